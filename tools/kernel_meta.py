#!/usr/bin/env python3
"""Register, spill, LDS and scratch figures of every kernel in a built library, from the code objects' metadata notes
(no GPU needed).  usage: tools/kernel_meta.py [lib.so] [name filter]
tests/test_abi.py uses kernel_table() to hold the default render kernel to its occupancy budget."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("name", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size")


def kernel_table(lib):
    """One dict per kernel: demangled `name`, and the integer fields of FIELDS."""
    notes = ""
    with tempfile.TemporaryDirectory() as td:
        # the device code objects sit in the fat-binary section of the host library, one bundle per translation unit
        fat = os.path.join(td, "fat.bin")
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("no .hip_fatbin section in %s: %s" % (lib, r.stderr))
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for n, a in enumerate(starts):
            b = starts[n + 1] if n + 1 < len(starts) else len(blob)
            one, out = os.path.join(td, "b%d.bin" % n), os.path.join(td, "b%d.co" % n)
            open(one, "wb").write(blob[a:b])
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + one,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("unbundle failed: " + r.stderr)
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", out], capture_output=True, text=True).stdout
    rows, kern = [], None
    for ln in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":            # first key of a kernel's record
            kern = {}
            rows.append(kern)
        elif kern is not None and k in FIELDS:
            kern[k] = v.strip("'") if k == "name" else int(v)
    mangled = [k.get("name", "?") for k in rows]
    demangled = subprocess.run(["c++filt"] + mangled, capture_output=True, text=True).stdout.splitlines() if mangled else []
    for k, d in zip(rows, demangled):
        k["name"] = d.strip()
    return rows


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "acgpathtracing_amd", "libacgpt_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in kernel_table(lib):
        if flt and flt not in k["name"]:
            continue
        print("%-110s vgpr %3s (spilled %s) sgpr %3s (spilled %s) lds %6s scratch %s" % (k["name"][:110], k.get("vgpr_count"), k.get("vgpr_spill_count"),
              k.get("sgpr_count"), k.get("sgpr_spill_count"), k.get("group_segment_fixed_size"), k.get("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
