#!/usr/bin/env python3
"""Register, spill, LDS and scratch figures of every kernel in a built library, from the code object's metadata notes.
usage: tools/kernel_meta.py [lib.so] [name filter]"""
import os
import re
import subprocess
import sys
import tempfile

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "acgpathtracing_amd", "libacgpt_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
llvm = "/opt/rocm/lib/llvm/bin"
notes = ""
with tempfile.TemporaryDirectory() as td:
    # the device code objects sit in the fat-binary section of the host library, one bundle per translation unit
    fat = os.path.join(td, "fat.bin")
    r = subprocess.run([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("no .hip_fatbin section in %s: %s" % (lib, r.stderr))
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    for n, a in enumerate(starts):
        b = starts[n + 1] if n + 1 < len(starts) else len(blob)
        one, out = os.path.join(td, "b%d.bin" % n), os.path.join(td, "b%d.co" % n)
        open(one, "wb").write(blob[a:b])
        r = subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + one,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit("unbundle failed: " + r.stderr)
        notes += subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", out], capture_output=True, text=True).stdout
kern = None
rows = []
for ln in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count":
        kern = {"agpr": v}
        rows.append(kern)
    elif kern is not None and k in ("name", "sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size"):
        kern[k] = v.strip("'")
for k in rows:
    name = subprocess.run(["c++filt", k.get("name", "?")], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print("%-110s vgpr %3s (spilled %s) sgpr %3s (spilled %s) lds %6s scratch %s" % (name[:110], k.get("vgpr_count"), k.get("vgpr_spill_count"), k.get("sgpr_count"),
          k.get("sgpr_spill_count"), k.get("group_segment_fixed_size"), k.get("private_segment_fixed_size")))
