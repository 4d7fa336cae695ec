#!/usr/bin/env python3
"""Price the vector instructions of a render kernel with the MEASURED issue cost of their opcode class.

tools/summarize_prof.py's `valu_issue_busy_frac(2cyc/instr)` prices every vector instruction at the 2 cycles of a full-rate
wave64 op.  profiles/r02_ubench_valu.txt (tools/ubench_valu.hip, MI355X) measures three classes at >= 2 waves per SIMD:
    full rate     2.5 cycles   add / sub / mul / fma / mac / and / or / xor / not / add_u32 / sub_u32 / mov
    half rate     4.2 cycles   min / max / min3 / max3 / med3 / compares / cndmask / shifts / conversions / bfi / bfe / perm /
                               alignbit / lshl_or / lshl_add / and_or / mad_u32 / mul_lo / 64-bit adds / v_fma_mix_f32 / packed ops
    quarter rate  8.2 cycles   rcp / rsq / sqrt / sin / cos / exp / log
This tool disassembles the kernel from the built library (no GPU needed), classifies every vector instruction of (a) its BVH loop
— node visits and triangle tests: from the first fp16 slab plane (v_fma_mix_f32; fp32-node kernels: the first node gather) to the
last triangle test's division — and (b) the rest (shade / regenerate phase, queue, prologue), and prints the mean cost per vector
instruction of each.  The dynamic mix is a blend of the two; tools/summarize_prof.py prices SQ_INSTS_VALU with the blend 0.7 / 0.3
(the phases' share of a wave's time, profiles/r03_wave_timeline_fast.txt: the two means differ by a few percent, so the weights
hardly matter) -> `valu_issue_busy_mix`; the same count at the datasheet's 2 / 4 / 8 cycles -> `valu_issue_busy_spec`.
tests/test_bench_roofline.py pins the class costs to the microbenchmark file.

usage: tools/valu_mix.py [lib.so] [kernel substring]      prints one JSON object"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
UBENCH = os.path.join(ROOT, "profiles", "r02_ubench_valu.txt")

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_mov_b32", "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_xnor_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"}
# representative rows of the microbenchmark file for each class (4 waves per SIMD column)
ROWS = {"full": ["v_add_f32", "v_mul_f32", "v_fma_f32", "v_and_b32", "v_add_u32", "v_sub_f32", "v_xor_b32", "v_mov_b32"],
        "half": ["v_fma_mix_f32 (f16 lo src0)", "v_fma_mix_f32 (f16 hi src0)", "v_alignbit_b32 (vgpr shift)", "v_max_f32", "v_max3_f32", "v_min3_f32", "v_lshlrev_b32",
                 "v_cmp_lt_f32 sgpr pair", "v_cndmask (sgpr pair, set)", "v_bfi_b32", "v_cvt_f32_u32", "v_lshl_or_b32", "v_mad_u32_u24", "v_mul_lo_u32"],
        "quarter": ["v_rcp_f32"]}


def class_costs(path=UBENCH, column=2):
    """Mean SIMD cycles per wave64 instruction of each class, from the microbenchmark's table (column 2 = 4 waves per SIMD)."""
    table = {}
    for ln in open(path):
        m = re.match(r"^(\S.*?)\s{2,}([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", ln)
        if m:
            table[m.group(1).strip()] = [float(m.group(k)) for k in range(2, 6)]
    out = {}
    for cls, rows in ROWS.items():
        vals = [table[r][column] for r in rows if r in table]
        if not vals:
            raise RuntimeError("no rows of class %s in %s" % (cls, path))
        out[cls] = sum(vals) / len(vals)
    return out


def classify(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op in FULL:
        return "full"
    if op in QUARTER:
        return "quarter"
    return "half"


def disassemble(lib, kernel_substr):
    """Instruction mnemonics of the first kernel whose demangled name contains kernel_substr, in program order."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True, capture_output=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for n, a in enumerate(starts):
            b = starts[n + 1] if n + 1 < len(starts) else len(blob)
            one, co = os.path.join(td, "b%d.bin" % n), os.path.join(td, "b%d.co" % n)
            open(one, "wb").write(blob[a:b])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + one,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
            names = re.findall(r"^[0-9a-f]+ <(\S+)>:$", text, flags=re.M)
            if not names:
                continue
            dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
            for mangled, d in zip(names, dem):
                if kernel_substr in d and "(" in d:
                    body = text.split("<%s>:" % mangled, 1)[1]
                    body = re.split(r"^\n?[0-9a-f]+ <", body, maxsplit=1, flags=re.M)[0]
                    ops = []
                    for ln in body.splitlines():
                        m = re.match(r"^\s+([a-z_0-9]+)\b", ln)
                        if m:
                            ops.append(m.group(1))
                    return d.strip(), ops
    raise RuntimeError("kernel %r not found in %s" % (kernel_substr, lib))


def mix(ops, costs):
    count = {"full": 0, "half": 0, "quarter": 0}
    for op in ops:
        if op.startswith("v_") and not op.startswith("v_readlane") and not op.startswith("v_writelane") and not op.startswith("v_readfirstlane"):
            count[classify(op)] += 1
    n = sum(count.values())
    mean = sum(count[c] * costs[c] for c in count) / max(1, n)
    return {"n_valu": n, "by_class": count, "mean_cycles": mean}


SPEC = {"full": 2.0, "half": 4.0, "quarter": 8.0}      # MI355X_MICROARCH.md: v_fma_f32 wave64 2 cycles (32 lanes per cycle); half / quarter rate classes
LOOP_SHARE = 0.7                                          # BVH loop's share of a wave's time (profiles/r03_wave_timeline_fast.txt: 70.5 %)


def kernel_mix(lib, kernel_substr):
    costs = class_costs()
    name, ops = disassemble(lib, kernel_substr)
    first = [i for i, op in enumerate(ops) if op == "v_fma_mix_f32"] or [i for i, op in enumerate(ops) if op == "global_load_dwordx4"]
    last = [i for i, op in enumerate(ops) if op.startswith("v_div_fixup_f32")]
    a = max(0, first[0] - 12) if first else 0
    b = min(len(ops), last[-1] + 30) if last else len(ops)
    loop, rest, whole = mix(ops[a:b], costs), mix(ops[:a] + ops[b:], costs), mix(ops, costs)
    def blend(c):
        l = sum(loop["by_class"][k] * c[k] for k in c) / max(1, loop["n_valu"])
        r = sum(rest["by_class"][k] * c[k] for k in c) / max(1, rest["n_valu"])
        return LOOP_SHARE * l + (1.0 - LOOP_SHARE) * r
    return {"kernel": name, "class_cycles": costs, "class_cycles_source": os.path.relpath(UBENCH, ROOT) + " (4 waves per SIMD column, class means)",
            "class_cycles_spec": SPEC, "bvh_loop": loop, "rest_of_kernel": rest, "whole_kernel": whole,
            "cycles_per_valu_priced": blend(costs), "cycles_per_valu_spec": blend(SPEC),
            "rule": "%.1f x BVH-loop mean + %.1f x mean of the rest (the phases' share of the wave-time)" % (LOOP_SHARE, 1.0 - LOOP_SHARE)}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "acgpathtracing_amd", "libacgpt_hip.so")
    kern = sys.argv[2] if len(sys.argv) > 2 else "k_render_pw<40, 16, 11, 256, 5, false, 0, 6, 2, false, 0, 0, 1>"
    print(json.dumps(kernel_mix(lib, kern), indent=1))
