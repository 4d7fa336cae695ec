#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output directory: kernel stats + per-launch PMC averages for
the render kernel, plus the derived ratios used in DESIGN.md."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    d = sys.argv[1]
    kern = sys.argv[2] if len(sys.argv) > 2 else "k_render"
    out = {"dir": os.path.basename(d.rstrip("/")), "kernel_filter": kern}
    if len(sys.argv) > 3:
        out["command"] = sys.argv[3]
    if len(sys.argv) > 4:
        out["steps_per_kernel_launch"] = int(sys.argv[4])
    try:        # what the profile was taken on: bench.py quotes it only for a library built from the same kernel sources
        from acgpathtracing_amd import _build
        out["kernel_source_hash"] = _build.kernel_source_hash()
    except Exception as e:
        out["kernel_source_hash"] = "unknown (%s)" % e
    st = os.path.join(d, "trace", "trace_kernel_stats.csv")
    if os.path.exists(st):
        for r in csv.DictReader(open(st)):
            if kern in r["Name"]:
                out["kernel_stats"] = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                       "min_ms": float(r["MinNs"]) / 1e6, "max_ms": float(r["MaxNs"]) / 1e6, "pct": float(r["Percentage"])}
                break
    pmc = {}
    for f in sorted(glob.glob(os.path.join(d, "pmc*", "pmc_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {"vgpr": int(r["VGPR_Count"]), "sgpr": int(r["SGPR_Count"]), "lds": int(r["LDS_Block_Size"]),
                        "grid": int(r["Grid_Size"]), "wg": int(r["Workgroup_Size"])}
        for k, v in agg.items():
            pmc[k] = sum(v) / len(v)
        if meta:
            out["dispatch"] = meta
    out["pmc_per_launch"] = pmc
    der = {}
    g = pmc.get
    if g("GRBM_GUI_ACTIVE"):
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        der["gpu_cycles"] = cyc
        if "kernel_stats" in out:
            der["clock_ghz"] = cyc / (out["kernel_stats"]["avg_ms"] * 1e6)
        if g("SQ_INSTS_VALU"):
            der["valu_issue_busy_frac(2cyc/instr,1024 SIMDs)"] = g("SQ_INSTS_VALU") * 2.0 / (cyc * 1024)
            # the same count priced by opcode class (tools/valu_mix.py: static mix of the kernel that ran, BVH loop and the rest blended by
            # their share of the wave-time): _mix at the datasheet's 2 / 4 / 8 cycles for full / half / quarter rate, _ubench at the
            # cycles the microbenchmark measures for each class in isolation (an upper bound: it can exceed 1, classes overlap in a mixed stream)
            try:
                import valu_mix
                kname = out.get("kernel_stats", {}).get("name", "")
                inst = kname[kname.index("k_render"):kname.index("(ptd::")] if "k_render" in kname and "(ptd::" in kname else kern
                lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "acgpathtracing_amd", os.environ.get("ACGPT_HIP_LIB", "libacgpt_hip.so"))
                vm = valu_mix.kernel_mix(lib, inst)
                out["valu_mix"] = vm
                spec, priced = vm["cycles_per_valu_spec"], vm["cycles_per_valu_priced"]
                if g("SQ_ACTIVE_INST_VALU2") is not None and g("SQ_INSTS_VALU_TRANS_F32") is not None:
                    # the DYNAMIC class shares, from counters calibrated with the microbenchmark (profiles/r04_valu_class_counters.txt):
                    # SQ_ACTIVE_INST_VALU2 advances 0.42 per full-rate instruction and not at all for the others; TRANS_F32 counts the quarter-rate ones
                    n = g("SQ_INSTS_VALU")
                    full = min(1.0, g("SQ_ACTIVE_INST_VALU2") / 0.42 / n)
                    quarter = g("SQ_INSTS_VALU_TRANS_F32") / n
                    half = max(0.0, 1.0 - full - quarter)
                    cs, cu = vm["class_cycles_spec"], vm["class_cycles"]
                    spec = full * cs["full"] + half * cs["half"] + quarter * cs["quarter"]
                    priced = full * cu["full"] + half * cu["half"] + quarter * cu["quarter"]
                    out["valu_class_shares"] = {"full_rate": full, "half_rate": half, "quarter_rate": quarter, "source": "SQ_ACTIVE_INST_VALU2 / 0.42, SQ_INSTS_VALU_TRANS_F32 (dynamic; calibration: profiles/r04_valu_class_counters.txt)",
                                                "static_mix_would_give": {"cycles_per_valu_spec": vm["cycles_per_valu_spec"]}}
                der["valu_issue_busy_mix"] = g("SQ_INSTS_VALU") * spec / (cyc * 1024)
                der["valu_issue_busy_ubench"] = g("SQ_INSTS_VALU") * priced / (cyc * 1024)
            except Exception as e:
                out["valu_mix"] = {"error": str(e)}
        if g("TA_TA_BUSY_sum"):
            der["ta_busy_frac(256 TAs)"] = g("TA_TA_BUSY_sum") / (cyc * 256)
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        der["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
        der["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
        der["l1_miss_per_access"] = g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")
    if g("FETCH_SIZE") is not None:
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE is in KB and under-reports wide coalesced streams 2x on gfx950
        der["hbm_read_bytes(FETCH_SIZE*1024*2)"] = g("FETCH_SIZE") * 1024 * 2
    if g("WRITE_SIZE") is not None:
        der["hbm_write_bytes(WRITE_SIZE*1024)"] = g("WRITE_SIZE") * 1024
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        der["hbm_bytes_per_launch"] = g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE") * 1024
    out["derived"] = der
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
