"""bench.py's roofline line without a GPU: the quote of a committed profile is attached only to the kernel instantiation and
the kernel sources it was measured on, the fraction is the stated flop model over the stated time, and the
profiles committed for the three bench configurations are those of the kernel sources in this tree."""
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from acgpathtracing_amd import _build, _native  # noqa: E402


def _args(config, **kw):
    pre = bench.PRESETS[config]
    a = types.SimpleNamespace(config=config, scene=pre[0], width=pre[3], height=pre[4], spp=bench.SPP_PER_LAUNCH, variant=-1,
                              blocks_per_cu=0, fuse=8, chunks=0, max_depth=pre[2], direct_lighting=pre[5], importance_sampling=pre[6])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _info(n_tris):
    return types.SimpleNamespace(n_tris=n_tris, half_node_bytes=32 * (n_tris - 1), node_bytes=64 * (n_tris - 1), tri_bytes=64 * n_tris)


@pytest.mark.parametrize("config", [0, 2, 3, 5])
def test_committed_profiles_belong_to_these_kernel_sources(config):
    summ, src = bench.pmc_summary(config)
    assert summ is not None, "no profiles/r*_c%d_summary.json" % config
    assert summ["kernel_source_hash"] == _build.kernel_source_hash(), \
        "%s was taken on other kernel sources: re-run tools/profile_bench.sh (bench.py would drop its quote)" % src
    lib = _native.hip()
    variant = 9 if config == 5 else 7
    kern = lib.pt_variant_kernel(variant, _native.MATH_FAST).decode()        # what bench.py --config N runs by default
    assert kern in summ["kernel_stats"]["name"], (kern, summ["kernel_stats"]["name"])
    assert "--no-ieee-leg" in summ["command"] and ("--config %d" % config) in summ["command"]
    assert summ["kernel_stats"]["calls"] >= 2 and summ["kernel_stats"]["avg_ms"] > 0


def test_profile_quote_is_tied_to_kernel_and_sources():
    summ, src = bench.pmc_summary(2)
    kern = summ["kernel_stats"]["name"]
    inst = kern[kern.index("k_render_pw"):kern.index("(ptd::")]
    h = summ["kernel_source_hash"]
    a = _args(2)
    rays = 3.68e9
    r = bench.roofline_block(a, _info(1264), 1, 8, [147.0, 147.2], rays, rays * 1.27, 8, "pw fp16 nodes", inst, h)
    assert r["traffic"] == summ["derived"]["hbm_bytes_per_launch"] and r["measured"]["source"] == src and "profile_dropped" not in r
    assert r["measured"]["lane_utilisation"] == summ["derived"]["valu_lane_utilisation"]
    # what binds, as measured: vector issue priced by opcode class, the texture-address path, live lanes — and which is larger
    iss = r["issue"]
    assert iss["valu_issue_busy_mix"] == summ["derived"]["valu_issue_busy_mix"] and iss["ta_busy"] == summ["derived"]["ta_busy_frac(256 TAs)"]
    assert 0.6 < iss["valu_issue_busy_mix"] < 1.0 and 0.6 < iss["ta_busy"] < 1.0 and 0.4 < iss["lane_utilisation"] < 0.6
    assert iss["valu_issue_busy_mix"] > summ["derived"]["valu_issue_busy_frac(2cyc/instr,1024 SIMDs)"]          # half- and quarter-rate instructions cost more than 2 cycles
    assert r["bound_measured"] == ("vector issue" if iss["valu_issue_busy_mix"] >= iss["ta_busy"] else "texture-address path")
    sh = summ["valu_class_shares"]
    assert abs(sh["full_rate"] + sh["half_rate"] + sh["quarter_rate"] - 1.0) < 1e-9 and 0.4 < sh["full_rate"] < 0.8 and 0.005 < sh["quarter_rate"] < 0.05
    # the stated model: 48 * ceil(log2 T) + 168 flops per traversed ray over the average kernel time, against 157.3 TFLOP/s
    flops = rays * (48 * 11 + 168)
    assert r["bound"] == "valu" and abs(r["achieved"] - flops / 0.1471 / 1e12) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 157.3) < 1e-12 and r["frac_counting_culled_rays"] > r["frac"] == r["frac_entering_scene"]
    # another instantiation (the IEEE twin), other sources, another command line: the quote drops out and says why
    twin = inst[:-2] + "0>"
    assert twin != inst
    r = bench.roofline_block(a, _info(1264), 1, 8, [161.0], rays, rays, 8, "pw fp16 nodes", twin, h)
    assert r["traffic"] is None and r["measured"] is None and "is not the" in r["profile_dropped"]["reason"]
    r = bench.roofline_block(a, _info(1264), 1, 8, [147.0], rays, rays, 8, "pw fp16 nodes", inst, "0123456789abcdef")
    assert r["traffic"] is None and "kernel sources" in r["profile_dropped"]["reason"]
    r = bench.roofline_block(_args(2, fuse=4), _info(1264), 1, 4, [75.0], rays / 2, rays / 2, 4, "pw fp16 nodes", inst, h)
    assert r["traffic"] is None and r["measured"] is None and "profile_dropped" not in r       # not the profiled command: nothing to quote
    # where the scene lives decides the roof: on chip -> vector flops; beyond the Infinity Cache -> HBM bytes of the node format that ran
    big = bench.roofline_block(_args(5, scene="x.obj"), _info(10_500_000), 1, 2, [167.0], 1.0e9, 1.0e9, 2, "pw fp16 nodes", inst, h)
    assert big["bound"] == "hbm" and big["unit"] == "GB/s" and big["scene_resident_in"] == "HBM" and big["algorithmic_bytes_per_ray_this_kernel"] == 32 * 24 + 48
    mid = bench.roofline_block(_args(5, scene="x.obj"), _info(1_310_732), 1, 2, [111.0], 1.0e9, 1.0e9, 2, "pw fp16 nodes", inst, h)
    assert mid["bound"] == "valu" and mid["scene_resident_in"] == "Infinity Cache"


def test_vector_instruction_pricing_is_pinned_to_the_microbenchmark():
    """tools/valu_mix.py: the cycles per opcode class come from profiles/r02_ubench_valu.txt (not from a constant in the tool), every
    opcode of the BVH loop lands in the class the microbenchmark measured it in, and the datasheet pricing is 2 / 4 / 8."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_mix
    costs = valu_mix.class_costs()
    assert 2.3 < costs["full"] < 2.9 and 4.0 < costs["half"] < 4.6 and 8.0 < costs["quarter"] < 8.7
    # re-derive one class mean from the file itself: the tool reads the table, it does not carry numbers
    rows = {}
    for ln in open(os.path.join(ROOT, "profiles", "r02_ubench_valu.txt")):
        p = ln.split()
        if len(p) >= 5 and p[0].startswith("v_"):
            try:
                rows[" ".join(p[:-4])] = float(p[-2])
            except ValueError:
                pass
    full = [rows[k] for k in ("v_add_f32", "v_mul_f32", "v_fma_f32", "v_and_b32", "v_add_u32", "v_sub_f32", "v_xor_b32", "v_mov_b32")]
    assert abs(sum(full) / len(full) - costs["full"]) < 1e-9
    for op, cls in (("v_fma_f32", "full"), ("v_sub_f32_e32", "full"), ("v_add_u32_e32", "full"), ("v_mov_b32_e32", "full"), ("v_and_b32_e32", "full"),
                    ("v_fma_mix_f32", "half"), ("v_alignbit_b32", "half"), ("v_max3_f32", "half"), ("v_min_f32_e32", "half"), ("v_cmp_le_f32_e64", "half"),
                    ("v_cndmask_b32_e64", "half"), ("v_lshl_add_u32", "half"), ("v_cvt_f32_f16_e32", "half"), ("v_rcp_f32_e32", "quarter"), ("v_sqrt_f32_e32", "quarter")):
        assert valu_mix.classify(op) == cls, op
    assert valu_mix.SPEC == {"full": 2.0, "half": 4.0, "quarter": 8.0}
    m = valu_mix.kernel_mix(_native.hip_library_path(), _native.hip().pt_variant_kernel(7, _native.MATH_FAST).decode())
    assert m["bvh_loop"]["n_valu"] > 200 and m["bvh_loop"]["by_class"]["half"] > m["bvh_loop"]["by_class"]["quarter"]
    assert m["class_cycles"] == costs and 2.5 < m["cycles_per_valu_spec"] < 3.5


def test_committed_bench_lines_are_of_these_sources():
    """profiles/r04_bench_c*.json: the JSON lines quoted in DESIGN.md / BASELINE.md carry the hash they were measured on."""
    for n in ("c0", "c2", "c3", "c5", "default_steps20"):
        j = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_%s.json" % n)))
        r = j["roofline"]
        assert r["kernel_source_hash"] == _build.kernel_source_hash(), n
        assert r["traffic"] is not None and "profile_dropped" not in r, n
        assert j["config"]["math"].startswith("fast") and j["other_math_mode"]["math"] == "ieee"
        assert 0.0 < r["frac"] < 1.0 and j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
        assert "issue" in r and r["bound_measured"] in ("vector issue", "texture-address path")
        assert j["Mray_per_s_entering_scene"] == j["config"]["Mray_per_s_entering_scene"] <= j["value"]
