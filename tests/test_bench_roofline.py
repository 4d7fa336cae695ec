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
    scene = {2: "cornell_box_diffuse.obj", 3: "cornell_box.obj", 5: "stress_1m.obj"}[config]
    a = types.SimpleNamespace(config=config, scene=scene, width=bench.WIDTH, height=bench.HEIGHT, spp=bench.SPP_PER_LAUNCH, variant=-1,
                              blocks_per_cu=0, fuse=8, chunks=0)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _info(n_tris):
    return types.SimpleNamespace(n_tris=n_tris, half_node_bytes=32 * (n_tris - 1), node_bytes=64 * (n_tris - 1), tri_bytes=64 * n_tris)


@pytest.mark.parametrize("config", [2, 3, 5])
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


def test_committed_bench_lines_are_of_these_sources():
    """profiles/r03_bench_c*.json: the JSON lines quoted in DESIGN.md / BASELINE.md carry the hash they were measured on."""
    for n in ("c2", "c3", "c5", "default_steps20"):
        j = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_%s.json" % n)))
        r = j["roofline"]
        assert r["kernel_source_hash"] == _build.kernel_source_hash(), n
        assert r["traffic"] is not None and "profile_dropped" not in r, n
        assert j["config"]["math"].startswith("fast") and j["other_math_mode"]["math"] == "ieee"
        assert 0.0 < r["frac"] < 1.0 and j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
