"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (task statement, item 3): bit-exact for integer / index work — PRNG-driven pixel seeds, Morton
codes, sort order, hit triangle index — and, because the ray/triangle arithmetic is specified down
to the fused operations, bit-exact hit distances too.  Floating-point image parity is stated as
BASELINE.json's north_star does: per-channel L2 (MSE over clamped linear radiance) < 1e-3; the GPU's
sinf/cosf/acosf/powf are ROCm's, the oracle's are glibc's, so single paths can flip.
"""
import ctypes as C
import os

import numpy as np
import pytest

import acgpathtracing_amd as pt
from acgpathtracing_amd import _native
from scene_utils import adversarial_rays, copy_params, flip_report, image_mse, image_mse_trimmed, make_params, random_rays, scene_arrays

pytestmark = pytest.mark.gpu

# north_star's bar is image L2 error < 1e-3; what is measured is 1e-13 ... 1e-9 (a handful of paths flip where ROCm's and
# glibc's sinf / cosf / acosf round differently; worst of a 30-camera soak 6e-7, profiles/r01_soak_images.txt).  The
# tests hold the measured level, not the headline bar: a few-percent error in any shading term is ~1e-4 and fails.
MSE_TOL = 1e-6
# PT_MATH_FAST against the (IEEE-level) oracle in UNIFORM-hemisphere mode: every sampled direction differs in its last bits, so
# each of the rare rays that skim their own wall (scene_utils.image_mse_trimmed) is a coin flip — a few pixels carry one path
# more or less.  Round 3 trimmed the 0.1 % worst pixels away and held the rest; since round 4 the flips are COUNTED and each is
# held to the weight of one path (assert_uniform_mode_fast below; levels measured per configuration and frame seed in
# profiles/r04_flip_levels.txt):
#   flipped paths per traced path: 1.3e-5 ... 2.2e-5 with direct lighting off (diffuse and full scene alike, 8 ... 256 spp, 64^2 ... 512^2
#   pixels), 2.0e-4 ... 3.1e-4 with direct lighting on (a grazing bounce ray AND its shadow ray start at the same rounded hit point)
FLIP_RATE = {False: 2.2e-5, True: 3.1e-4}          # by useDirectLighting: the measured upper ends
MSE_REST_TOL = 5e-9                                # the pixels without a flip: measured 1e-17 ... 5e-10
PATH_RADIANCE_MAX = 20.0                           # a path's radiance is at most Ke (1 + Kd) = 10 x 1.78, times a throughput <= 1


def assert_uniform_mode_fast(facc, ref, spp, direct_lighting, what=""):
    """The fast-math render of a uniform-hemisphere configuration against the oracle: (i) the pixels that differ by more than 1e-3 of
    full scale are counted, not trimmed, and their count is what the measured flip rate predicts for pixels x spp paths (twice
    the expectation plus five standard deviations of a Poisson count: a shading term that is off by a percent puts EVERY pixel
    out and fails here); (ii) the whole-image MSE is explained by those pixels at the weight of one path each — a flipped path
    moves a pixel by at most min(1, L / spp) per channel; (iii) all other pixels agree to rounding; (iv) and so do their means."""
    r = flip_report(facc, ref)
    pixels = facc.shape[0] * facc.shape[1]
    n_exp = FLIP_RATE[bool(direct_lighting)] * pixels * spp
    n_max = 2.0 * n_exp + 5.0 * n_exp ** 0.5 + 3.0
    w = min(1.0, (PATH_RADIANCE_MAX / spp) ** 2)      # squared error of a pixel that gained or lost one path, mean over the channels, clamped radiance
    msg = (what, r, n_max)
    assert r["n_out"] <= n_max, msg
    assert r["mse"] <= r["n_out"] * w / pixels + MSE_REST_TOL, msg
    assert r["mse_rest"] < MSE_REST_TOL, msg
    assert abs(r["rest_mean_a"] - r["rest_mean_b"]) <= 1e-4 * r["rest_mean_b"], msg
    assert r["mse"] < 1e-3, msg                     # north_star's own bar, whatever the model says
    return r
SAME_BITS_MIN = 0.70    # fraction of pixels whose fp32 accumulation is bit-identical to the oracle's (same summation order)
_DEFAULT_VARIANT = -1   # pt_set_tuning: chosen per scene (fp16 nodes for these scenes)
SCENE_FULL = pt.SCENES + "/cornell_box.obj"
SCENE_DIFFUSE = pt.SCENES + "/cornell_box_diffuse.obj"


import contextlib


@contextlib.contextmanager
def _math(state, mode):
    """The fixtures' contexts run in "ieee" mode (conftest.py); a block that wants the library's default arithmetic
    ("fast": pt_set_math_mode, include/acgpt.h) switches for its duration."""
    pt.setMathMode(state, mode)
    try:
        yield
    finally:
        pt.setMathMode(state, "ieee")


def _gpu_render(state, p, frames=1, out_buffer=None, fuse=1, first_frame=0):
    """Launch `frames` sub-launches through LaunchCurrentFrame, `fuse` of them per kernel launch; returns
    (accum, fb, [stats])."""
    state.params.width, state.params.height = p.width, p.height
    keep_accum, keep_handle = state.params.accumulationBuffer, state.params.handle
    C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
    state.params.accumulationBuffer, state.params.handle = keep_accum, keep_handle
    state.refreshAccumulationBuffer = True
    pt.updateState(out_buffer, state)
    if out_buffer is None:
        out_buffer = pt.OutputBuffer(pt.OutputBufferType.DEVICE, p.width, p.height, state)
    stats = []
    f = 0
    while f < frames:
        n = min(fuse, frames - f)
        state.params.currentFrameIdx = first_frame + f
        pt.LaunchCurrentFrame(out_buffer, state, n)
        stats.append(pt.getStats(state))
        f += n
    acc = pt.readAccumulation(state)
    fb = out_buffer.getHostPointer().copy()
    out_buffer.free()
    return acc, fb, stats


@pytest.fixture(scope="module")
def full(gpu_state_factory, oracle):
    state, obj = gpu_state_factory(SCENE_FULL, width=64, height=64)
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    return state, obj, sc


@pytest.fixture(scope="module")
def diffuse(gpu_state_factory, oracle):
    state, obj = gpu_state_factory(SCENE_DIFFUSE, width=64, height=64)
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    return state, obj, sc


@pytest.fixture(params=["ieee", "fast"])
def both_modes(request, full, diffuse):
    """For the tests that compare the GPU with itself (scheduling, partition, batches: bit-identical whatever the arithmetic):
    run them at the oracle's level and in the library's default math mode."""
    for st in (full[0], diffuse[0]):
        pt.setMathMode(st, request.param)
    yield request.param
    for st in (full[0], diffuse[0]):
        pt.setMathMode(st, "ieee")


def test_library_is_the_hip_one():
    L = _native.hip()
    assert L.pt_abi_version() == 4
    assert _native.hip_library_path().endswith("libacgpt_hip.so")


def test_morton_sort_bit_exact(full):
    """Integer work: 30-bit Morton codes and the stable LSD radix sort, restated in numpy."""
    state, obj, _ = full
    v, idx = scene_arrays(obj)
    T = idx.shape[0]
    codes = np.zeros(T, np.uint32); prims = np.zeros(T, np.uint32)
    assert _native.hip().pt_read_morton(state.context, codes.ctypes.data, prims.ctypes.data) == 0
    # restatement of k_prepare / k_morton (fp32 throughout)
    tri = v[idx][:, :, :3].astype(np.float32).copy()
    # the corners the builder boxes: v0, v0 + e1, v0 + e2 with the record's rounded edges e = v - v0 (lbvh_build.hip record_aabb)
    tri[:, 1] = (tri[:, 0] + (tri[:, 1] - tri[:, 0]).astype(np.float32)).astype(np.float32)
    tri[:, 2] = (tri[:, 0] + (tri[:, 2] - tri[:, 0]).astype(np.float32)).astype(np.float32)
    lo = tri.min(axis=1); hi = tri.max(axis=1)
    pad_abs = np.float32(max(1.0, float(np.abs(v[:, :3]).max()))) * np.float32(1.0 / 524288.0)
    pad = np.maximum(np.float32(1e-5) * np.maximum(np.float32(1.0), np.maximum(np.abs(lo), np.abs(hi))), pad_abs).astype(np.float32)
    lo = (lo - pad).astype(np.float32); hi = (hi + pad).astype(np.float32)
    slo = lo.min(axis=0); shi = hi.max(axis=0)
    c = (np.float32(0.5) * (lo + hi)).astype(np.float32)
    ext = (shi - slo).astype(np.float32)
    u = ((c - slo).astype(np.float32) / ext).astype(np.float32)
    q = np.minimum(np.maximum((u * np.float32(1024.0)).astype(np.float32), np.float32(0.0)), np.float32(1023.0)).astype(np.uint32)

    def expand(x):
        x = (x * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        x = (x * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        x = (x * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        x = (x * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return x
    with np.errstate(over="ignore"):
        expect = (expand(q[:, 0]) << np.uint32(2)) | (expand(q[:, 1]) << np.uint32(1)) | expand(q[:, 2])
    order = np.argsort(expect, kind="stable")
    assert np.array_equal(prims, order.astype(np.uint32))
    assert np.array_equal(codes, expect[order])
    info = pt.getBvhInfo(state)
    assert info.n_tris == T and info.n_nodes == T - 1
    assert np.allclose(np.array(info.scene_lo), slo) and np.allclose(np.array(info.scene_hi), shi)
    assert 8 <= info.stack_entries <= 64 and info.max_depth < info.stack_entries


def test_a_scene_keeps_one_node_array(gpu_state_factory):
    """the builder leaves ONE node array on the device — the one the chosen kernel reads (fp16: 32 B per node)
    — next to the triangle and shading records; the fp32 nodes come back on first use (a ray query, an fp32 kernel variant)
    from the topology and the records, and render the same bits as the fp16 kernel."""
    L = _native.hip()
    state, obj = gpu_state_factory(SCENE_FULL, width=48, height=48, direct_lighting=True, importance_sampling=True, spp=4, max_depth=4)
    T = obj.getIndexBuffer().size // 3
    info = pt.getBvhInfo(state)
    assert info.half_node_bytes == 32 * (T - 1) and info.node_bytes == 64 * (T - 1) and info.tri_bytes == 48 * T
    lean = info.half_node_bytes + info.tri_bytes + 16 * T
    assert info.device_bytes == lean, "after pt_set_scene: fp16 nodes + triangle records + shading records, nothing else"
    assert info.device_bytes <= 1.3 * (info.half_node_bytes + info.tri_bytes)      # bench.py's scene_bytes
    p = make_params(48, 48, 4, 4, True, True)
    want, _, st = _gpu_render(state, p)
    assert L.pt_variant_name(int(st[0].variant)).find(b"fp16") >= 0
    assert pt.getBvhInfo(state).device_bytes == lean, "rendering with the chosen kernel allocates no scene array"
    # a ray query walks the fp32 nodes: rebuilt now, counted from now on
    rays = random_rays(2000, 7)
    t = np.zeros(2000, np.float32); prim = np.zeros(2000, np.uint32)
    assert L.pt_trace_closest(state.context, rays.ctypes.data, 2000, t.ctypes.data, prim.ctypes.data) == 0
    assert pt.getBvhInfo(state).device_bytes == lean + info.node_bytes
    # ... and an fp32 kernel variant on those rebuilt nodes renders the fp16 kernel's bits
    try:
        assert L.pt_set_tuning(state.context, 0, 1) == 0
        got, _, st = _gpu_render(state, p)
        assert int(st[0].variant) == 1 and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    finally:
        assert L.pt_set_tuning(state.context, 0, -1) == 0
    # a scene that is set while an fp32 variant is selected keeps the fp32 nodes instead
    state2, _ = gpu_state_factory(SCENE_FULL, width=48, height=48, direct_lighting=True, importance_sampling=True, spp=4, max_depth=4)
    assert L.pt_set_tuning(state2.context, 0, 1) == 0
    pt.buildTheAccelarationStructure(state2, obj)
    i2 = pt.getBvhInfo(state2)
    assert i2.device_bytes == i2.node_bytes + i2.tri_bytes + 16 * T
    got2, _, _ = _gpu_render(state2, p)
    assert np.array_equal(got2.view(np.uint32), want.view(np.uint32))
    assert L.pt_set_tuning(state2.context, 0, -1) == 0
    got3, _, _ = _gpu_render(state2, p)          # back to the fp16 kernel: its nodes are derived from the fp32 ones on first use
    assert np.array_equal(got3.view(np.uint32), want.view(np.uint32))


def test_trace_closest_bit_exact(full):
    """Closest hit vs brute force over every triangle: same triangle index, same t, bit for bit."""
    state, obj, sc = full
    v, idx = scene_arrays(obj)
    rays = np.concatenate([random_rays(150000, 1), adversarial_rays(v, idx, 2),
                           random_rays(20000, 3, tmin=5.0, tmax=300.0)])
    n = rays.shape[0]
    t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32)
    assert _native.hip().pt_trace_closest(state.context, rays.ctypes.data, n, t.ctypes.data, prim.ctypes.data) == 0
    t_ref, prim_ref = sc.trace_closest(rays, use_bvh=False)
    assert np.array_equal(prim, prim_ref), "hit triangle differs on %d rays" % int((prim != prim_ref).sum())
    assert np.array_equal(t.view(np.uint32), t_ref.view(np.uint32)), "hit distance differs"
    assert (prim != 0xFFFFFFFF).mean() > 0.5


def test_karras_tree_gives_the_same_answers(gpu_state_factory, oracle):
    """Both hierarchy builders (Karras radix tree, PLOC) sit behind the same triangle test: identical hits."""
    state, obj = gpu_state_factory(SCENE_FULL, width=64, height=64, build_mode=0)
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    v, idx = scene_arrays(obj)
    rays = np.concatenate([random_rays(60000, 31), adversarial_rays(v, idx, 32)])
    n = rays.shape[0]
    t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32); hit = np.zeros(n, np.uint8)
    L = _native.hip()
    assert L.pt_trace_closest(state.context, rays.ctypes.data, n, t.ctypes.data, prim.ctypes.data) == 0
    assert L.pt_trace_any(state.context, rays.ctypes.data, n, hit.ctypes.data) == 0
    t_ref, prim_ref = sc.trace_closest(rays, use_bvh=False)
    assert np.array_equal(prim, prim_ref) and np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    assert np.array_equal(hit, sc.trace_any(rays, use_bvh=False))
    info = pt.getBvhInfo(state)
    assert info.n_nodes == idx.shape[0] - 1 and info.max_depth < info.stack_entries
    p = make_params(96, 64, 4, 6, True, True)
    acc, _, _ = _gpu_render(state, p)
    ref, _, _, _ = sc.render(copy_params(p), use_bvh=True)
    assert image_mse(acc, ref) < MSE_TOL


def test_every_tree_builder_renders_the_same_bits(gpu_state_factory):
    """Build mode 0 (Karras), 1 (PLOC), 2 (PLOC + insertion-based optimisation on the host, the default for small scenes): boxes only
    prune, so image bits and ray counters are the same whatever the tree; the optimised tree is not deeper than the lane stacks allow
    and has the smaller sum of inner-node areas to show for its 12 ms (fewer BVH-loop trips of the stats kernel)."""
    L = _native.hip()
    p = make_params(160, 96, 8, 8, True, True)
    imgs, trips = {}, {}
    for mode in (1, 2, 0):
        state, obj = gpu_state_factory(SCENE_FULL, width=160, height=96, build_mode=mode)
        info = pt.getBvhInfo(state)
        assert info.max_depth < info.stack_entries <= 28, (mode, info.max_depth, info.stack_entries)
        acc, fb, st = _gpu_render(state, p)
        imgs[mode] = (acc, fb, (int(st[0].radiance_rays), int(st[0].shadow_rays), int(st[0].paths)))
        assert L.pt_set_tuning(state.context, 0, 6) == 0          # the stats twin of the default loop: trips through the BVH loop
        _, _, st6 = _gpu_render(state, p)
        trips[mode] = int(st6[0].trav_wave_steps)
        assert L.pt_set_tuning(state.context, 0, _DEFAULT_VARIANT) == 0
    for mode in (2, 0):
        assert np.array_equal(imgs[mode][0].view(np.uint32), imgs[1][0].view(np.uint32)) and np.array_equal(imgs[mode][1], imgs[1][1]), mode
        assert imgs[mode][2] == imgs[1][2], mode
    print("BVH-loop trips: Karras %d, PLOC %d, PLOC + insertion-based optimisation %d" % (trips[0], trips[1], trips[2]))
    assert trips[2] < trips[1] < trips[0]


def test_trace_any_bit_exact(full):
    state, obj, sc = full
    v, idx = scene_arrays(obj)
    rays = np.concatenate([random_rays(100000, 4, tmin=0.01, tmax=250.0), adversarial_rays(v, idx, 5)])
    n = rays.shape[0]
    hit = np.zeros(n, np.uint8)
    assert _native.hip().pt_trace_any(state.context, rays.ctypes.data, n, hit.ctypes.data) == 0
    ref = sc.trace_any(rays, use_bvh=False)
    assert np.array_equal(hit, ref)
    assert 0.05 < hit.mean() < 0.95


def test_render_config1_diffuse(diffuse):
    """BASELINE config 1: 256x256, 16 spp, 3 bounces, diffuse-only, IS off, DL off."""
    state, obj, sc = diffuse
    p = make_params(256, 256, 16, 3, False, False)
    acc, fb, stats = _gpu_render(state, p)
    ref_acc, ref_fb, ref_stats, _ = sc.render(copy_params(p), use_bvh=True)
    mse = image_mse(acc, ref_acc)
    assert mse < MSE_TOL, mse
    same = np.all(acc.view(np.uint32) == ref_acc.view(np.uint32), axis=-1).mean()
    assert same > SAME_BITS_MIN, "only %.3f of the pixels are bit-identical" % same
    assert np.all(acc[..., 3] == 1.0)
    assert (np.abs(fb.astype(int) - ref_fb.astype(int)) <= 1).mean() > 0.995
    s = stats[0]
    assert s.paths == 256 * 256 * 16 and s.pixels == 256 * 256 and s.shadow_rays == 0 and s.math_mode == _native.MATH_IEEE
    assert abs(int(s.radiance_rays) - ref_stats["radiance_rays"]) <= 1e-3 * ref_stats["radiance_rays"]
    # the library's default arithmetic (the reference's --use_fast_math level): the same image by tolerance, not by bits
    with _math(state, "fast"):
        facc, ffb, fstats = _gpu_render(state, p)
    f = fstats[0]
    assert f.math_mode == _native.MATH_FAST and f.paths == s.paths and f.pixels == s.pixels and f.shadow_rays == 0
    # importance sampling is off here: uniform-hemisphere mode — flips counted and weighed, see assert_uniform_mode_fast
    rep = assert_uniform_mode_fast(facc, ref_acc, 16, False, "config 1")
    assert rep["mse"] < 2.5e-4 and np.all(facc[..., 3] == 1.0)       # measured 8.2e-5 ... 1.1e-4 over four frame seeds
    assert (np.abs(ffb.astype(int) - ref_fb.astype(int)) <= 1).mean() > 0.99
    assert abs(int(f.radiance_rays) - ref_stats["radiance_rays"]) <= 1e-3 * ref_stats["radiance_rays"]
    print("config 1: MSE vs oracle ieee %.3e, fast %.3e" % (mse, image_mse(facc, ref_acc)))


@pytest.mark.parametrize("dl,isamp,depth", [(True, True, 8), (False, True, 4), (True, False, 16), (False, False, 28), (True, True, 16)])
def test_render_all_bsdfs(full, dl, isamp, depth):
    """Refractive + conductor + diffuse, every toggle combination, shallow and deep paths."""
    state, obj, sc = full
    p = make_params(128, 96, 8, depth, dl, isamp)
    acc, fb, stats = _gpu_render(state, p)
    ref_acc, ref_fb, ref_stats, _ = sc.render(copy_params(p), use_bvh=True)
    assert np.isfinite(acc).all()
    mse = image_mse(acc, ref_acc)
    assert mse < MSE_TOL, mse
    same = np.all(acc.view(np.uint32) == ref_acc.view(np.uint32), axis=-1).mean()
    assert same > SAME_BITS_MIN, same     # the rest differ in the last bits: ROCm vs glibc sinf/cosf/acosf
    s = stats[0]
    assert s.paths == 128 * 96 * 8
    assert abs(int(s.radiance_rays) - ref_stats["radiance_rays"]) <= 2e-3 * ref_stats["radiance_rays"]
    assert abs(int(s.shadow_rays) - ref_stats["shadow_rays"]) <= 2e-3 * max(1, ref_stats["shadow_rays"])
    if not dl:
        assert s.shadow_rays == 0
    with _math(state, "fast"):             # the default arithmetic: tolerance and counters
        facc, _, fstats = _gpu_render(state, p)
    f = fstats[0]
    assert np.isfinite(facc).all()
    if isamp:
        assert image_mse(facc, ref_acc) < MSE_TOL, image_mse(facc, ref_acc)
    else:                                  # uniform-hemisphere mode: flips counted and weighed
        rep = assert_uniform_mode_fast(facc, ref_acc, 8, dl, "all bsdfs DL %d depth %d" % (dl, depth))
        assert rep["mse"] < 2e-4, rep      # measured 3e-7 ... 5.8e-5 over four frame seeds (one flipped path in a 128 x 96 x 8 image is 5.8e-5)
    assert f.math_mode == _native.MATH_FAST and f.paths == s.paths
    assert abs(int(f.radiance_rays) - ref_stats["radiance_rays"]) <= 2e-3 * ref_stats["radiance_rays"]
    assert abs(int(f.shadow_rays) - ref_stats["shadow_rays"]) <= 2e-3 * max(1, ref_stats["shadow_rays"])
    print("DL %d IS %d depth %d: MSE vs oracle ieee %.3e (%.1f%% pixels bit-identical), fast %.3e" % (dl, isamp, depth, mse, 100 * same, image_mse(facc, ref_acc)))


def test_reference_startup_workload_config0(full):
    """The ONE workload the reference's own program defines (bench.py --config 0): cornell_box.obj, 512 x 512, 128 samples per launch,
    maxDepth 4, direct lighting off, importance sampling off (PathTracerMain.cpp:43, 58-59, 653-657) — one launch of its frame loop,
    the whole image against the oracle, the reference's summation order (one run per pixel).  IEEE level: the untrimmed MSE < 1e-6.
    Default (fast) level: this is the uniform-hemisphere mode in which single grazing paths flip (assert_uniform_mode_fast):
    measured 640 of the 262 144 pixels carry a flipped path (1.9e-5 per traced path), MSE 1.6e-5, the other pixels agree to 2.5e-17
    (profiles/r04_flip_levels.txt); the whole-image bar is 10 x that figure."""
    state, obj, sc = full
    p = make_params(512, 512, 128, 4, False, False)
    ref, _, ref_st, _ = sc.render(copy_params(p), use_bvh=True)
    acc, _, st = _gpu_render(state, p)
    s = st[0]
    assert s.math_mode == _native.MATH_IEEE and s.sample_chunks == 1 and s.paths == 512 * 512 * 128 and s.shadow_rays == 0
    mse = image_mse(acc, ref)
    same = float(np.all(acc.view(np.uint32) == ref.view(np.uint32), axis=-1).mean())
    assert mse < MSE_TOL, mse
    assert abs(int(s.radiance_rays) - ref_st["radiance_rays"]) <= 1e-3 * ref_st["radiance_rays"]
    with _math(state, "fast"):
        facc, _, fst = _gpu_render(state, p)
    f = fst[0]
    assert f.math_mode == _native.MATH_FAST and f.paths == s.paths and f.shadow_rays == 0
    rep = assert_uniform_mode_fast(facc, ref, 128, False, "config 0")
    assert rep["mse"] < 1.6e-4, rep
    assert abs(int(f.radiance_rays) - ref_st["radiance_rays"]) <= 1e-3 * ref_st["radiance_rays"]
    print("config 0 (512 x 512, 128 spp, depth 4, DL off, IS off): MSE vs oracle ieee %.3e (%.1f %% of the pixels bit-identical), fast %.3e (%d pixels carry a flipped path, the rest %.1e)"
          % (mse, 100 * same, rep["mse"], rep["n_out"], rep["mse_rest"]))


def test_progressive_accumulation(full):
    """Running mean over currentFrameIdx (pathTracerPrograms.cu:803-811), three sub-launches."""
    state, obj, sc = full
    p = make_params(96, 64, 4, 4, True, True)
    acc, fb, _ = _gpu_render(state, p, frames=3)
    ref = None
    for f in range(3):
        q = copy_params(p); q.currentFrameIdx = f
        ref, ref_fb, _, _ = sc.render(q, accumulation=ref, use_bvh=True)
    assert image_mse(acc, ref) < MSE_TOL
    assert np.all(acc.view(np.uint32) == ref.view(np.uint32), axis=-1).mean() > SAME_BITS_MIN


@pytest.mark.parametrize("chunks", [0, 1, 4])
def test_frame_batches_equal_separate_launches(full, chunks, both_modes):
    """pt_launch_frames: n sub-frames in one kernel launch leave accumulation and framebuffer bit-identical to n
    pt_launch calls — batches of 2, 3 (not a power of two: padded work items), 5 and 8, and a batch that
    continues an accumulation (first frame > 0 folds into what is already there)."""
    state, obj, sc = full
    L = _native.hip()
    assert L.pt_set_sample_chunks(state.context, chunks) == 0
    try:
        p = make_params(160, 96, 8, 5, True, True)
        want_acc, want_fb, st1 = _gpu_render(state, p, frames=8)
        rays1 = sum(int(s.radiance_rays + s.shadow_rays) for s in st1)
        for fuse in (2, 3, 5, 8):
            acc, fb, st = _gpu_render(state, p, frames=8, fuse=fuse)
            assert np.array_equal(acc.view(np.uint32), want_acc.view(np.uint32)), "accumulation differs at fuse=%d" % fuse
            assert np.array_equal(fb, want_fb), fuse
            assert sum(int(s.radiance_rays + s.shadow_rays) for s in st) == rays1
            assert all(int(s.pixels) == 160 * 96 for s in st)
        # continuing: frames 0..2 separately, then 3..7 as one batch on top of the same buffer
        state.params.width, state.params.height = p.width, p.height
        out = pt.OutputBuffer(pt.OutputBufferType.DEVICE, p.width, p.height, state)
        for f in range(3):
            state.params.currentFrameIdx = f
            pt.LaunchCurrentFrame(out, state)
        state.params.currentFrameIdx = 3
        pt.LaunchCurrentFrame(out, state, 5)
        assert np.array_equal(pt.readAccumulation(state).view(np.uint32), want_acc.view(np.uint32))
        assert np.array_equal(out.getHostPointer(), want_fb)
        out.free()
        # the running mean itself still matches the oracle
        ref = None
        for f in range(8):
            q = copy_params(p); q.currentFrameIdx = f
            ref, _, _, _ = sc.render(q, accumulation=ref, use_bvh=True)
        assert image_mse(want_acc, ref) < MSE_TOL
        # a batch whose frame sums exceed the scratch limit runs as several kernel launches: same bits, summed counters
        assert L.pt_set_scratch_limit(state.context, 1 << 20) == 0            # 1 MiB = 4 sub-frames of 160 x 96 float4
        try:
            acc, fb, st = _gpu_render(state, p, frames=8, fuse=8)
            assert np.array_equal(acc.view(np.uint32), want_acc.view(np.uint32)) and np.array_equal(fb, want_fb)
            assert len(st) == 1 and int(st[0].paths) == 160 * 96 * 8 * 8 and int(st[0].radiance_rays + st[0].shadow_rays) == rays1
        finally:
            assert L.pt_set_scratch_limit(state.context, 1 << 30) == 0
        assert L.pt_set_scratch_limit(state.context, 1000) != 0
        assert L.pt_launch_frames(state.context, C.byref(state.params), 0) != 0 and L.pt_launch_frames(state.context, C.byref(state.params), 65) != 0
    finally:
        assert L.pt_set_sample_chunks(state.context, 1) == 0


@pytest.mark.parametrize("mode", ["ieee", "fast"])
def test_every_kernel_variant_gives_the_same_bits(full, mode):
    """The scheduler variants (segment-synchronous, persistent traversal at several thresholds, fp32 /
    quantised / LDS-staged nodes, the workgroup wavefront kernel) only change the interleaving between lanes: identical
    images — within each math mode (every product variant exists in both)."""
    state, obj, _ = full
    L = _native.hip()
    p = make_params(160, 96, 8, 8, True, True)
    ref = None
    tried = 0
    try:
        pt.setMathMode(state, mode)
        for v in range(64):
            name = L.pt_variant_name(v)
            if name is None:
                break
            if name.startswith(b"DIAG") or name.startswith(b"LIGHTS") or (name.startswith(b"TRIG") and mode == "ieee") or L.pt_set_tuning(state.context, 0, v) != 0:
                continue
            acc, fb, st = _gpu_render(state, p)
            assert st[0].math_mode == (_native.MATH_FAST if mode == "fast" else _native.MATH_IEEE) and st[0].variant == v
            tried += 1
            if ref is None:
                ref = (acc, fb, st[0].radiance_rays, st[0].shadow_rays)
            else:
                assert np.array_equal(acc.view(np.uint32), ref[0].view(np.uint32)), "variant %d differs" % v
                assert np.array_equal(fb, ref[1])
                assert (st[0].radiance_rays, st[0].shadow_rays) == (ref[2], ref[3])
    finally:
        assert L.pt_set_tuning(state.context, 0, _DEFAULT_VARIANT) == 0
        pt.setMathMode(state, "ieee")
    assert tried >= 4


def test_queue_order_and_pixel_classes_change_no_bit(diffuse, both_modes):
    """How the work queue is dealt over its eight shards (contiguous bands, rows or tiles round robin, or one queue) and whether pixels are classified beforehand (pixels whose rays cannot reach the scene box are settled when their
    grant is decoded; pixels whose rays all reach it skip the cull test) are scheduling matters: the accumulation, the
    framebuffer and the ray / path counters are the same bit for bit.  A 16:9 frame, so that both pixel classes and a band of
    unclassified pixels around the box's silhouette exist; sample runs on, two frames in one launch."""
    state, obj, _ = diffuse
    L = _native.hip()
    p = make_params(640, 360, 16, 6, True, True)
    ref = None
    try:
        assert L.pt_set_sample_chunks(state.context, 4) == 0
        for classes, order in ((0, 0), (1, 0), (0, 1), (1, 1), (1, 2), (1, 3)):
            assert L.pt_debug_pixel_classes(state.context, classes) == 0 and L.pt_debug_queue_order(state.context, order) == 0
            acc, fb, st = _gpu_render(state, p, frames=2, fuse=2)
            cnt = (int(st[0].radiance_rays), int(st[0].shadow_rays), int(st[0].paths), int(st[0].pixels))
            if ref is None:
                ref = (acc, fb, cnt, int(st[0].culled_rays))
                assert cnt[2] == 640 * 360 * 16 * 2 and cnt[3] == 640 * 360
                assert 0.40 < ref[3] / cnt[2] < 0.50          # the box fills the middle of the wide frame
            else:
                assert np.array_equal(acc.view(np.uint32), ref[0].view(np.uint32)), (classes, order)
                assert np.array_equal(fb, ref[1]) and cnt == ref[2], (classes, order)
                if classes:        # whole pixels settled without a cull test: never fewer than the per-ray test finds
                    assert int(st[0].culled_rays) >= ref[3] - 16 * 2 * 4 * 360
        assert L.pt_debug_queue_order(state.context, 4) != 0
        # how many items a wave takes per queue atomic (capped for small launches so that every wave fetches >= 16 grants,
        # profiles/r04_wave_timeline_c0.txt; ACGPT_GRANT overrides the rule): one group, the rule's choice, far more than a wave's share
        L.pt_debug_pixel_classes(state.context, 1); L.pt_debug_queue_order(state.context, 1)
        for grant in ("4", "64", "256", "1024"):           # 256 = 64 groups of 4 runs, the most a grant can hold; 1024 is refused (the rule stays)
            os.environ["ACGPT_GRANT"] = grant
            acc, fb, st = _gpu_render(state, p, frames=2, fuse=2)
            cnt = (int(st[0].radiance_rays), int(st[0].shadow_rays), int(st[0].paths), int(st[0].pixels))
            assert np.array_equal(acc.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(fb, ref[1]) and cnt == ref[2], grant
    finally:
        os.environ.pop("ACGPT_GRANT", None)
        L.pt_debug_pixel_classes(state.context, 1); L.pt_debug_queue_order(state.context, 1)
        L.pt_set_sample_chunks(state.context, 1)


def test_math_modes(full, diffuse):
    """pt_set_math_mode: "fast" (the library's default) is the arithmetic the reference's own build uses — nvcc --use_fast_math,
    /root/reference/CMakeLists.txt:267: approximate reciprocal, square root, sine and cosine in the shading code; "ieee" is the
    level the oracle is written at.  Same paths (counted), the same image within the parity tolerance against the oracle and
    against each other, other low bits; traversal untouched: a directly seen image (depth 0: the eye's rays, the hit, the
    emitter) is the same bit for bit but for the direction's own rounding.  The TRIG variant (hardware sin / cos in the cosine
    sampler only) sits between the two."""
    L = _native.hip()
    for name, (state, obj, sc) in (("glass + metal", full), ("diffuse", diffuse)):
        p = make_params(160, 96, 32, 8, True, True)
        base, _, st0 = _gpu_render(state, p)
        with _math(state, "fast"):
            fast, _, st1 = _gpu_render(state, p)
            fast2, _, _ = _gpu_render(state, p)
        ref, _, ref_stats, _ = sc.render(copy_params(p), use_bvh=True)
        assert st0[0].math_mode == _native.MATH_IEEE and st1[0].math_mode == _native.MATH_FAST and st0[0].variant == st1[0].variant
        assert st1[0].paths == st0[0].paths == 160 * 96 * 32
        for st in (st0[0], st1[0]):
            assert abs(int(st.radiance_rays) - ref_stats["radiance_rays"]) <= 2e-3 * ref_stats["radiance_rays"]
            assert abs(int(st.shadow_rays) - ref_stats["shadow_rays"]) <= 2e-3 * ref_stats["shadow_rays"]
        m_fast, m_base, m_between = image_mse(fast, ref), image_mse(base, ref), image_mse(fast, base)
        print("%s: MSE vs oracle: ieee %.3e, fast %.3e; fast vs ieee %.3e" % (name, m_base, m_fast, m_between))
        assert m_fast < MSE_TOL and m_base < MSE_TOL and m_between < MSE_TOL
        assert not np.array_equal(fast.view(np.uint32), base.view(np.uint32))
        assert np.array_equal(fast.view(np.uint32), fast2.view(np.uint32))          # deterministic in itself
        # the means agree far below the Monte-Carlo noise of either: no bias from the approximate instructions
        assert abs(float(fast[..., :3].mean()) - float(base[..., :3].mean())) < 1e-3 * float(base[..., :3].mean())
    state = full[0]
    assert L.pt_set_math_mode(state.context, 2) != 0 and b"PT_MATH" in L.pt_last_error(state.context)
    v = [i for i in range(64) if (L.pt_variant_name(i) or b"").startswith(b"TRIG")]
    assert len(v) == 1
    p = make_params(160, 96, 32, 8, True, True)
    base, _, st0 = _gpu_render(state, p)
    try:
        assert L.pt_set_tuning(state.context, 0, v[0]) == 0
        trig, _, st1 = _gpu_render(state, p)
    finally:
        assert L.pt_set_tuning(state.context, 0, _DEFAULT_VARIANT) == 0
    ref, _, _, _ = full[2].render(copy_params(p), use_bvh=True)
    assert st1[0].paths == st0[0].paths
    assert image_mse(trig, ref) < MSE_TOL and image_mse(trig, base) < MSE_TOL
    assert not np.array_equal(trig.view(np.uint32), base.view(np.uint32))


def test_fast_math_flips_fall_with_the_sample_count(diffuse):
    """Uniform-hemisphere mode in the default math mode: the pixels that differ from the oracle are single paths that took the
    other branch at a self-intersecting grazing bounce — unbiased coin flips, so their weight in the image
    falls with the sample count: the MSE against the oracle at 256 spp is a fraction of that at 16 spp, the image means agree,
    and with the few pixels that carry a flip set aside the images agree to rounding at either count."""
    state, obj, sc = diffuse
    res = {}
    for spp in (16, 256):
        p = make_params(64, 64, spp, 4, False, False)
        ref, _, ref_st, _ = sc.render(copy_params(p), use_bvh=True)
        with _math(state, "fast"):
            acc, _, st = _gpu_render(state, p)
        assert st[0].paths == 64 * 64 * spp and abs(int(st[0].radiance_rays) - ref_st["radiance_rays"]) <= 1e-3 * ref_st["radiance_rays"]
        rep = assert_uniform_mode_fast(acc, ref, spp, False, "64 x 64, %d spp" % spp)
        res[spp] = (rep["mse"], rep["mse_rest"], rep["mean_a"], rep["mean_b"], rep["n_out"])
        print("uniform mode, fast math, %3d spp: MSE vs oracle %.3e (%.3e over the pixels without a flip); means %.5f / %.5f; %d pixels carry a flipped path" % ((spp,) + res[spp]))
    for spp, (mse, rest, m_gpu, m_ref, n_out) in res.items():
        assert abs(m_gpu - m_ref) <= 1e-2 * m_ref, (spp, m_gpu, m_ref)      # one flipped path in 4096 pixels x 16 spp moves the mean by 3e-3 (measured up to 6.6e-3)
    # a flip weighs 1 / spp: measured 0 ... 1.8e-4 at 16 spp (0 ... 2 flips), 5e-6 ... 1.1e-5 at 256 spp (14 ... 23 flips of 1/16 the weight squared)
    assert res[16][0] < 5e-4 and res[256][0] < 4e-5, res


@pytest.mark.parametrize("chunks", [2, 8, 32, 0])
def test_sample_chunks(full, chunks):
    """pt_set_sample_chunks: the same samples summed as consecutive runs.  Against the oracle with the
    same association the usual bit-level agreement holds; against the reference's order (chunks = 1)
    only the last bits of the sums move.  0 = automatic."""
    state, obj, sc = full
    L = _native.hip()
    p = make_params(128, 80, 32, 6, True, True)
    base, _, base_st = _gpu_render(state, p)                   # chunks = 1
    try:
        assert L.pt_set_sample_chunks(state.context, chunks) == 0
        acc, fb, st = _gpu_render(state, p)
        used = st[0].sample_chunks
        assert used == (chunks if chunks else 8)      # automatic: 32 runs wanted for a small image, capped so that a run keeps >= 4 of the 32 samples
        assert (st[0].radiance_rays, st[0].shadow_rays, st[0].paths, st[0].pixels) == \
               (base_st[0].radiance_rays, base_st[0].shadow_rays, base_st[0].paths, base_st[0].pixels), "same paths, same rays"
        ref, ref_fb, _, _ = sc.render(copy_params(p), use_bvh=True, chunks=used)
        assert image_mse(acc, ref) < MSE_TOL
        assert np.all(acc.view(np.uint32) == ref.view(np.uint32), axis=-1).mean() > SAME_BITS_MIN
        rel = np.abs(acc[..., :3] - base[..., :3]) / np.maximum(np.abs(base[..., :3]), 1e-3)
        assert rel.max() < 1e-4, "re-association moves sums by ulps only"
        assert image_mse(acc, base) < 1e-10
        # 3 samples cannot be cut into 2 runs
        q = copy_params(state.params); q.samplesPerPixel = 3
        if chunks > 1:
            assert L.pt_launch(state.context, C.byref(q)) != 0
    finally:
        assert L.pt_set_sample_chunks(state.context, 1) == 0


def test_deterministic_and_zero_copy(full, both_modes):
    """Same inputs -> same bits, and the ZERO_COPY framebuffer mode sees the same pixels."""
    state, obj, _ = full
    p = make_params(80, 60, 4, 5, True, True)
    a1, f1, _ = _gpu_render(state, p)
    zc = pt.OutputBuffer(pt.OutputBufferType.ZERO_COPY, 80, 60, state)
    a2, f2, _ = _gpu_render(state, p, out_buffer=zc)
    assert np.array_equal(a1.view(np.uint32), a2.view(np.uint32))
    assert np.array_equal(f1, f2)


def test_tile_partition_is_exact(full, oracle, both_modes):
    """Two ranks' pixel sets (sutil/WorkDistribution.h:60-81) are disjoint, cover the image, and their
    sum over a zero-initialised buffer equals the single-GPU launch bit for bit."""
    state, obj, _ = full
    L = _native.hip()
    p = make_params(100, 52, 4, 4, True, True)     # neither a multiple of the 16x4 strip nor of 8
    assert L.pt_set_sample_chunks(state.context, 4) == 0    # the multi-GPU setting: runs of samples per lane
    whole, _, _ = _gpu_render(state, p)
    total = np.zeros_like(whole)
    cover = np.zeros(whole.shape[:2], np.int32)
    try:
        for world in (2,):
            for rank in range(world):
                assert L.pt_set_partition(state.context, rank, world) == 0
                state.refreshAccumulationBuffer = True
                pt.updateState(None, state)
                L.pt_device_memset(state.context, state.params.accumulationBuffer, 0, 100 * 52 * 16)
                ob = pt.OutputBuffer(pt.OutputBufferType.DEVICE, 100, 52, state)
                state.params.currentFrameIdx = 0
                pt.LaunchCurrentFrame(ob, state)
                part = pt.readAccumulation(state)
                ob.free()
                mine = part[..., 3] == 1.0
                cover += mine
                total += part
                expect = np.zeros_like(mine)
                for si in range(oracle.num_samples(world, 100, 52)):
                    x, y = oracle.sample_pixel(world, 100, rank, si)
                    if x < 100 and y < 52:
                        expect[y, x] = True
                assert np.array_equal(mine, expect)
                assert pt.getStats(state).pixels == int(expect.sum())
    finally:
        L.pt_set_partition(state.context, 0, 1)
        L.pt_set_sample_chunks(state.context, 1)
    assert np.all(cover == 1)
    assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))


def test_edge_cases(gpu_state_factory, oracle, tmp_path):
    L = _native.hip()
    # a single triangle, and an OBJ with no faces at all
    one = tmp_path / "one.obj"
    one.write_text("mtllib one.mtl\nv 100 100 300\nv 450 100 300\nv 278 450 300\nusemtl m\nf 1 2 3\n")
    (tmp_path / "one.mtl").write_text("newmtl m\nKd 0.5 0.6 0.7\nKe 1 2 3\n")
    state, obj = gpu_state_factory(str(one), width=32, height=32, max_depth=2, spp=2)
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    p = make_params(32, 32, 2, 2, True, True)
    acc, fb, _ = _gpu_render(state, p)
    ref, _, _, _ = sc.render(copy_params(p), use_bvh=False)
    assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    assert acc[..., :3].max() > 1.0          # Ke + Ke*Kd seen directly
    with _math(state, "fast"):               # the default arithmetic: the emitter is the only surface, every value is Ke + Ke * Kd or 0
        facc, _, fst = _gpu_render(state, p)
    assert image_mse(facc, ref) < MSE_TOL and np.array_equal(facc[..., :3] > 0, ref[..., :3] > 0) and fst[0].math_mode == _native.MATH_FAST
    # empty scene: every ray misses, image is black with alpha 1 — in both math modes
    v = np.zeros(4, np.float32)
    assert L.pt_set_scene(state.context, v.ctypes.data, 1, None, 0, None, None, 0) == 0
    state.params.handle = L.pt_scene_handle(state.context)
    for mode in ("ieee", "fast"):
        with _math(state, mode):
            acc, fb, st = _gpu_render(state, p)
        assert np.all(acc[..., :3] == 0.0) and np.all(acc[..., 3] == 1.0) and np.all(fb[..., :3] == 0) and np.all(fb[..., 3] == 255)
        assert st[0].radiance_rays == 32 * 32 * 2
    # argument checking (PathTracerMain.cpp:42, 122-128; pathTracerPrograms.cu:727)
    for field, bad in (("maxDepth", 0), ("maxDepth", 29), ("samplesPerPixel", 0), ("width", 0), ("height", 70000)):
        q = copy_params(state.params)
        setattr(q, field, bad)
        assert L.pt_launch(state.context, C.byref(q)) != 0
        assert L.pt_last_error(state.context)
    # a face whose material id is tinyobj's -1 is rejected at scene upload
    vv = np.array([0, 0, 0, 1, 1, 0, 0, 1, 0, 1, 0, 1], np.float32)
    ii = np.array([0, 1, 2], np.uint32); mm = np.array([0xFFFFFFFF], np.uint32)
    mats = (pt.Material * 1)()
    assert L.pt_set_scene(state.context, vv.ctypes.data, 3, ii.ctypes.data, 1, mm.ctypes.data, C.addressof(mats), 1) != 0
    assert b"material index" in L.pt_last_error(state.context)


def test_windowed_stack_on_a_deep_tree(gpu_state_factory, oracle, tmp_path):
    """The large-scene kernel keeps a sliding window of 16 stack entries per lane in LDS and moves deeper ones to global memory
    four at a time.  2^18 small triangles strung along the camera's axis, a little off it: a ray down the axis enters the boxes
    of both children at every level of an 18-level tree, so its stack of pending far children outgrows the window and shrinks
    again.  The windowed kernel must move entries (counted), and agree bit for bit with the kernels that keep the whole stack
    in LDS (fp32 nodes, four waves) and with the segment-synchronous one; and match the oracle."""
    n = 1 << 18
    k = np.arange(n, dtype=np.float64)
    ang = k * 2.399963                               # golden-angle spiral around the axis
    rad = 0.6 + 1.4 * ((k * 0.6180339887) % 1.0)
    cx, cy, cz = 278.0 + rad * np.cos(ang), 273.0 + rad * np.sin(ang), 20.0 + 500.0 * k / n
    v = np.empty((n, 3, 3))
    v[:, 0] = np.stack([cx, cy, cz], 1)
    v[:, 1] = np.stack([cx + 0.35, cy + 0.05, cz], 1)
    v[:, 2] = np.stack([cx + 0.05, cy + 0.35, cz], 1)
    with open(tmp_path / "line.obj", "w") as fh:
        fh.write("mtllib line.mtl\nusemtl white\n")
        np.savetxt(fh, v.reshape(-1, 3), fmt="v %.4f %.4f %.4f")
        idx = np.arange(1, 3 * n + 1).reshape(n, 3)
        np.savetxt(fh, idx, fmt="f %d %d %d")
        fh.write("usemtl light\nv 213 548 227\nv 343 548 227\nv 343 548 332\nv 213 548 332\nf -4 -3 -2 -1\n")
    (tmp_path / "line.mtl").write_text("newmtl white\nKd 0.7 0.7 0.7\nnewmtl light\nKd 0.8 0.8 0.8\nKe 10 10 10\n")
    L = _native.hip()
    state, obj = gpu_state_factory(str(tmp_path / "line.obj"), width=64, height=64)
    info = pt.getBvhInfo(state)
    assert info.n_tris == n + 2 and info.stack_entries > 18, info.stack_entries
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    p = make_params(96, 96, 4, 4, True, True)
    # a narrow view straight down the string: the image is 4 units wide where the string begins and 9 where it ends
    p.cameraEye = _native.Float3(278.0, 273.0, -400.0)
    p.cameraU, p.cameraV, p.cameraW = _native.Float3(-2.0, 0.0, 0.0), _native.Float3(0.0, 2.0, 0.0), _native.Float3(0.0, 0.0, 400.0)
    ref, _, ref_st, _ = sc.render(copy_params(p), use_bvh=True)
    for mode in ("ieee", "fast"):             # the oracle's level, and the library's default arithmetic
        imgs, moves = {}, 0
        try:
            pt.setMathMode(state, mode)
            for v_ in (9, 3, 0):              # windowed stack / fp32 nodes, whole stack in LDS / segment-synchronous
                assert L.pt_set_tuning(state.context, 0, v_) == 0, L.pt_last_error(state.context)
                acc, fb, st = _gpu_render(state, p)
                assert int(st[0].variant) == v_
                imgs[v_] = (acc, (int(st[0].radiance_rays), int(st[0].shadow_rays)))
                if v_ == 9:
                    d = (C.c_uint64 * 1)()
                    assert L.pt_debug_window_moves(state.context, d) == 0
                    moves = int(d[0])
        finally:
            assert L.pt_set_tuning(state.context, 0, _DEFAULT_VARIANT) == 0
            pt.setMathMode(state, "ieee")
        print("windowed stack (%s): %d wave-level moves between the LDS window and global memory; stack %d entries" % (mode, moves, info.stack_entries))
        assert moves > 0, "no ray outgrew the window: the test does not exercise what it is for"
        for v_, (acc, cnt) in imgs.items():
            assert np.array_equal(acc.view(np.uint32), imgs[0][0].view(np.uint32)), "variant %d differs from the segment-synchronous kernel" % v_
            assert cnt == imgs[0][1], v_
        assert image_mse(imgs[9][0], ref) < MSE_TOL
        assert abs(imgs[9][1][0] - ref_st["radiance_rays"]) <= 2e-3 * ref_st["radiance_rays"]
    sc.close()


def test_large_scene_properties(gpu_state_factory, oracle, tmp_path):
    """BASELINE config-5 size (1.3 M triangles, generated from a fixed seed): brute force is out of
    reach, so parity is checked (a) against brute force on a small ray sample, (b) against the
    oracle's own BVH on a large one, (c) through size-independent properties: any-hit == (closest
    hit exists) on the same interval, determinism across both hierarchy builders."""
    import sys
    sys.path.insert(0, pt.SCENES)
    import make_scenes
    path = str(tmp_path / "stress.obj")
    make_scenes.stress_scene(path)
    state, obj = gpu_state_factory(path, width=64, height=64)
    L = _native.hip()
    info = pt.getBvhInfo(state)
    assert info.n_tris == 64 * 20480 + 12 and info.n_nodes == info.n_tris - 1
    assert info.max_depth < info.stack_entries <= 128
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    rays = random_rays(120000, 41, lo=(20, 20, 20), hi=(530, 530, 540))
    n = rays.shape[0]
    t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32); hit = np.zeros(n, np.uint8)
    assert L.pt_trace_closest(state.context, rays.ctypes.data, n, t.ctypes.data, prim.ctypes.data) == 0
    assert L.pt_trace_any(state.context, rays.ctypes.data, n, hit.ctypes.data) == 0
    t_ref, p_ref = sc.trace_closest(rays, use_bvh=True)
    assert np.array_equal(prim, p_ref) and np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    assert np.array_equal(hit != 0, prim != 0xFFFFFFFF)                       # (c) any-hit <=> a closest hit exists
    tb, pb = sc.trace_closest(rays[:150], use_bvh=False)                       # (a) brute force sample
    assert np.array_equal(prim[:150], pb) and np.array_equal(t[:150].view(np.uint32), tb.view(np.uint32))
    # the node array and box test the render kernel itself walks on this scene — fp16 {centre, half extent}, a scale per axis, the reinserted
    # tree in depth-first order — through the ray-stream kernel (stream format 4): the same hits
    ts = np.zeros(n, np.float32); prims = np.zeros(n, np.uint32); ms = C.c_float()
    assert L.pt_bench_traversal(state.context, rays.ctypes.data, n, 1, 4, ts.ctypes.data, prims.ctypes.data, C.byref(ms), None) == 0
    assert np.array_equal(prims, p_ref) and np.array_equal(ts.view(np.uint32), t_ref.view(np.uint32))
    # the other builder, same answers
    assert L.pt_set_build_mode(state.context, 0) == 0
    pt.buildTheAccelarationStructure(state, obj)
    t2 = np.zeros(n, np.float32); prim2 = np.zeros(n, np.uint32)
    assert L.pt_trace_closest(state.context, rays.ctypes.data, n, t2.ctypes.data, prim2.ctypes.data) == 0
    assert np.array_equal(prim2, prim) and np.array_equal(t2.view(np.uint32), t.view(np.uint32))
    assert L.pt_set_build_mode(state.context, 1) == 0
    pt.buildTheAccelarationStructure(state, obj)
    # a small render against the oracle
    p = make_params(96, 54, 2, 4, True, True)
    acc, _, st = _gpu_render(state, p)
    ref, _, ref_st, _ = sc.render(copy_params(p), use_bvh=True)
    assert image_mse(acc, ref) < MSE_TOL
    assert abs(int(st[0].radiance_rays) - ref_st["radiance_rays"]) <= 5e-3 * ref_st["radiance_rays"]
    # BASELINE config 5's own geometry: 1920x1080, 2 x 128 spp, maxDepth 8, IS + DL, automatic sample runs, one batch of
    # two steps (what bench.py --config 5 times), against the oracle on two windows
    import oracle_lib
    assert L.pt_set_sample_chunks(state.context, 0) == 0
    p = make_params(1920, 1080, 128, 8, True, True)
    acc, _, st = _gpu_render(state, p, frames=2, fuse=2)
    assert st[0].paths == 1920 * 1080 * 128 * 2
    with _math(state, "fast"):             # what bench.py --config 5 times by default
        facc, _, fst = _gpu_render(state, p, frames=2, fuse=2)
    assert fst[0].paths == st[0].paths and fst[0].math_mode == _native.MATH_FAST and fst[0].variant == st[0].variant
    for name, win in (("centre", (944, 500, 32, 32)), ("upper right", (1200, 800, 32, 32))):
        r = None
        for f in range(2):
            q = copy_params(p); q.currentFrameIdx = f
            r, _, _ = oracle_lib.render_window(sc, q, win, accumulation=r, chunks=int(st[0].sample_chunks))
        x0, y0, ww, wh = win
        a, rr = acc[y0:y0 + wh, x0:x0 + ww], r[y0:y0 + wh, x0:x0 + ww]
        fa = facc[y0:y0 + wh, x0:x0 + ww]
        print("config 5 / %s: MSE %.3e (fast math %.3e), mean %.4f" % (name, image_mse(a, rr), image_mse(fa, rr), float(rr[..., :3].mean())))
        assert image_mse(a, rr) < MSE_TOL and image_mse(fa, rr) < MSE_TOL and rr[..., :3].mean() > 1e-3
    assert L.pt_set_sample_chunks(state.context, 1) == 0


def test_node_format_follows_the_geometry(gpu_state_factory, oracle, tmp_path):
    """pt_set_scene's choice between fp16 and fp32 nodes: spheres of ordinary size keep the fp16 planes' inflation of a box small
    and get the fp16 kernel; the same spheres shrunk to a fiftieth are finer than the planes (mean box inflation > 3: whole
    subtrees collapse onto the same fp16 planes) and get the fp32 kernel.  Both images agree with the oracle either way."""
    import sys
    sys.path.insert(0, pt.SCENES)
    import make_scenes
    L = _native.hip()
    for scale, want in ((1.0, "fp16"), (0.02, "fp32")):
        path = str(tmp_path / ("spheres_%g.obj" % scale))
        make_scenes.stress_scene(path, n_spheres=8, subdiv=4, radius_scale=scale)
        state, obj = gpu_state_factory(path, sample_chunks=1, width=64, height=48)
        info = pt.getBvhInfo(state)
        assert info.n_tris == 8 * 5120 + 12
        assert 1.0 <= info.half_area_ratio < 1.5
        assert (info.half_box_inflation <= 3.0) == (want == "fp16"), (scale, info.half_box_inflation)
        p = make_params(64, 48, 8, 5, True, True)
        acc, fb, st = _gpu_render(state, p)
        name = L.pt_variant_name(int(st[0].variant)).decode()
        print("radius x %g: area ratio %.4f, mean box inflation %.3f -> %s" % (scale, info.half_area_ratio, info.half_box_inflation, name))
        assert want in name, name
        sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
        ref, _, _, _ = sc.render(copy_params(p), use_bvh=True)
        assert image_mse(acc, ref) < MSE_TOL
        sc.close()


def test_headless_app_matches_the_python_path(full, tmp_path):
    """acgpt_main (the C++ mirror of PathTracerMain.cpp) and the Python mirror drive the same library: the
    same frames, the same bytes — including the key replay (toggle importance sampling, reset)."""
    import os
    import subprocess
    from PIL import Image
    exe = os.path.join(os.path.dirname(_native.hip_library_path()), "acgpt_main")
    if not os.path.exists(exe):
        from acgpathtracing_amd import _build
        _build.build_main()
    out = str(tmp_path / "app.ppm")
    r = subprocess.run([exe, "--obj", SCENE_FULL, "--width", "96", "--height", "64", "--spp-per-launch", "8", "--frames", "3",
                        "--max-depth", "5", "--direct-lighting", "--keys", "1", "--out", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Using Importance Sampling: yes" in r.stdout and "Total Samples 16" in r.stdout   # reset after the toggle: 2 frames x 8
    got = np.asarray(Image.open(out).convert("RGB"))
    # the same session through the Python mirror: frame 0 without IS is discarded by the reset
    state, obj, _ = full
    L = _native.hip()
    assert L.pt_set_sample_chunks(state.context, 0) == 0          # the app uses the library defaults
    try:
        p = make_params(96, 64, 8, 5, True, True)
        with _math(state, "fast"):
            acc, fb, _ = _gpu_render(state, p, frames=2)
        acc_i, fb_i, _ = _gpu_render(state, p, frames=2)
    finally:
        assert L.pt_set_sample_chunks(state.context, 1) == 0
    assert np.array_equal(got, fb[::-1, :, :3])
    # --math ieee: the oracle's arithmetic level, the mode the fixtures' contexts run in
    out_i = str(tmp_path / "app_ieee.ppm")
    r = subprocess.run([exe, "--obj", SCENE_FULL, "--width", "96", "--height", "64", "--spp-per-launch", "8", "--frames", "3", "--math", "ieee",
                        "--max-depth", "5", "--direct-lighting", "--keys", "1", "--out", out_i], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(np.asarray(Image.open(out_i).convert("RGB")), fb_i[::-1, :, :3])
    # --fuse-frames: 5 frames as batches of 4 + 1 (with a dump after frame 2 that splits the first batch): same bytes
    # as one launch per frame
    outs = {}
    for fuse in (1, 4):
        o = str(tmp_path / ("fuse%d.ppm" % fuse))
        r = subprocess.run([exe, "--obj", SCENE_FULL, "--width", "96", "--height", "64", "--spp-per-launch", "8", "--frames", "5", "--max-depth", "5",
                            "--direct-lighting", "--importance-sampling", "--fuse-frames", str(fuse), "--dump-every", "2", "--out", o],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert "Total Samples 40" in r.stdout
        outs[fuse] = (open(o, "rb").read(), open(o + ".2.ppm", "rb").read(), open(o + ".4.ppm", "rb").read())
    assert outs[1] == outs[4]


# Windows (x0, y0, w, h) of the 1920x1080 frame from the preset camera (image row 0 = bottom, U points to -x): the box
# front spans columns 431..1489, the ceiling light columns 861..1059 x rows 923..958, the glass sphere 605..840 x 84..320,
# the metal mesh 1024..1207 x 373..560 (projected from the OBJ).
HEADLINE_WINDOWS = {
    "back wall": (900, 500, 48, 48),
    "glass sphere": (700, 180, 48, 48),
    "metal mesh": (1090, 440, 48, 48),
    "light edge": (1036, 915, 48, 48),
    "box edge (aspect)": (415, 500, 32, 32),
    "outside the box": (100, 500, 32, 32),
}


# Fraction of a window's pixels whose fp32 accumulation must equal the oracle's bit for bit at 1024 spp (measured 28-82 %: a
# pixel's 1024-sample sum differs as soon as ONE of its ~10^4 sampled directions rounds differently in ROCm's and glibc's
# sinf / cosf / acosf).  A regression that halved the measured fraction would be a real change of the arithmetic.
HEADLINE_SAME_BITS_MIN = 0.20


def _headline_check(gpu_state_factory, oracle, scene, depth, frames, windows, label):
    """Full 1920x1080 render (16:9 camera: U scales with the aspect, sutil/Camera.cpp:34-45), 128 spp per step, the
    library's automatic sample runs and frame batches of 8 — what bench.py times — against the oracle on pixel windows."""
    import oracle_lib
    state, obj = gpu_state_factory(scene, sample_chunks=0, width=64, height=64)
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    W, H, S = 1920, 1080, 128
    p = make_params(W, H, S, depth, True, True)
    acc, fb, st = _gpu_render(state, p, frames=frames, fuse=8)
    chunks = int(st[0].sample_chunks)
    assert sum(int(t.paths) for t in st) == W * H * S * frames
    with _math(state, "fast"):             # what bench.py times by default: the same frame in the library's default arithmetic
        facc, _, fst = _gpu_render(state, p, frames=frames, fuse=8)
    assert sum(int(t.paths) for t in fst) == W * H * S * frames and int(fst[0].sample_chunks) == chunks and fst[0].math_mode == _native.MATH_FAST
    for name in ("radiance_rays", "shadow_rays"):
        n_i, n_f = sum(int(getattr(t, name)) for t in st), sum(int(getattr(t, name)) for t in fst)
        assert abs(n_i - n_f) <= 1e-4 * n_i, (name, n_i, n_f)
    c_i, c_f = sum(int(t.culled_rays) for t in st), sum(int(t.culled_rays) for t in fst)
    assert abs(c_i - c_f) <= 1e-5 * max(1, c_i), (c_i, c_f)     # a camera ray within rounding of the box's silhouette may fall either way
    assert np.isfinite(facc).all() and np.all(facc[..., 3] == 1.0)
    # the two math modes over the WHOLE 1920x1080 frame (the oracle follows on windows only): the same image
    whole = image_mse(facc, acc)
    m_i, m_f = float(acc[..., :3].mean()), float(facc[..., :3].mean())
    print("%s, whole frame: fast vs ieee MSE %.3e, means %.6f / %.6f, %.1f %% of the pixels bit-identical" %
          (label, whole, m_i, m_f, 100 * float(np.all(acc.view(np.uint32) == facc.view(np.uint32), axis=-1).mean())))
    assert whole < 1e-7 and abs(m_i - m_f) <= 1e-4 * m_i, (whole, m_i, m_f)
    worst = 0.0
    for name, (x0, y0, ww, wh) in windows.items():
        ref = None
        for f in range(frames):
            q = copy_params(p); q.currentFrameIdx = f
            ref, _, _ = oracle_lib.render_window(sc, q, (x0, y0, ww, wh), accumulation=ref, chunks=chunks)
        a, r = acc[y0:y0 + wh, x0:x0 + ww], ref[y0:y0 + wh, x0:x0 + ww]
        mse = image_mse(a, r)
        same = float(np.all(a.view(np.uint32) == r.view(np.uint32), axis=-1).mean())
        rel = float((np.abs(a[..., :3] - r[..., :3]) / np.maximum(np.abs(r[..., :3]), 1e-2)).max())
        print("%s / %s: MSE %.3e, max rel diff %.3e, %.1f%% pixels bit-identical, mean %.4f" % (label, name, mse, rel, 100 * same, float(r[..., :3].mean())))
        assert np.all(a[..., 3] == 1.0)
        assert mse < MSE_TOL, (name, mse)
        assert same >= HEADLINE_SAME_BITS_MIN, (name, same)
        fa = facc[y0:y0 + wh, x0:x0 + ww]
        fmse = image_mse(fa, r)
        frel = float((np.abs(fa[..., :3] - r[..., :3]) / np.maximum(np.abs(r[..., :3]), 1e-2)).max())
        print("%s / %s, fast math: MSE %.3e, max rel diff %.3e" % (label, name, fmse, frel))
        assert fmse < MSE_TOL, (name, fmse)
        if name == "outside the box":
            assert np.all(fa[..., :3] == 0.0)
        worst = max(worst, mse, fmse)
        if name == "outside the box":
            assert np.all(a[..., :3] == 0.0) and np.all(r[..., :3] == 0.0)
        else:
            assert r[..., :3].mean() > 1e-3, "window %s is empty: the camera mapping moved" % name
        if name == "box edge (aspect)":       # the box front begins at column 431: left part black, right part lit, on both sides
            assert np.all(r[:, :8, :3] == 0.0) and np.all(a[:, :8, :3] == 0.0) and a[:, 24:, :3].mean() > 0.02
    sc.close()
    return worst


def test_headline_config2_windows(gpu_state_factory, oracle):
    """BASELINE config 2 exactly: cornell_box_diffuse.obj, 1920x1080, 8 x 128 = 1024 spp, maxDepth 8, IS + DL."""
    w = {k: v for k, v in HEADLINE_WINDOWS.items() if k not in ("glass sphere", "metal mesh")}
    w["short box top"] = (700, 180, 48, 48)
    _headline_check(gpu_state_factory, oracle, SCENE_DIFFUSE, 8, 8, w, "config 2")


def test_headline_config3_windows(gpu_state_factory, oracle):
    """BASELINE config 3's exact setting (cornell_box.obj with the refractive sphere and the conductor mesh, IS + DL,
    maxDepth 16) at 1920x1080; 8 of its 32 steps (1024 of 4096 spp) keep the CPU side of the test within a minute."""
    w = {k: HEADLINE_WINDOWS[k] for k in ("glass sphere", "metal mesh", "light edge")}
    w = {k: (x, y, 32, 32) for k, (x, y, _, _) in w.items()}
    _headline_check(gpu_state_factory, oracle, SCENE_FULL, 16, 8, w, "config 3")


def test_config3_full_shape(gpu_state_factory, oracle):
    """BASELINE config 3 in its real shape: cornell_box.obj (glass sphere + conductor mesh), 1920x1080, 32 steps of 128 spp =
    4096 spp as 4 kernel launches of 8 steps (what `bench.py --config 3` times), maxDepth 16, IS + DL.  The oracle cannot
    follow at this size, so: (a) counter identities over the whole run; (b) the accumulation after the first 8 steps is the
    8-step render's (test_headline_config3_windows checks that one against the oracle) and every later batch continues the
    running mean: an 8 + 24 split equals the straight 32; (c) one 32 x 32 window on the glass sphere against the oracle at
    all 4096 spp."""
    import oracle_lib
    state, obj = gpu_state_factory(SCENE_FULL, sample_chunks=0, width=64, height=64)
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    W, H, S, D, F = 1920, 1080, 128, 16, 32
    p = make_params(W, H, S, D, True, True)
    acc8, _, st8 = _gpu_render(state, p, frames=8, fuse=8)
    # continue the same accumulation with steps 8 .. 31 (three more launches of 8)
    ob = pt.OutputBuffer(pt.OutputBufferType.DEVICE, W, H, state)
    st = list(st8)
    for first in (8, 16, 24):
        state.params.currentFrameIdx = first
        pt.LaunchCurrentFrame(ob, state, 8)
        st.append(pt.getStats(state))
    acc32 = pt.readAccumulation(state)
    fb32 = ob.getHostPointer().copy()
    ob.free()
    straight, fb_straight, st_s = _gpu_render(state, p, frames=F, fuse=8)
    assert len(st) == len(st_s) == 4
    assert sum(int(t.paths) for t in st_s) == W * H * S * F
    for a, b in zip(st, st_s):
        assert (int(a.radiance_rays), int(a.shadow_rays), int(a.paths), int(a.culled_rays)) == (int(b.radiance_rays), int(b.shadow_rays), int(b.paths), int(b.culled_rays))
    assert np.isfinite(straight).all() and np.all(straight[..., 3] == 1.0)
    assert np.array_equal(acc32.view(np.uint32), straight.view(np.uint32)) and np.array_equal(fb32, fb_straight)
    assert not np.array_equal(acc8.view(np.uint32), straight.view(np.uint32))
    # camera rays that end at the scene box: the 16:9 frame around the square box, 46.7 % of the pixels' rays
    culled = sum(int(t.culled_rays) for t in st_s)
    assert 0.44 < culled / float(W * H * S * F) < 0.49
    x0, y0, ww, wh = 700, 180, 32, 32
    chunks = int(st_s[0].sample_chunks)
    ref = None
    for f in range(F):
        q = copy_params(p); q.currentFrameIdx = f
        ref, _, _ = oracle_lib.render_window(sc, q, (x0, y0, ww, wh), accumulation=ref, chunks=chunks)
    a, r = straight[y0:y0 + wh, x0:x0 + ww], ref[y0:y0 + wh, x0:x0 + ww]
    mse = image_mse(a, r)
    same = float(np.all(a.view(np.uint32) == r.view(np.uint32), axis=-1).mean())
    print("config 3, glass sphere, 4096 spp: MSE %.3e, %.1f%% pixels bit-identical, mean %.4f" % (mse, 100 * same, float(r[..., :3].mean())))
    assert mse < MSE_TOL and r[..., :3].mean() > 1e-3
    sc.close()


def test_light_mode_scene_lights_and_mis(full, diffuse):
    """SURVEY.md section 8 f4, opt-in: pt_set_light_mode(1) takes the area light from the scene's emissive triangles and
    combines light and BSDF sampling by the power heuristic.  (a) GPU against the oracle's twin of the same estimator, every
    toggle combination, both scenes; (b) what the mode is for: direct lighting on / off and importance sampling on / off
    converge to the same image, which the reference's estimator (mode 0) does not; (c) mode 0 is untouched."""
    L = _native.hip()
    for name, (state, obj, sc) in (("diffuse", diffuse), ("glass + metal", full)):
        p0 = make_params(96, 72, 8, 8, True, True)
        before, _, _ = _gpu_render(state, p0)                      # mode 0
        n_lights = sc.set_light_mode(1)
        assert n_lights == 2                                       # the ceiling quad of the OBJ
        try:
            assert L.pt_set_light_mode(state.context, 1) == 0
            for dl, isamp in ((True, True), (False, True), (True, False), (False, False)):
                p = make_params(96, 72, 8, 8, dl, isamp)
                acc, _, st = _gpu_render(state, p)
                assert b"LIGHTS" in L.pt_variant_name(int(st[0].variant))
                ref, _, ref_st, _ = sc.render(copy_params(p), use_bvh=True)
                mse = image_mse(acc, ref)
                same = float(np.all(acc.view(np.uint32) == ref.view(np.uint32), axis=-1).mean())
                print("light mode 1, %s, DL %d IS %d: MSE %.3e, %.1f%% pixels bit-identical" % (name, dl, isamp, mse, 100 * same))
                assert np.isfinite(acc).all() and mse < MSE_TOL and same > 0.60
                assert abs(int(st[0].radiance_rays) - ref_st["radiance_rays"]) <= 2e-3 * ref_st["radiance_rays"]
                assert abs(int(st[0].shadow_rays) - ref_st["shadow_rays"]) <= 2e-3 * max(1, ref_st["shadow_rays"])
                with _math(state, "fast"):                         # the LIGHTS kernel's fast-math twin: by tolerance
                    facc, _, fst = _gpu_render(state, p)
                assert b"LIGHTS" in L.pt_variant_name(int(fst[0].variant)) and fst[0].math_mode == _native.MATH_FAST and fst[0].paths == st[0].paths
                assert np.isfinite(facc).all()
                if isamp:
                    assert image_mse(facc, ref) < MSE_TOL, (name, dl, isamp, image_mse(facc, ref))
                else:
                    rep = assert_uniform_mode_fast(facc, ref, 8, dl, "light mode 1 %s DL %d" % (name, dl))
                    assert rep["mse"] < 2e-4, rep
            if name == "diffuse":                                  # (b) 48 x 36 pixels x 1024 spp: image means
                means = {}
                for mode in (1, 0):
                    assert L.pt_set_light_mode(state.context, mode) == 0
                    for dl, isamp in ((True, True), (False, True), (True, False)):
                        acc, _, _ = _gpu_render(state, make_params(48, 36, 256, 12, dl, isamp), frames=4)
                        means[(mode, dl, isamp)] = float(acc[..., :3].mean())
                print("image means:", {k: round(v, 4) for k, v in means.items()})
                m1 = [means[(1, True, True)], means[(1, False, True)], means[(1, True, False)]]
                assert max(m1) / min(m1) < 1.03, "mode 1: the toggles must agree within noise"
                assert means[(0, True, True)] / means[(0, False, True)] > 1.2, "mode 0 counts the light twice with direct lighting on"
            assert L.pt_set_light_mode(state.context, 2) != 0
        finally:
            assert L.pt_set_light_mode(state.context, 0) == 0
            sc.set_light_mode(0)
        after, _, _ = _gpu_render(state, p0)                       # (c)
        assert np.array_equal(before.view(np.uint32), after.view(np.uint32))


def test_full_size_properties(diffuse, both_modes):
    """BASELINE config-2 geometry at its full 1920x1080 / 128 spp per launch: the oracle cannot follow
    at this size, so: (a) the segment-synchronous kernel and the default kernel agree bit for bit,
    counters included; (b) the 8-way tile partition sums to the whole image bit for bit; (c) counter
    identities hold (paths = W*H*spp, pixels = W*H); (d) two consecutive frames follow the running-mean
    rule exactly: acc_1 == lerp(acc_0, frame_1_alone, 1/2); (e) a batch of 8 steps in one kernel launch (what
    bench.py times) leaves the same bits as 8 launches."""
    state, obj, _ = diffuse
    L = _native.hip()
    W, H, S = 1920, 1080, 128
    p = make_params(W, H, S, 8, True, True)
    try:
        assert L.pt_set_sample_chunks(state.context, 0) == 0
        acc, fb, st = _gpu_render(state, p)
        s = st[0]
        assert s.paths == W * H * S and s.pixels == W * H and s.sample_chunks == 8
        assert s.radiance_rays >= s.paths and 0 < s.shadow_rays < s.radiance_rays
        assert np.isfinite(acc).all() and np.all(acc[..., 3] == 1.0)
        assert 0.05 < float(np.clip(acc[..., :3], 0, 1).mean()) < 0.6
        # (a) reference scheduler, same bits
        assert L.pt_set_tuning(state.context, 0, 0) == 0
        acc0, fb0, st0 = _gpu_render(state, p)
        assert L.pt_set_tuning(state.context, 0, _DEFAULT_VARIANT) == 0
        assert np.array_equal(acc0.view(np.uint32), acc.view(np.uint32)) and np.array_equal(fb0, fb)
        assert (st0[0].radiance_rays, st0[0].shadow_rays) == (s.radiance_rays, s.shadow_rays)
        # (b) 8 ranks' tiles (each picks 32 runs per pixel automatically -> force the whole-image setting)
        assert L.pt_set_sample_chunks(state.context, 8) == 0
        q = make_params(W, H, 16, 8, True, True)                  # lighter: 16 spp
        whole, _, _ = _gpu_render(state, q)
        total = np.zeros_like(whole)
        rays = 0
        for rank in range(8):
            assert L.pt_set_partition(state.context, rank, 8) == 0
            state.refreshAccumulationBuffer = True
            pt.updateState(None, state)
            L.pt_device_memset(state.context, state.params.accumulationBuffer, 0, W * H * 16)
            state.params.currentFrameIdx = 0
            pt.LaunchCurrentFrame(None, state)
            total += pt.readAccumulation(state)
            t = pt.getStats(state)
            assert t.pixels == W * H // 8
            rays += t.radiance_rays
        L.pt_set_partition(state.context, 0, 1)
        assert np.array_equal(total.view(np.uint32), whole.view(np.uint32))
        # (d) progressive accumulation identity on two frames
        two, _, _ = _gpu_render(state, q, frames=2)
        state.refreshAccumulationBuffer = True
        pt.updateState(None, state)
        L.pt_device_memset(state.context, state.params.accumulationBuffer, 0, W * H * 16)
        state.params.currentFrameIdx = 1          # frame 1 over a zero buffer: lerp(0, f1, 1/2) = f1/2 exactly
        pt.LaunchCurrentFrame(None, state)
        half_f1 = pt.readAccumulation(state)[..., :3]
        f1 = half_f1 * np.float32(2.0)
        expect = whole[..., :3] + np.float32(0.5) * (f1 - whole[..., :3])
        assert np.array_equal(expect.astype(np.float32).view(np.uint32), two[..., :3].view(np.uint32))
        # (e) the bench's default shape: 8 steps in one kernel launch == 8 launches, whole image and one rank's tiles
        assert L.pt_set_sample_chunks(state.context, 0) == 0
        for part in ((0, 1), (3, 8)):
            assert L.pt_set_partition(state.context, *part) == 0
            sep, sep_fb, st_sep = _gpu_render(state, q, frames=8)
            one, one_fb, st_one = _gpu_render(state, q, frames=8, fuse=8)
            assert len(st_one) == 1 and st_one[0].paths == sum(t.paths for t in st_sep)
            assert st_one[0].radiance_rays + st_one[0].shadow_rays == sum(t.radiance_rays + t.shadow_rays for t in st_sep)
            assert np.array_equal(sep.view(np.uint32), one.view(np.uint32)) and np.array_equal(sep_fb, one_fb), part
    finally:
        L.pt_set_partition(state.context, 0, 1)
        L.pt_set_tuning(state.context, 0, _DEFAULT_VARIANT)
        L.pt_set_sample_chunks(state.context, 1)


def test_trackball_orbit_changes_the_view(tmp_path):
    """acgpt_main --orbit / --zoom: Trackball -> Camera -> U,V,W -> a different (valid) image."""
    import os
    import subprocess
    from PIL import Image
    exe = os.path.join(os.path.dirname(_native.hip_library_path()), "acgpt_main")
    imgs = []
    for extra in ([], ["--orbit", "60,-30", "--zoom", "2"]):
        out = str(tmp_path / ("v%d.png" % len(imgs)))
        r = subprocess.run([exe, "--obj", SCENE_FULL, "--width", "128", "--height", "96", "--spp-per-launch", "16", "--frames", "1",
                            "--direct-lighting", "--importance-sampling", "--out", out] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        imgs.append(np.asarray(Image.open(out).convert("RGB")).astype(np.float32))
    assert imgs[0].shape == (96, 128, 3) and imgs[0].mean() > 10 and imgs[1].mean() > 10
    assert np.abs(imgs[0] - imgs[1]).mean() > 5.0


def test_lifetime_and_streams():
    """Contexts, scenes and launches can be created, replaced and destroyed repeatedly without leaking device
    memory; launches enqueued on a caller-provided stream (torch's) are ordered with the caller's work."""
    import torch
    L = _native.hip()
    try:
        torch.cuda.init()
    except RuntimeError as e:         # no retry: report what the box looks like and fail
        import subprocess
        diag = subprocess.run("rocminfo | head -40; ls -l /dev/kfd /dev/dri", shell=True, capture_output=True, text=True).stdout
        raise AssertionError("torch.cuda.init() failed: %s\n%s" % (e, diag))
    obj = pt.TinyObjWrapper(SCENE_FULL)
    free0 = None
    for it in range(7):
        if it == 1:          # round 0 pays the runtime's one-time allocations (code objects, queues); measure from here
            torch.cuda.synchronize(); torch.cuda.empty_cache()
            free0, _ = torch.cuda.mem_get_info()
        state = pt.PathTracerState()
        pt.createDeviceContext(state, 0)
        for _ in range(3):
            pt.buildTheAccelarationStructure(state, obj)          # replaces the previous scene
        p = make_params(64, 48, 4, 4, True, True)
        p.handle = state.params.handle
        acc = torch.zeros((48, 64, 4), dtype=torch.float32, device="cuda")
        fb = torch.zeros((48, 64, 4), dtype=torch.uint8, device="cuda")
        p.accumulationBuffer, p.frameBuffer = acc.data_ptr(), fb.data_ptr()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            assert L.pt_set_stream(state.context, C.c_void_p(side.cuda_stream)) == 0
            acc.fill_(7.0)                                            # on `side`, before the launch
            assert L.pt_launch(state.context, C.byref(p)) == 0
            total = acc[..., 3].sum()                                 # on `side`, after the launch
        side.synchronize()
        assert float(total) == 64 * 48                              # alpha 1 everywhere: the fill did not overtake the launch
        assert L.pt_set_stream(state.context, None) == 0
        stale = copy_params(p); stale.handle = 12345
        assert L.pt_launch(state.context, C.byref(stale)) != 0 and b"stale" in L.pt_last_error(state.context)
        state.params.accumulationBuffer = None
        L.pt_destroy(state.context)
        state.context = None
        del acc, fb
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, "device memory leaked: %d MiB" % ((free0 - free1) >> 20)


def test_parallel_reinsertion_is_deterministic_and_prunes(gpu_state_factory, oracle, tmp_path):
    """Build mode 2 above 16 384 triangles: the parallel reinsertion on the device (lbvh_build.hip section 13).  A 30 732-triangle scene built
    twice gives the same tree — the same node visits and triangle tests, lane by lane, for a fixed ray stream (gains and node numbers decide
    which moves win, not the order in which threads arrive) —, fewer of both than the PLOC tree it starts from, and the same hits as
    brute force."""
    import sys
    sys.path.insert(0, pt.SCENES)
    import make_scenes
    path = str(tmp_path / "mid.obj")
    make_scenes.stress_scene(path, n_spheres=24, subdiv=3, mtl_name="mid.mtl")
    L = _native.hip()
    closest = random_rays(150000, 51, lo=(20, 20, 20), hi=(530, 530, 540))
    anyr = random_rays(50000, 52, lo=(20, 20, 20), hi=(530, 530, 540), tmin=0.01, tmax=300.0); anyr[:, 7] *= -1.0
    rays = np.ascontiguousarray(np.concatenate([closest, anyr]).astype(np.float32))
    n = rays.shape[0]
    res = {}
    for name, mode in (("ploc", 1), ("reinserted", 2), ("reinserted again", 2)):
        state, obj = gpu_state_factory(path, width=64, height=64, build_mode=mode)
        info = pt.getBvhInfo(state)
        assert info.n_tris == 24 * 1280 + 12 > 16384 and info.max_depth < info.stack_entries
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32); ms = C.c_float(); cnt = (C.c_uint64 * 5)()
        assert L.pt_bench_traversal(state.context, rays.ctypes.data, n, 1, 0, t.ctypes.data, prim.ctypes.data, C.byref(ms), cnt) == 0
        res[name] = (t, prim, int(cnt[1]), int(cnt[2]), int(info.max_depth))
        t4 = np.zeros(n, np.float32); prim4 = np.zeros(n, np.uint32)         # and through the fp16 nodes the render kernel walks: the same answers
        assert L.pt_bench_traversal(state.context, rays.ctypes.data, n, 1, 4, t4.ctypes.data, prim4.ctypes.data, C.byref(ms), None) == 0
        assert np.array_equal(prim4, prim) and np.array_equal(t4.view(np.uint32), t.view(np.uint32)), name
        if mode == 1:
            sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
            tb, pb = sc.trace_closest(np.ascontiguousarray(rays[:2000]), use_bvh=False)
            sc.close()
    a, b, p = res["reinserted"], res["reinserted again"], res["ploc"]
    assert a[2:] == b[2:], "two builds of the same scene walk different trees: %s vs %s" % (a[2:], b[2:])
    for r in (a, b):
        assert np.array_equal(r[1], p[1]) and np.array_equal(r[0].view(np.uint32), p[0].view(np.uint32))
    assert np.array_equal(p[1][:2000], pb) and np.array_equal(p[0][:2000].view(np.uint32), tb.view(np.uint32))
    print("node visits / triangle tests summed over %d rays: PLOC %d / %d (height %d), after the parallel reinsertion %d / %d (height %d)" % (n, p[2], p[3], p[4], a[2], a[3], a[4]))
    assert a[2] < 0.99 * p[2] and a[3] <= p[3]


def test_thousands_of_coincident_triangles_build_and_answer(gpu_state_factory, oracle, tmp_path):
    """17 000 copies of one triangle (one box, one Morton code) plus a second triangle behind them: clustering by merged area, ties to the
    lower index, chains such a scene into a tree as deep as it is long, which the lane stacks cannot walk.  The builder notices that the merges
    stall and pairs ties by a hash (k_ploc_nn), and pt_set_scene would take the radix tree (equal codes split by index bits) rather than fail
    on a tree deeper than 128 levels.  Every ray that hits reports the FIRST copy (ties in t go to the lower primitive index, as in the
    oracle's brute force), any-hit agrees, and the render kernels' own node array gives the same answers."""
    path = str(tmp_path / "same.obj")
    with open(path, "w") as f:
        f.write("mtllib same.mtl\nusemtl white\nv 100 100 300\nv 400 100 300\nv 100 400 300\nv 0 0 500\nv 556 0 500\nv 0 548 500\n")
        f.write("f 1 2 3\n" * 17000 + "f 4 5 6\n")
    with open(str(tmp_path / "same.mtl"), "w") as f:
        f.write("newmtl white\nKd 0.7 0.7 0.7\n")
    L = _native.hip()
    state, obj = gpu_state_factory(path, width=64, height=64)
    info = pt.getBvhInfo(state)
    assert info.n_tris == 17001 and info.max_depth < info.stack_entries <= 128, (info.max_depth, info.stack_entries)
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    rays = random_rays(600, 61, lo=(50, 50, 0), hi=(450, 450, 250))
    rays[:, 3:6] = (rays[:, 3:6] * np.float32([0.3, 0.3, 0.0]) + np.float32([0, 0, 1])); rays[:, 3:6] /= np.linalg.norm(rays[:, 3:6], axis=1, keepdims=True)
    rays = np.ascontiguousarray(rays.astype(np.float32)); n = rays.shape[0]
    t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32); hit = np.zeros(n, np.uint8)
    assert L.pt_trace_closest(state.context, rays.ctypes.data, n, t.ctypes.data, prim.ctypes.data) == 0
    assert L.pt_trace_any(state.context, rays.ctypes.data, n, hit.ctypes.data) == 0
    t_ref, p_ref = sc.trace_closest(rays, use_bvh=False)
    assert np.array_equal(prim, p_ref) and np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    assert (prim == 0).sum() > 50 and (prim == 17000).sum() > 50 and np.array_equal(hit != 0, prim != 0xFFFFFFFF)
    ts = np.zeros(n, np.float32); prims = np.zeros(n, np.uint32); ms = C.c_float()
    assert L.pt_bench_traversal(state.context, rays.ctypes.data, n, 1, 4, ts.ctypes.data, prims.ctypes.data, C.byref(ms), None) == 0      # the render kernels' own nodes
    assert np.array_equal(prims, p_ref) and np.array_equal(ts.view(np.uint32), t_ref.view(np.uint32))
    sc.close()


def test_flat_scene_hits(gpu_state_factory, oracle, tmp_path):
    """A scene with no extent on one axis — 45 000 triangles of a grid in the plane y = 50 —: the fp16 nodes' y axis holds nothing but the
    boxes' pad (its scale is 2^19 / 1023 times the others'), the tree is optimised on the device, and rays from both sides, grazing ones
    included, hit what brute force says, through the query kernels and through the render kernels' own node array."""
    k = 150
    path = str(tmp_path / "flat.obj")
    with open(path, "w") as f:
        f.write("mtllib flat.mtl\nusemtl white\n")
        for j in range(k + 1):
            for i in range(k + 1):
                f.write("v %r 50 %r\n" % (i * 500.0 / k, j * 500.0 / k))
        for j in range(k):
            for i in range(k):
                a = j * (k + 1) + i + 1
                f.write("f %d %d %d\nf %d %d %d\n" % (a, a + 1, a + k + 2, a, a + k + 2, a + k + 1))
    with open(str(tmp_path / "flat.mtl"), "w") as f:
        f.write("newmtl white\nKd 0.7 0.7 0.7\n")
    L = _native.hip()
    state, obj = gpu_state_factory(path, width=64, height=64)
    info = pt.getBvhInfo(state)
    assert info.n_tris == 2 * k * k and info.max_depth < info.stack_entries
    sc = oracle.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    rng = np.random.default_rng(71)
    n = 400
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0] = rng.uniform(-50, 550, n); rays[:, 2] = rng.uniform(-50, 550, n)
    rays[:, 1] = 50 + rng.choice([-1.0, 1.0], n) * np.exp(rng.uniform(np.log(1e-2), np.log(300.0), n))        # a hair above / below the plane up to far away
    tgt = np.stack([rng.uniform(0, 500, n), np.full(n, 50.0), rng.uniform(0, 500, n)], axis=1)
    d = tgt - rays[:, 0:3]
    rays[:, 3:6] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 6] = 0.01; rays[:, 7] = 1e16
    rays = np.ascontiguousarray(rays)
    t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32); ms = C.c_float()
    assert L.pt_trace_closest(state.context, rays.ctypes.data, n, t.ctypes.data, prim.ctypes.data) == 0
    t_ref, p_ref = sc.trace_closest(rays, use_bvh=False)
    assert np.array_equal(prim, p_ref) and np.array_equal(t.view(np.uint32), t_ref.view(np.uint32))
    assert (prim != 0xFFFFFFFF).mean() > 0.9
    ts = np.zeros(n, np.float32); prims = np.zeros(n, np.uint32)
    assert L.pt_bench_traversal(state.context, rays.ctypes.data, n, 1, 4, ts.ctypes.data, prims.ctypes.data, C.byref(ms), None) == 0
    assert np.array_equal(prims, p_ref) and np.array_equal(ts.view(np.uint32), t_ref.view(np.uint32))
    sc.close()


def test_ray_stream_kernel_bit_exact(full):
    """pt_bench_traversal (persistent ray-stream kernel, closest and any-hit rays mixed in one launch,
    more rays than resident lanes so the in-loop refill runs) against brute force."""
    state, obj, sc = full
    v, idx = scene_arrays(obj)
    closest = np.concatenate([random_rays(300000, 41), adversarial_rays(v, idx, 42)])
    anyr = random_rays(250000, 43, tmin=0.01, tmax=250.0)
    rays = np.concatenate([closest, anyr]).astype(np.float32)
    rays[closest.shape[0]:, 7] *= -1.0                      # negative tmax marks an any-hit ray
    perm = np.random.default_rng(44).permutation(rays.shape[0])
    rays = np.ascontiguousarray(rays[perm])
    n = rays.shape[0]
    L = _native.hip()
    is_c = rays[:, 7] > 0
    t_ref, prim_ref = sc.trace_closest(np.ascontiguousarray(rays[is_c]), use_bvh=False)
    ar = np.ascontiguousarray(rays[~is_c]); ar[:, 7] *= -1.0
    any_ref = sc.trace_any(ar, use_bvh=False) != 0
    for fmt in (0, 1, 2, 3, 4):                              # two-child fp32 tree, four-wide 8-bit tree, two-child with the fma slab test, fp16 {lo, hi} nodes, fp16 {centre, half extent} nodes (what the default render kernels walk)
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32); ms = C.c_float()
        assert L.pt_bench_traversal(state.context, rays.ctypes.data, n, 2, fmt, t.ctypes.data, prim.ctypes.data, C.byref(ms), None) == 0
        assert np.array_equal(prim[is_c], prim_ref) and np.array_equal(t[is_c].view(np.uint32), t_ref.view(np.uint32)), fmt
        assert np.array_equal(prim[~is_c] != 0, any_ref), fmt
        assert ms.value > 0
    assert L.pt_bench_traversal(state.context, rays.ctypes.data, 0, 1, 0, t.ctypes.data, prim.ctypes.data, C.byref(ms), None) != 0   # empty stream: refused



