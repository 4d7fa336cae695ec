"""The GPU's own building blocks against the REFERENCE's golden vectors, directly (no oracle in between).

pt_selftest runs the device functions that pt_device.h / pt_shading.h inline into the render kernel on the
inputs of tests/golden/reference_vectors.npz, which tests/golden/make_golden.py generated from the reference's
host-compilable sources (cuda/random.h, cuda/helpers.h, sutil/vec_math.h, sutil/WorkDistribution.h).
Integer work and fp32 arithmetic made of IEEE-exact operations are compared bit for bit; make_color goes through
powf (ROCm's vs glibc's), so it gets the 8-bit bar: equal, except a stated handful of values one step apart."""
import ctypes as C
import os

import numpy as np
import pytest

import acgpathtracing_amd as pt
from acgpathtracing_amd import _native

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_vectors.npz"))


@pytest.fixture(scope="module")
def ctx(built):
    state = pt.PathTracerState()
    pt.createDeviceContext(state, 0)
    yield state.context
    _native.hip().pt_destroy(state.context)


def run(ctx, op, inp, n, out):
    inp = np.ascontiguousarray(inp)
    L = _native.hip()
    assert L.pt_selftest(ctx, op, inp.ctypes.data, n, out.ctypes.data) == 0, L.pt_last_error(ctx)
    return out


def same_bits_or_both_nan(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def test_gpu_tea4_and_lcg_bit_exact(ctx):
    pairs = np.ascontiguousarray(G["tea_in"], np.uint32)
    got = run(ctx, 0, pairs, pairs.shape[0], np.zeros(pairs.shape[0], np.uint32))
    assert np.array_equal(got, G["tea_out"])
    ka = run(ctx, 0, np.array([[0, 0], [12345, 7]], np.uint32), 2, np.zeros(2, np.uint32))
    assert ka.tolist() == [1576399551, 1964180806]                       # SURVEY.md §8c known answers
    for seed, st, va in zip(G["rnd_seeds"], G["rnd_states"], G["rnd_values"]):
        out = run(ctx, 1, np.array([int(seed), st.size], np.uint32), 1, np.zeros(2 * st.size, np.uint32))
        assert np.array_equal(out[:st.size], st)
        assert np.array_equal(out[st.size:], np.ascontiguousarray(va, np.float32).view(np.uint32))


def test_gpu_vec_math_and_refract_bit_exact(ctx):
    A, B, Cc, S = G["vec_a"], G["vec_b"], G["vec_c"], G["vec_s"]
    n = A.shape[0]
    rec = np.concatenate([A, B, Cc, S.reshape(-1, 1)], axis=1).astype(np.float32)
    for op, name in enumerate(["normalize", "reflect", "faceforward", "lerp", "cross", "divide"]):
        got = run(ctx, 3 + op, rec, n, np.zeros((n, 3), np.float32))
        assert same_bits_or_both_nan(got, G["vec_" + name]), name
    m = G["refract_i"].shape[0]
    rec = np.concatenate([G["refract_i"], G["refract_n"], G["refract_ior"].reshape(-1, 1)], axis=1).astype(np.float32)
    out = run(ctx, 9, rec, m, np.zeros((m, 4), np.float32))
    ok = out[:, 3].view(np.uint32) != 0
    assert np.array_equal(ok, G["refract_ok"] != 0)
    assert same_bits_or_both_nan(out[ok, :3], G["refract_r"][ok])       # r is only defined when refraction happens


def test_gpu_make_color(ctx):
    c = np.ascontiguousarray(G["color_in"], np.float32)
    got = run(ctx, 2, c, c.shape[0], np.zeros(c.shape[0], np.uint32)).view(np.uint8).reshape(-1, 4)
    want = G["color_out"]
    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
    assert d.max() <= 1, "sRGB quantisation more than one step from the reference"
    assert (d != 0).mean() < 2e-3, "fraction of channels one step apart: %.2e" % (d != 0).mean()
    assert np.all(got[:, 3] == 255)
    ka = run(ctx, 2, np.array([[0, 0.18, 1.5]], np.float32), 1, np.zeros(1, np.uint32)).view(np.uint8)
    assert tuple(ka) == (0, 118, 255, 255)


def test_gpu_work_distribution_exact(ctx):
    off = 0
    for (world, w, h), ns in zip(G["wd_cases"], G["wd_num_samples"]):
        m = G["wd_maps"][off:off + world * ns * 2].reshape(world, ns, 2); off += world * ns * 2
        for r in range(world):
            q = np.zeros((ns, 4), np.int32); q[:, 0] = world; q[:, 1] = w; q[:, 2] = r; q[:, 3] = np.arange(ns)
            got = run(ctx, 10, q, ns, np.zeros((ns, 2), np.int32))
            assert np.array_equal(got, m[r].astype(np.int32)), (world, w, h, r)


def test_gpu_sincos_equals_sin_and_cos(ctx):
    """The samplers take sin and cos of an angle from one sincosf call (one argument reduction instead of two).  That is
    only admissible if the pair is, bit for bit, what sinf(x) and cosf(x) return: checked over the ranges the samplers
    use (theta in [0, pi/2], phi in [0, 2 pi)), a general range, and special values."""
    rng = np.random.default_rng(5)
    xs = np.concatenate([
        np.linspace(0.0, 2.0 * np.pi, 4_000_001, dtype=np.float64).astype(np.float32),
        (rng.random(4_000_000) * (2.0 * np.pi)).astype(np.float32),
        np.arccos(np.sqrt(rng.random(4_000_000))).astype(np.float32),
        (rng.standard_normal(2_000_000) * 1e3).astype(np.float32),
        (rng.standard_normal(1_000_000) * 1e7).astype(np.float32),
        np.array([0.0, -0.0, 1e-30, -1e-30, 1e-45, np.pi, np.pi / 2, 2 * np.pi, 3.4e38, -3.4e38, np.inf, -np.inf, np.nan], np.float32)])
    out = run(ctx, 11, xs, xs.shape[0], np.zeros((xs.shape[0], 4), np.float32))
    assert same_bits_or_both_nan(out[:, 2], out[:, 0]), "sincosf's sine differs from sinf"
    assert same_bits_or_both_nan(out[:, 3], out[:, 1]), "sincosf's cosine differs from cosf"
    fin = np.isfinite(xs)
    assert np.allclose(out[fin & (np.abs(xs) < 1e4), 0], np.sin(xs[fin & (np.abs(xs) < 1e4)].astype(np.float64)), atol=2e-7)


def ulp_distance(a, b):
    """distance in units of the last place of the larger magnitude (0 for equal bits / both NaN)"""
    a = np.ascontiguousarray(a, np.float32).astype(np.float64); b = np.ascontiguousarray(b, np.float32).astype(np.float64)
    sp = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)
    d = np.abs(a - b) / sp
    return np.where(np.isnan(a) & np.isnan(b), 0.0, d)


def test_gpu_ptprog_math(ctx):
    """The device versions of the OptiX-free helpers of pathTracerPrograms.cu (pt_shading.h: the functions shade_hit
    inlines) against vectors from the reference's own text.  IEEE-only arithmetic (ONB, safeDivide, both Fresnel terms):
    bit for bit.  The three samplers go through libm (acosf / sinf / cosf: ROCm OCML here, glibc in the vectors) and
    return unit vectors built from intermediate angles in [0, 2 pi): a libm result that is 1 ulp off moves the angle by
    up to 2^-23 (ulp of a value in [1, 2)), hence a component by the same absolute amount however small that component
    is (cos(acos(x)) near x = 0 is the extreme case).  Bar: every component within 2 ulp of itself, or within 2^-22
    absolute = 2 ulp of the angle it was computed from."""
    n = G["math_onb_n"].shape[0]
    cat = lambda *cols: np.concatenate([np.asarray(c, np.float32).reshape(n, -1) for c in cols], axis=1)
    exact = [
        (12, cat(G["math_onb_n"], G["math_onb_p"]), 3, G["math_onb_out"], "OrthonormalBasis"),
        (13, cat(G["math_sdiv_a"], G["math_sdiv_b"]), 3, G["math_sdiv3_out"], "safeDivide"),
        (17, cat(G["math_fc_cos"], G["math_fc_eta"], G["math_fc_k"]), 3, G["math_fc_out"], "fresnelSchlickConductor"),
        (18, cat(G["math_fd_cos"], G["math_fd_etai"], G["math_fd_etat"]), 1, G["math_fd_out"].reshape(-1, 1), "FrDielectric"),
    ]
    for op, rec, ow, want, name in exact:
        got = run(ctx, op, rec, n, np.zeros((n, ow), np.float32))
        assert same_bits_or_both_nan(got, want), name
    libm = [
        (14, cat(G["math_u1"], G["math_u2"]), G["math_cosine_out"], "cosine_sample_hemisphere"),
        (15, cat(G["math_u1"], G["math_u2"]), G["math_uniform_out"], "uniform_sample_hemisphere"),
        (16, cat(G["math_u1"], G["math_u2"], G["math_ggx_rough"], G["math_onb_n"]), G["math_ggx_out"], "sampleGGX"),
    ]
    for op, rec, want, name in libm:
        got = run(ctx, op, rec, n, np.zeros((n, 3), np.float32))
        d = ulp_distance(got, want)
        absd = np.abs(got.astype(np.float64) - want.astype(np.float64))
        near = absd <= 2.0 ** -22
        ok = np.isfinite(want).all(axis=1)           # u2 = 1 in sampleGGX divides by zero in the reference too: NaN rows compare as NaN
        assert np.all((d[ok] <= 2.0) | near[ok]), "%s: max %.1f ulp" % (name, d[ok].max())
        assert same_bits_or_both_nan(got[~ok], want[~ok]) or np.isnan(got[~ok]).any(axis=1).all(), name
        print("%s: %.1f%% of components bit-identical, %.2f%% beyond 2 ulp of themselves, max |diff| %.3e" %
              (name, 100 * (d[ok] == 0).mean(), 100 * (d[ok] > 2).mean(), absd[ok].max()))


def test_gpu_fast_math_error_bounds(ctx):
    """PT_MATH_FAST (pt_set_math_mode, the library's default): the shading helpers with v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 /
    v_sin_f32 / v_cos_f32 — the kind of arithmetic nvcc --use_fast_math gives the reference's own build
    (/root/reference/CMakeLists.txt:267).  nvcc's approximate instructions cannot be reproduced here, so there are no golden
    vectors for this level; it is held against the SAME reference vectors as the IEEE level (test_gpu_ptprog_math), by error
    bounds: each approximate instruction is within 1 ulp, a quotient a * rcp(b) within 2, and the helpers' results stay
    within a few 1e-7 of the reference's values (components of unit vectors, Fresnel terms in [0, 1])."""
    n = G["math_onb_n"].shape[0]
    cat = lambda *cols: np.concatenate([np.asarray(c, np.float32).reshape(n, -1) for c in cols], axis=1)
    rng = np.random.default_rng(31)
    # primitives (op 30): a / b, sqrt(|a|), normalize((a, b, 1)) against float64
    m = 1 << 18
    a = (rng.standard_normal(m) * np.exp(rng.uniform(-20, 20, m))).astype(np.float32)
    b = (rng.standard_normal(m) * np.exp(rng.uniform(-20, 20, m))).astype(np.float32)
    b[b == 0] = 1.0
    got = run(ctx, 30, np.stack([a, b], axis=1), m, np.zeros((m, 4), np.float32))
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    q = (a64 / b64).astype(np.float32)
    ok = np.isfinite(q) & (np.abs(q) > 1e-30) & (np.abs(b) > 1e-30) & (np.abs(b) < 1e30)       # denormal reciprocals / quotients: out of the shading code's range
    assert ulp_distance(got[ok, 0], q[ok]).max() <= 2.0, ulp_distance(got[ok, 0], q[ok]).max()
    r = np.sqrt(np.abs(a64)).astype(np.float32)
    assert ulp_distance(got[:, 1], r).max() <= 1.0
    ln = np.sqrt(a64 * a64 + b64 * b64 + 1.0)
    fin = np.isfinite(a64 * a64 + b64 * b64) & (a64 * a64 + b64 * b64 < 1e37)
    assert np.abs(got[fin, 2] - a64[fin] / ln[fin]).max() <= 3e-7 and np.abs(got[fin, 3] - b64[fin] / ln[fin]).max() <= 3e-7
    # sin / cos of 2 pi u (op 31): the hardware instructions against float64, and the IEEE-level pair beside them
    u = np.concatenate([rng.random(m - 4).astype(np.float32), np.array([0.0, 0.25, 0.5, 0.99999994], np.float32)])
    got = run(ctx, 31, u, m, np.zeros((m, 4), np.float32))
    ang = 2.0 * np.pi * u.astype(np.float64)
    hw = max(np.abs(got[:, 0] - np.sin(ang)).max(), np.abs(got[:, 1] - np.cos(ang)).max())
    lib = max(np.abs(got[:, 2] - np.sin(ang)).max(), np.abs(got[:, 3] - np.cos(ang)).max())
    print("sin / cos of 2 pi u: max |error| hardware %.3e, libm path %.3e (the libm path rounds 2 pi u to fp32 first)" % (hw, lib))
    assert hw <= 1e-6 and lib <= 1e-6
    # the helpers at the fast level (ops 32..38) against the reference's own vectors
    cases = [
        (32, cat(G["math_onb_n"], G["math_onb_p"]), 3, G["math_onb_out"], "OrthonormalBasis", 4e-7),
        (33, cat(G["math_sdiv_a"], G["math_sdiv_b"]), 3, G["math_sdiv3_out"], "safeDivide x 3 (roulette)", 4e-7),
        # sin(acos(sqrt(e1))) as sqrt(1 - e1): the reference's own form loses digits where sqrt(e1) rounds next to 1 (theta ~ 0: acos
        # of a value 1 ulp off), so its vectors carry up to 1e-4 of their own rounding there; held against them loosely here and
        # against the exact values below
        (34, cat(G["math_u1"], G["math_u2"]), 3, G["math_cosine_out"], "cosine_sample_hemisphere", 3e-4),
        (35, cat(G["math_u1"], G["math_u2"]), 3, G["math_uniform_out"], "uniform_sample_hemisphere", 1e-6),
        # sinTheta = sqrt(1 - cosTheta^2) next to cosTheta = 1 (small u2) multiplies the quotient's 2 ulp by cos / sin: the form's own conditioning
        (36, cat(G["math_u1"], G["math_u2"], G["math_ggx_rough"], G["math_onb_n"]), 3, G["math_ggx_out"], "sampleGGX", 5e-5),
        (37, cat(G["math_fc_cos"], G["math_fc_eta"], G["math_fc_k"]), 3, G["math_fc_out"], "fresnelSchlickConductor", 4e-7),
        (38, cat(G["math_fd_cos"], G["math_fd_etai"], G["math_fd_etat"]), 1, G["math_fd_out"].reshape(-1, 1), "FrDielectric", 3e-5),      # cosThetaT = sqrt(1 - sinThetaT^2) next to total internal reflection: the same conditioning
    ]
    for op, rec, ow, want, name, tol in cases:
        got = run(ctx, op, rec, n, np.zeros((n, ow), np.float32))
        want = np.asarray(want, np.float32).reshape(n, ow)
        ok = np.isfinite(want).all(axis=1)
        # relative to max(1, |value|): unit-vector components and Fresnel terms are O(1); safeDivide's quotients can be large
        err = np.abs(got[ok].astype(np.float64) - want[ok].astype(np.float64)) / np.maximum(1.0, np.abs(want[ok].astype(np.float64)))
        print("%s at the fast level: max error %.3e (bound %.1e), %.1f%% of components bit-identical to the reference's vectors" %
              (name, err.max(), tol, 100 * (got[ok] == want[ok]).mean()))
        assert err.max() <= tol, (name, err.max())
        assert np.isfinite(got[ok]).all()
        if op == 34:
            e1, e2 = np.asarray(G["math_u1"], np.float64).reshape(-1), np.asarray(G["math_u2"], np.float64).reshape(-1)
            exact = np.stack([np.sqrt(1 - e1) * np.cos(2 * np.pi * e2), np.sqrt(1 - e1) * np.sin(2 * np.pi * e2), np.sqrt(e1)], axis=1)
            ex_err = np.abs(got.astype(np.float64) - exact).max()
            ref_err = np.abs(np.asarray(G["math_cosine_out"], np.float64).reshape(n, 3) - exact).max()
            print("cosine_sample_hemisphere against the exact values: fast level %.3e, the reference's vectors %.3e" % (ex_err, ref_err))
            assert ex_err <= 5e-7


def _exact_slab(o, d, lo, hi, tmin, tmax):
    """Ray / box in float64 on the very fp32 values the device gets: True where the ray meets the box within [tmin, tmax]."""
    o, d, lo, hi = (x.astype(np.float64) for x in (o, d, lo, hi))
    with np.errstate(divide="ignore", invalid="ignore"):
        t1, t2 = (lo - o) / d, (hi - o) / d
    par = d == 0.0                                      # parallel to the slab: inside it or never
    inside = (o >= lo) & (o <= hi)
    tn = np.where(par, np.where(inside, -np.inf, np.inf), np.minimum(t1, t2))
    tf = np.where(par, np.where(inside, np.inf, -np.inf), np.maximum(t1, t2))
    return np.maximum(tn.max(axis=1), tmin) <= np.minimum(tf.min(axis=1), tmax.astype(np.float64))


def test_gpu_fp16_slab_is_conservative(ctx):
    """The default kernel's box test (pt_selftest op 19: outward fp16 planes, rotate flags in the multipliers' low mantissa
    bits, one v_fma_mix per plane) must accept every ray that meets the box as it was before the builder's pad — triangles
    live inside that — for origins in and far outside the scene, grazing, axis-parallel and near-degenerate directions."""
    rng = np.random.default_rng(20261004)
    n = 1 << 21
    centre = np.array([100.0, -50.0, 30.0], np.float32)
    H = np.float32(300.0)
    e = int(np.frexp(H)[1])
    inv_scale = np.float32(2.0 ** (e - 10))
    coord_max = np.float32(np.abs(centre).max() + H)
    pad_abs = np.float32(coord_max / np.float32(524288.0))
    # inner boxes: log-uniform sizes, a quarter flat on one axis (axis-aligned triangles), a quarter touching the scene's rim
    size = (H * np.exp(rng.uniform(np.log(1e-4), 0.0, (n, 3)))).astype(np.float32)
    flat = rng.random(n) < 0.25
    size[flat, rng.integers(0, 3, flat.sum())] = 0.0
    lo0 = (centre - H + rng.random((n, 3), dtype=np.float32) * (2 * H - size)).astype(np.float32)
    rim = rng.random(n) < 0.25
    side = rng.integers(0, 2, (n, 3)).astype(bool)
    lo0 = np.where(rim[:, None] & side, centre + H - size, np.where(rim[:, None], centre - H, lo0)).astype(np.float32)
    hi0 = (lo0 + size).astype(np.float32)
    pad = np.maximum(np.float32(1e-5) * np.maximum(np.float32(1.0), np.maximum(np.abs(lo0), np.abs(hi0))), pad_abs).astype(np.float32)
    lo, hi = (lo0 - pad).astype(np.float32), (hi0 + pad).astype(np.float32)     # what lbvh_build.hip k_prepare hands on
    # origins: three quarters inside the scene, the rest up to 32 scene sizes away
    far = rng.random(n) < 0.25
    o = (centre + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.where(far[:, None], 64 * H, H)).astype(np.float32)
    # directions: aimed at a point of the inner box's surface (corners, edges, faces: grazing rays), some random
    u = rng.random((n, 3), dtype=np.float32)
    snap = rng.integers(0, 3, (n, 3))                   # 0 -> lo plane, 1 -> hi plane, 2 -> somewhere between
    target = np.where(snap == 0, lo0, np.where(snap == 1, hi0, lo0 + u * size)).astype(np.float32)
    d = (target - o).astype(np.float32)
    rnd_dir = rng.random(n) < 0.15
    d[rnd_dir] = rng.normal(size=(int(rnd_dir.sum()), 3)).astype(np.float32)
    d = (d / np.maximum(np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True), 1e-30)).astype(np.float32)
    # axis-parallel and near-degenerate components (incl. -0.0 and denormals)
    for frac, val in ((0.06, 0.0), (0.02, -0.0), (0.02, 1e-30), (0.02, -1e-38), (0.02, 1e-45)):
        m = rng.random(n) < frac
        d[m, rng.integers(0, 3, int(m.sum()))] = np.float32(val)
    keep = np.abs(d).max(axis=1) > 0
    d[~keep] = np.array([0.0, 1.0, 0.0], np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.float32(1e16), (np.linalg.norm((target - o).astype(np.float64), axis=1) * rng.uniform(0.5, 1.5, n)).astype(np.float32)).astype(np.float32)
    must = _exact_slab(o, d, lo0, hi0, 0.01, tmax)
    assert must.mean() > 0.3, "the case generator lost its hits"
    wide = _exact_slab(o, d, lo - np.float32(0.02) * H, hi + np.float32(0.02) * H, 0.01, tmax)
    assert (~wide).mean() > 0.05
    # the scale the builder uses — the scene's farthest plane goes to 1023, just below a power of two, where fp16 is finest
    # relative to the scene (lbvh_build.hip) — and a power-of-two scale (round 2's rule: exact products, coarser planes)
    half_ext = np.float32(np.abs(np.concatenate([lo, hi]) - centre).max())
    for inv_scale in (np.float32(half_ext / np.float32(1023.0)), np.float32(2.0 ** (e - 10))):
        rec = np.concatenate([o, d, lo, hi, np.broadcast_to(centre, (n, 3)), np.full((n, 1), inv_scale, np.float32), tmax[:, None]], axis=1).astype(np.float32)
        assert rec.shape == (n, 17)
        # op 19: {lo, hi} planes, rotated by the ray's sign (NODE_FMT 9); op 40: {centre, half extent} (NODE_FMT 11)
        for op, form in ((19, "lo / hi planes"), (40, "centre / half extent")):
            out = np.zeros((n, 3), np.uint32)
            run(ctx, op, np.ascontiguousarray(rec), n, out)
            accepted = out[:, 0] == 1
            missed = must & ~accepted
            assert not missed.any(), "%s, scale %r: box test rejected %d of %d rays that meet the unpadded box, first: %s" % (form, 1.0 / inv_scale, missed.sum(), must.sum(), rec[np.argmax(missed)])
            # and it is a test, not a constant: rays that miss the box inflated by 2 % of the scene are (almost) never accepted
            assert accepted[~wide].mean() < 1e-3, accepted[~wide].mean()
            print("fp16 slab test, %s, scaled by %.4f: %.2f %% of the rays that miss the box inflated by 2 %% of the scene are accepted, %.1f %% of all rays"
                  % (form, 1.0 / inv_scale, 100 * accepted[~wide].mean(), 100 * accepted.mean()))
    # op 41: {centre, half extent} with a scale PER AXIS, what the builder gives NODE_FMT 11 (every face of the scene box at |g| = 1023).  The same
    # cases in a scene squeezed to 1 / 3 and 1 / 40 of its size on y and z: boxes, origins and targets squeezed about the centre, directions re-aimed
    sq = np.array([1.0, 0.313881, 1.0 / 40.0], np.float32)        # (y: the lower face lands at g = -321.1 under x's scale, between two fp16 values)
    f = lambda p: (centre + (p - centre) * sq).astype(np.float32)
    lo0s, hi0s, os_ = f(lo0), f(hi0), f(o)
    ds = (d * sq).astype(np.float32)
    ds = (ds / np.maximum(np.linalg.norm(ds.astype(np.float64), axis=1, keepdims=True), 1e-30)).astype(np.float32)
    ds[np.abs(ds).max(axis=1) == 0] = np.array([0.0, 1.0, 0.0], np.float32)
    pads = np.maximum(np.float32(1e-5) * np.maximum(np.float32(1.0), np.maximum(np.abs(lo0s), np.abs(hi0s))), pad_abs).astype(np.float32)
    los, his = (lo0s - pads).astype(np.float32), (hi0s + pads).astype(np.float32)
    must = _exact_slab(os_, ds, lo0s, hi0s, 0.01, tmax)
    assert must.mean() > 0.2, "the squeezed cases lost their hits"
    wide = _exact_slab(os_, ds, los - np.float32(0.02) * H * sq, his + np.float32(0.02) * H * sq, 0.01, tmax)
    inv3 = (np.abs(np.concatenate([los, his]) - centre).max(axis=0) / np.float32(1023.0)).astype(np.float32)
    assert inv3[0] > 2.5 * inv3[1] > 25 * inv3[2] > 0
    rec = np.concatenate([os_, ds, los, his, np.broadcast_to(centre, (n, 3)), np.broadcast_to(inv3, (n, 3)), tmax[:, None]], axis=1).astype(np.float32)
    assert rec.shape == (n, 19)
    out = np.zeros((n, 3), np.uint32)
    run(ctx, 41, np.ascontiguousarray(rec), n, out)
    accepted = out[:, 0] == 1
    missed = must & ~accepted
    assert not missed.any(), "centre / half extent, a scale per axis: box test rejected %d of %d rays that meet the unpadded box, first: %s" % (missed.sum(), must.sum(), rec[np.argmax(missed)])
    assert accepted[~wide].mean() < 1e-3, accepted[~wide].mean()
    # and a FLAT scene: everything in the plane z = centre.z (the z axis holds nothing but the boxes' pad: its scale is 2^19 / 1023 times the x axis'),
    # rays through it from both sides, a thousandth of them in the plane
    sqf = np.array([1.0, 0.5, 0.0], np.float32)
    ff = lambda p: (centre + (p - centre) * sqf).astype(np.float32)
    lo0f, hi0f = ff(lo0), ff(hi0)
    of = (centre + (o - centre) * np.array([1.0, 0.5, 0.05], np.float32)).astype(np.float32)
    tgt = ff(target)
    df = (tgt - of).astype(np.float32)
    inplane = rng.random(n) < 1e-3
    df[inplane, 2] = 0.0; of[inplane, 2] = centre[2]
    df = (df / np.maximum(np.linalg.norm(df.astype(np.float64), axis=1, keepdims=True), 1e-30)).astype(np.float32)
    df[np.abs(df).max(axis=1) == 0] = np.array([0.0, 1.0, 0.0], np.float32)
    padf = np.maximum(np.float32(1e-5) * np.maximum(np.float32(1.0), np.maximum(np.abs(lo0f), np.abs(hi0f))), pad_abs).astype(np.float32)
    lof, hif = (lo0f - padf).astype(np.float32), (hi0f + padf).astype(np.float32)
    mustf = _exact_slab(of, df, lo0f, hi0f, 0.01, tmax)
    assert mustf.mean() > 0.1, "the flat cases lost their hits"
    inv3f = (np.abs(np.concatenate([lof, hif]) - centre).max(axis=0) / np.float32(1023.0)).astype(np.float32)
    assert inv3f[2] < 1e-4 * inv3f[0]
    recf = np.concatenate([of, df, lof, hif, np.broadcast_to(centre, (n, 3)), np.broadcast_to(inv3f, (n, 3)), tmax[:, None]], axis=1).astype(np.float32)
    outf = np.zeros((n, 3), np.uint32)
    run(ctx, 41, np.ascontiguousarray(recf), n, outf)
    missedf = mustf & ~(outf[:, 0] == 1)
    assert not missedf.any(), "flat scene, a scale per axis: box test rejected %d of %d rays that meet the unpadded box, first: %s" % (missedf.sum(), mustf.sum(), recf[np.argmax(missedf)])
    # a flat box on a face of the scene box is as thin as its pad: a ray that leaves it at 45 degrees is outside before tmin = 0.01 ... with one
    # scale for all axes (op 40) the face of a short axis lies between two fp16 values and the same ray is still inside
    wall = np.zeros((3, 19), np.float32)
    face_y = np.float32(centre[1] - inv3[1] * 1023)                    # the scene box's lower y face
    mid_u = np.float32((face_y + np.float32(1e-3) - centre[1]) / inv3[0])
    assert np.float32(np.float16(mid_u)) > mid_u + np.float32(0.05)           # under one scale the wall's centre rounds into the scene: the box reaches 0.03 units in
    for k, inv in enumerate((inv3, np.full(3, inv3[0], np.float32))):
        wall[k, 0:3] = (centre[0], face_y + np.float32(3e-3), centre[2]); wall[k, 3:6] = (0.70710678, 0.70710678, 0.0)
        wall[k, 6:9] = (centre[0] - 50, face_y, centre[2] - 1); wall[k, 9:12] = (centre[0] + 50, face_y + np.float32(2e-3), centre[2] + 1)
        wall[k, 12:15] = centre; wall[k, 15:18] = inv; wall[k, 18] = 1e16
    out = np.zeros((3, 3), np.uint32)
    run(ctx, 41, np.ascontiguousarray(wall), 3, out)
    assert out[0, 0] == 0 and out[1, 0] == 1, out[:2]
    print("fp16 slab test, centre / half extent with a scale per axis: %.2f %% of the rays that miss the inflated box are accepted; a wall on a face of the scene box is left before tmin" % (100 * accepted[~wide].mean()))


def test_gpu_shared_plane_slab_is_conservative(ctx):
    """The shared-plane kernel's box test (pt_selftest op 39; NODE_FMT 10: magnitudes measured inward from the root's planes, the sign
    bit naming the owning child, the other child's copy mirrored beyond the root plane) on the case generator of the fp16 slab test:
    no ray that meets the box as it was before the builder's pad is rejected, whether the box is child 0 (magnitudes as stored) or
    child 1 (negated), after the interval's trip over the stack; the sibling that inherits all six planes is the root box and is
    accepted exactly when the root is; and it is a test, not a constant."""
    rng = np.random.default_rng(20261005)
    n = 1 << 21
    centre = np.array([100.0, -50.0, 30.0], np.float32)
    H = np.float32(300.0)
    coord_max = np.float32(np.abs(centre).max() + H)
    pad_abs = np.float32(coord_max / np.float32(524288.0))
    size = (H * np.exp(rng.uniform(np.log(1e-4), 0.0, (n, 3)))).astype(np.float32)
    flat = rng.random(n) < 0.25
    size[flat, rng.integers(0, 3, flat.sum())] = 0.0
    lo0 = (centre - H + rng.random((n, 3), dtype=np.float32) * (2 * H - size)).astype(np.float32)
    rim = rng.random(n) < 0.25
    side = rng.integers(0, 2, (n, 3)).astype(bool)
    lo0 = np.where(rim[:, None] & side, centre + H - size, np.where(rim[:, None], centre - H, lo0)).astype(np.float32)
    hi0 = (lo0 + size).astype(np.float32)
    pad = np.maximum(np.float32(1e-5) * np.maximum(np.float32(1.0), np.maximum(np.abs(lo0), np.abs(hi0))), pad_abs).astype(np.float32)
    lo, hi = (lo0 - pad).astype(np.float32), (hi0 + pad).astype(np.float32)
    far = rng.random(n) < 0.25
    o = (centre + (rng.random((n, 3), dtype=np.float32) * 2 - 1) * np.where(far[:, None], 32 * H, H)).astype(np.float32)
    u = rng.random((n, 3), dtype=np.float32)
    snap = rng.integers(0, 3, (n, 3))
    target = np.where(snap == 0, lo0, np.where(snap == 1, hi0, lo0 + u * size)).astype(np.float32)
    d = (target - o).astype(np.float32)
    rnd_dir = rng.random(n) < 0.15
    d[rnd_dir] = rng.normal(size=(int(rnd_dir.sum()), 3)).astype(np.float32)
    d = (d / np.maximum(np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True), 1e-30)).astype(np.float32)
    for frac, val in ((0.06, 0.0), (0.02, -0.0), (0.02, 1e-30), (0.02, -1e-38), (0.02, 1e-45)):
        m = rng.random(n) < frac
        d[m, rng.integers(0, 3, int(m.sum()))] = np.float32(val)
    keep = np.abs(d).max(axis=1) > 0
    d[~keep] = np.array([0.0, 1.0, 0.0], np.float32)
    tmax = np.where(rng.random(n) < 0.5, np.float32(1e16), (np.linalg.norm((target - o).astype(np.float64), axis=1) * rng.uniform(0.5, 1.5, n)).astype(np.float32)).astype(np.float32)
    must = _exact_slab(o, d, lo0, hi0, 0.01, tmax)
    assert must.mean() > 0.3
    wide = _exact_slab(o, d, lo - np.float32(0.02) * H, hi + np.float32(0.02) * H, 0.01, tmax)
    assert (~wide).mean() > 0.05
    # the root's planes as lbvh_build.hip ensure_srecs sets them: the scene box (here: of all the padded boxes), 1e-4 of its extent + pad_abs outside
    s_lo, s_hi = lo.min(axis=0), hi.max(axis=0)
    ext = (s_hi - s_lo).astype(np.float32)
    rp = (np.maximum(ext, np.float32(1e-30)) * np.float32(1e-4) + pad_abs).astype(np.float32)
    Lr, Hr = (s_lo - rp).astype(np.float32), (s_hi + rp).astype(np.float32)
    inv_scale = np.float32((Hr - Lr).max() / np.float32(2046.0))
    rec = np.concatenate([o, d, lo, hi, np.broadcast_to(Lr, (n, 3)), np.broadcast_to(Hr, (n, 3)), np.full((n, 1), inv_scale, np.float32), tmax[:, None]], axis=1).astype(np.float32)
    assert rec.shape == (n, 20)
    out = np.zeros((n, 3), np.uint32)
    run(ctx, 39, np.ascontiguousarray(rec), n, out)
    for k, name in ((0, "child 0 (magnitudes as stored)"), (1, "child 1 (negated)")):
        accepted = out[:, k] == 1
        missed = must & ~accepted
        assert not missed.any(), "%s: box test rejected %d of %d rays that meet the unpadded box, first: %s" % (name, missed.sum(), must.sum(), rec[np.argmax(missed)])
        assert accepted[~wide].mean() < 2e-3, accepted[~wide].mean()
        print("shared-plane slab test, %s: %.2f %% of the rays that miss the box inflated by 2 %% of the scene are accepted, %.1f %% of all rays"
              % (name, 100 * accepted[~wide].mean(), 100 * accepted.mean()))
    root_ok = (out[:, 2] & 1) == 1
    assert np.array_equal((out[:, 2] & 2) != 0, root_ok) and np.array_equal((out[:, 2] & 4) != 0, root_ok), "the sibling that inherits every plane IS the root box"
    assert not (must & ~root_ok).any()
