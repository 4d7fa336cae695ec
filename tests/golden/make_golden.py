#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.npz from the REFERENCE'S OWN code.

Runs only where /root/reference exists: it loads oracle/_ref/libref.so, which oracle/Makefile
builds from the reference's host-compilable sources where they lie (cuda/random.h, cuda/helpers.h,
sutil/vec_math.h, sutil/Camera.cpp, sutil/Trackball.cpp, sutil/WorkDistribution.h,
PathTracer_Optix/TinyObjWrapper.cpp + util/tiny_obj_loader.h, and the OptiX-free sampling / BSDF helpers of
PathTracer_Optix/pathTracerPrograms.cu that oracle/extract_ptprog_math.py lifts verbatim into oracle/_ref at build time).  The outputs are DATA (inputs and
expected outputs); no reference source text is stored.  Deterministic: fixed numpy seeds.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import ref_lib  # noqa: E402


def main():
    if not ref_lib.available():
        raise SystemExit("oracle/_ref/libref.so missing: run `make -C oracle` where /root/reference exists")
    R = ref_lib.Ref()
    rng = np.random.default_rng(20240607)
    out = {}

    # ---- PRNG: tea<4> and the lcg/rnd stream (cuda/random.h) ---------------------------------
    pairs = np.concatenate([
        np.array([[0, 0], [1, 0], [0, 1], [12345, 7], [0xFFFFFFFF, 0xFFFFFFFF], [1920 * 1080 - 1, 31], [65535, 1]], np.uint64),
        rng.integers(0, 2 ** 32, size=(200, 2), dtype=np.uint64)]).astype(np.uint32)
    out["tea_in"] = pairs
    out["tea_out"] = np.array([R.tea4(int(a), int(b)) for a, b in pairs], np.uint32)
    seeds = np.array([0, 1, 1576399551, 0xFFFFFFFF, 1964180806, 0x9E3779B9], np.uint32)
    st = []; va = []
    for s in seeds:
        a, b = R.rnd_stream(int(s), 64)
        st.append(a); va.append(b)
    out["rnd_seeds"] = seeds
    out["rnd_states"] = np.stack(st)
    out["rnd_values"] = np.stack(va)

    # ---- make_color (cuda/helpers.h) ----------------------------------------------------------------
    special = np.array([[0, 0, 0], [1, 1, 1], [0.0031308, 0.0031307, 0.0031309], [0.18, 0.5, 0.999999], [-1, 2, 1e-8],
                        [255.0 / 256.0, 254.999 / 256.0, 0.5], [np.nan, np.inf, -np.inf], [1e-45, 1e-38, 3.0e-3]], np.float32)
    colors = np.concatenate([special, rng.random((2000, 3), dtype=np.float32) ** 3, (rng.random((500, 3), dtype=np.float32) * 4 - 1)]).astype(np.float32)
    out["color_in"] = colors
    out["color_out"] = R.make_color(colors)

    # ---- float3 algebra (sutil/vec_math.h) and refract (cuda/helpers.h) --------------------------------
    n = 400
    A = (rng.normal(size=(n, 3)) * rng.choice([1e-3, 1.0, 500.0], size=(n, 1))).astype(np.float32)
    B = rng.normal(size=(n, 3)).astype(np.float32)
    Cc = rng.normal(size=(n, 3)).astype(np.float32)
    S = rng.random(n).astype(np.float32) * 3 + np.float32(0.01)
    out["vec_a"], out["vec_b"], out["vec_c"], out["vec_s"] = A, B, Cc, S
    for op, name in enumerate(["normalize", "reflect", "faceforward", "lerp", "cross", "divide"]):
        out["vec_" + name] = np.stack([R.vec_op(op, A[i], B[i], Cc[i], float(S[i])) for i in range(n)])
    I = A / np.linalg.norm(A, axis=1, keepdims=True)
    N = B / np.linalg.norm(B, axis=1, keepdims=True)
    iors = rng.choice([1.0, 1.33, 1.5, 2.4, 0.75], size=n).astype(np.float32)
    rr = [R.refract(I[i].astype(np.float32), N[i].astype(np.float32), float(iors[i])) for i in range(n)]
    out["refract_i"], out["refract_n"], out["refract_ior"] = I.astype(np.float32), N.astype(np.float32), iors
    out["refract_r"] = np.stack([r[0] for r in rr])
    out["refract_ok"] = np.array([r[1] for r in rr], np.uint8)

    # ---- the OptiX-free helpers inside pathTracerPrograms.cu (:54-85, :265-284, :341-380, :455-476, :494-510, :534-559),
    # run from the reference's own text (oracle/extract_ptprog_math.py -> oracle/_ref/ptprog_math.inc -> libref.so) ----
    m = 3000
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    Nn = unit(rng.normal(size=(m, 3)))
    # ONB branch edge |n.x| vs |n.z|, axis-aligned normals, and sampleGGX's |N.z| < 0.999 switch from both sides
    Nn[:6] = np.array([[0, 1, 0], [0, -1, 0], [1, 0, 0], [0, 0, 1], [0, 0, -1], [-1, 0, 0]], np.float32)
    Nn[6:10] = unit(np.array([[0.5, 0.3, 0.5], [0.5, 0.3, -0.5], [-0.5, 0.1, 0.5], [0.5000001, 0.2, 0.5]], np.float64))
    for k, z in enumerate([0.999, 0.99900001, 0.9989999, 0.9990001, -0.999, -0.9990001, -0.9989999, 0.9995, 0.998]):
        s_xy = np.sqrt(max(0.0, 1.0 - z * z))
        Nn[10 + k] = np.array([s_xy * 0.6, s_xy * 0.8, z], np.float32)
    Pp = rng.normal(size=(m, 3)).astype(np.float32)
    out["math_onb_n"], out["math_onb_p"] = Nn, Pp
    out["math_onb_out"] = R.onb_transform(Nn, Pp)
    sa = (rng.normal(size=(m, 3)) * rng.choice([1e-3, 1.0, 1e4], size=(m, 1))).astype(np.float32)
    sb = rng.normal(size=m).astype(np.float32)
    sb[:40:4] = 0.0; sb[1:40:4] = -0.0; sb[2:40:4] = np.float32(1e-42); sb[3:12:4] = np.inf
    out["math_sdiv_a"], out["math_sdiv_b"] = sa, sb
    out["math_sdiv_out"] = R.safe_divide(sa[:, 0], sb)
    out["math_sdiv3_out"] = R.safe_divide3(sa, sb)
    # the samplers' inputs are rnd() outputs: k / 2^24, k in [0, 2^24)
    u1 = (rng.integers(0, 1 << 24, size=m).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)
    u2 = (rng.integers(0, 1 << 24, size=m).astype(np.float32) / np.float32(1 << 24)).astype(np.float32)
    edge = np.array([0.0, 1.0 / (1 << 24), 0.25, 0.5, 0.75, ((1 << 24) - 1) / float(1 << 24)], np.float32)
    u1[:36] = np.repeat(edge, 6); u2[:36] = np.tile(edge, 6)
    out["math_u1"], out["math_u2"] = u1, u2
    out["math_cosine_out"] = R.sample_hemisphere(0, u1, u2)
    out["math_uniform_out"] = R.sample_hemisphere(1, u1, u2)
    rough = np.full(m, 0.2, np.float32); rough[m // 2:] = rng.choice([0.05, 0.5, 1.0], size=m - m // 2).astype(np.float32)
    out["math_ggx_rough"] = rough
    out["math_ggx_out"] = R.sample_ggx(u1, u2, rough, Nn)
    ct = (rng.random(m) * 1.0).astype(np.float32); ct[:5] = [0.0, 1.0, 1e-4, 0.5, 0.9999999]
    eta = np.tile(np.array([1.45, 0.7, 1.55], np.float32), (m, 1)); kk = np.tile(np.array([3.0, 2.2, 3.5], np.float32), (m, 1))
    eta[m // 2:] = (rng.random((m - m // 2, 3)) * 3 + 0.1).astype(np.float32); kk[m // 2:] = (rng.random((m - m // 2, 3)) * 5).astype(np.float32)
    out["math_fc_cos"], out["math_fc_eta"], out["math_fc_k"] = ct, eta, kk
    out["math_fc_out"] = R.fresnel_conductor(ct, eta, kk)
    ci = (rng.random(m) * 2.4 - 1.2).astype(np.float32)          # beyond [-1, 1]: the clamp at :535
    ci[:8] = [0.0, -0.0, 1.0, -1.0, 1e-7, -1e-7, 0.3, -0.3]
    ei = np.ones(m, np.float32); et = rng.choice([1.0, 1.33, 1.5, 2.4, 0.75], size=m).astype(np.float32)
    ei[m // 2:] = rng.choice([1.0, 1.5, 2.4], size=m - m // 2).astype(np.float32)
    out["math_fd_cos"], out["math_fd_etai"], out["math_fd_etat"] = ci, ei, et
    out["math_fd_out"] = R.fr_dielectric(ci, ei, et)
    assert (out["math_fd_out"] == 1.0).mean() > 0.02, "total internal reflection must be covered"

    # ---- Camera::UVWFrame (sutil/Camera.cpp) ---------------------------------------------------------------
    cams = [((278, 273, -900), (278, 273, 330), (0, 1, 0), 35.0, 1.0),
            ((278, 273, -900), (278, 273, 330), (0, 1, 0), 35.0, np.float32(1920) / np.float32(1080)),
            ((278, 273, -900), (278, 273, 330), (0, 1, 0), 35.0, np.float32(100) / np.float32(52)),
            ((1, 1, 1), (0, 0, 0), (0, 1, 0), 35.0, 1.0),
            ((3.5, -2.25, 7), (0.5, 1, -1), (0.1, 0.9, 0.2), 60.0, 1.5),
            ((0, 5, 0), (0.001, 0, 0.002), (0, 0, 1), 90.0, 0.75)]
    cam_in = np.array([list(e) + list(l) + list(u) + [f, a] for e, l, u, f, a in cams], np.float32)
    out["camera_in"] = cam_in
    out["camera_uvw"] = np.stack([np.concatenate(R.camera_uvw(c[0:3], c[3:6], c[6:9], float(c[9]), float(c[10]))) for c in cam_in])

    # ---- Trackball (sutil/Trackball.cpp) ----------------------------------------------------------------------
    scripts = [
        [(0, 100, 100), (1, 130, 90), (1, 180, 140)],
        [(1, 10, 10), (1, 50, 20), (2, 1, 0), (2, 1, 0), (2, -1, 0)],
        [(0, 0, 0), (1, 400, 0), (1, 400, 500), (1, -300, 500), (2, -1, 0)],
    ]
    tb_out = []; tb_cfg = []
    ev_flat = []; ev_len = []
    for si, ev in enumerate(scripts):
        for view_mode in (0, 1):
            for gimbal in (0, 1):
                ci = si % 2 * 4
                c = cam_in[ci]
                tb_cfg.append([ci, view_mode, gimbal])
                tb_out.append(R.trackball_script(c[0:3], c[3:6], c[6:9], float(c[9]), float(c[10]), view_mode, 10.0, gimbal, 512, 512, ev))
                ev_flat += [list(e) for e in ev]; ev_len.append(len(ev))
    out["trackball_cfg"] = np.array(tb_cfg, np.int32)
    out["trackball_events"] = np.array(ev_flat, np.int32)
    out["trackball_event_counts"] = np.array(ev_len, np.int32)
    out["trackball_out"] = np.stack(tb_out)

    # ---- StaticWorkDistribution (sutil/WorkDistribution.h) ------------------------------------------------------------
    wd_cases = np.array([[1, 64, 32], [2, 100, 52], [3, 33, 9], [4, 128, 64], [8, 1920, 16], [8, 250, 37]], np.int32)
    out["wd_cases"] = wd_cases
    maps = []
    for world, w, h in wd_cases:
        ns = R.num_samples(int(world), int(w), int(h))
        m = np.zeros((world, ns, 2), np.int32)
        for r in range(world):
            for si in range(ns):
                m[r, si] = R.sample_pixel(int(world), int(w), int(h), r, si)
        maps.append(m.reshape(-1))
    out["wd_num_samples"] = np.array([R.num_samples(int(a), int(b), int(c)) for a, b, c in wd_cases], np.int32)
    out["wd_maps"] = np.concatenate(maps)

    # ---- TinyObjWrapper (PathTracer_Optix/TinyObjWrapper.cpp over tinyobjloader) -------------------------------------------
    objs = ["tests/golden/obj/quads_ngons.obj", "tests/golden/obj/no_mtl.obj", "tests/golden/obj/exponent_numbers.obj",
            "tests/golden/obj/tricky_order.obj",
            "acgpathtracing_amd/scenes/cornell_box.obj", "acgpathtracing_amd/scenes/cornell_box_diffuse.obj"]
    out["obj_names"] = np.array(objs)
    for k, rel in enumerate(objs):
        r = R.obj_load(os.path.join(ROOT, rel))
        assert r is not None, rel
        v, i, m, mm = r
        out["obj%d_verts" % k] = v
        out["obj%d_idx" % k] = i
        out["obj%d_mat_ids" % k] = m
        out["obj%d_mats" % k] = mm        # n_mats x 10 dwords (Material, 40 bytes), raw bits
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_vectors.npz"), "with", len(out), "arrays")


if __name__ == "__main__":
    main()
