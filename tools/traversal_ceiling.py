#!/usr/bin/env python3
"""How fast could BVH traversal alone go?  Builds a ray set with the render's own mix (camera rays,
cosine-sampled bounce rays from their hit points, shadow rays towards the area light), streams it through
pt_bench_traversal (persistent kernel, nothing but the BVH loop, 60 VGPRs, refill inside the loop) and
compares with the render kernel's rays per second.  Also bit-checks the stream kernel against
pt_trace_closest / pt_trace_any."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import acgpathtracing_amd as pt  # noqa: E402
from acgpathtracing_amd import _native  # noqa: E402
from scene_utils import make_params  # noqa: E402


def trace(L, ctx, rays):
    n = rays.shape[0]
    t = np.zeros(n, np.float32); p = np.zeros(n, np.uint32)
    assert L.pt_trace_closest(ctx, rays.ctypes.data, n, t.ctypes.data, p.ctypes.data) == 0
    return t, p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell_box_diffuse.obj")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--bounces", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5)
    a = ap.parse_args()
    L = _native.hip()
    import bench
    path = bench.scene_path(pt, a.scene)
    state, obj = pt.setup(path, width=64, height=64)
    rng = np.random.default_rng(1)
    p = make_params(a.width, a.height, 1, 8, True, True)
    eye = np.array(p.cameraEye.tuple(), np.float32)
    U, V, W = [np.array(x.tuple(), np.float32) for x in (p.cameraU, p.cameraV, p.cameraW)]
    ys, xs = np.mgrid[0:a.height, 0:a.width]
    dx = 2 * ((xs + rng.random(xs.shape)) / a.width) - 1
    dy = 2 * ((ys + rng.random(ys.shape)) / a.height) - 1
    d = dx[..., None] * U + dy[..., None] * V + W
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).reshape(-1, 3).astype(np.float32)
    n0 = d.shape[0]
    rays = np.zeros((n0, 8), np.float32); rays[:, 0:3] = eye; rays[:, 3:6] = d; rays[:, 6] = 0.01; rays[:, 7] = 1e16
    v = obj.getVerticesFloat().reshape(-1, 4)[:, :3]; idx = obj.getIndexBuffer().reshape(-1, 3)
    e1 = v[idx[:, 1]] - v[idx[:, 0]]; e2 = v[idx[:, 2]] - v[idx[:, 0]]
    nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    all_rays = [rays]
    cur = rays
    for b in range(a.bounces):
        t, prim = trace(L, state.context, cur)
        hit = prim != 0xFFFFFFFF
        P = cur[hit, 0:3] + t[hit, None] * cur[hit, 3:6]
        N = nrm[prim[hit]]
        N = np.where((np.sum(N * cur[hit, 3:6], axis=1) > 0)[:, None], -N, N).astype(np.float32)
        m = P.shape[0]
        # shadow rays to the area light (corner (343,547,227), v1 (0,0,105), v2 (-130,0,0))
        lp = np.array([343, 547, 227], np.float32) + rng.random((m, 1)).astype(np.float32) * np.array([0, 0, 105], np.float32) \
            + rng.random((m, 1)).astype(np.float32) * np.array([-130, 0, 0], np.float32)
        Ld = lp - P; dist = np.linalg.norm(Ld, axis=1); Ld /= dist[:, None]
        ok = (np.sum(N * Ld, axis=1) > 0) & (Ld[:, 1] > 0)
        sh = np.zeros((int(ok.sum()), 8), np.float32)
        sh[:, 0:3] = P[ok]; sh[:, 3:6] = Ld[ok]; sh[:, 6] = 0.01; sh[:, 7] = -(dist[ok] - 0.01)      # negative: any-hit
        all_rays.append(sh)
        # cosine-weighted bounce (roulette-like thinning: keep ~70 %)
        z1, z2 = rng.random(m), rng.random(m)
        th = np.arccos(np.sqrt(z1)); ph = 2 * np.pi * z2
        loc = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], axis=1)
        bn = np.where((np.abs(N[:, 0]) > np.abs(N[:, 2]))[:, None], np.stack([-N[:, 1], N[:, 0], 0 * N[:, 0]], 1), np.stack([0 * N[:, 0], -N[:, 2], N[:, 1]], 1))
        bn /= np.linalg.norm(bn, axis=1, keepdims=True)
        tg = np.cross(bn, N)
        nd = loc[:, 0:1] * tg + loc[:, 1:2] * bn + loc[:, 2:3] * N
        keep = rng.random(m) < 0.7
        nxt = np.zeros((int(keep.sum()), 8), np.float32)
        nxt[:, 0:3] = P[keep]; nxt[:, 3:6] = nd[keep]; nxt[:, 6] = 0.01; nxt[:, 7] = 1e16
        all_rays.append(nxt)
        cur = nxt
    R = np.ascontiguousarray(np.concatenate(all_rays, axis=0), np.float32)
    # interleave kinds the way lanes see them (shuffle within blocks of 4096 keeps locality of neighbours)
    n = R.shape[0]
    closest = R[:, 7] > 0
    tc, pc = trace(L, state.context, R[closest])
    anyr = np.ascontiguousarray(R[~closest]); anyr[:, 7] = -anyr[:, 7]
    h = np.zeros(anyr.shape[0], np.uint8)
    assert L.pt_trace_any(state.context, anyr.ctypes.data, anyr.shape[0], h.ctypes.data) == 0
    assert L.pt_bench_traversal(state.context, R.ctypes.data, 64, 1, 1, np.zeros(64, np.float32).ctypes.data, np.zeros(64, np.uint32).ctypes.data, C.byref(C.c_float()), None) == 0   # builds the four-wide tree
    info = pt.getBvhInfo(state)
    print("scene %s (%d triangles): %d rays (%d camera, %d shadow, %d bounce); closest-hit rate %.2f, occluded %.2f"
          % (a.scene, info.n_tris, n, n0, int((~closest).sum()), n - n0 - int((~closest).sum()), float((pc != 0xFFFFFFFF).mean()), float(h.mean())))
    print("two-child tree: %d nodes, depth %d, build %.2f ms; four-wide tree: %d nodes, depth %d, collapse %.2f ms (host)"
          % (info.n_nodes, info.max_depth, info.build_ms, info.wide_nodes, info.wide_depth, info.wide_ms))
    for fmt, name in ((0, "two-child fp32 nodes (64 B)"), (1, "four-wide 8-bit nodes (48 B)"), (2, "two-child, fma slab test"),
                      (3, "two-child fp16 {lo, hi} nodes (32 B)"), (4, "fp16 {centre, half}, scale per axis")):
        t_out = np.zeros(n, np.float32); p_out = np.zeros(n, np.uint32); ms = C.c_float()
        cnt = np.zeros(5, np.uint64)
        assert L.pt_bench_traversal(state.context, R.ctypes.data, n, a.repeats, fmt, t_out.ctypes.data, p_out.ctypes.data, C.byref(ms), cnt.ctypes.data) == 0, L.pt_last_error(state.context)
        bad_c = int((p_out[closest] != pc).sum()) + int((t_out[closest].view(np.uint32) != tc.view(np.uint32)).sum())
        bad_a = int(((p_out[~closest] != 0) != (h != 0)).sum())
        print("pure traversal, ray-stream kernel, %-30s: %8.3f ms -> %8.1f Mray/s   %s"
              % (name, ms.value, n / ms.value / 1e3, "bit-identical to pt_trace_closest / pt_trace_any" if bad_c + bad_a == 0 else "MISMATCH closest %d any %d" % (bad_c, bad_a)))
        it, vis, tri, vr, lr = [float(x) for x in cnt]
        print("      per ray: %.2f node visits, %.2f triangle tests; wave iterations %.3g (%.1f%% with a visit at %.1f lanes, %.1f%% with a triangle round at %.1f lanes)"
              % (vis / n, tri / n, it, 100 * vr / it, vis / max(vr, 1), 100 * lr / it, tri / max(lr, 1)))
    pt.CleanAllTheThings(state)


if __name__ == "__main__":
    main()
