#!/usr/bin/env python3
"""Many seeds of the adversarial + random ray sets through every traversal (query kernels, ray-stream kernel on both tree
formats, both hierarchy builders) against the oracle's brute-force loop.  Found the equal-distance pruning case fixed by
kTieWiden; run it after touching a box test.  usage: python tools/soak_rays.py [--seeds 40] [--first 100]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import acgpathtracing_amd as pt  # noqa: E402
from acgpathtracing_amd import _native  # noqa: E402
import oracle_lib  # noqa: E402
from scene_utils import adversarial_rays, random_rays, scene_arrays  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--first", type=int, default=100)
    ap.add_argument("--scene", default="cornell_box.obj")
    a = ap.parse_args()
    L = _native.hip()
    orc = oracle_lib.load()
    bad = 0
    for mode in (2, 1, 0):          # the default (PLOC + insertion-based optimisation), PLOC, Karras
        state, obj = pt.setup(os.path.join(pt.SCENES, a.scene), width=64, height=64, build_mode=mode)
        sc = orc.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
        v, idx = scene_arrays(obj)
        t0 = time.time()
        for s in range(a.first, a.first + a.seeds):
            rays = np.concatenate([random_rays(60000, 3 * s), adversarial_rays(v, idx, 3 * s + 1), random_rays(20000, 3 * s + 2, tmin=0.01, tmax=200.0)]).astype(np.float32)
            rays = np.ascontiguousarray(rays)
            n = rays.shape[0]
            t_ref, p_ref = sc.trace_closest(rays, use_bvh=False)
            a_ref = sc.trace_any(rays, use_bvh=False) != 0
            t = np.zeros(n, np.float32); p = np.zeros(n, np.uint32); h = np.zeros(n, np.uint8); ms = C.c_float()
            assert L.pt_trace_closest(state.context, rays.ctypes.data, n, t.ctypes.data, p.ctypes.data) == 0
            assert L.pt_trace_any(state.context, rays.ctypes.data, n, h.ctypes.data) == 0
            res = {"query": (t.copy(), p.copy())}
            for fmt in (0, 1, 2, 3, 4):
                assert L.pt_bench_traversal(state.context, rays.ctypes.data, n, 1, fmt, t.ctypes.data, p.ctypes.data, C.byref(ms), None) == 0
                res["stream fmt %d" % fmt] = (t.copy(), p.copy())
            t_b, p_b = sc.trace_closest(rays, use_bvh=True)
            res["oracle bvh"] = (t_b, p_b)
            for name, (tt, pp) in res.items():
                m = (pp != p_ref) | (tt.view(np.uint32) != t_ref.view(np.uint32))
                if m.any():
                    bad += int(m.sum())
                    i = int(np.nonzero(m)[0][0])
                    print("MISMATCH mode %d seed %d %s: %d rays, first %d: got (%r, %d) want (%r, %d) ray %s" % (mode, s, name, int(m.sum()), i, tt[i], pp[i], t_ref[i], p_ref[i], rays[i]))
            m = (h != 0) != a_ref
            if m.any():
                bad += int(m.sum()); print("MISMATCH any-hit mode %d seed %d: %d rays" % (mode, s, int(m.sum())))
            ar = rays.copy(); ar[:, 7] *= -1.0
            for fmt in (0, 1, 2, 3, 4):
                assert L.pt_bench_traversal(state.context, ar.ctypes.data, n, 1, fmt, t.ctypes.data, p.ctypes.data, C.byref(ms), None) == 0
                m = (p != 0) != a_ref
                if m.any():
                    bad += int(m.sum()); print("MISMATCH any-hit stream fmt %d mode %d seed %d: %d rays" % (fmt, mode, s, int(m.sum())))
            if (s - a.first) % 10 == 9:
                print("mode %d: %d seeds done, %.0f s, %d mismatching rays so far" % (mode, s - a.first + 1, time.time() - t0, bad)); sys.stdout.flush()
        sc.close()
        pt.CleanAllTheThings(state)
    print("TOTAL mismatching rays: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
