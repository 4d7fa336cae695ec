#!/usr/bin/env python3
"""Random cameras, toggles and depths: GPU image against the oracle (MSE < 1e-3 and the fraction of bit-identical
pixels), both the default sample-run setting and the reference's single chain.  usage: python tools/soak_images.py [--cases 30]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import acgpathtracing_amd as pt  # noqa: E402
from acgpathtracing_amd import _native  # noqa: E402
import oracle_lib  # noqa: E402
from scene_utils import copy_params, image_mse, image_mse_trimmed, make_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--scene", default="cornell_box.obj")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant (pt_set_tuning; -1 = the library's choice)")
    ap.add_argument("--math", default="fast", choices=["fast", "ieee"], help="pt_set_math_mode: fast (the library default) or ieee (the oracle's arithmetic level)")
    a = ap.parse_args()
    L = _native.hip()
    orc = oracle_lib.load()
    state, obj = pt.setup(os.path.join(pt.SCENES, a.scene), width=96, height=64)
    sc = orc.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    assert L.pt_set_tuning(state.context, 0, a.variant) == 0
    pt.setMathMode(state, a.math)
    rng = np.random.default_rng(2024)
    worst = 0.0
    hist = {"is_on": [], "is_off": [], "is_off_trimmed": []}
    for k in range(a.cases):
        w, h = int(rng.choice([64, 96, 130])), int(rng.choice([48, 64, 75]))
        spp = int(rng.choice([4, 8, 16])); depth = int(rng.integers(1, 17))
        dl, isamp = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        p = make_params(w, h, spp, depth, dl, isamp)
        cam = pt.Camera()
        inside = bool(rng.integers(0, 2))
        eye = rng.uniform((30, 30, 30), (520, 520, 520)) if inside else rng.uniform((-300, 0, -1200), (800, 600, -300))
        look = rng.uniform((100, 100, 100), (450, 450, 450))
        cam.setEye(tuple(float(x) for x in eye)); cam.setLookat(tuple(float(x) for x in look)); cam.setUp((0.0, 1.0, 0.0))
        cam.setFovY(float(rng.uniform(20, 80))); cam.setAspectRatio(w / h)
        U, V, W = cam.UVWFrame()
        for dst, src in ((p.cameraEye, cam.eye()), (p.cameraU, U), (p.cameraV, V), (p.cameraW, W)):
            dst.x, dst.y, dst.z = float(src[0]), float(src[1]), float(src[2])
        frames = int(rng.integers(1, 4))
        for chunks in (0, 1):
            assert L.pt_set_sample_chunks(state.context, chunks) == 0
            state.params.width, state.params.height = w, h
            keep_a, keep_h = state.params.accumulationBuffer, state.params.handle
            C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
            state.params.accumulationBuffer, state.params.handle = keep_a, keep_h
            state.refreshAccumulationBuffer = True
            pt.updateState(None, state)
            state.params.currentFrameIdx = 0
            pt.LaunchCurrentFrame(None, state, frames)
            acc = pt.readAccumulation(state)
            st = pt.getStats(state)
            ref = None
            for f in range(frames):
                q = copy_params(p); q.currentFrameIdx = f
                ref, _, rst, _ = sc.render(q, accumulation=ref, use_bvh=True, chunks=int(st.sample_chunks))
            mse = image_mse(acc, ref)
            differ = ~np.all(acc.view(np.uint32) == ref.view(np.uint32), axis=-1)
            same = 1.0 - float(differ.mean())
            worst = max(worst, mse)
            hist["is_on" if isamp else "is_off"].append(mse)
            if not isamp:
                hist["is_off_trimmed"].append(image_mse_trimmed(acc, ref, 1e-3))
            flag = "" if mse < 1e-3 and np.isfinite(acc).all() else "   <-- FAIL"
            print("case %2d %3dx%-3d spp %2d depth %2d DL %d IS %d %s frames %d runs %2d: MSE %.2e, %.1f %% pixels bit-identical%s"
                  % (k, w, h, spp, depth, dl, isamp, "inside " if inside else "outside", frames, st.sample_chunks, mse, 100 * same, flag))
            if mse > 1e-6:      # which pixels carry it: a path that took another branch somewhere shows as one or two pixels
                d2 = ((acc[..., :3].astype(np.float64) - ref[..., :3]) ** 2).sum(axis=-1)
                ys, xs = np.unravel_index(np.argsort(d2, axis=None)[::-1][:3], d2.shape)
                print("        %d pixels differ; largest: %s" % (int(differ.sum()), "; ".join("(%d,%d) gpu %s cpu %s" % (x, y, acc[y, x, :3], ref[y, x, :3]) for y, x in zip(ys, xs) if d2[y, x] > 0)))
            sys.stdout.flush()
    print("worst MSE %.3e (math mode %s)" % (worst, a.math))
    for k, v in hist.items():
        if v:
            v = np.sort(np.asarray(v))
            print("  %-15s n %3d: median %.2e, 90 %% %.2e, max %.2e, below 1e-6: %d" % (k, v.size, v[v.size // 2], v[int(0.9 * (v.size - 1))], v[-1], int((v < 1e-6).sum())))
    sc.close()
    pt.CleanAllTheThings(state)
    return 0 if worst < 1e-3 else 1


if __name__ == "__main__":
    sys.exit(main())
