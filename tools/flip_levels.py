#!/usr/bin/env python3
"""What the FAST math mode differs from the oracle by in UNIFORM-hemisphere mode, per test configuration and over several
frame seeds: whole-image MSE, the count of pixels that carry a flipped path (squared error > 1e-6), the MSE of the rest and the
image means (tests/scene_utils.flip_report).  The bars of tests/test_gpu_parity.py for that mode are set from this table
(profiles/r04_flip_levels.txt).  GPU box; usage: tools/flip_levels.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import acgpathtracing_amd as pt
from acgpathtracing_amd import _native
import oracle_lib
from scene_utils import copy_params, flip_report, make_params

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = _native.hip()
orc = oracle_lib.load()
CASES = [("cornell_box_diffuse.obj", 256, 256, 16, 3, False, False, "test_render_config1_diffuse"),
         ("cornell_box.obj", 128, 96, 8, 16, True, False, "test_render_all_bsdfs[DL 1 IS 0 depth 16]"),
         ("cornell_box.obj", 128, 96, 8, 28, False, False, "test_render_all_bsdfs[DL 0 IS 0 depth 28]"),
         ("cornell_box_diffuse.obj", 64, 64, 16, 4, False, False, "test_fast_math_flips 16 spp"),
         ("cornell_box_diffuse.obj", 64, 64, 256, 4, False, False, "test_fast_math_flips 256 spp"),
         ("cornell_box.obj", 512, 512, 128, 4, False, False, "config 0 (the reference's start-up workload)")]
print("%-46s %5s %9s %6s %9s %9s %9s %9s" % ("case", "frame", "mse", "n_out", "frac_out", "mse_rest", "d_mean/m", "d_rest/m"))
for scene, w, h, spp, depth, dl, isamp, name in CASES:
    state, obj = pt.setup(os.path.join(pt.SCENES, scene), width=w, height=h, max_depth=depth, direct_lighting=dl, importance_sampling=isamp, spp=spp, math_mode="fast")
    assert L.pt_set_sample_chunks(state.context, 1) == 0
    sc = orc.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    worst = None
    for f in range(frames if w * h * spp < 2e7 else 1):
        p = make_params(w, h, spp, depth, dl, isamp, frame=f)
        keep_a, keep_h = state.params.accumulationBuffer, state.params.handle
        C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
        state.params.accumulationBuffer, state.params.handle = keep_a, keep_h
        # frame f alone, into a zeroed buffer: both sides then hold lerp(0, frame, 1 / (f + 1)) = frame / (f + 1); scaled back
        L.pt_device_memset(state.context, state.params.accumulationBuffer, 0, w * h * 16)
        state.params.currentFrameIdx = f
        assert L.pt_launch_frames(state.context, C.byref(state.params), 1) == 0
        acc = pt.readAccumulation(state) * np.float32(f + 1)
        q = copy_params(p); q.currentFrameIdx = f
        ref, _, _, _ = sc.render(q, use_bvh=True)
        ref = ref * np.float32(f + 1)
        r = flip_report(acc, ref)
        print("%-46s %5d %9.3e %6d %9.2e %9.3e %9.2e %9.2e" % (name, f, r["mse"], r["n_out"], r["frac_out"], r["mse_rest"],
              abs(r["mean_a"] - r["mean_b"]) / max(1e-30, r["mean_b"]), abs(r["rest_mean_a"] - r["rest_mean_b"]) / max(1e-30, r["rest_mean_b"])), flush=True)
    sc.close()
    pt.CleanAllTheThings(state)
