#!/usr/bin/env python3
"""Quick A/B of the wavefront kernels against the default kernel: same bits, kernel times.  The wavefront kernels were measured and
lost; they live in the experiments library only (libacgpt_hip_exp.so: _build.build_hip(experiments=True)), which
this tool loads.  usage: tools/wf_check.py [width height spp frames depth] (GPU box)"""
import os
import sys

os.environ.setdefault("ACGPT_EXPERIMENTS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import acgpathtracing_amd as pt
from acgpathtracing_amd import _native
from scene_utils import make_params

W, H, S, F, D = [int(x) for x in (sys.argv[1:6] + ["160", "96", "8", "1", "8"][len(sys.argv) - 1:])][:5]
scene = os.environ.get("WF_SCENE", "cornell_box.obj")
variants = [int(v) for v in os.environ.get("WF_VARIANTS", "-1,10,11").split(",")]
chunk_list = [int(v) for v in os.environ.get("WF_CHUNKS", "1,0").split(",")]
L = _native.hip()
state, obj = pt.setup(os.path.join(pt.SCENES, scene), width=W, height=H, max_depth=D, direct_lighting=True, importance_sampling=True, spp=S)
p = make_params(W, H, S, D, True, True)
keep_a, keep_h = state.params.accumulationBuffer, state.params.handle
C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
state.params.accumulationBuffer, state.params.handle = keep_a, keep_h
ok = True
for chunks in chunk_list:
    assert L.pt_set_sample_chunks(state.context, chunks) == 0
    ref = None
    for v in variants:
        if L.pt_set_tuning(state.context, 0, v) != 0:
            print("variant %d refused: %s" % (v, L.pt_last_error(state.context).decode())); continue
        best = 1e9
        for rep in range(int(os.environ.get("WF_REPS", "2"))):
            L.pt_device_memset(state.context, state.params.accumulationBuffer, 0, W * H * 16)
            state.params.currentFrameIdx = 0
            rc = L.pt_launch_frames(state.context, C.byref(state.params), F)
            if rc != 0:
                print("variant %d chunks %d: LAUNCH FAILED: %s" % (v, chunks, L.pt_last_error(state.context).decode())); ok = False; break
            st = pt.getStats(state)
            best = min(best, st.kernel_ms)
        if rc != 0:
            continue
        acc = pt.readAccumulation(state)
        cnt = (int(st.radiance_rays), int(st.shadow_rays), int(st.paths), int(st.culled_rays))
        line = "chunks %2d variant %3d (ran %d, %d blocks): %9.3f ms  rays %s" % (chunks, v, st.variant, st.grid_blocks, best, cnt)
        if st.trav_wave_steps:
            line += "  lanes/trip %.1f  lanes/shade round %.1f" % (st.trav_lane_steps / st.trav_wave_steps, st.shade_lane_rounds / max(1, st.shade_wave_rounds))
        if ref is None:
            ref = (acc, cnt)
        else:
            same = np.array_equal(acc.view(np.uint32), ref[0].view(np.uint32))
            line += "  bits %s counters %s" % ("SAME" if same else "DIFFER (%d pixels, max abs %.3e)" % (int(np.any(acc != ref[0], axis=-1).sum()), float(np.abs(acc - ref[0]).max())), "same" if cnt == ref[1] else "DIFFER")
            ok = ok and same and cnt == ref[1]
        print(line, flush=True)
        if st.variant >= 10 and hasattr(L, "pt_debug_wf"):
            d = (C.c_uint64 * 17)()
            L.pt_debug_wf(state.context, d)
            d = [int(x) for x in d]
            tt, ti, stt, si = d[0], d[1], d[2], d[3]
            print("      trace waves idle %.1f %% | shade waves idle %.1f %%, deal %.1f %% (%d rounds, %.1f rec, %.2f us), hits %.1f %% (%d rounds, %.1f rec, %.2f us), accounting %.1f %% (%d rounds, %.1f rec, %.2f us)"
                  % (100.0 * ti / max(1, tt), 100.0 * si / max(1, stt),
                     100.0 * d[4] / max(1, stt), d[5], d[6] / max(1, d[5]), d[4] / max(1, d[5]) / 100.0,
                     100.0 * d[7] / max(1, stt), d[8], d[9] / max(1, d[8]), d[7] / max(1, d[8]) / 100.0,
                     100.0 * d[10] / max(1, stt), d[11], d[12] / max(1, d[11]), d[10] / max(1, d[11]) / 100.0), flush=True)
            if d[14]:
                print("      trace waves: %.1f %% of their time in exchanges (%d exchanges, %.1f records in, %.2f us each); %d trips, %.2f us per trip without the exchanges, %.2f trips per exchange"
                      % (100.0 * d[13] / max(1, tt), d[14], d[15] / d[14], d[13] / d[14] / 100.0, d[16], (tt - d[13] - ti) / max(1, d[16]) / 100.0, d[16] / d[14]), flush=True)
pt.CleanAllTheThings(state)
print("WF_CHECK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
