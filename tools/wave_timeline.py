#!/usr/bin/env python3
"""Where does a launch's time go outside the steady state?  Runs the stats variant of the render kernel and reads
three 100 MHz stamps per wave (start, queue found empty, end): ramp-up = spread of the starts, drain = what waves
do after the queue is empty.  usage: python tools/wave_timeline.py [--partition 0,8] [--fuse 8] [--chunks 0]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell_box_diffuse.obj")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--fuse", type=int, default=8)
    ap.add_argument("--chunks", type=int, default=0)
    ap.add_argument("--partition", default="0,1")
    ap.add_argument("--variant", type=int, default=6)
    ap.add_argument("--direct-lighting", type=int, default=1)
    ap.add_argument("--importance-sampling", type=int, default=1)
    a = ap.parse_args()
    L = _native.hip()
    state, obj = pt.setup(os.path.join(pt.SCENES, a.scene), width=a.width, height=a.height, max_depth=a.max_depth, direct_lighting=bool(a.direct_lighting), importance_sampling=bool(a.importance_sampling), spp=a.spp)
    p = make_params(a.width, a.height, a.spp, a.max_depth, bool(a.direct_lighting), bool(a.importance_sampling))
    keep_a, keep_h = state.params.accumulationBuffer, state.params.handle
    C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
    state.params.accumulationBuffer, state.params.handle = keep_a, keep_h
    rank, world = [int(x) for x in a.partition.split(",")]
    assert L.pt_set_partition(state.context, rank, world) == 0
    assert L.pt_set_sample_chunks(state.context, a.chunks) == 0
    assert L.pt_set_tuning(state.context, 0, a.variant) == 0
    for rep in range(2):
        state.params.currentFrameIdx = 0
        pt.LaunchCurrentFrame(None, state, a.fuse)
    s = pt.getStats(state)
    n = int(s.grid_blocks) * 4
    t = np.zeros(3 * n, np.uint64)
    assert L.pt_debug_wave_times(state.context, t.ctypes.data, n) == 0
    t = t.reshape(n, 3).astype(np.float64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    start, drain, end = (t[:, 0] - t0) / 1e5, (t[:, 1] - t0) / 1e5, (t[:, 2] - t0) / 1e5      # ms
    q = lambda x: " ".join("%8.3f" % v for v in np.percentile(x, [0, 1, 10, 50, 90, 99, 100]))
    print("kernel %.3f ms (HIP events), %d waves stamped, partition %s, %d steps per launch, %d sample runs" % (s.kernel_ms, t.shape[0], a.partition, a.fuse, s.sample_chunks))
    print("percentile              0        1       10       50       90       99      100   (ms from the first wave's start)")
    print("wave start       %s" % q(start))
    print("queue empty seen %s" % q(drain))
    print("wave end         %s" % q(end))
    print("after the queue was empty, per wave %s" % q(end - drain))
    busy = (end - start).sum(); tail = (end - drain).sum()
    print("wave-time spent after the queue was empty: %.2f %% of all wave-time; last wave ends %.3f ms after the median wave" % (100 * tail / busy, end.max() - np.median(end)))
    pr = np.zeros(2056, np.uint64)
    assert L.pt_debug_queue_progress(state.context, pr.ctypes.data) == 0
    print("share of the wave-time spent in the shade / regenerate phase (the rest is the BVH loop): %.1f %%" % (100.0 * float(pr[2048]) / 1e5 / busy))
    print("   of the wave-time: queue refill %.1f %%, finished runs (park / fold / write) %.1f %%, camera-path start incl. cull %.1f %%" %
          tuple(100.0 * float(pr[2049 + k]) / 1e5 / busy for k in range(3)))
    print("   shade rounds %d, lanes shaded per round %.1f; BVH loop trips %d at %.1f lanes" % (s.shade_wave_rounds, s.shade_lane_rounds / max(1, s.shade_wave_rounds), s.trav_wave_steps, s.trav_lane_steps / max(1, s.trav_wave_steps)))
    pr = pr[:2048].reshape(8, 256).astype(np.float64)
    print("work-queue progress (all shards advance together unless noted): time at which each tenth of the items had been handed out, and the rate between them")
    frac = []
    for k in range(0, 256, 16):
        col = pr[:, k]; col = col[col > 0]
        if col.size:
            frac.append((k / 256.0, (np.median(col) - t0) / 1e5, (col.max() - col.min()) / 1e5))
    prev = None
    for f, ms, spread in frac:
        rate = "" if prev is None else "  -> %.2f %% of the items per ms" % (100 * (f - prev[0]) / max(ms - prev[1], 1e-9))
        print("   %5.1f %% handed out at %8.3f ms (shards spread %.3f ms)%s" % (100 * f, ms, spread, rate))
        prev = (f, ms)
    print("if every wave had ended at the median end, the launch would take %.3f ms instead of %.3f" % (np.median(end), end.max()))
    pt.CleanAllTheThings(state)


if __name__ == "__main__":
    main()
