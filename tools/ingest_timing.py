#!/usr/bin/env python3
"""Scene set-up timing for a large OBJ: host ingest (TinyObjWrapper phases, by thread count) and the on-device
BVH build.  usage: python tools/ingest_timing.py [--scene stress_1m.obj] [--no-gpu]"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acgpathtracing_amd as pt  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="stress_1m.obj")
    ap.add_argument("--no-gpu", action="store_true")
    ap.add_argument("--threads", default="1,2,4,8,16", help="thread counts of the ingest sweep ('' skips it)")
    a = ap.parse_args()
    t = time.time()
    path = a.scene if os.path.isabs(a.scene) else bench.scene_path(pt, a.scene)
    print("scene %s: %.1f MB (generated / found in %.2f s), host threads available %d" % (a.scene, os.path.getsize(path) / 1e6, time.time() - t, os.cpu_count()))
    os.environ["ACGPT_OBJ_TIMING"] = "1"
    for thr in [int(x) for x in a.threads.split(",") if x]:
        os.environ["ACGPT_OBJ_THREADS"] = str(thr)
        best = 1e9
        for _ in range(3):
            t = time.time(); obj = pt.TinyObjWrapper(path); best = min(best, time.time() - t)
        print("ingest, %2d threads: %.3f s (best of 3, includes the copy into numpy arrays), %d triangles" % (thr, best, obj.getIndexBuffer().size // 3))
        sys.stdout.flush()
    os.environ.pop("ACGPT_OBJ_THREADS", None)
    if a.no_gpu:
        return
    for mode, name in ((2, "PLOC + insertion-based optimisation (the default: on the host up to 16 384 triangles, parallel reinsertion on the device above)"), (1, "PLOC"), (0, "Karras LBVH")):
        t = time.time()
        state, obj = pt.setup(path, width=64, height=64, build_mode=mode)
        wall = time.time() - t
        info = pt.getBvhInfo(state)
        print("set-up with %s: %.3f s wall (ingest + upload + build); device build %.2f ms, %d nodes, depth %d" % (name, wall, info.build_ms, info.n_nodes, info.max_depth))
        pt.CleanAllTheThings(state)


if __name__ == "__main__":
    main()
