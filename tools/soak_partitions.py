#!/usr/bin/env python3
"""Random image sizes x GPU counts x sample-run settings x frame batches: the ranks' tiles (pt_set_partition, each rank into its
own zero-filled buffer) must add up to the one-rank image bit for bit, and the ranks' ray / path / pixel counters to the
one-rank counters.  Everything on one GPU.  usage: python tools/soak_partitions.py [--cases 60]"""
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


def render(L, state, p, frames, rank, world):
    assert L.pt_set_partition(state.context, rank, world) == 0
    n = p.width * p.height * 16
    buf = C.c_void_p()
    assert L.pt_device_malloc(state.context, C.byref(buf), n) == 0
    assert L.pt_device_memset(state.context, buf, 0, n) == 0
    q = type(p)(); C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
    q.accumulationBuffer = buf.value; q.frameBuffer = None; q.handle = state.params.handle; q.currentFrameIdx = 0
    rc = L.pt_launch_frames(state.context, C.byref(q), frames)
    assert rc == 0, L.pt_last_error(state.context)
    acc = np.zeros((p.height, p.width, 4), np.float32)
    assert L.pt_copy_to_host(state.context, acc.ctypes.data, buf, n) == 0
    L.pt_device_free(state.context, buf)
    st = pt.getStats(state)
    return acc, np.array([int(st.radiance_rays), int(st.shadow_rays), int(st.paths), int(st.pixels), int(st.culled_rays)], np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    a = ap.parse_args()
    L = _native.hip()
    rng = np.random.default_rng(777)
    state, obj = pt.setup(os.path.join(pt.SCENES, "cornell_box.obj"), width=64, height=64)
    bad = 0
    for k in range(a.cases):
        w = int(rng.choice([1, 3, 7, 8, 9, 31, 64, 100, 257, int(rng.integers(1, 400))]))
        h = int(rng.choice([1, 2, 4, 5, 33, 64, 90, int(rng.integers(1, 300))]))
        world = int(rng.choice([2, 3, 4, 5, 7, 8]))
        chunks = int(rng.choice([0, 1, 4]))
        frames = int(rng.integers(1, 4))
        spp = int(rng.choice([4, 8, 16]))
        depth = int(rng.integers(1, 7))
        assert L.pt_set_sample_chunks(state.context, chunks) == 0
        p = make_params(w, h, spp, depth, True, True)
        # the whole image with the sample-run count the ranks will choose (the automatic choice depends on the pixels per rank)
        parts = [render(L, state, p, frames, r, world) for r in range(world)]
        runs = int(pt.getStats(state).sample_chunks)
        assert L.pt_set_sample_chunks(state.context, runs) == 0
        whole, cw = render(L, state, p, frames, 0, 1)
        total = np.zeros_like(whole); ct = np.zeros(5, np.int64)
        for acc, c in parts:
            total += acc; ct += c
        ok = np.array_equal(total.view(np.uint32), whole.view(np.uint32)) and np.array_equal(ct[:4], cw[:4])
        if not ok:
            bad += 1
            print("MISMATCH case %d: %dx%d world %d runs %d frames %d spp %d: %d pixels differ, counters %s vs %s"
                  % (k, w, h, world, runs, frames, spp, int(np.any(total != whole, axis=-1).sum()), ct, cw), flush=True)
    L.pt_set_partition(state.context, 0, 1)
    pt.CleanAllTheThings(state)
    print("%d cases, %d mismatches" % (a.cases, bad))
    print("SOAK_PARTITIONS", "OK" if bad == 0 else "FAILED")
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
