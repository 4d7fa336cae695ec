#!/usr/bin/env python3
"""Pixel classes against the per-ray test, random cameras: eye outside / inside / partly behind the scene box, narrow and wide
fields of view, wide and tall images, both scenes.  With the classes on (whole pixels that cannot reach the scene box settled
at grant decode, pixels that certainly reach it skip the cull test) and off, the accumulation must be the same bit for bit and
the ray / path / pixel counters equal.  usage: python tools/soak_pixel_classes.py [--cases 300]"""
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


def render(L, state, p, classes):
    assert L.pt_debug_pixel_classes(state.context, classes) == 0
    keep_h = state.params.handle
    n = p.width * p.height * 16
    buf = C.c_void_p()
    assert L.pt_device_malloc(state.context, C.byref(buf), n) == 0
    assert L.pt_device_memset(state.context, buf, 0, n) == 0
    q = type(p)(); C.memmove(C.byref(q), C.byref(p), C.sizeof(p))
    q.accumulationBuffer = buf.value; q.frameBuffer = None; q.handle = keep_h; q.currentFrameIdx = 0
    rc = L.pt_launch_frames(state.context, C.byref(q), 2)
    assert rc == 0, L.pt_last_error(state.context)
    acc = np.zeros((p.height, p.width, 4), np.float32)
    assert L.pt_copy_to_host(state.context, acc.ctypes.data, buf, n) == 0
    L.pt_device_free(state.context, buf)
    st = pt.getStats(state)
    return acc, (int(st.radiance_rays), int(st.shadow_rays), int(st.paths), int(st.pixels)), int(st.culled_rays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    a = ap.parse_args()
    L = _native.hip()
    rng = np.random.default_rng(31337)
    bad = 0
    settled = []
    for scene in ("cornell_box.obj", "cornell_box_diffuse.obj"):
        state, obj = pt.setup(os.path.join(pt.SCENES, scene), width=64, height=64)
        assert L.pt_set_sample_chunks(state.context, 0) == 0
        for k in range(a.cases // 2):
            w, h = int(rng.choice([64, 96, 160, 200, 33])), int(rng.choice([48, 64, 90, 120, 17]))
            spp = int(rng.choice([4, 8])); depth = int(rng.integers(1, 9))
            p = make_params(w, h, spp, depth, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
            kind = int(rng.integers(0, 4))
            if kind == 0:   eye = rng.uniform((-300, 0, -1500), (800, 600, -250))          # in front of the box, as the preset
            elif kind == 1: eye = rng.uniform((30, 30, 30), (520, 520, 520))              # inside the box
            elif kind == 2: eye = rng.uniform((-900, -600, -900), (1500, 1200, 1500))     # anywhere around it
            else:           eye = rng.uniform((-50, -50, -60), (610, 600, 20))            # at the box's faces: partly behind the eye
            look = rng.uniform((0, 0, 0), (556, 549, 559)) if rng.integers(0, 4) else rng.uniform((-2000, -2000, -2000), (2000, 2000, 2000))
            cam = pt.Camera()
            cam.setEye(tuple(float(x) for x in eye)); cam.setLookat(tuple(float(x) for x in look)); cam.setUp((0.0, 1.0, 0.0))
            cam.setFovY(float(rng.choice([8.0, 20.0, 35.0, 60.0, 100.0]))); cam.setAspectRatio(np.float32(w) / np.float32(h))
            U, V, W = cam.UVWFrame()
            p.cameraEye = _native.Float3(*cam.eye()); p.cameraU = _native.Float3(*U); p.cameraV = _native.Float3(*V); p.cameraW = _native.Float3(*W)
            off, c_off, culled_off = render(L, state, p, 0)
            on, c_on, culled_on = render(L, state, p, 1)
            same = np.array_equal(on.view(np.uint32), off.view(np.uint32)) and c_on == c_off
            settled.append((culled_on - culled_off) / max(1, c_on[2]))
            if not same:
                bad += 1
                print("MISMATCH scene %s case %d kind %d %dx%d eye %s look %s: %d pixels differ, counters %s vs %s"
                      % (scene, k, kind, w, h, np.round(eye, 1), np.round(look, 1), int(np.any(on != off, axis=-1).sum()), c_on, c_off), flush=True)
        pt.CleanAllTheThings(state)
    settled = np.array(settled)
    print("%d cases, %d mismatches; culled rays with the classes on minus off, as a fraction of the paths: min %.4f median %.4f max %.4f"
          % (len(settled), bad, settled.min(), np.median(settled), settled.max()))
    print("SOAK_PIXEL_CLASSES", "OK" if bad == 0 else "FAILED")
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
