#!/usr/bin/env python3
"""Far cameras: the default kernel (fma slab test, v_rcp) against variant 0 (subtract-multiply slab test, IEEE 1/d) and the
oracle, eye at k scene sizes from the box with a narrow field of view.  usage: python tools/far_camera_check.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import acgpathtracing_amd as pt  # noqa: E402
from acgpathtracing_amd import _native  # noqa: E402
import oracle_lib  # noqa: E402
from scene_utils import copy_params, image_mse, make_params  # noqa: E402


def main():
    L = _native.hip()
    orc = oracle_lib.load()
    state, obj = pt.setup(os.path.join(pt.SCENES, "cornell_box.obj"), width=128, height=128)
    sc = orc.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    assert L.pt_set_sample_chunks(state.context, 1) == 0
    for k in (2, 8, 30, 100, 1000):
        dist = 556.0 * k
        p = make_params(128, 128, 8, 6, True, True)
        cam = pt.Camera()
        cam.setEye((278.0 + 0.3 * dist, 273.0 + 0.2 * dist, 279.0 - dist)); cam.setLookat((278.0, 273.0, 279.0)); cam.setUp((0.0, 1.0, 0.0))
        cam.setFovY(float(np.degrees(2 * np.arctan(400.0 / dist)))); cam.setAspectRatio(1.0)
        U, V, W = cam.UVWFrame()
        for dst, src in ((p.cameraEye, cam.eye()), (p.cameraU, U), (p.cameraV, V), (p.cameraW, W)):
            dst.x, dst.y, dst.z = float(src[0]), float(src[1]), float(src[2])
        imgs = {}
        for variant in (1, 0):
            assert L.pt_set_tuning(state.context, 0, variant) == 0
            keep_a, keep_h = state.params.accumulationBuffer, state.params.handle
            C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
            state.params.accumulationBuffer, state.params.handle = keep_a, keep_h
            state.refreshAccumulationBuffer = True
            pt.updateState(None, state)
            state.params.currentFrameIdx = 0
            pt.LaunchCurrentFrame(None, state)
            imgs[variant] = pt.readAccumulation(state)
        ref, _, _, _ = sc.render(copy_params(p), use_bvh=True)
        brute, _, _, _ = sc.render(copy_params(p), use_bvh=False)
        same = np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
        print("eye at %5d scene sizes: default == variant 0 bit for bit: %s; MSE default vs oracle (BVH) %.2e, vs oracle (brute force) %.2e; oracle BVH vs brute %.2e; mean radiance %.3f"
              % (k, same, image_mse(imgs[1], ref), image_mse(imgs[1], brute), image_mse(ref, brute), float(np.clip(imgs[1][..., :3], 0, 1).mean())))
    sc.close()
    pt.CleanAllTheThings(state)


if __name__ == "__main__":
    main()
