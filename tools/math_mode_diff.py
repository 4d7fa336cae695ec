#!/usr/bin/env python3
"""PT_MATH_IEEE against PT_MATH_FAST (pt_set_math_mode) on the diffuse Cornell box, 256 x 256, 16 spp: per toggle pair and depth the
MSE between the two images, the fraction of pixels with other bits, the pixels that differ by more than 1e-3 of their value, and
the ray counters.  Shows where the two arithmetic levels trace different paths: in uniform-hemisphere mode one bounce in ~10^5
skims its own wall and is decided by the last bits of its direction (tests/scene_utils.image_mse_trimmed)."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acgpathtracing_amd as pt
from acgpathtracing_amd import _native
from scene_utils import make_params, image_mse
L = _native.hip()
state, obj = pt.setup(os.path.join(pt.SCENES, "cornell_box_diffuse.obj"), width=64, height=64)
assert L.pt_set_sample_chunks(state.context, 1) == 0
def render(p, mode):
    pt.setMathMode(state, mode)
    keep_a, keep_h = state.params.accumulationBuffer, state.params.handle
    C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
    state.params.accumulationBuffer, state.params.handle = keep_a, keep_h
    state.refreshAccumulationBuffer = True
    pt.updateState(None, state)
    ob = pt.OutputBuffer(pt.OutputBufferType.DEVICE, p.width, p.height, state)
    state.params.currentFrameIdx = 0
    pt.LaunchCurrentFrame(ob, state)
    acc = pt.readAccumulation(state); st = pt.getStats(state); ob.free()
    return acc, st
for isamp in (False, True):
    for dl in (False, True):
        for depth in (1, 2, 3, 8):
            p = make_params(256, 256, 16, depth, dl, isamp)
            a, sa = render(p, "ieee"); b, sb = render(p, "fast")
            diff = np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1)
            big = np.abs(a[..., :3] - b[..., :3]).max(axis=-1) > 1e-3 * np.maximum(1e-3, np.abs(a[..., :3]).max(axis=-1))
            if not isamp and not dl and depth in (1, 8):      # who carries the difference: the pixels that differ most, both values
                d2 = ((a[..., :3].astype(np.float64) - b[..., :3]) ** 2).sum(axis=-1)
                ys, xs = np.unravel_index(np.argsort(d2, axis=None)[::-1][:6], d2.shape)
                print("    largest differences at depth %d: %s" % (depth, "; ".join("(%d,%d) ieee %s fast %s" % (x, y, np.round(a[y, x, :3], 3), np.round(b[y, x, :3], 3)) for y, x in zip(ys, xs))))
                print("    pixels that see the emitter directly (red > 10): %d of %d; of the %d pixels differing by > 1e-3, on the emitter: %d"
                      % (int((a[..., 0] > 10).sum()), a.shape[0] * a.shape[1], int(big.sum()), int((big & (a[..., 0] > 10)).sum())))
            print("IS %d DL %d depth %d: MSE %.3e, pixels with other bits %.4f, pixels differing by > 1e-3 rel: %d; rays %d vs %d, shadow %d vs %d"
                  % (isamp, dl, depth, image_mse(a, b), diff.mean(), int(big.sum()), sa.radiance_rays, sb.radiance_rays, sa.shadow_rays, sb.shadow_rays))
