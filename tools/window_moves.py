import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, acgpathtracing_amd as pt
from acgpathtracing_amd import _native
from scene_utils import make_params
L = _native.hip()
path = bench.scene_path(pt, "stress_1m.obj")
state, obj = pt.setup(path, width=1920, height=1080, max_depth=8, direct_lighting=True, importance_sampling=True, spp=128)
assert L.pt_set_sample_chunks(state.context, 0) == 0
p = make_params(1920, 1080, 128, 8, True, True)
keep_a, keep_h = state.params.accumulationBuffer, state.params.handle
C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
state.params.accumulationBuffer, state.params.handle = keep_a, keep_h
state.params.currentFrameIdx = 0
assert L.pt_launch_frames(state.context, C.byref(state.params), 2) == 0
st = pt.getStats(state)
d = (C.c_uint64 * 1)(); L.pt_debug_window_moves(state.context, d)
rays = int(st.radiance_rays + st.shadow_rays - st.culled_rays)
print("variant %d, %.1f ms, %d traversed rays, %d wave-level window moves (each moves 4 entries of up to 64 lanes: <= %.1f MB), moves per 1000 rays %.3f"
      % (st.variant, st.kernel_ms, rays, int(d[0]), int(d[0]) * 1024 / 1e6, 1000.0 * int(d[0]) / rays))
pt.CleanAllTheThings(state)
