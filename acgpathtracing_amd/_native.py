"""ctypes bindings of the C ABI (include/acgpt.h) and of the host-side mirror library.

The product path is libacgpt_hip.so and nothing else: if it is missing or fails to load,
importing this module's `hip()` raises — there is no fallback.
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def __init__(self, x=0.0, y=0.0, z=0.0):
        super().__init__(float(x), float(y), float(z))

    def tuple(self):
        return (self.x, self.y, self.z)


class Material(C.Structure):
    """Material, PathTracer_Optix/TinyObjWrapper.h:33-40."""
    _fields_ = [("diffuse", Float3), ("emission", Float3), ("roughness", C.c_float),
                ("metallic", C.c_float), ("ior", C.c_float), ("bsdfType", C.c_int32)]


class AreaLight(C.Structure):
    """AreaLight, PathTracer_Optix/pathTracer.h:77-83."""
    _fields_ = [("corner", Float3), ("v1", Float3), ("v2", Float3), ("normal", Float3), ("emission", Float3)]


class PathTraceParams(C.Structure):
    """PathTraceParams, PathTracer_Optix/pathTracer.h:85-108 (168 bytes)."""
    _fields_ = [("currentFrameIdx", C.c_uint32),
                ("accumulationBuffer", C.c_void_p),
                ("frameBuffer", C.c_void_p),
                ("width", C.c_uint32), ("height", C.c_uint32),
                ("samplesPerPixel", C.c_uint32), ("maxDepth", C.c_uint32),
                ("cameraEye", Float3), ("cameraU", Float3), ("cameraV", Float3), ("cameraW", Float3),
                ("areaLight", AreaLight),
                ("handle", C.c_uint64),
                ("useDirectLighting", C.c_uint8), ("useImportanceSampling", C.c_uint8)]


class Stats(C.Structure):
    _fields_ = [("radiance_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("paths", C.c_uint64),
                ("kernel_ms", C.c_float), ("launch_ms", C.c_float), ("pixels", C.c_uint32), ("grid_blocks", C.c_uint32), ("sample_chunks", C.c_uint32), ("variant", C.c_uint32),
                ("trav_wave_steps", C.c_uint64), ("trav_lane_steps", C.c_uint64),
                ("shade_wave_rounds", C.c_uint64), ("shade_lane_rounds", C.c_uint64), ("culled_rays", C.c_uint64),
                ("math_mode", C.c_uint32), ("reserved", C.c_uint32)]


class BvhInfo(C.Structure):
    _fields_ = [("n_tris", C.c_uint32), ("n_nodes", C.c_uint32), ("max_depth", C.c_uint32), ("stack_entries", C.c_uint32),
                ("scene_lo", C.c_float * 3), ("scene_hi", C.c_float * 3), ("build_ms", C.c_float),
                ("node_bytes", C.c_uint32), ("tri_bytes", C.c_uint32),
                ("wide_nodes", C.c_uint32), ("wide_depth", C.c_uint32), ("wide_bytes", C.c_uint32), ("wide_ms", C.c_float),
                ("half_node_bytes", C.c_uint32), ("half_area_ratio", C.c_float), ("half_box_inflation", C.c_float),
                ("device_bytes", C.c_uint64)]


assert C.sizeof(PathTraceParams) == 168
assert C.sizeof(Material) == 40
assert C.sizeof(AreaLight) == 60
assert C.sizeof(Stats) == 96 and C.sizeof(BvhInfo) == 88          # ABI version 4 (include/acgpt.h)
ABI_VERSION = 4
MATH_IEEE, MATH_FAST = 0, 1                                        # pt_set_math_mode

# every symbol include/acgpt.h declares (the drop-in boundary) ...
ABI_SYMBOLS = [
    "pt_create", "pt_create_multi", "pt_device_count", "pt_destroy", "pt_last_error", "pt_set_scene", "pt_set_build_mode", "pt_scene_handle", "pt_get_bvh_info",
    "pt_launch", "pt_launch_frames", "pt_resolve_framebuffer", "pt_set_partition", "pt_set_sample_chunks", "pt_set_light_mode", "pt_set_math_mode", "pt_set_scratch_limit", "pt_set_tuning",
    "pt_variant_name", "pt_variant_kernel", "pt_kernel_source_hash", "pt_set_stream", "pt_get_stats",
    "pt_trace_closest", "pt_trace_any",
    "pt_device_malloc", "pt_device_free", "pt_device_memset", "pt_copy_to_host", "pt_copy_to_device",
    "pt_host_malloc_mapped", "pt_host_free_mapped", "pt_abi_version",
]
# ... and include/acgpt_test.h (test hooks and diagnostics; same library)
TEST_SYMBOLS = ["pt_bench_traversal", "pt_selftest", "pt_debug_wave_times", "pt_debug_queue_progress", "pt_debug_window_moves", "pt_debug_queue_order", "pt_debug_pixel_classes", "pt_debug_row_spans", "pt_read_morton"]

_hip = None
_host = None


def hip_library_path():
    # ACGPT_EXPERIMENTS=1 (tools/sweep_variants.py only): the build that also carries the kernel variants that were
    # measured and not adopted.  Same sources, same ABI; never what tests, smoke() or bench.py load by default.
    if os.environ.get("ACGPT_HIP_LIB"):          # A/B measurements of differently built libraries (tools/ only)
        return os.path.join(PKG, os.environ["ACGPT_HIP_LIB"])
    if os.environ.get("ACGPT_EXPERIMENTS") == "1":
        return os.path.join(PKG, "libacgpt_hip_exp.so")
    return os.path.join(PKG, "libacgpt_hip.so")


def host_library_path():
    return os.path.join(PKG, "libacgpt_host.so")


def _torch_first():
    """The load-order rule of INTEGRATION.md section 5, enforced instead of documented only.  PyTorch's ROCm wheel carries its own
    HIP / HSA runtime; libacgpt_hip.so links against /opt/rocm's.  With torch's runtime up first the two coexist; in a process
    that loaded this library first, a later torch.cuda.init() reports "No HIP GPUs are available".  So, before the library is
    loaded: a torch that is already imported gets its GPU side initialised; a torch that is installed but not imported yet is
    imported and initialised (ACGPT_TORCH_FIRST=1, the default), left alone (=0: a process that will never use torch and does
    not want the import), or refused with the reason (=error)."""
    import importlib.util
    import sys
    mode = os.environ.get("ACGPT_TORCH_FIRST", "1")
    if "torch" not in sys.modules:
        if mode == "0" or importlib.util.find_spec("torch") is None:
            return
        if mode == "error":
            raise RuntimeError("libacgpt_hip.so is about to be loaded before PyTorch's GPU runtime: `import torch; torch.cuda.is_available()` "
                               "must come first in a process that uses both (INTEGRATION.md section 5), or set ACGPT_TORCH_FIRST=0 "
                               "if this process never touches torch.cuda")
    import torch
    torch.cuda.is_available()


def hip():
    """The HIP library.  Raises if it is not built or cannot be loaded (no fallback)."""
    global _hip
    if _hip is not None:
        return _hip
    path = hip_library_path()
    if not os.path.exists(path):
        raise RuntimeError("libacgpt_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
    _torch_first()
    L = C.CDLL(path)
    vp, sz, u32p, f32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.pt_create.argtypes = [C.POINTER(vp), C.c_int]; L.pt_create.restype = C.c_int
    L.pt_create_multi.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]; L.pt_create_multi.restype = C.c_int
    L.pt_device_count.argtypes = [vp]; L.pt_device_count.restype = C.c_int
    L.pt_variant_kernel.argtypes = [C.c_int, C.c_int]; L.pt_variant_kernel.restype = C.c_char_p
    L.pt_kernel_source_hash.argtypes = []; L.pt_kernel_source_hash.restype = C.c_char_p
    L.pt_destroy.argtypes = [vp]; L.pt_destroy.restype = None
    L.pt_last_error.argtypes = [vp]; L.pt_last_error.restype = C.c_char_p
    L.pt_set_scene.argtypes = [vp, vp, sz, vp, sz, vp, vp, sz]; L.pt_set_scene.restype = C.c_int
    L.pt_set_build_mode.argtypes = [vp, C.c_int]; L.pt_set_build_mode.restype = C.c_int
    L.pt_scene_handle.argtypes = [vp]; L.pt_scene_handle.restype = C.c_uint64
    L.pt_get_bvh_info.argtypes = [vp, C.POINTER(BvhInfo)]; L.pt_get_bvh_info.restype = C.c_int
    L.pt_launch.argtypes = [vp, C.POINTER(PathTraceParams)]; L.pt_launch.restype = C.c_int
    L.pt_launch_frames.argtypes = [vp, C.POINTER(PathTraceParams), C.c_uint32]; L.pt_launch_frames.restype = C.c_int
    L.pt_resolve_framebuffer.argtypes = [vp, vp, vp, sz]; L.pt_resolve_framebuffer.restype = C.c_int
    L.pt_set_partition.argtypes = [vp, C.c_int, C.c_int]; L.pt_set_partition.restype = C.c_int
    L.pt_set_sample_chunks.argtypes = [vp, C.c_int]; L.pt_set_sample_chunks.restype = C.c_int
    L.pt_set_light_mode.argtypes = [vp, C.c_int]; L.pt_set_light_mode.restype = C.c_int
    L.pt_set_math_mode.argtypes = [vp, C.c_int]; L.pt_set_math_mode.restype = C.c_int
    L.pt_set_scratch_limit.argtypes = [vp, C.c_size_t]; L.pt_set_scratch_limit.restype = C.c_int
    L.pt_set_tuning.argtypes = [vp, C.c_int, C.c_int]; L.pt_set_tuning.restype = C.c_int
    L.pt_variant_name.argtypes = [C.c_int]; L.pt_variant_name.restype = C.c_char_p
    L.pt_set_stream.argtypes = [vp, vp]; L.pt_set_stream.restype = C.c_int
    L.pt_get_stats.argtypes = [vp, C.POINTER(Stats)]; L.pt_get_stats.restype = C.c_int
    L.pt_trace_closest.argtypes = [vp, vp, sz, vp, vp]; L.pt_trace_closest.restype = C.c_int
    L.pt_trace_any.argtypes = [vp, vp, sz, vp]; L.pt_trace_any.restype = C.c_int
    L.pt_bench_traversal.argtypes = [vp, vp, sz, C.c_int, C.c_int, vp, vp, C.POINTER(C.c_float), vp]; L.pt_bench_traversal.restype = C.c_int
    L.pt_selftest.argtypes = [vp, C.c_int, vp, sz, vp]; L.pt_selftest.restype = C.c_int
    L.pt_debug_wave_times.argtypes = [vp, vp, sz]; L.pt_debug_wave_times.restype = C.c_int
    L.pt_debug_queue_progress.argtypes = [vp, vp]; L.pt_debug_queue_progress.restype = C.c_int
    L.pt_read_morton.argtypes = [vp, vp, vp]; L.pt_read_morton.restype = C.c_int
    L.pt_debug_window_moves.argtypes = [vp, vp]; L.pt_debug_window_moves.restype = C.c_int
    if hasattr(L, "pt_debug_node_order"):      # experiments library only
        L.pt_debug_node_order.argtypes = [vp, C.c_int]; L.pt_debug_node_order.restype = C.c_int
    if hasattr(L, "pt_debug_wf"):      # experiments library only
        L.pt_debug_wf.argtypes = [vp, vp]; L.pt_debug_wf.restype = C.c_int
    L.pt_debug_queue_order.argtypes = [vp, C.c_int]; L.pt_debug_queue_order.restype = C.c_int
    L.pt_debug_pixel_classes.argtypes = [vp, C.c_int]; L.pt_debug_pixel_classes.restype = C.c_int
    L.pt_debug_row_spans.argtypes = [C.POINTER(PathTraceParams), vp, vp, vp]; L.pt_debug_row_spans.restype = C.c_int
    L.pt_device_malloc.argtypes = [vp, C.POINTER(vp), sz]; L.pt_device_malloc.restype = C.c_int
    L.pt_device_free.argtypes = [vp, vp]; L.pt_device_free.restype = C.c_int
    L.pt_device_memset.argtypes = [vp, vp, C.c_int, sz]; L.pt_device_memset.restype = C.c_int
    L.pt_copy_to_host.argtypes = [vp, vp, vp, sz]; L.pt_copy_to_host.restype = C.c_int
    L.pt_copy_to_device.argtypes = [vp, vp, vp, sz]; L.pt_copy_to_device.restype = C.c_int
    L.pt_host_malloc_mapped.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), sz]; L.pt_host_malloc_mapped.restype = C.c_int
    L.pt_host_free_mapped.argtypes = [vp, vp]; L.pt_host_free_mapped.restype = C.c_int
    L.pt_abi_version.argtypes = []; L.pt_abi_version.restype = C.c_uint32
    if L.pt_abi_version() != ABI_VERSION:
        raise RuntimeError("libacgpt_hip.so has ABI version %d, this binding expects %d: rebuild (__graft_entry__.build())" % (L.pt_abi_version(), ABI_VERSION))
    _hip = L
    return L


def host():
    """The host-side mirror library (OBJ ingest, Camera, Trackball): plain C++ on the CPU."""
    global _host
    if _host is not None:
        return _host
    path = host_library_path()
    if not os.path.exists(path):
        raise RuntimeError("libacgpt_host.so is not built (run __graft_entry__.build())")
    L = C.CDLL(path)
    vp, sz = C.c_void_p, C.c_size_t
    L.pth_obj_load.argtypes = [C.c_char_p]; L.pth_obj_load.restype = vp
    L.pth_obj_ok.argtypes = [vp]; L.pth_obj_ok.restype = C.c_int
    L.pth_obj_error.argtypes = [vp]; L.pth_obj_error.restype = C.c_char_p
    L.pth_obj_warning.argtypes = [vp]; L.pth_obj_warning.restype = C.c_char_p
    L.pth_obj_sizes.argtypes = [vp] + [C.POINTER(sz)] * 4; L.pth_obj_sizes.restype = None
    L.pth_obj_fill.argtypes = [vp, vp, vp, vp, vp]; L.pth_obj_fill.restype = None
    L.pth_obj_free.argtypes = [vp]; L.pth_obj_free.restype = None
    L.pth_camera_uvw.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]; L.pth_camera_uvw.restype = None
    L.pth_trackball_script.argtypes = [vp, vp, vp, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, vp, sz, vp]
    L.pth_trackball_script.restype = None
    L.pth_save_image.argtypes = [C.c_char_p, vp, C.c_int, C.c_int]; L.pth_save_image.restype = C.c_int
    _host = L
    return L
