"""ctypes loader of the CPU checker under oracle/ — TEST INFRASTRUCTURE (tests, smoke, and
the cpu_baseline leg of bench.py only)."""
import ctypes as C
import os

import numpy as np

from math_binding import MathMixin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def _cpu_has_fma():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    f = line.split()
                    return "fma" in f and "avx2" in f
    except OSError:
        pass
    return False


class Oracle(MathMixin):
    prefix = "orc_"

    def __init__(self, lib, path):
        self.lib, self.path = lib, path
        self._bind_math()
        vp, sz = C.c_void_p, C.c_size_t
        L = lib
        L.orc_tea4.argtypes = [C.c_uint32, C.c_uint32]; L.orc_tea4.restype = C.c_uint32
        L.orc_rnd_stream.argtypes = [C.c_uint32, sz, vp, vp]; L.orc_rnd_stream.restype = None
        L.orc_make_color.argtypes = [vp, sz, vp]; L.orc_make_color.restype = None
        L.orc_refract.argtypes = [vp, vp, C.c_float, vp, vp]; L.orc_refract.restype = None
        L.orc_vec_op.argtypes = [C.c_int, vp, vp, vp, C.c_float, vp]; L.orc_vec_op.restype = None
        L.orc_camera_uvw.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]; L.orc_camera_uvw.restype = None
        L.orc_num_samples.argtypes = [C.c_int] * 3; L.orc_num_samples.restype = C.c_int
        L.orc_sample_pixel.argtypes = [C.c_int] * 4 + [vp, vp]; L.orc_sample_pixel.restype = None
        L.orc_scene_create.argtypes = [vp, sz, vp, sz, vp, vp, sz]; L.orc_scene_create.restype = vp
        L.orc_scene_destroy.argtypes = [vp]; L.orc_scene_destroy.restype = None
        L.orc_trace_closest.argtypes = [vp, vp, sz, C.c_int, vp, vp]; L.orc_trace_closest.restype = None
        L.orc_trace_any.argtypes = [vp, vp, sz, C.c_int, vp]; L.orc_trace_any.restype = None
        L.orc_render.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]; L.orc_render.restype = C.c_double
        L.orc_render_window.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]; L.orc_render_window.restype = C.c_double
        L.orc_scene_set_light_mode.argtypes = [vp, C.c_int]; L.orc_scene_set_light_mode.restype = C.c_int
        L.orc_uses_hw_fma.argtypes = []; L.orc_uses_hw_fma.restype = C.c_int

    # -- small functions -------------------------------------------------------------
    def tea4(self, v0, v1):
        return int(self.lib.orc_tea4(v0 & 0xFFFFFFFF, v1 & 0xFFFFFFFF))

    def rnd_stream(self, seed, n):
        st = np.zeros(n, np.uint32); va = np.zeros(n, np.float32)
        self.lib.orc_rnd_stream(seed & 0xFFFFFFFF, n, st.ctypes.data, va.ctypes.data)
        return st, va

    def make_color(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
        out = np.zeros((rgb.shape[0], 4), np.uint8)
        self.lib.orc_make_color(rgb.ctypes.data, rgb.shape[0], out.ctypes.data)
        return out

    def refract(self, i, n, ior):
        i = np.ascontiguousarray(i, np.float32); n = np.ascontiguousarray(n, np.float32)
        r = np.zeros(3, np.float32); ok = C.c_int()
        self.lib.orc_refract(i.ctypes.data, n.ctypes.data, C.c_float(ior), r.ctypes.data, C.byref(ok))
        return r, bool(ok.value)

    def vec_op(self, op, a, b=None, c=None, s=0.0):
        a = np.ascontiguousarray(a, np.float32)
        b = None if b is None else np.ascontiguousarray(b, np.float32)
        c = None if c is None else np.ascontiguousarray(c, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.orc_vec_op(op, a.ctypes.data, None if b is None else b.ctypes.data, None if c is None else c.ctypes.data,
                            C.c_float(s), out.ctypes.data)
        return out

    def camera_uvw(self, eye, lookat, up, fovy, aspect):
        e = np.ascontiguousarray(eye, np.float32); l = np.ascontiguousarray(lookat, np.float32); u = np.ascontiguousarray(up, np.float32)
        U = np.zeros(3, np.float32); V = np.zeros(3, np.float32); W = np.zeros(3, np.float32)
        self.lib.orc_camera_uvw(e.ctypes.data, l.ctypes.data, u.ctypes.data, C.c_float(fovy), C.c_float(aspect),
                                U.ctypes.data, V.ctypes.data, W.ctypes.data)
        return U, V, W

    def num_samples(self, world, w, h):
        return int(self.lib.orc_num_samples(world, w, h))

    def sample_pixel(self, world, w, rank, si):
        x, y = C.c_int(), C.c_int()
        self.lib.orc_sample_pixel(world, w, rank, si, C.byref(x), C.byref(y))
        return x.value, y.value

    # -- scene / tracing / rendering ----------------------------------------------------
    def scene(self, verts, idx, mat_ids, mats):
        return OracleScene(self, verts, idx, mat_ids, mats)


class OracleScene:
    def __init__(self, orc, verts, idx, mat_ids, mats):
        self.orc = orc
        v = np.ascontiguousarray(verts, np.float32); i = np.ascontiguousarray(idx, np.uint32); m = np.ascontiguousarray(mat_ids, np.uint32)
        self._keep = (v, i, m, mats)
        self.h = orc.lib.orc_scene_create(v.ctypes.data, v.size // 4, i.ctypes.data, i.size // 3, m.ctypes.data,
                                          C.addressof(mats) if len(mats) else None, len(mats))
        if not self.h:
            raise ValueError("oracle: invalid scene (index out of range)")

    def set_light_mode(self, mode):
        """0: the reference's estimator; 1: scene lights + MIS.  Returns the number of emissive triangles."""
        return int(self.orc.lib.orc_scene_set_light_mode(self.h, int(mode)))

    def close(self):
        if self.h:
            self.orc.lib.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace_closest(self, rays, use_bvh=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        t = np.zeros(rays.shape[0], np.float32); p = np.zeros(rays.shape[0], np.uint32)
        self.orc.lib.orc_trace_closest(self.h, rays.ctypes.data, rays.shape[0], int(use_bvh), t.ctypes.data, p.ctypes.data)
        return t, p

    def trace_any(self, rays, use_bvh=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        o = np.zeros(rays.shape[0], np.uint8)
        self.orc.lib.orc_trace_any(self.h, rays.ctypes.data, rays.shape[0], int(use_bvh), o.ctypes.data)
        return o

    def render(self, params, accumulation=None, use_bvh=True, threads=0, rank=0, world=1, chunks=1):
        """One launch on the CPU.  Returns (accumulation[h,w,4] f32, framebuffer[h,w,4] u8, stats dict, seconds)."""
        h, w = int(params.height), int(params.width)
        if accumulation is None:
            accumulation = np.zeros((h, w, 4), np.float32)
        fb = np.zeros((h, w, 4), np.uint8)
        stats = np.zeros(3, np.uint64)
        if threads <= 0:
            threads = os.cpu_count() or 1
        secs = self.orc.lib.orc_render(self.h, C.byref(params), accumulation.ctypes.data, fb.ctypes.data,
                                       int(use_bvh), int(threads), int(rank), int(world), int(chunks), stats.ctypes.data)
        return accumulation, fb, {"radiance_rays": int(stats[0]), "shadow_rays": int(stats[1]), "paths": int(stats[2])}, float(secs)


_cached = None


def render_window(scene, params, window, accumulation=None, use_bvh=True, threads=0, chunks=1):
    """One launch on the CPU for the pixels of window = (x0, y0, w, h) only.  Returns (accumulation[H,W,4] with only the
    window filled, stats dict, seconds); pass the returned accumulation back in for the next frame of a progressive run."""
    h, w = int(params.height), int(params.width)
    if accumulation is None:
        accumulation = np.zeros((h, w, 4), np.float32)
    win = np.array(window, np.int32)
    stats = np.zeros(3, np.uint64)
    if threads <= 0:
        threads = os.cpu_count() or 1
    secs = scene.orc.lib.orc_render_window(scene.h, C.byref(params), accumulation.ctypes.data, None, int(use_bvh), int(threads),
                                           int(chunks), win.ctypes.data, stats.ctypes.data)
    return accumulation, {"radiance_rays": int(stats[0]), "shadow_rays": int(stats[1]), "paths": int(stats[2])}, float(secs)


def load(build_if_missing=True):
    global _cached
    if _cached is not None:
        return _cached
    name = "liboracle_pt_fma.so" if _cpu_has_fma() else "liboracle_pt.so"
    path = os.path.join(ORACLE_DIR, name)
    if not os.path.exists(path) and build_if_missing:
        import subprocess
        subprocess.run(["make", "-C", ORACLE_DIR, "liboracle_pt.so", "liboracle_pt_fma.so"], check=True, stdout=subprocess.DEVNULL)
    _cached = Oracle(C.CDLL(path), path)
    return _cached


def load_variant(name):
    return Oracle(C.CDLL(os.path.join(ORACLE_DIR, name)), name)
