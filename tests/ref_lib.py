"""ctypes loader of oracle/_ref/libref.so: the REFERENCE'S OWN host-compilable sources
(cuda/random.h, cuda/helpers.h, sutil/vec_math.h, sutil/Camera.cpp, sutil/Trackball.cpp,
sutil/WorkDistribution.h, PathTracer_Optix/TinyObjWrapper.cpp) built by oracle/Makefile.
Exists only where /root/reference exists; tests that need it skip otherwise and rely on the
committed fixtures in tests/golden/ instead."""
import ctypes as C
import os

import numpy as np

from math_binding import MathMixin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libref.so")


def available():
    return os.path.exists(PATH)


class Ref(MathMixin):
    prefix = "ref_"

    def __init__(self):
        L = self.lib = C.CDLL(PATH)
        self._bind_math()
        vp, sz = C.c_void_p, C.c_size_t
        L.ref_tea4.argtypes = [C.c_uint32, C.c_uint32]; L.ref_tea4.restype = C.c_uint32
        L.ref_rnd_stream.argtypes = [C.c_uint32, sz, vp, vp]; L.ref_rnd_stream.restype = None
        L.ref_make_color.argtypes = [vp, sz, vp]; L.ref_make_color.restype = None
        L.ref_refract.argtypes = [vp, vp, C.c_float, vp, vp]; L.ref_refract.restype = None
        L.ref_vec_op.argtypes = [C.c_int, vp, vp, vp, C.c_float, vp]; L.ref_vec_op.restype = None
        L.ref_camera_uvw.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]; L.ref_camera_uvw.restype = None
        L.ref_trackball_script.argtypes = [vp, vp, vp, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, vp, sz, vp]
        L.ref_trackball_script.restype = None
        L.ref_num_samples.argtypes = [C.c_int] * 3; L.ref_num_samples.restype = C.c_int
        L.ref_sample_pixel.argtypes = [C.c_int] * 5 + [vp, vp]; L.ref_sample_pixel.restype = None
        L.ref_obj_load.argtypes = [C.c_char_p]; L.ref_obj_load.restype = vp
        L.ref_obj_ok.argtypes = [vp]; L.ref_obj_ok.restype = C.c_int
        L.ref_obj_sizes.argtypes = [vp] + [C.POINTER(sz)] * 4; L.ref_obj_sizes.restype = None
        L.ref_obj_fill.argtypes = [vp, vp, vp, vp, vp]; L.ref_obj_fill.restype = None
        L.ref_obj_free.argtypes = [vp]; L.ref_obj_free.restype = None

    def tea4(self, a, b):
        return int(self.lib.ref_tea4(a & 0xFFFFFFFF, b & 0xFFFFFFFF))

    def rnd_stream(self, seed, n):
        st = np.zeros(n, np.uint32); va = np.zeros(n, np.float32)
        self.lib.ref_rnd_stream(seed & 0xFFFFFFFF, n, st.ctypes.data, va.ctypes.data)
        return st, va

    def make_color(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32).reshape(-1, 3)
        out = np.zeros((rgb.shape[0], 4), np.uint8)
        self.lib.ref_make_color(rgb.ctypes.data, rgb.shape[0], out.ctypes.data)
        return out

    def refract(self, i, n, ior):
        i = np.ascontiguousarray(i, np.float32); n = np.ascontiguousarray(n, np.float32)
        r = np.zeros(3, np.float32); ok = C.c_int()
        self.lib.ref_refract(i.ctypes.data, n.ctypes.data, C.c_float(ior), r.ctypes.data, C.byref(ok))
        return r, bool(ok.value)

    def vec_op(self, op, a, b=None, c=None, s=0.0):
        a = np.ascontiguousarray(a, np.float32)
        b = None if b is None else np.ascontiguousarray(b, np.float32)
        c = None if c is None else np.ascontiguousarray(c, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.ref_vec_op(op, a.ctypes.data, None if b is None else b.ctypes.data, None if c is None else c.ctypes.data,
                            C.c_float(s), out.ctypes.data)
        return out

    def camera_uvw(self, eye, lookat, up, fovy, aspect):
        e = np.ascontiguousarray(eye, np.float32); l = np.ascontiguousarray(lookat, np.float32); u = np.ascontiguousarray(up, np.float32)
        U = np.zeros(3, np.float32); V = np.zeros(3, np.float32); W = np.zeros(3, np.float32)
        self.lib.ref_camera_uvw(e.ctypes.data, l.ctypes.data, u.ctypes.data, C.c_float(fovy), C.c_float(aspect),
                                U.ctypes.data, V.ctypes.data, W.ctypes.data)
        return U, V, W

    def trackball_script(self, eye, lookat, up, fovy, aspect, view_mode, move_speed, gimbal_lock, cw, ch, events):
        e = np.ascontiguousarray(eye, np.float32); l = np.ascontiguousarray(lookat, np.float32); u = np.ascontiguousarray(up, np.float32)
        ev = np.ascontiguousarray(events, np.int32).reshape(-1, 3)
        out = np.zeros(9, np.float32)
        self.lib.ref_trackball_script(e.ctypes.data, l.ctypes.data, u.ctypes.data, C.c_float(fovy), C.c_float(aspect), view_mode,
                                      C.c_float(move_speed), gimbal_lock, cw, ch, ev.ctypes.data, ev.shape[0], out.ctypes.data)
        return out

    def num_samples(self, world, w, h):
        return int(self.lib.ref_num_samples(world, w, h))

    def sample_pixel(self, world, w, h, rank, si):
        x, y = C.c_int(), C.c_int()
        self.lib.ref_sample_pixel(world, w, h, rank, si, C.byref(x), C.byref(y))
        return x.value, y.value

    def obj_load(self, path):
        L = self.lib
        h = L.ref_obj_load(os.fsencode(path))
        try:
            if not L.ref_obj_ok(h):
                return None
            sizes = [C.c_size_t() for _ in range(4)]
            L.ref_obj_sizes(h, *[C.byref(s) for s in sizes])
            nv, ni, nm, nmat = [s.value for s in sizes]
            v = np.zeros(nv, np.float32); i = np.zeros(ni, np.uint32); m = np.zeros(nm, np.uint32); mm = np.zeros(nmat * 10, np.uint32)
            L.ref_obj_fill(h, v.ctypes.data, i.ctypes.data, m.ctypes.data, mm.ctypes.data)
            return v, i, m, mm
        finally:
            L.ref_obj_free(h)
