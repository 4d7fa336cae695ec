"""Shared ctypes binding for the seven sampling / BSDF helpers of pathTracerPrograms.cu (:54-85, :265-284, :341-380,
:455-476, :494-510, :534-559).  The oracle (prefix "orc_") and the reference build (prefix "ref_", oracle/_ref/libref.so
over the reference's own text) export them with identical signatures, so one mixin serves both loaders."""
import ctypes as C

import numpy as np


def _f(a, cols=None):
    a = np.ascontiguousarray(a, np.float32)
    return a.reshape(-1, cols) if cols else a.reshape(-1)


class MathMixin:
    """needs self.lib and self.prefix"""

    def _bind_math(self):
        vp, sz = C.c_void_p, C.c_size_t
        sig = {"onb_transform": [vp, vp, sz, vp], "safe_divide": [vp, vp, sz, vp], "safe_divide3": [vp, vp, sz, vp],
               "sample_hemisphere": [C.c_int, vp, vp, sz, vp], "sample_ggx": [vp, vp, vp, vp, sz, vp],
               "fresnel_conductor": [vp, vp, vp, sz, vp], "fr_dielectric": [vp, vp, vp, sz, vp]}
        for name, args in sig.items():
            fn = getattr(self.lib, self.prefix + name)
            fn.argtypes = args
            fn.restype = None

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def onb_transform(self, n3, p3):
        n3, p3 = _f(n3, 3), _f(p3, 3)
        out = np.zeros_like(n3)
        self._fn("onb_transform")(n3.ctypes.data, p3.ctypes.data, n3.shape[0], out.ctypes.data)
        return out

    def safe_divide(self, a, b):
        a, b = _f(a), _f(b)
        out = np.zeros_like(a)
        self._fn("safe_divide")(a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data)
        return out

    def safe_divide3(self, a3, b):
        a3, b = _f(a3, 3), _f(b)
        out = np.zeros_like(a3)
        self._fn("safe_divide3")(a3.ctypes.data, b.ctypes.data, b.size, out.ctypes.data)
        return out

    def sample_hemisphere(self, which, u1, u2):
        u1, u2 = _f(u1), _f(u2)
        out = np.zeros((u1.size, 3), np.float32)
        self._fn("sample_hemisphere")(int(which), u1.ctypes.data, u2.ctypes.data, u1.size, out.ctypes.data)
        return out

    def sample_ggx(self, u1, u2, roughness, n3):
        u1, u2, roughness, n3 = _f(u1), _f(u2), _f(roughness), _f(n3, 3)
        out = np.zeros_like(n3)
        self._fn("sample_ggx")(u1.ctypes.data, u2.ctypes.data, roughness.ctypes.data, n3.ctypes.data, u1.size, out.ctypes.data)
        return out

    def fresnel_conductor(self, cos_theta, eta3, k3):
        cos_theta, eta3, k3 = _f(cos_theta), _f(eta3, 3), _f(k3, 3)
        out = np.zeros_like(eta3)
        self._fn("fresnel_conductor")(cos_theta.ctypes.data, eta3.ctypes.data, k3.ctypes.data, cos_theta.size, out.ctypes.data)
        return out

    def fr_dielectric(self, cos_i, eta_i, eta_t):
        cos_i, eta_i, eta_t = _f(cos_i), _f(eta_i), _f(eta_t)
        out = np.zeros_like(cos_i)
        self._fn("fr_dielectric")(cos_i.ctypes.data, eta_i.ctypes.data, eta_t.ctypes.data, cos_i.size, out.ctypes.data)
        return out
