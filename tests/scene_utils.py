"""Shared helpers for the parity tests: the reference's launch presets and seeded ray sets."""
import numpy as np

import acgpathtracing_amd as pt


def make_params(w, h, spp, max_depth, direct_lighting, importance_sampling, frame=0):
    """PathTraceParams with the reference's camera (PathTracerMain.cpp:228-233, aspect = w/h)
    and hard-coded area light (PathTracerMain.cpp:154-158)."""
    p = pt.PathTraceParams()
    p.width, p.height, p.samplesPerPixel, p.maxDepth = w, h, spp, max_depth
    p.useDirectLighting = 1 if direct_lighting else 0
    p.useImportanceSampling = 1 if importance_sampling else 0
    p.currentFrameIdx = frame
    cam = pt.initCamera()
    cam.setAspectRatio(np.float32(w) / np.float32(h))
    U, V, W = cam.UVWFrame()
    f = lambda v: pt.Float3(float(v[0]), float(v[1]), float(v[2]))
    p.cameraEye, p.cameraU, p.cameraV, p.cameraW = f(cam.eye()), f(U), f(V), f(W)
    al = p.areaLight
    al.emission = pt.Float3(10.0, 10.0, 10.0)
    al.corner = pt.Float3(343.0, 547.0, 227.0)
    al.v1 = pt.Float3(0.0, 0.0, 105.0)
    al.v2 = pt.Float3(-130.0, 0.0, 0.0)
    al.normal = pt.Float3(0.0, -1.0, 0.0)      # normalize(cross(v1, v2))
    return p


def copy_params(p):
    q = pt.PathTraceParams()
    import ctypes
    ctypes.memmove(ctypes.byref(q), ctypes.byref(p), ctypes.sizeof(p))
    return q


def scene_arrays(obj):
    v = np.ascontiguousarray(obj.getVerticesFloat(), np.float32).reshape(-1, 4)
    idx = np.ascontiguousarray(obj.getIndexBuffer(), np.uint32).reshape(-1, 3)
    return v, idx


def random_rays(n, seed, lo=(0.0, 0.0, 0.0), hi=(556.0, 548.8, 559.2), tmin=0.01, tmax=1e16):
    """Origins uniform in the box, directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = o
    r[:, 3:6] = d.astype(np.float32)
    r[:, 6] = tmin
    r[:, 7] = tmax
    return r


def adversarial_rays(verts, idx, seed, n_per_kind=2000):
    """Rays that graze what a BVH gets wrong first: aimed exactly at vertices, edge midpoints and
    centroids of the scene's own triangles, axis-parallel rays, rays starting on surfaces, and
    rays with zero direction components (inf / NaN slabs)."""
    rng = np.random.default_rng(seed)
    T = idx.shape[0]
    v = verts[:, :3]
    out = []
    eye = np.array([278.0, 273.0, -900.0], np.float32)

    def aim(origins, targets):
        d = (targets - origins).astype(np.float32)
        r = np.zeros((len(d), 8), np.float32)
        r[:, 0:3] = origins; r[:, 3:6] = d; r[:, 6] = 0.01; r[:, 7] = 1e16
        return r

    pick = rng.integers(0, T, n_per_kind)
    tri = v[idx[pick]]                                    # [n,3,3]
    inside = rng.uniform((50, 50, 50), (500, 500, 500), size=(n_per_kind, 3)).astype(np.float32)
    out.append(aim(np.broadcast_to(eye, (n_per_kind, 3)), tri[:, 0]))                      # vertices from the camera
    out.append(aim(inside, tri[:, 1]))                                                     # vertices from inside
    out.append(aim(inside, ((tri[:, 0] + tri[:, 1]) * np.float32(0.5))))                   # edge midpoints
    out.append(aim(inside, tri.mean(axis=1).astype(np.float32)))                           # centroids
    # normalised versions (unit directions change the rounding)
    a = aim(inside, tri[:, 2]); a[:, 3:6] /= np.linalg.norm(a[:, 3:6], axis=1, keepdims=True); out.append(a)
    # axis-parallel rays with exact zeros in the direction
    for axis in range(3):
        for sign in (-1.0, 1.0):
            r = np.zeros((n_per_kind // 4, 8), np.float32)
            r[:, 0:3] = rng.uniform((1, 1, 1), (555, 547, 558), size=(n_per_kind // 4, 3))
            r[:, 3 + axis] = sign
            r[:, 6] = 0.01; r[:, 7] = 1e16
            out.append(r)
    # origins exactly on a wall plane, direction inside the plane (0 * inf slabs)
    r = np.zeros((n_per_kind // 4, 8), np.float32)
    r[:, 0:3] = rng.uniform((1, 0, 1), (555, 0, 558), size=(n_per_kind // 4, 3)); r[:, 1] = 0.0
    ang = rng.uniform(0, 2 * np.pi, n_per_kind // 4)
    r[:, 3] = np.cos(ang); r[:, 5] = np.sin(ang); r[:, 6] = 0.01; r[:, 7] = 1e16
    out.append(r)
    # short shadow-like segments between random surface points
    p0 = tri.mean(axis=1).astype(np.float32)
    p1 = v[idx[rng.integers(0, T, n_per_kind)]].mean(axis=1).astype(np.float32)
    d = p1 - p0
    dist = np.linalg.norm(d, axis=1).astype(np.float32)
    ok = dist > 1.0
    s = np.zeros((ok.sum(), 8), np.float32)
    s[:, 0:3] = p0[ok]; s[:, 3:6] = d[ok] / dist[ok, None]; s[:, 6] = 0.01; s[:, 7] = dist[ok] - 0.01
    out.append(s)
    return np.ascontiguousarray(np.concatenate(out, axis=0), np.float32)


def image_mse_trimmed(a, b, drop):
    """image_mse over all but the `drop` fraction of pixels that differ most.  For the comparison of two arithmetic levels in
    uniform-hemisphere mode: there three bounces in a thousand leave their surface at a slope below 3e-3, and the reference
    starts them AT the hit point with an absolute tmin of 0.01; a hit point that rounded to the far side of its surface sends
    such a ray through the surface's own plane again beyond tmin.  Which rays do is decided by the last bits of the hit point
    and of the direction, so two correct implementations that differ in those bits trace different paths for that sample — a
    whole path's radiance in one pixel, in either direction.  The trimmed value says that everything
    else agrees."""
    x = np.clip(a[..., :3].astype(np.float64), 0.0, 1.0)
    y = np.clip(b[..., :3].astype(np.float64), 0.0, 1.0)
    per_pixel = np.mean((x - y) ** 2, axis=-1).reshape(-1)
    k = int(np.ceil(drop * per_pixel.size))
    kept = np.sort(per_pixel)[:per_pixel.size - k] if k else per_pixel
    return float(kept.sum() / per_pixel.size)


def flip_report(a, b, pixel_tol=1e-6):
    """Two renders of the same samples at two arithmetic levels in uniform-hemisphere mode (image_mse_trimmed above says why single
    paths flip there): instead of trimming the worst pixels away silently, COUNT them.  A pixel is "out" when its squared error
    (mean over the channels, clamped linear radiance) exceeds pixel_tol — i.e. when it differs by more than 1e-3 of full scale.
    Returns a dict: mse (whole image), n_out, frac_out, mse_rest (the mean squared error of the other pixels, over the whole pixel
    count), mean_a / mean_b (whole image) and rest_mean_a / rest_mean_b (over the pixels that are not out)."""
    x = np.clip(a[..., :3].astype(np.float64), 0.0, 1.0)
    y = np.clip(b[..., :3].astype(np.float64), 0.0, 1.0)
    per_pixel = np.mean((x - y) ** 2, axis=-1)
    out = per_pixel > pixel_tol
    n = per_pixel.size
    rest = ~out
    return {"mse": float(per_pixel.mean()), "n_out": int(out.sum()), "frac_out": float(out.sum()) / n,
            "mse_rest": float(per_pixel[rest].sum() / n), "mean_a": float(x.mean()), "mean_b": float(y.mean()),
            "rest_mean_a": float(x[rest].mean()) if rest.any() else 0.0, "rest_mean_b": float(y[rest].mean()) if rest.any() else 0.0}


def image_mse(a, b):
    """Per-channel MSE of linear accumulation buffers clamped to [0,1] (SURVEY.md §8d parity metric)."""
    x = np.clip(a[..., :3].astype(np.float64), 0.0, 1.0)
    y = np.clip(b[..., :3].astype(np.float64), 0.0, 1.0)
    return float(np.mean((x - y) ** 2))
