#!/usr/bin/env python3
"""Author the scene assets (the reference ships none: its OBJ path is an absolute
C:\\ path, PathTracer_Optix/PathTracerMain.cpp:50).

Geometry: the classic 555-unit Cornell box implied by the reference's hard-coded
camera (PathTracerMain.cpp:228-233) and area light (PathTracerMain.cpp:154-158):
five walls, an emissive ceiling quad just above the y=547 light rectangle, the tall
and short blocks, a faceted glass icosphere front-left and a bumpy "monkey stand-in"
mesh on the short block (the reference screenshots show a Suzanne there).

Outputs (deterministic, committed):
  cornell_box.obj / .mtl            glass -> material "glass_Refractive", blob -> "purple_Metallic"
  cornell_box_diffuse.obj / .mtl    same geometry, material names without those substrings
                                    (BSDF is chosen by NAME, TinyObjWrapper.cpp:150-162)
The ~1.3 M-triangle stress scene is generated on demand by stress_scene() (not committed).
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def icosphere(subdiv):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
         (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    v = [tuple(np.array(p, dtype=np.float64) / np.linalg.norm(p)) for p in v]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
         (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
         (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(subdiv):
        cache = {}
        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (np.array(v[a]) + np.array(v[b])) * 0.5
                m /= np.linalg.norm(m)
                v.append(tuple(m))
                cache[key] = len(v) - 1
            return cache[key]
        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v, dtype=np.float64), np.array(f, dtype=np.int64)


def blob(nu, nv):
    """UV sphere displaced by a few sinusoids: a ~Suzanne-sized (2*nu*(nv-1) tris) closed mesh."""
    verts = [(0.0, 1.0, 0.0)]
    for j in range(1, nv):
        th = math.pi * j / nv
        for i in range(nu):
            ph = 2.0 * math.pi * i / nu
            verts.append((math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph)))
    verts.append((0.0, -1.0, 0.0))
    v = np.array(verts, dtype=np.float64)
    r = 1.0 + 0.18 * np.sin(5.0 * v[:, 0] + 1.0) * np.cos(4.0 * v[:, 1]) + 0.12 * np.sin(7.0 * v[:, 2] + 0.5)
    v = v * r[:, None]
    f = []
    for i in range(nu):
        f.append((0, 1 + (i + 1) % nu, 1 + i))
    for j in range(nv - 2):
        a0 = 1 + j * nu
        b0 = a0 + nu
        for i in range(nu):
            i1 = (i + 1) % nu
            f.append((a0 + i, a0 + i1, b0 + i1))
            f.append((a0 + i, b0 + i1, b0 + i))
    last = len(verts) - 1
    a0 = 1 + (nv - 2) * nu
    for i in range(nu):
        f.append((last, a0 + i, a0 + (i + 1) % nu))
    return v, np.array(f, dtype=np.int64)


class ObjWriter:
    def __init__(self, mtllib):
        self.lines = ["# authored by acgpathtracing_amd/scenes/make_scenes.py", "mtllib %s" % mtllib]
        self.nv = 0

    def verts(self, vs):
        base = self.nv + 1
        for p in vs:
            self.lines.append("v %.6f %.6f %.6f" % (p[0], p[1], p[2]))
        self.nv += len(vs)
        return base

    def obj(self, name, mtl):
        self.lines.append("o %s" % name)
        self.lines.append("usemtl %s" % mtl)

    def face(self, idx):
        self.lines.append("f " + " ".join(str(i) for i in idx))

    def quad(self, name, mtl, q):
        self.obj(name, mtl)
        b = self.verts(q)
        self.face([b, b + 1, b + 2, b + 3])

    def mesh(self, name, mtl, v, f):
        self.obj(name, mtl)
        b = self.verts(v)
        for t in f:
            self.face([b + int(t[0]), b + int(t[1]), b + int(t[2])])

    def text(self):
        return "\n".join(self.lines) + "\n"


def block_quads(pts):
    """pts: 5 quads (top + 4 sides), classic Cornell data."""
    return pts


def cornell(glass_name, metal_name, mtllib):
    w = ObjWriter(mtllib)
    w.quad("floor", "white", [(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)])
    w.quad("ceiling", "white", [(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)])
    w.quad("back", "white", [(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)])
    w.quad("right", "green", [(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)])
    w.quad("left", "red", [(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)])
    # emissive quad just above the hard-coded light rectangle at y = 547
    w.quad("lamp", "light", [(343, 548.3, 227), (343, 548.3, 332), (213, 548.3, 332), (213, 548.3, 227)])
    short = [
        [(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)],
        [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)],
        [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)],
        [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)],
        [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)],
    ]
    tall = [
        [(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)],
        [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)],
        [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)],
        [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)],
        [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)],
    ]
    for k, q in enumerate(short):
        w.quad("short_block_%d" % k, "white", q)
    for k, q in enumerate(tall):
        w.quad("tall_block_%d" % k, "white", q)
    # faceted glass sphere, front-left of the image (image-left is +x: cameraU points to -x)
    sv, sf = icosphere(2)
    w.mesh("glass_sphere", glass_name, sv * 70.0 + np.array([420.0, 70.5, 130.0]), sf)
    # bumpy blob on the short block (stand-in for the reference's Suzanne)
    bv, bf = blob(24, 20)
    w.mesh("blob", metal_name, bv * 52.0 + np.array([186.0, 165.0 + 62.0, 168.0]), bf)
    return w.text()


MTL_COMMON = """newmtl white
Kd 0.725 0.71 0.68
newmtl red
Kd 0.63 0.065 0.05
newmtl green
Kd 0.14 0.45 0.091
newmtl light
Kd 0.78 0.78 0.78
Ke 17 12 4
"""


def mtl(glass_name, metal_name):
    return ("# authored by acgpathtracing_amd/scenes/make_scenes.py\n" + MTL_COMMON +
            "newmtl %s\nKd 0.95 0.95 0.95\nNi 1.5\n" % glass_name +
            "newmtl %s\nKd 0.8 0.5 0.9\nPr 0.2\nPm 1.0\n" % metal_name)


def stress_scene(path_obj, n_spheres=64, subdiv=5, seed=0x9E3779B9, mtl_name="stress_scene.mtl", radius_scale=1.0):
    """Seeded ~1.3 M-triangle scene inside the Cornell shell: n_spheres icospheres of
    20*4^subdiv triangles each (64 x 20480 = 1 310 720).  Positions from an LCG
    (cuda/random.h constants) with a fixed seed.  Writes path_obj and a sibling .mtl."""
    state = seed & 0xFFFFFFFF

    def rnd():
        nonlocal state
        state = (1664525 * state + 1013904223) & 0xFFFFFFFF
        return (state & 0xFFFFFF) / 16777216.0

    w = ObjWriter(mtl_name)
    w.quad("floor", "white", [(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)])
    w.quad("ceiling", "white", [(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)])
    w.quad("back", "white", [(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)])
    w.quad("right", "green", [(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)])
    w.quad("left", "red", [(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)])
    w.quad("lamp", "light", [(343, 548.3, 227), (343, 548.3, 332), (213, 548.3, 332), (213, 548.3, 227)])
    sv, sf = icosphere(subdiv)
    mats = ["white", "red", "green"]
    for k in range(n_spheres):
        r = (25.0 + 30.0 * rnd()) * radius_scale      # radius_scale < 1: triangles below the fp16 planes' resolution (tools/make_big_scene.py)
        c = np.array([60.0 + 430.0 * rnd(), 40.0 + 440.0 * rnd(), 60.0 + 440.0 * rnd()])
        w.mesh("s%03d" % k, mats[k % 3], sv * r + c, sf)
    with open(path_obj, "w") as fh:
        fh.write(w.text())
    with open(os.path.join(os.path.dirname(os.path.abspath(path_obj)), mtl_name), "w") as fh:
        fh.write("# stress scene materials\n" + MTL_COMMON)


def main():
    out = HERE
    with open(os.path.join(out, "cornell_box.obj"), "w") as fh:
        fh.write(cornell("glass_Refractive", "purple_Metallic", "cornell_box.mtl"))
    with open(os.path.join(out, "cornell_box.mtl"), "w") as fh:
        fh.write(mtl("glass_Refractive", "purple_Metallic"))
    with open(os.path.join(out, "cornell_box_diffuse.obj"), "w") as fh:
        fh.write(cornell("glass", "purple", "cornell_box_diffuse.mtl"))
    with open(os.path.join(out, "cornell_box_diffuse.mtl"), "w") as fh:
        fh.write(mtl("glass", "purple"))
    if len(sys.argv) > 1 and sys.argv[1] == "--stress":
        stress_scene(sys.argv[2] if len(sys.argv) > 2 else os.path.join(out, "stress_1m.obj"))


if __name__ == "__main__":
    main()
