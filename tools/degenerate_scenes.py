#!/usr/bin/env python3
"""Large degenerate scenes through the hierarchy builders (PLOC as built, and the default with its tree optimisation): n copies of one triangle,
a flat grid of triangles in one plane, n long slivers a hair apart.  Build time and tree depth; a build that does not come back is the failure
looked for.  usage: python tools/degenerate_scenes.py {same|plane|strip|sliver} <n>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import acgpathtracing_amd as pt
from acgpathtracing_amd import _native
kind = sys.argv[1]; n = int(sys.argv[2])
path = "/tmp/degen_%s_%d.obj" % (kind, n)
with open(path, "w") as f:
    f.write("mtllib degen.mtl\nusemtl white\n")
    if kind == "same":
        f.write("v 100 100 100\nv 300 100 100\nv 100 300 100\n")
        for i in range(n): f.write("f 1 2 3\n")
    elif kind == "plane":
        k = int(np.sqrt(n / 2)) + 1
        for j in range(k + 1):
            for i in range(k + 1): f.write("v %g 50 %g\n" % (i * 500.0 / k, j * 500.0 / k))
        c = 0
        for j in range(k):
            for i in range(k):
                a = j * (k + 1) + i + 1
                f.write("f %d %d %d\nf %d %d %d\n" % (a, a + 1, a + k + 2, a, a + k + 2, a + k + 1)); c += 2
    elif kind == "strip":          # a long strip of identical unit quads at integer spacing (a fence, a staircase, a tessellated band): every merged area ties with its mirror image
        for i in range(n // 2 + 1): f.write("v %d 0 0\nv %d 1 0\n" % (i, i))
        for i in range(n // 2):
            a = 2 * i + 1
            f.write("f %d %d %d\nf %d %d %d\n" % (a, a + 2, a + 3, a, a + 3, a + 1))
    elif kind == "sliver":
        for i in range(n): f.write("v %g 0 0\nv %g 500 0.001\nv %g 0 500\n" % (i * 1e-3, i * 1e-3, i * 1e-3))
        for i in range(n): f.write("f %d %d %d\n" % (3 * i + 1, 3 * i + 2, 3 * i + 3))
open("/tmp/degen.mtl", "w").write("newmtl white\nKd 0.7 0.7 0.7\n")
os.environ["ACGPT_DEBUG_BUILD"] = "1"
for mode in (1, 2):
    t = time.time()
    state, obj = pt.setup(path, width=64, height=64, build_mode=mode)
    info = pt.getBvhInfo(state)
    print("%s n=%d mode %d: set-up %.2f s, build %.1f ms, depth %d, stack %d" % (kind, info.n_tris, mode, time.time() - t, info.build_ms, info.max_depth, info.stack_entries)); sys.stdout.flush()
    pt.CleanAllTheThings(state)
