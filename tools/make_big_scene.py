#!/usr/bin/env python3
"""A larger relative of the 1.31 M-triangle stress scene (acgpathtracing_amd/scenes/make_scenes.py stress_scene): the same
Cornell shell filled with seeded icospheres, e.g. --spheres 128 --subdiv 6 = 10.5 M triangles, whose nodes + triangle records
(840 MB) no longer fit the 256 MB Infinity Cache.  Prints the path; feed it to  bench.py --config 5 --scene <path>."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "acgpathtracing_amd", "scenes"))
import make_scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--spheres", type=int, default=128)
ap.add_argument("--subdiv", type=int, default=6)
ap.add_argument("--radius-scale", type=float, default=1.0, help="< 1: smaller spheres, i.e. triangles below the resolution of the fp16 planes")
ap.add_argument("--out", default="")
a = ap.parse_args()
out = a.out or os.path.join(tempfile.gettempdir(), "acgpt_scenes_%d" % os.getuid(), "stress_%ds%d%s.obj" % (a.spheres, a.subdiv, "" if a.radius_scale == 1.0 else "_r%g" % a.radius_scale))
os.makedirs(os.path.dirname(out), exist_ok=True)
if not os.path.exists(out):
    t0 = time.time()
    tmp = out + ".%d.tmp.obj" % os.getpid()
    make_scenes.stress_scene(tmp, n_spheres=a.spheres, subdiv=a.subdiv, radius_scale=a.radius_scale)
    os.replace(tmp, out)
    print("generated %d triangles in %.1f s (%.0f MB)" % (12 + a.spheres * 20 * 4 ** a.subdiv, time.time() - t0, os.path.getsize(out) / 1e6), file=sys.stderr)
print(out)
