"""acgpathtracing_amd — MI355X-native drop-in for the render path of fallinbryan/ACGPathTracing.

The hot path (per-pixel Monte-Carlo launch) is hand-written HIP for gfx950 behind the C ABI in
include/acgpt.h; this package holds that library, the C++ host-side mirror of the reference's
TinyObjWrapper / Camera / Trackball, and a thin Python mirror of PathTracerMain.cpp's functions.
"""
from . import _native  # noqa: F401
from .pathtracer import *  # noqa: F401,F403
