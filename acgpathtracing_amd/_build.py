"""Build recipes for the native pieces (all in-tree, nothing JIT-cached elsewhere).

    libacgpt_hip.so   HIP kernels + the C ABI of include/acgpt.h   (hipcc, gfx950)
    libacgpt_host.so  host-side mirror: OBJ ingest, Camera, Trackball (g++)
    acgpt_main        headless C++ app mirroring PathTracerMain.cpp  (g++, dlopens nothing:
                      links libacgpt_hip.so)
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")

HIP_SOURCES = ["capi.hip", "lbvh_build.hip", "wide_bvh.hip", "render_megakernel.hip", "selftest.hip"]
EXPERIMENT_SOURCES = ["render_wavefront.hip"]      # kernels that were measured and lost: libacgpt_hip_exp.so only
HIP_HEADERS = ["pt_device.h", "pt_shading.h", "lbvh_build.h", "render_megakernel.h", "render_common.h", "render_experiments.inc", "lbvh_experiments.inc", "selftest.h"]
HOST_SOURCES = ["host_capi.cpp", "TinyObjWrapper.cpp", "Camera.cpp", "Trackball.cpp", "ImageIO.cpp"]

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-fvisibility=hidden", "-std=c++17"]
HOST_FLAGS = ["-O2", "-std=c++14", "-fPIC", "-shared", "-ffp-contract=off", "-fvisibility=hidden", "-pthread"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _hipcc():
    for cand in ("hipcc", "/opt/rocm/bin/hipcc"):
        path = shutil.which(cand)
        if path:
            return path
    raise RuntimeError("hipcc not found")


KERNEL_SOURCES = ["render_megakernel.hip", "render_megakernel.h", "render_common.h", "pt_device.h", "pt_shading.h", "lbvh_build.hip", "lbvh_build.h"]


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources of the render kernels: compiled into the library
    (pt_kernel_source_hash) and recorded by tools/summarize_prof.py, so that bench.py quotes a committed profile only
    for the kernel code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def build_hip(force=False, verbose=False, experiments=False):
    """experiments=True builds libacgpt_hip_exp.so: the same library plus every kernel variant that was measured and
    not adopted (-DACGPT_EXPERIMENTS; tools/sweep_variants.py loads it via ACGPT_EXPERIMENTS=1).  Never the product.
    By default only the current round's experiments are compiled in; ACGPT_EXPERIMENTS_ALL=1 adds the ~70 of earlier rounds."""
    out = os.path.join(PKG, "libacgpt_hip_exp.so" if experiments else "libacgpt_hip.so")
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES + (EXPERIMENT_SOURCES if experiments else [])]
    deps = srcs + [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(ROOT, "include", "acgpt.h"), os.path.join(ROOT, "include", "acgpt_test.h")]
    if force or _stale(out, deps):
        cmd = [_hipcc()] + HIP_FLAGS + (["-DACGPT_EXPERIMENTS=%d" % (2 if os.environ.get("ACGPT_EXPERIMENTS_ALL") == "1" else 1)] if experiments else []) + \
              ['-DACGPT_KERNEL_SRC_HASH="%s"' % kernel_source_hash(), "-o", out] + srcs + ["-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT if not verbose else None)
    return out


def build_host(force=False, verbose=False):
    out = os.path.join(PKG, "libacgpt_host.so")
    srcs = [os.path.join(HOST, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(HOST, h) for h in os.listdir(HOST) if h.endswith(".h")]
    if force or _stale(out, deps):
        cmd = ["g++"] + HOST_FLAGS + ["-o", out] + srcs
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return out


def build_main(force=False, verbose=False):
    """Headless app; needs libacgpt_hip.so next to it (rpath $ORIGIN)."""
    src = os.path.join(HOST, "PathTracerMain.cpp")
    if not os.path.exists(src):
        return None
    out = os.path.join(PKG, "acgpt_main")
    srcs = [src] + [os.path.join(HOST, s) for s in ("TinyObjWrapper.cpp", "Camera.cpp", "Trackball.cpp", "ImageIO.cpp")]
    deps = srcs + [os.path.join(HOST, h) for h in os.listdir(HOST) if h.endswith(".h")] + [os.path.join(ROOT, "include", "acgpt.h")]
    if force or _stale(out, deps):
        cmd = ["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-o", out] + srcs + \
              ["-L" + PKG, "-lacgpt_hip", "-Wl,-rpath,$ORIGIN", "-pthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return out


def build_oracle(force=False, verbose=False):
    """Test infrastructure: the CPU checker (and, where /root/reference exists, oracle/_ref)."""
    odir = os.path.join(ROOT, "oracle")
    cmd = ["make", "-C", odir] + (["-B"] if force else [])
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return odir


def build_all(force=False, verbose=False):
    build_host(force, verbose)
    build_hip(force, verbose)
    build_main(force, verbose)
