"""The C-ABI library: it loads, exports every symbol include/acgpt.h declares, its POD layouts match
the reference's, and without a GPU it FAILS LOUDLY instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

from acgpathtracing_amd import _build, _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _build.build_hip()
    return _native.hip()


def header_symbols(name="acgpt.h"):
    text = open(os.path.join(ROOT, "include", name)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef ACGPT_EXPERIMENTS.*?#endif", "", text, flags=re.S)      # declared for libacgpt_hip_exp.so only: the product must NOT export them
    return sorted(set(re.findall(r"\b(pt_[a-z_]+)\s*\(", text)))


def test_exports_every_declared_symbol(lib):
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libacgpt_hip.so does not export %s" % s
    assert sorted(_native.ABI_SYMBOLS) == syms, "the Python binding and the header disagree"
    # test hooks and diagnostics live in their own header: a maintainer binding the render path does not see them
    hooks = header_symbols("acgpt_test.h")
    assert sorted(_native.TEST_SYMBOLS) == hooks and not set(hooks) & set(syms)
    for s in hooks:
        assert hasattr(lib, s), "libacgpt_hip.so does not export %s" % s
    out = subprocess.run(["nm", "-D", "--defined-only", _native.hip_library_path()], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (pt_[a-z_]+)", out))
    assert exported == set(syms) | set(hooks)
    assert lib.pt_abi_version() == 4 == _native.ABI_VERSION
    assert C.sizeof(_native.Stats) == 96 and C.sizeof(_native.BvhInfo) == 88


def test_pod_layouts_match_the_reference():
    assert C.sizeof(_native.PathTraceParams) == 168          # pathTracer.h:85-108 on LP64
    assert _native.PathTraceParams.accumulationBuffer.offset == 8
    assert _native.PathTraceParams.frameBuffer.offset == 16
    assert _native.PathTraceParams.width.offset == 24
    assert _native.PathTraceParams.cameraEye.offset == 40
    assert _native.PathTraceParams.areaLight.offset == 88
    assert _native.PathTraceParams.handle.offset == 152
    assert _native.PathTraceParams.useDirectLighting.offset == 160
    assert _native.PathTraceParams.useImportanceSampling.offset == 161
    assert C.sizeof(_native.Material) == 40 and _native.Material.bsdfType.offset == 36     # TinyObjWrapper.h:33-40
    assert C.sizeof(_native.AreaLight) == 60


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a machine without a GPU")
def test_no_gpu_means_loud_failure(lib):
    ctx = C.c_void_p()
    assert lib.pt_create(C.byref(ctx), 0) != 0
    assert not ctx.value
    assert b"no HIP device" in lib.pt_last_error(None)
    import acgpathtracing_amd as pt
    with pytest.raises(pt.PathTracerError):
        pt.createDeviceContext(pt.PathTracerState())


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "acgpathtracing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "liboracle" not in text and "oracle/" not in text.replace("oracle/oracle_pt.cpp", "").replace("oracle/_ref", "").replace("oracle/Makefile", ""), f


def test_render_kernels_keep_their_occupancy_budget(lib):
    """Read from the built code object (no GPU): the default render kernel and its deep-tree twin are compiled for five
    waves per SIMD — at most 96 vector registers — and the default spills none of them; the four-wave kernels stay within
    128 without spills.  A change that pushes the BVH loop over the budget shows up here, not as a silent 10 % on the bench."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_meta
    from acgpathtracing_amd import _native
    rows = [k for k in kernel_meta.kernel_table(_native.hip_library_path()) if "k_render_pw<" in k["name"]]
    assert len(rows) >= 8
    five, four = [], []
    for k in rows:
        args = [a.strip() for a in re.search(r"k_render_pw<([^>]*)>", k["name"]).group(1).split(",")]
        (five if args[4] == "5" else four).append((k, args))
    assert len(five) == 4, [k["name"] for k, _ in five]      # each in both math modes
    for k, args in five:            # the default and its large-scene twin (windowed stack): nothing in scratch memory
        assert k["vgpr_count"] <= 96, k
        assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    for k, args in four:
        assert k["vgpr_count"] <= 128 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k


def test_variant_kernel_names_are_the_code_objects(lib):
    """pt_variant_kernel(v, math mode) is what bench.py holds a committed profile against: it must be, character for character,
    the name a kernel trace prints for that variant's instantiation — i.e. a kernel that exists in the built code object; and the
    two math modes of a variant are two different kernels."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_meta
    names = [k["name"] for k in kernel_meta.kernel_table(_native.hip_library_path())]
    n = 0
    for v in range(64):
        if lib.pt_variant_name(v) is None:
            break
        kerns = [lib.pt_variant_kernel(v, m).decode() for m in (_native.MATH_IEEE, _native.MATH_FAST)]
        assert kerns[0] != kerns[1], (v, kerns)
        for kern in kerns:
            assert kern, "variant %d has no kernel name" % v
            hits = [x for x in names if ("ptd::" + kern + "(") in x]
            assert len(hits) == 1, (v, kern, [x for x in names if kern.split("<")[0] in x][:3])
        n += 1
    assert n >= 10
    assert re.fullmatch(r"[0-9a-f]{16}", lib.pt_kernel_source_hash().decode())
    assert lib.pt_kernel_source_hash().decode() == _build.kernel_source_hash()


def test_load_order_rule_is_enforced_not_only_documented():
    """torch's GPU runtime must be up before libacgpt_hip.so is loaded.  _native.hip() sees to it
    (default), leaves torch alone on request, or refuses with the reason — in fresh processes, where nothing is loaded yet."""
    import sys
    code = "import sys; from acgpathtracing_amd import _native; assert 'torch' not in sys.modules; _native.hip(); print('torch' in sys.modules)"
    def run(mode):
        env = dict(os.environ)
        env.pop("ACGPT_TORCH_FIRST", None)
        if mode is not None:
            env["ACGPT_TORCH_FIRST"] = mode
        return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    r = run(None)
    assert r.returncode == 0 and r.stdout.strip() == "True", r.stderr[-2000:]
    r = run("0")
    assert r.returncode == 0 and r.stdout.strip() == "False", r.stderr[-2000:]
    r = run("error")
    assert r.returncode != 0 and "INTEGRATION.md section 5" in r.stderr and "ACGPT_TORCH_FIRST=0" in r.stderr
