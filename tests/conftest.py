import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# PyTorch's ROCm wheel carries its own HIP / HSA runtime (torch/lib/libamdhip64.so, libhsa-runtime64.so); libacgpt_hip.so links
# against /opt/rocm's.  With torch's runtime up first the two coexist (bench.py's order); in a process that used the library first,
# a later torch.cuda.init() reports "No HIP GPUs are available" (measured on the GPU box: any pt_create before the first torch
# call).  Tests that use torch next to the library (streams, lifetime, multi-rank) therefore need torch initialised before the first
# context exists — whatever subset of the files is collected.  INTEGRATION.md section 5 states the same rule for callers.
try:
    import torch
    torch.cuda.is_available()
except Exception:       # no torch: those tests skip or fail on their own
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the host library and the oracle exist (cheap no-op when up to date)."""
    from acgpathtracing_amd import _build
    _build.build_host()
    _build.build_oracle()
    return True


@pytest.fixture(scope="session")
def oracle(built):
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def gpu_state_factory():
    """Creates PathTracerState objects on cuda:0 through the C ABI and cleans them up."""
    import acgpathtracing_amd as pt
    made = []

    def make(obj_path, sample_chunks=1, math_mode="ieee", **kw):
        """sample_chunks=1: the reference's own summation order; math_mode="ieee": the arithmetic level the oracle is written
        at (what most parity tests want: bits comparable with the oracle's).  The library's default, "fast" — the arithmetic of
        the reference's own --use_fast_math build — is held against the oracle by tolerance in the tests that name it."""
        from acgpathtracing_amd import _native
        state, obj = pt.setup(obj_path, math_mode=math_mode, **kw)
        made.append(state)
        assert _native.hip().pt_set_sample_chunks(state.context, sample_chunks) == 0
        return state, obj

    yield make
    for s in made:
        pt.CleanAllTheThings(s)
