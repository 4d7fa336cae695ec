"""The N > 1 path on hardware, as far as a one-GPU box allows.

(a) bench.py --gpus 2 under torch.distributed.run, exactly as the driver launches it, with both ranks on the one
    GPU (ACGPT_REHEARSE_SAME_GPU=1: RCCL refuses two ranks on one device, so the reduce goes through gloo on host
    copies; everything else — pt_set_partition, the per-rank frame batches, the zero-initialised buffers, the
    reduce(SUM) to rank 0, pt_resolve_framebuffer there, max-over-ranks timing, summed counters — is the product
    path): accumulation and framebuffer must equal the one-rank run bit for bit at equal sample runs.
(b) an RCCL collective on a device tensor executes on this hardware: world 1, backend "nccl", the same
    reduce / all_reduce calls distributed.py issues (tools/rccl_selfcheck.py).
Every child is a fresh process started from pytest (nothing that touched the GPU is re-exec'ed).  A real 2..8-GPU
RCCL run is the driver's to make; DESIGN.md §6 keeps the "unmeasured on 8 GPUs" label until it has."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--config", "3", "--width", "328", "--height", "204", "--spp", "16", "--steps", "5", "--fuse", "4", "--warmup", "1",
        "--chunks", "4", "--no-cpu-baseline"]      # 328 x 204: neither a multiple of the 16 x 4 two-rank strip nor of 8


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(cmd, env_extra, timeout=300):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, "%s\n--- stdout\n%s\n--- stderr\n%s" % (" ".join(cmd), r.stdout[-3000:], r.stderr[-3000:])
    return r


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_equal_one_rank(built, tmp_path):
    one_acc, one_fb = str(tmp_path / "one.npy"), str(tmp_path / "one.ppm")
    two_acc, two_fb = str(tmp_path / "two.npy"), str(tmp_path / "two.ppm")
    r1 = _run([sys.executable, "bench.py", "--gpus", "1"] + ARGS + ["--save-accum", one_acc, "--save", one_fb], {})
    j1 = _json_line(r1.stdout)
    port = _free_port()
    r2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2"] + ARGS + ["--save-accum", two_acc, "--save", two_fb],
              {"ACGPT_REHEARSE_SAME_GPU": "1", "MASTER_ADDR": "127.0.0.1"})
    j2 = _json_line(r2.stdout)
    a1, a2 = np.load(one_acc), np.load(two_acc)
    assert a1.shape == (204, 328, 4) and np.all(a1[..., 3] == 1.0)
    assert np.array_equal(a1.view(np.uint32), a2.view(np.uint32)), "two ranks' reduced accumulation differs from one rank's"
    assert open(one_fb, "rb").read() == open(two_fb, "rb").read()
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    for k in ("rays", "paths"):
        assert j2["config"][k] == j1["config"][k], k          # SUM over ranks = the one-rank counters
    assert j1["config"]["paths"] == 328 * 204 * 16 * 5
    assert j1["config"]["sample_runs_per_pixel"] == j2["config"]["sample_runs_per_pixel"] == 4
    for j in (j1, j2):
        r = j["roofline"]
        assert 0.0 < r["frac"] <= 1.0 and r["bound"] in ("valu", "hbm") and r["unit"] in ("TFLOP/s", "GB/s")


def test_rccl_collective_runs_on_this_gpu(built):
    r = _run([sys.executable, os.path.join("tools", "rccl_selfcheck.py")], {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    assert "NCCL_OK" in r.stdout, r.stdout + r.stderr
