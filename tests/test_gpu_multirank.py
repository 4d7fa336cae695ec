"""The N > 1 path on hardware, as far as a one-GPU box allows.

(a) bench.py --gpus 2 under torch.distributed.run, exactly as the driver launches it, with both ranks on the one
    GPU (ACGPT_REHEARSE_SAME_GPU=1: RCCL refuses two ranks on one device, so the reduce goes through gloo on host
    copies; everything else — pt_set_partition, the per-rank frame batches, the zero-initialised buffers, the
    reduce(SUM) to rank 0, pt_resolve_framebuffer there, max-over-ranks timing, summed counters — is the product
    path): accumulation and framebuffer must equal the one-rank run bit for bit at equal sample runs.
(b) an RCCL collective on a device tensor executes on this hardware: world 1, backend "nccl", the same
    reduce / all_reduce calls distributed.py issues (tools/rccl_selfcheck.py).
(c) the same split behind the C ABI (pt_create_multi, what a C++ caller of the drop-in uses): acgpt_main --gpus N with every
    rank on the one GPU and a sum kernel standing in for RCCL (a rehearsal, labelled so in include/acgpt.h), and the group
    context over ONE device, whose ncclReduce runs in-process through librccl with one rank; accumulation save / restore.
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


def _app(tmp_path, tag, extra, env=None, frames=5):
    """acgpt_main (the C++ mirror of PathTracerMain.cpp) on a 328 x 204 image, 16 spp per launch; returns (ppm bytes, accumulation dump bytes, stdout)."""
    exe = os.path.join(ROOT, "acgpathtracing_amd", "acgpt_main")
    if not os.path.exists(exe):
        from acgpathtracing_amd import _build
        _build.build_hip(); _build.build_main()
    out, acc = str(tmp_path / (tag + ".ppm")), str(tmp_path / (tag + ".acc"))
    cmd = [exe, "--obj", os.path.join(ROOT, "acgpathtracing_amd", "scenes", "cornell_box.obj"), "--width", "328", "--height", "204",
           "--spp-per-launch", "16", "--frames", str(frames), "--fuse-frames", "4", "--max-depth", "6", "--direct-lighting", "--importance-sampling",
           "--sample-chunks", "4", "--out", out, "--save-accum", acc] + extra
    r = _run(cmd, env or {})
    return open(out, "rb").read(), open(acc, "rb").read(), r.stdout


def test_group_context_behind_the_c_abi(built, tmp_path):
    """pt_create_multi: N devices behind the one context a caller of the drop-in holds."""
    reh = {"ACGPT_REHEARSE_SAME_GPU": "1"}
    one = _app(tmp_path, "one", [])
    assert len(one[1]) == 20 + 328 * 204 * 16 and "Total Samples 80" in one[2]
    # the group over one device: private buffer -> ncclReduce (librccl in-process, one rank) -> caller's buffer -> make_color
    solo = _app(tmp_path, "solo", ["--multi"])
    assert "Devices: 1" in solo[2]
    assert solo[:2] == one[:2], "the one-device group (RCCL reduce with one rank) differs from pt_create"
    # two and three ranks on the one GPU (rehearsal: a sum kernel stands in for RCCL); 3 exercises the per-strip rotation
    for n in (2, 3):
        grp = _app(tmp_path, "grp%d" % n, ["--gpus", str(n)], reh)
        assert "Devices: %d" % n in grp[2]
        assert grp[:2] == one[:2], "--gpus %d differs from one device" % n
    # the math mode fans out to every rank: the IEEE level on a group == on one device (and differs from the default's bits)
    one_i = _app(tmp_path, "one_ieee", ["--math", "ieee"])
    grp_i = _app(tmp_path, "grp_ieee", ["--gpus", "2", "--math", "ieee"], reh)
    assert grp_i[:2] == one_i[:2] and one_i[1] != one[1]
    # without the rehearsal switch a device listed twice is refused
    import ctypes as C
    from acgpathtracing_amd import _native
    L = _native.hip()
    ctx = C.c_void_p()
    assert "ACGPT_REHEARSE_SAME_GPU" not in os.environ
    assert L.pt_create_multi(C.byref(ctx), (C.c_int * 2)(0, 0), 2) != 0 and b"twice" in L.pt_last_error(None)
    # save / restore of the progressive state: 3 frames, then 2 more in a new process == 5 frames straight; on one device
    # and on a group (whose ranks take their own pixels of the restored buffer)
    for tag, extra, env in (("r1", [], {}), ("r2", ["--gpus", "2"], reh)):
        first = _app(tmp_path, tag + "a", extra, env, frames=3)
        rest = _app(tmp_path, tag + "b", extra + ["--restore-accum", str(tmp_path / (tag + "a.acc"))], env, frames=2)
        assert "Accumulation restored: 3 frames" in rest[2]
        assert first[1] != one[1] and rest[:2] == one[:2], tag


def _render_frames(state, frames, fuse=1):
    import ctypes as C
    import acgpathtracing_amd as pt
    for f in range(0, frames, fuse):
        pt.updateState(None, state)
        pt.LaunchCurrentFrame(None, state, sub_frames=min(fuse, frames - f))
        state.params.currentFrameIdx += min(fuse, frames - f)
    return pt.readAccumulation(state)


def test_group_context_reshaped_with_the_same_pixel_count(built):
    """a group context that goes from W x H to H x W (same pixel count, another tile layout) must not keep the old
    layout's pixels in its ranks' private buffers; and a restart at frame 0 on the same buffer starts from zero."""
    import ctypes as C
    import acgpathtracing_amd as pt
    from acgpathtracing_amd import _native
    L = _native.hip()
    obj_path = os.path.join(pt.SCENES, "cornell_box.obj")
    assert "ACGPT_REHEARSE_SAME_GPU" not in os.environ
    os.environ["ACGPT_REHEARSE_SAME_GPU"] = "1"
    try:
        grp, _ = pt.setup(obj_path, width=96, height=64, max_depth=4, direct_lighting=True, importance_sampling=True, spp=8, device_ids=[0, 0, 0])
    finally:
        del os.environ["ACGPT_REHEARSE_SAME_GPU"]
    one, _ = pt.setup(obj_path, width=96, height=64, max_depth=4, direct_lighting=True, importance_sampling=True, spp=8)
    try:
        assert L.pt_device_count(grp.context) == 3
        for st in (grp, one):
            assert L.pt_set_sample_chunks(st.context, 4) == 0
        want = _render_frames(one, 3)
        got = _render_frames(grp, 3)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
        # the same contexts, the transposed shape: 64 x 96 has the pixel count of 96 x 64 and another tile layout
        for st in (grp, one):
            cam = pt.initCamera(); cam.setAspectRatio(np.float32(64) / np.float32(96))
            U, V, W = cam.UVWFrame()
            st.params.width, st.params.height = 64, 96
            st.params.cameraU, st.params.cameraV, st.params.cameraW = pt.pathtracer._f3(U), pt.pathtracer._f3(V), pt.pathtracer._f3(W)
            st.refreshAccumulationBuffer = True
        want = _render_frames(one, 3)
        got = _render_frames(grp, 3)
        assert want.shape == (96, 64, 4) and np.array_equal(want.view(np.uint32), got.view(np.uint32)), "stale pixels of the old tile layout in the reduce"
        # restart at frame 0 into the SAME caller buffer (no reallocation): nothing of the previous sequence may survive
        for st in (grp, one):
            st.params.currentFrameIdx = 0
        want = _render_frames(one, 2)
        got = _render_frames(grp, 2)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32))
    finally:
        pt.CleanAllTheThings(grp); pt.CleanAllTheThings(one)


def test_python_save_restore_with_a_pending_refresh(built, tmp_path):
    """restoreAccumulation after setup(math_mode=...) — whose refresh flag is still pending — must not be discarded
    by the next updateState: 3 frames, save, new state, restore, 2 frames == 5 frames straight."""
    import acgpathtracing_amd as pt
    obj_path = os.path.join(pt.SCENES, "cornell_box.obj")
    kw = dict(width=80, height=48, max_depth=5, direct_lighting=True, importance_sampling=True, spp=8, math_mode="ieee")
    a, _ = pt.setup(obj_path, **kw)
    try:
        straight = _render_frames(a, 5)
    finally:
        pt.CleanAllTheThings(a)
    b, _ = pt.setup(obj_path, **kw)
    try:
        _render_frames(b, 3)
        dump = str(tmp_path / "three.acc")
        pt.saveAccumulation(b, dump)
    finally:
        pt.CleanAllTheThings(b)
    c, _ = pt.setup(obj_path, **kw)
    try:
        assert c.refreshAccumulationBuffer          # setMathMode left it pending
        pt.restoreAccumulation(c, dump)
        assert int(c.params.currentFrameIdx) == 3 and not c.refreshAccumulationBuffer
        rest = _render_frames(c, 2)
    finally:
        pt.CleanAllTheThings(c)
    assert np.array_equal(straight.view(np.uint32), rest.view(np.uint32))
