#!/usr/bin/env python3
"""bench.py — Mray/s and ms/frame of the path-trace hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path at the reference's launch granularity (LaunchCurrentFrame,
PathTracerMain.cpp:184-210): 128 samples per pixel over the whole 1920x1080 image, frame index
advancing, progressive accumulation.  8 steps are one 1024-spp output frame = BASELINE.json
configs[1] (Cornell-box OBJ, 1080p, 1024 spp, 8 bounces, importance sampling + direct lighting).
The scene and BVH are resident in HBM before the timed region.

--fuse F (default 8): F consecutive steps go into ONE kernel launch (pt_launch_frames): same seeds,
same per-step progressive blend, buffers bit-identical to F separate pt_launch calls
(tests/test_gpu_parity.py::test_frame_batches_equal_separate_launches); what disappears is the
ramp-up and drain of F - 1 launches.  --fuse 1 is the reference's one-launch-per-step loop.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank renders its own 8x4 pixel
tiles (sutil/WorkDistribution.h) of the same steps, then ONE RCCL reduce of the float4
accumulation buffer to rank 0 + make_color there, all inside the timed region.  Per-GPU work
shrinks with N -> "scaling": "strong".

One JSON line on rank 0.  `roofline` prices the megakernel against HBM (SURVEY.md §8d:
B_ray = 64*ceil(log2 T) + 64 bytes per ray + 36*W*H per launch); `cpu_baseline` times the CPU
oracle (oracle/, scalar C++ restatement, std::thread over the host cores) on a bounded sample of
the same workload — reported, not the target.
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

WIDTH, HEIGHT = 1920, 1080
SPP_PER_LAUNCH = 128
MAX_DEPTH = 8
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5],
                    help="BASELINE.json config: 2 = diffuse Cornell, 1024 spp, 8 bounces (default, the headline); "
                         "3 = glass + metal, 4096 spp, 16 bounces; 5 = 1.3 M-triangle stress scene, 256 spp, 8 bounces")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP_PER_LAUNCH)
    ap.add_argument("--max-depth", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU baseline sample (0 = per config)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--variant", type=int, default=-1, help="render kernel variant (-1 = library default)")
    ap.add_argument("--fuse", type=int, default=8, help="steps per kernel launch (pt_launch_frames); 1 = one launch per step")
    ap.add_argument("--save", default="", help="write the final framebuffer as PPM (rank 0)")
    a = ap.parse_args()
    preset = {2: ("cornell_box_diffuse.obj", 8, 8), 3: ("cornell_box.obj", 32, 16), 5: ("stress_1m.obj", 2, 8)}[a.config]
    if a.scene is None:
        a.scene = preset[0]
    if a.steps is None:
        a.steps = preset[1]
    if a.max_depth is None:
        a.max_depth = preset[2]
    if a.cpu_spp <= 0:
        a.cpu_spp = {2: 64, 3: 64, 5: 32}[a.config]      # ~10-20 s of CPU work on 16 host threads
    return a


def scene_path(pt, name):
    """Scenes ship with the package; the 1.3 M-triangle stress scene is generated on demand (seeded)."""
    if os.path.isabs(name) and os.path.exists(name):
        return name
    p = os.path.join(pt.SCENES, name)
    if os.path.exists(p):
        return p
    if name == "stress_1m.obj":
        import tempfile
        sys.path.insert(0, pt.SCENES)
        import make_scenes
        out = os.path.join(tempfile.gettempdir(), "acgpt_scenes_%d" % os.getuid(), name)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        if not os.path.exists(out):
            tmp = out + ".%d.tmp.obj" % os.getpid()
            make_scenes.stress_scene(tmp)
            os.replace(tmp, out)
        return out
    raise SystemExit("unknown scene %s" % name)


def cpu_baseline(pt, obj, params, cpu_spp):
    """The oracle timed on the host cores (bounded sample: same image size, depth and toggles,
    cpu_spp samples, one frame).  Only this function touches oracle/."""
    import oracle_lib
    from scene_utils import copy_params
    orc = oracle_lib.load()
    sc = orc.scene(obj.getVerticesFloat(), obj.getIndexBuffer(), obj.getMaterialIndices(), obj.getMaterials())
    q = copy_params(params)
    q.samplesPerPixel = cpu_spp
    q.currentFrameIdx = 0
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))      # the GPU box's CPU share for one GPU
    _, _, st, secs = sc.render(q, use_bvh=True, threads=cores)
    rays = st["radiance_rays"] + st["shadow_rays"]
    sc.close()
    return {"value": rays / secs / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": "%s %dx%d maxDepth %d IS+DL, %d spp x 1 frame (%.1f s, %d rays)" %
                      (os.path.basename(obj.path), q.width, q.height, q.maxDepth, cpu_spp, secs, rays)}, rays / max(1, q.width * q.height * cpu_spp)


def main():
    a = parse()
    import torch
    import acgpathtracing_amd as pt
    from acgpathtracing_amd import _native, distributed as D
    from scene_utils import make_params

    rank, world, local_rank = D.env_rank_world()
    if a.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (a.gpus, world))
    if a.gpus > 1 and world == 1:
        raise SystemExit("N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # Rehearsal on a one-GPU box (never used by the driver): every rank shares cuda:0 and the
    # reduce goes through gloo on host copies.  Exercises partition + reduce + resolve end to end.
    rehearse = os.environ.get("ACGPT_REHEARSE_SAME_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    D.init_process_group(("gloo" if rehearse else "nccl") if world > 1 else None)
    L = _native.hip()

    # ---- scene + context (outside the timed region: inputs resident in HBM) ------------------
    obj = pt.TinyObjWrapper(scene_path(pt, a.scene))
    obj.path = a.scene
    if not obj.dataLoaded:
        raise SystemExit("cannot load scene %s" % a.scene)
    state = pt.PathTracerState()
    pt.createDeviceContext(state, local_rank)
    pt.buildTheAccelarationStructure(state, obj)
    p = make_params(a.width, a.height, a.spp, a.max_depth, True, True)
    p.handle = state.params.handle
    accum = torch.zeros((a.height, a.width, 4), dtype=torch.float32, device=dev)
    fb = torch.zeros((a.height, a.width, 4), dtype=torch.uint8, device=dev)
    p.accumulationBuffer = accum.data_ptr()
    p.frameBuffer = fb.data_ptr() if world == 1 else None
    state.params = p
    assert L.pt_set_partition(state.context, rank, world) == 0
    assert L.pt_set_stream(state.context, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    if a.blocks_per_cu or a.variant >= 0:
        assert L.pt_set_tuning(state.context, a.blocks_per_cu, max(a.variant, 0)) == 0
    info = pt.getBvhInfo(state)

    fuse = max(1, min(a.fuse, 64, a.steps))

    def launch(first_frame, n):
        state.params.currentFrameIdx = first_frame
        rc = L.pt_launch_frames(state.context, C.byref(state.params), n)
        if rc != 0:
            raise SystemExit("pt_launch_frames failed: %s" % L.pt_last_error(state.context).decode())
        return pt.getStats(state)

    # warm-up: at least `warmup` steps, in launches of the same shape as the timed ones (so a kernel trace
    # of this command averages over equal launches)
    w = 0
    while w < a.warmup:
        launch(w, fuse)
        w += fuse
    if world > 1:          # bring up the RCCL channels of the reduce outside the timed region
        if rehearse:
            host = accum.cpu(); D.reduce_accumulation(host, dst=0)
        else:
            D.reduce_accumulation(accum, dst=0)
    accum.zero_()

    # ---- timed region ---------------------------------------------------------------------------
    D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    rays = shadow = paths = 0
    kernel_ms = []
    k = 0
    while k < a.steps:
        n = min(fuse, a.steps - k)
        s = launch(k, n)
        rays += int(s.radiance_rays); shadow += int(s.shadow_rays); paths += int(s.paths)
        kernel_ms.append(float(s.kernel_ms))
        k += n
    if world > 1:
        if rehearse:
            host = accum.cpu()
            D.reduce_accumulation(host, dst=0)
            accum.copy_(host)
        else:
            D.reduce_accumulation(accum, dst=0)
        if rank == 0:
            assert L.pt_resolve_framebuffer(state.context, C.c_void_p(accum.data_ptr()), C.c_void_p(fb.data_ptr()), a.width * a.height) == 0
    torch.cuda.synchronize(); D.barrier()
    elapsed = time.perf_counter() - t0
    cdev = dev if (world > 1 and not rehearse) else None
    elapsed = D.max_over_ranks(elapsed, cdev)
    tot_rays, tot_shadow, tot_paths = D.sum_over_ranks([rays, shadow, paths], cdev)

    # ---- report ---------------------------------------------------------------------------------------
    if rank == 0:
        all_rays = tot_rays + tot_shadow
        T = max(2, info.n_tris)
        b_ray = 64 * math.ceil(math.log2(T)) + 64
        f_ray = 48 * math.ceil(math.log2(T)) + 168
        k_avg_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        n_launches = max(1, len(kernel_ms))
        my_rays_per_launch = (rays + shadow) / n_launches
        my_pixels = a.width * a.height / world
        algo_bytes = my_rays_per_launch * b_ray + 36.0 * my_pixels * (a.steps / n_launches)
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms > 0 else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        default_workload = (a.config == 2 and a.scene == "cornell_box_diffuse.obj" and (a.width, a.height, a.spp, a.max_depth) == (WIDTH, HEIGHT, SPP_PER_LAUNCH, MAX_DEPTH)
                            and world == 1 and a.variant < 0 and not a.blocks_per_cu and fuse == 8 and a.steps % 8 == 0)
        if default_workload and os.path.exists(prof):     # PMC traffic was collected on exactly this workload
            try:
                traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mray/s at 1080p, %d spp, %d bounces (radiance + shadow rays per second)" % (a.spp * a.steps, a.max_depth),
            "value": all_rays / elapsed / 1e6,
            "unit": "Mray/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed * 1e3 / a.steps,
            "ms_per_frame": elapsed * 1e3,
            "spp_per_frame": a.spp * a.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s (%d triangles), %dx%d, %d spp per step x %d steps, maxDepth %d, importance sampling + direct lighting"
                                   % (a.scene, info.n_tris, a.width, a.height, a.spp, a.steps, a.max_depth),
                       "steps_per_kernel_launch": fuse, "kernel_launches": n_launches,
                       "parallelism": "pixel tiles 8x4 over %d GPU(s)%s" % (world, ", RCCL reduce of float4 accumulation" if world > 1 else ""),
                       "rays": int(all_rays), "paths": int(tot_paths), "rays_per_path": all_rays / max(1.0, tot_paths),
                       "bvh": {"nodes": info.n_nodes, "max_depth": info.max_depth, "stack_entries": info.stack_entries, "build_ms": info.build_ms}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": "k_render", "kernel_ms_avg": k_avg_ms,
                         "algorithmic_bytes_per_ray": b_ray, "algorithmic_bytes_per_launch": algo_bytes,
                         "valu_frac_secondary": (my_rays_per_launch * f_ray / (k_avg_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS) if k_avg_ms > 0 else 0.0},
        }
        if world == 1 and not a.no_cpu_baseline:
            base, _ = cpu_baseline(pt, obj, p, a.cpu_spp)
            out["cpu_baseline"] = base
        if a.save:
            img = fb.cpu().numpy()[::-1, :, :3]
            with open(a.save, "wb") as fh:
                fh.write(b"P6\n%d %d\n255\n" % (a.width, a.height))
                fh.write(np.ascontiguousarray(img).tobytes())
        print(json.dumps(out), flush=True)
    D.barrier()
    pt_ctx = state.context
    state.params.accumulationBuffer = None
    L.pt_destroy(pt_ctx)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
