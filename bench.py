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

One JSON line on rank 0.

`roofline` prices the megakernel against the resource that can bind it.  The kernel is BVH pointer
chasing + fp32 shading: no matrix work, so MFMA never applies.  HBM applies only when the scene
(nodes + triangle records) exceeds the 256 MB Infinity Cache; a scene that fits the 32 MB of L2 (every
Cornell-class input: 140 KB) or the Infinity Cache is served on chip, and the SURVEY.md §8d byte model
(B_ray = 64*ceil(log2 T) + 64 bytes per ray + 36*W*H per step) then describes cache traffic, not HBM —
it is reported as `hbm_model` and never as `frac`.  For those scenes `bound` is "valu": achieved =
algorithmic fp32 flops (F_ray = 48*ceil(log2 T) + 168 per ray) over the kernel's
HIP-event time, peak = 157.3 TFLOP/s fp32 vector.  Flops and bytes are priced on the rays that are
TRAVERSED: camera rays that end at the scene's bounding box (pt_stats.culled_rays; they are part of the
Mray/s figure, SURVEY.md §8d counts every radiance segment) carry none.  `traffic` = measured HBM bytes per
kernel launch and `measured` = the unit-busy fractions, both from the committed PMC summary of this
config's default command (profiles/, separate --pmc passes, gfx950 FETCH_SIZE correction) — attached only
if that profile was taken on the kernel that ran here: same instantiation (pt_variant_kernel) and same
kernel sources (pt_kernel_source_hash); a stale profile drops out.
`cpu_baseline` times the CPU oracle (oracle/, scalar C++ restatement, std::thread over the host cores)
on a bounded sample of the same workload — reported, not the target.
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
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: fp32 vector (non-matrix) peak
L2_BYTES = 8 * 4 * 1024 * 1024       # 4 MB per XCD
MALL_BYTES = 256 * 1024 * 1024       # Infinity Cache


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[0, 2, 3, 5],
                    help="BASELINE.json config: 2 = diffuse Cornell, 1024 spp, 8 bounces (default, the headline); "
                         "3 = glass + metal, 4096 spp, 16 bounces; 5 = 1.3 M-triangle stress scene, 256 spp, 8 bounces; "
                         "0 = the one workload the reference's own program defines (PathTracerMain.cpp:43, 58-59, 653-657): cornell_box.obj, "
                         "512x512, 128 spp per launch, maxDepth 4, direct lighting off, importance sampling off")
    ap.add_argument("--scene", default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--direct-lighting", type=int, default=None, choices=[0, 1], help="useDirectLighting (default: per config)")
    ap.add_argument("--importance-sampling", type=int, default=None, choices=[0, 1], help="useImportanceSampling (default: per config)")
    ap.add_argument("--spp", type=int, default=SPP_PER_LAUNCH)
    ap.add_argument("--max-depth", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="samples per pixel of the CPU baseline sample (0 = per config)")
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--variant", type=int, default=-1, help="render kernel variant (-1 = library default)")
    ap.add_argument("--math", default="fast", choices=["fast", "ieee"],
                    help="arithmetic of the shading code (pt_set_math_mode): fast = what the reference's own build uses (nvcc --use_fast_math, the library default); ieee = the CPU oracle's level")
    ap.add_argument("--no-ieee-leg", action="store_true", help="skip the untimed launches in the other math mode that the report quotes beside the timed one (N = 1 only)")
    ap.add_argument("--fuse", type=int, default=8, help="steps per kernel launch (pt_launch_frames); 1 = one launch per step")
    ap.add_argument("--chunks", type=int, default=0, help="sample runs per pixel (pt_set_sample_chunks); 0 = automatic")
    ap.add_argument("--save", default="", help="write the final framebuffer as PPM (rank 0)")
    ap.add_argument("--save-accum", default="", help="write the final float4 accumulation buffer as .npy (rank 0)")
    a = ap.parse_args()
    preset = PRESETS[a.config]
    if a.scene is None:
        a.scene = preset[0]
    if a.steps is None:
        a.steps = preset[1]
    if a.max_depth is None:
        a.max_depth = preset[2]
    if a.width is None:
        a.width = preset[3]
    if a.height is None:
        a.height = preset[4]
    if a.direct_lighting is None:
        a.direct_lighting = preset[5]
    if a.importance_sampling is None:
        a.importance_sampling = preset[6]
    if a.cpu_spp <= 0:
        a.cpu_spp = {0: 256, 2: 64, 3: 64, 5: 32}[a.config]      # ~10-20 s of CPU work on 16 host threads
    return a


# config -> (scene, steps, maxDepth, width, height, direct lighting, importance sampling).  Config 0 is the reference program's own
# start-up state (PathTracerMain.cpp:43 samples_per_launch 128, :58-59 512 x 512, :653-657 depth 4 / both toggles off); its 8
# steps are 8 iterations of the reference's frame loop (:700-730), whose "Frame Render Time" print (:726) is one step.
PRESETS = {0: ("cornell_box.obj", 8, 4, 512, 512, 0, 0), 2: ("cornell_box_diffuse.obj", 8, 8, WIDTH, HEIGHT, 1, 1),
           3: ("cornell_box.obj", 32, 16, WIDTH, HEIGHT, 1, 1), 5: ("stress_1m.obj", 2, 8, WIDTH, HEIGHT, 1, 1)}


def toggles_text(a):
    return {(1, 1): "importance sampling + direct lighting", (1, 0): "direct lighting, uniform hemisphere sampling",
            (0, 1): "importance sampling, no direct lighting", (0, 0): "no direct lighting, uniform hemisphere sampling (the reference's start-up toggles)"}[(int(a.direct_lighting), int(a.importance_sampling))]


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
            "sample": "%s %dx%d maxDepth %d DL %d IS %d, %d spp x 1 frame (%.1f s, %d rays)" %
                      (os.path.basename(obj.path), q.width, q.height, q.maxDepth, int(q.useDirectLighting), int(q.useImportanceSampling), cpu_spp, secs, rays)}, rays / max(1, q.width * q.height * cpu_spp)


def primary_miss_fraction(p, info):
    """Fraction of the pixels whose camera ray (through the pixel centre) misses the scene's bounding box: those
    paths are one ray that fails at the root.  Part of the workload BASELINE config 2 defines (a 16:9 frame around a
    square box), stated so that the Mray/s figure can be read with it."""
    W, H = int(p.width), int(p.height)
    xs = (np.arange(W, dtype=np.float64) + 0.5) / W * 2.0 - 1.0
    ys = (np.arange(H, dtype=np.float64) + 0.5) / H * 2.0 - 1.0
    f = lambda v: np.array([v.x, v.y, v.z], np.float64)
    eye, U, V, Wv = f(p.cameraEye), f(p.cameraU), f(p.cameraV), f(p.cameraW)
    d = xs[None, :, None] * U + ys[:, None, None] * V + Wv
    lo = np.array(info.scene_lo, np.float64); hi = np.array(info.scene_hi, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo - eye) / d; t1 = (hi - eye) / d
    tn = np.nanmax(np.minimum(t0, t1), axis=-1); tf = np.nanmin(np.maximum(t0, t1), axis=-1)
    return float(1.0 - ((tn <= tf) & (tf > 0)).mean())


def pmc_summary(config):
    """The committed PMC summary of this config's default bench command (tools/profile_bench.sh +
    tools/summarize_prof.py): newest profiles/r*_c<config>_summary.json."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_c%d_summary.json" % config)))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


GATHER_PEAK_GBS = 7400.0        # MI355X_MICROARCH.md: measured random-gather rate out of the Infinity Cache (7.4-7.9 TB/s, 151 MB table)


def roofline_block(a, info, world, fuse, kernel_ms, traced_per_launch, counted_per_launch, steps_per_launch, variant_name, variant_kernel, source_hash):
    """traced_per_launch: rays that enter the BVH loop (radiance + shadow - culled); counted_per_launch: every ray of the
    SURVEY.md 8d definition (what `value` counts)."""
    T = max(2, info.n_tris)
    levels = math.ceil(math.log2(T))
    b_ray = 64 * levels + 64
    f_ray = 48 * levels + 168
    k_avg_ms = sum(kernel_ms) / max(1, len(kernel_ms))
    ksec = k_avg_ms * 1e-3
    my_pixels = a.width * a.height / world
    algo_bytes = traced_per_launch * b_ray + 36.0 * my_pixels * steps_per_launch
    algo_flops = traced_per_launch * f_ray
    half = "fp16" in variant_name                    # the node array the kernel that ran walks: 32-byte fp16 or 64-byte fp32 nodes
    scene_bytes = int(info.half_node_bytes if half else info.node_bytes) + int(info.tri_bytes)
    kernel_b_ray = (32 if half else 64) * levels + 48   # what this kernel's formats move per ray of the SURVEY model: one node per level + one triangle record
    kernel_bytes = traced_per_launch * kernel_b_ray + 36.0 * my_pixels * steps_per_launch
    kernel_gbs = kernel_bytes / ksec / 1e9 if ksec > 0 else 0.0
    resident = "L2" if scene_bytes <= L2_BYTES else ("Infinity Cache" if scene_bytes <= MALL_BYTES else "HBM")
    model_gbs = algo_bytes / ksec / 1e9 if ksec > 0 else 0.0
    valu_tf = algo_flops / ksec / 1e12 if ksec > 0 else 0.0
    valu_tf_counted = counted_per_launch * f_ray / ksec / 1e12 if ksec > 0 else 0.0
    summ, src = pmc_summary(a.config)
    pre = PRESETS[a.config]
    default_cmd = (a.scene == pre[0] and (a.width, a.height, a.spp) == (pre[3], pre[4], SPP_PER_LAUNCH) and world == 1 and a.variant < 0
                   and (int(getattr(a, "direct_lighting", pre[5])), int(getattr(a, "importance_sampling", pre[6]))) == (pre[5], pre[6])
                   and getattr(a, "max_depth", pre[2]) == pre[2]
                   and not a.blocks_per_cu and a.fuse == 8 and a.chunks == 0)
    traffic, measured, fabric, dropped = None, None, None, None
    if summ and default_cmd:
        prof_kernel = summ.get("kernel_stats", {}).get("name", "")
        same_kernel = bool(variant_kernel) and variant_kernel in prof_kernel
        same_source = summ.get("kernel_source_hash") == source_hash
        if same_kernel and same_source:
            der = summ.get("derived", {})
            traffic = der.get("hbm_bytes_per_launch")
            prof_ms = summ.get("kernel_stats", {}).get("avg_ms")
            measured = {"source": src, "kernel": prof_kernel, "kernel_source_hash": source_hash,
                        "kernel_ms_avg_rocprof": prof_ms,
                        "steps_per_launch_profiled": summ.get("steps_per_kernel_launch", 8),
                        "ta_busy": der.get("ta_busy_frac(256 TAs)"),
                        "valu_issue_busy": der.get("valu_issue_busy_frac(2cyc/instr,1024 SIMDs)"),
                        "valu_issue_busy_mix": der.get("valu_issue_busy_mix"), "valu_issue_busy_ubench": der.get("valu_issue_busy_ubench"),
                        "lane_utilisation": der.get("valu_lane_utilisation"),
                        "l2_hit": der.get("l2_hit_rate"), "l1_miss_per_access": der.get("l1_miss_per_access"),
                        "hbm_GBps": (traffic / (prof_ms * 1e-3) / 1e9) if traffic and prof_ms else None}
            if resident == "Infinity Cache" and traffic and prof_ms:
                # a scene the Infinity Cache holds: what moves between L2 and the fabric is gather traffic, priced against the
                # guide's measured gather rate out of that cache, not against HBM
                gbs = traffic / (prof_ms * 1e-3) / 1e9
                fabric = {"GBps": gbs, "peak": GATHER_PEAK_GBS, "frac": gbs / GATHER_PEAK_GBS,
                          "note": "L2 <-> fabric bytes of the profiled launch over its rocprof time, against the 7.4-7.9 TB/s random-gather rate of the Infinity Cache (MI355X_MICROARCH.md)"}
        else:
            dropped = {"source": src, "reason": ("profiled kernel %r is not the %r that ran" % (prof_kernel[:120], variant_kernel)) if not same_kernel
                       else ("profile taken on kernel sources %s, this library is built from %s" % (summ.get("kernel_source_hash"), source_hash))}
    r = {"kernel": variant_name, "kernel_instantiation": variant_kernel, "kernel_source_hash": source_hash,
         "kernel_ms_avg": k_avg_ms, "scene_bytes": scene_bytes, "scene_resident_in": resident,
         "traffic": traffic, "measured": measured,
         "traffic_note": ("bytes between L2 and the fabric per kernel launch (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section); "
                          + ("reads that the 256 MB Infinity Cache serves are included: an upper bound on HBM traffic" if resident == "Infinity Cache"
                             else "HBM traffic")),
         "rays_priced": "traversed rays only (radiance + shadow - culled camera rays)",
         "hbm_model": {"algorithmic_bytes_per_ray": b_ray, "algorithmic_bytes_per_launch": algo_bytes, "GBps": model_gbs,
                       "note": "SURVEY.md 8d byte model; for a cache-resident scene these bytes are served by L1/L2/Infinity Cache, so this is not an HBM fraction"},
         "valu_model": {"algorithmic_flops_per_ray": f_ray, "TFLOPs": valu_tf, "frac_of_fp32_vector_peak": valu_tf / FP32_PEAK_TFLOPS}}
    if measured and measured.get("valu_issue_busy_mix") is not None and measured.get("ta_busy") is not None:
        # what binds, as measured: the SIMDs' issue slots (vector instruction count priced by opcode class: full / half / quarter rate
        # at 2 / 4 / 8 cycles; tools/valu_mix.py) against the texture-address units, at the fraction of lanes that do useful work
        vi, ta = measured["valu_issue_busy_mix"], measured["ta_busy"]
        r["issue"] = {"valu_issue_busy_mix": vi, "ta_busy": ta, "lane_utilisation": measured.get("lane_utilisation"),
                      "valu_issue_busy_ubench": measured.get("valu_issue_busy_ubench"),
                      "note": "valu_issue_busy_mix: SQ_INSTS_VALU priced by opcode class (2 / 4 / 8 cycles) over 1024 SIMDs x kernel cycles; _ubench: the same count at the "
                              "cycles profiles/r02_ubench_valu.txt measures per class in isolation (upper bound); ta_busy: TA_TA_BUSY over 256 units"}
        r["bound_measured"] = "vector issue" if vi >= ta else "texture-address path"
    if dropped:
        r["profile_dropped"] = dropped
    if fabric:
        r["fabric"] = fabric
    if resident == "HBM":
        # priced with the bytes of the node format that ran (an fp16 node is half the SURVEY model's 64 B), so that a compact
        # format does not read as bandwidth
        r.update({"bound": "hbm", "achieved": kernel_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kernel_gbs / HBM_PEAK_GBS,
                  "algorithmic_bytes_per_ray_this_kernel": kernel_b_ray})
    else:
        r.update({"bound": "valu", "achieved": valu_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu_tf / FP32_PEAK_TFLOPS,
                  "frac_entering_scene": valu_tf / FP32_PEAK_TFLOPS,
                  "frac_counting_culled_rays": valu_tf_counted / FP32_PEAK_TFLOPS})
    return r


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
    p = make_params(a.width, a.height, a.spp, a.max_depth, bool(a.direct_lighting), bool(a.importance_sampling))
    p.handle = state.params.handle
    accum = torch.zeros((a.height, a.width, 4), dtype=torch.float32, device=dev)
    fb = torch.zeros((a.height, a.width, 4), dtype=torch.uint8, device=dev)
    p.accumulationBuffer = accum.data_ptr()
    p.frameBuffer = fb.data_ptr() if world == 1 else None
    state.params = p
    assert L.pt_set_partition(state.context, rank, world) == 0
    assert L.pt_set_sample_chunks(state.context, a.chunks) == 0
    assert L.pt_set_stream(state.context, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    if a.blocks_per_cu or a.variant >= 0:
        assert L.pt_set_tuning(state.context, a.blocks_per_cu, a.variant if a.variant >= 0 else -1) == 0
    math_mode = _native.MATH_FAST if a.math == "fast" else _native.MATH_IEEE
    assert L.pt_set_math_mode(state.context, math_mode) == 0
    if os.environ.get("ACGPT_NODE_ORDER"):          # experiments library only (tools/): renumbered fp16 nodes, same bits
        assert L.pt_debug_node_order(state.context, int(os.environ["ACGPT_NODE_ORDER"])) == 0, L.pt_last_error(state.context)
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
    rays = shadow = paths = culled = 0
    kernel_ms = []
    last_stats = None
    k = 0
    while k < a.steps:
        n = min(fuse, a.steps - k)
        s = launch(k, n)
        rays += int(s.radiance_rays); shadow += int(s.shadow_rays); paths += int(s.paths); culled += int(s.culled_rays)
        kernel_ms.append(float(s.kernel_ms))
        last_stats = s
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
    tot_rays, tot_shadow, tot_paths, tot_culled = D.sum_over_ranks([rays, shadow, paths, culled], cdev)
    if a.variant >= 0:
        assert int(last_stats.variant) == a.variant, "the kernel variant that ran is not the one requested"
    assert int(last_stats.math_mode) == math_mode, "the math mode that ran is not the one requested"
    # the other math mode, untimed, beside the timed one: two launches of the timed shape, the second one's kernel time (N = 1 only)
    other_math = None
    if world == 1 and not a.no_ieee_leg:
        other = _native.MATH_IEEE if math_mode == _native.MATH_FAST else _native.MATH_FAST
        assert L.pt_set_math_mode(state.context, other) == 0
        keep, keep_fb = accum.clone(), fb.clone()
        launch(0, fuse)
        so = launch(0, fuse)
        accum.copy_(keep); fb.copy_(keep_fb)
        assert L.pt_set_math_mode(state.context, math_mode) == 0
        o_rays = int(so.radiance_rays) + int(so.shadow_rays)
        other_math = {"math": "ieee" if other == _native.MATH_IEEE else "fast", "kernel": (L.pt_variant_kernel(int(so.variant), other) or b"").decode(),
                      "kernel_ms_per_step": float(so.kernel_ms) / fuse, "Mray_per_s_kernel_time": o_rays / (float(so.kernel_ms) * 1e-3) / 1e6,
                      "note": "untimed: kernel time of one %d-step launch in the other math mode, HIP events" % fuse}

    # ---- report ---------------------------------------------------------------------------------------
    if rank == 0:
        all_rays = tot_rays + tot_shadow
        n_launches = max(1, len(kernel_ms))
        vname = L.pt_variant_name(int(last_stats.variant))
        vkern = L.pt_variant_kernel(int(last_stats.variant), int(last_stats.math_mode))
        roof = roofline_block(a, info, world, fuse, kernel_ms, (rays + shadow - culled) / n_launches, (rays + shadow) / n_launches, a.steps / n_launches,
                              vname.decode() if vname else "?", vkern.decode() if vkern else "", L.pt_kernel_source_hash().decode())
        miss = primary_miss_fraction(p, info)
        out = {
            "metric": "Mray/s at %s, %d spp, %d bounces (radiance + shadow rays per second; camera rays that miss the scene box are radiance segments too and are counted — the rays the kernel traverses are Mray_per_s_entering_scene beside `value`)"
                      % ("1080p" if (a.width, a.height) == (WIDTH, HEIGHT) else "%dx%d" % (a.width, a.height), a.spp * a.steps, a.max_depth),
            "value": all_rays / elapsed / 1e6,
            "unit": "Mray/s",
            "Mray_per_s_entering_scene": (all_rays - tot_culled) / elapsed / 1e6,      # the rays the kernel traverses: `value` minus camera rays settled at the scene box
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed * 1e3 / a.steps,
            "ms_per_frame": elapsed * 1e3,
            "spp_per_frame": a.spp * a.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s (%d triangles), %dx%d, %d spp per step x %d steps, maxDepth %d, %s"
                                   % (a.scene, info.n_tris, a.width, a.height, a.spp, a.steps, a.max_depth, toggles_text(a)),
                       "baseline_config": ("the reference program's own start-up workload (PathTracerMain.cpp:43, 58-59, 653-657); a step is one iteration of its frame loop (:700-730)"
                                           if a.config == 0 else "BASELINE.json configs[%d]" % (a.config - 1)),
                       "math": ("fast: the arithmetic of the reference's own build (nvcc --use_fast_math, CMakeLists.txt:267): v_rcp / v_sqrt / v_rsq / v_sin / v_cos in the shading code; traversal and triangle test as in ieee mode"
                                if math_mode == _native.MATH_FAST else "ieee: correctly rounded division / square root and libm sincosf / acosf in the shading code (the CPU oracle's level)"),
                       "steps_per_kernel_launch": fuse, "kernel_launches": n_launches,
                       "parallelism": "pixel tiles 8x4 over %d GPU(s)%s" % (world, ", RCCL reduce of float4 accumulation" if world > 1 else ""),
                       "rays": int(all_rays), "paths": int(tot_paths), "rays_per_path": all_rays / max(1.0, tot_paths),
                       "primary_miss_fraction": miss, "culled_rays": int(tot_culled), "rays_entering_scene": int(all_rays - tot_culled),
                       "Mray_per_s_entering_scene": (all_rays - tot_culled) / elapsed / 1e6,
                       "sample_runs_per_pixel": int(last_stats.sample_chunks),
                       "bvh": {"nodes": info.n_nodes, "max_depth": info.max_depth, "stack_entries": info.stack_entries, "build_ms": info.build_ms},
                       "scene_device_bytes": int(pt.getBvhInfo(state).device_bytes)},      # what the scene holds on the device after the run (one node array + triangle + shading records)
            "roofline": roof,
        }
        if other_math:
            out["other_math_mode"] = other_math
        if world == 1 and not a.no_cpu_baseline:
            base, _ = cpu_baseline(pt, obj, p, a.cpu_spp)
            out["cpu_baseline"] = base
        if a.save_accum:
            np.save(a.save_accum, accum.cpu().numpy())
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
