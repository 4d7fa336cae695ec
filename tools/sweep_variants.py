#!/usr/bin/env python3
"""Run every render-kernel variant on the bench workload in ONE process (interleaved rounds) and
print ms per launch, Mray/s, scheduler efficiencies and whether the image bits match variant 0."""
import argparse
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import acgpathtracing_amd as pt  # noqa: E402
from acgpathtracing_amd import _native  # noqa: E402
from scene_utils import make_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="cornell_box_diffuse.obj")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", default="all")
    ap.add_argument("--blocks-per-cu", default="0")
    ap.add_argument("--no-dl", action="store_true")
    ap.add_argument("--chunks", default="1", help="comma list of sample-chunk counts to sweep")
    ap.add_argument("--partition", default="0,1", help="rank,world pixel-tile partition")
    ap.add_argument("--build-mode", type=int, default=None, help="0 Karras LBVH, 1 PLOC (library default)")
    ap.add_argument("--fuse", type=int, default=1, help="sub-frames per kernel launch (pt_launch_frames)")
    ap.add_argument("--pixel-classes", type=int, default=1, help="0: every path start tests its camera ray against the scene box (pt_debug_pixel_classes)")
    ap.add_argument("--math", type=int, default=1, help="pt_set_math_mode: 1 fast (the library default), 0 ieee; experiment rows without a fast twin run ieee either way")
    ap.add_argument("--queue-order", type=int, default=1, help="1: tile-strip rows interleaved over the queue shards (pt_debug_queue_order)")
    a = ap.parse_args()
    L = _native.hip()
    path = a.scene if os.path.isabs(a.scene) else os.path.join(pt.SCENES, a.scene)
    state, obj = pt.setup(path, width=a.width, height=a.height, max_depth=a.max_depth,
                          direct_lighting=not a.no_dl, importance_sampling=True, spp=a.spp, build_mode=a.build_mode)
    p = make_params(a.width, a.height, a.spp, a.max_depth, not a.no_dl, True)
    keep_a, keep_h = state.params.accumulationBuffer, state.params.handle
    C.memmove(C.byref(state.params), C.byref(p), C.sizeof(p))
    state.params.accumulationBuffer, state.params.handle = keep_a, keep_h
    rank, world = [int(x) for x in a.partition.split(",")]
    assert L.pt_set_partition(state.context, rank, world) == 0
    assert L.pt_set_math_mode(state.context, a.math) == 0
    assert L.pt_debug_queue_order(state.context, a.queue_order) == 0
    assert L.pt_debug_pixel_classes(state.context, a.pixel_classes) == 0
    info = pt.getBvhInfo(state)
    print("scene %s: %d tris, depth %d, stack %d, build %.2f ms" % (a.scene, info.n_tris, info.max_depth, info.stack_entries, info.build_ms))
    variants = list(range(64)) if a.variants == "all" else [int(v) for v in a.variants.split(",")]
    bpcs = [int(v) for v in a.blocks_per_cu.split(",")]
    results = {}
    ref_hash = None
    chunk_list = [int(x) for x in a.chunks.split(",")]
    for r in range(a.rounds):
      for ch in chunk_list:
        assert L.pt_set_sample_chunks(state.context, ch) == 0
        for bpc in bpcs:
            for v in variants:
                if L.pt_variant_name(v) is None:
                    continue
                if L.pt_set_tuning(state.context, bpc, v) != 0:
                    if r == 0:
                        print("variant %d skipped: %s" % (v, L.pt_last_error(state.context).decode()))
                    continue
                state.params.currentFrameIdx = 0
                pt.LaunchCurrentFrame(None, state, a.fuse)
                s = pt.getStats(state)
                acc = pt.readAccumulation(state) if r == 0 else None
                h = hashlib.sha1(acc.tobytes()).hexdigest()[:12] if acc is not None else None
                if ref_hash is None:
                    ref_hash = h
                key = (v, bpc, ch)
                e = results.setdefault(key, {"ms": [], "launch": [], "hash": h, "stats": s})
                e["ms"].append(s.kernel_ms)
                e["launch"].append(s.launch_ms)
    print("%-4s %-4s %-3s %-6s %9s %9s %9s %8s %8s %8s  %s" % ("var", "bpc", "ch", "grid", "ms(min)", "ms(med)", "Mray/s", "travEff", "shadeEff", "steps/ray", "bits==first"))
    for (v, bpc, ch), e in sorted(results.items()):
        s = e["stats"]
        rays = s.radiance_rays + s.shadow_rays
        ms = sorted(e["ms"])
        te = s.trav_lane_steps / (64.0 * s.trav_wave_steps) if s.trav_wave_steps else float("nan")
        se = s.shade_lane_rounds / (64.0 * s.shade_wave_rounds) if s.shade_wave_rounds else float("nan")
        spr = s.trav_lane_steps / rays if s.trav_wave_steps else float("nan")
        if s.trav_wave_steps:
            pass
        print("      pt_launch wall (min) %.3f ms vs render kernel %.3f ms" % (min(e["launch"]), ms[0]))
        if s.trav_wave_steps:
            print("      wave-steps %.4g  shade rounds %.4g  rays %.4g  paths %.4g  pixels %d" % (s.trav_wave_steps, s.shade_wave_rounds, rays, s.paths, s.pixels))
        print("%-4d %-4d %-3d %-6d %9.3f %9.3f %9.1f %8.3f %8.3f %8.2f  %-5s  %s" %
              (v, bpc, ch, s.grid_blocks, ms[0], ms[len(ms) // 2], rays / ms[0] / 1e3, te, se, spr, e["hash"] == ref_hash, ("[fast] " if s.math_mode else "[ieee] ") + L.pt_variant_name(v).decode()))
    pt.CleanAllTheThings(state)


if __name__ == "__main__":
    main()
