/*
 * acgpt_test.h — test hooks and diagnostics of libacgpt_hip.so.
 *
 * NOT part of the drop-in boundary (include/acgpt.h): nothing here stands in for a function of the reference's
 * PathTracerMain.cpp.  The parity tests and the tools under tools/ use these to hold the device functions against golden
 * vectors and to look inside the scheduler; a binding of the render path never needs them.
 */
#ifndef ACGPT_TEST_H
#define ACGPT_TEST_H

#include "acgpt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: pure-traversal throughput.  Streams n HOST rays (same 8-float records; tmax < 0 marks an
 * any-hit ray of length |tmax|) through a persistent kernel that contains nothing but the BVH loop, `repeats`
 * times, and reports the fastest kernel time.  node_format 0: two-child fp32 tree; 1: four-wide 8-bit tree;
 * 2: two-child tree with the slab test as one fma per plane (the fp32 render kernels' form); 3: fp16 {lo, hi} nodes (rounds 2-3's default);
 * 4: fp16 {centre, half extent} nodes with a scale per axis — the array and the box test the default render kernels use.
 * Results: closest rays as pt_trace_closest; any-hit rays give t_out = prim_out = 1 when occluded, 0
 * otherwise.  counters_out (may be NULL): 5 values of the last repeat — loop iterations summed over waves,
 * node visits summed over lanes, triangle tests summed over lanes, iterations with a node visit, iterations
 * with a triangle round.  Not used by the render path.                                                   */
int pt_bench_traversal(pt_ctx* ctx, const float* rays, size_t n, int repeats, int node_format, float* t_out, uint32_t* prim_out, float* ms_out,
                       uint64_t* counters_out);
/* Test hook: evaluates the device functions the render kernel is built from on HOST inputs (n elements), so that
 * the GPU implementations can be held against golden vectors of the reference's own code directly.
 *   op 0  tea<4> (cuda/random.h:31-46)            in uint32[n][2]                 out uint32[n]
 *   op 1  lcg / rnd stream (cuda/random.h:49-67)  in {seed, count}, n = 1         out uint32 states[count], float values[count]
 *   op 2  make_color (cuda/helpers.h:35-62)       in float[n][3]                  out uchar4[n]
 *   op 3..8 normalize, reflect, faceforward, lerp, cross, a / s (sutil/vec_math.h)  in float[n][10] = a, b, c, s   out float[n][3]
 *   op 9  refract (cuda/helpers.h:107-137)        in float[n][7] = i, n, ior      out float[n][4] = r, ok (uint32)
 *   op 10 StaticWorkDistribution::getSamplePixel (sutil/WorkDistribution.h:59-81)  in int32[n][4] = world, width, rank, sample   out int32[n][2]
 *   op 11 sinf(x), cosf(x) and the pair sincosf(x) gives (the samplers use the latter)   in float[n]   out float[n][4]
 * the OptiX-free helpers of pathTracerPrograms.cu, golden vectors from the reference's own text (oracle/_ref):
 *   op 12 OrthonormalBasis(n).inverse_transform(p) (:54-85)        in float[n][6] = n, p            out float[n][3]
 *   op 13 safeDivide(float3, float) (:265-284)                      in float[n][4] = a, b            out float[n][3]
 *   op 14 cosine_sample_hemisphere (:341-353)                       in float[n][2] = eta1, eta2      out float[n][3]
 *   op 15 uniform_sample_hemisphere (:368-380)                      in float[n][2] = u1, u2          out float[n][3]
 *   op 16 sampleGGX (:455-476)                                      in float[n][6] = u1, u2, roughness, N   out float[n][3]
 *   op 17 fresnelSchlickConductor (:494-510)                        in float[n][7] = cosTheta, eta, k       out float[n][3]
 *   op 18 FrDielectric (:534-559)                                   in float[n][3] = cosThetaI, etaI, etaT  out float[n][1]
 * and the default kernel's own ray / box test (no reference counterpart: OptiX traverses there), end to end:
 *   op 19 fp16 slab test: box -> outward fp16 planes, ray -> per-axis multiplier / addend with the rotate flags in the
 *         multiplier's low bits, entry / exit distance          in float[n][17] = ray o, d, box lo, hi, scene centre, inv_scale
 *         (the builder's: scene half extent / 1023; any positive value), tmax    out uint32[n][3] = accepted, entry t (float bits), exit t (float bits).  Must accept every
 *         ray that meets the box shrunk by the builder's pad (tests/test_gpu_golden.py).
 * the arithmetic of PT_MATH_FAST (pt_set_math_mode; no reference vectors exist for it: nvcc's approximate instructions are not
 * reproducible here — the ops are held against their IEEE twins by error bounds):
 *   op 30 primitives      in float[n][2] = a, b     out float[n][4] = a * rcp(b), sqrt(|a|), x and y of normalize((a, b, 1))
 *   op 31 sin / cos of 2 pi u                       in float[n]     out float[n][4] = v_sin_f32(u), v_cos_f32(u), sinf(2 pi u), cosf(2 pi u)
 *   op 32..38 = ops 12..18 at that level, same records (op 33: the roulette's throughput scaling, one reciprocal for the three quotients)
 *   op 40 = op 19 (same records, same outputs) through the fp16 centre / half-extent form of the nodes (NODE_FMT 11: pack_centre_half, slab_hc)
 *   op 41 = op 40 with a scale per axis, as the builder uses it: in = ray o xyz, d xyz, box lo xyz, hi xyz, scene centre xyz, inv_scale xyz, tmax (19 floats)
 *   op 39 shared-plane slab test (NODE_FMT 10, pt_device.h SSpace): in float[n][20] = ray o xyz, d xyz, box lo xyz, hi xyz, root planes L xyz, H xyz,
 *         inv_scale, tmax; the box as child 0 and as child 1 of a node whose other child is the root box.  out uint[n][3] = accepted as child 0,
 *         accepted as child 1, root accepted | sibling (the root box) accepted << 1 | << 2 */
int pt_selftest(pt_ctx* ctx, int op, const void* in, size_t n, void* out);
/* Diagnostic: after a launch of a "+ scheduler stats" kernel variant, three 100 MHz stamps per wave (start,
 * first time it found the work queue empty, end; 0 = wave did not run), HOST output of 3 * max_waves values. */
int pt_debug_wave_times(pt_ctx* ctx, uint64_t* out, size_t max_waves);
/* ... and, per work-queue shard (8), the stamp of the first grant past each 1/256 of the shard: HOST output of
 * 8 * 256 values (0 = never reached by a stamped grant), followed by ONE value: the time all waves together spent
 * in the shade / regenerate phase, in 10 ns units and its split into queue refill / finished runs / camera-path start (2056 values in all: 2048 + 1 + 3, rest unused). */
int pt_debug_queue_progress(pt_ctx* ctx, uint64_t* out);
/* Order of the work queue over its eight shards (one per XCD).  0: each shard is a contiguous eighth of the 8x4-tile order
 * (round 2); 1 (default): tile-strip rows are dealt round robin over the shards (what sutil/WorkDistribution.h:60-81 does
 * across GPUs); 2: single tiles round robin; 3: one queue in image order.  Same image bits and counters in every mode. */
int pt_debug_queue_order(pt_ctx* ctx, int mode);
/* Pixel classes on (default) / off.  On: per image row, the host hands the kernel the columns outside which no ray of a pixel can
 * reach the scene's bounding box (those pixels' samples are booked as misses without being started) and the columns inside
 * which every ray does (their path starts skip the cull test).  Same image bits and the same ray / path counters either way;
 * pt_stats.culled_rays differs (it counts what was settled without a traversal). */
int pt_debug_pixel_classes(pt_ctx* ctx, int on);
/* The host computation behind them, callable without a context or a GPU: for the camera and image size of `params` and the box
 * [box_lo, box_hi], out[2 * y] = outer columns lo | hi << 16 and out[2 * y + 1] = inner columns lo | hi << 16 of image row y
 * (2 * height words).  0 = computed, 1 = no classes for this view (box not entirely in front of the eye, degenerate frame). */
int pt_debug_row_spans(const pt_params* params, const float* box_lo, const float* box_hi, uint32_t* out);
/* Diagnostic, one value of the last launch: a windowed-stack kernel's moves of stack entries between the LDS window and global
 * memory (wave-level events; 0 for the kernels that keep the whole stack in LDS). */
int pt_debug_window_moves(pt_ctx* ctx, uint64_t* out);
#ifdef ACGPT_EXPERIMENTS
/* Experiments library only: walk a renumbered copy of the fp16 nodes (0: the build's order; 1: the two children of a node in one 64-byte
 * line; 2: depth first).  Same bits; an A/B of the memory system on large scenes (profiles/r04_ab_node_order.txt). */
int pt_debug_node_order(pt_ctx* ctx, int mode);
/* Experiments library only (libacgpt_hip_exp.so, -DACGPT_EXPERIMENTS; never the product): 17 values after a launch of a wavefront
 * kernel variant (render_wavefront.hip), summed over the waves of the grid, times in 10 ns ticks: trace waves {total, idle},
 * shade waves {total, idle, deal time / rounds / records, hit-shading time / rounds / records, accounting time / rounds /
 * records}, trace waves {exchange time / exchanges / records taken in, loop trips}. */
int pt_debug_wf(pt_ctx* ctx, uint64_t* out);
#endif
/* Sorted (morton, triangle) pairs of the last build, HOST outputs of n_tris. */
int pt_read_morton(pt_ctx* ctx, uint32_t* codes_sorted, uint32_t* prims_sorted);

#ifdef __cplusplus
}
#endif
#endif /* ACGPT_TEST_H */
