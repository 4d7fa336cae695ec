// render_megakernel.h — launch interface of the persistent path-trace kernel.
#pragma once
#include <hip/hip_runtime.h>
#include "pt_device.h"

namespace ptd {

constexpr int kRenderThreads = 256;   // 4 waves per workgroup (segment-synchronous variant)
constexpr uint32_t kRenderFoldSlots = 128;  // fold slots per wave (render_megakernel.hip kFoldSlots)
constexpr uint32_t kMaxTimedWaves = 16384;   // stats variants stamp start / queue-empty / end per wave behind the 8 counters
// layout of RenderArgs::counters (64-bit words): [0, 8) ray / path / scheduler counters, [8, 8 + 3 * kMaxTimedWaves) wave stamps,
// then 2056 words of queue progress and phase times (stats variants), then kTailCounters more launch counters
constexpr uint32_t kCulledCounter = 8u + 3u * kMaxTimedWaves + 2056u;   // camera rays ended by the scene-box cull (they are part of counters[0] and [2] too)
#ifdef ACGPT_EXPERIMENTS
constexpr uint32_t kAbortCounter = kCulledCounter + 1u;                 // workgroups of a wavefront kernel (experiments build) that gave up (scheduling error or watchdog): the launch fails
#endif
constexpr uint32_t kWindowMoves = kCulledCounter + 20u;  // wave-level moves of stack entries between the LDS window and global memory (windowed-stack kernels)
#ifdef ACGPT_EXPERIMENTS
constexpr uint32_t kWfDiag = kCulledCounter + 2u;      // 17 words of wavefront-kernel diagnostics (render_wavefront.hip, pt_debug_wf; experiments build)
#endif
constexpr uint32_t kTailCounters = 24u;
constexpr uint32_t kCounterWords = kCulledCounter + kTailCounters;
// Variant indices of the product library (render_megakernel.hip kVariants).  pt_set_scene picks one per scene unless
// pt_set_tuning named one.  fp16 nodes unless the scene has geometry finer than their planes: the mean inflation of a child
// box by the outward fp16 rounding stays below kHalfInflationLimit (measured break-even on clusters of ever smaller triangles,
// profiles/r02_fp16_vs_fp32_nodes.txt: fp16 nodes win by 2 ... 35 % up to 2.6, lose 6 ... 18 % from 4.0) and the area-weighted
// ratio below kHalfAreaLimit.  Then: the five-waves-per-SIMD kernel; above kWindowSceneTris triangles, or when five workgroups'
// lane stacks would not fit a CU's LDS (trees deeper than ~28 levels), its large-scene twin: shade rounds at 24 parked lanes
// instead of 40 and a sliding 16-entry stack window in LDS (profiles/r03_sweep_large_scenes.txt: 18 ... 24 % faster from 82 k to
// 1.31 M triangles, 5 % slower at 20 k).  fp32 nodes otherwise, with triangle rounds at 8 lanes above kLargeSceneTris.
constexpr int kVariantSync = 0, kVariantF32 = 1, kVariantF32Stats = 2, kVariantF32Large = 3, kVariantTrig = 4;      // 4: the cosine sampler's trigonometry in hardware, rest as the math mode says
constexpr int kVariantF16 = 5, kVariantF16Stats = 6, kVariantF16W5 = 7, kVariantLights = 8, kVariantF16W5Deep = 9;
#ifdef ACGPT_EXPERIMENTS
constexpr int kVariantWf = 10, kVariantWfStats = 11;      // experiments build: the workgroup-level wavefront kernel (render_wavefront.hip) and its twin with time stamps
#endif
constexpr int kDefaultVariant = kVariantF16W5;
constexpr uint32_t kLargeSceneTris = 100000;
constexpr uint32_t kWindowSceneTris = 50000;
constexpr float kHalfAreaLimit = 1.5f, kHalfInflationLimit = 3.0f;

struct FastDiv { uint32_t mul, sh1, sh2; };     // n / d = (t + ((n - t) >> sh1)) >> sh2, t = mulhi(n, mul)  (render_megakernel.hip fast_div)

struct RenderArgs {
    DeviceScene scene;
    float4*   accum;        // PathTraceParams::accumulationBuffer
    uint32_t* fb;           // PathTraceParams::frameBuffer (uchar4 packed), may be null
    uint32_t  width, height, spp, maxDepth, frame;
    pt_float3 eye, U, V, W;
    pt_area_light light;
    float     light_area;      // |light.v1 x light.v2| (:1021), fp32, evaluated on the host
    uint32_t  useDL, useIS;
    int       rank, world;
    uint32_t  total_samples;   // queue length: StaticWorkDistribution::numSamples(world) << sub_shift
    uint32_t  shard_size;      // queue shard length (8 shards)
    uint32_t* queue_heads;     // 8 counters, zeroed before the launch
    unsigned long long* counters;   // [8 + 3 * kMaxTimedWaves] radiance rays, shadow rays, paths, pixels, traversal wave-steps, lane-steps, shade rounds, shade lanes
    uint32_t  stack_entries;
    uint32_t* stack_overflow;  // kernels with a capped LDS stack: [wave of the grid][stack_entries - cap][64] deeper entries
    uint32_t  n_lds_nodes;     // nodes staged into LDS (NODE_FMT 2), else 0
    // sample chunks: a pixel's spp samples may be split into 2^chunk_shift consecutive runs, each run
    // owned by its own lane (shortens the per-pixel serial chain when a GPU has few pixels).
    uint32_t  chunk_shift;     // 0 = one lane per pixel (the reference's summation order)
    uint32_t  chunk_spp;       // spp >> chunk_shift
    // frame batches: one launch renders sub-frames frame .. frame + n_frames - 1 (each spp samples per pixel,
    // its own tea<4>(pixel, frame) seeds) and k_finalize folds them into the accumulation buffer one after the
    // other, exactly as n_frames separate launches would.  A work item is (pixel, sub) with
    // sub = frame_in_batch << chunk_shift | chunk, 2^sub_shift subs per pixel (padded with empty items).
    uint32_t  n_frames;        // >= 1
    uint32_t  sub_shift;       // ceil_log2(n_frames) + chunk_shift
    uint32_t  lcg_mul[32];     // seed of chunk k = lcg_mul[k] * seed0 + lcg_add[k]  (2 * k * chunk_spp LCG steps)
    uint32_t  lcg_add[32];
    float4*   frame_sums;      // [pixel slot of this rank's tile order][n_frames] sums of the sub-frames of a batch: k_finalize blends them in order
    float*    wave_scratch;    // [wave of the grid][kFoldSlots << chunk_shift][3] partial sums of runs whose group is still open (chunk_shift > 0)
    uint32_t  grant;           // minimum work items taken per queue atomic (1 = exactly what is needed)
    uint32_t  strip_cols;      // tile-strip columns of StaticWorkDistribution for (width, world)
    FastDiv   div_cols, div_world;
    pt_float3 cull_lo, cull_hi;   // scene bounding box, enlarged: a camera ray that misses it ends its path without a traversal
    // pixel classes (capi.hip row_spans): per image row {x range outside which no ray of a pixel can reach the scene's bounding box,
    // x range inside which every ray of a pixel does}, two uint32 of two 16-bit columns each; null = unknown (every path start tests)
    const uint2* row_spans;
    uint32_t  row_interleave;     // 1 (experiment, pt_debug_queue_order): queue position -> tile-strip row 0, 8, 16, ..., 1, 9, ... so that every
    uint32_t  strip_rows;         // queue shard (one per XCD) holds rows from all over the image instead of a contiguous eighth
    // origin-triangle release (render_megakernel.hip): a ray that leaves a triangle at cos(theta) to its plane cannot be accepted
    // by that triangle's Moeller-Trumbore test once cos(theta) * tmin exceeds what rounding can put between the hit point and the
    // plane: kOriginEps * (largest scene coordinate + length of the segment that produced the hit point).  skip_base = the first term.
    float     skip_base;
};
constexpr float kOriginEps = 16.0f * 1.1920929e-7f;      // 16 * 2^-23

int render_variant_count();
const char* render_variant_name(int variant);
const char* render_variant_kernel(int variant, int math);   // the instantiation as a kernel trace prints it ("" for experiment variants); math: pt_set_math_mode
int render_variant_has_fast_math(int variant);   // 0: the variant exists with IEEE arithmetic only (experiment rows)
int render_variant_threads(int variant);
int render_variant_top_nodes(int variant);      // > 0: the variant stages that many nodes of the tree's top in LDS (experiments)
int render_variant_stack_cap(int variant);      // 0 = the whole stack in LDS
int render_variant_node_format(int variant);   // 0 fp32 two-child; 11 fp16 two-child as centre / half extent (the default); 7 / 8 / 9 fp16 two-child {lo, hi} (min-max / rotated / rotated, flags in the multipliers); experiments: 1/2/4 16-bit grid, 3 four-wide 8-bit, 10 / 12 shared-plane records
hipError_t render_occupancy(int variant, int math, uint32_t stack_entries, uint32_t n_nodes, int* blocks_per_cu);
hipError_t launch_render(int variant, int math, const RenderArgs& args, uint32_t grid_blocks, hipStream_t stream);
hipError_t launch_finalize(const RenderArgs& args, hipStream_t stream);
hipError_t launch_resolve(const float4* accum, uint32_t* fb, uint32_t n, hipStream_t stream);
hipError_t launch_keep_owned(float4* accum, uint32_t width, uint32_t height, int rank, int world, hipStream_t stream);
hipError_t launch_sum_ranks(float4* dst, const float4* const* srcs, int n_srcs, uint32_t n, hipStream_t stream);
hipError_t launch_trace_stream(int fmt, const DeviceScene& sc, uint32_t stack_entries, const float* d_rays, uint32_t n, uint32_t* d_head,
                               float* d_t, uint32_t* d_prim, unsigned long long* d_counters, uint32_t grid_blocks, hipStream_t stream);
hipError_t trace_stream_occupancy(int fmt, uint32_t stack_entries, int* blocks_per_cu);
hipError_t launch_trace_closest(const DeviceScene& sc, uint32_t stack_entries, const float* d_rays, uint32_t n,
                                float* d_t, uint32_t* d_prim, hipStream_t stream);
hipError_t launch_trace_any(const DeviceScene& sc, uint32_t stack_entries, const float* d_rays, uint32_t n,
                            uint8_t* d_hit, hipStream_t stream);

}  // namespace ptd
