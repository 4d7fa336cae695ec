/*
 * acgpt.h — C ABI of the MI355X-native path-trace hot path.
 *
 * This is the drop-in boundary for the render path of the reference's
 * PathTracer_Optix/PathTracerMain.cpp: every entry point below replaces one of
 * the plain functions that file calls over its `PathTracerState&`
 * (PathTracerMain.cpp:71-93).  The reference has no FFI layer of its own; these
 * are the symbols a cgo / JNI / ctypes / C++ binding for that path would bind.
 *
 * Conventions
 *   - every call returns 0 on success, non-zero on failure; the message is
 *     available from pt_last_error() (the reference throws sutil::Exception
 *     from CUDA_CHECK / OPTIX_CHECK, sutil/Exception.h:82-112; the C++ wrapper
 *     in acgpathtracing_amd/host re-throws to keep that convention).
 *   - the context owns every device allocation it makes; host arrays are
 *     borrowed for the duration of the call only.
 *   - a context is single-threaded (the reference is: one host thread, one
 *     stream, launch-then-sync per frame, PathTracerMain.cpp:184-210).
 *   - image row 0 is the BOTTOM row (pathTracerPrograms.cu:782-783 with
 *     sutil/Camera.cpp:34-45: V points up).
 *   - there is NO CPU fallback behind this ABI: without a HIP device
 *     pt_create() fails.
 */
#ifndef ACGPT_H
#define ACGPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_ctx pt_ctx;

typedef struct pt_float3 { float x, y, z; } pt_float3;

/* BSDFType, PathTracer_Optix/TinyObjWrapper.h:27-31 */
enum { PT_BSDF_DIFFUSE = 0, PT_BSDF_METALLIC = 1, PT_BSDF_REFRACTION = 2 };

/* Material, PathTracer_Optix/TinyObjWrapper.h:33-40 (40 bytes, same field order).
 * roughness and metallic are carried but ignored by the shading, exactly as in
 * the reference (pathTracerPrograms.cu:879-880). */
typedef struct pt_material {
    pt_float3 diffuse;
    pt_float3 emission;
    float     roughness;
    float     metallic;
    float     ior;
    int32_t   bsdfType;
} pt_material;

/* AreaLight, PathTracer_Optix/pathTracer.h:77-83 (60 bytes). */
typedef struct pt_area_light {
    pt_float3 corner;
    pt_float3 v1;
    pt_float3 v2;
    pt_float3 normal;
    pt_float3 emission;
} pt_area_light;

/* PathTraceParams, PathTracer_Optix/pathTracer.h:85-108 — field for field,
 * sizeof == 168 on LP64 like the reference's.  accumulationBuffer (float4[w*h],
 * linear radiance, alpha 1) and frameBuffer (uchar4[w*h], sRGB) are DEVICE
 * pointers supplied by the caller, as in the reference (PathTracerMain.cpp:
 * 145-148, 186-188); frameBuffer may also be pinned host memory mapped into
 * the device (the ZERO_COPY mode of sutil/CUDAOutputBuffer.h:45-51).
 * `handle` is the value returned by pt_scene_handle() (the reference stores
 * the OptixTraversableHandle there, PathTracerMain.cpp:159). */
typedef struct pt_params {
    uint32_t      currentFrameIdx;
    float*        accumulationBuffer;
    uint8_t*      frameBuffer;
    uint32_t      width;
    uint32_t      height;
    uint32_t      samplesPerPixel;
    uint32_t      maxDepth;
    pt_float3     cameraEye;
    pt_float3     cameraU;
    pt_float3     cameraV;
    pt_float3     cameraW;
    pt_area_light areaLight;
    uint64_t      handle;
    uint8_t       useDirectLighting;     /* bool */
    uint8_t       useImportanceSampling; /* bool */
} pt_params;

/* Counters of the most recent pt_launch (the reference counts nothing; Mray/s
 * is defined in SURVEY.md §8d as radiance segments + shadow rays per second). */
typedef struct pt_stats {
    uint64_t radiance_rays;   /* closest-hit segments traced                  */
    uint64_t shadow_rays;     /* next-event occlusion rays traced             */
    uint64_t paths;           /* camera paths started (= pixels * spp)        */
    float    kernel_ms;       /* megakernel duration, HIP events on its stream*/
    float    launch_ms;       /* pt_launch entry -> synchronised return       */
    uint32_t pixels;          /* pixels this launch rendered (tile partition) */
    uint32_t grid_blocks;     /* persistent workgroups launched                */
    uint32_t sample_chunks;   /* runs per pixel this launch used (pt_set_sample_chunks) */
    uint32_t variant;         /* render kernel variant that ran (pt_variant_name)      */
    /* scheduler diagnostics (0 for the segment-synchronous variant):          */
    uint64_t trav_wave_steps;   /* BVH loop iterations summed over waves       */
    uint64_t trav_lane_steps;   /* ... times lanes with a ray in flight        */
    uint64_t shade_wave_rounds; /* shade/regenerate rounds summed over waves   */
    uint64_t shade_lane_rounds; /* ... times lanes shaded in them              */
    uint64_t culled_rays;       /* camera rays that end at the scene's bounding box without a traversal: one radiance
                                 * segment that misses (pathTracerPrograms.cu:833-847); counted in radiance_rays and paths too */
    uint32_t math_mode;         /* arithmetic of the shading code this launch ran with (pt_set_math_mode)        */
    uint32_t reserved;
} pt_stats;

/* What the on-device LBVH build produced. */
typedef struct pt_bvh_info {
    uint32_t n_tris;
    uint32_t n_nodes;         /* internal nodes (n_tris - 1, or 1 if n_tris==1) */
    uint32_t max_depth;       /* longest root->leaf path, in internal nodes   */
    uint32_t stack_entries;   /* per-lane traversal stack entries reserved    */
    float    scene_lo[3];
    float    scene_hi[3];
    float    build_ms;        /* morton + sort + hierarchy + refit, device ms */
    uint32_t node_bytes;      /* bytes of the traversal node array            */
    uint32_t tri_bytes;       /* bytes of the leaf triangle array             */
    uint32_t wide_nodes;      /* four-wide tree: nodes                        */
    uint32_t wide_depth;      /* four-wide tree: levels                       */
    uint32_t wide_bytes;      /* four-wide tree: record array (nodes + triangles, 48 B each) */
    float    wide_ms;         /* collapse of the two-child tree, host ms      */
    uint32_t half_node_bytes; /* bytes of the fp16 node array (32 B per node)  */
    float    half_area_ratio; /* summed child-box area with fp16 planes / with fp32 planes (>= 1) */
    float    half_box_inflation; /* mean over the child boxes of their own fp16 / fp32 area (>= 1): large where geometry is finer than the fp16 planes */
    uint64_t device_bytes;    /* bytes of device memory the scene's arrays hold right now (ABI version 4).  A scene keeps ONE node array — the one
                               * its kernel reads: fp16 nodes (32 B per node) or fp32 nodes (64 B) — plus triangle records (48 B) and shading
                               * records (16 B); node_bytes / half_node_bytes above are the SIZES of the two formats, whichever is resident.
                               * The other array, and the experiment formats, are rebuilt on first use (a ray query, pt_set_tuning) and count from then on. */
} pt_bvh_info;

/* ---- lifetime -------------------------------------------------------------
 * pt_create   <- createDeviceContext(), PathTracerMain.cpp:240-258, plus
 *                createModule/ProgramGroups/Pipeline, :400-539 (nothing to JIT:
 *                the code object is built ahead of time for gfx950).
 * pt_destroy  <- CleanAllTheThings(), PathTracerMain.cpp:629-646.            */
int         pt_create(pt_ctx** out, int device_id);
/* A context over n_devices GPUs of one node (1 <= n <= 16; device_ids[0] is "rank 0": the caller's buffers live there).
 * Every call below then acts on the whole group: pt_set_scene builds the BVH on every device (the scene is replicated),
 * pt_launch / pt_launch_frames render with the pixel tiles of sutil/WorkDistribution.h:50-81 split over the devices — one
 * host thread and stream per device — and finish with ONE ncclReduce(SUM) (librccl, loaded on first use) of the ranks'
 * accumulation buffers into params->accumulationBuffer on rank 0, where make_color fills params->frameBuffer; every pixel
 * has exactly one non-zero term, so the buffers equal a one-device launch with the same sample-run setting bit for bit.
 * pt_get_stats sums the ranks' counters (kernel_ms: the slowest rank); queries and the memory helpers act on rank 0.
 * pt_set_partition is refused.  With ACGPT_REHEARSE_SAME_GPU=1 in the environment the same device id may be listed
 * more than once: every rank then shares that GPU and a sum kernel stands in for RCCL (a rehearsal for one-GPU boxes).   */
int         pt_create_multi(pt_ctx** out, const int* device_ids, int n_devices);
int         pt_device_count(pt_ctx* ctx);   /* devices behind the context: 1 for pt_create */
void        pt_destroy(pt_ctx* ctx);
const char* pt_last_error(pt_ctx* ctx);   /* ctx may be NULL: last global error */

/* ---- scene ------------------------------------------------------------------
 * pt_set_scene <- buildTheAccelarationStructure(), PathTracerMain.cpp:260-398
 *                 + createShaderBindingTable(), :544-627.
 * verts_xyzw: n_verts * 4 floats (w ignored; stride 16 B as :314);
 * idx: n_tris * 3 vertex indices; mat_ids: one per triangle (the reference's
 * sbtIndexOffsetBuffer, :325-328); mats: n_mats records.
 * A material id >= n_mats is an error (the reference would index past its SBT).
 * Builds the LBVH on the device; replaces any previous scene.               */
int      pt_set_scene(pt_ctx* ctx,
                      const float* verts_xyzw, size_t n_verts,
                      const uint32_t* idx, size_t n_tris,
                      const uint32_t* mat_ids,
                      const pt_material* mats, size_t n_mats);
/* Hierarchy builder used by the NEXT pt_set_scene: all start from the same on-device Morton radix
 * sort; 0 = Karras radix tree (classic LBVH), 1 = PLOC (locally-ordered clustering: fewer node visits
 * per ray), 2 = PLOC followed by an insertion-based optimisation of the tree — the default: for scenes
 * up to 16 384 triangles on the host, one node at a time (Bittner et al. 2013; 12 ms for the 1 264
 * triangles of the Cornell scenes, 2.5 % fewer visits), for larger ones on the device, all nodes at
 * once (parallel reinsertion, Meister & Bittner 2018; +73 ms at 1.31 M triangles, +0.6 s at 10.5 M,
 * render 6 % / 13 % faster); scenes above 50 000 triangles get their nodes in depth-first order.
 * Results of every query and every image bit are identical, only speed differs.                  */
int      pt_set_build_mode(pt_ctx* ctx, int mode);
uint64_t pt_scene_handle(pt_ctx* ctx);
int      pt_get_bvh_info(pt_ctx* ctx, pt_bvh_info* out);

/* ---- the hot call -------------------------------------------------------------
 * pt_launch <- LaunchCurrentFrame(), PathTracerMain.cpp:184-210: consumes a
 * PathTraceParams by value, runs the megakernel over width x height pixels and
 * returns synchronised (CUDA_SYNC_CHECK, :209).  Progressive accumulation
 * follows pathTracerPrograms.cu:803-811 (running mean over currentFrameIdx).
 * maxDepth outside [1,28] and samplesPerPixel == 0 are errors
 * (PathTracerMain.cpp:42, 122-128; the do{}while(--i) at
 * pathTracerPrograms.cu:727,780 requires spp >= 1).                           */
int pt_launch(pt_ctx* ctx, const pt_params* params);

/* A batch of sub-frames in ONE kernel launch: renders frames params->currentFrameIdx ..
 * currentFrameIdx + n_frames - 1 (each samplesPerPixel samples per pixel with its own
 * tea<4>(pixel, frame) seeds) and folds them into the accumulation buffer in frame order.
 * The buffers end up bit-identical to n_frames consecutive pt_launch calls (the loop
 * of PathTracerMain.cpp:700-730 without its per-launch synchronisation); what is saved is
 * the ramp-up and drain of n_frames - 1 launches.  n_frames in [1, 64].              */
int pt_launch_frames(pt_ctx* ctx, const pt_params* params, uint32_t n_frames);

/* Upper bound on the per-launch scratch of a frame batch (one float4 per pixel and sub-frame, held until the batch is
 * blended into the accumulation buffer): pt_launch_frames runs a batch whose sums would not fit as several kernel
 * launches — same bits.  Default 1 GiB (32 sub-frames at 1920x1080); at least 1 MiB.                                   */
int pt_set_scratch_limit(pt_ctx* ctx, size_t bytes);

/* Multi-GPU pixel partition: this context renders only the pixels that
 * sutil/WorkDistribution.h:50-81 assigns to `rank` of `world` (interleaved 8x4
 * tiles, rotated per strip row); other pixels of the buffers are left
 * untouched.  world == 1 (the default) renders everything.                   */
int pt_set_partition(pt_ctx* ctx, int rank, int world);

/* After a multi-GPU reduce the root holds the summed accumulation but no colours: this applies
 * make_color (cuda/helpers.h:58-63, as pathTracerPrograms.cu:814 does per pixel) to n_pixels of
 * a float4 DEVICE array and writes uchar4 into frameBuffer (device or mapped host).          */
int pt_resolve_framebuffer(pt_ctx* ctx, const float* accumulation_rgba, uint8_t* framebuffer_rgba, size_t n_pixels);

/* Light mode (strictly opt-in, the default 0 is the reference bit for bit).
 *   0  the reference's estimator: the rectangle of params->areaLight (hard-coded at PathTracerMain.cpp:154-158), light
 *      samples and BSDF-sampled light hits both counted (pathTracerPrograms.cu:992-1026), the emitter quirks of a8/a9;
 *   1  the area light is the scene's own emissive triangles (materials with Ke != 0; params->areaLight is ignored), light
 *      sampling by area, combined with BSDF sampling by the power heuristic; a seen light contributes Ke, uniform
 *      hemisphere sampling carries its 2 cos weight: direct lighting on / off and importance sampling on / off converge to
 *      the same image.  Same random draws per segment as mode 0.  One kernel variant serves it (pt_variant_name "LIGHTS").  */
int pt_set_light_mode(pt_ctx* ctx, int mode);

/* Arithmetic of the shading code (closest-hit, samplers, light sample, roulette, camera-ray set-up).
 *   PT_MATH_FAST (default): the arithmetic of the reference's own build.  /root/reference/CMakeLists.txt:267 compiles
 *      pathTracerPrograms.cu with nvcc --use_fast_math (-prec-div=false -prec-sqrt=false, sinf -> __sinf, cosf -> __cosf):
 *      a / b is a * rcp(b), sqrtf / 1 / sqrtf are the approximate instructions, sin / cos of 2 pi u the hardware ones.  Here:
 *      v_rcp_f32, v_sqrt_f32, v_rsq_f32, v_sin_f32, v_cos_f32 (1 ulp each) and sqrt(1 - z) for sin(acos(sqrt(z))).
 *   PT_MATH_IEEE: correctly rounded division and square root, the C library's sincosf / acosf — the level the CPU oracle
 *      (oracle/oracle_pt.cpp) is written at; the mode in which most pixels of an image equal the oracle's bit for bit.
 * BVH traversal and the triangle test are the same code in both modes (hit triangle and distance of a given ray are bit-exact
 * either way); the modes differ in the last bits of shading values, i.e. by less than the parity tolerance (image MSE < 1e-6
 * against the oracle in both; tests/test_gpu_parity.py).  Every kernel variant of the product library exists in both modes. */
#define PT_MATH_IEEE 0
#define PT_MATH_FAST 1
int pt_set_math_mode(pt_ctx* ctx, int mode);

/* Sample chunks (1, 2, 4, 8, 16, 32; 0 = automatic, the default: 8 runs per pixel, 16 when this rank
 * holds fewer than 2^20 pixels, reduced until every run keeps at least 4 samples).  With c > 1 a pixel's samplesPerPixel samples are cut into
 * c consecutive runs, each owned by its own lane with the PRNG skipped ahead to where the run
 * starts, and the runs' partial sums are added in run order.  Same samples, same paths; only the
 * association of the fp32 sum changes ((s1+..+sk) + (sk+1+..) instead of one left-to-right chain), so
 * images differ from c = 1 in the last bits.  It shortens the per-pixel serial chain, which is what
 * bounds a launch when a GPU holds few pixels (multi-GPU tiles), and keeps neighbouring lanes on
 * similar rays.  samplesPerPixel must divide by c.
 * WHICH SETTING REPRODUCES THE REFERENCE'S SUMMATION ORDER: c = 1, and only c = 1 — one left-to-right fp32 chain per pixel, as
 * pathTracerPrograms.cu:727-780 adds its samples.  A drop-in caller who wants the reference order's last bits asks for it
 * (pt_set_sample_chunks(ctx, 1); __graft_entry__.smoke() and most parity tests do).  bench.py does NOT: it times the
 * automatic setting (8 or 16 runs per pixel; the oracle is called with the same association wherever bits are compared),
 * which is 20-30 % faster on the headline configuration and differs from c = 1 by <= 1e-4 relative on any channel. */
int pt_set_sample_chunks(pt_ctx* ctx, int chunks);

/* Launch tuning: persistent workgroups per CU (0 = from the occupancy query) and the render
 * kernel variant: -1 = chosen per scene (the default: fp16 nodes unless the scene has geometry finer than their planes —
 * pt_bvh_info.half_box_inflation above 3 —, fp32 nodes then), 0 = segment-synchronous, n >= 1 = persistent traversal
 * with deferred shading, see csrc/render_megakernel.hip.  Every variant produces the same image bits.   */
int pt_set_tuning(pt_ctx* ctx, int blocks_per_cu, int variant);
/* Human-readable description of a kernel variant, NULL past the last one.  Names starting with "DIAG"
 * are timing experiments (some deliberately compute different bits) and are never selected by default; "FAST-MATH" and
 * "LIGHTS" (the kernel of pt_set_light_mode(1)) are opt-in and compute other bits than the reference's estimator. */
const char* pt_variant_name(int variant);
/* The variant's kernel in a math mode as a kernel trace prints it ("k_render_pw<40, 16, 11, 256, 5, false, 0, 6, 2, false, 0, 0, 1>":
 * the last argument is the math mode), and a hash of the kernel sources this library was built from: what bench.py checks a
 * committed profile against before quoting it. */
const char* pt_variant_kernel(int variant, int math_mode);
const char* pt_kernel_source_hash(void);

/* Stream the launches are enqueued on (a hipStream_t, e.g. torch's current
 * stream); NULL restores the context's own stream (PathTracerMain.cpp:161). */
int pt_set_stream(pt_ctx* ctx, void* hip_stream);
int pt_get_stats(pt_ctx* ctx, pt_stats* out);

/* ---- queries used by the parity tests ------------------------------------------
 * rays: n records of 8 floats (origin xyz, direction xyz, tmin, tmax), HOST.
 * closest: t_out[i] = hit distance or -1, prim_out[i] = triangle index (in the
 * caller's index-buffer order) or 0xFFFFFFFF; interval is open (tmin, tmax),
 * two-sided, ties -> lowest triangle index.  any: hit_out[i] = 1 if any
 * triangle is hit inside the interval (traceOcclusion, pathTracerPrograms.cu:
 * 651-684).                                                                    */
int pt_trace_closest(pt_ctx* ctx, const float* rays, size_t n, float* t_out, uint32_t* prim_out);
int pt_trace_any(pt_ctx* ctx, const float* rays, size_t n, uint8_t* hit_out);
/* ---- device memory helpers for bindings that have no HIP runtime of their own
 * (the reference app calls cudaMalloc/cudaMemcpy directly, :145-148).         */
int pt_device_malloc(pt_ctx* ctx, void** out, size_t bytes);
int pt_device_free(pt_ctx* ctx, void* ptr);
int pt_device_memset(pt_ctx* ctx, void* ptr, int value, size_t bytes);
int pt_copy_to_host(pt_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);
int pt_copy_to_device(pt_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int pt_host_malloc_mapped(pt_ctx* ctx, void** host_out, void** device_out, size_t bytes);
int pt_host_free_mapped(pt_ctx* ctx, void* host_ptr);

/* Version of this ABI. */
uint32_t pt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ACGPT_H */
