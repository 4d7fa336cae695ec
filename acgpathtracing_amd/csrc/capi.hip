// capi.hip — implementation of include/acgpt.h on the HIP runtime.
// Each export names the reference function it stands in for (see acgpt.h).  There is no
// CPU path in this library: every entry point fails if no HIP device is usable.
#include <hip/hip_runtime.h>
// RCCL: types only — librccl is loaded with dlopen by pt_create_multi, a single-GPU caller never touches it, and a box without
// the RCCL headers still builds the library (the handful of types and enumerators used below, with rccl.h's values)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/acgpt.h"
#include "../../include/acgpt_test.h"
#include "lbvh_build.h"
#include "pt_device.h"
#include "render_megakernel.h"
#include "selftest.h"

#define PT_API extern "C" __attribute__((visibility("default")))

static_assert(sizeof(pt_params) == 168, "pt_params must mirror PathTraceParams (168 bytes)");
static_assert(sizeof(pt_material) == 40, "pt_material must mirror Material (40 bytes)");
static_assert(sizeof(pt_area_light) == 60, "pt_area_light must mirror AreaLight (60 bytes)");
static_assert(sizeof(pt_stats) == 96 && sizeof(pt_bvh_info) == 88, "ABI version 4: a change of these layouts bumps pt_abi_version");

struct pt_multi;

struct pt_ctx {
    int device = 0;
    int n_cus = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // bracket the render kernel alone (k_finalize sits outside)
    ptd::LbvhResult bvh;
    ptd::DevMaterial* d_mats = nullptr;       // pt_material repacked into two aligned 16-byte halves (pt_device.h)
    uint32_t n_mats = 0;
    float4* d_lights = nullptr;               // emissive triangles of the scene (light mode 1), 5 float4 each
    uint32_t n_lights = 0;
    float light_area = 0.0f;
    int light_mode = 0;                       // 0 the reference's estimator, 1 scene lights + MIS (pt_set_light_mode)
    int math_mode = PT_MATH_FAST;             // arithmetic of the shading code (pt_set_math_mode): the reference's own build uses nvcc --use_fast_math
    uint32_t stack_entries = 8;
    int blocks_per_cu = 0;        // from the occupancy query for the current stack size
    int tune_blocks_per_cu = 0;   // user override
    int build_mode = 2;           // 0 Karras LBVH, 1 PLOC over the Morton order, 2 PLOC + insertion-based optimisation of small trees (the default)
    int variant = ptd::kDefaultVariant;   // render kernel variant (render_megakernel.hip)
    bool variant_auto = true;             // until pt_set_tuning picks one: chosen per scene size in pt_set_scene
    uint32_t* d_queue = nullptr;              // 8 shard heads
    unsigned long long* d_counters = nullptr; // 8 counters
    int rank = 0, world = 1;
    int chunks = 0;                           // sample chunks per pixel: 0 = automatic, else 1/2/4/8/16
    float4* d_frame_sums = nullptr; size_t frame_sums_bytes = 0;   // [pixel][sub-frame] of a frame batch
    float* d_wave_scratch = nullptr; size_t wave_scratch_bytes = 0;    // fold slots of every wave of the grid
    uint32_t* d_stack_ovf = nullptr; size_t stack_ovf_bytes = 0;       // stack entries beyond a kernel's LDS cap
    size_t scratch_limit = (size_t)1 << 30;                          // a frame batch is cut into launches whose frame sums fit
    // division constants of the tile order, valid for (div_width, div_world_n): built and verified once per image width
    uint32_t div_width = 0; int div_world_n = 0; ptd::FastDiv div_cols = {0, 0, 0}, div_world = {0, 0, 0};
    uint2* d_row_spans = nullptr; size_t row_spans_rows = 0;         // pixel classes per image row (row_spans below)
    std::vector<float> spans_key;                                    // what the spans on the device were computed from
    int pixel_classes = 1;                                           // 0: off (pt_debug_pixel_classes)
    int queue_order = 1;                      // tile-strip rows dealt round robin over the queue shards (render_common.h queue_slot; pt_debug_queue_order)
    pt_multi* multi = nullptr;                // pt_create_multi: this context is rank 0 of a group (below)
    pt_stats stats;
    uint64_t scene_serial = 0;
    std::string err;
};

static std::mutex g_err_mu;
static std::string g_err;

static int fail(pt_ctx* c, const std::string& m)
{
    if (c) c->err = m;
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = m;
    return 1;
}
#define CK(c, x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail((c), std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// ---- ROCTx ranges (tracing): scene build, launch, finalize, reduce show up labelled in a
// `rocprofv3 --marker-trace` of any caller.  The marker library is looked up once at run time; without it the ranges are no-ops.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        void* h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
static Roctx& roctx() { static Roctx r; return r; }
struct Range {
    bool on;
    explicit Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
};
}  // namespace

// ---- multi-GPU group (pt_create_multi): rank 0 is the context the caller holds; it owns the others -----------------
// One context, stream and host thread per device; the tile partition of sutil/WorkDistribution.h:50-81 per rank; each rank
// accumulates its own pixels in a private full-size float4 buffer that is zero elsewhere; ONE ncclReduce(SUM) per launch
// brings them into the caller's accumulation buffer on rank 0 (every pixel has exactly one non-zero term, so the sum is
// that term bit for bit) and rank 0 applies make_color.  Replaces the dormant multi-GPU branch of the reference
// (sutil/WorkDistribution.h, sutil/CUDAOutputBuffer.h CUDA_P2P).
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
struct pt_multi {
    std::vector<pt_ctx*> ranks;          // ranks[0] = the context the caller holds
    std::vector<float4*> accum;          // private accumulation buffer of each rank (its own pixels, zero elsewhere)
    uint32_t accum_w = 0, accum_h = 0;   // image shape the private buffers were allocated for
    bool rehearsal = false;              // all ranks on ONE device (one-GPU box): a sum kernel stands in for RCCL
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    const void* cont_accum = nullptr;    // the caller's buffer and the frame index a straight continuation would pass next
    uint32_t cont_frame = 0, cont_w = 0, cont_h = 0;
    float reduce_ms = 0.0f;
    pt_stats group_stats;
};

PT_API uint32_t pt_abi_version(void) { return 4u; }

PT_API const char* pt_last_error(pt_ctx* ctx)
{
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_err_mu);
    return g_err.c_str();
}

static int create_one(pt_ctx** out, int device_id)
{
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, "pt_create: no HIP device (this library has no CPU path)");
    if (device_id < 0 || device_id >= n) return fail(nullptr, "pt_create: device id out of range");
    CK(nullptr, hipSetDevice(device_id));
    hipDeviceProp_t prop;
    CK(nullptr, hipGetDeviceProperties(&prop, device_id));
    pt_ctx* c = new pt_ctx();
    c->device = device_id;
    c->n_cus = prop.multiProcessorCount;
    memset(&c->stats, 0, sizeof(c->stats));
    if (hipStreamCreate(&c->own_stream) != hipSuccess || hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipMalloc((void**)&c->d_queue, 8 * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&c->d_counters, (size_t)ptd::kCounterWords * sizeof(unsigned long long)) != hipSuccess) {
        delete c;
        return fail(nullptr, "pt_create: device resource allocation failed");
    }
    c->stream = c->own_stream;
    *out = c;
    return 0;
}

PT_API int pt_create(pt_ctx** out, int device_id)
{
    if (!out) return fail(nullptr, "pt_create: out is null");
    return create_one(out, device_id);
}

static void destroy_one(pt_ctx* c);

static bool load_rccl(RcclApi& r, std::string& err)
{
    r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))dlsym(r.lib, "ncclReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.Reduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) { err = "librccl lacks a collective entry point"; return false; }
    return true;
}

PT_API int pt_create_multi(pt_ctx** out, const int* device_ids, int n_devices)
{
    if (!out) return fail(nullptr, "pt_create_multi: out is null");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 16) return fail(nullptr, "pt_create_multi: need 1 to 16 device ids");
    bool dup = false;
    for (int i = 0; i < n_devices; i++) for (int j = 0; j < i; j++) dup = dup || device_ids[i] == device_ids[j];
    const char* reh = getenv("ACGPT_REHEARSE_SAME_GPU");
    if (dup && !(reh && reh[0] == '1'))
        return fail(nullptr, "pt_create_multi: a device id appears twice (allowed only as a rehearsal on a one-GPU box: ACGPT_REHEARSE_SAME_GPU=1)");
    pt_multi* m = new pt_multi();
    m->rehearsal = dup;
    for (int i = 0; i < n_devices; i++) {
        pt_ctx* c = nullptr;
        if (create_one(&c, device_ids[i]) != 0) {
            for (pt_ctx* k : m->ranks) destroy_one(k);
            delete m;
            return 1;
        }
        c->rank = i; c->world = n_devices;
        m->ranks.push_back(c);
    }
    m->accum.assign((size_t)n_devices, nullptr);
    if (!m->rehearsal) {
        std::string err;
        bool ok = load_rccl(m->rccl, err);
        if (ok) {
            m->comms.assign((size_t)n_devices, nullptr);
            const ncclResult_t r = m->rccl.CommInitAll(m->comms.data(), n_devices, device_ids);
            if (r != ncclSuccess) { ok = false; err = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r); m->comms.clear(); }
        }
        if (!ok) {
            for (pt_ctx* k : m->ranks) destroy_one(k);
            if (m->rccl.lib) dlclose(m->rccl.lib);
            delete m;
            return fail(nullptr, "pt_create_multi: " + err);
        }
    }
    m->ranks[0]->multi = m;
    *out = m->ranks[0];
    return 0;
}

PT_API int pt_device_count(pt_ctx* c) { return !c ? 0 : (c->multi ? (int)c->multi->ranks.size() : 1); }

// f(rank context, rank index) on every rank of a group, each on its own host thread (rank 0 on the caller's); the first
// failure's message becomes the group's
template <typename F>
static int on_every_rank(pt_ctx* c, F f)
{
    pt_multi* m = c->multi;
    const size_t n = m->ranks.size();
    std::vector<int> rc(n, 0);
    std::vector<std::thread> th;
    for (size_t i = 1; i < n; i++) th.emplace_back([&, i]() { rc[i] = f(m->ranks[i], (int)i); });
    rc[0] = f(m->ranks[0], 0);
    for (auto& t : th) t.join();
    for (size_t i = 0; i < n; i++)
        if (rc[i] != 0) return fail(c, "rank " + std::to_string(i) + " (device " + std::to_string(m->ranks[i]->device) + "): " + m->ranks[i]->err);
    return 0;
}

static void free_scene(pt_ctx* c)
{
    ptd::free_lbvh(c->bvh);
    if (c->d_mats) { (void)hipFree(c->d_mats); c->d_mats = nullptr; }
    c->n_mats = 0;
    if (c->d_lights) { (void)hipFree(c->d_lights); c->d_lights = nullptr; }
    c->n_lights = 0; c->light_area = 0.0f;
}

PT_API void pt_destroy(pt_ctx* c)
{
    if (!c) return;
    if (pt_multi* m = c->multi) {
        c->multi = nullptr;
        for (size_t i = 0; i < m->ranks.size(); i++) { (void)hipSetDevice(m->ranks[i]->device); (void)hipStreamSynchronize(m->ranks[i]->stream); }
        for (ncclComm_t k : m->comms) if (k) (void)m->rccl.CommDestroy(k);
        for (size_t i = 0; i < m->ranks.size(); i++) if (m->accum[i]) { (void)hipSetDevice(m->ranks[i]->device); (void)hipFree(m->accum[i]); }
        for (size_t i = 1; i < m->ranks.size(); i++) destroy_one(m->ranks[i]);
        // librccl stays loaded: unloading a library that owns threads and device state at this point buys nothing
        delete m;
    }
    destroy_one(c);
}

static void destroy_one(pt_ctx* c)
{
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    if (c->d_queue) (void)hipFree(c->d_queue);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_frame_sums) (void)hipFree(c->d_frame_sums);
    if (c->d_wave_scratch) (void)hipFree(c->d_wave_scratch);
    if (c->d_stack_ovf) (void)hipFree(c->d_stack_ovf);
    if (c->d_row_spans) (void)hipFree(c->d_row_spans);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

static int pick_variant(const pt_ctx* c)
{
    const bool half_ok = c->bvh.half_area_ratio <= ptd::kHalfAreaLimit && c->bvh.half_box_inflation <= ptd::kHalfInflationLimit;
    const bool large = c->bvh.n_tris > ptd::kLargeSceneTris;
    if (!half_ok) return large ? ptd::kVariantF32Large : ptd::kVariantF32;
    if (c->bvh.n_tris > ptd::kWindowSceneTris) return ptd::kVariantF16W5Deep;     // long rays: earlier shade rounds, windowed stack
    int w5_blocks = 0;      // do five workgroups of the five-wave kernel fit a CU with this tree's stack depth?
    if (ptd::render_occupancy(ptd::kVariantF16W5, c->math_mode, c->stack_entries, c->bvh.n_nodes, &w5_blocks) == hipSuccess && w5_blocks >= 5) return ptd::kVariantF16W5;
    return ptd::kVariantF16W5Deep;      // small scene, deep tree: the windowed stack keeps five workgroups on a CU
}

// dwords per lane: one push per internal node on a root-to-leaf path at most (two-child tree), or one
// 8-byte group per level (four-wide tree, if built); every kernel variant gets the larger of the two
static int size_stack(pt_ctx* c)
{
    uint32_t need = c->bvh.max_depth + 1u;
    if (c->bvh.wrecs && need < 2u * (c->bvh.wide_depth + 1u)) need = 2u * (c->bvh.wide_depth + 1u);
    if (need < 8u) need = 8u;
    need = (need + 3u) & ~3u;
    if (need > 128u) return fail(c, "pt_set_scene: BVH deeper than the traversal stack supports");
    c->stack_entries = need;
    CK(c, ptd::render_occupancy(c->variant, c->math_mode, c->stack_entries, c->bvh.n_nodes, &c->blocks_per_cu));
    if (c->blocks_per_cu < 1) return fail(c, "render kernel does not fit on a CU with this stack size");
    return 0;
}

// The four-wide tree is built on first use (a kernel variant or diagnostic that walks it): its host-side
// collapse is not part of the default scene set-up.
// ... and so is the breadth-first copy of the tree's top, for the variants that stage it in LDS
static int ensure_top(pt_ctx* c)
{
    if (ptd::render_variant_top_nodes(c->variant) == 0) return 0;
    std::string err;
    if (!ptd::ensure_hnodes(c->bvh, c->stream, err)) return fail(c, err);      // (the staged top of the tree is a copy of the {lo, hi} nodes)
    if (!ptd::build_top_nodes(c->bvh, c->stream, err)) return fail(c, err);
    return 0;
}

static int ensure_wide(pt_ctx* c)
{
    if (c->bvh.wrecs || c->bvh.n_tris == 0) return 0;
    std::string err;
    if (!ptd::ensure_nodes(c->bvh, c->stream, err)) return fail(c, err);
    if (!ptd::build_wide4(c->bvh, c->stream, err)) return fail(c, err);
    return size_stack(c);
}

// The node array a kernel variant traverses, present on the device before anything is launched on it.  A scene keeps one of
// the two — set_scene_one releases the other — and any of them comes back on first use (lbvh_build.h).
static int ensure_node_format(pt_ctx* c, int fmt)
{
    if (c->bvh.n_tris == 0) return 0;
    std::string err;
    bool ok = true;
    switch (fmt) {
        case 0: case 5: ok = ptd::ensure_nodes(c->bvh, c->stream, err); break;
        case 7: case 8: case 9: ok = ptd::ensure_hnodes(c->bvh, c->stream, err); break;
        case 1: case 2: case 4: ok = ptd::ensure_qnodes(c->bvh, c->stream, err); break;
        case 6: ok = ptd::ensure_cnodes(c->bvh, c->stream, err); break;
        case 3: return ensure_wide(c);
        case 11: ok = ptd::ensure_hcnodes(c->bvh, c->stream, err); break;
#ifdef ACGPT_EXPERIMENTS
        case 13: case 14: ok = ptd::ensure_hcnodes(c->bvh, c->stream, err); break;
        case 10: ok = ptd::ensure_srecs(c->bvh, false, c->stream, err); break;       // 15-bit child references: scenes up to ~8 000 triangles
        case 12: ok = ptd::ensure_srecs(c->bvh, true, c->stream, err); break;
#endif
        default: return fail(c, "unknown node format");
    }
    return ok ? 0 : fail(c, err);
}
static int ensure_variant_arrays(pt_ctx* c)
{
    if (int rc = ensure_node_format(c, ptd::render_variant_node_format(c->variant))) return rc;
    return ensure_top(c);
}

static int set_scene_one(pt_ctx* c, const float* verts_xyzw, size_t n_verts, const uint32_t* idx, size_t n_tris,
                        const uint32_t* mat_ids, const pt_material* mats, size_t n_mats)
{
    if (!c) return fail(nullptr, "pt_set_scene: null context");
    if (n_tris > 0 && (!verts_xyzw || !idx || !mat_ids || !mats)) return fail(c, "pt_set_scene: null array");
    // the render kernel addresses nodes (64 B) and triangle records (48 B) with 32-bit byte offsets
    if (n_tris >= (1ull << 26) || n_verts >= 0xFFFFFFFFull) return fail(c, "pt_set_scene: too many triangles (limit 2^26) or vertices");
    for (size_t i = 0; i < 3 * n_tris; i++)
        if (idx[i] >= n_verts) return fail(c, "pt_set_scene: vertex index out of range");
    for (size_t i = 0; i < n_tris; i++)
        if (mat_ids[i] >= n_mats) return fail(c, "pt_set_scene: material index out of range (a face without a known usemtl has id 0xFFFFFFFF)");
    for (size_t i = 0; i < n_mats; i++)
        if (mats[i].bsdfType < 0 || mats[i].bsdfType > 2) return fail(c, "pt_set_scene: unknown bsdfType");
    if (n_mats > (size_t)ptd::kShadeMatMask + 1u) return fail(c, "pt_set_scene: more than 2^24 materials");
    CK(c, hipSetDevice(c->device));
    CK(c, hipStreamSynchronize(c->stream));
    Range range("acgpt: scene upload + BVH build");
    free_scene(c);
    std::string err;
    if (!ptd::build_lbvh(verts_xyzw, n_verts, idx, n_tris, mat_ids, c->build_mode, c->stream, c->bvh, err)) return fail(c, "pt_set_scene: " + err);
    if (c->build_mode != 0 && c->bvh.max_depth + 1u > 128u) {
        // Clustering by merged area has nothing to go by when thousands of triangles coincide (one box, one Morton code).  The builder pairs
        // ties by a hash once it sees the merges stall (lbvh_build.hip k_ploc_nn: 20 000 copies of one triangle, 96 levels instead of 20 000);
        // should a tree still outgrow the 128-entry lane stacks (size_stack below), the radix tree — equal codes split by their index bits,
        // always logarithmic — is taken instead of an error.
        free_scene(c);
        if (!ptd::build_lbvh(verts_xyzw, n_verts, idx, n_tris, mat_ids, 0, c->stream, c->bvh, err)) return fail(c, "pt_set_scene: " + err);
    }
    if (n_mats) {
        std::vector<ptd::DevMaterial> dm(n_mats);
        for (size_t i = 0; i < n_mats; i++) {
            const pt_material& m = mats[i];
            uint32_t b = (uint32_t)m.bsdfType; float bf; memcpy(&bf, &b, 4);
            dm[i].kd_ior = make_float4(m.diffuse.x, m.diffuse.y, m.diffuse.z, m.ior);
            dm[i].ke_bsdf = make_float4(m.emission.x, m.emission.y, m.emission.z, bf);
        }
        CK(c, hipMalloc((void**)&c->d_mats, n_mats * sizeof(ptd::DevMaterial)));
        CK(c, hipMemcpy(c->d_mats, dm.data(), n_mats * sizeof(ptd::DevMaterial), hipMemcpyHostToDevice));
        if (!ptd::tag_shade_records(c->bvh, c->d_mats, c->stream, err)) return fail(c, "pt_set_scene: " + err);
    }
    c->n_mats = (uint32_t)n_mats;
    {   // light mode 1: every triangle with an emissive material, in triangle order; same fp32 operations as the oracle's
        // orc_scene_create (edges by one subtraction, cross / length / normalize of sutil/vec_math.h:533-549, running area sum)
        std::vector<float4> lights;
        float run = 0.0f;
        for (size_t i = 0; i < n_tris; i++) {
            const pt_float3 ke = mats[mat_ids[i]].emission;
            if (!(sqrtf(ke.x * ke.x + ke.y * ke.y + ke.z * ke.z) > 0.0f)) continue;
            const float* a = verts_xyzw + 4 * (size_t)idx[3 * i], *b = verts_xyzw + 4 * (size_t)idx[3 * i + 1], *cc = verts_xyzw + 4 * (size_t)idx[3 * i + 2];
            const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]};
            const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            const float len = sqrtf(cx * cx + cy * cy + cz * cz);
            const float area = 0.5f * len;
            if (!(area > 0.0f)) continue;
            const float inv = 1.0f / len;
            run += area;
            lights.push_back(make_float4(a[0], a[1], a[2], area));
            lights.push_back(make_float4(e1[0], e1[1], e1[2], run));
            lights.push_back(make_float4(e2[0], e2[1], e2[2], 0.0f));
            lights.push_back(make_float4(cx * inv, cy * inv, cz * inv, 0.0f));
            lights.push_back(make_float4(ke.x, ke.y, ke.z, 0.0f));
        }
        if (!lights.empty()) {
            CK(c, hipMalloc((void**)&c->d_lights, lights.size() * sizeof(float4)));
            CK(c, hipMemcpy(c->d_lights, lights.data(), lights.size() * sizeof(float4), hipMemcpyHostToDevice));
        }
        c->n_lights = (uint32_t)(lights.size() / 5);
        c->light_area = run;
    }
    if (int rc = size_stack(c)) return rc;              // stack depth first: the choice below depends on it
    if (c->variant_auto) { c->variant = pick_variant(c); if (int rc = size_stack(c)) return rc; }
    {   // one node array per scene: the one the chosen kernel reads (fp16: 32 B per node, fp32: 64 B); the other comes back on first use
        const int fmt = ptd::render_variant_node_format(c->variant);
        if (fmt == 11) ptd::keep_one_node_array(c->bvh, 2);
        else if (fmt == 7 || fmt == 8 || fmt == 9) ptd::keep_one_node_array(c->bvh, 1);
        else if (fmt == 0 || fmt == 5) ptd::keep_one_node_array(c->bvh, 0);
    }
    if (int rc = ensure_variant_arrays(c)) return rc;
    c->scene_serial++;
    return 0;
}

PT_API uint64_t pt_scene_handle(pt_ctx* c) { return c ? c->scene_serial : 0; }

PT_API int pt_get_bvh_info(pt_ctx* c, pt_bvh_info* out)
{
    if (!c || !out) return fail(c, "pt_get_bvh_info: null argument");
    memset(out, 0, sizeof(*out));
    out->n_tris = c->bvh.n_tris;
    out->n_nodes = c->bvh.n_nodes;
    out->max_depth = c->bvh.max_depth;
    out->stack_entries = c->stack_entries;
    for (int k = 0; k < 3; k++) { out->scene_lo[k] = c->bvh.scene_lo[k]; out->scene_hi[k] = c->bvh.scene_hi[k]; }
    out->build_ms = c->bvh.build_ms;
    out->node_bytes = c->bvh.n_nodes * (uint32_t)sizeof(ptd::BvhNode);
    out->tri_bytes = c->bvh.n_tris * (uint32_t)sizeof(ptd::TriRecord);
    out->wide_nodes = c->bvh.n_wnodes;
    out->wide_depth = c->bvh.wide_depth;
    out->wide_bytes = c->bvh.n_wrecs * 48u;
    out->wide_ms = c->bvh.wide_ms;
    out->half_node_bytes = c->bvh.n_nodes * (uint32_t)sizeof(ptd::HNode);
    out->half_area_ratio = c->bvh.half_area_ratio;
    out->half_box_inflation = c->bvh.half_box_inflation;
    out->device_bytes = (uint64_t)ptd::scene_device_bytes(c->bvh);
    return 0;
}

PT_API int pt_set_partition(pt_ctx* c, int rank, int world)
{
    if (!c) return fail(nullptr, "pt_set_partition: null context");
    if (c->multi) return fail(c, "pt_set_partition: a pt_create_multi group partitions the image itself");
    if (world < 1 || rank < 0 || rank >= world) return fail(c, "pt_set_partition: need 0 <= rank < world");
    c->rank = rank; c->world = world;
    return 0;
}

static int set_scratch_limit_one(pt_ctx* c, size_t bytes)
{
    if (!c) return fail(nullptr, "pt_set_scratch_limit: null context");
    if (bytes < ((size_t)1 << 20)) return fail(c, "pt_set_scratch_limit: at least 1 MiB");
    c->scratch_limit = bytes;
    return 0;
}

static int set_light_mode_one(pt_ctx* c, int mode)
{
    if (!c) return fail(nullptr, "pt_set_light_mode: null context");
    if (mode != 0 && mode != 1) return fail(c, "pt_set_light_mode: 0 = the reference's estimator (hard-coded rectangle, PathTracerMain.cpp:154-158), 1 = scene lights + MIS");
    c->light_mode = mode;
    return 0;
}

static int set_math_mode_one(pt_ctx* c, int mode)
{
    if (!c) return fail(nullptr, "pt_set_math_mode: null context");
    if (mode != PT_MATH_IEEE && mode != PT_MATH_FAST) return fail(c, "pt_set_math_mode: PT_MATH_IEEE (0) or PT_MATH_FAST (1)");
    c->math_mode = mode;
    CK(c, hipSetDevice(c->device));           // the twin's occupancy (its register count differs)
    CK(c, ptd::render_occupancy(c->variant, c->math_mode, c->stack_entries, c->bvh.n_nodes, &c->blocks_per_cu));
    if (c->blocks_per_cu < 1) return fail(c, "pt_set_math_mode: the kernel of this mode does not fit the current scene in LDS");
    return 0;
}

static int set_sample_chunks_one(pt_ctx* c, int chunks)
{
    if (!c) return fail(nullptr, "pt_set_sample_chunks: null context");
    if (chunks != 0 && chunks != 1 && chunks != 2 && chunks != 4 && chunks != 8 && chunks != 16 && chunks != 32) return fail(c, "pt_set_sample_chunks: 0 (automatic), 1, 2, 4, 8, 16 or 32");
    c->chunks = chunks;
    return 0;
}

static int set_tuning_one(pt_ctx* c, int blocks_per_cu, int variant)
{
    if (!c) return fail(nullptr, "pt_set_tuning: null context");
    if (blocks_per_cu < 0 || blocks_per_cu > 16) return fail(c, "pt_set_tuning: blocks_per_cu out of range");
    if (variant < -1 || variant >= ptd::render_variant_count()) return fail(c, "pt_set_tuning: unknown kernel variant");
    CK(c, hipSetDevice(c->device));
    c->tune_blocks_per_cu = blocks_per_cu;
    c->variant_auto = variant < 0;
    c->variant = variant < 0 ? pick_variant(c) : variant;
    if (int rc = ensure_variant_arrays(c)) return rc;
    CK(c, ptd::render_occupancy(c->variant, c->math_mode, c->stack_entries, c->bvh.n_nodes, &c->blocks_per_cu));
    if (c->blocks_per_cu < 1) return fail(c, "pt_set_tuning: this kernel variant does not fit the current scene in LDS");
    return 0;
}

static int set_build_mode_one(pt_ctx* c, int mode)
{
    if (!c) return fail(nullptr, "pt_set_build_mode: null context");
    if (mode != 0 && mode != 1 && mode != 2) return fail(c, "pt_set_build_mode: 0 = Karras LBVH, 1 = PLOC, 2 = PLOC + insertion-based optimisation");
    c->build_mode = mode;
    return 0;
}

PT_API int pt_set_scene(pt_ctx* c, const float* verts_xyzw, size_t n_verts, const uint32_t* idx, size_t n_tris,
                        const uint32_t* mat_ids, const pt_material* mats, size_t n_mats)
{
    if (c && c->multi) return on_every_rank(c, [&](pt_ctx* r, int) { return set_scene_one(r, verts_xyzw, n_verts, idx, n_tris, mat_ids, mats, n_mats); });
    return set_scene_one(c, verts_xyzw, n_verts, idx, n_tris, mat_ids, mats, n_mats);
}

PT_API int pt_set_scratch_limit(pt_ctx* c, size_t bytes)
{
    if (c && c->multi) return on_every_rank(c, [&](pt_ctx* r, int) { return set_scratch_limit_one(r, bytes); });
    return set_scratch_limit_one(c, bytes);
}

PT_API int pt_set_light_mode(pt_ctx* c, int mode)
{
    if (c && c->multi) return on_every_rank(c, [&](pt_ctx* r, int) { return set_light_mode_one(r, mode); });
    return set_light_mode_one(c, mode);
}

PT_API int pt_set_math_mode(pt_ctx* c, int mode)
{
    if (c && c->multi) return on_every_rank(c, [&](pt_ctx* r, int) { return set_math_mode_one(r, mode); });
    return set_math_mode_one(c, mode);
}

PT_API int pt_set_sample_chunks(pt_ctx* c, int chunks)
{
    if (c && c->multi) return on_every_rank(c, [&](pt_ctx* r, int) { return set_sample_chunks_one(r, chunks); });
    return set_sample_chunks_one(c, chunks);
}

PT_API int pt_set_tuning(pt_ctx* c, int blocks_per_cu, int variant)
{
    if (c && c->multi) return on_every_rank(c, [&](pt_ctx* r, int) { return set_tuning_one(r, blocks_per_cu, variant); });
    return set_tuning_one(c, blocks_per_cu, variant);
}

PT_API int pt_set_build_mode(pt_ctx* c, int mode)
{
    if (c && c->multi) return on_every_rank(c, [&](pt_ctx* r, int) { return set_build_mode_one(r, mode); });
    return set_build_mode_one(c, mode);
}

PT_API const char* pt_variant_name(int variant)
{
    return (variant >= 0 && variant < ptd::render_variant_count()) ? ptd::render_variant_name(variant) : nullptr;
}

PT_API const char* pt_variant_kernel(int variant, int math_mode)
{
    return (variant >= 0 && variant < ptd::render_variant_count()) ? ptd::render_variant_kernel(variant, math_mode) : nullptr;
}

#ifndef ACGPT_KERNEL_SRC_HASH
#define ACGPT_KERNEL_SRC_HASH "unknown"
#endif
PT_API const char* pt_kernel_source_hash(void) { return ACGPT_KERNEL_SRC_HASH; }

PT_API int pt_set_stream(pt_ctx* c, void* s)
{
    if (!c) return fail(nullptr, "pt_set_stream: null context");
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return 0;
}

static ptd::DeviceScene device_scene(pt_ctx* c)
{
    ptd::DeviceScene sc;
    sc.nodes = c->bvh.nodes; sc.qnodes = c->bvh.qnodes; sc.cnodes = c->bvh.cnodes; sc.hnodes = c->bvh.hnodes; sc.top = c->bvh.top_nodes; sc.n_top = c->bvh.n_top; sc.hspace = c->bvh.hspace; sc.grid = c->bvh.grid; sc.tris = c->bvh.tris; sc.shade = c->bvh.shade; sc.wrecs = c->bvh.wrecs; sc.hcnodes = c->bvh.hcnodes_alt ? c->bvh.hcnodes_alt : c->bvh.hcnodes; sc.srecs = c->bvh.srecs; sc.sspace = c->bvh.sspace; sc.mats = c->d_mats;
    sc.n_tris = c->bvh.n_tris; sc.n_mats = c->n_mats;
    sc.lights = c->d_lights; sc.n_lights = c->n_lights; sc.light_area = c->light_area;
    return sc;
}

// StaticWorkDistribution::numSamples, sutil/WorkDistribution.h:50-57
static uint32_t num_samples(int world, uint32_t w, uint32_t h)
{
    const uint32_t strip_w = 8u * (uint32_t)world, strip_h = 4u;
    const uint32_t cols = w / strip_w + (w % strip_w == 0 ? 0 : 1);
    const uint32_t rows = h / strip_h + (h % strip_h == 0 ? 0 : 1);
    return rows * cols * 32u;
}

// Granlund & Montgomery division by an invariant (N = 32): exact for every 32-bit numerator
static ptd::FastDiv make_fast_div(uint32_t d)
{
    uint32_t l = 0;
    while (((uint64_t)1 << l) < d) l++;
    ptd::FastDiv f;
    f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1u);
    f.sh1 = l < 1u ? l : 1u;
    f.sh2 = l < 1u ? 0u : l - 1u;
    return f;
}
static bool check_fast_div(const ptd::FastDiv& f, uint32_t d, uint32_t n_max)
{
    auto q = [&](uint32_t n) { const uint32_t t = (uint32_t)(((uint64_t)n * f.mul) >> 32); return (t + ((n - t) >> f.sh1)) >> f.sh2; };
    for (uint64_t m = 0; m <= (uint64_t)n_max + d; m += d)           // every multiple of d and its neighbours
        for (int k = -1; k <= 1; k++) {
            const uint64_t n = m + (uint64_t)(int64_t)k;
            if (n <= 0xFFFFFFFFull && (m != 0 || k >= 0) && q((uint32_t)n) != (uint32_t)(n / d)) return false;
        }
    return q(0xFFFFFFFFu) == 0xFFFFFFFFu / d;
}

// ---- pixel classes -------------------------------------------------------------------------------------------------------------
// The rays through a pixel are D = dx U + dy V + W with (dx, dy) in the pixel's square of the image plane
// (pathTracerPrograms.cu:730-737).  The rays that meet a convex box are those whose (dx, dy) fall into the convex hull of the
// box's eight projected corners.  Per image row: the columns outside which a pixel's square lies wholly outside that hull (by
// a quarter pixel and with the box grown by a thousandth of the scene: no ray through such a pixel can hit anything — the kernel
// books its samples as misses without starting them), and the columns inside which the square lies wholly inside it (path starts
// there skip the cull test; a hint only: a ray that misses after all is traversed and misses).  false: the box is not entirely
// in front of the eye, or the camera frame is degenerate — no classes for this launch.
static bool row_spans(const pt_params* p, const float lo[3], const float hi[3], std::vector<uint32_t>& out)
{
    const uint32_t W = p->width, H = p->height;
    out.assign((size_t)2 * H, 0u);
    const double U[3] = {p->cameraU.x, p->cameraU.y, p->cameraU.z}, V[3] = {p->cameraV.x, p->cameraV.y, p->cameraV.z}, Wv[3] = {p->cameraW.x, p->cameraW.y, p->cameraW.z};
    const double E[3] = {p->cameraEye.x, p->cameraEye.y, p->cameraEye.z};
    auto cross = [](const double* a, const double* b, double* r) { r[0] = a[1] * b[2] - a[2] * b[1]; r[1] = a[2] * b[0] - a[0] * b[2]; r[2] = a[0] * b[1] - a[1] * b[0]; };
    auto dot = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
    double VxW[3], WxU[3], UxV[3];
    cross(V, Wv, VxW); cross(Wv, U, WxU); cross(U, V, UxV);
    const double det = dot(U, VxW);
    const double scale = std::sqrt(dot(U, U) * dot(V, V) * dot(Wv, Wv));
    if (!(std::fabs(det) > 1e-9 * scale) || !std::isfinite(det)) return false;
    double ext = 0.0;
    for (int k = 0; k < 3; k++) ext = std::max(ext, (double)hi[k] - (double)lo[k]);
    if (!(ext >= 0.0) || !std::isfinite(ext)) return false;
    const double pad = 1e-3 * ext + 1e-6;
    std::vector<std::pair<double, double>> pts;
    for (int i = 0; i < 8; i++) {
        const double c[3] = {((i & 1) ? hi[0] + pad : lo[0] - pad) - E[0], ((i & 2) ? hi[1] + pad : lo[1] - pad) - E[1], ((i & 4) ? hi[2] + pad : lo[2] - pad) - E[2]};
        const double a = dot(c, VxW) / det, b = dot(c, WxU) / det, t = dot(c, UxV) / det;      // c = a U + b V + t W
        if (!(t > 1e-6 * (std::fabs(a) + std::fabs(b) + 1.0)) || !std::isfinite(a + b + t)) return false;   // at or behind the eye plane
        pts.emplace_back(a / t, b / t);
    }
    // convex hull (monotone chain), counter-clockwise
    std::sort(pts.begin(), pts.end());
    std::vector<std::pair<double, double>> hull(16);
    int k = 0;
    auto turn = [](const std::pair<double, double>& o, const std::pair<double, double>& a, const std::pair<double, double>& b) { return (a.first - o.first) * (b.second - o.second) - (a.second - o.second) * (b.first - o.first); };
    for (int i = 0; i < 8; i++) { while (k >= 2 && turn(hull[k - 2], hull[k - 1], pts[i]) <= 0) k--; hull[k++] = pts[i]; }
    for (int i = 6, t0 = k + 1; i >= 0; i--) { while (k >= t0 && turn(hull[k - 2], hull[k - 1], pts[i]) <= 0) k--; hull[k++] = pts[i]; }
    const int n = k - 1;
    if (n < 3) return false;
    // the hull's x interval on the horizontal line at height y; false: the line misses the hull
    auto interval = [&](double y, double& l, double& r) {
        l = 1e300; r = -1e300;
        for (int i = 0; i < n; i++) {
            const auto& a = hull[i]; const auto& b = hull[(i + 1) % n];
            const double ya = a.second, yb = b.second;
            if ((ya <= y && y <= yb) || (yb <= y && y <= ya)) {
                const double x = ya == yb ? a.first : a.first + (b.first - a.first) * (y - ya) / (yb - ya);
                l = std::min(l, ya == yb ? std::min(a.first, b.first) : x);
                r = std::max(r, ya == yb ? std::max(a.first, b.first) : x);
            }
        }
        return l <= r;
    };
    double ymin = 1e300, ymax = -1e300;
    for (int i = 0; i < n; i++) { ymin = std::min(ymin, hull[i].second); ymax = std::max(ymax, hull[i].second); }
    const double mx = 0.5 / W, my = 0.5 / H;                  // a quarter pixel in image-plane units (a pixel is 2 / W wide)
    for (uint32_t y = 0; y < H; y++) {
        const double y0 = 2.0 * y / H - 1.0, y1 = 2.0 * (y + 1) / H - 1.0;
        uint32_t out_lo = 0, out_hi = 0, in_lo = 0, in_hi = 0;
        if (!(y1 + my < ymin || y0 - my > ymax)) {
            // outer: the hull's extent over the band [y0 - my, y1 + my] — at the band's two edges (clamped into the hull's own
            // range) and at every hull vertex inside the band
            double l = 1e300, r = -1e300, a, b;
            const double e0 = std::max(y0 - my, ymin), e1 = std::min(y1 + my, ymax);
            if (interval(e0, a, b)) { l = std::min(l, a); r = std::max(r, b); }
            if (interval(e1, a, b)) { l = std::min(l, a); r = std::max(r, b); }
            for (int i = 0; i < n; i++) if (hull[i].second >= e0 && hull[i].second <= e1) { l = std::min(l, hull[i].first); r = std::max(r, hull[i].first); }
            if (l <= r) {
                const double fl = std::floor((l - mx + 1.0) * 0.5 * W), fh = std::ceil((r + mx + 1.0) * 0.5 * W);
                out_lo = (uint32_t)std::min(std::max(fl, 0.0), (double)W);
                out_hi = (uint32_t)std::min(std::max(fh, 0.0), (double)W);
            }
            // inner: columns whose square lies inside the hull — convex, so inside at both edges of the (grown) band is enough
            double l0, r0, l1, r1;
            if (interval(y0 - my, l0, r0) && interval(y1 + my, l1, r1)) {
                const double il = std::max(l0, l1) + mx, ir = std::min(r0, r1) - mx;
                const double cl = std::ceil((il + 1.0) * 0.5 * W), ch = std::floor((ir + 1.0) * 0.5 * W);
                if (cl < ch) { in_lo = (uint32_t)std::min(std::max(cl, 0.0), (double)W); in_hi = (uint32_t)std::min(std::max(ch, 0.0), (double)W); }
                if (in_lo < out_lo) in_lo = out_lo;
                if (in_hi > out_hi) in_hi = out_hi;
                if (in_lo >= in_hi) in_lo = in_hi = 0;
            }
        }
        out[2 * (size_t)y] = out_lo | (out_hi << 16);
        out[2 * (size_t)y + 1] = in_lo | (in_hi << 16);
    }
    return true;
}

PT_API int pt_launch(pt_ctx* c, const pt_params* p) { return pt_launch_frames(c, p, 1u); }

static int launch_batch(pt_ctx* c, const pt_params* p, uint32_t n_frames);
static int launch_frames_single(pt_ctx* c, const pt_params* p, uint32_t n_frames);
static int launch_frames_multi(pt_ctx* c, const pt_params* p, uint32_t n_frames);

PT_API int pt_launch_frames(pt_ctx* c, const pt_params* p, uint32_t n_frames)
{
    if (!c || !p) return fail(c, "pt_launch: null argument");
    if (n_frames < 1u || n_frames > 64u) return fail(c, "pt_launch_frames: n_frames must be in [1, 64]");
    Range range("acgpt: pt_launch_frames");
    return c->multi ? launch_frames_multi(c, p, n_frames) : launch_frames_single(c, p, n_frames);
}

// A group's launch: every rank renders its tiles of the same sub-frames into its private buffer (own host thread, own
// stream), then one reduce to rank 0 and make_color there.  Synchronised on return, like the single-GPU launch.
static int launch_frames_multi(pt_ctx* c, const pt_params* p, uint32_t n_frames)
{
    const auto t0 = std::chrono::steady_clock::now();
    pt_multi* m = c->multi;
    const int world = (int)m->ranks.size();
    if (p->width == 0 || p->height == 0) return fail(c, "pt_launch: empty image");
    if (!p->accumulationBuffer) return fail(c, "pt_launch: accumulationBuffer is null");
    const size_t pixels = (size_t)p->width * p->height;
    if (pixels > 0x7FFFFFFFull / 4u) return fail(c, "pt_launch: image too large for the group reduce");
    // private buffers: (re)allocated zero-filled; a launch that is not the straight continuation of the previous one with
    // frame > 0 (a restored accumulation) first takes the caller's values for the rank's own pixels
    // The tile layout depends on the image SHAPE, not on its pixel count (640x360 and 360x640 share one): keyed on (width, height).
    const bool fresh = m->accum_w != p->width || m->accum_h != p->height;
    const bool continues = !fresh && p->currentFrameIdx > 0u && p->accumulationBuffer == m->cont_accum && p->currentFrameIdx == m->cont_frame && p->width == m->cont_w && p->height == m->cont_h;
    if (!continues && p->currentFrameIdx > 0u) {      // the ranks are about to read the caller's buffer: what the caller queued on its stream (pt_set_stream) comes first
        CK(c, hipSetDevice(c->device));
        CK(c, hipStreamSynchronize(c->stream));
    }
    int rc = on_every_rank(c, [&](pt_ctx* r, int i) -> int {
        CK(r, hipSetDevice(r->device));
        if (fresh) {
            if (m->accum[(size_t)i]) { CK(r, hipStreamSynchronize(r->stream)); (void)hipFree(m->accum[(size_t)i]); m->accum[(size_t)i] = nullptr; }
            CK(r, hipMalloc((void**)&m->accum[(size_t)i], pixels * sizeof(float4)));
        }
        // every launch that is not a straight continuation starts from "own pixels or zero": the reduce relies on every pixel
        // having exactly one non-zero term, whatever an earlier launch of another shape or frame sequence left behind
        if (!continues) CK(r, hipMemsetAsync(m->accum[(size_t)i], 0, pixels * sizeof(float4), r->stream));
        if (p->currentFrameIdx > 0u && !continues) {
            CK(r, hipMemcpyAsync(m->accum[(size_t)i], p->accumulationBuffer, pixels * sizeof(float4), hipMemcpyDefault, r->stream));
            CK(r, ptd::launch_keep_owned(m->accum[(size_t)i], p->width, p->height, i, world, r->stream));
        }
        pt_params q = *p;
        q.accumulationBuffer = (float*)m->accum[(size_t)i];
        q.frameBuffer = nullptr;
        r->rank = i; r->world = world;
        return launch_frames_single(r, &q, n_frames);
    });
    if (rc) return rc;
    m->accum_w = p->width; m->accum_h = p->height;
    {
        Range range("acgpt: reduce of the accumulation buffers to rank 0");
        const auto tr = std::chrono::steady_clock::now();
        if (m->rehearsal) {         // every buffer lives on the one device: add them there
            std::vector<const float4*> src(m->accum.begin(), m->accum.end());
            CK(c, hipSetDevice(c->device));
            CK(c, ptd::launch_sum_ranks((float4*)p->accumulationBuffer, src.data(), world, (uint32_t)pixels, c->stream));
            CK(c, hipStreamSynchronize(c->stream));
        } else {
            // no early return between GroupStart and GroupEnd: an open group would leave the communicators unusable
            ncclResult_t r = m->rccl.GroupStart();
            hipError_t dev_err = hipSuccess;
            for (int i = 0; i < world && r == ncclSuccess && dev_err == hipSuccess; i++) {
                dev_err = hipSetDevice(m->ranks[(size_t)i]->device);
                if (dev_err != hipSuccess) break;
                r = m->rccl.Reduce(m->accum[(size_t)i], i == 0 ? (void*)p->accumulationBuffer : (void*)m->accum[(size_t)i], pixels * 4u, ncclFloat, ncclSum, 0,
                                   m->comms[(size_t)i], m->ranks[(size_t)i]->stream);
            }
            const ncclResult_t e = m->rccl.GroupEnd();
            if (r == ncclSuccess) r = e;
            if (dev_err != hipSuccess) return fail(c, std::string("pt_launch: hipSetDevice before ncclReduce: ") + hipGetErrorString(dev_err));
            if (r != ncclSuccess) return fail(c, std::string("pt_launch: ncclReduce: ") + m->rccl.GetErrorString(r));
            for (int i = world - 1; i >= 0; i--) { CK(c, hipSetDevice(m->ranks[(size_t)i]->device)); CK(c, hipStreamSynchronize(m->ranks[(size_t)i]->stream)); }
        }
        m->reduce_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - tr).count();
    }
    if (p->frameBuffer) {
        CK(c, hipSetDevice(c->device));
        CK(c, ptd::launch_resolve((const float4*)p->accumulationBuffer, (uint32_t*)p->frameBuffer, (uint32_t)pixels, c->stream));
        CK(c, hipStreamSynchronize(c->stream));
    }
    // the group's counters: rays, paths and pixels summed, the slowest rank's kernel time
    pt_stats total = m->ranks[0]->stats;
    for (int i = 1; i < world; i++) {
        const pt_stats& s = m->ranks[(size_t)i]->stats;
        total.radiance_rays += s.radiance_rays; total.shadow_rays += s.shadow_rays; total.paths += s.paths; total.culled_rays += s.culled_rays;
        total.pixels += s.pixels; total.grid_blocks += s.grid_blocks;
        total.trav_wave_steps += s.trav_wave_steps; total.trav_lane_steps += s.trav_lane_steps;
        total.shade_wave_rounds += s.shade_wave_rounds; total.shade_lane_rounds += s.shade_lane_rounds;
        if (s.kernel_ms > total.kernel_ms) total.kernel_ms = s.kernel_ms;
    }
    total.launch_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    m->cont_accum = p->accumulationBuffer; m->cont_frame = p->currentFrameIdx + n_frames; m->cont_w = p->width; m->cont_h = p->height;
    m->group_stats = total;
    return 0;
}

static int launch_frames_single(pt_ctx* c, const pt_params* p, uint32_t n_frames)
{
    // one kernel launch holds one float4 per (pixel, sub-frame) until k_finalize has blended them: batches whose sums
    // would exceed the scratch limit (1 GiB: 32 sub-frames at 1920x1080) run as several launches, same bits
    const size_t per_frame = (size_t)num_samples(c->world, p->width, p->height) * sizeof(float4);      // this rank's pixel slots
    uint32_t per_launch = per_frame ? (uint32_t)(c->scratch_limit / per_frame) : n_frames;
    if (per_launch < 1u) per_launch = 1u;
    if (per_launch >= n_frames) return launch_batch(c, p, n_frames);
    pt_params q = *p;
    pt_stats total;
    memset(&total, 0, sizeof(total));
    for (uint32_t done = 0; done < n_frames;) {
        const uint32_t n = n_frames - done < per_launch ? n_frames - done : per_launch;
        q.currentFrameIdx = p->currentFrameIdx + done;
        if (int rc = launch_batch(c, &q, n)) return rc;
        const pt_stats& s = c->stats;
        total.radiance_rays += s.radiance_rays; total.shadow_rays += s.shadow_rays; total.paths += s.paths; total.culled_rays += s.culled_rays;
        total.kernel_ms += s.kernel_ms; total.launch_ms += s.launch_ms;
        total.trav_wave_steps += s.trav_wave_steps; total.trav_lane_steps += s.trav_lane_steps;
        total.shade_wave_rounds += s.shade_wave_rounds; total.shade_lane_rounds += s.shade_lane_rounds;
        total.pixels = s.pixels; total.grid_blocks = s.grid_blocks; total.sample_chunks = s.sample_chunks; total.variant = s.variant; total.math_mode = s.math_mode;
        done += n;
    }
    c->stats = total;
    return 0;
}

static int launch_batch(pt_ctx* c, const pt_params* p, uint32_t n_frames)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (p->width == 0 || p->height == 0) return fail(c, "pt_launch: empty image");
    if (p->width > 65535u || p->height > 65535u) return fail(c, "pt_launch: width and height are limited to 65535 (work items pack them into 16 bits each)");
    if (p->samplesPerPixel == 0) return fail(c, "pt_launch: samplesPerPixel must be >= 1 (do{}while(--i), pathTracerPrograms.cu:727,780)");
    if (p->maxDepth < 1 || p->maxDepth > 28) return fail(c, "pt_launch: maxDepth must be in [1, 28] (PathTracerMain.cpp:42, 122-128)");
    if (!p->accumulationBuffer) return fail(c, "pt_launch: accumulationBuffer is null");
    if (p->handle != 0 && p->handle != c->scene_serial) return fail(c, "pt_launch: stale scene handle");
    CK(c, hipSetDevice(c->device));
    // light mode 1 has its own kernel (the estimator differs); every other choice is c->variant.  The node array that kernel
    // walks is on the device before the launch (a scene keeps one array; the lights kernel reads the fp16 nodes whatever the scene chose)
    const int variant = c->light_mode == 1 ? ptd::kVariantLights : c->variant;
    if (int rc = ensure_node_format(c, ptd::render_variant_node_format(variant))) return rc;

    ptd::RenderArgs a;
    memset(&a, 0, sizeof(a));
    a.scene = device_scene(c);
    a.accum = (float4*)p->accumulationBuffer;
    a.fb = (uint32_t*)p->frameBuffer;
    a.width = p->width; a.height = p->height; a.spp = p->samplesPerPixel; a.maxDepth = p->maxDepth; a.frame = p->currentFrameIdx;
    a.eye = p->cameraEye; a.U = p->cameraU; a.V = p->cameraV; a.W = p->cameraW;
    a.light = p->areaLight;
    {   // :1021 length(cross(v1, v2)) with the operations of sutil/vec_math.h:533-542 (this file is built with -ffp-contract=off)
        const pt_float3 u = p->areaLight.v1, v = p->areaLight.v2;
        const float cx = u.y * v.z - u.z * v.y, cy = u.z * v.x - u.x * v.z, cz = u.x * v.y - u.y * v.x;
        a.light_area = sqrtf(cx * cx + cy * cy + cz * cz);
    }
    a.useDL = p->useDirectLighting ? 1u : 0u;
    a.useIS = p->useImportanceSampling ? 1u : 0u;
    a.rank = c->rank; a.world = c->world;
    if (c->chunks > 0) {
        if (p->samplesPerPixel % (uint32_t)c->chunks != 0u) return fail(c, "pt_launch: samplesPerPixel is not a multiple of the sample-chunk count");
        while ((1 << a.chunk_shift) < c->chunks) a.chunk_shift++;
    } else {
        // automatic: 8 runs per pixel, 16 when this rank holds fewer than 2^20 pixels (shorter items keep the
        // tail of the launch short), as long as every run keeps >= 4 samples
        const uint32_t my_pixels = (uint32_t)(((uint64_t)p->width * p->height) / (uint64_t)c->world);
        uint32_t want = my_pixels < (1u << 20) ? 4u : 3u;      // (32 runs below 2^19 pixels until round 3: 22.4 ms for a rank of eight against 22.1 with 16, profiles/r03_sweep_partitions.txt)
        while (want > 0u && (p->samplesPerPixel % (1u << want) != 0u || (p->samplesPerPixel >> want) < 4u)) want--;
        a.chunk_shift = want;
    }
    // items per queue atomic.  A grant is decoded once, by up to 64 lanes side by side (one (pixel, step) group each), so it holds
    // at most 64 groups; within that, larger grants mean fewer atomics and decodes (64 -> 256 items at 8 runs: -1.3 % at full
    // size, -3.6 % when a rank holds 1/8 of the tiles, profiles/r02_sweep_grant.txt).  A scene that does not fit one XCD's 4 MB
    // of L2 wants the opposite: a wave's 64 lanes on as few pixels as possible, so that its camera and first-bounce rays walk the
    // same lines (1.31 M triangles: grant 64 is 6 % faster than 256, profiles/r02_sweep_grant_c5.txt).
    static const uint32_t grant_on_l2[6] = {32u, 64u, 128u, 256u, 256u, 256u};
    static const uint32_t grant_beyond_l2[6] = {16u, 16u, 32u, 64u, 64u, 64u};
    const size_t scene_bytes = (size_t)c->bvh.n_nodes * sizeof(ptd::HNode) + (size_t)c->bvh.n_tris * sizeof(ptd::TriRecord);
    a.grant = (scene_bytes <= ((size_t)4 << 20) ? grant_on_l2 : grant_beyond_l2)[a.chunk_shift];      // (capped for small launches once the grid is known, below)
    a.chunk_spp = p->samplesPerPixel >> a.chunk_shift;
    a.n_frames = n_frames;
    a.sub_shift = a.chunk_shift;
    while ((1u << (a.sub_shift - a.chunk_shift)) < n_frames) a.sub_shift++;
    {   // LCG skip-ahead: chunk k starts 2 * k * chunk_spp draws after the pixel seed (two jitter draws per sample, :730)
        uint32_t mul = 1u, add = 0u;
        for (uint32_t k = 0; k < 32u; k++) {
            a.lcg_mul[k] = mul; a.lcg_add[k] = add;
            for (uint32_t i = 0; i < 2u * a.chunk_spp; i++) { add = 1664525u * add + 1013904223u; mul = 1664525u * mul; }
        }
    }
    if (((uint64_t)num_samples(c->world, p->width, p->height) << a.sub_shift) >= 0x7FFFFFFFull)
        return fail(c, "pt_launch: image too large for this sample-chunk count and frame batch (2^31 work items)");
    a.total_samples = num_samples(c->world, p->width, p->height) << a.sub_shift;
    a.strip_cols = p->width / (8u * (uint32_t)c->world) + (p->width % (8u * (uint32_t)c->world) == 0 ? 0u : 1u);
    if (c->div_width != p->width || c->div_world_n != c->world) {      // the constants depend on (width, world) only: built and verified when those change
        const ptd::FastDiv dc = make_fast_div(a.strip_cols), dw = make_fast_div((uint32_t)c->world);
        if (!check_fast_div(dc, a.strip_cols, (65536u / 4u + 1u) * a.strip_cols) ||          // every tile-strip index of the tallest image
            !check_fast_div(dw, (uint32_t)c->world, 65536u + (uint32_t)c->world))
            return fail(c, "pt_launch: internal error (division constants)");
        c->div_cols = dc; c->div_world = dw; c->div_width = p->width; c->div_world_n = c->world;
    }
    a.div_cols = c->div_cols;
    a.div_world = c->div_world;
    a.shard_size = ((a.total_samples + 7u) / 8u + 63u) & ~63u;
    a.row_interleave = (uint32_t)c->queue_order;
    a.strip_rows = p->height / 4u + (p->height % 4u == 0 ? 0u : 1u);
    {   // one float4 per (pixel slot of this rank's tile order, sub-frame): what the megakernel hands to k_finalize
        const size_t need = (size_t)num_samples(c->world, p->width, p->height) * n_frames * sizeof(float4);
        if (need > c->frame_sums_bytes) {
            CK(c, hipStreamSynchronize(c->stream));
            if (c->d_frame_sums) { (void)hipFree(c->d_frame_sums); c->d_frame_sums = nullptr; c->frame_sums_bytes = 0; }
            CK(c, hipMalloc((void**)&c->d_frame_sums, need));
            c->frame_sums_bytes = need;
        }
        a.frame_sums = c->d_frame_sums;
    }
    {   // scene box for the camera-ray cull, enlarged by 2^-10 of its extent (+ a floor) beyond the roundings of reaches_scene()
        float ext = 0.0f;
        for (int k = 0; k < 3; k++) ext = fmaxf(ext, c->bvh.scene_hi[k] - c->bvh.scene_lo[k]);
        const float grow = ext * (1.0f / 1024.0f) + 1e-6f;
        a.cull_lo = {c->bvh.scene_lo[0] - grow, c->bvh.scene_lo[1] - grow, c->bvh.scene_lo[2] - grow};
        a.cull_hi = {c->bvh.scene_hi[0] + grow, c->bvh.scene_hi[1] + grow, c->bvh.scene_hi[2] + grow};
    }
    {   // origin-triangle release: the rounding of a hit point scales with the largest coordinate it can have
        float cm = 0.0f;
        for (int k = 0; k < 3; k++) cm = fmaxf(cm, fmaxf(fabsf(c->bvh.scene_lo[k]), fabsf(c->bvh.scene_hi[k])));
        a.skip_base = ptd::kOriginEps * cm;
    }
    a.row_spans = nullptr;
    if (c->pixel_classes && c->bvh.n_tris != 0u && p->height <= 32767u) {
        // recomputed only when the camera, the image size or the scene box change
        std::vector<float> key = {(float)p->width, (float)p->height, p->cameraEye.x, p->cameraEye.y, p->cameraEye.z, p->cameraU.x, p->cameraU.y, p->cameraU.z,
                                  p->cameraV.x, p->cameraV.y, p->cameraV.z, p->cameraW.x, p->cameraW.y, p->cameraW.z,
                                  c->bvh.scene_lo[0], c->bvh.scene_lo[1], c->bvh.scene_lo[2], c->bvh.scene_hi[0], c->bvh.scene_hi[1], c->bvh.scene_hi[2], 1.0f};
        // (the last element records whether the spans could be computed at all: not part of the comparison)
        if (key.size() != c->spans_key.size() || memcmp(key.data(), c->spans_key.data(), (key.size() - 1) * sizeof(float)) != 0) {
            std::vector<uint32_t> spans;
            const bool ok = row_spans(p, c->bvh.scene_lo, c->bvh.scene_hi, spans);
            key.back() = ok ? 1.0f : 0.0f;
            if (ok) {
                if (c->row_spans_rows < p->height) {
                    CK(c, hipStreamSynchronize(c->stream));
                    if (c->d_row_spans) { (void)hipFree(c->d_row_spans); c->d_row_spans = nullptr; c->row_spans_rows = 0; }
                    CK(c, hipMalloc((void**)&c->d_row_spans, (size_t)p->height * sizeof(uint2)));
                    c->row_spans_rows = p->height;
                }
                CK(c, hipStreamSynchronize(c->stream));       // a launch in flight may still read the old spans (launches return synchronised, so this is a formality)
                CK(c, hipMemcpy(c->d_row_spans, spans.data(), (size_t)p->height * sizeof(uint2), hipMemcpyHostToDevice));
            }
            c->spans_key = key;
        }
        if (c->spans_key.back() == 1.0f) a.row_spans = c->d_row_spans;
    }
    a.queue_heads = c->d_queue;
    a.counters = c->d_counters;
    a.stack_entries = c->stack_entries;
    a.n_lds_nodes = c->bvh.n_nodes;

    int fit = c->blocks_per_cu;
    if (variant != c->variant || fit < 1) {   // (fit < 1: no scene yet — empty world, every ray misses)
        CK(c, ptd::render_occupancy(variant, c->math_mode, c->stack_entries, c->bvh.n_nodes, &fit));
        if (variant == c->variant) c->blocks_per_cu = fit;
        if (fit < 1) fit = 1;
    }
    int bpc = c->tune_blocks_per_cu > 0 ? c->tune_blocks_per_cu : fit;
    if (bpc > fit) bpc = fit;
    uint32_t grid = (uint32_t)c->n_cus * (uint32_t)bpc;
    const uint32_t waves_needed = (a.total_samples + 63u) / 64u;
    const uint32_t wpb = (uint32_t)ptd::render_variant_threads(variant) / 64u;
    const uint32_t blocks_needed = (waves_needed + wpb - 1) / wpb;
    if (grid > blocks_needed) grid = blocks_needed;
    if (grid < 1) grid = 1;
    {   // a small launch wants small grants: a wave that has fetched its last grant works it off alone, so the launch ends one grant's
        // worth of work after the queue is empty.  The reference's own start-up workload in one launch (512 x 512 x 128 spp: 4.2 M items,
        // 820 per wave) with grants of 256 had a third of the items handed out in the first 60 us and its last wave finishing 1.9 ms after the
        // median one (5.2 ms a launch where eight fused steps take 2.8 each, profiles/r04_wave_timeline_c0.txt): keep >= 16 grants per wave.
        const uint32_t per_wave = a.total_samples / (grid * wpb);
        uint32_t cap = 1u << a.chunk_shift;                         // never below one (pixel, sub-frame) group
        while (cap * 2u * 16u <= per_wave) cap *= 2u;
        if (const char* e = getenv("ACGPT_GRANT")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1u && v <= (64u << a.chunk_shift)) { a.grant = v; cap = v; } }     // sweeps (a grant is decoded by 64 lanes, one group each: <= 64 groups)
        if (a.grant > cap) a.grant = cap;
    }

    if (a.chunk_shift) {   // fold slots for every wave of the grid
        const size_t need = (size_t)grid * wpb * ((size_t)ptd::kRenderFoldSlots << a.chunk_shift) * 3 * sizeof(float);
        if (need > c->wave_scratch_bytes) {
            CK(c, hipStreamSynchronize(c->stream));
            if (c->d_wave_scratch) { (void)hipFree(c->d_wave_scratch); c->d_wave_scratch = nullptr; c->wave_scratch_bytes = 0; }
            CK(c, hipMalloc((void**)&c->d_wave_scratch, need));
            c->wave_scratch_bytes = need;
        }
        a.wave_scratch = c->d_wave_scratch;
    }
    {   // kernels whose LDS stack is capped keep deeper entries here
        const int cap_signed = ptd::render_variant_stack_cap(variant);      // > 0: entries in LDS, the rest here; < 0: a sliding window of that many, every slot has a home here
        const uint32_t cap = (uint32_t)(cap_signed < 0 ? -cap_signed : cap_signed);
        if (cap > 0u && c->stack_entries > cap) {
            const size_t need = (size_t)grid * wpb * 64u * (cap_signed < 0 ? c->stack_entries : c->stack_entries - cap) * sizeof(uint32_t)
                                * ((ptd::render_variant_node_format(variant) == 10 || ptd::render_variant_node_format(variant) == 12) ? 2u : 1u);       // the shared-plane kernel's entries are {node, interval}
            if (need > c->stack_ovf_bytes) {
                CK(c, hipStreamSynchronize(c->stream));
                if (c->d_stack_ovf) { (void)hipFree(c->d_stack_ovf); c->d_stack_ovf = nullptr; c->stack_ovf_bytes = 0; }
                CK(c, hipMalloc((void**)&c->d_stack_ovf, need));
                c->stack_ovf_bytes = need;
            }
            a.stack_overflow = c->d_stack_ovf;
        }
    }
    CK(c, hipMemsetAsync(c->d_queue, 0, 8 * sizeof(uint32_t), c->stream));
    CK(c, hipMemsetAsync(c->d_counters, 0, (size_t)ptd::kCounterWords * sizeof(unsigned long long), c->stream));
    CK(c, hipEventRecord(c->ev0, c->stream));
    { Range range("acgpt: render megakernel (launch_batch)"); CK(c, ptd::launch_render(variant, c->math_mode, a, grid, c->stream)); }
    CK(c, hipEventRecord(c->ev1, c->stream));
    { Range range("acgpt: k_finalize"); CK(c, ptd::launch_finalize(a, c->stream)); }
    unsigned long long h[8], h_tail[2] = {0, 0};      // h_tail: culled camera rays; experiments build: workgroups of a wavefront kernel that gave up
    CK(c, hipMemcpyAsync(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipMemcpyAsync(h_tail, c->d_counters + ptd::kCulledCounter, sizeof(h_tail), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));            // CUDA_SYNC_CHECK, PathTracerMain.cpp:209
    const unsigned long long h_culled = h_tail[0];
#ifdef ACGPT_EXPERIMENTS
    if (h_tail[1] != 0) return fail(c, "pt_launch: the render kernel gave up (" + std::to_string(h_tail[1]) + " workgroups: scheduling error or watchdog); the image is incomplete");
#endif
    float ms = 0.0f;
    CK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->stats.radiance_rays = h[0];
    c->stats.shadow_rays = h[1];
    c->stats.paths = h[2];
    c->stats.culled_rays = h_culled;
    c->stats.pixels = (uint32_t)(h[3] / ((unsigned long long)n_frames << a.chunk_shift));
    c->stats.sample_chunks = 1u << a.chunk_shift;
    c->stats.trav_wave_steps = h[4];
    c->stats.trav_lane_steps = h[5];
    c->stats.shade_wave_rounds = h[6];
    c->stats.shade_lane_rounds = h[7];
    c->stats.grid_blocks = grid;
    c->stats.variant = (uint32_t)variant;
    c->stats.math_mode = (uint32_t)(c->math_mode != 0 && ptd::render_variant_has_fast_math(variant) ? PT_MATH_FAST : PT_MATH_IEEE);
    c->stats.kernel_ms = ms;
    c->stats.launch_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

PT_API int pt_resolve_framebuffer(pt_ctx* c, const float* accum, uint8_t* fb, size_t n_pixels)
{
    if (!c || !accum || !fb) return fail(c, "pt_resolve_framebuffer: null argument");
    if (n_pixels > 0x7FFFFFFFull) return fail(c, "pt_resolve_framebuffer: too many pixels");
    CK(c, hipSetDevice(c->device));
    if (n_pixels) CK(c, ptd::launch_resolve((const float4*)accum, (uint32_t*)fb, (uint32_t)n_pixels, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

PT_API int pt_get_stats(pt_ctx* c, pt_stats* out)
{
    if (!c || !out) return fail(c, "pt_get_stats: null argument");
    *out = c->multi ? c->multi->group_stats : c->stats;
    return 0;
}

template <typename F>
static int trace_common(pt_ctx* c, const float* rays, size_t n, size_t out_bytes_a, void* out_a, size_t out_bytes_b, void* out_b, F launch)
{
    if (n == 0) return 0;
    if (n > 0x7FFFFFFFull) return fail(c, "pt_trace: too many rays");
    CK(c, hipSetDevice(c->device));
    float* d_rays = nullptr; void* d_a = nullptr; void* d_b = nullptr;
    int rc = 0;
    hipError_t e = hipMalloc((void**)&d_rays, n * 32);
    if (e == hipSuccess) e = hipMalloc(&d_a, out_bytes_a);
    if (e == hipSuccess && out_bytes_b) e = hipMalloc(&d_b, out_bytes_b);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, n * 32, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch(d_rays, d_a, d_b);
    if (e == hipSuccess) e = hipMemcpyAsync(out_a, d_a, out_bytes_a, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && out_bytes_b) e = hipMemcpyAsync(out_b, d_b, out_bytes_b, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, std::string("pt_trace: ") + hipGetErrorString(e));
    if (d_rays) (void)hipFree(d_rays);
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    return rc;
}

PT_API int pt_trace_closest(pt_ctx* c, const float* rays, size_t n, float* t_out, uint32_t* prim_out)
{
    if (!c || (n && (!rays || !t_out || !prim_out))) return fail(c, "pt_trace_closest: null argument");
    CK(c, hipSetDevice(c->device));
    if (int rc = ensure_node_format(c, 0)) return rc;          // the query kernels walk the fp32 nodes
    const ptd::DeviceScene sc = device_scene(c);
    const uint32_t se = c->stack_entries;
    hipStream_t s = c->stream;
    return trace_common(c, rays, n, n * 4, t_out, n * 4, prim_out, [&](float* d_rays, void* a, void* b) {
        return ptd::launch_trace_closest(sc, se, d_rays, (uint32_t)n, (float*)a, (uint32_t*)b, s);
    });
}

PT_API int pt_trace_any(pt_ctx* c, const float* rays, size_t n, uint8_t* hit_out)
{
    if (!c || (n && (!rays || !hit_out))) return fail(c, "pt_trace_any: null argument");
    CK(c, hipSetDevice(c->device));
    if (int rc = ensure_node_format(c, 0)) return rc;
    const ptd::DeviceScene sc = device_scene(c);
    const uint32_t se = c->stack_entries;
    hipStream_t s = c->stream;
    return trace_common(c, rays, n, n, hit_out, 0, nullptr, [&](float* d_rays, void* a, void*) {
        return ptd::launch_trace_any(sc, se, d_rays, (uint32_t)n, (uint8_t*)a, s);
    });
}

PT_API int pt_bench_traversal(pt_ctx* c, const float* rays, size_t n, int repeats, int node_format, float* t_out, uint32_t* prim_out, float* ms_out,
                              uint64_t* counters_out)
{
    if (!c || !rays || !t_out || !prim_out || !ms_out || n == 0 || n > 0x7FFFFFFFull || repeats < 1 || node_format < 0 || node_format > 4)
        return fail(c, "pt_bench_traversal: bad argument");
    CK(c, hipSetDevice(c->device));
    if (int rc = ensure_node_format(c, node_format == 1 ? 3 : node_format == 3 ? 9 : node_format == 4 ? 11 : 0)) return rc;       // stream formats 0 / 2: fp32 nodes, 3: fp16 {lo, hi}, 4: fp16 {centre, half extent}, 1: four-wide
    const uint32_t entries = node_format == 1 ? (c->bvh.wide_depth + 1u) : c->stack_entries;
    float* d_rays = nullptr; float* d_t = nullptr; uint32_t* d_p = nullptr; uint32_t* d_head = nullptr;
    int bpc = 0;
    hipError_t e = ptd::trace_stream_occupancy(node_format, entries, &bpc);
    if (e == hipSuccess && bpc < 1) e = hipErrorInvalidValue;
    if (e == hipSuccess) e = hipMalloc((void**)&d_rays, n * 32);
    if (e == hipSuccess) e = hipMalloc((void**)&d_t, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_p, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&d_head, 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_rays, rays, n * 32, hipMemcpyHostToDevice, c->stream);
    float best = 1e30f;
    const ptd::DeviceScene sc = device_scene(c);
    for (int r = 0; r < repeats && e == hipSuccess; r++) {
        e = hipMemsetAsync(d_head, 0, 4, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(c->d_counters, 0, 8 * sizeof(unsigned long long), c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev0, c->stream);
        if (e == hipSuccess) e = ptd::launch_trace_stream(node_format, sc, entries, d_rays, (uint32_t)n, d_head, d_t, d_p, c->d_counters, (uint32_t)(c->n_cus * bpc), c->stream);
        if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
        if (ms < best) best = ms;
    }
    if (e == hipSuccess) e = hipMemcpy(t_out, d_t, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(prim_out, d_p, n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && counters_out) e = hipMemcpy(counters_out, c->d_counters, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    if (d_rays) (void)hipFree(d_rays);
    if (d_t) (void)hipFree(d_t);
    if (d_p) (void)hipFree(d_p);
    if (d_head) (void)hipFree(d_head);
    if (e != hipSuccess) return fail(c, std::string("pt_bench_traversal: ") + hipGetErrorString(e));
    *ms_out = best;
    return 0;
}

// in / out sizes per element, in dwords (op 1: in = {seed, count}, out = 2 * count)
PT_API int pt_selftest(pt_ctx* c, int op, const void* in, size_t n, void* out)
{
    //                            0  1  2   3   4   5   6   7   8  9 10 11 12 13 14 15 16 17 18  19  20 .. 29 unused           30 31 32 33 34 35 36 37 38  39  40  41
    static const int in_dw[42] = {2, 2, 3, 10, 10, 10, 10, 10, 10, 7, 4, 1, 6, 4, 2, 2, 6, 7, 3, 17, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 1, 6, 4, 2, 2, 6, 7, 3, 20, 17, 19},
                     out_dw[42] = {1, 0, 1, 3, 3, 3, 3, 3, 3, 4, 2, 4, 3, 3, 3, 3, 3, 3, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 3, 3, 3, 3, 3, 3, 1, 3, 3, 3};
    if (!c || !in || !out || op < 0 || op > 41 || in_dw[op] == 0 || n == 0 || n > (1u << 24)) return fail(c, "pt_selftest: bad argument");
    CK(c, hipSetDevice(c->device));
    size_t in_bytes = n * (size_t)in_dw[op] * 4, out_bytes = n * (size_t)out_dw[op] * 4;
    uint32_t launch_n = (uint32_t)n;
    if (op == 1) {
        const uint32_t count = ((const uint32_t*)in)[1];
        if (n != 1 || count == 0 || count > (1u << 22)) return fail(c, "pt_selftest: op 1 takes one {seed, count} record");
        in_bytes = 8; out_bytes = (size_t)count * 8; launch_n = 1;
    }
    uint32_t* d_in = nullptr; uint32_t* d_out = nullptr;
    hipError_t e = hipMalloc((void**)&d_in, in_bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, out_bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, in, in_bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = ptd::launch_selftest(op, d_in, launch_n, d_out, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, std::string("pt_selftest: ") + hipGetErrorString(e));
    return 0;
}

PT_API int pt_debug_wave_times(pt_ctx* c, uint64_t* out, size_t max_waves)
{
    if (!c || !out) return fail(c, "pt_debug_wave_times: null argument");
    if (max_waves > ptd::kMaxTimedWaves) max_waves = ptd::kMaxTimedWaves;
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpy(out, c->d_counters + 8, 3 * max_waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

PT_API int pt_debug_queue_progress(pt_ctx* c, uint64_t* out)
{
    if (!c || !out) return fail(c, "pt_debug_queue_progress: null argument");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpy(out, c->d_counters + 8 + 3 * (size_t)ptd::kMaxTimedWaves, 2056 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

PT_API int pt_debug_queue_order(pt_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 3) return fail(c, "pt_debug_queue_order: 0 = contiguous eighths of the tile order per shard; round robin over the shards in units of 1 = a tile-strip row (default), 2 = a tile; 3 = one queue in image order");
    c->queue_order = mode;
    if (c->multi) for (size_t i = 1; i < c->multi->ranks.size(); i++) c->multi->ranks[i]->queue_order = mode;
    return 0;
}

// The host side of the pixel classes alone (no context, no GPU): out = 2 * params->height words, {outer lo | hi << 16, inner lo | hi << 16}
// per image row.  Returns 0 when the spans could be computed, 1 when they could not (box not entirely in front of the eye, ...).
PT_API int pt_debug_row_spans(const pt_params* p, const float* box_lo, const float* box_hi, uint32_t* out)
{
    if (!p || !box_lo || !box_hi || !out || p->width == 0 || p->height == 0 || p->width > 65535u || p->height > 32767u) return 2;
    std::vector<uint32_t> spans;
    const bool ok = row_spans(p, box_lo, box_hi, spans);
    memcpy(out, spans.data(), spans.size() * sizeof(uint32_t));
    return ok ? 0 : 1;
}

PT_API int pt_debug_pixel_classes(pt_ctx* c, int on)
{
    if (!c) return fail(c, "pt_debug_pixel_classes: null context");
    c->pixel_classes = on ? 1 : 0;
    if (c->multi) for (size_t i = 1; i < c->multi->ranks.size(); i++) c->multi->ranks[i]->pixel_classes = c->pixel_classes;
    return 0;
}

PT_API int pt_debug_window_moves(pt_ctx* c, uint64_t* out)
{
    if (!c || !out) return fail(c, "pt_debug_window_moves: null argument");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpy(out, c->d_counters + ptd::kWindowMoves, sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}

#ifdef ACGPT_EXPERIMENTS
// experiments build only: renumber the resident fp16 nodes (0: the build's order, 1: sibling pairs share a 64-byte line, 2: depth first)
namespace ptd { bool reorder_records_dfs(LbvhResult& r, hipStream_t stream, std::string& err); }      // lbvh_experiments.inc
PT_API int pt_debug_node_order(pt_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 3) return fail(c, "pt_debug_node_order: mode 0, 1, 2 or 3");
    CK(c, hipSetDevice(c->device));
    CK(c, hipStreamSynchronize(c->stream));
    std::string err;
    if (mode == 3) {        // the triangle records in depth-first leaf order (in place; the nodes stay as they are)
        if (!ptd::reorder_records_dfs(c->bvh, c->stream, err)) return fail(c, "pt_debug_node_order: " + err);
        return 0;
    }
    if (!ptd::reorder_hcnodes(c->bvh, mode, c->stream, err)) return fail(c, "pt_debug_node_order: " + err);
    return 0;
}

// experiments build only: per-role times of the last launch of a wavefront kernel (render_wavefront.hip), 17 values (tools/wf_check.py)
PT_API int pt_debug_wf(pt_ctx* c, uint64_t* out)
{
    if (!c || !out) return fail(c, "pt_debug_wf: null argument");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpy(out, c->d_counters + ptd::kWfDiag, 17 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return 0;
}
#endif

PT_API int pt_read_morton(pt_ctx* c, uint32_t* codes_sorted, uint32_t* prims_sorted)
{
    if (!c || !codes_sorted || !prims_sorted) return fail(c, "pt_read_morton: null argument");
    if (c->bvh.n_tris == 0) return 0;
    CK(c, hipSetDevice(c->device));
    std::string err;
    if (!ptd::read_morton(c->bvh, c->stream, codes_sorted, prims_sorted, err)) return fail(c, "pt_read_morton: " + err);
    return 0;
}

PT_API int pt_device_malloc(pt_ctx* c, void** out, size_t bytes)
{
    if (!c || !out) return fail(c, "pt_device_malloc: null argument");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMalloc(out, bytes ? bytes : 4));
    return 0;
}
PT_API int pt_device_free(pt_ctx* c, void* ptr)
{
    if (!c) return fail(nullptr, "pt_device_free: null context");
    CK(c, hipSetDevice(c->device));
    CK(c, hipFree(ptr));
    return 0;
}
PT_API int pt_device_memset(pt_ctx* c, void* ptr, int value, size_t bytes)
{
    if (!c) return fail(nullptr, "pt_device_memset: null context");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemsetAsync(ptr, value, bytes, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}
PT_API int pt_copy_to_host(pt_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return fail(nullptr, "pt_copy_to_host: null context");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}
PT_API int pt_copy_to_device(pt_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c) return fail(nullptr, "pt_copy_to_device: null context");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}
PT_API int pt_host_malloc_mapped(pt_ctx* c, void** host_out, void** device_out, size_t bytes)
{
    if (!c || !host_out || !device_out) return fail(c, "pt_host_malloc_mapped: null argument");
    CK(c, hipSetDevice(c->device));
    CK(c, hipHostMalloc(host_out, bytes ? bytes : 4, hipHostMallocMapped | hipHostMallocPortable));
    CK(c, hipHostGetDevicePointer(device_out, *host_out, 0));
    return 0;
}
PT_API int pt_host_free_mapped(pt_ctx* c, void* host_ptr)
{
    if (!c) return fail(nullptr, "pt_host_free_mapped: null context");
    CK(c, hipHostFree(host_ptr));
    return 0;
}
