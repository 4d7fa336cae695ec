// render_megakernel.hip — the per-pixel Monte-Carlo launch as ONE persistent gfx950 kernel.
//
// Replaces __raygen__rg / __closesthit__diffuse__ch / __miss__ms
// (PathTracer_Optix/pathTracerPrograms.cu:707-816, 866-1031, 833-847) and the OptiX
// traversal underneath them (:600-613, :660-671).
//
// Kernels of one pt_launch, in stream order:
//   k_render_pw  the persistent megakernel (default; k_render = segment-synchronous baseline)
//   k_finalize   blends the launch's sub-frame(s) into the accumulation buffer in frame order, make_color
//
// Scheduling (wave64, persistent):
//   * the grid is sized to the chip (CUs x resident workgroups), never to the image;
//   * one lane owns one work item and walks its samples in the reference's order as a FLAT loop;
//     a lane whose path ends regenerates the next camera path right away, a lane whose item is
//     finished takes the next one from the queue (ballot of the idle lanes, one atomicAdd by the
//     first of them, popcount-prefix hand-out, wave-local reserve); 8 queue shards, one per XCD; an item is
//     decoded from its index alone (tile order -> pixel, tea<4> seed, LCG skip-ahead): no table in memory;
//   * the runs a pixel's samples are cut into are summed in run order by the wave that was granted them
//     (fold slots, below): no per-run buffer in memory either;
//   * a camera ray that misses the scene's bounding box ends its path on the spot;
//   * radiance and shadow rays share one BVH loop; lanes with a finished ray park until enough of
//     them are waiting, then that batch is shaded and re-armed (k_render_pw);
//   * the traversal stack lives in LDS, entry-major (pt_device.h), sized from the measured tree
//     height, with its top element cached in a register.
// Item order inside the queue is the 8x4-tile order of sutil/WorkDistribution.h:60-81 for
// (rank, world), which is also the multi-GPU partition.
#include "render_common.h"

namespace ptd {

// frame batches: for every pixel of this rank, blend the sub-frames' sums into the accumulation buffer in frame order —
// the arithmetic n_frames separate launches would do.  One thread per pixel slot of the rank's tile order.
constexpr uint32_t kFinThreads = 256;
__global__ void __launch_bounds__(kFinThreads) k_finalize(const RenderArgs A)
{
    const uint32_t slot = blockIdx.x * kFinThreads + threadIdx.x;
    if (slot >= (A.total_samples >> A.sub_shift)) return;
    int x, y;
    sample_pixel(A.world, (int)A.width, A.rank, (int)slot, x, y);
    if ((uint32_t)x >= A.width || (uint32_t)y >= A.height) return;
    const uint32_t pix = (uint32_t)y * A.width + (uint32_t)x;
    f3 accum = mk(0.0f);
    if (A.frame > 0u) { const float4 q = A.accum[pix]; accum = mk(q.x, q.y, q.z); }
    const float4* row = A.frame_sums + (size_t)slot * A.n_frames;
    for (uint32_t f = 0; f < A.n_frames; f++) {
        const float4 v = row[f];
        accum = blend_frame(accum, mk(v.x, v.y, v.z), A.spp, A.frame + f);
    }
    A.accum[pix] = make_float4(accum.x, accum.y, accum.z, 1.0f);
    if (A.fb) A.fb[pix] = make_color(accum);
}

// =================================================================================================
// Variant 0: segment-synchronous.  Every iteration: all live lanes trace one radiance segment to
// completion, shade, (some) trace a shadow ray, account.  Simple; lanes wait for the slowest ray.
// =================================================================================================
template <int MATH>
__global__ void __launch_bounds__(kRenderThreads)
k_render(const RenderArgsBox B)
{
    constexpr int FM = MATH ? 2 : 0;                  // arithmetic level of the shading code (pt_device.h)
    const RenderArgs& A = B.a[0];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    LaneStack st;
    st.base = lds_dyn + wave * (A.stack_entries * 64u) + lane;
    uint32_t* const lcg_skip = lds_dyn + (kRenderThreads / 64) * (A.stack_entries * 64u);      // 64 dwords behind the stacks
    if (threadIdx.x < 32u) { lcg_skip[2u * threadIdx.x] = A.lcg_mul[threadIdx.x]; lcg_skip[2u * threadIdx.x + 1u] = A.lcg_add[threadIdx.x]; }
    const WaveBook book = wave_book(lcg_skip + 64u + wave * kBookDwords, lane);
    __syncthreads();
    const DeviceScene sc = A.scene;
    const auto late = [&]() -> const RenderArgs& { return A; };
    const f3 eye = mk(A.eye), camU = mk(A.U), camV = mk(A.V), camW = mk(A.W);
    const float fw = (float)(int)A.width, fh = (float)(int)A.height;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    QueueState q; q.shard = A.row_interleave == 3u ? 0u : xcc_id(); q.shards_left = 8; q.res_first = 0; q.res_count = 0; q.grant_g0 = 0; q.grp_pxy = 0xFFFFFFFFu; q.grp_seed = 0; q.skipped = 0; q.free_top = kFoldSlots;
    unsigned long long n_radiance = 0, n_shadow = 0, n_paths = 0, n_pixels = 0, n_culled = 0;
    float* const scratch = A.wave_scratch + 3u * (size_t)(blockIdx.x * (kRenderThreads / 64) + wave) * ((size_t)kFoldSlots << A.chunk_shift);

    LanePixel lp; lp.alive = false; lp.new_path = false; lp.pxy = lp.seed = lp.samples_left = lp.tag = 0; lp.result = mk(0.0f);
    uint32_t pseed = 0;
    int depth = 0;
    f3 org = mk(0.0f), dir = mk(0.0f, 0.0f, 1.0f), att = mk(1.0f);

    for (;;) {
        refill_lanes(A, late, q, lane, below, lp, lcg_skip, book);
        if (q.skipped != 0u) {
            const unsigned long long n = (unsigned long long)q.skipped * A.spp;
            n_radiance += n; n_paths += n; n_culled += n;
            n_pixels += (unsigned long long)q.skipped << A.chunk_shift;
            q.skipped = 0u;
        }
        const unsigned long long live = vote(lp.alive);
        if (live == 0ull) { if (q.shards_left == 0u && q.res_count == 0u) break; else continue; }

        if (lp.alive && lp.new_path) {                                // camera path start, :727-745
            const float jx = rnd(lp.seed);
            const float jy = rnd(lp.seed);
            dir = m_normalize<FM>(camera_dir<FM>((float)(lp.pxy & 0xFFFFu), (float)(lp.pxy >> 16), jx, jy, fw, fh, camU, camV, camW));
            org = eye;
            att = mk(1.0f);
            pseed = lp.seed;
            depth = 0;
            lp.new_path = false;
        }

        HitRec hit;                                                   // radiance segment (:750-757)
        traverse<false>(sc, st, lp.alive, org, dir, 0.01f, 1e16f, hit);
        n_radiance += (unsigned long long)popc(live);

        f3 emission = mk(0.0f), P = mk(0.0f), L = mk(0.0f);
        float Ldist = 0.0f;
        Pending pd; pd.nxt_org = org; pd.nxt_dir = dir; pd.radiance = mk(0.0f); pd.weight = 0.0f;
        pd.done = true;                                               // __miss__ms :833-847
        bool want_shadow = false;
        if (lp.alive && hit.slot >= 0)
            want_shadow = shade_hit<FM>(sc, late, org, dir, hit.t, hit.slot, depth, pseed, att, emission, pd, P, L, Ldist);

        const unsigned long long shadow_mask = vote(want_shadow);
        if (shadow_mask != 0ull) {                                    // traceOcclusion :651-684
            HitRec sh;
            const bool occluded = traverse<true>(sc, st, want_shadow, P, L, 0.01f, Ldist - 0.01f, sh);
            n_shadow += (unsigned long long)popc(shadow_mask);
            if (want_shadow && !occluded) pd.radiance = m_madd<FM>(mk(A.light.emission), pd.weight, pd.radiance);
        }

        bool end = false, finished = false;
        if (lp.alive) {                                               // raygen :760-778
            lp.result += emission;
            add_segment<FM>(lp.result, pd.radiance, att);
            const float p = roulette_p<FM>(att);
            const bool rr = rnd(pseed) > p;
            end = pd.done || rr || (uint32_t)depth >= A.maxDepth;
            if (!end) {
                att = roulette_scale<FM>(att, p);
                org = pd.nxt_org;
                dir = pd.nxt_dir;
                ++depth;
            } else {
                lp.samples_left--;
                lp.new_path = true;
                if (lp.samples_left == 0u) { lp.alive = false; finished = true; }
            }
        }
        n_paths += (unsigned long long)popc(vote(end));
        n_pixels += (unsigned long long)popc(vote(finished));
        finish_runs(A, q, below, lp, finished, book, scratch);
    }
    if (lane == 0) {
        atomicAdd(&A.counters[0], n_radiance);
        atomicAdd(&A.counters[1], n_shadow);
        atomicAdd(&A.counters[2], n_paths);
        atomicAdd(&A.counters[3], n_pixels);
        if (n_culled) atomicAdd(&A.counters[kCulledCounter], n_culled);
    }
}

// Per-ray constants of the slab test for node format NODE_FMT (see k_render_pw): rinv multiplies a stored plane, gro is added.
// NODE_FMT 8 keeps, per axis, a rotate amount (0 or 16): the packed fp16 pair {lo, hi} of a node is rotated so that the
// low half is the plane the ray meets first.  Near and far planes then need no per-axis min / max.
struct AxisRot { uint32_t x, y, z; };
__device__ __forceinline__ AxisRot axis_rot(const f3& rinv)
{
    AxisRot r;
    r.x = rinv.x < 0.0f ? 16u : 0u; r.y = rinv.y < 0.0f ? 16u : 0u; r.z = rinv.z < 0.0f ? 16u : 0u;
    return r;
}

template <int NODE_FMT>
__device__ __forceinline__ void setup_ray(const f3& ro, const f3& rd, const QGrid& G, const HSpace& HS, f3& rinv, f3& gro)
{
    if (NODE_FMT == 0 || NODE_FMT == 6) {            // t = p * (1/d) + (-o/d)
        rinv = mk(finite_rcp(rd.x), finite_rcp(rd.y), finite_rcp(rd.z));
        gro = mk(-(ro.x * rinv.x), -(ro.y * rinv.y), -(ro.z * rinv.z));
    } else if (NODE_FMT == 9) {                      // the same with the rotate amounts in the multipliers (pt_device.h)
        setup_ray_h9(ro, rd, HS, rinv, gro);
    } else if (NODE_FMT == 11 || NODE_FMT == 13 || NODE_FMT == 14) {   // fp16 centre / half-extent nodes: the plain multiplier and addend
        setup_ray_hc(ro, rd, HS, rinv, gro);
    } else if (NODE_FMT == 7 || NODE_FMT == 8) {     // t = g * (1/d / scale) + (centre - o)/d, g = the fp16 plane
        const f3 r = mk(finite_rcp(rd.x), finite_rcp(rd.y), finite_rcp(rd.z));
        gro = mk((HS.cx - ro.x) * r.x, (HS.cy - ro.y) * r.y, (HS.cz - ro.z) * r.z);
        rinv = r * HS.inv_scale;
    } else {
        rinv = mk(fast_rcp(rd.x), fast_rcp(rd.y), fast_rcp(rd.z));
        if (NODE_FMT == 1 || NODE_FMT == 2 || NODE_FMT == 4) {
            gro = mk((ro.x - G.ox) * G.icx, (ro.y - G.oy) * G.icy, (ro.z - G.oz) * G.icz);
            rinv = mk(G.cx * rinv.x, G.cy * rinv.y, G.cz * rinv.z);
            if (NODE_FMT == 4) gro = mk(-(gro.x * rinv.x), -(gro.y * rinv.y), -(gro.z * rinv.z));   // t = q * rinv + gro
        }
    }
}

// =================================================================================================
// Variant 1: persistent traversal with deferred shading.  The BVH loop never waits for the slowest
// ray: a lane whose ray is finished parks; once SHADE_K lanes are parked (or nothing is left to
// traverse) the wave leaves the traversal loop, shades exactly those lanes — closest-hit, shadow
// resolve, roulette, next camera path or next pixel from the queue — gives each of them a new ray
// and re-enters the loop.  Radiance and shadow rays share the one traversal loop (a per-lane
// any-hit flag).  Each lane still performs its own operations in the reference's order, so the
// image is bit-identical to variant 0; only the interleaving between lanes changes.
// LEAF_K: triangle tests run when at least LEAF_K lanes sit at a leaf, or no lane has an inner node.
// =================================================================================================
// NODE_FMT: 0 = fp32 boxes, 64-byte nodes in global memory (4 x 16-byte loads per visit), slab test as one fma per plane
//           7 = fp16 boxes in a scene-centred space, 32-byte nodes (2 loads per visit), the same fma count: each plane is a
//               v_fma_mix_f32 reading the fp16 half in place (pt_device.h HNode), near / far by per-axis min / max
//           8 = the same nodes; each packed {lo, hi} pair is rotated by 0 or 16 bits first (v_alignbit_b32, per ray and axis),
//               so the low half is the near plane: 6 rotates replace 12 min / max; three registers of rotate amounts
//           9 = the same, with the rotate amount in the five lowest mantissa bits of the plane multiplier (setup_ray): no
//               register for it — the default of rounds 2-3
//          11 = the same 32-byte nodes holding {centre, half extent} per axis: near / far = c_t -+ h_t by a full-rate subtract / add, no
//               rotates, child references of inner nodes as byte offsets (round 4: -2 % on every configuration, same bits) — the default
//          10 / 12 = shared-plane records (experiments): 16-byte nodes, one gather per visit, the ray's interval carried on the stack
//           5 = the fp32 nodes, slab test as subtract + multiply per plane
//           4 = 16-bit grid nodes with the fma form
//           1 = 16-bit grid boxes, 32-byte nodes in global memory (2 loads per visit)
//           2 = the same 32-byte nodes staged into LDS by each workgroup (scenes whose node array
//               fits beside the lane stacks; 1024-thread workgroups so one copy serves 16 waves)
//           3 = four-wide tree, 8-bit child boxes, 48-byte records shared with the triangles (wide_bvh.hip):
//               3 loads per visit and about half the visits; the stack holds {base, child list} groups
// DIAG: 1 = 12 extra dependent VALU per inner step, 2 = two extra 16-byte loads per inner step (timing experiments only, never
// a product variant); 3 = arithmetic level 1 of the shading code (pt_device.h, FM: the cosine sampler's trigonometry in hardware).
// MATH (pt_set_math_mode): 0 = IEEE arithmetic in the shading code, the level the CPU oracle is written at; 1 = the arithmetic of
// the reference's own build (nvcc --use_fast_math, CMakeLists.txt:267; level 2 of pt_device.h).  Traversal and triangle test are
// the same in both: the rays that are traced for a given path prefix, and what they hit, do not depend on it.
// INNER: 0 = stack entirely in LDS, nested branches; 1 = stack top cached in a register (the LDS read of
// a pop is consumed one push/pop later, off the critical path) and child selection by selects; 2, 3 = the same with
// that many node visits per trip through the loop control.
template <int SHADE_K, int LEAF_K, int NODE_FMT, int THREADS, int MINW, bool STATS, int DIAG = 0, int INNER = 0, int LEAVES = 1, bool LIGHTS = false, int STACK_CAP = 0, int TOPN = 0, int MATH = 0>
__global__ void __launch_bounds__(THREADS, MINW)
k_render_pw(const RenderArgsBox B)
{
    const RenderArgs& A = B.a[0];                     // what the BVH loop, the queue and the wave set-up use: read once
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // STACK_CAP > 0: only the first STACK_CAP entries of a lane's stack live in LDS (so that a deep tree does not cost a
    // resident workgroup); the few rays that ever hold more pending nodes keep the rest in global memory (A.stack_overflow).
    // STACK_CAP < 0: a SLIDING WINDOW of -STACK_CAP entries (a power of two) in LDS, addressed circularly: slot k of a lane's
    // stack sits at position k mod window; slots below `wbase` have been moved to global memory.  Pushes and pops inside the
    // BVH loop are plain circular LDS accesses — no comparison, no branch; once per trip a lane whose stack pointer has come
    // within a trip's reach of either end of the window moves four entries out or back in (rare: rays seldom hold more than a
    // dozen pending nodes).  Any tree depth at a fixed LDS cost, and none of the per-access test that made the capped
    // kernel 10 % slower than the plain one on the same tree.
    constexpr bool WINDOW = STACK_CAP < 0;
    constexpr int FM = MATH ? 2 : (DIAG == 3 ? 1 : 0);         // arithmetic level of the shade phase
    // Origin-triangle release (experiment, DIAG 5): a bounce or shadow ray starts ON the triangle its path has just hit and inside
    // that triangle's box, so front-to-back traversal leads it to that leaf first, where it waits for a triangle round only to
    // fail on t < tmin.  When cos(theta_out) * tmin exceeds the rounding between hit point and plane (RenderArgs::skip_base),
    // Moeller-Trumbore cannot accept the origin triangle: a lane found sitting at that leaf when a trip ends moves on at once.
    constexpr bool SKIP = DIAG == 5 && INNER >= 1 && !LIGHTS;
    constexpr int WIN = WINDOW ? -STACK_CAP : 0;
    static_assert(!WINDOW || ((WIN & (WIN - 1)) == 0 && WIN >= 16), "the window wraps by masking and must hold two trips");
    const uint32_t lds_entries = WINDOW ? (uint32_t)WIN + 1u : ((STACK_CAP > 0 && A.stack_entries > (uint32_t)STACK_CAP) ? (uint32_t)STACK_CAP : A.stack_entries);   // WINDOW: entry WIN of a lane's column holds its window base
    const bool deep = WINDOW && A.stack_entries > (uint32_t)WIN;      // wave-uniform: can a stack outgrow the window at all?
    constexpr bool SHARED = NODE_FMT == 10 || NODE_FMT == 12;     // shared-plane records, 15-bit / 30-bit child references
    constexpr uint32_t ENT = SHARED ? 2u : 1u;      // dwords per stack entry: the shared-plane kernel keeps {node, interval}
    LaneStack st;
    st.base = lds_dyn + wave * (lds_entries * 64u * ENT) + lane;
    // the overflow region of this wave: a wave-uniform base (scalar registers) and, where an entry is touched, a 32-bit
    // offset from the entry number and the lane — a per-lane 64-bit pointer held across the kernel cost two vector registers
    // and, at the 96 of five waves per SIMD, spills whose scratch traffic was the 18 GB of fabric writes of round 2's profile
    uint32_t* const ovf = STACK_CAP != 0
        ? A.stack_overflow + (size_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (THREADS / 64) + wave)) * 64u * ENT * (WINDOW ? A.stack_entries : A.stack_entries - lds_entries) : nullptr;
    const auto push = [&](int at, int v) {
        if (WINDOW) st.push(at & (WIN - 1), v);
        else if (STACK_CAP == 0 || at < (int)lds_entries) st.push(at, v);
        else ovf[(uint32_t)(at - (int)lds_entries) * 64u + lane] = (uint32_t)v;
    };
    const auto pop = [&](int at) -> int {
        if (WINDOW) return st.pop(at & (WIN - 1));
        if (STACK_CAP == 0 || at < (int)lds_entries) return st.pop(at);
        return (int)ovf[(uint32_t)(at - (int)lds_entries) * 64u + lane];
    };
    LaneStack2 st2;                                   // NODE_FMT 3: the same LDS region as stack_entries / 2 groups; NODE_FMT 10: lds_entries 8-byte entries
    st2.base = (uint2*)(lds_dyn + wave * (lds_entries * 64u * ENT)) + lane;
    uint2* const ovf2 = (uint2*)ovf;                  // NODE_FMT 10: the overflow region as 8-byte entries
    DeviceScene sc = A.scene;
    if (NODE_FMT == 3) sc.tris = (const TriRecord*)A.scene.wrecs;      // triangles live in the record array
    const uint2* lds_nodes = (const uint2*)(lds_dyn + (THREADS / 64) * (lds_entries * 64u * ENT));
    if (NODE_FMT == 2) {
        uint4* dst = (uint4*)(lds_dyn + (THREADS / 64) * (lds_entries * 64u));
        const uint4* src = (const uint4*)sc.qnodes;
        for (uint32_t i = threadIdx.x; i < A.n_lds_nodes * 2u; i += THREADS) dst[i] = src[i];
        __syncthreads();
    }
    if (NODE_FMT == 13) {      // experiment: the FIRST 16 bytes of every fp16 node (child 0) staged in LDS, 16 bytes apart; child 1 still comes through the texture path
        uint4* dst = (uint4*)(lds_dyn + (THREADS / 64) * (lds_entries * 64u));
        const uint4* src = (const uint4*)sc.hcnodes;
        for (uint32_t i = threadIdx.x; i < A.n_lds_nodes; i += THREADS) dst[i] = src[2u * i];
        __syncthreads();
    }
    if (NODE_FMT == 14) {      // ... and the whole nodes (32 bytes apart, as in global memory): every node gather from LDS, the texture path sees triangles only
        uint4* dst = (uint4*)(lds_dyn + (THREADS / 64) * (lds_entries * 64u));
        const uint4* src = (const uint4*)sc.hcnodes;
        for (uint32_t i = threadIdx.x; i < A.n_lds_nodes * 2u; i += THREADS) dst[i] = src[i];
        __syncthreads();
    }
    // LCG skip-ahead table behind the stacks (and behind the LDS-staged nodes of NODE_FMT 2 / 13)
    uint32_t* const lcg_skip = lds_dyn + (THREADS / 64) * (lds_entries * 64u * ENT) + (NODE_FMT == 2 || NODE_FMT == 14 ? A.n_lds_nodes * 8u : NODE_FMT == 13 ? A.n_lds_nodes * 4u : 0u);
    if (threadIdx.x < 32u) { lcg_skip[2u * threadIdx.x] = A.lcg_mul[threadIdx.x]; lcg_skip[2u * threadIdx.x + 1u] = A.lcg_add[threadIdx.x]; }
    const WaveBook book = wave_book(lcg_skip + 64u + wave * kBookDwords, lane);
    // TOPN > 0 (experiment): the first TOPN nodes of the tree, breadth first, staged in LDS behind the books — every ray walks them;
    // a node reference with kTopNodeFlag is a position in that copy
    const uint4* const top_lds = (const uint4*)(lcg_skip + 64u + (THREADS / 64) * kBookDwords);
    if (TOPN > 0) {
        uint4* dst = (uint4*)(lcg_skip + 64u + (THREADS / 64) * kBookDwords);
        const uint4* src = (const uint4*)A.scene.top;
        const uint32_t n = (A.scene.n_top < (uint32_t)TOPN ? A.scene.n_top : (uint32_t)TOPN) * 2u;
        const uint32_t* ids = (const uint32_t*)(A.scene.top + kTopNodesMax);        // position -> index in hnodes
        for (uint32_t i = threadIdx.x; i < n; i += THREADS) {
            uint4 v = src[i];
            if ((int)v.w >= 0 && (v.w & kTopNodeFlag) && (v.w & 0xFFFFu) >= (uint32_t)TOPN) v.w = ids[v.w & 0xFFFFu];     // a child past this kernel's cut: back to its index in hnodes
            dst[i] = v;
        }
    }
    __syncthreads();
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int root = sc.n_tris ? ((TOPN > 0 && NODE_FMT == 9) ? (int)kTopNodeFlag : 0) : kSentinel;

    QueueState q; q.shard = A.row_interleave == 3u ? 0u : xcc_id(); q.shards_left = 8; q.res_first = 0; q.res_count = 0; q.grant_g0 = 0; q.grp_pxy = 0xFFFFFFFFu; q.grp_seed = 0; q.skipped = 0; q.free_top = kFoldSlots;
    float* const scratch = A.wave_scratch + 3u * (size_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (THREADS / 64) + wave)) * ((size_t)kFoldSlots << A.chunk_shift);   // wave-uniform: scalar registers
    unsigned long long n_radiance = 0, n_shadow = 0, n_paths = 0, n_pixels = 0, n_culled = 0;
    unsigned long long n_steps = 0, n_lane_steps = 0, n_rounds = 0, n_lane_rounds = 0;
    uint32_t n_moves = 0;                             // WINDOW: times this wave moved stack entries out of / back into the LDS window
    // STATS only: 100 MHz stamps of this wave's start, of the moment it found the queue empty, and of its end
    unsigned long long t_start = 0, t_drain = 0, t_phase = 0, t_in_shade = 0, t_refill = 0, t_finish = 0, t_newpath = 0, t_mark = 0;
    if (STATS) { t_start = __builtin_amdgcn_s_memrealtime(); t_phase = t_start; }

    LanePixel lp; lp.alive = false; lp.new_path = false; lp.pxy = lp.seed = lp.samples_left = lp.tag = 0; lp.result = mk(0.0f);
    uint32_t pseed = 0;
    int depth = 0;
    f3 att = mk(1.0f);
    // ray in flight (rinv / gro: reciprocal direction and origin, in grid space for quantised nodes)
    f3 ro = mk(0.0f), rd = mk(0.0f, 0.0f, 1.0f), rinv = mk(1.0f), gro = mk(0.0f);
    f3 gfar = mk(0.0f);                               // NODE_FMT 10 only: the root's far plane distances (gro: its near plane distances; rinv: |1 / d| / scale)
    float cur_tn = 0.0f, cur_tf = 0.0f;               // NODE_FMT 10 only: the ray's interval in the box of `node`
    uint32_t tos_iv = 0u;                             // ... and, packed as two fp16, in the box of the stack's top element
    AxisRot rot = {0u, 0u, 0u};                       // NODE_FMT 8 only
    constexpr float rtmin = 0.01f;      // both ray kinds start at 0.01 (:750-757 and :660-672): a literal, not a register
    float rtmax = 0.0f, best_t = 0.0f;
    int best_slot = -1; uint32_t best_prim = 0xFFFFFFFFu;
    int node = kSentinel, sp = 0, tos = kSentinel;
    uint32_t cur_base = 0, cur_list = 0;              // NODE_FMT 3: innermost group of pending children
    bool shadow_ray = false, shadow_hit = false;
    float prev_pdf = 0.0f;                            // LIGHTS (light mode 1) only: pdf of the last sampled direction where a light sample was taken
    bool fin_pending = false;                         // ran out of samples inside the camera cull: its run is finished at the next round's start
    // What the closest-hit left for after the shadow ray (Pending), held while that ray is in flight.  Light mode 0 keeps four values
    // instead of ten: the next bounce's direction (or, for a path that ends on an emitter, the emitter's Ke, which the light
    // sample is added to, :992-1000, 1015-1024) and the light sample's weight; the next bounce's origin is the shadow ray's own
    // origin P (diffuse) or P + R * 1e-4 (conductor, :948) and is recomputed with the closest-hit's operations.
    Pending pd_lights; pd_lights.nxt_org = mk(0.0f); pd_lights.nxt_dir = mk(0.0f, 0.0f, 1.0f); pd_lights.radiance = mk(0.0f); pd_lights.weight = 0.0f; pd_lights.done = true;   // LIGHTS only: the whole record
    f3 keep_dir = mk(0.0f, 0.0f, 1.0f); float keep_weight = 0.0f; bool keep_done = true, keep_metal = false;
    int origin_ref = kSentinel;                       // SKIP: leaf reference of the triangle the ray in flight starts on
    bool skip_now = false, keep_skip = false;         // ... may the ray in flight / the bounce after the shadow ray pass it by

    for (;;) {
        // =========================== shade / regenerate: lanes with no ray in flight ===============
        // launch constants the shade phase needs are read at their points of use (see RenderArgsBox above)
        const auto late = [&]() -> const RenderArgs& { return B.a[opaque_zero()]; };
        if (STATS) { n_rounds += 1; n_lane_rounds += (unsigned long long)popc(vote(lp.alive && node == kSentinel)); t_phase = __builtin_amdgcn_s_memrealtime(); }
        bool segment_done = false, started_shadow = false;
        bool skip_bounce = false;                                     // SKIP: may the bounce that starts in this round pass its origin triangle by
        f3 emission = mk(0.0f);
        Pending pd;                                                   // lives within one shade round (light mode 1: carried in pd_lights)
        if (LIGHTS) pd = pd_lights;
        else { pd.nxt_org = mk(0.0f); pd.nxt_dir = mk(0.0f, 0.0f, 1.0f); pd.radiance = mk(0.0f); pd.weight = 0.0f; pd.done = true; }
        if (lp.alive && node == kSentinel) {
            if (shadow_ray) {                                         // shadow ray back (:1015-1024)
                if (LIGHTS) { if (shadow_hit) pd.radiance = mk(0.0f); }                // the light sample parked there counts only unoccluded
                else {
                    pd.done = keep_done; pd.weight = keep_weight;
                    pd.nxt_dir = keep_dir;
                    pd.nxt_org = keep_metal ? ro + keep_dir * 1e-4f : ro;             // ro is the shadow ray's origin P
                    pd.radiance = keep_done ? keep_dir : mk(0.0f);
                    if (!shadow_hit) pd.radiance = m_madd<FM>(mk(late().light.emission), pd.weight, pd.radiance);
                }
                shadow_ray = false;
                segment_done = true;
                skip_bounce = keep_skip;
            } else {                                                  // radiance ray back
                bool want_shadow = false;
                f3 P, L; float Ldist;
                if (best_slot >= 0) {
                    if (LIGHTS) want_shadow = shade_hit_lights<FM>(sc, late, ro, rd, best_t, best_slot, depth, pseed, att, prev_pdf, pd, P, L, Ldist);
                    else want_shadow = shade_hit<FM, NODE_FMT == 3>(sc, late, ro, rd, best_t, best_slot, depth, pseed, att, emission, pd, P, L, Ldist);
                } else {                                              // __miss__ms :833-847
                    pd.radiance = mk(0.0f); pd.weight = 0.0f; pd.done = true;
                }
                lp.result += emission;                                // :760 (before the radiance term)
                if (SKIP) {
                    // the rays that start at this hit point: cos(theta_out) * tmin against the rounding between P and the triangle's plane
                    const float bound = late().skip_base + kOriginEps * best_t;
                    origin_ref = best_slot >= 0 ? ~best_slot : kSentinel;
                    skip_bounce = best_slot >= 0 && pd.cos_bounce * 0.01f > bound;
                    if (want_shadow) { skip_now = pd.cos_shadow * 0.01f > bound; keep_skip = skip_bounce; }
                }
                if (want_shadow) {
                    if (LIGHTS) pd_lights = pd;
                    else {
                        keep_done = pd.done; keep_weight = pd.weight;
                        keep_dir = pd.done ? pd.radiance : pd.nxt_dir;
                        keep_metal = !pd.done && !(pd.nxt_org.x == P.x && pd.nxt_org.y == P.y && pd.nxt_org.z == P.z);
                    }
                    ro = P; rd = L;
                    if (SHARED) { const RenderArgs& Rs = late(); setup_ray_s(ro, rd, Rs.scene.sspace, rinv, gro, gfar); }
                    else { const RenderArgs& Rs = late(); setup_ray<NODE_FMT>(ro, rd, Rs.scene.grid, Rs.scene.hspace, rinv, gro); }
                    if (NODE_FMT == 8) rot = axis_rot(rinv);
                    rtmax = Ldist - 0.01f; best_t = rtmax; best_slot = -1; best_prim = 0xFFFFFFFFu;
                    node = root; sp = 0; if (WINDOW && deep) { if (SHARED) st2.push(WIN, 0u, 0u); else st.push(WIN, 0); } cur_list = 0u; shadow_ray = true; shadow_hit = false; started_shadow = true;
                    if (SHARED) {       // the ray's interval in the root's box: the root planes' distances are the per-ray constants themselves
                        cur_tn = fmaxf(fmaxf(gro.x, gro.y), fmaxf(gro.z, rtmin)); cur_tf = fminf(fminf(gfar.x, gfar.y), fminf(gfar.z, rtmax));
                        if (!(cur_tn <= cur_tf * kFarWiden)) node = kSentinel;
                    }
                } else {
                    segment_done = true;
                }
            }
        }
        n_shadow += (unsigned long long)popc(vote(started_shadow));
        bool end = false, finished = false;
        if (segment_done) {                                           // raygen :761-778
            if (LIGHTS) lp.result += pd.radiance;                     // light mode 1: already times the throughput
            else add_segment<FM>(lp.result, pd.radiance, att);
            float p = roulette_p<FM>(att);
            if (LIGHTS) p = fminf(p, 1.0f);                           // the 2 cos weight can lift the throughput above 1; a survival probability is <= 1
            const bool rr = rnd(pseed) > p;
            end = pd.done || rr || (uint32_t)depth >= A.maxDepth;
            if (!end) {
                att = roulette_scale<FM>(att, p);
                ro = pd.nxt_org; rd = pd.nxt_dir;
                ++depth;
                if (SKIP) skip_now = skip_bounce;
            } else {
                lp.samples_left--;
                lp.new_path = true;
                if (lp.samples_left == 0u) { lp.alive = false; finished = true; }
            }
        }
        n_paths += (unsigned long long)popc(vote(end));
        finished = finished || fin_pending;
        fin_pending = false;
        n_pixels += (unsigned long long)popc(vote(finished));
        if (STATS) t_mark = __builtin_amdgcn_s_memrealtime();
        finish_runs(A, q, below, lp, finished, book, scratch);     // before the refill overwrites the lanes' items
        if (STATS) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); t_finish += now - t_mark; t_mark = now; }

        refill_lanes<STATS>(A, late, q, lane, below, lp, lcg_skip, book);
        if (q.skipped != 0u) {      // pixels that cannot reach the scene box: every sample is one radiance segment that misses, one path
            const unsigned long long n = (unsigned long long)q.skipped * A.spp;
            n_radiance += n; n_paths += n; n_culled += n;
            n_pixels += (unsigned long long)q.skipped << A.chunk_shift;
            q.skipped = 0u;
        }
        if (STATS) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); t_refill += now - t_mark; t_mark = now; }
        if (STATS && t_drain == 0ull && q.shards_left == 0u && q.res_count == 0u) t_drain = __builtin_amdgcn_s_memrealtime();

        bool start_radiance = segment_done && !end;
        uint32_t my_culled = 0u;                                      // per lane: the counters are wave-uniform and must not be touched under divergence
        if (lp.alive && lp.new_path) {                                // camera path start, :727-745
            const RenderArgs& Rc = late();
            const f3 eye = mk(Rc.eye), camU = mk(Rc.U), camV = mk(Rc.V), camW = mk(Rc.W);
            const float fw = (float)(int)Rc.width, fh = (float)(int)Rc.height;
            // camera-ray cull against the scene box (reaches_scene): corners relative to the eye; an empty scene is never reached
            const f3 elo = Rc.scene.n_tris ? mk(Rc.cull_lo) - eye : mk(1.0f), ehi = Rc.scene.n_tris ? mk(Rc.cull_hi) - eye : mk(-1.0f);
            f3 D;
            for (;;) {
                const float jx = rnd(lp.seed);
                const float jy = rnd(lp.seed);
                D = camera_dir<FM>((float)(lp.pxy & 0xFFFFu), (float)(lp.pxy >> 16), jx, jy, fw, fh, camU, camV, camW);
                // a camera ray that cannot reach the scene box: one radiance segment that misses (:833-847 adds nothing to
                // the result, done = true); its path ends here and the lane goes on to its next sample
                if ((lp.tag & (1u << 24)) != 0u || reaches_scene(D, elo, ehi)) break;      // bit 24: every ray of this pixel reaches the box
                my_culled++;
                lp.samples_left--;
                if (lp.samples_left == 0u) { lp.alive = false; fin_pending = true; break; }
            }
            if (lp.alive) {
                rd = m_normalize<FM>(D);
                ro = eye;
                att = mk(1.0f);
                pseed = lp.seed;
                depth = 0;
                prev_pdf = 0.0f;
                lp.new_path = false;
                start_radiance = true;
                if (SKIP) skip_now = false;                       // a camera ray starts on no triangle
            }
        }
        if (vote(my_culled != 0u) != 0ull) {                           // wave sum of the per-lane counts, bit plane by bit plane
            unsigned long long sum = 0ull;
            for (uint32_t b = 0; vote((my_culled >> b) != 0u) != 0ull; b++) sum += (unsigned long long)popc(vote(((my_culled >> b) & 1u) != 0u)) << b;
            n_radiance += sum; n_paths += sum; n_culled += sum;
        }
        if (STATS) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); t_newpath += now - t_mark; }
        if (vote(lp.alive) == 0ull) { if (q.shards_left == 0u && q.res_count == 0u && vote(fin_pending) == 0ull) break; else continue; }
        if (start_radiance) {                                         // traceRadiance :750-757
            if (SHARED) { const RenderArgs& Rs = late(); setup_ray_s(ro, rd, Rs.scene.sspace, rinv, gro, gfar); }
            else { const RenderArgs& Rs = late(); setup_ray<NODE_FMT>(ro, rd, Rs.scene.grid, Rs.scene.hspace, rinv, gro); }
            if (NODE_FMT == 8) rot = axis_rot(rinv);
            rtmax = 1e16f; best_t = rtmax; best_slot = -1; best_prim = 0xFFFFFFFFu;
            node = root; sp = 0; if (WINDOW && deep) { if (SHARED) st2.push(WIN, 0u, 0u); else st.push(WIN, 0); } cur_list = 0u; shadow_ray = false;
            if (SHARED) {
                cur_tn = fmaxf(fmaxf(gro.x, gro.y), fmaxf(gro.z, rtmin)); cur_tf = fminf(fminf(gfar.x, gfar.y), fminf(gfar.z, rtmax));
                if (!(cur_tn <= cur_tf * kFarWiden)) node = kSentinel;
            }
        }
        n_radiance += (unsigned long long)popc(vote(start_radiance));

        if (STATS) { const unsigned long long now = __builtin_amdgcn_s_memrealtime(); t_in_shade += now - t_phase; t_phase = now; }
        // =========================== traversal: until SHADE_K lanes are parked =====================
        const unsigned long long alive_mask = vote(lp.alive);          // fixed while the wave traverses
        for (;;) {
            const bool act = node != kSentinel;
            const unsigned long long am = vote(act);
            if (am == 0ull) break;
            if (popc(alive_mask & ~am) >= SHADE_K) break;              // parked lanes: scalar arithmetic on the two masks
            if (STATS) { n_steps += 1; n_lane_steps += (unsigned long long)popc(am); }
            if (NODE_FMT == 3) {
                const bool at_inner = act && node >= 0;
                const bool leaf_lane = node < 0;
                const unsigned long long lmask = vote(leaf_lane);
                const bool leaf_round = lmask != 0ull && (LEAF_K <= 1 || popc(lmask) >= LEAF_K || vote(at_inner) == 0ull);
                bool next = false;
                if (at_inner) {
                    uint32_t base;
                    const uint32_t list = wide_visit(sc.wrecs, node, ro, rinv, rtmin, best_t, base);
                    if (list != 0u) {
                        if (cur_list != 0u) { st2.push(sp, cur_base, cur_list); sp++; }
                        cur_base = base; cur_list = list;
                    }
                    next = true;
                }
                if (leaf_round && leaf_lane) {
                    const int slot = ~node;
                    const TriRecord* tp = sc.tris + slot;
                    const float4 r0 = tp->r0, r1 = tp->r1, r2 = tp->r2;
                    float t;
                    const bool ok = tri_test_lazy(ro, rd, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), rtmin, rtmax, t);
                    const uint32_t prim = __float_as_uint(r2.y);
                    if (ok) {
                        if (shadow_ray) { shadow_hit = true; cur_list = 0u; sp = 0; }
                        else if (t < best_t || (t == best_t && prim < best_prim)) { best_t = t; best_slot = slot; best_prim = prim; }
                    }
                    next = true;
                }
                if (next) {
                    if (cur_list == 0u && sp > 0) { sp--; const uint2 g = st2.pop(sp); cur_base = g.x; cur_list = g.y; }
                    if (cur_list != 0u) {
                        const uint32_t nib = cur_list & 15u;
                        cur_list >>= 4;
                        const int idx = (int)(cur_base + (nib & 3u));
                        node = (nib & 4u) ? ~idx : idx;
                    } else {
                        node = kSentinel;
                    }
                }
                continue;
            }
            if constexpr (SHARED) {
                // ---- shared-plane records: one 16-byte gather per visit, the ray's interval carried down and on the stack ----
                static_assert(!SHARED || (INNER >= 1 && STACK_CAP <= 0 && TOPN == 0 && !SKIP), "shared-plane kernel: register stack top, whole or windowed LDS stack");
                constexpr int TRIP = INNER >= 2 ? INNER : 1;
                const auto push8 = [&](int at, int ref, uint32_t iv) { st2.push(WINDOW ? (at & (WIN - 1)) : at, (uint32_t)ref, iv); };
                const auto pop8 = [&](int at, int& ref, uint32_t& iv) { const uint2 e = st2.pop(WINDOW ? (at & (WIN - 1)) : at); ref = (int)e.x; iv = e.y; };
                if (WINDOW && deep) {
                    static_assert(!WINDOW || WIN >= 2 * TRIP + LEAVES + 3, "window too small: moving entries out and back in would alternate");
                    int wbase = (int)st2.pop(WIN).x;
                    for (;;) {
                        const bool out = act && sp + TRIP > wbase + WIN;
                        const bool in = act && wbase > 0 && sp - (TRIP + LEAVES) < wbase;
                        if (vote(out || in) == 0ull) break;
                        n_moves += 1u;
                        if (out) {
#pragma unroll
                            for (int j = 0; j < 4; j++) ovf2[(uint32_t)(wbase + j) * 64u + lane] = st2.pop((wbase + j) & (WIN - 1));
                            wbase += 4;
                        } else if (in) {
                            wbase -= 4;
#pragma unroll
                            for (int j = 0; j < 4; j++) { const uint2 e = ovf2[(uint32_t)(wbase + j) * 64u + lane]; st2.push((wbase + j) & (WIN - 1), e.x, e.y); }
                        }
                        if (out || in) st2.push(WIN, (uint32_t)wbase, 0u);
                    }
                }
#pragma unroll
                for (int visit = 0; visit < TRIP; visit++)
                if ((uint32_t)node < (uint32_t)kSentinel) {
                    const uint4 q = *(const uint4*)((const char*)sc.srecs + (size_t)((uint32_t)node << 4));
                    cur_tf = vmin_raw(cur_tf, best_t * kTieWiden);
                    float n0, f0, n1, f1;
                    slab_s(q.x, q.y, q.z, rinv, gro, gfar, cur_tn, cur_tf, n0, f0, n1, f1);
                    // children: two 16-bit references, bit 15 = triangle (sign-extended: negative, as every leaf reference of this kernel)
                    int c0, c1;
                    if (NODE_FMT == 10) { c0 = (int)(short)(q.w & 0xFFFFu); c1 = (int)q.w >> 16; }
                    else {      // one 30-bit index: child 1 follows child 0 (a triangle is three records); bit 31 / 30: child 0 / 1 is a triangle
                        const uint32_t t = q.w >> 30, base = q.w & kSBaseMask;
                        c0 = (int)(q.w & ~kSLeaf1);
                        c1 = (int)((base + (t & 2u) + 1u) | (t << 31));
                    }
                    const bool h0 = n0 <= f0 * kFarWiden, h1 = n1 <= f1 * kFarWiden;
                    const bool first0 = n0 <= n1;
                    const bool pick0 = h0 && (first0 || !h1);
                    if (h0 && h1) { push8(sp, tos, tos_iv); tos = first0 ? c1 : c0; tos_iv = pack_interval(first0 ? n1 : n0, first0 ? f1 : f0); sp++; }
                    if (h0 || h1) {
                        node = pick0 ? c0 : c1; cur_tn = pick0 ? n0 : n1; cur_tf = pick0 ? f0 : f1;
                    } else {
                        node = sp ? tos : kSentinel;
                        unpack_interval(tos_iv, cur_tn, cur_tf);
                        sp = sp ? sp - 1 : 0;
                        pop8(sp, tos, tos_iv);
                    }
                }
                const bool at_leaf = node < 0;
                const unsigned long long lm = vote(at_leaf);
                if (lm != 0ull && (LEAF_K <= 1 || popc(lm) >= LEAF_K || vote(node >= 0 && node != kSentinel) == 0ull)) {
#pragma unroll
                    for (int leaf = 0; leaf < LEAVES; leaf++)
                    if (node < 0) {
                        const uint4* tp = (const uint4*)((const char*)sc.srecs + (size_t)(((uint32_t)node & (NODE_FMT == 10 ? 0x7FFFu : 0x7FFFFFFFu)) << 4));
                        const uint4 u0 = tp[0], u1 = tp[1], u2 = tp[2];
                        const float4 r0 = make_float4(__uint_as_float(u0.x), __uint_as_float(u0.y), __uint_as_float(u0.z), __uint_as_float(u0.w));
                        const float4 r1 = make_float4(__uint_as_float(u1.x), __uint_as_float(u1.y), __uint_as_float(u1.z), __uint_as_float(u1.w));
                        float t;
                        const bool ok = tri_test_lazy(ro, rd, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, __uint_as_float(u2.x)), rtmin, rtmax, t);
                        const uint32_t prim = u2.y;
                        bool stop = false;
                        if (ok) {
                            if (shadow_ray) { shadow_hit = true; stop = true; }
                            else if (t < best_t || (t == best_t && prim < best_prim)) { best_t = t; best_slot = (int)u2.w; best_prim = prim; }
                        }
                        node = (stop || sp == 0) ? kSentinel : tos;
                        unpack_interval(tos_iv, cur_tn, cur_tf);
                        sp = sp ? sp - 1 : 0;
                        pop8(sp, tos, tos_iv);
                    }
                }
                continue;
            }
            if (WINDOW && deep) {
                // a trip pushes at most TRIP entries (slots sp .. sp + TRIP - 1 must lie inside the window) and pops at most
                // TRIP + LEAVES (down to slot sp - TRIP - LEAVES, which must not have been moved out)
                constexpr int TRIP = INNER >= 2 ? INNER : 1;
                static_assert(!WINDOW || WIN >= 2 * TRIP + LEAVES + 3, "window too small: moving entries out and back in would alternate");
                // slots [0, wbase) of this lane's stack are in global memory; wbase lives in LDS (entry WIN of the lane's column): one
                // conflict-free read per trip instead of a vector register held across the shade phase
                int wbase = st.pop(WIN);
                for (;;) {
                    const bool out = act && sp + TRIP > wbase + WIN;
                    const bool in = act && wbase > 0 && sp - (TRIP + LEAVES) < wbase;
                    if (vote(out || in) == 0ull) break;
                    n_moves += 1u;
                    if (out) {
#pragma unroll
                        for (int j = 0; j < 4; j++) ovf[(uint32_t)(wbase + j) * 64u + lane] = (uint32_t)st.pop((wbase + j) & (WIN - 1));
                        wbase += 4;
                    } else if (in) {
                        wbase -= 4;
#pragma unroll
                        for (int j = 0; j < 4; j++) st.push((wbase + j) & (WIN - 1), (int)ovf[(uint32_t)(wbase + j) * 64u + lane]);
                    }
                    if (out || in) st.push(WIN, wbase);
                }
            }
            // INNER == 2: two node visits per trip through the loop control (a lane that reaches a leaf or runs dry in the
            // first sits out the second)
#pragma unroll
            for (int visit = 0; visit < (INNER >= 2 ? INNER : 1); visit++)
            if ((uint32_t)node < (uint32_t)kSentinel) {
                float n0, f0, n1, f1; int c0, c1;
                if (NODE_FMT == 5) {        // the two-step slab test (p - o) * (1/d): comparison variant
                    const BvhNode* np = sc.nodes + node;
                    const float4 a = np->a, b = np->b, c = np->c;
                    const int4 ch = np->d;
                    c0 = ch.x; c1 = ch.y;
                    float x0 = (a.x - ro.x) * rinv.x, x1 = (a.w - ro.x) * rinv.x;
                    float y0 = (a.y - ro.y) * rinv.y, y1 = (b.x - ro.y) * rinv.y;
                    float z0 = (a.z - ro.z) * rinv.z, z1 = (b.y - ro.z) * rinv.z;
                    n0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), rtmin));
                    f0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * kFarWiden;
                    float u0 = (b.z - ro.x) * rinv.x, u1 = (c.y - ro.x) * rinv.x;
                    float v0 = (b.w - ro.y) * rinv.y, v1 = (c.z - ro.y) * rinv.y;
                    float w0 = (c.x - ro.z) * rinv.z, w1 = (c.w - ro.z) * rinv.z;
                    n1 = fmaxf(fmaxf(fminf(u0, u1), fminf(v0, v1)), fmaxf(fminf(w0, w1), rtmin));
                    f1 = fminf(fminf(fmaxf(u0, u1), fmaxf(v0, v1)), fmaxf(w0, w1)) * kFarWiden;
                } else if (NODE_FMT == 0) {
                    // fp32 nodes, slab planes as one full-rate fma each: t = p * (1/d) + (-o/d); lbvh_build.hip's pad_abs
                    // covers the single rounding of -o/d.  32-bit byte offset (scalar base + vector offset addressing; the
                    // scene size limit in pt_set_scene keeps it below 4 GB)
                    const BvhNode* np = (const BvhNode*)((const char*)sc.nodes + (size_t)((uint32_t)node << 6));
                    const float4 a = np->a, b = np->b, c = np->c;
                    const int4 ch = np->d;
                    c0 = ch.x; c1 = ch.y;
                    const float x0 = __builtin_fmaf(a.x, rinv.x, gro.x), x1 = __builtin_fmaf(a.w, rinv.x, gro.x);
                    const float y0 = __builtin_fmaf(a.y, rinv.y, gro.y), y1 = __builtin_fmaf(b.x, rinv.y, gro.y);
                    const float z0 = __builtin_fmaf(a.z, rinv.z, gro.z), z1 = __builtin_fmaf(b.y, rinv.z, gro.z);
                    n0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), rtmin));
                    f0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * kFarWiden;
                    const float u0 = __builtin_fmaf(b.z, rinv.x, gro.x), u1 = __builtin_fmaf(c.y, rinv.x, gro.x);
                    const float v0 = __builtin_fmaf(b.w, rinv.y, gro.y), v1 = __builtin_fmaf(c.z, rinv.y, gro.y);
                    const float w0 = __builtin_fmaf(c.x, rinv.z, gro.z), w1 = __builtin_fmaf(c.w, rinv.z, gro.z);
                    n1 = fmaxf(fmaxf(fminf(u0, u1), fminf(v0, v1)), fmaxf(fminf(w0, w1), rtmin));
                    f1 = fminf(fminf(fmaxf(u0, u1), fmaxf(v0, v1)), fmaxf(w0, w1)) * kFarWiden;
                } else if (NODE_FMT == 7) {
                    // fp16 planes, two 16-byte loads; every plane is one v_fma_mix_f32 (the fp16 -> fp32 conversion is part of it)
                    const uint4* np = (const uint4*)((const char*)sc.hnodes + (size_t)((uint32_t)node << 5));
                    const uint4 qa = np[0], qb = np[1];
                    c0 = (int)qa.w; c1 = (int)qb.w;
                    const float x0 = fma_h_lo(qa.x, rinv.x, gro.x), x1 = fma_h_hi(qa.x, rinv.x, gro.x);
                    const float y0 = fma_h_lo(qa.y, rinv.y, gro.y), y1 = fma_h_hi(qa.y, rinv.y, gro.y);
                    const float z0 = fma_h_lo(qa.z, rinv.z, gro.z), z1 = fma_h_hi(qa.z, rinv.z, gro.z);
                    n0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), rtmin));
                    f0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * kFarWiden;
                    const float u0 = fma_h_lo(qb.x, rinv.x, gro.x), u1 = fma_h_hi(qb.x, rinv.x, gro.x);
                    const float v0 = fma_h_lo(qb.y, rinv.y, gro.y), v1 = fma_h_hi(qb.y, rinv.y, gro.y);
                    const float w0 = fma_h_lo(qb.z, rinv.z, gro.z), w1 = fma_h_hi(qb.z, rinv.z, gro.z);
                    n1 = fmaxf(fmaxf(fminf(u0, u1), fminf(v0, v1)), fmaxf(fminf(w0, w1), rtmin));
                    f1 = fminf(fminf(fmaxf(u0, u1), fmaxf(v0, v1)), fmaxf(w0, w1)) * kFarWiden;
                } else if (NODE_FMT == 8) {
                    // fp16 planes as NODE_FMT 7; each packed {lo, hi} pair is rotated by the ray's per-axis amount first, so the
                    // low half is always the near plane: 6 rotates replace 12 min / max
                    const uint4* np = (const uint4*)((const char*)sc.hnodes + (size_t)((uint32_t)node << 5));
                    const uint4 qa = np[0], qb = np[1];
                    c0 = (int)qa.w; c1 = (int)qb.w;
                    const uint32_t ax = rot16(qa.x, rot.x), ay = rot16(qa.y, rot.y), az = rot16(qa.z, rot.z);
                    n0 = fmaxf(fmaxf(fma_h_lo(ax, rinv.x, gro.x), fma_h_lo(ay, rinv.y, gro.y)), fmaxf(fma_h_lo(az, rinv.z, gro.z), rtmin));
                    f0 = fminf(fminf(fma_h_hi(ax, rinv.x, gro.x), fma_h_hi(ay, rinv.y, gro.y)), fma_h_hi(az, rinv.z, gro.z)) * kFarWiden;
                    const uint32_t bx = rot16(qb.x, rot.x), by = rot16(qb.y, rot.y), bz = rot16(qb.z, rot.z);
                    n1 = fmaxf(fmaxf(fma_h_lo(bx, rinv.x, gro.x), fma_h_lo(by, rinv.y, gro.y)), fmaxf(fma_h_lo(bz, rinv.z, gro.z), rtmin));
                    f1 = fminf(fminf(fma_h_hi(bx, rinv.x, gro.x), fma_h_hi(by, rinv.y, gro.y)), fma_h_hi(bz, rinv.z, gro.z)) * kFarWiden;
                } else if (NODE_FMT == 9) {
                    // NODE_FMT 8 with the rotate amounts read from the low bits of the plane multipliers (setup_ray)
                    uint4 qa, qb;
                    if (TOPN > 0 && ((uint32_t)node & kTopNodeFlag)) {             // the top of the tree: from LDS
                        const uint4* tp = top_lds + 2u * ((uint32_t)node & 0xFFFFu);
                        qa = tp[0]; qb = tp[1];
                    } else {
                        const uint4* np = (const uint4*)((const char*)sc.hnodes + (size_t)((uint32_t)node << 5));
                        qa = np[0]; qb = np[1];
                    }
                    c0 = (int)qa.w; c1 = (int)qb.w;
                    slab_h9(qa.x, qa.y, qa.z, rinv, gro, rtmin, n0, f0);
                    slab_h9(qb.x, qb.y, qb.z, rinv, gro, rtmin, n1, f1);
                } else if (NODE_FMT == 14) {
                    const uint4* np = (const uint4*)((const char*)lds_nodes + (uint32_t)node);
                    const uint4 qa = np[0], qb = np[1];
                    c0 = (int)qa.w; c1 = (int)qb.w;
                    slab_hc(qa.x, qa.y, qa.z, rinv, gro, rtmin, n0, f0);
                    slab_hc(qb.x, qb.y, qb.z, rinv, gro, rtmin, n1, f1);
                } else if (NODE_FMT == 13) {
                    // NODE_FMT 11 with child 0's half of the node from LDS (a node's byte offset halved is its place there), child 1's through the texture path
                    const uint4 qa = *(const uint4*)((const char*)lds_nodes + ((uint32_t)node >> 1));
                    const uint4 qb = *(const uint4*)((const char*)sc.hcnodes + (size_t)(uint32_t)node + 16u);
                    c0 = (int)qa.w; c1 = (int)qb.w;
                    slab_hc(qa.x, qa.y, qa.z, rinv, gro, rtmin, n0, f0);
                    slab_hc(qb.x, qb.y, qb.z, rinv, gro, rtmin, n1, f1);
                } else if (NODE_FMT == 11) {
                    // fp16 centre / half-extent nodes (pt_device.h): no rotates; child references of inner nodes are byte offsets
                    const uint4* np = (const uint4*)((const char*)sc.hcnodes + (size_t)(uint32_t)node);
                    const uint4 qa = np[0], qb = np[1];
                    c0 = (int)qa.w; c1 = (int)qb.w;
                    slab_hc(qa.x, qa.y, qa.z, rinv, gro, rtmin, n0, f0);
                    slab_hc(qb.x, qb.y, qb.z, rinv, gro, rtmin, n1, f1);
                } else if (NODE_FMT == 6) {
                    // centre / half-extent nodes: near = (c - o)/d - h/|d|, far = (c - o)/d + h/|d|: full-rate arithmetic only,
                    // the |.| is a source modifier
                    const BvhNode* np = (const BvhNode*)((const char*)sc.cnodes + (size_t)((uint32_t)node << 6));
                    const float4 a = np->a, b = np->b, c = np->c;
                    const int4 ch = np->d;
                    c0 = ch.x; c1 = ch.y;
                    const float ax = fabsf(rinv.x), ay = fabsf(rinv.y), az = fabsf(rinv.z);
                    const float cx0 = __builtin_fmaf(a.x, rinv.x, gro.x), hx0 = a.w * ax;
                    const float cy0 = __builtin_fmaf(a.y, rinv.y, gro.y), hy0 = b.x * ay;
                    const float cz0 = __builtin_fmaf(a.z, rinv.z, gro.z), hz0 = b.y * az;
                    n0 = fmaxf(fmaxf(cx0 - hx0, cy0 - hy0), fmaxf(cz0 - hz0, rtmin));
                    f0 = fminf(fminf(cx0 + hx0, cy0 + hy0), cz0 + hz0) * kFarWiden;
                    const float cx1 = __builtin_fmaf(b.z, rinv.x, gro.x), hx1 = c.y * ax;
                    const float cy1 = __builtin_fmaf(b.w, rinv.y, gro.y), hy1 = c.z * ay;
                    const float cz1 = __builtin_fmaf(c.x, rinv.z, gro.z), hz1 = c.w * az;
                    n1 = fmaxf(fmaxf(cx1 - hx1, cy1 - hy1), fmaxf(cz1 - hz1, rtmin));
                    f1 = fminf(fminf(cx1 + hx1, cy1 + hy1), cz1 + hz1) * kFarWiden;
                } else if (NODE_FMT == 4) {
                    // 16-bit grid nodes, one conversion + one fma per plane (the grid's one-cell outward rounding covers the
                    // fma form's error, which is below 0.01 cell)
                    const QNode* np = sc.qnodes + node;
                    const uint4 qa = np->a, qb = np->b;
                    c0 = (int)qa.w; c1 = (int)qb.w;
                    const float x0 = __builtin_fmaf((float)(qa.x & 0xFFFFu), rinv.x, gro.x), x1 = __builtin_fmaf((float)(qa.y >> 16), rinv.x, gro.x);
                    const float y0 = __builtin_fmaf((float)(qa.x >> 16), rinv.y, gro.y), y1 = __builtin_fmaf((float)(qa.z & 0xFFFFu), rinv.y, gro.y);
                    const float z0 = __builtin_fmaf((float)(qa.y & 0xFFFFu), rinv.z, gro.z), z1 = __builtin_fmaf((float)(qa.z >> 16), rinv.z, gro.z);
                    n0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), rtmin));
                    f0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * kFarWiden;
                    const float u0 = __builtin_fmaf((float)(qb.x & 0xFFFFu), rinv.x, gro.x), u1 = __builtin_fmaf((float)(qb.y >> 16), rinv.x, gro.x);
                    const float v0 = __builtin_fmaf((float)(qb.x >> 16), rinv.y, gro.y), v1 = __builtin_fmaf((float)(qb.z & 0xFFFFu), rinv.y, gro.y);
                    const float w0 = __builtin_fmaf((float)(qb.y & 0xFFFFu), rinv.z, gro.z), w1 = __builtin_fmaf((float)(qb.z >> 16), rinv.z, gro.z);
                    n1 = fmaxf(fmaxf(fminf(u0, u1), fminf(v0, v1)), fmaxf(fminf(w0, w1), rtmin));
                    f1 = fminf(fminf(fmaxf(u0, u1), fmaxf(v0, v1)), fmaxf(w0, w1)) * kFarWiden;
                } else {
                    uint4 qa, qb;
                    if (NODE_FMT == 2) {
                        const uint2* p = lds_nodes + 4 * node;
                        const uint2 t0 = p[0], t1 = p[1], t2 = p[2], t3 = p[3];
                        qa = make_uint4(t0.x, t0.y, t1.x, t1.y); qb = make_uint4(t2.x, t2.y, t3.x, t3.y);
                    } else {
                        const QNode* np = sc.qnodes + node;
                        qa = np->a; qb = np->b;
                    }
                    c0 = (int)qa.w; c1 = (int)qb.w;
                    float x0 = ((float)(qa.x & 0xFFFFu) - gro.x) * rinv.x, x1 = ((float)(qa.y >> 16) - gro.x) * rinv.x;
                    float y0 = ((float)(qa.x >> 16) - gro.y) * rinv.y, y1 = ((float)(qa.z & 0xFFFFu) - gro.y) * rinv.y;
                    float z0 = ((float)(qa.y & 0xFFFFu) - gro.z) * rinv.z, z1 = ((float)(qa.z >> 16) - gro.z) * rinv.z;
                    n0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), rtmin));
                    f0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * kFarWiden;
                    float u0 = ((float)(qb.x & 0xFFFFu) - gro.x) * rinv.x, u1 = ((float)(qb.y >> 16) - gro.x) * rinv.x;
                    float v0 = ((float)(qb.x >> 16) - gro.y) * rinv.y, v1 = ((float)(qb.z & 0xFFFFu) - gro.y) * rinv.y;
                    float w0 = ((float)(qb.y & 0xFFFFu) - gro.z) * rinv.z, w1 = ((float)(qb.z >> 16) - gro.z) * rinv.z;
                    n1 = fmaxf(fmaxf(fminf(u0, u1), fminf(v0, v1)), fmaxf(fminf(w0, w1), rtmin));
                    f1 = fminf(fminf(fmaxf(u0, u1), fmaxf(v0, v1)), fmaxf(w0, w1)) * kFarWiden;
                }
                if (DIAG == 1) {
                    float d = n0;
#pragma unroll
                    for (int k = 0; k < 12; k++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(d) : "v"(f0));
                    asm volatile("" :: "v"(d));
                }
                if (DIAG == 2) {
                    const float4* xp = (const float4*)(sc.nodes + node);
    typedef float v4f __attribute__((ext_vector_type(4)));
                    v4f e0, e1;
                    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:32\n\ts_waitcnt vmcnt(0)"
                                 : "=&v"(e0), "=&v"(e1) : "v"(xp) : "memory");
                    asm volatile("" :: "v"(e0), "v"(e1));
                }
                f0 = fminf(f0, best_t * kTieWiden);
                f1 = fminf(f1, best_t * kTieWiden);
                const bool h0 = n0 <= f0, h1 = n1 <= f1;
                if (INNER == 0) {
                    if (h0 && h1) {
                        const bool first0 = n0 <= n1;
                        push(sp, first0 ? c1 : c0);
                        sp++;
                        node = first0 ? c0 : c1;
                    } else if (h0) {
                        node = c0;
                    } else if (h1) {
                        node = c1;
                    } else {
                        if (sp == 0) node = kSentinel; else { sp--; node = pop(sp); }
                    }
                } else {
                    // elements e_1..e_sp, e_sp in `tos`, e_k (k < sp) in LDS slot k
                    const bool first0 = n0 <= n1;
                    const int near_c = (h0 && (first0 || !h1)) ? c0 : c1;
                    const int far_c = first0 ? c1 : c0;
                    if (h0 && h1) { push(sp, tos); tos = far_c; sp++; }
                    if (h0 || h1) {
                        node = near_c;
                    } else {
                        node = sp ? tos : kSentinel;
                        sp = sp ? sp - 1 : 0;
                        tos = pop(sp);
                    }
                }
            }
            if (SKIP) {
                if (skip_now && node == origin_ref) {     // sitting at the triangle the ray started on: it cannot be hit (see SKIP above)
                    node = sp ? tos : kSentinel;
                    sp = sp ? sp - 1 : 0;
                    tos = pop(sp);
                }
            }
            const bool at_leaf = node < 0;       // kSentinel is positive
            const unsigned long long lm = vote(at_leaf);
            if (lm != 0ull && (LEAF_K <= 1 || popc(lm) >= LEAF_K || vote(node >= 0 && node != kSentinel) == 0ull)) {
#pragma unroll
                for (int leaf = 0; leaf < LEAVES; leaf++)          // LEAVES == 2: a lane whose next node is a leaf again tests it in the same round
                if (node < 0) {
                    const int slot = ~node;
                    const TriRecord* tp = (const TriRecord*)((const char*)sc.tris + (size_t)((uint32_t)slot * 48u));
                    const float4 r0 = tp->r0, r1 = tp->r1, r2 = tp->r2;
                    float t;
                    const bool ok = tri_test_lazy(ro, rd, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), rtmin, rtmax, t);
                    const uint32_t prim = __float_as_uint(r2.y);
                    bool stop = false;
                    if (ok) {
                        if (shadow_ray) { shadow_hit = true; stop = true; }
                        else if (t < best_t || (t == best_t && prim < best_prim)) { best_t = t; best_slot = slot; best_prim = prim; }
                    }
                    if (INNER == 0) {
                        if (stop || sp == 0) node = kSentinel; else { sp--; node = pop(sp); }
                    } else {
                        node = (stop || sp == 0) ? kSentinel : tos;
                        sp = sp ? sp - 1 : 0;
                        tos = pop(sp);
                    }
                }
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&A.counters[0], n_radiance);
        atomicAdd(&A.counters[1], n_shadow);
        atomicAdd(&A.counters[2], n_paths);
        atomicAdd(&A.counters[3], n_pixels);
        atomicAdd(&A.counters[4], n_steps);
        atomicAdd(&A.counters[5], n_lane_steps);
        atomicAdd(&A.counters[6], n_rounds);
        atomicAdd(&A.counters[7], n_lane_rounds);
        if (n_culled) atomicAdd(&A.counters[kCulledCounter], n_culled);
        if (WINDOW && n_moves) atomicAdd(&A.counters[kWindowMoves], (unsigned long long)n_moves);
        if (STATS) {
            const uint32_t w = blockIdx.x * (THREADS / 64) + wave;
            if (w < kMaxTimedWaves) {
                A.counters[8 + 3 * w] = t_start;
                A.counters[8 + 3 * w + 1] = t_drain;
                atomicAdd(&A.counters[8 + 3 * kMaxTimedWaves + 2048], t_in_shade);       // 10 ns units, summed over waves
                atomicAdd(&A.counters[8 + 3 * kMaxTimedWaves + 2049], t_refill);         // ... of which: queue refill,
                atomicAdd(&A.counters[8 + 3 * kMaxTimedWaves + 2050], t_finish);         // finished runs (park / fold / write),
                atomicAdd(&A.counters[8 + 3 * kMaxTimedWaves + 2051], t_newpath);        // camera-path start incl. the cull
                A.counters[8 + 3 * w + 2] = __builtin_amdgcn_s_memrealtime();
            }
        }
    }
}

__global__ void k_resolve(const float4* __restrict__ accum, uint32_t* __restrict__ fb, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float4 a = accum[i]; fb[i] = make_color(mk(a.x, a.y, a.z)); }
}

// multi-GPU group (capi.hip pt_multi): a rank's private accumulation buffer holds its own pixels and zero elsewhere.
// k_keep_owned re-establishes that after the caller's buffer was copied in (a restored accumulation); the owner of a pixel is
// the inverse of StaticWorkDistribution::getSamplePixel (sutil/WorkDistribution.h:60-81).
__global__ void k_keep_owned(float4* __restrict__ accum, uint32_t width, uint32_t n, int rank, int world)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = i % width, y = i / width;
    const int col = (int)((x >> 3) % (uint32_t)world), row = (int)((y >> 2) % (uint32_t)world);
    const int owner = (col - row + world) % world;
    if (owner != rank) accum[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
// rehearsal of the reduce on a one-GPU box (all ranks' buffers on one device): dst = srcs[0] + srcs[1] + ...
struct SumSources { const float4* p[16]; int n; };
__global__ void k_sum_ranks(float4* __restrict__ dst, const SumSources src, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = src.p[0][i];
    for (int k = 1; k < src.n; k++) { const float4 b = src.p[k][i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    dst[i] = a;
}
hipError_t launch_keep_owned(float4* accum, uint32_t width, uint32_t height, int rank, int world, hipStream_t stream)
{
    const uint32_t n = width * height;
    k_keep_owned<<<(n + 255) / 256, 256, 0, stream>>>(accum, width, n, rank, world);
    return hipGetLastError();
}
hipError_t launch_sum_ranks(float4* dst, const float4* const* srcs, int n_srcs, uint32_t n, hipStream_t stream)
{
    if (n_srcs < 1 || n_srcs > 16) return hipErrorInvalidValue;
    SumSources s; s.n = n_srcs;
    for (int k = 0; k < 16; k++) s.p[k] = k < n_srcs ? srcs[k] : nullptr;
    k_sum_ranks<<<(n + 255) / 256, 256, 0, stream>>>(dst, s, n);
    return hipGetLastError();
}

hipError_t launch_finalize(const RenderArgs& args, hipStream_t stream)
{
    const uint32_t slots = args.total_samples >> args.sub_shift;
    k_finalize<<<(slots + kFinThreads - 1) / kFinThreads, kFinThreads, 0, stream>>>(args);
    return hipGetLastError();
}

hipError_t launch_resolve(const float4* accum, uint32_t* fb, uint32_t n, hipStream_t stream)
{
    k_resolve<<<(n + 255) / 256, 256, 0, stream>>>(accum, fb, n);
    return hipGetLastError();
}

// ---- standalone ray queries (parity tests): same traversal, one ray per lane -------------
__global__ void __launch_bounds__(256)
k_trace_closest(const DeviceScene sc, uint32_t stack_entries, const float* __restrict__ rays, uint32_t n,
                float* __restrict__ t_out, uint32_t* __restrict__ prim_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    LaneStack st;
    st.base = lds_dyn + (threadIdx.x >> 6) * (stack_entries * 64u) + (threadIdx.x & 63u);
    const bool active = i < n;
    f3 o = mk(0.0f), d = mk(0.0f, 0.0f, 1.0f); float tmin = 0.0f, tmax = 0.0f;
    if (active) { const float* r = rays + 8ull * i; o = mk(r[0], r[1], r[2]); d = mk(r[3], r[4], r[5]); tmin = r[6]; tmax = r[7]; }
    HitRec h;
    traverse<false>(sc, st, active, o, d, tmin, tmax, h);
    if (active) { t_out[i] = h.slot >= 0 ? h.t : -1.0f; prim_out[i] = h.prim; }
}

__global__ void __launch_bounds__(256)
k_trace_any(const DeviceScene sc, uint32_t stack_entries, const float* __restrict__ rays, uint32_t n, uint8_t* __restrict__ hit_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    LaneStack st;
    st.base = lds_dyn + (threadIdx.x >> 6) * (stack_entries * 64u) + (threadIdx.x & 63u);
    const bool active = i < n;
    f3 o = mk(0.0f), d = mk(0.0f, 0.0f, 1.0f); float tmin = 0.0f, tmax = 0.0f;
    if (active) { const float* r = rays + 8ull * i; o = mk(r[0], r[1], r[2]); d = mk(r[3], r[4], r[5]); tmin = r[6]; tmax = r[7]; }
    HitRec h;
    const bool f = traverse<true>(sc, st, active, o, d, tmin, tmax, h);
    if (active) hit_out[i] = f ? 1 : 0;
}

// ---- traversal-ceiling diagnostic -----------------------------------------------------------------
// A ray-stream kernel with nothing but the BVH loop: persistent waves pull rays (origin, direction,
// tmin, tmax; tmax < 0 marks an any-hit ray with |tmax|) from a global array, a lane that finishes
// writes its result and takes the next ray inside the loop (ballot-prefix hand-out from a wave-local
// grant).  No path state, no shading: few registers, full occupancy, lanes (almost) never idle.  It
// answers one question — how fast could traversal alone go on this scene and ray mix — and is
// bit-checked against pt_trace_closest / pt_trace_any.
template <int FETCH_K, int LEAF_K, int FMT = 0>
__global__ void __launch_bounds__(256)
k_trace_stream(const DeviceScene sc, uint32_t stack_entries, const float4* __restrict__ rays, uint32_t n,
               uint32_t* __restrict__ head, float* __restrict__ t_out, uint32_t* __restrict__ prim_out,
               unsigned long long* __restrict__ counters)
{
    const uint32_t lane = threadIdx.x & 63u;
    LaneStack st;
    st.base = lds_dyn + (threadIdx.x >> 6) * (stack_entries * 64u) + lane;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    uint32_t res_first = 0, res_count = 0;      // wave-local grant
    bool drained = false;
    // ray in flight
    uint32_t rid = 0xFFFFFFFFu;
    f3 ro = mk(0.0f), rd = mk(0.0f, 0.0f, 1.0f), rinv = mk(1.0f), roi = mk(0.0f);
    float rtmin = 0.0f, rtmax = 0.0f, best_t = 0.0f;
    int best_slot = -1; uint32_t best_prim = 0xFFFFFFFFu;
    int node = kSentinel, sp = 0, tos = kSentinel;
    bool any_ray = false, any_hit = false;
    unsigned long long n_iter = 0, n_visit = 0, n_tri = 0, n_vround = 0, n_lround = 0;
    for (;;) {
        // retire finished rays, fetch new ones
        const bool idle_lane = node == kSentinel;
        unsigned long long idle = vote(idle_lane);
        const unsigned long long busy = ~idle;
        if (idle != 0ull && (popc(idle) >= FETCH_K || busy == 0ull)) {
            if (idle_lane && rid != 0xFFFFFFFFu) {
                if (any_ray) { t_out[rid] = any_hit ? 1.0f : 0.0f; prim_out[rid] = any_hit ? 1u : 0u; }
                else { t_out[rid] = best_slot >= 0 ? best_t : -1.0f; prim_out[rid] = best_prim; }
                rid = 0xFFFFFFFFu;
            }
            while (idle != 0ull && !(drained && res_count == 0u)) {
                if (res_count == 0u) {
                    const uint32_t leader = (uint32_t)__ffsll((long long)idle) - 1u;
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(head, 256u);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
                    res_first = base;
                    res_count = base < n ? min(256u, n - base) : 0u;
                    if (res_count < 256u) drained = true;
                    if (res_count == 0u) break;
                }
                const uint32_t want = (uint32_t)popc(idle);
                const uint32_t take = want < res_count ? want : res_count;
                const uint32_t rank = (uint32_t)popc(idle & below);
                if (idle_lane && rid == 0xFFFFFFFFu && rank < take) {
                    rid = res_first + rank;
                    const float4 a = rays[2ull * rid], b = rays[2ull * rid + 1];
                    ro = mk(a.x, a.y, a.z); rd = mk(a.w, b.x, b.y); rtmin = b.z;
                    any_ray = b.w < 0.0f; rtmax = fabsf(b.w);
                    if (FMT == 2) setup_ray<0>(ro, rd, sc.grid, sc.hspace, rinv, roi);
                    else if (FMT == 3) setup_ray<7>(ro, rd, sc.grid, sc.hspace, rinv, roi);
                    else if (FMT == 4) setup_ray<11>(ro, rd, sc.grid, sc.hspace, rinv, roi);
                    else rinv = mk(fast_rcp(rd.x), fast_rcp(rd.y), fast_rcp(rd.z));
                    best_t = rtmax; best_slot = -1; best_prim = 0xFFFFFFFFu; any_hit = false;
                    node = sc.n_tris ? 0 : kSentinel; sp = 0;
                }
                res_first += take; res_count -= take;
                idle = vote(rid == 0xFFFFFFFFu);
            }
            if (vote(node != kSentinel) == 0ull) {
                if (vote(rid != 0xFFFFFFFFu) == 0ull && drained && res_count == 0u) break;   // nothing in flight, nothing left
                continue;       // rays of an empty scene retire on the next turn
            }
        }
        // one traversal step
        { const unsigned long long vm = vote(node >= 0 && node != kSentinel); n_iter++; n_visit += (unsigned long long)popc(vm); n_vround += vm ? 1u : 0u; }
        if (node >= 0 && node != kSentinel) {
            float x0, x1, y0, y1, z0, z1, u0, u1, v0, v1, w0, w1;
            float n0, f0, n1, f1;
            int2 ch;
            if (FMT == 4) {        // the default render kernels' nodes and box test (NODE_FMT 11): fp16 {centre, half extent}, a scale per axis, child references as byte offsets
                const uint4* hp = (const uint4*)((const char*)sc.hcnodes + (size_t)(uint32_t)node);
                const uint4 qa = hp[0], qb = hp[1];
                ch = make_int2((int)qa.w, (int)qb.w);
                slab_hc(qa.x, qa.y, qa.z, rinv, roi, rtmin, n0, f0);
                slab_hc(qb.x, qb.y, qb.z, rinv, roi, rtmin, n1, f1);
            } else {
            if (FMT == 3) {        // fp16 nodes, two loads, v_fma_mix_f32 planes: the render kernel's NODE_FMT 7
                const uint4* hp = (const uint4*)(sc.hnodes + node);
                const uint4 qa = hp[0], qb = hp[1];
                ch = make_int2((int)qa.w, (int)qb.w);
                x0 = fma_h_lo(qa.x, rinv.x, roi.x); x1 = fma_h_hi(qa.x, rinv.x, roi.x);
                y0 = fma_h_lo(qa.y, rinv.y, roi.y); y1 = fma_h_hi(qa.y, rinv.y, roi.y);
                z0 = fma_h_lo(qa.z, rinv.z, roi.z); z1 = fma_h_hi(qa.z, rinv.z, roi.z);
                u0 = fma_h_lo(qb.x, rinv.x, roi.x); u1 = fma_h_hi(qb.x, rinv.x, roi.x);
                v0 = fma_h_lo(qb.y, rinv.y, roi.y); v1 = fma_h_hi(qb.y, rinv.y, roi.y);
                w0 = fma_h_lo(qb.z, rinv.z, roi.z); w1 = fma_h_hi(qb.z, rinv.z, roi.z);
            } else {
            const BvhNode* np = sc.nodes + node;
            const float4 a = np->a, b = np->b, c = np->c;
            ch = make_int2(np->d.x, np->d.y);
            if (FMT == 2) {        // t = p * (1/d) - o/d, one full-rate fma per plane: the render kernel's form (NODE_FMT 0)
                x0 = __builtin_fmaf(a.x, rinv.x, roi.x); x1 = __builtin_fmaf(a.w, rinv.x, roi.x);
                y0 = __builtin_fmaf(a.y, rinv.y, roi.y); y1 = __builtin_fmaf(b.x, rinv.y, roi.y);
                z0 = __builtin_fmaf(a.z, rinv.z, roi.z); z1 = __builtin_fmaf(b.y, rinv.z, roi.z);
                u0 = __builtin_fmaf(b.z, rinv.x, roi.x); u1 = __builtin_fmaf(c.y, rinv.x, roi.x);
                v0 = __builtin_fmaf(b.w, rinv.y, roi.y); v1 = __builtin_fmaf(c.z, rinv.y, roi.y);
                w0 = __builtin_fmaf(c.x, rinv.z, roi.z); w1 = __builtin_fmaf(c.w, rinv.z, roi.z);
            } else {
                x0 = (a.x - ro.x) * rinv.x; x1 = (a.w - ro.x) * rinv.x;
                y0 = (a.y - ro.y) * rinv.y; y1 = (b.x - ro.y) * rinv.y;
                z0 = (a.z - ro.z) * rinv.z; z1 = (b.y - ro.z) * rinv.z;
                u0 = (b.z - ro.x) * rinv.x; u1 = (c.y - ro.x) * rinv.x;
                v0 = (b.w - ro.y) * rinv.y; v1 = (c.z - ro.y) * rinv.y;
                w0 = (c.x - ro.z) * rinv.z; w1 = (c.w - ro.z) * rinv.z;
            }
            }
            n0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), rtmin));
            f0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * kFarWiden;
            n1 = fmaxf(fmaxf(fminf(u0, u1), fminf(v0, v1)), fmaxf(fminf(w0, w1), rtmin));
            f1 = fminf(fminf(fmaxf(u0, u1), fmaxf(v0, v1)), fmaxf(w0, w1)) * kFarWiden;
            }
            f0 = fminf(f0, best_t * kTieWiden);
            f1 = fminf(f1, best_t * kTieWiden);
            const bool h0 = n0 <= f0, h1 = n1 <= f1;
            const bool first0 = n0 <= n1;
            const int near_c = (h0 && (first0 || !h1)) ? ch.x : ch.y;
            const int far_c = first0 ? ch.y : ch.x;
            if (h0 && h1) { st.push(sp, tos); tos = far_c; sp++; }
            if (h0 || h1) node = near_c;
            else { node = sp ? tos : kSentinel; sp = sp ? sp - 1 : 0; tos = st.pop(sp); }
        }
        const bool at_leaf = node < 0;
        const unsigned long long lm = vote(at_leaf);
        if (lm != 0ull && (popc(lm) >= LEAF_K || vote(node >= 0 && node != kSentinel) == 0ull)) {
            n_tri += (unsigned long long)popc(lm); n_lround++;
            if (at_leaf) {
                const int slot = ~node;
                const TriRecord* tp = sc.tris + slot;
                const float4 r0 = tp->r0, r1 = tp->r1, r2 = tp->r2;
                float t;
                const bool ok = tri_test_lazy(ro, rd, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), rtmin, rtmax, t);
                const uint32_t prim = __float_as_uint(r2.y);
                bool stop = false;
                if (ok) {
                    if (any_ray) { any_hit = true; stop = true; }
                    else if (t < best_t || (t == best_t && prim < best_prim)) { best_t = t; best_slot = slot; best_prim = prim; }
                }
                node = (stop || sp == 0) ? kSentinel : tos;
                sp = sp ? sp - 1 : 0;
                tos = st.pop(sp);
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&counters[0], n_iter); atomicAdd(&counters[1], n_visit); atomicAdd(&counters[2], n_tri);
        atomicAdd(&counters[3], n_vround); atomicAdd(&counters[4], n_lround);
    }
}

// The same ray-stream kernel over the four-wide tree (wide_bvh.hip): a lane's current work item is a wide
// node (>= 0), a triangle record (~index) or nothing; the children still to do at each level wait as
// {base, near-to-far nibble list} groups, the innermost in registers, the rest on the LDS stack.
template <int FETCH_K, int LEAF_K>
__global__ void __launch_bounds__(256)
k_trace_stream_w4(const DeviceScene sc, uint32_t stack_entries, const float4* __restrict__ rays, uint32_t n,
                  uint32_t* __restrict__ head, float* __restrict__ t_out, uint32_t* __restrict__ prim_out,
               unsigned long long* __restrict__ counters)
{
    const uint32_t lane = threadIdx.x & 63u;
    LaneStack2 st;
    st.base = (uint2*)lds_dyn + (threadIdx.x >> 6) * (stack_entries * 64u) + lane;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint4* __restrict__ R = sc.wrecs;
    const TriRecord* __restrict__ T = (const TriRecord*)sc.wrecs;
    uint32_t res_first = 0, res_count = 0;
    bool drained = false;
    uint32_t rid = 0xFFFFFFFFu;
    f3 ro = mk(0.0f), rd = mk(0.0f, 0.0f, 1.0f), rinv = mk(1.0f);
    float rtmin = 0.0f, rtmax = 0.0f, best_t = 0.0f;
    int best_slot = -1; uint32_t best_prim = 0xFFFFFFFFu;
    int node = kSentinel, sp = 0;
    uint32_t cur_base = 0, cur_list = 0;
    bool any_ray = false, any_hit = false;
    unsigned long long n_iter = 0, n_visit = 0, n_tri = 0, n_vround = 0, n_lround = 0;
    for (;;) {
        const bool idle_lane = node == kSentinel;
        unsigned long long idle = vote(idle_lane);
        const unsigned long long busy = ~idle;
        if (idle != 0ull && (popc(idle) >= FETCH_K || busy == 0ull)) {
            if (idle_lane && rid != 0xFFFFFFFFu) {
                if (any_ray) { t_out[rid] = any_hit ? 1.0f : 0.0f; prim_out[rid] = any_hit ? 1u : 0u; }
                else { t_out[rid] = best_slot >= 0 ? best_t : -1.0f; prim_out[rid] = best_prim; }
                rid = 0xFFFFFFFFu;
            }
            while (idle != 0ull && !(drained && res_count == 0u)) {
                if (res_count == 0u) {
                    const uint32_t leader = (uint32_t)__ffsll((long long)idle) - 1u;
                    uint32_t base = 0;
                    if (lane == leader) base = atomicAdd(head, 256u);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
                    res_first = base;
                    res_count = base < n ? min(256u, n - base) : 0u;
                    if (res_count < 256u) drained = true;
                    if (res_count == 0u) break;
                }
                const uint32_t want = (uint32_t)popc(idle);
                const uint32_t take = want < res_count ? want : res_count;
                const uint32_t rank = (uint32_t)popc(idle & below);
                if (idle_lane && rid == 0xFFFFFFFFu && rank < take) {
                    rid = res_first + rank;
                    const float4 a = rays[2ull * rid], b = rays[2ull * rid + 1];
                    ro = mk(a.x, a.y, a.z); rd = mk(a.w, b.x, b.y); rtmin = b.z;
                    any_ray = b.w < 0.0f; rtmax = fabsf(b.w);
                    rinv = mk(fast_rcp(rd.x), fast_rcp(rd.y), fast_rcp(rd.z));
                    best_t = rtmax; best_slot = -1; best_prim = 0xFFFFFFFFu; any_hit = false;
                    node = sc.n_tris ? 0 : kSentinel; sp = 0; cur_list = 0;
                }
                res_first += take; res_count -= take;
                idle = vote(rid == 0xFFFFFFFFu);
            }
            if (vote(node != kSentinel) == 0ull) {
                if (vote(rid != 0xFFFFFFFFu) == 0ull && drained && res_count == 0u) break;
                continue;
            }
        }
        const bool at_leaf = node < 0;
        const bool at_inner = node >= 0 && node != kSentinel;
        const unsigned long long lm = vote(at_leaf);
        const bool leaf_round = lm != 0ull && (popc(lm) >= LEAF_K || vote(at_inner) == 0ull);
        { const unsigned long long vm = vote(at_inner); n_iter++; n_visit += (unsigned long long)popc(vm); n_vround += vm ? 1u : 0u;
          if (leaf_round) { n_tri += (unsigned long long)popc(lm); n_lround++; } }
        bool next = false;
        if (at_inner) {
            uint32_t base;
            const uint32_t list = wide_visit(R, node, ro, rinv, rtmin, best_t, base);
            if (list != 0u) {
                if (cur_list != 0u) { st.push(sp, cur_base, cur_list); sp++; }
                cur_base = base; cur_list = list;
            }
            next = true;
        }
        if (leaf_round && at_leaf) {
            const int slot = ~node;
            const TriRecord* tp = T + slot;
            const float4 r0 = tp->r0, r1 = tp->r1, r2 = tp->r2;
            float t;
            const bool ok = tri_test_lazy(ro, rd, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), rtmin, rtmax, t);
            const uint32_t prim = __float_as_uint(r2.y);
            if (ok) {
                if (any_ray) { any_hit = true; cur_list = 0u; sp = 0; }
                else if (t < best_t || (t == best_t && prim < best_prim)) { best_t = t; best_slot = slot; best_prim = prim; }
            }
            next = true;
        }
        if (next) {
            if (cur_list == 0u && sp > 0) { sp--; const uint2 g = st.pop(sp); cur_base = g.x; cur_list = g.y; }
            if (cur_list != 0u) {
                const uint32_t nib = cur_list & 15u;
                cur_list >>= 4;
                const int idx = (int)(cur_base + (nib & 3u));
                node = (nib & 4u) ? ~idx : idx;
            } else {
                node = kSentinel;
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&counters[0], n_iter); atomicAdd(&counters[1], n_visit); atomicAdd(&counters[2], n_tri);
        atomicAdd(&counters[3], n_vround); atomicAdd(&counters[4], n_lround);
    }
}

// fmt 0: two-child fp32 tree (stack_entries dwords per lane); fmt 1: four-wide tree (stack_entries 8-byte groups);
// fmt 2: two-child fp32 tree, fma slab test; fmt 3: two-child fp16 {lo, hi} nodes; fmt 4: fp16 {centre, half extent} nodes, what the default render kernels walk
typedef void (*StreamKernel)(const DeviceScene, uint32_t, const float4*, uint32_t, uint32_t*, float*, uint32_t*, unsigned long long*);
static StreamKernel stream_kernel(int fmt)
{
    switch (fmt) {
        case 1: return k_trace_stream_w4<8, 8>;
        case 2: return k_trace_stream<8, 8, 2>;
        case 3: return k_trace_stream<8, 8, 3>;
        case 4: return k_trace_stream<8, 8, 4>;
        default: return k_trace_stream<8, 8, 0>;
    }
}
hipError_t launch_trace_stream(int fmt, const DeviceScene& sc, uint32_t stack_entries, const float* d_rays, uint32_t n, uint32_t* d_head,
                               float* d_t, uint32_t* d_prim, unsigned long long* d_counters, uint32_t grid_blocks, hipStream_t stream)
{
    const size_t lds = (size_t)(fmt == 1 ? 8 : 4) * stack_entries * 256u;
    const StreamKernel k = stream_kernel(fmt);
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid_blocks), dim3(256), lds, stream, sc, stack_entries, (const float4*)d_rays, n, d_head, d_t, d_prim, d_counters);
    return hipGetLastError();
}

hipError_t trace_stream_occupancy(int fmt, uint32_t stack_entries, int* blocks_per_cu)
{
    const size_t lds = (size_t)(fmt == 1 ? 8 : 4) * stack_entries * 256u;
    const StreamKernel k = stream_kernel(fmt);
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void*)k, 256, lds);
}

// ---- host-side launchers ------------------------------------------------------------------
// k / kernel: the instantiation with IEEE arithmetic in the shading code; k_fast / kernel_fast: its twin with the arithmetic of the
// reference's own build (pt_set_math_mode; nullptr: the variant exists at the IEEE level only — experiment rows)
struct VariantDesc { RenderKernel k; int threads; int node_fmt; const char* name; int stack_cap = 0; const char* kernel = ""; int wf = -1; int top_n = 0; RenderKernel k_fast = nullptr; const char* kernel_fast = ""; };   // wf >= 0 (experiments build only): index into render_wavefront.hip's table

// Render kernel variants.  0: segment-synchronous (fp32 nodes).  Others: persistent traversal
// <SHADE_K, LEAF_K, NODE_FMT, THREADS, MINW, STATS, DIAG, INNER, LEAVES, LIGHTS, STACK_CAP, TOPN>.  The product library carries the
// variants a user can meaningfully pick (indices fixed: render_megakernel.h); everything that was measured on the way and lost is
// compiled only with -DACGPT_EXPERIMENTS (acgpathtracing_amd/_build.py build_hip(experiments=True), tools/sweep_variants.py).
// PW(...): the instantiation and its name as a kernel trace prints it (all twelve arguments spelled out), so that a profile
// can be tied to the variant that ran (pt_variant_kernel, bench.py).
#define PW(...) k_render_pw<__VA_ARGS__>
#define PWN(...) "k_render_pw<" #__VA_ARGS__ ">"
// a product row: the twelve arguments once, the IEEE instantiation (MATH 0) and its fast-math twin (MATH 1) from them
#define ROW(threads, fmt, name, cap, ...) {PW(__VA_ARGS__, 0), threads, fmt, name, cap, PWN(__VA_ARGS__, 0), -1, 0, PW(__VA_ARGS__, 1), PWN(__VA_ARGS__, 1)}
static const VariantDesc kVariants[] = {
    {k_render<0>, 256, 0, "sync fp32-nodes", 0, "k_render<0>", -1, 0, k_render<1>, "k_render<1>"},
    ROW(256, 0, "pw K44 L16 fp32 nodes w4, register stack top, two visits and two triangle tests per loop trip", 0, 44, 16, 0, 256, 4, false, 0, 2, 2, false, 0, 0),
    ROW(256, 0, "pw K44 L16 fp32 nodes + scheduler stats", 0, 44, 16, 0, 256, 4, true, 0, 2, 2, false, 0, 0),
    ROW(256, 0, "pw K48 L8 fp32 nodes w4, two visits per loop trip (large scenes whose fp16 planes would be too coarse)", 0, 48, 8, 0, 256, 4, false, 0, 2, 1, false, 0, 0),
    ROW(256, 0, "TRIG fp32 nodes w4 with the cosine sampler's sin / cos / acos on v_sin_f32 / v_cos_f32 / sqrt (IEEE mode: everything else IEEE; other bits than its neighbours there)", 0, 48, 12, 0, 256, 4, false, 3, 1, 1, false, 0, 0),
    ROW(256, 8, "pw K44 L16 fp16 nodes (32 B), sign-rotated v_fma_mix planes, w4, three visits and two triangle tests per loop trip", 0, 44, 16, 8, 256, 4, false, 0, 3, 2, false, 0, 0),
    ROW(256, 11, "pw K44 L16 fp16 centre / half-extent nodes, six visits per trip + scheduler stats (the default kernel's loop at four waves)", 0, 44, 16, 11, 256, 4, true, 0, 6, 2, false, 0, 0),
    ROW(256, 11, "pw K40 L16 fp16 nodes (32 B) as centre / half extent per axis: two v_fma_mix_f32 and a full-rate subtract / add per axis and child, no rotates; FIVE waves per SIMD (96 registers), six visits and two triangle tests per loop trip", 0, 40, 16, 11, 256, 5, false, 0, 6, 2, false, 0, 0),
    ROW(256, 11, "LIGHTS scene-driven area lights + MIS (light mode 1, opt-in: not the reference's estimator), fp16 centre / half-extent nodes w4", 0, 44, 16, 11, 256, 4, false, 0, 5, 2, true, 0, 0),
    ROW(256, 11, "pw K24 L16 fp16 centre / half-extent nodes, five waves per SIMD, for large scenes and deep trees: shade rounds at 24 parked lanes (rays are long there), a sliding window of 16 stack entries per lane in LDS, deeper ones moved to global memory four at a time", -16, 24, 16, 11, 256, 5, false, 0, 5, 2, false, -16, 0),
#ifdef ACGPT_EXPERIMENTS
#include "render_experiments.inc"
#endif
};
int render_variant_count() { return (int)(sizeof(kVariants) / sizeof(kVariants[0])); }
// a table row, with the rows that stand for a wavefront kernel filled in from that kernel's own description
static VariantDesc variant_desc(int v)
{
    VariantDesc d = kVariants[v];
#ifdef ACGPT_EXPERIMENTS
    if (d.wf >= 0) {
        const WfDesc* w = wf_variant(d.wf);
        d.k = w->k; d.threads = (w->nt + w->ns) * 64; d.name = w->name; d.stack_cap = w->stack_cap; d.kernel = w->kernel;
        d.k_fast = w->k_fast; d.kernel_fast = w->kernel_fast;
    }
#endif
    return d;
}
// the instantiation a math mode runs: the fast twin where the variant has one (experiment rows exist at the IEEE level only)
static RenderKernel variant_kernel(const VariantDesc& d, int math) { return (math != 0 && d.k_fast != nullptr) ? d.k_fast : d.k; }
const char* render_variant_name(int v) { return (v >= 0 && v < render_variant_count()) ? variant_desc(v).name : "?"; }
int render_variant_node_format(int v) { return (v >= 0 && v < render_variant_count()) ? kVariants[v].node_fmt : -1; }
int render_variant_threads(int v) { return (v >= 0 && v < render_variant_count()) ? variant_desc(v).threads : 0; }
int render_variant_stack_cap(int v) { return (v >= 0 && v < render_variant_count()) ? variant_desc(v).stack_cap : 0; }
int render_variant_top_nodes(int v) { return (v >= 0 && v < render_variant_count()) ? kVariants[v].top_n : 0; }
const char* render_variant_kernel(int v, int math)
{
    if (v < 0 || v >= render_variant_count()) return "";
    const VariantDesc d = variant_desc(v);
    return (math != 0 && d.k_fast != nullptr) ? d.kernel_fast : d.kernel;
}
int render_variant_has_fast_math(int v) { return (v >= 0 && v < render_variant_count()) ? variant_desc(v).k_fast != nullptr : 0; }

static size_t variant_lds(const VariantDesc& d, uint32_t stack_entries, uint32_t n_nodes)
{
#ifdef ACGPT_EXPERIMENTS
    if (d.wf >= 0) return wf_lds_bytes(*wf_variant(d.wf), stack_entries);
#endif
    if (d.stack_cap > 0 && stack_entries > (uint32_t)d.stack_cap) stack_entries = (uint32_t)d.stack_cap;
    if (d.stack_cap < 0) stack_entries = (uint32_t)(-d.stack_cap) + 1u;      // sliding window: that many entries, whatever the tree, + the window base
    const uint32_t ent = (d.node_fmt == 10 || d.node_fmt == 12) ? 2u : 1u;     // the shared-plane kernel's stack entries are 8 bytes
    size_t lds = (size_t)(d.threads / 64) * (stack_entries * 256u * ent + kBookDwords * 4u) + 256u + (size_t)d.top_n * sizeof(HNode);      // lane stacks, fold bookkeeping, LCG skip-ahead table, staged top of the tree
    if (d.node_fmt == 2) lds += (size_t)n_nodes * sizeof(QNode);
    if (d.node_fmt == 13) lds += (size_t)n_nodes * 16u;
    if (d.node_fmt == 14) lds += (size_t)n_nodes * 32u;
    return lds;
}

hipError_t render_occupancy(int variant, int math, uint32_t stack_entries, uint32_t n_nodes, int* blocks_per_cu)
{
    if (variant < 0 || variant >= render_variant_count()) return hipErrorInvalidValue;
    VariantDesc d = variant_desc(variant);
    d.k = variant_kernel(d, math);
    const size_t lds = variant_lds(d, stack_entries, n_nodes);
    *blocks_per_cu = 0;
    if (lds > 160u * 1024u) return hipSuccess;      // does not fit: 0 blocks, caller reports it
    hipError_t e = hipFuncSetAttribute((const void*)d.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void*)d.k, d.threads, lds);
}

hipError_t launch_render(int variant, int math, const RenderArgs& args, uint32_t grid_blocks, hipStream_t stream)
{
    if (variant < 0 || variant >= render_variant_count()) return hipErrorInvalidValue;
    VariantDesc d = variant_desc(variant);
    d.k = variant_kernel(d, math);
    const size_t lds = variant_lds(d, args.stack_entries, args.n_lds_nodes);
    RenderArgsBox box;
    box.a[0] = args;
    hipLaunchKernelGGL(d.k, dim3(grid_blocks), dim3(d.threads), lds, stream, box);
    return hipGetLastError();
}

hipError_t launch_trace_closest(const DeviceScene& sc, uint32_t stack_entries, const float* d_rays, uint32_t n,
                                float* d_t, uint32_t* d_prim, hipStream_t stream)
{
    const size_t lds = (size_t)4 * stack_entries * 256u;
    hipError_t e = hipFuncSetAttribute((const void*)k_trace_closest, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_trace_closest<<<(n + 255) / 256, 256, lds, stream>>>(sc, stack_entries, d_rays, n, d_t, d_prim);
    return hipGetLastError();
}

hipError_t launch_trace_any(const DeviceScene& sc, uint32_t stack_entries, const float* d_rays, uint32_t n,
                            uint8_t* d_hit, hipStream_t stream)
{
    const size_t lds = (size_t)4 * stack_entries * 256u;
    hipError_t e = hipFuncSetAttribute((const void*)k_trace_any, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_trace_any<<<(n + 255) / 256, 256, lds, stream>>>(sc, stack_entries, d_rays, n, d_hit);
    return hipGetLastError();
}

}  // namespace ptd
