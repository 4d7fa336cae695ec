// render_wavefront.hip — the per-pixel Monte-Carlo launch as a WORKGROUP-level wavefront kernel for gfx950.
//
// Same work, same bits as k_render_pw (render_megakernel.hip): replaces __raygen__rg / __closesthit__diffuse__ch / __miss__ms
// (PathTracer_Optix/pathTracerPrograms.cu:707-816, 866-1031, 833-847) and the OptiX traversal under them.  What differs is who
// executes what.  In k_render_pw a lane owns a pixel run from its first camera ray to its last: it traverses, parks until
// enough lanes of its wave have a finished ray, shades, traverses again — a BVH visit runs at 38 of 64 lanes, the closest-hit
// code at about half of a shade round's lanes.  Here the waves of a workgroup SPECIALISE and exchange rays through LDS:
//
//   * a ray and its path are one RECORD of 23 dwords (pixel, run, sums, throughput, PRNG, ray, hit, what the closest-hit left
//     for after the shadow ray);
//   * TRACE waves hold one record per lane in registers and do nothing but the BVH loop.  A lane whose ray is finished swaps
//     its record, inside the loop, against a ready one from the pool (field by field through the slot the ready one sits in),
//     so the loop's lanes stay full;
//   * SHADE waves take 64 records of one kind at a time — rays that hit something (closest-hit shading, light sample) or rays
//     that need accounting (shadow ray back, miss: roulette, next bounce or next camera path) — so the long closest-hit code
//     runs on full waves of hits;
//   * runs that have used up their samples are folded, and new work items taken from the launch's queue, 64 at a time by one
//     wave under a workgroup lock (the DEAL round: the item decode, the fold slots and the free lists need no atomics).
//
// Five rings in LDS carry slot numbers between the roles: ready to trace, hit to shade, to account, run finished, empty.
// Every lane performs its own record's operations in the reference's order, and the runs of a pixel are summed in run order
// as before: the image is bit-identical to every other kernel variant (test_every_kernel_variant_gives_the_same_bits).
//
// Safety: every wait in here is on a queue another resident wave of the same workgroup feeds without waiting for us; spins are
// bounded and a watchdog on the shader clock ends the kernel with an error flag (pt_launch fails) instead of hanging the GPU.
#include "render_common.h"

namespace ptd {

constexpr uint32_t kWfNone = 0xFFFFFFFFu;

// ---- the record -------------------------------------------------------------------------------------------------------------
enum : uint32_t { F_PXY = 0, F_LSEED, F_RUN, F_TAGF, F_RESX, F_RESY, F_RESZ, F_PSEED, F_ATTX, F_ATTY, F_ATTZ,
                  F_ROX, F_ROY, F_ROZ, F_RDX, F_RDY, F_RDZ, F_TMAX, F_HIT, F_NDX, F_NDY, F_NDZ, F_WEIGHT, F_COUNT };
// F_RUN : samples left in the run (16 bits) | fold slot of the run's group << 16 (0xFFFF: one run per pixel, no slot)
// F_TAGF: sub = sub-frame << chunk_shift | run (16 bits) | depth << 16 (5 bits) | flags
// F_TMAX: the ray's tmax when it is ready to trace, the closest hit's distance when it comes back
// F_HIT : radiance ray back: leaf slot of the closest hit, -1 = miss; shadow ray back: 0 = occluded, -1 = free
// F_ND* : while a shadow ray is in flight: the direction of the next bounce (pd.nxt_dir) — or, for a path that ends on an
//         emitter (kTagDone), the emitter's Ke, which the light sample is added to (:992-1000, 1015-1024); F_WEIGHT = pd.weight.
//         The next bounce's origin is recomputed from the shadow ray's origin P: P for a diffuse hit, P + R * 1e-4 for a conductor (kTagMetal).
constexpr uint32_t kTagDepthShift = 16u, kTagShadow = 1u << 21, kTagDone = 1u << 22, kTagMetal = 1u << 23;
enum : int { QT = 0, QH = 1, QM = 2, QR = 3, QF = 4, kRings = 5 };   // ready to trace, hit to shade, to account, run finished, empty slot

struct WfRing { uint32_t head, tail; int count; uint32_t pad; };
struct WfCtl {
    WfRing q[kRings];
    uint32_t deal_lock, done, abort_flag, drained;
    int n_live;                                    // runs dealt and not yet folded
    // the dealer's state between DEAL rounds (owned by whoever holds deal_lock)
    uint32_t shard, shards_left, res_first, res_count, grant_g0, free_top;
    uint32_t grp_pxy[64], grp_seed[64];
    uint16_t gslot[64];
};

__device__ __forceinline__ int lds_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Lanes with `p` append `v`: one reservation per wave, cells written, then the count published (a consumer that finds a
// cell of its reservation still empty waits for it: its producer is between these two steps).
template <uint32_t CAP>
__device__ __forceinline__ void ring_push(WfRing* r, uint32_t* cells, bool p, uint32_t v, uint32_t lane, unsigned long long below)
{
    const unsigned long long m = vote(p);
    if (m == 0ull) return;
    const uint32_t n = (uint32_t)popc(m), leader = (uint32_t)__ffsll((long long)m) - 1u;
    uint32_t pos = 0;
    if (lane == leader) pos = __hip_atomic_fetch_add(&r->tail, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)leader);
    if (p) __hip_atomic_store(&cells[(pos + (uint32_t)popc(m & below)) & (CAP - 1u)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (lane == leader) __hip_atomic_fetch_add(&r->count, (int)n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Lanes with `want` take an entry each while there are any (kWfNone otherwise).
template <uint32_t CAP>
__device__ __forceinline__ uint32_t ring_pop(WfRing* r, uint32_t* cells, bool want, uint32_t lane, unsigned long long below, bool& trouble)
{
    const unsigned long long m = vote(want);
    if (m == 0ull) return kWfNone;
    const int n = popc(m);
    const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
    uint32_t got = 0, pos = 0;
    if (lane == leader && lds_load(&r->count) > 0) {
        const int c = __hip_atomic_fetch_sub(&r->count, n, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        got = c <= 0 ? 0u : (uint32_t)(c < n ? c : n);
        if ((int)got < n) __hip_atomic_fetch_add(&r->count, n - (int)got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (got) pos = __hip_atomic_fetch_add(&r->head, got, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    got = (uint32_t)__builtin_amdgcn_readlane((int)got, (int)leader);
    pos = (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)leader);
    uint32_t v = kWfNone;
    const uint32_t rank = (uint32_t)popc(m & below);
    if (want && rank < got) {
        uint32_t* c = &cells[(pos + rank) & (CAP - 1u)];
        uint32_t spins = 0;
        do { v = __hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); } while (v == kWfNone && ++spins < (1u << 20));
        __hip_atomic_store(c, kWfNone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v == kWfNone) trouble = true;
    }
    return v;
}

// ---- fold bookkeeping of a workgroup (the DEAL round holds the lock: plain LDS accesses, one wave at a time) -----------------
struct WgBook {
    uint32_t* tick;       // [FOLD / 4] runs parked so far, one byte per fold slot
    uint16_t* free16;     // [FOLD] stack of free fold slots
    uint16_t* gslot;      // [64] fold slot of group grant_g0 + j of the current grant
    __device__ __forceinline__ uint32_t bump(uint32_t slot) const
    { return (atomicAdd(&tick[slot >> 2], 1u << (8u * (slot & 3u))) >> (8u * (slot & 3u))) & 0xFFu; }
    __device__ __forceinline__ void clear(uint32_t slot) const { atomicAnd(&tick[slot >> 2], ~(0xFFu << (8u * (slot & 3u)))); }
};
constexpr uint32_t kNoFold = 0xFFFFu;

// what a lane of the DEAL round holds of its record
struct RunLane {
    bool alive, new_path;
    uint32_t pxy, seed, samples_left, sub, fold;
    f3 result;
};

// finish_runs of render_common.h for a workgroup's book: lanes with `finished` park their run's sum; the lane that completes
// a group adds its runs in run order, writes the (pixel, sub-frame) sum and frees the fold slot.
__device__ __forceinline__ void wf_finish_runs(const RenderArgs& A, QueueState& q, unsigned long long below, const RunLane& lp, bool finished,
                                               const WgBook& book, float* __restrict__ scratch)
{
    if (vote(finished) == 0ull) return;
    const uint32_t cs = A.chunk_shift, runs = 1u << cs;
    if (cs == 0u) {
        if (finished) write_frame_sum(A, lp.pxy, lp.sub, lp.result);
        return;
    }
    float* group = scratch + 3u * ((size_t)lp.fold << cs);
    bool folder = false;
    if (finished) {
        float* mine = group + 3u * (lp.sub & (runs - 1u));
        mine[0] = lp.result.x; mine[1] = lp.result.y; mine[2] = lp.result.z;
        folder = book.bump(lp.fold) == runs - 1u;
    }
    const unsigned long long folders = vote(folder);
    if (folders == 0ull) return;
    // partial sums of earlier DEAL rounds were stored by other waves of this workgroup (same CU, complete before they released
    // the lock); ours of this round: wait for the stores, then read everything back past the L1 (sc0)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0) only
    if (folder) {
        f3 sum = mk(0.0f);
        if (runs == 2u) {
            v3f_t p0, p1;
            load2_coherent(group, p0, p1);
            sum = mk(p0.x, p0.y, p0.z);
            sum += mk(p1.x, p1.y, p1.z);
        } else {
            for (uint32_t k0 = 0; k0 < runs; k0 += 4u) {
                v3f_t p0, p1, p2, p3;
                load4_coherent(group + 3u * k0, p0, p1, p2, p3);
                if (k0 == 0u) sum = mk(p0.x, p0.y, p0.z); else sum += mk(p0.x, p0.y, p0.z);
                sum += mk(p1.x, p1.y, p1.z);
                sum += mk(p2.x, p2.y, p2.z);
                sum += mk(p3.x, p3.y, p3.z);
            }
        }
        write_frame_sum(A, lp.pxy, lp.sub >> cs, sum);
        book.clear(lp.fold);
        book.free16[q.free_top + (uint32_t)popc(folders & below)] = (uint16_t)lp.fold;
    }
    q.free_top += (uint32_t)popc(folders);
}

// refill_lanes of render_common.h for a workgroup's book.  `idle_in`: lanes of the round that need an item.
__device__ __forceinline__ void wf_refill(const RenderArgs& A, QueueState& q, uint32_t lane, unsigned long long below, RunLane& lp,
                                          const uint32_t* lcg_skip, const WgBook& book)
{
    const uint32_t cs = A.chunk_shift, run_mask = (1u << cs) - 1u;
    const uint32_t fshift = A.sub_shift - cs, fmask = (1u << fshift) - 1u;
    unsigned long long idle = vote(!lp.alive);
    while (idle != 0ull && (q.res_count != 0u || q.shards_left != 0u)) {
        if (q.res_count == 0u) {
            const uint32_t leader = (uint32_t)__ffsll((long long)idle) - 1u;
            const uint32_t idle_n = (uint32_t)popc(idle);
            uint32_t req = idle_n > A.grant ? idle_n : A.grant;
            req = (req + run_mask) & ~run_mask;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&A.queue_heads[q.shard], req);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
            const uint32_t shard_begin = q.shard * A.shard_size;
            uint32_t shard_end = shard_begin + A.shard_size;
            if (shard_end > A.total_samples) shard_end = A.total_samples;
            if (shard_begin > A.total_samples) shard_end = shard_begin;
            const uint32_t first = shard_begin + base;
            uint32_t avail = first < shard_end ? shard_end - first : 0u;
            if (avail > req) avail = req;
            if (avail < req) { q.shard = (q.shard + 1u) & 7u; q.shards_left--; }
            q.res_first = first; q.res_count = avail;
            if (avail == 0u) continue;
            q.grant_g0 = first >> cs;
            const uint32_t g = q.grant_g0 + lane;
            uint32_t x, y;
            sample_pixel_fast(A, queue_slot(A, g >> fshift), x, y);
            const uint32_t f = g & fmask;
            const bool ok = (g << cs) < first + avail && x < A.width && y < A.height && f < A.n_frames;
            q.grp_pxy = ok ? (x | (y << 16)) : 0xFFFFFFFFu;
            q.grp_seed = tea4(y * A.width + x, A.frame + f);
        }
        const uint32_t want = (uint32_t)popc(idle);
        const uint32_t take = want < q.res_count ? want : q.res_count;
        const uint32_t rank = (uint32_t)popc(idle & below);
        const uint32_t item = q.res_first + rank;
        {
            const uint32_t run = item & run_mask;
            const uint32_t gj = (item >> cs) - q.grant_g0;
            const int src = (int)(gj << 2);
            const uint32_t pxy = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)q.grp_pxy);
            const uint32_t seed0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)q.grp_seed);
            const bool mine = !lp.alive && rank < take && pxy != 0xFFFFFFFFu;
            uint32_t sl = kNoFold;
            if (cs != 0u) {
                const unsigned long long opens = vote(mine && run == 0u);
                if (opens != 0ull) {
                    if (mine && run == 0u) book.gslot[gj] = book.free16[q.free_top - 1u - (uint32_t)popc(opens & below)];
                    q.free_top -= (uint32_t)popc(opens);
                }
                if (mine) sl = (uint32_t)book.gslot[gj];
            }
            if (mine) {
                lp.pxy = pxy;
                lp.sub = (((item >> cs) & fmask) << cs) | run;
                lp.fold = sl;
                lp.seed = lcg_skip[2u * run] * seed0 + lcg_skip[2u * run + 1u];
                lp.result = mk(0.0f);
                lp.samples_left = A.chunk_spp;
                lp.alive = true;
                lp.new_path = true;
            }
        }
        q.res_first += take; q.res_count -= take;
        idle = vote(!lp.alive);
    }
}

// =================================================================================================================================
// NT trace waves + NS shade waves per workgroup; POOL record slots (a power of two); STACK_CAP entries of a lane's traversal
// stack in LDS (deeper ones in global memory); a trace wave exchanges records when FETCH_K lanes are idle; triangle rounds at
// LEAF_K lanes; VISITS node visits per trip through the loop control.  fp16 nodes (HNode, the slab test of NODE_FMT 9).
// =================================================================================================================================
template <int NT, int NS, int POOL, int STACK_CAP, int FETCH_K, int LEAF_K, int VISITS, int MINB, bool DIAG, int MATH = 0>
__global__ void __launch_bounds__((NT + NS) * 64, MINB)
k_render_wf(const RenderArgsBox B)
{
    constexpr int FM = MATH ? 2 : 0;                  // arithmetic level of the shading code (pt_device.h; pt_set_math_mode)
    constexpr int THREADS = (NT + NS) * 64;
    constexpr uint32_t FOLD = (uint32_t)((NT * 64 + POOL + 64 + 3) & ~3);
    static_assert((POOL & (POOL - 1)) == 0, "ring indices wrap by masking");
    static_assert(FOLD <= (uint32_t)(THREADS / 64) * kRenderFoldSlots, "the launch sizes the partial-sum scratch as 128 fold slots per wave");
    const RenderArgs& A = B.a[0];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // ---- LDS ----------------------------------------------------------------------------------------------------------------
    WfCtl* const ctl = (WfCtl*)lds_dyn;
    uint32_t* lp_ = lds_dyn + (sizeof(WfCtl) + 3u) / 4u;
    uint32_t* const lcg_skip = lp_;                 lp_ += 64;
    uint32_t* const tick = lp_;                     lp_ += FOLD / 4u;
    uint16_t* const free16 = (uint16_t*)lp_;        lp_ += FOLD / 2u;
    uint32_t* const cells = lp_;                    lp_ += kRings * POOL;
    uint32_t* const pool = lp_;                     lp_ += F_COUNT * POOL;
    uint32_t* const stacks = lp_;
    const uint32_t lds_entries = A.stack_entries > (uint32_t)STACK_CAP ? (uint32_t)STACK_CAP : A.stack_entries;

    for (uint32_t i = threadIdx.x; i < (sizeof(WfCtl) + 3u) / 4u; i += THREADS) lds_dyn[i] = 0u;
    for (uint32_t i = threadIdx.x; i < FOLD / 4u; i += THREADS) tick[i] = 0u;
    for (uint32_t i = threadIdx.x; i < FOLD; i += THREADS) free16[i] = (uint16_t)i;
    for (uint32_t i = threadIdx.x; i < kRings * POOL; i += THREADS) cells[i] = i >= QF * POOL ? i - QF * POOL : kWfNone;     // every slot starts empty
    if (threadIdx.x < 32u) { lcg_skip[2u * threadIdx.x] = A.lcg_mul[threadIdx.x]; lcg_skip[2u * threadIdx.x + 1u] = A.lcg_add[threadIdx.x]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ctl->q[QF].tail = POOL; ctl->q[QF].count = POOL;
        ctl->shard = xcc_id(); ctl->shards_left = 8u; ctl->free_top = FOLD;
    }
    if (threadIdx.x < 64u) { ctl->grp_pxy[threadIdx.x] = 0xFFFFFFFFu; ctl->grp_seed[threadIdx.x] = 0u; }
    __syncthreads();

    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    constexpr unsigned long long kWatchdogTicks = 2000000000ull;       // 20 s at 100 MHz: a launch takes 0.1 ... 1 s
    unsigned long long n_radiance = 0, n_shadow = 0, n_paths = 0, n_pixels = 0, n_culled = 0;
    unsigned long long n_steps = 0, n_lane_steps = 0, n_rounds = 0, n_lane_rounds = 0;
    // diagnostics (10 ns ticks and counts, summed over the waves of the grid): trace waves: total, idle; shade waves: total, idle,
    // deal / hit / accounting rounds: time, rounds, records
    unsigned long long d_total = 0, d_idle = 0, d_deal = 0, d_hit = 0, d_acct = 0, c_deal = 0, c_hit = 0, c_acct = 0, l_deal = 0, l_hit = 0, l_acct = 0;
    bool trouble = false;
    const auto late = [&]() -> const RenderArgs& { return B.a[opaque_zero()]; };
    const auto poll_stop = [&]() -> bool {
        if (lds_load(&ctl->abort_flag) != 0u) return true;
        if (__builtin_amdgcn_s_memrealtime() - t_begin > kWatchdogTicks) { if (lane == 0) __hip_atomic_store(&ctl->abort_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); return true; }
        return false;
    };

    if (wave >= (uint32_t)NS) {
        // ======================================= TRACE wave ==================================================================
        const uint32_t tw = wave - (uint32_t)NS;
        const HNode* const hnodes = A.scene.hnodes;
        const TriRecord* const tris = A.scene.tris;
        const int root = A.scene.n_tris ? 0 : kSentinel;
        LaneStack st;
        st.base = stacks + tw * (lds_entries * 64u) + lane;
        uint32_t* const ovf = A.stack_entries > lds_entries
            ? A.stack_overflow + (size_t)(blockIdx.x * (THREADS / 64) + wave) * 64u * (A.stack_entries - lds_entries) + lane : nullptr;
        const auto push = [&](int at, int v) { if (at < (int)lds_entries) st.push(at, v); else ovf[(at - (int)lds_entries) * 64] = (uint32_t)v; };
        const auto pop = [&](int at) -> int { if (at < (int)lds_entries) return st.pop(at); return (int)ovf[(at - (int)lds_entries) * 64]; };
        uint32_t rec[F_COUNT];
#pragma unroll
        for (int f = 0; f < (int)F_COUNT; f++) rec[f] = 0u;
        bool has_rec = false, shadow_ray = false, shadow_hit = false;
        f3 ro = mk(0.0f), rd = mk(0.0f, 0.0f, 1.0f), rinv = mk(1.0f), gro = mk(0.0f);
        constexpr float rtmin = 0.01f;
        float best_t = 0.0f;
        int best_slot = -1; uint32_t best_prim = 0xFFFFFFFFu;
        int node = kSentinel, sp = 0, tos = kSentinel;
        int cooldown = 0;
        for (;;) {
            const bool act = node != kSentinel;
            const unsigned long long am = vote(act);
            const int n_idle = 64 - popc(am);
            if ((n_idle >= FETCH_K && cooldown <= 0) || am == 0ull) {
                // ---- exchange: finished records out, ready records in -------------------------------------------------------
                const unsigned long long t_ex = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;
                const bool idle = !act;
                const bool fin = idle && has_rec;
                uint32_t s = ring_pop<POOL>(&ctl->q[QT], cells + QT * POOL, idle, lane, below, trouble);
                const bool swap_in = s != kWfNone;
                if (vote(fin && !swap_in) != 0ull) {                       // nothing ready for them: drop the record into an empty slot
                    const uint32_t s2 = ring_pop<POOL>(&ctl->q[QF], cells + QF * POOL, fin && !swap_in, lane, below, trouble);
                    if (s2 != kWfNone) s = s2;
                }
                const bool moved = s != kWfNone;
                const bool is_hit = !shadow_ray && best_slot >= 0;
                if (fin && moved) {
                    rec[F_TMAX] = __float_as_uint(best_t);
                    rec[F_HIT] = shadow_ray ? (shadow_hit ? 0u : 0xFFFFFFFFu) : (uint32_t)best_slot;
                }
                if (moved) {
                    uint32_t* const at = pool + s;
#pragma unroll
                    for (int f = 0; f < (int)F_COUNT; f++) {
                        uint32_t in = 0u;
                        if (swap_in) in = at[f * POOL];
                        if (fin) at[f * POOL] = rec[f];
                        if (swap_in) rec[f] = in;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                ring_push<POOL>(&ctl->q[QH], cells + QH * POOL, fin && moved && is_hit, s, lane, below);
                ring_push<POOL>(&ctl->q[QM], cells + QM * POOL, fin && moved && !is_hit, s, lane, below);
                ring_push<POOL>(&ctl->q[QF], cells + QF * POOL, !fin && swap_in, s, lane, below);      // took a record, had none to leave: the slot is empty now
                if (fin && moved) has_rec = false;
                if (swap_in) {
                    has_rec = true;
                    ro = mk(__uint_as_float(rec[F_ROX]), __uint_as_float(rec[F_ROY]), __uint_as_float(rec[F_ROZ]));
                    rd = mk(__uint_as_float(rec[F_RDX]), __uint_as_float(rec[F_RDY]), __uint_as_float(rec[F_RDZ]));
                    setup_ray_h9(ro, rd, late().scene.hspace, rinv, gro);
                    shadow_ray = (rec[F_TAGF] & kTagShadow) != 0u;
                    shadow_hit = false;
                    best_t = __uint_as_float(rec[F_TMAX]); best_slot = -1; best_prim = 0xFFFFFFFFu;
                    node = root; sp = 0;
                }
                const unsigned long long now_active = vote(node != kSentinel);
                cooldown = vote(swap_in) != 0ull ? 0 : 4;                  // nothing came: look again a few trips later
                if (DIAG) { d_deal += __builtin_amdgcn_s_memrealtime() - t_ex; c_deal += 1; l_deal += (unsigned long long)popc(vote(swap_in)); }
                if (now_active == 0ull) {
                    if (vote(has_rec) == 0ull && lds_load(&ctl->done) != 0u) break;
                    if (vote(trouble) != 0ull) { if (lane == 0) __hip_atomic_store(&ctl->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                    if (poll_stop()) break;
                    if (DIAG) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_sleep(8); d_idle += __builtin_amdgcn_s_memrealtime() - t0; } else __builtin_amdgcn_s_sleep(8);
                    continue;
                }
            }
            cooldown--;
            n_steps += 1; n_lane_steps += (unsigned long long)popc(vote(node != kSentinel));
            const float rtmax = shadow_ray ? best_t : 1e16f;               // a shadow ray's interval end never moves; a radiance ray's is open (:750-757)
#pragma unroll
            for (int visit = 0; visit < VISITS; visit++)
            if ((uint32_t)node < (uint32_t)kSentinel) {
                float n0, f0, n1, f1; int c0, c1;
                const uint4* np = (const uint4*)((const char*)hnodes + (size_t)((uint32_t)node << 5));
                const uint4 qa = np[0], qb = np[1];
                c0 = (int)qa.w; c1 = (int)qb.w;
                slab_h9(qa.x, qa.y, qa.z, rinv, gro, rtmin, n0, f0);
                slab_h9(qb.x, qb.y, qb.z, rinv, gro, rtmin, n1, f1);
                f0 = fminf(f0, best_t * kTieWiden);
                f1 = fminf(f1, best_t * kTieWiden);
                const bool h0 = n0 <= f0, h1 = n1 <= f1;
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
            const bool at_leaf = node < 0;
            const unsigned long long lm = vote(at_leaf);
            if (lm != 0ull && (popc(lm) >= LEAF_K || vote(node >= 0 && node != kSentinel) == 0ull)) {
#pragma unroll
                for (int leaf = 0; leaf < 2; leaf++)
                if (node < 0) {
                    const int slot = ~node;
                    const TriRecord* tp = (const TriRecord*)((const char*)tris + (size_t)((uint32_t)slot * 48u));
                    const float4 r0 = tp->r0, r1 = tp->r1, r2 = tp->r2;
                    float t;
                    const bool ok = tri_test_lazy(ro, rd, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), rtmin, rtmax, t);
                    const uint32_t prim = __float_as_uint(r2.y);
                    bool stop = false;
                    if (ok) {
                        if (shadow_ray) { shadow_hit = true; stop = true; }
                        else if (t < best_t || (t == best_t && prim < best_prim)) { best_t = t; best_slot = slot; best_prim = prim; }
                    }
                    node = (stop || sp == 0) ? kSentinel : tos;
                    sp = sp ? sp - 1 : 0;
                    tos = pop(sp);
                }
            }
        }
    } else {
        // ======================================= SHADE wave ==================================================================
        WgBook book; book.tick = tick; book.free16 = free16; book.gslot = ctl->gslot;
        constexpr int kLowWater = POOL / 4;            // ready rays below which the trace waves are about to run dry: shade whatever is there
        for (;;) {
            if (lds_load(&ctl->done) != 0u) break;
            if (vote(trouble) != 0ull) { if (lane == 0) __hip_atomic_store(&ctl->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
            if (poll_stop()) break;
            const int nt = lds_load(&ctl->q[QT].count), nh = lds_load(&ctl->q[QH].count), nm = lds_load(&ctl->q[QM].count);
            const int nr = lds_load(&ctl->q[QR].count), nf = lds_load(&ctl->q[QF].count);
            const bool drained = lds_load(&ctl->drained) != 0u;
            const int n_deal = nr + (drained ? 0 : nf);
            const int thr = nt < kLowWater ? 1 : 64;
            int choice = -1;                                  // 0 deal, 1 hits, 2 accounting
            if (n_deal >= thr && n_deal >= nh && n_deal >= nm) {
                uint32_t got = 0;
                if (lane == 0) got = __hip_atomic_exchange(&ctl->deal_lock, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u ? 1u : 0u;
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                if (got) choice = 0;
            }
            if (choice < 0) {
                if (nh >= thr && nh >= nm) choice = 1;
                else if (nm >= thr) choice = 2;
                else if (nh >= thr) choice = 1;
            }
            if (choice < 0) { if (DIAG) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_sleep(4); d_idle += __builtin_amdgcn_s_memrealtime() - t0; } else __builtin_amdgcn_s_sleep(4); continue; }
            const unsigned long long t_round = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;

            if (choice == 0) {
                // ---------------------------------- DEAL round (lock held) ----------------------------------------------------
                const RenderArgs& Ad = late();          // the launch constants of the deal are read here, not held across the kernel
                float* const scratch = Ad.wave_scratch + 3u * (size_t)(blockIdx.x * (THREADS / 64)) * ((size_t)kRenderFoldSlots << Ad.chunk_shift);
                QueueState q;
                q.shard = ctl->shard; q.shards_left = ctl->shards_left; q.res_first = ctl->res_first; q.res_count = ctl->res_count;
                q.grant_g0 = ctl->grant_g0; q.free_top = ctl->free_top; q.grp_pxy = ctl->grp_pxy[lane]; q.grp_seed = ctl->grp_seed[lane];
                int live_delta = 0;
                uint32_t slot = ring_pop<POOL>(&ctl->q[QR], cells + QR * POOL, true, lane, below, trouble);
                const bool from_run = slot != kWfNone;
                if (q.res_count != 0u || q.shards_left != 0u) {
                    const uint32_t s2 = ring_pop<POOL>(&ctl->q[QF], cells + QF * POOL, !from_run, lane, below, trouble);
                    if (!from_run) slot = s2;
                }
                const bool have = slot != kWfNone;
                RunLane lp; lp.alive = !have; lp.new_path = false; lp.pxy = lp.seed = lp.samples_left = lp.sub = 0u; lp.fold = kNoFold; lp.result = mk(0.0f);
                if (from_run) {
                    const uint32_t* at = pool + slot;
                    lp.pxy = at[F_PXY * POOL];
                    lp.sub = at[F_TAGF * POOL] & 0xFFFFu;
                    lp.fold = at[F_RUN * POOL] >> 16;
                    lp.result = mk(__uint_as_float(at[F_RESX * POOL]), __uint_as_float(at[F_RESY * POOL]), __uint_as_float(at[F_RESZ * POOL]));
                }
                bool finished = from_run;
                f3 ro = mk(0.0f), rd = mk(0.0f, 0.0f, 1.0f);
                for (int pass = 0; pass < 8; pass++) {
                    const int n_fin = popc(vote(finished));
                    n_pixels += (unsigned long long)n_fin; live_delta -= n_fin;
                    wf_finish_runs(Ad, q, below, lp, finished, book, scratch);
                    finished = false;
                    const unsigned long long before = vote(lp.alive);
                    wf_refill(Ad, q, lane, below, lp, lcg_skip, book);
                    live_delta += popc(vote(lp.alive) & ~before);
                    // camera path start (:727-745) with the scene-box cull, as in k_render_pw
                    uint32_t my_culled = 0u;
                    if (have && lp.alive && lp.new_path) {
                        const RenderArgs& Rc = late();
                        const f3 eye = mk(Rc.eye), camU = mk(Rc.U), camV = mk(Rc.V), camW = mk(Rc.W);
                        const float fw = (float)(int)Rc.width, fh = (float)(int)Rc.height;
                        const f3 elo = Rc.scene.n_tris ? mk(Rc.cull_lo) - eye : mk(1.0f), ehi = Rc.scene.n_tris ? mk(Rc.cull_hi) - eye : mk(-1.0f);
                        f3 D;
                        for (;;) {
                            const float jx = rnd(lp.seed);
                            const float jy = rnd(lp.seed);
                            D = camera_dir<FM>((float)(lp.pxy & 0xFFFFu), (float)(lp.pxy >> 16), jx, jy, fw, fh, camU, camV, camW);
                            if (reaches_scene(D, elo, ehi)) break;
                            my_culled++;
                            lp.samples_left--;
                            if (lp.samples_left == 0u) { lp.alive = false; finished = true; break; }
                        }
                        if (lp.alive) { rd = m_normalize<FM>(D); ro = eye; lp.new_path = false; }
                    }
                    if (vote(my_culled != 0u) != 0ull) {
                        unsigned long long sum = 0ull;
                        for (uint32_t b = 0; vote((my_culled >> b) != 0u) != 0ull; b++) sum += (unsigned long long)popc(vote(((my_culled >> b) & 1u) != 0u)) << b;
                        n_radiance += sum; n_paths += sum; n_culled += sum;
                    }
                    if (vote(have && !lp.alive) == 0ull) break;                                   // every slot of the round has a ray
                    if (vote(finished) == 0ull && q.res_count == 0u && q.shards_left == 0u) break;  // no items left
                }
                {   // a run that ran out of samples in the last pass's cull still has to be folded
                    const int n_fin = popc(vote(finished));
                    n_pixels += (unsigned long long)n_fin; live_delta -= n_fin;
                    wf_finish_runs(Ad, q, below, lp, finished, book, scratch);
                }
                const bool armed = have && lp.alive;
                n_radiance += (unsigned long long)popc(vote(armed));
                if (armed) {
                    uint32_t* const at = pool + slot;
                    at[F_PXY * POOL] = lp.pxy; at[F_LSEED * POOL] = lp.seed; at[F_RUN * POOL] = lp.samples_left | (lp.fold << 16); at[F_TAGF * POOL] = lp.sub;
                    at[F_RESX * POOL] = __float_as_uint(lp.result.x); at[F_RESY * POOL] = __float_as_uint(lp.result.y); at[F_RESZ * POOL] = __float_as_uint(lp.result.z);
                    at[F_PSEED * POOL] = lp.seed;
                    at[F_ATTX * POOL] = at[F_ATTY * POOL] = at[F_ATTZ * POOL] = __float_as_uint(1.0f);
                    at[F_ROX * POOL] = __float_as_uint(ro.x); at[F_ROY * POOL] = __float_as_uint(ro.y); at[F_ROZ * POOL] = __float_as_uint(ro.z);
                    at[F_RDX * POOL] = __float_as_uint(rd.x); at[F_RDY * POOL] = __float_as_uint(rd.y); at[F_RDZ * POOL] = __float_as_uint(rd.z);
                    at[F_TMAX * POOL] = __float_as_uint(1e16f); at[F_HIT * POOL] = 0xFFFFFFFFu;
                    at[F_NDX * POOL] = at[F_NDY * POOL] = at[F_NDZ * POOL] = at[F_WEIGHT * POOL] = 0u;
                }
                const bool items_left = q.res_count != 0u || q.shards_left != 0u;
                // the dealer's state back into LDS; the partial sums of this round are complete before the lock opens
                ctl->grp_pxy[lane] = q.grp_pxy; ctl->grp_seed[lane] = q.grp_seed;
                if (lane == 0) {
                    ctl->shard = q.shard; ctl->shards_left = q.shards_left; ctl->res_first = q.res_first; ctl->res_count = q.res_count;
                    ctl->grant_g0 = q.grant_g0; ctl->free_top = q.free_top;
                    const int live = ctl->n_live + live_delta;
                    ctl->n_live = live;
                    if (!items_left) {
                        __hip_atomic_store(&ctl->drained, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (live == 0) __hip_atomic_store(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_s_waitcnt(0x0F70);
                ring_push<POOL>(&ctl->q[QT], cells + QT * POOL, armed, slot, lane, below);
                ring_push<POOL>(&ctl->q[QF], cells + QF * POOL, have && !armed, slot, lane, below);
                if (lane == 0) __hip_atomic_store(&ctl->deal_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (DIAG) { d_deal += __builtin_amdgcn_s_memrealtime() - t_round; c_deal += 1; l_deal += (unsigned long long)popc(vote(have)); }
                continue;
            }

            // ---------------------------------- SHADE round: 64 records of one kind ------------------------------------------
            const int qid = choice == 1 ? QH : QM;
            const uint32_t slot = ring_pop<POOL>(&ctl->q[qid], cells + qid * POOL, true, lane, below, trouble);
            const bool have = slot != kWfNone;
            if (vote(have) == 0ull) continue;
            n_rounds += 1; n_lane_rounds += (unsigned long long)popc(vote(have));
            uint32_t r[F_COUNT];
            {
                const uint32_t* at = pool + (have ? slot : 0u);
#pragma unroll
                for (int f = 0; f < (int)F_COUNT; f++) r[f] = at[f * POOL];
            }
            uint32_t pxy = r[F_PXY], lseed = r[F_LSEED], samples_left = r[F_RUN] & 0xFFFFu, tagf = r[F_TAGF], pseed = r[F_PSEED];
            const uint32_t fold = r[F_RUN] >> 16;
            f3 result = mk(__uint_as_float(r[F_RESX]), __uint_as_float(r[F_RESY]), __uint_as_float(r[F_RESZ]));
            f3 att = mk(__uint_as_float(r[F_ATTX]), __uint_as_float(r[F_ATTY]), __uint_as_float(r[F_ATTZ]));
            f3 ro = mk(__uint_as_float(r[F_ROX]), __uint_as_float(r[F_ROY]), __uint_as_float(r[F_ROZ]));
            f3 rd = mk(__uint_as_float(r[F_RDX]), __uint_as_float(r[F_RDY]), __uint_as_float(r[F_RDZ]));
            float tmax = __uint_as_float(r[F_TMAX]);
            const int hit = (int)r[F_HIT];
            f3 nd = mk(__uint_as_float(r[F_NDX]), __uint_as_float(r[F_NDY]), __uint_as_float(r[F_NDZ]));
            float weight = __uint_as_float(r[F_WEIGHT]);
            int depth = (int)((tagf >> kTagDepthShift) & 31u);
            const bool was_shadow = (tagf & kTagShadow) != 0u;
            tagf &= 0xFFFFu;                                   // sub stays; depth and flags are rebuilt below

            bool segment_done = false, started_shadow = false;
            Pending pd; pd.nxt_org = mk(0.0f); pd.nxt_dir = mk(0.0f, 0.0f, 1.0f); pd.radiance = mk(0.0f); pd.weight = 0.0f; pd.done = true;
            if (have) {
                if (was_shadow) {                                         // shadow ray back (:1015-1024)
                    pd.done = (r[F_TAGF] & kTagDone) != 0u;
                    pd.weight = weight;
                    pd.radiance = pd.done ? nd : mk(0.0f);
                    pd.nxt_dir = nd;
                    pd.nxt_org = (r[F_TAGF] & kTagMetal) ? ro + nd * 1e-4f : ro;      // ro is the shadow ray's origin P (:926, :948)
                    if (hit < 0) pd.radiance = m_madd<FM>(mk(late().light.emission), pd.weight, pd.radiance);
                    segment_done = true;
                } else {                                                  // radiance ray back
                    bool want_shadow = false;
                    f3 P, L; float Ldist;
                    f3 emission = mk(0.0f);
                    if (hit >= 0) {
                        want_shadow = shade_hit<FM, false>(late().scene, late, ro, rd, tmax, hit, depth, pseed, att, emission, pd, P, L, Ldist);
                    } else {                                              // __miss__ms :833-847
                        pd.radiance = mk(0.0f); pd.weight = 0.0f; pd.done = true;
                    }
                    result += emission;                                   // :760 (before the radiance term)
                    if (want_shadow) {
                        // is the next bounce's origin P itself, or the conductor's offset one?  (refraction takes no light sample)
                        const bool metal = !(pd.nxt_org.x == P.x && pd.nxt_org.y == P.y && pd.nxt_org.z == P.z) && !pd.done;
                        ro = P; rd = L; tmax = Ldist - 0.01f;
                        nd = pd.done ? pd.radiance : pd.nxt_dir;
                        weight = pd.weight;
                        tagf |= kTagShadow | (pd.done ? kTagDone : 0u) | (metal ? kTagMetal : 0u);
                        started_shadow = true;
                    } else {
                        segment_done = true;
                    }
                }
            }
            n_shadow += (unsigned long long)popc(vote(started_shadow));
            bool end = false, finished = false, new_path = false, start_radiance = false;
            if (segment_done) {                                           // raygen :761-778
                add_segment<FM>(result, pd.radiance, att);
                const float p = roulette_p<FM>(att);
                const bool rr = rnd(pseed) > p;
                end = pd.done || rr || (uint32_t)depth >= late().maxDepth;
                if (!end) {
                    att = roulette_scale<FM>(att, p);
                    ro = pd.nxt_org; rd = pd.nxt_dir;
                    ++depth;
                    start_radiance = true;
                } else {
                    samples_left--;
                    new_path = true;
                    if (samples_left == 0u) finished = true;
                }
            }
            n_paths += (unsigned long long)popc(vote(end));
            uint32_t my_culled = 0u;
            if (new_path && !finished) {                                  // camera path start, :727-745
                const RenderArgs& Rc = late();
                const f3 eye = mk(Rc.eye), camU = mk(Rc.U), camV = mk(Rc.V), camW = mk(Rc.W);
                const float fw = (float)(int)Rc.width, fh = (float)(int)Rc.height;
                const f3 elo = Rc.scene.n_tris ? mk(Rc.cull_lo) - eye : mk(1.0f), ehi = Rc.scene.n_tris ? mk(Rc.cull_hi) - eye : mk(-1.0f);
                f3 D;
                for (;;) {
                    const float jx = rnd(lseed);
                    const float jy = rnd(lseed);
                    D = camera_dir<FM>((float)(pxy & 0xFFFFu), (float)(pxy >> 16), jx, jy, fw, fh, camU, camV, camW);
                    if (reaches_scene(D, elo, ehi)) break;
                    my_culled++;
                    samples_left--;
                    if (samples_left == 0u) { finished = true; break; }
                }
                if (!finished) {
                    rd = m_normalize<FM>(D);
                    ro = eye;
                    att = mk(1.0f);
                    pseed = lseed;
                    depth = 0;
                    start_radiance = true;
                }
            }
            if (vote(my_culled != 0u) != 0ull) {
                unsigned long long sum = 0ull;
                for (uint32_t b = 0; vote((my_culled >> b) != 0u) != 0ull; b++) sum += (unsigned long long)popc(vote(((my_culled >> b) & 1u) != 0u)) << b;
                n_radiance += sum; n_paths += sum; n_culled += sum;
            }
            n_radiance += (unsigned long long)popc(vote(start_radiance));
            if (start_radiance) { tmax = 1e16f; nd = mk(0.0f); weight = 0.0f; }
            if (have) {
                uint32_t* const at = pool + slot;
                at[F_LSEED * POOL] = lseed; at[F_RUN * POOL] = samples_left | (fold << 16);
                at[F_TAGF * POOL] = tagf | ((uint32_t)depth << kTagDepthShift);
                at[F_RESX * POOL] = __float_as_uint(result.x); at[F_RESY * POOL] = __float_as_uint(result.y); at[F_RESZ * POOL] = __float_as_uint(result.z);
                at[F_PSEED * POOL] = pseed;
                at[F_ATTX * POOL] = __float_as_uint(att.x); at[F_ATTY * POOL] = __float_as_uint(att.y); at[F_ATTZ * POOL] = __float_as_uint(att.z);
                at[F_ROX * POOL] = __float_as_uint(ro.x); at[F_ROY * POOL] = __float_as_uint(ro.y); at[F_ROZ * POOL] = __float_as_uint(ro.z);
                at[F_RDX * POOL] = __float_as_uint(rd.x); at[F_RDY * POOL] = __float_as_uint(rd.y); at[F_RDZ * POOL] = __float_as_uint(rd.z);
                at[F_TMAX * POOL] = __float_as_uint(tmax);
                at[F_NDX * POOL] = __float_as_uint(nd.x); at[F_NDY * POOL] = __float_as_uint(nd.y); at[F_NDZ * POOL] = __float_as_uint(nd.z);
                at[F_WEIGHT * POOL] = __float_as_uint(weight);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            ring_push<POOL>(&ctl->q[QT], cells + QT * POOL, have && !finished, slot, lane, below);
            ring_push<POOL>(&ctl->q[QR], cells + QR * POOL, have && finished, slot, lane, below);
            if (DIAG && choice == 1) { d_hit += __builtin_amdgcn_s_memrealtime() - t_round; c_hit += 1; l_hit += (unsigned long long)popc(vote(have)); }
            else if (DIAG) { d_acct += __builtin_amdgcn_s_memrealtime() - t_round; c_acct += 1; l_acct += (unsigned long long)popc(vote(have)); }
        }
    }
    if (DIAG) d_total = __builtin_amdgcn_s_memrealtime() - t_begin;

    if (lane == 0) {
        atomicAdd(&late().counters[0], n_radiance);
        atomicAdd(&late().counters[1], n_shadow);
        atomicAdd(&late().counters[2], n_paths);
        atomicAdd(&late().counters[3], n_pixels);
        atomicAdd(&late().counters[4], n_steps);
        atomicAdd(&late().counters[5], n_lane_steps);
        atomicAdd(&late().counters[6], n_rounds);
        atomicAdd(&late().counters[7], n_lane_rounds);
        if (n_culled) atomicAdd(&late().counters[kCulledCounter], n_culled);
        unsigned long long* dg = late().counters + kWfDiag;
        if (!DIAG) {}
        else if (wave >= (uint32_t)NS) { atomicAdd(&dg[0], d_total); atomicAdd(&dg[1], d_idle); atomicAdd(&dg[13], d_deal); atomicAdd(&dg[14], c_deal); atomicAdd(&dg[15], l_deal); atomicAdd(&dg[16], n_steps); }
        else {
            atomicAdd(&dg[2], d_total); atomicAdd(&dg[3], d_idle);
            atomicAdd(&dg[4], d_deal); atomicAdd(&dg[5], c_deal); atomicAdd(&dg[6], l_deal);
            atomicAdd(&dg[7], d_hit); atomicAdd(&dg[8], c_hit); atomicAdd(&dg[9], l_hit);
            atomicAdd(&dg[10], d_acct); atomicAdd(&dg[11], c_acct); atomicAdd(&dg[12], l_acct);
        }
        const uint32_t ab = lds_load(&ctl->abort_flag);
        if (ab != 0u) atomicAdd(&late().counters[kAbortCounter], 1ull);         // pt_launch fails: the image is not complete
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
#define WF(...) k_render_wf<__VA_ARGS__>
#define WFN(...) "k_render_wf<" #__VA_ARGS__ ">"
static const WfDesc kWfVariants[] = {
    {WF(12, 4, 512, 16, 16, 16, 5, 1, false, 0), 12, 4, 512, 16, "wavefront: 12 trace + 4 shade waves per workgroup, 512 record slots in LDS, fp16 nodes, one workgroup per CU (a measured experiment: never chosen automatically)", WFN(12, 4, 512, 16, 16, 16, 5, 1, false, 0), WF(12, 4, 512, 16, 16, 16, 5, 1, false, 1), WFN(12, 4, 512, 16, 16, 16, 5, 1, false, 1)},
    {WF(12, 4, 512, 16, 16, 16, 5, 1, true, 0), 12, 4, 512, 16, "wavefront 12 + 4 with per-role time stamps (pt_debug_wf)", WFN(12, 4, 512, 16, 16, 16, 5, 1, true, 0), WF(12, 4, 512, 16, 16, 16, 5, 1, true, 1), WFN(12, 4, 512, 16, 16, 16, 5, 1, true, 1)},
#ifdef ACGPT_EXPERIMENTS
    {WF(8, 2, 256, 16, 16, 16, 5, 2, false, 0), 8, 2, 256, 16, "wavefront: 8 trace + 2 shade waves per workgroup, 256 record slots, fp16 nodes, two workgroups per CU (five waves per SIMD)", WFN(8, 2, 256, 16, 16, 16, 5, 2, false, 0), WF(8, 2, 256, 16, 16, 16, 5, 2, false, 1), WFN(8, 2, 256, 16, 16, 16, 5, 2, false, 1)},
    {WF(8, 8, 512, 16, 16, 16, 5, 1, true, 0), 8, 8, 512, 16, "wavefront 8 + 8 with per-role time stamps", WFN(8, 8, 512, 16, 16, 16, 5, 1, true, 0), WF(8, 8, 512, 16, 16, 16, 5, 1, true, 1), WFN(8, 8, 512, 16, 16, 16, 5, 1, true, 1)},
    {WF(10, 6, 512, 16, 16, 16, 5, 1, true, 0), 10, 6, 512, 16, "wavefront 10 + 6 with per-role time stamps", WFN(10, 6, 512, 16, 16, 16, 5, 1, true, 0), WF(10, 6, 512, 16, 16, 16, 5, 1, true, 1), WFN(10, 6, 512, 16, 16, 16, 5, 1, true, 1)},
#endif
};
int wf_variant_count() { return (int)(sizeof(kWfVariants) / sizeof(kWfVariants[0])); }
const WfDesc* wf_variant(int i) { return (i >= 0 && i < wf_variant_count()) ? &kWfVariants[i] : nullptr; }
size_t wf_lds_bytes(const WfDesc& d, uint32_t stack_entries)
{
    const uint32_t entries = stack_entries > (uint32_t)d.stack_cap ? (uint32_t)d.stack_cap : stack_entries;
    const uint32_t fold = (uint32_t)((d.nt * 64 + d.pool + 64 + 3) & ~3);
    size_t dw = (sizeof(WfCtl) + 3u) / 4u + 64u + fold / 4u + fold / 2u + (size_t)kRings * d.pool + (size_t)F_COUNT * d.pool + (size_t)d.nt * entries * 64u;
    return dw * 4u;
}

}  // namespace ptd
