// render_common.h — device helpers shared by the render kernels (render_megakernel.hip, render_wavefront.hip): the pixel
// queue and its item decode, sample-run folding, frame sums, the camera ray and its scene-box cull, and how the launch
// constants are read.  Included by kernel translation units only.
#pragma once
#include "pt_device.h"
#include "render_megakernel.h"
#include "pt_shading.h"

namespace ptd {

extern __shared__ uint32_t lds_dyn[];

// ---- pixel queue ---------------------------------------------------------------------------------
// Work item = (pixel, sub-frame of the batch, run of samples); index = pixel slot << sub_shift | sub-frame << chunk_shift
// | run.  The pixel slot is the position in the 8x4-tile order of sutil/WorkDistribution.h for (rank, world); everything
// a lane needs (pixel coordinates, seed = tea<4>(pixel, frame) skipped ahead to the run's first sample) follows from
// the index, so taking an item touches no memory.
// Grants: the first idle lane (ffs of the ballot) takes max(idle lanes, A.grant) consecutive items — rounded up to
// whole (pixel, sub-frame) groups — of the wave's queue shard with ONE atomicAdd; the wave hands them out by
// popcount-prefix and keeps the rest as a reserve, so most refills touch no atomic at all.  Shards are per XCD;
// drained shards are stolen from round-robin.
//
// Fold slots: all runs of one (pixel, sub-frame) are granted to ONE wave (grants are whole groups).  A group takes a slot
// of the wave's scratch when its first run is dealt to a lane (popped from the wave's free list in LDS and left in the
// book for the lanes that get the group's other runs; not when the grant is decoded: a grant of 32 groups would hold 32
// slots for groups that are not being worked on yet, and the slots in use are what has to stay in the L2); a lane that
// finishes a run parks its partial sum there and bumps the slot's ticket (an LDS counter, one ds_add_rtn for all the lanes
// that finish in a round); the lane that brings the ticket to the run count adds the partial sums in run order — the
// association orc_render(chunks) uses —, writes the sum and pushes the slot back.  Open groups per wave <= 64 (one per
// lane in flight), slots 128.
constexpr uint32_t kFoldSlots = 128u;
constexpr uint32_t kNoSlot = 0xFFu;
constexpr uint32_t kBookDwords = kFoldSlots / 2u + 16u;       // per wave in LDS: tickets and free-slot stack (a byte each), slot of each group of the grant (a byte each)

struct WaveBook {
    uint32_t* tick;      // [kFoldSlots / 4] runs parked so far, one byte per slot (<= 32 runs)
    uint8_t*  free;      // [kFoldSlots] stack of free slots
    uint8_t*  gslot;     // [64] fold slot of group grant_g0 + j of the current grant, once its first run has been dealt
    // park one more run in `slot`; returns how many were parked before (ds_add_rtn_u32 on the byte's dword)
    __device__ __forceinline__ uint32_t bump(uint32_t slot) const
    { return (atomicAdd(&tick[slot >> 2], 1u << (8u * (slot & 3u))) >> (8u * (slot & 3u))) & 0xFFu; }
    __device__ __forceinline__ void clear(uint32_t slot) const { atomicAnd(&tick[slot >> 2], ~(0xFFu << (8u * (slot & 3u)))); }
};
__device__ __forceinline__ WaveBook wave_book(uint32_t* lds, uint32_t lane)
{
    WaveBook b; b.tick = lds; b.free = (uint8_t*)(lds + kFoldSlots / 4u); b.gslot = (uint8_t*)(lds + kFoldSlots / 2u);
    if (lane < kFoldSlots / 4u) b.tick[lane] = 0u;
    b.free[lane] = (uint8_t)lane; b.free[64u + lane] = (uint8_t)(64u + lane);
    return b;
}

struct QueueState {
    uint32_t shard, shards_left, res_first, res_count;
    uint32_t grant_g0;                    // first group of the current grant
    uint32_t free_top;                    // entries on the wave's stack of free fold slots (WaveBook::free)
    uint32_t grp_pxy, grp_seed;           // PER LANE: lane j holds pixel (x | y << 16, bit 31: every ray of the pixel reaches the scene box; 0xFFFFFFFF = padding) and tea<4> seed of group grant_g0 + j
    uint32_t skipped;                     // groups of pixels that cannot reach the scene box, settled when their grant was decoded (the kernel books their samples)
};

struct LanePixel {
    bool alive, new_path;
    uint32_t pxy;          // px | py << 16
    uint32_t seed, samples_left;
    uint32_t tag;          // sub-frame << chunk_shift | run, fold slot << 16
    f3 result;
};

// n / d for a launch constant d by multiply-high and shifts (Granlund & Montgomery 1994, N = 32: exact for every
// 32-bit n); capi.hip builds {mul, sh1, sh2} and checks them.  Integer division has no scalar instruction and costs
// ~40 vector ones; this is 4, and on wave-uniform operands they are scalar.
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& d)
{
    const uint32_t t = __umulhi(n, d.mul);
    return (t + ((n - t) >> d.sh1)) >> d.sh2;
}
// StaticWorkDistribution::getSamplePixel (sutil/WorkDistribution.h:60-81) with the two divisions by launch constants
// (tile-strip columns, GPU count) as fast_div; same results as sample_pixel()
__device__ __forceinline__ void sample_pixel_fast(const RenderArgs& A, uint32_t sample_idx, uint32_t& px, uint32_t& py)
{
    const uint32_t world = (uint32_t)A.world;
    const uint32_t tile_strip_idx = sample_idx >> 5;                       // 8 x 4 pixels per tile
    const uint32_t tile_strip_y = fast_div(tile_strip_idx, A.div_cols);
    const uint32_t tile_strip_x = tile_strip_idx - tile_strip_y * A.strip_cols;
    const uint32_t tile_pixel_idx = sample_idx & 31u;
    const uint32_t a = (uint32_t)A.rank + (tile_strip_y - fast_div(tile_strip_y, A.div_world) * world);   // gpu_idx + tile_strip_y % num_gpus
    const uint32_t tile_offset_x = (a - fast_div(a, A.div_world) * world) * 8u;
    py = tile_strip_y * 4u + (tile_pixel_idx >> 3);
    px = tile_strip_x * (8u * world) + (tile_pixel_idx & 7u) + tile_offset_x;
}

// the sum of one (pixel, sub-frame) is complete: it is parked per (pixel, sub-frame); k_finalize blends the sub-frames of
// the launch into the accumulation buffer in frame order and applies make_color (the megakernel carries neither: their
// powf code would be inlined at every place a lane can finish)
// The sums are indexed by the pixel's slot in this rank's tile order (the inverse of sample_pixel_fast), so a rank that
// holds 1/world of the tiles holds 1/world of the sums.
__device__ __forceinline__ uint32_t pixel_slot(const RenderArgs& A, uint32_t pxy)
{
    const uint32_t px = pxy & 0xFFFFu, py = pxy >> 16;
    const uint32_t strip_x = fast_div(px >> 3, A.div_world);              // px / (8 * world)
    return (((py >> 2) * A.strip_cols + strip_x) << 5) | ((py & 3u) << 3) | (px & 7u);
}
__device__ __forceinline__ void write_frame_sum(const RenderArgs& A, uint32_t pxy, uint32_t f, const f3& sum)
{
    A.frame_sums[(size_t)pixel_slot(A, pxy) * A.n_frames + f] = make_float4(sum.x, sum.y, sum.z, 0.0f);
}

// Queue position of a pixel slot -> the pixel slot it stands for.  The queue is cut into eight contiguous shards, one per XCD;
// with the identity (A.row_interleave 0) a shard is a contiguous band of the image, and bands differ in cost: XCDs drift apart
// by up to a fifth of the launch and even out only by stealing at the end (profiles/r02_wave_timeline.txt).  Otherwise the
// image is dealt in units — 1: a tile-strip row (4 pixel rows), 2: one 8x4 tile — and the units go to the shards round robin
// (unit u to shard u mod 8, what sutil/WorkDistribution.h:60-81 does across GPUs): queue order = units 0, 8, 16, ..., 1, 9, ...
// (All units are the same size, so this is a permutation of the pixel slots.)  3: the identity, with every wave starting at
// shard 0 — one queue in image order.
__device__ __forceinline__ uint32_t queue_slot(const RenderArgs& A, uint32_t pos)
{
    const uint32_t mode = A.row_interleave;
    if (mode != 1u && mode != 2u) return pos;
    const uint32_t tile = pos >> 5;
    uint32_t k, within, unit_tiles, n_units;           // unit position in queue order, tile within the unit
    if (mode == 2u) { k = tile; within = 0u; unit_tiles = 1u; n_units = A.strip_rows * A.strip_cols; }
    else { k = fast_div(tile, A.div_cols); unit_tiles = A.strip_cols; within = tile - k * unit_tiles; n_units = A.strip_rows; }
    uint32_t unit = k;
    for (uint32_t s = 0; s < 8u; s++) {
        const uint32_t n_s = n_units > s ? (n_units - s + 7u) >> 3 : 0u;     // units congruent to s modulo 8
        if (k < n_s) { unit = k * 8u + s; break; }
        k -= n_s;
    }
    return ((unit * unit_tiles + within) << 5) | (pos & 31u);
}

// lcg_skip: {multiplier, increment} of the LCG skip-ahead per run, staged in LDS by the kernel (a per-lane table look-up
// in the kernel-argument segment would be a global load on the deal's critical path)
// late(): the launch constants again, for the grant decode — once per 256 items, so they are read there (kernel-argument segment,
// scalar cache) instead of being held in scalar registers across the whole persistent kernel (see RenderArgsBox below)
template <bool STATS = false, typename Late>
__device__ __forceinline__ void refill_lanes(const RenderArgs& A, Late late, QueueState& q, uint32_t lane, unsigned long long below, LanePixel& lp,
                                             const uint32_t* lcg_skip, const WaveBook& book)
{
    const uint32_t cs = A.chunk_shift, run_mask = (1u << cs) - 1u;
    const uint32_t fshift = A.sub_shift - cs, fmask = (1u << fshift) - 1u;      // group index = pixel slot << fshift | sub-frame
    unsigned long long idle = vote(!lp.alive);
    while (idle != 0ull && (q.res_count != 0u || q.shards_left != 0u)) {
        if (q.res_count == 0u) {                                      // wave-uniform: fetch a grant
            const RenderArgs& G = late();
            const uint32_t leader = (uint32_t)__ffsll((long long)idle) - 1u;
            const uint32_t idle_n = (uint32_t)popc(idle);
            uint32_t req = idle_n > G.grant ? idle_n : G.grant;       // at least what is needed now,
            req = (req + run_mask) & ~run_mask;                       // in whole groups (shards begin on group boundaries)
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&G.queue_heads[q.shard], req);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
            if (STATS && lane == leader && base < G.shard_size) {       // progress of the shard: first grant past each 1/256
                const uint32_t slot = (uint32_t)(((unsigned long long)base << 8) / G.shard_size);
                unsigned long long* pr = G.counters + 8 + 3 * kMaxTimedWaves + 256u * q.shard + slot;
                if (*pr == 0ull) *pr = __builtin_amdgcn_s_memrealtime();
            }
            const uint32_t shard_begin = q.shard * G.shard_size;
            uint32_t shard_end = shard_begin + G.shard_size;
            if (shard_end > G.total_samples) shard_end = G.total_samples;
            if (shard_begin > G.total_samples) shard_end = shard_begin;
            const uint32_t first = shard_begin + base;
            uint32_t avail = first < shard_end ? shard_end - first : 0u;
            if (avail > req) avail = req;
            if (avail < req) { q.shard = (q.shard + 1u) & 7u; q.shards_left--; }   // shard drained: steal from the next
            q.res_first = first; q.res_count = avail;
            if (avail == 0u) continue;
            {
                // decode the grant's groups side by side, one per lane (<= 64 of them): tile order -> pixel, tea<4> seed (:721).
                // Once per grant instead of one serial tea<4> chain per group on the deal's critical path.
                q.grant_g0 = first >> cs;
                const uint32_t g = q.grant_g0 + lane;
                uint32_t x, y;
                sample_pixel_fast(G, queue_slot(G, g >> fshift), x, y);
                const uint32_t f = g & fmask;
                const bool ok = (g << cs) < first + avail && x < G.width && y < G.height && f < G.n_frames;   // else: padding of the tile / batch grid
                q.grp_seed = tea4(y * G.width + x, G.frame + f);
                uint32_t pxy = ok ? (x | (y << 16)) : 0xFFFFFFFFu;
                if (G.row_spans != nullptr) {
                    // the pixel's class (capi.hip row_spans).  Outside the row's outer span no ray through the pixel reaches the scene's
                    // bounding box: all its samples are the reference's __miss__ms case (:833-847), their sum is zero — written here, the
                    // group never becomes work items.  Inside the inner span every ray reaches the box (bit 31): path starts skip the cull test.
                    const uint2 sp = G.row_spans[ok ? (pxy >> 16) : 0u];
                    const uint32_t px = pxy & 0xFFFFu;
                    const bool outside = ok && (px < (sp.x & 0xFFFFu) || px >= (sp.x >> 16));
                    if (outside) write_frame_sum(G, pxy, f, mk(0.0f));
                    q.skipped += (uint32_t)popc(vote(outside));
                    if (ok && px >= (sp.y & 0xFFFFu) && px < (sp.y >> 16)) pxy |= 0x80000000u;
                    if (outside) pxy = 0xFFFFFFFFu;
                }
                q.grp_pxy = pxy;
                // a grant whose groups are all settled (pixels that cannot reach the scene box) or padding holds nothing to deal: drop
                // it whole instead of handing out its items, 64 at a time, to lanes that find nothing in them
                if (vote(pxy != 0xFFFFFFFFu) == 0ull) { q.res_count = 0u; continue; }
            }
        }
        const uint32_t want = (uint32_t)popc(idle);
        const uint32_t take = want < q.res_count ? want : q.res_count;
        const uint32_t rank = (uint32_t)popc(idle & below);
        const uint32_t item = q.res_first + rank;
        {
            // a lane fetches pixel and seed of its group from the lane that decoded it when the grant was taken, and skips the
            // LCG ahead to its run (one run per group: a skip of zero steps, no slot)
            const uint32_t run = item & run_mask;
            const uint32_t gj = (item >> cs) - q.grant_g0;
            const int src = (int)(gj << 2);                                    // ds_bpermute takes a byte index
            const uint32_t pxy_f = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)q.grp_pxy);
            const uint32_t seed0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)q.grp_seed);
            const bool mine = !lp.alive && rank < take && pxy_f != 0xFFFFFFFFu;
            const uint32_t pxy = pxy_f & 0x7FFFFFFFu;
            uint32_t sl = kNoSlot;
            if (cs != 0u) {
                // fold slot of the group: the lane that gets run 0 pops one (the n-th such lane of this deal the n-th entry from
                // the top of the free stack) and leaves it in the book; runs are dealt in item order, so the lanes with the
                // group's other runs — in this deal or a later one — find it there (LDS operations of a wave are in order)
                const unsigned long long opens = vote(mine && run == 0u);
                if (opens != 0ull) {
                    if (mine && run == 0u) book.gslot[gj] = book.free[q.free_top - 1u - (uint32_t)popc(opens & below)];
                    q.free_top -= (uint32_t)popc(opens);
                }
                if (mine) sl = (uint32_t)book.gslot[gj];
            }
            if (mine) {
                lp.pxy = pxy;
                lp.tag = (((item >> cs) & fmask) << cs) | run | (sl << 16) | ((pxy_f >> 31) << 24);      // bit 24: no cull test needed
                lp.seed = lcg_skip[2u * run] * seed0 + lcg_skip[2u * run + 1u];     // skip the jitter draws of the samples before this run (2 per sample)
                lp.result = mk(0.0f);
                lp.samples_left = A.chunk_spp;
                lp.alive = true;
                lp.new_path = true;
            }
        }
        q.res_first += take; q.res_count -= take;
        idle = vote(!lp.alive);          // lanes that drew a padding item try again
    }
}

// mean over spp, progressive lerp (:782-811)
__device__ __forceinline__ f3 blend_frame(const f3& prev, const f3& result, uint32_t spp, uint32_t frame)
{
    f3 accum = result / (float)spp;
    if (frame > 0u) {
        const float a = 1.0f / (float)(frame + 1u);
        accum = lerp3(prev, accum, a);
    }
    return accum;
}

// Two / four consecutive 16-byte loads served by the L2 (sc0: past this CU's L1, whatever an earlier use of the addresses
// left there; the partial sums were written by this very CU, so the L2 of its XCD holds them and nothing has to leave it), all in flight together and waited for inside the same asm block (the compiler never sees a
// register whose load has not landed).
typedef float v3f_t __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void load2_coherent(const float* p, v3f_t& a, v3f_t& b)
{
    asm volatile("global_load_dwordx3 %0, %2, off sc0\n\tglobal_load_dwordx3 %1, %2, off offset:12 sc0\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(p) : "memory");
}
__device__ __forceinline__ void load4_coherent(const float* p, v3f_t& a, v3f_t& b, v3f_t& c, v3f_t& d)
{
    asm volatile("global_load_dwordx3 %0, %4, off sc0\n\tglobal_load_dwordx3 %1, %4, off offset:12 sc0\n\t"
                 "global_load_dwordx3 %2, %4, off offset:24 sc0\n\tglobal_load_dwordx3 %3, %4, off offset:36 sc0\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(p) : "memory");
}

// Lanes with `finished` set have completed their run of samples.  One run per pixel: write.  Several: park the partial
// sum in the group's fold slot, bump the slot's ticket, and the lane that completes the group adds the runs in order.
// scratch: this wave's (kFoldSlots << chunk_shift) partial sums, three floats each.
__device__ __forceinline__ void finish_runs(const RenderArgs& A, QueueState& q, unsigned long long below, const LanePixel& lp, bool finished, const WaveBook& book, float* __restrict__ scratch)
{
    if (vote(finished) == 0ull) return;
    const uint32_t cs = A.chunk_shift, runs = 1u << cs;
    const uint32_t sub = lp.tag & 0xFFFFu;
    if (cs == 0u) {
        if (finished) write_frame_sum(A, lp.pxy, sub, lp.result);
        return;
    }
    const uint32_t slot = (lp.tag >> 16) & 0xFFu;
    float* group = scratch + 3u * ((size_t)slot << cs);
    bool folder = false;
    if (finished) {
        float* mine = group + 3u * (sub & (runs - 1u));
        mine[0] = lp.result.x; mine[1] = lp.result.y; mine[2] = lp.result.z;
        folder = book.bump(slot) == runs - 1u;                       // LDS: lanes of one group that finish together get distinct counts
    }
    const unsigned long long folders = vote(folder);
    if (folders == 0ull) return;
    // the partial sums were stored by lanes of this wave through this CU's L1: wait for the stores, then read them back
    // past the L1 (sc0 loads), whatever lines an earlier use of the slot left there
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0) only
    if (folder) {
        // the loads of four runs in flight at once (one latency per four runs, not one per run)
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
                if (k0 == 0u) sum = mk(p0.x, p0.y, p0.z); else sum += mk(p0.x, p0.y, p0.z);      // the chain starts at run 0, not at zero
                sum += mk(p1.x, p1.y, p1.z);
                sum += mk(p2.x, p2.y, p2.z);
                sum += mk(p3.x, p3.y, p3.z);
            }
        }
        write_frame_sum(A, lp.pxy, sub >> cs, sum);
        book.clear(slot);
        book.free[q.free_top + (uint32_t)popc(folders & below)] = (uint8_t)slot;          // back on the free stack
    }
    q.free_top += (uint32_t)popc(folders);
}

// Camera ray through pixel (px, py) with jitter (jx, jy), unnormalised (:730-737)
template <int FM = 0>
__device__ __forceinline__ f3 camera_dir(float px, float py, float jx, float jy, float fw, float fh, const f3& U, const f3& V, const f3& W)
{
    if (FM >= 2) {
        const float dx = __builtin_fmaf(2.0f, m_div<FM>(px + jx, fw), -1.0f), dy = __builtin_fmaf(2.0f, m_div<FM>(py + jy, fh), -1.0f);
        return m_madd<FM>(V, dy, m_madd<FM>(U, dx, W));
    }
    const float dx = 2.0f * m_div<FM>(px + jx, fw) - 1.0f;
    const float dy = 2.0f * m_div<FM>(py + jy, fh) - 1.0f;
    return dx * U + dy * V + W;
}
// Can a ray from the eye along D (any length) reach the scene's bounding box?  elo / ehi: box corners minus the eye, the box
// itself enlarged on the host beyond every rounding below.  false = certain miss: no triangle can be hit, the path is the
// reference's __miss__ms case (:833-847) without a traversal.
__device__ __forceinline__ bool reaches_scene(const f3& D, const f3& elo, const f3& ehi)
{
    const float rx = finite_rcp(D.x), ry = finite_rcp(D.y), rz = finite_rcp(D.z);
    const float x0 = elo.x * rx, x1 = ehi.x * rx, y0 = elo.y * ry, y1 = ehi.y * ry, z0 = elo.z * rz, z1 = ehi.z * rz;
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * kFarWiden;
    return tn <= tf;
}

// The launch constants live in the kernel-argument segment.  Left to itself the compiler loads all of them once, keeps
// them in scalar registers for the whole kernel and, out of registers, spills the BVH loop's own pointers.  The persistent
// kernel therefore re-reads what only the shade phase needs at the start of every shade round, through an index the
// compiler cannot see through (a zero made by an opaque instruction), so those values never live across the BVH loop.
struct RenderArgsBox { RenderArgs a[1]; };
__device__ __forceinline__ uint32_t opaque_zero() { uint32_t z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }


// ---- host-side interface of the kernel translation units --------------------------------------------------------------------
typedef void (*RenderKernel)(const RenderArgsBox);
#ifdef ACGPT_EXPERIMENTS
// workgroup-level wavefront kernels (render_wavefront.hip; measured, lost, experiments build only)
struct WfDesc { RenderKernel k; int nt, ns, pool, stack_cap; const char* name; const char* kernel; RenderKernel k_fast; const char* kernel_fast; };
int wf_variant_count();
const WfDesc* wf_variant(int i);
size_t wf_lds_bytes(const WfDesc& d, uint32_t stack_entries);
#endif

}  // namespace ptd
