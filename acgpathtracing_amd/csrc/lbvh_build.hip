// lbvh_build.hip — on-device LBVH construction for gfx950.
//
// Replaces optixAccelBuild + compaction (PathTracerMain.cpp:329-397); the reference holds
// no BVH code of its own (its comment at :300-302), so the structure is ours:
//   1. k_prepare    per triangle: edges, padded AABB, scene bounds (ordered-uint atomics)
//   2. k_morton     30-bit Morton code of the AABB centre inside the scene bounds
//   3. radix sort   LSD, 4 passes x 8 bits, stable (ties keep triangle order):
//                   k_hist -> k_scan -> k_scatter, ranks by wave64 ballot match
//   4. k_hierarchy  Karras 2012 radix tree over the sorted codes (ties broken by index)
//   5. k_refit      bottom-up: the second thread to reach a node writes its 64-byte
//                   traversal node (both child boxes) and height; agent-scope acq_rel
//                   counters carry the child boxes between CUs
// Everything is deterministic: the sorted order and therefore the tree are unique for a
// given triangle list.
#include "pt_device.h"
#include "lbvh_build.h"
#include <hip/hip_fp16.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

namespace ptd {

__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t u)
{
    uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f; memcpy(&f, &b, 4); return f;
#endif
}

// --- 1. triangle records + bounds -------------------------------------------------
// Padded AABB of a triangle FROM ITS RECORD (v0, e1, e2): the corners are v0, v0 + e1, v0 + e2 — the triangle the
// Moeller-Trumbore test works on — so that the boxes can be recomputed from the records alone, bit for bit, after the fp32
// node array has been released (refit_nodes below).  pad: the triangle test accepts rays a few ulps outside the exact triangle
// (relative term), and the render kernel's slab test t = p * (1/d) - o/d rounds -o/d once, i.e. moves a plane by up to
// |o| * 2^-24; pad_abs = 2^-19 of the largest |coordinate| of the scene covers that for ray origins up to 32 scene sizes away.
__device__ __forceinline__ void record_aabb(const TriRecord& r, float pad_abs, float lo[3], float hi[3])
{
    const float pa[3] = {r.r0.x, r.r0.y, r.r0.z};
    const float pb[3] = {pa[0] + r.r0.w, pa[1] + r.r1.x, pa[2] + r.r1.y};
    const float pc[3] = {pa[0] + r.r1.z, pa[1] + r.r1.w, pa[2] + r.r2.x};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float l = fminf(pa[k], fminf(pb[k], pc[k])), h = fmaxf(pa[k], fmaxf(pb[k], pc[k]));
        const float pad = fmaxf(1e-5f * fmaxf(1.0f, fmaxf(fabsf(l), fabsf(h))), pad_abs);
        lo[k] = l - pad; hi[k] = h + pad;
    }
}

__global__ void k_prepare(const float4* __restrict__ verts, const uint32_t* __restrict__ idx,
                          const uint32_t* __restrict__ mat_ids, uint32_t n_tris,
                          TriRecord* __restrict__ tri_unsorted, float4* __restrict__ tri_lo, float4* __restrict__ tri_hi,
                          uint32_t* __restrict__ scene_bounds /*6 ordered uints: lo xyz, hi xyz*/, float pad_abs)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n_tris) {
        const float4 a = verts[idx[3 * i]], b = verts[idx[3 * i + 1]], c = verts[idx[3 * i + 2]];
        TriRecord r;
        r.r0 = make_float4(a.x, a.y, a.z, b.x - a.x);
        r.r1 = make_float4(b.y - a.y, b.z - a.z, c.x - a.x, c.y - a.y);
        r.r2 = make_float4(c.z - a.z, __uint_as_float(i), __uint_as_float(mat_ids[i]), 0.0f);
        tri_unsorted[i] = r;
        record_aabb(r, pad_abs, lo, hi);
        tri_lo[i] = make_float4(lo[0], lo[1], lo[2], 0.0f);
        tri_hi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    }
    // wave reduction, then one atomic per wave and component
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float l = lo[k], h = hi[k];
        for (int off = 32; off > 0; off >>= 1) {
            l = fminf(l, __shfl_xor(l, off));
            h = fmaxf(h, __shfl_xor(h, off));
        }
        if ((threadIdx.x & 63) == 0) {
            if (l <= h) {
                atomicMin(&scene_bounds[k], f2ord(l));
                atomicMax(&scene_bounds[3 + k], f2ord(h));
            }
        }
    }
}

// --- 2. Morton codes -----------------------------------------------------------------
__device__ __forceinline__ uint32_t expand10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton30(const float l[3], const float h[3], const float slo[3], const float shi[3])
{
    const float c[3] = {0.5f * (l[0] + h[0]), 0.5f * (l[1] + h[1]), 0.5f * (l[2] + h[2])};
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float ext = shi[k] - slo[k];
        float u = ext > 0.0f ? (c[k] - slo[k]) / ext : 0.0f;
        u = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
        q[k] = (uint32_t)u;
    }
    return (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
}
__global__ void k_morton(const float4* __restrict__ tri_lo, const float4* __restrict__ tri_hi, uint32_t n_tris,
                         const uint32_t* __restrict__ scene_bounds, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    const float slo[3] = {ord2f(scene_bounds[0]), ord2f(scene_bounds[1]), ord2f(scene_bounds[2])};
    const float shi[3] = {ord2f(scene_bounds[3]), ord2f(scene_bounds[4]), ord2f(scene_bounds[5])};
    const float4 l4 = tri_lo[i], h4 = tri_hi[i];
    const float l[3] = {l4.x, l4.y, l4.z}, h[3] = {h4.x, h4.y, h4.z};
    keys[i] = morton30(l, h, slo, shi);
    vals[i] = i;
}
// the (code, triangle) pairs of the sorted order, recomputed from the records (pt_read_morton: the build keeps no copy of its keys)
struct Bounds6 { float v[6]; };      // lo xyz, hi xyz
__global__ void k_morton_of_records(const TriRecord* __restrict__ tris, uint32_t n_tris, float pad_abs, const Bounds6 bounds,
                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ prims)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tris) return;
    const TriRecord r = tris[i];
    float l[3], h[3];
    record_aabb(r, pad_abs, l, h);
    const float slo[3] = {bounds.v[0], bounds.v[1], bounds.v[2]}, shi[3] = {bounds.v[3], bounds.v[4], bounds.v[5]};
    keys[i] = morton30(l, h, slo, shi);
    prims[i] = __float_as_uint(r.r2.y);
}

// --- 3. LSD radix sort, 8 bits per pass ---------------------------------------------
constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;
constexpr int kSortTile = kSortThreads * kSortItems;

__global__ void __launch_bounds__(kSortThreads)
k_hist(const uint32_t* __restrict__ keys, uint32_t n, int shift, uint32_t n_blocks, uint32_t* __restrict__ block_hist)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t i = base + r * kSortThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    block_hist[threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `count` values in place, one workgroup of 1024 threads
__global__ void __launch_bounds__(1024) k_scan(uint32_t* __restrict__ data, uint32_t count)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (count + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per, e = min(count, b + per);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; i++) s += data[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = b; i < e; i++) { uint32_t v = data[i]; data[i] = run; run += v; }
}

__global__ void __launch_bounds__(kSortThreads)
k_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t n, int shift,
          uint32_t n_blocks, const uint32_t* __restrict__ block_offs,
          uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t running[256];          // keys of this digit already placed by this block
    __shared__ uint32_t wave_cnt[4][256];      // per-round, per-wave digit counts
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    running[tid] = block_offs[tid * n_blocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; w++) wave_cnt[w][tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortItems; r++) {
        const uint32_t i = base + r * kSortThreads + tid;
        const bool valid = i < n;
        uint32_t key = 0, val = 0, digit = 0;
        if (valid) { key = keys_in[i]; val = vals_in[i]; digit = (key >> shift) & 255u; }
        // lanes of this wave holding the same digit (ballot match over the 8 digit bits)
        unsigned long long same = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bal = __ballot((digit >> b) & 1u);
            same &= ((digit >> b) & 1u) ? bal : ~bal;
        }
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const uint32_t rank = __popcll(same & below);
        if (valid && rank == 0) wave_cnt[wave][digit] = __popcll(same);
        __syncthreads();
        if (valid) {
            uint32_t off = running[digit] + rank;
            for (uint32_t w = 0; w < wave; w++) off += wave_cnt[w][digit];
            keys_out[off] = key;
            vals_out[off] = val;
        }
        __syncthreads();
        {
            uint32_t s = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) { s += wave_cnt[w][tid]; wave_cnt[w][tid] = 0; }
            running[tid] += s;
        }
        __syncthreads();
    }
}

// --- 4. Karras radix tree -----------------------------------------------------------
__device__ __forceinline__ int delta(const uint32_t* __restrict__ keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clz(a ^ b);
}
// children: >= 0 internal node, < 0 leaf (~sorted slot).  parent arrays for the refit.
__global__ void k_hierarchy(const uint32_t* __restrict__ keys, int n, int2* __restrict__ children,
                            int* __restrict__ node_parent, int* __restrict__ leaf_parent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    int left, right;
    if (lo == gamma) { left = ~gamma; leaf_parent[gamma] = i; } else { left = gamma; node_parent[gamma] = i; }
    if (hi == gamma + 1) { right = ~(gamma + 1); leaf_parent[gamma + 1] = i; } else { right = gamma + 1; node_parent[gamma + 1] = i; }
    children[i] = make_int2(left, right);
    if (i == 0) node_parent[0] = -1;
}

// --- quantisation grid: 16 bits per axis over the (slightly enlarged) scene box -------------------
__host__ __device__ inline QGrid make_qgrid_f(const float lo[3], const float hi[3])
{
    QGrid g;
    float c[3], o[3];
    for (int k = 0; k < 3; k++) {
        const float ext = hi[k] - lo[k];
        c[k] = (ext * 1.0001f + 1e-30f) / 65531.0f;     // all boxes land in cells [2, 65533]
        o[k] = lo[k] - 2.0f * c[k];
    }
    g.ox = o[0]; g.oy = o[1]; g.oz = o[2];
    g.cx = c[0]; g.cy = c[1]; g.cz = c[2];
    g.icx = 1.0f / c[0]; g.icy = 1.0f / c[1]; g.icz = 1.0f / c[2];
    return g;
}
// outward rounding plus one cell of slack on each side: the slack absorbs the rounding of the
// ray's own world->grid transform (a few 1e-3 cells), see DESIGN.md
__device__ __forceinline__ uint32_t q_lo(float w, float o, float ic)
{
    const float g = floorf((w - o) * ic) - 1.0f;
    return (uint32_t)fminf(fmaxf(g, 0.0f), 65535.0f);
}
__device__ __forceinline__ uint32_t q_hi(float w, float o, float ic)
{
    const float g = ceilf((w - o) * ic) + 1.0f;
    return (uint32_t)fminf(fmaxf(g, 0.0f), 65535.0f);
}
__device__ __forceinline__ QNode quantise_node(const QGrid g, const float4 l0, const float4 h0, const float4 l1, const float4 h1, int c0, int c1)
{
    QNode q;
    // an empty box (lo = +inf, hi = -inf) quantises to lo = 65535, hi = 0: never hit
    q.a = make_uint4(q_lo(l0.x, g.ox, g.icx) | (q_lo(l0.y, g.oy, g.icy) << 16),
                     q_lo(l0.z, g.oz, g.icz) | (q_hi(h0.x, g.ox, g.icx) << 16),
                     q_hi(h0.y, g.oy, g.icy) | (q_hi(h0.z, g.oz, g.icz) << 16), (uint32_t)c0);
    q.b = make_uint4(q_lo(l1.x, g.ox, g.icx) | (q_lo(l1.y, g.oy, g.icy) << 16),
                     q_lo(l1.z, g.oz, g.icz) | (q_hi(h1.x, g.ox, g.icx) << 16),
                     q_hi(h1.y, g.oy, g.icy) | (q_hi(h1.z, g.oz, g.icz) << 16), (uint32_t)c1);
    return q;
}

// --- 5. sorted leaves + bottom-up refit ------------------------------------------------
__global__ void k_gather_leaves(const uint32_t* __restrict__ vals_sorted, uint32_t n, const TriRecord* __restrict__ tri_unsorted,
                                TriRecord* __restrict__ tris, float4* __restrict__ shade)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const TriRecord r = tri_unsorted[vals_sorted[i]];
    tris[i] = r;
    // what closest-hit shading needs of the triangle: N_0 = normalize(cross(v1 - v0, v2 - v0)) (pathTracerPrograms.cu:890; the
    // same device functions the shade phase would call, so the same bits) and the material id
    const f3 n0 = normalize(cross(mk(r.r0.w, r.r1.x, r.r1.y), mk(r.r1.z, r.r1.w, r.r2.x)));
    shade[i] = make_float4(n0.x, n0.y, n0.z, r.r2.z);
}

__global__ void k_refit(int n, const uint32_t* __restrict__ vals_sorted, const float4* __restrict__ tri_lo, const float4* __restrict__ tri_hi,
                        const int2* __restrict__ children, const int* __restrict__ node_parent, const int* __restrict__ leaf_parent,
                        uint32_t* __restrict__ visit, float4* __restrict__ node_lo, float4* __restrict__ node_hi /* .w = height */,
                        BvhNode* __restrict__ nodes)
{
    const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    int cur = leaf_parent[leaf];
    while (cur >= 0) {
        // release our subtree's boxes, acquire the sibling's: the second arrival proceeds
        const uint32_t prev = __hip_atomic_fetch_add(&visit[cur], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0) return;
        const int2 ch = children[cur];
        float4 l0, h0, l1, h1; float hgt0, hgt1;
        if (ch.x < 0) { const uint32_t p = vals_sorted[~ch.x]; l0 = tri_lo[p]; h0 = tri_hi[p]; hgt0 = 0.0f; }
        else { l0 = node_lo[ch.x]; h0 = node_hi[ch.x]; hgt0 = h0.w; }
        if (ch.y < 0) { const uint32_t p = vals_sorted[~ch.y]; l1 = tri_lo[p]; h1 = tri_hi[p]; hgt1 = 0.0f; }
        else { l1 = node_lo[ch.y]; h1 = node_hi[ch.y]; hgt1 = h1.w; }
        BvhNode nd;
        nd.a = make_float4(l0.x, l0.y, l0.z, h0.x);
        nd.b = make_float4(h0.y, h0.z, l1.x, l1.y);
        nd.c = make_float4(l1.z, h1.x, h1.y, h1.z);
        nd.d = make_int4(ch.x, ch.y, 0, 0);
        nodes[cur] = nd;
        node_lo[cur] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
        node_hi[cur] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), 1.0f + fmaxf(hgt0, hgt1));
        cur = node_parent[cur];
    }
}

// Single-triangle scene: one node whose second child is an empty box.
__global__ void k_single_node(const float4* __restrict__ tri_lo, const float4* __restrict__ tri_hi, BvhNode* __restrict__ nodes,
                              float4* __restrict__ node_hi)
{
    const float4 l = tri_lo[0], h = tri_hi[0];
    BvhNode nd;
    nd.a = make_float4(l.x, l.y, l.z, h.x);
    nd.b = make_float4(h.y, h.z, INFINITY, INFINITY);
    nd.c = make_float4(INFINITY, -INFINITY, -INFINITY, -INFINITY);
    nd.d = make_int4(~0, ~0, 0, 0);
    nodes[0] = nd;
    node_hi[0] = make_float4(h.x, h.y, h.z, 1.0f);
}

// --- 6. PLOC: parallel locally-ordered clustering over the Morton order (Meister & Bittner 2018) ---
// Same front end as the LBVH (Morton codes + radix sort); instead of splitting the sorted sequence by
// code bits (Karras), clusters are merged bottom-up: every cluster looks kPlocRadius neighbours to
// each side in the current order for the partner with the smallest merged surface area, mutual
// nearest neighbours merge, the sequence is compacted, repeat until one cluster is left.  Quality
// is that of a full SAH sweep (-29 % inner-node visits per ray on the Cornell scene against the
// Karras tree, profiles/r01_bvh_quality.txt).  Deterministic: ties go to the lower index (to hashed pairs once ties hold the merges up, k_ploc_nn) and node
// numbers come from prefix sums, not atomics.  Node 0 is the root (numbers are handed out downwards).
constexpr int kPlocRadius = 8;
constexpr uint32_t kReinsertIterations = 12;     // parallel reinsertion (13): at most this many find / lock / apply / refit rounds
constexpr uint32_t kDepthFirstTris = 50000;      // scenes above this many triangles get their nodes in depth-first order (6a): where the tree outgrows the caches

struct Cluster {
    float4 lo;   // .w = node reference as int bits (>= 0 inner node, < 0 leaf ~slot)
    float4 hi;   // .w = height (0 for a leaf)
};

__global__ void k_ploc_init(const uint32_t* __restrict__ vals_sorted, uint32_t n, const float4* __restrict__ tri_lo,
                            const float4* __restrict__ tri_hi, Cluster* __restrict__ c)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = vals_sorted[i];
    float4 l = tri_lo[p], h = tri_hi[p];
    l.w = __int_as_float(~(int)i);
    h.w = 0.0f;
    Cluster cl; cl.lo = l; cl.hi = h;
    c[i] = cl;
}

__device__ __forceinline__ float merged_area(const float4& al, const float4& ah, const float4& bl, const float4& bh)
{
    const float dx = fmaxf(ah.x, bh.x) - fminf(al.x, bl.x);
    const float dy = fmaxf(ah.y, bh.y) - fminf(al.y, bl.y);
    const float dz = fmaxf(ah.z, bh.z) - fminf(al.z, bl.z);
    return dx * dy + dy * dz + dz * dx;
}

// The round loop runs on the device's own counters: PlocState holds the current cluster count and the next free node number,
// every kernel of a round reads them, the last one advances them.  The host enqueues rounds in batches and looks at the
// count once per batch, instead of synchronising (and copying four words back) after every round.
struct PlocState { uint32_t m, next_top, kept, made, rounds, hashed_ties, pad1, pad2; };

__global__ void k_ploc_nn(const Cluster* __restrict__ c, const PlocState* __restrict__ st, uint32_t* __restrict__ nn)
{
    const uint32_t m = st->m;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || m < 2u) return;
    const float4 l = c[i].lo, h = c[i].hi;
    const uint32_t a = i > (uint32_t)kPlocRadius ? i - kPlocRadius : 0u;
    const uint32_t b = min(m, i + kPlocRadius + 1u);
    // Ties in the merged area go to the lower index.  That rule makes every cluster of a regular sequence — a strip of identical quads, a fence,
    // a staircase: each has two equally good neighbours — point the same way, so that one pair per round is mutual and the build takes
    // (almost) as many rounds as there are triangles: 100 000 triangles of a strip 1.77 s, tools/degenerate_scenes.py.  When the host sees
    // a batch of rounds that hardly shrank the sequence it sets st->hashed_ties: ties then go to the pair with the smaller hash of its lower
    // position — a key both partners compute alike —, a third of the positions are local minima of it and merge in the same round (3 ms).
    // Ordinary scenes never switch: their trees are those of rounds 1-3 (the hash as the general rule cost the 1.31 M-triangle scene 2.5 %).
    const bool hashed = st->hashed_ties != 0u;
    float best = INFINITY; uint32_t bj = i, bkey = 0xFFFFFFFFu;
    for (uint32_t j = a; j < b; j++) {
        if (j == i) continue;
        const float ar = merged_area(l, h, c[j].lo, c[j].hi);
        const uint32_t key = hashed ? (j < i ? j : i) * 0x9E3779B1u : 0u;
        if (ar < best || (ar == best && key < bkey)) { best = ar; bj = j; bkey = key; }      // ascending j: without the hash, ties keep the lower index
    }
    nn[i] = bj;
}

// keep[i] = 1 if position i survives (unmerged, or the lower partner of a merge); made[i] = 1 if a node is created at i
__global__ void k_ploc_flags(const uint32_t* __restrict__ nn, const PlocState* __restrict__ st, uint32_t* __restrict__ keep, uint32_t* __restrict__ made)
{
    const uint32_t m = st->m;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || m < 2u) return;
    const uint32_t j = nn[i];
    const bool mutual = (j != i) && (nn[j] == i);
    keep[i] = (!mutual || i < j) ? 1u : 0u;
    made[i] = (mutual && i < j) ? 1u : 0u;
}

// exclusive scans of keep[] and made[] over the current clusters, in place, one workgroup; totals into the state
__global__ void __launch_bounds__(1024) k_ploc_scan(uint32_t* __restrict__ keep, uint32_t* __restrict__ made, PlocState* __restrict__ st)
{
    __shared__ uint32_t pk[1024], pm[1024];
    const uint32_t count = st->m;
    if (count < 2u) return;
    const uint32_t per = (count + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per, e = min(count, b + per);
    uint32_t sk = 0, sm = 0;
    for (uint32_t i = b; i < e; i++) { sk += keep[i]; sm += made[i]; }
    pk[threadIdx.x] = sk; pm[threadIdx.x] = sm;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t vk = threadIdx.x >= off ? pk[threadIdx.x - off] : 0u, vm = threadIdx.x >= off ? pm[threadIdx.x - off] : 0u;
        __syncthreads();
        pk[threadIdx.x] += vk; pm[threadIdx.x] += vm;
        __syncthreads();
    }
    uint32_t rk = pk[threadIdx.x] - sk, rm = pm[threadIdx.x] - sm;
    for (uint32_t i = b; i < e; i++) { const uint32_t vk = keep[i], vm = made[i]; keep[i] = rk; made[i] = rm; rk += vk; rm += vm; }
    if (threadIdx.x == 1023u) { st->kept = pk[1023]; st->made = pm[1023]; }
}

__global__ void k_ploc_merge(const Cluster* __restrict__ cin, const uint32_t* __restrict__ nn, const PlocState* __restrict__ st,
                             const uint32_t* __restrict__ keep_pos, const uint32_t* __restrict__ made_pos,
                             Cluster* __restrict__ cout, BvhNode* __restrict__ nodes)
{
    const uint32_t m = st->m;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < 2u) { if (i == 0u) cout[0] = cin[0]; return; }     // finished: the host's blind ping-pong still finds the root cluster
    if (i >= m) return;
    const uint32_t next_node_top = st->next_top;                  // highest free node number
    const uint32_t j = nn[i];
    const bool mutual = (j != i) && (nn[j] == i);
    if (mutual && i > j) return;                   // absorbed by its partner
    Cluster me = cin[i];
    if (mutual) {
        const Cluster other = cin[j];
        const uint32_t node = next_node_top - made_pos[i];
        const int c0 = __float_as_int(me.lo.w), c1 = __float_as_int(other.lo.w);
        BvhNode nd;
        nd.a = make_float4(me.lo.x, me.lo.y, me.lo.z, me.hi.x);
        nd.b = make_float4(me.hi.y, me.hi.z, other.lo.x, other.lo.y);
        nd.c = make_float4(other.lo.z, other.hi.x, other.hi.y, other.hi.z);
        nd.d = make_int4(c0, c1, 0, 0);
        nodes[node] = nd;
        Cluster u;
        u.lo = make_float4(fminf(me.lo.x, other.lo.x), fminf(me.lo.y, other.lo.y), fminf(me.lo.z, other.lo.z), __int_as_float((int)node));
        u.hi = make_float4(fmaxf(me.hi.x, other.hi.x), fmaxf(me.hi.y, other.hi.y), fmaxf(me.hi.z, other.hi.z), 1.0f + fmaxf(me.hi.w, other.hi.w));
        me = u;
    }
    cout[keep_pos[i]] = me;
}

// after the merge of a round: the survivors are the new sequence
__global__ void k_ploc_advance(PlocState* __restrict__ st)
{
    if (st->m < 2u) return;
    st->next_top -= st->made;
    st->m = st->kept;
    st->rounds += 1u;
}


// --- 6a. depth-first numbering of the nodes (large scenes) ---------------------------------------------------------------------
// PLOC hands out node numbers by merge round (root last = 0), Karras by position in the sorted order: neither keeps a root-to-leaf
// descent inside few cache lines.  In depth-first (pre-)order a node is followed by its first child's whole subtree, so the deeper
// a ray is, the closer together the nodes it visits next: on the scenes that do not fit the L2 the render kernel's L2 hit rate goes
// from 0.61 to 0.68 (10.5 M triangles; 0.73 -> 0.76 at 1.31 M) and the bytes between L2 and fabric drop by a fifth, for 2.8 % / 0.8 % of the
// time (profiles/r04_ab_node_order.txt; sibling pairs in one 64-byte line: nothing).  Done on the fp32 nodes, before any other array
// is derived from them, so every format shares the numbering.  pre(child0) = pre(parent) + 1, pre(child1) = pre(parent) + 1 +
// |inner nodes under child0|: subtree sizes bottom-up (the second arrival at a node proceeds, as in k_refit), then every node
// walks up to the root adding what lies before it (tree depth steps).
__global__ void k_dfs_parents(const BvhNode* __restrict__ nodes, uint32_t n_nodes, int* __restrict__ parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int4 ch = nodes[i].d;
    if (ch.x >= 0) parent[ch.x] = (int)i;
    if (ch.y >= 0) parent[ch.y] = (int)i;
    if (i == 0u) parent[0] = -1;
}
__global__ void k_dfs_sizes(const BvhNode* __restrict__ nodes, uint32_t n_nodes, const int* __restrict__ parent, uint32_t* __restrict__ visit, uint32_t* __restrict__ size)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    int4 ch = nodes[i].d;
    if (ch.x >= 0 || ch.y >= 0) return;                      // start at the nodes whose children are both triangles
    int cur = (int)i;
    uint32_t s = 1u;
    for (;;) {
        size[cur] = s;
        const int p = parent[cur];
        if (p < 0) return;
        ch = nodes[p].d;
        const bool both_inner = ch.x >= 0 && ch.y >= 0;
        if (both_inner) {
            // release this subtree's size, acquire the sibling's: the second arrival proceeds
            const uint32_t prev = __hip_atomic_fetch_add(&visit[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == 0u) return;
            s = 1u + __hip_atomic_load(&size[ch.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + __hip_atomic_load(&size[ch.y], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            s = 1u + s;                                       // the other child is a triangle
        }
        cur = p;
    }
}
__global__ void k_dfs_index(const BvhNode* __restrict__ nodes, uint32_t n_nodes, const int* __restrict__ parent, const uint32_t* __restrict__ size, uint32_t* __restrict__ newidx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    uint32_t before = 0u;
    int cur = (int)i;
    for (int p = parent[cur]; p >= 0; cur = p, p = parent[cur]) {
        const int4 ch = nodes[p].d;
        before += 1u + ((ch.y == cur && ch.x >= 0) ? size[ch.x] : 0u);      // the parent itself, and the first child's subtree if this is the second
    }
    newidx[i] = before;
}
__global__ void k_dfs_permute(const BvhNode* __restrict__ nodes, uint32_t n_nodes, const uint32_t* __restrict__ newidx, BvhNode* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    BvhNode nd = nodes[i];
    if (nd.d.x >= 0) nd.d.x = (int)newidx[nd.d.x];
    if (nd.d.y >= 0) nd.d.y = (int)newidx[nd.d.y];
    out[newidx[i]] = nd;
}

// --- 6b. 16-bit grid copy of the nodes (experiment formats 1 / 2 / 4; built on first use from the fp32 nodes) -------------
__global__ void k_quant_nodes(const BvhNode* __restrict__ nodes, uint32_t n, const QGrid g, QNode* __restrict__ qn)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BvhNode nd = nodes[i];
    // child 0: lo (a.x a.y a.z) hi (a.w b.x b.y); child 1: lo (b.z b.w c.x) hi (c.y c.z c.w)
    qn[i] = quantise_node(g, make_float4(nd.a.x, nd.a.y, nd.a.z, 0.0f), make_float4(nd.a.w, nd.b.x, nd.b.y, 0.0f),
                          make_float4(nd.b.z, nd.b.w, nd.c.x, 0.0f), make_float4(nd.c.y, nd.c.z, nd.c.w, 0.0f), nd.d.x, nd.d.y);
}

// --- 7. centre / half-extent copy of the nodes ------------------------------------------------------
// Slab planes from a centre c and a half extent h need no per-axis min / max (half-rate instructions):
// near = (c - o)/d - h/|d|, far = (c - o)/d + h/|d|.  h is inflated by 1e-6 so that [c - h, c + h] contains the
// fp32 [lo, hi] it came from whatever way c and h rounded.  Same slots as BvhNode: lo -> c, hi -> h.
__device__ __forceinline__ void centre_half(float lo, float hi, float& c, float& h)
{
    if (!(lo <= hi)) { c = 0.0f; h = -1.0f; return; }       // empty child (single-triangle scene): never hit
    c = 0.5f * lo + 0.5f * hi;
    h = fmaxf(c - lo, hi - c) * 1.000001f + 1e-30f;
}
__global__ void k_centre_nodes(const BvhNode* __restrict__ nodes, uint32_t n, BvhNode* __restrict__ cn)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BvhNode nd = nodes[i];
    BvhNode o;
    // child 0: lo (a.x a.y a.z) hi (a.w b.x b.y); child 1: lo (b.z b.w c.x) hi (c.y c.z c.w)
    centre_half(nd.a.x, nd.a.w, o.a.x, o.a.w);
    centre_half(nd.a.y, nd.b.x, o.a.y, o.b.x);
    centre_half(nd.a.z, nd.b.y, o.a.z, o.b.y);
    centre_half(nd.b.z, nd.c.y, o.b.z, o.c.y);
    centre_half(nd.b.w, nd.c.z, o.b.w, o.c.z);
    centre_half(nd.c.x, nd.c.w, o.c.x, o.c.w);
    o.d = nd.d;
    cn[i] = o;
}

// --- 8. fp16 copy of the nodes -------------------------------------------------------------------------
// Planes go to the scene-centred, scaled space of HSpace and are rounded OUTWARD to fp16 (lo down, hi up),
// so every fp16 box contains its fp32 box.  area[0] / area[1] accumulate the child-box surface areas before / after, area[2] /
// area[3] the number of boxes and the sum of their own after / before ratios: the measures by which pt_set_scene decides whether
// the coarser planes are acceptable for this scene (render_megakernel.h kHalfAreaLimit, kHalfInflationLimit).
__global__ void k_half_nodes(const BvhNode* __restrict__ nodes, uint32_t n, HSpace sp, HNode* __restrict__ hn, float* __restrict__ area)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float before = 0.0f, after = 0.0f, boxes = 0.0f, inflation = 0.0f;
    if (i < n) {
        const BvhNode nd = nodes[i];
        const float scale = 1.0f / sp.inv_scale;
        float l[6], h[6];
        HNode o;
        // child 0: lo (a.x a.y a.z) hi (a.w b.x b.y); child 1: lo (b.z b.w c.x) hi (c.y c.z c.w)
        o.a = make_uint4(pack_planes(nd.a.x, nd.a.w, sp.cx, scale, l[0], h[0]), pack_planes(nd.a.y, nd.b.x, sp.cy, scale, l[1], h[1]),
                         pack_planes(nd.a.z, nd.b.y, sp.cz, scale, l[2], h[2]), (uint32_t)nd.d.x);
        o.b = make_uint4(pack_planes(nd.b.z, nd.c.y, sp.cx, scale, l[3], h[3]), pack_planes(nd.b.w, nd.c.z, sp.cy, scale, l[4], h[4]),
                         pack_planes(nd.c.x, nd.c.w, sp.cz, scale, l[5], h[5]), (uint32_t)nd.d.y);
        hn[i] = o;
        const float e0[3] = {nd.a.w - nd.a.x, nd.b.x - nd.a.y, nd.b.y - nd.a.z}, e1[3] = {nd.c.y - nd.b.z, nd.c.z - nd.b.w, nd.c.w - nd.c.x};
        const float b0 = e0[0] * e0[1] + e0[1] * e0[2] + e0[2] * e0[0], b1 = e1[0] * e1[1] + e1[1] * e1[2] + e1[2] * e1[0];
        if (e0[0] >= 0.0f) before += b0;
        if (e1[0] >= 0.0f) before += b1;
        const float g0[3] = {(h[0] - l[0]) * sp.inv_scale, (h[1] - l[1]) * sp.inv_scale, (h[2] - l[2]) * sp.inv_scale};
        const float g1[3] = {(h[3] - l[3]) * sp.inv_scale, (h[4] - l[4]) * sp.inv_scale, (h[5] - l[5]) * sp.inv_scale};
        const float a0 = g0[0] * g0[1] + g0[1] * g0[2] + g0[2] * g0[0], a1 = g1[0] * g1[1] + g1[1] * g1[2] + g1[2] * g1[0];
        if (e0[0] >= 0.0f) after += a0;
        if (e1[0] >= 0.0f) after += a1;
        // per box, unweighted: a cluster of triangles below the planes' resolution inflates its own boxes many times over
        // without moving the area sums, which the scene's large boxes dominate
        if (e0[0] >= 0.0f && b0 > 0.0f) { boxes += 1.0f; inflation += fminf(a0 / b0, 1e4f); }
        if (e1[0] >= 0.0f && b1 > 0.0f) { boxes += 1.0f; inflation += fminf(a1 / b1, 1e4f); }
    }
    for (int off = 32; off > 0; off >>= 1) {
        before += __shfl_xor(before, off); after += __shfl_xor(after, off);
        boxes += __shfl_xor(boxes, off); inflation += __shfl_xor(inflation, off);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&area[0], before); atomicAdd(&area[1], after); atomicAdd(&area[2], boxes); atomicAdd(&area[3], inflation); }
}

// --- 8b. fp16 centre / half-extent copy (NODE_FMT 11): child references of inner nodes as byte offsets (index * 32) -------------
__global__ void k_hc_nodes(const BvhNode* __restrict__ nodes, uint32_t n, HSpace sp, HNode* __restrict__ hn)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BvhNode nd = nodes[i];
    const float sx = 1.0f / sp.isx, sy = 1.0f / sp.isy, sz = 1.0f / sp.isz;
    HNode o;
    // child 0: lo (a.x a.y a.z) hi (a.w b.x b.y); child 1: lo (b.z b.w c.x) hi (c.y c.z c.w)
    o.a = make_uint4(pack_centre_half(nd.a.x, nd.a.w, sp.cx, sx), pack_centre_half(nd.a.y, nd.b.x, sp.cy, sy), pack_centre_half(nd.a.z, nd.b.y, sp.cz, sz),
                     nd.d.x >= 0 ? (uint32_t)nd.d.x << 5 : (uint32_t)nd.d.x);
    o.b = make_uint4(pack_centre_half(nd.b.z, nd.c.y, sp.cx, sx), pack_centre_half(nd.b.w, nd.c.z, sp.cy, sy), pack_centre_half(nd.c.x, nd.c.w, sp.cz, sz),
                     nd.d.y >= 0 ? (uint32_t)nd.d.y << 5 : (uint32_t)nd.d.y);
    hn[i] = o;
}

// --- 9. the top of the tree, breadth first ---------------------------------------------------------------------------------
// The first `cap` inner nodes in breadth-first order from the root, as a small array of their own: a child that is in the
// array too is referenced as kTopNodeFlag | position, every other child as in `hn`.  Render kernels that stage the top of the
// tree in LDS (every ray walks it) traverse this copy.  One thread: 255 nodes at most.
// ids[i] = index in `hn` of the node at position i (a kernel that stages fewer than `cap` nodes turns references past its
// own cut back into those).
__global__ void k_top_nodes(const HNode* __restrict__ hn, HNode* __restrict__ top, uint32_t* __restrict__ ids, uint32_t cap, uint32_t* __restrict__ n_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t count = 1;
    ids[0] = 0u;
    for (uint32_t i = 0; i < count; i++) {
        HNode nd = hn[ids[i]];
        const int c0 = (int)nd.a.w, c1 = (int)nd.b.w;
        if (c0 >= 0 && count < cap) { ids[count] = (uint32_t)c0; nd.a.w = kTopNodeFlag | count; count++; }
        if (c1 >= 0 && count < cap) { ids[count] = (uint32_t)c1; nd.b.w = kTopNodeFlag | count; count++; }
        top[i] = nd;
    }
    *n_out = count;
}

// --- 10. the fp32 nodes again, from what a scene keeps -----------------------------------------------------------------------
// A scene whose kernel reads the fp16 nodes does not keep the 64-byte fp32 nodes on the device (they are twice the size of
// what is traversed).  Whoever needs them later (the ray queries of the parity tests, a switch to an fp32 kernel variant, the
// experiment formats derived from them) gets them back from the topology in the fp16 nodes and the triangle records: leaf boxes by
// record_aabb — the very function the build used — and unions bottom-up, which are exact and order-independent: the same bits
// as the build's own array.
// (shift: 0 for the child references of HNode, 5 for those of the centre / half-extent copy, whose inner references are byte offsets)
__device__ __forceinline__ int h_child(uint32_t w, int shift) { const int c = (int)w; return c >= 0 ? c >> shift : c; }
__global__ void k_parents_of(const HNode* __restrict__ hn, int shift, uint32_t n_nodes, int* __restrict__ node_parent, int* __restrict__ leaf_parent)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int c0 = h_child(hn[i].a.w, shift), c1 = h_child(hn[i].b.w, shift);
    if (c0 >= 0) node_parent[c0] = (int)i; else leaf_parent[~c0] = (int)i;
    if (c1 >= 0) node_parent[c1] = (int)i; else leaf_parent[~c1] = (int)i;
    if (i == 0u) node_parent[0] = -1;
}
__global__ void k_refit_records(int n, const TriRecord* __restrict__ tris, float pad_abs, const HNode* __restrict__ hn, int shift,
                                const int* __restrict__ node_parent, const int* __restrict__ leaf_parent, uint32_t* __restrict__ visit,
                                float4* __restrict__ node_lo, float4* __restrict__ node_hi, BvhNode* __restrict__ nodes)
{
    const int leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n) return;
    int cur = leaf_parent[leaf];
    while (cur >= 0) {
        const uint32_t prev = __hip_atomic_fetch_add(&visit[cur], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);      // as k_refit: the second arrival proceeds
        if (prev == 0) return;
        const int c0 = h_child(hn[cur].a.w, shift), c1 = h_child(hn[cur].b.w, shift);
        float l0[3], h0[3], l1[3], h1[3];
        if (c0 < 0) record_aabb(tris[~c0], pad_abs, l0, h0);
        else { const float4 a = node_lo[c0], b = node_hi[c0]; l0[0] = a.x; l0[1] = a.y; l0[2] = a.z; h0[0] = b.x; h0[1] = b.y; h0[2] = b.z; }
        if (c1 < 0) record_aabb(tris[~c1], pad_abs, l1, h1);
        else { const float4 a = node_lo[c1], b = node_hi[c1]; l1[0] = a.x; l1[1] = a.y; l1[2] = a.z; h1[0] = b.x; h1[1] = b.y; h1[2] = b.z; }
        BvhNode nd;
        nd.a = make_float4(l0[0], l0[1], l0[2], h0[0]);
        nd.b = make_float4(h0[1], h0[2], l1[0], l1[1]);
        nd.c = make_float4(l1[2], h1[0], h1[1], h1[2]);
        nd.d = make_int4(c0, c1, 0, 0);
        nodes[cur] = nd;
        node_lo[cur] = make_float4(fminf(l0[0], l1[0]), fminf(l0[1], l1[1]), fminf(l0[2], l1[2]), 0.0f);
        node_hi[cur] = make_float4(fmaxf(h0[0], h1[0]), fmaxf(h0[1], h1[1]), fmaxf(h0[2], h1[2]), 0.0f);
        cur = node_parent[cur];
    }
}
// single-triangle scene: k_single_node's node from the record
__global__ void k_single_node_record(const TriRecord* __restrict__ tris, float pad_abs, BvhNode* __restrict__ nodes)
{
    float l[3], h[3];
    record_aabb(tris[0], pad_abs, l, h);
    BvhNode nd;
    nd.a = make_float4(l[0], l[1], l[2], h[0]);
    nd.b = make_float4(h[1], h[2], INFINITY, INFINITY);
    nd.c = make_float4(INFINITY, -INFINITY, -INFINITY, -INFINITY);
    nd.d = make_int4(~0, ~0, 0, 0);
    nodes[0] = nd;
}

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return false; } } while (0)

namespace {
// scratch allocations of one build, released on every exit path
struct Scratch {
    std::vector<void*> ptrs;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    template <typename T> hipError_t alloc(T** p, size_t bytes)
    {
        hipError_t e = hipMalloc((void**)p, bytes ? bytes : 4);
        if (e == hipSuccess) ptrs.push_back((void*)*p);
        return e;
    }
    ~Scratch()
    {
        for (void* p : ptrs) (void)hipFree(p);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
    }
};
}  // namespace

void free_lbvh(LbvhResult& r)
{
    if (r.nodes) (void)hipFree(r.nodes);
    if (r.qnodes) (void)hipFree(r.qnodes);
    if (r.cnodes) (void)hipFree(r.cnodes);
    if (r.hnodes) (void)hipFree(r.hnodes);
    if (r.top_nodes) (void)hipFree(r.top_nodes);
    if (r.tris) (void)hipFree(r.tris);
    if (r.shade) (void)hipFree(r.shade);
    if (r.wrecs) (void)hipFree(r.wrecs);
    if (r.srecs) (void)hipFree(r.srecs);
    if (r.hcnodes) (void)hipFree(r.hcnodes);
    if (r.hcnodes_alt) (void)hipFree(r.hcnodes_alt);
    r = LbvhResult();
}

// --- 12. insertion-based optimisation of small trees (build mode 2; host) --------------------------------------------------------
// Bittner, Hapala, Havran 2013: take a node out of the tree (its parent goes with it, its sibling moves up), then put its two
// subtrees back where they enlarge the tree least — found by a best-first search over the tree with the surface area the insertion
// would add along the way as the bound.  Large flat triangles (walls) that PLOC's local merges buried deep move up next to the root;
// subtrees of small triangles get boxes that overlap less.  The sum of the inner nodes' surface areas — what a random ray pays in
// visits — drops by 4.9 % on the Cornell scenes after two passes (a third finds nothing: PLOC's tree is close to this optimum), the
// render's BVH-loop trips by 2.5 %, its time by 2.5 % / 1.9 % on configs 2 / 3 (profiles/r04_ab_tree_optimisation.txt).  Scenes up to
// kOptimizeMaxTris triangles: their tree is a few hundred KB, the passes take milliseconds on one host thread; deterministic (no
// hashing, no threads; ties by index).  Boxes only prune: every hit stays bit-exact whatever the tree.  The default build (mode 2) for
// scenes up to kOptimizeMaxTris triangles; larger scenes, and mode 1, keep the PLOC tree as it is.
constexpr uint32_t kOptimizeMaxTris = 16384;      // ~0.2 s of host time at the limit; 12 ms for the 1 264 triangles of the Cornell scenes
namespace {
struct OBox { float lo[3], hi[3]; };
inline OBox obox_union(const OBox& a, const OBox& b)
{ OBox r; for (int k = 0; k < 3; k++) { r.lo[k] = fminf(a.lo[k], b.lo[k]); r.hi[k] = fmaxf(a.hi[k], b.hi[k]); } return r; }
inline float obox_area(const OBox& b)
{ const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2]; return x * y + y * z + z * x; }
inline bool obox_same(const OBox& a, const OBox& b) { return memcmp(&a, &b, sizeof(OBox)) == 0; }

struct OTree {
    int m = 0;                                   // inner nodes 0 .. m-1, root = 0; a reference >= 0 is an inner node, < 0 a leaf ~slot
    std::vector<OBox> nbox, lbox;                // boxes of the inner nodes / of the leaves
    std::vector<int> nparent, lparent;           // parent node of an inner node (-1: root) / of a leaf
    std::vector<int> child;                      // 2 per inner node
    const OBox& box(int r) const { return r >= 0 ? nbox[(size_t)r] : lbox[(size_t)~r]; }
    int& parent(int r) { return r >= 0 ? nparent[(size_t)r] : lparent[(size_t)~r]; }
    void refit_from(int node)
    {
        while (node >= 0) {
            const OBox b = obox_union(box(child[2 * (size_t)node]), box(child[2 * (size_t)node + 1]));
            if (obox_same(b, nbox[(size_t)node])) return;
            nbox[(size_t)node] = b;
            node = nparent[(size_t)node];
        }
    }
    double area_sum() const { double s = 0.0; for (int i = 0; i < m; i++) s += obox_area(nbox[(size_t)i]); return s; }
    // where does a subtree with box `b` enlarge the tree least?  best-first over the induced cost (area added to the ancestors)
    int find_insertion(const OBox& b, std::vector<std::pair<float, int>>& heap) const
    {
        const float ab = obox_area(b);
        float best = INFINITY; int best_ref = 0;
        heap.clear();
        heap.emplace_back(0.0f, 0);
        const auto cmp = [](const std::pair<float, int>& x, const std::pair<float, int>& y) { return x.first > y.first || (x.first == y.first && x.second > y.second); };
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            const std::pair<float, int> e = heap.back(); heap.pop_back();
            if (e.first + ab >= best) break;
            const float direct = obox_area(obox_union(box(e.second), b));
            const float total = e.first + direct;
            if (total < best) { best = total; best_ref = e.second; }
            if (e.second >= 0) {
                const float induced = total - obox_area(nbox[(size_t)e.second]);
                if (induced + ab < best) {
                    heap.emplace_back(induced, child[2 * (size_t)e.second]); std::push_heap(heap.begin(), heap.end(), cmp);
                    heap.emplace_back(induced, child[2 * (size_t)e.second + 1]); std::push_heap(heap.begin(), heap.end(), cmp);
                }
            }
        }
        return best_ref;
    }
    // hang subtree `sub` and the reference `at` under the free node `q`, which takes `at`'s place
    void insert_at(int at, int sub, int q)
    {
        if (at == 0) {          // the root keeps number 0: its content moves into q, and the root becomes the parent of q and sub
            child[2 * (size_t)q] = child[0]; child[2 * (size_t)q + 1] = child[1];
            parent(child[0]) = q; parent(child[1]) = q;
            nbox[(size_t)q] = nbox[0];
            child[0] = q; child[1] = sub;
            nparent[(size_t)q] = 0; parent(sub) = 0;
            nbox[0] = obox_union(nbox[(size_t)q], box(sub));
            return;
        }
        const int p = parent(at);
        child[2 * (size_t)p + (child[2 * (size_t)p] == at ? 0 : 1)] = q;
        nparent[(size_t)q] = p;
        child[2 * (size_t)q] = at; child[2 * (size_t)q + 1] = sub;
        parent(at) = q; parent(sub) = q;
        nbox[(size_t)q] = obox_union(box(at), box(sub));
        refit_from(p);
    }
};
}  // namespace

// nodes: the fp32 nodes of the build (host copy), rewritten in place; returns the tree height (inner nodes on the longest root-to-leaf path)
static uint32_t optimize_tree_host(std::vector<BvhNode>& nodes, uint32_t n_tris, double* area_before, double* area_after, int* passes_done)
{
    OTree t;
    t.m = (int)nodes.size();
    t.nbox.resize((size_t)t.m); t.nparent.assign((size_t)t.m, -1); t.child.resize(2 * (size_t)t.m);
    t.lbox.resize(n_tris); t.lparent.assign(n_tris, -1);
    for (int i = 0; i < t.m; i++) {
        const BvhNode& nd = nodes[(size_t)i];
        // child 0: lo (a.x a.y a.z) hi (a.w b.x b.y); child 1: lo (b.z b.w c.x) hi (c.y c.z c.w)
        const OBox b0 = {{nd.a.x, nd.a.y, nd.a.z}, {nd.a.w, nd.b.x, nd.b.y}}, b1 = {{nd.b.z, nd.b.w, nd.c.x}, {nd.c.y, nd.c.z, nd.c.w}};
        const int c[2] = {nd.d.x, nd.d.y};
        const OBox* cb[2] = {&b0, &b1};
        for (int k = 0; k < 2; k++) {
            t.child[2 * (size_t)i + k] = c[k];
            if (c[k] >= 0) { t.nbox[(size_t)c[k]] = *cb[k]; t.nparent[(size_t)c[k]] = i; }
            else { t.lbox[(size_t)~c[k]] = *cb[k]; t.lparent[(size_t)~c[k]] = i; }
        }
        if (i == 0) t.nbox[0] = obox_union(b0, b1);
    }
    *area_before = t.area_sum();
    std::vector<std::pair<float, int>> heap;
    std::vector<std::pair<float, int>> order;
    double prev = *area_before;
    int pass = 0;
    const int max_passes = getenv("ACGPT_OPT_PASSES") ? atoi(getenv("ACGPT_OPT_PASSES")) : 8;
    const double stop = getenv("ACGPT_OPT_STOP") ? atof(getenv("ACGPT_OPT_STOP")) : 0.995;
    for (; pass < max_passes; pass++) {
        // largest nodes first: they are the ones a wrong place costs most
        order.clear();
        for (int i = 1; i < t.m; i++) order.emplace_back(-obox_area(t.nbox[(size_t)i]), i);
        std::sort(order.begin(), order.end());
        for (const auto& oc : order) {
            const int nn = oc.second;
            const int p = t.nparent[(size_t)nn];
            if (p <= 0) continue;                                   // children of the root stay (the root keeps its number and its place)
            const int g = t.nparent[(size_t)p];
            const int sib = t.child[2 * (size_t)p + (t.child[2 * (size_t)p] == nn ? 1 : 0)];
            const int l = t.child[2 * (size_t)nn], r = t.child[2 * (size_t)nn + 1];
            // take nn and its parent out: the sibling moves up
            t.child[2 * (size_t)g + (t.child[2 * (size_t)g] == p ? 0 : 1)] = sib;
            t.parent(sib) = g;
            t.refit_from(g);
            // ... and put nn's two subtrees back, the larger first, where they add least
            const bool l_first = obox_area(t.box(l)) >= obox_area(t.box(r));
            const int first = l_first ? l : r, second = l_first ? r : l;
            t.insert_at(t.find_insertion(t.box(first), heap), first, p);
            t.insert_at(t.find_insertion(t.box(second), heap), second, nn);
        }
        const double now = t.area_sum();
        if (!(now < prev * stop)) { pass++; break; }
        prev = now;
    }
    *area_after = t.area_sum();
    *passes_done = pass;
    // back into the node array; the tree's height by an explicit stack
    for (int i = 0; i < t.m; i++) {
        const int c0 = t.child[2 * (size_t)i], c1 = t.child[2 * (size_t)i + 1];
        const OBox& b0 = t.box(c0); const OBox& b1 = t.box(c1);
        BvhNode nd;
        nd.a = make_float4(b0.lo[0], b0.lo[1], b0.lo[2], b0.hi[0]);
        nd.b = make_float4(b0.hi[1], b0.hi[2], b1.lo[0], b1.lo[1]);
        nd.c = make_float4(b1.lo[2], b1.hi[0], b1.hi[1], b1.hi[2]);
        nd.d = make_int4(c0, c1, 0, 0);
        nodes[(size_t)i] = nd;
    }
    uint32_t height = 0, leaves = 0, inner = 0;
    std::vector<std::pair<int, uint32_t>> st(1, std::make_pair(0, 1u));
    while (!st.empty()) {
        const std::pair<int, uint32_t> e = st.back(); st.pop_back();
        inner++;
        if (e.second > height) height = e.second;
        for (int k = 0; k < 2; k++) {
            const int c = t.child[2 * (size_t)e.first + k];
            if (c >= 0) st.emplace_back(c, e.second + 1u); else leaves++;
        }
        if (inner > (uint32_t)t.m) break;
    }
    return (inner == (uint32_t)t.m && leaves == n_tris) ? height : 0u;      // 0: the tree lost a node (a bug): the caller keeps the unoptimised one
}

// --- 13. parallel reinsertion on the device (build mode 2, scenes above kOptimizeMaxTris) ---------------------------------------
// The same optimisation for trees too large for one host thread (1.31 M triangles: 24 s there), after Meister & Bittner 2018
// ("Parallel reinsertion for bounding volume hierarchy optimization"): every node looks — read only — for the place where moving
// it, with its subtree, shrinks the tree most; moves whose paths through the tree (node up to the lowest common ancestor and down to
// the target) do not touch are applied together, the larger gain winning a contested node; boxes are refitted; repeat.
//   find   walk up from the node: at every ancestor A the node's side shrinks (gain: the parent disappears, the ancestors below A
//          lose the node's box) and the subtree of A's OTHER child is searched depth first, stackless (parent links), for the
//          target y that pays least: area(B u y) for the new node + the growth of the nodes between A and y; pruned by the bound
//          that what is left of the gain cannot beat the best found.
//   lock   atomicMax of (gain, node): an `edit` word on the six nodes the move rewrites, a `through` word on the nodes it only grows (below).
//   apply  the moves that hold all their locks: the sibling takes the parent's place, the parent becomes the new node (y, x) where y was.
//   refit  bottom-up from the leaves, the second arrival at a node proceeds.
// Unified node ids: inner nodes 0 .. m-1 (root 0, never moved), leaf slot s = m + s.  Deterministic (the maximum key is unique).
struct ReinsTree {
    int* parent;        // [m + n]
    int2* child;        // [m] unified ids
    float4* lo;         // [m + n]
    float4* hi;
    uint32_t m, n;
};
__device__ __forceinline__ float ri_area(const float4& l, const float4& h) { const float x = h.x - l.x, y = h.y - l.y, z = h.z - l.z; return x * y + y * z + z * x; }
__device__ __forceinline__ float ri_union_area(const float4& l, const float4& h, const float4& bl, const float4& bh)
{
    const float x = fmaxf(h.x, bh.x) - fminf(l.x, bl.x), y = fmaxf(h.y, bh.y) - fminf(l.y, bl.y), z = fmaxf(h.z, bh.z) - fminf(l.z, bl.z);
    return x * y + y * z + z * x;
}
__global__ void k_ri_init(const BvhNode* __restrict__ nodes, ReinsTree t)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.m) return;
    const BvhNode nd = nodes[i];
    // child 0: lo (a.x a.y a.z) hi (a.w b.x b.y); child 1: lo (b.z b.w c.x) hi (c.y c.z c.w)
    const int c0 = nd.d.x >= 0 ? nd.d.x : (int)t.m + ~nd.d.x, c1 = nd.d.y >= 0 ? nd.d.y : (int)t.m + ~nd.d.y;
    t.child[i] = make_int2(c0, c1);
    t.parent[c0] = (int)i; t.parent[c1] = (int)i;
    t.lo[c0] = make_float4(nd.a.x, nd.a.y, nd.a.z, 0.0f); t.hi[c0] = make_float4(nd.a.w, nd.b.x, nd.b.y, 0.0f);
    t.lo[c1] = make_float4(nd.b.z, nd.b.w, nd.c.x, 0.0f); t.hi[c1] = make_float4(nd.c.y, nd.c.z, nd.c.w, 0.0f);
    if (i == 0u) {
        t.parent[0] = -1;
        t.lo[0] = make_float4(fminf(nd.a.x, nd.b.z), fminf(nd.a.y, nd.b.w), fminf(nd.a.z, nd.c.x), 0.0f);
        t.hi[0] = make_float4(fmaxf(nd.a.w, nd.c.y), fmaxf(nd.b.x, nd.c.z), fmaxf(nd.b.y, nd.c.w), 0.0f);
    }
}
// best move of node x: target[x] = y (or -1), pivot[x] = the lowest common ancestor of the move, gain[x]
__global__ void k_ri_find(ReinsTree t, int* __restrict__ target, int* __restrict__ pivot_out, float* __restrict__ gain_out)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= t.m + t.n) return;
    target[x] = -1; gain_out[x] = 0.0f; pivot_out[x] = -1;
    if (x == 0u) return;
    const int p = t.parent[x];
    if (p <= 0) return;                                  // children of the root stay: the root keeps its number
    const float4 bl = t.lo[x], bh = t.hi[x];
    const float ab = ri_area(bl, bh);
    float best = 0.0f; int best_y = -1, best_pivot = -1;
    const int2 pc = t.child[p];
    int sib = pc.x == (int)x ? pc.y : pc.x;
    float4 wl = t.lo[sib], wh = t.hi[sib];               // box of the subtree that takes p's place
    float gain = ri_area(t.lo[p], t.hi[p]);              // p disappears
    int cur = p, other = sib, piv = p;
    for (int level = 0; level < 64; level++) {
        // depth-first, stackless, over the subtree of `other`; induced = growth of the nodes from `other` down to the parent of `node`
        int node = other;
        float induced = 0.0f;
        bool down = true;
        for (int guard = 0; guard < (1 << 20); guard++) {
            if (down) {
                const float4 nl = t.lo[node], nh = t.hi[node];
                const float direct = ri_union_area(nl, nh, bl, bh);
                const float g = gain - induced - direct;
                if (g > best && !(piv == p && node == sib)) { best = g; best_y = node; best_pivot = piv; }
                const float growth = direct - ri_area(nl, nh);
                if (node < (int)t.m && gain - (induced + growth) - ab > best) { induced += growth; node = t.child[node].x; }
                else down = false;
            }
            if (!down) {
                if (node == other) break;
                const int par = t.parent[node];
                const int2 cc = t.child[par];
                if (cc.x == node) { node = cc.y; down = true; }
                else {
                    induced -= ri_union_area(t.lo[par], t.hi[par], bl, bh) - ri_area(t.lo[par], t.hi[par]);
                    if (induced < 0.0f) induced = 0.0f;
                    node = par;
                }
            }
        }
        // one level up: the parent of `cur` becomes the pivot.  `cur` then lies strictly between p and the pivot and shrinks to its box without
        // x (wl, wh) — unless it is p itself, which is gone altogether and already counted
        const int a = t.parent[cur];
        if (a < 0) break;
        if (cur != p) gain += ri_area(t.lo[cur], t.hi[cur]) - ri_area(wl, wh);
        const int2 ac = t.child[a];
        other = ac.x == cur ? ac.y : ac.x;
        {   // the box of `a` without x, for when `a` itself lies below a later pivot
            const float4 ol = t.lo[other], oh = t.hi[other];
            wl = make_float4(fminf(wl.x, ol.x), fminf(wl.y, ol.y), fminf(wl.z, ol.z), 0.0f);
            wh = make_float4(fmaxf(wh.x, oh.x), fmaxf(wh.y, oh.y), fmaxf(wh.z, oh.z), 0.0f);
        }
        cur = a; piv = a;
        if (gain - ab <= best) break;                    // even a free insertion further up could not beat the best found
    }
    if (best_y >= 0 && best > 0.0f) { target[x] = best_y; pivot_out[x] = best_pivot; gain_out[x] = best; }
}

// What two moves must not share.  A move EDITS six nodes — x, its parent p (which becomes the new node), p's parent g and x's sibling (the
// sibling takes p's place), the target y and y's parent — and it INSERTS THROUGH the nodes between y's parent and the pivot (their
// boxes grow; nothing is written to them).  Two moves with disjoint edit sets write disjoint words.  A cycle — x under x' and x' under x
// — needs each of the two moved roots to lie on the other's insertion path, so a move also loses against a stronger one that inserts
// through a node it edits, and against a stronger one that edits a node it inserts through.  Two lock words per node (atomicMax of
// (gain, node)): `edit` and `through`.  Far fewer conflicts than locking the whole path from x over the pivot to y (what the first
// version did: 30 k moves of 2.6 M nodes in the first round on 1.31 M triangles).
struct RiEdit { int n[6]; };
__device__ __forceinline__ RiEdit ri_edit_set(const ReinsTree& t, int x, int y)
{
    RiEdit e;
    const int p = t.parent[x];
    const int2 pc = t.child[p];
    e.n[0] = x; e.n[1] = p; e.n[2] = t.parent[p]; e.n[3] = pc.x == x ? pc.y : pc.x; e.n[4] = y; e.n[5] = t.parent[y];
    return e;
}
template <typename F>
__device__ __forceinline__ void ri_for_through(const ReinsTree& t, int y, int pivot, F f)
{
    int guard = 0;          // (at most the tree's height; the bound turns a corrupted tree into a wrong result instead of a hang)
    for (int a = t.parent[y]; a != pivot && a >= 0 && guard < 4096; a = t.parent[a], guard++) f(a);
}
__global__ void k_ri_lock(ReinsTree t, const int* __restrict__ target, const int* __restrict__ pivot, const float* __restrict__ gain,
                          unsigned long long* __restrict__ lock_edit, unsigned long long* __restrict__ lock_through)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= t.m + t.n || target[x] < 0) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(gain[x]) << 32) | x;       // gains are positive: their bits order as they do
    const RiEdit e = ri_edit_set(t, (int)x, target[x]);
#pragma unroll
    for (int k = 0; k < 6; k++) if (e.n[k] >= 0) atomicMax(&lock_edit[e.n[k]], key);
    ri_for_through(t, target[x], pivot[x], [&](int a) { atomicMax(&lock_through[a], key); });
}
__global__ void k_ri_check(ReinsTree t, const int* __restrict__ target, const int* __restrict__ pivot, const float* __restrict__ gain,
                           const unsigned long long* __restrict__ lock_edit, const unsigned long long* __restrict__ lock_through,
                           uint32_t* __restrict__ winner, uint32_t* __restrict__ n_winners)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= t.m + t.n) return;
    winner[x] = 0u;
    if (target[x] < 0) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(gain[x]) << 32) | x;
    const RiEdit e = ri_edit_set(t, (int)x, target[x]);
    bool all = true;
#pragma unroll
    for (int k = 0; k < 6; k++) if (e.n[k] >= 0) all = all && lock_edit[e.n[k]] == key && lock_through[e.n[k]] <= key;
    ri_for_through(t, target[x], pivot[x], [&](int a) { all = all && lock_edit[a] <= key; });
    if (all) { winner[x] = 1u; atomicAdd(n_winners, 1u); }
}
// the moves that hold all their locks; every node they write is theirs alone (k_ri_check read the tree before any of this)
__global__ void k_ri_apply(ReinsTree t, const int* __restrict__ target, const uint32_t* __restrict__ winner)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= t.m + t.n || winner[x] == 0u) return;
    const int y = target[x];
    const int p = t.parent[x];
    const int g = t.parent[p];
    const int2 pc = t.child[p];
    const int sib = pc.x == (int)x ? pc.y : pc.x;
    // the sibling takes p's place under g
    int2 gc = t.child[g];
    if (gc.x == p) gc.x = sib; else gc.y = sib;
    t.child[g] = gc;
    t.parent[sib] = g;
    // p becomes the new node (y, x) where y was (y's parent is read AFTER the step above: it may be g)
    const int yp = t.parent[y];
    int2 yc = t.child[yp];
    if (yc.x == y) yc.x = p; else yc.y = p;
    t.child[yp] = yc;
    t.parent[p] = yp;
    t.child[p] = make_int2(y, (int)x);
    t.parent[y] = p;
}
__global__ void k_ri_refit(ReinsTree t, uint32_t* __restrict__ visit)
{
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= t.n) return;
    int cur = t.parent[t.m + leaf];
    for (int guard = 0; cur >= 0 && guard < 4096; guard++) {
        const uint32_t prev = __hip_atomic_fetch_add(&visit[cur], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == 0u) return;
        const int2 c = t.child[cur];
        const float4 l0 = t.lo[c.x], h0 = t.hi[c.x], l1 = t.lo[c.y], h1 = t.hi[c.y];
        t.lo[cur] = make_float4(fminf(l0.x, l1.x), fminf(l0.y, l1.y), fminf(l0.z, l1.z), 0.0f);
        t.hi[cur] = make_float4(fmaxf(h0.x, h1.x), fmaxf(h0.y, h1.y), fmaxf(h0.z, h1.z), fmaxf(h0.w, h1.w) + 1.0f);     // .w of hi: height (0 for a leaf)
        cur = t.parent[cur];
    }
}
__global__ void k_ri_area(ReinsTree t, double* __restrict__ sum)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    double a = i < t.m ? (double)ri_area(t.lo[i], t.hi[i]) : 0.0;
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if ((threadIdx.x & 63) == 0 && a != 0.0) atomicAdd(sum, a);
}
__global__ void k_ri_write(ReinsTree t, BvhNode* __restrict__ nodes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.m) return;
    const int2 c = t.child[i];
    const float4 l0 = t.lo[c.x], h0 = t.hi[c.x], l1 = t.lo[c.y], h1 = t.hi[c.y];
    BvhNode nd;
    nd.a = make_float4(l0.x, l0.y, l0.z, h0.x);
    nd.b = make_float4(h0.y, h0.z, l1.x, l1.y);
    nd.c = make_float4(l1.z, h1.x, h1.y, h1.z);
    nd.d = make_int4(c.x < (int)t.m ? c.x : ~(c.x - (int)t.m), c.y < (int)t.m ? c.y : ~(c.y - (int)t.m), 0, 0);
    nodes[i] = nd;
}

static bool build_impl(const float* h_verts_xyzw, size_t n_verts, const uint32_t* h_idx, uint32_t n,
                       const uint32_t* h_mat_ids, int mode, hipStream_t stream, LbvhResult& out, std::string& err)
{
    Scratch sc;
    float4* d_verts; uint32_t* d_idx; uint32_t* d_mat;
    TriRecord* d_unsorted; float4 *d_tlo, *d_thi, *d_nlo, *d_nhi;
    uint32_t *d_bounds, *d_keys[2], *d_vals[2], *d_hist, *d_visit;
    int2* d_children; int *d_nparent, *d_lparent;
    const uint32_t n_nodes = n > 1 ? n - 1 : 1;
    const uint32_t blocks = (n + 255) / 256;
    const uint32_t sort_blocks = (n + kSortTile - 1) / kSortTile;
    const uint32_t init_bounds[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    uint32_t h_bounds[6];
    float4 root_hi;
    int cur = 0;

    float coord_max = 1.0f;
    for (size_t i = 0; i < n_verts; i++)
        for (int k = 0; k < 3; k++) { const float a = fabsf(h_verts_xyzw[4 * i + k]); if (a > coord_max && a < INFINITY) coord_max = a; }
    const float pad_abs = coord_max * (1.0f / 524288.0f);
    HIPCK(sc.alloc(&d_verts, n_verts * 16));
    HIPCK(sc.alloc(&d_idx, (size_t)n * 12));
    HIPCK(sc.alloc(&d_mat, (size_t)n * 4));
    HIPCK(sc.alloc(&d_unsorted, (size_t)n * sizeof(TriRecord)));
    HIPCK(sc.alloc(&d_tlo, (size_t)n * 16));
    HIPCK(sc.alloc(&d_thi, (size_t)n * 16));
    HIPCK(sc.alloc(&d_nlo, (size_t)n_nodes * 16));
    HIPCK(sc.alloc(&d_nhi, (size_t)n_nodes * 16));
    HIPCK(sc.alloc(&d_bounds, 24));
    for (int k = 0; k < 2; k++) { HIPCK(sc.alloc(&d_keys[k], (size_t)n * 4)); HIPCK(sc.alloc(&d_vals[k], (size_t)n * 4)); }
    HIPCK(sc.alloc(&d_hist, (size_t)256 * sort_blocks * 4));
    HIPCK(sc.alloc(&d_visit, (size_t)n_nodes * 4));
    HIPCK(sc.alloc(&d_children, (size_t)n_nodes * 8));
    HIPCK(sc.alloc(&d_nparent, (size_t)n_nodes * 4));
    HIPCK(sc.alloc(&d_lparent, (size_t)n * 4));
    // what a scene keeps: the fp32 nodes (the build's own output), their fp16 copy, the triangle and shading records.  The caller
    // releases the node arrays its kernel does not read (keep_one_node_array); any of them comes back on first use
    // (ensure_nodes / ensure_hnodes), and so do the experiment formats (ensure_qnodes, ensure_cnodes).
    HIPCK(hipMalloc((void**)&out.nodes, (size_t)n_nodes * sizeof(BvhNode)));
    HIPCK(hipMalloc((void**)&out.hnodes, (size_t)n_nodes * sizeof(HNode)));
    HIPCK(hipMalloc((void**)&out.tris, (size_t)n * sizeof(TriRecord)));
    HIPCK(hipMalloc((void**)&out.shade, (size_t)n * sizeof(float4)));
    out.pad_abs = pad_abs;

    HIPCK(hipMemcpyAsync(d_verts, h_verts_xyzw, n_verts * 16, hipMemcpyHostToDevice, stream));
    HIPCK(hipMemcpyAsync(d_idx, h_idx, (size_t)n * 12, hipMemcpyHostToDevice, stream));
    HIPCK(hipMemcpyAsync(d_mat, h_mat_ids, (size_t)n * 4, hipMemcpyHostToDevice, stream));
    HIPCK(hipMemcpyAsync(d_bounds, init_bounds, 24, hipMemcpyHostToDevice, stream));
    HIPCK(hipMemsetAsync(d_visit, 0, (size_t)n_nodes * 4, stream));
    HIPCK(hipEventCreate(&sc.ev0));
    HIPCK(hipEventCreate(&sc.ev1));
    HIPCK(hipEventRecord(sc.ev0, stream));

    k_prepare<<<blocks, 256, 0, stream>>>(d_verts, d_idx, d_mat, n, d_unsorted, d_tlo, d_thi, d_bounds, pad_abs);
    k_morton<<<blocks, 256, 0, stream>>>(d_tlo, d_thi, n, d_bounds, d_keys[0], d_vals[0]);
    for (int pass = 0; pass < 4; pass++) {
        const int shift = 8 * pass;
        k_hist<<<sort_blocks, kSortThreads, 0, stream>>>(d_keys[cur], n, shift, sort_blocks, d_hist);
        k_scan<<<1, 1024, 0, stream>>>(d_hist, 256u * sort_blocks);
        k_scatter<<<sort_blocks, kSortThreads, 0, stream>>>(d_keys[cur], d_vals[cur], n, shift, sort_blocks, d_hist,
                                                             d_keys[cur ^ 1], d_vals[cur ^ 1]);
        cur ^= 1;
    }
    k_gather_leaves<<<blocks, 256, 0, stream>>>(d_vals[cur], n, d_unsorted, out.tris, out.shade);
    if (n > 1 && mode >= 1) {
        // PLOC over the sorted order; the cluster arrays ping-pong, flags/positions reuse scratch
        Cluster* d_c[2]; uint32_t *d_nn, *d_keep, *d_made;
        HIPCK(sc.alloc(&d_c[0], (size_t)n * sizeof(Cluster)));
        HIPCK(sc.alloc(&d_c[1], (size_t)n * sizeof(Cluster)));
        HIPCK(sc.alloc(&d_nn, (size_t)n * 4));
        HIPCK(sc.alloc(&d_keep, ((size_t)n + 1) * 4));
        HIPCK(sc.alloc(&d_made, ((size_t)n + 1) * 4));
        k_ploc_init<<<blocks, 256, 0, stream>>>(d_vals[cur], n, d_tlo, d_thi, d_c[0]);
        PlocState* d_st;
        HIPCK(sc.alloc(&d_st, sizeof(PlocState)));
        PlocState h_st = {n, n - 2u, 0u, 0u, 0u, 0u, 0u, 0u};      // node numbers n-2 ... 0, root last = 0
        HIPCK(hipMemcpyAsync(d_st, &h_st, sizeof(h_st), hipMemcpyHostToDevice, stream));
        int pc = 0;
        uint32_t m_ub = n;              // what the host knows the cluster count does not exceed: sizes the grids
        const int kBatch = 8;           // rounds enqueued between two looks at the device's count
        for (int guard = 0; m_ub > 1u; guard++) {
            if (guard > 4096) { err = "PLOC made no progress"; return false; }
            const uint32_t mb = (m_ub + 255) / 256;
            for (int r = 0; r < kBatch; r++) {
                k_ploc_nn<<<mb, 256, 0, stream>>>(d_c[pc], d_st, d_nn);
                k_ploc_flags<<<mb, 256, 0, stream>>>(d_nn, d_st, d_keep, d_made);
                k_ploc_scan<<<1, 1024, 0, stream>>>(d_keep, d_made, d_st);
                k_ploc_merge<<<mb, 256, 0, stream>>>(d_c[pc], d_nn, d_st, d_keep, d_made, d_c[pc ^ 1], out.nodes);
                k_ploc_advance<<<1, 1, 0, stream>>>(d_st);
                pc ^= 1;
            }
            HIPCK(hipMemcpyAsync(&h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, stream));
            HIPCK(hipStreamSynchronize(stream));
            if (h_st.m >= m_ub && h_st.m > 1u) { err = "PLOC made no progress"; return false; }
            if (h_st.hashed_ties == 0u && m_ub > 1024u && h_st.m > m_ub - m_ub / 4u) {      // eight rounds took less than a quarter off (ordinary scenes: a third per ROUND): ties are holding the merges up (k_ploc_nn)
                const uint32_t one = 1u;
                HIPCK(hipMemcpyAsync(&d_st->hashed_ties, &one, 4, hipMemcpyHostToDevice, stream));
                if (getenv("ACGPT_DEBUG_BUILD")) fprintf(stderr, "[acgpt build] PLOC: %u -> %u clusters in %d rounds, ties go to hashed pairs from here\n", m_ub, h_st.m, kBatch);
            }
            m_ub = h_st.m;
        }
        out.build_iterations = h_st.rounds;
        // the last cluster carries the root's height
        HIPCK(hipMemcpyAsync(d_nhi, &d_c[pc][0].hi, 16, hipMemcpyDeviceToDevice, stream));
    } else if (n > 1) {
        k_hierarchy<<<(n - 1 + 255) / 256, 256, 0, stream>>>(d_keys[cur], (int)n, d_children, d_nparent, d_lparent);
        k_refit<<<blocks, 256, 0, stream>>>((int)n, d_vals[cur], d_tlo, d_thi, d_children, d_nparent, d_lparent, d_visit,
                                            d_nlo, d_nhi, out.nodes);
    } else {
        k_single_node<<<1, 1, 0, stream>>>(d_tlo, d_thi, out.nodes, d_nhi);
    }
    HIPCK(hipGetLastError());
    uint32_t opt_height = 0;
    const uint32_t opt_max_host = kOptimizeMaxTris;
    if (mode == 2 && n > 2 && n <= opt_max_host) {        // small scenes: insertion-based optimisation of the PLOC tree (12), on the host
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<BvhNode> h_nodes(n_nodes);
        HIPCK(hipMemcpyAsync(h_nodes.data(), out.nodes, (size_t)n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost, stream));
        HIPCK(hipStreamSynchronize(stream));
        std::vector<BvhNode> keep = h_nodes;
        double a0 = 0.0, a1 = 0.0; int passes = 0;
        opt_height = optimize_tree_host(h_nodes, n, &a0, &a1, &passes);
        // a tree that got deeper must still leave room for five workgroups' lane stacks in a CU's LDS (28 entries per lane: height <= 27);
        // otherwise the render falls to the windowed-stack kernel and loses more than the better tree gains
        HIPCK(hipMemcpy(&root_hi, d_nhi, 16, hipMemcpyDeviceToHost));
        if (opt_height > 27u && opt_height > (uint32_t)root_hi.w) opt_height = 0u;
        if (opt_height != 0u) {
            HIPCK(hipMemcpyAsync(out.nodes, h_nodes.data(), (size_t)n_nodes * sizeof(BvhNode), hipMemcpyHostToDevice, stream));
            HIPCK(hipStreamSynchronize(stream));
            out.opt_area_before = (float)a0; out.opt_area_after = (float)a1; out.opt_passes = (uint32_t)passes;
        }
        out.opt_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (getenv("ACGPT_DEBUG_BUILD"))
            fprintf(stderr, "[acgpt build] insertion-based optimisation: %d passes, inner-node area sum %.6g -> %.6g (%.1f %%), height %u, %.1f ms on the host\n",
                    passes, a0, a1, 100.0 * a1 / (a0 > 0.0 ? a0 : 1.0), opt_height, out.opt_ms);
    }
    if (mode == 2 && n > opt_max_host && n_nodes > 2) {       // larger scenes: parallel reinsertion on the device (13)
        ReinsTree t;
        t.m = n_nodes; t.n = n;
        const size_t N = (size_t)n_nodes + n;
        int *d_target, *d_pivot; float* d_gain; unsigned long long *d_lock, *d_lock_w; uint32_t *d_win, *d_nwin, *d_rvisit; double* d_area;
        HIPCK(sc.alloc(&t.parent, N * 4));
        HIPCK(sc.alloc(&t.child, (size_t)n_nodes * 8));
        HIPCK(sc.alloc(&t.lo, N * 16));
        HIPCK(sc.alloc(&t.hi, N * 16));
        HIPCK(sc.alloc(&d_target, N * 4));
        HIPCK(sc.alloc(&d_pivot, N * 4));
        HIPCK(sc.alloc(&d_gain, N * 4));
        HIPCK(sc.alloc(&d_lock, N * 8));
        HIPCK(sc.alloc(&d_lock_w, N * 8));
        HIPCK(sc.alloc(&d_win, N * 4));
        HIPCK(sc.alloc(&d_nwin, 4));
        HIPCK(sc.alloc(&d_rvisit, (size_t)n_nodes * 4));
        HIPCK(sc.alloc(&d_area, 8));
        const uint32_t nbm = (n_nodes + 255) / 256, nbn = (uint32_t)((N + 255) / 256);
        const auto t0 = std::chrono::steady_clock::now();
        k_ri_init<<<nbm, 256, 0, stream>>>(out.nodes, t);
        const auto area_now = [&](double& a) -> hipError_t {
            hipError_t e = hipMemsetAsync(d_area, 0, 8, stream);
            if (e != hipSuccess) return e;
            k_ri_area<<<nbm, 256, 0, stream>>>(t, d_area);
            e = hipMemcpyAsync(&a, d_area, 8, hipMemcpyDeviceToHost, stream);
            return e == hipSuccess ? hipStreamSynchronize(stream) : e;
        };
        // (the first refit also gives every inner node its box: k_ri_init only knows the children's)
        HIPCK(hipMemsetAsync(d_rvisit, 0, (size_t)n_nodes * 4, stream));
        k_ri_refit<<<blocks, 256, 0, stream>>>(t, d_rvisit);
        double a_first = 0.0, a_prev = 0.0, a_now = 0.0;
        HIPCK(area_now(a_first));
        a_prev = a_first;
        uint32_t iters = 0, moved = 0;
        // 12 rounds / stop below 0.2 % gain per round, from profiles/r04_ab_tree_optimisation.txt: 1.31 M triangles are flat after 4 rounds
        // (100.7 ms a frame against 101.2 after 16), 10.5 M still gain at 16 (131.5 against 133.7 after 8) at 55 ms a round; the two
        // environment variables are for that sweep, like ACGPT_OPT_PASSES above
        const uint32_t max_it = getenv("ACGPT_RI_ITERS") ? (uint32_t)atoi(getenv("ACGPT_RI_ITERS")) : kReinsertIterations;
        const double stop_at = getenv("ACGPT_RI_STOP") ? atof(getenv("ACGPT_RI_STOP")) : 0.998;
        for (; iters < max_it; iters++) {
            uint32_t winners = 0;
            k_ri_find<<<nbn, 256, 0, stream>>>(t, d_target, d_pivot, d_gain);
            HIPCK(hipMemsetAsync(d_lock, 0, N * 8, stream));
            HIPCK(hipMemsetAsync(d_lock_w, 0, N * 8, stream));
            HIPCK(hipMemsetAsync(d_nwin, 0, 4, stream));
            k_ri_lock<<<nbn, 256, 0, stream>>>(t, d_target, d_pivot, d_gain, d_lock, d_lock_w);
            k_ri_check<<<nbn, 256, 0, stream>>>(t, d_target, d_pivot, d_gain, d_lock, d_lock_w, d_win, d_nwin);
            k_ri_apply<<<nbn, 256, 0, stream>>>(t, d_target, d_win);
            HIPCK(hipMemsetAsync(d_rvisit, 0, (size_t)n_nodes * 4, stream));
            k_ri_refit<<<blocks, 256, 0, stream>>>(t, d_rvisit);
            HIPCK(hipGetLastError());
            HIPCK(hipMemcpyAsync(&winners, d_nwin, 4, hipMemcpyDeviceToHost, stream));
            HIPCK(area_now(a_now));
            moved += winners;
            if (getenv("ACGPT_DEBUG_BUILD"))
                fprintf(stderr, "[acgpt build] parallel reinsertion, iteration %u: %u moves, inner-node area sum %.6g (%.2f %% of the PLOC tree's)\n", iters + 1, winners, a_now, 100.0 * a_now / (a_first > 0.0 ? a_first : 1.0));
            if (winners == 0u || !(a_now < a_prev * stop_at)) { iters++; break; }
            a_prev = a_now;
        }
        float4 ploc_hi;
        HIPCK(hipMemcpyAsync(&root_hi, t.hi, 16, hipMemcpyDeviceToHost, stream));      // the root's height after the last refit
        HIPCK(hipMemcpyAsync(&ploc_hi, d_nhi, 16, hipMemcpyDeviceToHost, stream));     // and the PLOC tree's
        HIPCK(hipStreamSynchronize(stream));
        // the lane stacks hold 128 entries at most (capi.hip size_stack): a reinserted tree that outgrows them where the PLOC tree
        // did not is not taken (never seen: the trees get shallower, 35 -> 34 and 40 -> 38 levels on the 1.31 M / 10.5 M-triangle scenes)
        const bool too_deep = (uint32_t)root_hi.w > 120u && (uint32_t)root_hi.w > (uint32_t)ploc_hi.w;
        if (!too_deep) {
            k_ri_write<<<nbm, 256, 0, stream>>>(t, out.nodes);
            HIPCK(hipGetLastError());
            HIPCK(hipStreamSynchronize(stream));
            opt_height = (uint32_t)root_hi.w;
        }
        out.opt_area_before = (float)a_first; out.opt_area_after = too_deep ? (float)a_first : (float)a_now; out.opt_passes = iters;
        out.opt_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (getenv("ACGPT_DEBUG_BUILD"))
            fprintf(stderr, "[acgpt build] parallel reinsertion: %u iterations, %u moves, height %u, %.1f ms\n", iters, moved, opt_height, out.opt_ms);
    }
    if (n > kDepthFirstTris) {         // large scenes: the nodes in depth-first order (6a)
        int* d_par; uint32_t *d_size, *d_new; BvhNode* d_sorted = nullptr;
        HIPCK(sc.alloc(&d_par, (size_t)n_nodes * 4));
        HIPCK(sc.alloc(&d_size, (size_t)n_nodes * 4));
        HIPCK(sc.alloc(&d_new, (size_t)n_nodes * 4));
        HIPCK(hipMemsetAsync(d_visit, 0, (size_t)n_nodes * 4, stream));
        const uint32_t nbk = (n_nodes + 255) / 256;
        k_dfs_parents<<<nbk, 256, 0, stream>>>(out.nodes, n_nodes, d_par);
        k_dfs_sizes<<<nbk, 256, 0, stream>>>(out.nodes, n_nodes, d_par, d_visit, d_size);
        k_dfs_index<<<nbk, 256, 0, stream>>>(out.nodes, n_nodes, d_par, d_size, d_new);
        HIPCK(hipMalloc((void**)&d_sorted, (size_t)n_nodes * sizeof(BvhNode)));
        k_dfs_permute<<<nbk, 256, 0, stream>>>(out.nodes, n_nodes, d_new, d_sorted);
        hipError_t e_ = hipGetLastError();
        if (e_ == hipSuccess) e_ = hipStreamSynchronize(stream);
        if (e_ != hipSuccess) { (void)hipFree(d_sorted); err = std::string("depth-first numbering: ") + hipGetErrorString(e_); return false; }
        (void)hipFree(out.nodes);
        out.nodes = d_sorted;
    }
    HIPCK(hipEventRecord(sc.ev1, stream));
    HIPCK(hipMemcpyAsync(h_bounds, d_bounds, 24, hipMemcpyDeviceToHost, stream));
    HIPCK(hipMemcpyAsync(&root_hi, d_nhi, 16, hipMemcpyDeviceToHost, stream));
    HIPCK(hipStreamSynchronize(stream));
    HIPCK(hipEventElapsedTime(&out.build_ms, sc.ev0, sc.ev1));
    for (int k = 0; k < 3; k++) { out.scene_lo[k] = ord2f(h_bounds[k]); out.scene_hi[k] = ord2f(h_bounds[3 + k]); }
    {   // fp16 nodes: centre of the scene box, scaled so that the farthest plane sits at 1023
        float half_ext = 0.0f;
        HSpace sp;
        float* cc = &sp.cx;
        for (int k = 0; k < 3; k++) { cc[k] = 0.5f * out.scene_lo[k] + 0.5f * out.scene_hi[k]; half_ext = fmaxf(half_ext, fmaxf(out.scene_hi[k] - cc[k], cc[k] - out.scene_lo[k])); }
        // fp16 resolves 2^-11 of a value's own binade, so the rim of the scene should sit just BELOW a power of two: the farthest
        // plane goes to 1023 (a power-of-two scale leaves it anywhere in [512, 1024): Cornell's 278 became 556, where a step is
        // 0.5 — twice as coarse, 0.25 world units against 0.136 now).  Neither pack_planes() nor the kernels need the scale to be a
        // power of two: its reciprocal's rounding and the product's (2^-23 of a coordinate) sit inside pack_planes' 2^-18 guard
        // (test_gpu_fp16_slab_is_conservative runs both scales; profiles/r03_ab_hspace_scale.txt: -0.6 % / -0.3 % / 0 on configs 2 / 3 / 5).
        sp.inv_scale = (half_ext > 0.0f && half_ext < INFINITY ? half_ext : 1.0f) / 1023.0f;
        {   // the {centre, half extent} nodes: a scale per axis, every face of the scene box at |g| = 1023 (pt_device.h pack_centre_half).  An axis
            // the scene is (almost) flat on keeps at least 2^-20 of the longest extent; ACGPT_HC_UNIFORM=1: one scale for all (the A/B,
            // profiles/r04_ab_axis_scales.txt)
            float* is = &sp.isx;
            for (int k = 0; k < 3; k++) {
                const float hk = fmaxf(out.scene_hi[k] - cc[k], cc[k] - out.scene_lo[k]);
                is[k] = (hk > half_ext * 0x1p-20f && hk < INFINITY && !getenv("ACGPT_HC_UNIFORM") ? hk : (half_ext > 0.0f && half_ext < INFINITY ? half_ext : 1.0f)) / 1023.0f;
            }
            sp.pad_ = 0.0f;
        }
        float* d_area;
        float h_area[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        HIPCK(sc.alloc(&d_area, 16));
        HIPCK(hipMemsetAsync(d_area, 0, 16, stream));
        k_half_nodes<<<(n_nodes + 255) / 256, 256, 0, stream>>>(out.nodes, n_nodes, sp, out.hnodes, d_area);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(h_area, d_area, 16, hipMemcpyDeviceToHost, stream));
        HIPCK(hipStreamSynchronize(stream));
        HIPCK(hipMalloc((void**)&out.hcnodes, (size_t)n_nodes * sizeof(HNode)));
        k_hc_nodes<<<(n_nodes + 255) / 256, 256, 0, stream>>>(out.nodes, n_nodes, sp, out.hcnodes);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(stream));
        out.hspace = sp;
        out.half_area_ratio = h_area[0] > 0.0f ? h_area[1] / h_area[0] : 1.0f;
        out.half_box_inflation = h_area[2] > 0.0f ? h_area[3] / h_area[2] : 1.0f;
    }
    out.n_nodes = n_nodes;
    out.max_depth = opt_height ? opt_height : (uint32_t)root_hi.w;
    out.grid = make_qgrid_f(out.scene_lo, out.scene_hi);
    return true;
}

__global__ void k_tag_shade(float4* __restrict__ shade, uint32_t n, const DevMaterial* __restrict__ mats)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = __float_as_uint(shade[i].w) & kShadeMatMask;
    const float4 m1 = mats[id].ke_bsdf;
    const bool has_ke = !(m1.x == 0.0f && m1.y == 0.0f && m1.z == 0.0f);          // a NaN component counts: the fetch it stands for would return it
    shade[i].w = __uint_as_float(id | ((__float_as_uint(m1.w) & 3u) << kShadeBsdfShift) | (has_ke ? kShadeHasKe : 0u));
}

bool tag_shade_records(LbvhResult& r, const DevMaterial* d_mats, hipStream_t stream, std::string& err)
{
    if (!r.shade || r.n_tris == 0 || !d_mats) return true;
    k_tag_shade<<<(r.n_tris + 255u) / 256u, 256, 0, stream>>>(r.shade, r.n_tris, d_mats);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { err = std::string("shade records: ") + hipGetErrorString(e); return false; }
    return true;
}

// The breadth-first copy of the tree's top (k_top_nodes): built on first use by a kernel variant that stages it in LDS — an
// experiment; not part of the default scene set-up.
bool build_top_nodes(LbvhResult& r, hipStream_t stream, std::string& err)
{
    if (r.top_nodes || !r.hnodes || r.n_tris == 0) return true;
    uint32_t* d_ntop = nullptr;
    HIPCK(hipMalloc((void**)&d_ntop, 4));
    hipError_t e = hipMalloc((void**)&r.top_nodes, (size_t)kTopNodesMax * (sizeof(HNode) + sizeof(uint32_t)));      // nodes, then their indices in hnodes
    if (e == hipSuccess) {
        k_top_nodes<<<1, 64, 0, stream>>>(r.hnodes, r.top_nodes, (uint32_t*)(r.top_nodes + kTopNodesMax), kTopNodesMax, d_ntop);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&r.n_top, d_ntop, 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_ntop);
    if (e != hipSuccess) { err = std::string("top nodes: ") + hipGetErrorString(e); return false; }
    return true;
}

// ---- node arrays on demand ------------------------------------------------------------------------------------------------
static bool sync_ok(hipStream_t stream, const char* what, std::string& err)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { err = std::string(what) + ": " + hipGetErrorString(e); return false; }
    return true;
}

void keep_one_node_array(LbvhResult& r, int keep)
{
    // keep: 0 the fp32 nodes, 1 the fp16 {lo, hi} nodes, 2 the fp16 {centre, half extent} nodes; never the last copy of the topology
    const bool have = keep == 0 ? r.nodes != nullptr : (keep == 1 ? r.hnodes != nullptr : r.hcnodes != nullptr);
    if (!have) return;
    if (keep != 0) {
        if (r.nodes) { (void)hipFree(r.nodes); r.nodes = nullptr; }
        if (r.qnodes) { (void)hipFree(r.qnodes); r.qnodes = nullptr; }
        if (r.cnodes) { (void)hipFree(r.cnodes); r.cnodes = nullptr; }
    }
    if (keep != 1) {
        if (r.hnodes) { (void)hipFree(r.hnodes); r.hnodes = nullptr; }
        if (r.top_nodes) { (void)hipFree(r.top_nodes); r.top_nodes = nullptr; r.n_top = 0; }
    }
    if (keep != 2 && r.hcnodes) { (void)hipFree(r.hcnodes); r.hcnodes = nullptr; }
}

bool ensure_nodes(LbvhResult& r, hipStream_t stream, std::string& err)
{
    if (r.nodes || r.n_tris == 0) return true;
    const HNode* topo = r.hnodes ? r.hnodes : r.hcnodes;       // either fp16 array holds the topology
    const int shift = r.hnodes ? 0 : 5;
    if (!topo || !r.tris) { err = "fp32 nodes: no node array is present"; return false; }
    Scratch sc;
    const uint32_t n = r.n_tris, n_nodes = r.n_nodes;
    HIPCK(hipMalloc((void**)&r.nodes, (size_t)n_nodes * sizeof(BvhNode)));
    if (n == 1) {
        k_single_node_record<<<1, 1, 0, stream>>>(r.tris, r.pad_abs, r.nodes);
    } else {
        int *d_np, *d_lp; uint32_t* d_visit; float4 *d_lo, *d_hi;
        HIPCK(sc.alloc(&d_np, (size_t)n_nodes * 4));
        HIPCK(sc.alloc(&d_lp, (size_t)n * 4));
        HIPCK(sc.alloc(&d_visit, (size_t)n_nodes * 4));
        HIPCK(sc.alloc(&d_lo, (size_t)n_nodes * 16));
        HIPCK(sc.alloc(&d_hi, (size_t)n_nodes * 16));
        HIPCK(hipMemsetAsync(d_visit, 0, (size_t)n_nodes * 4, stream));
        k_parents_of<<<(n_nodes + 255) / 256, 256, 0, stream>>>(topo, shift, n_nodes, d_np, d_lp);
        k_refit_records<<<(n + 255) / 256, 256, 0, stream>>>((int)n, r.tris, r.pad_abs, topo, shift, d_np, d_lp, d_visit, d_lo, d_hi, r.nodes);
    }
    return sync_ok(stream, "fp32 nodes", err);
}

bool ensure_hnodes(LbvhResult& r, hipStream_t stream, std::string& err)
{
    if (r.hnodes || r.n_tris == 0) return true;
    if (!ensure_nodes(r, stream, err)) return false;
    Scratch sc;
    float* d_area;
    HIPCK(sc.alloc(&d_area, 16));
    HIPCK(hipMemsetAsync(d_area, 0, 16, stream));
    HIPCK(hipMalloc((void**)&r.hnodes, (size_t)r.n_nodes * sizeof(HNode)));
    k_half_nodes<<<(r.n_nodes + 255) / 256, 256, 0, stream>>>(r.nodes, r.n_nodes, r.hspace, r.hnodes, d_area);
    return sync_ok(stream, "fp16 nodes", err);
}

// experiment formats, derived from the fp32 nodes on first use by a kernel variant that reads them
bool ensure_qnodes(LbvhResult& r, hipStream_t stream, std::string& err)
{
    if (r.qnodes || r.n_tris == 0) return true;
    if (!ensure_nodes(r, stream, err)) return false;
    HIPCK(hipMalloc((void**)&r.qnodes, (size_t)r.n_nodes * sizeof(QNode)));
    k_quant_nodes<<<(r.n_nodes + 255) / 256, 256, 0, stream>>>(r.nodes, r.n_nodes, r.grid, r.qnodes);
    return sync_ok(stream, "grid nodes", err);
}

bool ensure_cnodes(LbvhResult& r, hipStream_t stream, std::string& err)
{
    if (r.cnodes || r.n_tris == 0) return true;
    if (!ensure_nodes(r, stream, err)) return false;
    HIPCK(hipMalloc((void**)&r.cnodes, (size_t)r.n_nodes * sizeof(BvhNode)));
    k_centre_nodes<<<(r.n_nodes + 255) / 256, 256, 0, stream>>>(r.nodes, r.n_nodes, r.cnodes);
    return sync_ok(stream, "centre nodes", err);
}

bool ensure_hcnodes(LbvhResult& r, hipStream_t stream, std::string& err)
{
    if (r.hcnodes || r.n_tris == 0) return true;
    if (!ensure_nodes(r, stream, err)) return false;
    HIPCK(hipMalloc((void**)&r.hcnodes, (size_t)r.n_nodes * sizeof(HNode)));
    k_hc_nodes<<<(r.n_nodes + 255) / 256, 256, 0, stream>>>(r.nodes, r.n_nodes, r.hspace, r.hcnodes);
    return sync_ok(stream, "fp16 centre / half-extent nodes", err);
}

#ifdef ACGPT_EXPERIMENTS
#include "lbvh_experiments.inc"      // builders of formats that were measured and not adopted; not part of the product's kernel-source hash
#endif

// (Morton code, original triangle index) of every leaf slot, as the sort saw them: recomputed from the records, HOST outputs
bool read_morton(const LbvhResult& r, hipStream_t stream, uint32_t* h_codes, uint32_t* h_prims, std::string& err)
{
    if (r.n_tris == 0) return true;
    Scratch sc;
    uint32_t *d_k, *d_p;
    HIPCK(sc.alloc(&d_k, (size_t)r.n_tris * 4));
    HIPCK(sc.alloc(&d_p, (size_t)r.n_tris * 4));
    Bounds6 b;
    for (int k = 0; k < 3; k++) { b.v[k] = r.scene_lo[k]; b.v[3 + k] = r.scene_hi[k]; }
    k_morton_of_records<<<(r.n_tris + 255) / 256, 256, 0, stream>>>(r.tris, r.n_tris, r.pad_abs, b, d_k, d_p);
    HIPCK(hipGetLastError());
    HIPCK(hipMemcpyAsync(h_codes, d_k, (size_t)r.n_tris * 4, hipMemcpyDeviceToHost, stream));
    HIPCK(hipMemcpyAsync(h_prims, d_p, (size_t)r.n_tris * 4, hipMemcpyDeviceToHost, stream));
    HIPCK(hipStreamSynchronize(stream));
    return true;
}

size_t scene_device_bytes(const LbvhResult& r)
{
    size_t b = 0;
    if (r.nodes) b += (size_t)r.n_nodes * sizeof(BvhNode);
    if (r.hnodes) b += (size_t)r.n_nodes * sizeof(HNode);
    if (r.qnodes) b += (size_t)r.n_nodes * sizeof(QNode);
    if (r.cnodes) b += (size_t)r.n_nodes * sizeof(BvhNode);
    if (r.top_nodes) b += (size_t)kTopNodesMax * (sizeof(HNode) + sizeof(uint32_t));
    if (r.tris) b += (size_t)r.n_tris * sizeof(TriRecord);
    if (r.shade) b += (size_t)r.n_tris * sizeof(float4);
    if (r.wrecs) b += (size_t)r.n_wrecs * 48u;
    if (r.srecs) b += (size_t)r.n_srecs * sizeof(uint4);
    if (r.hcnodes) b += (size_t)r.n_nodes * sizeof(HNode);
    b += r.hcnodes_alt_bytes;
    return b;
}

bool build_lbvh(const float* h_verts_xyzw, size_t n_verts, const uint32_t* h_idx, size_t n_tris,
                const uint32_t* h_mat_ids, int mode, hipStream_t stream, LbvhResult& out, std::string& err)
{
    out = LbvhResult();
    out.n_tris = (uint32_t)n_tris;
    out.mode = mode;
    if (n_tris == 0) return true;
    if (!build_impl(h_verts_xyzw, n_verts, h_idx, (uint32_t)n_tris, h_mat_ids, mode, stream, out, err)) {
        (void)hipStreamSynchronize(stream);
        free_lbvh(out);
        return false;
    }
    return true;
}

}  // namespace ptd
