// pt_device.h — device-side data layout and the per-lane building blocks of the gfx950
// path-trace kernels: float3 algebra, PRNG, ray/triangle test, LDS-stack BVH traversal.
//
// HBM layout (all arrays owned by the context, built once per scene):
//   nodes  : BvhNode[n_nodes], 64 B each, two child boxes + two child references.
//            child >= 0 : internal node index;  child < 0 : leaf, ~child = slot in `tris`.
//   tris   : TriRecord[n_tris], 48 B each, in Morton order: v0, e1 = v1-v0, e2 = v2-v0,
//            original triangle index, material id.  Edges are the single fp32 subtraction
//            the reference's closest-hit performs (pathTracerPrograms.cu:890), so shading
//            needs no second fetch through the index buffer.
//   shade  : float4[n_tris], same order: the triangle's geometric normal — normalize(cross(e1, e2)), the very operations of
//            pathTracerPrograms.cu:890, done once by the builder — and its material id: one 16-byte fetch per shaded hit
//            instead of the record's three, and no normalisation (a sqrt and a division) in the shade phase.  The id's upper
//            byte carries the material's bsdfType and whether its emission is non-zero (tag_shade_records), so a diffuse,
//            non-emissive hit — nearly all of them — needs ONE further fetch: {diffuse, ior}.
//   mats   : DevMaterial[n_mats]: pt_material (40 B, HitGroupData's payload, pathTracer.h:118-127) repacked at upload into two
//            aligned 16-byte halves, {diffuse, ior} and {emission, bsdfType}; roughness and metallic, which the reference's
//            shading ignores (:879-880), are not carried.
// Compile with -ffp-contract=off: the only fused multiply-adds are the explicit ones — in tri_test(), which must match
// oracle/oracle_pt.cpp bit for bit, in the slab tests, and in the shading helpers at arithmetic level 2 (m_dot, m_cross, m_madd below).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/acgpt.h"

namespace ptd {

struct __attribute__((aligned(16))) BvhNode {
    float4 a;   // lo0.x lo0.y lo0.z hi0.x
    float4 b;   // hi0.y hi0.z lo1.x lo1.y
    float4 c;   // lo1.z hi1.x hi1.y hi1.z
    int4   d;   // child0 child1 - -
};
static_assert(sizeof(BvhNode) == 64, "node size");

struct __attribute__((aligned(16))) TriRecord {
    float4 r0;  // v0.x v0.y v0.z e1.x
    float4 r1;  // e1.y e1.z e2.x e2.y
    float4 r2;  // e2.z prim(bits) mat(bits) -
};
static_assert(sizeof(TriRecord) == 48, "tri size");

// Quantised node, 32 B = two 16-byte loads: each child box as six 16-bit grid coordinates
// (rounded outward by one cell, so the box test stays conservative) + the child reference.
//   a = { lo0.x | lo0.y<<16, lo0.z | hi0.x<<16, hi0.y | hi0.z<<16, child0 }
//   b = { lo1.x | lo1.y<<16, lo1.z | hi1.x<<16, hi1.y | hi1.z<<16, child1 }
// Rays are moved into grid space once (QGrid), so decoding is an integer->float convert.
struct __attribute__((aligned(16))) QNode {
    uint4 a;
    uint4 b;
};
static_assert(sizeof(QNode) == 32, "qnode size");

// Half-precision node, 32 B = two 16-byte loads: each child box as six fp16 planes in a scene-centred, scaled
// space g = (w - centre) * scale (the scene's farthest plane at 1023, where fp16 is finest relative to the scene), lo rounded down and hi rounded up so the fp16 box contains the fp32 one.
//   a = { lo0.x | hi0.x << 16, lo0.y | hi0.y << 16, lo0.z | hi0.z << 16, child0 }
//   b = { lo1.x | hi1.x << 16, lo1.y | hi1.y << 16, lo1.z | hi1.z << 16, child1 }
// A slab plane is t = g * (1/d / scale) + (centre - o)/d: one v_fma_mix_f32 per plane, which reads the fp16 half of the
// register directly — the decode costs no instruction, the node half the loads of the fp32 format.
struct __attribute__((aligned(16))) HNode {
    uint4 a;
    uint4 b;
};
static_assert(sizeof(HNode) == 32, "hnode size");
struct HSpace { float cx, cy, cz, inv_scale;        // world = g * inv_scale + centre; the builder puts the scene's farthest plane at g = 1023
                float isx, isy, isz, pad_; };       // the {centre, half extent} nodes (NODE_FMT 11) have a scale per axis: every face of the scene box at |g| = 1023 (below)

// Shared-plane node (NODE_FMT 10, round 4), 16 B = ONE 16-byte load per visit.  A child box is its parent's box cut by new planes, and
// because the parent's box is the union of its two children, each of the parent's six planes is inherited by at least one child: the two
// children together have at most SIX planes the parent does not.  A node stores exactly those — per axis the new plane on the lo side
// and the new plane on the hi side, each owned by ONE child — and the traversal carries the ray's interval [tn, tf] in the node's own
// box (which accounts for every inherited plane, bit for bit: same fma, same constants) down to the children and on the stack.
//   record = { x: lo_new | hi_new << 16,  y: ...,  z: ...,  children }
// A plane is an fp16 MAGNITUDE measured inward from the root's plane of its own side — lo planes from the root's lo corner, hi planes
// from its hi corner, both >= 0 and rounded toward 0, i.e. outward —; the SIGN bit says which child owns it (0: child 0, 1: child 1).
// Child 0 evaluates the six halves as stored, child 1 negated (a free source modifier of v_fma_mix_f32): a plane owned by the other
// child then has a NEGATIVE magnitude, i.e. lies beyond the root's plane of its side, where it can neither raise tn nor lower tf — it
// drops out without a select, a mask or a NaN.  12 v_fma_mix_f32 and 3 rotates per visit (the 32-byte HNode: 12 and 6), one gather.
// children: record index of child 0; child 1 follows it (a triangle is three records: the TriRecord itself); bit 30 / 31: child 0 / 1
// is a triangle.  Inner nodes and triangles share the one array, every pair of siblings is contiguous.
struct SSpace { float lx, ly, lz, hx, hy, hz, inv_scale; };      // the root's planes (world) and world units per unit of magnitude
constexpr uint32_t kSLeaf0 = 1u << 31, kSLeaf1 = 1u << 30, kSBaseMask = (1u << 30) - 1u;
constexpr uint32_t kTopNodeFlag = 0x40000000u;      // node reference into the breadth-first copy of the tree's top (DeviceScene::top)
constexpr uint32_t kTopNodesMax = 255u;

// world -> grid: g = (w - origin) * inv_cell ; cell sizes per axis
struct QGrid {
    float ox, oy, oz;
    float icx, icy, icz;
    float cx, cy, cz;
};

// pt_material as the kernels read it (capi.hip repacks at upload): two aligned 16-byte loads instead of 40 unaligned bytes
struct DevMaterial { float4 kd_ior; float4 ke_bsdf; };      // diffuse.xyz + ior | emission.xyz + bsdfType (as bits)
// shade record .w: material id in the low 24 bits, bsdfType above, and a flag: the material's emission has a non-zero component
constexpr uint32_t kShadeMatMask = 0x00FFFFFFu, kShadeBsdfShift = 24u, kShadeHasKe = 1u << 26;

struct DeviceScene {
    const BvhNode*     nodes;
    const QNode*       qnodes;
    const BvhNode*     cnodes;    // the same tree with every child box as centre + half extent (layout of BvhNode: lo -> centre, hi -> half extent)
    const HNode*       hnodes;    // the same tree with fp16 boxes (32-byte nodes)
    const HNode*       top;       // its first n_top inner nodes, breadth first (lbvh_build.hip k_top_nodes)
    uint32_t           n_top;
    HSpace             hspace;
    QGrid              grid;
    const TriRecord*   tris;
    const float4*      shade;     // per leaf slot: geometric normal normalize(cross(e1, e2)) (:890) and material id — what closest-hit shading reads
    const uint4*       wrecs;     // four-wide tree: 48-byte records, wide nodes and triangles in one array (wide_bvh.hip)
    const HNode*       hcnodes;   // the fp16 tree with every child box as centre | half extent per axis (NODE_FMT 11), child references as byte offsets
    const uint4*       srecs;     // shared-plane tree (NODE_FMT 10): 16-byte records, nodes and triangles in one array; record 0 = the root
    SSpace             sspace;
    const DevMaterial* mats;
    uint32_t n_tris;
    uint32_t n_mats;
    // light mode 1 (pt_set_light_mode): the scene's emissive triangles, 5 float4 each:
    //   {v0.xyz, area} {e1.xyz, cdf} {e2.xyz, -} {n.xyz, -} {Ke.xyz, -}; cdf = running sum of the areas, light_area = its last value
    const float4* lights;
    uint32_t n_lights;
    float light_area;
};

constexpr int   kSentinel  = 0x7FFFFFFF;   // stack bottom marker (never a valid node index)
constexpr float kFarWiden  = 1.000001f;    // conservative slab test (Ize 2013) incl. the 1-ulp v_rcp_f32 direction
// Pruning against the best hit so far uses best_t * kTieWiden: a triangle whose computed distance ties with
// (or is a few ulps below) the current best must still be reached, and its box entry distance and its
// Moeller-Trumbore distance round independently.  The triangle test itself is exact about (tmin, tmax).
constexpr float kTieWiden  = 1.00002f;
constexpr float kPIf       = 3.14159265358979323846f;

// ------------------------------------------------------------------ float3 ----
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk(float x, float y, float z) { f3 r = {x, y, z}; return r; }
__device__ __forceinline__ f3 mk(float s) { return mk(s, s, s); }
__device__ __forceinline__ f3 mk(const pt_float3& p) { return mk(p.x, p.y, p.z); }
__device__ __forceinline__ f3 operator-(const f3& a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator+(const f3& a, const f3& b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(const f3& a, const f3& b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(const f3& a, const f3& b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(const f3& a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, const f3& a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator/(const f3& a, const f3& b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ f3 operator/(const f3& a, float s) { float inv = 1.0f / s; return a * inv; }
__device__ __forceinline__ void operator+=(f3& a, const f3& b) { a.x += b.x; a.y += b.y; a.z += b.z; }
__device__ __forceinline__ void operator*=(f3& a, const f3& b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; }
__device__ __forceinline__ float dot(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(const f3& a, const f3& b)
{ return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float length(const f3& v) { return sqrtf(dot(v, v)); }
__device__ __forceinline__ f3 normalize(const f3& v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }
// Arithmetic level of the shading code (template parameter FM of the functions below and of pt_shading.h):
//   0  IEEE: correctly rounded division and square root, OCML sincosf / acosf — the level the CPU oracle is written at;
//   1  level 0 with the cosine sampler's trigonometry on v_sin_f32 / v_cos_f32 and sqrt(1 - z1) for sin(acos(sqrt(z1)));
//   2  what nvcc --use_fast_math makes of the reference's own build (CMakeLists.txt:267: -prec-div=false -prec-sqrt=false,
//      sinf -> __sinf, cosf -> __cosf, and nvcc's default -fmad=true): a / b = a * v_rcp_f32(b), v_sqrt_f32, v_rsq_f32,
//      v_sin_f32 / v_cos_f32 (1 ulp each), and multiply-adds fused where they are spelled out below (m_dot, m_cross, m_madd:
//      nvcc contracts where it sees fit; here the placement is explicit, so every kernel variant computes the same bits).
// Traversal and the triangle test are the same at every level (hits stay bit-exact); only shading values move in their last bits.
template <int FM> __device__ __forceinline__ float m_div(float a, float b) { return FM >= 2 ? a * __builtin_amdgcn_rcpf(b) : a / b; }
template <int FM> __device__ __forceinline__ float m_sqrt(float x) { return FM >= 2 ? __builtin_amdgcn_sqrtf(x) : sqrtf(x); }
template <int FM> __device__ __forceinline__ float m_dot(const f3& a, const f3& b)
{ return FM >= 2 ? __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)) : a.x * b.x + a.y * b.y + a.z * b.z; }
template <int FM> __device__ __forceinline__ f3 m_cross(const f3& a, const f3& b)
{
    if (FM >= 2) return mk(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)), __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// a * s + b
template <int FM> __device__ __forceinline__ f3 m_madd(const f3& a, float s, const f3& b)
{
    if (FM >= 2) return mk(__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z));
    return mk(a.x * s + b.x, a.y * s + b.y, a.z * s + b.z);
}
// a * b + c, component by component
template <int FM> __device__ __forceinline__ f3 m_madd(const f3& a, const f3& b, const f3& c)
{
    if (FM >= 2) return mk(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y), __builtin_fmaf(a.z, b.z, c.z));
    return mk(a.x * b.x + c.x, a.y * b.y + c.y, a.z * b.z + c.z);
}
template <int FM> __device__ __forceinline__ float m_length(const f3& v) { return m_sqrt<FM>(m_dot<FM>(v, v)); }
template <int FM> __device__ __forceinline__ f3 m_normalize(const f3& v)
{
    if (FM >= 2) return v * __builtin_amdgcn_rsqf(m_dot<FM>(v, v));
    const float invLen = 1.0f / sqrtf(dot(v, v));
    return v * invLen;
}
// sin and cos of 2 pi u, u in [0, 1): the hardware instructions take their argument in revolutions
template <int FM> __device__ __forceinline__ void m_sincos_2pi(float u, float& s, float& c)
{
    if (FM >= 2) { s = __builtin_amdgcn_sinf(u); c = __builtin_amdgcn_cosf(u); }
    else sincosf(2.0f * kPIf * u, &s, &c);
}
__device__ __forceinline__ f3 reflect(const f3& i, const f3& n) { return i - 2.0f * n * dot(n, i); }
__device__ __forceinline__ f3 faceforward(const f3& n, const f3& i, const f3& nref) { return n * copysignf(1.0f, dot(i, nref)); }
__device__ __forceinline__ f3 lerp3(const f3& a, const f3& b, float t) { return a + t * (b - a); }
__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

// -------------------------------------------------------------------- PRNG ----
// tea<4>, cuda/random.h:31-46
__device__ __forceinline__ uint32_t tea4(uint32_t val0, uint32_t val1)
{
    uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
    for (uint32_t n = 0; n < 4; n++) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    return v0;
}
// lcg + rnd, cuda/random.h:49-55, 64-67
__device__ __forceinline__ float rnd(uint32_t& prev)
{
    prev = 1664525u * prev + 1013904223u;
    return (float)(prev & 0x00FFFFFFu) / (float)0x01000000;
}

// ------------------------------------------------------ ray / triangle test ----
__device__ __forceinline__ float dot_fma(const f3& a, const f3& b)
{ return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross_fma(const f3& a, const f3& b)
{
    return mk(__builtin_fmaf(a.y, b.z, -(a.z * b.y)),
              __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
              __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
// Moeller-Trumbore, two-sided, open interval (tmin, tmax); same operations in the same
// order as tri_test() in oracle/oracle_pt.cpp.
__device__ __forceinline__ bool tri_test(const f3& o, const f3& d, const f3& v0, const f3& e1, const f3& e2,
                                         float tmin, float tmax, float& t_out)
{
    f3 p = cross_fma(d, e2);
    float det = dot_fma(e1, p);
    f3 s = o - v0;
    float U = dot_fma(s, p);
    f3 q = cross_fma(s, e1);
    float V = dot_fma(d, q);
    float T = dot_fma(e2, q);
    if (det < 0.0f) { det = -det; U = -U; V = -V; T = -T; }
    bool ok = (det > 0.0f) && !(U < 0.0f || V < 0.0f || U + V > det);
    float t = T / det;
    ok = ok && (t > tmin && t < tmax);
    t_out = t;
    return ok;
}

// Same test, with the division (the only expensive operation) issued only for lanes that passed
// the barycentric test; accepted hits are bit-identical to tri_test().
__device__ __forceinline__ bool tri_test_lazy(const f3& o, const f3& d, const f3& v0, const f3& e1, const f3& e2,
                                              float tmin, float tmax, float& t_out)
{
    f3 p = cross_fma(d, e2);
    float det = dot_fma(e1, p);
    f3 s = o - v0;
    float U = dot_fma(s, p);
    f3 q = cross_fma(s, e1);
    float V = dot_fma(d, q);
    // if (det < 0) negate det, U, V, T: as sign-bit arithmetic, without a branch (negating T afterwards gives the bits
    // of the dot product with q negated: rounding is symmetric).  For det = -0.0 this flips where the comparison would
    // not, but a zero determinant is rejected either way.
    const uint32_t sgn = __float_as_uint(det) & 0x80000000u;
    det = __uint_as_float(__float_as_uint(det) ^ sgn);
    U = __uint_as_float(__float_as_uint(U) ^ sgn);
    V = __uint_as_float(__float_as_uint(V) ^ sgn);
    bool ok = (det > 0.0f) && !(U < 0.0f || V < 0.0f || U + V > det);
    t_out = 0.0f;
    if (ok) {
        const float T = __uint_as_float(__float_as_uint(dot_fma(e2, q)) ^ sgn);
        const float t = T / det;
        ok = (t > tmin && t < tmax);
        t_out = t;
    }
    return ok;
}

// 1-ulp reciprocal (v_rcp_f32) for quantities outside the bit-exact contract (box tests).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// The same, kept finite: the fma form of the slab test, t = p * (1/d) - o/d, must not see inf - inf for a box that
// straddles a coordinate plane when d is 0 (the subtract-then-multiply form gets -inf / +inf there and is fine).  With
// +-1e30 a zero direction component still sends both planes to +-huge with the signs of (p - o), up to the rounding of
// o/d that lbvh_build.hip's pad_abs covers.
__device__ __forceinline__ float finite_rcp(float x) { return fminf(fmaxf(__builtin_amdgcn_rcpf(x), -1e30f), 1e30f); }

// fp16 plane (low / high half of a packed dword) times a, plus b, in fp32: compiles to one v_fma_mix_f32
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fma_h_lo(uint32_t packed, float a, float b) { return __builtin_fmaf((float)__builtin_bit_cast(half2_t, packed).x, a, b); }
__device__ __forceinline__ float fma_h_hi(uint32_t packed, float a, float b) { return __builtin_fmaf((float)__builtin_bit_cast(half2_t, packed).y, a, b); }

// ---- the slab test of the default render kernel on fp16 planes (NODE_FMT 9 of render_megakernel.hip) ----
// pack_planes: builder side.  Planes go to the scene-centred, scaled space of HSpace and are rounded OUTWARD.
__device__ __forceinline__ uint32_t half_bits(__half h) { return (uint32_t)__half_as_ushort(h); }
__device__ __forceinline__ uint32_t pack_planes(float lo, float hi, float c, float scale, float& glo, float& ghi)
{
    if (!(lo <= hi)) { glo = 0.0f; ghi = 0.0f; return 0x7C00u | (0xFC00u << 16); }      // empty child: lo = +inf, hi = -inf
    // (w - c) * scale rounds twice in fp32, scale is the rounded reciprocal of the kernels' inv_scale and their plane multiplier
    // (1/d) * inv_scale rounds once more (2^-22 of the coordinate together), and that multiplier carries the ray's
    // rotate flags in its five lowest mantissa bits (setup_ray, NODE_FMT 9: 2^-19 of the coordinate): each plane goes outward
    // by 2^-18 of its own coordinate before it is rounded outward to fp16 (whose step is 2^-11 of it)
    const float a = (lo - c) * scale, b = (hi - c) * scale;
    __half hl = __float2half_rd(a - fabsf(a) * 3.9e-6f), hh = __float2half_ru(b + fabsf(b) * 3.9e-6f);
    glo = __half2float(hl); ghi = __half2float(hh);
    return half_bits(hl) | (half_bits(hh) << 16);
}

// Per-ray constants: t = g * mul + add for an fp16 plane g, mul = (1/d) / scale, add = (centre - o) / d.  The five lowest
// mantissa bits of mul carry the rotate amount of its axis — 16 where the ray runs against the axis, else 0 —, which is all
// v_alignbit_b32 reads of its shift operand: the packed {lo, hi} pair of a node is rotated so that its low half is the plane
// the ray meets first, and near / far need no per-axis min / max and no register for the amounts.  Forcing those bits moves
// mul by at most 31 ulp (2^-19 relative), i.e. a plane by 2^-19 of its own coordinate; pack_planes() pads by 2^-18 of it.
__device__ __forceinline__ void setup_ray_h9(const f3& ro, const f3& rd, const HSpace& HS, f3& mul, f3& add)
{
    const f3 r = mk(finite_rcp(rd.x), finite_rcp(rd.y), finite_rcp(rd.z));
    add = mk((HS.cx - ro.x) * r.x, (HS.cy - ro.y) * r.y, (HS.cz - ro.z) * r.z);
    mul = r * HS.inv_scale;
    mul.x = __uint_as_float((__float_as_uint(mul.x) & ~31u) | (r.x < 0.0f ? 16u : 0u));
    mul.y = __uint_as_float((__float_as_uint(mul.y) & ~31u) | (r.y < 0.0f ? 16u : 0u));
    mul.z = __uint_as_float((__float_as_uint(mul.z) & ~31u) | (r.z < 0.0f ? 16u : 0u));
}
__device__ __forceinline__ uint32_t rot16(uint32_t v, uint32_t by) { return __builtin_amdgcn_alignbit(v, v, by); }
// entry and exit distance of one child box {px, py, pz} (packed {lo, hi} per axis): 3 rotates, 6 v_fma_mix_f32, max3 / min3
__device__ __forceinline__ void slab_h9(uint32_t px, uint32_t py, uint32_t pz, const f3& mul, const f3& add, float rtmin, float& tn, float& tf)
{
    const uint32_t ax = rot16(px, __float_as_uint(mul.x)), ay = rot16(py, __float_as_uint(mul.y)), az = rot16(pz, __float_as_uint(mul.z));
    tn = fmaxf(fmaxf(fma_h_lo(ax, mul.x, add.x), fma_h_lo(ay, mul.y, add.y)), fmaxf(fma_h_lo(az, mul.z, add.z), rtmin));
    tf = fminf(fminf(fma_h_hi(ax, mul.x, add.x), fma_h_hi(ay, mul.y, add.y)), fma_h_hi(az, mul.z, add.z)) * kFarWiden;
}

// ---- fp16 centre / half-extent nodes (NODE_FMT 11, round 4) ----
// The HNode layout with {centre, half extent} per axis instead of {lo, hi}: near = c_t - h_t, far = c_t + h_t with c_t = c * (1/d / scale) +
// (centre - o)/d and h_t = h * |1/d / scale|.  Four instructions per axis and child — two v_fma_mix_f32, one subtract, one add — against the
// rotated form's two v_fma_mix_f32 and half a v_alignbit_b32; but the subtract and the add are FULL-rate fp32 instructions, which the SIMD executes
// beside a half-rate neighbour almost for nothing (profiles/r04_ubench_valu.txt: v_max_f32 + v_fma_f32 as a pair 4.7 cycles, alone 4.1 + 2.6),
// while v_alignbit_b32 is one more half-rate instruction in a loop that is made of them.  No rotate flags in the multipliers either.
// builder side: c rounded to nearest, h rounded up so that [c - h, c + h] holds the fp32 interval plus the same 2^-18 relative guard as pack_planes.
// A scale PER AXIS (HSpace isx / isy / isz; the ray parameter t does not care how an axis is scaled, the loop is unchanged): each face of the
// scene box sits at |g| = 1023, an fp16 value, so a flat box on such a face — a wall of a closed room, and every inner box that touches it —
// has c exact and h = its pad: as thin as the fp32 box, where a centre between two fp16 values makes it up to one step (1 / 2046 of the
// extent) thick and every ray that leaves the wall stays inside it beyond tmin and tests the wall's triangles for nothing.
__device__ __forceinline__ uint32_t pack_centre_half(float lo, float hi, float c0, float scale)
{
    if (!(lo <= hi)) return half_bits(__float2half_rn(0.0f)) | (half_bits(__float2half_rn(-1.0f)) << 16);      // empty child: a negative half extent is never hit
    const float a = (lo - c0) * scale, b = (hi - c0) * scale;
    const __half hc = __float2half_rn(0.5f * a + 0.5f * b);
    const float c = __half2float(hc);
    const float guard = fmaxf(fabsf(a), fabsf(b)) * 3.9e-6f;
    const float h = fmaxf(b - c, c - a) + guard;
    return half_bits(hc) | (half_bits(__float2half_ru(h + fabsf(h) * 1e-6f)) << 16);
}
__device__ __forceinline__ void setup_ray_hc(const f3& ro, const f3& rd, const HSpace& HS, f3& mul, f3& add)
{
    const f3 r = mk(finite_rcp(rd.x), finite_rcp(rd.y), finite_rcp(rd.z));
    add = mk((HS.cx - ro.x) * r.x, (HS.cy - ro.y) * r.y, (HS.cz - ro.z) * r.z);
    mul = mk(r.x * HS.isx, r.y * HS.isy, r.z * HS.isz);
}
__device__ __forceinline__ void slab_hc(uint32_t px, uint32_t py, uint32_t pz, const f3& mul, const f3& add, float rtmin, float& tn, float& tf)
{
    const half2_t hx = __builtin_bit_cast(half2_t, px), hy = __builtin_bit_cast(half2_t, py), hz = __builtin_bit_cast(half2_t, pz);
    const float cx = __builtin_fmaf((float)hx.x, mul.x, add.x), cy = __builtin_fmaf((float)hy.x, mul.y, add.y), cz = __builtin_fmaf((float)hz.x, mul.z, add.z);
    const float ex = __builtin_fmaf((float)hx.y, fabsf(mul.x), 0.0f), ey = __builtin_fmaf((float)hy.y, fabsf(mul.y), 0.0f), ez = __builtin_fmaf((float)hz.y, fabsf(mul.z), 0.0f);
    tn = fmaxf(fmaxf(cx - ex, cy - ey), fmaxf(cz - ez, rtmin));
    tf = fminf(fminf(cx + ex, cy + ey), cz + ez) * kFarWiden;
}

// ---- the slab test on shared-plane nodes (NODE_FMT 10; SSpace / record layout above) ----
// builder side: magnitude of a plane at distance `dist` (>= 0, world units) inside the root's plane of its side, rounded toward 0
// after the same 2^-18 relative guard as pack_planes (product and reciprocal roundings, rotate flags in the multiplier)
__device__ __forceinline__ uint32_t pack_magnitude(float dist, float scale, bool owner1)
{
    float a = fmaxf(dist, 0.0f) * scale;
    a = a - a * 3.9e-6f;
    if (!(a < 65504.0f)) a = 65504.0f;                       // (an empty child's "planes": as far inside as the format reaches)
    return half_bits(__float2half_rz(a)) | (owner1 ? 0x8000u : 0u);
}
// the new plane on one side of one axis: the child whose plane is NOT the parent's owns it (at least one child's is: the parent's box
// is the union); magnitudes are compared after rounding, so "inherits" means "the same fp16 plane".  d0 / d1: distance of child 0's /
// child 1's plane inside the root's plane of this side; the parent's is the smaller of the two
__device__ __forceinline__ uint32_t s_new_plane(float d0, float d1, float scale)
{
    const uint32_t m0 = pack_magnitude(d0, scale, false), m1 = pack_magnitude(d1, scale, false);
    return m0 > m1 ? m0 : (m1 > m0 ? (m1 | 0x8000u) : m0);        // the larger magnitude lies further inside: that child's own plane
}
// Per-ray constants.  With r = 1 / d on an axis, a lo plane of magnitude g sits at t = (L + g s - o) r = g (s r) + (L - o) r and a hi plane at
// t = (H - g s - o) r = g (-s r) + (H - o) r.  The packed {lo, hi} pair of a node is rotated so that its low half is the plane the ray
// meets first (rotate amount in the five lowest mantissa bits of the multiplier, as in setup_ray_h9); then, whatever the ray's
// sign: near = g_low * mm + an, far = g_high * (-mm) + af with mm = |s r|, an / af = the root's near / far plane distances.
// an, af are also the ray's interval in the root's box: max3 / min3 of them start the traversal.
__device__ __forceinline__ void setup_ray_s(const f3& ro, const f3& rd, const SSpace& S, f3& mm, f3& an, f3& af)
{
    const f3 r = mk(finite_rcp(rd.x), finite_rcp(rd.y), finite_rcp(rd.z));
    const f3 alo = mk((S.lx - ro.x) * r.x, (S.ly - ro.y) * r.y, (S.lz - ro.z) * r.z), ahi = mk((S.hx - ro.x) * r.x, (S.hy - ro.y) * r.y, (S.hz - ro.z) * r.z);
    const bool nx = r.x < 0.0f, ny = r.y < 0.0f, nz = r.z < 0.0f;
    an = mk(nx ? ahi.x : alo.x, ny ? ahi.y : alo.y, nz ? ahi.z : alo.z);
    af = mk(nx ? alo.x : ahi.x, ny ? alo.y : ahi.y, nz ? alo.z : ahi.z);
    mm = mk(fabsf(r.x) * S.inv_scale, fabsf(r.y) * S.inv_scale, fabsf(r.z) * S.inv_scale);
    mm.x = __uint_as_float((__float_as_uint(mm.x) & ~31u) | (nx ? 16u : 0u));
    mm.y = __uint_as_float((__float_as_uint(mm.y) & ~31u) | (ny ? 16u : 0u));
    mm.z = __uint_as_float((__float_as_uint(mm.z) & ~31u) | (nz ? 16u : 0u));
}
// v_max_f32 / v_min_f32 as they are: fmaxf / fminf on a value the compiler cannot prove canonical (a select, a loop-carried value, an
// LDS word) first cost a v_max_f32 x, x each — two half-rate instructions per visit for the carried interval, which is never a NaN
__device__ __forceinline__ float vmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// both children of a record: entry / exit distances, given the ray's interval [tn, tf] in the node's own box
__device__ __forceinline__ void slab_s(uint32_t px, uint32_t py, uint32_t pz, const f3& mm, const f3& an, const f3& af, float tn, float tf,
                                       float& n0, float& f0, float& n1, float& f1)
{
    const uint32_t ax = rot16(px, __float_as_uint(mm.x)), ay = rot16(py, __float_as_uint(mm.y)), az = rot16(pz, __float_as_uint(mm.z));
    const half2_t hx = __builtin_bit_cast(half2_t, ax), hy = __builtin_bit_cast(half2_t, ay), hz = __builtin_bit_cast(half2_t, az);
    n0 = fmaxf(fmaxf(__builtin_fmaf((float)hx.x, mm.x, an.x), __builtin_fmaf((float)hy.x, mm.y, an.y)), vmax_raw(__builtin_fmaf((float)hz.x, mm.z, an.z), tn));
    f0 = fminf(fminf(__builtin_fmaf((float)hx.y, -mm.x, af.x), __builtin_fmaf((float)hy.y, -mm.y, af.y)), vmin_raw(__builtin_fmaf((float)hz.y, -mm.z, af.z), tf));
    n1 = fmaxf(fmaxf(__builtin_fmaf(-(float)hx.x, mm.x, an.x), __builtin_fmaf(-(float)hy.x, mm.y, an.y)), vmax_raw(__builtin_fmaf(-(float)hz.x, mm.z, an.z), tn));
    f1 = fminf(fminf(__builtin_fmaf((float)hx.y, mm.x, af.x), __builtin_fmaf((float)hy.y, mm.y, af.y)), vmin_raw(__builtin_fmaf((float)hz.y, mm.z, af.z), tf));
}
// an interval on the traversal stack: two fp16 in one dword, tn rounded down and tf rounded up (both are >= 0: v_cvt_pkrtz rounds toward 0,
// one unit in the last place is added to tf; 65504 and beyond become +inf)
__device__ __forceinline__ uint32_t pack_interval(float tn, float tf)
{
    typedef __fp16 h2v __attribute__((ext_vector_type(2)));
    const h2v p = __builtin_amdgcn_cvt_pkrtz(tn, tf);
    return __builtin_bit_cast(uint32_t, p) + 0x00010000u;
}
__device__ __forceinline__ void unpack_interval(uint32_t p, float& tn, float& tf)
{
    const half2_t h = __builtin_bit_cast(half2_t, p);
    tn = (float)h.x; tf = (float)h.y;
}

// ------------------------------------------------------------ wave votes ----
// __ballot(int) first turns the predicate into 0 / 1 in a VGPR and compares it again (two half-rate vector
// instructions per vote); the lane mask of a bool is already there.  Counts are taken as int so that the
// comparison with a threshold stays on the scalar unit (the 64-bit form is a vector compare).
__device__ __forceinline__ unsigned long long vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int popc(unsigned long long m) { return __builtin_popcount((uint32_t)m) + __builtin_popcount((uint32_t)(m >> 32)); }

// ------------------------------------------------------------ LDS lane stack ----
// One stack per lane, entry-major / lane-minor so a wave-wide push or pop touches 64
// consecutive dwords (conflict-free ds_read_b32 / ds_write_b32).
struct LaneStack {
    uint32_t* base;   // &lds[wave_region + lane]
    __device__ __forceinline__ void push(int sp, int v) const { base[sp * 64] = (uint32_t)v; }
    __device__ __forceinline__ int pop(int sp) const { return (int)base[sp * 64]; }
};

struct HitRec { float t; int slot; uint32_t prim; };

// One ray per lane through the two-child BVH.  ANY_HIT: return at the first triangle hit
// (traceOcclusion, pathTracerPrograms.cu:651-684).  Otherwise closest hit, equal t -> lowest
// original triangle index.  `active` lanes only; inactive lanes fall straight through.
template <bool ANY_HIT>
__device__ __forceinline__ bool traverse(const DeviceScene& sc, const LaneStack& st, bool active,
                                         const f3& o, const f3& d, float tmin, float tmax, HitRec& hit)
{
    hit.t = tmax; hit.slot = -1; hit.prim = 0xFFFFFFFFu;
    if (sc.n_tris == 0) return false;
    const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    int sp = 0;
    int node = active ? 0 : kSentinel;
    bool found = false;
    while (node != kSentinel) {
        if (node >= 0) {
            const BvhNode* np = sc.nodes + node;
            const float4 a = np->a, b = np->b, c = np->c;
            const int4 ch = np->d;
            // child 0: lo (a.x a.y a.z) hi (a.w b.x b.y); child 1: lo (b.z b.w c.x) hi (c.y c.z c.w)
            float x0 = (a.x - o.x) * ix, x1 = (a.w - o.x) * ix;
            float y0 = (a.y - o.y) * iy, y1 = (b.x - o.y) * iy;
            float z0 = (a.z - o.z) * iz, z1 = (b.y - o.z) * iz;
            float n0 = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), tmin));
            float f0 = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1)) * kFarWiden;
            f0 = fminf(f0, hit.t * kTieWiden);
            float u0 = (b.z - o.x) * ix, u1 = (c.y - o.x) * ix;
            float v0 = (b.w - o.y) * iy, v1 = (c.z - o.y) * iy;
            float w0 = (c.x - o.z) * iz, w1 = (c.w - o.z) * iz;
            float n1 = fmaxf(fmaxf(fminf(u0, u1), fminf(v0, v1)), fmaxf(fminf(w0, w1), tmin));
            float f1 = fminf(fminf(fmaxf(u0, u1), fmaxf(v0, v1)), fmaxf(w0, w1)) * kFarWiden;
            f1 = fminf(f1, hit.t * kTieWiden);
            const bool h0 = n0 <= f0, h1 = n1 <= f1;
            if (h0 && h1) {
                const bool first0 = n0 <= n1;
                st.push(sp, first0 ? ch.y : ch.x);
                sp++;
                node = first0 ? ch.x : ch.y;
            } else if (h0) {
                node = ch.x;
            } else if (h1) {
                node = ch.y;
            } else {
                if (sp == 0) node = kSentinel; else { sp--; node = st.pop(sp); }
            }
        } else {
            const int slot = ~node;
            const TriRecord* tp = sc.tris + slot;
            const float4 r0 = tp->r0, r1 = tp->r1, r2 = tp->r2;
            float t;
            const bool ok = tri_test(o, d, mk(r0.x, r0.y, r0.z), mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x), tmin, tmax, t);
            if (ANY_HIT) {
                if (ok) { found = true; node = kSentinel; continue; }
            } else {
                const uint32_t prim = __float_as_uint(r2.y);
                if (ok && (t < hit.t || (t == hit.t && prim < hit.prim))) { hit.t = t; hit.slot = slot; hit.prim = prim; found = true; }
            }
            if (sp == 0) node = kSentinel; else { sp--; node = st.pop(sp); }
        }
    }
    return found;
}

// ------------------------------------------------------------ four-wide tree ----
// Stack of {base, list} groups, one per tree level at most (see wide_bvh.hip); entry-major like LaneStack.
struct LaneStack2 {
    uint2* base;      // &lds[wave_region + lane]
    __device__ __forceinline__ void push(int sp, uint32_t b, uint32_t l) const { base[sp * 64] = make_uint2(b, l); }
    __device__ __forceinline__ uint2 pop(int sp) const { return base[sp * 64]; }
};

__device__ __forceinline__ float ubyte_f(uint32_t v, int k) { return (float)((v >> (8 * k)) & 0xFFu); }   // v_cvt_f32_ubyteN

// Visit wide node `node`: test its (up to four) child boxes, return the hit children near-to-far as a list of
// nibbles (8 | leaf << 2 | k; 0 terminates) and the record index of child 0.
// Plane distance = byte * (scale / d) + (origin - o) / d: the scale is a power of two, so the only roundings
// are those of (origin - o) / d and the fma itself; wide_bvh.hip's margin covers them.  A NaN (0 * inf for an
// axis-parallel ray) drops that plane from the max / min, which only widens the test.
__device__ __forceinline__ uint32_t wide_visit(const uint4* __restrict__ R, int node, const f3& ro, const f3& rinv,
                                               float rtmin, float best_t, uint32_t& base)
{
    const uint4* rp = R + 3u * (uint32_t)node;
    const uint4 h0 = rp[0], h1 = rp[1], h2 = rp[2];
    const uint32_t em = h0.w;
    const float ax = __uint_as_float((em & 0xFFu) << 23) * rinv.x;
    const float ay = __uint_as_float(((em >> 8) & 0xFFu) << 23) * rinv.y;
    const float az = __uint_as_float(((em >> 16) & 0xFFu) << 23) * rinv.z;
    const float bx = (__uint_as_float(h0.x) - ro.x) * rinv.x;
    const float by = (__uint_as_float(h0.y) - ro.y) * rinv.y;
    const float bz = (__uint_as_float(h0.z) - ro.z) * rinv.z;
    const uint32_t n_inner = (em >> 24) & 7u, n_child = em >> 27;
    const float prune_t = best_t * kTieWiden;
    base = h1.x;
    // lo planes: h1.z h1.w h2.x ; hi planes: h2.y h2.z h2.w ; near = lo for a positive direction
    const bool ngx = rinv.x < 0.0f, ngy = rinv.y < 0.0f, ngz = rinv.z < 0.0f;
    const uint32_t qnx = ngx ? h2.y : h1.z, qfx = ngx ? h1.z : h2.y;
    const uint32_t qny = ngy ? h2.z : h1.w, qfy = ngy ? h1.w : h2.z;
    const uint32_t qnz = ngz ? h2.w : h2.x, qfz = ngz ? h2.x : h2.w;
    uint32_t key[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float tnx = __builtin_fmaf(ubyte_f(qnx, k), ax, bx), tfx = __builtin_fmaf(ubyte_f(qfx, k), ax, bx);
        const float tny = __builtin_fmaf(ubyte_f(qny, k), ay, by), tfy = __builtin_fmaf(ubyte_f(qfy, k), ay, by);
        const float tnz = __builtin_fmaf(ubyte_f(qnz, k), az, bz), tfz = __builtin_fmaf(ubyte_f(qfz, k), az, bz);
        const float nn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, rtmin));
        const float ff = fminf(fminf(fminf(tfx, tfy), tfz) * kFarWiden, prune_t);
        const bool hit = (nn <= ff) && ((uint32_t)k < n_child);
        const uint32_t tag = 8u | ((uint32_t)k >= n_inner ? 4u : 0u) | (uint32_t)k;
        key[k] = hit ? ((__float_as_uint(nn) & ~15u) | tag) : 0xFFFFFFF0u;
    }
    // sorting network (ascending); misses carry a zero nibble and sort last
#define PT_CAS(i, j) { const uint32_t lo_ = min(key[i], key[j]), hi_ = max(key[i], key[j]); key[i] = lo_; key[j] = hi_; }
    PT_CAS(0, 1) PT_CAS(2, 3) PT_CAS(0, 2) PT_CAS(1, 3) PT_CAS(1, 2)
#undef PT_CAS
    return (key[0] & 15u) | ((key[1] & 15u) << 4) | ((key[2] & 15u) << 8) | ((key[3] & 15u) << 12);
}

}  // namespace ptd
