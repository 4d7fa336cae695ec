// wide_bvh.hip — four-wide BVH with 8-bit child boxes, derived from the two-child tree of lbvh_build.hip.
//
// Why: the render kernel is bound by the texture-address unit's gather rate (≈0.65 cycles per lane per
// 16-byte load, profiles/r01_ubench_gather.txt), so what counts is 16-byte fetches per ray.  A two-child
// fp32 node costs 4 fetches per visit and a Cornell-class ray visits ≈10 of them; a four-wide node with
// boxes quantised to a node-local 8-bit grid (Ylitie, Karras, Laine 2017) is 48 bytes = 3 fetches and
// a ray visits about half as many.
//
// One array of 48-byte records holds BOTH the wide nodes and the triangles (TriRecord is 48 bytes too):
// the children of a node — inner nodes first, then triangles — are consecutive records, so a child is
// addressed as base + k and a traversal-stack entry is {base, sorted list of k}.  Record 0 is the root.
//
//   WideNode (12 dwords):
//     0..2  origin x y z (fp32)
//     3     ex | ey << 8 | ez << 16 | n_inner << 24 | n_children << 27      scale_a = 2^(e_a - 127)
//     4     base: record index of child 0
//     5     -
//     6..8  lo.x[4] lo.y[4] lo.z[4]      one byte per child, plane = origin + scale * byte
//     9..11 hi.x[4] hi.y[4] hi.z[4]
//
// Quantisation is outward and checked in double precision against the decoded planes, with a margin
// (kMarginRel x largest scene coordinate) that covers the fp32 rounding of the kernel's
// t = byte * (scale / d) + (origin - o) / d form; the box test therefore stays conservative and hits
// remain bit-identical to the two-child tree (same triangle test, same tie rule).
//
// The collapse runs on the host (depth-first layout, widest-area child opened first); it is part of scene
// set-up, timed separately in LbvhResult::wide_ms.
#include "lbvh_build.h"
#include <chrono>
#include <cmath>
#include <vector>

namespace ptd {

namespace {

constexpr double kMarginRel = 1.0 / 1048576.0;    // 2^-20 of the largest |coordinate| / extent of the scene

struct Item { int ref; float lo[3], hi[3]; };

inline double area(const Item& it)
{
    const double dx = (double)it.hi[0] - it.lo[0], dy = (double)it.hi[1] - it.lo[1], dz = (double)it.hi[2] - it.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

inline void children_of(const BvhNode& nd, Item out[2], int& n_out)
{
    n_out = 0;
    Item a; a.ref = nd.d.x; a.lo[0] = nd.a.x; a.lo[1] = nd.a.y; a.lo[2] = nd.a.z; a.hi[0] = nd.a.w; a.hi[1] = nd.b.x; a.hi[2] = nd.b.y;
    Item b; b.ref = nd.d.y; b.lo[0] = nd.b.z; b.lo[1] = nd.b.w; b.lo[2] = nd.c.x; b.hi[0] = nd.c.y; b.hi[1] = nd.c.z; b.hi[2] = nd.c.w;
    if (a.lo[0] <= a.hi[0] && a.lo[1] <= a.hi[1] && a.lo[2] <= a.hi[2]) out[n_out++] = a;
    if (b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2]) out[n_out++] = b;     // single-triangle scene: empty box
}

struct Rec { uint32_t w[12]; };
static_assert(sizeof(Rec) == 48 && sizeof(TriRecord) == 48, "record size");

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// Encode one wide node.  Returns false if a plane cannot be represented (never for finite boxes).
bool encode(const Item* it, int n_items, int n_inner, uint32_t base, double margin, Rec& out)
{
    memset(&out, 0, sizeof(out));
    uint32_t ebits[3];
    uint8_t qlo[3][4], qhi[3][4];
    float origin[3];
    for (int a = 0; a < 3; a++) {
        double lo_min = 1e300, hi_max = -1e300;
        for (int k = 0; k < n_items; k++) { lo_min = std::fmin(lo_min, (double)it[k].lo[a] - margin); hi_max = std::fmax(hi_max, (double)it[k].hi[a] + margin); }
        float o = (float)lo_min;
        if ((double)o > lo_min) o = std::nextafterf(o, -INFINITY);
        origin[a] = o;
        const double ext = hi_max - (double)o;
        int e = (int)std::ceil(std::log2(ext / 255.0));
        if (e < -100) e = -100;
        for (;; e++) {
            if (e > 120) return false;
            const double scale = std::ldexp(1.0, e);
            bool ok = true;
            for (int k = 0; k < 4; k++) { qlo[a][k] = 255; qhi[a][k] = 0; }          // unused slots: inverted
            for (int k = 0; k < n_items && ok; k++) {
                const double l = (double)it[k].lo[a] - margin, h = (double)it[k].hi[a] + margin;
                double ql = std::floor((l - (double)o) / scale), qh = std::ceil((h - (double)o) / scale);
                if (ql < 0) ql = 0;
                while (ql > 0 && (double)o + ql * scale > l) ql -= 1;
                while ((double)o + qh * scale < h) qh += 1;
                if (qh > 255 || ql > 255) { ok = false; break; }
                qlo[a][k] = (uint8_t)ql; qhi[a][k] = (uint8_t)qh;
            }
            if (ok) break;
        }
        ebits[a] = (uint32_t)(e + 127);
    }
    out.w[0] = f2u(origin[0]); out.w[1] = f2u(origin[1]); out.w[2] = f2u(origin[2]);
    out.w[3] = ebits[0] | (ebits[1] << 8) | (ebits[2] << 16) | ((uint32_t)n_inner << 24) | ((uint32_t)n_items << 27);
    out.w[4] = base;
    for (int a = 0; a < 3; a++) {
        out.w[6 + a] = qlo[a][0] | (qlo[a][1] << 8) | (qlo[a][2] << 16) | ((uint32_t)qlo[a][3] << 24);
        out.w[9 + a] = qhi[a][0] | (qhi[a][1] << 8) | (qhi[a][2] << 16) | ((uint32_t)qhi[a][3] << 24);
    }
    return true;
}

}  // namespace

bool build_wide4(LbvhResult& r, hipStream_t stream, std::string& err)
{
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = r.n_tris;
    if (n == 0) return true;
    if (!ensure_nodes(r, stream, err)) return false;
    std::vector<BvhNode> nodes(r.n_nodes);
    std::vector<TriRecord> tris(n);
    hipError_t e = hipMemcpyAsync(nodes.data(), r.nodes, (size_t)r.n_nodes * sizeof(BvhNode), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tris.data(), r.tris, (size_t)n * sizeof(TriRecord), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { err = std::string("build_wide4 read-back: ") + hipGetErrorString(e); return false; }

    double big = 1.0;
    for (int k = 0; k < 3; k++) {
        big = std::fmax(big, std::fabs((double)r.scene_lo[k]));
        big = std::fmax(big, std::fabs((double)r.scene_hi[k]));
        big = std::fmax(big, (double)r.scene_hi[k] - (double)r.scene_lo[k]);
    }
    const double margin = big * kMarginRel;

    struct Todo { int bin; uint32_t rec; uint32_t depth; };
    std::vector<Rec> recs;
    recs.reserve((size_t)n + n / 2 + 4);
    recs.resize(1);
    std::vector<Todo> todo;
    todo.reserve(n / 2 + 4);
    todo.push_back({0, 0u, 1u});
    uint32_t n_wnodes = 0, depth_max = 0;
    // depth-first: a node's children block is laid down when the node is reached, and its first inner child is
    // reached next, so a root-to-leaf descent walks forward through nearby memory (matters once the tree
    // outgrows the L2: the 1.3 M-triangle scene)
    while (!todo.empty()) {
        const Todo td = todo.back();
        todo.pop_back();
        Item items[4]; int n_items = 0;
        { Item c[2]; int nc; children_of(nodes[td.bin], c, nc); for (int k = 0; k < nc; k++) items[n_items++] = c[k]; }
        while (n_items < 4) {
            int pick = -1; double best = -1.0;
            for (int k = 0; k < n_items; k++) if (items[k].ref >= 0) { const double a = area(items[k]); if (a > best) { best = a; pick = k; } }
            if (pick < 0) break;
            Item c[2]; int nc; children_of(nodes[items[pick].ref], c, nc);
            if (nc == 0) { items[pick] = items[--n_items]; continue; }
            items[pick] = c[0];
            if (nc == 2) items[n_items++] = c[1];
        }
        if (n_items == 0) { err = "build_wide4: node without children"; return false; }
        // inner children first, then triangles (stable)
        Item ord[4]; int n_inner = 0, m = 0;
        for (int k = 0; k < n_items; k++) if (items[k].ref >= 0) ord[m++] = items[k];
        n_inner = m;
        for (int k = 0; k < n_items; k++) if (items[k].ref < 0) ord[m++] = items[k];
        const uint32_t base = (uint32_t)recs.size();
        recs.resize(recs.size() + (size_t)n_items);
        for (int k = n_items - 1; k >= 0; k--) {
            if (k < n_inner) todo.push_back({ord[k].ref, base + (uint32_t)k, td.depth + 1});
            else memcpy(&recs[base + k], &tris[(size_t)(~ord[k].ref)], 48);
        }
        Rec nd;
        if (!encode(ord, n_items, n_inner, base, margin, nd)) { err = "build_wide4: box not representable"; return false; }
        recs[td.rec] = nd;
        n_wnodes++;
        if (td.depth > depth_max) depth_max = td.depth;
    }
    if (recs.size() != (size_t)n_wnodes + n) { err = "build_wide4: record count mismatch"; return false; }
    if (recs.size() >= 0x7FFFFFF0ull) { err = "build_wide4: too many records"; return false; }

    e = hipMalloc((void**)&r.wrecs, recs.size() * 48);
    if (e == hipSuccess) e = hipMemcpyAsync(r.wrecs, recs.data(), recs.size() * 48, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { err = std::string("build_wide4 upload: ") + hipGetErrorString(e); return false; }
    r.n_wrecs = (uint32_t)recs.size();
    r.n_wnodes = n_wnodes;
    r.wide_depth = depth_max;
    r.wide_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

}  // namespace ptd
