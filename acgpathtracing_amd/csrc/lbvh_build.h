// lbvh_build.h — host entry of the on-device LBVH build (see lbvh_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstring>
#include "pt_device.h"

namespace ptd {

struct LbvhResult {
    // A scene keeps ONE of the node arrays nodes / hnodes / hcnodes — the one its render kernel reads; the others are released after the
    // build (keep_one_node_array) and come back, bit for bit, on first use (ensure_nodes / ensure_hnodes / ensure_hcnodes).
    BvhNode*   nodes = nullptr;        // device, n_nodes (fp32 boxes, 64 B)
    HNode*     hnodes = nullptr;       // device, n_nodes (fp16 {lo, hi} boxes, 32 B)
    QNode*     qnodes = nullptr;       // device, n_nodes (16-bit grid boxes, 32 B): experiment formats, on first use (ensure_qnodes)
    BvhNode*   cnodes = nullptr;       // device, n_nodes (centre + half-extent boxes, 64 B): experiment format, on first use (ensure_cnodes)
    float      pad_abs = 0.0f;         // absolute pad of the triangle boxes (record_aabb): 2^-19 of the scene's largest |coordinate|
    HNode*     top_nodes = nullptr;    // device, kTopNodesMax: the first n_top inner nodes breadth first (children inside the array: kTopNodeFlag | position)
    uint32_t   n_top = 0;
    HSpace     hspace = {0, 0, 0, 1, 1, 1, 1, 0};
    float      half_area_ratio = 0.0f; // sum of child-box areas after fp16 outward rounding / before (what a random ray pays)
    float      half_box_inflation = 0.0f; // mean over the child boxes of their own area after / before (what a ray through the finest geometry pays)
    QGrid      grid = {};              // world -> grid transform of qnodes
    TriRecord* tris = nullptr;         // device, n_tris, Morton order
    float4*    shade = nullptr;        // device, n_tris, same order: geometric normal + material id (pt_device.h DeviceScene::shade)
    HNode*     hcnodes_alt = nullptr;  // experiments build: a renumbered copy of hcnodes (reorder_hcnodes), what the kernels walk while it exists
    size_t     hcnodes_alt_bytes = 0;
    HNode*     hcnodes = nullptr;      // device, n_nodes: fp16 centre | half-extent boxes, child references as byte offsets (NODE_FMT 11; ensure_hcnodes)
    uint4*     srecs = nullptr;        // device, n_srecs x 16 B: shared-plane nodes + triangles in one array (NODE_FMT 10; ensure_srecs)
    uint32_t   n_srecs = 0;
    bool       srecs_wide = false;     // child references: one 30-bit index + two flags (NODE_FMT 12) instead of two 15-bit references (NODE_FMT 10)
    SSpace     sspace = {0, 0, 0, 0, 0, 0, 1};
    uint4*     wrecs = nullptr;        // device, n_wrecs x 48 B: four-wide nodes + triangles (wide_bvh.hip)
    uint32_t   n_wrecs = 0, n_wnodes = 0, wide_depth = 0;
    float      wide_ms = 0.0f;         // host collapse + upload
    uint32_t   n_tris = 0, n_nodes = 0, max_depth = 0;
    float      scene_lo[3] = {0, 0, 0}, scene_hi[3] = {0, 0, 0};
    float      build_ms = 0.0f;
    int        mode = 0;               // 0 Karras radix tree, 1 PLOC, 2 PLOC + insertion-based optimisation on the host (scenes up to 16 384 triangles; the default)
    float      opt_area_before = 0.0f, opt_area_after = 0.0f, opt_ms = 0.0f;      // mode 2: sum of the inner nodes' surface areas before / after, host time
    uint32_t   opt_passes = 0;
    uint32_t   build_iterations = 0;   // PLOC merge rounds
};

// Host arrays in, device BVH out.  Synchronous on return.  false + err on failure.
// mode 0: Karras radix tree over the sorted Morton codes; mode 1: PLOC over the same order.
bool build_lbvh(const float* h_verts_xyzw, size_t n_verts, const uint32_t* h_idx, size_t n_tris,
                const uint32_t* h_mat_ids, int mode, hipStream_t stream, LbvhResult& out, std::string& err);

void free_lbvh(LbvhResult& r);

// One node array per scene: release the one the chosen kernel does not read (no-op when it is the only copy of the topology) ...
void keep_one_node_array(LbvhResult& r, int keep);      // keep: 0 fp32 nodes, 1 fp16 {lo, hi} nodes, 2 fp16 {centre, half extent} nodes; frees the others and what derives from them
// ... and get any of them back on first use.  fp32 nodes: from the topology in the fp16 nodes + the triangle records, the same bits as
// the build's own (unions are exact); the others from the fp32 nodes.  Synchronous on return.
bool ensure_nodes(LbvhResult& r, hipStream_t stream, std::string& err);
bool ensure_hnodes(LbvhResult& r, hipStream_t stream, std::string& err);
bool ensure_qnodes(LbvhResult& r, hipStream_t stream, std::string& err);
bool ensure_cnodes(LbvhResult& r, hipStream_t stream, std::string& err);
bool ensure_hcnodes(LbvhResult& r, hipStream_t stream, std::string& err);    // fp16 centre / half-extent nodes (NODE_FMT 11), from the fp32 nodes
#ifdef ACGPT_EXPERIMENTS
bool ensure_srecs(LbvhResult& r, bool wide_refs, hipStream_t stream, std::string& err);      // shared-plane records (NODE_FMT 10 / 12), from the fp32 nodes
// experiment: renumber the centre / half-extent nodes (0: the build's order; 1: sibling pairs in one 64-byte line; 2: depth first)
bool reorder_hcnodes(LbvhResult& r, int mode, hipStream_t stream, std::string& err);
#endif      // shared-plane records (NODE_FMT 10), from the fp32 nodes
// (Morton code, original triangle index) per leaf slot in sorted order, recomputed from the records (the build keeps no copy of its sort keys)
bool read_morton(const LbvhResult& r, hipStream_t stream, uint32_t* h_codes, uint32_t* h_prims, std::string& err);
// bytes of device memory the scene's arrays hold right now
size_t scene_device_bytes(const LbvhResult& r);

// Writes each material's bsdfType and whether it emits into the upper byte of the shade records' material word (pt_device.h
// kShadeBsdfShift, kShadeHasKe), once the repacked materials are on the device: what lets closest-hit shading fetch
// {diffuse, ior} alone for a diffuse, non-emissive hit.  Synchronous on return.
bool tag_shade_records(LbvhResult& r, const DevMaterial* d_mats, hipStream_t stream, std::string& err);

// The first kTopNodesMax inner nodes breadth first (LbvhResult::top_nodes), on first use (experiment variants only).
bool build_top_nodes(LbvhResult& r, hipStream_t stream, std::string& err);

// Collapse the two-child tree into the four-wide, 8-bit-quantised record array (wide_bvh.hip).
bool build_wide4(LbvhResult& r, hipStream_t stream, std::string& err);

}  // namespace ptd
