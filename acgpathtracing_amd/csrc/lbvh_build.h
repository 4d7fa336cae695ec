// lbvh_build.h — host entry of the on-device LBVH build (see lbvh_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstring>
#include "pt_device.h"

namespace ptd {

struct LbvhResult {
    BvhNode*   nodes = nullptr;        // device, n_nodes (fp32 boxes, 64 B)
    QNode*     qnodes = nullptr;       // device, n_nodes (16-bit grid boxes, 32 B)
    BvhNode*   cnodes = nullptr;       // device, n_nodes (centre + half-extent boxes, 64 B)
    HNode*     hnodes = nullptr;       // device, n_nodes (fp16 boxes, 32 B)
    HNode*     top_nodes = nullptr;    // device, kTopNodesMax: the first n_top inner nodes breadth first (children inside the array: kTopNodeFlag | position)
    uint32_t   n_top = 0;
    HSpace     hspace = {0, 0, 0, 1};
    float      half_area_ratio = 0.0f; // sum of child-box areas after fp16 outward rounding / before (what a random ray pays)
    float      half_box_inflation = 0.0f; // mean over the child boxes of their own area after / before (what a ray through the finest geometry pays)
    QGrid      grid = {};              // world -> grid transform of qnodes
    TriRecord* tris = nullptr;         // device, n_tris, Morton order
    float4*    shade = nullptr;        // device, n_tris, same order: geometric normal + material id (pt_device.h DeviceScene::shade)
    uint4*     wrecs = nullptr;        // device, n_wrecs x 48 B: four-wide nodes + triangles (wide_bvh.hip)
    uint32_t   n_wrecs = 0, n_wnodes = 0, wide_depth = 0;
    float      wide_ms = 0.0f;         // host collapse + upload
    uint32_t*  keys_sorted = nullptr;  // device, n_tris
    uint32_t*  vals_sorted = nullptr;  // device, n_tris (original triangle index per slot)
    uint32_t   n_tris = 0, n_nodes = 0, max_depth = 0;
    float      scene_lo[3] = {0, 0, 0}, scene_hi[3] = {0, 0, 0};
    float      build_ms = 0.0f;
    int        mode = 0;               // 0 Karras radix tree, 1 PLOC
    uint32_t   build_iterations = 0;   // PLOC merge rounds
};

// Host arrays in, device BVH out.  Synchronous on return.  false + err on failure.
// mode 0: Karras radix tree over the sorted Morton codes; mode 1: PLOC over the same order.
bool build_lbvh(const float* h_verts_xyzw, size_t n_verts, const uint32_t* h_idx, size_t n_tris,
                const uint32_t* h_mat_ids, int mode, hipStream_t stream, LbvhResult& out, std::string& err);

void free_lbvh(LbvhResult& r);

// Writes each material's bsdfType and whether it emits into the upper byte of the shade records' material word (pt_device.h
// kShadeBsdfShift, kShadeHasKe), once the repacked materials are on the device: what lets closest-hit shading fetch
// {diffuse, ior} alone for a diffuse, non-emissive hit.  Synchronous on return.
bool tag_shade_records(LbvhResult& r, const DevMaterial* d_mats, hipStream_t stream, std::string& err);

// The first kTopNodesMax inner nodes breadth first (LbvhResult::top_nodes), on first use (experiment variants only).
bool build_top_nodes(LbvhResult& r, hipStream_t stream, std::string& err);

// Collapse the two-child tree into the four-wide, 8-bit-quantised record array (wide_bvh.hip).
bool build_wide4(LbvhResult& r, hipStream_t stream, std::string& err);

}  // namespace ptd
