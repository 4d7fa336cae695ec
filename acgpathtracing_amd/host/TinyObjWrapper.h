// TinyObjWrapper.h — OBJ/MTL ingest with the public API of the reference's
// PathTracer_Optix/TinyObjWrapper.h:77-115 (same method names and return types, same
// Material / BSDFType layout), on top of a small parser of our own instead of
// util/tiny_obj_loader.h.  Output contract (TinyObjWrapper.cpp:138-244):
//   getVerticesFloat()   x y z 1.0 per `v` line, file order
//   getIndexBuffer()     vertex indices of all triangulated faces, file order
//   getMaterialIndices() one id per triangle; a face with no (or an unknown) usemtl
//                        gets 0xFFFFFFFF (tinyobj's -1)
//   getMaterials()       one record per newmtl; BSDF from the NAME: contains
//                        "Refractive" -> refraction, else "Metallic" -> metallic
// Triangulation follows tinyobjloader v2.0.0 (util/tiny_obj_loader.h:1510-1965): quads
// are split along the shorter diagonal, larger polygons are ear-clipped.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "vec_types.h"

namespace acgpt {

struct Vertex { float x, y, z; };

enum BSDFType { BSDF_DIFFUSE, BSDF_METALLIC, BSDF_REFRACTION };

struct Material {
    float3 diffuse;
    float3 emission;
    float roughness;
    float metallic;
    float ior;
    BSDFType bsdfType;
};
static_assert(sizeof(Material) == 40, "Material must match pt_material");

class TinyObjWrapper {
public:
    TinyObjWrapper() {}
    explicit TinyObjWrapper(const std::string& filename);
    ~TinyObjWrapper() {}

    bool loadFile(const std::string& filename);

    std::vector<float> getVerticesFloat() const;
    std::vector<Material> getMaterials() const;
    std::vector<uint32_t> getMaterialIndices() const;
    std::vector<uint32_t> getIndexBuffer() const;
    size_t getNumMaterials() const;

    // extras (not in the reference): diagnostics of the last load
    const std::string& warning() const { return _warn; }
    const std::string& error() const { return _err; }
    bool loaded() const { return dataLoaded; }

private:
    bool dataLoaded = false;
    std::vector<float> _vertices;
    std::vector<Material> _materials;
    std::vector<uint32_t> _materialIndices;
    std::vector<uint32_t> _indexBuffer;
    std::string _warn, _err;
};

}  // namespace acgpt
