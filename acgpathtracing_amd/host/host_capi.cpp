// host_capi.cpp — C exports of the host-side mirror (OBJ ingest, Camera, Trackball) so the
// Python tests and bench can drive the same C++ the headless app uses.  CPU only.
#include <cstdint>
#include <cstring>
#include "Camera.h"
#include "ImageIO.h"
#include "TinyObjWrapper.h"
#include "Trackball.h"

#define HOST_API extern "C" __attribute__((visibility("default")))
using namespace acgpt;

HOST_API void* pth_obj_load(const char* path)
{
    TinyObjWrapper* w = new TinyObjWrapper();
    w->loadFile(path);
    return w;
}
HOST_API int pth_obj_ok(void* h) { return ((TinyObjWrapper*)h)->loaded() ? 1 : 0; }
HOST_API const char* pth_obj_error(void* h) { return ((TinyObjWrapper*)h)->error().c_str(); }
HOST_API const char* pth_obj_warning(void* h) { return ((TinyObjWrapper*)h)->warning().c_str(); }
HOST_API void pth_obj_sizes(void* h, size_t* n_vert_floats, size_t* n_indices, size_t* n_mat_ids, size_t* n_mats)
{
    TinyObjWrapper* w = (TinyObjWrapper*)h;
    *n_vert_floats = w->getVerticesFloat().size();
    *n_indices = w->getIndexBuffer().size();
    *n_mat_ids = w->getMaterialIndices().size();
    *n_mats = w->getNumMaterials();
}
HOST_API void pth_obj_fill(void* h, float* verts, uint32_t* indices, uint32_t* mat_ids, void* mats40)
{
    TinyObjWrapper* w = (TinyObjWrapper*)h;
    std::vector<float> v = w->getVerticesFloat();
    std::vector<uint32_t> i = w->getIndexBuffer(), m = w->getMaterialIndices();
    std::vector<Material> mm = w->getMaterials();
    if (!v.empty()) memcpy(verts, v.data(), v.size() * 4);
    if (!i.empty()) memcpy(indices, i.data(), i.size() * 4);
    if (!m.empty()) memcpy(mat_ids, m.data(), m.size() * 4);
    if (!mm.empty()) memcpy(mats40, mm.data(), mm.size() * sizeof(Material));
}
HOST_API void pth_obj_free(void* h) { delete (TinyObjWrapper*)h; }

HOST_API void pth_camera_uvw(const float* eye, const float* lookat, const float* up, float fovY, float aspect,
                             float* U3, float* V3, float* W3)
{
    Camera cam(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
               make_float3(up[0], up[1], up[2]), fovY, aspect);
    float3 U, V, W;
    cam.UVWFrame(U, V, W);
    U3[0] = U.x; U3[1] = U.y; U3[2] = U.z; V3[0] = V.x; V3[1] = V.y; V3[2] = V.z; W3[0] = W.x; W3[1] = W.y; W3[2] = W.z;
}

// events: (kind, a, b) triples; kind 0 startTracking(a,b), 1 updateTracking(a,b,w,h), 2 wheelEvent(a)
HOST_API void pth_trackball_script(const float* eye, const float* lookat, const float* up, float fovY, float aspect,
                                   int view_mode, float move_speed, int gimbal_lock, int canvas_w, int canvas_h,
                                   const int* events, size_t n_events, float* out9)
{
    Camera cam(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
               make_float3(up[0], up[1], up[2]), fovY, aspect);
    Trackball tb;
    tb.setCamera(&cam);
    tb.setMoveSpeed(move_speed);
    tb.setReferenceFrame(make_float3(1.0f, 0.0f, 0.0f), make_float3(0.0f, 0.0f, 1.0f), make_float3(0.0f, 1.0f, 0.0f));
    tb.setGimbalLock(gimbal_lock != 0);
    tb.setViewMode(view_mode == 0 ? Trackball::EyeFixed : Trackball::LookAtFixed);
    for (size_t i = 0; i < n_events; i++) {
        const int* e = events + 3 * i;
        if (e[0] == 0) tb.startTracking(e[1], e[2]);
        else if (e[0] == 1) tb.updateTracking(e[1], e[2], canvas_w, canvas_h);
        else if (e[0] == 2) tb.wheelEvent(e[1]);
    }
    float3 a = cam.eye(), b = cam.lookat(), c = cam.up();
    out9[0] = a.x; out9[1] = a.y; out9[2] = a.z; out9[3] = b.x; out9[4] = b.y; out9[5] = b.z; out9[6] = c.x; out9[7] = c.y; out9[8] = c.z;
}

// rgba: width*height*4 bytes, row 0 = bottom; suffix selects PPM or PNG (sutil::saveImage conventions)
HOST_API int pth_save_image(const char* path, const unsigned char* rgba, int width, int height)
{
    return saveImage(path, rgba, width, height) ? 0 : 1;
}
