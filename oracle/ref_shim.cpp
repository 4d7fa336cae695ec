/*
 * ref_shim.cpp — thin C exports over the REFERENCE'S OWN sources, compiled from where
 * they lie under /root/reference (never copied).  Builds only in the authoring container
 * (the reference does not exist on the GPU box); output goes to oracle/_ref/libref.so,
 * which is git-ignored.  Used by tests/golden/make_golden.py to generate fixtures and by
 * tests/test_oracle_golden.py to pin oracle_pt.cpp function by function.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Reference pieces compiled here (all build as-is with g++ and the CUDA *headers* that
 * ship inside the image's triton wheel — no stand-in headers are written):
 *   cuda/random.h            tea<4>, lcg, rnd
 *   cuda/helpers.h           make_color, refract
 *   sutil/vec_math.h         normalize, reflect, faceforward, lerp, cross, operator/
 *   sutil/Camera.cpp         Camera::UVWFrame
 *   sutil/Trackball.cpp      Trackball (orbit / zoom)
 *   sutil/WorkDistribution.h StaticWorkDistribution
 *   PathTracer_Optix/TinyObjWrapper.cpp (+ util/tiny_obj_loader.h)
 * NOT compiled: PathTracer_Optix/pathTracerPrograms.cu and pathTracer.h include
 * <optix.h> (OptiX SDK, absent) — unbuildable here.
 */
#include <cuda_runtime.h>
#include <cuda/random.h>
#include <sutil/vec_math.h>
#include <cuda/helpers.h>
#include <sutil/Camera.h>
#include <sutil/Trackball.h>
#include <sutil/WorkDistribution.h>
#include <PathTracer_Optix/TinyObjWrapper.h>
#include <cstring>
#include <cstdint>

#define REF_API extern "C" __attribute__((visibility("default")))

REF_API uint32_t ref_tea4(uint32_t v0, uint32_t v1) { return tea<4>(v0, v1); }

REF_API void ref_rnd_stream(uint32_t seed, size_t n, uint32_t* states_out, float* values_out)
{
    for (size_t i = 0; i < n; i++) { float v = rnd(seed); if (states_out) states_out[i] = seed; if (values_out) values_out[i] = v; }
}

REF_API void ref_make_color(const float* rgb, size_t n, uint8_t* rgba_out)
{
    for (size_t i = 0; i < n; i++) {
        uchar4 c = make_color(make_float3(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]));
        rgba_out[4 * i] = c.x; rgba_out[4 * i + 1] = c.y; rgba_out[4 * i + 2] = c.z; rgba_out[4 * i + 3] = c.w;
    }
}

REF_API void ref_refract(const float* i3, const float* n3, float ior, float* r3, int* ok)
{
    float3 r; bool b = refract(r, make_float3(i3[0], i3[1], i3[2]), make_float3(n3[0], n3[1], n3[2]), ior);
    r3[0] = r.x; r3[1] = r.y; r3[2] = r.z; *ok = b;
}

/* op: 0 normalize(a) 1 reflect(a,b) 2 faceforward(a,b,c) 3 lerp(a,b,s) 4 cross(a,b) 5 a/s */
REF_API void ref_vec_op(int op, const float* a, const float* b, const float* c, float s, float* out)
{
    float3 A = make_float3(a[0], a[1], a[2]);
    float3 B = b ? make_float3(b[0], b[1], b[2]) : make_float3(0.f);
    float3 C = c ? make_float3(c[0], c[1], c[2]) : make_float3(0.f);
    float3 r = make_float3(0.f);
    switch (op) {
    case 0: r = normalize(A); break;
    case 1: r = reflect(A, B); break;
    case 2: r = faceforward(A, B, C); break;
    case 3: r = lerp(A, B, s); break;
    case 4: r = cross(A, B); break;
    case 5: r = A / s; break;
    }
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

REF_API void ref_camera_uvw(const float* eye, const float* lookat, const float* up, float fovY, float aspect,
                            float* U3, float* V3, float* W3)
{
    sutil::Camera cam(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
                      make_float3(up[0], up[1], up[2]), fovY, aspect);
    float3 U, V, W; cam.UVWFrame(U, V, W);
    U3[0] = U.x; U3[1] = U.y; U3[2] = U.z; V3[0] = V.x; V3[1] = V.y; V3[2] = V.z; W3[0] = W.x; W3[1] = W.y; W3[2] = W.z;
}

/* Trackball driven by a script of events: each event = (kind, x, y); kind 0 startTracking,
 * 1 updateTracking, 2 wheelEvent(dir = x).  out: eye(3) lookat(3) up(3) after the script. */
REF_API void ref_trackball_script(const float* eye, const float* lookat, const float* up, float fovY, float aspect,
                                  int view_mode, float move_speed, int gimbal_lock,
                                  int canvas_w, int canvas_h, const int* events, size_t n_events, float* out9)
{
    sutil::Camera cam(make_float3(eye[0], eye[1], eye[2]), make_float3(lookat[0], lookat[1], lookat[2]),
                      make_float3(up[0], up[1], up[2]), fovY, aspect);
    sutil::Trackball tb;
    tb.setCamera(&cam);
    tb.setMoveSpeed(move_speed);
    tb.setReferenceFrame(make_float3(1.0f, 0.0f, 0.0f), make_float3(0.0f, 0.0f, 1.0f), make_float3(0.0f, 1.0f, 0.0f));
    tb.setGimbalLock(gimbal_lock != 0);
    tb.setViewMode(view_mode == 0 ? sutil::Trackball::EyeFixed : sutil::Trackball::LookAtFixed);
    for (size_t i = 0; i < n_events; i++) {
        const int* e = events + 3 * i;
        if (e[0] == 0) tb.startTracking(e[1], e[2]);
        else if (e[0] == 1) tb.updateTracking(e[1], e[2], canvas_w, canvas_h);
        else if (e[0] == 2) tb.wheelEvent(e[1]);
    }
    float3 a = cam.eye(), b = cam.lookat(), c = cam.up();
    out9[0] = a.x; out9[1] = a.y; out9[2] = a.z; out9[3] = b.x; out9[4] = b.y; out9[5] = b.z; out9[6] = c.x; out9[7] = c.y; out9[8] = c.z;
}

REF_API int ref_num_samples(int num_gpus, int width, int height)
{
    StaticWorkDistribution wd; wd.setRasterSize(width, height); wd.setNumGPUs(num_gpus);
    return wd.numSamples(0);
}
REF_API void ref_sample_pixel(int num_gpus, int width, int height, int gpu_idx, int sample_idx, int* px, int* py)
{
    StaticWorkDistribution wd; wd.setRasterSize(width, height); wd.setNumGPUs(num_gpus);
    int2 p = wd.getSamplePixel(gpu_idx, sample_idx); *px = p.x; *py = p.y;
}

/* TinyObjWrapper: two-call protocol (sizes, then fill). */
struct RefObj { TinyObjWrapper w; bool ok; };
REF_API void* ref_obj_load(const char* path)
{
    RefObj* o = new RefObj(); o->ok = o->w.loadFile(path); return o;
}
REF_API int ref_obj_ok(void* h) { return ((RefObj*)h)->ok; }
REF_API void ref_obj_sizes(void* h, size_t* n_vert_floats, size_t* n_indices, size_t* n_mat_ids, size_t* n_mats)
{
    RefObj* o = (RefObj*)h;
    *n_vert_floats = o->w.getVerticesFloat().size(); *n_indices = o->w.getIndexBuffer().size();
    *n_mat_ids = o->w.getMaterialIndices().size(); *n_mats = o->w.getNumMaterials();
}
REF_API void ref_obj_fill(void* h, float* verts, uint32_t* indices, uint32_t* mat_ids, void* mats40)
{
    RefObj* o = (RefObj*)h;
    auto v = o->w.getVerticesFloat(); auto i = o->w.getIndexBuffer(); auto m = o->w.getMaterialIndices(); auto mm = o->w.getMaterials();
    if (!v.empty()) memcpy(verts, v.data(), v.size() * 4);
    if (!i.empty()) memcpy(indices, i.data(), i.size() * 4);
    if (!m.empty()) memcpy(mat_ids, m.data(), m.size() * 4);
    static_assert(sizeof(Material) == 40, "Material layout");
    if (!mm.empty()) memcpy(mats40, mm.data(), mm.size() * sizeof(Material));
}
REF_API void ref_obj_free(void* h) { delete (RefObj*)h; }
