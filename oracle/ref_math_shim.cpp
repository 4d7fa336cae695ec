/*
 * ref_math_shim.cpp — C exports over the OptiX-free sampling / BSDF helpers of the REFERENCE'S OWN
 * PathTracer_Optix/pathTracerPrograms.cu (:54-85, :265-284, :341-380, :455-476, :494-510, :534-559).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Builds only in the authoring container.
 *
 * The .cu file as a whole includes <optix.h> and is unbuildable here; oracle/extract_ptprog_math.py
 * copies the definitions above verbatim into oracle/_ref/ptprog_math.inc at build time (git-ignored,
 * gpurun-ignored, never committed), and this file compiles that text against the reference's own
 * sutil/vec_math.h and the CUDA headers of the image's triton wheel — no stand-in headers.
 *
 * Host-compilation caveat: `abs(N.z)` at :470 must resolve to the float overload, as it
 * does in nvcc device code; on the host a bare `abs` would pick int abs(int).  `using std::abs` below
 * puts the <cmath> float overload into the global namespace.
 */
#include <cuda_runtime.h>
#include <cmath>
#include <cstdlib>
#include <cstddef>
#include <cstdint>
using std::abs;
#include <sutil/vec_math.h>

#include "_ref/ptprog_math.inc"

#define REF_API extern "C" __attribute__((visibility("default")))

static inline float3 ld3(const float* p) { return make_float3(p[0], p[1], p[2]); }
static inline void st3(float* p, const float3& v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

/* OrthonormalBasis(n).inverse_transform(p), n rows */
REF_API void ref_onb_transform(const float* n3, const float* p3, size_t n, float* out3)
{
    for (size_t i = 0; i < n; i++) {
        OrthonormalBasis onb(ld3(n3 + 3 * i));
        float3 p = ld3(p3 + 3 * i);
        onb.inverse_transform(p);
        st3(out3 + 3 * i, p);
    }
}

REF_API void ref_safe_divide(const float* a, const float* b, size_t n, float* out)
{
    for (size_t i = 0; i < n; i++) out[i] = safeDivide(a[i], b[i]);
}
REF_API void ref_safe_divide3(const float* a3, const float* b, size_t n, float* out3)
{
    for (size_t i = 0; i < n; i++) st3(out3 + 3 * i, safeDivide(ld3(a3 + 3 * i), b[i]));
}

/* which: 0 cosine_sample_hemisphere, 1 uniform_sample_hemisphere */
REF_API void ref_sample_hemisphere(int which, const float* u1, const float* u2, size_t n, float* out3)
{
    for (size_t i = 0; i < n; i++) {
        float3 p = make_float3(0.0f);
        if (which == 0) cosine_sample_hemisphere(u1[i], u2[i], p);
        else            uniform_sample_hemisphere(u1[i], u2[i], p);
        st3(out3 + 3 * i, p);
    }
}

REF_API void ref_sample_ggx(const float* u1, const float* u2, const float* roughness, const float* n3, size_t n, float* out3)
{
    for (size_t i = 0; i < n; i++) st3(out3 + 3 * i, sampleGGX(u1[i], u2[i], roughness[i], ld3(n3 + 3 * i)));
}

REF_API void ref_fresnel_conductor(const float* cos_theta, const float* eta3, const float* k3, size_t n, float* out3)
{
    for (size_t i = 0; i < n; i++) st3(out3 + 3 * i, fresnelSchlickConductor(cos_theta[i], ld3(eta3 + 3 * i), ld3(k3 + 3 * i)));
}

REF_API void ref_fr_dielectric(const float* cos_i, const float* eta_i, const float* eta_t, size_t n, float* out)
{
    for (size_t i = 0; i < n; i++) out[i] = FrDielectric(cos_i[i], eta_i[i], eta_t[i]);
}
