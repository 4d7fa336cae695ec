// pt_shading.h — per-lane shading pieces shared by the render kernels: the reference's sampling,
// BSDF and colour code restated for gfx950 (plain fp32; citations inline).  Every function takes the arithmetic level FM of
// pt_device.h: 0 = IEEE operations without contraction, the oracle's level; 2 = the reference build's fast-math kind.
#pragma once
#include "pt_device.h"

namespace ptd {

// sutil/WorkDistribution.h:60-81
__device__ __forceinline__ void sample_pixel(int num_gpus, int width, int gpu_idx, int sample_idx, int& px, int& py)
{
    const int TILE_WIDTH = 8, TILE_HEIGHT = 4;
    const int tile_strip_width = TILE_WIDTH * num_gpus;
    const int num_tile_strip_cols = width / tile_strip_width + (width % tile_strip_width == 0 ? 0 : 1);
    const int tile_strip_idx = sample_idx / (TILE_WIDTH * TILE_HEIGHT);
    const int tile_strip_y = tile_strip_idx / num_tile_strip_cols;
    const int tile_strip_x = tile_strip_idx - tile_strip_y * num_tile_strip_cols;
    const int tile_pixel_idx = sample_idx - tile_strip_idx * (TILE_WIDTH * TILE_HEIGHT);
    const int tile_pixel_y = tile_pixel_idx / TILE_WIDTH;
    const int tile_pixel_x = tile_pixel_idx - tile_pixel_y * TILE_WIDTH;
    const int tile_offset_x = (gpu_idx + tile_strip_y % num_gpus) % num_gpus * TILE_WIDTH;
    py = tile_strip_y * TILE_HEIGHT + tile_pixel_y;
    px = tile_strip_x * tile_strip_width + tile_pixel_x + tile_offset_x;
}

// ---- sampling / BSDF pieces, restated from pathTracerPrograms.cu ---------------------
// OrthonormalBasis :54-85
template <int FM = 0>
__device__ __forceinline__ void onb_transform(const f3& n, f3& p)
{
    f3 bn;
    if (fabsf(n.x) > fabsf(n.z)) bn = mk(-n.y, n.x, 0.0f);
    else                         bn = mk(0.0f, -n.z, n.y);
    bn = m_normalize<FM>(bn);
    const f3 tg = m_cross<FM>(bn, n);
    p = FM >= 2 ? m_madd<FM>(n, p.z, m_madd<FM>(bn, p.y, p.x * tg)) : p.x * tg + p.y * bn + p.z * n;
}
// cosine_sample_hemisphere :341-353.  sincosf shares one argument reduction between the sine and the cosine of an
// angle; OCML's sinf / cosf are that same reduction + kernel with one output selected, so the values are the ones
// sinf(x) and cosf(x) return (checked bit for bit over the argument ranges: pt_selftest op 11, test_gpu_golden.py)
__device__ __forceinline__ f3 cosine_sample_hemisphere(float eta1, float eta2)
{
    const float theta = acosf(sqrtf(eta1));
    const float phi = 2.0f * kPIf * eta2;
    float st, ct, sp, cp;
    sincosf(theta, &st, &ct);
    sincosf(phi, &sp, &cp);
    return mk(st * cp, st * sp, ct);
}
// the same sampler with v_sin_f32 / v_cos_f32 and sqrt(1 - z1) for sin(acos(sqrt(z1))): arithmetic levels 1 and 2 (pt_device.h)
template <int FM>
__device__ __forceinline__ f3 cosine_sample_hemisphere_fast(float eta1, float eta2)
{
    const float ct = m_sqrt<FM>(eta1), stt = m_sqrt<FM>(1.0f - eta1);
    return mk(stt * __builtin_amdgcn_cosf(eta2), stt * __builtin_amdgcn_sinf(eta2), ct);
}
// uniform_sample_hemisphere :368-380 (the theta computed at :372 is unused there)
template <int FM = 0>
__device__ __forceinline__ f3 uniform_sample_hemisphere(float u1, float u2)
{
    float sp, cp;
    m_sincos_2pi<FM>(u2, sp, cp);                                    // phi = 2 pi u2
    return mk(cp * m_sqrt<FM>(1 - u1 * u1), sp * m_sqrt<FM>(1 - u1 * u1), u1);
}
// sampleGGX :455-476 (roughness is the literal 0.2 of :880)
template <int FM = 0>
__device__ __forceinline__ f3 sample_ggx(float u1, float u2, float roughness, const f3& N)
{
    const float cosTheta = m_sqrt<FM>(m_div<FM>(1.0f - u2, 1.0f + (roughness * roughness - 1.0f) * u2));
    const float sinTheta = m_sqrt<FM>(1.0f - cosTheta * cosTheta);
    float sp, cp;
    m_sincos_2pi<FM>(u1, sp, cp);            // phi = 2 pi u1: the values of sinf(phi), cosf(phi) with one argument reduction (see shade_hit)
    const f3 H = mk(sinTheta * cp, sinTheta * sp, cosTheta);
    // :470 compares in double against 0.999; 0.999f rounds up, so the float test is identical
    const f3 up = fabsf(N.z) < 0.999f ? mk(0.0f, 0.0f, 1.0f) : mk(1.0f, 0.0f, 0.0f);
    const f3 tangent = m_normalize<FM>(cross(up, N));
    const f3 bitangent = cross(N, tangent);
    return m_normalize<FM>(H.x * tangent + H.y * bitangent + H.z * N);
}
// fresnelSchlickConductor :494-510
template <int FM = 0>
__device__ __forceinline__ f3 fresnel_conductor(float cosTheta, const f3& eta, const f3& k)
{
    const f3 eta2 = eta * eta, k2 = k * k;
    const f3 c2 = mk(cosTheta * cosTheta);
    const f3 t1 = eta2 - k2 - c2;
    const f3 a2plusb2 = mk(m_sqrt<FM>(t1.x * t1.x + 4 * eta2.x * k2.x), m_sqrt<FM>(t1.y * t1.y + 4 * eta2.y * k2.y),
                           m_sqrt<FM>(t1.z * t1.z + 4 * eta2.z * k2.z));
    const f3 t2 = a2plusb2 + c2;
    const f3 rs_n = t2 - 2 * eta * cosTheta + c2, rs_d = t2 + 2 * eta * cosTheta + c2;
    const f3 Rs = mk(m_div<FM>(rs_n.x, rs_d.x), m_div<FM>(rs_n.y, rs_d.y), m_div<FM>(rs_n.z, rs_d.z));
    const f3 rp_n = Rs * (t2 - 2 * eta * cosTheta + mk(1.0f)), rp_d = t2 + 2 * eta * cosTheta + mk(1.0f);
    const f3 Rp = mk(m_div<FM>(rp_n.x, rp_d.x), m_div<FM>(rp_n.y, rp_d.y), m_div<FM>(rp_n.z, rp_d.z));
    return (Rs + Rp) * 0.5f;
}
// FrDielectric :534-559
template <int FM = 0>
__device__ __forceinline__ float fr_dielectric(float cosThetaI, float etaI, float etaT)
{
    cosThetaI = clampf(cosThetaI, -1.0f, 1.0f);
    if (!(cosThetaI > 0.0f)) { const float t = etaI; etaI = etaT; etaT = t; cosThetaI = fabsf(cosThetaI); }
    const float sinThetaI = m_sqrt<FM>(fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI));
    const float sinThetaT = m_div<FM>(etaI, etaT) * sinThetaI;
    if (sinThetaT >= 1.0f) return 1.0f;
    const float cosThetaT = m_sqrt<FM>(fmaxf(0.0f, 1.0f - sinThetaT * sinThetaT));
    const float rParl = m_div<FM>((etaT * cosThetaI) - (etaI * cosThetaT), (etaT * cosThetaI) + (etaI * cosThetaT));
    const float rPerp = m_div<FM>((etaI * cosThetaI) - (etaT * cosThetaT), (etaI * cosThetaI) + (etaT * cosThetaT));
    return (rParl * rParl + rPerp * rPerp) / 2.0f;
}
// refract, cuda/helpers.h:107-137
template <int FM = 0>
__device__ __forceinline__ bool refract_dir(f3& r, const f3& i, const f3& n, float ior)
{
    f3 nn = n;
    float negNdotV = dot(i, nn);
    float eta;
    if (negNdotV > 0.0f) { eta = ior; nn = -n; negNdotV = -negNdotV; }
    else                 { eta = m_div<FM>(1.f, ior); }
    const float k = 1.f - eta * eta * (1.f - negNdotV * negNdotV);
    if (k < 0.0f) { r = mk(0.f); return false; }
    r = m_normalize<FM>(eta * i - (eta * negNdotV + m_sqrt<FM>(k)) * nn);
    return true;
}
template <int FM = 0>
__device__ __forceinline__ float safe_div(float a, float b) { return b == 0.0f ? 0.0f : m_div<FM>(a, b); }
// What a finished segment adds to the pixel (raygen :761-762: result += radiance * attenuation) and the roulette's survival
// probability (:766: the throughput's luminance); multiply-adds fused at the fast level
template <int FM = 0>
__device__ __forceinline__ void add_segment(f3& result, const f3& radiance, const f3& att)
{
    if (FM >= 2) result = m_madd<FM>(radiance, att, result);
    else result += radiance * att;
}
template <int FM = 0>
__device__ __forceinline__ float roulette_p(const f3& att) { return m_dot<FM>(att, mk(0.30f, 0.59f, 0.11f)); }
// Throughput of a path that survives the roulette (:771-777: each component safeDivide'd by the survival probability); at the
// fast arithmetic level the three quotients share one reciprocal
template <int FM = 0>
__device__ __forceinline__ f3 roulette_scale(const f3& att, float p)
{
    if (FM >= 2) { const float ip = p == 0.0f ? 0.0f : __builtin_amdgcn_rcpf(p); return att * ip; }
    return mk(safe_div(att.x, p), safe_div(att.y, p), safe_div(att.z, p));
}

// make_color, cuda/helpers.h:35-62
__device__ __forceinline__ float to_srgb1(float c)
{
    const float invGamma = 1.0f / 2.4f;
    const float powed = powf(c, invGamma);
    return c < 0.0031308f ? 12.92f * c : 1.055f * powed - 0.055f;
}
__device__ __forceinline__ uint32_t quantize8(float x)
{
    x = clampf(x, 0.0f, 1.0f);
    const uint32_t v = (uint32_t)(x * 256.0f);
    return v < 255u ? v : 255u;
}
__device__ __forceinline__ uint32_t make_color(const f3& c)
{
    const uint32_t r = quantize8(to_srgb1(clampf(c.x, 0.0f, 1.0f)));
    const uint32_t g = quantize8(to_srgb1(clampf(c.y, 0.0f, 1.0f)));
    const uint32_t b = quantize8(to_srgb1(clampf(c.z, 0.0f, 1.0f)));
    return r | (g << 8) | (b << 16) | (255u << 24);
}

__device__ __forceinline__ uint32_t xcc_id()
{
    // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4); only used as an affinity hint
    return (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
}


// What the shading needs of the launch constants (toggles, area light) is read where it is used, through `late()`: a
// callable that returns the kernel's RenderArgs.  The persistent kernel passes one that re-reads the kernel-argument
// segment at that point (render_megakernel.hip, RenderArgsBox), so that these ~20 values are never held in scalar
// registers across the rest of the kernel; a plain `return A` keeps the compiler's usual hoisting.

// What a closest-hit leaves behind for raygen (RadiancePayloadRayData, pathTracer.h:19-32), minus
// what is consumed on the spot.
struct Pending {
    f3 nxt_org, nxt_dir, radiance;
    float weight;     // NEE contribution factor if the shadow ray is unoccluded
    bool done;
    // cosines of the shadow ray and of the next bounce against the plane of the triangle that was hit (0: unknown / not a ray
    // that starts ON that triangle): what the origin-triangle release of the persistent kernel decides by
    float cos_shadow = 0.0f, cos_bounce = 0.0f;
};

// __closesthit__diffuse__ch, pathTracerPrograms.cu:866-1031, for one lane.  Returns true when a
// shadow ray (P, L, 0.01, Ldist - 0.01) has to be traced before the segment can be accounted.
// FM: arithmetic level (pt_device.h): 0 IEEE, 1 the cosine sampler's trigonometry in hardware, 2 the arithmetic of the reference's
// own build (nvcc --use_fast_math, CMakeLists.txt:267).  Levels 1 and 2 give different low bits than level 0, the same image
// within the parity tolerance (test_fast_math_variant).
// FROM_RECORD: normal and material from the triangle record itself (`slot` indexes sc.tris; the four-wide experiment, whose
// slots are positions in its own record array) instead of the builder's shading record
template <int FM = 0, bool FROM_RECORD = false, typename Late>
__device__ __forceinline__ bool shade_hit(const DeviceScene& sc, Late late, const f3& org, const f3& dir,
                                          float t_hit, int slot, int depth, uint32_t& pseed, f3& att, f3& emission,
                                          Pending& pd, f3& P, f3& L, float& Ldist)
{
    float4 sr;                                                                   // :890 N_0 = normalize(cross(v1 - v0, v2 - v0)), material id
    if (FROM_RECORD) {
        const TriRecord* tp = sc.tris + slot;
        const float4 r0 = tp->r0, r1 = tp->r1, r2 = tp->r2;
        const f3 n0 = normalize(cross(mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x)));
        sr = make_float4(n0.x, n0.y, n0.z, r2.z);
    } else {
        sr = sc.shade[slot];                                                     // computed once by the builder (k_gather_leaves)
    }
    // the builder's record names the material, its bsdfType and whether it emits: {diffuse, ior} is the one fetch every hit
    // needs, the emission is fetched for emitters only (zero otherwise: what the fetch would return)
    const uint32_t mw = __float_as_uint(sr.w);
    const DevMaterial* mp = sc.mats + (mw & kShadeMatMask);
    const float4 m0 = mp->kd_ior;
    const f3 Kd = mk(m0.x, m0.y, m0.z);
    const float IOR = m0.w;
    f3 Ke = mk(0.0f);
    int bsdf = (int)((mw >> kShadeBsdfShift) & 3u);
    if (FROM_RECORD || (mw & kShadeHasKe) != 0u) {       // FROM_RECORD: a plain material id from the triangle record
        const float4 m1 = mp->ke_bsdf;
        Ke = mk(m1.x, m1.y, m1.z);
        if (FROM_RECORD) bsdf = (int)__float_as_uint(m1.w);
    }
    const f3 N0 = mk(sr.x, sr.y, sr.z);
    const f3 N = FM >= 2 ? N0 * copysignf(1.0f, -m_dot<FM>(dir, N0)) : faceforward(N0, -dir, N0);
    P = FM >= 2 ? m_madd<FM>(dir, t_hit, org) : org + t_hit * dir;              // :894
    emission = depth == 0 ? Ke : mk(0.0f);                                       // :898-901
    uint32_t s = pseed;
    pd.nxt_org = org; pd.nxt_dir = dir;
    if (bsdf == PT_BSDF_DIFFUSE) {                                               // :907-930
        const float z1 = rnd(s);
        const float z2 = rnd(s);
        f3 w_in;
        if (late().useIS) w_in = FM >= 1 ? cosine_sample_hemisphere_fast<FM>(z1, z2) : cosine_sample_hemisphere(z1, z2);
        else         w_in = uniform_sample_hemisphere<FM>(z1, z2);
        pd.cos_bounce = w_in.z;                                                  // cosine to the (flipped) geometric normal, before the change of basis
        onb_transform<FM>(N, w_in);
        pd.nxt_dir = w_in;
        pd.nxt_org = P;
        att *= Kd;
    } else if (bsdf == PT_BSDF_METALLIC) {                                       // :931-953
        const float z1 = rnd(s);
        const float z2 = rnd(s);
        const f3 mn = sample_ggx<FM>(z1, z2, 0.2f, N);
        const f3 R = reflect(dir, mn);
        pd.nxt_dir = R;
        pd.nxt_org = P + R * 1e-4f;
        const f3 eta = mk(1.45f, 0.7f, 1.55f), kk = mk(3.0f, 2.2f, 3.5f);
        const float cosTheta = fmaxf(dot(mn, -dir), 0.0f);
        const f3 F = fresnel_conductor<FM>(cosTheta, eta, kk);
        att *= F * Kd;
    } else if (bsdf == PT_BSDF_REFRACTION) {                                     // :954-982
        const f3 inc = m_normalize<FM>(dir);
        const float cos_theta = dot(m_normalize<FM>(-dir), N0);
        const float F = fr_dielectric<FM>(cos_theta, 1.0f, IOR);
        if (rnd(s) < F) {
            pd.nxt_dir = reflect(inc, N0);
        } else {
            f3 rd;
            pd.nxt_dir = refract_dir<FM>(rd, inc, N0, IOR) ? rd : reflect(inc, N0);
        }
        pd.nxt_org = P + pd.nxt_dir * 1e-3f;
        att *= Kd;
    }
    const float z1 = rnd(s);                                                     // :985-987
    const float z2 = rnd(s);
    pseed = s;
    // :992-1000 tests length(Ke) > 0: sqrt(s) > 0 exactly when s > 0 (s = +0, a denormal, inf and NaN included), so the
    // correctly rounded square root of the reference is not needed for the decision
    if (dot(Ke, Ke) > 0.0f) { pd.radiance = Ke; pd.done = true; }
    else                   { pd.radiance = mk(0.0f); pd.done = false; }
    pd.weight = 0.0f;
    bool want_shadow = false;
    const auto& La = late();
    if (La.useDL && bsdf != PT_BSDF_REFRACTION) {                                // :1003-1026
        const f3 light_pos = FM >= 2 ? m_madd<FM>(mk(La.light.v2), z2, m_madd<FM>(mk(La.light.v1), z1, mk(La.light.corner)))
                                     : mk(La.light.corner) + mk(La.light.v1) * z1 + mk(La.light.v2) * z2;
        Ldist = m_length<FM>(light_pos - P);
        L = m_normalize<FM>(light_pos - P);
        const float nDl = m_dot<FM>(N, L);
        const float LnDl = -m_dot<FM>(mk(La.light.normal), L);
        want_shadow = nDl > 0.0f && LnDl > 0.0f;
        pd.cos_shadow = nDl;
        pd.weight = m_div<FM>(nDl * LnDl * La.light_area, kPIf * Ldist * Ldist);         // :1021-1022 (|v1 x v2| from the host), used only if unoccluded
    }
    return want_shadow;
}

// ---- light mode 1 (pt_set_light_mode; NOT the reference's estimator) ----------------------
// Emissive triangles of the scene as the area light, next-event estimation and BSDF-sampled emitter hits combined with
// the power heuristic, uniform hemisphere sampling with its 2 cos weight, emitters without the Kd quirks.  Same draws in
// the same order as mode 0.  Operation for operation the twin of closesthit_scene_lights() in oracle/oracle_pt.cpp.
// Out: pd.radiance = what this segment adds to the pixel, already times the throughput — at once for an emitter hit
// (pd.done), or, when the function returns true, only if the shadow ray (P, L, 0.01, Ldist - 0.01) is unoccluded.
// att becomes the throughput of the continuation; prev_pdf the solid-angle pdf of the sampled direction where a
// light sample was taken (0 elsewhere: a later emitter hit then counts in full).
template <int FM = 0, typename Late>
__device__ __forceinline__ bool shade_hit_lights(const DeviceScene& sc, Late late, const f3& org, const f3& dir,
                                                 float t_hit, int slot, int depth, uint32_t& pseed, f3& att, float& prev_pdf,
                                                 Pending& pd, f3& P, f3& L, float& Ldist)
{
    const float4 sr = sc.shade[slot];
    const uint32_t mw = __float_as_uint(sr.w);
    const DevMaterial* mp = sc.mats + (mw & kShadeMatMask);
    const float4 m0 = mp->kd_ior;
    const f3 Kd = mk(m0.x, m0.y, m0.z);
    const float ior = m0.w;
    f3 Ke = mk(0.0f);
    if ((mw & kShadeHasKe) != 0u) { const float4 m1 = mp->ke_bsdf; Ke = mk(m1.x, m1.y, m1.z); }
    const int bsdf = (int)((mw >> kShadeBsdfShift) & 3u);
    const f3 N0 = mk(sr.x, sr.y, sr.z);
    const f3 N = faceforward(N0, -dir, N0);
    P = org + t_hit * dir;
    const auto& La = late();
    const bool useDL = La.useDL != 0u && sc.n_lights != 0u;
    const bool useIS = La.useIS != 0u;
    const float area_total = sc.light_area;
    uint32_t s = pseed;
    pd.radiance = mk(0.0f); pd.weight = 0.0f;
    pd.nxt_org = org; pd.nxt_dir = dir;
    if (dot(Ke, Ke) > 0.0f) {                         // length(Ke) > 0, without the square root (see shade_hit)
        float w = 1.0f;
        if (depth > 0 && prev_pdf > 0.0f) {
            const float cos_l = fabsf(dot(N0, dir));
            const float p_l = m_div<FM>(t_hit * t_hit, area_total * cos_l);
            w = cos_l > 0.0f ? m_div<FM>(prev_pdf * prev_pdf, prev_pdf * prev_pdf + p_l * p_l) : 1.0f;
        }
        pd.radiance = att * Ke * w;
        if (bsdf == PT_BSDF_REFRACTION) (void)rnd(s); else { (void)rnd(s); (void)rnd(s); }
        (void)rnd(s); (void)rnd(s);
        pseed = s;
        pd.done = true;
        return false;
    }
    pd.done = false;
    const f3 att_in = att;
    float bsdf_pdf = 0.0f;
    if (bsdf == PT_BSDF_DIFFUSE) {
        const float z1 = rnd(s);
        const float z2 = rnd(s);
        f3 w_in = useIS ? (FM >= 1 ? cosine_sample_hemisphere_fast<FM>(z1, z2) : cosine_sample_hemisphere(z1, z2)) : uniform_sample_hemisphere<FM>(z1, z2);
        const float cos_out = w_in.z;
        onb_transform<FM>(N, w_in);
        pd.nxt_dir = w_in;
        pd.nxt_org = P;
        if (useIS) { att = att_in * Kd; bsdf_pdf = FM >= 2 ? cos_out * (1.0f / kPIf) : cos_out / kPIf; }
        else       { att = att_in * Kd * (2.0f * cos_out); bsdf_pdf = 1.0f / (2.0f * kPIf); }
    } else if (bsdf == PT_BSDF_METALLIC) {
        const float z1 = rnd(s);
        const float z2 = rnd(s);
        const f3 mn = sample_ggx<FM>(z1, z2, 0.2f, N);
        const f3 R = reflect(dir, mn);
        pd.nxt_dir = R;
        pd.nxt_org = P + R * 1e-4f;
        const f3 eta = mk(1.45f, 0.7f, 1.55f), kk = mk(3.0f, 2.2f, 3.5f);
        const float cosTheta = fmaxf(dot(mn, -dir), 0.0f);
        att = att_in * (fresnel_conductor<FM>(cosTheta, eta, kk) * Kd);
    } else if (bsdf == PT_BSDF_REFRACTION) {
        const f3 inc = m_normalize<FM>(dir);
        const float cos_theta = dot(m_normalize<FM>(-dir), N0);
        const float F = fr_dielectric<FM>(cos_theta, 1.0f, ior);
        if (rnd(s) < F) {
            pd.nxt_dir = reflect(inc, N0);
        } else {
            f3 rd;
            pd.nxt_dir = refract_dir<FM>(rd, inc, N0, ior) ? rd : reflect(inc, N0);
        }
        pd.nxt_org = P + pd.nxt_dir * 1e-3f;
        att = att_in * Kd;
    }
    const float z1 = rnd(s);
    const float z2 = rnd(s);
    pseed = s;
    prev_pdf = 0.0f;
    bool want_shadow = false;
    if (useDL && bsdf == PT_BSDF_DIFFUSE) {
        const float target = z1 * area_total;
        uint32_t k = 0;
        while (k + 1u < sc.n_lights && !(target < sc.lights[5u * k + 1u].w)) k++;
        const float4 l0 = sc.lights[5u * k], l1 = sc.lights[5u * k + 1u], l2 = sc.lights[5u * k + 2u], l3 = sc.lights[5u * k + 3u], l4 = sc.lights[5u * k + 4u];
        const float lo = k ? sc.lights[5u * (k - 1u) + 1u].w : 0.0f;
        const float u = fminf(fmaxf(m_div<FM>(target - lo, l0.w), 0.0f), 0.99999994f);
        const float su = m_sqrt<FM>(u);
        const f3 light_pos = mk(l0.x, l0.y, l0.z) + mk(l1.x, l1.y, l1.z) * (su * (1.0f - z2)) + mk(l2.x, l2.y, l2.z) * (su * z2);
        const f3 Lv = light_pos - P;
        const float dist2 = dot(Lv, Lv);
        Ldist = m_sqrt<FM>(dist2);
        L = FM >= 2 ? Lv * __builtin_amdgcn_rsqf(dist2) : Lv / Ldist;
        const float nDl = dot(N, L);
        const float LnDl = fabsf(dot(mk(l3.x, l3.y, l3.z), L));
        prev_pdf = bsdf_pdf;
        want_shadow = nDl > 0.0f && LnDl > 0.0f;
        const float p_l = m_div<FM>(dist2, area_total * LnDl);
        const float p_b = useIS ? (FM >= 2 ? nDl * (1.0f / kPIf) : nDl / kPIf) : 1.0f / (2.0f * kPIf);
        const float w = m_div<FM>(p_l * p_l, p_l * p_l + p_b * p_b);
        const float geom = m_div<FM>(nDl * LnDl * area_total, kPIf * dist2);
        if (want_shadow) pd.radiance = att_in * Kd * mk(l4.x, l4.y, l4.z) * (geom * w);      // counted only if the shadow ray finds nothing
    }
    return want_shadow;
}

}  // namespace ptd
