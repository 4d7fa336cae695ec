// selftest.hip — evaluates the device-side building blocks of the render kernel (the very functions
// pt_device.h / pt_shading.h inline into it) on caller-supplied inputs, so that tests can hold the GPU
// implementations against the golden vectors generated from the reference's own host-compilable sources
// (tests/golden/reference_vectors.npz: cuda/random.h, cuda/helpers.h, sutil/vec_math.h,
// sutil/WorkDistribution.h).  Not on the render path.
#include "pt_device.h"
#include "pt_shading.h"
#include "selftest.h"

namespace ptd {

// op: see pt_selftest in include/acgpt.h
__global__ void k_selftest(int op, const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (op == 1) {                                   // one LCG stream: in = {seed, count}; out = states[count], values[count]
        if (i != 0) return;
        uint32_t seed = in[0];
        const uint32_t count = in[1];
        for (uint32_t k = 0; k < count; k++) { const float v = rnd(seed); out[k] = seed; out[count + k] = __float_as_uint(v); }
        return;
    }
    if (i >= n) return;
    const float* fin = (const float*)in;
    float* fout = (float*)out;
    if (op == 0) { out[i] = tea4(in[2 * i], in[2 * i + 1]); return; }
    if (op == 2) { out[i] = make_color(mk(fin[3 * i], fin[3 * i + 1], fin[3 * i + 2])); return; }
    if (op >= 3 && op <= 8) {                        // in: a xyz, b xyz, c xyz, s
        const float* r = fin + 10 * i;
        const f3 a = mk(r[0], r[1], r[2]), b = mk(r[3], r[4], r[5]), c = mk(r[6], r[7], r[8]);
        const float s = r[9];
        f3 o;
        switch (op) {
            case 3: o = normalize(a); break;
            case 4: o = reflect(a, b); break;
            case 5: o = faceforward(a, b, c); break;
            case 6: o = lerp3(a, b, s); break;
            case 7: o = cross(a, b); break;
            default: o = a / s; break;
        }
        fout[3 * i] = o.x; fout[3 * i + 1] = o.y; fout[3 * i + 2] = o.z;
        return;
    }
    if (op == 9) {                                   // in: i xyz, n xyz, ior; out: r xyz, ok
        const float* r = fin + 7 * i;
        f3 o = mk(0.0f);
        const bool ok = refract_dir(o, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), r[6]);
        fout[4 * i] = o.x; fout[4 * i + 1] = o.y; fout[4 * i + 2] = o.z; out[4 * i + 3] = ok ? 1u : 0u;
        return;
    }
    if (op == 11) {                                  // in: x; out: sinf(x), cosf(x), and the pair sincosf(x) returns
        const float x = fin[i];
        float sv, cv;
        sincosf(x, &sv, &cv);
        fout[4 * i] = sinf(x); fout[4 * i + 1] = cosf(x); fout[4 * i + 2] = sv; fout[4 * i + 3] = cv;
        return;
    }
    if (op >= 12 && op <= 18) {                      // pathTracerPrograms.cu helpers (see acgpt.h for the record layouts)
        const int in_w = (op == 12 || op == 16) ? 6 : op == 13 ? 4 : op == 17 ? 7 : op == 18 ? 3 : 2;
        const float* r = fin + in_w * i;
        f3 o = mk(0.0f);
        switch (op) {
            case 12: o = mk(r[3], r[4], r[5]); onb_transform(mk(r[0], r[1], r[2]), o); break;
            case 13: o = mk(safe_div(r[0], r[3]), safe_div(r[1], r[3]), safe_div(r[2], r[3])); break;
            case 14: o = cosine_sample_hemisphere(r[0], r[1]); break;
            case 15: o = uniform_sample_hemisphere(r[0], r[1]); break;
            case 16: o = sample_ggx(r[0], r[1], r[2], mk(r[3], r[4], r[5])); break;
            case 17: o = fresnel_conductor(r[0], mk(r[1], r[2], r[3]), mk(r[4], r[5], r[6])); break;
            default: fout[i] = fr_dielectric(r[0], r[1], r[2]); return;
        }
        fout[3 * i] = o.x; fout[3 * i + 1] = o.y; fout[3 * i + 2] = o.z;
        return;
    }
    if (op >= 32 && op <= 38) {                      // ops 12..18 at the arithmetic level of PT_MATH_FAST (pt_device.h, level 2): same records
        const int base = op - 20;
        const int in_w = (base == 12 || base == 16) ? 6 : base == 13 ? 4 : base == 17 ? 7 : base == 18 ? 3 : 2;
        const float* r = fin + in_w * i;
        f3 o = mk(0.0f);
        switch (base) {
            case 12: o = mk(r[3], r[4], r[5]); onb_transform<2>(mk(r[0], r[1], r[2]), o); break;
            case 13: o = roulette_scale<2>(mk(r[0], r[1], r[2]), r[3]); break;
            case 14: o = cosine_sample_hemisphere_fast<2>(r[0], r[1]); break;
            case 15: o = uniform_sample_hemisphere<2>(r[0], r[1]); break;
            case 16: o = sample_ggx<2>(r[0], r[1], r[2], mk(r[3], r[4], r[5])); break;
            case 17: o = fresnel_conductor<2>(r[0], mk(r[1], r[2], r[3]), mk(r[4], r[5], r[6])); break;
            default: fout[i] = fr_dielectric<2>(r[0], r[1], r[2]); return;
        }
        fout[3 * i] = o.x; fout[3 * i + 1] = o.y; fout[3 * i + 2] = o.z;
        return;
    }
    if (op == 30) {                                  // the primitives of PT_MATH_FAST: in a, b; out a / b, sqrt(|a|), normalize((a, b, 1)).x, .y
        const float a = fin[2 * i], b = fin[2 * i + 1];
        const f3 nv = m_normalize<2>(mk(a, b, 1.0f));
        fout[4 * i] = m_div<2>(a, b); fout[4 * i + 1] = m_sqrt<2>(fabsf(a)); fout[4 * i + 2] = nv.x; fout[4 * i + 3] = nv.y;
        return;
    }
    if (op == 31) {                                  // in u in [0, 1); out sin(2 pi u), cos(2 pi u) at level 2 (v_sin_f32 / v_cos_f32), then at level 0
        float sv, cv, s0, c0;
        m_sincos_2pi<2>(fin[i], sv, cv);
        m_sincos_2pi<0>(fin[i], s0, c0);
        fout[4 * i] = sv; fout[4 * i + 1] = cv; fout[4 * i + 2] = s0; fout[4 * i + 3] = c0;
        return;
    }
    if (op == 19) {
        // the default kernel's box test end to end: in = ray o xyz, d xyz, box lo xyz, hi xyz (fp32, as the builder holds it),
        // scene centre xyz, inv_scale, tmax; out = accepted (n <= min(f, tmax)), n, f.  The box goes through
        // pack_planes() (outward to fp16), the ray through setup_ray_h9(), the test is slab_h9() — what k_render_pw executes.
        const float* r = fin + 17 * i;
        HSpace hs; hs.cx = r[12]; hs.cy = r[13]; hs.cz = r[14]; hs.inv_scale = r[15];
        const float scale = 1.0f / hs.inv_scale;
        float gl, gh;
        const uint32_t px = pack_planes(r[6], r[9], hs.cx, scale, gl, gh), py = pack_planes(r[7], r[10], hs.cy, scale, gl, gh),
                       pz = pack_planes(r[8], r[11], hs.cz, scale, gl, gh);
        f3 mul, add;
        setup_ray_h9(mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), hs, mul, add);
        float tn, tf;
        slab_h9(px, py, pz, mul, add, 0.01f, tn, tf);
        out[3 * i] = tn <= fminf(tf, r[16]) ? 1u : 0u; fout[3 * i + 1] = tn; fout[3 * i + 2] = tf;
        return;
    }
    if (op == 40) {
        // op 19's records through the fp16 centre / half-extent form (NODE_FMT 11): pack_centre_half, setup_ray_hc, slab_hc
        const float* r = fin + 17 * i;
        // (one scale on every axis here; op 41 takes a scale per axis, what the builder uses)
        HSpace hs; hs.cx = r[12]; hs.cy = r[13]; hs.cz = r[14]; hs.inv_scale = r[15]; hs.isx = hs.isy = hs.isz = r[15]; hs.pad_ = 0.0f;
        const float scale = 1.0f / hs.inv_scale;
        const uint32_t px = pack_centre_half(r[6], r[9], hs.cx, scale), py = pack_centre_half(r[7], r[10], hs.cy, scale), pz = pack_centre_half(r[8], r[11], hs.cz, scale);
        f3 mul, add;
        setup_ray_hc(mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), hs, mul, add);
        float tn, tf;
        slab_hc(px, py, pz, mul, add, 0.01f, tn, tf);
        out[3 * i] = tn <= fminf(tf, r[16]) ? 1u : 0u; fout[3 * i + 1] = tn; fout[3 * i + 2] = tf;
        return;
    }
    if (op == 41) {
        // op 40 with a scale per axis: in = ray o xyz, d xyz, box lo xyz, hi xyz, scene centre xyz, inv_scale xyz, tmax (19 floats)
        const float* r = fin + 19 * i;
        HSpace hs; hs.cx = r[12]; hs.cy = r[13]; hs.cz = r[14]; hs.inv_scale = r[15]; hs.isx = r[15]; hs.isy = r[16]; hs.isz = r[17]; hs.pad_ = 0.0f;
        const uint32_t px = pack_centre_half(r[6], r[9], hs.cx, 1.0f / hs.isx), py = pack_centre_half(r[7], r[10], hs.cy, 1.0f / hs.isy), pz = pack_centre_half(r[8], r[11], hs.cz, 1.0f / hs.isz);
        f3 mul, add;
        setup_ray_hc(mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), hs, mul, add);
        float tn, tf;
        slab_hc(px, py, pz, mul, add, 0.01f, tn, tf);
        out[3 * i] = tn <= fminf(tf, r[18]) ? 1u : 0u; fout[3 * i + 1] = tn; fout[3 * i + 2] = tf;
        return;
    }
    if (op == 39) {
        // the shared-plane kernel's box test end to end (NODE_FMT 10): in = ray o xyz, d xyz, box lo xyz, hi xyz (fp32, as the builder
        // holds it), root planes L xyz, H xyz, inv_scale, tmax.  The box is once child 0 and once child 1 of a node whose other child
        // is the root box itself (so the box owns all six new planes and the sibling inherits all six): s_new_plane / pack_magnitude,
        // setup_ray_s, the root interval from the per-ray constants, slab_s — what k_render_pw<..., 10, ...> executes — and the
        // interval's trip over the stack (pack_interval / unpack_interval).  out = accepted as child 0, as child 1, root accepted | sibling accepted << 1
        const float* r = fin + 20 * i;
        SSpace sp; sp.lx = r[12]; sp.ly = r[13]; sp.lz = r[14]; sp.hx = r[15]; sp.hy = r[16]; sp.hz = r[17]; sp.inv_scale = r[18];
        const float scale = 1.0f / sp.inv_scale, tmax = r[19];
        f3 mm, an, af;
        setup_ray_s(mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), sp, mm, an, af);
        float tn = fmaxf(fmaxf(an.x, an.y), fmaxf(an.z, 0.01f)), tf = fminf(fminf(af.x, af.y), fminf(af.z, tmax));
        const bool root_ok = tn <= tf * kFarWiden;
        { float a, b; unpack_interval(pack_interval(fmaxf(tn, 0.0f), fmaxf(tf, 0.0f)), a, b); if (root_ok) { tn = a; tf = b; } }      // as if it had waited on the stack
        uint32_t acc[2]; bool sib[2];
#pragma unroll
        for (int as1 = 0; as1 < 2; as1++) {
            uint32_t w[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float dl = r[6 + k] - r[12 + k], dh = r[15 + k] - r[9 + k];
                w[k] = as1 ? (s_new_plane(0.0f, dl, scale) | (s_new_plane(0.0f, dh, scale) << 16)) : (s_new_plane(dl, 0.0f, scale) | (s_new_plane(dh, 0.0f, scale) << 16));
            }
            float n0, f0, n1, f1;
            slab_s(w[0], w[1], w[2], mm, an, af, tn, tf, n0, f0, n1, f1);
            const bool a0 = n0 <= f0 * kFarWiden, a1 = n1 <= f1 * kFarWiden;
            acc[as1] = root_ok && (as1 ? a1 : a0) ? 1u : 0u;
            sib[as1] = root_ok && (as1 ? a0 : a1);
        }
        out[3 * i] = acc[0]; out[3 * i + 1] = acc[1]; out[3 * i + 2] = (root_ok ? 1u : 0u) | (sib[0] ? 2u : 0u) | (sib[1] ? 4u : 0u);
        return;
    }
    if (op == 10) {                                  // in: world, width, rank, sample; out: x, y
        const int* r = (const int*)in + 4 * i;
        int x, y;
        sample_pixel(r[0], r[1], r[2], r[3], x, y);
        ((int*)out)[2 * i] = x; ((int*)out)[2 * i + 1] = y;
        return;
    }
}

hipError_t launch_selftest(int op, const uint32_t* d_in, uint32_t n, uint32_t* d_out, hipStream_t stream)
{
    k_selftest<<<(n + 255) / 256, 256, 0, stream>>>(op, d_in, n, d_out);
    return hipGetLastError();
}

}  // namespace ptd
