// render_megakernel.hip — the per-pixel Monte-Carlo launch as ONE persistent gfx950 kernel.
//
// Replaces __raygen__rg / __closesthit__diffuse__ch / __miss__ms
// (PathTracer_Optix/pathTracerPrograms.cu:707-816, 866-1031, 833-847) and the OptiX
// traversal underneath them (:600-613, :660-671).
//
// Scheduling (wave64, persistent):
//   * the grid is sized to the chip (CUs x resident blocks), never to the image;
//   * one lane owns one pixel and walks its samplesPerPixel paths in the reference's order
//     (so the per-pixel fp32 sum is the reference's sum), as a FLAT loop: every iteration
//     each live lane traces exactly one path segment; a lane whose path ends regenerates the
//     next camera path in the same iteration, so lanes never wait for the longest path;
//   * a lane whose pixel is finished is refilled from a global pixel queue: __ballot() of the
//     idle lanes, one atomicAdd by the first idle lane (ffs), popcount-prefix to hand out
//     consecutive pixels.  The queue has 8 shards, one per XCD (HW_REG_XCC_ID), so
//     neighbouring pixels share an L2 and the atomics do not contend; empty shards are
//     stolen from round-robin;
//   * the BVH traversal stack lives in LDS, entry-major (pt_device.h), sized from the
//     measured tree height.
// Pixel order inside the queue is the 8x4-tile order of sutil/WorkDistribution.h:60-81 for
// (rank, world), which is also the multi-GPU partition.
#include "pt_device.h"
#include "render_megakernel.h"

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
__device__ __forceinline__ void onb_transform(const f3& n, f3& p)
{
    f3 bn;
    if (fabsf(n.x) > fabsf(n.z)) bn = mk(-n.y, n.x, 0.0f);
    else                         bn = mk(0.0f, -n.z, n.y);
    bn = normalize(bn);
    const f3 tg = cross(bn, n);
    p = p.x * tg + p.y * bn + p.z * n;
}
// sampleGGX :455-476 (roughness is the literal 0.2 of :880)
__device__ __forceinline__ f3 sample_ggx(float u1, float u2, float roughness, const f3& N)
{
    const float phi = 2.0f * kPIf * u1;
    const float cosTheta = sqrtf((1.0f - u2) / (1.0f + (roughness * roughness - 1.0f) * u2));
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    const f3 H = mk(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
    // :470 compares in double against 0.999; 0.999f rounds up, so the float test is identical
    const f3 up = fabsf(N.z) < 0.999f ? mk(0.0f, 0.0f, 1.0f) : mk(1.0f, 0.0f, 0.0f);
    const f3 tangent = normalize(cross(up, N));
    const f3 bitangent = cross(N, tangent);
    return normalize(H.x * tangent + H.y * bitangent + H.z * N);
}
// fresnelSchlickConductor :494-510
__device__ __forceinline__ f3 fresnel_conductor(float cosTheta, const f3& eta, const f3& k)
{
    const f3 eta2 = eta * eta, k2 = k * k;
    const f3 c2 = mk(cosTheta * cosTheta);
    const f3 t1 = eta2 - k2 - c2;
    const f3 a2plusb2 = mk(sqrtf(t1.x * t1.x + 4 * eta2.x * k2.x), sqrtf(t1.y * t1.y + 4 * eta2.y * k2.y),
                           sqrtf(t1.z * t1.z + 4 * eta2.z * k2.z));
    const f3 t2 = a2plusb2 + c2;
    const f3 Rs = (t2 - 2 * eta * cosTheta + c2) / (t2 + 2 * eta * cosTheta + c2);
    const f3 Rp = Rs * (t2 - 2 * eta * cosTheta + mk(1.0f)) / (t2 + 2 * eta * cosTheta + mk(1.0f));
    return (Rs + Rp) * 0.5f;
}
// FrDielectric :534-559
__device__ __forceinline__ float fr_dielectric(float cosThetaI, float etaI, float etaT)
{
    cosThetaI = clampf(cosThetaI, -1.0f, 1.0f);
    if (!(cosThetaI > 0.0f)) { const float t = etaI; etaI = etaT; etaT = t; cosThetaI = fabsf(cosThetaI); }
    const float sinThetaI = sqrtf(fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI));
    const float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1.0f) return 1.0f;
    const float cosThetaT = sqrtf(fmaxf(0.0f, 1.0f - sinThetaT * sinThetaT));
    const float rParl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    const float rPerp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (rParl * rParl + rPerp * rPerp) / 2.0f;
}
// refract, cuda/helpers.h:107-137
__device__ __forceinline__ bool refract_dir(f3& r, const f3& i, const f3& n, float ior)
{
    f3 nn = n;
    float negNdotV = dot(i, nn);
    float eta;
    if (negNdotV > 0.0f) { eta = ior; nn = -n; negNdotV = -negNdotV; }
    else                 { eta = 1.f / ior; }
    const float k = 1.f - eta * eta * (1.f - negNdotV * negNdotV);
    if (k < 0.0f) { r = mk(0.f); return false; }
    r = normalize(eta * i - (eta * negNdotV + sqrtf(k)) * nn);
    return true;
}
__device__ __forceinline__ float safe_div(float a, float b) { return b == 0.0f ? 0.0f : a / b; }

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

extern __shared__ uint32_t lds_dyn[];

__global__ void __launch_bounds__(kRenderThreads)
k_render(const RenderArgs A)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    LaneStack st;
    st.base = lds_dyn + wave * (A.stack_entries * 64u) + lane;
    const DeviceScene sc = A.scene;

    const f3 eye = mk(A.eye), camU = mk(A.U), camV = mk(A.V), camW = mk(A.W);
    const f3 Lc = mk(A.light.corner), Lv1 = mk(A.light.v1), Lv2 = mk(A.light.v2), Ln = mk(A.light.normal), Le = mk(A.light.emission);
    const float lightA = length(cross(Lv1, Lv2));                    // :1021
    const float fw = (float)(int)A.width, fh = (float)(int)A.height;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    // queue state (wave-uniform)
    uint32_t shard = xcc_id();
    uint32_t shards_left = 8;
    // counters (wave-uniform, flushed once)
    unsigned long long n_radiance = 0, n_shadow = 0, n_paths = 0, n_pixels = 0;

    // lane state
    bool alive = false, new_path = false;
    uint32_t pix = 0, px = 0, py = 0, seed = 0, pseed = 0, samples_left = 0;
    int depth = 0;
    f3 result = mk(0.0f), org = mk(0.0f), dir = mk(0.0f, 0.0f, 1.0f), att = mk(1.0f);

    for (;;) {
        // ---- refill idle lanes from the pixel queue --------------------------------
        unsigned long long idle = __ballot(!alive);
        while (idle != 0ull && shards_left != 0u) {
            const uint32_t want = (uint32_t)__popcll(idle);
            const uint32_t leader = (uint32_t)__ffsll((long long)idle) - 1u;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&A.queue_heads[shard], want);
            base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);   // wave-uniform
            const uint32_t shard_begin = shard * A.shard_size;
            uint32_t shard_end = shard_begin + A.shard_size;
            if (shard_end > A.total_samples) shard_end = A.total_samples;
            if (shard_begin > A.total_samples) shard_end = shard_begin;
            const uint32_t first = shard_begin + base;
            const uint32_t avail = first < shard_end ? shard_end - first : 0u;
            if (!alive) {
                const uint32_t rank = (uint32_t)__popcll(idle & below);
                if (rank < avail) {
                    int x, y;
                    sample_pixel(A.world, (int)A.width, A.rank, (int)(first + rank), x, y);
                    if ((uint32_t)x < A.width && (uint32_t)y < A.height) {
                        px = (uint32_t)x; py = (uint32_t)y;
                        pix = py * A.width + px;
                        seed = tea4(pix, A.frame);                       // :721
                        result = mk(0.0f);
                        samples_left = A.spp;
                        alive = true;
                        new_path = true;
                    }
                }
            }
            if (avail < want) { shard = (shard + 1u) & 7u; shards_left--; }   // shard drained: steal next
            idle = __ballot(!alive);
            if (avail >= want) break;   // everyone asked was served (padding pixels stay idle till next turn)
        }
        const unsigned long long live = __ballot(alive);
        if (live == 0ull) { if (shards_left == 0u) break; else continue; }

        // ---- camera path start, :727-745 ---------------------------------------------
        if (alive && new_path) {
            const float jx = rnd(seed);
            const float jy = rnd(seed);
            const float dx = 2.0f * (((float)px + jx) / fw) - 1.0f;
            const float dy = 2.0f * (((float)py + jy) / fh) - 1.0f;
            dir = normalize(dx * camU + dy * camV + camW);
            org = eye;
            att = mk(1.0f);
            pseed = seed;
            depth = 0;
            new_path = false;
        }

        // ---- radiance segment: closest hit, tmin 0.01, tmax 1e16 (:750-757) -----------
        HitRec hit;
        traverse<false>(sc, st, alive, org, dir, 0.01f, 1e16f, hit);
        n_radiance += (unsigned long long)__popcll(live);

        f3 emission = mk(0.0f), radiance = mk(0.0f), P = mk(0.0f), N = mk(0.0f), new_org = org, new_dir = dir;
        bool done = true;             // __miss__ms :833-847: radiance = background (0), done
        bool want_shadow = false;
        f3 L = mk(0.0f); float Ldist = 0.0f, nDl = 0.0f, LnDl = 0.0f;
        const bool is_hit = alive && hit.slot >= 0;
        if (is_hit) {
            // ---- __closesthit__diffuse__ch :866-1031 -----------------------------------
            const TriRecord* tp = sc.tris + hit.slot;
            const float4 r0 = tp->r0, r1 = tp->r1, r2 = tp->r2;
            const pt_material* mp = sc.mats + __float_as_uint(r2.z);
            const f3 Kd = mk(mp->diffuse), Ke = mk(mp->emission);
            const float IOR = mp->ior;
            const int bsdf = mp->bsdfType;
            const f3 N0 = normalize(cross(mk(r0.w, r1.x, r1.y), mk(r1.z, r1.w, r2.x)));   // :890
            N = faceforward(N0, -dir, N0);
            P = org + hit.t * dir;                                                       // :894
            emission = depth == 0 ? Ke : mk(0.0f);                                       // :898-901
            uint32_t s = pseed;
            if (bsdf == PT_BSDF_DIFFUSE) {                                               // :907-930
                const float z1 = rnd(s);
                const float z2 = rnd(s);
                f3 w_in;
                if (A.useIS) {                                                           // :341-353
                    const float theta = acosf(sqrtf(z1));
                    const float phi = 2.0f * kPIf * z2;
                    w_in = mk(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
                } else {                                                                 // :368-380
                    const float phi = 2.0f * kPIf * z2;
                    w_in = mk(cosf(phi) * sqrtf(1 - z1 * z1), sinf(phi) * sqrtf(1 - z1 * z1), z1);
                }
                onb_transform(N, w_in);
                new_dir = w_in;
                new_org = P;
                att *= Kd;
            } else if (bsdf == PT_BSDF_METALLIC) {                                       // :931-953
                const float z1 = rnd(s);
                const float z2 = rnd(s);
                const f3 mn = sample_ggx(z1, z2, 0.2f, N);
                const f3 R = reflect(dir, mn);
                new_dir = R;
                new_org = P + R * 1e-4f;
                const f3 eta = mk(1.45f, 0.7f, 1.55f), kk = mk(3.0f, 2.2f, 3.5f);
                const float cosTheta = fmaxf(dot(mn, -dir), 0.0f);
                const f3 F = fresnel_conductor(cosTheta, eta, kk);
                att *= F * Kd;
            } else if (bsdf == PT_BSDF_REFRACTION) {                                     // :954-982
                const f3 inc = normalize(dir);
                const float cos_theta = dot(normalize(-dir), N0);
                const float F = fr_dielectric(cos_theta, 1.0f, IOR);
                if (rnd(s) < F) {
                    new_dir = reflect(inc, N0);
                } else {
                    f3 rd;
                    new_dir = refract_dir(rd, inc, N0, IOR) ? rd : reflect(inc, N0);
                }
                new_org = P + new_dir * 1e-3f;
                att *= Kd;
            }
            const float z1 = rnd(s);                                                     // :985-987
            const float z2 = rnd(s);
            pseed = s;
            if (length(Ke) > 0.0f) { radiance = Ke; done = true; }                       // :992-1000
            else                   { radiance = mk(0.0f); done = false; }
            if (A.useDL && bsdf != PT_BSDF_REFRACTION) {                                 // :1003-1026
                const f3 light_pos = Lc + Lv1 * z1 + Lv2 * z2;
                Ldist = length(light_pos - P);
                L = normalize(light_pos - P);
                nDl = dot(N, L);
                LnDl = -dot(Ln, L);
                want_shadow = nDl > 0.0f && LnDl > 0.0f;
            }
        }

        // ---- occlusion ray (traceOcclusion :651-684): any hit occludes -----------------
        const unsigned long long shadow_mask = __ballot(want_shadow);
        if (shadow_mask != 0ull) {
            HitRec sh;
            const bool occluded = traverse<true>(sc, st, want_shadow, P, L, 0.01f, Ldist - 0.01f, sh);
            n_shadow += (unsigned long long)__popcll(shadow_mask);
            if (want_shadow && !occluded) {
                const float weight = nDl * LnDl * lightA / (kPIf * Ldist * Ldist);
                radiance += Le * weight;
            }
        }

        // ---- back in raygen: accumulate, roulette, advance (:760-778) ------------------
        bool end = false, finished = false;
        if (alive) {
            result += emission;
            result += radiance * att;
            const float p = dot(att, mk(0.30f, 0.59f, 0.11f));
            const bool rr = rnd(pseed) > p;
            end = done || rr || (uint32_t)depth >= A.maxDepth;
            if (!end) {
                att = mk(safe_div(att.x, p), safe_div(att.y, p), safe_div(att.z, p));
                org = new_org;
                dir = new_dir;
                ++depth;
            } else {
                samples_left--;
                new_path = true;
                if (samples_left == 0u) {
                    // ---- pixel finished (:782-814) ---------------------------------------
                    f3 accum = result / (float)A.spp;
                    if (A.frame > 0u) {
                        const float a = 1.0f / (float)(A.frame + 1u);
                        const float4 prev = A.accum[pix];
                        accum = lerp3(mk(prev.x, prev.y, prev.z), accum, a);
                    }
                    A.accum[pix] = make_float4(accum.x, accum.y, accum.z, 1.0f);
                    if (A.fb) A.fb[pix] = make_color(accum);
                    alive = false;
                    finished = true;
                }
            }
        }
        // wave-uniform counters: ballots taken with the whole wave converged
        n_paths += (unsigned long long)__popcll(__ballot(end));
        n_pixels += (unsigned long long)__popcll(__ballot(finished));
    }
    if (lane == 0) {
        atomicAdd(&A.counters[0], n_radiance);
        atomicAdd(&A.counters[1], n_shadow);
        atomicAdd(&A.counters[2], n_paths);
        atomicAdd(&A.counters[3], n_pixels);
    }
}

// ---- standalone ray queries (parity tests): same traversal, one ray per lane -------------
__global__ void __launch_bounds__(256)
k_trace_closest(const DeviceScene sc, uint32_t stack_entries, const float* __restrict__ rays, uint32_t n,
                float* __restrict__ t_out, uint32_t* __restrict__ prim_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    LaneStack st;
    st.base = lds_dyn + (threadIdx.x >> 6) * (stack_entries * 64u) + (threadIdx.x & 63u);
    const bool active = i < n;
    f3 o = mk(0.0f), d = mk(0.0f, 0.0f, 1.0f); float tmin = 0.0f, tmax = 0.0f;
    if (active) { const float* r = rays + 8ull * i; o = mk(r[0], r[1], r[2]); d = mk(r[3], r[4], r[5]); tmin = r[6]; tmax = r[7]; }
    HitRec h;
    traverse<false>(sc, st, active, o, d, tmin, tmax, h);
    if (active) { t_out[i] = h.slot >= 0 ? h.t : -1.0f; prim_out[i] = h.prim; }
}

__global__ void __launch_bounds__(256)
k_trace_any(const DeviceScene sc, uint32_t stack_entries, const float* __restrict__ rays, uint32_t n, uint8_t* __restrict__ hit_out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    LaneStack st;
    st.base = lds_dyn + (threadIdx.x >> 6) * (stack_entries * 64u) + (threadIdx.x & 63u);
    const bool active = i < n;
    f3 o = mk(0.0f), d = mk(0.0f, 0.0f, 1.0f); float tmin = 0.0f, tmax = 0.0f;
    if (active) { const float* r = rays + 8ull * i; o = mk(r[0], r[1], r[2]); d = mk(r[3], r[4], r[5]); tmin = r[6]; tmax = r[7]; }
    HitRec h;
    const bool f = traverse<true>(sc, st, active, o, d, tmin, tmax, h);
    if (active) hit_out[i] = f ? 1 : 0;
}

__global__ void k_resolve(const float4* __restrict__ accum, uint32_t* __restrict__ fb, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float4 a = accum[i]; fb[i] = make_color(mk(a.x, a.y, a.z)); }
}

hipError_t launch_resolve(const float4* accum, uint32_t* fb, uint32_t n, hipStream_t stream)
{
    k_resolve<<<(n + 255) / 256, 256, 0, stream>>>(accum, fb, n);
    return hipGetLastError();
}

// ---- host-side launchers ------------------------------------------------------------------
hipError_t render_occupancy(uint32_t stack_entries, int* blocks_per_cu)
{
    const size_t lds = (size_t)(kRenderThreads / 64) * stack_entries * 256u;
    hipError_t e = hipFuncSetAttribute((const void*)k_render, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, (const void*)k_render, kRenderThreads, lds);
}

hipError_t launch_render(const RenderArgs& args, uint32_t grid_blocks, hipStream_t stream)
{
    const size_t lds = (size_t)(kRenderThreads / 64) * args.stack_entries * 256u;
    k_render<<<grid_blocks, kRenderThreads, lds, stream>>>(args);
    return hipGetLastError();
}

hipError_t launch_trace_closest(const DeviceScene& sc, uint32_t stack_entries, const float* d_rays, uint32_t n,
                                float* d_t, uint32_t* d_prim, hipStream_t stream)
{
    const size_t lds = (size_t)4 * stack_entries * 256u;
    hipError_t e = hipFuncSetAttribute((const void*)k_trace_closest, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_trace_closest<<<(n + 255) / 256, 256, lds, stream>>>(sc, stack_entries, d_rays, n, d_t, d_prim);
    return hipGetLastError();
}

hipError_t launch_trace_any(const DeviceScene& sc, uint32_t stack_entries, const float* d_rays, uint32_t n,
                            uint8_t* d_hit, hipStream_t stream)
{
    const size_t lds = (size_t)4 * stack_entries * 256u;
    hipError_t e = hipFuncSetAttribute((const void*)k_trace_any, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    k_trace_any<<<(n + 255) / 256, 256, lds, stream>>>(sc, stack_entries, d_rays, n, d_hit);
    return hipGetLastError();
}

}  // namespace ptd
