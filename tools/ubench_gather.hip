// ubench_gather.hip — how expensive is a divergent per-lane gather on gfx950?
// Measures CU-cycles per wave-level load instruction for global (L1/L2-resident table) and LDS
// gathers as a function of load width and of the number of active lanes.  Informs the BVH node
// format of acgpathtracing_amd/csrc (measure, don't guess).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = 1664525u * s + 1013904223u; return s; }

template <int WIDTH>   // dwords per lane per load: 1, 2, 4
__global__ void __launch_bounds__(256) k_global(const uint32_t* __restrict__ table, uint32_t mask_dw, int iters, int active, uint32_t* out)
{
    const int lane = threadIdx.x & 63;
    uint32_t s = blockIdx.x * 256 + threadIdx.x + 1;
    uint32_t acc = 0;
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
            uint32_t a[4];
#pragma unroll
            for (int k = 0; k < 4; k++) a[k] = ((lcg(s) >> 4) & mask_dw) & ~(uint32_t)(WIDTH - 1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (WIDTH == 4) { uint4 v = *(const uint4*)(table + a[k]); acc += v.x ^ v.y ^ v.z ^ v.w; }
                else if (WIDTH == 2) { uint2 v = *(const uint2*)(table + a[k]); acc += v.x ^ v.y; }
                else acc += table[a[k]];
            }
            s += acc & 1;   // dependent chain like a traversal
        }
    }
    if (acc == 0x12345678) out[0] = acc;
}

// node-like: consecutive dwordx4 loads from one 64-byte record
__global__ void __launch_bounds__(256) k_global_node(const uint32_t* __restrict__ table, uint32_t mask_dw, int iters, int active, int nloads, uint32_t* out)
{
    const int lane = threadIdx.x & 63;
    uint32_t s = blockIdx.x * 256 + threadIdx.x + 1;
    uint32_t acc = 0;
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
            uint32_t a = ((lcg(s) >> 4) & mask_dw) & ~15u;
            for (int k = 0; k < nloads; k++) { uint4 v = *(const uint4*)(table + a + 4 * k); acc += v.x ^ v.y ^ v.z ^ v.w; }
            s += acc & 1;
        }
    }
    if (acc == 0x12345678) out[0] = acc;
}

template <int WIDTH>
__global__ void __launch_bounds__(256) k_lds(const uint32_t* __restrict__ table, uint32_t mask_dw, int iters, int active, uint32_t* out)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i <= mask_dw; i += 256) lds[i] = table[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t s = blockIdx.x * 256 + threadIdx.x + 1;
    uint32_t acc = 0;
    if (lane < active) {
        for (int i = 0; i < iters; i++) {
            uint32_t a[4];
#pragma unroll
            for (int k = 0; k < 4; k++) a[k] = ((lcg(s) >> 4) & mask_dw) & ~(uint32_t)(WIDTH - 1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (WIDTH == 4) { uint4 v = *(const uint4*)(lds + a[k]); acc += v.x ^ v.y ^ v.z ^ v.w; }
                else if (WIDTH == 2) { uint2 v = *(const uint2*)(lds + a[k]); acc += v.x ^ v.y; }
                else acc += lds[a[k]];
            }
            s += acc & 1;
        }
    }
    if (acc == 0x12345678) out[0] = acc;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, %.2f GHz\n", prop.name, cus, ghz);
    const size_t tbytes = 1 << 20;
    std::vector<uint32_t> h(tbytes / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u);
    uint32_t *d_t, *d_o; CK(hipMalloc(&d_t, tbytes)); CK(hipMalloc(&d_o, 64));
    CK(hipMemcpy(d_t, h.data(), tbytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    const int blocks_per_cu = 4;   // 16 waves per CU like the render kernel
    auto run = [&](const char* name, auto launch, int active, int loads_per_iter) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms * 1e-3 * ghz * 1e9;
        const double winstr_per_cu = (double)blocks_per_cu * 4 * iters * loads_per_iter;
        printf("%-34s active %2d : %8.3f ms  %6.2f cycles per wave-load per CU  %6.3f cycles per lane-load\n",
               name, active, ms, cyc / winstr_per_cu, cyc / winstr_per_cu / active);
    };
    const int grid = cus * blocks_per_cu;
    for (uint32_t kb : {16u, 128u}) {
        const uint32_t mask = kb * 1024 / 4 - 1;
        printf("--- global table %u KB\n", kb);
        for (int active : {64, 32, 16, 8}) {
            char nm[64];
            snprintf(nm, 64, "global dword   %uKB", kb); run(nm, [&] { k_global<1><<<grid, 256>>>(d_t, mask, iters, active, d_o); }, active, 4);
            snprintf(nm, 64, "global dwordx2 %uKB", kb); run(nm, [&] { k_global<2><<<grid, 256>>>(d_t, mask, iters, active, d_o); }, active, 4);
            snprintf(nm, 64, "global dwordx4 %uKB", kb); run(nm, [&] { k_global<4><<<grid, 256>>>(d_t, mask, iters, active, d_o); }, active, 4);
            snprintf(nm, 64, "global node 4x dwordx4 %uKB", kb); run(nm, [&] { k_global_node<<<grid, 256>>>(d_t, mask, iters, active, 4, d_o); }, active, 4);
            snprintf(nm, 64, "global node 2x dwordx4 %uKB", kb); run(nm, [&] { k_global_node<<<grid, 256>>>(d_t, mask, iters, active, 2, d_o); }, active, 2);
        }
    }
    {
        const uint32_t kb = 32, mask = kb * 1024 / 4 - 1;
        printf("--- LDS table %u KB per block\n", kb);
        for (int active : {64, 32, 16, 8}) {
            run("lds b32", [&] { k_lds<1><<<grid, 256, kb * 1024>>>(d_t, mask, iters, active, d_o); }, active, 4);
            run("lds b64", [&] { k_lds<2><<<grid, 256, kb * 1024>>>(d_t, mask, iters, active, d_o); }, active, 4);
            run("lds b128", [&] { k_lds<4><<<grid, 256, kb * 1024>>>(d_t, mask, iters, active, d_o); }, active, 4);
        }
    }
    return 0;
}
