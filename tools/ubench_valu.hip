// ubench_valu.hip — SIMD cycles per wave64 vector instruction on gfx950, for the instruction kinds the BVH
// loop is made of, at 1, 2, 4 and 8 waves per SIMD.  Each kernel runs ITERS x 64 instances of one instruction on
// 8 independent registers per lane (no dependent-issue stall beyond what the hardware imposes).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(int iters, float* out) { \
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    float a = 1.0001f, b = 0.5f; uint32_t m = 0x0F0F0F0Fu; \
    for (int i = 0; i < iters; i++) { \
        _Pragma("unroll") for (int k = 0; k < 8; k++) { \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "v"(m) : "vcc", "s20", "s21"); \
        } \
    } \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0; \
}
#define A_ADD(n)   "v_add_f32 %" #n ", %" #n ", %8\n"
#define A_MUL(n)   "v_mul_f32 %" #n ", %" #n ", %8\n"
#define A_FMA(n)   "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_MAX(n)   "v_max_f32 %" #n ", %" #n ", %8\n"
#define A_MAX3(n)  "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_AND(n)   "v_and_b32 %" #n ", %" #n ", %10\n"
#define A_ADDU(n)  "v_add_u32 %" #n ", %" #n ", %10\n"
#define A_SHL(n)   "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define A_MINU(n)  "v_min_u32 %" #n ", %" #n ", %10\n"
#define A_CVTB(n)  "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define A_CVTU(n)  "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define A_CND(n)   "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_CMP(n)   "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define A_CMPS(n)  "v_cmp_lt_f32 s[20:21], %" #n ", %8\n"
#define A_RCP(n)   "v_rcp_f32 %" #n ", %" #n "\n"
#define A_MOV(n)   "v_mov_b32 %" #n ", %8\n"
#define A_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 4, %10\n"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %10, %10\n"
#define A_BFI(n)   "v_bfi_b32 %" #n ", %10, %" #n ", %8\n"
#define A_MIN3(n)  "v_min3_f32 %" #n ", %" #n ", %8, %9\n"
#define A_SUB(n)   "v_sub_f32 %" #n ", %" #n ", %8\n"
#define A_XOR(n)   "v_xor_b32 %" #n ", %" #n ", %10\n"
#define A_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %10, %10\n"
#define A_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
BODY(k_add, A_ADD) BODY(k_mul, A_MUL) BODY(k_fma, A_FMA) BODY(k_max, A_MAX) BODY(k_max3, A_MAX3) BODY(k_and, A_AND)
BODY(k_addu, A_ADDU) BODY(k_shl, A_SHL) BODY(k_minu, A_MINU) BODY(k_cvtb, A_CVTB) BODY(k_cvtu, A_CVTU) BODY(k_cnd, A_CND)
BODY(k_bfi, A_BFI) BODY(k_min3, A_MIN3) BODY(k_sub, A_SUB) BODY(k_xor, A_XOR) BODY(k_mad24, A_MAD24)
#define A_FMAMIX(n) "v_fma_mix_f32 %" #n ", %" #n ", %8, %9 op_sel_hi:[1,0,0]\n"
#define A_FMAMIXH(n) "v_fma_mix_f32 %" #n ", %" #n ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define A_CVTH(n)  "v_cvt_f32_f16 %" #n ", %" #n "\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", %10\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %" #n ", %10\n"
#define A_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
BODY(k_alignbit, A_ALIGNBIT) BODY(k_perm, A_PERM) BODY(k_med3, A_MED3)
BODY(k_fmamix, A_FMAMIX) BODY(k_fmamixh, A_FMAMIXH) BODY(k_cvth, A_CVTH)
// round 4: what a plane-sharing node visit would add (packed 16-bit integer max as a NaN router, fp16 interval packing), and two
// mixed streams: do the class costs add when full- and half-rate instructions alternate?
#define A_PKMAXI(n) "v_pk_max_i16 %" #n ", %" #n ", %10\n"
#define A_PKMAXU(n) "v_pk_max_u16 %" #n ", %" #n ", %10\n"
#define A_CVTPK(n)  "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %8\n"
#define A_OR(n)     "v_or_b32 %" #n ", %" #n ", %10\n"
#define A_MIXFMA(n) "v_fma_mix_f32 %" #n ", %" #n ", %8, %9 op_sel_hi:[1,0,0]\nv_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_MAXFMA(n) "v_max_f32 %" #n ", %" #n ", %8\nv_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_MIXABS(n) "v_fma_mix_f32 %" #n ", |%" #n "|, %8, %9 op_sel_hi:[1,0,0]\n"
BODY(k_pkmaxi, A_PKMAXI) BODY(k_pkmaxu, A_PKMAXU) BODY(k_cvtpk, A_CVTPK) BODY(k_or, A_OR) BODY(k_mixfma, A_MIXFMA) BODY(k_maxfma, A_MAXFMA) BODY(k_mixabs, A_MIXABS)
BODY(k_cmp, A_CMP) BODY(k_rcp, A_RCP) BODY(k_mov, A_MOV) BODY(k_lshlor, A_LSHLOR) BODY(k_andor, A_ANDOR)

__global__ void __launch_bounds__(256) k_cmps(int iters, float* out) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f, b = 0.5f; uint32_t m = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            asm volatile(A_CMPS(0) A_CMPS(1) A_CMPS(2) A_CMPS(3) A_CMPS(4) A_CMPS(5) A_CMPS(6) A_CMPS(7)
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "v"(m) : "s20", "s21");
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;
}
// v_cndmask variants: mask in an initialised SGPR pair; vcc written by a v_cmp right before each select; v_bfi as a select
__global__ void __launch_bounds__(256) k_cnd_s(int iters, float* out) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f;
    asm volatile("s_mov_b64 s[20:21], 0x55555555\n" ::: "s20", "s21");
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\nv_cndmask_b32 %1, %1, %8, s[20:21]\nv_cndmask_b32 %2, %2, %8, s[20:21]\nv_cndmask_b32 %3, %3, %8, s[20:21]\n"
                         "v_cndmask_b32 %4, %4, %8, s[20:21]\nv_cndmask_b32 %5, %5, %8, s[20:21]\nv_cndmask_b32 %6, %6, %8, s[20:21]\nv_cndmask_b32 %7, %7, %8, s[20:21]\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "s20", "s21");
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;
}
__global__ void __launch_bounds__(256) k_cnd_init(int iters, float* out) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f;
    for (int i = 0; i < iters; i++) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %1\n" :: "v"(r0), "v"(a) : "vcc");
#pragma unroll
        for (int k = 0; k < 8; k++)
            asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : );
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;
}
__global__ void __launch_bounds__(256) k_cmp_cnd(int iters, float* out) {     // counts PAIRS
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cmp_lt_f32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_cmp_lt_f32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_cmp_lt_f32 vcc, %5, %8\nv_cndmask_b32 %5, %5, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %6, %8\nv_cndmask_b32 %6, %6, %8, vcc\nv_cmp_lt_f32 vcc, %7, %8\nv_cndmask_b32 %7, %7, %8, vcc\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;
}
// one compare feeding four selects (a swap of two pairs): counts GROUPS of 5 instructions
__global__ void __launch_bounds__(256) k_cmp_cnd4(int iters, float* out) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;
}
// vcc mask, VOP3 encoding of the selects
__global__ void __launch_bounds__(256) k_cmp_cnd4_e64(int iters, float* out) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\nv_cndmask_b32_e64 %0, %0, %8, vcc\nv_cndmask_b32_e64 %1, %1, %8, vcc\nv_cndmask_b32_e64 %2, %2, %8, vcc\nv_cndmask_b32_e64 %3, %3, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\nv_cndmask_b32_e64 %4, %4, %8, vcc\nv_cndmask_b32_e64 %5, %5, %8, vcc\nv_cndmask_b32_e64 %6, %6, %8, vcc\nv_cndmask_b32_e64 %7, %7, %8, vcc\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;
}
// vcc mask, selects with DISTINCT destination (dst != src0): is it the in-place update?
__global__ void __launch_bounds__(256) k_cmp_cnd4_d(int iters, float* out) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f, b = 2.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            asm volatile("v_cmp_lt_f32 vcc, %8, %9\nv_cndmask_b32 %0, %8, %9, vcc\nv_cndmask_b32 %1, %8, %9, vcc\nv_cndmask_b32 %2, %8, %9, vcc\nv_cndmask_b32 %3, %8, %9, vcc\n"
                         "v_cmp_lt_f32 vcc, %9, %8\nv_cndmask_b32 %4, %8, %9, vcc\nv_cndmask_b32 %5, %8, %9, vcc\nv_cndmask_b32 %6, %8, %9, vcc\nv_cndmask_b32 %7, %8, %9, vcc\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;
}
// same with the mask in an SGPR pair
__global__ void __launch_bounds__(256) k_cmps_cnd4(int iters, float* out) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 32; k++)
            asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\nv_cndmask_b32 %0, %0, %8, s[20:21]\nv_cndmask_b32 %1, %1, %8, s[20:21]\nv_cndmask_b32 %2, %2, %8, s[20:21]\nv_cndmask_b32 %3, %3, %8, s[20:21]\n"
                         "v_cmp_lt_f32 s[22:23], %4, %8\nv_cndmask_b32 %4, %4, %8, s[22:23]\nv_cndmask_b32 %5, %5, %8, s[22:23]\nv_cndmask_b32 %6, %6, %8, s[22:23]\nv_cndmask_b32 %7, %7, %8, s[22:23]\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "s20", "s21", "s22", "s23");
    }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) out[0] = r0;
}
typedef float v2f __attribute__((ext_vector_type(2)));
#define PKBODY(NAME, OP) \
__global__ void __launch_bounds__(256) NAME(int iters, float* out) { \
    v2f r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f; \
    v2f a = {1.0001f, 0.9999f}, b = {0.5f, 0.25f}; \
    for (int i = 0; i < iters; i++) { \
        _Pragma("unroll") for (int k = 0; k < 8; k++) { \
            asm volatile(OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" \
                         OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n" \
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)); \
        } \
    } \
    v2f s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; if (s.x + s.y == 12345.678f) out[0] = s.x; \
}
#define PKBODY2(NAME, OP) \
__global__ void __launch_bounds__(256) NAME(int iters, float* out) { \
    v2f r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f; \
    v2f a = {1.0001f, 0.9999f}; \
    for (int i = 0; i < iters; i++) { \
        _Pragma("unroll") for (int k = 0; k < 8; k++) { \
            asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
                         OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a)); \
        } \
    } \
    v2f s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7; if (s.x + s.y == 12345.678f) out[0] = s.x; \
}
PKBODY(k_pkfma, "v_pk_fma_f32") PKBODY2(k_pkmul, "v_pk_mul_f32") PKBODY2(k_pkadd, "v_pk_add_f32")
// 64-bit address arithmetic as the compiler emits it for `base + index * size` (register pairs)
#define B64BODY(NAME, ASM) \
__global__ void __launch_bounds__(256) NAME(int iters, float* out) { \
    unsigned long long r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    unsigned long long b = 0x1000; uint32_t m = 48; \
    for (int i = 0; i < iters; i++) { \
        _Pragma("unroll") for (int k = 0; k < 8; k++) { \
            asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7) \
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(m) : "vcc"); \
        } \
    } \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345678ull) out[0] = 1.0f; \
}
#define A_SHL64(n)   "v_lshlrev_b64 %" #n ", 6, %" #n "\n"
#define A_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %8\n"
#define A_MAD64(n)   "v_mad_u64_u32 %" #n ", vcc, %9, 48, %" #n "\n"
B64BODY(k_shl64, A_SHL64) B64BODY(k_lshladd64, A_LSHLADD64) B64BODY(k_mad64, A_MAD64)
#define A_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %10\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %10\n"
BODY(k_mul24, A_MUL24) BODY(k_mullo, A_MULLO)
// scalar: s_and_b64 / s_bcnt1 chain next to nothing else
__global__ void __launch_bounds__(256) k_salu(int iters, float* out) {
    unsigned long long x = 0x123456789ull + blockIdx.x; uint32_t c = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 64; k++) asm volatile("s_and_b64 %0, %0, exec\n" : "+s"(x) : : "scc");
    }
    if (x == 0x1234567 && c == 7) out[0] = 1.0f;
}

#define PAIR_ALT_0 1
#define PAIR_ALT_1 0
#define PAIR_ALT_2 3
#define PAIR_ALT_3 2
#define PAIR_ALT_4 5
#define PAIR_ALT_5 4
#define PAIR_ALT_6 7
#define PAIR_ALT_7 6
#define PAIR_CALL(M, n) M(n)
// round 4: PAIRS — which instructions share an execution pipe?  cost(pair) ~ cost(a) + cost(b): same pipe; ~ max(cost(a), cost(b)): they overlap.
// The two instructions of a pair alternate on DIFFERENT registers (n and n^1 of the eight), so that no dependence links them.
#define A_PAIR_ALIGNBIT_FMA(n) A_ALIGNBIT(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_alignbit_fma, A_PAIR_ALIGNBIT_FMA)
#define A_PAIR_MAX_FMA(n) A_MAX(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_max_fma, A_PAIR_MAX_FMA)
#define A_PAIR_MAX3_FMA(n) A_MAX3(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_max3_fma, A_PAIR_MAX3_FMA)
#define A_PAIR_MIN3_FMA(n) A_MIN3(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_min3_fma, A_PAIR_MIN3_FMA)
#define A_PAIR_CMPS_FMA(n) A_CMPS(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_cmps_fma, A_PAIR_CMPS_FMA)
#define A_PAIR_PKMAXI_FMA(n) A_PKMAXI(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_pkmaxi_fma, A_PAIR_PKMAXI_FMA)
#define A_PAIR_SHL_FMA(n) A_SHL(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_shl_fma, A_PAIR_SHL_FMA)
#define A_PAIR_ADDU_FMA(n) A_ADDU(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_addu_fma, A_PAIR_ADDU_FMA)
#define A_PAIR_AND_FMA(n) A_AND(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_and_fma, A_PAIR_AND_FMA)
#define A_PAIR_MOV_FMA(n) A_MOV(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_mov_fma, A_PAIR_MOV_FMA)
#define A_PAIR_CVTH_FMA(n) A_CVTH(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_cvth_fma, A_PAIR_CVTH_FMA)
#define A_PAIR_FMAMIX_FMA(n) A_FMAMIX(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_fmamix_fma, A_PAIR_FMAMIX_FMA)
#define A_PAIR_MUL_FMA(n) A_MUL(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_mul_fma, A_PAIR_MUL_FMA)
#define A_PAIR_BFI_FMA(n) A_BFI(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_bfi_fma, A_PAIR_BFI_FMA)
#define A_PAIR_RCP_FMA(n) A_RCP(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_rcp_fma, A_PAIR_RCP_FMA)
#define A_PAIR_LSHLOR_FMA(n) A_LSHLOR(n) PAIR_CALL(A_FMA, PAIR_ALT_##n)
BODY(k_pair_lshlor_fma, A_PAIR_LSHLOR_FMA)
#define A_PAIR_ALIGNBIT_MAX(n) A_ALIGNBIT(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_alignbit_max, A_PAIR_ALIGNBIT_MAX)
#define A_PAIR_MAX3_MAX(n) A_MAX3(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_max3_max, A_PAIR_MAX3_MAX)
#define A_PAIR_MIN3_MAX(n) A_MIN3(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_min3_max, A_PAIR_MIN3_MAX)
#define A_PAIR_CMPS_MAX(n) A_CMPS(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_cmps_max, A_PAIR_CMPS_MAX)
#define A_PAIR_PKMAXI_MAX(n) A_PKMAXI(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_pkmaxi_max, A_PAIR_PKMAXI_MAX)
#define A_PAIR_SHL_MAX(n) A_SHL(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_shl_max, A_PAIR_SHL_MAX)
#define A_PAIR_ADDU_MAX(n) A_ADDU(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_addu_max, A_PAIR_ADDU_MAX)
#define A_PAIR_AND_MAX(n) A_AND(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_and_max, A_PAIR_AND_MAX)
#define A_PAIR_MOV_MAX(n) A_MOV(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_mov_max, A_PAIR_MOV_MAX)
#define A_PAIR_CVTH_MAX(n) A_CVTH(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_cvth_max, A_PAIR_CVTH_MAX)
#define A_PAIR_FMAMIX_MAX(n) A_FMAMIX(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_fmamix_max, A_PAIR_FMAMIX_MAX)
#define A_PAIR_MUL_MAX(n) A_MUL(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_mul_max, A_PAIR_MUL_MAX)
#define A_PAIR_BFI_MAX(n) A_BFI(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_bfi_max, A_PAIR_BFI_MAX)
#define A_PAIR_RCP_MAX(n) A_RCP(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_rcp_max, A_PAIR_RCP_MAX)
#define A_PAIR_LSHLOR_MAX(n) A_LSHLOR(n) PAIR_CALL(A_MAX, PAIR_ALT_##n)
BODY(k_pair_lshlor_max, A_PAIR_LSHLOR_MAX)
#define A_PAIR_ALIGNBIT_FMAMIX(n) A_ALIGNBIT(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_alignbit_fmamix, A_PAIR_ALIGNBIT_FMAMIX)
#define A_PAIR_MAX_FMAMIX(n) A_MAX(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_max_fmamix, A_PAIR_MAX_FMAMIX)
#define A_PAIR_MAX3_FMAMIX(n) A_MAX3(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_max3_fmamix, A_PAIR_MAX3_FMAMIX)
#define A_PAIR_MIN3_FMAMIX(n) A_MIN3(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_min3_fmamix, A_PAIR_MIN3_FMAMIX)
#define A_PAIR_CMPS_FMAMIX(n) A_CMPS(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_cmps_fmamix, A_PAIR_CMPS_FMAMIX)
#define A_PAIR_PKMAXI_FMAMIX(n) A_PKMAXI(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_pkmaxi_fmamix, A_PAIR_PKMAXI_FMAMIX)
#define A_PAIR_SHL_FMAMIX(n) A_SHL(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_shl_fmamix, A_PAIR_SHL_FMAMIX)
#define A_PAIR_ADDU_FMAMIX(n) A_ADDU(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_addu_fmamix, A_PAIR_ADDU_FMAMIX)
#define A_PAIR_AND_FMAMIX(n) A_AND(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_and_fmamix, A_PAIR_AND_FMAMIX)
#define A_PAIR_MOV_FMAMIX(n) A_MOV(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_mov_fmamix, A_PAIR_MOV_FMAMIX)
#define A_PAIR_CVTH_FMAMIX(n) A_CVTH(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_cvth_fmamix, A_PAIR_CVTH_FMAMIX)
#define A_PAIR_MUL_FMAMIX(n) A_MUL(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_mul_fmamix, A_PAIR_MUL_FMAMIX)
#define A_PAIR_BFI_FMAMIX(n) A_BFI(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_bfi_fmamix, A_PAIR_BFI_FMAMIX)
#define A_PAIR_RCP_FMAMIX(n) A_RCP(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_rcp_fmamix, A_PAIR_RCP_FMAMIX)
#define A_PAIR_LSHLOR_FMAMIX(n) A_LSHLOR(n) PAIR_CALL(A_FMAMIX, PAIR_ALT_##n)
BODY(k_pair_lshlor_fmamix, A_PAIR_LSHLOR_FMAMIX)

// round 4: does a full-rate instruction overlap a half-rate NEIGHBOUR (same wave, adjacent), or any half-rate instruction in flight on the
// SIMD (another wave's)?  The same 4 + 4 instructions per group of eight, once alternating (the PAIR rows) and once in blocks of four.
#define A_BLK_FMA_MAX_0(n) A_FMA(n)
#define A_BLK_SEL(n) A_BLK_SEL_##n
#define A_BLK4(n) PAIR_CALL(A_BLK4_, n)
#define A_BLK4_0 A_FMA(0)
#define A_BLK4_1 A_FMA(1)
#define A_BLK4_2 A_FMA(2)
#define A_BLK4_3 A_FMA(3)
#define A_BLK4_4 A_MAX(4)
#define A_BLK4_5 A_MAX(5)
#define A_BLK4_6 A_MAX(6)
#define A_BLK4_7 A_MAX(7)
#define A_BLK4X(n) A_BLK4_##n
BODY(k_blk4_fma_max, A_BLK4X)
#define A_BLKM_0 A_FMAMIX(0)
#define A_BLKM_1 A_FMAMIX(1)
#define A_BLKM_2 A_FMAMIX(2)
#define A_BLKM_3 A_FMAMIX(3)
#define A_BLKM_4 A_SUB(4)
#define A_BLKM_5 A_SUB(5)
#define A_BLKM_6 A_SUB(6)
#define A_BLKM_7 A_SUB(7)
#define A_BLKMX(n) A_BLKM_##n
BODY(k_blk4_mix_sub, A_BLKMX)
#define A_ALT_0 A_FMA(0)
#define A_ALT_1 A_MAX(1)
#define A_ALT_2 A_FMA(2)
#define A_ALT_3 A_MAX(3)
#define A_ALT_4 A_FMA(4)
#define A_ALT_5 A_MAX(5)
#define A_ALT_6 A_FMA(6)
#define A_ALT_7 A_MAX(7)
#define A_ALTX(n) A_ALT_##n
BODY(k_alt_fma_max, A_ALTX)

typedef void (*Kern)(int, float*);
struct Desc { const char* name; Kern k; };

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, %.2f GHz\n", prop.gcnArchName, cus, ghz);
    float* d; CK(hipMalloc(&d, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const Desc ks[] = {{"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_fma_f32", k_fma}, {"v_fma_mix_f32 (f16 lo src0)", k_fmamix}, {"v_fma_mix_f32 (f16 hi src0)", k_fmamixh}, {"v_cvt_f32_f16", k_cvth}, {"v_alignbit_b32 (vgpr shift)", k_alignbit}, {"v_perm_b32 (vgpr selector)", k_perm}, {"v_med3_f32", k_med3}, {"v_max_f32", k_max}, {"v_max3_f32", k_max3},
                       {"v_and_b32", k_and}, {"v_add_u32", k_addu}, {"v_lshlrev_b32", k_shl}, {"v_min_u32", k_minu}, {"v_cvt_f32_ubyte1", k_cvtb},
                       {"v_cvt_f32_u32", k_cvtu}, {"v_cndmask_b32 (vcc)", k_cnd}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr pair", k_cmps},
                       {"v_cndmask (sgpr pair, set)", k_cnd_s}, {"v_cndmask (vcc from v_cmp)", k_cnd_init}, {"v_cmp + v_cndmask PAIR", k_cmp_cnd},
                       {"1 v_cmp vcc + 4 v_cndmask (x0.2)", k_cmp_cnd4}, {"1 v_cmp sgpr + 4 v_cndmask (x0.2)", k_cmps_cnd4}, {"1 v_cmp vcc + 4 cndmask_e64 vcc", k_cmp_cnd4_e64}, {"1 v_cmp vcc + 4 cndmask dst!=src", k_cmp_cnd4_d},
                       {"v_pk_fma_f32 (2 fma)", k_pkfma}, {"v_pk_mul_f32 (2 mul)", k_pkmul}, {"v_pk_add_f32 (2 add)", k_pkadd},
                       {"v_lshlrev_b64", k_shl64}, {"v_lshl_add_u64", k_lshladd64}, {"v_mad_u64_u32", k_mad64}, {"v_mul_u32_u24", k_mul24}, {"v_mul_lo_u32", k_mullo},
                       {"v_bfi_b32", k_bfi}, {"v_min3_f32", k_min3}, {"v_sub_f32", k_sub}, {"v_xor_b32", k_xor}, {"v_mad_u32_u24", k_mad24}, {"v_rcp_f32", k_rcp}, {"v_mov_b32", k_mov}, {"v_lshl_or_b32", k_lshlor}, {"v_and_or_b32", k_andor}, {"s_and_b64 (scalar)", k_salu},
                       {"v_pk_max_i16", k_pkmaxi}, {"v_pk_max_u16", k_pkmaxu}, {"v_cvt_pkrtz_f16_f32", k_cvtpk}, {"v_or_b32", k_or}, {"v_fma_mix_f32 |src0| (abs modifier)", k_mixabs},
                       {"PAIR v_fma_mix_f32 + v_fma_f32", k_mixfma}, {"PAIR v_max_f32 + v_fma_f32", k_maxfma},
                       {"GROUP of 8 (x0.125): 4 v_fma_f32 then 4 v_max_f32", k_blk4_fma_max}, {"GROUP of 8 (x0.125): fma max fma max ...", k_alt_fma_max}, {"GROUP of 8 (x0.125): 4 v_fma_mix_f32 then 4 v_sub_f32", k_blk4_mix_sub},
                       {"PAIR v_alignbit_b32 + v_fma_f32", k_pair_alignbit_fma},
                       {"PAIR v_max_f32 + v_fma_f32", k_pair_max_fma},
                       {"PAIR v_max3_f32 + v_fma_f32", k_pair_max3_fma},
                       {"PAIR v_min3_f32 + v_fma_f32", k_pair_min3_fma},
                       {"PAIR v_cmp_lt_f32 sgpr + v_fma_f32", k_pair_cmps_fma},
                       {"PAIR v_pk_max_i16 + v_fma_f32", k_pair_pkmaxi_fma},
                       {"PAIR v_lshlrev_b32 + v_fma_f32", k_pair_shl_fma},
                       {"PAIR v_add_u32 + v_fma_f32", k_pair_addu_fma},
                       {"PAIR v_and_b32 + v_fma_f32", k_pair_and_fma},
                       {"PAIR v_mov_b32 + v_fma_f32", k_pair_mov_fma},
                       {"PAIR v_cvt_f32_f16 + v_fma_f32", k_pair_cvth_fma},
                       {"PAIR v_fma_mix_f32 + v_fma_f32", k_pair_fmamix_fma},
                       {"PAIR v_mul_f32 + v_fma_f32", k_pair_mul_fma},
                       {"PAIR v_bfi_b32 + v_fma_f32", k_pair_bfi_fma},
                       {"PAIR v_rcp_f32 + v_fma_f32", k_pair_rcp_fma},
                       {"PAIR v_lshl_or_b32 + v_fma_f32", k_pair_lshlor_fma},
                       {"PAIR v_alignbit_b32 + v_max_f32", k_pair_alignbit_max},
                       {"PAIR v_max3_f32 + v_max_f32", k_pair_max3_max},
                       {"PAIR v_min3_f32 + v_max_f32", k_pair_min3_max},
                       {"PAIR v_cmp_lt_f32 sgpr + v_max_f32", k_pair_cmps_max},
                       {"PAIR v_pk_max_i16 + v_max_f32", k_pair_pkmaxi_max},
                       {"PAIR v_lshlrev_b32 + v_max_f32", k_pair_shl_max},
                       {"PAIR v_add_u32 + v_max_f32", k_pair_addu_max},
                       {"PAIR v_and_b32 + v_max_f32", k_pair_and_max},
                       {"PAIR v_mov_b32 + v_max_f32", k_pair_mov_max},
                       {"PAIR v_cvt_f32_f16 + v_max_f32", k_pair_cvth_max},
                       {"PAIR v_fma_mix_f32 + v_max_f32", k_pair_fmamix_max},
                       {"PAIR v_mul_f32 + v_max_f32", k_pair_mul_max},
                       {"PAIR v_bfi_b32 + v_max_f32", k_pair_bfi_max},
                       {"PAIR v_rcp_f32 + v_max_f32", k_pair_rcp_max},
                       {"PAIR v_lshl_or_b32 + v_max_f32", k_pair_lshlor_max},
                       {"PAIR v_alignbit_b32 + v_fma_mix_f32", k_pair_alignbit_fmamix},
                       {"PAIR v_max_f32 + v_fma_mix_f32", k_pair_max_fmamix},
                       {"PAIR v_max3_f32 + v_fma_mix_f32", k_pair_max3_fmamix},
                       {"PAIR v_min3_f32 + v_fma_mix_f32", k_pair_min3_fmamix},
                       {"PAIR v_cmp_lt_f32 sgpr + v_fma_mix_f32", k_pair_cmps_fmamix},
                       {"PAIR v_pk_max_i16 + v_fma_mix_f32", k_pair_pkmaxi_fmamix},
                       {"PAIR v_lshlrev_b32 + v_fma_mix_f32", k_pair_shl_fmamix},
                       {"PAIR v_add_u32 + v_fma_mix_f32", k_pair_addu_fmamix},
                       {"PAIR v_and_b32 + v_fma_mix_f32", k_pair_and_fmamix},
                       {"PAIR v_mov_b32 + v_fma_mix_f32", k_pair_mov_fmamix},
                       {"PAIR v_cvt_f32_f16 + v_fma_mix_f32", k_pair_cvth_fmamix},
                       {"PAIR v_mul_f32 + v_fma_mix_f32", k_pair_mul_fmamix},
                       {"PAIR v_bfi_b32 + v_fma_mix_f32", k_pair_bfi_fmamix},
                       {"PAIR v_rcp_f32 + v_fma_mix_f32", k_pair_rcp_fmamix},
                       {"PAIR v_lshl_or_b32 + v_fma_mix_f32", k_pair_lshlor_fmamix}};
    const int iters = 2000;
    printf("%-26s %10s %10s %10s %10s   SIMD cycles per wave-instruction\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    for (const Desc& kd : ks) {
        printf("%-34s", kd.name);
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;      // 256 threads = 4 waves = one per SIMD
            kd.k<<<blocks, 256>>>(10, d);
            CK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                CK(hipEventRecord(e0));
                kd.k<<<blocks, 256>>>(iters, d);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double instr_per_simd = (double)iters * 64.0 * wps;
            printf(" %10.2f", best * 1e-3 * ghz * 1e9 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
