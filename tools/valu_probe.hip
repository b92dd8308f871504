// valu_probe.hip -- issue cost of the vector instructions the time-serial scan kernels are built from, on one MI355X box:
// plain / packed fp32 FMA, v_exp_f32, DPP adds (row shifts, bank-masked rotates, quad permutes), v_permlane32/16_swap,
// v_readlane, ds_bpermute, the 16-bit conversions, and the state-update mix (1 exp : 4 FMA) -- at 1..4 waves per SIMD.
// Every body is 32 instructions over 8 independent registers (inline asm, so the compiler cannot fold or reorder them).
// Output: one line per (body, waves/SIMD): wave-cycles per instruction (s_memtime, one wave's view) and SIMD-ns per
// instruction (wall time / instructions issued on one SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/_bin/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define X4(S) S S S S
#define REGS "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)

#define I_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define I_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n\t"
#define I_EXP(i) "v_exp_f32 %" #i ", %" #i "\n\t"
#define I_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define I_FMAS(i) "v_fma_f32 %" #i ", %" #i ", %10, %9\n\t"
#define I_DPP_SHR(i) "v_add_f32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPP_ROR8(i) "v_add_f32_dpp %" #i ", %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
#define I_DPP_QP(i) "v_add_f32_dpp %" #i ", %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define I_SHL(i) "v_lshlrev_b32 %" #i ", 16, %" #i "\n\t"
#define I_CVT(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %8\n\t"
#define PK4 "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
#define I_BPERM(i) "ds_bpermute_b32 %" #i ", %8, %" #i "\n\t"

enum { K_FMA, K_MUL, K_EXP, K_RCP, K_FMAS, K_PKFMA, K_DPP_SHR, K_DPP_ROR8, K_DPP_QP, K_SHL, K_CVT, K_SWAP32, K_SWAP16, K_READLANE, K_BPERM,
       K_MIX, K_MIXPK, K_DEP_FMA, K_DEP_PK, K_DEP_PK2, K_DEP_DPP, K_DEP_EXP, K_LDS_RT, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32", "v_mul_f32", "v_exp_f32", "v_rcp_f32", "v_fma_f32 (sgpr src)", "v_pk_fma_f32 (2 flop-lanes)", "v_add_f32_dpp row_shr",
                                      "v_add_f32_dpp row_ror:8 bank_mask", "v_add_f32_dpp quad_perm", "v_lshlrev_b32", "v_cvt_pk_bf16_f32",
                                      "v_permlane32_swap", "v_permlane16_swap", "v_readlane_b32", "ds_bpermute_b32",
                                      "mix 8 exp + 24 fma", "mix 8 exp + 12 pk_fma", "DEPENDENT v_fma_f32 chain", "DEPENDENT v_pk_fma_f32 chain (s_nop 0 between)",
                                      "two interleaved dependent v_pk_fma_f32 chains", "DEPENDENT v_add_f32_dpp chain (s_nop 1 between)",
                                      "DEPENDENT v_exp_f32 chain", "ds_read_b64 + s_waitcnt round trip"};

template <int KIND> __global__ __launch_bounds__(256) void k_probe(float* out, long long* cyc, int iters, float a, float b) {
    float r0 = threadIdx.x * 1e-3f, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, pa = {a, a}, pb = {b, b};
    const float sa = __builtin_amdgcn_readfirstlane(a);
    int sacc = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_FMA) asm volatile(X4(R8(I_FMA)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_MUL) asm volatile(X4(R8(I_MUL)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_EXP) asm volatile(X4(R8(I_EXP)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_RCP) asm volatile(X4(R8(I_RCP)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_FMAS) asm volatile(X4(R8(I_FMAS)) : REGS : "v"(a), "v"(b), "s"(sa));
        if constexpr (KIND == K_DPP_SHR) asm volatile(X4(R8(I_DPP_SHR)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_DPP_ROR8) asm volatile(X4(R8(I_DPP_ROR8)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_DPP_QP) asm volatile(X4(R8(I_DPP_QP)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_SHL) asm volatile(X4(R8(I_SHL)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_CVT) asm volatile(X4(R8(I_CVT)) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_BPERM) asm volatile(X4(R8(I_BPERM)) "s_waitcnt lgkmcnt(0)\n\t" : REGS : "v"((int)(threadIdx.x * 4 ^ 128)), "v"(b));
        if constexpr (KIND == K_PKFMA)
            asm volatile(X4(PK4 PK4) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        if constexpr (KIND == K_SWAP32)
            asm volatile(X4(X4("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t")) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        if constexpr (KIND == K_SWAP16)
            asm volatile(X4(X4("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t")) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        if constexpr (KIND == K_READLANE) {
            int s0, s1, s2, s3;
            asm volatile(X4(X4("v_readlane_b32 %0, %4, 5\n\tv_readlane_b32 %1, %5, 9\n\t")) "v_readlane_b32 %2, %6, 1\n\tv_readlane_b32 %3, %7, 2\n\t"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(r0), "v"(r1), "v"(r2), "v"(r3));
            sacc += s0 + s1 + s2 + s3;
        }
        if constexpr (KIND == K_DEP_FMA) asm volatile(X4(X4("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %0, %0, %8, %9\n\t")) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_DEP_PK)
            asm volatile(X4(X4("v_pk_fma_f32 %0, %0, %4, %5\n\ts_nop 0\n\tv_pk_fma_f32 %0, %0, %4, %5\n\ts_nop 0\n\t")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        if constexpr (KIND == K_DEP_PK2)
            asm volatile(X4(X4("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\t")) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
        if constexpr (KIND == K_DEP_DPP)
            asm volatile(X4(X4("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t")) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_DEP_EXP) asm volatile(X4(X4("v_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\t")) : REGS : "v"(a), "v"(b));
        if constexpr (KIND == K_LDS_RT) {
            f2 q;
            asm volatile(X4(X4("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b64 %0, %1 offset:64\n\ts_waitcnt lgkmcnt(0)\n\t")) : "=&v"(q) : "v"(0));
            p0 += q;
        }
        if constexpr (KIND == K_MIX) {      // per state: exp, then 3 dependent-free fma/mul -- the scan's forward mix (1 : 3), 8 + 24
            asm volatile(R8(I_EXP) R8(I_FMA) R8(I_MUL) R8(I_FMA) : REGS : "v"(a), "v"(b));
        }
        if constexpr (KIND == K_MIXPK) {
            asm volatile(R8(I_EXP) X4("v_pk_fma_f32 %8, %8, %14, %15\n\tv_pk_mul_f32 %9, %9, %14\n\tv_pk_fma_f32 %10, %10, %14, %15\n\t")
                         : REGS, "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(a), "v"(b), "v"(pa), "v"(pb));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)sacc;
    if (s == 123.456f) out[0] = s;
}

static int insts_per_iter(int kind) {
    switch (kind) {
        case K_READLANE: return 34;
        case K_MIXPK: return 8 + 12;
        default: return 32;
    }
}

template <int KIND> void run(float* out, long long* cyc, int waves_per_simd) {
    const int iters = 20000;
    const int grid = 256 * waves_per_simd;      // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_probe<KIND><<<grid, 256>>>(out, cyc, 100, 0.999f, 1e-3f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k_probe<KIND><<<grid, 256>>>(out, cyc, iters, 0.999f, 1e-3f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long c;
    CK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    const double n = (double)iters * insts_per_iter(KIND);
    printf("%-36s waves/SIMD %d : %7.2f wave-cycles/inst   %7.3f SIMD-ns/inst   (%.3f ms)\n", kNames[KIND], waves_per_simd, (double)c / n,
           ms * 1e6 / (n * waves_per_simd), ms);
}

template <int KIND> void sweep(float* out, long long* cyc) {
    for (int w : {1, 2, 3, 4}) run<KIND>(out, cyc, w);
}

int main() {
    float* out;
    long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    sweep<K_FMA>(out, cyc); sweep<K_MUL>(out, cyc); sweep<K_FMAS>(out, cyc); sweep<K_PKFMA>(out, cyc); sweep<K_EXP>(out, cyc); sweep<K_RCP>(out, cyc);
    sweep<K_MIX>(out, cyc); sweep<K_MIXPK>(out, cyc);
    sweep<K_DPP_SHR>(out, cyc); sweep<K_DPP_ROR8>(out, cyc); sweep<K_DPP_QP>(out, cyc);
    sweep<K_SWAP32>(out, cyc); sweep<K_SWAP16>(out, cyc); sweep<K_READLANE>(out, cyc); sweep<K_BPERM>(out, cyc);
    sweep<K_SHL>(out, cyc); sweep<K_CVT>(out, cyc);
    sweep<K_DEP_FMA>(out, cyc); sweep<K_DEP_PK>(out, cyc); sweep<K_DEP_PK2>(out, cyc); sweep<K_DEP_DPP>(out, cyc); sweep<K_DEP_EXP>(out, cyc);
    sweep<K_LDS_RT>(out, cyc);
    return 0;
}
