// mfma_probe.hip -- what dense bf16 MFMA rate does THIS chip sustain?  (round 5: every GEMM variant of the package and the library's land at
// 1.0-1.17 PFLOP/s on the projection shapes; the ring kernel with its loads, fragment reads and barriers compiled away runs at 1.07.)
// One workgroup per CU (a large dynamic LDS request keeps it that way), 1 or 2 waves per SIMD, every wave issues v_mfma_f32_16x16x32_bf16
// back to back on NACC independent accumulator tiles -- no memory traffic at all inside the timed loop.  Operands: random bf16 (what a GEMM
// on real activations toggles) or zeros (the same instruction stream with nothing switching: the clock the power limit allows differs).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/_bin/mfma_probe && tools/_bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512, 1) void k_mfma(const bf8v* __restrict__ in, float* __restrict__ out, int iters) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x;
    bf8v a[4], b[4];
    for (int q = 0; q < 4; ++q) {
        a[q] = in[(q * 2) * 1024 + lane % 1024];
        b[q] = in[(q * 2 + 1) * 1024 + lane % 1024];
    }
    f4v acc[NACC];
    for (int q = 0; q < NACC; ++q) acc[q] = f4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q & 3], b[(q >> 2) & 3], acc[q], 0, 0, 0);
    }
    f4v s = acc[0];
    for (int q = 1; q < NACC; ++q) s += acc[q];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[blockIdx.x * blockDim.x + lane] = s[0];      // keeps the loop alive
    if (lane == 0 && lds[0] == 77) out[0] = 1.f;
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main() {
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    const int n = 8 * 1024 * 8;
    std::vector<unsigned short> h(n);
    srand(1);
    bf8v* din;
    float* dout;
    hipMalloc(&din, n * 2);
    hipMalloc(&dout, ncu * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 8000, NACC = 16;
    hipFuncSetAttribute((const void*)k_mfma<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int zero = 0; zero < 2; ++zero) {
        for (int i = 0; i < n; ++i) h[i] = zero ? 0 : f2bf((float)rand() / RAND_MAX * 2.f - 1.f);
        hipMemcpy(din, h.data(), n * 2, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 2; ++wps) {
            const int threads = 256 * wps;
            hipLaunchKernelGGL(k_mfma<NACC>, dim3(ncu), dim3(threads), 100 * 1024, 0, din, dout, 200);
            hipDeviceSynchronize();
            float best = 1e30f, tot = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_mfma<NACC>, dim3(ncu), dim3(threads), 100 * 1024, 0, din, dout, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
                tot += ms;
            }
            const double flops = 2.0 * 16 * 16 * 32 * (double)NACC * iters * (threads / 64) * ncu;
            const double per_mfma_ns = best * 1e6 / ((double)NACC * iters * wps);         // per MFMA and SIMD
            printf("%-6s operands, %d wave(s) per SIMD, %d CUs: %.3f ms best / %.3f ms mean  -> %.0f TFLOP/s best, %.0f mean; %.2f ns per MFMA and SIMD "
                   "(= 16 cycles at %.2f GHz)\n", zero ? "zero" : "random", wps, ncu, best, tot / 5, flops / best / 1e9, flops / (tot / 5) / 1e9,
                   per_mfma_ns, 16.0 / per_mfma_ns);
        }
    }
    return 0;
}
