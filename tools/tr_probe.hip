// what ds_read_b64_tr_b16 returns: LDS holds lds[i] = i (16-bit); every lane passes its own address (lane * 8 bytes: 4 consecutive elements);
// prints, per lane, the four elements it got.  hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o tools/_bin/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4 lds_s4;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(&lds[l * 4]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
