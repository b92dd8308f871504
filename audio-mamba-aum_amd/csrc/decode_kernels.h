// decode_kernels.h -- the two per-token kernels of streaming inference (Mamba.step, MS:313-358): the conv window update
// (causal_conv1d_update of the causal_conv1d wheel, MS:328-334) and the single-step state update (selective_state_update,
// ops/triton/selective_state_update.py:157-192).  One thread per (batch entry, channel); the caches (conv window, SSM state) are fp32 and
// updated in place; everything a thread touches is a few dozen bytes -- these launches are latency, not bandwidth.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode_args.h"

namespace aumdec {

template <class T> __device__ __forceinline__ float ld(const void* p, int64_t i);
template <> __device__ __forceinline__ float ld<float>(const void* p, int64_t i) { return static_cast<const float*>(p)[i]; }
template <> __device__ __forceinline__ float ld<__bf16>(const void* p, int64_t i) { return (float)static_cast<const __bf16*>(p)[i]; }
template <> __device__ __forceinline__ float ld<_Float16>(const void* p, int64_t i) { return (float)static_cast<const _Float16*>(p)[i]; }
template <class T> __device__ __forceinline__ void st(void* p, int64_t i, float v) { static_cast<T*>(p)[i] = (T)v; }

// window <- (window[1:], x);  out = act(<window, weight> + bias)
template <class T>
__global__ void k_conv_update(AumConvUpdateArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)a.batch * a.dim) return;
    const int d = (int)(i % a.dim);
    float* win = a.conv_state + i * a.width;
    const float* w = a.weight + (int64_t)d * a.width;
    float acc = a.bias ? a.bias[d] : 0.f;
    for (int k = 0; k + 1 < a.width; ++k) {
        const float v = win[k + 1];
        win[k] = v;
        acc += v * w[k];
    }
    const float xn = ld<T>(a.x, i);
    win[a.width - 1] = xn;
    acc += xn * w[a.width - 1];
    if (a.flags & AUM_CONV_SILU) acc = acc / (1.f + __expf(-acc));
    st<T>(a.out, i, acc);
}

// h <- exp(dt A) h + (dt x) B;  y = <h, C> + D x;  out = y * silu(z)
template <class T>
__global__ void k_state_update(AumStateUpdateArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)a.batch * a.dim) return;
    const int d = (int)(i % a.dim);
    const int64_t b = i / a.dim;
    float dt = ld<T>(a.dt, i) + (a.dt_bias ? a.dt_bias[d] : 0.f);
    if (a.flags & AUM_SCAN_SOFTPLUS) dt = dt > 20.f ? dt : log1pf(__expf(dt));
    const float x = ld<T>(a.x, i);
    float* h = a.state + i * a.dstate;
    const float* A = a.A + (int64_t)d * a.dstate;
    float y = 0.f;
    for (int n = 0; n < a.dstate; ++n) {
        const float hn = __expf(dt * A[n]) * h[n] + dt * x * ld<T>(a.B, b * a.dstate + n);
        h[n] = hn;
        y += hn * ld<T>(a.C, b * a.dstate + n);
    }
    if (a.D) y += a.D[d] * x;
    if (a.z) {
        const float z = ld<T>(a.z, i);
        y *= z / (1.f + __expf(-z));
    }
    st<T>(a.out, i, y);
}

}  // namespace aumdec
