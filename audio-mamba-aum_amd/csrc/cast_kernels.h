// cast_kernels.h -- the 16-bit copies of a model's fp32 master weights, one launch per group of equally shaped matrices (ABI 13,
// aum_cast_bank).  Under autocast the reference casts every Linear weight on every forward (torch.autocast's cast of F.linear's
// operands, MS:185-189 / SSI:467-468, 517); here the casts of all blocks are hoisted to the top of the forward (ssi.step_cache), and the
// data-gradient GEMMs want some of the weights transposed as well.  Done with torch that is a multi-tensor copy (2.5 TB/s) plus a
// strided transposing copy (1 TB/s): 0.35 ms per step of AuM-Base (profiles/r06_step_timeline.txt).  This kernel reads every fp32
// matrix once and writes the 16-bit copy and, if asked, its transpose: 64 x 64 tiles, the transpose through LDS, every global access
// 16 bytes per lane on the read side and 8 / 16 bytes on the write side.  Rounding: round-to-nearest-even (what Tensor.to() does).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cast_args.h"

namespace aumc {

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));

template <bool BF16> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{a, b}, bf2v));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{a, b}, h2v));
}

constexpr int CT_PITCH = CT * 2 + 16;   // bytes per LDS row of the 16-bit tile (144: nine 16-byte chunks -> column reads spread over the banks)

struct CastBank {
    const uint64_t* src;        // device array of n addresses: fp32 (rows, cols) contiguous matrices
    void* bank;                 // (n, rows, cols) 16-bit
    void* bank_t;               // (n, cols, rows) 16-bit, or null
    int32_t n, rows, cols;
    int32_t tiles_r, tiles_c;
};

// workgroup = 256 threads = one 64 x 64 tile of one matrix; rows % 8 == 0, cols % 4 == 0 (ragged tiles: whole 4-column / 8-row groups are
// inside or outside)
template <bool BF16>
__global__ __launch_bounds__(256) void k_cast_bank(CastBank a) {
    __shared__ __attribute__((aligned(16))) char lds[CT * CT_PITCH];
    const int tid = (int)threadIdx.x;
    const int per = a.tiles_r * a.tiles_c;
    const int m = (int)blockIdx.x / per, t = (int)blockIdx.x - m * per;
    const int tr = t / a.tiles_c, tc = t - tr * a.tiles_c;
    const float* src = reinterpret_cast<const float*>(a.src[m]);
    const int64_t mat = (int64_t)a.rows * a.cols;
    char* dst = static_cast<char*>(a.bank) + (int64_t)m * mat * 2;
    // read: thread = (row r0 + 16 q, 4 columns), q = 0 .. 3; all four loads first
    const int rr = tid >> 4, c4 = (tid & 15) * 4;
    const int col = tc * CT + c4;
    f4v v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = tr * CT + rr + 16 * q;
        v[q] = (row < a.rows && col < a.cols) ? *reinterpret_cast<const f4v*>(src + (int64_t)row * a.cols + col) : f4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = tr * CT + rr + 16 * q;
        const u2v p = {pack2<BF16>(v[q][0], v[q][1]), pack2<BF16>(v[q][2], v[q][3])};
        if (row < a.rows && col < a.cols) *reinterpret_cast<u2v*>(dst + ((int64_t)row * a.cols + col) * 2) = p;
        if (a.bank_t) *reinterpret_cast<u2v*>(lds + (rr + 16 * q) * CT_PITCH + c4 * 2) = p;
    }
    if (!a.bank_t) return;          // uniform over the launch
    __syncthreads();
    // write the transpose: thread = (tile column c, 8 tile rows from 8 g), two such pieces per thread; element (row, c) sits at lds[row][c]
    char* dt = static_cast<char*>(a.bank_t) + (int64_t)m * mat * 2;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int idx = tid + 256 * h;          // 0 .. 511 = 64 columns x 8 row groups
        const int c = idx >> 3, g = idx & 7;
        const int ocol = tc * CT + c, orow = tr * CT + g * 8;
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = *reinterpret_cast<const uint16_t*>(lds + (g * 8 + 2 * k) * CT_PITCH + c * 2);
            const uint32_t hi = *reinterpret_cast<const uint16_t*>(lds + (g * 8 + 2 * k + 1) * CT_PITCH + c * 2);
            w[k] = lo | (hi << 16);
        }
        if (ocol < a.cols && orow < a.rows) *reinterpret_cast<u4v*>(dt + ((int64_t)ocol * a.rows + orow) * 2) = u4v{w[0], w[1], w[2], w[3]};
    }
}

}  // namespace aumc
