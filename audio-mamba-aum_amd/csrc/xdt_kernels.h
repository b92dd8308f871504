// xdt_kernels.h -- x_proj and dt_proj of the token-major block in ONE pass over conv_out (round 3; ABI 9 aum_xdt_tm_fwd).
//
//   x_dbl[M][C] = u[M][E] . W_x[C][E]^T            SSI:467        (C = dt_rank + 2 d_state = 80 for AuM-Base)
//   delta[M][E] = x_dbl[M][0:R] . W_dt[E][R]^T     SSI:468        (R = dt_rank)
//
// Both are bound by one stream each -- the first reads an activation tensor (100 MB at the bench shape) to produce 5 MB, the second
// turns those 5 MB into an activation tensor -- and as two launches the small tensor makes a round trip through memory in between.
// Here a workgroup of 4 waves takes 128 tokens through both: W_x (245 KB) passes through LDS in two K-halves shared by the four waves
// (rows padded by 16 bytes: fragment reads spread over the banks), a wave accumulates its 32 tokens x 80 columns on the matrix pipe
// (v_mfma_f32_16x16x32: rows = x_dbl columns, columns = tokens), rounds them into an LDS tile of its own -- from which x_dbl leaves in
// 16-byte pieces and the dt block comes back as the MFMA B operand -- and then streams W_dt through the same LDS space in two channel
// halves, storing delta 64 contiguous bytes per token row and instruction (the store layout of dtproj_kernels.h).
// u is fetched 32 contiguous bytes per lane (a full 128-byte line per token row and K-step of 64) four K-steps ahead of its use; the k
// order inside a step is permuted the same way on both operands (lane group kg holds k0 + 16 kg .. + 15), which the product does not see.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xdt_args.h"

namespace aumx {

typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u2v __attribute__((ext_vector_type(2)));

template <bool BF16> __device__ __forceinline__ f4v mfma(s8v a, s8v b, f4v c) {
    if constexpr (BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <bool BF16> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{a, b}, bf2v));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2v{a, b}, h2v));
}

template <bool BF16> __device__ __forceinline__ float lo16(uint32_t v) {
    if constexpr (BF16) return __builtin_bit_cast(float, v << 16);
    else return (float)__builtin_bit_cast(h2v, v)[0];
}
template <bool BF16> __device__ __forceinline__ float hi16(uint32_t v) {
    if constexpr (BF16) return __builtin_bit_cast(float, v & 0xffff0000u);
    else return (float)__builtin_bit_cast(h2v, v)[1];
}

constexpr int NCF = XDT_COLS / 16;                        // 5 column fragments of x_dbl
constexpr int NTF = XDT_TOK_W / 16;                       // token fragments per wave
constexpr int XP = XDT_COLS * 2 + 16;                     // bytes per row of a wave's x_dbl tile (176: 11 x 16, odd -> conflict-free reads)
constexpr int XT_BYTES = XDT_TOK_W * XP;                  // one wave's tile
constexpr int WDP = 144;                                  // bytes per W_dt row in LDS (rank <= 64: 128 + 16)
__host__ __device__ constexpr int slab_pitch(int kh) { return kh * 2 + 16; }          // bytes per W_x row of a K-half
__host__ __device__ constexpr int slab_bytes(int dim) {
    const int a = XDT_COLS * slab_pitch(dim / 2), b = (dim / 2) * WDP;
    return a > b ? a : b;
}
__host__ __device__ constexpr int lds_bytes(int dim, int nw) { return slab_bytes(dim) + nw * XT_BYTES; }

// rows x chunks 16-byte pieces of a row-major matrix (row pitch src_pitch bytes) -> LDS rows of dst_pitch bytes, by all 256 threads: eight
// loads in flight per thread before the first LDS write (written as one load-store pair per iteration, each iteration was a round trip to
// L2 on the critical path: 30 of them per W_x half -- 119 us for the kernel instead of 60)
template <int NT>
__device__ __forceinline__ void stage_rows(char* dst, int dst_pitch, const char* src, int64_t src_pitch, int rows, int chunks, int tid) {
    const int total = rows * chunks;
    for (int base = tid; base < total; base += NT * 8) {
        u4v r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = base + j * NT;
            if (idx < total) {
                const int row = idx / chunks, ch = idx - row * chunks;
                r[j] = *reinterpret_cast<const u4v*>(src + row * src_pitch + ch * 16);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = base + j * NT;
            if (idx < total) {
                const int row = idx / chunks, ch = idx - row * chunks;
                *reinterpret_cast<u4v*>(dst + row * dst_pitch + ch * 16) = r[j];
            }
        }
    }
}

// NW waves of 16 tokens per workgroup, one workgroup per CU.  The token count decides NW (xdt_waves): 64 x 513 tokens are 2052 fragments of
// 16 -- 8 waves make 257 workgroups, a second round for ONE workgroup on 256 CUs (97 us, cold caches); 9 waves make 228 of them (one round).
// NC = columns of x_dbl: 80 (AuM-Base: dt_rank 48 + 2 x 16) or 56 (AuM-Small: 24 + 32).  56 is three and a half column fragments: the
// fourth fragment multiplies eight weight rows that do not exist (whatever the slab holds behind row 55) -- their products are columns 56..63
// of the wave's tile, inside the row's padding, which nothing reads.
template <bool BF16, int KS, int NW, int NC>
__global__ __launch_bounds__(NW * 64, 1) void k_xdt_tm_fwd(AumXdtArgs g) {
    constexpr int NCF = (NC + 15) / 16;
    constexpr int XP = NC == 80 ? 176 : 144;             // bytes per tile row: an odd number of 16-byte chunks (conflict-free fragment reads)
    static_assert(NC % 8 == 0 && NCF * 32 <= XP && XP <= aumx::XP, "tile row holds every fragment column");
    __shared__ __attribute__((aligned(16))) char lds[lds_bytes(XDT_MAX_DIM, NW)];       // 143 KB at 8 waves: one workgroup per CU
    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rho = lane & 15, kg = lane >> 4;
    const int E = g.dim, KH = E / 2, SP = slab_pitch(KH);
    char* slab = lds;
    char* xt = lds + slab_bytes(E) + w * XT_BYTES;
    const int64_t t0 = (int64_t)blockIdx.x * (NW * XDT_TOK_W) + w * XDT_TOK_W;
    const s8v zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const char* ub = static_cast<const char*>(g.u);
    bool tok_ok[NTF];
    const char* urow[NTF];
    for (int tf = 0; tf < NTF; ++tf) {
        tok_ok[tf] = t0 + tf * 16 + rho < g.ntok;
        urow[tf] = ub + ((t0 + tf * 16 + rho) * g.ldu + kg * 16) * 2;
    }

    f4v acc[NTF][NCF];
#pragma unroll
    for (int tf = 0; tf < NTF; ++tf)
#pragma unroll
        for (int f = 0; f < NCF; ++f) acc[tf][f] = f4v{0.f, 0.f, 0.f, 0.f};

    // ---- stage A: x_dbl = u . W_x^T, K in two halves of W_x through LDS -------------------------------------------------------
    auto load_u = [&](int k0, s8v (&uf)[NTF][2]) {
#pragma unroll
        for (int tf = 0; tf < NTF; ++tf) {
            uf[tf][0] = tok_ok[tf] ? *reinterpret_cast<const s8v*>(urow[tf] + (int64_t)k0 * 2) : zero;
            uf[tf][1] = tok_ok[tf] ? *reinterpret_cast<const s8v*>(urow[tf] + (int64_t)k0 * 2 + 16) : zero;
        }
    };
    // conv_out rows arrive through a ring of four register sets indexed by the (unrolled) step: a set is refilled, four steps ahead, right
    // after the MFMAs that read it were issued.  (Rotating three sets through register copies -- the first version -- made every copy wait
    // for its load: the prefetch distance collapsed to one step.)
    const int nsteps = KH / 64, total = E / 64;               // steps per K-half; dim % 256 == 0 -> total % 4 == 0
    s8v ur[4][NTF][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < total) load_u(j * 64, ur[j]);
    const char* wrd = slab + rho * SP + kg * 32;              // + f * 16 rows, + step-in-half * 128 bytes
    for (int s0 = 0; s0 < total; s0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = s0 + j;
            if (s == 0 || s == nsteps) {                      // a K-half of W_x through LDS
                __syncthreads();                              // everybody is done with the previous contents of the slab
                stage_rows<NW * 64>(slab, SP, static_cast<const char*>(g.wx) + (int64_t)(s / nsteps) * KH * 2, (int64_t)g.ldwx * 2, NC, KH / 8, tid);
                __syncthreads();
            }
            const int sl = s >= nsteps ? s - nsteps : s;
#pragma unroll
            for (int f = 0; f < NCF; ++f) {
                const s8v w0 = *reinterpret_cast<const s8v*>(wrd + f * 16 * SP + sl * 128);
                const s8v w1 = *reinterpret_cast<const s8v*>(wrd + f * 16 * SP + sl * 128 + 16);
#pragma unroll
                for (int tf = 0; tf < NTF; ++tf) {
                    acc[tf][f] = mfma<BF16>(w0, ur[j][tf][0], acc[tf][f]);
                    acc[tf][f] = mfma<BF16>(w1, ur[j][tf][1], acc[tf][f]);
                }
            }
            if (s + 4 < total) load_u((s + 4) * 64, ur[j]);
        }
    }
    // ---- the wave's x_dbl tile: rounded once, [token][column] in LDS; x_dbl leaves from there in 16-byte pieces --------------------
    // lane (kg, token rho) holds columns 16 f + 4 kg + r of token fragment tf
#pragma unroll
    for (int tf = 0; tf < NTF; ++tf)
#pragma unroll
        for (int f = 0; f < NCF; ++f) {
            u2v v;
            v.x = pack2<BF16>(acc[tf][f][0], acc[tf][f][1]);
            v.y = pack2<BF16>(acc[tf][f][2], acc[tf][f][3]);
            *reinterpret_cast<u2v*>(xt + (tf * 16 + rho) * XP + (f * 16 + kg * 4) * 2) = v;
        }
    __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the wave's own LDS writes have landed (nobody else reads this tile)
    {
        char* xo = static_cast<char*>(g.x_dbl);
        constexpr int PIECES = NC * 2 / 16;                          // 10 / 7 per token row
        for (int idx = lane; idx < XDT_TOK_W * PIECES; idx += 64) {
            const int tk = idx / PIECES, pc = idx - tk * PIECES;
            if (t0 + tk < g.ntok)
                *reinterpret_cast<u4v*>(xo + ((t0 + tk) * g.ldx) * 2 + pc * 16) = *reinterpret_cast<const u4v*>(xt + tk * XP + pc * 16);
        }
    }
    // ---- stage B: delta = x_dbl[:, :R] . W_dt^T, the channels in two halves of W_dt through the same LDS space -------------------
    s8v xf[NTF][KS];
#pragma unroll
    for (int tf = 0; tf < NTF; ++tf)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = ks * 32 + kg * 8;
            xf[tf][ks] = k < g.rank ? *reinterpret_cast<const s8v*>(xt + (tf * 16 + rho) * XP + k * 2) : zero;
        }
    const bool k_ok[2] = {kg * 8 < g.rank, 32 + kg * 8 < g.rank};
    const int CH = E / 2, wchunks = g.rank / 8;
    char* ob = static_cast<char*>(g.delta) + ((t0 + rho) * g.ldd + kg * 8) * 2;
    const int64_t otf = (int64_t)16 * g.ldd * 2;
    // fragment j of a channel pair reads weight rows c0 + 8 (rho >> 2) + 4 j + (rho & 3): accumulator rows 4 kg + r of fragments 0, 1 are
    // channels c0 + 8 kg + 0..7
    const char* wdr = slab + ((rho >> 2) * 8 + (rho & 3)) * WDP + kg * 16;
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        stage_rows<NW * 64>(slab, WDP, static_cast<const char*>(g.wdt) + (int64_t)half * CH * g.ldwdt * 2, (int64_t)g.ldwdt * 2, CH, wchunks, tid);
        __syncthreads();
        const int npairs = CH / 32;
#pragma unroll 2
        for (int p = 0; p < npairs; ++p) {
            s8v wf[2][KS];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    wf[j][ks] = k_ok[ks] ? *reinterpret_cast<const s8v*>(wdr + (p * 32 + j * 4) * WDP + ks * 64) : zero;
#pragma unroll
            for (int tf = 0; tf < NTF; ++tf) {
                f4v a2[2] = {f4v{0.f, 0.f, 0.f, 0.f}, f4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) a2[j] = mfma<BF16>(wf[j][ks], xf[tf][ks], a2[j]);
                if (tok_ok[tf]) {
                    u4v o;
                    o.x = pack2<BF16>(a2[0][0], a2[0][1]);
                    o.y = pack2<BF16>(a2[0][2], a2[0][3]);
                    o.z = pack2<BF16>(a2[1][0], a2[1][1]);
                    o.w = pack2<BF16>(a2[1][2], a2[1][3]);
                    *reinterpret_cast<u4v*>(ob + tf * otf + ((int64_t)half * CH + p * 32) * 2) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Backward row pass (round 4; aum_xdt_tm_bwd): the three token-parallel pieces of the x_proj / dt_proj gradients in ONE pass
//   dx_dbl[M][0:R] = ddelta[M][E] . W_dt[E][R]          SSI:587      (rounded to the activations' type, as the GEMM's output is)
//   dx_dbl[M][R:C] = dB | dC                            SSI:570-574  (the scan backward's fp32 rows)
//   du[M][E]      += dx_dbl[M][C] . W_x[C][E]           SSI:590      (in place)
// As three launches (a GEMM, a strided cast-copy, an addmm) they took 50 + 8 + 78 us at the bench shape for 300 MB of traffic
// (48 us at the copy rate).  The arrangement is the forward kernel's with the roles exchanged: stage A contracts the channels -- W_dt^T
// ([R][E], two K-halves through LDS) against ddelta rows fetched 32 bytes per lane four K-steps ahead -- into the wave's 16-token
// x tile in LDS, where the dB | dC columns join; dx_dbl leaves from the tile in 16-byte pieces; stage B streams W_x^T ([E][C], four
// channel quarters through LDS) against the tile and adds each lane's eight channels of a token into du with one 16-byte load and store.
// W_x^T rows sit at a 256-byte pitch with their 16-byte chunks XOR-ed by the row's position in its fragment (s = 4 (row / 8 % 4) + row % 4):
// the 16 rows a fragment read touches share their first bank, the XOR spreads them over all 64.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int WXP = 256;                                  // bytes per W_x^T row in LDS (80 columns = 10 chunks, swizzled inside 16)

template <bool BF16, int RF, int NW, int PD>
__global__ __launch_bounds__(NW * 64, 1) void k_xdt_tm_bwd(AumXdtBwdArgs g) {
    __shared__ __attribute__((aligned(16))) char lds[lds_bytes(XDT_MAX_DIM, NW)];
    constexpr int R = RF * 16, KS = 3;
    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rho = lane & 15, kg = lane >> 4;
    const int E = g.dim, KH = E / 2, SP = slab_pitch(KH);
    char* slab = lds;
    char* xt = lds + slab_bytes(XDT_MAX_DIM) + w * XT_BYTES;
    const int64_t t0 = (int64_t)blockIdx.x * (NW * XDT_TOK_W) + w * XDT_TOK_W;
    const s8v zero = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool tok_ok = t0 + rho < g.ntok;
    const char* drow = static_cast<const char*>(g.ddelta) + ((t0 + rho) * g.ldd + kg * 16) * 2;

    f4v acc[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) acc[f] = f4v{0.f, 0.f, 0.f, 0.f};
    // ---- stage A: dx_dbl[:, :R] = ddelta . W_dt ------------------------------------------------------------------------------
    auto load_d = [&](int k0, s8v (&df)[2]) {
        df[0] = tok_ok ? *reinterpret_cast<const s8v*>(drow + (int64_t)k0 * 2) : zero;
        df[1] = tok_ok ? *reinterpret_cast<const s8v*>(drow + (int64_t)k0 * 2 + 16) : zero;
    };
    const int nsteps = KH / 64, total = E / 64;
    s8v dr[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < total) load_d(j * 64, dr[j]);
    const char* wrd = slab + rho * SP + kg * 32;
    for (int s0 = 0; s0 < total; s0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = s0 + j;
            if (s == 0 || s == nsteps) {
                __syncthreads();
                stage_rows<NW * 64>(slab, SP, static_cast<const char*>(g.wdt_t) + (int64_t)(s / nsteps) * KH * 2, (int64_t)g.ldwdt * 2, R, KH / 8, tid);
                __syncthreads();
            }
            const int sl = s >= nsteps ? s - nsteps : s;
#pragma unroll
            for (int f = 0; f < RF; ++f) {
                const s8v w0 = *reinterpret_cast<const s8v*>(wrd + f * 16 * SP + sl * 128);
                const s8v w1 = *reinterpret_cast<const s8v*>(wrd + f * 16 * SP + sl * 128 + 16);
                acc[f] = mfma<BF16>(w0, dr[j][0], acc[f]);
                acc[f] = mfma<BF16>(w1, dr[j][1], acc[f]);
            }
            if (s + 4 < total) load_d((s + 4) * 64, dr[j]);
        }
    }
    // ---- the wave's dx_dbl tile [token][column]: the dt block rounded once, dB | dC behind it ---------------------------------------
#pragma unroll
    for (int f = 0; f < RF; ++f) {
        u2v v;
        v.x = pack2<BF16>(acc[f][0], acc[f][1]);
        v.y = pack2<BF16>(acc[f][2], acc[f][3]);
        *reinterpret_cast<u2v*>(xt + rho * XP + (f * 16 + kg * 4) * 2) = v;
    }
    {       // 16 tokens x (C - R) fp32 columns: C - R = 32 -> lane = (token lane / 4, eight columns (lane % 4) * 8)
        const int tk = lane >> 2, c8 = (lane & 3) * 8;
        u4v o = {0u, 0u, 0u, 0u};
        if (t0 + tk < g.ntok) {
            const float* src = g.dbc + (t0 + tk) * (int64_t)g.lddbc + c8;
            const f4v a = *reinterpret_cast<const f4v*>(src), b = *reinterpret_cast<const f4v*>(src + 4);
            o.x = pack2<BF16>(a[0], a[1]);
            o.y = pack2<BF16>(a[2], a[3]);
            o.z = pack2<BF16>(b[0], b[1]);
            o.w = pack2<BF16>(b[2], b[3]);
        }
        *reinterpret_cast<u4v*>(xt + tk * XP + (R + c8) * 2) = o;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);                              // lgkmcnt(0): the wave's own tile is complete (nobody else reads it)
    {
        char* xo = static_cast<char*>(g.dx_dbl);
        constexpr int PIECES = XDT_COLS * 2 / 16;
        for (int idx = lane; idx < XDT_TOK_W * PIECES; idx += 64) {
            const int tk = idx / PIECES, pc = idx - tk * PIECES;
            if (t0 + tk < g.ntok)
                *reinterpret_cast<u4v*>(xo + ((t0 + tk) * g.ldx) * 2 + pc * 16) = *reinterpret_cast<const u4v*>(xt + tk * XP + pc * 16);
        }
    }
    // ---- stage B: du += dx_dbl . W_x, the channels in four quarters of W_x^T through the slab ---------------------------------------
    s8v xf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = ks * 32 + kg * 8;
        xf[ks] = k < XDT_COLS ? *reinterpret_cast<const s8v*>(xt + rho * XP + k * 2) : zero;
    }
    const int CH = E / 4;
    char* ob = static_cast<char*>(g.du) + ((t0 + rho) * g.ldu + kg * 8) * 2;
    const char* wdr = slab + ((rho >> 2) * 8 + (rho & 3)) * WXP;
    int qoff[KS];                                                     // this lane's (swizzled) chunk of a row for every k-step
    bool q_ok[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int q = ks * 4 + kg;
        q_ok[ks] = q < XDT_COLS / 8;
        qoff[ks] = ((q ^ rho) & 15) * 16;
    }
    for (int quarter = 0; quarter < 4; ++quarter) {
        __syncthreads();
        {       // CH rows x 10 chunks of W_x^T -> swizzled rows of 256 bytes; eight loads in flight per thread
            const char* src = static_cast<const char*>(g.wx_t) + (int64_t)quarter * CH * g.ldwx * 2;
            constexpr int CHK = XDT_COLS / 8;
            const int totalc = CH * CHK;
            for (int base = tid; base < totalc; base += NW * 64 * 8) {
                u4v r[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = base + j * NW * 64;
                    if (idx < totalc) {
                        const int row = idx / CHK, ch = idx - row * CHK;
                        r[j] = *reinterpret_cast<const u4v*>(src + (int64_t)row * g.ldwx * 2 + ch * 16);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = base + j * NW * 64;
                    if (idx < totalc) {
                        const int row = idx / CHK, ch = idx - row * CHK;
                        const int sw = ((row >> 3) & 3) * 4 + (row & 3);
                        *reinterpret_cast<u4v*>(slab + row * WXP + ((ch ^ sw) & 15) * 16) = r[j];
                    }
                }
            }
        }
        __syncthreads();
        const int npairs = CH / 32;                                   // a multiple of PD: dim % 256 == 0 -> CH % 64 == 0 (PD 2), dim % 512 == 0 -> PD 4
        char* oq = ob + (int64_t)quarter * CH * 2;
        // this lane's 16 bytes of du for the next PD channel pairs are in flight while a pair is multiplied (two in flight, the first
        // version, left the read side of the read-modify-write at the memory latency)
        u4v dq[PD];
#pragma unroll
        for (int j = 0; j < PD; ++j) dq[j] = tok_ok ? *reinterpret_cast<const u4v*>(oq + j * 64) : u4v{0u, 0u, 0u, 0u};
        for (int p0 = 0; p0 < npairs; p0 += PD) {
#pragma unroll
            for (int pj = 0; pj < PD; ++pj) {
                const int p = p0 + pj;
                f4v a2[2] = {f4v{0.f, 0.f, 0.f, 0.f}, f4v{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const s8v wf = q_ok[ks] ? *reinterpret_cast<const s8v*>(wdr + (p * 32 + j * 4) * WXP + qoff[ks]) : zero;
                        a2[j] = mfma<BF16>(wf, xf[ks], a2[j]);
                    }
                if (tok_ok) {
                    const u4v d = dq[pj];
                    u4v o;
                    o.x = pack2<BF16>(a2[0][0] + lo16<BF16>(d.x), a2[0][1] + hi16<BF16>(d.x));
                    o.y = pack2<BF16>(a2[0][2] + lo16<BF16>(d.y), a2[0][3] + hi16<BF16>(d.y));
                    o.z = pack2<BF16>(a2[1][0] + lo16<BF16>(d.z), a2[1][1] + hi16<BF16>(d.z));
                    o.w = pack2<BF16>(a2[1][2] + lo16<BF16>(d.w), a2[1][3] + hi16<BF16>(d.w));
                    *reinterpret_cast<u4v*>(oq + p * 64) = o;
                    if (p + PD < npairs) dq[pj] = *reinterpret_cast<const u4v*>(oq + (p + PD) * 64);
                }
            }
        }
    }
}

}  // namespace aumx
