// proj_kernels.h -- the skinny projections around the scan, as MFMA kernels on channel-major activations.
//
// Reference (vim-mamba_ssm/mamba_ssm/ops/selective_scan_interface.py = "SSI"):
//   forward   SSI:467  x_dbl = conv1d_out^T @ W_x^T            (B*L, R+2N)     K = d_inner
//             SSI:468  delta = W_dt @ x_dbl[:, :R]^T           (d_inner, B*L)  K = R
//   backward  SSI:586  dW_dt = ddelta @ x_dbl[:, :R]           K = B*L
//             SSI:587  dx_dbl[:, :R] = ddelta^T @ W_dt         K = d_inner
//             SSI:589  dW_x = dx_dbl^T @ conv1d_out^T          K = B*L
//             SSI:590  dconv1d_out += W_x^T @ dx_dbl^T         K = R+2N
// With R+2N = 80 and R = 48 these are 5% of the block's GEMM flops and 35% of its GEMM time under hipBLASLt (skinny
// M or K): every one of them streams one (d_inner x B*L) activation and is HBM-bound, so they
// are written here as streaming kernels: 64 tokens per wavefront, v_mfma_f32_16x16x32 with TOKENS as the MFMA row
// dimension (a lane's 4 accumulator rows are 4 consecutive tokens -> 8-byte stores along the contiguous axis).
//
// Layouts (all "channel-major", token index t = b*L + l contiguous):
//   conv_out, delta, ddelta, dconv : [d_inner][ntok]       x_dbl, dx_dbl : [R+2N][ntok]  (rows: dt | B | C)
// so B and C of the scan are the row blocks of x_dbl (batch stride L, state stride ntok): no transposes, no slices.
//
// MFMA operand facts used (cdna4 16x16x32, 64 lanes): A[row = lane&15][k-group = lane>>4, 8 values],
// B[k-group = lane>>4, 8 values][col = lane&15], D[row = 4*(lane>>4) + reg][col = lane&15].  The order of the 8 k
// values inside a group is irrelevant as long as A and B agree (it is a sum), which the loaders below guarantee.
#pragma once
#include "wave.h"
#include "../../include/aum_hip.h"

namespace aum {

constexpr int PJ_TT = 64;        // tokens per wavefront tile
constexpr int PJ_PITCH = 68;     // LDS row pitch in 16-bit elements: 34 dwords -> the 4 k-groups of a gather land 16 banks apart
constexpr int PJ_KS = 64;        // channels staged per step (two MFMA k-extents)
constexpr int PJ_MAXRB = 5;      // (R + 2N) <= 80
constexpr int PJ_MAXRC = 3;      // k-chunks of 32 over R+2N (<= 96)
constexpr int PJ_XROWS = 96;
constexpr int PJ_NW = 4;         // wavefronts per workgroup (forward / data-gradient kernels)
constexpr int PJ_STAGE = PJ_KS * PJ_PITCH;                     // one staging buffer, 16-bit elements
constexpr int PJ_WPITCH = 72;     // weight-slice row pitch in LDS (144 B: 16-byte aligned rows, conflict-free ds_read_b128)
constexpr int PJ_FPITCH = 68;     // fp32 scratch row pitch (floats) of the data-gradient epilogue
constexpr int PJ_WSTAGE = 16 * PJ_MAXRB * PJ_WPITCH;            // one weight-slice buffer (80 rows x 64 channels)
constexpr int PJ_LDS_ELEMS = 2 * PJ_STAGE + 2 * PJ_WSTAGE + PJ_XROWS * PJ_PITCH;   // 53.5 KB per workgroup
constexpr int PJW_KS = 64;      // tokens per step of the weight-gradient kernel (two MFMA k-extents: full 128-byte lines per row)
constexpr int PJW_NW = 4;        // wavefronts per workgroup of the weight-gradient kernel (each owns 64 channel rows)
constexpr int PJW_EB = 2;        // 16-row channel blocks per wavefront
constexpr int PJW_ROWS = PJW_NW * PJW_EB * 16;                 // 256 channel rows per workgroup
constexpr int PJW_YPITCH = 72;                                 // y-slice row pitch in LDS (144 B: 16-byte aligned rows, conflict-free b128 reads)
constexpr int PJW_YSTAGE = 16 * PJ_MAXRB * PJW_YPITCH;          // one y-slice buffer (80 rows x 64 tokens)
constexpr int PJW_LDS_ELEMS = 2 * PJW_YSTAGE;                  // 23 KB per workgroup

#ifndef AUM_EMU
typedef short s8v __attribute__((ext_vector_type(8)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
using frag8 = s8v;
using quad16 = s4v;
using acc4 = f4v;
AUM_DEV frag8 frag_zero() { return s8v{0, 0, 0, 0, 0, 0, 0, 0}; }
AUM_DEV acc4 acc_zero() { return f4v{0.f, 0.f, 0.f, 0.f}; }
AUM_DEV vf acc_get(const acc4& a, int r) { return a[r]; }
AUM_DEV acc4 mfma16(bf16_t, frag8 a, frag8 b, acc4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}
AUM_DEV acc4 mfma16(f16_t, frag8 a, frag8 b, acc4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <class T> AUM_DEV frag8 gload_frag(const T* p, vi idx) {
    const pk16_t<2> raw = *reinterpret_cast<const pk16_t<2>*>(p + idx);
    return __builtin_bit_cast(s8v, raw);
}
template <class T> AUM_DEV frag8 gload_frag_n(const T* p, vi idx, vi n) {   // first min(n, 8) elements, rest zero
    if (n >= 8) return gload_frag(p, idx);
    frag8 r = frag_zero();
    AUM_UNROLL
    for (int j = 0; j < 8; ++j) if (j < n) r[j] = (short)p[idx + j].bits;
    return r;
}
template <class T> AUM_DEV void gstore_frag_n(T* p, vi idx, frag8 v, vi n) {
    if (n >= 8) {
        *reinterpret_cast<pk16_t<2>*>(p + idx) = __builtin_bit_cast(pk16_t<2>, v);
    } else {
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) if (j < n) p[idx + j].bits = (uint16_t)v[j];
    }
}
struct __attribute__((packed, aligned(2))) pk8_t { uint8_t b[8]; };
template <class T> AUM_DEV void gstore_quad_n(T* p, vi idx, quad16 v, vi n) {
    if (n >= 4) {
        *reinterpret_cast<pk8_t*>(p + idx) = __builtin_bit_cast(pk8_t, v);
    } else {
        AUM_UNROLL
        for (int j = 0; j < 4; ++j) if (j < n) p[idx + j].bits = (uint16_t)v[j];
    }
}
template <class T> AUM_DEV void gload_quad_n(const T* p, vi idx, vi n, vf (&o)[4]) {
    T e[4];
    if (n >= 4) {
        const pk8_t raw = *reinterpret_cast<const pk8_t*>(p + idx);
        __builtin_memcpy(e, &raw, 8);
    } else {
        AUM_UNROLL
        for (int j = 0; j < 4; ++j) e[j].bits = (j < n) ? p[idx + j].bits : (uint16_t)0;
    }
    AUM_UNROLL
    for (int j = 0; j < 4; ++j) o[j] = elem_to_f32(e[j]);
}
// fp32 -> 16-bit with the hardware converters (v_cvt_pk_bf16_f32 / v_cvt_pkrtz is NOT used: RNE like torch)
AUM_DEV quad16 cvt4(bf16_t, vf a, vf b, vf c, vf d) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(s4v, b4{(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d});
}
AUM_DEV quad16 cvt4(f16_t, vf a, vf b, vf c, vf d) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(s4v, h4{(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d});
}
// LDS holding 16-bit elements.  idx of the 4- and 8-element accesses is a multiple of 4 (8-byte aligned).
typedef uint64_t __attribute__((may_alias)) u64a_t;
AUM_DEV void lds16_write8(uint16_t* lds, vi idx, frag8 v) {
    typedef uint64_t u2 __attribute__((ext_vector_type(2)));
    const u2 q = __builtin_bit_cast(u2, v);
    *reinterpret_cast<u64a_t*>(lds + idx) = q[0];
    *reinterpret_cast<u64a_t*>(lds + idx + 4) = q[1];
}
AUM_DEV frag8 lds16_read8(const uint16_t* lds, vi idx) {
    typedef uint64_t u2 __attribute__((ext_vector_type(2)));
    const u2 q = {*reinterpret_cast<const u64a_t*>(lds + idx), *reinterpret_cast<const u64a_t*>(lds + idx + 4)};
    return __builtin_bit_cast(s8v, q);
}
// 16-byte aligned forms (idx % 8 == 0 on a 16-byte aligned array): one ds_write_b128 / ds_read_b128
AUM_DEV void lds16_write8_a(uint16_t* lds, vi idx, frag8 v) { *reinterpret_cast<s8v*>(lds + idx) = v; }
AUM_DEV frag8 lds16_read8_a(const uint16_t* lds, vi idx) { return *reinterpret_cast<const s8v*>(lds + idx); }
AUM_DEV void lds16_write4(uint16_t* lds, vi idx, quad16 v) { *reinterpret_cast<u64a_t*>(lds + idx) = __builtin_bit_cast(uint64_t, v); }
AUM_DEV void lds16_write1(uint16_t* lds, vi idx, vi bits) { lds[idx] = (uint16_t)bits; }
AUM_DEV frag8 lds16_gather8(const uint16_t* lds, vi idx0, int stride) {
    frag8 r;
    AUM_UNROLL
    for (int j = 0; j < 8; ++j) r[j] = (short)lds[idx0 + j * stride];
    return r;
}
AUM_DEV frag8 lds16_gather8_n(const uint16_t* lds, vi idx0, int stride, vi n) {   // all 8 reads in range; j >= n zeroed
    frag8 r;
    AUM_UNROLL
    for (int j = 0; j < 8; ++j) {
        const short v = (short)lds[idx0 + j * stride];
        r[j] = (j < n) ? v : (short)0;
    }
    return r;
}
template <class T> AUM_DEV void gstore_frag(T* p, vi idx, frag8 v) { *reinterpret_cast<pk16_t<2>*>(p + idx) = __builtin_bit_cast(pk16_t<2>, v); }
template <class T> AUM_DEV void gstore_quad(T* p, vi idx, quad16 v) { *reinterpret_cast<pk8_t*>(p + idx) = __builtin_bit_cast(pk8_t, v); }
template <class T> AUM_DEV void gload_quad(const T* p, vi idx, vf (&o)[4]) {
    const pk8_t raw = *reinterpret_cast<const pk8_t*>(p + idx);
    T e[4];
    __builtin_memcpy(e, &raw, 8);
    AUM_UNROLL
    for (int j = 0; j < 4; ++j) o[j] = elem_to_f32(e[j]);
}
template <class T> AUM_DEV vi f32_to_bits16(T, vf x) { T e; f32_to_elem(x, e); return (int)e.bits; }
#else
struct frag8 { uint16_t v[WAVE][8]; };
struct quad16 { uint16_t v[WAVE][4]; };
struct acc4 { float v[WAVE][4]; };
inline frag8 frag_zero() { frag8 r; std::memset(&r, 0, sizeof r); return r; }
inline acc4 acc_zero() { acc4 r; std::memset(&r, 0, sizeof r); return r; }
inline vf acc_get(const acc4& a, int r) { vf o; AUM_LANES o.v[l] = a.v[l][r]; return o; }
template <class T> inline acc4 mfma16(T, const frag8& a, const frag8& b, const acc4& c) {
    acc4 d = c;
    for (int i = 0; i < 16; ++i)
        for (int n = 0; n < 16; ++n) {
            double s = 0.0;
            for (int g = 0; g < 4; ++g)
                for (int j = 0; j < 8; ++j) {
                    T ea, eb;
                    ea.bits = a.v[i + 16 * g][j];
                    eb.bits = b.v[n + 16 * g][j];
                    s += (double)elem_to_f32(ea) * (double)elem_to_f32(eb);
                }
            d.v[n + 16 * (i / 4)][i % 4] = c.v[n + 16 * (i / 4)][i % 4] + (float)s;
        }
    return d;
}
template <class T> inline frag8 gload_frag_n(const T* p, const vi& idx, const vi& n) {
    frag8 r = frag_zero();
    AUM_LANES for (int j = 0; j < 8; ++j) if (j < n.v[l]) r.v[l][j] = p[idx.v[l] + j].bits;
    return r;
}
template <class T> inline frag8 gload_frag(const T* p, const vi& idx) { return gload_frag_n(p, idx, spl_i(8)); }
template <class T> inline void gstore_frag_n(T* p, const vi& idx, const frag8& v, const vi& n) {
    AUM_LANES for (int j = 0; j < 8; ++j) if (j < n.v[l]) p[idx.v[l] + j].bits = v.v[l][j];
}
template <class T> inline void gstore_quad_n(T* p, const vi& idx, const quad16& v, const vi& n) {
    AUM_LANES for (int j = 0; j < 4; ++j) if (j < n.v[l]) p[idx.v[l] + j].bits = v.v[l][j];
}
template <class T> inline void gload_quad_n(const T* p, const vi& idx, const vi& n, vf (&o)[4]) {
    for (int j = 0; j < 4; ++j) AUM_LANES o[j].v[l] = (j < n.v[l]) ? elem_to_f32(p[idx.v[l] + j]) : 0.f;
}
template <class T> inline quad16 cvt4(T, const vf& a, const vf& b, const vf& c, const vf& d) {
    quad16 r;
    AUM_LANES {
        T e;
        f32_to_elem(a.v[l], e); r.v[l][0] = e.bits;
        f32_to_elem(b.v[l], e); r.v[l][1] = e.bits;
        f32_to_elem(c.v[l], e); r.v[l][2] = e.bits;
        f32_to_elem(d.v[l], e); r.v[l][3] = e.bits;
    }
    return r;
}
inline void lds16_write8(uint16_t* lds, const vi& idx, const frag8& v) { AUM_LANES for (int j = 0; j < 8; ++j) lds[idx.v[l] + j] = v.v[l][j]; }
inline frag8 lds16_read8(const uint16_t* lds, const vi& idx) { frag8 r; AUM_LANES for (int j = 0; j < 8; ++j) r.v[l][j] = lds[idx.v[l] + j]; return r; }
inline void lds16_write8_a(uint16_t* lds, const vi& idx, const frag8& v) { lds16_write8(lds, idx, v); }
inline frag8 lds16_read8_a(const uint16_t* lds, const vi& idx) { return lds16_read8(lds, idx); }
inline void lds16_write4(uint16_t* lds, const vi& idx, const quad16& v) { AUM_LANES for (int j = 0; j < 4; ++j) lds[idx.v[l] + j] = v.v[l][j]; }
inline void lds16_write1(uint16_t* lds, const vi& idx, const vi& bits) { AUM_LANES lds[idx.v[l]] = (uint16_t)bits.v[l]; }
inline frag8 lds16_gather8_n(const uint16_t* lds, const vi& idx0, int stride, const vi& n) {
    frag8 r = frag_zero();
    AUM_LANES for (int j = 0; j < 8; ++j) if (j < n.v[l]) r.v[l][j] = lds[idx0.v[l] + j * stride];
    return r;
}
inline frag8 lds16_gather8(const uint16_t* lds, const vi& idx0, int stride) { return lds16_gather8_n(lds, idx0, stride, spl_i(8)); }
template <class T> inline void gstore_frag(T* p, const vi& idx, const frag8& v) { gstore_frag_n(p, idx, v, spl_i(8)); }
template <class T> inline void gstore_quad(T* p, const vi& idx, const quad16& v) { gstore_quad_n(p, idx, v, spl_i(4)); }
template <class T> inline void gload_quad(const T* p, const vi& idx, vf (&o)[4]) { gload_quad_n(p, idx, spl_i(4), o); }
template <class T> inline vi f32_to_bits16(T, const vf& x) { vi r; AUM_LANES { T e; f32_to_elem(x.v[l], e); r.v[l] = e.bits; } return r; }
#endif

AUM_DEV vi clamp_i(vi x, int lo, int hi) { return vmin_i(vmax_i(x, lo), hi); }
#ifndef AUM_EMU
template <class T> AUM_DEV void frag_to_f32(T, frag8 v, vf (&o)[8]) {
    AUM_UNROLL
    for (int j = 0; j < 8; ++j) { T e; e.bits = (uint16_t)v[j]; o[j] = elem_to_f32(e); }
}
template <class T> AUM_DEV frag8 f32_to_frag(T t, const vf (&x)[8]) {
    const quad16 a = cvt4(t, x[0], x[1], x[2], x[3]), b = cvt4(t, x[4], x[5], x[6], x[7]);
    return s8v{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
#else
template <class T> inline void frag_to_f32(T, const frag8& v, vf (&o)[8]) {
    for (int j = 0; j < 8; ++j) AUM_LANES { T e; e.bits = v.v[l][j]; o[j].v[l] = elem_to_f32(e); }
}
template <class T> inline frag8 f32_to_frag(T t, const vf (&x)[8]) {
    const quad16 a = cvt4(t, x[0], x[1], x[2], x[3]), b = cvt4(t, x[4], x[5], x[6], x[7]);
    frag8 r;
    AUM_LANES for (int j = 0; j < 4; ++j) { r.v[l][j] = a.v[l][j]; r.v[l][4 + j] = b.v[l][j]; }
    return r;
}
#endif

// FULL (wave-uniform template flag) = every token of the tile exists: plain vector accesses, no per-lane branches.
// Edge tiles (the last workgroup when ntok % 64 != 0) take the masked element-wise forms.
template <bool FULL, class T> AUM_DEV frag8 gload_frag_f(const T* p, vi idx, vi n) {
    if constexpr (FULL) return gload_frag(p, idx); else return gload_frag_n(p, idx, n);
}
template <bool FULL, class T> AUM_DEV void gstore_frag_f(T* p, vi idx, const frag8& v, vi n) {
    if constexpr (FULL) gstore_frag(p, idx, v); else gstore_frag_n(p, idx, v, n);
}
template <bool FULL, class T> AUM_DEV void gstore_quad_f(T* p, vi idx, const quad16& v, vi n) {
    if constexpr (FULL) gstore_quad(p, idx, v); else gstore_quad_n(p, idx, v, n);
}
template <bool FULL, class T> AUM_DEV void gload_quad_f(const T* p, vi idx, vi n, vf (&o)[4]) {
    if constexpr (FULL) gload_quad(p, idx, o); else gload_quad_n(p, idx, n, o);
}

// ---- shared pieces ----------------------------------------------------------------------------------------------
// Forward and data-gradient kernels: one 4-wavefront workgroup per 64-token tile.
//   K loop (channels): the four waves stage 64 channel rows x 64 tokens per step into a double-buffered LDS tile (two
//   16-byte global loads per lane, every row segment one full 128-byte line, prefetched two steps ahead in registers),
//   then wave w takes token block w (16 tokens): two A fragments gathered from the tile + 2*NCB weight fragments from
//   L1/L2 (prefetched one step ahead) -> 2*NCB MFMAs.  One workgroup barrier per step.
//   Epilogue GEMM (K = dt_rank or R+2N): wave w takes channel blocks eb = w, w+4, ... for all 64 tokens, so the four
//   8-byte stores of a lane's accumulator rows complete whole 128-byte lines of the output.
// Per-wave state that lives across barrier-separated phases is declared with AUM_PER_WAVE / indexed with AUM_W (wave.h).
// Weight fragments are loaded unconditionally from clamped (in-bounds) addresses: columns past the matrix produce
// accumulator columns that are never stored, k positions past the matrix meet zeros in the A fragment.
AUM_DEV vi pj_tok_valid8(int64_t ntok, int tok0) { return clamp_i(spl_i((int)ntok - tok0) - (lane_id() & 7) * 8, 0, 8); }

// staging slot of a lane: rows 8w + (lane>>3) + 32i (i = 0, 1), tokens 8*(lane&7)..+7
template <bool FULL, class T>
AUM_DEV void pj_stage_load(const T* act, int64_t ntok, int ch0, int tok0, int w, frag8 (&st)[2]) {
    const vi lane = lane_id();
    const vi ntv = pj_tok_valid8(ntok, tok0);
    AUM_UNROLL
    for (int i = 0; i < 2; ++i) {
        const T* base = act + (int64_t)(ch0 + w * 8 + 32 * i) * ntok + tok0;    // wave-uniform 64-bit part of the address
        st[i] = gload_frag_f<FULL>(base, (lane >> 3) * (int)ntok + (lane & 7) * 8, ntv);
    }
}
AUM_DEV void pj_stage_store(uint16_t* buf, int w, const frag8 (&st)[2]) {
    const vi lane = lane_id();
    AUM_UNROLL
    for (int i = 0; i < 2; ++i) lds16_write8(buf, ((lane >> 3) + w * 8 + 32 * i) * PJ_PITCH + (lane & 7) * 8, st[i]);
}
// The weight slice of a step (16*NCB rows = output columns, 64 channels, channels contiguous in global memory) is loaded
// ONCE per workgroup -- every row one 128-byte line, 16 bytes per lane, (NCB+1)/2 loads per lane -- and shared through
// LDS: per-wave fragment loads straight from L1/L2 cost 4x the traffic and were 65 of the kernel's 127 us.
template <class T, int NCB> AUM_DEV void pj_w_load(const T* wmat, int pitch, int ch0, int ncols, int w, frag8 (&wr)[(NCB + 1) / 2]) {
    const vi lane = lane_id();
    AUM_UNROLL
    for (int i = 0; i < (NCB + 1) / 2; ++i) {
        const int qbase = w * 64 + 256 * i;                 // wave-uniform: a wave is entirely inside or outside the slice
        wr[i] = frag_zero();
        if (qbase < NCB * 128) {
            const vi q = lane + qbase;
            wr[i] = gload_frag(wmat, vmin_i(q >> 3, ncols - 1) * pitch + (q & 7) * 8 + ch0);
        }
    }
}
template <int NCB> AUM_DEV void pj_w_store(uint16_t* wbuf, int w, const frag8 (&wr)[(NCB + 1) / 2]) {
    const vi lane = lane_id();
    AUM_UNROLL
    for (int i = 0; i < (NCB + 1) / 2; ++i) {
        const int qbase = w * 64 + 256 * i;
        if (qbase < NCB * 128) {
            const vi q = lane + qbase;
            lds16_write8_a(wbuf, (q >> 3) * PJ_WPITCH + (q & 7) * 8, wr[i]);
        }
    }
}
template <class T, int NCB> AUM_DEV void pj_mma(const uint16_t* buf, const uint16_t* wbuf, int w, acc4 (&acc)[NCB]) {
    const vi lane = lane_id();
    const vi t16 = lane & 15, g = lane >> 4;
    const vi base = (g * 8) * PJ_PITCH + t16 + w * 16;
    const frag8 a0 = lds16_gather8(buf, base, PJ_PITCH);
    const frag8 a1 = lds16_gather8(buf, base + 32 * PJ_PITCH, PJ_PITCH);
    AUM_UNROLL
    for (int cb = 0; cb < NCB; ++cb) {
        const vi wo = (t16 + cb * 16) * PJ_WPITCH + g * 8;
        acc[cb] = mfma16(T{}, a0, lds16_read8_a(wbuf, wo), acc[cb]);
        acc[cb] = mfma16(T{}, a1, lds16_read8_a(wbuf, wo + 32), acc[cb]);
    }
}

// acc[w][cb] (tokens 16w..16w+15  x  columns 16cb..) = sum over all channels of act^T[token][ch] * W[col][ch].
// Software pipeline: activation rows are loaded two steps ahead (register sets sa / sb alternate), the weight slice one
// step ahead; both are published to the other half of their double-buffered LDS tiles one step before use.
template <bool FULL, class T, int NCB>
AUM_DEV void pj_reduce_channels(const T* act, int64_t ntok, int dim, int tok0, const T* wmat, int ncols, uint16_t* stage,
                                uint16_t* wstage, acc4 (&acc)[AUM_PER_WAVE(PJ_NW)][NCB]) {
    const int nks = dim / PJ_KS;
    frag8 sa[AUM_PER_WAVE(PJ_NW)][2], sb[AUM_PER_WAVE(PJ_NW)][2];
    frag8 wr[AUM_PER_WAVE(PJ_NW)][(NCB + 1) / 2];
    AUM_FOR_EACH_WAVE(w, PJ_NW) {
        AUM_UNROLL
        for (int cb = 0; cb < NCB; ++cb) acc[AUM_W(w)][cb] = acc_zero();
        frag8 s0[2];
        pj_stage_load<FULL>(act, ntok, 0, tok0, w, s0);
        pj_stage_store(stage, w, s0);
        pj_w_load<T, NCB>(wmat, dim, 0, ncols, w, wr[AUM_W(w)]);
        pj_w_store<NCB>(wstage, w, wr[AUM_W(w)]);
        if (nks > 1) pj_stage_load<FULL>(act, ntok, PJ_KS, tok0, w, sa[AUM_W(w)]);
        if (nks > 2) pj_stage_load<FULL>(act, ntok, 2 * PJ_KS, tok0, w, sb[AUM_W(w)]);
        if (nks > 1) pj_w_load<T, NCB>(wmat, dim, PJ_KS, ncols, w, wr[AUM_W(w)]);
    }
    AUM_WG_BARRIER();
    // one step: publish the rows and the weight slice of step KS+1, fetch the rows of step KS+3 and the weights of step
    // KS+2, multiply step KS.  C1 / C2 / C3 are the "step KS+1 / KS+2 / KS+3 exists" conditions: literal `true` in the
    // steady-state loop (straight-line code, so the compiler's s_waitcnt counts stay exact and the prefetches stay in
    // flight across the MFMAs), runtime in the tail.
#define PJ_STEP(KS, SCUR, C1, C2, C3)                                                                              \
        AUM_FOR_EACH_WAVE(w, PJ_NW) {                                                                              \
            const int cb_ = (KS) & 1, nb_ = ((KS) + 1) & 1;                                                        \
            if (C1) pj_stage_store(stage + nb_ * PJ_STAGE, w, SCUR[AUM_W(w)]);                                     \
            if (C1) pj_w_store<NCB>(wstage + nb_ * PJ_WSTAGE, w, wr[AUM_W(w)]);                                    \
            if (C3) pj_stage_load<FULL>(act, ntok, ((KS) + 3) * PJ_KS, tok0, w, SCUR[AUM_W(w)]);                   \
            if (C2) pj_w_load<T, NCB>(wmat, dim, ((KS) + 2) * PJ_KS, ncols, w, wr[AUM_W(w)]);                      \
            pj_mma<T, NCB>(stage + cb_ * PJ_STAGE, wstage + cb_ * PJ_WSTAGE, w, acc[AUM_W(w)]);                    \
        }                                                                                                          \
        AUM_WG_BARRIER_LDS();
    int ks = 0;
    for (; ks + 6 <= nks; ks += 2) {
        PJ_STEP(ks, sa, true, true, true)
        PJ_STEP(ks + 1, sb, true, true, true)
    }
    for (; ks < nks; ks += 2) {
        PJ_STEP(ks, sa, ks + 1 < nks, ks + 2 < nks, ks + 3 < nks)
        if (ks + 1 < nks) { PJ_STEP(ks + 1, sb, ks + 2 < nks, ks + 3 < nks, ks + 4 < nks) }
    }
#undef PJ_STEP
}

// wave w's accumulators (tokens 16w.. x cols) -> LDS tile rows = cols, 16-bit, rounded once
template <class T, int NCB> AUM_DEV void pj_acc_to_tile(uint16_t* xt, int w, const acc4 (&acc)[NCB]) {
    const vi lane = lane_id();
    const vi t16 = lane & 15, g = lane >> 4;
    AUM_UNROLL
    for (int cb = 0; cb < NCB; ++cb) {
        const acc4& d = acc[cb];
        lds16_write4(xt, (t16 + cb * 16) * PJ_PITCH + g * 4 + w * 16,
                     cvt4(T{}, acc_get(d, 0), acc_get(d, 1), acc_get(d, 2), acc_get(d, 3)));
    }
}
// LDS tile rows [0, nrows) -> global [nrows][ntok] at token tok0: wave w takes rows 8w + 32i + (lane>>3), 16 bytes per lane
template <bool FULL, class T> AUM_DEV void pj_tile_to_global(const uint16_t* xt, T* dst, int64_t ntok, int tok0, int nrows, int w) {
    const vi lane = lane_id();
    const vi srow = lane >> 3, stok = (lane & 7) * 8;
    const vi ntv = pj_tok_valid8(ntok, tok0);
    for (int r0 = w * 8; r0 < nrows; r0 += 8 * PJ_NW) {
        const vi row = srow + r0;
        const frag8 v = lds16_read8(xt, vmin_i(row, PJ_XROWS - 1) * PJ_PITCH + stok);
        T* base = dst + (int64_t)r0 * ntok + tok0;
        if (FULL && r0 + 8 <= nrows) gstore_frag(base, srow * (int)ntok + stok, v);
        else gstore_frag_n(base, srow * (int)ntok + stok, v, vsel_i(row < nrows, ntv, spl_i(0)));
    }
}
// A fragments (16 tokens of block nb x 32 rows of the tile) for the epilogue GEMM; rows >= nrows read as zero
AUM_DEV frag8 pj_tile_frag(const uint16_t* xt, int chunk, int nrows, int nb) {
    const vi lane = lane_id();
    const vi row0 = (lane >> 4) * 8 + chunk * 32;
    const vi n = clamp_i(spl_i(nrows) - row0, 0, 8);
    return lds16_gather8_n(xt, vmin_i(row0, PJ_XROWS - 8) * PJ_PITCH + (lane & 15) + nb * 16, PJ_PITCH, n);
}

// ---- forward: x_dbl = W_x conv_out ; delta = W_dt x_dbl[:R] -----------------------------------------------------
// NCB = ceil((R+2N)/16) column blocks of x_dbl, TWO = (R > 32): compile-time so every loop below is straight-line
template <bool FULL, class T, int NCB, bool TWO> AUM_DEV void proj_fwd_tile(const AumProjArgs& p, int wg, uint16_t* lds) {
    uint16_t* stage = lds;
    uint16_t* wstage = lds + 2 * PJ_STAGE;
    uint16_t* xt = lds + 2 * PJ_STAGE + 2 * PJ_WSTAGE;
    const int64_t ntok = p.ntok;
    const int tok0 = wg * PJ_TT, E = p.dim, R = p.dt_rank, RT = p.dt_rank + 2 * p.dstate;
    const T* w_dt = static_cast<const T*>(p.w_dt);
    T* delta = static_cast<T*>(p.out_act);

    acc4 acc[AUM_PER_WAVE(PJ_NW)][NCB];
    pj_reduce_channels<FULL, T, NCB>(static_cast<const T*>(p.act), ntok, E, tok0, static_cast<const T*>(p.w_x), RT, stage, wstage, acc);
    AUM_FOR_EACH_WAVE(w, PJ_NW) { pj_acc_to_tile<T, NCB>(xt, w, acc[AUM_W(w)]); }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, PJ_NW) {
        const vi lane = lane_id();
        const vi t16 = lane & 15, g = lane >> 4;
        pj_tile_to_global<FULL>(xt, static_cast<T*>(p.x_dbl), ntok, tok0, RT, w);
        // delta^T[token][e] = sum_r dt^T[token][r] * W_dt[e][r]
        constexpr bool two = TWO;
        frag8 adt[2][4];
        AUM_UNROLL
        for (int nb = 0; nb < 4; ++nb) {
            adt[0][nb] = pj_tile_frag(xt, 0, R, nb);
            adt[1][nb] = two ? pj_tile_frag(xt, 1, R, nb) : frag_zero();
        }
        const int neb = E / 16;
        // the accumulator layout (4 consecutive tokens of ONE channel per lane) would store 32-byte pieces; each block is
        // turned through a wave-private LDS scratch (the staging area, free after the K loop) into 16 bytes per lane,
        // 8 lanes per 128-byte line
        uint16_t* scr = stage + w * (16 * PJ_WPITCH);
        const vi srow = lane >> 3, stok = (lane & 7) * 8;
        const vi ntv = pj_tok_valid8(ntok, tok0);
        const int LD = p.w_ld;                                        // % 8 == 0: a lane's 8 k values never straddle a row
        const int wmax = E * LD - 8;                                  // lanes whose k group lies past dt_rank (zero A) read here
        // weights of the next channel block are loaded while this one is multiplied; ping-pong sets, no copies
        frag8 ba[2], bb[2];
#define PJ_LOAD_WDT(dst, EB)                                                                 \
        dst[0] = gload_frag(w_dt, vmin_i((t16 + (EB) * 16) * LD + g * 8, wmax));             \
        dst[1] = two ? gload_frag(w_dt, vmin_i((t16 + (EB) * 16) * LD + 32 + g * 8, wmax)) : frag_zero();
#define PJ_DELTA_BLOCK(EB, BCUR, BNXT)                                                                               \
        {                                                                                                            \
            if ((EB) + PJ_NW < neb) { PJ_LOAD_WDT(BNXT, (EB) + PJ_NW) }                                              \
            AUM_UNROLL                                                                                               \
            for (int nb = 0; nb < 4; ++nb) {                                                                         \
                acc4 d = mfma16(T{}, adt[0][nb], BCUR[0], acc_zero());                                               \
                if (two) d = mfma16(T{}, adt[1][nb], BCUR[1], d);                                                    \
                lds16_write4(scr, t16 * PJ_WPITCH + g * 4 + nb * 16, cvt4(T{}, acc_get(d, 0), acc_get(d, 1), acc_get(d, 2), acc_get(d, 3))); \
            }                                                                                                        \
            wave_lds_fence();                                                                                        \
            T* drow = delta + (int64_t)(EB) * 16 * ntok + tok0;                                                      \
            AUM_UNROLL                                                                                               \
            for (int i = 0; i < 2; ++i)                                                                              \
                gstore_frag_f<FULL>(drow, (srow + 8 * i) * (int)ntok + stok, lds16_read8_a(scr, (srow + 8 * i) * PJ_WPITCH + stok), ntv); \
            wave_lds_fence();                                                                                        \
        }
        PJ_LOAD_WDT(ba, w)
        for (int eb = w; eb < neb; eb += 2 * PJ_NW) {
            PJ_DELTA_BLOCK(eb, ba, bb)
            if (eb + PJ_NW < neb) PJ_DELTA_BLOCK(eb + PJ_NW, bb, ba)
        }
#undef PJ_LOAD_WDT
#undef PJ_DELTA_BLOCK
    }
}
template <class T, int NCB, bool TWO> AUM_DEV void proj_fwd_wg(const AumProjArgs& p, int wg, uint16_t* lds) {
    if ((int64_t)(wg + 1) * PJ_TT <= p.ntok) proj_fwd_tile<true, T, NCB, TWO>(p, wg, lds);
    else proj_fwd_tile<false, T, NCB, TWO>(p, wg, lds);
}

// ---- backward, data: dx_dbl[:R] = W_dt^T ddelta ; dx_dbl[R:] = (dB | dC) ; dconv += W_x^T dx_dbl ---------------------
// NDB = ceil(R/16) column blocks of the dt gradient, NRC = ceil((R+2N)/32) k-chunks of the second GEMM
template <bool FULL, class T, int NDB, int NRC> AUM_DEV void proj_bwd_data_tile(const AumProjArgs& p, int wg, uint16_t* lds) {
    uint16_t* stage = lds;
    uint16_t* wstage = lds + 2 * PJ_STAGE;
    uint16_t* xt = lds + 2 * PJ_STAGE + 2 * PJ_WSTAGE;
    const int64_t ntok = p.ntok;
    const int tok0 = wg * PJ_TT, E = p.dim, R = p.dt_rank, N = p.dstate, RT = R + 2 * N, L = p.len;
    const T* w_xT = static_cast<const T*>(p.w_x);            // [E][w_ld]
    T* dconv = static_cast<T*>(p.out_act);

    acc4 acc[AUM_PER_WAVE(PJ_NW)][NDB];
    pj_reduce_channels<FULL, T, NDB>(static_cast<const T*>(p.act), ntok, E, tok0, static_cast<const T*>(p.w_dt), R, stage, wstage, acc);
    // rows 0..16*ceil(R/16)-1 from the accumulators (columns >= R hold don't-care values), then rows R..RT-1 = dB | dC on
    // top of them; rows >= RT are never read back unmasked
    AUM_FOR_EACH_WAVE(w, PJ_NW) { pj_acc_to_tile<T, NDB>(xt, w, acc[AUM_W(w)]); }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, PJ_NW) {   // fp32 (batch, dstate, len) -> 16-bit, one token per lane, states n = w, w+4, ...
        const vi lane = lane_id();
        const vi tok = lane + tok0;
        const vm ok = tok < (int)ntok;
        const vi tc = vmin_i(tok, (int)ntok - 1);
        const vi b = tc / L, l = tc - b * L;
        for (int n = w; n < N; n += PJ_NW) {
            const vf vb = gload_u(p.dB, b * (int)p.dB_bs + n * (int)p.dB_ns + l);
            const vf vc = gload_u(p.dC, b * (int)p.dC_bs + n * (int)p.dC_ns + l);
            lds16_write1(xt, lane + (R + n) * PJ_PITCH, f32_to_bits16(T{}, vsel(ok, vb, splat(0.f))));
            lds16_write1(xt, lane + (R + N + n) * PJ_PITCH, f32_to_bits16(T{}, vsel(ok, vc, splat(0.f))));
        }
    }
    AUM_WG_BARRIER();
    AUM_FOR_EACH_WAVE(w, PJ_NW) {
        const vi lane = lane_id();
        const vi t16 = lane & 15, g = lane >> 4;
        pj_tile_to_global<FULL>(xt, static_cast<T*>(p.x_dbl), ntok, tok0, RT, w);
        frag8 ax[NRC][4];
        AUM_UNROLL
        for (int c = 0; c < NRC; ++c)
            AUM_UNROLL
            for (int nb = 0; nb < 4; ++nb) ax[c][nb] = pj_tile_frag(xt, c, RT, nb);
        const int neb = E / 16;
        // fp32 accumulator blocks are turned through a wave-private LDS scratch (the staging area, free after the K loop)
        // so that dconv is read and written 16 bytes per lane, 8 lanes per 128-byte line, and rounded once
        float* scrf = reinterpret_cast<float*>(stage) + w * (16 * PJ_FPITCH);
        const vi srow = lane >> 3, stok = (lane & 7) * 8;
        const vi ntv = pj_tok_valid8(ntok, tok0);
        const int LD = p.w_ld;
        const int wmax = E * LD - 8;
        frag8 bw[NRC], bn[NRC];
#define PJ_LOAD_WX(dst, EB)                                                                                          \
        AUM_UNROLL                                                                                                   \
        for (int c = 0; c < NRC; ++c)                                                                                \
            dst[c] = gload_frag(w_xT, vmin_i((t16 + (EB) * 16) * LD + c * 32 + g * 8, wmax));
#define PJ_DCONV_BLOCK(EB, BCUR, BNXT)                                                                               \
        {                                                                                                            \
            if ((EB) + PJ_NW < neb) { PJ_LOAD_WX(BNXT, (EB) + PJ_NW) }                                               \
            T* drow = dconv + (int64_t)(EB) * 16 * ntok + tok0;                                                      \
            frag8 old[2];                                                                                            \
            AUM_UNROLL                                                                                               \
            for (int i = 0; i < 2; ++i) old[i] = gload_frag_f<FULL>(drow, (srow + 8 * i) * (int)ntok + stok, ntv);   \
            AUM_UNROLL                                                                                               \
            for (int nb = 0; nb < 4; ++nb) {                                                                         \
                acc4 d = acc_zero();                                                                                 \
                AUM_UNROLL                                                                                           \
                for (int c = 0; c < NRC; ++c) d = mfma16(T{}, ax[c][nb], BCUR[c], d);                                \
                AUM_UNROLL                                                                                           \
                for (int q = 0; q < 4; ++q) lds_write(scrf, t16 * PJ_FPITCH + g * 4 + nb * 16 + q, acc_get(d, q));   \
            }                                                                                                        \
            wave_lds_fence();                                                                                        \
            AUM_UNROLL                                                                                               \
            for (int i = 0; i < 2; ++i) {                                                                            \
                vf o[8], r[8];                                                                                       \
                frag_to_f32(T{}, old[i], o);                                                                         \
                AUM_UNROLL                                                                                           \
                for (int j = 0; j < 8; ++j) r[j] = o[j] + lds_read(scrf, (srow + 8 * i) * PJ_FPITCH + stok + j);     \
                gstore_frag_f<FULL>(drow, (srow + 8 * i) * (int)ntok + stok, f32_to_frag(T{}, r), ntv);              \
            }                                                                                                        \
            wave_lds_fence();                                                                                        \
        }
        PJ_LOAD_WX(bw, w)
        for (int eb = w; eb < neb; eb += 2 * PJ_NW) {
            PJ_DCONV_BLOCK(eb, bw, bn)
            if (eb + PJ_NW < neb) PJ_DCONV_BLOCK(eb + PJ_NW, bn, bw)
        }
#undef PJ_DCONV_BLOCK
#undef PJ_LOAD_WX
    }
}
template <class T, int NDB, int NRC> AUM_DEV void proj_bwd_data_wg(const AumProjArgs& p, int wg, uint16_t* lds) {
    if ((int64_t)(wg + 1) * PJ_TT <= p.ntok) proj_bwd_data_tile<true, T, NDB, NRC>(p, wg, lds);
    else proj_bwd_data_tile<false, T, NDB, NRC>(p, wg, lds);
}

// ---- backward, weights: out[e][r] (or out[r][e]) = sum_t X[e][t] * Y[r][t], split over tokens ------------------------
// One workgroup = 256 channel rows (4 waves x 4 sixteen-row blocks) x one token range.  Per 32-token step the y slice
// ([<= 80 rows][32 tokens], the operand every channel block needs) is loaded ONCE per workgroup into a double-buffered LDS
// tile and read as fragments by all four waves (per-wave y fragments from L1/L2 were 126 MB of extra L2 traffic next to
// the 100 MB x stream); the x fragments of a wave (its own rows, 8 consecutive tokens per lane) come straight from
// global memory, two steps ahead in ping-pong register sets.  One barrier per step, one fp32 partial per workgroup.
template <class T> AUM_DEV void pjw_y_load(const T* Y, int64_t ntok, int nr, int64_t t0, int64_t t_end, int w, frag8 (&yr)[3]) {
    const vi lane = lane_id();
    AUM_UNROLL
    for (int i = 0; i < 3; ++i) {
        const int qbase = w * 64 + 256 * i;                     // lane-load q: row q>>3, tokens 8*(q&7)..+7 of the step
        yr[i] = frag_zero();
        if (qbase < 16 * PJ_MAXRB * 8) {
            const vi q = lane + qbase;
            const vi row = q >> 3;
            const vi n = clamp_i(spl_i((int)(t_end - t0)) - (q & 7) * 8, 0, 8);
            if (t0 + PJW_KS <= t_end) yr[i] = gload_frag_n(Y + t0, vmin_i(row, nr - 1) * (int)ntok + (q & 7) * 8, vsel_i(row < nr, spl_i(8), spl_i(0)));
            else yr[i] = gload_frag_n(Y + t0, vmin_i(row, nr - 1) * (int)ntok + (q & 7) * 8, vsel_i(row < nr, n, spl_i(0)));
        }
    }
}
AUM_DEV void pjw_y_store(uint16_t* ybuf, int w, const frag8 (&yr)[3]) {
    const vi lane = lane_id();
    AUM_UNROLL
    for (int i = 0; i < 3; ++i) {
        const int qbase = w * 64 + 256 * i;
        if (qbase < 16 * PJ_MAXRB * 8) {
            const vi q = lane + qbase;
            lds16_write8_a(ybuf, (q >> 3) * PJW_YPITCH + (q & 7) * 8, yr[i]);
        }
    }
}
template <class T> AUM_DEV void pjw_x_load(const T* xb, int64_t ntok, int rows_left, int64_t t0, int64_t t_end, frag8 (&xa)[2 * PJW_EB]) {
    const vi lane = lane_id();
    const vi t16 = lane & 15, g = lane >> 4;
    AUM_UNROLL
    for (int eb = 0; eb < PJW_EB; ++eb) {
        AUM_UNROLL
        for (int c = 0; c < 2; ++c) {                       // the two 32-token k-extents of the step: together one 128-byte line per row
            xa[2 * eb + c] = frag_zero();
            if (eb * 16 < rows_left) {
                const vi off = (eb * 16 + t16) * (int)ntok + g * 8 + 32 * c;
                if (t0 + PJW_KS <= t_end) xa[2 * eb + c] = gload_frag(xb + t0, off);
                else xa[2 * eb + c] = gload_frag_n(xb + t0, off, clamp_i(spl_i((int)(t_end - t0)) - g * 8 - 32 * c, 0, 8));
            }
        }
    }
}
template <class T>
AUM_DEV void proj_bwd_weight_wg(const AumProjWArgs& p, int chunk, int wg, uint16_t* lds) {
    const int64_t ntok = p.ntok;
    const int nec = (p.dim + PJW_ROWS - 1) / PJW_ROWS;
    const int ec = wg % nec, split = wg / nec;
    const int nr = p.nrows, nrb = (nr + 15) / 16;
    const T* X = static_cast<const T*>(p.x);
    const T* Y = static_cast<const T*>(p.y);
    const int64_t t_begin = (int64_t)split * chunk;
    const int64_t t_end = t_begin + chunk < ntok ? t_begin + chunk : ntok;
    const int nsteps = t_begin < t_end ? (int)((t_end - t_begin + PJW_KS - 1) / PJW_KS) : 0;
    acc4 acc[AUM_PER_WAVE(PJW_NW)][PJW_EB][PJ_MAXRB];
    frag8 xa[AUM_PER_WAVE(PJW_NW)][2 * PJW_EB], xb2[AUM_PER_WAVE(PJW_NW)][2 * PJW_EB];
    frag8 yr[AUM_PER_WAVE(PJW_NW)][3];
    AUM_FOR_EACH_WAVE(w, PJW_NW) {
        const int row0 = ec * PJW_ROWS + w * 16 * PJW_EB;          // this wave's first channel
        const T* xb = X + (int64_t)(row0 < p.dim ? row0 : 0) * ntok;
        const int rows_left = p.dim - row0;
        AUM_UNROLL
        for (int eb = 0; eb < PJW_EB; ++eb)
            AUM_UNROLL
            for (int rb = 0; rb < PJ_MAXRB; ++rb) acc[AUM_W(w)][eb][rb] = acc_zero();
        if (nsteps > 0) {
            pjw_y_load(Y, ntok, nr, t_begin, t_end, w, yr[AUM_W(w)]);
            pjw_y_store(lds, w, yr[AUM_W(w)]);
            pjw_x_load(xb, ntok, rows_left, t_begin, t_end, xa[AUM_W(w)]);
            if (nsteps > 1) {
                pjw_y_load(Y, ntok, nr, t_begin + PJW_KS, t_end, w, yr[AUM_W(w)]);
                pjw_x_load(xb, ntok, rows_left, t_begin + PJW_KS, t_end, xb2[AUM_W(w)]);
            }
        }
    }
    AUM_WG_BARRIER();
    // step KS: publish y(KS+1), fetch y(KS+2) and x(KS+2) (into the x set used at KS), multiply x(KS) with y(KS) from LDS
#define PJW_STEP(KS, XCUR)                                                                                           \
    AUM_FOR_EACH_WAVE(w, PJW_NW) {                                                                                   \
        const vi lane = lane_id();                                                                                   \
        const vi t16 = lane & 15, g = lane >> 4;                                                                     \
        const int row0 = ec * PJW_ROWS + w * 16 * PJW_EB;                                                            \
        const T* xb = X + (int64_t)(row0 < p.dim ? row0 : 0) * ntok;                                                 \
        const int rows_left = p.dim - row0;                                                                          \
        const uint16_t* ycur = lds + ((KS) & 1) * PJW_YSTAGE;                                                        \
        if ((KS) + 1 < nsteps) pjw_y_store(lds + (((KS) + 1) & 1) * PJW_YSTAGE, w, yr[AUM_W(w)]);                    \
        frag8 xc[2 * PJW_EB];                                                                                        \
        AUM_UNROLL                                                                                                   \
        for (int eb = 0; eb < 2 * PJW_EB; ++eb) xc[eb] = XCUR[AUM_W(w)][eb];                                         \
        if ((KS) + 2 < nsteps) {                                                                                     \
            pjw_y_load(Y, ntok, nr, t_begin + (int64_t)((KS) + 2) * PJW_KS, t_end, w, yr[AUM_W(w)]);                 \
            pjw_x_load(xb, ntok, rows_left, t_begin + (int64_t)((KS) + 2) * PJW_KS, t_end, XCUR[AUM_W(w)]);          \
        }                                                                                                            \
        AUM_UNROLL                                                                                                   \
        for (int rb = 0; rb < PJ_MAXRB; ++rb) {                                                                      \
            if (rb < nrb) {                                                                                          \
                const frag8 y0 = lds16_read8_a(ycur, (t16 + rb * 16) * PJW_YPITCH + g * 8);                          \
                const frag8 y1 = lds16_read8_a(ycur, (t16 + rb * 16) * PJW_YPITCH + g * 8 + 32);                     \
                AUM_UNROLL                                                                                           \
                for (int eb = 0; eb < PJW_EB; ++eb) {                                                                \
                    acc[AUM_W(w)][eb][rb] = mfma16(T{}, xc[2 * eb], y0, acc[AUM_W(w)][eb][rb]);                      \
                    acc[AUM_W(w)][eb][rb] = mfma16(T{}, xc[2 * eb + 1], y1, acc[AUM_W(w)][eb][rb]);                  \
                }                                                                                                    \
            }                                                                                                        \
        }                                                                                                            \
    }                                                                                                                \
    AUM_WG_BARRIER_LDS();
    for (int ks = 0; ks < nsteps; ks += 2) {
        PJW_STEP(ks, xa)
        if (ks + 1 < nsteps) { PJW_STEP(ks + 1, xb2) }
    }
#undef PJW_STEP
    AUM_FOR_EACH_WAVE(w, PJW_NW) {
        const vi lane = lane_id();
        const vi t16 = lane & 15, g = lane >> 4;
        float* outp = p.out + (int64_t)split * p.dim * nr;
        const int row0 = ec * PJW_ROWS + w * 16 * PJW_EB;
        AUM_UNROLL
        for (int eb = 0; eb < PJW_EB; ++eb) {
            AUM_UNROLL
            for (int rb = 0; rb < PJ_MAXRB; ++rb) {
                if (rb < nrb && row0 + eb * 16 < p.dim) {
                    const vi r = t16 + rb * 16;
                    AUM_UNROLL
                    for (int q = 0; q < 4; ++q) {
                        const vi e = g * 4 + (row0 + eb * 16 + q);        // D row = channel, D col = r
                        const vi idx = p.transpose_out ? r * p.dim + e : e * nr + r;
                        gstore(outp, vsel_i(r < nr, idx, spl_i(0)), acc_get(acc[AUM_W(w)][eb][rb], q), r < nr);
                    }
                }
            }
        }
    }
}

}  // namespace aum
