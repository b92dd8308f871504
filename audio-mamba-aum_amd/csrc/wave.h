// wave.h -- the 64-lane wavefront programming layer every kernel in this library is written against.
//
// Each kernel body is ONE source, written as straight-line "per-lane" code over the types vf / vi / vm
// and the free functions below.  It is compiled two ways:
//
//   * device build (hipcc --offload-arch=gfx950):  vf = float, vi = int, vm = bool; the functions map
//     1:1 onto CDNA4 instructions (v_exp_f32, v_rcp_f32, DPP row_shr/row_shl/wave_shr, v_readlane_b32,
//     ds_read/ds_write, global_atomic_add_f32 ...).  This is the product.
//
//   * lane-array build (-DAUM_EMU, host clang++):  vf = struct{float v[64]} etc. and every function
//     applies the gfx950 semantics lane by lane.  It exists ONLY so tests/ can check a kernel's index
//     arithmetic, tails, carries and reductions on a machine without a GPU (tests/emu).  It is never
//     loaded by the product path (aum_hip/_lib.py loads libaum_hip.so or fails).
//
// Cross-lane traffic inside a wavefront is DPP / readlane / wave-synchronous LDS.  Most kernels use one-wave
// workgroups; the production scan kernels use 8-wave workgroups written as barrier-separated phases
// (AUM_FOR_EACH_WAVE below) so the lane-array build can step the waves one after another.
#pragma once
#include <stdint.h>

#ifdef AUM_EMU
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define AUM_DEV inline
#define AUM_UNROLL _Pragma("unroll")
#else
#include <hip/hip_runtime.h>
#define AUM_DEV __device__ __forceinline__
#define AUM_UNROLL _Pragma("unroll")
#endif
#ifdef AUM_EMU
#define AUM_HOSTDEV inline
#else
#define AUM_HOSTDEV __host__ __device__ __forceinline__
#endif

namespace aum {

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------------
// Element types of the activation tensors.  fp32 internal math always (SSI:101-103).
// ------------------------------------------------------------------------------------------------
struct bf16_t { uint16_t bits; };
struct f16_t { uint16_t bits; };

AUM_DEV float bits_to_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
AUM_DEV uint32_t f32_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

AUM_DEV float elem_to_f32(float x) { return x; }
AUM_DEV float elem_to_f32(bf16_t x) { return bits_to_f32(((uint32_t)x.bits) << 16); }
AUM_DEV float elem_to_f32(f16_t x) { return (float)__builtin_bit_cast(_Float16, x.bits); }

AUM_DEV void f32_to_elem(float f, float& o) { o = f; }
AUM_DEV void f32_to_elem(float f, bf16_t& o) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u = f32_to_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) { o.bits = (uint16_t)((u >> 16) | 0x40u); return; }
    u += 0x7fffu + ((u >> 16) & 1u);
    o.bits = (uint16_t)(u >> 16);
}
AUM_DEV void f32_to_elem(float f, f16_t& o) { o.bits = __builtin_bit_cast(uint16_t, (_Float16)f); }

#ifndef AUM_EMU
// =================================================================================================
// Device backend: one lane per thread.
// =================================================================================================
using vf = float;
using vi = int;
using vm = bool;

AUM_DEV vi lane_id() { return (int)(threadIdx.x & 63u); }
// index of this wavefront inside a multi-wave workgroup (wave-uniform, lives in an SGPR)
AUM_DEV int wave_in_wg() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// Multi-wave workgroups are written as PHASES: `AUM_FOR_EACH_WAVE(w, NW) { ...phase body for wave w... }
// AUM_WG_BARRIER();`.  On the device the loop runs once for this thread's own wave and the barrier is s_barrier;
// the lane-array build steps the NW waves one after another (legal because phases only communicate through LDS
// across the barrier, or through commutative LDS atomics inside a phase).
#define AUM_FOR_EACH_WAVE(w, NW) for (int w = aum::wave_in_wg(), aum_once_ = 1; aum_once_; aum_once_ = 0)
#define AUM_WG_BARRIER() __syncthreads()
// Barrier issued INSIDE a wave phase (every wave of the workgroup reaches it the same number of times).  The lane-array
// build runs the waves of a phase one after another, where it is a no-op: sequential execution is one legal schedule,
// so arithmetic is checked there while race-freedom under concurrency is argued at the call site and checked on the GPU.
#define AUM_WG_BARRIER_IN_PHASE() __syncthreads()
// Workgroup barrier that only orders LDS traffic: wait for this wave's LDS operations, then s_barrier.  __syncthreads()
// is a release/acquire fence pair around the barrier and therefore also drains vmcnt -- every global load a wave has
// prefetched for a LATER step is waited for at each step's barrier, which serialises software-pipelined K loops on
// memory latency.  Use only where the data exchanged between the waves lives in LDS.
#define AUM_WG_BARRIER_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
// compiler scheduling fence (no instruction): keeps what was issued before it ahead of what follows
#define AUM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// issue priority of this wave among the waves of its SIMD (0..3; the arbiter picks by priority, then by age), and the slot number
// the hardware gave the wave on its SIMD (HW_ID.wave_id: distinct for the waves that share a SIMD)
#define AUM_SET_PRIO(p) __builtin_amdgcn_s_setprio(p)
AUM_DEV int wave_slot_on_simd() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 4) & 15u); }
// wait for every outstanding scalar-cache (and LDS) request of this wave.  Scalar loads return out of order, so the only count
// the hardware can wait for is zero: a wave that prefetches row s+1 while it still has to wait for row s waits for both.  Placing
// this BEFORE the next prefetch is issued makes the wait cover only requests that have had a whole step to arrive.
#define AUM_WAIT_SCALAR_LOADS() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Per-wave state that must survive from one phase to the next (registers on the device): declare `T name[AUM_PER_WAVE(NW)]...`
// and index it with AUM_W(w).  One slot on the device; the lane-array build keeps a slot per wave it steps through.
#define AUM_PER_WAVE(NW) 1
#define AUM_W(w) 0
AUM_DEV vf splat(float x) { return x; }
AUM_DEV vi spl_i(int x) { return x; }
AUM_DEV vf vfma(vf a, vf b, vf c) { return __builtin_fmaf(a, b, c); }
AUM_DEV vf vexp2(vf x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
AUM_DEV vf vlog2(vf x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
AUM_DEV vf vrcp(vf x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32
AUM_DEV vf vrsqrt(vf x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32
AUM_DEV vf vdiv(vf a, vf b) { return a / b; }                  // IEEE division (correctly rounded)
AUM_DEV vf vsqrt(vf x) { return __builtin_sqrtf(x); }
AUM_DEV vf vmax(vf a, vf b) { return __builtin_fmaxf(a, b); }
AUM_DEV vf vsel(vm m, vf a, vf b) { return m ? a : b; }
AUM_DEV vi vsel_i(vm m, vi a, vi b) { return m ? a : b; }
AUM_DEV bool any_lane(vm m) { return __any(m); }
AUM_DEV vi vcvt_i(vf x) { return (int)x; }
AUM_DEV vi vmin_i(vi a, int b) { return a < b ? a : b; }
AUM_DEV vi vmax_i(vi a, int b) { return a > b ? a : b; }

// Packed pair of fp32 lanes-values: the VALU of gfx950 executes a wave64 fp32 instruction in 4 cycles (measured:
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.2-4.5 on the scan kernels) and the packed forms v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 do TWO fp32 per lane in the same 4 cycles -- the 157 TFLOP/s vector peak is a packed-math number.
// The scan kernels therefore carry their two rows per wave as one vf2.
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef vf2 vf2_raw;
AUM_DEV vf2 mk2(vf a, vf b) { return vf2{a, b}; }
AUM_DEV vf2 spl2(vf a) { return vf2{a, a}; }
AUM_DEV vf lo2(vf2 v) { return v.x; }
AUM_DEV vf hi2(vf2 v) { return v.y; }
AUM_DEV vf2 vfma2(vf2 a, vf2 b, vf2 c) { return __builtin_elementwise_fma(a, b, c); }
AUM_DEV vf2 vexp2_2(vf2 x) { return vf2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
AUM_DEV vf2 vsel2(vm m, vf2 a, vf2 b) { return m ? a : b; }

// Element offsets are taken as UNSIGNED 32-bit values (every caller clamps them into the row): the access then is
// `global_* v, v_offset, s[base]` -- an SGPR base plus a 32-bit VGPR offset -- instead of a sign-extended 64-bit per-lane address
// (two VGPRs and a v_lshl_add_u64 per access, and they get hoisted out of loops and spilled).
// CONTRACT: on every ACTIVE lane the index is >= 0 (a negative index -- an address before the row pointer -- would wrap to +4 G elements
// here).  Callers clamp with vmax_i or mask such lanes; the lane-array build asserts it (aum_emu_idx_ok below) so the host tests catch a
// violation the device would turn into a wild access.
template <class T> AUM_DEV vf gload(const T* p, vi idx, vm m) { return m ? elem_to_f32(p[(uint32_t)idx]) : 0.f; }
template <class T> AUM_DEV void gstore(T* p, vi idx, vf v, vm m) { if (m) f32_to_elem(v, p[(uint32_t)idx]); }
// unconditional load (caller clamps idx into range): no exec-mask branch, so several can be in flight
template <class T> AUM_DEV vf gload_u(const T* p, vi idx) { return elem_to_f32(p[(uint32_t)idx]); }
AUM_DEV void gatomic_add(float* p, vi idx, vf v, vm m) { if (m) atomicAdd(p + idx, v); }
// 8 consecutive elements per lane as ONE (2-byte types) or TWO (fp32) 16-byte vector accesses.  Rows of the
// (batch, dim, len) tensors start at arbitrary element offsets (len = 513), so the address is only element-aligned:
// the packed/aligned(2|4) struct makes hipcc emit global_load/store_dwordx4 in unaligned-access mode (HSA default).
template <int BYTES> struct __attribute__((packed, aligned(BYTES))) pk16_t { uint8_t b[16]; };
template <class T> AUM_DEV void gload8(const T* p, vi idx, vm m, vf (&o)[8]) {
    if (m) {
        if constexpr (sizeof(T) == 2) {
            const pk16_t<2> raw = *reinterpret_cast<const pk16_t<2>*>(p + (uint32_t)idx);
            T e[8];
            __builtin_memcpy(e, &raw, 16);
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) o[j] = elem_to_f32(e[j]);
        } else {
            const pk16_t<4> r0 = *reinterpret_cast<const pk16_t<4>*>(p + (uint32_t)idx);
            const pk16_t<4> r1 = *reinterpret_cast<const pk16_t<4>*>(p + (uint32_t)idx + 4);
            float e[8];
            __builtin_memcpy(e, &r0, 16);
            __builtin_memcpy(e + 4, &r1, 16);
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) o[j] = e[j];
        }
    } else {
        AUM_UNROLL
        for (int j = 0; j < 8; ++j) o[j] = 0.f;
    }
}
// two fp32 -> one packed pair of 16-bit elements.  bf16: v_cvt_pk_bf16_f32 (gfx950; round-to-nearest-even, quiet NaN -- the
// same result as f32_to_elem(bf16_t), which costs ~5 VALU instructions per element)
typedef __bf16 aum_bf16x2 __attribute__((ext_vector_type(2)));
template <class T> AUM_DEV uint32_t f32x2_to_elem2(float a, float b) {
    if constexpr (sizeof(T) == 2 && __is_same(T, bf16_t)) {
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(vf2_raw{a, b}, aum_bf16x2));
    } else {
        T ea, eb;
        f32_to_elem(a, ea);
        f32_to_elem(b, eb);
        return (uint32_t)ea.bits | ((uint32_t)eb.bits << 16);
    }
}
template <class T> AUM_DEV void gstore8(T* p, vi idx, const vf (&v)[8], vm m) {
    if (m) {
        if constexpr (sizeof(T) == 2) {
            uint32_t e[4];
            AUM_UNROLL
            for (int j = 0; j < 4; ++j) e[j] = f32x2_to_elem2<T>(v[2 * j], v[2 * j + 1]);
            pk16_t<2> raw;
            __builtin_memcpy(&raw, e, 16);
            *reinterpret_cast<pk16_t<2>*>(p + (uint32_t)idx) = raw;
        } else {
            float e[8];
            AUM_UNROLL
            for (int j = 0; j < 8; ++j) e[j] = v[j];
            pk16_t<4> r0, r1;
            __builtin_memcpy(&r0, e, 16);
            __builtin_memcpy(&r1, e + 4, 16);
            *reinterpret_cast<pk16_t<4>*>(p + (uint32_t)idx) = r0;
            *reinterpret_cast<pk16_t<4>*>(p + (uint32_t)idx + 4) = r1;
        }
    }
}
// L1-bypassing accesses for scratch that this wave wrote earlier in the same launch
AUM_DEV vf gload_coherent(const float* p, vi idx, vm m) {
    return m ? __hip_atomic_load(p + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
}
AUM_DEV void gstore_coherent(float* p, vi idx, vf v, vm m) {
    if (m) __hip_atomic_store(p + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
AUM_DEV vf lds_read(const float* lds, vi idx) { return lds[idx]; }
AUM_DEV void lds_write(float* lds, vi idx, vf v) { lds[idx] = v; }
AUM_DEV void lds_write_m(float* lds, vi idx, vf v, vm m) { if (m) lds[idx] = v; }
AUM_DEV void lds_atomic_add(float* lds, vi idx, vf v) {   // ds_add_f32 (no return)
    __hip_atomic_fetch_add(lds + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
AUM_DEV void wave_sync() { __syncthreads(); }  // single-wave workgroup: orders LDS traffic, ~free
// Orders one wave's own LDS traffic inside a multi-wave workgroup (lane A's ds_write seen by lane B's later ds_read of
// the same wave).  The LDS executes a wave's instructions in order, so only the compiler has to be held: a wavefront-scope
// fence + scheduling barrier, no s_barrier.
AUM_DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int CTRL> AUM_DEV vf dpp_mov(vf x, vf old) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                                __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// lane i <- lane i-N of its 16-lane row; lanes with (i%16) < N keep `old`
template <int N> AUM_DEV vf dpp_row_shr(vf x, vf old) { return dpp_mov<0x110 + N>(x, old); }
// lane i <- lane i+N of its row; lanes with (i%16)+N > 15 keep `old`
template <int N> AUM_DEV vf dpp_row_shl(vf x, vf old) { return dpp_mov<0x100 + N>(x, old); }
// lane i <- lane (i + N) mod 16 of its own 16-lane row (rotate right by N: every lane has a source)
template <int N> AUM_DEV vf dpp_row_ror(vf x) { return dpp_mov<0x120 + N>(x, x); }
// lane i <- lane src[i] (any lane): ds_bpermute_b32, one pass through the LDS crossbar, no LDS memory
AUM_DEV vf lane_gather(vf x, vi src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, x)));
}
AUM_DEV vf dpp_wave_shr1(vf x, vf old) { return dpp_mov<0x138>(x, old); }  // lane i <- i-1, lane 0 keeps old
AUM_DEV vf dpp_wave_shl1(vf x, vf old) { return dpp_mov<0x130>(x, old); }  // lane i <- i+1, lane 63 keeps old
AUM_DEV float readlane(vf x, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
// lane `lane` (wave-uniform) of v <- the wave-uniform value s.  (clang has no v_writelane_b32 builtin and inline asm would hide the
// SGPR-lane-select hazard from the compiler: a compare + select, two VALU instructions.)
AUM_DEV vf writelane(vf v, float s, int lane) { return (int)(threadIdx.x & 63u) == lane ? s : v; }
// wave-uniform element of a row (every lane reads the same address)
template <class T> AUM_DEV float gload_s(const T* p, int idx) { return elem_to_f32(p[idx]); }
// a value the optimiser must treat as unknown at this point: address arithmetic that depends on it is not hoisted out of
// the enclosing loop (where it would occupy registers for the whole loop)
AUM_DEV vi opaque_i(vi x) { asm volatile("" : "+v"(x)); return x; }
// the value exists in a register HERE: keeps the optimiser from sinking a chain of accumulator updates to their distant use (and
// carrying the operands of all of them in between)
AUM_DEV void pin_value(vf& x) { asm volatile("" : "+v"(x)); }
AUM_DEV void pin_value2(vf2& x) { asm volatile("" : "+v"(x)); }
// (lo, lo) / (hi, hi) of a pair: operand modifiers (op_sel) of the packed instruction that consumes them
AUM_DEV vf2 bc_lo(vf2 a) { return __builtin_shufflevector(a, a, 0, 0); }
AUM_DEV vf2 bc_hi(vf2 a) { return __builtin_shufflevector(a, a, 1, 1); }
// sixteen per-lane values addressed by a wave-uniform RUN-TIME index: a register vector the compiler indexes through M0
// (s_set_gpr_idx / v_movrel), so a loop over state pairs need not be unrolled to keep its per-pair carries in registers
struct vf16 { float __attribute__((ext_vector_type(16))) r; };
AUM_DEV vf vf16_get(const vf16& a, int i) { return a.r[i]; }
AUM_DEV void vf16_set(vf16& a, int i, vf v) { a.r[i] = v; }
// a wave-uniform pointer pinned to an SGPR pair: what is added to it afterwards (a 32-bit per-lane offset) stays the `voffset` of a
// `global_* v, voffset, s[base]` access -- without the pin the optimiser folds the uniform part into a 64-bit per-lane address
// (v_lshl_add_u64 per access)
template <class P> AUM_DEV P* uniform_ptr(P* p) { asm volatile("" : "+s"(p)); return p; }
// accesses through a pinned pointer: the pin hides where the pointer came from, so the global address space is stated here
// (a generic pointer would make these flat_load / flat_store with 64-bit per-lane addresses)
template <class T> AUM_DEV vf gload_g(const T* p, vi idx) {
    if constexpr (sizeof(T) == 4) {
        return ((const __attribute__((address_space(1))) float*)p)[(uint32_t)idx];
    } else {
        T e;
        e.bits = ((const __attribute__((address_space(1))) uint16_t*)p)[(uint32_t)idx];
        return elem_to_f32(e);
    }
}
template <class T> AUM_DEV void gstore_g(T* p, vi idx, vf v) {
    if constexpr (sizeof(T) == 4) {
        ((__attribute__((address_space(1))) float*)p)[(uint32_t)idx] = v;
    } else {
        auto* q = (__attribute__((address_space(1))) uint16_t*)p;
        if constexpr (__is_same(T, bf16_t)) q[(uint32_t)idx] = (uint16_t)f32x2_to_elem2<T>(v, v);
        else { T e; f32_to_elem(v, e); q[(uint32_t)idx] = e.bits; }
    }
}
// Buffer-resource accesses (buffer_load/store ... offen): address = descriptor base + per-lane byte offset (VGPR, constant over a
// kernel's row walk) + wave-uniform byte offset (SGPR, one s_add per step).  No 64-bit address arithmetic on either ALU.
template <class T> struct gbuf { __amdgpu_buffer_rsrc_t r; };
template <class T> AUM_DEV gbuf<T> make_gbuf(const T* p) {
    gbuf<T> b;
    b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), 0, 0x7fffffff, 0x00020000);
    return b;
}
template <class T> AUM_DEV vf gbuf_load(const gbuf<T>& b, vi voff_bytes, int soff_bytes) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff_bytes, soff_bytes, 0));
    } else {
        T e;
        e.bits = __builtin_amdgcn_raw_buffer_load_b16(b.r, voff_bytes, soff_bytes, 0);
        return elem_to_f32(e);
    }
}
// the same access with the element left as it was loaded (its bits in the low end of a register): a prefetch whose widening --
// and with it the s_waitcnt -- is written where the value is used, not where the load is issued
template <class T> AUM_DEV vi gbuf_load_raw(const gbuf<T>& b, vi voff_bytes, int soff_bytes) {
    if constexpr (sizeof(T) == 4) return (int)__builtin_amdgcn_raw_buffer_load_b32(b.r, voff_bytes, soff_bytes, 0);
    else return (int)__builtin_amdgcn_raw_buffer_load_b16(b.r, voff_bytes, soff_bytes, 0);
}
template <class T> AUM_DEV vf raw_to_f32(vi raw) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(float, raw);
    } else {
        T e;
        e.bits = (uint16_t)raw;
        return elem_to_f32(e);
    }
}
// two fp32 values as one dword of two 16-bit elements of T (round to nearest even; lo in the low half): the packed state checkpoint of the
// token-major scan, read back with lds_pair_to_f32<T>
template <class T> AUM_DEV void gbuf_store_pair16(const gbuf<float>& b, vi voff_bytes, int soff_bytes, vf lo, vf hi) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    __builtin_amdgcn_raw_buffer_store_b32(f32x2_to_elem2<T>(lo, hi), b.r, voff_bytes, soff_bytes, 0);
}
template <class T> AUM_DEV void gbuf_store(const gbuf<T>& b, vi voff_bytes, int soff_bytes, vf v) {
    if constexpr (sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), b.r, voff_bytes, soff_bytes, 0);
    } else if constexpr (__is_same(T, bf16_t)) {
        __builtin_amdgcn_raw_buffer_store_b16((uint16_t)f32x2_to_elem2<T>(v, v), b.r, voff_bytes, soff_bytes, 0);
    } else {
        T e;
        f32_to_elem(v, e);
        __builtin_amdgcn_raw_buffer_store_b16(e.bits, b.r, voff_bytes, soff_bytes, 0);
    }
}
typedef float aum_f2 __attribute__((ext_vector_type(2)));
typedef float aum_f4 __attribute__((ext_vector_type(4)));
// two consecutive elements of a row, raw (one dword for the 16-bit types, a dwordx2 for float), and their widening -- separate so that
// a prefetch is not waited for where it is issued
struct vpair_raw { vi w[2]; };
template <class T> AUM_DEV vpair_raw gbuf_load_pair_raw(const gbuf<T>& b, vi voff_bytes, int soff_bytes) {
    vpair_raw r;
    if constexpr (sizeof(T) == 4) {
        typedef int i2 __attribute__((ext_vector_type(2)));
        const i2 v = __builtin_bit_cast(i2, __builtin_amdgcn_raw_buffer_load_b64(b.r, voff_bytes, soff_bytes, 0));
        r.w[0] = v.x;
        r.w[1] = v.y;
    } else {
        r.w[0] = (int)__builtin_amdgcn_raw_buffer_load_b32(b.r, voff_bytes, soff_bytes, 0);
        r.w[1] = 0;
    }
    return r;
}
template <class T> AUM_DEV void pair_raw_to_f32(const vpair_raw& r, vf& lo, vf& hi) {
    if constexpr (sizeof(T) == 4) {
        lo = __builtin_bit_cast(float, r.w[0]);
        hi = __builtin_bit_cast(float, r.w[1]);
    } else if constexpr (__is_same(T, bf16_t)) {
        lo = bits_to_f32((uint32_t)r.w[0] << 16);
        hi = bits_to_f32((uint32_t)r.w[0] & 0xffff0000u);
    } else {
        lo = (float)__builtin_bit_cast(_Float16, (uint16_t)((uint32_t)r.w[0] & 0xffffu));
        hi = (float)__builtin_bit_cast(_Float16, (uint16_t)((uint32_t)r.w[0] >> 16));
    }
}
// two consecutive elements of a row as fp32 (one dword for the 16-bit types, one dwordx2 for float)
template <class T> AUM_DEV void gbuf_load_pair(const gbuf<T>& b, vi voff_bytes, int soff_bytes, vf& lo, vf& hi) {
    if constexpr (sizeof(T) == 4) {
        const auto w = __builtin_amdgcn_raw_buffer_load_b64(b.r, voff_bytes, soff_bytes, 0);
        static_assert(sizeof(w) == 8, "dwordx2");
        const aum_f2 f = __builtin_bit_cast(aum_f2, w);
        lo = f.x;
        hi = f.y;
    } else {
        const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(b.r, voff_bytes, soff_bytes, 0);
        if constexpr (__is_same(T, bf16_t)) {
            lo = bits_to_f32(w << 16);
            hi = bits_to_f32(w & 0xffff0000u);
        } else {
            lo = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
            hi = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
        }
    }
}
// LDS: a pair per lane (ds_write_b64; idx even) and four consecutive words read by every lane from ONE wave-uniform index
// (ds_read_b128, all lanes the same address: a broadcast)
AUM_DEV void lds_write2(float* lds, vi idx, vf a, vf b) { *reinterpret_cast<aum_f2*>(lds + idx) = aum_f2{a, b}; }
AUM_DEV void lds_read2_u(const float* lds, int idx, vf (&o)[2]) {
    const aum_f2 v = *reinterpret_cast<const aum_f2*>(lds + idx);
    o[0] = v.x; o[1] = v.y;
}
AUM_DEV void lds_read4_u(const float* lds, int idx, vf (&o)[4]) {
    const aum_f4 v = *reinterpret_cast<const aum_f4*>(lds + idx);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
// 16 bytes per lane: global <-> registers (buffer_load/store_dwordx4 ... offen) and registers <-> LDS (ds_write/read_b128), and one
// element of T per lane out of / into an LDS tile (ds_read_u16 / ds_read_b32, ds_write_b16 / ds_write_b32).  Byte offsets.
struct vq { vi w[4]; };
template <class T> AUM_DEV vq gbuf_load16(const gbuf<T>& b, vi voff_bytes, int soff_bytes) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const auto r = __builtin_amdgcn_raw_buffer_load_b128(b.r, voff_bytes, soff_bytes, 0);
    static_assert(sizeof(r) == 16, "dwordx4");
    const u4 u = __builtin_bit_cast(u4, r);
    vq q;
    q.w[0] = (int)u.x; q.w[1] = (int)u.y; q.w[2] = (int)u.z; q.w[3] = (int)u.w;
    return q;
}
template <class T> AUM_DEV void gbuf_store16(const gbuf<T>& b, vi voff_bytes, int soff_bytes, const vq& q) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 u = {(uint32_t)q.w[0], (uint32_t)q.w[1], (uint32_t)q.w[2], (uint32_t)q.w[3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(b.r, 0, 0, 0)), u), b.r, voff_bytes,
                                           soff_bytes, 0);
}
template <class T> AUM_DEV void gbuf_store16_m(const gbuf<T>& b, vi voff_bytes, int soff_bytes, const vq& q, vm m) {
    if (m) gbuf_store16(b, voff_bytes, soff_bytes, q);
}
// the 16 / sizeof(T) consecutive elements of a 16-byte access as fp32, and back (round to nearest even)
template <class T> AUM_DEV void vq_unpack(const vq& q, vf (&o)[16 / sizeof(T)]) {
    if constexpr (sizeof(T) == 4) {
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) o[i] = __builtin_bit_cast(float, q.w[i]);
    } else {
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) {
            vpair_raw r;
            r.w[0] = q.w[i];
            r.w[1] = 0;
            pair_raw_to_f32<T>(r, o[2 * i], o[2 * i + 1]);
        }
    }
}
template <class T> AUM_DEV vq vq_pack(const vf (&v)[16 / sizeof(T)]) {
    vq q;
    if constexpr (sizeof(T) == 4) {
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) q.w[i] = __builtin_bit_cast(int, v[i]);
    } else {
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) q.w[i] = (int)f32x2_to_elem2<T>(v[2 * i], v[2 * i + 1]);
    }
    return q;
}
// 8 bytes per lane of a 16-bit type (four elements): the narrow form of the 16-byte accesses above, for kernels that want twice the waves at
// half the registers (conv_tm_kernels.h: the backward)
struct vh { vi w[2]; };
template <class T> AUM_DEV vh gbuf_load8(const gbuf<T>& b, vi voff_bytes, int soff_bytes) {
    typedef int i2 __attribute__((ext_vector_type(2)));
    const i2 v = __builtin_bit_cast(i2, __builtin_amdgcn_raw_buffer_load_b64(b.r, voff_bytes, soff_bytes, 0));
    vh q;
    q.w[0] = v.x; q.w[1] = v.y;
    return q;
}
template <class T> AUM_DEV void gbuf_store8(const gbuf<T>& b, vi voff_bytes, int soff_bytes, const vh& q) {
    typedef int i2 __attribute__((ext_vector_type(2)));
    const i2 u = {q.w[0], q.w[1]};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(b.r, 0, 0, 0)), u), b.r, voff_bytes, soff_bytes, 0);
}
template <class T> AUM_DEV void gbuf_store8_m(const gbuf<T>& b, vi voff_bytes, int soff_bytes, const vh& q, vm m) {
    typedef int i2 __attribute__((ext_vector_type(2)));
    const i2 u = {q.w[0], q.w[1]};
    if (m) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(b.r, 0, 0, 0)), u), b.r, voff_bytes, soff_bytes, 0);
}
template <class T> AUM_DEV void vh_unpack(const vh& q, vf (&o)[4]) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    AUM_UNROLL
    for (int i = 0; i < 2; ++i) {
        vpair_raw r;
        r.w[0] = q.w[i];
        r.w[1] = 0;
        pair_raw_to_f32<T>(r, o[2 * i], o[2 * i + 1]);
    }
}
template <class T> AUM_DEV vh vh_pack(const vf (&v)[4]) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    vh q;
    AUM_UNROLL
    for (int i = 0; i < 2; ++i) q.w[i] = (int)f32x2_to_elem2<T>(v[2 * i], v[2 * i + 1]);
    return q;
}
AUM_DEV void lds_write16(float* lds, vi byte_off, const vq& q) {
    typedef int i4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<i4*>(reinterpret_cast<char*>(lds) + byte_off) = i4{q.w[0], q.w[1], q.w[2], q.w[3]};
}
AUM_DEV vq lds_read16(const float* lds, vi byte_off) {
    typedef int i4 __attribute__((ext_vector_type(4)));
    const i4 v = *reinterpret_cast<const i4*>(reinterpret_cast<const char*>(lds) + byte_off);
    vq q;
    q.w[0] = v.x; q.w[1] = v.y; q.w[2] = v.z; q.w[3] = v.w;
    return q;
}
// the bits of a raw word as fp32 and back (the fp32 rows that travel as 16-byte pieces)
AUM_DEV vf int_as_f32(vi x) { return __builtin_bit_cast(float, x); }
AUM_DEV vi f32_as_int(vf x) { return __builtin_bit_cast(int, x); }
// Global -> LDS without registers (buffer_load_dword / _dwordx4 ... lds): lane l's 4 or 16 bytes land at lds_dst + l * 4 / l * 16, the
// data never passes through VGPRs and nothing has to be "parked" later.  Completion is counted in vmcnt like any load: AUM_WAIT_VM(n)
// (memory operations complete in issue order, n = the number of YOUNGER ones that may stay in flight) before the LDS data is read.
// (the LDS address-space cast does not parse in the host pass of the same source, which never runs these bodies)
template <class T> AUM_DEV void gbuf_load16_lds(const gbuf<T>& b, vi voff_bytes, int soff_bytes, float* lds_dst) {
#ifdef __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff_bytes, soff_bytes, 0, 0);
#endif
}
template <class T> AUM_DEV void gbuf_load4_lds(const gbuf<T>& b, vi voff_bytes, int soff_bytes, float* lds_dst) {
#ifdef __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_dst, 4, voff_bytes, soff_bytes, 0, 0);
#endif
}
#define AUM_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// the two consecutive elements of T this lane fetched with gbuf_load4_lds (one dword for the 16-bit types; two for float, the second
// WAVE dwords further)
template <class T> AUM_DEV void lds_pair_to_f32(const float* lds, vf& lo, vf& hi) {
    const int l4 = (int)(threadIdx.x & 63u) * 4;
    vpair_raw r;
    r.w[0] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(lds) + l4);
    r.w[1] = 0;
    if constexpr (sizeof(T) == 4) r.w[1] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(lds) + l4 + WAVE * 4);
    pair_raw_to_f32<T>(r, lo, hi);
}
template <class T> AUM_DEV vi lds_read_raw(const float* lds, vi byte_off) {
    if constexpr (sizeof(T) == 4) return *reinterpret_cast<const int*>(reinterpret_cast<const char*>(lds) + byte_off);
    else return (int)*reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(lds) + byte_off);
}
template <class T> AUM_DEV void lds_write_elem(float* lds, vi byte_off, vf v) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + byte_off) = v;
    } else {
        uint16_t bits;
        if constexpr (__is_same(T, bf16_t)) bits = (uint16_t)f32x2_to_elem2<T>(v, v);
        else { T e; f32_to_elem(v, e); bits = e.bits; }
        *reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(lds) + byte_off) = bits;
    }
}
#define AUM_LDS(type, name, count) __shared__ type name[count]

#else
// =================================================================================================
// Lane-array backend (tests only): gfx950 semantics applied lane by lane on the host.
// =================================================================================================
struct vf { float v[WAVE]; };
struct vi { int v[WAVE]; };
struct vm { bool v[WAVE]; };

#define AUM_LANES for (int l = 0; l < WAVE; ++l)
inline vi lane_id() { vi r; AUM_LANES r.v[l] = l; return r; }
inline vf splat(float x) { vf r; AUM_LANES r.v[l] = x; return r; }
inline vf splat(const vf& x) { return x; }      // a value that is already per-lane (wave-uniform by construction at the call site)
inline vi spl_i(int x) { vi r; AUM_LANES r.v[l] = x; return r; }

#define AUM_BINOP_F(op)                                                                         \
    inline vf operator op(const vf& a, const vf& b) { vf r; AUM_LANES r.v[l] = a.v[l] op b.v[l]; return r; } \
    inline vf operator op(const vf& a, float b) { vf r; AUM_LANES r.v[l] = a.v[l] op b; return r; }          \
    inline vf operator op(float a, const vf& b) { vf r; AUM_LANES r.v[l] = a op b.v[l]; return r; }
AUM_BINOP_F(+) AUM_BINOP_F(-) AUM_BINOP_F(*)
inline vf operator-(const vf& a) { vf r; AUM_LANES r.v[l] = -a.v[l]; return r; }
inline vf& operator+=(vf& a, const vf& b) { AUM_LANES a.v[l] += b.v[l]; return a; }
#define AUM_CMP_F(op)                                                                          \
    inline vm operator op(const vf& a, const vf& b) { vm r; AUM_LANES r.v[l] = a.v[l] op b.v[l]; return r; } \
    inline vm operator op(const vf& a, float b) { vm r; AUM_LANES r.v[l] = a.v[l] op b; return r; }
AUM_CMP_F(<) AUM_CMP_F(>) AUM_CMP_F(<=) AUM_CMP_F(>=) AUM_CMP_F(==)
#define AUM_BINOP_I(op)                                                                         \
    inline vi operator op(const vi& a, const vi& b) { vi r; AUM_LANES r.v[l] = a.v[l] op b.v[l]; return r; } \
    inline vi operator op(const vi& a, int b) { vi r; AUM_LANES r.v[l] = a.v[l] op b; return r; }            \
    inline vi operator op(int a, const vi& b) { vi r; AUM_LANES r.v[l] = a op b.v[l]; return r; }
AUM_BINOP_I(+) AUM_BINOP_I(-) AUM_BINOP_I(*) AUM_BINOP_I(/) AUM_BINOP_I(%) AUM_BINOP_I(&) AUM_BINOP_I(>>) AUM_BINOP_I(<<)
#define AUM_CMP_I(op)                                                                          \
    inline vm operator op(const vi& a, const vi& b) { vm r; AUM_LANES r.v[l] = a.v[l] op b.v[l]; return r; } \
    inline vm operator op(const vi& a, int b) { vm r; AUM_LANES r.v[l] = a.v[l] op b; return r; }
AUM_CMP_I(<) AUM_CMP_I(>) AUM_CMP_I(<=) AUM_CMP_I(>=) AUM_CMP_I(==) AUM_CMP_I(!=)
inline vm operator&&(const vm& a, const vm& b) { vm r; AUM_LANES r.v[l] = a.v[l] && b.v[l]; return r; }
inline vm operator&&(const vm& a, bool b) { vm r; AUM_LANES r.v[l] = a.v[l] && b; return r; }
inline vm operator||(const vm& a, const vm& b) { vm r; AUM_LANES r.v[l] = a.v[l] || b.v[l]; return r; }
inline vm operator!(const vm& a) { vm r; AUM_LANES r.v[l] = !a.v[l]; return r; }

inline vf vfma(const vf& a, const vf& b, const vf& c) { vf r; AUM_LANES r.v[l] = std::fmaf(a.v[l], b.v[l], c.v[l]); return r; }
inline vf vfma(const vf& a, float b, const vf& c) { return vfma(a, splat(b), c); }
inline vf vfma(float a, const vf& b, const vf& c) { return vfma(splat(a), b, c); }
inline vf vfma(const vf& a, const vf& b, float c) { return vfma(a, b, splat(c)); }
inline vf vexp2(const vf& x) { vf r; AUM_LANES r.v[l] = std::exp2(x.v[l]); return r; }
inline vf vlog2(const vf& x) { vf r; AUM_LANES r.v[l] = std::log2(x.v[l]); return r; }
inline vf vrcp(const vf& x) { vf r; AUM_LANES r.v[l] = 1.0f / x.v[l]; return r; }
inline vf vrsqrt(const vf& x) { vf r; AUM_LANES r.v[l] = 1.0f / std::sqrt(x.v[l]); return r; }
inline vf vsqrt(const vf& x) { vf r; AUM_LANES r.v[l] = std::sqrt(x.v[l]); return r; }
inline vf vdiv(const vf& a, const vf& b) { vf r; AUM_LANES r.v[l] = a.v[l] / b.v[l]; return r; }
inline vf vmax(const vf& a, const vf& b) { vf r; AUM_LANES r.v[l] = std::fmax(a.v[l], b.v[l]); return r; }
inline vf vsel(const vm& m, const vf& a, const vf& b) { vf r; AUM_LANES r.v[l] = m.v[l] ? a.v[l] : b.v[l]; return r; }
inline vf vsel(const vm& m, const vf& a, float b) { return vsel(m, a, splat(b)); }
inline vf vsel(const vm& m, float a, const vf& b) { return vsel(m, splat(a), b); }
inline vi vsel_i(const vm& m, const vi& a, const vi& b) { vi r; AUM_LANES r.v[l] = m.v[l] ? a.v[l] : b.v[l]; return r; }
inline bool any_lane(const vm& m) { bool r = false; AUM_LANES r = r || m.v[l]; return r; }
inline vi vcvt_i(const vf& x) { vi r; AUM_LANES r.v[l] = (int)x.v[l]; return r; }
inline vi vmin_i(const vi& a, int b) { vi r; AUM_LANES r.v[l] = a.v[l] < b ? a.v[l] : b; return r; }
inline vi vmax_i(const vi& a, int b) { vi r; AUM_LANES r.v[l] = a.v[l] > b ? a.v[l] : b; return r; }
// scalar (wave-uniform) overloads so kernel code can mix uniform floats freely
inline float vfma(float a, float b, float c) { return std::fmaf(a, b, c); }
inline float vexp2(float x) { return std::exp2(x); }
inline float vlog2(float x) { return std::log2(x); }
inline float vrcp(float x) { return 1.0f / x; }
inline float vsel(bool m, float a, float b) { return m ? a : b; }

struct vf2 { vf x, y; };
inline vf2 mk2(const vf& a, const vf& b) { return vf2{a, b}; }
inline vf2 spl2(const vf& a) { return vf2{a, a}; }
inline vf2 spl2(float a) { return vf2{splat(a), splat(a)}; }
inline vf lo2(const vf2& v) { return v.x; }
inline vf hi2(const vf2& v) { return v.y; }
inline vf2 operator*(const vf2& a, const vf2& b) { return vf2{a.x * b.x, a.y * b.y}; }
inline vf2 operator+(const vf2& a, const vf2& b) { return vf2{a.x + b.x, a.y + b.y}; }
inline vf2 operator-(const vf2& a, const vf2& b) { return vf2{a.x - b.x, a.y - b.y}; }
inline vf2 vfma2(const vf2& a, const vf2& b, const vf2& c) { return vf2{vfma(a.x, b.x, c.x), vfma(a.y, b.y, c.y)}; }
inline vf2 vexp2_2(const vf2& v) { return vf2{vexp2(v.x), vexp2(v.y)}; }
inline vf2 vsel2(const vm& m, const vf2& a, const vf2& b) { return vf2{vsel(m, a.x, b.x), vsel(m, a.y, b.y)}; }

// the device helpers index with (uint32_t)idx: an active lane must not carry a negative index (see the contract at the device versions)
inline void aum_emu_idx_ok(int idx, const char* what) {
    if (idx < 0) { std::fprintf(stderr, "aum emu: negative element index %d on an active lane in %s\n", idx, what); std::abort(); }
}
template <class T> inline vf gload(const T* p, const vi& idx, const vm& m) {
    vf r; AUM_LANES { if (m.v[l]) aum_emu_idx_ok(idx.v[l], "gload"); r.v[l] = m.v[l] ? elem_to_f32(p[idx.v[l]]) : 0.f; } return r;
}
template <class T> inline vf gload_u(const T* p, const vi& idx) { vf r; AUM_LANES { aum_emu_idx_ok(idx.v[l], "gload_u"); r.v[l] = elem_to_f32(p[idx.v[l]]); } return r; }
template <class T> inline void gstore(T* p, const vi& idx, const vf& v, const vm& m) {
    AUM_LANES if (m.v[l]) { aum_emu_idx_ok(idx.v[l], "gstore"); f32_to_elem(v.v[l], p[idx.v[l]]); }
}
inline void gatomic_add(float* p, const vi& idx, const vf& v, const vm& m) { AUM_LANES if (m.v[l]) p[idx.v[l]] += v.v[l]; }
template <class T> inline void gload8(const T* p, const vi& idx, const vm& m, vf (&o)[8]) {
    AUM_LANES if (m.v[l]) aum_emu_idx_ok(idx.v[l], "gload8");
    for (int j = 0; j < 8; ++j) AUM_LANES o[j].v[l] = m.v[l] ? elem_to_f32(p[idx.v[l] + j]) : 0.f;
}
template <class T> inline void gstore8(T* p, const vi& idx, const vf (&v)[8], const vm& m) {
    AUM_LANES if (m.v[l]) aum_emu_idx_ok(idx.v[l], "gstore8");
    for (int j = 0; j < 8; ++j) AUM_LANES if (m.v[l]) f32_to_elem(v[j].v[l], p[idx.v[l] + j]);
}
inline vf gload_coherent(const float* p, const vi& idx, const vm& m) { return gload(p, idx, m); }
inline void gstore_coherent(float* p, const vi& idx, const vf& v, const vm& m) { gstore(p, idx, v, m); }
inline vf lds_read(const float* lds, const vi& idx) { vf r; AUM_LANES r.v[l] = lds[idx.v[l]]; return r; }
inline void lds_write(float* lds, const vi& idx, const vf& v) { AUM_LANES lds[idx.v[l]] = v.v[l]; }
inline void lds_write_m(float* lds, const vi& idx, const vf& v, const vm& m) { AUM_LANES if (m.v[l]) lds[idx.v[l]] = v.v[l]; }
inline void lds_atomic_add(float* lds, const vi& idx, const vf& v) { AUM_LANES lds[idx.v[l]] += v.v[l]; }
#define AUM_FOR_EACH_WAVE(w, NW) for (int w = 0; w < (NW); ++w)
#define AUM_WG_BARRIER() do { } while (0)
#define AUM_WG_BARRIER_IN_PHASE() do { } while (0)
#define AUM_WG_BARRIER_LDS() do { } while (0)
#define AUM_SCHED_FENCE() do { } while (0)
#define AUM_SET_PRIO(p) do { } while (0)
inline int wave_slot_on_simd() { return 0; }
#define AUM_WAIT_SCALAR_LOADS() do { } while (0)
#define AUM_PER_WAVE(NW) (NW)
#define AUM_W(w) (w)
inline void wave_sync() {}
inline void wave_lds_fence() {}

template <int N> inline vf dpp_row_shr(const vf& x, const vf& old) {
    vf r; AUM_LANES r.v[l] = ((l & 15) >= N) ? x.v[l - N] : old.v[l]; return r;
}
template <int N> inline vf dpp_row_shl(const vf& x, const vf& old) {
    vf r; AUM_LANES r.v[l] = ((l & 15) + N <= 15) ? x.v[l + N] : old.v[l]; return r;
}
template <int N> inline vf dpp_row_ror(const vf& x) { vf r; AUM_LANES r.v[l] = x.v[(l & ~15) | ((l + N) & 15)]; return r; }
inline vf lane_gather(const vf& x, const vi& src) { vf r; AUM_LANES r.v[l] = x.v[src.v[l] & (WAVE - 1)]; return r; }
inline vf dpp_wave_shr1(const vf& x, const vf& old) { vf r; AUM_LANES r.v[l] = l >= 1 ? x.v[l - 1] : old.v[l]; return r; }
inline vf dpp_wave_shl1(const vf& x, const vf& old) { vf r; AUM_LANES r.v[l] = l < WAVE - 1 ? x.v[l + 1] : old.v[l]; return r; }
template <int N> inline vf dpp_row_shr(const vf& x, float old) { return dpp_row_shr<N>(x, splat(old)); }
template <int N> inline vf dpp_row_shl(const vf& x, float old) { return dpp_row_shl<N>(x, splat(old)); }
inline vf dpp_wave_shr1(const vf& x, float old) { return dpp_wave_shr1(x, splat(old)); }
inline vf dpp_wave_shl1(const vf& x, float old) { return dpp_wave_shl1(x, splat(old)); }
inline float readlane(const vf& x, int lane) { return x.v[lane]; }
inline vf writelane(const vf& v, float s, int lane) { vf r = v; r.v[lane] = s; return r; }
template <class T> inline float gload_s(const T* p, int idx) { return elem_to_f32(p[idx]); }
inline vi opaque_i(const vi& x) { return x; }
inline void pin_value(vf&) {}
inline void pin_value2(vf2&) {}
inline vf2 bc_lo(const vf2& a) { return vf2{a.x, a.x}; }
inline vf2 bc_hi(const vf2& a) { return vf2{a.y, a.y}; }
struct vf16 { vf v[16]; };
inline vf vf16_get(const vf16& a, int i) { return a.v[i]; }
inline void vf16_set(vf16& a, int i, const vf& v) { a.v[i] = v; }
template <class P> inline P* uniform_ptr(P* p) { return p; }
template <class T> inline vf gload_g(const T* p, const vi& idx) { return gload_u(p, idx); }
template <class T> inline void gstore_g(T* p, const vi& idx, const vf& v) { AUM_LANES f32_to_elem(v.v[l], p[idx.v[l]]); }
template <class T> struct gbuf { T* p; };
template <class T> inline gbuf<T> make_gbuf(const T* p) { return gbuf<T>{const_cast<T*>(p)}; }
template <class T> inline vf gbuf_load(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes) {
    vf r; AUM_LANES r.v[l] = elem_to_f32(*(const T*)((const char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes)); return r;
}
template <class T> inline vi gbuf_load_raw(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes) {
    vi r;
    AUM_LANES {
        const T* q = (const T*)((const char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes);
        if constexpr (sizeof(T) == 4) std::memcpy(&r.v[l], q, 4);
        else r.v[l] = (int)q->bits;
    }
    return r;
}
template <class T> inline vf raw_to_f32(const vi& raw) {
    vf r;
    AUM_LANES {
        if constexpr (sizeof(T) == 4) std::memcpy(&r.v[l], &raw.v[l], 4);
        else { T e; e.bits = (uint16_t)raw.v[l]; r.v[l] = elem_to_f32(e); }
    }
    return r;
}
template <class T> inline void gbuf_store(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes, const vf& v) {
    AUM_LANES f32_to_elem(v.v[l], *(T*)((char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes));
}
template <class T> inline void gbuf_store_pair16(const gbuf<float>& b, const vi& voff_bytes, int soff_bytes, const vf& lo, const vf& hi) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    AUM_LANES {
        T e[2];
        f32_to_elem(lo.v[l], e[0]);
        f32_to_elem(hi.v[l], e[1]);
        std::memcpy((char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, e, 4);
    }
}
struct vpair_raw { vf lo, hi; };
template <class T> inline vpair_raw gbuf_load_pair_raw(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes) {
    vpair_raw r;
    AUM_LANES {
        const T* q = (const T*)((const char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes);
        r.lo.v[l] = elem_to_f32(q[0]);
        r.hi.v[l] = elem_to_f32(q[1]);
    }
    return r;
}
template <class T> inline void pair_raw_to_f32(const vpair_raw& r, vf& lo, vf& hi) { lo = r.lo; hi = r.hi; }
template <class T> inline void gbuf_load_pair(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes, vf& lo, vf& hi) {
    AUM_LANES {
        const T* q = (const T*)((const char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes);
        lo.v[l] = elem_to_f32(q[0]);
        hi.v[l] = elem_to_f32(q[1]);
    }
}
inline void lds_write2(float* lds, const vi& idx, const vf& a, const vf& b) { AUM_LANES { lds[idx.v[l]] = a.v[l]; lds[idx.v[l] + 1] = b.v[l]; } }
inline void lds_read4_u(const float* lds, int idx, vf (&o)[4]) { for (int k = 0; k < 4; ++k) o[k] = splat(lds[idx + k]); }
inline void lds_read2_u(const float* lds, int idx, vf (&o)[2]) { for (int k = 0; k < 2; ++k) o[k] = splat(lds[idx + k]); }
struct vq { vi w[4]; };
template <class T> inline vq gbuf_load16(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes) {
    vq q;
    AUM_LANES {
        int t[4];
        std::memcpy(t, (const char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, 16);
        for (int k = 0; k < 4; ++k) q.w[k].v[l] = t[k];
    }
    return q;
}
template <class T> inline void gbuf_store16(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes, const vq& q) {
    AUM_LANES {
        int t[4];
        for (int k = 0; k < 4; ++k) t[k] = q.w[k].v[l];
        std::memcpy((char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, t, 16);
    }
}
template <class T> inline void gbuf_store16_m(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes, const vq& q, const vm& m) {
    AUM_LANES if (m.v[l]) {
        int t[4];
        for (int k = 0; k < 4; ++k) t[k] = q.w[k].v[l];
        std::memcpy((char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, t, 16);
    }
}
template <class T> inline void vq_unpack(const vq& q, vf (&o)[16 / sizeof(T)]) {
    constexpr int V = 16 / sizeof(T);
    AUM_LANES {
        int w[4];
        for (int k = 0; k < 4; ++k) w[k] = q.w[k].v[l];
        T e[V];
        std::memcpy(e, w, 16);
        for (int k = 0; k < V; ++k) o[k].v[l] = elem_to_f32(e[k]);
    }
}
template <class T> inline vq vq_pack(const vf (&v)[16 / sizeof(T)]) {
    constexpr int V = 16 / sizeof(T);
    vq q;
    AUM_LANES {
        T e[V];
        for (int k = 0; k < V; ++k) f32_to_elem(v[k].v[l], e[k]);
        int w[4];
        std::memcpy(w, e, 16);
        for (int k = 0; k < 4; ++k) q.w[k].v[l] = w[k];
    }
    return q;
}
struct vh { vi w[2]; };
template <class T> inline vh gbuf_load8(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes) {
    vh q;
    AUM_LANES {
        int t[2];
        std::memcpy(t, (const char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, 8);
        for (int k = 0; k < 2; ++k) q.w[k].v[l] = t[k];
    }
    return q;
}
template <class T> inline void gbuf_store8_m(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes, const vh& q, const vm& m) {
    AUM_LANES if (m.v[l]) {
        int t[2];
        for (int k = 0; k < 2; ++k) t[k] = q.w[k].v[l];
        std::memcpy((char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, t, 8);
    }
}
template <class T> inline void gbuf_store8(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes, const vh& q) {
    AUM_LANES {
        int t[2];
        for (int k = 0; k < 2; ++k) t[k] = q.w[k].v[l];
        std::memcpy((char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, t, 8);
    }
}
template <class T> inline void vh_unpack(const vh& q, vf (&o)[4]) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    AUM_LANES {
        int w[2];
        for (int k = 0; k < 2; ++k) w[k] = q.w[k].v[l];
        T e[4];
        std::memcpy(e, w, 8);
        for (int k = 0; k < 4; ++k) o[k].v[l] = elem_to_f32(e[k]);
    }
}
template <class T> inline vh vh_pack(const vf (&v)[4]) {
    static_assert(sizeof(T) == 2, "16-bit element types");
    vh q;
    AUM_LANES {
        T e[4];
        for (int k = 0; k < 4; ++k) f32_to_elem(v[k].v[l], e[k]);
        int w[2];
        std::memcpy(w, e, 8);
        for (int k = 0; k < 2; ++k) q.w[k].v[l] = w[k];
    }
    return q;
}
inline void lds_write16(float* lds, const vi& byte_off, const vq& q) {
    AUM_LANES {
        int t[4];
        for (int k = 0; k < 4; ++k) t[k] = q.w[k].v[l];
        std::memcpy((char*)lds + byte_off.v[l], t, 16);
    }
}
inline vq lds_read16(const float* lds, const vi& byte_off) {
    vq q;
    AUM_LANES {
        int t[4];
        std::memcpy(t, (const char*)lds + byte_off.v[l], 16);
        for (int k = 0; k < 4; ++k) q.w[k].v[l] = t[k];
    }
    return q;
}
template <class T> inline void gbuf_load16_lds(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes, float* lds_dst) {
    AUM_LANES std::memcpy((char*)lds_dst + l * 16, (const char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, 16);
}
template <class T> inline void gbuf_load4_lds(const gbuf<T>& b, const vi& voff_bytes, int soff_bytes, float* lds_dst) {
    AUM_LANES std::memcpy((char*)lds_dst + l * 4, (const char*)b.p + (int64_t)voff_bytes.v[l] + soff_bytes, 4);
}
#define AUM_WAIT_VM(n) do { } while (0)
template <class T> inline void lds_pair_to_f32(const float* lds, vf& lo, vf& hi) {
    AUM_LANES {
        if constexpr (sizeof(T) == 4) {
            lo.v[l] = lds[l];
            hi.v[l] = lds[WAVE + l];
        } else {
            T e[2];
            std::memcpy(e, &lds[l], 4);
            lo.v[l] = elem_to_f32(e[0]);
            hi.v[l] = elem_to_f32(e[1]);
        }
    }
}
inline vf int_as_f32(const vi& x) { vf r; AUM_LANES std::memcpy(&r.v[l], &x.v[l], 4); return r; }
inline vi f32_as_int(const vf& x) { vi r; AUM_LANES std::memcpy(&r.v[l], &x.v[l], 4); return r; }
template <class T> inline vi lds_read_raw(const float* lds, const vi& byte_off) {
    vi r;
    AUM_LANES {
        const char* q = (const char*)lds + byte_off.v[l];
        if constexpr (sizeof(T) == 4) std::memcpy(&r.v[l], q, 4);
        else { uint16_t h; std::memcpy(&h, q, 2); r.v[l] = (int)h; }
    }
    return r;
}
template <class T> inline void lds_write_elem(float* lds, const vi& byte_off, const vf& v) {
    AUM_LANES {
        T e;
        f32_to_elem(v.v[l], e);
        std::memcpy((char*)lds + byte_off.v[l], &e, sizeof(T));
    }
}
#define AUM_LDS(type, name, count) type name[count]
#endif  // AUM_EMU

// ------------------------------------------------------------------------------------------------
// Wave-uniform loads through the scalar cache (s_load_dword*): the row is the same for all 64 lanes, so its values live in
// SGPRs and enter VALU instructions as scalar operands -- no vector registers, no LDS.  On the device the pointer is re-typed to
// the constant address space, which is what lets the compiler select SMEM for memory the kernel does not write (a plain global
// load of a uniform address stays a vector load: the kernel's own stores might alias it).  Rows must be 4-byte aligned.
// ------------------------------------------------------------------------------------------------
#ifndef AUM_EMU
typedef const __attribute__((address_space(4))) uint32_t* aum_cptr32;
AUM_DEV uint32_t sload_u32(const void* row, int dword) { return ((aum_cptr32)(uintptr_t)row)[dword]; }
#else
inline uint32_t sload_u32(const void* row, int dword) {
    uint32_t v;
    std::memcpy(&v, (const char*)row + 4 * (size_t)dword, 4);
    return v;
}
#endif
// dword `dword` of the wave-uniform row at byte offset `row_bytes` (a multiple of 4) from `base`: s_load_dword* sdst, s[base], soffset
#ifndef AUM_EMU
AUM_DEV uint32_t sload_u32_at(const void* base, int row_bytes, int dword) {
    return *(aum_cptr32)((uintptr_t)base + (uint32_t)row_bytes + 4u * (uint32_t)dword);
}
#else
inline uint32_t sload_u32_at(const void* base, int row_bytes, int dword) {
    uint32_t v;
    std::memcpy(&v, (const char*)base + row_bytes + 4 * (size_t)dword, 4);
    return v;
}
#endif
// elements 2j and 2j+1 of a wave-uniform row of T
AUM_DEV void sload_pair(const float* row, int j, float& a, float& b) {
    a = bits_to_f32(sload_u32(row, 2 * j));
    b = bits_to_f32(sload_u32(row, 2 * j + 1));
}
AUM_DEV void sload_pair(const bf16_t* row, int j, float& a, float& b) {
    const uint32_t w = sload_u32(row, j);
    a = bits_to_f32(w << 16);
    b = bits_to_f32(w & 0xffff0000u);
}
AUM_DEV void sload_pair(const f16_t* row, int j, float& a, float& b) {
    const uint32_t w = sload_u32(row, j);
    a = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
    b = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
}

// one element per lane, converted with the packed hardware conversion where there is one (bf16: v_cvt_pk_bf16_f32, same
// rounding as f32_to_elem)
template <class T> AUM_DEV void gstore1(T* p, vi idx, vf v, vm m) {
#ifndef AUM_EMU
    if constexpr (sizeof(T) == 2 && __is_same(T, bf16_t)) {
        if (m) p[(uint32_t)idx].bits = (uint16_t)f32x2_to_elem2<T>(v, v);
        return;
    }
#endif
    gstore(p, idx, v, m);
}

// ------------------------------------------------------------------------------------------------
// Backend-independent helpers built from the primitives above.
// ------------------------------------------------------------------------------------------------
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

AUM_DEV vf vexp(vf x) { return vexp2(x * LOG2E); }
AUM_DEV vf vsigmoid(vf x) { return vrcp(splat(1.0f) + vexp2(x * (-LOG2E))); }

// a / b from v_rcp_f32 plus one residual correction (4 instructions, <= 1 ulp here) instead of the ~11-instruction IEEE
// expansion: the softplus below runs once per element in the prologue of every scan kernel
AUM_DEV vf vdiv_nr(vf a, vf b) {
    const vf r = vrcp(b);
    const vf q = a * r;
    return vfma(vfma(-q, b, a), r, q);
}

// torch softplus(beta=1, threshold=20) (SSI:106-107): x > 20 ? x : log1p(exp(x)), with an accurate
// log1p for small exp(x) (log(1+e) * e / ((1+e) - 1), exact e when 1+e rounds to 1).
AUM_DEV vf vsoftplus(vf x) {
    vf e = vexp(x);
    vf w = e + 1.0f;
    vf d = w - 1.0f;
    vf lg = vlog2(w) * LN2;
    vf r = vsel(d == 0.0f, e, lg * vdiv_nr(e, vsel(d == 0.0f, splat(1.0f), d)));
    return vsel(x > 20.0f, x, r);
}

#ifdef AUM_EMU
// wave-uniform (plain float) forms for the lane-array build; on the device vf IS float and the functions above serve both
inline float vexp(float x) { return vexp2(x * LOG2E); }
inline float vsigmoid(float x) { return vrcp(1.0f + vexp2(x * (-LOG2E))); }
inline float vdiv_nr(float a, float b) {
    const float r = vrcp(b);
    const float q = a * r;
    return vfma(vfma(-q, b, a), r, q);
}
inline float vsoftplus(float x) {
    const float e = vexp(x);
    const float w = e + 1.0f;
    const float d = w - 1.0f;
    const float lg = vlog2(w) * LN2;
    const float r = d == 0.0f ? e : lg * vdiv_nr(e, d == 0.0f ? 1.0f : d);
    return x > 20.0f ? x : r;
}
#endif

// Packed-pair forms of the two element-wise functions of the scan prologues/epilogues: the adds, multiplies and the
// residual step run as v_pk_* on both halves, only v_exp_f32 / v_log_f32 / v_rcp_f32 and the selects stay per half.
AUM_DEV vf2 vsigmoid2(vf2 x) {
    const vf2 e = vexp2_2(x * spl2(splat(-LOG2E)));
    const vf2 w = e + spl2(splat(1.0f));
    return mk2(vrcp(lo2(w)), vrcp(hi2(w)));
}
AUM_DEV vf2 vsoftplus2(vf2 x) {
    const vf2 one = spl2(splat(1.0f));
    const vf2 e = vexp2_2(x * spl2(splat(LOG2E)));
    const vf2 w = e + one;
    const vf2 d = w - one;
    const vf2 lg = mk2(vlog2(lo2(w)), vlog2(hi2(w))) * spl2(splat(LN2));
    const vm z0 = lo2(d) == 0.0f, z1 = hi2(d) == 0.0f;
    const vf2 ds = mk2(vsel(z0, splat(1.0f), lo2(d)), vsel(z1, splat(1.0f), hi2(d)));
    const vf2 r = mk2(vrcp(lo2(ds)), vrcp(hi2(ds)));
    const vf2 q = e * r;
    const vf2 qq = vfma2(vfma2(spl2(splat(0.f)) - q, ds, e), r, q);
    const vf2 v = lg * qq;
    return mk2(vsel(lo2(x) > 20.0f, lo2(x), vsel(z0, lo2(e), lo2(v))), vsel(hi2(x) > 20.0f, hi2(x), vsel(z1, hi2(e), hi2(v))));
}

// Sum over each 16-lane row, result in every lane of the row: 4 DPP rotate-and-add steps.
AUM_DEV vf row_sum16(vf x) {
    x = x + dpp_row_ror<8>(x);
    x = x + dpp_row_ror<4>(x);
    x = x + dpp_row_ror<2>(x);
    x = x + dpp_row_ror<1>(x);
    return x;
}
// lane 16q + n holds the partial of 16-lane row q for item n (n = 0..15): sum the four q's, every lane gets its item's total
AUM_DEV vf sum_rows4(vf x) {
    const vi lane = lane_id();
    x = x + lane_gather(x, (lane + 16) & (WAVE - 1));
    x = x + lane_gather(x, (lane + 32) & (WAVE - 1));
    return x;
}

// Sum over the 64 lanes, result wave-uniform.  4 DPP row steps + 3 cross-row v_readlane.
AUM_DEV float wave_sum(vf x) {
    x = x + dpp_row_shr<1>(x, splat(0.f));
    x = x + dpp_row_shr<2>(x, splat(0.f));
    x = x + dpp_row_shr<4>(x, splat(0.f));
    x = x + dpp_row_shr<8>(x, splat(0.f));
    return (readlane(x, 15) + readlane(x, 31)) + (readlane(x, 47) + readlane(x, 63));
}

// ------------------------------------------------------------------------------------------------
// Sum of 32 per-lane values over the 64 lanes with the results spread over the lanes (a transposing butterfly): on return lane l
// holds the total of value k(l) = 2 * (l & 15) + ((l >> 4) & 1); lanes l and l ^ 32 hold the same total.
// Each level pairs a lane with the partner that differs in one bit of the lane number; a lane keeps the half of the values its own
// bit selects and adds the partner's partial sums of that half (and hands the partner the other half), so 32 values cost
// 16 + 8 + 4 + 2 + 1 exchanges instead of 32 x 6.  Bits 3, 2, 1, 0 stay inside a 16-lane row (DPP), bit 4 goes through
// v_permlane16_swap, and the last level adds the two halves of the wave (v_permlane32_swap).
// ------------------------------------------------------------------------------------------------
constexpr int wave_sum32_value_of_lane(int l) { return 2 * (l & 15) + ((l >> 4) & 1); }
// The same for 16 values, in place and without temporaries: lane l holds the total of value
// 8 * bit3(l) + 4 * bit2(l) + 2 * bit4(l) + bit5(l) (the four lanes of a quad hold the same total).  The two widest levels pair lanes
// inside a 16-lane row and are ONE masked v_add_f32_dpp per lane half and result (bank_mask write-enables the lanes whose bit
// selects the value; the others keep what they have): 2 instructions per result instead of two selects, a move and an add.  The
// next two levels cross rows with v_permlane16_swap / v_permlane32_swap, which exchange exactly the halves a transposing level
// hands over; bits 1 and 0 are plain sums.  32 instructions for 16 values.
constexpr int wave_sum16_value_of_lane(int l) { return 8 * ((l >> 3) & 1) + 4 * ((l >> 2) & 1) + 2 * ((l >> 4) & 1) + ((l >> 5) & 1); }
#ifdef AUM_EMU
inline vf wave_sum16(vf (&v)[16]) {
    vf r;
    AUM_LANES {
        float acc = 0.f;
        for (int m = 0; m < WAVE; ++m) acc += v[wave_sum16_value_of_lane(l)].v[m];
        r.v[l] = acc;
    }
    return r;
}
// the butterfly in two parts (device build: the 24 masked DPP adds / the swaps and quad sums of TWO butterflies interleaved); here the head
// already forms the totals
inline void wave_sum16_head(vf (&v)[16], vf (&h)[4]) {
    h[0] = wave_sum16(v);
    h[1] = h[2] = h[3] = splat(0.f);
}
inline void wave_sum16_tail2(vf (&a)[4], vf (&b)[4], vf& ra, vf& rb) {
    ra = a[0];
    rb = b[0];
}
inline vf wave_sum32(vf (&v)[32]) {
    vf r;
    AUM_LANES {
        const int k = wave_sum32_value_of_lane(l);
        float acc = 0.f;
        for (int m = 0; m < WAVE; ++m) acc += v[k].v[m];
        r.v[l] = acc;
    }
    return r;
}
#else
template <int CTRL> AUM_DEV vf dpp_fetch(vf x) { return dpp_mov<CTRL>(x, x); }
// levels over lane bits 2, 1, 0 of the butterfly: 8 values in v[0..7] -> v[0] (lane l: the value (l & 7) of the eight)
AUM_DEV void wave_sum_low3(vf (&v)[32], int lane) {
    {   // bit 2 (xor 4: lanes with the bit clear take lane + 4, the others lane - 4), 4 results
        const bool up = (lane & 4) != 0;
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) {
            const vf keep = up ? v[4 + i] : v[i], send = up ? v[i] : v[4 + i];
            const vf fromhi = dpp_mov<0x104>(send, send);      // row_shl:4  lane i <- i + 4
            const vf fromlo = dpp_mov<0x114>(send, send);      // row_shr:4  lane i <- i - 4
            v[i] = keep + (up ? fromlo : fromhi);
        }
    }
    {   // bit 1 (xor 2: quad_perm [2,3,0,1]), 2 results
        const bool up = (lane & 2) != 0;
        AUM_UNROLL
        for (int i = 0; i < 2; ++i) {
            const vf keep = up ? v[2 + i] : v[i], send = up ? v[i] : v[2 + i];
            v[i] = keep + dpp_fetch<0x4E>(send);
        }
    }
    {   // bit 0 (xor 1: quad_perm [1,0,3,2])
        const bool up = (lane & 1) != 0;
        const vf keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
        v[0] = keep + dpp_fetch<0xB1>(send);
    }
}
AUM_DEV vf wave_sum16(vf (&v)[16]) {
    // levels over lane bits 3 and 2.  Inline assembly is not covered by the compiler's hazard recogniser: a DPP or permlane read
    // needs two wait states after a VALU write of the same register, hence the s_nop at both ends and around the swaps; in the DPP
    // levels no instruction reads a register written by one of the two before it.  (The swaps are written here as well: the
    // builtin with two DIFFERENT operands came back from the compiler as a sum of the first result with itself.)
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        // bit 4: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second, so the sum of the
        // two results is (value i summed over the row pair) in even rows and (value i + 2) in odd rows; bit 5 likewise with the
        // halves of the wave (v_permlane32_swap); bits 1 and 0 are plain sums inside a quad
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %2\n\t"
        "v_permlane16_swap_b32 %1, %3\n\t"
        "s_nop 1\n\t"
        "v_add_f32 %0, %0, %2\n\t"
        "v_add_f32 %1, %1, %3\n\t"
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %1\n\t"
        "s_nop 1\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
        : "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    return v[0];
}
// wave_sum16 in two parts, so that the serial tails of TWO butterflies (the pass's dC and dB sums) interleave: the levels over lane bits
// 3 and 2 (24 independent masked DPP adds: 16 values -> 4 registers) ...
AUM_DEV void wave_sum16_head(vf (&v)[16], vf (&h)[4]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %3, %3 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
        : "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    h[0] = v[0];
    h[1] = v[1];
    h[2] = v[2];
    h[3] = v[3];
}
// ... and the levels over bits 4 and 5 (v_permlane16_swap / v_permlane32_swap + sums) and 1, 0 (quad sums) of two butterflies at once: each
// butterfly's tail alone is a chain of six dependent steps with two idle issue slots in front of every swap / DPP read; interleaved, the
// other butterfly's instruction fills one of them
AUM_DEV void wave_sum16_tail2(vf (&a)[4], vf (&b)[4], vf& ra, vf& rb) {
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %2\n\t"
        "v_permlane16_swap_b32 %4, %6\n\t"
        "v_permlane16_swap_b32 %1, %3\n\t"
        "v_permlane16_swap_b32 %5, %7\n\t"
        "v_add_f32 %0, %0, %2\n\t"
        "v_add_f32 %4, %4, %6\n\t"
        "v_add_f32 %1, %1, %3\n\t"
        "v_add_f32 %5, %5, %7\n\t"
        "s_nop 0\n\t"
        "v_permlane32_swap_b32 %0, %1\n\t"
        "v_permlane32_swap_b32 %4, %5\n\t"
        "s_nop 0\n\t"
        "v_add_f32 %0, %0, %1\n\t"
        "v_add_f32 %4, %4, %5\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
    ra = a[0];
    rb = b[0];
}
AUM_DEV vf wave_sum32(vf (&v)[32]) {
    const int lane = (int)(threadIdx.x & 63u);
    // bit 3 (xor 8: row_ror:8), 16 results
    {
        const bool up = (lane & 8) != 0;
        AUM_UNROLL
        for (int i = 0; i < 16; ++i) {
            const vf keep = up ? v[16 + i] : v[i], send = up ? v[i] : v[16 + i];
            v[i] = keep + dpp_fetch<0x128>(send);
        }
    }
    // bit 2 (xor 4: lanes with the bit clear take lane + 4, the others lane - 4), 8 results
    {
        const bool up = (lane & 4) != 0;
        AUM_UNROLL
        for (int i = 0; i < 8; ++i) {
            const vf keep = up ? v[8 + i] : v[i], send = up ? v[i] : v[8 + i];
            const vf fromhi = dpp_mov<0x104>(send, send);      // row_shl:4  lane i <- i + 4
            const vf fromlo = dpp_mov<0x114>(send, send);      // row_shr:4  lane i <- i - 4
            v[i] = keep + (up ? fromlo : fromhi);
        }
    }
    // bit 1 (xor 2: quad_perm [2,3,0,1]), 4 results
    {
        const bool up = (lane & 2) != 0;
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) {
            const vf keep = up ? v[4 + i] : v[i], send = up ? v[i] : v[4 + i];
            v[i] = keep + dpp_fetch<0x4E>(send);
        }
    }
    // bit 0 (xor 1: quad_perm [1,0,3,2]), 2 results
    {
        const bool up = (lane & 1) != 0;
        AUM_UNROLL
        for (int i = 0; i < 2; ++i) {
            const vf keep = up ? v[2 + i] : v[i], send = up ? v[i] : v[2 + i];
            v[i] = keep + dpp_fetch<0xB1>(send);
        }
    }
    // bit 4 (xor 16): v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second
    vf r;
    {
        const bool up = (lane & 16) != 0;
        const vf keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, send), __builtin_bit_cast(unsigned, send), false, false);
        // first result: rows (r0, r0, r2, r2) of `send`; second: rows (r1, r1, r3, r3): an even row wants its odd neighbour and vice versa
        r = keep + __builtin_bit_cast(float, up ? sw[0] : sw[1]);
    }
    // the two halves of the wave
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, r), __builtin_bit_cast(unsigned, r), false, false);
        // first result: halves (lo, lo) of r; second: (hi, hi)
        r = r + __builtin_bit_cast(float, (lane & 32) ? sw[0] : sw[1]);
    }
    return r;
}
#endif

// (Round 5's sums through an LDS transposition tile and round 4's sums on the matrix pipe -- lsum_*, wave_sum_mfma_* -- lived here: parity-green,
// measured slower than the butterflies of wave_sum16 / wave_sum32, removed in round 6: HISTORY.md, profiles/r05_ab_lsum.txt, r04_ab_msum.txt.)
// ------------------------------------------------------------------------------------------------
// The associative scan of the selective-scan recurrence x' = a*x + b over the 64 lanes.
// Each lane holds the composition (P, S) of its own K steps:  x_out = P * x_in + S.
// Operator (earlier) o (later):  (P1,S1) o (P2,S2) = (P1*P2, P2*S1 + S2)      (SURVEY 8a')
//
// REV = false: time runs lane 0 -> 63 (inclusive prefix scan); REV = true: 63 -> 0 (suffix scan).
// On return S holds, for every lane, the state AFTER that lane's last step given a zero state before
// the first lane (a carry-in is folded into the first lane's S by the caller).  P is left holding
// partial products and must not be used afterwards.
// Intra-row: 4 Hillis-Steele steps with DPP row_shr / row_shl (identity filled into invalid lanes).
// Cross-row: the three row totals travel through SGPRs via v_readlane_b32.
// ------------------------------------------------------------------------------------------------
#ifdef AUM_EMU
template <bool REV> AUM_DEV void wave_scan_affine(vf& P, vf& S) {
#define AUM_SCAN_STEP(N)                                                               \
    {                                                                                  \
        vf Pp = REV ? dpp_row_shl<N>(P, splat(1.f)) : dpp_row_shr<N>(P, splat(1.f));   \
        vf Sp = REV ? dpp_row_shl<N>(S, splat(0.f)) : dpp_row_shr<N>(S, splat(0.f));   \
        S = vfma(P, Sp, S);                                                            \
        P = P * Pp;                                                                    \
    }
    AUM_SCAN_STEP(1) AUM_SCAN_STEP(2) AUM_SCAN_STEP(4) AUM_SCAN_STEP(8)
#undef AUM_SCAN_STEP
    // row totals sit in the last (first, for REV) lane of each 16-lane row
    const int t0 = REV ? 48 : 15, t1 = REV ? 32 : 31, t2 = REV ? 16 : 47;
    const float P0 = readlane(P, t0), S0 = readlane(S, t0);
    const float P1 = readlane(P, t1), S1 = readlane(S, t1);
    const float P2 = readlane(P, t2), S2 = readlane(S, t2);
    // exclusive composition entering rows 1,2,3 (in scan order)
    const float E1P = P0, E1S = S0;
    const float E2P = P0 * P1, E2S = vfma(P1, S0, S1);
    const float E3S = vfma(P2, E2S, S2);
    const vi row = lane_id() >> 4;                        // 0..3
    const vi ord = REV ? (3 - row) : row;                 // position of this row in scan order
    const vf inS = vsel(ord == 1, splat(E1S), vsel(ord == 2, splat(E2S), vsel(ord == 3, splat(E3S), splat(0.f))));
    S = vfma(P, inS, S);
    (void)E1P; (void)E2P;
}
#else
// Device version: every Hillis-Steele step is two DPP-fused VOP2 instructions,
//     v_fmac_f32_dpp S, S, P <shift>   ; S[l] += S[l-d] * P[l]      (lanes without a source are not written)
//     v_mul_f32_dpp  P, P, P <shift>   ; P[l] *= P[l-d]
// instead of {v_mov old, v_mov_dpp} x2 + v_fma + v_mul.  The prefix form finishes across the 16-lane rows with
// row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3); the suffix form has no broadcast in that direction and keeps
// the v_readlane cross-row step.  One s_nop per step covers the "VALU write -> DPP read" hazard (2 wait states);
// the leading s_nop 1 covers operands the compiler may have written just before the statement.
template <bool REV> AUM_DEV void wave_scan_affine(vf& P, vf& S) {
    if constexpr (!REV) {
        asm volatile(
            "s_nop 1\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
            "s_nop 0"
            : "+v"(P), "+v"(S));
    } else {
        asm volatile(
            "s_nop 1\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0\n\t"
            "v_fmac_f32_dpp %1, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 0"
            : "+v"(P), "+v"(S));
        // rows in scan order are 3,2,1,0; their totals sit in the first lane of each row
        const float P0 = readlane(P, 48), S0 = readlane(S, 48);
        const float P1 = readlane(P, 32), S1 = readlane(S, 32);
        const float S2 = readlane(S, 16), P2 = readlane(P, 16);
        const float E2S = vfma(P1, S0, S1);
        const float E3S = vfma(P2, E2S, S2);
        const vi ord = 3 - (lane_id() >> 4);
        const vf inS = vsel(ord == 1, S0, vsel(ord == 2, E2S, vsel(ord == 3, E3S, 0.f)));
        S = vfma(P, inS, S);
        (void)P0;
    }
}
#endif

// Two independent scans at once (the two rows a wave carries).  On the device the two chains are interleaved in one
// statement, which also fills the DPP wait states that the single-chain version pads with s_nop.
#ifdef AUM_EMU
template <bool REV> inline void wave_scan_affine2(vf2& P, vf2& S) {
    wave_scan_affine<REV>(P.x, S.x);
    wave_scan_affine<REV>(P.y, S.y);
}
#else
#define AUM_SCAN2_STEP(ctrl)                                              \
    "v_fmac_f32_dpp %2, %2, %0 " ctrl "\n\t"                              \
    "v_mul_f32_dpp %0, %0, %0 " ctrl "\n\t"                               \
    "v_fmac_f32_dpp %3, %3, %1 " ctrl "\n\t"                              \
    "v_mul_f32_dpp %1, %1, %1 " ctrl "\n\t"
template <bool REV> AUM_DEV void wave_scan_affine2(vf2& P, vf2& S) {
    float p0 = P.x, p1 = P.y, s0 = S.x, s1 = S.y;
    if constexpr (!REV) {
        asm volatile("s_nop 1\n\t"
                     AUM_SCAN2_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
                     AUM_SCAN2_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                     AUM_SCAN2_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
                     AUM_SCAN2_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                     AUM_SCAN2_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                     AUM_SCAN2_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                     "s_nop 0"
                     : "+v"(p0), "+v"(p1), "+v"(s0), "+v"(s1));
    } else {
        asm volatile("s_nop 1\n\t"
                     AUM_SCAN2_STEP("row_shl:1 row_mask:0xf bank_mask:0xf")
                     AUM_SCAN2_STEP("row_shl:2 row_mask:0xf bank_mask:0xf")
                     AUM_SCAN2_STEP("row_shl:4 row_mask:0xf bank_mask:0xf")
                     AUM_SCAN2_STEP("row_shl:8 row_mask:0xf bank_mask:0xf")
                     "s_nop 0"
                     : "+v"(p0), "+v"(p1), "+v"(s0), "+v"(s1));
        const vi ord = 3 - (lane_id() >> 4);
        {
            const float S0 = readlane(s0, 48), P1 = readlane(p0, 32), S1 = readlane(s0, 32), P2 = readlane(p0, 16), S2 = readlane(s0, 16);
            const float E2S = vfma(P1, S0, S1), E3S = vfma(P2, E2S, S2);
            s0 = vfma(p0, vsel(ord == 1, S0, vsel(ord == 2, E2S, vsel(ord == 3, E3S, 0.f))), s0);
        }
        {
            const float S0 = readlane(s1, 48), P1 = readlane(p1, 32), S1 = readlane(s1, 32), P2 = readlane(p1, 16), S2 = readlane(s1, 16);
            const float E2S = vfma(P1, S0, S1), E3S = vfma(P2, E2S, S2);
            s1 = vfma(p1, vsel(ord == 1, S0, vsel(ord == 2, E2S, vsel(ord == 3, E3S, 0.f))), s1);
        }
    }
    P = vf2{p0, p1};
    S = vf2{s0, s1};
}
#undef AUM_SCAN2_STEP
#endif

}  // namespace aum
