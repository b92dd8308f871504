// scan_state_kernels.h -- backward selective scan for the AuM row shape L = 513, dstate <= 16: WAVES OWN STATES, ROWS STREAM.
//
// scan_half_kernels.h / scan_row_kernels.h give every wave whole rows: each wave loads a row, prepares it (softplus, gate), walks
// the 16 states, writes the row's gradients.  Measured on MI355X (B = 64, bf16, profiles/r02_*): the vector ALU is busy 55-73 % of
// the time.  (a) The dB/dC tiles are shared by the rows of a workgroup and need an ordered read-add-write -- a rotated state order
// with a barrier per state step -- which keeps the 12 waves of a CU in lock-step, so their memory phases (two HBM round trips per
// row) coincide instead of hiding behind each other's arithmetic; (b) 4 fp32 tiles = 148 KB of LDS: one workgroup per CU, 165
// VGPRs: three waves per SIMD.
//
// Here one workgroup = 16 waves = the 16 states.  Wave n keeps B[b][n][:], C[b][n][:] and its dB/dC accumulators IN REGISTERS for
// the whole workgroup (no LDS tiles, no read-add-write, no ordering between waves), and the rows of the workgroup stream past:
//   P1  (4 waves, 128 steps each, two steps per lane, packed math)  raw u, delta, dout, z, out_pre of row r+1 -- fetched a row
//       ahead into registers, in storage format -- -> delta = softplus(.), delta*u, dy = dout*silu(z), dz (stored), softplus' -> LDS
//   P2  (all 16 waves)  state n of row r, both directions: states from the lane-entry checkpoint (x_lane, fetched two rows
//       ahead), adjoint scan, dB/dC accumulation in registers, this state's share of G = sum_n g B and DA = sum_n A g a x -> LDS
//   P3  (4 waves)  sum the 16 shares, du and ddelta of row r-1 (stored), dD / ddelta_bias partial sums
//   tail step (t = 512) of all (direction, state) pairs: one wave, lane 16*d + n, before (P1) and after (P3) the main steps, coupled
//       to them through wave-uniform carries in LDS
// Two workgroup barriers per row instead of sixteen, ~112 KB of LDS, <= 128 VGPRs -> 16 waves per CU.  Needs the forward's
// lane-entry checkpoint (scan_row_kernels.h).
//
// STATUS (round 2, measured on MI355X, B = 64, bf16): parity-green (emulator + GPU, all three direction modes) but NOT the default --
// opt-in through AUM_DBG_STATE_BWD.  One direction: 107 VGPRs, no scratch, 1.16 ms (scanh_bwd: 0.92 ms); both directions fused: the
// second direction body costs ~25 VGPRs more than the 128 of a 16-wave workgroup (36 B of scratch per lane), 2.08 ms (scanh_bwd
// 1.41 ms).  Phase stamps (tools/scans_trace.py, profiles/r02_state_kernel_phase_trace.txt): an iteration takes 7.7 k cycles where
// the state work of four waves per SIMD is ~2.3 k at full VALU rate -- the waves that carry a tail role reach the first barrier last
// (3.0 k state work + 1.7 k of serial, wave-uniform tail arithmetic), and barrier + share hand-off + barrier add ~2 k.  What fixed
// the first version (1.38 ms, 2.6 us per row with the state work ablated): one definition point for every prefetch register (per-role
// branches around the loads made hipcc merge them through v_mov copies behind `s_waitcnt vmcnt(0)`), row parameters in LDS
// instead of per-row parameter loads, batched LDS reads in P3, opaque lane ids in the role code (hoisted 64-bit addresses cost 20
// VGPRs).  Next: tail roles off the critical path, one barrier per row (counter instead of barrier 1), the second
// direction inside 128 VGPRs.
#pragma once
#include "scan_row_kernels.h"

namespace aum {

constexpr int SCANS_NW = 16;                            // waves per workgroup = states
// floats per plane: steps (8 l + i) / (8 l + 4 + i), i < 4, of every lane l; +32 words so that the two planes sit 32 banks apart
// (the two-steps-per-lane accesses of P1 / P3 alternate between the planes every second lane)
constexpr int SCANS_PLANE = 4 * WAVE + 32;
constexpr int SCANS_ROWBUF = 2 * SCANS_PLANE;           // one per-step array of a row: [plane][lane][4]
constexpr int SCANS_NPREP = 5;                          // prepared arrays: delta, delta*u, dy, u, softplus'
constexpr int SCANS_TC = 32;                            // tail lanes: 16*d + n
// LDS map (floats).  Prepared rows and tail inputs are triple-buffered: in one iteration P1 writes row r+1, P2 reads row r, P3
// reads row r-1; the tail-out area is double-buffered (P2 of row r writes while P3-tail reads row r-1).
constexpr int SCANS_OFF_PREP = 0;                                               // [3][NPREP][ROWBUF]
constexpr int SCANS_OFF_PART = SCANS_OFF_PREP + 3 * SCANS_NPREP * SCANS_ROWBUF; // [NW][2][ROWBUF]   G | DA shares
constexpr int SCANS_OFF_TIN = SCANS_OFF_PART + SCANS_NW * 2 * SCANS_ROWBUF;     // [3][4][TC]        a_t | b_t | cc_t | scalars
constexpr int SCANS_TOUT = 2 * SCANS_TC + 4 * SCANS_TC;                         // [TC] x_last | [TC] ga_last | [TC][4] dA shares
constexpr int SCANS_OFF_TOUT = SCANS_OFF_TIN + 3 * 4 * SCANS_TC;                // [2][TOUT]
constexpr int SCANS_OFF_TCONST = SCANS_OFF_TOUT + 2 * SCANS_TOUT;               // [TC] B_512 | [TC] C_512 | [TC] tail dB | [TC] tail dC
// per-row parameters of the workgroup's rows, loaded once: no wave waits for a parameter load inside the row loop
constexpr int SCANS_OFF_ROWC = SCANS_OFF_TCONST + 4 * SCANS_TC;                 // [rows][2]          delta_bias | D
constexpr int SCANS_OFF_ATAB = SCANS_OFF_ROWC + 2 * 64;                         // [rows][2][16]      A | A_b
constexpr int SCANS_LDS_FLOATS = SCANS_OFF_ATAB + 64 * 2 * 16;
enum { SCANS_TS_DL = 0, SCANS_TS_U = 1, SCANS_TS_DY = 2, SCANS_TS_DSP = 3 };   // scalars of the tail step, P1-tail -> P3-tail
AUM_HOSTDEV constexpr int scans_rows() { return 64; }   // rows per workgroup (a row costs every wave one P2 step)
constexpr int SCANS_NSLICE = 5;                         // dD / ddelta_bias partials per (batch, row): 4 P3 slices + the tail

// workspace of this kernel (floats): per-workgroup dB/dC partial rows, per-batch dA, per-(batch, slice) dD / ddelta_bias
// debug trace (AUM_DBG_TRACE, workgroup 0): shader-clock stamps of the pipeline phases, [wave][iteration][8 slots] uint32
constexpr int SCANS_TRACE_ITERS = 24, SCANS_TRACE_SLOTS = 8;
constexpr int SCANS_TRACE_FLOATS = SCANS_NW * SCANS_TRACE_ITERS * SCANS_TRACE_SLOTS;
struct ScanSWs { int64_t pB, pC, pA, pAb, pD, pbias, trace, total; int gpb; };
AUM_HOSTDEV ScanSWs scans_ws_layout(int batch, int dim, int len, int N, bool bidir) {
    ScanSWs w;
    w.gpb = (dim + scans_rows() - 1) / scans_rows();
    const int64_t tile = (int64_t)w.gpb * batch * N * len;
    int64_t o = 0;
    w.pB = o; o += tile;
    w.pC = o; o += tile;
    w.pA = o; o += (int64_t)batch * dim * N;
    w.pAb = o; o += bidir ? (int64_t)batch * dim * N : 0;
    w.pD = o; o += (int64_t)batch * SCANS_NSLICE * dim;
    w.pbias = o; o += (int64_t)batch * SCANS_NSLICE * dim;
    w.trace = o; o += SCANS_TRACE_FLOATS;
    w.total = o;
    return w;
}

// two consecutive elements of a row (element offset t0, only element-aligned) <-> a packed pair.  Fetched values stay in their
// storage format (ScansRaw2) until they are used a row later, so that no conversion waits for the load at the fetch.
template <class T> struct __attribute__((packed, aligned(sizeof(T) < 4 ? sizeof(T) : 4))) scans_pair_t { T e[2]; };
#ifdef AUM_EMU
template <class T> struct ScansRaw2 { vf2 v; };
template <class T> AUM_DEV ScansRaw2<T> scans_fetch2(const T* rp, vi t0) { return ScansRaw2<T>{mk2(gload_u(rp, t0), gload_u(rp, t0 + 1))}; }
template <class T> AUM_DEV vf2 scans_unpack(const ScansRaw2<T>& r) { return r.v; }
template <class T> AUM_DEV float scans_unpack_hi(const ScansRaw2<T>& r) { return r.v.y.v[0]; }
template <class T> AUM_DEV void scans_store2(T* rp, vi t0, vf2 v) {
    gstore(rp, t0, lo2(v), lane_id() >= 0);
    gstore(rp, t0 + 1, hi2(v), lane_id() >= 0);
}
#define AUM_MEM_FENCE() do { } while (0)
#else
// raw bits exactly as the load instruction delivers them (one dword for a pair of 16-bit elements, a zero-extended ushort for a
// single one): nothing is computed on a fetched value until it is used, so no s_waitcnt follows the fetch
struct __attribute__((packed, aligned(2))) scans_u32_t { uint32_t v; };
template <class T> struct ScansRaw2 { uint32_t w0, w1; };
template <class T> AUM_DEV ScansRaw2<T> scans_fetch2(const T* rp, vi t0) {
    ScansRaw2<T> r;
    if constexpr (sizeof(T) == 2) {
        r.w0 = reinterpret_cast<const scans_u32_t*>(rp + (uint32_t)t0)->v;
        r.w1 = 0;
    } else {
        r.w0 = __builtin_bit_cast(uint32_t, rp[(uint32_t)t0]);
        r.w1 = __builtin_bit_cast(uint32_t, rp[(uint32_t)t0 + 1]);
    }
    return r;
}
template <class T> AUM_DEV float scans_bits_to_f32(uint32_t lo16) {
    if constexpr (__is_same(T, bf16_t)) return bits_to_f32(lo16 << 16);
    else return (float)__builtin_bit_cast(_Float16, (uint16_t)lo16);
}
template <class T> AUM_DEV vf2 scans_unpack(const ScansRaw2<T>& r) {
    if constexpr (sizeof(T) == 2) {
        if constexpr (__is_same(T, bf16_t)) return mk2(bits_to_f32(r.w0 << 16), bits_to_f32(r.w0 & 0xffff0000u));
        else return mk2(scans_bits_to_f32<T>(r.w0 & 0xffffu), scans_bits_to_f32<T>(r.w0 >> 16));
    } else {
        return mk2(bits_to_f32(r.w0), bits_to_f32(r.w1));
    }
}
// second element of a fetched pair as a wave-uniform value (the tail wave fetches steps 511 | 512 in every lane)
template <class T> AUM_DEV float scans_unpack_hi(const ScansRaw2<T>& r) {
    const vf2 v = scans_unpack<T>(r);
    return readlane(hi2(v), 0);
}
template <class T> AUM_DEV void scans_store2(T* rp, vi t0, vf2 v) {
    scans_pair_t<T> r;
    if constexpr (sizeof(T) == 2) {
        const uint32_t pk = f32x2_to_elem2<T>(lo2(v), hi2(v));
        __builtin_memcpy(&r, &pk, 4);
    } else {
        f32_to_elem(lo2(v), r.e[0]);
        f32_to_elem(hi2(v), r.e[1]);
    }
    *reinterpret_cast<scans_pair_t<T>*>(rp + (uint32_t)t0) = r;
}
// compiler-only memory fence: LDS values read before it are read again after it (no common-subexpression across it), which is
// what keeps a row's prepared arrays out of the registers between their uses
#define AUM_MEM_FENCE() asm volatile("" ::: "memory")
#endif
// LDS word of step t (< 512) inside a ROWBUF: plane (t & 7) >> 2, lane t >> 3, slot t & 3
AUM_DEV vi scans_word(vi t) { return ((t & 7) >> 2) * SCANS_PLANE + (t >> 3) * 4 + (t & 3); }
// a lane's 8 steps of a ROWBUF array <-> half-packed slots (two 16-byte LDS accesses)
AUM_DEV void scans_lds_read8(const float* buf, vf2 (&m)[4]) {
    const vi lane = lane_id();
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) m[i] = mk2(lds_read(buf, lane * 4 + i), lds_read(buf, lane * 4 + (SCANS_PLANE + i)));
}
AUM_DEV vf2 scans_lds_read_slot(const float* buf, int i) {
    const vi lane = lane_id();
    return mk2(lds_read(buf, lane * 4 + i), lds_read(buf, lane * 4 + (SCANS_PLANE + i)));
}
AUM_DEV void scans_lds_write8(float* buf, const vf2 (&m)[4]) {
    const vi lane = lane_id();
    AUM_UNROLL
    for (int i = 0; i < 4; ++i) {
        lds_write(buf, lane * 4 + i, lo2(m[i]));
        lds_write(buf, lane * 4 + (SCANS_PLANE + i), hi2(m[i]));
    }
}
AUM_DEV vf2 scans_lds_read2(const float* buf, vi word) { return mk2(lds_read(buf, word), lds_read(buf, word + 1)); }
AUM_DEV void scans_lds_write2(float* buf, vi word, vf2 v) {
    lds_write(buf, word, lo2(v));
    lds_write(buf, word + 1, hi2(v));
}

template <bool V> struct ScansBool { static constexpr bool value = V; };
#ifdef AUM_EMU
#define AUM_NOUNROLL
#else
#define AUM_NOUNROLL _Pragma("nounroll")
#endif

// registers a wave carries from row to row
template <class T> struct ScansWave {
    vf2 Bn[4], Cn[4], dBacc[4], dCacc[4];   // this wave's state: B, C of the batch entry; dB, dC summed over the workgroup's rows
    vf2 G[4], DA[4];                        // this state's shares of the current row, between the state work and their hand-off
    vf xin[1][2];                           // lane-entry states of the next row (direction slot)
    ScansRaw2<T> raw[5];                    // u, delta, dout, z, out_pre of the next row, two steps per lane (P1 waves: their
                                            // slice; the P1-tail wave: steps 511 | 512 in every lane)
};

template <class T, int MODE>
AUM_DEV void scans_bwd(const AumScanBwdArgs& p, int wg, float* lds) {
    constexpr bool BI = MODE == 2;
    constexpr int ND = BI ? 2 : 1;
    constexpr int NW = SCANS_NW;
    constexpr bool REV0 = scanr_slot_rev<MODE>(0);
    const int N = p.dstate;
    const ScanSWs L = scans_ws_layout(p.batch, p.dim, p.len, N, BI);
    float* ws = (float*)p.workspace;
    const int b = wg / L.gpb, g_idx = wg % L.gpb;
    const int eb = g_idx * scans_rows();
    const int R = (p.dim - eb) < scans_rows() ? (p.dim - eb) : scans_rows();      // rows of this workgroup
    const bool softplus = (p.flags & AUM_SCAN_SOFTPLUS) != 0;
    const float ndir = BI ? 2.f : 1.f;
    float* prep = lds + SCANS_OFF_PREP;
    float* part = lds + SCANS_OFF_PART;
    float* tin = lds + SCANS_OFF_TIN;
    float* tout = lds + SCANS_OFF_TOUT;
    float* tconst = lds + SCANS_OFF_TCONST;
    float* rowc = lds + SCANS_OFF_ROWC;
    float* atab = lds + SCANS_OFF_ATAB;
    static_assert(scans_rows() == 64, "row tables are sized for 64 rows per workgroup");
    auto row_bias = [&](int r) { return readlane(lds_read(rowc, spl_i(2 * r)), 0); };
    auto row_D = [&](int r) { return readlane(lds_read(rowc, spl_i(2 * r + 1)), 0); };
    ScansWave<T> st[AUM_PER_WAVE(NW)];

    auto urow = [&](int r) { return row_ptr<T>(p.u, (int64_t)b * p.u_bs + (int64_t)(eb + r) * p.u_ds); };
    auto drow = [&](int r) { return row_ptr<T>(p.delta, (int64_t)b * p.delta_bs + (int64_t)(eb + r) * p.delta_ds); };
    auto grow = [&](int r) { return row_ptr<T>(p.dout, (int64_t)b * p.dout_bs + (int64_t)(eb + r) * p.dout_ds); };
    auto zrow = [&](int r) { return row_ptr<T>(p.z, (int64_t)b * p.z_bs + (int64_t)(eb + r) * p.z_ds); };
    auto yrow = [&](int r) { return row_ptr<T>(p.out_pre, (int64_t)b * p.out_bs + (int64_t)(eb + r) * p.out_ds); };
    // Prefetch of row r for the P1 roles.  EVERY wave executes the same five loads (waves 0-3: their 128-step slice, two steps per
    // lane; wave 8: steps 511 | 512 in every lane; the others repeat a slice and never look at the result): the registers that
    // carry the data to the next iteration then have ONE definition per iteration -- with per-role branches around the loads hipcc
    // merges the variants through v_mov copies at the end of the branch, behind an s_waitcnt vmcnt(0) for the loads just issued.
    auto fetch_raw = [&](ScansWave<T>& S, int w, int r) {
        const int rc = r < R ? r : R - 1;
        const vi lane = opaque_i(lane_id());
        const vi t0 = w == 8 ? spl_i(SCANR_LEN - 2) : lane * 2 + (w & 3) * 128;
        S.raw[0] = scans_fetch2<T>(urow(rc), t0);
        S.raw[1] = scans_fetch2<T>(drow(rc), t0);
        S.raw[2] = scans_fetch2<T>(grow(rc), t0);
        if (p.z) {
            S.raw[3] = scans_fetch2<T>(zrow(rc), t0);
            S.raw[4] = scans_fetch2<T>(yrow(rc), t0);
        }
    };
    // A of row r for the tail roles, lane 16*d + n, from the workgroup's table
    auto tail_A = [&](int r) {
        const vi lane = lane_id();
        const vi tl = vmin_i(lane, SCANS_TC - 1);
        const vm tvalid = (lane < 16 * ND) && ((lane & 15) < N);
        return vsel(tvalid, lds_read(atab, tl + r * 32), splat(0.f));
    };
    auto fetch_xin = [&](ScansWave<T>& S, int w, int r) {
        constexpr int PAR = 0;
        const int rc = r < R ? r : R - 1;
        const int n = w < N ? w : 0;
        const float* ck = p.x_lane + ((int64_t)b * p.dim + eb + rc) * ND * N * WAVE;
        AUM_UNROLL
        for (int d = 0; d < ND; ++d) {
            S.xin[PAR][d] = gload(ck + ((int64_t)d * N + n) * WAVE, opaque_i(lane_id()), lane_id() >= 0);
        }
    };
    // ---- P1: prepare row r (slice w of the main steps) ----
    auto p1_slice = [&](ScansWave<T>& S, int w, int r) {
        const vi t0 = opaque_i(lane_id()) * 2 + w * 128;
        const vi word = scans_word(t0);
        float* pb = prep + (r % 3) * SCANS_NPREP * SCANS_ROWBUF;
        const float bias = row_bias(r);
        const vf2 one2 = spl2(splat(1.f));
        const vf2 uu = scans_unpack<T>(S.raw[0]);
        const vf2 dr = scans_unpack<T>(S.raw[1]) + spl2(splat(bias));
        const vf2 dl = softplus ? vsoftplus2(dr) : dr;
        vf2 dsp = one2;
        if (softplus) {
            const vf2 sg = vsigmoid2(dr);
            dsp = mk2(vsel(lo2(dr) > 20.f, splat(1.f), lo2(sg)), vsel(hi2(dr) > 20.f, splat(1.f), hi2(sg)));
        }
        vf2 go = scans_unpack<T>(S.raw[2]);
        if (p.z) {
            const vf2 zz = scans_unpack<T>(S.raw[3]), yp = scans_unpack<T>(S.raw[4]);
            const vf2 sg = vsigmoid2(zz);
            const vf2 dzv = go * yp * sg * vfma2(zz, one2 - sg, one2);
            go = go * zz * sg;
            scans_store2<T>(row_ptr_w<T>(p.dz, (int64_t)b * p.dz_bs + (int64_t)(eb + r) * p.dz_ds), t0, dzv);
        }
        scans_lds_write2(pb + 0 * SCANS_ROWBUF, word, dl);
        scans_lds_write2(pb + 1 * SCANS_ROWBUF, word, dl * uu);
        scans_lds_write2(pb + 2 * SCANS_ROWBUF, word, go);
        scans_lds_write2(pb + 3 * SCANS_ROWBUF, word, uu);
        scans_lds_write2(pb + 4 * SCANS_ROWBUF, word, dsp);
    };
    // ---- P1, tail step of row r: a_t, b_t, cc_t for every (direction slot, state), the scalars for P3-tail, dz_512 ----
    auto p1_tail = [&](ScansWave<T>& S, int r) {
        const vi lane = opaque_i(lane_id());
        const vi tlc = vmin_i(lane, SCANS_TC - 1);
        const vf B_t = lds_read(tconst, tlc), C_t = lds_read(tconst + SCANS_TC, tlc);
        float* tb = tin + (r % 3) * 4 * SCANS_TC;
        const float bias = row_bias(r);
        const float u_t = scans_unpack_hi<T>(S.raw[0]), raw_t = scans_unpack_hi<T>(S.raw[1]) + bias;
        float go_t = scans_unpack_hi<T>(S.raw[2]);
        const float dl_t = softplus ? vsoftplus(raw_t) : raw_t;
        const float dsp_t = (softplus && !(raw_t > 20.f)) ? vsigmoid(raw_t) : 1.f;
        if (p.z) {
            const float z_t = scans_unpack_hi<T>(S.raw[3]), yp_t = scans_unpack_hi<T>(S.raw[4]);
            const float sg = vsigmoid(z_t);
            const float dz_t = go_t * yp_t * sg * vfma(z_t, 1.f - sg, 1.f);
            go_t = go_t * z_t * sg;
            gstore(row_ptr_w<T>(p.dz, (int64_t)b * p.dz_bs + (int64_t)(eb + r) * p.dz_ds), spl_i(SCANR_LEN - 1), splat(dz_t), lane == 0);
        }
        const vm tl = lane < SCANS_TC;
        lds_write_m(tb + 0 * SCANS_TC, tlc, vexp2(tail_A(r) * splat(dl_t * LOG2E)), tl);
        lds_write_m(tb + 1 * SCANS_TC, tlc, B_t * splat(dl_t * u_t), tl);
        lds_write_m(tb + 2 * SCANS_TC, tlc, C_t * splat(go_t), tl);
        vf sc = splat(0.f);
        sc = vsel(lane == SCANS_TS_DL, splat(dl_t), sc);
        sc = vsel(lane == SCANS_TS_U, splat(u_t), sc);
        sc = vsel(lane == SCANS_TS_DY, splat(go_t), sc);
        sc = vsel(lane == SCANS_TS_DSP, splat(dsp_t), sc);
        lds_write_m(tb + 3 * SCANS_TC, tlc, sc, tl);
    };
    // ---- P2: state w of row r, every direction slot ----
    auto p2_state = [&](ScansWave<T>& S, int w, int r) {
        constexpr int PAR = 0;
        const vi lane = lane_id();
        const float* pb = prep + (r % 3) * SCANS_NPREP * SCANS_ROWBUF;
        const float* tb = tin + (r % 3) * 4 * SCANS_TC;
        float* to = tout + (r & 1) * SCANS_TOUT;
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) { S.G[i] = spl2(splat(0.f)); S.DA[i] = spl2(splat(0.f)); }
        vf xin_cur[ND];
        float Acur[ND];
        AUM_UNROLL
        for (int d = 0; d < ND; ++d) {
            xin_cur[d] = S.xin[PAR][d];
            Acur[d] = readlane(lds_read(atab, spl_i((r * 2 + d) * 16 + (w < N ? w : 0))), 0);
        }
        AUM_SCHED_FENCE();
        fetch_xin(S, w, r + 1);              // entry states and A of the next row: one row of arithmetic ahead, every wave
        AUM_SCHED_FENCE();
        if (w >= N || (p.flags & AUM_DBG_SKIP_STATES)) return;
        vf2 cc[4];
        {   // c_t = dy_t C_t serves both directions; delta, delta*u and dy themselves are read from LDS again where they are
            // needed (LDS reads are cheap here, registers are not: 128 VGPRs = 16 waves per CU)
            vf2 dy[4];
            scans_lds_read8(pb + 2 * SCANS_ROWBUF, dy);
            AUM_UNROLL
            for (int i = 0; i < 4; ++i) cc[i] = dy[i] * S.Cn[i];
        }
        // one direction slot; REV is a compile-time property of the slot.  The two slots of the bidirectional kernel run as the
        // iterations of a loop the compiler may not unroll: in one basic block it interleaves them and needs ~50 VGPRs more
        // (183 instead of ~130: measured with hipcc 7.2), i.e. 200 bytes of scratch per lane at the 128 of a 16-wave workgroup
        auto dir_body = [&](auto rev_tag, int d) {
            constexpr bool rev = decltype(rev_tag)::value;
            AUM_MEM_FENCE();
            const float Araw = d == 0 ? Acur[0] : Acur[ND - 1];
            const float An = Araw * LOG2E;
            vf2 a[4], x[4], m[4], g[4];
            {
                vf2 dl[4];
                scans_lds_read8(pb + 0 * SCANS_ROWBUF, dl);
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) a[i] = vexp2_2(dl[i] * spl2(splat(An)));
            }
            const vf x_in = d == 0 ? xin_cur[0] : xin_cur[ND - 1];
            vf gin, gS;
            {
                vf2 bb[4];      // b_t = delta_t u_t B_t: only the state recurrence needs it
                scans_lds_read8(pb + 1 * SCANS_ROWBUF, bb);
                AUM_UNROLL
                for (int i = 0; i < 4; ++i) bb[i] = bb[i] * S.Bn[i];
                scanr_states_from_entry<rev>(a, bb, x_in, x);
            }
            AUM_MEM_FENCE();
            if constexpr (!rev) {
                const vf a_edge = lds_read(tb + 0 * SCANS_TC, spl_i(16 * d + w));
                const vf gcin = lds_read(tb + 2 * SCANS_TC, spl_i(16 * d + w));
                m[0] = a[1]; m[1] = a[2]; m[2] = a[3];
                m[3] = mk2(hi2(a[0]), dpp_wave_shl1(lo2(a[0]), a_edge));
                scanr_affine<true, true>(m, cc, gcin, g, gin, gS);
                lds_write_m(to, spl_i(16 * d + w), hi2(x[3]), lane == WAVE - 1);          // x_511
            } else {
                m[0] = mk2(dpp_wave_shr1(hi2(a[3]), splat(1.f)), lo2(a[3]));
                m[1] = a[0]; m[2] = a[1]; m[3] = a[2];
                scanr_affine<false, false>(m, cc, splat(0.f), g, gin, gS);
            }
            (void)gin; (void)gS;
            AUM_MEM_FENCE();
            vf2 dAl = spl2(splat(0.f));
            const vf2 Ar = spl2(splat(Araw));
            AUM_UNROLL
            for (int i = 0; i < 4; ++i) {
                const vf2 dlu_i = scans_lds_read_slot(pb + 1 * SCANS_ROWBUF, i);
                const vf2 dy_i = scans_lds_read_slot(pb + 2 * SCANS_ROWBUF, i);
                const vf2 dl_i = scans_lds_read_slot(pb + 0 * SCANS_ROWBUF, i);
                vf2 xprev;
                if (!rev) xprev = i == 0 ? mk2(x_in, lo2(x[3])) : x[i > 0 ? i - 1 : 0];
                else xprev = i == 3 ? mk2(hi2(x[0]), x_in) : x[i < 3 ? i + 1 : 3];
                const vf2 ga = g[i] * a[i];
                if (rev && i == 3) lds_write_m(to + SCANS_TC, spl_i(16 * d + w), hi2(ga), lane == WAVE - 1);      // a_511 g_511
                const vf2 h = ga * xprev;
                S.G[i] = vfma2(g[i], S.Bn[i], S.G[i]);
                S.DA[i] = vfma2(Ar, h, S.DA[i]);
                S.dBacc[i] = vfma2(g[i], dlu_i, S.dBacc[i]);
                S.dCacc[i] = vfma2(dy_i, x[i], S.dCacc[i]);
                dAl = vfma2(dl_i, h, dAl);
            }
            // dA of (row, state, slot): the four 16-lane-row sums, added up by the P3-tail wave
            lds_write_m(to + 2 * SCANS_TC, (lane >> 4) + (16 * d + w) * 4, row_sum16(lo2(dAl) + hi2(dAl)), (lane & 15) == 0);
        };
        if constexpr (!BI) {
            dir_body(ScansBool<REV0>{}, 0);
        } else {
            const int nd_rt = p.A_b ? 2 : 1;         // always 2 here; opaque to the optimiser on purpose
            AUM_NOUNROLL
            for (int d = 0; d < nd_rt; ++d) {
                if (d == 0) dir_body(ScansBool<false>{}, 0);
                else dir_body(ScansBool<true>{}, 1);
            }
        }
    };
    // hand-off of a row's P2 results (after the barrier that ends P3 of the previous row)
    auto p2_publish = [&](ScansWave<T>& S, int w) {
        scans_lds_write8(part + (w * 2 + 0) * SCANS_ROWBUF, S.G);
        scans_lds_write8(part + (w * 2 + 1) * SCANS_ROWBUF, S.DA);
    };
    // ---- P3: du, ddelta of row r (slice w), dD / ddelta_bias partial sums ----
    auto p3_slice = [&](int w, int r) {
        const vi lane = opaque_i(lane_id());
        const vi t0 = lane * 2 + w * 128;
        const vi word = scans_word(t0);
        const float* pb = prep + (r % 3) * SCANS_NPREP * SCANS_ROWBUF;
        const int e = eb + r;
        // the 16 shares (idle states publish zeros), eight 8-byte reads in flight at a time: one read per s_waitcnt -- what the
        // compiler makes of the plain loop -- is 32 serial LDS round trips, ~4000 cycles per row on the critical path
        vf2 G = spl2(splat(0.f)), DA = spl2(splat(0.f));
        AUM_UNROLL
        for (int s0 = 0; s0 < SCANS_NW; s0 += 4) {
            vf2 tg[4], td[4];
            AUM_UNROLL
            for (int k = 0; k < 4; ++k) {
                tg[k] = scans_lds_read2(part + ((s0 + k) * 2 + 0) * SCANS_ROWBUF, word);
                td[k] = scans_lds_read2(part + ((s0 + k) * 2 + 1) * SCANS_ROWBUF, word);
            }
            AUM_SCHED_FENCE();
            G = G + ((tg[0] + tg[1]) + (tg[2] + tg[3]));
            DA = DA + ((td[0] + td[1]) + (td[2] + td[3]));
            AUM_SCHED_FENCE();
        }
        const vf2 dl = scans_lds_read2(pb + 0 * SCANS_ROWBUF, word), dy = scans_lds_read2(pb + 2 * SCANS_ROWBUF, word);
        const vf2 uu = scans_lds_read2(pb + 3 * SCANS_ROWBUF, word), dsp = scans_lds_read2(pb + 4 * SCANS_ROWBUF, word);
        const float Dn = ndir * row_D(r);
        const vf2 du = vfma2(dl, G, dy * spl2(splat(Dn)));
        const vf2 dd = vfma2(uu, G, DA) * dsp;
        scans_store2<T>(row_ptr_w<T>(p.du, (int64_t)b * p.du_bs + (int64_t)e * p.du_ds), t0, du);
        scans_store2<T>(row_ptr_w<T>(p.ddelta, (int64_t)b * p.ddelta_bs + (int64_t)e * p.ddelta_ds), t0, dd);
        const vf2 dDl = dy * uu;
        const float sD = ndir * wave_sum(lo2(dDl) + hi2(dDl));
        const float sb = wave_sum(lo2(dd) + hi2(dd));
        gstore(ws + L.pD + ((int64_t)b * SCANS_NSLICE + w) * p.dim + e, spl_i(0), splat(sD), lane == 0);
        gstore(ws + L.pbias + ((int64_t)b * SCANS_NSLICE + w) * p.dim + e, spl_i(0), splat(sb), lane == 0);
    };
    // ---- P3, tail step of row r ----
    auto p3_tail = [&](ScansWave<T>& S, int r) {
        const vi lane = opaque_i(lane_id());
        const vi tlc = vmin_i(lane, SCANS_TC - 1);
        const vf A_t = tail_A(r), B_t = lds_read(tconst, tlc);
        const float* to = tout + (r & 1) * SCANS_TOUT;
        const float* tb = tin + (r % 3) * 4 * SCANS_TC;
        const int e = eb + r;
        const vm tvalid = (lane < 16 * ND) && ((lane & 15) < N);
        const vm tfwd = MODE == 0 ? (lane >= 0) : MODE == 1 ? (lane < 0) : (lane < 16);
        const vf a_t = lds_read(tb + 0 * SCANS_TC, tlc), b_t = lds_read(tb + 1 * SCANS_TC, tlc), cc_t = lds_read(tb + 2 * SCANS_TC, tlc);
        const vf sc = lds_read(tb + 3 * SCANS_TC, tlc);          // the four scalars of the tail step: one LDS round trip
        const float dl_t = readlane(sc, SCANS_TS_DL), u_t = readlane(sc, SCANS_TS_U);
        const float dy_t = readlane(sc, SCANS_TS_DY), dsp_t = readlane(sc, SCANS_TS_DSP);
        const vf xm = lds_read(to, tlc), gm = lds_read(to + SCANS_TC, tlc);
        const vf x_t = vsel(tfwd, vfma(a_t, xm, b_t), b_t);
        const vf g_t = vsel(tfwd, cc_t, cc_t + gm);
        const vf h_t = vsel(tfwd && tvalid, g_t * a_t * xm, splat(0.f));
        const float G_t = wave_sum(vsel(tvalid, g_t * B_t, splat(0.f)));
        const float DA_t = wave_sum(A_t * h_t);
        // tail dB / dC of the workgroup's rows: this wave's private accumulators in LDS
        lds_write_m(tconst + 2 * SCANS_TC, tlc, lds_read(tconst + 2 * SCANS_TC, tlc) + vsel(tvalid, g_t * splat(dl_t * u_t), splat(0.f)), lane < SCANS_TC);
        lds_write_m(tconst + 3 * SCANS_TC, tlc, lds_read(tconst + 3 * SCANS_TC, tlc) + vsel(tvalid, x_t * splat(dy_t), splat(0.f)), lane < SCANS_TC);
        // dA of the row: the four 16-lane-row shares of every (direction slot, state) + the tail step of the forward-time slot
        vf dAv = h_t * splat(dl_t);
        AUM_UNROLL
        for (int q = 0; q < 4; ++q) dAv = dAv + lds_read(to + 2 * SCANS_TC, tlc * 4 + q);
        if (!(p.flags & AUM_DBG_SKIP_PARTIALS)) {
            const vi tn = vmin_i(lane & 15, N - 1);
            gstore(ws + L.pA + ((int64_t)b * p.dim + e) * N, tn, dAv, tvalid && (lane < 16));
            if (BI) gstore(ws + L.pAb + ((int64_t)b * p.dim + e) * N, tn, dAv, tvalid && (lane >= 16));
        }
        const float Dn = ndir * row_D(r);
        const float du_t = vfma(dl_t, G_t, dy_t * Dn);
        const float dd_t = vfma(u_t, G_t, DA_t) * dsp_t;
        gstore(row_ptr_w<T>(p.du, (int64_t)b * p.du_bs + (int64_t)e * p.du_ds), spl_i(SCANR_LEN - 1), splat(du_t), lane == 0);
        gstore(row_ptr_w<T>(p.ddelta, (int64_t)b * p.ddelta_bs + (int64_t)e * p.ddelta_ds), spl_i(SCANR_LEN - 1), splat(dd_t), lane == 0);
        gstore(ws + L.pD + ((int64_t)b * SCANS_NSLICE + 4) * p.dim + e, spl_i(0), splat(ndir * dy_t * u_t), lane == 0);
        gstore(ws + L.pbias + ((int64_t)b * SCANS_NSLICE + 4) * p.dim + e, spl_i(0), splat(dd_t), lane == 0);
    };
    // tail column of B / C of this batch entry, lane 16*d + n
    auto tail_bc = [&](const void* base, int64_t bs, int64_t ns) {
        const vi lane = lane_id();
        const vi tn = vmin_i(lane & 15, N - 1);
        return gload(row_ptr<T>(base, (int64_t)b * bs), tn * (int)ns + (SCANR_LEN - 1), (lane < 16 * ND) && ((lane & 15) < N));
    };

    // ---- set-up: this wave's B / C row, zero accumulators, first fetches; P1 of row 0 ----
    AUM_FOR_EACH_WAVE(w, NW) {
        ScansWave<T>& S = st[AUM_W(w)];
        const int n = w < N ? w : 0;
        scanr_row_read<T>(row_ptr<T>(p.B, (int64_t)b * p.B_bs + (int64_t)n * p.B_ns), S.Bn);
        scanr_row_read<T>(row_ptr<T>(p.C, (int64_t)b * p.C_bs + (int64_t)n * p.C_ns), S.Cn);
        AUM_UNROLL
        for (int i = 0; i < 4; ++i) { S.dBacc[i] = spl2(splat(0.f)); S.dCacc[i] = spl2(splat(0.f)); S.G[i] = spl2(splat(0.f)); S.DA[i] = spl2(splat(0.f)); }
        AUM_UNROLL
        for (int i = 0; i < 5; ++i) S.raw[i] = ScansRaw2<T>{};
        S.xin[0][0] = S.xin[0][1] = splat(0.f);
        fetch_xin(S, w, 0);
        {   // the row tables: delta_bias, D and A (A_b) of this workgroup's rows
            const vi i = lane_id() + w * WAVE;               // 16 waves x 64 lanes = 1024 >= 64 rows x 16 states
            const vi rr = vmin_i(i >> 4, R - 1), nn = vmin_i(i & 15, N - 1);
            const vm ok = (i < 64 * 16) && ((i & 15) < N) && ((i >> 4) < R);
            lds_write_m(atab, (i >> 4) * 32 + (i & 15), gload(p.A + (int64_t)eb * N, rr * N + nn, ok), i < 64 * 16);
            if (BI) lds_write_m(atab, (i >> 4) * 32 + 16 + (i & 15), gload(p.A_b + (int64_t)eb * N, rr * N + nn, ok), i < 64 * 16);
            if (w == 0) {
                const vi rl = vmin_i(lane_id(), R - 1);
                const vm okr = lane_id() < R;
                lds_write(rowc, lane_id() * 2, p.delta_bias ? gload(p.delta_bias + eb, rl, okr) : splat(0.f));
                lds_write(rowc, lane_id() * 2 + 1, p.D ? gload(p.D + eb, rl, okr) : splat(0.f));
            }
        }
        if (w == 8) {       // constants and accumulators of the two tail roles; the tail-out areas start at zero (idle states stay zero)
            const vi tlc = vmin_i(lane_id(), SCANS_TC - 1);
            const vm tl = lane_id() < SCANS_TC;
            lds_write_m(tconst, tlc, tail_bc(p.B, p.B_bs, p.B_ns), tl);
            lds_write_m(tconst + SCANS_TC, tlc, tail_bc(p.C, p.C_bs, p.C_ns), tl);
            lds_write_m(tconst + 2 * SCANS_TC, tlc, splat(0.f), tl);
            lds_write_m(tconst + 3 * SCANS_TC, tlc, splat(0.f), tl);
            for (int i0 = 0; i0 < 2 * SCANS_TOUT; i0 += WAVE) lds_write_m(tout, lane_id() + i0, splat(0.f), lane_id() + i0 < 2 * SCANS_TOUT);
            wave_lds_fence();
        }
        fetch_raw(S, w, 0);
    }
    AUM_WG_BARRIER();              // row tables and tail constants are in LDS
    AUM_FOR_EACH_WAVE(w, NW) {
        ScansWave<T>& S = st[AUM_W(w)];
        if (w < 4) p1_slice(S, w, 0);
        else if (w == 8) p1_tail(S, 0);
        AUM_SCHED_FENCE();
        fetch_raw(S, w, 1);
    }
    AUM_WG_BARRIER();
    // ---- the rows: iteration r runs P2(r) on all waves, P1(r+1) on waves 0-3 and 8 (AFTER their P2: the raw loads of row r+1
    // were issued an iteration ago and get the whole P2 to land), P3(r-1) on waves 4-7 and 9 (BEFORE their P2: its inputs are
    // complete at the barrier) ----
#ifdef AUM_EMU
#define SCANS_STAMP(w, r, slot) do { } while (0)
#else
    const bool tracing = (p.flags & AUM_DBG_TRACE) && wg == 0;
    uint32_t* trace = (uint32_t*)(ws + L.trace);
#define SCANS_STAMP(w, r, slot)                                                                                              \
    do {                                                                                                                      \
        if (tracing && (r) < SCANS_TRACE_ITERS && (threadIdx.x & 63) == 0)                                                    \
            trace[((w) * SCANS_TRACE_ITERS + (r)) * SCANS_TRACE_SLOTS + (slot)] = (uint32_t)__builtin_readcyclecounter();    \
    } while (0)
#endif
    for (int r = 0; r <= R; ++r) {
        AUM_FOR_EACH_WAVE(w, NW) {
            ScansWave<T>& S = st[AUM_W(w)];
            SCANS_STAMP(w, r, 0);
            if (w >= 4 && w < 8) {
                if (r >= 1) p3_slice(w - 4, r - 1);
            } else if (w == 9) {
                if (r >= 1) p3_tail(S, r - 1);
            }
            SCANS_STAMP(w, r, 1);
            if (r < R) p2_state(S, w, r);
            SCANS_STAMP(w, r, 2);
            if (w < 4) {
                if (r + 1 < R) p1_slice(S, w, r + 1);
            } else if (w == 8) {
                if (r + 1 < R) p1_tail(S, r + 1);
            }
            SCANS_STAMP(w, r, 3);
            // the fetch must not be scheduled above the last use of the registers it refills
            AUM_SCHED_FENCE();
            fetch_raw(S, w, r + 2);
            AUM_SCHED_FENCE();
            SCANS_STAMP(w, r, 4);
        }
        // LDS-only barriers: everything the waves exchange per row lives in LDS, and __syncthreads() would also drain vmcnt --
        // i.e. wait at every barrier for the prefetches issued this iteration, putting the HBM latency back on the critical path
        AUM_WG_BARRIER_LDS();      // P3(r-1) has read the shares of row r-1; P1(r+1) is complete
        AUM_FOR_EACH_WAVE(w, NW) {
            SCANS_STAMP(w, r, 5);
            if (r < R) p2_publish(st[AUM_W(w)], w);
            SCANS_STAMP(w, r, 6);
        }
        AUM_WG_BARRIER_LDS();      // the shares of row r are visible
        AUM_FOR_EACH_WAVE(w, NW) { SCANS_STAMP(w, r, 7); }
    }
    // ---- dB / dC of this workgroup: one partial row per state, the tail column from the P3-tail wave ----
    AUM_FOR_EACH_WAVE(w, NW) {
        ScansWave<T>& S = st[AUM_W(w)];
        const vi lane = lane_id();
        float* dBp = ws + L.pB + ((int64_t)g_idx * p.batch + b) * N * p.len;
        float* dCp = ws + L.pC + ((int64_t)g_idx * p.batch + b) * N * p.len;
        if (w < N) {
            vf v[8];
            AUM_UNROLL
            for (int i = 0; i < 4; ++i) { v[i] = lo2(S.dBacc[i]); v[4 + i] = hi2(S.dBacc[i]); }
            gstore8(dBp + (int64_t)w * p.len, lane * 8, v, lane >= 0);
            AUM_UNROLL
            for (int i = 0; i < 4; ++i) { v[i] = lo2(S.dCacc[i]); v[4 + i] = hi2(S.dCacc[i]); }
            gstore8(dCp + (int64_t)w * p.len, lane * 8, v, lane >= 0);
        }
        if (w == 9) {
            const vi tlc = vmin_i(lane, SCANS_TC - 1);
            vf sB = vsel(lane < SCANS_TC, lds_read(tconst + 2 * SCANS_TC, tlc), splat(0.f));
            vf sC = vsel(lane < SCANS_TC, lds_read(tconst + 3 * SCANS_TC, tlc), splat(0.f));
            if (BI) {
                sB = sB + lane_gather(sB, (lane + 16) & (WAVE - 1));
                sC = sC + lane_gather(sC, (lane + 16) & (WAVE - 1));
            }
            const vi tn = vmin_i(lane, N - 1);
            gstore(dBp, tn * p.len + (SCANR_LEN - 1), sB, lane < N);
            gstore(dCp, tn * p.len + (SCANR_LEN - 1), sC, lane < N);
        }
    }
}

}  // namespace aum
