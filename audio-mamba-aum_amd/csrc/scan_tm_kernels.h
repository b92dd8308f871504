// scan_tm_kernels.h -- the selective scan on TOKEN-MAJOR activations, time-serial (round 3; include/aum_hip.h "ABI 7").
//
// Division of the work (the opposite of scan_row_kernels.h / scan_half_kernels.h, where lanes run along time):
//   * a wavefront owns 64 CHANNELS of one batch entry; lane = channel.  With activations stored [token][channel] (what F.linear
//     writes) the 64 values a wave needs at one step are one 128-byte line: every load and store is coalesced, nothing is staged
//     through LDS, and 513 is just a loop count.
//   * time is serial inside the lane: the 16 states of the channel live in 16 registers and are updated step by step,
//         a = exp2(delta_t * A_n * log2e);  x_n = a * x_n + (delta_t u_t) B_n,t;  y_t += C_n,t * x_n
//     -- 4 multiply-adds and one v_exp_f32 per (state, step): no first pass / wave scan / second pass, no product chains.  The
//     16 independent chains per lane give the instruction-level parallelism that three resident waves per SIMD then interleave.
//   * memory: a wave moves a block of 8 steps x 64 channels per tensor with ONE 16-byte-per-lane access (buffer descriptor + per-lane
//     offset + scalar row cursor: no 64-bit address arithmetic) and keeps it in an LDS strip of its own (no barrier: nobody else reads
//     it); a step takes its channel's element out of the tile.  The next block's six tensors are in flight while a block computes.
//   * B_t and C_t are the same for all 64 channels: parked as fp32 [step][B | C] in the strip, a step reads its 32 values back as
//     eight broadcast ds_read_b128 -- no scalar-cache traffic, no SGPR rows.
//   * Fo-Bi (A_b given): the forward-time and the reverse-time recurrence of a channel group are the two waves of one workgroup.
//     Each runs its first half of the sequence writing its partial y into `out`, they meet at one barrier, and each then runs its
//     second half adding the other's partial, the 2 D u skip and the gate: out_z is written once, out_pre once, and the only
//     extra traffic is one read of the partials.  3072 waves at B = 64, E = 1536: exactly three per SIMD, one round.
//   * backward: the same walk downward in blocks of 8 steps from the state checkpoint the forward leaves every 8 steps.
// Reference: SSI:37, 62-65, 499-507, 541-561 (selective_scan_cuda.fwd/.bwd call sites), SSI:86-152 (selective_scan_ref).
#pragma once
#include "scan_kernels.h"

namespace aum {

// timing experiments only (tools/build_variant.sh -DAUM_SCANT_ABL=<bits>; never set in the product build): 1 no v_exp_f32, 2 no B/C
// row loads, 4 no stores, 8 no activation loads, 16 no per-step LDS reads, 32 no per-step LDS writes, 64 no state checkpoints,
// 128 a checkpoint every second block only (= the forward's side of a 16-step spacing)
#ifndef AUM_SCANT_ABL
#define AUM_SCANT_ABL 0
#endif
#ifndef AUM_SCANT_FWD_PAIRS
#define AUM_SCANT_FWD_PAIRS 1      // 0 (A/B builds): the forward's blocks step by step, as until round 5
#endif
constexpr int SCANT_N = 16;                    // states per channel (Mamba's d_state; the only instantiation)
constexpr int SCANT_CK = AUM_SCAN_TM_CK;       // steps per checkpoint block
constexpr int SCANT_G = SCANT_CK / 2;          // steps per prefetch group (two groups in flight, ping-pong)
AUM_HOSTDEV constexpr int scant_nblocks(int len) { return (len + SCANT_CK - 1) / SCANT_CK; }
// (the last block's exit state is never needed; + SCANT_CK - 1: a direction whose iterations are numbered from scant_grid_shift() instead
// of 0 can have one block more)
AUM_HOSTDEV constexpr int scant_nck(int len) { return scant_nblocks(len + SCANT_CK - 1) - 1; }
// Rows of one checkpoint (dwords per channel).  fp32 activations: the 16 states as they are.  16-bit activations: 8 rows of PAIRS in the
// activations' own type (states 2j, 2j+1 in one dword) -- the training forward writes 0.8 GB of checkpoints per launch at the bench shape and was HBM-bound on it
// (1.83 GB in 0.39 ms = 4.7 TB/s); rounding the entry state of an 8-step block to the precision its inputs already have halves that
// traffic on both sides.  The backward walks the states in pairs, so a pass fetches exactly one row.
#ifdef AUM_SCANT_CK_F32      // A/B builds only (tools/build_variant.sh): fp32 checkpoints for every dtype, the layout before the packed form
template <class T> AUM_HOSTDEV constexpr int scant_ck_rows() { return SCANT_N; }
#else
template <class T> AUM_HOSTDEV constexpr int scant_ck_rows() { return sizeof(T) == 2 ? SCANT_N / 2 : SCANT_N; }
#endif
AUM_HOSTDEV bool scant_supported(int dim, int dstate) { return dstate == SCANT_N && dim % WAVE == 0; }

// ------------------------------------------------------------------------------------------------
// Block staging.  A wave moves the 8 steps x 64 channels of a block between HBM and a 1 KB (16-bit) / 2 KB (fp32) LDS tile as
// 16 bytes per lane (lane = (row r of the block in memory order, 16-byte chunk c of the row): one buffer_load/store_dwordx4 per
// tensor and block instead of a 2-byte access per tensor and STEP), and reads / writes its own channel's element of one step
// with a 2- or 4-byte LDS access.  Row i of a tile is iteration base + i of the block: memory row r is i = r (forward time) or
// 7 - r (reverse time), so per-lane global offsets are never negative.
// ------------------------------------------------------------------------------------------------
template <bool V> struct ScanTTag { static constexpr bool value = V; };
template <class T> struct ScanTTile {
    static constexpr int ES = (int)sizeof(T);
    static constexpr int ROWB = WAVE * ES;                 // bytes per tile row
    static constexpr int NLD = ES / 2;                     // 16-byte accesses per lane and tile (8 rows x ROWB / 1024)
    static constexpr int FLOATS = SCANT_CK * ROWB / 4;
};
constexpr int SCANT_BC_ROW = 2 * SCANT_N;                          // floats per step: B_0..B_15 | C_0..C_15
constexpr int SCANT_BC_BLOCK = SCANT_CK * SCANT_BC_ROW;            // floats per block
// LDS strip of one wave: four input tiles (u, delta, z, partial), two output tiles (out, out_pre), two B/C blocks
template <class T> AUM_HOSTDEV constexpr int scant_lds_wave_floats() { return 6 * ScanTTile<T>::FLOATS + 2 * SCANT_BC_BLOCK; }

template <class T> struct ScanTStage { vq q[ScanTTile<T>::NLD]; };
struct ScanTRaw { vi u, d, z, part; };       // one step's inputs as read from the tiles (widened where they are used)

// One direction of one channel group over scan-order iterations [it0, it1) of the row; iteration `it` is step
// t = t0 + it * tstep (tstep = +1 forward time, -1 reverse time).
// PHASE 0: the whole direction alone -- out = gate * (y + D u), out_pre = y + D u.
// PHASE 1: first half of a direction pair -- out <- partial y (no skip, no gate).
// PHASE 2: second half -- tot = y + partial(out) + dmul D u;  out_pre <- tot;  out <- gate * tot.
// PHASE 3 / 4 (time segments, scant_seg_*): carry passes that write nothing per step and keep, next to the state, the product of
//   the decays a over the range in pacc (out only: exp2(A2 * the sum of the range's deltas)).  3: the state recurrence itself, x = a x + (delta u) B.  4: the ADJOINT recurrence of the
//   backward, h = a (h + dy C) with dy = dout * gate(z), walked in the order the backward walks -- the caller passes dout as `u`, C
//   as `B` and the direction's time mapping reversed.
// Blocks of SCANT_CK steps aligned to multiples of SCANT_CK in `it`.  At the top of a block the wave requests the NEXT block's
// inputs (u, delta, z, partial, B, C: six 16-byte-per-lane loads) and holds them in registers while it computes the current block
// out of LDS; at the end of the block it flushes the output tiles (two 16-byte-per-lane stores) and parks the new inputs in the
// tiles.  Inside a block the next step's B and inputs are read from LDS while the current step computes.
// (First design: B_t / C_t through the scalar cache and 2-byte per-lane loads per step and tensor.  s_waitcnt can only wait for ALL
// outstanding scalar loads and the row latency exceeds a step; a 2-byte access per lane is one 128-byte request per instruction
// and the memory pipeline, not the vector ALU, set the pace -- 0.37 ms, 45-60 % VALU busy.)
// Blocks inside the phase take the fast form (row offsets from one scalar cursor); the ragged blocks at the ends of a phase clamp
// their rows per lane and mask their stores.
template <class T, int N, int PHASE, bool SP, bool HAS_Z, bool HAS_PRE>
AUM_DEV void scant_fwd_run(const AumScanTmFwdArgs& p, int b, int e0, int dir, int t0, int tstep, int it0, int it1, const float* Aptr,
                           float dmul, vf2 (&x)[N / 2], float* lds, unsigned long long* scant_trace_acc = nullptr, vf2* pacc = nullptr) {
    using TL = ScanTTile<T>;
    constexpr int ES = TL::ES, ROWB = TL::ROWB, NLD = TL::NLD;
    static_assert(N == 16 && SCANT_CK == 8, "lane = (row, chunk) staging below assumes 8 steps x 8 chunks / state pairs");
    constexpr bool CARRY = PHASE >= 3;
    constexpr bool LD_Z = HAS_Z && PHASE != 1 && PHASE != 3, LD_PART = PHASE == 2, ST_PRE = HAS_PRE && PHASE != 1 && !CARRY;
    const int L = p.len;
    const vi lane = lane_id();
    const vi ec = lane + e0;                     // dim % 64 == 0: every lane is a channel
    vf2 A2[N / 2];                               // A * log2(e), states (2j, 2j+1)
    AUM_UNROLL
    for (int j = 0; j < N / 2; ++j) A2[j] = mk2(gload_u(Aptr, ec * N + 2 * j) * LOG2E, gload_u(Aptr, ec * N + 2 * j + 1) * LOG2E);
    const vf biasv = p.delta_bias ? gload_u(p.delta_bias, ec) : splat(0.f);
    const vf Dv = p.D ? gload_u(p.D, ec) * dmul : splat(0.f);
    const gbuf<T> ubuf = make_gbuf(row_ptr<T>(p.u, (int64_t)b * p.u_bs));
    const gbuf<T> dbuf = make_gbuf(row_ptr<T>(p.delta, (int64_t)b * p.delta_bs));
    const gbuf<T> zbuf = make_gbuf(HAS_Z ? row_ptr<T>(p.z, (int64_t)b * p.z_bs) : row_ptr<T>(p.u, 0));
    const gbuf<T> obuf = make_gbuf(row_ptr<T>(p.out, (int64_t)b * p.out_bs));
    const gbuf<T> pbuf = make_gbuf(HAS_PRE ? row_ptr<T>(p.out_pre, (int64_t)b * p.pre_bs) : row_ptr<T>(p.out, 0));
    const gbuf<T> Bbuf = make_gbuf(row_ptr<T>(p.B, (int64_t)b * p.B_bs));
    const gbuf<T> Cbuf = make_gbuf(row_ptr<T>(p.C, (int64_t)b * p.C_bs));
    const int nck = scant_nck(L);
    const bool want_ck = p.ckpt != nullptr;
    constexpr int CKR = scant_ck_rows<T>();
    const gbuf<float> ckbuf = make_gbuf(want_ck ? p.ckpt + ((int64_t)dir * p.batch + b) * nck * CKR * p.dim : (const float*)Aptr);
    // byte strides per step of time
    const int u_tb = (int)p.u_ts * ES, d_tb = (int)p.delta_ts * ES, z_tb = HAS_Z ? (int)p.z_ts * ES : 0, o_tb = (int)p.out_ts * ES,
              p_tb = HAS_PRE ? (int)p.pre_ts * ES : 0, B_tb = (int)p.B_ts * ES, C_tb = (int)p.C_ts * ES;
    auto tok = [&](int it) { return t0 + it * tstep; };
    // LDS strip
    float* t_u = lds;
    float* t_d = lds + TL::FLOATS;
    float* t_z = lds + 2 * TL::FLOATS;
    float* t_p = lds + 3 * TL::FLOATS;
    float* t_o = lds + 4 * TL::FLOATS;
    float* t_pre = lds + 5 * TL::FLOATS;
    float* t_bc = lds + 6 * TL::FLOATS;
    // staging lanes: memory row r, 16-byte chunk c (+ 8 for the second access of fp32 tiles); iteration index i of the row
    const vi st_r = lane >> 3, st_c = lane & 7;
    const vi st_i = tstep > 0 ? st_r : (SCANT_CK - 1) - st_r;
    const vi st_gcol = st_c * 16 + e0 * ES;                 // byte offset of the chunk in a global row
    const vi st_lds = st_i * ROWB + st_c * 16;              // byte offset of the chunk in a tile
    const vi el_off = lane * ES;                            // this lane's element inside a tile row
    const vi bc_slot = st_i * SCANT_BC_ROW + st_c * 2;      // B pair st_c of the row's step in a B/C block; the C pair at + N
    const vi vo4 = ec * 4;

    struct Next { ScanTStage<T> u, d, z, part; vpair_raw bp, cp; };
    // request block `blk` (rows clamped into the phase unless the whole block lies inside it)
    auto request = [&](int blk, Next& n) {
        const int base = blk * SCANT_CK;
        const bool inside = base >= it0 && base + SCANT_CK <= it1;
        vi rowt;                 // time step of this lane's row (ragged blocks)
        int t_lo = 0;            // lowest time step of the block (blocks inside the phase)
        if (inside) {
            t_lo = tstep > 0 ? tok(base) : tok(base + SCANT_CK - 1);
            rowt = st_r;
        } else {
            const vi it = vmax_i(vmin_i(st_i + base, it1 - 1), it0);
            rowt = it * tstep + t0;
        }
        AUM_UNROLL
        for (int i = 0; i < NLD; ++i) {
            const vi col = st_gcol + 128 * i;
            if (!(AUM_SCANT_ABL & 8)) {
                n.u.q[i] = gbuf_load16(ubuf, rowt * u_tb + col, t_lo * u_tb);
                n.d.q[i] = gbuf_load16(dbuf, rowt * d_tb + col, t_lo * d_tb);
                if (LD_Z) n.z.q[i] = gbuf_load16(zbuf, rowt * z_tb + col, t_lo * z_tb);
                if (LD_PART) n.part.q[i] = gbuf_load16(obuf, rowt * o_tb + col, t_lo * o_tb);
            } else {
                AUM_UNROLL
                for (int k = 0; k < 4; ++k) n.u.q[i].w[k] = n.d.q[i].w[k] = n.z.q[i].w[k] = n.part.q[i].w[k] = spl_i(0x3c003c00);
            }
        }
        n.bp = gbuf_load_pair_raw(Bbuf, rowt * B_tb + st_c * (2 * ES), t_lo * B_tb);
        n.cp = gbuf_load_pair_raw(Cbuf, rowt * C_tb + st_c * (2 * ES), t_lo * C_tb);
    };
    // park a requested block in the tiles / the B/C block of its parity
    auto park = [&](int blk, const Next& n) {
        AUM_UNROLL
        for (int i = 0; i < NLD; ++i) {
            const vi off = st_lds + 128 * i;
            lds_write16(t_u, off, n.u.q[i]);
            lds_write16(t_d, off, n.d.q[i]);
            if (LD_Z) lds_write16(t_z, off, n.z.q[i]);
            if (LD_PART) lds_write16(t_p, off, n.part.q[i]);
        }
        float* bc = t_bc + (blk & 1) * SCANT_BC_BLOCK;
        vf b0, b1, c0, c1;
        pair_raw_to_f32<T>(n.bp, b0, b1);
        pair_raw_to_f32<T>(n.cp, c0, c1);
        if (AUM_SCANT_ABL & 2) b0 = b1 = c0 = c1 = splat(1.f);
        lds_write2(bc, bc_slot, b0, b1);
        lds_write2(bc, bc_slot + N, c0, c1);
        wave_lds_fence();
    };
    // write the output tiles of block `blk` back (rows outside the phase are not written)
    auto flush = [&](int blk) {
        if ((AUM_SCANT_ABL & 4) || CARRY) return;
        const int base = blk * SCANT_CK;
        const bool inside = base >= it0 && base + SCANT_CK <= it1;
        wave_lds_fence();
        vi rowt;
        int t_lo = 0;
        vm valid = lane >= 0;
        if (inside) {
            t_lo = tstep > 0 ? tok(base) : tok(base + SCANT_CK - 1);
            rowt = st_r;
        } else {
            const vi it = st_i + base;
            valid = (it >= it0) && (it < it1);
            rowt = vmax_i(vmin_i(it, it1 - 1), it0) * tstep + t0;
        }
        AUM_UNROLL
        for (int i = 0; i < NLD; ++i) {
            const vi off = st_lds + 128 * i, col = st_gcol + 128 * i;
            const vq qo = lds_read16(t_o, off);
            if (inside) {
                gbuf_store16(obuf, rowt * o_tb + col, t_lo * o_tb, qo);
                if (ST_PRE) gbuf_store16(pbuf, rowt * p_tb + col, t_lo * p_tb, lds_read16(t_pre, off));
            } else if (any_lane(valid)) {
                vq qp = qo;
                if (ST_PRE) qp = lds_read16(t_pre, off);
                gbuf_store16_m(obuf, rowt * o_tb + col, 0, qo, valid);
                if (ST_PRE) gbuf_store16_m(pbuf, rowt * p_tb + col, 0, qp, valid);
            }
        }
    };
    auto read_B = [&](const float* bcrow, vf2 (&Bp)[N / 2]) {
        AUM_UNROLL
        for (int k = 0; k < N / 4; ++k) {
            vf q[4];
            if (AUM_SCANT_ABL & 16) q[0] = q[1] = q[2] = q[3] = splat(1.f);
            else lds_read4_u(bcrow, 4 * k, q);
            Bp[2 * k] = mk2(q[0], q[1]);
            Bp[2 * k + 1] = mk2(q[2], q[3]);
        }
    };
    auto read_raw = [&](int s, ScanTRaw& r) {
        const vi off = el_off + s * ROWB;
        if (AUM_SCANT_ABL & 16) {
            r.u = r.d = r.z = r.part = lane + s;
            return;
        }
        r.u = lds_read_raw<T>(t_u, off);
        r.d = lds_read_raw<T>(t_d, off);
        if (LD_Z) r.z = lds_read_raw<T>(t_z, off);
        if (LD_PART) r.part = lds_read_raw<T>(t_p, off);
    };
    auto delta_of = [&](const ScanTRaw& r) {
        vf dl = raw_to_f32<T>(r.d) + biasv;
        if (SP) dl = vsoftplus(dl);
        return dl;
    };
    // step s of the current block.  r / dl / Bp: this step's inputs, its delta and its B (read from LDS / computed during the previous
    // step); rn / dln / Bn <- the next step's: requested here together with this step's C, ahead of the arithmetic that hides the round
    // trips, and the next delta (the softplus chain) after this step's state updates.
    // The arithmetic is written stage by stage over the eight state pairs -- all exponents, then all sixteen v_exp_f32, then the
    // updates -- because the compiler keeps source order: pair by pair, every v_pk_fma waited for the v_exp right in front of it.
    vf dsum = splat(0.f);       // carry passes: the sum of this range's deltas
    auto step = [&](int s, const float* bc, bool prefetch, const ScanTRaw& r, vf dl, vf2 (&Bp)[N / 2], ScanTRaw& rn, vf& dln, vf2 (&Bn)[N / 2]) {
        const float* bcrow = bc + s * SCANT_BC_ROW;
        vf2 Cp[N / 2];
        if (!CARRY) {
            AUM_UNROLL
            for (int k = 0; k < N / 4; ++k) {
                vf q[4];
                if (AUM_SCANT_ABL & 16) q[0] = q[1] = q[2] = q[3] = splat(1.f);
                else lds_read4_u(bcrow, N + 4 * k, q);
                Cp[2 * k] = mk2(q[0], q[1]);
                Cp[2 * k + 1] = mk2(q[2], q[3]);
            }
        }
        if (prefetch) {
            read_B(bcrow + SCANT_BC_ROW, Bn);
            read_raw(s + 1, rn);
        }
        AUM_SCHED_FENCE();
        const vf uu = raw_to_f32<T>(r.u);
        const vf du = dl * uu;
        const vf2 dl2 = spl2(dl), du2 = spl2(du);
        vf2 a[N / 2];
        AUM_UNROLL
        for (int j = 0; j < N / 2; ++j) a[j] = dl2 * A2[j];
        if (CARRY) {
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) a[j] = vexp2_2(a[j]);
            if (PHASE == 3) {
                AUM_UNROLL
                for (int j = 0; j < N / 2; ++j) x[j] = vfma2(a[j], x[j], du2 * Bp[j]);
            } else {        // uu = dout of the step, Bp = its C row
                vf dy = uu;
                if (LD_Z) {
                    const vf zz = raw_to_f32<T>(r.z);
                    dy = uu * (zz * vsigmoid(zz));
                }
                const vf2 dy2 = spl2(dy);
                AUM_UNROLL
                for (int j = 0; j < N / 2; ++j) x[j] = a[j] * vfma2(dy2, Bp[j], x[j]);
            }
            dsum = dsum + dl;          // the product of the decays is exp2(A2 * sum of delta): formed once, after the last block
            if (prefetch) dln = delta_of(rn);
            return;
        }
        vf zz = splat(0.f), ez = splat(0.f);
        if (LD_Z) {
            zz = raw_to_f32<T>(r.z);
            ez = zz * (-LOG2E);
        }
        AUM_UNROLL
        for (int j = 0; j < N / 2; ++j) Bp[j] = du2 * Bp[j];
        if (!(AUM_SCANT_ABL & 1)) {
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) a[j] = vexp2_2(a[j]);
        }
        if (LD_Z) ez = vexp2(ez);
        AUM_UNROLL
        for (int j = 0; j < N / 2; ++j) x[j] = vfma2(a[j], x[j], Bp[j]);
        vf sg = splat(1.f);
        if (LD_Z) sg = vrcp(ez + 1.0f);
        vf2 y2[4];
        AUM_UNROLL
        for (int j = 0; j < 4; ++j) y2[j] = x[j] * Cp[j];
        AUM_UNROLL
        for (int j = 4; j < N / 2; ++j) y2[j & 3] = vfma2(x[j], Cp[j], y2[j & 3]);
        if (prefetch) dln = delta_of(rn);
        const vf2 ysum = (y2[0] + y2[1]) + (y2[2] + y2[3]);
        const vf ys = lo2(ysum) + hi2(ysum);
        const vi off = el_off + s * ROWB;
        if (PHASE == 1) {
            if (!(AUM_SCANT_ABL & 32)) lds_write_elem<T>(t_o, off, ys);
            else x[0] = x[0] + spl2(ys * 1e-30f);
            return;
        }
        vf tot = vfma(uu, Dv, ys);
        if (PHASE == 2) tot = tot + raw_to_f32<T>(r.part);
        if (ST_PRE && !(AUM_SCANT_ABL & 32)) lds_write_elem<T>(t_pre, off, tot);
        if (HAS_Z) tot = tot * (zz * sg);
        if (!(AUM_SCANT_ABL & 32)) lds_write_elem<T>(t_o, off, tot);
        else x[0] = x[0] + spl2(tot * 1e-30f);
    };
    // Four steps [s0, s0 + 4) of a block that lies inside the phase (round 6): the per-step values -- unpack, softplus, delta u, the gate -- are
    // formed for PAIRS of steps on packed fp32 ahead of the recurrence (the same IEEE operations in the same order as `step`, bit-equal), and a
    // pair's outputs (D u skip, partial, gate) leave together behind its second step.  Bq: B of step s0 on entry, of step s0 + 4 on exit
    // (more_b).  (r04_isa_mix_tm.txt: 28 of a step's ~80 vector-ALU instructions were this per-step work, one step at a time.)
    auto steps4 = [&](int s0, const float* bc, vf2 (&Bq)[N / 2], bool more_b) {
        vi ru[4], rd[4], rz[4], rp[4];
        AUM_UNROLL
        for (int q = 0; q < 4; ++q) {
            const vi off = el_off + (s0 + q) * ROWB;
            ru[q] = lds_read_raw<T>(t_u, off);
            rd[q] = lds_read_raw<T>(t_d, off);
            if (LD_Z) rz[q] = lds_read_raw<T>(t_z, off);
            if (LD_PART) rp[q] = lds_read_raw<T>(t_p, off);
        }
        vf2 Pd[2], Pu[2], Us[2], G[2];
        const vf2 bias2 = spl2(biasv), Dv2 = spl2(Dv);
        AUM_UNROLL
        for (int i = 0; i < 2; ++i) {
            Us[i] = mk2(raw_to_f32<T>(ru[2 * i]), raw_to_f32<T>(ru[2 * i + 1]));
            vf2 d = mk2(raw_to_f32<T>(rd[2 * i]), raw_to_f32<T>(rd[2 * i + 1])) + bias2;
            if (SP) d = vsoftplus2(d);
            Pd[i] = d;
            Pu[i] = d * Us[i];
            if (LD_Z) {
                const vf2 zz = mk2(raw_to_f32<T>(rz[2 * i]), raw_to_f32<T>(rz[2 * i + 1]));
                G[i] = zz * vsigmoid2(zz);
            }
        }
        vf ys_even = splat(0.f);
        AUM_UNROLL
        for (int q = 0; q < 4; ++q) {
            const int s = s0 + q;
            const float* bcrow = bc + s * SCANT_BC_ROW;
            vf2 Cp[N / 2], Bn[N / 2];
            AUM_UNROLL
            for (int k = 0; k < N / 4; ++k) {
                vf c4[4];
                lds_read4_u(bcrow, N + 4 * k, c4);
                Cp[2 * k] = mk2(c4[0], c4[1]);
                Cp[2 * k + 1] = mk2(c4[2], c4[3]);
            }
            const bool pf = q < 3 || more_b;
            if (pf) read_B(bcrow + SCANT_BC_ROW, Bn);
            AUM_SCHED_FENCE();
            const vf2 dl2 = (q & 1) ? bc_hi(Pd[q >> 1]) : bc_lo(Pd[q >> 1]);
            const vf2 du2 = (q & 1) ? bc_hi(Pu[q >> 1]) : bc_lo(Pu[q >> 1]);
            vf2 a[N / 2];
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) a[j] = dl2 * A2[j];
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) Bq[j] = du2 * Bq[j];
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) a[j] = vexp2_2(a[j]);
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) x[j] = vfma2(a[j], x[j], Bq[j]);
            vf2 y2[4];
            AUM_UNROLL
            for (int j = 0; j < 4; ++j) y2[j] = x[j] * Cp[j];
            AUM_UNROLL
            for (int j = 4; j < N / 2; ++j) y2[j & 3] = vfma2(x[j], Cp[j], y2[j & 3]);
            const vf2 ysum = (y2[0] + y2[1]) + (y2[2] + y2[3]);
            const vf ys = lo2(ysum) + hi2(ysum);
            if (PHASE == 1) {
                lds_write_elem<T>(t_o, el_off + s * ROWB, ys);
            } else if ((q & 1) == 0) {
                ys_even = ys;
            } else {
                const int i = q >> 1;
                const vi off0 = el_off + (s - 1) * ROWB, off1 = el_off + s * ROWB;
                vf2 tot = vfma2(Us[i], Dv2, mk2(ys_even, ys));
                if (PHASE == 2) tot = tot + mk2(raw_to_f32<T>(rp[q - 1]), raw_to_f32<T>(rp[q]));
                if (ST_PRE) {
                    lds_write_elem<T>(t_pre, off0, lo2(tot));
                    lds_write_elem<T>(t_pre, off1, hi2(tot));
                }
                if (HAS_Z) tot = tot * G[i];
                lds_write_elem<T>(t_o, off0, lo2(tot));
                lds_write_elem<T>(t_o, off1, hi2(tot));
            }
            if (pf) {
                AUM_UNROLL
                for (int j = 0; j < N / 2; ++j) Bq[j] = Bn[j];
            }
        }
    };
    auto ckpt_store = [&](int blk) {
        int off = blk * CKR * p.dim * 4;
        if constexpr (CKR == N) {
            AUM_UNROLL
            for (int n = 0; n < N; ++n) {
                gbuf_store(ckbuf, vo4, off, (n & 1) ? hi2(x[n >> 1]) : lo2(x[n >> 1]));
                off += p.dim * 4;
            }
        } else {
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) {
                gbuf_store_pair16<T>(ckbuf, vo4, off, lo2(x[j]), hi2(x[j]));
                off += p.dim * 4;
            }
        }
    };

    if (it0 >= it1) return;
    const int blk0 = it0 / SCANT_CK, blk1 = (it1 + SCANT_CK - 1) / SCANT_CK;
#if defined(AUM_SCANT_TRACE) && !defined(AUM_EMU)
#define AUM_TM_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); scant_trace_acc[k] += now_ - stamp_; stamp_ = now_; } while (0)
    unsigned long long stamp_ = __builtin_readcyclecounter();
#else
#define AUM_TM_STAMP(k) do { } while (0)
#endif
    Next nx;
    request(blk0, nx);
    park(blk0, nx);
    AUM_TM_STAMP(0);
#ifndef AUM_SCANT_PRIO
#define AUM_SCANT_PRIO 1
#endif
#ifndef AUM_SCANT_PRIO_SHIFT    // the turns change every 2^SHIFT blocks (A/B builds)
#define AUM_SCANT_PRIO_SHIFT 0
#endif
#ifndef AUM_SCANT_PRIO_A        // the levels of the three turns (A/B builds)
#define AUM_SCANT_PRIO_A 2
#endif
#ifndef AUM_SCANT_PRIO_B
#define AUM_SCANT_PRIO_B 1
#endif
    // Three waves share a SIMD and the arbiter serves equal priorities oldest first: left alone, one wave runs at full speed and
    // finishes at 0.55 of the kernel's time, the second at 0.75, and the last runs the final quarter alone at less than half the
    // vector ALU's rate (measured: wave durations 185 / 254 / 325 us on every SIMD).  Each wave therefore walks through the
    // priorities 2, 1, 0 block by block, staggered by its slot number on the SIMD, so the three take turns and finish together.
    const int wslot = AUM_SCANT_PRIO ? wave_slot_on_simd() : 0;
    for (int blk = blk0; blk < blk1; ++blk) {
        const int base = blk * SCANT_CK;
        const bool more = blk + 1 < blk1;
        if (AUM_SCANT_PRIO) {
            const int turn = ((blk >> AUM_SCANT_PRIO_SHIFT) + wslot) % 3;
            if (turn == 0) AUM_SET_PRIO(AUM_SCANT_PRIO_A);
            else if (turn == 1) AUM_SET_PRIO(AUM_SCANT_PRIO_B);
            else AUM_SET_PRIO(0);
        }
        if (more) request(blk + 1, nx);
        AUM_TM_STAMP(1);
        const float* bc = t_bc + (blk & 1) * SCANT_BC_BLOCK;
        if (base >= it0 && base + SCANT_CK <= it1 && !CARRY && !AUM_SCANT_ABL && AUM_SCANT_FWD_PAIRS) {
            vf2 Bq[N / 2];
            read_B(bc, Bq);
            steps4(0, bc, Bq, true);
            steps4(4, bc, Bq, false);
            if (want_ck && blk < nck) ckpt_store(blk);
        } else if (base >= it0 && base + SCANT_CK <= it1) {
            ScanTRaw r;
            vf2 Bq[N / 2];
            read_raw(0, r);
            read_B(bc, Bq);
            vf dl = delta_of(r);
            AUM_UNROLL
            for (int s = 0; s < SCANT_CK; ++s) {
                ScanTRaw rn;
                vf2 Bn[N / 2];
                vf dln;
                step(s, bc, s + 1 < SCANT_CK, r, dl, Bq, rn, dln, Bn);
                if (s + 1 < SCANT_CK) {
                    r = rn;
                    dl = dln;
                    AUM_UNROLL
                    for (int j = 0; j < N / 2; ++j) Bq[j] = Bn[j];
                }
            }
            if (want_ck && !CARRY && blk < nck && !(AUM_SCANT_ABL & 64) && !((AUM_SCANT_ABL & 128) && (blk & 1) == 0)) ckpt_store(blk);
        } else {        // ragged: steps outside the phase are skipped
            for (int s = 0; s < SCANT_CK; ++s) {
                if (base + s < it0 || base + s >= it1) continue;
                ScanTRaw r, rn;
                vf2 Bq[N / 2], Bn[N / 2];
                vf dln;
                read_raw(s, r);
                read_B(bc + s * SCANT_BC_ROW, Bq);
                step(s, bc, false, r, delta_of(r), Bq, rn, dln, Bn);
            }
            // the block's exit state is complete only in the phase that ran its last step
            if (want_ck && !CARRY && blk < nck && base + SCANT_CK - 1 >= it0 && base + SCANT_CK - 1 < it1) ckpt_store(blk);
        }
        AUM_TM_STAMP(2);
        flush(blk);
        AUM_TM_STAMP(3);
        if (more) park(blk + 1, nx);
        AUM_TM_STAMP(4);
    }
#undef AUM_TM_STAMP
    if (CARRY) {
        const vf2 ds2 = spl2(dsum);
        AUM_UNROLL
        for (int j = 0; j < N / 2; ++j) pacc[j] = vexp2_2(ds2 * A2[j]);
    }
}

// iterations the forward-time wave (dir 0) and the reverse-time wave (dir 1) of a pair run before they meet: together they cover
// every step exactly once
AUM_HOSTDEV constexpr int scant_first_half(int len, int dir) { return dir == 0 ? len / 2 : len - len / 2; }
// A direction pair's meeting point on a block boundary.  L = 513 is 256 + 257 iterations: numbered from 0, the reverse direction's boundary falls
// one step into its block 32, which is then a ragged block in BOTH of its phases (1 + 7 live steps) next to the ragged last block -- and a
// ragged block of the backward costs 1.25 whole ones (all exponentials, both butterflies, per-step conditions).  So a direction's iterations
// are NUMBERED from io = (SCANT_CK - first_half % SCANT_CK) % SCANT_CK instead of 0 (the callers pass it0 + io, it1 + io and the time origin
// moved by io steps; the runners see ordinary ranges): the meeting point is a multiple of SCANT_CK, the direction's first block holds its
// first SCANT_CK - io steps, 2 ragged blocks per pair instead of 4.  No extra value lives in the kernels (k_scant_bwd is at 254 registers
// with two more holding spilled scalars: one more scalar and it spills vector registers, whose scratch accesses break every counted vmcnt
// wait -- profiles/r06_scan_grid_shift.txt; csrc/build.py refuses such a build).
#ifndef AUM_SCANT_GRID_SHIFT
#define AUM_SCANT_GRID_SHIFT 1      // 0: A/B build -- iterations numbered from 0 (as until round 6)
#endif
AUM_HOSTDEV constexpr int scant_grid_shift(int len, int dir) {
    return AUM_SCANT_GRID_SHIFT && len >= 8 * SCANT_CK ? (SCANT_CK - scant_first_half(len, dir) % SCANT_CK) % SCANT_CK : 0;
}

// workgroup = four waves, one per SIMD: two channel groups x two directions (BIDIR: even wave = A forward time, odd wave = A_b reverse
// time of the same group) or four channel groups.  Units (batch entry, channel group) are numbered batch-major.
constexpr int SCANT_NW = 4;
template <bool BIDIR> AUM_HOSTDEV constexpr int scant_units_per_wg() { return BIDIR ? SCANT_NW / 2 : SCANT_NW; }
// workgroups of the backward: four one-direction units, or three Fo-Bi pairs in three stages (scant_bwd)
template <bool BIDIR> AUM_HOSTDEV constexpr int scant_bwd_wgs(int units) { return BIDIR ? (units + 2) / 3 : (units + SCANT_NW - 1) / SCANT_NW; }

template <class T, bool SP, bool HAS_Z, bool HAS_PRE, bool BIDIR>
AUM_DEV void scant_fwd(const AumScanTmFwdArgs& p, int wg, float* lds, unsigned long long* tacc = nullptr) {
    constexpr int N = SCANT_N;
    constexpr int NW = SCANT_NW;
    constexpr int UPW = scant_units_per_wg<BIDIR>();
    const int gpb = p.dim / WAVE;
    const int units = p.batch * gpb;
    const int L = p.len;
    vf2 x[AUM_PER_WAVE(NW)][N / 2];
    if (!BIDIR) {
        const bool rev = (p.flags & AUM_SCAN_REVERSE) != 0;
        AUM_FOR_EACH_WAVE(w, NW) {
            const int unit = wg * UPW + w;
            if (unit < units) {
                AUM_UNROLL
                for (int j = 0; j < N / 2; ++j) x[AUM_W(w)][j] = spl2(splat(0.f));
                scant_fwd_run<T, N, 0, SP, HAS_Z, HAS_PRE>(p, unit / gpb, (unit % gpb) * WAVE, 0, rev ? L - 1 : 0, rev ? -1 : 1, 0, L, p.A, 1.f,
                                                           x[AUM_W(w)], lds + w * scant_lds_wave_floats<T>(), tacc);
            }
        }
        return;
    }
    AUM_FOR_EACH_WAVE(w, NW) {
        const int unit = wg * UPW + (w >> 1), d = w & 1, io = scant_grid_shift(L, d);
        if (unit < units) {
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) x[AUM_W(w)][j] = spl2(splat(0.f));
            scant_fwd_run<T, N, 1, SP, HAS_Z, HAS_PRE>(p, unit / gpb, (unit % gpb) * WAVE, d, d ? L - 1 + io : -io, d ? -1 : 1, io, scant_first_half(L, d) + io,
                                                       d ? p.A_b : p.A, 2.f, x[AUM_W(w)], lds + w * scant_lds_wave_floats<T>(), tacc);
        }
    }
    AUM_WG_BARRIER();      // also orders this workgroup's partial stores before the other wave's loads of them (same CU, same L2)
    AUM_FOR_EACH_WAVE(w, NW) {
        const int unit = wg * UPW + (w >> 1), d = w & 1, io = scant_grid_shift(L, d);
        if (unit < units)
            scant_fwd_run<T, N, 2, SP, HAS_Z, HAS_PRE>(p, unit / gpb, (unit % gpb) * WAVE, d, d ? L - 1 + io : -io, d ? -1 : 1, scant_first_half(L, d) + io, L + io,
                                                       d ? p.A_b : p.A, 2.f, x[AUM_W(w)], lds + w * scant_lds_wave_floats<T>(), tacc);
    }
}


// ------------------------------------------------------------------------------------------------
// Time segments (round 4): long rows at a small batch.  One wave per (batch entry, channel group, direction) is 8 x 24 x 2 = 384
// waves at the long-form shape (B = 8, L = 4097) -- 0.4 per SIMD, each a 4097-step latency chain.  The recurrence is affine in the
// state, so a direction's iterations are cut into `nseg` segments of seg_len steps (a multiple of the checkpoint block) that run as
// separate waves:
//   carry pass   every segment from a zero entry state: X_s = its exit state, P_s = the product of its decays (PHASE 3);
//   main pass    segment s starts from  x = P_{s-1} ( ... (P_0 0 + X_0) ... ) + X_{s-1}  (s multiply-adds per state in the wave's
//                prologue) and runs the ordinary phases.  Fo-Bi: direction 0 writes its partial y for the whole row (PHASE 1) in
//                one launch, direction 1 finishes the row (PHASE 2) in the next -- the hand-over a barrier orders inside the
//                unsegmented kernel is ordered by the stream here.
// The backward is the same construction on the adjoint h (affine, the same decays): PHASE 4 is its carry pass.
// ------------------------------------------------------------------------------------------------
struct ScanTSeg {
    float* carry;          // [ndir][batch][nseg][2][N][dim] fp32: P rows, then X rows of a segment
    int nseg, seg_len;     // seg_len % SCANT_CK == 0
    int dir0, ndl;         // this launch runs directions dir0 .. dir0 + ndl - 1
};
AUM_HOSTDEV constexpr int scant_seg_len(int len, int nseg) { return ((len + nseg - 1) / nseg + SCANT_CK - 1) / SCANT_CK * SCANT_CK; }
AUM_HOSTDEV constexpr int64_t scant_seg_carry_floats(int batch, int dim, int nseg, bool bidir) {
    return (int64_t)(bidir ? 2 : 1) * batch * nseg * 2 * SCANT_N * dim;
}

// workgroup = four independent waves; item = ((unit * nseg) + segment) * ndl + local direction
template <class T, int PHASE, bool SP, bool HAS_Z, bool HAS_PRE>
AUM_DEV void scant_seg_fwd(const AumScanTmFwdArgs& p, const ScanTSeg& sg, int wg, float* lds) {
    constexpr int N = SCANT_N;
    constexpr int NW = SCANT_NW;
    const int gpb = p.dim / WAVE;
    const int L = p.len;
    const int items = p.batch * gpb * sg.nseg * sg.ndl;
    const bool bidir = p.A_b != nullptr;
    vf2 x[AUM_PER_WAVE(NW)][N / 2], P[AUM_PER_WAVE(NW)][N / 2];
    AUM_FOR_EACH_WAVE(w, NW) {
        const int item = wg * NW + w;
        // the carries of a direction's last segment (state) / first segment (adjoint) are never used
        const int s = item < items ? (item / sg.ndl) % sg.nseg : 0;
        const bool idle = item >= items || (PHASE == 3 && s == sg.nseg - 1) || (PHASE == 4 && s == 0);
        if (!idle) {
            const int d = sg.dir0 + item % sg.ndl, unit = item / (sg.ndl * sg.nseg);
            const int b = unit / gpb, e0 = (unit % gpb) * WAVE;
            const bool rev = bidir ? d == 1 : (p.flags & AUM_SCAN_REVERSE) != 0;
            const float* Aptr = d ? p.A_b : p.A;
            const int it0 = s * sg.seg_len < L ? s * sg.seg_len : L;
            const int it1 = it0 + sg.seg_len < L ? it0 + sg.seg_len : L;
            const vi ec = lane_id() + e0;
            float* cb = sg.carry + ((int64_t)d * p.batch + b) * sg.nseg * (2 * N) * p.dim;
            float* lw = lds + w * scant_lds_wave_floats<T>();
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) x[AUM_W(w)][j] = spl2(splat(0.f));
            if (PHASE >= 3) {
                AUM_UNROLL
                for (int j = 0; j < N / 2; ++j) P[AUM_W(w)][j] = spl2(splat(1.f));
                if (PHASE == 3)
                    scant_fwd_run<T, N, 3, SP, HAS_Z, HAS_PRE>(p, b, e0, d, rev ? L - 1 : 0, rev ? -1 : 1, it0, it1, Aptr, 1.f, x[AUM_W(w)], lw, nullptr,
                                                               P[AUM_W(w)]);
                else        // the adjoint walks the direction's iterations downward: iteration L - 1 - it of the reversed mapping
                    scant_fwd_run<T, N, 4, SP, HAS_Z, HAS_PRE>(p, b, e0, d, rev ? 0 : L - 1, rev ? 1 : -1, L - it1, L - it0, Aptr, 1.f, x[AUM_W(w)], lw,
                                                               nullptr, P[AUM_W(w)]);
                float* cs = cb + (int64_t)s * (2 * N) * p.dim;
                AUM_UNROLL
                for (int n = 0; n < N; ++n) {
                    gstore(cs + (int64_t)n * p.dim, ec, (n & 1) ? hi2(P[AUM_W(w)][n >> 1]) : lo2(P[AUM_W(w)][n >> 1]), ec >= 0);
                    gstore(cs + (int64_t)(N + n) * p.dim, ec, (n & 1) ? hi2(x[AUM_W(w)][n >> 1]) : lo2(x[AUM_W(w)][n >> 1]), ec >= 0);
                }
            } else {
                for (int sp = 0; sp < s; ++sp) {
                    const float* cs = cb + (int64_t)sp * (2 * N) * p.dim;
                    AUM_UNROLL
                    for (int j = 0; j < N / 2; ++j) {
                        const vf2 pj = mk2(gload_u(cs + (int64_t)(2 * j) * p.dim, ec), gload_u(cs + (int64_t)(2 * j + 1) * p.dim, ec));
                        const vf2 xj = mk2(gload_u(cs + (int64_t)(N + 2 * j) * p.dim, ec), gload_u(cs + (int64_t)(N + 2 * j + 1) * p.dim, ec));
                        x[AUM_W(w)][j] = vfma2(pj, x[AUM_W(w)][j], xj);
                    }
                }
                scant_fwd_run<T, N, PHASE, SP, HAS_Z, HAS_PRE>(p, b, e0, d, rev ? L - 1 : 0, rev ? -1 : 1, it0, it1, Aptr, bidir ? 2.f : 1.f,
                                                               x[AUM_W(w)], lw);
            }
        }
    }
}

// ================================================================================================
// Backward.  The same division of the work: lane = channel, a wave owns one direction of a channel group and walks its blocks of
// 8 steps in REVERSE scan order.  A block is processed in passes over state pairs (two states at a time):
//   forward sweep over the block from the checkpointed entry state:  a = exp2(delta A2), w = a x, x = w + (delta u) B -- keeps w_i
//     (= a_i x_{i-1}) in registers and forms the dC products dy_i x_i;
//   reverse sweep with the adjoint  g_i = dy_i C_i + a_{i+1} g_{i+1}  carried from block to block:  dB products g (delta u),
//     S1_i += g . B_i,  S2_i += A2 . (g w_i),  dA += delta_i g w_i;
//   the 16 + 16 products of the pass are summed over the 64 channels by two transposing butterflies (wave_sum16) and collected in
//     an LDS tile [step][dB_0..15 | dC_0..15] that leaves as ONE partial row block per wave and block (no atomics; a second
//     kernel sums the channel-group partials).
// After the passes: du_i = delta_i S1_i + D dy_i, ddelta_i = u_i S1_i + ln2 S2_i, dz_i = dout_i ytot_i d(gate)/dz.
// Fo-Bi: the two direction waves of a channel group walk time in opposite order; as in the forward each first writes its half of
// du / ddelta as partials, they meet at one barrier, and each then finishes the other's half (adding the partial, applying the
// softplus derivative, dz, dD, ddelta_bias).
// Memory: every tensor of a block arrives by global -> LDS loads that bypass the registers (buffer_load ... lds), issued a whole
// block (eight passes) ahead: the input tiles are free as soon as a block has turned them into per-step registers (delta, delta u,
// dy) before its first pass, the checkpointed entry states of the next block replace the current block's rows pair by pair as the
// passes consume them, and A sits in LDS for the whole kernel.  Loads parked through registers one pass ahead cost 0.6 ms of a
// 1.4 ms kernel: one pass is shorter than a loaded HBM round trip and two waves per SIMD cannot hide the difference.
// LDS tiles are planes of 8 rows x 128 bytes (lane l's 16 bytes of a load at l * 16), the order the direct loads write in.
// ================================================================================================
// (Round 5 built the dB / dC channel sums of a pass through a per-wave LDS transposition tile, round 4 on the matrix pipe: both parity-green,
// both measured slower than the transposing butterflies below -- profiles/r05_ab_lsum.txt, r04_ab_msum.txt, HISTORY.md.  Removed in round 6.)
template <class T> AUM_HOSTDEV constexpr int scant_bwd_lds_wave_floats() {
    // u, delta, z, dout, ypre | du, ddelta, dz | B/C block | dB/dC block | u (odd blocks) | entry states [16][64] | A [16][64] | raw B, C pairs
    // (+ fp32 activations: the other direction's dB | dC row block of the second phase; 16-bit activations park it in the half of the
    // entry-state strip their packed checkpoints leave free)
    return 9 * ScanTTile<T>::FLOATS + 2 * SCANT_BC_BLOCK + 2 * SCANT_N * WAVE + 2 * ScanTTile<T>::NLD * WAVE + (sizeof(T) == 4 ? SCANT_BC_BLOCK : 0);
}
// Fo-Bi: the two directions of a channel group hand their dB | dC partial rows over like their du / ddelta partials -- the wave that runs a
// time step in its SECOND phase adds the row block the other direction wrote there in its first -- so the reduce kernel sums one row per
// channel group instead of two (round 5: 201 -> 101 MB read by k_scant_bwd_reduce).  Slots of a token's partial rows: [0, nparts / 2)
// finished rows (second phase), [nparts / 2, nparts) first-phase partials.  Off in the -DAUM_SCANT_CK_F32 A/B build (it uses the same LDS).
#ifndef AUM_SCANT_DBC_MERGE
#if defined(AUM_SCANT_CK_F32)
#define AUM_SCANT_DBC_MERGE 0
#else
#define AUM_SCANT_DBC_MERGE 1       // -DAUM_SCANT_DBC_MERGE=0: both directions' rows go to the reduce kernel, as until round 4 (A/B builds)
#endif
#endif
AUM_HOSTDEV constexpr int scant_dbc_rows_to_sum(int nparts, bool bidir) { return bidir && AUM_SCANT_DBC_MERGE ? nparts / 2 : nparts; }
// partials of one (batch entry, direction, channel group) wave
struct ScanTBwdOut {
    float* dbc;        // [batch][len][nparts][2N] fp32 partial rows
    float* dA;         // [ndir][batch][N][dim]
    float* dD;         // [ndir][batch][dim]
    float* dbias;      // [ndir][batch][dim]
    int nparts;
    float* carry;      // bidirectional: [workgroup][direction][34][64] -- the carries of the pair whose two phases run on different waves
    unsigned long long* trace;      // -DAUM_SCANT_TRACE builds (tools/tm_trace.py): 16 x uint64 per wave behind the partials; else unused
};

#ifndef AUM_SCANT_TAIL2
#define AUM_SCANT_TAIL2 1     // 0 (A/B builds): each butterfly complete where its terms exist, as in round 3
#endif
#ifndef AUM_SCANT_BABL
#define AUM_SCANT_BABL 0      // timing experiments only (wrong results): 1 no butterflies, 2 no exponentials, 4 carries at a fixed register index,
#endif                        // 8 no loads of the next block, 16 no entry-state loads, 32 no B/C reads from LDS
template <class T, int N, int PHASE, bool SP, bool HAS_Z>
AUM_DEV void scant_bwd_run(const AumScanTmBwdArgs& p, const ScanTBwdOut& wo, int b, int e0, int dir, int part, int t0, int tstep, int it0,
                           int it1, const float* Aptr, float dmul, vf16& hh, vf16& dAacc, vf& dDacc, vf& dbacc, float* lds,
                           unsigned long long* tacc = nullptr) {
    using TL = ScanTTile<T>;
    constexpr int ES = TL::ES, NLD = TL::NLD;
    constexpr int LROW = 128, PLANE = SCANT_CK * LROW;          // bytes: a tile row segment, a plane of eight of them
    constexpr bool FINAL = PHASE != 1, LD_PART = PHASE == 2;
    static_assert(N == 16 && SCANT_CK == 8, "8 steps x 8 state pairs");
    const int L = p.len;
    const vi lane = lane_id();
    const vi ec = lane + e0;
    const vf biasv = p.delta_bias ? gload_u(p.delta_bias, ec) : splat(0.f);
    const vf Dv = p.D ? gload_u(p.D, ec) * dmul : splat(0.f);
    const gbuf<T> ubuf = make_gbuf(row_ptr<T>(p.u, (int64_t)b * p.u_bs));
    const gbuf<T> dbuf = make_gbuf(row_ptr<T>(p.delta, (int64_t)b * p.delta_bs));
    const gbuf<T> zbuf = make_gbuf(HAS_Z ? row_ptr<T>(p.z, (int64_t)b * p.z_bs) : row_ptr<T>(p.u, 0));
    const gbuf<T> gbuf_ = make_gbuf(row_ptr<T>(p.dout, (int64_t)b * p.dout_bs));
    const gbuf<T> ybuf = make_gbuf(HAS_Z ? row_ptr<T>(p.out_pre, (int64_t)b * p.pre_bs) : row_ptr<T>(p.u, 0));
    const gbuf<T> dubuf = make_gbuf(row_ptr<T>(p.du, (int64_t)b * p.du_bs));
    const gbuf<T> ddbuf = make_gbuf(row_ptr<T>(p.ddelta, (int64_t)b * p.ddelta_bs));
    const gbuf<T> dzbuf = make_gbuf(HAS_Z ? row_ptr<T>(p.dz, (int64_t)b * p.dz_bs) : row_ptr<T>(p.du, 0));
    const gbuf<T> Bbuf = make_gbuf(row_ptr<T>(p.B, (int64_t)b * p.B_bs));
    const gbuf<T> Cbuf = make_gbuf(row_ptr<T>(p.C, (int64_t)b * p.C_bs));
    const gbuf<float> Abuf = make_gbuf(Aptr);
    const int nck = scant_nck(L);
    constexpr int CKR = scant_ck_rows<T>();        // rows of a checkpoint: the 16 states (fp32 activations) or 8 bf16 pairs (16-bit activations)
    constexpr int RPP = CKR / (N / 2);             // rows a pass over one state pair consumes
    const gbuf<float> ckbuf = make_gbuf(p.ckpt + ((int64_t)dir * p.batch + b) * (nck > 0 ? nck : 1) * CKR * p.dim);
    const gbuf<float> dbcbuf = make_gbuf(wo.dbc + (int64_t)b * L * wo.nparts * (2 * N));
    const int u_tb = (int)p.u_ts * ES, d_tb = (int)p.delta_ts * ES, z_tb = HAS_Z ? (int)p.z_ts * ES : 0, g_tb = (int)p.dout_ts * ES,
              y_tb = HAS_Z ? (int)p.pre_ts * ES : 0, du_tb = (int)p.du_ts * ES, dd_tb = (int)p.ddelta_ts * ES,
              dz_tb = HAS_Z ? (int)p.dz_ts * ES : 0, B_tb = (int)p.B_ts * ES, C_tb = (int)p.C_ts * ES;
    // `part`: the channel group's slot.  Fo-Bi with merged hand-over: first phase -> slot nparts / 2 + part, second phase reads that slot
    // of the OTHER direction (same group) and writes slot part
    constexpr bool DBC_MERGE = AUM_SCANT_DBC_MERGE && PHASE != 0;
    const int dbc_tb = wo.nparts * (2 * N) * 4;
    const int dbc_col_in = (wo.nparts / 2 + part) * (2 * N) * 4;
    const int dbc_col = DBC_MERGE && PHASE == 1 ? dbc_col_in : part * (2 * N) * 4;
    auto tok = [&](int it) { return t0 + it * tstep; };
    // u is read again in a block's last lines, after the next block's u arrived: two tiles, by block parity
    float* t_u0 = lds;
    float* t_d = lds + TL::FLOATS;
    float* t_z = lds + 2 * TL::FLOATS;
    float* t_g = lds + 3 * TL::FLOATS;
    float* t_y = lds + 4 * TL::FLOATS;
    float* t_du = lds + 5 * TL::FLOATS;
    float* t_dd = lds + 6 * TL::FLOATS;
    float* t_dz = lds + 7 * TL::FLOATS;
    float* t_bc = lds + 8 * TL::FLOATS;
    float* t_dbc = t_bc + SCANT_BC_BLOCK;
    float* t_u1 = t_dbc + SCANT_BC_BLOCK;
    float* t_ck = t_u1 + TL::FLOATS;               // entry state of the block: row n = state n, one float per lane (16-bit activations: row j = the bf16 pair of states 2j, 2j+1)
    float* t_A = t_ck + N * WAVE;                  // A[e][n] likewise
    float* t_rawB = t_A + N * WAVE;                // the next block's B / C pairs as loaded (NLD dwords per lane and tensor)
    float* t_rawC = t_rawB + NLD * WAVE;
    // the other direction's dB | dC row block of this block (second phase): [step][dB | dC] fp32 like t_dbc
    float* t_dbi = sizeof(T) == 4 ? t_rawC + NLD * WAVE : t_ck + CKR * WAVE;
    auto t_u_of = [&](int blk) { return (blk & 1) ? t_u1 : t_u0; };
    // lane = (row st_i, 16-byte column st_c) of a 8 x 128-byte plane; the row is the block's step st_i IN ITERATION ORDER, which a
    // reversed direction finds at memory row 7 - st_i of the block
    const vi st_i = lane >> 3, st_c = lane & 7;
    const vi st_r = tstep > 0 ? st_i : (SCANT_CK - 1) - st_i;
    const vi st_gcol = st_c * 16 + e0 * ES;
    const vi st_lds = lane * 16;
    const vi el_off = ((lane * ES) >> 7) * PLANE + ((lane * ES) & (LROW - 1));       // this lane's channel in row 0 of a tile
    const vi bc_slot = st_i * SCANT_BC_ROW + st_c * 2;
    const vi vo4 = ec * 4;
    const vi dbc_slot = (((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 4) & 1)) * SCANT_BC_ROW + ((lane >> 5) & 1);
    // per-lane row of block `blk` in a global tensor: memory row st_r for blocks inside the phase (plus the scalar offset of the
    // block's lowest time step), the clamped row's time step for ragged ones
    struct Rows { vi rowt; int t_lo; vm valid; bool inside; };
    auto rows_of = [&](int blk) {
        Rows r;
        const int base = blk * SCANT_CK;
        r.inside = base >= it0 && base + SCANT_CK <= it1;
        r.t_lo = 0;
        r.valid = lane >= 0;
        if (r.inside) {
            r.t_lo = tstep > 0 ? tok(base) : tok(base + SCANT_CK - 1);
            r.rowt = st_r;
        } else {
            // loads: any real step of the direction (the forward sweep of a block re-runs its steps before the phase's first one -- in
            // PHASE 1; the other phases start at the direction's first step: nothing exists in front of it0; none of the steps behind
            // it1 - 1 is used); stores: the phase's own steps only
            const vi it = st_i + base;
            r.valid = (it >= it0) && (it < it1);
            r.rowt = vmax_i(vmin_i(it, it1 - 1), PHASE != 1 ? it0 : 0) * tstep + t0;
        }
        return r;
    };
    auto load_tile = [&](const gbuf<T>& buf, int tb, const Rows& r, float* tile) {
        AUM_UNROLL
        for (int i = 0; i < NLD; ++i) gbuf_load16_lds(buf, r.rowt * tb + st_gcol + LROW * i, r.t_lo * tb, tile + (PLANE / 4) * i);
    };
    auto store_tile = [&](const gbuf<T>& buf, int tb, const float* tile, const Rows& r) {
        AUM_UNROLL
        for (int i = 0; i < NLD; ++i) {
            const vq q = lds_read16(tile, st_lds + PLANE * i);
            if (r.inside) gbuf_store16(buf, r.rowt * tb + st_gcol + LROW * i, r.t_lo * tb, q);
            else gbuf_store16_m(buf, r.rowt * tb + st_gcol + LROW * i, 0, q, r.valid);
        }
    };
    // the input tensors of block `blk` (7 x NLD loads); B / C as the pairs each lane will convert
    auto request_block = [&](int blk, const Rows& r) {
        load_tile(ubuf, u_tb, r, t_u_of(blk));
        load_tile(dbuf, d_tb, r, t_d);
        load_tile(gbuf_, g_tb, r, t_g);
        if (HAS_Z) {
            load_tile(zbuf, z_tb, r, t_z);
            load_tile(ybuf, y_tb, r, t_y);
        }
        AUM_UNROLL
        for (int i = 0; i < NLD; ++i) {
            gbuf_load4_lds(Bbuf, r.rowt * B_tb + st_c * (2 * ES) + 4 * i, r.t_lo * B_tb, t_rawB + WAVE * i);
            gbuf_load4_lds(Cbuf, r.rowt * C_tb + st_c * (2 * ES) + 4 * i, r.t_lo * C_tb, t_rawC + WAVE * i);
        }
    };
    // rows [n0, n1) of the entry state of block `blk` (the checkpoint after block blk - 1; zero for the first block)
    auto request_entry = [&](int blk, int n0, int n1) {
        if (AUM_SCANT_BABL & 16) return;
        for (int n = n0; n < n1; ++n) {
            if (blk > 0) gbuf_load4_lds(ckbuf, vo4, ((blk - 1) * CKR + n) * p.dim * 4, t_ck + n * WAVE);
            else lds_write(t_ck, lane + n * WAVE, splat(0.f));
        }
    };
    // the raw B / C pairs in LDS -> fp32 [step][B_0..15 | C_0..15]
    auto convert_bc = [&]() {
        vf b0, b1, c0, c1;
        lds_pair_to_f32<T>(t_rawB, b0, b1);
        lds_pair_to_f32<T>(t_rawC, c0, c1);
        lds_write2(t_bc, bc_slot, b0, b1);
        lds_write2(t_bc, bc_slot + N, c0, c1);
    };

    if (it0 >= it1) return;
    const int blk_lo = it0 / SCANT_CK, blk_hi = (it1 + SCANT_CK - 1) / SCANT_CK;       // blocks blk_hi-1 down to blk_lo
#if defined(AUM_SCANT_TRACE) && !defined(AUM_EMU)
#define AUM_TMB_STAMP(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); tacc[k] += now_ - stamp_; stamp_ = now_; } while (0)
    unsigned long long stamp_ = __builtin_readcyclecounter();
#elif defined(AUM_SCANT_MARKS) && !defined(AUM_EMU)
    // listings only (tools/isa_mix_tm.py --marks): a comment line in the assembly where a stage of the block begins
#define AUM_TMB_STAMP(k) asm volatile("; AUM_MARK " #k)
#else
#define AUM_TMB_STAMP(k) do { } while (0)
#endif
    // prologue of the phase: everything the first block needs, and A
    wave_lds_fence();
    request_block(blk_hi - 1, rows_of(blk_hi - 1));
    request_entry(blk_hi - 1, 0, CKR);
    for (int n = 0; n < N; ++n) gbuf_load4_lds(Abuf, ec * (N * 4), n * 4, t_A + n * WAVE);
    // Memory operations complete in issue order, and s_waitcnt vmcnt(n) returns once at most n are in flight: every wait below names
    // exactly the operations YOUNGER than the data it needs.  Issue order around a block:
    //   [previous block's passes: 16 entry rows of this block] [its end: NST stores] [this block's first lines: RBN loads of the next
    //   block's tensors, PART loads of this block's partial du / ddelta] [this block's passes: 2 entry rows of the next block each]
    constexpr int NST = 2 * NLD + ((HAS_Z && FINAL) ? NLD : 0) + 1;       // stores at the end of a block: du, ddelta, dz, dB/dC
    constexpr int RBN = ((HAS_Z ? 5 : 3) + 2) * NLD;                      // request_block
    constexpr int PART = LD_PART ? 2 * NLD + (DBC_MERGE ? 1 : 0) : 0;
    AUM_TMB_STAMP(0);
    // one block; FULL: all eight steps belong to the phase (no per-step conditions)
// Two waves share a SIMD for the whole kernel and the arbiter serves equal priorities oldest first: left alone, the older wave of every
// SIMD runs ahead, finishes at ~0.7 of the kernel's span and leaves the younger one to run the rest alone -- at the issue rate of ONE wave
// (profiles/r04_trace_bwd.txt: wave durations 841 ... 1188 us in a 1189 us launch).  So the two take turns: priority 3 / 0, exchanged with
// every 8-step block, offset by the wave's slot on its SIMD (round 5; same box, bench launch: 1.016 -> 0.962 ms; by pass instead of by
// block 0.976; the odd slot at a fixed priority 1.004; levels 1 / 0 0.949 against 0.943 for 3 / 0 on another box: profiles/r05_ab_bwd_prio.txt).
// -DAUM_SCANT_BPRIO=0: no priorities (round 4); 2: the odd slot holds priority 1; 3: the turn changes with the pass.
#ifndef AUM_SCANT_BPRIO
#define AUM_SCANT_BPRIO 1
#endif
#ifndef AUM_SCANT_BPRIO_SHIFT
#define AUM_SCANT_BPRIO_SHIFT 0
#endif
#ifndef AUM_SCANT_BPRIO_HI
#define AUM_SCANT_BPRIO_HI 3
#endif
    const int bslot = AUM_SCANT_BPRIO ? wave_slot_on_simd() : 0;
    auto do_block = [&](auto full_tag, int blk) {
        constexpr bool FULL = decltype(full_tag)::value;
        if (AUM_SCANT_BPRIO == 1) {
            if (((blk >> AUM_SCANT_BPRIO_SHIFT) + bslot) & 1) AUM_SET_PRIO(AUM_SCANT_BPRIO_HI);
            else AUM_SET_PRIO(0);
        } else if (AUM_SCANT_BPRIO == 2) {
            if (bslot & 1) AUM_SET_PRIO(1);
        }
        const int base = blk * SCANT_CK;
        const int s_lo = FULL ? 0 : (it0 > base ? it0 - base : 0);                       // steps [s_lo, s_hi) of the block are this phase's
        const int s_hi = FULL ? SCANT_CK : (it1 < base + SCANT_CK ? it1 - base : SCANT_CK);
        const int s_min = FULL || PHASE == 1 ? 0 : s_lo;       // a phase that starts at the direction's first step: no steps in front of it
        const bool more = blk > blk_lo;
        const bool cknext = more && blk - 1 > 0;        // the next block has a checkpoint to fetch (else its entry rows are zeroed)
        // the block's tensors (requested in the previous block's first lines) are older than that block's partials, the entry rows
        // its passes requested and its stores
        if (blk == blk_hi - 1) AUM_WAIT_VM(0);
        else if (blk > 0) AUM_WAIT_VM(PART + CKR + NST);
        else AUM_WAIT_VM(PART + NST);
        wave_lds_fence();
        AUM_TMB_STAMP(9);
        convert_bc();
        // ---- per-step registers of the block; dz of its steps --------------------------------------------------
        // Pd[i] = (delta_2i, delta_2i+1), Pu[i] = (delta u)_2i, _2i+1, Q[i] = (dy_2i, dy_2i+1): pairs of DIFFERENT values -- a packed instruction
        // broadcasts either half of a register pair as an operand modifier, whereas a (d, d) operand kept across the pass loop is two registers.
        // Pairs over STEPS (round 6; until round 5 one pair per step: (delta_s, (delta u)_s)): the arithmetic of this stage and of the block's
        // last lines then runs on two steps per packed instruction -- the same IEEE operations in the same order, bit-equal results -- and
        // the passes pick a step's half exactly as they always did for Q.  (profiles/r06_isa_mix_tm.txt: a quarter of the kernel's vector-ALU
        // instructions are outside the passes.)
        vf2 Pd[SCANT_CK / 2], Pu[SCANT_CK / 2], Q[SCANT_CK / 2];
        float* const t_u = t_u_of(blk);
        {
            // all LDS reads first (raw: 40 registers that are free here), then the arithmetic pair of steps by pair of steps
            vi ru[SCANT_CK], rd[SCANT_CK], rg[SCANT_CK], rz[SCANT_CK], ry[SCANT_CK];
            AUM_UNROLL
            for (int s = 0; s < SCANT_CK; ++s) {
                const vi off = el_off + s * LROW;
                ru[s] = lds_read_raw<T>(t_u, off);
                rd[s] = lds_read_raw<T>(t_d, off);
                rg[s] = lds_read_raw<T>(t_g, off);
                if (HAS_Z) rz[s] = lds_read_raw<T>(t_z, off);
                if (HAS_Z && FINAL) ry[s] = lds_read_raw<T>(t_y, off);
            }
            AUM_SCHED_FENCE();
            const vf2 bias2 = spl2(biasv), one2 = spl2(splat(1.f));
            AUM_UNROLL
            for (int i = 0; i < SCANT_CK / 2; ++i) {
                const int s0 = 2 * i, s1 = 2 * i + 1;
                const vf2 us = mk2(raw_to_f32<T>(ru[s0]), raw_to_f32<T>(ru[s1]));
                vf2 d = mk2(raw_to_f32<T>(rd[s0]), raw_to_f32<T>(rd[s1])) + bias2;
                if (SP) d = vsoftplus2(d);
                Pd[i] = d;
                Pu[i] = d * us;
                const vf2 go = mk2(raw_to_f32<T>(rg[s0]), raw_to_f32<T>(rg[s1]));
                if (HAS_Z) {
                    const vf2 zz = mk2(raw_to_f32<T>(rz[s0]), raw_to_f32<T>(rz[s1]));
                    const vf2 sg = vsigmoid2(zz);
                    Q[i] = go * (zz * sg);
                    if (FINAL) {        // dz = dout ytot d(z sigmoid(z))/dz  (direction-independent: written by whoever finishes the step)
                        const vf2 yt = mk2(raw_to_f32<T>(ry[s0]), raw_to_f32<T>(ry[s1]));
                        const vf2 dzv = go * yt * (sg * vfma2(zz, one2 - sg, one2));
                        lds_write_elem<T>(t_dz, el_off + s0 * LROW, lo2(dzv));
                        lds_write_elem<T>(t_dz, el_off + s1 * LROW, hi2(dzv));
                    }
                } else {
                    Q[i] = go;
                }
            }
        }
        // a step's delta / delta u as a broadcast operand
        auto DS = [&](int s) { return (s & 1) ? bc_hi(Pd[s >> 1]) : bc_lo(Pd[s >> 1]); };
        auto US = [&](int s) { return (s & 1) ? bc_hi(Pu[s >> 1]) : bc_lo(Pu[s >> 1]); };
        AUM_TMB_STAMP(10);
        vf2 S1[SCANT_CK], S2[SCANT_CK];      // per state PAIR: one packed fma per step and sum; the two halves are added once per block
        AUM_UNROLL
        for (int s = 0; s < SCANT_CK; ++s) S1[s] = S2[s] = spl2(splat(0.f));
        wave_lds_fence();
        const Rows rc = rows_of(blk);
        // the input tiles are registers now: the next block's tensors, and (second phase) the other direction's partial du / ddelta of
        // THIS block into the output tiles, which idle until the block's last lines
        if (!(AUM_SCANT_BABL & 8)) {
            if (more) request_block(blk - 1, rows_of(blk - 1));
            if (LD_PART) {
                load_tile(dubuf, du_tb, rc, t_du);
                load_tile(ddbuf, dd_tb, rc, t_dd);
                if constexpr (DBC_MERGE)        // lane = (step st_i, 16-byte chunk st_c) of the [8][2N] fp32 block: the layout of the store below
                    gbuf_load16_lds(dbcbuf, rc.rowt * dbc_tb + st_c * 16 + dbc_col_in, rc.t_lo * dbc_tb, t_dbi);
            }
        }
        AUM_TMB_STAMP(1);
        // ---- passes over the state pairs: a real loop (unrolled, the optimiser spread its work over all eight passes and needed more
        // than 340 registers); the pairs' carries sit in register vectors indexed by the loop counter
        _Pragma("nounroll")
        for (int j = 0; j < N / 2; ++j) {
            // B and C of the pass's state pair for all eight steps, the pair's entry state and A, requested before anything else: the
            // exponentials cover their latency (read step by step inside the sweeps, every read was waited for on the spot:
            // twelve LDS round trips per pass in a chain that two waves per SIMD cannot hide)
            // this pass's entry rows: younger are the later rows (CKR - RPP (j + 1)), the previous block's stores, this block's requests and
            // the RPP j rows of the next block requested so far -- a constant; fewer were issued near the ends of a phase
            if (AUM_SCANT_BPRIO == 3) {          // A/B: the turn changes with the pass, not the block
                if (((j >> AUM_SCANT_BPRIO_SHIFT) + bslot) & 1) AUM_SET_PRIO(AUM_SCANT_BPRIO_HI);
                else AUM_SET_PRIO(0);
            }
            if (blk != blk_hi - 1) {
                if (more && cknext) AUM_WAIT_VM((CKR - RPP) + NST + RBN + PART);
                else if (more) AUM_WAIT_VM(NST + RBN + PART);
                else AUM_WAIT_VM(NST + PART);
            }
            // (the LDS answers in order: A first, the exponentials wait for nothing else)
            const vf2 Aj = mk2(lds_read(t_A, lane + (2 * j) * WAVE), lds_read(t_A, lane + (2 * j + 1) * WAVE));
            vf2 x;
            if constexpr (RPP == 2) {
                x = mk2(lds_read(t_ck, lane + (2 * j) * WAVE), lds_read(t_ck, lane + (2 * j + 1) * WAVE));
            } else {
                vf x_lo, x_hi;
                lds_pair_to_f32<T>(t_ck + j * WAVE, x_lo, x_hi);
                x = mk2(x_lo, x_hi);
            }
            vf qb[SCANT_CK][2], qc[SCANT_CK][2];
            AUM_UNROLL
            for (int s = 0; s < SCANT_CK; ++s) {
                if (AUM_SCANT_BABL & 32) {
                    qb[s][0] = qb[s][1] = qc[s][0] = qc[s][1] = biasv;
                    continue;
                }
                lds_read2_u(t_bc, s * SCANT_BC_ROW + 2 * j, qb[s]);
                lds_read2_u(t_bc, s * SCANT_BC_ROW + N + 2 * j, qc[s]);
            }
            const vf2 A2j = Aj * spl2(splat(LOG2E));
            AUM_TMB_STAMP(2);
            AUM_SCHED_FENCE();        // passes are scheduled one by one: across them the scheduler's reordering costs registers (2.5 KB of scratch)
            // the per-step values are loop invariants, and a packed operand (d, d) built OUTSIDE the loop is a real register pair
            // (48 registers for 24 values); opaque to the compiler here, the broadcast is an operand modifier of the packed instruction
            AUM_UNROLL
            for (int i = 0; i < SCANT_CK / 2; ++i) {
                pin_value2(Pd[i]);
                pin_value2(Pu[i]);
                pin_value2(Q[i]);
            }
            const int jr = (AUM_SCANT_BABL & 4) ? 0 : j;
            vf2 hj = mk2(vf16_get(hh, 2 * jr), vf16_get(hh, 2 * jr + 1)), dAj = mk2(vf16_get(dAacc, 2 * jr), vf16_get(dAacc, 2 * jr + 1));
            vf2 w[SCANT_CK], a[SCANT_CK];
            vf pc[16], pb[16];
            // a = exp2(delta A2) of the eight steps: independent of the recurrences, used by both sweeps
            AUM_UNROLL
            for (int s = 0; s < SCANT_CK; ++s) a[s] = (AUM_SCANT_BABL & 2) ? DS(s) * A2j : vexp2_2(DS(s) * A2j);
            // forward sweep: steps 0 .. s_hi-1
            AUM_UNROLL
            for (int s = 0; s < SCANT_CK; ++s) {
                if (!FULL) {
                    pc[2 * s] = pc[2 * s + 1] = splat(0.f);
                    pb[2 * s] = pb[2 * s + 1] = splat(0.f);
                }
                vf2 pcs = spl2(splat(0.f));
                if (FULL || (s >= s_min && s < s_hi)) {
                    w[s] = a[s] * x;
                    x = vfma2(US(s), mk2(qb[s][0], qb[s][1]), w[s]);
                    if (FULL || s >= s_lo) {
                        pcs = ((s & 1) ? bc_hi(Q[s >> 1]) : bc_lo(Q[s >> 1])) * x;
                        pc[2 * s] = lo2(pcs);
                        pc[2 * s + 1] = hi2(pcs);
                    }
                }
            }
            // the entry rows this pass consumed make room for the next block's
            if (cknext) request_entry(blk - 1, RPP * j, RPP * j + RPP);
            AUM_SCHED_FENCE();
            AUM_TMB_STAMP(3);
            vf dCsum = splat(0.f);
            vf hc[4], hb[4];
            // the dC butterfly's 24 independent levels now (16 term registers -> 4); its serial tail runs after the reverse sweep,
            // interleaved with the dB butterfly's (wave.h, wave_sum16_tail2)
            if (AUM_SCANT_BABL & 1) hc[0] = pc[0] + pc[5];
            else if (AUM_SCANT_TAIL2) wave_sum16_head(pc, hc);
            else dCsum = wave_sum16(pc);
            AUM_SCHED_FENCE();
            AUM_TMB_STAMP(4);
            // reverse sweep: steps s_hi-1 .. s_lo
            AUM_UNROLL
            for (int s = SCANT_CK - 1; s >= 0; --s) {
                vf2 pbs = spl2(splat(0.f));
                if (FULL || (s >= s_lo && s < s_hi)) {
                    const vf2 g = vfma2((s & 1) ? bc_hi(Q[s >> 1]) : bc_lo(Q[s >> 1]), mk2(qc[s][0], qc[s][1]), hj);
                    pbs = g * US(s);
                    pb[2 * s] = lo2(pbs);
                    pb[2 * s + 1] = hi2(pbs);
                    S1[s] = vfma2(g, mk2(qb[s][0], qb[s][1]), S1[s]);
                    const vf2 r = g * w[s];
                    S2[s] = vfma2(A2j, r, S2[s]);
                    dAj = vfma2(DS(s), r, dAj);
                    hj = a[s] * g;
                }
            }
            AUM_SCHED_FENCE();
            AUM_TMB_STAMP(5);
            vf dBsum;
            if (AUM_SCANT_BABL & 1) {
                dBsum = pb[0] + pb[7];
                dCsum = hc[0];
            } else if (AUM_SCANT_TAIL2) {
                wave_sum16_head(pb, hb);
                wave_sum16_tail2(hc, hb, dCsum, dBsum);
            } else {
                dBsum = wave_sum16(pb);
            }
            // lane l holds the totals of value k = wave_sum16_value_of_lane(l) = (step k >> 1, state 2j + (k & 1)).  The four lanes of a
            // quad hold the same total and all write it to the same slot: a lane mask here is a branch, and everything the scheduler
            // sinks below it loses its packed-operand broadcasts (they are folded per basic block)
            const vi slot = dbc_slot + 2 * j;
            lds_write(t_dbc, slot, dBsum);
            lds_write(t_dbc, slot + N, dCsum);
            vf16_set(hh, 2 * jr, lo2(hj));
            vf16_set(hh, 2 * jr + 1, hi2(hj));
            vf16_set(dAacc, 2 * jr, lo2(dAj));
            vf16_set(dAacc, 2 * jr + 1, hi2(dAj));
            AUM_TMB_STAMP(6);
        }
        // ---- the block's du, ddelta; partials / finish ----------------------------------------------------------
        // this block's partial du / ddelta are back (younger: the CKR entry rows requested during the passes)
        if (cknext) AUM_WAIT_VM(CKR);
        else AUM_WAIT_VM(0);
        wave_lds_fence();
        if (more && !cknext) request_entry(0, 0, CKR);        // the first block starts from zero
        vi rus[SCANT_CK], rpu[SCANT_CK], rpd[SCANT_CK];
        AUM_UNROLL
        for (int s = 0; s < SCANT_CK; ++s) {
            const vi off = el_off + s * LROW;
            rus[s] = lds_read_raw<T>(t_u, off);
            if (FINAL && LD_PART) {
                rpu[s] = lds_read_raw<T>(t_du, off);
                rpd[s] = lds_read_raw<T>(t_dd, off);
            }
        }
        AUM_SCHED_FENCE();
        if constexpr (FULL) {
            // two steps per packed instruction; the two running sums (dD, dbias) keep their step order (one fma / add per step): bit-equal to the
            // step-by-step form below
            const vf2 Dv2 = spl2(Dv), one2 = spl2(splat(1.f));
            AUM_UNROLL
            for (int i = 0; i < SCANT_CK / 2; ++i) {
                const int s0 = 2 * i, s1 = 2 * i + 1;
                const vf2 dls = Pd[i], dys = Q[i];
                const vf2 us = mk2(raw_to_f32<T>(rus[s0]), raw_to_f32<T>(rus[s1]));
                const vf2 s1v = mk2(lo2(S1[s0]) + hi2(S1[s0]), lo2(S1[s1]) + hi2(S1[s1]));
                const vf2 s2v = mk2(lo2(S2[s0]) + hi2(S2[s0]), lo2(S2[s1]) + hi2(S2[s1]));
                vf2 du = dls * s1v;
                vf2 dd = vfma2(us, s1v, s2v * spl2(splat(LN2)));
                if (FINAL) {
                    du = vfma2(Dv2, dys, du);
                    dDacc = vfma(lo2(dys), lo2(us), dDacc);
                    dDacc = vfma(hi2(dys), hi2(us), dDacc);
                    if (LD_PART) {
                        du = du + mk2(raw_to_f32<T>(rpu[s0]), raw_to_f32<T>(rpu[s1]));
                        dd = dd + mk2(raw_to_f32<T>(rpd[s0]), raw_to_f32<T>(rpd[s1]));
                    }
                    if (SP) dd = dd * (one2 - vexp2_2(dls * spl2(splat(-LOG2E))));      // sigmoid(raw) = 1 - exp(-softplus(raw))
                    dbacc = dbacc + lo2(dd);
                    dbacc = dbacc + hi2(dd);
                }
                lds_write_elem<T>(t_du, el_off + s0 * LROW, lo2(du));
                lds_write_elem<T>(t_du, el_off + s1 * LROW, hi2(du));
                lds_write_elem<T>(t_dd, el_off + s0 * LROW, lo2(dd));
                lds_write_elem<T>(t_dd, el_off + s1 * LROW, hi2(dd));
            }
        } else {
            AUM_UNROLL
            for (int s = 0; s < SCANT_CK; ++s) {
                if (s >= s_lo && s < s_hi) {
                    const vi off = el_off + s * LROW;
                    const vf dls = (s & 1) ? hi2(Pd[s >> 1]) : lo2(Pd[s >> 1]), dys = (s & 1) ? hi2(Q[s >> 1]) : lo2(Q[s >> 1]);
                    const vf us = raw_to_f32<T>(rus[s]);
                    const vf s1 = lo2(S1[s]) + hi2(S1[s]), s2 = lo2(S2[s]) + hi2(S2[s]);
                    vf du = dls * s1;
                    vf dd = vfma(us, s1, s2 * LN2);
                    if (FINAL) {
                        du = vfma(Dv, dys, du);
                        dDacc = vfma(dys, us, dDacc);
                        if (LD_PART) {
                            du = du + raw_to_f32<T>(rpu[s]);
                            dd = dd + raw_to_f32<T>(rpd[s]);
                        }
                        if (SP) dd = dd * (splat(1.f) - vexp2(dls * (-LOG2E)));      // sigmoid(raw) = 1 - exp(-softplus(raw))
                        dbacc = dbacc + dd;
                    }
                    lds_write_elem<T>(t_du, off, du);
                    lds_write_elem<T>(t_dd, off, dd);
                }
            }
        }
        wave_lds_fence();
        AUM_TMB_STAMP(7);
        store_tile(dubuf, du_tb, t_du, rc);
        store_tile(ddbuf, dd_tb, t_dd, rc);
        if (HAS_Z && FINAL) store_tile(dzbuf, dz_tb, t_dz, rc);
        {       // dB / dC partial rows of the block: [step][2N] fp32 = 16 bytes per lane
            vq q = lds_read16(t_dbc, st_i * (SCANT_BC_ROW * 4) + st_c * 16);
            if constexpr (DBC_MERGE && LD_PART) {
                const vq qi = lds_read16(t_dbi, st_i * (SCANT_BC_ROW * 4) + st_c * 16);
                AUM_UNROLL
                for (int k = 0; k < 4; ++k) q.w[k] = f32_as_int(int_as_f32(q.w[k]) + int_as_f32(qi.w[k]));
            }
            if (rc.inside) gbuf_store16(dbcbuf, rc.rowt * dbc_tb + st_c * 16 + dbc_col, rc.t_lo * dbc_tb, q);
            else gbuf_store16_m(dbcbuf, rc.rowt * dbc_tb + st_c * 16 + dbc_col, 0, q, rc.valid);
        }
        wave_lds_fence();
        AUM_TMB_STAMP(8);
    };
    for (int blk = blk_hi - 1; blk >= blk_lo; --blk) {
        if (blk * SCANT_CK >= it0 && blk * SCANT_CK + SCANT_CK <= it1) do_block(ScanTTag<true>{}, blk);
        else do_block(ScanTTag<false>{}, blk);
    }
#undef AUM_TMB_STAMP
}

// workgroup = four waves as in the forward.  Workspace partials are summed by scant_bwd_reduce.
template <class T, bool SP, bool HAS_Z, bool BIDIR>
AUM_DEV void scant_bwd(const AumScanTmBwdArgs& p, const ScanTBwdOut& wo, int wg, float* lds, unsigned long long* tacc = nullptr) {
    constexpr int N = SCANT_N;
    constexpr int NW = SCANT_NW;
    constexpr int UPW = scant_units_per_wg<BIDIR>();
    const int gpb = p.dim / WAVE;
    const int units = p.batch * gpb;
    const int L = p.len;
    vf16 hh[AUM_PER_WAVE(NW)], dA[AUM_PER_WAVE(NW)];
    vf dD[AUM_PER_WAVE(NW)], dbias[AUM_PER_WAVE(NW)];
    auto finish = [&](int w, int unit, int d) {      // per-wave partial sums of dA, dD, ddelta_bias
        const int b = unit / gpb, e0 = (unit % gpb) * WAVE;
        const vi ec = lane_id() + e0;
        float* pa = wo.dA + ((int64_t)d * p.batch + b) * N * p.dim;
        AUM_UNROLL
        for (int n = 0; n < N; ++n) gstore(pa + (int64_t)n * p.dim, ec, vf16_get(dA[AUM_W(w)], n), ec >= 0);
        gstore(wo.dD + ((int64_t)d * p.batch + b) * p.dim, ec, dD[AUM_W(w)] * (BIDIR ? 2.f : 1.f), ec >= 0);
        gstore(wo.dbias + ((int64_t)d * p.batch + b) * p.dim, ec, dbias[AUM_W(w)], ec >= 0);
    };
    auto init = [&](int w) {
        AUM_UNROLL
        for (int n = 0; n < N; ++n) {
            vf16_set(hh[AUM_W(w)], n, splat(0.f));
            vf16_set(dA[AUM_W(w)], n, splat(0.f));
        }
        dD[AUM_W(w)] = dbias[AUM_W(w)] = splat(0.f);
    };
    if (!BIDIR) {
        const bool rev = (p.flags & AUM_SCAN_REVERSE) != 0;
        AUM_FOR_EACH_WAVE(w, NW) {
            const int unit = wg * UPW + w;
            if (unit < units) {
                init(w);
                scant_bwd_run<T, N, 0, SP, HAS_Z>(p, wo, unit / gpb, (unit % gpb) * WAVE, 0, unit % gpb, rev ? L - 1 : 0, rev ? -1 : 1, 0, L, p.A, 1.f,
                                                  hh[AUM_W(w)], dA[AUM_W(w)], dD[AUM_W(w)], dbias[AUM_W(w)], lds + w * scant_bwd_lds_wave_floats<T>(), tacc);
                finish(w, unit, 0);
            }
        }
        return;
    }
    // Bidirectional.  The backward walks a direction's iterations from the last to the first: its first phase is the iterations the
    // forward's second phase ran, [first_half, L), its second phase [0, first_half).  The kernel holds 256 registers per lane, two
    // waves per SIMD, and B = 64, E = 1536 is THREE direction waves per SIMD: a workgroup per channel-group pair would leave the last
    // third of the run to lone waves that take as long as paired ones (the pass is a latency chain).  So a workgroup of four waves
    // takes three pairs X, Y, Z through three stages of one phase each -- 512 workgroups, two per CU, every SIMD holds two waves for
    // the whole kernel:
    //     waves 0,1:  X phase 1 | X phase 2 | Y phase 2          waves 2,3:  Y phase 1 | Z phase 1 | Z phase 2
    // Y's carries (adjoint state, dA, dD, ddelta_bias partial sums: 34 values per lane) cross from waves 2,3 to waves 0,1 through
    // global memory, ordered by the two barriers in between like the du / ddelta partials of a pair.
    const int npairs = units;
    _Pragma("nounroll")
    for (int stage = 0; stage < 3; ++stage) {
        AUM_FOR_EACH_WAVE(w, NW) {
            const int h = w >> 1, d = w & 1, io = scant_grid_shift(L, d);
            const int slot = h == 0 ? (stage == 2 ? 1 : 0) : (stage == 0 ? 1 : 2);       // X, Y, Z = 0, 1, 2
            const int phase = h == 0 ? (stage == 0 ? 1 : 2) : (stage == 2 ? 2 : 1);
            const int unit = wg * 3 + slot;
            if (unit < npairs) {
                const int b = unit / gpb, e0 = (unit % gpb) * WAVE, prt = AUM_SCANT_DBC_MERGE ? unit % gpb : (unit % gpb) * 2 + d;
                float* cy = wo.carry + ((int64_t)wg * 2 + d) * (2 * N + 2) * WAVE;
                const vi lane = lane_id();
                if (phase == 1) {
                    init(w);
                    scant_bwd_run<T, N, 1, SP, HAS_Z>(p, wo, b, e0, d, prt, d ? L - 1 + io : -io, d ? -1 : 1, scant_first_half(L, d) + io, L + io, d ? p.A_b : p.A, 2.f,
                                                      hh[AUM_W(w)], dA[AUM_W(w)], dD[AUM_W(w)], dbias[AUM_W(w)],
                                                      lds + w * scant_bwd_lds_wave_floats<T>(), tacc);
                    if (slot == 1) {        // Y: phase 2 runs on the other two waves
                        AUM_UNROLL
                        for (int n = 0; n < N; ++n) {
                            gstore(cy + n * WAVE, lane, vf16_get(hh[AUM_W(w)], n), lane >= 0);
                            gstore(cy + (N + n) * WAVE, lane, vf16_get(dA[AUM_W(w)], n), lane >= 0);
                        }
                        gstore(cy + 2 * N * WAVE, lane, dD[AUM_W(w)], lane >= 0);
                        gstore(cy + (2 * N + 1) * WAVE, lane, dbias[AUM_W(w)], lane >= 0);
                    }
                } else {
                    if (slot == 1) {
                        AUM_UNROLL
                        for (int n = 0; n < N; ++n) {
                            vf16_set(hh[AUM_W(w)], n, gload_u(cy + n * WAVE, lane));
                            vf16_set(dA[AUM_W(w)], n, gload_u(cy + (N + n) * WAVE, lane));
                        }
                        dD[AUM_W(w)] = gload_u(cy + 2 * N * WAVE, lane);
                        dbias[AUM_W(w)] = gload_u(cy + (2 * N + 1) * WAVE, lane);
                    }
                    scant_bwd_run<T, N, 2, SP, HAS_Z>(p, wo, b, e0, d, prt, d ? L - 1 + io : -io, d ? -1 : 1, io, scant_first_half(L, d) + io, d ? p.A_b : p.A, 2.f,
                                                      hh[AUM_W(w)], dA[AUM_W(w)], dD[AUM_W(w)], dbias[AUM_W(w)],
                                                      lds + w * scant_bwd_lds_wave_floats<T>(), tacc);
                    finish(w, unit, d);
                }
            }
        }
        if (stage < 2) AUM_WG_BARRIER();
    }
}

// Time segments of the backward (see scant_seg_fwd): segment s of a direction starts from the adjoint
//   h = P_{nseg-1} ( ... ) ... : h <- P_s' h + H_s' for s' = nseg - 1 down to s + 1 (PHASE 4 carries: H_s' = the adjoint at the segment's
// first step from zero behind its last), and leaves its partial sums of dA / dD / ddelta_bias in rows of their own: the reduce
// kernel sums batch * nseg rows per direction.  PHASE 0: one direction alone; 1 / 2: direction 0 / direction 1 of a Fo-Bi pair, each
// over the whole row, in two launches.
template <class T, int PHASE, bool SP, bool HAS_Z>
AUM_DEV void scant_seg_bwd(const AumScanTmBwdArgs& p, const ScanTBwdOut& wo, const ScanTSeg& sg, int wg, float* lds) {
    constexpr int N = SCANT_N;
    constexpr int NW = SCANT_NW;
    const int gpb = p.dim / WAVE;
    const int L = p.len;
    const int items = p.batch * gpb * sg.nseg;
    const bool bidir = p.A_b != nullptr;
    vf16 hh[AUM_PER_WAVE(NW)], dA[AUM_PER_WAVE(NW)];
    vf dD[AUM_PER_WAVE(NW)], dbias[AUM_PER_WAVE(NW)];
    AUM_FOR_EACH_WAVE(w, NW) {
        const int item = wg * NW + w;
        if (item < items) {
            const int d = sg.dir0, s = item % sg.nseg, unit = item / sg.nseg;
            const int b = unit / gpb, e0 = (unit % gpb) * WAVE, prt = bidir && !AUM_SCANT_DBC_MERGE ? (unit % gpb) * 2 + d : unit % gpb;
            const bool rev = bidir ? d == 1 : (p.flags & AUM_SCAN_REVERSE) != 0;
            const int it0 = s * sg.seg_len < L ? s * sg.seg_len : L;
            const int it1 = it0 + sg.seg_len < L ? it0 + sg.seg_len : L;
            const vi ec = lane_id() + e0;
            AUM_UNROLL
            for (int n = 0; n < N; ++n) {
                vf16_set(hh[AUM_W(w)], n, splat(0.f));
                vf16_set(dA[AUM_W(w)], n, splat(0.f));
            }
            dD[AUM_W(w)] = dbias[AUM_W(w)] = splat(0.f);
            const float* cb = sg.carry + ((int64_t)d * p.batch + b) * sg.nseg * (2 * N) * p.dim;
            for (int sp = sg.nseg - 1; sp > s; --sp) {
                const float* cs = cb + (int64_t)sp * (2 * N) * p.dim;
                AUM_UNROLL
                for (int n = 0; n < N; ++n)
                    vf16_set(hh[AUM_W(w)], n, vfma(gload_u(cs + (int64_t)n * p.dim, ec), vf16_get(hh[AUM_W(w)], n), gload_u(cs + (int64_t)(N + n) * p.dim, ec)));
            }
            scant_bwd_run<T, N, PHASE, SP, HAS_Z>(p, wo, b, e0, d, prt, rev ? L - 1 : 0, rev ? -1 : 1, it0, it1, d ? p.A_b : p.A, bidir ? 2.f : 1.f,
                                                  hh[AUM_W(w)], dA[AUM_W(w)], dD[AUM_W(w)], dbias[AUM_W(w)], lds + w * scant_bwd_lds_wave_floats<T>());
            const int64_t q = ((int64_t)d * p.batch + b) * sg.nseg + s;
            float* pa = wo.dA + q * N * p.dim;
            AUM_UNROLL
            for (int n = 0; n < N; ++n) gstore(pa + (int64_t)n * p.dim, ec, vf16_get(dA[AUM_W(w)], n), ec >= 0);
            gstore(wo.dD + q * p.dim, ec, dD[AUM_W(w)] * (bidir ? 2.f : 1.f), ec >= 0);
            gstore(wo.dbias + q * p.dim, ec, dbias[AUM_W(w)], ec >= 0);
        }
    }
}

}  // namespace aum
