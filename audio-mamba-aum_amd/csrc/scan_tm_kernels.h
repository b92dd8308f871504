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
// row loads, 4 no stores, 8 no activation loads, 16 no per-step LDS reads, 32 no per-step LDS writes
#ifndef AUM_SCANT_ABL
#define AUM_SCANT_ABL 0
#endif
constexpr int SCANT_N = 16;                    // states per channel (Mamba's d_state; the only instantiation)
constexpr int SCANT_CK = AUM_SCAN_TM_CK;       // steps per checkpoint block
constexpr int SCANT_G = SCANT_CK / 2;          // steps per prefetch group (two groups in flight, ping-pong)
AUM_HOSTDEV constexpr int scant_nblocks(int len) { return (len + SCANT_CK - 1) / SCANT_CK; }
AUM_HOSTDEV constexpr int scant_nck(int len) { return scant_nblocks(len) - 1; }        // the last block's exit state is never needed
AUM_HOSTDEV bool scant_supported(int dim, int dstate) { return dstate == SCANT_N && dim % WAVE == 0; }

// ------------------------------------------------------------------------------------------------
// Block staging.  A wave moves the 8 steps x 64 channels of a block between HBM and a 1 KB (16-bit) / 2 KB (fp32) LDS tile as
// 16 bytes per lane (lane = (row r of the block in memory order, 16-byte chunk c of the row): one buffer_load/store_dwordx4 per
// tensor and block instead of a 2-byte access per tensor and STEP), and reads / writes its own channel's element of one step
// with a 2- or 4-byte LDS access.  Row i of a tile is iteration base + i of the block: memory row r is i = r (forward time) or
// 7 - r (reverse time), so per-lane global offsets are never negative.
// ------------------------------------------------------------------------------------------------
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
                           float dmul, vf2 (&x)[N / 2], float* lds, unsigned long long* scant_trace_acc = nullptr) {
    using TL = ScanTTile<T>;
    constexpr int ES = TL::ES, ROWB = TL::ROWB, NLD = TL::NLD;
    static_assert(N == 16 && SCANT_CK == 8, "lane = (row, chunk) staging below assumes 8 steps x 8 chunks / state pairs");
    constexpr bool LD_Z = HAS_Z && PHASE != 1, LD_PART = PHASE == 2, ST_PRE = HAS_PRE && PHASE != 1;
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
    const gbuf<float> ckbuf = make_gbuf(want_ck ? p.ckpt + ((int64_t)dir * p.batch + b) * nck * N * p.dim : (const float*)Aptr);
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
        if (AUM_SCANT_ABL & 4) return;
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
    auto step = [&](int s, const float* bc, bool prefetch, const ScanTRaw& r, vf dl, vf2 (&Bp)[N / 2], ScanTRaw& rn, vf& dln, vf2 (&Bn)[N / 2]) {
        const float* bcrow = bc + s * SCANT_BC_ROW;
        vf2 Cp[N / 2];
        AUM_UNROLL
        for (int k = 0; k < N / 4; ++k) {
            vf q[4];
            if (AUM_SCANT_ABL & 16) q[0] = q[1] = q[2] = q[3] = splat(1.f);
            else lds_read4_u(bcrow, N + 4 * k, q);
            Cp[2 * k] = mk2(q[0], q[1]);
            Cp[2 * k + 1] = mk2(q[2], q[3]);
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
    auto ckpt_store = [&](int blk) {
        int off = blk * N * p.dim * 4;
        AUM_UNROLL
        for (int n = 0; n < N; ++n) {
            gbuf_store(ckbuf, vo4, off, (n & 1) ? hi2(x[n >> 1]) : lo2(x[n >> 1]));
            off += p.dim * 4;
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
    // Three waves share a SIMD and the arbiter serves equal priorities oldest first: left alone, one wave runs at full speed and
    // finishes at 0.55 of the kernel's time, the second at 0.75, and the last runs the final quarter alone at less than half the
    // vector ALU's rate (measured: wave durations 185 / 254 / 325 us on every SIMD).  Each wave therefore walks through the
    // priorities 2, 1, 0 block by block, staggered by its slot number on the SIMD, so the three take turns and finish together.
    const int wslot = AUM_SCANT_PRIO ? wave_slot_on_simd() : 0;
    for (int blk = blk0; blk < blk1; ++blk) {
        const int base = blk * SCANT_CK;
        const bool more = blk + 1 < blk1;
        if (AUM_SCANT_PRIO) {
            const int turn = (blk + wslot) % 3;
            if (turn == 0) AUM_SET_PRIO(2);
            else if (turn == 1) AUM_SET_PRIO(1);
            else AUM_SET_PRIO(0);
        }
        if (more) request(blk + 1, nx);
        AUM_TM_STAMP(1);
        const float* bc = t_bc + (blk & 1) * SCANT_BC_BLOCK;
        if (base >= it0 && base + SCANT_CK <= it1) {
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
            if (want_ck && blk < nck) ckpt_store(blk);
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
            if (want_ck && blk < nck && base + SCANT_CK - 1 >= it0 && base + SCANT_CK - 1 < it1) ckpt_store(blk);
        }
        AUM_TM_STAMP(2);
        flush(blk);
        AUM_TM_STAMP(3);
        if (more) park(blk + 1, nx);
        AUM_TM_STAMP(4);
    }
#undef AUM_TM_STAMP
}

// iterations the forward-time wave (dir 0) and the reverse-time wave (dir 1) of a pair run before they meet: together they cover
// every step exactly once
AUM_HOSTDEV constexpr int scant_first_half(int len, int dir) { return dir == 0 ? len / 2 : len - len / 2; }

// workgroup = four waves, one per SIMD: two channel groups x two directions (BIDIR: even wave = A forward time, odd wave = A_b reverse
// time of the same group) or four channel groups.  Units (batch entry, channel group) are numbered batch-major.
constexpr int SCANT_NW = 4;
template <bool BIDIR> AUM_HOSTDEV constexpr int scant_units_per_wg() { return BIDIR ? SCANT_NW / 2 : SCANT_NW; }

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
        const int unit = wg * UPW + (w >> 1), d = w & 1;
        if (unit < units) {
            AUM_UNROLL
            for (int j = 0; j < N / 2; ++j) x[AUM_W(w)][j] = spl2(splat(0.f));
            scant_fwd_run<T, N, 1, SP, HAS_Z, HAS_PRE>(p, unit / gpb, (unit % gpb) * WAVE, d, d ? L - 1 : 0, d ? -1 : 1, 0, scant_first_half(L, d),
                                                       d ? p.A_b : p.A, 2.f, x[AUM_W(w)], lds + w * scant_lds_wave_floats<T>(), tacc);
        }
    }
    AUM_WG_BARRIER();      // also orders this workgroup's partial stores before the other wave's loads of them (same CU, same L2)
    AUM_FOR_EACH_WAVE(w, NW) {
        const int unit = wg * UPW + (w >> 1), d = w & 1;
        if (unit < units)
            scant_fwd_run<T, N, 2, SP, HAS_Z, HAS_PRE>(p, unit / gpb, (unit % gpb) * WAVE, d, d ? L - 1 : 0, d ? -1 : 1, scant_first_half(L, d), L,
                                                       d ? p.A_b : p.A, 2.f, x[AUM_W(w)], lds + w * scant_lds_wave_floats<T>(), tacc);
    }
}

}  // namespace aum
