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
//   * B_t and C_t are the same for all 64 channels: a wave fetches the rows of its next 8 steps with one coalesced access, parks
//     them as fp32 in a 2 KB LDS strip of its own and reads a step's 32 values back as broadcasts (no barrier: nobody else reads it).
//   * addresses: one buffer descriptor per tensor; the per-lane byte offset is a constant VGPR, the row offset an SGPR cursor that
//     moves by one stride per step (one s_add).  No 64-bit address arithmetic on either ALU.
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
// row loads, 4 no stores, 8 no activation loads
#ifndef AUM_SCANT_ABL
#define AUM_SCANT_ABL 0
#endif
constexpr int SCANT_N = 16;                    // states per channel (Mamba's d_state; the only instantiation)
constexpr int SCANT_CK = AUM_SCAN_TM_CK;       // steps per checkpoint block
constexpr int SCANT_G = SCANT_CK / 2;          // steps per prefetch group (two groups in flight, ping-pong)
AUM_HOSTDEV constexpr int scant_nblocks(int len) { return (len + SCANT_CK - 1) / SCANT_CK; }
AUM_HOSTDEV constexpr int scant_nck(int len) { return scant_nblocks(len) - 1; }        // the last block's exit state is never needed
AUM_HOSTDEV bool scant_supported(int dim, int dstate) { return dstate == SCANT_N && dim % WAVE == 0; }

// inputs of one step as loaded (widened where they are used, so that the wait for a prefetch sits at the use)
struct ScanTRaw { vi u, d, z, part; };

// B_t / C_t of a block of SCANT_CK steps, staged by the wave for itself: [step][B_0..B_{N-1} | C_0..C_{N-1}] fp32 in LDS, two blocks
constexpr int SCANT_BC_ROW = 2 * SCANT_N;                          // floats per step
constexpr int SCANT_BC_BLOCK = SCANT_CK * SCANT_BC_ROW;            // floats per block
constexpr int SCANT_LDS_WAVE_FLOATS = 2 * SCANT_BC_BLOCK;          // per wave (2 KB)

// One direction of one channel group over scan-order iterations [it0, it1) of the row; iteration `it` is step
// t = t0 + it * tstep (tstep = +1 forward time, -1 reverse time).
// PHASE 0: the whole direction alone -- out = gate * (y + D u), out_pre = y + D u.
// PHASE 1: first half of a direction pair -- out <- partial y (no skip, no gate).
// PHASE 2: second half -- tot = y + partial(out) + dmul D u;  out_pre <- tot;  out <- gate * tot.
// Blocks of SCANT_CK steps aligned to multiples of SCANT_CK in `it`.  Activations: the inputs of half a block are in flight while the
// other half is computed (two register sets, ping-pong).  B_t / C_t: the rows of the NEXT block are fetched by the wave as one
// coalesced access at the top of a block (lane = (step, state pair)), widened to fp32 and parked in the wave's own LDS strip half a
// block later; a step reads its 32 values back with eight wave-uniform ds_read_b128 (every lane the same address: a broadcast).
// (Scalar-cache loads into SGPRs were the first design: with one row per step they cost 0.10 of 0.37 ms -- s_waitcnt can only wait
// for ALL outstanding scalar loads, the row latency exceeds a step, and 32 live SGPR rows pushed the kernel into SGPR spills.)
// Runs of blocks that lie inside the phase together with the block after them take the fast body: row offsets are cursors advanced
// by one stride per step and nothing is conditional.  The ragged blocks at the ends of a phase take the same body with clamped row
// indices and per-step conditions.
template <class T, int N, int PHASE, bool SP, bool HAS_Z, bool HAS_PRE>
AUM_DEV void scant_fwd_run(const AumScanTmFwdArgs& p, int b, int e0, int dir, int t0, int tstep, int it0, int it1, const float* Aptr,
                           float dmul, vf2 (&x)[N / 2], float* lds) {
    constexpr int ES = (int)sizeof(T);
    static_assert(N == 16 && SCANT_CK == 8, "lane = (step, state pair) staging below assumes 8 steps x 8 pairs");
    const int L = p.len;
    const vi lane = lane_id();
    const vi ec = lane + e0;                     // dim % 64 == 0: every lane is a channel
    const vi vo = ec * ES, vo4 = ec * 4;         // per-lane byte offsets into a row of T / of float
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
    // byte strides per step of time / per iteration
    const int u_tb = (int)p.u_ts * ES, d_tb = (int)p.delta_ts * ES, z_tb = HAS_Z ? (int)p.z_ts * ES : 0, o_tb = (int)p.out_ts * ES,
              p_tb = HAS_PRE ? (int)p.pre_ts * ES : 0, B_tb = (int)p.B_ts * ES, C_tb = (int)p.C_ts * ES;
    const int su = tstep * u_tb, sd = tstep * d_tb, sz = tstep * z_tb, so = tstep * o_tb, spre = tstep * p_tb;
    auto tok = [&](int it) { return t0 + it * tstep; };
    auto clamp_it = [&](int it) { return it < it0 ? it0 : (it < it1 ? it : it1 - 1); };
    // B/C staging: lane -> (memory row r of the block, state pair j); the row's iteration inside the block is r (forward time) or
    // 7 - r (reverse time), so per-lane offsets are never negative
    const vi st_r = lane >> 3, st_j = lane & 7;
    const vi st_i = tstep > 0 ? st_r : (SCANT_CK - 1) - st_r;
    const vi st_slot = st_i * SCANT_BC_ROW + st_j * 2;                  // LDS word of B pair j of that step; C pair at + N
    const vi st_voB = st_r * B_tb + st_j * (2 * ES), st_voC = st_r * C_tb + st_j * (2 * ES);

    auto load_at = [&](int uo, int dofs, int zo, int oo, ScanTRaw& r) {
        if (AUM_SCANT_ABL & 8) {
            r.u = r.d = r.z = r.part = spl_i(0x3f80);
            return;
        }
        r.u = gbuf_load_raw(ubuf, vo, uo);
        r.d = gbuf_load_raw(dbuf, vo, dofs);
        if (PHASE != 1 && HAS_Z) r.z = gbuf_load_raw(zbuf, vo, zo);
        if (PHASE == 2) r.part = gbuf_load_raw(obuf, vo, oo);
    };
    // B of a step: four broadcast reads of its LDS row
    auto read_B = [&](const float* bcrow, vf2 (&Bp)[N / 2]) {
        AUM_UNROLL
        for (int k = 0; k < N / 4; ++k) {
            vf q[4];
            if (AUM_SCANT_ABL & 2) q[0] = q[1] = q[2] = q[3] = splat(1.f);
            else lds_read4_u(bcrow, 4 * k, q);
            Bp[2 * k] = mk2(q[0], q[1]);
            Bp[2 * k + 1] = mk2(q[2], q[3]);
        }
    };
    // one step.  Bp: this step's B (read from LDS during the previous step); Bn <- the next step's B, requested here together
    // with this step's C, ahead of the arithmetic that hides both round trips
    auto step_at = [&](const ScanTRaw& r, const float* bcrow, const float* bcrow_next, const vf2 (&Bp)[N / 2], vf2 (&Bn)[N / 2], int oo,
                       int po) {
        vf2 Cp[N / 2];
        AUM_UNROLL
        for (int k = 0; k < N / 4; ++k) {
            vf q[4];
            if (AUM_SCANT_ABL & 2) q[0] = q[1] = q[2] = q[3] = splat(1.f);
            else lds_read4_u(bcrow, N + 4 * k, q);
            Cp[2 * k] = mk2(q[0], q[1]);
            Cp[2 * k + 1] = mk2(q[2], q[3]);
        }
        read_B(bcrow_next, Bn);
        AUM_SCHED_FENCE();
        const vf uu = raw_to_f32<T>(r.u);
        vf dl = raw_to_f32<T>(r.d) + biasv;
        if (SP) dl = vsoftplus(dl);
        const vf du = dl * uu;
        const vf2 dl2 = spl2(dl), du2 = spl2(du);
        AUM_UNROLL
        for (int j = 0; j < N / 2; ++j) {       // states in pairs (2j, 2j+1)
            const vf2 e = dl2 * A2[j];
            const vf2 a = (AUM_SCANT_ABL & 1) ? e : vexp2_2(e);
            x[j] = vfma2(a, x[j], du2 * Bp[j]);
        }
        vf2 y2[2] = {spl2(splat(0.f)), spl2(splat(0.f))};
        AUM_UNROLL
        for (int j = 0; j < N / 2; ++j) y2[j & 1] = vfma2(x[j], Cp[j], y2[j & 1]);
        const vf2 ysum = y2[0] + y2[1];
        const vf ys = lo2(ysum) + hi2(ysum);
        if (PHASE == 1) {
            if (!(AUM_SCANT_ABL & 4)) gbuf_store(obuf, vo, oo, ys);
            return;
        }
        vf tot = vfma(uu, Dv, ys);
        if (PHASE == 2) tot = tot + raw_to_f32<T>(r.part);
        if (HAS_PRE && !(AUM_SCANT_ABL & 4)) gbuf_store(pbuf, vo, po, tot);
        if (HAS_Z) {
            const vf zz = raw_to_f32<T>(r.z);
            tot = tot * (zz * vsigmoid(zz));
        }
        if (!(AUM_SCANT_ABL & 4)) gbuf_store(obuf, vo, oo, tot);
    };
    auto ckpt_store = [&](int blk) {
        int off = blk * N * p.dim * 4;
        AUM_UNROLL
        for (int n = 0; n < N; ++n) {
            gbuf_store(ckbuf, vo4, off, (n & 1) ? hi2(x[n >> 1]) : lo2(x[n >> 1]));
            off += p.dim * 4;
        }
    };
    // rows of block `blk` -> registers.  fast: all 8 iterations inside the phase, one cursor; else per-lane clamped rows
    struct BCRegs { vf b0, b1, c0, c1; };
    auto bc_load_fast = [&](int blk, BCRegs& g) {
        const int base = blk * SCANT_CK;
        const int t_lo = tstep > 0 ? tok(base) : tok(base + SCANT_CK - 1);
        gbuf_load_pair(Bbuf, st_voB, t_lo * B_tb, g.b0, g.b1);
        gbuf_load_pair(Cbuf, st_voC, t_lo * C_tb, g.c0, g.c1);
    };
    auto bc_load_slow = [&](int blk, BCRegs& g) {
        const int base = blk * SCANT_CK;
        vi it = st_i + base;
        it = vmax_i(vmin_i(it, it1 - 1), it0);
        const vi t = it * tstep + t0;
        gbuf_load_pair(Bbuf, t * B_tb + st_j * (2 * ES), 0, g.b0, g.b1);
        gbuf_load_pair(Cbuf, t * C_tb + st_j * (2 * ES), 0, g.c0, g.c1);
    };
    auto bc_stage = [&](int blk, const BCRegs& g) {
        float* dst = lds + (blk & 1) * SCANT_BC_BLOCK;
        lds_write2(dst, st_slot, g.b0, g.b1);
        lds_write2(dst, st_slot + N, g.c0, g.c1);
        wave_lds_fence();
    };
    auto blk_inside = [&](int blk) { return blk * SCANT_CK >= it0 && blk * SCANT_CK + SCANT_CK <= it1; };
    auto is_fast = [&](int blk) { return blk_inside(blk) && blk_inside(blk + 1); };

    if (it0 >= it1) return;
    ScanTRaw ra[SCANT_G], rb[SCANT_G];
    BCRegs bcn;                          // rows of the next block, between their load and their staging
    const int blk0 = it0 / SCANT_CK, blk1 = (it1 + SCANT_CK - 1) / SCANT_CK;
    // prologue of the phase: the first half block and the first block's B/C rows
    AUM_UNROLL
    for (int s = 0; s < SCANT_G; ++s) {
        const int t = tok(clamp_it(blk0 * SCANT_CK + s));
        load_at(t * u_tb, t * d_tb, t * z_tb, t * o_tb, ra[s]);
    }
    bc_load_slow(blk0, bcn);
    bc_stage(blk0, bcn);
    // the ragged block `blk`: row offsets from clamped iteration numbers, every step conditional
    auto slow_block = [&](int blk) {
        const int base = blk * SCANT_CK;
        const float* cur = lds + (blk & 1) * SCANT_BC_BLOCK;
        auto load_half = [&](int first, ScanTRaw (&r)[SCANT_G]) {
            AUM_UNROLL
            for (int s = 0; s < SCANT_G; ++s) {
                const int t = tok(clamp_it(first + s));
                load_at(t * u_tb, t * d_tb, t * z_tb, t * o_tb, r[s]);
            }
        };
        auto step_half = [&](int first, const ScanTRaw (&r)[SCANT_G]) {
            AUM_UNROLL
            for (int s = 0; s < SCANT_G; ++s) {
                const int it = first + s;
                if (it >= it0 && it < it1) {
                    const int t = tok(it);
                    const float* row = cur + (it - base) * SCANT_BC_ROW;
                    vf2 Bp[N / 2], Bn[N / 2];
                    read_B(row, Bp);
                    step_at(r[s], row, row, Bp, Bn, t * o_tb, t * p_tb);
                }
            }
        };
        if (blk + 1 < blk1) bc_load_slow(blk + 1, bcn);
        load_half(base + SCANT_G, rb);
        step_half(base, ra);
        if (blk + 1 < blk1) bc_stage(blk + 1, bcn);
        load_half(base + SCANT_CK, ra);
        step_half(base + SCANT_G, rb);
        // the block's exit state is complete only in the phase that ran its last step
        if (want_ck && blk < nck && base + SCANT_CK - 1 >= it0 && base + SCANT_CK - 1 < it1) ckpt_store(blk);
    };

    int blk = blk0;
    while (blk < blk1) {
        if (!is_fast(blk)) {
            slow_block(blk);
            ++blk;
            continue;
        }
        // a run of fast blocks: cursors persist across them.  Loads run half a block ahead of the steps, B/C rows a block ahead.
        const int base = blk * SCANT_CK;
        const int tl = tok(base + SCANT_G), ts = tok(base);
        int lu = tl * u_tb, ld = tl * d_tb, lz = tl * z_tb, lo = tl * o_tb;
        int co = ts * o_tb, cp = ts * p_tb;
        auto load_half = [&](ScanTRaw (&r)[SCANT_G]) {
            AUM_UNROLL
            for (int s = 0; s < SCANT_G; ++s) {
                load_at(lu, ld, lz, lo, r[s]);
                lu += su;
                ld += sd;
                lz += sz;
                lo += so;
            }
        };
        vf2 Bq[N / 2];
        read_B(lds + (blk & 1) * SCANT_BC_BLOCK, Bq);
        // Bq: B of the next step to compute (requested a step ahead).  `rows`: the four rows of this half; `after`: the row that
        // follows them (the other half of this block, or the first row of the next block -- staged before this half began)
        auto step_half = [&](const float* rows, const float* after, const ScanTRaw (&r)[SCANT_G]) {
            AUM_UNROLL
            for (int s = 0; s < SCANT_G; ++s) {
                vf2 Bn[N / 2];
                step_at(r[s], rows + s * SCANT_BC_ROW, s + 1 < SCANT_G ? rows + (s + 1) * SCANT_BC_ROW : after, Bq, Bn, co, cp);
                AUM_UNROLL
                for (int j = 0; j < N / 2; ++j) Bq[j] = Bn[j];
                co += so;
                cp += spre;
            }
        };
        do {
            const float* cur = lds + (blk & 1) * SCANT_BC_BLOCK;
            const float* nxt = lds + ((blk + 1) & 1) * SCANT_BC_BLOCK;
            bc_load_fast(blk + 1, bcn);
            load_half(rb);
            step_half(cur, cur + SCANT_G * SCANT_BC_ROW, ra);
            bc_stage(blk + 1, bcn);
            load_half(ra);
            step_half(cur + SCANT_G * SCANT_BC_ROW, nxt, rb);
            if (want_ck && blk < nck) ckpt_store(blk);
            ++blk;
        } while (blk < blk1 && is_fast(blk));
    }
}

// iterations the forward-time wave (dir 0) and the reverse-time wave (dir 1) of a pair run before they meet: together they cover
// every step exactly once
AUM_HOSTDEV constexpr int scant_first_half(int len, int dir) { return dir == 0 ? len / 2 : len - len / 2; }

// workgroup = four waves, one per SIMD: two channel groups x two directions (BIDIR: even wave = A forward time, odd wave = A_b reverse
// time of the same group) or four channel groups.  Units (batch entry, channel group) are numbered batch-major.
constexpr int SCANT_NW = 4;
template <bool BIDIR> AUM_HOSTDEV constexpr int scant_units_per_wg() { return BIDIR ? SCANT_NW / 2 : SCANT_NW; }

template <class T, bool SP, bool HAS_Z, bool HAS_PRE, bool BIDIR>
AUM_DEV void scant_fwd(const AumScanTmFwdArgs& p, int wg, float* lds) {
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
                                                           x[AUM_W(w)], lds + w * SCANT_LDS_WAVE_FLOATS);
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
                                                       d ? p.A_b : p.A, 2.f, x[AUM_W(w)], lds + w * SCANT_LDS_WAVE_FLOATS);
        }
    }
    AUM_WG_BARRIER();      // also orders this workgroup's partial stores before the other wave's loads of them (same CU, same L2)
    AUM_FOR_EACH_WAVE(w, NW) {
        const int unit = wg * UPW + (w >> 1), d = w & 1;
        if (unit < units)
            scant_fwd_run<T, N, 2, SP, HAS_Z, HAS_PRE>(p, unit / gpb, (unit % gpb) * WAVE, d, d ? L - 1 : 0, d ? -1 : 1, scant_first_half(L, d), L,
                                                       d ? p.A_b : p.A, 2.f, x[AUM_W(w)], lds + w * SCANT_LDS_WAVE_FLOATS);
    }
}

}  // namespace aum
