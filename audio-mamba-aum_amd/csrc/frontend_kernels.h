// frontend_kernels.h -- waveform -> token sequence in ONE launch: log-mel frames (fbank_kernels.h, wave-per-frame form), the
// 16 x 16 patch embedding as an MFMA GEMM on the tile while it is still in LDS, + bias, + position rows, cls row in place.
//
// Reference chain replaced (src/ of the reference): dataloader.py:134-147, 206-228 (fbank, pad, SpecAug, normalise, noise, roll)
// -> models/mamba_models.py:509-541 (unsqueeze/transpose, patch_embed, cls token in the middle, + pos_embed) with
// utilities/tokenization.py:278-310 (FlexiPatchEmbed default branch: conv2d(kernel = stride = 16) -> flatten -> (B, N, Dm)).
// Neither the (B, 1024, 128) fp32 spectrogram nor an im2col copy of it reaches HBM; the only optional extra output is the
// 16-bit patch matrix the weight gradient needs (half the bytes of the spectrogram, written straight from the MFMA operands).
//
// Workgroup = 8 wavefronts = one clip x 64 output frames (4 time blocks) -> 8 x 4 = 32 tokens.
//   phase 1  wave w computes frames 8w .. 8w+7 (one frame per wave at a time, see fbank_frame_wave) and drops the 128 log-mel
//            values of each, rounded to the GEMM's 16-bit type, into tile[mel][frame] (pitch 72: 16-byte aligned rows).
//   phase 2  tokens are the MFMA COLUMNS (col = lane & 15 -> f_block = col & 7, t_block = col >> 3 [+2 for the second tile]),
//            embedding channels the rows: a lane's 4 accumulators are 4 consecutive channels of one token -> 16-byte stores.
//            k = 16 i + j (i: mel row inside the patch, j: frame inside the patch) is the flattened conv weight's own order,
//            so a k-group of 8 is 8 consecutive frames of one mel row = one ds_read_b128 from the tile.
//            Wave w owns the channel tiles c = w, w + 8, ...; the weight fragments stream from L2 (393 KB for Dm = 768).
#pragma once
#include "fbank_kernels.h"
#include "proj_kernels.h"

namespace aum {

constexpr int FT_NW = 8;
constexpr int FT_FRAMES = 64;
constexpr int FT_MEL = 128;
constexpr int FT_K = 256;                      // 16 x 16 patch
constexpr int FT_TPITCH = 72;
constexpr int FT_TILE = FT_MEL * FT_TPITCH;    // 16-bit elements
constexpr int FT_LDS_FLOATS = FT_NW * FBW_WAVE_FLOATS;
constexpr int FT_FPW = FT_FRAMES / FT_NW;

#ifndef AUM_EMU
template <class T> AUM_DEV vf vround16(T, vf x) { T e; f32_to_elem(x, e); return elem_to_f32(e); }
AUM_DEV void gload4_f32(const float* p, vi idx, vf (&o)[4]) {
    const f4v v = *reinterpret_cast<const f4v*>(p + idx);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
AUM_DEV void gstore4(float* p, vi idx, vf a, vf b, vf c, vf d) { *reinterpret_cast<f4v*>(p + idx) = f4v{a, b, c, d}; }
template <class T> AUM_DEV void gstore4(T* p, vi idx, vf a, vf b, vf c, vf d) { gstore_quad(p, idx, cvt4(T{}, a, b, c, d)); }
#else
template <class T> inline vf vround16(T, const vf& x) { vf r; AUM_LANES { T e; f32_to_elem(x.v[l], e); r.v[l] = elem_to_f32(e); } return r; }
inline void gload4_f32(const float* p, const vi& idx, vf (&o)[4]) { for (int j = 0; j < 4; ++j) AUM_LANES o[j].v[l] = p[idx.v[l] + j]; }
inline void gstore4(float* p, const vi& idx, const vf& a, const vf& b, const vf& c, const vf& d) {
    AUM_LANES { p[idx.v[l]] = a.v[l]; p[idx.v[l] + 1] = b.v[l]; p[idx.v[l] + 2] = c.v[l]; p[idx.v[l] + 3] = d.v[l]; }
}
template <class T> inline void gstore4(T* p, const vi& idx, const vf& a, const vf& b, const vf& c, const vf& d) {
    gstore_quad(p, idx, cvt4(T{}, a, b, c, d));
}
#endif

// T: 16-bit type of the GEMM operands (weight, patches); TO: element type of the token sequence
template <class T, class TO>
AUM_DEV void frontend_tokens_wg(const AumFrontendArgs& a, int wg, float* lds_f, uint16_t* tile) {
    const AumFbankArgs& fb = a.fbank;
    const int T_len = fb.target_length;
    const int nt = T_len / 16, nf = FT_MEL / 16, n_patches = nt * nf;
    const int blocks = T_len / FT_FRAMES;
    const int b = wg / blocks, tq = wg % blocks;
    const FbankAug ag = fbank_aug(fb, b);
    const bool has_cls = a.cls_row != nullptr;
    TO* tokens = reinterpret_cast<TO*>(a.tokens) + (int64_t)b * a.tokens_bs;
    // ---- phase 1: 64 frames -> tile[mel][frame]
    AUM_FOR_EACH_WAVE(w, FT_NW) {
        FbwTw tw;
        fbw_twiddles(fb, tw);
        for (int q = 0; q < FT_FPW; ++q) {
            const int tl = w * FT_FPW + q;
            int frame = (tq * FT_FRAMES + tl - ag.roll) % T_len;          // out[(frame + roll) mod T] = in[frame]
            if (frame < 0) frame += T_len;
            fbank_frame_wave(fb, b, frame, ag, tw, lds_f + w * FBW_WAVE_FLOATS,
                             [&](vi m, vf v, vm ok) { lds16_write1(tile, m * FT_TPITCH + tl, f32_to_bits16(T{}, v)); });
        }
        if (has_cls && tq == 0) {                                        // the cls row of this clip: cls_token + pos_embed[0]
            for (int d0 = w * WAVE; d0 < a.dim; d0 += FT_NW * WAVE) {
                const vi d = lane_id() + d0;
                const vm ok = d < a.dim;
                gstore(tokens, a.cls_pos * a.dim + d, gload(a.cls_row, d, ok), ok);
            }
        }
    }
    AUM_WG_BARRIER();
    // ---- phase 2: tokens[32][dim] = patches[32][256] x W^T, + bias, rounded as the 16-bit conv output, + position row
    const T* W = reinterpret_cast<const T*>(a.weight);
    T* patches = a.patches ? reinterpret_cast<T*>(a.patches) + (int64_t)b * n_patches * FT_K : nullptr;
    AUM_FOR_EACH_WAVE(w, FT_NW) {
        const vi lane = lane_id();
        const vi col = lane & 15, g = lane >> 4;
        const vi fblk = col & 7;
        frag8 P[2][8];
        vi cell[2], seq[2];
        AUM_UNROLL
        for (int tt = 0; tt < 2; ++tt) {
            const vi tb = (col >> 3) + 2 * tt;                           // time block inside the workgroup's 64 frames
            const vi tbg = tb + tq * (FT_FRAMES / 16);
            cell[tt] = fblk * nt + tbg;                                  // the model's token index f * n_t + t (MM:516)
            vi sq = (a.flags & AUM_FRONTEND_TIME_MAJOR) ? tbg * nf + fblk : cell[tt];
            seq[tt] = has_cls ? vsel_i(sq >= a.cls_pos, sq + 1, sq) : sq;
            AUM_UNROLL
            for (int s = 0; s < 8; ++s) {
                const vi row = fblk * 16 + (g >> 1) + 2 * s;
                P[tt][s] = lds16_read8_a(tile, row * FT_TPITCH + tb * 16 + (g & 1) * 8);
            }
        }
        if (patches) {                                                   // k-step w of both token tiles: 16-byte stores
            AUM_UNROLL
            for (int s = 0; s < 8; ++s) {
                if (s != w % 8) continue;
                AUM_UNROLL
                for (int tt = 0; tt < 2; ++tt) gstore_frag(patches, cell[tt] * FT_K + 32 * s + g * 8, P[tt][s]);
            }
        }
        for (int c = w; c < a.dim / 16; c += FT_NW) {
            acc4 acc[2] = {acc_zero(), acc_zero()};
            AUM_UNROLL
            for (int s = 0; s < 8; ++s) {
                const frag8 wf = gload_frag(W, (col + c * 16) * FT_K + 32 * s + g * 8);     // A operand: row = channel
                acc[0] = mfma16(T{}, wf, P[0][s], acc[0]);
                acc[1] = mfma16(T{}, wf, P[1][s], acc[1]);
            }
            const vi dm = g * 4 + c * 16;
            vf bias[4];
            gload4_f32(a.bias, dm, bias);
            AUM_UNROLL
            for (int tt = 0; tt < 2; ++tt) {
                vf pe[4], o[4];
                gload4_f32(a.pos, cell[tt] * a.dim + dm, pe);
                AUM_UNROLL
                for (int r = 0; r < 4; ++r) o[r] = vround16(T{}, acc_get(acc[tt], r) + bias[r]) + pe[r];
                gstore4(tokens, seq[tt] * a.dim + dm, o[0], o[1], o[2], o[3]);
            }
        }
    }
}

}  // namespace aum
