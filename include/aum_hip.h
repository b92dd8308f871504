/*
 * aum_hip.h -- C ABI of libaum_hip.so: the MI355X (gfx950) implementation of the native boundary that
 * kaistmm/Audio-Mamba-AuM reaches through two un-vendored CUDA wheels.
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   selective_scan_cuda.fwd / .bwd            call sites vim-mamba_ssm/mamba_ssm/ops/selective_scan_interface.py:37,
 *                                             62-65, 213-215, 247-251, 354-356, 389-393, 499-505, 541-552
 *   causal_conv1d_cuda.causal_conv1d_fwd/_bwd call sites selective_scan_interface.py:177, 239, 281-283, 318, 380,
 *                                             425-427, 463, 532, 594-596 (arithmetic: modules/mamba_simple.py:272)
 *   Triton _layer_norm_fwd_1pass_kernel / _layer_norm_bwd_kernel (fused residual-add + RMSNorm)
 *                                             vim-mamba_ssm/mamba_ssm/ops/triton/layernorm.py:123-177, 293-377
 *
 * Conventions (same as the reference's extension ABI, SURVEY.md 8b "Conventions"):
 *   - every pointer is a DEVICE pointer; nothing is allocated, freed or synchronised inside the library;
 *     every call only enqueues kernels on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - activation tensors (u, delta, z, B, C, x, dout, ...) share one element type `dtype`;
 *     A, D, delta_bias, conv weights/bias, norm weights and every parameter gradient are fp32;
 *   - the time axis is always unit-stride; batch / channel strides are explicit, in ELEMENTS;
 *   - return value 0 on success, a negative AUM_E_* code on invalid arguments (no exceptions cross the ABI).
 */
#ifndef AUM_HIP_H
#define AUM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AUM_ABI_VERSION 13  /* 2: x_ck (chunk-entry state checkpoint) appended to the two scan argument structs;
                               3: x_lane (lane-entry state checkpoint of the L = 513 row kernels) appended after it;
                               4: aug / noise (per-clip augmentation in the log-mel kernel's epilogue) appended to AumFbankArgs;
                               5: aum_frontend_tokens_fwd (waveform -> token sequence in one launch);
                               6: aum_sum_rows (fixed-order sum of partial results);
                               7: aum_scan_tm_fwd / _bwd (time-serial selective scan on token-major activations);
                               8: aum_conv1d_tm_fwd / _bwd (the causal conv on token-major activations);
                               9: aum_gemm_tn (the dense in_proj / out_proj GEMMs on token-major activations); aum_dtproj_tm_fwd, aum_xdt_tm_fwd; aum_scan_tm_ckpt_rows
                                  (packed 16-bit state checkpoints of the token-major scan for 16-bit activations);
                               10: aum_gemm_wgrad (weight gradients of the projections), aum_scan_tm_seg_fwd / _bwd (time segments: long rows at a small
                                  batch), aum_causal_conv1d_update / aum_selective_state_update (streaming inference);
                               11: aum_sum_rows_multi (several partial sets summed in one launch);
                               12: experiment switches and entry points removed from the boundary (AUM_GEMM_STAGGERED / _PERSISTENT / _NO_COUNTED_WAIT /
                                   _NO_PREFETCH / _W4 / _RING, aum_gemm_tn_sk, aum_gemm_tn_sk_workspace_bytes, aum_scan_tm_bwd_matrix_sums); AUM_GEMM_PACED;
                                   aum_sum_rows / aum_sum_rows_multi take any float address as destination;
                               13: aum_cast_bank (the 16-bit copies -- and transposes -- of a group of fp32 master weights in one launch);
                                   aum_rmsnorm_bwd_partial_rows (the vectorised norm backward leaves an eighth of the partial rows) */

enum { AUM_F32 = 0, AUM_BF16 = 1, AUM_F16 = 2 };

enum {
    AUM_OK = 0,
    AUM_E_NULL = -1,        /* a required pointer is NULL */
    AUM_E_SHAPE = -2,       /* non-positive or unsupported size */
    AUM_E_DTYPE = -3,       /* unknown dtype enum */
    AUM_E_UNSUPPORTED = -4, /* combination not implemented (e.g. dstate > 256) */
    AUM_E_WORKSPACE = -5,   /* workspace missing or too small */
    AUM_E_LAUNCH = -6       /* hipLaunchKernel reported an error */
};

/* flags */
#define AUM_SCAN_SOFTPLUS 1u /* delta = softplus(delta + delta_bias)  (delta_softplus=True)                     */
#define AUM_SCAN_REVERSE 2u  /* run the recurrence from t=len-1 down to 0 (replaces the .flip([-1]) copies of   */
                             /* selective_scan_interface.py:503-507,547-561 and mamba_simple.py:229-246)        */
#define AUM_SCAN_GENERIC 4u  /* force the generic single-wave kernels (any dstate <= 256); default: 8-wave workgroup */
#define AUM_SCAN_ROWPAIR 8u  /* debug / A-B: keep the general row-pair kernels where a specialised one would run (one-row backward for
                                 512 + tail rows, chunked kernels for rows of 512*m + 1 steps, forward and backward) */
                             /* kernels when dstate <= 16                                                        */
#define AUM_SCAN_ACCUMULATE 16u /* long rows (the chunked kernels; anything else: AUM_E_UNSUPPORTED): add to out (forward) / du, ddelta, dz
                                  (backward) instead of overwriting them -- the second direction of a bidirectional layer lands on
                                  the first one's tensors, replacing the element-wise sums of selective_scan_interface.py:507,554-559 */
#define AUM_CONV_SILU 1u
#define AUM_CONV_REVERSE 2u  /* anti-causal: y[l] = act(b + sum_w W[w] x[l+(W-1)-w])                            */
#define AUM_CONV_GENERIC 4u  /* force the any-width kernel (default: the vectorised width-4 kernel when width == 4)  */
/* aum_gemm_tn picks its kernel by the length of the K loop (the paced-store kernel from k = 448 on, else one workgroup per tile); these name one: */
#define AUM_GEMM_LOCKSTEP 1u        /* one workgroup per 256 x 256 tile, one barrier per K-step */
#define AUM_GEMM_PIPELINED 32u      /* one workgroup per tile, fragment reads of the next half K-step under the current half's MFMAs */
#define AUM_GEMM_PACED 256u         /* one workgroup per CU walking a tile list as one stream of K-steps (hand-placed schedule, AGPR accumulators); a tile's
                                       stores leave under the next tile's first K-steps; k >= 448 (else AUM_E_UNSUPPORTED) */
#define AUM_NORM_PRENORM 1u  /* also return residual_out                                                        */
#define AUM_NORM_GENERIC 2u  /* force the any-cols kernel (default: register-cached vector kernel, cols <= 2048)  */

/*
 * Selective scan forward (selective_scan_cuda.fwd).
 *   u, delta, z, out, out_pre : (batch, dim, len)        B, C : (batch, dstate, len)   [the G=1 (B,1,N,L) case]
 *   A, A_b : (dim, dstate) fp32    D, delta_bias : (dim) fp32 or NULL     last_state : (batch, dim, dstate) fp32 or NULL
 * A_b != NULL selects the direction-fused bidirectional form of BiMambaInnerFn.forward
 * (selective_scan_interface.py:499-507): out = scan(A, forward) + scan(A_b, time-reversed), inputs read once, no flip
 * copies.  z == NULL: no gate.  out_pre (optional) receives the pre-gate value y + D*u summed over directions (what
 * the reference saves as `out` / `out_f`,`out_b` for the backward's dz).
 * workspace: unused by the forward (kept for ABI symmetry).
 * x_ck (optional, output): the `x` checkpoint tensor of selective_scan_cuda.fwd in this library's chunking -- the state
 * entering every 512-step chunk in scan order, (batch, dim, len/512, dstate) fp32.  Only rows the chunked kernels take
 * (aum_selective_scan_ckpt_bytes(...) > 0, one direction, neither AUM_SCAN_GENERIC nor AUM_SCAN_ROWPAIR) have one;
 * passing it for any other call is AUM_E_UNSUPPORTED.  Handed to the backward it replaces the backward's own pre-pass.
 */
typedef struct AumScanFwdArgs {
    const void *u, *delta, *z, *B, *C;
    const float *A, *A_b, *D, *delta_bias;
    void *out, *out_pre;
    float *last_state;
    void *workspace;
    int64_t workspace_bytes;
    int64_t u_bs, u_ds;         /* batch / channel strides of u (elements) */
    int64_t delta_bs, delta_ds; /* ... of delta */
    int64_t z_bs, z_ds;
    int64_t B_bs, B_ns;         /* batch / state strides of B */
    int64_t C_bs, C_ns;
    int64_t out_bs, out_ds;     /* strides of out and out_pre */
    int32_t batch, dim, len, dstate;
    int32_t dtype;
    uint32_t flags;
    float *x_ck;                /* ABI 2 */
    float *x_lane;              /* ABI 3 (optional, output): see aum_selective_scan_lane_ckpt_bytes */
} AumScanFwdArgs;

/*
 * Selective scan backward (selective_scan_cuda.bwd).  dout is the gradient of `out`.
 *   du, ddelta, dz : (batch, dim, len) in `dtype` (written; dz only when z != NULL)
 *   dA, dA_b : (dim, dstate); dD, ddelta_bias : (dim); dB, dC : (batch, dstate, len) -- all fp32, ACCUMULATED INTO
 *   (the caller zero-fills them; the fused bidirectional call adds both directions, selective_scan_interface.py:554-559).
 * out_pre is the tensor saved by the forward (required when z != NULL; it feeds dz).
 * workspace (required; size from aum_selective_scan_workspace_bytes(..., backward=1)): per-workgroup dB/dC partial
 * tiles and per-batch dA/dD/ddelta_bias partials that a second kernel of the same call reduces -- the backward issues
 * no global atomics.
 * The gradient is the mathematically complete one (autograd of bimamba_inner_ref), see DESIGN.md "dz".
 */
typedef struct AumScanBwdArgs {
    const void *u, *delta, *z, *B, *C, *dout, *out_pre;
    const float *A, *A_b, *D, *delta_bias;
    void *du, *ddelta, *dz;
    float *dA, *dA_b, *dB, *dC, *dD, *ddelta_bias;
    void *workspace;
    int64_t workspace_bytes;
    int64_t u_bs, u_ds, delta_bs, delta_ds, z_bs, z_ds, B_bs, B_ns, C_bs, C_ns;
    int64_t dout_bs, dout_ds, out_bs, out_ds; /* out_* = strides of out_pre */
    int64_t du_bs, du_ds, ddelta_bs, ddelta_ds, dz_bs, dz_ds;
    int64_t dB_bs, dB_ns, dC_bs, dC_ns;
    int32_t batch, dim, len, dstate;
    int32_t dtype;
    uint32_t flags;
    const float *x_ck;          /* ABI 2: the forward's checkpoint of the same call shape and direction, or NULL */
    const float *x_lane;        /* ABI 3: the forward's lane-entry checkpoint of the same call (shape, direction mode), or NULL */
} AumScanBwdArgs;

int aum_selective_scan_fwd(const AumScanFwdArgs* args, void* stream);
int aum_selective_scan_bwd(const AumScanBwdArgs* args, void* stream);
/* longest `len` handled in one pass per row (no workspace needed, direction fusion available) */
int aum_scan_max_single_pass_len(void);
int64_t aum_selective_scan_workspace_bytes(int32_t batch, int32_t dim, int32_t len, int32_t dstate, int32_t bidirectional,
                                           int32_t backward);
/* bytes of the x_ck checkpoint for this shape (one direction), 0 when the shape has none */
int64_t aum_selective_scan_ckpt_bytes(int32_t batch, int32_t dim, int32_t len, int32_t dstate);
/*
 * x_lane: the second form of the `x` tensor selective_scan_cuda.fwd returns for its backward (SSI:37-45), at the granularity
 * this library's L = 513 row kernels want it: the state entering every lane's 8-step block, per (row, direction, state) --
 * (batch, dim, directions, dstate, 64) fp32, directions = 2 for the fused bidirectional call.  Rows the row kernels take
 * (len == 513, dstate <= 16, neither AUM_SCAN_GENERIC nor AUM_SCAN_ROWPAIR) have one; bytes = 0 otherwise, and passing x_lane for
 * such a call is AUM_E_UNSUPPORTED.  Handed to the backward of the same call it replaces the recomputation of the forward scan
 * (the backward without it runs the previous-generation kernels).
 */
int64_t aum_selective_scan_lane_ckpt_bytes(int32_t batch, int32_t dim, int32_t len, int32_t dstate, int32_t bidirectional);

/*
 * Depthwise causal conv1d + bias + SiLU (causal_conv1d_cuda.causal_conv1d_fwd / _bwd).
 *   x, y, dy, dx : (batch, dim, len) in `dtype`;  weight : (dim, width) fp32;  bias : (dim) fp32 or NULL
 *   dweight (dim, width), dbias (dim): fp32, accumulated into (caller zero-fills).
 */
typedef struct AumConvArgs {
    const void *x, *dy;  /* dy: backward only */
    const float *weight, *bias;
    void *y;             /* forward: output.  backward: unused */
    void *dx;            /* backward */
    float *dweight, *dbias;
    int64_t x_bs, x_ds, y_bs, y_ds, dy_bs, dy_ds, dx_bs, dx_ds;
    int32_t batch, dim, len, width;
    int32_t dtype;
    uint32_t flags;
} AumConvArgs;

int aum_causal_conv1d_fwd(const AumConvArgs* args, void* stream);
int aum_causal_conv1d_bwd(const AumConvArgs* args, void* stream);

/*
 * Fused residual-add + RMSNorm (rms_norm_fn -> LayerNormFn, layernorm.py:380-478).
 *   x (rows, cols) in x_dtype; residual (rows, cols) in res_dtype or NULL; weight (cols) fp32;
 *   y (rows, cols) in y_dtype; residual_out (rows, cols) in res_dtype or NULL; rstd (rows) fp32.
 * backward: dy (rows, cols) in y_dtype; dresidual_out in res_dtype or NULL; x = the saved residual_out (or the input x
 * when no residual stream exists) in res_dtype; writes dx in x_dtype, dresidual_in (res_dtype, optional, receives the same
 * values as dx), and per-workgroup partial sums dweight_partial (n_partials, cols) fp32 that the caller reduces
 * (layernorm.py:333-372 does the same); n_partials = aum_rmsnorm_bwd_partials(rows).
 */
typedef struct AumNormArgs {
    const void *x, *residual, *dy, *dresidual_out;
    const float *weight, *rstd_in;
    void *y, *residual_out, *dx, *dresidual_in;
    float *rstd_out, *dweight_partial;
    int64_t row_stride_x, row_stride_res, row_stride_y, row_stride_res_out;
    int64_t row_stride_dy, row_stride_dres_out, row_stride_dx, row_stride_dres_in;
    float eps;
    int32_t rows, cols;
    int32_t x_dtype, res_dtype, y_dtype;
    uint32_t flags;
} AumNormArgs;

int aum_rmsnorm_fwd(const AumNormArgs* args, void* stream);
int aum_rmsnorm_bwd(const AumNormArgs* args, void* stream);
int aum_rmsnorm_bwd_partials(int32_t rows);
/* ABI 13: rows of dweight_partial that hold sums after aum_rmsnorm_bwd (<= aum_rmsnorm_bwd_partials(rows), which remains the size to
 * allocate): the vectorised kernels (cols <= 2048, no AUM_NORM_GENERIC) add eight waves' sums inside the workgroup, in wave order. */
int aum_rmsnorm_bwd_partial_rows(int32_t rows, int32_t cols, uint32_t flags);

/*
 * Log-mel filterbank frontend (replaces torchaudio.compliance.kaldi.fbank + pad + normalise on the CPU DataLoader
 * workers, src/dataloader.py:134-147, 220-221).
 *   wave : (batch, n_samples) fp32, already mean-removed (dataloader.py:101);  out : (batch, target_length, num_mel) fp32
 *   window (win), twiddle (padded/2 complex pairs exp(-2 pi i k / padded)), mel_start_f / mel_count_f (num_mel, small
 *   integers stored as fp32), mel_w (num_mel, mel_wstride): tables built by the host (aum/frontend.py).
 *   frames >= num_frames are written as the reference's zero padding after normalisation.
 */
typedef struct AumFbankArgs {
    const float *wave, *window, *twiddle, *mel_start_f, *mel_count_f, *mel_w;
    float *out;
    int64_t wave_bs, out_bs;
    int32_t batch, n_samples, win, shift, padded, num_frames, target_length, num_mel, mel_wstride;
    float preemph, norm_mean, norm_inv2std, log_floor;
    /* ABI 4, both optional.  aug: (batch, AUM_FBANK_AUG) fp32 per-clip table applied in the kernel's store (the training-time
     * chain of src/dataloader.py:139-145, 206-228 without extra passes over the spectrogram):
     *   [0] frames of this clip (ragged batches: frames >= it are the reference's zero padding; < 0: use num_frames)
     *   [1],[2] frequency band [lo, hi) and [3],[4] time band [lo, hi) set to 0 BEFORE normalisation (SpecAug; empty when lo >= hi)
     *   [5] roll along time (torch.roll shift, applied last)   [6] noise amplitude: out += noise[b][t][m] * amp, before the roll
     * noise: (batch, target_length, num_mel) fp32 uniform numbers from the caller's generator, or NULL (no noise). */
    const float *aug, *noise;
} AumFbankArgs;
#define AUM_FBANK_AUG 8
int aum_fbank_fwd(const AumFbankArgs* args, void* stream);

/*
 * Waveform -> token sequence in one launch (ABI 5): the log-mel frontend above, the 16 x 16 patch embedding, its bias, the
 * position rows and the cls row, i.e. src/dataloader.py:134-147, 206-228 -> src/models/mamba_models.py:509-541 with
 * src/utilities/tokenization.py:278-310 (FlexiPatchEmbed, kernel = stride = 16) -- without the spectrogram or an im2col copy
 * of it in HBM.
 *   fbank   : as for aum_fbank_fwd (aug / noise included); .out is ignored.  Needs padded == 512, num_mel == 128 and
 *             target_length % 64 == 0 (AUM_E_UNSUPPORTED otherwise: call aum_fbank_fwd and a GEMM instead).
 *   weight  : (dim, 256) in `dtype` (bf16 / f16) = the conv weight (dim, 1, 16, 16) flattened: k = 16 * mel_row + frame
 *   bias    : (dim) fp32      pos : (n_patches, dim) fp32, row f_block * n_t + t_block (pos_embed rows 1 ..)
 *   cls_row : (dim) fp32 = cls_token + pos_embed row 0, written as token `cls_pos` of every clip; NULL = no cls row
 *   tokens  : (batch, n_patches + (cls_row != NULL), dim) in `out_dtype`, batch stride tokens_bs elements:
 *             round16(patch . weight + bias) + pos   (the 16-bit conv output of the autocast reference, then the fp32 add)
 *   patches : optional (batch * n_patches, 256) in `dtype`, rows in f_block * n_t + t_block order: the GEMM's input, saved
 *             for the weight gradient.
 *   flags   : AUM_FRONTEND_TIME_MAJOR = patch tokens in time-major order (transpose_token_sequence, MM:545-566).
 */
typedef struct AumFrontendArgs {
    AumFbankArgs fbank;
    const void *weight;
    const float *bias, *pos, *cls_row;
    void *tokens, *patches;
    int64_t tokens_bs;
    int32_t dim, cls_pos, dtype, out_dtype;
    uint32_t flags;
} AumFrontendArgs;
#define AUM_FRONTEND_TIME_MAJOR 1u
int aum_frontend_tokens_fwd(const AumFrontendArgs* args, void* stream);

/*
 * The skinny projections around the scan on CHANNEL-MAJOR activations (token t = b*len + l contiguous):
 *   conv_out / delta / ddelta / dconv : [dim][ntok]     x_dbl / dx_dbl : [dt_rank + 2*dstate][ntok], rows = dt | B | C
 * 16-bit activations only (AUM_BF16 / AUM_F16, fp32 accumulation on the MFMA units); weights in the same dtype.
 * Limits: dim % 64 == 0, dt_rank <= 64, dt_rank + 2*dstate <= 80, ntok * 80 < 2^31; otherwise AUM_E_UNSUPPORTED and the
 * caller uses its library GEMMs.
 *
 * aum_proj_fwd       replaces SSI:467-468 (F.linear x_proj, delta_proj matmul) and the B/C slicing + transposes of
 *                    SSI:471-493:   act = conv_out (in), w_x [dt_rank+2*dstate][dim], w_dt [dim][dt_rank],
 *                    x_dbl (out), out_act = delta (out, no bias / softplus: the scan applies them).
 * aum_proj_bwd_data  replaces SSI:570-574 (dB/dC scatter into dx_dbl), SSI:587 and SSI:590:
 *                    act = ddelta (in), w_dt = W_dt^T [dt_rank][dim], w_x = W_x^T [dim][dt_rank+2*dstate],
 *                    dB/dC fp32 (batch, dstate, len) with element strides, x_dbl = dx_dbl (out),
 *                    out_act = dconv, ACCUMULATED in place on top of the scan's du.
 */
typedef struct AumProjArgs {
    const void *act, *w_x, *w_dt;
    void *x_dbl, *out_act;
    const float *dB, *dC;
    int64_t dB_bs, dB_ns, dC_bs, dC_ns;
    int64_t ntok;
    int32_t dim, dt_rank, dstate, len, dtype;
    int32_t w_ld;   /* row pitch (elements, multiple of 8) of the matrix whose rows run along the SHORT k dimension:
                       forward: w_dt [dim][w_ld >= dt_rank]; backward: w_x = W_x^T [dim][w_ld >= dt_rank + 2*dstate].
                       Pad columns are never multiplied in (the other operand is zero there) but must be readable. */
} AumProjArgs;
int aum_proj_fwd(const AumProjArgs* args, void* stream);
int aum_proj_bwd_data(const AumProjArgs* args, void* stream);
/*
 * aum_proj_bwd_weight  replaces SSI:586 and SSI:589: partial[s][...] = sum over the s-th token range of
 *                    x[e][t] * y[r][t]  (x: [dim][ntok] 16-bit, y: [nrows][ntok] 16-bit, nrows <= 80), written as
 *                    [dim][nrows] (transpose_out = 0) or [nrows][dim] (transpose_out = 1) fp32 per split; the caller
 *                    sums the nsplit partials.  nsplit = aum_proj_bwd_weight_splits(dim, ntok).
 */
typedef struct AumProjWArgs {
    const void *x, *y;
    float* out;                 /* [nsplit][dim * nrows] */
    int64_t ntok;
    int32_t dim, nrows, nsplit, transpose_out, dtype;
} AumProjWArgs;
int aum_proj_bwd_weight(const AumProjWArgs* args, void* stream);
int aum_proj_bwd_weight_splits(int32_t dim, int64_t ntok);

/*
 * Selective scan on TOKEN-MAJOR activations, time-serial (ABI 7).  Same arithmetic as aum_selective_scan_fwd/_bwd
 * (selective_scan_cuda.fwd / .bwd, SSI:37, 62-65, 499-507, 541-561; oracle selective_scan_ref SSI:86-152), different division of the
 * work: a wavefront owns 64 CHANNELS of one batch entry and walks the sequence step by step with the 16 states of every channel in
 * registers; B_t / C_t are wave-uniform and come through the scalar cache.  No associative scan, no LDS tiles, any length.
 *   u, delta, z, out, out_pre, dout, du, ddelta, dz : element (b, t, e) at  b * X_bs + t * X_ts + e   (channels contiguous -- the
 *       natural output of F.linear; z / dz may be the second half of an xz row: pass the offset pointer and X_ts = 2 * dim)
 *   B, C : element (b, t, n) at  b * X_bs + t * X_ts + n, same dtype; rows 4-byte aligned (even strides / offsets for 16-bit dtypes)
 *   A, A_b : (dim, dstate) fp32;  D, delta_bias : (dim) fp32 or NULL
 * A_b != NULL: both directions in one launch (BiMambaInnerFn, SSI:499-507): out = gate * (y_fwd + y_rev + 2 D u); the two
 * direction waves of a channel group meet in the middle of the sequence and exchange their halves through `out`.
 * AUM_SCAN_REVERSE (A_b == NULL): the recurrence runs from t = len-1 down to 0.
 * ckpt (optional, forward output / backward input): the state entering every AUM_SCAN_TM_CK-step block in scan order,
 *   (directions, batch, nck, rows, dim) dwords with nck = aum_scan_tm_nck(len) and rows = aum_scan_tm_ckpt_rows(dtype) -- the `x` tensor
 *   of selective_scan_cuda.fwd at this kernel's granularity: the dstate states as fp32 for fp32 activations, dstate / 2 rows of PAIRS in the
 *   activations' own 16-bit type (states 2j, 2j+1 in one dword) for 16-bit activations (ABI 9; the entry state of an 8-step block rounded to the precision of the block's
 *   own inputs: half the checkpoint traffic of the training forward, which is HBM-bound on it).  A buffer sized for dstate rows is always
 *   large enough; forward and backward of one call must see the same dtype.  The backward requires it.
 * Limits: dstate == 16, dim % 64 == 0, len * X_ts * sizeof(element) < 2^31 for every tensor; otherwise AUM_E_UNSUPPORTED (callers
 * use aum_selective_scan_*).
 */
#define AUM_SCAN_TM_CK 8
typedef struct AumScanTmFwdArgs {
    const void *u, *delta, *z, *B, *C;
    const float *A, *A_b, *D, *delta_bias;
    void *out, *out_pre;
    float *ckpt;
    int64_t u_bs, u_ts, delta_bs, delta_ts, z_bs, z_ts, B_bs, B_ts, C_bs, C_ts, out_bs, out_ts, pre_bs, pre_ts;
    int32_t batch, dim, len, dstate;
    int32_t dtype;
    uint32_t flags;
} AumScanTmFwdArgs;
int aum_scan_tm_fwd(const AumScanTmFwdArgs* args, void* stream);
int32_t aum_scan_tm_nck(int32_t len);
int32_t aum_scan_tm_ckpt_rows(int32_t dtype);

/*
 * Backward of the above.  du, ddelta, dz in `dtype` (written).  dBC: (batch, len, 2 * dstate) fp32 = dB | dC per token, written
 * (the sum over the channel-group partials the kernel leaves in `workspace`).  dA, dA_b (dim, dstate), dD, ddelta_bias (dim): fp32,
 * written.  out_pre: the pre-gate sum saved by the forward (required with z).  workspace: aum_scan_tm_workspace_bytes(...) bytes.
 */
typedef struct AumScanTmBwdArgs {
    const void *u, *delta, *z, *B, *C, *dout, *out_pre;
    const float *A, *A_b, *D, *delta_bias;
    const float *ckpt;
    void *du, *ddelta, *dz;
    float *dA, *dA_b, *dBC, *dD, *ddelta_bias;
    void *workspace;
    int64_t workspace_bytes;
    int64_t u_bs, u_ts, delta_bs, delta_ts, z_bs, z_ts, B_bs, B_ts, C_bs, C_ts, dout_bs, dout_ts, pre_bs, pre_ts;
    int64_t du_bs, du_ts, ddelta_bs, ddelta_ts, dz_bs, dz_ts;
    int32_t batch, dim, len, dstate;
    int32_t dtype;
    uint32_t flags;
    /* ABI 10, optional (NULL: not written): dA .* A and dA_b .* A_b, (dim, dstate) fp32 -- the gradient with respect to A_log when
       A = -exp(A_log) (mamba_simple.py:190, 204), written by the same partial-sum launch that writes dA / dA_b */
    float *dA_xA, *dA_b_xA;
} AumScanTmBwdArgs;
int aum_scan_tm_bwd(const AumScanTmBwdArgs* args, void* stream);
int64_t aum_scan_tm_workspace_bytes(int32_t batch, int32_t dim, int32_t len, int32_t dstate, int32_t bidirectional);

/*
 * The same two operators on rows cut into TIME SEGMENTS (ABI 10): long rows at a small batch -- the reference's long-form setting
 * (B = 8, L = 4097: SURVEY section 8 config 5) is 384 (batch entry, channel group, direction) waves on 1024 SIMDs, each a 4097-step
 * serial chain.  The recurrence x_t = a_t x_{t-1} + b_t is affine in the state, so every direction is cut into `segments` ranges of
 * steps that run as waves of their own: a carry pass leaves each range's exit state from a zero entry and the product of its decays,
 * every range then starts from the composition of the carries before it (selective_scan_fn's results: SSI:86-152 restated on
 * ranges; nothing is approximated -- the sums are re-associated in fp32).  The backward cuts the adjoint recurrence the same way.
 *   segments: 2 .. AUM_SCAN_TM_MAX_SEGMENTS; ranges are ceil(len / segments) steps rounded up to a multiple of AUM_SCAN_TM_CK.
 *   carry: aum_scan_tm_seg_carry_bytes(...) bytes of scratch (forward); the backward keeps its carries in its workspace, which is
 *   aum_scan_tm_seg_workspace_bytes(...) bytes.  ckpt has the unsegmented layout: the two forms of the forward may be mixed with the
 *   two forms of the backward.  Everything else as aum_scan_tm_fwd / aum_scan_tm_bwd (base).
 */
#define AUM_SCAN_TM_MAX_SEGMENTS 32
typedef struct AumScanTmSegFwdArgs {
    AumScanTmFwdArgs base;
    float* carry;
    int64_t carry_bytes;
    int32_t segments;
    int32_t reserved;
} AumScanTmSegFwdArgs;
typedef struct AumScanTmSegBwdArgs {
    AumScanTmBwdArgs base;
    int32_t segments;
    int32_t reserved;
} AumScanTmSegBwdArgs;
int aum_scan_tm_seg_fwd(const AumScanTmSegFwdArgs* args, void* stream);
int aum_scan_tm_seg_bwd(const AumScanTmSegBwdArgs* args, void* stream);
int64_t aum_scan_tm_seg_carry_bytes(int32_t batch, int32_t dim, int32_t len, int32_t dstate, int32_t bidirectional, int32_t segments);
int64_t aum_scan_tm_seg_workspace_bytes(int32_t batch, int32_t dim, int32_t len, int32_t dstate, int32_t bidirectional, int32_t segments);

/*
 * Depthwise causal conv1d (+ SiLU) on TOKEN-MAJOR activations (ABI 8): the same operator as aum_causal_conv1d_fwd / _bwd
 * (MS:272 causal_conv1d_fn; SSI:463 forward and SSI:594-596 backward call sites) for tensors laid out (batch, len, dim) with the
 * channel contiguous -- the layout of the in_proj output rows [x | z] and of the time-serial scan's operands, so x / dx may be the
 * first half of a (batch, len, 2 dim) tensor (strides in ELEMENTS: *_bs batch, *_ts token; channel stride 1).
 *   x, y, dy, dx in `dtype`; weight (dim, width) fp32, width <= 4; bias (dim) fp32 or NULL; flags: AUM_CONV_SILU, AUM_CONV_REVERSE.
 *   backward: dx written; dw_part [nparts][dim][width] (the weight's layout) and db_part [nparts][dim] (NULL without bias) fp32 are per-wave partial sums
 *   (nparts = aum_conv1d_tm_nparts(batch, len)) that the caller adds up in a fixed order (aum_sum_rows): no atomics.
 * Limits: dim % (16 / sizeof(dtype)) == 0, 16-byte aligned pointers (weight and bias included) and row strides.
 */
typedef struct AumConvTmArgs {
    const void *x, *dy;      /* dy: backward only */
    const float *weight, *bias;
    void *y;                 /* forward */
    void *dx;                /* backward */
    float *dw_part, *db_part;
    int64_t x_bs, x_ts, y_bs, y_ts, dy_bs, dy_ts, dx_bs, dx_ts;
    int32_t batch, dim, len, width;
    int32_t dtype;
    uint32_t flags;
} AumConvTmArgs;
int aum_conv1d_tm_fwd(const AumConvTmArgs* args, void* stream);
int aum_conv1d_tm_bwd(const AumConvTmArgs* args, void* stream);
int32_t aum_conv1d_tm_nparts(int32_t batch, int32_t len);

/*
 * Dense projection GEMM on token-major activations (ABI 9): the in_proj / out_proj matrix products of the Mamba block and their
 * data gradients, which the reference leaves to cuBLAS through F.linear (vim-mamba_ssm/mamba_ssm/modules/mamba_simple.py:185-189;
 * selective_scan_interface.py:517 out_proj forward, :540 its data gradient; the in_proj data gradient is autograd's).
 *   c[m][n] = sum over k of a[m][k] * b[n][k]      ("TN": both operands K-contiguous; fp32 accumulation, one rounding at the store)
 *   a: (m, k) row pitch lda;  b: (n, k) row pitch ldb;  c: (m, n) row pitch ldc -- pitches in ELEMENTS, all three tensors `dtype`
 *   (AUM_BF16 or AUM_F16; AUM_F32: AUM_E_DTYPE).  m is arbitrary (the token count); n % 256 == 0 and k % 64 == 0 (the model widths
 *   768 / 1536 / 3072; else AUM_E_UNSUPPORTED); pointers 16-byte aligned, pitches % 8 == 0, every tensor below 2 GiB per 256-row block
 *   (32-bit buffer offsets).  forward: a = activations, b = the weight as nn.Linear stores it; data gradient: a = d out, b = the weight's transpose.
 */
typedef struct AumGemmArgs {
    const void *a, *b;
    void *c;
    int32_t m, n, k;
    int32_t lda, ldb, ldc;
    int32_t dtype;
    uint32_t flags;
} AumGemmArgs;
int aum_gemm_tn(const AumGemmArgs* args, void* stream);

/*
 * Weight-gradient GEMM of the in / out projections on token-major operands (ABI 10; autograd of mamba_simple.py:185-189 and
 * selective_scan_interface.py:563: d W = d out^T . input):
 *   part[s][n][k] = sum over the tokens t of split s of  y[t][n] * x[t][k]          s = 0 .. splits - 1
 *   y: (t, n) rows of pitch ldy (the output gradient);  x: (t, k) rows of pitch ldx (the layer's input) -- pitches in ELEMENTS, both `dtype`
 *   (AUM_BF16 / AUM_F16; else AUM_E_DTYPE);  part: (splits, n, k) fp32, contiguous: the caller sums the splits in a fixed order (aum_sum_rows).
 *   Split s takes tokens [s c, min(t, (s + 1) c)) with c = ceil(ceil(t / splits) / 64) * 64 (a split may be empty: its tile is zero).
 *   n % 256 == 0, k % 256 == 0 or k in {48, 80} (the skinny operands of the dt_proj / x_proj weight gradients, SSI:586, 589), splits <= 64, pitches % 8 == 0, 16-byte aligned pointers, c * pitch * 2 < 2 GiB (else AUM_E_UNSUPPORTED: callers
 *   use a library GEMM).  fp32 accumulation; no atomics: the result is bitwise repeatable.
 */
typedef struct AumGemmWArgs {
    const void *y, *x;
    float* part;
    int64_t t;
    int64_t ldy, ldx;
    int32_t n, k, splits;
    int32_t dtype;
} AumGemmWArgs;
int aum_gemm_wgrad(const AumGemmWArgs* args, void* stream);

/*
 * dt projection of the token-major block (ABI 9): delta = x_dbl[:, :dt_rank] . dt_proj.weight^T (selective_scan_interface.py:468 without
 * its transposes; the bias and the softplus stay in the scan, as in the reference).
 *   x: (ntok, >= rank) rows of pitch ldx, the first `rank` columns are read (x_dbl: the dt block sits in front of B and C);
 *   w: (dim, rank) rows of pitch ldw (dt_proj.weight as nn.Linear stores it);  out: (ntok, dim) rows of pitch ldo.  Pitches in ELEMENTS, all
 *   three tensors `dtype` (AUM_BF16 / AUM_F16; else AUM_E_DTYPE); dim % 32 == 0, rank % 8 == 0, rank <= 64, pitches % 8 == 0, 16-byte
 *   aligned pointers (else AUM_E_UNSUPPORTED: callers use a library GEMM).  fp32 accumulation, one rounding at the store.
 */
typedef struct AumDtProjArgs {
    const void *x, *w;
    void *out;
    int64_t ntok;
    int32_t dim, rank;
    int32_t ldx, ldw, ldo;
    int32_t dtype;
} AumDtProjArgs;
int aum_dtproj_tm_fwd(const AumDtProjArgs* args, void* stream);

/*
 * x_proj and dt_proj of the token-major block in one pass over conv_out (ABI 9; selective_scan_interface.py:467-468 without their
 * transposes): x_dbl = u . x_proj.weight^T, delta = x_dbl[:, :rank] . dt_proj.weight^T (bias and softplus stay in the scan).
 *   u: (ntok, dim) rows of pitch ldu (conv_out);  wx: (ncols, dim) pitch ldwx;  wdt: (dim, rank) pitch ldwdt;
 *   x_dbl (out): (ntok, ncols) pitch ldx;  delta (out): (ntok, dim) pitch ldd.  Pitches in ELEMENTS, all tensors `dtype` (AUM_BF16 / AUM_F16).
 *   Built for ncols == 80 or 56 (dt_rank + 2 d_state of AuM-Base / AuM-Small), dim % 256 == 0, dim <= 1536, rank % 8 == 0, rank <= 64, pitches % 8 == 0,
 *   16-byte aligned pointers; anything else AUM_E_UNSUPPORTED (callers use a GEMM for x_dbl and aum_dtproj_tm_fwd).  delta is computed from
 *   the ROUNDED x_dbl, as the two separate products do.
 */
typedef struct AumXdtArgs {
    const void *u, *wx, *wdt;
    void *x_dbl, *delta;
    int64_t ntok;
    int32_t dim, rank, ncols;
    int32_t ldu, ldwx, ldwdt, ldx, ldd;
    int32_t dtype;
} AumXdtArgs;
int aum_xdt_tm_fwd(const AumXdtArgs* args, void* stream);

/*
 * The token-parallel pieces of the x_proj / dt_proj gradients in one pass (ABI 10; selective_scan_interface.py:570-574, :587, :590):
 *   dx_dbl[:, :rank] = ddelta . dt_proj.weight  (rounded to `dtype`),  dx_dbl[:, rank:] = dB | dC (the scan backward's fp32 rows),
 *   du += dx_dbl . x_proj.weight  (in place).
 *   ddelta: (ntok, dim) pitch ldd;  dbc: (ntok, ncols - rank) fp32 pitch lddbc;  wdt_t: dt_proj.weight TRANSPOSED, (rank, dim) pitch ldwdt;
 *   wx_t: x_proj.weight TRANSPOSED, (dim, ncols) pitch ldwx;  du (in / out): (ntok, dim) pitch ldu;  dx_dbl (out): (ntok, ncols) pitch ldx.
 *   Built for ncols == 80, rank == 48 (AuM-Base), dim % 256 == 0, dim <= 1536, pitches % 8 == 0 (lddbc % 4 == 0), 16-byte aligned pointers;
 *   anything else AUM_E_UNSUPPORTED (callers use the three library calls).
 */
typedef struct AumXdtBwdArgs {
    const void *ddelta;
    const float *dbc;
    const void *wdt_t, *wx_t;
    void *du, *dx_dbl;
    int64_t ntok;
    int32_t dim, rank, ncols;
    int32_t ldd, lddbc, ldwdt, ldwx, ldu, ldx;
    int32_t dtype;
} AumXdtBwdArgs;
int aum_xdt_tm_bwd(const AumXdtBwdArgs* args, void* stream);

/*
 * Streaming inference, one token per call (ABI 10; mamba_simple.py:313-358 `Mamba.step`).  The caches are fp32, contiguous, updated in place.
 *
 * aum_causal_conv1d_update -- `causal_conv1d_update(x, conv_state, weight, bias, activation)` of the causal_conv1d wheel (call site
 *   mamba_simple.py:328-334; arithmetic = the fallback at :322-327): conv_state (batch, dim, width) <- (conv_state[..., 1:], x);
 *   out (batch, dim) = act(sum_k conv_state[..., k] * weight[dim][k] + bias).  x / out: `dtype`, contiguous; weight (dim, width), bias (dim) fp32;
 *   flags: AUM_CONV_SILU.  width <= 8.
 * aum_selective_state_update -- `selective_state_update(state, x, dt, A, B, C, D, z, dt_bias, dt_softplus)` (ops/triton/selective_state_update.py:
 *   157-192, call site mamba_simple.py:352-354): dt' = softplus?(dt + dt_bias); state (batch, dim, dstate) <- exp(dt' A) state + dt' x B;
 *   out (batch, dim) = (<state, C> + D x) * silu(z).  x, dt, z, out (batch, dim) and B, C (batch, dstate): `dtype`, contiguous; A (dim, dstate),
 *   D, dt_bias (dim) fp32 (D, dt_bias, z may be NULL);  flags: AUM_SCAN_SOFTPLUS.  dstate <= 256.
 */
typedef struct AumConvUpdateArgs {
    const void* x;
    float* conv_state;
    const float *weight, *bias;
    void* out;
    int32_t batch, dim, width;
    int32_t dtype;
    uint32_t flags;
} AumConvUpdateArgs;
int aum_causal_conv1d_update(const AumConvUpdateArgs* args, void* stream);
typedef struct AumStateUpdateArgs {
    float* state;
    const void *x, *dt, *z, *B, *C;
    const float *A, *D, *dt_bias;
    void* out;
    int32_t batch, dim, dstate;
    int32_t dtype;
    uint32_t flags;
} AumStateUpdateArgs;
int aum_selective_state_update(const AumStateUpdateArgs* args, void* stream);

/* Self-tests and calibration (used by tests/ and bench.py; not part of the reference's surface). */
int aum_abi_version(void);
/* runs wave_scan_affine<rev> on 64 (P,S) pairs: in/out are device arrays of 128 floats (P[0..63], S[0..63]) */
int aum_selftest_wave_scan(const float* in, float* out, int rev, void* stream);
/* runs wave_sum32 (transposing butterfly sum over the 64 lanes) on 32 x 64 values in[k][lane]; out[lane] = the total of value
 * 2 * (lane & 15) + ((lane >> 4) & 1); out[64 + lane] = wave_sum16 of values 0..15; out[128 + lane] / out[192 + lane] = the matrix-pipe
 * sums (terms rounded to bf16, round 4) of values 0..15 / 16..31: the total of value 4 * (lane >> 4) + bit3(lane) + 2 * bit2(lane) of
 * the tile.  out: 256 floats */
int aum_selftest_wave_sum32(const float* in, float* out, void* stream);
/* float4 streaming copy, the measured-HBM-roofline denominator of SURVEY.md 8(d) */
int aum_hbm_copy(const void* src, void* dst, int64_t bytes, void* stream);

/*
 * dst[b][i] = sum over o of src[b][o][i] in fp32 (ABI 6; batch = 1 for a plain sum, > 1 for the first stage of a two-stage sum): the fixed-order sum of the per-workgroup partial results that
 * aum_rmsnorm_bwd (dweight_partial) and aum_proj_bwd_weight (out) leave to the caller, and of split-K GEMM partial products
 * (SSI:563, 586, 589; the reference leaves the same kind of sum to torch: LN:333-372).  src: (batch, outer, inner) contiguous in
 * src_dtype (AUM_F32 / AUM_BF16 / AUM_F16), dst: (batch, inner) fp32.  inner % 8 == 0; src 16-byte aligned, dst any float address (round 6: a parameter
 * gradient's view inside a DistributedDataParallel bucket may sit behind an odd-sized parameter).
 */
int aum_sum_rows(const void* src, float* dst, int64_t batch, int64_t outer, int64_t inner, int32_t src_dtype, void* stream);

/*
 * ABI 11: up to AUM_SUM_MAX_JOBS independent fixed-order sums of fp32 partial results in ONE launch -- a layer's backward leaves five partial sets behind
 * (the out_proj weight-gradient splits and the two skinny partial sets of aum_gemm_wgrad, SSI:563, 586, 589; the conv weight and bias partials of
 * aum_conv1d_tm_bwd), and a 5 us launch per set is 5 us of the step.  Job q:  dst[i] = sum over o < outer of src[o][i],  i < inner  -- each job with
 * the row grouping aum_sum_rows (batch = 1) picks for it, i.e. the same additions in the same order as a launch of its own: bitwise the same result.  tr_cols > 0: the summed (inner / tr_cols, tr_cols) matrix is stored transposed, dst
 * (tr_cols, inner / tr_cols) -- the x_proj weight gradient leaves aum_gemm_wgrad as (dim, R + 2N) and the parameter is (R + 2N, dim).
 * 1 <= njobs <= AUM_SUM_MAX_JOBS, inner % 8 == 0, inner % tr_cols == 0; src 16-byte aligned, dst any float address; `jobs` is read on the host during the call.
 */
#define AUM_SUM_MAX_JOBS 8
typedef struct AumSumJob {
    const float* src;       /* (outer, inner) contiguous */
    float* dst;             /* (inner), or (tr_cols, inner / tr_cols) */
    int64_t outer, inner;
    int32_t tr_cols;
    int32_t reserved;
} AumSumJob;
int aum_sum_rows_multi(const AumSumJob* jobs, int32_t njobs, void* stream);

/* The 16-bit working copies of fp32 master weights (ABI 13).  Replaces: torch.autocast's per-call cast of F.linear's weight operand (the
 * reference's in_proj / x_proj / dt_proj / out_proj under --mixed_precision, MS:185-189, SSI:467-468, 517) -- hoisted to one launch per
 * group of equally shaped matrices at the top of a forward (ssi.step_cache).
 *   src     DEVICE array of n addresses, each an fp32 (rows, cols) contiguous matrix
 *   bank    (n, rows, cols) in `dtype` (AUM_BF16 / AUM_F16), round-to-nearest-even (what Tensor.to() does)
 *   bank_t  (n, cols, rows): the transposes, or NULL
 * rows % 8 == 0, cols % 4 == 0; bank / bank_t 16-byte aligned. */
int aum_cast_bank(const uint64_t* src, int32_t n, int32_t rows, int32_t cols, void* bank, void* bank_t, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUM_HIP_H */
