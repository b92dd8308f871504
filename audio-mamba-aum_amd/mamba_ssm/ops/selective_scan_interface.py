"""Drop-in for the reference's mamba_ssm.ops.selective_scan_interface
(/root/reference/vim-mamba_ssm/mamba_ssm/ops/selective_scan_interface.py = "SSI"): same public names, argument
meaning and return conventions, backed by the hand-written gfx950 kernels of libaum_hip.so (aum_hip) instead of the
selective_scan_cuda / causal_conv1d_cuda wheels.

What is different by design (MI355X-first):
  * the reverse direction of Fo-Bi / Bi-Bi is a flag on the kernels; none of the .flip([-1]) copies of SSI:504-507,
    547-561 or MS:229-246 exist;
  * Fo-Bi runs ONE direction-fused scan launch forward and ONE backward (inputs read once, out_z written once);
  * activations between the kernels and the projections are stored channel-major [E][B][L] so x_proj / dt_proj /
    out_proj and all their gradients are transpose-free GEMMs (hipBLASLt -> MFMA);
  * nothing is recomputed in backward (SSI:218-219 drops conv1d_out/delta to save memory on 16-80 GB parts; on 288 GB
    of HBM3E the recompute costs more than the bytes), so conv1d_out, delta, out_z and the pre-gate sum are saved;
  * the backward returns the mathematically complete dz (autograd of bimamba_inner_ref); the reference's fused v1
    backward drops the reverse-direction dz term at SSI:560/599.  AUM_REF_DZ_DROP=1 reproduces that for A/B runs.

The *_ref functions are this package's own pure-PyTorch statements of the same arithmetic (public names of the
reference module; usable on CPU).  They are NOT used by the fused path.
"""
import contextlib
import os

import torch
import torch.nn.functional as F

import aum_hip

# Host-side switches (README.md "Switches"), read ONCE at import: a variable set later in a running job changes nothing.  As in
# aum_hip.debug, the ones that change WHICH kernels or GEMMs run are honoured only under AUM_DEBUG=1 (A/B runs, tools/); a production job
# cannot be steered onto another path by a stray variable.
_DBG = os.environ.get("AUM_DEBUG") == "1"


def _dbg_env(name, default):
    return os.environ.get(name, default) if _DBG else default


_STEP_CACHE_ON = _dbg_env("AUM_STEP_CACHE", "1") != "0"
_WGRAD_SPLIT_ON = _dbg_env("AUM_WGRAD_SPLIT", "1") != "0"
_REF_DZ_DROP = os.environ.get("AUM_REF_DZ_DROP", "0") == "1"        # a numerics option (the reference's dz), not a kernel switch
# AUM_TOKEN_MAJOR=0: the blocks keep their activations channel-major [E][B*L] (rounds 1-2: the row kernels); default: token-major
# [B*L][E] rows (the time-serial scan kernels, the register-window conv, plain row-major GEMMs) where the shape allows it
TOKEN_MAJOR = _dbg_env("AUM_TOKEN_MAJOR", "1") != "0"

_custom_fwd = torch.amp.custom_fwd(device_type="cuda")
_custom_bwd = torch.amp.custom_bwd(device_type="cuda")


def _autocast_dtype():
    return torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else None


# =================================================================================================
# gradient homes
# =================================================================================================
# Under DistributedDataParallel(gradient_as_bucket_view=True) a parameter's gradient lives in a view of a flat bucket.  With
# zero_grad(set_to_none=True) (torch's default) every backward produces a NEW gradient tensor and the reducer's per-parameter hook copies
# it into the bucket: 271 copy launches and 2 x 368 MB of traffic per AuM-Base step, 1.3 ms of GPU time with no wire at all
# (profiles/r06_ddp_overhead.txt).  A parameter that carries `_aum_grad_home` (adopt_grad_homes: the bucket view the reducer gave it in an
# earlier step) has its gradient WRITTEN there by the kernel that finishes it -- the weight-gradient sums, the scan's parameter sums, the norm's
# weight sum -- and what the autograd function returns is a fresh alias of that view: the accumulator keeps it without a copy, the
# reducer's hook finds a gradient that already aliases its bucket and copies nothing.  A stale home (buckets rebuilt, another wrapper) only
# costs the copy it was meant to save: the reducer then copies as before.
def grad_home(p):
    """where parameter p's gradient is wanted, or None.  Only while p.grad is None: a backward that ACCUMULATES (gradient accumulation,
    no_sync) must not overwrite what it is added to."""
    h = getattr(p, "_aum_grad_home", None) if p is not None else None
    if h is None or p.grad is not None or h.dtype != torch.float32 or p.dtype != torch.float32 or h.device != p.device or h.shape != p.shape \
            or not h.is_contiguous():
        return None
    return h


def _homed(t, home):
    """what the backward returns for a gradient `t` that was written into `home`: a fresh tensor object on the same storage (the one the
    gradient accumulator may keep); anything else unchanged"""
    if home is not None and t is not None and t.data_ptr() == home.data_ptr() and t.numel() == home.numel():
        HOME_HITS[0] += 1
        return home.view(home.shape)
    return t


HOME_HITS = [0]         # gradients handed over inside their home since import (tests, bench.py's dist record)


def adopt_grad_homes(module):
    """after a backward under DistributedDataParallel(gradient_as_bucket_view=True): remember each parameter's bucket view as the place its
    next gradient is written to.  Returns the number of parameters that have one.  (TT:39, TT:168: the reference wraps the model in the
    same reducer; its CUDA kernels cannot be told where to write.)"""
    n = 0
    for p in module.parameters():
        g = p.grad
        if g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape and g._base is not None:
            p._aum_grad_home = g
            n += 1
    return n


class GradHomes:
    """adopt_grad_homes on a schedule: after each of the first backward passes (the reducer rebuilds its buckets once, after the first
    iteration) and every 64th one after that (a stale home costs a copy, never a wrong gradient)"""

    def __init__(self, module):
        self.module, self.calls = module, 0

    def after_backward(self):
        self.calls += 1
        if self.calls <= 3 or self.calls % 64 == 0:
            return adopt_grad_homes(self.module)
        return None


def drop_grad_homes(module):
    for p in module.parameters():
        if hasattr(p, "_aum_grad_home"):
            del p._aum_grad_home


_A_OWNER = {}           # storage address of a cached A (neg_exp) -> the A_log parameter it came from, for the open forward


# =================================================================================================
# per-forward weight cache
# =================================================================================================
# What autocast does per call -- one fp32 -> 16-bit cast kernel per projection weight and layer, plus this package's transposed
# copies for the data-gradient kernels and the two -exp(A_log) chains -- is ~250 launches of 3-6 us per AuM-Base step.  A model
# that owns many blocks can do all of it in a handful of launches at the top of its forward (`with step_cache(mixers, dtype)`); the
# blocks then find their 16-bit weights, the transposes and A here.  Entries live for one forward (the autograd graph keeps what the
# backward needs); a Mamba block used on its own finds nothing and casts per call, exactly as before.
# NOT thread-safe: the dict is process-global and keyed by id(parameter); two forwards of the SAME model in two threads would pop each
# other's entries (the result stays right -- a missing entry is a per-call cast -- but the saving is lost).  One forward per process at a
# time is what the launcher and bench.py do.
_STEP_CACHE = {}
_CAST_HIP = _dbg_env("AUM_CAST_LIB", "0") != "1"         # AUM_DEBUG=1 AUM_CAST_LIB=1: the step cache's casts / transposes as torch copies (A/B)


@contextlib.contextmanager
def step_cache(mixers, dtype):
    groups, a_logs = {}, []
    if not _STEP_CACHE_ON:                                      # A/B switch: per-call casts, as a block used on its own does
        mixers = []
    for m in mixers:
        for name in ("in_proj", "x_proj", "dt_proj", "out_proj", "x_proj_b", "dt_proj_b"):
            lin = getattr(m, name, None)
            if lin is not None and dtype is not None and lin.weight.dtype != dtype:
                # data gradient on the MFMA kernel: g [tokens, out] @ W [out, in] = gemm_tn(g, W^T [in, out]) -> (N, K) = (in, out)
                want_t = name in ("x_proj", "dt_proj", "x_proj_b", "dt_proj_b") or (
                    torch.is_grad_enabled() and lin.weight.is_cuda and lin.weight.dim() == 2
                    and _hip_gemm_ok(lin.weight.new_empty(0, dtype=dtype), lin.weight.shape[1], lin.weight.shape[0]))
                groups.setdefault((want_t, tuple(lin.weight.shape), lin.weight.device), []).append(lin.weight)
        a_logs += [p for p in (getattr(m, "A_log", None), getattr(m, "A_b_log", None)) if p is not None]
    mine, a_ptrs = [], []
    _DA_XA.clear()
    with torch.no_grad():
        for (want_t, shape, dev), ps in groups.items():
            if _CAST_HIP and dev.type == "cuda" and aum_hip.cast_bank_supported(ps, dtype):
                bank, bank_t = aum_hip.cast_bank([p.detach() for p in ps], dtype, want_t)      # every matrix read once, both copies in one launch
            else:
                bank = torch.empty((len(ps),) + shape, dtype=dtype, device=dev)
                torch._foreach_copy_(list(bank.unbind(0)), [p.detach() for p in ps])          # one multi-tensor cast
                bank_t = bank.transpose(1, 2).contiguous() if want_t else None
            for i, p in enumerate(ps):
                _STEP_CACHE[id(p)] = (dtype, bank[i], None if bank_t is None else bank_t[i])
                mine.append(id(p))
        by_shape = {}
        for p in a_logs:
            by_shape.setdefault((tuple(p.shape), p.device), []).append(p)
        for ps in by_shape.values():
            A = -torch.exp(torch.stack([p.detach().float() for p in ps]))
            for i, p in enumerate(ps):
                _STEP_CACHE[id(p)] = ("A", A[i], None)
                mine.append(id(p))
                a_ptrs.append(A[i].data_ptr())
    _A_CACHE_PTRS.update(a_ptrs)
    try:
        yield
    finally:
        for k in mine:                      # only this context's entries: another model's forward may be open around this one
            _STEP_CACHE.pop(k, None)
        _A_CACHE_PTRS.difference_update(a_ptrs)
        for a_ in a_ptrs:
            _A_OWNER.pop(a_, None)


def _cast(w, dtype):
    """w in `dtype` (None: unchanged): the cached copy of the running forward if there is one, else a cast."""
    if w is None or dtype is None or w.dtype == dtype:
        return w
    c = _STEP_CACHE.get(id(w))
    return c[1] if c is not None and c[0] == dtype else w.to(dtype)


def _cast_t(w, dtype):
    """contiguous transpose of the 2-D weight w in `dtype` (None: w's own)"""
    c = _STEP_CACHE.get(id(w))
    if c is not None and c[0] == dtype and c[2] is not None:
        return c[2]
    return (w if dtype is None else w.to(dtype)).t().contiguous()


# d A_log = d A .* A.  The token-major scan backward writes that product next to d A in its partial-sum launch (aum_scan_tm_bwd: dA_xA);
# it is handed over here, keyed by the STORAGE of the d A tensor the block returned -- the entry holds that tensor, so the address cannot
# be reused while the entry lives; a gradient that is not that very tensor (summed with another use of A, copied) misses and is
# multiplied as before.  Entries nobody claimed are dropped when the next forward opens its cache.
_DA_XA = {}
_A_CACHE_PTRS = set()           # storage addresses of the A / A_b tensors the open step caches hold (what neg_exp hands out)


def _da_xa_put(dA, prod, A):
    if not _DA_XA:
        # unclaimed entries (and the (E, N) tensors they pin) go when this backward pass ends, not when the next forward opens a cache
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_DA_XA.clear)
        except RuntimeError:
            pass                 # not inside a backward pass (a test calling the block's backward helper directly): step_cache() clears
    _DA_XA[dA.data_ptr()] = (dA, prod, A)


class _NegExpFn(torch.autograd.Function):
    """A = -exp(A_log) with the value taken from the cache: d A / d A_log = A (one launch in the backward, none when the scan backward
    already formed the product)"""

    @staticmethod
    def forward(ctx, A_log, A):
        ctx.save_for_backward(A)
        ctx.A_log = A_log
        return A.view_as(A)

    @staticmethod
    def backward(ctx, g):
        ent = _DA_XA.pop(g.data_ptr(), None)
        if ent is not None and ent[0].data_ptr() == g.data_ptr() and ent[0].shape == g.shape and ent[0].dtype == g.dtype \
                and ent[2].data_ptr() == ctx.saved_tensors[0].data_ptr():
            return _homed(ent[1], grad_home(ctx.A_log)), None
        return g * ctx.saved_tensors[0], None


def neg_exp(A_log):
    """-exp(A_log.float())  (MS:190, 204, 220)"""
    c = _STEP_CACHE.get(id(A_log))
    if c is not None and c[0] == "A":
        _A_OWNER[c[1].data_ptr()] = A_log
        return _NegExpFn.apply(A_log, c[1])
    return -torch.exp(A_log.float())


# =================================================================================================
# selective_scan_fn  (SSI:14-83)
# =================================================================================================
class SelectiveScanFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False, reverse=False):
        if B.dim() < 3 or C.dim() < 3 or A.is_complex():
            raise NotImplementedError("HIP selective scan implements real A with input-dependent B/C "
                                      "(the only form Mamba/AuM uses, SSI:473-493)")
        fix = lambda t: t if t is None or t.stride(-1) == 1 else t.contiguous()      # SSI:19-30
        u, delta, B, C, z = fix(u), fix(delta), fix(B), fix(C), fix(z)
        ctx.squeeze_B, ctx.squeeze_C = B.dim() == 3, C.dim() == 3
        # long-form rows: the `x` checkpoint of SSI:37-45 (chunk-entry states), kept for the backward
        x_ck = aum_hip.scan_ckpt(u, A.shape[1]) if any(ctx.needs_input_grad) else None
        out, out_pre, last = aum_hip.scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus, reverse,
                                              want_out_pre=z is not None, want_last_state=return_last_state, x_ck=x_ck)
        ctx.delta_softplus, ctx.reverse, ctx.has_z = delta_softplus, reverse, z is not None
        ctx.save_for_backward(u, delta, A, B, C, D, z, delta_bias, out_pre, x_ck)
        if return_last_state:
            ctx.mark_non_differentiable(last)       # SSI:79-82: no gradient through last_state
            return out, last
        return out

    @staticmethod
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, z, delta_bias, out_pre, x_ck = ctx.saved_tensors
        if dout.stride(-1) != 1:
            dout = dout.contiguous()
        g = aum_hip.scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout.to(u.dtype), out_pre, ctx.delta_softplus,
                             ctx.reverse, x_ck=x_ck)
        dB = g["dB"].to(B.dtype)
        dC = g["dC"].to(C.dtype)
        if not ctx.squeeze_B:
            dB = dB.unsqueeze(1)
        if not ctx.squeeze_C:
            dC = dC.unsqueeze(1)
        return (g["du"], g["ddelta"], g["dA"].to(A.dtype), dB, dC,
                g["dD"].to(D.dtype) if D is not None else None,
                g["dz"],
                g["ddelta_bias"].to(delta_bias.dtype) if delta_bias is not None else None,
                None, None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """if return_last_state is True, returns (out, last_state); last_state is (batch, dim, dstate) and carries no
    gradient (SSI:77-83)."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state, False)


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                       return_last_state=False):
    """Pure-PyTorch statement of the recurrence (same contract as SSI:86-152; real A, variable B/C as (B,N,L) or
    (B,G,N,L)).  fp32 internal math, result cast back to u.dtype."""
    dtype_in = u.dtype
    u32, dt = u.float(), delta.float()
    if delta_bias is not None:
        dt = dt + delta_bias.float()[:, None]
    if delta_softplus:
        dt = F.softplus(dt)
    batch, dim, L = u32.shape
    N = A.shape[1]

    def per_channel(M):      # -> (batch, dim, N, L)
        M = M.float()
        if M.dim() == 3:
            return M[:, None].expand(batch, dim, N, L)
        return M.repeat_interleave(dim // M.shape[1], dim=1)

    Bf, Cf = per_channel(B), per_channel(C)
    decay = torch.exp(dt[:, :, None, :] * A.float()[None, :, :, None])      # (batch, dim, N, L)
    drive = (dt * u32)[:, :, None, :] * Bf
    h = torch.zeros(batch, dim, N, dtype=torch.float32, device=u.device)
    ys = torch.empty(batch, dim, L, dtype=torch.float32, device=u.device)
    for t in range(L):
        h = decay[..., t] * h + drive[..., t]
        ys[..., t] = (h * Cf[..., t]).sum(-1)
    out = ys if D is None else ys + u32 * D.float()[:, None]
    if z is not None:
        out = out * F.silu(z.float())
    out = out.to(dtype_in)
    return (out, h) if return_last_state else out


# =================================================================================================
# fused inner blocks  (SSI:155-633)
# =================================================================================================
# The token count of an AuM batch is batch * 513 -- never a multiple of a GEMM tile.  The library's kernels launch one workgroup per
# output tile, so the 64 tokens past 32768 (B = 64) add a whole extra round of workgroups to a 3-7 round GEMM (cold-cache, one
# box: in_proj forward 175 vs 144 us, out_proj data gradient 138 vs 90 us for 32832 vs 32768 tokens, profiles/
# r02_sweep_gemm_tokens.txt).  The four big projection GEMMs are therefore issued as a tile-aligned GEMM over the first
# n0 = 256 * floor(ntok / 256) tokens plus a small one over the remainder, both writing slices of one output.  AUM_GEMM_TOKEN_SPLIT is
# a bit mask for A/B runs (1 in_proj forward, 2 out_proj forward, 4 out_proj data gradient, 8 in_proj data gradient).  Same-box A/B of the
# step: mask 0 / 1 / 2 / 4 / 8 / 15 / 13 = 80.06 / 79.67 / 80.10 / 79.53 / 79.63 / 78.97 / 78.87 ms -> default 13 in rounds 2-4 (the out_proj
# forward loses what it gains to its 17 us remainder GEMM).  Round 5 re-measured it on the token-major block, where the out_proj data gradient
# runs on aum_gemm_tn (ragged rows in its stride) and the in_proj data gradient is a different library call than in round 2
# (profiles/r05_gemm_dispatch_ab.txt, same box x3, ms per step): 13: 61.58 / 61.54 / 61.54, 0: 61.20 / 61.26 / 61.17, 1: 61.17 / 61.26 / 61.19,
# 8: 61.75 / 61.61 / 61.74 -- the in_proj data gradient's split now COSTS 0.4 ms, the in_proj forward's is level -> default 0 (single GEMMs).
_TOKEN_SPLIT = int(_dbg_env("AUM_GEMM_TOKEN_SPLIT", "0"))


def _tok_n0(ntok, bit, t):
    n0 = ntok // 256 * 256
    return n0 if (_TOKEN_SPLIT & bit) and t.is_cuda and ntok >= 8192 and 0 < n0 < ntok else 0


def _mm_tokens_cols(a, bt, bit):
    """a [M, K] @ bt[ntok, K]^T -> [M, ntok] (tokens = output columns)"""
    ntok = bt.shape[0]
    n0 = _tok_n0(ntok, bit, bt)
    if not n0:
        return torch.matmul(a, bt.t())
    out = torch.empty((a.shape[0], ntok), dtype=a.dtype, device=a.device)
    torch.matmul(a, bt[:n0].t(), out=out[:, :n0])
    torch.matmul(a, bt[n0:].t(), out=out[:, n0:])
    return out


def _mm_tokens_rows(at, b, bit):
    """at[K, ntok]^T @ b [K, N] -> [ntok, N] (tokens = output rows); `at` is the channel-major activation"""
    ntok = at.shape[1]
    n0 = _tok_n0(ntok, bit, at)
    if not n0:
        return torch.matmul(at.t(), b)
    out = torch.empty((ntok, b.shape[1]), dtype=at.dtype, device=at.device)
    torch.matmul(at[:, :n0].t(), b, out=out[:n0])
    torch.matmul(at[:, n0:].t(), b, out=out[n0:])
    return out


# split-K counts of the in / out projection weight gradients, in order of preference: the first that divides K = batch * len is
# used (B = 64, L = 513: 6 and 9; the long-form 8 x 4097 tokens: 4 and 8).  Same-box A/B of the step with the counts of round 1
# (4, 8): 82.42 -> 81.63 ms; cold-cache sweep of the two GEMMs incl. their partial sums in profiles/r02_sweep_wgrad_splits.txt
# (out_proj 8 / 9 splits: 123 / 98 us).  AUM_WGRAD_SPLITS="in,out" forces a pair for sweeps.
_WGRAD_SPLITS = ((6, 4, 8, 2), (9, 8, 4, 2))
if _dbg_env("AUM_WGRAD_SPLITS", ""):
    _WGRAD_SPLITS = tuple((int(v),) for v in _dbg_env("AUM_WGRAD_SPLITS", "").split(","))
    if len(_WGRAD_SPLITS) != 2:
        raise ValueError("AUM_WGRAD_SPLITS takes two counts, 'in_proj,out_proj' (e.g. AUM_WGRAD_SPLITS=6,9)")


def _pick_splits(K, prefs):
    return next((s for s in prefs if K % s == 0 and K // s >= 1024), 1)


def split_k_wgrad(a_mk, b_kn, splits, out_dtype=None):
    """a_mk [M, K] @ b_kn [K, N] with a long K (= batch*len tokens) and a small [M, N] result: the weight-gradient GEMMs
    of the in/out projections.  hipBLASLt's best single-GEMM solutions keep the matrix pipe 20-31 % busy on these shapes
    (profiles/r01_mfma_busy.txt: 18-48 output tiles for 256 CUs); splitting K into `splits` batched GEMMs on strided
    views (no copies) and summing the partial products in fp32 fills the chip (sweep on one box, in/out splits: 4/8 85.0,
    8/8 86.1, 16/8 85.5, 4/4 85.8, 4/16 85.4, 2/8 85.9 ms per step; single GEMMs 87.9).  AUM_WGRAD_SPLIT=0 restores the single GEMM.
    out_dtype: the parameter's dtype -- the fp32 sum is handed over as it is instead of being rounded to 16 bits and widened again by
    autograd (two cast launches per weight; the reference's autocast GEMM rounds its weight gradient to 16 bits, this one does not)."""
    K = a_mk.shape[1]
    out_dtype = out_dtype or a_mk.dtype
    if splits <= 1 or K % splits or K // splits < 1024 or not _WGRAD_SPLIT_ON or not a_mk.is_cuda:
        return torch.matmul(a_mk, b_kn).to(out_dtype)
    kc = K // splits
    a3 = a_mk.unflatten(1, (splits, kc)).permute(1, 0, 2)          # [S, M, kc], strided view
    b3 = b_kn.unflatten(0, (splits, kc))                            # [S, kc, N]
    return aum_hip.sum_rows(torch.bmm(a3, b3)).to(out_dtype)


# The weight gradients of the token-major block's in / out projections (d W = d out^T . input, both operands token-major) on the hand-written
# kernel aum_gemm_wgrad (csrc/gemm_kernels.h: transposing LDS reads, fp32 partial tiles over token splits summed by aum_sum_rows) instead of the
# library's strided batched GEMMs.  AUM_DEBUG=1 AUM_WGRAD=lib restores those (A/B runs).
_HIP_WGRAD = _dbg_env("AUM_WGRAD", "hip") != "lib"


def _wgrad_tm(dy2d, x2d, splits_hint, out_dtype, pending=None, home=None):
    """dW [N, K] = dy2d [T, N]^T @ x2d [T, K] for token-major operands.  pending: a _PendingSums of the caller -- the kernel's partial tiles
    join it and the gradient is what its run() returns at the index this function returns (fp32 parameters only).  home: where the sum is
    wanted (grad_home of the parameter)"""
    if _HIP_WGRAD and _HIP_GEMM and dy2d.is_cuda and aum_hip.gemm_wgrad_supported(dy2d, x2d):
        if pending is not None and out_dtype == torch.float32:
            return pending.add(aum_hip.gemm_wgrad(dy2d, x2d, partials=True), out=home)
        return aum_hip.gemm_wgrad(dy2d, x2d, out=home if out_dtype == torch.float32 else None).to(out_dtype)
    return split_k_wgrad(dy2d.t(), x2d, _pick_splits(x2d.shape[0], splits_hint), out_dtype)


class _PendingSums:
    """the partial sets a layer's backward leaves behind (weight-gradient splits, per-wave conv partials), summed by ONE launch at the end of
    the backward function instead of one 5-12 us launch each (aum_hip.sum_rows_multi; every set is added in the order a launch of its own uses)"""

    def __init__(self):
        self.parts, self.tr, self.outs = [], [], []

    def add(self, part, tr_cols=0, out=None):
        self.parts.append(part)
        self.tr.append(tr_cols)
        self.outs.append(out)
        return _PendingIndex(len(self.parts) - 1)

    def run(self):
        return aum_hip.sum_rows_multi(self.parts, self.tr, outs=self.outs) if self.parts else []


class _PendingIndex(int):
    pass


class InProjFn(torch.autograd.Function):
    """xz2d [2E, B*L] = W [2E, D] @ hidden2d[B*L, D]^T (MS:185-189: matmul and BLH -> HBL transpose in one GEMM) with the
    split-K weight gradient above; autocast casts both operands like F.linear would."""

    @staticmethod
    def forward(ctx, weight, hidden2d):
        w = _cast(weight, _autocast_dtype())
        h = hidden2d.to(w.dtype)
        ctx.save_for_backward(w, h)
        ctx.wdtype, ctx.hdtype = weight.dtype, hidden2d.dtype
        return _mm_tokens_cols(w, h, 1)

    @staticmethod
    def backward(ctx, dxz2d):
        w, h = ctx.saved_tensors
        dxz2d = dxz2d.to(w.dtype)
        dh = _mm_tokens_rows(dxz2d, w, 8) if ctx.needs_input_grad[1] else None
        dw = split_k_wgrad(dxz2d, h, _pick_splits(h.shape[0], _WGRAD_SPLITS[0]), ctx.wdtype) if ctx.needs_input_grad[0] else None
        return dw, (None if dh is None else dh.to(ctx.hdtype))


def _mm_rows(a, b, bit):
    """a [ntok, K] @ b [K, N] -> [ntok, N] with the tile-aligned token split of _mm_tokens_rows (tokens = rows of a)"""
    ntok = a.shape[0]
    n0 = _tok_n0(ntok, bit, a)
    if not n0:
        return torch.matmul(a, b)
    out = torch.empty((ntok, b.shape[1]), dtype=a.dtype, device=a.device)
    torch.matmul(a[:n0], b, out=out[:n0])
    torch.matmul(a[n0:], b, out=out[n0:])
    return out


# The K-contiguous projection GEMMs of the token-major block -- in_proj / out_proj forward, and their data gradients against the cached
# transposed weight -- can run on the hand-written MFMA kernel (aum_hip.gemm_tn, csrc/gemm_kernels.h) whenever the operands qualify
# (16-bit, device, widths that are multiples of 256 / 64).  Which of them do is decided by the step, not by the kernel alone
# (profiles/r03_gemm_step_ab.txt, same box, ms per step): library only 67.49; + out_proj data gradient (N = 1536, K = 768) 66.77;
# + in_proj forward (3072, 768) 66.20 -- alone the kernel is 10 % behind the library on that one, but it takes the 64 ragged rows of
# 64 x 513 tokens in its stride where the library needs a second launch (152.6 + 19 us); + out_proj forward (768, 1536) 66.93; all four
# 67.0-70.1 (N = 768 is two tiles per CU: tile quantisation).  The default ("auto") therefore sends the two N >= 1536 shapes
# to the kernel; AUM_DEBUG=1 AUM_GEMM=hip sends all four, AUM_GEMM=lib none, AUM_GEMM_SHAPES="NxK,..." another set (A/B runs).  The weight
# gradients (token-contiguous operands) and everything that does not qualify stay library GEMMs.
_V2_STREAMS = _dbg_env("AUM_V2_STREAMS", "1") != "0"        # Bi-Bi: the second pipeline on a side stream (mamba_simple.py); 0: in line (A/B)
_side_streams = {}
_main_streams = {}
def v2_two_streams(params=(), module=None):
    """Bi-Bi's second pipeline on a side stream?  Autograd runs each pipeline's backward on the stream of its forward, so anything that
    consumes a parameter gradient INSIDE backward sees two producer streams.  DistributedDataParallel's reducer orders a bucket's
    all-reduce behind the stream of the LAST gradient hook only, and a bucket holds gradients of both pipelines: under an initialised
    process group (any world size) the side streams are used only by blocks of a model whose OWN wrapper exchanges gradients through
    `ddp_join_streams_hook` -- `register_ddp_join_streams(ddp)` (what aum.train.compress_gradients calls for models with Bi-Bi blocks)
    registers the hook and marks that wrapper's blocks (`module._aum_streams_joined`); a second wrapper in the same process without the hook,
    or a hook that was built but never registered, leaves its blocks in line.  Parameters that carry post-accumulate-grad hooks (FSDP,
    optimizer-in-backward, user hooks: `params` = every parameter touched inside the two side-stream regions) keep the pipelines in line too."""
    if not _V2_STREAMS:
        return False
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and not getattr(module, "_aum_streams_joined", False):
        return False
    return not any(getattr(p_, "_post_accumulate_grad_hooks", None) for p_ in params)


def side_streams(device):
    """the two streams Bi-Bi's pipelines run on.  BOTH pipelines leave the stream the block was called on: under DistributedDataParallel the
    gradient accumulators of every parameter were created on that stream (the reducer's constructor touches them all), so autograd makes
    it wait for the producer of every parameter gradient -- with one pipeline ON that stream, that pipeline's own backward then queues
    behind the other pipeline's (measured: no gain left from the second stream under DDP); with both pipelines elsewhere the calling
    stream only waits, and the pipelines overlap."""
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    _main_streams[device] = torch.cuda.current_stream(device)       # the stream the block was called on (joins happen there)
    return s


def ddp_join_streams_hook(hook=None):
    """A DistributedDataParallel communication hook that first makes the CURRENT stream (the one the reducer orders the bucket's
    collective behind) wait for the other streams Bi-Bi's backward runs on, then hands the bucket to `hook` (default: the plain
    all-reduce + mean).  Use `register_ddp_join_streams`, which also tells the wrapper's blocks that they may leave the calling stream.
    TT:39, TT:168."""
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    inner = hook or default_hooks.allreduce_hook

    def joined(state, bucket):
        buf = bucket.buffer()
        if buf.is_cuda:
            cur = torch.cuda.current_stream(buf.device)
            for s in tuple(_side_streams.get(buf.device) or ()) + (_main_streams.get(buf.device),):
                if s is not None and s != cur:
                    cur.wait_stream(s)
        return inner(state, bucket)

    return joined


def register_ddp_join_streams(ddp, state=None, hook=None):
    """register `ddp_join_streams_hook(hook)` on THIS DistributedDataParallel wrapper and allow the Bi-Bi blocks it wraps to run their two
    pipelines on side streams (the permission is a flag on each block, not a process-wide latch: ADVICE r5)."""
    ddp.register_comm_hook(state, ddp_join_streams_hook(hook))
    for m in ddp.module.modules():
        if getattr(m, "bimamba_type", None) == "v2":
            m._aum_streams_joined = True


_XDT_BWD_HIP = _dbg_env("AUM_XDT_BWD_LIB", "0") != "1"      # AUM_DEBUG=1 AUM_XDT_BWD_LIB=1: the x_proj / dt_proj gradients as five library calls (A/B)
_XDT_HIP = _dbg_env("AUM_XDT_LIB", "0") != "1"              # AUM_DEBUG=1 AUM_XDT_LIB=1: x_proj as a library GEMM + the dt projection kernel (A/B)
_DTPROJ_HIP = _dbg_env("AUM_DTPROJ_LIB", "0") != "1"        # AUM_DEBUG=1 AUM_DTPROJ_LIB=1: the dt projection back on the library GEMM (A/B)
_GEMM_MODE = _dbg_env("AUM_GEMM", "auto")
if _GEMM_MODE not in ("auto", "hip", "lib"):
    raise ValueError("AUM_GEMM takes auto, hip or lib")
_HIP_GEMM = _GEMM_MODE != "lib"
# (N, K) of aum_gemm_tn calls that make the STEP faster than the library's solutions do.  Round 5 re-measured the in_proj forward (3072, 768), the default
# since round 3 on a step A/B "level within the box spread": same box, three alternating rounds (profiles/r05_gemm_dispatch_ab.txt), ms per step --
# out_proj data gradient only 62.62 / 62.54 / 62.54, both 62.67 / 62.73 / 62.88, in_proj forward only 63.65 / 63.63 / 63.40, library only
# 63.14 / 63.12 / 62.99: the kernel is 5-8 % behind the library standalone on that shape (155-160 vs 146-148 us) and does not win it back in the
# step.  The library keeps it; the out_proj data gradient (84 vs 91 us standalone, -0.5 ms in the step) stays on the kernel.
_HIP_GEMM_FASTER = {(1536, 768), (3072, 768)}
if _dbg_env("AUM_GEMM_SHAPES", ""):         # A/B runs: another set, "NxK,NxK"
    _HIP_GEMM_FASTER = {tuple(int(v) for v in sh.split("x")) for sh in _dbg_env("AUM_GEMM_SHAPES", "").replace("+", ",").split(",")}


def _hip_gemm_ok(a, n, k):
    return (_HIP_GEMM and a.is_cuda and a.dtype in (torch.bfloat16, torch.float16) and n % aum_hip.GEMM_BN == 0 and k % aum_hip.GEMM_BK == 0
            and (_GEMM_MODE == "hip" or (n, k) in _HIP_GEMM_FASTER))


def _gemm_rows(a, w_nk, bit):
    """a [ntok, K] @ w_nk [N, K]^T -> [ntok, N]"""
    if _hip_gemm_ok(a, w_nk.shape[0], w_nk.shape[1]) and aum_hip.gemm_tn_supported(a, w_nk):
        return aum_hip.gemm_tn(a, w_nk)
    return _mm_rows(a, w_nk.t(), bit)


def _weight_t_for_dgrad(w_param, w_cast, a_is_cuda):
    """the (K, N) -> (N, K) transposed 16-bit copy of a projection weight for the data-gradient GEMM on the MFMA kernel (from the step
    cache when a model filled it), or None when that GEMM stays with the library (which takes the weight as stored)"""
    n, k = w_cast.shape[1], w_cast.shape[0]          # the data gradient multiplies by w (k = out features, n = in features)
    if not (_HIP_GEMM and a_is_cuda and w_cast.dtype in (torch.bfloat16, torch.float16) and n % aum_hip.GEMM_BN == 0 and k % aum_hip.GEMM_BK == 0
            and (_GEMM_MODE == "hip" or (n, k) in _HIP_GEMM_FASTER)):
        return None
    return _cast_t(w_param, w_cast.dtype)


def _gemm_dgrad(g, w_cast, w_t, bit):
    """g [ntok, K] @ w_cast [K, N] -> [ntok, N]; w_t = w_cast^T contiguous or None"""
    if w_t is not None and aum_hip.gemm_tn_supported(g, w_t):
        return aum_hip.gemm_tn(g, w_t)
    return _mm_rows(g, w_cast, bit)


class InProjTmFn(torch.autograd.Function):
    """xz2d [B*L, 2E] = hidden2d [B*L, D] @ W^T: the token-major form of InProjFn (MS:185-189 without the BLH -> HBL transpose: the
    token-major kernels read the rows as they are)."""

    @staticmethod
    def forward(ctx, weight, hidden2d):
        w = _cast(weight, _autocast_dtype())
        h = hidden2d.to(w.dtype)
        w_t = _weight_t_for_dgrad(weight, w, h.is_cuda) if ctx.needs_input_grad[1] else None
        ctx.save_for_backward(w, h, w_t)
        ctx.wdtype, ctx.hdtype = weight.dtype, hidden2d.dtype
        ctx.wparam = weight
        return _gemm_rows(h, w, 1)

    @staticmethod
    def backward(ctx, dxz2d):
        w, h, w_t = ctx.saved_tensors
        dxz2d = dxz2d.to(w.dtype)
        dh = _gemm_dgrad(dxz2d, w, w_t, 8) if ctx.needs_input_grad[1] else None
        home = grad_home(ctx.wparam)
        dw = _homed(_wgrad_tm(dxz2d, h, _WGRAD_SPLITS[0], ctx.wdtype, home=home), home) if ctx.needs_input_grad[0] else None
        return dw, (None if dh is None else dh.to(ctx.hdtype))


class OutProjTmFn(torch.autograd.Function):
    """out2d [B*L, D] = y2d [B*L, E] @ W_out^T on token-major rows (SSI:517 for the Bi-Bi block, whose two pipelines share one out_proj,
    MS:240-246): the same GEMM dispatch as inside the fused inner functions -- forward and data gradient on aum_gemm_tn where the
    step is faster with it, the weight gradient as split-K batches."""

    @staticmethod
    def forward(ctx, weight, y2d):
        w = _cast(weight, _autocast_dtype())
        y = y2d.to(w.dtype)
        w_t = _weight_t_for_dgrad(weight, w, y.is_cuda) if ctx.needs_input_grad[1] else None
        ctx.save_for_backward(w, y, w_t)
        ctx.wdtype, ctx.ydtype = weight.dtype, y2d.dtype
        ctx.wparam = weight
        return _gemm_rows(y, w, 2)

    @staticmethod
    def backward(ctx, dout2d):
        w, y, w_t = ctx.saved_tensors
        dout2d = dout2d.to(w.dtype)
        dy = _gemm_dgrad(dout2d, w, w_t, 4) if ctx.needs_input_grad[1] else None
        home = grad_home(ctx.wparam)
        dw = _homed(_wgrad_tm(dout2d, y, _WGRAD_SPLITS[1], ctx.wdtype, home=home), home) if ctx.needs_input_grad[0] else None
        return dw, (None if dy is None else dy.to(ctx.ydtype))


def token_major_ok(d_inner, d_state, d_conv, dt_rank, dtype=None):
    """shapes the token-major kernels take (include/aum_hip.h: aum_scan_tm_*, aum_conv1d_tm_*); B and C are read in place from the
    x_proj output rows, so the dt block in front of them has to keep them 16-byte aligned"""
    es = 2 if dtype in (torch.bfloat16, torch.float16) else 4
    return (aum_hip.scan_tm_supported(d_inner, d_state) and d_conv <= 4 and (dt_rank * es) % 16 == 0
            and ((dt_rank + 2 * d_state) * es) % 16 == 0)


# The time-serial kernels give one wave to 64 channels of one batch entry and direction and walk the WHOLE sequence with it: they need
# batch * (d_inner / 64) * directions waves to fill 256 CUs x 4 SIMDs -- 1.5 per SIMD for the forward (three resident), 2 per SIMD when a
# backward follows (two resident at 256 VGPRs).  Short of that (long-form clips at batch 8, single-clip inference, AuM-Small training at
# batch 64) the chunk-parallel channel-major kernels are the better division: AuM-Small, B = 64 (1536 waves), same box: training 41.2 ms
# token-major vs 39.7 ms channel-major, forward only 11.3 vs 11.9 ms (profiles/r03_variants_bench.json).  AUM_TM_MIN_WAVES overrides
# the forward threshold (the training one is 4/3 of it).
_TM_MIN_WAVES = int(_dbg_env("AUM_TM_MIN_WAVES", "1536"))


# Long rows at a small batch (the long-form clips: B = 8, L = 4097) are cut into time segments that run as waves of their own
# (aum_hip.scan_tm_segments, aum_scan_tm_seg_*): AUM_TM_SEGMENTS=0 keeps them on the channel-major kernels, a number > 1 forces it.
_TM_SEGMENTS = int(_dbg_env("AUM_TM_SEGMENTS", "-1"))


def tm_segments(batch, d_inner, seqlen, bidirectional, training, device=None):
    if _TM_SEGMENTS == 0 or seqlen is None:
        return 1
    if _TM_SEGMENTS > 1:
        return min(_TM_SEGMENTS, aum_hip.SCAN_TM_MAX_SEGMENTS)
    return aum_hip.scan_tm_segments(batch, d_inner, seqlen, bidirectional, training, device=device)


def token_major_preferred(batch, d_inner, bidirectional, training=None, seqlen=None):
    training = torch.is_grad_enabled() if training is None else training
    # (the 4/3 applies to the Fo-Bi backward -- three direction pairs per workgroup in three stages; a one-direction launch of 1536 waves
    # trains faster token-major: Fo-Fo AuM-Base at batch 64, 54.7 ms against 60.4 ms channel-major, profiles/r03 / r04_variants_bench.json)
    need = -(-_TM_MIN_WAVES * 4 // 3) if training and bidirectional else _TM_MIN_WAVES
    if batch * (d_inner // 64) * (2 if bidirectional else 1) >= need:
        return True
    return tm_segments(batch, d_inner, seqlen, bidirectional, training) > 1


def _is_tm(xz):
    """(B, 2E, L) logical tensor stored token-major: the transpose view of a (B, L, 2E) tensor with contiguous rows"""
    return xz.dim() == 3 and xz.stride(1) == 1 and xz.shape[1] > 1


def _inner_forward_tm(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A, A_b, D,
                      delta_bias, delta_softplus, reverse):
    """_inner_forward on token-major activations: every tensor is (B, L, X) with X contiguous.
      conv (SSI:463) -> x_dbl = conv_out W_x^T (SSI:467) -> delta = x_dbl[:, :R] W_dt^T (SSI:468) -> B, C = column blocks of x_dbl read in
      place (SSI:473-493) -> one selective-scan launch, both directions (SSI:499-507) -> out_proj (SSI:517)."""
    act = _autocast_dtype()
    ctx.out_proj_wdtype = out_proj_weight.dtype if out_proj_weight is not None else None
    x_proj_param, delta_proj_param = x_proj_weight, delta_proj_weight
    x_proj_weight, delta_proj_weight = _cast(x_proj_weight, act), _cast(delta_proj_weight, act)
    out_proj_param = out_proj_weight
    out_proj_weight, out_proj_bias = _cast(out_proj_weight, act), _cast(out_proj_bias, act)
    out_proj_wt = None
    if out_proj_weight is not None and any(ctx.needs_input_grad):
        out_proj_wt = _weight_t_for_dgrad(out_proj_param, out_proj_weight.to(xz.dtype), xz.is_cuda)
    xz_t = xz.transpose(1, 2)                                      # (B, L, 2E), rows contiguous
    Bsz, L, two_e = xz_t.shape
    E = two_e // 2
    R = delta_proj_weight.shape[1]
    N = A.shape[-1]
    x, z = xz_t[:, :, :E], xz_t[:, :, E:]
    conv_w = conv1d_weight.reshape(E, -1)
    conv_out = aum_hip.conv1d_tm_fwd(x, conv_w, conv1d_bias, True, reverse)                 # SSI:463  (B, L, E)
    conv2d = conv_out.view(Bsz * L, E)
    w_x, w_dt = x_proj_weight.to(conv2d.dtype), delta_proj_weight.to(conv2d.dtype)
    if _XDT_HIP and conv2d.is_cuda and aum_hip.xdt_tm_supported(conv2d, w_x, w_dt):
        x_dbl, delta = aum_hip.xdt_tm_fwd(conv2d, w_x, w_dt)                                 # SSI:467-468 in one pass over conv_out
    else:
        x_dbl = torch.matmul(conv2d, w_x.t())                                                # SSI:467  (BL, R+2N)
        delta = None
    if delta is not None:
        pass
    elif _DTPROJ_HIP and x_dbl.is_cuda and aum_hip.dtproj_tm_supported(x_dbl, R, w_dt):
        delta = aum_hip.dtproj_tm_fwd(x_dbl, R, w_dt)                                        # SSI:468  (BL, E): the write-bound MFMA kernel
    else:
        delta = torch.matmul(x_dbl[:, :R], w_dt.t())
    x3 = x_dbl.view(Bsz, L, R + 2 * N)
    Bm, Cm = x3[:, :, R:R + N], x3[:, :, R + N:]                                            # SSI:479  views, no copies
    need_bwd = any(ctx.needs_input_grad)
    # the backward's row pass (aum_xdt_tm_bwd) multiplies by the two small weights the other way round: their transposes, from the
    # step cache when the model filled it
    w_x_t = w_dt_t = None
    if need_bwd and _XDT_BWD_HIP and conv2d.is_cuda and R + 2 * N == aum_hip.XDT_COLS and R == aum_hip.XDT_COLS - 32 and E % 256 == 0 \
            and E <= aum_hip.XDT_MAX_DIM and conv2d.dtype in (torch.bfloat16, torch.float16):
        w_x_t, w_dt_t = _cast_t(x_proj_param, conv2d.dtype), _cast_t(delta_proj_param, conv2d.dtype)
    ckpt = aum_hip.scan_tm_ckpt(Bsz, L, E, N, A_b is not None, xz.device, dtype=conv_out.dtype) if need_bwd else None
    waves = Bsz * (E // 64) * (2 if A_b is not None else 1)
    cut = waves < (-(-_TM_MIN_WAVES * 4 // 3) if need_bwd and A_b is not None else _TM_MIN_WAVES)
    out_z, out_pre = aum_hip.scan_tm_fwd(conv_out, delta.view(Bsz, L, E), A, Bm, Cm, D, z, delta_bias, delta_softplus,
                                         reverse if A_b is None else False, A_b=A_b, want_out_pre=need_bwd, ckpt=ckpt,
                                         segments=tm_segments(Bsz, E, L, A_b is not None, False, device=xz.device) if cut else 1)
    ctx.tm_cut = cut
    # the scan backward forms d A .* A only for an A that neg_exp took out of THIS forward's cache (its _NegExpFn node is the consumer)
    ctx.A_cached = (need_bwd and A.dtype == torch.float32 and A.is_contiguous() and A.data_ptr() in _A_CACHE_PTRS
                    and (A_b is None or (A_b.is_contiguous() and A_b.data_ptr() in _A_CACHE_PTRS)))
    ctx.tm = True
    # the parameters behind this call's gradients (grad_home looks at them in the backward)
    ctx.params = dict(conv_w=conv1d_weight, conv_b=conv1d_bias, x_proj=x_proj_param, dt_proj=delta_proj_param, out_proj=out_proj_param, D=D,
                      dt_bias=delta_bias, A_log=_A_OWNER.get(A.data_ptr()), A_b_log=None if A_b is None else _A_OWNER.get(A_b.data_ptr()))
    ctx.delta_softplus, ctx.reverse = delta_softplus, reverse
    ctx.has_out_proj = out_proj_weight is not None
    ctx.out_proj_bias_is_None = out_proj_bias is None
    ctx.save_for_backward(xz, conv_w, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, out_proj_weight, conv_out, delta, A, A_b, D,
                          delta_bias, out_pre, out_z, ckpt, out_proj_wt, w_x_t, w_dt_t)
    if out_proj_weight is None:
        return out_z.transpose(1, 2)                                                         # SSI:224  (B, E, L) logical
    out = _gemm_rows(out_z.view(Bsz * L, E), out_proj_weight.to(out_z.dtype), 2)             # SSI:517
    if out_proj_bias is not None:
        out = out + out_proj_bias
    return out.reshape(Bsz, L, -1)


def _inner_backward_tm(ctx, dout):
    (xz, conv_w, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, out_proj_weight, conv_out, delta, A, A_b, D, delta_bias, out_pre,
     out_z, ckpt, out_proj_wt, w_x_t, w_dt_t) = ctx.saved_tensors
    xz_t = xz.transpose(1, 2)
    Bsz, L, two_e = xz_t.shape
    E = two_e // 2
    R = delta_proj_weight.shape[1]
    N = A.shape[-1]
    x, z = xz_t[:, :, :E], xz_t[:, :, E:]
    dxz_t = torch.empty((Bsz, L, two_e), dtype=xz.dtype, device=xz.device)                   # SSI:537
    dx, dz = dxz_t[:, :, :E], dxz_t[:, :, E:]
    dout_proj_weight = dout_proj_bias = None
    pend = _PendingSums()
    H = {k: grad_home(v) for k, v in getattr(ctx, "params", {}).items()}          # where each parameter's gradient is wanted (or None)
    H = {k: H.get(k) for k in ("conv_w", "conv_b", "x_proj", "dt_proj", "out_proj", "D", "dt_bias", "A_log", "A_b_log")}
    if ctx.has_out_proj:
        dout2 = dout.reshape(Bsz * L, -1).to(out_proj_weight.dtype)
        dout_z = _gemm_dgrad(dout2, out_proj_weight, out_proj_wt, 4).view(Bsz, L, E).to(conv_out.dtype)     # SSI:540
        dout_proj_weight = _wgrad_tm(dout2, out_z.view(Bsz * L, E), _WGRAD_SPLITS[1], ctx.out_proj_wdtype, pend, home=H["out_proj"])     # SSI:563
        dout_proj_bias = dout2.sum(0) if not ctx.out_proj_bias_is_None else None
    else:
        dout_z = dout.transpose(1, 2)
        dout_z = (dout_z if dout_z.stride(2) == 1 else dout_z.contiguous()).to(xz.dtype)
    x3 = x_dbl.view(Bsz, L, R + 2 * N)
    g = aum_hip.scan_tm_bwd(conv_out, delta.view(Bsz, L, E), A, x3[:, :, R:R + N], x3[:, :, R + N:], D, z, delta_bias, dout_z, out_pre,
                            ckpt, ctx.delta_softplus, ctx.reverse if A_b is None else False, A_b=A_b, dz_out=dz,
                            segments=tm_segments(Bsz, E, L, A_b is not None, True, device=conv_out.device) if ctx.tm_cut else 1,
                            want_dA_xA=ctx.A_cached,
                            param_out=dict(dD=H["D"], ddelta_bias=H["dt_bias"], dA_xA=H["A_log"], dA_b_xA=H["A_b_log"]))        # SSI:541-561
    if ctx.A_cached:            # A came out of the forward's cache (neg_exp): its d A_log is ready (see _NegExpFn)
        _da_xa_put(g["dA"], g["dA_xA"], A)
        if A_b is not None:
            _da_xa_put(g["dA_b"], g["dA_b_xA"], A_b)
    du2, ddelta2 = g["du"].view(Bsz * L, E), g["ddelta"].view(Bsz * L, E)
    dbc2 = g["dBC"].view(Bsz * L, 2 * N)
    conv2d = conv_out.view(Bsz * L, E)
    if w_x_t is not None and aum_hip.xdt_tm_bwd_supported(ddelta2, dbc2, w_dt_t, w_x_t, du2):
        # SSI:570-574, 587, 590 in one pass over ddelta and du (aum_xdt_tm_bwd); the two weight gradients on the skinny form of the
        # weight-gradient kernel (aum_gemm_wgrad, k = 48 / 80): every activation tensor is read once per product
        dx_dbl = aum_hip.xdt_tm_bwd(ddelta2, dbc2, w_dt_t, w_x_t, du2)
        # (a shape the kernel's own argument check would refuse -- e.g. a token split beyond 32-bit byte offsets -- takes the library's
        # split-K products instead of raising inside backward)
        splits = _pick_splits(Bsz * L, _WGRAD_SPLITS[1])
        if aum_hip.gemm_wgrad_supported(ddelta2, x_dbl[:, :R]) and aum_hip.gemm_wgrad_supported(conv2d, dx_dbl):
            # the partial sets join the function's one sum launch, which also stores the x_proj gradient in the parameter's (R + 2N, E) layout
            ddelta_proj_weight = pend.add(aum_hip.gemm_wgrad(ddelta2, x_dbl[:, :R], partials=True), out=H["dt_proj"])          # SSI:586  (E, R) fp32
            dx_proj_weight = pend.add(aum_hip.gemm_wgrad(conv2d, dx_dbl, partials=True), R + 2 * N, out=H["x_proj"])          # SSI:589  (R + 2N, E) fp32
        else:
            ddelta_proj_weight = split_k_wgrad(ddelta2.t(), x_dbl[:, :R], splits, torch.float32)
            dx_proj_weight = split_k_wgrad(dx_dbl.t(), conv2d, splits, torch.float32)
    else:
        dx_dbl = torch.empty_like(x_dbl)
        dx_dbl[:, R:].copy_(dbc2)                                                            # SSI:570-574
        dx_dbl[:, :R].copy_(torch.matmul(ddelta2, delta_proj_weight.to(ddelta2.dtype)))      # SSI:587
        splits = _pick_splits(Bsz * L, _WGRAD_SPLITS[1])
        ddelta_proj_weight = split_k_wgrad(ddelta2.t(), x_dbl[:, :R], splits, torch.float32)     # SSI:586
        dx_proj_weight = split_k_wgrad(dx_dbl.t(), conv2d, splits, torch.float32)            # SSI:589
        du2.addmm_(dx_dbl, x_proj_weight.to(dx_dbl.dtype))                                   # SSI:590
    _, dconv_w, dconv_b = aum_hip.conv1d_tm_bwd(x, conv_w, conv1d_bias, g["du"], True, ctx.reverse, dx_out=dx, partials=True)   # SSI:594
    dconv_w, dconv_b = pend.add(dconv_w, out=H["conv_w"]), (None if dconv_b is None else pend.add(dconv_b, out=H["conv_b"]))
    sums = pend.run()
    dout_proj_weight, ddelta_proj_weight, dx_proj_weight, dconv_w, dconv_b = (
        sums[v] if isinstance(v, _PendingIndex) else v for v in (dout_proj_weight, ddelta_proj_weight, dx_proj_weight, dconv_w, dconv_b))
    return dict(dxz=dxz_t.transpose(1, 2), dconv_w=_homed(dconv_w.reshape(E, 1, -1), H["conv_w"]), dconv_b=_homed(dconv_b, H["conv_b"]),
                dx_proj_w=_homed(dx_proj_weight, H["x_proj"]), ddt_proj_w=_homed(ddelta_proj_weight, H["dt_proj"]),
                dout_proj_w=_homed(dout_proj_weight, H["out_proj"]), dout_proj_b=dout_proj_bias,
                dA=g["dA"], dA_b=g.get("dA_b"), dD=_homed(g["dD"], H["D"]), ddelta_bias=_homed(g["ddelta_bias"], H["dt_bias"]),
                dB_proj_bias=None, dC_proj_bias=None)


def _dm2d(t):
    """(B, E, L) channel-major tensor -> its [E, B*L] 2-D view (no copy)."""
    Bsz, E, L = t.shape
    return t.permute(1, 0, 2).reshape(E, Bsz * L)


def _is_dmajor(t):
    Bsz, E, L = t.shape
    return t.stride(2) == 1 and t.stride(0) == L and t.stride(1) == Bsz * L


def _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                   out_proj_bias, A, A_b, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, reverse):
    if A.is_complex():
        raise NotImplementedError("real A only (SSI:502 asserts the same for the bidirectional path)")
    ctx.tm = False
    # (AUM_REF_DZ_DROP needs the two directions' gate inputs separately: only the channel-major block keeps them, so the option selects it)
    if (TOKEN_MAJOR and not (_REF_DZ_DROP and A_b is not None) and _is_tm(xz) and B_proj_bias is None and C_proj_bias is None
            and xz.stride(2) == xz.shape[1]
            and token_major_ok(xz.shape[1] // 2, A.shape[-1], conv1d_weight.shape[-1], delta_proj_weight.shape[1], xz.dtype)
            and aum_hip.conv1d_tm_supported(xz.transpose(1, 2)[:, :, :xz.shape[1] // 2], conv1d_weight.shape[-1])):
        return _inner_forward_tm(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                                 A, A_b, D, delta_bias, delta_softplus, reverse)
    act = _autocast_dtype()
    ctx.out_proj_wdtype = out_proj_weight.dtype if out_proj_weight is not None else None
    # SSI:452-457: only the projection weights are cast.  The transposes feed the data-gradient kernel (SSI:587, 590).
    x_proj_wt = _cast_t(x_proj_weight, act) if any(ctx.needs_input_grad) else None
    delta_proj_wt = _cast_t(delta_proj_weight, act) if any(ctx.needs_input_grad) else None
    x_proj_weight, delta_proj_weight = _cast(x_proj_weight, act), _cast(delta_proj_weight, act)
    out_proj_weight, out_proj_bias = _cast(out_proj_weight, act), _cast(out_proj_bias, act)
    if xz.stride(-1) != 1:
        xz = xz.contiguous()
    Bsz, two_e, L = xz.shape
    E = two_e // 2
    R = delta_proj_weight.shape[1]
    N = A.shape[-1]
    x, z = xz[:, :E], xz[:, E:]
    conv_w = conv1d_weight.reshape(E, -1)
    conv_out = aum_hip.conv1d_fwd(x, conv_w, conv1d_bias, True, reverse, dmajor=True)        # SSI:463
    conv2d = _dm2d(conv_out)                                                                 # [E, BL]
    # the MFMA projection kernels (16-bit activations, no B/C projection bias -- the only form Mamba uses): x_dbl stays
    # channel-major [R+2N][BL], so B and C are row blocks of it (batch stride L, state stride BL) -- no slicing copies
    ctx.proj_kernels = (B_proj_bias is None and C_proj_bias is None and x_proj_weight.dtype == conv2d.dtype
                        and delta_proj_weight.dtype == conv2d.dtype and conv2d.is_contiguous()
                        and aum_hip.proj_supported(E, R, N, Bsz * L, conv2d.dtype))
    if ctx.proj_kernels:
        x_dbl, delta = aum_hip.proj_fwd(conv2d, x_proj_weight.contiguous(), delta_proj_weight.contiguous(), N)   # SSI:467-468
        delta = delta.view(E, Bsz, L).permute(1, 0, 2)
        Bm = x_dbl[R:R + N].view(N, Bsz, L).permute(1, 0, 2)                                 # SSI:479  (B, N, L) view
        Cm = x_dbl[R + N:].view(N, Bsz, L).permute(1, 0, 2)
    else:
        x_dbl = torch.matmul(conv2d.t(), x_proj_weight.t().to(conv2d.dtype))                 # SSI:467  (BL, R+2N)
        delta = torch.matmul(delta_proj_weight.to(x_dbl.dtype), x_dbl[:, :R].t())            # SSI:468  [E, BL]
        delta = delta.reshape(E, Bsz, L).permute(1, 0, 2)
        Bm = x_dbl[:, R:R + N]
        Cm = x_dbl[:, R + N:R + 2 * N]
        if B_proj_bias is not None:
            Bm = Bm + B_proj_bias.to(Bm.dtype)
        if C_proj_bias is not None:
            Cm = Cm + C_proj_bias.to(Cm.dtype)
        Bm = Bm.reshape(Bsz, L, N).transpose(1, 2).contiguous()                              # SSI:479  (B, N, L)
        Cm = Cm.reshape(Bsz, L, N).transpose(1, 2).contiguous()
    # the pre-gate sum is only needed by the backward: skipped (one 2-byte write per element and layer) when no input of
    # this call requires a gradient, i.e. in inference
    need_bwd = any(ctx.needs_input_grad)
    bidir_fused = A_b is not None and L <= aum_hip.get().max_single_pass_len
    if A_b is None or bidir_fused:
        ck_f = aum_hip.scan_ckpt(conv_out, A.shape[1]) if need_bwd and A_b is None else None    # long one-direction rows only
        out_z, out_pre, _ = aum_hip.scan_fwd(conv_out, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus,
                                             reverse if A_b is None else False,        # Fo-Bi: the directions are A (forward), A_b (reverse)
                                             A_b=A_b, want_out_pre=need_bwd, dmajor=True, x_ck=ck_f)   # SSI:499-507, one launch
        out_pre_b = ck_b = None
    else:   # long rows: two reverse-flag launches, still no flip copies; the chunked kernels checkpoint the chunk-entry states
        ck_f = aum_hip.scan_ckpt(conv_out, A.shape[1]) if need_bwd else None
        ck_b = aum_hip.scan_ckpt(conv_out, A.shape[1]) if need_bwd else None
        of, out_pre, _ = aum_hip.scan_fwd(conv_out, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, False,
                                          want_out_pre=need_bwd, dmajor=True, x_ck=ck_f)
        acc = aum_hip.scan_accumulates(conv_out, A.shape[1])      # the reverse launch adds to the forward launch's output
        ob, out_pre_b, _ = aum_hip.scan_fwd(conv_out, delta, A_b, Bm, Cm, D, z, delta_bias, delta_softplus, True,
                                            want_out_pre=need_bwd, dmajor=True, x_ck=ck_b, accumulate_into=of if acc else None)
        out_z = ob if acc else of + ob
    ctx.delta_softplus, ctx.reverse, ctx.bidir_fused = delta_softplus, reverse, bidir_fused
    ctx.has_out_proj = out_proj_weight is not None
    ctx.out_proj_bias_is_None = out_proj_bias is None
    ctx.B_proj_bias_is_None, ctx.C_proj_bias_is_None = B_proj_bias is None, C_proj_bias is None
    ctx.save_for_backward(xz, conv_w, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, out_proj_weight,
                          conv_out, delta, A, A_b, Bm, Cm, D, delta_bias, out_pre, out_pre_b, out_z, ck_f, ck_b,
                          x_proj_wt, delta_proj_wt)
    if out_proj_weight is None:
        return out_z                                                                          # SSI:224
    out = _mm_tokens_rows(_dm2d(out_z), out_proj_weight.t(), 2)                               # SSI:517
    if out_proj_bias is not None:
        out = out + out_proj_bias
    return out.reshape(Bsz, L, -1)


def _inner_backward(ctx, dout):
    if ctx.tm:
        return _inner_backward_tm(ctx, dout)
    (xz, conv_w, conv1d_bias, x_dbl, x_proj_weight, delta_proj_weight, out_proj_weight, conv_out, delta, A, A_b, Bm,
     Cm, D, delta_bias, out_pre, out_pre_b, out_z, ck_f, ck_b, x_proj_wt, delta_proj_wt) = ctx.saved_tensors
    Bsz, two_e, L = xz.shape
    E = two_e // 2
    R = delta_proj_weight.shape[1]
    N = A.shape[-1]
    x, z = xz[:, :E], xz[:, E:]
    dxz = torch.empty_like(xz)                      # same (channel-major) strides as xz      SSI:537
    dx, dz = dxz[:, :E], dxz[:, E:]
    dout_proj_weight = dout_proj_bias = None
    if ctx.has_out_proj:
        dout2 = dout.reshape(Bsz * L, -1).to(out_proj_weight.dtype)
        dout_z = _mm_tokens_cols(out_proj_weight.t(), dout2, 4).reshape(E, Bsz, L).permute(1, 0, 2)   # SSI:540
        dout_proj_weight = split_k_wgrad(dout2.t(), _dm2d(out_z).t(), _pick_splits(dout2.shape[0], _WGRAD_SPLITS[1]),
                                         ctx.out_proj_wdtype)                                                  # SSI:563
        dout_proj_bias = dout2.sum(0) if not ctx.out_proj_bias_is_None else None
    else:
        dout_z = dout if dout.stride(-1) == 1 else dout.contiguous()
        dout_z = dout_z.to(xz.dtype)
    drop = _REF_DZ_DROP and A_b is not None
    if A_b is None or ctx.bidir_fused:
        g = aum_hip.scan_bwd(conv_out, delta, A, Bm, Cm, D, z, delta_bias, dout_z, out_pre, ctx.delta_softplus,
                             ctx.reverse if A_b is None else False, A_b=A_b, dz_out=dz, dmajor=True, x_ck=ck_f)        # SSI:541-561, one launch
    else:
        g = aum_hip.scan_bwd(conv_out, delta, A, Bm, Cm, D, z, delta_bias, dout_z, out_pre, ctx.delta_softplus, False,
                             dz_out=dz, dmajor=True, x_ck=ck_f)
        acc = not drop and aum_hip.scan_accumulates(conv_out, A.shape[1])   # SSI:554-559 inside the reverse launch
        gb = aum_hip.scan_bwd(conv_out, delta, A_b, Bm, Cm, D, z, delta_bias, dout_z, out_pre_b, ctx.delta_softplus,
                              True, dmajor=True, x_ck=ck_b, accumulate_into=g if acc else None)
        if not acc:
            for k in ("du", "ddelta", "dB", "dC", "dD", "ddelta_bias"):
                g[k] = g[k] + gb[k]
            if not drop:
                dz.add_(gb["dz"])
        g["dA_b"] = gb["dA"]
    if drop and ctx.bidir_fused:
        # reproduce SSI:560/599 for A/B runs against a CUDA run of the reference: keep only the forward term
        gf = aum_hip.scan_bwd(conv_out, delta, A, Bm, Cm, D, z, delta_bias, dout_z,
                              aum_hip.scan_fwd(conv_out, delta, A, Bm, Cm, D, z, delta_bias, ctx.delta_softplus, False,
                                               want_out_pre=True, dmajor=True)[1],
                              ctx.delta_softplus, False, dz_out=dz, dmajor=True)
        del gf
    dconv_out, ddelta = g["du"], g["ddelta"]
    if ctx.proj_kernels:
        ddelta2, dconv2 = _dm2d(ddelta), _dm2d(dconv_out)
        dx_dbl = aum_hip.proj_bwd_data(ddelta2, delta_proj_wt, x_proj_wt, g["dB"], g["dC"], dconv2, L)   # SSI:570-574, 587, 590
        ddelta_proj_weight = aum_hip.proj_bwd_weight(ddelta2, x_dbl[:R], False)              # SSI:586
        dx_proj_weight = aum_hip.proj_bwd_weight(_dm2d(conv_out), dx_dbl, True)              # SSI:589
        _, dconv_w, dconv_b = aum_hip.conv1d_bwd(x, conv_w, conv1d_bias, dconv_out, True, ctx.reverse, dx_out=dx)  # SSI:594
        return dict(dxz=dxz, dconv_w=dconv_w.reshape(E, 1, -1), dconv_b=dconv_b, dx_proj_w=dx_proj_weight,
                    ddt_proj_w=ddelta_proj_weight, dout_proj_w=dout_proj_weight, dout_proj_b=dout_proj_bias,
                    dA=g["dA"], dA_b=g.get("dA_b"), dD=g["dD"], ddelta_bias=g["ddelta_bias"],
                    dB_proj_bias=None, dC_proj_bias=None)
    dx_dbl = torch.empty_like(x_dbl)
    dx_dbl3 = dx_dbl.view(Bsz, L, -1)
    dx_dbl3[:, :, R:R + N].copy_(g["dB"].transpose(1, 2))                                    # SSI:570-574
    dx_dbl3[:, :, R + N:R + 2 * N].copy_(g["dC"].transpose(1, 2))
    dB_proj_bias = g["dB"].sum((0, 2)) if not ctx.B_proj_bias_is_None else None
    dC_proj_bias = g["dC"].sum((0, 2)) if not ctx.C_proj_bias_is_None else None
    ddelta2 = _dm2d(ddelta)                                                                   # [E, BL]
    ddelta_proj_weight = torch.matmul(ddelta2, x_dbl[:, :R])                                  # SSI:586
    dx_dbl[:, :R] = torch.matmul(ddelta2.t(), delta_proj_weight.to(ddelta2.dtype))            # SSI:587
    dx_proj_weight = torch.matmul(dx_dbl.t(), _dm2d(conv_out).t())                            # SSI:589
    dconv2 = _dm2d(dconv_out)
    dconv2.addmm_(x_proj_weight.t().to(dx_dbl.dtype), dx_dbl.t())                             # SSI:590
    _, dconv_w, dconv_b = aum_hip.conv1d_bwd(x, conv_w, conv1d_bias, dconv_out, True, ctx.reverse, dx_out=dx)  # SSI:594
    return dict(dxz=dxz, dconv_w=dconv_w.reshape(E, 1, -1), dconv_b=dconv_b, dx_proj_w=dx_proj_weight,
                ddt_proj_w=ddelta_proj_weight, dout_proj_w=dout_proj_weight, dout_proj_b=dout_proj_bias,
                dA=g["dA"], dA_b=g.get("dA_b"), dD=g["dD"], ddelta_bias=g["ddelta_bias"],
                dB_proj_bias=dB_proj_bias, dC_proj_bias=dC_proj_bias)


def _check_variable_bc(B, C):
    if B is not None or C is not None:
        raise NotImplementedError("the fused HIP path implements input-dependent B and C (B=None, C=None), "
                                  "the only form mamba_simple.Mamba passes (MS:208-209)")


class MambaInnerFnNoOutProj(torch.autograd.Function):
    """SSI:155-289.  Extra trailing argument `reverse` folds the xz.flip(-1)/out.flip(-1) sandwich of MS:229-246."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None, D=None,
                delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True, checkpoint_lvl=1,
                reverse=False):
        _check_variable_bc(B, C)
        return _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, None, None, A,
                              None, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, reverse)

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        g = _inner_backward(ctx, dout)
        return (g["dxz"], g["dconv_w"], g["dconv_b"], g["dx_proj_w"], g["ddt_proj_w"], g["dA"], None, None, g["dD"],
                g["ddelta_bias"], g["dB_proj_bias"], g["dC_proj_bias"], None, None, None)


class MambaInnerFn(torch.autograd.Function):
    """SSI:292-434."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True,
                checkpoint_lvl=1, reverse=False):
        _check_variable_bc(B, C)
        return _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                              out_proj_bias, A, None, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, reverse)

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        g = _inner_backward(ctx, dout)
        return (g["dxz"], g["dconv_w"], g["dconv_b"], g["dx_proj_w"], g["ddt_proj_w"], g["dout_proj_w"],
                g["dout_proj_b"], g["dA"], None, None, g["dD"], g["ddelta_bias"], g["dB_proj_bias"],
                g["dC_proj_bias"], None, None, None)


class BiMambaInnerFn(torch.autograd.Function):
    """SSI:437-603 (Fo-Bi / bimamba_type='v1'): shared conv/x_proj/dt_proj, forward scan with A plus time-reversed
    scan with A_b, summed, one out_proj."""

    @staticmethod
    @_custom_fwd
    def forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                delta_softplus=True, checkpoint_lvl=1, reverse=False):
        """`reverse` (extension): the block applied to the time-reversed sequence and reversed back -- flip(fn(flip(xz))) -- without the
        copies: the conv runs anti-causally and the two scan directions trade their A matrices (MM:623-638, the odd layers of an
        `if_bidirectional` model)."""
        _check_variable_bc(B, C)
        ctx.swapped = bool(reverse)
        if reverse:
            A, A_b = A_b, A
        return _inner_forward(ctx, xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                              out_proj_bias, A, A_b, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, bool(reverse))

    @staticmethod
    @_custom_bwd
    def backward(ctx, dout):
        g = _inner_backward(ctx, dout)
        dA, dA_b = (g["dA_b"], g["dA"]) if ctx.swapped else (g["dA"], g["dA_b"])
        return (g["dxz"], g["dconv_w"], g["dconv_b"], g["dx_proj_w"], g["ddt_proj_w"], g["dout_proj_w"],
                g["dout_proj_b"], dA, dA_b, None, None, g["dD"], g["ddelta_bias"], g["dB_proj_bias"],
                g["dC_proj_bias"], None, None, None)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                   A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                   delta_softplus=True, reverse=False):
    """`reverse=True` (extension) == flip(fn(flip(xz))) without the two copies."""
    return MambaInnerFn.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                              out_proj_bias, A, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, 1, reverse)


def bimamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                     A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                     delta_softplus=True, reverse=False):
    """`reverse=True` (extension) == flip(fn(flip(xz))) without the two copies."""
    return BiMambaInnerFn.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                                out_proj_bias, A, A_b, B, C, D, delta_bias, B_proj_bias, C_proj_bias, delta_softplus, 1, reverse)


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None,
                               D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True,
                               reverse=False):
    """`reverse=True` (extension) == flip(fn(flip(xz)))) without the two copies."""
    return MambaInnerFnNoOutProj.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D,
                                       delta_bias, B_proj_bias, C_proj_bias, delta_softplus, 1, reverse)


# =================================================================================================
# pure-PyTorch statements of the inner blocks (public names of the reference module: SSI:636-709)
# =================================================================================================
def _conv_ref(x, weight, bias):
    w = weight.reshape(weight.shape[0], -1)
    y = F.conv1d(x, w[:, None, :], bias, padding=w.shape[1] - 1, groups=x.shape[1])[..., :x.shape[-1]]
    return F.silu(y)


def _inner_pre_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, N, B_proj_bias, C_proj_bias):
    Bsz, two_e, L = xz.shape
    E = two_e // 2
    R = delta_proj_weight.shape[1]
    x, z = xz[:, :E], xz[:, E:]
    xc = _conv_ref(x, conv1d_weight, conv1d_bias)
    proj = xc.transpose(1, 2).reshape(Bsz * L, E) @ x_proj_weight.t()
    delta = (proj[:, :R] @ delta_proj_weight.t()).reshape(Bsz, L, E).transpose(1, 2)
    Bm, Cm = proj[:, R:R + N], proj[:, R + N:R + 2 * N]
    if B_proj_bias is not None:
        Bm = Bm + B_proj_bias.to(Bm.dtype)
    if C_proj_bias is not None:
        Cm = Cm + C_proj_bias.to(Cm.dtype)
    Bm = Bm.reshape(Bsz, L, N).transpose(1, 2).contiguous()
    Cm = Cm.reshape(Bsz, L, N).transpose(1, 2).contiguous()
    return xc, z, delta, Bm, Cm


def mamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                    A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                    delta_softplus=True):
    xc, z, delta, Bm, Cm = _inner_pre_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                          A.shape[-1], B_proj_bias, C_proj_bias)
    y = selective_scan_ref(xc, delta, A, Bm if B is None else B, Cm if C is None else C, D, z=z,
                           delta_bias=delta_bias, delta_softplus=True)
    return F.linear(y.transpose(1, 2), out_proj_weight, out_proj_bias)


def bimamba_inner_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight,
                      out_proj_bias, A, A_b, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None,
                      C_proj_bias=None, delta_softplus=True):
    xc, z, delta, Bm, Cm = _inner_pre_ref(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight,
                                          A.shape[-1], B_proj_bias, C_proj_bias)
    Bm = Bm if B is None else B
    Cm = Cm if C is None else C
    y = selective_scan_ref(xc, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias, delta_softplus=True)
    fl = lambda t: t.flip([-1])
    y_b = selective_scan_ref(fl(xc), fl(delta), A_b, fl(Bm), fl(Cm), D, fl(z), delta_bias, delta_softplus=True)
    return F.linear((y + fl(y_b)).transpose(1, 2), out_proj_weight, out_proj_bias)
