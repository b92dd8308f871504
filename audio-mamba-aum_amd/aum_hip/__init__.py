"""aum_hip -- ctypes binding of libaum_hip.so (the C ABI in include/aum_hip.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here (device memory, streams); every op below enqueues hand-written gfx950 kernels on the
current HIP stream through the C ABI.  There is NO fallback: if the shared library is missing the import of
the product path fails loudly (`python audio-mamba-aum_amd/csrc/build.py` builds it with hipcc).

The functions mirror the reference's extension modules:
  scan_fwd / scan_bwd          <- selective_scan_cuda.fwd / .bwd   (SSI:37, 62-65, 499-505, 541-552)
  conv1d_fwd / conv1d_bwd      <- causal_conv1d_cuda.causal_conv1d_fwd / _bwd (SSI:463, 594-596)
  rmsnorm_fwd / rmsnorm_bwd    <- _layer_norm_fwd / _layer_norm_bwd (LN:123-177, 293-377)
"""
import ctypes as C
import math
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libaum_hip.so")

AUM_F32, AUM_BF16, AUM_F16 = 0, 1, 2
SCAN_SOFTPLUS, SCAN_REVERSE, SCAN_GENERIC, SCAN_ROWPAIR, SCAN_ACCUMULATE = 1, 2, 4, 8, 16
CONV_SILU, CONV_REVERSE = 1, 2
_DT = {torch.float32: AUM_F32, torch.bfloat16: AUM_BF16, torch.float16: AUM_F16}
_ERR = {-1: "AUM_E_NULL", -2: "AUM_E_SHAPE", -3: "AUM_E_DTYPE", -4: "AUM_E_UNSUPPORTED", -5: "AUM_E_WORKSPACE",
        -6: "AUM_E_LAUNCH"}

_i64, _i32, _u32, _vp, _fp = C.c_int64, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p
ABI_VERSION = 13


class _Debug:
    """Kernel-selection switches for A/B runs and ablations (tools/kbench.py, tests).  They are attributes set by the tool that
    wants them; the environment is consulted ONCE, at import, and only when AUM_DEBUG=1 -- a stray variable in a training
    environment cannot change which kernel runs."""

    def __init__(self):
        self.ablate = 0            # AUM_DBG_* bits (upper half of the kernel `flags`)
        self.rowpair = False       # general row-pair kernels instead of the specialised ones
        self.no_ckpt = False       # long rows: no chunk-entry checkpoint
        self.no_accumulate = False  # long rows: second direction does not accumulate into the first one's tensors
        self.no_lane_ckpt = False  # L = 513 rows: scan_lane_ckpt() hands out no checkpoint
        self.proj_splits = 0       # token splits of the projection weight-gradient kernel (0: aum_proj_bwd_weight_splits)
        self.torch_sums = False    # partial results summed by torch.sum instead of aum_sum_rows
        self.sums_one_by_one = False   # sum_rows_multi: one aum_sum_rows launch per partial set (the launches of rounds 2-4; A/B)
        if os.environ.get("AUM_DEBUG") == "1":
            self.ablate = int(os.environ.get("AUM_ABLATE", "0"))
            self.rowpair = os.environ.get("AUM_SCAN_ROWPAIR") == "1"
            self.no_ckpt = os.environ.get("AUM_SCAN_NO_CKPT") == "1"
            self.no_accumulate = os.environ.get("AUM_SCAN_NO_ACCUMULATE") == "1"
            self.no_lane_ckpt = os.environ.get("AUM_SCAN_NO_LANE_CKPT") == "1"
            self.torch_sums = os.environ.get("AUM_TORCH_SUMS") == "1"
            self.sums_one_by_one = os.environ.get("AUM_SUMS_ONE_BY_ONE") == "1"


debug = _Debug()


class ScanFwdArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("u", "delta", "z", "B", "C", "A", "A_b", "D", "delta_bias", "out", "out_pre",
                                    "last_state", "workspace")]
                + [(n, _i64) for n in ("workspace_bytes", "u_bs", "u_ds", "delta_bs", "delta_ds", "z_bs", "z_ds", "B_bs",
                                       "B_ns", "C_bs", "C_ns", "out_bs", "out_ds")]
                + [(n, _i32) for n in ("batch", "dim", "len", "dstate", "dtype")] + [("flags", _u32), ("x_ck", _vp), ("x_lane", _vp)])


class ScanBwdArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("u", "delta", "z", "B", "C", "dout", "out_pre", "A", "A_b", "D", "delta_bias", "du",
                                    "ddelta", "dz", "dA", "dA_b", "dB", "dC", "dD", "ddelta_bias", "workspace")]
                + [(n, _i64) for n in ("workspace_bytes", "u_bs", "u_ds", "delta_bs", "delta_ds", "z_bs", "z_ds", "B_bs",
                                       "B_ns", "C_bs", "C_ns", "dout_bs", "dout_ds", "out_bs", "out_ds", "du_bs", "du_ds",
                                       "ddelta_bs", "ddelta_ds", "dz_bs", "dz_ds", "dB_bs", "dB_ns", "dC_bs", "dC_ns")]
                + [(n, _i32) for n in ("batch", "dim", "len", "dstate", "dtype")] + [("flags", _u32), ("x_ck", _vp), ("x_lane", _vp)])


class ScanTmFwdArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("u", "delta", "z", "B", "C", "A", "A_b", "D", "delta_bias", "out", "out_pre", "ckpt")]
                + [(n, _i64) for n in ("u_bs", "u_ts", "delta_bs", "delta_ts", "z_bs", "z_ts", "B_bs", "B_ts", "C_bs", "C_ts",
                                       "out_bs", "out_ts", "pre_bs", "pre_ts")]
                + [(n, _i32) for n in ("batch", "dim", "len", "dstate", "dtype")] + [("flags", _u32)])


class ScanTmBwdArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("u", "delta", "z", "B", "C", "dout", "out_pre", "A", "A_b", "D", "delta_bias", "ckpt", "du",
                                    "ddelta", "dz", "dA", "dA_b", "dBC", "dD", "ddelta_bias", "workspace")]
                + [(n, _i64) for n in ("workspace_bytes", "u_bs", "u_ts", "delta_bs", "delta_ts", "z_bs", "z_ts", "B_bs", "B_ts",
                                       "C_bs", "C_ts", "dout_bs", "dout_ts", "pre_bs", "pre_ts", "du_bs", "du_ts", "ddelta_bs",
                                       "ddelta_ts", "dz_bs", "dz_ts")]
                + [(n, _i32) for n in ("batch", "dim", "len", "dstate", "dtype")] + [("flags", _u32), ("dA_xA", _vp), ("dA_b_xA", _vp)])


class ScanTmSegFwdArgs(C.Structure):
    _fields_ = [("base", ScanTmFwdArgs), ("carry", _vp), ("carry_bytes", _i64), ("segments", _i32), ("reserved", _i32)]


class ScanTmSegBwdArgs(C.Structure):
    _fields_ = [("base", ScanTmBwdArgs), ("segments", _i32), ("reserved", _i32)]


class ConvArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("x", "dy", "weight", "bias", "y", "dx", "dweight", "dbias")]
                + [(n, _i64) for n in ("x_bs", "x_ds", "y_bs", "y_ds", "dy_bs", "dy_ds", "dx_bs", "dx_ds")]
                + [(n, _i32) for n in ("batch", "dim", "len", "width", "dtype")] + [("flags", _u32)])


class ConvTmArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("x", "dy", "weight", "bias", "y", "dx", "dw_part", "db_part")]
                + [(n, _i64) for n in ("x_bs", "x_ts", "y_bs", "y_ts", "dy_bs", "dy_ts", "dx_bs", "dx_ts")]
                + [(n, _i32) for n in ("batch", "dim", "len", "width", "dtype")] + [("flags", _u32)])


class NormArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("x", "residual", "dy", "dresidual_out", "weight", "rstd_in", "y", "residual_out",
                                    "dx", "dresidual_in", "rstd_out", "dweight_partial")]
                + [(n, _i64) for n in ("row_stride_x", "row_stride_res", "row_stride_y", "row_stride_res_out",
                                       "row_stride_dy", "row_stride_dres_out", "row_stride_dx", "row_stride_dres_in")]
                + [("eps", C.c_float)] + [(n, _i32) for n in ("rows", "cols", "x_dtype", "res_dtype", "y_dtype")]
                + [("flags", _u32)])


class FbankArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("wave", "window", "twiddle", "mel_start_f", "mel_count_f", "mel_w", "out")]
                + [(n, _i64) for n in ("wave_bs", "out_bs")]
                + [(n, _i32) for n in ("batch", "n_samples", "win", "shift", "padded", "num_frames", "target_length",
                                       "num_mel", "mel_wstride")]
                + [(n, C.c_float) for n in ("preemph", "norm_mean", "norm_inv2std", "log_floor")]
                + [("aug", _vp), ("noise", _vp)])


class FrontendArgs(C.Structure):
    _fields_ = ([("fbank", FbankArgs)] + [(n, _vp) for n in ("weight", "bias", "pos", "cls_row", "tokens", "patches")]
                + [("tokens_bs", _i64)] + [(n, _i32) for n in ("dim", "cls_pos", "dtype", "out_dtype")] + [("flags", _u32)])


class ProjArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("act", "w_x", "w_dt", "x_dbl", "out_act", "dB", "dC")]
                + [(n, _i64) for n in ("dB_bs", "dB_ns", "dC_bs", "dC_ns", "ntok")]
                + [(n, _i32) for n in ("dim", "dt_rank", "dstate", "len", "dtype", "w_ld")])


class ProjWArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("x", "y", "out")] + [("ntok", _i64)]
                + [(n, _i32) for n in ("dim", "nrows", "nsplit", "transpose_out", "dtype")])


class GemmArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("a", "b", "c")] + [(n, _i32) for n in ("m", "n", "k", "lda", "ldb", "ldc", "dtype")] + [("flags", _u32)])


class XdtBwdArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("ddelta", "dbc", "wdt_t", "wx_t", "du", "dx_dbl")] + [("ntok", _i64)]
                + [(n, _i32) for n in ("dim", "rank", "ncols", "ldd", "lddbc", "ldwdt", "ldwx", "ldu", "ldx", "dtype")])


class GemmWArgs(C.Structure):
    _fields_ = [("y", C.c_void_p), ("x", C.c_void_p), ("part", C.c_void_p), ("t", C.c_int64), ("ldy", C.c_int64), ("ldx", C.c_int64),
                ("n", C.c_int32), ("k", C.c_int32), ("splits", C.c_int32), ("dtype", C.c_int32)]


SUM_MAX_JOBS = 8        # AUM_SUM_MAX_JOBS


class SumJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("outer", C.c_int64), ("inner", C.c_int64), ("tr_cols", C.c_int32), ("reserved", C.c_int32)]


class ConvUpdateArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("conv_state", C.c_void_p), ("weight", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("batch", C.c_int32), ("dim", C.c_int32), ("width", C.c_int32), ("dtype", C.c_int32), ("flags", C.c_uint32)]


class StateUpdateArgs(C.Structure):
    _fields_ = [("state", C.c_void_p), ("x", C.c_void_p), ("dt", C.c_void_p), ("z", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("A", C.c_void_p), ("D", C.c_void_p), ("dt_bias", C.c_void_p), ("out", C.c_void_p),
                ("batch", C.c_int32), ("dim", C.c_int32), ("dstate", C.c_int32), ("dtype", C.c_int32), ("flags", C.c_uint32)]


class DtProjArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("x", "w", "out")] + [("ntok", _i64)] + [(n, _i32) for n in ("dim", "rank", "ldx", "ldw", "ldo", "dtype")])


class XdtArgs(C.Structure):
    _fields_ = ([(n, _vp) for n in ("u", "wx", "wdt", "x_dbl", "delta")] + [("ntok", _i64)]
                + [(n, _i32) for n in ("dim", "rank", "ncols", "ldu", "ldwx", "ldwdt", "ldx", "ldd", "dtype")])


EXPORTS = ["aum_gemm_tn", "aum_dtproj_tm_fwd", "aum_xdt_tm_fwd", "aum_proj_fwd", "aum_proj_bwd_data", "aum_proj_bwd_weight", "aum_proj_bwd_weight_splits", "aum_fbank_fwd", "aum_frontend_tokens_fwd", "aum_abi_version", "aum_selective_scan_fwd", "aum_selective_scan_bwd", "aum_scan_max_single_pass_len",
           "aum_selective_scan_workspace_bytes", "aum_selective_scan_ckpt_bytes", "aum_selective_scan_lane_ckpt_bytes", "aum_causal_conv1d_fwd", "aum_causal_conv1d_bwd", "aum_rmsnorm_fwd",
           "aum_rmsnorm_bwd", "aum_rmsnorm_bwd_partials", "aum_selftest_wave_scan", "aum_hbm_copy", "aum_sum_rows", "aum_sum_rows_multi",
           "aum_scan_tm_fwd", "aum_scan_tm_nck", "aum_scan_tm_ckpt_rows", "aum_scan_tm_bwd", "aum_scan_tm_workspace_bytes", "aum_scan_tm_seg_fwd", "aum_scan_tm_seg_bwd",
           "aum_scan_tm_seg_carry_bytes", "aum_scan_tm_seg_workspace_bytes", "aum_selftest_wave_sum32",
           "aum_conv1d_tm_fwd", "aum_conv1d_tm_bwd", "aum_conv1d_tm_nparts", "aum_gemm_wgrad", "aum_xdt_tm_bwd", "aum_causal_conv1d_update", "aum_selective_state_update", "aum_cast_bank", "aum_rmsnorm_bwd_partial_rows"]


class Lib:
    """A loaded C-ABI library.  `host=True` marks the tests-only lane-array build that takes host pointers."""

    def __init__(self, path, host=False):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: the HIP extension is not built.  Run `python audio-mamba-aum_amd/csrc/build.py` "
                "(hipcc, gfx950).  There is no CPU fallback on the product path.")
        self.path, self.host = path, host
        self.c = C.CDLL(path)
        for name in EXPORTS:
            getattr(self.c, name)   # AttributeError if a declared symbol is missing
        self.c.aum_selective_scan_workspace_bytes.restype = _i64
        self.c.aum_selective_scan_workspace_bytes.argtypes = [_i32] * 6
        for n in ("aum_selective_scan_fwd", "aum_selective_scan_bwd", "aum_causal_conv1d_fwd", "aum_causal_conv1d_bwd",
                  "aum_rmsnorm_fwd", "aum_rmsnorm_bwd"):
            getattr(self.c, n).argtypes = [_vp, _vp]
        self.c.aum_scan_tm_fwd.argtypes = [_vp, _vp]
        self.c.aum_scan_tm_bwd.argtypes = [_vp, _vp]
        self.c.aum_scan_tm_seg_fwd.argtypes = [_vp, _vp]
        self.c.aum_scan_tm_seg_bwd.argtypes = [_vp, _vp]
        self.c.aum_scan_tm_nck.argtypes = [_i32]
        self.c.aum_scan_tm_ckpt_rows.argtypes = [_i32]
        self.c.aum_conv1d_tm_fwd.argtypes = [_vp, _vp]
        self.c.aum_conv1d_tm_bwd.argtypes = [_vp, _vp]
        self.c.aum_conv1d_tm_nparts.argtypes = [_i32, _i32]
        self.c.aum_gemm_tn.argtypes = [_vp, _vp]
        self.c.aum_gemm_wgrad.argtypes = [_vp, _vp]
        self.c.aum_causal_conv1d_update.argtypes = [_vp, _vp]
        self.c.aum_selective_state_update.argtypes = [_vp, _vp]
        self.c.aum_dtproj_tm_fwd.argtypes = [_vp, _vp]
        self.c.aum_xdt_tm_fwd.argtypes = [_vp, _vp]
        self.c.aum_xdt_tm_bwd.argtypes = [_vp, _vp]
        self.c.aum_scan_tm_workspace_bytes.restype = _i64
        self.c.aum_scan_tm_workspace_bytes.argtypes = [_i32] * 5
        for fn in (self.c.aum_scan_tm_seg_carry_bytes, self.c.aum_scan_tm_seg_workspace_bytes):
            fn.restype = _i64
            fn.argtypes = [_i32] * 6
        self.c.aum_fbank_fwd.argtypes = [_vp, _vp]
        self.c.aum_frontend_tokens_fwd.argtypes = [_vp, _vp]
        for n in ("aum_proj_fwd", "aum_proj_bwd_data", "aum_proj_bwd_weight"):
            getattr(self.c, n).argtypes = [_vp, _vp]
        self.c.aum_proj_bwd_weight_splits.argtypes = [_i32, _i64]
        self.c.aum_selftest_wave_scan.argtypes = [_vp, _vp, C.c_int, _vp]
        self.c.aum_hbm_copy.argtypes = [_vp, _vp, _i64, _vp]
        self.c.aum_cast_bank.argtypes = [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp]
        self.c.aum_selftest_wave_sum32.argtypes = [_vp, _vp, _vp]
        self.c.aum_sum_rows.argtypes = [_vp, _vp, _i64, _i64, _i64, _i32, _vp]
        self.c.aum_sum_rows_multi.argtypes = [_vp, _i32, _vp]
        self.c.aum_selective_scan_ckpt_bytes.restype = _i64
        self.c.aum_selective_scan_ckpt_bytes.argtypes = [_i32] * 4
        self.c.aum_selective_scan_lane_ckpt_bytes.restype = _i64
        self.c.aum_selective_scan_lane_ckpt_bytes.argtypes = [_i32] * 5
        if self.c.aum_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {self.c.aum_abi_version()}, this binding speaks {ABI_VERSION} -- rebuild the library "
                               "(python audio-mamba-aum_amd/csrc/build.py)")
        self.max_single_pass_len = int(self.c.aum_scan_max_single_pass_len())

    def stream(self, t):
        if self.host:
            return None
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def check_tensor(self, t):
        if t is None:
            return
        if self.host:
            if t.device.type != "cpu":
                raise RuntimeError("lane-array (test) library takes host tensors")
        elif t.device.type != "cuda":
            raise RuntimeError("libaum_hip.so needs device (HIP) tensors; there is no CPU path in the product library")


_product = None


def get():
    """The product library (libaum_hip.so).  Raises ImportError if it has not been built."""
    global _product
    if _product is None:
        so = _SO
        if os.environ.get("AUM_DEBUG") == "1" and os.environ.get("AUM_HIP_LIB"):     # A/B builds (tools/ only)
            so = os.environ["AUM_HIP_LIB"]
        _product = Lib(so, host=False)
    return _product


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {_ERR.get(rc, rc)}")


class LaunchTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are enqueued on (torch's current
    stream).  bench.py enables it to compute the live roofline numbers; off by default (zero overhead)."""

    def __init__(self):
        self.enabled = False
        self.only = None        # optional set of kernel names to time (two events per launch cost ~8 us of host time)
        self.every = 1          # bracket every n-th launch of a name only: an event on the stream is a barrier packet, ~6 us of idle GPU each
                                # (profiles/r06_step_timeline.txt: two 5.8 us gaps around every bracketed launch)
        self.records = {}       # name -> list of (start_event, end_event, meta)
        self.seen = {}          # name -> launches seen since reset()

    def reset(self):
        self.records = {}
        self.seen = {}

    def want(self, name):
        n = self.seen.get(name, 0)
        self.seen[name] = n + 1
        return n % self.every == 0

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            out[name] = {"launches": len(ms), "avg_ms": sum(ms) / max(len(ms), 1), "meta": evs[-1][2] if evs else None}
        return out


timer = LaunchTimer()


def _launch(fn, args, stream_tensor, lib, name, meta=None):
    if timer.enabled and not lib.host and (timer.only is None or name in timer.only) and timer.want(name):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(C_byref(args), lib.stream(stream_tensor))
        b.record()
        timer.records.setdefault(name, []).append((a, b, meta))
    else:
        rc = fn(C_byref(args), lib.stream(stream_tensor))
    _chk(rc, name)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _unit(t, name):
    if t is not None and t.stride(-1) != 1 and t.shape[-1] != 1:
        raise RuntimeError(f"{name}: the time axis must be unit-stride (SSI:19-30)")


def _f32c(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _bc3(t):
    """(batch, 1, dstate, len) -> (batch, dstate, len) view (the G=1 case, SSI:31-36)."""
    if t.dim() == 4:
        if t.shape[1] != 1:
            raise RuntimeError("only one B/C group is supported (G=1), as used by Mamba (SSI:473-493)")
        t = t[:, 0]
    return t


def _alloc(batch, dim, length, dtype, device, dmajor):
    """(batch, dim, len) tensor; dmajor=True stores it channel-major [dim][batch][len] (the layout of the reference's
    xz view, MS:185-189) so the projections on either side of the kernels are transpose-free GEMMs."""
    if dmajor:
        return torch.empty((dim, batch, length), dtype=dtype, device=device).permute(1, 0, 2)
    return torch.empty((batch, dim, length), dtype=dtype, device=device)


def scan_ckpt(u, dstate, lib=None):
    """An empty `x` checkpoint tensor (batch, dim, len/512, dstate) fp32 for rows the chunked kernels take (long-form clips,
    L = 512 m + 1), else None.  Pass it as x_ck to scan_fwd (filled) and then to scan_bwd of the same direction (which then
    skips its pre-pass); not with generic=/rowpair=.  debug.no_ckpt (A/B runs) or debug.rowpair: no checkpoint."""
    lib = lib or get()
    batch, dim, length = u.shape
    if debug.rowpair or debug.no_ckpt:
        return None
    if lib.c.aum_selective_scan_ckpt_bytes(batch, dim, length, dstate) <= 0:
        return None
    return torch.empty((batch, dim, length // 512, dstate), dtype=torch.float32, device=u.device)


def scan_lane_ckpt(u, dstate, bidir, lib=None):
    """An empty lane-entry checkpoint (batch, dim, directions, dstate, 64) fp32 for rows the L = 513 row kernels take, else None.
    Pass it as x_lane to scan_fwd (filled) and to scan_bwd of the same call, which then skips recomputing the forward scan."""
    lib = lib or get()
    batch, dim, length = u.shape
    if debug.rowpair or debug.no_lane_ckpt:
        return None
    if lib.c.aum_selective_scan_lane_ckpt_bytes(batch, dim, length, dstate, int(bool(bidir))) <= 0:
        return None
    return torch.empty((batch, dim, 2 if bidir else 1, dstate, 64), dtype=torch.float32, device=u.device)


def scan_accumulates(u, dstate, lib=None):
    """True when a second-direction call on this shape can add to the first one's tensors (accumulate_into=): the rows the
    chunked kernels take, i.e. exactly when scan_ckpt() gives a checkpoint."""
    lib = lib or get()
    batch, dim, length = u.shape
    if debug.rowpair or debug.no_accumulate:
        return False
    return lib.c.aum_selective_scan_ckpt_bytes(batch, dim, length, dstate) > 0


def scan_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, reverse=False, A_b=None,
             want_out_pre=False, want_last_state=False, dmajor=False, generic=False, rowpair=False, x_ck=None,
             accumulate_into=None, x_lane=None, lib=None):
    """selective_scan_cuda.fwd.  Returns (out, out_pre|None, last_state|None).  A_b != None: fused bidirectional.
    x_ck: a scan_ckpt() tensor to fill.  accumulate_into: the `out` of the other direction's call (long rows only, see
    scan_accumulates): this call adds to it and returns it."""
    lib = lib or get()
    B, C = _bc3(B), _bc3(C)
    for n, t in (("u", u), ("delta", delta), ("z", z), ("B", B), ("C", C)):
        _unit(t, n)
        lib.check_tensor(t)
    if u.dtype not in _DT or any(t is not None and t.dtype != u.dtype for t in (delta, z, B, C)):
        raise RuntimeError("u, delta, z, B, C must share one dtype in {fp32, bf16, fp16}")
    batch, dim, length = u.shape
    dstate = A.shape[1]
    if A.shape != (dim, dstate) or B.shape != (batch, dstate, length) or C.shape != B.shape:
        raise RuntimeError("shape mismatch in selective scan arguments")
    A, A_b, D, delta_bias = _f32c(A), _f32c(A_b), _f32c(D), _f32c(delta_bias)
    out = accumulate_into if accumulate_into is not None else _alloc(batch, dim, length, u.dtype, u.device, dmajor)
    assert out.shape == u.shape and out.dtype == u.dtype and out.stride(2) == 1
    out_pre = _alloc(batch, dim, length, u.dtype, u.device, dmajor) if want_out_pre else None
    last = torch.empty((batch, dim, dstate), dtype=torch.float32, device=u.device) if want_last_state else None
    a = ScanFwdArgs()
    a.u, a.delta, a.z, a.B, a.C = _ptr(u), _ptr(delta), _ptr(z), _ptr(B), _ptr(C)
    a.A, a.A_b, a.D, a.delta_bias = _ptr(A), _ptr(A_b), _ptr(D), _ptr(delta_bias)
    a.out, a.out_pre, a.last_state = _ptr(out), _ptr(out_pre), _ptr(last)
    if x_ck is not None:
        assert x_ck.dtype == torch.float32 and x_ck.is_contiguous() and x_ck.shape == (batch, dim, length // 512, dstate)
        lib.check_tensor(x_ck)
        a.x_ck = _ptr(x_ck)
    if x_lane is not None:
        assert x_lane.dtype == torch.float32 and x_lane.is_contiguous() and x_lane.shape == (batch, dim, 2 if A_b is not None else 1, dstate, 64)
        lib.check_tensor(x_lane)
        a.x_lane = _ptr(x_lane)
    a.u_bs, a.u_ds = u.stride(0), u.stride(1)
    a.delta_bs, a.delta_ds = delta.stride(0), delta.stride(1)
    if z is not None:
        a.z_bs, a.z_ds = z.stride(0), z.stride(1)
    a.B_bs, a.B_ns = B.stride(0), B.stride(1)
    a.C_bs, a.C_ns = C.stride(0), C.stride(1)
    a.out_bs, a.out_ds = out.stride(0), out.stride(1)
    a.batch, a.dim, a.len, a.dstate, a.dtype = batch, dim, length, dstate, _DT[u.dtype]
    a.flags = ((SCAN_SOFTPLUS if delta_softplus else 0) | (SCAN_REVERSE if reverse else 0) | (SCAN_GENERIC if generic else 0)
               | (SCAN_ROWPAIR if rowpair or debug.rowpair else 0)
               | (SCAN_ACCUMULATE if accumulate_into is not None else 0)
               | (debug.ablate << 16))   # kernel-ablation bits, set by tools/kbench.py only
    _launch(lib.c.aum_selective_scan_fwd, a, u, lib, "scan_fwd_bidir" if A_b is not None else "scan_fwd",
            (batch, dim, length, dstate, u.element_size(), want_out_pre))
    return out, out_pre, last


def C_byref(s):
    return C.cast(C.byref(s), C.c_void_p)


def scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, out_pre, delta_softplus=False, reverse=False, A_b=None,
             dz_out=None, dmajor=False, generic=False, rowpair=False, x_ck=None, accumulate_into=None, x_lane=None, lib=None):
    """selective_scan_cuda.bwd.  Returns dict(du, ddelta, dA, dA_b, dB, dC, dD, dz, ddelta_bias); dB/dC fp32
    (batch, dstate, len).  dz_out: optional preallocated (possibly strided) tensor written in place (SSI:537-545).
    accumulate_into: the dict the other direction's call returned (long rows only, see scan_accumulates): this call adds its
    du, ddelta, dz, dB, dC, dD, ddelta_bias to those tensors and returns them, with its own dA."""
    lib = lib or get()
    B, C = _bc3(B), _bc3(C)
    for n, t in (("u", u), ("delta", delta), ("z", z), ("B", B), ("C", C), ("dout", dout), ("out_pre", out_pre)):
        _unit(t, n)
        lib.check_tensor(t)
    batch, dim, length = u.shape
    dstate = A.shape[1]
    dev = u.device
    A, A_b, D, delta_bias = _f32c(A), _f32c(A_b), _f32c(D), _f32c(delta_bias)
    acc = accumulate_into
    du = acc["du"] if acc else _alloc(batch, dim, length, u.dtype, dev, dmajor)
    ddelta = acc["ddelta"] if acc else _alloc(batch, dim, length, u.dtype, dev, dmajor)
    dz = None
    if z is not None:
        dz = acc["dz"] if acc else dz_out if dz_out is not None else _alloc(batch, dim, length, u.dtype, dev, dmajor)
        _unit(dz, "dz")
    f32 = dict(dtype=torch.float32, device=dev)
    # one zero-fill for all accumulate-into outputs (dA, dA_b, dD, ddelta_bias, dB, dC) instead of six
    nA, nBC = dim * dstate, batch * dstate * length
    sizes = [nA, nA if A_b is not None else 0, dim if D is not None and not acc else 0,
             dim if delta_bias is not None and not acc else 0, 0 if acc else nBC, 0 if acc else nBC]
    zbuf = torch.zeros((sum(sizes),), **f32)
    parts = torch.split(zbuf, sizes)
    dA = parts[0].view(dim, dstate)
    dA_b = parts[1].view(dim, dstate) if A_b is not None else None
    dD = parts[2] if D is not None else None
    dbias = parts[3] if delta_bias is not None else None
    dB = parts[4].view(batch, dstate, length) if not acc else None
    dC = parts[5].view(batch, dstate, length) if not acc else None
    if acc:      # the reduce stage of the call adds its partial sums to what is there
        assert A_b is None
        dB, dC, dD, dbias = acc["dB"], acc["dC"], acc["dD"], acc["ddelta_bias"]
    ws_bytes = int(lib.c.aum_selective_scan_workspace_bytes(batch, dim, length, dstate, int(A_b is not None), 1))
    ws = torch.empty((max(ws_bytes, 4) // 4,), **f32) if ws_bytes else None
    a = ScanBwdArgs()
    a.u, a.delta, a.z, a.B, a.C, a.dout, a.out_pre = map(_ptr, (u, delta, z, B, C, dout, out_pre))
    a.A, a.A_b, a.D, a.delta_bias = _ptr(A), _ptr(A_b), _ptr(D), _ptr(delta_bias)
    a.du, a.ddelta, a.dz = _ptr(du), _ptr(ddelta), _ptr(dz)
    a.dA, a.dA_b, a.dB, a.dC, a.dD, a.ddelta_bias = map(_ptr, (dA, dA_b, dB, dC, dD, dbias))
    a.workspace, a.workspace_bytes = _ptr(ws), ws_bytes
    if x_ck is not None:
        assert x_ck.dtype == torch.float32 and x_ck.is_contiguous() and x_ck.shape == (batch, dim, length // 512, dstate)
        a.x_ck = _ptr(x_ck)
    if x_lane is not None:
        assert x_lane.dtype == torch.float32 and x_lane.is_contiguous() and x_lane.shape == (batch, dim, 2 if A_b is not None else 1, dstate, 64)
        lib.check_tensor(x_lane)
        a.x_lane = _ptr(x_lane)
    a.u_bs, a.u_ds = u.stride(0), u.stride(1)
    a.delta_bs, a.delta_ds = delta.stride(0), delta.stride(1)
    if z is not None:
        a.z_bs, a.z_ds = z.stride(0), z.stride(1)
        a.out_bs, a.out_ds = out_pre.stride(0), out_pre.stride(1)
        a.dz_bs, a.dz_ds = dz.stride(0), dz.stride(1)
    a.B_bs, a.B_ns, a.C_bs, a.C_ns = B.stride(0), B.stride(1), C.stride(0), C.stride(1)
    a.dout_bs, a.dout_ds = dout.stride(0), dout.stride(1)
    a.du_bs, a.du_ds = du.stride(0), du.stride(1)
    a.ddelta_bs, a.ddelta_ds = ddelta.stride(0), ddelta.stride(1)
    a.dB_bs, a.dB_ns, a.dC_bs, a.dC_ns = dB.stride(0), dB.stride(1), dC.stride(0), dC.stride(1)
    a.batch, a.dim, a.len, a.dstate, a.dtype = batch, dim, length, dstate, _DT[u.dtype]
    a.flags = ((SCAN_SOFTPLUS if delta_softplus else 0) | (SCAN_REVERSE if reverse else 0) | (SCAN_GENERIC if generic else 0)
               | (SCAN_ROWPAIR if rowpair or debug.rowpair else 0)
               | (SCAN_ACCUMULATE if acc else 0)
               | (debug.ablate << 16))   # kernel-ablation bits, set by tools/kbench.py only
    _launch(lib.c.aum_selective_scan_bwd, a, u, lib, "scan_bwd_bidir" if A_b is not None else "scan_bwd",
            (batch, dim, length, dstate, u.element_size(), True))
    return dict(du=du, ddelta=ddelta, dA=dA, dA_b=dA_b, dB=dB, dC=dC, dD=dD, dz=dz, ddelta_bias=dbias)


# ---- time-serial scan on token-major activations (aum_scan_tm_*, ABI 7) -------------------------------------------------
SCAN_TM_CK = 8


def scan_tm_supported(dim, dstate):
    """the limits of aum_scan_tm_fwd / _bwd (include/aum_hip.h); outside them callers use scan_fwd / scan_bwd"""
    return dstate == 16 and dim % 64 == 0


def _tm3(t, name, last):
    """(batch, len, X) tensor with unit stride along X -> (batch stride, token stride) in elements"""
    if t.dim() != 3 or t.shape[2] != last or (t.stride(2) != 1 and last != 1):
        raise RuntimeError(f"{name}: expected (batch, len, {last}) with the last axis contiguous (token-major)")
    ts = t.stride(1) if t.shape[1] > 1 else last                # the stride of a size-1 axis is arbitrary: normalise
    bs = t.stride(0) if t.shape[0] > 1 else ts * t.shape[1]
    return bs, ts


def scan_tm_ckpt(batch, length, dim, dstate, bidir, device, lib=None, dtype=torch.float32):
    """empty state checkpoint (directions, batch, nck, rows, dim) dwords for scan_tm_fwd to fill and scan_tm_bwd to read; dtype = the
    activations' dtype: fp32 states (rows = dstate) for fp32, pairs of states in the activations' own type (rows = dstate / 2) for 16-bit activations"""
    lib = lib or get()
    nck = int(lib.c.aum_scan_tm_nck(length))
    rows = int(lib.c.aum_scan_tm_ckpt_rows(_DT[dtype])) if dstate == 16 else dstate
    return torch.empty((2 if bidir else 1, batch, max(nck, 1), rows, dim), dtype=torch.float32, device=device)


SCAN_TM_MAX_SEGMENTS = 32


def scan_tm_segments(batch, dim, length, bidir, training=False, nsimd=None, device=None):
    """time segments a token-major launch of this shape is cut into (1: not cut).  One wave per (batch entry, 64 channels, direction)
    leaves most SIMDs idle at a small batch while every wave walks the whole row; segments multiply the waves at the price of one
    carry pass (the recurrence once more, without outputs).  Rows are cut only when they are long (>= 1024 steps) and the uncut launch is
    under one wave per two SIMDs, into as many ranges as give every SIMD three waves per launch (each direction of a Fo-Bi pair is a
    launch of its own), ranges no shorter than 128 steps.  Measured at B = 8, L = 4097, E = 1536 (profiles/r04_seg_time.json): 16
    ranges are the fastest cut for the forward (three resident waves per SIMD: one round) AND for the backward (two resident: 11 or
    12 ranges leave a tail round, 16 is 1.5 rounds of shorter waves) -- 0.68 / 1.39 ms against 1.29 / 5.03 ms uncut."""
    nsimd = 4 * cu_count(device) if nsimd is None else nsimd          # MI355X: 256 CUs x 4 SIMDs; `device`: the operands' (default: current)
    per_dir = batch * (dim // 64)
    waves = per_dir * (2 if bidir else 1)
    if per_dir <= 0 or waves * 2 > nsimd or length < 1024:
        return 1
    want = -(-nsimd * 3 // per_dir)
    seg = min(want, length // 128, SCAN_TM_MAX_SEGMENTS)
    return seg if seg >= 2 else 1


def scan_tm_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, reverse=False, A_b=None,
                want_out_pre=False, ckpt=None, out=None, lib=None, segments=1):
    """Selective scan forward on token-major tensors: u, delta, z (batch, len, dim) with channels contiguous (row strides free: z
    may be a slice of an xz tensor); B, C (batch, len, dstate) in u's dtype.  A_b != None: both directions (Fo-Bi).
    segments > 1: the rows are cut into that many time ranges that run as waves of their own (aum_scan_tm_seg_fwd).
    Returns (out, out_pre|None), both (batch, len, dim) contiguous."""
    lib = lib or get()
    batch, length, dim = u.shape
    dstate = A.shape[1]
    for t in (u, delta, z, B, C):
        lib.check_tensor(t)
    if u.dtype not in _DT or delta.dtype != u.dtype or (z is not None and z.dtype != u.dtype):
        raise RuntimeError("u, delta, z must share one dtype in {fp32, bf16, fp16}")
    if B.dtype != u.dtype or C.dtype != u.dtype:
        raise RuntimeError("B, C must have u's dtype")
    A, A_b, D, delta_bias = _f32c(A), _f32c(A_b), _f32c(D), _f32c(delta_bias)
    a = ScanTmFwdArgs()
    a.u_bs, a.u_ts = _tm3(u, "u", dim)
    a.delta_bs, a.delta_ts = _tm3(delta, "delta", dim)
    if z is not None:
        a.z_bs, a.z_ts = _tm3(z, "z", dim)
    a.B_bs, a.B_ts = _tm3(B, "B", dstate)
    a.C_bs, a.C_ts = _tm3(C, "C", dstate)
    if out is None:
        out = torch.empty((batch, length, dim), dtype=u.dtype, device=u.device)
    out_pre = torch.empty((batch, length, dim), dtype=u.dtype, device=u.device) if want_out_pre else None
    a.out_bs, a.out_ts = _tm3(out, "out", dim)
    if out_pre is not None:
        a.pre_bs, a.pre_ts = _tm3(out_pre, "out_pre", dim)
    a.u, a.delta, a.z, a.B, a.C = _ptr(u), _ptr(delta), _ptr(z), _ptr(B), _ptr(C)
    a.A, a.A_b, a.D, a.delta_bias = _ptr(A), _ptr(A_b), _ptr(D), _ptr(delta_bias)
    a.out, a.out_pre = _ptr(out), _ptr(out_pre)
    if ckpt is not None:
        assert ckpt.dtype == torch.float32 and ckpt.is_contiguous()
        rows = int(lib.c.aum_scan_tm_ckpt_rows(_DT[u.dtype])) if dstate == 16 else dstate
        assert ckpt.numel() >= (2 if A_b is not None else 1) * batch * max(int(lib.c.aum_scan_tm_nck(length)), 1) * rows * dim, "ckpt too small"
        lib.check_tensor(ckpt)
        a.ckpt = _ptr(ckpt)
    a.batch, a.dim, a.len, a.dstate, a.dtype = batch, dim, length, dstate, _DT[u.dtype]
    a.flags = (SCAN_SOFTPLUS if delta_softplus else 0) | (SCAN_REVERSE if reverse else 0)
    if segments > 1:
        sa = ScanTmSegFwdArgs()
        sa.base = a
        sa.segments = int(segments)
        sa.carry_bytes = int(lib.c.aum_scan_tm_seg_carry_bytes(batch, dim, length, dstate, int(A_b is not None), int(segments)))
        if sa.carry_bytes <= 0:
            raise RuntimeError(f"scan_tm_fwd: segments={segments} not supported for this shape")
        carry = torch.empty((sa.carry_bytes // 4,), dtype=torch.float32, device=u.device)
        sa.carry = _ptr(carry)
        _launch(lib.c.aum_scan_tm_seg_fwd, sa, u, lib, "scan_tm_seg_fwd_bidir" if A_b is not None else "scan_tm_seg_fwd",
                (batch, dim, length, dstate, u.element_size(), want_out_pre, int(segments)))
        return out, out_pre
    _launch(lib.c.aum_scan_tm_fwd, a, u, lib, "scan_tm_fwd_bidir" if A_b is not None else "scan_tm_fwd",
            (batch, dim, length, dstate, u.element_size(), want_out_pre))
    return out, out_pre


def scan_tm_bwd(u, delta, A, B, C, D, z, delta_bias, dout, out_pre, ckpt, delta_softplus=False, reverse=False, A_b=None, dz_out=None,
                lib=None, segments=1, want_dA_xA=False, param_out=None):
    """Backward of scan_tm_fwd (same tensor conventions; ckpt: the tensor scan_tm_fwd filled).  Returns dict(du, ddelta, dz (batch, len,
    dim) in u's dtype, dBC (batch, len, 2 * dstate) fp32 = dB | dC, dA, dA_b (dim, dstate), dD, ddelta_bias (dim) fp32)."""
    lib = lib or get()
    batch, length, dim = u.shape
    dstate = A.shape[1]
    dev = u.device
    for t in (u, delta, z, B, C, dout, out_pre, ckpt):
        lib.check_tensor(t)
    # the kernel is told ONE element type: a tensor of another one would be read past its end (e.g. a 16-bit dout under fp32 rows)
    if u.dtype not in _DT:
        raise RuntimeError("u must be fp32, bf16 or fp16")
    for name, t in (("delta", delta), ("z", z), ("B", B), ("C", C), ("dout", dout), ("out_pre", out_pre), ("dz_out", dz_out)):
        if t is not None and t.dtype != u.dtype:
            raise RuntimeError(f"{name} must have u's dtype ({u.dtype}), got {t.dtype}")
    for name, t in (("delta", delta), ("z", z), ("dout", dout), ("out_pre", out_pre), ("dz_out", dz_out)):
        if t is not None and tuple(t.shape) != (batch, length, dim):
            raise RuntimeError(f"{name}: expected shape {(batch, length, dim)}, got {tuple(t.shape)}")
    if z is not None and out_pre is None:
        raise RuntimeError("out_pre (the forward's pre-gate sum) is required when z is given")
    A, A_b, D, delta_bias = _f32c(A), _f32c(A_b), _f32c(D), _f32c(delta_bias)
    bidir = A_b is not None
    if ckpt is None or ckpt.dtype != torch.float32 or not ckpt.is_contiguous():
        raise RuntimeError("ckpt: the contiguous fp32 tensor scan_tm_fwd filled")
    rows_ = int(lib.c.aum_scan_tm_ckpt_rows(_DT[u.dtype])) if dstate == 16 else dstate
    if ckpt.numel() < (2 if bidir else 1) * batch * max(int(lib.c.aum_scan_tm_nck(length)), 1) * rows_ * dim:
        raise RuntimeError("ckpt too small for this launch")
    du = torch.empty((batch, length, dim), dtype=u.dtype, device=dev)
    ddelta = torch.empty((batch, length, dim), dtype=u.dtype, device=dev)
    dz = None
    if z is not None:
        dz = dz_out if dz_out is not None else torch.empty((batch, length, dim), dtype=u.dtype, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    dBC = torch.empty((batch, length, 2 * dstate), **f32)
    # param_out: {"dD" | "ddelta_bias" | "dA_xA" | "dA_b_xA": destination} -- where a parameter's gradient is wanted (its bucket view under
    # DistributedDataParallel); anything missing or unfit is a new tensor
    po = param_out or {}
    dA = torch.empty((dim, dstate), **f32)
    dA_b = torch.empty((dim, dstate), **f32) if bidir else None
    dD = _sum_out(po.get("dD"), (dim,), dev) if D is not None else None
    dbias = _sum_out(po.get("ddelta_bias"), (dim,), dev) if delta_bias is not None else None
    # want_dA_xA: dA .* A (and dA_b .* A_b) from the same partial-sum launch -- the gradient of A_log where A = -exp(A_log)
    dA_xA = _sum_out(po.get("dA_xA"), (dim, dstate), dev) if want_dA_xA else None
    dA_b_xA = _sum_out(po.get("dA_b_xA"), (dim, dstate), dev) if want_dA_xA and bidir else None
    if segments > 1:
        ws_bytes = int(lib.c.aum_scan_tm_seg_workspace_bytes(batch, dim, length, dstate, int(bidir), int(segments)))
        if ws_bytes <= 0:
            raise RuntimeError(f"scan_tm_bwd: segments={segments} not supported for this shape")
    else:
        ws_bytes = int(lib.c.aum_scan_tm_workspace_bytes(batch, dim, length, dstate, int(bidir)))
    ws = torch.empty((max(ws_bytes, 4) // 4,), **f32)
    a = ScanTmBwdArgs()
    a.u, a.delta, a.z, a.B, a.C, a.dout, a.out_pre = map(_ptr, (u, delta, z, B, C, dout, out_pre))
    a.A, a.A_b, a.D, a.delta_bias, a.ckpt = _ptr(A), _ptr(A_b), _ptr(D), _ptr(delta_bias), _ptr(ckpt)
    a.du, a.ddelta, a.dz = _ptr(du), _ptr(ddelta), _ptr(dz)
    a.dA, a.dA_b, a.dBC, a.dD, a.ddelta_bias = map(_ptr, (dA, dA_b, dBC, dD, dbias))
    a.dA_xA, a.dA_b_xA = _ptr(dA_xA), _ptr(dA_b_xA)
    a.workspace, a.workspace_bytes = _ptr(ws), ws_bytes
    a.u_bs, a.u_ts = _tm3(u, "u", dim)
    a.delta_bs, a.delta_ts = _tm3(delta, "delta", dim)
    a.B_bs, a.B_ts = _tm3(B, "B", dstate)
    a.C_bs, a.C_ts = _tm3(C, "C", dstate)
    a.dout_bs, a.dout_ts = _tm3(dout, "dout", dim)
    a.du_bs, a.du_ts = _tm3(du, "du", dim)
    a.ddelta_bs, a.ddelta_ts = _tm3(ddelta, "ddelta", dim)
    if z is not None:
        a.z_bs, a.z_ts = _tm3(z, "z", dim)
        a.pre_bs, a.pre_ts = _tm3(out_pre, "out_pre", dim)
        a.dz_bs, a.dz_ts = _tm3(dz, "dz", dim)
    a.batch, a.dim, a.len, a.dstate, a.dtype = batch, dim, length, dstate, _DT[u.dtype]
    a.flags = (SCAN_SOFTPLUS if delta_softplus else 0) | (SCAN_REVERSE if reverse else 0)
    if segments > 1:
        sa = ScanTmSegBwdArgs()
        sa.base = a
        sa.segments = int(segments)
        _launch(lib.c.aum_scan_tm_seg_bwd, sa, u, lib, "scan_tm_seg_bwd_bidir" if bidir else "scan_tm_seg_bwd",
                (batch, dim, length, dstate, u.element_size(), True, int(segments)))
    else:
        _launch(lib.c.aum_scan_tm_bwd, a, u, lib, "scan_tm_bwd_bidir" if bidir else "scan_tm_bwd", (batch, dim, length, dstate, u.element_size(), True))
    return dict(du=du, ddelta=ddelta, dz=dz, dBC=dBC, dA=dA, dA_b=dA_b, dD=dD, ddelta_bias=dbias, dA_xA=dA_xA, dA_b_xA=dA_b_xA, _ws=ws)


GEMM_BN, GEMM_BK = 256, 64


_dev_cu = {}


def cu_count(device=None):
    """compute units of the device the kernels run on (MI355X: 256; a partitioned or different part reports its own) -- the host-side
    dispatch rules (time segments, token splits) are stated in CUs / SIMDs and must agree with what the library sees (csrc: cu_count())"""
    if not torch.cuda.is_available():
        return 256
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    n = _dev_cu.get(idx)
    if n is None:
        n = _dev_cu[idx] = int(torch.cuda.get_device_properties(idx).multi_processor_count)
    return n


def gemm_wgrad_splits(n, k, ncu=None, device=None):
    """token splits that give every CU one workgroup: (n / 256) (k / 256) output tiles x splits ~ the CU count"""
    ncu = cu_count(device) if ncu is None else ncu
    tiles = (n // 256) * max(1, k // 256)          # a skinny operand (k = 48 / 80) is one column tile
    return max(1, min(64, ncu // max(tiles, 1)))


def gemm_wgrad_supported(y, x, splits=None):
    """shapes aum_gemm_wgrad takes (include/aum_hip.h, ABI 10; the C side's gemm_wgrad_check, rule for rule): 16-bit 2-D token-major
    operands (rows = tokens, unit column stride), y's width a multiple of 256, x's a multiple of 256 or one of the skinny widths 48 / 80,
    at most 64 token splits, and a split's rows within 32-bit byte offsets of both operands"""
    if not (y.dim() == 2 and x.dim() == 2 and y.dtype == x.dtype and y.dtype in (torch.bfloat16, torch.float16) and y.shape[0] == x.shape[0]
            and y.stride(1) == 1 and x.stride(1) == 1 and y.shape[1] % 256 == 0 and (x.shape[1] % 256 == 0 or x.shape[1] in (48, 80))
            and y.stride(0) % 8 == 0 and y.stride(0) >= y.shape[1] and x.stride(0) >= x.shape[1]
            and x.stride(0) % 8 == 0 and y.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and y.shape[0] > 0):
        return False
    splits = splits or gemm_wgrad_splits(y.shape[1], x.shape[1], device=y.device if y.is_cuda else None)
    if not 0 < splits <= 64:
        return False
    chunk = (-(-y.shape[0] // splits) + 63) // 64 * 64
    return chunk * y.stride(0) * 2 < (1 << 31) and chunk * x.stride(0) * 2 < (1 << 31)


def gemm_wgrad(y, x, splits=None, lib=None, partials=False, out=None):
    """dW (n, k) fp32 = y (t, n)^T @ x (t, k): the weight gradient of a projection from token-major operands (aum_gemm_wgrad: hand-written
    MFMA kernel with transposing LDS reads, fp32 partial tiles over `splits` token ranges summed in a fixed order by aum_sum_rows)."""
    lib = lib or get()
    t, n = y.shape
    k = x.shape[1]
    splits = splits or gemm_wgrad_splits(n, k, device=y.device if y.is_cuda else None)
    if not gemm_wgrad_supported(y, x, splits):
        raise RuntimeError("gemm_wgrad: operands outside the kernel's limits (see gemm_wgrad_supported)")
    lib.check_tensor(y)
    lib.check_tensor(x)
    part = torch.empty((splits, n, k), dtype=torch.float32, device=y.device)
    a = GemmWArgs()
    a.y, a.x, a.part = _ptr(y), _ptr(x), _ptr(part)
    a.t, a.ldy, a.ldx, a.n, a.k, a.splits, a.dtype = t, y.stride(0), x.stride(0), n, k, splits, _DT[y.dtype]
    _launch(lib.c.aum_gemm_wgrad, a, y, lib, "gemm_wgrad", (t, n, k, y.element_size()))
    if partials:
        return part
    if splits == 1:
        return part[0] if out is None else sum_rows(part, lib=lib, out=out)
    return sum_rows(part, lib=lib, out=out)


def gemm_tn_supported(a, b):
    """shapes aum_gemm_tn takes (include/aum_hip.h, ABI 9): 16-bit 2-D operands with contiguous K, n % 256 == 0, k % 64 == 0"""
    if a.dim() != 2 or b.dim() != 2 or a.dtype != b.dtype or a.dtype not in (torch.bfloat16, torch.float16):
        return False
    m, k = a.shape
    n = b.shape[0]
    return (b.shape[1] == k and m > 0 and n % GEMM_BN == 0 and k % GEMM_BK == 0 and a.stride(1) == 1 and b.stride(1) == 1
            and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and 512 * a.stride(0) < 2 ** 31 and 512 * b.stride(0) < 2 ** 31)


GEMM_LOCKSTEP, GEMM_PIPELINED, GEMM_PACED = 1, 32, 256


def gemm_tn(a, b, out=None, lib=None, flags=0):
    """out[m][n] = sum_k a[m][k] b[n][k]: the in_proj / out_proj GEMM and their data gradients on token-major activations (MS:185-189,
    SSI:517, 540).  a (m, k), b (n, k): 16-bit, K contiguous; out (m, n) rows contiguous (may be a column block of a wider tensor).
    flags: 0 (the library picks: the paced-store kernel from k = 448 on) or one of GEMM_LOCKSTEP / GEMM_PIPELINED / GEMM_PACED."""
    lib = lib or get()
    lib.check_tensor(a)
    lib.check_tensor(b)
    if not gemm_tn_supported(a, b):
        raise RuntimeError(f"gemm_tn: unsupported operands {tuple(a.shape)} {a.dtype} x {tuple(b.shape)} {b.dtype} (need 16-bit, contiguous K, "
                           "n % 256 == 0, k % 64 == 0, 16-byte aligned rows)")
    m, k = a.shape
    n = b.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=a.dtype, device=a.device)
    elif (out.shape != (m, n) or out.dtype != a.dtype or out.stride(1) != 1 or out.stride(0) % 8 or out.data_ptr() % 16
          or out.device != a.device):
        raise RuntimeError("gemm_tn: out must be an (m, n) tensor of the operands' dtype with contiguous, 16-byte aligned rows")
    g = GemmArgs()
    g.a, g.b, g.c = _ptr(a), _ptr(b), _ptr(out)
    g.m, g.n, g.k, g.lda, g.ldb, g.ldc, g.dtype = m, n, k, a.stride(0), b.stride(0), out.stride(0), _DT[a.dtype]
    g.flags = flags
    _launch(lib.c.aum_gemm_tn, g, a, lib, "gemm_tn", (m, n, k))
    return out


def conv1d_update(x, conv_state, weight, bias=None, silu=True, lib=None):
    """one token of the causal conv (aum_causal_conv1d_update): x (batch, dim); conv_state (batch, dim, width) fp32 contiguous, shifted and
    extended IN PLACE; weight (dim, width); returns act(<window, weight> + bias) (batch, dim) in x's dtype"""
    lib = lib or get()
    if x.dim() != 2 or conv_state.dim() != 3 or conv_state.shape[:2] != x.shape or conv_state.dtype != torch.float32 or not conv_state.is_contiguous():
        raise RuntimeError("conv1d_update: x (batch, dim), conv_state (batch, dim, width) fp32 contiguous")
    for t in (x, conv_state):
        lib.check_tensor(t)
    x = x.contiguous()
    weight, bias = _f32c(weight.reshape(x.shape[1], -1)), _f32c(bias)
    out = torch.empty_like(x)
    a = ConvUpdateArgs()
    a.x, a.conv_state, a.weight, a.bias, a.out = _ptr(x), _ptr(conv_state), _ptr(weight), _ptr(bias), _ptr(out)
    a.batch, a.dim, a.width, a.dtype, a.flags = x.shape[0], x.shape[1], conv_state.shape[2], _DT[x.dtype], (CONV_SILU if silu else 0)
    _launch(lib.c.aum_causal_conv1d_update, a, x, lib, "conv1d_update")
    return out


def state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False, lib=None):
    """one token of the selective scan (aum_selective_state_update): state (batch, dim, dstate) fp32 contiguous, advanced IN PLACE;
    x, dt, z (batch, dim), B, C (batch, dstate) in one dtype; A (dim, dstate), D, dt_bias (dim); returns (batch, dim) in x's dtype"""
    lib = lib or get()
    if state.dim() != 3 or state.dtype != torch.float32 or not state.is_contiguous() or x.shape != state.shape[:2]:
        raise RuntimeError("state_update: state (batch, dim, dstate) fp32 contiguous, x (batch, dim)")
    for t in (state, x, dt, z, B, C):
        lib.check_tensor(t)
    dty = x.dtype
    x, dt, B, C = x.contiguous(), dt.to(dty).contiguous(), B.to(dty).contiguous(), C.to(dty).contiguous()
    z = None if z is None else z.to(dty).contiguous()
    if dt.shape != x.shape or B.shape != (x.shape[0], state.shape[2]) or C.shape != B.shape or (z is not None and z.shape != x.shape):
        raise RuntimeError("state_update: dt, z (batch, dim); B, C (batch, dstate)")
    A, D, dt_bias = _f32c(A), _f32c(D), _f32c(dt_bias)
    out = torch.empty_like(x)
    a = StateUpdateArgs()
    a.state, a.x, a.dt, a.z, a.B, a.C = _ptr(state), _ptr(x), _ptr(dt), _ptr(z), _ptr(B), _ptr(C)
    a.A, a.D, a.dt_bias, a.out = _ptr(A), _ptr(D), _ptr(dt_bias), _ptr(out)
    a.batch, a.dim, a.dstate, a.dtype, a.flags = x.shape[0], x.shape[1], state.shape[2], _DT[dty], (SCAN_SOFTPLUS if dt_softplus else 0)
    _launch(lib.c.aum_selective_state_update, a, x, lib, "state_update")
    return out


def dtproj_tm_supported(x_dbl, rank, w):
    """shapes aum_dtproj_tm_fwd takes (include/aum_hip.h, ABI 9): 16-bit row-major x_dbl (ntok, >= rank) and dt_proj.weight (dim, rank)"""
    return (x_dbl.dim() == 2 and w.dim() == 2 and x_dbl.dtype == w.dtype and x_dbl.dtype in (torch.bfloat16, torch.float16)
            and w.shape[1] == rank and x_dbl.shape[1] >= rank and rank % 8 == 0 and rank <= 64 and w.shape[0] % 32 == 0
            and x_dbl.stride(1) == 1 and w.stride(1) == 1 and x_dbl.stride(0) % 8 == 0 and w.stride(0) % 8 == 0
            and x_dbl.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def dtproj_tm_fwd(x_dbl, rank, w, lib=None):
    """delta (ntok, dim) = x_dbl[:, :rank] @ w^T (SSI:468 on token-major rows): w = dt_proj.weight (dim, rank) in x_dbl's 16-bit dtype"""
    lib = lib or get()
    lib.check_tensor(x_dbl)
    lib.check_tensor(w)
    if not dtproj_tm_supported(x_dbl, rank, w):
        raise RuntimeError(f"dtproj_tm_fwd: unsupported operands {tuple(x_dbl.shape)} {x_dbl.dtype}, rank {rank}, weight {tuple(w.shape)} {w.dtype}")
    ntok, dim = x_dbl.shape[0], w.shape[0]
    out = torch.empty((ntok, dim), dtype=x_dbl.dtype, device=x_dbl.device)
    a = DtProjArgs()
    a.x, a.w, a.out = _ptr(x_dbl), _ptr(w), _ptr(out)
    a.ntok, a.dim, a.rank, a.ldx, a.ldw, a.ldo, a.dtype = ntok, dim, rank, x_dbl.stride(0), w.stride(0), dim, _DT[x_dbl.dtype]
    _launch(lib.c.aum_dtproj_tm_fwd, a, x_dbl, lib, "dtproj_tm_fwd", (ntok, dim, rank))
    return out


XDT_COLS, XDT_MAX_DIM = 80, 1536
XDT_COLS_FWD = (80, 56)        # x_dbl widths aum_xdt_tm_fwd is built for (AuM-Base, AuM-Small)


def xdt_tm_supported(u, wx, wdt):
    """shapes aum_xdt_tm_fwd takes (include/aum_hip.h, ABI 9): 16-bit row-major conv_out (ntok, dim), x_proj.weight (80, dim),
    dt_proj.weight (dim, rank)"""
    if not (u.dim() == 2 and wx.dim() == 2 and wdt.dim() == 2 and u.dtype == wx.dtype == wdt.dtype and u.dtype in (torch.bfloat16, torch.float16)):
        return False
    dim, rank = u.shape[1], wdt.shape[1]
    ok_t = lambda t: t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
    return (wx.shape[0] in XDT_COLS_FWD and wx.shape[1] == dim and wdt.shape[0] == dim and dim % 256 == 0 and dim <= XDT_MAX_DIM and rank % 8 == 0
            and rank <= 64 and rank <= wx.shape[0]
            and ok_t(u) and ok_t(wx) and ok_t(wdt))


def xdt_tm_bwd_supported(ddelta, dbc, wdt_t, wx_t, du):
    """shapes aum_xdt_tm_bwd takes (include/aum_hip.h, ABI 10): 16-bit row-major ddelta / du (ntok, dim), fp32 dB | dC rows (ntok, 32),
    dt_proj.weight^T (48, dim), x_proj.weight^T (dim, 80)"""
    if not (ddelta.dim() == 2 and du.shape == ddelta.shape and ddelta.dtype == du.dtype == wdt_t.dtype == wx_t.dtype
            and ddelta.dtype in (torch.bfloat16, torch.float16) and dbc.dtype == torch.float32 and dbc.dim() == 2):
        return False
    ntok, dim = ddelta.shape
    ok_t = lambda t, m=8: t.stride(1) == 1 and t.stride(0) % m == 0 and t.data_ptr() % 16 == 0
    return (wx_t.shape == (dim, XDT_COLS) and wdt_t.shape == (XDT_COLS - 32, dim) and dbc.shape == (ntok, 32) and dim % 256 == 0
            and dim <= XDT_MAX_DIM and ok_t(ddelta) and ok_t(du) and ok_t(wdt_t) and ok_t(wx_t) and ok_t(dbc, 4))


def xdt_tm_bwd(ddelta, dbc, wdt_t, wx_t, du, lib=None):
    """dx_dbl (ntok, 80) = [ddelta @ wdt_t^T | dbc] in ddelta's dtype, and du += dx_dbl @ wx_t^T IN PLACE (SSI:570-574, 587, 590 on
    token-major rows: one pass over ddelta and du).  Returns dx_dbl."""
    lib = lib or get()
    for t in (ddelta, dbc, wdt_t, wx_t, du):
        lib.check_tensor(t)
    if not xdt_tm_bwd_supported(ddelta, dbc, wdt_t, wx_t, du):
        raise RuntimeError(f"xdt_tm_bwd: unsupported operands {tuple(ddelta.shape)} {ddelta.dtype}, {tuple(dbc.shape)}, {tuple(wdt_t.shape)}, {tuple(wx_t.shape)}")
    ntok, dim = ddelta.shape
    dx_dbl = torch.empty((ntok, XDT_COLS), dtype=ddelta.dtype, device=ddelta.device)
    a = XdtBwdArgs()
    a.ddelta, a.dbc, a.wdt_t, a.wx_t, a.du, a.dx_dbl = _ptr(ddelta), _ptr(dbc), _ptr(wdt_t), _ptr(wx_t), _ptr(du), _ptr(dx_dbl)
    a.ntok, a.dim, a.rank, a.ncols = ntok, dim, XDT_COLS - 32, XDT_COLS
    a.ldd, a.lddbc, a.ldwdt, a.ldwx, a.ldu, a.ldx = ddelta.stride(0), dbc.stride(0), wdt_t.stride(0), wx_t.stride(0), du.stride(0), XDT_COLS
    a.dtype = _DT[ddelta.dtype]
    _launch(lib.c.aum_xdt_tm_bwd, a, ddelta, lib, "xdt_tm_bwd", (ntok, dim))
    return dx_dbl


def xdt_tm_fwd(u, wx, wdt, lib=None):
    """(x_dbl (ntok, 80), delta (ntok, dim)) = (u @ wx^T, x_dbl[:, :rank] @ wdt^T) in one pass over u (SSI:467-468 on token-major rows)"""
    lib = lib or get()
    for t in (u, wx, wdt):
        lib.check_tensor(t)
    if not xdt_tm_supported(u, wx, wdt):
        raise RuntimeError(f"xdt_tm_fwd: unsupported operands {tuple(u.shape)} {u.dtype}, {tuple(wx.shape)}, {tuple(wdt.shape)}")
    ntok, dim = u.shape
    rank = wdt.shape[1]
    ncols = wx.shape[0]
    x_dbl = torch.empty((ntok, ncols), dtype=u.dtype, device=u.device)
    delta = torch.empty((ntok, dim), dtype=u.dtype, device=u.device)
    a = XdtArgs()
    a.u, a.wx, a.wdt, a.x_dbl, a.delta = _ptr(u), _ptr(wx), _ptr(wdt), _ptr(x_dbl), _ptr(delta)
    a.ntok, a.dim, a.rank, a.ncols = ntok, dim, rank, ncols
    a.ldu, a.ldwx, a.ldwdt, a.ldx, a.ldd, a.dtype = u.stride(0), wx.stride(0), wdt.stride(0), ncols, dim, _DT[u.dtype]
    _launch(lib.c.aum_xdt_tm_fwd, a, u, lib, "xdt_tm_fwd", (ntok, dim, rank))
    return x_dbl, delta


def conv1d_tm_supported(x, width):
    """the token-major conv takes (batch, len, dim) views with contiguous channels, 16-byte aligned rows and width <= 4"""
    if x.dim() != 3 or x.dtype not in _DT or width > 4 or (x.stride(2) != 1 and x.shape[2] != 1):
        return False
    es = x.element_size()
    bs, ts = _tm3(x, "x", x.shape[2])
    return x.shape[2] % (16 // es) == 0 and x.data_ptr() % 16 == 0 and (ts * es) % 16 == 0 and (bs * es) % 16 == 0


def _al16(t):
    """the kernels read weights with 16-byte accesses: a parameter that is a misaligned view gets its own storage"""
    return t if t is None or t.data_ptr() % 16 == 0 else t.clone()


def conv1d_tm_fwd(x, weight, bias=None, silu=True, reverse=False, lib=None):
    """causal_conv1d_fn on a token-major (batch, len, dim) view (x may be the first half of the in_proj output rows) -> y (batch, len,
    dim) contiguous.  weight (dim, width) / (dim, 1, width)."""
    lib = lib or get()
    lib.check_tensor(x)
    batch, length, dim = x.shape
    weight = _al16(_f32c(weight.reshape(dim, -1)))
    bias = _al16(_f32c(bias))
    y = torch.empty((batch, length, dim), dtype=x.dtype, device=x.device)
    a = ConvTmArgs()
    a.x, a.weight, a.bias, a.y = _ptr(x), _ptr(weight), _ptr(bias), _ptr(y)
    a.x_bs, a.x_ts = _tm3(x, "x", dim)
    a.y_bs, a.y_ts = _tm3(y, "y", dim)
    a.batch, a.dim, a.len, a.width, a.dtype = batch, dim, length, weight.shape[1], _DT[x.dtype]
    a.flags = (CONV_SILU if silu else 0) | (CONV_REVERSE if reverse else 0)
    _launch(lib.c.aum_conv1d_tm_fwd, a, x, lib, "conv_tm_fwd", (batch, dim, length, x.element_size()))
    return y


def conv1d_tm_bwd(x, weight, bias, dy, silu=True, reverse=False, dx_out=None, lib=None, partials=False):
    """-> (dx (batch, len, dim), dweight (dim, width) fp32, dbias (dim) fp32 | None); dx_out may be a strided token-major view (the
    first half of d(in_proj output)).  The per-wave partial sums of dweight / dbias are added in a fixed order (aum_sum_rows_multi);
    partials=True: -> (dx, dw_part (nparts, dim, width), db_part (nparts, dim) | None) for a caller that sums them with other sets."""
    lib = lib or get()
    lib.check_tensor(x)
    batch, length, dim = x.shape
    weight = _al16(_f32c(weight.reshape(dim, -1)))
    bias = _al16(_f32c(bias))
    width = weight.shape[1]
    dy = dy if dy.stride(2) == 1 else dy.contiguous()
    dx = dx_out if dx_out is not None else torch.empty((batch, length, dim), dtype=x.dtype, device=x.device)
    nparts = int(lib.c.aum_conv1d_tm_nparts(batch, length))
    f32 = dict(dtype=torch.float32, device=x.device)
    dw_part = torch.empty((nparts, dim, width), **f32)
    db_part = torch.empty((nparts, dim), **f32) if bias is not None else None
    a = ConvTmArgs()
    a.x, a.dy, a.weight, a.bias, a.dx, a.dw_part, a.db_part = map(_ptr, (x, dy, weight, bias, dx, dw_part, db_part))
    a.x_bs, a.x_ts = _tm3(x, "x", dim)
    a.dy_bs, a.dy_ts = _tm3(dy, "dy", dim)
    a.dx_bs, a.dx_ts = _tm3(dx, "dx", dim)
    a.batch, a.dim, a.len, a.width, a.dtype = batch, dim, length, width, _DT[x.dtype]
    a.flags = (CONV_SILU if silu else 0) | (CONV_REVERSE if reverse else 0)
    _launch(lib.c.aum_conv1d_tm_bwd, a, x, lib, "conv_tm_bwd", (batch, dim, length, x.element_size()))
    if partials:
        return dx, dw_part, db_part
    if db_part is None:
        return dx, sum_rows(dw_part, lib=lib), None           # (dim, width): the partial rows are in the weight's own layout
    dweight, dbias = sum_rows_multi([dw_part, db_part], lib=lib)      # both partial sets in one launch
    return dx, dweight, dbias


def conv1d_fwd(x, weight, bias=None, silu=True, reverse=False, dmajor=False, generic=False, lib=None):
    """causal_conv1d_cuda.causal_conv1d_fwd(x, weight(dim,width), bias, None, silu) -> y (batch, dim, len) contiguous."""
    lib = lib or get()
    _unit(x, "x")
    lib.check_tensor(x)
    batch, dim, length = x.shape
    weight = _f32c(weight.reshape(dim, -1))
    bias = _f32c(bias)
    y = _alloc(batch, dim, length, x.dtype, x.device, dmajor)
    a = ConvArgs()
    a.x, a.weight, a.bias, a.y = _ptr(x), _ptr(weight), _ptr(bias), _ptr(y)
    a.x_bs, a.x_ds, a.y_bs, a.y_ds = x.stride(0), x.stride(1), y.stride(0), y.stride(1)
    a.batch, a.dim, a.len, a.width, a.dtype = batch, dim, length, weight.shape[1], _DT[x.dtype]
    a.flags = (CONV_SILU if silu else 0) | (CONV_REVERSE if reverse else 0) | (4 if generic else 0)
    _launch(lib.c.aum_causal_conv1d_fwd, a, x, lib, "conv_fwd", (batch, dim, length, x.element_size()))
    return y


def conv1d_bwd(x, weight, bias, dy, silu=True, reverse=False, dx_out=None, generic=False, lib=None):
    """causal_conv1d_cuda.causal_conv1d_bwd -> (dx, dweight(dim,width) fp32, dbias fp32|None); dx_out may be a
    preallocated strided view written in place (SSI:594-596)."""
    lib = lib or get()
    _unit(x, "x")
    _unit(dy, "dy")
    lib.check_tensor(x)
    lib.check_tensor(dy)
    batch, dim, length = x.shape
    weight = _f32c(weight.reshape(dim, -1))
    bias = _f32c(bias)
    dx = dx_out if dx_out is not None else torch.empty((batch, dim, length), dtype=x.dtype, device=x.device)
    _unit(dx, "dx")
    # dweight / dbias accumulate (one fp32 atomic per wave and tap): one zero fill for both
    zb = torch.zeros((weight.numel() + (dim if bias is not None else 0),), dtype=torch.float32, device=x.device)
    dw = zb[:weight.numel()].view_as(weight)
    db = zb[weight.numel():] if bias is not None else None
    a = ConvArgs()
    a.x, a.dy, a.weight, a.bias, a.dx, a.dweight, a.dbias = map(_ptr, (x, dy, weight, bias, dx, dw, db))
    a.x_bs, a.x_ds, a.dy_bs, a.dy_ds = x.stride(0), x.stride(1), dy.stride(0), dy.stride(1)
    a.dx_bs, a.dx_ds = dx.stride(0), dx.stride(1)
    a.batch, a.dim, a.len, a.width, a.dtype = batch, dim, length, weight.shape[1], _DT[x.dtype]
    a.flags = (CONV_SILU if silu else 0) | (CONV_REVERSE if reverse else 0) | (4 if generic else 0)
    _launch(lib.c.aum_causal_conv1d_bwd, a, x, lib, "conv_bwd", (batch, dim, length, x.element_size()))
    return dx, dw, db


def rmsnorm_fwd(x, weight, residual=None, eps=1e-5, residual_dtype=None, generic=False, lib=None):
    """_layer_norm_fwd(is_rms_norm=True) (LN:123-177).  x: (rows, cols).  Returns (y, rstd, residual_out) where
    residual_out is x itself when no new residual tensor is needed (LN:176-177)."""
    lib = lib or get()
    lib.check_tensor(x)
    rows, cols = x.shape
    assert x.stride(1) == 1
    if residual is not None:
        residual_dtype = residual.dtype
        assert residual.stride(1) == 1 and residual.shape == x.shape
    weight = _f32c(weight)
    y = torch.empty_like(x, memory_format=torch.contiguous_format)
    need_res_out = residual is not None or (residual_dtype is not None and residual_dtype != x.dtype)
    res_out = torch.empty((rows, cols), dtype=residual_dtype, device=x.device) if need_res_out else None
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device)
    a = NormArgs()
    a.x, a.residual, a.weight, a.y, a.residual_out, a.rstd_out = map(_ptr, (x, residual, weight, y, res_out, rstd))
    a.row_stride_x, a.row_stride_y = x.stride(0), y.stride(0)
    if residual is not None:
        a.row_stride_res = residual.stride(0)
    if res_out is not None:
        a.row_stride_res_out = res_out.stride(0)
    a.eps, a.rows, a.cols = eps, rows, cols
    a.x_dtype = a.y_dtype = _DT[x.dtype]
    a.res_dtype = _DT[residual_dtype] if need_res_out else _DT[x.dtype]
    a.flags = 2 if generic else 0
    _launch(lib.c.aum_rmsnorm_fwd, a, x, lib, "rmsnorm_fwd", (rows, cols, x.element_size()))
    return y, rstd, (res_out if res_out is not None else x)


def rmsnorm_bwd(dy, x_saved, weight, rstd, dresidual=None, has_residual=False, x_dtype=None, generic=False, lib=None, dw_out=None):
    """_layer_norm_bwd(is_rms_norm=True) (LN:293-377).  x_saved = residual_out of the forward.  Returns
    (dx [x_dtype], dweight fp32, dresidual_in | None)."""
    lib = lib or get()
    lib.check_tensor(dy)
    rows, cols = x_saved.shape
    x_dtype = x_dtype or x_saved.dtype
    assert dy.stride(1) == 1 and x_saved.stride(1) == 1 and dy.dtype == x_dtype
    weight = _f32c(weight)
    dx = torch.empty((rows, cols), dtype=x_dtype, device=dy.device)
    dres_in = torch.empty_like(x_saved) if (has_residual and x_dtype != x_saved.dtype) else None
    n_part = int(lib.c.aum_rmsnorm_bwd_partial_rows(rows, cols, 2 if generic else 0))     # rows the kernel leaves sums in
    dwp = torch.empty((n_part, cols), dtype=torch.float32, device=dy.device)
    a = NormArgs()
    a.x, a.dy, a.dresidual_out, a.weight, a.rstd_in = map(_ptr, (x_saved, dy, dresidual, weight, rstd))
    a.dx, a.dresidual_in, a.dweight_partial = _ptr(dx), _ptr(dres_in), _ptr(dwp)
    a.row_stride_x, a.row_stride_dy, a.row_stride_dx = x_saved.stride(0), dy.stride(0), dx.stride(0)
    if dresidual is not None:
        assert dresidual.dtype == x_saved.dtype and dresidual.stride(1) == 1
        a.row_stride_dres_out = dresidual.stride(0)
    if dres_in is not None:
        a.row_stride_dres_in = dres_in.stride(0)
    a.rows, a.cols = rows, cols
    a.x_dtype = a.y_dtype = _DT[x_dtype]
    a.res_dtype = _DT[x_saved.dtype]
    a.flags = 2 if generic else 0
    _launch(lib.c.aum_rmsnorm_bwd, a, dy, lib, "rmsnorm_bwd", (rows, cols, dy.element_size()))
    dw = sum_rows(dwp, lib, out=dw_out)
    if has_residual and dres_in is None:
        dres_in = dx
    return dx, dw, dres_in


FBANK_AUG = 8     # columns of the per-clip augmentation table (include/aum_hip.h AUM_FBANK_AUG)


def fbank_fwd(wave, tables, target_length, norm_mean, norm_std, preemph=0.97, aug=None, noise=None, lib=None):
    """Log-mel frontend.  wave: (batch, n_samples) fp32 mean-removed; tables: dict(window, twiddle, mel_start_f,
    mel_count_f, mel_w [num_mel, stride], win, shift, padded) of device tensors built by aum.frontend.FbankTables.
    aug: optional (batch, 8) fp32 per-clip table [frames, f_lo, f_hi, t_lo, t_hi, roll, noise_amp, 0] applied in the kernel's
    store (ragged padding, SpecAug bands, noise, roll); noise: (batch, target_length, num_mel) fp32 uniform numbers."""
    lib = lib or get()
    batch, num_mel = wave.shape[0], tables["mel_w"].shape[0]
    out = torch.empty((batch, target_length, num_mel), dtype=torch.float32, device=wave.device)
    a = FbankArgs()
    _fill_fbank(a, lib, wave, tables, target_length, norm_mean, norm_std, preemph, aug, noise)
    a.out, a.out_bs = _ptr(out), out.stride(0)
    _launch(lib.c.aum_fbank_fwd, a, wave, lib, "fbank_fwd", (batch, target_length, num_mel))
    return out


def _fill_fbank(a, lib, wave, tables, target_length, norm_mean, norm_std, preemph, aug, noise):
    lib.check_tensor(wave)
    assert wave.dtype == torch.float32 and wave.stride(1) == 1
    batch, n = wave.shape
    win, shift, padded = tables["win"], tables["shift"], tables["padded"]
    num_frames = 0 if n < win else min(target_length, 1 + (n - win) // shift)
    num_mel, stride = tables["mel_w"].shape
    a.wave = _ptr(wave)
    a.window, a.twiddle = _ptr(tables["window"]), _ptr(tables["twiddle"])
    a.mel_start_f, a.mel_count_f, a.mel_w = _ptr(tables["mel_start_f"]), _ptr(tables["mel_count_f"]), _ptr(tables["mel_w"])
    a.wave_bs = wave.stride(0)
    a.batch, a.n_samples, a.win, a.shift, a.padded = batch, n, win, shift, padded
    a.num_frames, a.target_length, a.num_mel, a.mel_wstride = num_frames, target_length, num_mel, stride
    a.preemph, a.norm_mean, a.norm_inv2std = preemph, norm_mean, 1.0 / (2.0 * norm_std)
    a.log_floor = 1.1920928955078125e-07
    if aug is not None:
        assert aug.dtype == torch.float32 and aug.is_contiguous() and aug.shape == (batch, FBANK_AUG)
        lib.check_tensor(aug)
        a.aug = _ptr(aug)
    if noise is not None:
        assert aug is not None and noise.dtype == torch.float32 and noise.is_contiguous()
        assert noise.shape == (batch, target_length, num_mel)
        lib.check_tensor(noise)
        a.noise = _ptr(noise)


FRONTEND_TIME_MAJOR = 1


def frontend_tokens_supported(tables, target_length, dim, dtype):
    """the limits of aum_frontend_tokens_fwd (include/aum_hip.h); outside them callers run fbank_fwd and a GEMM"""
    return (dtype in (torch.bfloat16, torch.float16) and tables["padded"] == 512 and tables["mel_w"].shape[0] == 128
            and target_length % 64 == 0 and dim % 16 == 0)


def frontend_tokens(wave, tables, target_length, norm_mean, norm_std, weight, bias, pos, cls_row, cls_pos, out_dtype=torch.float32,
                    time_major=False, save_patches=False, preemph=0.97, aug=None, noise=None, lib=None):
    """Waveform -> token sequence in one launch (aum_frontend_tokens_fwd).  weight: (dim, 256) bf16/f16 (the flattened 16 x 16
    conv weight); bias (dim), pos (n_patches, dim), cls_row (dim) or None: fp32.  Returns (tokens (batch, n_patches [+1], dim)
    in out_dtype, patches (batch * n_patches, 256) in weight.dtype or None)."""
    lib = lib or get()
    batch = wave.shape[0]
    dim = weight.shape[0]
    n_patches = target_length // 16 * 8
    assert weight.shape == (dim, 256) and weight.is_contiguous() and weight.dtype in (torch.bfloat16, torch.float16)
    assert bias.shape == (dim,) and bias.dtype == torch.float32 and bias.is_contiguous()
    assert pos.shape == (n_patches, dim) and pos.dtype == torch.float32 and pos.is_contiguous()
    for t in (weight, bias, pos):
        lib.check_tensor(t)
    n_tok = n_patches + (cls_row is not None)
    tokens = torch.empty((batch, n_tok, dim), dtype=out_dtype, device=wave.device)
    patches = torch.empty((batch * n_patches, 256), dtype=weight.dtype, device=wave.device) if save_patches else None
    a = FrontendArgs()
    _fill_fbank(a.fbank, lib, wave, tables, target_length, norm_mean, norm_std, preemph, aug, noise)
    a.weight, a.bias, a.pos, a.tokens = _ptr(weight), _ptr(bias), _ptr(pos), _ptr(tokens)
    if cls_row is not None:
        assert cls_row.shape == (dim,) and cls_row.dtype == torch.float32 and cls_row.is_contiguous()
        lib.check_tensor(cls_row)
        a.cls_row, a.cls_pos = _ptr(cls_row), cls_pos
    if patches is not None:
        a.patches = _ptr(patches)
    a.tokens_bs, a.dim, a.dtype, a.out_dtype = tokens.stride(0), dim, _DT[weight.dtype], _DT[out_dtype]
    a.flags = FRONTEND_TIME_MAJOR if time_major else 0
    _launch(lib.c.aum_frontend_tokens_fwd, a, wave, lib, "frontend_tokens", (batch, target_length, dim))
    return tokens, patches


def proj_supported(dim, dt_rank, dstate, ntok, dtype):
    """the limits of include/aum_hip.h for the MFMA projection kernels; outside them callers use library GEMMs"""
    return (dtype in (torch.bfloat16, torch.float16) and dim % 64 == 0 and dt_rank <= 64 and dt_rank + 2 * dstate <= 80
            and ntok * 80 < 2 ** 31)


def _proj_common(a, ntok, dim, dt_rank, dstate, dtype):
    a.ntok, a.dim, a.dt_rank, a.dstate, a.dtype = ntok, dim, dt_rank, dstate, _DT[dtype]


def _pad_cols8(w):
    """[rows][cols] -> [rows][ceil8(cols)] zero-padded (the kernels read the short k dimension in groups of 8)"""
    pad = (-w.shape[1]) % 8
    return w.contiguous() if pad == 0 else torch.nn.functional.pad(w, (0, pad)).contiguous()


def proj_fwd(conv_out2d, w_x, w_dt, dstate, lib=None):
    """conv_out2d [dim][ntok], w_x [dt_rank+2*dstate][dim], w_dt [dim][dt_rank] (all one 16-bit dtype, contiguous)
    -> x_dbl [dt_rank+2*dstate][ntok] (rows dt | B | C) and delta [dim][ntok] = w_dt @ dt   (SSI:467-468)."""
    lib = lib or get()
    for t in (conv_out2d, w_x, w_dt):
        lib.check_tensor(t)
        assert t.is_contiguous() and t.dtype == conv_out2d.dtype
    dim, ntok = conv_out2d.shape
    rt, dt_rank = w_x.shape[0], w_dt.shape[1]
    assert w_x.shape == (dt_rank + 2 * dstate, dim) and w_dt.shape == (dim, dt_rank)
    w_dt = _pad_cols8(w_dt)
    x_dbl = torch.empty((rt, ntok), dtype=conv_out2d.dtype, device=conv_out2d.device)
    delta = torch.empty_like(conv_out2d)
    a = ProjArgs()
    a.act, a.w_x, a.w_dt, a.x_dbl, a.out_act = _ptr(conv_out2d), _ptr(w_x), _ptr(w_dt), _ptr(x_dbl), _ptr(delta)
    _proj_common(a, ntok, dim, dt_rank, dstate, conv_out2d.dtype)
    a.len, a.w_ld = ntok, w_dt.shape[1]
    _launch(lib.c.aum_proj_fwd, a, conv_out2d, lib, "proj_fwd", (dim, ntok, rt))
    return x_dbl, delta


def proj_bwd_data(ddelta2d, w_dt_t, w_x_t, dB, dC, dconv2d, length, lib=None):
    """ddelta2d [dim][ntok], w_dt_t = W_dt^T [dt_rank][dim], w_x_t = W_x^T [dim][rt], dB/dC fp32 (batch, dstate, len)
    -> dx_dbl [rt][ntok]; dconv2d [dim][ntok] += W_x^T dx_dbl in place   (SSI:570-574, 587, 590)."""
    lib = lib or get()
    for t in (ddelta2d, w_dt_t, w_x_t, dconv2d):
        lib.check_tensor(t)
        assert t.is_contiguous() and t.dtype == ddelta2d.dtype
    lib.check_tensor(dB), lib.check_tensor(dC)
    assert dB.dtype == torch.float32 and dC.dtype == torch.float32 and dB.stride(2) == 1 and dC.stride(2) == 1
    dim, ntok = ddelta2d.shape
    dt_rank, rt = w_dt_t.shape[0], w_x_t.shape[1]
    dstate = dB.shape[1]
    assert rt == dt_rank + 2 * dstate and dB.shape[0] * length == ntok and dB.shape[2] == length
    w_x_t = _pad_cols8(w_x_t)
    dx_dbl = torch.empty((rt, ntok), dtype=ddelta2d.dtype, device=ddelta2d.device)
    a = ProjArgs()
    a.act, a.w_x, a.w_dt, a.x_dbl, a.out_act = _ptr(ddelta2d), _ptr(w_x_t), _ptr(w_dt_t), _ptr(dx_dbl), _ptr(dconv2d)
    a.dB, a.dC = _ptr(dB), _ptr(dC)
    a.dB_bs, a.dB_ns, a.dC_bs, a.dC_ns = dB.stride(0), dB.stride(1), dC.stride(0), dC.stride(1)
    _proj_common(a, ntok, dim, dt_rank, dstate, ddelta2d.dtype)
    a.len, a.w_ld = length, w_x_t.shape[1]
    _launch(lib.c.aum_proj_bwd_data, a, ddelta2d, lib, "proj_bwd_data", (dim, ntok, rt))
    return dx_dbl


def proj_bwd_weight(x2d, y2d, transpose_out, lib=None):
    """sum_t x[e][t] * y[r][t] -> [dim][nrows] (or [nrows][dim] with transpose_out) fp32   (SSI:586, 589)."""
    lib = lib or get()
    lib.check_tensor(x2d), lib.check_tensor(y2d)
    assert x2d.is_contiguous() and y2d.stride(1) == 1 and y2d.stride(0) == x2d.shape[1] and x2d.dtype == y2d.dtype
    dim, ntok = x2d.shape
    nrows = y2d.shape[0]
    nsplit = debug.proj_splits or int(lib.c.aum_proj_bwd_weight_splits(dim, ntok))
    if nsplit <= 0:
        raise RuntimeError("aum_proj_bwd_weight: unsupported shape")
    part = torch.empty((nsplit, nrows, dim) if transpose_out else (nsplit, dim, nrows), dtype=torch.float32, device=x2d.device)
    a = ProjWArgs()
    a.x, a.y, a.out, a.ntok = _ptr(x2d), _ptr(y2d), _ptr(part), ntok
    a.dim, a.nrows, a.nsplit, a.transpose_out, a.dtype = dim, nrows, nsplit, int(transpose_out), _DT[x2d.dtype]
    _launch(lib.c.aum_proj_bwd_weight, a, x2d, lib, "proj_bwd_weight", (dim, ntok, nrows))
    return sum_rows(part, lib) if nsplit > 1 else part[0]


def selftest_wave_scan(P, S, rev=False, lib=None):
    lib = lib or get()
    inp = torch.cat([P.float().reshape(64), S.float().reshape(64)]).contiguous()
    out = torch.empty_like(inp)
    _chk(lib.c.aum_selftest_wave_scan(_ptr(inp), _ptr(out), int(rev), lib.stream(inp)), "aum_selftest_wave_scan")
    return out[:64], out[64:]


def selftest_wave_sum32(values, lib=None):
    """values: (32, 64) fp32 -> (2, 64): row 0 the totals in the lane order of wave_sum32 (lane l: value 2 * (l & 15) + ((l >> 4) & 1)),
    row 1 wave_sum16 of the first 16 values"""
    lib = lib or get()
    inp = values.float().contiguous()
    out = torch.empty((2, 64), dtype=torch.float32, device=inp.device)
    _chk(lib.c.aum_selftest_wave_sum32(_ptr(inp), _ptr(out), lib.stream(inp)), "aum_selftest_wave_sum32")
    return out


def _sum_out(out, shape, device):
    """the destination of a sum: `out` (a contiguous fp32 tensor of that many elements on that device -- e.g. the bucket view a parameter's
    gradient lives in under DistributedDataParallel) or a new tensor"""
    if out is not None and out.dtype == torch.float32 and out.is_contiguous() and out.device == device and out.numel() == math.prod(shape):
        return out.view(shape)          # (4-byte aligned is enough: a view in a bucket behind an odd-sized parameter)
    return torch.empty(shape, dtype=torch.float32, device=device)


def sum_rows(t, lib=None, out=None):
    """t: (outer, ...) contiguous fp32 / bf16 / fp16 -> fp32 sum over dim 0, in a fixed order (aum_sum_rows): the partial results of
    rmsnorm_bwd / proj_bwd_weight and split-K GEMM partial products.  Tall and narrow inputs (the 4096 x 768 norm partials: too few
    columns for one pass to fill the chip) are summed in two stages, 32 slices first.  Shapes the kernel does not take go through torch.
    out: where to put the sum (see _sum_out); the result is returned either way."""
    lib = lib or get()
    inner = t[0].numel() if t.shape[0] > 0 else 0
    if debug.torch_sums or t.dtype not in _DT or not t.is_contiguous() or inner % 8 or inner == 0 or t.data_ptr() % 16:
        r = t.sum(0, dtype=torch.float32)
        if out is not None and out.shape == r.shape and out.dtype == r.dtype:
            return out.copy_(r)
        return r
    lib.check_tensor(t)
    outer, shape = t.shape[0], t.shape[1:]
    stream = lib.stream(t)
    if outer >= 1024 and inner < 8192 and outer % 32 == 0:
        mid = torch.empty((32, inner), dtype=torch.float32, device=t.device)
        _chk(lib.c.aum_sum_rows(_ptr(t), _ptr(mid), 32, outer // 32, inner, _DT[t.dtype], stream), "aum_sum_rows")
        t, outer = mid, 32
    out = _sum_out(out, shape, t.device)
    _chk(lib.c.aum_sum_rows(_ptr(t), _ptr(out), 1, outer, inner, _DT[t.dtype], stream), "aum_sum_rows")
    return out


def _two_stage(t):
    """sum_rows' rule for a tall and narrow partial set (two stages, 32 slices first)"""
    return t.shape[0] >= 1024 and t[0].numel() < 8192 and t.shape[0] % 32 == 0


def sum_rows_multi(parts, tr_cols=None, lib=None, outs=None):
    """parts: up to SUM_MAX_JOBS (outer, ...) contiguous fp32 tensors of partial results on one device -> their fp32 sums over dim 0 in ONE launch
    (aum_sum_rows_multi; bitwise the sums sum_rows gives one by one).  tr_cols[q] > 0: part q is (outer, rows, tr_cols) and its sum is
    returned transposed, (tr_cols, rows) contiguous.  Shapes the entry does not take go through sum_rows / torch one by one."""
    lib = lib or get()
    tr_cols = list(tr_cols or [0] * len(parts))
    outs = list(outs or [None] * len(parts))
    # (a set that sum_rows would add in two stages keeps that order: it goes through sum_rows, and so does every set of its call)
    ok = 0 < len(parts) <= SUM_MAX_JOBS and not debug.torch_sums and not debug.sums_one_by_one and all(
        t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] > 0 and t[0].numel() > 0 and t[0].numel() % 8 == 0 and t.data_ptr() % 16 == 0
        and not _two_stage(t) and (not tc or (t.dim() == 3 and t.shape[2] == tc)) for t, tc in zip(parts, tr_cols))
    if not ok:
        res = []
        for t, tc, o in zip(parts, tr_cols, outs):
            if tc:
                r = sum_rows(t, lib=lib).t()
                res.append(o.copy_(r) if o is not None and o.shape == r.shape else r.contiguous())
            else:
                res.append(sum_rows(t, lib=lib, out=o))
        return res
    for t in parts:
        lib.check_tensor(t)
    outs = [_sum_out(o, (t.shape[2], t.shape[1]) if tc else tuple(t.shape[1:]), t.device) for t, tc, o in zip(parts, tr_cols, outs)]
    jobs = (SumJob * len(parts))()
    for j, t, o, tc in zip(jobs, parts, outs, tr_cols):
        j.src, j.dst, j.outer, j.inner, j.tr_cols = _ptr(t), _ptr(o), t.shape[0], t[0].numel(), tc
    _chk(lib.c.aum_sum_rows_multi(C.cast(jobs, C.c_void_p), len(parts), lib.stream(parts[0])), "aum_sum_rows_multi")
    return outs


_cast_tables = {}


def cast_bank(params, dtype, want_t=False, lib=None):
    """params: equally shaped 2-D fp32 contiguous matrices on one device -> (bank (n, rows, cols) in `dtype`, bank_t (n, cols, rows) or None):
    every matrix read once, both copies written by ONE launch (aum_cast_bank).  The device table of the n addresses is kept per address
    tuple (parameters are updated in place: their addresses are stable across steps)."""
    lib = lib or get()
    p0 = params[0]
    rows, cols = p0.shape
    key = (p0.device, tuple(p.data_ptr() for p in params))
    tab = _cast_tables.get(key)
    if tab is None:
        if len(_cast_tables) > 64:
            _cast_tables.clear()
        tab = _cast_tables[key] = torch.tensor(key[1], dtype=torch.int64, device=p0.device)
    for p in params:
        lib.check_tensor(p)
    bank = torch.empty((len(params), rows, cols), dtype=dtype, device=p0.device)
    bank_t = torch.empty((len(params), cols, rows), dtype=dtype, device=p0.device) if want_t else None
    _chk(lib.c.aum_cast_bank(_ptr(tab), len(params), rows, cols, _ptr(bank), _ptr(bank_t), _DT[dtype], lib.stream(p0)), "aum_cast_bank")
    return bank, bank_t


def cast_bank_supported(params, dtype):
    p0 = params[0]
    return (dtype in (torch.bfloat16, torch.float16) and p0.dim() == 2 and p0.shape[0] % 8 == 0 and p0.shape[1] % 4 == 0
            and all(p.dtype == torch.float32 and p.is_contiguous() and p.shape == p0.shape and p.device == p0.device and p.data_ptr() % 16 == 0
                    for p in params))


def hbm_copy(src, dst, lib=None):
    lib = lib or get()
    _chk(lib.c.aum_hbm_copy(_ptr(src), _ptr(dst), src.numel() * src.element_size(), lib.stream(src)), "aum_hbm_copy")
