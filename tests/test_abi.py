"""CPU: the C-ABI shared library loads and exports every symbol include/aum_hip.h declares, and the ctypes
structure layouts match the header.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

import aum_hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aum_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aum_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def so_path():
    import importlib.util
    spec = importlib.util.spec_from_file_location("aum_build", os.path.join(ROOT, "audio-mamba-aum_amd", "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def test_every_declared_symbol_is_exported(so_path):
    names = declared_functions()
    assert len(names) >= 12 and set(aum_hip.EXPORTS) == set(names)
    lib = ctypes.CDLL(so_path)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.aum_abi_version() == aum_hip.ABI_VERSION == 13
    assert lib.aum_scan_max_single_pass_len() == 576
    assert lib.aum_rmsnorm_bwd_partials(32832) == 4096
    assert lib.aum_rmsnorm_bwd_partial_rows(32832, 768, 0) == 512 and lib.aum_rmsnorm_bwd_partial_rows(32832, 768, 2) == 4096
    assert lib.aum_rmsnorm_bwd_partial_rows(5, 768, 0) == 1 and lib.aum_rmsnorm_bwd_partial_rows(5, 4096, 0) == 5


def test_struct_layouts_match_header(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with the ctypes mirrors."""
    probes = {
        "AumScanFwdArgs": (aum_hip.ScanFwdArgs, ["u", "A", "out", "workspace_bytes", "u_bs", "out_ds", "batch", "flags", "x_ck", "x_lane"]),
        "AumScanBwdArgs": (aum_hip.ScanBwdArgs, ["u", "dout", "A", "du", "dA", "workspace_bytes", "dC_ns", "batch", "flags", "x_ck", "x_lane"]),
        "AumConvArgs": (aum_hip.ConvArgs, ["x", "weight", "y", "dweight", "x_bs", "dx_ds", "batch", "flags"]),
        "AumNormArgs": (aum_hip.NormArgs, ["x", "weight", "y", "rstd_out", "row_stride_x", "eps", "rows", "flags"]),
        "AumFbankArgs": (aum_hip.FbankArgs, ["wave", "mel_w", "out", "wave_bs", "batch", "mel_wstride", "preemph", "log_floor", "aug", "noise"]),
        "AumFrontendArgs": (aum_hip.FrontendArgs, ["fbank", "weight", "bias", "pos", "cls_row", "tokens", "patches", "tokens_bs", "dim", "cls_pos",
                                                   "dtype", "out_dtype", "flags"]),
        "AumProjArgs": (aum_hip.ProjArgs, ["act", "w_dt", "out_act", "dB", "dC_ns", "ntok", "dim", "dtype", "w_ld"]),
        "AumProjWArgs": (aum_hip.ProjWArgs, ["x", "y", "out", "ntok", "dim", "nsplit", "dtype"]),
        "AumScanTmFwdArgs": (aum_hip.ScanTmFwdArgs, ["u", "C", "A", "delta_bias", "out", "out_pre", "ckpt", "u_bs", "C_ts", "pre_ts", "batch",
                                                     "dstate", "dtype", "flags"]),
        "AumScanTmBwdArgs": (aum_hip.ScanTmBwdArgs, ["u", "dout", "out_pre", "A", "ckpt", "du", "dz", "dA", "dBC", "ddelta_bias", "workspace",
                                                     "workspace_bytes", "u_bs", "pre_ts", "du_bs", "dz_ts", "batch", "dtype", "flags", "dA_xA", "dA_b_xA"]),
        "AumConvTmArgs": (aum_hip.ConvTmArgs, ["x", "dy", "weight", "bias", "y", "dx", "dw_part", "db_part", "x_bs", "dx_ts", "batch", "width",
                                               "dtype", "flags"]),
        "AumDtProjArgs": (aum_hip.DtProjArgs, ["x", "w", "out", "ntok", "dim", "rank", "ldx", "ldw", "ldo", "dtype"]),
        "AumXdtArgs": (aum_hip.XdtArgs, ["u", "wx", "wdt", "x_dbl", "delta", "ntok", "dim", "rank", "ncols", "ldu", "ldwx", "ldwdt", "ldx", "ldd", "dtype"]),
        "AumXdtBwdArgs": (aum_hip.XdtBwdArgs, ["ddelta", "dbc", "wdt_t", "wx_t", "du", "dx_dbl", "ntok", "dim", "rank", "ncols", "ldd", "lddbc", "ldwdt", "ldwx",
                                               "ldu", "ldx", "dtype"]),
        "AumGemmArgs": (aum_hip.GemmArgs, ["a", "b", "c", "m", "n", "k", "lda", "ldb", "ldc", "dtype", "flags"]),
        "AumGemmWArgs": (aum_hip.GemmWArgs, ["y", "x", "part", "t", "ldy", "ldx", "n", "k", "splits", "dtype"]),
        "AumSumJob": (aum_hip.SumJob, ["src", "dst", "outer", "inner", "tr_cols", "reserved"]),
        "AumScanTmSegFwdArgs": (aum_hip.ScanTmSegFwdArgs, ["base", "carry", "carry_bytes", "segments"]),
        "AumScanTmSegBwdArgs": (aum_hip.ScanTmSegBwdArgs, ["base", "segments"]),
        "AumConvUpdateArgs": (aum_hip.ConvUpdateArgs, ["x", "conv_state", "weight", "bias", "out", "batch", "dim", "width", "dtype", "flags"]),
        "AumStateUpdateArgs": (aum_hip.StateUpdateArgs, ["state", "x", "dt", "z", "B", "C", "A", "D", "dt_bias", "out", "batch", "dim", "dstate", "dtype", "flags"]),
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void){']
    for cname, (_, fields) in probes.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines += ['return 0;}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, (cls, fields) in probes.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f in fields:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, (cname, f)


def test_product_library_refuses_host_tensors(so_path):
    import torch
    lib = aum_hip.Lib(so_path)
    with pytest.raises(RuntimeError, match="no CPU path"):
        aum_hip.conv1d_fwd(torch.zeros(1, 2, 8), torch.zeros(2, 4), None, lib=lib)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ImportError, match="no CPU fallback"):
        aum_hip.Lib(str(tmp_path / "libaum_hip.so"))


def test_build_refuses_counted_wait_kernels_that_spill():
    """csrc/build.py reads hipcc's kernel-resource-usage remarks: a kernel with hand-counted vmcnt waits (k_scant_*, k_gemm_tn*, k_gemm_wgrad*)
    that uses scratch memory is reported (and the build raises); the channel-major kernels, which wait through the compiler, may spill"""
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("aum_build", os.path.join(here, "..", "audio-mamba-aum_amd", "csrc", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    log = """
x.hip:1:1: remark: Function Name: _ZN3aum11k_scant_bwdINS_6bf16_tELb1ELb1ELb1EEEv16AumScanTmBwdArgsNS_11ScanTBwdOutE [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     ScratchSize [bytes/lane]: 16 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark: Function Name: _ZN3aum12k_scanwg_fwdIfLi9ELi0ELi0EEEv14AumScanFwdArgsi [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     ScratchSize [bytes/lane]: 224 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark: Function Name: _ZN4aumg12k_gemm_tn_psILb1EEEvNS_10GemmLaunchE [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]
"""
    assert b.spilling_kernels(log) == [("_ZN3aum11k_scant_bwdINS_6bf16_tELb1ELb1ELb1EEEv16AumScanTmBwdArgsNS_11ScanTBwdOutE", 16)]
    assert b.spilling_kernels("") == []
