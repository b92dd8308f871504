"""TEST INFRASTRUCTURE: build tests/emu/_build/libaum_emu.so (host lane-array build of the kernel sources)."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "audio-mamba-aum_amd", "csrc")
CXX = os.environ.get("AUM_EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
DEPS = [os.path.join(HERE, "aum_emu.cpp")] + [os.path.join(CSRC, f) for f in
        ("aum_api.inc", "wave.h", "scan_kernels.h", "scan_wg_kernels.h", "scan_half_kernels.h", "scan_row_kernels.h", "fbank_kernels.h", "frontend_kernels.h", "proj_kernels.h", "conv_rows_kernels.h", "conv_norm_kernels.h", "scan_tm_kernels.h", "conv_tm_kernels.h", "aum_api_tm.inc", "gemm_args.h", "dtproj_args.h", "xdt_args.h", "decode_args.h", "cast_args.h")] + [os.path.join(ROOT, "include", "aum_hip.h")]


def build(extra_flags=(), tag=""):
    """tag / extra_flags: an opt-in build variant of the kernel sources (e.g. -DAUM_SCANT_DBC_MERGE=0) next to the default library"""
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, f"libaum_emu{tag}.so")
    h = hashlib.sha256()
    for d in DEPS:
        h.update(open(d, "rb").read())
    extra = os.environ.get("AUM_EXTRA_CXXFLAGS", "").split() + list(extra_flags)
    h.update(" ".join(extra).encode())
    dig = h.hexdigest()
    stamp = os.path.join(out_dir, "digest" + tag)
    if os.path.exists(so) and os.path.exists(stamp) and open(stamp).read() == dig:
        return so
    subprocess.check_call([CXX, "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-Wno-unused-variable", "-Wno-unused-function"] + extra +
                          [os.path.join(HERE, "aum_emu.cpp"), "-o", so])
    open(stamp, "w").write(dig)
    return so


if __name__ == "__main__":
    print(build())
