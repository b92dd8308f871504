#!/usr/bin/env python3
"""A/B builds of the time-serial scan kernels: recompiles only the bf16 objects of the tm forward (part 5) and backward (part 6)
with extra -D flags and links them with the other objects of the regular build into
audio-mamba-aum_amd/aum_hip/variants/libaum_hip_<name>.so (git-ignored, travels with the gpurun snapshot; tools/tm_time.py
and tools/tm_gpu_check.py take the variant name).

  python tools/tm_variant.py <name> [-DFOO=1 ...] [--asm] [--api]
--asm also writes /tmp/aumt/<name>_p6.s; --api recompiles the host part as well (flags that change it: AUM_SCANT_TRACE)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "audio-mamba-aum_amd", "csrc")
sys.path.insert(0, CSRC)
import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    defs = [a for a in sys.argv[2:] if a.startswith("-D")]
    B.build()
    vdir = os.path.join(B.OUT_DIR, "variants")
    odir = os.path.join(B.OBJ_DIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    os.makedirs(odir, exist_ok=True)
    objs, procs = [], []
    for oname, odefs in B.parts():
        if oname in ("scantm_p5_d1.o", "scantm_p6_d1.o") or (oname == "api.o" and "--api" in sys.argv):
            o = os.path.join(odir, f"{name}_{oname}")
            procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + odefs + defs + ["-c", B.SRC, "-o", o]))
            if "--asm" in sys.argv and "p6" in oname:
                os.makedirs("/tmp/aumt", exist_ok=True)
                procs.append(subprocess.Popen([B.HIPCC] + B.FLAGS + odefs + defs + ["-S", "--cuda-device-only", B.SRC, "-o",
                                                                                     f"/tmp/aumt/{name}_p6.s"]))
            objs.append(o)
        else:
            objs.append(os.path.join(B.OBJ_DIR, oname))
    for p in procs:
        if p.wait() != 0:
            raise SystemExit("hipcc failed")
    so = os.path.join(vdir, f"libaum_hip_{name}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print(so)


if __name__ == "__main__":
    main()
