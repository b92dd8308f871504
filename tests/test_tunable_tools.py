"""CPU: the GEMM solution file tooling -- aum/tunable.py's scratch copy and environment, tools/merge_tunable.py's merge rule."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSV = os.path.join(ROOT, "audio-mamba-aum_amd", "aum", "tunableop_gfx950.csv")


def test_recorded_file_has_the_single_launch_shapes_of_the_headline_step():
    """the library GEMMs the bench's step issues since round 5 (one launch per projection: 64 x 513 = 32 832 token rows) have recorded
    solutions -- a missing one would be tuned online in every bench run's warm-up"""
    keys = {tuple(ln.split(",")[:2]) for ln in open(CSV).read().splitlines() if ln and not ln.startswith("Validator,")}
    for op, shape in (("GemmTunableOp_BFloat16_TN", "tn_3072_32832_768_ld_768_768_3072"),        # in_proj forward
                      ("GemmTunableOp_BFloat16_NN", "nn_768_32832_3072_ld_768_3072_768"),        # in_proj data gradient
                      ("GemmTunableOp_BFloat16_TN", "tn_768_32832_1536_ld_1536_1536_768"),       # out_proj forward
                      ("GemmTunableOp_BFloat16_NN", "nn_1536_32832_768_ld_1536_768_1536")):      # out_proj data gradient (AUM_GEMM=lib)
        assert (op, shape) in keys, shape
    assert len(keys) == sum(1 for ln in open(CSV).read().splitlines() if ln and not ln.startswith("Validator,")), "one line per (operation, shape)"


def test_merge_adds_missing_shapes_only(tmp_path):
    work = tmp_path / "repo"
    (work / "tools").mkdir(parents=True)
    (work / "audio-mamba-aum_amd" / "aum").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tools", "merge_tunable.py"), work / "tools" / "merge_tunable.py")
    dst = work / "audio-mamba-aum_amd" / "aum" / "tunableop_gfx950.csv"
    dst.write_text("Validator,PT_VERSION,2.10.0\nGemmTunableOp_BFloat16_TN,tn_8_8_8_ld_8_8_8,Default,0.01\n")
    new = tmp_path / "new.csv"
    new.write_text("Validator,PT_VERSION,2.10.0\nGemmTunableOp_BFloat16_TN,tn_8_8_8_ld_8_8_8,Gemm_Rocblas_1,0.02\n"
                   "GemmTunableOp_BFloat16_NN,nn_16_8_8_ld_16_8_16,Gemm_Hipblaslt_7,0.03\n")
    run = lambda *a: subprocess.run([sys.executable, str(work / "tools" / "merge_tunable.py"), *a], capture_output=True, text=True)
    r = run(str(new))
    assert r.returncode == 0 and "1 added, 0 replaced" in r.stdout, r.stdout + r.stderr
    lines = dst.read_text().splitlines()
    assert lines[1].endswith("Default,0.01") and lines[2].startswith("GemmTunableOp_BFloat16_NN,nn_16_8_8")
    r = run(str(new), "--replace")
    assert "0 added, 1 replaced" in r.stdout and dst.read_text().splitlines()[1].endswith("Gemm_Rocblas_1,0.02")
    other = tmp_path / "other.csv"
    other.write_text("Validator,PT_VERSION,2.11.0\nGemmTunableOp_BFloat16_NN,nn_1_1_1_ld_1_1_1,Default,0.01\n")
    r = run(str(other))
    assert r.returncode != 0 and "validators differ" in (r.stdout + r.stderr)


def test_enable_copies_the_recorded_file_and_sets_the_environment(tmp_path, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "audio-mamba-aum_amd"))
    from aum import tunable
    for k in [k for k in os.environ if k.startswith("PYTORCH_TUNABLEOP_")]:
        monkeypatch.delenv(k)
    monkeypatch.delenv("AUM_NO_TUNABLEOP", raising=False)
    monkeypatch.delenv("AUM_TUNABLEOP_DUMP", raising=False)
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    import tempfile
    tempfile.tempdir = None
    try:
        d = tunable.enable(3)
        assert os.environ["PYTORCH_TUNABLEOP_ENABLED"] == "1" and os.environ["PYTORCH_TUNABLEOP_TUNING"] == "1"
        assert os.environ["PYTORCH_TUNABLEOP_FILENAME"] == os.path.join(d, "results.csv")
        for ordinal in (0, 3):          # TunableOp appends the device ordinal: rank-local copy and the masked-device case
            assert open(os.path.join(d, f"results{ordinal}.csv")).read() == open(CSV).read()
        monkeypatch.setenv("AUM_NO_TUNABLEOP", "1")
        assert tunable.enable(0) is None
    finally:
        tempfile.tempdir = None
        for k in [k for k in os.environ if k.startswith("PYTORCH_TUNABLEOP_")]:
            os.environ.pop(k, None)
