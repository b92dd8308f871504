"""N>1 path on CPU: world_size=2, gloo backend (replicas + one gradient exchange per step)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="2", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"DDP_RESULT worst_rel_grad_err=(\S+) grads_identical=(\S+) weights_identical=(\S+)", r.stdout)
    assert m, r.stdout[-2000:]
    return float(m.group(1)), m.group(2) == "True", m.group(3) == "True"


def test_ddp_two_ranks_gloo():
    err, same_g, same_w = _run({})
    assert err < 1e-4 and same_g and same_w


def test_ddp_two_ranks_gloo_bf16_gradient_exchange():
    """--grad_compress bf16: buckets travel in bf16 (half the bytes on the xGMI ring); the reduced gradient is the global-batch
    gradient to bf16 rounding, and still bit-identical on every rank."""
    err, same_g, same_w = _run({"AUM_TEST_COMPRESS": "bf16"})
    assert 1e-7 < err < 1.2e-2 and same_g and same_w
