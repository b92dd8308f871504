"""N>1 path on CPU: world_size=2, gloo backend (replicas + one gradient exchange per step)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ddp_two_ranks_gloo():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "ddp_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"DDP_RESULT worst_rel_grad_err=(\S+) grads_identical=(\S+) weights_identical=(\S+)", r.stdout)
    assert m, r.stdout[-2000:]
    assert float(m.group(1)) < 1e-4 and m.group(2) == "True" and m.group(3) == "True", m.group(0)
