"""The multi-rank path of bench.py on the GPU box: a fresh two-process torch.distributed.run of the real bench (gloo for the
control exchanges, both ranks on the one GPU the box has).  No scaling claim -- this checks that the rank path runs and reports."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
           "--steps", "1", "--warmup", "0", "--batch", "512", "--cpu-sample", "0"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 1 and r["value"] > 0 and r["scaling"] == "weak"
    assert r["config"]["batch_per_gpu"] == 512
    assert 0.0 < r["converged_frac"] < 1.0 and r["frames_correct_frac"] == r["converged_frac"]
    assert r["roofline"]["frac"] > 0 and "cpu_baseline" not in r
