"""The driver's N = 2 command on a one-GPU box: both ranks on cuda:0, torch.distributed over gloo instead of RCCL (RCCL refuses two
ranks on one device).  Everything else is the real multi-GPU path of bench.py: env / replay shards per rank, the gradient-exporting
fb_vec_step, one all-reduce of the flat gradient per step, fb_qnet_apply_adam, barrier + MAX-over-ranks timing -- and the run's own
check that the replicas are still bit-identical afterwards."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu_over_gloo():
    env = dict(os.environ, FB_BENCH_SINGLE_DEVICE="1", FB_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3",
           "--no-kernel-legs", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                # rank 0 alone prints, one line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "weak"
    assert out["config"]["replicas_bit_identical"] is True
    assert out["value"] > 0 and out["config"]["train_only"]["grad_steps_per_sec"] > 0
