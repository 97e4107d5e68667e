"""Throughput of BASELINE.json's other single-GPU configurations through the device-resident loop (VecBrain):
configs[2] Double-DQN 4096 envs batch 256 (fp32 training, fp32 or bf16 acting), configs[3] prioritized replay with a
1 M-slot SumTree and 4096 envs, configs[4]'s per-GPU share (dueling, 4096 envs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vecbrain import VecBrain
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rows = [("configs[2] double, B=256, fp32 acting", dict(algo="double", batch=256), None),
        ("configs[2] double, B=256, bf16 acting", dict(algo="double", batch=256), "bf16"),
        ("configs[3] PER 1M slots (exact order)", dict(algo="per", batch=32), "exact"),
        ("configs[3] PER 1M slots (FB_PER_FAST)", dict(algo="per", batch=32), "fast"),
        ("configs[4] dueling, per-GPU share", dict(algo="nature", arch="dueling", batch=32), None)]
for name, kw, mode in rows:
    vb = VecBrain(4096, capacity=1_000_000, observe=20, seed=1, **kw)
    if mode == "bf16":
        vb.net.set_inference_dtype("bf16")
    if mode in ("exact", "fast"):
        vb.replay.set_per_mode(mode)
    n = steps if mode != "exact" else max(steps // 8, 30)
    vb.run(30, log_every=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vb.run(n, log_every=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(vb.last_loss).all()
    print(f"{name:42s} {dt / n * 1e6:8.1f} us/step  {4096 * n / dt / 1e6:6.2f} M env-steps/s  {n / dt:8.0f} grad-steps/s", flush=True)
    del vb
