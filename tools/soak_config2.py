"""Soak run of BASELINE configs[2] (Double-DQN, 4096 envs, batch 256) in fp32 and bf16: finite loss / parameters, no env or replay error."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vecbrain import VecBrain
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for dtype in ("f32", "bf16"):
    vb = VecBrain(4096, algo="double", batch=256, capacity=1_000_000, observe=20, seed=3)
    vb.set_dtype(dtype)
    t0 = time.perf_counter()
    vb.run(steps, log_every=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = vb.last_loss.item()
    assert loss == loss and abs(loss) < 1e6, loss
    assert vb.env.error_count() == 0
    assert torch.isfinite(vb.net.store_params()).all()
    print(f"configs[2] {dtype}: {steps} steps  {4096 * steps / dt / 1e6:.2f} M env-steps/s  {dt / steps * 1e6:.1f} us/step  loss {loss:.4g}", flush=True)
