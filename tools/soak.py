"""Soak run: the device-resident loop for a few thousand steps per algorithm (ring wrap, target syncs, PER updates),
checking that the loss stays finite, the env reports no invalid action and the replay indices stay in range."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vecbrain import VecBrain
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
import numpy as np
for algo, arch, fast in (("dqn", "plain", False), ("nature", "plain", False), ("double", "dueling", False), ("per", "plain", True), ("per", "plain", False)):
    vb = VecBrain(1024, algo=algo, arch=arch, capacity=200_000, observe=50, seed=3)
    if fast:
        vb.replay.set_per_mode("fast")
    t0 = time.perf_counter()
    vb.run(steps, log_every=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    loss = vb.last_loss.item()
    ep, ssum, smax, pipes = vb.stats.tolist()
    assert loss == loss and abs(loss) < 1e6, loss
    assert vb.env.error_count() == 0 and len(vb.replay) == 200_000
    p = vb.net.store_params()
    assert torch.isfinite(p).all()
    if algo == "per":                                # the SumTree after thousands of stores / samples / batch_updates (reference order: through the side stream)
        tree, ptr, size, beta = vb.replay.per_state()
        cap = 200_000
        assert size == cap and np.isclose(tree[0], tree[cap - 1:].sum(), rtol=1e-9), (size, tree[0], tree[cap - 1:].sum())
        i = np.arange(0, cap - 1)
        np.testing.assert_allclose(tree[i], tree[2 * i + 1] + tree[2 * i + 2], rtol=1e-9, atol=1e-12)
        assert beta == 1.0 or steps < 700
    print(f"{algo:7s} {arch:8s} {steps} steps  {1024 * steps / dt / 1e6:.2f} M env-steps/s  loss {loss:.4g}  episodes {ep}  max score {smax}", flush=True)
