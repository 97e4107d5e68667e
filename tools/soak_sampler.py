"""fb_vec_step's riding sampler against CPython's random.sample over thousands of steps (block regenerations, duplicate
candidates -> the serial fallback, a saturating ring): every step's indices must be the reference stream's."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
N, B, CAP = 256, 32, 3000                     # small population: a duplicate candidate every few draws
env, rep, net = VecGameState(N, seed=1), VecReplay(CAP, N), QNet(max_batch=N)
rep.seed(123, "cpython"); net.init_params(0)
env.track_state(); env.observe(); rep.reset(env.frame_bits)
one = VecStep(env, rep, net, B, "dqn")
rng, size, bad = random.Random(123), 0, 0
for step in range(steps):
    one(0.1, seed=7, step=step)
    size = min(size + N, CAP)
    if one.idx.tolist() != rng.sample(range(size), B):
        bad += 1
        print("mismatch at step", step); break
assert bad == 0 and env.error_count() == 0 and len(rep) == CAP
assert torch.isfinite(net.store_params()).all()
print(f"{steps} steps: sampled indices identical to random.sample; loss {one.loss.item():.4g}")
