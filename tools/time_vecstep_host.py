"""fb_vec_step loop (bench workload): host issue time per step (before the sync) and wall time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
n, B = int(os.environ.get("FB_TRACE_ENVS", "1024")), int(os.environ.get("FB_TRACE_BATCH", "32"))
env, replay, net = VecGameState(n, seed=0), VecReplay(1_000_000, n), QNet(max_batch=n)
replay.seed(0, "cpython"); net.init_params(0)
nib = env.track_state(); env.observe(); replay.reset(env.frame_bits)
one = VecStep(env, replay, net, B, "dqn")
for step in range(300):
    one(0.03, seed=0, step=step, train=step >= 10)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    for step in range(400):
        one(0.03, seed=0, step=1000 * rep + step, train=True)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"us per step {dt / 400 * 1e6:.1f}   host issue {th / 400 * 1e6:.1f}   split stats {net.split_stats()}", flush=True)
