"""Train-only leg (sample -> gather -> train_step), eager then hipGraph -- target for rocprofv3 --kernel-trace.
tools/trace_gaps.py turns the trace into per-kernel durations and the idle gap in front of each launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay
n, B = 1024, int(os.environ.get("FB_TRACE_BATCH", "32"))
algo = os.environ.get("FB_TRACE_ALGO", "dqn")
env, replay, net = VecGameState(n, seed=0), VecReplay(1_000_000, n), QNet(max_batch=n)
net.init_params(0)
env.observe(); replay.reset(env.frame_bits)
acts = (torch.rand(n, device="cuda") < 0.1).to(torch.uint8)
for t in range(40):
    env.frame_step(acts, want_u8=False)
    replay.push(env.frame_bits, acts, env.reward, env.terminal)


def train():
    idx, _ = replay.sample(B)
    s, aa, r, s2, tt = replay.gather(idx)
    net.train_step(algo, s, aa, r, s2, tt, want_aux=False)


for _ in range(5):
    train()
torch.cuda.synchronize()
if os.environ.get("FB_TRACE_GRAPH", "1") == "1":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            train()
    for _ in range(3):
        g.replay()
else:
    for _ in range(30):
        train()
torch.cuda.synchronize()
