"""The train-only leg of bench.py (fb_train_steps(10) captured in one hipGraph, replayed) -- target for rocprofv3 --kernel-trace;
tools/trace_gaps.py CSV 300 then gives the in-situ duration of every kernel of a step and the idle gap in front of it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet, TrainSteps, VecGameState, VecReplay
n, B = 1024, 32
env, replay, net = VecGameState(n, seed=0), VecReplay(1_000_000, n), QNet(max_batch=n)
replay.seed(0, "cpython")
net.init_params(0)
env.observe(); replay.reset(env.frame_bits)
acts = (torch.rand(n, device="cuda") < 0.1).to(torch.uint8)
for t in range(60):
    env.frame_step(acts, want_u8=False)
    replay.push(env.frame_bits, acts, env.reward, env.terminal)
ts = TrainSteps(replay, net, B, os.environ.get("FB_TRACE_ALGO", "dqn"))
ts(2)
torch.cuda.synchronize()
if os.environ.get("FB_TRACE_GRAPH", "1") == "1":
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ts(10)
    for _ in range(6):
        g.replay()
else:
    for _ in range(6):
        ts(10)
torch.cuda.synchronize()
