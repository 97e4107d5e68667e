"""The full loop of bench.py (act -> env step -> push -> sample -> gather -> train), 60 steps -- target for
rocprofv3 --kernel-trace; tools/trace_gaps.py then gives per-kernel durations and idle gaps in situ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay
n, B = 1024, 32
env, replay, net = VecGameState(n, seed=0), VecReplay(1_000_000, n), QNet(max_batch=n)
replay.seed(0, "cpython")
net.init_params(0)
nib = env.track_state()
env.observe(); replay.reset(env.frame_bits)
for step in range(60):
    actions = net.act_nib(nib, 0.03, seed=0, step=step)
    env.frame_step(actions, want_u8=False)
    replay.push(env.frame_bits, actions, env.reward, env.terminal)
    idx, _ = replay.sample(B)
    s, a, r, s2, t = replay.gather(idx)
    net.train_step("dqn", s, a, r, s2, t, want_aux=False)
torch.cuda.synchronize()
