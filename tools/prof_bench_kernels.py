"""The kernels of one bench step, a few launches each -- target for rocprofv3 --pmc passes (HBM traffic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, train_from_replay
n, B = 1024, 32
env, replay, net = VecGameState(n, seed=0), VecReplay(1_000_000, n), QNet(max_batch=n)
net.init_params(0)
nib = env.track_state()
env.observe(); replay.reset(env.frame_bits)
acts = (torch.rand(n, device="cuda") < 0.1).to(torch.uint8)
for t in range(40):                                   # fill a little of the ring
    env.frame_step(acts, want_u8=False)
    replay.push(env.frame_bits, acts, env.reward, env.terminal)
for t in range(5):
    a = net.act_nib(nib, 0.03, step=t)
    env.frame_step(a, want_u8=False)
    replay.push(env.frame_bits, a, env.reward, env.terminal)
    idx, _ = replay.sample(B)
    s, aa, r, s2, tt = replay.gather(idx)
    net.train_step("dqn", s, aa, r, s2, tt, want_aux=False)
    train_from_replay(replay, net, "dqn", idx)        # the ring-fed step both loops run (conv23_t_kernel<3, true, true>, conv_dw21_kernel<2, true>)
big = torch.randint(0, 40000, (4096,), dtype=torch.int64, device="cuda")
mid = torch.randint(0, 40000, (256,), dtype=torch.int64, device="cuda")
for _ in range(3):
    replay.gather(big)
    replay.gather(mid)
torch.cuda.synchronize()
