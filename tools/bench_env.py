"""Quick env-only throughput probe (not the contract bench; see bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import VecGameState

for n in (1024, 4096, 32768):
    env = VecGameState(n, seed=0)
    acts = (torch.rand(n, device="cuda") < 0.1).to(torch.uint8)
    for want_u8 in (True, False):
        for _ in range(20):
            env.frame_step(acts, want_u8=want_u8)
        torch.cuda.synchronize()
        T = 300
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(T):
            env.frame_step(acts, want_u8=want_u8)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / T
        print(f"n_envs={n} u8={want_u8} {ms*1e3:.1f} us/step  {n/ms*1e3/1e6:.2f} M env-steps/s  "
              f"write {n*(6400*want_u8+800)/ms/1e6:.1f} GB/s")
