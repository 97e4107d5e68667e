"""Run the acting forward (n = 1024) and one train step a few times -- target for rocprofv3 --pmc."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet
n = 1024
net = QNet(max_batch=n); net.init_params(0)
states = ((torch.rand((n, 80, 80, 4), device="cuda") < 0.37).to(torch.uint8) * 255).contiguous()
for _ in range(5):
    net.act(states, 0.03)
B = 32
s = states[:B].contiguous(); s2 = states[B:2 * B].contiguous()
a = torch.zeros(B, dtype=torch.uint8, device="cuda"); r = torch.full((B,), 0.1, device="cuda"); t = torch.zeros(B, dtype=torch.uint8, device="cuda")
for _ in range(5):
    net.train_step("dqn", s, a, r, s2, t)
torch.cuda.synchronize()
