"""A few acting forwards on real game states (for rocprofv3 --pmc runs: tools/pmc_act.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet, VecGameState
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = VecGameState(n, seed=0)
nib = env.track_state()
env.observe()
for t in range(60):
    env.frame_step((torch.rand(n, device="cuda") < 0.1).to(torch.uint8), want_u8=False)
net = QNet(max_batch=n); net.init_params(0)
for _ in range(30):
    net.act_nib(nib, 0.0)
torch.cuda.synchronize()
