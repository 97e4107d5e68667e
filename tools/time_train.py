"""HIP-event time of every kernel of one train step (B = 32, plain DQN), like bench.py's kernel leg."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd import _lib as L
if os.environ.get("FB_LIB"):                      # an ablation / tuning build of the library
    L.LIB_PATH = os.path.abspath(os.environ["FB_LIB"])
from dqnflappybird_amd.vec import QNet
lib = L.lib()
R = 100
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ALGO = {"dqn": 0, "nature": 1, "double": 2}[sys.argv[2] if len(sys.argv) > 2 else "dqn"]
net = QNet(max_batch=1024); net.init_params(0)
s = ((torch.rand((B, 80, 80, 4), device="cuda") < 0.37).to(torch.uint8) * 255).contiguous()
s2 = ((torch.rand((B, 80, 80, 4), device="cuda") < 0.37).to(torch.uint8) * 255).contiguous()
a = torch.zeros(B, dtype=torch.uint8, device="cuda"); r = torch.full((B,), 0.1, device="cuda"); t = torch.zeros(B, dtype=torch.uint8, device="cuda")
loss = torch.zeros(1, device="cuda")
for _ in range(50):
    net.train_step(ALGO, s, a, r, s2, t, want_aux=False)
st = L.current_stream()
out, tot = [], 0.0
for k in range(64):
    name = lib.fb_qnet_kernel_name(k).decode()
    if not name:
        break
    def run():
        L.check(lib.fb_qnet_profile_kernel(net.h, k, R, ALGO, B, L.ptr(s), L.ptr(a), L.ptr(r), L.ptr(s2), L.ptr(t), L.ptr(loss), st), "profile")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / R
    tot += us
    out.append(f"{name.replace('_kernel', '')} {us:.1f}")
print("  ".join(out), f" sum {tot:.1f}", flush=True)
