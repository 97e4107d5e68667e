"""random.sample(range(n), k) on the device, k = 32 and 256, from a 1 M-slot memory -- target for tools/trace_run.sh (kernel durations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import VecReplay
rep = VecReplay(1_000_000, 1024)
rep.seed(0, "cpython")
bits = torch.zeros((1024, 100), dtype=torch.int64, device="cuda")
a = torch.zeros(1024, dtype=torch.uint8, device="cuda"); r = torch.zeros(1024, device="cuda"); t = torch.zeros(1024, dtype=torch.uint8, device="cuda")
rep.reset(bits)
for _ in range(990):
    rep.push(bits, a, r, t)
torch.cuda.synchronize()
ks = [int(x) for x in os.environ.get("FB_TRACE_KS", "32,256").split(",")]
for k in ks:
    for _ in range(100):
        rep.sample(k)
    torch.cuda.synchronize()
