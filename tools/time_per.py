"""HIP-event time of the prioritized-replay kernels (1 M-slot tree): store of N envs (reference order / FB_PER_FAST), sample, update."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import VecReplay
for N in (1024, 4096):
    per = VecReplay(1_000_000, N, prioritized=True)
    per.seed(0, "numpy")
    bits = torch.zeros((N, 100), dtype=torch.int64, device="cuda")
    a = torch.zeros(N, dtype=torch.uint8, device="cuda"); r = torch.zeros(N, device="cuda"); t = torch.zeros(N, dtype=torch.uint8, device="cuda")
    per.reset(bits)
    for _ in range(80):
        per.push(bits, a, r, t)
    idx, _ = per.sample(32)
    err = torch.rand(32, device="cuda")
    per.update_priorities(idx, abs_err=err)
    def ev(fn, R=50):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / R
    out = [f"N={N}"]
    for mode in ("exact", "fast"):
        per.set_per_mode(mode)
        out.append(f"{mode}: push+store {ev(lambda: per.push(bits, a, r, t)):.1f}  sample {ev(lambda: per.sample(32)):.1f}  update {ev(lambda: per.update_priorities(idx, abs_err=err)):.1f} us")
    print("  ".join(out), flush=True)
    del per
