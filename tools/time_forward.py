"""HIP-event time of every kernel of the acting forward (u8 states) at a few batch sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd import _lib as L
from dqnflappybird_amd.vec import QNet
lib = L.lib()
R = 50
for n in [int(x) for x in (sys.argv[1:] or ["1024", "4096"])]:
    net = QNet(max_batch=n); net.init_params(0)
    states = ((torch.rand((n, 80, 80, 4), device="cuda") < 0.37).to(torch.uint8) * 255).contiguous()
    net.act(states, 0.0)
    st = L.current_stream()
    out = []
    for k in range(5):
        def run():
            L.check(lib.fb_qnet_profile_kernel(net.h, k, R, -1, n, L.ptr(states), None, None, None, None, None, st), "profile")
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        out.append(f"{lib.fb_qnet_kernel_name(k).decode().replace('_kernel', '')} {e0.elapsed_time(e1) * 1e3 / R:.1f}")
    print(f"n={n}: " + "  ".join(out), flush=True)
