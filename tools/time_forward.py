"""HIP-event time of every kernel of the acting forward on real game states (the env kernel's nibble state
after 60 random steps), at a few env counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd import _lib as L
from dqnflappybird_amd.vec import QNet, VecGameState
lib = L.lib()
R = 200
for n in [int(x) for x in (sys.argv[1:] or ["1024", "4096"])]:
    env = VecGameState(n, seed=0)
    nib = env.track_state()
    env.observe()
    for t in range(60):
        env.frame_step((torch.rand(n, device="cuda") < 0.1).to(torch.uint8), want_u8=False)
    net = QNet(max_batch=n); net.init_params(0)
    for _ in range(3000 * 1024 // n):                     # ~0.5 s of load: let the clocks ramp before timing
        net.act_nib(nib, 0.0)
    torch.cuda.synchronize()
    st = L.current_stream()
    out = []
    for k in range(5):
        def run():
            L.check(lib.fb_qnet_profile_kernel(net.h, k, R, -2, n, L.ptr(nib), None, None, None, None, None, st), "profile")
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        out.append(f"{lib.fb_qnet_kernel_name(k).decode().replace('_kernel', '')} {e0.elapsed_time(e1) * 1e3 / R:.1f}")
    print(f"n={n}: " + "  ".join(out), flush=True)
    del net, env
