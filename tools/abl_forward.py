"""time_forward.py against an ablation build of the library: python tools/abl_forward.py <lib.so> [n ...].
(Ablation builds compute wrong results on purpose; they only answer "what does the kernel wait for".)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dqnflappybird_amd import _lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from dqnflappybird_amd.vec import QNet, VecGameState
lib = L.lib()
R = 200
for n in [int(x) for x in (sys.argv[2:] or ["1024"])]:
    env = VecGameState(n, seed=0)
    nib = env.track_state()
    env.observe()
    for t in range(60):
        env.frame_step((torch.rand(n, device="cuda") < 0.1).to(torch.uint8), want_u8=False)
    net = QNet(max_batch=n); net.init_params(0)
    for _ in range(3000 * 1024 // n):
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
    print(f"{os.path.basename(sys.argv[1])} n={n}: " + "  ".join(out), flush=True)
    del net, env
