"""HIP-event time of every kernel of one ring-fed train step (what fb_vec_step launches): python tools/time_train_ring.py [B] [algo] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqnflappybird_amd import _lib as L
if os.environ.get("FB_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["FB_LIB"])
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, train_from_replay
lib = L.lib()
R = 100
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
algo = sys.argv[2] if len(sys.argv) > 2 else "dqn"
ALGO = {"dqn": 0, "nature": 1, "double": 2}[algo]
N = 1024
env, rep = VecGameState(N, seed=3), VecReplay(200000, N)
env.observe(); rep.reset(env.frame_bits)
rng = np.random.default_rng(0)
for _ in range(100):
    acts = torch.from_numpy((rng.random(N) < 0.1).astype(np.uint8)).cuda()
    env.frame_step(acts, want_u8=False); rep.push(env.frame_bits, acts, env.reward, env.terminal)
net = QNet(max_batch=max(B, 256)); net.init_params(0); net.sync_target()
if len(sys.argv) > 3:
    net.set_train_dtype(sys.argv[3])
idx = torch.from_numpy(rng.integers(0, len(rep), B)).cuda()
for _ in range(50):
    loss, a, r, t = train_from_replay(rep, net, algo, idx)
st = L.current_stream()
out, tot = [], 0.0
for k in range(64):
    name = lib.fb_qnet_kernel_name(k).decode()
    if not name:
        break
    def run():
        L.check(lib.fb_profile_ring_kernel(rep.h, net.h, k, R, ALGO, B, L.ptr(idx), L.ptr(a), L.ptr(r), L.ptr(t), L.ptr(loss), st), "profile")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / R - 0.0
    tot += us
    out.append(f"{k}:{us:.1f}")
print(f"B={B} {algo}: " + "  ".join(out), f" sum {tot:.1f}   (0 conv1 1 trunk/conv23 3 fc1 4 head 5 loss 6 fc1_bwd 7 conv_bx 8 conv_dw21 10 slab 11 adam; each includes ~2 x 2 us of guarded re-split launches)", flush=True)
