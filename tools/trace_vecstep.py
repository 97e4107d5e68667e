"""The loop bench.py times (fb_vec_step: act -> env -> push[+sample] -> gather[+draw ahead] -> train, one host call per
step), 150 steps -- target for rocprofv3 --kernel-trace; tools/trace_gaps.py CSV 400 then gives the in-situ duration of
every kernel of a step and the idle gap in front of it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
n, B = int(os.environ.get("FB_TRACE_ENVS", "1024")), int(os.environ.get("FB_TRACE_BATCH", "32"))
algo = os.environ.get("FB_TRACE_ALGO", "dqn")                  # "per": prioritized memory (reference-order tree), the loop of configs[3]
env, replay, net = VecGameState(n, seed=0), VecReplay(1_000_000, n, prioritized=algo == "per"), QNet(max_batch=n)
replay.seed(0, "numpy" if algo == "per" else "cpython")
net.init_params(0)
if os.environ.get("FB_TRACE_DTYPE", "f32") == "bf16":       # configs[2]'s stated dtype: bf16 acting and training
    net.set_inference_dtype("bf16"); net.set_train_dtype("bf16")
nib = env.track_state()
env.observe(); replay.reset(env.frame_bits)
one = VecStep(env, replay, net, B, algo)
train = os.environ.get("FB_TRACE_TRAIN", "1") == "1"          # 0: act -> env -> push only (W_fc1's planes never go stale)
for step in range(int(os.environ.get("FB_TRACE_STEPS", "150"))):      # (the split schedule's minibatches are mostly clean only once the memory holds ~300 steps)
    one(0.03, seed=0, step=step, train=train)
torch.cuda.synchronize()
