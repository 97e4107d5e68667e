"""A few hundred data-parallel steps at world size 1 for rocprofv3 --kernel-trace: python3 tools/trace_dp.py [plain|overlap] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from dqnflappybird_amd.dist import OverlappedAllReduce
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
mode = sys.argv[1] if len(sys.argv) > 1 else "overlap"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
env, rep, net = VecGameState(1024, seed=0), VecReplay(1_000_000, 1024), QNet(max_batch=1024)
rep.seed(0, "cpython"); net.init_params(0)
env.track_state(); env.observe(); rep.reset(env.frame_bits)
g = torch.zeros(net.n_params, device="cuda")
one = VecStep(env, rep, net, 32, "dqn", flat_grad=g)
red = OverlappedAllReduce(net, g, False, force=True) if mode == "overlap" else None
for i in range(K):
    one(0.03, seed=0, step=i)
    if red is not None:
        red()
    else:
        dist.all_reduce(g)
    net.apply_adam(g)
torch.cuda.synchronize()
dist.destroy_process_group()
