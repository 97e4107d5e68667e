"""World-size-1 cost of the data-parallel step (the N > 1 hot path of bench.py / VecBrain.step) against the fused single-GPU step:
    fused          fb_vec_step()                                   Adam inside the train step
    dp             fb_vec_step(flat_grad = g) -> all_reduce(g) -> fb_qnet_apply_adam(g)      (what every rank runs when N > 1)
    dp_no_rccl     the same without the collective call            (what the gradient export + the stand-alone Adam cost by themselves)
    dp_native      fb_vec_step_dp: the all-reduce issued from the C side on the library's own RCCL communicator, on the step's stream
    dp_native_ov   the same in two pieces, the W_fc1 / head part on a side stream behind the fc1 backward launch (fb_dist_set_overlap)
    dp_overlap     dist.OverlappedAllReduce: the W_fc1 / head part of the gradient reduced on a side stream behind the fc1 backward launch
with a one-rank RCCL process group, so the all-reduce is a real RCCL call on the step's stream.  python tools/time_dp_step.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep

K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


NATIVE = None


def build(with_grad, native=False):
    global NATIVE
    env, rep, net = VecGameState(1024, seed=0), VecReplay(1_000_000, 1024), QNet(max_batch=1024)
    rep.seed(0, "cpython"); net.init_params(0)
    env.track_state(); env.observe(); rep.reset(env.frame_bits)
    g = torch.zeros(net.n_params, device="cuda") if with_grad else None
    if native and NATIVE is None:
        from dqnflappybird_amd.dist import NativeDP
        NATIVE = NativeDP()
    if native:
        from dqnflappybird_amd import _lib as L
        L.check(L.lib().fb_dist_set_overlap(NATIVE.handle, int(native == "overlap")), "fb_dist_set_overlap")
    return env, rep, net, g, VecStep(env, rep, net, 32, "dqn", flat_grad=g, dist=NATIVE if native else None)


def run(name, with_grad, rccl):
    env, rep, net, g, one = build(with_grad, native={"native": "inline", "native_overlap": "overlap"}.get(rccl, False))
    red = None
    if rccl == "overlap":
        from dqnflappybird_amd.dist import OverlappedAllReduce
        red = OverlappedAllReduce(net, g, False, force=True)

    def step(i):
        one(0.03, seed=0, step=i)
        if with_grad and rccl not in ("native", "native_overlap"):
            if red is not None:
                red()
            elif rccl:
                dist.all_reduce(g)
            net.apply_adam(g)
    for i in range(60):
        step(i)
    ts, hs = [], []
    for rep_ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K):
            step(60 + rep_ * K + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / K * 1e6); hs.append((t1 - t0) / K * 1e6)
    print(f"{name:12s} {sorted(ts)[2]:7.1f} us/step  (min {min(ts):.1f}, max {max(ts):.1f});  host issue time {sorted(hs)[2]:.1f} us/step", flush=True)


run("fused", False, False)
run("dp_no_rccl", True, False)
run("dp", True, True)
run("dp_overlap", True, "overlap")
run("dp_native", True, "native")
run("dp_native_ov", True, "native_overlap")
dist.destroy_process_group()
