"""why is configs[3] slow inside bench.py?  python tools/dbg_per_slow.py A|B|C"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as bc
mode = sys.argv[1]
def row(i):
    r = bc.gpu_row(i, 25 if i == 3 else 50, min_total=0.1)
    print(mode, r["config"][:50], r["us_per_step"], flush=True)
if mode == "A":            # the other rows first
    row(1); row(2); row(3)
elif mode == "B":          # a split net first
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
    env, rep, net = VecGameState(1024, seed=0), VecReplay(1_000_000, 1024), QNet(max_batch=1024)
    rep.seed(0, "cpython"); net.init_params(0); env.track_state(); env.observe(); rep.reset(env.frame_bits)
    one = VecStep(env, rep, net, 32, "dqn")
    for s in range(200): one(0.03, seed=0, step=s)
    torch.cuda.synchronize(); print("split stats", net.split_stats())
    row(3)
    del one, env, rep, net
    import gc; gc.collect(); torch.cuda.synchronize()
    row(3)
elif mode == "C":          # a hipGraph first
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, TrainSteps
    env, rep, net = VecGameState(1024, seed=0), VecReplay(1_000_000, 1024), QNet(max_batch=1024)
    rep.seed(0, "cpython"); net.init_params(0); env.track_state(); env.observe(); rep.reset(env.frame_bits)
    acts = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    for s in range(100):
        env.frame_step(acts, want_u8=False); rep.push(env.frame_bits, acts, env.reward, env.terminal)
    g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    t10 = TrainSteps(rep, net, 32, "dqn")
    with torch.cuda.stream(side): t10(1)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    with torch.cuda.graph(g): t10(10)
    g.replay(); torch.cuda.synchronize()
    row(3)
if mode.startswith("D"):   # k live streams created first: does the prioritized memory's side stream land on the caller's hardware queue?
    k = int(mode[1:])
    keep = [torch.cuda.Stream() for _ in range(k)]
    for s_ in keep:
        with torch.cuda.stream(s_):
            torch.zeros(8, device="cuda").add_(1)
    torch.cuda.synchronize()
    row(int(os.environ.get("DBG_ROW", "3")))
