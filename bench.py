#!/usr/bin/env python3
"""bench.py -- the contract benchmark of the MI355X Flappy-Bird DQN hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): per GPU 1024 vectorised envs + BrainDQN (uniform replay,
1 M-slot ring, batch 32, fp32).  One *step* = one pass of the whole hot path over one batch of
envs: getAction for 1024 envs from their currentState (forward + epsilon-greedy) -> frame_step (render +
preprocess fused) -> store -> random.sample(32) -> minibatch gather -> _trainQNetwork (target
forward, forward, backward, Adam; with N > 1 one RCCL all-reduce of the flat gradient).
`value` = env-steps/s over all ranks in that loop.  The train-only leg (random.sample -> train step fed from the frame ring)
gives `grad_steps_per_sec` (also inside `config.train_only` and `roofline`, where the driver's record keeps it), the env-only leg
`env_only_steps_per_sec`.  Everything is device
resident when the timed region starts; nothing crosses PCIe inside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS, BATCH, CAPACITY = 1024, 32, 1_000_000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 / fp16 MFMA, dense (the F16 forms take the same cycles)

# algorithmic flops / bytes per launch (n = samples in the launch)
FWD_FLOP = {"conv1_pool_kernel": 2 * 400 * 32 * 256, "conv2_kernel": 2 * 25 * 64 * 512,
            "conv3_kernel": 2 * 25 * 64 * 576, "fc1_kernel": 2 * 1600 * 512, "head_kernel": 2 * 512 * 2}
FWD_FLOP["conv23_t_kernel"] = FWD_FLOP["conv2_kernel"] + FWD_FLOP["conv3_kernel"]      # conv2 + conv3 of one state in one launch
FWD_FLOP["fc1_fk_kernel"] = FWD_FLOP["fc1_kernel"] + FWD_FLOP["head_kernel"]           # fc1 + the head's per-tile shares
BWD_FLOP = {"fc1_bwd2_kernel": 2 * 2 * 1600 * 512,                # loss + head + fc1 dW + dX in one launch
            # the whole conv backward per sample in one launch: conv3^T + conv2^T data gradients, conv3 / conv2 / conv1 weight gradients
            # (+ W_fc1's Adam span: HBM)
            "conv_bw_kernel": 2 * 25 * 576 * 64 + 2 * 25 * 512 * 64 + 2 * 25 * 576 * 64 + 2 * 25 * 512 * 64 + 2 * 400 * 256 * 32}
GATHER_BYTES = 102_417          # per sampled transition
ADAM_BYTES = 28                 # per parameter
ENV_BYTES = 6_400 + 64          # per env-step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-legs", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[2] / configs[3] legs (tools/profile_round.sh: the rocprofv3 kernel statistics then hold the headline workload's launches only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dqnflappybird_amd import _lib as L
    from dqnflappybird_amd.vec import QNet, TrainSteps, VecGameState, VecReplay, VecStep

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    L.require_gpu()
    # rehearsal switches for a one-GPU box (the real multi-GPU run uses neither): all ranks on cuda:0 over gloo
    dev_index = 0 if os.environ.get("FB_BENCH_SINGLE_DEVICE") else local_rank
    backend = os.environ.get("FB_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- the other single-GPU configurations (rank 0, N = 1 only)
    # BASELINE.json configs[2] (Double-DQN, 4096 envs, B = 256, in fp32 and in its stated bf16) and configs[3] (prioritized replay, 1 M-slot
    # SumTree in the reference's update order, 4096 envs) through the same device-resident loop (VecBrain: act -> env -> store -> sample ->
    # train, one train step per env step), >= 100 ms timed each, median.  Reported under config.other_configs; never part of `value`.
    # They run FIRST, while the process holds next to no streams.  (HIP deals streams out over a handful of hardware queues and pipes: with
    # the headline pipeline's streams alive the prioritized memory's side stream once landed beside the caller's queue in a way that cost
    # 680 us per step against 258 - 277; the library now checks every side stream against its caller's and replaces one that fails --
    # DESIGN.md section 4 "Streams", tools/dbg_per_slow.py -- the order stays as the simpler guarantee.)
    other = None
    if rank == 0 and world == 1 and not args.no_kernel_legs and not args.no_other_configs:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs as bc
        other = []
        for i in (1, 2, 3):
            r_ = bc.gpu_row(i, 50 if i != 3 else 25, min_total=0.1)
            other.append({"workload": r_["config"], "us_per_step": r_["us_per_step"], "env_steps_per_s": r_["env_steps_per_s"],
                          "grad_steps_per_s": r_["grad_steps_per_s"], "dtype": "bf16" if r_["mode"] == "bf16" else "f32", "repeats": r_["repeats"]})
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- build the pipeline
    seed = 0
    env = VecGameState(N_ENVS, seed=seed + rank)            # envs shard by rank, own Philox streams
    replay = VecReplay(CAPACITY, N_ENVS)
    replay.seed(seed + rank, "cpython")                     # bit-exact random.sample stream
    net = QNet(2, 512, "plain", max_batch=N_ENVS)
    net.init_params(seed=seed)                              # identical replicas on every rank
    # rehearsal switch for a one-GPU box: run the N > 1 code path (gradient export, fb_vec_step_dp, the library's RCCL communicator) at world size 1
    force_dp = os.environ.get("FB_BENCH_FORCE_DP") == "1"
    grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda") if world > 1 or force_dp else None
    nib = env.track_state()                                 # agents' 4-frame stacks, kept by the env kernel
    env.observe()
    replay.reset(env.frame_bits)
    eps = 0.03                                              # INITIAL_EPSILON (BrainDQN.py:25)

    from dqnflappybird_amd.vec import train_from_replay

    def train(step):                                        # the separate calls: random.sample -> train step fed from the ring
        idx, _ = replay.sample(BATCH)
        if world > 1:
            train_from_replay(replay, net, "dqn", idx, flat_grad=grad)
            dist.all_reduce(grad)                           # sum loss (BrainDQN.py:162) -> plain sum
            net.apply_adam(grad)
        else:
            train_from_replay(replay, net, "dqn", idx)

    # the whole step as one host call (fb_vec_step: act on the nibble states -> env -> store + random.sample ->
    # train from the ring; the same launches as the separate calls, without the interpreter between them).
    # N > 1: fb_vec_step_dp -- the same step exporting its gradient, the all-reduce (sum loss, BrainDQN.py:162 -> plain sum) through the
    # library's own RCCL communicator in two pieces (W_fc1 / head part on a side stream behind the fc1 backward launch, conv part on the
    # step's stream) and Adam, still one host call -- with FB_DP_NATIVE=1 (opt-in: that path has never run at world size > 1).  Default:
    # torch.distributed's all-reduce between fb_vec_step and fb_qnet_apply_adam (FB_DP_OVERLAP=1: in two pieces, dist.OverlappedAllReduce).
    from dqnflappybird_amd.dist import NativeDP, NativeUnavailable, OverlappedAllReduce, native_wanted
    native = None
    if (world > 1 or force_dp) and backend == "nccl" and (native_wanted() or force_dp):
        try:                                                 # all ranks succeed, or all ranks get NativeUnavailable (dist.NativeDP)
            native = NativeDP(rank, world)
        except NativeUnavailable as e:                       # pragma: no cover  (no RCCL to load on some rank)
            print(f"[bench] rank {rank}: the library's RCCL communicator is unavailable ({e}); torch.distributed's all-reduce", file=sys.stderr)
    one_step = VecStep(env, replay, net, BATCH, "dqn", flat_grad=grad, dist=native)
    want_overlap = os.environ.get("FB_DP_OVERLAP", "0") == "1"
    reduce_grad = OverlappedAllReduce(net, grad, mean_loss=False) if world > 1 and backend == "nccl" and want_overlap and native is None else None

    def full_step(step):
        one_step(eps, seed=seed + rank, step=step)
        if world > 1 and native is None:
            if reduce_grad is not None:
                reduce_grad()
            else:
                dist.all_reduce(grad)
            net.apply_adam(grad)

    local_times = []

    def timed(fn, k, first=0):
        barrier()
        t0 = time.perf_counter()
        for i in range(k):
            fn(first + i)
        barrier()
        dt = time.perf_counter() - t0
        local_times.append(dt)                              # (this rank's own clock, before the MAX over ranks)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        return dt

    def timed_repeat(fn, k, first=0, min_total=0.2, max_rep=400):
        """The contract's measurement -- exactly k steps between barrier + synchronize, MAX over ranks -- repeated until the
        measurements add up to >= min_total seconds: with the driver's --steps 20 one measurement is a 3 ms single shot that one
        scheduler hiccup can spoil.  Returns the list of per-measurement times (identical on every rank, so every rank takes
        the same number of repeats); callers report the median."""
        dts = []
        while sum(dts) < min_total and len(dts) < max_rep:
            dts.append(timed(fn, k, first + len(dts) * k))
        return dts

    def median(xs):
        xs = sorted(xs)
        return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])

    # ---------------------------------------------------------------- leg A: the full loop (value)
    for i in range(args.warmup):
        full_step(i)
    dts_full = timed_repeat(full_step, args.steps, first=args.warmup)
    dts_full_local = list(local_times)
    dt = median(dts_full)
    ms_per_step = dt / args.steps * 1e3
    env_steps_per_s = world * N_ENVS * args.steps / dt
    # which schedule the loop ran in: fb_vec_step's split schedule (train chain on the caller's stream BESIDE acting + env on the net's
    # side stream; same results as one stream, bit for bit) -- how many steps took it, and how many of their minibatches could start
    # beside their env step (the rest hold a transition the step itself appends and wait for it).  Then the same loop on ONE stream,
    # for the record (A/B switch fb_vec_step_set_schedule; not part of `value`).
    split_issued, split_clean = net.split_stats()             # (raises if a wait between the two streams gave up)
    L.check(L.lib().fb_vec_step_set_schedule(0), "fb_vec_step_set_schedule")
    for i in range(args.warmup):
        full_step(i)
    n_before = len(local_times)
    dts_one = timed_repeat(full_step, args.steps, first=args.warmup, min_total=0.1)
    del local_times[n_before:]
    L.check(L.lib().fb_vec_step_set_schedule(1), "fb_vec_step_set_schedule")
    one_stream_ms = median(dts_one) / args.steps * 1e3

    # ---------------------------------------------------------------- leg B: env only
    acts = (torch.rand(N_ENVS, device="cuda") < 0.1).to(torch.uint8)
    dts_env = timed_repeat(lambda i: env.frame_step(acts, want_u8=False), args.steps)
    dt_env = median(dts_env)
    env_only = world * N_ENVS * args.steps / dt_env

    env_by_n = {str(N_ENVS): round(env_only, 1)}
    if rank == 0 and not args.no_kernel_legs:                # env-only at N = 4096 and 32768 as well
        for n_big in (4096, 32768):
            eb = VecGameState(n_big, seed=seed)
            eb.observe()
            ab = (torch.rand(n_big, device="cuda") < 0.1).to(torch.uint8)
            for _ in range(20):
                eb.frame_step(ab, want_u8=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                eb.frame_step(ab, want_u8=False)
            torch.cuda.synchronize()
            env_by_n[str(n_big)] = round(n_big * 100 / (time.perf_counter() - t0), 1)
            del eb

    # ---------------------------------------------------------------- leg C: train only (sample -> gather -> train)
    graph_used = False
    if world == 1:
        try:                                                # one hipGraph = 10 train steps, no host work inside
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            train10 = TrainSteps(replay, net, BATCH, "dqn")
            with torch.cuda.stream(side):
                train10(1)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                train10(10)                                  # fb_train_steps: 10 x (sample -> gather -> train), one host call
            g.replay()
            torch.cuda.synchronize()
            graph_used = True
        except Exception as e:                              # pragma: no cover
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); eager train leg", file=sys.stderr)
    if graph_used:
        reps = max(1, args.steps // 10)
        dts_tr = timed_repeat(lambda i: g.replay(), reps)
        grad_steps = 10 * reps
    else:
        dts_tr = timed_repeat(train, args.steps)
        grad_steps = args.steps
    dt_tr = median(dts_tr)
    grad_steps_per_s = world * grad_steps / dt_tr
    dts_tr_eager = timed_repeat(train, args.steps)
    dt_tr_eager = median(dts_tr_eager)
    grad_steps_eager = world * args.steps / dt_tr_eager

    # ---------------------------------------------------------------- N > 1: the data-parallel invariant, checked on the run itself
    # every rank started from the same parameters and applied the same reduced gradients in legs A and C, so the replicas must still be
    # bit-identical (element-wise MIN == MAX over ranks; a NaN fails it too).  Outside every timed region.
    replicas_identical = None
    if world > 1:
        lo = net.store_params()
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))
        if not replicas_identical and rank == 0:
            print(f"[bench] REPLICAS DIVERGED: {int((lo != hi).sum())} of {lo.numel()} parameters differ between ranks", file=sys.stderr)

    # ---------------------------------------------------------------- N > 1: what the step's collective costs, and which path ran
    # (so that a scaling record explains its own efficiency: the all-reduce alone, bracketed by HIP events on the stream it is issued
    # from -- torch.distributed's goes through RCCL's own stream and back, which the events then include -- median of 30; and every
    # rank's own median step time, min / max over ranks)
    dp_path, allreduce_us, rank_ms = None, None, None
    if grad is not None:
        dp_path = ("native_overlap" if native.overlap else "native") if native is not None else ("torch_overlap" if reduce_grad is not None else "torch")
    if world > 1:
        scratch_g = torch.zeros_like(grad)
        ts = []
        for i in range(35):
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if native is not None:
                native.all_reduce(scratch_g)
            else:
                dist.all_reduce(scratch_g)
            e1.record()
            torch.cuda.synchronize()
            if i >= 5:
                ts.append(e0.elapsed_time(e1) * 1e3)
        tt = torch.tensor([median(ts)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        allreduce_us = round(tt.item(), 2)
        mine = torch.tensor([median(dts_full_local) / args.steps * 1e3], dtype=torch.float64, device="cuda")
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        rank_ms = [round(lo.item(), 4), round(hi.item(), 4)]

    # ---------------------------------------------------------------- leg D: per-kernel HIP-event timing
    kernels = []
    roofline = None
    if rank == 0 and not args.no_kernel_legs:
        def ev_time(fn, reps):
            # (>= 20 ms of the same launches first: an idle GPU clocks down within milliseconds, and a cold first millisecond of a 30 us
            # kernel measures 15 % long)
            t_w = time.perf_counter()
            while True:
                fn()
                torch.cuda.synchronize()
                if time.perf_counter() - t_w > 0.02:
                    break
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                                     # torch's current stream == the stream we launch on
            fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps         # us per launch

        tpath = os.path.join(ROOT, "profiles", "traffic.json")      # PMC bytes per launch, rocprofv3 --pmc passes
        pmc = json.load(open(tpath)) if os.path.exists(tpath) else {}

        def add(name, us, per_step, bound, work, split=0):
            """bound: what limits the kernel as DESIGN.md section 4 states it -- "mfma" | "hbm" (achieved / peak / frac priced on the
            ALGORITHMIC flops / bytes of SURVEY 8d) | "lds" | "latency" (dependent chains: the figure of merit is the time per
            launch; achieved = frac = null, no bandwidth is claimed).
            split = s > 0: the kernel computes every fp32 product as s fp16 x fp16 MFMA products of two-plane operands (DESIGN.md
            section 4: conv1 2 -- its u8 input is exact in fp16 --, conv2 / conv3 / fc1 3).  `achieved` / `frac` stay ALGORITHMIC
            (fp32-equivalent flops / time / dense fp16 MFMA peak); `achieved_issued` / `frac_issued` count the s products the matrix
            pipe really executes."""
            split = split or (2 if name.startswith("conv1_pool_kernel") else 3 if name.startswith("conv23_t_kernel") else 0)
            k = {"kernel": name, "us": round(us, 3), "launches_per_step": per_step, "bound": bound, "traffic": pmc.get(name)}
            if bound in ("lds", "latency"):
                k.update({"achieved": None, "peak": None, "unit": "us per launch", "frac": None})
                if work:
                    k["algorithmic_GBps"] = round(work / us / 1e3, 1)
            else:
                peak = HBM_PEAK_GBS if bound == "hbm" else (MFMA_BF16_PEAK_TF if split else MFMA_F32_PEAK_TF)
                ach = work / us / 1e3 if bound == "hbm" else work / us / 1e6      # GB/s | TFLOP/s
                k.update({"achieved": round(ach, 3), "peak": peak, "unit": "GB/s" if bound == "hbm" else "TFLOP/s", "frac": round(ach / peak, 5)})
                if split:
                    k["dtype"] = f"f16x{split:g} (fp32 result)"
                    k["frac_basis"] = "algorithmic: SURVEY 8(d) fp32-equivalent flops / time / dense fp16 MFMA peak"
                    k["achieved_issued"] = round(split * ach, 3)
                    k["frac_issued"] = round(split * ach / peak, 5)
                    k["note"] = (f"the kernel issues {split:.3g} fp16 MFMA products per fp32 MAC (frac_issued); peak = nominal dense fp16/bf16 MFMA; "
                                 "tools/mb/mb_mfma3.hip measures 1.47 PFLOP/s sustained on random operands with all CUs busy, 2.3 on zeros")
            if name.startswith("gather_kernel"):
                # the replay gather: the figure is the PMC one (HBM bytes the counters saw / time / 8 TB/s); the logical-byte basis of
                # SURVEY 8(d) prices bytes the launch never moves (the ring stores frames as bits and shares them between s and s') and
                # can exceed 1 -- kept as a footnote, never as `frac`
                k["logical_GBps"], k["logical_frac"] = k["achieved"], k["frac"]
                k["logical_basis"] = "SURVEY 8d: 102 417 B per sampled transition (footnote: more bytes than the launch moves)"
                if k["traffic"]:
                    k["achieved"] = round(k["traffic"] / us / 1e3, 3)
                    k["frac"] = round(k["traffic"] / us / 1e3 / HBM_PEAK_GBS, 5)
                    k["frac_basis"] = "PMC HBM bytes per launch (profiles/traffic.json) / time / 8 TB/s"
                else:
                    k["achieved"] = k["frac"] = None
            kernels.append(k)

        scratch = QNet(2, 512, "plain", max_batch=N_ENVS)   # profile on a scratch net (Adam really steps)
        scratch.init_params(seed=1)
        idx, _ = replay.sample(BATCH)
        s, a, r, s2, t = replay.gather(idx)
        states = replay.current_state()
        loss = torch.zeros(1, device="cuda")
        scratch.train_step("dqn", s, a, r, s2, t, want_aux=False)
        R = 50
        lib = L.lib()
        st = L.current_stream
        # acting forward, n = 1024
        scratch.act_nib(nib, 0.0)
        # (>= 256 states, forward only: the two-plane fp16 kernels; conv2 + conv3 in one launch)
        # (>= 256 nibble states: conv1 + conv2 + conv3 in ONE launch, conv1's output handed to conv2 in LDS; then fc1 on K slices)
        c1f, c23f = FWD_FLOP["conv1_pool_kernel"], FWD_FLOP["conv2_kernel"] + FWD_FLOP["conv3_kernel"]
        act = [(0, "conv1_sp_kernel<nib>", c1f, 2),                  # (a launch of its own only with FB_ACT_FUSED=0)
               (1, "conv23_sp_kernel<C1>[conv1 + pool + conv2 + conv3, 5 states per workgroup]", c1f + c23f, (2 * c1f + 3 * c23f) / (c1f + c23f)),
               (3, "fc1_sp_kernel", FWD_FLOP["fc1_kernel"], 3), (4, "head_kernel", FWD_FLOP["head_kernel"], 0)]
        for k, name, flop, split in act:
            us = ev_time(lambda: L.check(lib.fb_qnet_profile_kernel(scratch.h, k, R, -2, N_ENVS, L.ptr(nib), None, None, None,
                                                                  None, None, st()), "profile"), R)
            if us < 0.8:                                     # not a launch of this plan
                continue
            if k == 1 and any(x["kernel"].startswith("conv1_sp_kernel") for x in kernels):      # FB_ACT_FUSED=0: the two-launch form
                name, flop, split = "conv23_sp_kernel", c23f, 3
            # (inside fb_vec_step the head does not get a launch of its own: it rides in the env step launch)
            add(name + "[act n=1024]", us, 0 if name == "head_kernel" else 1, "mfma", flop * N_ENVS, split)
            if k == 1 and split != 3:
                kernels[-1]["dtype"] = "f16x2 (conv1) / f16x3 (conv2, conv3), fp32 result"
        # train step, B = 32 (forward kernels see 2B samples: s and s')
        scratch.train_step("dqn", s, a, r, s2, t, want_aux=False)
        for k in range(64):
            name = lib.fb_qnet_kernel_name(k).decode()
            if not name:
                break
            us = ev_time(lambda: L.check(lib.fb_qnet_profile_kernel(scratch.h, k, R, 0, BATCH, L.ptr(s), L.ptr(a), L.ptr(r),
                                                                  L.ptr(s2), L.ptr(t), L.ptr(loss), st()), "profile"), R)
            if us < 0.8:                                     # not a launch of this plan (rides in a neighbour at B = 32, see fb_qnet_kernel_name)
                continue
            if name in ("conv1_pool_kernel", "conv23_t_kernel"):
                # the gathered-minibatch form (fb_qnet_train_step on u8 states); both measured loops run the ring-fed trunk below instead
                add(name + "[train 2B=64, gathered minibatch]", us, 0, "mfma", FWD_FLOP[name] * 2 * BATCH)
            elif name in FWD_FLOP:
                add(name + "[train 2B=64]", us, 1, "mfma", FWD_FLOP[name] * 2 * BATCH)
            elif name == "conv_bw_kernel":                    # carries W_fc1's Adam update (22.9 MB of HBM traffic) as extra workgroups
                add(name + "[gathered minibatch: per-sample conv3^T / conv2^T chain + conv3 / conv2 / conv1 dW + Adam of W_fc1 (HBM part priced)]", us, 0, "hbm", ADAM_BYTES * 1600 * 512)      # (full loop: the <ring> variant below)
            elif name in BWD_FLOP:
                add(name, us, 1, "mfma", BWD_FLOP[name] * BATCH)
            elif name == "adam_fused_kernel":                 # 80 K parameters + their slab sums: a dependent-load chain, not a stream
                add(name + "[all but W_fc1; emits the conv planes]", us, 1, "latency", ADAM_BYTES * (net.n_params - 1600 * 512))
            else:
                add(name, us, 1, "latency", 0)
        us = ev_time(lambda: L.check(lib.fb_replay_profile_gather(replay.h, BATCH, L.ptr(idx), L.ptr(s), L.ptr(s2), L.ptr(a), L.ptr(r),
                                                                  L.ptr(t), R, st()), "gather"), R)
        add("gather_kernel<false>[B=32]", us, 0, "hbm", GATHER_BYTES * BATCH)      # (no loop launches it any more; callers that want the u8 minibatch)
        # what the full loop launches instead of gather + conv1 + conv2/3: the conv trunk of every sampled state, fed from the 1-bit ring
        for k, nm, flop in ((1, "conv23_t_kernel<ring>[train 2B=64: conv1 + pool + conv2 + conv3 from the frame ring]",
                             (FWD_FLOP["conv1_pool_kernel"] + FWD_FLOP["conv23_t_kernel"]) * 2 * BATCH),
                            (7, "conv_bw_kernel<ring>[per-sample conv3^T / conv2^T chain + conv3 / conv2 / conv1 dW; Adam of W_fc1 rides (22.9 MB)]", BWD_FLOP["conv_bw_kernel"] * BATCH)):
            us = ev_time(lambda: L.check(lib.fb_profile_ring_kernel(replay.h, scratch.h, k, R, 0, BATCH, L.ptr(idx), L.ptr(a), L.ptr(r),
                                                                    L.ptr(t), L.ptr(loss), st()), "profile ring"), R)
            c1, c23 = FWD_FLOP["conv1_pool_kernel"], FWD_FLOP["conv23_t_kernel"]
            add(nm, us, 1, "mfma", flop, (2 * c1 + 3 * c23) / (c1 + c23) if k == 1 else 0)
            if k == 1:
                kernels[-1]["dtype"] = "f16x2 (conv1) / f16x3 (conv2, conv3), fp32 result"
        mid = torch.randint(0, 100000, (256,), dtype=torch.int64, device="cuda")
        midrep = [torch.empty((256, 80, 80, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        mm = [torch.empty(256, dtype=dt_, device="cuda") for dt_ in (torch.uint8, torch.float32, torch.uint8)]
        us = ev_time(lambda: L.check(lib.fb_replay_profile_gather(replay.h, 256, L.ptr(mid), L.ptr(midrep[0]), L.ptr(midrep[1]),
                                                                  L.ptr(mm[0]), L.ptr(mm[1]), L.ptr(mm[2]), R, st()), "gather"), R)
        add("gather_kernel<false>[B=256]", us, 0, "hbm", GATHER_BYTES * 256)
        big = torch.randint(0, 100000, (4096,), dtype=torch.int64, device="cuda")
        bigrep = [torch.empty((4096, 80, 80, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        bm = [torch.empty(4096, dtype=dt_, device="cuda") for dt_ in (torch.uint8, torch.float32, torch.uint8)]
        us = ev_time(lambda: L.check(lib.fb_replay_profile_gather(replay.h, 4096, L.ptr(big), L.ptr(bigrep[0]), L.ptr(bigrep[1]),
                                                                  L.ptr(bm[0]), L.ptr(bm[1]), L.ptr(bm[2]), 10, st()), "gather"), 10)
        add("gather_kernel<false>[B=4096]", us, 0, "hbm", GATHER_BYTES * 4096)
        us = ev_time(lambda: [replay.current_state() for _ in range(R)], R)
        add("gather_kernel<true>[currentState n=1024]", us, 0, "hbm", 2 * 25_600 * N_ENVS)
        us = ev_time(lambda: [env.frame_step(acts, want_u8=False) for _ in range(R)], R)
        add("env_kernel<true>[n=1024]", us, 1, "lds", ENV_BYTES * N_ENVS)      # row-mask table lookups in LDS; 6.4 KB per env-step is no HBM load
        # SURVEY 8(d) asks for the env step's LDS-bandwidth fraction.  LDS bytes per env-step, counted statically from csrc/fb_env.hip (no
        # counter reports LDS bytes): a workgroup stages 29.6 KB of tables once, then per env moves ~17 KB (rows' table reads 3.6, frame
        # assembly 4.6, nibble / ring / frame reads of the assembled words 8.0, collision 0.8).  One env per workgroup up to 2048 envs.
        lds_peak = 256 * 128 * 2.4e9 / 1e12                  # TB/s: 256 CUs x 128 B / clk x 2.4 GHz
        def lds_frac(n, rate):                               # rate = env-steps/s alone at n envs
            per_wg = max(1, -(-n // 2048))
            b = 17_000 + 29_600 / per_wg
            return {"n_envs": n, "lds_bytes_per_env_step": round(b), "achieved_TBps": round(b * rate / 1e12, 2), "peak_TBps": round(lds_peak, 1),
                    "frac": round(b * rate / 1e12 / lds_peak, 4)}
        kernels[-1]["lds"] = [lds_frac(int(n), r) for n, r in env_by_n.items()]
        kernels[-1]["lds_note"] = "static byte count, not a counter; at 6-11 % of the LDS peak the step is latency-bound (dependent chain per env), not LDS-bandwidth-bound"
        # prioritized replay (config 4: 1 M-slot SumTree in HBM): latency-bound tree walks -> us per batch, no BW fraction
        per = VecReplay(CAPACITY, N_ENVS, prioritized=True)
        per.seed(seed, "numpy")
        per.reset(env.frame_bits)
        for _ in range(64):                                  # 65 536 stored transitions, priorities all at the store maximum
            per.push(env.frame_bits, acts, env.reward, env.terminal)
        pidx, _ = per.sample(BATCH)
        perr = torch.rand(BATCH, device="cuda")
        per.update_priorities(pidx, abs_err=perr)
        us = ev_time(lambda: [per.sample(BATCH) for _ in range(R)], R)
        add("per_sample_kernel[B=32, 1M-slot tree]", us, 0, "latency", 0)
        us = ev_time(lambda: [per.update_priorities(pidx, abs_err=perr) for _ in range(R)], R)
        add("per_update_kernel[B=32, 1M-slot tree]", us, 0, "latency", 0)
        us = ev_time(lambda: [per.push(env.frame_bits, acts, env.reward, env.terminal) for _ in range(R)], R)
        add("push_kernel + per_store_kernel[n=1024, exact order]", us, 0, "latency", 0)
        per.set_per_mode("fast")                             # level-wise recomputation instead of ordered running sums
        us = ev_time(lambda: [per.push(env.frame_bits, acts, env.reward, env.terminal) for _ in range(R)], R)
        add("push_kernel + per_store_fast_kernel[n=1024]", us, 0, "latency", 0)
        us = ev_time(lambda: [per.update_priorities(pidx, abs_err=perr) for _ in range(R)], R)
        add("per_update_fast_kernel[B=32, 1M-slot tree]", us, 0, "latency", 0)
        del per
        dom = max(kernels, key=lambda k: k["us"] * k["launches_per_step"])
        roofline = {k: dom[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic")}
        roofline["us"] = dom["us"]
        for f in ("dtype", "frac_basis", "achieved_issued", "frac_issued", "note"):
            if f in dom:
                roofline[f] = dom[f]
        # the north-star HBM figure: the replay gather (102 417 logical bytes per sampled transition).  Per batch size the
        # PMC figure comes FIRST (HBM bytes the counters saw, profiles/traffic.json, / time / 8 TB/s), the logical figure second (SURVEY's
        # s and s' as 2 x 25 600 B read + written per transition; the ring stores a frame as 800 B of bits and shares frames between s and
        # s', so the launch really moves far less).  B = 32 is 1 us of traffic inside a ~4 us dependent launch: latency-bound, no bandwidth
        # claim.  The measured loops launch no gather at all: the train step reads the sampled transitions' bits in the ring directly.
        gk = {k["kernel"]: k for k in kernels}
        roofline["replay_gather"] = {"in_loop": "none: the conv trunk kernel reads 4 x 800 B of frame bits per state from the ring (conv23_t_kernel<ring>), "
                                                "in fb_vec_step and in fb_train_steps alike"}
        for b, n in (("B=32", "gather_kernel<false>[B=32]"), ("B=256", "gather_kernel<false>[B=256]"), ("B=4096", "gather_kernel<false>[B=4096]")):
            kk = gk[n]
            g_ = {"us": kk["us"], "traffic": kk["traffic"]}
            if kk["traffic"]:
                g_["pmc_GBps"] = round(kk["traffic"] / kk["us"] / 1e3, 1)
                g_["pmc_frac"] = round(kk["traffic"] / kk["us"] / 1e3 / HBM_PEAK_GBS, 4)
            g_["footnote_logical_GBps"] = kk["logical_GBps"]          # (not a roofline figure: prices bytes the launch does not move)
            g_["footnote_logical_basis"] = "SURVEY 8d: 102 417 B per sampled transition"
            if b == "B=32":
                g_["bound"] = "latency (1 us of traffic in a dependent launch; an EMPTY launch costs 2.8-3.3 us here, tools/mb/mb_launch.hip): the fractions are reported, not claimed"
            roofline["replay_gather"][b] = g_
        roofline["grad_steps_per_sec"] = round(grad_steps_per_s, 1)      # (the train-only half of BASELINE.json's metric, where the driver's record keeps it)
        # `us` above is the kernel ALONE on its launch stream (HIP events).  In the measured loop it runs on the net's side stream beside
        # the train chain (split schedule) and both slow each other: the committed rocprofv3 --kernel-trace --stats summary of this command
        # (tools/profile_round.sh) averages its in-loop and its isolated launches -- reported beside the isolated figure, same flop count
        spath = os.path.join(ROOT, "profiles", "r04_kernel_stats.csv")
        if os.path.exists(spath) and dom["kernel"].startswith("conv23_sp_kernel<C1>"):
            import csv
            for row in csv.DictReader(open(spath)):
                if "conv23_sp_kernel<3, 5, true>" in row["Name"]:
                    avg = float(row["AverageNs"]) / 1e3
                    roofline["rocprof_summary"] = {"file": "profiles/r04_kernel_stats.csv", "calls": int(row["Calls"]), "avg_us": round(avg, 3),
                                                   "frac": round(dom["frac"] * dom["us"] / avg, 5),
                                                   "note": "in-loop launches (beside the train chain, ~33 us) and isolated ones (~29 us) averaged"}
                    break

    # ---------------------------------------------------------------- CPU baseline (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc                    # the checker, timed as the reported baseline
        res = orc.reference_loop(observe_steps=300, train_steps=60, replay_cap=50000, seed=0)
        cpu = {"value": round(res["env_steps_per_s"], 2), "unit": "env-steps/s", "cores": 1, "kind": "port",
               "grad_steps_per_sec": round(res["grad_steps_per_s"], 3),
               "sample": f"oracle single-env loop of FlappyBirdDQN.py:72-76 (batch-1 act, full render + preprocess, "
                         f"store, random.sample(32), train): {res['env_steps']} env steps incl. {res['grad_steps']} "
                         f"train steps in {res['seconds']:.1f} s, fp64-accumulating scalar C, 1 thread, no 30 FPS cap"}
        # Memory.sample restated faithfully: get_min_prob (a min over all filled leaves) inside the per-sample loop
        # (BrainPrioritizedReplyDQN.py:70-71,141), capacity 50 000, full memory
        mem = orc.Memory(50000)
        mem.store(50000)
        t0 = time.perf_counter()
        for _ in range(5):
            mem.sample(32, u=np.full(32, 0.5))
        cpu["per_sample_ms_faithful"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
        # (ii) all host cores: the reference has ONE env and ONE session, so "all cores" = that many independent
        # replicas of the same loop (one per core, no shared state); the 30 FPS cap of the reference (tick(30),
        # wrapped_flappy_bird.py:179) bounds each replica at 30 env-steps/s and 30 grad-steps/s analytically.
        import concurrent.futures as cf
        ncpu = min(len(os.sched_getaffinity(0)), 16)         # the box's CPU share for one GPU is 16 cores
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(ncpu) as ex:              # ctypes releases the GIL inside the C loop
            rs = list(ex.map(lambda i: orc.reference_loop(observe_steps=100, train_steps=20, replay_cap=50000, seed=i), range(ncpu)))
        wall = time.perf_counter() - t0
        cpu["all_cores"] = {"cores": ncpu, "value": round(sum(r["env_steps"] for r in rs) / wall, 2), "unit": "env-steps/s",
                            "grad_steps_per_sec": round(sum(r["grad_steps"] for r in rs) / wall, 3),
                            "sample": f"{ncpu} independent replicas (100 observe + 20 train steps each) in {wall:.1f} s",
                            "fps_capped_bound": {"env_steps_per_s": 30 * ncpu, "grad_steps_per_s": 30 * ncpu}}

    if rank == 0:
        out = {
            "metric": "env steps/sec (whole node) + DQN grad-steps/sec, 80x80x4 batch=32",
            "value": round(env_steps_per_s, 1), "unit": "env-steps/s",
            "grad_steps_per_sec": round(grad_steps_per_s, 1),
            "grad_steps_per_sec_eager": round(grad_steps_eager, 1),
            "env_only_steps_per_sec": round(env_only, 1), "env_only_steps_per_sec_by_n_envs": env_by_n,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # every leg: the K-step measurement of the contract repeated until >= 200 ms have been timed; value / ms_per_step /
            # grad_steps_per_sec are the MEDIAN measurement, min and max beside it (ms per step of the full loop)
            "repeats": {"full_loop": len(dts_full), "env_only": len(dts_env), "train_only": len(dts_tr), "train_only_eager": len(dts_tr_eager)},
            "ms_per_step_min_max": [round(min(dts_full) / args.steps * 1e3, 4), round(max(dts_full) / args.steps * 1e3, 4)],
            "timed_seconds": round(sum(dts_full) + sum(dts_env) + sum(dts_tr) + sum(dts_tr_eager), 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 1024 vectorised envs + BrainDQN uniform replay, batch 32, fp32, per GPU",
                       "n_envs_per_gpu": N_ENVS, "batch": BATCH, "replay_slots": CAPACITY, "fc_width": 512,
                       "sampler": "cpython-mt19937 (bit-exact random.sample)", "epsilon": eps,
                       "schedule": {"full_loop": "split: the train chain on the step's stream beside acting + env on the net's side stream, handed over through device words "
                                                 "(fb_vec_step, include/fbdqn.h); results bit-identical to the one-stream order" if split_issued else "one stream",
                                    "steps_split": split_issued, "minibatches_started_beside_their_env_step": split_clean,
                                    "one_stream_ms_per_step": round(one_stream_ms, 4), "one_stream_env_steps_per_s": round(world * N_ENVS / one_stream_ms * 1e3, 1)},
                       "train_leg": "fb_train_steps(10) in one hipGraph: 10 x (random.sample -> ring-fed five-launch train step)" if graph_used else "eager",
                       "train_only": {"grad_steps_per_sec": round(grad_steps_per_s, 1), "us_per_grad_step": round(1e6 * world / grad_steps_per_s, 2),
                                      "grad_steps_per_sec_eager": round(grad_steps_eager, 1)},
                       "env_only_steps_per_sec": round(env_only, 1),
                       "replicas_bit_identical": replicas_identical,       # (N > 1 only: parameters compared across ranks after the timed legs)
                       "other_configs": other,
                       # N > 1 (or FB_BENCH_FORCE_DP=1): which data-parallel path the step took, the collective alone, per-rank step time
                       "dp_path": dp_path, "allreduce_us": allreduce_us, "rank_ms_per_step_min_max": rank_ms,
                       "parallelism": (f"dp{world}: envs + replay sharded per rank, one RCCL all-reduce of the flat gradient per step"
                                       + (" (fb_vec_step_dp: issued from the C side in two pieces, the W_fc1 / head part overlapped with the conv backward)"
                                          if native is not None else " (torch.distributed)")) if world > 1 else "single GPU"},
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if native is not None:
        torch.cuda.synchronize()
        native.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
