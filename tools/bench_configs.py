"""One JSON line per BASELINE.json configuration, through the device-resident loop (VecBrain: act -> env step -> store ->
sample -> gather -> train, one train step per env step), so that DESIGN.md's per-config table is reproducible:

    python tools/bench_configs.py [--steps N] [--only i] [--out FILE]
    rocprofv3 --kernel-trace --stats -d gpurun_out/cfg_i -- python3 tools/bench_configs.py --only i     (kernel stats of one row)

Rows: configs[1] (1024 envs, BrainDQN, uniform 1 M-slot replay, B = 32: bench.py's workload, here through VecBrain),
configs[2] (Double-DQN, 4096 envs, B = 256; fp32 and bf16 arithmetic), configs[3] (prioritized replay, 1 M-slot SumTree, 4096
envs; reference-order tree and FB_PER_FAST), configs[4]'s per-GPU share (dueling head, 4096 envs, B = 32), and BASELINE.md's C4
(CPU, oracle): the reference's Memory.sample restated faithfully -- get_min_prob, a min over all filled leaves, inside the
per-sample loop (BrainPrioritizedReplyDQN.py:70-71,141) -- at 50 000 and 1 000 000 slots."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROWS = [
    ("configs[1] BrainDQN, 1024 envs, uniform replay, B=32, fp32", dict(n_envs=1024, algo="dqn", batch=32), None),
    ("configs[2] Double-DQN, 4096 envs, B=256, fp32", dict(n_envs=4096, algo="double", batch=256), None),
    ("configs[2] Double-DQN, 4096 envs, B=256, bf16", dict(n_envs=4096, algo="double", batch=256), "bf16"),
    ("configs[3] PER, 1M-slot SumTree, 4096 envs, B=32, reference-order tree", dict(n_envs=4096, algo="per", batch=32), "exact"),
    ("configs[3] PER, 1M-slot SumTree, 4096 envs, B=32, FB_PER_FAST", dict(n_envs=4096, algo="per", batch=32), "fast"),
    ("configs[4] per-GPU share: dueling, 4096 envs, B=32", dict(n_envs=4096, algo="nature", arch="dueling", batch=32), None),
    ("configs[4] per-GPU share: dueling Double-DQN, 4096 envs, B=32", dict(n_envs=4096, algo="double", arch="dueling", batch=32), None),
    ("configs[4] whole-node env count on ONE GPU: dueling Double-DQN, 32768 envs, B=32", dict(n_envs=32768, algo="double", arch="dueling", batch=32), None),
]


def gpu_row(i, steps, min_total=0.3):
    import torch
    from dqnflappybird_amd.vecbrain import VecBrain
    name, kw, mode = ROWS[i]
    kw = dict(kw)
    n_envs = kw.pop("n_envs")
    vb = VecBrain(n_envs, capacity=1_000_000, observe=20, seed=1, **kw)
    if mode == "bf16":
        vb.set_dtype("bf16")
    if mode in ("exact", "fast"):
        vb.replay.set_per_mode(mode)
    vb.run(40, log_every=0)                      # 20 observe + 20 train steps of warm-up
    torch.cuda.synchronize()
    times = []
    total = 0.0
    while total < min_total and len(times) < 50:       # measurements of `steps` steps until >= min_total s are timed; report the median
        t0 = time.perf_counter()
        vb.run(steps, log_every=0)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        total += times[-1]
    assert torch.isfinite(vb.last_loss).all()
    dt = sorted(times)[len(times) // 2]
    overflow = vb.net.overflow_count()
    assert overflow == 0, f"{overflow} wave(s) met an activation beyond the two-plane fp16 range"
    del vb
    torch.cuda.synchronize()
    return {"config": name, "n_envs": n_envs, "algo": kw["algo"], "arch": kw.get("arch", "plain"), "batch": kw["batch"], "mode": mode or "fp32",
            "replay_slots": 1_000_000, "steps": steps, "repeats": len(times), "us_per_step": round(dt / steps * 1e6, 2),
            "env_steps_per_s": round(n_envs * steps / dt, 1), "grad_steps_per_s": round(steps / dt, 1),
            "gpu": torch.cuda.get_device_name(0), "data": "synthetic"}


def cpu_c4():
    import numpy as np
    from oracle import oracle as orc
    out = []
    for cap in (50_000, 1_000_000):
        mem = orc.Memory(cap)
        mem.store(cap)
        reps = 5 if cap == 50_000 else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            mem.sample(32, u=np.full(32, 0.5))
        out.append({"config": f"BASELINE.md C4: Memory.sample(32), faithful O(size) min inside the loop, {cap} slots (CPU oracle, 1 thread)",
                    "ms_per_sample_batch": round((time.perf_counter() - t0) / reps * 1e3, 3), "kind": "port"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--cpu", action="store_true", help="also time BASELINE.md C4 on the host")
    a = ap.parse_args()
    rows = []
    for i in range(len(ROWS)) if a.only is None else [a.only]:
        steps = a.steps if "reference-order" not in ROWS[i][0] else max(a.steps // 4, 20)
        rows.append(gpu_row(i, steps))
        print(json.dumps(rows[-1]), flush=True)
    if a.cpu:
        for r in cpu_c4():
            rows.append(r)
            print(json.dumps(r), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
