"""Does the vectorised loop LEARN?  

Every kernel of the loop is parity-checked against the oracle; what kernel-level parity cannot see is the plumbing between them in the
vectorised form -- reward / terminal routing through the replay ring, the target-sync timing, the epsilon schedule, the frame stack
across episode ends.  A score curve can.  The reference's own evidence is logs_bird/eval_images/mean.png (mean episode score 60 .. 135 after
2 - 4 M single-env steps, BrainDQNNature.py:149-197); no target score is set here, only "clearly above the untrained policy's".

    python tools/learn_curve.py [--envs 16,1024] [--lrs 1e-6,1e-5] [--steps 2000000] [--window 50000] [--algo nature] [--out FILE]

One VecBrain run per (envs, lr); one train step per loop step once onlineTimeStep > OBSERVE, as in the reference.  Every `window`
steps the device stats buffer (episodes ended, score sum, score max, pipes passed: kept by the env kernel) is read and zeroed, so each
row is the window's own figure, not a running average.  Rows go to stdout and to --out as they are produced.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from dqnflappybird_amd.vecbrain import VecBrain  # noqa: E402


def run(n_envs, lr, steps, window, algo, arch, seed, out, budget_s, explore):
    vb = VecBrain(n_envs, algo=algo, arch=arch, capacity=1_000_000, seed=seed, explore=explore)
    vb.net.set_hparams(lr=lr)
    head = f"# envs {n_envs}  algo {algo}/{arch}  lr {lr:g}  batch {vb.batch}  observe {vb.observe}  explore {vb.explore}  eps {vb.initial_epsilon} -> {vb.final_epsilon}  target sync / {vb.replace_target_iter}"
    cols = "#   train_steps   env_steps  epsilon  episodes  mean_score  max_score  pipes/episode      loss   steps/s"
    for f in (sys.stdout, out):
        print(head, file=f); print(cols, file=f); f.flush()
    t_run = time.perf_counter()
    done = 0
    rows = []
    while done < steps:
        n = min(window, steps - done)
        t0 = time.perf_counter()
        vb.run(n, log_every=0)
        ep, ssum, smax, pipes = vb.stats.tolist()            # (the host sync of the window)
        dt = time.perf_counter() - t0
        vb.stats.zero_()
        done += n
        loss = vb.last_loss.item() if vb.last_loss is not None else float("nan")
        trained = max(0, vb.onlineTimeStep - vb.observe - 1)
        row = (trained, done * n_envs, vb.epsilon, ep, ssum / max(ep, 1), smax, pipes / max(ep, 1), loss, n / dt)
        rows.append(row)
        line = f"  {row[0]:13d} {row[1]:11d}  {row[2]:7.5f} {row[3]:9d}  {row[4]:10.3f} {row[5]:10d}  {row[6]:13.3f} {row[7]:9.4g} {row[8]:9.0f}"
        for f in (sys.stdout, out):
            print(line, file=f); f.flush()
        if budget_s and time.perf_counter() - t_run > budget_s:
            for f in (sys.stdout, out):
                print(f"# stopped at the time budget of {budget_s} s", file=f); f.flush()
            break
    assert vb.env.error_count() == 0
    issued, clean = vb.net.split_stats() if hasattr(vb.net, "split_stats") else (0, 0)      # (raises if a wait between the two streams of the split schedule gave up)
    first = rows[0][4]
    best = max(r[4] for r in rows)
    last = sum(r[4] for r in rows[-3:]) / len(rows[-3:])
    for f in (sys.stdout, out):
        print(f"# summary envs {n_envs} lr {lr:g}: mean score first window {first:.3f}, best window {best:.3f}, last three windows {last:.3f}, max score of the run {max(r[5] for r in rows)}; split schedule: {issued} steps, {clean} minibatches beside their env step\n", file=f)
        f.flush()
    del vb
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", default="16,1024")
    ap.add_argument("--lrs", default="1e-6,1e-5")
    ap.add_argument("--steps", type=int, default=2_000_000)
    ap.add_argument("--window", type=int, default=50_000)
    ap.add_argument("--algo", default="nature")
    ap.add_argument("--arch", default="plain")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--explore", type=int, default=1_000_000)
    ap.add_argument("--budget-s", type=float, default=0.0, help="stop a run after this many seconds (0 = run all its steps)")
    ap.add_argument("--out", default="gpurun_out/learning.txt")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "a") as out:
        print(f"# tools/learn_curve.py on {torch.cuda.get_device_name(0)}: {' '.join(sys.argv[1:])}", file=out)
        for n_envs in [int(x) for x in a.envs.split(",")]:
            for lr in [float(x) for x in a.lrs.split(",")]:
                run(n_envs, lr, a.steps, a.window, a.algo, a.arch, a.seed, out, a.budget_s, a.explore)


if __name__ == "__main__":
    main()
