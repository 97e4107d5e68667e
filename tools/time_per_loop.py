import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from dqnflappybird_amd.vecbrain import VecBrain
for n in (1024, 2048):
    vb = VecBrain(n, algo="per", capacity=1_000_000, observe=20, seed=1)
    vb.run(60, log_every=0); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); vb.run(200, log_every=0); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 200 * 1e6)
    print(n, "envs:", round(sorted(ts)[3], 1), "us/step")
