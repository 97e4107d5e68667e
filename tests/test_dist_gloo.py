"""The N > 1 path on CPU: world_size-2 `gloo` process group, rank-local batches, one all-reduce of
the flat gradient, identical replicas afterwards (dqnflappybird_amd/dist.py).  Compute = oracle."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _batch(rng, B):
    s = (rng.random((B, 80, 80, 4)) < 0.37).astype(np.uint8) * 255
    s2 = (rng.random((B, 80, 80, 4)) < 0.37).astype(np.uint8) * 255
    a = rng.integers(0, 2, B).astype(np.uint8)
    r = rng.choice(np.array([0.1, 3, -3], np.float32), B)
    return s, a, r, s2, (r == -3).astype(np.uint8)


def _grads(o, cfg, p, kind, batch):
    s, a, r, s2, t = batch
    q, acts = o.forward(p, cfg, s, keep=True)
    qn = o.forward(p, cfg, s2).max(1)
    _, loss, _, dq = o.dqn_loss(kind, q, qn, a, r, t)
    return o.backward(p, cfg, s, acts, dq)


def _worker(rank, world, port, mean_loss, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dqnflappybird_amd import dist as fdist
    from oracle import oracle as o
    r, lr, w = fdist.init("gloo")
    assert (r, w) == (rank, world)
    assert list(fdist.shard_envs(32768, rank, world)) == list(range(rank * 16384, (rank + 1) * 16384))
    cfg = o.qcfg()
    p = o.init_params(cfg, seed=0) * 3.0                         # identical replicas
    pt = torch.from_numpy(p.copy() if rank == 0 else np.zeros_like(p))
    fdist.broadcast_params(pt, src=0)
    assert np.array_equal(pt.numpy(), p)
    B = 4
    batch = _batch(np.random.default_rng(100 + rank), B)         # rank-local replay shard
    g = torch.from_numpy(_grads(o, cfg, p, 1 if mean_loss else 0, batch))
    fdist.allreduce_gradients(g, mean_loss)
    opt = o.Adam(p.size, lr=1e-4)
    opt.step(p, g.numpy())
    np.save(os.path.join(out_dir, f"p{rank}.npy"), p)
    np.save(os.path.join(out_dir, f"g{rank}.npy"), g.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mean_loss", [False, True])
def test_two_rank_allreduce_equals_global_batch(tmp_path, mean_loss):
    from oracle import oracle as o
    port = 29700 + int(mean_loss) + (os.getpid() % 200)
    mp.spawn(_worker, args=(2, port, mean_loss, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    g0 = np.load(tmp_path / "g0.npy")
    assert np.array_equal(p0, p1)                                # replicas stay bit-identical
    # single-process reference on the concatenated (global) batch
    cfg = o.qcfg()
    p = o.init_params(cfg, seed=0) * 3.0
    b0, b1 = _batch(np.random.default_rng(100), 4), _batch(np.random.default_rng(101), 4)
    glob = tuple(np.concatenate([x, y]) for x, y in zip(b0, b1))
    gref = _grads(o, cfg, p, 1 if mean_loss else 0, glob)        # sum loss: sum; mean loss: mean over 2B
    scale = np.abs(gref).max()
    np.testing.assert_allclose(g0, gref, rtol=1e-4, atol=1e-6 * scale)


# ------------------------------------------------------------------ the real VecBrain rank logic, two processes
VB = dict(n_envs=2, batch=4, capacity=64, fc_width=128, seed=5, observe=2, explore=1000, initial_epsilon=0.25,
          final_epsilon=0.0, replace_target_iter=3)
VB_STEPS = 9                                   # observe 3 steps, train 6; timeStep % 3 == 0 syncs the target net twice


def _vecbrain_worker(rank, world, port, algo, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dqnflappybird_amd import dist as fdist
    from dqnflappybird_amd.vecbrain import VecBrain
    from tests.cpu_backend import CpuVecBackend
    r, _, w = fdist.init("gloo")
    vb = VecBrain(algo=algo, rank=r, world=w, backend=CpuVecBackend(), **VB)
    trace = []
    for _ in range(VB_STEPS):
        vb.step()
        trace.append(np.concatenate([vb.one_step.actions, vb.one_step.idx if vb.one_step.idx is not None else []]))
    np.save(os.path.join(out_dir, f"p{rank}.npy"), vb.net.p[0])
    np.save(os.path.join(out_dir, f"t{rank}.npy"), vb.net.p[1])
    np.save(os.path.join(out_dir, f"m{rank}.npy"), vb.net.opt.m)
    np.save(os.path.join(out_dir, f"trace{rank}.npy"), np.concatenate(trace))
    np.save(os.path.join(out_dir, f"meta{rank}.npy"), np.array([vb.timeStep, vb.onlineTimeStep, vb.net.syncs, vb.epsilon]))
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["dqn", "nature"])
def test_vecbrain_rank_logic_two_processes(tmp_path, algo):
    """VecBrain(rank, world = 2) through the CPU stand-in backend under gloo: replicas stay bit-identical, the two ranks play
    DIFFERENT games and draw DIFFERENT minibatches (env / replay / acting seeds are offset by rank), the update equals one
    optimizer step on the global batch (sum for BrainDQN's sum loss, mean for the mean losses), and the target net syncs on
    timeStep % replace_target_iter while training."""
    import torch
    from dqnflappybird_amd.vecbrain import MEAN_LOSS, VecBrain
    from tests.cpu_backend import CpuVecBackend
    port = 29900 + (os.getpid() % 97) + (0 if algo == "dqn" else 100)
    mp.spawn(_vecbrain_worker, args=(2, port, algo, str(tmp_path)), nprocs=2, join=True)
    ld = lambda n: [np.load(tmp_path / f"{n}{r}.npy") for r in (0, 1)]
    p, t, m, trace, meta = ld("p"), ld("t"), ld("m"), ld("trace"), ld("meta")
    assert np.array_equal(p[0], p[1]) and np.array_equal(t[0], t[1]) and np.array_equal(m[0], m[1])
    assert not np.array_equal(trace[0], trace[1])                # rank-local envs and replay shards
    assert np.array_equal(meta[0], meta[1]) and meta[0][0] == VB_STEPS
    want_syncs = sum(1 for ts in range(VB_STEPS) if ts > VB["observe"] and algo == "nature" and ts % VB["replace_target_iter"] == 0)
    assert meta[0][2] == want_syncs
    # the same thing in ONE process: both ranks' brains side by side, gradients combined by hand
    brains = [VecBrain(algo=algo, rank=r, world=1, backend=CpuVecBackend(), **VB) for r in (0, 1)]
    for r, b in enumerate(brains):                               # what world = 2 changes in the constructor, by hand
        b.rank = r
        b.env = CpuVecBackend().env(VB["n_envs"], VB["seed"] + 1000003 * r)
        b.replay = CpuVecBackend().replay(VB["capacity"], VB["n_envs"], False)
        b.replay.seed(VB["seed"] + r)
        b.env.observe(); b.replay.reset(b.env.frame_bits)
        b.grad = torch.zeros(b.net.n_params)
        b.one_step = CpuVecBackend().step(b.env, b.replay, b.net, VB["batch"], algo, b.gamma, b.grad)
    for step in range(VB_STEPS):
        training = step > VB["observe"]
        for b in brains:
            if training and algo == "nature" and b.timeStep % VB["replace_target_iter"] == 0:
                b.net.sync_target()
            b.one_step(b.epsilon, seed=VB["seed"] + b.rank, step=b.timeStep, train=training)
        if training:
            g = brains[0].grad + brains[1].grad
            if MEAN_LOSS[algo]:
                g = g / 2
            for b in brains:
                b.net.apply_adam(g)
        for b in brains:
            if b.epsilon > b.final_epsilon and b.onlineTimeStep > b.observe:
                b.epsilon -= (b.initial_epsilon - b.final_epsilon) / b.explore
            b.timeStep += 1; b.onlineTimeStep += 1
    assert np.array_equal(brains[0].net.p[0], p[0]) and np.array_equal(brains[1].net.p[0], p[0])
    assert np.array_equal(brains[0].net.p[1], t[0])


def _save_load_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dqnflappybird_amd import dist as fdist
    from dqnflappybird_amd.vecbrain import VecBrain
    from tests.cpu_backend import CpuVecBackend
    r, _, w = fdist.init("gloo")
    ck = os.path.join(out_dir, "ck")

    def run(vb, n):
        tr = []
        for _ in range(n):
            vb.step()
            tr.append(np.concatenate([vb.one_step.actions, vb.one_step.idx if vb.one_step.idx is not None else []]))
        return np.concatenate(tr)

    a = VecBrain(algo="nature", rank=r, world=w, backend=CpuVecBackend(), **VB)
    run(a, 6)                                              # observe 3 steps, train 3
    a.save(ck)                                             # (returns behind a barrier: every file is complete on every rank)
    assert os.path.exists(ck + ".npz") and os.path.exists(f"{ck}.rank0.npz") and os.path.exists(f"{ck}.rank1.npz")
    want = run(a, 4)
    b = VecBrain(algo="nature", rank=r, world=w, backend=CpuVecBackend(), **dict(VB, seed=VB["seed"] + 50))      # other games, weights, generator
    b.load(ck)
    got = run(b, 4)
    same = (np.array_equal(want, got) and np.array_equal(a.net.p[0], b.net.p[0]) and np.array_equal(a.net.p[1], b.net.p[1])
            and np.array_equal(a.net.opt.m, b.net.opt.m) and np.array_equal(a.net.opt.v, b.net.opt.v)
            and (a.timeStep, a.onlineTimeStep, a.epsilon) == (b.timeStep, b.onlineTimeStep, b.epsilon)
            and np.array_equal(a.env.get_state(), b.env.get_state()))
    np.save(os.path.join(out_dir, f"sl{rank}.npy"), np.array([int(same)]))
    np.save(os.path.join(out_dir, f"sltrace{rank}.npy"), got)
    np.save(os.path.join(out_dir, f"slp{rank}.npy"), b.net.p[0])
    dist.destroy_process_group()


def test_vecbrain_save_load_two_ranks(tmp_path):
    """VecBrain.save / load at world size 2: rank 0 alone writes the replicated nets + Adam to `path`, every
    rank its own envs / frame stacks / replay shard to `<path>.rank<r>.npz`; a fresh VecBrain per rank (other seed) that loads the
    checkpoint continues with the same actions, sampled indices, parameters and Adam slots as the brain that saved -- on BOTH ranks,
    whose games and minibatches differ from each other."""
    port = 29500 + (os.getpid() % 180)
    mp.spawn(_save_load_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ok = [int(np.load(tmp_path / f"sl{r}.npy")[0]) for r in (0, 1)]
    assert ok == [1, 1]
    t0, t1 = np.load(tmp_path / "sltrace0.npy"), np.load(tmp_path / "sltrace1.npy")
    assert not np.array_equal(t0, t1)                            # rank-local games / shards really were restored per rank
    assert np.array_equal(np.load(tmp_path / "slp0.npy"), np.load(tmp_path / "slp1.npy"))      # replicas still identical
    l0, l1 = np.load(tmp_path / "ck.rank0.npz"), np.load(tmp_path / "ck.rank1.npz")
    assert not np.array_equal(l0["env_state"], l1["env_state"]) and "online" not in l0.files and "env_state" not in np.load(tmp_path / "ck.npz").files


def test_single_process_is_a_no_op():
    from dqnflappybird_amd import dist as fdist
    g = torch.ones(10)
    assert fdist.allreduce_gradients(g, True) is g and torch.equal(g, torch.ones(10))
    with pytest.raises(ValueError):
        fdist.shard_envs(10, 0, 3)


# ------------------------------------------------------------------ NativeDP: the failure path of the agreement protocol
def _native_worker(rank, world, port, fail_rank, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from dqnflappybird_amd import _lib as L
    from dqnflappybird_amd import dist as fdist
    fdist.init("gloo")
    lib = L.lib()                                        # (the shared object loads without a GPU; nothing is launched here)
    if rank == fail_rank:                                # this rank "cannot load RCCL"
        class _Lib:                                      # the real library with one entry point failing
            def __getattr__(self, name):
                if name == "fb_dist_probe":
                    return lambda path: -1
                return getattr(lib, name)
        L._lib = _Lib()
    try:
        fdist.NativeDP(rank, world)
        outcome = "created"
    except fdist.NativeUnavailable:
        outcome = "unavailable"
    except Exception as e:                               # noqa: BLE001
        outcome = type(e).__name__
    # every rank is still in step: a later collective of the job matches up (a rank stuck in / skipped past a broadcast would
    # deadlock or mismatch here; mp.spawn's join would never return)
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    with open(os.path.join(out_dir, f"o{rank}.txt"), "w") as f:
        f.write(f"{outcome} {t.item()}")
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [0, 1])
def test_native_dp_one_rank_failing_makes_every_rank_fall_back(tmp_path, fail_rank):
    """ADVICE (round 2): NativeDP used to do rank-local fallible work between collectives -- a rank that failed before the id broadcast
    went on to the agreement all-reduce while its peers sat in the broadcast (mismatched collectives: a hang).  Now: local work first,
    one all_reduce(MIN), and only then the id exchange; with one rank unable to load RCCL EVERY rank raises NativeUnavailable before
    anything of RCCL's has been exchanged, and the job's next collective still matches up."""
    port = 29300 + (os.getpid() % 150) + 200 * fail_rank
    mp.spawn(_native_worker, args=(2, port, fail_rank, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        outcome, total = (tmp_path / f"o{r}.txt").read_text().split()
        assert outcome == "unavailable", (r, outcome)
        assert float(total) == 3.0


def test_native_is_opt_in(monkeypatch):
    from dqnflappybird_amd import dist as fdist
    monkeypatch.delenv("FB_DP_NATIVE", raising=False)
    assert not fdist.native_wanted()
    monkeypatch.setenv("FB_DP_NATIVE", "1")
    assert fdist.native_wanted()
    assert fdist.NativeDP.agree(True, 1) and not fdist.NativeDP.agree(False, 1)
