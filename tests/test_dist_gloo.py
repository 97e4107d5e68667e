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


def test_single_process_is_a_no_op():
    from dqnflappybird_amd import dist as fdist
    g = torch.ones(10)
    assert fdist.allreduce_gradients(g, True) is g and torch.equal(g, torch.ones(10))
    with pytest.raises(ValueError):
        fdist.shard_envs(10, 0, 3)
