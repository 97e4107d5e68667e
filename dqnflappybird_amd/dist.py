"""Data-parallel glue: one process per GPU, one all-reduce of the flat gradient per train step.

Envs and replay shards are rank-local (no data-path collective); parameters
and Adam state are replicated and stay bit-identical because every rank applies the same reduced
gradient.  `torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the
CPU tests (tests/test_dist_gloo.py).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_envs(n_envs_total, rank, world):
    """env id -> rank: contiguous blocks, e // (N / world)."""
    if n_envs_total % world:
        raise ValueError(f"{n_envs_total} envs do not split evenly over {world} ranks")
    per = n_envs_total // world
    return range(rank * per, (rank + 1) * per)


def allreduce_gradients(flat_grad, mean_loss):
    """Sum the flat fp32 gradient over ranks (3.59 MB).  The sum-of-squares loss of BrainDQN
    (BrainDQN.py:162) over the global batch is the plain sum of the rank losses; the mean losses
    (BrainDQNNature.py:119, BrainPrioritizedReplyDQN.py:251) need the sum divided by the world size."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    if mean_loss:
        flat_grad.div_(dist.get_world_size())
    return flat_grad


def broadcast_params(flat_params, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src=src)
    return flat_params
