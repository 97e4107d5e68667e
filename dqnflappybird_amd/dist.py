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


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def broadcast_params(flat_params, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src=src)
    return flat_params


class OverlappedAllReduce:
    """allreduce_gradients(flat_grad, mean_loss) for the HIP path, with the large part off the critical path.

    A gradient-exporting step records an event right behind its fc1 backward launch (fb_qnet_set_grad_event): from there on
    flat_grad[split:] -- W_fc1, b_fc1 and the head, 3.3 of the 3.59 MB -- is final while three more launches (the conv backward and
    the slab reduction) still run.  Calling the object after the step has been ISSUED (the host runs ahead of the GPU) therefore
        side stream:  wait(event) -> all-reduce flat_grad[split:]         (overlaps the conv backward)
        step stream:  all-reduce flat_grad[:split] (0.3 MB)  -> wait(side stream)
    and returns with the step's stream ordered behind both; fb_qnet_apply_adam follows as before.  Element for element the same sums as
    one all-reduce of the whole vector, so replicas stay bit-identical.  `force` runs the choreography at world size 1 too (tests,
    tools/time_dp_step.py).

    OPT-IN (FB_DP_OVERLAP=1), not the default: through torch.distributed every collective costs two cross-stream hops (step stream ->
    RCCL's stream -> back), the event and the join add two more, and at world size 1 -- where RCCL's all-reduce is a copy -- the split
    step measures 27 us SLOWER than the plain one (152 vs 125 us; fused single-GPU step 114).  It can only pay where the 3.3 MB
    all-reduce itself takes longer than that, which no hardware available to this build could show."""

    def __init__(self, net, flat_grad, mean_loss, force=False):
        from . import _lib as L
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.on = dist.is_initialized() and (self.world > 1 or force)
        self.mean = bool(mean_loss) and self.world > 1
        self.net, self.grad = net, flat_grad
        if not self.on:
            return
        lib = L.lib()
        split = int(lib.fb_qnet_grad_split(net.h))
        self.front, self.tail = flat_grad[:split], flat_grad[split:]
        self.side = torch.cuda.Stream()
        self.event = torch.cuda.Event()
        self.event.record()                                  # (creates the underlying hipEvent_t: torch makes events lazily)
        L.check(lib.fb_qnet_set_grad_event(net.h, self.event.cuda_event), "fb_qnet_set_grad_event")

    def __call__(self):
        if not self.on:
            return self.grad
        main = torch.cuda.current_stream()
        self.side.wait_event(self.event)
        with torch.cuda.stream(self.side):
            dist.all_reduce(self.tail, op=dist.ReduceOp.SUM)
            if self.mean:
                self.tail.div_(self.world)
        dist.all_reduce(self.front, op=dist.ReduceOp.SUM)
        if self.mean:
            self.front.div_(self.world)
        main.wait_stream(self.side)
        return self.grad

    def close(self):
        if self.on:
            from . import _lib as L
            L.lib().fb_qnet_set_grad_event(self.net.h, None)
            self.on = False


class NativeUnavailable(RuntimeError):
    """raised by NativeDP() on EVERY rank alike when some rank cannot take the native path: fall back to torch.distributed"""


def native_wanted():
    """The library's own communicator is OPT-IN (FB_DP_NATIVE=1): its multi-rank path (csrc/fb_dist.hip, fb_vec_step_dp) has only ever
    run at world size 1 -- no multi-GPU node has been available to this build -- so torch.distributed's all-reduce, which has, stays the
    default until a 2-rank run has compared replicas bit for bit."""
    return os.environ.get("FB_DP_NATIVE", "0") == "1"


class NativeDP:
    """The library's own RCCL communicator (csrc/fb_dist.hip): the data-parallel step's all-reduce issued from the C side, straight onto
    the step's HIP stream, no detour through torch.distributed's RCCL stream (two cross-stream hops per collective); optionally in two
    pieces with the W_fc1 / head part of the gradient on a side stream behind the fc1 backward launch.  torch.distributed only carries the 128-byte unique id
    from rank 0 to the others (or nothing at world size 1).  handle: fb_dist_t for vec.VecStep(dist=...)."""

    def __init__(self, rank=None, world=None, overlap=None):
        """overlap: False = one all-reduce of the whole gradient on the step's stream (default), True = in two pieces with the large one
        on a side stream behind the fc1 backward launch (default from FB_DP_OVERLAP=1).  See csrc/fb_dist.hip for when which wins.

        COLLECTIVE over the ranks of torch.distributed, in three phases, so that no rank can be left waiting in a collective its peers
        never enter:
          1. rank-local, fallible, non-collective: every rank loads RCCL (fb_dist_probe), rank 0 also draws the unique id;
          2. agreement: all_reduce(MIN) of "phase 1 worked here" -- if any rank failed, EVERY rank raises NativeUnavailable here (the
             caller falls back to torch.distributed's all-reduce, all ranks alike); nothing of RCCL's own has been exchanged yet;
          3. the id is broadcast and every rank enters ncclCommInitRank (fb_dist_create).  From the broadcast on a failure is FATAL
             (FbError): a rank that fell back now would leave its peers blocked inside ncclCommInitRank."""
        from . import _lib as L
        self.handle = None
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        lib = L.lib()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = path.encode() if os.path.exists(path) else None
        uid = torch.zeros(128, dtype=torch.uint8)
        # -- 1: what can fail on one rank alone
        err = None
        try:
            L.check(lib.fb_dist_probe(path), "fb_dist_probe")
            if self.rank == 0:
                L.check(lib.fb_dist_unique_id(path, uid.data_ptr()), "fb_dist_unique_id")
        except Exception as e:                               # noqa: BLE001  (any failure here means "not on this rank")
            err = e
        on_gpu = self.world > 1 and dist.get_backend() == "nccl"
        # -- 2: every rank learns whether every rank got this far
        if not self.agree(err is None, self.world, on_gpu):
            raise NativeUnavailable(f"rank {self.rank}: " + (f"{type(err).__name__}: {err}" if err else "another rank cannot load RCCL"))
        # -- 3: collective set-up; no way back from here
        if self.world > 1:
            t = uid.cuda() if on_gpu else uid
            dist.broadcast(t, src=0)
            uid = t.cpu()
        self.handle = lib.fb_dist_create(path, self.rank, self.world, uid.data_ptr())
        if not self.handle:
            raise L.FbError("fb_dist_create failed after the unique id was exchanged (fatal: the other ranks are inside "
                            "ncclCommInitRank): " + lib.fb_last_error().decode("utf-8", "replace"))
        self.overlap = bool(overlap) if overlap is not None else os.environ.get("FB_DP_OVERLAP", "0") == "1"
        L.check(lib.fb_dist_set_overlap(self.handle, int(self.overlap)), "fb_dist_set_overlap")

    @staticmethod
    def agree(ok_here, world, on_gpu=False):
        """True when `ok_here` holds on EVERY rank (one all_reduce(MIN); world size 1: ok_here itself)."""
        if world <= 1 or not dist.is_initialized():
            return bool(ok_here)
        t = torch.tensor([1 if ok_here else 0], dtype=torch.int32)
        if on_gpu:
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def all_reduce(self, flat):
        """the collective alone (sum, in place) on torch's current stream"""
        from . import _lib as L
        L.check(L.lib().fb_dist_all_reduce(self.handle, L.ptr(flat), flat.numel(), L.current_stream()), "fb_dist_all_reduce")
        return flat

    def close(self):
        if self.handle:
            from . import _lib as L
            L.lib().fb_dist_destroy(self.handle)
            self.handle = None

