"""VecBrain: the reference's training loop (FlappyBirdDQN.py:72-76 + Brain*.setPerception) for N
envs per GPU, entirely device resident: no tensor leaves HBM between getAction and the Adam update.

    step:  currentState (A1) -> getAction for N envs (A2) -> frame_step (E*, P1 fused) -> store (R1/R4)
           -> sample (R1/R4) -> gather (R2) -> _trainQNetwork (Q*) [-> all-reduce over ranks]

Host-side schedule follows the reference: training starts once onlineTimeStep > OBSERVE, epsilon
decays by (INITIAL - FINAL) / EXPLORE per step after that, Nature/Double sync the target net when
timeStep % 500 == 0, PER never does.
"""
from . import dist as fdist

MEAN_LOSS = {"dqn": False, "nature": True, "double": True, "per": True}


class HipVecBackend:
    """What VecBrain computes with: the HIP library through vec.py.  (tests/cpu_backend.py holds a CPU stand-in with the
    same five factories, so that the rank logic below -- seed offsets, gradient averaging, target-sync placement -- runs
    in the world-size-2 gloo tests without a GPU; the product never uses anything else than this class.)"""
    name = "hip-gfx950"
    per_one_step = True                                      # fb_vec_step runs the prioritized step too (store, Memory.sample, train, batch_update)

    def env(self, n_envs, seed):
        from .vec import VecGameState
        return VecGameState(n_envs, seed=seed)

    def replay(self, capacity, n_envs, prioritized):
        from .vec import VecReplay
        return VecReplay(capacity, n_envs, prioritized=prioritized)

    def net(self, actions, fc_width, arch, max_batch):
        from .vec import QNet
        return QNet(actions, fc_width, arch, max_batch=max_batch)

    def step(self, env, replay, net, batch, algo, gamma, flat_grad, dist=None, mean_loss=False):
        from .vec import VecStep
        return VecStep(env, replay, net, batch, algo, gamma, flat_grad=flat_grad, dist=dist, mean_loss=mean_loss)

    def native(self, rank, world):
        """the library's own RCCL communicator (dist.NativeDP): with it the whole data-parallel step -- all-reduce and Adam included --
        is the one host call fb_vec_step_dp.  Opt-in (FB_DP_NATIVE=1, dist.native_wanted); otherwise torch.distributed's all-reduce
        sits between the step and Adam.  NativeDP() itself makes the ranks agree: it either succeeds on every rank or raises
        NativeUnavailable on every rank (fallback, all alike); a failure after its id exchange is fatal and propagates."""
        import torch.distributed as tdist
        if not fdist.native_wanted() or not tdist.is_initialized() or tdist.get_backend() != "nccl":
            return None
        try:
            return fdist.NativeDP(rank, world)
        except fdist.NativeUnavailable:
            return None

    def zeros(self, n):
        import torch
        return torch.zeros(n, dtype=torch.float32, device="cuda")

    def to_device(self, x):
        import numpy as np
        import torch
        return torch.from_numpy(np.ascontiguousarray(x)).cuda()

    def synchronize(self):
        import torch
        torch.cuda.synchronize()

    def train_from_replay(self, replay, net, algo, idx, isw, gamma, flat_grad, want_abs_err):
        """minibatch assembly + train step from the sampled indices on, without gathered copies -> (loss, abs_err or None)"""
        from .vec import train_from_replay
        out = train_from_replay(replay, net, algo, idx, gamma, flat_grad, isw=isw, want_abs_err=want_abs_err)
        return out[0], (out[4] if want_abs_err else None)

    def reducer(self, net, flat_grad, mean_loss):
        """-> callable that all-reduces flat_grad over the ranks: one collective; with FB_DP_OVERLAP=1 two, the large one overlapped
        with the conv backward (dist.OverlappedAllReduce -- off by default, see its docstring)"""
        import os
        if os.environ.get("FB_DP_OVERLAP", "0") == "1":
            return fdist.OverlappedAllReduce(net, flat_grad, mean_loss)
        return lambda: fdist.allreduce_gradients(flat_grad, mean_loss)


class VecBrain:
    def __init__(self, n_envs, algo="dqn", arch="plain", batch=32, capacity=1_000_000, fc_width=512, seed=0,
                 observe=1000, explore=1_000_000, initial_epsilon=0.03, final_epsilon=0.0, gamma=0.99,
                 replace_target_iter=500, sampler=None, rank=0, world=1, backend=None):
        be = backend or HipVecBackend()
        self.be = be
        self.n, self.algo, self.batch, self.gamma = n_envs, algo, batch, gamma
        self.rank, self.world = rank, world
        self.observe, self.explore = observe, explore
        self.epsilon, self.initial_epsilon, self.final_epsilon = initial_epsilon, initial_epsilon, final_epsilon
        self.replace_target_iter = replace_target_iter
        self.seed = seed
        self.env = be.env(n_envs, seed + 1000003 * rank)     # envs shard by rank: every rank plays its own games
        self.replay = be.replay(capacity, n_envs, algo == "per")     # ... into its own replay shard
        self.replay.seed(seed + rank, sampler)
        self.net = be.net(2, fc_width, arch, max(n_envs, batch))
        self.net.init_params(seed=seed, which=0)             # the same draw on every rank
        self.net.init_params(seed=seed + 1, which=1)
        if world > 1:                                        # replicas start from rank 0's parameters, bit for bit
            for which in (0, 1):
                flat = self.net.store_params(which)
                fdist.broadcast_params(flat, src=0)
                self.net.load_params(flat, which)
        self.grad = be.zeros(self.net.n_params) if world > 1 else None
        # one all-reduce of the flat gradient per train step (the HIP backend can split it in two: dist.OverlappedAllReduce)
        self.reduce = None
        if self.grad is not None:
            mk = getattr(be, "reducer", None)
            self.reduce = mk(self.net, self.grad, MEAN_LOSS[algo]) if mk else (lambda: fdist.allreduce_gradients(self.grad, MEAN_LOSS[algo]))
        self.timeStep = 0
        self.onlineTimeStep = 0
        self.nib = self.env.track_state()                    # currentState of every env, maintained by the env kernel
        self.env.observe()
        self.replay.reset(self.env.frame_bits)
        self.stats = self.env.track_stats()                  # [episodes, score sum, score max, pipes passed], kept by the env kernel
        self.last_loss = None
        self.dtype = "f32"
        # the whole step is one host call (fb_vec_step): uniform replay with the head, random.sample and the Memory append riding in the
        # env launch; prioritized replay with store -> Memory.sample -> weighted train -> batch_update as launches of the same call
        self.native = None
        if self.grad is not None and algo != "per" and hasattr(be, "native"):
            self.native = be.native(rank, world)
        if algo == "per" and not getattr(be, "per_one_step", False):
            self.one_step = None                             # (a backend without the fused prioritized step: the separate calls below)
        elif self.native is not None:
            self.one_step = be.step(self.env, self.replay, self.net, batch, algo, gamma, self.grad, dist=self.native, mean_loss=MEAN_LOSS[algo])
        else:
            self.one_step = be.step(self.env, self.replay, self.net, batch, algo, gamma, self.grad)

    def train_step(self, idx=None):
        if self.algo in ("nature", "double") and self.timeStep % self.replace_target_iter == 0:
            self.net.sync_target()
        isw = None
        if idx is None:
            idx, isw = self.replay.sample(self.batch)
        if hasattr(self.be, "train_from_replay") and self.batch <= 256:
            # the conv trunk reads the sampled transitions' frame bits in the ring: no gather launch, no u8 copies
            loss, abs_err = self.be.train_from_replay(self.replay, self.net, self.algo, idx, isw, self.gamma, self.grad, self.algo == "per")
        else:
            s, a, r, s2, t = self.replay.gather(idx)
            loss, abs_err, _ = self.net.train_step(self.algo, s, a, r, s2, t, isw=isw, gamma=self.gamma, flat_grad=self.grad,
                                                   want_aux=self.algo == "per")
        if self.grad is not None:
            self.reduce()
            self.net.apply_adam(self.grad)
        if self.algo == "per":
            self.replay.update_priorities(idx, abs_err=abs_err)
        self.last_loss = loss

    def step(self):
        if self.one_step is not None:
            training = self.onlineTimeStep > self.observe
            if training and self.algo in ("nature", "double") and self.timeStep % self.replace_target_iter == 0:
                self.net.sync_target()                       # acting reads the online net only: same result as syncing before training
            self.one_step(self.epsilon, seed=self.seed + self.rank, step=self.timeStep, train=training)
            if self.epsilon > self.final_epsilon and self.onlineTimeStep > self.observe:
                self.epsilon -= (self.initial_epsilon - self.final_epsilon) / self.explore
            if training:
                if self.grad is not None and self.native is None:      # (fb_vec_step_dp has reduced and applied already)
                    self.reduce()
                    self.net.apply_adam(self.grad)
                self.last_loss = self.one_step.loss
            self.timeStep += 1
            self.onlineTimeStep += 1
            return
        actions = self.net.act_nib(self.nib, self.epsilon, seed=self.seed + self.rank, step=self.timeStep)
        if self.epsilon > self.final_epsilon and self.onlineTimeStep > self.observe:
            self.epsilon -= (self.initial_epsilon - self.final_epsilon) / self.explore
        _, reward, terminal, _ = self.env.frame_step(actions, want_u8=False)
        training = self.onlineTimeStep > self.observe
        idx = None
        if training and self.algo != "per":                  # store + random.sample in one launch (same indices)
            idx = self.replay.push_sample(self.env.frame_bits, actions, reward, terminal, self.batch)
        else:
            self.replay.push(self.env.frame_bits, actions, reward, terminal)
        if training:
            self.train_step(idx)
        self.timeStep += 1
        self.onlineTimeStep += 1

    def set_dtype(self, dtype="f32"):
        """'bf16' = BASELINE.json configs[2]'s arithmetic for acting AND training (fp32 master weights / Adam); 'f32' = default."""
        self.net.set_inference_dtype(dtype)
        self.net.set_train_dtype(dtype)
        self.dtype = dtype

    # ------------------------------------------------------------------ checkpoint / resume of the WHOLE loop
    @staticmethod
    def _npz(path):
        return path if str(path).endswith(".npz") else str(path) + ".npz"

    def _local_path(self, path):
        """where this rank's LOCAL state goes: `<path>.rank<r>.npz` (envs, frame stacks, replay shard and stats differ per rank)"""
        base = str(path)[:-4] if str(path).endswith(".npz") else str(path)
        return f"{base}.rank{self.rank}.npz"

    def save(self, path):
        """Everything the device-resident loop needs to continue bit for bit: both nets + Adam slots (what the reference saves,
        BrainDQN.py:227-233), the three scalars, AND what it forgets (:176-192): the replay memory (frame ring, a / r / t, sampler
        generator, SumTree), onlineTimeStep, every env's state and the agents' frame stacks.

        world = 1: one file, `path`.  world > 1: the REPLICATED part (nets, Adam, scalars -- bit-identical on every rank) is written by
        rank 0 alone to `path`; every rank writes its own envs / frame stacks / replay shard / stats to `<path>.rank<r>.npz`.  (All ranks
        writing `path` -- round 3 -- overwrote each other's rank-local state.)  A barrier closes the call: when it returns on any rank,
        every file of the checkpoint is complete."""
        import numpy as np
        host = lambda t: t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)
        local = dict(env_state=self.env.get_state(), stats=host(self.stats), replay=self.replay.state_blob())
        if self.nib is not None:
            local["nib"] = host(self.nib)
        shared = None
        if self.rank == 0:
            m, v, pows = self.net.adam_state()
            shared = dict(online=host(self.net.store_params(0)), target=host(self.net.store_params(1)), adam_m=host(m), adam_v=host(v),
                          beta_pows=np.asarray(pows, np.float32), scalars=np.array([self.timeStep, self.onlineTimeStep, self.world, self.seed], np.int64),
                          epsilon=np.array([self.epsilon], np.float64))
        if self.world == 1:
            np.savez(self._npz(path), **shared, **local)
            return
        np.savez(self._local_path(path), **local)
        if self.rank == 0:
            np.savez(self._npz(path), **shared)
        fdist.barrier()

    def load(self, path):
        """the inverse of save(); at world > 1 every rank reads the replicated part from `path` and its own `<path>.rank<r>.npz`
        (the world size must be the one the checkpoint was written with: the shards are rank-local)."""
        import numpy as np
        z = np.load(self._npz(path))
        saved_world = int(z["scalars"][2]) if len(z["scalars"]) > 2 else 1
        if saved_world != self.world:
            raise ValueError(f"checkpoint {path} was written by {saved_world} rank(s), this job has {self.world}")
        zl = np.load(self._local_path(path)) if self.world > 1 else z
        dev = self.be.to_device if hasattr(self.be, "to_device") else np.ascontiguousarray
        self.net.load_params(z["online"], 0)
        self.net.load_params(z["target"], 1)
        self.net.set_adam_state(dev(z["adam_m"]), dev(z["adam_v"]), z["beta_pows"])
        self.env.set_state(zl["env_state"])
        if self.nib is not None:
            self.nib.copy_(dev(zl["nib"]))
        self.stats[...] = dev(zl["stats"])
        self.replay.load_state_blob(zl["replay"])
        self.timeStep, self.onlineTimeStep = int(z["scalars"][0]), int(z["scalars"][1])
        if len(z["scalars"]) > 3:
            self.seed = int(z["scalars"][3])                     # the key of the acting (epsilon-greedy) stream: (seed + rank, timeStep)
        self.epsilon = float(z["epsilon"][0])
        if hasattr(self.be, "synchronize"):
            self.be.synchronize()

    def run(self, steps, log_every=100):
        for i in range(steps):
            self.step()
            if log_every and (i + 1) % log_every == 0:
                loss = self.last_loss.item() if self.last_loss is not None else float("nan")
                ep, ssum, smax, pipes = self.stats.tolist()      # the only host sync of the loop, once per log line
                if hasattr(self.net, "check_range"):
                    self.net.check_range()                       # (the same sync: an activation beyond the two-plane fp16 range raises here)
                if hasattr(self.net, "split_stats"):
                    self.net.split_stats()                       # (... and so does a wait between the two streams of fb_vec_step's split schedule that gave up)
                print(f"TIMESTEP {self.timeStep} / ENVS {self.n} / EPSILON {self.epsilon:.6f} / GAME_TIMES {ep} / "
                      f"MEAN_SCORE {ssum / max(ep, 1):.3f} / MAX_SCORE {smax} / PIPES {pipes} / LOSS {loss:.6g}", flush=True)
