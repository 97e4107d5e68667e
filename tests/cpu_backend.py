"""A CPU stand-in for dqnflappybird_amd.backend.HipBackend, built on the oracle (tests only): lets the
`-m "not gpu"` suite run the Brain classes' host logic (schedules, RNG order, API surface)."""
import numpy as np

from oracle import oracle as o


class CpuNet:
    def __init__(self, actions, fc_width, arch, max_batch):
        self.cfg = o.qcfg(fc_width, actions, arch == "dueling")
        self.n_params = o.nparams(self.cfg)
        self.p = [np.zeros(self.n_params, np.float32), np.zeros(self.n_params, np.float32)]
        self.opt = o.Adam(self.n_params)
        self.syncs = 0
        self.train_calls = []

    def init_params(self, seed=0, which=0):
        self.p[which] = o.init_params(self.cfg, seed)

    def forward(self, states, which=0):
        return o.forward(self.p[which], self.cfg, states)

    def sync_target(self):
        self.p[1] = self.p[0].copy()
        self.syncs += 1

    def train_step(self, algo, s, a, r, s2, t, isw=None, gamma=0.99, flat_grad=None, want_aux=True):
        q, acts = o.forward(self.p[0], self.cfg, s, keep=True)
        if algo == "dqn":
            qn = o.forward(self.p[0], self.cfg, s2).max(1)
        elif algo == "double":
            am = o.forward(self.p[0], self.cfg, s2).argmax(1)
            qn = o.forward(self.p[1], self.cfg, s2)[np.arange(len(am)), am]
        else:
            qn = o.forward(self.p[1], self.cfg, s2).max(1)
        kind = {"dqn": 0, "nature": 1, "double": 1, "per": 2}[algo]
        y, loss, ae, dq = o.dqn_loss(kind, q, qn, a, r, t, isw=None if isw is None else np.asarray(isw, np.float32), gamma=gamma)
        g = o.backward(self.p[0], self.cfg, s, acts, dq)
        self.train_calls.append(algo)
        if flat_grad is None:
            self.opt.step(self.p[0], g)
        else:
            flat_grad[...] = g
        return np.float32(loss), ae, y

    def pg_step(self, states, actions, weights, n_total=None, flat_grad=None):
        q, acts = o.forward(self.p[0], self.cfg, states, keep=True)
        loss, dq = o.pg_loss(q, actions, weights, n_total)
        g = o.backward(self.p[0], self.cfg, states, acts, dq)
        self.train_calls.append("pg")
        if flat_grad is None:
            self.opt.step(self.p[0], g)
        else:
            flat_grad[...] = g
        return np.float32(loss)

    def apply_adam(self, g):
        self.opt.step(self.p[0], g)

    def load_params(self, flat, which=0):
        self.p[which] = np.array(flat, np.float32)

    def store_params(self, which=0):
        return self.p[which].copy()

    def adam_state(self):
        return self.opt.m.copy(), self.opt.v.copy(), np.array([self.opt.b1p.value, self.opt.b2p.value], np.float32)

    def set_adam_state(self, m, v, pows):
        self.opt.m[:], self.opt.v[:] = m, v
        self.opt.b1p.value, self.opt.b2p.value = float(pows[0]), float(pows[1])


class CpuReplay:
    """deque semantics of BrainDQN.py:66-72 for one env (+ the oracle Memory for PER)."""

    def __init__(self, capacity, prioritized):
        self.cap, self.prioritized = capacity, prioritized
        self.mem, self.state = [], None
        self.per = o.Memory(capacity) if prioritized else None
        self.slots = {}

    def reset(self, frames):
        f = np.asarray(frames).reshape(80, 80)
        self.state = np.stack([f] * 4, axis=2)
        self.mem = []

    def push(self, frames, a, r, t):
        new = np.append(self.state[:, :, 1:], np.asarray(frames).reshape(80, 80, 1), axis=2)
        tr = (self.state, int(np.asarray(a)[0]), float(np.asarray(r)[0]), new, int(np.asarray(t)[0]))
        if self.prioritized:
            self.slots[self.per.data_pointer] = tr
            self.per.store()
        else:
            self.mem.append(tr)
            if len(self.mem) > self.cap:
                self.mem.pop(0)
        self.state = new

    def sample(self, n, uniforms=None):
        idx, isw = self.per.sample(n, u=np.asarray(uniforms))
        return idx.astype(np.int64), isw

    def gather(self, idx):
        idx = np.asarray(idx)
        trs = [self.slots[int(i) - (self.cap - 1)] for i in idx] if self.prioritized else [self.mem[int(i)] for i in idx]
        s = np.stack([x[0] for x in trs]); s2 = np.stack([x[3] for x in trs])
        return (s, np.array([x[1] for x in trs], np.uint8), np.array([x[2] for x in trs], np.float32), s2,
                np.array([x[4] for x in trs], np.uint8))

    def update_priorities(self, idx, abs_err=None, priorities=None):
        if priorities is not None:
            self.per.batch_update_p(np.asarray(idx, np.int32), priorities)
        else:
            self.per.batch_update(np.asarray(idx, np.int32), abs_err)

    def per_state(self, want_tree=True):
        return self.per.tree.copy(), self.per.data_pointer, self.per.size, self.per.beta


class CpuBackend:
    name = "cpu-oracle (tests only)"

    def make_net(self, actions, fc_width, arch, max_batch):
        return CpuNet(actions, fc_width, arch, max_batch)

    def make_replay(self, capacity, prioritized):
        return CpuReplay(capacity, prioritized)

    @staticmethod
    def dev(x, dtype=None):
        return np.ascontiguousarray(x)

    @staticmethod
    def host(t):
        return np.asarray(t)

    @staticmethod
    def zeros(n):
        return np.zeros(n, np.float32)


# --------------------------------------------------------------------------------------------------------------------
# CPU stand-ins for the vectorised loop (dqnflappybird_amd.vecbrain.HipVecBackend), tests only: N oracle envs, the deque N envs
# appending in env order build, CPython's random.sample, the oracle network.  They mirror what fb_vec_step does per step so
# that VecBrain's rank logic (env / replay / acting seeds per rank, gradient sum vs mean over ranks, target-sync placement,
# parameter broadcast) runs under a world-size-2 gloo group without a GPU.
class CpuVecEnv:
    def __init__(self, n_envs, seed):
        self.n, self.seed = n_envs, seed
        self.envs = [o.GameState(seed=seed, env_id=e) for e in range(n_envs)]
        self.frame_bits = None
        self.episodes = 0

    def track_state(self):
        return None

    def track_stats(self):
        return np.zeros(4, np.int64)

    def observe(self):
        self.frame_bits = np.stack([g.frame80() for g in self.envs])
        return self.frame_bits

    def get_state(self):
        """the envs' C structs, byte for byte (ctypes); what fb_env_get_state is to the HIP path"""
        import ctypes as C
        return np.stack([np.frombuffer(bytes(g.e), np.uint8) for g in self.envs])

    def set_state(self, state):
        import ctypes as C
        for g, row in zip(self.envs, np.asarray(state, np.uint8)):
            C.memmove(C.byref(g.e), row.tobytes(), C.sizeof(g.e))
        self.observe()

    def step(self, actions):
        out = [g.step(int(a)) for g, a in zip(self.envs, actions)]
        self.frame_bits = np.stack([g.frame80() for g in self.envs])
        self.reward = np.array([x[0] for x in out], np.float32)
        self.terminal = np.array([x[1] for x in out], np.uint8)
        return self.frame_bits, self.reward, self.terminal


class CpuVecReplay:
    """uniform memory of N envs: one transition per env and step, appended in env order (fb_replay.hip's deque order)"""

    def __init__(self, capacity, n_envs, prioritized):
        assert not prioritized
        self.cap, self.n = capacity, n_envs
        self.mem, self.states, self.rng = [], None, None

    def seed(self, seed, rng=None):
        self.rng = o.PyRandom(seed)

    def reset(self, frames):
        self.states = [np.stack([f] * 4, axis=2) for f in frames]
        self.mem = []

    def current_state(self):
        return np.stack(self.states)

    def push(self, frames, a, r, t):
        for e in range(self.n):
            new = np.append(self.states[e][:, :, 1:], frames[e].reshape(80, 80, 1), axis=2)
            self.mem.append((self.states[e], int(a[e]), float(r[e]), new, int(t[e])))
            self.states[e] = new
        del self.mem[:max(0, len(self.mem) - self.cap)]

    def sample(self, batch):
        return self.rng.sample(len(self.mem), batch)

    def state_blob(self):
        """the whole memory as one uint8 array: frame stacks, transitions, the sampler's generator (fb_replay_save_state's counterpart)"""
        import ctypes as C
        mem = np.zeros((len(self.mem), 2 * 25600 + 8), np.uint8)
        for i, (s0, a, r, s1, t) in enumerate(self.mem):
            mem[i, :25600], mem[i, 25600:51200] = s0.ravel(), s1.ravel()
            mem[i, 51200] = a; mem[i, 51201] = t; mem[i, 51204:51208] = np.frombuffer(np.float32(r).tobytes(), np.uint8)
        head = np.frombuffer(np.int64(len(self.mem)).tobytes(), np.uint8)
        return np.concatenate([head, np.frombuffer(bytes(self.rng.s), np.uint8), np.stack(self.states).ravel(), mem.ravel()])

    def load_state_blob(self, blob):
        import ctypes as C
        blob = np.asarray(blob, np.uint8)
        n = int(np.frombuffer(blob[:8].tobytes(), np.int64)[0]); o_ = 8
        sz = C.sizeof(self.rng.s)
        C.memmove(C.byref(self.rng.s), blob[o_:o_ + sz].tobytes(), sz); o_ += sz
        self.states = [x.copy() for x in blob[o_:o_ + self.n * 25600].reshape(self.n, 80, 80, 4)]; o_ += self.n * 25600
        mem = blob[o_:].reshape(n, 2 * 25600 + 8)
        self.mem = [(m[:25600].reshape(80, 80, 4).copy(), int(m[51200]), float(np.frombuffer(m[51204:51208].tobytes(), np.float32)[0]),
                     m[25600:51200].reshape(80, 80, 4).copy(), int(m[51201])) for m in mem]

    def gather(self, idx):
        trs = [self.mem[int(i)] for i in idx]
        return (np.stack([x[0] for x in trs]), np.array([x[1] for x in trs], np.uint8), np.array([x[2] for x in trs], np.float32),
                np.stack([x[3] for x in trs]), np.array([x[4] for x in trs], np.uint8))


class CpuVecNet(CpuNet):
    """CpuNet speaking torch CPU tensors at its boundary (what torch.distributed moves)"""

    def store_params(self, which=0):
        import torch
        return torch.from_numpy(self.p[which].copy())

    def load_params(self, flat, which=0):
        self.p[which] = np.array(flat, np.float32)

    def apply_adam(self, g):
        self.opt.step(self.p[0], np.asarray(g, np.float32))


class CpuVecStep:
    """fb_vec_step on the stand-ins: act (Q + the Philox epsilon-greedy stream of fb_head.h) -> step -> store ->
    random.sample -> gather -> train"""

    def __init__(self, env, replay, net, batch, algo, gamma, flat_grad):
        self.env, self.replay, self.net = env, replay, net
        self.batch, self.algo, self.gamma, self.flat_grad = batch, algo, gamma, flat_grad
        self.loss, self.idx, self.actions = None, None, None

    def __call__(self, epsilon, seed=0, step=0, train=True):
        q = self.net.forward(self.replay.current_state())
        acts = np.argmax(q, axis=1).astype(np.uint8)
        for e in range(self.env.n):
            ph = o.philox(seed & 0xFFFFFFFF, seed >> 32, e, step & 0xFFFFFFFF, 1, step >> 32)      # FB_STREAM_EPS
            u = np.float32(int(ph[0]) >> 8) * np.float32(1.0 / 16777216.0)
            if u <= np.float32(epsilon):
                acts[e] = (int(ph[1]) * 2) >> 32
        frames, r, t = self.env.step(acts)
        self.replay.push(frames, acts, r, t)
        self.actions = acts
        if train:
            self.idx = self.replay.sample(self.batch)
            s, a, r, s2, t = self.replay.gather(self.idx)
            g = None if self.flat_grad is None else np.empty(self.net.n_params, np.float32)
            self.loss, _, _ = self.net.train_step(self.algo, s, a, r, s2, t, gamma=self.gamma, flat_grad=g)
            if g is not None:
                self.flat_grad.copy_(__import__("torch").from_numpy(g))
        return acts


class CpuVecBackend:
    name = "cpu-oracle (tests only)"

    def env(self, n_envs, seed):
        return CpuVecEnv(n_envs, seed)

    def replay(self, capacity, n_envs, prioritized):
        return CpuVecReplay(capacity, n_envs, prioritized)

    def net(self, actions, fc_width, arch, max_batch):
        return CpuVecNet(actions, fc_width, arch, max_batch)

    def step(self, env, replay, net, batch, algo, gamma, flat_grad):
        return CpuVecStep(env, replay, net, batch, algo, gamma, flat_grad)

    def zeros(self, n):
        import torch
        return torch.zeros(n, dtype=torch.float32)
