"""Drop-in for BrainPrioritizedReplyDQN.py: SumTree, Memory and the agent
(reference BrainPrioritizedReplyDQN.py:32-151,155-366).

The tree lives in HBM as the reference's fp64 array heap and is updated with the reference's own
sequence of floating-point operations, so `tree` bytes, sampled `b_idx` and IS weights match the
reference's classes (tests/test_gpu_replay.py).  Kept quirks: beta is incremented before the
first sample, ISWeights are normalised by min_prob (not by the max weight), the target net is
NEVER synchronised (this class's _trainQNetwork does not run target_replace_op, :277-329)."""
import numpy as np

from .BrainDQNNature import BrainDQNNature

FRAME_PER_ACTION = 1
BATCH_SIZE = 32
OBSERVE = 1000.
EXPLORE = 1000000.
GAMMA = 0.99
FINAL_EPSILON = 0
INITIAL_EPSILON = 0.03
REPLAY_MEMORY = 50000
N_FEATURES = 80 * 80 * 4


class SumTree(object):
    """Reference :32-104 on top of a prioritized device memory without payload frames."""

    def __init__(self, capacity, backend=None, _replay=None):
        if backend is None:
            from .backend import HipBackend
            backend = HipBackend()
        self._be = backend
        self.capacity = capacity
        self._rep = _replay if _replay is not None else backend.make_replay(capacity, True)
        self.data = np.zeros(capacity, dtype=object)
        self.size = 0
        self.data_pointer = 0
        self._tree = None

    @property
    def tree(self):
        if self._tree is None:
            self._tree = self._rep.per_state()[0]
        return self._tree

    @property
    def total_p(self):
        return self.tree[0]

    def add(self, p, data):
        tree_idx = self.data_pointer + (self.capacity - 1)
        self.data[self.data_pointer] = data
        self.update(tree_idx, p)
        self.data_pointer += 1
        if self.data_pointer >= self.capacity:
            self.data_pointer = 0
        if self.size < self.capacity:
            self.size += 1

    def update(self, tree_idx, p):
        self._rep.update_priorities(self._be.dev(np.array([tree_idx], np.int64)),
                                    priorities=self._be.dev(np.array([p], np.float32)))
        self._tree = None

    def get_min_prob(self):
        return min(self.tree[self.capacity - 1: self.capacity + self.size - 1]) / self.total_p

    def get_leaf(self, v):
        tree, parent_idx = self.tree, 0
        while True:
            cl_idx = 2 * parent_idx + 1
            if cl_idx >= len(tree):
                leaf_idx = parent_idx
                break
            if v <= tree[cl_idx]:
                parent_idx = cl_idx
            else:
                v -= tree[cl_idx]
                parent_idx = cl_idx + 1
        return leaf_idx, tree[leaf_idx], self.data[leaf_idx - self.capacity + 1]


class Memory(object):
    """Reference :107-151.  `store` takes the reference's transition tuple
    (state, action, reward, newState, terminal); consecutive transitions must be consecutive
    steps of one env (newState = state shifted by one frame), which is the only way the
    reference uses it -- the ring stores each 80x80 frame once."""
    epsilon = 0.01
    alpha = 0.6
    beta = 0.4
    beta_increment_per_sampling = 0.001
    abs_err_upper = 1.

    def __init__(self, capacity, backend=None):
        if backend is None:
            from .backend import HipBackend
            backend = HipBackend()
        self._be = backend
        self.capacity = capacity
        self._rep = backend.make_replay(capacity, True)
        self.sum_tree = SumTree(capacity, backend, _replay=self._rep)
        self._started = False

    def reset(self, first_frame):
        self._rep.reset(self._be.dev(np.ascontiguousarray(first_frame.reshape(1, 80, 80), np.uint8)))
        self._started = True

    def store(self, transition):
        state, action, reward, new_state, terminal = transition
        if not self._started:
            self.reset(np.asarray(state)[:, :, 3])
        frame = np.ascontiguousarray(np.asarray(new_state)[:, :, 3].reshape(1, 80, 80), np.uint8)
        self._rep.push(self._be.dev(frame), self._be.dev(np.array([int(np.argmax(action))], np.uint8)),
                       self._be.dev(np.array([reward], np.float32)), self._be.dev(np.array([1 if terminal else 0], np.uint8)))
        t = self.sum_tree
        t.data_pointer = (t.data_pointer + 1) % self.capacity
        t.size = min(t.size + 1, self.capacity)
        t._tree = None

    def sample_device(self, n):
        """Memory.sample with np.random.uniform's stream, everything left on the device."""
        u = np.array([np.random.uniform(0.0, 1.0) for _ in range(n)])          # consumes numpy's global stream like :136
        idx, isw = self._rep.sample(n, uniforms=self._be.dev(u))
        self.beta = min(1., self.beta + self.beta_increment_per_sampling)
        return idx, isw

    def sample(self, n):
        idx, isw = self.sample_device(n)
        s, a, r, s2, t = (self._be.host(x) for x in self._rep.gather(idx))
        b_memory = np.empty((n,), dtype=object)
        for i in range(n):
            onehot = np.zeros(2)
            onehot[a[i]] = 1
            b_memory[i] = (s[i], onehot, float(r[i]), s2[i], bool(t[i]))
        return self._be.host(idx).astype(np.int32), b_memory, self._be.host(isw).reshape(n, 1)

    def batch_update(self, tree_idx, abs_errors):
        idx = tree_idx if hasattr(tree_idx, "is_cuda") else self._be.dev(np.asarray(tree_idx, np.int64))
        if hasattr(abs_errors, "is_cuda"):
            self._rep.update_priorities(idx, abs_err=abs_errors)
        else:
            e = self._be.dev(np.asarray(abs_errors, np.float32))
            self._rep.update_priorities(idx, abs_err=e)
            abs_errors[...] = self._be.host(e)                # the reference mutates its argument (+= epsilon)
        self.sum_tree._tree = None


class BrainPrioritizedReplyDQN(BrainDQNNature):
    ALGO = "per"
    PRIORITIZED = True
    DIR_NAME = "/prioritized_reply_dqn/"

    def __init__(self, actionNum, gameName, **kw):
        super().__init__(actionNum, gameName, **kw)
        self.replayMemory = Memory(capacity=self.REPLAY_MEMORY, backend=self._be)

    def setInitState(self, observ):
        observ = np.ascontiguousarray(np.asarray(observ).reshape(80, 80), np.uint8)
        self.currentState = np.stack((observ, observ, observ, observ), axis=2)
        self.replayMemory.reset(observ)
        self._n_stored = 0

    def _store(self, nextObserv, action, reward, terminal):
        newState = np.append(self.currentState[:, :, 1:], np.asarray(nextObserv).reshape(80, 80, 1), axis=2)
        self.replayMemory.store((self.currentState, action, reward, newState, terminal))
        self._n_stored += 1

    def _pre_train(self):
        pass                                                   # never syncs the target net (reference :277-329)

    def _trainQNetwork(self):
        tree_idx, isw = self.replayMemory.sample_device(self.BATCH_SIZE)
        s, a, r, s2, t = self.replayMemory._rep.gather(tree_idx)
        loss, abs_errors, y = self.net.train_step("per", s, a, r, s2, t, isw=isw, gamma=self.GAMMA)
        self.replayMemory.batch_update(tree_idx, abs_errors)
        self._after_train(loss, y)                               # reference :316-329
