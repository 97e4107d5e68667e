"""Vectorised front end of the HIP hot path: N envs / N-env replay / Q-network as torch-ROCm
tensors over the C ABI (include/fbdqn.h).  PyTorch is plumbing here (device memory, streams,
torch.distributed); every computation is a kernel of libfbdqn.so.

Reference counterparts:
    VecGameState  -> game/wrapped_flappy_bird.py GameState, N at a time, preprocess fused in
    VecReplay     -> BrainDQN.replayMemory (deque) / BrainPrioritizedReplyDQN.Memory
    QNet          -> BrainDQN._createQNetwork / _trainQNetwork and the variants' versions
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _dev_check(*tensors):
    for t in tensors:
        if t is not None and not (t.is_cuda and t.is_contiguous()):
            raise ValueError("tensors handed to the HIP path must be contiguous CUDA(ROCm) tensors")


class VecGameState:
    """N independent Flappy Bird games on the GPU (game/wrapped_flappy_bird.py:58-183)."""

    def __init__(self, n_envs, seed=0, device="cuda"):
        L.require_gpu()
        self.n = int(n_envs)
        self.device = torch.device(device)
        self.h = C.c_void_p()
        blob = L.sprite_blob()
        L.check(L.lib().fb_env_create(self.n, seed, 0, blob, len(blob), C.byref(self.h)), "fb_env_create")
        dev = self.device
        self.frames = torch.empty((self.n, 80, 80), dtype=torch.uint8, device=dev)
        self.frame_bits = torch.empty((self.n, 100), dtype=torch.int64, device=dev)
        self.reward = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.terminal = torch.empty(self.n, dtype=torch.uint8, device=dev)
        self.score = torch.empty(self.n, dtype=torch.int32, device=dev)

    def __del__(self):
        try:                                     # at interpreter shutdown the module globals may already be gone
            if getattr(self, "h", None) and self.h.value:
                L.lib().fb_env_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def reset(self):
        L.check(L.lib().fb_env_reset(self.h, L.current_stream()), "fb_env_reset")

    def frame_step(self, actions, want_u8=True, frame_bits=None):
        """actions: uint8[N] (0 = nothing, 1 = flap).  Returns (frames u8[N,80,80] or None,
        reward f32[N], terminal u8[N], score i32[N]); the packed frame is in self.frame_bits
        (or in `frame_bits` when given, e.g. a replay-ring slot)."""
        _dev_check(actions, frame_bits)
        if actions.dtype != torch.uint8 or actions.numel() != self.n:
            raise ValueError("actions must be uint8[N]")
        fb = self.frame_bits if frame_bits is None else frame_bits
        L.check(L.lib().fb_env_step(self.h, L.ptr(actions), L.ptr(self.frames) if want_u8 else None, L.ptr(fb),
                                    L.ptr(self.reward), L.ptr(self.terminal), L.ptr(self.score),
                                    L.current_stream()), "fb_env_step")
        return (self.frames if want_u8 else None), self.reward, self.terminal, self.score

    def track_state(self):
        """Keep the agents' 4-frame stacks on the device in nibble form (fb_env_set_nib_buffer); returns the
        u8[N, FB_NIB_STRIDE] tensor (conv1's SAME padding included, include/fbdqn.h) that QNet.act_nib consumes.
        Call before observe()."""
        self.nib = torch.zeros((self.n, L.NIB_STRIDE), dtype=torch.uint8, device=self.device)
        L.check(L.lib().fb_env_set_nib_buffer(self.h, L.ptr(self.nib)), "fb_env_set_nib_buffer")
        return self.nib

    def track_stats(self):
        """Device-side episode counters (fb_env_set_stats_buffer): returns the int64[4] tensor
        [episodes ended, sum of their scores, max score, pipes passed]; read it whenever you log."""
        self.stats = torch.zeros(4, dtype=torch.int64, device=self.device)
        L.check(L.lib().fb_env_set_stats_buffer(self.h, L.ptr(self.stats)), "fb_env_set_stats_buffer")
        return self.stats

    def observe(self):
        L.check(L.lib().fb_env_observe(self.h, L.ptr(self.frames), L.ptr(self.frame_bits), L.current_stream()),
                "fb_env_observe")
        return self.frames

    def get_state(self):
        out = np.empty((self.n, 16), np.int32)
        L.check(L.lib().fb_env_get_state(self.h, L.ptr(out)), "fb_env_get_state")
        return out

    def set_state(self, state):
        state = np.ascontiguousarray(state, np.int32)
        assert state.shape == (self.n, 16)
        L.check(L.lib().fb_env_set_state(self.h, L.ptr(state)), "fb_env_set_state")

    def set_gap_tape(self, tape):
        if tape is None:
            L.check(L.lib().fb_env_set_gap_tape(self.h, None, 0), "fb_env_set_gap_tape")
            return
        tape = np.ascontiguousarray(tape, np.int8)
        assert tape.ndim == 2 and tape.shape[0] == self.n
        L.check(L.lib().fb_env_set_gap_tape(self.h, L.ptr(tape), tape.shape[1]), "fb_env_set_gap_tape")

    def render_full(self, env_id=0):
        out = torch.empty((288, 512, 3), dtype=torch.uint8, device=self.device)
        L.check(L.lib().fb_env_render_full(self.h, env_id, L.ptr(out), L.current_stream()), "fb_env_render_full")
        return out

    def error_count(self):
        v = C.c_int64()
        L.check(L.lib().fb_env_error_count(self.h, C.byref(v)), "fb_env_error_count")
        return v.value


class VecReplay:
    """Replay memory in HBM for N envs (BrainDQN.py:36,69-72,197-201;
    BrainPrioritizedReplyDQN.py:32-151).  Frames are stored once, 1 bit per pixel."""

    def __init__(self, capacity, n_envs=1, prioritized=False, device="cuda"):
        L.require_gpu()
        self.capacity, self.n, self.prioritized = int(capacity), int(n_envs), bool(prioritized)
        self.device = torch.device(device)
        self.h = C.c_void_p()
        L.check(L.lib().fb_replay_create(self.capacity, self.n, L.REPLAY_PER if prioritized else L.REPLAY_UNIFORM,
                                         C.byref(self.h)), "fb_replay_create")
        self._buf = {}

    def __del__(self):
        try:                                     # at interpreter shutdown the module globals may already be gone
            if getattr(self, "h", None) and self.h.value:
                L.lib().fb_replay_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def seed(self, seed, rng=None):
        """rng: 'cpython' (random.seed), 'numpy' (np.random.seed) or 'philox'."""
        kind = {"cpython": L.RNG_CPYTHON, "philox": L.RNG_PHILOX, "numpy": L.RNG_NUMPY}[
            rng or ("numpy" if self.prioritized else "cpython")]
        L.check(L.lib().fb_replay_seed(self.h, kind, seed), "fb_replay_seed")

    @staticmethod
    def _split(frames):
        """u8[N,80,80] -> (frames, None); int64[N,100] packed bits -> (None, bits)."""
        _dev_check(frames)
        if frames.dtype == torch.uint8:
            return frames, None
        if frames.dtype == torch.int64:
            return None, frames
        raise ValueError("frames must be uint8[N,80,80] or int64[N,100] (packed)")

    def reset(self, first_frames):
        f, b = self._split(first_frames)
        L.check(L.lib().fb_replay_reset(self.h, L.ptr(f), L.ptr(b), L.current_stream()), "fb_replay_reset")

    def push(self, next_frames, actions, rewards, terminals):
        f, b = self._split(next_frames)
        _dev_check(actions, rewards, terminals)
        L.check(L.lib().fb_replay_push(self.h, L.ptr(f), L.ptr(b), L.ptr(actions), L.ptr(rewards), L.ptr(terminals),
                                       L.current_stream()), "fb_replay_push")

    def push_sample(self, next_frames, actions, rewards, terminals, batch):
        """push(...) then sample(batch) of a uniform memory in one launch (fb_replay_push_sample) -> idx int64[B]."""
        if self.prioritized:
            raise ValueError("push_sample is for uniform memories (PER needs the importance weights: push + sample)")
        f, b = self._split(next_frames)
        _dev_check(actions, rewards, terminals)
        idx = self._get(f"idx{batch}", (batch,), torch.int64)
        L.check(L.lib().fb_replay_push_sample(self.h, L.ptr(f), L.ptr(b), L.ptr(actions), L.ptr(rewards), L.ptr(terminals),
                                              batch, L.ptr(idx), L.current_stream()), "fb_replay_push_sample")
        return idx

    def _get(self, name, shape, dtype):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buf[name] = t
        return t

    def current_state(self):
        out = self._get("cur", (self.n, 80, 80, 4), torch.uint8)
        L.check(L.lib().fb_replay_current_state(self.h, L.ptr(out), L.current_stream()), "fb_replay_current_state")
        return out

    def sample(self, batch, uniforms=None, out=None):
        """-> idx int64[B] (deque positions, or SumTree indices for PER), isw float64[B] or None.
        `out`: an int64[B] device tensor to receive the indices (default: a buffer reused by every call)."""
        idx = self._get(f"idx{batch}", (batch,), torch.int64) if out is None else out
        _dev_check(idx)
        isw = self._get(f"isw{batch}", (batch,), torch.float64) if self.prioritized else None
        _dev_check(uniforms)
        L.check(L.lib().fb_replay_sample(self.h, batch, L.ptr(uniforms), L.ptr(idx), L.ptr(isw), L.current_stream()),
                "fb_replay_sample")
        return idx, isw

    def gather(self, idx):
        B = idx.numel()
        s = self._get(f"s{B}", (B, 80, 80, 4), torch.uint8)
        s2 = self._get(f"s2{B}", (B, 80, 80, 4), torch.uint8)
        a = self._get(f"a{B}", (B,), torch.uint8)
        r = self._get(f"r{B}", (B,), torch.float32)
        t = self._get(f"t{B}", (B,), torch.uint8)
        L.check(L.lib().fb_replay_gather(self.h, B, L.ptr(idx), L.ptr(s), L.ptr(s2), L.ptr(a), L.ptr(r), L.ptr(t),
                                         L.current_stream()), "fb_replay_gather")
        return s, a, r, s2, t

    def update_priorities(self, idx, abs_err=None, priorities=None):
        _dev_check(idx, abs_err, priorities)
        L.check(L.lib().fb_replay_update_priorities(self.h, idx.numel(), L.ptr(idx), L.ptr(abs_err), L.ptr(priorities),
                                                    L.current_stream()), "fb_replay_update_priorities")

    def set_per_mode(self, mode="exact"):
        """'exact' (reference-order running sums, bit-identical tree) or 'fast' (level-wise recomputation)."""
        L.check(L.lib().fb_replay_set_per_mode(self.h, {"exact": L.PER_EXACT, "fast": L.PER_FAST}[mode]),
                "fb_replay_set_per_mode")

    def __len__(self):
        v = C.c_int64()
        L.check(L.lib().fb_replay_size(self.h, C.byref(v)), "fb_replay_size")
        return v.value

    def state_blob(self):
        """The whole memory as one numpy uint8 blob (fb_replay_save_state): frame ring, a / r / t rows, counters, sampler
        generator, SumTree heaps -- what a resumed run needs to continue the index stream bit for bit."""
        n = C.c_size_t()
        L.check(L.lib().fb_replay_state_bytes(self.h, C.byref(n)), "fb_replay_state_bytes")
        blob = np.empty(n.value, np.uint8)
        L.check(L.lib().fb_replay_save_state(self.h, L.ptr(blob), n.value), "fb_replay_save_state")
        return blob

    def load_state_blob(self, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        L.check(L.lib().fb_replay_load_state(self.h, L.ptr(blob), blob.size), "fb_replay_load_state")

    def per_state(self, want_tree=True):
        tree = np.empty(2 * self.capacity - 1, np.float64) if want_tree else None
        ptr_, size, beta = C.c_int64(), C.c_int64(), C.c_double()
        L.check(L.lib().fb_replay_per_tree(self.h, L.ptr(tree), C.byref(ptr_), C.byref(size), C.byref(beta)),
                "fb_replay_per_tree")
        return tree, ptr_.value, size.value, beta.value


ALGOS = {"dqn": L.ALGO_DQN, "nature": L.ALGO_NATURE, "double": L.ALGO_DOUBLE, "per": L.ALGO_PER, "pg": L.ALGO_PG}


class QNet:
    """The reference Q-network (BrainDQN.py:119-163) with forward, backward and TF-Adam as HIP
    kernels.  `arch='dueling'` builds the head of BrainDuelingDQN.py:78-86."""

    def __init__(self, actions=2, fc_width=512, arch="plain", max_batch=32, device="cuda"):
        L.require_gpu()
        self.A, self.FC, self.max_batch = int(actions), int(fc_width), int(max_batch)
        self.dueling = arch == "dueling"
        self.device = torch.device(device)
        self.h = C.c_void_p()
        L.check(L.lib().fb_qnet_create(L.ARCH_DUELING if self.dueling else L.ARCH_PLAIN, self.FC, self.A, self.max_batch,
                                       C.byref(self.h)), "fb_qnet_create")
        n = C.c_int64()
        L.check(L.lib().fb_qnet_num_params(self.h, C.byref(n)), "fb_qnet_num_params")
        self.n_params = n.value
        self._buf = {}

    def __del__(self):
        try:                                     # at interpreter shutdown the module globals may already be gone
            if getattr(self, "h", None) and self.h.value:
                L.lib().fb_qnet_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass

    def _get(self, name, shape, dtype):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buf[name] = t
        return t

    # -- parameters -----------------------------------------------------------------
    def init_params(self, seed=0, which=L.NET_ONLINE):
        L.check(L.lib().fb_qnet_init_params(self.h, which, seed, L.current_stream()), "fb_qnet_init_params")

    def load_params(self, flat, which=L.NET_ONLINE):
        if not torch.is_tensor(flat):
            flat = torch.from_numpy(np.ascontiguousarray(flat, np.float32)).to(self.device)
        _dev_check(flat)
        if flat.dtype != torch.float32 or flat.numel() != self.n_params:
            raise ValueError(f"expected float32[{self.n_params}]")
        L.check(L.lib().fb_qnet_load_params(self.h, which, L.ptr(flat), L.current_stream()), "fb_qnet_load_params")
        torch.cuda.current_stream().synchronize()     # `flat` may be a temporary

    def store_params(self, which=L.NET_ONLINE):
        out = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        L.check(L.lib().fb_qnet_store_params(self.h, which, L.ptr(out), L.current_stream()), "fb_qnet_store_params")
        return out

    def set_hparams(self, lr=1e-6, beta1=0.9, beta2=0.999, eps=1e-8):
        L.check(L.lib().fb_qnet_set_hparams(self.h, lr, beta1, beta2, eps), "fb_qnet_set_hparams")

    def overflow_count(self, reset=False):
        """waves that split an ACTIVATION beyond the two-plane fp16 range (|x| >= 32768, include/fbdqn.h) since creation / the last
        reset: 0 = every forward / train step so far computed in range.  Synchronous (one device word)."""
        v = C.c_int64()
        L.check(L.lib().fb_qnet_overflow_count(self.h, int(bool(reset)), C.byref(v)), "fb_qnet_overflow_count")
        return v.value

    def split_stats(self):
        """(steps fb_vec_step issued with the train step on its own stream, how many of those started beside their env step)."""
        a, b = C.c_int64(), C.c_int64()
        L.check(L.lib().fb_qnet_split_stats(self.h, C.byref(a), C.byref(b)), "fb_qnet_split_stats")
        return a.value, b.value

    def check_range(self):
        """raise FbError if a launch of this net met an activation beyond the fp32-equivalent path's range (its results are then
        inf / NaN / wrong where TensorFlow's fp32 would have carried on)"""
        n = self.overflow_count()
        if n:
            raise L.FbError(f"{n} wave(s) of the Q-network's fp32-equivalent (two-plane fp16) kernels met an activation of magnitude >= 32768: "
                            "those Q-values / gradients are not to be trusted.  Use set_inference_dtype('bf16') / set_train_dtype('bf16') "
                            "(fp32's exponent range) for this net, or rescale its weights (include/fbdqn.h, fb_qnet_overflow_count).")

    def set_inference_dtype(self, dtype="f32"):
        """'f32' (default) or 'bf16': arithmetic of forward / act / act_nib on >= 256 states (BASELINE config 3)."""
        L.check(L.lib().fb_qnet_set_inference_dtype(self.h, {"f32": L.DTYPE_F32, "bf16": L.DTYPE_BF16}[dtype]),
                "fb_qnet_set_inference_dtype")

    def set_train_dtype(self, dtype="f32"):
        """'f32' (default) or 'bf16': arithmetic of train_step (bf16 operands, fp32 accumulation, fp32 master weights + Adam)."""
        L.check(L.lib().fb_qnet_set_train_dtype(self.h, {"f32": L.DTYPE_F32, "bf16": L.DTYPE_BF16}[dtype]), "fb_qnet_set_train_dtype")

    def adam_state(self):
        m = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        v = torch.empty_like(m)
        pows = np.empty(2, np.float32)
        L.check(L.lib().fb_qnet_get_adam_state(self.h, L.ptr(m), L.ptr(v), L.ptr(pows)), "fb_qnet_get_adam_state")
        return m, v, pows

    def set_adam_state(self, m, v, pows):
        pows = np.ascontiguousarray(pows, np.float32)
        _dev_check(m, v)
        L.check(L.lib().fb_qnet_set_adam_state(self.h, L.ptr(m), L.ptr(v), L.ptr(pows)), "fb_qnet_set_adam_state")

    def sync_target(self):
        L.check(L.lib().fb_qnet_sync_target(self.h, L.current_stream()), "fb_qnet_sync_target")

    # -- compute --------------------------------------------------------------------
    def forward(self, states, which=L.NET_ONLINE):
        _dev_check(states)
        B = states.shape[0]
        if states.dtype != torch.uint8 or tuple(states.shape[1:]) != (80, 80, 4):
            raise ValueError("states must be uint8[B,80,80,4]")
        q = self._get(f"q{B}", (B, self.A), torch.float32)
        L.check(L.lib().fb_qnet_forward(self.h, which, L.ptr(states), B, L.ptr(q), L.current_stream()), "fb_qnet_forward")
        return q

    def act(self, states, epsilon, seed=0, step=0, want_q=False):
        _dev_check(states)
        n = states.shape[0]
        actions = self._get(f"act{n}", (n,), torch.uint8)
        q = self._get(f"qa{n}", (n, self.A), torch.float32) if want_q else None
        L.check(L.lib().fb_qnet_act(self.h, L.ptr(states), n, float(epsilon), seed, step, L.ptr(actions), L.ptr(q),
                                    L.current_stream()), "fb_qnet_act")
        return (actions, q) if want_q else actions

    def act_nib(self, nib_states, epsilon, seed=0, step=0, want_q=False):
        """getAction for N envs straight from VecGameState.track_state()'s nibble states."""
        _dev_check(nib_states)
        n = nib_states.shape[0]
        actions = self._get(f"act{n}", (n,), torch.uint8)
        q = self._get(f"qa{n}", (n, self.A), torch.float32) if want_q else None
        L.check(L.lib().fb_qnet_act_nib(self.h, L.ptr(nib_states), n, float(epsilon), seed, step, L.ptr(actions), L.ptr(q),
                                        L.current_stream()), "fb_qnet_act_nib")
        return (actions, q) if want_q else actions

    def train_step(self, algo, s, a, r, s2, t, isw=None, gamma=0.99, flat_grad=None, want_aux=True):
        """One _trainQNetwork step.  Returns (loss f32[1], abs_err f32[B], q_target f32[B]) device tensors.
        flat_grad=None applies Adam; a float32[n_params] tensor receives the gradients instead."""
        _dev_check(s, a, r, s2, t, isw, flat_grad)
        B = s.shape[0]
        loss = self._get("loss", (1,), torch.float32)
        ae = self._get(f"ae{B}", (B,), torch.float32) if want_aux else None
        y = self._get(f"y{B}", (B,), torch.float32) if want_aux else None
        if isw is not None and isw.dtype != torch.float32:
            isw = isw.to(torch.float32)          # ISWeights is fed to a float32 placeholder
        L.check(L.lib().fb_qnet_train_step(self.h, ALGOS[algo] if isinstance(algo, str) else algo, B, L.ptr(s), L.ptr(a),
                                           L.ptr(r), L.ptr(s2), L.ptr(t), L.ptr(isw), float(gamma), L.ptr(loss), L.ptr(ae),
                                           L.ptr(y), L.ptr(flat_grad), L.current_stream()), "fb_qnet_train_step")
        return loss, ae, y

    def pg_step(self, states, actions, weights, n_total=None, flat_grad=None):
        """One policy-gradient step (FB_ALGO_PG) on <= 128 states: loss = sum over them of softmax_cross_entropy(logits, action) x
        weight / n_total (n_total defaults to the chunk size = a plain mean).  flat_grad=None applies Adam; a float32[n_params] tensor
        receives the chunk's gradient instead (episodes longer than 128: add the chunks' gradients, then apply_adam once).
        -> loss f32[1] (device)."""
        B = states.shape[0]
        zeros = self._get(f"pgt{B}", (B,), torch.uint8)
        zeros.zero_()
        loss, _, _ = self.train_step("pg", states, actions, weights, states, zeros, gamma=float(n_total or B), flat_grad=flat_grad, want_aux=False)
        return loss

    def apply_adam(self, flat_grad):
        _dev_check(flat_grad)
        L.check(L.lib().fb_qnet_apply_adam(self.h, L.ptr(flat_grad), L.current_stream()), "fb_qnet_apply_adam")


def train_from_replay(replay, net, algo, idx, gamma=0.99, flat_grad=None, isw=None, want_abs_err=False):
    """replay.gather(idx) + net.train_step(...) without the gathered copies (fb_train_from_replay): the conv trunk reads the sampled
    transitions' 1-bit frames in the ring directly.  Same results as the two calls (batch <= 256).  Prioritized replay: idx are the
    SumTree leaf indices of replay.sample, isw its importance weights; want_abs_err returns |TD error| for update_priorities.
    -> (loss f32[1], a u8[B], r f32[B], t u8[B][, abs_err f32[B]]) on the device."""
    if (replay.prioritized or algo == "per") and isw is None:
        raise ValueError("the prioritized step needs the importance weights (isw)")
    _dev_check(idx, flat_grad, isw)
    B, dev = int(idx.numel()), idx.device
    a = torch.empty(B, dtype=torch.uint8, device=dev); r = torch.empty(B, dtype=torch.float32, device=dev)
    t = torch.empty(B, dtype=torch.uint8, device=dev); loss = torch.zeros(1, dtype=torch.float32, device=dev)
    ae = torch.empty(B, dtype=torch.float32, device=dev) if want_abs_err else None
    if isw is not None and isw.dtype != torch.float32:
        isw = isw.to(torch.float32)                      # ISWeights is fed to a float32 placeholder
    L.check(L.lib().fb_train_from_replay(replay.h, net.h, ALGOS[algo], B, L.ptr(idx), L.ptr(isw), L.ptr(a), L.ptr(r), L.ptr(t), float(gamma),
                                         L.ptr(loss), L.ptr(ae), L.ptr(flat_grad), L.current_stream()),
            "fb_train_from_replay")
    return (loss, a, r, t, ae) if want_abs_err else (loss, a, r, t)


class TrainSteps:
    """n x (random.sample -> minibatch -> _trainQNetwork) on a uniform memory that is not being pushed to, as one host call
    (fb_train_steps): the separate calls' results, with the next step's random.sample riding in the conv3 backward launch."""

    def __init__(self, replay, net, batch=32, algo="dqn", gamma=0.99):
        if replay.prioritized or algo == "per":
            raise ValueError("TrainSteps is for uniform replay (PER needs the importance weights: use the separate calls)")
        self.replay, self.net, self.batch, self.algo, self.gamma = replay, net, batch, ALGOS[algo], float(gamma)
        dev, B = replay.device, batch
        self.idx = torch.zeros(2 * B, dtype=torch.int64, device=dev)     # two buffers, used alternately
        self.s = torch.empty((B, 80, 80, 4), dtype=torch.uint8, device=dev)
        self.s2 = torch.empty((B, 80, 80, 4), dtype=torch.uint8, device=dev)
        self.a = torch.empty(B, dtype=torch.uint8, device=dev)
        self.r = torch.empty(B, dtype=torch.float32, device=dev)
        self.t = torch.empty(B, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)

    def __call__(self, n_steps=1):
        L.check(L.lib().fb_train_steps(self.replay.h, self.net.h, self.algo, self.batch, int(n_steps), L.ptr(self.idx), L.ptr(self.s),
                                       L.ptr(self.s2), L.ptr(self.a), L.ptr(self.r), L.ptr(self.t), L.ptr(self.loss), self.gamma,
                                       L.current_stream()), "fb_train_steps")
        return self.loss


class VecStep:
    """One whole step of the vectorised loop (FlappyBirdDQN.py:72-76 for N envs) as a single
    host call, fb_vec_step: getAction -> frame_step -> store + sample -> minibatch -> _trainQNetwork (prioritized memories: ->
    Memory.batch_update as well).
    The same C-ABI calls in the same order as the separate VecGameState / VecReplay / QNet methods (identical
    results); the pointers are bound once, so the interpreter spends one ctypes call per step instead of five."""

    def __init__(self, env, replay, net, batch=32, algo="dqn", gamma=0.99, flat_grad=None, dist=None, mean_loss=False):
        """flat_grad: export the gradient instead of applying Adam (data parallel; the caller all-reduces and calls net.apply_adam).
        dist: a dist.NativeDP -- then the call is fb_vec_step_dp: the step, the all-reduce of flat_grad through the library's own
        RCCL communicator (overlapped with the conv backward) and Adam, all in the one host call; mean_loss divides by the world size."""
        if replay.prioritized != (algo == "per"):
            raise ValueError("algo 'per' goes with a prioritized memory, every other algo with a uniform one")
        if dist is not None and flat_grad is None:
            raise ValueError("VecStep(dist=...) needs the flat_grad buffer the gradient is reduced in")
        self.dist, self.mean_loss = dist, int(bool(mean_loss))
        if getattr(env, "nib", None) is None:
            raise ValueError("call env.track_state() first: the acting path reads the env kernel's nibble states")
        _dev_check(flat_grad)
        self.env, self.replay, self.net = env, replay, net
        self.batch, self.algo, self.gamma, self.flat_grad = batch, ALGOS[algo], float(gamma), flat_grad
        dev, B = env.device, batch
        self.actions = torch.zeros(env.n, dtype=torch.uint8, device=dev)
        self.idx = torch.zeros(B, dtype=torch.int64, device=dev)
        self.s = torch.empty((B, 80, 80, 4), dtype=torch.uint8, device=dev)
        self.s2 = torch.empty((B, 80, 80, 4), dtype=torch.uint8, device=dev)
        self.a = torch.empty(B, dtype=torch.uint8, device=dev)
        self.r = torch.empty(B, dtype=torch.float32, device=dev)
        self.t = torch.empty(B, dtype=torch.uint8, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        per = algo == "per"                                # Memory.sample's weights (f64, and as the float32 placeholder takes them), |TD errors|
        self.isw = torch.zeros(B, dtype=torch.float64, device=dev) if per else None
        self.isw32 = torch.zeros(B, dtype=torch.float32, device=dev) if per else None
        self.abs_err = torch.zeros(B, dtype=torch.float32, device=dev) if per else None
        p = lambda x: None if x is None else x.data_ptr()
        self.buf = L.StepBuffers(p(env.nib), p(self.actions), p(env.frame_bits), p(env.reward), p(env.terminal), p(env.score),
                                 p(self.idx), p(self.s), p(self.s2), p(self.a), p(self.t), p(self.r), p(self.loss), p(flat_grad),
                                 p(self.isw), p(self.isw32), p(self.abs_err))

    def __call__(self, epsilon, seed=0, step=0, train=True):
        """-> actions uint8[N] (device); rewards / terminals / scores are the env's tensors, loss is self.loss."""
        if self.dist is not None:
            L.check(L.lib().fb_vec_step_dp(self.dist.handle, self.env.h, self.replay.h, self.net.h, C.byref(self.buf), self.env.n, self.algo,
                                           self.batch, float(epsilon), int(seed), int(step), int(bool(train)), self.gamma, self.mean_loss,
                                           L.current_stream()), "fb_vec_step_dp")
            return self.actions
        L.check(L.lib().fb_vec_step(self.env.h, self.replay.h, self.net.h, C.byref(self.buf), self.env.n, self.algo, self.batch,
                                    float(epsilon), int(seed), int(step), int(bool(train)), self.gamma, L.current_stream()),
                "fb_vec_step")
        return self.actions
