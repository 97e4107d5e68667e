"""Host logic of the drop-in Brain classes on CPU (compute delegated to tests/cpu_backend.py):
the reference's schedules and RNG consumption order (BrainDQN.py:66-116,195-223 and the variants)."""
import os
import random
import tempfile

import numpy as np
import pytest

from tests.cpu_backend import CpuBackend


def frames_source(oracle, seed):
    env = oracle.GameState(seed=seed)
    env.step(0)
    first = env.frame80()

    def step(action):
        r, t, s = env.step(int(action))
        return env.frame80().reshape(80, 80, 1), (0.1 if abs(r - 0.1) < 1e-6 else int(r)), t, s
    return first, step


def make(cls, **kw):
    kw.setdefault("save_root", "/nonexistent/saved_parameters")
    kw.setdefault("logs_root", os.path.join(tempfile.mkdtemp(), "logs_"))
    b = cls(2, 'bird', backend=CpuBackend(), verbose=False, seed=1, **kw)
    b.OBSERVE, b.BATCH_SIZE = 12., 8          # instance attributes: small so that training starts quickly
    return b


def run(brain, step_env, first, n):
    brain.setInitState(first)
    for _ in range(n):
        a = brain.getAction()
        assert a.dtype == np.float64 and a.sum() == 1 and a.shape == (2,)
        obs, r, t, s = step_env(int(a[1]))
        brain.setPerception(obs, a, r, t, s)


def test_dqn_schedule_and_rng_order(oracle):
    """observe -> train gate on onlineTimeStep, epsilon decay, and the exact `random` call sequence."""
    from dqnflappybird_amd.BrainDQN import BrainDQN
    first, step_env = frames_source(oracle, 3)
    random.seed(77)
    brain = make(BrainDQN)
    N = 30
    run(brain, step_env, first, N)
    assert brain.timeStep == N and brain.onlineTimeStep == N and len(brain) == N
    n_train = N - 13                           # steps with onlineTimeStep in 13..29
    assert len(brain.net.train_calls) == n_train and set(brain.net.train_calls) == {"dqn"}
    assert brain.net.syncs == 0
    # epsilon: decremented in getAction once onlineTimeStep > OBSERVE  (reference :113-114)
    assert abs(brain.epsilon - (0.03 - n_train * 0.03 / 1e6)) < 1e-15
    # replay the reference's call order on a twin stream: seed draw in __init__, then per step
    # random() [+ randrange on explore], and random.sample(range(len), B) per train step
    random_state_after = random.getstate()
    random.seed(77)
    random.getrandbits(48)                     # the constructor's weight-init seed draw is skipped (seed given) -> undo
    random.seed(77)
    eps, explored = 0.03, 0
    for t in range(N):
        if random.random() <= eps:
            random.randrange(2)
            explored += 1
        if t > 12:
            eps -= 0.03 / 1e6
            random.sample(range(t + 1), 8)
    assert random.getstate() == random_state_after


def test_current_state_is_newest_last_and_never_reset(oracle):
    from dqnflappybird_amd.BrainDQN import BrainDQN
    first, step_env = frames_source(oracle, 4)
    brain = make(BrainDQN)
    brain.setInitState(first)
    assert brain.currentState.shape == (80, 80, 4) and all(np.array_equal(brain.currentState[:, :, k], first) for k in range(4))
    hist = [first]
    for i in range(70):                        # never flapping crashes within 70 steps: the stack must not reset
        a = np.array([1., 0.])
        obs, r, t, s = step_env(0)
        hist.append(obs[:, :, 0])
        brain.setPerception(obs, a, r, t, s)
        for k in range(4):
            assert np.array_equal(brain.currentState[:, :, k], hist[max(0, len(hist) - 4 + k)])
    assert brain.gameTimes >= 1


def test_nature_target_sync_on_timestep_multiple(oracle):
    from dqnflappybird_amd.BrainDQNNature import BrainDQNNature
    first, step_env = frames_source(oracle, 5)
    brain = make(BrainDQNNature)
    brain.REPLACE_TARGET_ITER = 10
    run(brain, step_env, first, 45)
    # trained at timeStep 13..44; syncs at 20, 30, 40  (reference :151-152)
    assert brain.net.syncs == 3 and set(brain.net.train_calls) == {"nature"}
    assert not np.array_equal(brain.net.p[0], brain.net.p[1])


def test_ddqn_and_dueling_are_nature_unless_asked(oracle):
    from dqnflappybird_amd.BrainDoubleDQN import BrainDoubleDQN
    from dqnflappybird_amd.BrainDuelingDQN_CC import BrainDuelingDQN
    first, step_env = frames_source(oracle, 6)
    b = make(BrainDoubleDQN)
    run(b, step_env, first, 16)
    assert set(b.net.train_calls) == {"nature"} and b.dir_name == "/double_dqn/"
    b2 = make(BrainDoubleDQN, faithful=False)
    run(b2, step_env, first, 16)
    assert set(b2.net.train_calls) == {"double"}
    d = make(BrainDuelingDQN)
    assert d.net.cfg.dueling == 0 and d.net.n_params == 898722
    d2 = make(BrainDuelingDQN, faithful=False)
    assert d2.net.cfg.dueling == 1 and d2.net.n_params == 899235


def test_per_brain_never_syncs_and_uses_numpy_stream(oracle):
    from dqnflappybird_amd.BrainPrioritizedReplyDQN import BrainPrioritizedReplyDQN
    first, step_env = frames_source(oracle, 7)
    np.random.seed(5)
    brain = make(BrainPrioritizedReplyDQN)
    brain.REPLACE_TARGET_ITER = 5
    run(brain, step_env, first, 20)
    assert brain.net.syncs == 0 and set(brain.net.train_calls) == {"per"}
    assert abs(brain.replayMemory.beta - (0.4 + 7 * 0.001)) < 1e-12          # 7 train steps
    after = np.random.get_state()[1].copy(), np.random.get_state()[2]
    np.random.seed(5)
    for _ in range(7 * 8):
        np.random.uniform(0, 1)
    twin = np.random.get_state()
    assert np.array_equal(after[0], twin[1]) and after[1] == twin[2]


def test_game_module_surface():
    """constants of game/wrapped_flappy_bird.py:14-50 and the hit masks, without a GPU."""
    from dqnflappybird_amd.game import flappy_bird_utils as fu
    images, sounds, hit = fu.load()
    assert images["pipe"][0].shape == (320, 52, 4) and images["player"][0].shape == (24, 34, 4)
    assert sounds == {} and len(hit["pipe"]) == 2 and len(hit["player"]) == 3
    assert len(hit["pipe"][0]) == 52 and len(hit["pipe"][0][0]) == 320


def test_hitmasks_equal_reference(golden):
    from dqnflappybird_amd.game import flappy_bird_utils as fu
    g = golden("game_trajectories.npz")
    _, _, hit = fu.load()
    assert np.array_equal(np.array(hit["pipe"][0], np.uint8), g["hit_pipe_upper"])
    assert np.array_equal(np.array(hit["pipe"][1], np.uint8), g["hit_pipe_lower"])
    assert np.array_equal(np.array(hit["player"], np.uint8), g["hit_player"])


def test_checkpoint_resume_roundtrip(oracle, tmp_path):
    """reference BrainDQN.py:176-192,227-233: parameters + Adam slots + the three pickled scalars; the
    scalar file has the reference's exact format (three consecutive pickles)."""
    import pickle
    from dqnflappybird_amd.BrainDQNNature import BrainDQNNature
    first, step_env = frames_source(oracle, 8)
    random.seed(3)
    root = str(tmp_path / "saved_parameters")
    a = make(BrainDQNNature, save_root=root)
    a.SAVE_EVERY = 7
    run(a, step_env, first, 30)                       # trains from step 13, saves at timeStep 14, 21, 28
    assert sorted(os.listdir(root + "/dqn_nature/")) == ["bird-14.npz", "bird-21.npz", "bird-28.npz",
                                                          "bird-saved-parameters.txt", "checkpoint"]
    with open(root + "/dqn_nature/bird-saved-parameters.txt", "rb") as f:
        game_times, time_step, eps = pickle.load(f), pickle.load(f), pickle.load(f)
    assert time_step == 28 and isinstance(game_times, int) and 0.0299 < eps < 0.03
    b = make(BrainDQNNature, save_root=root)          # a fresh Brain picks the newest checkpoint up
    assert (b.timeStep, b.gameTimes, b.epsilon) == (time_step, game_times, eps) and b.onlineTimeStep == 0
    z = np.load(root + "/dqn_nature/bird-28.npz")
    assert np.array_equal(b.net.p[0], z["online"]) and np.array_equal(b.net.p[1], z["target"])
    assert np.array_equal(b.net.opt.m, z["adam_m"]) and b.net.opt.b1p.value == z["beta_pows"][0]
    assert not np.array_equal(z["online"], oracle.init_params(b.net.cfg, 1))     # it really trained


def test_reference_log_streams_and_plots(oracle, tmp_path):
    """reference BrainDQN.py:36-56,224-235,242-324: the five text streams (names, space-separated append format,
    lists cleared after a flush, q_targets as str(list)) and the four PNGs of _record_by_pic."""
    from dqnflappybird_amd.BrainDQN import BrainDQN
    first, step_env = frames_source(oracle, 5)
    random.seed(4)
    logs = str(tmp_path / "logs_")
    b = make(BrainDQN, save_root=str(tmp_path / "saved"), logs_root=logs)
    b.SAVE_EVERY = 10
    run(b, step_env, first, 35)                       # trains from step 13; flushes at timeStep 20 and 30
    d = logs + "bird/dqn/"
    assert sorted(os.listdir(d)) == ["lost_hist.txt", "q_targets.txt", "reward_every_time_step.txt",
                                     "score_every_episode.txt", "time_steps_when_episode_end.txt"]
    loss, scores, ends, rewards, q = b._get_loss_score_timestep_reward_qtarget_from_file()
    # train steps 13..30, BATCH 8; the flush at timeStep 30 runs inside that step's training, before its reward is appended
    assert len(loss) == 18 and len(q) == 18 * 8 and len(rewards) == 30
    assert len(scores) == len(ends) and all(r in (0.1, 3.0, -3.0) for r in rewards)
    assert len(b.lost_hist) == 4 and len(b.reward_every_time_step) == 5       # steps 31..34 / 30..34 are still in memory
    with open(d + "q_targets.txt") as f:
        assert f.read().startswith("[")                                          # str(list) entries, like the reference
    b._record_by_pic()
    pngs = sorted(n for n in os.listdir(d) if n.endswith(".png"))
    assert pngs == [f"35_{n}" for n in ("lost_hist_total.png", "q_target_total.png", "scores_episode_total.png",
                                       "scores_time_step_total.png")]
    assert b.lost_hist == []                                                      # _record_by_pic flushes first


def test_standalone_dueling_recipe(oracle, tmp_path, monkeypatch):
    """dqnflappybird_amd/BrainDuelingDQN.py, the reference's stand-alone script (BrainDuelingDQN.py:140-324):
    newest-first stack, t > OBSERVE gate, epsilon schedule, target sync on t % REPLACE_TARGET_ITER, checkpoints under
    the global step t + step, resume from the newest checkpoint's name, the 1 000 000-step backup copy (also at
    global step 0), and the Q function in the reference's channel order."""
    from dqnflappybird_amd import BrainDuelingDQN as R
    monkeypatch.setattr(R, "OBSERVE", 10.)
    monkeypatch.setattr(R, "BATCH", 8)
    monkeypatch.setattr(R, "SAVER_ITER", 12)
    monkeypatch.setattr(R, "REPLACE_TARGET_ITER", 6)
    assert (R.INITIAL_EPSILON, R.FINAL_EPSILON, R.EXPLORE, R.REPLAY_MEMORY) == (0.1, 0.0001, 3000000., 50000)

    class Game:                                    # game.GameState stand-in: frames from the oracle env
        def __init__(self):
            self.env = oracle.GameState(seed=3)
            self.frames = []

        def frame_step(self, a):
            assert a.sum() == 1
            r, t, s = self.env.step(int(a[1]))
            self.frames.append(self.env.frame80().copy())
            return self.frames[-1], (0.1 if abs(r - 0.1) < 1e-6 else int(r)), t, s

    game = Game()
    random.seed(7)
    handles = R.createNetwork(backend=CpuBackend(), seed=5)
    net = handles[0]
    assert len(handles) == 6 and all(h is net for h in handles)
    synced = []
    orig_sync = net.net.sync_target
    monkeypatch.setattr(net.net, "sync_target", lambda: (synced.append(1), orig_sync())[1])
    sp, sb = str(tmp_path / "saved") + "/", str(tmp_path / "back") + "/"
    os.makedirs(sp)
    t, eps, obs = R.trainNetwork(*handles, sess=None, game_state=game, preprocess=lambda x: x, max_steps=30, verbose=False,
                                 save_path=sp, save_back_path=sb)
    assert t == 30
    # newest first: channel 0 is the last frame, channel 3 the one three steps earlier
    for c in range(4):
        assert np.array_equal(obs[:, :, c], game.frames[-1 - c])
    # epsilon decays once per step while t > OBSERVE (t = 11..29)
    assert abs(eps - (0.1 - 19 * (0.1 - 0.0001) / 3000000.)) < 1e-12
    assert len(synced) == 3                         # t = 12, 18, 24
    assert sorted(os.listdir(sp)) == ["bird-dqn-12.npz", "bird-dqn-24.npz", "checkpoint"]
    assert os.listdir(sb) == ["0"]                  # (step + t) % 1e6 == 0 at the very first iteration, like the reference
    assert len(net) == 30
    # Q in the reference's channel order == the kernel-order network on the reversed stack
    q = net.q_values(obs)
    p_ref = net.get_params(0)
    p_int = np.asarray(net.be.host(net.net.store_params(0)))
    assert np.array_equal(p_ref[8192:], p_int[8192:]) and not np.array_equal(p_ref[:8192], p_int[:8192])
    assert np.array_equal(p_ref[:8192].reshape(8, 8, 4, 32)[:, :, ::-1, :], p_int[:8192].reshape(8, 8, 4, 32))
    cfg = oracle.qcfg(512, 2, True)
    want = oracle.forward(p_int, cfg, np.ascontiguousarray(obs[None, :, :, ::-1]))[0]
    np.testing.assert_allclose(q, want, atol=1e-5)
    # resume: the global step comes back from the checkpoint's name, parameters from its content
    h2 = R.createNetwork(backend=CpuBackend(), seed=99)
    saver, step = R.store_parameters(h2[0], sp, verbose=False)
    assert step == 24
    z = np.load(sp + "bird-dqn-24.npz")
    assert np.array_equal(h2[0].get_params(0), z["online"]) and np.array_equal(h2[0].get_params(1), z["target"])
    # counter_add: the reference's averaging (COUNTERS_SIZE = 2)
    del R.average_score[:]
    c = []
    R.counter_add(c, 4, 10, logs_path=str(tmp_path / "logs") + "/")
    R.counter_add(c, 6, 11, logs_path=str(tmp_path / "logs") + "/")
    assert c == [] and R.average_score == [5.0]
