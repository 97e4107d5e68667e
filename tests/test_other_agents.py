"""BrainPolicyGradient / BrainDQNActorCritic: the host logic on the CPU stand-in backend, and the
policy-gradient loss of the HIP path (FB_ALGO_PG) against the oracle.  Both agents are broken upstream; DESIGN.md section 8 records
what was decided about each bug, and these tests pin those decisions."""
import random

import numpy as np
import pytest

from tests.test_brain_host_logic import frames_source


def drive(brain, oracle, steps, seed=4):
    """FlappyBirdDQN.py:60-76 with the oracle env"""
    first, step_env = frames_source(oracle, seed)
    brain.setInitState(first)
    ends = []
    for n in range(steps):
        a = brain.getAction()
        assert a.dtype == np.float64 and a.sum() == 1
        obs, r, term, score = step_env(int(a[1]))
        brain.setPerception(obs, a, r, term, score)
        if term:
            ends.append(n)
    return ends


def test_pg_loss_oracle_vs_torch(oracle):
    """fbo_pg_loss == mean(softmax_cross_entropy_with_logits(q, onehot(a)) * w) and its gradient (BrainPolicyGradient.py:96-100)"""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    B = 37
    q = rng.standard_normal((B, 2)).astype(np.float32) * 3
    a = rng.integers(0, 2, B).astype(np.uint8)
    w = rng.choice(np.array([0.1, 3, -3], np.float32), B)
    loss, dq = oracle.pg_loss(q, a, w)
    qt = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    L = (F.cross_entropy(qt, torch.tensor(a.astype(np.int64)), reduction="none") * torch.tensor(w, dtype=torch.float64)).mean()
    L.backward()
    np.testing.assert_allclose(loss, L.item(), rtol=1e-6)
    np.testing.assert_allclose(dq, qt.grad.numpy(), rtol=1e-5, atol=1e-9)
    # a chunk of a larger batch carries its share of the mean
    loss2, dq2 = oracle.pg_loss(q[:10], a[:10], w[:10], n_total=B)
    np.testing.assert_allclose(dq2, dq[:10], rtol=1e-6)


@pytest.mark.parametrize("faithful", [True, False])
def test_policy_gradient_host_logic(oracle, tmp_path, faithful):
    """episode memory, one train step per episode on ALL its states as one batch (:131-137), the raw-rewards bug behind `faithful`,
    actions drawn from np.random's global stream (:186)."""
    from dqnflappybird_amd.BrainPolicyGradient import BrainPolicyGradient
    from tests.cpu_backend import CpuBackend
    np.random.seed(5)
    b = BrainPolicyGradient(2, "bird", backend=CpuBackend(), verbose=False, seed=1, save_root=str(tmp_path / "sp"),
                            logs_root=str(tmp_path / "logs_"), faithful=faithful)
    fed = []
    orig = b.net.pg_step
    b.net.pg_step = lambda s, a, w, n=None, flat_grad=None: (fed.append((len(s), np.asarray(w).copy(), n)), orig(s, a, w, n, flat_grad))[1]
    p0 = b.net.p[0].copy()
    seen_rewards = []
    first, step_env = frames_source(oracle, 4)
    b.setInitState(first)
    steps = 0
    while b.gameTimes < 2 and steps < 400:
        a = b.getAction()
        obs, r, term, score = step_env(int(a[1]))
        seen_rewards.append(r)
        b.setPerception(obs, a, r, term, score)
        steps += 1
    assert b.gameTimes == 2 and b.timeStep == steps and not b.ep_states      # the episode memory is emptied by the train step
    assert len(b.lost_hist) == 2 and len(fed) >= 2
    n0 = fed[0][0]
    assert fed[0][2] == n0 and b.time_steps_when_episode_end[0] == n0 - 1     # the whole first episode, one batch
    ep0 = np.asarray(seen_rewards[:n0], np.float32)
    if faithful:
        assert np.array_equal(fed[0][1], ep0) and ep0[-1] == -3               # RAW rewards, as the reference's code feeds them (:136)
    else:
        w = fed[0][1]
        assert abs(w.mean()) < 1e-5 and abs(w.std() - 1) < 1e-3               # discounted + normalised returns (:200-211)
    assert not np.array_equal(b.net.p[0], p0)
    b.timeStep = 100000                                                        # the save the reference would die in (self.epsilon, :147)
    b.ep_states, b.ep_acts, b.ep_rewards = [b.currentState, b.currentState], [np.array([1., 0.]), np.array([0., 1.])], [0.1, -3]
    b.trainQNetwork()
    b2 = BrainPolicyGradient(2, "bird", backend=CpuBackend(), verbose=False, seed=9, save_root=str(tmp_path / "sp"), logs_root=str(tmp_path / "logs_"))
    assert b2.timeStep == 100000 and np.array_equal(b2.net.p[0], b.net.p[0])


def test_policy_gradient_long_episode_is_one_batch_in_chunks(oracle, tmp_path):
    """an episode longer than 128 states: chunks export their share of the MEAN's gradient, one Adam step -- equal to the one-batch step"""
    from dqnflappybird_amd.BrainPolicyGradient import BrainPolicyGradient
    from tests.cpu_backend import CpuBackend
    rng = np.random.default_rng(1)
    mk = lambda: BrainPolicyGradient(2, "bird", backend=CpuBackend(), verbose=False, seed=3, save_root=str(tmp_path / "sp"),
                                     logs_root=str(tmp_path / "logs_"), record_logs=False)
    a, b = mk(), mk()
    n = 150
    states = [(rng.random((80, 80, 4)) < 0.3).astype(np.uint8) * 255 for _ in range(n)]
    acts = [np.eye(2)[rng.integers(0, 2)] for _ in range(n)]
    rews = [float(x) for x in rng.choice([0.1, 3.0, -3.0], n)]
    for br in (a, b):
        br.ep_states, br.ep_acts, br.ep_rewards, br.timeStep = list(states), list(acts), list(rews), 7
    a.trainQNetwork()                                           # 128 + 22
    import dqnflappybird_amd.BrainPolicyGradient as M
    old, M.CHUNK = M.CHUNK, 1000
    try:
        b.trainQNetwork()                                       # one batch of 150 (the CPU stand-in has no chunk limit)
    finally:
        M.CHUNK = old
    np.testing.assert_allclose(a.lost, b.lost, rtol=1e-5)
    assert np.abs(a.net.p[0] - b.net.p[0]).max() < 2e-7         # Adam normalises: both moved by ~lr per element, the same way
    assert len(a.net.train_calls) == 2 and len(b.net.train_calls) == 1


@pytest.mark.parametrize("faithful", [True, False])
def test_actor_critic_host_logic(oracle, tmp_path, faithful):
    """one critic step (td^2, V(s') from the critic itself) and one actor step per env step (:194-211); the actor's sign as written
    upstream behind `faithful`."""
    from dqnflappybird_amd.BrainActorCritic import BrainDQNActorCritic
    from tests.cpu_backend import CpuBackend
    np.random.seed(2)
    b = BrainDQNActorCritic(2, "bird", backend=CpuBackend(), verbose=False, seed=1, save_root=str(tmp_path / "sp"), logs_root=str(tmp_path / "logs_"),
                            faithful=faithful)
    assert b.critic.cfg.actions == 1 and b.actor.cfg.actions == 2
    first, step_env = frames_source(oracle, 6)
    b.setInitState(first)
    for n in range(6):
        s = b.currentState.copy()
        a = b.getAction()
        obs, r, term, score = step_env(int(a[1]))
        v_s = float(b.critic.forward(s[None])[0, 0])
        nxt = np.append(s[:, :, 1:], obs, axis=2)
        v_n = float(b.critic.forward(nxt[None])[0, 0])
        logp = float(np.log(b.action_prob(s)[int(a[1])]))
        b.setPerception(obs, a, r, term, score)
        td = (0.1 if r == np.float32(0.1) else r) + 0.99 * v_n - v_s
        assert abs(b.td_error - td) < 1e-5
        assert abs(b.lost_hist_critic[-1] - td * td) < 1e-5 * max(1, td * td)
        want = logp * td if faithful else -logp * td              # faithful: the reference's own loss number, log pi * td (:99)
        assert abs(b.lost_hist_actor[-1] - want) < 1e-4 * max(1, abs(want))
        assert abs(b.q_target_critic_list[-1] - (td + v_s)) < 1e-5
    assert b.timeStep == 6 and b.actor.train_calls == ["pg"] * 6 and b.critic.train_calls == ["dqn"] * 6


@pytest.mark.gpu
@pytest.mark.parametrize("dueling,B,n_total", [(False, 32, None), (False, 100, 400), (True, 17, None), (False, 1, 1)])
def test_pg_step_on_device_matches_oracle(oracle, dueling, B, n_total):
    """FB_ALGO_PG through the C ABI: loss and every gradient tensor vs the oracle (fbo_pg_loss + fbo_qnet_backward)."""
    import torch
    import zlib
    assert torch.cuda.is_available()
    from dqnflappybird_amd.vec import QNet
    from tests.test_gpu_qnet import rand_states, trained_like_params
    cfg = oracle.qcfg(512, 2, dueling)
    p = trained_like_params(oracle, cfg, 3)
    net = QNet(2, 512, "dueling" if dueling else "plain", max_batch=128)
    net.load_params(p, 0)
    for attempt in range(60):
        rng = np.random.default_rng(zlib.crc32(f"pg-{dueling}-{B}-{attempt}".encode()))
        s = rand_states(rng, B)
        q, acts = oracle.forward(p, cfg, s, keep=True)
        if oracle.last_margin() > 2e-5:
            break
    else:
        pytest.fail("no kink-free batch found")
    a = rng.integers(0, 2, B).astype(np.uint8)
    w = rng.choice(np.array([0.1, 3, -3], np.float32), B, p=[0.6, 0.2, 0.2])
    d = lambda x: torch.from_numpy(x).cuda()
    grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda")
    loss = net.pg_step(d(s), d(a), d(w), n_total, flat_grad=grad)
    loss0, dq = oracle.pg_loss(q, a, w, n_total)
    g0 = oracle.backward(p, cfg, s, acts, dq)
    np.testing.assert_allclose(loss.item(), loss0, rtol=2e-5, atol=1e-6)
    g = grad.cpu().numpy()
    bounds = [0, 8192, 8224, 40992, 41056, 77920, 77984, 77984 + 1600 * 512, 77984 + 1600 * 512 + 512, net.n_params]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        scale = np.abs(g0[lo:hi]).max()
        assert scale > 0
        np.testing.assert_allclose(g[lo:hi], g0[lo:hi], rtol=2e-3, atol=2e-5 * scale, err_msg=f"params[{lo}:{hi}]")
    assert np.array_equal(net.store_params().cpu().numpy(), p)
    with pytest.raises(ValueError):                               # chunks of at most 128
        net2 = QNet(max_batch=200)
        net2.pg_step(d(rand_states(rng, 200)), d(np.zeros(200, np.uint8)), d(np.ones(200, np.float32)))


@pytest.mark.gpu
def test_other_agents_run_on_the_device(tmp_path, oracle):
    import torch
    assert torch.cuda.is_available()
    from dqnflappybird_amd.BrainActorCritic import BrainDQNActorCritic
    from dqnflappybird_amd.BrainPolicyGradient import BrainPolicyGradient
    np.random.seed(1)
    kw = dict(verbose=False, seed=2, save_root=str(tmp_path / "sp"), logs_root=str(tmp_path / "logs_"), record_logs=False)
    pg = BrainPolicyGradient(2, "bird", **kw)
    p0 = pg.net.store_params().clone()
    drive(pg, oracle, 90)
    assert pg.gameTimes >= 1 and np.isfinite(pg.lost) and not torch.equal(pg.net.store_params(), p0)
    ac = BrainDQNActorCritic(2, "bird", **kw)
    a0, c0 = ac.actor.store_params().clone(), ac.critic.store_params().clone()
    drive(ac, oracle, 12)
    assert ac.timeStep == 12 and np.isfinite(ac.lost_hist_actor).all() and np.isfinite(ac.lost_hist_critic).all()
    assert not torch.equal(ac.actor.store_params(), a0) and not torch.equal(ac.critic.store_params(), c0)
