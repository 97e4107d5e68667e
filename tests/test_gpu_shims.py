"""The drop-in Python surface on the GPU: game.wrapped_flappy_bird.GameState, preprocess,
Brain* classes (FlappyBirdDQN.py:36-76 loop) and the vectorised VecBrain loop."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.mark.parametrize("name,seed", [("traj0_never", 1), ("traj2_every7", 3), ("traj6_seek", 7)])
def test_gamestate_dropin_reproduces_reference_with_python_random(torch_cuda, golden, name, seed):
    """random.seed(s); GameState(); frame_step(one-hot)... == the reference's own module run with the
    same seed: pipes come from `random` in the reference's order, so states/rewards match step by step."""
    from dqnflappybird_amd.game import wrapped_flappy_bird as game
    g = golden("game_trajectories.npz")
    assert (game.SCREENWIDTH, game.SCREENHEIGHT, game.PIPE_WIDTH, game.PIPE_HEIGHT, game.PLAYER_WIDTH,
            game.PLAYER_HEIGHT, game.PIPEGAPSIZE, game.FPS) == (288, 512, 52, 320, 34, 24, 100, 30)
    assert abs(game.BASEY - 404.48) < 1e-12
    random.seed(seed)
    gs = game.GameState()
    st = gs._env.get_state()
    st[0, 13] = int(g[name + "_cyc0"][0])        # PLAYER_INDEX_GEN phase the fixture's process had
    gs._env.set_state(st)
    T = min(len(g[name + "_action"]), 400)
    for t in range(T):
        a = int(g[name + "_action"][t])
        onehot = np.zeros(2)
        onehot[a] = 1
        img, r, term, score = gs.frame_step(onehot)
        assert img.shape == (288, 512, 3) and img.dtype == np.uint8
        assert r == pytest.approx(float(g[name + "_reward"][t])) and isinstance(r, (int, float))
        assert term == bool(g[name + "_terminal"][t]) and score == g[name + "_score"][t]
        want = g[name + "_state"][t + 1]
        assert (gs.playery, gs.playerVelY, gs.playerIndex, gs.score) == (want[0], want[1], want[2], want[5])
        assert [p["x"] for p in gs.upperPipes] == [x for x in want[7:10] if x != -9999]
        assert [p["y"] for p in gs.upperPipes] == list(want[10:10 + want[6]])
    # the global stream advanced exactly like the reference's: the next draw equals a twin's
    nxt = random.random()
    random.seed(seed)
    for _ in range(len(g[name + "_draws"]) if T == len(g[name + "_action"]) else 0):
        random.randint(0, 7)
    if T == len(g[name + "_action"]):
        assert nxt == random.random()
    with pytest.raises(ValueError, match="Multiple input actions!"):
        gs.frame_step(np.array([1, 1]))


def test_preprocess_kernel_matches_oracle(torch_cuda, oracle):
    from dqnflappybird_amd.FlappyBirdDQN import preprocess
    env = oracle.GameState(seed=2)
    for t in range(40):
        env.step(int(t % 8 == 0))
        if t % 5 == 0:
            rgb = env.render_full()
            out = preprocess(rgb)
            assert out.shape == (80, 80, 1) and np.array_equal(out[:, :, 0], oracle.preprocess(rgb))
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 256, (288, 512, 3), dtype=np.uint8)      # arbitrary image, not a game frame
    assert np.array_equal(preprocess(noise)[:, :, 0], oracle.preprocess(noise))


@pytest.mark.parametrize("model", ["dqn", "dqnnature", "ddqn", "duelingdqn", "prioritydqn"])
def test_driver_loop_runs_every_model(torch_cuda, model):
    """FlappyBirdDQN.py's loop end to end on the GPU for each --model (OBSERVE shortened)."""
    from dqnflappybird_amd import FlappyBirdDQN as drv
    cls = drv.model_class(model)
    old = cls.OBSERVE
    cls.OBSERVE = 40.
    try:
        random.seed(1)
        np.random.seed(1)
        brain = drv.playFlappyBird(model, steps=60, verbose=False)
    finally:
        cls.OBSERVE = old
    assert brain.timeStep == 60 and len(brain) == 60
    assert np.isfinite(brain.lost.item())
    assert brain.currentState.shape == (80, 80, 4)
    assert abs(brain.epsilon - (0.03 - 19 * 3e-8)) < 1e-12


def test_brain_on_gpu_tracks_cpu_backend(torch_cuda, oracle):
    """Same seeds, same frames: the HIP-backed BrainDQN and the oracle-backed one make the same
    decisions and reach the same loss (first train steps)."""
    from dqnflappybird_amd.BrainDQN import BrainDQN
    from tests.cpu_backend import CpuBackend
    from tests.test_brain_host_logic import frames_source
    outs = []
    for backend in (None, CpuBackend()):
        first, step_env = frames_source(oracle, 9)
        random.seed(5)
        b = BrainDQN(2, 'bird', backend=backend, verbose=False, seed=3)
        b.OBSERVE, b.BATCH_SIZE = 10., 8
        if backend is None:
            b.net.load_params(oracle.init_params(oracle.qcfg(), 3) * 3.0)
        else:
            b.net.p[0] = oracle.init_params(oracle.qcfg(), 3) * 3.0
        b.setInitState(first)
        acts, losses = [], []
        for _ in range(16):
            a = b.getAction()
            obs, r, t, s = step_env(int(a[1]))
            b.setPerception(obs, a, r, t, s)
            acts.append(int(a[1]))
            if b.lost is not None:
                losses.append(float(b.lost.item() if hasattr(b.lost, "item") else b.lost))
        outs.append((acts, losses))
    assert outs[0][0] == outs[1][0]
    np.testing.assert_allclose(outs[0][1][:2], outs[1][1][:2], rtol=1e-3)


@pytest.mark.parametrize("algo", ["dqn", "nature", "per"])
def test_vecbrain_device_resident_loop(torch_cuda, algo):
    from dqnflappybird_amd.vecbrain import VecBrain
    vb = VecBrain(64, algo=algo, capacity=20000, observe=5, seed=2)
    vb.run(80, log_every=0)
    assert vb.timeStep == 80 and len(vb.replay) == 80 * 64
    assert np.isfinite(vb.last_loss.item())
    # device-side episode counters (fb_env_set_stats_buffer): an untrained agent crashes within ~50 steps
    ep, ssum, smax, pipes = vb.stats.tolist()
    assert ep >= 64 and 0 <= smax <= ssum <= pipes and smax <= 2


def test_checkpoint_resume_on_device(torch_cuda, oracle, tmp_path):
    """save_checkpoint / _load_saved_parameters through the HIP backend (reference BrainDQN.py:176-192,227-233)."""
    from dqnflappybird_amd.BrainDQN import BrainDQN
    from tests.test_brain_host_logic import frames_source
    first, step_env = frames_source(oracle, 2)
    root = str(tmp_path / "saved_parameters")
    a = BrainDQN(2, 'bird', verbose=False, seed=5, save_root=root)
    a.OBSERVE, a.BATCH_SIZE = 8., 8
    a.setInitState(first)
    for _ in range(14):
        act = a.getAction()
        obs, r, t, s = step_env(int(act[1]))
        a.setPerception(obs, act, r, t, s)
    a.save_checkpoint()
    b = BrainDQN(2, 'bird', verbose=False, seed=99, save_root=root)
    assert torch_cuda.equal(a.net.store_params(0), b.net.store_params(0))
    assert torch_cuda.equal(a.net.adam_state()[0], b.net.adam_state()[0])
    assert (b.timeStep, b.epsilon, b.gameTimes) == (a.timeStep, a.epsilon, a.gameTimes)
    # the restored net computes the same Q (incl. the refreshed bf16 split of conv1)
    st = torch_cuda.from_numpy(a.currentState[None]).cuda()
    assert torch_cuda.equal(a.net.forward(st), b.net.forward(st))


def test_checkpoint_with_replay_memory_resumes_the_index_stream(torch_cuda, oracle, tmp_path):
    """save_replay=True keeps what the reference forgets (BrainDQN.py:176-192): replay memory, onlineTimeStep, frame stack and the
    generators.  The resumed Brain does not observe again and, fed the same frames, draws the same minibatch indices, takes the same
    actions and ends with the same parameters as the one that never stopped."""
    import random
    from dqnflappybird_amd.BrainDQN import BrainDQN
    from tests.test_brain_host_logic import frames_source
    root = str(tmp_path / "saved_parameters")

    def run(brain, env, n, log):
        for _ in range(n):
            act = brain.getAction()
            obs, r, t, s = env(int(act[1]))
            brain.setPerception(obs, act, r, t, s)
            log.append((int(act[1]), float(brain._be.host(brain.lost).reshape(-1)[0]) if brain.lost is not None else None))

    random.seed(4); np.random.seed(4)
    first, env_a = frames_source(oracle, 2)
    a = BrainDQN(2, 'bird', verbose=False, seed=5, save_root=root, save_replay=True)
    a.OBSERVE, a.BATCH_SIZE = 8., 8
    a.setInitState(first)
    log_a = []
    run(a, env_a, 14, log_a)
    a.save_checkpoint()
    tail_a = []
    run(a, env_a, 10, tail_a)
    # the same environment, replayed to the checkpoint's position for the resumed brain
    _, env_b = frames_source(oracle, 2)
    for act, _ in log_a:
        env_b(act)
    random.seed(999); np.random.seed(999)                                # the checkpoint restores the generators
    b = BrainDQN(2, 'bird', verbose=False, seed=99, save_root=root, save_replay=True)
    b.OBSERVE, b.BATCH_SIZE = 8., 8
    assert b.onlineTimeStep == 14 and len(b) == 14                       # no second OBSERVE phase
    tail_b = []
    run(b, env_b, 10, tail_b)
    assert tail_a == tail_b                                              # actions and losses step by step
    assert torch_cuda.equal(a.net.store_params(0), b.net.store_params(0))


def test_vecbrain_save_load_continues_bit_for_bit(torch_cuda, tmp_path):
    """VecBrain.save / load: the device-resident loop (256 envs) resumed from a checkpoint reproduces the actions, the sampled
    indices and the parameters of the run that never stopped -- env states, frame stacks, ring, sampler generator, Adam slots."""
    torch = torch_cuda
    from dqnflappybird_amd.vecbrain import VecBrain
    kw = dict(algo="nature", batch=32, capacity=20000, observe=6, seed=3, replace_target_iter=4)
    a = VecBrain(256, **kw)
    a.run(20, log_every=0)
    a.save(str(tmp_path / "ck"))
    eps_saved = a.epsilon
    ta = []
    for _ in range(12):
        a.step(); ta.append((a.one_step.actions.clone(), a.one_step.idx.clone(), a.one_step.loss.clone()))
    b = VecBrain(256, **dict(kw, seed=77))                               # different games, weights and generator until load()
    b.load(str(tmp_path / "ck"))
    assert (b.timeStep, b.onlineTimeStep, b.epsilon) == (20, 20, eps_saved)
    b.seed = a.seed                                                      # the acting stream's key is a constructor argument, not state
    for i in range(12):
        b.step()
        assert torch.equal(b.one_step.actions, ta[i][0]) and torch.equal(b.one_step.idx, ta[i][1]) and torch.equal(b.one_step.loss, ta[i][2]), i
    assert torch.equal(a.net.store_params(0), b.net.store_params(0)) and torch.equal(a.net.store_params(1), b.net.store_params(1))
    assert (a.env.get_state() == b.env.get_state()).all() and len(a.replay) == len(b.replay)


@pytest.mark.parametrize("N", [7, 64, 300, 301, 1027])      # < 256: the small-batch kernels; >= 256: the two-plane fp16 acting kernels
def test_nibble_state_equals_current_state_and_act_nib_is_bit_identical(torch_cuda, N):
    """The env kernel's running 4-frame nibble state == the replay ring's currentState (BrainDQN.py:68,238-239),
    and the acting forward that consumes it gives bit-identical Q / actions to the u8 path.  (>= 256 states the nibble path is the
    FUSED trunk -- conv1 + conv2 + conv3 in one launch, four states per workgroup, conv1's output handed over in LDS -- and the u8
    path the conv1 launch + the five-states-per-workgroup conv2 / conv3 launch with fp16 planes in HBM between them: same bits.
    301 / 1027: a last workgroup with one / three states.)"""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay
    env, rep, net = VecGameState(N, seed=3), VecReplay(5000, N), QNet(max_batch=max(N, 8))
    net.init_params(seed=1)
    nib = env.track_state()
    env.observe()
    rep.reset(env.frame_bits)

    def unpack(nibt):
        from dqnflappybird_amd import _lib as L                     # [N][FB_NIB_STRIDE]: 84 padded rows of 4 + 40 bytes
        raw = nibt.cpu().numpy()
        full = raw[:, :L.NIB_ROWS * L.NIB_PITCH].reshape(N, L.NIB_ROWS, L.NIB_PITCH)
        assert not full[:, :2].any() and not full[:, 82:].any() and not full[:, :, :4].any()      # conv1's zero padding
        assert not raw[:, L.NIB_ROWS * L.NIB_PITCH:].any()
        b = np.ascontiguousarray(full[:, 2:82, 4:])
        px = np.stack([b & 0x0F, b >> 4], axis=-1).reshape(N, 6400)   # nibble per pixel
        return (((px[..., None] >> np.arange(4)) & 1) * 255).astype(np.uint8).reshape(N, 80, 80, 4)

    rng = np.random.default_rng(0)
    for t in range(12):
        states = rep.current_state()
        assert np.array_equal(unpack(nib), states.cpu().numpy()), t
        a0, q0 = net.act(states, 0.0, want_q=True)
        a0, q0 = a0.clone(), q0.clone()
        a1, q1 = net.act_nib(nib, 0.0, want_q=True)
        assert torch.equal(q0, q1) and torch.equal(a0, a1)
        acts = torch.from_numpy((rng.random(N) < 0.2).astype(np.uint8)).cuda()
        env.frame_step(acts, want_u8=False)
        rep.push(env.frame_bits, acts, env.reward, env.terminal)


@pytest.mark.parametrize("N,steps", [(256, 40), (1024, 30), (2304, 24)])      # 1024 envs / batch 32: the bench's own workload
def test_vec_step_single_call_equals_separate_calls(torch_cuda, N, steps):
    """fb_vec_step (one host call per step; head, random.sample and the Memory append riding in the env launch) ==
    act_nib -> frame_step -> push_sample -> gather -> train_step: same actions, env states, sampled indices and
    parameters after `steps` steps (8 of them in the OBSERVE phase).  2304 envs: more envs than env workgroups (the
    head keeps its own launch, the env kernel strides) and a ring that wraps."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
    B = 32

    def make():
        env, rep, net = VecGameState(N, seed=5), VecReplay(20000, N), QNet(max_batch=N)
        rep.seed(9, "cpython"); net.init_params(3)
        nib = env.track_state(); env.observe(); rep.reset(env.frame_bits)
        return env, rep, net, nib

    e1, r1, n1, nib1 = make()
    e2, r2, n2, nib2 = make()
    one = VecStep(e2, r2, n2, B, "dqn")
    for step in range(steps):
        train = step >= 8
        a1 = n1.act_nib(nib1, 0.05, seed=1, step=step)
        e1.frame_step(a1, want_u8=False)
        if train:
            idx = r1.push_sample(e1.frame_bits, a1, e1.reward, e1.terminal, B)
            s, a, r, s2, t = r1.gather(idx)
            loss, _, _ = n1.train_step("dqn", s, a, r, s2, t, want_aux=False)
        else:
            r1.push(e1.frame_bits, a1, e1.reward, e1.terminal)
        a2 = one(0.05, seed=1, step=step, train=train)
        assert torch.equal(a1, a2)
        if train:
            assert torch.equal(idx, one.idx) and torch.equal(loss, one.loss)
    assert (e1.get_state() == e2.get_state()).all()
    assert torch.equal(n1.store_params(), n2.store_params())
    assert torch.equal(nib1, nib2)
    # the split schedule (train chain on the caller's stream beside acting + env on the net's side stream) ran where it applies, and
    # BOTH of its cases did: minibatches that started beside their env step and ones that had to wait for it (a transition of the
    # step itself was drawn); split_stats raises if any of the bounded waits between the chains gave up
    issued, clean = n2.split_stats()
    assert issued == steps - 8
    if N <= 1024:
        assert 0 < clean < issued, (issued, clean)  # (2304 envs in a 20 000-slot memory: a clean draw is a 2 % event)
    # the ring the riders filled holds what the push kernel stores: oldest, newest and a stride of positions in between
    assert len(r1) == len(r2)
    probe = torch.cat([torch.arange(0, 48), torch.arange(100, len(r1) - 48, 331), torch.arange(len(r1) - 48, len(r1))]).cuda()
    for x, y in zip(r1.gather(probe), r2.gather(probe)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("N,algo,B,steps,dtype", [(1024, "dqn", 32, 400, "f32"), (512, "double", 64, 200, "f32"), (4096, "nature", 32, 120, "f32"),
                                                   (1024, "nature", 32, 150, "bf16")])
def test_split_schedule_equals_one_stream_over_many_steps(torch_cuda, N, algo, B, steps, dtype):
    """fb_vec_step's split schedule (train chain on the caller's stream beside acting + env on the net's side stream, handed over through
    device words) against the SAME loop kept on one stream (fb_vec_step_set_schedule(0)), two pipelines stepped alternately: actions,
    indices and loss at every step, parameters / Adam slots / env states / ring contents at the end, bit for bit -- over enough steps
    that both kinds of minibatch occur many times (started beside the env step; waited for it because a transition of the step itself
    was drawn), with target syncs on the caller's stream in between (ordered against the train chain by stream order alone)."""
    torch = torch_cuda
    from dqnflappybird_amd import _lib as L
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep

    def make():
        env, rep, net = VecGameState(N, seed=11), VecReplay(60000, N), QNet(max_batch=max(N, B))
        rep.seed(4, "cpython"); net.init_params(5, which=0); net.init_params(6, which=1)
        if dtype == "bf16":                                 # (bf16 acting + training: the one-plane instantiations of both trunks)
            net.set_inference_dtype("bf16"); net.set_train_dtype("bf16")
        nib = env.track_state(); env.observe(); rep.reset(env.frame_bits)
        return env, rep, net, nib, VecStep(env, rep, net, B, algo)

    e1, r1, n1, nib1, one = make()
    e2, r2, n2, nib2, two = make()
    try:
        for step in range(steps):
            train = step >= 4 and step % 9 != 5            # (a step without training now and then: it takes the one-stream path, in between split ones)
            if algo != "dqn" and train and step % 25 == 0:
                n1.sync_target(); n2.sync_target()
            L.check(L.lib().fb_vec_step_set_schedule(0), "schedule")
            a1 = one(0.05, seed=2, step=step, train=train).clone()
            L.check(L.lib().fb_vec_step_set_schedule(1), "schedule")
            a2 = two(0.05, seed=2, step=step, train=train)
            assert torch.equal(a1, a2), step
            if train:
                assert torch.equal(one.idx, two.idx) and torch.equal(one.loss, two.loss), step
    finally:
        L.check(L.lib().fb_vec_step_set_schedule(1), "schedule")
    assert (e1.get_state() == e2.get_state()).all() and torch.equal(nib1, nib2)
    assert torch.equal(n1.store_params(), n2.store_params()) and torch.equal(n1.store_params(1), n2.store_params(1))
    (m1, v1, p1), (m2, v2, p2) = n1.adam_state(), n2.adam_state()
    assert torch.equal(m1, m2) and torch.equal(v1, v2) and np.array_equal(p1, p2)
    assert np.array_equal(r1.state_blob(), r2.state_blob())
    assert n1.split_stats() == (0, 0)
    issued, clean = n2.split_stats()                      # (raises if a wait between the two streams gave up)
    assert issued == sum(1 for k in range(steps) if k >= 4 and k % 9 != 5) and 0 < clean < issued, (issued, clean)


@pytest.mark.parametrize("algo,B,dtype", [("dqn", 32, "f32"), ("nature", 17, "f32"), ("double", 64, "f32"), ("double", 256, "f32"),
                                          ("nature", 32, "bf16"), ("double", 160, "bf16")])
def test_train_from_replay_equals_gather_plus_train_step(torch_cuda, algo, B, dtype):
    """fb_train_from_replay (the conv trunk reads the sampled transitions' 1-bit frames in the ring; no gathered copies) ==
    fb_replay_gather + fb_qnet_train_step, bit for bit below 256 samples: a / r / t, loss, parameters and Adam slots over several steps, on a ring
    that has wrapped, with indices at both ends of the deque, right after a target sync and after a stand-alone Adam (stale planes)."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, train_from_replay
    N = 64
    env, rep = VecGameState(N, seed=3), VecReplay(1500, N)
    env.observe(); rep.reset(env.frame_bits)
    rng = np.random.default_rng(0)
    for _ in range(40):                                   # 2560 transitions through a 1500-slot memory: the ring wraps
        acts = torch.from_numpy((rng.random(N) < 0.15).astype(np.uint8)).cuda()
        env.frame_step(acts, want_u8=False)
        rep.push(env.frame_bits, acts, env.reward, env.terminal)
    n1, n2 = QNet(max_batch=B), QNet(max_batch=B)
    for n in (n1, n2):
        n.init_params(7); n.sync_target(); n.set_train_dtype(dtype)     # (bf16: the one-plane instantiations of the same kernels)
    size = len(rep)
    # B = 256: the separate calls run the large-batch kernels (another summation order in conv2 / conv3), so equal to rounding only
    same = (lambda x, y: torch.equal(x, y)) if B < 256 else (lambda x, y: torch.allclose(x, y, rtol=2e-4, atol=2e-6))
    for step in range(6):
        idx = torch.from_numpy(rng.integers(0, size, B)).cuda()
        idx[0], idx[1] = 0, size - 1                      # oldest and newest transition
        s, a, r, s2, t = rep.gather(idx)
        l1, _, _ = n1.train_step(algo, s, a, r, s2, t, want_aux=False)
        l2, a2, r2, t2 = train_from_replay(rep, n2, algo, idx)
        assert torch.equal(a, a2) and torch.equal(r, r2) and torch.equal(t, t2)
        assert same(l1, l2), (step, l1, l2)
        if step == 2:
            n1.sync_target(); n2.sync_target()
        if step == 3:                                     # a stand-alone Adam leaves the planes stale: the call has to notice
            g = torch.full((n1.n_params,), 1e-3, device="cuda")
            n1.apply_adam(g); n2.apply_adam(g)
    assert same(n1.store_params(), n2.store_params())
    m1, v1, _ = n1.adam_state(); m2, v2, _ = n2.adam_state()
    assert same(m1, m2) and same(v1, v2)
    with pytest.raises(Exception):
        train_from_replay(rep, n2, algo, torch.zeros(257, dtype=torch.int64, device="cuda"))


@pytest.mark.parametrize("algo,dueling,B", [("dqn", False, 32), ("double", True, 32), ("per", False, 32)])
def test_ring_fed_train_step_gradients_match_oracle(torch_cuda, oracle, algo, dueling, B):
    """The ring-fed train step (fb_train_from_replay: conv trunk straight from the 1-bit frame ring -- what fb_vec_step and
    fb_train_steps launch) against the ORACLE directly, not through its equality with gather + train_step: the oracle gets the u8
    states the gather kernel expands from the same ring (themselves tied to the DequeModel in test_gpu_replay.py), the device never
    builds them.  Targets within 1e-4, every gradient tensor within 2e-3 of the fp64-accumulating oracle."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, train_from_replay
    from tests.test_gpu_qnet import oracle_train_grads, trained_like_params
    N = 64
    env, rep = VecGameState(N, seed=11), VecReplay(4000, N)
    env.observe(); rep.reset(env.frame_bits)
    rng = np.random.default_rng(5)
    for _ in range(50):
        acts = torch.from_numpy((rng.random(N) < 0.12).astype(np.uint8)).cuda()
        env.frame_step(acts, want_u8=False)
        rep.push(env.frame_bits, acts, env.reward, env.terminal)
    cfg = oracle.qcfg(512, 2, dueling)
    p_on, p_tg = trained_like_params(oracle, cfg, 1), trained_like_params(oracle, cfg, 2)
    net = QNet(2, 512, "dueling" if dueling else "plain", max_batch=B)
    net.load_params(p_on, 0); net.load_params(p_tg, 1)
    size = len(rep)
    for attempt in range(60):                            # kink-free minibatch (see test_train_step_gradients_match_oracle)
        idx = torch.from_numpy(rng.integers(0, size, B)).cuda()
        s, a, r, s2, t = (x.cpu().numpy() for x in rep.gather(idx))
        oracle.forward(p_on, cfg, s)
        if oracle.last_margin(nonzero=True) > 2e-5:      # (game frames: pool windows over identical pixels tie exactly, in every arithmetic)
            break
    else:
        pytest.fail("no kink-free batch found")
    isw = rng.random(B).astype(np.float32) if algo == "per" else None
    iswd = None if isw is None else torch.from_numpy(isw).cuda()
    grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda")
    out = train_from_replay(rep, net, algo, idx, flat_grad=grad, isw=iswd, want_abs_err=True)
    loss, ae = out[0], out[4]
    y0, loss0, ae0, g0 = oracle_train_grads(oracle, cfg, p_on, p_tg, algo, s, a, r, s2, t, isw)
    np.testing.assert_allclose(loss.item(), loss0, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ae.cpu().numpy(), ae0, rtol=0, atol=2e-4)
    g = grad.cpu().numpy()
    bounds = [0, 8192, 8224, 40992, 41056, 77920, 77984, 77984 + 1600 * 512, 77984 + 1600 * 512 + 512, net.n_params]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        scale = np.abs(g0[lo:hi]).max()
        assert scale > 0
        np.testing.assert_allclose(g[lo:hi], g0[lo:hi], rtol=2e-3, atol=2e-5 * scale, err_msg=f"params[{lo}:{hi}]")
    assert np.array_equal(net.store_params().cpu().numpy(), p_on)          # gradient-only mode


@pytest.mark.parametrize("N,steps,algo", [(256, 40, "dqn"), (2304, 24, "nature")])
def test_vec_step_data_parallel_path_equals_fused(torch_cuda, N, steps, algo):
    """The N > 1 hot path at world size 1 (bench.py full_step / VecBrain.step): fb_vec_step(flat_grad = g) followed by
    fb_qnet_apply_adam(g) == fb_vec_step() with Adam fused, bit for bit and step by step -- actions (so the acting forward
    saw the weights the stand-alone Adam wrote, re-split into bf16 planes), sampled indices, loss -- and at the end the
    parameters, both Adam slots and beta powers, the env states and the agents' frame stacks."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
    B = 32

    def make(dp):
        env, rep, net = VecGameState(N, seed=5), VecReplay(20000, N), QNet(max_batch=N)
        rep.seed(9, "cpython"); net.init_params(3, which=0); net.init_params(4, which=1)
        nib = env.track_state(); env.observe(); rep.reset(env.frame_bits)
        grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda") if dp else None
        return env, rep, net, nib, grad, VecStep(env, rep, net, B, algo, flat_grad=grad)

    e1, r1, n1, nib1, _, fused = make(False)
    e2, r2, n2, nib2, grad, split = make(True)
    for step in range(steps):
        train = step >= 8
        if train and algo == "nature" and step % 5 == 0:
            n1.sync_target(); n2.sync_target()
        a1 = fused(0.05, seed=1, step=step, train=train)
        a2 = split(0.05, seed=1, step=step, train=train)
        if train:
            n2.apply_adam(grad)
        assert torch.equal(a1, a2), step
        if train:
            assert torch.equal(fused.idx, split.idx) and torch.equal(fused.loss, split.loss), step
    assert (e1.get_state() == e2.get_state()).all() and torch.equal(nib1, nib2)
    assert torch.equal(n1.store_params(), n2.store_params()) and torch.equal(n1.store_params(1), n2.store_params(1))
    (m1, v1, p1), (m2, v2, p2) = n1.adam_state(), n2.adam_state()
    assert torch.equal(m1, m2) and torch.equal(v1, v2) and np.array_equal(p1, p2)
    assert p1[0] < 0.9 ** (steps - 8)                                    # and the optimizer really stepped steps - 8 times


@pytest.mark.parametrize("dp,N,steps", [(False, 256, 30), (True, 256, 30), (False, 4096, 14)])
def test_vec_step_prioritized_equals_the_separate_calls(torch_cuda, dp, N, steps):
    """fb_vec_step on a prioritized memory (BrainPrioritizedReplyDQN.py:277-329 for N envs in one host call: act -> env -> Memory.store ->
    Memory.sample -> weighted train -> Memory.batch_update) == the separate calls, bit for bit: actions, tree indices, importance
    weights, loss and |TD errors| step by step, parameters, the raw tree bytes and beta at the end (reference-order tree)."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep, train_from_replay
    # (4096 envs: Memory.batch_update runs ahead on the memory's side stream as well -- from that env count on it is the default)
    B = 32

    def make():
        env, rep, net = VecGameState(N, seed=5), VecReplay(32 * N, N, prioritized=True), QNet(max_batch=N)
        rep.seed(9, "numpy"); net.init_params(3, which=0); net.init_params(4, which=1)
        nib = env.track_state(); env.observe(); rep.reset(env.frame_bits)
        return env, rep, net, nib

    e1, r1, n1, nib1 = make()
    e2, r2, n2, nib2 = make()
    # dp: the data-parallel form of the same step -- the gradient is exported, the priorities are updated from the local |TD errors|
    # inside the call, Adam follows from outside (after the all-reduce a multi-rank run would put in between)
    grad = torch.zeros(n2.n_params, dtype=torch.float32, device="cuda") if dp else None
    one = VecStep(e2, r2, n2, B, "per", flat_grad=grad)
    for step in range(steps):
        train = step >= 8
        a1 = n1.act_nib(nib1, 0.05, seed=1, step=step)
        e1.frame_step(a1, want_u8=False)
        r1.push(e1.frame_bits, a1, e1.reward, e1.terminal)
        if train:
            idx, isw = r1.sample(B)
            loss, a_, r_, t_, ae = train_from_replay(r1, n1, "per", idx, isw=isw, want_abs_err=True)
            r1.update_priorities(idx, abs_err=ae)
        a2 = one(0.05, seed=1, step=step, train=train)
        if train and dp:
            n2.apply_adam(grad)
        assert torch.equal(a1, a2), step
        if train:
            assert torch.equal(idx, one.idx) and torch.equal(isw, one.isw), step
            # (|TD errors|: the stand-alone batch_update adds its 0.01 in place, :147; inside fb_vec_step it runs ahead on the memory's
            #  side stream and leaves the caller's array as the loss wrote it)
            assert torch.equal(loss, one.loss) and torch.equal(ae, one.abs_err + 0.01), step      # (fb_vec_step leaves |TD error| as the loss wrote it, whichever form its batch_update took; the stand-alone call adds the reference's 0.01 in place)
    assert (e1.get_state() == e2.get_state()).all() and torch.equal(nib1, nib2)
    assert torch.equal(n1.store_params(), n2.store_params())
    b1, b2 = r1.state_blob(), r2.state_blob()
    assert np.array_equal(np.asarray(b1), np.asarray(b2))             # ring, rows, counters, generator, SumTree + max / min heaps, beta
    with pytest.raises(ValueError):
        VecStep(e2, r2, n2, B, "dqn")                                  # a uniform algo on a prioritized memory


def test_train_from_replay_prioritized_equals_the_separate_calls(torch_cuda):
    """The prioritized step through fb_train_from_replay (SumTree leaf indices, importance weights in, |TD errors| out) ==
    fb_replay_gather + fb_qnet_train_step(isw) + the same priorities update, bit for bit over several sample / train / update rounds."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, train_from_replay
    N, B = 64, 32
    env = VecGameState(N, seed=3); env.observe()
    reps = [VecReplay(4096, N, prioritized=True) for _ in range(2)]
    for rp in reps:
        rp.seed(5, "numpy"); rp.reset(env.frame_bits)
    rng = np.random.default_rng(0)
    for _ in range(30):
        acts = torch.from_numpy((rng.random(N) < 0.15).astype(np.uint8)).cuda()
        env.frame_step(acts, want_u8=False)
        for rp in reps:
            rp.push(env.frame_bits, acts, env.reward, env.terminal)
    n1, n2 = QNet(max_batch=B), QNet(max_batch=B)
    for n in (n1, n2):
        n.init_params(7); n.sync_target()
    for step in range(5):
        (i1, w1), (i2, w2) = reps[0].sample(B), reps[1].sample(B)
        assert torch.equal(i1, i2) and torch.equal(w1, w2)
        s, a, r, s2, t = reps[0].gather(i1)
        l1, ae1, _ = n1.train_step("per", s, a, r, s2, t, isw=w1, want_aux=True)
        l2, a2, r2, t2, ae2 = train_from_replay(reps[1], n2, "per", i2, isw=w2, want_abs_err=True)
        assert torch.equal(a, a2) and torch.equal(r, r2) and torch.equal(t, t2)
        assert torch.equal(l1, l2) and torch.equal(ae1, ae2), step
        reps[0].update_priorities(i1, abs_err=ae1); reps[1].update_priorities(i2, abs_err=ae2)
    assert torch.equal(n1.store_params(), n2.store_params())


def test_overlapped_all_reduce_equals_the_plain_data_parallel_step(torch_cuda):
    """dist.OverlappedAllReduce (the gradient's W_fc1 / head part reduced on a side stream that waits for the event fb_vec_step records
    behind its fc1 backward launch, the conv part on the step's stream, then a join) == one all-reduce of the whole vector, through a
    real one-rank RCCL group: actions, indices, loss step by step, parameters and Adam slots at the end.  Also checks that the event
    really sits where the header says: the side stream's copy of the tail, taken at the event, already holds the final values."""
    torch = torch_cuda
    import os
    import torch.distributed as tdist
    from dqnflappybird_amd.dist import OverlappedAllReduce
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
    N, B, steps = 512, 32, 24
    created = not tdist.is_initialized()
    if created:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
        tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        def make():
            env, rep, net = VecGameState(N, seed=5), VecReplay(20000, N), QNet(max_batch=N)
            rep.seed(9, "cpython"); net.init_params(3, which=0); net.init_params(4, which=1)
            env.track_state(); env.observe(); rep.reset(env.frame_bits)
            grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda")
            return env, rep, net, grad, VecStep(env, rep, net, B, "nature", flat_grad=grad)

        e1, r1, n1, g1, plain = make()
        e2, r2, n2, g2, over = make()
        red = OverlappedAllReduce(n2, g2, mean_loss=True, force=True)
        assert red.on and red.tail.numel() > 10 * red.front.numel()
        snap = torch.zeros_like(red.tail)
        for step in range(steps):
            train = step >= 8
            a1 = plain(0.05, seed=1, step=step, train=train)
            if train:
                tdist.all_reduce(g1)
                n1.apply_adam(g1)
            a2 = over(0.05, seed=1, step=step, train=train)
            if train:
                red.side.wait_event(red.event)
                with torch.cuda.stream(red.side):
                    snap.copy_(red.tail)                      # what a side stream sees AT the event
                red()
                n2.apply_adam(g2)
                torch.cuda.current_stream().wait_stream(red.side)
                assert torch.equal(snap, g1[red.front.numel():]), step
            assert torch.equal(a1, a2), step
            if train:
                assert torch.equal(plain.idx, over.idx) and torch.equal(plain.loss, over.loss), step
        assert torch.equal(n1.store_params(), n2.store_params())
        (m1, v1, p1), (m2, v2, p2) = n1.adam_state(), n2.adam_state()
        assert torch.equal(m1, m2) and torch.equal(v1, v2) and np.array_equal(p1, p2)
        red.close()
    finally:
        if created:
            tdist.destroy_process_group()


@pytest.mark.parametrize("N,algo,overlap", [(512, "dqn", True), (256, "nature", True), (256, "dqn", False)])
def test_native_dp_step_equals_the_plain_data_parallel_step(torch_cuda, N, algo, overlap):
    """fb_vec_step_dp (the step, the all-reduce through the library's OWN RCCL communicator -- in line, or in two pieces: tail on a side
    stream behind the gradient-ready event, front on the step's stream -- and Adam, one host call) == fb_vec_step(flat_grad) + fb_qnet_apply_adam at world
    size 1, bit for bit: actions, indices and loss step by step, parameters / Adam slots / env states at the end.  (One rank: the
    collective is RCCL's one-rank copy; what this pins is the choreography -- events, stream order, nothing read before it is final.)"""
    torch = torch_cuda
    from dqnflappybird_amd.dist import NativeDP
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
    B, steps = 32, 30
    nd = NativeDP(rank=0, world=1, overlap=overlap)
    try:
        def make(native):
            env, rep, net = VecGameState(N, seed=5), VecReplay(20000, N), QNet(max_batch=N)
            rep.seed(9, "cpython"); net.init_params(3, which=0); net.init_params(4, which=1)
            env.track_state(); env.observe(); rep.reset(env.frame_bits)
            grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda")
            return env, rep, net, grad, VecStep(env, rep, net, B, algo, flat_grad=grad, dist=nd if native else None, mean_loss=algo != "dqn")

        e1, r1, n1, g1, plain = make(False)
        e2, r2, n2, g2, native = make(True)
        for step in range(steps):
            train = step >= 8
            if train and algo == "nature" and step % 5 == 0:
                n1.sync_target(); n2.sync_target()
            a1 = plain(0.05, seed=1, step=step, train=train)
            if train:
                n1.apply_adam(g1)
            a2 = native(0.05, seed=1, step=step, train=train)
            assert torch.equal(a1, a2), (step, n1.split_stats(), n2.split_stats())
            if train:
                assert torch.equal(plain.idx, native.idx) and torch.equal(plain.loss, native.loss), step
                assert torch.equal(g1, g2), step
        assert (e1.get_state() == e2.get_state()).all()
        assert torch.equal(n1.store_params(), n2.store_params())
        (m1, v1, p1), (m2, v2, p2) = n1.adam_state(), n2.adam_state()
        assert torch.equal(m1, m2) and torch.equal(v1, v2) and np.array_equal(p1, p2)
        assert p1[0] < 0.9 ** (steps - 9)
    finally:
        torch.cuda.synchronize()
        nd.close()


def test_graph_replayed_train_steps_refresh_the_acting_weights(torch_cuda):
    """A train step replayed from a captured hipGraph changes the parameters without the host handle noticing: the acting
    forward on >= 256 states (bf16 hi/mid/lo split of the weights) must still re-split them.  Staleness is decided on the
    device (AdamDev::pver / wver), so: capture TrainSteps -> act live -> replay -> act == the same sequence run eagerly, bit for
    bit; and an acting forward captured while the split was fresh re-splits when it is replayed after the parameters moved."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, TrainSteps, VecGameState, VecReplay
    N, B = 256, 32

    def make():
        env, rep, net = VecGameState(N, seed=5), VecReplay(20000, N), QNet(max_batch=N)
        rep.seed(9, "cpython"); net.init_params(3); net.set_hparams(lr=1e-3)     # steps large enough to move Q visibly
        nib = env.track_state(); env.observe(); rep.reset(env.frame_bits)
        acts = (torch.rand(N, generator=torch.Generator().manual_seed(1)) < 0.1).to(torch.uint8).cuda()
        for _ in range(12):
            env.frame_step(acts, want_u8=False)
            rep.push(env.frame_bits, acts, env.reward, env.terminal)
        ts = TrainSteps(rep, net, B, "dqn")
        ts(1)                                                            # warm-up: every kernel loaded before any capture
        torch.cuda.synchronize()
        return env, rep, net, nib, ts

    # eager
    _, _, n1, nib1, ts1 = make()
    q1a = n1.act_nib(nib1, 0.0, want_q=True)[1].clone()
    ts1(2)
    q1b = n1.act_nib(nib1, 0.0, want_q=True)[1].clone()
    ts1(2)
    q1c = n1.act_nib(nib1, 0.0, want_q=True)[1].clone()
    # graphs
    _, _, n2, nib2, ts2 = make()
    g_train, g_act = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_train):
        ts2(2)
    q2a = n2.act_nib(nib2, 0.0, want_q=True)[1].clone()                  # live act: the split is fresh from here on
    g_train.replay()                                                     # parameters move on the device only
    q2b = n2.act_nib(nib2, 0.0, want_q=True)[1].clone()
    with torch.cuda.graph(g_act):                                        # captured while the split is fresh
        q_cap = n2.act_nib(nib2, 0.0, want_q=True)[1]
    g_train.replay()
    g_act.replay()
    torch.cuda.synchronize()
    q2c = q_cap.clone()
    assert torch.equal(q1a, q2a) and torch.equal(q1b, q2b) and torch.equal(q1c, q2c)
    assert not torch.equal(q1a, q1b) and not torch.equal(q1b, q1c)       # the comparison is not vacuous
    assert torch.equal(n1.store_params(), n2.store_params())
    m1, v1, p1 = n1.adam_state(); m2, v2, p2 = n2.adam_state()
    assert torch.equal(m1, m2) and torch.equal(v1, v2) and np.array_equal(p1, p2)


def test_vec_step_rejects_bad_arguments_before_anything_moves(torch_cuda):
    """fb_vec_step validates everything up front: a rejected call leaves the replay's push counter, the env and the
    network exactly as they were (it used to count the push before the train step refused the batch)."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay, VecStep
    N = 64
    env, rep, net = VecGameState(N, seed=5), VecReplay(5000, N), QNet(max_batch=N)
    rep.seed(9, "cpython"); net.init_params(3)
    env.track_state(); env.observe(); rep.reset(env.frame_bits)
    good, bad = VecStep(env, rep, net, 32, "dqn"), VecStep(env, rep, net, 300, "dqn")      # 300 > MAXTB = 256 and > max_batch
    for step in range(3):
        good(0.0, step=step, train=False)
    size, state, params = len(rep), env.get_state().copy(), net.store_params().clone()
    with pytest.raises(ValueError):
        bad(0.0, step=3, train=True)
    assert len(rep) == size and (env.get_state() == state).all() and torch.equal(net.store_params(), params)
    good(0.0, step=3, train=True)                                        # and the pipeline carries on
    assert len(rep) == size + N


def test_standalone_dueling_recipe_on_device(torch_cuda, tmp_path, monkeypatch):
    """the reference's stand-alone BrainDuelingDQN.py loop end to end on the HIP backend: drop-in game module,
    preprocess kernel, HBM replay, dueling net, checkpoints."""
    import os
    from dqnflappybird_amd import BrainDuelingDQN as R
    monkeypatch.setattr(R, "OBSERVE", 40.)
    monkeypatch.setattr(R, "SAVER_ITER", 25)
    monkeypatch.setattr(R, "REPLACE_TARGET_ITER", 10)
    random.seed(2)
    handles = R.createNetwork(seed=3)
    net = handles[0]
    p0, t0 = net.get_params(0), net.get_params(1)
    sp = str(tmp_path / "saved") + "/"
    t, eps, obs = R.trainNetwork(*handles, sess=None, max_steps=60, verbose=False, save_path=sp, save_back_path=str(tmp_path / "back") + "/")
    assert t == 60 and obs.shape == (80, 80, 4) and set(np.unique(obs)) <= {0, 255}
    assert sorted(os.listdir(sp)) == ["bird-dqn-25.npz", "bird-dqn-50.npz", "checkpoint"]
    assert not np.array_equal(net.get_params(0), p0)                       # 19 train steps happened
    z = np.load(sp + "bird-dqn-50.npz")              # saved when t became 50, before the first sync (iteration t = 50 > OBSERVE)
    assert np.array_equal(z["target"], t0) and not np.array_equal(net.get_params(1), t0)
    q = net.q_values(obs)
    assert q.shape == (2,) and np.isfinite(q).all()


def test_train_steps_single_call_equals_separate_calls(torch_cuda):
    """fb_train_steps (n x sample -> gather -> train in one call, the next step's random.sample riding in the conv3 backward
    launch) == the separate calls: same index stream, same parameters."""
    torch = torch_cuda
    from dqnflappybird_amd import _lib as L
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay
    N, B = 256, 32

    def make():
        env, rep, net = VecGameState(N, seed=5), VecReplay(20000, N), QNet(max_batch=N)
        rep.seed(9, "cpython"); net.init_params(3)
        env.observe(); rep.reset(env.frame_bits)
        acts = (torch.rand(N, generator=torch.Generator().manual_seed(1)) < 0.1).to(torch.uint8).cuda()
        for _ in range(12):
            env.frame_step(acts, want_u8=False)
            rep.push(env.frame_bits, acts, env.reward, env.terminal)
        return rep, net

    r1, n1 = make()
    r2, n2 = make()
    seen = []
    for _ in range(7):
        idx, _ = r1.sample(B)
        seen.append(idx.clone())
        s, a, r, s2, t = r1.gather(idx)
        n1.train_step("dqn", s, a, r, s2, t, want_aux=False)
    idx2 = torch.zeros(2 * B, dtype=torch.int64, device="cuda")
    s, a, r, s2, t = r2.gather(idx2[:B].contiguous())                    # buffers of the right shapes
    loss = torch.zeros(1, device="cuda")
    for n in (3, 4):                                                     # two calls: the stream continues across them
        L.check(L.lib().fb_train_steps(r2.h, n2.h, 0, B, n, L.ptr(idx2), L.ptr(s), L.ptr(s2), L.ptr(a), L.ptr(r), L.ptr(t), L.ptr(loss),
                                       0.99, L.current_stream()), "fb_train_steps")
        last = n - 1
        want = seen[2] if n == 3 else seen[6]
        assert torch.equal(idx2[(last & 1) * B:(last & 1) * B + B], want)
    assert torch.equal(n1.store_params(), n2.store_params())
    nxt1, _ = r1.sample(B); nxt2, _ = r2.sample(B)                       # and the generators are in the same place
    assert torch.equal(nxt1, nxt2)
