"""HIP env kernel (through the C ABI) vs the oracle and the reference-derived trajectories."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TAPES = ["traj0_never", "traj1_always", "traj2_every7", "traj3_every9", "traj4_random10",
         "traj5_random50", "traj6_seek", "traj7_seek"]


def dev_state_to_snapshot(st):
    """i32[16] device layout -> the oracle/fixture snapshot layout."""
    out = np.zeros(16, np.int32)
    out[0:7] = st[0:7]
    n = st[6]
    for i in range(3):
        live = i < n
        gy = 100 + 10 * st[10 + i]
        out[7 + i] = st[7 + i] if live else -9999
        out[10 + i] = gy - 320 if live else 0
        out[13 + i] = gy + 100 if live else 0
    return out


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def test_reference_trajectories_all_tapes_in_one_launch(torch_cuda, golden, oracle):
    """All 8 reference tapes run side by side as 8 envs of one VecGameState: state, reward,
    terminal, score bit-exact vs the reference's own game module; frames bit-exact vs the oracle."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecGameState
    g = golden("game_trajectories.npz")
    T = max(len(g[n + "_action"]) for n in TAPES)
    D = max(len(g[n + "_draws"]) for n in TAPES)
    env = VecGameState(len(TAPES), seed=0)
    tape = np.zeros((len(TAPES), D), np.int8)
    for i, n in enumerate(TAPES):
        d = g[n + "_draws"]
        tape[i, :len(d)] = d
    env.set_gap_tape(tape)
    env.reset()
    st = env.get_state()
    for i, n in enumerate(TAPES):
        st[i, 13] = int(g[n + "_cyc0"][0])
    env.set_state(st)
    orc = [oracle.GameState(tape=g[n + "_draws"], cyc_pos=int(g[n + "_cyc0"][0])) for n in TAPES]
    for i, n in enumerate(TAPES):
        assert np.array_equal(dev_state_to_snapshot(env.get_state()[i]), g[n + "_state"][0][:16])
    alive = [True] * len(TAPES)
    cursor_at_end = {}
    frame_checks = 0
    for t in range(T):
        acts = np.zeros(len(TAPES), np.uint8)
        for i, n in enumerate(TAPES):
            a = g[n + "_action"]
            alive[i] = t < len(a)
            acts[i] = a[t] if alive[i] else 0
        frames, rew, term, score = env.frame_step(torch.from_numpy(acts).cuda())
        st = env.get_state()
        rew, term, score = rew.cpu().numpy(), term.cpu().numpy(), score.cpu().numpy()
        fr = frames.cpu().numpy() if (t % 7 == 0 or term.any()) else None
        bits = env.frame_bits.cpu().numpy() if fr is not None else None
        for i, n in enumerate(TAPES):
            if not alive[i]:
                continue
            assert rew[i] == g[n + "_reward"][t], (n, t)
            assert term[i] == g[n + "_terminal"][t], (n, t)
            assert score[i] == g[n + "_score"][t], (n, t)
            assert np.array_equal(dev_state_to_snapshot(st[i]), g[n + "_state"][t + 1][:16]), (n, t)
            orc[i].step(int(acts[i]))
            if t == len(g[n + "_action"]) - 1:
                cursor_at_end[i] = st[i, 15]
            if fr is not None:
                want = orc[i].frame80()
                assert np.array_equal(fr[i], want), (n, t, np.argwhere(fr[i] != want)[:5])
                packed = np.packbits((want.reshape(-1) != 0), bitorder="little").view(np.int64)
                assert np.array_equal(bits[i], packed), (n, t)
                frame_checks += 1
    assert frame_checks > 1000
    for i, n in enumerate(TAPES):
        assert cursor_at_end[i] == len(g[n + "_draws"])      # every recorded draw consumed, none invented


def test_philox_envs_match_oracle_1024(torch_cuda, oracle):
    """1024 envs on their own Philox gap streams, random actions: every env, every step,
    state + outputs + frame bit-exact vs the oracle."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecGameState
    N, T, seed = 1024, 200, 0x1234_5678_9ABC
    env = VecGameState(N, seed=seed)
    orc = [oracle.GameState(seed=seed, env_id=e) for e in range(N)]
    rng = np.random.default_rng(0)
    for e in range(0, N, 97):
        assert np.array_equal(dev_state_to_snapshot(env.get_state()[e]), orc[e].snapshot())
    n_term = 0
    for t in range(T):
        acts = (rng.random(N) < 0.1).astype(np.uint8)
        frames, rew, term, score = env.frame_step(torch.from_numpy(acts).cuda())
        rew, term, score = rew.cpu().numpy(), term.cpu().numpy(), score.cpu().numpy()
        st = env.get_state()
        check_frames = t % 20 == 19
        fr = frames.cpu().numpy() if check_frames else None
        for e in range(N):
            r, te, sc = orc[e].step(int(acts[e]))
            assert (rew[e], bool(term[e]), score[e]) == (np.float32(r), te, sc), (t, e)
            n_term += te
            if check_frames and e % 16 == 0:
                assert np.array_equal(fr[e], orc[e].frame80()), (t, e)
        if t % 10 == 0:
            for e in range(0, N, 13):
                assert np.array_equal(dev_state_to_snapshot(st[e]), orc[e].snapshot()), (t, e)
    assert n_term > 500


def test_full_render_matches_oracle(torch_cuda, oracle):
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecGameState
    env = VecGameState(4, seed=9)
    orc = [oracle.GameState(seed=9, env_id=e) for e in range(4)]
    for t in range(120):
        a = np.array([t % 9 == 0, t % 5 == 0, 0, 1], np.uint8)
        env.frame_step(torch.from_numpy(a).cuda())
        for e in range(4):
            orc[e].step(int(a[e]))
        if t % 30 == 29:
            for e in range(4):
                assert np.array_equal(env.render_full(e).cpu().numpy(), orc[e].render_full()), (t, e)


def test_invalid_action_is_counted_and_env_untouched(torch_cuda):
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecGameState
    env = VecGameState(3, seed=1)
    before = env.get_state()
    env.frame_step(torch.tensor([0, 2, 1], dtype=torch.uint8).cuda())
    after = env.get_state()
    assert env.error_count() == 1
    assert np.array_equal(before[1], after[1]) and not np.array_equal(before[0], after[0])


def test_bad_blob_is_rejected_loudly(torch_cuda):
    import ctypes as C
    from dqnflappybird_amd import _lib as L
    h = C.c_void_p()
    rc = L.lib().fb_env_create(4, 0, 0, b"x" * 100, 100, C.byref(h))
    assert rc == -1 and b"sprite blob" in L.lib().fb_last_error()


def test_stats_buffer_counts_episodes_and_scores(torch_cuda):
    """fb_env_set_stats_buffer: [episodes ended, sum / max of their scores, pipes passed] == the same quantities
    accumulated on the host from the per-step terminal / score / reward outputs."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecGameState
    env = VecGameState(200, seed=11)
    stats = env.track_stats()
    env.observe()
    g = torch.Generator(device="cuda").manual_seed(1)
    ep = ssum = smax = pipes = 0
    for _ in range(300):
        a = (torch.rand(200, device="cuda", generator=g) < 0.08).to(torch.uint8)
        _, r, t, sc = env.frame_step(a, want_u8=False)
        tm = t.bool()
        ep += int(tm.sum()); ssum += int(sc[tm].sum()); pipes += int((r == 3.0).sum())
        if tm.any():
            smax = max(smax, int(sc[tm].max()))
    assert stats.tolist() == [ep, ssum, smax, pipes] and ep > 100 and pipes > 0


def test_env_trajectory_is_independent_of_the_env_count(torch_cuda):
    """BASELINE's 32 768 envs: env e of a big launch is bit-identical (state, frame, reward, terminal, score) to env e
    of a small launch with the same seed -- envs own their Philox streams, nothing depends on the grid."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecGameState
    big, small = VecGameState(32768, seed=21), VecGameState(48, seed=21)
    big.observe(); small.observe()
    assert torch.equal(big.frame_bits[:48], small.frame_bits)
    g = torch.Generator(device="cuda").manual_seed(5)
    for _ in range(120):
        a = (torch.rand(32768, device="cuda", generator=g) < 0.09).to(torch.uint8)
        big.frame_step(a, want_u8=False)
        small.frame_step(a[:48].contiguous(), want_u8=False)
        assert torch.equal(big.frame_bits[:48], small.frame_bits)
        assert torch.equal(big.reward[:48], small.reward) and torch.equal(big.terminal[:48], small.terminal)
        assert torch.equal(big.score[:48], small.score)
    assert (big.get_state()[:48] == small.get_state()).all()
    assert int(big.terminal.sum()) >= 0 and big.error_count() == 0
