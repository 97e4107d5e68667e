"""Oracle env vs trajectories produced by the reference's own game module
(tests/golden/game_trajectories.npz, made by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

TAPES = ["traj0_never", "traj1_always", "traj2_every7", "traj3_every9", "traj4_random10",
         "traj5_random50", "traj6_seek", "traj7_seek"]


def test_constants(golden):
    with open(os.path.join(os.path.dirname(__file__), "golden", "game_constants.json")) as f:
        c = json.load(f)
    assert (c["SCREENWIDTH"], c["SCREENHEIGHT"], c["PIPE_WIDTH"], c["PIPE_HEIGHT"]) == (288, 512, 52, 320)
    assert (c["PLAYER_WIDTH"], c["PLAYER_HEIGHT"], c["PIPEGAPSIZE"], c["FPS"]) == (34, 24, 100, 30)
    assert abs(c["BASEY"] - 404.48) < 1e-12
    assert c["multiple_actions_error"] == "Multiple input actions!"


def test_hitmasks_match_reference(oracle, golden):
    """getHitmask (game/flappy_bird_utils.py:103-124) incl. the 180-degree upper pipe."""
    g = golden("game_trajectories.npz")
    up, lo, pl = oracle.hitmasks()
    assert np.array_equal(up, g["hit_pipe_upper"])
    assert np.array_equal(lo, g["hit_pipe_lower"])
    assert np.array_equal(pl, g["hit_player"])


@pytest.mark.parametrize("name", TAPES)
def test_trajectory_bit_exact(oracle, golden, name):
    g = golden("game_trajectories.npz")
    st, act = g[name + "_state"], g[name + "_action"]
    env = oracle.GameState(tape=g[name + "_draws"], cyc_pos=int(g[name + "_cyc0"][0]))
    assert np.array_equal(env.snapshot(), st[0][:16])
    for t, a in enumerate(act):
        r, term, score = env.step(int(a))
        assert r == g[name + "_reward"][t], (name, t)
        assert term == bool(g[name + "_terminal"][t]), (name, t)
        assert score == g[name + "_score"][t], (name, t)
        assert np.array_equal(env.snapshot(), st[t + 1][:16]), (name, t, env.snapshot(), st[t + 1])
    # every recorded draw was consumed, none invented
    assert env.e.tape_pos == len(g[name + "_draws"])


def test_fixture_covers_the_interesting_events(golden):
    g = golden("game_trajectories.npz")
    terms = sum(int(g[n + "_terminal"].sum()) for n in TAPES)
    scores = sum(int((g[n + "_reward"] == 3).sum()) for n in TAPES)
    assert terms >= 50 and scores >= 100
    # pipe (not ground) crashes are present: a terminal whose pre-crash y was well above the ground
    pipe_crash = 0
    for n in TAPES:
        st, te = g[n + "_state"], g[n + "_terminal"]
        for t in np.nonzero(te)[0]:
            if st[t][0] + 10 < 370:
                pipe_crash += 1
    assert pipe_crash >= 10


def test_invalid_action_raises(oracle):
    env = oracle.GameState(seed=1)
    with pytest.raises(ValueError, match="Multiple input actions!"):
        env.step(2)


def test_full_render_matches_standin_frames(oracle, golden):
    """NOT a reference output (SDL is absent): the stand-in blit of make_golden.py is a second,
    independent statement of 'overwrite where alpha != 0'; rendering parity stays unpinned."""
    g = golden("game_trajectories.npz")
    f = golden("standin_frames.npz")
    checked = 0
    for name in ("traj4_random10", "traj6_seek"):
        want = {int(k.split("_t")[-1]): f[k] for k in f.files if k.startswith(name + "_t")}
        env = oracle.GameState(tape=g[name + "_draws"], cyc_pos=int(g[name + "_cyc0"][0]))
        for t, a in enumerate(g[name + "_action"]):
            env.step(int(a))
            if t in want:
                assert np.array_equal(env.render_full(), want[t]), (name, t)
                checked += 1
    assert checked >= 20


def test_preprocess_shape_and_binary(oracle):
    env = oracle.GameState(seed=5)
    lit = []
    for t in range(300):
        env.step(int(t % 9 == 0))
        fr = env.frame80()
        assert set(np.unique(fr)) <= {0, 255}
        lit.append((fr == 255).mean())
    # about 37 % of the pixels are lit
    assert 0.25 < np.mean(lit) < 0.5


def test_resize_taps_match_survey_probe(oracle):
    """cv2.resize geometry: row taps start 1,4,8,12..., col taps 2,9,15,21...;
    check through an impulse response of the oracle's preprocess."""
    rows, cols = [], []
    for r in range(4):
        # find which source rows influence output row r: light a full source row
        hit = []
        for sx in range(0, 20):
            img = np.zeros((288, 512, 3), np.uint8)
            img[sx, :, :] = 255
            if oracle.preprocess(img)[r].any():
                hit.append(sx)
        rows.append(hit[0])
    for c in range(4):
        hit = []
        for sy in range(0, 30):
            img = np.zeros((288, 512, 3), np.uint8)
            img[:, sy, :] = 255
            if oracle.preprocess(img)[:, c].any():
                hit.append(sy)
        cols.append(hit[0])
    assert rows == [1, 4, 8, 12]
    assert cols == [2, 9, 15, 21]
