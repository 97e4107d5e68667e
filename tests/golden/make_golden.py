#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (build container only).

What runs here and why it is a fixture generator, not product code:

* ``game/wrapped_flappy_bird.py`` and ``game/flappy_bird_utils.py`` of the
  reference are *imported unchanged* from /root/reference.  They need pygame
  1.9.4, which is not installed, so a small stand-in module (below: ``Rect`` with
  pygame's ``clip`` rule, PIL-backed ``image.load``/``get_at``, 180-degree
  ``transform.rotate``, a numpy ``Surface.blit`` for opaque/transparent pixels,
  no-op clock/display/event) is put into ``sys.modules`` first.  Physics, pipe
  spawn, scoring, hit-mask construction and ``checkCrash``/``pixelCollision`` are
  then the reference's own code (game/wrapped_flappy_bird.py:59-300).  The
  *rendered pixels* come from the stand-in's blit, so they are NOT a reference
  output: rendering/preprocess parity stays "unpinned" (SURVEY.md section 8c);
  the full-frame fixture is only a second, independent statement of
  "overwrite where alpha != 0 at int-truncated positions".
* ``SumTree``/``Memory`` are imported from the reference's
  BrainPrioritizedReplyDQN.py with an empty ``tensorflow`` module in
  ``sys.modules`` (those two classes only use NumPy,
  BrainPrioritizedReplyDQN.py:32-151).
* CPython's own ``random`` gives the uniform-replay index vectors
  (``random.sample``, BrainDQN.py:197) and the epsilon-greedy draw order
  (BrainDQN.py:103-104).

Nothing from /root/reference is copied: the outputs are numeric arrays only.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz, *.json
"""
import json
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- #
# stand-in pygame (only what game/*.py touches)
# --------------------------------------------------------------------------- #
def install_fake_pygame():
    from PIL import Image

    pg = types.ModuleType("pygame")

    class Rect:
        def __init__(self, x, y, w, h):
            self.x, self.y, self.width, self.height = int(x), int(y), int(w), int(h)

        def clip(self, B):
            A = self
            # pygame's rect.c: pg_rect_clip
            if A.x >= B.x and A.x < B.x + B.width:
                x = A.x
            elif B.x >= A.x and B.x < A.x + A.width:
                x = B.x
            else:
                return Rect(A.x, A.y, 0, 0)
            if A.x + A.width > B.x and A.x + A.width <= B.x + B.width:
                w = A.x + A.width - x
            elif B.x + B.width > A.x and B.x + B.width <= A.x + A.width:
                w = B.x + B.width - x
            else:
                return Rect(A.x, A.y, 0, 0)
            if A.y >= B.y and A.y < B.y + B.height:
                y = A.y
            elif B.y >= A.y and B.y < A.y + A.height:
                y = B.y
            else:
                return Rect(A.x, A.y, 0, 0)
            if A.y + A.height > B.y and A.y + A.height <= B.y + B.height:
                h = A.y + A.height - y
            elif B.y + B.height > A.y and B.y + B.height <= A.y + A.height:
                h = B.y + B.height - y
            else:
                return Rect(A.x, A.y, 0, 0)
            return Rect(x, y, w, h)

    class Surface:
        def __init__(self, arr):  # arr: [h][w][4] uint8 RGBA
            self.a = arr

        def convert_alpha(self):
            return self

        def convert(self):
            return self

        def get_width(self):
            return self.a.shape[1]

        def get_height(self):
            return self.a.shape[0]

        def get_at(self, xy):
            x, y = xy
            return tuple(int(v) for v in self.a[y, x])

        def blit(self, src, pos):
            x0, y0 = int(pos[0]), int(pos[1])
            h, w = src.a.shape[:2]
            H, W = self.a.shape[:2]
            xs, ys = max(x0, 0), max(y0, 0)
            xe, ye = min(x0 + w, W), min(y0 + h, H)
            if xs >= xe or ys >= ye:
                return
            sub = src.a[ys - y0:ye - y0, xs - x0:xe - x0]
            dst = self.a[ys:ye, xs:xe]
            m = sub[..., 3] != 0
            dst[m] = sub[m]

    screen = {}

    pg.Rect = Rect
    pg.init = lambda: None
    pg.time = types.SimpleNamespace(Clock=lambda: types.SimpleNamespace(tick=lambda fps: 0))
    pg.event = types.SimpleNamespace(pump=lambda: None)

    def set_mode(size):
        screen["s"] = Surface(np.zeros((size[1], size[0], 4), np.uint8))
        screen["s"].a[..., 3] = 255
        return screen["s"]

    pg.display = types.SimpleNamespace(
        set_mode=set_mode, set_caption=lambda t: None, update=lambda: None,
        get_surface=lambda: screen["s"])
    pg.image = types.SimpleNamespace(
        load=lambda p: Surface(np.array(Image.open(p).convert("RGBA"))))
    pg.transform = types.SimpleNamespace(
        rotate=lambda s, deg: Surface(np.ascontiguousarray(s.a[::-1, ::-1])) if deg == 180 else None)
    sa = types.ModuleType("pygame.surfarray")
    # array3d -> [x][y][rgb]
    sa.array3d = lambda s: np.ascontiguousarray(s.a[..., :3].transpose(1, 0, 2))
    pg.surfarray = sa
    loc = types.ModuleType("pygame.locals")
    pg.locals = loc
    sys.modules["pygame"] = pg
    sys.modules["pygame.surfarray"] = sa
    sys.modules["pygame.locals"] = loc


# --------------------------------------------------------------------------- #
# game trajectories
# --------------------------------------------------------------------------- #
def snapshot(g, game):
    up, lo = g.upperPipes, g.lowerPipes
    px = [int(p["x"]) for p in up] + [-9999] * (3 - len(up))
    uy = [int(p["y"]) for p in up] + [0] * (3 - len(up))
    ly = [int(p["y"]) for p in lo] + [0] * (3 - len(lo))
    for a, b in zip(up, lo):
        assert a["x"] == b["x"] and float(a["x"]).is_integer()
    assert float(g.playery).is_integer()
    return [int(g.playery), int(g.playerVelY), int(g.playerIndex), int(g.loopIter), int(g.basex),
            int(g.score), len(up)] + px + uy + ly


def policy_action(kind, t, g, rng):
    if kind == "never":
        return 0
    if kind == "always":
        return 1
    if kind.startswith("every"):
        return int(t % int(kind[5:]) == 0)
    if kind == "random10":
        return int(rng.random() < 0.1)
    if kind == "random50":
        return int(rng.random() < 0.5)
    if kind == "seek":
        # aim the bird at the centre of the next gap (keeps episodes long -> score events)
        nxt = None
        for u in g.upperPipes:
            if u["x"] + 52 > 57:
                nxt = u
                break
        target = (nxt["y"] + 320 + 50) if nxt is not None else 200
        return int(g.playery + 12 > target + 8 and g.playerVelY >= 0)
    raise ValueError(kind)


def game_fixtures():
    install_fake_pygame()
    os.chdir(REF)
    sys.path.insert(0, os.path.join(REF, "game"))
    sys.path.insert(0, REF)
    import game.wrapped_flappy_bird as game  # the reference's own module text

    consts = dict(FPS=game.FPS, SCREENWIDTH=game.SCREENWIDTH, SCREENHEIGHT=game.SCREENHEIGHT,
                  PIPEGAPSIZE=game.PIPEGAPSIZE, BASEY=game.BASEY, PLAYER_WIDTH=game.PLAYER_WIDTH,
                  PLAYER_HEIGHT=game.PLAYER_HEIGHT, PIPE_WIDTH=game.PIPE_WIDTH,
                  PIPE_HEIGHT=game.PIPE_HEIGHT, BACKGROUND_WIDTH=game.BACKGROUND_WIDTH)

    # hitmasks as the reference builds them (mask[x][y]) -> packed [x][y] uint8
    hm = {
        "hit_pipe_upper": np.array(game.HITMASKS["pipe"][0], np.uint8),
        "hit_pipe_lower": np.array(game.HITMASKS["pipe"][1], np.uint8),
        "hit_player": np.array(game.HITMASKS["player"], np.uint8),
    }

    draws = []
    real_randint = random.randint

    def rec_randint(a, b):
        v = real_randint(a, b)
        draws.append(v)
        return v

    random.randint = rec_randint
    out = {}
    frames = {}
    try:
        # PLAYER_INDEX_GEN is a module global that is never reset
        # (game/wrapped_flappy_bird.py:52): tapes run back to back in one process
        # and each records the cycle phase it started with.
        cyc_calls = [0]
        import itertools
        base_cycle = itertools.cycle([0, 1, 2, 1])

        def counting_gen():
            while True:
                cyc_calls[0] += 1
                yield next(base_cycle)

        game.PLAYER_INDEX_GEN = counting_gen()

        tapes = [("never", 120, 1), ("always", 150, 2), ("every7", 400, 3), ("every9", 400, 4),
                 ("random10", 1500, 5), ("random50", 600, 6), ("seek", 4000, 7), ("seek", 2500, 8)]
        for i, (kind, T, seed) in enumerate(tapes):
            random.seed(seed)
            prng = random.Random(1000 + seed)
            del draws[:]
            g = game.GameState()
            rows, acts, rew, term, sc = [snapshot(g, game)], [], [], [], []
            cyc0 = cyc_calls[0] % 4
            keep = {}
            for t in range(T):
                a = policy_action(kind, t, g, prng)
                onehot = np.zeros(2)
                onehot[a] = 1
                img, r, te, s = g.frame_step(onehot)
                assert img.shape == (288, 512, 3)
                rows.append(snapshot(g, game))
                acts.append(a)
                rew.append(float(r))
                term.append(bool(te))
                sc.append(int(s))
                if i in (4, 6) and (t % 97 == 0 or te) and len(keep) < 12:
                    keep[t] = img.copy()
            name = f"traj{i}_{kind}"
            out[name + "_state"] = np.array(rows, np.int32)
            out[name + "_action"] = np.array(acts, np.uint8)
            out[name + "_reward"] = np.array(rew, np.float32)
            out[name + "_terminal"] = np.array(term, np.uint8)
            out[name + "_score"] = np.array(sc, np.int32)
            out[name + "_draws"] = np.array(draws, np.int8)
            out[name + "_cyc0"] = np.array([cyc0], np.int32)
            for t, img in keep.items():
                # stand-in render: store losslessly but small (row-run-length is overkill:
                # np.savez_compressed handles these mostly-black frames)
                frames[f"{name}_t{t}"] = img
        # ValueError contract (game/wrapped_flappy_bird.py:99-100)
        g = game.GameState()
        try:
            g.frame_step(np.array([1, 1]))
            raised = False
        except ValueError as e:
            raised = str(e)
        consts["multiple_actions_error"] = raised
    finally:
        random.randint = real_randint
    np.savez_compressed(os.path.join(OUT, "game_trajectories.npz"), **out, **hm)
    np.savez_compressed(os.path.join(OUT, "standin_frames.npz"), **frames)
    with open(os.path.join(OUT, "game_constants.json"), "w") as f:
        json.dump(consts, f, indent=1, sort_keys=True)
    n_term = sum(int(out[k].sum()) for k in out if k.endswith("_terminal"))
    n_score = sum(int((out[k] == 3).sum()) for k in out if k.endswith("_reward"))
    print("game: tapes", len(tapes), "terminals", n_term, "score events", n_score, "frames", len(frames))


# --------------------------------------------------------------------------- #
# SumTree / Memory
# --------------------------------------------------------------------------- #
def per_fixtures():
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    import BrainPrioritizedReplyDQN as P  # the reference's own SumTree / Memory
    os.chdir(cwd)

    out = {}
    meta = {}
    cases = [("cap8", 8, 11, 60), ("cap50", 50, 12, 300), ("cap1000", 1000, 13, 2500),
             ("cap50000", 50000, 14, 52000)]
    for name, cap, seed, n_store in cases:
        np.random.seed(seed)
        aux = np.random.RandomState(900 + seed)  # abs-error source, separate stream
        mem = P.Memory(cap)
        ops = []          # (kind, n)
        b_idx, isw, errs, betas, pss = [], [], [], [], []
        stored = 0
        n = 4 if cap == 8 else 32
        # fill a little, then interleave store / sample / batch_update like
        # setPerception does (BrainPrioritizedReplyDQN.py:332-345, 277-315)
        warm = min(cap, max(n, 40)) if cap > 8 else 5
        for _ in range(warm):
            mem.store(stored)
            stored += 1
        ops.append(("store", warm))
        n_rounds = 0
        while stored < n_store:
            k = 1 if cap <= 1000 else 997
            for _ in range(k):
                mem.store(stored)
                stored += 1
            ops.append(("store", k))
            idx, data, w = mem.sample(n)
            for i, d in zip(idx, data):
                # data slot <-> tree index relation (BrainPrioritizedReplyDQN.py:99)
                assert mem.sum_tree.data[i - cap + 1] == d
            e = aux.uniform(0, 1.5, size=n).astype(np.float32)
            if n_rounds % 5 == 0:
                e[: n // 4] = 0.0  # exercise the +epsilon floor
            errs.append(e.copy())
            # what Memory.batch_update computes from e (BrainPrioritizedReplyDQN.py:147-149), kept
            # because NumPy's float32 power is not correctly rounded (differs from libm by 1 ulp)
            pss.append(np.power(np.minimum(e + np.float32(0.01), np.float32(1.0)), 0.6).astype(np.float32))
            mem.batch_update(idx, e)  # note: mutates e in place (+= 0.01)
            for ti, pv in zip(idx, pss[-1]):
                assert mem.sum_tree.tree[ti] == pv or list(idx).count(ti) > 1
            b_idx.append(idx.copy())
            isw.append(w[:, 0].copy())
            betas.append(float(mem.beta))
            ops.append(("sample_update", n))
            n_rounds += 1
        t = mem.sum_tree
        out[name + "_b_idx"] = np.array(b_idx, np.int32)
        out[name + "_isw"] = np.array(isw, np.float64)
        out[name + "_abs_err"] = np.array(errs, np.float32)
        out[name + "_ps"] = np.array(pss, np.float32)
        out[name + "_beta"] = np.array(betas, np.float64)
        if cap <= 1000:
            out[name + "_tree"] = t.tree.copy()
        else:
            # 50 000-leaf tree: keep the top 1023 nodes, every 97th node and an
            # order-sensitive checksum of the raw bytes
            out[name + "_tree_top"] = t.tree[:1023].copy()
            out[name + "_tree_stride97"] = t.tree[::97].copy()
            out[name + "_tree_xor"] = np.array(
                [np.bitwise_xor.reduce(t.tree.view(np.uint64) * (np.arange(t.tree.size, dtype=np.uint64) | np.uint64(1)))],
                np.uint64)
        meta[name] = dict(capacity=cap, seed=seed, n=n, ops=ops, size=int(t.size),
                          data_pointer=int(t.data_pointer), total_p=float(t.total_p),
                          rounds=n_rounds)
    # get_leaf on a hand-made tree, incl. v == boundary and v > total (BrainPrioritizedReplyDQN.py:85-100)
    t = P.SumTree(6)
    for i, p in enumerate([0.5, 1.0, 0.25, 2.0, 0.125, 4.0]):
        t.add(p, i)
    vs = np.array([0.0, 0.5, 0.5000001, 1.5, 1.75, 3.75, 3.875, 7.875, 7.9, 100.0, -1.0])
    out["hand_tree"] = t.tree.copy()
    out["hand_v"] = vs
    out["hand_leaf"] = np.array([t.get_leaf(v)[0] for v in vs], np.int32)
    out["hand_min_prob"] = np.array([t.get_min_prob()])
    np.savez_compressed(os.path.join(OUT, "per_sumtree.npz"), **out)
    with open(os.path.join(OUT, "per_sumtree.json"), "w") as f:
        json.dump(meta, f, sort_keys=True)
    print("per:", {k: v["rounds"] for k, v in meta.items()})


# --------------------------------------------------------------------------- #
# CPython `random` (uniform replay + epsilon-greedy stream)
# --------------------------------------------------------------------------- #
def cpython_random_fixtures():
    out = {}
    for seed in (0, 1, 12345, 2 ** 40 + 7):
        for n in (33, 277, 278, 1002, 1046, 50000, 1_000_000):
            for k in (32, 256):
                if k > n:
                    continue
                random.seed(seed)
                rounds = [random.sample(range(n), k) for _ in range(3)]
                out[f"sample_s{seed}_n{n}_k{k}"] = np.array(rounds, np.int64)
        random.seed(seed)
        # the reference's per-step consumption order: random() then randrange(2) on explore,
        # randint(0, 7) on pipe spawn (BrainDQN.py:103-104, game/wrapped_flappy_bird.py:212)
        seq = []
        for _ in range(64):
            seq.append(random.random())
            seq.append(float(random.randrange(2)))
            seq.append(float(random.randint(0, 7)))
        out[f"stream_s{seed}"] = np.array(seq, np.float64)
        random.seed(seed)
        out[f"bits32_s{seed}"] = np.array([random.getrandbits(32) for _ in range(1300)], np.uint32)
    # numpy legacy stream used by Memory.sample (BrainPrioritizedReplyDQN.py:136)
    for seed in (0, 11, 14):
        np.random.seed(seed)
        out[f"np_uniform_s{seed}"] = np.array([np.random.uniform(0.25 * i, 0.25 * (i + 1)) for i in range(700)])
    np.savez_compressed(os.path.join(OUT, "cpython_random.npz"), **out)
    print("cpython random:", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["random", "per", "game"]
    cwd = os.getcwd()
    if "random" in which:
        cpython_random_fixtures()
    if "per" in which:
        per_fixtures()
    if "game" in which:
        game_fixtures()
    os.chdir(cwd)
