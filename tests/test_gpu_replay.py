"""HIP replay memory (through the C ABI): deque semantics, bit-exact CPython sampling, gather,
and the prioritized SumTree/Memory bit-exact against the reference's own classes."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def rand_frames(rng, n):
    return (rng.random((n, 80, 80)) < 0.37).astype(np.uint8) * 255


class DequeModel:
    """The reference's replayMemory for N envs appending in env order (BrainDQN.py:66-72)."""

    def __init__(self, cap, first):                  # first: u8[N,80,80]
        self.cap = cap
        self.state = np.stack([first] * 4, axis=-1)  # setInitState, BrainDQN.py:238-239
        self.mem = []

    def push(self, nxt, a, r, t):
        new = np.concatenate([self.state[..., 1:], nxt[..., None]], axis=-1)   # BrainDQN.py:68
        for e in range(len(a)):
            self.mem.append((self.state[e], a[e], r[e], new[e], t[e]))
            if len(self.mem) > self.cap:
                self.mem.pop(0)
        self.state = new


@pytest.mark.parametrize("n_envs,cap,steps", [(1, 50, 130), (4, 50, 40), (3, 100, 80), (8, 1000, 200)])
def test_uniform_deque_semantics_and_gather(torch_cuda, n_envs, cap, steps):
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    rng = np.random.default_rng(n_envs * 1000 + cap)
    rep = VecReplay(cap, n_envs)
    first = rand_frames(rng, n_envs)
    rep.reset(torch.from_numpy(first).cuda())
    model = DequeModel(cap, first)
    assert len(rep) == 0
    rep.seed(12345)
    random.seed(12345)
    for t in range(steps):
        nxt = rand_frames(rng, n_envs)
        a = rng.integers(0, 2, n_envs).astype(np.uint8)
        r = rng.choice(np.array([0.1, 3, -3], np.float32), n_envs)
        te = (r == -3).astype(np.uint8)
        rep.push(torch.from_numpy(nxt).cuda(), torch.from_numpy(a).cuda(), torch.from_numpy(r).cuda(),
                 torch.from_numpy(te).cuda())
        model.push(nxt, a, r, te)
        assert np.array_equal(rep.current_state().cpu().numpy(), model.state)
        if len(model.mem) >= 32 and t % 3 == 0:
            assert len(rep) == len(model.mem)
            idx, _ = rep.sample(32)
            want_idx = random.sample(range(len(model.mem)), 32)          # BrainDQN.py:197
            assert idx.cpu().tolist() == want_idx
            s, aa, rr, s2, tt = (x.cpu().numpy() for x in rep.gather(idx))
            for b, j in enumerate(want_idx):
                ms, ma, mr, ms2, mt = model.mem[j]
                assert np.array_equal(s[b], ms) and np.array_equal(s2[b], ms2), (t, b, j)
                assert (aa[b], rr[b], tt[b]) == (ma, mr, mt)


def test_cpython_sample_golden_vectors(torch_cuda, golden):
    """random.sample(range(n), k) for the fixture's (seed, n, k) incl. the pool path (n <= setsize)."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    g = golden("cpython_random.npz")
    z8 = torch.zeros((1, 80, 80), dtype=torch.uint8, device="cuda")
    za, zr = torch.zeros(1, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.float32, device="cuda")
    checked = 0
    for key in sorted(g.files):
        if not key.startswith("sample_"):
            continue
        _, s, n, k = key.split("_")
        seed, n, k = int(s[1:]), int(n[1:]), int(k[1:])
        if n > 2000:
            continue                                  # those sizes run in the N-env test below
        rep = VecReplay(n, 1)
        rep.reset(z8)
        for _ in range(n):
            rep.push(z8, za, zr, za)
        rep.seed(seed)
        for want in g[key]:
            idx, _ = rep.sample(k)
            assert idx.cpu().tolist() == list(want), key
        checked += 1
    assert checked >= 16


@pytest.mark.parametrize("n", [50000, 1000000])
def test_cpython_sample_large_memories(torch_cuda, golden, n):
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    g = golden("cpython_random.npz")
    N = 1000
    rep = VecReplay(n, N)
    z8 = torch.zeros((N, 100), dtype=torch.int64, device="cuda")
    za, zr = torch.zeros(N, dtype=torch.uint8, device="cuda"), torch.zeros(N, dtype=torch.float32, device="cuda")
    rep.reset(z8)
    for _ in range(n // N):
        rep.push(z8, za, zr, za)
    assert len(rep) == n
    for seed in (0, 1, 12345, 2 ** 40 + 7):
        for k in (32, 256):
            rep.seed(seed)
            for want in g[f"sample_s{seed}_n{n}_k{k}"]:
                idx, _ = rep.sample(k)
                assert idx.cpu().tolist() == list(want)


def test_push_sample_equals_push_then_sample(torch_cuda):
    """fb_replay_push_sample (sampler as an extra workgroup of the push launch) == the two calls in a row:
    same indices, same MT19937 consumption, same ring contents -- also across the ring wrap and for u8 frames."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    N, cap, B = 7, 300, 32
    a_, b_ = VecReplay(cap, N), VecReplay(cap, N)
    a_.seed(11, "cpython"); b_.seed(11, "cpython")
    g = torch.Generator(device="cuda").manual_seed(3)
    first = (torch.rand((N, 80, 80), device="cuda", generator=g) < 0.4).to(torch.uint8) * 255
    a_.reset(first); b_.reset(first)
    for step in range(120):
        fr = (torch.rand((N, 80, 80), device="cuda", generator=g) < 0.4).to(torch.uint8) * 255
        ac = (torch.rand(N, device="cuda", generator=g) < 0.5).to(torch.uint8)
        rw = torch.rand(N, device="cuda", generator=g)
        tm = (torch.rand(N, device="cuda", generator=g) < 0.1).to(torch.uint8)
        if (step + 1) * N < B:                     # too few transitions for a sample of 32: plain pushes
            a_.push(fr, ac, rw, tm); b_.push(fr, ac, rw, tm)
            continue
        a_.push(fr, ac, rw, tm)
        ia, _ = a_.sample(B)
        ib = b_.push_sample(fr, ac, rw, tm, B)
        assert ia.cpu().tolist() == ib.cpu().tolist()
        if step % 17 == 0:
            for x, y in zip(a_.gather(ia), b_.gather(ib)):
                assert torch.equal(x, y)
    assert len(a_) == len(b_) == cap


def test_sample_larger_than_population_fails_loudly(torch_cuda):
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    from dqnflappybird_amd._lib import FbError
    rep = VecReplay(100, 1)
    z8 = torch.zeros((1, 80, 80), dtype=torch.uint8, device="cuda")
    za, zr = torch.zeros(1, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.float32, device="cuda")
    rep.reset(z8)
    for _ in range(10):
        rep.push(z8, za, zr, za)
    rep.sample(32)
    with pytest.raises(FbError, match="Sample larger than population"):
        len(rep)


def test_philox_sampler_in_range_and_reproducible(torch_cuda):
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    N = 64
    reps = [VecReplay(5000, N), VecReplay(5000, N)]
    z8 = torch.zeros((N, 100), dtype=torch.int64, device="cuda")
    za, zr = torch.zeros(N, dtype=torch.uint8, device="cuda"), torch.zeros(N, dtype=torch.float32, device="cuda")
    out = []
    for rep in reps:
        rep.reset(z8)
        for _ in range(50):
            rep.push(z8, za, zr, za)
        rep.seed(7, "philox")
        out.append(torch.stack([rep.sample(256)[0].clone() for _ in range(20)]).cpu().numpy())
    assert np.array_equal(out[0], out[1])
    assert out[0].min() >= 0 and out[0].max() < 3200 and len(np.unique(out[0])) > 2500


# ----------------------------------------------------------------------------- PER
def per_replay_case(torch, golden, name, n_envs=1):
    from dqnflappybird_amd.vec import VecReplay
    g = golden("per_sumtree.npz")
    with open(os.path.join(os.path.dirname(__file__), "golden", "per_sumtree.json")) as f:
        meta = json.load(f)[name]
    cap, n = meta["capacity"], meta["n"]
    rep = VecReplay(cap, n_envs, prioritized=True)
    z8 = torch.zeros((n_envs, 100), dtype=torch.int64, device="cuda")
    za = torch.zeros(n_envs, dtype=torch.uint8, device="cuda")
    zr = torch.zeros(n_envs, dtype=torch.float32, device="cuda")
    rep.reset(z8)
    rep.seed(meta["seed"], "numpy")
    rnd = 0
    for kind, k in meta["ops"]:
        if kind == "store":
            assert k % n_envs == 0
            for _ in range(k // n_envs):
                rep.push(z8, za, zr, za)
        else:
            idx, isw = rep.sample(n)
            assert idx.cpu().tolist() == list(g[name + "_b_idx"][rnd]), (name, rnd)
            np.testing.assert_allclose(isw.cpu().numpy(), g[name + "_isw"][rnd], rtol=1e-13)
            rep.update_priorities(idx, priorities=torch.from_numpy(g[name + "_ps"][rnd]).cuda())
            rnd += 1
    assert rnd == meta["rounds"]
    tree, ptr, size, beta = rep.per_state()
    assert (ptr, size) == (meta["data_pointer"], meta["size"])
    assert beta == g[name + "_beta"][-1]
    return tree, g, meta


@pytest.mark.parametrize("name", ["cap8", "cap50", "cap1000"])
def test_per_bit_exact_small(torch_cuda, golden, name):
    tree, g, meta = per_replay_case(torch_cuda, golden, name)
    assert np.array_equal(tree.view(np.uint64), g[name + "_tree"].view(np.uint64))


def test_per_bit_exact_reference_capacity(torch_cuda, golden):
    """capacity 50 000 (REPLAY_MEMORY): two leaf levels, ring wrap, 53 sample/update rounds."""
    name = "cap50000"
    tree, g, meta = per_replay_case(torch_cuda, golden, name)
    assert np.array_equal(tree[:1023].view(np.uint64), g[name + "_tree_top"].view(np.uint64))
    assert np.array_equal(tree[::97].view(np.uint64), g[name + "_tree_stride97"].view(np.uint64))
    x = np.bitwise_xor.reduce(tree.view(np.uint64) * (np.arange(tree.size, dtype=np.uint64) | np.uint64(1)))
    assert x == g[name + "_tree_xor"][0]


def test_per_vector_store_equals_sequential(torch_cuda, golden):
    """N envs storing per step == the same transitions stored one by one (997 = 997 x 1 here,
    cap50000 tape with n_envs = 997)."""
    name = "cap50000"
    g = golden("per_sumtree.npz")
    with open(os.path.join(os.path.dirname(__file__), "golden", "per_sumtree.json")) as f:
        meta = json.load(f)[name]
    # the tape's warm-up store count (40) is not a multiple of 997: replay it with n_envs = 1 first
    # is covered above; here check pure stores with N = 997 against the oracle
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    from oracle import oracle as o
    N, cap = 997, 50000
    rep = VecReplay(cap, N, prioritized=True)
    z8 = torch.zeros((N, 100), dtype=torch.int64, device="cuda")
    za, zr = torch.zeros(N, dtype=torch.uint8, device="cuda"), torch.zeros(N, dtype=torch.float32, device="cuda")
    rep.reset(z8)
    mem = o.Memory(cap)
    rng = np.random.default_rng(5)
    for step in range(60):                       # 59 820 stores: wraps the ring and crosses the level split
        rep.push(z8, za, zr, za)
        mem.store(N)
        if step % 7 == 3:
            u = rng.random(32)
            idx, isw = rep.sample(32, uniforms=torch.from_numpy(u).cuda())
            oi, ow = mem.sample(32, u=u)
            assert idx.cpu().tolist() == list(oi)
            np.testing.assert_allclose(isw.cpu().numpy(), ow, rtol=1e-13)
            ps = (rng.random(32).astype(np.float32) * 1.2 + 0.01).clip(max=1.0) ** np.float32(0.6)
            rep.update_priorities(idx, priorities=torch.from_numpy(ps).cuda())
            mem.batch_update_p(oi, ps)
    tree, ptr, size, beta = rep.per_state()
    assert (ptr, size) == (mem.data_pointer, mem.size)
    assert np.array_equal(tree.view(np.uint64), mem.tree.view(np.uint64))


@pytest.mark.parametrize("N,cap", [(997, 50000), (4096, 100000), (64, 300)])
def test_per_fast_mode_tree_is_the_exact_pairwise_sum(torch_cuda, N, cap):
    """FB_PER_FAST keeps the heaps by recomputation (node = left + right), not by the reference's running
    sums: the LEAVES, pointer and size are those of the reference-order oracle, every internal node is the
    correctly rounded sum of its two children (so the root is within fp64 rounding of the oracle's total),
    and the sampled indices are exactly what SumTree.get_leaf returns on the device's own tree."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    from oracle import oracle as o
    rep = VecReplay(cap, N, prioritized=True)
    rep.set_per_mode("fast")
    z8 = torch.zeros((N, 100), dtype=torch.int64, device="cuda")
    za, zr = torch.zeros(N, dtype=torch.uint8, device="cuda"), torch.zeros(N, dtype=torch.float32, device="cuda")
    rep.reset(z8)
    mem = o.Memory(cap)
    rng = np.random.default_rng(N)

    def get_leaf(tree, v):                        # BrainPrioritizedReplyDQN.py:85-100
        i = 0
        while 2 * i + 1 < len(tree):
            l = 2 * i + 1
            if v <= tree[l]:
                i = l
            else:
                v -= tree[l]
                i = l + 1
        return i

    for step in range(2 * cap // N + 3):           # wraps the ring, crosses the leaf-depth split
        rep.push(z8, za, zr, za)
        mem.store(N)
        if step % 5 == 2:
            tree, _, _, _ = rep.per_state()
            u = rng.random(32)
            idx, isw = rep.sample(32, uniforms=torch.from_numpy(u).cuda())
            seg = tree[0] / 32                     # Memory.sample: v = uniform(seg*i, seg*(i+1)) with the injected u
            want = [get_leaf(tree, seg * i + (seg * (i + 1) - seg * i) * u[i]) for i in range(32)]
            assert idx.cpu().tolist() == want
            ps = (rng.random(32).astype(np.float32) * 1.2 + 0.01).clip(max=1.0) ** np.float32(0.6)
            ps[5] = ps[0]
            idx[5] = idx[0]                        # a duplicate index: the later entry wins in both
            rep.update_priorities(idx, priorities=torch.from_numpy(ps).cuda())
            mem.batch_update_p(idx.cpu().numpy(), ps)
    tree, ptr, size, beta = rep.per_state()
    assert (ptr, size) == (mem.data_pointer, mem.size)
    assert np.array_equal(tree[cap - 1:].view(np.uint64), mem.tree[cap - 1:].view(np.uint64))      # leaves
    inner = np.arange(cap - 1)
    assert np.array_equal(tree[inner], tree[2 * inner + 1] + tree[2 * inner + 2])                  # exact fp64 sums
    assert abs(tree[0] - mem.tree[0]) <= 1e-9 * mem.tree[0]
    rep.set_per_mode("exact")                      # the tree stays valid for the exact kernels
    rep.push(z8, za, zr, za)
    t2, _, _, _ = rep.per_state()
    assert abs(t2[0] - t2[cap - 1:].sum()) <= 1e-9 * t2[0]


def test_per_device_pow_close_to_numpy(torch_cuda, golden):
    """abs_err path: (min(|e|+0.01, 1))^0.6 computed on the device, within 1 fp32 ulp of NumPy's
    (which itself is not correctly rounded), and abs_err updated in place like the reference."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    g = golden("per_sumtree.npz")
    e = g["cap1000_abs_err"].reshape(-1)[:256].copy()
    want = g["cap1000_ps"].reshape(-1)[:256]
    rep = VecReplay(256, 1, prioritized=True)
    z8 = torch.zeros((1, 100), dtype=torch.int64, device="cuda")
    za, zr = torch.zeros(1, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.float32, device="cuda")
    rep.reset(z8)
    for _ in range(256):
        rep.push(z8, za, zr, za)
    idx = torch.arange(255, 511, dtype=torch.int64, device="cuda")
    ed = torch.from_numpy(e).cuda()
    rep.update_priorities(idx, abs_err=ed)
    np.testing.assert_array_equal(ed.cpu().numpy(), e + np.float32(0.01))
    tree, _, _, _ = rep.per_state()
    got = tree[255:].astype(np.float32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1


def test_full_size_ring_addressing_one_million_slots(torch_cuda):
    """BASELINE's 1 M-slot replay with 1024 envs, filled past the wrap (1 100 steps = 1 126 400 stores): the gathered
    (s, a, r, s', t) of sampled positions equal a closed-form transition generator -- a size-independent property of
    the ring addressing (deque position j <-> g = oldest + j, (step, env) = divmod(g, N), frame window s = t-3..t)."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    N, cap, steps = 1024, 1_000_000, 1100
    rep = VecReplay(cap, N)
    env_id = torch.arange(N, device="cuda", dtype=torch.int64)

    def bits(step):                                  # frame of (step, env): 100 words, a cheap injective hash
        w = torch.arange(100, device="cuda", dtype=torch.int64)
        x = (step * 1315423911 + env_id[:, None] * 2654435761 + w[None, :] * 97 + 12345) % 2147483647
        return (x * 6364136223846793005 + (x << 17)).contiguous()

    def meta(step):
        a = ((env_id * 7 + step) % 2).to(torch.uint8)
        r = ((env_id + 3 * step) % 5).to(torch.float32) * 0.5
        t = ((env_id * 13 + step) % 11 == 0).to(torch.uint8)
        return a, r, t

    rep.reset(bits(0))                               # frame index 0 = the initial observation
    for st in range(1, steps + 1):
        a, r, t = meta(st)
        rep.push(bits(st), a, r, t)                  # transition of step st: s = frames st-4..st-1 (clamped at 0), s' = st-3..st
    assert len(rep) == cap
    total = steps * N
    oldest = total - cap
    rng = np.random.default_rng(0)
    j = np.concatenate([rng.integers(0, cap, 60), [0, 1, cap - 1, cap - 2, N - 1, N]])
    idx = torch.from_numpy(j).cuda()
    s, a, r, s2, t = rep.gather(idx)
    g = oldest + j
    stp, env = g // N + 1, g % N                     # transition pushed at step stp by env
    fb = {int(k): bits(int(k)) for k in set(np.concatenate([np.maximum(stp - 4 + d, 0) for d in range(5)]).tolist())}

    def unpack(step, e):                             # u8[80,80] of the frame (bit p of the 6400-bit row-major image)
        words = fb[int(step)][int(e)].cpu().numpy().astype(np.uint64)
        return (np.unpackbits(words.view(np.uint8), bitorder="little").reshape(80, 80) * 255).astype(np.uint8)

    for b in range(len(j)):
        for f in range(4):
            assert np.array_equal(s[b, :, :, f].cpu().numpy(), unpack(max(stp[b] - 4 + f, 0), env[b])), (b, f)
            assert np.array_equal(s2[b, :, :, f].cpu().numpy(), unpack(max(stp[b] - 3 + f, 0), env[b])), (b, f)
        am, rm, tm = meta(int(stp[b]))
        assert (int(a[b]), float(r[b]), int(t[b])) == (int(am[env[b]]), float(rm[env[b]]), int(tm[env[b]]))


@pytest.mark.parametrize("cap,B,pool", [(50, 200, 30), (1000, 256, 90), (50000, 200, 5000), (50000, 32, 12)])
def test_per_batch_update_long_lists_with_duplicates(torch_cuda, cap, B, pool):
    """Memory.batch_update (BrainPrioritizedReplyDQN.py:146-151) on lists longer than a wave and full of duplicates (the leaves are drawn
    from a pool smaller than the list), leaves on both levels of the heap: the tree bytes stay the reference's (later duplicates win, every
    ancestor's running sum in list order), and so do the max / min heaps -- checked through what reads them: Memory.sample's importance
    weights divide by the minimum leaf, Memory.store writes the maximum leaf."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import VecReplay
    from oracle import oracle as o
    rep = VecReplay(cap, 1, prioritized=True)
    z8 = torch.zeros((1, 100), dtype=torch.int64, device="cuda")
    za, zr = torch.zeros(1, dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.float32, device="cuda")
    rep.reset(z8)
    mem = o.Memory(cap)
    fill = min(cap, 3000)
    for _ in range(fill):
        rep.push(z8, za, zr, za)
    mem.store(fill)
    rng = np.random.default_rng(cap + B)
    for rnd in range(6):
        leaves = rng.choice(fill, size=min(pool, fill), replace=False)
        idx = (rng.choice(leaves, size=B) + cap - 1).astype(np.int64)
        if rnd == 3:
            idx[:] = idx[0]                                         # one leaf, B times
        ps = (rng.random(B).astype(np.float32) * 1.3 + 0.005).clip(max=1.0) ** np.float32(0.6)
        rep.update_priorities(torch.from_numpy(idx).cuda(), priorities=torch.from_numpy(ps).cuda())
        mem.batch_update_p(idx.astype(np.int32), ps)
        tree, ptr, size, beta = rep.per_state()
        assert np.array_equal(tree.view(np.uint64), mem.tree.view(np.uint64)), rnd
        u = rng.random(32)
        di, dw = rep.sample(32, uniforms=torch.from_numpy(u).cuda())
        oi, ow = mem.sample(32, u=u)
        assert di.cpu().tolist() == list(oi)
        np.testing.assert_allclose(dw.cpu().numpy(), ow, rtol=1e-13)      # (p / total / min_prob)^-beta: the MINIMUM heap
        rep.push(z8, za, zr, za)                                     # stores the MAXIMUM leaf
        mem.store(1)
        tree, ptr, size, beta = rep.per_state()
        assert (ptr, size) == (mem.data_pointer, mem.size)
        assert np.array_equal(tree.view(np.uint64), mem.tree.view(np.uint64)), rnd
