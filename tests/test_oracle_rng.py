"""Oracle RNG restatements vs CPython / NumPy themselves (tests/golden/cpython_random.npz
and, since both are importable here, live)."""
import random

import numpy as np
import pytest


def test_mt_bits_match_cpython(oracle, golden):
    g = golden("cpython_random.npz")
    for seed in (0, 1, 12345, 2 ** 40 + 7):
        r = oracle.PyRandom(seed)
        got = np.array([r.getrandbits(32) for _ in range(1300)], np.uint32)
        assert np.array_equal(got, g[f"bits32_s{seed}"])


def test_random_sample_indices_bit_exact(oracle, golden):
    """random.sample(range(n), k): BrainDQN.py:197 (the sampled value is the deque index)."""
    g = golden("cpython_random.npz")
    n_checked = 0
    for key in g.files:
        if not key.startswith("sample_"):
            continue
        _, s, n, k = key.split("_")
        seed, n, k = int(s[1:]), int(n[1:]), int(k[1:])
        r = oracle.PyRandom(seed)
        for want in g[key]:
            assert np.array_equal(r.sample(n, k), want), key
        n_checked += 1
    assert n_checked >= 40


def test_stream_order_random_randrange_randint(oracle, golden):
    """epsilon draw, randrange(2), randint(0,7) in the reference's per-step order."""
    g = golden("cpython_random.npz")
    for seed in (0, 1, 12345, 2 ** 40 + 7):
        r = oracle.PyRandom(seed)
        seq = []
        for _ in range(64):
            seq += [r.random(), float(r.randbelow(2)), float(r.randbelow(8))]
        assert np.array_equal(np.array(seq), g[f"stream_s{seed}"])


def test_live_against_this_interpreter(oracle):
    for seed in (3, 99, 2 ** 33 + 1):
        random.seed(seed)
        r = oracle.PyRandom(seed)
        for n, k in ((1002, 32), (50000, 32), (300, 32), (100, 32), (2000, 256), (900, 256)):
            assert list(r.sample(n, k)) == random.sample(range(n), k)
    with pytest.raises(ValueError):
        oracle.PyRandom(0).sample(10, 32)


def test_numpy_legacy_uniform(oracle, golden):
    g = golden("cpython_random.npz")
    for seed in (0, 11, 14):
        r = oracle.NpRandom(seed)
        got = np.array([r.uniform(0.25 * i, 0.25 * (i + 1)) for i in range(700)])
        assert np.array_equal(got, g[f"np_uniform_s{seed}"])


def test_philox_known_answer(oracle):
    # Random123 kat_vectors: philox4x32-10, ctr = key = 0 / all ones / pi digits
    assert list(oracle.philox(0, 0, 0, 0, 0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(oracle.philox(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)) == \
        [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(oracle.philox(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
