"""Oracle SumTree/Memory vs the reference's own classes (tests/golden/per_sumtree.npz):
sampled tree indices, IS weights and the raw tree bytes, bit for bit."""
import json
import os

import numpy as np
import pytest

CASES = ["cap8", "cap50", "cap1000", "cap50000"]


def replay_case(oracle, golden, name, powf="numpy"):
    g = golden("per_sumtree.npz")
    with open(os.path.join(os.path.dirname(__file__), "golden", "per_sumtree.json")) as f:
        meta = json.load(f)[name]
    cap, n = meta["capacity"], meta["n"]
    mem = oracle.Memory(cap)
    rng = oracle.NpRandom(meta["seed"])
    rnd = 0
    for kind, k in meta["ops"]:
        if kind == "store":
            mem.store(k)
        else:
            idx, isw = mem.sample(n, np_rng=rng)
            assert np.array_equal(idx, g[name + "_b_idx"][rnd]), (name, rnd)
            # np.power vs libm pow: both double, allow 2 ulp
            np.testing.assert_allclose(isw, g[name + "_isw"][rnd], rtol=4e-16, atol=0)
            assert mem.beta == g[name + "_beta"][rnd]
            if powf == "numpy":
                # the p values the reference's np.power produced (not correctly rounded, see fbo_replay.c)
                mem.batch_update_p(idx, g[name + "_ps"][rnd])
            else:
                e = g[name + "_abs_err"][rnd].copy()
                mem.batch_update(idx, e)
                np.testing.assert_array_equal(e, g[name + "_abs_err"][rnd] + np.float32(0.01))  # in-place +=
            rnd += 1
    assert rnd == meta["rounds"]
    assert mem.size == meta["size"] and mem.data_pointer == meta["data_pointer"]
    return mem, g, meta


@pytest.mark.parametrize("name", CASES[:3])
def test_small_trees_bit_exact(oracle, golden, name):
    mem, g, meta = replay_case(oracle, golden, name)
    assert np.array_equal(mem.tree.view(np.uint64), g[name + "_tree"].view(np.uint64))
    assert mem.tree[0] == meta["total_p"]


def test_reference_capacity_50000_bit_exact(oracle, golden):
    name = "cap50000"
    mem, g, meta = replay_case(oracle, golden, name)
    t = mem.tree
    assert np.array_equal(t[:1023].view(np.uint64), g[name + "_tree_top"].view(np.uint64))
    assert np.array_equal(t[::97].view(np.uint64), g[name + "_tree_stride97"].view(np.uint64))
    x = np.bitwise_xor.reduce(t.view(np.uint64) * (np.arange(t.size, dtype=np.uint64) | np.uint64(1)))
    assert x == g[name + "_tree_xor"][0]


def test_get_leaf_boundaries(oracle, golden):
    """v == left sum goes left; v beyond total falls off to the last leaf; v < 0 goes left
    (BrainPrioritizedReplyDQN.py:85-100), on a capacity-6 (two leaf levels) tree."""
    g = golden("per_sumtree.npz")
    mem = oracle.Memory(6)
    for i, p in enumerate([0.5, 1.0, 0.25, 2.0, 0.125, 4.0]):
        mem.update(i + 5, p)
    assert np.array_equal(mem.tree, g["hand_tree"])
    got = [mem.get_leaf(v) for v in g["hand_v"]]
    assert got == list(g["hand_leaf"])


def test_libm_powf_path_within_one_ulp(oracle, golden):
    """Memory.batch_update through libm powf: (|e|+0.01 clipped)^0.6 within 1 fp32 ulp of NumPy's."""
    g = golden("per_sumtree.npz")
    e = g["cap1000_abs_err"].reshape(-1).copy()
    want = g["cap1000_ps"].reshape(-1)
    mem = oracle.Memory(len(e))
    idx = np.arange(len(e), dtype=np.int32) + len(e) - 1
    mem.batch_update(idx, e)
    got = mem.tree[len(e) - 1:].astype(np.float32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    assert (ulp == 0).mean() > 0.8   # NumPy float32 power is not correctly rounded
