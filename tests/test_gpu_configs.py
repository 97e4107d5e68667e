"""BASELINE.json configs[2..4] as parity / smoke cases at their full sizes (the bench line is configs[1]):
  [2] BrainDoubleDQN target-net, 4096 envs, batch 256   (fp32 here; bf16 is deferred)
  [3] BrainPrioritizedReplyDQN GPU SumTree, 1M-slot replay, 4096 envs
  [4] BrainDuelingDQN, 4096 envs per rank (x8 ranks on the real node)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def test_config2_double_dqn_batch256_gradients(torch_cuda, oracle):
    """largest supported minibatch (256) through the real Double-DQN update vs the oracle."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    from tests.test_gpu_qnet import oracle_train_grads, rand_states, trained_like_params
    B = 256
    cfg = oracle.qcfg()
    p_on, p_tg = trained_like_params(oracle, cfg, 1), trained_like_params(oracle, cfg, 2)
    net = QNet(max_batch=B)
    net.load_params(p_on, 0)
    net.load_params(p_tg, 1)
    rng = np.random.default_rng(256)
    s, s2 = rand_states(rng, B), rand_states(rng, B)
    a = rng.integers(0, 2, B).astype(np.uint8)
    r = rng.choice(np.array([0.1, 3, -3], np.float32), B, p=[0.8, 0.1, 0.1])
    t = (r == -3).astype(np.uint8)
    d = lambda x: torch.from_numpy(x).cuda()
    grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda")
    loss, ae, y = net.train_step("double", d(s), d(a), d(r), d(s2), d(t), flat_grad=grad)
    y0, loss0, ae0, g0 = oracle_train_grads(oracle, cfg, p_on, p_tg, "double", s, a, r, s2, t, None)
    np.testing.assert_allclose(y.cpu().numpy(), y0, rtol=0, atol=1e-4)
    np.testing.assert_allclose(loss.item(), loss0, rtol=1e-4)
    g = grad.cpu().numpy()
    # with 256 samples a handful of units sit on a ReLU kink (see test_gpu_qnet): compare in aggregate
    rel = np.linalg.norm(g - g0) / np.linalg.norm(g0)
    assert rel < 2e-2, rel
    for lo, hi in ((77984, 77984 + 1600 * 512), (77984 + 1600 * 512 + 512, net.n_params)):   # fc1 / head: kink-insensitive
        np.testing.assert_allclose(g[lo:hi], g0[lo:hi], rtol=2e-3, atol=2e-5 * np.abs(g0[lo:hi]).max())


@pytest.mark.parametrize("algo,arch,batch,prior", [("double", "plain", 256, False), ("per", "plain", 32, True),
                                                   ("nature", "dueling", 32, False)])
def test_configs_run_at_4096_envs(torch_cuda, algo, arch, batch, prior):
    from dqnflappybird_amd.vecbrain import VecBrain
    vb = VecBrain(4096, algo=algo, arch=arch, batch=batch, capacity=1_000_000, observe=2, seed=4)
    vb.run(8, log_every=0)
    assert vb.timeStep == 8 and len(vb.replay) == 8 * 4096
    assert np.isfinite(vb.last_loss.item())
    if prior:
        tree, ptr, size, beta = vb.replay.per_state()
        assert (ptr, size) == (8 * 4096, 8 * 4096) and abs(beta - (0.4 + 5 * 0.001)) < 1e-12
        assert np.isclose(tree[0], tree[1_000_000 - 1:].sum(), rtol=1e-12)
        # the heap invariant holds level by level (every inner node = sum of its children, up to rounding)
        i = np.arange(0, 999_999)
        np.testing.assert_allclose(tree[i], tree[2 * i + 1] + tree[2 * i + 2], rtol=1e-9, atol=1e-12)
