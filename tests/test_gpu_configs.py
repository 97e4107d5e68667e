"""BASELINE.json configs[2..4] as parity / smoke cases at their full sizes (the bench line is configs[1]):
  [2] BrainDoubleDQN target-net, 4096 envs, batch 256   (fp32 and the configuration's stated dtype, bf16)
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


# bf16 training (configs[2]'s dtype): every GEMM operand rounded to bf16 (8 significant bits, relative rounding error <= 2^-9), fp32
# accumulation, fp32 master weights and Adam.  1e-4 absolute is out of reach by construction; the bounds below are RELATIVE and per
# tensor: independent operand roundings of 2^-9 average out over the reduction length, leaving ~0.5 % per layer, compounding over
# the 4 (forward) + 4 (backward) layers a gradient has gone through -- a few per cent on the conv1 end, ~1 % near the head.
BF16_Q_REL = 0.03          # Q / targets: fraction of the Q scale
BF16_GRAD_REL = 0.06       # per-tensor ||g_bf16 - g_fp32|| / ||g_fp32||


@pytest.mark.parametrize("B,algo", [(256, "double"), (32, "nature")])      # the LDS-staged large-batch kernels / the small-batch ones
def test_bf16_training_gradients_within_relative_bound(torch_cuda, oracle, B, algo):
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    from tests.test_gpu_qnet import oracle_train_grads, rand_states, trained_like_params
    cfg = oracle.qcfg()
    p_on, p_tg = trained_like_params(oracle, cfg, 1), trained_like_params(oracle, cfg, 2)
    net = QNet(max_batch=B)
    net.load_params(p_on, 0); net.load_params(p_tg, 1)
    rng = np.random.default_rng(B)
    s, s2 = rand_states(rng, B), rand_states(rng, B)
    a = rng.integers(0, 2, B).astype(np.uint8)
    r = rng.choice(np.array([0.1, 3, -3], np.float32), B, p=[0.8, 0.1, 0.1])
    t = (r == -3).astype(np.uint8)
    d = lambda x: torch.from_numpy(x).cuda()
    g32, g16 = (torch.zeros(net.n_params, dtype=torch.float32, device="cuda") for _ in range(2))
    loss32, _, y32 = net.train_step(algo, d(s), d(a), d(r), d(s2), d(t), flat_grad=g32)
    loss32, y32 = loss32.item(), y32.cpu().numpy().copy()
    net.set_train_dtype("bf16")
    loss16, _, y16 = net.train_step(algo, d(s), d(a), d(r), d(s2), d(t), flat_grad=g16)
    loss16, y16 = loss16.item(), y16.cpu().numpy().copy()
    net.set_train_dtype("f32")
    g_again = torch.zeros_like(g32)
    net.train_step(algo, d(s), d(a), d(r), d(s2), d(t), flat_grad=g_again)
    assert torch.equal(g_again, g32)                                    # switching back restores the fp32 arithmetic bit for bit
    y0, loss0, _, g0 = oracle_train_grads(oracle, cfg, p_on, p_tg, algo, s, a, r, s2, t, None)
    scale = np.abs(y0).max()
    assert np.abs(y32 - y0).max() < 1e-4                                 # fp32 mode: the north-star bound
    assert 0 < np.abs(y16 - y0).max() < BF16_Q_REL * scale               # bf16 mode: its own, relative bound (and it IS a different arithmetic)
    assert abs(loss16 - loss0) < 3 * BF16_Q_REL * abs(loss0)
    g16n, g32n = g16.cpu().numpy(), g32.cpu().numpy()
    bounds = [0, 8192, 8224, 40992, 41056, 77920, 77984, 77984 + 1600 * 512, 77984 + 1600 * 512 + 512, net.n_params]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ref = g32n[lo:hi]                                                # against the fp32 DEVICE gradient: same ReLU / pool masks up to the bf16 flips
        rel = np.linalg.norm(g16n[lo:hi] - ref) / np.linalg.norm(ref)
        assert 0 < rel < BF16_GRAD_REL, (lo, hi, rel)
    assert np.linalg.norm(g16n - g0) / np.linalg.norm(g0) < BF16_GRAD_REL      # and against the oracle, whole vector
    # the master weights stay fp32: nothing was applied in gradient-only mode
    assert np.array_equal(net.store_params().cpu().numpy(), p_on)


def test_config2_end_to_end_in_bf16_at_4096_envs(torch_cuda):
    """configs[2] as stated: Double-DQN, 4096 envs, batch 256, bf16 acting AND training, through the device-resident loop.  The
    run stays finite, the optimizer moves fp32 master weights, and 25 bf16 train steps land within a few per cent (of the update)
    of the same 25 steps in fp32 from the same start."""
    torch = torch_cuda
    from dqnflappybird_amd.vecbrain import VecBrain
    kw = dict(algo="double", batch=256, capacity=1_000_000, observe=3, seed=4)
    a, b = VecBrain(4096, **kw), VecBrain(4096, **kw)
    b.set_dtype("bf16")
    for vb in (a, b):
        vb.net.set_hparams(lr=1e-4)                                      # visible steps
    p0 = a.net.store_params().clone()
    a.run(28, log_every=0); b.run(28, log_every=0)
    assert b.dtype == "bf16" and np.isfinite(b.last_loss.item()) and b.env.error_count() == 0
    pa, pb = a.net.store_params(), b.net.store_params()
    assert pb.dtype == torch.float32 and torch.isfinite(pb).all()
    upd_a, upd_b = (pa - p0).norm().item(), (pb - p0).norm().item()
    assert upd_b > 0 and abs(upd_b / upd_a - 1) < 0.1                   # same amount of learning
    # (the two runs' games diverge once a bf16 Q flips an argmax, so parameters are compared as whole-vector statistics only)
    assert (pb - pa).norm().item() < 1.5 * upd_a


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


def test_config4_whole_node_env_count_on_one_gpu(torch_cuda):
    """configs[4] names 32 768 envs over 8 GPUs (4096 per rank: test_configs_run_at_4096_envs); no 8-GPU node is available to this
    build, so the whole env count runs here on ONE GPU through the same loop: dueling Double-DQN, batch 32, 1 M-slot memory (31 pushes of
    32 768 transitions fill it, so it wraps inside the run)."""
    from dqnflappybird_amd.vecbrain import VecBrain
    vb = VecBrain(32768, algo="double", arch="dueling", batch=32, capacity=1_000_000, observe=2, seed=4)
    vb.run(40, log_every=0)
    assert vb.timeStep == 40 and len(vb.replay) == 1_000_000             # 40 x 32 768 pushes: the memory is full and has wrapped
    assert np.isfinite(vb.last_loss.item()) and vb.env.error_count() == 0
    assert torch_cuda.isfinite(vb.net.store_params()).all()
