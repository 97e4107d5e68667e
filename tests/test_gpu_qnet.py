"""HIP Q-network (through the C ABI) vs the oracle: Q within 1e-4 (the north-star tolerance, fp32),
gradients, losses / targets, TF-style Adam, target sync, epsilon-greedy acting."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

Q_ATOL = 1e-4          # BASELINE.json north_star: "Q-values ... within 1e-4 fp32"


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def rand_states(rng, B):
    return (rng.random((B, 80, 80, 4)) < 0.37).astype(np.uint8) * 255


def trained_like_params(oracle, cfg, seed):
    """sigma=0.01 init gives Q ~ 0.01 everywhere; scale the weights so activations are O(1..10)
    like a trained net, which is the regime the 1e-4 bound has to hold in."""
    p = oracle.init_params(cfg, seed=seed)
    p *= 3.0
    return p


@pytest.mark.parametrize("dueling", [False, True])
@pytest.mark.parametrize("B", [1, 7, 32, 100])
def test_forward_q_within_1e4(torch_cuda, oracle, dueling, B):
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(B + 100 * dueling)
    cfg = oracle.qcfg(512, 2, dueling)
    p = trained_like_params(oracle, cfg, seed=B)
    net = QNet(2, 512, "dueling" if dueling else "plain", max_batch=128)
    assert net.n_params == oracle.nparams(cfg)
    net.load_params(p)
    s = rand_states(rng, B)
    q = net.forward(torch.from_numpy(s).cuda()).cpu().numpy()
    want = oracle.forward(p, cfg, s)
    assert np.abs(want).max() > 0.05            # the comparison is not vacuous
    np.testing.assert_allclose(q, want, rtol=0, atol=Q_ATOL)
    # store_params round trip
    assert np.array_equal(net.store_params().cpu().numpy(), p)


@pytest.mark.parametrize("dueling,N", [(False, 256), (False, 333), (True, 1000)])
def test_large_batch_forward_split_bf16_path(torch_cuda, oracle, dueling, N):
    """>= 256 states, forward only: the split-bf16 path (three bf16 planes per fp32 value, six bf16 MFMAs per
    product).  Same 1e-4 bound against the oracle, and within 2e-6 relative of the fp32-MFMA kernels that the
    same states take when they are fed in slices of 128."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(N)
    cfg = oracle.qcfg(512, 2, dueling)
    p = trained_like_params(oracle, cfg, seed=N)
    net = QNet(2, 512, "dueling" if dueling else "plain", max_batch=N)
    net.load_params(p)
    s = rand_states(rng, N)
    sd = torch.from_numpy(s).cuda()
    q = net.forward(sd).cpu().numpy()
    want = oracle.forward(p, cfg, s)
    assert np.abs(want).max() > 0.05
    np.testing.assert_allclose(q, want, rtol=0, atol=Q_ATOL)
    q32 = np.concatenate([net.forward(sd[i:i + 128]).cpu().numpy() for i in range(0, N, 128)])
    assert np.abs(q - q32).max() <= 2e-6 * np.abs(want).max()
    # the error of both against the fp64-accumulating oracle is of the same order
    assert np.abs(q - want).max() <= 4 * np.abs(q32 - want).max() + 1e-7


def test_bf16_inference_mode(torch_cuda, oracle):
    """BASELINE config 3's dtype: plain bf16 operands (hi planes only), fp32 accumulation, on the >= 256-state
    forward path.  bf16 keeps 8 significant bits, so the bound is relative (3 % of the Q scale); the greedy
    actions still agree with the fp32 ones except where the two Q-values are closer than that."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(77)
    cfg = oracle.qcfg()
    p = trained_like_params(oracle, cfg, seed=3)
    net = QNet(max_batch=512)
    net.load_params(p)
    s = rand_states(rng, 512)
    sd = torch.from_numpy(s).cuda()
    q32 = net.forward(sd).cpu().numpy()
    net.set_inference_dtype("bf16")
    q16 = net.forward(sd).cpu().numpy()
    net.set_inference_dtype("f32")
    assert np.array_equal(net.forward(sd).cpu().numpy(), q32)          # switching back restores the fp32 path
    scale = np.abs(q32).max()
    err = np.abs(q16 - q32).max()
    assert 0 < err <= 0.03 * scale
    margin = np.abs(q32[:, 0] - q32[:, 1])
    agree = q16.argmax(1) == q32.argmax(1)
    assert agree[margin > 0.06 * scale].all()
    with pytest.raises(ValueError):
        from dqnflappybird_amd import _lib as L
        L.check(L.lib().fb_qnet_set_inference_dtype(net.h, 7), "fb_qnet_set_inference_dtype")


def oracle_train_grads(oracle, cfg, p_on, p_tg, algo, s, a, r, s2, t, isw):
    q, acts = oracle.forward(p_on, cfg, s, keep=True)
    if algo == "dqn":
        qn = oracle.forward(p_on, cfg, s2).max(1)
    elif algo == "double":
        am = oracle.forward(p_on, cfg, s2).argmax(1)
        qn = oracle.forward(p_tg, cfg, s2)[np.arange(len(am)), am]
    else:
        qn = oracle.forward(p_tg, cfg, s2).max(1)
    kind = {"dqn": 0, "nature": 1, "double": 1, "per": 2}[algo]
    y, loss, ae, dq = oracle.dqn_loss(kind, q, qn, a, r, t, isw=isw)
    g = oracle.backward(p_on, cfg, s, acts, dq)
    return y, loss, ae, g


@pytest.mark.parametrize("algo,dueling,B", [("dqn", False, 32), ("nature", False, 32), ("double", False, 32),
                                            ("per", False, 32), ("nature", True, 32), ("dqn", False, 5),
                                            ("double", True, 64)])
def test_train_step_gradients_match_oracle(torch_cuda, oracle, algo, dueling, B):
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    # A gradient check through ReLUs / max-pools is only meaningful away from their kinks: if one of the
    # ~100 000 unit inputs per layer sits within rounding distance of 0 (or a pool winner within rounding
    # distance of its runner-up), the fp32 device and the fp64-accumulating oracle disagree on that unit's
    # mask and its whole gradient column differs.  Draw data until the oracle reports a safe margin.
    import zlib
    cfg = oracle.qcfg(512, 2, dueling)
    p_on, p_tg = trained_like_params(oracle, cfg, 1), trained_like_params(oracle, cfg, 2)
    net = QNet(2, 512, "dueling" if dueling else "plain", max_batch=64)
    net.load_params(p_on, 0)
    net.load_params(p_tg, 1)
    for attempt in range(50):
        rng = np.random.default_rng(zlib.crc32(f"{algo}-{dueling}-{B}-{attempt}".encode()))
        s, s2 = rand_states(rng, B), rand_states(rng, B)
        oracle.forward(p_on, cfg, s)
        if oracle.last_margin() > 2e-5:          # device vs oracle pre-activations differ by ~1e-6
            break
    else:
        pytest.fail("no kink-free batch found")
    a = rng.integers(0, 2, B).astype(np.uint8)
    r = rng.choice(np.array([0.1, 3, -3], np.float32), B, p=[0.8, 0.1, 0.1])
    t = (r == -3).astype(np.uint8)
    isw = rng.random(B).astype(np.float32) if algo == "per" else None
    d = lambda x: None if x is None else torch.from_numpy(x).cuda()
    grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda")
    loss, ae, y = net.train_step(algo, d(s), d(a), d(r), d(s2), d(t), isw=d(isw), flat_grad=grad)
    y0, loss0, ae0, g0 = oracle_train_grads(oracle, cfg, p_on, p_tg, algo, s, a, r, s2, t, isw)
    np.testing.assert_allclose(y.cpu().numpy(), y0, rtol=0, atol=Q_ATOL)
    np.testing.assert_allclose(ae.cpu().numpy(), ae0, rtol=0, atol=2 * Q_ATOL)
    np.testing.assert_allclose(loss.item(), loss0, rtol=1e-4, atol=1e-6)
    g = grad.cpu().numpy()
    # per-tensor comparison so that a small tensor cannot hide behind a large one
    bounds = [0, 8192, 8224, 40992, 41056, 77920, 77984, 77984 + 1600 * 512, 77984 + 1600 * 512 + 512, net.n_params]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        scale = np.abs(g0[lo:hi]).max()
        assert scale > 0
        np.testing.assert_allclose(g[lo:hi], g0[lo:hi], rtol=2e-3, atol=2e-5 * scale, err_msg=f"params[{lo}:{hi}]")
    # parameters untouched in gradient-only mode
    assert np.array_equal(net.store_params().cpu().numpy(), p_on)


def test_adam_updates_match_oracle_over_10_steps(torch_cuda, oracle):
    """fused train step (Adam applied on the device) vs oracle forward/backward/fbo_adam_step, 10 steps.  The gradients of
    the two sides differ by fp32 vs fp64 accumulation, so this end-to-end check is relative; the Adam ARITHMETIC itself
    (epsilon placement, bias correction, beta powers) is held bit for bit by test_adam_arithmetic_bit_exact below."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(11)
    cfg = oracle.qcfg()
    p = trained_like_params(oracle, cfg, 5)
    net = QNet(max_batch=32)
    net.load_params(p, 0)
    net.set_hparams(lr=1e-4)                      # larger than the reference's 1e-6 so the step is visible in fp32
    opt = oracle.Adam(p.size, lr=1e-4)
    d = lambda x: torch.from_numpy(x).cuda()
    p_ref = p.copy()
    for step in range(10):
        B = 32
        for _ in range(50):                       # stay away from ReLU / pool kinks (see the gradient test)
            s, s2 = rand_states(rng, B), rand_states(rng, B)
            oracle.forward(p_ref, cfg, s)
            if oracle.last_margin() > 2e-5:
                break
        a = rng.integers(0, 2, B).astype(np.uint8)
        r = rng.choice(np.array([0.1, 3, -3], np.float32), B)
        t = (r == -3).astype(np.uint8)
        net.train_step("dqn", d(s), d(a), d(r), d(s2), d(t))
        _, _, _, g = oracle_train_grads(oracle, cfg, p_ref, p_ref, "dqn", s, a, r, s2, t, None)
        opt.step(p_ref, g)
        got = net.store_params().cpu().numpy()
        # Adam normalises the step to ~lr per element: compare against that scale
        assert np.abs(got - p_ref).max() < 0.05 * 1e-4 * (step + 1), step
        assert np.abs(got - p_ref).mean() < 2e-4 * 1e-4 * (step + 1), step      # and the typical element is far closer
    m, v, pows = net.adam_state()
    np.testing.assert_allclose(pows, [opt.b1p.value, opt.b2p.value], rtol=1e-6)
    np.testing.assert_allclose(m.cpu().numpy(), opt.m, rtol=5e-3, atol=1e-6 * np.abs(opt.m).max())
    np.testing.assert_allclose(v.cpu().numpy(), opt.v, rtol=1e-2, atol=1e-6 * np.abs(opt.v).max())


@pytest.mark.parametrize("lr", [1e-6, 1e-4])          # the reference's learning rate (BrainDQN.py:163) and a visible one
def test_adam_arithmetic_bit_exact(torch_cuda, oracle, lr):
    """TF ApplyAdam on the device == fbo_adam_step on the SAME gradients, bit for bit: parameters, m, v and both beta
    powers after every one of 6 updates.  The gradients are the device's own (gradient-only train step), so nothing but
    the optimizer arithmetic is compared: alpha = lr * sqrt(1 - b2^t) / (1 - b1^t), m += (g - m)(1 - b1),
    v += (g^2 - v)(1 - b2), p -= m * alpha / (sqrt(v) + eps) -- epsilon OUTSIDE the square root, no PyTorch-style bias-corrected
    denominators.  Also covers both ways an update reaches the parameters: fb_qnet_apply_adam and the fused step."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(21)
    cfg = oracle.qcfg()
    p0 = trained_like_params(oracle, cfg, 6)
    nets = [QNet(max_batch=32), QNet(max_batch=32)]
    for n in nets:
        n.load_params(p0, 0)
        n.set_hparams(lr=lr)
    opt = oracle.Adam(p0.size, lr=lr)
    p_ref = p0.copy()
    grad = torch.zeros(nets[0].n_params, dtype=torch.float32, device="cuda")
    d = lambda x: torch.from_numpy(x).cuda()
    for step in range(6):
        s, s2 = rand_states(rng, 32), rand_states(rng, 32)
        a = rng.integers(0, 2, 32).astype(np.uint8)
        r = rng.choice(np.array([0.1, 3, -3], np.float32), 32)
        t = (r == -3).astype(np.uint8)
        nets[0].train_step("dqn", d(s), d(a), d(r), d(s2), d(t), flat_grad=grad)
        nets[0].apply_adam(grad)
        nets[1].train_step("dqn", d(s), d(a), d(r), d(s2), d(t))             # fused: Adam inside the step
        opt.step(p_ref, grad.cpu().numpy())
        for n in nets:
            m, v, pows = n.adam_state()
            assert np.array_equal(pows, np.array([opt.b1p.value, opt.b2p.value], np.float32)), step
            assert np.array_equal(m.cpu().numpy(), opt.m), step
            assert np.array_equal(v.cpu().numpy(), opt.v), step
            assert np.array_equal(n.store_params().cpu().numpy(), p_ref), step
    assert not np.array_equal(p_ref, p0)
    if lr == 1e-6:                                # the first update moves a weight by ~lr: far above one ulp of a 0.03-sized weight
        assert np.abs(p_ref - p0).max() > 3e-6


def test_act_epsilon_stream_is_the_documented_philox(torch_cuda, oracle):
    """the epsilon-greedy draws of fb_qnet_act are Philox4x32-10 with key = seed, counter = (env, step lo, FB_STREAM_EPS = 1,
    step hi): word 0 -> random.random(), word 1 -> randrange(A) (fb_head.h), pinned here against the oracle's Philox for
    seeds / steps that exercise all four 32-bit words."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(5)
    net = QNet(max_batch=300)
    net.init_params(seed=2)
    N = 300
    sd = torch.from_numpy(rand_states(rng, N)).cuda()
    greedy = net.act(sd, epsilon=0.0).cpu().numpy().copy()
    for seed, step in [(0, 0), (7, 123456), ((9 << 32) | 5, 3), (1, (1 << 32) + 17), ((0xDEADBEEF << 32) | 0x12345678, (77 << 32) | 99)]:
        for eps in (1.0, 0.25):
            got = net.act(sd, epsilon=eps, seed=seed, step=step).cpu().numpy()
            want = greedy.copy()
            for e in range(N):
                ph = oracle.philox(seed & 0xFFFFFFFF, seed >> 32, e, step & 0xFFFFFFFF, 1, step >> 32)
                u = np.float32(int(ph[0]) >> 8) * np.float32(1.0 / 16777216.0)
                if u <= np.float32(eps):
                    want[e] = (int(ph[1]) * 2) >> 32
            assert np.array_equal(got, want), (seed, step, eps)


def test_data_parallel_path_equals_fused_path(torch_cuda, oracle):
    """train_step(flat_grad) + apply_adam == train_step() with Adam fused, bit for bit."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(3)
    cfg = oracle.qcfg()
    p = trained_like_params(oracle, cfg, 9)
    nets = [QNet(max_batch=32), QNet(max_batch=32)]
    for n in nets:
        n.load_params(p, 0)
        n.load_params(p, 1)
    d = lambda x: torch.from_numpy(x).cuda()
    grad = torch.zeros(nets[0].n_params, dtype=torch.float32, device="cuda")
    for step in range(3):
        s, s2 = rand_states(rng, 32), rand_states(rng, 32)
        a = rng.integers(0, 2, 32).astype(np.uint8)
        r = rng.choice(np.array([0.1, 3, -3], np.float32), 32)
        t = (r == -3).astype(np.uint8)
        nets[0].train_step("nature", d(s), d(a), d(r), d(s2), d(t))
        nets[1].train_step("nature", d(s), d(a), d(r), d(s2), d(t), flat_grad=grad)
        nets[1].apply_adam(grad)
        assert torch.equal(nets[0].store_params(), nets[1].store_params())
    # and the run is reproducible bit for bit (no atomics in the reductions)
    assert torch.equal(nets[0].adam_state()[0], nets[1].adam_state()[0])


def test_target_sync_and_independent_init(torch_cuda):
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    net = QNet(max_batch=8)
    net.init_params(seed=1, which=0)
    net.init_params(seed=2, which=1)               # BrainDQNNature: target net initialised independently
    p0, p1 = net.store_params(0), net.store_params(1)
    assert not torch.equal(p0, p1)
    w = p0[77984:77984 + 1600 * 512]
    assert w.abs().max() <= 0.02 + 1e-7 and abs(w.std().item() - 0.008796) < 2e-4
    assert torch.all(p0[8192:8224] == 0.01) and torch.all(p0[-2:] == 0.01)
    net.sync_target()
    assert torch.equal(net.store_params(0), net.store_params(1))


def test_act_epsilon_greedy(torch_cuda, oracle):
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(8)
    cfg = oracle.qcfg()
    p = trained_like_params(oracle, cfg, 4)
    net = QNet(max_batch=2048)
    net.load_params(p)
    N = 2048
    s = rand_states(rng, N)
    sd = torch.from_numpy(s).cuda()
    act, q = net.act(sd, epsilon=0.0, seed=1, step=0, want_q=True)
    q = q.cpu().numpy()
    want = oracle.forward(p, cfg, s)                                  # ALL 2048 rows of the >= 256-state path against the oracle
    np.testing.assert_allclose(q, want, rtol=0, atol=Q_ATOL)
    assert net.overflow_count() == 0
    assert np.array_equal(act.cpu().numpy(), q.argmax(1))            # greedy == np.argmax
    a1 = net.act(sd, epsilon=1.0, seed=1, step=5).cpu().numpy().copy()
    a2 = net.act(sd, epsilon=1.0, seed=1, step=6).cpu().numpy().copy()
    a1b = net.act(sd, epsilon=1.0, seed=1, step=5).cpu().numpy().copy()
    assert 0.4 < a1.mean() < 0.6 and not np.array_equal(a1, a2) and np.array_equal(a1, a1b)
    a3 = net.act(sd, epsilon=0.03, seed=1, step=7).cpu().numpy()     # INITIAL_EPSILON
    frac = (a3 != q.argmax(1)).mean()
    assert 0.003 < frac < 0.04                                        # ~ eps/2 of the envs deviate


def test_fused_acting_trunk_all_rows_against_the_oracle(torch_cuda, oracle):
    """The acting path VecBrain runs (nibble states -> conv1 + conv2 + conv3 in one launch -> fc1 on K slices -> head) compared with
    the oracle DIRECTLY on every one of 1027 states (a last workgroup of three), not through its bit-identity with the u8 path."""
    torch = torch_cuda
    from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay
    N = 1027
    cfg = oracle.qcfg()
    p = trained_like_params(oracle, cfg, 6)
    env, rep, net = VecGameState(N, seed=9), VecReplay(5000, N), QNet(max_batch=N)
    net.load_params(p)
    nib = env.track_state()
    env.observe()
    rep.reset(env.frame_bits)
    rng = np.random.default_rng(2)
    for _ in range(7):                                                # (distinct frames in the four stack positions)
        acts = torch.from_numpy((rng.random(N) < 0.3).astype(np.uint8)).cuda()
        env.frame_step(acts, want_u8=False)
        rep.push(env.frame_bits, acts, env.reward, env.terminal)
    _, q = net.act_nib(nib, 0.0, want_q=True)
    want = oracle.forward(p, cfg, rep.current_state().cpu().numpy())
    np.testing.assert_allclose(q.cpu().numpy(), want, rtol=0, atol=Q_ATOL)
    assert net.overflow_count() == 0


@pytest.mark.parametrize("N", [40, 1024])
def test_activations_beyond_the_two_plane_range_are_reported_not_silently_wrong(torch_cuda, oracle, N):
    """the path labelled f32 carries activations as two fp16 planes, unscaled -- exact for |x| < 32768 only,
    where TensorFlow's fp32 (BrainDQN.py:119-155) has no such limit.  A net with weights x 40 (pooled conv1 outputs ~ 1e5, conv2 /
    conv3 outputs far beyond) used to come back as inf / NaN or as finite wrong numbers.  Now: the library COUNTS the event
    (fb_qnet_overflow_count), QNet.check_range raises, and the same net in bf16 mode (fp32's exponent range) still tracks the oracle
    on the Q scale.  In range (weights x 1) the counter stays 0."""
    torch = torch_cuda
    from dqnflappybird_amd import _lib as L
    from dqnflappybird_amd.vec import QNet
    rng = np.random.default_rng(5)
    cfg = oracle.qcfg()
    p1 = trained_like_params(oracle, cfg, 3)
    s = rand_states(rng, N)
    sd = torch.from_numpy(s).cuda()
    net = QNet(max_batch=max(N, 64))
    net.load_params(p1)
    net.forward(sd)
    assert net.overflow_count() == 0
    net.check_range()
    big = (p1 * 40.0).astype(np.float32)
    want = oracle.forward(big, cfg, s)
    assert np.abs(want).max() > 1e6                                   # (the oracle's own activations pass 65504 from conv1 / conv2 on)
    net.load_params(big)
    net.forward(sd)
    n = net.overflow_count()
    assert n > 0
    with pytest.raises(L.FbError, match="32768"):
        net.check_range()
    assert net.overflow_count(reset=True) == n and net.overflow_count() == 0
    if N >= 256:                                                      # the way out the message names: bf16 planes have fp32's exponent range
        net.set_inference_dtype("bf16")
        q = net.forward(sd).cpu().numpy()
        assert np.isfinite(q).all() and net.overflow_count() == 0
        assert np.abs(q - want).max() < 0.03 * np.abs(want).max()     # (the bf16 bound of tests/test_gpu_configs.py, on the Q scale)


# ---------------------------------------------------------------------------------------------------------------------------
# The reference's OWN numeric regime: weights as BrainDQN.py:123-152 draws them (truncated normal, sigma = 0.01, biases 0.01 --
# where a run sits for its first millions of steps at lr 1e-6), the mean losses of BrainDQNNature.py:119 /
# BrainPrioritizedReplyDQN.py:249-251 (PER: x ISWeights << 1), rewards 0.1 / 3 / -3.  There the conv-layer data gradients are
# 1e-6 .. 1e-9, far below fp16's normal range: the two-plane fp16 operands of the backward kernels are pre-scaled by exact powers
# of two (fb_qnet.hip pow2_scale).  The yardstick is what PLAIN fp32 arithmetic achieves on the same problem: the same backward pass
# in torch-CPU float32 (fp32 products, fp32 accumulation), both measured against the fp64-accumulating oracle, per tensor.
TENSORS = [("W_conv1", 0, 8192), ("b_conv1", 8192, 8224), ("W_conv2", 8224, 40992), ("b_conv2", 40992, 41056),
           ("W_conv3", 41056, 77920), ("b_conv3", 77920, 77984), ("W_fc1", 77984, 77984 + 1600 * 512),
           ("b_fc1", 77984 + 1600 * 512, 77984 + 1600 * 512 + 512), ("head", 77984 + 1600 * 512 + 512, None)]
FP32_FACTOR = 3.0       # a tensor's error may be at most this many times plain fp32's own error on it (measured: 0.3 .. 1.4; round 2's unscaled planes: up to 64) ...
FP32_FLOOR = 3e-7       # ... where that error is itself not below a few fp32 roundings (relative L2)


def torch_fp32_grads(params, s, dq, dueling):
    """the backward pass of sum(Q * dq) in plain float32 on the CPU (tests/test_oracle_qnet.py's statement of the graph)"""
    import torch
    from tests.test_oracle_qnet import torch_forward
    torch.set_num_threads(8)
    pt = torch.tensor(params, dtype=torch.float32, requires_grad=True)
    q = torch_forward(pt, torch.tensor(s, dtype=torch.float32), dueling=dueling)
    (q * torch.tensor(dq, dtype=torch.float32)).sum().backward()
    return pt.grad.numpy()


def _regime_report(lines):
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "grad_regime.txt"), "a") as f:
            f.write("\n".join(lines) + "\n")


@pytest.mark.parametrize("algo,dueling,B", [("nature", False, 32), ("per", False, 32), ("dqn", False, 32), ("nature", True, 32),
                                            ("double", False, 256)])
def test_train_step_gradients_in_the_reference_regime(torch_cuda, oracle, algo, dueling, B):
    torch = torch_cuda
    import zlib
    from dqnflappybird_amd.vec import QNet
    cfg = oracle.qcfg(512, 2, dueling)
    p_on, p_tg = oracle.init_params(cfg, seed=1), oracle.init_params(cfg, seed=2)       # UNSCALED: sigma 0.01, biases 0.01
    net = QNet(2, 512, "dueling" if dueling else "plain", max_batch=B)
    net.load_params(p_on, 0)
    net.load_params(p_tg, 1)
    # kink-free data (see test_train_step_gradients_match_oracle); pre-activations are O(0.01 .. 25) here and the device's differ
    # from the oracle's by ~1e-7 of that
    need = 2e-6 if B <= 32 else 3e-7
    for attempt in range(40 if B <= 32 else 8):
        rng = np.random.default_rng(zlib.crc32(f"regime-{algo}-{dueling}-{B}-{attempt}".encode()))
        s, s2 = rand_states(rng, B), rand_states(rng, B)
        oracle.forward(p_on, cfg, s)
        if oracle.last_margin() > need:
            break
    else:
        pytest.fail("no kink-free batch found")
    a = rng.integers(0, 2, B).astype(np.uint8)
    r = rng.choice(np.array([0.1, 3, -3], np.float32), B, p=[0.8, 0.1, 0.1])
    t = (r == -3).astype(np.uint8)
    isw = (0.02 + 0.2 * rng.random(B)).astype(np.float32) if algo == "per" else None       # ISWeights << 1, like early in a run (beta 0.4)
    d = lambda x: None if x is None else torch.from_numpy(x).cuda()
    grad = torch.zeros(net.n_params, dtype=torch.float32, device="cuda")
    loss, ae, y = net.train_step(algo, d(s), d(a), d(r), d(s2), d(t), isw=d(isw), flat_grad=grad)
    y0, loss0, ae0, g0 = oracle_train_grads(oracle, cfg, p_on, p_tg, algo, s, a, r, s2, t, isw)
    np.testing.assert_allclose(y.cpu().numpy(), y0, rtol=0, atol=Q_ATOL)
    np.testing.assert_allclose(loss.item(), loss0, rtol=1e-5)
    # plain fp32 on the same dQ
    q = oracle.forward(p_on, cfg, s)
    qn = (oracle.forward(p_on, cfg, s2).max(1) if algo == "dqn" else
          oracle.forward(p_tg, cfg, s2)[np.arange(B), oracle.forward(p_on, cfg, s2).argmax(1)] if algo == "double" else
          oracle.forward(p_tg, cfg, s2).max(1))
    _, _, _, dq = oracle.dqn_loss({"dqn": 0, "nature": 1, "double": 1, "per": 2}[algo], q, qn, a, r, t, isw=isw)
    g32 = torch_fp32_grads(p_on, s, dq, dueling)
    g = grad.cpu().numpy()
    lines = [f"[{algo} dueling={dueling} B={B}]  max|dQ| {np.abs(dq).max():.3e}"]
    worst = 0.0
    for name, lo, hi in TENSORS:
        ref = g0[lo:hi].astype(np.float64)
        n0 = np.linalg.norm(ref)
        assert n0 > 0
        e_dev = np.linalg.norm(g[lo:hi] - ref) / n0
        e_f32 = np.linalg.norm(g32[lo:hi] - ref) / n0
        m_dev = np.abs(g[lo:hi] - ref).max() / np.abs(ref).max()
        lines.append(f"  {name:8s} |g|max {np.abs(ref).max():.2e}  relL2 device {e_dev:.2e}  plain fp32 {e_f32:.2e}  ratio {e_dev / max(e_f32, 1e-30):5.2f}   max-err/max {m_dev:.2e}")
        worst = max(worst, e_dev / max(e_f32, FP32_FLOOR))
    _regime_report(lines)
    for name, lo, hi in TENSORS:
        ref = g0[lo:hi].astype(np.float64)
        n0 = np.linalg.norm(ref)
        e_dev = np.linalg.norm(g[lo:hi] - ref) / n0
        e_f32 = np.linalg.norm(g32[lo:hi] - ref) / n0
        assert e_dev <= FP32_FACTOR * max(e_f32, FP32_FLOOR), "\n".join(lines)
