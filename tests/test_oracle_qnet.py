"""Oracle Q-network / losses / TF-Adam vs an independent torch-CPU statement of the same graph
(BrainDQN.py:119-163).  TensorFlow itself is absent: parity vs TF is unpinned (see fbo_qnet.c)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


def split(p, fc=512, A=2, dueling=False):
    o, out = 0, {}
    shapes = [("w1", (8, 8, 4, 32)), ("b1", (32,)), ("w2", (4, 4, 32, 64)), ("b2", (64,)),
              ("w3", (3, 3, 64, 64)), ("b3", (64,)), ("wf1", (1600, fc)), ("bf1", (fc,))]
    shapes += ([("wv", (fc, 1)), ("bv", (1,))] if dueling else []) + [("wq", (fc, A)), ("bq", (A,))]
    for n, s in shapes:
        k = int(np.prod(s))
        out[n] = p[o:o + k].view(s)
        o += k
    assert o == p.numel()
    return out


def torch_forward(p, states, fc=512, A=2, dueling=False):
    """TF semantics in torch: NHWC/HWIO, SAME padding 2/1/1, NHWC flatten."""
    w = split(p, fc, A, dueling)
    x = states.permute(0, 3, 1, 2)
    h = F.relu(F.conv2d(x, w["w1"].permute(3, 2, 0, 1), w["b1"], stride=4, padding=2))
    h = F.max_pool2d(h, 2, 2)
    h = F.relu(F.conv2d(h, w["w2"].permute(3, 2, 0, 1), w["b2"], stride=2, padding=1))
    h = F.relu(F.conv2d(h, w["w3"].permute(3, 2, 0, 1), w["b3"], stride=1, padding=1))
    h = h.permute(0, 2, 3, 1).reshape(h.shape[0], 1600)
    h = F.relu(h @ w["wf1"] + w["bf1"])
    if dueling:
        v = h @ w["wv"] + w["bv"]
        a = h @ w["wq"] + w["bq"]
        return v + (a - a.mean(dim=1, keepdim=True))
    return h @ w["wq"] + w["bq"]


def rand_states(rng, B):
    return (rng.random((B, 80, 80, 4)) < 0.37).astype(np.uint8) * 255


@pytest.mark.parametrize("dueling", [False, True])
def test_forward_backward_vs_torch(oracle, dueling):
    rng = np.random.default_rng(0)
    cfg = oracle.qcfg(512, 2, dueling)
    assert oracle.nparams(cfg) == (899235 if dueling else 898722)
    params = oracle.init_params(cfg, seed=7)
    # scale the weights up so that ReLUs switch and Q is O(1), like a trained net
    params *= 3.0
    B = 6
    s = rand_states(rng, B)
    q, acts = oracle.forward(params, cfg, s, keep=True)
    pt = torch.tensor(params, dtype=torch.float64, requires_grad=True)
    qt = torch_forward(pt, torch.tensor(s, dtype=torch.float64), dueling=dueling)
    np.testing.assert_allclose(q, qt.detach().numpy(), rtol=0, atol=2e-5 * max(1.0, np.abs(q).max()))
    dq = rng.standard_normal((B, 2)).astype(np.float32)
    g = oracle.backward(params, cfg, s, acts, dq)
    (qt * torch.tensor(dq, dtype=torch.float64)).sum().backward()
    gt = pt.grad.numpy()
    scale = np.abs(gt).max()
    np.testing.assert_allclose(g, gt, rtol=2e-4, atol=2e-6 * scale)


def test_init_distribution(oracle):
    """tf.truncated_normal(stddev=0.01) + biases 0.01 (BrainDQN.py:123-152)."""
    cfg = oracle.qcfg()
    p = oracle.init_params(cfg, seed=0)
    w = p[77984:77984 + 1600 * 512]
    assert np.abs(w).max() <= 0.02 + 1e-9
    assert abs(w.std() - 0.01 * 0.8796) < 2e-4          # std of N(0,1) truncated at 2 sigma
    assert abs(w.mean()) < 5e-5
    assert np.all(p[8192:8224] == np.float32(0.01)) and np.all(p[-2:] == np.float32(0.01))
    assert not np.array_equal(p, oracle.init_params(cfg, seed=1))


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_loss_variants(oracle, kind):
    """sum (BrainDQN.py:162) / mean (BrainDQNNature.py:119) / IS-weighted mean + abs_errors
    (BrainPrioritizedReplyDQN.py:249-251); y per BrainDQN.py:210-215."""
    rng = np.random.default_rng(kind)
    B = 32
    q = rng.standard_normal((B, 2)).astype(np.float32)
    qn = rng.standard_normal(B).astype(np.float32)
    a = rng.integers(0, 2, B).astype(np.uint8)
    r = rng.choice(np.array([0.1, 3, -3], np.float32), B)
    t = (r == -3).astype(np.uint8)
    w = rng.random(B).astype(np.float32)
    y, loss, ae, dq = oracle.dqn_loss(kind, q, qn, a, r, t, isw=w)
    # the reference's python-float arithmetic
    rp = [0.1 if x == np.float32(0.1) else float(x) for x in r]
    # (NumPy 1.x, the reference's era: python float * np.float32 scalar -> float64; NEP 50 changed that)
    y_ref = np.array([rp[i] if t[i] else rp[i] + 0.99 * float(qn[i]) for i in range(B)]).astype(np.float32)
    assert np.array_equal(y, y_ref)
    qt = torch.tensor(q, dtype=torch.float64, requires_grad=True)
    qe = (qt * F.one_hot(torch.tensor(a.astype(np.int64)), 2)).sum(1)
    d = torch.tensor(y_ref, dtype=torch.float64) - qe
    L = (d ** 2).sum() if kind == 0 else ((d ** 2).mean() if kind == 1 else (torch.tensor(w, dtype=torch.float64) * d ** 2).mean())
    L.backward()
    np.testing.assert_allclose(loss, L.item(), rtol=1e-6)
    np.testing.assert_allclose(dq, qt.grad.numpy(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(ae, np.abs(d.detach().numpy()), rtol=1e-6, atol=1e-9)


def test_tf_adam_formula(oracle):
    """ApplyAdam: alpha = lr*sqrt(1-b2^t)/(1-b1^t); eps OUTSIDE the sqrt, not PyTorch's placement."""
    rng = np.random.default_rng(3)
    n = 1000
    p = rng.standard_normal(n).astype(np.float32)
    p64, m, v = p.astype(np.float64), np.zeros(n), np.zeros(n)
    opt = oracle.Adam(n, lr=1e-3)
    for t in range(1, 6):
        g = rng.standard_normal(n).astype(np.float32)
        opt.step(p, g)
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g.astype(np.float64) ** 2
        alpha = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        p64 = p64 - alpha * m / (np.sqrt(v) + 1e-8)
        np.testing.assert_allclose(p, p64, rtol=0, atol=2e-6)
    assert abs(opt.b1p.value - 0.9 ** 6) < 1e-6
