/*
 * oracle/fbo_qnet.c -- TEST INFRASTRUCTURE (see fbo.h).
 *
 * CPU restatement of the reference Q-network, its losses and TF-style Adam:
 *   network      BrainDQN.py:119-155 (same graph in BrainDQNNature.py:35-95,
 *                BrainPrioritizedReplyDQN.py:164-232)
 *   dueling head BrainDuelingDQN.py:78-86
 *   losses       BrainDQN.py:159-162 (sum), BrainDQNNature.py:118-119 (mean),
 *                BrainPrioritizedReplyDQN.py:247-251 (IS-weighted mean, abs_errors)
 *   targets      BrainDQN.py:210-215 (Python float64, fed as float32)
 *   optimizer    tf.train.AdamOptimizer(1e-6), BrainDQN.py:163
 * TensorFlow 1.12 is a third-party dependency that is absent here, so this is
 * a restatement of its published op semantics (NHWC / HWIO conv2d with SAME
 * padding, max_pool 2x2/2, NHWC flatten, ApplyAdam:  m += (g-m)(1-b1);
 * v += (g*g-v)(1-b2); var -= m*alpha/(sqrt(v)+eps), alpha = lr*sqrt(1-b2^t)/(1-b1^t)).
 * PARITY UNPINNED vs TF; cross-checked against torch-CPU in tests/.
 *
 * Dot products accumulate in double and round once to float per output, so
 * the oracle sits between any two fp32 summation orders (TF-Eigen's and ours).
 *
 * Flat parameter order (the reference's Variable .. Variable_9 creation order):
 *   W_conv1[8][8][4][32] b_conv1[32] W_conv2[4][4][32][64] b_conv2[64]
 *   W_conv3[3][3][64][64] b_conv3[64] W_fc1[1600][FC] b_fc1[FC]
 *   plain:   W_fc2[FC][A] b_fc2[A]
 *   dueling: W_v[FC][1] b_v[1] W_a[FC][A] b_a[A]
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fbo.h"

/* Smallest distance of any ReLU input from 0 / of any pool winner from its runner-up seen by the last
 * fbo_qnet_forward: gradient checks through ReLUs are only meaningful away from the kinks, the tests use
 * this to pick data with a safe margin. */
static __thread float g_margin = 1e30f, g_margin_nz = 1e30f;
float fbo_qnet_last_margin(void) { return g_margin; }
/* the same without the EXACT ties / zeros: game frames have pool windows over identical pixels (four equal conv outputs: margin 0, the
 * same first-maximum in every arithmetic, and equal contributions whichever position wins), which random frames never have */
float fbo_qnet_last_margin_nonzero(void) { return g_margin_nz; }
static inline void margin(float v) {
    float a = v < 0 ? -v : v;
    if (a < g_margin) g_margin = a;
    if (a > 0 && a < g_margin_nz) g_margin_nz = a;
}

enum { O_W1 = 0, O_B1 = 8192, O_W2 = 8224, O_B2 = 40992, O_W3 = 41056, O_B3 = 77920, O_WF1 = 77984 };

typedef struct { size_t wf1, bf1, wq, bq, wv, bv, n; } offs_t;

static offs_t offs(fbo_qcfg c) {
    offs_t o;
    o.wf1 = O_WF1; o.bf1 = o.wf1 + (size_t)1600 * c.fc;
    size_t p = o.bf1 + c.fc;
    if (c.dueling) { o.wv = p; o.bv = p + c.fc; p = o.bv + 1; } else { o.wv = o.bv = 0; }
    o.wq = p; o.bq = p + (size_t)c.fc * c.actions;
    o.n = o.bq + c.actions;
    return o;
}

size_t fbo_qnet_nparams(fbo_qcfg c) { return offs(c).n; }

/* per-sample activation record kept for backward */
enum { A_H1 = 0, A_P1 = 12800, A_H2 = 16000, A_H3 = 17600, A_HF = 19200 };
size_t fbo_qnet_act_floats(fbo_qcfg c) { return (size_t)A_HF + c.fc; }

static void conv(const float *in, int H, int W, int Ci, const float *wt, const float *bias, int K,
                 int stride, int pad, int Ho, int Wo, int Co, float *out) {
    double acc[64];                                        /* Co <= 64; co innermost so gcc vectorises */
    for (int oy = 0; oy < Ho; oy++)
        for (int ox = 0; ox < Wo; ox++) {
            for (int co = 0; co < Co; co++) acc[co] = 0;
            for (int ky = 0; ky < K; ky++) {
                int iy = oy * stride + ky - pad;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < K; kx++) {
                    int ix = ox * stride + kx - pad;
                    if (ix < 0 || ix >= W) continue;
                    const float *ip = in + ((size_t)iy * W + ix) * Ci;
                    const float *wp = wt + ((size_t)(ky * K + kx) * Ci) * Co;
                    for (int ci = 0; ci < Ci; ci++) {
                        double xv = ip[ci];
                        if (xv == 0) continue;
                        const float *wr = wp + (size_t)ci * Co;
                        for (int co = 0; co < Co; co++) acc[co] += xv * (double)wr[co];
                    }
                }
            }
            for (int co = 0; co < Co; co++) {
                float v = (float)acc[co] + bias[co];       /* conv output is a float tensor, then + b */
                margin(v);
                out[((size_t)oy * Wo + ox) * Co + co] = v > 0 ? v : 0;
            }
        }
}

static void head(const float *P, fbo_qcfg c, offs_t o, const float *hf, float *q) {
    int A = c.actions;
    if (!c.dueling) {
        for (int a = 0; a < A; a++) {
            double acc = 0;
            for (int j = 0; j < c.fc; j++) acc += (double)hf[j] * (double)P[o.wq + (size_t)j * A + a];
            q[a] = (float)acc + P[o.bq + a];
        }
    } else {                                               /* BrainDuelingDQN.py:78-86 */
        double accv = 0;
        for (int j = 0; j < c.fc; j++) accv += (double)hf[j] * (double)P[o.wv + j];
        float V = (float)accv + P[o.bv];
        float adv[16], mean = 0;
        for (int a = 0; a < A; a++) {
            double acc = 0;
            for (int j = 0; j < c.fc; j++) acc += (double)hf[j] * (double)P[o.wq + (size_t)j * A + a];
            adv[a] = (float)acc + P[o.bq + a];
            mean += adv[a];
        }
        mean /= (float)A;
        for (int a = 0; a < A; a++) q[a] = V + (adv[a] - mean);
    }
}

void fbo_qnet_forward(const float *P, fbo_qcfg c, const uint8_t *states, int B, float *q, float *acts) {
    offs_t o = offs(c);
    size_t AF = fbo_qnet_act_floats(c);
    float *x = (float *)malloc(sizeof(float) * 25600);
    float *tmp = acts ? NULL : (float *)malloc(sizeof(float) * AF);
    g_margin = 1e30f; g_margin_nz = 1e30f;
    for (int b = 0; b < B; b++) {
        float *a = acts ? acts + (size_t)b * AF : tmp;
        const uint8_t *s = states + (size_t)b * 25600;
        for (int i = 0; i < 25600; i++) x[i] = (float)s[i];   /* fed as 0/255, not normalised */
        conv(x, 80, 80, 4, P + O_W1, P + O_B1, 8, 4, 2, 20, 20, 32, a + A_H1);
        for (int py = 0; py < 10; py++)                        /* max_pool 2x2 s2, BrainDQN.py:128 */
            for (int px = 0; px < 10; px++)
                for (int ch = 0; ch < 32; ch++) {
                    float m = -INFINITY, m2 = -INFINITY;
                    for (int dy = 0; dy < 2; dy++)
                        for (int dx = 0; dx < 2; dx++) {
                            float v = a[A_H1 + ((2 * py + dy) * 20 + 2 * px + dx) * 32 + ch];
                            if (v > m) { m2 = m; m = v; } else if (v > m2) m2 = v;
                        }
                    if (m > 0) margin(m - m2);             /* how clearly the pool winner wins */
                    a[A_P1 + (py * 10 + px) * 32 + ch] = m;
                }
        conv(a + A_P1, 10, 10, 32, P + O_W2, P + O_B2, 4, 2, 1, 5, 5, 64, a + A_H2);
        conv(a + A_H2, 5, 5, 64, P + O_W3, P + O_B3, 3, 1, 1, 5, 5, 64, a + A_H3);
        for (int j = 0; j < c.fc; j++) {                       /* NHWC flatten -> fc1 */
            double acc = 0;
            for (int i = 0; i < 1600; i++) acc += (double)a[A_H3 + i] * (double)P[o.wf1 + (size_t)i * c.fc + j];
            float v = (float)acc + P[o.bf1 + j];
            margin(v);
            a[A_HF + j] = v > 0 ? v : 0;
        }
        head(P, c, o, a + A_HF, q + (size_t)b * c.actions);
    }
    free(x); free(tmp);
}

/* dW / dB (accumulated into the caller's double buffers) and dIn for one sample */
static void conv_bwd(const float *in, int H, int W, int Ci, const float *wt, int K, int stride, int pad,
                     int Ho, int Wo, int Co, const float *dout /* already relu-masked */,
                     double *dW, double *dB, float *din /* may be NULL; overwritten */) {
    double *dind = din ? (double *)calloc((size_t)H * W * Ci, sizeof(double)) : NULL;
    double g[64];
    for (int oy = 0; oy < Ho; oy++)
        for (int ox = 0; ox < Wo; ox++) {
            const float *dp = dout + ((size_t)oy * Wo + ox) * Co;
            int any = 0;
            for (int co = 0; co < Co; co++) { g[co] = dp[co]; dB[co] += g[co]; any |= dp[co] != 0; }
            if (!any) continue;
            for (int ky = 0; ky < K; ky++) {
                int iy = oy * stride + ky - pad;
                if (iy < 0 || iy >= H) continue;
                for (int kx = 0; kx < K; kx++) {
                    int ix = ox * stride + kx - pad;
                    if (ix < 0 || ix >= W) continue;
                    const float *ip = in + ((size_t)iy * W + ix) * Ci;
                    for (int ci = 0; ci < Ci; ci++) {
                        size_t wb = ((size_t)(ky * K + kx) * Ci + ci) * Co;
                        double xv = ip[ci];
                        if (xv != 0) { double *dw = dW + wb; for (int co = 0; co < Co; co++) dw[co] += xv * g[co]; }
                        if (dind) {
                            const float *wr = wt + wb;
                            double d = 0;
                            for (int co = 0; co < Co; co++) d += (double)wr[co] * g[co];
                            dind[((size_t)iy * W + ix) * Ci + ci] += d;
                        }
                    }
                }
            }
        }
    if (din) { for (size_t i = 0; i < (size_t)H * W * Ci; i++) din[i] = (float)dind[i]; free(dind); }
}

void fbo_qnet_backward(const float *P, fbo_qcfg c, const uint8_t *states, int B, const float *acts,
                       const float *dq, float *grads) {
    offs_t o = offs(c);
    size_t AF = fbo_qnet_act_floats(c);
    int A = c.actions, FC = c.fc;
    double *G = (double *)calloc(o.n, sizeof(double));
    float *x = (float *)malloc(sizeof(float) * 25600);
    float *dhf = (float *)malloc(sizeof(float) * FC);
    float dh3[1600], dh2[1600], dp1[3200], dh1[12800];
    for (int b = 0; b < B; b++) {
        const float *a = acts + (size_t)b * AF;
        const float *g = dq + (size_t)b * A;
        double dadv[16], dV = 0;
        if (c.dueling) {                                   /* Q = V + (A - mean A) */
            double s = 0;
            for (int k = 0; k < A; k++) s += g[k];
            dV = s;
            for (int k = 0; k < A; k++) dadv[k] = (double)g[k] - s / A;
            G[o.bv] += dV;
        } else {
            for (int k = 0; k < A; k++) dadv[k] = g[k];
        }
        for (int k = 0; k < A; k++) G[o.bq + k] += dadv[k];
        for (int j = 0; j < FC; j++) {
            double hj = a[A_HF + j], d = 0;
            for (int k = 0; k < A; k++) {
                G[o.wq + (size_t)j * A + k] += hj * dadv[k];
                d += dadv[k] * (double)P[o.wq + (size_t)j * A + k];
            }
            if (c.dueling) { G[o.wv + j] += hj * dV; d += dV * (double)P[o.wv + j]; }
            dhf[j] = hj > 0 ? (float)d : 0.f;
        }
        for (int j = 0; j < FC; j++) G[o.bf1 + j] += dhf[j];
        for (int i = 0; i < 1600; i++) {
            double hi = a[A_H3 + i], d = 0;
            const float *wr = P + o.wf1 + (size_t)i * FC;
            double *gr = G + o.wf1 + (size_t)i * FC;
            for (int j = 0; j < FC; j++) { gr[j] += hi * (double)dhf[j]; d += (double)dhf[j] * (double)wr[j]; }
            dh3[i] = hi > 0 ? (float)d : 0.f;
        }
        conv_bwd(a + A_H2, 5, 5, 64, P + O_W3, 3, 1, 1, 5, 5, 64, dh3, G + O_W3, G + O_B3, dh2);
        for (int i = 0; i < 1600; i++) if (!(a[A_H2 + i] > 0)) dh2[i] = 0;
        conv_bwd(a + A_P1, 10, 10, 32, P + O_W2, 4, 2, 1, 5, 5, 64, dh2, G + O_W2, G + O_B2, dp1);
        memset(dh1, 0, sizeof(dh1));
        for (int py = 0; py < 10; py++)                     /* MaxPoolGrad: first max in window scan order */
            for (int px = 0; px < 10; px++)
                for (int ch = 0; ch < 32; ch++) {
                    int best = 0; float m = -INFINITY;
                    for (int k = 0; k < 4; k++) {
                        float v = a[A_H1 + ((2 * py + (k >> 1)) * 20 + 2 * px + (k & 1)) * 32 + ch];
                        if (v > m) { m = v; best = k; }
                    }
                    int idx = ((2 * py + (best >> 1)) * 20 + 2 * px + (best & 1)) * 32 + ch;
                    dh1[idx] = a[A_H1 + idx] > 0 ? dp1[(py * 10 + px) * 32 + ch] : 0.f;   /* relu1 */
                }
        const uint8_t *s = states + (size_t)b * 25600;
        for (int i = 0; i < 25600; i++) x[i] = (float)s[i];
        conv_bwd(x, 80, 80, 4, P + O_W1, 8, 4, 2, 20, 20, 32, dh1, G + O_W1, G + O_B1, NULL);
    }
    for (size_t i = 0; i < o.n; i++) grads[i] = (float)G[i];
    free(G); free(x); free(dhf);
}

/* TF 1.12 core/kernels/training_ops.cc ApplyAdam (non-Nesterov), all fp32 */
void fbo_adam_step(float *p, float *m, float *v, const float *g, size_t n, float lr, float b1,
                   float b2, float eps, float *b1pow, float *b2pow) {
    const float alpha = lr * sqrtf(1.f - *b2pow) / (1.f - *b1pow);
    for (size_t i = 0; i < n; i++) {
        m[i] += (g[i] - m[i]) * (1.f - b1);
        v[i] += (g[i] * g[i] - v[i]) * (1.f - b2);
        p[i] -= (m[i] * alpha) / (sqrtf(v[i]) + eps);
    }
    *b1pow *= b1;                                           /* AdamOptimizer._finish */
    *b2pow *= b2;
}

void fbo_dqn_loss(int kind, int B, int A, const float *q, const float *q_next_sel, const uint8_t *action,
                  const float *reward, const uint8_t *terminal, const float *isw, double gamma,
                  float *y, float *loss, float *abs_err, float *dq) {
    double L = 0;
    for (int b = 0; b < B; b++) {
        /* BrainDQN.py:210-215: python float64 arithmetic, then fed to a float32 placeholder */
        /* the reference's rewards are the Python numbers 0.1, 3, -3 (wrapped_flappy_bird.py:95,148,162) */
        double r = (reward[b] == 0.1f) ? 0.1 : (double)reward[b];
        double yd = terminal[b] ? r : r + gamma * (double)q_next_sel[b];
        y[b] = (float)yd;
        float qe = q[(size_t)b * A + action[b]];               /* reduce_sum(Q * onehot) */
        float d = y[b] - qe;
        float w = (kind == 2) ? isw[b] : 1.f;
        L += (double)(w * d * d);
        if (abs_err) abs_err[b] = fabsf(d);
        float scale = (kind == 0) ? 2.f : 2.f / (float)B;
        for (int a = 0; a < A; a++) dq[(size_t)b * A + a] = 0.f;
        dq[(size_t)b * A + action[b]] = -scale * w * d;
    }
    *loss = (float)(kind == 0 ? L : L / B);
}

/* Policy-gradient loss of BrainPolicyGradient.py:96-100 (and the actor of BrainActorCritic.py:96-100):
 * neg_log_prob = softmax_cross_entropy_with_logits(logits = Q, labels = onehot(action)); loss = mean over n_total of neg_log_prob x w.
 * q holds the logits of B of those n_total samples; loss / dq are this chunk's share. */
void fbo_pg_loss(int B, int A, const float *q, const uint8_t *action, const float *w, double n_total, float *loss, float *dq) {
    double L = 0;
    for (int b = 0; b < B; b++) {
        const float *z = q + (size_t)b * A;
        double mx = z[0], se = 0;
        for (int a = 1; a < A; a++) if (z[a] > mx) mx = z[a];
        for (int a = 0; a < A; a++) se += exp((double)z[a] - mx);
        const double nlp = log(se) - ((double)z[action[b]] - mx);
        L += nlp * (double)w[b] / n_total;
        for (int a = 0; a < A; a++)
            dq[(size_t)b * A + a] = (float)((exp((double)z[a] - mx) / se - (a == action[b] ? 1.0 : 0.0)) * (double)w[b] / n_total);
    }
    *loss = (float)L;
}

/* tf.truncated_normal(stddev=0.01): N(0, 0.01) re-drawn beyond 2 sigma; biases 0.01
 * (BrainDQN.py:123-152).  Our own Philox stream 4 + Box-Muller: the reference
 * is unseeded, so there is no draw sequence to match -- only the distribution. */
void fbo_trunc_normal_init(float *P, fbo_qcfg c, uint32_t seed_lo, uint32_t seed_hi) {
    offs_t o = offs(c);
    uint32_t ctr = 0;
    for (size_t i = 0; i < o.n; i++) {
        for (;;) {
            uint32_t r[4];
            fbo_philox4x32(seed_lo, seed_hi, (uint32_t)i, ctr, 4u, 0u, r);
            float u1 = ((r[0] >> 8) + 1) * (1.0f / 16777216.0f), u2 = (r[1] >> 8) * (1.0f / 16777216.0f);
            float z = sqrtf(-2.f * logf(u1)) * cosf(6.28318530717958647692f * u2);
            if (fabsf(z) <= 2.f) { P[i] = 0.01f * z; ctr = 0; break; }
            ctr++;
        }
    }
    size_t bias[] = {O_B1, 32, O_B2, 64, O_B3, 64, o.bf1, (size_t)c.fc, o.bq, (size_t)c.actions};
    for (int k = 0; k < 5; k++) for (size_t i = 0; i < bias[2 * k + 1]; i++) P[bias[2 * k] + i] = 0.01f;
    if (c.dueling) P[o.bv] = 0.01f;
}
