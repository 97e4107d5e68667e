/*
 * oracle/fbo_loop.c -- TEST INFRASTRUCTURE (see fbo.h): the CPU baseline.
 *
 * Single-env restatement of the reference's training loop, used only by the
 * `cpu_baseline` leg of bench.py ("kind": "port"):
 *   FlappyBirdDQN.py:60-76   driver: getAction -> frame_step -> preprocess -> setPerception
 *   BrainDQN.py:99-116       epsilon-greedy on a batch-1 forward
 *   BrainDQN.py:66-96        4-frame stack, deque store, train once onlineTimeStep > OBSERVE
 *   BrainDQN.py:195-223      random.sample(32) -> Q(s') with the same net -> y -> sum loss -> Adam
 * The reference's 30 FPS sleep (game/wrapped_flappy_bird.py:179) is NOT
 * reproduced: this is the compute-bound bound of that loop on one host core.
 */
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "fbo.h"

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int fbo_reference_loop(int observe_steps, int train_steps, int replay_cap, uint32_t seed,
                       fbo_loop_result *out) {
    const int B = 32;
    fbo_qcfg c = {512, 2, 0};
    size_t NP = fbo_qnet_nparams(c), AF = fbo_qnet_act_floats(c);
    int total = observe_steps + train_steps + 1;
    int cap = replay_cap < total ? replay_cap : total;
    if (cap < B) return -1;
    float *P = (float *)malloc(sizeof(float) * NP), *G = (float *)malloc(sizeof(float) * NP);
    float *M = (float *)calloc(NP, sizeof(float)), *V = (float *)calloc(NP, sizeof(float));
    float b1p = 0.9f, b2p = 0.999f;
    fbo_trunc_normal_init(P, c, seed, 0);
    /* states[t % (cap+1)] : transition t = (states[t], a[t], r[t], states[t+1], term[t]) */
    uint8_t *states = (uint8_t *)malloc((size_t)(cap + 1) * 25600);
    uint8_t *act = (uint8_t *)malloc(cap + 1), *term = (uint8_t *)malloc(cap + 1);
    float *rew = (float *)malloc(sizeof(float) * (cap + 1));
    uint8_t *sb = (uint8_t *)malloc((size_t)B * 25600), *s2b = (uint8_t *)malloc((size_t)B * 25600);
    float *acts = (float *)malloc(sizeof(float) * AF * B);
    uint8_t frame[6400];
    fbo_mt mt;
    fbo_mt_seed_python(&mt, seed);
    fbo_env env;
    fbo_env_init(&env, seed, 0, 0, NULL, 0, 0);

    float r; int te, sc;
    fbo_env_step(&env, 0, &r, &te, &sc);                   /* FlappyBirdDQN.py:65-69 */
    fbo_env_frame80(&env, frame);
    uint8_t *s0 = states;
    for (int i = 0; i < 6400; i++) for (int k = 0; k < 4; k++) s0[i * 4 + k] = frame[i];

    double eps = 0.03, t0 = now_s(), t_train0 = 0;
    int64_t n_tr = 0, grad_steps = 0;
    for (int t = 0; t < observe_steps + train_steps; t++) {
        if (t == observe_steps) t_train0 = now_s();
        const uint8_t *cur = states + (size_t)(t % (cap + 1)) * 25600;
        uint8_t *nxt = states + (size_t)((t + 1) % (cap + 1)) * 25600;
        float q[2];
        fbo_qnet_forward(P, c, cur, 1, q, NULL);           /* getAction */
        int a;
        if (fbo_mt_random(&mt) <= eps) a = (int)fbo_py_randbelow(&mt, 2);
        else a = q[1] > q[0] ? 1 : 0;
        if (eps > 0 && t > observe_steps) eps -= 0.03 / 1e6;
        fbo_env_step(&env, a, &r, &te, &sc);
        fbo_env_frame80(&env, frame);                      /* full render + preprocess */
        for (int i = 0; i < 6400; i++) {                   /* BrainDQN.py:68 */
            nxt[i * 4 + 0] = cur[i * 4 + 1]; nxt[i * 4 + 1] = cur[i * 4 + 2];
            nxt[i * 4 + 2] = cur[i * 4 + 3]; nxt[i * 4 + 3] = frame[i];
        }
        act[t % (cap + 1)] = (uint8_t)a; rew[t % (cap + 1)] = r; term[t % (cap + 1)] = (uint8_t)te;
        n_tr = (n_tr < cap) ? n_tr + 1 : cap;
        if (t > observe_steps) {                           /* _trainQNetwork */
            int64_t idx[32];
            uint8_t ab[32], tb[32];
            float rb[32], qn[64], qs[64], qsel[32], y[32], dq[64], loss;
            fbo_py_sample(&mt, n_tr, B, idx);
            int64_t oldest = (int64_t)t + 1 - n_tr;
            for (int b = 0; b < B; b++) {
                int64_t tt = oldest + idx[b];
                memcpy(sb + (size_t)b * 25600, states + (size_t)(tt % (cap + 1)) * 25600, 25600);
                memcpy(s2b + (size_t)b * 25600, states + (size_t)((tt + 1) % (cap + 1)) * 25600, 25600);
                ab[b] = act[tt % (cap + 1)]; rb[b] = rew[tt % (cap + 1)]; tb[b] = term[tt % (cap + 1)];
            }
            fbo_qnet_forward(P, c, s2b, B, qn, NULL);
            for (int b = 0; b < B; b++) qsel[b] = qn[2 * b] > qn[2 * b + 1] ? qn[2 * b] : qn[2 * b + 1];
            fbo_qnet_forward(P, c, sb, B, qs, acts);
            fbo_dqn_loss(0, B, 2, qs, qsel, ab, rb, tb, NULL, 0.99, y, &loss, NULL, dq);
            fbo_qnet_backward(P, c, sb, B, acts, dq, G);
            fbo_adam_step(P, M, V, G, NP, 1e-6f, 0.9f, 0.999f, 1e-8f, &b1p, &b2p);
            grad_steps++;
        }
    }
    double t1 = now_s();
    out->seconds = t1 - t0;
    out->env_steps = observe_steps + train_steps;
    out->grad_steps = grad_steps;
    out->env_steps_per_s = out->env_steps / (t1 - t0);
    out->grad_steps_per_s = grad_steps ? grad_steps / (t1 - t_train0) : 0;
    free(P); free(G); free(M); free(V); free(states); free(act); free(term); free(rew);
    free(sb); free(s2b); free(acts);
    return 0;
}
