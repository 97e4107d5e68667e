/*
 * oracle/fbo_replay.c -- TEST INFRASTRUCTURE (see fbo.h).
 *
 * CPU restatement of the prioritized replay of the reference:
 *   SumTree  BrainPrioritizedReplyDQN.py:32-104
 *   Memory   BrainPrioritizedReplyDQN.py:107-151
 * fp64 array heap, history-dependent running sums -- kept operation for
 * operation so that tree bytes and sampled indices are bit-exact with the
 * reference's own classes (tests/golden/per_sumtree.npz).
 * (The uniform replay needs no restatement beyond fbo_py_sample in fbo_rng.c:
 * the sampled value IS the deque index, BrainDQN.py:197.)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fbo.h"

fbo_per *fbo_per_create(int64_t capacity) {
    fbo_per *p = (fbo_per *)calloc(1, sizeof(fbo_per));
    p->capacity = capacity;
    p->tree = (double *)calloc((size_t)(2 * capacity - 1), sizeof(double));   /* :41 */
    p->beta = 0.4;                                                            /* :114 */
    return p;
}

void fbo_per_destroy(fbo_per *p) { if (p) { free(p->tree); free(p); } }

void fbo_per_update(fbo_per *p, int64_t tree_idx, double pr) {               /* :62-68 */
    double change = pr - p->tree[tree_idx];
    p->tree[tree_idx] = pr;
    while (tree_idx != 0) {
        tree_idx = (tree_idx - 1) / 2;
        p->tree[tree_idx] += change;
    }
}

void fbo_per_store(fbo_per *p) {                                             /* Memory.store :121-125 + add :50-60 */
    const double *leaves = p->tree + (p->capacity - 1);
    double max_p = leaves[0];
    for (int64_t i = 1; i < p->capacity; i++) if (leaves[i] > max_p) max_p = leaves[i];
    if (max_p == 0) max_p = 1.0;                                             /* abs_err_upper */
    fbo_per_update(p, p->data_pointer + p->capacity - 1, max_p);
    p->data_pointer += 1;
    if (p->data_pointer >= p->capacity) p->data_pointer = 0;
    if (p->size < p->capacity) p->size += 1;
}

int64_t fbo_per_get_leaf(const fbo_per *p, double v) {                       /* :73-100 */
    int64_t len = 2 * p->capacity - 1, parent = 0;
    for (;;) {
        int64_t cl = 2 * parent + 1, cr = cl + 1;
        if (cl >= len) return parent;
        if (v <= p->tree[cl]) parent = cl;
        else { v -= p->tree[cl]; parent = cr; }
    }
}

double fbo_per_min_prob(const fbo_per *p) {                                  /* :70-71 */
    const double *leaves = p->tree + (p->capacity - 1);
    double m = leaves[0];
    for (int64_t i = 1; i < p->size; i++) if (leaves[i] < m) m = leaves[i];
    return m / p->tree[0];
}

void fbo_per_sample(fbo_per *p, int n, fbo_mt *np_rng, const double *u_or_null,
                    int32_t *tree_idx, double *isw) {                        /* :127-144 */
    double total = p->tree[0];
    double pri_seg = total / n;
    double nb = p->beta + 0.001;
    p->beta = nb < 1.0 ? nb : 1.0;
    for (int i = 0; i < n; i++) {
        double a = pri_seg * i, b = pri_seg * (i + 1);
        double u = u_or_null ? u_or_null[i] : fbo_mt_random(np_rng);
        double v = a + (b - a) * u;                                          /* np.random.uniform(a, b) */
        int64_t idx = fbo_per_get_leaf(p, v);
        double prob = p->tree[idx] / p->tree[0];
        double min_prob = fbo_per_min_prob(p);
        isw[i] = pow(prob / min_prob, -p->beta);
        tree_idx[i] = (int32_t)idx;
    }
}

void fbo_per_batch_update(fbo_per *p, int n, const int32_t *tree_idx, float *abs_err) {   /* :146-151 */
    for (int i = 0; i < n; i++) {
        abs_err[i] += 0.01f;                                                 /* in place, fp32 */
        float c = abs_err[i] < 1.0f ? abs_err[i] : 1.0f;
        float ps = powf(c, 0.6f);
        fbo_per_update(p, tree_idx[i], (double)ps);
    }
}

/* Same with the priorities already raised to alpha: NumPy's float32 power is a
 * SIMD routine that is not correctly rounded and differs from libm's powf by
 * 1 ulp on some inputs, so bit-exact replays of the reference inject the
 * p values NumPy produced (tests/golden/per_sumtree.npz *_ps). */
void fbo_per_batch_update_p(fbo_per *p, int n, const int32_t *tree_idx, const float *ps) {
    for (int i = 0; i < n; i++) fbo_per_update(p, tree_idx[i], (double)ps[i]);
}
