/*
 * oracle/fbo_rng.c -- TEST INFRASTRUCTURE (see fbo.h).
 *
 * The reference draws everything from two third-party generators that are not
 * part of its source tree:
 *   - CPython's `random` module (MT19937): epsilon-greedy BrainDQN.py:103-104,
 *     pipe gaps game/wrapped_flappy_bird.py:212, uniform replay BrainDQN.py:197
 *   - NumPy's legacy global RandomState (the same MT19937):
 *     BrainPrioritizedReplyDQN.py:136
 * Both are restated here from their published algorithms (Matsumoto & Nishimura
 * mt19937ar.c; CPython Lib/random.py + Modules/_randommodule.c; numpy
 * random/mtrand legacy seeding) and pinned against CPython / NumPy themselves
 * in tests/test_oracle_rng.py via tests/golden/cpython_random.npz.
 *
 * Philox4x32-10 (Salmon et al., SC'11) is this framework's own per-env stream;
 * there is no reference counterpart (the reference has one env and one stream).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "fbo.h"

#define MT_N 624
#define MT_M 397

void fbo_mt_init_genrand(fbo_mt *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < MT_N; i++)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = MT_N;
}

void fbo_mt_init_by_array(fbo_mt *s, const uint32_t *key, int key_length) {
    fbo_mt_init_genrand(s, 19650218u);
    uint32_t *mt = s->mt;
    int i = 1, j = 0;
    int k = MT_N > key_length ? MT_N : key_length;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
        if (j >= key_length) j = 0;
    }
    for (k = MT_N - 1; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= MT_N) { mt[0] = mt[MT_N - 1]; i = 1; }
    }
    mt[0] = 0x80000000u;
    s->idx = MT_N;
}

/* random.seed(n) for a non-negative int: key = 32-bit little-endian digits of n */
void fbo_mt_seed_python(fbo_mt *s, uint64_t seed) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    fbo_mt_init_by_array(s, key, key[1] ? 2 : 1);
}

uint32_t fbo_mt_u32(fbo_mt *s) {
    uint32_t *mt = s->mt, y;
    if (s->idx >= MT_N) {
        int kk;
        for (kk = 0; kk < MT_N - MT_M; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; kk < MT_N - 1; kk++) {
            y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
            mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
        mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        s->idx = 0;
    }
    y = mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* _randommodule.c random_random == numpy legacy rk_double */
double fbo_mt_random(fbo_mt *s) {
    uint32_t a = fbo_mt_u32(s) >> 5, b = fbo_mt_u32(s) >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

uint32_t fbo_py_getrandbits(fbo_mt *s, int k) { return fbo_mt_u32(s) >> (32 - k); }

static int bit_length(uint32_t n) { int k = 0; while (n) { k++; n >>= 1; } return k; }

/* Lib/random.py _randbelow_with_getrandbits */
uint32_t fbo_py_randbelow(fbo_mt *s, uint32_t n) {
    int k = bit_length(n);
    uint32_t r = fbo_py_getrandbits(s, k);
    while (r >= n) r = fbo_py_getrandbits(s, k);
    return r;
}

/* Lib/random.py Random.sample on a range(n) population: the deque index the
 * reference's random.sample(self.replayMemory, BATCH_SIZE) reads (BrainDQN.py:197) */
int fbo_py_sample(fbo_mt *s, int64_t n, int k, int64_t *out) {
    if (k < 0 || k > n) return -1;
    int64_t setsize = 21;
    if (k > 5) setsize += (int64_t)pow(4.0, ceil(log((double)k * 3.0) / log(4.0)));
    if (n <= setsize) {
        int64_t *pool = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
        for (int64_t i = 0; i < n; i++) pool[i] = i;
        for (int i = 0; i < k; i++) {
            uint32_t j = fbo_py_randbelow(s, (uint32_t)(n - i));
            out[i] = pool[j];
            pool[j] = pool[n - i - 1];
        }
        free(pool);
    } else {
        for (int i = 0; i < k; i++) {
            for (;;) {
                int64_t j = fbo_py_randbelow(s, (uint32_t)n);
                int dup = 0;
                for (int q = 0; q < i; q++) if (out[q] == j) { dup = 1; break; }
                if (!dup) { out[i] = j; break; }
            }
        }
    }
    return 0;
}

double fbo_np_uniform(fbo_mt *s, double lo, double hi) { return lo + (hi - lo) * fbo_mt_random(s); }

void fbo_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
