/*
 * oracle/fbo.h -- CPU restatement ("oracle") of the Flappy-Bird DQN hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: it is
 * imported / linked only by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py, as the checker or the reported CPU baseline.
 * The product path (dqnflappybird_amd/ + libfbdqn.so) never calls into it and
 * fails loudly when the HIP library is missing.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout).  Pinning status (see DESIGN.md section "Oracle"):
 *   - game physics / spawn / score / collision : PINNED against trajectories
 *     produced by the reference's own game module (tests/golden/game_trajectories.npz)
 *   - SumTree / Memory                         : PINNED (bit-exact) against the
 *     reference's own classes (tests/golden/per_sumtree.npz)
 *   - random.sample / random() / randint       : PINNED against CPython itself
 *   - rendering (SDL blit) and cv2 preprocess  : PARITY UNPINNED (pygame / cv2
 *     are not installable here); restated from the published algorithms
 *   - Q-network / TF Adam                      : PARITY UNPINNED against
 *     TensorFlow 1.12 (not installable); cross-checked against torch-CPU in
 *     tests/test_oracle_qnet.py
 */
#ifndef FBO_H
#define FBO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- constants */
#define FBO_SCREENW 288          /* game/wrapped_flappy_bird.py:16 */
#define FBO_SCREENH 512          /* :17 */
#define FBO_PIPE_W 52            /* :48 */
#define FBO_PIPE_H 320           /* :49 */
#define FBO_PLAYER_W 34          /* :46 */
#define FBO_PLAYER_H 24          /* :47 */
#define FBO_BASE_W 336
#define FBO_BASE_H 112
#define FBO_PIPEGAP 100          /* :43 */
#define FBO_PLAYERX 57           /* int(288*0.2), :61 */
#define FBO_OBS 80               /* FlappyBirdDQN.py:32 */

/* ---------------------------------------------------------------- RNG */
typedef struct { uint32_t mt[624]; int idx; } fbo_mt;

void fbo_mt_init_genrand(fbo_mt *s, uint32_t seed);                 /* numpy legacy np.random.seed(int) */
void fbo_mt_init_by_array(fbo_mt *s, const uint32_t *key, int n);
void fbo_mt_seed_python(fbo_mt *s, uint64_t seed);                  /* random.seed(int >= 0) */
uint32_t fbo_mt_u32(fbo_mt *s);
double fbo_mt_random(fbo_mt *s);                                    /* random.random() == np random_sample */
uint32_t fbo_py_getrandbits(fbo_mt *s, int k);                      /* k <= 32 */
uint32_t fbo_py_randbelow(fbo_mt *s, uint32_t n);
int fbo_py_sample(fbo_mt *s, int64_t n, int k, int64_t *out);       /* random.sample(range(n), k) */
double fbo_np_uniform(fbo_mt *s, double lo, double hi);             /* np.random.uniform */

void fbo_philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                    uint32_t out[4]);

/* ---------------------------------------------------------------- assets */
int fbo_assets_load(const uint8_t *blob, size_t n);                 /* 0 ok */
/* hit masks in the reference's mask[x][y] convention, for the fixture check */
void fbo_hitmask_pipe(int upper, uint8_t *out /*[52][320]*/);
void fbo_hitmask_player(int pose, uint8_t *out /*[34][24]*/);

/* ---------------------------------------------------------------- env */
typedef struct {
    double playery;              /* float only in the crash frame (380.48) */
    int32_t vely, player_index, loop_iter, basex, score, cyc_pos, n_pipes;
    int32_t pipe_x[3], pipe_gap[3];   /* gap index 0..7, gapY = 100 + 10*idx */
    /* pipe-gap draw source: tape (if tape != NULL) else Philox stream 0 */
    const int8_t *tape; int64_t tape_len, tape_pos;
    uint32_t seed_lo, seed_hi, env_id, rng_ctr;
} fbo_env;

void fbo_env_init(fbo_env *e, uint32_t seed_lo, uint32_t seed_hi, uint32_t env_id,
                  const int8_t *tape, int64_t tape_len, int cyc_pos);
void fbo_env_reset(fbo_env *e);
/* returns 0, or -1 for an invalid action vector (ValueError in the reference) */
int fbo_env_step(fbo_env *e, int action, float *reward, int *terminal, int *score_return);
void fbo_env_render_full(const fbo_env *e, uint8_t *rgb /*[288][512][3] = array3d*/);
void fbo_preprocess(const uint8_t *rgb /*[288][512][3]*/, uint8_t *out /*[80][80]*/);
void fbo_env_frame80(const fbo_env *e, uint8_t *out /*[80][80]*/);
/* flat int32[16] snapshot used by the tests: y, vely, idx, loop, basex, score, npipes, x[3], uy[3], ly[3] */
void fbo_env_snapshot(const fbo_env *e, int32_t out[16]);

/* ---------------------------------------------------------------- replay */
typedef struct {
    int64_t capacity; double *tree; int64_t size, data_pointer;
    double beta;
} fbo_per;

fbo_per *fbo_per_create(int64_t capacity);
void fbo_per_destroy(fbo_per *p);
void fbo_per_update(fbo_per *p, int64_t tree_idx, double pr);
void fbo_per_store(fbo_per *p);                                     /* Memory.store */
int64_t fbo_per_get_leaf(const fbo_per *p, double v);
double fbo_per_min_prob(const fbo_per *p);
/* Memory.sample: uniforms in [0,1) injected (u[i]) or drawn from the numpy legacy stream */
void fbo_per_sample(fbo_per *p, int n, fbo_mt *np_rng, const double *u_or_null,
                    int32_t *tree_idx, double *isw);
void fbo_per_batch_update(fbo_per *p, int n, const int32_t *tree_idx, float *abs_err /* mutated */);
void fbo_per_batch_update_p(fbo_per *p, int n, const int32_t *tree_idx, const float *ps);

/* ---------------------------------------------------------------- Q network */
typedef struct { int fc, actions, dueling; } fbo_qcfg;
size_t fbo_qnet_nparams(fbo_qcfg c);
/* acts: NULL or workspace of fbo_qnet_act_floats(c)*B floats kept for backward */
size_t fbo_qnet_act_floats(fbo_qcfg c);
void fbo_qnet_forward(const float *params, fbo_qcfg c, const uint8_t *states, int B, float *q, float *acts);
float fbo_qnet_last_margin(void);     /* min |ReLU input| / pool win margin of the last forward (same thread) */
float fbo_qnet_last_margin_nonzero(void);   /* the same, exact ties / zeros (pool windows over identical pixels) left out */
void fbo_qnet_backward(const float *params, fbo_qcfg c, const uint8_t *states, int B,
                       const float *acts, const float *dq, float *grads);
void fbo_adam_step(float *p, float *m, float *v, const float *g, size_t n, float lr, float b1,
                   float b2, float eps, float *b1pow, float *b2pow);
/* y, loss, dq for the four loss variants; kind: 0 sum (BrainDQN), 1 mean (Nature), 2 IS-weighted mean (PER) */
void fbo_dqn_loss(int kind, int B, int A, const float *q, const float *q_next_sel, const uint8_t *action,
                  const float *reward, const uint8_t *terminal, const float *isw, double gamma,
                  float *y, float *loss, float *abs_err, float *dq);
/* policy-gradient loss (BrainPolicyGradient.py:96-100): mean over n_total of softmax cross-entropy x weight; this chunk's share */
void fbo_pg_loss(int B, int A, const float *q, const uint8_t *action, const float *w, double n_total, float *loss, float *dq);
void fbo_trunc_normal_init(float *params, fbo_qcfg c, uint32_t seed_lo, uint32_t seed_hi);

/* ---------------------------------------------------------------- single-env loop (CPU baseline) */
typedef struct {
    double env_steps_per_s, grad_steps_per_s, seconds;
    int64_t env_steps, grad_steps;
} fbo_loop_result;
int fbo_reference_loop(int observe_steps, int train_steps, int replay_cap, uint32_t seed,
                       fbo_loop_result *out);

#ifdef __cplusplus
}
#endif
#endif
