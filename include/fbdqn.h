/*
 * include/fbdqn.h -- C ABI of libfbdqn.so, the MI355X (gfx950) Flappy-Bird DQN hot path.
 *
 * This is the drop-in boundary.  The reference (angela000/DQNFlappyBird) is pure
 * Python and has no FFI layer of its own; its boundary is the duck-typed surface
 * FlappyBirdDQN.py uses (GameState.frame_step, Brain.getAction / setPerception).
 * Each entry point below names the reference function it stands behind
 * (paths relative to the reference checkout); the modules under dqnflappybird_amd/ bind them
 * with ctypes and re-creates the reference's class surface on top.
 *
 * Conventions
 *   - every pointer marked [dev] is device memory owned by the CALLER (e.g.
 *     torch.Tensor.data_ptr()); the library never frees caller memory;
 *   - [host] pointers are ordinary host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every
 *     call is asynchronous on it unless documented otherwise; nothing here
 *     allocates or synchronises inside the step functions, so a caller may
 *     capture them into a hipGraph;
 *   - return value: 0 = FB_OK, negative = error; fb_last_error() gives the text
 *     (thread-local).  No C++ exception crosses the boundary;
 *   - one handle per GPU; handles are not thread-safe; distinct handles are
 *     independent.
 *
 * Environment variables libfbdqn.so reads (each ONCE, at the first call that consults it).  Every default is the product
 * path; the other value selects a form the tests pin to the default bit for bit (an A/B or tuning switch, never a result):
 *   FB_ACT_FUSED=0            acting forward as two launches (conv1, then conv2 + conv3) instead of the fused trunk
 *   FB_VEC_HEAD_RIDER=0       fb_vec_step: head_kernel as its own launch instead of riding in the env launch
 *   FB_VEC_SAMPLE_RIDER=0     fb_vec_step: random.sample as its own launch instead of riding in the env launch
 *   FB_VEC_SPLIT=0            fb_vec_step keeps acting + env on the caller's stream in front of the train step instead of beside it on a second stream
 *   FB_SPLIT_LAST_ROUND=0     split schedule at >= 4096 envs: the train chain starts with the draw instead of with the acting trunk's last round of workgroups
 *   FB_ACT_SPW=4              four states per workgroup of the fused acting trunk outside the split schedule (default 5)
 *   FB_VEC_RING=0             fb_vec_step trains through fb_replay_gather + fb_qnet_train_step (u8 minibatch) instead of from the ring
 *   FB_TRAIN_STEPS_GATHER=1   the same for fb_train_steps
 *   FB_BW_MERGED=0            small-batch conv backward as two launches (conv_bx, conv_dw21) instead of conv_bw_kernel
 *   FB_SPAN_SPLIT=p, FB_SPAN_BLOCKS=n   (two-launch form only) share / workgroup cap of W_fc1's Adam span in the first launch
 *   FB_PER_STORE_AHEAD=0      Memory.store's tree update in line instead of on the memory's side stream (and with it the run-ahead sample)
 *   FB_PER_SAMPLE_AHEAD=0     Memory.sample in line
 *   FB_PER_UPDATE_AHEAD=0|1   Memory.batch_update in line / on the side stream whatever the env count (default: ahead from 4096 envs on)
 *   FB_ENV_GRID_CAP=n, FB_ENV_GRID=n    env workgroups before they stride over envs
 *   FB_ABORT_LOG=path         file the SIGABRT hook appends the native back-trace to (fb_debug_abort_backtrace)
 *   FB_SIDE_PROBE_DEBUG=1     print what the side-stream checks measured (fb_streams_concurrent, csrc/fb_common.hip) to stderr
 * The Python side reads FB_LIB (another build of this library), FB_DP_NATIVE / FB_DP_OVERLAP (which data-parallel path, dist.py).
 * Modes that DO change results (FB_PER_FAST, the train dtype, pipelined acting) are API calls below, never variables.
 */
#ifndef FBDQN_H
#define FBDQN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FB_OK 0
#define FB_ERR_INVALID (-1)      /* bad argument / shape */
#define FB_ERR_HIP (-2)          /* a HIP runtime call failed */
#define FB_ERR_NOMEM (-3)
#define FB_ERR_STATE (-4)        /* call order (e.g. sample before enough pushes) */

const char *fb_last_error(void);
int fb_version(void);
/* number of visible HIP devices, or a negative error; used by the shim to fail loudly */
int fb_device_count(void);
/* Diagnostics: install a SIGABRT handler that writes the aborting thread's native call stack to stderr -- and to the file the
 * environment variable FB_ABORT_LOG names, if set: a test runner may have captured descriptor 2 -- before the previous handler
 * runs (the GPU runtimes abort() on fatal errors, not always with a message).  Idempotent. */
int fb_debug_abort_backtrace(void);

/* ------------------------------------------------------------------ environment
 * N independent games stepped per launch, render + 80x80 preprocess fused in.
 *   GameState.__init__    game/wrapped_flappy_bird.py:59-85
 *   GameState.frame_step  game/wrapped_flappy_bird.py:87-183
 *   getRandomPipe         game/wrapped_flappy_bird.py:208-221
 *   checkCrash/pixelCollision  :244-300
 *   flappy_bird_utils.load/getHitmask  game/flappy_bird_utils.py:16-124 (-> sprite blob)
 *   preprocess            FlappyBirdDQN.py:31-34
 */
typedef struct fb_env *fb_env_t;

#define FB_ENV_STATE_INTS 16     /* layout of one env in get/set_state, see below */

/* sprite_blob: the packed sprites (tools/make_assets.py layout, 57 756 bytes) [host]. */
int fb_env_create(int n_envs, uint64_t seed, uint32_t flags, const void *sprite_blob, size_t blob_bytes,
                  fb_env_t *out);
int fb_env_destroy(fb_env_t h);
/* GameState.__init__ for every env (draws two pipe gaps each). */
int fb_env_reset(fb_env_t h, void *stream);
/* frame_step for every env.
 *   actions   [dev] u8[N]      0 = do nothing, 1 = flap (index of the one-hot's 1); any other
 *                              value is the reference's ValueError: the env is left untouched
 *                              and counted in fb_env_error_count()
 *   frames    [dev] u8[N,80,80] or NULL: preprocess(image_data) = {0,255}, [x_small][y_small]
 *   frame_bits[dev] u64[N,100] or NULL: the same frame, 1 bit per pixel (bit i of the row-major
 *                              pixel index i lives in word i/64, bit i%64)
 *   reward    [dev] f32[N]     0.1 / 3 / -3
 *   terminal  [dev] u8[N]
 *   score     [dev] i32[N]     score_return (captured before the reset on a crash)
 */
int fb_env_step(fb_env_t h, const uint8_t *actions, uint8_t *frames, uint64_t *frame_bits, float *reward,
                uint8_t *terminal, int32_t *score, void *stream);
/* Observation of the current state without stepping (used for the very first frame only by
 * tests; the reference obtains it with a do-nothing frame_step, FlappyBirdDQN.py:65-66). */
int fb_env_observe(fb_env_t h, uint8_t *frames, uint64_t *frame_bits, void *stream);
/* Parity / checkpoint access; synchronous.  i32[N][16] [host]:
 *   0 playery 1 velY 2 playerIndex 3 loopIter 4 basex 5 score 6 nPipes
 *   7..9 pipe x 10..12 pipe gap index (0..7) 13 PLAYER_INDEX_GEN phase 14 rng counter 15 tape cursor */
int fb_env_get_state(fb_env_t h, int32_t *state_host);
int fb_env_set_state(fb_env_t h, const int32_t *state_host);
/* Replace the Philox pipe-gap stream by an explicit tape of random.randint(0,7) results
 * (i8[N][tape_len], [host], copied); tape_len = 0 switches back to Philox. Synchronous. */
int fb_env_set_gap_tape(fb_env_t h, const int8_t *tape_host, int tape_len);
/* pygame.surfarray.array3d of one env: u8[288,512,3] [dev] (debug / parity). */
int fb_env_render_full(fb_env_t h, int env_id, uint8_t *rgb, void *stream);
/* Register (or clear with NULL) a caller-owned buffer u8[N][FB_NIB_STRIDE] [dev] that every following fb_env_observe /
 * fb_env_step keeps equal to the agent's 4-frame stack (BrainDQN.py:68,238-239) in "nibble" form: one byte = two
 * horizontally adjacent pixels, bit 4*px + f = frame f of the stack (f = 3 newest).  The image is stored with conv1's
 * SAME padding around it, so that the acting conv1 reads every tap at a fixed offset from one base address with no
 * bounds check: FB_NIB_ROWS rows of FB_NIB_PITCH bytes = [4 zero bytes][40 bytes: pixels 0..79 of the row]; image row r
 * is buffer row r + 2 (two zero rows above, two below; the right-hand padding of a row is the zero prefix of the next),
 * i.e. pixels (r, 2q), (r, 2q+1) live in byte (r + 2) * FB_NIB_PITCH + 4 + q.  observe zeroes the padding and fills all
 * four frames with the observation (setInitState); a step shifts and appends.  fb_qnet_act_nib consumes it, which
 * removes the currentState expansion from the acting path.  Synchronous. */
#define FB_NIB_PITCH 44
#define FB_NIB_ROWS 84
#define FB_NIB_STRIDE 3712      /* 84 * 44 = 3696, + 16 so that the last row's right-hand taps stay inside */
int fb_env_set_nib_buffer(fb_env_t h, uint8_t *nib_states);
/* Register (or clear with NULL) a caller-owned u64[4] [dev] that every following fb_env_step updates with atomics:
 * [0] episodes ended (the reference's gameTimes, BrainDQN.py:92), [1] sum and [2] maximum of their scores
 * (score_every_episode, :94), [3] pipes passed (reward 3 events).  The vectorised loop reads it back whenever it
 * logs, instead of syncing per step.  The caller zeroes it.  Synchronous. */
int fb_env_set_stats_buffer(fb_env_t h, uint64_t *stats);
/* number of invalid actions seen so far (synchronous). */
int fb_env_error_count(fb_env_t h, int64_t *count_host);
/* preprocess() of FlappyBirdDQN.py:31-34 (cv2.resize -> BGR2GRAY -> threshold) for frames the caller
 * holds as array3d images: rgb u8[n,288,512,3] [dev] -> out u8[n,80,80] [dev].  The fused env step
 * never needs it; it exists so that `preprocess(observ)` stays a drop-in. */
int fb_preprocess_rgb(fb_env_t h, const uint8_t *rgb, int n_frames, uint8_t *out, void *stream);

/* ------------------------------------------------------------------ replay memory
 * HBM ring of single frames (1 bit / pixel: preprocess only emits 0 or 255) + per-transition
 * action / reward / terminal; a transition is the 5-frame window (s = t-3..t, s' = t-2..t+1).
 *   deque store / popleft     BrainDQN.py:36,69-72      (REPLAY_MEMORY = 50000, :26)
 *   frame stack               BrainDQN.py:68,238-239    (newest last, never reset)
 *   random.sample             BrainDQN.py:197
 *   minibatch assembly        BrainDQN.py:198-201
 *   SumTree / Memory          BrainPrioritizedReplyDQN.py:32-151
 */
typedef struct fb_replay *fb_replay_t;

#define FB_REPLAY_UNIFORM 0
#define FB_REPLAY_PER 1

#define FB_RNG_CPYTHON 0         /* MT19937 + Lib/random.py: bit-exact random.sample(range(n), B) */
#define FB_RNG_PHILOX 1          /* counter based, with replacement, fully parallel */
#define FB_RNG_NUMPY 2           /* MT19937 legacy np.random.seed(int): Memory.sample's uniform() */

int fb_replay_create(int64_t capacity, int n_envs, int kind, fb_replay_t *out);
int fb_replay_destroy(fb_replay_t h);
int fb_replay_seed(fb_replay_t h, int rng_kind, uint64_t seed);          /* synchronous */
/* setInitState: the first observation becomes all four frames of every env's stack. */
int fb_replay_reset(fb_replay_t h, const uint8_t *frames /*[dev] u8[N,80,80] or NULL*/,
                    const uint64_t *frame_bits /*[dev] u64[N,100] or NULL*/, void *stream);
/* setPerception's store: one transition per env (env order = deque order within a step).
 * Exactly one of frames / frame_bits is given (the NEXT observation).  The handle counts pushes on
 * the host (the kernels receive the step index by value), so a push must not be replayed from a captured
 * hipGraph, and a captured fb_replay_gather / fb_replay_current_state addresses the memory as it was filled at
 * capture time (fine for replaying train steps on a memory that is not being pushed to; sample / train have no
 * such restriction). */
int fb_replay_push(fb_replay_t h, const uint8_t *frames, const uint64_t *frame_bits, const uint8_t *actions,
                   const float *rewards, const uint8_t *terminals, void *stream);
/* fb_replay_push followed by fb_replay_sample(batch) of a uniform memory, in one launch: identical results
 * (the sample depends on the memory's size after the push, not on the pushed data; same RNG consumption),
 * but the single-wave sampler runs beside the copy instead of after it.  idx i64[batch] [dev].  For a
 * prioritized memory or a non-CPython RNG it simply performs the two calls in a row (isw is not returned:
 * use the separate calls for PER). */
int fb_replay_push_sample(fb_replay_t h, const uint8_t *frames, const uint64_t *frame_bits, const uint8_t *actions,
                          const float *rewards, const uint8_t *terminals, int batch, int64_t *idx, void *stream);
/* currentState of every env: u8[N,80,80,4] [dev] (newest frame last). */
int fb_replay_current_state(fb_replay_t h, uint8_t *states, void *stream);
/* Uniform: idx = deque positions (0 = oldest) exactly as random.sample(range(len), B).
 * PER: idx = SumTree tree indices (b_idx of Memory.sample), isw = ISWeights[:,0] (f64),
 *      uniforms = B doubles in [0,1) [dev] replacing np.random.uniform's stream, or NULL. */
int fb_replay_sample(fb_replay_t h, int batch, const double *uniforms, int64_t *idx, double *isw, void *stream);
/* s, s2: u8[B,80,80,4]; a: u8[B]; r: f32[B]; t: u8[B]  (all [dev]) */
int fb_replay_gather(fb_replay_t h, int batch, const int64_t *idx, uint8_t *s, uint8_t *s2, uint8_t *a,
                     float *r, uint8_t *t, void *stream);
/* Measurement aid (bench.py roofline): fb_replay_gather launched `reps` times back to back on `stream`. */
int fb_replay_profile_gather(fb_replay_t h, int batch, const int64_t *idx, uint8_t *s, uint8_t *s2, uint8_t *a, float *r,
                             uint8_t *t, int reps, void *stream);
/* Memory.batch_update(tree_idx, abs_errors): abs_err f32[B] [dev] is updated in place (+= 0.01)
 * like the reference does; priorities_or_null f32[B] [dev] injects the p values instead of
 * computing (min(|e|+0.01, 1))^0.6 on the device. */
int fb_replay_update_priorities(fb_replay_t h, int batch, const int64_t *idx, float *abs_err,
                                const float *priorities_or_null, void *stream);
/* How the SumTree is maintained (prioritized memories only):
 *   FB_PER_EXACT (default) the reference's running sums in the reference's update order: tree bytes and sampled
 *                indices bit-identical to BrainPrioritizedReplyDQN.SumTree (the parity mode); Memory.store of
 *                N envs costs N ordered tree walks
 *   FB_PER_FAST  every touched node recomputed as left + right, level by level: order independent, same values
 *                up to fp64 rounding of the sums, ~20x faster stores for thousands of envs.  Switch only while the
 *                memory is empty or between steps; the tree stays valid in both modes. */
#define FB_PER_EXACT 0
#define FB_PER_FAST 1
int fb_replay_set_per_mode(fb_replay_t h, int mode);
/* host-side queries (synchronous): len(replayMemory); PER: tree copy f64[2*cap-1] [host] */
int fb_replay_size(fb_replay_t h, int64_t *size_host);
int fb_replay_per_tree(fb_replay_t h, double *tree_host, int64_t *data_pointer, int64_t *size, double *beta);
/* Checkpoint of the memory -- what the reference forgets (BrainDQN.py:176-192,227-233 save the network and three scalars; a
 * resumed run observes for OBSERVE steps again).  One opaque [host] blob of fb_replay_state_bytes() bytes holds the frame ring, the
 * action / reward / terminal rows, the counters, the sampler's generator state and the SumTree heaps: after fb_replay_load_state
 * into a memory created with the same capacity / env count / kind, sample / gather / push continue bit for bit.  Synchronous. */
int fb_replay_state_bytes(fb_replay_t h, size_t *bytes_host);
int fb_replay_save_state(fb_replay_t h, void *blob_host, size_t bytes);
int fb_replay_load_state(fb_replay_t h, const void *blob_host, size_t bytes);

/* ------------------------------------------------------------------ Q network
 *   network        BrainDQN.py:119-155 (conv 8x8/4 -> pool -> conv 4x4/2 -> conv 3x3/1 -> fc -> A)
 *   dueling head   BrainDuelingDQN.py:78-86
 *   getAction      BrainDQN.py:99-116
 *   _trainQNetwork BrainDQN.py:195-223, BrainDQNNature.py:149-182, BrainDoubleDQN.py:37-68,
 *                  BrainPrioritizedReplyDQN.py:277-315
 *   Adam           BrainDQN.py:163 (tf.train.AdamOptimizer(1e-6))
 *   target sync    BrainDQNNature.py:107-111,151-152
 * Flat fp32 parameter order = the reference's variable creation order:
 *   W_conv1[8,8,4,32] b[32] W_conv2[4,4,32,64] b[64] W_conv3[3,3,64,64] b[64] W_fc1[1600,FC] b[FC]
 *   then  W_fc2[FC,A] b[A]            (plain)
 *   or    W_v[FC,1] b_v[1] W_a[FC,A] b_a[A]   (dueling)
 */
typedef struct fb_qnet *fb_qnet_t;

#define FB_ARCH_PLAIN 0
#define FB_ARCH_DUELING 1
#define FB_NET_ONLINE 0
#define FB_NET_TARGET 1
#define FB_ALGO_DQN 0            /* BrainDQN: target from the same net, loss = sum */
#define FB_ALGO_NATURE 1         /* BrainDQNNature: frozen target net, loss = mean */
#define FB_ALGO_DOUBLE 2         /* BrainDoubleDQN.trainQNetwork: argmax online, value target, mean */
#define FB_ALGO_PER 3            /* BrainPrioritizedReplyDQN: target net, mean(ISW * sq), abs_errors */
/* Policy gradient (BrainPolicyGradient.py:96-100; the actor of BrainActorCritic.py:96-100): the net's outputs are LOGITS,
 * loss = mean over N samples of softmax_cross_entropy(logits, action) x weight.  In fb_qnet_train_step: r = the weights (what the
 * reference feeds as tf_rewards / td_error), gamma = N as a double -- a batch larger than 128 (one whole episode) goes in chunks of
 * <= 128 that export their gradient (flat_grad), the caller adds them up and calls fb_qnet_apply_adam once; s2 / t are ignored (pass
 * s / zeros), abs_err / q_target are not meaningful; loss = this chunk's share of the mean.  Not available through fb_vec_step. */
#define FB_ALGO_PG 4

int fb_qnet_create(int arch, int fc_width, int n_actions, int max_batch, fb_qnet_t *out);
int fb_qnet_destroy(fb_qnet_t h);
int fb_qnet_num_params(fb_qnet_t h, int64_t *n_host);
/* tf.truncated_normal(stddev=0.01) weights, 0.01 biases, for `which` net. */
int fb_qnet_init_params(fb_qnet_t h, int which, uint64_t seed, void *stream);
int fb_qnet_load_params(fb_qnet_t h, int which, const float *flat /*[dev]*/, void *stream);
int fb_qnet_store_params(fb_qnet_t h, int which, float *flat /*[dev]*/, void *stream);
/* Adam slots m, v (f32[n] [dev]) and beta powers ([host] f32[2]); synchronous. */
int fb_qnet_get_adam_state(fb_qnet_t h, float *m, float *v, float *beta_pows_host);
int fb_qnet_set_adam_state(fb_qnet_t h, const float *m, const float *v, const float *beta_pows_host);
int fb_qnet_set_hparams(fb_qnet_t h, float lr, float beta1, float beta2, float eps);
/* Arithmetic of the forward-only path on >= 256 states (fb_qnet_forward / fb_qnet_act / fb_qnet_act_nib):
 *   FB_DTYPE_F32  (default) fp32-equivalent: every fp32 product as three fp16 MFMA products of two planes (x = h + l/4096).  The
 *                 pair carries x to 2^-24 relative for 2^-14 <= |x| < 65504 (fp16's normal range): weights and activations of this
 *                 network live there; gradient operands (1e-6 .. 1e-9 in the reference's regime) are brought there by an exact
 *                 power-of-two pre-scale taken from the operand block's maximum, folded back in the epilogue (csrc/fb_qnet.hip
 *                 pow2_scale; tests/test_gpu_qnet.py::test_train_step_gradients_in_the_reference_regime)
 *   FB_DTYPE_BF16 plain bf16 inference (config 3 of BASELINE.json: "bf16"): activations and weights rounded to
 *                 bf16, fp32 accumulation.  Training and batches < 256 always compute in fp32. */
#define FB_DTYPE_F32 0
#define FB_DTYPE_BF16 1
int fb_qnet_set_inference_dtype(fb_qnet_t h, int dtype);
/* Range guard of FB_DTYPE_F32.  ACTIVATIONS are split into their two fp16 planes unscaled, so the form is exact only for
 * |x| < FB_F16_RANGE = 32768 (h overflows from 65520 on, l already from 32768 on).  TensorFlow's fp32 (BrainDQN.py:119-155) has no such
 * limit; the reference network's activations are O(1 .. 100) and stay far inside, a net loaded from outside need not.  Every kernel that
 * splits an activation (acting trunk, training trunk, the large-batch weight-gradient kernels) therefore counts the waves that met
 * |x| >= FB_F16_RANGE in a device word instead of silently producing inf / NaN (or a finite wrong number behind the next relu):
 *   fb_qnet_overflow_count -> [host] the count since creation / the last reset (synchronous; 0 = every result so far is in range).
 * A non-zero count means: the Q-values / gradients of those launches are NOT to be trusted; switch the net to FB_DTYPE_BF16 (fp32's
 * exponent range) or rescale its weights.  VecBrain.run and the Brain* classes raise on it at their log cadence. */
int fb_qnet_overflow_count(fb_qnet_t h, int reset, int64_t *count_host);
/* Arithmetic of fb_qnet_train_step (BASELINE.json configs[2]: "bf16"):
 *   FB_DTYPE_F32  (default) fp32: the fc1 GEMMs of small batches on the fp32-input matrix instruction, everything else on two-plane fp16
 *                 (as above, gradient operands pre-scaled)
 *   FB_DTYPE_BF16 bf16 training: every GEMM operand (activations, weights, incoming gradients; conv1's u8 input is exact anyway)
 *                 is rounded to bf16, products accumulate in fp32, the master weights and both Adam slots stay fp32.  Gradients
 *                 then agree with fp32 ones to a few per cent per tensor (tests/test_gpu_configs.py states the bound). */
int fb_qnet_set_train_dtype(fb_qnet_t h, int dtype);
/* QValue.eval: states u8[B,80,80,4] -> q f32[B,A] */
int fb_qnet_forward(fb_qnet_t h, int which, const uint8_t *states, int batch, float *q, void *stream);
/* getAction for N envs: forward + epsilon-greedy (Philox stream 1, counter = step).
 *   epsilon f32 by value; actions u8[N] out; q f32[N,A] out or NULL */
int fb_qnet_act(fb_qnet_t h, const uint8_t *states, int n, float epsilon, uint64_t seed, uint64_t step,
                uint8_t *actions, float *q, void *stream);
/* One _trainQNetwork step on a gathered minibatch.
 *   isw f32[B] (PER) or NULL; loss f32[1]; abs_err f32[B] or NULL; q_target f32[B] or NULL (all [dev])
 *   flat_grad NULL : gradients are applied with Adam at once (single GPU)
 *   flat_grad [dev] f32[n_params]: gradients are only written there (data parallel: all-reduce
 *             them, then fb_qnet_apply_adam) */
/* fb_qnet_act on the env kernel's nibble states u8[n][FB_NIB_STRIDE] (fb_env_set_nib_buffer) */
int fb_qnet_act_nib(fb_qnet_t h, const uint8_t *nib_states, int n, float epsilon, uint64_t seed, uint64_t step,
                    uint8_t *actions, float *q, void *stream);
int fb_qnet_train_step(fb_qnet_t h, int algo, int batch, const uint8_t *s, const uint8_t *a, const float *r,
                       const uint8_t *s2, const uint8_t *t, const float *isw, double gamma, float *loss,
                       float *abs_err, float *q_target, float *flat_grad, void *stream);
int fb_qnet_apply_adam(fb_qnet_t h, const float *flat_grad, void *stream);
/* Data parallel, all-reduce off the critical path: a step that exports its gradient (flat_grad != NULL; fb_qnet_train_step,
 * fb_vec_step, fb_train_from_replay) records `event` (a hipEvent_t, or NULL to switch this off) on its stream right behind the
 * fc1 backward launch.  From that point flat_grad[fb_qnet_grad_split() ..) -- W_fc1, b_fc1 and the head, 91 % of the bytes -- is
 * final: reduce it on a side stream that waits for the event while the conv backward (3 more launches) still runs, reduce the
 * small front part [0, fb_qnet_grad_split()) on the step's stream afterwards, join, fb_qnet_apply_adam. */
int fb_qnet_set_grad_event(fb_qnet_t h, void *event);
int64_t fb_qnet_grad_split(fb_qnet_t h);
int fb_qnet_sync_target(fb_qnet_t h, void *stream);
/* Measurement aid (bench.py roofline): re-launch ONE kernel of the train-step plan `reps` times on
 * `stream` with the geometry the real step uses, on the workspace a preceding fb_qnet_train_step
 * of the same shape left behind.  Kernel ids count from 0; fb_qnet_kernel_name() returns "" past the
 * last one.  algo = -1 / -2 selects the acting forward (batch states as u8 / as nibble states).  The Adam kernel really updates the parameters: use a scratch network. */
int fb_qnet_profile_kernel(fb_qnet_t h, int kernel, int reps, int algo, int batch, const uint8_t *s, const uint8_t *a,
                           const float *r, const uint8_t *s2, const uint8_t *t, float *loss, void *stream);
const char *fb_qnet_kernel_name(int kernel);

/* ------------------------------------------------------------------ one whole step of the vectorised loop
 * FlappyBirdDQN.py:72-76 for N envs in ONE call: getAction (fb_qnet_act_nib) -> frame_step
 * (fb_env_step, packed frames) -> store + random.sample (fb_replay_push_sample) -> minibatch (fb_replay_gather)
 * -> _trainQNetwork (fb_qnet_train_step).  The results of exactly those calls in that order on `stream`.  It saves the
 * host's per-call overhead between launches (the GPU otherwise idles ~15 us per step waiting for the interpreter) and
 * four launches: the head of the acting forward (fc2 + epsilon-greedy action), random.sample and the Memory append
 * ride inside the env step launch (uniform memory, CPython generator, <= 2048 envs and 2 actions for the head; anything
 * else keeps its own launch), bit-identical to the separate calls.
 * With >= 256 envs the step skips the gather launch as well: the train step's first kernel reads the sampled
 * transitions' 1-bit frames in the ring itself (fb_train_from_replay below) -- same results, b->s / b->s2 stay untouched.
 * A prioritized memory (BrainPrioritizedReplyDQN.py:277-329) runs store -> Memory.sample -> train with the importance weights ->
 * Memory.batch_update in the same call (fb_replay_push, fb_replay_sample, fb_train_from_replay, fb_replay_update_priorities: no riders).
 * All pointers [dev], caller owned; nib is the buffer given to fb_env_set_nib_buffer.  train = 0 stops after the
 * store (the reference's OBSERVE phase).  flat_grad as in fb_qnet_train_step (data parallel: all-reduce it, then
 * fb_qnet_apply_adam).
 *
 * hipGraph capture.  State that decides what a launch does lives on the device (parameter / plane versions, Adam's step counter, the
 * sampler's generator, SumTree pointer / size, beta), so a captured call replays correctly -- with ONE exception: the replay memory's
 * push counter also has a host-side mirror (it is passed by value into the launches that address the frame ring), so a captured
 * launch that pushes or addresses the ring is only valid while the memory holds the number of pushes it held at capture time.
 *   capturable, replayable any number of times:  fb_qnet_forward / _act / _act_nib, fb_qnet_train_step, fb_qnet_apply_adam,
 *       fb_qnet_sync_target, fb_env_step, fb_replay_sample, fb_replay_update_priorities, and -- on a memory that is NOT pushed to
 *       between capture and the last replay -- fb_replay_gather, fb_train_from_replay and fb_train_steps (what bench.py's
 *       train-only leg does: fb_train_steps(10) in one graph)
 *   NOT capturable for replay:  fb_replay_push / _push_sample, fb_vec_step, fb_vec_step_dp (they advance the push counter: a replay
 *       would write the same ring slot again and sample a memory of the captured size); synchronous calls (get / set state, seeds,
 *       hyper-parameters, checkpoints) synchronise the device and must stay outside a capture. */
typedef struct {
    uint8_t *nib;                                   /* u8[N,FB_NIB_STRIDE] */
    uint8_t *actions;                               /* u8[N] out */
    uint64_t *frame_bits;                           /* u64[N,100] out */
    float *reward; uint8_t *terminal; int32_t *score;   /* [N] out */
    int64_t *idx;                                   /* i64[B] out */
    uint8_t *s, *s2, *a, *t; float *r;              /* gathered minibatch: u8[B,80,80,4] x2, u8[B], u8[B], f32[B] */
    float *loss;                                    /* f32[1] out */
    float *flat_grad;                               /* f32[n_params] or NULL */
    /* prioritized replay (algo = FB_ALGO_PER) only, else NULL: Memory.sample's importance weights as it returns them (f64[B]) and as the
     * float32 placeholder takes them (f32[B]), and the |TD errors| Memory.batch_update receives (f32[B]).
     * With the reference-order tree and 4096 envs or more Memory.batch_update of a step runs on the memory's own side stream, beside the NEXT step's acting
     * forward (its result is first needed by that step's Memory.store, which follows it there): idx and abs_err are read after
     * fb_vec_step has returned and must stay valid -- and unwritten by the caller -- until the next call on this memory.  Whichever
     * form Memory.batch_update takes (side stream or in line), abs_err holds |TD error| as the loss left it when fb_vec_step returns
     * (fb_replay_update_priorities, the stand-alone call, adds its 0.01 in place as the reference does,
     * BrainPrioritizedReplyDQN.py:147).  Any later call that touches the memory's tree joins that stream first. */
    double *isw; float *isw32; float *abs_err;
} fb_step_buffers;
/* The split schedule.  For a uniform memory with the CPython generator, 256 <= n_envs <= 8192, batch < 256, a 2-action net and a stream
 * that is not being captured, fb_vec_step keeps the train step on `stream` and puts the acting forward and the env step on a stream of the
 * net's own BESIDE it: in the reference's loop both read the weights the previous step's Adam left (FlappyBirdDQN.py:72-76), and the
 * minibatch depends on the env step only when it holds one of the n_envs transitions this very step appends -- the draw decides that on
 * the device and then waits for the env step itself (~3 % of the steps at 1024 envs / 1 M slots).  The two chains hand over through
 * device words that kernels store and single waves poll; every wait is bounded at 1 s and counted.  Results are those of the
 * one-stream order bit for bit.
 * On return everything the step produced is ordered on `stream`, so callers need not know -- with one exception: a step that exports
 * its gradient (flat_grad) is completed by fb_qnet_apply_adam / fb_dist_reduce_apply on the same stream, and only behind THAT call is
 * `stream` ordered behind the step's env launch (actions, rewards, terminals, scores, frame bits).
 * The first call with a given `stream` checks that the net's side stream really runs beside it (HIP may map both to one hardware queue
 * or pipe): it synchronises both streams a few times, ~1 ms, once; if no side stream passes, the step stays on one stream.
 *   fb_qnet_split_stats -> [host] steps issued that way / how many of their minibatches started beside the env step (synchronous);
 *                          FB_ERR_INVALID with the per-site counts if any wait between the chains gave up. */
int fb_qnet_split_stats(fb_qnet_t net, int64_t *steps_host, int64_t *clean_host);
/* Process-wide A/B switch of the above (both schedules give the same results): split = 0 keeps every later fb_vec_step on one stream,
 * 1 (the default; the environment variable FB_VEC_SPLIT=0 starts the process with 0) takes the split schedule where it applies. */
int fb_vec_step_set_schedule(int split);
int fb_vec_step(fb_env_t env, fb_replay_t replay, fb_qnet_t net, const fb_step_buffers *b, int n_envs, int algo, int batch,
                float epsilon, uint64_t seed, uint64_t step, int train, double gamma, void *stream);

/* fb_replay_gather + fb_qnet_train_step WITHOUT the gathered copies (BrainDQN.py:197-223 from the indices on): the minibatch is
 * described by its indices, the conv trunk reads the replay's 1-bit frames directly (5 x 800 B per transition instead of writing and
 * re-reading 51 KB of u8 expansion) and fills a, r, t (u8 / f32 / u8 [batch], [dev] out).  Bit-identical to the two separate calls.
 * batch <= 256.  isw f32[batch] (the prioritized step's importance weights, idx then are SumTree leaf indices) or NULL; abs_err f32[batch]
 * out (|TD error|, for fb_replay_update_priorities) or NULL; flat_grad as in fb_qnet_train_step. */
int fb_train_from_replay(fb_replay_t replay, fb_qnet_t net, int algo, int batch, const int64_t *idx, const float *isw, uint8_t *a, float *r,
                         uint8_t *t, double gamma, float *loss, float *abs_err, float *flat_grad, void *stream);
/* Measurement aid: kernel `kernel` (ids of fb_qnet_kernel_name) of that step's plan, `reps` times, like fb_qnet_profile_kernel. */
int fb_profile_ring_kernel(fb_replay_t replay, fb_qnet_t net, int kernel, int reps, int algo, int batch, const int64_t *idx, uint8_t *a,
                           float *r, uint8_t *t, float *loss, void *stream);

/* ------------------------------------------------------------------ data parallel: one process per GPU, RCCL over xGMI
 * The only exchange between ranks is the all-reduce of the flat gradient between the backward pass and Adam (envs and replay shards are
 * rank-local).  fb_vec_step_dp = fb_vec_step(flat_grad) + that all-reduce + fb_qnet_apply_adam in one call, with RCCL called directly
 * on the step's own stream (no detour through another library's stream).  Optionally (fb_dist_set_overlap) in two pieces: the W_fc1 /
 * head part of the gradient (91 % of the bytes, final behind the fc1 backward launch) on a side stream while the conv backward still
 * runs, the conv part on the step's stream, then a join and Adam -- the same sums as one all-reduce of the whole vector.  mean_loss != 0 divides by the world size afterwards (the mean losses of
 * BrainDQNNature.py:119 / BrainPrioritizedReplyDQN.py:251; BrainDQN's sum loss, BrainDQN.py:162, is a plain sum).
 * Set-up: rank 0 calls fb_dist_unique_id, hands the 128 bytes to every rank by whatever channel the launcher has (the Python side
 * broadcasts them, dqnflappybird_amd/dist.py), every rank calls fb_dist_create (collective: ncclCommInitRank) with its HIP device current.
 * librccl_path: the librccl.so the process already holds (a Python host's framework usually bundles one), or NULL for the default search. */
typedef struct fb_dist *fb_dist_t;
/* rank-local, non-collective: 0 when this process can load RCCL and resolve its symbols.  Every rank calls it (and rank 0
 * fb_dist_unique_id) BEFORE the ranks agree to take this path; once the id has been exchanged a failure of fb_dist_create on one
 * rank leaves its peers inside ncclCommInitRank, so from there on a failure must end the job, not fall back. */
int fb_dist_probe(const char *librccl_path);
int fb_dist_unique_id(const char *librccl_path, uint8_t *id128 /*[host] out*/);
fb_dist_t fb_dist_create(const char *librccl_path, int rank, int world, const uint8_t *id128 /*[host]*/);
void fb_dist_destroy(fb_dist_t d);
/* 0 (default): one all-reduce of the whole gradient on the step's stream; 1: the two-piece schedule described above.  Every cross-stream
 * dependency costs 5-8 us on this hardware, so the two-piece schedule only pays when the 3.3 MB all-reduce takes longer than ~20 us. */
int fb_dist_set_overlap(fb_dist_t d, int overlap);
int fb_vec_step_dp(fb_dist_t d, fb_env_t env, fb_replay_t replay, fb_qnet_t net, const fb_step_buffers *b, int n_envs, int algo,
                   int batch, float epsilon, uint64_t seed, uint64_t step, int train, double gamma, int mean_loss, void *stream);
/* The reduction + Adam alone, for a gradient some other call exported (fb_qnet_train_step, fb_train_from_replay) after
 * fb_qnet_set_grad_event(net, fb_dist_grad_event(d)). */
int fb_dist_reduce_apply(fb_dist_t d, fb_qnet_t net, float *flat_grad /*[dev]*/, int mean_loss, void *stream);
void *fb_dist_grad_event(fb_dist_t d);
/* The collective alone: sum all-reduce of buf[count] (f32, [dev], in place) over the communicator, on `stream`.  For measurement
 * (bench.py config.allreduce_us) and for callers that schedule their own step. */
int fb_dist_all_reduce(fb_dist_t d, float *buf, int64_t count, void *stream);

/* n_steps x (fb_replay_sample -> fb_replay_gather -> fb_qnet_train_step) on a uniform memory in ONE call, same results: only
 * the first draw and the first gather are launches of their own, the draw of step i + 1 rides in step i's conv3 backward
 * launch and its gather in step i's Adam launch (CPython generator; other generators keep their launches).  idx: i64[2 * batch] [dev], two buffers used alternately (step i: idx + (i & 1) * batch);
 * s, s2, a, r, t, loss as in fb_qnet_train_step. */
int fb_train_steps(fb_replay_t replay, fb_qnet_t net, int algo, int batch, int n_steps, int64_t *idx, uint8_t *s, uint8_t *s2,
                   uint8_t *a, float *r, uint8_t *t, float *loss, double gamma, void *stream);

#ifdef __cplusplus
}
#endif
#endif
