// Shared host/device helpers of libfbdqn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/fbdqn.h"

// ---------------------------------------------------------------- errors
extern thread_local char fb_err_buf[512];
int fb_set_error(int code, const char *fmt, ...);

#define FB_CHECK_HIP(expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fb_set_error(FB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                           \
    } while (0)

#define FB_REQUIRE(cond, ...)                                          \
    do {                                                               \
        if (!(cond)) return fb_set_error(FB_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define FB_LAUNCH_CHECK() FB_CHECK_HIP(hipGetLastError())

static inline hipStream_t fb_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Hand-offs between kernels of two streams through device words (fb_vec_step's split schedule).  A word only ever grows (the step
// number).  Stores and polls are relaxed agent-scope atomics (they go to the coherent level, past the XCD's own L2); a reader that goes
// on to READ what the other kernel wrote adds fb_flag_acquire() -- the L2s of the eight XCDs are not coherent with each other inside a
// launch -- and the writer of such data stores the word from a kernel BEHIND the one that wrote the data (its end-of-kernel release has
// written the data back).  Waits are bounded: after ~1 s a wave counts a timeout and goes on, so none can spin forever.
__device__ __forceinline__ unsigned long long fb_flag_load(const unsigned long long *flag) { return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fb_flag_store(unsigned long long *flag, unsigned long long v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fb_flag_acquire() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }
// A launch's GATE workgroup: one extra workgroup at the end of the grid whose first thread does nothing but wait for a word -- so the
// launch does not retire, and the next launch on its stream does not start, before the other stream's kernel has.  One wave spins; the
// kernels behind the gate never do (waves that spin must not hold what the kernels they wait for need: a launch whose every workgroup
// waited for the acting trunk to retire kept that trunk, which wants whole SIMDs, from ever being placed).
struct FbGate { const unsigned long long *flag; unsigned long long val; unsigned *timeouts; };      // flag == NULL: no gate workgroup in the grid
__device__ __forceinline__ void fb_flag_wait(const unsigned long long *flag, unsigned long long v, unsigned *timeouts);
__device__ __forceinline__ bool fb_gate_workgroup(const FbGate &g) {          // true: this workgroup was the gate (it has waited; return)
    if (!g.flag || blockIdx.x != gridDim.x - 1) return false;
    if (threadIdx.x == 0) fb_flag_wait(g.flag, g.val, g.timeouts);
    return true;
}
__device__ __forceinline__ void fb_flag_wait(const unsigned long long *flag, unsigned long long v, unsigned *timeouts) {
    if (fb_flag_load(flag) >= v) return;
    const long long t0 = wall_clock64();
    while (fb_flag_load(flag) < v) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 100000000LL) { atomicAdd(timeouts, 1u); break; }      // (100 MHz counter)
    }
}

// ---------------------------------------------------------------- Philox4x32-10
// This framework's own counter-based stream (the reference has a single env and
// a single shared MT19937): key = seed, counter = (entity id, draw counter, stream id, 0).
#define FB_STREAM_GAP 0u      // pipe gaps, random.randint(0,7)   game/wrapped_flappy_bird.py:212
#define FB_STREAM_EPS 1u      // epsilon-greedy                     BrainDQN.py:103-104
#define FB_STREAM_SAMPLE 2u   // uniform replay (FB_RNG_PHILOX)
#define FB_STREAM_PER 3u      // PER segment uniforms (FB_RNG_PHILOX)
#define FB_STREAM_INIT 4u     // truncated-normal weight init

struct fb_u4 { uint32_t x, y, z, w; };

__host__ __device__ static inline fb_u4 fb_philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                                  uint32_t c2, uint32_t c3) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return fb_u4{c0, c1, c2, c3};
}

// ---------------------------------------------------------------- MT19937 (device side)
// CPython `random` / numpy legacy RandomState share this generator; state lives in HBM
// (625 words: mt[624], idx) and is advanced by a single wave.
struct FbMT { uint32_t mt[624]; uint32_t idx; };

// what a draw needs: the generator, the population size n = len(memory) at the time of the call, an error flag
// gate (split schedule of fb_vec_step, or NULL): the draw is that step's first launch on the caller's stream (it stores c_entry = gate_val
// on arrival) and the gate of the train chain behind it: it does not retire before the env step its minibatch depends on has (the
// previous step's when none of the drawn positions is >= newest_from, else this step's own)
struct FbSplitFlags;
struct FbSampleCtx { FbMT *mt; int *error; long long n; FbSplitFlags *gate; unsigned long long gate_val; long long newest_from; int wait_last_round; };
// random.sample(range(n), k) -> out[k] as a rider of another module's launch (fb_sampler.h; k == 0: no rider)
struct FbSampleRider { FbSampleCtx ctx; int k; long long setsize; long long *out; };
// fc2 / dueling head + epsilon-greedy action of one state from the fc1 partial sums (fb_head.h): the acting path's last
// kernel, which fb_vec_step lets ride in the env step launch (every env workgroup computes its own action first).
struct NetOff { int bf1, wv, bv, wq, bq, n; };
struct HeadCore {
    const float *hf; int stot, nks; float *q; int FC, A, dueling; NetOff off;
    uint8_t *actions; float epsilon; uint32_t seed_lo, seed_hi, step_lo, step_hi;
};
// on_arrival (split schedule, or NULL): a word the launch that carries the rider stores when it arrives -- the launch in front of it (the
// acting forward's fc1 launch) has retired then
struct FbHeadRider { HeadCore c; const float *params; int on; unsigned long long *on_arrival; unsigned long long arrival_val; };
// fb_replay_gather as a rider of another module's launch (fb_gather.h; B == 0: no rider): what the kernel needs of the ring
struct FbGatherCtx {
    long long cap; int n_envs, t_f, kind;
    const unsigned long long *bits; const uint8_t *act; const float *rew; const uint8_t *term; int *error;
};
struct FbGatherRider { FbGatherCtx c; long long steps; int B; const long long *idx; uint8_t *s, *s2, *a; float *r; uint8_t *t; };
int fb_replay_gather_rider(fb_replay_t h, int batch, const int64_t *idx, uint8_t *s, uint8_t *s2, uint8_t *a, float *r, uint8_t *t,
                           FbGatherRider *rider);
// A sampled minibatch described by where it lives in the frame ring instead of by gathered copies: the train step's first kernel reads
// the 1-bit frames itself (no gather launch, no u8 expansion) and fills a / r / t (u8 / f32 / u8 [B], [dev]) for the loss.
struct FbRingSrc { FbGatherCtx c; long long steps; const long long *idx; uint8_t *a; float *r; uint8_t *t; };
int fb_replay_ring_src(fb_replay_t h, int batch, const int64_t *idx, uint8_t *a, float *r, uint8_t *t, FbRingSrc *out);
// fb_qnet_train_step on such a minibatch (isw / abs_err: the prioritized step's importance weights in, |TD errors| out; else NULL).  The split conv planes of both nets must be current: true
// after an acting forward of >= 256 states in the same stream order (fb_vec_step), which is the only caller.
// rider: random.sample for the NEXT step in the conv3 backward launch (fb_train_steps), or NULL.
// split (the split schedule's train chain, or NULL): the fc1 backward launch gets a gate workgroup waiting for trunk_done (W_fc1's Adam span
// rides in the launch behind it), the conv backward launch one waiting for fc1_done (the Adam launch follows), and the Adam launch's
// last thread waits for env_done
int fb_qnet_train_step_ring(fb_qnet_t h, int algo, int batch, const FbRingSrc *ring, const float *isw, double gamma, float *loss,
                            float *abs_err, float *flat_grad, void *stream, const FbSampleRider *rider = nullptr, const struct FbSplitCtx *split = nullptr);
// The split schedule of fb_vec_step (fb_common.hip): the train step of a vector step on a stream of its own BESIDE the acting forward and
// the env step -- both read the weights the previous step's Adam left; only the replay push connects them, and only when the minibatch
// holds one of the transitions this very step appends (the sampler decides that on the device and opens `gate` itself when it does not).
struct FbSplitFlags {                          // [dev], one word per 64 bytes; each holds the number of the last step that reached the point
    unsigned long long c_entry, p0[7];         // the caller's stream has reached this step's first launch (the draw): all it held before is done
    unsigned long long trunk_done, p2[7];      // the acting trunk has retired (stored by the fc1 launch behind it)
    unsigned long long fc1_done, p3[7];        // the acting forward's fc1 launch has retired
    unsigned long long env_done, p4[7];        // the env step (with the push and the head riding in it) has retired
    unsigned long long last_round, p8[7];      // (acting trunks of more than one round of workgroups) the first workgroup of the LAST round has been placed
    unsigned clean_count;                      // minibatches that started beside their env step
    unsigned timeouts[7];                      // waits that gave up (must stay 0), per site: 0 the draw (env_done) 1 the side stream's entry (c_entry) 2 the fc1 backward launch's gate (trunk_done) 3 the conv backward launch's gate (fc1_done) 4 the Adam launch's last wait (env_done)
};
struct FbSplitCtx {
    hipStream_t tstream;                       // the side stream: acting forward + env step (the train chain stays on the caller's stream)
    FbSplitFlags *f;
    unsigned long long seq;                    // steps issued so far
    // HIP multiplexes streams onto a few hardware queues and promises no concurrency between two streams: on a shared queue a kernel
    // that waits for a word a LATER launch stores would sit in front of it until its time-out.  So the pair (side stream, caller's
    // stream) shakes hands once, both ways, with 20 ms waits, before the schedule is used with it (fb_split_probe)
    const void *probed_stream; int probed_ok;
};
int fb_split_probe(FbSplitCtx *ctx, void *stream);      // 1: the side stream and `stream` make progress independently of each other (synchronises both, once per stream)
// HIP deals a process's streams out over a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in turn and promises no concurrency
// between two of them: a side stream that lands on its caller's queue runs IN LINE with it (the prioritized memory's run-ahead tree work
// then cost configs[3] 680 us per step instead of 260), and a kernel on it that polls a word a later launch of the caller's stream
// stores would wait until its time-out.  fb_streams_concurrent: a hand-shake both ways with 20 ms waits (synchronises both streams);
// fb_side_stream_beside: `current` if it passes, else up to six fresh non-blocking streams of that priority (`current` is destroyed
// when it is replaced); *ok says whether the stream returned passes.
int fb_streams_concurrent(hipStream_t side, hipStream_t caller);
hipStream_t fb_side_stream_beside(hipStream_t caller, int priority, hipStream_t current, int *ok);
FbSplitCtx *fb_qnet_split_ctx(fb_qnet_t h);    // created on first use; NULL when the runtime lacks stream memory operations (the caller falls back)
// random.sample(range(n after the coming push), batch) -> idx on `stream`, opening ctx->gate at ctx->seq when the draw is clean; 1 when launched
// wait_last_round: the draw also waits until the acting trunk on the side stream has placed the first workgroup of its last round
int fb_replay_sample_gated(fb_replay_t h, int batch, int64_t *idx, const FbSplitCtx *ctx, void *stream, int wait_last_round = 0);
// one-wave launches on `stream`: wait until *flag >= v (bounded) / *flag = v behind whatever the stream holds
int fb_split_wait(const FbSplitCtx *ctx, const unsigned long long *flag, unsigned long long v, void *stream);
int fb_split_set(const FbSplitCtx *ctx, unsigned long long *flag, unsigned long long v, void *stream);
int fb_qnet_refresh_planes(fb_qnet_t h, void *stream);      // re-split whichever net's planes are stale (decided on the device)
int fb_qnet_profile_ring(fb_qnet_t h, int kernel, int reps, int algo, int batch, const FbRingSrc *ring, float *loss, void *stream);
// Memory append as a rider of the env step: every env workgroup stores its new frame / action / reward / terminal straight
// into the ring slot of the coming push (bits: slot of env 0's frame, +100 words per env; act / rew / term: row of the
// step, +1 per env; bits == NULL: no rider).  steps_dev receives steps_new (the device mirror of the push counter).
struct FbPushRider { unsigned long long *bits; uint8_t *act; float *rew; uint8_t *term; long long *steps_dev; long long steps_new; };
// library-internal (C++ linkage): the env step with the replay sampler as an extra workgroup, and the replay side of it
int fb_env_step_rider(fb_env_t h, const uint8_t *actions, uint8_t *frames, uint64_t *frame_bits, float *reward, uint8_t *terminal,
                      int32_t *score, const FbSampleRider *rider, const FbPushRider *push, const FbHeadRider *head, void *stream);
int fb_qnet_num_actions(fb_qnet_t h);
void *fb_qnet_get_grad_event(fb_qnet_t h);                 // the event fb_qnet_set_grad_event installed, or NULL
// the event the LAST gradient-exporting train step recorded behind its fc1 backward launch (NULL: none was recorded); reading clears it
void *fb_qnet_take_grad_event_recorded(fb_qnet_t h);
// argument checks fb_vec_step makes BEFORE any counter moves or any launch goes out (0 = fine, else the error code with
// fb_last_error set): the batch the train step would reject / the env count the acting forward would reject
int fb_qnet_check_step(fb_qnet_t h, int n_envs, int train_batch);
int fb_env_num_envs(fb_env_t h);
int fb_replay_num_envs(fb_replay_t h);
int fb_replay_is_prioritized(fb_replay_t h);
int fb_replay_update_priorities_keep(fb_replay_t h, int batch, const int64_t *idx, const float *abs_err, void *stream);       // in line, abs_err left untouched (fb_vec_step)
int fb_replay_update_priorities_ahead(fb_replay_t h, int batch, const int64_t *idx, const float *abs_err, void *stream);      // batch_update on the side stream (see fb_replay.hip); 1 when issued
int fb_replay_per_store_ahead(fb_replay_t h, void *stream);
int fb_replay_sample_ahead(fb_replay_t h, int batch, int64_t *idx, double *isw, float *isw32, void *stream);      // Memory.sample behind that store, on the same stream; 1 when issued      // the tree part of the coming push, ahead of it on a side stream (see fb_replay.hip)
int fb_env_can_carry_head(fb_env_t h);        // 1 when an env workgroup of the step launch has a wave per env it walks (<= 4 envs per workgroup)
// fb_qnet_act_nib without its last launch: conv1 .. fc1 are launched, *head describes the head_kernel work left over
// split (or NULL): the split schedule's context -- the fc1 launch stores trunk_done on arrival (the trunk in front of it has retired)
int fb_qnet_act_nib_rider(fb_qnet_t h, const uint8_t *nib_states, int n, float epsilon, uint64_t seed, uint64_t step,
                          uint8_t *actions, FbHeadRider *head, void *stream, const FbSplitCtx *split = nullptr);
// the rider for the frame / scalar part of "fb_replay_push" (returns 1, fills *push and COUNTS the push: the env launch that carries it
// must follow, and fb_replay_finish_push behind that launch: Memory.store's tree update of a prioritized memory -- joined if it ran
// ahead on the side stream, launched otherwise; nothing for a uniform memory)
int fb_replay_begin_push_rider(fb_replay_t h, FbPushRider *push);
int fb_replay_finish_push(fb_replay_t h, void *stream);
// fb_replay_sample that also leaves the importance weights as float32 (the loss's placeholder type), or NULL
int fb_replay_sample_f32(fb_replay_t h, int batch, const double *uniforms, int64_t *idx, double *isw, float *isw32, void *stream);
// the rider for "fb_replay_push; fb_replay_sample(batch) -> idx" (memory as it will be after `pushes_ahead` more pushes).  Returns 1 and
// fills *rider for a uniform memory with the CPython generator, 0 when the sampler cannot ride (PER, other generators).
int fb_replay_sample_rider(fb_replay_t h, int batch, int64_t *idx, FbSampleRider *rider, int pushes_ahead = 1);
// fb_qnet_train_step (fused Adam) with a random.sample rider in its conv3 backward launch and the NEXT step's gather in its
// Adam launch (either may be NULL)
int fb_qnet_train_step_rider(fb_qnet_t h, int algo, int batch, const uint8_t *s, const uint8_t *a, const float *r, const uint8_t *s2,
                             const uint8_t *t, double gamma, float *loss, const FbSampleRider *rider, const FbGatherRider *gather,
                             void *stream);
void fb_mt_init_genrand_host(FbMT *s, uint32_t seed);
void fb_mt_init_by_array_host(FbMT *s, const uint32_t *key, int n);
