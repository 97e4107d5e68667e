// fb_common.hip -- error channel, version, host-side MT19937 seeding.
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdlib.h>
#include <unistd.h>
#include "fb_common.h"

thread_local char fb_err_buf[512] = "";

int fb_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(fb_err_buf, sizeof(fb_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *fb_last_error(void) { return fb_err_buf; }

// Diagnostics: a process that dies in abort() (the HIP / ROCr runtimes do that on fatal errors, sometimes without a message) leaves its
// native call stack on stderr first -- and, when FB_ABORT_LOG names a file, in that file too: a test runner that captures file
// descriptor 2 (pytest's default) takes the runtime's own message and this trace down with it when the process dies.
// Async-signal-safe calls only; chains to whatever handler was installed before (Python's faulthandler prints the interpreter's
// stack after this one).  Idempotent: a second call keeps the first call's saved handler (saving our own handler as the
// "previous" one would make the handler chain to itself).
static struct sigaction fb_prev_abrt;
static bool fb_abrt_installed = false;
static int fb_abrt_log_fd = -1;
static void fb_abort_backtrace(int sig, siginfo_t *info, void *ctx) {
    static const char msg[] = "\n[libfbdqn] SIGABRT: native stack of the aborting thread\n";
    void *frames[64];
    const int n = backtrace(frames, 64);
    const int fds[2] = {2, fb_abrt_log_fd};
    for (int k = 0; k < 2; k++) {
        if (fds[k] < 0) continue;
        if (write(fds[k], msg, sizeof(msg) - 1) < 0) {}
        backtrace_symbols_fd(frames, n, fds[k]);
    }
    if (fb_prev_abrt.sa_flags & SA_SIGINFO) { if (fb_prev_abrt.sa_sigaction && fb_prev_abrt.sa_sigaction != fb_abort_backtrace) fb_prev_abrt.sa_sigaction(sig, info, ctx); }
    else if (fb_prev_abrt.sa_handler != SIG_DFL && fb_prev_abrt.sa_handler != SIG_IGN) fb_prev_abrt.sa_handler(sig);
}
extern "C" int fb_debug_abort_backtrace(void) {
    if (fb_abrt_installed) return FB_OK;
    void *warm[4];
    (void)backtrace(warm, 4);                        // (loads libgcc now: the first call allocates, which a signal handler must not)
    const char *path = getenv("FB_ABORT_LOG");
    if (path && *path) fb_abrt_log_fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = fb_abort_backtrace;
    sa.sa_flags = SA_SIGINFO | SA_RESETHAND;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGABRT, &sa, &fb_prev_abrt) != 0) return fb_set_error(FB_ERR_STATE, "sigaction(SIGABRT) failed");
    fb_abrt_installed = true;
    return FB_OK;
}
extern "C" int fb_version(void) { return 100; }

extern "C" int fb_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fb_set_error(FB_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

// Matsumoto & Nishimura mt19937ar.c seeding (CPython random.seed / numpy legacy seed use these)
void fb_mt_init_genrand_host(FbMT *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

void fb_mt_init_by_array_host(FbMT *s, const uint32_t *key, int key_length) {
    fb_mt_init_genrand_host(s, 19650218u);
    uint32_t *mt = s->mt;
    int i = 1, j = 0, k = 624 > key_length ? 624 : key_length;
    for (; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
        if (j >= key_length) j = 0;
    }
    for (k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000u;
    s->idx = 624;
}

// n x (random.sample -> minibatch -> _trainQNetwork) on a memory that is not being pushed to, as one host call.  Each step is the six
// launches of the ring-fed train step (conv trunk of the 2B states straight from the 1-bit frame ring -> fc1 -> loss + fc1 backward ->
// conv data gradients -> conv weight gradients -> Adam): no gather, no u8 minibatch.  Only the first draw gets a launch of its own: the
// draw for step i + 1 rides in step i's conv3 backward launch (it needs the generator and len(memory), nothing of step i).  What lets
// a step start with the trunk: the Adam launch leaves the split planes of W_conv1 / W_conv2 / W_conv3 behind (adam_fused_kernel), and
// every other writer of the parameters re-splits eagerly, so the conv planes are current without an acting forward in between.
// idx holds two index buffers used alternately; the indices of step i end up in idx[(i & 1) * batch ..).  s / s2 (the u8 minibatch
// of the gathered form) are only written when FB_TRAIN_STEPS_GATHER=1 selects that form (A/B knob: 7 launches per step).
extern "C" int fb_train_steps(fb_replay_t replay, fb_qnet_t net, int algo, int batch, int n_steps, int64_t *idx, uint8_t *s,
                              uint8_t *s2, uint8_t *a, float *r, uint8_t *t, float *loss, double gamma, void *stream) {
    FB_REQUIRE(replay && net && idx && s && s2 && a && r && t && loss && n_steps >= 1, "fb_train_steps: bad argument");
    FB_REQUIRE(algo != FB_ALGO_PER, "fb_train_steps: prioritized replay needs the importance weights: use the separate calls");
    static const bool gathered_form = getenv("FB_TRAIN_STEPS_GATHER") && atoi(getenv("FB_TRAIN_STEPS_GATHER")) == 1;
    int rc = fb_replay_sample(replay, batch, nullptr, idx, nullptr, stream);
    if (!gathered_form) {
        for (int i = 0; rc == FB_OK && i < n_steps; i++) {
            int64_t *cur = idx + (size_t)(i & 1) * batch, *nxt = idx + (size_t)((i + 1) & 1) * batch;
            FbRingSrc ring;
            rc = fb_replay_ring_src(replay, batch, cur, a, r, t, &ring);
            if (rc != FB_OK) break;
            FbSampleRider srider;
            const int rides = i + 1 < n_steps && fb_replay_sample_rider(replay, batch, nxt, &srider, 0);
            rc = fb_qnet_train_step_ring(net, algo, batch, &ring, nullptr, gamma, loss, nullptr, nullptr, stream, rides ? &srider : nullptr);
            if (rc == FB_OK && i + 1 < n_steps && !rides) rc = fb_replay_sample(replay, batch, nullptr, nxt, nullptr, stream);
        }
        return rc;
    }
    bool gathered = false;                               // the minibatch of step i is already in s / s2 / a / r / t
    for (int i = 0; rc == FB_OK && i < n_steps; i++) {
        int64_t *cur = idx + (size_t)(i & 1) * batch, *nxt = idx + (size_t)((i + 1) & 1) * batch;
        if (!gathered) rc = fb_replay_gather(replay, batch, cur, s, s2, a, r, t, stream);
        if (rc != FB_OK) break;
        FbSampleRider srider;
        FbGatherRider grider;
        const int rides = i + 1 < n_steps && fb_replay_sample_rider(replay, batch, nxt, &srider, 0) &&
                          fb_replay_gather_rider(replay, batch, nxt, s, s2, a, r, t, &grider);
        rc = fb_qnet_train_step_rider(net, algo, batch, s, a, r, s2, t, gamma, loss, rides ? &srider : nullptr,
                                      rides ? &grider : nullptr, stream);
        gathered = rides;
        if (rc == FB_OK && i + 1 < n_steps && !rides) rc = fb_replay_sample(replay, batch, nullptr, nxt, nullptr, stream);
    }
    return rc;
}

// ---- do two streams make progress independently of each other?  (see fb_common.h)
namespace {
__global__ void probe_set_kernel(unsigned long long *flag, unsigned long long v) { fb_flag_store(flag, v); }
__global__ void probe_wait_kernel(const unsigned long long *flag, unsigned long long v, unsigned *fail) {      // one wave, up to 20 ms
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    while (fb_flag_load(flag) < v) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 2000000LL) { atomicAdd(fail, 1u); break; }
    }
}
// the dispatch test below: a launch with far more workgroups than fit at once (60 KB of LDS each: two per CU) keeps its queue's dispatcher
// busy for its whole length; every workgroup stamps the 100 MHz clock on arrival (min) and on leaving (max)
__global__ __launch_bounds__(64) void probe_hold_kernel(unsigned long long *first, unsigned long long *last, long long ticks) {
    __shared__ char pad[60000];
    pad[threadIdx.x * 900] = 0;
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    atomicMin(first, (unsigned long long)t0);
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    atomicMax(last, (unsigned long long)wall_clock64() + (pad[0] ? 1 : 0));
}
__global__ void probe_stamp_kernel(unsigned long long *when) { *when = (unsigned long long)wall_clock64(); }
struct ProbeWords { unsigned long long a, pa[7], b, pb[7]; unsigned fail; unsigned pad; unsigned long long first, last, when, first2, last2; };
ProbeWords *probe_words[64] = {};            // per device (the current one of the calling thread)
unsigned long long probe_seq = 0;
}

int fb_streams_concurrent(hipStream_t S, hipStream_t C) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!probe_words[dev] && (hipMalloc((void **)&probe_words[dev], sizeof(ProbeWords)) != hipSuccess || hipMemset(probe_words[dev], 0, sizeof(ProbeWords)) != hipSuccess)) {
        probe_words[dev] = nullptr; (void)hipGetLastError(); return 0;
    }
    ProbeWords *W = probe_words[dev];
    const unsigned long long v = ++probe_seq;
    unsigned before = 0, after = 0;
    if (hipStreamSynchronize(S) != hipSuccess || hipStreamSynchronize(C) != hipSuccess) return 0;
    if (hipMemcpy(&before, &W->fail, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    // the waiter is launched FIRST each time: on a shared queue it sits in front of the launch that would release it
    hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(64), 0, S, (const unsigned long long *)&W->a, v, &W->fail);
    hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, C, &W->a, v);
    if (hipStreamSynchronize(S) != hipSuccess || hipStreamSynchronize(C) != hipSuccess) return 0;
    hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(64), 0, C, (const unsigned long long *)&W->b, v, &W->fail);
    hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, S, &W->b, v);
    if (hipStreamSynchronize(S) != hipSuccess || hipStreamSynchronize(C) != hipSuccess) return 0;
    if (hipMemcpy(&after, &W->fail, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (after != before) return 0;
    // ... and are they DISPATCHED beside each other?  Two queues on one pipe of the command processor take turns per launch: while the
    // caller's stream is handing out a grid that does not fit at once, nothing of the side stream starts -- both make progress (the
    // hand-shake above passes) and every kernel of the pair runs at a fraction of its speed (a vector step of 4096 envs: 478 us instead
    // of 204, tools/dbg_per_slow.py D7).  Test: a one-wave launch on the side stream issued right behind a 2048-workgroup launch of ~35 us
    // on the caller's must start in that launch's first half; best of three (a host hiccup between the two launches only delays it).
    double best = 1e9;
    for (int trial = 0; trial < 3 && best >= 0.5; trial++) {
        const unsigned long long init[3] = {~0ull, 0ull, 0ull};
        if (hipMemcpy(&W->first, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) return 0;
        hipLaunchKernelGGL(probe_hold_kernel, dim3(2048), dim3(64), 0, C, &W->first, &W->last, 800LL);
        hipLaunchKernelGGL(probe_stamp_kernel, dim3(1), dim3(1), 0, S, &W->when);
        if (hipStreamSynchronize(S) != hipSuccess || hipStreamSynchronize(C) != hipSuccess) return 0;
        unsigned long long st[3];
        if (hipMemcpy(st, &W->first, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return 0;
        if (st[1] <= st[0]) return 0;
        const double frac = st[2] <= st[0] ? 0.0 : (double)(st[2] - st[0]) / (double)(st[1] - st[0]);
        if (frac < best) best = frac;
    }
    // ... and the other way round: while ONE long-running workgroup sits on the side stream (the prioritized memory's tree kernels are exactly
    // that, on a highest-priority stream), a chain of small dependent launches on the caller's must keep its pace.  On a bad pairing --
    // seen with a highest-priority side stream and seven other live streams in the process -- every launch of the caller's stream takes
    // ~50 us whatever its size while the side stream is busy (configs[3]: 680 us per step instead of 258).  Test: eight one-wave launches
    // back to back on the caller's stream, alone and beside an 80 us workgroup on the side stream; at most 2 x + 10 us; best of three
    double worst_ratio = 1e9, alone_us = 0, beside_us = 0;
    for (int trial = 0; trial < 3 && worst_ratio >= 1.0; trial++) {
        unsigned long long st[5];
        const unsigned long long init[5] = {~0ull, 0ull, 0ull, ~0ull, 0ull};
        if (hipMemcpy(&W->first, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) return 0;
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(probe_hold_kernel, dim3(1), dim3(64), 0, C, &W->first, &W->last, 50LL);
        if (hipStreamSynchronize(C) != hipSuccess || hipMemcpy(st, &W->first, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess || st[1] <= st[0]) return 0;
        const double alone = (double)(st[1] - st[0]) / 100.0;
        if (hipMemcpy(&W->first, init, sizeof(init), hipMemcpyHostToDevice) != hipSuccess) return 0;
        hipLaunchKernelGGL(probe_hold_kernel, dim3(1), dim3(64), 0, S, &W->first2, &W->last2, 8000LL);        // one workgroup, 80 us
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(probe_hold_kernel, dim3(1), dim3(64), 0, C, &W->first, &W->last, 50LL);
        if (hipStreamSynchronize(S) != hipSuccess || hipStreamSynchronize(C) != hipSuccess) return 0;
        if (hipMemcpy(st, &W->first, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess || st[1] <= st[0]) return 0;
        const double beside = (double)(st[1] - st[0]) / 100.0, ratio = beside / (2.0 * alone + 10.0);
        if (ratio < worst_ratio) { worst_ratio = ratio; alone_us = alone; beside_us = beside; }
    }
    if (getenv("FB_SIDE_PROBE_DEBUG"))
        fprintf(stderr, "[fb] side stream %p beside %p: a launch on it started %.2f of the way through a long launch on the caller's; eight small launches on the caller's took %.1f us beside "
                        "a long workgroup on it, %.1f alone\n", (void *)S, (void *)C, best, beside_us, alone_us);
    return best < 0.5 && worst_ratio < 1.0;
}

hipStream_t fb_side_stream_beside(hipStream_t C, int priority, hipStream_t current, int *ok) {
    bool good = current && fb_streams_concurrent(current, C);
    for (int attempt = 0; !good && attempt < 6; attempt++) {
        hipStream_t ts = nullptr;
        if (hipStreamCreateWithPriority(&ts, hipStreamNonBlocking, priority) != hipSuccess) break;
        if (current) { (void)hipStreamSynchronize(current); (void)hipStreamDestroy(current); }
        current = ts;
        good = fb_streams_concurrent(current, C);
    }
    (void)hipGetLastError();
    if (ok) *ok = good ? 1 : 0;
    return current;
}

// which schedule fb_vec_step uses where both apply: 1 (default; FB_VEC_SPLIT=0 starts the process with 0) = the split schedule
static int fb_vec_split_flag = -1;
static bool fb_vec_split_enabled() {
    if (fb_vec_split_flag < 0) fb_vec_split_flag = !(getenv("FB_VEC_SPLIT") && atoi(getenv("FB_VEC_SPLIT")) == 0);
    return fb_vec_split_flag != 0;
}
extern "C" int fb_vec_step_set_schedule(int split) { fb_vec_split_flag = split ? 1 : 0; return FB_OK; }

// One step of the vectorised loop as a single host call: the five C-ABI calls of FlappyBirdDQN.py:72-76 back to back.
extern "C" int fb_vec_step(fb_env_t env, fb_replay_t replay, fb_qnet_t net, const fb_step_buffers *b, int n_envs, int algo,
                           int batch, float epsilon, uint64_t seed, uint64_t step, int train, double gamma, void *stream) {
    FB_REQUIRE(env && replay && net && b, "fb_vec_step: NULL handle");
    FB_REQUIRE(b->nib && b->actions && b->frame_bits && b->reward && b->terminal && b->score, "fb_vec_step: NULL env buffer");
    const bool per = algo == FB_ALGO_PER;
    if (per && train) FB_REQUIRE(b->isw && b->isw32 && b->abs_err, "fb_vec_step: the prioritized step needs the isw / isw32 / abs_err buffers");
    // every argument check of the calls below happens HERE, before the replay's push counter moves or anything is launched: a
    // rejected step must leave the handles exactly as they were (a counted push without its env launch would make every later
    // gather address a ring slot that was never written)
    FB_REQUIRE(algo >= 0 && algo <= 3, "fb_vec_step: unknown algo %d", algo);
    FB_REQUIRE(per == (fb_replay_is_prioritized(replay) != 0), "fb_vec_step: algo %d and the memory's kind (uniform / prioritized) do not match", algo);
    FB_REQUIRE(n_envs == fb_env_num_envs(env) && n_envs == fb_replay_num_envs(replay), "fb_vec_step: n_envs %d does not match the env (%d) / replay (%d) handles",
               n_envs, fb_env_num_envs(env), fb_replay_num_envs(replay));
    if (train) FB_REQUIRE(b->idx && b->s && b->s2 && b->a && b->r && b->t && b->loss, "fb_vec_step: NULL training buffer");
    {
        const int rc0 = fb_qnet_check_step(net, n_envs, train ? batch : -1);
        if (rc0 != FB_OK) return rc0;
    }
    // ---- The split schedule (uniform memory, small batches): act(k) and train(k) of the reference's loop BOTH read the weights Adam(k - 1)
    // left -- what orders them is the replay append between them (FlappyBirdDQN.py:72-76, BrainDQN.py:236-240), and that only matters when
    // the minibatch holds one of the n_envs transitions this very step appends (32 draws from a million slots: ~3 % of the steps).  So
    // acting + env go to a second stream beside the train chain; the draw (its population size is known on the host) decides on the
    // device whether the minibatch is clean, and waits for the env step itself when it is not -- the step then runs in the old order.
    // Same kernels, same inputs, same results bit for bit (the tests that pin fb_vec_step to the separate calls run through here); the
    // acting trunk's five states per workgroup leave a fifth of the chip to the chain beside it.
    // Hazards between the two streams, and what covers each: W_fc1's Adam span (conv backward launch) against the trunk's riding re-split
    // of W_fc1's planes -> the launch in front of it does not retire before the trunk has; the Adam launch (conv planes, biases, head
    // parameters, the version word) against the trunk and the fc1 launch (which records the version and copies the head's parameters
    // for the env launch's head rider) -> the launch in front of it does not retire before the fc1 launch has; workspaces -> the fused
    // acting forward has its own (hf_act / hp_act); the acting forward against the previous step's Adam and whatever else the caller's
    // stream held at entry -> c_entry.
    if (fb_vec_split_enabled() && train && !per && n_envs >= 256 && batch < 256 && fb_env_can_carry_head(env) && fb_qnet_num_actions(net) == 2) {
        hipStream_t A = fb_stream(stream);
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(A, &cap);
        FbSampleRider probe;
        FbSplitCtx *sc = cap == hipStreamCaptureStatusNone && fb_replay_sample_rider(replay, batch, b->idx, &probe) ? fb_qnet_split_ctx(net) : nullptr;
        if (sc && !fb_split_probe(sc, stream)) sc = nullptr;      // (the two streams share a hardware queue: one stream, as before)
        if (sc) {
            // Which chain goes where: the TRAIN chain stays on the caller's stream -- it is the longer one, step k + 1's follows step k's
            // in stream order with no hop, and whatever the caller puts on `stream` between calls (a target sync, reads of idx / loss) is
            // ordered against it for free.  Acting + env go to the net's side stream S.
            // How the chains hand over: through device words, not events -- a cross-stream event costs 6 - 13 us per use on this part even
            // when it has long been signalled (tools/mb/mb_gate.hip, profiles/r04_notes.md), a word that a kernel stores and another polls
            // ~1 (FbSplitFlags; every wait is bounded):
            //   S's first launch is one wave waiting for c_entry: the draw -- this call's first launch on `stream` -- has started, so the
            //     previous Adam and everything else the caller's stream held are done (its reads of the previous step's rewards included);
            //   the acting forward's fc1 launch stores trunk_done on arrival, the env launch behind it fc1_done, a one-thread launch behind that env_done;
            //   the draw does not retire before env_done (of the previous step: nothing else orders `stream` behind that push; of this step
            //     when the draw was not clean) -- one wave, so that nothing that spins holds what the kernels it waits for need --; a gate
            //     workgroup at the end of the fc1 backward launch waits for trunk_done (W_fc1's Adam span rides in the next launch), one at
            //     the end of the conv backward launch for fc1_done (Adam follows), and the Adam launch's last thread for env_done: on
            //     return every result of the step is ordered on `stream` as before.
            hipStream_t C = A, S = sc->tstream;
            FbSplitFlags *F = sc->f;
            sc->seq += 1;
            // (issue order on the host: the draw first -- it heads the longer chain, and the side stream's entry wave waits for it anyway)
            // (from 4096 envs on the acting side is the longer chain by far and its trunk takes several rounds of workgroups, the last one
            // partial -- 4096 envs: 52 of 820 -- : the train chain then starts with that last round, when most of the chip falls idle, and
            // the full rounds run undisturbed: 205 -> 200 us per step at 4096 envs; at 2048 envs what is left of the side chain by then is
            // shorter than the train chain, 123 -> 125, hence the threshold)
            static const bool last_round_on = !(getenv("FB_SPLIT_LAST_ROUND") && atoi(getenv("FB_SPLIT_LAST_ROUND")) == 0);      // A/B knob
            const int trunk_wgs = (n_envs + 4) / 5, wait_last_round = last_round_on && n_envs >= 4096 && trunk_wgs % 256 != 0;
            if (!fb_replay_sample_gated(replay, batch, b->idx, sc, C, wait_last_round)) { sc->seq -= 1; return fb_set_error(FB_ERR_HIP, "fb_vec_step: the gated draw could not be launched"); }
            int rc = fb_split_wait(sc, &F->c_entry, sc->seq, S);
            FbHeadRider hrider;
            if (rc == FB_OK) rc = fb_qnet_act_nib_rider(net, b->nib, n_envs, epsilon, seed, step, b->actions, &hrider, S, sc);
            // (a failure from here on leaves waits behind that nothing will satisfy: they give up after 1 s each, the error is returned)
            if (rc != FB_OK) { (void)hipStreamSynchronize(S); return rc; }
            hrider.on_arrival = &F->fc1_done; hrider.arrival_val = sc->seq;      // (the env launch is the next one on S: it stores fc1_done when it arrives)
            FbPushRider prider;
            fb_replay_begin_push_rider(replay, &prider);       // (counts the push: the minibatch below is addressed in the memory as it will be)
            FbRingSrc ring;
            rc = fb_replay_ring_src(replay, batch, b->idx, b->a, b->r, b->t, &ring);
            // (flat_grad: the data-parallel step -- the gradient is exported, and the fb_qnet_apply_adam / fb_dist_reduce_apply that completes
            // the step takes the Adam launch's place in the hand-over: until then `stream` is NOT yet ordered behind this step's env launch)
            if (rc == FB_OK) rc = fb_qnet_train_step_ring(net, algo, batch, &ring, nullptr, gamma, b->loss, nullptr, b->flat_grad, C, nullptr, sc);
            const int rc2 = fb_env_step_rider(env, b->actions, nullptr, b->frame_bits, b->reward, b->terminal, b->score, nullptr, &prider, &hrider, S);
            const int rc3 = fb_split_set(sc, &F->env_done, sc->seq, S);
            return rc != FB_OK ? rc : rc2 != FB_OK ? rc2 : rc3;
        }
    }
    // prioritized memory: Memory.store's tree update of this step's push goes out FIRST, on the memory's side stream -- it depends on the
    // tree as the previous step left it and on the env count, nothing else -- and runs beside the acting forward and the env step
    static const bool store_ahead = !(getenv("FB_PER_STORE_AHEAD") && atoi(getenv("FB_PER_STORE_AHEAD")) == 0);      // A/B knob
    int sampled = 0;                                 // ... and, behind it on that stream, Memory.sample of this step (it needs that tree and the memory's generator)
    if (per && store_ahead && fb_replay_per_store_ahead(replay, stream) && train)
        sampled = fb_replay_sample_ahead(replay, batch, b->idx, b->isw, b->isw32, stream);
    // the acting path's last kernel (fc2 + epsilon-greedy action, one wave per env) rides in the env launch as well when
    // an env workgroup has a wave for each of its envs there (up to four envs per workgroup: 8192 envs)
    FbHeadRider hrider;
    static const bool head_rides = !(getenv("FB_VEC_HEAD_RIDER") && atoi(getenv("FB_VEC_HEAD_RIDER")) == 0);      // tuning knob
    const int have_h = head_rides && fb_env_can_carry_head(env) && fb_qnet_num_actions(net) == 2;
    int rc = have_h ? fb_qnet_act_nib_rider(net, b->nib, n_envs, epsilon, seed, step, b->actions, &hrider, stream)
                    : fb_qnet_act_nib(net, b->nib, n_envs, epsilon, seed, step, b->actions, nullptr, stream);
    if (rc != FB_OK) return rc;
    // Riders of the env launch (uniform memory): random.sample of this step -- it only needs the size the memory will have
    // after the push, not the frames -- and the push itself: every env workgroup stores its transition straight into the
    // ring.  Memories that cannot ride keep their own launches (same results).
    FbSampleRider srider;
    FbPushRider prider;
    static const bool sample_rides = !(getenv("FB_VEC_SAMPLE_RIDER") && atoi(getenv("FB_VEC_SAMPLE_RIDER")) == 0);      // A/B knob
    const int have_s = train && sample_rides ? fb_replay_sample_rider(replay, batch, b->idx, &srider) : 0;      // before the push is counted
    const int have_p = fb_replay_begin_push_rider(replay, &prider);
    rc = fb_env_step_rider(env, b->actions, nullptr, b->frame_bits, b->reward, b->terminal, b->score, have_s ? &srider : nullptr,
                           have_p ? &prider : nullptr, have_h ? &hrider : nullptr, stream);
    if (rc != FB_OK) return rc;
    if (!have_p) {
        if (train && !have_s && !per) rc = fb_replay_push_sample(replay, nullptr, b->frame_bits, b->actions, b->reward, b->terminal, batch, b->idx, stream);
        else rc = fb_replay_push(replay, nullptr, b->frame_bits, b->actions, b->reward, b->terminal, stream);
    } else {
        rc = fb_replay_finish_push(replay, stream);      // (prioritized memory: the tree part of the push the env launch carried)
        if (rc == FB_OK && train && !have_s && !per) rc = fb_replay_sample(replay, batch, nullptr, b->idx, nullptr, stream);
    }
    if (rc != FB_OK || !train) return rc;
    if (per) {
        // BrainPrioritizedReplyDQN.py:277-329 from the sample on: importance weights -> weighted loss -> |TD errors| back into the tree
        // (fb_replay_sample above wrote the tree indices; its weights come as f64, the loss takes them as the float32 placeholder did)
        if (!sampled) rc = fb_replay_sample_f32(replay, batch, nullptr, b->idx, b->isw, b->isw32, stream);      // (else: joined with the push)
        if (rc != FB_OK) return rc;
        if (n_envs >= 256) {
            FbRingSrc ring;
            rc = fb_replay_ring_src(replay, batch, b->idx, b->a, b->r, b->t, &ring);
            if (rc == FB_OK) rc = fb_qnet_train_step_ring(net, algo, batch, &ring, b->isw32, gamma, b->loss, b->abs_err, b->flat_grad, stream);
        } else {
            rc = fb_replay_gather(replay, batch, b->idx, b->s, b->s2, b->a, b->r, b->t, stream);
            if (rc == FB_OK) rc = fb_qnet_train_step(net, algo, batch, b->s, b->a, b->r, b->s2, b->t, b->isw32, gamma, b->loss, b->abs_err, nullptr, b->flat_grad, stream);
        }
        if (rc != FB_OK) return rc;
        // Memory.batch_update: on the memory's side stream when it can run ahead (reference-order tree), in line otherwise; b->abs_err keeps
        // |TD error| as the loss left it in BOTH forms (the in-place `abs_errors += epsilon` of :147 stays inside the kernel; the
        // stand-alone fb_replay_update_priorities does it in the caller's array as the reference does)
        if (fb_replay_update_priorities_ahead(replay, batch, b->idx, b->abs_err, stream)) return FB_OK;
        return fb_replay_update_priorities_keep(replay, batch, b->idx, b->abs_err, stream);
    }
    // No gather (256 envs or more).  The train step's first launch reads the sampled transitions' 1-bit frames in the ring itself
    // (conv trunk per state) and leaves a / r / t behind; the split conv planes it needs are current because the acting forward above
    // has just refreshed them.  (b->s / b->s2 stay untouched then.)
    static const bool ring_on = !(getenv("FB_VEC_RING") && atoi(getenv("FB_VEC_RING")) == 0);      // tuning / A-B knob
    if (ring_on && n_envs >= 256) {
        FbRingSrc ring;
        rc = fb_replay_ring_src(replay, batch, b->idx, b->a, b->r, b->t, &ring);
        if (rc != FB_OK) return rc;
        return fb_qnet_train_step_ring(net, algo, batch, &ring, nullptr, gamma, b->loss, nullptr, b->flat_grad, stream);
    }
    rc = fb_replay_gather(replay, batch, b->idx, b->s, b->s2, b->a, b->r, b->t, stream);
    if (rc != FB_OK) return rc;
    return fb_qnet_train_step(net, algo, batch, b->s, b->a, b->r, b->s2, b->t, nullptr, gamma, b->loss, nullptr, nullptr, b->flat_grad,
                              stream);
}

extern "C" int fb_train_from_replay(fb_replay_t replay, fb_qnet_t net, int algo, int batch, const int64_t *idx, const float *isw, uint8_t *a,
                                    float *r, uint8_t *t, double gamma, float *loss, float *abs_err, float *flat_grad, void *stream) {
    FB_REQUIRE(replay && net && idx && a && r && t && loss, "fb_train_from_replay: NULL argument");
    FB_REQUIRE(algo >= 0 && algo <= 3, "fb_train_from_replay: unknown algo %d", algo);
    FB_REQUIRE(algo != FB_ALGO_PER || isw, "fb_train_from_replay: the prioritized step needs the importance weights");
    FB_REQUIRE(batch >= 1 && batch <= 256, "fb_train_from_replay: batch must be in 1..256");
    FbRingSrc ring;
    int rc = fb_replay_ring_src(replay, batch, idx, a, r, t, &ring);
    if (rc != FB_OK) return rc;
    // the conv planes are always current (adam_fused_kernel; init / load / sync re-split eagerly).  Only a batch of >= 256 also reads
    // W_fc1's planes (fc1_sp_kernel), which Adam leaves stale: re-split them on sight (two guarded launches)
    if (batch >= 256) {
        rc = fb_qnet_refresh_planes(net, stream);
        if (rc != FB_OK) return rc;
    }
    return fb_qnet_train_step_ring(net, algo, batch, &ring, isw, gamma, loss, abs_err, flat_grad, stream);
}

extern "C" int fb_profile_ring_kernel(fb_replay_t replay, fb_qnet_t net, int kernel, int reps, int algo, int batch, const int64_t *idx,
                                      uint8_t *a, float *r, uint8_t *t, float *loss, void *stream) {
    FB_REQUIRE(replay && net && idx && a && r && t && loss && reps >= 1, "fb_profile_ring_kernel: bad argument");
    FbRingSrc ring;
    int rc = fb_replay_ring_src(replay, batch, idx, a, r, t, &ring);
    if (rc != FB_OK) return rc;
    return fb_qnet_profile_ring(net, kernel, reps, algo, batch, &ring, loss, stream);
}
