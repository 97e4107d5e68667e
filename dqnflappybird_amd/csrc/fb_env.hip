// fb_env.hip -- batched headless Flappy Bird with the 80x80 preprocess fused in (gfx950).
//
// Reference semantics (paths relative to the reference checkout):
//   GameState.__init__ / frame_step   game/wrapped_flappy_bird.py:59-183
//   getRandomPipe                     game/wrapped_flappy_bird.py:208-221
//   checkCrash / pixelCollision       game/wrapped_flappy_bird.py:244-300
//   preprocess                        FlappyBirdDQN.py:31-34
//
// Design (DESIGN.md "Kernel E"):  one 256-thread workgroup walks envs e = blockIdx.x, +gridDim.x.
// The bird sprite (palette indices), the palette, the resize tap tables, the pre-rendered pipe / bird / ground
// row tables and the sprites' hit masks live in LDS (31 KB, staged once per workgroup).
// The 288x512x3 canvas of the reference is never built: every output pixel of the 80x80
// observation reads the four canvas pixels cv2.resize would read, resolved directly against the
// sprite that is on top there, then applies OpenCV's fixed-point bilinear / gray / threshold.
// Output columns c >= 63 only ever see the ground sprite (taps y >= 405, base drawn last over the
// pipes, bird never below y = 403), so they depend on basex alone (12 values) and come from a
// table computed at create time.  Pixels are emitted 64 at a time as one ballot word (the replay
// ring stores 1 bit / pixel) and optionally as u8 {0,255}.
#include <stdlib.h>
#include "fb_common.h"

namespace {

#include "fb_sampler.h"           // the replay sampler can ride in the step launch (fb_vec_step)
#include "fb_head.h"              // ... and so can the acting path's head (fc2 + epsilon-greedy action of this env)

constexpr int SW = 288, SH = 512, PIPE_W = 52, PIPE_H = 320, BIRD_W = 34, BIRD_H = 24;
#ifndef ENV_HITMASK
#define ENV_HITMASK 1   // 0: the per-pixel collision test against the sprite in global memory (A/B builds)
#endif
#ifndef ENV_EXIT
#define ENV_EXIT 0      // measurement builds only (tools/abl_env.sh): leave the step kernel after phase 1..4
#endif
constexpr int BASE_W = 336, BASE_H = 112, BASEY_I = 404, PLAYERX = 57, GAP = 100, OBS = 80;
#ifndef ENV_THREADS_N
#define ENV_THREADS_N 256
#endif
constexpr int ENV_THREADS = ENV_THREADS_N;  // 4 waves (measured: 512 threads 17.2 us in the loop at 1024 envs, 256: 11.8, 128: 13.7, 64: 18.5)
constexpr int NIB_PRE = (800 + ENV_THREADS - 1) / ENV_THREADS;   // a thread's words of the old nibble image
constexpr int GROUND_C0 = 63;    // first observation column whose taps all lie in the ground sprite
constexpr int PIPE_DX = PIPE_W + 1, BIRD_ROWS = 12;
constexpr size_t BLOB_BYTES = 8 + 4 + 1024 + PIPE_H * PIPE_W + 3 * BIRD_H * BIRD_W + BASE_H * BASE_W;

struct alignas(16) EnvLds {      // staged into LDS by every workgroup
    uint32_t pal[256];
    uint8_t bird[3 * BIRD_H * BIRD_W];
    uint32_t ground_bits[12 * OBS];      // [basex / -4][r] bit (c - 63)
    int16_t xo[OBS], xb0[OBS], xb1[OBS]; // game-x taps per output row r (cv vertical pass)
    int16_t yo[OBS], ya0[OBS], ya1[OBS]; // game-y taps per output column c (cv horizontal pass)
    // pre-rendered rows (fb_env_create): the 63 non-ground columns of an observation row that meets a pipe pair
    // with gap index g at x offset dx - 1 = xo[r] - pipe_x, for the row phase r % 5 (the taps repeat every 5 rows:
    // 3.6 * 5 = 18), and the up to 8 columns a bird at y % 32 lights in bird row rb (its x is fixed; the column
    // pattern repeats every 32 source pixels = 5 columns)
    unsigned long long pipe_mask[8 * PIPE_DX * 5];
    uint8_t bird_pat[3 * BIRD_ROWS * 32];
    uint8_t bird_cb[32];                 // first column of that pattern for y % 32
    int16_t bird_r0, bird_nr, pad_[6];   // observation rows whose x taps touch the bird
    // hit masks (wrapped_flappy_bird.py:285-300, getHitmask): bit c of row r = the sprite's pixel (r, c) is opaque
    unsigned long long pipe_hit[PIPE_H];
    unsigned long long bird_hit[3 * BIRD_H];
};
static_assert(sizeof(EnvLds) % 16 == 0, "EnvLds must be a multiple of 16 bytes");

struct EnvConst {
    EnvLds l;
    // the raw pipe sprite stays in global memory (L2): with the pre-rendered row masks only the pixel-exact collision
    // test and the rows where bird and pipe meet still read it, and those touch a few hundred bytes of it per step
    uint8_t pipe[PIPE_H * PIPE_W];
    uint8_t base[BASE_H * BASE_W];       // only fb_env_render_full needs the raw ground sprite
};

struct EnvParams {
    int n_envs;
    uint32_t seed_lo, seed_hi;
    int tape_len;
    const int8_t *tape;                  // [n_envs][tape_len] or null
    int32_t *state;                      // [n_envs][16]
    const EnvConst *cst;
    unsigned long long *err_count;
    uint8_t *nib;                        // optional [n_envs][FB_NIB_STRIDE]: 2 pixels x last 4 frames per byte, SAME-padded (caller owned)
    unsigned long long *stats;           // optional [4]: episodes ended, sum / max of their scores, pipes passed (caller owned)
};

// ------------------------------------------------------------------ device helpers
__device__ __forceinline__ int gap_y(int idx) { return 100 + 10 * idx; }   // 20+10*idx + int(404.48*0.2)

__device__ __forceinline__ int draw_gap(const EnvParams &p, int env, int32_t *st) {
    if (p.tape_len > 0) {
        int cur = st[15];
        int v = cur < p.tape_len ? p.tape[(size_t)env * p.tape_len + cur] : 0;
        st[15] = cur + 1;
        return v & 7;
    }
    fb_u4 o = fb_philox(p.seed_lo, p.seed_hi, (uint32_t)env, (uint32_t)st[14], FB_STREAM_GAP, 0u);
    st[14] += 1;
    return (int)(o.x >> 29);
}

__device__ __forceinline__ void env_reset(const EnvParams &p, int env, int32_t *st) {   // :59-85
    st[0] = 244; st[1] = 0; st[2] = 0; st[3] = 0; st[4] = 0; st[5] = 0; st[6] = 2;
    int g1 = draw_gap(p, env, st), g2 = draw_gap(p, env, st);
    st[7] = 288; st[8] = 432; st[9] = 0;
    st[10] = g1; st[11] = g2; st[12] = 0;
}

// palette index of the topmost pipe pixel at canvas (x, y), 0 if none; y < BASEY_I
__device__ __forceinline__ int pipe_at(const uint8_t *__restrict__ pipe, const int32_t *st, int x, int y) {
    int idx = 0;
    const int n = st[6];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        int col = x - st[7 + i];
        if (i < n && col >= 0 && col < PIPE_W) {
            int gy = gap_y(st[10 + i]);
            if (y < gy) idx = pipe[(gy - 1 - y) * PIPE_W + (PIPE_W - 1 - col)];     // upper = rotated by 180
            else if (y >= gy + GAP) idx = pipe[(y - gy - GAP) * PIPE_W + col];
        }
    }
    return idx;
}

__device__ __forceinline__ int bird_at(const EnvLds &L, const int32_t *st, int x, int y) {
    int bx = x - PLAYERX, by = y - st[0];
    if (bx >= 0 && bx < BIRD_W && by >= 0 && by < BIRD_H) return L.bird[(st[2] * BIRD_H + by) * BIRD_W + bx];
    return 0;
}

// OpenCV 8-bit INTER_LINEAR on 3 channels + BGR2GRAY (on RGB-ordered data) + threshold(>1)
__device__ __forceinline__ int resize_gray_bit(uint32_t s00, uint32_t s01, uint32_t s10, uint32_t s11,
                                               int a0, int a1, int b0, int b1) {
    int v[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int h0 = (int)((s00 >> (8 * k)) & 255u) * a0 + (int)((s01 >> (8 * k)) & 255u) * a1;
        int h1 = (int)((s10 >> (8 * k)) & 255u) * a0 + (int)((s11 >> (8 * k)) & 255u) * a1;
        v[k] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    }
    int gray = (v[0] * 1868 + v[1] * 9617 + v[2] * 4899 + (1 << 13)) >> 14;
    return gray > 1;
}

// 32-bit word of the padded nibble image (include/fbdqn.h, FB_NIB_*) that holds the 8 pixels of unpadded word w
__device__ __forceinline__ int nib_word(int w) { return (w / 10 + 2) * (FB_NIB_PITCH / 4) + 1 + w % 10; }

// ------------------------------------------------------------------ the step kernel
template <bool STEP>
__global__ __launch_bounds__(ENV_THREADS, 8) void env_kernel(EnvParams p, const uint8_t *__restrict__ actions,
                                                  uint8_t *__restrict__ frames,
                                                  unsigned long long *__restrict__ frame_bits,
                                                  float *__restrict__ reward, uint8_t *__restrict__ terminal,
                                                  int32_t *__restrict__ score, FbSampleRider rider, FbPushRider push,
                                                  FbHeadRider head) {
    __shared__ EnvLds L;
    __shared__ unsigned long long fw[100];          // the frame being assembled, 1 bit / pixel
    __shared__ unsigned long long rowm[OBS];        // columns 0..62 of every output row
    __shared__ int slow_rows[OBS];
    __shared__ int nslow;
    __shared__ int act_mail[ENV_THREADS / 64];      // the head rider's actions: wave w's result for the workgroup's w-th env
    // everything the first env of this workgroup needs from global memory is requested BEFORE the sprite tables are
    // waited for: state, action and the old nibble words travel together with the 25 KB of tables (one round trip
    // instead of three dependent ones)
    // fb_vec_step: the replay memory's random.sample rides as the FIRST workgroup (one wave of it).  The draw needs the
    // generator and the memory's size after the push that follows this step, not the frames produced here, and its
    // ~6 us dependent chain is shorter than this kernel -- in the push launch it was the longest chain of the launch.
    if (STEP && head.on_arrival && blockIdx.x == 0 && threadIdx.x == 0) fb_flag_store(head.on_arrival, head.arrival_val);
    const int rid = STEP && rider.k ? 1 : 0, nblk = gridDim.x - rid, bid = (int)blockIdx.x - rid;     // workgroup 0 = the rider
    if (bid < 0) {
        static_assert(sizeof(EnvLds) >= FB_SAMPLE_LDS_WORDS * 4, "the sampler borrows the sprite tables' LDS");
        uint32_t *words = reinterpret_cast<uint32_t *>(&L);
        if (threadIdx.x < 64) sample_cpython_body(rider.ctx, rider.k, rider.setsize, rider.out, words, reinterpret_cast<int *>(words + 624));
        return;
    }
    int32_t st0[16];
    int act0 = 0;
    uint32_t nib0[NIB_PRE] = {};
    {
        const int env = bid;                        // < n_envs: the grid never exceeds the env count
#pragma unroll
        for (int i = 0; i < 16; i++) st0[i] = p.state[(size_t)env * 16 + i];
        if (STEP && !head.on) act0 = actions[env];
        if (STEP && p.nib) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(p.nib + (size_t)env * FB_NIB_STRIDE);
#pragma unroll
            for (int k = 0; k < NIB_PRE; k++) {
                const int w = threadIdx.x + k * ENV_THREADS;
                nib0[k] = src[nib_word(w < 800 ? w : 0)];
            }
        }
        const uint4 *src = reinterpret_cast<const uint4 *>(&p.cst->l);
        uint4 *dst = reinterpret_cast<uint4 *>(&L);
        for (int i = threadIdx.x; i < (int)(sizeof(EnvLds) / 16); i += ENV_THREADS) dst[i] = src[i];
        // fb_vec_step: the acting path's head rides here -- wave 0 turns this env's fc1 partial sums into its Q values and
        // its epsilon-greedy action (head_one: head_kernel's own code) while the tables above are on their way
        // (up to ENV_THREADS / 64 envs per workgroup: wave w takes the w-th of them, env + w * nblk -- the host makes sure there are no more)
        if (STEP && head.on) {
            const int hw = threadIdx.x >> 6, he = env + hw * nblk;
            if (he < p.n_envs) {
                const int a = head_one_t<2>(head.c, head.params, he, threadIdx.x & 63);      // the game has two actions (the host checks)
                if ((threadIdx.x & 63) == 0) act_mail[hw] = a;     // handed to the other waves through LDS
            }
        }
    }
    __syncthreads();
#if ENV_EXIT == 1
    if (STEP) return;
#elif ENV_EXIT == 5     // ... with the head rider's action and the state loads kept alive
    if (STEP) { if (threadIdx.x == 0) score[bid] = st0[0] + (head.on ? act_mail[0] : act0); return; }
#elif ENV_EXIT == 6     // ... and the staged tables
    if (STEP) { if (threadIdx.x == 0) score[bid] = st0[0] + (head.on ? act_mail[0] : act0) + (int)L.bird_hit[st0[2] * BIRD_H] + (int)L.pal[st0[3]]; return; }
#elif ENV_EXIT == 7     // the staged tables without the head rider
    if (STEP) { if (threadIdx.x == 0) score[bid] = (int)L.bird_hit[(bid & 1) * BIRD_H] + (int)L.pal[bid & 255]; return; }
#endif
    if (STEP && head.on) act0 = act_mail[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (int env = bid; env < p.n_envs; env += nblk) {
        const bool first = env == bid;
        int32_t st[16];
#pragma unroll
        for (int i = 0; i < 16; i++) st[i] = st0[i];
        if (!first) {
#pragma unroll
            for (int i = 0; i < 16; i++) st[i] = p.state[(size_t)env * 16 + i];
        }

        float rew = 0.1f;
        int term = 0, score_ret = st[5], bad = 0, act = 0;
        if (STEP) {
            act = first ? act0 : head.on ? act_mail[(env - bid) / nblk] : (int)actions[env];
            bad = act > 1;                                         // ValueError('Multiple input actions!'), :99-100
            if (!bad) {
                int flapped = 0;
                if (act == 1 && st[0] > -2 * BIRD_H) { st[1] = -9; flapped = 1; }     // :105-108
                if (st[1] < 10 && !flapped) st[1] += 1;                               // :110-111
                const int ynew = st[0] + st[1];                   // min(velY, 380.48 - y), :115
                const int ground = ynew >= 380;                   // y + 24 >= 403.48, :252
                st[0] = ynew < 0 ? 0 : ynew;                      // :116-117
                if ((st[3] + 1) % 3 == 0) {                       // :120-121, cycle([0,1,2,1])
                    const int cyc = st[13] & 3;
                    st[2] = cyc == 3 ? 1 : cyc;
                    st[13] = (cyc + 1) & 3;
                }
                st[3] = (st[3] + 1) % 30;                         // :122
                st[4] = -((-st[4] + 100) % 48);                   // :123
#pragma unroll
                for (int i = 0; i < 3; i++) if (i < st[6]) st[7 + i] -= 4;            // :126-128
                if (0 < st[7] && st[7] < 5) {                     // :131-134
                    const int g = draw_gap(p, env, st);
                    if (st[6] == 2) { st[9] = 298; st[12] = g; } else { st[8] = 298; st[11] = g; }
                    st[6] += 1;
                }
                if (st[7] < -PIPE_W) {                            // :137-139
                    st[7] = st[8]; st[10] = st[11]; st[8] = st[9]; st[11] = st[12]; st[9] = 0; st[12] = 0;
                    st[6] -= 1;
                }
#pragma unroll
                for (int i = 0; i < 3; i++)                       // :142-148: x+26 <= 74 < x+30
                    if (i < st[6] && st[7 + i] <= 48 && st[7 + i] > 44) { st[5] += 1; rew = 3.0f; }
                // pixel-exact pipe collision, :255-273 -- every bird pixel against the pipe on top of it
                int hit = 0;
#if ENV_HITMASK
                // one thread per bird row: the row's 34 hit bits against the hit bits of every pipe under them (checkCrash, :255-273: a hit
                // on ANY pipe's mask counts), all from LDS -- no dependent global loads
                if (!ground && threadIdx.x < BIRD_H) {
                    const int by = threadIdx.x, y = st[0] + by;
                    const unsigned long long bm = L.bird_hit[st[2] * BIRD_H + by];
                    unsigned long long under = 0ull;              // bit bx: an opaque pipe pixel under bird column bx
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const int d = PLAYERX - st[7 + i], gy = gap_y(st[10 + i]);        // pipe column of bird column bx: bx + d
                        if (i < st[6] && d > -64 && d < PIPE_W && (y < gy || y >= gy + GAP)) {
                            const int row = y < gy ? gy - 1 - y : y - gy - GAP;
                            const unsigned long long raw = L.pipe_hit[row < PIPE_H ? row : PIPE_H - 1];
                            const unsigned long long pm = y < gy ? __brevll(raw) >> (64 - PIPE_W) : raw;     // upper pipe: rotated by 180
                            under |= d >= 0 ? pm >> d : pm << -d;
                        }
                    }
                    hit = y < BASEY_I && (under & bm) != 0ull;
                }
#else
                if (!ground) {
                    for (int i = threadIdx.x; i < BIRD_W * BIRD_H; i += ENV_THREADS) {
                        const int bx = i % BIRD_W, by = i / BIRD_W;
                        if (L.bird[(st[2] * BIRD_H + by) * BIRD_W + bx]) {
                            const int y = st[0] + by;
                            if (y < BASEY_I && pipe_at(p.cst->pipe, st, PLAYERX + bx, y)) hit = 1;
                        }
                    }
                }
#endif
                const int crash = __syncthreads_or(hit | ground);
#if ENV_EXIT == 2
                if (STEP) return;
#endif
                score_ret = st[5];                                // :155
                if (crash) { term = 1; env_reset(p, env, st); rew = -3.0f; }          // :157-162
            }
        }
        if (threadIdx.x == 0) {
            if (STEP) {
#pragma unroll
                for (int i = 0; i < 16; i++) p.state[(size_t)env * 16 + i] = st[i];
                reward[env] = bad ? 0.f : rew;
                terminal[env] = (uint8_t)term;
                score[env] = score_ret;
                if (push.bits) {         // Memory append (fb_replay_push) riding in this launch: the transition's scalars
                    push.act[env] = (uint8_t)act; push.rew[env] = bad ? 0.f : rew; push.term[env] = (uint8_t)term;
                    if (env == 0) *push.steps_dev = push.steps_new;
                }
                if (bad) atomicAdd(p.err_count, 1ull);
                if (p.stats) {           // device-side counters behind the reference's GAME_TIMES / score log lines: no host sync per step
                    if (term) { atomicAdd(&p.stats[0], 1ull); atomicAdd(&p.stats[1], (unsigned long long)score_ret); atomicMax(&p.stats[2], (unsigned long long)score_ret); }
                    if (rew == 3.0f) atomicAdd(&p.stats[3], 1ull);
                }
            }
        }
        // ---- observation, one wave per output row r (lane = column c < 63; columns >= 63 come from the ground
        // table).  Everything that depends on the row only -- which sprite column each of the two x taps falls
        // in -- is computed once per row in scalar registers; rows whose taps touch neither the bird nor a pipe
        // are all background and skip the per-pixel work.  The 80 bits of a row are OR-ed into the frame's 100
        // packed words in LDS (a row straddles two or three words), then written out coalesced.
        const int gidx = (-st[4]) >> 2;                           // basex in {0,-4,..,-44}
        for (int w = threadIdx.x; w < 100; w += ENV_THREADS) fw[w] = 0ull;
        if (threadIdx.x == 0) nslow = 0;
        __syncthreads();
        const int py = __builtin_amdgcn_readfirstlane(st[0]), pidx = __builtin_amdgcn_readfirstlane(st[2]);
        const int npipes = __builtin_amdgcn_readfirstlane(st[6]);
        int pxs[3], gys[3], gis[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            pxs[i] = __builtin_amdgcn_readfirstlane(st[7 + i]); gis[i] = __builtin_amdgcn_readfirstlane(st[10 + i]);
            gys[i] = gap_y(gis[i]);
        }
        // ---- observation.  Thread r < 80 owns output row r.  A row that meets only pipes, or only the bird, is a table
        // lookup (EnvLds::pipe_mask / bird_pat: the same four-tap arithmetic, done once on the host for every sprite
        // offset and tap phase); a row whose taps touch the bird AND a pipe goes to the per-pixel path below, which
        // resolves every tap against the sprite on top (at most the 9 bird rows, only while a pipe passes the bird).
        for (int r = threadIdx.x; r < OBS; r += ENV_THREADS) {
            const int x0 = L.xo[r], ph = r % 5;
            unsigned long long m = 0ull;
            bool haspipe = false;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int dx = x0 - pxs[i] + 1;
                if (i < npipes && dx >= 0 && dx < PIPE_DX) { haspipe = true; m |= L.pipe_mask[(gis[i] * PIPE_DX + dx) * 5 + ph]; }
            }
            const int rb = r - L.bird_r0;
            const bool hasbird = rb >= 0 && rb < L.bird_nr;
            if (hasbird && (haspipe || py < 0 || py >= 384)) slow_rows[atomicAdd(&nslow, 1)] = r;   // (y out of the table's range: only via set_state)
            else if (hasbird) {
                const int rmd = py & 31;
                m |= ((unsigned long long)L.bird_pat[(pidx * BIRD_ROWS + rb) * 32 + rmd] << (L.bird_cb[rmd] + 5 * (py >> 5))) & 0x7FFFFFFFFFFFFFFFull;
            }
            rowm[r] = m;
        }
        __syncthreads();
        const int wv = __builtin_amdgcn_readfirstlane(wave), ns = nslow;
        for (int q = wv; q < ns; q += ENV_THREADS / 64) {
            const int r = slow_rows[q];
            const int x0 = L.xo[r];
            int bcol[2], pcol[2], pgy[2];                         // per x tap: bird column / pipe column (or -1) and its gap
#pragma unroll
            for (int tx = 0; tx < 2; tx++) {
                const int x = x0 + tx;
                bcol[tx] = (x >= PLAYERX && x < PLAYERX + BIRD_W) ? x - PLAYERX : -1;
                pcol[tx] = -1; pgy[tx] = 0;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const int col = x - pxs[i];
                    if (i < npipes && col >= 0 && col < PIPE_W) { pcol[tx] = col; pgy[tx] = gys[i]; }
                }
            }
            const int c = lane < GROUND_C0 ? lane : 0;
            const int y0 = L.yo[c];
            int idx[2][2];
#pragma unroll
            for (int tx = 0; tx < 2; tx++)
#pragma unroll
                for (int ty = 0; ty < 2; ty++) {
                    const int y = y0 + ty, by = y - py;
                    int v = 0;
                    if (bcol[tx] >= 0 && by >= 0 && by < BIRD_H) v = L.bird[(pidx * BIRD_H + by) * BIRD_W + bcol[tx]];
                    if (!v && pcol[tx] >= 0) {
                        if (y < pgy[tx]) v = p.cst->pipe[(pgy[tx] - 1 - y) * PIPE_W + (PIPE_W - 1 - pcol[tx])];
                        else if (y >= pgy[tx] + GAP) v = p.cst->pipe[(y - pgy[tx] - GAP) * PIPE_W + pcol[tx]];
                    }
                    idx[tx][ty] = v;
                }
            int bit = 0;
            if (idx[0][0] | idx[0][1] | idx[1][0] | idx[1][1])
                bit = resize_gray_bit(L.pal[idx[0][0]], L.pal[idx[0][1]], L.pal[idx[1][0]], L.pal[idx[1][1]], L.ya0[c],
                                      L.ya1[c], L.xb0[r], L.xb1[r]);
            const unsigned long long m = __ballot(bit && lane < GROUND_C0);
            if (lane == 0) rowm[r] = m;
        }
        __syncthreads();
#if ENV_EXIT == 3
        if (STEP) return;
#endif
        // a row's 80 bits (63 rendered + 17 ground) are OR-ed into the frame's 100 packed words (a row straddles two or three)
        for (int r = threadIdx.x; r < OBS; r += ENV_THREADS) {
            const unsigned long long g = L.ground_bits[gidx * OBS + r];           // 17 bits, columns 63..79
            const int p0 = r * OBS, w0 = p0 >> 6, sh = p0 & 63;                   // row = bits [p0, p0 + 80)
            const unsigned long long lo = rowm[r] | (g << GROUND_C0);             // columns 0..63 (bit 63 = column 63)
            const unsigned long long hi = g >> 1;                                 // columns 64..79
            atomicOr(&fw[w0], lo << sh);
            if (sh) atomicOr(&fw[w0 + 1], (lo >> (64 - sh)) | (hi << sh));
            else atomicOr(&fw[w0 + 1], hi);
            if (sh > 48) atomicOr(&fw[w0 + 2], hi >> (64 - sh));
        }
        __syncthreads();
#if ENV_EXIT == 4
        if (STEP) return;
#endif
        if (frame_bits) for (int w = threadIdx.x; w < 100; w += ENV_THREADS) frame_bits[(size_t)env * 100 + w] = fw[w];
        if (STEP && push.bits) for (int w = threadIdx.x; w < 100; w += ENV_THREADS) push.bits[(size_t)env * 100 + w] = fw[w];   // ... and its frame
        if (p.nib) {
            // the agent's 4-frame stack (BrainDQN.py:68,238-239: newest last, never reset) as one nibble per pixel:
            // bit f of a pixel's nibble = frame f of the stack; a step shifts the nibbles down and puts the new
            // frame on top, the initial observation fills all four frames.  One u32 = 8 pixels.
            uint32_t *dst = reinterpret_cast<uint32_t *>(p.nib + (size_t)env * FB_NIB_STRIDE);
            if (!STEP) {                      // the initial observation also (re)writes conv1's zero padding around the image
                for (int w = threadIdx.x; w < FB_NIB_STRIDE / 4; w += ENV_THREADS) {
                    const int row = w / (FB_NIB_PITCH / 4), col = w - row * (FB_NIB_PITCH / 4);
                    if (row < 2 || row >= 82 || col == 0) dst[w] = 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < NIB_PRE; k++) {
                const int w = threadIdx.x + k * ENV_THREADS;
                if (w >= 800) break;
                const unsigned t = (unsigned)(fw[w >> 3] >> ((w & 7) * 8)) & 0xFFu;       // the 8 new pixel bits
                uint32_t top = 0;
#pragma unroll
                for (int m = 0; m < 8; m++) top |= ((t >> m) & 1u) << (4 * m + 3);
                uint32_t old = 0u;
                const int pw = nib_word(w);
                if (STEP) old = first ? nib0[k] : dst[pw];                                // (k is a compile-time index: the loop is unrolled)
                dst[pw] = STEP ? (((old >> 1) & 0x77777777u) | top) : (top | (top >> 1) | (top >> 2) | (top >> 3));
            }
        }
        if (frames) {
            for (int q = threadIdx.x; q < 1600; q += ENV_THREADS) {                       // 4 pixels -> one 32-bit store
                const unsigned int nib = (unsigned int)(fw[q >> 4] >> ((q & 15) * 4)) & 0xFu;
                const unsigned int v = ((nib & 1u) * 0xFFu) | (((nib >> 1) & 1u) * 0xFF00u) | (((nib >> 2) & 1u) * 0xFF0000u) |
                                       (((nib >> 3) & 1u) * 0xFF000000u);
                reinterpret_cast<unsigned int *>(frames + (size_t)env * 6400)[q] = v;
            }
        }
        __syncthreads();                                          // fw is reused by the next env of this workgroup
    }
}

// array3d of one env, for parity with the reference's image_data (:177)
__global__ __launch_bounds__(256) void render_full_kernel(EnvParams p, int env, uint8_t *__restrict__ rgb) {
    const EnvConst &C = *p.cst;
    int32_t st[16];
#pragma unroll
    for (int i = 0; i < 16; i++) st[i] = p.state[(size_t)env * 16 + i];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < SW * SH; i += gridDim.x * 256) {
        const int x = i / SH, y = i - x * SH;
        int idx = bird_at(C.l, st, x, y);
        if (!idx) {
            if (y >= BASEY_I) { const int by = y - BASEY_I; idx = by < BASE_H ? C.base[by * BASE_W + (x - st[4])] : 0; }
            else idx = pipe_at(C.pipe, st, x, y);
        }
        const uint32_t c = C.l.pal[idx];
        rgb[(size_t)i * 3 + 0] = c & 255; rgb[(size_t)i * 3 + 1] = (c >> 8) & 255; rgb[(size_t)i * 3 + 2] = (c >> 16) & 255;
    }
}

// preprocess() of FlappyBirdDQN.py:31-34 for arbitrary array3d frames u8[n][288][512][3]
// (the fused env kernel never needs this; it is the drop-in for callers that hold RGB frames)
__global__ __launch_bounds__(256) void preprocess_kernel(const EnvConst *__restrict__ cst, const uint8_t *__restrict__ rgb,
                                                         int n, uint8_t *__restrict__ out) {
    const EnvLds &T = cst->l;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < (long long)n * 6400; idx += (long long)gridDim.x * 256) {
        const int f = (int)(idx / 6400), pix = (int)(idx - (long long)f * 6400), r = pix / OBS, c = pix - r * OBS;
        const uint8_t *img = rgb + (size_t)f * SW * SH * 3;
        const int x0 = T.xo[r], y0 = T.yo[c];
        uint32_t s[2][2];
#pragma unroll
        for (int tx = 0; tx < 2; tx++)
#pragma unroll
            for (int ty = 0; ty < 2; ty++) {
                const uint8_t *p = img + ((size_t)(x0 + tx) * SH + (y0 + ty)) * 3;
                s[tx][ty] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
            }
        out[idx] = resize_gray_bit(s[0][0], s[0][1], s[1][0], s[1][1], T.ya0[c], T.ya1[c], T.xb0[r], T.xb1[r]) ? 255 : 0;
    }
}

__global__ void env_reset_kernel(EnvParams p) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= p.n_envs) return;
    int32_t st[16];
#pragma unroll
    for (int i = 0; i < 16; i++) st[i] = 0;
    env_reset(p, env, st);
#pragma unroll
    for (int i = 0; i < 16; i++) p.state[(size_t)env * 16 + i] = st[i];
}

// ------------------------------------------------------------------ host side
// cv2.resize(.., (80,80)) INTER_LINEAR tap tables (OpenCV resize.cpp, 8-bit path)
void linear_tab(int dst, int src, int16_t *ofs, int16_t *c0, int16_t *c1) {
    const double scale = 1.0 / ((double)dst / src);
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { s = 0; f = 0; }
        if (s >= src - 1) { s = src - 1; f = 0; }
        ofs[d] = (int16_t)s;
        c0[d] = (int16_t)lrintf((1.f - f) * 2048.f);
        c1[d] = (int16_t)lrintf(f * 2048.f);
    }
}

int host_gray_bit(uint32_t s00, uint32_t s01, uint32_t s10, uint32_t s11, int a0, int a1, int b0, int b1) {
    int v[3];
    for (int k = 0; k < 3; k++) {
        int h0 = (int)((s00 >> (8 * k)) & 255u) * a0 + (int)((s01 >> (8 * k)) & 255u) * a1;
        int h1 = (int)((s10 >> (8 * k)) & 255u) * a0 + (int)((s11 >> (8 * k)) & 255u) * a1;
        v[k] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    }
    return ((v[0] * 1868 + v[1] * 9617 + v[2] * 4899 + (1 << 13)) >> 14) > 1;
}

}  // namespace

struct fb_env {
    EnvParams p;
    EnvConst *d_const;
    int8_t *d_tape; size_t tape_bytes;
    int grid;
};

extern "C" int fb_env_create(int n_envs, uint64_t seed, uint32_t flags, const void *blob, size_t blob_bytes,
                             fb_env_t *out) {
    (void)flags;
    FB_REQUIRE(out != nullptr, "fb_env_create: out is NULL");
    FB_REQUIRE(n_envs > 0 && n_envs <= (1 << 22), "fb_env_create: n_envs=%d out of range", n_envs);
    FB_REQUIRE(blob && blob_bytes == BLOB_BYTES && memcmp(blob, "FBSPR001", 8) == 0,
               "fb_env_create: sprite blob must be %zu bytes with magic FBSPR001 (tools/make_assets.py)", BLOB_BYTES);
    EnvConst *hc = new EnvConst();
    const uint8_t *b = (const uint8_t *)blob + 12;
    memcpy(hc->l.pal, b, 1024); b += 1024;
    memcpy(hc->pipe, b, sizeof(hc->pipe)); b += sizeof(hc->pipe);
    memcpy(hc->l.bird, b, sizeof(hc->l.bird)); b += sizeof(hc->l.bird);
    memcpy(hc->base, b, sizeof(hc->base));
    linear_tab(OBS, SH, hc->l.yo, hc->l.ya0, hc->l.ya1);      // cv columns = game y (512)
    linear_tab(OBS, SW, hc->l.xo, hc->l.xb0, hc->l.xb1);      // cv rows    = game x (288)
    // the ground-only region: every tap of columns >= GROUND_C0 must be inside the ground sprite
    if (hc->l.yo[GROUND_C0] < BASEY_I || hc->l.yo[GROUND_C0 - 1] + 1 >= BASEY_I) {
        delete hc;
        return fb_set_error(FB_ERR_INVALID, "fb_env_create: ground column split does not match the tap table");
    }
    for (int g = 0; g < 12; g++)
        for (int r = 0; r < OBS; r++) {
            uint32_t bits = 0;
            for (int c = GROUND_C0; c < OBS; c++) {
                uint32_t s[2][2];
                for (int tx = 0; tx < 2; tx++)
                    for (int ty = 0; ty < 2; ty++) {
                        int x = hc->l.xo[r] + tx, y = hc->l.yo[c] + ty;
                        s[tx][ty] = hc->l.pal[hc->base[(y - BASEY_I) * BASE_W + (x + 4 * g)]];
                    }
                bits |= (uint32_t)host_gray_bit(s[0][0], s[0][1], s[1][0], s[1][1], hc->l.ya0[c], hc->l.ya1[c],
                                                hc->l.xb0[r], hc->l.xb1[r]) << (c - GROUND_C0);
            }
            hc->l.ground_bits[g * OBS + r] = bits;
        }
    // pre-rendered rows.  The tap tables repeat: 5 rows = 18 source pixels, 5 columns = 32 source pixels.
    for (int r = 0; r + 5 < OBS; r++)
        if (hc->l.xo[r + 5] != hc->l.xo[r] + 18 || hc->l.xb0[r + 5] != hc->l.xb0[r] || hc->l.xb1[r + 5] != hc->l.xb1[r] ||
            hc->l.yo[r + 5] != hc->l.yo[r] + 32 || hc->l.ya0[r + 5] != hc->l.ya0[r] || hc->l.ya1[r + 5] != hc->l.ya1[r]) {
            delete hc;
            return fb_set_error(FB_ERR_INVALID, "fb_env_create: tap tables are not 5-periodic");
        }
    auto pipe_px = [&](int g, int col, int y) -> int {         // wrapped_flappy_bird.py:165-170, upper pipe rotated by 180
        if (col < 0 || col >= PIPE_W) return 0;
        const int gy = 100 + 10 * g;
        if (y < gy) return hc->pipe[(gy - 1 - y) * PIPE_W + (PIPE_W - 1 - col)];
        if (y >= gy + GAP) return hc->pipe[(y - gy - GAP) * PIPE_W + col];
        return 0;
    };
    for (int g = 0; g < 8; g++)
        for (int dx = 0; dx < PIPE_DX; dx++)
            for (int ph = 0; ph < 5; ph++) {
                unsigned long long m = 0;
                for (int c = 0; c < GROUND_C0; c++) {
                    uint32_t s2[2][2];
                    for (int tx = 0; tx < 2; tx++)
                        for (int ty = 0; ty < 2; ty++) s2[tx][ty] = hc->l.pal[pipe_px(g, dx - 1 + tx, hc->l.yo[c] + ty)];
                    m |= (unsigned long long)host_gray_bit(s2[0][0], s2[0][1], s2[1][0], s2[1][1], hc->l.ya0[c], hc->l.ya1[c],
                                                           hc->l.xb0[ph], hc->l.xb1[ph]) << c;
                }
                hc->l.pipe_mask[(g * PIPE_DX + dx) * 5 + ph] = m;
            }
    for (int r = 0; r < PIPE_H; r++) {
        unsigned long long m = 0;
        for (int c = 0; c < PIPE_W; c++) m |= (unsigned long long)(hc->pipe[r * PIPE_W + c] != 0) << c;
        hc->l.pipe_hit[r] = m;
    }
    for (int r = 0; r < 3 * BIRD_H; r++) {
        unsigned long long m = 0;
        for (int c = 0; c < BIRD_W; c++) m |= (unsigned long long)(hc->l.bird[r * BIRD_W + c] != 0) << c;
        hc->l.bird_hit[r] = m;
    }
    int br0 = -1, bnr = 0;
    for (int r = 0; r < OBS; r++)
        if (hc->l.xo[r] + 1 >= PLAYERX && hc->l.xo[r] < PLAYERX + BIRD_W) { if (br0 < 0) br0 = r; bnr = r - br0 + 1; }
    if (br0 < 0 || bnr > BIRD_ROWS) { delete hc; return fb_set_error(FB_ERR_INVALID, "fb_env_create: bird row table too small"); }
    hc->l.bird_r0 = (int16_t)br0; hc->l.bird_nr = (int16_t)bnr;
    for (int rmd = 0; rmd < 32; rmd++) {
        int cb = 0;
        while (cb < OBS - 8 && hc->l.yo[cb] + 1 < rmd) cb++;     // first column with a tap at y >= rmd
        hc->l.bird_cb[rmd] = (uint8_t)cb;
        for (int pi = 0; pi < 3; pi++)
            for (int rb = 0; rb < bnr; rb++) {
                const int r = br0 + rb;
                unsigned pat = 0;
                for (int k = 0; k < 8; k++) {
                    const int c = cb + k;
                    uint32_t s2[2][2];
                    for (int tx = 0; tx < 2; tx++)
                        for (int ty = 0; ty < 2; ty++) {
                            const int bx = hc->l.xo[r] + tx - PLAYERX, by = hc->l.yo[c] + ty - rmd;
                            const int v = bx >= 0 && bx < BIRD_W && by >= 0 && by < BIRD_H ? hc->l.bird[(pi * BIRD_H + by) * BIRD_W + bx] : 0;
                            s2[tx][ty] = hc->l.pal[v];
                        }
                    pat |= (unsigned)host_gray_bit(s2[0][0], s2[0][1], s2[1][0], s2[1][1], hc->l.ya0[c], hc->l.ya1[c], hc->l.xb0[r],
                                                   hc->l.xb1[r]) << k;
                }
                // the pattern must fit its 8 columns: nothing of the bird may reach column cb + 8
                hc->l.bird_pat[(pi * BIRD_ROWS + rb) * 32 + rmd] = (uint8_t)pat;
            }
        if (hc->l.yo[cb + 8] <= rmd + BIRD_H - 1) { delete hc; return fb_set_error(FB_ERR_INVALID, "fb_env_create: bird pattern wider than 8 columns"); }
    }
    fb_env *h = new fb_env();
    memset(h, 0, sizeof(*h));
    hipError_t e = hipMalloc(&h->d_const, sizeof(EnvConst));
    if (e == hipSuccess) e = hipMemcpy(h->d_const, hc, sizeof(EnvConst), hipMemcpyHostToDevice);
    delete hc;
    if (e == hipSuccess) e = hipMalloc(&h->p.state, sizeof(int32_t) * 16 * (size_t)n_envs);
    if (e == hipSuccess) e = hipMalloc(&h->p.err_count, sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(h->p.err_count, 0, sizeof(unsigned long long));
    if (e != hipSuccess) {
        fb_set_error(FB_ERR_HIP, "fb_env_create: %s", hipGetErrorString(e));
        fb_env_destroy(h);
        return FB_ERR_HIP;
    }
    h->p.n_envs = n_envs;
    h->p.seed_lo = (uint32_t)seed; h->p.seed_hi = (uint32_t)(seed >> 32);
    h->p.cst = h->d_const;
    int cap = 2048;                                  // workgroups of the step launch: one per env up to here, then they stride
    if (const char *g = getenv("FB_ENV_GRID_CAP")) { int v = atoi(g); if (v >= 256 && v <= 65536) cap = v; }      // tuning knob
    h->grid = n_envs < cap ? n_envs : cap;
    if (const char *g = getenv("FB_ENV_GRID")) { int v = atoi(g); if (v > 0 && v < h->grid) h->grid = v; }   // tuning knob
    *out = h;
    return fb_env_reset(h, nullptr);
}

extern "C" int fb_env_destroy(fb_env_t h) {
    if (!h) return FB_OK;
    if (h->d_const) (void)hipFree(h->d_const);
    if (h->p.state) (void)hipFree(h->p.state);
    if (h->p.err_count) (void)hipFree(h->p.err_count);
    if (h->d_tape) (void)hipFree(h->d_tape);
    delete h;
    return FB_OK;
}

extern "C" int fb_env_reset(fb_env_t h, void *stream) {
    FB_REQUIRE(h, "fb_env_reset: NULL handle");
    hipLaunchKernelGGL(env_reset_kernel, dim3((h->p.n_envs + 255) / 256), dim3(256), 0, fb_stream(stream), h->p);
    FB_LAUNCH_CHECK();
    return FB_OK;
}

extern "C" int fb_env_step(fb_env_t h, const uint8_t *actions, uint8_t *frames, uint64_t *frame_bits, float *reward,
                           uint8_t *terminal, int32_t *score, void *stream) {
    return fb_env_step_rider(h, actions, frames, frame_bits, reward, terminal, score, nullptr, nullptr, nullptr, stream);
}

int fb_env_can_carry_head(fb_env_t h) { return h && (long long)h->grid * (ENV_THREADS / 64) >= h->p.n_envs; }      // a wave per env of the workgroup
int fb_env_num_envs(fb_env_t h) { return h ? h->p.n_envs : 0; }

int fb_env_step_rider(fb_env_t h, const uint8_t *actions, uint8_t *frames, uint64_t *frame_bits, float *reward, uint8_t *terminal,
                      int32_t *score, const FbSampleRider *rider, const FbPushRider *push, const FbHeadRider *head, void *stream) {
    FB_REQUIRE(h && actions && reward && terminal && score, "fb_env_step: NULL argument");
    FB_REQUIRE(!head || (fb_env_can_carry_head(h) && head->c.A == 2), "fb_env_step: the head rider needs at most %d envs per workgroup and a 2-action net", ENV_THREADS / 64);
    FbSampleRider r;
    FbPushRider q;
    FbHeadRider hd;
    memset(&r, 0, sizeof(r)); memset(&q, 0, sizeof(q)); memset(&hd, 0, sizeof(hd));
    if (rider) r = *rider;
    if (push) q = *push;
    if (head) hd = *head;
    hipLaunchKernelGGL(env_kernel<true>, dim3(h->grid + (r.k ? 1 : 0)), dim3(ENV_THREADS), 0, fb_stream(stream), h->p, actions, frames,
                       (unsigned long long *)frame_bits, reward, terminal, score, r, q, hd);
    FB_LAUNCH_CHECK();
    return FB_OK;
}

extern "C" int fb_env_observe(fb_env_t h, uint8_t *frames, uint64_t *frame_bits, void *stream) {
    FB_REQUIRE(h && (frames || frame_bits), "fb_env_observe: NULL argument");
    hipLaunchKernelGGL(env_kernel<false>, dim3(h->grid), dim3(ENV_THREADS), 0, fb_stream(stream), h->p,
                       (const uint8_t *)nullptr, frames, (unsigned long long *)frame_bits, (float *)nullptr,
                       (uint8_t *)nullptr, (int32_t *)nullptr, FbSampleRider{}, FbPushRider{}, FbHeadRider{});
    FB_LAUNCH_CHECK();
    return FB_OK;
}

extern "C" int fb_env_get_state(fb_env_t h, int32_t *state_host) {
    FB_REQUIRE(h && state_host, "fb_env_get_state: NULL argument");
    FB_CHECK_HIP(hipDeviceSynchronize());
    FB_CHECK_HIP(hipMemcpy(state_host, h->p.state, sizeof(int32_t) * 16 * (size_t)h->p.n_envs, hipMemcpyDeviceToHost));
    return FB_OK;
}

extern "C" int fb_env_set_state(fb_env_t h, const int32_t *state_host) {
    FB_REQUIRE(h && state_host, "fb_env_set_state: NULL argument");
    for (int e = 0; e < h->p.n_envs; e++) {
        const int32_t *s = state_host + (size_t)e * 16;
        FB_REQUIRE(s[6] >= 1 && s[6] <= 3 && s[2] >= 0 && s[2] <= 2 && s[4] <= 0 && s[4] > -48 && (s[4] & 3) == 0 &&
                       s[10] >= 0 && s[10] < 8 && s[11] >= 0 && s[11] < 8 && s[12] >= 0 && s[12] < 8,
                   "fb_env_set_state: env %d has an impossible state", e);
    }
    FB_CHECK_HIP(hipDeviceSynchronize());
    FB_CHECK_HIP(hipMemcpy(h->p.state, state_host, sizeof(int32_t) * 16 * (size_t)h->p.n_envs, hipMemcpyHostToDevice));
    return FB_OK;
}

extern "C" int fb_env_set_gap_tape(fb_env_t h, const int8_t *tape_host, int tape_len) {
    FB_REQUIRE(h && tape_len >= 0 && (tape_len == 0 || tape_host), "fb_env_set_gap_tape: bad argument");
    FB_CHECK_HIP(hipDeviceSynchronize());
    // the single-env GameState shim hands over a fresh 3-entry tape before every frame_step: keep the allocation while the
    // size fits and only copy
    const size_t n = (size_t)h->p.n_envs * tape_len;
    h->p.tape = nullptr; h->p.tape_len = 0;
    if (tape_len > 0) {
        if (n > h->tape_bytes) {
            if (h->d_tape) { (void)hipFree(h->d_tape); h->d_tape = nullptr; h->tape_bytes = 0; }
            FB_CHECK_HIP(hipMalloc(&h->d_tape, n));
            h->tape_bytes = n;
        }
        FB_CHECK_HIP(hipMemcpy(h->d_tape, tape_host, n, hipMemcpyHostToDevice));
        h->p.tape = h->d_tape; h->p.tape_len = tape_len;
    }
    return FB_OK;
}

extern "C" int fb_env_render_full(fb_env_t h, int env_id, uint8_t *rgb, void *stream) {
    FB_REQUIRE(h && rgb && env_id >= 0 && env_id < h->p.n_envs, "fb_env_render_full: bad argument");
    hipLaunchKernelGGL(render_full_kernel, dim3(144), dim3(256), 0, fb_stream(stream), h->p, env_id, rgb);
    FB_LAUNCH_CHECK();
    return FB_OK;
}

extern "C" int fb_preprocess_rgb(fb_env_t h, const uint8_t *rgb, int n_frames, uint8_t *out, void *stream) {
    FB_REQUIRE(h && rgb && out && n_frames >= 1, "fb_preprocess_rgb: bad argument");
    const long long total = (long long)n_frames * 6400;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(preprocess_kernel, dim3(grid), dim3(256), 0, fb_stream(stream), h->d_const, rgb, n_frames, out);
    FB_LAUNCH_CHECK();
    return FB_OK;
}

extern "C" int fb_env_set_nib_buffer(fb_env_t h, uint8_t *nib_states) {
    FB_REQUIRE(h, "fb_env_set_nib_buffer: NULL handle");
    FB_CHECK_HIP(hipDeviceSynchronize());
    h->p.nib = nib_states;
    return FB_OK;
}

extern "C" int fb_env_set_stats_buffer(fb_env_t h, uint64_t *stats) {
    FB_REQUIRE(h, "fb_env_set_stats_buffer: NULL handle");
    FB_CHECK_HIP(hipDeviceSynchronize());
    h->p.stats = reinterpret_cast<unsigned long long *>(stats);
    return FB_OK;
}

extern "C" int fb_env_error_count(fb_env_t h, int64_t *count_host) {
    FB_REQUIRE(h && count_host, "fb_env_error_count: NULL argument");
    FB_CHECK_HIP(hipDeviceSynchronize());
    unsigned long long v = 0;
    FB_CHECK_HIP(hipMemcpy(&v, h->p.err_count, sizeof(v), hipMemcpyDeviceToHost));
    *count_host = (int64_t)v;
    return FB_OK;
}
