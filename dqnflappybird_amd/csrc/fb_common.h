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

void fb_mt_init_genrand_host(FbMT *s, uint32_t seed);
void fb_mt_init_by_array_host(FbMT *s, const uint32_t *key, int n);
