/*
 * oracle/fbo_env.c -- TEST INFRASTRUCTURE (see fbo.h).
 *
 * CPU restatement of the reference environment and observation preprocess:
 *   GameState.__init__   game/wrapped_flappy_bird.py:59-85
 *   GameState.frame_step game/wrapped_flappy_bird.py:87-183
 *   getRandomPipe        game/wrapped_flappy_bird.py:208-221
 *   checkCrash           game/wrapped_flappy_bird.py:244-275
 *   pixelCollision       game/wrapped_flappy_bird.py:278-300
 *   load / getHitmask    game/flappy_bird_utils.py:16-124
 *   preprocess           FlappyBirdDQN.py:31-34
 * It deliberately stays close to the reference's *order of operations*
 * (full 288x512x3 canvas, then resize, then gray, then threshold), unlike the
 * HIP kernel which never materialises the canvas.
 *
 * pygame's Rect.clip and SDL's blit are third-party (pygame 1.9.4, absent):
 * restated from pygame's rect.c rule and "overwrite where alpha != 0" (all
 * sprite alphas are 0/255, tools/make_assets.py checks it).  cv2.resize /
 * cvtColor / threshold are third-party (opencv-python, version unpinned,
 * absent): restated from OpenCV's resize.cpp (8-bit INTER_LINEAR: 11-bit
 * coefficients, FixedPtCast<int,uchar,22>) and color_rgb (14-bit BGR2GRAY).
 */
#include <math.h>
#include <string.h>
#include "fbo.h"

/* ------------------------------------------------------------------ assets */
static uint32_t g_pal[256];
static uint8_t g_pipe[FBO_PIPE_H][FBO_PIPE_W];               /* lower pipe, [y][x] */
static uint8_t g_bird[3][FBO_PLAYER_H][FBO_PLAYER_W];
static uint8_t g_base[FBO_BASE_H][FBO_BASE_W];
static int g_assets_ok = 0;

int fbo_assets_load(const uint8_t *blob, size_t n) {
    const size_t need = 8 + 4 + 1024 + sizeof(g_pipe) + sizeof(g_bird) + sizeof(g_base);
    if (n != need || memcmp(blob, "FBSPR001", 8) != 0) return -1;
    memcpy(g_pal, blob + 12, 1024);
    const uint8_t *p = blob + 12 + 1024;
    memcpy(g_pipe, p, sizeof(g_pipe)); p += sizeof(g_pipe);
    memcpy(g_bird, p, sizeof(g_bird)); p += sizeof(g_bird);
    memcpy(g_base, p, sizeof(g_base));
    g_assets_ok = 1;
    return 0;
}

/* pipe[0] of the reference is the image rotated by 180 deg (flappy_bird_utils.py:68-72) */
static inline uint8_t pipe_px(int upper, int x, int y) {
    return upper ? g_pipe[FBO_PIPE_H - 1 - y][FBO_PIPE_W - 1 - x] : g_pipe[y][x];
}

void fbo_hitmask_pipe(int upper, uint8_t *out) {          /* mask[x][y], flappy_bird_utils.py:113-118 */
    for (int x = 0; x < FBO_PIPE_W; x++)
        for (int y = 0; y < FBO_PIPE_H; y++) out[x * FBO_PIPE_H + y] = pipe_px(upper, x, y) != 0;
}
void fbo_hitmask_player(int pose, uint8_t *out) {
    for (int x = 0; x < FBO_PLAYER_W; x++)
        for (int y = 0; y < FBO_PLAYER_H; y++) out[x * FBO_PLAYER_H + y] = g_bird[pose][y][x] != 0;
}

/* ------------------------------------------------------------------ env */
static const int CYC[4] = {0, 1, 2, 1};                    /* wrapped_flappy_bird.py:52 */

static int draw_gap(fbo_env *e) {                          /* random.randint(0, 7), :212 */
    if (e->tape) {
        int v = (e->tape_pos < e->tape_len) ? e->tape[e->tape_pos] : 0;
        e->tape_pos++;
        return v & 7;
    }
    uint32_t o[4];
    fbo_philox4x32(e->seed_lo, e->seed_hi, e->env_id, e->rng_ctr++, 0u, 0u, o);
    return (int)(o[0] >> 29);
}

void fbo_env_reset(fbo_env *e) {                           /* __init__, :59-85 */
    e->score = e->player_index = e->loop_iter = 0;
    e->playery = (double)(int)((FBO_SCREENH - FBO_PLAYER_H) / 2);   /* 244 */
    e->basex = 0;
    int g1 = draw_gap(e), g2 = draw_gap(e);
    e->n_pipes = 2;
    e->pipe_x[0] = FBO_SCREENW;                            /* 288 */
    e->pipe_x[1] = FBO_SCREENW + FBO_SCREENW / 2;          /* 432.0 */
    e->pipe_gap[0] = g1; e->pipe_gap[1] = g2;
    e->pipe_x[2] = 0; e->pipe_gap[2] = 0;
    e->vely = 0;
}

void fbo_env_init(fbo_env *e, uint32_t seed_lo, uint32_t seed_hi, uint32_t env_id,
                  const int8_t *tape, int64_t tape_len, int cyc_pos) {
    memset(e, 0, sizeof(*e));
    e->seed_lo = seed_lo; e->seed_hi = seed_hi; e->env_id = env_id;
    e->tape = tape; e->tape_len = tape_len; e->cyc_pos = cyc_pos & 3;
    fbo_env_reset(e);
}

static inline int gap_y(int idx) { return 20 + 10 * idx + (int)(FBO_SCREENH * 0.79 * 0.2); }  /* :211-215 */

typedef struct { int x, y, w, h; } rect_t;

static rect_t rect_clip(rect_t A, rect_t B) {              /* pygame rect.c clip rule */
    rect_t r = {A.x, A.y, 0, 0};
    int x, y, w, h;
    if (A.x >= B.x && A.x < B.x + B.w) x = A.x;
    else if (B.x >= A.x && B.x < A.x + A.w) x = B.x;
    else return r;
    if (A.x + A.w > B.x && A.x + A.w <= B.x + B.w) w = A.x + A.w - x;
    else if (B.x + B.w > A.x && B.x + B.w <= A.x + A.w) w = B.x + B.w - x;
    else return r;
    if (A.y >= B.y && A.y < B.y + B.h) y = A.y;
    else if (B.y >= A.y && B.y < A.y + A.h) y = B.y;
    else return r;
    if (A.y + A.h > B.y && A.y + A.h <= B.y + B.h) h = A.y + A.h - y;
    else if (B.y + B.h > A.y && B.y + B.h <= A.y + A.h) h = B.y + B.h - y;
    else return r;
    r.x = x; r.y = y; r.w = w; r.h = h;
    return r;
}

static int pixel_collision(rect_t r1, rect_t r2, int pose, int upper) {   /* :278-300 */
    rect_t r = rect_clip(r1, r2);
    if (r.w == 0 || r.h == 0) return 0;
    int x1 = r.x - r1.x, y1 = r.y - r1.y, x2 = r.x - r2.x, y2 = r.y - r2.y;
    for (int x = 0; x < r.w; x++)
        for (int y = 0; y < r.h; y++)
            if (g_bird[pose][y1 + y][x1 + x] && pipe_px(upper, x2 + x, y2 + y)) return 1;
    return 0;
}

static int check_crash(const fbo_env *e) {                 /* :244-275 */
    const double BASEY = FBO_SCREENH * 0.79;
    if (e->playery + FBO_PLAYER_H >= BASEY - 1) return 1;
    rect_t pr = {FBO_PLAYERX, (int)e->playery, FBO_PLAYER_W, FBO_PLAYER_H};
    for (int i = 0; i < e->n_pipes; i++) {
        int gy = gap_y(e->pipe_gap[i]);
        rect_t ur = {e->pipe_x[i], gy - FBO_PIPE_H, FBO_PIPE_W, FBO_PIPE_H};
        rect_t lr = {e->pipe_x[i], gy + FBO_PIPEGAP, FBO_PIPE_W, FBO_PIPE_H};
        if (pixel_collision(pr, ur, e->player_index, 1) || pixel_collision(pr, lr, e->player_index, 0))
            return 1;
    }
    return 0;
}

int fbo_env_step(fbo_env *e, int action, float *reward, int *terminal, int *score_return) {
    const double BASEY = FBO_SCREENH * 0.79;               /* 404.48, :44 */
    if (action != 0 && action != 1) return -1;             /* sum(input_actions) != 1, :99-100 */
    float rew = 0.1f;
    int term = 0, flapped = 0;
    if (action == 1 && e->playery > -2 * FBO_PLAYER_H) { e->vely = -9; flapped = 1; }   /* :105-108 */
    if (e->vely < 10 && !flapped) e->vely += 1;            /* :110-111 */
    {                                                      /* :115-117, Python min(a, b) */
        double room = BASEY - e->playery - FBO_PLAYER_H;
        double d = (room < (double)e->vely) ? room : (double)e->vely;
        e->playery += d;
        if (e->playery < 0) e->playery = 0;
    }
    if ((e->loop_iter + 1) % 3 == 0) {                     /* :120-121 */
        e->player_index = CYC[e->cyc_pos];
        e->cyc_pos = (e->cyc_pos + 1) & 3;
    }
    e->loop_iter = (e->loop_iter + 1) % 30;                /* :122 */
    e->basex = -((-e->basex + 100) % 48);                  /* :123 */
    for (int i = 0; i < e->n_pipes; i++) e->pipe_x[i] -= 4;   /* :126-128 */
    if (0 < e->pipe_x[0] && e->pipe_x[0] < 5) {            /* :131-134 */
        int g = draw_gap(e);
        e->pipe_x[e->n_pipes] = FBO_SCREENW + 10;
        e->pipe_gap[e->n_pipes] = g;
        e->n_pipes++;
    }
    if (e->pipe_x[0] < -FBO_PIPE_W) {                      /* :137-139 */
        for (int i = 1; i < e->n_pipes; i++) { e->pipe_x[i - 1] = e->pipe_x[i]; e->pipe_gap[i - 1] = e->pipe_gap[i]; }
        e->n_pipes--;
    }
    {                                                      /* :142-148 */
        double mid = FBO_PLAYERX + FBO_PLAYER_W / 2.0;
        for (int i = 0; i < e->n_pipes; i++) {
            double pm = e->pipe_x[i] + FBO_PIPE_W / 2.0;
            if (pm <= mid && mid < pm + 4) { e->score += 1; rew = 3.0f; }
        }
    }
    int crash = check_crash(e);                            /* :151-153 */
    *score_return = e->score;                              /* :155 */
    if (crash) { term = 1; fbo_env_reset(e); rew = -3.0f; }   /* :157-162 */
    *reward = rew; *terminal = term;
    return 0;
}

void fbo_env_snapshot(const fbo_env *e, int32_t o[16]) {
    o[0] = (int)e->playery; o[1] = e->vely; o[2] = e->player_index; o[3] = e->loop_iter;
    o[4] = e->basex; o[5] = e->score; o[6] = e->n_pipes;
    for (int i = 0; i < 3; i++) {
        int live = i < e->n_pipes;
        int gy = gap_y(e->pipe_gap[i]);
        o[7 + i] = live ? e->pipe_x[i] : -9999;
        o[10 + i] = live ? gy - FBO_PIPE_H : 0;
        o[13 + i] = live ? gy + FBO_PIPEGAP : 0;
    }
}

/* ------------------------------------------------------------------ render */
static inline void put(uint8_t *rgb, int x, int y, uint8_t idx) {
    if (!idx || x < 0 || x >= FBO_SCREENW || y < 0 || y >= FBO_SCREENH) return;
    uint32_t c = g_pal[idx];
    uint8_t *p = rgb + ((size_t)x * FBO_SCREENH + y) * 3;
    p[0] = c & 255; p[1] = (c >> 8) & 255; p[2] = (c >> 16) & 255;
}

void fbo_env_render_full(const fbo_env *e, uint8_t *rgb) {  /* :165-177, array3d -> [x][y][rgb] */
    memset(rgb, 0, (size_t)FBO_SCREENW * FBO_SCREENH * 3);  /* background-black.png is all zero */
    for (int i = 0; i < e->n_pipes; i++) {
        int gy = gap_y(e->pipe_gap[i]);
        for (int y = 0; y < FBO_PIPE_H; y++)
            for (int x = 0; x < FBO_PIPE_W; x++) {
                put(rgb, e->pipe_x[i] + x, gy - FBO_PIPE_H + y, pipe_px(1, x, y));
            }
        for (int y = 0; y < FBO_PIPE_H; y++)
            for (int x = 0; x < FBO_PIPE_W; x++)
                put(rgb, e->pipe_x[i] + x, gy + FBO_PIPEGAP + y, pipe_px(0, x, y));
    }
    const int basey = (int)(FBO_SCREENH * 0.79);            /* blit truncates 404.48 -> 404 */
    for (int y = 0; y < FBO_BASE_H; y++)
        for (int x = 0; x < FBO_BASE_W; x++) put(rgb, e->basex + x, basey + y, g_base[y][x]);
    for (int y = 0; y < FBO_PLAYER_H; y++)
        for (int x = 0; x < FBO_PLAYER_W; x++)
            put(rgb, FBO_PLAYERX + x, (int)e->playery + y, g_bird[e->player_index][y][x]);
}

/* ------------------------------------------------------------------ preprocess */
/* cv2.resize(img[288][512][3], (80, 80)) INTER_LINEAR, 8-bit path of OpenCV's
 * resizeGeneric_/HResizeLinear/VResizeLinear: rows of the cv image are game x. */
static void linear_tab(int dst, int src, int *ofs, short *coef /*[dst][2]*/) {
    double inv = (double)dst / src, scale = 1.0 / inv;
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { s = 0; f = 0; }
        if (s >= src - 1) { s = src - 1; f = 0; }
        ofs[d] = s;
        coef[2 * d] = (short)lrintf((1.f - f) * 2048.f);
        coef[2 * d + 1] = (short)lrintf(f * 2048.f);
    }
}

void fbo_preprocess(const uint8_t *rgb, uint8_t *out) {    /* FlappyBirdDQN.py:31-34 */
    int xo[FBO_OBS], yo[FBO_OBS];
    short xa[2 * FBO_OBS], yb[2 * FBO_OBS];
    linear_tab(FBO_OBS, FBO_SCREENH, xo, xa);              /* cv columns = game y (512) */
    linear_tab(FBO_OBS, FBO_SCREENW, yo, yb);              /* cv rows    = game x (288) */
    for (int r = 0; r < FBO_OBS; r++) {
        const uint8_t *S0 = rgb + (size_t)yo[r] * FBO_SCREENH * 3;
        const uint8_t *S1 = rgb + (size_t)(yo[r] + 1 < FBO_SCREENW ? yo[r] + 1 : yo[r]) * FBO_SCREENH * 3;
        int b0 = yb[2 * r], b1 = yb[2 * r + 1];
        for (int c = 0; c < FBO_OBS; c++) {
            int sx = xo[c], sx1 = sx + 1 < FBO_SCREENH ? sx + 1 : sx;
            int a0 = xa[2 * c], a1 = xa[2 * c + 1];
            int ch[3];
            for (int k = 0; k < 3; k++) {
                int h0 = S0[sx * 3 + k] * a0 + S0[sx1 * 3 + k] * a1;      /* HResizeLinear */
                int h1 = S1[sx * 3 + k] * a0 + S1[sx1 * 3 + k] * a1;
                int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;   /* VResizeLinear */
                ch[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
            /* COLOR_BGR2GRAY applied to RGB-ordered data: channel 0 takes the B weight */
            int gray = (ch[0] * 1868 + ch[1] * 9617 + ch[2] * 4899 + (1 << 13)) >> 14;
            out[r * FBO_OBS + c] = gray > 1 ? 255 : 0;     /* cv2.threshold(.,1,255,THRESH_BINARY) */
        }
    }
}

void fbo_env_frame80(const fbo_env *e, uint8_t *out) {
    static __thread uint8_t canvas[FBO_SCREENW * FBO_SCREENH * 3];
    fbo_env_render_full(e, canvas);
    fbo_preprocess(canvas, out);
}
