"""Bounds audit of the B = 256 Double-DQN train plan (the launch sequence of the one unexplained SIGABRT, round 2): every global
and LDS address expression of its kernels, re-derived here from csrc/fb_qnet.hip for every (workgroup, thread) of the real launch
geometry, against the sizes fb_qnet_create allocates (max_batch = 256, FC = 512).  Prints one row per (kernel, buffer): the
launch geometry, min / max element index touched, the allocation, and OK / OUT OF BOUNDS.  CPU only (numpy); `python tools/audit_bounds.py`.

The expressions are transcribed from the kernels (line references in each block) -- this is the "launch-parameter table" of
DESIGN.md, in executable form.  NS = 3 is the fp32-equivalent two-plane mode, NS = 1 the bf16 mode; both are checked because the
aborting sequence ran fp32 -> bf16 -> fp32 on the same buffers."""
import itertools
import sys

import numpy as np

B, FC, A, MAXTB, ZMAX = 256, 512, 2, 256, 64
S = 3 * B                                               # workspace rows (fb_qnet_create: S = 3 * max_batch)
OFF_W1, OFF_B1, OFF_W2, OFF_B2, OFF_W3, OFF_B3, OFF_WF1 = 0, 8192, 8224, 40992, 41056, 77920, 77984
CONV_PARAMS, CONV1_PARAMS = OFF_WF1, OFF_W2
NPARAMS = OFF_WF1 + 1600 * FC + FC + FC * A + A
WSP_W2, WSP_W3 = 0, 64 * 3 * 64
WSP_WF1 = WSP_W3 + 72 * 3 * 64
W3T = WSP_WF1 + (200 + 4) * 3 * FC
W2T = W3T + 72 * 3 * 64
WSP_TOTAL = W2T + 128 * 3 * 32                          # uint4 units

ALLOC = {                                              # elements of the unit named in each check
    "a1s(half)": S * 3200 * 6 // 2, "a3s(half)": (S * 1600 * 6 + 256) // 2, "p1(f32)": S * 3200, "amax(u8)": S * 3200,
    "h2(f32)": S * 1600, "h3(f32)": S * 1600, "hf(f32)": S * FC * 5, "q(f32)": S * 8, "dhf(f32)": B * FC, "dh3(f32)": B * 1600,
    "dh2(f32)": B * 1600, "dp1(f32)": B * 3200, "slabs(f32)": ZMAX * CONV_PARAMS, "slabs1(f32)": 2 * min(B, MAXTB) * CONV1_PARAMS,
    "wsp(uint4)": WSP_TOTAL, "grad(f32)": NPARAMS, "params(f32)": NPARAMS, "states(u8)": B * 25600, "w1s(uint4)": 3 * 8192 * 2 // 16,
    "gmax(f32)": FC // 16,
}
rows, bad = [], 0


def check(kernel, geom, buf, idx, width=1, lds=None):
    """idx: array of first-element indices of accesses `width` elements wide, in `buf` (or an LDS pool of `lds` elements)."""
    global bad
    idx = np.asarray(idx).ravel()
    lo, hi = int(idx.min()), int(idx.max()) + width - 1
    size = lds if lds is not None else ALLOC[buf]
    ok = lo >= 0 and hi < size
    bad += not ok
    rows.append((kernel, geom, buf if lds is None else f"LDS {buf}", lo, hi, size, "OK" if ok else "OUT OF BOUNDS"))


def grid(*dims):
    return np.meshgrid(*[np.arange(d) for d in dims], indexing="ij")


def drow(r, lane):
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)


# ---------------------------------------------------------------- conv1_sp_kernel<false>  (fb_qnet.hip conv1_sp_kernel)
# train plan, slices 0+1 in one pass (rows = 512, side.per = 256), slice 2 in a second pass (rows = 256, s_off = 512)
for s_off, rows_n, per in ((0, 2 * B, B), (2 * B, B, 0)):
    npool = rows_n * 100
    ntiles = (npool + 7) // 8
    nblk = min(256, (ntiles + 11) // 12)
    geom = f"grid {nblk} x 768 thr, {ntiles} tiles"
    tile, lane = grid(ntiles, 64)
    hl, j = lane >> 5, lane & 31
    pp, pos = j >> 2, j & 3
    P0 = tile * 8 + pp
    P = np.where(P0 < npool, P0, 0)
    b = P // 100
    rem = P - b * 100
    py, px = rem // 10, rem % 10
    oy, ox = 2 * py + (pos >> 1), 2 * px + (pos & 1)
    if per:
        b = b - (b // per) * per                          # state index inside its block (s / s2 are separate buffers of `per` states)
    for ky, kq in itertools.product(range(8), range(2)):
        iy = oy * 4 + ky - 2
        ix = ox * 4 - 2 + 4 * kq + 2 * hl
        ok = (P < npool) & (iy >= 0) & (iy < 80) & (ix >= 0) & (ix < 80)
        a = ((b * 80 + np.where(ok, iy, 0)) * 80) * 4 + np.where(ok, ix, 0) * 4
        check("conv1_sp<false>", geom, "states(u8)", a[ok], 8)
    Pp = tile * 8 + 2 * (j & 3) + hl
    live = Pp < npool
    so = (s_off * 100 + Pp) * 32 + 4 * (j >> 2)
    check("conv1_sp<false>", geom, "p1(f32)", so[live], 4)
    check("conv1_sp<false>", geom, "amax(u8)", so[live], 4)
    for plane in (0, 1):
        check("conv1_sp<false>", geom, "a1s(half)", so[live] + plane * S * 3200, 4)

# ---------------------------------------------------------------- conv23_sp_kernel<NS>
for NS in (3, 1):
    NPL, P0w = (2, 0) if NS == 3 else (1, 2)
    IN_P, C2_P = 2000, 1000
    ZOFF = NPL * IN_P
    RING = ZOFF + 16
    RSZ = 4 * NPL * 64
    XCH = RING + 6 * RSZ
    LDS = XCH + 2048
    for row0, rows_n in ((0, 2 * B), (2 * B, B)):
        nb = (rows_n + 4) // 5
        geom = f"NS={NS} grid {nb} x 512 thr"
        blk, tid = grid(nb, 512)
        lane, wave = tid & 63, tid >> 6
        w4, grp, hl, j = wave & 3, wave >> 2, lane >> 5, lane & 31
        s0 = blk * 5
        nloc = np.minimum(rows_n - s0, 5)
        for p, r in itertools.product(range(NPL), range(4)):        # input images
            q = tid + 512 * r
            ok = q < nloc * 400
            check(f"conv23_sp<{NS}>", geom, "a1s(half)", ((p * S * 3200 + (row0 + s0) * 3200) // 8 + q)[ok] * 8, 8)
            pix = q >> 2
            okl = q < IN_P
            check(f"conv23_sp<{NS}>", geom, "smem", (p * IN_P + pix * 4 + ((q + (pix >> 2)) & 3))[okl], 1, LDS)
        for i, q in itertools.product(range(17), range(2 if NPL == 2 else 1)):     # weight stream
            e = w4 + 4 * q
            k8, pl = e // NPL, e % NPL
            check(f"conv23_sp<{NS}>", geom, "wsp(uint4)", WSP_W2 + (2 * i + grp) * 768 + (k8 * 3 + P0w + pl) * 64 + lane)
        for slot in range(3):
            check(f"conv23_sp<{NS}>", geom, "smem", RING + grp * 3 * RSZ + slot * RSZ + w4 * 64 + lane + (256 if NPL == 2 else 0), 1, LDS)
        ml = w4 * 32 + j
        rowok = ml < 125
        bl = ml // 25
        for piece in range(8):                                          # conv2 output image + conv3 output planes
            check(f"conv23_sp<{NS}>", geom, "smem", ((NPL - 1) * C2_P + ml * 8 + (piece ^ ((ml >> 1) & 7)))[rowok], 1, LDS)
            live = rowok & (bl < nloc)
            for p in range(NPL):
                check(f"conv23_sp<{NS}>", geom, "a3s(half)", (p * S * 1600 + ((row0 + s0) * 25 + ml) * 64 + piece * 8 + 4 * hl)[live], 4)
            check(f"conv23_sp<{NS}>", geom, "h3(f32)", (((row0 + s0) * 25 + ml) * 64 + (piece // 4) * 32 + 8 * (piece & 3) + 4 * hl)[live], 4)
        check(f"conv23_sp<{NS}>", geom, "smem exchange (f32)", XCH * 4 + (wave * 16 + 15) * 64 + lane, 1, LDS * 4)     # exchange area (floats)

# ---------------------------------------------------------------- fc1_sp_kernel<NS>
for NS in (3, 1):
    NPL, P0w = (2, 0) if NS == 3 else (1, 2)
    for row0, M in ((0, 2 * B), (2 * B, B)):
        nblk = ((M + 127) // 128) * (FC // 64) * 4
        geom = f"NS={NS} grid {nblk} x 256 thr"
        blk, tid = grid(nblk, 256)
        lane, wave = tid & 63, tid >> 6
        xcd, w, nth = blk & 7, blk >> 3, FC >> 7
        ks, m0, n0 = xcd & 3, (w // nth) * 128, ((xcd >> 2) * nth + w % nth) * 64
        cbase = ks * 12 + np.minimum(ks, 2)
        for i in range(2):
            r = np.minimum(m0 + 64 * i + (tid >> 2), M - 1)
            for p, c in itertools.product(range(NPL), range(13)):       # a 12-chunk slice still LOADS a 13th
                check(f"fc1_sp<{NS}>", geom, "a3s(half)", p * S * 1600 + (row0 + r) * 1600 + cbase * 32 + (tid & 3) * 8 + c * 32, 8)
        for q in range(NPL):
            e = wave + 4 * q
            k8, pl = e // NPL, e % NPL
            for c in range(13):
                check(f"fc1_sp<{NS}>", geom, "wsp(uint4)", WSP_WF1 + ((cbase * 4 + k8) * 3 + P0w + pl) * FC + n0 + lane + c * 12 * FC)
        for ct, r in itertools.product(range(2), range(16)):
            mr = m0 + wave * 32 + drow(r, lane)
            live = mr < M
            check(f"fc1_sp<{NS}>", geom, "hf(f32)", (row0 * FC + (ks * S + mr) * FC + n0 + ct * 32 + (lane & 31))[live])

# ---------------------------------------------------------------- loss_head_kernel (publishes dhf + its per-workgroup maximum)
blk, tid = grid(FC // 16, 256)
jj = blk * 16 + (tid & 15)
for u in range(16):
    b = (tid >> 4) + 16 * u
    check("loss_head", f"grid {FC // 16} x 256 thr", "dhf(f32)", (b * FC + jj)[b < B])
    for ks in range(5):
        check("loss_head", f"grid {FC // 16} x 256 thr", "hf(f32)", ((min(ks, 3) if ks < 4 else 0) * S + np.where(b < B, b, tid >> 4)) * FC + jj)
check("loss_head", f"grid {FC // 16} x 256 thr", "gmax(f32)", blk)
check("loss_head", f"grid {FC // 16} x 256 thr", "q(f32)", (2 * B + np.minimum(tid, B - 1)) * A + A - 1)

# ---------------------------------------------------------------- fc1_bwd_big_kernel<NS, 4, 2>
ndx = ((B + 31) // 32) * 50
ntile = ndx + 50 * (FC // 32)
geom = f"grid {ntile} x 512 thr (KX 4, KW 2)"
tile, tid = grid(ndx, 512)
lane, wave = tid & 63, tid >> 6
hl, r = lane >> 5, lane & 31
mt, it = tile // 50, tile % 50
m = mt * 32 + r
for ks, half in itertools.product(range(4), range(2)):
    off = 8 * hl + (wave * 4) * 16 + ks * 16 + half * 4
    check("fc1_bwd_big dX", geom, "dhf(f32)", np.where(m < B, m, 0) * FC + off, 4)
    check("fc1_bwd_big dX", geom, "params(f32)", OFF_WF1 + (it * 32 + r) * FC + off, 4)
for q in range(2):
    rr = 2 * (wave & 7) + q
    mr = mt * 32 + drow(rr, lane)
    check("fc1_bwd_big dX", geom, "dh3(f32)", (mr * 1600 + it * 32 + r)[mr < B])
    check("fc1_bwd_big dX", geom, "red", ((wave * 16 + 15) * 64 + lane), 1, 8 * 16 * 64)
t, tid = grid(50 * (FC // 32), 512)
lane, wave = tid & 63, tid >> 6
hl, r = lane >> 5, lane & 31
it, nt = t // (FC // 32), t % (FC // 32)
for ks, jx in itertools.product(range(2), range(8)):
    b0 = 16 * (wave * 2 + ks) + 8 * hl + jx
    bc = np.where(b0 < B, b0, 0)
    check("fc1_bwd_big dW", geom, "h3(f32)", it * 32 + r + bc * 1600)
    check("fc1_bwd_big dW", geom, "dhf(f32)", nt * 32 + r + bc * FC)
for rr in range(16):
    check("fc1_bwd_big dW", geom, "grad(f32)", OFF_WF1 + (it * 32 + drow(rr, lane)) * FC + nt * 32 + r)

# ---------------------------------------------------------------- conv_bx_kernel<NS> (B per-sample chains; no dW tiles, no Adam span at B = 256 in export mode)
for NS in (3, 1):
    NPL, P0w = (2, 0) if NS == 3 else (1, 2)
    U4 = 2 * NPL * 200 + 16 + 2048 + 4
    geom = f"NS={NS} grid {B} x 512 thr"
    b, tid = grid(B, 512)
    lane, wave = tid & 63, tid >> 6
    hl, j = lane >> 5, lane & 31
    rowok = j < 25
    i = np.where(tid < 400, tid, 0)
    check(f"conv_bx<{NS}>", geom, "dh3(f32)", b * 1600 + 4 * i, 4)
    chA = (wave & 1) * 32 + 8 * (wave >> 1) + 4 * hl
    check(f"conv_bx<{NS}>", geom, "h2(f32)", (b * 25 + np.where(rowok, j, 0)) * 64 + chA, 4)
    check(f"conv_bx<{NS}>", geom, "dh2(f32)", ((b * 25 + j) * 64 + chA)[rowok], 4)
    cls, half = wave & 3, wave >> 2
    qy, qx = j // 5, j % 5
    pixB = ((cls >> 1) + 2 * qy) * 10 + (cls & 1) + 2 * qx
    ciB = 16 * half + 4 * hl
    for o8 in (0, 8):
        check(f"conv_bx<{NS}>", geom, "p1(f32)", (b * 100 + np.where(rowok, pixB, 0)) * 32 + ciB + o8, 4)
        check(f"conv_bx<{NS}>", geom, "dp1(f32)", ((b * 100 + pixB) * 32 + ciB + o8)[rowok], 4)
    for cc in range(17):
        for p in range(NPL):
            if cc < 9:
                ct, kq = wave & 1, wave >> 1
                check(f"conv_bx<{NS}>", geom, "wsp(uint4)", W3T + ((8 * cc + 2 * kq + hl) * 3 + P0w + p) * 64 + ct * 32 + j)
            else:
                c2 = cc - 9
                a2, b2, coh, ks = c2 >> 2, (c2 >> 1) & 1, c2 & 1, wave >> 2
                tap = ((((cls >> 1) + 1) & 1) + 2 * a2) * 4 + (((cls & 1) + 1) & 1) + 2 * b2
                check(f"conv_bx<{NS}>", geom, "wsp(uint4)", W2T + ((tap * 8 + coh * 4 + 2 * ks + hl) * 3 + P0w + p) * 32 + j)
    pix, q16 = i >> 4, i & 15
    check(f"conv_bx<{NS}>", geom, "pool", (NPL - 1) * 200 + pix * 8 + ((q16 >> 1) ^ ((pix >> 1) & 7)), 1, U4)
    check(f"conv_bx<{NS}>", geom, "pool red (f32)", (2 * NPL * 200 + 16) * 4 + (7 * 16 + 15) * 64 + lane, 1, U4 * 4)
    check(f"conv_bx<{NS}>", geom, "pool scale words (f32)", (2 * NPL * 200 + 16 + 2048) * 4 + wave, 1, U4 * 4)

# ---------------------------------------------------------------- conv_dwg_kernel<NS>  (conv_dwg_body, LAYER 3 then LAYER 2)
z3 = B // 16
n3, n2 = z3 * 4, z3 * 8
for NS in (3, 1):
    NPL = 2 if NS == 3 else 1
    PLH = 25 * 32 * 16
    U4 = 2 * NPL * PLH // 8 + 64 + 128 + 2048 + 4
    for LAYER, nb in ((3, n3), (2, n2)):
        geom = f"NS={NS} LAYER {LAYER}: {nb} of grid {n3 + n2} x 512 thr"
        blk, tid = grid(nb, 512)
        lane, wave = tid & 63, tid >> 6
        hl, r = lane >> 5, lane & 31
        cot = blk & 1
        mid = (blk >> 1) & 1 if LAYER == 3 else (blk >> 1) & 3
        g = blk >> 2 if LAYER == 3 else blk >> 3
        py, px = mid >> 1, mid & 1
        for q in range(13):
            w = wave + 8 * q
            wc = np.where(w < 100, w, 0)
            pos, bp = wc % 25, wc // 25 + 4 * hl
            s0 = g * 16 + 2 * bp
            if LAYER == 3:
                xo = (s0 * 25 + pos) * 64 + mid * 32 + r
                check(f"conv_dwg<{NS}>", geom, "h2(f32)", xo)
                check(f"conv_dwg<{NS}>", geom, "h2(f32)", xo + 1600)
            else:
                qy, qx = pos // 5, pos % 5
                xo = (s0 * 100 + (py + 2 * qy) * 10 + px + 2 * qx) * 32 + r
                check(f"conv_dwg<{NS}>", geom, "p1(f32)", xo)
                check(f"conv_dwg<{NS}>", geom, "p1(f32)", xo + 3200)
            yo = (s0 * 25 + pos) * 64 + cot * 32 + r
            ybuf = "dh3(f32)" if LAYER == 3 else "dh2(f32)"
            check(f"conv_dwg<{NS}>", geom, ybuf, yo)
            check(f"conv_dwg<{NS}>", geom, ybuf, yo + 1600)
            o = (pos * 32 + r) * 16 + 2 * bp                            # halves, 4-byte store
            for plane in range(NPL):
                check(f"conv_dwg<{NS}>", geom, "pool X", plane * PLH + o, 2, U4 * 8)
                check(f"conv_dwg<{NS}>", geom, "pool DY", NPL * PLH + plane * PLH + o, 2, U4 * 8)
        check(f"conv_dwg<{NS}>", geom, "pool zero page", 2 * NPL * PLH + 8 * hl, 8, U4 * 8)
        check(f"conv_dwg<{NS}>", geom, "pool frag", (NPL * 2 - 1) * PLH + (24 * 32 + r) * 16 + 8 * hl, 8, U4 * 8)
        part0 = (2 * NPL * PLH // 8 + 64) * 4                         # floats
        check(f"conv_dwg<{NS}>", geom, "pool part", part0 + tid, 1, U4 * 4)
        check(f"conv_dwg<{NS}>", geom, "pool red", part0 + 512 + (7 * 16 + 15) * 64 + lane, 1, U4 * 4)
        check(f"conv_dwg<{NS}>", geom, "pool scale words", part0 + 512 + 8192 + wave, 1, U4 * 4)
        base = g * CONV_PARAMS
        if LAYER == 3:
            for q in range(16):
                check(f"conv_dwg<{NS}>", geom, "slabs(f32)", base + OFF_W3 + (wave * 64 + mid * 32 + drow(q, lane)) * 64 + cot * 32 + r)
                check(f"conv_dwg<{NS}>", geom, "slabs(f32)", base + OFF_W3 + (8 * 64 + mid * 32 + drow(q, lane)) * 64 + cot * 32 + r)
            check(f"conv_dwg<{NS}>", geom, "slabs(f32)", base + OFF_B3 + cot * 32 + r)
        else:
            t4 = wave & 3
            ky, kx = ((py + 1) & 1) + 2 * (t4 >> 1), ((px + 1) & 1) + 2 * (t4 & 1)
            for q in range(16):
                check(f"conv_dwg<{NS}>", geom, "slabs(f32)", base + OFF_W2 + ((ky * 4 + kx) * 32 + drow(q, lane)) * 64 + cot * 32 + r)
            check(f"conv_dwg<{NS}>", geom, "slabs(f32)", base + OFF_B2 + cot * 32 + r)

# ---------------------------------------------------------------- conv_dw21_kernel<2, false> (conv1_dw2_body, two workgroups per sample -> sub-slabs) + slab_fold_kernel
NSP, SW = 2, 15
DW1_IMG_W, DW1_IMG_H = 104, 84
DW1_IMG = DW1_IMG_H * DW1_IMG_W * 4
U4 = DW1_IMG // 16 + SW * 2 * 64 + 64 + 4
geom = f"grid {2 * B} x 512 thr"
blk, tid = grid(2 * B, 512)
lane, wave = tid & 63, tid >> 6
hl, c = lane >> 5, lane & 31
b, part = blk // NSP, blk % NSP
s0 = part * SW
for u in range(2):
    ls = wave + 8 * u
    g = 2 * (s0 + ls) + hl
    oy, ox0 = g // 3, (g % 3) * 8
    for q in range(4):
        px = (ox0 >> 1) + q
        ok = (ls < SW) & (px < 10)
        po = (b * 100 + np.where(ok, (oy >> 1) * 10 + px, 0)) * 32 + c
        check("conv_dw21<2,false>", geom, "dp1(f32)", po)
        check("conv_dw21<2,false>", geom, "amax(u8)", po)
for k in range(4):
    ii = tid + 512 * k
    check("conv_dw21<2,false>", geom, "states(u8)", b * 25600 + np.where(ii < 1600, ii, 0) * 16, 16)
    row, col4 = ii // 20, ii % 20
    check("conv_dw21<2,false>", geom, "pool image", (((row + 2) * DW1_IMG_W + 4 + 4 * col4) * 4)[ii < 1600], 16, U4 * 16)
ky, kx, ci = wave, c >> 2, c & 3
for ls in range(SW):
    g = 2 * (s0 + ls) + hl
    oy, ox0 = g // 3, (g % 3) * 8
    tap = (ky * DW1_IMG_W + kx + 2) * 4 + ci + ((4 * oy) * DW1_IMG_W + 4 * ox0) * 4
    check("conv_dw21<2,false>", geom, "pool image", tap + 16 * 7, 1, DW1_IMG)
    check("conv_dw21<2,false>", geom, "pool dY fragments", DW1_IMG // 16 + (ls * 2 + 1) * 64 + lane, 1, U4)
check("conv_dw21<2,false>", geom, "pool bias sums", (DW1_IMG // 16 + SW * 2 * 64) * 4 + wave * 32 + c, 1, U4 * 4)
check("conv_dw21<2,false>", geom, "pool scale words", (DW1_IMG // 16 + SW * 2 * 64 + 64) * 4 + wave, 1, U4 * 4)
for r in range(16):
    check("conv_dw21<2,false>", geom, "slabs1(f32)", blk * CONV1_PARAMS + OFF_W1 + (ky * 32 + drow(r, lane)) * 32 + c)
check("conv_dw21<2,false>", geom, "slabs1(f32)", blk * CONV1_PARAMS + OFF_B1 + c)
fold = (2 * B + ZMAX - 1) // ZMAX
z1 = (2 * B + fold - 1) // fold
bx, s_, tid = grid((CONV1_PARAMS + 255) // 256, z1, 256)
idx = bx * 256 + tid
for q in range(fold):
    z = s_ * fold + q
    ok = (idx < CONV1_PARAMS) & (z < 2 * B)
    check("slab_fold", f"grid ({(CONV1_PARAMS + 255) // 256}, {z1}) x 256 thr, fold {fold}", "slabs1(f32)", (np.where(ok, z, 0) * CONV1_PARAMS + idx)[idx < CONV1_PARAMS])
check("slab_fold", f"grid ({(CONV1_PARAMS + 255) // 256}, {z1}) x 256 thr, fold {fold}", "slabs(f32)", (s_ * CONV_PARAMS + idx)[idx < CONV1_PARAMS])
# slab_reduce_kernel: z1 slabs of conv1, B / 16 of conv2 / conv3
bx, tid = grid((CONV_PARAMS + 255) // 256, 256)
idx = (bx * 256 + tid).ravel()
idx = idx[idx < CONV_PARAMS]
zz = np.where(idx < OFF_W2, z1, z3)
check("slab_reduce", f"grid {(CONV_PARAMS + 255) // 256} x 256 thr (z1 {z1}, z2 = z3 {z3})", "slabs(f32)", (zz - 1) * CONV_PARAMS + idx)
check("slab_reduce", f"grid {(CONV_PARAMS + 255) // 256} x 256 thr", "grad(f32)", idx)

# ---------------------------------------------------------------- report (one line per kernel / buffer: the widest range seen)
agg = {}
for k, g, buf, lo, hi, size, verdict in rows:
    key = (k, buf, size)
    if key not in agg:
        agg[key] = [g, lo, hi, size, verdict]
    else:
        a = agg[key]
        a[1], a[2] = min(a[1], lo), max(a[2], hi)
        if verdict != "OK":
            a[4] = verdict
print(f"{'kernel':22s} {'buffer':22s} {'min':>10s} {'max':>10s} {'size':>10s}  verdict   geometry")
for (k, buf, _), (g, lo, hi, size, verdict) in agg.items():
    print(f"{k:22s} {buf:22s} {lo:10d} {hi:10d} {size:10d}  {verdict:8s}  {g}")
print(f"\n{len(rows)} address expressions checked, {bad} out of bounds")
sys.exit(1 if bad else 0)
