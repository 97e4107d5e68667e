#!/usr/bin/env python3
"""Pack the five Flappy-Bird sprites the hot path uses into one palettised blob.

Run in the build container only (it reads the PNG *data files* of the reference
checkout; no reference code is imported or executed):

    python tools/make_assets.py [/root/reference/assets/sprites] \
        [dqnflappybird_amd/assets/sprites.bin]

The blob is what `flappy_bird_utils.load()` of the reference produces
(game/flappy_bird_utils.py:16-100) reduced to what the fused kernel needs:
RGB per opaque pixel and opacity (= hitmask, alpha != 0,
game/flappy_bird_utils.py:103-124).  All sprite alphas are 0 or 255 and the
background is pure black (checked below), so blitting is "overwrite where
opaque" and the background needs no storage.

Layout (little endian), total 57 756 B:
    0     char[8]  magic "FBSPR001"
    8     u32      n_colours (palette entries in use, entry 0 = transparent)
    12    u32[256] palette, R | G<<8 | B<<16
    1036  u8[320][52]   pipe-green.png, [y][x] palette index (lower pipe; the
                        upper pipe is this image rotated by 180 degrees,
                        game/flappy_bird_utils.py:68-72)
    17676 u8[3][24][34] redbird-{up,mid,down}flap.png, [pose][y][x]
    20124 u8[112][336]  base.png, [y][x]
"""
import struct
import sys

import numpy as np
from PIL import Image

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets/sprites"
DST = sys.argv[2] if len(sys.argv) > 2 else "dqnflappybird_amd/assets/sprites.bin"


def rgba(name):
    return np.array(Image.open(f"{SRC}/{name}.png").convert("RGBA"))


def main():
    bg = rgba("background-black")
    assert bg.shape == (512, 288, 4) and not bg[..., :3].any(), "background must be pure black"
    pipe = rgba("pipe-green")
    birds = [rgba(f"redbird-{p}flap") for p in ("up", "mid", "down")]
    base = rgba("base")
    assert pipe.shape == (320, 52, 4) and base.shape == (112, 336, 4)
    assert all(b.shape == (24, 34, 4) for b in birds)

    palette = [0]  # index 0: transparent
    lut = {}

    def index(img):
        a = img[..., 3]
        assert set(np.unique(a)) <= {0, 255}, "alpha must be binary"
        out = np.zeros(img.shape[:2], np.uint8)
        for (y, x), al in np.ndenumerate(a):
            if al == 0:
                continue
            r, g, b = (int(v) for v in img[y, x, :3])
            key = r | g << 8 | b << 16
            if key not in lut:
                lut[key] = len(palette)
                palette.append(key)
            out[y, x] = lut[key]
        return out

    pipe_i = index(pipe)
    bird_i = np.stack([index(b) for b in birds])
    base_i = index(base)
    assert len(palette) <= 256
    assert (base_i != 0).all(), "base must be fully opaque"

    pal = np.zeros(256, np.uint32)
    pal[: len(palette)] = palette
    blob = b"FBSPR001" + struct.pack("<I", len(palette)) + pal.tobytes()
    blob += pipe_i.tobytes() + bird_i.tobytes() + base_i.tobytes()
    assert len(blob) == 8 + 4 + 1024 + 320 * 52 + 3 * 24 * 34 + 112 * 336
    with open(DST, "wb") as f:
        f.write(blob)
    print(f"{DST}: {len(blob)} B, {len(palette)} palette entries")


if __name__ == "__main__":
    main()
