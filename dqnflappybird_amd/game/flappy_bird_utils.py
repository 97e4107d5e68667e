"""Drop-in for game/flappy_bird_utils.py of the reference (load :16-100, getHitmask :103-124).

The reference decodes PNGs through pygame at import time; here the sprites come from the packed
blob the HIP kernels use (dqnflappybird_amd/assets/sprites.bin, made by tools/make_assets.py), so
IMAGES are numpy RGBA arrays [h][w][4] instead of pygame surfaces and SOUNDS stays empty (the
reference has its sounds commented out, :81-85).  HITMASKS keep the reference's mask[x][y] layout.
"""
import numpy as np

from .. import _lib as L

_PIPE, _BIRD, _BASE = (320, 52), (3, 24, 34), (112, 336)


def _unpack():
    blob = np.frombuffer(L.sprite_blob(), np.uint8)
    assert bytes(blob[:8]) == b"FBSPR001"
    pal = blob[12:12 + 1024].view(np.uint32)
    o = 12 + 1024
    pipe = blob[o:o + 320 * 52].reshape(_PIPE); o += 320 * 52
    bird = blob[o:o + 3 * 24 * 34].reshape(_BIRD); o += 3 * 24 * 34
    base = blob[o:o + 112 * 336].reshape(_BASE)
    return pal, pipe, bird, base


def _rgba(idx, pal):
    c = pal[idx]
    out = np.zeros(idx.shape + (4,), np.uint8)
    out[..., 0], out[..., 1], out[..., 2] = c & 255, (c >> 8) & 255, (c >> 16) & 255
    out[..., 3] = np.where(idx != 0, 255, 0)
    return out


def getHitmask(image):
    """mask[x][y] = alpha != 0 (reference :103-124); `image` is an RGBA array [h][w][4]."""
    return [[bool(image[y, x, 3]) for y in range(image.shape[0])] for x in range(image.shape[1])]


def load():
    pal, pipe, bird, base = _unpack()
    lower = _rgba(pipe, pal)
    IMAGES = {
        "numbers": (),                                   # only showScore used them; its call is commented out
        "base": _rgba(base, pal),
        "background": np.zeros((512, 288, 4), np.uint8), # background-black.png
        "player": tuple(_rgba(bird[i], pal) for i in range(3)),
        "pipe": (np.ascontiguousarray(lower[::-1, ::-1]), lower),   # [0] rotated by 180 degrees (:68-72)
    }
    IMAGES["background"][..., 3] = 255
    SOUNDS = {}
    HITMASKS = {
        "pipe": (getHitmask(IMAGES["pipe"][0]), getHitmask(IMAGES["pipe"][1])),
        "player": tuple(getHitmask(im) for im in IMAGES["player"]),
    }
    return IMAGES, SOUNDS, HITMASKS
