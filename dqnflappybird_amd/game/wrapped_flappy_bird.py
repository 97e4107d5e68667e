"""Drop-in for game/wrapped_flappy_bird.py of the reference: the same module surface
(constants :14-52, GameState :58-183, getRandomPipe :208-221), with frame_step running on the
MI355X (one env of the batched HIP kernel) instead of pygame.

Differences a caller can observe:
  * no SDL window, no 30 FPS sleep (FPSCLOCK.tick, :179): frame_step returns as fast as the GPU does;
  * PLAYER_INDEX_GEN's phase lives in the env state (one env per GameState) -- for a single
    GameState per process, which is all the reference ever creates, this is the same thing;
  * pipe gaps are drawn from Python's `random` exactly where the reference draws them, so
    `random.seed(s)` reproduces the reference's pipe sequence.
"""
import random

import numpy as np
import torch

from ..vec import VecGameState
from . import flappy_bird_utils

FPS = 30
SCREENWIDTH = 288
SCREENHEIGHT = 512

IMAGES, SOUNDS, HITMASKS = flappy_bird_utils.load()
PIPEGAPSIZE = 100
BASEY = SCREENHEIGHT * 0.79

PLAYER_WIDTH = IMAGES['player'][0].shape[1]
PLAYER_HEIGHT = IMAGES['player'][0].shape[0]
PIPE_WIDTH = IMAGES['pipe'][0].shape[1]
PIPE_HEIGHT = IMAGES['pipe'][0].shape[0]
BACKGROUND_WIDTH = IMAGES['background'].shape[1]


def getRandomPipe():
    """returns a randomly generated pipe (reference :208-221)"""
    gapYs = [20, 30, 40, 50, 60, 70, 80, 90]
    index = random.randint(0, len(gapYs) - 1)
    gapY = gapYs[index] + int(BASEY * 0.2)
    pipeX = SCREENWIDTH + 10
    return [{'x': pipeX, 'y': gapY - PIPE_HEIGHT}, {'x': pipeX, 'y': gapY + PIPEGAPSIZE}]


class GameState:
    def __init__(self):
        self._env = VecGameState(1, seed=0)
        self._tape = np.zeros((1, 3), np.int8)
        self._act = torch.zeros(1, dtype=torch.uint8, device=self._env.device)
        self._reinit()

    def _reinit(self):
        # GameState.__init__ draws two pipes from `random` (:67-68)
        self._tape[0, :2] = [random.randint(0, 7), random.randint(0, 7)]
        self._env.set_gap_tape(self._tape)
        st = self._env.get_state()
        phase = st[0, 13]
        self._env.reset()
        st = self._env.get_state()
        st[0, 13], st[0, 15] = phase, 0
        self._env.set_state(st)

    # the reference exposes these as attributes; they are read back from the device on demand
    def _state(self):
        return self._env.get_state()[0]

    score = property(lambda self: int(self._state()[5]))
    playery = property(lambda self: int(self._state()[0]))
    playerVelY = property(lambda self: int(self._state()[1]))
    playerIndex = property(lambda self: int(self._state()[2]))

    @property
    def upperPipes(self):
        st = self._state()
        return [{'x': int(st[7 + i]), 'y': 100 + 10 * int(st[10 + i]) - PIPE_HEIGHT} for i in range(st[6])]

    @property
    def lowerPipes(self):
        st = self._state()
        return [{'x': int(st[7 + i]), 'y': 100 + 10 * int(st[10 + i]) + PIPEGAPSIZE} for i in range(st[6])]

    def frame_step(self, input_actions):
        if sum(input_actions) != 1:
            raise ValueError('Multiple input actions!')
        # Up to three draws can be consumed inside one step (a spawn, then two on a crash):
        # offer three from `random`, then rewind to what was really used so that the stream
        # advances exactly like the reference's.
        rs = random.getstate()
        self._tape[0] = [random.randint(0, 7) for _ in range(3)]
        self._env.set_gap_tape(self._tape)
        st = self._env.get_state()
        st[0, 15] = 0
        self._env.set_state(st)
        self._act[0] = 1 if input_actions[1] == 1 else 0
        _, reward, terminal, score = self._env.frame_step(self._act, want_u8=True)
        used = int(self._env.get_state()[0, 15])
        random.setstate(rs)
        for _ in range(used):
            random.randint(0, 7)
        image_data = self._env.render_full(0).cpu().numpy()          # array3d layout [x][y][rgb]
        r = reward.item()
        reward_py = 0.1 if abs(r - 0.1) < 1e-6 else int(r)            # the reference returns 0.1 / 3 / -3
        return image_data, reward_py, bool(terminal.item()), int(score.item())

    def observation80(self):
        """The fused 80x80 {0,255} observation of the last step (what preprocess(image_data) gives)."""
        return self._env.frames[0].cpu().numpy()
