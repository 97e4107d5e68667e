"""Drop-in for BrainDQNNature.py: BrainDQN + frozen target net, hard sync every 500 steps
(checked on timeStep, so the first sync of a fresh run is at step 1500), loss = mean
(reference BrainDQNNature.py:30,107-111,118-119,149-152)."""
from .BrainDQN import BrainDQN

FRAME_PER_ACTION = 1
BATCH_SIZE = 32
OBSERVE = 1000.
EXPLORE = 1000000.
GAMMA = 0.99
FINAL_EPSILON = 0
INITIAL_EPSILON = 0.03
REPLAY_MEMORY = 50000
SAVER_ITER = 10000
RECORD_STEP = (500000, 1000000, 1500000, 2000000, 2500000)
REPLACE_TARGET_ITER = 500


class BrainDQNNature(BrainDQN):
    ALGO = "nature"
    DIR_NAME = "/dqn_nature/"
    REPLACE_TARGET_ITER = REPLACE_TARGET_ITER

    def _pre_train(self):
        if self.timeStep % self.REPLACE_TARGET_ITER == 0:        # reference :151-152
            self.net.sync_target()
