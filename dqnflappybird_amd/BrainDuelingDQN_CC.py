"""Drop-in for BrainDuelingDQN_CC.py (what `--model duelingdqn` imports, FlappyBirdDQN.py:19).

Reference behaviour: the dueling head is defined in `createQNetwork` (BrainDuelingDQN_CC.py:37-147)
but the base class calls `_createQNetwork`, so the live graph is the plain Nature network; its own
setPerception (:219-245) calls the class's trainQNetwork (:171-215 = the Nature update) and prints
without the score.  `faithful=False` builds the real dueling head (V + (A - mean A),
BrainDuelingDQN.py:78-86) instead."""
from .BrainDQNNature import BrainDQNNature


class BrainDuelingDQN(BrainDQNNature):
    DIR_NAME = "/dueling_dqn/"

    def __init__(self, actionNum, gameName, faithful=True, **kw):
        self.ARCH = "plain" if faithful else "dueling"
        super().__init__(actionNum, gameName, **kw)

    def trainQNetwork(self):
        self._trainQNetwork()
