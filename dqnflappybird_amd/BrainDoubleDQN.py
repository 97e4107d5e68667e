"""Drop-in for BrainDoubleDQN.py.

In the reference `--model ddqn` does NOT run Double-DQN: the class defines `trainQNetwork`
(BrainDoubleDQN.py:37) while the base class calls `_trainQNetwork` (BrainDQN.py:75), so the
Nature-DQN update runs with the log directory '/double_dqn/'.  `faithful=True` (default) keeps
that; `faithful=False` runs the update the class was written for (argmax from the online net,
value from the target net, :51-61)."""
from .BrainDQNNature import BrainDQNNature

REPLACE_TARGET_ITER = 500


class BrainDoubleDQN(BrainDQNNature):
    DIR_NAME = "/double_dqn/"

    def __init__(self, actionNum, gameName, faithful=True, **kw):
        self.ALGO = "nature" if faithful else "double"
        super().__init__(actionNum, gameName, **kw)

    def trainQNetwork(self):
        """The reference's (dead) Double-DQN step, callable explicitly."""
        algo, self.ALGO = self.ALGO, "double"
        try:
            self._trainQNetwork()
        finally:
            self.ALGO = algo
