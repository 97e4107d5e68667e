"""Drop-in for BrainDQN.py of the reference: same class name, constructor and method surface
(BrainDQN.py:31-239), with the replay memory in HBM and the Q-network / Adam as HIP kernels.

Kept exactly as in the reference (cited lines are the reference's):
  * hyper-parameters :19-28; epsilon-greedy and its schedule :99-116 (epsilon keeps being
    decremented while > 0 and so ends a hair below 0, like dqn.log shows);
  * one shared `random` stream: random.random() / randrange per action, random.sample per train step;
  * training starts when onlineTimeStep > OBSERVE, one train step per env step :73-75;
  * loss = sum of squares (not mean) :162; target from the SAME network :205.
Not reproduced: TensorBoard graph dump, matplotlib plots and TF checkpoints.
"""
import random

import numpy as np

# Hyper Parameters (reference :19-28)
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


class BrainDQN:
    ALGO = "dqn"            # -> fb_qnet_train_step algo
    ARCH = "plain"
    PRIORITIZED = False
    DIR_NAME = "/dqn/"
    # module-level constants are looked up through the class so that subclasses in other modules
    # (which re-declare them, like the reference does) keep their own values
    OBSERVE, EXPLORE, BATCH_SIZE, GAMMA = OBSERVE, EXPLORE, BATCH_SIZE, GAMMA
    INITIAL_EPSILON, FINAL_EPSILON, REPLAY_MEMORY = INITIAL_EPSILON, FINAL_EPSILON, REPLAY_MEMORY

    def __init__(self, actionNum, gameName, backend=None, fc_width=512, verbose=True, seed=None):
        self.actionNum = actionNum
        self.gameName = gameName
        if backend is None:
            from .backend import HipBackend
            backend = HipBackend()
        self._be = backend
        self.verbose = verbose
        self.replayMemory = backend.make_replay(self.REPLAY_MEMORY, self.PRIORITIZED)
        self._n_stored = 0
        self.onlineTimeStep = 0
        self.gameTimes = 0
        self.timeStep = 0
        self.epsilon = self.INITIAL_EPSILON
        self._setDirName()
        self.lost_hist, self.q_target_list = [], []
        self.score_every_episode, self.time_steps_when_episode_end, self.reward_every_time_step = [], [], []
        self._fc_width = fc_width
        # tf.truncated_normal is unseeded in the reference: take a seed from `random` unless one is given
        self._seed = random.getrandbits(48) if seed is None else seed
        self._createQNetwork()

    def _setDirName(self):
        self.dir_name = self.DIR_NAME

    def _createQNetwork(self):
        self.net = self._be.make_net(self.actionNum, self._fc_width, self.ARCH, self.BATCH_SIZE)
        self.net.init_params(seed=self._seed, which=0)
        self.net.init_params(seed=self._seed + 1, which=1)      # target net: independent draw (BrainDQNNature.py:64-95)
        self.lost = None

    def __len__(self):
        return min(self._n_stored, self.REPLAY_MEMORY)

    # ------------------------------------------------------------------ reference surface
    def setInitState(self, observ):
        observ = np.ascontiguousarray(np.asarray(observ).reshape(80, 80), np.uint8)
        self.currentState = np.stack((observ, observ, observ, observ), axis=2)
        self.replayMemory.reset(self._be.dev(observ[None]))
        self._n_stored = 0

    def _q_values(self, state):
        return self._be.host(self.net.forward(self._be.dev(state[None])))[0]

    def getAction(self):
        QValue = self._q_values(self.currentState)
        action = np.zeros(self.actionNum)
        if self.timeStep % FRAME_PER_ACTION == 0:
            if random.random() <= self.epsilon:
                action_index = random.randrange(self.actionNum)
                action[action_index] = 1
            else:
                action_index = np.argmax(QValue)
                action[action_index] = 1
        else:
            action[0] = 1
        if self.epsilon > self.FINAL_EPSILON and self.onlineTimeStep > self.OBSERVE:
            self.epsilon -= (self.INITIAL_EPSILON - self.FINAL_EPSILON) / self.EXPLORE
        return action

    def _store(self, nextObserv, action, reward, terminal):
        frame = np.ascontiguousarray(np.asarray(nextObserv).reshape(1, 80, 80), np.uint8)
        self.replayMemory.push(self._be.dev(frame), self._be.dev(np.array([int(np.argmax(action))], np.uint8)),
                               self._be.dev(np.array([reward], np.float32)),
                               self._be.dev(np.array([1 if terminal else 0], np.uint8)))
        self._n_stored += 1

    def setPerception(self, nextObserv, action, reward, terminal, curScore):
        newState = np.append(self.currentState[:, :, 1:], np.asarray(nextObserv).reshape(80, 80, 1), axis=2)
        self._store(nextObserv, action, reward, terminal)
        if self.onlineTimeStep > self.OBSERVE:
            self._trainQNetwork()
        if self.verbose:
            if self.onlineTimeStep <= self.OBSERVE:
                state = "observe"
            elif self.onlineTimeStep <= self.OBSERVE + self.EXPLORE:
                state = "explore"
            else:
                state = "train"
            print("TIMESTEP", self.timeStep, "/ STATE", state, "/ ACTION", action[1], "/ EPSILON", self.epsilon,
                  "/ REWARD", reward, "/ SCORE", curScore)
        self.reward_every_time_step.append(reward)
        if terminal:
            self.gameTimes += 1
            if self.verbose:
                print("GAME_TIMES:" + str(self.gameTimes))
            self.score_every_episode.append(curScore)
            self.time_steps_when_episode_end.append(self.timeStep)
        self.currentState = newState
        self.timeStep += 1
        self.onlineTimeStep += 1

    # ------------------------------------------------------------------ training
    def _sample_indices(self):
        # random.sample(self.replayMemory, BATCH_SIZE) consumes `random` exactly like sampling the
        # index range does; the index IS the deque position (reference :197)
        return np.array(random.sample(range(len(self)), self.BATCH_SIZE), np.int64)

    def _pre_train(self):
        pass

    def _trainQNetwork(self):
        self._pre_train()
        idx = self._be.dev(self._sample_indices())
        s, a, r, s2, t = self.replayMemory.gather(idx)
        loss, _, y = self.net.train_step(self.ALGO, s, a, r, s2, t, gamma=self.GAMMA)
        self.lost = loss
        self._last_q_target = y
