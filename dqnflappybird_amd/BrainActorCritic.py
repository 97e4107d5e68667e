"""Drop-in for the reference's BrainActorCritic.py (online one-step actor-critic on two DQN trunks) -- SURVEY section 8(f) rank 4.

The reference's class cannot run: `trainQNetwork` (:181-191) starts with a leftover of the DQN classes that references names which do not
exist (`self.QValue`, `next_state_batch`, `BATCH_SIZE`, `minibatch`) and raises at the first step (its own actorcritic.log shows it).
What IS well defined there is the two graphs (:58-148) and the two `sess.run` calls (:194-211); this class executes those, on the HIP
path, with the following decisions:

  * the dead block :181-191 is dropped;
  * critic (:103-148): V(s) from a net with ONE output, td_error = r + GAMMA * V(s') - V(s), loss = td_error^2, batch of one, Adam 1e-6.
    The reference feeds `next_state_value_c: next_state` -- the next STATE's pixels where its placeholder [1, 1] wants the next state's
    VALUE (a shape error in TensorFlow).  Here V(s') is evaluated with the critic itself before the step, which is the standard one-step
    critic and what the placeholder's name says.  On the device this is FB_ALGO_DQN with one action: y = r + GAMMA * max_a Q(s') over a
    single a, sum loss, never terminal (the reference has no terminal case).
  * actor (:58-100): loss_a = mean(log pi(a|s) * td_error), MINIMISED as written -- which lowers the probability of actions with a
    positive TD error.  `faithful=True` (default) keeps that sign (FB_ALGO_PG with weight -td_error, since the PG loss is
    -log pi * weight); `faithful=False` uses +td_error (gradient ASCENT on log pi * td, the textbook actor).
  * `self.epsilon` (:221) is pickled but never set: set to 0.0 here.
"""
import os
import pickle

import numpy as np

GAMMA = 0.99
DIR_NAME = '/actor_critic/'
RECORD_STEP = (500000, 1000000, 1500000, 2000000, 2500000)


class BrainDQNActorCritic:
    def __init__(self, actionNum, gameName, backend=None, fc_width=512, verbose=True, seed=None, save_root="./saved_parameters",
                 logs_root="./logs_", record_logs=True, faithful=True):
        self.actionNum, self.gameName, self.faithful, self.verbose, self.record_logs = actionNum, gameName, faithful, verbose, record_logs
        if backend is None:
            from .backend import HipBackend
            backend = HipBackend()
        self._be = backend
        self.onlineTimeStep = self.gameTimes = self.timeStep = 0
        self.epsilon = 0.0
        self.save_path = save_root + DIR_NAME
        self.saved_parameters_file_path = self.save_path + self.gameName + '-saved-parameters.txt'
        self.logs_path = logs_root + self.gameName + DIR_NAME
        self.lost_hist_actor, self.lost_hist_critic, self.scores, self.q_target_critic_list = [], [], [], []
        self.score_every_episode, self.time_steps_when_episode_end, self.reward_every_time_step = [], [], []
        import random
        s0 = random.getrandbits(48) if seed is None else seed
        self.actor = backend.make_net(actionNum, fc_width, "plain", 1)      # scope 'Actor'  (:58-100)
        self.critic = backend.make_net(1, fc_width, "plain", 1)             # scope 'critic' (:103-148): W_fc2 is [512, 1]
        self.actor.init_params(seed=s0, which=0)
        self.critic.init_params(seed=s0 + 1, which=0)
        self.td_error = None

    def setInitState(self, observ):
        self.currentState = np.stack((observ, observ, observ, observ), axis=2)

    def _dev_state(self, state):
        return self._be.dev(np.ascontiguousarray(state[None], np.uint8))

    def action_prob(self, state):
        q = np.asarray(self._be.host(self.actor.forward(self._dev_state(state))), np.float32)[0]
        e = np.exp(q - q.max(), dtype=np.float32)
        return e / e.sum(dtype=np.float32)

    def getAction(self):
        action = np.zeros(self.actionNum)
        act_prob = self.action_prob(self.currentState)
        action_index = np.random.choice(range(act_prob.shape[0]), p=act_prob.ravel())      # reference :246
        action[action_index] = 1
        return action

    def trainQNetwork(self, action, reward, next_state):
        s, s2 = self._dev_state(self.currentState), self._dev_state(next_state)
        v_s = float(np.asarray(self._be.host(self.critic.forward(s))).reshape(-1)[0])          # state_value_c, before the update
        a0, r, t0 = self._be.dev(np.zeros(1, np.uint8)), self._be.dev(np.array([reward], np.float32)), self._be.dev(np.zeros(1, np.uint8))
        lost_c, _, y = self.critic.train_step("dqn", s, a0, r, s2, t0, gamma=GAMMA)              # td^2, one Adam step (:194-201)
        y = float(np.asarray(self._be.host(y)).reshape(-1)[0])
        td_error = y - v_s                                                                     # r + GAMMA * V(s') - V(s)
        act = self._be.dev(np.array([int(np.argmax(action))], np.uint8))
        w = self._be.dev(np.array([-td_error if self.faithful else td_error], np.float32))
        lost_a = self.actor.pg_step(s, act, w, 1)                                              # (:203-210)
        lost_a = float(np.asarray(self._be.host(lost_a)).reshape(-1)[0])     # = -log pi(a|s) * w: faithful (w = -td) -> log pi * td, the reference's own number
        self.td_error = td_error
        self.lost_hist_actor.append(lost_a)
        self.lost_hist_critic.append(float(np.asarray(self._be.host(lost_c)).reshape(-1)[0]))
        self.q_target_critic_list.append(td_error + v_s)
        if self.timeStep % 100000 == 0:
            self.save_checkpoint()

    def setPerception(self, nextObserv, action, reward, terminal, curScore):
        newState = np.append(self.currentState[:, :, 1:], nextObserv, axis=2)
        if self.verbose:
            print("TIMESTEP", self.timeStep, "/ ACTION", action[1], "/ REWARD", reward)
        self.trainQNetwork(action, reward, newState)
        if terminal:
            self.gameTimes += 1
            if self.verbose:
                print("GAME_TIMES:" + str(self.gameTimes))
            self.scores.append(curScore)
        self.currentState = newState
        self.timeStep += 1
        self.onlineTimeStep += 1

    def save_checkpoint(self):
        os.makedirs(self.save_path, exist_ok=True)
        name = f"{self.gameName}-{self.timeStep}.npz"
        out = {}
        for tag, net in (("actor", self.actor), ("critic", self.critic)):
            m, v, pows = net.adam_state()
            out.update({tag: self._be.host(net.store_params(0)), tag + "_m": self._be.host(m), tag + "_v": self._be.host(v),
                        tag + "_pows": np.asarray(pows, np.float32)})
        np.savez(os.path.join(self.save_path, name), **out)
        with open(os.path.join(self.save_path, "checkpoint"), "w") as f:
            f.write(name + "\n")
        with open(self.saved_parameters_file_path, 'wb') as f:
            pickle.dump(self.gameTimes, f)
            pickle.dump(self.timeStep, f)
            pickle.dump(self.epsilon, f)
