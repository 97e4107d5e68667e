"""Drop-in for the reference's BrainPolicyGradient.py (REINFORCE on the DQN trunk) -- SURVEY section 8(f) rank 4.

Same surface as the reference class (`BrainPolicyGradient(actionNum, gameName)`, `setInitState`, `getAction`, `setPerception`,
the ep_* lists, timeStep / onlineTimeStep / gameTimes, the four log streams); the network, its softmax cross-entropy loss and Adam run
on the HIP path (FB_ALGO_PG: vec.QNet.pg_step).  The reference's class has bugs; what was decided about each:

  * :129 vs :136 -- trainQNetwork computes the discounted, normalised returns and then feeds the RAW rewards as `tf_rewards`.
    `faithful=True` (default) does what the reference's code does (raw rewards); `faithful=False` feeds the normalised returns, which
    is evidently what was meant (and what the method's name says).
  * :147 -- `self.epsilon` is pickled but never set: the reference would raise at its first save.  Here epsilon = 0.0 exists, so the
    save works (there is no epsilon-greedy in this agent: actions are sampled from the softmax).
  * :206 -- `_record_by_pic` calls a method that does not exist; here it calls the one that does
    (`_save_loss_score_timestep_reward_to_file`).
  * :168 -- the transition stores `newState` (the state AFTER the action) with the action and reward; kept as is.
An episode is trained on as ONE batch (`:131-137`): episodes longer than 128 states go to the device in chunks that export their
share of the mean's gradient; one Adam step per episode, as in the reference.
"""
import os
import pickle

import numpy as np

GAMMA = 0.99                                        # decay rate of past observations (reference :17)
SAVE_PATH = "./saved_parameters/policy_gradient/"
RECORD_STEP = (500000, 1000000, 1500000, 2000000, 2500000)
DIR_NAME = '/policy_gradient/'
CHUNK = 128                                         # FB_ALGO_PG takes <= 128 states per call


class BrainPolicyGradient:
    def __init__(self, actionNum, gameName, backend=None, fc_width=512, verbose=True, seed=None, save_root="./saved_parameters",
                 logs_root="./logs_", record_logs=True, faithful=True):
        self.actionNum, self.gameName, self.faithful, self.verbose, self.record_logs = actionNum, gameName, faithful, verbose, record_logs
        if backend is None:
            from .backend import HipBackend
            backend = HipBackend()
        self._be = backend
        self.ep_states, self.ep_acts, self.ep_rewards = [], [], []
        self.onlineTimeStep = self.gameTimes = self.timeStep = 0
        self.epsilon = 0.0                          # (pickled by the reference's save, never set there: see the module docstring)
        self.save_path = save_root + DIR_NAME
        self.saved_parameters_file_path = self.save_path + self.gameName + '-saved-parameters.txt'
        self.logs_path = logs_root + self.gameName + DIR_NAME
        self.lost_hist, self.score_every_episode, self.time_steps_when_episode_end, self.reward_every_time_step = [], [], [], []
        self.lost_hist_file_path = self.logs_path + 'lost_hist.txt'
        self.score_every_episode_file_path = self.logs_path + 'score_every_episode.txt'
        self.time_steps_when_episode_end_file_path = self.logs_path + 'time_steps_when_episode_end.txt'
        self.reward_every_time_step_file_path = self.logs_path + 'reward_every_time_step.txt'
        import random
        self.net = backend.make_net(actionNum, fc_width, "plain", CHUNK)
        self.net.init_params(seed=random.getrandbits(48) if seed is None else seed, which=0)
        self.lost = None
        self.counters = []
        self._load_saved_parameters()

    # ------------------------------------------------------------------ checkpoint (reference :104-124, :141-150)
    def _load_saved_parameters(self):
        marker = os.path.join(self.save_path, "checkpoint")
        if not os.path.exists(marker):
            if self.verbose:
                print("Could not find old network weights")
            return False
        from .BrainDQN import _load_scalar
        with open(marker) as f:
            z = np.load(os.path.join(self.save_path, f.read().strip()))
        self.net.load_params(z["online"], 0)
        self.net.set_adam_state(self._be.dev(z["adam_m"]), self._be.dev(z["adam_v"]), z["beta_pows"])
        if os.path.exists(self.saved_parameters_file_path) and os.path.getsize(self.saved_parameters_file_path) > 0:
            with open(self.saved_parameters_file_path, 'rb') as f:
                self.gameTimes, self.timeStep, self.epsilon = _load_scalar(f), _load_scalar(f), _load_scalar(f)
        return True

    def save_checkpoint(self):
        os.makedirs(self.save_path, exist_ok=True)
        name = f"{self.gameName}-{self.timeStep}.npz"
        m, v, pows = self.net.adam_state()
        np.savez(os.path.join(self.save_path, name), online=self._be.host(self.net.store_params(0)), adam_m=self._be.host(m),
                 adam_v=self._be.host(v), beta_pows=np.asarray(pows, np.float32))
        with open(os.path.join(self.save_path, "checkpoint"), "w") as f:
            f.write(name + "\n")
        with open(self.saved_parameters_file_path, 'wb') as f:
            pickle.dump(self.gameTimes, f)
            pickle.dump(self.timeStep, f)
            pickle.dump(self.epsilon, f)

    # ------------------------------------------------------------------ reference surface
    def setInitState(self, observ):
        self.currentState = np.stack((observ, observ, observ, observ), axis=2)

    def act_prob(self, state):
        """self.act_prob.eval: softmax of the logits, in float32 like the graph (reference :91-94)"""
        q = np.asarray(self._be.host(self.net.forward(self._be.dev(np.ascontiguousarray(state[None], np.uint8)))), np.float32)[0]
        e = np.exp(q - q.max(), dtype=np.float32)
        return e / e.sum(dtype=np.float32)

    def getAction(self):
        action = np.zeros(self.actionNum)
        act_prob = self.act_prob(self.currentState)
        action_index = np.random.choice(range(act_prob.shape[0]), p=act_prob.ravel())      # reference :186: numpy's global stream
        action[action_index] = 1
        return action

    def store_transition_in_episode(self, state, action, reward):
        self.ep_states.append(state)
        self.ep_acts.append(action)
        self.ep_rewards.append(reward)

    def _discount_and_norm_rewards(self):
        """reference :200-211"""
        discounted_ep_rs = np.zeros_like(self.ep_rewards, dtype=np.float64)      # (the reference's zeros_like of an all-int reward list would be an int array and raise two lines down)
        running_add = 0
        for t in reversed(range(0, len(self.ep_rewards))):
            running_add = running_add * GAMMA + self.ep_rewards[t]
            discounted_ep_rs[t] = running_add
        discounted_ep_rs -= np.mean(discounted_ep_rs)
        discounted_ep_rs /= np.std(discounted_ep_rs)
        return discounted_ep_rs

    def trainQNetwork(self):
        discounted_ep_rewards_norm = self._discount_and_norm_rewards()
        weights = np.asarray(self.ep_rewards if self.faithful else discounted_ep_rewards_norm, np.float32)      # reference :136 feeds ep_rewards
        states = np.ascontiguousarray(np.stack(self.ep_states), np.uint8)
        acts = np.argmax(np.asarray(self.ep_acts), axis=1).astype(np.uint8)             # labels = the one-hot actions (:97,99)
        n = len(weights)
        if n <= CHUNK:
            loss = self.net.pg_step(self._be.dev(states), self._be.dev(acts), self._be.dev(weights), n)
            self.lost = float(np.asarray(self._be.host(loss)).reshape(-1)[0])
        else:                                       # one batch = one episode (:131-137): chunks export their share of the mean's gradient
            total, g, loss_sum = self._be.zeros(self.net.n_params), self._be.zeros(self.net.n_params), 0.0
            for lo in range(0, n, CHUNK):
                hi = min(n, lo + CHUNK)
                loss = self.net.pg_step(self._be.dev(states[lo:hi]), self._be.dev(acts[lo:hi]), self._be.dev(weights[lo:hi]), n, flat_grad=g)
                total += g
                loss_sum += float(np.asarray(self._be.host(loss)).reshape(-1)[0])
            self.net.apply_adam(total)
            self.lost = loss_sum
        self.lost_hist.append(self.lost)
        self.ep_states, self.ep_acts, self.ep_rewards = [], [], []
        if self.timeStep % 100000 == 0:             # reference :141
            self.save_checkpoint()
            if self.record_logs:
                self._save_loss_score_timestep_reward_to_file()
        if self.timeStep in RECORD_STEP and self.record_logs:
            self._record_by_pic()

    def setPerception(self, nextObserv, action, reward, terminal, curScore):
        newState = np.append(self.currentState[:, :, 1:], nextObserv, axis=2)
        self.store_transition_in_episode(newState, action, reward)
        if self.verbose:
            print("TIMESTEP", self.timeStep, "/ ACTION", action[1], "/ REWARD", reward)
        self.reward_every_time_step.append(reward)
        if terminal:
            self.trainQNetwork()
            self.gameTimes += 1
            self.score_every_episode.append(curScore)
            self.time_steps_when_episode_end.append(self.timeStep)
            if self.verbose:
                print("GAME_TIMES:" + str(self.gameTimes))
        self.currentState = newState
        self.timeStep += 1
        self.onlineTimeStep += 1

    # ------------------------------------------------------------------ logs (reference :214-270)
    def _save_loss_score_timestep_reward_to_file(self):
        os.makedirs(self.logs_path, exist_ok=True)
        for path, data in ((self.lost_hist_file_path, self.lost_hist), (self.score_every_episode_file_path, self.score_every_episode),
                           (self.time_steps_when_episode_end_file_path, self.time_steps_when_episode_end),
                           (self.reward_every_time_step_file_path, self.reward_every_time_step)):
            with open(path, 'a') as f:
                for x in data:
                    f.write(str(x) + ' ')
            del data[:]

    def _get_loss_score_timestep_reward_from_file(self):
        def numbers(path):
            with open(path) as f:
                return [float(x) for x in f.readline().split(" ")[0:-1]]
        return (numbers(self.lost_hist_file_path), numbers(self.score_every_episode_file_path),
                numbers(self.time_steps_when_episode_end_file_path), numbers(self.reward_every_time_step_file_path))

    def _record_by_pic(self):
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        self._save_loss_score_timestep_reward_to_file()       # (the reference names a method here that does not exist, :206)
        loss, scores, when, _ = self._get_loss_score_timestep_reward_from_file()
        for ys, xs, yl, xl, name in ((loss, None, 'loss', 'time_step', "_lost_hist_total.png"), (scores, None, 'score', 'episode', "_scores_episode_total.png"),
                                     (scores, when, 'score', 'time_step', "_scores_time_step_total.png")):
            plt.figure()
            plt.plot(ys, '-') if xs is None else plt.plot(xs, ys, '-')
            plt.ylabel(yl)
            plt.xlabel(xl)
            plt.savefig(self.logs_path + str(self.timeStep) + name)
            plt.close()
