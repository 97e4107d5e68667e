"""Drop-in for BrainDQN.py of the reference: same class name, constructor and method surface
(BrainDQN.py:31-239), with the replay memory in HBM and the Q-network / Adam as HIP kernels.

Kept exactly as in the reference (cited lines are the reference's):
  * hyper-parameters :19-28; epsilon-greedy and its schedule :99-116 (epsilon keeps being
    decremented while > 0 and so ends a hair below 0, like dqn.log shows);
  * one shared `random` stream: random.random() / randrange per action, random.sample per train step;
  * training starts when onlineTimeStep > OBSERVE, one train step per env step :73-75;
  * loss = sum of squares (not mean) :162; target from the SAME network :205.
Checkpoint / resume (reference :176-192,227-233): every 100 000 steps the parameters go to
"./saved_parameters<dir_name>bird-<timeStep>.npz" (flat fp32 vectors instead of a TF Saver-V2 bundle,
plus the Adam slots the reference also saves) and gameTimes / timeStep / epsilon are pickled, in the
reference's order, into "bird-saved-parameters.txt" -- that file is interchangeable with the reference's.
A new Brain restores both if they exist; like the reference, the replay memory and onlineTimeStep are
not saved, so a resumed run observes for OBSERVE steps again.
Logs (reference :36-56,224-226,233-235,242-324): loss per train step, the q_target batch per train step, score and
time step of every finished episode and the reward of every step are kept in the reference's five lists, appended
to the reference's five text files under "./logs_<game><dir_name>" (same names, same space-separated format) at
every checkpoint, and plotted into the reference's four PNGs at RECORD_STEP.  Keeping them costs one host read of
the loss and of the 32 targets per train step, which the single-env path (a PCIe round trip per step anyway)
does like the reference; `record_logs=False` switches it off.  Not reproduced: the TensorBoard graph dump.
"""
import os
import pickle
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


class _ScalarUnpickler(pickle.Unpickler):
    """bird-saved-parameters.txt holds three pickled scalars (reference :230-232).  A pickle can name any
    callable; this one may name none, so a saved_parameters directory from anywhere (the file format is the
    reference's) cannot run code here."""

    def find_class(self, module, name):
        raise pickle.UnpicklingError(f"saved parameters may only hold plain numbers, not {module}.{name}")


def _load_scalar(f):
    v = _ScalarUnpickler(f).load()
    if isinstance(v, bool) or not isinstance(v, (int, float)):
        raise pickle.UnpicklingError(f"saved parameters may only hold plain numbers, not {type(v).__name__}")
    return v


class BrainDQN:
    ALGO = "dqn"            # -> fb_qnet_train_step algo
    ARCH = "plain"
    PRIORITIZED = False
    DIR_NAME = "/dqn/"
    # module-level constants are looked up through the class so that subclasses in other modules
    # (which re-declare them, like the reference does) keep their own values
    OBSERVE, EXPLORE, BATCH_SIZE, GAMMA = OBSERVE, EXPLORE, BATCH_SIZE, GAMMA
    INITIAL_EPSILON, FINAL_EPSILON, REPLAY_MEMORY = INITIAL_EPSILON, FINAL_EPSILON, REPLAY_MEMORY

    def __init__(self, actionNum, gameName, backend=None, fc_width=512, verbose=True, seed=None,
                 save_root="./saved_parameters", logs_root="./logs_", record_logs=True, save_replay=False, checkpoint_format="npz"):
        # checkpoint_format: "npz" (flat fp32 vectors, the default) or "tf" -- the reference's own format, a TensorFlow Saver-V2 bundle
        # (bird-<timeStep>.index / .data-00000-of-00001 with the reference's variable names, tf_bundle.py) that `saver.restore` of
        # the reference reads.  Loading recognises both, whichever the `checkpoint` file names.
        self.checkpoint_format = checkpoint_format
        # save_replay: also checkpoint what the reference forgets (:176-192) -- the replay memory, onlineTimeStep, the current
        # frame stack and the `random` / np.random generator states -- so that a resumed run does not observe again and continues
        # the sampled-index stream bit for bit.  Default False = the reference's behaviour.
        self.save_replay = save_replay
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
        self.save_path = save_root + self.dir_name                      # reference :45
        self.saved_parameters_file_path = self.save_path + self.gameName + '-saved-parameters.txt'
        self.logs_path = logs_root + self.gameName + self.dir_name                 # reference :49
        self.record_logs = record_logs
        self.lost_hist_file_path = self.logs_path + 'lost_hist.txt'
        self.q_target_file_path = self.logs_path + 'q_targets.txt'
        self.score_every_episode_file_path = self.logs_path + 'score_every_episode.txt'
        self.time_steps_when_episode_end_file_path = self.logs_path + 'time_steps_when_episode_end.txt'
        self.reward_every_time_step_file_path = self.logs_path + 'reward_every_time_step.txt'
        self._createQNetwork()
        self._load_saved_parameters()

    def _setDirName(self):
        self.dir_name = self.DIR_NAME

    def _createQNetwork(self):
        self.net = self._be.make_net(self.actionNum, self._fc_width, self.ARCH, self.BATCH_SIZE)
        self.net.init_params(seed=self._seed, which=0)
        self.net.init_params(seed=self._seed + 1, which=1)      # target net: independent draw (BrainDQNNature.py:64-95)
        self.lost = None

    def __len__(self):
        return min(self._n_stored, self.REPLAY_MEMORY)

    # ------------------------------------------------------------------ checkpoint / resume
    SAVE_EVERY = 100000                                   # reference :227

    def _load_saved_parameters(self):
        """reference :176-192: restore the newest checkpoint of this algorithm's directory, if any."""
        marker = os.path.join(self.save_path, "checkpoint")
        if not os.path.exists(marker):
            if self.verbose:
                print("Could not find old network weights")
            return False
        with open(marker) as f:
            first = f.readline().strip()
        tf_style = first.startswith("model_checkpoint_path:")        # what tf.train.Saver writes (reference :178-181)
        ckpt = os.path.join(self.save_path, os.path.basename(first.split(":", 1)[1].strip().strip('"')) if tf_style else first)
        if tf_style or os.path.exists(ckpt + ".index"):
            from . import tf_bundle
            z = tf_bundle.load_flat(ckpt)                              # the reference's own checkpoint format
            if z["target"] is None:                                    # single-net graph (BrainDQN): the target copy is the net itself
                z["target"] = z["online"]
        else:
            z = np.load(ckpt)
        if z["online"].size != self.net.n_params:
            raise ValueError(f"{ckpt} holds {z['online'].size} parameters, this network has {self.net.n_params}")
        self.net.load_params(z["online"], 0)
        self.net.load_params(z["target"], 1)
        if z["adam_m"] is not None:
            self.net.set_adam_state(self._be.dev(z["adam_m"]), self._be.dev(z["adam_v"]), z["beta_pows"])
        if self.verbose:
            print("Successfully loaded:", ckpt)
        if os.path.exists(self.saved_parameters_file_path) and os.path.getsize(self.saved_parameters_file_path) > 0:
            with open(self.saved_parameters_file_path, 'rb') as f:      # three plain numbers, nothing executable
                self.gameTimes = _load_scalar(f)
                self.timeStep = _load_scalar(f)
                self.epsilon = _load_scalar(f)
        extra = (ckpt[:-4] if ckpt.endswith(".npz") else ckpt) + "-replay.npz"
        if self.save_replay and os.path.exists(extra) and hasattr(self.replayMemory, "load_state_blob"):
            e = np.load(extra)
            self.replayMemory.load_state_blob(e["replay"])
            self.onlineTimeStep, self._n_stored = int(e["scalars"][0]), int(e["scalars"][1])
            self.currentState = e["currentState"]
            st = e["py_random"]
            random.setstate((int(st[0]), tuple(int(x) for x in st[1:626]), None if st[626] < 0 else float(e["py_gauss"][0])))
            np.random.set_state(("MT19937", e["np_keys"], int(e["np_pos"][0]), int(e["np_pos"][1]), float(e["np_gauss"][0])))
            if self.verbose:
                print("Successfully loaded the replay memory:", extra)
        return True

    def save_checkpoint(self):
        """reference :227-233 (saver.save + three pickle.dump calls)."""
        os.makedirs(self.save_path, exist_ok=True)
        m, v, pows = self.net.adam_state()
        if self.checkpoint_format == "tf":
            # saver.save(sess, SAVE_PATH + gameName, global_step=timeStep): the bundle + the `checkpoint` state file (no .meta: there is
            # no graph).  One net (BrainDQN's names) unless the algorithm keeps a target net (the eval_net / target_net scopes)
            from . import tf_bundle
            name = f"{self.gameName}-{self.timeStep}"
            two = getattr(self, "ALGO", "dqn") != "dqn"
            tf_bundle.save_flat(os.path.join(self.save_path, name), self._be.host(self.net.store_params(0)),
                                self._be.host(self.net.store_params(1)) if two else None, self._be.host(m), self._be.host(v), pows)
            with open(os.path.join(self.save_path, "checkpoint"), "w") as f:
                f.write(f'model_checkpoint_path: "{name}"\nall_model_checkpoint_paths: "{name}"\n')
            name += ".npz"                                             # (the replay side file below keeps its name scheme)
        else:
            name = f"{self.gameName}-{self.timeStep}.npz"
            np.savez(os.path.join(self.save_path, name), online=self._be.host(self.net.store_params(0)),
                     target=self._be.host(self.net.store_params(1)), adam_m=self._be.host(m), adam_v=self._be.host(v),
                     beta_pows=np.asarray(pows, np.float32))
            with open(os.path.join(self.save_path, "checkpoint"), "w") as f:
                f.write(name + "\n")
        with open(self.saved_parameters_file_path, 'wb') as f:
            pickle.dump(self.gameTimes, f)
            pickle.dump(self.timeStep, f)
            pickle.dump(self.epsilon, f)
        if self.save_replay and hasattr(self.replayMemory, "state_blob"):
            ver, keys, gauss = random.getstate()
            nps = np.random.get_state()
            np.savez(os.path.join(self.save_path, name[:-4] + "-replay.npz"), replay=self.replayMemory.state_blob(),
                     scalars=np.array([self.onlineTimeStep, self._n_stored], np.int64), currentState=self.currentState,
                     py_random=np.array([ver, *keys, -1 if gauss is None else 1], np.int64), py_gauss=np.array([gauss or 0.0]),
                     np_keys=nps[1], np_pos=np.array([nps[2], nps[3]], np.int64), np_gauss=np.array([nps[4]]))

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
        self._after_train(loss, y)

    def _after_train(self, loss, y):
        """reference :224-235: keep loss / targets, checkpoint + flush the logs every SAVE_EVERY steps, plots at RECORD_STEP"""
        self.lost = loss
        self._last_q_target = y
        if self.record_logs:
            self.lost_hist.append(float(self._be.host(loss).reshape(-1)[0]))
            self.q_target_list.append([float(v) for v in self._be.host(y).reshape(-1)])
        if self.timeStep % self.SAVE_EVERY == 0:
            self.save_checkpoint()
            if self.record_logs:
                self._save_loss_score_timestep_reward_qtarget_to_file()
        if self.record_logs and self.timeStep in RECORD_STEP:
            self._record_by_pic()

    # ------------------------------------------------------------------ logs (reference :242-324)
    def _save_loss_score_timestep_reward_qtarget_to_file(self):
        """Append the five lists to their files ('value value ... ', one line per file, like the reference) and
        clear them."""
        os.makedirs(self.logs_path, exist_ok=True)
        for path, values in ((self.lost_hist_file_path, self.lost_hist),
                             (self.score_every_episode_file_path, self.score_every_episode),
                             (self.time_steps_when_episode_end_file_path, self.time_steps_when_episode_end),
                             (self.reward_every_time_step_file_path, self.reward_every_time_step),
                             (self.q_target_file_path, self.q_target_list)):
            with open(path, 'a') as f:
                for v in values:
                    f.write(str(v) + ' ')
            del values[:]

    def _get_loss_score_timestep_reward_qtarget_from_file(self):
        def numbers(path, brackets=False):
            if not os.path.exists(path):
                return []
            with open(path) as f:
                line = f.readline()
            if brackets:                                              # q_targets.txt holds str(list) entries
                line = line.replace('[', '').replace(']', '').replace(',', '')
            return [float(tok) for tok in line.split(' ') if tok]
        return (numbers(self.lost_hist_file_path), numbers(self.score_every_episode_file_path),
                numbers(self.time_steps_when_episode_end_file_path), numbers(self.reward_every_time_step_file_path),
                numbers(self.q_target_file_path, brackets=True))

    def _record_by_pic(self):
        """The reference's four plots (loss / score per episode / q_target / score over time steps)."""
        self._save_loss_score_timestep_reward_qtarget_to_file()
        loss, scores, ends, _, q_target = self._get_loss_score_timestep_reward_qtarget_from_file()
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        for ys, xs, ylabel, xlabel, name in ((loss, None, 'loss', 'time_step', '_lost_hist_total.png'),
                                             (scores, None, 'score', 'episode', '_scores_episode_total.png'),
                                             (q_target, None, 'q_target', 'BATCH * time_step', '_q_target_total.png'),
                                             (scores, ends, 'score', 'time_step', '_scores_time_step_total.png')):
            plt.figure()
            if xs is None:
                plt.plot(ys, '-')
            else:
                plt.plot(xs, ys, '-')
            plt.ylabel(ylabel)
            plt.xlabel(xlabel)
            plt.savefig(self.logs_path + str(self.timeStep) + name)
            plt.close()
