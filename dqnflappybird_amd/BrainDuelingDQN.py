"""Drop-in for the reference's stand-alone script BrainDuelingDQN.py (a second training recipe on the same kernels):

    python -m dqnflappybird_amd.BrainDuelingDQN [--steps N] [--quiet]

Same module surface -- constants, `createNetwork`, `trainNetwork`, `store_parameters`, `counter_add`,
`epsilon_select_action` -- and the same schedule, each with the reference line it follows:
  * dueling head Q = V + (A - mean A) on the usual trunk, eval net + target net          :47-139
  * loss = mean((q_target - q_eval)^2), Adam 1e-6, targets max_a Q_target(s')            :158-160,229-250
  * OBSERVE 10 000, epsilon 0.1 -> 1e-4 over 3 000 000 steps, decayed while t > OBSERVE    :27-31,189-190
  * the frame stack is NEWEST FIRST (np.append(x_t1, observation[:, :, :3]))               :216
  * target sync when t % 500 == 0 (session-local t, only while training)                   :225-227
  * checkpoint every 10 000 steps under global step t + step, `step` parsed back from the
    newest checkpoint's name; a copy of the checkpoint directory at every 1 000 000th
    global step -- including step 0 of a fresh run, like the reference                      :180-183,203-204,270-271,288-301
  * status line "TIMESTEP t / STATE s / EPSILON e / ACTION i / REWARD r / Q_MAX %e"         :282-283
  * `counter_add` (evaluation averages; its call is commented out in the reference too)    :304-324

Frame order.  The HBM replay ring and the kernels keep stacks newest LAST.  A network fed newest-first
stacks is the same function as one fed newest-last stacks with W_conv1's input-channel axis reversed, so this
recipe trains the usual way and reverses that axis whenever parameters cross its boundary (`get_params` /
`set_params` / checkpoints are in the REFERENCE's channel order).  `observation` (what the loop carries, what
`epsilon_select_action` takes) is newest first like the reference's.

The TF objects the reference passes around (placeholders, readouts) are replaced by one `Network` object; the
six-tuple `createNetwork` returns keeps its arity so that `trainNetwork(*createNetwork(), sess)` still reads the same.
"""
from __future__ import print_function

import os
import random
import shutil

import numpy as np

GAME = 'bird'  # the name of the game being played for log files
ACTIONS = 2  # number of valid actions
FRAME_PER_ACTION = 1  # number of frames per action
BATCH = 32  # size of minibatch

OBSERVE = 10000.  # timesteps to observe before training
EXPLORE = 3000000.  # frames over which to anneal epsilon
GAMMA = 0.99  # decay rate of past observations
FINAL_EPSILON = 0.0001  # final value of epsilon
INITIAL_EPSILON = 0.1  # starting value of epsilon
REPLAY_MEMORY = 50000  # number of previous transitions to remember
REPLACE_TARGET_ITER = 500  # number of steps when target net parameters update

SAVER_ITER = 10000  # number of steps when save checkpoint
COUNTERS_SIZE = 2  # the number of episodes to average for evaluation
AVERAGE_SIZE = 400  # the length of average_score to print a png

# Evaluation: store the average scores of the last episodes.
average_score = []

LOGS_PATH = "./logs_" + GAME + "/dueling_dqn/"
SAVE_PATH = "./saved_parameters/dueling_dqn/"
SAVE_BACK_PATH = "./saved_back/dueling_dqn/"

_W1 = 8 * 8 * 4 * 32           # W_conv1 [8][8][4][32] opens the flat parameter vector


def _flip_frames(flat):
    """reference channel order (newest first) <-> kernel order (newest last): reverse W_conv1's input-channel axis"""
    out = np.array(flat, np.float32, copy=True)
    out[:_W1] = out[:_W1].reshape(8, 8, 4, 32)[:, :, ::-1, :].reshape(-1)
    return out


class Network:
    """eval net + target net (reference :47-139) and the replay memory D (:148), on the compute backend."""

    def __init__(self, backend=None, seed=None, fc_width=512):
        if backend is None:
            from .backend import HipBackend
            backend = HipBackend()
        self.be = backend
        self.net = backend.make_net(ACTIONS, fc_width, "dueling", BATCH)
        seed = random.getrandbits(48) if seed is None else seed      # tf.truncated_normal is unseeded in the reference
        self.net.init_params(seed=seed, which=0)
        self.net.init_params(seed=seed + 1, which=1)                  # target_net: its own draw until the first sync
        self.D = backend.make_replay(REPLAY_MEMORY, False)
        self.stored = 0

    # parameters in the reference's layout (newest-first conv1 channels)
    def get_params(self, which=0):
        return _flip_frames(self.be.host(self.net.store_params(which)))

    def set_params(self, flat, which=0):
        self.net.load_params(_flip_frames(flat), which)

    def q_values(self, observation, which=0):
        """readout.eval(feed_dict={input: [observation]})[0]; observation uint8[80,80,4] newest first"""
        s = np.ascontiguousarray(np.asarray(observation, np.uint8)[None, :, :, ::-1])
        return self.be.host(self.net.forward(self.be.dev(s), which))[0]

    def __len__(self):
        return min(self.stored, REPLAY_MEMORY)


def createNetwork(backend=None, seed=None):
    """-> (eval_net_input, target_net_input, readout_eval, readout_target, h_fc1_eval, h_fc1_target): one Network
    object stands in for all six TF handles."""
    net = Network(backend, seed)
    return net, net, net, net, net, net


_net = None          # the reference's module globals readout_eval / eval_net_input, set by trainNetwork


def store_parameters(net, save_path=None, verbose=True):
    """reference :288-301: restore the newest checkpoint if there is one; -> (saver, step)."""
    save_path = SAVE_PATH if save_path is None else save_path
    marker = os.path.join(save_path, "checkpoint")
    step = 0
    if os.path.exists(marker):
        with open(marker) as f:
            name = f.read().strip()
        z = np.load(os.path.join(save_path, name))
        net.set_params(z["online"], 0)
        net.set_params(z["target"], 1)
        net.net.set_adam_state(net.be.dev(z["adam_m"]), net.be.dev(z["adam_v"]), z["beta_pows"])
        if verbose:
            print("Successfully loaded:", os.path.join(save_path, name))
        step = int(os.path.splitext(name)[0].split('-')[-1])
    elif verbose:
        print("Could not find old network weights")

    def saver(global_step):
        os.makedirs(save_path, exist_ok=True)
        name = GAME + '-dqn-' + str(global_step) + '.npz'
        m, v, pows = net.net.adam_state()
        np.savez(os.path.join(save_path, name), online=net.get_params(0), target=net.get_params(1),
                 adam_m=net.be.host(m), adam_v=net.be.host(v), beta_pows=np.asarray(pows, np.float32))
        with open(marker, "w") as f:
            f.write(name + "\n")
    return saver, step


def counter_add(counters, count, steps, logs_path=None):
    """reference :304-324, verbatim host logic"""
    counters.append(count)
    # calculate the mean score and clear the counter.
    if len(counters) >= COUNTERS_SIZE:
        average_score.append(np.mean(counters))
        # get a scores file and clear average_score.
        if steps >= 1000000:
            a = steps // 1000000
            max_size = AVERAGE_SIZE // (2 ** a)
        else:
            max_size = AVERAGE_SIZE
        if len(average_score) >= max_size:
            logs_path = LOGS_PATH if logs_path is None else logs_path
            os.makedirs(logs_path, exist_ok=True)
            with open(logs_path + str(steps) + "_average_score.txt", "w") as fo:
                fo.write(str(average_score))
            del average_score[:]
        del counters[:]


def epsilon_select_action(step, epsilon, observation, net=None):
    """reference :327-342: -> (a_t one-hot, action_q_value, action_index)"""
    net = _net if net is None else net
    action_q_value = net.q_values(observation)
    a_t = np.zeros([ACTIONS])
    action_index = 0
    if step % FRAME_PER_ACTION == 0:
        # epsilon-greedy to balance exploration and exploitation.
        if random.random() <= epsilon:
            action_index = random.randrange(ACTIONS)
            a_t[action_index] = 1
        else:
            action_index = np.argmax(action_q_value)
            a_t[action_index] = 1
    else:
        a_t[0] = 1  # do nothing
    return a_t, action_q_value, action_index


def trainNetwork(eval_net_input, target_net_input=None, readout_eval=None, readout_target=None, h_fc1_eval=None,
                 h_fc1_target=None, sess=None, game_state=None, preprocess=None, max_steps=None, verbose=True,
                 save_path=None, save_back_path=None):
    """reference :140-283.  `game_state` / `preprocess` default to the HIP drop-ins (game.wrapped_flappy_bird and
    FlappyBirdDQN.preprocess); `max_steps` ends the reference's endless loop (tests, smoke runs)."""
    global _net
    net = _net = eval_net_input
    be = net.be
    save_path = SAVE_PATH if save_path is None else save_path
    save_back_path = SAVE_BACK_PATH if save_back_path is None else save_back_path
    if game_state is None:
        from .game import wrapped_flappy_bird as game
        game_state = game.GameState()
    if preprocess is None:
        from .FlappyBirdDQN import preprocess
    counter = []                                             # noqa: F841 -- the reference keeps it; its only use is commented out

    # get the first state by doing nothing and preprocess the image to 80x80x4
    do_nothing = np.zeros(ACTIONS)
    do_nothing[0] = 1
    x_t, r_0, terminal, score_current = game_state.frame_step(do_nothing)
    x_t = np.asarray(preprocess(x_t)).reshape(80, 80)
    observation = np.stack((x_t, x_t, x_t, x_t), axis=2)     # observation 80x80x4
    net.D.reset(be.dev(np.ascontiguousarray(x_t.reshape(1, 80, 80), np.uint8)))

    # saving and loading networks, step determines the global step of a checkpoint
    saver, step = store_parameters(net, save_path, verbose)

    # start training
    epsilon = INITIAL_EPSILON
    t = 0
    while max_steps is None or t < max_steps:
        # choose an action epsilon greedily
        a_t, action_q_value, action_index = epsilon_select_action(t, epsilon, observation, net)

        # scale down epsilon
        if epsilon > FINAL_EPSILON and t > OBSERVE:
            epsilon -= (INITIAL_EPSILON - FINAL_EPSILON) / EXPLORE

        # run the selected action and observe next state and reward
        x_t1_colored, r_t, terminal, score_current = game_state.frame_step(a_t)

        if (step + t) % 1000000 == 0 and os.path.isdir(save_path):
            dst = save_back_path + str(step + t)
            if not os.path.exists(dst):
                shutil.copytree(save_path, dst)

        # preprocess the image; the stack is newest FIRST
        x_t1 = np.reshape(np.asarray(preprocess(x_t1_colored)), (80, 80, 1))
        observation_ = np.append(x_t1, observation[:, :, :3], axis=2)  # (80x80x4)

        # store the last 50000(REPLAY_MEMORY) transitions in D (frames once, in HBM)
        net.D.push(be.dev(np.ascontiguousarray(x_t1.reshape(1, 80, 80), np.uint8)),
                   be.dev(np.array([int(np.argmax(a_t))], np.uint8)), be.dev(np.array([r_t], np.float32)),
                   be.dev(np.array([1 if terminal else 0], np.uint8)))
        net.stored += 1

        # only train if done observing
        if t > OBSERVE:
            # check to replace target parameters
            if t % REPLACE_TARGET_ITER == 0:
                net.net.sync_target()
                if verbose:
                    print('\ntarget_params_replaced\n')
            # sample a minibatch to train on: random.sample(D, BATCH) draws deque positions
            idx = be.dev(np.array(random.sample(range(len(net)), BATCH), np.int64))
            s, a, r, s2, term = net.D.gather(idx)
            net.net.train_step("nature", s, a, r, s2, term, gamma=GAMMA, want_aux=False)     # mean loss, max_a Q_target(s')

        # update the old values
        observation = observation_
        t += 1

        # save progress every 10000 iterations
        if t % SAVER_ITER == 0:
            saver(t + step)

        # print info
        if verbose:
            if t <= OBSERVE:
                state = "observe"
            elif OBSERVE < t <= OBSERVE + EXPLORE:
                state = "explore"
            else:
                state = "train"
            print("TIMESTEP", t, "/ STATE", state, "/ EPSILON", epsilon, "/ ACTION", action_index,
                  "/ REWARD", r_t, "/ Q_MAX %e" % np.max(action_q_value))
    return t, epsilon, observation


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()
    handles = createNetwork()
    trainNetwork(*handles, sess=None, max_steps=args.steps, verbose=not args.quiet)


if __name__ == "__main__":
    main()
