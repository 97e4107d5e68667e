"""VecBrain: the reference's training loop (FlappyBirdDQN.py:72-76 + Brain*.setPerception) for N
envs per GPU, entirely device resident: no tensor leaves HBM between getAction and the Adam update.

    step:  currentState (A1) -> getAction for N envs (A2) -> frame_step (E*, P1 fused) -> store (R1/R4)
           -> sample (R1/R4) -> gather (R2) -> _trainQNetwork (Q*) [-> all-reduce over ranks]

Host-side schedule follows the reference: training starts once onlineTimeStep > OBSERVE, epsilon
decays by (INITIAL - FINAL) / EXPLORE per step after that, Nature/Double sync the target net when
timeStep % 500 == 0, PER never does.
"""
import torch

from . import dist as fdist
from .vec import QNet, VecGameState, VecReplay, VecStep

MEAN_LOSS = {"dqn": False, "nature": True, "double": True, "per": True}


class VecBrain:
    def __init__(self, n_envs, algo="dqn", arch="plain", batch=32, capacity=1_000_000, fc_width=512, seed=0,
                 observe=1000, explore=1_000_000, initial_epsilon=0.03, final_epsilon=0.0, gamma=0.99,
                 replace_target_iter=500, sampler=None, rank=0, world=1):
        self.n, self.algo, self.batch, self.gamma = n_envs, algo, batch, gamma
        self.rank, self.world = rank, world
        self.observe, self.explore = observe, explore
        self.epsilon, self.initial_epsilon, self.final_epsilon = initial_epsilon, initial_epsilon, final_epsilon
        self.replace_target_iter = replace_target_iter
        self.seed = seed
        self.env = VecGameState(n_envs, seed=seed + 1000003 * rank)
        self.replay = VecReplay(capacity, n_envs, prioritized=(algo == "per"))
        self.replay.seed(seed + rank, sampler)
        self.net = QNet(2, fc_width, arch, max_batch=max(n_envs, batch))
        self.net.init_params(seed=seed, which=0)             # the same draw on every rank
        self.net.init_params(seed=seed + 1, which=1)
        self.grad = torch.zeros(self.net.n_params, dtype=torch.float32, device="cuda") if world > 1 else None
        self.timeStep = 0
        self.onlineTimeStep = 0
        self.nib = self.env.track_state()                    # currentState of every env, maintained by the env kernel
        self.env.observe()
        self.replay.reset(self.env.frame_bits)
        self.stats = self.env.track_stats()                  # [episodes, score sum, score max, pipes passed], kept by the env kernel
        self.last_loss = None
        # uniform replay: the whole step is one host call (fb_vec_step), with the head, random.sample and the Memory append
        # riding in the env launch; PER keeps the separate calls (its sample returns the importance weights)
        self.one_step = VecStep(self.env, self.replay, self.net, batch, algo, gamma, flat_grad=self.grad) if algo != "per" else None

    def train_step(self, idx=None):
        if self.algo in ("nature", "double") and self.timeStep % self.replace_target_iter == 0:
            self.net.sync_target()
        isw = None
        if idx is None:
            idx, isw = self.replay.sample(self.batch)
        s, a, r, s2, t = self.replay.gather(idx)
        loss, abs_err, _ = self.net.train_step(self.algo, s, a, r, s2, t, isw=isw, gamma=self.gamma, flat_grad=self.grad,
                                               want_aux=self.algo == "per")
        if self.grad is not None:
            fdist.allreduce_gradients(self.grad, MEAN_LOSS[self.algo])
            self.net.apply_adam(self.grad)
        if self.algo == "per":
            self.replay.update_priorities(idx, abs_err=abs_err)
        self.last_loss = loss

    def step(self):
        if self.one_step is not None:
            training = self.onlineTimeStep > self.observe
            if training and self.algo in ("nature", "double") and self.timeStep % self.replace_target_iter == 0:
                self.net.sync_target()                       # acting reads the online net only: same result as syncing before training
            self.one_step(self.epsilon, seed=self.seed + self.rank, step=self.timeStep, train=training)
            if self.epsilon > self.final_epsilon and self.onlineTimeStep > self.observe:
                self.epsilon -= (self.initial_epsilon - self.final_epsilon) / self.explore
            if training:
                if self.grad is not None:
                    fdist.allreduce_gradients(self.grad, MEAN_LOSS[self.algo])
                    self.net.apply_adam(self.grad)
                self.last_loss = self.one_step.loss
            self.timeStep += 1
            self.onlineTimeStep += 1
            return
        actions = self.net.act_nib(self.nib, self.epsilon, seed=self.seed + self.rank, step=self.timeStep)
        if self.epsilon > self.final_epsilon and self.onlineTimeStep > self.observe:
            self.epsilon -= (self.initial_epsilon - self.final_epsilon) / self.explore
        _, reward, terminal, _ = self.env.frame_step(actions, want_u8=False)
        training = self.onlineTimeStep > self.observe
        idx = None
        if training and self.algo != "per":                  # store + random.sample in one launch (same indices)
            idx = self.replay.push_sample(self.env.frame_bits, actions, reward, terminal, self.batch)
        else:
            self.replay.push(self.env.frame_bits, actions, reward, terminal)
        if training:
            self.train_step(idx)
        self.timeStep += 1
        self.onlineTimeStep += 1

    def run(self, steps, log_every=100):
        for i in range(steps):
            self.step()
            if log_every and (i + 1) % log_every == 0:
                loss = self.last_loss.item() if self.last_loss is not None else float("nan")
                ep, ssum, smax, pipes = self.stats.tolist()      # the only host sync of the loop, once per log line
                print(f"TIMESTEP {self.timeStep} / ENVS {self.n} / EPSILON {self.epsilon:.6f} / GAME_TIMES {ep} / "
                      f"MEAN_SCORE {ssum / max(ep, 1):.3f} / MAX_SCORE {smax} / PIPES {pipes} / LOSS {loss:.6g}", flush=True)
