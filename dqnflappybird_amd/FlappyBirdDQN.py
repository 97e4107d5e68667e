"""Drop-in for FlappyBirdDQN.py of the reference (driver :25-82): `--model` dispatch, `preprocess`,
and the getAction -> frame_step -> preprocess -> setPerception loop, on the MI355X.

    python -m dqnflappybird_amd.FlappyBirdDQN --model dqn [--steps N] [--quiet]
    python -m dqnflappybird_amd.FlappyBirdDQN --model dqn --vec 1024 --steps N      (vectorised loop)

`actorcritic` / `policygradient` are out of scope (broken in the reference).
"""
import argparse

import numpy as np

from . import _lib as L


def preprocess(observ, _env=[None]):
    """cv2.resize(observ, (80, 80)) -> COLOR_BGR2GRAY -> threshold(1, 255) (reference :31-34) as one
    HIP kernel; observ is the array3d frame u8[288,512,3]."""
    import ctypes as C
    import torch
    from .vec import VecGameState
    if _env[0] is None:
        _env[0] = VecGameState(1)
    rgb = torch.from_numpy(np.ascontiguousarray(observ, np.uint8)).cuda()
    out = torch.empty((80, 80), dtype=torch.uint8, device="cuda")
    L.check(L.lib().fb_preprocess_rgb(_env[0].h, L.ptr(rgb), 1, L.ptr(out), L.current_stream()), "fb_preprocess_rgb")
    return np.reshape(out.cpu().numpy(), (80, 80, 1))


def model_class(name):
    from .BrainDQN import BrainDQN
    from .BrainDQNNature import BrainDQNNature
    from .BrainDoubleDQN import BrainDoubleDQN
    from .BrainDuelingDQN_CC import BrainDuelingDQN
    from .BrainPrioritizedReplyDQN import BrainPrioritizedReplyDQN
    from .BrainActorCritic import BrainDQNActorCritic
    from .BrainPolicyGradient import BrainPolicyGradient
    table = {"dqn": BrainDQN, "ddqn": BrainDoubleDQN, "dqnnature": BrainDQNNature, "duelingdqn": BrainDuelingDQN,
             "prioritydqn": BrainPrioritizedReplyDQN, "actorcritic": BrainDQNActorCritic, "policygradient": BrainPolicyGradient}
    if name not in table:
        print("invalid model!")
        raise SystemExit(1)
    return table[name]


def playFlappyBird(model, steps=None, verbose=True):
    from .game import wrapped_flappy_bird as game
    brain = model_class(model)(2, 'bird', verbose=verbose)
    flappyBird = game.GameState()
    action0 = np.array([1, 0])
    observation0, reward0, terminal, curScore = flappyBird.frame_step(action0)
    observation0 = preprocess(observation0).reshape(80, 80)
    brain.setInitState(observation0)
    n = 0
    while steps is None or n < steps:
        action = brain.getAction()
        nextObserv, reward, terminal, curScore = flappyBird.frame_step(action)
        nextObserv = preprocess(nextObserv)
        brain.setPerception(nextObserv, action, reward, terminal, curScore)
        n += 1
    return brain


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--model")
    parser.add_argument("--steps", type=int, default=None)
    parser.add_argument("--quiet", action="store_true")
    parser.add_argument("--vec", type=int, default=0, help="run N vectorised envs (device-resident loop)")
    args = parser.parse_args()
    if args.vec:
        # one process per GPU (python -m torch.distributed.run --nproc-per-node N -m dqnflappybird_amd.FlappyBirdDQN ...):
        # --vec envs PER RANK, rank-local replay, one RCCL all-reduce of the flat gradient per train step
        from . import dist as fdist
        from .vecbrain import VecBrain
        rank, _, world = fdist.init()
        if args.model in ("actorcritic", "policygradient"):
            raise SystemExit("--vec runs the DQN family; the actor-critic / policy-gradient agents are single-env (as in the reference)")
        algo = {"dqn": "dqn", "ddqn": "nature", "dqnnature": "nature", "duelingdqn": "nature", "prioritydqn": "per"}[args.model]
        vb = VecBrain(args.vec, algo=algo, rank=rank, world=world)
        vb.run(args.steps or 1000, log_every=0 if (args.quiet or rank) else 100)
    else:
        playFlappyBird(args.model, args.steps, verbose=not args.quiet)


if __name__ == '__main__':
    main()
