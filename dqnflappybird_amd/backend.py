"""Compute backend of the single-env Brain classes: the HIP library through vec.QNet / vec.VecReplay.

The Brain classes keep the reference's host-side control flow (epsilon schedule, OBSERVE gate,
target-sync schedule, RNG consumption order) in Python and talk to this narrow interface, which is
what lets tests/ run that host logic on a CPU box against a CPU stand-in that lives in tests/.
"""
import numpy as np
import torch

from .vec import QNet, VecReplay


class HipBackend:
    name = "hip-gfx950"

    def make_net(self, actions, fc_width, arch, max_batch):
        return QNet(actions, fc_width, arch, max_batch=max_batch)

    def make_replay(self, capacity, prioritized):
        return VecReplay(capacity, 1, prioritized=prioritized)

    # ---- host <-> device plumbing (numpy at the reference's boundary)
    @staticmethod
    def dev(x, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(x))
        if dtype is not None:
            t = t.to(dtype)
        return t.cuda()

    @staticmethod
    def host(t):
        return t.cpu().numpy()

    @staticmethod
    def zeros(n):
        return torch.zeros(n, dtype=torch.float32, device="cuda")
