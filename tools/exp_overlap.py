"""Feasibility of running train(k) BESIDE act(k) + env(k) (both read the weights Adam(k - 1) left; only the replay push connects them):
timing only -- the two chains go to two HIP streams, optionally CU-masked (hipExtStreamCreateWithCUMask), with the cross-stream events the
real schedule would need.  Results are NOT checked here (the net's scratch buffers are shared between the chains).
    python tools/exp_overlap.py [n_cus_for_acting ...]      0 = two plain streams; seq = everything on one stream"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqnflappybird_amd import _lib as L
from dqnflappybird_amd.vec import QNet, VecGameState, VecReplay
lib = L.lib()
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
N, B, STEPS = int(os.environ.get("EXP_N", "1024")), int(os.environ.get("EXP_B", "32")), 400
ALGO = int(os.environ.get("EXP_ALGO", "0"))
env, rep, net = VecGameState(N, seed=0), VecReplay(1_000_000, N), QNet(max_batch=max(N, B))
rep.seed(0, "cpython"); net.init_params(0); net.sync_target()
nib = env.track_state(); env.observe(); rep.reset(env.frame_bits)
acts = torch.zeros(N, dtype=torch.uint8, device="cuda")
idx = torch.zeros(B, dtype=torch.int64, device="cuda")
a, r, t = torch.zeros(B, dtype=torch.uint8, device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(B, dtype=torch.uint8, device="cuda")
loss = torch.zeros(1, device="cuda")
P = L.ptr


def mk_stream(ncu, lo):
    """a stream on CUs [lo, lo + ncu) of the mask's bit order (ncu = 0: a plain non-blocking stream)"""
    s = C.c_void_p()
    if ncu == 0:
        assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
        return s
    bits = np.zeros(8, np.uint32)
    for i in range(lo, lo + ncu):
        bits[i // 32] |= np.uint32(1 << (i % 32))
    assert hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, bits.ctypes.data_as(C.c_void_p)) == 0
    return s


def mk_event():
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0           # hipEventDisableTiming
    return e


def act_env(st, step):
    L.check(lib.fb_qnet_act_nib(net.h, P(nib), N, C.c_float(0.03), 0, step, P(acts), None, st), "act")
    L.check(lib.fb_env_step(env.h, P(acts), None, P(env.frame_bits), P(env.reward), P(env.terminal), P(env.score), st), "env")
    L.check(lib.fb_replay_push(rep.h, None, P(env.frame_bits), P(acts), P(env.reward), P(env.terminal), st), "push")


def train(st):
    L.check(lib.fb_replay_sample(rep.h, B, None, P(idx), None, st), "sample")
    L.check(lib.fb_train_from_replay(rep.h, net.h, ALGO, B, P(idx), None, P(a), P(r), P(t), C.c_double(0.99), P(loss), None, None, st), "train")


def run(mode):
    torch.cuda.synchronize()
    if mode == "seq":
        sA = sT = mk_stream(0, 0)
    else:
        ncu = int(mode)
        sA, sT = mk_stream(ncu, 0), mk_stream(256 - ncu if ncu else 0, ncu)
    evA, evT = [mk_event() for _ in range(4)], [mk_event() for _ in range(4)]
    for k in range(60):
        act_env(sA, k)                                                  # fill the memory a little
    hip.hipStreamSynchronize(sA)
    out = []
    for rep_ in range(3):
        t0 = time.perf_counter()
        for k in range(STEPS):
            if sA is not sT:
                # act(k): needs Adam(k - 1) = train(k - 1) done;  train(k): needs push(k - 1) (the clean-minibatch case) and Adam(k - 1) (same stream)
                if k: hip.hipStreamWaitEvent(sA, evT[(k - 1) & 3], 0)
                act_env(sA, k); hip.hipEventRecord(evA[k & 3], sA)
                if k: hip.hipStreamWaitEvent(sT, evA[(k - 1) & 3], 0)
                train(sT); hip.hipEventRecord(evT[k & 3], sT)
            else:
                act_env(sA, k); train(sA)
        th = time.perf_counter() - t0
        hip.hipStreamSynchronize(sA); hip.hipStreamSynchronize(sT)
        dt = time.perf_counter() - t0
        out.append(f"{dt / STEPS * 1e6:.1f} (host {th / STEPS * 1e6:.1f})")
    print(f"mode {mode:>4}: us per step " + "  ".join(out), flush=True)


for m in (sys.argv[1:] or ["seq", "0", "224", "208", "192", "176", "160"]):
    run(m)
