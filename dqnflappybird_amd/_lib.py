"""ctypes binding of libfbdqn.so (include/fbdqn.h).

The HIP library is the product: there is no CPU fallback.  `lib()` raises if the
shared object is missing, and `require_gpu()` raises if no MI355X is visible, so a
GPU box can never silently run anything else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FB_LIB") or os.path.join(_HERE, "libfbdqn.so")      # FB_LIB: an ablation / tuning build (tools/)
ASSET_BLOB = os.path.join(_HERE, "assets", "sprites.bin")

FB_OK = 0
REPLAY_UNIFORM, REPLAY_PER = 0, 1
RNG_CPYTHON, RNG_PHILOX, RNG_NUMPY = 0, 1, 2
ARCH_PLAIN, ARCH_DUELING = 0, 1
NET_ONLINE, NET_TARGET = 0, 1
ALGO_DQN, ALGO_NATURE, ALGO_DOUBLE, ALGO_PER, ALGO_PG = 0, 1, 2, 3, 4
DTYPE_F32, DTYPE_BF16 = 0, 1
PER_EXACT, PER_FAST = 0, 1
NIB_PITCH, NIB_ROWS, NIB_STRIDE = 44, 84, 3712       # include/fbdqn.h FB_NIB_*

_vp, _i, _i64, _u64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t

# name -> argtypes; every symbol include/fbdqn.h declares (tests/test_capi_symbols.py checks the match)
SIGNATURES = {
    "fb_last_error": [],
    "fb_version": [],
    "fb_device_count": [],
    "fb_debug_abort_backtrace": [],
    "fb_env_create": [_i, _u64, C.c_uint32, _vp, _sz, _vp],
    "fb_env_destroy": [_vp],
    "fb_env_reset": [_vp, _vp],
    "fb_env_step": [_vp] * 8,
    "fb_env_observe": [_vp] * 4,
    "fb_env_get_state": [_vp, _vp],
    "fb_env_set_state": [_vp, _vp],
    "fb_env_set_gap_tape": [_vp, _vp, _i],
    "fb_env_render_full": [_vp, _i, _vp, _vp],
    "fb_env_error_count": [_vp, _vp],
    "fb_env_set_nib_buffer": [_vp, _vp],
    "fb_env_set_stats_buffer": [_vp, _vp],
    "fb_preprocess_rgb": [_vp, _vp, _i, _vp, _vp],
    "fb_replay_create": [_i64, _i, _i, _vp],
    "fb_replay_destroy": [_vp],
    "fb_replay_seed": [_vp, _i, _u64],
    "fb_replay_reset": [_vp, _vp, _vp, _vp],
    "fb_replay_push": [_vp] * 7,
    "fb_replay_push_sample": [_vp] * 6 + [_i, _vp, _vp],
    "fb_replay_current_state": [_vp, _vp, _vp],
    "fb_replay_sample": [_vp, _i, _vp, _vp, _vp, _vp],
    "fb_replay_gather": [_vp, _i] + [_vp] * 7,
    "fb_replay_profile_gather": [_vp, _i] + [_vp] * 6 + [_i, _vp],
    "fb_replay_update_priorities": [_vp, _i, _vp, _vp, _vp, _vp],
    "fb_replay_set_per_mode": [_vp, _i],
    "fb_replay_size": [_vp, _vp],
    "fb_replay_per_tree": [_vp] * 5,
    "fb_replay_state_bytes": [_vp, _vp],
    "fb_replay_save_state": [_vp, _vp, _sz],
    "fb_replay_load_state": [_vp, _vp, _sz],
    "fb_qnet_create": [_i, _i, _i, _i, _vp],
    "fb_qnet_destroy": [_vp],
    "fb_qnet_num_params": [_vp, _vp],
    "fb_qnet_init_params": [_vp, _i, _u64, _vp],
    "fb_qnet_load_params": [_vp, _i, _vp, _vp],
    "fb_qnet_store_params": [_vp, _i, _vp, _vp],
    "fb_qnet_get_adam_state": [_vp] * 4,
    "fb_qnet_set_adam_state": [_vp] * 4,
    "fb_qnet_set_hparams": [_vp, _f, _f, _f, _f],
    "fb_qnet_set_inference_dtype": [_vp, _i],
    "fb_qnet_overflow_count": [_vp, _i, _vp],
    "fb_qnet_split_stats": [_vp, _vp, _vp],
    "fb_vec_step_set_schedule": [_i],
    "fb_qnet_set_train_dtype": [_vp, _i],
    "fb_qnet_forward": [_vp, _i, _vp, _i, _vp, _vp],
    "fb_qnet_act": [_vp, _vp, _i, _f, _u64, _u64, _vp, _vp, _vp],
    "fb_qnet_act_nib": [_vp, _vp, _i, _f, _u64, _u64, _vp, _vp, _vp],
    "fb_qnet_train_step": [_vp, _i, _i] + [_vp] * 6 + [_d] + [_vp] * 5,
    "fb_qnet_apply_adam": [_vp, _vp, _vp],
    "fb_train_steps": [_vp, _vp, _i, _i, _i] + [_vp] * 7 + [_d, _vp],
    "fb_qnet_sync_target": [_vp, _vp],
    "fb_qnet_profile_kernel": [_vp, _i, _i, _i, _i] + [_vp] * 7,
    "fb_dist_probe": [C.c_char_p],
    "fb_dist_unique_id": [C.c_char_p, _vp],
    "fb_dist_create": [C.c_char_p, _i, _i, _vp],
    "fb_dist_destroy": [_vp],
    "fb_dist_set_overlap": [_vp, _i],
    "fb_vec_step_dp": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _u64, _i, _d, _i, _vp],
    "fb_dist_reduce_apply": [_vp, _vp, _vp, _i, _vp],
    "fb_dist_grad_event": [_vp],
    "fb_dist_all_reduce": [_vp, _vp, _i64, _vp],
    "fb_qnet_set_grad_event": [_vp, _vp],
    "fb_qnet_grad_split": [_vp],
    "fb_train_from_replay": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _vp],
    "fb_profile_ring_kernel": [_vp, _vp, _i, _i, _i, _i] + [_vp] * 6,
    "fb_qnet_kernel_name": [_i],
    "fb_vec_step": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _u64, _i, _d, _vp],
}


class StepBuffers(C.Structure):
    """fb_step_buffers (include/fbdqn.h)"""
    _fields_ = [(n, C.c_void_p) for n in ("nib", "actions", "frame_bits", "reward", "terminal", "score", "idx", "s", "s2", "a", "t",
                                           "r", "loss", "flat_grad", "isw", "isw32", "abs_err")]

_lib = None


class FbError(RuntimeError):
    pass


def lib():
    """Load libfbdqn.so; fail loudly when it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FbError(f"{LIB_PATH} is missing: build it with `make -C dqnflappybird_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
        # PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7, same as /opt/rocm's).
        # Import torch FIRST so that the process has ONE HIP runtime and libfbdqn.so binds to it;
        # loading /opt/rocm's copy first leaves torch without devices ("No HIP GPUs are available").
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError = stale library: rebuild it
            fn.argtypes = args
            fn.restype = C.c_char_p if name in ("fb_last_error", "fb_qnet_kernel_name") else C.c_int64 if name == "fb_qnet_grad_split" else C.c_void_p if name in ("fb_dist_create", "fb_dist_grad_event") else None if name == "fb_dist_destroy" else C.c_int
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != FB_OK:
        msg = lib().fb_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise FbError(f"{what} failed ({rc}): {msg}")


def require_gpu():
    n = lib().fb_device_count()
    if n <= 0:
        raise FbError("no HIP device visible: the MI355X path cannot run (there is no CPU fallback): "
                      + lib().fb_last_error().decode("utf-8", "replace"))
    return n


def sprite_blob():
    with open(ASSET_BLOB, "rb") as f:
        return f.read()


def ptr(t):
    """device/host pointer of a torch tensor / numpy array / None as c_void_p."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return t.ctypes.data_as(C.c_void_p)


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
