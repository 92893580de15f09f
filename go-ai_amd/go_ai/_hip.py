"""ctypes binding of libgoai_hip.so (include/goai_hip.h).

This is the only place the package touches the native library.  PyTorch-ROCm is used for
device memory and streams only: tensors are handed over as raw `data_ptr()`s.  There is no
CPU fallback: if the library is missing the import of this module raises.
"""
import ctypes
import os

import torch  # noqa: F401  -- first, so that libgoai_hip binds to the HIP runtime torch already loaded

_PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("GOAI_HIP_LIB", os.path.join(_PKG_DIR, "libgoai_hip.so"))   # override: diagnostic builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C go-ai_amd/csrc` (or __graft_entry__.build()); "
        "this package has no CPU fallback")

lib = ctypes.CDLL(LIB_PATH)

c_void_p, c_int, c_int64, c_uint64, c_float, c_double = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, ctypes.c_double)

lib.goai_last_error.restype = ctypes.c_char_p
lib.goai_last_error.argtypes = []
lib.goai_version.restype = c_int
lib.goai_version.argtypes = []

_SIGNATURES = {
    "goai_ctx_create": [ctypes.POINTER(c_void_p), c_int, c_int],
    "goai_ctx_destroy": [c_void_p],
    "goai_ctx_set_stream": [c_void_p, c_void_p],
    "goai_ctx_sync": [c_void_p],
    "goai_transition": [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int],
    "goai_children": [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int],
    "goai_score": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "goai_liberties": [c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "goai_nn_init": [c_void_p, c_int, c_int, c_int, c_int],
    "goai_load_weights": [c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), c_int],
    "goai_nn_conv_algo": [c_void_p],
    "goai_nn_convs_per_launch": [c_void_p],
    "goai_nn_forward": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int],
    "goai_nn_forward_static": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int],
    "goai_mcts_init": [c_void_p, c_int, c_int, c_int],
    "goai_mcts_reset": [c_void_p, c_void_p],
    "goai_mcts_select": [c_void_p, c_void_p],
    "goai_mcts_backup": [c_void_p, c_void_p, c_void_p, c_void_p, c_int],
    "goai_mcts_expand": [c_void_p, c_void_p, c_void_p],
    "goai_mcts_backup_q": [c_void_p, c_void_p, c_void_p, c_int],
    "goai_mcts_child_values": [c_void_p, c_void_p],
    "goai_mcts_policy_value": [c_void_p, c_double, c_int, c_void_p],
    "goai_mcts_run": [c_void_p, c_int],
    "goai_mcts_steps": [c_void_p, c_int],
    "goai_mcts_steps_in_one_launch": [c_void_p],
    "goai_mcts_readout": [c_void_p, c_void_p, c_void_p, c_void_p],
    "goai_mcts_policy": [c_void_p, c_double, c_void_p],
    "goai_uniform_policy": [c_void_p, c_int, c_void_p],
    "goai_sample_actions": [c_void_p, c_void_p, c_uint64, c_uint64, c_int, c_void_p],
    "goai_mcts_advance": [c_void_p, c_void_p],
    "goai_mcts_set_active": [c_void_p, c_int],
    "goai_sample_actions_keyed": [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_int, c_void_p],
    "goai_mcts_node_capacity": [c_void_p],
    "goai_mcts_export_tree": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "goai_mcts_set_refill": [c_void_p, c_int],
    "goai_mcts_game_stats": [c_void_p, ctypes.POINTER(c_int64)],
    "goai_mcts_roots": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    "goai_mcts_counters": [c_void_p, ctypes.POINTER(c_int64)],
    "goai_train_init": [c_void_p, c_int, c_int, c_int],
    "goai_train_load": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int],
    "goai_train_export": [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int],
    "goai_train_reset_optimizer": [c_void_p],
    "goai_train_step": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_float, ctypes.c_float,
                        ctypes.POINTER(ctypes.c_float)],
    "goai_train_params": [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64)],
    "goai_train_sync_inference": [c_void_p],
    "goai_comm_unique_id": [c_void_p],
    "goai_comm_init": [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)],
    "goai_comm_destroy": [c_void_p, c_void_p],
    "goai_gather_stats": [c_void_p, c_void_p, ctypes.POINTER(c_double), c_int, ctypes.POINTER(c_int64)],
    "goai_profile_enable": [c_void_p, c_int],
    "goai_profile_read": [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_int64)],
}

for _name, _args in _SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = library older than this binding
    _fn.restype = c_int
    _fn.argtypes = _args


class GoaiError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise GoaiError(f"libgoai_hip error {rc}: {lib.goai_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "libgoai_hip takes contiguous device tensors"
    return c_void_p(t.data_ptr())
