"""ctypes binding of libmzsearch.so (include/mzsearch.h).

There is no CPU or PyTorch fallback: if the HIP library is missing or a call
fails, this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MZS_LIB selects an alternative build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("MZS_LIB") or os.path.join(_HERE, "libmzsearch.so")

ENGINE_FUSED = 0
ENGINE_FUSED_GENERIC = 2
OPT_ADAM, OPT_SGD = 0, 1

TREE_VISIT, TREE_VALUE_SUM, TREE_PRIOR, TREE_REWARD, TREE_CHILD_BASE = 0, 1, 2, 3, 4
TREE_SIM_LEAF, TREE_MINMAX, TREE_HIDDEN, TREE_STATUS = 5, 6, 7, 8
TREE_ROOT_VISITS, TREE_ROOT_VALUE, TREE_ROOT_PRIOR, TREE_CONSUMED, TREE_PATH_LEN = 9, 10, 11, 12, 13


class MzsError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("n_actions", C.c_int32), ("state_dim", C.c_int32),
                ("hidden_dim", C.c_int32), ("n_mid_layers", C.c_int32), ("model_kind", C.c_int32),
                ("precision", C.c_int32)]


class ArenaConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("env", C.c_int32), ("n_slots", C.c_int32), ("n_rows", C.c_int32),
                ("max_moves", C.c_int32)]


ARENA_LENGTH, ARENA_OBS, ARENA_ACTION, ARENA_REWARD, ARENA_ROOT_VALUE, ARENA_POLICY = 0, 1, 2, 3, 4, 5
ARENA_ALIVE, ARENA_STATE, ARENA_STREAM_SEED, ARENA_SLOT_ROW, ARENA_QUEUE = 6, 7, 8, 9, 10
ENV_CARTPOLE = 0
MODEL_MLP, MODEL_LSTM, MODEL_VISION = 0, 1, 2
PRECISION_F32, PRECISION_BF16, PRECISION_BF16X3 = 0, 1, 2


class Config(C.Structure):
    _fields_ = [("dims", Dims), ("max_games", C.c_int32), ("max_simulations", C.c_int32),
                ("device", C.c_int32), ("engine", C.c_int32), ("discount", C.c_double),
                ("root_exploration_fraction", C.c_double), ("pbc_table", C.c_void_p)]


_SIGNATURES = {
    "mzs_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "mzs_destroy": (None, [C.c_void_p]),
    "mzs_weight_count": (C.c_size_t, [C.POINTER(Dims)]),
    "mzs_set_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mzs_search": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_int64, C.c_void_p]),
    "mzs_search_kernel_name": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_size_t]),
    "mzs_fill_streams": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_double,
                                   C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mzs_search_streams": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                     C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p]),
    "mzs_search_injected": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int64, C.c_void_p]),
    "mzs_selftest_reciprocals": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "mzs_set_result_sink": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mzs_results": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "mzs_tree_array": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "mzs_tree_stride": (C.c_int32, [C.c_void_p]),
    "mzs_hidden_stride": (C.c_int32, [C.c_void_p]),
    "mzs_initial_inference": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "mzs_recurrent_inference": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "mzs_prediction_inference": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "mzs_train_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mzs_train_grad": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p]),
    "mzs_train_grad_static": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "mzs_train_set_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mzs_train_set_momentum": (C.c_int, [C.c_void_p, C.c_double]),
    "mzs_train_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p]),
    "mzs_train_dropout_count": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "mzs_train_set_dropout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64]),
    "mzs_get_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mzs_arena_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mzs_arena_destroy": (None, [C.c_void_p]),
    "mzs_arena_obs_dim": (C.c_int32, [C.c_void_p]),
    "mzs_arena_n_actions": (C.c_int32, [C.c_void_p]),
    "mzs_arena_begin": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_void_p]),
    "mzs_arena_enqueue": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_void_p]),
    "mzs_arena_play": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                 C.c_double, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                 C.c_void_p, C.c_void_p]),
    "mzs_arena_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "mzs_arena_make_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mzs_arena_priority": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                     C.c_void_p, C.c_void_p]),
    "mzs_arena_priority_packed": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "mzs_arena_status": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "mzs_arena_array": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "mzs_replay_sample_positions": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "mzs_replay_update_value": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]),
    "mzs_last_error": (C.c_char_p, []),
    "mzs_version": (C.c_char_p, []),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


def load():
    """Load libmzsearch.so (built in-tree by `make -C muzero_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MzsError(f"{LIB_PATH} not found: build it with `make -C muzero_amd/csrc` "
                           "(or `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "there is no fallback path")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().mzs_last_error()
        raise MzsError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
