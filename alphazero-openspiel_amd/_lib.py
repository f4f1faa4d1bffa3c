"""ctypes binding of libaz_engine.so — the C ABI declared in include/az_engine.h.

There is NO fallback: if the HIP library is missing, importing the engine fails loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AZ_ENGINE_LIB") or os.path.join(HERE, "libaz_engine.so")  # env override: profiling builds

GAME_CONNECT_FOUR, GAME_BREAKTHROUGH = 0, 1
BACKUPS = {"on-policy": 0, "soft-Z": 1, "A0C": 2, "off-policy": 3}
RNG_PHILOX, RNG_INJECTED = 0, 1
FAULTS = {1: "POOL_EXHAUSTED", 2: "PLY_OVERFLOW", 4: "NO_VISITS", 8: "BAD_PRIOR", 16: "ILLEGAL_ACTION", 32: "VISIT_RANGE"}
SELECT_PUCT, SELECT_UCT = 0, 1
ACTION_NONE, ACTION_SEARCH_AGAIN = -1, -2
ARENA_AGENTS = {None: 0, "zero": 1, "net": 2}
OPPONENTS = {None: 0, "random": 1, "uct": 2, "external": 3}


class AzConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("game", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("n_slots", C.c_int32), ("n_playouts", C.c_int32), ("use_dirichlet", C.c_int32),
                ("keep_search_tree", C.c_int32), ("backup", C.c_int32), ("rng_mode", C.c_int32),
                ("max_sims_per_tick", C.c_int32), ("device", C.c_int32), ("manual_moves", C.c_int32),
                ("chain_window_us", C.c_int32), ("nodes_per_slot", C.c_int64), ("max_games", C.c_int64),
                ("c_puct", C.c_double), ("dirichlet_ratio", C.c_double), ("dirichlet_alpha", C.c_double),
                ("temperature", C.c_double), ("seed", C.c_uint64),
                ("arena_agent", C.c_int32), ("arena_opponent", C.c_int32), ("opponent_sims", C.c_int32),
                ("arena_flip", C.c_int32), ("opponent_uct_c", C.c_double),
                ("select_rule", C.c_int32), ("arena_probabilistic", C.c_int32),
                ("num_probabilistic_actions", C.c_int32), ("spare_pools", C.c_int32)]


class AzSizes(C.Structure):
    _fields_ = [("num_actions", C.c_int32), ("obs_planes", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("max_children", C.c_int32), ("max_plies", C.c_int32), ("n_slots", C.c_int32),
                ("spare_pools", C.c_int32), ("nodes_per_slot", C.c_int64), ("max_games", C.c_int64),
                ("device_bytes", C.c_int64)]


class AzProgress(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("games_started", "games_done", "moves", "sims", "evals", "terminal_hits",
                                          "sum_depth", "sum_children", "nodes_allocated", "compactions",
                                          "slots_waiting", "slots_idle", "slots_search_done")] + \
               [("error_flags", C.c_uint32), ("reserved", C.c_uint32)]


class AzExampleView(C.Structure):
    _fields_ = [("n_games", C.c_int64), ("max_plies", C.c_int32), ("max_children", C.c_int32),
                ("game_len", C.POINTER(C.c_int32)), ("game_ret0", C.POINTER(C.c_float)),
                ("states", C.POINTER(C.c_uint64)), ("move", C.POINTER(C.c_uint16)),
                ("n_children", C.POINTER(C.c_uint8)), ("child_action", C.POINTER(C.c_uint16)),
                ("child_visits", C.POINTER(C.c_uint32)), ("value", C.POINTER(C.c_double))]


class AzSlotInfo(C.Structure):
    _fields_ = [("phase", C.c_int32), ("game_id", C.c_int32), ("ply", C.c_int32), ("sims_done", C.c_int32),
                ("root", C.c_uint32), ("alloc", C.c_uint32), ("bb", C.c_uint64 * 2), ("leaf_bb", C.c_uint64 * 2),
                ("leaf_ply", C.c_int32), ("depth", C.c_int32)]


class AzNetDesc(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("in_planes", C.c_int32),
                ("n_filters", C.c_int32), ("n_blocks", C.c_int32), ("num_actions", C.c_int32), ("device", C.c_int32),
                ("precision", C.c_int32), ("reserved", C.c_int32),
                ("conv_w", C.POINTER(C.c_uint16)), ("conv_epi", C.POINTER(C.c_float)),
                ("in_affine", C.POINTER(C.c_float)), ("skip_w", C.POINTER(C.c_float)),
                ("fc_w", C.POINTER(C.c_uint16)), ("fc_b", C.POINTER(C.c_float)),
                ("conv_w_lo", C.POINTER(C.c_uint16)), ("fc_w_lo", C.POINTER(C.c_uint16))]
NET_PREC = {"f16": 0, "f32x": 1}


class AzReplayConfig(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("game", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("device", C.c_int32), ("reserved", C.c_int32), ("max_games", C.c_int64), ("max_examples", C.c_int64)]


class AzReplayStats(C.Structure):
    _fields_ = [("n_games", C.c_int64), ("n_examples", C.c_int64), ("n_unique", C.c_int64), ("games_dropped", C.c_int64),
                ("fault_flags", C.c_int64)]


# every symbol include/az_engine.h, include/az_net.h and include/az_replay.h declare: (name, restype, argtypes)
_vp = C.c_void_p
PROTOTYPES = [
    ("az_engine_create", C.c_int, [C.POINTER(AzConfig), C.POINTER(_vp)]),
    ("az_engine_destroy", C.c_int, [_vp]),
    ("az_last_error", C.c_char_p, [_vp]),
    ("az_engine_sizes", C.c_int, [_vp, C.POINTER(AzSizes)]),
    ("az_engine_reset", C.c_int, [_vp, C.c_uint64, C.c_int64, _vp]),
    ("az_engine_set_injected_rng", C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64]),
    ("az_engine_set_start_prefix", C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int32]),
    ("az_engine_advance", C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    ("az_engine_advance_slots", C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    ("az_engine_compact_rows", C.c_int, [_vp, C.POINTER(C.c_int32), _vp]),
    ("az_engine_advance_rows", C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, _vp]),
    ("az_engine_opponent_moves", C.c_int, [_vp, _vp]),
    ("az_engine_exchange_moves", C.c_int, [_vp, _vp, _vp]),
    ("az_engine_update_root", C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int32, _vp]),
    ("az_engine_progress", C.c_int, [_vp, C.POINTER(AzProgress), _vp]),
    ("az_engine_poll", C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_uint32), _vp]),
    ("az_engine_export", C.c_int, [_vp, C.POINTER(AzExampleView), _vp]),
    ("az_engine_export_device_bytes", C.c_int64, [_vp]),
    ("az_engine_export_device", C.c_int, [_vp, _vp, C.c_int64, _vp]),
    ("az_engine_read_root", C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double)]),
    ("az_engine_read_slot", C.c_int, [_vp, C.c_int32, C.POINTER(AzSlotInfo)]),
    ("az_engine_read_tree", C.c_int64, [_vp, C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("az_net_create", C.c_int, [C.POINTER(AzNetDesc), C.POINTER(_vp)]),
    ("az_net_destroy", C.c_int, [_vp]),
    ("az_net_last_error", C.c_char_p, [_vp]),
    ("az_net_forward", C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp]),
    ("az_net_reserve", C.c_int, [_vp, C.c_int32]),
    ("az_net_read_tower", C.c_int, [_vp, C.POINTER(C.c_float), C.c_int32]),
    ("az_net_issued_mfma_per_board", C.c_int, [_vp, C.c_int32, C.POINTER(C.c_double)]),
    ("az_net_kernel_label", C.c_char_p, [_vp, C.c_int32]),
    ("az_replay_create", C.c_int, [C.POINTER(AzReplayConfig), C.POINTER(_vp)]),
    ("az_replay_destroy", C.c_int, [_vp]),
    ("az_replay_last_error", C.c_char_p, [_vp]),
    ("az_replay_set_capacity", C.c_int, [_vp, C.c_int64]),
    ("az_replay_append_engine", C.c_int, [_vp, _vp, _vp]),
    ("az_replay_append_host", C.c_int, [_vp, C.POINTER(AzExampleView), C.c_int32, _vp]),
    ("az_replay_append_device", C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _vp]),
    ("az_replay_dedupe", C.c_int, [_vp, _vp]),
    ("az_replay_sample", C.c_int, [_vp, _vp, C.c_int32, C.c_uint64, _vp, _vp, _vp, _vp]),
    ("az_replay_stats_get", C.c_int, [_vp, C.POINTER(AzReplayStats)]),
    ("az_replay_read_unique", C.c_int64, [_vp, C.c_int64, C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_int32)]),
    ("az_replay_debug_set_key", C.c_int, [_vp, C.c_int64, C.c_uint64]),
    ("az_replay_read_example", C.c_int, [_vp, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
]

_lib = None


def load():
    """Load the HIP engine library.  Raises ImportError (with the build command) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "alphazero_openspiel_amd: %s is missing — the HIP engine is not built and there is no CPU "
            "fallback.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C alphazero-openspiel_amd/csrc`." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in PROTOTYPES:
        fn = getattr(lib, name)  # AttributeError here = header/library skew
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
