"""ctypes binding of include/gjx.h — the thin C-ABI layer between the Python host API and the
hand-written HIP kernels (libgjx_hip.so).

This module only declares structs/prototypes and checks status codes.  It never picks a library
by itself: `runtime.py` loads `lib/libgjx_hip.so` and fails loudly if it is missing.  (The test
suite loads the CPU oracle through the same class to drive the identical host logic on CPU
tensors; the product never does.)
"""

from __future__ import annotations

import ctypes as C
import os

c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_uint64)
c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)

GJX_OK = 0
STATUS = {
    0: "GJX_OK",
    -1: "GJX_ERR_INVALID",
    -2: "GJX_ERR_UNSUPPORTED",
    -3: "GJX_ERR_WORKSPACE",
    -4: "GJX_ERR_LAUNCH",
    -5: "GJX_ERR_NO_DEVICE",
    -6: "GJX_ERR_JIT",
}

RNG_THREEFRY = 0
RNG_PHILOX = 1

LSE_RECORD_WORDS = 65  # include/gjx.h: GJX_LSE_RECORD_WORDS
DIST_NORMAL, DIST_GAMMA, DIST_BETA, DIST_BERNOULLI, DIST_CATEGORICAL = range(5)
ARG_CONST, ARG_SITE, ARG_INPUT, ARG_TABLE, ARG_STATE, ARG_OBS, ARG_PARAM, ARG_EXPR = range(8)
# postfix programs as distribution arguments (gjx.h: GJX_ARG_EXPR / gjx_expr_op)
(EXPR_CONST, EXPR_SITE, EXPR_INPUT, EXPR_PARAM, EXPR_STATE, EXPR_OBS, EXPR_ADD, EXPR_SUB, EXPR_MUL, EXPR_NEG, EXPR_DIV, EXPR_EXP,
 EXPR_LOG, EXPR_SQRT, EXPR_ABS, EXPR_MAX, EXPR_MIN, EXPR_LT, EXPR_LE, EXPR_EQ, EXPR_SELECT) = range(21)
EXPR_UNARY = (EXPR_NEG, EXPR_EXP, EXPR_LOG, EXPR_SQRT, EXPR_ABS)
MAP_EXP, MAP_LOG, MAP_DIV, MAP_RDIV, MAP_SQRT, MAP_ABS = range(6)
MAX_EXPR_OPS, MAX_EXPR_DEPTH = 32, 8
MAX_PARAMS = 64
SMC_MAX_STATE, SMC_MAX_OBS = 4, 8
OP_LOGSUMEXP, OP_CATEGORICAL_INDEX, OP_RESAMPLE, OP_SMC = range(4)
MAX_SITES = 64
PLAN_FAST_MATH = 1  # gjx.h: GJX_PLAN_FAST_MATH


class GjxError(RuntimeError):
    def __init__(self, fn: str, code: int):
        super().__init__(f"{fn} failed: {STATUS.get(code, code)}")
        self.code = code


class Keys(C.Structure):
    _fields_ = [
        ("impl", C.c_int32),
        ("mode", C.c_int32),
        ("keys", C.c_void_p),
        ("parent", C.c_uint32 * 2),
        ("first", C.c_uint64),
        ("has_fold", C.c_int32),
        ("fold", C.c_uint32),
        ("parent_lane", C.c_uint64),
    ]


class Scope(C.Structure):
    """include/gjx.h gjx_scope: one nested `@gen` call of a plan (the range of the flat site table it produced)."""

    _fields_ = [("parent", C.c_int32), ("begin", C.c_int32), ("end", C.c_int32)]


MAX_SCOPES = 16


class LseOut(C.Structure):
    """gjx_lse_out: where a fused importance launch leaves the pass's log-sum-exp."""

    _fields_ = [("e", C.c_void_p), ("q", C.c_void_p), ("lse", C.c_void_p), ("record", C.c_void_p),
                ("tickets", C.c_void_p), ("lse_shifted", C.c_void_p), ("shift", C.c_float)]


LSE_TICKET_WORDS = 17 * 64  # include/gjx.h: GJX_LSE_TICKET_WORDS


class EstimateIO(C.Structure):
    """include/gjx.h gjx_estimate_io: what one gjx_importance_estimate call reads besides the key."""

    _fields_ = [("plan", C.c_void_p), ("n", C.c_uint64), ("input_cols", C.c_void_p), ("n_input_cols", C.c_int32), ("impl", C.c_int32),
                ("row_e", C.c_void_p), ("row_s", C.c_void_p), ("lse", LseOut)]


def key_words(impl: int) -> int:
    """Words per materialised key (gjx.h: GJX_KEY_WORDS): threefry 2, philox 4 (cipher key + lane)."""
    return 4 if impl == RNG_PHILOX else 2


class F32(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("scalar", C.c_float)]


class Arg(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("ref", C.c_int32),
        ("scale", C.c_float),
        ("offset", C.c_float),
        ("table", C.c_void_p),
    ]


class ExprOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("ref", C.c_int32), ("value", C.c_float)]


def expr_arg(prog, keep: list) -> Arg:
    """`prog`: [(opcode, ref, value)] -> a GJX_ARG_EXPR argument.  The ctypes array goes into `keep` (the library copies
    the program at plan creation; until then the caller keeps it alive)."""
    arr = (ExprOp * len(prog))(*[ExprOp(int(o), int(r), float(v)) for o, r, v in prog])
    keep.append(arr)
    return Arg(ARG_EXPR, len(prog), 0.0, 0.0, C.addressof(arr))


class Site(C.Structure):
    _fields_ = [
        ("dist", C.c_int32),
        ("observed", C.c_int32),
        ("out_col", C.c_int32),
        ("n_cat", C.c_int32),
        ("n_rows", C.c_int32),
        ("cat_mode", C.c_int32),
        ("arg", Arg * 2),
        ("obs", Arg),
        ("logits", C.c_void_p),
    ]


class SmcModel(C.Structure):
    _fields_ = [
        ("init_sites", C.POINTER(Site)),
        ("n_init_sites", C.c_int32),
        ("step_sites", C.POINTER(Site)),
        ("n_step_sites", C.c_int32),
        ("init_state", Arg * 4),
        ("next_state", Arg * 4),
        ("n_state", C.c_int32),
        ("n_obs", C.c_int32),
    ]


class ScanModel(C.Structure):
    _fields_ = [
        ("step_sites", C.POINTER(Site)),
        ("n_step_sites", C.c_int32),
        ("next_state", Arg * 4),
        ("n_state", C.c_int32),
        ("n_obs", C.c_int32),
    ]


class ScanIO(C.Structure):
    _fields_ = [
        ("particle_keys", C.c_void_p),
        ("n", C.c_uint64),
        ("n_steps", C.c_int32),
        ("obs", C.c_void_p),
        ("carry0", C.c_void_p),
        ("carry0_cols", C.c_void_p),
        ("value_cols", C.c_void_p),
        ("n_value_cols", C.c_int32),
        ("col_stride", C.c_uint64),
        ("carry_out", C.c_void_p),
        ("score", C.c_void_p),
        ("logw", C.c_void_p),
        ("max_partials", C.c_void_p),
        ("row_e", C.c_void_p),
        ("row_s", C.c_void_p),
        ("lse", C.c_void_p),
    ]


class TileRec(C.Structure):
    """gjx_tile_rec: a tile's mass relative to its own power-of-two anchor."""
    _fields_ = [("s", C.c_uint64), ("e", C.c_int32), ("pad", C.c_int32)]


TILE_REC_WORDS, TILE_SUB_WORDS, TILE_ESS_WORDS = 2, 16, 2  # gjx_tile_rec / gjx_tile_sub / gjx_tile_ess as int64 words
TILE_FRAC = 30  # gjx.h: GJX_TILE_FRAC
TILE_EMPTY = -(1 << 30)


class SmcPop(C.Structure):
    """gjx_smc_pop: a population between two steps."""
    _fields_ = [
        ("state", C.c_void_p * 4),
        ("qw", C.c_void_p),
        ("logw", C.c_void_p),
        ("recs", C.c_void_p),
        ("subs", C.c_void_p),
        ("ess", C.c_void_p),
        ("prefix", C.c_void_p),
    ]


class ShardedIO(C.Structure):
    _fields_ = [
        ("pop", SmcPop * 2),
        ("out_e", C.c_void_p),
        ("out_q", C.c_void_p),
        ("ancestors", C.c_void_p),
        ("ranges", C.c_void_p),
        ("shuffle", C.c_int32),
        ("received", C.POINTER(C.c_uint64)),
        ("stage", C.c_void_p),            # r04: dev [world, tiles_local, 32 B]: adaptive filters on a collective transport
    ]


class Seg(C.Structure):
    """gjx_seg: elements [a, b) of a global column travel to / from `peer`."""
    _fields_ = [("peer", C.c_int32), ("a", C.c_uint64), ("b", C.c_uint64)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32,
                          C.POINTER(Seg), C.c_int32, C.POINTER(Seg), C.c_int32, C.c_void_p)
STREAM_SYNC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


COMM_ID_BYTES = 128


class Lgssm(C.Structure):
    _fields_ = [
        ("x0_loc", C.c_float),
        ("x0_scale", C.c_float),
        ("a", C.c_float),
        ("q", C.c_float),
        ("r", C.c_float),
    ]


class Hmm(C.Structure):
    _fields_ = [
        ("n_states", C.c_int32),
        ("init_state", C.c_int32),
        ("trans_logits", C.c_void_p),
        ("obs_logits", C.c_void_p),
    ]


MAX_PEERS = 8


class SmcPeers(C.Structure):
    """gjx_smc_peers: the peer transport of a sharded filter (arenas of identical layout; delta[o] = byte distance from this
    rank's arena to rank o's as mapped in this process; flags / error inside the arena)."""
    _fields_ = [
        ("world", C.c_int32),
        ("rank", C.c_int32),
        ("delta", C.c_int64 * MAX_PEERS),
        ("flags", C.c_void_p),
        ("error", C.c_void_p),
        ("wait_value", C.c_uint64),
        ("timeout_ms", C.c_uint32),
        ("pad", C.c_int32),
        ("signal_recs", C.c_void_p),      # a deferred signal (gjx.h): set by the sharded drivers
        ("signal_ess", C.c_void_p),
        ("signal_first_tile", C.c_uint64),
        ("signal_n_tiles", C.c_uint64),
        ("signal_value", C.c_uint64),
    ]


class SmcConfig(C.Structure):
    _fields_ = [
        ("impl", C.c_int32),
        ("n_total", C.c_uint64),
        ("first_slot", C.c_uint64),
        ("n_local", C.c_uint64),
        ("n_steps", C.c_int32),
        ("step_keys", C.c_void_p),
        ("resample_keys", C.c_void_p),
        ("n_filters", C.c_int32),
        ("filter_stride", C.c_uint64),
        ("ess_threshold", C.c_float),     # 0 / >= 1: resample at every step; (0, 1): only when ESS < threshold * N
        ("resampled_out", C.c_void_p),    # dev int32[T] / [F, T]: 1 where a step began with a resampling
        ("peers", C.POINTER(SmcPeers)),   # r04, nullable: the source population lives in the peers' arenas
    ]


_P = C.c_void_p
_KP = C.POINTER(Keys)

# name -> (restype, argtypes).  Every symbol include/gjx.h declares must be listed here;
# tests/test_abi_symbols.py cross-checks the header against this table and the built .so.
PROTOTYPES = {
    "gjx_version": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gjx_backend_name": (C.c_char_p, []),
    "gjx_rng_keys": (C.c_int, [_KP, C.c_uint64, _P, _P]),
    "gjx_rng_split_each": (C.c_int, [_KP, C.c_uint64, C.c_uint32, _P, _P]),
    "gjx_rng_bits": (C.c_int, [_KP, C.c_uint32, C.c_uint64, _P, _P]),
    "gjx_sample_logpdf_normal": (C.c_int, [_KP, F32, F32, _P, _P, C.c_uint64, _P]),
    "gjx_sample_logpdf_gamma": (C.c_int, [_KP, F32, F32, _P, _P, C.c_uint64, _P]),
    "gjx_sample_logpdf_beta": (C.c_int, [_KP, F32, F32, _P, _P, C.c_uint64, _P]),
    "gjx_sample_logpdf_bernoulli": (C.c_int, [_KP, F32, _P, _P, C.c_uint64, _P]),
    "gjx_sample_logpdf_categorical": (
        C.c_int,
        [_KP, _P, C.c_uint64, C.c_uint32, _P, C.c_int, _P, _P, C.c_uint64, _P],
    ),
    "gjx_map_f32": (C.c_int, [C.c_int, _P, C.c_float, _P, C.c_uint64, _P]),
    "gjx_logpdf_normal": (C.c_int, [F32, F32, F32, _P, C.c_uint64, _P]),
    "gjx_logpdf_gamma": (C.c_int, [F32, F32, F32, _P, C.c_uint64, _P]),
    "gjx_logpdf_beta": (C.c_int, [F32, F32, F32, _P, C.c_uint64, _P]),
    "gjx_logpdf_bernoulli": (C.c_int, [_P, C.c_int, F32, _P, C.c_uint64, _P]),
    "gjx_logpdf_categorical": (
        C.c_int,
        [_P, C.c_int, _P, C.c_uint64, C.c_uint32, _P, _P, C.c_uint64, _P],
    ),
    "gjx_plan_create": (C.c_int, [C.POINTER(Site), C.c_int, C.POINTER(_P)]),
    "gjx_plan_create_ex": (C.c_int, [C.POINTER(Site), C.c_int, C.c_uint32, C.POINTER(_P)]),
    "gjx_plan_create_scoped": (C.c_int, [C.POINTER(Site), C.c_int, C.POINTER(Scope), C.c_int, C.c_uint32, C.POINTER(_P)]),
    "gjx_plan_destroy": (C.c_int, [_P]),
    "gjx_plan_set_params": (C.c_int, [_P, _P, C.c_int]),
    "gjx_plan_specialized_source": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gjx_plan_compile_check": (C.c_int, [_P, C.c_int]),
    "gjx_plan_prepare": (C.c_int, [_P, _KP]),
    "gjx_jit_stats": (C.c_int, [_P, _P, _P]),
    "gjx_jit_routes": (C.c_int, [_P, _P, _P, _P]),
    "gjx_smc_run_graph_stats": (C.c_int, [_P, _P]),
    "gjx_jit_compile_source": (C.c_int, [C.c_char_p]),
    "gjx_importance_run": (
        C.c_int,
        [_P, _KP, C.POINTER(_P), C.c_int, C.POINTER(_P), C.c_int, _P, _P, C.c_uint64, _P, _P, _P, _P, _P],
    ),
    "gjx_importance_estimate": (C.c_int, [C.POINTER(EstimateIO), C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_float, C.c_void_p]),
    "gjx_importance_run_batch": (
        C.c_int,
        [_P, _KP, C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(_P), C.c_int, C.POINTER(_P), C.c_int, _P, _P, C.c_uint64,
         _P, _P, _P, _P],
    ),
    "gjx_workspace_bytes": (C.c_size_t, [C.c_int, C.c_uint64]),
    "gjx_frac_bits": (C.c_int, [C.c_uint64]),
    "gjx_num_tiles": (C.c_uint64, [C.c_uint64]),
    "gjx_num_max_partials": (C.c_uint64, [C.c_uint64]),
    "gjx_row_stats": (C.c_int, [_P, C.c_uint64, _P, _P, _P]),
    "gjx_lse_rows": (C.c_int, [_P, _P, C.c_uint64, _P, _P, _P, _P, _P]),
    "gjx_lse_rows_batch": (C.c_int, [_P, _P, C.c_uint64, C.c_int32, C.c_uint64, _P, _P, _P, _P, _P]),
    "gjx_lse_combine": (C.c_int, [_P, C.c_int32, C.c_uint64, C.c_int32, C.c_uint64, _P, _P, _P, _P, _P]),
    "gjx_max_f32": (C.c_int, [_P, C.c_uint64, _P, _P, _P, C.c_size_t, _P]),
    "gjx_expsum_fix": (C.c_int, [_P, C.c_uint64, _P, C.c_int, _P, _P, C.c_size_t, _P]),
    "gjx_lse_finish": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "gjx_logsumexp_f32": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "gjx_categorical_index": (C.c_int, [_KP, _P, C.c_uint64, _P, C.c_int, _P, C.c_size_t, _P]),
    "gjx_resample_systematic": (
        C.c_int,
        [_KP, _P, C.c_uint64, C.c_uint64, _P, _P, _P, _P, C.c_size_t, _P],
    ),
    "gjx_resample_multinomial": (
        C.c_int,
        [_KP, _P, C.c_uint64, C.c_uint64, _P, _P, _P, _P, C.c_size_t, _P],
    ),
    "gjx_gather_cols": (C.c_int, [_P, C.c_uint64, C.POINTER(_P), C.POINTER(_P), C.c_int, _P]),
    "gjx_smc_tile": (C.c_uint64, []),
    "gjx_smc_run_lgssm": (
        C.c_int,
        [C.POINTER(SmcConfig), C.POINTER(Lgssm), _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P],
    ),
    "gjx_smc_run_hmm": (
        C.c_int,
        [C.POINTER(SmcConfig), C.POINTER(Hmm), _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P],
    ),
    "gjx_smc_plan_create": (C.c_int, [C.POINTER(SmcModel), C.POINTER(_P)]),
    "gjx_smc_plan_create_scoped": (C.c_int, [C.POINTER(SmcModel), C.POINTER(Scope), C.c_int, C.POINTER(Scope), C.c_int, C.POINTER(_P)]),
    "gjx_smc_plan_destroy": (C.c_int, [_P]),
    "gjx_smc_plan_compile_check": (C.c_int, [_P, C.c_int]),
    "gjx_scan_plan_create": (C.c_int, [C.POINTER(ScanModel), C.c_uint32, C.POINTER(_P)]),
    "gjx_scan_plan_create_scoped": (C.c_int, [C.POINTER(ScanModel), C.POINTER(Scope), C.c_int, C.c_uint32, C.POINTER(_P)]),
    "gjx_scan_plan_destroy": (C.c_int, [_P]),
    "gjx_scan_plan_compile_check": (C.c_int, [_P, C.c_int]),
    "gjx_scan_run": (C.c_int, [_P, C.POINTER(ScanIO), _P]),
    "gjx_comm_unique_id": (C.c_int, [_P]),
    "gjx_comm_init_rccl": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(_P)]),
    "gjx_comm_group_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gjx_comm_group_destroy": (C.c_int, [_P]),
    "gjx_comm_init_local": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "gjx_comm_destroy": (C.c_int, [_P]),
    "gjx_comm_rank": (C.c_int, [_P]),
    "gjx_comm_world": (C.c_int, [_P]),
    "gjx_comm_lse_combine": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "gjx_smc_sharded_run_lgssm": (C.c_int, [_P, C.POINTER(SmcConfig), C.POINTER(Lgssm), _P, C.POINTER(ShardedIO), _P]),
    "gjx_smc_sharded_run_hmm": (C.c_int, [_P, C.POINTER(SmcConfig), C.POINTER(Hmm), _P, _P, _P, C.POINTER(ShardedIO), _P]),
    "gjx_smc_sharded_run_plan": (C.c_int, [_P, C.POINTER(SmcConfig), _P, _P, C.POINTER(ShardedIO), _P]),
    "gjx_smc_run_plan": (
        C.c_int,
        [C.POINTER(SmcConfig), _P, _P, _P, _P, C.POINTER(_P), _P, _P, _P, C.c_size_t, _P],
    ),
    "gjx_smc_lgssm_step": (
        C.c_int,
        [C.POINTER(SmcConfig), C.POINTER(Lgssm), C.c_int, C.c_float, C.POINTER(SmcPop), C.POINTER(SmcPop), _P, _P, _P, _P],
    ),
    "gjx_smc_hmm_step": (
        C.c_int,
        [C.POINTER(SmcConfig), C.POINTER(Hmm), C.c_int, C.c_int32, C.POINTER(SmcPop), C.POINTER(SmcPop), _P, _P, _P, _P, _P, _P],
    ),
    "gjx_smc_plan_step": (C.c_int, [C.POINTER(SmcConfig), _P, C.c_int, _P, C.POINTER(SmcPop), C.POINTER(SmcPop), _P, _P, _P, _P]),
    "gjx_smc_finish": (C.c_int, [C.POINTER(SmcConfig), _P, _P, _P, _P]),
    "gjx_smc_source_ranges": (C.c_int, [C.POINTER(SmcConfig), _P, _P, C.c_int, C.c_int64, _P, _P]),
    "gjx_tile_weights": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P]),
    "gjx_tile_merge": (C.c_int, [_P, C.c_uint64, _P, _P, _P]),
    "gjx_comm_init_callbacks": (C.c_int, [C.c_int, C.c_int, ALLGATHER_FN, EXCHANGE_FN, STREAM_SYNC_FN, _P, C.POINTER(_P)]),
    "gjx_categorical_index_batch": (C.c_int, [_KP, C.c_int32, _P, C.c_uint64, C.c_uint64, _P, _P]),
    "gjx_comm_init_peers": (C.c_int, [C.POINTER(SmcPeers), _P, C.c_int, C.POINTER(_P)]),
    "gjx_smc_peer_signal": (C.c_int, [C.POINTER(SmcPeers), _P, _P, C.c_uint64, C.c_uint64, C.c_uint64, _P]),
    "gjx_smc_peer_wait": (C.c_int, [C.POINTER(SmcPeers), C.c_uint64, _P]),
    "gjx_smc_peer_signal_fused": (C.c_int, [C.POINTER(SmcConfig)]),
    "gjx_smc_records_pack": (C.c_int, [C.POINTER(SmcConfig), C.c_int, C.c_int, _P, _P, _P, _P]),
    "gjx_hmm_alias_words": (C.c_uint64, [C.c_int32]),
    "gjx_hmm_prepare": (C.c_int, [C.POINTER(Hmm), _P, _P, _P]),
}

_NO_STATUS = {
    "gjx_backend_name",
    "gjx_workspace_bytes",
    "gjx_frac_bits",
    "gjx_smc_tile",
    "gjx_num_tiles",
    "gjx_num_max_partials",
    "gjx_hmm_alias_words",
    "gjx_comm_rank",
    "gjx_comm_world",
    "gjx_smc_peer_signal_fused",
}


# The (major, minor) of include/gjx.h these bindings were written for.  Struct layouts and the sampling spec change behind
# unchanged entry points from minor to minor (0.7 -> 0.8: LseOut / SmcConfig / ShardedIO layouts, the Philox Normal spec), so a
# library of another version is refused at load: with mismatched layouts it would read garbage pointers (silent corruption,
# or a GPU fault on a shared box).
ABI_VERSION = (0, 10)


class AbiVersionMismatch(RuntimeError):
    pass


class GjxLib:
    """A loaded implementation of include/gjx.h.  `device_type` is the torch device type whose
    memory the library's "dev" pointers refer to ("cuda" for libgjx_hip.so)."""

    def __init__(self, path: str, device_type: str):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.device_type = device_type
        self._dll = C.CDLL(path)
        # the version FIRST (gjx_version exists in every build): a library of another minor may lack newer symbols, and the
        # useful error is "wrong version", not "missing symbol"
        vfn = self._dll.gjx_version
        vfn.restype, vfn.argtypes = PROTOTYPES["gjx_version"]
        major, minor = C.c_int(-1), C.c_int(-1)
        vfn(C.byref(major), C.byref(minor))
        if (major.value, minor.value) != ABI_VERSION:
            raise AbiVersionMismatch(
                f"{path} implements gjx.h {major.value}.{minor.value}; these bindings are written for "
                f"{ABI_VERSION[0]}.{ABI_VERSION[1]} (struct layouts and the sampling spec differ between minors): rebuild the "
                "library from this tree, or run the other library with its own tree's bindings")
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(self._dll, name)  # AttributeError => ABI symbol missing: fail loudly
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)
        major, minor = C.c_int(), C.c_int()
        self.call("gjx_version", C.byref(major), C.byref(minor))
        self.version = (major.value, minor.value)
        self.name = self._gjx_backend_name().decode()

    def call(self, name: str, *args):
        rc = getattr(self, "_" + name)(*args)
        if name not in _NO_STATUS and rc != GJX_OK:
            raise GjxError(name, rc)
        return rc
