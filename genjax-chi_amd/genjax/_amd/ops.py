"""Tensor-level wrappers over the C-ABI (include/gjx.h).

`Ops` owns a loaded `GjxLib` and turns torch tensors into raw pointers + sizes.  torch is only the
device-memory holder and stream provider here; all arithmetic happens inside the library.  Every
method checks that its tensors live on the library's device type, are contiguous and have the
dtype the ABI documents — operand shapes are validated on the host before any kernel launches.
"""

from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import torch

from . import abi
from .abi import GjxLib, Keys, F32


@dataclass(frozen=True)
class KeyBatch:
    """Host-side description of `n` per-particle keys (mirrors gjx_keys)."""

    impl: int
    mode: int  # 0 explicit tensor [n, key_words] (int32 view of u32), 1 lazy split, 2 literal
    tensor: torch.Tensor | None = None
    parent: tuple[int, int] = (0, 0)
    first: int = 0
    fold: int | None = None
    parent_lane: int = 0  # philox: lane of the parent key (modes 1, 2)

    def with_fold(self, fold: int) -> "KeyBatch":
        if self.fold is not None:
            raise ValueError("key batch already carries a fold")
        return KeyBatch(self.impl, self.mode, self.tensor, self.parent, self.first, int(fold), self.parent_lane)


class Ops:
    def __init__(self, lib: GjxLib):
        self.lib = lib
        self.device_type = lib.device_type
        self.tile = int(lib.call("gjx_smc_tile"))
        self._ws: dict = {}
        # allocations name the device TYPE: torch resolves "the current device" itself (a Python-level
        # torch.cuda.current_device() per allocation was a tenth of an eager call's host time)
        self._alloc_device = torch.device(self.device_type)

    # ---- plumbing ---------------------------------------------------------------------------
    def device(self) -> torch.device:
        if self.device_type == "cuda":
            return torch.device("cuda", torch.cuda.current_device())
        return torch.device("cpu")

    def stream(self):
        if self.device_type == "cuda":
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return C.c_void_p(0)

    def _chk(self, t: torch.Tensor, dtype, n: int | None = None, name: str = "tensor"):
        if t.device.type != self.device_type:
            raise ValueError(f"{name}: expected a {self.device_type} tensor, got {t.device}")
        if t.dtype != dtype:
            raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError(f"{name}: must be contiguous")
        if n is not None and t.numel() != n:
            raise ValueError(f"{name}: expected {n} elements, got {t.numel()}")
        return C.c_void_p(t.data_ptr())

    def empty(self, n, dtype):
        return torch.empty(n, dtype=dtype, device=self._alloc_device)

    def empty_columns(self, n: int, dtypes: list) -> list[torch.Tensor]:
        """len(dtypes) columns of n 4-byte elements carved from ONE allocation (rows of a [k, n] block)."""
        if not dtypes:
            return []
        if any(dt not in (torch.float32, torch.int32) for dt in dtypes):
            return [self.empty(n, dt) for dt in dtypes]
        cols = torch.empty((len(dtypes), n), dtype=torch.float32, device=self._alloc_device).unbind(0)
        return [c if dt == torch.float32 else c.view(torch.int32) for c, dt in zip(cols, dtypes)]

    def workspace(self, op: int, n: int) -> tuple[torch.Tensor, int]:
        """Scratch for one call, from torch's caching allocator: a block is handed out again only to work ordered after
        this call on the same stream, so calls on different streams / threads never share scratch (a cache of our own,
        keyed by operation, let two concurrent filters of different streams write the same tile sums)."""
        nbytes = int(self.lib.call("gjx_workspace_bytes", op, n))
        return torch.empty(max(nbytes, 1024), dtype=torch.uint8, device=self._alloc_device), nbytes

    def _keys(self, kb: KeyBatch, n: int) -> Keys:
        k = Keys()
        k.impl = kb.impl
        k.mode = kb.mode
        if kb.mode == 0:
            assert kb.tensor is not None
            k.keys = self._chk(kb.tensor, torch.int32, abi.key_words(kb.impl) * n, "keys").value
        else:
            k.keys = None
            k.parent[0], k.parent[1] = kb.parent
            k.first = kb.first
            k.parent_lane = kb.parent_lane
        k.has_fold = 0 if kb.fold is None else 1
        k.fold = 0 if kb.fold is None else kb.fold & 0xFFFFFFFF
        return k

    def _f32(self, v, n: int, name: str) -> F32:
        if isinstance(v, torch.Tensor):
            if v.numel() == 1 and v.dim() == 0:
                return F32(None, float(v))
            return F32(self._chk(v, torch.float32, n, name).value, 0.0)
        return F32(None, float(v))

    def num_tiles(self, n: int) -> int:
        return int(self.lib.call("gjx_num_tiles", n))

    def num_max_partials(self, n: int) -> int:
        return int(self.lib.call("gjx_num_max_partials", n))

    def frac_bits(self, n_total: int) -> int:
        return int(self.lib.call("gjx_frac_bits", n_total))

    # ---- RNG --------------------------------------------------------------------------------
    def rng_keys(self, kb: KeyBatch, n: int) -> torch.Tensor:
        out = self.empty((n, abi.key_words(kb.impl)), torch.int32)
        self.lib.call("gjx_rng_keys", C.byref(self._keys(kb, n)), n, C.c_void_p(out.data_ptr()), self.stream())
        return out

    def rng_split_each(self, kb: KeyBatch, n: int, m: int) -> torch.Tensor:
        out = self.empty((n * m, abi.key_words(kb.impl)), torch.int32)
        self.lib.call("gjx_rng_split_each", C.byref(self._keys(kb, n)), n, m, C.c_void_p(out.data_ptr()), self.stream())
        return out

    def rng_bits(self, kb: KeyBatch, n: int, sub: int = 0) -> torch.Tensor:
        out = self.empty(n, torch.int32)
        self.lib.call("gjx_rng_bits", C.byref(self._keys(kb, n)), sub, n, C.c_void_p(out.data_ptr()), self.stream())
        return out

    # ---- distributions ------------------------------------------------------------------------
    def sample_logpdf(self, dist: str, kb: KeyBatch, n: int, a, b=None, want_score=True):
        score = self.empty(n, torch.float32) if want_score else None
        sp = C.c_void_p(score.data_ptr()) if want_score else None
        k = C.byref(self._keys(kb, n))
        if dist in ("normal", "gamma", "beta"):
            val = self.empty(n, torch.float32)
            self.lib.call(f"gjx_sample_logpdf_{dist}", k, self._f32(a, n, "arg0"), self._f32(b, n, "arg1"),
                          C.c_void_p(val.data_ptr()), sp, n, self.stream())
        elif dist == "bernoulli":
            val = self.empty(n, torch.uint8)
            self.lib.call("gjx_sample_logpdf_bernoulli", k, self._f32(a, n, "probs"),
                          C.c_void_p(val.data_ptr()), sp, n, self.stream())
        else:
            raise ValueError(dist)
        return val, score

    def sample_logpdf_categorical(self, kb: KeyBatch, n: int, logits: torch.Tensor, row_index=None,
                                  mode: int = 1, want_score=True):
        if logits.dim() != 2:
            raise ValueError("logits must be [n_rows, n_cat]")
        n_rows, n_cat = logits.shape
        if row_index is None and n_rows not in (1, n):
            raise ValueError("logits rows must be 1 or n when no row_index is given")
        lp = self._chk(logits, torch.float32, name="logits")
        rp = None if row_index is None else self._chk(row_index, torch.int32, n, "row_index")
        val = self.empty(n, torch.int32)
        score = self.empty(n, torch.float32) if want_score else None
        self.lib.call("gjx_sample_logpdf_categorical", C.byref(self._keys(kb, n)), lp, n_rows, n_cat, rp, mode,
                      C.c_void_p(val.data_ptr()), C.c_void_p(score.data_ptr()) if want_score else None, n,
                      self.stream())
        return val, score

    def logpdf(self, dist: str, n: int, value, a, b=None) -> torch.Tensor:
        score = self.empty(n, torch.float32)
        sp = C.c_void_p(score.data_ptr())
        if dist in ("normal", "gamma", "beta"):
            self.lib.call(f"gjx_logpdf_{dist}", self._f32(value, n, "value"), self._f32(a, n, "arg0"),
                          self._f32(b, n, "arg1"), sp, n, self.stream())
        elif dist == "bernoulli":
            if isinstance(value, torch.Tensor) and value.numel() > 1:
                vp, vs = self._chk(value, torch.uint8, n, "value"), 0
            else:
                vp, vs = None, int(bool(value))
            self.lib.call("gjx_logpdf_bernoulli", vp, vs, self._f32(a, n, "probs"), sp, n, self.stream())
        else:
            raise ValueError(dist)
        return score

    def map_f32(self, op: int, x: torch.Tensor, c: float = 0.0) -> torch.Tensor:
        """gjx_map_f32 over a tensor of any shape: the spec's exp / log, x / c, c / x (abi.MAP_*) — the bits the same
        operation computes inside a fused plan (GJX_EXPR_EXP / _LOG / _DIV)."""
        x = x.to(torch.float32)
        if x.device.type != self.device_type:
            x = x.to(self.device())
        x = x.contiguous()
        out = torch.empty_like(x)
        n = x.numel()
        if n:
            self.lib.call("gjx_map_f32", int(op), C.c_void_p(x.data_ptr()), float(c), C.c_void_p(out.data_ptr()), n, self.stream())
        return out

    def logpdf_categorical(self, n: int, value, logits: torch.Tensor, row_index=None) -> torch.Tensor:
        n_rows, n_cat = logits.shape
        if row_index is None and n_rows not in (1, n):
            raise ValueError("logits rows must be 1 or n when no row_index is given")
        score = self.empty(n, torch.float32)
        if isinstance(value, torch.Tensor) and value.numel() > 1:
            vp, vs = self._chk(value, torch.int32, n, "value"), 0
        else:
            vp, vs = None, int(value)
        rp = None if row_index is None else self._chk(row_index, torch.int32, n, "row_index")
        self.lib.call("gjx_logpdf_categorical", vp, vs, self._chk(logits, torch.float32, name="logits"), n_rows,
                      n_cat, rp, C.c_void_p(score.data_ptr()), n, self.stream())
        return score

    # ---- plans ----------------------------------------------------------------------------------
    def plan_create(self, sites: list[abi.Site], fast_math: bool = False, scopes=()) -> "Plan":
        """`fast_math`: gjx.h GJX_PLAN_FAST_MATH — hardware transcendentals for the continuous parts of the walk
        (<= 1e-5 relative on values and log-weights; no longer bit-identical to the oracle).
        `scopes`: (parent, begin, end) of every nested `@gen` call, in call order (gjx.h gjx_scope)."""
        arr = (abi.Site * len(sites))(*sites)
        handle = C.c_void_p()
        flags = abi.PLAN_FAST_MATH if fast_math else 0
        if scopes:
            sc = (abi.Scope * len(scopes))(*[abi.Scope(*k) for k in scopes])
            self.lib.call("gjx_plan_create_scoped", arr, len(sites), sc, len(scopes), flags, C.byref(handle))
        else:
            self.lib.call("gjx_plan_create_ex", arr, len(sites), flags, C.byref(handle))
        return Plan(self, handle, len(sites))

    def jit_stats(self) -> dict:
        """gjx_jit_stats: hiprtc compilations so far, code objects loaded now, code objects evicted."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.lib.call("gjx_jit_stats", C.byref(a), C.byref(b), C.byref(c))
        return dict(compiles=a.value, cached_modules=b.value, evictions=c.value)

    def jit_routes(self) -> dict:
        """gjx_jit_routes: which route compiled this process's generated kernels (the helper process / in-process hiprtc),
        and how many helpers failed or could not be started."""
        v = [C.c_uint64() for _ in range(4)]
        self.lib.call("gjx_jit_routes", *[C.byref(x) for x in v])
        return dict(child_compiles=v[0].value, inproc_compiles=v[1].value, child_failures=v[2].value, spawn_failures=v[3].value)

    def smc_run_graph_stats(self) -> dict:
        """gjx_smc_run_graph_stats: whole runs replayed as one hipGraph (captures = graphs instantiated, replays = runs that were
        one graph launch)."""
        a, b = C.c_uint64(), C.c_uint64()
        self.lib.call("gjx_smc_run_graph_stats", C.byref(a), C.byref(b))
        return dict(captures=a.value, replays=b.value)

    def tickets(self) -> torch.Tensor:
        """The zeroed ticket words of fused log-sum-exp launches on the current stream (gjx_lse_out.tickets:
        every launch leaves them zero, launches sharing them must be stream-ordered)."""
        key = ("tickets", str(self.device()), self.stream().value)
        t = self._ws.get(key)
        if t is None:
            t = torch.zeros(abi.LSE_TICKET_WORDS, dtype=torch.int32, device=self.device())
            self._ws[key] = t
        return t

    def importance_run(self, plan: "Plan", kb: KeyBatch, n: int, input_cols: list[torch.Tensor],
                       value_dtypes: list, want_score=True, want_max_partials=True, want_rows=False, fuse_lse=False):
        """One fused `@gen` walk.  With want_rows the launch also emits the row-anchored partial sums of its
        log-weights (`lse_rows(rows)` folds them); with fuse_lse it folds them itself (rows.lse / e_out / q_out)."""
        if kb.fold is not None:
            raise ValueError("particle keys must not carry a fold")
        ins = (C.c_void_p * max(1, len(input_cols)))()
        for i, t in enumerate(input_cols):
            ins[i] = self._chk(t, torch.float32, n, f"input_cols[{i}]").value
        # value columns, score and log-weights: rows of one block (n a multiple of 4 keeps every row 16-byte aligned,
        # which the four-particles-per-lane kernel form asks for)
        block = self.empty_columns(n, list(value_dtypes) + [torch.float32] * (2 if want_score else 1))
        vals = block[:len(value_dtypes)]
        outs = (C.c_void_p * max(1, len(vals)))()
        for i, t in enumerate(vals):
            outs[i] = t.data_ptr()
        score = block[-2] if want_score else None
        logw = block[-1]
        mp = self.empty(self.num_max_partials(n), torch.float32) if want_max_partials else None
        rows, lse = None, None
        if want_rows:
            # the fold is a separate one-workgroup launch (`lse_rows`, on demand).  gjx_importance_run can also
            # fold inside the launch (gjx_lse_out, `fuse_lse=True`): one launch fewer, no faster at 1e6 particles,
            # and it leans on an in-launch hand-off between workgroups — the eager API keeps the plain form.
            rows = RowStats(self.empty(self.num_max_partials(n), torch.int32),
                            self.empty(self.num_max_partials(n), torch.int64), n)
            if fuse_lse:
                rows.lse, rows.e_out, rows.q_out = (self.empty(1, torch.float32), self.empty(1, torch.int32),
                                                    self.empty(1, torch.int64))
                lse = abi.LseOut(rows.e_out.data_ptr(), rows.q_out.data_ptr(), rows.lse.data_ptr(), None,
                                 self.tickets().data_ptr())
        self.lib.call("gjx_importance_run", plan.handle, C.byref(self._keys(kb, n)), ins, len(input_cols), outs,
                      len(vals), C.c_void_p(score.data_ptr()) if want_score else None,
                      C.c_void_p(logw.data_ptr()), n, C.c_void_p(mp.data_ptr()) if mp is not None else None,
                      self._p(rows.e) if rows else None, self._p(rows.s) if rows else None,
                      C.byref(lse) if lse is not None else None, self.stream())
        if want_rows:
            return vals, score, logw, mp, rows
        return vals, score, logw, mp

    # ---- row-anchored log-sum-exp ----------------------------------------------
    def row_stats(self, x: torch.Tensor) -> "RowStats":
        n = x.numel()
        rows = RowStats(self.empty(self.num_max_partials(n), torch.int32), self.empty(self.num_max_partials(n), torch.int64), n)
        self.lib.call("gjx_row_stats", self._chk(x, torch.float32, n, "x"), n, self._p(rows.e), self._p(rows.s),
                      self.stream())
        return rows

    def lse_rows(self, rows: "RowStats", record: torch.Tensor | None = None):
        """-> (lse f32[1], e i32[1], q i64[1]) on device; `record` (int64[65]) also receives the
        exchangeable (anchor, buckets) summary of these rows (gjx.h: GJX_LSE_RECORD_WORDS)."""
        if record is None and rows.lse is not None:
            return rows.lse, rows.e_out, rows.q_out  # already folded by the launch that produced the rows
        lse, e, q = self.empty(1, torch.float32), self.empty(1, torch.int32), self.empty(1, torch.int64)
        self.lib.call("gjx_lse_rows", self._p(rows.e), self._p(rows.s), rows.e.numel(), self._p(e), self._p(q),
                      self._p(lse), None if record is None else self._chk(record, torch.int64, abi.LSE_RECORD_WORDS, "record"),
                      self.stream())
        return lse, e, q

    def lse_combine(self, records: torch.Tensor, n_batch: int | None = None):
        """records int64[n_records, batch, 65] (the all-gather of per-rank [batch, 65] blocks) ->
        (lse f32[n_batch], e i32[n_batch], q i64[n_batch]): the merged log-sum-exp of the first
        n_batch passes, bit-identical to one lse_rows over the whole population."""
        g, b, w = records.shape
        if w != abi.LSE_RECORD_WORDS or records.dtype != torch.int64 or not records.is_contiguous():
            raise ValueError("records must be contiguous int64[n_records, batch, 65]")
        n_batch = b if n_batch is None else n_batch
        lse, e, q = self.empty(n_batch, torch.float32), self.empty(n_batch, torch.int32), self.empty(n_batch, torch.int64)
        self.lib.call("gjx_lse_combine", self._p(records), g, b * w, n_batch, w, self._p(e), self._p(q), self._p(lse),
                      None, self.stream())
        return lse, e, q

    @staticmethod
    def log_z_from_rows(e: torch.Tensor, q: torch.Tensor, n_total: int) -> float:
        """float64 log Z = e ln2 + log(q) - 30 ln2 - log N from the exact (e, q) pair."""
        qi = int(q.cpu())
        if qi <= 0:
            return float("-inf")  # no mass: every weight underflowed (or was -inf)
        return int(e.cpu()) * math.log(2.0) + math.log(qi) - 30 * math.log(2.0) - math.log(n_total)

    def prepare_importance(self, plan: "Plan", kb, n: int, input_cols: list[torch.Tensor],
                           value_dtypes: list, with_lse: bool = True, fold_batch: int = 1,
                           estimate_only: bool = False) -> "PreparedImportance":
        """Pre-bind one importance pass (+ its log-sum-exp) to persistent output buffers: a launch
        is then two C calls with no allocation or marshalling on the host (what a latency-bound
        1e6-particle step needs; it is also what a HIP-graph capture of the step would replay)."""
        return PreparedImportance(self, plan, kb, n, input_cols, value_dtypes, with_lse, fold_batch, estimate_only)

    # ---- weights --------------------------------------------------------------------------------
    def max_f32(self, x: torch.Tensor | None, n: int, max_partials=None, out=None) -> torch.Tensor:
        out = self.empty(1, torch.float32) if out is None else out
        ws, nb = self.workspace(abi.OP_LOGSUMEXP, n)
        self.lib.call("gjx_max_f32", None if x is None else self._chk(x, torch.float32, n, "x"), n,
                      None if max_partials is None else self._chk(max_partials, torch.float32, self.num_max_partials(n)),
                      C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), nb, self.stream())
        return out

    def expsum_fix(self, x: torch.Tensor, max_dev: torch.Tensor, frac: int, out=None) -> torch.Tensor:
        n = x.numel()
        out = self.empty(1, torch.int64) if out is None else out
        ws, nb = self.workspace(abi.OP_LOGSUMEXP, n)
        self.lib.call("gjx_expsum_fix", self._chk(x, torch.float32, n, "x"), n,
                      self._chk(max_dev, torch.float32, 1, "max"), frac, C.c_void_p(out.data_ptr()),
                      C.c_void_p(ws.data_ptr()), nb, self.stream())
        return out

    def lse_finish(self, max_dev: torch.Tensor, q_dev: torch.Tensor, frac: int) -> torch.Tensor:
        out = self.empty(1, torch.float32)
        self.lib.call("gjx_lse_finish", self._chk(max_dev, torch.float32, 1), self._chk(q_dev, torch.int64, 1),
                      frac, C.c_void_p(out.data_ptr()), self.stream())
        return out

    def logsumexp(self, x: torch.Tensor, max_partials=None):
        """-> (lse f32[1], max f32[1], q i64[1]) on device."""
        n = x.numel()
        lse, m, q = self.empty(1, torch.float32), self.empty(1, torch.float32), self.empty(1, torch.int64)
        ws, nb = self.workspace(abi.OP_LOGSUMEXP, n)
        self.lib.call("gjx_logsumexp_f32", self._chk(x, torch.float32, n, "x"), n,
                      None if max_partials is None else self._chk(max_partials, torch.float32, self.num_max_partials(n)),
                      C.c_void_p(lse.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(q.data_ptr()),
                      C.c_void_p(ws.data_ptr()), nb, self.stream())
        return lse, m, q

    def categorical_index(self, kb: KeyBatch, logits: torch.Tensor, mode: int = 0) -> torch.Tensor:
        n = logits.numel()
        out = self.empty(1, torch.int64)
        ws, nb = self.workspace(abi.OP_CATEGORICAL_INDEX, n)
        self.lib.call("gjx_categorical_index", C.byref(self._keys(kb, 1)), self._chk(logits, torch.float32, n), n,
                      C.c_void_p(out.data_ptr()), mode, C.c_void_p(ws.data_ptr()), nb, self.stream())
        return out

    def categorical_index_batch(self, kbs: list, logits: torch.Tensor, n: int) -> torch.Tensor:
        """gjx_categorical_index_batch: one Gumbel-max draw per row of `logits` [B, stride] (the first n entries of each
        row) under the scalar keys `kbs` — ONE launch; entry b equals categorical_index(kbs[b], logits[b, :n])."""
        B = len(kbs)
        arr = (abi.Keys * B)(*[self._keys(k, 1) for k in kbs])
        out = self.empty(B, torch.int64)
        self.lib.call("gjx_categorical_index_batch", arr, B, C.c_void_p(logits.data_ptr()), int(n), int(logits.stride(0)),
                      C.c_void_p(out.data_ptr()), self.stream())
        return out

    def resample(self, kind: str, kb: KeyBatch, logw: torch.Tensor, n_out: int | None = None):
        """-> (ancestors int32[n_out], anchor, q i64[1]): systematic — tile-anchored weights, anchor = the merged
        power-of-two exponent e int32[1] (lse = e ln 2 + log(q 2^-30)); multinomial — anchor = max f32[1]
        (lse = max + log(q 2^-frac_bits(n)))."""
        n = logw.numel()
        n_out = n if n_out is None else n_out
        anc = self.empty(n_out, torch.int32)
        m, q = self.empty(1, torch.int32 if kind == "systematic" else torch.float32), self.empty(1, torch.int64)
        ws, nb = self.workspace(abi.OP_RESAMPLE, max(n, n_out))
        self.lib.call(f"gjx_resample_{kind}", C.byref(self._keys(kb, 1)), self._chk(logw, torch.float32, n), n,
                      n_out, C.c_void_p(anc.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(q.data_ptr()),
                      C.c_void_p(ws.data_ptr()), nb, self.stream())
        return anc, m, q

    def tile_weights(self, x: torch.Tensor):
        """-> (qw int32[n], recs int64[tiles, 2], subs int64[tiles, 16], ess int64[tiles, 2]): the tile-anchored
        fixed-point weights (u32 bit patterns) and the three dense record arrays (gjx_tile_rec: S_t, e_t in the low half
        of word 1; gjx_tile_sub: 16 sub-prefixes; gjx_tile_ess: ESS sums) of arbitrary log-weights."""
        n, nt = x.numel(), self.num_tiles(x.numel())
        qw, recs = self.empty(n, torch.int32), self.empty((nt, abi.TILE_REC_WORDS), torch.int64)
        subs, ess = self.empty((nt, abi.TILE_SUB_WORDS), torch.int64), self.empty((nt, abi.TILE_ESS_WORDS), torch.int64)
        self.lib.call("gjx_tile_weights", self._chk(x, torch.float32, n), n, self._p(qw), self._p(recs), self._p(subs),
                      self._p(ess), self.stream())
        return qw, recs, subs, ess

    def tile_merge(self, recs: torch.Tensor):
        """-> (e int32[1], q int64[1]): merged anchor and total mass of tile records."""
        e, q = self.empty(1, torch.int32), self.empty(1, torch.int64)
        self.lib.call("gjx_tile_merge", self._p(recs), recs.shape[0], self._p(e), self._p(q), self.stream())
        return e, q

    def gather_cols(self, ancestors: torch.Tensor, cols: list[torch.Tensor]) -> list[torch.Tensor]:
        n_out = ancestors.numel()
        self._chk(ancestors, torch.int32, name="ancestors")
        src = (C.c_void_p * max(1, len(cols)))()
        dst = (C.c_void_p * max(1, len(cols)))()
        outs = []
        for i, c in enumerate(cols):
            if c.element_size() != 4 or c.dim() != 1 or not c.is_contiguous() or c.device.type != self.device_type:
                raise ValueError("gather_cols: columns must be contiguous 1-D 4-byte tensors on the device")
            o = torch.empty(n_out, dtype=c.dtype, device=c.device)
            src[i], dst[i] = c.data_ptr(), o.data_ptr()
            outs.append(o)
        self.lib.call("gjx_gather_cols", C.c_void_p(ancestors.data_ptr()), n_out, src, dst, len(cols), self.stream())
        return outs

    # ---- fused SMC --------------------------------------------------------------------------------
    def _smc_cfg(self, impl, n_total, first, n_local, step_keys, resample_keys, ess_threshold: float = 0.0):
        """`step_keys` / `resample_keys`: [T, 2] for one filter, [F, T, 2] for F filters stepping in the same
        launches (gjx_smc_config.n_filters).  `ess_threshold` in (0, 1): ESS-adaptive resampling — the config then
        carries a device int32[T] / [F, T] `cfg._flags` (1 where a step began with a resampling)."""
        import numpy as np

        sk = np.ascontiguousarray(np.asarray(step_keys, dtype=np.uint32))
        rk = np.ascontiguousarray(np.asarray(resample_keys, dtype=np.uint32))
        F = sk.shape[0] if sk.ndim == 3 else 1
        T = sk.shape[-2]
        cfg = abi.SmcConfig()
        cfg.impl, cfg.n_total, cfg.first_slot, cfg.n_local, cfg.n_steps = impl, n_total, first, n_local, T
        cfg.step_keys, cfg.resample_keys = sk.ctypes.data, rk.ctypes.data
        cfg.n_filters = F if F > 1 else 0
        cfg.filter_stride = self.num_tiles(n_total) * self.tile if F > 1 else 0
        cfg._keep = (sk, rk)  # keep host arrays alive
        cfg._filters = F
        cfg.ess_threshold = float(ess_threshold)
        cfg._adaptive = 0.0 < float(ess_threshold) < 1.0
        cfg._flags = None
        if cfg._adaptive:
            cfg._flags = torch.zeros((F, T) if F > 1 else (T,), dtype=torch.int32, device=self.device())
            cfg.resampled_out = cfg._flags.data_ptr()
        return cfg

    def _smc_buffers(self, cfg, n, state_dtype, want_ancestors):
        """Outputs of a whole-run call: one filter -> [T], [n]; F filters -> [F, T], [F, stride] (views [:, :n])."""
        F, T = cfg._filters, cfg.n_steps
        if F == 1:
            return (self.empty(T, torch.int32), self.empty(T, torch.int64), self.empty(n, state_dtype),
                    self.empty(n, torch.float32), self.empty((T, n), torch.int32) if want_ancestors else None,
                    self.workspace(abi.OP_SMC, n))
        stride = cfg.filter_stride
        ws_one = int(self.lib.call("gjx_workspace_bytes", abi.OP_SMC, n))
        ws = torch.empty(F * ws_one, dtype=torch.uint8, device=self._alloc_device)  # (per call: see workspace())
        return (self.empty((F, T), torch.int32), self.empty((F, T), torch.int64), self.empty((F, stride), state_dtype),
                self.empty((F, stride), torch.float32),
                self.empty((T, F, stride), torch.int32) if want_ancestors else None, (ws, F * ws_one))

    def smc_run_lgssm(self, impl, n, step_keys, resample_keys, model: abi.Lgssm, y, want_ancestors=False,
                      ess_threshold: float = 0.0, want_flags: bool = False):
        """-> (out_e, out_q, state, logw, ancestors[, resampled flags or None])."""
        import numpy as np

        cfg = self._smc_cfg(impl, n, 0, n, step_keys, resample_keys, ess_threshold)
        yh = np.ascontiguousarray(np.asarray(y, dtype=np.float32))
        assert yh.size == cfg.n_steps
        out_e, out_q, state, logw, anc, (ws, nb) = self._smc_buffers(cfg, n, torch.float32, want_ancestors)
        self.lib.call("gjx_smc_run_lgssm", C.byref(cfg), C.byref(model), C.c_void_p(yh.ctypes.data),
                      C.c_void_p(out_e.data_ptr()), C.c_void_p(out_q.data_ptr()), C.c_void_p(state.data_ptr()),
                      C.c_void_p(logw.data_ptr()), C.c_void_p(anc.data_ptr()) if anc is not None else None,
                      C.c_void_p(ws.data_ptr()), nb, self.stream())
        return (out_e, out_q, state, logw, anc, cfg._flags) if want_flags else (out_e, out_q, state, logw, anc)

    def smc_run_hmm(self, impl, n, step_keys, resample_keys, n_states, init_state, trans_logits, obs_logits, y,
                    want_ancestors=False, ess_threshold: float = 0.0, want_flags: bool = False):
        import numpy as np

        cfg = self._smc_cfg(impl, n, 0, n, step_keys, resample_keys, ess_threshold)
        yh = np.ascontiguousarray(np.asarray(y, dtype=np.int32))
        assert yh.size == cfg.n_steps
        mdl = abi.Hmm()
        mdl.n_states, mdl.init_state = n_states, init_state
        mdl.trans_logits = self._chk(trans_logits, torch.float32, n_states * n_states).value
        mdl.obs_logits = self._chk(obs_logits, torch.float32, n_states * n_states).value
        out_e, out_q, state, logw, anc, (ws, nb) = self._smc_buffers(cfg, n, torch.int32, want_ancestors)
        self.lib.call("gjx_smc_run_hmm", C.byref(cfg), C.byref(mdl), C.c_void_p(yh.ctypes.data),
                      C.c_void_p(out_e.data_ptr()), C.c_void_p(out_q.data_ptr()), C.c_void_p(state.data_ptr()),
                      C.c_void_p(logw.data_ptr()), C.c_void_p(anc.data_ptr()) if anc is not None else None,
                      C.c_void_p(ws.data_ptr()), nb, self.stream())
        return (out_e, out_q, state, logw, anc, cfg._flags) if want_flags else (out_e, out_q, state, logw, anc)

    def hmm_model(self, n_states: int, init_state: int, trans_logits: torch.Tensor, obs_logits: torch.Tensor) -> abi.Hmm:
        mdl = abi.Hmm()
        mdl.n_states, mdl.init_state = n_states, init_state
        mdl.trans_logits = self._chk(trans_logits, torch.float32, n_states * n_states).value
        mdl.obs_logits = self._chk(obs_logits, torch.float32, n_states * n_states).value
        mdl._keep = (trans_logits, obs_logits)
        return mdl

    def hmm_prepare(self, n_states: int, init_state: int, trans_logits: torch.Tensor, obs_logits: torch.Tensor):
        """-> (trans_alias int32[K, K] packed alias-table entries, obs_logp f32[K, K]) — the tables of gjx.h."""
        return self.hmm_prepare_model(self.hmm_model(n_states, init_state, trans_logits, obs_logits))

    def hmm_prepare_model(self, mdl: abi.Hmm):
        n_states = mdl.n_states
        words = int(self.lib.call("gjx_hmm_alias_words", n_states))
        cdf = torch.zeros(words, dtype=torch.int32, device=self.device())
        logp = self.empty((n_states, n_states), torch.float32)
        self.lib.call("gjx_hmm_prepare", C.byref(mdl), self._p(cdf), self._p(logp), self.stream())
        return cdf.view(n_states, -1), logp

    # ---- bootstrap SMC for a user model (init + step site tables) --------------------------------
    def smc_plan_create(self, init_sites, step_sites, init_state, next_state, n_obs: int, init_scopes=(), step_scopes=()) -> "SmcPlan":
        m = abi.SmcModel()
        ia = (abi.Site * len(init_sites))(*init_sites)
        sa = (abi.Site * len(step_sites))(*step_sites)
        m.init_sites, m.n_init_sites, m.step_sites, m.n_step_sites = ia, len(init_sites), sa, len(step_sites)
        for k, a in enumerate(init_state):
            m.init_state[k] = a
        for k, a in enumerate(next_state):
            m.next_state[k] = a
        m.n_state, m.n_obs = len(next_state), n_obs
        handle = C.c_void_p()
        if init_scopes or step_scopes:  # nested `@gen` calls inside init / step
            si = (abi.Scope * max(1, len(init_scopes)))(*[abi.Scope(*k) for k in init_scopes])
            ss = (abi.Scope * max(1, len(step_scopes)))(*[abi.Scope(*k) for k in step_scopes])
            self.lib.call("gjx_smc_plan_create_scoped", C.byref(m), si, len(init_scopes), ss, len(step_scopes), C.byref(handle))
        else:
            self.lib.call("gjx_smc_plan_create", C.byref(m), C.byref(handle))
        return SmcPlan(self, handle, len(next_state), n_obs)

    # ---- importance over a Scan model: T steps per particle in one launch ------------------------
    def scan_plan_create(self, step_sites, next_state, n_obs: int, fast_math: bool = False, scopes=()) -> "ScanPlan":
        m = abi.ScanModel()
        sa = (abi.Site * len(step_sites))(*step_sites)
        m.step_sites, m.n_step_sites = sa, len(step_sites)
        for k, a in enumerate(next_state):
            m.next_state[k] = a
        m.n_state, m.n_obs = len(next_state), n_obs
        handle = C.c_void_p()
        flags = abi.PLAN_FAST_MATH if fast_math else 0
        if scopes:  # nested `@gen` calls inside the step kernel
            sc = (abi.Scope * len(scopes))(*[abi.Scope(*k) for k in scopes])
            self.lib.call("gjx_scan_plan_create_scoped", C.byref(m), sc, len(scopes), flags, C.byref(handle))
        else:
            self.lib.call("gjx_scan_plan_create", C.byref(m), flags, C.byref(handle))
        return ScanPlan(self, handle, len(next_state), n_obs)

    def scan_run(self, plan: "ScanPlan", kb: KeyBatch, n: int, T: int, obs, carry0, value_dtypes: list,
                 want_score=True, want_rows=True, out=None):
        """One launch for the whole scan: -> dict(values [T, n] each, score, logw, carry [n] each, max_partials, rows).
        `obs` [T, n_obs] (host array or device tensor); `carry0`: per component a float or a device f32[n] column.
        `out`: a dict from an earlier call whose buffers are reused (benchmarks)."""
        import numpy as np

        if kb.fold is not None:
            raise ValueError("particle keys must not carry a fold")
        D = plan.n_state
        if len(carry0) != D:
            raise ValueError(f"carry0 must have {D} components")
        if plan.n_obs:
            obs_d = obs if isinstance(obs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(obs, dtype=np.float32)))
            obs_d = obs_d.to(device=self.device(), dtype=torch.float32).reshape(T, plan.n_obs).contiguous()
        else:
            obs_d = None
        c0 = np.zeros(D, dtype=np.float32)
        c0cols = (C.c_void_p * D)()
        keep = []
        for k, v in enumerate(carry0):
            if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.numel() > 1:
                keep.append(v.to(device=self.device(), dtype=torch.float32).contiguous())
                c0cols[k] = self._chk(keep[-1], torch.float32, n, f"carry0[{k}]").value
            else:
                c0[k] = float(v)
        if out is None:
            out = dict(values=[self.empty((T, n), dt) for dt in value_dtypes],
                       score=self.empty(n, torch.float32) if want_score else None, logw=self.empty(n, torch.float32),
                       carry=[self.empty(n, torch.float32) for _ in range(D)],
                       max_partials=self.empty(self.num_max_partials(n), torch.float32),
                       rows=RowStats(self.empty(self.num_max_partials(n), torch.int32),
                                     self.empty(self.num_max_partials(n), torch.int64), n) if want_rows else None)
        outs = (C.c_void_p * max(1, len(out["values"])))(*[t.data_ptr() for t in out["values"]])
        couts = (C.c_void_p * D)(*[t.data_ptr() for t in out["carry"]])
        keys = self._keys(kb, n)
        io = abi.ScanIO()
        io.particle_keys = C.cast(C.pointer(keys), C.c_void_p)
        io.n, io.n_steps = n, T
        io.obs = obs_d.data_ptr() if obs_d is not None else None
        io.carry0 = c0.ctypes.data
        io.carry0_cols = C.cast(c0cols, C.c_void_p)
        io.value_cols, io.n_value_cols, io.col_stride = C.cast(outs, C.c_void_p), len(out["values"]), n
        io.carry_out = C.cast(couts, C.c_void_p)
        io.score = out["score"].data_ptr() if out["score"] is not None else None
        io.logw = out["logw"].data_ptr()
        io.max_partials = out["max_partials"].data_ptr()
        if out["rows"] is not None:
            io.row_e, io.row_s = out["rows"].e.data_ptr(), out["rows"].s.data_ptr()
        self.lib.call("gjx_scan_run", plan.handle, C.byref(io), self.stream())
        out["_keep"] = (obs_d, keep)
        return out

    def smc_run_plan(self, plan: "SmcPlan", impl, n, step_keys, resample_keys, obs, want_ancestors=False,
                     ess_threshold: float = 0.0, want_flags: bool = False):
        """`step_keys` / `resample_keys` [T, 2]: one filter -> (e [T], q [T], state columns [n], logw [n], ancestors
        [T, n]); [F, T, 2]: F filters (same observations, own keys) stepping in the same launches -> ([F, T], [F, T],
        columns [F, stride], [F, stride], [T, F, stride]), filter f equal to its own single run bit for bit."""
        import numpy as np

        cfg = self._smc_cfg(impl, n, 0, n, step_keys, resample_keys, ess_threshold)
        T, F = cfg.n_steps, cfg._filters
        oh = np.ascontiguousarray(np.asarray(obs, dtype=np.float32).reshape(T, max(plan.n_obs, 1))[:, :plan.n_obs])
        stride = cfg.filter_stride if F > 1 else n
        shape = (lambda *tail: (F, *tail)) if F > 1 else (lambda *tail: tail)
        out_e, out_q = self.empty(shape(T), torch.int32), self.empty(shape(T), torch.int64)
        states = [self.empty(shape(stride), torch.float32) for _ in range(plan.n_state)]
        sp = (C.c_void_p * plan.n_state)(*[t.data_ptr() for t in states])
        logw = self.empty(shape(stride), torch.float32)
        anc = self.empty((T, F, stride) if F > 1 else (T, n), torch.int32) if want_ancestors else None
        ws, nb = self.workspace(abi.OP_SMC, F * stride)
        self.lib.call("gjx_smc_run_plan", C.byref(cfg), plan.handle, C.c_void_p(oh.ctypes.data) if plan.n_obs else None,
                      self._p(out_e), self._p(out_q), sp, self._p(logw), self._p(anc), C.c_void_p(ws.data_ptr()), nb,
                      self.stream())
        return (out_e, out_q, states, logw, anc, cfg._flags) if want_flags else (out_e, out_q, states, logw, anc)

    # ---- step-level SMC pieces (multi-device driver: dist.py) -----------------------------------
    def smc_config(self, impl, n_total, first, n_local, step_keys, resample_keys, ess_threshold: float = 0.0):
        """Step-level config (one filter)."""
        return self._smc_cfg(impl, n_total, first, n_local, step_keys, resample_keys, ess_threshold)

    @staticmethod
    def _p(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def smc_pop(self, n_total: int, state_dtypes: list, adaptive: bool, want_logw: bool = True) -> "SmcPopulation":
        """A GLOBAL-size population (gjx_smc_pop): state columns, fixed-point weights, log-weights, tile records."""
        return SmcPopulation(self, n_total, state_dtypes, adaptive, want_logw)

    def smc_lgssm_step(self, cfg, model: abi.Lgssm, t: int, y_t: float, prev, out, prev_e_out=None, prev_q_out=None,
                       ancestors_out=None):
        """`prev` / `out`: abi.SmcPop structs (SmcPopulation.struct(): `out` offset to the rank's own block)."""
        self.lib.call("gjx_smc_lgssm_step", C.byref(cfg), C.byref(model), t, float(y_t),
                      C.byref(prev) if prev is not None else None, C.byref(out), self._p(prev_e_out), self._p(prev_q_out),
                      self._p(ancestors_out), self.stream())

    def smc_hmm_step(self, cfg, model: abi.Hmm, t: int, y_t: int, prev, out, trans_alias, obs_logp, prev_e_out=None,
                     prev_q_out=None, ancestors_out=None):
        self.lib.call("gjx_smc_hmm_step", C.byref(cfg), C.byref(model), t, int(y_t),
                      C.byref(prev) if prev is not None else None, C.byref(out), self._p(prev_e_out), self._p(prev_q_out),
                      self._p(trans_alias), self._p(obs_logp), self._p(ancestors_out), self.stream())

    def smc_plan_step(self, cfg, plan: "SmcPlan", t: int, obs_t, prev, out, prev_e_out=None, prev_q_out=None,
                      ancestors_out=None):
        """One step of a plan-driven filter; `obs_t` the step's observation constants."""
        import numpy as np

        oh = np.ascontiguousarray(np.asarray(obs_t, dtype=np.float32).reshape(-1)[:plan.n_obs])
        self.lib.call("gjx_smc_plan_step", C.byref(cfg), plan.handle, t, C.c_void_p(oh.ctypes.data) if plan.n_obs else None,
                      C.byref(prev) if prev is not None else None, C.byref(out), self._p(prev_e_out), self._p(prev_q_out),
                      self._p(ancestors_out), self.stream())

    def smc_records_pack(self, cfg, world: int, unpack: bool, recs, ess, stage):
        """gjx_smc_records_pack: a rank's records + ESS sums <-> its slot of the one-message staging buffer."""
        self.lib.call("gjx_smc_records_pack", C.byref(cfg), int(world), 1 if unpack else 0, self._p(recs), self._p(ess),
                      self._p(stage), self.stream())

    def smc_finish(self, cfg, recs, e_out, q_out):
        self.lib.call("gjx_smc_finish", C.byref(cfg), self._p(recs), self._p(e_out), self._p(q_out), self.stream())

    def smc_source_ranges(self, cfg, recs, ess, world: int, out_ranges, ticket: int = 0):
        """out_ranges int64[2 world + 1]: the source tiles each of `world` equal output blocks can draw from, then
        the ticket (stored last; a pinned host buffer can be polled for it)."""
        self.lib.call("gjx_smc_source_ranges", C.byref(cfg), self._p(recs), self._p(ess), int(world), int(ticket),
                      self._p(out_ranges), self.stream())

    def log_z_from_pairs(self, out_e: torch.Tensor, out_q: torch.Tensor, n_total: int, resampled=None) -> float:
        """log Z = sum_t (e_t ln 2 + log(q_t 2^-30) - log N), evaluated in float64 on the host from the exact
        per-step (merged anchor, fixed-point sum) pairs.  `resampled` (int32[T], ESS-adaptive filters):
        the sum runs over the steps that END an epoch of accumulated weights — a resampling follows (resampled[t + 1]
        == 1) or t = T - 1."""
        e = out_e.detach().cpu().double()
        q = out_q.detach().cpu().double()
        terms = (e - abi.TILE_FRAC) * math.log(2.0) + torch.log(q) - math.log(n_total)
        if resampled is not None:
            r = resampled.detach().cpu().bool()
            ends = torch.ones_like(r)
            ends[:-1] = r[1:]
            terms = terms[ends]
        return float(terms.sum())


class BlockCarver:
    """Carves arrays out of ONE block of memory at 256-byte aligned offsets — the same offsets for the same sequence of
    requests, which is what makes the arenas of the peer transport (gjx_smc_peers) identical in layout on every rank.
    `block` None: a dry run that only counts bytes."""

    def __init__(self, block: "torch.Tensor | None"):
        self.block, self.off = block, 0

    def take(self, shape, dtype) -> "torch.Tensor | None":
        shape = (shape,) if isinstance(shape, int) else tuple(shape)
        nbytes = int(math.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        off = (self.off + 255) // 256 * 256
        self.off = off + nbytes
        if self.block is None:
            return None
        if self.off > self.block.numel():
            raise ValueError("arena block too small")
        return self.block[off:off + nbytes].view(dtype).view(shape)


class SmcPopulation:
    """Device buffers of one population of a step-level (sharded) filter, all GLOBAL size (gjx.h gjx_smc_pop).  `carver`:
    carve them from a caller's block (an arena of the peer transport) instead of allocating."""

    def __init__(self, ops: "Ops", n_total: int, state_dtypes: list, adaptive: bool, want_logw: bool = True,
                 carver: "BlockCarver | None" = None):
        self.ops, self.n, self.adaptive = ops, n_total, adaptive
        nt = ops.num_tiles(n_total)
        if carver is None:
            self.state = [ops.empty(n_total, dt) for dt in state_dtypes]
            self.qw = ops.empty(n_total, torch.int32)
            self.logw = ops.empty(n_total, torch.float32) if (want_logw or adaptive) else None
            self.recs = torch.zeros((nt, abi.TILE_REC_WORDS), dtype=torch.int64, device=ops.device())
            self.subs = torch.zeros((nt, abi.TILE_SUB_WORDS), dtype=torch.int64, device=ops.device())
            self.ess = torch.zeros((nt, abi.TILE_ESS_WORDS), dtype=torch.int64, device=ops.device()) if adaptive else None
            self.prefix = ops.empty(nt + 4, torch.int64) if nt > 1024 else None
        else:
            self.state = [carver.take(n_total, dt) for dt in state_dtypes]
            self.qw = carver.take(n_total, torch.int32)
            self.logw = carver.take(n_total, torch.float32) if (want_logw or adaptive) else None
            self.recs = carver.take((nt, abi.TILE_REC_WORDS), torch.int64)
            self.subs = carver.take((nt, abi.TILE_SUB_WORDS), torch.int64)
            self.ess = carver.take((nt, abi.TILE_ESS_WORDS), torch.int64) if adaptive else None
            self.prefix = carver.take(nt + 4, torch.int64) if nt > 1024 else None

    def columns(self, with_logw: bool | None = None) -> list[torch.Tensor]:
        """The per-particle columns an ancestor shuffle moves: state, fixed-point weights (and log-weights when adaptive)."""
        cols = list(self.state) + [self.qw]
        if self.adaptive if with_logw is None else with_logw:
            cols.append(self.logw)
        return cols

    def struct(self, first: int = 0, with_logw: bool = True) -> abi.SmcPop:
        """abi.SmcPop: per-particle arrays start at slot `first` (a rank's own block for the population a step WRITES;
        0 for the one it READS); the record arrays are always the global ones."""
        p = abi.SmcPop()
        for k, c in enumerate(self.state):
            p.state[k] = c.data_ptr() + first * 4
        p.qw = self.qw.data_ptr() + first * 4
        p.logw = (self.logw.data_ptr() + first * 4) if (self.logw is not None and with_logw) else None
        p.recs = self.recs.data_ptr()
        p.subs = self.subs.data_ptr()
        p.ess = self.ess.data_ptr() if self.ess is not None else None
        p.prefix = self.prefix.data_ptr() if self.prefix is not None else None
        p._keep = self
        return p


class SmcPlan:
    def __init__(self, ops: "Ops", handle, n_state: int, n_obs: int):
        self.ops, self.handle, self.n_state, self.n_obs = ops, handle, n_state, n_obs

    def __del__(self):
        try:
            if self.handle:
                self.ops.lib.call("gjx_smc_plan_destroy", self.handle)
                self.handle = None
        except Exception:
            pass


class ScanPlan:
    def __init__(self, ops: "Ops", handle, n_state: int, n_obs: int):
        self.ops, self.handle, self.n_state, self.n_obs = ops, handle, n_state, n_obs

    def compile_check(self, impl: int) -> int:
        return self.ops.lib._gjx_scan_plan_compile_check(self.handle, impl)

    def __del__(self):
        try:
            if self.handle:
                self.ops.lib.call("gjx_scan_plan_destroy", self.handle)
                self.handle = None
        except Exception:
            pass


@dataclass
class RowStats:
    """Row-anchored partial sums of a log-weight column (one (e, S) pair per 256 particles)."""

    e: torch.Tensor  # int32[rows]
    s: torch.Tensor  # int64[rows]
    n: int
    lse: torch.Tensor | None = None    # f32[1]: log-sum-exp of the rows when the producing launch folded them
    e_out: torch.Tensor | None = None  # int32[1] anchor of that fold
    q_out: torch.Tensor | None = None  # int64[1] fixed-point sum of that fold


class Plan:
    def __init__(self, ops: Ops, handle, n_sites: int):
        self.ops, self.handle, self.n_sites = ops, handle, n_sites
        self.params_owner = None

    def set_params(self, values) -> "Plan":
        """Values of the plan's GJX_ARG_PARAM references for the launches that follow (observations, model arguments:
        one specialised kernel serves every dataset)."""
        import numpy as np

        v = np.ascontiguousarray(np.asarray(values, dtype=np.float32).reshape(-1))
        self.ops.lib.call("gjx_plan_set_params", self.handle, C.c_void_p(v.ctypes.data) if v.size else None, int(v.size))
        self.params_owner = None  # (whoever set them may claim them afterwards: see ImportanceK._fast_estimate)
        return self

    def __del__(self):
        try:
            if self.handle:
                self.ops.lib.call("gjx_plan_destroy", self.handle)
                self.handle = None
        except Exception:
            pass


class PreparedImportance:
    """A fully marshalled `gjx_importance_run` (+ its log-sum-exp) on persistent buffers.

    `fold_batch` B >= 1 keeps B slots of row sums: passes write their (anchor, sum) pairs into consecutive
    slots and ONE `gjx_lse_rows_batch` launch folds them all (`launch_fold`) — the ~5 us latency of a
    one-workgroup kernel is paid once per B passes.
    `kb` may be a LIST of L key batches (L <= 32, lazy, differing in their parent key): `launch_passes`
    then runs L independent passes in ONE launch (`gjx_importance_run_batch`) — a single 1e6-particle pass
    is under two rounds of the machine, L passes keep it full.  Every pass of a launch has its own trace
    buffers (`values[c][p]`, `logw[p]`, `score[p]`); successive launches reuse them."""

    def __init__(self, ops: Ops, plan: Plan, kb, n: int, input_cols, value_dtypes, with_lse=True, fold_batch: int = 1,
                 estimate_only: bool = False):
        kbs = list(kb) if isinstance(kb, (list, tuple)) else [kb]
        if any(k.fold is not None for k in kbs):
            raise ValueError("particle keys must not carry a fold")
        L = self.launch_passes_n = len(kbs)
        if fold_batch % L:
            raise ValueError("fold_batch must be a multiple of the passes per launch")
        self.ops, self.plan, self.n, self.fold_batch = ops, plan, n, fold_batch
        self.inputs = [t for t in input_cols]
        stride = self.pass_stride = -(-n // 256) * 256  # even, and every pass 1 KiB-aligned
        self.values_all = [ops.empty((L, stride), dt) for dt in value_dtypes]
        cols_n = 0 if estimate_only else stride  # (an estimate-only pass writes no per-particle column)
        self.score_all, self.logw_all = ops.empty((L, cols_n), torch.float32), ops.empty((L, cols_n), torch.float32)
        self.values = [v[0, :n] for v in self.values_all]  # pass 0 (the single-pass views)
        self.score, self.logw = self.score_all[0, :n], self.logw_all[0, :n]
        R = self.n_rows = ops.num_max_partials(n)
        self.max_partials_all = ops.empty((L, R if not estimate_only else 0), torch.float32)
        self.max_partials = self.max_partials_all[0]
        self.row_e_all, self.row_s_all = ops.empty((fold_batch, R), torch.int32), ops.empty((fold_batch, R), torch.int64)
        self.rows = RowStats(self.row_e_all[0], self.row_s_all[0], n)  # slot 0
        self.lse_all, self.e_all, self.q_all = (ops.empty(fold_batch, torch.float32), ops.empty(fold_batch, torch.int32),
                                                ops.empty(fold_batch, torch.int64))
        self.lse, self.row_e_out, self.row_q_out = self.lse_all[:1], self.e_all[:1], self.q_all[:1]
        self.max, self.q = ops.empty(1, torch.float32), ops.empty(1, torch.int64)
        self._keys_arr = (abi.Keys * L)(*[ops._keys(k, n) for k in kbs])
        self._keys = self._keys_arr[0]
        for k in self._keys_arr:
            ops.lib.call("gjx_plan_prepare", plan.handle, C.byref(k))  # build the specialised kernel now
        self._ins = (C.c_void_p * max(1, len(self.inputs)))(*[ops._chk(t, torch.float32, n).value for t in self.inputs])
        self._outs = (C.c_void_p * max(1, len(self.values_all)))(*[t.data_ptr() for t in self.values_all])
        self._ws, self._nb = ops.workspace(abi.OP_LOGSUMEXP, n)  # this object's own (persistent) scratch
        lib = ops.lib
        self._run, self._run_batch, self._lse, self._lse_rows, self._fold = (
            lib._gjx_importance_run, lib._gjx_importance_run_batch, lib._gjx_logsumexp_f32, lib._gjx_lse_rows,
            lib._gjx_lse_rows_batch)
        head = (plan.handle, C.byref(self._keys), self._ins, len(self.inputs), self._outs, len(self.values_all),
                C.c_void_p(self.score_all.data_ptr()), C.c_void_p(self.logw_all.data_ptr()), n,
                C.c_void_p(self.max_partials_all.data_ptr()))
        if estimate_only:  # only the row sums leave the kernel (a plan without value columns): no score / logw / maxima
            if value_dtypes:
                raise ValueError("an estimate-only pass has no value columns")
            head = head[:6] + (None, None, n, None)
        self._args_run = [head + (C.c_void_p(self.row_e_all[b].data_ptr()), C.c_void_p(self.row_s_all[b].data_ptr()))
                          for b in range(fold_batch)]
        self._batch_head = (plan.handle, self._keys_arr)
        self._batch_mid = (stride, R, self._ins, len(self.inputs), self._outs, len(self.values_all),
                           C.c_void_p(self.score_all.data_ptr()), C.c_void_p(self.logw_all.data_ptr()), n,
                           C.c_void_p(self.max_partials_all.data_ptr()))
        self._batch_rows = [(C.c_void_p(self.row_e_all[b].data_ptr()), C.c_void_p(self.row_s_all[b].data_ptr()))
                            for b in range(fold_batch)]
        # fused form: the importance launch folds its own row sums (results in .lse / .row_e_out / .row_q_out)
        self._tickets = torch.zeros(abi.LSE_TICKET_WORDS, dtype=torch.int32, device=ops.device())
        self._lse_out = abi.LseOut(self.row_e_out.data_ptr(), self.row_q_out.data_ptr(), self.lse.data_ptr(), None,
                                   self._tickets.data_ptr(), None, 0.0)
        self._lse_ref = C.byref(self._lse_out)
        self._lse_with_record: dict = {}
        self._args_lse_rows = (C.c_void_p(self.rows.e.data_ptr()), C.c_void_p(self.rows.s.data_ptr()), R,
                               C.c_void_p(self.row_e_out.data_ptr()), C.c_void_p(self.row_q_out.data_ptr()),
                               C.c_void_p(self.lse.data_ptr()))
        self._args_fold = (C.c_void_p(self.row_e_all.data_ptr()), C.c_void_p(self.row_s_all.data_ptr()), R)
        self._fold_outs = (C.c_void_p(self.e_all.data_ptr()), C.c_void_p(self.q_all.data_ptr()),
                           C.c_void_p(self.lse_all.data_ptr()))
        self._no_record = C.c_void_p(None)
        self._args_lse = (C.c_void_p(self.logw.data_ptr()), n, C.c_void_p(self.max_partials.data_ptr()),
                          C.c_void_p(self.lse.data_ptr()), C.c_void_p(self.max.data_ptr()), C.c_void_p(self.q.data_ptr()),
                          C.c_void_p(self._ws.data_ptr()), self._nb)
        self.with_lse = with_lse

    def launch_importance(self, stream=None, slot: int = 0):
        """One pass (key batch 0); its row sums go to slot `slot` (folded later by launch_lse_rows / launch_fold)."""
        rc = self._run(*self._args_run[slot], None, stream if stream is not None else self.ops.stream())
        if rc:
            raise abi.GjxError("gjx_importance_run", rc)

    def launch_passes(self, slot0: int = 0, count: int | None = None, stream=None):
        """`count` (default: all L) independent passes in ONE launch; pass p's row sums go to slot slot0 + p."""
        count = self.launch_passes_n if count is None else count
        rc = self._run_batch(*self._batch_head, count, *self._batch_mid, *self._batch_rows[slot0],
                             stream if stream is not None else self.ops.stream())
        if rc:
            raise abi.GjxError("gjx_importance_run_batch", rc)

    def launch_fold(self, count: int, stream=None, records_ptr=None):
        """Fold the row sums of slots [0, count) in ONE launch: lse_all / e_all / q_all[:count]; `records_ptr`
        (C.c_void_p of a dev int64[count, 65]) also receives each pass's exchangeable record."""
        rc = self._fold(*self._args_fold, count, self.n_rows, *self._fold_outs,
                        self._no_record if records_ptr is None else records_ptr,
                        stream if stream is not None else self.ops.stream())
        if rc:
            raise abi.GjxError("gjx_lse_rows_batch", rc)

    def launch_fused(self, stream=None, record: torch.Tensor | None = None):
        """Walk + log-sum-exp in ONE launch: .lse, .row_e_out, .row_q_out (and `record`, a dev int64[65]
        slot, when given) are written by the workgroup that finishes last."""
        if record is None:
            ref = self._lse_ref
        else:
            ent = self._lse_with_record.get(record.data_ptr())
            if ent is None:
                out = abi.LseOut(self.row_e_out.data_ptr(), self.row_q_out.data_ptr(), self.lse.data_ptr(),
                                 record.data_ptr(), self._tickets.data_ptr(), None, 0.0)
                ent = self._lse_with_record[record.data_ptr()] = (out, C.byref(out), record)
            ref = ent[1]
        rc = self._run(*self._args_run[0], ref, stream if stream is not None else self.ops.stream())
        if rc:
            raise abi.GjxError("gjx_importance_run", rc)

    def launch_fused_shifted(self, out: torch.Tensor, shift: float, stream=None):
        """Walk + log-sum-exp in ONE launch, `out[0] = lse - shift` (f32): with shift = log K the log-marginal estimate
        logsumexp(lw) - log K of inference/smc.py:96-97 lands in a tensor of the caller's, no further kernel."""
        lo = self.__dict__.get("_lse_shifted")
        if lo is None:
            lo = self._lse_shifted = abi.LseOut(self.row_e_out.data_ptr(), self.row_q_out.data_ptr(), self.lse.data_ptr(), None,
                                                self._tickets.data_ptr(), None, 0.0)
            self._lse_shifted_ref = C.byref(lo)
        lo.lse_shifted, lo.shift = out.data_ptr(), shift
        rc = self._run(*self._args_run[0], self._lse_shifted_ref, stream if stream is not None else self.ops.stream())
        if rc:
            raise abi.GjxError("gjx_importance_run", rc)

    def launch_estimate(self, k0: int, k1: int, lane: int, out: torch.Tensor, shift: float):
        """gjx_importance_estimate: the reference's key derivation from the caller's scalar key, the estimate-only walk, the fold
        and `out[()] = lse - shift` as ONE library call (one launch) on the current stream."""
        io = self.__dict__.get("_est_io")
        if io is None:
            io = self._est_io = abi.EstimateIO(self.plan.handle.value, self.n, C.addressof(self._ins), len(self.inputs), self._keys.impl,
                                               self.rows.e.data_ptr(), self.rows.s.data_ptr(),
                                               abi.LseOut(self.row_e_out.data_ptr(), self.row_q_out.data_ptr(), self.lse.data_ptr(), None,
                                                          self._tickets.data_ptr(), None, 0.0))
            self._est_ref, self._est_fn = C.byref(io), self.ops.lib._gjx_importance_estimate
            self._est_cuda = self.ops.device_type == "cuda"
        stream = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()) if self._est_cuda else 0
        rc = self._est_fn(self._est_ref, k0, k1, lane, out.data_ptr(), shift, stream)
        if rc:
            raise abi.GjxError("gjx_importance_estimate", rc)

    def launch_lse(self, stream=None):
        rc = self._lse(*self._args_lse, stream if stream is not None else self.ops.stream())
        if rc:
            raise abi.GjxError("gjx_logsumexp_f32", rc)

    def launch_lse_rows(self, stream=None, record_ptr=None):
        """Row-anchored log-sum-exp of slot 0 from the partial sums the importance kernel emitted: one tiny
        kernel, no pass over logw.  Results in .lse, .row_e_out, .row_q_out; `record_ptr`
        (C.c_void_p of a dev int64[65]) also receives the shard's exchangeable record."""
        rc = self._lse_rows(*self._args_lse_rows, self._no_record if record_ptr is None else record_ptr,
                            stream if stream is not None else self.ops.stream())
        if rc:
            raise abi.GjxError("gjx_lse_rows", rc)

    def launch(self, stream=None):
        """One pass (+ the fold of its row sums, a second one-workgroup launch, when with_lse)."""
        st = stream if stream is not None else self.ops.stream()
        self.launch_importance(st)
        if self.with_lse:
            self.launch_lse_rows(st)


class _Hip:
    """The few HIP runtime calls the pipeline needs, bound through ctypes on the runtime torch has
    already loaded (torch.cuda.Event costs ~5 us of Python per call; these cost ~1)."""

    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            import os

            path = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
            lib = C.CDLL(path)
            lib.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            lib.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
            lib.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
            lib.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
            lib.hipEventSynchronize.argtypes = [C.c_void_p]
            lib.hipEventDestroy.argtypes = [C.c_void_p]
            cls._lib = lib
        return cls._lib


class HipEvent:
    """A raw hipEvent_t (timing enabled unless `timing=False`)."""

    def __init__(self, timing: bool = True):
        self.h = C.c_void_p()
        rc = _Hip.lib().hipEventCreateWithFlags(C.byref(self.h), 0 if timing else 2)  # 2 = hipEventDisableTiming
        if rc:
            raise RuntimeError(f"hipEventCreateWithFlags: {rc}")

    def record(self, stream_handle):
        _Hip.lib().hipEventRecord(self.h, stream_handle)

    def elapsed_ms(self, end: "HipEvent") -> float:
        out = C.c_float()
        _Hip.lib().hipEventSynchronize(end.h)
        rc = _Hip.lib().hipEventElapsedTime(C.byref(out), self.h, end.h)
        if rc:
            raise RuntimeError(f"hipEventElapsedTime: {rc}")
        return out.value

    def __del__(self):
        try:
            _Hip.lib().hipEventDestroy(self.h)
        except Exception:
            pass
