"""Multi-device execution: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on the GPU box; "gloo" in the CPU tests).  The reference has no distributed layer at all;
this is a new design around two properties of the single-device path:

  * keys are counter-based and indexed by GLOBAL particle slot, and
  * weight sums are exact integers,

so a population sharded over G ranks produces bit-identical particles, ancestors and log-Z for
every G — the exchange is pure data movement, never a change of arithmetic.

ImportanceK: rank r runs a row-aligned block of slots with no data-path communication.  The importance
kernel emits row-anchored partial sums of its own log-weights; one tiny kernel folds them into a
65-word record (anchor exponent + 64 exact integer buckets) and ONE all-gather of that
record per pass yields the global log-normaliser — no second pass over the log-weights, no all-reduce
pair, and the merged result is bit-identical for every number of ranks.

Bootstrap SMC: resampling is global.  Per step each rank (1) resamples + propagates + weights its own
output slots reading the GLOBAL previous population, (2) all-reduce(max) of the per-tile maxima,
(3) computes its tiles' fixed-point masses, (4) all-gathers the new particles, weights and tile masses.
All-gathers over 7 point-to-point xGMI links use every link at once; their volume (8 B per particle
per step per rank) is what bounds weak scaling — see DESIGN.md §6 for the direct peer-read design
that replaces step (4) with reads of only the ancestor ranges actually needed.
"""

from __future__ import annotations

import torch

from . import abi, prng, workloads as W
from .ops import Ops


def _dist():
    import torch.distributed as dist

    return dist


ROW = 256  # particles per row of the row-anchored partial sums (gjx_num_max_partials)


def shard_rows(n_total: int, rank: int, world: int) -> tuple[int, int]:
    """(first slot, slot count) of rank's block when the population is split at row boundaries (what
    keeps the row-anchored log-normaliser independent of the number of ranks)."""
    rows = -(-n_total // ROW)
    r0, r1 = rows * rank // world, rows * (rank + 1) // world
    first = r0 * ROW
    return first, min(r1 * ROW, n_total) - first


class BatchedImportance:
    """Sharded ImportanceK passes with *bucketed, overlapped* collectives.

    A pass = the importance kernel, which leaves its row sums in slot b; one `gjx_lse_rows_batch` launch
    per batch folds them into the shard's 65-word records, a [batch, 65] block.  After `batch` passes the block is all-gathered ONCE, asynchronously (RCCL runs on
    its own stream), while the next batch already computes into the other block of a double buffer; the
    wait is deferred until that block is reused or its results are read.  A small-message RCCL collective
    costs tens of microseconds — more than the 28 us kernel — so this is the xGMI analogue of gradient
    bucketing with compute/communication overlap: per pass the exchange costs 1/batch of one collective
    and none of the device's time.  Passes reuse one set of trace buffers (48 MB at 1e6 particles)."""

    def __init__(self, ops: Ops, wl, batch: int = 8, world: int | None = None, depth: int = 2,
                 always_exchange: bool = False, passes: int = 1):
        dist = _dist()
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        if self.world > 1 and (wl.first % ROW or (wl.n % ROW and wl.first + wl.n != wl.n_total)):
            raise ValueError("shards must be split at 256-particle row boundaries (dist.shard_rows)")
        self.ops, self.wl, self.batch, self.depth = ops, wl, batch, depth
        self.passes = passes  # independent passes per importance launch (gjx_importance_run_batch)
        self.prep = wl.prepare(fold_batch=batch, passes=passes)
        dev, words = ops.device(), abi.LSE_RECORD_WORDS
        self.local = [torch.zeros((batch, words), dtype=torch.int64, device=dev) for _ in range(depth)]
        self.exchange = self.world > 1 or always_exchange  # a one-rank group still goes through the collective
        self.gathered = [torch.zeros((self.world, batch, words), dtype=torch.int64, device=dev) if self.exchange
                         else t.view(1, batch, words) for t in self.local]
        import ctypes as C

        self._rec = [C.c_void_p(t.data_ptr()) for t in self.local]
        self._work = [None] * depth
        self._count = [0] * depth
        self._slot = 0

    def run(self, count: int | None = None, on_launch=None) -> int:
        """Enqueue `count` (<= batch) passes and the exchange of their records; returns the buffer
        index to hand to `results`.  `on_launch(phase, n_passes)` brackets every importance launch."""
        count = self.batch if count is None else count
        d, self._slot = self._slot, (self._slot + 1) % self.depth
        self.wait(d)  # the block's previous exchange must have read it before it is overwritten
        st, done = self.ops.stream(), 0
        while done < count:
            c = min(self.passes, count - done)
            if on_launch:
                on_launch(0, c)
            if self.passes == 1:
                self.prep.launch_importance(st, done)  # row sums into slot `done`
            else:
                self.prep.launch_passes(done, c, st)  # c passes, row sums into slots done .. done + c - 1
            if on_launch:
                on_launch(1, c)
            done += c
        self.prep.launch_fold(count, st, self._rec[d])  # one launch folds the batch into its records
        if self.exchange:
            self._work[d] = _dist().all_gather_into_tensor(self.gathered[d].view(-1, self.gathered[d].shape[-1]),
                                                            self.local[d], async_op=True)
        self._count[d] = count
        return d

    def wait(self, d: int | None = None):
        for i in (range(self.depth) if d is None else (d,)):
            if self._work[i] is not None:
                self._work[i].wait()
                self._work[i] = None

    def results(self, d: int):
        """(lse f32[count], e i32[count], q i64[count]) of buffer d's passes (device tensors)."""
        self.wait(d)
        return self.ops.lse_combine(self.gathered[d], self._count[d])

    def log_z(self, d: int, b: int = 0) -> float:
        _, e, q = self.results(d)
        return self.ops.log_z_from_rows(e[b], q[b], self.wl.n_total)


def importance_log_z(ops: Ops, wl: "W.Gaussian10"):
    """One sharded ImportanceK pass -> (log_z float64, local logw, e, q).  `wl` was built with the
    rank's (first, n_local) from `shard_rows` and the global n_total."""
    pipe = BatchedImportance(ops, wl, batch=1, depth=1)
    d = pipe.run(1)
    _, e, q = pipe.results(d)
    return ops.log_z_from_rows(e[0], q[0], wl.n_total), pipe.prep.logw, e, q


class ShardedLgssmSMC:
    """Bootstrap SMC on the linear-Gaussian model with the population sharded over ranks."""

    def __init__(self, ops: Ops, impl: int, seed: int, n_total: int, T: int, rank: int, world: int,
                 record_ancestors: bool = False):
        tile = ops.tile
        if n_total % (world * tile) != 0:
            raise ValueError(f"n_total must be a multiple of world*{tile}")
        self.ops, self.impl, self.T, self.rank, self.world = ops, impl, T, rank, world
        self.n_total, self.n_local = n_total, n_total // world
        self.first = rank * self.n_local
        self.y = W.lgssm_data(T)
        sk, rk = W.smc_key_schedule(prng.key(seed, impl), T)
        self.cfg = ops.smc_config(impl, n_total, self.first, self.n_local, sk, rk)
        self.model = W.lgssm_model()
        dev = ops.device()
        nt = ops.num_tiles(n_total)
        self.state = [torch.empty(n_total, dtype=torch.float32, device=dev) for _ in range(2)]
        self.logw = [torch.empty(n_total, dtype=torch.float32, device=dev) for _ in range(2)]
        self.tile_sums = torch.zeros(nt, dtype=torch.int64, device=dev)
        self.max_partials = torch.empty(nt, dtype=torch.float32, device=dev)
        self.out_max = torch.empty(T, dtype=torch.float32, device=dev)
        self.out_q = torch.zeros(T, dtype=torch.int64, device=dev)
        self.ancestors = torch.empty((T, self.n_local), dtype=torch.int32, device=dev) if record_ancestors else None

    def _gather(self, full: torch.Tensor, lo: int, hi: int):
        dist = _dist()
        if self.world == 1:
            return
        local = full[lo:hi]
        if full.device.type != "cuda":
            local = local.clone()  # gloo: keep input and output distinct
        dist.all_gather_into_tensor(full, local)

    def run(self):
        ops, dist = self.ops, _dist()
        lo, hi = self.first, self.first + self.n_local
        tl, th = lo // ops.tile, hi // ops.tile
        for t in range(self.T):
            cur, prv = t & 1, (t & 1) ^ 1
            ops.smc_lgssm_step_a(
                self.cfg, self.model, t, float(self.y[t]),
                self.state[prv] if t else None, self.logw[prv] if t else None,
                self.out_max[t - 1:t] if t else None, self.tile_sums if t else None,
                self.out_q[t - 1:t] if t else None,
                self.state[cur][lo:hi], self.logw[cur][lo:hi], self.max_partials,
                None if self.ancestors is None else self.ancestors[t])
            if self.world > 1:
                dist.all_reduce(self.max_partials, op=dist.ReduceOp.MAX)
            ops.smc_step_b(self.cfg, self.logw[cur][lo:hi], self.max_partials, self.out_max[t:t + 1], self.tile_sums)
            self._gather(self.tile_sums, tl, th)
            self._gather(self.state[cur], lo, hi)
            self._gather(self.logw[cur], lo, hi)
        ops.smc_finish(self.cfg, self.tile_sums, self.out_q[self.T - 1:self.T])
        last = (self.T - 1) & 1
        return dict(out_max=self.out_max, out_q=self.out_q, state=self.state[last][lo:hi],
                    logw=self.logw[last][lo:hi], ancestors=self.ancestors,
                    log_z=ops.log_z_from_pairs(self.out_max, self.out_q, self.n_total),
                    log_z_exact=W.lgssm_exact_log_z(self.y))
