"""Multi-device execution: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on the GPU box; "gloo" in the CPU tests).  The reference has no distributed layer at all;
this is a new design around two properties of the single-device path:

  * keys are counter-based and indexed by GLOBAL particle slot, and
  * weight sums are exact integers,

so a population sharded over G ranks produces bit-identical particles, ancestors and log-Z for
every G — the exchange is pure data movement, never a change of arithmetic.

ImportanceK: rank r runs slots [r n, (r+1) n) with no communication; the global log-normaliser needs
one all-reduce(max) of a float and one all-reduce(sum) of an int64 (exact, order-independent).

Bootstrap SMC: resampling is global.  Per step each rank (1) resamples + propagates + weights its own
output slots reading the GLOBAL previous population, (2) all-reduce(max) of the per-tile maxima,
(3) computes its tiles' fixed-point masses, (4) all-gathers the new particles, weights and tile masses.
All-gathers over 7 point-to-point xGMI links use every link at once; their volume (8 B per particle
per step per rank) is what bounds weak scaling — see DESIGN.md §6 for the direct peer-read design
that replaces step (4) with reads of only the ancestor ranges actually needed.
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import abi, prng, workloads as W
from .ops import Ops


def _dist():
    import torch.distributed as dist

    return dist


def importance_log_z(ops: Ops, wl: "W.Gaussian10", prep=None):
    """Sharded ImportanceK pass -> (log_z float64, local logw).  `wl` was built with first / n_total."""
    dist = _dist()
    if prep is None:
        prep = wl.prepare()
    prep.launch_importance()
    m = ops.max_f32(None, wl.n, max_partials=prep.max_partials)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    q = ops.expsum_fix(prep.logw, m, wl.frac)
    dist.all_reduce(q, op=dist.ReduceOp.SUM)
    log_z = float(m.cpu()) + math.log(int(q.cpu())) - wl.frac * math.log(2.0) - math.log(wl.n_total)
    return log_z, prep.logw, m, q


class BatchedImportance:
    """Sharded ImportanceK passes with *bucketed* collectives: B independent passes are issued back to
    back, their B local maxima are all-reduced as ONE message, then the B fixed-point sums as one more.
    A small-message RCCL all-reduce costs tens of microseconds of latency (and as much host time to
    enqueue), more than the 29 us kernel; bucketing B passes amortises it B-fold — the xGMI analogue
    of gradient bucketing.  Each pass keeps its own output buffers (B x 48 MB at 1e6 particles)."""

    def __init__(self, ops: Ops, make_workload, batch: int = 8):
        self.ops, self.batch = ops, batch
        dev = ops.device()
        self.m_all = torch.empty(batch, dtype=torch.float32, device=dev)
        self.q_all = torch.empty(batch, dtype=torch.int64, device=dev)
        self.preps = []
        self.wl = None
        for _ in range(batch):
            wl = make_workload()
            self.wl = self.wl or wl
            self.preps.append((wl, ops.prepare_importance(wl.plan, wl.keys, wl.n, [], [torch.float32] * W.G10_LATENTS,
                                                          with_lse=False)))
        self.m_views = [self.m_all[b:b + 1] for b in range(batch)]
        self.q_views = [self.q_all[b:b + 1] for b in range(batch)]

    def run(self, count: int | None = None, on_kernel=None):
        """`count` (<= batch) passes; afterwards m_all[:count], q_all[:count] hold the global pairs."""
        dist, ops = _dist(), self.ops
        count = self.batch if count is None else count
        for b in range(count):
            wl, prep = self.preps[b]
            if on_kernel:
                on_kernel(b, 0)
            prep.launch_importance()
            if on_kernel:
                on_kernel(b, 1)
            ops.max_f32(None, wl.n, max_partials=prep.max_partials, out=self.m_views[b])
        dist.all_reduce(self.m_all[:count], op=dist.ReduceOp.MAX)
        for b in range(count):
            wl, prep = self.preps[b]
            ops.expsum_fix(prep.logw, self.m_views[b], wl.frac, out=self.q_views[b])
        dist.all_reduce(self.q_all[:count], op=dist.ReduceOp.SUM)

    def log_z(self, b: int = 0) -> float:
        return (float(self.m_all[b].cpu()) + math.log(int(self.q_all[b].cpu())) - self.wl.frac * math.log(2.0)
                - math.log(self.wl.n_total))


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
