"""Bootstrap SMC on the fused state-space kernels (`gjx_smc_run_lgssm` / `gjx_smc_run_hmm`).

The reference's SMC module has no resampling step or SMC loop; the north star
asks for bootstrap SMC with systematic resampling and an ancestor gather.  This is that driver:
one call enqueues the whole T-step filter — per step one fused resample+gather+propagate+weight
kernel and one tile-sum kernel — with no host synchronisation.  The model classes are the
fixed-structure equivalents of the `@scan`/`@gen` kernels

    x' = normal(a * x, q) @ "x";  normal(x', r) @ "y"                  (LinearGaussianSSM)
    z' = categorical(T[z, :]) @ "z";  categorical(O[z', :]) @ "x"      (DiscreteHMM, exact_testbed.py:62-68)
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from . import abi, prng
from .choicemap import ChoiceMap
from .runtime import get_ops
from .smc_plan import StateSpaceModel, build_smc_plan, observation_matrix


@dataclass(frozen=True)
class LinearGaussianSSM:
    x0_loc: float = 0.0
    x0_scale: float = 1.0
    a: float = 0.9
    q: float = 1.0
    r: float = 0.5


@dataclass(frozen=True)
class DiscreteHMM:
    trans_logits: torch.Tensor  # [K, K], row = previous state
    obs_logits: torch.Tensor  # [K, K], row = state
    init_state: int = 0


@dataclass
class SMCResult:
    log_marginal_likelihood: float  # float64 from the exact per-step (max, fixed-point sum) pairs
    step_e: torch.Tensor  # int32[T]: per-step merged anchor e_t (lse_t = e_t ln 2 + log(q_t 2^-30))
    step_q: torch.Tensor  # i64[T]
    particles: torch.Tensor  # final-step particles [n] (a tuple of columns for a multi-component carry)
    log_weights: torch.Tensor  # their incremental log-weights [n]
    ancestors: torch.Tensor | None  # int32[T, n] (row 0 = identity)
    resampled: torch.Tensor | None = None  # int32[T] (ESS-adaptive filters): 1 where a step began with a resampling

    def get_log_marginal_likelihood_estimate(self) -> float:
        return self.log_marginal_likelihood


def smc_key_schedule(key: prng.PRNGKey, T: int):
    """step t propagates with fold_in(key, 2t) and resamples with fold_in(key, 2t+1) (fresh lane-0 keys;
    for threefry the same words as split(key, 2T)[2t], [2t+1])."""
    w = prng.fold_words(key, 2 * T)
    return w[0::2].copy(), w[1::2].copy()


class BootstrapSMC:
    """Bootstrap particle filter with systematic resampling: at every step (default), or — `ess_threshold` in (0, 1) —
    only when the effective sample size of the current weights falls below `ess_threshold * n_particles`; between
    resamplings the log-weights accumulate (gjx.h: gjx_smc_config.ess_threshold)."""

    def __init__(self, model, observations, n_particles: int, record_ancestors: bool = False, ess_threshold: float = 0.0):
        self.model, self.n, self.record_ancestors = model, int(n_particles), record_ancestors
        self.ess_threshold = float(ess_threshold)
        self._plan = None
        if isinstance(model, StateSpaceModel):
            if not isinstance(observations, ChoiceMap):
                raise TypeError("observations for a StateSpaceModel are a ChoiceMap of length-T sequences")
            self._obs_chm = observations
            self.observations = None
        else:
            self.observations = np.asarray(observations)

    def get_num_particles(self):
        return self.n

    def run(self, key: prng.PRNGKey) -> SMCResult:
        ops = get_ops()
        if self.observations is not None:
            T = len(self.observations)
            sk, rk = smc_key_schedule(key, T)
        if isinstance(self.model, LinearGaussianSSM):
            m = self.model
            out = ops.smc_run_lgssm(key.impl, self.n, sk, rk, abi.Lgssm(m.x0_loc, m.x0_scale, m.a, m.q, m.r),
                                    self.observations.astype(np.float32), self.record_ancestors,
                                    ess_threshold=self.ess_threshold, want_flags=True)
        elif isinstance(self.model, DiscreteHMM):
            m = self.model
            dev = ops.device()
            tl = torch.as_tensor(m.trans_logits, dtype=torch.float32).to(dev).contiguous()
            ol = torch.as_tensor(m.obs_logits, dtype=torch.float32).to(dev).contiguous()
            out = ops.smc_run_hmm(key.impl, self.n, sk, rk, int(tl.shape[0]), int(m.init_state), tl, ol,
                                  self.observations.astype(np.int32), self.record_ancestors,
                                  ess_threshold=self.ess_threshold, want_flags=True)
        elif isinstance(self.model, StateSpaceModel):
            if self._plan is None:
                self._obs_addrs = [a for a, _ in self._obs_chm.leaves()]
                self._plan, self._n_state = build_smc_plan(self.model, self._obs_addrs)
                self._obs = observation_matrix(self._obs_chm, self._obs_addrs)
            T = self._obs.shape[0]
            sk, rk = smc_key_schedule(key, T)
            om, oq, states, logw, anc, fl = ops.smc_run_plan(self._plan, key.impl, self.n, sk, rk, self._obs,
                                                             self.record_ancestors, ess_threshold=self.ess_threshold,
                                                             want_flags=True)
            out = (om, oq, states[0] if self._n_state == 1 else tuple(states), logw, anc, fl)
        else:
            raise TypeError(f"no fused SMC kernel for {type(self.model).__name__}")
        step_e, step_q, state, logw, anc, flags = out
        return SMCResult(ops.log_z_from_pairs(step_e, step_q, self.n, flags), step_e, step_q, state, logw, anc, flags)

    def run_many(self, keys) -> list:
        """`vmap(self.run)(keys)`: one independent filter per key.  Up to 16 filters step in the same kernel launches
        (`gjx_smc_config.n_filters`: a 1e6-particle step alone is under one round of an MI355X), for the hand-written
        models and for generated ones alike; element b equals `self.run(keys[b])` bit for bit."""
        keys = list(keys)
        if not isinstance(self.model, (LinearGaussianSSM, DiscreteHMM, StateSpaceModel)) or len(keys) < 2:
            return [self.run(k) for k in keys]
        ops, out = get_ops(), []
        if isinstance(self.model, StateSpaceModel) and self._plan is None:
            self.run(keys[0])  # builds the plan and the observation matrix
        T = len(self.observations) if self.observations is not None else self._obs.shape[0]
        ess = dict(ess_threshold=self.ess_threshold, want_flags=True)
        for lo in range(0, len(keys), 16):
            chunk = keys[lo:lo + 16]
            if len(chunk) == 1:
                out.append(self.run(chunk[0]))
                continue
            try:
                out.extend(self._run_chunk(ops, chunk, T, ess))
            except abi.GjxError as e:
                # populations too large for a filter batch (more than 2048 tiles per filter, or a workspace the
                # device cannot hold): the documented contract is "element b equals self.run(keys[b])" — run them so
                if e.code not in (-2, -3):  # GJX_ERR_UNSUPPORTED, GJX_ERR_WORKSPACE
                    raise
                out.extend(self.run(k) for k in chunk)
        return out

    def _run_chunk(self, ops, chunk, T, ess) -> list:
        out = []
        pairs = [smc_key_schedule(k, T) for k in chunk]
        sk, rk = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
        m, impl = self.model, chunk[0].impl
        if isinstance(m, StateSpaceModel):
            om, oq, states, logw, anc, fl = ops.smc_run_plan(self._plan, impl, self.n, sk, rk, self._obs,
                                                             self.record_ancestors, **ess)
            for f in range(len(chunk)):
                cols = [c[f, :self.n] for c in states]
                ff = None if fl is None else fl[f]
                out.append(SMCResult(ops.log_z_from_pairs(om[f], oq[f], self.n, ff), om[f], oq[f],
                                     cols[0] if self._n_state == 1 else tuple(cols), logw[f, :self.n],
                                     None if anc is None else anc[:, f, :self.n], ff))
            return out
        if isinstance(m, LinearGaussianSSM):
            res = ops.smc_run_lgssm(impl, self.n, sk, rk, abi.Lgssm(m.x0_loc, m.x0_scale, m.a, m.q, m.r),
                                    self.observations.astype(np.float32), self.record_ancestors, **ess)
        else:
            dev = ops.device()
            tl = torch.as_tensor(m.trans_logits, dtype=torch.float32).to(dev).contiguous()
            ol = torch.as_tensor(m.obs_logits, dtype=torch.float32).to(dev).contiguous()
            res = ops.smc_run_hmm(impl, self.n, sk, rk, int(tl.shape[0]), int(m.init_state), tl, ol,
                                  self.observations.astype(np.int32), self.record_ancestors, **ess)
        step_e, step_q, state, logw, anc, fl = res
        for f in range(len(chunk)):
            ff = None if fl is None else fl[f]
            out.append(SMCResult(ops.log_z_from_pairs(step_e[f], step_q[f], self.n, ff), step_e[f], step_q[f],
                                 state[f, :self.n], logw[f, :self.n], None if anc is None else anc[:, f, :self.n], ff))
        return out

    def log_marginal_likelihood_estimate(self, key: prng.PRNGKey) -> float:
        return self.run(key).log_marginal_likelihood
