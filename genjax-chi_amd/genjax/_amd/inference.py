"""`Target`, `Algorithm`, `Marginal` and the SMC algorithm family on vectorised traces.

Mirrors inference/sp.py:52-273 and inference/smc.py:56-465 of the reference: same constructors,
methods, key-derivation order (SURVEY §3.5 including the key-reuse quirks) and error behaviour.
What differs is the execution model: where the reference `vmap`s a per-particle function, a key
batch of K keys drives ONE run of the model over K-wide device columns; `logsumexp`, the
categorical draw of `sample_particle`, resampling and ancestor gathers are C-ABI kernels.
"""

from __future__ import annotations

import math
import threading

import torch

from . import abi, prng
from .choicemap import ChoiceMap, Selection
from .lang import Distribution, DistributionTrace, GenerativeFunction, ParticleKeys, Trace, split
from .ops import KeyBatch
from .runtime import fast_math_enabled, get_ops


# =================================================================================================
# Target  (inference/sp.py:52-94)
# =================================================================================================
class Target:
    """Unnormalised posterior: a generative function, its arguments and a constraint."""

    def __init__(self, p: GenerativeFunction, args: tuple, constraint: ChoiceMap):
        if isinstance(p, Marginal):  # sp.py:46-49,79
            raise TypeError("Target does not support Marginal generative functions.")
        if not isinstance(p, GenerativeFunction):
            raise TypeError(f"Target: p must be a GenerativeFunction, got {type(p).__name__}")
        if not isinstance(args, tuple):
            raise TypeError("Target: args must be a tuple")
        if not isinstance(constraint, ChoiceMap):
            raise TypeError("Target: constraint must be a ChoiceMap")
        self.p, self.args, self.constraint = p, args, constraint

    def importance(self, key, constraint: ChoiceMap):
        merged = self.constraint.merge(constraint)
        return self.p.importance(key, merged, self.args)

    def filter_to_unconstrained(self, choice_map: ChoiceMap) -> ChoiceMap:
        return choice_map.filter(~self.constraint.get_selection())

    def __getitem__(self, addr):
        return self.constraint[addr]


# =================================================================================================
# Algorithm / SampleDistribution  (inference/sp.py:101-199)
# =================================================================================================
class SampleDistribution(Distribution):
    """Distributions whose return value is a ChoiceMap."""

    def canonical_args(self, args, kwargs):
        return tuple(args)

    def simulate(self, key, args):
        w, v = self.random_weighted(key, *args)
        return DistributionTrace(self, args, v, w)

    def __call__(self, *args, **kwargs):
        return super().__call__(*args, **kwargs)


class Algorithm(SampleDistribution):
    def random_weighted(self, key, *args):
        raise NotImplementedError

    def estimate_logpdf(self, key, v, *args):
        raise NotImplementedError

    def estimate_normalizing_constant(self, key, target):
        raise NotImplementedError

    def estimate_reciprocal_normalizing_constant(self, key, target, latent_choices, w):
        raise NotImplementedError


# =================================================================================================
# ParticleCollection  (inference/smc.py:76-109)
# =================================================================================================
class ParticleCollection:
    """Weighted particles: a trace whose leaves carry the particle axis, log-weights f32[K]."""

    def __init__(self, particles: Trace, log_weights: torch.Tensor, is_valid=True, max_partials=None,
                 row_stats=None):
        self.particles, self.log_weights, self.is_valid = particles, log_weights, is_valid
        self._max_partials = max_partials
        self._rows = row_stats  # row-anchored partial sums emitted by the kernel that produced log_weights
        self._lse = None

    def get_particles(self) -> Trace:
        return self.particles

    def get_log_weights(self) -> torch.Tensor:
        return self.log_weights

    def __len__(self):
        return int(self.log_weights.shape[0])

    def _lse_triple(self):
        """(lse, e, q): row-anchored log-sum-exp — from the producer kernel's
        partial sums when it emitted them, else one pass over the weights."""
        if self._lse is None:
            ops = get_ops()
            rows = self._rows if self._rows is not None else ops.row_stats(self.log_weights.contiguous())
            self._lse = ops.lse_rows(rows)
        return self._lse

    def get_log_marginal_likelihood_estimate(self) -> torch.Tensor:
        """logsumexp(log_weights) - log K  (smc.py:96-97), f32 scalar tensor on the device."""
        lse, _, _ = self._lse_triple()
        return lse[0] - math.log(len(self))

    def log_marginal_likelihood_estimate_f64(self) -> float:
        """The same estimate evaluated in float64 from the exact (anchor, fixed-point sum) pair."""
        _, e, q = self._lse_triple()
        return get_ops().log_z_from_rows(e, q, len(self))

    def get_particle(self, idx) -> Trace:
        """tree_map(lambda v: v[idx]) over the trace (smc.py:90-91)."""
        if isinstance(idx, torch.Tensor):
            idx = int(idx.reshape(-1)[0].cpu())
        n = len(self)
        return self.particles.map_leaves(lambda v: v[idx] if _has_particle_axis(v, n) else v)

    def __getitem__(self, idx):
        return self.get_particle(idx), self.log_weights[idx]

    def sample_particle(self, key) -> Trace:
        """One particle with probability proportional to its weight (smc.py:102-109): a
        categorical draw over log_weights - logsumexp(log_weights) — the normaliser is a constant
        shift, which neither Gumbel-max nor the inverse-CDF draw depends on."""
        ops = get_ops()
        kb = _literal_key(key)
        idx = ops.categorical_index(kb, self.log_weights.contiguous(), mode=0)
        return self.get_particle(idx)

    # -- resampling (not in the reference library: SURVEY F3/E2) -----------------------------------
    def resample(self, key, method: str = "systematic", n_out: int | None = None) -> "ParticleCollection":
        """New equally weighted collection whose particles are drawn with probability proportional
        to the weights.  log-weights become logsumexp(lw) - log K, preserving the estimate."""
        ops = get_ops()
        n = len(self)
        anc, _, _ = ops.resample(method, _literal_key(key), self.log_weights.contiguous(), n_out)
        new = gather_trace(self.particles, anc, n)
        lz = self.get_log_marginal_likelihood_estimate()
        k = anc.shape[0]
        lw = torch.zeros(k, dtype=torch.float32, device=self.log_weights.device) + lz
        out = ParticleCollection(new, lw, self.is_valid)
        out.ancestors = anc
        return out

    def effective_sample_size(self) -> float:
        lw = self.log_weights.double()
        lw = lw - lw.max()
        w = torch.exp(lw)
        return float((w.sum() ** 2 / (w * w).sum()).cpu())


def _has_particle_axis(v, n) -> bool:
    return isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n


def _literal_key(key) -> KeyBatch:
    if isinstance(key, prng.PRNGKey):
        return key.literal()
    raise TypeError("expected a scalar PRNG key")


def gather_trace(trace: Trace, ancestors: torch.Tensor, n: int) -> Trace:
    """trace[ancestors] for every leaf with the particle axis; 4-byte columns go through the
    gather kernel in one launch."""
    ops = get_ops()
    cols, seen = [], {}

    def collect(v):
        if _has_particle_axis(v, n) and v.dim() == 1 and v.element_size() == 4 and v.is_contiguous():
            if id(v) not in seen:
                seen[id(v)] = len(cols)
                cols.append(v)
        return v

    trace.map_leaves(collect)
    gathered = ops.gather_cols(ancestors, cols) if cols else []
    idx64 = None

    def apply(v):
        nonlocal idx64
        if id(v) in seen:
            return gathered[seen[id(v)]]
        if _has_particle_axis(v, n):
            if idx64 is None:
                idx64 = ancestors.long()
            return v[idx64]
        return v

    return trace.map_leaves(apply)


def stack_to_first_dim(a, b):
    """CSMC stacking helper (smc.py:56-68): concatenate along the first axis, then squeeze."""
    ops = get_ops()
    a = torch.as_tensor(a, device=ops.device()) if not isinstance(a, torch.Tensor) else a
    b = torch.as_tensor(b, device=a.device) if not isinstance(b, torch.Tensor) else b.to(a.device)
    if a.dim() <= 1:
        a = a.reshape(-1, 1)
    if b.dim() <= 1:
        b = b.reshape(-1, 1)
    if a.dtype != b.dtype:
        b = b.to(a.dtype)
    return torch.cat([a, b], dim=0).squeeze()


def _col_of(v, n: int):
    """Broadcast a per-trace scalar leaf to a [n] column (constrained constants are stored once)."""
    ops = get_ops()
    if isinstance(v, torch.Tensor):
        if v.dim() >= 1 and v.shape[0] == n:
            return v
        return v.reshape(1).expand(n).contiguous() if v.dim() == 0 else v
    if isinstance(v, bool):
        return torch.full((n,), v, dtype=torch.bool, device=ops.device())
    if isinstance(v, int):
        return torch.full((n,), v, dtype=torch.int32, device=ops.device())
    if isinstance(v, float):
        return torch.full((n,), v, dtype=torch.float32, device=ops.device())
    return v


def _stack_leaf(a, na: int, b):
    a = _col_of(a, na)
    if not isinstance(a, torch.Tensor):
        return a
    b = torch.as_tensor(b, device=a.device).to(a.dtype).reshape((1,) + tuple(a.shape[1:]))
    return torch.cat([a, b], dim=0)


def _stack_traces(a: Trace, na: int, b: Trace) -> Trace:
    """tree_map(stack_to_first_dim, a, b): `a` carries na particles, `b` one; b goes LAST."""
    from .lang import MaterialTrace

    ca, cb = dict(a.get_choices().leaves()), dict(b.get_choices().leaves())
    choices = ChoiceMap.from_mapping([(addr, _stack_leaf(v, na, cb[addr])) for addr, v in ca.items()])
    ra, rb = a.get_retval(), b.get_retval()
    retval = _stack_leaf(ra, na, rb) if isinstance(ra, (torch.Tensor, bool, int, float)) else ra
    return MaterialTrace(a.get_gen_fn(), a.get_args(), retval, choices, _stack_leaf(a.get_score(), na, b.get_score()))


def _expand0(trace: Trace) -> Trace:
    return trace.map_leaves(lambda v: v[None] if isinstance(v, torch.Tensor) else v)


def _as_col(v, n=1):
    ops = get_ops()
    if isinstance(v, torch.Tensor):
        return v.reshape(-1).to(torch.float32)
    return torch.full((n,), float(v), dtype=torch.float32, device=ops.device())


# =================================================================================================
# SMC algorithms  (inference/smc.py:117-465)
# =================================================================================================
class SMCAlgorithm(Algorithm):
    """Abstract SMC algorithm: `run_smc`, `run_csmc`, and the derived estimators."""

    def get_num_particles(self) -> int:
        raise NotImplementedError

    def get_final_target(self) -> Target:
        raise NotImplementedError

    def run_smc(self, key) -> ParticleCollection:
        raise NotImplementedError

    def run_csmc(self, key, retained: ChoiceMap) -> ParticleCollection:
        raise NotImplementedError

    def log_marginal_likelihood_estimate(self, key, target: Target | None = None):
        algorithm = ChangeTarget(self, target) if target else self
        key, sub_key = split(key)
        return algorithm.run_smc(sub_key).get_log_marginal_likelihood_estimate()

    # -- GenSP ---------------------------------------------------------------------------------------
    def random_weighted(self, key, *args):
        assert isinstance(args[0], Target)
        target = args[0]
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        collection = algorithm.run_smc(key)
        particle = collection.sample_particle(sub_key)
        log_density_estimate = particle.get_score() - collection.get_log_marginal_likelihood_estimate()
        chm = target.filter_to_unconstrained(particle.get_choices())
        return log_density_estimate, chm

    def random_weighted_batch(self, keys, *args):
        """`vmap(self.random_weighted, in_axes=(0, None))(keys, target)` in a bounded number of launches, or None when this
        algorithm has no batched form (jaxlike.vmap then runs key by key).  Element b of every result equals
        `self.random_weighted(keys[b], target)` bit for bit."""
        return None

    def estimate_logpdf(self, key, v: ChoiceMap, *args):
        assert isinstance(args[0], Target)
        target = args[0]
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        collection = algorithm.run_csmc(key, v)
        particle = collection.sample_particle(sub_key)
        return particle.get_score() - collection.get_log_marginal_likelihood_estimate()

    # -- VI via GRASP ----------------------------------------------------------------------------------
    def estimate_normalizing_constant(self, key, target: Target):
        algorithm = ChangeTarget(self, target)
        key, sub_key = split(key)
        return algorithm.run_smc(sub_key).get_log_marginal_likelihood_estimate()

    def estimate_reciprocal_normalizing_constant(self, key, target: Target, latent_choices: ChoiceMap, w):
        algorithm = ChangeTarget(self, target)
        return algorithm.run_csmc_for_normalizing_constant(key, latent_choices, w)


class Importance(SMCAlgorithm):
    """One-particle importance sampling, optional proposal `q` (smc.py:233-279)."""

    def __init__(self, target: Target, q: SampleDistribution | None = None):
        self.target, self.q = target, q

    def get_num_particles(self):
        return 1

    def get_final_target(self):
        return self.target

    def run_smc(self, key):
        key, sub_key = split(key)
        if self.q is not None:
            log_weight, choice = self.q.random_weighted(sub_key, self.target)
            tr, target_score = self.target.importance(key, choice)
        else:
            log_weight = 0.0
            tr, target_score = self.target.importance(key, ChoiceMap.empty())
        return ParticleCollection(_expand0(tr), _as_col(target_score - log_weight), True)

    def run_csmc(self, key, retained: ChoiceMap):
        key, sub_key = split(key)
        q_score = self.q.estimate_logpdf(sub_key, retained, self.target) if self.q else 0.0
        tr, target_score = self.target.importance(key, retained)
        return ParticleCollection(_expand0(tr), _as_col(target_score - q_score), True)


class _FastEstimate:
    """What ImportanceK.log_marginal_likelihood_estimate keeps between calls (see _fast_state)."""

    __slots__ = ("ops", "fast_math", "tensors", "plan", "tracer", "params", "log_k", "n", "preps")

    def __init__(self, ops, fast_math, tensors, plan, tracer, params, log_k, n):
        self.ops, self.fast_math, self.tensors, self.plan, self.tracer = ops, fast_math, tensors, plan, tracer
        self.params, self.log_k, self.n, self.preps = params, log_k, n, {}


class ImportanceK(SMCAlgorithm):
    """K-particle importance sampling (smc.py:282-351): `k_particles` keys, ONE batched run."""

    def __init__(self, target: Target, q: SampleDistribution | None = None, k_particles: int = 2):
        self.target, self.q, self.k_particles = target, q, int(k_particles)

    def get_num_particles(self):
        return self.k_particles

    def get_final_target(self):
        return self.target

    def run_smc(self, key):
        key, sub_key = split(key)
        sub_keys = split(sub_key, self.get_num_particles())
        if self.q is not None:
            # the SAME sub_keys drive the proposal and the target (smc.py:302-305)
            log_weights, choices = _batched_random_weighted(self.q, sub_keys, self.target)
            trs, target_scores = self.target.importance(sub_keys, choices)
            return ParticleCollection(trs, target_scores - log_weights, True)
        trs, target_scores = self.target.importance(sub_keys, ChoiceMap.empty())
        return ParticleCollection(trs, target_scores, True, max_partials=getattr(trs, "max_partials", None),
                                  row_stats=getattr(trs, "row_stats", None))

    # -- the literal call `ImportanceK(target, k_particles=K).log_marginal_likelihood_estimate(key)` (smc.py:145-156, the
    # README's and tests/inference/test_smc.py's form) on a plan-able target without a proposal.  The general route builds
    # the whole `ParticleCollection` first (trace objects per site, freshly allocated columns, plan and trace cache
    # look-ups): ~85 us of host work around a 19 us kernel.  This route keeps, on the algorithm object, the traced body,
    # its plan and ONE set of persistent buffers with pre-marshalled launches, and per call only derives the key, points
    # the launch at it, enqueues the importance kernel and the fold of its row sums, and subtracts log K — the same
    # kernels on the same keys, so the same bits.
    def _fast_state(self):
        ops = get_ops()
        # one state PER HOST THREAD: the plan takes its parameters set-then-run, and the plan cache behind
        # _make_plan is thread-local for that reason — a second thread must not reuse (or re-parameterise) the first one's plan
        states = self.__dict__.setdefault("_fast", {})
        tid = threading.get_ident()
        st = states.get(tid)
        if st is not None and st.ops is ops and st.fast_math == fast_math_enabled():
            for t, v in st.tensors:
                if t._version != v:
                    break
            else:
                return st
        states[tid] = None
        from .lang import StaticGenerativeFunction
        from .plan import _make_plan, _needs_eager, _traced

        p_, n = self.target.p, self.get_num_particles()
        if self.q is not None or not isinstance(p_, StaticGenerativeFunction) or n < 2:
            return None
        if any(_needs_eager(a) for a in self.target.args):
            return None
        merged = self.target.constraint.merge(ChoiceMap.empty())
        traced = _traced(p_, merged, n, self.target.args)
        if traced is None:
            return None
        tracer = traced[0]
        # the estimate needs logsumexp(lw) alone: the walk without value columns, score or log-weights — its kernel stores
        # one (anchor, sum) pair per 256 particles and the fold of them; same draws, same weights, same fixed-point sums
        plan = _make_plan(tracer, estimate_only=True)  # (sets the launch parameters: observations and scalar arguments)
        tensors = [t for t in list(self.target.args) + [v for _, v in merged.leaves()] if isinstance(t, torch.Tensor)]
        st = _FastEstimate(ops, fast_math_enabled(), tuple((t, t._version) for t in tensors), plan, tracer, list(tracer.params),
                           math.log(n), n)
        states[tid] = st
        return st

    def _fast_estimate(self, key):
        if type(key) is not prng.PRNGKey:
            return None
        st = self._fast_state()
        if st is None:
            return None
        impl = key.impl
        slot = (impl, threading.get_ident())  # (persistent buffers: one set per generator and host thread)
        prep = st.preps.get(slot)
        if prep is None:
            kb = prng.split_lazy(prng.split_at(prng.split_at(key, 1), 1), st.n)
            try:
                prep = st.preps[slot] = st.ops.prepare_importance(st.plan, kb, st.n, st.tracer.inputs, [], estimate_only=True)
            except abi.GjxError as e:
                from .runtime import compiler_switched_off

                if compiler_switched_off(e):
                    return None
                raise
        plan = st.plan
        if st.params and plan.params_owner is not st:  # (another algorithm object may share the cached plan)
            plan.set_params(st.params)
            plan.params_owner = st
        # ONE library call = ONE launch (gjx_importance_estimate): key, sub_key = split(key) [the estimate]; key, sub_key =
        # split(sub_key); sub_keys = split(sub_key, K), lazily [run_smc]; the walk, the fold of its row sums by the workgroup
        # that finishes last, and lse - log K (one f32 subtraction, as `lse[0] - math.log(K)` on the general route)
        out = torch.empty((), dtype=torch.float32, device=st.ops._alloc_device)
        try:
            prep.launch_estimate(key.k0, key.k1, key.lane, out, st.log_k)
        except abi.GjxError as e:
            from .runtime import compiler_switched_off

            if compiler_switched_off(e):  # GJX_PLAN_JIT=0 and a body with programs / nested calls: the general route
                return None
            raise
        return out

    def log_marginal_likelihood_estimate(self, key, target: Target | None = None):
        if target is None:
            fast = self._fast_estimate(key)
            if fast is not None:
                return fast
        return super().log_marginal_likelihood_estimate(key, target)

    def random_weighted_batch(self, keys, *args):
        """The README's call (/root/reference/README.md:111-113: `jax.vmap(alg.random_weighted, in_axes=(0, None))(sub_keys,
        posterior_target)`, 50 trials of K = 50 particles) as TWO launches per 32 trials instead of a run per trial: the
        trials' importance passes share one `gjx_importance_run_batch` launch (+ one fold of their row sums), their particle
        draws one `gjx_categorical_index_batch` launch.  Same key derivation as the scalar path (smc.py:162-179: key, sub_key =
        split(key); run_smc(key) -> split, split(sub, K); sample_particle(sub_key)); the target must be THIS algorithm's target
        (ChangeTarget's identity shortcut), no custom proposal, a flat plan-able body, K <= 1024."""
        from .lang import StaticGenerativeFunction
        from .plan import fused_generate_batch

        keys = list(keys)
        k = self.get_num_particles()
        if (len(args) != 1 or args[0] is not self.target or self.q is not None or k > 1024 or len(keys) < 2
                or not isinstance(self.target.p, StaticGenerativeFunction) or any(type(key) is not prng.PRNGKey for key in keys)):
            return None
        ops = get_ops()
        ests, cols = [], None
        for lo in range(0, len(keys), 32):
            chunk = keys[lo:lo + 32]
            pks, draw_keys = [], []
            for key in chunk:
                key_run, sub_key = split(key)        # random_weighted
                _, sub2 = split(key_run)             # run_smc
                pks.append(split(sub2, k))
                draw_keys.append(_literal_key(sub_key))
            res = fused_generate_batch(self.target.p, pks, self.target.constraint, self.target.args)
            if res is None:
                return None
            idx = ops.categorical_index_batch(draw_keys, res["logw"], k).reshape(-1, 1)
            score = res["score"].gather(1, idx).reshape(-1)
            ests.append(score - (res["lse"] - math.log(k)))
            picked = [(addr, v.gather(1, idx).reshape(-1)) for addr, v in res["values"]]
            cols = picked if cols is None else [(a, torch.cat([c0, c1])) for (a, c0), (_, c1) in zip(cols, picked)]
        return torch.cat(ests), ChoiceMap.from_mapping(cols)

    def log_marginal_likelihood_estimates(self, keys):
        """`vmap(self.log_marginal_likelihood_estimate)(keys)`: one independent estimate per key, as a
        float32 tensor [len(keys)].  A plan-able target without a custom proposal runs up to 32 estimates per
        kernel launch (a single 1e6-particle pass does not keep an MI355X full for long enough); anything
        else runs key by key.  Element b equals `self.log_marginal_likelihood_estimate(keys[b])` bit for bit."""
        from .lang import StaticGenerativeFunction
        from .plan import fused_log_weights_batch

        keys = list(keys)
        k = self.get_num_particles()
        out = []
        fusable = self.q is None and isinstance(self.target.p, StaticGenerativeFunction) and len(keys) > 1
        for lo in range(0, len(keys), 32):
            chunk, res = keys[lo:lo + 32], None
            if fusable:
                pks = []
                for key in chunk:  # log_marginal_likelihood_estimate: split; run_smc: split, split(sub, K)
                    _, sub_key = split(key)
                    _, sub_key = split(sub_key)
                    pks.append(split(sub_key, k))
                res = fused_log_weights_batch(self.target.p, pks, self.target.constraint, self.target.args)
            if res is None:
                fusable = False
                out.extend(torch.as_tensor(self.log_marginal_likelihood_estimate(key)).reshape(1) for key in chunk)
            else:
                out.append(res[1] - math.log(k))
        return torch.cat([t.to(out[0].device) for t in out])

    def run_csmc(self, key, retained: ChoiceMap):
        k = self.get_num_particles()
        key, sub_key = split(key)
        sub_keys = split(sub_key, k - 1)
        if self.q:
            log_scores, choices = _batched_random_weighted(self.q, sub_keys, self.target)
            retained_choice_score = self.q.estimate_logpdf(key, retained, self.target)
            stacked_choices = _stack_choice_maps(choices, k - 1, retained)
            stacked_scores = _stack_leaf(log_scores, k - 1, retained_choice_score)
            sub_keys = split(key, k)
            target_traces, target_scores = self.target.importance(sub_keys, stacked_choices)
        else:
            if k > 1:
                ignored_traces, ignored_scores = self.target.importance(sub_keys, ChoiceMap.empty())
            retained_trace, retained_choice_score = self.target.importance(key, retained)
            if k > 1:
                target_scores = _stack_leaf(ignored_scores, k - 1, retained_choice_score)
                target_traces = _stack_traces(ignored_traces, k - 1, retained_trace)  # retained particle LAST
            else:
                target_scores = _as_col(retained_choice_score)
                target_traces = _expand0(retained_trace)
            stacked_scores = 0.0
        return ParticleCollection(target_traces, (target_scores - stacked_scores).reshape(-1), True)


def _batched_random_weighted(q, sub_keys: ParticleKeys, target: Target):
    """vmap(q.random_weighted, in_axes=(0, None))(sub_keys, target)."""
    if hasattr(q, "batched_random_weighted"):
        return q.batched_random_weighted(sub_keys, target)
    try:  # proposals built from `@gen` functions (Marginal) run once over the whole key batch
        w, chm = q.random_weighted(sub_keys, target)
        if isinstance(w, torch.Tensor) and w.dim() == 1 and w.shape[0] == sub_keys.n:
            return w, chm
    except (TypeError, NotImplementedError):
        pass
    # generic proposals run once per key (the reference vmaps them); their results are stacked
    ws, chms = [], []
    for k in sub_keys:
        w, chm = q.random_weighted(k, target)
        ws.append(torch.as_tensor(w, dtype=torch.float32, device=get_ops().device()).reshape(()))
        chms.append(dict(chm.leaves()))
    choices = ChoiceMap.from_mapping(
        [(addr, torch.stack([torch.as_tensor(c[addr], device=get_ops().device()) for c in chms])) for addr in chms[0]])
    return torch.stack(ws), choices


def _stack_choice_maps(batched: ChoiceMap, n: int, single: ChoiceMap) -> ChoiceMap:
    sdict = dict(single.leaves())
    return ChoiceMap.from_mapping([(addr, _stack_leaf(v, n, sdict[addr])) for addr, v in batched.leaves()])


class ChangeTarget(SMCAlgorithm):
    """Re-weight a collection for a new target (smc.py:359-465)."""

    def __init__(self, prev: SMCAlgorithm, target: Target):
        self.prev, self.target = prev, target

    def get_num_particles(self):
        return self.prev.get_num_particles()

    def get_final_target(self):
        return self.target

    def _reweight(self, key, collection: ParticleCollection, n_keys: int, upto: int | None = None):
        """vmap(_reweight)(split(key, n), particles, weights): constrain the new target to the
        particles' latent choices, w' = new_weight - old_score + w."""
        particles, weights = collection.get_particles(), collection.get_log_weights()
        if upto is not None:
            n_all = len(collection)
            particles = particles.map_leaves(lambda v: v[:upto] if _has_particle_axis(v, n_all) else v)
            weights = weights[:upto]
        if self.target is self.prev.get_final_target() and upto is None:
            # Identical target object: new_weight is the particle's full score recomputed by the same
            # kernels, so new_weight - score + w == w bit for bit; skip the second pass.
            return particles, weights
        latents = self.prev.get_final_target().filter_to_unconstrained(particles.get_choices())
        sub_keys = split(key, n_keys)
        new_trace, new_weight = self.target.importance(sub_keys, latents)
        return new_trace, new_weight - particles.get_score() + weights

    def run_smc(self, key):
        collection = self.prev.run_smc(key)
        new_particles, new_weights = self._reweight(key, collection, self.get_num_particles())
        same = new_weights is collection.log_weights
        return ParticleCollection(new_particles, new_weights, True,
                                  max_partials=collection._max_partials if same else None,
                                  row_stats=collection._rows if same else None)

    def run_csmc(self, key, retained: ChoiceMap):
        collection = self.prev.run_csmc(key, retained)
        new_particles, new_weights = self._reweight(key, collection, self.get_num_particles())
        return ParticleCollection(new_particles, new_weights, True)

    def run_csmc_for_normalizing_constant(self, key, latent_choices: ChoiceMap, w):
        key, sub_key = split(key)
        collection = self.prev.run_csmc(sub_key, latent_choices)
        num_particles = self.get_num_particles()
        retained_score = collection.get_particle(-1).get_score()
        retained_weight = collection.get_log_weights()[-1]
        tail = _as_col(w - retained_score + retained_weight)
        if num_particles > 1:
            _, rejected = self._reweight_always(key, collection, num_particles - 1, upto=num_particles - 1)
            all_weights = torch.cat([rejected.reshape(-1), tail])
        else:
            all_weights = tail
        lse, _, _ = get_ops().logsumexp(all_weights.contiguous())
        return retained_score - (lse[0] - math.log(num_particles))

    def _reweight_always(self, key, collection, n_keys, upto):
        n_all = len(collection)
        particles = collection.get_particles().map_leaves(lambda v: v[:upto] if _has_particle_axis(v, n_all) else v)
        weights = collection.get_log_weights()[:upto]
        latents = self.prev.get_final_target().filter_to_unconstrained(particles.get_choices())
        new_trace, new_weight = self.target.importance(split(key, n_keys), latents)
        return new_trace, new_weight - particles.get_score() + weights


# =================================================================================================
# Marginal  (inference/sp.py:207-273)
# =================================================================================================
class Marginal(SampleDistribution):
    """Marginal of a generative function over a selection, optionally with a custom algorithm."""

    def __init__(self, gen_fn: GenerativeFunction, selection: Selection = Selection.all(), algorithm=None):
        self.gen_fn, self.selection, self.algorithm = gen_fn, selection, algorithm

    def random_weighted(self, key, *args):
        key, sub_key = split(key)
        tr = self.gen_fn.simulate(sub_key, args)
        choices = tr.get_choices()
        latent_choices = choices.filter(self.selection)
        key, sub_key = split(key)
        weight = tr.project(sub_key, ~self.selection)
        if self.algorithm is None:
            return weight, latent_choices
        target = Target(self.gen_fn, args, latent_choices)
        other_choices = choices.filter(~self.selection)
        z = self.algorithm.estimate_reciprocal_normalizing_constant(key, target, other_choices, weight)
        return z, latent_choices

    def estimate_logpdf(self, key, v: ChoiceMap, *args):
        if self.algorithm is None:
            _, weight = self.gen_fn.importance(key, v, args)
            return weight
        target = Target(self.gen_fn, args, v)
        return self.algorithm.estimate_normalizing_constant(key, target)


def marginal(selection: Selection = Selection.all(), algorithm=None):
    def decorator(gen_fn: GenerativeFunction) -> Marginal:
        return Marginal(gen_fn, selection, algorithm)

    return decorator
