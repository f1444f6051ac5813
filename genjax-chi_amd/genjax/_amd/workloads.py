"""The BASELINE.json workloads expressed directly on the C-ABI wrappers (`Ops`): synthetic inputs,
key schedules, fused-kernel drivers and the closed-form log-Z each one is checked against.

These are the fixed-structure fast paths the host API (`ImportanceK`, `BootstrapSMC`) lowers to;
bench.py, __graft_entry__.smoke() and the parity tests call them with either backend.
  C2  gaussian10_importance  ImportanceK on a 10-latent Gaussian model
  C3  lgssm_smc              bootstrap SMC, linear-Gaussian state space (C3)
  C5  hmm_smc                bootstrap SMC, 256-state HMM (C5; model shape exact_testbed.py:62-68)
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import abi, prng
from .ops import Ops

# ---------------------------------------------------------------------------------------------
# C2: ImportanceK, z_i ~ N(0,1), y_i ~ N(z_i, 0.5), i < 10
# ---------------------------------------------------------------------------------------------
G10_LATENTS = 10
G10_OBS_SCALE = 0.5


def gaussian10_data():
    rng = np.random.default_rng(0)
    z = rng.standard_normal(G10_LATENTS)
    y = z + G10_OBS_SCALE * rng.standard_normal(G10_LATENTS)
    return y.astype(np.float32)


def gaussian10_exact_log_z(y=None) -> float:
    y = gaussian10_data() if y is None else y
    var = 1.0 + G10_OBS_SCALE**2
    yd = y.astype(np.float64)
    return float(np.sum(-0.5 * yd * yd / var - 0.5 * math.log(2 * math.pi * var)))


def gaussian10_sites(y) -> list[abi.Site]:
    """Site table of
        @gen
        def model():
            for i in range(10):
                z = normal(0.0, 1.0) @ f"z{i}"
                normal(z, 0.5) @ f"y{i}"          # constrained to y[i]
    in program order (static.py:349-352 numbers the sites 1..20)."""
    sites = []
    for i in range(G10_LATENTS):
        z = abi.Site()
        z.dist, z.observed, z.out_col = abi.DIST_NORMAL, 0, i
        z.arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, 0.0, None)
        z.arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, 1.0, None)
        sites.append(z)
        o = abi.Site()
        o.dist, o.observed, o.out_col = abi.DIST_NORMAL, 1, -1
        o.arg[0] = abi.Arg(abi.ARG_SITE, 2 * i, 1.0, 0.0, None)
        o.arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, G10_OBS_SCALE, None)
        o.obs = abi.Arg(abi.ARG_CONST, 0, 0.0, float(y[i]), None)
        sites.append(o)
    return sites


def importance_particle_keys(root: prng.PRNGKey, n_local: int, first: int = 0):
    """Key tree of ImportanceK(...).log_marginal_likelihood_estimate(key):
    (_, k1) = split(key) [smc.py:154]; (_, k2) = split(k1) [smc.py:299]; p_i = split(k2, N)[i]."""
    k1 = prng.split(root)[1]
    k2 = prng.split(k1)[1]
    return prng.split_lazy(k2, n_local, first)


class Gaussian10:
    """Reusable state of the C2 workload (plan + key batch), so a bench step is kernels only."""

    def __init__(self, ops: Ops, impl: int, seed: int, n_local: int, first: int = 0, n_total: int | None = None,
                 fast_math: bool = False):
        self.ops, self.n, self.first, self.impl, self.seed = ops, n_local, first, impl, seed
        self.n_total = n_local if n_total is None else n_total
        self.y = gaussian10_data()
        self.plan = ops.plan_create(gaussian10_sites(self.y), fast_math=fast_math)
        self.keys = importance_particle_keys(prng.key(seed, impl), n_local, first)
        self.frac = ops.frac_bits(self.n_total)

    def prepare(self, fold_batch: int = 1, passes: int = 1):
        """Persistent-buffer form of `step` (no host allocation per pass); `fold_batch` passes share one
        log-sum-exp launch and `passes` independent passes (seeds seed, seed+1, ...) share one importance
        launch (ops.PreparedImportance)."""
        prep = getattr(self, "_prep", None)
        if prep is None or prep.fold_batch != fold_batch or prep.launch_passes_n != passes:
            keys = self.keys if passes == 1 else [
                importance_particle_keys(prng.key(self.seed + p, self.impl), self.n, self.first) for p in range(passes)]
            self._prep = self.ops.prepare_importance(self.plan, keys, self.n, [], [torch.float32] * G10_LATENTS,
                                                     fold_batch=fold_batch)
        return self._prep

    def step(self):
        """One ImportanceK pass on this rank: trace columns, score, log-weights and the local
        (max, fixed-point sum) pair — all device tensors, no host sync."""
        vals, score, logw, mp, rows = self.ops.importance_run(
            self.plan, self.keys, self.n, [], [torch.float32] * G10_LATENTS, want_score=True, want_max_partials=True,
            want_rows=True)
        if self.n_total == self.n:
            lse, m, q = self.ops.logsumexp(logw, max_partials=mp)  # max-anchored form (shardable, §3.5)
            rlse, re, rq = self.ops.lse_rows(rows)  # row-anchored form (fused into the kernel, §3.5b)
            return dict(values=vals, score=score, logw=logw, lse=lse, max=m, q=q, rows=rows, row_lse=rlse,
                        row_e=re, row_q=rq)
        # a shard of a larger population: its exchangeable record (dist.BatchedImportance merges them)
        record = torch.zeros(abi.LSE_RECORD_WORDS, dtype=torch.int64, device=logw.device)
        self.ops.lse_rows(rows, record=record)
        return dict(values=vals, score=score, logw=logw, rows=rows, record=record, max_partials=mp)


def gaussian10_importance(ops: Ops, impl: int, seed: int, n: int, fast_math: bool = False):
    w = Gaussian10(ops, impl, seed, n, fast_math=fast_math)
    out = w.step()
    q, m = int(out["q"].cpu()), float(out["max"].cpu())
    log_z = m + math.log(q) - w.frac * math.log(2.0) - math.log(n)
    return dict(logw=out["logw"], score=out["score"], values=out["values"], q=q, max=m, lse=float(out["lse"].cpu()),
                log_z=log_z, log_z_exact=gaussian10_exact_log_z(w.y), rows=out["rows"],
                row_lse=float(out["row_lse"].cpu()), row_e=int(out["row_e"].cpu()), row_q=int(out["row_q"].cpu()),
                log_z_rows=ops.log_z_from_rows(out["row_e"], out["row_q"], n))


# ---------------------------------------------------------------------------------------------
# The rejection samplers under ImportanceK (north star: "fused Beta/Gamma/Normal/Categorical samplers"): the README's
# beta-bernoulli model (/root/reference/README.md:89-93) and a Gamma-Normal model, as site tables
# ---------------------------------------------------------------------------------------------
def beta_bernoulli_sites(obs: bool = True, alpha: float = 2.0, beta: float = 2.0) -> list[abi.Site]:
    """@gen def beta_bernoulli(a, b): p = beta(a, b) @ "p"; v = flip(p) @ "v"; return v     (v constrained to `obs`)"""
    p = abi.Site()
    p.dist, p.observed, p.out_col = abi.DIST_BETA, 0, 0
    p.arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, alpha, None)
    p.arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, beta, None)
    v = abi.Site()
    v.dist, v.observed, v.out_col = abi.DIST_BERNOULLI, 1, -1
    v.arg[0] = abi.Arg(abi.ARG_SITE, 0, 1.0, 0.0, None)
    v.obs = abi.Arg(abi.ARG_CONST, 0, 0.0, 1.0 if obs else 0.0, None)
    return [p, v]


def beta_bernoulli_exact_log_z(obs: bool = True, alpha: float = 2.0, beta: float = 2.0) -> float:
    return math.log((alpha if obs else beta) / (alpha + beta))


GN_Y = np.array([0.8, -1.9, 0.35, 2.6, -0.7], dtype=np.float32)


def gamma_normal_sites(y=GN_Y, conc: float = 3.0, rate: float = 2.0) -> list[abi.Site]:
    """@gen def model(): s = gamma(3, 2) @ "s"; for i: normal(0, s) @ f"y{i}"      (a Gamma prior on the scale; y observed)"""
    s_ = abi.Site()
    s_.dist, s_.observed, s_.out_col = abi.DIST_GAMMA, 0, 0
    s_.arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, conc, None)
    s_.arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, rate, None)
    sites = [s_]
    for v in y:
        o = abi.Site()
        o.dist, o.observed, o.out_col = abi.DIST_NORMAL, 1, -1
        o.arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, 0.0, None)
        o.arg[1] = abi.Arg(abi.ARG_SITE, 0, 1.0, 0.0, None)
        o.obs = abi.Arg(abi.ARG_CONST, 0, 0.0, float(v), None)
        sites.append(o)
    return sites


def gamma_normal_exact_log_z(y=GN_Y, conc: float = 3.0, rate: float = 2.0) -> float:
    from scipy import integrate, stats

    yd = np.asarray(y, dtype=np.float64)
    f = lambda s: stats.gamma.pdf(s, conc, scale=1.0 / rate) * np.prod(stats.norm.pdf(yd, 0.0, s))  # noqa: E731
    return float(math.log(integrate.quad(f, 0.0, np.inf, limit=200)[0]))


class SiteModel:
    """An ImportanceK workload given by a site table (plan + key batch + persistent buffers), like Gaussian10."""

    def __init__(self, ops: Ops, impl: int, seed: int, n: int, sites: list, value_dtypes: list, log_z_exact: float):
        self.ops, self.n, self.impl, self.seed = ops, n, impl, seed
        self.first, self.n_total = 0, n
        self.plan = ops.plan_create(sites)
        self.value_dtypes = value_dtypes
        self.keys = importance_particle_keys(prng.key(seed, impl), n)
        self.log_z_exact = log_z_exact

    def prepare(self, fold_batch: int = 1, passes: int = 1):
        prep = getattr(self, "_prep", None)
        if prep is None or prep.fold_batch != fold_batch or prep.launch_passes_n != passes:
            keys = self.keys if passes == 1 else [importance_particle_keys(prng.key(self.seed + p, self.impl), self.n) for p in range(passes)]
            self._prep = self.ops.prepare_importance(self.plan, keys, self.n, [], self.value_dtypes, fold_batch=fold_batch)
        return self._prep

    def step(self):
        vals, score, logw, mp, rows = self.ops.importance_run(self.plan, self.keys, self.n, [], self.value_dtypes, want_score=True,
                                                            want_max_partials=True, want_rows=True)
        rlse, re, rq = self.ops.lse_rows(rows)
        return dict(values=vals, score=score, logw=logw, row_e=re, row_q=rq)


def beta_bernoulli_model(ops: Ops, impl: int, seed: int, n: int) -> SiteModel:
    return SiteModel(ops, impl, seed, n, beta_bernoulli_sites(), [torch.float32], beta_bernoulli_exact_log_z())


def gamma_normal_model(ops: Ops, impl: int, seed: int, n: int) -> SiteModel:
    return SiteModel(ops, impl, seed, n, gamma_normal_sites(), [torch.float32], gamma_normal_exact_log_z())


# ---------------------------------------------------------------------------------------------
# C3: bootstrap SMC on x_0~N(0,1), x_t~N(0.9 x_{t-1}, 1), y_t~N(x_t, 0.5)
# ---------------------------------------------------------------------------------------------
LGSSM = dict(x0_loc=0.0, x0_scale=1.0, a=0.9, q=1.0, r=0.5)


def lgssm_model() -> abi.Lgssm:
    return abi.Lgssm(LGSSM["x0_loc"], LGSSM["x0_scale"], LGSSM["a"], LGSSM["q"], LGSSM["r"])


def lgssm_data(T: int):
    rng = np.random.default_rng(1)
    x = np.empty(T)
    y = np.empty(T)
    for t in range(T):
        x[t] = (LGSSM["x0_loc"] + LGSSM["x0_scale"] * rng.standard_normal()) if t == 0 else (
            LGSSM["a"] * x[t - 1] + LGSSM["q"] * rng.standard_normal())
        y[t] = x[t] + LGSSM["r"] * rng.standard_normal()
    return y.astype(np.float32)


def lgssm_exact_log_z(y) -> float:
    """Scalar Kalman filter in float64."""
    m, p = LGSSM["x0_loc"], LGSSM["x0_scale"] ** 2
    ll = 0.0
    for t, yt in enumerate(np.asarray(y, dtype=np.float64)):
        if t > 0:
            m, p = LGSSM["a"] * m, LGSSM["a"] ** 2 * p + LGSSM["q"] ** 2
        s = p + LGSSM["r"] ** 2
        ll += -0.5 * (yt - m) ** 2 / s - 0.5 * math.log(2 * math.pi * s)
        k = p / s
        m, p = m + k * (yt - m), (1 - k) * p
    return ll


def smc_key_schedule(root: prng.PRNGKey, T: int):
    """step_keys[t], resample_keys[t] = fold_in(root, 2t), fold_in(root, 2t+1): fresh lane-0 keys (for
    threefry the same words as split(root, 2T)[2t], [2t+1])."""
    w = prng.fold_words(root, 2 * T)
    return w[0::2].copy(), w[1::2].copy()


def filter_key_schedules(seed: int, impl: int, T: int, filters: int):
    """Key schedule(s): [T, 2] arrays for one filter, [F, T, 2] for F filters with seeds seed .. seed+F-1."""
    if filters == 1:
        return smc_key_schedule(prng.key(seed, impl), T)
    pairs = [smc_key_schedule(prng.key(seed + f, impl), T) for f in range(filters)]
    return np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])


def smc_result(ops: Ops, out, n: int, filters: int, log_z_exact: float):
    out_e, out_q, state, logw, anc = out[:5]
    flags = out[5] if len(out) > 5 else None  # ESS-adaptive filters: 1 where a step began with a resampling
    if filters == 1:
        return dict(out_e=out_e, out_q=out_q, state=state, logw=logw, ancestors=anc, resampled=flags,
                    log_z=ops.log_z_from_pairs(out_e, out_q, n, flags), log_z_exact=log_z_exact)
    return dict(out_e=out_e, out_q=out_q, state=state[:, :n], logw=logw[:, :n],
                ancestors=None if anc is None else anc[:, :, :n], resampled=flags,
                log_z=[ops.log_z_from_pairs(out_e[f], out_q[f], n, None if flags is None else flags[f]) for f in range(filters)],
                log_z_exact=log_z_exact)


class LgssmSMC:
    """Reusable state of the C3 workload: data, key schedule and exact log Z are prepared once, so
    `run()` is only the enqueue of the fused filter (2 kernels per step, no host sync)."""

    def __init__(self, ops: Ops, impl: int, seed: int, n: int, T: int, want_ancestors: bool = False, filters: int = 1,
                 ess_threshold: float = 0.0, y=None, model=None):
        """`filters` > 1: that many independent filters (seeds seed, seed+1, ...) step in the same launches.
        `ess_threshold` in (0, 1): resample only when ESS < threshold * n (gjx_smc_config)."""
        self.ops, self.impl, self.n, self.T, self.want_ancestors, self.filters = ops, impl, n, T, want_ancestors, filters
        self.ess = ess_threshold
        self.y = lgssm_data(T) if y is None else np.asarray(y, dtype=np.float32)
        self.sk, self.rk = filter_key_schedules(seed, impl, T, filters)
        self.model = lgssm_model() if model is None else model
        self.log_z_exact = lgssm_exact_log_z(self.y) if model is None else float("nan")

    def run(self):
        return self.ops.smc_run_lgssm(self.impl, self.n, self.sk, self.rk, self.model, self.y, self.want_ancestors,
                                      ess_threshold=self.ess, want_flags=True)

    def result(self, out):
        return smc_result(self.ops, out, self.n, self.filters, self.log_z_exact)


def lgssm_smc(ops: Ops, impl: int, seed: int, n: int, T: int, want_ancestors: bool = False, **kw):
    w = LgssmSMC(ops, impl, seed, n, T, want_ancestors, **kw)
    return w.result(w.run())


# ---------------------------------------------------------------------------------------------
# C5: HMM with circulant logits (construction of distributions/custom/discrete_hmm.py:42-86)
# ---------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------
# C3 as the reference literally runs it when given ImportanceK over `step.scan(n=T)` (no resampling,
# [N, T] leaves; scan.py:237-294): the whole scan of every particle in one launch (gjx_scan_run)
# ---------------------------------------------------------------------------------------------
def lgssm_scan_sites():
    """Site table of the scan kernel
        @gen
        def step(x, _):
            x2 = normal(a * x, q) @ "x"
            normal(x2, r) @ "y"            # constrained to y[t]
            return x2, x2
    -> (sites, next_state)."""
    m = LGSSM
    x = abi.Site()
    x.dist, x.observed, x.out_col = abi.DIST_NORMAL, 0, 0
    x.arg[0] = abi.Arg(abi.ARG_STATE, 0, m["a"], 0.0, None)
    x.arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, m["q"], None)
    y = abi.Site()
    y.dist, y.observed, y.out_col = abi.DIST_NORMAL, 1, -1
    y.arg[0] = abi.Arg(abi.ARG_SITE, 0, 1.0, 0.0, None)
    y.arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, m["r"], None)
    y.obs = abi.Arg(abi.ARG_OBS, 0, 1.0, 0.0, None)
    return [x, y], [abi.Arg(abi.ARG_SITE, 0, 1.0, 0.0, None)]


class LgssmScan:
    """Reusable state of the scan-importance workload (plan, keys, observation table, output buffers)."""

    def __init__(self, ops: Ops, impl: int, seed: int, n: int, T: int, fast_math: bool = False, x0: float = 0.0):
        self.ops, self.impl, self.n, self.T, self.x0 = ops, impl, n, T, x0
        sites, nxt = lgssm_scan_sites()
        self.plan = ops.scan_plan_create(sites, nxt, 1, fast_math=fast_math)
        self.y = lgssm_data(T)
        self.obs = torch.from_numpy(self.y.reshape(T, 1).copy()).to(ops.device())
        self.kb = importance_particle_keys(prng.key(seed, impl), n)
        self.out = None

    def run(self):
        self.out = self.ops.scan_run(self.plan, self.kb, self.n, self.T, self.obs, [self.x0], [torch.float32], out=self.out)
        return self.out

    def result(self):
        o = self.out
        lse, e, q = self.ops.lse_rows(o["rows"])
        return dict(x=o["values"][0], logw=o["logw"], score=o["score"], carry=o["carry"][0], max_partials=o["max_partials"],
                    row_e=o["rows"].e, row_s=o["rows"].s, log_z=Ops.log_z_from_rows(e, q, self.n))


def lgssm_scan(ops: Ops, impl: int, seed: int, n: int, T: int, **kw):
    wl = LgssmScan(ops, impl, seed, n, T, **kw)
    wl.run()
    return wl.result()


class HmmScan:
    """ImportanceK over the HMM as a scan (the reference's literal configs[4] semantics: no resampling, `[T, N]` state
    trajectories): kernel `z' ~ categorical(trans[z]); y ~ categorical(obs[z'])` in one launch."""

    def __init__(self, ops: Ops, impl: int, seed: int, n: int, T: int, n_states=None, cat_mode: int = 1):
        self.ops, self.impl, self.n, self.T = ops, impl, n, T
        trans, obs = hmm_tables(n_states)
        self.k = trans.shape[0]
        dev = ops.device()
        self.trans = torch.from_numpy(trans).to(dev).contiguous()
        self.obs_l = torch.from_numpy(obs).to(dev).contiguous()
        z = abi.Site()
        z.dist, z.observed, z.out_col = abi.DIST_CATEGORICAL, 0, 0
        z.n_cat, z.n_rows, z.cat_mode = self.k, self.k, cat_mode
        z.arg[0] = abi.Arg(abi.ARG_STATE, 0, 1.0, 0.0, None)
        z.logits = self.trans.data_ptr()
        y = abi.Site()
        y.dist, y.observed, y.out_col = abi.DIST_CATEGORICAL, 1, -1
        y.n_cat, y.n_rows, y.cat_mode = self.k, self.k, cat_mode
        y.arg[0] = abi.Arg(abi.ARG_SITE, 0, 1.0, 0.0, None)
        y.obs = abi.Arg(abi.ARG_OBS, 0, 1.0, 0.0, None)
        y.logits = self.obs_l.data_ptr()
        self.plan = ops.scan_plan_create([z, y], [abi.Arg(abi.ARG_SITE, 0, 1.0, 0.0, None)], 1)
        self.y = hmm_data(T, n_states)
        self.obs = torch.from_numpy(self.y.astype(np.float32).reshape(T, 1)).to(dev)
        self.z0 = float(HMM["init_state"] % self.k)
        self.kb = importance_particle_keys(prng.key(seed, impl), n)
        self.out = None

    def run(self):
        self.out = self.ops.scan_run(self.plan, self.kb, self.n, self.T, self.obs, [self.z0], [torch.int32], out=self.out)
        return self.out

    def result(self):
        o = self.out
        lse, e, q = self.ops.lse_rows(o["rows"])
        return dict(z=o["values"][0], logw=o["logw"], score=o["score"], carry=o["carry"][0],
                    log_z=Ops.log_z_from_rows(e, q, self.n))


def scaled_circulant(n: int, k: int, epsilon: float, delta: float) -> np.ndarray:
    """Row-circulant matrix whose first column is eps^|i| within distance k of the diagonal
    (wrapping) and -delta elsewhere — restated from the reference's description, float32."""
    src = np.empty(n, dtype=np.float64)
    for i in range(n):
        if i <= k:
            src[i] = epsilon ** abs(i)
        elif i - n >= -k:
            src[i] = epsilon ** abs(i - n)
        else:
            src[i] = -delta
    idx = (np.arange(n)[:, None] - np.arange(n)[None, :]) % n  # scipy.linalg.circulant layout
    return src[idx].astype(np.float32)


HMM = dict(n_states=256, adjacency=8, sigma_trans=0.5, sigma_obs=0.5, init_state=128)


def hmm_tables(n_states=None):
    k = HMM["n_states"] if n_states is None else n_states
    adj = min(HMM["adjacency"], max(1, k // 4))
    trans = scaled_circulant(k, adj, HMM["sigma_trans"], 1.0 / HMM["sigma_trans"])
    obs = scaled_circulant(k, adj, HMM["sigma_obs"], 1.0 / HMM["sigma_obs"])
    return trans, obs


def _softmax64(l):
    l = l.astype(np.float64)
    l = l - l.max(axis=1, keepdims=True)
    p = np.exp(l)
    return p / p.sum(axis=1, keepdims=True)


def hmm_data(T: int, n_states=None, init_state=None):
    trans, obs = hmm_tables(n_states)
    k = trans.shape[0]
    z = (HMM["init_state"] if init_state is None else init_state) % k
    pt, po = _softmax64(trans), _softmax64(obs)
    rng = np.random.default_rng(2)
    ys = np.empty(T, dtype=np.int32)
    for t in range(T):
        z = rng.choice(k, p=pt[z])
        ys[t] = rng.choice(k, p=po[z])
    return ys


def hmm_exact_log_z(y, n_states=None, init_state=None) -> float:
    """Forward algorithm in float64 (the quantity discrete_hmm.py:118-143 computes)."""
    trans, obs = hmm_tables(n_states)
    k = trans.shape[0]
    pt, po = _softmax64(trans), _softmax64(obs)
    alpha = np.zeros(k)
    alpha[(HMM["init_state"] if init_state is None else init_state) % k] = 1.0
    ll = 0.0
    for yt in y:
        alpha = (alpha @ pt) * po[:, yt]
        s = alpha.sum()
        ll += math.log(s)
        alpha /= s
    return ll


class HmmSMC:
    """Reusable state of the C5 workload (tables resident on the device, data and keys prepared)."""

    def __init__(self, ops: Ops, impl: int, seed: int, n: int, T: int, n_states=None, want_ancestors: bool = False,
                 filters: int = 1, ess_threshold: float = 0.0):
        trans, obs = hmm_tables(n_states)
        self.ops, self.impl, self.n, self.T, self.want_ancestors, self.filters = ops, impl, n, T, want_ancestors, filters
        self.ess = ess_threshold
        self.k = trans.shape[0]
        self.init = HMM["init_state"] % self.k
        self.y = hmm_data(T, n_states)
        dev = ops.device()
        self.tl = torch.from_numpy(trans).to(dev).contiguous()
        self.ol = torch.from_numpy(obs).to(dev).contiguous()
        self.sk, self.rk = filter_key_schedules(seed, impl, T, filters)
        self.log_z_exact = hmm_exact_log_z(self.y, n_states)

    def run(self):
        return self.ops.smc_run_hmm(self.impl, self.n, self.sk, self.rk, self.k, self.init, self.tl, self.ol, self.y,
                                    self.want_ancestors, ess_threshold=self.ess, want_flags=True)

    def result(self, out):
        return smc_result(self.ops, out, self.n, self.filters, self.log_z_exact)


def hmm_smc(ops: Ops, impl: int, seed: int, n: int, T: int, n_states=None, want_ancestors: bool = False, **kw):
    w = HmmSMC(ops, impl, seed, n, T, n_states, want_ancestors, **kw)
    return w.result(w.run())
