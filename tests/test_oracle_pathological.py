"""The oracle on values outside every domain, without a GPU: defined results, indices in range, no
undefined float -> integer conversion.  The GPU suite compares the HIP library with the oracle on the same inputs
(test_gpu_parity_abi.py: test_pathological_weights, test_sites_with_invalid_parameters, ...); this file is what the
sanitizer build of the oracle (`make -C oracle sanitize`, ASan + UBSan + float-cast-overflow) runs over them."""

import numpy as np
import pytest
import torch

from genjax._amd import abi, prng, workloads as W
from genjax._amd.ops import KeyBatch

BAD = [float("nan"), float("inf"), float("-inf"), 0.0, -1.0, 1e-45, 3e38]


def weights(n):
    g = torch.Generator().manual_seed(0)
    lw = torch.randn(n, generator=g) * 2
    cases = {"all -inf": torch.full((n,), float("-inf")), "all nan": torch.full((n,), float("nan")),
             "all -3e38": torch.full((n,), -3e38), "huge": lw * 1e37}
    a = lw.clone(); a[::7] = float("nan"); cases["some nan"] = a
    a = lw.clone(); a[123] = float("inf"); cases["one +inf"] = a
    a = lw.clone(); a[123] = float("inf"); a[4000] = float("inf"); a[9] = float("nan"); cases["two +inf, nan"] = a
    return cases


@pytest.mark.parametrize("impl", [0, 1])
def test_weights(oracle_ops, impl):
    n = 5000
    for name, w in weights(n).items():
        key = KeyBatch(impl, 2, parent=(5, 1))
        for kind in ("systematic", "multinomial"):
            anc, m, q = oracle_ops.resample(kind, key, w, n)
            assert 0 <= int(anc.min()) and int(anc.max()) < n, name
        for mode in (0, 1):
            assert 0 <= int(oracle_ops.categorical_index(key, w, mode)) < n
        oracle_ops.logsumexp(w)
        oracle_ops.lse_rows(oracle_ops.row_stats(w))
    anc, _, _ = oracle_ops.resample("systematic", KeyBatch(impl, 2, parent=(5, 1)), weights(n)["all nan"], n)
    assert torch.equal(anc, torch.arange(n, dtype=anc.dtype))  # zero total mass: the population is kept
    anc, _, _ = oracle_ops.resample("systematic", KeyBatch(impl, 2, parent=(5, 1)), weights(n)["all -inf"], n)
    assert torch.equal(anc, torch.arange(n, dtype=anc.dtype))  # likewise
    anc, _, _ = oracle_ops.resample("systematic", KeyBatch(impl, 2, parent=(5, 1)), weights(n)["all -inf"], 3 * n)
    assert torch.equal(anc, torch.arange(3 * n, dtype=anc.dtype) // 3)  # ... floor(j n / n_out) when n_out != n
    anc, _, _ = oracle_ops.resample("systematic", KeyBatch(impl, 2, parent=(5, 1)), weights(n)["one +inf"], n)
    assert int(anc.min()) == int(anc.max()) == 123


@pytest.mark.parametrize("impl", [0, 1])
def test_sites(oracle_ops, impl):
    n = 400
    key = KeyBatch(impl, 2, parent=(8, 1))
    for dist in ("normal", "gamma", "beta", "bernoulli"):
        for a in BAD + [1.5]:
            for b in ([None] if dist == "bernoulli" else BAD + [0.7]):
                oracle_ops.sample_logpdf(dist, key, n, a, b)
                for v in ((0.3, -2.0, float("nan"), float("inf"), 0.0, 1.0) if dist != "bernoulli" else (0, 1)):
                    oracle_ops.logpdf(dist, n, v, a, b) if dist != "bernoulli" else oracle_ops.logpdf(dist, n, v, a)
    assert torch.isnan(oracle_ops.logpdf("normal", 4, 0.0, 0.0, float("nan"))).all()
    assert torch.isnan(oracle_ops.logpdf("normal", 4, 0.0, 0.0, -1.0)).all()          # log of a negative scale
    assert bool((oracle_ops.logpdf("gamma", 4, 1.0, -1.0, 1.0) == float("-inf")).all())  # lgamma(-1) = +inf
    K = 7
    g = torch.Generator().manual_seed(3)
    base = torch.randn(1, K, generator=g)
    for logits in (torch.full((1, K), float("-inf")), torch.full((1, K), float("nan")), base * 1e38,
                   base.clone().index_fill_(1, torch.tensor([4]), float("inf"))):
        for mode in (0, 1):
            v, s = oracle_ops.sample_logpdf_categorical(KeyBatch(impl, 2, parent=(8, 2)), n, logits, mode=mode)
            assert 0 <= int(v.min()) and int(v.max()) < K
        for v in (0, 4, -1, K):
            oracle_ops.logpdf_categorical(n, v, logits)


@pytest.mark.parametrize("impl", [0, 1])
def test_filters_and_plans(oracle_ops, impl):
    n = 3000
    for bad in (float("nan"), float("inf"), 1e30):
        y = np.array([0.1, bad, 0.3, 0.2], dtype=np.float32)
        sk, rk = W.smc_key_schedule(prng.key(11, impl), 4)
        out = oracle_ops.smc_run_lgssm(impl, n, sk, rk, abi.Lgssm(0.0, 1.0, 0.9, 1.0, 0.5), y, True)
        assert 0 <= int(out[4].min()) and int(out[4].max()) < n
    K = 12
    g = torch.Generator().manual_seed(4)
    t, o = torch.randn(K, K, generator=g), torch.randn(K, K, generator=g)
    t[2, 7] = float("nan"); t[4, :] = float("nan"); t[5, :] = float("-inf"); t[:, ::3] = float("-inf")
    o[1, :] = float("nan"); o[:, 8] = float("inf"); o[:, 3] = float("-inf")
    sk, rk = W.smc_key_schedule(prng.key(13, impl), 6)
    out = oracle_ops.smc_run_hmm(impl, n, sk, rk, K, 5, t, o, np.array([3, 1, 8, 3, 0, 11], dtype=np.int32), True)
    assert 0 <= int(out[2].min()) and int(out[2].max()) < K and 0 <= int(out[4].min()) and int(out[4].max()) < n
    kb = W.importance_particle_keys(prng.key(5, impl), n)
    for variant in range(4):
        sites = W.gaussian10_sites(W.gaussian10_data())[:8]
        if variant == 0:
            sites[1].obs = abi.Arg(abi.ARG_CONST, 0, 0.0, float("nan"), None)
        elif variant == 1:
            sites[3].arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, 0.0, None)
        elif variant == 2:
            sites[2].arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, -1.0, None)
        else:
            sites[4].arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, float("inf"), None)
        plan = oracle_ops.plan_create(sites)
        vals, score, logw, mp = oracle_ops.importance_run(plan, kb, n, [], [torch.float32] * 4)
        oracle_ops.lse_rows(oracle_ops.row_stats(logw))
        oracle_ops.logsumexp(logw)
