"""Randomised HIP-vs-oracle parity of the generic resamplers and weight reductions (one MI355X):
    python tests/fuzz_resample.py [seconds] [seed]
Random population sizes (1 .. 3e6, ragged), output sizes, both generators, and weight patterns that steer the collapse-proof
paths of `resample_body`: flat, one / a few particles with most of the mass (heavy tiles with and without idle tiles),
geometric decay, long runs of -inf, a heavy LAST tile, everything in one tile.  Ancestors, maxima and fixed-point sums
must be equal bit for bit.  (test infrastructure: it loads the oracle; `test_random_resampling_fuzz` runs it for 20 s)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genjax._amd.abi import GjxLib  # noqa: E402
from genjax._amd.ops import KeyBatch, Ops  # noqa: E402
from genjax._amd.runtime import load_hip_ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
hip = load_hip_ops()
ora = Ops(GjxLib(os.path.join(ROOT, "oracle", "libgjx_oracle.so"), "cpu"))


def weights(n):
    kind = rng.choice(["flat", "normal", "one", "few", "decay", "holes", "last", "tile", "two_scales"])
    lw = rng.normal(0, 1.0, n).astype(np.float32)
    if kind == "flat":
        lw[:] = 0.0
    elif kind == "one":
        lw[int(rng.integers(n))] += float(rng.uniform(5, 40))
    elif kind == "few":
        for _ in range(int(rng.integers(2, 40))):
            lw[int(rng.integers(n))] += float(rng.uniform(3, 25))
    elif kind == "decay":
        lw -= np.arange(n, dtype=np.float32) * float(rng.uniform(1e-5, 1e-2))
    elif kind == "holes":
        a = int(rng.integers(n)); b = min(n, a + int(rng.integers(1, max(2, n // 2))))
        lw[a:b] = -np.inf
        if not np.isfinite(lw).any():
            lw[0] = 0.0
    elif kind == "last":
        lw[max(0, n - int(rng.integers(1, 1500))):] += float(rng.uniform(8, 30))
    elif kind == "tile":
        a = int(rng.integers(n)); lw[:] = -80.0; lw[a:a + int(rng.integers(1, 1024))] = 0.0
    else:
        lw[rng.random(n) < 0.01] += float(rng.uniform(4, 12))
    return kind, torch.from_numpy(lw)


t_end, cases = time.time() + budget, 0
while time.time() < t_end:
    n = int(rng.choice([1, 2, 255, 1024, 1025, 4097, 30000, 200_000, 1_000_000, 2_200_000, int(rng.integers(1, 3_000_000))]))
    n_out = n if rng.random() < 0.6 else int(rng.choice([1, 7, 1024, 50_000, int(rng.integers(1, 3_000_000))]))
    impl = int(rng.integers(2))
    kind, lw = weights(n)
    key = KeyBatch(impl, 2, parent=(int(rng.integers(1 << 30)), cases))
    ctx = dict(case=cases, n=n, n_out=n_out, impl=impl, weights=kind)
    for how in ("systematic", "multinomial"):
        if how == "multinomial" and n_out > 300_000:
            continue  # (the oracle's multinomial walk is O(n_out log n): keep the case quick)
        ha, hm, hq = hip.resample(how, key, lw.cuda(), n_out)
        oa, om, oq = ora.resample(how, key, lw, n_out)
        if not (torch.equal(ha.cpu(), oa) and torch.equal(hq.cpu(), oq) and torch.equal(hm.cpu().view(torch.int32), om.view(torch.int32))):
            d = (ha.cpu() != oa).nonzero().flatten()[:5]
            print("MISMATCH", how, ctx, "first differing slots", d.tolist(), ha.cpu()[d].tolist(), oa[d].tolist())
            sys.exit(1)
    for a, b in zip(hip.logsumexp(lw.cuda()), ora.logsumexp(lw)):
        if not torch.equal(a.cpu().view(torch.int32) if a.dtype == torch.float32 else a.cpu(), b.view(torch.int32) if b.dtype == torch.float32 else b):
            print("MISMATCH logsumexp", ctx)
            sys.exit(1)
    # a short LGSSM filter whose observations jump (weight collapse at random steps), every-step or ESS-adaptive
    if cases % 3 == 0:
        from genjax._amd import abi, prng, workloads as W

        T = int(rng.integers(2, 7))
        nn = int(rng.choice([1024, 5000, 70_000, 300_000]))
        y = rng.normal(0, 1, T).astype(np.float32)
        y[rng.random(T) < 0.4] *= float(rng.uniform(5, 60))
        mdl = abi.Lgssm(0.0, 1.0, float(rng.uniform(0.5, 1.0)), float(rng.uniform(0.3, 1.5)), float(rng.choice([0.05, 0.5, 2.0])))
        sk, rk = W.smc_key_schedule(prng.key(int(rng.integers(1 << 30)), impl), T)
        ess = float(rng.choice([0.0, 0.0, 0.5, 0.9]))
        h = hip.smc_run_lgssm(impl, nn, sk, rk, mdl, y, True, ess_threshold=ess, want_flags=True)
        o = ora.smc_run_lgssm(impl, nn, sk, rk, mdl, y, True, ess_threshold=ess, want_flags=True)
        for i, (a, b) in enumerate(zip(h, o)):
            if a is None and b is None:
                continue
            a, b = a.cpu(), b.cpu()
            same = torch.equal(a, b) or (a.dtype.is_floating_point and torch.equal(a.view(torch.int32), b.view(torch.int32)))
            if not same:
                print("MISMATCH filter output", i, dict(ctx, T=T, nn=nn, ess=ess, y=y.tolist()))
                sys.exit(1)
    cases += 1
print(f"resample fuzz ok: {cases} random cases, HIP == oracle bit for bit")
