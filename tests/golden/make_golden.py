#!/usr/bin/env python3
"""Generates tests/golden/*.json.  Run from the repo root: python tests/golden/make_golden.py

Sources of the expected values (none come from the reference: it cannot be imported here,
SURVEY F6, and its tests hold no random golden vectors):
  rng_kat.json        Random123 known-answer vectors for Threefry2x32-20 / Philox4x32-10
                      (Salmon et al. SC'11 distribution, kat_vectors) — literals below.
  logpdf_scipy.json   scipy.stats float64 log-densities on a fixed grid.
  reference_kat.json  closed-form answers the reference's own tests assert
                      (tests/inference/test_smc.py:32-87, tests/generative_functions/
                      test_static_gen_fn.py:317-318, README.md:121-123 analytic means).
  oracle_regression.json  outputs of the in-repo oracle for fixed counters — regression pins of
                      the arithmetic spec (both the oracle and the HIP library must reproduce them).
  oracle_regression_r02.json  the same for what round 2 added: the one-launch scan (LGSSM and HMM kernels), the
                      ESS-adaptive schedule, launch parameters, the collapsing-weights filter.
"""
import json
import math
import os
import sys

import numpy as np
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


dump("rng_kat.json", {
    "threefry2x32_20": [
        {"key": ["00000000", "00000000"], "ctr": ["00000000", "00000000"], "out": ["6b200159", "99ba4efe"]},
        {"key": ["ffffffff", "ffffffff"], "ctr": ["ffffffff", "ffffffff"], "out": ["1cb996fc", "bb002be7"]},
        {"key": ["13198a2e", "03707344"], "ctr": ["243f6a88", "85a308d3"], "out": ["c4923a9c", "483df7a0"]},
    ],
    "philox4x32_10": [
        {"key": ["00000000", "00000000"], "ctr": ["00000000"] * 4,
         "out": ["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]},
        {"key": ["ffffffff", "ffffffff"], "ctr": ["ffffffff"] * 4,
         "out": ["408f276d", "41c83b0e", "a20bc7c6", "6d5451fd"]},
        {"key": ["a4093822", "299f31d0"], "ctr": ["243f6a88", "85a308d3", "13198a2e", "03707344"],
         "out": ["d16cfe09", "94fdcceb", "5001e420", "24126ea1"]},
    ],
})

xs = [-3.0, -0.5, 0.0, 0.3, 1.5, 4.0]
pos = [0.01, 0.3, 1.0, 1.5, 7.0]
unit = [0.001, 0.3, 0.5, 0.9, 0.999]
dump("logpdf_scipy.json", {
    "normal": [{"x": x, "loc": l, "scale": s, "logpdf": float(stats.norm.logpdf(x, l, s))}
               for x in xs for l, s in [(0.0, 1.0), (1.0, 0.5), (-2.0, 3.0)]],
    "gamma": [{"x": x, "concentration": a, "rate": b, "logpdf": float(stats.gamma.logpdf(x, a, scale=1 / b))}
              for x in pos for a, b in [(2.0, 3.0), (0.5, 1.0), (9.0, 0.5)]],
    "beta": [{"x": x, "a": a, "b": b, "logpdf": float(stats.beta.logpdf(x, a, b))}
             for x in unit for a, b in [(2.0, 2.0), (0.5, 3.0), (5.0, 1.5)]],
    "bernoulli": [{"x": e, "p": p, "logpdf": float(math.log(p if e else 1 - p))} for e in (0, 1) for p in (0.1, 0.5, 0.7)],
    "categorical": [{"logits": [-0.3, -0.5], "logpdf": [-0.59813887, -0.79813887]}],
})

dump("reference_kat.json", {
    "flip_flip_trivial_logZ": math.log(0.7),
    "flip_flip_logZ": math.log(0.5 * 0.9 + 0.5 * 0.3),
    "assess_two_std_normals_y1_1_y2_m1": -2.837877,
    "normal_logpdf_0p5_0_1": -1.0439385332,
    "beta_bernoulli_posterior_mean_obs_true": 0.6,
    "beta_bernoulli_posterior_mean_obs_false": 0.4,
    "beta_bernoulli_logZ": math.log(0.5),
})

# ---- regression pins produced by the oracle ---------------------------------------------------
import torch  # noqa: E402
from genjax._amd.abi import GjxLib  # noqa: E402
from genjax._amd.ops import KeyBatch, Ops  # noqa: E402
from genjax._amd import workloads as W  # noqa: E402

ora = Ops(GjxLib(os.path.join(ROOT, "oracle", "libgjx_oracle.so"), "cpu"))
reg = {}
for impl, nm in ((0, "threefry"), (1, "philox")):
    kb = KeyBatch(impl, 1, parent=(0, 42), first=0)
    r = {}
    r["keys"] = ora.rng_keys(kb, 4).view(-1).tolist()
    r["bits_fold1"] = ora.rng_bits(kb.with_fold(1), 4).tolist()
    v, s = ora.sample_logpdf("normal", kb.with_fold(1), 4, 0.0, 1.0)
    r["normal_bits"] = v.view(torch.int32).tolist()
    r["normal_score_bits"] = s.view(torch.int32).tolist()
    v, s = ora.sample_logpdf("gamma", kb.with_fold(2), 4, 0.7, 2.0)
    r["gamma_bits"] = v.view(torch.int32).tolist()
    v, s = ora.sample_logpdf("beta", kb.with_fold(3), 4, 2.0, 2.0)
    r["beta_bits"] = v.view(torch.int32).tolist()
    v, s = ora.sample_logpdf("bernoulli", kb.with_fold(4), 8, 0.3)
    r["bernoulli"] = v.tolist()
    lw = torch.linspace(-3, 2, 37)
    a, m, q = ora.resample("systematic", KeyBatch(impl, 2, parent=(5, 6)), lw)
    r["systematic_ancestors"] = a.tolist()
    r["systematic_q"], r["systematic_e"] = int(q), int(m)
    a, _, _ = ora.resample("multinomial", KeyBatch(impl, 2, parent=(5, 6)), lw, 12)
    r["multinomial_ancestors"] = a.tolist()
    r["categorical_index_gumbel"] = int(ora.categorical_index(KeyBatch(impl, 2, parent=(5, 6)), lw, 0))
    r["categorical_index_invcdf"] = int(ora.categorical_index(KeyBatch(impl, 2, parent=(5, 6)), lw, 1))
    g = W.gaussian10_importance(ora, impl, seed=3, n=2048)
    r["gaussian10_q"], r["gaussian10_max_bits"] = g["q"], int(torch.tensor(g["max"], dtype=torch.float32).view(torch.int32))
    r["gaussian10_logw_head_bits"] = g["logw"][:4].view(torch.int32).tolist()
    s_ = W.lgssm_smc(ora, impl, seed=4, n=2048, T=6, want_ancestors=True)
    r["lgssm_q"], r["lgssm_e"] = s_["out_q"].tolist(), s_["out_e"].tolist()
    r["lgssm_anc_t5_head"] = s_["ancestors"][5, :16].tolist()
    h = W.hmm_smc(ora, impl, seed=5, n=2048, T=6, n_states=16)
    r["hmm_q"], r["hmm_e"] = h["out_q"].tolist(), h["out_e"].tolist()
    reg[nm] = r
dump("oracle_regression.json", reg)

# ---- round 2 additions (a separate file).  Round 3 revised the resampling weight spec (tile-anchored
# records, one launch per SMC step), so the resampling / SMC entries of BOTH files were regenerated then; everything else
# reproduced unchanged. ----
reg2 = {}
for impl, nm in ((0, "threefry"), (1, "philox")):
    r = {}
    sc = W.lgssm_scan(ora, impl, seed=6, n=1500, T=9)
    r["scan_lgssm_logw_head_bits"] = sc["logw"][:6].view(torch.int32).tolist()
    r["scan_lgssm_x_t8_head_bits"] = sc["x"][8, :6].view(torch.int32).tolist()
    hs = W.HmmScan(ora, impl, 7, 1200, 8, n_states=16, cat_mode=1)
    hs.run()
    hr = hs.result()
    r["scan_hmm_z_t7_head"] = hr["z"][7, :12].tolist()
    r["scan_hmm_logw_head_bits"] = hr["logw"][:4].view(torch.int32).tolist()
    ad = W.lgssm_smc(ora, impl, seed=8, n=3000, T=16, want_ancestors=True, ess_threshold=0.5)
    r["ess_flags"] = ad["resampled"].tolist()
    r["ess_q"], r["ess_e"] = ad["out_q"].tolist(), ad["out_e"].tolist()
    r["ess_logw_head_bits"] = ad["logw"][:4].view(torch.int32).tolist()
    from genjax._amd import abi, prng  # noqa: E402
    import numpy as np  # noqa: E402,F811

    y = np.array([0.1, 25.0, -40.0, -39.5, 60.0, 60.2], dtype=np.float32)
    sk, rk = W.smc_key_schedule(prng.key(11, impl), 6)
    col = ora.smc_run_lgssm(impl, 20000, sk, rk, abi.Lgssm(0.0, 1.0, 0.9, 1.0, 0.05), y, True)
    r["collapse_q"], r["collapse_e"] = col[1].tolist(), col[0].tolist()
    r["collapse_anc_t2_distinct"] = int(col[4][2].unique().numel())
    r["collapse_anc_t5_head"] = col[4][5, :8].tolist()
    reg2[nm] = r
dump("oracle_regression_r02.json", reg2)
