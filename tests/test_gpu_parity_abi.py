"""GPU parity tests: every C-ABI entry point of libgjx_hip.so against the CPU oracle on the same
seeded inputs.  Integer / index outputs and — because the math spec fixes every rounding — all f32
outputs are compared BIT-EXACTLY (tolerance 0), which is stronger than the north star's 1e-5
relative bound on log-weights."""

import math
import os

import numpy as np
import pytest
import torch

from genjax._amd import abi, prng, workloads as W
from genjax._amd.ops import KeyBatch

pytestmark = pytest.mark.gpu

IMPLS = [0, 1]
SIZES = [1, 63, 1024, 1025, 40000]


def dev(t, ops):
    return t.to(ops.device()).contiguous()


def same(a, b, what=""):
    a, b = a.cpu(), b.cpu()
    if a.dtype.is_floating_point:
        ok = torch.equal(a.view(torch.int32), b.view(torch.int32)) or torch.equal(a, b)
    else:
        ok = torch.equal(a, b)
    if not ok:
        bad = (a != b).nonzero().flatten()[:5]
        raise AssertionError(f"{what}: {int((a != b).sum())} of {a.numel()} differ, first at {bad.tolist()}: "
                             f"{a.flatten()[bad].tolist()} vs {b.flatten()[bad].tolist()}")


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", SIZES)
def test_rng_keys_and_bits(hip_ops, oracle_ops, impl, n):
    kb = KeyBatch(impl, 1, parent=(0x13198A2E, 0x03707344), first=(1 << 33) + 5)
    for fold in (None, 1, 0xFFFFFFFF):
        k = kb if fold is None else kb.with_fold(fold)
        same(hip_ops.rng_keys(k, n), oracle_ops.rng_keys(k, n), "rng_keys")
        for sub in (0, 7):
            same(hip_ops.rng_bits(k, n, sub), oracle_ops.rng_bits(k, n, sub), "rng_bits")
    # explicit keys (mode 0) must reproduce the lazy batch
    mat = oracle_ops.rng_keys(kb, n)
    ek_o = KeyBatch(impl, 0, tensor=mat)
    ek_h = KeyBatch(impl, 0, tensor=dev(mat, hip_ops))
    same(hip_ops.rng_bits(ek_h.with_fold(2), n), oracle_ops.rng_bits(kb.with_fold(2), n), "explicit vs lazy")
    same(oracle_ops.rng_bits(ek_o.with_fold(2), n), oracle_ops.rng_bits(kb.with_fold(2), n))
    lit = KeyBatch(impl, 2, parent=(9, 10))
    same(hip_ops.rng_bits(lit, n, 3), oracle_ops.rng_bits(lit, n, 3), "literal key")
    for m in (1, 3):
        same(hip_ops.rng_split_each(kb, n, m), oracle_ops.rng_split_each(kb, n, m), "rng_split_each")
        same(hip_ops.rng_split_each(ek_h, n, m), oracle_ops.rng_split_each(kb, n, m), "rng_split_each explicit")
    se = oracle_ops.rng_split_each(kb, n, 3).view(n, 3, -1)
    assert torch.equal(se[:, 0], oracle_ops.rng_keys(KeyBatch(impl, 0, tensor=mat).with_fold(0), n)) or impl == 1
    if impl == 1:  # philox keys with a lane: as parent of a lazy batch (children are hashed) and as a literal
        for lk in (KeyBatch(1, 1, parent=(7, 8), first=3, parent_lane=(1 << 40) + 9),
                   KeyBatch(1, 2, parent=(7, 8), parent_lane=12345)):
            for k in (lk, lk.with_fold(6)):
                same(hip_ops.rng_keys(k, n), oracle_ops.rng_keys(k, n), "laned rng_keys")
                same(hip_ops.rng_bits(k, n, 0), oracle_ops.rng_bits(k, n, 0), "laned rng_bits")
                same(hip_ops.rng_bits(k, n, 5), oracle_ops.rng_bits(k, n, 5), "laned rng_bits sub")
            same(hip_ops.rng_split_each(lk, n, 2), oracle_ops.rng_split_each(lk, n, 2), "laned split_each")


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", SIZES)
def test_sample_logpdf_scalar_args(hip_ops, oracle_ops, impl, n):
    kb = KeyBatch(impl, 1, parent=(1, 2), first=0).with_fold(4)
    for dist, a, b in [("normal", 0.3, 1.7), ("gamma", 2.5, 3.0), ("gamma", 0.3, 0.7), ("beta", 2.0, 2.0),
                       ("beta", 0.5, 3.0), ("bernoulli", 0.3, None)]:
        hv, hs = hip_ops.sample_logpdf(dist, kb, n, a, b)
        ov, os_ = oracle_ops.sample_logpdf(dist, kb, n, a, b)
        same(hv, ov, f"{dist} value")
        same(hs, os_, f"{dist} score")


@pytest.mark.parametrize("impl", IMPLS)
def test_sample_logpdf_tensor_args(hip_ops, oracle_ops, impl):
    n = 30001
    g = torch.Generator().manual_seed(0)
    loc = torch.randn(n, generator=g)
    pos = torch.rand(n, generator=g) * 4 + 0.05
    pos2 = torch.rand(n, generator=g) * 4 + 0.05
    prob = torch.rand(n, generator=g)
    kb = KeyBatch(impl, 1, parent=(3, 4), first=10).with_fold(1)
    for dist, a, b in [("normal", loc, pos), ("gamma", pos, pos2), ("beta", pos, pos2), ("bernoulli", prob, None)]:
        hv, hs = hip_ops.sample_logpdf(dist, kb, n, dev(a, hip_ops), None if b is None else dev(b, hip_ops))
        ov, os_ = oracle_ops.sample_logpdf(dist, kb, n, a, b)
        same(hv, ov, f"{dist} value")
        same(hs, os_, f"{dist} score")
        # logpdf of the produced values reproduces the fused score
        hl = hip_ops.logpdf(dist, n, hv, dev(a, hip_ops), None if b is None else dev(b, hip_ops))
        ol = oracle_ops.logpdf(dist, n, ov, a, b)
        same(hl, ol, f"{dist} logpdf")
        same(hl, hs, f"{dist} logpdf == fused score")


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("mode", [0, 1])
def test_categorical(hip_ops, oracle_ops, impl, mode):
    n, k = 5000, 7
    g = torch.Generator().manual_seed(1)
    shared = torch.randn(1, k, generator=g)
    shared[0, 2] = float("-inf")  # a zero-probability category
    per = torch.randn(n, k, generator=g) * 2
    table = torch.randn(5, k, generator=g)
    rows = torch.randint(0, 5, (n,), generator=g, dtype=torch.int32)
    kb = KeyBatch(impl, 1, parent=(8, 9), first=0).with_fold(2)
    for logits, ri in [(shared, None), (per, None), (table, rows)]:
        hv, hs = hip_ops.sample_logpdf_categorical(kb, n, dev(logits, hip_ops), None if ri is None else dev(ri, hip_ops), mode)
        ov, os_ = oracle_ops.sample_logpdf_categorical(kb, n, logits, ri, mode)
        same(hv, ov, "categorical value")
        same(hs, os_, "categorical score")
        hl = hip_ops.logpdf_categorical(n, hv, dev(logits, hip_ops), None if ri is None else dev(ri, hip_ops))
        same(hl, hs, "categorical logpdf == fused score")
        if logits is shared:
            assert not bool((hv.cpu() == 2).any()), "a -inf logit must never be drawn"


@pytest.fixture(params=["specialized", "pair", "interpreter"])
def plan_mode(request, monkeypatch):
    """The importance kernels: the hiprtc-specialised straight-line kernel (default: Philox lazy batches take the
    form with FOUR adjacent particles per lane, one wave per 256-particle row, whenever n and the buffers are 16-byte
    aligned), the same with two particles per lane (GJX_JIT_FORM=pair) and the site-table interpreter
    (GJX_PLAN_JIT=0).  All three must give the oracle's bits."""
    monkeypatch.setenv("GJX_PLAN_JIT", "0" if request.param == "interpreter" else "1")
    if request.param == "pair":
        monkeypatch.setenv("GJX_JIT_FORM", "pair")
    else:
        monkeypatch.delenv("GJX_JIT_FORM", raising=False)
    return request.param


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", [1, 1000, 1024, 70001, 70004])
def test_importance_gaussian10(hip_ops, oracle_ops, impl, n, plan_mode):
    h = W.gaussian10_importance(hip_ops, impl, seed=11, n=n)
    o = W.gaussian10_importance(oracle_ops, impl, seed=11, n=n)
    same(h["logw"], o["logw"], "logw")
    same(h["score"], o["score"], "score")
    for a, b in zip(h["values"], o["values"]):
        same(a, b, "latent column")
    assert h["q"] == o["q"] and h["max"] == o["max"] and h["lse"] == o["lse"]
    # row-anchored partial sums emitted by the importance kernel itself
    same(h["rows"].e, o["rows"].e, "row anchors"); same(h["rows"].s, o["rows"].s, "row sums")
    assert (h["row_e"], h["row_q"], h["row_lse"]) == (o["row_e"], o["row_q"], o["row_lse"])
    assert abs(h["log_z_rows"] - h["log_z"]) < 1e-6, "the two fixed-point forms agree to fixed-point resolution"


@pytest.mark.parametrize("impl", IMPLS)
def test_importance_key_forms_agree(hip_ops, oracle_ops, impl):
    """One population, every way of handing over its keys: the paired kernel (lazy batch, even first), the
    one-particle-per-lane kernel (odd first: pairs straddle lanes), explicit key arrays, and the eager
    per-site kernel all draw the same values — Philox Normal sites pair particles by key lane, wherever the
    partner's word has to be derived."""
    import torch as T

    n, first = 3001, 10
    sites = W.gaussian10_sites(W.gaussian10_data())
    root = prng.key(77, impl)
    k2 = prng.split(prng.split(root)[1])[1]

    def run(ops, kb, m):
        plan = ops.plan_create(sites)
        vals, score, logw, _ = ops.importance_run(plan, kb, m, [], [T.float32] * W.G10_LATENTS)
        return T.stack([v.cpu() for v in vals]), score.cpu(), logw.cpu()

    lazy = prng.split_lazy(k2, n, first)
    ref = run(oracle_ops, lazy, n)
    got = run(hip_ops, lazy, n)
    for a, b in zip(got, ref):
        same(a, b, "lazy, even first (paired kernel)")
    odd = run(hip_ops, prng.split_lazy(k2, n - 1, first + 1), n - 1)  # the same particles minus the first one
    for a, b in zip(odd, ref):
        same(a, b[..., 1:], "lazy, odd first (generic kernel)")
    mat = hip_ops.rng_keys(lazy, n)
    exp = run(hip_ops, KeyBatch(impl, 0, tensor=mat), n)
    for a, b in zip(exp, ref):
        same(a, b, "explicit key array")
    # eager per-site kernel, site 0 (fold: threefry counter 1, philox draw index 0)
    v0, _ = hip_ops.sample_logpdf("normal", lazy.with_fold(1 if impl == 0 else 0), n, 0.0, 1.0)
    same(v0, ref[0][0], "eager site kernel")
    # a scalar run with keys[i] draws what element i of the batch draws
    i = 7
    one = run(hip_ops, prng.split_at(k2, first + i).literal(), 1)
    same(one[0][:, 0], ref[0][:, i], "scalar key == batch element")


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", [2, 1000, 1001, 70002, 70004])
def test_importance_passes_in_one_launch(hip_ops, oracle_ops, impl, n, plan_mode):
    """gjx_importance_run_batch: L independent passes (seeds s, s+1, ...) in one launch write what L separate
    passes write — trace columns, scores, log-weights, row sums and folded log-marginals — and equal the
    oracle's passes bit for bit; n odd takes the one-particle-per-lane kernel, the interpreter one launch per pass."""
    L = 3
    wl_h = W.Gaussian10(hip_ops, impl, seed=21, n_local=n)
    prep = wl_h.prepare(fold_batch=2 * L, passes=L)
    prep.launch_passes(L, L)  # second half of the slots
    prep.launch_fold(2 * L)
    for p_ in range(L):
        ref = W.Gaussian10(oracle_ops, impl, seed=21 + p_, n_local=n).step()
        same(prep.logw_all[p_, :n], ref["logw"], f"logw pass {p_}")
        same(prep.score_all[p_, :n], ref["score"], f"score pass {p_}")
        for c, col in enumerate(ref["values"]):
            same(prep.values_all[c][p_, :n], col, f"column {c} pass {p_}")
        same(prep.row_e_all[L + p_], ref["rows"].e, "row anchors"); same(prep.row_s_all[L + p_], ref["rows"].s, "row sums")
        same(prep.lse_all[L + p_:L + p_ + 1], ref["row_lse"], "folded lse")
        same(prep.e_all[L + p_:L + p_ + 1], ref["row_e"]); same(prep.q_all[L + p_:L + p_ + 1], ref["row_q"])
    # a ragged launch (2 of 3 passes) into the first slots
    prep.launch_passes(0, 2)
    prep.launch_fold(2)
    for p_ in range(2):
        same(prep.q_all[p_:p_ + 1], W.Gaussian10(oracle_ops, impl, seed=21 + p_, n_local=n).step()["row_q"], "ragged launch")


@pytest.mark.parametrize("impl", IMPLS)
def test_importance_mixed_plan(hip_ops, oracle_ops, impl, plan_mode):
    """A plan touching every distribution and argument kind: beta-bernoulli, gamma-scaled normal,
    categorical selecting a table row, input columns, observed input column."""
    n = 20000
    g = torch.Generator().manual_seed(3)
    xin = torch.randn(n, generator=g)
    yobs = torch.randn(n, generator=g)
    means = torch.tensor([0.0, 10.0, 11.0])
    logits = torch.log(torch.tensor([[0.5, 0.25, 0.25]]))

    def build(ops):
        A = abi.Arg
        s = []
        p = abi.Site(); p.dist, p.observed, p.out_col = abi.DIST_BETA, 0, 0
        p.arg[0], p.arg[1] = A(abi.ARG_CONST, 0, 0, 2.0, None), A(abi.ARG_CONST, 0, 0, 2.0, None); s.append(p)
        v = abi.Site(); v.dist, v.observed, v.out_col = abi.DIST_BERNOULLI, 1, -1
        v.arg[0] = A(abi.ARG_SITE, 0, 1.0, 0.0, None); v.obs = A(abi.ARG_CONST, 0, 0, 1.0, None); s.append(v)
        gm = abi.Site(); gm.dist, gm.observed, gm.out_col = abi.DIST_GAMMA, 0, 1
        gm.arg[0], gm.arg[1] = A(abi.ARG_CONST, 0, 0, 0.7, None), A(abi.ARG_SITE, 0, 2.0, 0.5, None); s.append(gm)
        c = abi.Site(); c.dist, c.observed, c.out_col = abi.DIST_CATEGORICAL, 0, 2
        c.n_cat, c.n_rows, c.cat_mode = 3, 1, 1
        c.arg[0] = A(abi.ARG_CONST, 0, 0, 0.0, None)
        lg = dev(logits, ops); c.logits = lg.data_ptr(); s.append(c)
        mt = dev(means, ops)
        x = abi.Site(); x.dist, x.observed, x.out_col = abi.DIST_NORMAL, 0, 3
        x.arg[0], x.arg[1] = A(abi.ARG_TABLE, 3, 0, 0, mt.data_ptr()), A(abi.ARG_SITE, 2, 1.0, 0.1, None); s.append(x)
        y = abi.Site(); y.dist, y.observed, y.out_col = abi.DIST_NORMAL, 1, -1
        y.arg[0], y.arg[1] = A(abi.ARG_INPUT, 0, 0.5, 1.0, None), A(abi.ARG_CONST, 0, 0, 2.0, None)
        y.obs = A(abi.ARG_INPUT, 1, 0, 0, None); s.append(y)
        fl = abi.Site(); fl.dist, fl.observed, fl.out_col = abi.DIST_BERNOULLI, 0, 4
        fl.arg[0] = A(abi.ARG_CONST, 0, 0, 0.25, None); s.append(fl)
        return ops.plan_create(s), (lg, mt)

    kb = KeyBatch(impl, 1, parent=(21, 22), first=7)
    dts = [torch.float32, torch.float32, torch.int32, torch.float32, torch.int32]
    ph, keep_h = build(hip_ops)
    po, keep_o = build(oracle_ops)
    hv, hs, hw, hmp = hip_ops.importance_run(ph, kb, n, [dev(xin, hip_ops), dev(yobs, hip_ops)], dts)
    ov, os_, ow, omp = oracle_ops.importance_run(po, kb, n, [xin, yobs], dts)
    for a, b in zip(hv, ov):
        same(a, b, "value column")
    same(hs, os_, "score")
    same(hw, ow, "logw")
    same(hmp, omp, "max partials")


def test_map_f32(hip_ops, oracle_ops):
    """gjx_map_f32: the spec's exp / log and the IEEE division by (of) a number over a column, HIP == oracle bit for bit
    (edge cases included: zeros, infinities, NaN, subnormals, both ends of exp's range)."""
    from genjax._amd import abi
    from test_oracle_pinning import map_inputs

    x = torch.cat([map_inputs(), torch.randn(1 << 20) * 30])
    xd = dev(x, hip_ops)
    for op in (abi.MAP_EXP, abi.MAP_LOG, abi.MAP_ABS):
        same(hip_ops.map_f32(op, xd), oracle_ops.map_f32(op, x), f"map {op}")
    same_or_both_nan(hip_ops.map_f32(abi.MAP_SQRT, xd), oracle_ops.map_f32(abi.MAP_SQRT, x), "sqrt")  # (the NaN of a negative argument: its sign is the platform's)
    for c in (3.0, 0.1, -7.25, 1e-30, 2.263):
        same(hip_ops.map_f32(abi.MAP_DIV, xd, c), oracle_ops.map_f32(abi.MAP_DIV, x, c), "x / c")
        same(hip_ops.map_f32(abi.MAP_RDIV, xd, c), oracle_ops.map_f32(abi.MAP_RDIV, x, c), "c / x")


@pytest.mark.parametrize("n", [1, 5, 1023, 1024, 1025, 50000, 1 << 20])
def test_logsumexp(hip_ops, oracle_ops, n):
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * 5 - 3
    if n > 5:
        x[3] = float("-inf")
    hl, hm, hq = hip_ops.logsumexp(dev(x, hip_ops))
    ol, om, oq = oracle_ops.logsumexp(x)
    same(hm, om, "max"); same(hq, oq, "q"); same(hl, ol, "lse")
    ref = torch.logsumexp(x.double(), 0)
    assert abs(float(hl.cpu()) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    # split form used by the multi-device path
    m = hip_ops.max_f32(dev(x, hip_ops), n)
    frac = hip_ops.frac_bits(n)
    q = hip_ops.expsum_fix(dev(x, hip_ops), m, frac)
    same(q, oq, "expsum_fix"); same(hip_ops.lse_finish(m, q, frac), ol, "lse_finish")
    # row-anchored form
    hr, orr = hip_ops.row_stats(dev(x, hip_ops)), oracle_ops.row_stats(x)
    same(hr.e, orr.e, "row_stats e"); same(hr.s, orr.s, "row_stats s")
    for a, b in zip(hip_ops.lse_rows(hr), oracle_ops.lse_rows(orr)):
        same(a, b, "lse_rows")
    assert abs(float(hip_ops.lse_rows(hr)[0].cpu()) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", [1, 255, 256, 257, 5000, 1_000_000, 1_100_003])
def test_fused_lse_tail(hip_ops, oracle_ops, impl, n):
    """The importance launch folds its own row sums (last-workgroup tail): the fused (e, q, lse, record)
    equal a separate gjx_lse_rows over the rows it wrote, equal the oracle, on every one of many
    back-to-back launches sharing one ticket buffer (1_100_003 particles: > 4096 rows, the looped fold)."""
    from genjax._amd import abi

    wl_h = W.Gaussian10(hip_ops, impl, seed=4, n_local=n)
    prep = wl_h.prepare()
    reps = 40 if n <= 5000 else 12
    recs = torch.zeros((reps, abi.LSE_RECORD_WORDS), dtype=torch.int64, device=hip_ops.device())
    for r in range(reps):
        prep.launch_fused(record=recs[r])
    fused = (prep.lse.clone(), prep.row_e_out.clone(), prep.row_q_out.clone())
    ref_rec = torch.zeros(abi.LSE_RECORD_WORDS, dtype=torch.int64, device=hip_ops.device())
    prep.rows.lse = None
    sep = hip_ops.lse_rows(prep.rows, record=ref_rec)
    for a, b in zip(fused, sep):
        same(a, b, "fused vs separate fold")
    assert all(torch.equal(recs[r], ref_rec) for r in range(reps)), "a launch read stale row sums"
    assert int(prep._tickets.abs().sum().cpu()) == 0  # left zero for the next launch
    out = W.Gaussian10(oracle_ops, impl, seed=4, n_local=n).step()
    same(fused[0], out["row_lse"], "lse vs oracle"); same(fused[1], out["row_e"]); same(fused[2], out["row_q"])
    # the shifted output of the same launch (gjx_lse_out.lse_shifted): lse - shift as ONE f32 subtraction
    shifted = torch.empty(1, dtype=torch.float32, device=hip_ops.device())
    prep.launch_fused_shifted(shifted, 2.5)
    same(shifted, fused[0] - 2.5, "lse - shift"); same(prep.lse, fused[0], "lse beside the shifted output")
    o_prep = W.Gaussian10(oracle_ops, impl, seed=4, n_local=n).prepare()
    o_shift = torch.empty(1, dtype=torch.float32)
    o_prep.launch_fused_shifted(o_shift, 2.5)
    same(shifted, o_shift, "shifted lse vs oracle")


@pytest.mark.parametrize("k", [3, 17, 256])
def test_hmm_tables(hip_ops, oracle_ops, k):
    """gjx_hmm_prepare: alias tables and observation log-probabilities, HIP == oracle bit for bit (the
    distribution the tables encode is pinned on the CPU side: test_oracle_pinning.check_hmm_alias)."""
    from test_oracle_pinning import check_hmm_alias

    tl, ol = W.hmm_tables(k)
    tl, ol = torch.from_numpy(tl).contiguous(), torch.from_numpy(ol).contiguous()
    hc, hp = hip_ops.hmm_prepare(k, 0, dev(tl, hip_ops), dev(ol, hip_ops))
    oc, op_ = oracle_ops.hmm_prepare(k, 0, tl, ol)
    same(hc, oc, "trans_alias"); same(hp, op_, "obs_logp")
    check_hmm_alias(hip_ops, k)


def test_lse_records(hip_ops, oracle_ops):
    """gjx_lse_rows records / gjx_lse_combine: shards merge exactly, and HIP == oracle bit for bit."""
    from test_oracle_pinning import check_lse_records

    got = check_lse_records(hip_ops, lambda t: dev(t, hip_ops))
    want = check_lse_records(oracle_ops)
    for (gr, gl), (wr, wl) in zip(got, want):
        assert torch.equal(gr, wr) and torch.equal(gl, wl)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n,n_out", [(1, 1), (7, 20), (1024, 1024), (1025, 300), (30000, 30000), (200000, 200000)])
def test_resample_and_gather(hip_ops, oracle_ops, impl, n, n_out):
    g = torch.Generator().manual_seed(n)
    lw = torch.randn(n, generator=g) * 3
    if n > 10:
        lw[1] = float("-inf")
        lw[5] = 9.0  # one heavy particle: many offspring from a single source tile
    key = KeyBatch(impl, 2, parent=(5, n))
    for kind in ("systematic", "multinomial"):
        ha, hm, hq = hip_ops.resample(kind, key, dev(lw, hip_ops), n_out)
        oa, om, oq = oracle_ops.resample(kind, key, lw, n_out)
        same(ha, oa, f"{kind} ancestors"); same(hm, om); same(hq, oq)
    ha, _, _ = hip_ops.resample("systematic", key, dev(lw, hip_ops), n_out)
    a = ha.cpu()
    assert bool((a[1:] >= a[:-1]).all()), "systematic ancestors must be monotone"
    cnt = torch.bincount(a.long(), minlength=n).double()
    w = torch.softmax(lw.double(), 0) * n_out
    assert float((cnt - w).abs().max()) < 1.0 + 1e-6, "systematic offspring counts must be floor/ceil of N w"
    cols = [torch.randn(n, generator=g), torch.randint(0, 100, (n,), generator=g, dtype=torch.int32)]
    hg = hip_ops.gather_cols(ha, [dev(c, hip_ops) for c in cols])
    for c, o in zip(cols, hg):
        same(o, c[a.long()], "gather")


def pathological_weights(n):
    g = torch.Generator().manual_seed(n)
    lw = torch.randn(n, generator=g) * 2
    cases = {}
    cases["all -inf"] = torch.full((n,), float("-inf"))
    a = lw.clone(); a[::7] = float("nan"); cases["some nan"] = a
    cases["all nan"] = torch.full((n,), float("nan"))
    a = lw.clone(); a[n // 3] = float("inf"); cases["one +inf"] = a
    a = lw.clone(); a[n // 3] = float("inf"); a[n - 2] = float("inf"); a[9] = float("nan"); cases["two +inf and a nan"] = a
    cases["all -3e38"] = torch.full((n,), -3e38)
    return cases


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", [5000, 300000])
def test_pathological_weights(hip_ops, oracle_ops, impl, n):
    """Weights no filter should produce but a caller can pass: all -inf (uniform), NaNs (weight 0), +inf (all the mass),
    all NaN (zero total: every slot takes the last particle).  Defined by the spec's comparisons (fixw, teeth_below), the
    same on both sides, indices always in range."""
    for name, lw in pathological_weights(n).items():
        key = KeyBatch(impl, 2, parent=(5, n))
        for kind in ("systematic", "multinomial"):
            ha, hm, hq = hip_ops.resample(kind, key, dev(lw, hip_ops), n)
            oa, om, oq = oracle_ops.resample(kind, key, lw, n)
            same(ha, oa, f"{name}: {kind} ancestors"); same(hq, oq, f"{name}: q")
            assert torch.equal(hm.cpu().isnan(), om.isnan()) and torch.equal(hm.cpu().nan_to_num(), om.nan_to_num()), name
            assert 0 <= int(oa.min()) and int(oa.max()) < n
        for mode in (0, 1):
            same(hip_ops.categorical_index(key, dev(lw, hip_ops), mode), oracle_ops.categorical_index(key, lw, mode), name)
        # the log-normalisers: max-anchored (logsumexp) and row-anchored (row_stats + lse_rows)
        for a, b in zip(hip_ops.logsumexp(dev(lw, hip_ops)), oracle_ops.logsumexp(lw)):
            same(a, b, f"{name}: logsumexp")
        for a, b in zip(hip_ops.lse_rows(hip_ops.row_stats(dev(lw, hip_ops))), oracle_ops.lse_rows(oracle_ops.row_stats(lw))):
            same(a, b, f"{name}: lse_rows")


def same_or_both_nan(a, b, what=""):
    """Bit-equal where a number comes out; where the arithmetic gives NaN, a NaN on both sides (its sign / payload is
    the platform's)."""
    a, b = a.cpu(), b.cpu()
    if a.dtype.is_floating_point:
        assert torch.equal(a.isnan(), b.isnan()), f"{what}: NaN in different places ({int(a.isnan().sum())} vs {int(b.isnan().sum())})"
        same(a.nan_to_num(nan=0.0), b.nan_to_num(nan=0.0), what)
    else:
        same(a, b, what)


BAD_ARGS = [float("nan"), float("inf"), float("-inf"), 0.0, -1.0, 1e-45, 3e38]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("dist", ["normal", "gamma", "beta", "bernoulli"])
def test_sites_with_invalid_parameters(hip_ops, oracle_ops, impl, dist):
    """Parameters outside a distribution's domain (zero / negative / infinite / NaN scale, rate, concentration,
    probability): the reference's TFP arithmetic returns NaN or inf there and raises nothing; here too the result is
    whatever the spec's f32 operations give — the same on both sides, samplers bounded (64 rejection rounds)."""
    n = 3000
    key = KeyBatch(impl, 2, parent=(8, 1))
    for a in BAD_ARGS + [1.5]:
        for b in ([None] if dist == "bernoulli" else BAD_ARGS + [0.7]):
            hv, hs = hip_ops.sample_logpdf(dist, key, n, a, b)
            ov, os_ = oracle_ops.sample_logpdf(dist, key, n, a, b)
            same_or_both_nan(hv, ov, f"{dist}({a}, {b}) value")
            same_or_both_nan(hs, os_, f"{dist}({a}, {b}) score")
            if dist != "bernoulli":
                for v in (0.3, -2.0, float("nan"), float("inf"), 0.0, 1.0):
                    same_or_both_nan(hip_ops.logpdf(dist, n, v, a, b), oracle_ops.logpdf(dist, n, v, a, b), f"{dist}.logpdf({v}; {a}, {b})")
            else:
                for v in (0, 1):
                    same_or_both_nan(hip_ops.logpdf(dist, n, v, a), oracle_ops.logpdf(dist, n, v, a), f"bernoulli.logpdf({v}; {a})")


@pytest.mark.parametrize("impl", IMPLS)
def test_importance_plan_with_invalid_parameters(hip_ops, oracle_ops, impl, plan_mode):
    """The specialised importance kernels on a model whose parameters / observations are invalid here and there: a NaN
    observation, a zero and a negative scale, an infinite mean, a gamma with a negative concentration feeding a scale.
    Values, scores, weights and the row-anchored log-normaliser records equal the oracle's (NaN where it has NaN)."""
    n = 20000
    kb = W.importance_particle_keys(prng.key(5, impl), n)
    for variant in range(6):
        sites = W.gaussian10_sites(W.gaussian10_data())[:8]
        if variant == 0:
            sites[1].obs = abi.Arg(abi.ARG_CONST, 0, 0.0, float("nan"), None)
        elif variant == 1:
            sites[3].arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, 0.0, None)       # observation scale 0
        elif variant == 2:
            sites[2].arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, -1.0, None)      # a latent's scale negative
        elif variant == 3:
            sites[4].arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, float("inf"), None)
        elif variant == 4:
            g = abi.Site()
            g.dist, g.observed, g.out_col = abi.DIST_GAMMA, 0, 4
            g.arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, -2.0, None)
            g.arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, 1.0, None)
            sites.append(g)
            o = abi.Site()
            o.dist, o.observed, o.out_col = abi.DIST_NORMAL, 1, -1
            o.arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, 0.0, None)
            o.arg[1] = abi.Arg(abi.ARG_SITE, 8, 1.0, 0.0, None)
            o.obs = abi.Arg(abi.ARG_CONST, 0, 0.0, 0.3, None)
            sites.append(o)
        else:
            sites[5].obs = abi.Arg(abi.ARG_CONST, 0, 0.0, 3e38, None)          # log-weights near -inf
        nlat = sum(1 for st in sites if not st.observed)
        out = []
        for ops in (hip_ops, oracle_ops):
            plan = ops.plan_create(sites)
            vals, score, logw, mp = ops.importance_run(plan, kb, n, [], [torch.float32] * nlat)
            rows = ops.row_stats(logw)
            out.append((vals, score, logw, mp, rows.e, rows.s, ops.lse_rows(rows), ops.logsumexp(logw)))
        h, o = out
        for a, b in zip(h[0], o[0]):
            same_or_both_nan(a, b, f"variant {variant}: values")
        for k, what in ((1, "score"), (2, "logw"), (3, "max partials"), (4, "row anchors"), (5, "row sums")):
            same_or_both_nan(h[k], o[k], f"variant {variant}: {what}")
        for a, b in zip(h[6] + h[7], o[6] + o[7]):
            same_or_both_nan(a, b, f"variant {variant}: log-normaliser")


@pytest.mark.parametrize("impl", IMPLS)
def test_expression_arguments(hip_ops, oracle_ops, impl, monkeypatch):
    """GJX_ARG_EXPR at the ABI: postfix programs over sites, an input column, parameters and literals as distribution
    arguments (`w * x + b`, a product of two sites, a negation); every kernel form (n aligned for four particles per lane,
    n odd) equals the oracle; the table interpreter refuses such a plan (GJX_ERR_UNSUPPORTED) instead of guessing."""
    from genjax._amd.abi import GjxError

    keep = []
    E = lambda prog: abi.expr_arg(prog, keep)  # noqa: E731
    c = lambda v: abi.Arg(abi.ARG_CONST, 0, 0.0, v, None)  # noqa: E731

    def site(dist, a0, a1, out_col=-1, obs=None):
        st = abi.Site()
        st.dist, st.observed, st.out_col = dist, 0 if obs is None else 1, out_col
        st.arg[0], st.arg[1] = a0, a1
        if obs is not None:
            st.obs = obs
        return st

    S, K, P_, I = abi.EXPR_SITE, abi.EXPR_CONST, abi.EXPR_PARAM, abi.EXPR_INPUT
    ADD, SUB, MUL, NEG = abi.EXPR_ADD, abi.EXPR_SUB, abi.EXPR_MUL, abi.EXPR_NEG
    sites = [
        site(abi.DIST_NORMAL, c(0.0), c(1.0), 0),                                                    # w
        site(abi.DIST_NORMAL, c(0.5), c(2.0), 1),                                                    # b
        site(abi.DIST_BERNOULLI, c(0.3), c(0.0), 2),                                                 # k
        site(abi.DIST_NORMAL, E([(S, 0, 0), (P_, 0, 0), (MUL, 0, 0), (S, 1, 0), (ADD, 0, 0)]), c(0.5), obs=abi.Arg(abi.ARG_PARAM, 1, 1.0, 0.0, None)),
        site(abi.DIST_NORMAL, E([(S, 0, 0), (S, 1, 0), (SUB, 0, 0), (S, 0, 0), (S, 1, 0), (SUB, 0, 0), (MUL, 0, 0), (K, 0, 1.0), (SUB, 0, 0)]), c(0.7), 3),
        site(abi.DIST_GAMMA, E([(S, 0, 0), (S, 0, 0), (MUL, 0, 0), (K, 0, 0.5), (ADD, 0, 0)]), E([(I, 0, 0), (I, 0, 0), (MUL, 0, 0), (K, 0, 1.0), (ADD, 0, 0)]), 4),
        site(abi.DIST_NORMAL, E([(S, 2, 0), (K, 0, 2.0), (MUL, 0, 0), (S, 5, 0), (SUB, 0, 0), (NEG, 0, 0)]), c(1.0), obs=c(0.4)),
    ]
    dtypes = [torch.float32, torch.float32, torch.int32, torch.float32, torch.float32]
    for n in (20000, 20001, 3):
        g = torch.Generator().manual_seed(n)
        col = torch.randn(n, generator=g)
        kb = W.importance_particle_keys(prng.key(31, impl), n)
        outs = []
        for ops in (hip_ops, oracle_ops):
            plan = ops.plan_create(sites)
            plan.set_params([1.25, -0.3])
            vals, score, logw, mp, rows = ops.importance_run(plan, kb, n, [dev(col, ops)], dtypes, want_rows=True)
            outs.append(vals + [score, logw, mp, rows.e, rows.s])
        for i, (a, b) in enumerate(zip(*outs)):
            same(a, b, f"n {n}: output {i}")
    monkeypatch.setenv("GJX_PLAN_JIT", "0")
    plan = hip_ops.plan_create(sites)
    plan.set_params([1.25, -0.3])
    with pytest.raises(GjxError) as ei:
        hip_ops.importance_run(plan, W.importance_particle_keys(prng.key(31, impl), 64), 64, [dev(torch.zeros(64), hip_ops)], dtypes)
    assert ei.value.code == -2  # GJX_ERR_UNSUPPORTED


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("mode", [0, 1])
def test_categorical_with_invalid_logits(hip_ops, oracle_ops, impl, mode):
    n, K = 2000, 7
    key = KeyBatch(impl, 2, parent=(8, 2))
    g = torch.Generator().manual_seed(3)
    base = torch.randn(1, K, generator=g)
    rows = {"all -inf": torch.full((1, K), float("-inf")), "all nan": torch.full((1, K), float("nan")),
            "one nan": base.clone().index_fill_(1, torch.tensor([2]), float("nan")),
            "one +inf": base.clone().index_fill_(1, torch.tensor([4]), float("inf")),
            "huge": base * 1e38}
    for name, logits in rows.items():
        hv, hs = hip_ops.sample_logpdf_categorical(key, n, dev(logits, hip_ops), mode=mode)
        ov, os_ = oracle_ops.sample_logpdf_categorical(key, n, logits, mode=mode)
        same(hv, ov, f"{name}: value")
        same_or_both_nan(hs, os_, f"{name}: score")
        assert 0 <= int(ov.min()) and int(ov.max()) < K
        for v in (0, 4, -1, K):
            same_or_both_nan(hip_ops.logpdf_categorical(n, v, dev(logits, hip_ops)), oracle_ops.logpdf_categorical(n, v, logits), f"{name}: logpdf({v})")


@pytest.mark.parametrize("impl", IMPLS)
def test_hmm_with_invalid_tables(hip_ops, oracle_ops, impl):
    """HMM tables with forbidden transitions (-inf), an unreachable-and-unleavable state (a row of -inf), NaN entries and
    an observation no state can emit: the prepared alias / log-probability tables and the whole filter equal the oracle's."""
    K, n, T = 12, 30000, 6
    g = torch.Generator().manual_seed(4)
    base_t, base_o = torch.randn(K, K, generator=g), torch.randn(K, K, generator=g)
    variants = []
    t = base_t.clone(); t[:, ::3] = float("-inf"); variants.append(("forbidden transitions", t, base_o))
    t = base_t.clone(); t[5, :] = float("-inf"); variants.append(("a row of -inf", t, base_o))
    t = base_t.clone(); t[2, 7] = float("nan"); t[4, :] = float("nan"); variants.append(("NaN transitions", t, base_o))
    o = base_o.clone(); o[:, 3] = float("-inf"); variants.append(("an observation no state emits", base_t, o))
    o = base_o.clone(); o[1, :] = float("nan"); o[:, 8] = float("inf"); variants.append(("NaN / +inf emissions", base_t, o))
    y = np.array([3, 1, 8, 3, 0, 11], dtype=np.int32)
    sk, rk = W.smc_key_schedule(prng.key(13, impl), T)
    for name, tl, ol in variants:
        for init in (0, 5):
            ha, hl = hip_ops.hmm_prepare(K, init, dev(tl, hip_ops), dev(ol, hip_ops))
            oa, ol_ = oracle_ops.hmm_prepare(K, init, tl, ol)
            same(ha, oa, f"{name}: alias table"); same_or_both_nan(hl, ol_, f"{name}: emission log-probabilities")
            h = hip_ops.smc_run_hmm(impl, n, sk, rk, K, init, dev(tl, hip_ops), dev(ol, hip_ops), y, True)
            o_ = oracle_ops.smc_run_hmm(impl, n, sk, rk, K, init, tl, ol, y, True)
            for a, b, what in zip(h, o_, ("step max", "step q", "state", "logw", "ancestors")):
                same_or_both_nan(a, b, f"{name} (init {init}): {what}")
            assert 0 <= int(o_[2].min()) and int(o_[2].max()) < K and 0 <= int(o_[4].min()) and int(o_[4].max()) < n


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("bad", [float("nan"), float("inf"), 1e30])
def test_smc_with_an_impossible_observation(hip_ops, oracle_ops, impl, bad):
    """One observation is NaN / inf / absurd: that step's weights are all NaN or -inf.  The filter goes on (zero total:
    every slot takes the last particle; all -inf: uniform) and HIP equals the oracle at every step."""
    n, T = 70000, 5
    y = np.array([0.1, bad, 0.3, 0.2, -0.4], dtype=np.float32)
    sk, rk = W.smc_key_schedule(prng.key(11, impl), T)
    mdl = abi.Lgssm(0.0, 1.0, 0.9, 1.0, 0.5)
    h = hip_ops.smc_run_lgssm(impl, n, sk, rk, mdl, y, True)
    o = oracle_ops.smc_run_lgssm(impl, n, sk, rk, mdl, y, True)
    for a, b, what in zip(h, o, ("step max", "step q", "state", "logw", "ancestors")):
        a, b = a.cpu(), b.cpu()
        if a.is_floating_point():
            assert torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(), b.nan_to_num()), what
        else:
            same(a, b, what)
    assert 0 <= int(o[4].min()) and int(o[4].max()) < n


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n", [1, 100, 5000, 100000])
def test_categorical_index(hip_ops, oracle_ops, impl, mode, n):
    g = torch.Generator().manual_seed(n + mode)
    lw = torch.randn(n, generator=g) * 2
    for s in range(3):
        key = KeyBatch(impl, 2, parent=(s, 99))
        same(hip_ops.categorical_index(key, dev(lw, hip_ops), mode), oracle_ops.categorical_index(key, lw, mode),
             "categorical_index")


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n,T", [(1024, 5), (5000, 20), (100000, 30), (2_200_000, 3)])  # last: > 2048 tiles
def test_smc_lgssm(hip_ops, oracle_ops, impl, n, T):
    h = W.lgssm_smc(hip_ops, impl, seed=7, n=n, T=T, want_ancestors=True)
    o = W.lgssm_smc(oracle_ops, impl, seed=7, n=n, T=T, want_ancestors=True)
    same(h["ancestors"], o["ancestors"], "ancestors")
    same(h["out_e"], o["out_e"], "per-step max"); same(h["out_q"], o["out_q"], "per-step q")
    same(h["state"], o["state"], "final particles"); same(h["logw"], o["logw"], "final log-weights")
    assert h["log_z"] == o["log_z"]
    # without ancestor output the run must not change
    h2 = W.lgssm_smc(hip_ops, impl, seed=7, n=n, T=T, want_ancestors=False)
    same(h2["out_q"], o["out_q"]); same(h2["state"], o["state"])


def degenerate_lgssm_run(ops, impl, n, T=8, **kw):
    """A filter whose weights collapse: a sharp observation model and observations that jump by tens of
    standard deviations, so that at some steps a handful of particles (in a few tiles) carry all the mass
    and most tiles have mass 0 — the resampler's empty-tile and many-copies-of-one-source paths."""
    from genjax._amd import abi, prng

    y = np.array([0.1, 25.0, -40.0, -39.5, 60.0, 60.2, 0.0, 3.0][:T], dtype=np.float32)
    sk, rk = W.smc_key_schedule(prng.key(11, impl), T)
    return ops.smc_run_lgssm(impl, n, sk, rk, abi.Lgssm(0.0, 1.0, 0.9, 1.0, 0.05), y, True, **kw)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", [4096, 50000])
def test_smc_degenerate_weights(hip_ops, oracle_ops, impl, n):
    h, o = degenerate_lgssm_run(hip_ops, impl, n), degenerate_lgssm_run(oracle_ops, impl, n)
    for a, b, what in zip(h, o, ("step max", "step q", "state", "logw", "ancestors")):
        same(a, b, what)
    anc = o[4]
    assert int(anc[2].unique().numel()) < n // 100  # the collapse really happens: few distinct ancestors


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n", [20000, 300000, 2_200_000])  # 2.2e6: through the precomputed tile prefix (> 1024 tiles)
def test_smc_collapse_helpers(hip_ops, oracle_ops, impl, n):
    """Weight collapse at sizes where ONE source tile owns thousands of output slots: every output tile inside its run
    reads that one tile's stored CDF (the output-centric step is collapse-proof by construction) — ancestors, particles
    and weights are the oracle's bits."""
    T = 6 if n > 1_000_000 else 8
    h, o = degenerate_lgssm_run(hip_ops, impl, n, T), degenerate_lgssm_run(oracle_ops, impl, n, T)
    for a, b, what in zip(h, o, ("step max", "step q", "state", "logw", "ancestors")):
        same(a, b, what)
    anc = o[4]
    counts = torch.bincount(anc[2].long() // 1024, minlength=(n + 1023) // 1024)
    assert int(counts.max()) > 4096  # some tile really is heavy
    # a filter batch takes the same path per filter
    if n == 20000:
        from genjax._amd import abi, prng
        y = np.array([0.1, 25.0, -40.0, -39.5, 60.0, 60.2], dtype=np.float32)
        pairs = [W.smc_key_schedule(prng.key(11 + f, impl), 6) for f in range(5)]
        sk, rk = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
        mdl = abi.Lgssm(0.0, 1.0, 0.9, 1.0, 0.05)
        hb = hip_ops.smc_run_lgssm(impl, n, sk, rk, mdl, y, True)
        ob = oracle_ops.smc_run_lgssm(impl, n, sk, rk, mdl, y, True)
        for a, b, what in zip(hb, ob, ("step max", "step q", "state", "logw", "ancestors")):
            if what in ("state", "logw"):  # [F, stride]: the padding behind each filter's n particles is never written
                a, b = a[:, :n], b[:, :n]
            elif what == "ancestors":
                a, b = a[:, :, :n], b[:, :, :n]
            same(a, b, what + " (batch of 5)")


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n,n_out", [(300_000, 300_000), (70_004, 70_004), (50_000, 200_000)])
def test_resample_heavy_tile_without_idle_tiles(hip_ops, oracle_ops, impl, n, n_out):
    """One particle holds most of the mass and EVERY other particle has some (the pattern that left the source-centric
    kernels of rounds 1-2 without idle workgroups to delegate to).  Ancestors are the oracle's; also with two heavy
    tiles and with more output slots than particles."""
    kb = KeyBatch(impl, 2, parent=(5, 6))
    for heavy in ([n // 3], [7, n - 5]):
        lw = torch.zeros(n)
        for i in heavy:
            lw[i] = math.log(1.5 * n / len(heavy))
        a, m, q = hip_ops.resample("systematic", kb, lw.to(hip_ops.device()), n_out)
        b, mo, qo = oracle_ops.resample("systematic", kb, lw, n_out)
        same(a, b, "ancestors")
        same(q, qo, "total mass")
        cnt = torch.bincount(b.long(), minlength=n)
        assert int(cnt.max()) > 0.25 * n_out and int((cnt > 0).sum()) > 0.2 * min(n, n_out)


def test_smc_per_slot_search_path(hip_ops):
    """The other way an output tile finds its ancestors — every slot searching the merged tile prefix and one stored
    in-tile CDF, taken when a tile has more than 16 source tiles — forced for EVERY tile (GJX_SMC_SCAN_MAX=0, read when
    the library loads: a child process): collapsing and ordinary filters, the generic resampler with n_out != n, and a
    population beyond 1024 tiles (precomputed prefix) equal the oracle bit for bit."""
    import subprocess
    import sys

    child = r"""
import sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import torch
from genjax._amd.abi import GjxLib
from genjax._amd.ops import KeyBatch, Ops
from genjax._amd.runtime import load_hip_ops
from genjax._amd import workloads as W
from test_gpu_parity_abi import degenerate_lgssm_run
hip, ora = load_hip_ops(), Ops(GjxLib(sys.argv[3], "cpu"))
for impl in (0, 1):
    for n, T in ((300_000, 8), (20_000, 8), (1_200_000, 4)):
        h, o = degenerate_lgssm_run(hip, impl, n, T), degenerate_lgssm_run(ora, impl, n, T)
        for a, b in zip(h, o):
            assert torch.equal(a.cpu(), b.cpu())
    h = W.lgssm_smc(hip, impl, seed=7, n=70_001, T=12, want_ancestors=True)
    o = W.lgssm_smc(ora, impl, seed=7, n=70_001, T=12, want_ancestors=True)
    for k in ("ancestors", "out_e", "out_q", "state", "logw"):
        assert torch.equal(h[k].cpu(), o[k].cpu()), k
    g = torch.Generator().manual_seed(3)
    lw = torch.randn(50_000, generator=g) * 4
    for n_out in (50_000, 7, 130_001):
        a, e, q = hip.resample("systematic", KeyBatch(impl, 2, parent=(5, 6)), lw.cuda(), n_out)
        b, eo, qo = ora.resample("systematic", KeyBatch(impl, 2, parent=(5, 6)), lw, n_out)
        assert torch.equal(a.cpu(), b) and int(e) == int(eo) and int(q) == int(qo)
print("ok")
"""
    from conftest import ORACLE_LIB, ROOT

    env = dict(os.environ, GJX_SMC_SCAN_MAX="0")
    r = subprocess.run([sys.executable, "-c", child, os.path.join(ROOT, "genjax-chi_amd"), os.path.join(ROOT, "tests"), ORACLE_LIB],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n,n_out", [(1_000_000, 1_000_000), (300_000, 300_000), (70_004, 20_000)])
def test_resample_many_light_source_tiles(hip_ops, oracle_ops, impl, n, n_out):
    """One particle holds 99.9 % of the mass and the rest is spread evenly: the thousand slots of the light particles
    fall into one or two OUTPUT tiles whose sources are hundreds of tiles with a tooth each — those tiles take the
    per-slot search, every other one the marks; a step's cost is bounded either way.  Ancestors are the oracle's."""
    kb = KeyBatch(impl, 2, parent=(5, 6))
    for frac_light in (1e-3, 5e-2):
        lw = torch.zeros(n)
        lw[n // 2 + 3] = math.log((1.0 - frac_light) / frac_light * n)
        a, e, q = hip_ops.resample("systematic", kb, lw.to(hip_ops.device()), n_out)
        b, eo, qo = oracle_ops.resample("systematic", kb, lw, n_out)
        same(a, b, "ancestors")
        same(q, qo, "total mass"); same(e, eo, "anchor")
        cnt = torch.bincount(b.long(), minlength=n)
        assert int(cnt.max()) > 0.9 * n_out and int((cnt > 0).sum()) > 0.5 * frac_light * n_out


ESS_CASES = [("lgssm", 5000, 30, 0.5, 1), ("lgssm", 1024, 12, 0.9, 1), ("lgssm", 70000, 25, 0.3, 1), ("lgssm", 3000, 20, 0.5, 5),
             ("hmm", 6000, 30, 0.5, 1), ("hmm", 2048, 16, 0.25, 3), ("lgssm", 2_200_000, 5, 0.5, 1), ("lgssm", 4096, 10, 1e-6, 1)]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("kind,n,T,thr,F", ESS_CASES)
def test_smc_ess_adaptive(hip_ops, oracle_ops, impl, kind, n, T, thr, F):
    """gjx_smc_config.ess_threshold: resample only when ESS < thr * N.  The decision is a function of exact integer
    sums, so HIP and oracle take the same one at every step: flags, ancestors (identity on kept steps), accumulated
    log-weights, per-step (e, q) and log Z are equal bit for bit — single filters, filter batches, > 1024 tiles."""
    mk = (lambda ops: W.LgssmSMC(ops, impl, 5, n, T, want_ancestors=True, filters=F, ess_threshold=thr)) if kind == "lgssm" else (
        lambda ops: W.HmmSMC(ops, impl, 5, n, T, n_states=16, want_ancestors=True, filters=F, ess_threshold=thr))
    hw, ow = mk(hip_ops), mk(oracle_ops)
    h, o = hw.result(hw.run()), ow.result(ow.run())
    for key in ("resampled", "out_e", "out_q", "state", "logw", "ancestors"):
        same(h[key], o[key], key)
    assert h["log_z"] == o["log_z"]
    fl = o["resampled"] if F == 1 else o["resampled"][0]
    anc = o["ancestors"] if F == 1 else o["ancestors"][:, 0]
    assert int(fl[0]) == 0
    kept = [t for t in range(1, T) if int(fl[t]) == 0]
    if thr < 1e-3:
        assert len(kept) == T - 1  # never resamples
    elif thr < 0.9:
        assert 0 < len(kept) < T - 1  # a genuinely adaptive schedule
    for t in kept[:3]:
        assert torch.equal(anc[t], torch.arange(n, dtype=torch.int32))
    if kind == "lgssm" and F == 1 and 0.2 < thr < 0.9 and n >= 5000:
        assert abs(o["log_z"] - o["log_z_exact"]) < 0.8  # still an estimate of the evidence


@pytest.mark.parametrize("n", [1, 1023, 1024, 1025, 5000, 1_500_000])
def test_tile_records(hip_ops, oracle_ops, n):
    """gjx_tile_weights / gjx_tile_merge: fixed-point weights, tile records and their merge of arbitrary
    log-weights — wide dynamic range, -inf / NaN / +inf entries, a ragged last tile — are the oracle's bits, and the
    merged (e, Q) pair reproduces logsumexp."""
    g = torch.Generator().manual_seed(n)
    lw = torch.randn(n, generator=g) * 30
    if n > 10:
        lw[3] = float("-inf"); lw[7] = float("nan")
    hc, hr, hs, he_ = hip_ops.tile_weights(lw.to(hip_ops.device()))
    oc, orr, os_, oe_ = oracle_ops.tile_weights(lw)
    same(hc, oc, "fixed-point weights"); same(hr, orr, "records"); same(hs, os_, "sub-prefixes"); same(he_, oe_, "ESS sums")
    he, hq = hip_ops.tile_merge(hr)
    oe, oq = oracle_ops.tile_merge(orr)
    same(he, oe, "merged anchor"); same(hq, oq, "merged mass")
    ref = float(torch.logsumexp(torch.nan_to_num(lw.double(), nan=float("-inf")), 0))
    assert (int(oe) - 30) * math.log(2) + math.log(int(oq)) == pytest.approx(ref, abs=1e-5)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n,T,k", [(2048, 6, 16), (30000, 25, 256)])
def test_smc_hmm(hip_ops, oracle_ops, impl, n, T, k):
    h = W.hmm_smc(hip_ops, impl, seed=9, n=n, T=T, n_states=k, want_ancestors=True)
    o = W.hmm_smc(oracle_ops, impl, seed=9, n=n, T=T, n_states=k, want_ancestors=True)
    same(h["ancestors"], o["ancestors"], "ancestors")
    same(h["out_e"], o["out_e"]); same(h["out_q"], o["out_q"])
    same(h["state"], o["state"]); same(h["logw"], o["logw"])


def _smc_plans(ops):
    """(LGSSM as a plan, a 2-state plan with gamma / bernoulli / table arguments)."""
    A, m = abi.Arg, W.LGSSM

    def site(dist, a0, a1=None, obs=None):
        s_ = abi.Site()
        s_.dist, s_.observed, s_.out_col = dist, 0 if obs is None else 1, -1
        s_.arg[0] = a0
        if a1 is not None:
            s_.arg[1] = a1
        if obs is not None:
            s_.obs = obs
        return s_

    c = lambda v: A(abi.ARG_CONST, 0, 0.0, v, None)
    lg = ops.smc_plan_create(
        [site(abi.DIST_NORMAL, c(m["x0_loc"]), c(m["x0_scale"])),
         site(abi.DIST_NORMAL, A(abi.ARG_SITE, 0, 1.0, 0.0, None), c(m["r"]), A(abi.ARG_OBS, 0, 1.0, 0.0, None))],
        [site(abi.DIST_NORMAL, A(abi.ARG_STATE, 0, m["a"], 0.0, None), c(m["q"])),
         site(abi.DIST_NORMAL, A(abi.ARG_SITE, 0, 1.0, 0.0, None), c(m["r"]), A(abi.ARG_OBS, 0, 1.0, 0.0, None))],
        [A(abi.ARG_SITE, 0, 1.0, 0.0, None)], [A(abi.ARG_SITE, 0, 1.0, 0.0, None)], 1)
    rich = ops.smc_plan_create(
        [site(abi.DIST_NORMAL, c(0.0), c(1.0)), site(abi.DIST_GAMMA, c(2.0), c(2.0)),
         site(abi.DIST_NORMAL, A(abi.ARG_SITE, 0, 1.0, 0.0, None), c(0.7), A(abi.ARG_OBS, 0, 1.0, 0.0, None))],
        [site(abi.DIST_NORMAL, A(abi.ARG_STATE, 0, 0.8, 0.0, None), c(0.5)),
         site(abi.DIST_GAMMA, c(0.6), A(abi.ARG_STATE, 1, 1.0, 1.0, None)),
         site(abi.DIST_BERNOULLI, c(0.3)),
         site(abi.DIST_BETA, c(2.0), A(abi.ARG_SITE, 2, 1.5, 0.5, None)),
         site(abi.DIST_NORMAL, A(abi.ARG_SITE, 0, 1.0, 0.0, None), A(abi.ARG_SITE, 3, 1.0, 0.2, None),
              A(abi.ARG_OBS, 0, 1.0, 0.0, None)),
         site(abi.DIST_BERNOULLI, A(abi.ARG_SITE, 3, 1.0, 0.0, None), None, A(abi.ARG_OBS, 1, 1.0, 0.0, None))],
        [A(abi.ARG_SITE, 0, 1.0, 0.0, None), A(abi.ARG_SITE, 1, 1.0, 0.0, None)],
        [A(abi.ARG_SITE, 0, 1.0, 0.0, None), A(abi.ARG_SITE, 1, 0.5, 0.1, None)], 2)
    return lg, rich


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("kind,n,T", [("lgssm", 5000, 12), ("lgssm", 1024, 4), ("hmm", 3000, 9)])
def test_smc_filters_in_one_launch(hip_ops, oracle_ops, impl, kind, n, T):
    """gjx_smc_config.n_filters: F independent filters (seeds s, s+1, ...) stepping in the same launches equal
    F separate runs bit for bit — per-step (max, q), final particles and weights, ancestors — on both backends."""
    F = 3 if n > 2000 else 16  # (16 = the most filters one launch takes)
    mk = (lambda ops, f, seed: W.LgssmSMC(ops, impl, seed, n, T, want_ancestors=True, filters=f)) if kind == "lgssm" else (
        lambda ops, f, seed: W.HmmSMC(ops, impl, seed, n, T, n_states=16, want_ancestors=True, filters=f))
    hb = mk(hip_ops, F, 7)
    got = hb.result(hb.run())
    ob = mk(oracle_ops, F, 7)
    want = ob.result(ob.run())
    for key in ("out_e", "out_q", "state", "logw", "ancestors"):
        same(got[key], want[key], f"{key} (batched, HIP vs oracle)")
    for f in range(F):
        one = mk(hip_ops, 1, 7 + f)
        ref = one.result(one.run())
        same(got["out_e"][f], ref["out_e"], "out_e"); same(got["out_q"][f], ref["out_q"], "out_q")
        same(got["state"][f], ref["state"], "state"); same(got["logw"][f], ref["logw"], "logw")
        same(got["ancestors"][:, f], ref["ancestors"], "ancestors")
        assert got["log_z"][f] == ref["log_z"]


@pytest.mark.parametrize("impl", IMPLS)
def test_smc_plans(hip_ops, oracle_ops, impl):
    """Plan-driven bootstrap SMC (hiprtc-generated policy in the fused resample kernel) against the
    oracle's plan interpreter, and against the hand-written LGSSM kernel."""
    from genjax._amd import prng

    T, n = 14, 30000
    y = W.lgssm_data(T)
    obs2 = np.stack([y, (np.arange(T) % 2).astype(np.float32)], axis=1)
    sk, rk = W.smc_key_schedule(prng.key(13, impl), T)
    hl, hr = _smc_plans(hip_ops)
    ol, orr = _smc_plans(oracle_ops)
    for hp, op_, obs in ((hl, ol, y), (hr, orr, obs2)):
        h = hip_ops.smc_run_plan(hp, impl, n, sk, rk, obs, True)
        o = oracle_ops.smc_run_plan(op_, impl, n, sk, rk, obs, True)
        same(h[0], o[0], "step max"); same(h[1], o[1], "step q"); same(h[3], o[3], "logw"); same(h[4], o[4], "ancestors")
        for a, b in zip(h[2], o[2]):
            same(a, b, "state column")
    # three filters of the two-state model in the same launches == their own runs == the oracle's
    pairs = [W.smc_key_schedule(prng.key(20 + f, impl), T) for f in range(3)]
    skf, rkf = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    hb = hip_ops.smc_run_plan(hr, impl, n, skf, rkf, obs2, True)
    ob = oracle_ops.smc_run_plan(orr, impl, n, skf, rkf, obs2, True)
    same(hb[0], ob[0], "batched step max"); same(hb[1], ob[1], "batched step q"); same(hb[4][:, :, :n], ob[4][:, :, :n], "batched ancestors")  # (slots n .. stride of a filter are padding)
    for f in range(3):
        one = hip_ops.smc_run_plan(hr, impl, n, pairs[f][0], pairs[f][1], obs2, True)
        same(hb[1][f], one[1], "q of filter"); same(hb[3][f, :n], one[3], "logw of filter")
        same(hb[4][:, f, :n], one[4], "ancestors of filter")
        for a, b in zip(hb[2], one[2]):
            same(a[f, :n], b, "state column of filter")
    fixed = hip_ops.smc_run_lgssm(impl, n, sk, rk, W.lgssm_model(), y, True)
    gen_ = hip_ops.smc_run_plan(hl, impl, n, sk, rk, y, True)
    same(gen_[1], fixed[1], "generated vs hand-written LGSSM q"); same(gen_[2][0], fixed[2], "particles")
    same(gen_[4], fixed[4], "ancestors")


def _smc_plan_scoped(ops):
    """A filter whose step is composed of sub-models: transition(x) = {eps ~ normal, g ~ gamma} (scope 1), the body's own
    x ~ normal(0.8 x_prev + 0.3 eps, 0.4 + 0.1 g) and k ~ flip, emission = {y ~ normal(x, 0.6) observed} (scope 2); init has a
    nested prior {x0 ~ normal} (scope 1) and an observed y."""
    A = abi.Arg
    c = lambda v: A(abi.ARG_CONST, 0, 0.0, v, None)

    def site(dist, a0, a1=None, obs=None):
        s_ = abi.Site()
        s_.dist, s_.observed, s_.out_col = dist, 0 if obs is None else 1, -1
        s_.arg[0] = a0
        if a1 is not None:
            s_.arg[1] = a1
        if obs is not None:
            s_.obs = obs
        return s_

    keep = []
    mean = abi.expr_arg([(abi.EXPR_STATE, 0, 0.0), (abi.EXPR_CONST, 0, 0.8), (abi.EXPR_MUL, 0, 0.0), (abi.EXPR_SITE, 0, 0.0),
                         (abi.EXPR_CONST, 0, 0.3), (abi.EXPR_MUL, 0, 0.0), (abi.EXPR_ADD, 0, 0.0)], keep)
    init = [site(abi.DIST_NORMAL, c(0.0), c(1.0)),
            site(abi.DIST_NORMAL, A(abi.ARG_SITE, 0, 1.0, 0.0, None), c(0.6), A(abi.ARG_OBS, 0, 1.0, 0.0, None))]
    step = [site(abi.DIST_NORMAL, c(0.0), c(1.0)), site(abi.DIST_GAMMA, c(2.0), c(2.0)),
            site(abi.DIST_NORMAL, mean, A(abi.ARG_SITE, 1, 0.1, 0.4, None)), site(abi.DIST_BERNOULLI, c(0.3)),
            site(abi.DIST_NORMAL, A(abi.ARG_SITE, 2, 1.0, 0.0, None), c(0.6), A(abi.ARG_OBS, 0, 1.0, 0.0, None))]
    plan = ops.smc_plan_create(init, step, [A(abi.ARG_SITE, 0, 1.0, 0.0, None)], [A(abi.ARG_SITE, 2, 1.0, 0.0, None)], 1,
                               init_scopes=[(0, 0, 1)], step_scopes=[(0, 0, 2), (0, 4, 5)])
    plan._keep_progs = keep
    return plan


@pytest.mark.parametrize("impl", IMPLS)
def test_smc_plans_with_nested_calls(hip_ops, oracle_ops, impl):
    """gjx_smc_plan_create_scoped: a generated filter whose init / step bodies call sub-models — per-scope keys under the
    slot key, the quad blocks for the body's own sites only — equals the oracle's walk bit for bit (every-step and
    ESS-adaptive, one filter and three per launch); without the compiler such a filter is refused."""
    import os

    from genjax._amd import prng
    from genjax._amd.abi import GjxError

    T, n = 11, 9000
    y = W.lgssm_data(T)
    sk, rk = W.smc_key_schedule(prng.key(17, impl), T)
    hp, op_ = _smc_plan_scoped(hip_ops), _smc_plan_scoped(oracle_ops)
    for thr in (0.0, 0.5):
        h = hip_ops.smc_run_plan(hp, impl, n, sk, rk, y, True, ess_threshold=thr)
        o = oracle_ops.smc_run_plan(op_, impl, n, sk, rk, y, True, ess_threshold=thr)
        same(h[0], o[0], "step max"); same(h[1], o[1], "step q"); same(h[3], o[3], "logw"); same(h[4], o[4], "ancestors")
        for a, b in zip(h[2], o[2]):
            same(a, b, "state column")
    pairs = [W.smc_key_schedule(prng.key(30 + f, impl), T) for f in range(3)]
    skf, rkf = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    hb = hip_ops.smc_run_plan(hp, impl, n, skf, rkf, y, True)
    ob = oracle_ops.smc_run_plan(op_, impl, n, skf, rkf, y, True)
    same(hb[1], ob[1], "batched step q"); same(hb[4][:, :, :n], ob[4][:, :, :n], "batched ancestors")
    os.environ["GJX_PLAN_JIT"] = "0"
    try:
        with pytest.raises(GjxError):
            hip_ops.smc_run_plan(_smc_plan_scoped(hip_ops), impl, n, sk, rk, y, False)
    finally:
        del os.environ["GJX_PLAN_JIT"]


@pytest.mark.parametrize("impl", IMPLS)
def test_smc_plans_without_the_compiler(hip_ops, oracle_ops, impl):
    """GJX_PLAN_JIT=0: generated filters run through the table-walking policy (k_smc_interp_*: the site table read at run time
    inside the same fused resample kernel) — the same bits as the oracle (hence as the compiled policy), every-step and
    ESS-adaptive, one filter and three per launch; a filter whose model holds a program (GJX_ARG_EXPR) has no such route
    and says so."""
    import os

    from genjax._amd import prng
    from genjax._amd.abi import GjxError

    T, n = 9, 6000
    y = W.lgssm_data(T)
    obs2 = np.stack([y, (np.arange(T) % 2).astype(np.float32)], axis=1)
    sk, rk = W.smc_key_schedule(prng.key(13, impl), T)
    ol, orr = _smc_plans(oracle_ops)
    before = hip_ops.jit_stats()["compiles"]
    os.environ["GJX_PLAN_JIT"] = "0"
    try:
        hl, hr = _smc_plans(hip_ops)
        for hp, op_, obs in ((hl, ol, y), (hr, orr, obs2)):
            for thr in (0.0, 0.5):
                h = hip_ops.smc_run_plan(hp, impl, n, sk, rk, obs, True, ess_threshold=thr)
                o = oracle_ops.smc_run_plan(op_, impl, n, sk, rk, obs, True, ess_threshold=thr)
                same(h[0], o[0], "step max"); same(h[1], o[1], "step q"); same(h[3], o[3], "logw"); same(h[4], o[4], "ancestors")
                for a, b in zip(h[2], o[2]):
                    same(a, b, "state column")
        pairs = [W.smc_key_schedule(prng.key(20 + f, impl), T) for f in range(3)]
        skf, rkf = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
        hb = hip_ops.smc_run_plan(hr, impl, n, skf, rkf, obs2, True)
        ob = oracle_ops.smc_run_plan(orr, impl, n, skf, rkf, obs2, True)
        same(hb[1], ob[1], "batched step q"); same(hb[4][:, :, :n], ob[4][:, :, :n], "batched ancestors")
        # a program in the model: compiled, never interpreted
        A = abi.Arg
        keep = []
        prog = abi.expr_arg([(abi.EXPR_STATE, 0, 0.0), (abi.EXPR_STATE, 0, 0.0), (abi.EXPR_MUL, 0, 0.0)], keep)
        s0, s1 = abi.Site(), abi.Site()
        s0.dist, s0.out_col = abi.DIST_NORMAL, -1
        s0.arg[0], s0.arg[1] = A(abi.ARG_CONST, 0, 0.0, 0.0, None), A(abi.ARG_CONST, 0, 0.0, 1.0, None)
        s1.dist, s1.out_col = abi.DIST_NORMAL, -1
        s1.arg[0], s1.arg[1] = prog, A(abi.ARG_CONST, 0, 0.0, 1.0, None)
        pe = hip_ops.smc_plan_create([s0], [s1], [A(abi.ARG_SITE, 0, 1.0, 0.0, None)], [A(abi.ARG_SITE, 0, 1.0, 0.0, None)], 0)
        with pytest.raises(GjxError):
            hip_ops.smc_run_plan(pe, impl, n, sk, rk, np.zeros((T, 1), np.float32), False)
    finally:
        del os.environ["GJX_PLAN_JIT"]
    assert hip_ops.jit_stats()["compiles"] == before  # nothing was compiled on the way


def _scan_plans(ops):
    """(the LGSSM scan kernel, a two-component carry with gamma / beta / bernoulli sites, an input and two observations)."""
    A = abi.Arg
    c = lambda v: A(abi.ARG_CONST, 0, 0.0, v, None)

    def site(dist, out_col, a0, a1=None, obs=None):
        s_ = abi.Site()
        s_.dist, s_.observed, s_.out_col = dist, 0 if obs is None else 1, out_col
        s_.arg[0] = a0
        if a1 is not None:
            s_.arg[1] = a1
        if obs is not None:
            s_.obs = obs
        return s_

    sites, nxt = W.lgssm_scan_sites()
    lg = ops.scan_plan_create(sites, nxt, 1)
    rich = ops.scan_plan_create(
        [site(abi.DIST_NORMAL, 0, A(abi.ARG_STATE, 0, 0.8, 0.0, None), A(abi.ARG_OBS, 2, 1.0, 0.0, None)),
         site(abi.DIST_GAMMA, 1, c(0.6), A(abi.ARG_STATE, 1, 1.0, 1.0, None)),
         site(abi.DIST_BERNOULLI, 2, c(0.3)),
         site(abi.DIST_BETA, 3, c(2.0), A(abi.ARG_SITE, 2, 1.5, 0.5, None)),
         site(abi.DIST_NORMAL, -1, A(abi.ARG_SITE, 0, 1.0, 0.0, None), A(abi.ARG_SITE, 3, 1.0, 0.2, None), A(abi.ARG_OBS, 0, 1.0, 0.0, None)),
         site(abi.DIST_BERNOULLI, -1, A(abi.ARG_SITE, 3, 1.0, 0.0, None), None, A(abi.ARG_OBS, 1, 1.0, 0.0, None))],
        [A(abi.ARG_SITE, 0, 1.0, 0.0, None), A(abi.ARG_SITE, 1, 0.5, 0.1, None)], 3)
    return lg, rich


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n,T", [(70004, 13), (5000, 40), (257, 1)])
def test_scan_run(hip_ops, oracle_ops, impl, n, T):
    """gjx_scan_run (Scan.generate under ImportanceK, scan.py:237-294) — the whole T-step walk of every particle in one
    launch — against the oracle: time-major value columns, weights, scores, final carry, row maxima and row-anchored
    sums, with lazy and materialised particle keys and a per-particle initial carry."""
    from genjax._amd import prng

    y = W.lgssm_data(T)
    obs3 = np.stack([y, (np.arange(T) % 2).astype(np.float32), np.linspace(0.5, 1.5, T).astype(np.float32)], axis=1)
    kb = W.importance_particle_keys(prng.key(3, impl), n)
    x0 = torch.linspace(-1.0, 1.0, n)
    hl, hr = _scan_plans(hip_ops)
    ol, orr = _scan_plans(oracle_ops)
    runs = [(hl, ol, y.reshape(T, 1), [0.25], [torch.float32]),
            (hl, ol, y.reshape(T, 1), [x0], [torch.float32]),
            (hr, orr, obs3, [x0, 0.5], [torch.float32, torch.float32, torch.int32, torch.float32])]
    for hp, op_, obs, c0, dts in runs:
        for keys in (kb, "explicit"):
            got, want = [], []
            for ops, plan, dst in ((hip_ops, hp, got), (oracle_ops, op_, want)):
                k = keys
                if keys == "explicit":
                    k = KeyBatch(impl, 0, tensor=ops.rng_keys(kb, n))
                c0d = [v.to(ops.device()) if isinstance(v, torch.Tensor) else v for v in c0]
                o = ops.scan_run(plan, k, n, T, obs, c0d, dts)
                dst.extend(o["values"] + o["carry"] + [o["score"], o["logw"], o["max_partials"], o["rows"].e, o["rows"].s])
            for i, (a, b) in enumerate(zip(got, want)):
                same(a, b, f"scan output {i}")


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("cat_mode", [0, 1])
def test_scan_run_hmm(hip_ops, oracle_ops, impl, cat_mode):
    """The HMM as a scan kernel (categorical rows chosen by the carried state and by the new state; the reference's
    configs[4] semantics under ImportanceK): one launch == the oracle, Gumbel-max and inverse-CDF categoricals."""
    n, T = 6000, 12
    h, o = W.HmmScan(hip_ops, impl, 4, n, T, n_states=16, cat_mode=cat_mode), W.HmmScan(oracle_ops, impl, 4, n, T, n_states=16, cat_mode=cat_mode)
    h.run(); o.run()
    rh, ro = h.result(), o.result()
    for key in ("z", "logw", "score", "carry"):
        same(rh[key], ro[key], key)
    assert rh["log_z"] == ro["log_z"]
    z = ro["z"]
    assert int(z.min()) >= 0 and int(z.max()) < 16


def test_scan_full_size(hip_ops):
    """N = 1e6, T = 100 (the reference's literal config-3 semantics under ImportanceK): weights and scores of the
    one-launch scan agree with a float64 recomputation from the stored [T, N] trajectories."""
    n, T = 1_000_000, 100
    m = W.LGSSM
    for impl in IMPLS:
        r = W.lgssm_scan(hip_ops, impl, seed=2, n=n, T=T)
        x = r["x"].double()  # [T, n]
        y = torch.from_numpy(W.lgssm_data(T)).to(x.device).double()[:, None]
        prev = torch.cat([torch.zeros(1, n, device=x.device, dtype=torch.float64), x[:-1]], 0)
        lp = lambda v, mu, s: -0.5 * ((v - mu) / s) ** 2 - math.log(s) - 0.5 * math.log(2 * math.pi)
        w = lp(y, x, m["r"]).sum(0)
        sc = w + lp(x, m["a"] * prev, m["q"]).sum(0)
        assert float((r["logw"].double() - w).abs().max()) < 2e-3 * T ** 0.5
        assert float((r["score"].double() - sc).abs().max()) < 4e-3 * T ** 0.5
        assert torch.equal(r["carry"], r["x"][-1])
        ref = float(torch.logsumexp(r["logw"].double(), 0).cpu()) - np.log(n)
        assert abs(r["log_z"] - ref) < 1e-5
        std = x.std(1)  # the prior's stationary spread: 1 / sqrt(1 - a^2)
        assert abs(float(std[-1]) - 1.0 / math.sqrt(1.0 - m["a"] ** 2)) < 0.02


@pytest.mark.parametrize("impl", IMPLS)
def test_smc_plans_ess_adaptive(hip_ops, oracle_ops, impl):
    """ESS-adaptive resampling in GENERATED filters (user models): flags, per-step pairs, particles, accumulated weights
    and ancestors equal the oracle's, for one filter and for a batch of three."""
    from genjax._amd import prng

    T, n = 16, 20000
    y = W.lgssm_data(T)
    obs2 = np.stack([y, (np.arange(T) % 2).astype(np.float32)], axis=1)
    _, hr = _smc_plans(hip_ops)
    _, orr = _smc_plans(oracle_ops)
    sk, rk = W.smc_key_schedule(prng.key(13, impl), T)
    h = hip_ops.smc_run_plan(hr, impl, n, sk, rk, obs2, True, ess_threshold=0.5, want_flags=True)
    o = oracle_ops.smc_run_plan(orr, impl, n, sk, rk, obs2, True, ess_threshold=0.5, want_flags=True)
    same(h[5], o[5], "resampled flags")
    assert 0 < int((o[5][1:] == 0).sum()) < T - 1
    same(h[0], o[0], "step max"); same(h[1], o[1], "step q"); same(h[3], o[3], "logw"); same(h[4], o[4], "ancestors")
    for a, b in zip(h[2], o[2]):
        same(a, b, "state column")
    pairs = [W.smc_key_schedule(prng.key(20 + f, impl), T) for f in range(3)]
    skf, rkf = np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])
    hb = hip_ops.smc_run_plan(hr, impl, n, skf, rkf, obs2, True, ess_threshold=0.5, want_flags=True)
    ob = oracle_ops.smc_run_plan(orr, impl, n, skf, rkf, obs2, True, ess_threshold=0.5, want_flags=True)
    same(hb[5], ob[5], "batched flags"); same(hb[1], ob[1], "batched step q"); same(hb[4][:, :, :n], ob[4][:, :, :n], "batched ancestors")


def test_full_size_properties(hip_ops):
    """BASELINE sizes (1e6 particles) through size-independent properties: closed-form log Z within
    Monte-Carlo error, weight-sum invariants, monotone ancestors with floor/ceil offspring counts."""
    n = 1_000_000
    for impl in IMPLS:
        r = W.gaussian10_importance(hip_ops, impl, seed=0, n=n)
        assert abs(r["log_z"] - r["log_z_exact"]) < 0.2, (r["log_z"], r["log_z_exact"])
        lw = r["logw"]
        ref = float(torch.logsumexp(lw.double(), 0).cpu()) - np.log(n)
        assert abs(r["log_z"] - ref) < 1e-5, "fixed-point log-sum-exp vs float64 log-sum-exp"
        s = W.lgssm_smc(hip_ops, impl, seed=1, n=n, T=100)
        assert abs(s["log_z"] - s["log_z_exact"]) < 0.2, (s["log_z"], s["log_z_exact"])  # ~6 sigma of the estimator
        a, _, _ = hip_ops.resample("systematic", KeyBatch(impl, 2, parent=(1, 1)), lw)
        a = a.cpu()
        assert bool((a[1:] >= a[:-1]).all())
        cnt = torch.bincount(a.long(), minlength=n).double()
        w = torch.softmax(lw.double().cpu(), 0) * n
        assert float((cnt - w).abs().max()) < 1.0 + 1e-4


def test_full_size_hmm(hip_ops):
    """BASELINE configs[4] at full size as a test (not only in bench.py): 1e6 particles, 256 states, T = 500 —
    log Z against the forward algorithm (float64), ancestors monotone at a late step, states within range."""
    n, T = 1_000_000, 500
    for impl in IMPLS:
        r = W.hmm_smc(hip_ops, impl, seed=2, n=n, T=T, want_ancestors=False)
        assert abs(r["log_z"] - r["log_z_exact"]) < 0.35, (r["log_z"], r["log_z_exact"])
        st = r["state"]
        assert int(st.min()) >= 0 and int(st.max()) < 256
        assert bool(torch.isfinite(r["out_e"]).all()) and bool((r["out_q"] > 0).all())


def test_full_size_batches(hip_ops):
    """The benchmark's launches at full size: 8 independent 1e6-particle ImportanceK passes in one launch and 16
    bootstrap filters (T = 20) in the same launches.  Every pass / filter has its own seed: the estimates differ,
    each agrees with the float64 log-sum-exp of its own log-weights, and their spread around the closed form is
    the Monte-Carlo error of one estimate."""
    n, L = 1_000_000, 8
    wl = W.Gaussian10(hip_ops, 1, seed=100, n_local=n)
    prep = wl.prepare(fold_batch=L, passes=L)
    prep.launch_passes(0, L)
    prep.launch_fold(L)
    exact = W.gaussian10_exact_log_z(wl.y)
    zs = [hip_ops.log_z_from_rows(prep.e_all[p], prep.q_all[p], n) for p in range(L)]
    assert len({round(z, 9) for z in zs}) == L, "independent passes"
    for p in range(L):
        ref = float(torch.logsumexp(prep.logw_all[p, :n].double(), 0).cpu()) - np.log(n)
        assert abs(zs[p] - ref) < 1e-5
    err = np.array(zs) - exact
    assert abs(err.mean()) < 0.1 and err.std() < 0.1, (err.mean(), err.std())
    # the variance of the weights fixes the estimator's standard deviation: sqrt(var(w) / n) / mean(w).  Both
    # sides are noisy (8 estimates; heavy-tailed weights in 10 dimensions): same order of magnitude is the claim
    w = torch.exp(prep.logw_all[0, :n].double() - zs[0])
    assert 0.2 < float(w.std() / np.sqrt(n)) / err.std() < 5.0, (float(w.std() / np.sqrt(n)), err.std())
    f = W.LgssmSMC(hip_ops, 1, 200, n, 20, filters=16)
    r = f.result(f.run())
    zf = np.array(r["log_z"])
    assert len({round(z, 9) for z in zf}) == 16
    assert abs(zf.mean() - W.lgssm_exact_log_z(f.y)) < 0.1 and zf.std() < 0.1, (zf, W.lgssm_exact_log_z(f.y))


@pytest.mark.parametrize("form", ["pair", "quad", "one"])
@pytest.mark.parametrize("n", [1000, 70004, 1_000_000])
def test_importance_fast_math_tolerance(hip_ops, oracle_ops, n, form, monkeypatch):
    """GJX_PLAN_FAST_MATH (hardware log / exp / sqrt / sin / cos for the continuous parts of the walk): the same
    particles from the same counters, every latent value and log-weight within the north star's 1e-5 relative
    bound of the exact specification (the oracle), log Z within 1e-4 absolute."""
    monkeypatch.setenv("GJX_PLAN_JIT", "1")
    monkeypatch.setenv("GJX_JIT_FORM", form)
    h = W.gaussian10_importance(hip_ops, 1, seed=5, n=n, fast_math=True)
    o = W.gaussian10_importance(oracle_ops, 1, seed=5, n=n)
    lw_h, lw_o = h["logw"].cpu().double(), o["logw"].double()
    rel = ((lw_h - lw_o).abs() / lw_o.abs().clamp_min(1e-30)).max().item()
    assert rel <= 1e-5, f"log-weights: max relative deviation {rel:.3g}"
    sc_h, sc_o = h["score"].cpu().double(), o["score"].double()
    assert ((sc_h - sc_o).abs() / sc_o.abs().clamp_min(1e-30)).max().item() <= 1e-5
    for a, b in zip(h["values"], o["values"]):
        a, b = a.cpu().double(), b.double()
        # a standard normal near 0 has no meaningful relative error: 1e-5 relative OR 2e-6 absolute (|z| <~ 6)
        err = (a - b).abs()
        assert bool(((err <= 1e-5 * b.abs()) | (err <= 2e-6)).all()), f"latent values: max abs deviation {err.max().item():.3g}"
    assert abs(h["log_z_rows"] - o["log_z_rows"]) <= 1e-4
    if form == "pair":  # the exact plan on the same device still gives the oracle's bits (the flag is per plan)
        same(W.gaussian10_importance(hip_ops, 1, seed=5, n=n)["logw"], o["logw"], "exact plan next to a fast one")


def test_population_near_the_index_limit(hip_ops, oracle_ops):
    """2 000 000 001 particles (ancestors are int32: the cap is 2^31 - 1): two steps of the LGSSM filter — byte offsets
    beyond 2^32, 1.95 million tiles, the precomputed tile prefix — equal the oracle bit for bit.  32 GB on the device and
    as much on the host; skipped where that does not fit."""
    import psutil

    n, T = 2_000_000_001, 2
    free, _ = torch.cuda.mem_get_info()
    if free < 48 * 2**30 or psutil.virtual_memory().available < 96 * 2**30:
        pytest.skip("needs 48 GiB of device memory and 96 GiB of host memory")
    h = W.lgssm_smc(hip_ops, 1, seed=3, n=n, T=T)
    got = {k: h[k].cpu() for k in ("out_e", "out_q", "state", "logw")}
    log_z = h["log_z"]
    del h
    torch.cuda.empty_cache()
    o = W.lgssm_smc(oracle_ops, 1, seed=3, n=n, T=T)
    for k, v in got.items():
        assert torch.equal(v, o[k]), k
    assert log_z == o["log_z"] and abs(log_z - o["log_z_exact"]) < 1e-3


def test_importance_beyond_4g_bytes_per_column(hip_ops, oracle_ops):
    """1 200 000 004 particles of a two-site model: every output column is 4.8 GB, so the specialised kernel's 16-byte
    stores land beyond byte offset 2^32.  Values, scores, weights and the row records equal the oracle's."""
    import psutil

    n = 1_200_000_004
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2**30 or psutil.virtual_memory().available < 96 * 2**30:
        pytest.skip("needs 40 GiB of device memory and 96 GiB of host memory")
    sites = W.gaussian10_sites(W.gaussian10_data())[:2]
    kb = W.importance_particle_keys(prng.key(9, 1), n)

    def run(ops):
        plan = ops.plan_create(sites)
        vals, score, logw, mp, rows = ops.importance_run(plan, kb, n, [], [torch.float32], want_rows=True)
        lse = ops.lse_rows(rows)
        return [vals[0].cpu(), score.cpu(), logw.cpu(), mp.cpu(), rows.e.cpu(), rows.s.cpu()] + [x.cpu() for x in lse]

    got = run(hip_ops)
    torch.cuda.empty_cache()
    want = run(oracle_ops)
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"output {i}"


def test_random_resampling_fuzz(hip_ops):
    """tests/fuzz_resample.py: random sizes and weight patterns through the generic resamplers, the log-sum-exp and short
    collapsing filters — the heavy-tile / idle-tile / extra-workgroup paths of the resampling kernel against the oracle."""
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_resample.py"), "20", "5"], capture_output=True, text=True,
                       timeout=400)
    assert r.returncode == 0 and "resample fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seconds,seed,p_invalid", [("25", "7", "0"), ("15", "3", "0.2")])
def test_random_plans_fuzz(hip_ops, seconds, seed, p_invalid):
    """A short run of tests/fuzz_parity.py (random site tables, sizes, generators; importance, scan and generated-SMC
    plans): every output of the HIP library equals the oracle's bit for bit — also when a fifth of the constants,
    observations, parameters and carries are NaN / infinite / out of the domain.  (A child process: the tool
    owns its backends.)"""
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), seconds, seed, p_invalid], capture_output=True,
                       text=True, timeout=400)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("kind", ["lgssm", "hmm"])
@pytest.mark.parametrize("ess", [0.0, 0.5])
def test_smc_runs_replayed_as_one_graph(hip_ops, oracle_ops, impl, kind, ess):
    """r04: from the second run of a shape on, a whole one-filter run is ONE hipGraph launch whose keys, observations and
    model scalars come from a device block (gjx_smc_run_graph_stats).  Five runs over the SAME buffers (the caching allocator
    hands a released output back), each with another seed — and, for the LGSSM, another model and other observations: every
    run equals the oracle's bit for bit, and at least three of them were replays."""
    import gc

    n, T = 50_000, 12
    before = hip_ops.smc_run_graph_stats()
    for rep in range(5):
        seed = 31 + rep
        if kind == "lgssm":
            y = (W.lgssm_data(T) + np.float32(0.25 * rep)).astype(np.float32)
            mdl = abi.Lgssm(0.1 * rep, 1.0 + 0.1 * rep, 0.9 - 0.05 * rep, 1.0, 0.8 + 0.1 * rep)
            h = W.LgssmSMC(hip_ops, impl, seed, n, T, True, ess_threshold=ess, y=y, model=mdl)
            o = W.LgssmSMC(oracle_ops, impl, seed, n, T, True, ess_threshold=ess, y=y, model=mdl)
        else:
            h = W.HmmSMC(hip_ops, impl, seed, n, T, n_states=16, want_ancestors=True, ess_threshold=ess)
            o = W.HmmSMC(oracle_ops, impl, seed, n, T, n_states=16, want_ancestors=True, ess_threshold=ess)
        hr, orr = h.result(h.run()), o.result(o.run())
        torch.cuda.synchronize()
        same(hr["ancestors"], orr["ancestors"], f"ancestors, run {rep}")
        same(hr["out_e"], orr["out_e"]); same(hr["out_q"], orr["out_q"])
        same(hr["state"], orr["state"]); same(hr["logw"], orr["logw"])
        if ess > 0:
            same(hr["resampled"], orr["resampled"], "resampling flags")
        assert hr["log_z"] == orr["log_z"]
        del hr, h
        gc.collect()
    after = hip_ops.smc_run_graph_stats()
    if os.environ.get("GJX_SMC_GRAPH") != "0":
        assert after["replays"] - before["replays"] >= 3, (before, after)


def test_smc_run_graphs_are_evicted_and_rebuilt(hip_ops, oracle_ops):
    """More shapes than the library keeps graphs for (16): every shape runs three times (plain, capture, replay), old graphs are
    destroyed as new ones come, and a shape that comes back is captured again — each result equal to the oracle's."""
    import gc

    if os.environ.get("GJX_SMC_GRAPH") == "0":
        pytest.skip("run graphs switched off")
    before = hip_ops.smc_run_graph_stats()
    shapes = [4096 + 1024 * i for i in range(20)] + [4096, 5120]
    for n in shapes:
        o = W.lgssm_smc(oracle_ops, 1, seed=3, n=n, T=4)
        for _ in range(3):
            h = W.lgssm_smc(hip_ops, 1, seed=3, n=n, T=4)
            torch.cuda.synchronize()
            same(h["state"], o["state"]); same(h["out_q"], o["out_q"])
            assert h["log_z"] == o["log_z"]
            del h
            gc.collect()
    after = hip_ops.smc_run_graph_stats()
    assert after["captures"] - before["captures"] >= 20 and after["replays"] - before["replays"] >= 20, (before, after)
