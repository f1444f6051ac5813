"""CPU tests that pin the ORACLE (the checker) before it is trusted: Random123 known answers,
scipy float64 log-densities, accuracy of the math spec, the reference tests' closed-form answers,
distributional checks of the samplers, resampling invariants, and the committed regression vectors."""

import ctypes as C
import json
import math
import os

import numpy as np
import pytest
import torch
from scipy import special, stats

from genjax._amd import workloads as W
from genjax._amd.ops import KeyBatch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def raw(oracle_ops):
    return oracle_ops.lib._dll


def test_cipher_known_answers(raw):
    kat = gold("rng_kat.json")
    for v in kat["threefry2x32_20"]:
        k = (C.c_uint32 * 2)(*[int(x, 16) for x in v["key"]])
        c = (C.c_uint32 * 2)(*[int(x, 16) for x in v["ctr"]])
        o = (C.c_uint32 * 2)()
        raw.gjo_threefry2x32(k, c, o)
        assert [f"{x:08x}" for x in o] == v["out"]
    for v in kat["philox4x32_10"]:
        k = (C.c_uint32 * 2)(*[int(x, 16) for x in v["key"]])
        c = (C.c_uint32 * 4)(*[int(x, 16) for x in v["ctr"]])
        o = (C.c_uint32 * 4)()
        raw.gjo_philox4x32(k, c, o)
        assert [f"{x:08x}" for x in o] == v["out"]


def test_jax_key_tree_is_threefry_of_counter(oracle_ops, raw):
    """split(k, n)[i] == fold_in(k, i) == threefry2x32(k, (0, i)) — jax_threefry_partitionable
    semantics."""
    key = (0x13198A2E, 0x03707344)
    ks = oracle_ops.rng_keys(KeyBatch(0, 1, parent=key, first=0), 5).numpy().view(np.uint32)
    for i in range(5):
        k = (C.c_uint32 * 2)(*key)
        c = (C.c_uint32 * 2)(0, i)
        o = (C.c_uint32 * 2)()
        raw.gjo_threefry2x32(k, c, o)
        assert tuple(ks[i]) == (o[0], o[1])
        f = oracle_ops.rng_keys(KeyBatch(0, 2, parent=key).with_fold(i), 1).numpy().view(np.uint32)[0]
        assert tuple(f) == (o[0], o[1])
    # 32 random bits of a key = hi ^ lo of block (0, 0)
    b = oracle_ops.rng_bits(KeyBatch(0, 2, parent=key), 1).numpy().view(np.uint32)[0]
    k = (C.c_uint32 * 2)(*key); c = (C.c_uint32 * 2)(0, 0); o = (C.c_uint32 * 2)()
    raw.gjo_threefry2x32(k, c, o)
    assert b == o[0] ^ o[1]


def _math(raw, fn, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    raw.gjo_math(fn, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), C.c_uint64(x.size))
    return y


def test_box_muller_against_float64(raw):
    """The table-driven Box-Muller transform of the PHILOX Normal sites (gjx_device.hpp bm_pair): against float64 on the
    same two words the normals are within 2.5e-7 of the radius (edge words, every table boundary, u -> 1, 2e6 random
    pairs); u = 1 gives radius 0; and the constants in both headers are the generator's (tools/gen_bm_tables.py)."""
    import subprocess
    import sys

    root = os.path.dirname(HERE)
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_bm_tables.py"), "--check"]).returncode == 0
    rng = np.random.default_rng(0)
    n = 2_000_000
    wr = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    wa = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    edge = np.array([0, 1, 2, 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFF00, 0xFFFFFE00, 0xFFFF0000, 0x80000000, 0x7FFFFFFF,
                     0xB504F333, 0xB504F334], dtype=np.uint32)
    wr[:edge.size] = edge
    k = np.arange(256, dtype=np.uint64)
    wa[100:356] = (k << 24).astype(np.uint32)
    wa[400:656] = ((k << 24) | 0xFFFFFF).astype(np.uint32)
    wa[700:956] = ((k << 24) | 0x800000).astype(np.uint32)
    wr[1000:200000] = (2**32 - 1 - rng.integers(0, 2**26, 199000)).astype(np.uint32)  # u close to 1
    zc, zs = np.empty(n, np.float32), np.empty(n, np.float32)
    raw.gjo_bm_pair(wr.ctypes.data_as(C.c_void_p), wa.ctypes.data_as(C.c_void_p), zc.ctypes.data_as(C.c_void_p),
                    zs.ctypes.data_as(C.c_void_p), C.c_uint64(n))
    u = ((wr.astype(np.float32) + np.float32(1.0)) * np.float32(2.3283064365386963e-10)).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(u))
    th = 2.0 * np.pi * (wa >> 8).astype(np.float64) / 2.0**24
    assert np.isfinite(zc).all() and np.isfinite(zs).all()
    assert zc[3] == 0.0 and zs[3] == 0.0  # u == 1
    for got, ref in ((zc, r * np.cos(th)), (zs, r * np.sin(th))):
        assert np.max(np.abs(got - ref) - 2.5e-7 * r) <= 0.0
    assert stats.kstest(zc[200000:].astype(np.float64), "norm").pvalue > 1e-3
    assert stats.kstest(zs[200000:].astype(np.float64), "norm").pvalue > 1e-3


def test_math_spec_accuracy(raw):
    rng = np.random.default_rng(0)
    x = np.exp(rng.uniform(-80, 80, 100000)).astype(np.float32)
    assert np.max(np.abs(_math(raw, 0, x) - np.log(x.astype(np.float64))) / np.maximum(1, np.abs(np.log(x.astype(np.float64))))) < 2e-7
    x = rng.uniform(-85, 85, 100000).astype(np.float32)
    assert np.max(np.abs(_math(raw, 1, x) / np.exp(x.astype(np.float64)) - 1)) < 3e-7
    x = rng.uniform(-1, 1, 100000).astype(np.float32)
    ref = special.erfinv(x.astype(np.float64))
    assert np.max(np.abs(_math(raw, 2, x) - ref) / np.maximum(1e-3, np.abs(ref))) < 1e-6
    x = np.exp(rng.uniform(-6, 6, 100000)).astype(np.float32)
    ref = special.gammaln(x.astype(np.float64))
    assert np.max(np.abs(_math(raw, 3, x) - ref) / np.maximum(1, np.abs(ref))) < 1e-5
    assert _math(raw, 0, [0.0])[0] == -np.inf and _math(raw, 1, [-200.0])[0] == 0.0


def map_inputs():
    import torch

    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(5000, generator=g) * 20, torch.rand(3000, generator=g) * 1e-3, torch.randn(2000, generator=g) * 90,
                   torch.tensor([0.0, -0.0, 1.0, -1.0, float("inf"), float("-inf"), float("nan"), 1e-45, 1e-39, 3.4e38, 88.0,
                                 88.5, 88.72, 88.73, 89.0, -86.0, -87.5, -104.0, 0.5, 2.0, 3.0])]).to(torch.float32)
    return x


def test_map_f32_against_numpy(oracle_ops):
    """gjx_map_f32 (what `exp` / `log` / a division by a number between the sites of a model body compute, on the per-site
    path and — as GJX_EXPR_EXP / _LOG / _DIV — inside fused plans): exp and log within 2 ulp of float64, the IEEE edge cases
    (log 0 = -inf, log of a negative = NaN, exp overflow = +inf, NaN stays NaN), divisions equal to numpy's f32 bit for bit."""
    import torch

    from genjax._amd import abi

    x = map_inputs()
    xn = x.numpy()
    with np.errstate(all="ignore"):
        for op, ref in ((abi.MAP_EXP, np.exp(xn.astype(np.float64))), (abi.MAP_LOG, np.log(xn.astype(np.float64)))):
            got = oracle_ops.map_f32(op, x).numpy()
            r32 = ref.astype(np.float32)
            fin = np.isfinite(r32) & (np.abs(r32) > 1.2e-38)
            assert np.array_equal(np.isnan(got), np.isnan(r32))
            big = ~np.isnan(r32) & ~fin & (np.abs(r32) > 1.0)  # overflow to +-inf
            assert np.array_equal(got[big], r32[big])
            ulp = np.spacing(np.abs(r32[fin]))
            tol = np.where((xn[fin] > 88.0) | (xn[fin] < -86.0), 4.0, 2.0) if op == abi.MAP_EXP else 2.0  # (the tails square exp(x / 2))
            assert np.all(np.abs(got[fin].astype(np.float64) - ref[fin]) <= tol * ulp), op
        for c in (3.0, 0.1, -7.25, 1e-30):
            assert np.array_equal(oracle_ops.map_f32(abi.MAP_DIV, x, c).numpy(), xn / np.float32(c), equal_nan=True)
            assert np.array_equal(oracle_ops.map_f32(abi.MAP_RDIV, x, c).numpy(), np.float32(c) / xn, equal_nan=True)
        assert np.array_equal(oracle_ops.map_f32(abi.MAP_SQRT, x).numpy(), np.sqrt(xn), equal_nan=True)  # correctly rounded
        assert np.array_equal(oracle_ops.map_f32(abi.MAP_ABS, x).numpy(), np.abs(xn), equal_nan=True)
    assert oracle_ops.map_f32(abi.MAP_EXP, torch.zeros(3, 4)).shape == (3, 4)


def test_logpdfs_against_scipy(oracle_ops):
    g = gold("logpdf_scipy.json")
    for r in g["normal"]:
        got = float(oracle_ops.logpdf("normal", 1, r["x"], r["loc"], r["scale"]))
        assert got == pytest.approx(r["logpdf"], rel=2e-6, abs=2e-6)
    for r in g["gamma"]:
        got = float(oracle_ops.logpdf("gamma", 1, r["x"], r["concentration"], r["rate"]))
        assert got == pytest.approx(r["logpdf"], rel=1e-5, abs=1e-5)
    for r in g["beta"]:
        got = float(oracle_ops.logpdf("beta", 1, r["x"], r["a"], r["b"]))
        assert got == pytest.approx(r["logpdf"], rel=1e-5, abs=2e-5)
    for r in g["bernoulli"]:
        got = float(oracle_ops.logpdf("bernoulli", 1, bool(r["x"]), r["p"]))
        assert got == pytest.approx(r["logpdf"], rel=1e-6)
    for r in g["categorical"]:
        lg = torch.tensor([r["logits"]])
        for k, want in enumerate(r["logpdf"]):
            assert float(oracle_ops.logpdf_categorical(1, k, lg)) == pytest.approx(want, rel=1e-6)


def test_reference_known_answers(oracle_ops):
    r = gold("reference_kat.json")
    # tests/generative_functions/test_static_gen_fn.py:317-318
    s = float(oracle_ops.logpdf("normal", 1, 1.0, 0.0, 1.0)) + float(oracle_ops.logpdf("normal", 1, -1.0, 0.0, 1.0))
    assert s == pytest.approx(r["assess_two_std_normals_y1_1_y2_m1"], abs=1e-6)
    assert float(oracle_ops.logpdf("normal", 1, 0.5, 0.0, 1.0)) == pytest.approx(r["normal_logpdf_0p5_0_1"], abs=1e-6)
    assert float(oracle_ops.logpdf("bernoulli", 1, True, 0.7)) == pytest.approx(r["flip_flip_trivial_logZ"], abs=1e-7)


@pytest.mark.parametrize("impl", [0, 1])
def test_sampler_distributions(oracle_ops, impl):
    n = 400000
    kb = KeyBatch(impl, 1, parent=(123, 456), first=0)
    v, _ = oracle_ops.sample_logpdf("normal", kb.with_fold(1), n, 1.0, 2.0)
    assert stats.kstest(v.numpy().astype(np.float64), "norm", args=(1.0, 2.0)).pvalue > 1e-3
    for a, b in ((2.5, 3.0), (0.3, 1.0), (1.0, 0.5)):
        v, _ = oracle_ops.sample_logpdf("gamma", kb.with_fold(2), n, a, b)
        assert stats.kstest(v.numpy().astype(np.float64), "gamma", args=(a, 0, 1 / b)).pvalue > 1e-3
    for a, b in ((2.0, 2.0), (0.5, 3.0)):
        v, _ = oracle_ops.sample_logpdf("beta", kb.with_fold(3), n, a, b)
        assert stats.kstest(v.numpy().astype(np.float64), "beta", args=(a, b)).pvalue > 1e-3
    v, _ = oracle_ops.sample_logpdf("bernoulli", kb.with_fold(4), n, 0.3)
    assert abs(float(v.float().mean()) - 0.3) < 4 * math.sqrt(0.21 / n)
    lg = torch.tensor([[0.1, -1.0, 2.0, 0.5, float("-inf")]])
    p = torch.softmax(lg[0].double(), 0).numpy()
    for mode in (0, 1):
        v, s = oracle_ops.sample_logpdf_categorical(kb.with_fold(5), n, lg, None, mode)
        cnt = np.bincount(v.numpy(), minlength=5)
        assert cnt[4] == 0
        assert stats.chisquare(cnt[:4], p[:4] * n).pvalue > 1e-3
        assert np.allclose(s.numpy(), np.log(p)[v.numpy()], atol=1e-6)


@pytest.mark.parametrize("impl", [0, 1])
def test_closed_form_log_z(oracle_ops, impl):
    g = W.gaussian10_importance(oracle_ops, impl, seed=0, n=400000)
    assert abs(g["log_z"] - g["log_z_exact"]) < 0.35
    ref = float(torch.logsumexp(g["logw"].double(), 0)) - math.log(400000)
    assert abs(g["log_z"] - ref) < 1e-6 and abs(g["lse"] - math.log(400000) - ref) < 1e-5
    # bootstrap-filter log Z: unbiased, std ~0.03 at this size (measured over seeds) -> 6 sigma
    s = W.lgssm_smc(oracle_ops, impl, seed=1, n=200000, T=40)
    assert abs(s["log_z"] - s["log_z_exact"]) < 0.2
    h = W.hmm_smc(oracle_ops, impl, seed=2, n=20000, T=30, n_states=32)
    assert abs(h["log_z"] - h["log_z_exact"]) < 0.15


@pytest.mark.parametrize("impl", [0, 1])
def test_resampling_invariants(oracle_ops, impl):
    g = torch.Generator().manual_seed(7)
    for n, n_out in ((1, 1), (10, 50), (5000, 5000), (3000, 1000)):
        lw = torch.randn(n, generator=g) * 3
        if n > 5:
            lw[2] = float("-inf")
        key = KeyBatch(impl, 2, parent=(n, 1))
        a, m, q = oracle_ops.resample("systematic", key, lw, n_out)
        assert a.shape[0] == n_out and bool((a[1:] >= a[:-1]).all())
        cnt = torch.bincount(a.long(), minlength=n).double()
        w = torch.softmax(lw.double(), 0) * n_out
        assert float((cnt - w).abs().max()) < 1 + 1e-6  # counts in {floor, ceil}
        if n > 5:
            assert cnt[2] == 0
        # the merged anchor is the power of two just above the maximum; the (e, q) pair reproduces logsumexp
        assert int(m) == math.ceil(float(lw.max() * torch.tensor(1.44269504088896341, dtype=torch.float32)))
        lse = (int(m) - 30) * math.log(2) + math.log(int(q))
        assert lse == pytest.approx(float(torch.logsumexp(lw.double(), 0)), abs=1e-6)
        a2, _, _ = oracle_ops.resample("multinomial", key, lw, n_out)
        assert int(a2.min()) >= 0 and int(a2.max()) < n
    # multinomial frequencies
    lw = torch.tensor([0.0, 1.0, 2.0, -1.0])
    a, _, _ = oracle_ops.resample("multinomial", KeyBatch(impl, 2, parent=(9, 9)), lw, 200000)
    cnt = np.bincount(a.numpy(), minlength=4)
    assert stats.chisquare(cnt, torch.softmax(lw.double(), 0).numpy() * 200000).pvalue > 1e-3


def test_logsumexp_edge_cases(oracle_ops):
    for x in ([0.0], [-1e30, 0.0], [5.0] * 7, [float("-inf"), -3.0], [-1e4, -1e4 + 1.0], [300.0, 299.0]):
        t = torch.tensor(x)
        lse, m, q = oracle_ops.logsumexp(t)
        ref = float(torch.logsumexp(t.double(), 0))
        assert float(lse) == pytest.approx(ref, abs=1e-6, rel=1e-6)
        rl, e, rq = oracle_ops.lse_rows(oracle_ops.row_stats(t))  # row-anchored form
        assert float(rl) == pytest.approx(ref, abs=1e-6, rel=1e-6)
        assert oracle_ops.log_z_from_rows(e, rq, 1) == pytest.approx(ref, abs=1e-6, rel=1e-7)
    assert float(oracle_ops.lse_rows(oracle_ops.row_stats(torch.full((700,), float("-inf"))))[0]) == float("-inf")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(100000, generator=g) * 30  # rows with very different anchors
    rl, e, rq = oracle_ops.lse_rows(oracle_ops.row_stats(x))
    assert oracle_ops.log_z_from_rows(e, rq, 1) == pytest.approx(float(torch.logsumexp(x.double(), 0)), abs=1e-7)


def lse_record_cases():
    """Row-stat sets whose anchors spread over 0..70 binades (rows beyond 63 drop out), with empty rows."""
    from genjax._amd.ops import RowStats

    g = torch.Generator().manual_seed(11)
    for n_rows, spread in ((1, 0), (9, 3), (700, 70), (5000, 5), (5000, 70)):
        e = (torch.randint(0, spread + 1, (n_rows,), generator=g) * (torch.rand(n_rows, generator=g) < 0.3) - 40).int()
        s = torch.randint(1 << 29, 1 << 38, (n_rows,), generator=g)
        if n_rows > 5:
            e[2], s[2] = -(1 << 30), 0  # an empty row (all -inf)
        yield RowStats(e, s, n_rows * 256)


def check_lse_records(ops, to_dev=lambda t: t):
    """Shared with the GPU suite: records of row-aligned shards merge into exactly the unsharded result."""
    from genjax._amd.ops import RowStats

    out = []
    for rows in lse_record_cases():
        n_rows = rows.e.numel()
        e_all, s_all = to_dev(rows.e), to_dev(rows.s)
        rec = to_dev(torch.zeros(65, dtype=torch.int64))
        lse, e, q = ops.lse_rows(RowStats(e_all, s_all, rows.n), record=rec)
        assert int(rec[0].cpu()) == int(e.cpu())
        assert sum(int(b) >> d for d, b in enumerate(rec[1:].cpu().tolist())) == int(q.cpu())
        for cuts in ((0, n_rows), (0, n_rows // 3, n_rows), (0, 1, n_rows // 2, n_rows // 2 + 1, n_rows)):
            cuts = sorted(set(cuts))
            recs = to_dev(torch.zeros((len(cuts) - 1, 2, 65), dtype=torch.int64))  # batch of 2: slot 1 used
            for r, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                ops.lse_rows(RowStats(e_all[a:b].clone(), s_all[a:b].clone(), (b - a) * 256), record=recs[r, 1])
            recs[:, 0, 0] = -(1 << 30)  # slot 0: empty records
            cl, ce, cq = ops.lse_combine(recs)
            assert float(cl[0].cpu()) == float("-inf") and int(cq[0].cpu()) == 0
            assert int(ce[1].cpu()) == int(e.cpu()) and int(cq[1].cpu()) == int(q.cpu())
            assert torch.equal(cl[1:2].cpu(), lse.cpu())
        # float64 check of the bucket arithmetic itself
        ed, sd = rows.e.double(), rows.s.double()
        live = rows.e > -(1 << 30)
        ref = torch.logsumexp(ed[live] * math.log(2.0) + torch.log(sd[live]) - 30 * math.log(2.0), 0)
        assert ops.log_z_from_rows(e, q, 1) == pytest.approx(float(ref), abs=1e-6)
        out.append((rec.cpu().clone(), lse.cpu().clone()))
    return out


def test_lse_records_merge_exactly(oracle_ops):
    check_lse_records(oracle_ops)


def test_regression_vectors(oracle_ops):
    check_regression(oracle_ops)
    check_regression_r02(oracle_ops)


def check_regression(ops):
    """Shared with the GPU suite: the committed oracle outputs must be reproduced exactly."""
    reg = gold("oracle_regression.json")
    dev = ops.device()
    for impl, nm in ((0, "threefry"), (1, "philox")):
        r = reg[nm]
        kb = KeyBatch(impl, 1, parent=(0, 42), first=0)
        assert ops.rng_keys(kb, 4).cpu().view(-1).tolist() == r["keys"]
        assert ops.rng_bits(kb.with_fold(1), 4).cpu().tolist() == r["bits_fold1"]
        v, s = ops.sample_logpdf("normal", kb.with_fold(1), 4, 0.0, 1.0)
        assert v.cpu().view(torch.int32).tolist() == r["normal_bits"]
        assert s.cpu().view(torch.int32).tolist() == r["normal_score_bits"]
        v, _ = ops.sample_logpdf("gamma", kb.with_fold(2), 4, 0.7, 2.0)
        assert v.cpu().view(torch.int32).tolist() == r["gamma_bits"]
        v, _ = ops.sample_logpdf("beta", kb.with_fold(3), 4, 2.0, 2.0)
        assert v.cpu().view(torch.int32).tolist() == r["beta_bits"]
        v, _ = ops.sample_logpdf("bernoulli", kb.with_fold(4), 8, 0.3)
        assert v.cpu().tolist() == r["bernoulli"]
        lw = torch.linspace(-3, 2, 37).to(dev)
        a, m, q = ops.resample("systematic", KeyBatch(impl, 2, parent=(5, 6)), lw)
        assert a.cpu().tolist() == r["systematic_ancestors"] and int(q.cpu()) == r["systematic_q"] and int(m.cpu()) == r["systematic_e"]
        a, _, _ = ops.resample("multinomial", KeyBatch(impl, 2, parent=(5, 6)), lw, 12)
        assert a.cpu().tolist() == r["multinomial_ancestors"]
        assert int(ops.categorical_index(KeyBatch(impl, 2, parent=(5, 6)), lw, 0).cpu()) == r["categorical_index_gumbel"]
        assert int(ops.categorical_index(KeyBatch(impl, 2, parent=(5, 6)), lw, 1).cpu()) == r["categorical_index_invcdf"]
        g = W.gaussian10_importance(ops, impl, seed=3, n=2048)
        assert g["q"] == r["gaussian10_q"]
        assert g["logw"][:4].cpu().view(torch.int32).tolist() == r["gaussian10_logw_head_bits"]
        s_ = W.lgssm_smc(ops, impl, seed=4, n=2048, T=6, want_ancestors=True)
        assert s_["out_q"].cpu().tolist() == r["lgssm_q"] and s_["out_e"].cpu().tolist() == r["lgssm_e"]
        assert s_["ancestors"][5, :16].cpu().tolist() == r["lgssm_anc_t5_head"]
        h = W.hmm_smc(ops, impl, seed=5, n=2048, T=6, n_states=16)
        assert h["out_q"].cpu().tolist() == r["hmm_q"] and h["out_e"].cpu().tolist() == r["hmm_e"]


def check_regression_r02(ops):
    """Round-2 pins (tests/golden/oracle_regression_r02.json): the one-launch scan, ESS-adaptive schedule, collapse."""
    import numpy as np

    from genjax._amd import abi, prng

    reg = gold("oracle_regression_r02.json")
    for impl, nm in ((0, "threefry"), (1, "philox")):
        r = reg[nm]
        sc = W.lgssm_scan(ops, impl, seed=6, n=1500, T=9)
        assert sc["logw"][:6].cpu().view(torch.int32).tolist() == r["scan_lgssm_logw_head_bits"]
        assert sc["x"][8, :6].cpu().view(torch.int32).tolist() == r["scan_lgssm_x_t8_head_bits"]
        hs = W.HmmScan(ops, impl, 7, 1200, 8, n_states=16, cat_mode=1)
        hs.run()
        hr = hs.result()
        assert hr["z"][7, :12].cpu().tolist() == r["scan_hmm_z_t7_head"]
        assert hr["logw"][:4].cpu().view(torch.int32).tolist() == r["scan_hmm_logw_head_bits"]
        ad = W.lgssm_smc(ops, impl, seed=8, n=3000, T=16, want_ancestors=True, ess_threshold=0.5)
        assert ad["resampled"].cpu().tolist() == r["ess_flags"] and ad["out_q"].cpu().tolist() == r["ess_q"]
        assert ad["out_e"].cpu().tolist() == r["ess_e"]
        assert ad["logw"][:4].cpu().view(torch.int32).tolist() == r["ess_logw_head_bits"]
        y = np.array([0.1, 25.0, -40.0, -39.5, 60.0, 60.2], dtype=np.float32)
        sk, rk = W.smc_key_schedule(prng.key(11, impl), 6)
        col = ops.smc_run_lgssm(impl, 20000, sk, rk, abi.Lgssm(0.0, 1.0, 0.9, 1.0, 0.05), y, True)
        assert col[1].cpu().tolist() == r["collapse_q"] and col[0].cpu().tolist() == r["collapse_e"]
        assert int(col[4][2].unique().numel()) == r["collapse_anc_t2_distinct"]
        assert col[4][5, :8].cpu().tolist() == r["collapse_anc_t5_head"]


def check_hmm_alias(ops, k):
    """The alias table of every transition row encodes the row's softmax: summing, over the K columns, the
    accept mass of the column and the reject mass handed to its alias reproduces softmax(logits) to the table's
    24-bit threshold resolution; thresholds are below 2^24 and aliases are valid states."""
    from genjax._amd import workloads as W

    tl, ol = W.hmm_tables(k)
    dev = ops.device()
    tab, logp = ops.hmm_prepare(k, 0, torch.from_numpy(tl).contiguous().to(dev), torch.from_numpy(ol).contiguous().to(dev))
    tab = tab.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert tab.shape == (k, k)
    thr, alias = tab >> 8, tab & 255
    assert (alias < k).all() and (thr <= 0xFFFFFF).all()
    p = np.exp(tl.astype(np.float64) - tl.max(axis=1, keepdims=True))
    p /= p.sum(axis=1, keepdims=True)
    acc = thr / float(1 << 24)
    acc[(alias == np.arange(k)[None, :])] = 1.0  # a column that is its own alias always yields itself
    got = np.zeros((k, k))
    for r in range(k):
        np.add.at(got[r], np.arange(k), acc[r] / k)
        np.add.at(got[r], alias[r], (1.0 - acc[r]) / k)
    assert np.abs(got - p).max() < 4e-7  # cat_fix resolution 2^-23 per weight, threshold resolution 2^-24 / K
    lp = ol.astype(np.float64) - ol.max(axis=1, keepdims=True)
    lp = lp - np.log(np.exp(lp).sum(axis=1, keepdims=True))
    assert np.abs(logp.cpu().numpy() - lp).max() < 1e-5


@pytest.mark.parametrize("k", [2, 3, 17, 64, 256])
def test_hmm_alias_tables(oracle_ops, k):
    check_hmm_alias(oracle_ops, k)
