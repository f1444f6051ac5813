"""Host-API test bodies shared by the CPU suite (oracle backend injected with `use_ops`) and the
GPU suite (the product's HIP backend).  They read like the reference's own tests for the path:
tests/inference/test_smc.py, tests/generative_functions/{test_static_gen_fn,test_distributions,
test_scan_combinator}.py and README.md:89-123."""

import math
import warnings

import pytest
import torch

import genjax
from genjax import ChoiceMap, ChoiceMapBuilder as C, SelectionBuilder as S, Target, beta, categorical, flip, gamma, gen, normal
from genjax._amd import jaxlike
from genjax._amd.lang import StaticTrace
from genjax._amd.plan import try_fused_generate
from genjax.inference.smc import (BootstrapSMC, ChangeTarget, Importance, ImportanceK, LinearGaussianSSM,
                                  StateSpaceModel)

jax = jaxlike
jnp = jaxlike.jnp


def f(x):
    return float(x.detach().cpu()) if isinstance(x, torch.Tensor) else float(x)


# ---- tests/inference/test_smc.py ------------------------------------------------------------------
def case_exact_flip_flip_trivial(impl):
    @gen
    def flip_flip_trivial():
        _ = flip(0.5) @ "x"
        _ = flip(0.7) @ "y"

    key = genjax.random.key(314159, impl)
    problem = Target(flip_flip_trivial, (), C["y"].set(True))
    z_exact = f(flip.assess(problem.constraint.get_submap("y"), (0.7,))[0])
    assert z_exact == pytest.approx(math.log(0.7), abs=1e-7)
    z = Importance(problem).log_marginal_likelihood_estimate(key)
    assert f(z) == pytest.approx(z_exact, rel=1e-1)
    z = ImportanceK(problem, k_particles=1000).log_marginal_likelihood_estimate(key)
    assert f(z) == pytest.approx(z_exact, rel=1e-3)


def case_exact_flip_flip(impl):
    @gen
    def flip_flip():
        v1 = flip(0.5) @ "x"
        p = jax.lax.cond(v1, lambda: 0.9, lambda: 0.3)
        _ = flip(p) @ "y"

    key = genjax.random.key(314159, impl)
    problem = Target(flip_flip, (), C["y"].set(True))
    assert problem["y"] is True
    z = ImportanceK(problem, k_particles=2000).log_marginal_likelihood_estimate(key)
    assert f(z) == pytest.approx(math.log(0.6), rel=1e-1)
    z = ImportanceK(problem, k_particles=200000).log_marginal_likelihood_estimate(key)
    assert f(z) == pytest.approx(math.log(0.6), abs=1e-2)
    # r03: the reference's own body (tests/inference/test_smc.py:59-66: `jax.lax.cond` on a flip value) is ONE fused kernel
    # (the selection is GJX_EXPR_SELECT), equal to the per-site path bit for bit
    import torch

    from genjax._amd.lang import GenerateHandler

    keys = genjax.random.split(genjax.random.key(7, impl), 5000)
    fused = try_fused_generate(flip_flip, keys, C["y"].set(True), ())
    assert fused is not None, "lax.cond on a traced flip value must lower to the fused kernel"
    h = GenerateHandler(keys, C["y"].set(True))
    h.run(flip_flip.source, ())
    assert torch.equal(fused[1], h.weight) and torch.equal(fused[0].get_choices()["x"], h.traces["x"].get_choices().get_value())


def case_non_marginal_target(impl):
    @gen
    def model():
        idx = categorical(probs=[0.5, 0.25, 0.25]) @ "idx"
        means = jnp.array([0.0, 10.0, 11.0])
        vars_ = jnp.array([1.0, 1.0, 1.0])
        x = normal(means[idx], vars_[idx]) @ "x"
        y = normal(means[idx], vars_[idx]) @ "y"
        return x, y

    marginal_model = model.marginal(selection=S["x"] | S["y"])
    with pytest.raises(TypeError):
        Target(marginal_model, (), C["x"].set(1.0))
    # the model itself is a fine target
    t = Target(model, (), C["x"].set(10.5))
    z = ImportanceK(t, k_particles=50000).log_marginal_likelihood_estimate(genjax.random.key(1, impl))
    exact = math.log(0.5 * _npdf(10.5, 0) + 0.25 * _npdf(10.5, 10) + 0.25 * _npdf(10.5, 11))
    assert f(z) == pytest.approx(exact, abs=0.05)


def _npdf(x, m, s=1.0):
    return math.exp(-0.5 * ((x - m) / s) ** 2) / (s * math.sqrt(2 * math.pi))


# ---- README.md:89-123 -------------------------------------------------------------------------------
def case_readme_beta_bernoulli(impl):
    @gen
    def beta_bernoulli(a, b):
        p = beta(a, b) @ "p"
        v = flip(p) @ "v"
        return v

    def run_inference(obs: bool):
        posterior_target = Target(beta_bernoulli, (2.0, 2.0), ChoiceMap.d({"v": obs}))
        alg = ImportanceK(posterior_target, k_particles=50)
        key = jax.random.key(314159, impl)
        sub_keys = jax.random.split(key, 50)
        est, p_chm = jax.vmap(alg.random_weighted, in_axes=(0, None))(sub_keys, posterior_target)
        assert p_chm["p"].shape == (50,) and est.shape == (50,)
        # r04: the trial axis runs batched (two launches per 32 trials); element b equals the scalar call bit for bit
        assert alg.random_weighted_batch(sub_keys, posterior_target) is not None
        for b in (0, 1, 17, 31, 32, 49):
            e_b, c_b = alg.random_weighted(sub_keys[b], posterior_target)
            assert torch.equal(torch.as_tensor(e_b).reshape(()).cpu(), est[b].cpu()) and torch.equal(c_b["p"].reshape(()).cpu(), p_chm["p"][b].cpu())
        return f(jnp.mean(p_chm["p"]))

    assert run_inference(True) == pytest.approx(0.6, abs=0.07)  # README prints 0.6039314
    assert run_inference(False) == pytest.approx(0.4, abs=0.07)  # README prints 0.3679334
    t = Target(beta_bernoulli, (2.0, 2.0), ChoiceMap.d({"v": True}))
    z = ImportanceK(t, k_particles=100000).log_marginal_likelihood_estimate(jax.random.key(2, impl))
    assert f(z) == pytest.approx(math.log(0.5), abs=1e-2)


# ---- tests/generative_functions/test_static_gen_fn.py --------------------------------------------
def case_static_gen_fn(impl):
    @gen
    def model():
        y1 = normal(0.0, 1.0) @ "y1"
        y2 = normal(0.0, 1.0) @ "y2"
        return y1 + y2

    key = genjax.random.key(314159, impl)
    score, retval = model.assess(C.kw(y1=1.0, y2=-1.0), ())  # test_static_gen_fn.py:317-318
    assert f(score) == pytest.approx(-2.837877, abs=1e-6) and f(retval) == 0.0
    tr = model.simulate(key, ())
    ch = tr.get_choices()
    assert f(tr.get_score()) == pytest.approx(f(model.assess(ch, ())[0]), rel=1e-6)
    assert f(tr.get_retval()) == pytest.approx(f(ch["y1"]) + f(ch["y2"]), rel=1e-6)
    # importance: weight = sum of constrained-site log-densities; 0 without constraints (441-489)
    tr2, w = model.importance(key, C["y2"].set(0.5), ())
    assert f(w) == pytest.approx(f(normal.logpdf(0.5, 0.0, 1.0)), rel=1e-6)
    assert f(tr2.get_choices()["y2"]) == 0.5
    assert f(tr2.get_score()) == pytest.approx(f(w) + f(normal.logpdf(tr2.get_choices()["y1"], 0.0, 1.0)), rel=1e-5)
    tr3, w0 = model.importance(key, C.n(), ())
    assert f(w0) == 0.0
    # address reuse (static.py:213-216)
    @gen
    def bad():
        normal(0.0, 1.0) @ "x"
        normal(0.0, 1.0) @ "x"

    with pytest.raises(genjax.AddressReuse):
        bad.simulate(key, ())
    with pytest.raises(genjax.MissingAddress):
        model.assess(C.kw(y1=1.0), ())
    # tuple addresses and nested calls (test_core.py:27-38, test_static_gen_fn.py:253-256)
    @gen
    def inner(m):
        return normal(m, 1.0) @ "z"

    @gen
    def outer():
        a = normal(0.0, 1.0) @ ("x", "x0")
        b = inner(a) @ "sub"
        return b

    tr = outer.simulate(key, ())
    ch = tr.get_choices()
    assert ("x", "x0") in ch and ("sub", "z") in ch and "x" not in ch
    tr4, w4 = outer.importance(key, C["sub", "z"].set(2.0), ())
    assert f(w4) == pytest.approx(f(normal.logpdf(2.0, f(tr4.get_choices()["x", "x0"]), 1.0)), rel=1e-5)
    # kwargs to distributions (test_distributions.py:490-520)
    @gen
    def kw():
        a = normal(loc=0.0, scale=0.1) @ "a"
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            c = categorical([0.0, 1.0]) @ "c"
            assert any(issubclass(r.category, DeprecationWarning) for r in rec)
        d = categorical(logits=[0.0, 1.0]) @ "d"
        return a

    kw.simulate(key, ())


# ---- tests/generative_functions/test_static_gen_fn.py:500-700, test_distributions.py update tests ----
def case_update(impl):
    from genjax import Diff, Update

    key = genjax.random.key(314159, impl)

    # distributions (distribution.py:179-258): re-score at new arguments, or replace the value
    key, sub_key = jax.random.split(key)
    tr = normal.simulate(sub_key, (0.0, 1.0))
    v = f(tr.get_retval())
    new_tr, w, rd, discard = tr.update(key, C.n())
    assert f(w) == 0.0 and rd.tangent is genjax.NoChange and discard.static_is_empty()
    new_tr, w, rd, discard = tr.update(key, C.n(), Diff.unknown_change((1.0, 2.0)))
    assert f(new_tr.get_retval()) == v and rd.tangent is genjax.NoChange
    assert f(w) == pytest.approx(f(normal.logpdf(v, 1.0, 2.0)) - f(tr.get_score()), abs=1e-6)
    assert f(new_tr.get_score()) == pytest.approx(f(normal.logpdf(v, 1.0, 2.0)), abs=1e-6)
    new_tr, w, rd, discard = tr.update(key, C.v(1.5))
    assert f(new_tr.get_retval()) == 1.5 and rd.tangent is genjax.UnknownChange and f(discard.get_value()) == v
    assert f(w) == pytest.approx(f(normal.logpdf(1.5, 0.0, 1.0)) - f(tr.get_score()), abs=1e-6)

    # test_simple_normal_update / test_simple_linked_normal_update
    @gen
    def simple_linked_normal():
        y1 = normal(0.0, 1.0) @ "y1"
        y2 = normal(y1, 1.0) @ "y2"
        y3 = normal(y1 + y2, 1.0) @ "y3"
        return y1 + y2 + y3

    key, sub_key = jax.random.split(key)
    tr = simple_linked_normal.simulate(sub_key, ())
    original_choice, original_score = tr.get_choices(), f(tr.get_score())
    new = C["y1"].set(2.0)
    key, sub_key = jax.random.split(key)
    updated, w, rd, discard = simple_linked_normal.update(sub_key, tr, new, ())
    _, w_edit, _, bwd = tr.edit(sub_key, Update(new))  # update_weight_correctness_general_assertions: same move
    assert f(w_edit) == f(w) and isinstance(bwd, Update)
    uc = updated.get_choices()
    y1, y2, y3 = f(uc["y1"]), f(uc["y2"]), f(uc["y3"])
    assert y1 == 2.0 and y2 == f(original_choice["y2"]) and y3 == f(original_choice["y3"])
    test_score = f(normal.assess(C.v(y1), (0.0, 1.0))[0]) + f(normal.assess(C.v(y2), (y1, 1.0))[0]) + \
        f(normal.assess(C.v(y3), (y1 + y2, 1.0))[0])
    assert f(original_choice["y1"]) == f(discard["y1"]) and "y2" not in discard
    assert f(updated.get_score()) == pytest.approx(original_score + f(w), rel=1e-5)
    assert f(updated.get_score()) == pytest.approx(test_score, rel=1e-5)
    assert f(updated.get_retval()) == pytest.approx(y1 + y2 + y3, rel=1e-6) and rd.tangent is genjax.UnknownChange
    # weight correctness: w = sum over changed factors of new − old log-density
    oy1 = f(original_choice["y1"])
    expect = (f(normal.logpdf(2.0, 0.0, 1.0)) - f(normal.logpdf(oy1, 0.0, 1.0))
              + f(normal.logpdf(y2, 2.0, 1.0)) - f(normal.logpdf(y2, oy1, 1.0))
              + f(normal.logpdf(y3, 2.0 + y2, 1.0)) - f(normal.logpdf(y3, oy1 + y2, 1.0)))
    assert f(w) == pytest.approx(expect, abs=1e-4)
    # the backward request undoes the move
    back, w_back, _, _ = bwd.edit(sub_key, updated, ())
    assert f(back.get_choices()["y1"]) == oy1 and f(w_back) == pytest.approx(-f(w), abs=1e-4)
    # two addresses at once
    upd2, w2, _, d2 = simple_linked_normal.update(sub_key, tr, C["y1"].set(2.0).at["y2"].set(3.0), ())
    assert f(upd2.get_score()) == pytest.approx(original_score + f(w2), rel=1e-5) and "y2" in d2 and "y3" not in d2

    # test_simple_hierarchical_normal: nested generative functions, argument changes flow into them
    @gen
    def _inner(x):
        return normal(x, 1.0) @ "y1"

    @gen
    def simple_hierarchical_normal():
        y1 = normal(0.0, 1.0) @ "y1"
        y2 = _inner(y1) @ "y2"
        y3 = _inner(y1 + y2) @ "y3"
        return y1 + y2 + y3

    key, sub_key = jax.random.split(key)
    tr = simple_hierarchical_normal.simulate(sub_key, ())
    oc, os_ = tr.get_choices(), f(tr.get_score())
    updated, w, _, discard = simple_hierarchical_normal.update(sub_key, tr, C["y1"].set(2.0), ())
    uc = updated.get_choices()
    assert f(uc["y1"]) == 2.0 and f(uc["y2", "y1"]) == f(oc["y2", "y1"]) and f(uc["y3", "y1"]) == f(oc["y3", "y1"])
    y2, y3 = f(uc["y2", "y1"]), f(uc["y3", "y1"])
    test_score = f(normal.logpdf(2.0, 0.0, 1.0)) + f(normal.logpdf(y2, 2.0, 1.0)) + f(normal.logpdf(y3, 2.0 + y2, 1.0))
    assert f(oc["y1"]) == f(discard["y1"])
    assert f(updated.get_score()) == pytest.approx(os_ + f(w), rel=1e-5)
    assert f(updated.get_score()) == pytest.approx(test_score, rel=1e-5)

    # argdiffs: a model argument changes, no constraint (generative_function.py:496-560 docstring example)
    @gen
    def model(var):
        v1 = normal(0.0, 1.0) @ "v1"
        v2 = normal(v1, var) @ "v2"
        return v2

    tr = model.simulate(sub_key, (1.0,))
    new_tr, w, rd, _ = tr.edit(key, Update(C.n()), Diff.unknown_change((3.0,)))
    ch = tr.get_choices()
    assert new_tr.get_args() == (3.0,) and f(new_tr.get_choices()["v2"]) == f(ch["v2"])
    assert f(w) == pytest.approx(f(normal.logpdf(f(ch["v2"]), f(ch["v1"]), 3.0)) - f(normal.logpdf(f(ch["v2"]), f(ch["v1"]), 1.0)), abs=1e-5)
    same, w0, rd0, _ = tr.update(key, C.n())
    assert f(w0) == 0.0 and rd0.tangent is genjax.NoChange
    with pytest.raises(genjax.NotSupportedEditRequest):
        tr.edit(key, genjax._amd.edit.PrimitiveEditRequest())  # (an unknown primitive request; Regenerate & co. are answered)
    with pytest.raises(genjax.MissingAddress):  # a previous trace that never visited "y1" (static.py:432-436)
        simple_linked_normal.update(sub_key, StaticTrace(simple_linked_normal, (), None, {}), C.n(), ())

    # a whole population at once: per-particle keys, [n] columns, one log-density kernel per site
    n = 4096
    keys = jax.random.split(key, n)
    ptr = simple_linked_normal.simulate(keys, ())
    pupd, pw, _, pdisc = ptr.update(keys, C["y2"].set(0.25))
    y1c, y3c = ptr.get_choices()["y1"], ptr.get_choices()["y3"]
    assert pw.shape == (n,) and torch.equal(pupd.get_choices()["y1"], y1c) and torch.equal(pdisc["y2"], ptr.get_choices()["y2"])
    y2o = ptr.get_choices()["y2"]
    expect = (normal.logpdf(torch.full_like(y1c, 0.25), y1c, 1.0) - normal.logpdf(y2o, y1c, 1.0)
              + normal.logpdf(y3c, y1c + 0.25, 1.0) - normal.logpdf(y3c, y1c + y2o, 1.0))
    assert torch.allclose(pw, expect, atol=1e-4)
    assert torch.allclose(pupd.get_score(), ptr.get_score() + pw, atol=1e-4)
    # a masked constraint (distribution.py:214-243): the value is replaced on SOME particles only
    from genjax import Mask

    flag = (torch.arange(n, device=y2o.device) % 3) == 0
    mupd, mw, _, mdisc = ptr.update(keys, C["y2"].set(Mask(torch.full_like(y2o, 0.25), flag)))
    y2n = mupd.get_choices()["y2"]
    assert torch.equal(y2n[flag], torch.full_like(y2o, 0.25)[flag]) and torch.equal(y2n[~flag], y2o[~flag])
    assert torch.equal(mw[~flag], torch.zeros_like(mw)[~flag])  # untouched particles: weight exactly 0
    assert torch.allclose(mw[flag], pw[flag], atol=1e-5) and torch.allclose(mupd.get_score(), ptr.get_score() + mw, atol=1e-4)
    dm = mdisc["y2"]
    assert isinstance(dm, Mask) and torch.equal(dm.flag, flag) and torch.equal(dm.value, y2o)
    same_tr, w_none, _, _ = ptr.update(keys, C["y2"].set(Mask(torch.full_like(y2o, 0.25), False)))  # a static False flag: no-op
    assert torch.equal(same_tr.get_choices()["y2"], y2o) and float(torch.as_tensor(w_none).abs().max()) == 0.0

    # combinators answer Update by re-generation (scan: test_scan_combinator.py update tests)
    @gen
    def step(carry, x):
        z = normal(carry + x, 1.0) @ "z"
        return z, z

    scanner = genjax.scan(n=3)(step)
    xs = torch.tensor([0.5, -0.5, 1.0])
    str_ = scanner.simulate(sub_key, (0.0, xs))
    supd, sw, _, sdisc = str_.update(key, C[1, "z"].set(0.75))
    zs = str_.get_choices()[:, "z"] if False else [f(str_.get_choices()[i, "z"]) for i in range(3)]
    assert f(supd.get_choices()[1, "z"]) == 0.75 and f(supd.get_choices()[0, "z"]) == zs[0]
    exp = (f(normal.logpdf(0.75, zs[0] - 0.5, 1.0)) - f(normal.logpdf(zs[1], zs[0] - 0.5, 1.0))
           + f(normal.logpdf(zs[2], 0.75 + 1.0, 1.0)) - f(normal.logpdf(zs[2], zs[1] + 1.0, 1.0)))
    assert f(sw) == pytest.approx(exp, abs=1e-4) and f(sdisc[1, "z"]) == zs[1]


# ---- tests/generative_functions/test_distributions.py:25-60 ---------------------------------------
def case_distributions(impl):
    key = genjax.random.key(314159, impl)
    for dist, args, v in ((normal, (0.0, 1.0), 0.5), (gamma, (2.0, 3.0), 1.5), (beta, (2.0, 2.0), 0.3), (flip, (0.3,), True)):
        tr = dist.simulate(key, args)
        assert f(tr.get_score()) == pytest.approx(f(dist.assess(tr.get_choices(), args)[0]), rel=1e-6)
        tr, w = dist.importance(key, C.n(), args)
        assert f(w) == 0.0
        tr, w = dist.importance(key, C.v(v), args)
        assert f(w) == f(tr.get_score()) == pytest.approx(f(dist.logpdf(v, *args)), rel=1e-6)
    assert f(normal.logpdf(0.5, 0.0, 1.0)) == pytest.approx(-1.0439385332, abs=1e-6)
    assert f(beta.logpdf(0.3, 2.0, 2.0)) == pytest.approx(0.2311117210, abs=1e-5)
    assert f(gamma.logpdf(1.5, 2.0, 3.0)) == pytest.approx(-1.8973103146, abs=1e-5)
    lp = [f(categorical.logpdf(k, logits=[-0.3, -0.5])) for k in (0, 1)]
    assert lp == pytest.approx([-0.59813887, -0.79813887], abs=1e-6)
    # the same scalar key gives the same draw; batches give per-particle draws
    a, b = normal.sample(key, 0.0, 1.0), normal.sample(key, 0.0, 1.0)
    assert f(a) == f(b)
    col = normal.sample(genjax.random.split(key, 1000), 0.0, 1.0)
    assert col.shape == (1000,) and abs(f(col.mean())) < 0.15


# ---- batched execution: fused kernel == per-site column kernels, bit for bit -------------------------
def case_uniform(impl):
    """`genjax.uniform(low, high)` (tensorflow_probability/__init__.py:294): range, moments, log-density, inside a model."""
    from genjax import uniform

    key = genjax.random.key(1, impl)
    assert 0.0 <= f(uniform.sample(key, 0.0, 1.0)) < 1.0
    assert f(uniform.logpdf(0.3, 0.0, 2.0)) == pytest.approx(-math.log(2.0)) and f(uniform.logpdf(3.0, 0.0, 2.0)) == -math.inf
    keys = genjax.random.split(key, 40000)
    u = uniform.sample(keys, -1.0, 3.0)
    assert f(u.min()) >= -1.0 and f(u.max()) < 3.0
    assert f(u.mean()) == pytest.approx(1.0, abs=0.03) and f(u.var()) == pytest.approx(16.0 / 12.0, abs=0.03)

    @gen
    def m():
        a = uniform(0.0, 2.0) @ "a"
        _ = normal(a, 0.5) @ "y"
        return a

    tr, w = m.importance(keys, C["y"].set(1.0), ())
    a = tr.get_choices()["a"]
    assert torch.allclose(w, normal.logpdf(1.0, a, 0.5), atol=1e-5)
    assert torch.allclose(tr.get_score(), w - math.log(2.0), atol=1e-5)


def case_fused_equals_eager(impl):
    @gen
    def model(a):
        p = beta(2.0, a) @ "p"
        v = flip(p) @ "v"
        g = gamma(0.7, p * 2.0 + 0.5) @ "g"
        x = normal(g * 0.5, 1.5) @ "x"
        y = normal(x - 1.0, 0.5) @ "y"
        return x

    n = 5000
    keys = genjax.random.split(genjax.random.key(7, impl), n)
    ycol = torch.linspace(-1, 1, n)
    for chm in (C.n(), C["v"].set(True) | C["y"].set(0.25), C["y"].set(ycol.to(_dev())) | C["p"].set(0.4)):
        fused = try_fused_generate(model, keys, chm, (3.0,))
        assert fused is not None, "this model must lower to the fused kernel"
        from genjax._amd.lang import GenerateHandler

        h = GenerateHandler(keys, chm)
        retval = h.run(model.source, (3.0,))
        eager = StaticTrace(model, (3.0,), retval, h.traces)
        ftr, fw = fused
        ew = h.weight
        if isinstance(ew, torch.Tensor):
            assert torch.equal(fw, ew)
        else:
            assert float(fw.abs().max()) == 0.0 and ew == 0.0
        assert torch.equal(ftr.get_score(), eager.get_score())
        fc, ec = dict(ftr.get_choices().leaves()), dict(eager.get_choices().leaves())
        assert fc.keys() == ec.keys()
        for k in fc:
            a_, b_ = fc[k], ec[k]
            if isinstance(a_, torch.Tensor) and a_.dim():
                assert torch.equal(a_, b_.to(a_.dtype) if isinstance(b_, torch.Tensor) else torch.full_like(a_, b_)), k
        assert torch.equal(ftr.get_retval(), eager.get_retval())
        # a site score recomputed on demand from the fused trace equals the eager one
        assert torch.equal(ftr.get_subtrace("x").get_score(), eager.get_subtrace("x").get_score())


def case_expression_arguments(impl):
    """Site arguments that are arithmetic over SEVERAL earlier sites / inputs / arguments — `normal(w * x + b, s)`, a
    product of two traced values, a parameter times a site — lower to postfix programs (gjx.h GJX_ARG_EXPR) and run as ONE
    fused kernel; weights, scores, values, recomputed site scores and the returned expression equal the per-site column
    path bit for bit (every operator is one f32 rounding, in the order the body wrote it)."""
    from genjax._amd import plan as P
    from genjax._amd.lang import GenerateHandler

    xs = [0.5, -1.25, 2.0, 0.1]

    @gen
    def regression(s, slope_scale):
        w = normal(0.0, 1.0) @ "w"
        b = normal(0.0, 2.0) @ "b"
        k = flip(0.3) @ "k"
        for i, x in enumerate(xs):
            normal(w * x + b, s) @ ("y", i)                       # two sites in one argument
        q = normal((w - b) * (w - b) - 1.0, 0.7) @ "q"          # a product of traced values
        r = normal(slope_scale * w + 0.25 - q * 3.0, s * 2.0) @ "r"   # a launch parameter times a site, three terms
        u = gamma(w * w + 0.5, b * b + 1.0) @ "u"
        t = normal(k * 2.0 - u, 1.0) @ "t"                       # an integer-valued site in arithmetic
        d = normal(w / u - b / (u + 2.0), q * q + 0.1) @ "d"      # quotients of traced values: IEEE divisions
        return w * 2.0 + b * b - 1.0, -(q * r) / (d * d + 1.0)

    n = 4000
    keys = genjax.random.split(genjax.random.key(21, impl), n)
    col = torch.linspace(-2, 2, n).to(_dev())
    for chm in (C.n(),
                C["y", 0].set(0.3) | C["y", 1].set(-0.2) | C["y", 2].set(1.1) | C["y", 3].set(0.05) | C["t"].set(0.4),
                C["q"].set(col) | C["r"].set(-0.75) | C["y", 2].set(0.0)):
        fused = try_fused_generate(regression, keys, chm, (0.5, 1.75))
        assert fused is not None, "expression arguments must lower to the fused kernel"
        ftr, fw = fused
        h = GenerateHandler(keys, chm)
        retval = h.run(regression.source, (0.5, 1.75))
        eager = StaticTrace(regression, (0.5, 1.75), retval, h.traces)
        ew = h.weight
        if isinstance(ew, torch.Tensor):
            assert torch.equal(fw, ew)
        assert torch.equal(ftr.get_score(), eager.get_score())
        fc, ec = dict(ftr.get_choices().leaves()), dict(eager.get_choices().leaves())
        assert fc.keys() == ec.keys()
        for key_ in fc:
            a_, b_ = fc[key_], ec[key_]
            if isinstance(a_, torch.Tensor) and a_.dim():
                assert torch.equal(a_, b_.to(a_.dtype) if isinstance(b_, torch.Tensor) else torch.full_like(a_, b_)), key_
        for fr, er in zip(ftr.get_retval(), eager.get_retval()):
            assert torch.equal(fr, er)
        for addr in ("q", "r", "u", "t", "d", ("y", 1)):
            assert torch.equal(torch.as_tensor(ftr.get_subtrace(addr).get_score()).to(torch.float32).expand(n),
                               torch.as_tensor(eager.get_subtrace(addr).get_score()).to(torch.float32).expand(n)), addr
    # the same structure on another dataset: parameters, not a new kernel (the plan cache keys on the programs' CONTENT)
    tr1 = P._traced(regression, C["y", 0].set(0.3), n, (0.5, 1.75))
    tr2 = P._traced(regression, C["y", 0].set(-0.9), n, (0.8, -0.3))
    assert tr1 is not None and tr2 is not None and P._make_plan(tr1[0]) is P._make_plan(tr2[0])
    # r03: what reference bodies write between sites — exp / log of a traced value, a division by (or of) a NUMBER — lowers
    # too (GJX_EXPR_EXP / _LOG / _DIV), and the per-site path computes the same bits (lang.SpecTensor -> gjx_map_f32:
    # torch's device exp is not the spec's, and torch multiplies a device tensor by the reciprocal instead of dividing)
    @gen
    def ratio():
        a = normal(0.0, 1.0) @ "a"
        c = gamma(2.0, 1.0) @ "c"
        return normal(a, torch.exp(c * 0.1)) @ "d"

    @gen
    def halves():
        a = normal(0.0, 1.0) @ "a"
        b = normal(a / 3.0, 1.0) @ "b"
        return normal(2.0 / (b * b + 1.5), (a * a + 0.5).log().exp()) @ "d"

    @gen
    def logscale(s0):
        ls = normal(s0, 0.3) @ "log_sigma"
        x = normal(0.0, torch.exp(ls)) @ "x"
        _ = normal(x / 2, torch.log(torch.exp(ls) + 1.0)) @ "y"
        return torch.div(x, 4.0)

    def same_trace(fn, args, chm):
        fused = try_fused_generate(fn, keys, chm, args)
        assert fused is not None, f"{fn.source.__name__}: exp / log / division by a number must lower to the fused kernel"
        ftr, fw = fused
        h = GenerateHandler(keys, chm)
        retval = h.run(fn.source, args)
        eager = StaticTrace(fn, args, retval, h.traces)
        if isinstance(h.weight, torch.Tensor):
            assert torch.equal(fw, h.weight), fn.source.__name__
        assert torch.equal(ftr.get_score(), eager.get_score()), fn.source.__name__
        fc, ec = dict(ftr.get_choices().leaves()), dict(eager.get_choices().leaves())
        assert fc.keys() == ec.keys()
        for key_ in fc:
            a_, b_ = fc[key_], ec[key_]
            if isinstance(a_, torch.Tensor) and a_.dim():
                assert torch.equal(a_, b_.to(a_.dtype) if isinstance(b_, torch.Tensor) else torch.full_like(a_, b_)), key_
        assert torch.equal(torch.as_tensor(ftr.get_retval()), torch.as_tensor(eager.get_retval())), fn.source.__name__
        for addr in fc:
            a = addr if not (isinstance(addr, tuple) and len(addr) == 1) else addr[0]
            assert torch.equal(torch.as_tensor(ftr.get_subtrace(a).get_score()).to(torch.float32).expand(n),
                               torch.as_tensor(eager.get_subtrace(a).get_score()).to(torch.float32).expand(n)), addr

    xs_l = [-1.5, -0.3, 0.4, 1.2, 2.0]

    @gen
    def logistic(scale):  # Bayesian logistic regression: flip(sigmoid(w x + b)) per datum; sqrt / abs / reciprocal / square too
        w = normal(0.0, scale) @ "w"
        b = normal(0.0, 1.0) @ "b"
        for i, x in enumerate(xs_l):
            flip(torch.sigmoid(w * x + b)) @ ("y", i)
        r = gamma(torch.sqrt(w ** 2 + 1.0), torch.abs(b) + 0.5 + (b * 0.1) ** 3) @ "r"
        s2 = normal(torch.clamp(w, min=-0.5, max=0.7), torch.nn.functional.softplus(b)) @ "s2"
        _ = normal(torch.maximum(s2, torch.tensor(0.1)), torch.minimum(r, torch.tensor(1.5)) + 0.2) @ "s3"
        return normal(torch.reciprocal(r + 1.0), torch.square(w).sqrt() + 0.1) @ "t"

    @gen
    def mixture(sep):  # jnp.where on a flip value and on comparisons of traced values; conditions combined with & | ~
        k = flip(0.3) @ "k"
        u = normal(0.0, 1.0) @ "u"
        m = torch.where(k, sep, -sep)
        x = normal(m + torch.where(u > 0.5, u, u * 0.25), 0.6) @ "x"
        c = (x > -1.0) & (x < 1.0) | ~(u <= 0.0)
        s_ = torch.where(c, torch.tensor(0.4), torch.tensor(1.3))
        _ = normal(torch.where(torch.eq(k, 1), x, u), s_) @ "y"
        return torch.where(torch.ne(k, 0) & (x >= u), x, torch.minimum(x, u))

    @gen
    def long_args(a):  # arguments of 17-32 program entries (r02's limit was 16): softplus / sigmoid / where of affine forms
        w = normal(0.0, 1.0) @ "w"
        b = normal(0.0, 1.0) @ "b"
        s_ = gamma(2.0, 2.0) @ "s"
        y = normal(torch.nn.functional.softplus(w * a + b), torch.sigmoid(w * 0.5 - b) + torch.where(s_ > 1.0, s_, 1.0 / (s_ + 0.5))) @ "y"
        return torch.nn.functional.softplus(y * w + b * s_)

    for chm_m in (C.n(), C["y"].set(0.7)):
        same_trace(long_args, (0.8,), chm_m)

    ys = C["y", 0].set(False) | C["y", 1].set(False) | C["y", 2].set(True) | C["y", 3].set(True) | C["y", 4].set(True)
    for chm_m in (C.n(), C["y"].set(0.3), C["k"].set(torch.tensor(True)) | C["y"].set(-0.2)):
        same_trace(mixture, (1.2,), chm_m)
    for fn, args, chm in ((ratio, (), C["d"].set(0.1)), (halves, (), C.n()), (halves, (), C["d"].set(-0.3)),
                          (logscale, (0.2,), C["y"].set(0.7)), (logscale, (0.2,), C.n()), (logistic, (1.5,), ys),
                          (logistic, (1.5,), C.n()), (logistic, (1.5,), ys | C["t"].set(0.3))):
        same_trace(fn, args, chm)
    tr, w = ratio.importance(keys, C["d"].set(0.1), ())
    assert w.shape == (n,) and bool(torch.isfinite(w).all())


def _dev():
    from genjax._amd.runtime import get_ops

    return get_ops().device()


# ---- structure vs data: observations and scalar arguments are launch parameters -----------------------
def case_params_equal_constants(impl):
    """The same model on several datasets: (a) lowering observations / scalar arguments to GJX_ARG_PARAM gives the very
    bits the constant-folded plan gives, (b) a body that uses an argument non-affinely is traced with constants
    instead (and still fuses), (c) on the HIP build, a new dataset costs no compilation."""
    from genjax._amd import plan as P
    from genjax._amd.runtime import get_ops

    @gen
    def model(a, s):
        p = beta(2.0, a) @ "p"
        v = flip(p) @ "v"
        g = gamma(a * 0.5 + 0.2, p * 2.0 + 0.5) @ "g"
        x = normal(g * 0.5, s) @ "x"
        y = normal(x - 1.0, 0.5) @ "y"
        z = normal(y * 2.0, s * 3.0) @ "z"  # an observed value feeding a later site
        return y

    @gen
    def curved(s):
        x = normal(0.0, s ** 2) @ "x"
        y = normal(x, 1.0) @ "y"
        return x

    n = 3000
    keys = genjax.random.split(genjax.random.key(11, impl), n)
    ops = get_ops()

    def run(args, chm, use_params):
        orig = P.PlanTracer.__init__

        def init(self, constraint, n_, use_params_=True):
            orig(self, constraint, n_, use_params_ and use_params)

        P.PlanTracer.__init__ = init
        try:
            out = try_fused_generate(model, keys, chm, args)
        finally:
            P.PlanTracer.__init__ = orig
        assert out is not None
        tr, w = out
        ch = tr.get_choices()
        return w, tr.get_score(), ch["p"], ch["g"], ch["x"], tr.get_retval()

    datasets = [((3.0, 1.5), C["y"].set(0.25) | C["z"].set(-0.5) | C["v"].set(True)),
                ((2.25, 0.75), C["y"].set(-1.5) | C["z"].set(2.0) | C["v"].set(False)),
                ((7.0, 0.3), C["y"].set(3.25) | C["z"].set(0.125) | C["v"].set(True))]
    run(*datasets[0], True)  # (compiles the structure's kernel on the HIP build)
    before = ops.jit_stats()
    for args, chm in datasets:
        a, b = run(args, chm, True), run(args, chm, False)
        for u, v in zip(a, b):
            if isinstance(u, torch.Tensor):
                assert torch.equal(u, v)
            else:
                assert u == v
    after = ops.jit_stats()
    if ops.lib.device_type == "cuda":
        # the three constant-folded plans compile (at most) three kernels; the parameterised plan none
        assert after["compiles"] - before["compiles"] <= 3
        c0 = ops.jit_stats()["compiles"]
        run((1.125, 2.5), C["y"].set(0.7) | C["z"].set(0.9) | C["v"].set(True), True)
        assert ops.jit_stats()["compiles"] == c0, "a new dataset must not recompile the model's kernel"
    # non-affine use of an argument: second trace, with constants
    out = try_fused_generate(curved, keys, C["y"].set(0.5), (1.5,))
    assert out is not None
    tr, w = out
    from genjax._amd.lang import GenerateHandler

    h = GenerateHandler(keys, C["y"].set(0.5))
    h.run(curved.source, (1.5,))
    assert torch.equal(w, h.weight)


# ---- ParticleCollection / ChangeTarget / CSMC ---------------------------------------------------------
def case_trace_cache(impl):
    """The traced form of a body is reused only for an exact repetition of the call (same function, constraint object and
    argument values): repeated calls equal uncached ones bit for bit, another constraint / argument or an in-place change
    of a constraint tensor is traced afresh."""
    from genjax._amd import plan as P

    @gen
    def model(a):
        z = normal(0.0, a) @ "z"
        _ = normal(z * 0.5, 0.7) @ "y"
        _ = normal(z, 1.5) @ "w"
        return z

    n = 2000
    keys = genjax.random.split(genjax.random.key(17, impl), n)
    wcol = torch.linspace(-1.0, 1.0, n).to(_dev())
    chm = C["y"].set(0.3) | C["w"].set(wcol)

    def run(c, a):
        tr, w = model.importance(keys, c, (a,))
        return w.clone(), tr.get_choices()["z"].clone()

    P.TRACE_CACHE = False
    try:
        ref1, ref2 = run(chm, 2.0), run(chm, 3.0)
    finally:
        P.TRACE_CACHE = True
    for _ in range(3):  # the second and third calls hit the cache
        got = run(chm, 2.0)
        assert torch.equal(got[0], ref1[0]) and torch.equal(got[1], ref1[1])
    got = run(chm, 3.0)  # another argument value
    assert torch.equal(got[0], ref2[0]) and torch.equal(got[1], ref2[1])
    chm2 = C["y"].set(-0.4) | C["w"].set(wcol)  # another constraint
    assert not torch.equal(run(chm2, 2.0)[0], ref1[0])
    wcol.add_(0.5)  # the SAME objects, a tensor changed in place: must not reuse the column traced before
    P.TRACE_CACHE = False
    try:
        ref3 = run(chm, 2.0)
    finally:
        P.TRACE_CACHE = True
    got = run(chm, 2.0)
    assert torch.equal(got[0], ref3[0]) and not torch.equal(got[0], ref1[0])


def case_particle_collection(impl):
    @gen
    def model():
        x = normal(0.0, 1.0) @ "x"
        _ = normal(x, 0.5) @ "y"
        return x

    key = genjax.random.key(11, impl)
    t = Target(model, (), C["y"].set(1.0))
    alg = ImportanceK(t, k_particles=20000)
    coll = alg.run_smc(key)
    lw = coll.get_log_weights()
    assert lw.shape == (20000,) and len(coll) == 20000
    z = f(coll.get_log_marginal_likelihood_estimate())
    exact = math.log(_npdf(1.0, 0.0, math.sqrt(1.25)))
    assert z == pytest.approx(exact, abs=0.03)
    assert coll.log_marginal_likelihood_estimate_f64() == pytest.approx(z, abs=1e-5)
    assert z == pytest.approx(f(torch.logsumexp(lw.double(), 0)) - math.log(20000), abs=1e-5)
    p = coll.sample_particle(genjax.random.key(3, impl))
    assert p.get_choices()["x"].dim() == 0 and f(p.get_choices()["y"]) == 1.0
    tr5, w5 = coll[5]
    assert f(w5) == f(lw[5]) and f(tr5.get_choices()["x"]) == f(coll.get_particles().get_choices()["x"][5])
    # posterior mean of x | y=1 is 0.8: importance estimate and resampled estimate
    xs = coll.get_particles().get_choices()["x"]
    w = torch.softmax(lw.double(), 0)
    assert f((w * xs.double()).sum()) == pytest.approx(0.8, abs=0.03)
    for method in ("systematic", "multinomial"):
        rs = coll.resample(genjax.random.key(4, impl), method)
        assert f(rs.get_particles().get_choices()["x"].mean()) == pytest.approx(0.8, abs=0.03)
        assert f(rs.get_log_marginal_likelihood_estimate()) == pytest.approx(z, abs=1e-4)
        a = rs.ancestors.long()
        assert torch.equal(rs.get_particles().get_choices()["x"], xs[a])
    assert 1.0 < coll.effective_sample_size() < 20000
    # ChangeTarget to a different observation re-weights exactly by the likelihood ratio
    t2 = Target(model, (), C["y"].set(-0.5))
    c2 = ChangeTarget(alg, t2).run_smc(key)
    want = lw + normal.logpdf(-0.5, xs, 0.5) - normal.logpdf(1.0, xs, 0.5)
    assert torch.allclose(c2.get_log_weights(), want, atol=2e-5)
    z2 = f(alg.log_marginal_likelihood_estimate(key, t2))
    assert z2 == pytest.approx(math.log(_npdf(-0.5, 0.0, math.sqrt(1.25))), abs=0.1)
    # random_weighted: density estimate = score - log Z_hat
    score, chm = alg.random_weighted(genjax.random.key(5, impl), t)
    assert "x" in chm and "y" not in chm
    # conditional SMC keeps the retained particle last
    retained = C["x"].set(0.123)
    cs = ImportanceK(t, k_particles=16).run_csmc(genjax.random.key(6, impl), retained)
    assert len(cs) == 16 and f(cs.get_particles().get_choices()["x"][-1]) == pytest.approx(0.123)
    # estimate_logpdf mirrors smc.py:181-198 literally: score of a particle drawn from the conditional
    # collection minus the log-marginal estimate (finite, and a posterior log-density of SOME particle)
    est = ImportanceK(t, k_particles=64).estimate_logpdf(genjax.random.key(8, impl), retained, t)
    assert math.isfinite(f(est)) and f(est) < math.log(_npdf(0.8, 0.8, math.sqrt(0.2))) + 0.5
    one = Importance(t).run_csmc(genjax.random.key(9, impl), retained)
    assert len(one) == 1


# ---- custom proposals q (smc.py:256-258, 301-305; sp.py:217-238) ------------------------------------
def case_custom_proposal(impl):
    @gen
    def model():
        x = normal(0.0, 1.0) @ "x"
        _ = normal(x, 0.5) @ "y"
        return x

    @gen
    def exact_posterior(target):
        y = target["y"]
        _ = normal(0.8 * y, math.sqrt(0.2)) @ "x"  # the exact posterior of x | y

    t = Target(model, (), C["y"].set(1.0))
    q = exact_posterior.marginal()
    key = genjax.random.key(21, impl)
    # Mirrored reference behaviour (sp.py:217-230): with the default selection (all) a Marginal's
    # random_weighted returns weight = tr.project(~all) = 0, so the collection's weights are the
    # target's importance weights at the PROPOSED choices: log p(x) + log p(y | x).
    keys = genjax.random.split(key, 64)
    wq, chm = q.random_weighted(keys, t)
    assert f(torch.as_tensor(wq).abs().max()) == 0.0 and chm["x"].shape == (64,)
    assert f(chm["x"].mean()) == pytest.approx(0.8, abs=0.25)
    coll = ImportanceK(t, q, k_particles=4096).run_smc(key)
    lw, xs = coll.get_log_weights(), coll.get_particles().get_choices()["x"]
    assert lw.shape == (4096,)
    assert torch.allclose(lw, normal.logpdf(xs, 0.0, 1.0) + normal.logpdf(1.0, xs, 0.5), atol=2e-5)
    assert f(xs.mean()) == pytest.approx(0.8, abs=0.03) and f(xs.std()) == pytest.approx(math.sqrt(0.2), abs=0.02)
    # the batched proposal run equals the reference's per-key vmap, key by key
    k1, sub = genjax.random.split(key)
    pks = genjax.random.split(sub, 4096)
    for i in (0, 17, 4095):
        wi, ci = q.random_weighted(pks[i], t)
        assert f(ci["x"]) == f(xs[i])
    one = Importance(t, q).run_smc(key)
    x1 = one.get_particles().get_choices()["x"][0]
    assert f(one.get_log_weights()[0]) == pytest.approx(f(normal.logpdf(x1, 0.0, 1.0) + normal.logpdf(1.0, x1, 0.5)), abs=2e-5)
    cs = ImportanceK(t, q, k_particles=8).run_csmc(key, C["x"].set(0.5))
    assert len(cs) == 8 and f(cs.get_particles().get_choices()["x"][-1]) == pytest.approx(0.5)


# ---- scan (tests/generative_functions/test_scan_combinator.py:54-61) ------------------------------------
def case_scan(impl):
    @genjax.scan(n=10)
    @gen
    def chain(x, _):
        z = normal(x, 1.0) @ "z"
        return z, None

    key = genjax.random.key(314159, impl)
    tr = chain.simulate(key, (0.0, None))
    zs = tr.get_choices()["z"]
    assert zs.shape == (10,)
    assert f(tr.get_score()) == pytest.approx(f(chain.assess(tr.get_choices(), (0.0, None))[0]), rel=1e-5)
    for i in (0, 3, 9):
        tr, w = chain.importance(key, C[i, "z"].set(0.5), (0.0, None))
        zs = tr.get_choices()["z"]
        assert f(zs[i]) == 0.5
        prev = 0.0 if i == 0 else f(zs[i - 1])
        assert f(w) == pytest.approx(f(normal.assess(C.v(0.5), (prev, 1.0))[0]), rel=1e-5)
    # batched over particles: [n, T] leaves
    keys = genjax.random.split(key, 64)
    trb, wb = chain.importance(keys, C[2, "z"].set(0.5), (0.0, None))
    assert trb.get_choices()["z"].shape == (64, 10) and wb.shape == (64,)
    assert trb.get_score().shape == (64,)


def case_scan_edge_cases(impl):
    """tests/generative_functions/test_scan_combinator.py:380-461: the length inferred from the scanned inputs, a zero-length
    scan (no choices, importance with them is a no-op), inputs of different leading sizes, a population of keys."""
    key = genjax.random.key(314159, impl)

    @gen
    def walk_step(x, std):
        new_x = normal(x, std) @ "x"
        return new_x, new_x

    args = (0.0, torch.tensor([2.0, 4.0, 3.0, 5.0, 1.0]))
    for sc in (walk_step.scan(n=5), walk_step.scan()):
        tr = sc.simulate(key, args)
        assert torch.allclose(tr.get_choices()[:, "x"], tr.get_retval()[1]) and tuple(tr.get_choices()[:, "x"].shape) == (5,)
    with pytest.raises(ValueError, match="disagrees with leading axis sizes"):
        walk_step.scan(n=4).simulate(key, args)
    keys = genjax.random.split(key, 10)
    many = walk_step.scan().simulate(keys, args)
    assert tuple(many.get_score().shape) == (10,) and tuple(many.get_choices()[:, "x"].shape) == (10, 5)

    @gen
    def step(state, sigma):
        new_x = normal(state, sigma) @ "x"
        return (new_x, new_x + 1)

    empty = step.scan(n=0).simulate(key, (2.0, torch.arange(0, dtype=torch.float32)))
    assert empty.get_choices().static_is_empty() and f(empty.get_score()) == 0.0
    _, w = step.scan().importance(genjax.random.split(key)[1], empty.get_choices(), (2.0, 2.0 + torch.arange(0, dtype=torch.float32)))
    assert f(w) == 0.0

    @gen
    def foo(shift, d):
        x = normal(d["loc"], d["scale"]) @ "x"
        return x + shift, None

    with pytest.raises(ValueError, match="scan got values with different leading axis sizes: 2, 1."):
        foo.scan().simulate(key, (torch.tensor([1.0]), {"loc": torch.tensor([10.0, 12.0]), "scale": torch.tensor([1.0])}))


def case_scan_fused_equals_loop(impl):
    """The one-launch scan (`gjx_scan_run`) against the host loop of per-site launches (scan.py:237-294): same key
    chain, same arithmetic -> the same trace, weights, score and return value, bit for bit; plus ImportanceK over a
    `.scan(n=T)` target end to end."""
    from genjax._amd import combinators as CB

    @gen
    def lg_step(x, _):
        x2 = normal(0.9 * x, 1.0) @ "x"
        _ = normal(x2, 0.5) @ "y"
        return x2, x2

    @gen
    def rich_step(carry, u):  # tuple carry, a scanned input, discrete and positive sites, an output built from a site
        x, s = carry
        b = flip(0.3) @ "b"
        g = gamma(2.0, 1.5) @ "g"
        x2 = normal(0.5 * x, u) @ "x"  # (an argument is affine in ONE traced value: plan.py)
        p = beta(2.0, 3.0) @ "p"
        _ = normal(x2 * 2.0, 1.25) @ "y"
        _ = flip(p) @ "z"
        return (x2, s - 1.0), (g * 0.5 + 1.0, b)

    @gen
    def controlled_step(carry, u):  # arguments over SEVERAL traced values (postfix programs, gjx.h GJX_ARG_EXPR): a
        x, v = carry                # controlled, two-component state-space kernel
        v2 = normal(0.8 * v + 0.3 * u, 0.5) @ "v"
        x2 = normal(x + 0.1 * v2 - 0.05 * u * u, 0.25) @ "x"
        _ = normal(x2 * x2 - v2, 1.0) @ "y"
        return (x2, v2), (x2 - v2 * 2.0, v2)

    @gen
    def kinematic_step(carry, u):  # a DETERMINISTIC update of a carried component: the next carry is an expression
        x, v = carry
        v2 = normal(0.9 * v + 0.2 * u, 0.4) @ "x"
        _ = normal(x + 0.5 * v2, 0.6) @ "y"
        return (x + 0.5 * v2 - 0.01 * x * x, v2), v2 * 2.0 - u

    torch.manual_seed(0)
    trans, emit = torch.randn(6, 6).to(_dev()), torch.randn(6, 6).to(_dev())

    @gen
    def hmm_step(z, _):  # the shape of BASELINE configs[4]: categorical rows chosen by the carried state / the new state
        z2 = categorical(logits=trans[z]) @ "z"
        _ = categorical(logits=emit[z2]) @ "y"
        return z2, z2

    n, T = 3000, 9
    keys = genjax.random.split(genjax.random.key(21, impl), n)
    ys = torch.linspace(-1.0, 1.5, T)
    us = torch.linspace(0.5, 1.25, T).to(_dev())
    zs = torch.tensor([True, False, True, True, False, False, True, False, True])
    x0 = torch.linspace(-2.0, 2.0, n).to(_dev())
    runs = [(lg_step.scan(n=T), C["y"].set(ys), (0.0, None)),
            (lg_step.scan(n=T), C[torch.arange(T), "y"].set(ys), (x0, None)),  # per-particle initial carry
            (lg_step.scan(n=T), C.n(), (0.25, None)),
            (rich_step.scan(), C["y"].set(ys) | C["z"].set(zs), ((0.5, 3.0), us)),
            (rich_step.scan(), C["y"].set(ys), ((x0, 1.0), us)),
            (controlled_step.scan(), C["y"].set(ys), ((0.0, 1.0), us)),
            (controlled_step.scan(), C.n(), ((x0, -0.5), us)),
            (kinematic_step.scan(), C["y"].set(ys), ((0.0, 1.0), us)),
            (kinematic_step.scan(), C.n(), ((x0, 0.25), us)),
            (hmm_step.scan(n=T), C["y"].set(torch.tensor([1, 0, 3, 5, 2, 2, 4, 0, 1])), (0, None)),
            (hmm_step.scan(n=T), C.n(), (2, None))]

    def same(a, b):
        if isinstance(a, (tuple, list)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
            a, b = torch.as_tensor(a), torch.as_tensor(b)
            return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a.cpu(), b.cpu())
        return a == b

    for model, chm, args in runs:
        out = {}
        for fused in (True, False):
            CB.FUSED_SCAN = fused
            try:
                tr, w = model.generate(keys, chm, args)
                sim = model.simulate(keys, args)
            finally:
                CB.FUSED_SCAN = True
            assert isinstance(tr, CB.FusedScanTrace) == fused and isinstance(sim, CB.FusedScanTrace) == fused
            out[fused] = (tr, w, sim)
        (ta, wa, sa), (tb, wb, sb) = out[True], out[False]
        wb = wb if isinstance(wb, torch.Tensor) else torch.zeros(n, device=_dev()) + wb
        assert same(wa, wb) and same(ta.get_score(), tb.get_score()) and same(ta.get_retval(), tb.get_retval())
        ca, cb = dict(ta.get_choices().leaves()), dict(tb.get_choices().leaves())
        assert ca.keys() == cb.keys()
        for k in ca:
            assert same(ca[k], cb[k]), k
        assert same(sa.get_score(), sb.get_score()) and same(sa.get_retval(), sb.get_retval())
        for k, v in dict(sa.get_choices().leaves()).items():
            assert same(v, dict(sb.get_choices().leaves())[k]), k
        # the general ScanTrace methods work on the fused trace (per-step traces rebuilt on demand)
        lat = "x" if "x" in ta.get_choices() else "z"
        pa, pb = ta.project(keys, S[lat]), tb.project(keys, S[lat])
        assert torch.allclose(torch.as_tensor(pa), torch.as_tensor(pb), atol=1e-4)
        idx = torch.arange(0, n, 7, device=_dev())
        sub = ta.map_leaves(lambda v: v[idx] if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == n else v)
        assert same(sub.get_choices()[lat], ta.get_choices()[lat][idx]) and same(sub.get_score(), ta.get_score()[idx])
    # ImportanceK over a scan target: exact log-marginal of the linear-Gaussian chain (Kalman filter, float64)
    T = 5
    ys = torch.tensor([0.3, -0.2, 0.5, 0.1, -0.4])
    target = Target(lg_step.scan(n=T), (0.0, None), C["y"].set(ys))
    coll = ImportanceK(target, k_particles=200_000).run_smc(genjax.random.key(4, impl))
    m, p, logz = 0.0, 0.0, 0.0
    for y in ys.tolist():
        m, p = 0.9 * m, 0.81 * p + 1.0
        s = p + 0.25
        logz += _lpdf(y, m, math.sqrt(s))
        k = p / s
        m, p = m + k * (y - m), (1 - k) * p
    assert f(coll.get_log_marginal_likelihood_estimate()) == pytest.approx(logz, abs=0.02)
    assert isinstance(coll.get_particles(), CB.FusedScanTrace)
    part = coll.sample_particle(genjax.random.key(5, impl))
    assert tuple(part.get_choices()["x"].shape) == (T,)


# ---- vmap / repeat (tests/generative_functions/test_vmap_combinator.py:60-79) ---------------------------
def case_vmap(impl):
    @gen
    def point(x, s):
        y = normal(x, s) @ "y"
        return y

    vm = point.vmap(in_axes=(0, None))
    key = genjax.random.key(314159, impl)
    xs = torch.tensor([0.0, 1.0, 2.0])
    tr = vm.simulate(key, (xs, 0.1))
    assert tr.get_choices()["y"].shape == (3,) and tr.get_retval().shape == (3,)
    assert f(tr.get_score()) == pytest.approx(f(vm.assess(tr.get_choices(), (xs, 0.1))[0]), rel=1e-5)
    obs = torch.tensor([3.0, 2.0, 3.0])
    tr, w = vm.importance(key, C[:, "y"].set(obs), (xs, 1.0))
    want = sum(f(normal.logpdf(v, m, 1.0)) for v, m in zip(obs.tolist(), xs.tolist()))
    assert f(w) == pytest.approx(want, rel=1e-5)  # weight == sum of the three normal log-densities
    keys = genjax.random.split(key, 7)
    trb, wb = vm.importance(keys, C[:, "y"].set(obs), (xs, 1.0))
    assert wb.shape == (7,) and trb.get_choices()["y"].shape == (7, 3)
    assert torch.allclose(wb, torch.full_like(wb, want), rtol=1e-5)
    sim = vm.simulate(keys, (xs, 1.0))
    ys = sim.get_choices()["y"]
    assert ys.shape == (7, 3) and len(set(ys.flatten().tolist())) == 21  # independent draws per (particle, element)
    rep = point.repeat(n=4).simulate(keys, (0.0, 1.0))
    assert rep.get_choices()["y"].shape == (7, 4)

    @gen
    def model():
        mu = normal(0.0, 1.0) @ "mu"
        return point.vmap(in_axes=(0, None))(torch.zeros(3), 1.0) @ "ys"

    tr, w = model.importance(keys, C["ys", :, "y"].set(obs), ())
    assert w.shape == (7,) and tr.get_choices()["ys", "y"].shape == (7, 3)


# ---- fused bootstrap SMC ---------------------------------------------------------------------------------
def case_vmap_edge_cases(impl):
    """tests/generative_functions/test_vmap_combinator.py:158-243: assess == the simulated score, argument validation,
    a population of keys over a vmapped model, a zero-length mapped axis."""
    key = genjax.random.key(314159, impl)

    @genjax.vmap(in_axes=(0,))
    @gen
    def model(x):
        z = normal(x, 1.0) @ "z"
        return z

    xs = torch.arange(0, 50, dtype=torch.float32)
    tr = model.simulate(key, (xs,))
    assert f(model.assess(tr.get_choices(), (xs,))[0]) == pytest.approx(f(tr.get_score()), rel=1e-6)

    @gen
    def foo(loc, scale):
        return normal(loc, scale) @ "x"

    with pytest.raises(ValueError, match="vmap was requested to map its argument along axis 0, which implies that its rank "
                                         "should be at least 1, but is only 0"):
        foo.vmap(in_axes=(0, None)).simulate(key, (10.0, torch.arange(3.0)))
    with pytest.raises(IndexError):
        foo.vmap(in_axes=0).simulate(key, (torch.arange(2.0), torch.arange(3.0)))
    keys = genjax.random.split(key, 10)
    res = model.simulate(keys, (torch.arange(5, dtype=torch.float32),))
    assert tuple(res.get_score().shape) == (10,) and tuple(res.get_choices()[:, "z"].shape) == (10, 5)

    @gen
    def step(state, sigma):
        new_x = normal(state, sigma) @ "x"
        return (new_x, new_x + 1)

    empty = step.vmap(in_axes=(None, 0)).simulate(genjax.random.key(20, impl), (2.0, torch.arange(0, dtype=torch.float32)))
    assert empty.get_choices().static_is_empty() and f(empty.get_score()) == 0.0


def case_vmap_indexed_constraints(impl):
    """test_vmap_combinator.py:61-122: constraints on some / all indices of a vmapped site."""
    @genjax.vmap(in_axes=(0,))
    @gen
    def kernel(x):
        z = normal(x, 1.0) @ "z"
        return z

    key = genjax.random.key(314159, impl)
    map_over = torch.arange(0, 3, dtype=torch.float32)
    chm = C[jnp.arange(3), "z"].set(jnp.array([3.0, 2.0, 3.0]))  # one entry per index == the whole axis
    _, w = kernel.importance(key, chm, (map_over,))
    expect = sum(f(normal.assess(C.v(v), (m, 1.0))[0]) for v, m in ((3.0, 0.0), (2.0, 1.0), (3.0, 2.0)))
    assert f(w) == pytest.approx(expect, rel=1e-6)
    _, w_whole = kernel.importance(key, C[:, "z"].set(jnp.array([3.0, 2.0, 3.0])), (map_over,))
    assert f(w_whole) == pytest.approx(f(w), rel=1e-6)
    # a single index: only that element is constrained, the others are sampled (weight excludes them)
    key, sub_key = jax.random.split(key)
    tr, w = kernel.importance(sub_key, C[0, "z"].set(3.0), (map_over,))
    assert f(w) == pytest.approx(f(normal.assess(C.v(3.0), (0.0, 1.0))[0]), rel=1e-6)
    ch = tr.get_choices()
    assert f(ch[0, "z"]) == 3.0 and f(ch[1, "z"]) != 3.0
    assert f(tr.get_score()) == pytest.approx(sum(f(normal.logpdf(f(ch[i, "z"]), float(i), 1.0)) for i in range(3)), rel=1e-5)
    zv = jnp.array([3.0, -1.0, 2.0])
    tr, _ = kernel.importance(sub_key, C[jnp.arange(3), "z"].set(zv), (map_over,))
    for i in range(3):
        assert f(tr.get_choices()[i, "z"]) == f(zv[i])
    # the unconstrained elements draw what an unconstrained run draws (same keys, same folds)
    free = kernel.simulate(sub_key, (map_over,)).get_choices()
    tr, _ = kernel.importance(sub_key, C[1, "z"].set(0.5), (map_over,))
    assert f(tr.get_choices()[0, "z"]) == f(free[0, "z"]) and f(tr.get_choices()[2, "z"]) == f(free[2, "z"])
    # nested vmaps (test_vmap_combinator.py:108-122)
    @genjax.vmap(in_axes=(0,))
    @gen
    def higher_model(x):
        return kernel(x) @ "outer"

    _, w = higher_model.importance(key, C[0, "outer", 1, "z"].set(1.0), (torch.ones(3, 3),))
    assert f(w) == pytest.approx(f(normal.assess(C.v(1.0), (1.0, 1.0))[0]), rel=1e-6)
    # a population of particles over the same partially constrained site
    n = 512
    ptr, pw = kernel.importance(jax.random.split(key, n), C[2, "z"].set(0.25), (map_over,))
    assert pw.shape == (n,) and torch.allclose(pw, torch.full_like(pw, f(normal.logpdf(0.25, 2.0, 1.0))), atol=1e-6)
    pz = ptr.get_choices()[:, "z"] if False else ptr.inner.get_choices()["z"].reshape(n, 3)
    assert bool((pz[:, 2] == 0.25).all()) and float(pz[:, 0].std()) > 0.5
    # masks built by hand (choice_map.py mask): a flag column selects which particles are constrained
    flag = torch.arange(n) % 2 == 0
    tr, w = normal.importance(jax.random.split(key, n), C.v(torch.full((n,), 1.5)).mask(flag), (0.0, 1.0))
    lp = f(normal.logpdf(1.5, 0.0, 1.0))
    assert torch.equal(w != 0, flag.to(w.device)) and torch.allclose(w[::2], torch.full_like(w[::2], lp))
    assert bool((tr.get_retval()[::2] == 1.5).all()) and float(tr.get_retval()[1::2].std()) > 0.5


def case_batched_estimates(impl):
    """vmap of ImportanceK.log_marginal_likelihood_estimate over keys: several estimates per launch, each
    equal to the single call bit for bit; non-plan-able targets run key by key with the same results."""
    @gen
    def model(s):
        z = normal(0.0, 1.0) @ "z"
        p_ = beta(2.0, 2.0) @ "p"
        _ = flip(p_) @ "v"
        _ = normal(z, s) @ "y"

    target = Target(model, (0.5,), C["y"].set(0.3).at["v"].set(True))
    alg = ImportanceK(target, k_particles=4000)
    root = genjax.random.key(99, impl)
    keys = list(jax.random.split(root, 5))
    got = alg.log_marginal_likelihood_estimates(keys)
    assert got.shape == (5,)
    for b, key in enumerate(keys):
        assert f(got[b]) == f(alg.log_marginal_likelihood_estimate(key))
    exact = math.log(0.5) + f(normal.logpdf(0.3, 0.0, math.sqrt(1.25)))
    assert f(got.mean()) == pytest.approx(exact, abs=0.05)

    @gen
    def nested():  # a nested call is not plan-able: the same API, key by key
        z = normal(0.0, 1.0) @ "z"
        _ = model(1.0) @ "m"
        return z

    alg2 = ImportanceK(Target(nested, (), C["m", "y"].set(0.3)), k_particles=500)
    got2 = alg2.log_marginal_likelihood_estimates(keys[:3])
    for b in range(3):
        assert f(got2[b]) == f(alg2.log_marginal_likelihood_estimate(keys[b]))


# ---- GenSP estimators (smc.py:181-225, 432-465; sp.py:207-252) -----------------------------------------
def _gaussian_pair():
    @gen
    def model():
        z = normal(0.0, 1.0) @ "z"
        _ = normal(z, 0.5) @ "y"
        return z

    return model


def _lpdf(x, m, s):  # float64 log N(x; m, s)
    return -0.5 * ((x - m) / s) ** 2 - math.log(s) - 0.5 * math.log(2 * math.pi)


def case_gensp_estimators(impl):
    """estimate_normalizing_constant / estimate_reciprocal_normalizing_constant / run_csmc_for_normalizing_constant /
    estimate_logpdf on the conjugate pair z ~ N(0,1), y ~ N(z, 0.5): closed-form Z = N(y; 0, sqrt(1.25)), the
    unbiasedness identities of the estimators, and an independent float64 restatement of the reference's
    formulas (key order: split, then split(key, K-1); retained particle LAST) on explicitly derived particles."""
    model = _gaussian_pair()
    y0 = 0.7
    log_z = _lpdf(y0, 0.0, math.sqrt(1.25))
    target = Target(model, (), C["y"].set(y0))
    alg = ImportanceK(target, k_particles=50000)
    key = genjax.random.key(17, impl)
    # (1) estimate_normalizing_constant == ChangeTarget(alg, target).run_smc(second split).lml  (smc.py:204-212)
    z1 = f(alg.estimate_normalizing_constant(key, target))
    assert z1 == pytest.approx(log_z, abs=0.02)
    assert z1 == f(alg.log_marginal_likelihood_estimate(key, target))  # the same key path (smc.py:145-156)
    # an EQUAL but distinct target object takes the re-weighting pass: same estimate up to f32 rounding
    z1b = f(alg.estimate_normalizing_constant(key, Target(model, (), C["y"].set(y0))))
    assert z1b == pytest.approx(z1, abs=2e-5)
    # a DIFFERENT target re-weights by the likelihood ratio
    t2 = Target(model, (), C["y"].set(-0.4))
    assert f(alg.estimate_normalizing_constant(key, t2)) == pytest.approx(_lpdf(-0.4, 0.0, math.sqrt(1.25)), abs=0.03)
    # unbiasedness: E[exp(estimate)] = Z for ANY k (here k = 8, 400 keys)
    small = ImportanceK(target, k_particles=8)
    ks = genjax.random.split(genjax.random.key(3, impl), 400)
    ests = torch.stack([torch.as_tensor(small.estimate_normalizing_constant(ks[i], target)).reshape(()).double().cpu()
                        for i in range(400)])
    ratio = torch.exp(ests - log_z)
    assert f(ratio.mean()) == pytest.approx(1.0, abs=5 * f(ratio.std()) / 20.0)

    # (2) run_csmc + the reciprocal estimator.  Explicit derivation of the conditional collection:
    K = 64
    alg_k = ImportanceK(target, k_particles=K)
    z_star = 0.31
    retained = C["z"].set(z_star)
    kk = genjax.random.key(23, impl)
    k_a, k_b = genjax.random.split(kk)            # run_csmc_for_normalizing_constant: key, sub_key = split(key)
    coll = alg_k.run_csmc(k_b, retained)          # prev.run_csmc(sub_key, latent_choices)
    zs = coll.get_particles().get_choices()["z"].double().cpu()
    lw = coll.get_log_weights().double().cpu()
    assert len(coll) == K and f(zs[-1]) == pytest.approx(z_star, abs=1e-7)  # retained LAST
    # fresh particles are prior draws weighted by the likelihood; the retained one carries its full score (all
    # of its choices are constrained: the reference's weight, smc.py:333-346)
    for i in (0, 5, K - 2):
        assert f(lw[i]) == pytest.approx(_lpdf(y0, f(zs[i]), 0.5), abs=1e-5)
    assert f(lw[-1]) == pytest.approx(_lpdf(z_star, 0.0, 1.0) + _lpdf(y0, z_star, 0.5), abs=1e-5)
    # the first K-1 particles are those of split(second split of k_b, K-1): smc.py:318-319
    k_b1, k_b2 = genjax.random.split(k_b)
    fresh = target.importance(genjax.random.split(k_b2, K - 1), ChoiceMap.empty())[0].get_choices()["z"].double().cpu()
    assert torch.equal(fresh, zs[:-1])
    # float64 restatement of smc.py:432-465 for the SAME target (rejected weights re-scored against it) and w
    w_in = _lpdf(y0, z_star, 0.5)                 # properly weighted for the target under the prior proposal
    rej = torch.tensor([_lpdf(y0, f(zs[i]), 0.5) for i in range(K - 1)], dtype=torch.float64)
    retained_score = _lpdf(z_star, 0.0, 1.0) + _lpdf(y0, z_star, 0.5)
    tail = w_in - retained_score + f(lw[-1])
    want = retained_score - (f(torch.logsumexp(torch.cat([rej, torch.tensor([tail], dtype=torch.float64)]), 0)) - math.log(K))
    got = f(alg_k.estimate_reciprocal_normalizing_constant(kk, Target(model, (), C["y"].set(y0)), retained, w_in))
    assert got == pytest.approx(want, abs=3e-5)
    assert got == pytest.approx(f(ChangeTarget(alg_k, target).run_csmc_for_normalizing_constant(kk, retained, w_in)), abs=3e-5)
    # conditional-SMC identity: with z* ~ p(z | y) and the proper w, E[exp(estimate - log p(z*, y))] = 1 / Z
    post_m, post_s = y0 / 1.25, math.sqrt(0.2)
    g = torch.Generator().manual_seed(7)
    inv = []
    small = ImportanceK(target, k_particles=6)
    ks = genjax.random.split(genjax.random.key(29, impl), 400)
    for i in range(400):
        zi = post_m + post_s * float(torch.randn((), generator=g))
        joint = _lpdf(zi, 0.0, 1.0) + _lpdf(y0, zi, 0.5)
        e = f(small.estimate_reciprocal_normalizing_constant(ks[i], target, C["z"].set(zi), _lpdf(y0, zi, 0.5)))
        inv.append(math.exp(e - joint + log_z))   # = Z / Z_hat_csmc
    inv = torch.tensor(inv, dtype=torch.float64)
    assert f(inv.mean()) == pytest.approx(1.0, abs=5 * f(inv.std()) / 20.0)
    # k_particles = 1: the collection is the retained particle alone
    one = ImportanceK(target, k_particles=1)
    e1 = f(one.estimate_reciprocal_normalizing_constant(kk, target, retained, w_in))
    assert e1 == pytest.approx(retained_score - w_in, abs=2e-5)
    assert f(one.estimate_logpdf(kk, retained, target)) == pytest.approx(0.0, abs=2e-5)  # score - its own weight

    # (3) estimate_logpdf (smc.py:181-198): score of a particle drawn from run_csmc(key, v) minus that collection's
    # log-marginal estimate; restated on the explicitly built collection
    k1, k2 = genjax.random.split(kk)
    cs = ChangeTarget(alg_k, target).run_csmc(k1, retained)
    part = cs.sample_particle(k2)
    want_lp = f(part.get_score()) - f(cs.get_log_marginal_likelihood_estimate())
    assert f(alg_k.estimate_logpdf(kk, retained, target)) == pytest.approx(want_lp, abs=1e-6)
    zp = f(part.get_choices()["z"])
    assert f(part.get_score()) == pytest.approx(_lpdf(zp, 0.0, 1.0) + _lpdf(y0, zp, 0.5), abs=2e-5)


def case_marginal_with_algorithm(impl):
    """Marginal(selection, algorithm=ImportanceK(...)) (sp.py:207-252): `estimate_logpdf` is the algorithm's
    normalising-constant estimate of the target constrained to the value; `random_weighted` simulates the model and
    scores the selected choices with the reciprocal estimator.  Used as a proposal `q` it plugs into ImportanceK
    (custom_proposal.ipynb cell 22 shape)."""
    model = _gaussian_pair()
    K = 4000
    template = Target(model, (), C["y"].set(0.0))  # the algorithm's own target: same constrained addresses
    marg = genjax.marginal(S["y"], algorithm=ImportanceK(template, k_particles=K))(model)
    key = genjax.random.key(41, impl)
    # estimate_logpdf(y) ~ log p(y) = log N(y; 0, sqrt(1.25)); and it IS estimate_normalizing_constant on the target
    for yv in (0.3, -1.1):
        est = f(marg.estimate_logpdf(key, C["y"].set(yv)))
        assert est == pytest.approx(_lpdf(yv, 0.0, math.sqrt(1.25)), abs=0.06)
        again = f(ImportanceK(template, k_particles=K).estimate_normalizing_constant(key, Target(model, (), C["y"].set(yv))))
        assert est == again
    # random_weighted: key order of sp.py:222-236 — simulate with the 2nd output of the first split, project with the
    # 2nd output of the second split, the remaining key goes to the algorithm
    wgt, chm = marg.random_weighted(key)
    assert "y" in chm and "z" not in chm
    k_rest, k_sim = genjax.random.split(key)
    tr = model.simulate(k_sim, ())
    assert f(chm["y"]) == f(tr.get_choices()["y"])
    k_alg, k_proj = genjax.random.split(k_rest)
    z_sim, y_sim = f(tr.get_choices()["z"]), f(tr.get_choices()["y"])
    w_proj = f(tr.project(k_proj, ~S["y"]))
    assert w_proj == pytest.approx(_lpdf(z_sim, 0.0, 1.0), abs=2e-5)  # score of the unselected choice
    direct = f(ImportanceK(template, k_particles=K).estimate_reciprocal_normalizing_constant(
        k_alg, Target(model, (), C["y"].set(tr.get_choices()["y"])), C["z"].set(tr.get_choices()["z"]), tr.project(k_proj, ~S["y"])))
    assert f(wgt) == direct
    # the formula of smc.py:432-465 in float64 on the explicitly derived conditional collection (reference quirks
    # kept: the retained score / weight are those under the algorithm's OWN target, y = 0)
    alg = ImportanceK(template, k_particles=K)
    ka, kb = genjax.random.split(k_alg)
    coll = alg.run_csmc(kb, C["z"].set(tr.get_choices()["z"]))
    zs = coll.get_particles().get_choices()["z"].double().cpu()
    rej = torch.tensor([_lpdf(y_sim, float(v), 0.5) for v in zs[:-1]], dtype=torch.float64)  # re-scored for y = y_sim
    ret_score = _lpdf(z_sim, 0.0, 1.0) + _lpdf(0.0, z_sim, 0.5)
    tail = w_proj - ret_score + f(coll.get_log_weights()[-1])
    want = ret_score - (f(torch.logsumexp(torch.cat([rej, torch.tensor([tail], dtype=torch.float64)]), 0)) - math.log(K))
    assert f(wgt) == pytest.approx(want, abs=5e-5)
    # round trip: the value it proposed has an estimated log-density close to the analytic marginal
    rt = f(marg.estimate_logpdf(genjax.random.key(43, impl), chm))
    assert rt == pytest.approx(_lpdf(y_sim, 0.0, math.sqrt(1.25)), abs=0.06)
    # as a proposal q of ImportanceK over a target whose latent is "y" of the marginal's model
    @gen
    def outer():
        y = normal(0.0, math.sqrt(1.25)) @ "y"
        _ = normal(y, 1.0) @ "obs"

    @gen
    def pair_for(target):  # a proposal receives the target it proposes for (sp.py:217-238, notebook cell 20)
        z = normal(0.0, 1.0) @ "z"
        _ = normal(z, 0.5) @ "y"

    t_outer = Target(outer, (), C["obs"].set(0.5))
    template_q = Target(pair_for, (t_outer,), C["y"].set(0.0))
    small = genjax.marginal(S["y"], algorithm=ImportanceK(template_q, k_particles=32))(pair_for)
    coll = ImportanceK(t_outer, q=small, k_particles=64).run_smc(genjax.random.key(47, impl))
    ys, lw = coll.get_particles().get_choices()["y"], coll.get_log_weights()
    assert ys.shape == (64,) and lw.shape == (64,) and bool(torch.isfinite(lw).all())
    # weights = target score at the proposed choice - q's own (reciprocal-estimator) score of it, particle i using
    # the SAME sub-key for q and for the target (smc.py:299-305).  (With the reference's `project(~selection)` weight
    # the score q reports is not the marginal density — mirrored, not fixed: DESIGN.md §8 — so only the plumbing
    # is pinned here, not a posterior.)
    _, sub = genjax.random.split(genjax.random.key(47, impl))
    pks = genjax.random.split(sub, 64)
    for i in (0, 9, 63):
        wi, ci = small.random_weighted(pks[i], t_outer)
        yi = f(ci["y"])
        assert yi == f(ys[i])
        assert f(lw[i]) == pytest.approx(_lpdf(yi, 0.0, math.sqrt(1.25)) + _lpdf(0.5, yi, 1.0) - f(wi), abs=3e-5)



def case_bootstrap_smc(impl):
    from genjax._amd import workloads as W

    y = W.lgssm_data(30)
    smc = BootstrapSMC(LinearGaussianSSM(), y, n_particles=20000, record_ancestors=True)
    res = smc.run(genjax.random.key(3, impl))
    assert res.log_marginal_likelihood == pytest.approx(W.lgssm_exact_log_z(y), abs=0.4)  # estimator std ~0.07
    assert res.ancestors.shape == (30, 20000) and res.particles.shape == (20000,)
    a = res.ancestors[7]
    assert bool((a[1:] >= a[:-1]).all())
    res2 = smc.run(genjax.random.key(3, impl))
    assert torch.equal(res.step_q, res2.step_q)  # counter-based: runs are replayable
    # several filters in the same launches == the filters run one by one
    many = smc.run_many([genjax.random.key(s_, impl) for s_ in (3, 4, 5)])
    assert torch.equal(many[0].step_q, res.step_q) and torch.equal(many[0].ancestors, res.ancestors)
    assert torch.equal(many[0].particles, res.particles)
    one = smc.run(genjax.random.key(5, impl))
    assert torch.equal(many[2].step_q, one.step_q) and many[2].log_marginal_likelihood == one.log_marginal_likelihood
    assert torch.equal(many[2].log_weights, one.log_weights)


def case_general_smc(impl):
    """A user-written state-space model lowered to the fused SMC kernels (smc_plan.py)."""
    from genjax._amd import workloads as W

    @gen
    def init():
        x = normal(0.0, 1.0) @ "x"
        normal(x, 0.5) @ "y"
        return x

    @gen
    def step(x):
        x2 = normal(0.9 * x, 1.0) @ "x"
        normal(x2, 0.5) @ "y"
        return x2

    y = W.lgssm_data(25)
    key = genjax.random.key(3, impl)
    a = BootstrapSMC(StateSpaceModel(init, step), C["y"].set(torch.tensor(y)), 8192, record_ancestors=True).run(key)
    b = BootstrapSMC(LinearGaussianSSM(), y, 8192, record_ancestors=True).run(key)
    # the generated kernel and the hand-written LGSSM kernel are the same filter, bit for bit (threefry: per-slot
    # keys; philox: both draw word (slot & 3) of the quad's block 0 and pair Box-Muller inside the quad)
    assert torch.equal(a.step_q, b.step_q) and torch.equal(a.step_e, b.step_e)
    assert torch.equal(a.particles, b.particles) and torch.equal(a.ancestors, b.ancestors)
    assert a.log_marginal_likelihood == pytest.approx(W.lgssm_exact_log_z(y), abs=0.5)
    assert b.log_marginal_likelihood == pytest.approx(W.lgssm_exact_log_z(y), abs=0.5)

    @gen
    def init2():
        m = normal(0.0, 1.0) @ "m"
        g = gamma(2.0, 2.0) @ "g"
        normal(m, 0.7) @ "y"
        return m, g

    @gen
    def step2(c):
        m, g = c
        m2 = normal(0.8 * m, 0.5) @ "m"
        g2 = gamma(2.0, g + 1.0) @ "g"
        normal(m2, 0.7) @ "y"
        return m2, g2

    r = BootstrapSMC(StateSpaceModel(init2, step2), C["y"].set(torch.tensor(y)), 4096).run(key)
    assert len(r.particles) == 2 and r.particles[0].shape == (4096,) and bool((r.particles[1] > 0).all())
    assert math.isfinite(r.log_marginal_likelihood)
    # the m-chain is linear-Gaussian and independent of g: its exact evidence is a Kalman filter
    mm, pp, ll = 0.0, 1.0, 0.0
    for t, yt in enumerate(y.astype("float64")):
        if t:
            mm, pp = 0.8 * mm, 0.64 * pp + 0.25
        sv = pp + 0.49
        ll += -0.5 * (yt - mm) ** 2 / sv - 0.5 * math.log(2 * math.pi * sv)
        kk = pp / sv
        mm, pp = mm + kk * (yt - mm), (1 - kk) * pp
    assert r.log_marginal_likelihood == pytest.approx(ll, abs=0.6)
    # a COUPLED two-component kernel: arguments over several traced values (postfix programs, gjx.h GJX_ARG_EXPR) —
    # position / velocity, the observation sees the position; exact evidence by a 2-d Kalman filter (float64)
    @gen
    def init3():
        p = normal(0.0, 1.0) @ "p"
        v = normal(0.0, 0.5) @ "v"
        normal(p, 0.6) @ "y"
        return p, v

    @gen
    def step3(c):
        p, v = c
        v2 = normal(0.9 * v - 0.1 * p, 0.3) @ "v"
        p2 = normal(p + 0.5 * v2, 0.2) @ "p"
        normal(p2, 0.6) @ "y"
        return p2, v2

    import numpy as np

    r3 = BootstrapSMC(StateSpaceModel(init3, step3), C["y"].set(torch.tensor(y)), 65536).run(key)
    mu, P_ = np.zeros(2), np.diag([1.0, 0.25])
    # x = (p, v);  v' = 0.9 v - 0.1 p + e_v (0.3);  p' = p + 0.5 v' + e_p (0.2)
    A_ = np.array([[1.0 - 0.05, 0.45], [-0.1, 0.9]])
    G_ = np.array([[0.2, 0.5 * 0.3], [0.0, 0.3]])  # noise loading of (e_p, e_v)
    Q_ = G_ @ G_.T
    H_, R_ = np.array([[1.0, 0.0]]), 0.36
    ll3 = 0.0
    for t, yt in enumerate(y.astype("float64")):
        if t:
            mu, P_ = A_ @ mu, A_ @ P_ @ A_.T + Q_
        sv = (H_ @ P_ @ H_.T).item() + R_
        ll3 += -0.5 * (yt - mu[0]) ** 2 / sv - 0.5 * math.log(2 * math.pi * sv)
        kk = (P_ @ H_.T) / sv
        mu, P_ = mu + kk[:, 0] * (yt - mu[0]), P_ - kk @ H_ @ P_
    assert r3.log_marginal_likelihood == pytest.approx(ll3, abs=0.25)

    # constant-velocity tracking: the position is a DETERMINISTIC function of the carried state and the sampled velocity (the
    # carry component is an expression); exact evidence by the same Kalman filter with a singular process noise
    @gen
    def init4():
        p = normal(0.0, 1.0) @ "p"
        v = normal(0.0, 0.5) @ "v"
        normal(p, 0.6) @ "y"
        return p, v

    @gen
    def step4(c):
        p, v = c
        v2 = normal(0.95 * v, 0.3) @ "v"
        normal(p + 0.5 * v2, 0.6) @ "y"
        return p + 0.5 * v2, v2

    r4 = BootstrapSMC(StateSpaceModel(init4, step4), C["y"].set(torch.tensor(y)), 65536).run(key)
    mu, P_ = np.zeros(2), np.diag([1.0, 0.25])
    A4 = np.array([[1.0, 0.5 * 0.95], [0.0, 0.95]])
    G4 = np.array([[0.5 * 0.3], [0.3]])
    Q4 = G4 @ G4.T
    ll4 = 0.0
    for t, yt in enumerate(y.astype("float64")):
        if t:
            mu, P_ = A4 @ mu, A4 @ P_ @ A4.T + Q4
        sv = (H_ @ P_ @ H_.T).item() + R_
        ll4 += -0.5 * (yt - mu[0]) ** 2 / sv - 0.5 * math.log(2 * math.pi * sv)
        kk = (P_ @ H_.T) / sv
        mu, P_ = mu + kk[:, 0] * (yt - mu[0]), P_ - kk @ H_ @ P_
    assert r4.log_marginal_likelihood == pytest.approx(ll4, abs=0.3)
    # vmap over keys: the filters of a user model step in the same launches, each equal to its own run
    smc2 = BootstrapSMC(StateSpaceModel(init2, step2), C["y"].set(torch.tensor(y)), 4096, record_ancestors=True)
    ks = [genjax.random.key(s_, impl) for s_ in (3, 8, 9)]
    many = smc2.run_many(ks)
    assert many[0].log_marginal_likelihood == r.log_marginal_likelihood
    for k_, m_ in zip(ks, many):
        one = smc2.run(k_)
        assert torch.equal(m_.step_q, one.step_q) and torch.equal(m_.ancestors, one.ancestors)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(m_.particles, one.particles))
    with pytest.raises(ValueError):
        BootstrapSMC(StateSpaceModel(init, step), C["nope"].set(torch.tensor(y)), 1024).run(key)
    # run_many's contract when the library refuses a filter batch (population too large for one: GJX_ERR_UNSUPPORTED,
    # or a workspace the device cannot hold): element b is still self.run(keys[b])
    from genjax._amd import abi as _abi

    calls = []

    def refuse(ops_, chunk, T_, ess_):
        calls.append(len(chunk))
        raise _abi.GjxError("gjx_smc_run_plan", -2)

    orig = smc2._run_chunk
    smc2._run_chunk = refuse
    try:
        fb = smc2.run_many(ks)
    finally:
        smc2._run_chunk = orig
    assert calls == [3]
    for a_, b_ in zip(fb, many):
        assert a_.log_marginal_likelihood == b_.log_marginal_likelihood and torch.equal(a_.ancestors, b_.ancestors)

    def broken(ops_, chunk, T_, ess_):
        raise _abi.GjxError("gjx_smc_run_plan", -4)  # anything else is an error, not a reason to fall back

    smc2._run_chunk = broken
    try:
        with pytest.raises(_abi.GjxError):
            smc2.run_many(ks)
    finally:
        smc2._run_chunk = orig


# ---- Regenerate / StaticRequest / Rejuvenate (requests.py:46-66; static.py:505-715; rejuvenate.py:45-94) -----------------
def case_regenerate_and_rejuvenate(impl):
    """Mirrors tests/inference/test_requests.py:37-196: a selected site takes a fresh draw and the weight is the change of
    the target density; the backward request restores the old trace with the opposite weight; Metropolis-Hastings with
    either request converges on a sharply observed value — run here over a population of independent chains."""
    from genjax import Regenerate, StaticRequest
    from genjax.inference.requests import Rejuvenate

    @gen
    def simple_normal():
        y1 = normal(0.0, 1.0) @ "y1"
        y2 = normal(0.0, 1.0) @ "y2"
        return y1 + y2

    key = genjax.random.key(314159, impl)
    key, sub_key = genjax.random.split(key)
    tr = simple_normal.simulate(sub_key, ())
    for addr, sel in (("y1", S["y1"]), ("y2", S["y2"])):
        old_v = tr.get_choices()[addr]
        new_tr, fwd_w, _, bwd = Regenerate(sel).edit(key, tr, ())
        new_v = new_tr.get_choices()[addr]
        assert f(old_v) != f(new_v) and f(fwd_w) != 0.0
        assert f(fwd_w) == pytest.approx(f(normal.logpdf(new_v, 0.0, 1.0)) - f(normal.logpdf(old_v, 0.0, 1.0)), abs=1e-6)
        other = "y2" if addr == "y1" else "y1"
        assert f(new_tr.get_choices()[other]) == f(tr.get_choices()[other])
        old_tr, bwd_w, _, _ = bwd.edit(sub_key, new_tr, ())
        assert f(fwd_w) + f(bwd_w) == pytest.approx(0.0, abs=1e-6)
        assert f(old_tr.get_choices()[addr]) == f(old_v)
    new_tr, fwd_w, _, bwd = Regenerate(S["y1"] | S["y2"]).edit(key, tr, ())
    assert f(new_tr.get_choices()["y2"]) != f(tr.get_choices()["y2"])
    old_tr, bwd_w, _, _ = bwd.edit(key, new_tr, ())
    assert f(fwd_w) + f(bwd_w) == pytest.approx(0.0, abs=1e-6) and f(old_tr.get_choices()["y2"]) == f(tr.get_choices()["y2"])

    # StaticRequest composition, tuple addresses and nested functions (test_static_gen_fn.py:889-961)
    @gen
    def submodel():
        return normal(0.0, 1.0) @ "y2"

    @gen
    def nested():
        y1 = normal(0.0, 1.0) @ ("y1", "y3")
        y2 = submodel() @ "y2"
        return y1 + y2

    for mdl, req, addr in ((simple_normal, StaticRequest({"y1": Regenerate(genjax.Selection.all()), "y2": genjax.Update(C.v(3.0))}), "y2"),
                           (nested, StaticRequest({("y1", "y3"): Regenerate(S.all),
                                                   "y2": StaticRequest({"y2": genjax.Update(C.v(3.0))})}), ("y2", "y2"))):
        t0 = mdl.simulate(genjax.random.key(0, impl), ())
        k_a, k_b = genjax.random.split(genjax.random.key(0, impl))
        new_t, w, _, bwd_req = req.edit(k_a, t0, ())
        assert f(new_t.get_choices()[addr]) == 3.0 and f(w) != 0.0
        old_t, w_, _, _ = bwd_req.edit(k_b, new_t, ())
        assert f(old_t.get_choices()[addr]) == f(t0.get_choices()[addr]) and f(w_) != 0.0
        assert f(w) + f(w_) == pytest.approx(0.0, abs=1e-6)

    @gen
    def linked_normal(s):
        y1 = normal(0.0, 3.0) @ "y1"
        _ = normal(y1, s) @ "y2"

    tr = linked_normal.simulate(sub_key, (1.0,))
    dens = lambda t: f(normal.logpdf(t.get_choices()["y1"], 0.0, 3.0)) + f(normal.logpdf(t.get_choices()["y2"], t.get_choices()["y1"], 1.0))
    new_tr, fwd_w, _, _ = Regenerate(S["y1"]).edit(key, tr, (1.0,))
    assert f(fwd_w) == pytest.approx(dens(new_tr) - dens(tr), abs=1e-5) and f(fwd_w) != 0.0

    # Metropolis-Hastings over a population of chains: y2 observed at 3.0 with a sharp likelihood
    n = 128

    def mh(request, sd, steps, k0):
        keys = genjax.random.split(genjax.random.key(k0, impl), n)
        t, _ = linked_normal.importance(keys, C.kw(y2=3.0), (sd,))
        for i in range(steps):
            ka = genjax.random.split(genjax.random.key(1000 * k0 + 2 * i, impl), n)
            kb = genjax.random.split(genjax.random.key(1000 * k0 + 2 * i + 1, impl), n)
            new_t, w, _, _ = request.edit(ka, t, (sd,))
            accept = torch.log(beta.sample(kb, 1.0, 1.0)) < w
            # accepted chains move (the per-particle masked update of 8f-3), the others keep their trace
            t, _, _, _ = t.update(ka, C["y1"].set(new_t.get_choices()["y1"]).mask(accept))
        return t.get_choices()["y1"]

    y1 = mh(Regenerate(S["y1"]), 0.01, 120, 7)
    assert f((y1 - 3.0).abs().median()) < 0.05
    # Rejuvenate with the prior as the proposal: the move is symmetric, its weight vanishes
    @gen
    def one_normal():
        _ = normal(0.0, 1.0) @ "y1"

    tr1 = one_normal.simulate(sub_key, ())
    req = StaticRequest({"y1": Rejuvenate(normal, lambda chm: (0.0, 1.0))})
    new_tr, w, _, _ = req.edit(sub_key, tr1, ())
    assert f(new_tr.get_choices()["y1"]) != f(tr1.get_choices()["y1"]) and abs(f(w)) < 1e-6
    # ... and a random-walk proposal around the current value converges
    walk = StaticRequest({"y1": Rejuvenate(normal, lambda chm: (chm.get_value(), 0.3))})
    y1 = mh(walk, 0.001, 60, 9)
    assert f((y1 - 3.0).abs().median()) < 0.02
    # a combinator answers Regenerate element-wise: weight = change of the total score, unselected choices stay
    @gen
    def kstep(x, _):
        z = normal(x, 1.0) @ "z"
        u = normal(z, 0.5) @ "u"
        return z, u

    keys = genjax.random.split(genjax.random.key(12, impl), 500)
    st = kstep.scan(n=4).simulate(keys, (0.0, None))
    keys2 = genjax.random.split(genjax.random.key(13, impl), 500)  # (the same keys would redraw the same values)
    new_st, w, _, bwd = Regenerate(S["u"]).edit(keys2, st, (0.0, None))
    assert torch.equal(new_st.get_choices()["z"], st.get_choices()["z"])
    assert not torch.equal(new_st.get_choices()["u"], st.get_choices()["u"])
    assert torch.allclose(w, new_st.get_score() - st.get_score(), atol=1e-5)
    back, wb, _, _ = bwd.edit(keys, new_st, (0.0, None))
    assert torch.equal(back.get_choices()["u"], st.get_choices()["u"]) and torch.allclose(w + wb, torch.zeros_like(w), atol=1e-4)


def case_vector_valued_sites(impl):
    """A distribution site whose arguments carry an event axis (`normal(mu_vec, 1.0) @ "x"`): values `[d]` (`[n, d]` over a
    population), the score summed over the axis (distribution.py:392-396).  simulate / importance / assess / update, a scalar
    key and a population; means of the draws; flip over a probability vector."""
    @gen
    def model(mu):
        x = normal(mu, 1.0) @ "x"
        _ = normal(x.sum(-1), 0.5) @ "y"
        return x

    mu = torch.tensor([0.0, 1.0, -2.0]).to(_dev())
    key = genjax.random.key(3, impl)
    tr = model.simulate(key, (mu,))
    x = tr.get_choices()["x"]
    assert tuple(x.shape) == (3,) and tuple(tr.get_retval().shape) == (3,)
    want = sum(_lpdf(f(x[i]), f(mu[i]), 1.0) for i in range(3)) + _lpdf(f(tr.get_choices()["y"]), f(x.sum()), 0.5)
    assert f(tr.get_score()) == pytest.approx(want, abs=1e-5)
    # (keyless `assess` reads 1-D columns as a population; the explicit `vmap` form carries the event structure)
    @gen
    def model_v(mu):
        x = normal.vmap(in_axes=(0, None))(mu, 1.0) @ "x"
        _ = normal(x.sum(-1), 0.5) @ "y"
        return x

    s, _ = model_v.assess(tr.get_choices(), (mu,))
    assert f(s) == pytest.approx(want, abs=1e-5)
    cx = torch.tensor([0.5, 0.5, 0.5])
    tr2, w = model.importance(key, C["x"].set(cx) | C["y"].set(1.0), (mu,))
    assert torch.equal(tr2.get_choices()["x"].cpu(), cx)
    assert f(w) == pytest.approx(sum(_lpdf(0.5, f(mu[i]), 1.0) for i in range(3)) + _lpdf(1.0, 1.5, 0.5), abs=1e-5)
    tr3, wu, _, disc = tr.update(key, C["x"].set(torch.tensor([1.0, 1.0, 1.0])))
    assert tuple(tr3.get_choices()["x"].shape) == (3,) and f(wu) == pytest.approx(f(tr3.get_score()) - f(tr.get_score()), abs=1e-4)
    keys = genjax.random.split(key, 4000)
    trp, wp = model.importance(keys, C["y"].set(-1.0), (mu,))
    xs = trp.get_choices()["x"]
    assert tuple(xs.shape) == (4000, 3) and tuple(wp.shape) == (4000,) and tuple(trp.get_score().shape) == (4000,)
    assert torch.allclose(xs.mean(0).cpu(), mu.cpu(), atol=0.08)
    assert torch.allclose(wp.cpu(), normal.logpdf(-1.0, xs.sum(-1), 0.5).cpu(), atol=1e-5)

    @gen
    def coins():
        b = flip(torch.tensor([0.1, 0.9, 0.5, 0.5])) @ "b"
        return b

    t = coins.simulate(keys, ())
    assert torch.allclose(t.get_choices()["b"].float().mean(0).cpu(), torch.tensor([0.1, 0.9, 0.5, 0.5]), atol=0.04)


def case_index_request(impl):
    """`IndexRequest(idx, request)` on `Scan` and `Vmap` traces inside a static model (test_scan_combinator.py:463-534,
    test_vmap_combinator.py:273-330): the weight is the change of density of the touched element; the others keep their
    values; out-of-range indices are refused; the backward request restores the trace."""
    from genjax import IndexRequest, Regenerate, StaticRequest, Update

    @gen
    def kernel(carry, _):
        z = normal(0.0, 1.0) @ "z"
        return z, None

    @gen
    def scanned_normal():
        y1 = normal(0.0, 1.0) @ "y1"
        _ = normal(0.0, 1.0) @ "y2"
        return kernel.scan(n=10)(y1, None) @ "kernel"

    key = genjax.random.key(314159, impl)
    key, sub_key = genjax.random.split(key)
    tr = scanned_normal.simulate(sub_key, ())
    new_tr, w, _, _ = Regenerate(S["y1"]).edit(key, tr, ())
    assert f(w) == pytest.approx(f(normal.logpdf(new_tr.get_choices()["y1"], 0.0, 1.0)) - f(normal.logpdf(tr.get_choices()["y1"], 0.0, 1.0)), abs=1e-5)
    for idx in range(10):
        old_z = tr.get_choices()["kernel", idx, "z"]
        req = StaticRequest({"kernel": IndexRequest(torch.tensor(idx), Regenerate(S["z"]))})
        new_tr, w, _, bwd = req.edit(key, tr, ())
        new_z = new_tr.get_choices()["kernel", idx, "z"]
        assert f(new_z) != f(old_z)
        assert f(w) == pytest.approx(f(normal.logpdf(new_z, 0.0, 1.0)) - f(normal.logpdf(old_z, 0.0, 1.0)), abs=2e-5)
        other = (idx + 1) % 10
        assert f(new_tr.get_choices()["kernel", other, "z"]) == f(tr.get_choices()["kernel", other, "z"])
        back, wb, _, _ = bwd.edit(key, new_tr, ())
        assert f(back.get_choices()["kernel", idx, "z"]) == f(old_z) and f(w) + f(wb) == pytest.approx(0.0, abs=2e-5)
    with pytest.raises(IndexError):
        StaticRequest({"kernel": IndexRequest(11, Regenerate(S["z"]))}).edit(key, tr, ())

    @gen
    def model():
        x = normal(0.0, 1.0) @ "x"
        _ = normal.vmap()(torch.zeros(100), torch.ones(100)) @ "a"
        return x

    tr = model.simulate(sub_key, ())
    for idx in range(0, 100, 13):
        old_a = tr.get_choices()["a", idx]
        new_tr, w, _, _ = StaticRequest({"a": IndexRequest(idx, Regenerate(S.all))}).edit(key, tr, ())
        new_a = new_tr.get_choices()["a", idx]
        assert f(w) == pytest.approx(f(normal.logpdf(new_a, 0.0, 1.0)) - f(normal.logpdf(old_a, 0.0, 1.0)), abs=5e-5)
        new_tr, w, _, _ = StaticRequest({"a": IndexRequest(idx, Update(C.v(idx + 7.0)))}).edit(key, tr, ())
        assert f(new_tr.get_choices()["a", idx]) == idx + 7.0
        assert f(w) == pytest.approx(f(normal.logpdf(idx + 7.0, 0.0, 1.0)) - f(normal.logpdf(old_a, 0.0, 1.0)), rel=1e-5, abs=5e-5)
        keep = (idx + 1) % 100
        assert f(new_tr.get_choices()["a", keep]) == f(tr.get_choices()["a", keep])


def case_fast_estimate_path(impl):
    """`ImportanceK(...).log_marginal_likelihood_estimate(key)` on a plan-able target keeps its traced body, plan and one
    set of persistent buffers on the algorithm object (inference.py: _fast_estimate): the same kernels on the same keys as
    the general route through `run_smc` — equal bit for bit, call after call, for several models and both generators; an
    in-place change of a tensor argument is seen; targets that are not plan-able, proposals and `target=` take the
    general route."""
    import torch

    from genjax._amd import inference as I

    @gen
    def gauss(mu0):
        zs = []
        for i in range(4):
            z = normal(mu0, 1.0) @ f"z{i}"
            _ = normal(z * 0.5 + 0.1, 0.5) @ f"y{i}"
            zs.append(z)
        return zs[0]

    @gen
    def mixed(a):
        p = beta(2.0, a) @ "p"
        v = flip(p) @ "v"
        g = gamma(2.0, 1.5) @ "g"
        _ = normal(p * 2.0 + g, 1.5) @ "x"

    t1 = Target(gauss, (0.3,), C["y0"].set(0.2) | C["y1"].set(-0.4) | C["y2"].set(1.1) | C["y3"].set(0.0))
    t2 = Target(mixed, (3.0,), C["x"].set(0.3) | C["v"].set(True))
    for t, k in ((t1, 3000), (t2, 2500), (t1, 2), (t1, 66_000)):  # (66_000: 258 rows, a fold over several workgroups)
        alg = ImportanceK(t, k_particles=k)
        for rep in range(3):
            key = genjax.random.key(50 + rep, impl)
            a = alg.log_marginal_likelihood_estimate(key)
            b = I.SMCAlgorithm.log_marginal_likelihood_estimate(alg, key)
            assert torch.equal(a, b), (k, rep)
            if rep == 0:
                first = a.clone()
        assert any(v is not None for v in alg.__dict__.get("_fast", {}).values())
        assert torch.equal(first, I.SMCAlgorithm.log_marginal_likelihood_estimate(alg, genjax.random.key(50, impl)))  # results are not views of the buffers
    # a tensor argument changed in place is seen (the cached trace is keyed by its version)
    mu = torch.tensor(0.3, device=first.device)
    alg = ImportanceK(Target(gauss, (mu,), t1.constraint), k_particles=2000)
    key = genjax.random.key(7, impl)
    a0 = alg.log_marginal_likelihood_estimate(key)
    mu.add_(1.5)
    a1 = alg.log_marginal_likelihood_estimate(key)
    assert not torch.equal(a0, a1) and torch.equal(a1, I.SMCAlgorithm.log_marginal_likelihood_estimate(alg, key))
    # the general route: a proposal, another target
    @gen
    def prop(target):
        for i in range(4):
            _ = normal(0.0, 1.5) @ f"z{i}"

    alg = ImportanceK(t1, q=prop.marginal() if hasattr(prop, "marginal") else None, k_particles=500)
    z = alg.log_marginal_likelihood_estimate(genjax.random.key(9, impl))
    assert not any(v is not None for v in (alg.__dict__.get("_fast") or {}).values()) and math.isfinite(f(z))
    z = ImportanceK(t1, k_particles=500).log_marginal_likelihood_estimate(genjax.random.key(9, impl), t1)
    assert math.isfinite(f(z))


def case_nested_calls(impl):
    """Nested `@gen` calls inside a fused plan (static.py:175-193, 349-352, 374-380): `callee(args) @ "addr"` takes ONE counter
    of its caller, runs under fold_in(key, counter) and numbers its own sites afresh; its weight and score are ITS totals,
    added to the caller's when it returns.  The body — callees in line, two levels deep, one of them without sites, one
    called three times — is ONE kernel (gjx_plan_create_scoped), and the trace it gives (hierarchical choices, nested
    scores and return values, the weight) equals the per-site path's bit for bit; the estimator built on it likewise."""
    import torch

    from genjax._amd import plan as P
    from genjax._amd.lang import GenerateHandler

    @gen
    def noise(scale):
        a = normal(0.0, scale) @ "a"
        b = gamma(2.0, 1.5) @ "b"
        return a * b

    @gen
    def inner(mu):
        e = noise(0.5) @ "n"
        x = normal(mu + e, 1.0) @ "x"
        y = normal(x * 0.5, 0.8) @ "y"
        return x + y

    @gen
    def empty(v):
        return v * 2.0

    @gen
    def model(t):
        z = normal(t, 1.0) @ "z"
        y1 = inner(z) @ "i1"
        k = flip(0.4) @ "k"
        w = empty(y1) @ "e"
        y2 = inner(w + 1.0) @ "i2"
        u = noise(2.0) @ "u"
        _ = normal(y1 + y2 + u, 0.7) @ "obs"
        return y2, k

    n = 3000
    keys = genjax.random.split(genjax.random.key(3, impl), n)
    col = torch.linspace(-1, 1, n).to(_dev())
    for chm in (C.n(),
                C["obs"].set(0.3) | C["i1", "x"].set(0.2) | C["i1", "y"].set(-0.1) | C["i1", "n", "a"].set(0.05) | C["i2", "n", "b"].set(1.1),
                C["z"].set(0.1) | C["u", "a"].set(-0.4) | C["i2", "y"].set(col)):
        fused = try_fused_generate(model, keys, chm, (0.25,))
        assert fused is not None, "nested @gen calls must lower to the fused kernel"
        ftr, fw = fused
        h = GenerateHandler(keys, chm)
        retval = h.run(model.source, (0.25,))
        eager = StaticTrace(model, (0.25,), retval, h.traces)
        if isinstance(h.weight, torch.Tensor):
            assert torch.equal(fw, h.weight)
        assert torch.equal(ftr.get_score(), eager.get_score())
        fc, ec = dict(ftr.get_choices().leaves()), dict(eager.get_choices().leaves())
        assert fc.keys() == ec.keys() and ("i2", "n", "b") in fc
        for key_ in fc:
            a_, b_ = torch.as_tensor(fc[key_]), torch.as_tensor(ec[key_])
            if a_.dim():
                assert torch.equal(a_, b_.to(a_.dtype).expand_as(a_)), key_
        assert torch.equal(ftr.get_retval()[0], eager.get_retval()[0]) and torch.equal(ftr.get_retval()[1], eager.get_retval()[1])
        for addr in ("i1", "i2", "u", "e"):
            fs, es = ftr.get_subtrace(addr), eager.get_subtrace(addr)
            assert torch.equal(torch.as_tensor(fs.get_score()).to(torch.float32).expand(n), torch.as_tensor(es.get_score()).to(torch.float32).expand(n)), addr
            assert torch.equal(torch.as_tensor(fs.get_retval()), torch.as_tensor(es.get_retval())), addr
        assert torch.equal(ftr.get_subtrace("i2").get_subtrace("n").get_score(), eager.get_subtrace("i2").get_subtrace("n").get_score())
    # the estimator on a target with nested calls takes the one-launch route and equals the general one
    alg = ImportanceK(Target(model, (0.25,), C["obs"].set(0.3) | C["i1", "y"].set(-0.1)), k_particles=2048)
    key = genjax.random.key(9, impl)
    from genjax._amd import inference as I
    assert torch.equal(alg.log_marginal_likelihood_estimate(key), I.SMCAlgorithm.log_marginal_likelihood_estimate(alg, key))
    assert any(v is not None for v in alg.__dict__.get("_fast", {}).values())
    # an address used twice inside ONE callee body is still an error; the same address in two different calls is not
    @gen
    def twice():
        normal(0.0, 1.0) @ "a"
        return normal(0.0, 1.0) @ "a"

    @gen
    def outer():
        return twice() @ "t"

    try:
        outer.importance(keys, C.n(), ())
        raise AssertionError("address reuse inside a callee went unnoticed")
    except Exception as e:  # noqa: BLE001
        assert "a" in str(e) or type(e).__name__ == "AddressReuse"
    tr = P._traced(model, C.n(), n, (0.25,))
    assert tr is not None and len(tr[0].scopes) == 6
    # a scan whose step kernel is composed of sub-models (a transition and an emission `@gen` function): ONE launch for all T
    # steps, equal to the host loop of per-site launches; constraints inside the callees; per-step traces rebuilt on demand
    from genjax._amd import combinators as CB

    @gen
    def transition(x, u):
        e = normal(0.0, 0.3) @ "eps"
        g = gamma(2.0, 2.0) @ "g"
        return x * 0.9 + u + e * g

    @gen
    def emission(x):
        return normal(x, 0.5) @ "y"

    @gen
    def step(carry, u):
        x, v = carry
        x2 = transition(x, u) @ "tr"
        k = flip(0.3) @ "k"
        y = emission(x2 / 2.0) @ "em"
        return (x2, v * 0.5 + torch.exp(x2 * 0.1)), (y, k)

    T, ns = 5, 1200
    us = torch.linspace(0.2, 1.4, T).to(_dev())
    skeys = genjax.random.split(genjax.random.key(5, impl), ns)
    for chm in (C.n(), C["em", "y"].set(torch.linspace(-1, 1, T)) | C["tr", "g"].set(torch.linspace(0.5, 1.5, T))):
        out = {}
        for fused in (True, False):
            CB.FUSED_SCAN = fused
            try:
                out[fused] = step.scan().generate(skeys, chm, ((0.25, -0.5), us))
            finally:
                CB.FUSED_SCAN = True
        (ta, wa), (tb, wb) = out[True], out[False]
        assert isinstance(ta, CB.FusedScanTrace), "a step kernel with nested calls must lower to the one-launch scan"
        if isinstance(wb, torch.Tensor):
            assert torch.equal(wa, wb)
        assert torch.equal(ta.get_score(), tb.get_score())
        ca, cb = dict(ta.get_choices().leaves()), dict(tb.get_choices().leaves())
        assert ca.keys() == cb.keys() and ("tr", "eps") in ca
        for key_ in ca:
            a_, b_ = torch.as_tensor(ca[key_]), torch.as_tensor(cb[key_])
            assert torch.equal(a_.float().cpu(), b_.float().cpu().expand_as(a_)), key_
        assert torch.equal(ta.get_retval()[0][0].cpu(), tb.get_retval()[0][0].cpu())
        assert torch.equal(torch.as_tensor(ta.get_retval()[1][0]).cpu().expand(ns, T), torch.as_tensor(tb.get_retval()[1][0]).cpu().expand(ns, T))
        assert torch.equal(ta.step_traces[2].get_score().cpu(), tb.step_traces[2].get_score().cpu())
    # a bootstrap filter whose init / step are composed of sub-models: generated kernels with per-scope keys
    # (gjx_smc_plan_create_scoped); the same LGSSM as the hand-written filter, so the evidence is Kalman's
    from genjax._amd import workloads as W

    @gen
    def prior():
        return normal(0.0, 1.0) @ "x"

    @gen
    def emit(x):
        return normal(x, 0.5) @ "y"

    @gen
    def trans(x):
        return normal(0.9 * x, 1.0) @ "x"

    @gen
    def f_init():
        x = prior() @ "p"
        emit(x) @ "e"
        return x

    @gen
    def f_step(x):
        x2 = trans(x) @ "t"
        emit(x2) @ "e"
        return x2

    yy = W.lgssm_data(25)
    res = BootstrapSMC(StateSpaceModel(f_init, f_step), C["e", "y"].set(torch.tensor(yy)), 16384).run(genjax.random.key(3, impl))
    assert res.log_marginal_likelihood == pytest.approx(W.lgssm_exact_log_z(yy), abs=0.5)


ALL_CASES = [case_nested_calls, case_fast_estimate_path, case_exact_flip_flip_trivial, case_exact_flip_flip, case_non_marginal_target, case_readme_beta_bernoulli,
             case_static_gen_fn, case_distributions, case_uniform, case_fused_equals_eager, case_expression_arguments, case_params_equal_constants, case_trace_cache, case_particle_collection, case_custom_proposal,
             case_scan, case_scan_edge_cases, case_scan_fused_equals_loop, case_vmap, case_vmap_edge_cases, case_vmap_indexed_constraints, case_batched_estimates, case_gensp_estimators,
             case_marginal_with_algorithm, case_bootstrap_smc, case_general_smc, case_update, case_regenerate_and_rejuvenate, case_vector_valued_sites, case_index_request]
