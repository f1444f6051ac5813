"""The host-API cases on the product path: libgjx_hip.so on cuda:0, plus HIP-vs-oracle equality of
whole host-level runs."""

import pytest
import torch

import genjax
import host_api_cases as H
from genjax import ChoiceMapBuilder as C, Target, gen, normal, beta, flip
from genjax._amd.runtime import use_ops
from genjax.inference.smc import ImportanceK

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("impl", ["threefry", "philox"])
@pytest.mark.parametrize("case", H.ALL_CASES, ids=lambda c: c.__name__)
def test_host_api(hip_ops, case, impl):
    with use_ops(hip_ops):
        case(impl)


@pytest.mark.parametrize("impl", ["threefry", "philox"])
def test_host_level_run_matches_oracle(hip_ops, oracle_ops, impl):
    @gen
    def model(a):
        p = beta(2.0, a) @ "p"
        v = flip(p) @ "v"
        x = normal(p * 2.0, 1.5) @ "x"
        y = normal(x - 1.0, 0.5) @ "y"
        return x

    def run(ops):
        with use_ops(ops):
            t = Target(model, (3.0,), C["y"].set(0.3) | C["v"].set(True))
            coll = ImportanceK(t, k_particles=30000).run_smc(genjax.random.key(5, impl))
            ch = coll.get_particles().get_choices()
            part = coll.sample_particle(genjax.random.key(6, impl))
            rs = coll.resample(genjax.random.key(7, impl))
            return (coll.get_log_weights().cpu(), ch["p"].cpu(), ch["x"].cpu(), float(part.get_choices()["x"].cpu()),
                    rs.ancestors.cpu(), float(coll.get_log_marginal_likelihood_estimate().cpu()))

    g, o = run(hip_ops), run(oracle_ops)
    for a, b in zip(g, o):
        if isinstance(a, torch.Tensor):
            assert torch.equal(a, b)
        else:
            assert a == b


@pytest.mark.parametrize("impl", ["threefry", "philox"])
def test_nested_calls_and_transcendentals_match_oracle(hip_ops, oracle_ops, impl):
    """r03: a body with nested `@gen` calls (two levels, a callee called twice), `exp` / `log` and divisions by numbers
    between its sites — ONE generated kernel with per-scope keys on the device, the site walk with a key stack in the
    oracle: weights, nested choices and the estimate equal across the backends bit for bit; likewise through the per-site
    path (gjx_map_f32) and under a one-launch scan whose carry goes through exp / a division."""
    from genjax import gamma
    from genjax._amd.lang import GenerateHandler
    from genjax._amd.plan import try_fused_generate

    @gen
    def noise(scale):
        a = normal(0.0, scale) @ "a"
        b = gamma(2.0, 1.5) @ "b"
        return a * b

    @gen
    def inner(mu):
        e = noise(torch.exp(mu * 0.1)) @ "n"
        return normal(mu / 3.0 + e, 1.0) @ "x"

    @gen
    def model(t):
        z = normal(t, 1.0) @ "z"
        y1 = inner(z) @ "i1"
        y2 = inner(torch.log(y1 * y1 + 1.5)) @ "i2"
        _ = normal(2.0 / (y2 * y2 + 1.0) + y1, 0.7) @ "obs"
        return y2

    @gen
    def step(x, _):
        x2 = normal(torch.exp(x * 0.2) / 2.5, 0.4) @ "x"
        normal(x2, 0.5) @ "y"
        return x2, x2

    def run(ops):
        with use_ops(ops):
            n = 20000
            keys = genjax.random.split(genjax.random.key(11, impl), n)
            chm = C["obs"].set(0.3) | C["i2", "n", "b"].set(1.1)
            fused = try_fused_generate(model, keys, chm, (0.25,))
            assert fused is not None
            ftr, fw = fused
            h = GenerateHandler(keys, chm)
            h.run(model.source, (0.25,))
            est = ImportanceK(Target(model, (0.25,), chm), k_particles=n).log_marginal_likelihood_estimate(genjax.random.key(12, impl))
            ch = ftr.get_choices()
            T = 6
            tr, w = step.scan(n=T).generate(keys, C["y"].set(torch.linspace(-0.5, 0.5, T)), (0.1, None))
            return (fw.cpu(), h.weight.cpu(), ch["i1", "n", "a"].cpu(), ch["i2", "x"].cpu(), ftr.get_score().cpu(), float(est.cpu()),
                    w.cpu(), tr.get_choices()["x"].cpu())

    g, o = run(hip_ops), run(oracle_ops)
    for k, (a, b) in enumerate(zip(g, o)):
        _eq(a, b, k)
    assert torch.equal(g[0], g[1])  # fused == per-site on the device too


def _eq(a, b, what):
    if isinstance(a, torch.Tensor):
        assert torch.equal(a.cpu(), b.cpu()), what
    else:
        assert a == b, what


@pytest.mark.parametrize("impl", ["threefry", "philox"])
def test_host_level_flows_match_oracle(hip_ops, oracle_ops, impl):
    """Whole host-level runs, HIP tensors `torch.equal` the oracle's: conditional SMC and the GenSP estimators, a
    custom proposal, a trace update (incl. a masked per-particle update), partially constrained vmapped sites and a
    Scan model under ImportanceK — the f-rows of SURVEY 8(f), compared across backends instead of against expressions
    evaluated by the same backend."""
    import genjax as gj
    from genjax import SelectionBuilder as S, gamma
    from genjax.inference.smc import ChangeTarget

    @gen
    def model():
        z = normal(0.0, 1.0) @ "z"
        g = gamma(2.0, 1.5) @ "g"
        _ = normal(z * 0.5 + g, 0.5) @ "y"
        return z

    @gen
    def proposal(target):
        _ = normal(0.4 * target["y"], 0.8) @ "z"
        _ = gamma(2.5, 1.5) @ "g"

    @gj.vmap(in_axes=(0,))
    @gen
    def vk(x):
        return normal(x, 1.0) @ "z"

    @gj.scan(n=6)
    @gen
    def chain(x, o):
        z = normal(0.8 * x, 1.0) @ "z"
        _ = normal(z, 0.7) @ "obs"
        return z, z

    def run(ops):
        out = {}
        with use_ops(ops):
            key = gj.random.key(31, impl)
            t = Target(model, (), C["y"].set(0.9))
            alg = ImportanceK(t, k_particles=2000)
            # conditional SMC + estimators
            retained = C["z"].set(0.2) | C["g"].set(1.1)
            cs = alg.run_csmc(key, retained)
            out["csmc_lw"] = cs.get_log_weights()
            out["csmc_z"] = cs.get_particles().get_choices()["z"]
            out["est_logpdf"] = float(torch.as_tensor(alg.estimate_logpdf(key, retained, t)).cpu())
            out["est_z"] = float(torch.as_tensor(alg.estimate_normalizing_constant(key, Target(model, (), C["y"].set(0.1)))).cpu())
            out["est_recip"] = float(torch.as_tensor(alg.estimate_reciprocal_normalizing_constant(
                key, Target(model, (), C["y"].set(0.1)), retained, -1.25)).cpu())
            # custom proposal (batched) + its conditional form
            q = proposal.marginal()
            cq = ImportanceK(t, q, k_particles=1500).run_smc(key)
            out["q_lw"], out["q_g"] = cq.get_log_weights(), cq.get_particles().get_choices()["g"]
            out["q_csmc_lw"] = ImportanceK(t, q, k_particles=9).run_csmc(key, retained).get_log_weights()
            # different-target re-weighting
            out["ct_lw"] = ChangeTarget(alg, Target(model, (), C["y"].set(-0.3))).run_smc(key).get_log_weights()
            # update over a population: new observation, and a masked per-particle replacement of z
            keys = gj.random.split(key, 700)
            tr, w0 = model.importance(keys, C["y"].set(0.9), ())
            tr2, w, _, disc = tr.update(keys, C["y"].set(0.2))
            out["upd_w"], out["upd_score"] = w, tr2.get_score()
            flag = (torch.arange(700) % 3 == 0).to(w.device)
            tr3, w3, _, _ = tr.update(keys, C["z"].set(torch.full((700,), 0.5, device=w.device)).mask(flag))
            out["mask_w"], out["mask_z"] = w3, tr3.get_choices()["z"]
            # partially constrained vmapped site over a population
            ptr, pw = vk.importance(gj.random.split(key, 300), C[1, "z"].set(0.25), (torch.arange(3.0, device=w.device),))
            out["vmap_w"], out["vmap_z"] = pw, ptr.inner.get_choices()["z"]
            # Scan model under ImportanceK (the reference's own semantics of the state-space configs)
            obs = torch.tensor([0.3, -0.2, 0.5, 1.0, 0.1, -0.4], device=w.device)
            ts = Target(chain, (0.0, None), C[:, "obs"].set(obs))
            sc = ImportanceK(ts, k_particles=800).run_smc(key)
            out["scan_lw"] = sc.get_log_weights()
            out["scan_z"] = sc.get_particles().get_choices()["z"]
            out["scan_logz"] = float(sc.get_log_marginal_likelihood_estimate().cpu())
            # Regenerate and a Rejuvenate random walk over a population (rejuvenation moves)
            from genjax import Regenerate, StaticRequest
            from genjax.inference.requests import Rejuvenate

            k2 = gj.random.split(gj.random.key(77, impl), 700)
            rg, rw, _, bwd = Regenerate(S["z"]).edit(k2, tr, ())
            out["regen_w"], out["regen_z"] = rw, rg.get_choices()["z"]
            back, bw, _, _ = bwd.edit(k2, rg, ())
            out["regen_back_w"], out["regen_back_z"] = bw, back.get_choices()["z"]
            walk = StaticRequest({"z": Rejuvenate(normal, lambda chm: (chm.get_value(), 0.25))})
            rj, jw, _, _ = walk.edit(k2, tr, ())
            out["rejuv_w"], out["rejuv_z"] = jw, rj.get_choices()["z"]
        return out

    g, o = run(hip_ops), run(oracle_ops)
    assert g.keys() == o.keys()
    for k in g:
        _eq(g[k], o[k], k)


def test_regression_vectors_on_gpu(hip_ops):
    from test_oracle_pinning import check_regression, check_regression_r02

    check_regression(hip_ops)
    check_regression_r02(hip_ops)


@pytest.mark.parametrize("impl", ["threefry", "philox"])
def test_fast_math_context(hip_ops, impl):
    """`with genjax.fast_math():` — the opt-in hardware-transcendental mode of fused `@gen` and `Scan` plans at the host
    API: log-weights within 1e-5 relative (north star tolerance for paths without resampling), values within 1e-5
    absolute of the bit-exact mode; outside the context nothing changes."""
    @gen
    def model(a):
        z = normal(0.0, a) @ "z"
        w = normal(z * 0.5, 1.5) @ "w"
        _ = normal(w - 1.0, 0.5) @ "y"
        return w

    @genjax.scan(n=12)
    @gen
    def chain(x, _):
        z = normal(0.8 * x, 1.0) @ "z"
        _ = normal(z, 0.7) @ "obs"
        return z, z

    with use_ops(hip_ops):
        keys = genjax.random.split(genjax.random.key(3, impl), 50_000)
        obs = torch.linspace(-0.5, 0.5, 12)

        def run():
            tr, w = model.importance(keys, C["y"].set(0.3), (2.0,))
            st, sw = chain.importance(keys, C["obs"].set(obs), (0.0, None))
            return w, tr.get_choices()["w"], sw, st.get_choices()["z"]

        exact = run()
        with genjax.fast_math():
            fast = run()
        again = run()
    for a, b in zip(exact, again):
        assert torch.equal(a, b)
    for k, (a, b) in enumerate(zip(exact, fast)):
        if k % 2 == 0:  # log-weights
            assert float(((a - b).abs() / a.abs().clamp_min(1e-3)).max()) < 1e-5
        else:
            assert float((a - b).abs().max()) < 2e-5
    if impl == "philox":
        assert not torch.equal(exact[0], fast[0])  # the fast plan really ran


@pytest.mark.parametrize("impl", [0, 1])
def test_random_models_fused_equals_per_site(hip_ops, impl):
    """tests/fuzz_models.py on the device: random `@gen` bodies — the specialised kernel a body lowers to equals the per-site
    column path (the library's per-site kernels + torch's device arithmetic between the sites) bit for bit."""
    import fuzz_models

    with use_ops(hip_ops):
        compared, skipped = fuzz_models.run(12.0, 31 + impl, impl, n=3000)
    assert compared > 5 and skipped < compared


@pytest.mark.parametrize("impl", [0, 1])
def test_random_scan_kernels_fused_equals_loop(hip_ops, impl):
    import fuzz_models

    with use_ops(hip_ops):
        compared, skipped = fuzz_models.run_scans(12.0, 41 + impl, impl, n=2000)
    assert compared > 3 and skipped < compared


@pytest.mark.parametrize("impl", ["threefry", "philox"])
def test_tables_updated_in_place_between_calls(hip_ops, oracle_ops, impl):
    """`categorical(logits=trans[z])` passes the user's own device tensor to the plan, and the library
    derives per-plan tables (CDFs, guides, log-probabilities) from it at the plan's first compilation.  An in-place update
    of the tensor (an EM step) keeps its address: the plan cache must not hand back the plan built from the old contents —
    the second estimate equals the oracle's on the NEW table (and differs from the first)."""
    from genjax import categorical

    def run(ops):
        with use_ops(ops):
            dev = ops.device()
            g = torch.Generator().manual_seed(3)
            trans = (torch.randn(6, 6, generator=g) * 1.5).to(dev)
            emit = (torch.randn(6, 6, generator=g) * 1.5).to(dev)

            @gen
            def model():
                z0 = categorical(logits=trans[0]) @ "z0"
                z1 = categorical(logits=trans[z0]) @ "z1"
                _ = categorical(logits=emit[z1]) @ "y"
                return z1

            t = Target(model, (), C["y"].set(2))
            out = []
            for step in range(2):
                alg = ImportanceK(t, k_particles=20000)
                out.append(float(alg.log_marginal_likelihood_estimate(genjax.random.key(11, impl))))
                coll = alg.run_smc(genjax.random.key(12, impl))
                out.append(coll.get_particles().get_choices()["z1"].cpu())
                trans.mul_(-0.7).add_(0.1)  # in place: same address, new contents
            return out

    h, o = run(hip_ops), run(oracle_ops)
    for a, b in zip(h, o):
        _eq(a, b, "estimate / particles before and after the in-place update")
    assert h[0] != h[2]


@pytest.mark.parametrize("impl", ["threefry", "philox"])
def test_another_table_of_the_same_shape_costs_no_compilation(hip_ops, oracle_ops, impl):
    """device tables (categorical logits and the per-row tables derived from them) are kernel
    ARGUMENTS of the generated kernels, so the source is keyed by the model's structure: running the same model with a
    different transition tensor (another address, other contents) reuses the compiled module — `gjx_jit_stats().compiles`
    does not move — and still equals the oracle."""
    from genjax import categorical

    def run(ops, scale):
        with use_ops(ops):
            dev = ops.device()
            g = torch.Generator().manual_seed(5)
            trans = (torch.randn(7, 7, generator=g) * scale).to(dev)
            emit = (torch.randn(7, 7, generator=g)).to(dev)

            @gen
            def model():
                z0 = categorical(logits=trans[1]) @ "z0"
                z1 = categorical(logits=trans[z0]) @ "z1"
                _ = categorical(logits=emit[z1]) @ "y"
                return z1

            t = Target(model, (), C["y"].set(3))
            return float(ImportanceK(t, k_particles=10000).log_marginal_likelihood_estimate(genjax.random.key(21, impl)))

    first = run(hip_ops, 1.0)
    c0 = hip_ops.jit_stats()["compiles"]
    second = run(hip_ops, 2.5)  # new tensors: new addresses, new contents, the same structure
    assert hip_ops.jit_stats()["compiles"] == c0
    assert first == run(oracle_ops, 1.0) and second == run(oracle_ops, 2.5) and first != second


NOJIT_CHILD = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import torch
import genjax
from genjax import ChoiceMapBuilder as C, Target, gen, normal, flip
from genjax._amd.abi import GjxLib
from genjax._amd.ops import Ops
from genjax._amd.runtime import load_hip_ops, use_ops
from genjax.inference.smc import ImportanceK

hip, ora = load_hip_ops(), Ops(GjxLib(sys.argv[3], "cpu"))

@gen
def inner(m):
    a = normal(m, 0.5) @ "a"
    b = normal(a * 0.5 + m, torch.exp(a * 0.1)) @ "b"       # a program between two sites
    return a + b

@gen
def model(s):
    z = normal(0.0, 1.0) @ "z"
    u = inner(z * s + 0.25) @ "u"                                  # a nested call: its own scope of site keys
    k = flip(torch.sigmoid(u * 0.5)) @ "k"
    normal(torch.where(k, z, u), 0.7) @ "obs"
    return z

@gen
def step(carry, x):
    v = normal(carry * 0.9 + x, 1.0) @ "v"
    normal(v * v * 0.1, 0.5) @ "y"
    return v, v

def run(ops, impl):
    out = {}
    with use_ops(ops):
        dev = ops.device()
        n = 3000
        keys = genjax.random.split(genjax.random.key(3, impl), n)
        chm = C["obs"].set(0.3) | C["u", "a"].set(torch.linspace(-1, 1, n).to(dev))
        tr, w = model.importance(keys, chm, (0.25,))
        ch = tr.get_choices()
        out["w"], out["score"], out["z"], out["b"], out["k"] = w, tr.get_score(), ch["z"], ch["u", "b"], ch["k"]
        out["lz"] = ImportanceK(Target(model, (0.25,), C["obs"].set(0.3)), k_particles=5000).log_marginal_likelihood_estimate(genjax.random.key(5, impl))
        T = 7
        xs = torch.linspace(0.0, 1.0, T).to(dev)
        skeys = genjax.random.split(genjax.random.key(9, impl), 2000)
        schm = C[2, "y"].set(0.4) | C[5, "y"].set(-0.2)
        str_, sw = step.scan().importance(skeys, schm, (0.5, xs))
        out["sw"], out["sscore"] = sw, str_.get_score()
    return {k: torch.as_tensor(v).detach().cpu() for k, v in out.items()}

c0 = hip.jit_stats()["compiles"]
for impl in (0, 1):
    a, b = run(hip, impl), run(ora, impl)
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), (impl, k, a[k].reshape(-1)[:4], b[k].reshape(-1)[:4])
assert hip.jit_stats()["compiles"] == c0, "GJX_PLAN_JIT=0 must not compile anything"
print("ok")
"""


def test_bodies_with_programs_and_nested_calls_run_without_the_compiler():
    """with the plan compiler switched off (GJX_PLAN_JIT=0) the table interpreters refuse plans that hold
    programs (arithmetic between sites) or nested `@gen` calls — such bodies, importance plans and one-launch scans alike,
    then take the PER-SITE path on the device (the documented route without a compiler: runtime.compiler_switched_off).  A
    body with a nested call, programs (`exp`, `sigmoid`, `where`) and a flip, an `ImportanceK` estimate over it, and a scan
    kernel with a program: choices, scores, weights and the estimate equal the ORACLE's (which walks the fused plan itself)
    bit for bit under both generators, and nothing is compiled."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GJX_PLAN_JIT="0")
    r = subprocess.run([sys.executable, "-c", NOJIT_CHILD, os.path.join(root, "genjax-chi_amd"), os.path.join(root, "tests"),
                        os.path.join(root, "oracle", "libgjx_oracle.so")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
