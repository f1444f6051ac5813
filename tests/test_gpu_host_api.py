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


def test_regression_vectors_on_gpu(hip_ops):
    from test_oracle_pinning import check_regression

    check_regression(hip_ops)
