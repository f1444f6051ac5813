"""Host logic (`@gen`, ChoiceMap, Target, ImportanceK, ...) on CPU: the oracle is injected as the
backend through the test hook `use_ops` so that the SAME host code the GPU runs is exercised here."""

import pytest

import host_api_cases as H
from genjax._amd.runtime import use_ops


@pytest.mark.parametrize("impl", ["threefry", "philox"])
@pytest.mark.parametrize("case", H.ALL_CASES, ids=lambda c: c.__name__)
def test_host_api(oracle_ops, case, impl):
    with use_ops(oracle_ops):
        case(impl)


@pytest.mark.parametrize("impl", [0, 1])
def test_random_models_fused_equals_per_site(oracle_ops, impl):
    """tests/fuzz_models.py: random `@gen` bodies — the fused kernel a body lowers to equals the per-site column path."""
    import fuzz_models

    with use_ops(oracle_ops):
        compared, skipped = fuzz_models.run(6.0, 11 + impl, impl, min_compared=201)
    assert compared > 200 and skipped < compared


@pytest.mark.parametrize("impl", [0, 1])
def test_random_scan_kernels_fused_equals_loop(oracle_ops, impl):
    """tests/fuzz_models.py: random scan kernels (tuple carry, a scanned input, expressions over carry / sites / input) — the
    one-launch scan equals the host loop of per-site launches."""
    import fuzz_models

    with use_ops(oracle_ops):
        compared, skipped = fuzz_models.run_scans(6.0, 17 + impl, impl, min_compared=101)
    assert compared > 100 and skipped < compared
