"""The sharded filter's protocol on the HIP kernels of ONE device: virtual ranks (threads sharing the device
and its stream, `dist.ThreadComm`) run the same per-step sequence the RCCL ranks run — resample only the
source tiles that feed the rank's slots, max exchange, tile masses, ancestor shuffle by ranges — with every
region a rank did not receive poisoned.  Particles, ancestors and log Z must equal the CPU oracle's
single-rank filter bit for bit, and at full size the HIP single-rank filter."""

import pytest
import torch

from genjax._amd import workloads as W
from test_distributed_gloo import _run_virtual_ranks, check_degenerate_sharded, check_virtual_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("kind", ["lgssm", "hmm"])
@pytest.mark.parametrize("world,exchange", [(2, "ranges"), (4, "ranges"), (3, "allgather")])
def test_virtual_ranks_equal_oracle(hip_ops, oracle_ops, impl, kind, world, exchange):
    n_total, T = 1024 * world * 5, 11
    res = check_virtual_ranks(hip_ops, kind, impl, world, n_total, T, exchange, ref_ops=oracle_ops)
    if exchange == "ranges":
        assert all(0 < r["received"] < (T - 1) * (n_total - n_total // world) // 2 for r in res)


def test_virtual_ranks_full_size(hip_ops):
    """2 x 1e6 particles (rounded to whole tiles), T=20: sharded == single-device on the same kernels, and the
    shuffle moves a small fraction of what an all-gather would."""
    world, T = 2, 20
    n_total = 2 * 977 * 1024
    res = _run_virtual_ranks(hip_ops, "lgssm", 1, world, n_total, T, "ranges")
    ref = W.lgssm_smc(hip_ops, 1, 5, n_total, T, True)
    assert torch.equal(torch.cat([r["state"] for r in res]), ref["state"])
    assert torch.equal(torch.cat([r["ancestors"] for r in res], dim=1), ref["ancestors"])
    for r in res:
        assert torch.equal(r["out_q"], ref["out_q"]) and r["log_z"] == ref["log_z"]
        assert r["received"] < 0.05 * (T - 1) * (n_total // world)


def test_source_ranges(hip_ops):
    from test_distributed_gloo import check_source_ranges

    check_source_ranges(hip_ops)


@pytest.mark.parametrize("world", [2, 4])
def test_virtual_ranks_with_collapsing_weights(hip_ops, world):
    from test_distributed_gloo import check_degenerate_sharded

    check_degenerate_sharded(hip_ops, 1, world)


@pytest.mark.parametrize("native", [False, True])
def test_virtual_ranks_with_an_impossible_observation(hip_ops, native):
    from test_distributed_gloo import check_impossible_observation_sharded

    check_impossible_observation_sharded(hip_ops, 1, 3, native)


@pytest.mark.parametrize("impl", [0, 1])
def test_virtual_ranks_generated_filter(hip_ops, impl):
    from test_distributed_gloo import check_sharded_plan
    from test_gpu_parity_abi import _smc_plans

    check_sharded_plan(hip_ops, impl, 2, _smc_plans)
    check_sharded_plan(hip_ops, impl, 4, _smc_plans)


@pytest.mark.parametrize("kind", ["lgssm", "hmm"])
def test_sharded_adaptive_on_device(hip_ops, oracle_ops, kind):
    """ESS-adaptive resampling sharded over 3 virtual ranks on the HIP kernels: the per-rank decisions, the identity
    source ranges of kept steps and the accumulated weights equal the single-rank ORACLE filter bit for bit."""
    res = check_virtual_ranks(hip_ops, kind, 1, 3, 1024 * 6, 14, "ranges", ref_ops=oracle_ops, ess_threshold=0.5)
    assert 0 < int((res[0]["resampled"][1:] == 0).sum()) < 13


# ---- the library's own communicator and C driver (gjx.h "multi-GPU") -------------------------------------------------
@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("kind", ["lgssm", "hmm"])
@pytest.mark.parametrize("world,exchange", [(2, "ranges"), (4, "ranges"), (3, "allgather")])
def test_native_sharded_run_on_device(hip_ops, oracle_ops, impl, kind, world, exchange):
    """`gjx_smc_sharded_run_*` on the HIP kernels: virtual ranks of a `gjx_comm` local group share the device and its
    stream; the C driver polls the range kernel's ticket in pinned memory.  Equal to the single-rank ORACLE filter."""
    check_virtual_ranks(hip_ops, kind, impl, world, 1024 * world * 5, 11, exchange, ref_ops=oracle_ops, native=True)


def test_native_sharded_adaptive_plan_and_lse_on_device(hip_ops, oracle_ops):
    from test_distributed_gloo import check_native_lse_combine, check_sharded_plan
    from test_gpu_parity_abi import _smc_plans

    check_virtual_ranks(hip_ops, "lgssm", 1, 3, 1024 * 6, 14, "ranges", ref_ops=oracle_ops, ess_threshold=0.5, native=True)
    check_sharded_plan(hip_ops, 1, 2, _smc_plans, native=True)
    check_native_lse_combine(hip_ops)


def test_rccl_communicator_one_rank(hip_ops):
    """The RCCL transport on the one GPU of the box (world = 1): the library loads RCCL, builds a communicator from a
    unique id, and the sharded entry points run through it — a one-rank run equals the single-device filter, and the
    log-Z combine of one record is the record's own fold.  (More ranks need more GPUs: the protocol itself is covered
    by the virtual-rank tests above.)"""
    from genjax._amd import dist as gdist

    comm = gdist.NativeComm.rccl(hip_ops, 0, 1)
    assert (comm.rank, comm.world) == (0, 1)
    n, T = 1024 * 9, 12
    smc = gdist.ShardedSMC(hip_ops, "lgssm", 1, 5, n, T, 0, 1, True)
    got = smc.run_native(comm)
    ref = W.lgssm_smc(hip_ops, 1, 5, n, T, True)
    assert torch.equal(got["state"], ref["state"]) and torch.equal(got["ancestors"], ref["ancestors"])
    assert torch.equal(got["out_q"], ref["out_q"]) and got["log_z"] == ref["log_z"]
    whole = W.gaussian10_importance(hip_ops, 1, seed=4, n=256 * 11)
    shard = W.Gaussian10(hip_ops, 1, seed=4, n_local=256 * 11, first=0, n_total=256 * 11 + 1).step()  # (record form)
    lse, e, q = comm.lse_combine(shard["record"].reshape(1, -1).contiguous())
    assert (int(e.cpu()[0]), int(q.cpu()[0])) == (whole["row_e"], whole["row_q"])


def test_sharded_collapse_without_idle_tiles(hip_ops, oracle_ops):
    """a sharded filter whose weights collapse onto one tile while EVERY other tile keeps some
    mass.  The output-centric step needs no idle workgroups: every rank's output tiles inside the heavy run read the
    one heavy source tile (received through the shuffle) — equal to the single-rank oracle filter bit for bit."""
    check_degenerate_sharded(hip_ops, 1, 4)
    check_virtual_ranks(hip_ops, "lgssm", 1, 3, 1024 * 3 * 5, 9, "ranges", ref_ops=oracle_ops)
    check_virtual_ranks(hip_ops, "hmm", 0, 2, 1024 * 2 * 7, 9, "ranges", ref_ops=oracle_ops, ess_threshold=0.5)


def test_random_sharded_configurations_on_device(hip_ops, oracle_ops):
    """Random sharded filters over virtual ranks on the HIP kernels (Python and native drivers), each against the ORACLE's
    single-rank run."""
    from test_distributed_gloo import random_sharded_configs

    random_sharded_configs(hip_ops, 9, 10, ref_ops=oracle_ops)
