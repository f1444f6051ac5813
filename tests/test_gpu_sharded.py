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


# ---- r04: the peer transport on the device ----------------------------------------------------------
@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("kind", ["lgssm", "hmm"])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_peer_transport_on_device(hip_ops, oracle_ops, impl, kind, world):
    """`gjx_comm_init_peers` on the HIP kernels: `world` virtual ranks whose arenas are slices of one allocation.  Per step and
    rank ONE step launch (bounded wait for the peers' arrival words, then resampling that reads remote source windows where
    they live) and ONE signal launch (records into every arena, system-scope release, arrival word) — no collective, no
    exchange, nothing decided on the host.  Equal to the single-rank ORACLE filter bit for bit."""
    check_virtual_ranks(hip_ops, kind, impl, world, 1024 * world * 3, 9, "ranges", ref_ops=oracle_ops, native="peers")


def test_peer_transport_adaptive_plan_collapse_and_reuse_on_device(hip_ops, oracle_ops):
    from genjax._amd import dist as gdist
    from test_distributed_gloo import check_sharded_plan
    from test_gpu_parity_abi import _smc_plans

    res = check_virtual_ranks(hip_ops, "lgssm", 1, 3, 1024 * 6, 14, "ranges", ref_ops=oracle_ops, ess_threshold=0.5, native="peers")
    assert 0 < int((res[0]["resampled"][1:] == 0).sum()) < 13
    check_virtual_ranks(hip_ops, "hmm", 0, 2, 1024 * 2 * 7, 9, "ranges", ref_ops=oracle_ops, ess_threshold=0.5, native="peers")
    check_sharded_plan(hip_ops, 0, 2, _smc_plans, native="peers")
    check_sharded_plan(hip_ops, 1, 4, _smc_plans, native="peers")
    # one communicator, run after run: the arrival words only grow
    world, n_total = 3, 1024 * 3 * 4
    arenas = gdist.PeerArena.virtual(hip_ops, world, n_total, [torch.float32], False)
    comms = gdist.NativeComm.peers_virtual(hip_ops, arenas, timeout_ms=20000)
    for seed, T in ((5, 6), (6, 7), (7, 3)):
        got = _run_virtual_ranks(hip_ops, "lgssm", 1, world, n_total, T, "ranges", seed=seed, native="peers", peer_arenas=arenas,
                                 peer_comms=comms)
        ref = W.lgssm_smc(oracle_ops, 1, seed, n_total, T, True)
        assert torch.equal(torch.cat([r["state"] for r in got]).cpu(), ref["state"])
        assert torch.equal(torch.cat([r["ancestors"] for r in got], dim=1).cpu(), ref["ancestors"])
        assert all(r["log_z"] == ref["log_z"] for r in got)
    assert [int(x) for x in arenas[1].flags[:world].cpu()] == [6 + 1 + 7 + 1 + 3 + 1] * world


def test_peer_transport_collapsing_and_impossible_weights_on_device(hip_ops):
    """The worst cases of the single-device step through the peer transport: weights collapsing onto one tile (every rank's
    output tiles read ONE remote heavy tile) and an impossible observation (zero total mass: the population is kept)."""
    from test_distributed_gloo import check_degenerate_sharded, check_impossible_observation_sharded

    check_degenerate_sharded(hip_ops, 1, 4, native="peers")
    check_impossible_observation_sharded(hip_ops, 1, 3, "peers")


def test_peer_transport_full_size(hip_ops):
    """2 x 1e6 particles over two virtual ranks (1954 tiles: the precomputed-prefix route with its wait launch), T = 12: the
    peer transport equals the single-device filter on the same kernels."""
    world, T = 2, 12
    n_total = 2 * 977 * 1024
    res = _run_virtual_ranks(hip_ops, "lgssm", 1, world, n_total, T, "ranges", native="peers")
    ref = W.lgssm_smc(hip_ops, 1, 5, n_total, T, True)
    assert torch.equal(torch.cat([r["state"] for r in res]), ref["state"])
    assert torch.equal(torch.cat([r["ancestors"] for r in res], dim=1), ref["ancestors"])
    for r in res:
        assert torch.equal(r["out_q"], ref["out_q"]) and r["log_z"] == ref["log_z"]


def test_peer_transport_lost_peer_on_device(hip_ops):
    """A peer that never arrives: every workgroup's bounded wait gives up, the launches end, the error word is set and the
    Python layer raises — the device is not left spinning."""
    from genjax._amd import dist as gdist

    world, n_total, T = 2, 1024 * 2 * 3, 3
    arenas = gdist.PeerArena.virtual(hip_ops, world, n_total, [torch.float32], False)
    comm = gdist.NativeComm.peers(hip_ops, arenas[0], None, False, timeout_ms=50)  # rank 1 never runs
    smc = gdist.ShardedSMC(hip_ops, "lgssm", 1, 5, n_total, T, 0, world, False, arena=arenas[0])
    with pytest.raises(RuntimeError, match="timed out"):
        smc.run_native(comm)
    assert float(torch.ones(100, device="cuda").sum().item()) == 100.0


IPC_CHILD = r"""
import os, sys
root, rank, world, outdir, port, fine, big = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
sys.path.insert(0, os.path.join(root, "genjax-chi_amd"))
import torch
import torch.distributed as tdist
from genjax._amd import dist as gdist
from genjax._amd.runtime import load_hip_ops

ops = load_hip_ops()
tdist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
kind, impl, seed, n_total, T = "lgssm", 1, 5, 1024 * world * (576 if big else 6), (6 if big else 8)
# the product's own arena set-up (dist.PeerArena.ipc: hipMalloc or fine-grained hipExtMallocWithFlags, hipIpcGetMemHandle, the
# handles over the process group, hipIpcOpenMemHandle, a vote on failures) — what bench.py runs at N > 1
arena = gdist.PeerArena.ipc(ops, rank, world, n_total, [torch.float32], False, fine_grained=bool(fine))
# small: a wait launch per step (the ranks share the device, and a step's 12 workgroups would wait inside the step launch);
# big (1152 tiles): the step's first launch is the group-record launch — it carries the previous step's DEFERRED signal and
# waits for the peer itself (a handful of workgroups: the peer's launches run beside it), so no wait launch and no signal launch
comm = gdist.NativeComm.peers(ops, arena, None, not big, timeout_ms=60000)
for run in range(2):  # two runs on one communicator: the arrival words keep growing
    smc = gdist.ShardedSMC(ops, kind, impl, seed + run, n_total, T, rank, world, True, arena=arena)
    res = smc.run_native(comm)
    torch.cuda.synchronize()
    torch.save({k: (v.cpu().clone() if isinstance(v, torch.Tensor) else v) for k, v in res.items()
                if k in ("state", "logw", "ancestors", "out_e", "out_q", "log_z")}, os.path.join(outdir, f"res{run}_{rank}.pt"))
    tdist.barrier()
tdist.barrier()   # nobody unmaps / frees while a peer may still read
tdist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.parametrize("fine,big", [(0, 0), (1, 0), (0, 1)])
def test_peer_transport_two_processes_one_gpu_ipc(tmp_path, oracle_ops, fine, big):
    """two REAL processes share the one GPU; each allocates its arena (ordinary device memory, or
    fine-grained: bench.py's second chance), hands it to the other through hipIpcGetMemHandle / hipIpcOpenMemHandle
    (`dist.PeerArena.ipc`, the handles over a gloo process group), and runs `gjx_smc_sharded_run_lgssm` over the peer
    transport, twice on one communicator (wait launches in front of the steps, as the ranks compete for one device).
    Particles, ancestors, (e, q) and log Z equal the single-rank oracle filter bit for bit."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    world = 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", IPC_CHILD, root, str(r), str(world), str(tmp_path), str(port), str(fine), str(big)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for run in range(2):
        res = [torch.load(os.path.join(tmp_path, f"res{run}_{r}.pt")) for r in range(world)]
        ref = W.lgssm_smc(oracle_ops, 1, 5 + run, 1024 * world * (576 if big else 6), 6 if big else 8, True)
        assert torch.equal(torch.cat([r["state"] for r in res]), ref["state"])
        assert torch.equal(torch.cat([r["logw"] for r in res]), ref["logw"])
        assert torch.equal(torch.cat([r["ancestors"] for r in res], dim=1), ref["ancestors"])
        for r in res:
            assert torch.equal(r["out_q"], ref["out_q"]) and torch.equal(r["out_e"], ref["out_e"]) and r["log_z"] == ref["log_z"]


@pytest.mark.gpu
def test_bench_two_rank_line_rehearsal():
    """`python bench.py --gpus 2 --workload smc_lgssm` end to end with both rank processes on the one device
    (GJX_BENCH_REHEARSE: gloo process group, the reference log Z from the single-rank filter): the N > 1 line's own code —
    rank processes started by the parent, brackets with the max over ranks, the peer transport mapped through hipIpc and
    validated bit for bit before it is used, one JSON line from rank 0 with `cpu_baseline` — runs before an 8-GPU node sees it."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GJX_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", GJX_BENCH_BRACKET_MIN_S="0.002")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "smc_lgssm", "--steps", "2",
                        "--warmup", "1", "--particles", "200000"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["transport"].startswith("peers"), d["config"]
    assert "roofline" in d and d.get("cpu_baseline") and d["cpu_baseline"]["value"] > 0
    assert abs(d["log_z"] - d["log_z_exact"]) < 1.0
