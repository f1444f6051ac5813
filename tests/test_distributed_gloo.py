"""The N>1 path on CPU: world_size-2 `gloo` processes drive the SAME host code that runs under
RCCL on the GPU box (the oracle is injected as the backend), and the sharded results must equal the
single-rank run bit for bit — populations are indexed by global slot and weight sums are exact."""

import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "libgjx_oracle.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, impl, n_total, n_imp_total, T, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
    import torch.distributed as dist

    from genjax._amd import dist as gdist, workloads as W
    from genjax._amd.abi import GjxLib
    from genjax._amd.ops import Ops

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    ops = Ops(GjxLib(ORACLE_LIB, "cpu"))
    first, n_imp = gdist.shard_rows(n_imp_total, rank, world)  # uneven row-aligned blocks, ragged tail
    wl = W.Gaussian10(ops, impl, seed=3, n_local=n_imp, first=first, n_total=n_imp_total)
    log_z, logw, e, q = gdist.importance_log_z(ops, wl)
    logw = logw.clone()
    pipe = gdist.BatchedImportance(ops, wl, batch=3)  # 5 passes through a double-buffered batch of 3
    d0 = pipe.run()
    d1 = pipe.run(2)
    for d, cnt in ((d0, 3), (d1, 2)):
        _, eb, qb = pipe.results(d)
        assert eb.numel() == cnt and all(torch.equal(eb[b:b + 1], e) and torch.equal(qb[b:b + 1], q) for b in range(cnt))
    d2 = pipe.run(1)  # reuses the first block after its exchange completed
    assert pipe.log_z(d2) == log_z
    # several passes per launch (seeds 3, 4): pass 0 is the estimate above, pass 1 an independent one
    pipe2 = gdist.BatchedImportance(ops, wl, batch=4, passes=2)
    _, e2, q2 = pipe2.results(pipe2.run(3))  # launches of 2 + 1 passes
    assert torch.equal(e2[0:1], e) and torch.equal(q2[0:1], q) and torch.equal(e2[2:3], e) and torch.equal(q2[2:3], q)
    logw_seed4 = pipe2.prep.logw_all[1, :n_imp].clone()
    smc = gdist.ShardedLgssmSMC(ops, impl, seed=5, n_total=n_total, T=T, rank=rank, world=world,
                                record_ancestors=True).run()
    torch.save(dict(log_z=log_z, logw=logw, e=e, q=q, e_seed4=e2[1:2].clone(), q_seed4=q2[1:2].clone(), logw_seed4=logw_seed4, smc_max=smc["out_max"], smc_q=smc["out_q"],
                    smc_state=smc["state"].clone(), smc_anc=smc["ancestors"], smc_log_z=smc["log_z"]),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("impl", [0, 1])
def test_two_ranks_equal_one_rank(tmp_path, oracle_ops, impl):
    from genjax._amd import workloads as W

    world, n_total, T = 2, 8192, 7
    n_imp_total = 256 * 37 + 100  # 38 rows (the last one ragged) split 19 / 19
    mp.spawn(_worker, args=(world, _free_port(), impl, n_total, n_imp_total, T, str(tmp_path)), nprocs=world,
             join=True)
    parts = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    # single-rank references
    ref_imp = W.Gaussian10(oracle_ops, impl, seed=3, n_local=n_imp_total).step()
    ref_smc = W.lgssm_smc(oracle_ops, impl, seed=5, n=n_total, T=T, want_ancestors=True)
    assert torch.equal(torch.cat([p["logw"] for p in parts]), ref_imp["logw"])
    for p in parts:
        assert torch.equal(p["e"], ref_imp["row_e"]) and torch.equal(p["q"], ref_imp["row_q"])
        assert p["log_z"] == oracle_ops.log_z_from_rows(ref_imp["row_e"], ref_imp["row_q"], n_imp_total)
        assert torch.equal(p["smc_max"], ref_smc["out_max"]) and torch.equal(p["smc_q"], ref_smc["out_q"])
        assert p["smc_log_z"] == ref_smc["log_z"]
    assert parts[0]["log_z"] == parts[1]["log_z"]
    ref4 = W.Gaussian10(oracle_ops, impl, seed=4, n_local=n_imp_total).step()  # the second pass of a launch
    assert torch.equal(torch.cat([p["logw_seed4"] for p in parts]), ref4["logw"])
    for p in parts:
        assert torch.equal(p["e_seed4"], ref4["row_e"]) and torch.equal(p["q_seed4"], ref4["row_q"])
    assert torch.equal(torch.cat([p["smc_state"] for p in parts]), ref_smc["state"])
    assert torch.equal(torch.cat([p["smc_anc"] for p in parts], dim=1), ref_smc["ancestors"])
