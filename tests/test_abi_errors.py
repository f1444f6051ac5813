"""Error behaviour at the C-ABI (include/gjx.h: every entry point returns a status, never crashes on a bad argument):
invalid arguments are refused with GJX_ERR_INVALID by the oracle build and — validation happens on the host, before
any launch — by libgjx_hip.so without a GPU."""

import ctypes as C
import os

import numpy as np
import pytest
import torch

from genjax._amd import abi, prng, workloads as W
from genjax._amd.abi import GjxError, GjxLib
from genjax._amd.ops import Ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB = os.path.join(ROOT, "genjax-chi_amd", "lib", "libgjx_hip.so")


@pytest.fixture(scope="module", params=["oracle", "hip-nogpu"])
def lib_ops(request, oracle_ops):
    if request.param == "oracle":
        return oracle_ops
    if not os.path.exists(HIP_LIB):
        import __graft_entry__ as g

        g.build()
    return Ops(GjxLib(HIP_LIB, "cuda"))  # host-side validation only: nothing below reaches a launch


def _site(dist, a0, a1=None, obs=None, out_col=-1):
    s = abi.Site()
    s.dist, s.observed, s.out_col = dist, 0 if obs is None else 1, out_col
    s.arg[0] = a0
    if a1 is not None:
        s.arg[1] = a1
    if obs is not None:
        s.obs = obs
    return s


A = abi.Arg
c = lambda v: A(abi.ARG_CONST, 0, 0.0, v, None)


def test_plan_creation_refuses_bad_tables(lib_ops):
    ops = lib_ops
    with pytest.raises(GjxError):  # a site that refers to a LATER site
        ops.plan_create([_site(abi.DIST_NORMAL, A(abi.ARG_SITE, 1, 1.0, 0.0, None), c(1.0)), _site(abi.DIST_NORMAL, c(0.0), c(1.0))])
    with pytest.raises(GjxError):  # unknown distribution
        ops.plan_create([_site(17, c(0.0), c(1.0))])
    with pytest.raises(GjxError):  # parameter index out of range
        ops.plan_create([_site(abi.DIST_NORMAL, A(abi.ARG_PARAM, abi.MAX_PARAMS, 1.0, 0.0, None), c(1.0))])
    with pytest.raises(GjxError):  # STATE arguments belong to SMC / scan plans
        ops.plan_create([_site(abi.DIST_NORMAL, A(abi.ARG_STATE, 0, 1.0, 0.0, None), c(1.0))])
    plan = ops.plan_create([_site(abi.DIST_NORMAL, A(abi.ARG_PARAM, 2, 1.0, 0.0, None), c(1.0), out_col=0)])
    with pytest.raises(GjxError):  # fewer values than the highest referenced parameter
        plan.set_params([0.5, 0.25])
    plan.set_params([0.5, 0.25, 0.125])


def test_scan_plan_creation_refuses_bad_models(lib_ops):
    ops = lib_ops
    step = [_site(abi.DIST_NORMAL, A(abi.ARG_STATE, 0, 0.9, 0.0, None), c(1.0), out_col=0)]
    nxt = [A(abi.ARG_SITE, 0, 1.0, 0.0, None)]
    ops.scan_plan_create(step, nxt, 0)
    with pytest.raises(GjxError):  # the carry refers to a state component that does not exist
        ops.scan_plan_create([_site(abi.DIST_NORMAL, A(abi.ARG_STATE, 1, 1.0, 0.0, None), c(1.0), out_col=0)], nxt, 0)
    with pytest.raises(GjxError):  # an observation index beyond n_obs
        ops.scan_plan_create(step + [_site(abi.DIST_NORMAL, A(abi.ARG_SITE, 0, 1.0, 0.0, None), c(0.5), obs=A(abi.ARG_OBS, 2, 1.0, 0.0, None))], nxt, 1)
    with pytest.raises(GjxError):  # more observation columns than GJX_SMC_MAX_OBS
        ops.scan_plan_create(step, nxt, abi.SMC_MAX_OBS + 1)
    with pytest.raises(GjxError):  # a table lookup cannot be a carry component
        ops.scan_plan_create(step, [A(abi.ARG_TABLE, 0, 0.0, 0.0, 0x1000)], 0)


def test_scan_run_refuses_bad_io(oracle_ops):
    """(run-time validation, on the oracle: the same checks head gjx_scan_run in libgjx_hip.so)"""
    ops = oracle_ops
    sites, nxt = W.lgssm_scan_sites()
    plan = ops.scan_plan_create(sites, nxt, 1)
    kb = W.importance_particle_keys(prng.key(1, 1), 64)
    y = np.zeros((3, 1), dtype=np.float32)
    ops.scan_run(plan, kb, 64, 3, y, [0.0], [torch.float32])
    with pytest.raises(GjxError):  # T < 1
        ops.scan_run(plan, kb, 64, 0, np.zeros((0, 1), dtype=np.float32), [0.0], [torch.float32])
    with pytest.raises(GjxError):  # the latent site's value column is missing
        ops.scan_run(plan, kb, 64, 3, y, [0.0], [])
    with pytest.raises(ValueError):  # a folded key batch
        ops.scan_run(plan, kb.with_fold(1), 64, 3, y, [0.0], [torch.float32])
    with pytest.raises(ValueError):
        ops.scan_run(plan, kb, 64, 3, y, [0.0, 1.0], [torch.float32])


def test_comm_entry_points_validate(lib_ops):
    ops = lib_ops
    g = C.c_void_p()
    with pytest.raises(GjxError):
        ops.lib.call("gjx_comm_group_create", 0, C.byref(g))
    with pytest.raises(GjxError):
        ops.lib.call("gjx_comm_group_create", 1000, C.byref(g))
    ops.lib.call("gjx_comm_group_create", 2, C.byref(g))
    h = C.c_void_p()
    with pytest.raises(GjxError):
        ops.lib.call("gjx_comm_init_local", g, 2, C.byref(h))  # rank outside the group
    ops.lib.call("gjx_comm_init_local", g, 1, C.byref(h))
    assert ops.lib.call("gjx_comm_rank", h) == 1 and ops.lib.call("gjx_comm_world", h) == 2
    assert ops.lib.call("gjx_comm_rank", None) == -1
    with pytest.raises(GjxError):  # no communicator
        ops.lib.call("gjx_smc_sharded_run_lgssm", None, None, None, None, None, None)
    ops.lib.call("gjx_comm_destroy", h)
    ops.lib.call("gjx_comm_group_destroy", g)
    if ops.lib.device_type == "cpu":  # the oracle has no RCCL side
        with pytest.raises(GjxError):
            ops.lib.call("gjx_comm_unique_id", C.c_void_p(0))


def test_expression_programs_are_validated(lib_ops):
    """GJX_ARG_EXPR (gjx.h): malformed postfix programs are refused at plan creation by both builds."""
    ops = lib_ops
    keep = []
    E = lambda prog: abi.expr_arg(prog, keep)
    z = _site(abi.DIST_NORMAL, c(0.0), c(1.0), out_col=0)
    ok = [(abi.EXPR_SITE, 0, 0.0), (abi.EXPR_CONST, 0, 2.0), (abi.EXPR_MUL, 0, 0.0), (abi.EXPR_PARAM, 1, 0.0), (abi.EXPR_ADD, 0, 0.0),
          (abi.EXPR_CONST, 0, 3.0), (abi.EXPR_DIV, 0, 0.0)]
    plan = ops.plan_create([z, _site(abi.DIST_NORMAL, E(ok), c(1.0), out_col=1)])
    with pytest.raises(GjxError):  # the program reads parameter 1: two values are needed
        plan.set_params([0.5])
    plan.set_params([0.5, 0.25])
    bad = {
        "a value left on the stack": [(abi.EXPR_SITE, 0, 0.0), (abi.EXPR_CONST, 0, 2.0)],
        "an operator without operands": [(abi.EXPR_SITE, 0, 0.0), (abi.EXPR_ADD, 0, 0.0)],
        "a later site": [(abi.EXPR_SITE, 1, 0.0)],
        "an unknown opcode": [(abi.EXPR_SITE, 0, 0.0), (99, 0, 0.0)],
        "a state operand in an importance plan": [(abi.EXPR_STATE, 0, 0.0)],
        "too long": [(abi.EXPR_CONST, 0, 1.0)] + [(abi.EXPR_CONST, 0, 1.0), (abi.EXPR_ADD, 0, 0.0)] * 16,
        "too deep": [(abi.EXPR_CONST, 0, 1.0)] * 9 + [(abi.EXPR_ADD, 0, 0.0)] * 8,
    }
    for what, prog in bad.items():
        with pytest.raises(GjxError):
            ops.plan_create([z, _site(abi.DIST_NORMAL, E(prog), c(1.0), out_col=1)])
    with pytest.raises(GjxError):  # an empty program
        ops.plan_create([z, _site(abi.DIST_NORMAL, A(abi.ARG_EXPR, 0, 0.0, 0.0, None), c(1.0), out_col=1)])
    with pytest.raises(GjxError):  # not as an observed value
        ops.plan_create([z, _site(abi.DIST_NORMAL, c(0.0), c(1.0), obs=E(ok))])
    cat = _site(abi.DIST_CATEGORICAL, E([(abi.EXPR_SITE, 0, 0.0)]), out_col=1)
    logits = torch.zeros(2, 3)
    cat.n_cat, cat.n_rows, cat.cat_mode, cat.logits = 3, 2, 1, logits.data_ptr()
    with pytest.raises(GjxError):  # not as the row of a categorical site
        ops.plan_create([z, cat])


def test_scoped_plans_are_validated(lib_ops):
    """gjx_plan_create_scoped (nested `@gen` calls): scopes must be ranges of the site table in call order, each inside its
    caller, at most four deep; anything else is GJX_ERR_INVALID from both builds."""
    c = abi.Arg(abi.ARG_CONST, 0, 0.0, 1.0, None)
    sites = [_site(abi.DIST_NORMAL, c, c, out_col=k) for k in range(6)]
    good = [[(0, 1, 3)], [(0, 0, 6)], [(0, 1, 5), (1, 2, 4), (2, 2, 3)], [(0, 2, 2), (0, 2, 4)], [(0, 1, 3), (1, 3, 3), (0, 3, 5)],
            [(0, 6, 6)]]
    for sc in good:
        lib_ops.plan_create(sites, scopes=sc)
    bad = [[(0, 3, 1)], [(0, 0, 7)], [(1, 0, 2)], [(0, 2, 4), (0, 1, 3)], [(0, 1, 3), (1, 2, 4)], [(0, 1, 3), (0, 2, 5)],
           [(0, 0, 6), (1, 0, 6), (2, 0, 6), (3, 0, 6), (4, 0, 6)], [(-1, 0, 2)], [(0, 1, 3), (1, 4, 5)]]
    for sc in bad:
        with pytest.raises(GjxError) as e:
            lib_ops.plan_create(sites, scopes=sc)
        assert e.value.code == -1, sc  # GJX_ERR_INVALID
    with pytest.raises(GjxError):
        lib_ops.plan_create(sites, scopes=[(0, 0, 0)] * (abi.MAX_SCOPES + 1))


def test_importance_estimate_validates(lib_ops):
    """gjx_importance_estimate: an estimate-only plan (no value columns), scratch and tickets present, a lane only under
    PHILOX — anything else is GJX_ERR_INVALID before any launch."""
    c = abi.Arg(abi.ARG_CONST, 0, 0.0, 1.0, None)
    plan_cols = lib_ops.plan_create([_site(abi.DIST_NORMAL, c, c, out_col=0)])
    plan_est = lib_ops.plan_create([_site(abi.DIST_NORMAL, c, c, out_col=-1)])
    fn = lib_ops.lib._gjx_importance_estimate
    tick = (C.c_uint32 * abi.LSE_TICKET_WORDS)()
    buf = (C.c_uint64 * 64)()
    out = (C.c_float * 1)()

    def io(plan, impl=1, tickets=True, rows=True):
        lse = abi.LseOut(None, None, None, None, C.addressof(tick) if tickets else None, None, 0.0)
        return abi.EstimateIO(plan.handle.value, 256, None, 0, impl, C.addressof(buf) if rows else None, C.addressof(buf) if rows else None, lse)

    for bad, lane in ((io(plan_cols), 0), (io(plan_est, tickets=False), 0), (io(plan_est, rows=False), 0), (io(plan_est, impl=0), 5),
                      (io(plan_est, impl=7), 0)):
        assert fn(C.byref(bad), 1, 2, lane, C.addressof(out), 0.0, None) == -1
    assert fn(None, 1, 2, 0, C.addressof(out), 0.0, None) == -1
    assert fn(C.byref(io(plan_est)), 1, 2, 0, None, 0.0, None) == -1
