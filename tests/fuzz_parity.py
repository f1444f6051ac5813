"""Randomised HIP-vs-oracle parity (one MI355X):  python tests/fuzz_parity.py [seconds] [seed] [p_invalid]
(test infrastructure: it loads the oracle; `test_random_plans_fuzz` runs it for 25 s, and again with p_invalid = 0.12:
constants, observations, launch parameters and carries are then replaced, with that probability each, by NaN / +-inf / 0 /
negative / denormal / huge values — DESIGN 3.11; where the arithmetic gives NaN, a NaN is required on both sides)
Random site tables (all distributions, CONST / SITE / INPUT / PARAM / TABLE / EXPR arguments, observed and latent sites), random
population sizes (ragged rows included), both generators, lazy and materialised keys — importance plans, scan plans and
generated SMC filters.  Every output must be equal bit for bit.  Prints the failing case and exits 1 on a mismatch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "genjax-chi_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genjax._amd import abi, prng, workloads as W  # noqa: E402
from genjax._amd.abi import GjxLib  # noqa: E402
from genjax._amd.ops import KeyBatch, Ops  # noqa: E402
from genjax._amd.runtime import load_hip_ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
P_BAD = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
BAD = [float("nan"), float("inf"), float("-inf"), 0.0, -1.0, 1e-45, 3e38, -3e38]


def maybe_bad(x):
    return float(rng.choice(BAD)) if P_BAD and rng.random() < P_BAD else x
hip = load_hip_ops()
ora = Ops(GjxLib(os.path.join(ROOT, "oracle", "libgjx_oracle.so"), "cpu"))
A = abi.Arg


EXPR_KEEP: list = []  # the programs' ctypes arrays (the libraries copy them at plan creation)


def const(lo, hi):
    return A(abi.ARG_CONST, 0, 0.0, maybe_bad(float(rng.uniform(lo, hi))), None)


def random_sites(n_sites, mode, n_state=0, n_obs=0, n_inputs=0, tables=None):
    """mode: 'imp' | 'scan' | 'smc'.  Positive-valued arguments are kept positive by construction."""
    sites, kinds = [], []  # kinds[q]: 'real' | 'pos' | 'unit' | 'int'
    out_col = 0
    for q in range(n_sites):
        dist = int(rng.choice([abi.DIST_NORMAL, abi.DIST_NORMAL, abi.DIST_GAMMA, abi.DIST_BETA, abi.DIST_BERNOULLI] +
                              ([abi.DIST_CATEGORICAL] if tables is not None else [])))
        s = abi.Site()
        s.dist = dist
        if dist == abi.DIST_CATEGORICAL:  # a logits table; the row is constant or chosen by an earlier 0/1 site
            K = int(rng.choice([3, 17, 64]))
            ints = [i for i, kk in enumerate(kinds) if kk == "int01"]
            rows = 2 if ints and rng.random() < 0.6 else 1
            tab = rng.normal(0, 1.5, (rows, K)).astype(np.float32)
            tables.append(tab)
            s.n_cat, s.n_rows, s.cat_mode = K, rows, int(rng.integers(2))
            s.arg[0] = A(abi.ARG_SITE, int(rng.choice(ints)), 1.0, 0.0, None) if rows == 2 else A(abi.ARG_CONST, 0, 0.0, 0.0, None)
            s.logits = len(tables) - 1  # (index; replaced by the device pointer of each backend)
            observed = rng.random() < 0.35
            s.observed = int(observed)
            if observed:
                s.obs = A(abi.ARG_CONST, 0, 0.0, float(rng.integers(K)), None)
                s.out_col = -1
            else:
                s.out_col = out_col if mode != "smc" else -1
                out_col += 1 if mode != "smc" else 0
            kinds.append("cat")
            sites.append(s)
            continue

        def loc_arg():
            opts = ["const"]
            if any(k in ("real", "pos", "unit") for k in kinds):
                opts += ["site"] * 2
            if n_state:
                opts.append("state")
            if n_inputs and mode == "imp":
                opts.append("input")
            if mode == "imp":
                opts.append("param")
            k = rng.choice(opts)
            sc, off = float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1, 1))
            if k == "site":
                ref = int(rng.choice([i for i, kk in enumerate(kinds) if kk in ("real", "pos", "unit")]))
                return A(abi.ARG_SITE, ref, sc, off, None)
            if k == "state":
                return A(abi.ARG_STATE, int(rng.integers(n_state)), sc, off, None)
            if k == "input":
                return A(abi.ARG_INPUT, int(rng.integers(n_inputs)), sc, off, None)
            if k == "param":
                return A(abi.ARG_PARAM, int(rng.integers(4)), sc, off, None)
            return const(-2, 2)

        def pos_arg():
            opts = ["const", "const"]
            if any(k in ("pos", "unit") for k in kinds):
                opts.append("site")
            if mode == "imp":
                opts.append("param_pos")
            k = rng.choice(opts)
            if k == "site":
                ref = int(rng.choice([i for i, kk in enumerate(kinds) if kk in ("pos", "unit")]))
                return A(abi.ARG_SITE, ref, float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.2, 1.0)), None)
            if k == "param_pos":
                return A(abi.ARG_PARAM, 4 + int(rng.integers(2)), float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.1, 0.5)), None)
            return const(0.3, 2.5)

        def expr_loc():
            """A postfix program over earlier real-valued sites / state / inputs / parameters / constants (GJX_ARG_EXPR)."""
            prog, depth = [], 0

            def operand():
                opts = ["const"]
                reals = [i for i, kk in enumerate(kinds) if kk in ("real", "pos", "unit", "int01")]
                if reals:
                    opts += ["site"] * 3
                if n_state:
                    opts.append("state")
                if mode in ("scan", "smc") and n_obs:
                    opts.append("obs")
                if n_inputs and mode == "imp":
                    opts.append("input")
                if mode == "imp":
                    opts.append("param")
                k = rng.choice(opts)
                if k == "site":
                    return (abi.EXPR_SITE, int(rng.choice(reals)), 0.0)
                if k == "state":
                    return (abi.EXPR_STATE, int(rng.integers(n_state)), 0.0)
                if k == "obs":
                    return (abi.EXPR_OBS, 0, 0.0)
                if k == "input":
                    return (abi.EXPR_INPUT, int(rng.integers(n_inputs)), 0.0)
                if k == "param":
                    return (abi.EXPR_PARAM, int(rng.integers(4)), 0.0)
                return (abi.EXPR_CONST, 0, maybe_bad(float(rng.uniform(-2, 2))))

            for _ in range(int(rng.integers(2, 5))):
                prog.append(operand())
                depth += 1
                while depth >= 2 and rng.random() < 0.7:
                    prog.append((int(rng.choice([abi.EXPR_ADD, abi.EXPR_SUB, abi.EXPR_MUL, abi.EXPR_ADD, abi.EXPR_MAX, abi.EXPR_MIN])), 0, 0.0))
                    depth -= 1
                if rng.random() < 0.15:
                    prog.append((int(rng.choice([abi.EXPR_NEG, abi.EXPR_NEG, abi.EXPR_ABS])), 0, 0.0))
                if rng.random() < 0.06:
                    prog += [(abi.EXPR_ABS, 0, 0.0), (abi.EXPR_SQRT, 0, 0.0)]
                if rng.random() < 0.08 and len(prog) < 9:  # where(top < c, t, f) with the stack's top as the compared value
                    prog += [(abi.EXPR_CONST, 0, float(rng.uniform(-1, 1))), (int(rng.choice([abi.EXPR_LT, abi.EXPR_LE, abi.EXPR_EQ])), 0, 0.0), operand(),
                             (abi.EXPR_CONST, 0, float(rng.uniform(-1, 1))), (abi.EXPR_SELECT, 0, 0.0)]
                if rng.random() < 0.12 and len(prog) < 9:  # exp / log of what is on the stack (r03: GJX_EXPR_EXP / _LOG)
                    prog += [(abi.EXPR_CONST, 0, float(rng.uniform(-0.4, 0.4))), (abi.EXPR_MUL, 0, 0.0), (abi.EXPR_EXP, 0, 0.0)]
                    if rng.random() < 0.5:  # log(exp(c x) + k), k > 0: a positive argument (p_invalid runs reach the others)
                        prog += [(abi.EXPR_CONST, 0, maybe_bad(float(rng.uniform(0.1, 2.0)))), (abi.EXPR_ADD, 0, 0.0), (abi.EXPR_LOG, 0, 0.0)]
                if rng.random() < 0.2 and len(prog) < 12:  # a division by a constant away from zero
                    prog += [(abi.EXPR_CONST, 0, maybe_bad(float(rng.choice([-1, 1]) * rng.uniform(0.5, 3.0)))), (abi.EXPR_DIV, 0, 0.0)]
            while depth >= 2:
                prog.append((int(rng.choice([abi.EXPR_ADD, abi.EXPR_SUB, abi.EXPR_MUL])), 0, 0.0))
                depth -= 1
            if len(prog) > abi.MAX_EXPR_OPS:  # (too long for one argument: draw another)
                return expr_loc()
            return abi.expr_arg(prog, EXPR_KEEP)

        if dist == abi.DIST_NORMAL:
            s.arg[0], s.arg[1] = (expr_loc() if rng.random() < 0.3 else loc_arg()), pos_arg()
            kinds.append("real")
        elif dist == abi.DIST_GAMMA:
            s.arg[0], s.arg[1] = pos_arg(), pos_arg()
            kinds.append("pos")
        elif dist == abi.DIST_BETA:
            s.arg[0], s.arg[1] = pos_arg(), pos_arg()
            kinds.append("unit")
        else:
            if any(k == "unit" for k in kinds) and rng.random() < 0.5:
                ref = int(rng.choice([i for i, kk in enumerate(kinds) if kk == "unit"]))
                s.arg[0] = A(abi.ARG_SITE, ref, 1.0, 0.0, None)
            else:
                s.arg[0] = const(0.1, 0.9)
            kinds.append("int01")
        observed = rng.random() < 0.35
        s.observed = int(observed)
        if observed:
            val = {"real": rng.uniform(-2, 2), "pos": rng.uniform(0.2, 3), "unit": rng.uniform(0.1, 0.9), "int01": float(rng.integers(2))}[kinds[-1]]
            if mode in ("scan", "smc") and n_obs and rng.random() < 0.7 and kinds[-1] in ("real", "int01"):
                s.obs = A(abi.ARG_OBS, int(rng.integers(n_obs)), 1.0, 0.0, None)
                if kinds[-1] == "int01":
                    s.obs = A(abi.ARG_OBS, n_obs - 1, 1.0, 0.0, None)  # (the last observation column holds 0/1 values)
            else:
                s.obs = A(abi.ARG_CONST, 0, 0.0, maybe_bad(float(val)) if kinds[-1] != "int01" else float(val), None)
            s.out_col = -1
        else:
            s.out_col = out_col if mode != "smc" else -1
            out_col += 1 if mode != "smc" else 0
        sites.append(s)
    return sites, kinds, out_col


def bind_tables(ops, sites, tables):
    """Copies of the site table with every categorical site's logits pointing at this backend's copy of its table."""
    keep, out = [], []
    for s_ in sites:
        t = abi.Site()
        C_ = __import__("ctypes")
        C_.memmove(C_.byref(t), C_.byref(s_), C_.sizeof(abi.Site))
        if s_.dist == abi.DIST_CATEGORICAL:
            dev = torch.from_numpy(tables[int(s_.logits or 0)]).to(ops.device()).contiguous()
            keep.append(dev)
            t.logits = dev.data_ptr()
        out.append(t)
    return out, keep


def dtypes_for(sites, kinds):
    out = []
    for s, k in zip(sites, kinds):
        if s.out_col >= 0:
            out.append(torch.int32 if k in ("int01", "cat") else torch.float32)
    return out


def eq(a, b, what, ctx):
    a, b = (a.cpu() if isinstance(a, torch.Tensor) else a), (b.cpu() if isinstance(b, torch.Tensor) else b)
    ok = torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b
    if not ok:
        # NaN == NaN bitwise
        if isinstance(a, torch.Tensor) and a.dtype.is_floating_point and torch.equal(a.view(torch.int32), b.view(torch.int32)):
            return
        # invalid-value runs: a NaN on both sides, whatever its sign / payload
        if P_BAD and isinstance(a, torch.Tensor) and a.dtype.is_floating_point and torch.equal(a.isnan(), b.isnan()) and \
                torch.equal(a.nan_to_num(nan=0.0).view(torch.int32), b.nan_to_num(nan=0.0).view(torch.int32)):
            return
        print("MISMATCH", what, ctx)
        if isinstance(a, torch.Tensor):
            d = (a != b) & ~(a.isnan() & b.isnan()) if a.dtype.is_floating_point else a != b
            idx = d.flatten().nonzero().flatten()[:6]
            print(f"  {int(d.sum())} of {a.numel()} differ; first at {idx.tolist()}: {a.flatten()[idx].tolist()} vs {b.flatten()[idx].tolist()}")
        sys.exit(1)


def random_scopes(n_sites):
    """Random nested calls over a flat table of n_sites sites (gjx_scope, in call order): up to three top-level calls, each
    maybe with one callee of its own, some without any site."""
    if rng.random() < 0.5 or n_sites < 1:
        return []
    out, pos = [], 0
    for _ in range(int(rng.integers(1, 4))):
        if pos > n_sites:
            break
        b = int(rng.integers(pos, n_sites + 1))
        e = int(rng.integers(b, n_sites + 1))
        out.append((0, b, e))
        me = len(out)
        if e > b and rng.random() < 0.5:  # a callee of this call
            ib = int(rng.integers(b, e + 1))
            ie = int(rng.integers(ib, e + 1))
            out.append((me, ib, ie))
        pos = e
    return out


t_end, cases = time.time() + budget, 0
while time.time() < t_end:
    impl = int(rng.integers(2))
    n = int(rng.choice([1, 3, 255, 256, 257, 1000, 4099, 20000, 70004]))
    mode = str(rng.choice(["imp", "imp", "scan", "smc"]))
    seed = int(rng.integers(1 << 30))
    ctx = dict(mode=mode, impl=impl, n=n, seed=seed, case=cases)
    kb = W.importance_particle_keys(prng.key(seed, impl), n)
    if mode == "imp":
        n_inputs = int(rng.integers(0, 3))
        tables = []
        sites, kinds, n_out = random_sites(int(rng.integers(1, 12)), "imp", n_inputs=n_inputs, tables=tables)
        params = [maybe_bad(float(x)) for x in rng.uniform(-1, 1, 4)] + [maybe_bad(float(x)) for x in rng.uniform(0.5, 2.0, 2)]
        cols = [torch.from_numpy(rng.uniform(-1, 1, n).astype(np.float32)) for _ in range(n_inputs)]
        outs = []
        scopes = random_scopes(len(sites))
        ctx["scopes"] = scopes
        for ops in (hip, ora):
            bound, keep = bind_tables(ops, sites, tables)
            plan = ops.plan_create(bound, scopes=scopes)
            plan.set_params(params)
            keys = kb if rng.random() < 0.7 or True else kb
            vals, score, logw, mp, rows = ops.importance_run(plan, keys, n, [c_.to(ops.device()) for c_ in cols], dtypes_for(sites, kinds),
                                                             want_rows=True)
            outs.append(vals + [score, logw, mp, rows.e, rows.s])
        for i, (a, b) in enumerate(zip(*outs)):
            eq(a, b, f"importance output {i}", ctx)
    elif mode == "scan":
        n_state, n_obs, T = int(rng.integers(1, 3)), 2, int(rng.integers(1, 9))
        tables = []
        sites, kinds, n_out = random_sites(int(rng.integers(1, 7)), "scan", n_state=n_state, n_obs=n_obs, tables=tables)
        real = [i for i, k in enumerate(kinds) if k in ("real", "pos", "unit")]
        nxt = [A(abi.ARG_SITE, int(rng.choice(real)), float(rng.uniform(-1, 1)), float(rng.uniform(-0.5, 0.5)), None) if real and rng.random() < 0.8
               else A(abi.ARG_STATE, int(rng.integers(n_state)), 0.5, 0.1, None) for _ in range(n_state)]
        if real and rng.random() < 0.4:  # a carry component that is an expression over a site, the carry and the input
            nxt[0] = abi.expr_arg([(abi.EXPR_STATE, 0, 0.0), (abi.EXPR_SITE, int(rng.choice(real)), 0.0), (abi.EXPR_CONST, 0, float(rng.uniform(-0.5, 0.5))),
                                   (abi.EXPR_MUL, 0, 0.0), (abi.EXPR_ADD, 0, 0.0), (abi.EXPR_OBS, 0, 0.0), (abi.EXPR_SUB, 0, 0.0)], EXPR_KEEP)
        obs = np.stack([[maybe_bad(float(x)) for x in rng.uniform(-1, 1, T)], rng.integers(0, 2, T)], axis=1).astype(np.float32)
        carry0 = [maybe_bad(float(rng.uniform(-1, 1))) for _ in range(n_state)]
        outs = []
        scopes = random_scopes(len(sites))
        ctx["scopes"] = scopes
        for ops in (hip, ora):
            bound, keep = bind_tables(ops, sites, tables)
            plan = ops.scan_plan_create(bound, nxt, n_obs, scopes=scopes)
            o = ops.scan_run(plan, kb, n, T, obs, carry0, dtypes_for(sites, kinds))
            outs.append(o["values"] + o["carry"] + [o["score"], o["logw"], o["max_partials"], o["rows"].e, o["rows"].s])
        for i, (a, b) in enumerate(zip(*outs)):
            eq(a, b, f"scan output {i}", ctx)
    else:
        if n < 256:
            continue
        n_state, n_obs, T = int(rng.integers(1, 3)), 2, int(rng.integers(2, 7))
        init, ik, _ = random_sites(int(rng.integers(1, 4)), "smc", n_state=0, n_obs=n_obs)
        step, sk_, _ = random_sites(int(rng.integers(1, 5)), "smc", n_state=n_state, n_obs=n_obs)
        ireal = [i for i, k in enumerate(ik) if k not in ("int01", "cat")]
        sreal = [i for i, k in enumerate(sk_) if k not in ("int01", "cat")]
        if not ireal or not sreal:
            continue
        istate = [A(abi.ARG_SITE, int(rng.choice(ireal)), 1.0, 0.0, None) for _ in range(n_state)]
        nstate = [A(abi.ARG_SITE, int(rng.choice(sreal)), float(rng.uniform(0.5, 1.0)), 0.0, None) for _ in range(n_state)]
        if rng.random() < 0.4:
            nstate[0] = abi.expr_arg([(abi.EXPR_STATE, 0, 0.0), (abi.EXPR_CONST, 0, 0.9), (abi.EXPR_MUL, 0, 0.0), (abi.EXPR_SITE, int(rng.choice(sreal)), 0.0),
                                      (abi.EXPR_CONST, 0, float(rng.uniform(0.1, 0.6))), (abi.EXPR_MUL, 0, 0.0), (abi.EXPR_ADD, 0, 0.0)], EXPR_KEEP)
            istate[0] = abi.expr_arg([(abi.EXPR_SITE, int(rng.choice(ireal)), 0.0), (abi.EXPR_OBS, 0, 0.0), (abi.EXPR_CONST, 0, 0.25), (abi.EXPR_MUL, 0, 0.0),
                                      (abi.EXPR_ADD, 0, 0.0)], EXPR_KEEP)
        obs = np.stack([[maybe_bad(float(x)) for x in rng.uniform(-1, 1, T)], rng.integers(0, 2, T)], axis=1).astype(np.float32)
        skeys, rkeys = W.smc_key_schedule(prng.key(seed, impl), T)
        ess = float(rng.choice([0.0, 0.0, 0.5]))
        outs = []
        isc, ssc = random_scopes(len(init)), random_scopes(len(step))
        ctx["scopes"] = (isc, ssc)
        for ops in (hip, ora):
            plan = ops.smc_plan_create(init, step, istate, nstate, n_obs, init_scopes=isc, step_scopes=ssc)
            r = ops.smc_run_plan(plan, impl, n, skeys, rkeys, obs, True, ess_threshold=ess, want_flags=True)
            outs.append([r[0], r[1], *r[2], r[3], r[4]] + ([r[5]] if r[5] is not None else []))
        for i, (a, b) in enumerate(zip(*outs)):
            eq(a, b, f"smc output {i}", dict(ctx, ess=ess, T=T))
    cases += 1
    del EXPR_KEEP[:]
print(f"fuzz ok: {cases} random cases (p_invalid {P_BAD}), HIP == oracle bit for bit")
