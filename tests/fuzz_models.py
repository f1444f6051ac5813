"""Randomised `@gen` model bodies through the host API: the fused kernel a body lowers to (plan.py: affine arguments,
postfix programs, launch parameters, per-particle inputs) against the per-site column path (GenerateHandler) — weights,
scores, every choice and the returned expression must be equal bit for bit on whatever backend is installed.
Used by tests/test_host_api_cpu.py (oracle backend) and tests/test_gpu_host_api.py (HIP backend):
    run(seconds, seed) -> (models compared, models that did not lower)"""
import time

import numpy as np
import torch

import genjax
from genjax import ChoiceMapBuilder as C, beta, flip, gamma, gen, normal
from genjax._amd.lang import GenerateHandler, StaticTrace
from genjax._amd.plan import try_fused_generate
from genjax._amd.runtime import get_ops


def _expr(rng, reals, params, depth=0):
    """A random f32 expression over earlier real-valued sites, model arguments and literals."""
    r = rng.random()
    if depth >= 3 or r < 0.3 or not reals:
        k = rng.random()
        if reals and k < 0.6:
            return str(rng.choice(reals))
        if params and k < 0.8:
            return str(rng.choice(params))
        return repr(round(float(rng.uniform(-2, 2)), 3))
    if r < 0.4:
        return f"(-{_expr(rng, reals, params, depth + 1)})"
    if r < 0.52:  # what reference bodies write between sites: exp / log of a traced value, a division by (of) a number
        x = str(rng.choice(reals))
        lit = round(float(rng.uniform(0.3, 3.0)), 3)
        y = str(rng.choice(reals))
        return str(rng.choice([f"torch.where({x} > {round(lit - 1.5, 3)}, {x}, {y} * 0.5)", f"torch.where(({x} < {y}) | ({y} >= {lit}), {lit}, {x})",
                               f"torch.clamp({x}, min={-lit}, max={lit})", f"torch.nn.functional.softplus({x})",
                               f"torch.maximum({x}, torch.tensor({round(lit - 1.5, 3)}))", f"torch.sigmoid({x} * {lit})", f"torch.sqrt({x} * {x} + {lit})", f"torch.abs({x})", f"({x} + {lit}).abs().sqrt()",
                               f"torch.reciprocal({x} * {x} + {lit})",
                               f"torch.exp({x} * {round(float(rng.uniform(-0.5, 0.5)), 3)})", f"torch.log({x} * {x} + {lit})",
                               f"({x} * {x} + {lit}).log()", f"({_expr(rng, reals, params, depth + 1)} / {lit})",
                               f"({lit} / ({x} * {x} + {lit}))", f"torch.div({x}, {lit})"]))
    op = rng.choice(["+", "-", "*", "*", "+"])
    return f"({_expr(rng, reals, params, depth + 1)} {op} {_expr(rng, reals, params, depth + 1)})"


def _pos(rng, reals, poss, params):
    k = rng.random()
    if poss and k < 0.35:
        return f"({rng.choice(poss)} * {round(float(rng.uniform(0.3, 2.0)), 3)} + {round(float(rng.uniform(0.1, 1.0)), 3)})"
    if reals and k < 0.6:
        a = _expr(rng, reals, params, 2)
        return f"({a} * {a} + {round(float(rng.uniform(0.2, 1.5)), 3)})"
    if params and k < 0.75:
        return "s"
    return repr(round(float(rng.uniform(0.3, 2.5)), 3))


def random_model(rng):
    """-> (source of `def model(s, t): ...`, site names with kinds)."""
    lines, reals, poss, units, ints, sites = [], [], [], [], [], []
    params = ["t"]  # `s` is a positive argument (scales), `t` any real
    helpers = []
    for q in range(int(rng.integers(2, 9))):
        name = f"v{q}"
        kind = rng.choice(["normal", "normal", "normal", "gamma", "beta", "flip"])
        if rng.random() < 0.2 and len(helpers) < 3:  # a nested `@gen` call: its own sites, maybe a call of its own
            hname, inner = f"h{len(helpers)}", []
            if helpers and rng.random() < 0.4:
                inner.append(f"    c = {helpers[-1]}(a * 0.5) @ 'c'")
            hs = round(float(rng.uniform(0.3, 1.5)), 3)
            inner.append(f"    p = normal(a{' + c' if inner else ''}, {hs}) @ 'p'")
            if rng.random() < 0.5:
                inner.append(f"    g = gamma(p * p + {hs}, 1.5) @ 'g'")
                inner.append("    return p * g")
            else:
                inner.append("    return p + a")
            lines_h = f"@gen\ndef {hname}(a):\n" + "\n".join(inner) + "\n"
            helpers.append(hname)
            pool0 = reals + poss + units
            lines.append(f"    {name} = {hname}({_expr(rng, pool0, params)}) @ '{name}'")
            reals.append(name)
            sites.append((name, "call"))
            globals().setdefault("_HELPER_SRC", {})[hname] = lines_h
            lines.append(f"    #helper {hname}")
            continue
        # (a flip value enters arithmetic times a literal: torch has no bool - bool, and neither has numpy)
        pool = reals + poss + units + ([f"({i} * {round(float(rng.uniform(0.5, 2.0)), 2)})" for i in ints] if rng.random() < 0.3 else [])
        if kind == "normal":
            lines.append(f"    {name} = normal({_expr(rng, pool, params)}, {_pos(rng, pool, poss, params)}) @ '{name}'")
            reals.append(name)
        elif kind == "gamma":
            lines.append(f"    {name} = gamma({_pos(rng, pool, poss, params)}, {_pos(rng, pool, poss, params)}) @ '{name}'")
            poss.append(name)
        elif kind == "beta":
            lines.append(f"    {name} = beta({_pos(rng, pool, poss, params)}, {_pos(rng, pool, poss, params)}) @ '{name}'")
            units.append(name)
        else:
            p = str(rng.choice(units)) if units and rng.random() < 0.5 else repr(round(float(rng.uniform(0.1, 0.9)), 3))
            lines.append(f"    {name} = flip({p}) @ '{name}'")
            ints.append(name)
        sites.append((name, kind))
    ret = _expr(rng, reals + poss + units, params) if (reals + poss + units) else "0.0"
    pre = "".join(globals().get("_HELPER_SRC", {})[h] for h in helpers)
    src = pre + "def model(s, t):\n" + "\n".join(lines) + f"\n    return {ret}, {sites[-1][0]}\n"
    return src, sites


def random_constraint(rng, sites, n, dev):
    chm = C.n()
    for name, kind in sites:
        if kind == "call":  # (a nested call: sometimes constrain its first site)
            if rng.random() < 0.3:
                chm = chm | C[name, "p"].set(float(rng.uniform(-1, 1)))
            continue
        if rng.random() < 0.35:
            if kind == "flip":
                v = bool(rng.integers(2))
            else:
                v = {"normal": float(rng.uniform(-2, 2)), "gamma": float(rng.uniform(0.2, 3)), "beta": float(rng.uniform(0.1, 0.9))}[kind]
                if rng.random() < 0.3:  # a per-particle column
                    lo, hi = {"normal": (-2, 2), "gamma": (0.2, 3), "beta": (0.1, 0.9)}[kind]
                    v = torch.linspace(lo, hi, n).to(dev)
            chm = chm | C[name].set(v)
    return chm


def _same(a, b):
    if isinstance(a, (tuple, list)):
        return all(_same(x, y) for x, y in zip(a, b))
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    if a.dtype != b.dtype and not (a.dtype.is_floating_point and b.dtype.is_floating_point):
        b = b.to(a.dtype)
    a, b = torch.broadcast_tensors(a.cpu(), b.cpu().to(a.dtype))
    return torch.equal(a, b) or (a.dtype.is_floating_point and torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(), b.nan_to_num()))


def run(seconds: float, seed: int, impl: int = 1, n: int = 1500, min_compared: int = 0):
    rng = np.random.default_rng(seed)
    dev = get_ops().device()
    # (`min_compared`: on a loaded machine the budget of seconds alone may cover only a handful of bodies — keep going, up to
    # ten budgets, until that many have been compared)
    t_end, compared, skipped = time.time() + seconds, 0, 0
    t_cap = time.time() + 10.0 * seconds
    while time.time() < t_end or (compared < min_compared and time.time() < t_cap):
        src, sites = random_model(rng)
        ns = {"normal": normal, "gamma": gamma, "beta": beta, "flip": flip, "torch": torch, "gen": gen}
        exec(src, ns)  # noqa: S102 - generated by random_model above
        model = gen(ns["model"])
        keys = genjax.random.split(genjax.random.key(int(rng.integers(1 << 30)), impl), n)
        chm = random_constraint(rng, sites, n, dev)
        args = (round(float(rng.uniform(0.3, 2.0)), 3), round(float(rng.uniform(-1.5, 1.5)), 3))
        fused = try_fused_generate(model, keys, chm, args)
        if fused is None:
            skipped += 1
            continue
        ftr, fw = fused
        h = GenerateHandler(keys, chm)
        retval = h.run(model.source, args)
        eager = StaticTrace(model, args, retval, h.traces)
        ctx = f"\n{src}\nconstraint {dict(chm.leaves()).keys()} args {args} seed {seed}"
        ew = h.weight
        assert _same(fw, ew if isinstance(ew, torch.Tensor) else torch.zeros(n) + ew), "weights differ" + ctx
        assert _same(ftr.get_score(), eager.get_score()), "scores differ" + ctx
        fc, ec = dict(ftr.get_choices().leaves()), dict(eager.get_choices().leaves())
        assert fc.keys() == ec.keys(), "addresses differ" + ctx
        for k in fc:
            assert _same(fc[k], ec[k]), f"choice {k} differs" + ctx
        assert _same(ftr.get_retval(), eager.get_retval()), "return values differ" + ctx
        compared += 1
    return compared, skipped


# ---- random scan kernels: the one-launch scan (gjx_scan_run) against the host loop of per-site launches ------------------
def random_step(rng):
    lines, reals, poss, units, sites = [], ["x", "v"], [], [], []
    params = ["u"]
    for q in range(int(rng.integers(2, 6))):
        name = f"s{q}"
        kind = rng.choice(["normal", "normal", "gamma", "beta"])
        pool = reals + poss + units
        if kind == "normal":
            lines.append(f"    {name} = normal({_expr(rng, pool, params)}, {_pos(rng, pool, poss, [])}) @ '{name}'")
            reals.append(name)
        elif kind == "gamma":
            lines.append(f"    {name} = gamma({_pos(rng, pool, poss, [])}, {_pos(rng, pool, poss, [])}) @ '{name}'")
            poss.append(name)
        else:
            lines.append(f"    {name} = beta({_pos(rng, pool, poss, [])}, {_pos(rng, pool, poss, [])}) @ '{name}'")
            units.append(name)
        sites.append((name, kind))
    new = [n for n, _ in sites]
    cx, cv = str(rng.choice(new)), str(rng.choice(new + ["v"]))
    if rng.random() < 0.4:
        cv = f"({cv} * {round(float(rng.uniform(0.5, 1.5)), 2)})"
    if rng.random() < 0.4:  # a deterministic update: the next carry is an expression over the carry, a site and the input
        cx = f"(x + {round(float(rng.uniform(0.1, 0.9)), 2)} * {cx} - u * {round(float(rng.uniform(0.0, 0.3)), 2)})"
    y = _expr(rng, new + (["x"] if rng.random() < 0.4 else []), params)  # (sometimes y_t reads the carry it was given)
    src = ("def step(carry, u):\n    x, v = carry\n" + "\n".join(lines) + f"\n    return ({cx}, {cv}), ({y}, {new[0]})\n")
    return src, sites


def run_scans(seconds: float, seed: int, impl: int = 1, n: int = 800, min_compared: int = 0):
    from genjax._amd import combinators as CB

    rng = np.random.default_rng(seed)
    dev = get_ops().device()
    # (`min_compared`: on a loaded machine the budget of seconds alone may cover only a handful of bodies — keep going, up to
    # ten budgets, until that many have been compared)
    t_end, compared, skipped = time.time() + seconds, 0, 0
    t_cap = time.time() + 10.0 * seconds
    while time.time() < t_end or (compared < min_compared and time.time() < t_cap):
        src, sites = random_step(rng)
        ns = {"normal": normal, "gamma": gamma, "beta": beta, "flip": flip, "torch": torch}
        exec(src, ns)  # noqa: S102 - generated by random_step above
        T = int(rng.integers(1, 7))
        model = gen(ns["step"]).scan()
        us = torch.linspace(0.2, 1.4, T).to(dev)
        keys = genjax.random.split(genjax.random.key(int(rng.integers(1 << 30)), impl), n)
        chm = C.n()
        for name, kind in sites:
            if rng.random() < 0.4:
                lo, hi = {"normal": (-2, 2), "gamma": (0.2, 3), "beta": (0.1, 0.9)}[kind]
                chm = chm | C[name].set(torch.linspace(lo, hi, T))
        x0 = torch.linspace(-1, 1, n).to(dev) if rng.random() < 0.5 else 0.25
        args = ((x0, -0.5), us)
        out = {}
        for fused in (True, False):
            CB.FUSED_SCAN = fused
            try:
                tr, w = model.generate(keys, chm, args)
            finally:
                CB.FUSED_SCAN = True
            out[fused] = (tr, w)
        if not isinstance(out[True][0], CB.FusedScanTrace):
            skipped += 1
            continue
        (ta, wa), (tb, wb) = out[True], out[False]
        ctx = f"\n{src}\nT {T} constraint {list(dict(chm.leaves()).keys())} seed {seed}"
        wb = wb if isinstance(wb, torch.Tensor) else torch.zeros(n) + wb
        assert _same(wa, wb), "weights differ" + ctx
        assert _same(ta.get_score(), tb.get_score()), "scores differ" + ctx
        ca, cb = dict(ta.get_choices().leaves()), dict(tb.get_choices().leaves())
        assert ca.keys() == cb.keys(), "addresses differ" + ctx
        for k in ca:
            assert _same(ca[k], cb[k]), f"choice {k} differs" + ctx
        assert _same(ta.get_retval(), tb.get_retval()), "return values differ" + ctx
        compared += 1
    return compared, skipped
