"""Lowering a `@gen` body to ONE fused kernel (`gjx_importance_run`).

The body is run once with symbolic site values.  If every `@` site is a supported distribution and
every distribution argument is (a) a constant, (b) `c * v`, `v + d`, `c * v + d`, `v - d` or `-v`
of ONE earlier site value or constraint column, or (c) `table[v]` of an integer-valued earlier
site, the walk of static.py:340-399 becomes a site table for the HIP interpreter kernel.  Anything
else (nested generative functions, non-affine arithmetic, data-dependent Python control flow)
raises `PlanUnsupported` and the caller falls back to the per-site column kernels — which compute
the same numbers bit for bit, only with one launch per site.

Restricting the affine forms to at most one multiply and one add is deliberate: that is exactly
the f32 operation sequence the column path would execute for the same expression, so both routes
round identically.

Structure versus data.  Scalar observations and scalar floating-point model arguments are lowered to
LAUNCH-UNIFORM PARAMETERS (`GJX_ARG_PARAM`, `gjx_plan_set_params`), not to constants: the specialised kernel
of a model is keyed by its structure alone, so the same model on another dataset (other observed values,
other hyper-parameters) reuses the compiled kernel instead of paying a hiprtc compilation.  A body that does
more with an argument than the affine forms above (`sigma ** 2`, `if mu > 0`) is traced a second time with
its arguments as plain constants — same numbers, one compilation per distinct value.
"""

from __future__ import annotations

import threading

import torch

from . import abi
from .choicemap import ChoiceMap
from .lang import (_spec_wrap, Distribution, ParticleKeys, StaticTrace, ValueTrace, _Handler, normal, gamma, beta, flip,
                   bernoulli, categorical, Categorical)
from .runtime import get_ops

MAX_INPUT_COLS = 16


class PlanUnsupported(Exception):
    pass


class Sym:
    """Symbolic f32 value: scale * source + offset, source = ("site", idx) | ("input", idx)."""

    __slots__ = ("src", "scale", "offset", "has_mul", "has_add", "is_int", "tracer")

    def __init__(self, tracer, src, scale=1.0, offset=0.0, has_mul=False, has_add=False, is_int=False):
        self.tracer, self.src, self.scale, self.offset = tracer, src, scale, offset
        self.has_mul, self.has_add, self.is_int = has_mul, has_add, is_int

    # -- affine algebra (one multiply, then one add — in that order) -------------------------------
    @staticmethod
    def _const(c):
        if isinstance(c, (bool, int, float)):
            return float(c)
        if isinstance(c, torch.Tensor) and c.dim() == 0:
            return float(c)
        raise PlanUnsupported("non-constant operand")

    @staticmethod
    def _is_const(c) -> bool:
        return isinstance(c, (bool, int, float)) or (isinstance(c, torch.Tensor) and c.dim() == 0)

    def __mul__(self, c):
        if not self._is_const(c) or self.has_mul or self.has_add:  # beyond one affine step: a postfix program
            return SymExpr.binop(abi.EXPR_MUL, self, c)
        c = self._const(c)
        return Sym(self.tracer, self.src, float(torch.tensor(c, dtype=torch.float32)), 0.0, True, False, False)

    def __rmul__(self, c):
        if not self._is_const(c) or self.has_mul or self.has_add:
            return SymExpr.binop(abi.EXPR_MUL, c, self)
        return self.__mul__(c)

    def __add__(self, c):
        if not self._is_const(c) or self.has_add:
            return SymExpr.binop(abi.EXPR_ADD, self, c)
        c = self._const(c)
        return Sym(self.tracer, self.src, self.scale, float(torch.tensor(c, dtype=torch.float32)), self.has_mul,
                   True, False)

    def __radd__(self, c):
        if not self._is_const(c) or self.has_add:
            return SymExpr.binop(abi.EXPR_ADD, c, self)
        return self.__add__(c)

    def __sub__(self, c):
        if not self._is_const(c) or self.has_add:
            return SymExpr.binop(abi.EXPR_SUB, self, c)
        return self.__add__(-self._const(c))

    def __rsub__(self, c):
        return SymExpr.binop(abi.EXPR_SUB, c, self)

    def __neg__(self):
        if self.has_mul or self.has_add:
            return SymExpr.unop(abi.EXPR_NEG, self)
        return self.__mul__(-1.0)

    def _no(self, *a, **k):
        raise PlanUnsupported("unsupported operation on a traced site value")

    def __truediv__(self, c):  # (x / c is NOT x * (1 / c): an IEEE division — lang.SpecTensor does the same per site)
        return SymExpr.binop(abi.EXPR_DIV, self, c)

    def __rtruediv__(self, c):
        return SymExpr.binop(abi.EXPR_DIV, c, self)

    def exp(self):
        return SymExpr.unop(abi.EXPR_EXP, self)

    def log(self):
        return SymExpr.unop(abi.EXPR_LOG, self)

    def sqrt(self):
        return SymExpr.unop(abi.EXPR_SQRT, self)

    def abs(self):
        return SymExpr.unop(abi.EXPR_ABS, self)

    __abs__ = abs

    def sigmoid(self):
        return _sym_sigmoid(self)

    def reciprocal(self):
        return SymExpr.binop(abi.EXPR_DIV, 1.0, self)

    def square(self):
        return SymExpr.binop(abi.EXPR_MUL, self, self)

    def __pow__(self, e):
        return _sym_pow(self, e)

    __rpow__ = __bool__ = __float__ = __int__ = _no
    __index__ = __array__ = __len__ = __iter__ = _no

    # comparisons give conditions (0 / 1 values of the program: `torch.where(x > 0.5, a, b)`); `==` stays Python's identity
    def __lt__(self, o): return SymExpr.binop(abi.EXPR_LT, self, o)  # noqa: E704
    def __le__(self, o): return SymExpr.binop(abi.EXPR_LE, self, o)  # noqa: E704
    def __gt__(self, o): return SymExpr.binop(abi.EXPR_LT, o, self)  # noqa: E704
    def __ge__(self, o): return SymExpr.binop(abi.EXPR_LE, o, self)  # noqa: E704
    def __and__(self, o): return _logical(abi.EXPR_MUL, self, o)  # noqa: E704
    def __or__(self, o): return _logical(abi.EXPR_MAX, self, o)  # noqa: E704
    def __invert__(self): return _logical(abi.EXPR_SUB, 1.0, self)  # noqa: E704
    __rand__, __ror__ = __and__, __or__

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        r = _sym_torch_function(func, args, kwargs)
        if r is not NotImplemented:
            return r
        # `table[sym]`: a constant tensor indexed by a traced integer-valued value — a 1-D table gives a distribution
        # argument (`means[idx]`), a 2-D one the logits / probs ROW of a categorical site (`trans[z]`)
        if func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[0], torch.Tensor) \
                and not isinstance(args[0], Sym) and isinstance(args[1], Sym):
            table, idx = args
            if idx.has_mul or idx.has_add:
                raise PlanUnsupported("table index must be a plain traced value")
            if table.dim() == 1:
                return TableProxy(table)[idx]
            if table.dim() == 2:
                return _Rows(idx.tracer, table, idx)
        raise PlanUnsupported(f"torch.{getattr(func, '__name__', func)} on a traced site value")


class _Rows:
    """`table2d[sym]`: row `sym` of a constant [rows, K] tensor — the parameters of a categorical site."""

    def __init__(self, tracer, table: torch.Tensor, idx: Sym):
        self.tracer, self.table, self.idx = tracer, table, idx


class SymExpr:
    """A traced f32 value beyond one affine step — `w * x + b` over two sites, `(z - m) * (z - m)`, ... — as the postfix
    program of `abi.ARG_EXPR`: operands push a value, `+ - * /` pop two, unary minus one; every operator is one f32 rounding,
    in the order the model body wrote it (what the per-site column path computes with torch's f32 tensor arithmetic)."""

    __slots__ = ("tracer", "prog", "cond")

    def __init__(self, tracer, prog):
        self.tracer, self.prog = tracer, prog
        self.cond = bool(prog) and prog[-1][0] in (abi.EXPR_LT, abi.EXPR_LE, abi.EXPR_EQ)  # a 0 / 1 value (see _logical)
        if len(prog) > abi.MAX_EXPR_OPS:
            raise PlanUnsupported("expression too long for a plan argument")
        depth = deepest = 0
        for op, _, _ in prog:
            depth += 1 if op <= abi.EXPR_OBS else (0 if op in abi.EXPR_UNARY else (-2 if op == abi.EXPR_SELECT else -1))
            deepest = max(deepest, depth)
        if deepest > abi.MAX_EXPR_DEPTH:
            raise PlanUnsupported("expression too deep for a plan argument")

    _LEAF = {"site": abi.EXPR_SITE, "input": abi.EXPR_INPUT, "state": abi.EXPR_STATE, "obs": abi.EXPR_OBS}

    @staticmethod
    def _f32(c) -> float:
        return float(torch.tensor(float(c), dtype=torch.float32))

    @classmethod
    def program_of(cls, x, tracer):
        """Operand -> (tracer, program).  Traced values keep their own rounding steps (a `Sym` is its source, then its
        multiply, then its add); numbers become f32 literals; launch parameters stay parameters."""
        if isinstance(x, SymExpr):
            return x.tracer, list(x.prog)
        if isinstance(x, Sym):
            prog = [(cls._LEAF[x.src[0]], x.src[1], 0.0)]
            if x.has_mul:
                prog += [(abi.EXPR_CONST, 0, x.scale), (abi.EXPR_MUL, 0, 0.0)]
            if x.has_add:
                prog += [(abi.EXPR_CONST, 0, x.offset), (abi.EXPR_ADD, 0, 0.0)]
            return x.tracer, prog
        if isinstance(x, ParamVal):
            if tracer is None or not getattr(tracer, "use_params", False):
                raise PlanUnsupported("a launch parameter outside an importance plan")
            return tracer, [(abi.EXPR_PARAM, tracer.param_slot(x), 0.0)]
        if Sym._is_const(x):
            return tracer, [(abi.EXPR_CONST, 0, cls._f32(x))]
        raise PlanUnsupported("unsupported operand in an expression over traced site values")

    @classmethod
    def binop(cls, op, a, b):
        # (`/` is an IEEE division whatever its operands: torch divides a device tensor by a NUMBER as a multiplication by
        # the number's reciprocal, so the per-site path routes that case through gjx_map_f32 — lang.SpecTensor)
        tr = next((v.tracer for v in (a, b) if isinstance(v, (Sym, SymExpr))), None)
        _, pa = cls.program_of(a, tr)
        _, pb = cls.program_of(b, tr)
        return cls(tr, pa + pb + [(op, 0, 0.0)])

    @classmethod
    def unop(cls, op, a):
        tr, pa = cls.program_of(a, getattr(a, "tracer", None))
        return cls(tr, pa + [(op, 0, 0.0)])

    @classmethod
    def select(cls, c, t, f):
        """`where(c, t, f)`: c a traced condition (a comparison, a flip value); a constant condition selects on the host."""
        if not isinstance(c, (Sym, SymExpr)):
            if isinstance(c, (bool, int)) or (isinstance(c, torch.Tensor) and c.dim() == 0):
                return t if bool(c) else f
            raise PlanUnsupported("where() with a condition that is neither traced nor a scalar")
        tr = c.tracer
        _, pc = cls.program_of(c, tr)
        _, pt = cls.program_of(t, tr)
        _, pf = cls.program_of(f, tr)
        return cls(tr, pc + pt + pf + [(abi.EXPR_SELECT, 0, 0.0)])

    def __add__(self, o): return SymExpr.binop(abi.EXPR_ADD, self, o)  # noqa: E704
    def __radd__(self, o): return SymExpr.binop(abi.EXPR_ADD, o, self)  # noqa: E704
    def __sub__(self, o): return SymExpr.binop(abi.EXPR_SUB, self, o)  # noqa: E704
    def __rsub__(self, o): return SymExpr.binop(abi.EXPR_SUB, o, self)  # noqa: E704
    def __mul__(self, o): return SymExpr.binop(abi.EXPR_MUL, self, o)  # noqa: E704
    def __rmul__(self, o): return SymExpr.binop(abi.EXPR_MUL, o, self)  # noqa: E704
    def __neg__(self): return SymExpr.unop(abi.EXPR_NEG, self)  # noqa: E704
    def __truediv__(self, o): return SymExpr.binop(abi.EXPR_DIV, self, o)  # noqa: E704
    def __rtruediv__(self, o): return SymExpr.binop(abi.EXPR_DIV, o, self)  # noqa: E704
    def exp(self): return SymExpr.unop(abi.EXPR_EXP, self)  # noqa: E704
    def log(self): return SymExpr.unop(abi.EXPR_LOG, self)  # noqa: E704
    def sqrt(self): return SymExpr.unop(abi.EXPR_SQRT, self)  # noqa: E704
    def abs(self): return SymExpr.unop(abi.EXPR_ABS, self)  # noqa: E704
    __abs__ = abs
    def sigmoid(self): return _sym_sigmoid(self)  # noqa: E704
    def reciprocal(self): return SymExpr.binop(abi.EXPR_DIV, 1.0, self)  # noqa: E704
    def square(self): return SymExpr.binop(abi.EXPR_MUL, self, self)  # noqa: E704

    def _no(self, *a, **k):
        raise PlanUnsupported("unsupported operation on a traced expression")

    def __pow__(self, e):
        return _sym_pow(self, e)

    __rpow__ = __bool__ = __float__ = __int__ = _no
    __index__ = __array__ = __len__ = __iter__ = _no

    def __lt__(self, o): return SymExpr.binop(abi.EXPR_LT, self, o)  # noqa: E704
    def __le__(self, o): return SymExpr.binop(abi.EXPR_LE, self, o)  # noqa: E704
    def __gt__(self, o): return SymExpr.binop(abi.EXPR_LT, o, self)  # noqa: E704
    def __ge__(self, o): return SymExpr.binop(abi.EXPR_LE, o, self)  # noqa: E704
    def __and__(self, o): return _logical(abi.EXPR_MUL, self, o)  # noqa: E704
    def __or__(self, o): return _logical(abi.EXPR_MAX, self, o)  # noqa: E704
    def __invert__(self): return _logical(abi.EXPR_SUB, 1.0, self)  # noqa: E704
    __rand__, __ror__ = __and__, __or__

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        r = _sym_torch_function(func, args, kwargs)
        if r is not NotImplemented:
            return r
        raise PlanUnsupported(f"torch.{getattr(func, '__name__', func)} on a traced expression")

    def evaluate(self, leaf):
        """The program over concrete operands: `leaf(kind, ref)` -> tensor / number.  Same f32 steps as the kernel."""
        st = []
        for op, ref, val in self.prog:
            if op == abi.EXPR_CONST:
                st.append(torch.tensor(val, dtype=torch.float32))
            elif op <= abi.EXPR_OBS:
                v = leaf(op, ref)
                v = v.to(torch.float32) if isinstance(v, torch.Tensor) else torch.tensor(float(v), dtype=torch.float32)
                st.append(v)
            elif op == abi.EXPR_NEG:
                st[-1] = -st[-1]
            elif op in _UNARY_MAP:
                st[-1] = get_ops().map_f32(_UNARY_MAP[op], st[-1])
            else:
                b = st.pop()
                a = st.pop()
                if op == abi.EXPR_DIV and (a.dim() == 0 or b.dim() == 0):  # a NUMBER on either side: the IEEE division
                    if b.dim() == 0:
                        st.append(get_ops().map_f32(abi.MAP_DIV, a, float(b)))
                    else:
                        st.append(get_ops().map_f32(abi.MAP_RDIV, b, float(a)))
                    continue
                if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.device != b.device:
                    a, b = (a.to(b.device), b) if a.dim() == 0 else (a, b.to(a.device))
                if op == abi.EXPR_SELECT:  # (b is f, a is t: the condition lies below them)
                    c = st.pop()
                    dev = next((v.device for v in (c, a, b) if v.dim()), c.device)  # (0-dim constants follow the columns)
                    c, a, b = torch.broadcast_tensors(c.to(dev), a.to(dev), b.to(dev))
                    st.append(torch.where(c != 0, a, b))
                    continue
                if op in (abi.EXPR_LT, abi.EXPR_LE, abi.EXPR_EQ):
                    if a.device != b.device:
                        a, b = (a.to(b.device), b) if a.dim() == 0 else (a, b.to(a.device))
                    st.append((a < b if op == abi.EXPR_LT else (a <= b if op == abi.EXPR_LE else a == b)).to(torch.float32))
                    continue
                if op in (abi.EXPR_MAX, abi.EXPR_MIN):  # (torch's own maximum / minimum: exact, a NaN if either is one)
                    a, b = torch.broadcast_tensors(a, b)
                    st.append(torch.maximum(a, b) if op == abi.EXPR_MAX else torch.minimum(a, b))
                    continue
                st.append(a + b if op == abi.EXPR_ADD else (a - b if op == abi.EXPR_SUB else (a * b if op == abi.EXPR_MUL else a / b)))
        return st[0]


_UNARY_MAP = {abi.EXPR_EXP: abi.MAP_EXP, abi.EXPR_LOG: abi.MAP_LOG, abi.EXPR_SQRT: abi.MAP_SQRT, abi.EXPR_ABS: abi.MAP_ABS}
_UNARY_FUNCS = {torch.exp: abi.EXPR_EXP, torch.Tensor.exp: abi.EXPR_EXP, torch.log: abi.EXPR_LOG, torch.Tensor.log: abi.EXPR_LOG,
                torch.sqrt: abi.EXPR_SQRT, torch.Tensor.sqrt: abi.EXPR_SQRT, torch.abs: abi.EXPR_ABS, torch.Tensor.abs: abi.EXPR_ABS,
                torch.absolute: abi.EXPR_ABS}
_DIV_FUNCS = (torch.div, torch.true_divide, torch.Tensor.div, torch.Tensor.true_divide)


_T = torch.Tensor
_ARITH_FUNCS = {torch.add: (abi.EXPR_ADD, False), _T.add: (abi.EXPR_ADD, False), _T.__add__: (abi.EXPR_ADD, False),
                _T.__radd__: (abi.EXPR_ADD, True), torch.sub: (abi.EXPR_SUB, False), _T.sub: (abi.EXPR_SUB, False),
                _T.__sub__: (abi.EXPR_SUB, False), _T.__rsub__: (abi.EXPR_SUB, True), torch.mul: (abi.EXPR_MUL, False),
                _T.mul: (abi.EXPR_MUL, False), _T.__mul__: (abi.EXPR_MUL, False), _T.__rmul__: (abi.EXPR_MUL, True),
                torch.multiply: (abi.EXPR_MUL, False), torch.subtract: (abi.EXPR_SUB, False),
                _T.__truediv__: (abi.EXPR_DIV, False), _T.__rtruediv__: (abi.EXPR_DIV, True)}
_CMP_FUNCS = {torch.lt: (abi.EXPR_LT, False), torch.Tensor.lt: (abi.EXPR_LT, False), torch.le: (abi.EXPR_LE, False),
              torch.Tensor.le: (abi.EXPR_LE, False), torch.gt: (abi.EXPR_LT, True), torch.Tensor.gt: (abi.EXPR_LT, True),
              torch.ge: (abi.EXPR_LE, True), torch.Tensor.ge: (abi.EXPR_LE, True), torch.eq: (abi.EXPR_EQ, False),
              torch.Tensor.eq: (abi.EXPR_EQ, False)}


def _sym_pow(x, e):
    """x ** 2 and x ** 3 are products (what torch's pow computes for these exponents, on every backend); other exponents go
    through the device library's pow on the per-site path: not lowered."""
    if isinstance(e, (int, float)) and not isinstance(e, bool) and float(e) in (2.0, 3.0):
        sq = SymExpr.binop(abi.EXPR_MUL, x, x)
        return sq if float(e) == 2.0 else SymExpr.binop(abi.EXPR_MUL, sq, x)
    raise PlanUnsupported("a power other than 2 or 3 of a traced value")


def _is_condition(v) -> bool:
    """A 0 / 1 value: a comparison (or a combination of conditions), a flip site, a boolean constant."""
    if isinstance(v, SymExpr):
        return v.cond
    if isinstance(v, Sym):
        m = v.tracer.meta[v.src[1]] if v.src[0] == "site" and not (v.has_mul or v.has_add) else None
        return m is not None and m["dtype"] == torch.bool
    return isinstance(v, bool) or (isinstance(v, torch.Tensor) and v.dim() == 0 and v.dtype == torch.bool)


def _logical(op, a, b):
    """and = product, or = maximum, not = 1 - c — of CONDITIONS only (`&` on other integers is bitwise: not lowered)."""
    if not all(_is_condition(v) for v in (a, b) if not (isinstance(v, float) and op == abi.EXPR_SUB)):
        raise PlanUnsupported("& | ~ on values that are not conditions")
    r = SymExpr.binop(op, a, b)
    r.cond = True
    return r


def _sym_sigmoid(x):
    """sigmoid(x) = 1 / (1 + exp(-x)) in the spec's steps (one rounding each): what lang.SpecTensor computes per site."""
    return SymExpr.binop(abi.EXPR_DIV, 1.0, SymExpr.binop(abi.EXPR_ADD, SymExpr.unop(abi.EXPR_EXP, SymExpr.unop(abi.EXPR_NEG, x)), 1.0))


def _sym_torch_function(func, args, kwargs):
    """torch functions of traced values that lower to the expression program (NotImplemented: not one of them)."""
    if kwargs:
        if func in (torch.clamp, torch.Tensor.clamp, torch.clip, torch.Tensor.clip) and len(args) == 1 and set(kwargs) <= {"min", "max"}:
            return _sym_torch_function(func, (args[0], kwargs.get("min"), kwargs.get("max")), None)
        return NotImplemented
    if func in _UNARY_FUNCS and len(args) == 1:
        return SymExpr.unop(_UNARY_FUNCS[func], args[0])
    if func in _DIV_FUNCS and len(args) == 2:
        return SymExpr.binop(abi.EXPR_DIV, args[0], args[1])
    if func in _ARITH_FUNCS and len(args) == 2:  # (a constant TENSOR on the left: `tensor + traced` arrives here)
        op, swap = _ARITH_FUNCS[func]
        return SymExpr.binop(op, args[1], args[0]) if swap else SymExpr.binop(op, args[0], args[1])
    if func in (torch.neg, torch.Tensor.neg, torch.negative, torch.Tensor.__neg__) and len(args) == 1:
        return -args[0]
    if func in (torch.sigmoid, torch.Tensor.sigmoid, torch.nn.functional.sigmoid) and len(args) == 1:
        return _sym_sigmoid(args[0])
    if func in (torch.reciprocal, torch.Tensor.reciprocal) and len(args) == 1:
        return SymExpr.binop(abi.EXPR_DIV, 1.0, args[0])
    if func in (torch.square, torch.Tensor.square) and len(args) == 1:
        return SymExpr.binop(abi.EXPR_MUL, args[0], args[0])
    if func in (torch.pow, torch.Tensor.pow, torch.Tensor.__pow__) and len(args) == 2 and isinstance(args[0], (Sym, SymExpr)):
        return _sym_pow(args[0], args[1])
    if func in (torch.maximum, torch.Tensor.maximum, torch.minimum, torch.Tensor.minimum) and len(args) == 2:
        return SymExpr.binop(abi.EXPR_MAX if func in (torch.maximum, torch.Tensor.maximum) else abi.EXPR_MIN, args[0], args[1])
    if func in (torch.clamp, torch.Tensor.clamp, torch.clip, torch.Tensor.clip) and 1 <= len(args) <= 3:
        x = args[0]
        lo = args[1] if len(args) > 1 else None
        hi = args[2] if len(args) > 2 else None
        if lo is not None:
            x = SymExpr.binop(abi.EXPR_MAX, x, lo)
        if hi is not None:
            x = SymExpr.binop(abi.EXPR_MIN, x, hi)
        return x
    if func in _CMP_FUNCS and len(args) == 2:
        op, swap = _CMP_FUNCS[func]
        return SymExpr.binop(op, args[1], args[0]) if swap else SymExpr.binop(op, args[0], args[1])
    if func in (torch.ne, torch.Tensor.ne) and len(args) == 2:
        return _logical(abi.EXPR_SUB, 1.0, SymExpr.binop(abi.EXPR_EQ, args[0], args[1]))
    if func in (torch.where, torch.Tensor.where) and len(args) == 3:  # jnp.where(cond, t, f)
        return SymExpr.select(args[0], args[1], args[2])
    if func in (torch.logical_and, _T.logical_and, torch.bitwise_and, _T.bitwise_and, _T.__and__, _T.__rand__) and len(args) == 2:
        return _logical(abi.EXPR_MUL, args[0], args[1])
    if func in (torch.logical_or, _T.logical_or, torch.bitwise_or, _T.bitwise_or, _T.__or__, _T.__ror__) and len(args) == 2:
        return _logical(abi.EXPR_MAX, args[0], args[1])
    if func in (torch.logical_not, _T.logical_not, torch.bitwise_not, _T.bitwise_not, _T.__invert__) and len(args) == 1:
        return _logical(abi.EXPR_SUB, 1.0, args[0])
    if func is torch.nn.functional.softplus and len(args) == 1:
        # max(x, 0) + log(1 + exp(-|x|)): no overflow for large x (what lang._spec_softplus computes per site)
        x = args[0]
        t = SymExpr.unop(abi.EXPR_LOG, SymExpr.binop(abi.EXPR_ADD, SymExpr.unop(abi.EXPR_EXP, SymExpr.unop(abi.EXPR_NEG, SymExpr.unop(abi.EXPR_ABS, x))), 1.0))
        return SymExpr.binop(abi.EXPR_ADD, SymExpr.binop(abi.EXPR_MAX, x, 0.0), t)
    return NotImplemented


class ParamVal:
    """A launch-uniform scalar (an observation, a scalar model argument, or host arithmetic on those): it carries the
    concrete value and does its arithmetic on the host with the operands' own types — exactly what the per-site column
    path would compute — but reaches the kernel as a PARAMETER, so its value is not part of the kernel's source.
    Anything that would turn it into structure (a comparison, `float()`, an index, mixing with a traced site value)
    raises `PlanUnsupported`; the body is then traced again with plain constants."""

    __slots__ = ("value", "slot")

    def __init__(self, value):
        # (a tensor does its host arithmetic as a site value does on the per-site path: lang.SpecTensor)
        self.value, self.slot = _spec_wrap(value), None

    @staticmethod
    def _v(x):
        if isinstance(x, ParamVal):
            return x.value
        if isinstance(x, (bool, int, float)) or (isinstance(x, torch.Tensor) and x.dim() == 0):
            return x
        return None

    def _bin(op, swap=False):  # noqa: N805
        def f(self, other):
            o = ParamVal._v(other)
            if o is None:
                return NotImplemented  # (a traced value: its reflected method builds the expression)
            return ParamVal(op(o, self.value) if swap else op(self.value, o))

        return f

    __add__, __radd__ = _bin(lambda a, b: a + b), _bin(lambda a, b: a + b, True)
    __sub__, __rsub__ = _bin(lambda a, b: a - b), _bin(lambda a, b: a - b, True)
    __mul__, __rmul__ = _bin(lambda a, b: a * b), _bin(lambda a, b: a * b, True)
    __truediv__, __rtruediv__ = _bin(lambda a, b: a / b), _bin(lambda a, b: a / b, True)
    __pow__, __rpow__ = _bin(lambda a, b: a ** b), _bin(lambda a, b: a ** b, True)
    del _bin

    def __neg__(self):
        return ParamVal(-self.value)

    def __abs__(self):
        return ParamVal(abs(self.value))

    def _no(self, *a, **k):
        raise PlanUnsupported("a launch parameter used as structure")

    __bool__ = __float__ = __int__ = __index__ = __lt__ = __le__ = __gt__ = __ge__ = __len__ = __iter__ = _no
    __array__ = __floordiv__ = __rfloordiv__ = __mod__ = __rmod__ = _no
    __hash__ = object.__hash__

    def __eq__(self, other):
        raise PlanUnsupported("a launch parameter used as structure")

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        raise PlanUnsupported(f"torch.{getattr(func, '__name__', func)} on a launch parameter")


class _Table:
    """`table[sym]` with a 1-D constant table and an integer-valued site."""

    def __init__(self, tracer, table: torch.Tensor, idx: Sym):
        self.tracer, self.table, self.idx = tracer, table, idx


_DIST_IDS = {id(normal): abi.DIST_NORMAL, id(gamma): abi.DIST_GAMMA, id(beta): abi.DIST_BETA,
             id(flip): abi.DIST_BERNOULLI}


class PlanTracer(_Handler):
    def __init__(self, constraint: ChoiceMap, n: int, use_params: bool = True):
        super().__init__()
        self.constraint, self.n, self.use_params = constraint, n, use_params
        self.params: list[float] = []  # launch-uniform parameter values (observations, model arguments)
        self.sites: list[abi.Site] = []
        self.meta: list[dict] = []  # per site: addr, gen_fn, args (symbolic), out_col, observed
        self.inputs: list[torch.Tensor] = []
        self.input_orig: dict = {}  # input column -> the constrained value it was made from (its own dtype)
        self.keep: list = []  # device tensors (and expression programs) the site table points into
        self.expr_progs: dict = {}  # address of a program's ctypes array -> the program
        self.n_out = 0
        # nested `@gen` calls (static.py:175-193): the callee's body is traced in line; the site table stays flat and
        # `scopes` says which range of it each call produced (gjx.h gjx_scope)
        self.scopes: list[list] = []  # [parent scope, begin, end]
        self.items: list = []         # this body's `@` sites in program order: ("site", index) | ("call", record)
        self.cur_scope, self.depth, self.allow_scopes = 0, 0, True
        self.prefix: tuple = ()  # the addresses of the calls the body being traced sits in

    # -- argument encoding ---------------------------------------------------------------------------
    def param_slot(self, pv: ParamVal) -> int:
        """The parameter slot of `pv` (allocated on first use as a distribution argument / observed value)."""
        if pv.slot is None:
            if len(self.params) >= abi.MAX_PARAMS:
                raise PlanUnsupported("too many launch parameters")
            v = pv.value
            self.params.append(float(torch.as_tensor(float(v) if not isinstance(v, torch.Tensor) else v,
                                                     dtype=torch.float32)))
            pv.slot = len(self.params) - 1
        return pv.slot

    def wrap_args(self, args):
        """Top-level scalar floating-point model arguments become parameters (ints and bools stay Python values: they
        are structure — loop bounds, switches)."""
        if not self.use_params:
            return tuple(args)
        return tuple(ParamVal(a) if isinstance(a, float) or (isinstance(a, torch.Tensor) and a.dim() == 0
                                                               and a.is_floating_point()) else a for a in args)

    def _arg(self, v) -> abi.Arg:
        if isinstance(v, SymExpr):
            a = abi.expr_arg(v.prog, self.keep)
            self.expr_progs[a.table] = tuple(v.prog)  # (by address: the plan cache keys on the program, not on where it lies)
            return a
        if isinstance(v, Sym):
            kind = abi.ARG_SITE if v.src[0] == "site" else abi.ARG_INPUT
            return abi.Arg(kind, v.src[1], v.scale, v.offset, None)
        if isinstance(v, ParamVal):
            return abi.Arg(abi.ARG_PARAM, self.param_slot(v), 1.0, 0.0, None)
        if isinstance(v, _Table):
            return abi.Arg(abi.ARG_TABLE, v.idx.src[1], 0.0, 0.0, v.table.data_ptr())
        if isinstance(v, torch.Tensor) and v.dim() == 1 and v.shape[0] == self.n and self.n > 1:
            return abi.Arg(abi.ARG_INPUT, self._input(v), 1.0, 0.0, None)
        return abi.Arg(abi.ARG_CONST, 0, 0.0, Sym._const(v), None)

    def _input(self, col: torch.Tensor) -> int:
        if len(self.inputs) >= MAX_INPUT_COLS:
            raise PlanUnsupported("too many per-particle input columns")
        ops = get_ops()
        self.inputs.append(col.to(device=ops.device(), dtype=torch.float32).contiguous())
        return len(self.inputs) - 1

    def _call(self, addr, gen_fn, args):
        """`callee(*args) @ addr`: the callee takes one counter of this body and numbers its own sites afresh under
        fold_in(key, counter) — the kernel derives that key per particle (gjx_plan_create_scoped)."""
        from .lang import StaticGenerativeFunction

        if not self.allow_scopes or not isinstance(gen_fn, StaticGenerativeFunction):
            raise PlanUnsupported("nested generative function")
        if len(self.scopes) >= abi.MAX_SCOPES or self.depth >= 3:
            raise PlanUnsupported("too many / too deep nested calls")
        a = addr if isinstance(addr, tuple) else (addr,)
        self.record(addr, None)
        k = len(self.scopes)
        self.scopes.append([self.cur_scope, len(self.sites), None])
        rec = dict(addr=addr, gen_fn=gen_fn, args=args, items=[], retval=None)
        self.items.append(("call", rec))
        saved = (self.constraint, self.traces, self.cur_scope, self.items, self.prefix)
        self.constraint, self.traces, self.cur_scope, self.items = self._callee_constraint(a), {}, k + 1, rec["items"]
        self.prefix = self.prefix + a
        self.depth += 1
        try:
            rec["retval"] = gen_fn.source(*_spec_wrap(tuple(args)))  # (`@` inside reaches this tracer: it is the stack's top)
        finally:
            self.constraint, self.traces, self.cur_scope, self.items, self.prefix = saved
            self.depth -= 1
        self.scopes[k][2] = len(self.sites)
        return rec["retval"]

    def _callee_constraint(self, a: tuple) -> ChoiceMap:
        return self.constraint.get_submap(*a)

    def handle_trace(self, addr, gen_fn, args):
        if not isinstance(gen_fn, Distribution):
            return self._call(addr, gen_fn, args)
        if len(self.sites) >= abi.MAX_SITES:
            raise PlanUnsupported("too many sites")
        a = addr if isinstance(addr, tuple) else (addr,)
        sub = self.constraint.get_submap(*a)
        obs = sub.get_value()
        if obs is None and not sub.static_is_empty():
            raise PlanUnsupported("structured constraint at a distribution address")
        site = abi.Site()
        site.observed = 0 if obs is None else 1
        site.out_col = -1
        is_int = False
        if id(gen_fn) in _DIST_IDS:
            site.dist = _DIST_IDS[id(gen_fn)]
            site.arg[0] = self._arg(args[0])
            if site.dist != abi.DIST_BERNOULLI:
                site.arg[1] = self._arg(args[1])
            is_int = site.dist == abi.DIST_BERNOULLI
        elif gen_fn is bernoulli:
            kind, v = args[0]
            if kind != "probs":
                raise PlanUnsupported("bernoulli(logits=) in a fused plan")
            site.dist = abi.DIST_BERNOULLI
            site.arg[0] = self._arg(v)
            is_int = True
        elif isinstance(gen_fn, Categorical):
            kind, v = args[0] if isinstance(args[0], tuple) else ("logits", args[0])
            if isinstance(v, (Sym, _Table, ParamVal)):
                raise PlanUnsupported("data-dependent categorical parameters")
            site.dist = abi.DIST_CATEGORICAL
            site.cat_mode = 0 if gen_fn.sampling == "gumbel" else 1
            if isinstance(v, _Rows):  # `categorical(logits=trans[z])`: the row is chosen by an earlier value
                ops = get_ops()
                logits = torch.as_tensor(v.table, dtype=torch.float32).to(ops.device())
                if kind == "probs":
                    logits = torch.log(logits)
                logits = logits.contiguous()
                site.n_cat, site.n_rows = int(logits.shape[1]), int(logits.shape[0])
                site.arg[0] = self._arg(v.idx)
            else:
                logits = Categorical._logits((kind, v), self.n)
                if logits.shape[0] != 1:
                    raise PlanUnsupported("per-particle categorical parameters")
                site.n_cat, site.n_rows = int(logits.shape[1]), 1
                site.arg[0] = abi.Arg(abi.ARG_CONST, 0, 0.0, 0.0, None)
            self.keep.append(logits)
            site.logits = logits.data_ptr()
            is_int = True
        else:
            raise PlanUnsupported(f"distribution {gen_fn!r} has no fused sampler")
        idx = len(self.sites)
        obs_sym = None
        if obs is not None:
            if isinstance(obs, torch.Tensor) and obs.dim() == 1 and obs.shape[0] == self.n and self.n > 1:
                site.obs = abi.Arg(abi.ARG_INPUT, self._input(obs.to(torch.float32)), 1.0, 0.0, None)
            else:
                if self.use_params and not isinstance(obs, ParamVal):
                    Sym._const(obs)  # (rejects non-scalars)
                    obs_sym = ParamVal(obs)
                    site.obs = abi.Arg(abi.ARG_PARAM, self.param_slot(obs_sym), 1.0, 0.0, None)
                else:
                    site.obs = abi.Arg(abi.ARG_CONST, 0, 0.0, float(Sym._const(obs)), None)
        else:
            site.out_col = self.n_out
            self.n_out += 1
        self.sites.append(site)
        self.items.append(("site", idx))
        self.meta.append(dict(addr=addr, gen_fn=gen_fn, args=args, obs=obs, out_col=site.out_col, is_int=is_int,
                              dtype=gen_fn.value_dtype, path=self.prefix + a))
        self.record(addr, None)
        if obs is not None:
            # a constrained value may feed later sites: constants stay constants, columns become inputs
            if isinstance(obs, torch.Tensor) and obs.dim() == 1 and obs.shape[0] == self.n and self.n > 1:
                self.input_orig[site.obs.ref] = obs
                return Sym(self, ("input", site.obs.ref), is_int=is_int)
            return obs if obs_sym is None else obs_sym
        return _IntSym(self, ("site", idx)) if is_int else Sym(self, ("site", idx))


class _IntSym(Sym):
    """Integer-valued site (flip / categorical): usable as a table index or as a 0/1 value."""

    def __init__(self, tracer, src):
        super().__init__(tracer, src, is_int=True)


class TableProxy:
    """Wrap a constant 1-D tensor so that `proxy[int_site]` is traceable (`means[idx]`)."""

    def __init__(self, table):
        self.table = torch.as_tensor(table, dtype=torch.float32)

    def __getitem__(self, idx):
        if isinstance(idx, Sym):
            if not idx.is_int and idx.src[0] not in ("state", "obs"):  # (a carry / observation is rounded to nearest)
                raise PlanUnsupported("table index must be an integer-valued site")
            ops = get_ops()
            t = self.table.to(ops.device()).contiguous()
            idx.tracer.keep.append(t)
            return _Table(idx.tracer, t, idx)
        if isinstance(idx, torch.Tensor):
            return self.table.to(idx.device)[idx.long()]
        return self.table[idx]


def _contains_sym(v) -> bool:
    if isinstance(v, (Sym, _Table, _Rows, ParamVal)):
        return True
    if isinstance(v, (tuple, list)):
        return any(_contains_sym(x) for x in v)
    if isinstance(v, dict):
        return any(_contains_sym(x) for x in v.values())
    return False


def _trace(gen_fn, constraint: ChoiceMap, n: int, args):
    """-> (tracer, retval) of the body, observations and scalar arguments as launch parameters when the body only
    uses them affinely, as constants otherwise; None when the body is not plan-able at all."""
    for use_params in (True, False):
        tracer = PlanTracer(constraint, n, use_params)
        try:
            retval = tracer.run(gen_fn.source, tracer.wrap_args(args))
        except Exception:
            # PlanUnsupported, or any error provoked by feeding symbolic values to code that expects
            # tensors: the column path re-runs the body and raises genuine model errors itself.
            continue
        if not tracer.sites:
            return None
        return tracer, retval
    return None


_PLANS = threading.local()  # per thread: a plan's parameters are set-then-run, not a concurrent object
_PLAN_CACHE_MAX = 32


def _make_plan(tracer, estimate_only: bool = False):
    """The plan of a traced body.  Plans are cached by their site table (the model's STRUCTURE — observations and scalar
    arguments are parameters, set per call): a repeated call skips plan creation and, in the library, regenerating and
    looking up the specialised kernel's source.  `estimate_only`: the same walk with no value column (every site's
    out_col = -1) — what an estimate of the log-marginal alone needs: its kernel stores row sums and nothing else."""
    import ctypes as C
    from collections import OrderedDict

    from .runtime import fast_math_enabled

    ops, fast = get_ops(), fast_math_enabled()
    cache = getattr(_PLANS, "cache", None)
    if cache is None:
        cache = _PLANS.cache = OrderedDict()
    key_attr = "_table_key_est" if estimate_only else "_table_key"
    raw = getattr(tracer, key_attr, None)  # (a tracer that the trace cache hands back has been keyed before)
    arr = None
    sites = tracer.sites
    if estimate_only:
        sites = (abi.Site * len(tracer.sites))(*tracer.sites)  # (array construction copies the structures)
        for st in sites:
            st.out_col = -1
        sites = list(sites)
    if raw is None:
        arr = (abi.Site * len(sites))(*sites)
        raw = bytes(memoryview(arr))
    if arr is not None and tracer.expr_progs:  # programs enter the key by content: their addresses differ from trace to trace
        tmp = (abi.Site * len(tracer.sites)).from_buffer_copy(raw)
        progs = []
        for q in range(len(tracer.sites)):
            for k in range(2):
                if tmp[q].arg[k].kind == abi.ARG_EXPR:
                    progs.append((q, k, tracer.expr_progs[tmp[q].arg[k].table]))
                    tmp[q].arg[k].table = None
        raw = bytes(memoryview(tmp)) + repr(progs).encode()
    setattr(tracer, key_attr, raw)
    # Tables the site table points into (categorical logits, transition rows) enter the key by identity AND version: the
    # library derives per-plan tables from them at the plan's first compilation (CDFs, guides, log-probabilities), so an
    # in-place update of such a tensor (an EM / optimiser step) must not find the plan built from its old contents.
    tables = tuple((id(t), t._version) for t in tracer.keep if isinstance(t, torch.Tensor))
    key = (id(ops), fast, raw, tables, tuple(tuple(k) for k in tracer.scopes))
    hit = cache.get(key)
    if hit is not None:
        cache.move_to_end(key)
        plan = hit[0]
    else:
        plan = ops.plan_create(sites, fast_math=fast, scopes=[tuple(k) for k in tracer.scopes])
        cache[key] = (plan, tracer.keep)  # (the tables the site table points into stay alive with the plan)
        while len(cache) > _PLAN_CACHE_MAX:
            cache.popitem(last=False)
    if tracer.params:
        plan.set_params(tracer.params)
    return plan


TRACE_CACHE = True  # reuse the traced form of a body when it is called again with the SAME target (see _traced)
_TRACE_CACHE_MAX = 16


def _leaf_key(v):
    if isinstance(v, torch.Tensor):
        return ("t", id(v), v._version)  # (an in-place change of a tensor invalidates what was traced from it)
    if isinstance(v, (bool, int, float, str, type(None))):
        return ("v", type(v).__name__, v)
    if isinstance(v, (tuple, list)):  # containers by CONTENT: a list mutated in place must not hit its old trace
        return ("c", type(v).__name__, tuple(_leaf_key(x) for x in v))
    if isinstance(v, dict):
        return ("d", tuple((k, _leaf_key(x)) for k, x in v.items()))
    raise TypeError("an argument of a type the trace cache cannot key by value")  # (-> the call is traced afresh)


def _traced(gen_fn, constraint: ChoiceMap, n: int, args):
    """`_trace` with a small per-thread cache for the repeated-call pattern `alg.log_marginal_likelihood_estimate(key)`
    in a loop: the SAME generative function, constraint object and argument values.  Only an exact repetition hits (the
    key holds the identities and versions of every tensor involved and strong references to the objects, so an identity is
    never recycled while cached); like a jitted function, a body that reads a mutable global sees the value it had when it
    was first traced — `plan.TRACE_CACHE = False` turns the cache off."""
    if not TRACE_CACHE:
        return _trace(gen_fn, constraint, n, args)
    from collections import OrderedDict

    from .runtime import fast_math_enabled

    cache = getattr(_PLANS, "traces", None)
    if cache is None:
        cache = _PLANS.traces = OrderedDict()
    try:
        key = (id(gen_fn), id(constraint), n, fast_math_enabled(), id(get_ops()), tuple(_leaf_key(a) for a in args),
               tuple(_leaf_key(v) for _, v in constraint.leaves()))
        hash(key)
    except TypeError:
        return _trace(gen_fn, constraint, n, args)
    hit = cache.get(key)
    if hit is not None:
        cache.move_to_end(key)
        return hit[0]
    traced = _trace(gen_fn, constraint, n, args)
    cache[key] = (traced, gen_fn, constraint, args)  # (strong references: the identities in the key stay taken)
    while len(cache) > _TRACE_CACHE_MAX:
        cache.popitem(last=False)
    return traced


def _try_fused_generate_impl(gen_fn, pk: ParticleKeys, constraint: ChoiceMap, args):
    """-> (trace, weight) through the fused kernel, or None when the body is not plan-able."""
    if pk.kb.fold is not None or any(_needs_eager(a) for a in args):
        return None
    traced = _traced(gen_fn, constraint, pk.n, args)
    if traced is None:
        return None
    tracer, retval = traced
    ops = get_ops()
    plan = _make_plan(tracer)
    dtypes = [torch.float32] * tracer.n_out
    for m in tracer.meta:
        if m["out_col"] >= 0 and m["is_int"]:
            dtypes[m["out_col"]] = torch.int32
    vals, score, logw, mp, rows = ops.importance_run(plan, pk.kb, pk.n, tracer.inputs, dtypes, want_score=True,
                                                     want_max_partials=True, want_rows=True)
    site_vals = []
    for m in tracer.meta:
        if m["out_col"] >= 0:
            v = vals[m["out_col"]]
            if m["dtype"] == torch.bool:
                v = v != 0
            site_vals.append(v)
        else:
            site_vals.append(m["obs"])

    def resolve(x):
        if isinstance(x, Sym):
            base = site_vals[x.src[1]] if x.src[0] == "site" else tracer.inputs[x.src[1]]
            if not (x.has_mul or x.has_add):
                # the value itself, in its presented dtype (bool for flip, int32 for categorical; a constrained column
                # as it was given)
                return tracer.input_orig.get(x.src[1], base) if x.src[0] == "input" else base
            base = base.to(torch.float32) if isinstance(base, torch.Tensor) else float(base)
            out = base
            if x.has_mul:
                out = out * x.scale
            if x.has_add:
                out = out + x.offset
            return out
        if isinstance(x, SymExpr):
            def leaf(kind, ref):
                if kind == abi.EXPR_SITE:
                    return site_vals[ref]
                if kind == abi.EXPR_INPUT:
                    return tracer.inputs[ref]
                return tracer.params[ref]
            return x.evaluate(leaf)
        if isinstance(x, _Table):
            return x.table[site_vals[x.idx.src[1]].long()]
        if isinstance(x, _Rows):
            base = site_vals[x.idx.src[1]] if x.idx.src[0] == "site" else tracer.inputs[x.idx.src[1]]
            return x.table.to(base.device)[base.long()]
        if isinstance(x, ParamVal):
            return x.value
        if isinstance(x, tuple):
            return tuple(resolve(y) for y in x)
        if isinstance(x, list):
            return [resolve(y) for y in x]
        return x

    def build(items):  # a body's sub-traces in program order; a nested call's trace sums its sites' scores on demand
        sub = {}
        for kind, x in items:
            if kind == "site":
                m = tracer.meta[x]
                sub[m["addr"]] = ValueTrace(m["gen_fn"], (lambda a=m["args"]: tuple(resolve(y) for y in a)), site_vals[x])
            else:
                sub[x["addr"]] = StaticTrace(x["gen_fn"], tuple(resolve(y) for y in x["args"]), resolve(x["retval"]), build(x["items"]))
        return sub

    tr = StaticTrace(gen_fn, args, resolve(retval), build(tracer.items), score=score)
    tr.max_partials = mp  # lets a max-anchored log-sum-exp skip its max pass
    tr.row_stats = rows  # row-anchored partial sums: the log-marginal needs one tiny kernel more
    return tr, logw


def _fused_log_weights_batch_impl(gen_fn, pks: list, constraint: ChoiceMap, args):
    """Several independent importance passes of one plan-able body in ONE launch
    (`gjx_importance_run_batch`) -> (log-weights [B, n], lse f32[B]) or None.  The passes differ only in
    their particle keys (lazy children of B parent keys); the log-sum-exp of every pass is folded by one
    `gjx_lse_rows_batch` launch from the row sums the kernel emitted."""
    n = pks[0].n
    if any(pk.n != n or pk.kb.fold is not None or pk.kb.mode != 1 for pk in pks) or any(_needs_eager(a) for a in args):
        return None
    traced = _trace(gen_fn, constraint, n, args)
    if traced is None:
        return None
    tracer = traced[0]
    ops = get_ops()
    plan = _make_plan(tracer)
    dtypes = [torch.float32] * tracer.n_out
    for m in tracer.meta:
        if m["out_col"] >= 0 and m["is_int"]:
            dtypes[m["out_col"]] = torch.int32
    prep = ops.prepare_importance(plan, [pk.kb for pk in pks], n, tracer.inputs, dtypes, fold_batch=len(pks))
    prep.launch_passes(0, len(pks))
    prep.launch_fold(len(pks))
    return prep.logw_all[:, :n], prep.lse_all


def _fused_generate_batch_impl(gen_fn, pks: list, constraint: ChoiceMap, args):
    """B independent importance passes of one plan-able FLAT body in ONE launch, with everything a particle draw per pass
    needs: -> dict(values=[per latent site: (addr, tensor [B, n] in its presented dtype)], score [B, n], logw [B, stride] (the
    first n of each row), lse f32[B]) or None (not plan-able this way: nested calls, eager arguments, explicit keys)."""
    n = pks[0].n
    if any(pk.n != n or pk.kb.fold is not None or pk.kb.mode != 1 for pk in pks) or any(_needs_eager(a) for a in args):
        return None
    traced = _trace(gen_fn, constraint, n, args)
    if traced is None:
        return None
    tracer = traced[0]
    if any(kind != "site" for kind, _ in tracer.items):
        return None
    ops = get_ops()
    plan = _make_plan(tracer)
    dtypes = [torch.float32] * tracer.n_out
    for m in tracer.meta:
        if m["out_col"] >= 0 and m["is_int"]:
            dtypes[m["out_col"]] = torch.int32
    prep = ops.prepare_importance(plan, [pk.kb for pk in pks], n, tracer.inputs, dtypes, fold_batch=len(pks))
    prep.launch_passes(0, len(pks))
    prep.launch_fold(len(pks))
    values = []
    for m in tracer.meta:
        if m["out_col"] >= 0:
            v = prep.values_all[m["out_col"]][:, :n]
            values.append((m["addr"], (v != 0) if m["dtype"] == torch.bool else v))
    return dict(values=values, score=prep.score_all[:, :n], logw=prep.logw_all, lse=prep.lse_all, _keep=prep)


def _or_per_site(impl):
    """The fused route, or None (= the per-site path) when the library refuses the plan because the compiler is switched off
    (runtime.compiler_switched_off)."""
    import functools

    @functools.wraps(impl)
    def run(*a, **kw):
        from .runtime import compiler_switched_off

        try:
            return impl(*a, **kw)
        except abi.GjxError as e:
            if compiler_switched_off(e):
                return None
            raise

    return run


try_fused_generate = _or_per_site(_try_fused_generate_impl)
fused_log_weights_batch = _or_per_site(_fused_log_weights_batch_impl)
fused_generate_batch = _or_per_site(_fused_generate_batch_impl)


def _needs_eager(a) -> bool:
    return isinstance(a, torch.Tensor) and a.dim() >= 1 and a.numel() > 1
