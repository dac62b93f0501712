"""A device cell model from a user's gotran ``.ode`` file: ``beat.models.from_ode(path)``.

The reference takes ANY ``fun(states, t, parameters, dt)`` -- normally the function gotranx generates from an ``.ode`` file
(demos/niederer_benchmark.py:82-99: ``gotranx.load_ode`` + ``gotranx.cli.gotran2py.get_code(..., scheme=[generalized_rush_larsen])``)
-- and evaluates it with NumPy on the host (src/beat/odesolver.py:67-79).  The models this package ships (FHN, TP06, ToR-ORd) are
hand-written kernels; a model it does not ship used to run as a Python callable on host arrays, with the whole state array making
a round trip per step.  ``from_ode`` closes that: the file is parsed (``ast``: the format is Python syntax), every assignment is
turned into a SymPy expression, the right-hand sides f_i and the TOTAL self-derivatives J_ii = d f_i / d y_i (intermediates
resolved: the variant the reference's Niederer table pins) are put through common-subexpression elimination and printed
as one C++ ``Model`` struct for ``csrc/beat_ode_kernel.h``; the library compiles it at first use (``beat_ode_model_register``,
run-time ``hipcc --genco``, cached like the sparse-row instances) and every entry point that takes a model id takes this one:
the fused split step, the pending update, the library's step loop, per-node parameter rows, parameter classes and cell types in
one launch (``DolfinMultiODESolver``), the in-kernel time loop (``run`` / ``single_cell.get_steady_state``).  Schemes: ``generalized_rush_larsen`` (gotranx's GRL1:
y_i += f_i / J_ii (exp(J_ii dt) - 1) where |J_ii| > 1e-8, forward Euler elsewhere) and ``forward_euler``.  ``exp`` in the
kernel is the library's table-driven evaluation (what the shipped models use: <= 1 ulp, 13 instead of ~50 VALU instructions; the
argument is kept inside double range); ``fast_exp=False`` prints libm's.

The same expressions, lambdified for NumPy, are the handle's HOST evaluation (``numpy_step``): what the tests compare the kernel
with, and what runs where no GPU / compiler is to be had.  Parameters: a (P,) vector or per node (P, N); what a generated model
does not get is the instance compiled for a FEW varying rows (beat_ode_step_rows: a (P, N) array of a generated model is read
whole, or runs as classes when its columns are few).  SymPy is needed (it is what gotranx itself builds on).
"""

from __future__ import annotations

import ast
import hashlib
from pathlib import Path

import numpy as np

from ._base import DeviceModel

_SCHEMES = ("generalized_rush_larsen", "forward_euler")
# gotranx's own names for the same two schemes (gotranx.schemes: what its code generator is asked for)
_SCHEME_ALIASES = {"forward_generalized_rush_larsen": "generalized_rush_larsen", "forward_explicit_euler": "forward_euler",
                   "explicit_euler": "forward_euler"}


def _walk(text: str, where: str):
    tree = ast.parse(text, filename=where)
    for node in tree.body:
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Call) and isinstance(node.value.func, ast.Name):
            yield "call", node.value.func.id, node.value
        elif isinstance(node, ast.Assign):
            if len(node.targets) != 1 or not isinstance(node.targets[0], ast.Name):
                raise ValueError(f"{where}:{node.lineno}: only plain assignments `name = expression` are understood")
            yield "assign", node.targets[0].id, node.value
        elif isinstance(node, ast.Expr):  # a docstring / bare constant
            continue
        else:
            raise ValueError(f"{where}:{node.lineno}: statement not understood in an .ode file")


_EXPR_NODES = (ast.Expression, ast.BinOp, ast.UnaryOp, ast.Call, ast.Name, ast.Constant, ast.Load, ast.Add, ast.Sub, ast.Mult, ast.Div,
               ast.Pow, ast.USub, ast.UAdd, ast.Compare, ast.Lt, ast.LtE, ast.Gt, ast.GtE, ast.Eq, ast.NotEq, ast.BoolOp, ast.And, ast.Or,
               ast.Not, ast.IfExp, ast.Mod)


def _check_expression(node, where: str, known) -> None:
    """The right-hand side of an assignment is evaluated (with SymPy objects for the names) to build the expression tree: only
    arithmetic, comparisons and calls of the functions an .ode file may use pass -- no attribute access, subscripts, lambdas,
    comprehensions or keyword tricks: an .ode file is data, and this is not the place where it gets to run code."""
    for n in ast.walk(node):
        if not isinstance(n, _EXPR_NODES):
            raise ValueError(f"{where}:{getattr(n, 'lineno', '?')}: `{type(n).__name__}` is not understood in an expression of an .ode file")
        if isinstance(n, ast.Call):
            if not isinstance(n.func, ast.Name) or n.func.id not in known or n.keywords:
                raise ValueError(f"{where}:{n.lineno}: call of an unknown function in an .ode file"
                                 + (f": {n.func.id}" if isinstance(n.func, ast.Name) else ""))
        if isinstance(n, ast.Constant) and not isinstance(n.value, (int, float)):
            raise ValueError(f"{where}:{n.lineno}: only numbers are understood in an expression of an .ode file")


class _ComparisonsAsNumbers(ast.NodeTransformer):
    """gotran lets a comparison stand in arithmetic as 0 or 1 -- ``gammas*((zetas > 0)*zetas + (zetas < -1)*(-zetas - 1))`` in the
    Land model, ``Gt(Zetas, 0)*Zetas`` in the reference's own file: a comparison (infix or Lt / Le / Gt / Ge / Eq / And / Or / Not)
    that is an operand of an arithmetic operator becomes ``_indicator(comparison)``
    (Piecewise((1, c), (0, True))); as an argument of Conditional / And / Or / Not it stays a condition."""

    @staticmethod
    def _wrap(node):
        relation = isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in ("Lt", "Le", "Gt", "Ge", "Eq", "And", "Or", "Not")
        if relation or isinstance(node, (ast.Compare, ast.BoolOp)):
            return ast.copy_location(ast.Call(func=ast.Name(id="_indicator", ctx=ast.Load()), args=[node], keywords=[]), node)
        return node

    def visit_BinOp(self, node):
        self.generic_visit(node)
        node.left, node.right = self._wrap(node.left), self._wrap(node.right)
        return node

    def visit_UnaryOp(self, node):
        self.generic_visit(node)
        if not isinstance(node.op, ast.Not):
            node.operand = self._wrap(node.operand)
        return node


def _number(node) -> float:
    if isinstance(node, ast.Call):  # ScalarParam(value, unit=..., ...)
        return float(ast.literal_eval(node.args[0]))
    return float(ast.literal_eval(node))


def _dependency_order(assignments):
    """gotran files may use an intermediate before the line that defines it: order by dependencies (stable)."""
    defined = {name for name, _ in assignments}
    deps = {name: ({n.id for n in ast.walk(node) if isinstance(n, ast.Name)} & defined) - {name} for name, node in assignments}
    out, done, pending = [], set(), list(assignments)
    while pending:
        rest = [(n, e) for n, e in pending if not deps[n] <= done]
        ready = [(n, e) for n, e in pending if deps[n] <= done]
        if not ready:
            raise ValueError(f"cyclic definitions: {[n for n, _ in rest]}")
        for n, e in ready:
            out.append((n, e))
            done.add(n)
        pending = rest
    return out


class OdeFileModel(DeviceModel):
    """A :class:`DeviceModel` generated from an ``.ode`` file (see the module docstring)."""

    def __init__(self, path, scheme="generalized_rush_larsen", v_name=None, name=None, fast_exp=True):
        import sympy

        self.fast_exp = bool(fast_exp)

        scheme = _SCHEME_ALIASES.get(scheme, scheme)
        if scheme not in _SCHEMES:
            raise ValueError(f"scheme must be one of {_SCHEMES}, got {scheme!r}")
        self.path = Path(path)
        self.scheme = scheme
        text = self.path.read_text()
        states, params, assigns = {}, {}, []
        for kind, nm, node in _walk(text, str(self.path)):
            if kind == "call" and nm in ("states", "parameters"):
                target = states if nm == "states" else params
                for kw in node.keywords:
                    target[kw.arg] = _number(kw.value)
            elif kind == "assign":
                assigns.append((nm, node))
            # (expressions(...), component(...), comment(...): grouping only)
        if not states:
            raise ValueError(f"{self.path}: no states(...) found")
        assigns = _dependency_order(assigns)
        if v_name is None:
            v_name = next((s for s in states if s.lower() in ("v", "vm", "v_m")), None)
        if v_name is not None and v_name not in states:
            raise KeyError(f"{v_name!r} is not a state of {self.path.name}")
        stem = "".join(c if c.isalnum() else "_" for c in (name or self.path.stem))
        super().__init__(f"{stem}_{scheme}", -1, states, params, v_name)
        # ---- symbolic right-hand sides and total self-derivatives
        ysym = {s: sympy.Symbol(s, real=True) for s in states}
        psym = {p: sympy.Symbol(p, real=True) for p in params}
        tsym, dtsym = sympy.Symbol("time", real=True), sympy.Symbol("dt", real=True)

        def cond(c):
            return c

        def piecewise(c, a, b):
            return sympy.Piecewise((a, c), (b, True))

        ns = {"exp": sympy.exp, "log": sympy.log, "sqrt": sympy.sqrt, "floor": sympy.floor, "abs": sympy.Abs, "Abs": sympy.Abs,
              "pow": sympy.Pow, "sin": sympy.sin, "cos": sympy.cos, "tan": sympy.tan, "tanh": sympy.tanh, "sinh": sympy.sinh,
              "cosh": sympy.cosh, "atan": sympy.atan, "asin": sympy.asin, "acos": sympy.acos,
              "Conditional": piecewise, "Lt": sympy.Lt, "Le": sympy.Le, "Gt": sympy.Gt, "Ge": sympy.Ge, "Eq": sympy.Eq,
              "And": sympy.And, "Or": sympy.Or, "Not": sympy.Not, "time": tsym, "t": tsym, "pi": sympy.pi,
              "_indicator": lambda c: sympy.Piecewise((sympy.Integer(1), c), (sympy.Integer(0), True))}
        ns.update(psym)
        ns.update(ysym)
        rhs = {}
        callables = {k for k, v in ns.items() if callable(v) and not isinstance(v, sympy.Basic)}
        for nm, node in assigns:
            _check_expression(node, str(self.path), callables - {"_indicator"})
            node = ast.fix_missing_locations(_ComparisonsAsNumbers().visit(node))
            try:
                expr = sympy.sympify(eval(compile(ast.Expression(node), str(self.path), "eval"), {"__builtins__": {}}, ns))
            except Exception as exc:  # noqa: BLE001
                raise ValueError(f"{self.path}: cannot turn `{nm} = ...` into an expression: {exc}") from exc
            ns[nm] = expr
            if nm.startswith("d") and nm.endswith("_dt") and nm[1:-3] in states:
                rhs[nm[1:-3]] = expr
        missing = [s for s in states if s not in rhs]
        if missing:
            raise ValueError(f"{self.path}: no d<state>_dt for {missing}")
        names = list(states)
        f = [rhs[s] for s in names]
        J = [sympy.diff(rhs[s], ysym[s]) if scheme == "generalized_rush_larsen" else sympy.Integer(0) for s in names]
        self._grl = [bool(j != 0) for j in J]
        repl, red = sympy.cse(f + J, symbols=sympy.numbered_symbols("x_"), optimizations="basic")
        self._sym = dict(y=[ysym[s] for s in names], p=[psym[p] for p in params], t=tsym, dt=dtsym, repl=repl,
                         f=red[: len(names)], J=red[len(names):])
        self._numpy_fn = None
        self._registered = None   # the library's id, once the compiled kernel has passed _self_check
        self._library_id = None   # the id beat_ode_model_register gave (the source is handed over once per process)
        self.source = self._cxx(stem)
        self.key = f"{self.cxx_name}"

    # ------------------------------------------------------------------------------------------------ C++
    def _cxx(self, stem: str) -> str:
        import os

        import sympy
        from sympy.printing.c import C99CodePrinter

        class Printer(C99CodePrinter):
            def _print_Pow(self, expr):  # small integer powers as products (pow() on fp64 is a long software routine)
                b, e = expr.as_base_exp()
                if e.is_Integer and 2 <= int(e) <= 4:
                    pb = self.parenthesize(b, 1000)  # as an atom
                    return "(" + "*".join([pb] * int(e)) + ")"
                if e.is_Integer and -4 <= int(e) <= -1:
                    pb = self.parenthesize(b, 1000)
                    return "(1.0/(" + "*".join([pb] * (-int(e))) + "))"
                return super()._print_Pow(expr)

            def _print_Piecewise(self, expr):
                # BRANCH-FREE: both arms evaluated, the result selected (v_cndmask).  As `c ? a : b` with the arms inline the
                # compiler builds divergent branches; a heavily spilled kernel (the reference's ToR-ORd files generate 250 - 300 spilled
                # SGPRs and 450 spilled VGPRs) then reloads registers it spilled under another lane mask -- wrong values on exactly the
                # nodes that take the other arm, in the states behind that arm, different from run to run (ROCm 7.2; seen in round 1
                # on the generated ToR-ORd kernel and reproduced in round 5: tools/diag_spill.py, profiles/r05_generated_spills.md).
                # A lane computes both arms of a divergent branch anyway.
                if os.environ.get("BEAT_ODE_BRANCHES") == "1":  # the form that was seen miscompiled (tests of the self checks)
                    return super()._print_Piecewise(expr)
                args = list(expr.args)
                out = self._print(args[-1][0])
                if args[-1][1] != True:  # noqa: E712 -- no default arm: what C's chained ?: would leave undefined
                    out = f"beat_sel({self._print(args[-1][1])}, {out}, 0.0)"
                for e, c in reversed(args[:-1]):
                    out = f"beat_sel({self._print(c)}, {self._print(e)}, {out})"
                return out

        # exp(): the library's table-driven evaluation (FastMath::exp of csrc/ionic_models.h, what the shipped models use: 13 VALU
        # instructions, <= 1 ulp, the 256-entry table in LDS) with the argument kept inside double range -- or libm's
        exp_name = "fexp" if self.fast_exp else "exp"
        pr = Printer({"contract": False, "user_functions": {"exp": exp_name}})
        y, p = self._sym["y"], self._sym["p"]
        sub = {s: sympy.Symbol(f"y_{k}") for k, s in enumerate(y)}
        sub.update({s: sympy.Symbol(f"p_{k}") for k, s in enumerate(p)})
        sub[self._sym["t"]] = sympy.Symbol("t")
        used = set()
        ns_, np_ = len(y), len(p)
        vi = self.state_index(self.v_name) if self.v_name else 0
        # Order: state by state, each one's common subexpressions right ahead of its update (depth first, those not yet emitted),
        # then the update and its store -- a temporary is defined next to its first use and a state's result leaves the registers
        # when it is final.  (All temporaries first and all updates last -- the order the elimination returns them in -- keeps every
        # one of them and all NS results alive to the end: the 48-state test model needed 256 + 256 registers and 450 B of scratch
        # per lane that way.)
        temp = {lhs: e.xreplace(sub) for lhs, e in self._sym["repl"]}
        emitted, lines = set(), []

        def emit(expr):
            stack = [(sym, False) for sym in sorted(expr.free_symbols, key=str, reverse=True) if sym in temp]
            while stack:
                sym, done = stack.pop()
                if sym in emitted:
                    continue
                if done:
                    emitted.add(sym)
                    lines.append(f"    const double {sym} = {pr.doprint(temp[sym])};")
                    continue
                stack.append((sym, True))
                for dep in sorted(temp[sym].free_symbols, key=str, reverse=True):
                    if dep in temp and dep not in emitted:
                        stack.append((dep, False))

        order = list(range(ns_))
        if os.environ.get("BEAT_ODE_V_LAST", "1") == "1" and self.v_name:  # the potential's equation sums every current: last, when the currents exist
            order = [k for k in order if k != vi] + [vi]
        if os.environ.get("BEAT_ODE_EMIT") == "global":  # every temporary first, in the elimination's order (tests: the heavily spilled form)
            for lhs in temp:
                emit(lhs)
            order = list(range(ns_))
        for k in order:
            fk, jk = self._sym["f"][k].xreplace(sub), self._sym["J"][k].xreplace(sub)
            emit(fk)
            emit(jk)
            used |= fk.free_symbols | jk.free_symbols
            if self._grl[k]:
                if os.environ.get("BEAT_ODE_BRANCHES") == "1":  # (the former output: see _print_Piecewise)
                    lines.append(f"    {{ const double f = {pr.doprint(fk)}; const double J = {pr.doprint(jk)};\n"
                                 f"      io.store({k}, y_{k} + (fabs(J) > 1e-8 ? f / J * ({exp_name}(J * dt) - 1.0) : f * dt)); }}")
                    continue
                lines.append(f"    {{ const double f = {pr.doprint(fk)}; const double J = {pr.doprint(jk)};\n"
                             f"      io.store({k}, y_{k} + beat_sel(fabs(J) > 1e-8, f / J * ({exp_name}(J * dt) - 1.0), f * dt)); }}")
            else:
                lines.append(f"    io.store({k}, y_{k} + dt * ({pr.doprint(fk)}));")
        for e in emitted:
            used |= temp[e].free_symbols
        body = []
        loads = [f"    const double y_{k} = io.load({k});" for k in range(ns_)]
        pl = [f"    const double p_{k} = p[{k}];" for k in range(np_) if sympy.Symbol(f"p_{k}") in used]
        digest = hashlib.sha1(("\n".join(lines + body) + self.scheme).encode()).hexdigest()[:12]
        self.cxx_name = f"Ode_{stem}_{digest}"
        return (f"// generated by beat.models.from_ode from {self.path.name} ({self.scheme}): {ns_} states, {np_} parameters\n"
                f"struct {self.cxx_name} {{\n"
                f"  static constexpr int NS = {ns_}, NP = {max(np_, 1)}, V_INDEX = {vi};\n"
                "  static constexpr bool REGISTER_LOOP = true;\n"
                "  static constexpr int WAVES = 1, WAVES_PER_NODE = 1;\n"
                "  struct Derived { double unused; };\n"
                "  template <class P> __host__ __device__ static Derived derive(const P&) { return Derived{0.0}; }\n"
                "  template <class IO, class P>\n"
                "  __device__ static __forceinline__ void step(const IO& io, const P& p, const Derived&, const FastMath& fm, double t, double dt) {\n"
                + ("    // (what libm / NumPy give everywhere: NaN stays NaN, overflow is inf, underflow goes through the subnormals to 0)\n"
                   "    // (k = round(x 256 / ln 2) must fit 32 bits: the clamp; inside it v_ldexp_f64 underflows to 0 through the subnormals and\n"
                   "    // overflows to inf as libm does; the clamp turns a NaN argument into a number, hence the select)\n"
                   "    const auto fexp = [&fm](double x) { const double e = fm.exp(fmin(fmax(x, -1.0e6), 1.0e3)); return x != x ? x : e; };\n" if self.fast_exp else "")
                + "    const auto beat_sel = [](bool c, double a, double b) { return c ? a : b; };\n"
                + "\n".join(loads + pl + lines + body) + "\n  }\n};\n")

    # ------------------------------------------------------------------------------------------------ NumPy
    def numpy_step(self, states, t, parameters, dt):
        """One step on the HOST with NumPy: the same expressions as the kernel's (the reference's way of evaluating ``fun``)."""
        import sympy

        if self._numpy_fn is None:
            s = self._sym
            # (the common subexpressions found at construction are handed to lambdify as they are: local assignments in the generated function)
            self._numpy_fn = sympy.lambdify(s["y"] + s["p"] + [s["t"]], [s["f"], s["J"]], "numpy", cse=lambda exprs: (s["repl"], exprs))
        y = np.asarray(states, dtype=np.float64)
        one_d = y.ndim == 1
        y2 = y.reshape(y.shape[0], -1)
        p = np.asarray(parameters, dtype=np.float64)  # (P,) or per node (P, N)
        if p.ndim == 2 and p.shape[1] != y2.shape[1]:
            raise ValueError(f"per-node parameters must have shape ({len(p)}, {y2.shape[1]}), got {p.shape}")
        # (every argument an array of the nodes' shape: NumPy's and / or of a condition on parameters alone with one on a state
        # would otherwise be a reduction over a ragged pair)
        n = y2.shape[1]
        pn = [np.broadcast_to(np.asarray(p[k], dtype=np.float64), (n,)) for k in range(len(p))]
        with np.errstate(all="ignore"):
            f, J = self._numpy_fn(*[y2[k] for k in range(y2.shape[0])], *pn, np.full(n, float(t)))
        out = np.empty_like(y2)
        for k in range(y2.shape[0]):
            fk = np.broadcast_to(np.asarray(f[k], dtype=np.float64), y2[k].shape)
            if self._grl[k]:
                Jk = np.broadcast_to(np.asarray(J[k], dtype=np.float64), y2[k].shape)
                with np.errstate(all="ignore"):
                    out[k] = y2[k] + np.where(np.abs(Jk) > 1e-8, fk / np.where(Jk == 0, 1.0, Jk) * (np.exp(Jk * dt) - 1.0), fk * dt)
            else:
                out[k] = y2[k] + dt * fk
        return out[:, 0].copy() if one_d else out

    # ------------------------------------------------------------------------------------------------ device
    def register(self) -> int:
        """The library's id for this model (beat_ode_model_register: the source is handed over once per process; the kernel is
        compiled when a step first needs it).  Raises BeatHipError where the library cannot compile at run time."""
        import ctypes as C

        from .. import _hip

        if self._registered is None:
            if self._library_id is None:
                lib = _hip.load()
                mid = C.c_int(-1)
                _hip.check(lib.beat_ode_model_register(self.cxx_name.encode(), self.source.encode(), self.num_states,
                                                       max(self.num_parameters, 1), self.state_index(self.v_name) if self.v_name else 0,
                                                       C.byref(mid)))
                self._library_id = int(mid.value)
            # the id is handed out only once the compiled kernel has passed its check: a caller that catches the check's error and
            # asks again gets the check again, not the id of a kernel known to be wrong
            self.model_id = self._library_id  # (the check itself steps the model through DeviceModel.__call__)
            try:
                self._self_check()
            except BaseException:
                self.model_id = -1
                raise
            self._registered = self._library_id
        return self._registered

    def _sample_states(self, n: int, seed: int = 0) -> np.ndarray:
        """States to try a kernel on: the initial values, the potential spread over [-100, 60] mV, values in [0, 1] (gates) spread
        over [0, 1], everything else within +-30 % of its initial value."""
        rng = np.random.default_rng(seed)
        y = np.repeat(self.init_state_values()[:, None], n, axis=1)
        for k, name in enumerate(self.state_names):
            y0 = y[k, 0]
            if name == self.v_name:
                y[k] = rng.uniform(-100.0, 60.0, n)
            elif 0.0 <= y0 <= 1.0:
                y[k] = rng.uniform(0.0, 1.0, n)
            else:
                y[k] = y0 * rng.uniform(0.7, 1.3, n)
        y[:, 0] = self.init_state_values()
        return y

    def _self_check(self) -> None:
        """The model's PLAIN kernel instance (uniform parameters, no pending update) against the NumPy evaluation of the same
        expressions on 2048 sample states (to 1e-6 of each value), once per process: a big generated kernel is heavily spilled, and such a kernel has been
        seen miscompiled (round 1; reproduced in round 5, tools/diag_spill.py) -- wrong values on part of the nodes, silently.
        The other instances (per-node rows, pending update, classes) are held against this one by the library at their first
        launch (csrc/beat_ode_jit.hip: custom_cross_check).  BEAT_JIT_SELF_CHECK=0 skips both."""
        import os

        if os.environ.get("BEAT_JIT_SELF_CHECK", "1") == "0" or getattr(self, "_verified", False):
            return
        y = self._sample_states(2048)
        p = self.init_parameter_values()
        worst = 0.0
        for t, dt in ((0.0, 0.01), (0.37, 0.05)):
            dev = DeviceModel.__call__(self, states=y, t=t, parameters=p, dt=dt)
            with np.errstate(all="ignore"):
                ref = self.numpy_step(y, t, p, dt)
            ok = np.isfinite(ref)
            with np.errstate(all="ignore"):  # (inf - inf where both sides overflow: masked by `ok`)
                err = np.abs(dev - ref)[ok] / np.maximum(np.maximum(np.abs(ref), np.abs(y)), 1e-12)[ok]
            # (1e-6 of the value: a sample far from the model's physiological range may sit where exp(J dt) - 1 cancels, and the two
            # exp()s differ in the last bit -- 3e-8 seen; a miscompiled kernel is wrong by O(1) on thousands of values)
            if not np.isfinite(dev[ok]).all() or (err.size and err.max() > 1e-6):
                bad = np.argwhere(~(np.abs(np.where(ok, dev - ref, 0.0)) <= 1e-6 * np.maximum(np.maximum(np.abs(ref), np.abs(y)), 1e-12)))
                k, i = (int(bad[0][0]), int(bad[0][1])) if len(bad) else (0, 0)
                raise RuntimeError(
                    f"{self.name}: the compiled kernel differs from the NumPy evaluation of the same expressions (state {self.state_names[k]}, "
                    f"sample {i}: {dev[k, i]!r} against {ref[k, i]!r}; {len(bad)} of {ok.sum()} values) -- a miscompiled (heavily spilled) kernel; "
                    "other compiler flags (BEAT_JIT_EXTRA_FLAGS) may help, BEAT_JIT_SELF_CHECK=0 skips this check")
            worst = max(worst, float(err.max()) if err.size else 0.0)
        self._verified = True
        self.self_check_error = worst

    def __call__(self, states=None, t=0.0, parameters=None, dt=None, **kwargs):
        if dt is None:
            raise TypeError("dt is required")
        try:
            import torch

            on_gpu = torch.cuda.is_available()
        except Exception:  # noqa: BLE001
            on_gpu = False
        if on_gpu:
            self.register()
            return super().__call__(states=states, t=t, parameters=parameters, dt=dt, **kwargs)
        return self.numpy_step(states, t, parameters, dt)

    def run(self, states, parameters, dt, nsteps, nbeats=1, t0=0.0, track_indices=None, save_freq=1):
        """nbeats x nsteps steps inside ONE launch (beat_ode_run: ``ode_run_kernel`` instantiated for this model at first use), as
        for the shipped models -- ``single_cell.get_steady_state`` with a generated model paces on the device."""
        self.register()
        return super().run(states, parameters, dt, nsteps, nbeats=nbeats, t0=t0, track_indices=track_indices, save_freq=save_freq)


def from_ode(path, scheme: str = "generalized_rush_larsen", v_name: str | None = None, name: str | None = None,
             fast_exp: bool = True) -> OdeFileModel:
    """A device cell model from a gotran ``.ode`` file: pass the result as ``fun`` to ``DolfinODESolver`` (and use its
    ``init_state_values`` / ``init_parameter_values`` / ``state_index`` / ``parameter_index`` as those of a gotranx module)."""
    return OdeFileModel(path, scheme=scheme, v_name=v_name, name=name, fast_exp=fast_exp)
