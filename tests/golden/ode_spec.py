"""Independent evaluator of a gotran ``.ode`` model specification (fixture generator helper).

Runs ONLY in the build container (it reads the reference's ``.ode`` text under
/root/reference, which never travels to the GPU box).  It is deliberately independent of
``oracle/ionic.py`` and of the HIP kernels: the ``.ode`` file is walked with ``ast``, every
assignment is evaluated (a) numerically with NumPy and (b) symbolically with SymPy holding
earlier intermediates as opaque symbols, which is how the GRL1 linearisation
``d(dX_dt)/dX`` of the published gotranx scheme is defined.
"""

from __future__ import annotations

import ast
from pathlib import Path

import numpy as np
import sympy


def _walk(path: Path):
    tree = ast.parse(Path(path).read_text())
    for node in tree.body:
        if isinstance(node, ast.Expr) and isinstance(node.value, ast.Call):
            yield "call", node.value.func.id, node.value
        elif isinstance(node, ast.Assign):
            assert len(node.targets) == 1 and isinstance(node.targets[0], ast.Name)
            yield "assign", node.targets[0].id, node.value
        elif isinstance(node, ast.Expr):  # docstring / bare constant
            continue
        else:  # pragma: no cover
            raise ValueError(ast.dump(node))


def _kwvalue(node):
    # value or ScalarParam(value, unit=...)
    if isinstance(node, ast.Call):
        return float(ast.literal_eval(node.args[0]))
    return float(ast.literal_eval(node))


class OdeSpec:
    def __init__(self, path):
        self.path = Path(path)
        self.states: dict[str, float] = {}
        self.parameters: dict[str, float] = {}
        self.assignments: list[tuple[str, ast.AST]] = []
        for kind, name, node in _walk(self.path):
            if kind == "call" and name in ("states", "parameters"):
                target = self.states if name == "states" else self.parameters
                for kw in node.keywords:
                    target[kw.arg] = _kwvalue(kw.value)
            elif kind == "assign":
                self.assignments.append((name, node))
        self.state_names = list(self.states)
        self.parameter_names = list(self.parameters)
        self.assignments = self._dependency_order(self.assignments)

    @staticmethod
    def _dependency_order(assignments):
        """gotran files may use an intermediate before the line that defines it: order by dependencies
        (stable: a line keeps its place unless it has to wait for a later definition)."""
        defined_here = {name for name, _ in assignments}
        deps = {name: {n.id for n in ast.walk(node) if isinstance(n, ast.Name)} & defined_here - {name}
                for name, node in assignments}
        out, done, pending = [], set(), list(assignments)
        while pending:
            progressed = False
            rest = []
            for name, node in pending:
                if deps[name] <= done:
                    out.append((name, node))
                    done.add(name)
                    progressed = True
                else:
                    rest.append((name, node))
            if not progressed:
                raise ValueError(f"cyclic definitions: {[n for n, _ in rest]}")
            pending = rest
        return out

    # ---------------------------------------------------------------- numeric
    def evaluate(self, states: dict, parameters: dict, t: float) -> dict:
        def conditional(c, a, b):
            return np.where(c, a, b)

        ns = {
            "exp": np.exp, "log": np.log, "sqrt": np.sqrt, "floor": np.floor, "abs": np.abs, "Abs": np.abs,
            "Conditional": conditional, "Lt": np.less, "Le": np.less_equal, "Gt": np.greater,
            "Ge": np.greater_equal, "Eq": np.equal, "And": np.logical_and, "Or": np.logical_or,
            "time": t, "pi": np.pi,
        }
        ns.update(parameters)
        ns.update(states)
        out = {}
        with np.errstate(all="ignore"):
            for name, node in self.assignments:
                val = eval(compile(ast.Expression(node), str(self.path), "eval"), {"__builtins__": {}}, ns)
                ns[name] = val
                out[name] = val
        return out

    # ---------------------------------------------------------------- symbolic
    def linearized(self, total: bool = True) -> dict[str, sympy.Expr]:
        """d(dX_dt)/dX of each state derivative.  ``total=True``: every intermediate expression is
        resolved first (total self-derivative, the variant the Niederer table pins); ``total=False``:
        intermediates are held as opaque symbols (derivative of the expression as written)."""
        sym = {n: sympy.Symbol(n, real=True) for n in list(self.states) + list(self.parameters)}

        class Rel:
            """A relation that may also be used as a number (``Gt(x, 0)*x``, ToRORd_dynCl_endo_Land.ode): 1 where it
            holds, 0 elsewhere."""

            def __init__(self, rel):
                self.rel = rel

            def number(self):
                return sympy.Piecewise((1, self.rel), (0, True))

            def __mul__(self, other):
                return self.number() * other

            __rmul__ = __mul__

        def rel(fn):
            return lambda a, b: Rel(fn(*(x.number() if isinstance(x, Rel) else x for x in (a, b))))

        def cond(x):
            return x.rel if isinstance(x, Rel) else x

        ns = {
            "exp": sympy.exp, "log": sympy.log, "sqrt": sympy.sqrt, "floor": sympy.floor,
            "abs": sympy.Abs, "Abs": sympy.Abs,
            "Conditional": lambda c, a, b: sympy.Piecewise((a, cond(c)), (b, True)),
            "Lt": rel(sympy.Lt), "Le": rel(sympy.Le), "Gt": rel(sympy.Gt), "Ge": rel(sympy.Ge), "Eq": rel(sympy.Eq),
            "And": lambda *a: sympy.And(*map(cond, a)), "Or": lambda *a: sympy.Or(*map(cond, a)),
            "time": sympy.Symbol("time", real=True), "pi": sympy.pi,
        }
        ns.update(sym)
        out = {}
        for name, node in self.assignments:
            expr = eval(compile(ast.Expression(node), str(self.path), "eval"), {"__builtins__": {}}, ns)
            ns[name] = sympy.sympify(expr) if total else sympy.Symbol(name)
            if name.startswith("d") and name.endswith("_dt") and name[1:-3] in self.states:
                out[name[1:-3]] = sympy.diff(sympy.sympify(expr), sym[name[1:-3]])
        return out

    def grl1(self, states: dict, parameters: dict, t: float, dt: float, delta: float = 1e-8, total: bool = True):
        """Returns (rhs, lin, new_states) dicts keyed by state name."""
        vals = self.evaluate(states, parameters, t)
        lin = self.linearized(total)
        env = dict(parameters)
        env.update(states)
        env.update(vals)
        env["time"] = t
        rhs, J, new = {}, {}, {}
        for s in self.state_names:
            f = np.asarray(vals[f"d{s}_dt"], dtype=float) + 0.0 * np.asarray(states[s])
            rhs[s] = f
            e = lin[s]
            if e == 0:
                J[s] = None
                new[s] = states[s] + f * dt
                continue
            syms = sorted(e.free_symbols, key=lambda x: x.name)
            fn = sympy.lambdify(syms, e, "numpy", cse=True)
            # arguments of one shape: conditions on parameters alone sit next to conditions on states in the
            # lambdified select() lists
            args = [np.broadcast_to(np.asarray(env[x.name], dtype=float), np.shape(f)) for x in syms]
            with np.errstate(all="ignore"):
                Jv = np.asarray(fn(*args), dtype=float) + 0.0 * f
                J[s] = Jv
                new[s] = states[s] + np.where(np.abs(Jv) > delta, f * (np.exp(Jv * dt) - 1) / Jv, f * dt)
        return rhs, J, new
