#!/usr/bin/env python3
"""Cell-model kernel generator (development tool, run in the build container).

Reads a gotran ``.ode`` model specification (the reference keeps them under odes/), and emits

  * a C++ header with a model struct for csrc/beat_ode.hip: one generalized-Rush-Larsen (GRL1) step per
    node with the total self-derivatives d f_i / d y_i obtained by forward-mode differentiation through
    the model's own intermediate expressions (chain rule per state; no expression is expanded), and
  * a Python data module with the state / parameter names and defaults.

usage: gen_cell_model.py <model.ode> <StructName> <out_header> <out_py>

The generated files are ordinary source; the tool only has to be re-run when a model is added.  It is the way to
bring a new model up quickly (correct values, golden-test green) -- not the way to a fast kernel: ToR-ORd-dynCl's
generated step kept 192 doubles live and ran at one wave per SIMD with spills (9.4 ms at 256^3); the kernel the library
ships for that model is hand-organised (fenicsx-beat_amd/csrc/torord_dyncl.h, 3.7 ms), and only the names / defaults
module this tool wrote (beat/models/_torord_dyncl_data.py) is still in use.
"""

from __future__ import annotations

import ast
import sys
from pathlib import Path

import sympy
from sympy.printing.c import C99CodePrinter

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests" / "golden"))
from ode_spec import OdeSpec  # noqa: E402  (ast walker + dependency ordering of the assignments)


class Printer(C99CodePrinter):
    def _print_Float(self, e):
        return repr(float(e))

    def _print_Integer(self, e):
        return f"{int(e)}.0"

    def _print_Rational(self, e):
        return f"({int(e.p)}.0 / {int(e.q)}.0)"

    def _print_Pow(self, e):
        b, x = e.base, e.exp
        if x.is_Float and float(x) == int(float(x)):
            x = sympy.Integer(int(float(x)))
        if x.is_Float and float(2 * x) == int(float(2 * x)):
            x = sympy.Rational(int(float(2 * x)), 2)
        if x.is_Rational and not x.is_Integer and x.q == 2 and abs(x.p) in (3, 5):
            bs = self.parenthesize(b, 1000)
            body = f"({bs} * sqrt({self._print(b)}))" if abs(x.p) == 3 else f"({bs} * {bs} * sqrt({self._print(b)}))"
            return body if x.p > 0 else f"beat_rcp({body})"
        if x.is_Integer:
            n = int(x)
            bs = self.parenthesize(b, 1000)
            if n == -1:
                return f"beat_rcp({self._print(b)})"
            if 1 < abs(n) <= 4:
                prod = " * ".join([bs] * abs(n))
                return f"({prod})" if n > 0 else f"beat_rcp({prod})"
        if x == sympy.Rational(1, 2):
            return f"sqrt({self._print(b)})"
        if x == -sympy.Rational(1, 2):
            return f"beat_rcp(sqrt({self._print(b)}))"
        return f"pow({self._print(b)}, {self._print(x)})"

    def _print_Mul(self, e):
        # a / b is emitted as a * beat_rcp(b) (v_rcp_f64 + one cubic Newton step, <= 1 ulp) instead of the
        # ~12-instruction IEEE division sequence; equal denominators are merged by the compiler's CSE
        num, den = [], []
        for a in e.args:
            if a.is_Pow and a.exp.is_Number and a.exp.is_negative:
                den.append(sympy.Pow(a.base, -a.exp))
            else:
                num.append(a)
        if not den:
            return super()._print_Mul(e)
        sign = ""
        if num and num[0].is_Number and num[0].is_negative:
            sign = "-"
            num[0] = -num[0]
            if num[0] == 1:
                num = num[1:]
        d = sympy.Mul(*den)
        r = f"beat_rcp({self._print(d)})"
        if not num:
            return f"{sign}{r}"
        n = sympy.Mul(*num)
        ns = self._print(n)
        if n.is_Add:
            ns = f"({ns})"
        return f"{sign}{ns}*{r}"

    def _print_exp(self, e):
        return f"fm.exp({self._print(e.args[0])})"

    def _print_log(self, e):
        return f"fm.log({self._print(e.args[0])})"

    def _print_beat_guard(self, e):
        return f"beat_guard({self._print(e.args[0])})"

    def _print_Abs(self, e):
        return f"fabs({self._print(e.args[0])})"

    def _print_floor(self, e):
        return f"floor({self._print(e.args[0])})"

    def _print_Piecewise(self, e):
        out = None
        for val, cond in reversed(e.args):
            v = self._print(val)
            out = v if out is None and cond == True else f"(({self._print(cond)}) ? {v} : {out if out is not None else '0.0'})"  # noqa: E712
        return out

    def _print_Relational(self, e):
        op = {"==": "==", "!=": "!=", "<": "<", "<=": "<=", ">": ">", ">=": ">="}[e.rel_op]
        return f"{self._print(e.lhs)} {op} {self._print(e.rhs)}"

    def _print_And(self, e):
        return " && ".join(f"({self._print(a)})" for a in e.args)

    def _print_Or(self, e):
        return " || ".join(f"({self._print(a)})" for a in e.args)

    def _print_Symbol(self, e):
        return SYMBOL_NAMES.get(e.name, e.name)


class beat_guard(sympy.Function):
    """The membrane potential kept a hair away from a removable singularity (printed as the device function of that
    name); its derivative is taken as 1."""

    nargs = 1

    def fdiff(self, argindex=1):
        return sympy.Integer(1)


def ghk_singular_inputs(exprs, states):
    """Names of the intermediates through which Goldman-Hodgkin-Katz fluxes  w g / (exp(u) - 1)  see the membrane
    potential (w, u proportional to v: 0/0 at v = 0), and the potential's symbol.

    The value of such a flux is fine next to the singularity, but its v-derivative is a difference of two O(1/u)
    terms whose relative error grows like ulp / u^2: a node that passes within ~1e-10 mV of 0 gets a self-derivative
    of ~1e8 of either sign and the Rush-Larsen exponential overflows (seen with the hand-written TP06 kernel at
    256^3, where the singular potential is 15 mV).  The generated kernels therefore evaluate these intermediates at
    a potential that is never closer than beat_guard's 1e-4 mV to the singular value."""
    names, vsym = set(), None
    for e in exprs.values():
        for q in e.atoms(sympy.Pow):
            if not (q.exp.is_Number and float(q.exp) < 0 and q.base.is_Add and len(q.base.args) == 2):
                continue
            ex = [t for t in q.base.args if isinstance(t, sympy.exp)]
            if len(ex) != 1 or not any(t.is_Number and float(t) == -1.0 for t in q.base.args):
                continue
            inputs = {sy.name for sy in ex[0].args[0].free_symbols if sy.name in exprs}
            for name in inputs:  # u must vanish with exactly one state (the potential), and so must a co-factor
                d = exprs[name]
                sv = [sy for sy in d.free_symbols if sy.name in states]
                if len(sv) == 1 and d.subs(sv[0], 0) == 0:
                    vsym = sv[0]
                    names.add(name)
    if vsym is not None:  # numerators: every other intermediate that is a multiple of the same potential alone
        for name, d in exprs.items():
            sv = [sy for sy in d.free_symbols if sy.name in states]
            if sv == [vsym] and not any(sy.name in exprs for sy in d.free_symbols) and d.subs(vsym, 0) == 0 \
                    and sympy.simplify(d / vsym).is_constant(vsym):
                names.add(name)
    return names, vsym


SYMBOL_NAMES: dict[str, str] = {}
WAVES = 1
# scheduling fences: after every block whose closure has more than LIGHT_CLOSURE intermediates, and after every
# FENCE_GROUP-th light block (light gate blocks may overlap each other's exp/table-lookup latencies)
import os  # noqa: E402
FENCE_GROUP = int(os.environ.get("GEN_FENCE_GROUP", "1"))
LIGHT_CLOSURE = int(os.environ.get("GEN_LIGHT_CLOSURE", "12"))


def build(spec: OdeSpec):
    states, params = spec.state_names, spec.parameter_names
    sym = {n: sympy.Symbol(n, real=True) for n in states + params}
    ns = {
        "exp": sympy.exp, "log": sympy.log, "sqrt": sympy.sqrt, "floor": sympy.floor, "abs": sympy.Abs,
        "Conditional": lambda c, a, b: sympy.Piecewise((a, c), (b, True)),
        "Lt": sympy.Lt, "Le": sympy.Le, "Gt": sympy.Gt, "Ge": sympy.Ge, "Eq": sympy.Eq, "And": sympy.And,
        "Or": sympy.Or, "time": sympy.Symbol("t", real=True), "pi": sympy.pi,
    }
    ns.update(sym)
    exprs: dict[str, sympy.Expr] = {}
    order: list[str] = []
    for name, node in spec.assignments:
        e = sympy.sympify(eval(compile(ast.Expression(node), "<ode>", "eval"), {"__builtins__": {}}, ns))
        exprs[name] = e
        order.append(name)
        ns[name] = sympy.Symbol(name, real=True)
    guarded, vsym = ghk_singular_inputs(exprs, states)
    if guarded:
        vg = sympy.Symbol(f"{vsym.name}_ghk", real=True)
        for name in guarded:
            exprs[name] = exprs[name].subs(vsym, vg)
        exprs[vg.name] = beat_guard(vsym)
        order.insert(0, vg.name)
        print(f"Goldman-Hodgkin-Katz inputs evaluated at the guarded potential {vg.name}:", ", ".join(sorted(guarded)))
    # transitive state dependencies of every intermediate
    dep: dict[str, set[str]] = {}
    for name in order:
        d = set()
        for s in exprs[name].free_symbols:
            if s.name in states:
                d.add(s.name)
            elif s.name in dep:
                d |= dep[s.name]
        dep[name] = d
    return exprs, order, dep


def derivative_statements(exprs, order, dep, states):
    """Forward-mode chain rule: for each state y, statements d<u>_d<y> for the intermediates u between y and
    d<y>_dt, ending with J_<y>."""
    stmts: list[tuple[str, sympy.Expr]] = []
    jac: dict[str, sympy.Expr | None] = {}
    for y in states:
        target = f"d{y}_dt"
        ysym = sympy.Symbol(y, real=True)
        # intermediates the target depends on (transitively)
        needed, stack = set(), [target]
        while stack:
            u = stack.pop()
            if u in needed:
                continue
            needed.add(u)
            stack += [s.name for s in exprs[u].free_symbols if s.name in exprs]
        dsym: dict[str, sympy.Expr] = {}
        for u in order:
            if u not in needed or y not in dep[u]:
                continue
            e = exprs[u]
            total = sympy.diff(e, ysym)
            for w in e.free_symbols:
                if w.name in dsym:
                    total += sympy.diff(e, w) * dsym[w.name]
            if u == target:
                jac[y] = total
            else:
                name = f"d{u}_d{y}"
                if total == 0:
                    continue
                stmts.append((name, total))
                dsym[u] = sympy.Symbol(name, real=True)
        jac.setdefault(y, None)
        if jac[y] is not None and jac[y] == 0:
            jac[y] = None
    return stmts, jac


def needed_intermediates(exprs, order, roots):
    need, stack = set(), list(roots)
    while stack:
        u = stack.pop()
        if u in need or u not in exprs:
            continue
        need.add(u)
        stack += [s.name for s in exprs[u].free_symbols if s.name in exprs]
    return [u for u in order if u in need]


def main():
    ode, struct, out_h, out_py = sys.argv[1:5]
    spec = OdeSpec(ode)
    states, params = spec.state_names, spec.parameter_names
    exprs, order, dep = build(spec)
    dstmts, jac = derivative_statements(exprs, order, dep, states)
    dnames = {n for n, _ in dstmts}
    roots = [f"d{y}_dt" for y in states]
    for _, e in dstmts:
        roots += [s.name for s in e.free_symbols if s.name in exprs]
    for y, j in jac.items():
        if j is not None:
            roots += [s.name for s in j.free_symbols if s.name in exprs]
    live = needed_intermediates(exprs, order, roots)

    for i, p in enumerate(params):
        SYMBOL_NAMES[p] = f"p[{i}]"
    pr = Printer()
    lines = []
    w = lines.append
    w(f"// GENERATED by tools/gen_cell_model.py from the model specification {Path(ode).name} -- do not edit.")
    w("// One generalized Rush-Larsen step per node; J_i = total d f_i / d y_i by forward-mode differentiation")
    w("// through the model's intermediate expressions (see oracle/ionic.py for the scheme and its pin).")
    w("#pragma once")
    w('#include "../ionic_models.h"')
    w("")
    w(f"struct {struct} {{")
    v_index = next((i for i, name in enumerate(states) if name.lower() == "v"), 0)
    w(f"  static constexpr int NS = {len(states)}, NP = {len(params)}, V_INDEX = {v_index};  // V_INDEX: membrane potential")
    w("  static constexpr bool REGISTER_LOOP = false;  // see ode_run_kernel")
    w(f"  static constexpr int WAVES = {WAVES};           // waves per SIMD the kernels are compiled for")
    w("  struct Derived {};")
    w("  __host__ __device__ static Derived derive(const double*) { return {}; }")
    w("  template <class IO>")
    w("  __device__ static __forceinline__ void step(const IO& io, const double* p, const Derived&, const FastMath& fm, double t,")
    w("                              double dt) {")
    for i, s in enumerate(states):
        w(f"    const double {s} = io.load({i});")
    # Emission order: states sorted by how many intermediates their update needs (the V-only gates first, the
    # membrane potential and the ion concentrations last); before each state's block only the intermediates
    # not emitted yet are computed, so few values are live at any point.  Each block ends with the store and
    # a scheduling fence.
    def closure(names):
        need, stack = set(), list(names)
        while stack:
            u = stack.pop()
            if u in need or u not in exprs:
                continue
            need.add(u)
            stack += [sy.name for sy in exprs[u].free_symbols if sy.name in exprs]
        return need

    per_state = {}
    for y in states:
        roots_y = [f"d{y}_dt"]
        for name, e in dstmts:
            if name.endswith(f"_d{y}"):
                roots_y += [sy.name for sy in e.free_symbols if sy.name in exprs]
        if jac[y] is not None:
            roots_y += [sy.name for sy in jac[y].free_symbols if sy.name in exprs]
        per_state[y] = closure(roots_y)
    emitted = set()
    nblock = 0
    for y in sorted(states, key=lambda q: len(per_state[q])):
        i = states.index(y)
        for u in order:
            if u in per_state[y] and u not in emitted:
                w(f"    const double {u} = {pr.doprint(exprs[u])};")
                emitted.add(u)
        w("    {")
        for name, e in dstmts:
            if name.endswith(f"_d{y}"):
                w(f"      const double {name} = {pr.doprint(e)};")
        if jac[y] is None:
            w(f"      io.store({i}, {y} + dt * d{y}_dt);")
        else:
            w(f"      const double J = {pr.doprint(jac[y])};")
            # f / J in closed form where that is simpler than f (gates: (inf - y)/tau over -1/tau = y - inf)
            f_e = exprs[f"d{y}_dt"]
            ratio = sympy.cancel(f_e / jac[y]) if sympy.count_ops(jac[y]) <= 6 else None
            if ratio is not None and sympy.count_ops(ratio) < sympy.count_ops(f_e):
                quot = f"({pr.doprint(ratio)})"
            else:
                quot = f"d{y}_dt * beat_rcp(J)"
            w(f"      io.store({i}, {y} + ((fabs(J) > 1e-8) ? {quot} * (fm.exp(J * dt) - 1.0) : d{y}_dt * dt));")
        w("    }")
        nblock += 1
        if len(per_state[y]) > LIGHT_CLOSURE or nblock % FENCE_GROUP == 0:
            w("    __builtin_amdgcn_sched_barrier(0);")
    w("  }")
    w("};")
    Path(out_h).parent.mkdir(parents=True, exist_ok=True)
    Path(out_h).write_text("\n".join(lines) + "\n")

    py = [f'"""GENERATED by tools/gen_cell_model.py from {Path(ode).name}: state / parameter names and defaults',
          '(order of appearance in the model specification)."""', "", "STATES = {"]
    py += [f"    {s!r}: {spec.states[s]!r}," for s in states] + ["}", "", "PARAMETERS = {"]
    py += [f"    {p!r}: {spec.parameters[p]!r}," for p in params] + ["}", ""]
    Path(out_py).write_text("\n".join(py))
    nz = sum(1 for y in states if jac[y] is not None)
    print(f"{struct}: {len(states)} states ({nz} Rush-Larsen, {len(states) - nz} forward Euler), {len(params)} parameters, "
          f"{len(live)} intermediates, {len(dstmts)} derivative statements ({len(dnames)} names)")


if __name__ == "__main__":
    main()
