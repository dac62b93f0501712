"""Minimal unit registry with the ``pint`` surface the reference's helpers use
(src/beat/units.py, conductivities.py:63-98, stimulation.py:114-207): ``ureg("uA/cm**2")``,
``value * ureg("S/m")``, ``q.to("uA/mV")``, ``q.magnitude``, arithmetic between quantities.
pint itself is not a dependency of this package."""

from __future__ import annotations

import ast
import operator

_DIMS = ("length", "current", "voltage", "time")


class Unit:
    __slots__ = ("factor", "dims")

    def __init__(self, factor, dims):
        self.factor = float(factor)
        self.dims = tuple(dims)

    def __mul__(self, o):
        return Unit(self.factor * o.factor, [a + b for a, b in zip(self.dims, o.dims)])

    def __truediv__(self, o):
        return Unit(self.factor / o.factor, [a - b for a, b in zip(self.dims, o.dims)])

    def __pow__(self, k):
        return Unit(self.factor**k, [a * k for a in self.dims])


def _u(factor, **dims):
    return Unit(factor, [dims.get(d, 0) for d in _DIMS])


_BASE = {
    "m": _u(1.0, length=1), "cm": _u(1e-2, length=1), "mm": _u(1e-3, length=1), "um": _u(1e-6, length=1),
    "A": _u(1.0, current=1), "mA": _u(1e-3, current=1), "uA": _u(1e-6, current=1), "nA": _u(1e-9, current=1),
    "pA": _u(1e-12, current=1),
    "V": _u(1.0, voltage=1), "mV": _u(1e-3, voltage=1),
    "s": _u(1.0, time=1), "ms": _u(1e-3, time=1),
    "S": _u(1.0, current=1, voltage=-1), "mS": _u(1e-3, current=1, voltage=-1), "uS": _u(1e-6, current=1, voltage=-1),
    "F": _u(1.0, current=1, time=1, voltage=-1), "uF": _u(1e-6, current=1, time=1, voltage=-1),
    "pF": _u(1e-12, current=1, time=1, voltage=-1),
    "dimensionless": _u(1.0),
}
_OPS = {ast.Mult: operator.mul, ast.Div: operator.truediv, ast.Pow: operator.pow}


def _parse(text: str) -> Unit:
    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.Name):
            if node.id not in _BASE:
                raise ValueError(f"unknown unit {node.id!r}")
            return _BASE[node.id]
        if isinstance(node, ast.Constant):
            return node.value
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, ast.USub):
            return -ev(node.operand)
        if isinstance(node, ast.BinOp) and type(node.op) in _OPS:
            a, b = ev(node.left), ev(node.right)
            if isinstance(a, (int, float)) and isinstance(b, Unit):  # e.g. 1/cm
                a = Unit(a, [0] * len(_DIMS))
            return _OPS[type(node.op)](a, b)
        raise ValueError(f"cannot parse unit expression {text!r}")

    return ev(ast.parse(text.strip().replace("^", "**"), mode="eval"))


class Quantity:
    def __init__(self, magnitude, unit: Unit, text: str = ""):
        self.magnitude = magnitude
        self._unit = unit
        self._text = text

    @property
    def units(self):
        return self._text

    def to(self, unit):
        target = unit._unit if isinstance(unit, Quantity) else _parse(unit)
        if any(abs(a - b) > 1e-12 for a, b in zip(target.dims, self._unit.dims)):
            raise ValueError(f"cannot convert {self._text or self._unit.dims} to {unit}")
        return Quantity(self.magnitude * (self._unit.factor / target.factor), target, unit if isinstance(unit, str) else unit._text)

    def _coerce(self, o):
        return o if isinstance(o, Quantity) else Quantity(o, _BASE["dimensionless"])

    @staticmethod
    def _label(a: str, op: str, b: str) -> str:
        if not b:
            return a
        if not a:
            return b if op == "*" else f"1/({b})"
        return f"({a}){op}({b})"

    def __mul__(self, o):
        o = self._coerce(o)
        return Quantity(self.magnitude * o.magnitude, self._unit * o._unit, self._label(self._text, "*", o._text))

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = self._coerce(o)
        return Quantity(self.magnitude / o.magnitude, self._unit / o._unit, self._label(self._text, "/", o._text))

    def __eq__(self, o):
        """Equal physical quantities compare equal whatever units they are written in (as in pint)."""
        o = self._coerce(o) if not isinstance(o, Quantity) else o
        if any(abs(a - b) > 1e-12 for a, b in zip(self._unit.dims, o._unit.dims)):
            return False
        a, b = self.magnitude * self._unit.factor, o.magnitude * o._unit.factor
        return bool(abs(a - b) <= 1e-12 * max(abs(a), abs(b), 1e-300))

    __hash__ = None

    def __rtruediv__(self, o):
        return self._coerce(o) / self

    def __add__(self, o):
        o = self._coerce(o)
        return Quantity(self.magnitude + o.to(self).magnitude, self._unit, self._text)

    __radd__ = __add__

    def __sub__(self, o):
        o = self._coerce(o)
        return Quantity(self.magnitude - o.to(self).magnitude, self._unit, self._text)

    def __pow__(self, k):
        return Quantity(self.magnitude**k, self._unit**k)

    def __float__(self):
        return float(self.magnitude)

    def __repr__(self):
        return f"<Quantity({self.magnitude}, '{self._text}')>"


class UnitRegistry:
    Quantity = Quantity

    def __call__(self, text: str) -> Quantity:
        return Quantity(1.0, _parse(text), text)


ureg = UnitRegistry()


def to_quantity(value, unit: str) -> Quantity:
    if isinstance(value, Quantity):
        return value.to(unit)
    return value * ureg(unit)
