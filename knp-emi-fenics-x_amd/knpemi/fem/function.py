"""`Function`-style I/O objects (the "FEniCSx Function I/O" of the drop-in boundary).

The reference passes `dolfinx.fem.Function`s around and only ever touches
`.x.array`, `.x.scatter_forward()`, `.name`, `.function_space`, `.interpolate`
(`utils.py:99-100,253-254,292-293`, `odeSolver.py:142,159-164`,
`emiWeakForm.py:66-79`).  These classes expose exactly that surface over flat
fp64 numpy arrays: `Function.x.array` has one entry per local (owned + ghost)
vertex of the sub-mesh, in sub-mesh vertex order.
"""
from __future__ import annotations

import numpy as np


class Constant:
    """`dolfinx.fem.Constant` stand-in: `float(c)` and `c.value`."""

    def __init__(self, mesh, value):
        self.mesh = mesh
        self.value = np.asarray(value, dtype=np.float64)

    def __float__(self):
        return float(self.value)

    def __add__(self, other):
        return float(self) + float(other)

    __radd__ = __add__

    def __repr__(self):
        return f"Constant({self.value!r})"


def as_float(v):
    """Scalars, `Constant`s and 0-d arrays -> float."""
    return float(v.value) if isinstance(v, Constant) else float(v)


class _Element:
    def __init__(self, space):
        self._space = space

    @property
    def interpolation_points(self):
        return None


class FunctionSpace:
    """CG-1 space (P1 on simplices, Q1 on tensor cells): dofs == mesh vertices."""

    def __init__(self, mesh, element=("CG", 1)):
        family, degree = element
        if family not in ("CG", "Lagrange", "P", "Q") or degree != 1:
            raise NotImplementedError("the MI355X hot path implements CG-1 only "
                                      "(the reference runs degree=1 everywhere)")
        self.mesh = mesh
        self.element = _Element(self)

    @property
    def num_dofs(self):
        return self.mesh.num_vertices

    def tabulate_dof_coordinates(self):
        x = np.zeros((self.mesh.num_vertices, 3))
        x[:, :self.mesh.gdim] = self.mesh.x
        return x

    def clone(self):
        return FunctionSpace(self.mesh)


def functionspace(mesh, element=("CG", 1)):
    return FunctionSpace(mesh, element)


class Vector:
    """`Function.x`: `.array` plus `scatter_forward()` (owner -> ghost update).

    Every access to `.array` bumps `version` (the caller may write through the
    returned view), which is how the device layer knows when a host array has
    to be uploaded again; the device layer itself reads/writes `_a` directly.
    """

    def __init__(self, n, mesh):
        self._a = np.zeros(n, dtype=np.float64)
        self._mesh = mesh
        self.version = 0

    @property
    def array(self):
        self.version += 1
        return self._a

    @array.setter
    def array(self, value):
        self.version += 1
        self._a[:] = value

    def scatter_forward(self):
        halo = getattr(self._mesh, "halo", None)
        if halo is not None:
            self.version += 1
            halo.forward_host(self._a)


class Function:
    def __init__(self, V, name="f"):
        self.function_space = V
        self.x = Vector(V.num_dofs, V.mesh)
        self.name = name

    def interpolate(self, f):
        """`f` is a callable of x with shape (3, n) (DOLFINx convention)."""
        xt = self.function_space.tabulate_dof_coordinates().T
        self.x.array[:] = np.asarray(f(xt), dtype=np.float64)

    def copy(self):
        g = Function(self.function_space, self.name)
        g.x.array[:] = self.x.array
        return g
