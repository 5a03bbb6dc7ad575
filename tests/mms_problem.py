"""The manufactured EMI problem of the reference's `tests/run_mms_emi.py:100-252` (unit constants,
exact solutions :166-179), with the source terms derived symbolically here (the reference derives
them in UFL, :182-214) and exposed as numpy callables for knpemi.mms."""
import numpy as np
import sympy as sp

from knpemi.fem import Constant

x, y, nx, ny = sp.symbols("x y n_x n_y")
pi = sp.pi
D = psi = F = C_phi = 1
z = dict(a=1, b=-1, c=1)

# The reference's concentrations a = sin(2 pi y) cos(2 pi x), b = cos(2 pi y) sin(2 pi x) give
# kappa = a + b + c = 2 b, which changes sign inside the domain: the manufactured problem is
# degenerate and neither the reference formulation nor this one converges on it (oracle and GPU
# reproduce the same non-converging errors to all digits).  The gate therefore shifts a and b by a
# constant so that kappa = 2 (OFFSET + cos sin) >= 2; the potentials are the reference's.
OFFSET = 2
a_i = OFFSET + sp.sin(2 * pi * y) * sp.cos(2 * pi * x)
b_i = OFFSET + sp.cos(2 * pi * y) * sp.sin(2 * pi * x)
c_i = -1 / sp.Integer(z["c"]) * (z["a"] * a_i + z["b"] * b_i)
a_e, b_e, c_e = a_i, b_i, c_i
phi_i = sp.cos(2 * pi * x) * sp.cos(2 * pi * y)
phi_e = sp.sin(2 * pi * x) * sp.sin(2 * pi * y)


def grad(f):
    return sp.Matrix([sp.diff(f, x), sp.diff(f, y)])


def div(v):
    return sp.diff(v[0], x) + sp.diff(v[1], y)


def flux(k, zk, phi):
    return -D * grad(k) - zk * D * psi * k * grad(phi)


J_i = {n: flux(k, z[n], phi_i) for n, k in (("a", a_i), ("b", b_i), ("c", c_i))}
J_e = {n: flux(k, z[n], phi_e) for n, k in (("a", a_e), ("b", b_e), ("c", c_e))}
f_phi_i = F * sum(z[n] * div(J_i[n]) for n in "abc")
f_phi_e = F * sum(z[n] * div(J_e[n]) for n in "abc")
nvec = sp.Matrix([nx, ny])
Im_intra = (F * sum((z[n] * J_i[n] for n in "abc"), sp.zeros(2, 1))).dot(nvec)
Im_extra = -(F * sum((z[n] * J_e[n] for n in "abc"), sp.zeros(2, 1))).dot(nvec)
f_phi_m = (phi_i - phi_e) - Im_intra / C_phi
f_I_M = Im_intra + Im_extra


def _fn(expr, with_normal=False):
    args = (x, y, nx, ny) if with_normal else (x, y)
    f = sp.lambdify(args, expr, "numpy")
    if with_normal:
        return lambda X, N: f(X[0], X[1], N[0], N[1]) + 0.0 * X[0]
    return lambda X: f(X[0], X[1]) + 0.0 * X[0]


MMS = {"f_phi_i": _fn(f_phi_i), "f_phi_e": _fn(f_phi_e), "f_phi_m": _fn(f_phi_m, True), "f_I_M": _fn(f_I_M, True),
       "phi_i_exact": _fn(phi_i), "phi_e_exact": _fn(phi_e)}
CONC = {"a": (_fn(a_e), _fn(a_i)), "b": (_fn(b_e), _fn(b_i)), "c": (_fn(c_e), _fn(c_i))}


class MMSMembraneModel:   # tests/run_mms_emi.py:34-35
    pass
