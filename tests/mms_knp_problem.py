"""A manufactured KNP problem (the reference's own `tests/run_mms.py` is unfinished): steady
concentrations c_k = A_k + B_k u and potential phi = P u, u = cos(pi x) cos(pi y), on the unit square with
no-flux boundaries (grad u . n = 0 there) and unit constants.  One implicit-Euler step of knpWeakForm.py:123-166
started from c_prev = c_exact with the source f_k = div J_k, J_k = -D grad c_k - z_k psi D c_k grad phi, must return
c_exact up to the discretisation error: second order in L2.  This pins the mass, diffusion, drift (sign and scale)
and source terms of the KNP forms; the membrane terms are not part of it (the mesh has no cells)."""
import numpy as np
import sympy as sp

x, y = sp.symbols("x y")
D = PSI = 1.0
DT = 0.5
Z = (1.0, -1.0, 1.0)              # two solved ions and the eliminated one
A = (2.0, 3.0)
B = (0.5, -0.7)
P0 = 0.8
u = sp.cos(sp.pi * x) * sp.cos(sp.pi * y)
phi = P0 * u
c = [A[k] + B[k] * u for k in range(2)]
c_elim = -(Z[0] * c[0] + Z[1] * c[1]) / Z[2]


def _div_flux(ck, zk):
    J = [-D * sp.diff(ck, v) - zk * PSI * D * ck * sp.diff(phi, v) for v in (x, y)]
    return sp.diff(J[0], x) + sp.diff(J[1], y)


def _fn(expr):
    f = sp.lambdify((x, y), expr, "numpy")
    return lambda X: f(X[0], X[1]) + 0.0 * X[0]


PHI = _fn(phi)
C_EXACT = [_fn(ck) for ck in c]
C_ELIM = _fn(c_elim)
F_SOURCE = [_fn(_div_flux(c[k], Z[k])) for k in range(2)]


def l2_error_p1(mesh, values, exact):
    """L2 norm of (P1 function - exact) with a degree-4 rule on every triangle."""
    pts = np.array([[1 / 3, 1 / 3], [0.6, 0.2], [0.2, 0.6], [0.2, 0.2]])
    wts = np.array([-27 / 96, 25 / 96, 25 / 96, 25 / 96])
    X = mesh.x[mesh.cells]                      # [nc, 3, 2]
    e1, e2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
    det = np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])
    err = 0.0
    for (a, b), w in zip(pts, wts):
        lam = np.array([1 - a - b, a, b])
        xq = np.einsum("j,cjd->cd", lam, X)
        uh = values[mesh.cells] @ lam
        err += w * np.sum(det * (uh - exact(xq.T)) ** 2)
    return float(np.sqrt(err))
