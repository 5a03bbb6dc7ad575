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


# ---------------------------------------------------------------------------------------------------------------
# Membrane variant: the unit square with ICS = [0.25, 0.75]^2 (make_mesh_mms).  u = cos(2 pi x) cos(2 pi y) has a
# vanishing gradient at the four corners of the ICS, so the normal fluxes below are single-valued at every Q node.
#   c_k = A_k + B_k u on both sides, phi_i = P u, phi_e = P u - PHI0   ([phi] = PHI0),
#   phi_M(previous step) = PHI0 + delta(x),
#   membrane law  F z_k J_k . n_i = I_ch_k + alpha_k C_M ([phi] - phi_M_prev) / dt     (knpWeakForm.py:178-214)
#     => I_ch_k = F z_k J_k . n_i + alpha_k C_M delta / dt,    alpha_k = D z_k^2 c_k / sum_j D z_j^2 c_j,
#   (alpha_k evaluated with the previous-step fields, as the forms do) and c_prev = c + dt div J_k absorbs the volume source on both sides (the forms have no ICS source term).
# One implicit Euler step must return c: this pins the rational membrane terms (alpha, C, g, the signs on either
# side) of b_knp.  Splitting scheme: the potential handed over is the one the ODE step has already advanced,
# phi_M_prev - (dt / C_M) sum_k I_ch_k (all three ions).
# ---------------------------------------------------------------------------------------------------------------
F_CONST = C_M = 1.0
DT_M = 0.01                       # small enough for c_prev = c + dt div J to stay positive
A_M = (2.0, 4.0)
B_M = (0.5, -0.7)
PHI0 = 0.3
DELTA0 = 0.2
um = sp.cos(2 * sp.pi * x) * sp.cos(2 * sp.pi * y)
phim = P0 * um
cm = [A_M[0] + B_M[0] * um, A_M[1] + B_M[1] * um]
cm.append(-(Z[0] * cm[0] + Z[1] * cm[1]) / Z[2])
delta = DELTA0 * (1 + sp.cos(2 * sp.pi * x) ** 2 + sp.cos(2 * sp.pi * y) ** 2) / 3
Jm = [[-D * sp.diff(cm[k], v) - Z[k] * PSI * D * cm[k] * sp.diff(phim, v) for v in (x, y)] for k in range(3)]
divJ = [sp.diff(Jm[k][0], x) + sp.diff(Jm[k][1], y) for k in range(3)]
# the forms evaluate alpha with the fields of the previous step (c_prev of the solved ions, the eliminated ion as given)
cprev = [cm[0] + DT_M * divJ[0], cm[1] + DT_M * divJ[1], cm[2]]
asum = sum(D * Z[k] ** 2 * cprev[k] for k in range(3))
alpha = [D * Z[k] ** 2 * cprev[k] / asum for k in range(3)]

M_C = [_fn(ck) for ck in cm]
M_CPREV = [_fn(cprev[k]) for k in range(2)]
M_PHI = _fn(phim)
M_DELTA = _fn(delta)
_Jx = [_fn(Jm[k][0]) for k in range(3)]
_Jy = [_fn(Jm[k][1]) for k in range(3)]
_AL = [_fn(alpha[k]) for k in range(3)]


def ics_normals(X):
    """Outward unit normal of [0.25, 0.75]^2 at boundary points X [2, n] (corners: either side, the fluxes vanish)."""
    d = np.stack([np.abs(X[0] - 0.25), np.abs(X[0] - 0.75), np.abs(X[1] - 0.25), np.abs(X[1] - 0.75)])
    side = np.argmin(d, axis=0)
    n = np.zeros_like(X)
    n[0] = np.where(side == 0, -1.0, np.where(side == 1, 1.0, 0.0))
    n[1] = np.where(side == 2, -1.0, np.where(side == 3, 1.0, 0.0))
    return n


def channel_currents(XQ):
    """I_ch_k (k = 0, 1, 2) at the membrane points XQ [2, n]."""
    n = ics_normals(XQ)
    out = []
    for k in range(3):
        jn = _Jx[k](XQ) * n[0] + _Jy[k](XQ) * n[1]
        out.append(F_CONST * Z[k] * jn + _AL[k](XQ) * C_M * M_DELTA(XQ) / DT_M)
    return out


def membrane_potential_prev(XQ, splitting):
    pm = PHI0 + M_DELTA(XQ)
    if splitting:
        pm = pm - (DT_M / C_M) * sum(channel_currents(XQ))
    return pm


# ---------------------------------------------------------------------------------------------------------------
# 3D volume variant (tetrahedra and Q1 hexahedra): the same steady state on the unit cube,
# u = cos(pi x) cos(pi y) cos(pi z), no-flux boundary.
# ---------------------------------------------------------------------------------------------------------------
zz = sp.symbols("z")
u3 = sp.cos(sp.pi * x) * sp.cos(sp.pi * y) * sp.cos(sp.pi * zz)
phi3 = P0 * u3
c3 = [A[k] + B[k] * u3 for k in range(2)]
c3_elim = -(Z[0] * c3[0] + Z[1] * c3[1]) / Z[2]


def _fn3(expr):
    f = sp.lambdify((x, y, zz), expr, "numpy")
    return lambda X: f(X[0], X[1], X[2]) + 0.0 * X[0]


def _div_flux3(ck, zk):
    J = [-D * sp.diff(ck, v) - zk * PSI * D * ck * sp.diff(phi3, v) for v in (x, y, zz)]
    return sum(sp.diff(J[i], v) for i, v in enumerate((x, y, zz)))


PHI3 = _fn3(phi3)
C3_EXACT = [_fn3(ck) for ck in c3]
C3_ELIM = _fn3(c3_elim)
F3_SOURCE = [_fn3(_div_flux3(c3[k], Z[k])) for k in range(2)]


def nodal_rms_error(values, exact_values):
    """Root mean square of the nodal error (uniform grids: a discrete L2 norm)."""
    return float(np.sqrt(np.mean((values - exact_values) ** 2)))


# ---------------------------------------------------------------------------------------------------------------
# EMI volume check without a source term: with c_2, c_3 constant and c_1 = C exp(-psi phi) - (D_2 c_2 + D_3 c_3) / D_1
# (z_k^2 = 1) the identity  F sum_k z_k D_k grad c_k = -kappa grad phi,  kappa = F psi sum_k z_k^2 D_k c_k,  holds
# pointwise, so the EMI equation  div(kappa grad phi) + F sum_k z_k div(D_k grad c_k) = 0  (emiWeakForm.py:138-241)
# is satisfied by phi itself, natural boundary condition included: assembling A_emi, b_emi from these
# concentrations and solving must return phi up to a constant at second order.
# ---------------------------------------------------------------------------------------------------------------
EMI_C = 10.0
EMI_C23 = (2.0, 2.0)


def emi_exact(X):
    """(phi, [c_1, c_2, c_3]) at points X [d, n] (d = 2 or 3)."""
    u_ = np.prod(np.cos(np.pi * X), axis=0)
    ph = P0 * u_
    c1 = EMI_C * np.exp(-PSI * ph) - (EMI_C23[0] + EMI_C23[1])
    return ph, [c1, np.full_like(ph, EMI_C23[0]), np.full_like(ph, EMI_C23[1])]


# ---------------------------------------------------------------------------------------------------------------
# 3D membrane variant: unit cube, ICS = [0.25, 0.75]^3, u = cos(2 pi x) cos(2 pi y) cos(2 pi z) (its gradient vanishes
# on the edges and corners of the inner cube, so the normal fluxes are single-valued at every membrane vertex).
# Same construction as the 2D membrane variant; checks the triangular (tetrahedra) and quadrilateral (hexahedra)
# membrane-facet kernels.
# ---------------------------------------------------------------------------------------------------------------
um3 = sp.cos(2 * sp.pi * x) * sp.cos(2 * sp.pi * y) * sp.cos(2 * sp.pi * zz)
phim3 = P0 * um3
cm3 = [A_M[0] + B_M[0] * um3, A_M[1] + B_M[1] * um3]
cm3.append(-(Z[0] * cm3[0] + Z[1] * cm3[1]) / Z[2])
delta3 = DELTA0 * (1 + sp.cos(2 * sp.pi * x) ** 2 + sp.cos(2 * sp.pi * y) ** 2 + sp.cos(2 * sp.pi * zz) ** 2) / 4
DT_M3 = 0.005
Jm3 = [[-D * sp.diff(cm3[k], v) - Z[k] * PSI * D * cm3[k] * sp.diff(phim3, v) for v in (x, y, zz)] for k in range(3)]
divJ3 = [sum(sp.diff(Jm3[k][i], v) for i, v in enumerate((x, y, zz))) for k in range(3)]
cprev3 = [cm3[0] + DT_M3 * divJ3[0], cm3[1] + DT_M3 * divJ3[1], cm3[2]]
asum3 = sum(D * Z[k] ** 2 * cprev3[k] for k in range(3))
alpha3 = [D * Z[k] ** 2 * cprev3[k] / asum3 for k in range(3)]
M3_C = [_fn3(ck) for ck in cm3]
M3_CPREV = [_fn3(cprev3[k]) for k in range(2)]
M3_PHI = _fn3(phim3)
M3_DELTA = _fn3(delta3)
_J3 = [[_fn3(Jm3[k][i]) for i in range(3)] for k in range(3)]
_AL3 = [_fn3(alpha3[k]) for k in range(3)]


def ics_normals3(X):
    """Outward unit normal of [0.25, 0.75]^3 at boundary points X [3, n] (edges / corners: any adjacent face)."""
    d = np.stack([np.abs(X[a] - v) for a in range(3) for v in (0.25, 0.75)])
    side = np.argmin(d, axis=0)
    n = np.zeros_like(X)
    for a in range(3):
        n[a] = np.where(side == 2 * a, -1.0, np.where(side == 2 * a + 1, 1.0, 0.0))
    return n


def channel_currents3(XQ):
    n = ics_normals3(XQ)
    out = []
    for k in range(3):
        jn = sum(_J3[k][i](XQ) * n[i] for i in range(3))
        out.append(F_CONST * Z[k] * jn + _AL3[k](XQ) * C_M * M3_DELTA(XQ) / DT_M3)
    return out


def membrane_potential_prev3(XQ, splitting):
    pm = PHI0 + M3_DELTA(XQ)
    if splitting:
        pm = pm - (DT_M3 / C_M) * sum(channel_currents3(XQ))
    return pm
