"""Manufactured problems for the DG(P1)+SIP variant, written against a small backend interface so
that the same cases run on the CPU restatement (oracle/knpemi_dg_oracle.py, tests/test_dg_oracle.py) and on the HIP
kernels (knpemi.dg.DGProblem, tests/test_dg_gpu.py).  The analytic fields are those of tests/mms_knp_problem.py.

A backend provides
    X          (nc, nv, d)  coordinates of the broken dofs,          XM (nmf, nf, d) of the membrane nodes,
    cell_sub   (nc,),                                                vol (nc,) cell measures,
    emi(params, ions, c_all, phi_M, I_ch, splitting)  -> (A csr, b),
    knp(params, ions, c_all, phi, phi_M, I_ch, splitting, f_source) -> ([A_k], b (K-1, n)).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import mms_knp_problem as K


def make_mesh(dim, M, membrane, cell="simplex"):
    """Unit square / cube with M cells per edge; cell = "hexahedron": Q1 hexahedra (dim 3)."""
    from knpemi.fem import create_box, create_unit_square
    from knpemi.fem.idealized import _tag
    mesh = create_unit_square(None, M, M) if dim == 2 else create_box(
        None, [np.zeros(3), np.ones(3)], (M, M, M), "hexahedron" if cell == "hexahedron" else "tetrahedron")
    nfv = mesh.facets.shape[1]
    if membrane:
        ct, ft = _tag(mesh, [([0.25] * dim, [0.75] * dim)], [1], full_facet_tags=False)
        sel = ft.values == 1
        return mesh, ct.dense(), mesh.facets[ft.indices[sel]], ft.values[sel]
    return mesh, np.zeros(mesh.num_cells, np.int32), np.zeros((0, nfv), np.int32), np.zeros(0, np.int32)


def l2_error(be, uh, exact):
    """L2 norm of (broken P1 field uh (nc, nv) - exact) with the vertex + centroid rule (degree 2 / 3); broken Q1 on
    hexahedra: the 3 x 3 x 3 Gauss rule through the trilinear map."""
    nv = uh.shape[1]
    if nv == 8:
        g, w = np.polynomial.legendre.leggauss(3)
        g, w = 0.5 * (g + 1.0), 0.5 * w
        tot = 0.0
        for a, wa in zip(g, w):
            for b_, wb in zip(g, w):
                for c_, wc in zip(g, w):
                    N = np.array([(a if j & 1 else 1 - a) * (b_ if j & 2 else 1 - b_) * (c_ if j & 4 else 1 - c_)
                                  for j in range(8)])
                    Xq = np.einsum("j,cjd->cd", N, be.X)
                    tot = tot + wa * wb * wc * be.vol * (uh @ N - exact(Xq.T)) ** 2
        return float(np.sqrt(np.sum(tot)))
    d = nv - 1
    Xc = be.X.mean(axis=1)
    ev = (uh - exact(be.X.reshape(-1, d).T).reshape(uh.shape)) ** 2
    ec = (uh.mean(axis=1) - exact(Xc.T)) ** 2
    wv, wc = {2: (1 / 12, 3 / 4), 3: (1 / 20, 4 / 5)}[d]
    return float(np.sqrt(np.sum(be.vol * (wv * ev.sum(axis=1) + wc * ec))))


def solve(A, b):
    """Sparse LU for small systems, ILU-preconditioned BiCGStab (to 1e-12) for the 3D ones where LU fill explodes."""
    A = A.tocsc()
    if A.shape[0] < 20000:
        return spla.spsolve(A, b)
    ilu = spla.spilu(A, drop_tol=1e-4, fill_factor=10)
    x, info = spla.bicgstab(A, b, M=spla.LinearOperator(A.shape, ilu.solve), rtol=1e-12, atol=0.0, maxiter=2000)
    assert info == 0 and np.linalg.norm(A @ x - b) <= 1e-10 * np.linalg.norm(b)
    return x


def solve_pinned(A, b):
    """Pure Neumann potential problem: pin dof 0 to zero (the right-hand side is compatible)."""
    A = sp.lil_matrix(A)
    A[0, :] = 0.0
    A[0, 0] = 1.0
    b = b.copy()
    b[0] = 0.0
    return solve(A, b)


def ions_unit(n_sub=2):
    return [dict(name=n, z=z, D=[K.D] * n_sub) for n, z in zip("abc", K.Z)]


def emi_boltzmann(be):
    """Concentrations in Boltzmann equilibrium with phi = P u: phi solves the potential equation (no membrane)."""
    d = be.X.shape[2]
    ph, cs = K.emi_exact(be.X.reshape(-1, d).T)
    shape = be.X.shape[:2]
    params = dict(dt=1.0, F=1.0, psi=K.PSI, C_M=1.0)
    nf = be.XM.shape[1]
    A, b = be.emi(params, ions_unit(), [c.reshape(shape) for c in cs], np.zeros((0, nf)), [np.zeros((0, nf))] * 3, True)
    x = solve_pinned(A, b).reshape(shape)
    exact = lambda X: K.emi_exact(X)[0]
    w = np.repeat(be.vol / shape[1], shape[1])
    shift = np.sum(w * (x.ravel() - ph)) / np.sum(w)
    return l2_error(be, x - shift, exact), A, b


def emi_membrane(be, splitting):
    """Boltzmann equilibrium on both sides and phi_M_prev = PHI0: phi_e = P u, phi_i = P u + PHI0."""
    d = be.X.shape[2]
    shape = be.X.shape[:2]
    ph, cs = K.emi_exact(be.X.reshape(-1, d).T)
    exact = ph.reshape(shape) + np.where(be.cell_sub[:, None] > 0, K.PHI0, 0.0)
    nmf, nf = be.XM.shape[:2]
    phi_M = np.full((nmf, nf), K.PHI0)
    I_ch = [np.zeros((nmf, nf)) for _ in range(3)]
    if not splitting:   # emiWeakForm.py:236: g = phi_M_prev - I_ch / C_phi, compensated in phi_M_prev
        I_tot = 0.3 + 0.2 * np.cos(2 * np.pi * be.XM[:, :, 0])
        I_ch[0], I_ch[2] = 0.7 * I_tot, 0.3 * I_tot
        phi_M = phi_M + I_tot
    params = dict(dt=1.0, F=1.0, psi=K.PSI, C_M=1.0)
    A, b = be.emi(params, ions_unit(), [c.reshape(shape) for c in cs], phi_M, I_ch, splitting)
    x = solve_pinned(A, b).reshape(shape)
    w = be.vol[:, None] / shape[1] * np.ones(shape)
    shift = np.sum(w * (x - exact)) / np.sum(w)
    err = float(np.sqrt(np.sum(w * (x - shift - exact) ** 2)))
    ins = be.cell_sub > 0
    jump = (np.sum(w[ins] * (x - exact)[ins]) / np.sum(w[ins])) - (np.sum(w[~ins] * (x - exact)[~ins]) / np.sum(w[~ins]))
    return err, abs(jump), A, b


def knp_volume(be):
    """One implicit Euler step from the manufactured steady state with its source term returns that state."""
    d = be.X.shape[2]
    shape = be.X.shape[:2]
    Xf = be.X.reshape(-1, d).T
    if d == 2:
        ce, cel, ph, src = K.C_EXACT, K.C_ELIM, K.PHI, K.F_SOURCE
    else:
        ce, cel, ph, src = K.C3_EXACT, K.C3_ELIM, K.PHI3, K.F3_SOURCE
    c_all = [ce[0](Xf).reshape(shape), ce[1](Xf).reshape(shape), cel(Xf).reshape(shape)]
    params = dict(dt=K.DT, F=1.0, psi=K.PSI, C_M=1.0)
    f = {k: src[k](Xf).reshape(shape) for k in range(2)}
    nf = be.XM.shape[1]
    As, b = be.knp(params, ions_unit(), c_all, ph(Xf).reshape(shape), np.zeros((0, nf)), [np.zeros((0, nf))] * 3, True, f)
    errs = []
    for k in range(2):
        x = solve(As[k], b[k]).reshape(shape)
        errs.append(l2_error(be, x, ce[k]))
    return errs, As, b


def knp_membrane(be, splitting):
    """Membrane variant: pins the rational membrane terms of b_knp on both sides."""
    d = be.X.shape[2]
    shape = be.X.shape[:2]
    Xf = be.X.reshape(-1, d).T
    nmf, nf = be.XM.shape[:2]
    XQ = be.XM.reshape(-1, d).T
    if d == 2:
        C, CP, PH, cur, pm, dt = K.M_C, K.M_CPREV, K.M_PHI, K.channel_currents, K.membrane_potential_prev, K.DT_M
    else:
        C, CP, PH, cur, pm, dt = K.M3_C, K.M3_CPREV, K.M3_PHI, K.channel_currents3, K.membrane_potential_prev3, K.DT_M3
    # the normals of the manufactured currents are those of the facets, so evaluate them facet by facet
    I = [i.reshape(nmf, nf) for i in cur_on_facets(be, cur)]
    phi_M = pm_on_facets(be, pm, splitting, I, dt)
    c_all = [CP[0](Xf).reshape(shape), CP[1](Xf).reshape(shape), C[2](Xf).reshape(shape)]
    phi = PH(Xf).reshape(shape) - np.where(be.cell_sub[:, None] == 0, K.PHI0, 0.0)
    params = dict(dt=dt, F=K.F_CONST, psi=K.PSI, C_M=K.C_M)
    As, b = be.knp(params, ions_unit(), c_all, phi, phi_M, I, splitting, None)
    errs = []
    for k in range(2):
        x = solve(As[k], b[k]).reshape(shape)
        errs.append(l2_error(be, x, C[k]))
    return errs, As, b


def cur_on_facets(be, cur):
    d = be.XM.shape[2]
    nmf, nf = be.XM.shape[:2]
    # tests/mms_knp_problem.py picks the normal from the nearest side of the inner box; evaluating at a point pulled
    # slightly towards the facet centroid makes that choice the facet's own side at edge and corner nodes too
    cen = be.XM.mean(axis=1, keepdims=True)
    Xin = (be.XM + 1e-6 * (cen - be.XM)).reshape(-1, d).T
    return cur(Xin)


def pm_on_facets(be, pm, splitting, I, dt):
    d = be.XM.shape[2]
    nmf, nf = be.XM.shape[:2]
    XQ = be.XM.reshape(-1, d).T
    base = pm(XQ, False).reshape(nmf, nf)
    if splitting:
        base = base - (dt / K.C_M) * sum(I)
    return base


class OracleBackend:
    def __init__(self, dim, M, membrane, gamma=10.0, cell="simplex"):
        import knpemi_dg_oracle as dg
        mesh, cell_sub, mfac, mtag = make_mesh(dim, M, membrane, cell)
        self.o = dg.make_dg_oracle(mesh.x, mesh.cells, mesh.cell_type, cell_sub, mfac, mtag)
        self.X = mesh.x[mesh.cells]
        self.XM = mesh.x[mfac].reshape(len(mfac), mfac.shape[1], dim)
        self.cell_sub = np.asarray(cell_sub)
        self.vol = self.o.vol
        self.gamma = gamma

    def emi(self, params, ions, c_all, phi_M, I_ch, splitting):
        return self.o.assemble_emi(params, ions, c_all, phi_M, I_ch, splitting_scheme=splitting, gamma=self.gamma)

    def knp(self, params, ions, c_all, phi, phi_M, I_ch, splitting, f_source):
        return self.o.assemble_knp(params, ions, c_all, phi, phi_M, I_ch, splitting_scheme=splitting, gamma=self.gamma,
                                   f_source=f_source)
