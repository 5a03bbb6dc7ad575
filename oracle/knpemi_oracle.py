"""CPU oracle for the knpemi hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the shipped package (`knp-emi-fenics-x_amd/knpemi`) never does.

What it is: a numpy/scipy restatement of the arithmetic the reference
(`adajel/knp-emi-fenics-x`) delegates to DOLFINx/FFCx/PETSc/numbalsoda for one
time step -- quadrature-based element integrals of the UFL forms, COO->CSR
assembly, nodal traces, end-of-step updates and an LSODA sweep over the
membrane dofs.  It deliberately works the way FFCx does (tabulate basis
functions at quadrature points, integrate numerically, scatter-add), i.e.
*differently* from the HIP kernels (closed-form P1 element matrices gathered
row by row), so that agreement between the two is meaningful.

PARITY UNPINNED: the reference has no golden vectors, no asserted numbers and
cannot be executed here (dolfinx, ufl, ffcx, basix, scifem, petsc4py,
numbalsoda, numba are absent).  The only
reference-supplied known answers are the analytic MMS solutions
(`tests/run_mms_emi.py:166-176`) and the calibrated ODE initial state
(`examples/idealized_geometries/mm_hh.py:12-16`); `tests/test_oracle.py` checks
the oracle against those, against two manufactured KNP solutions of this repo
(`tests/mms_knp_problem.py`: volume terms; membrane terms in both splitting
modes) and against form-independent invariants.

Third-party algorithm restated here (un-vendored, un-pinned in the reference:
`pyproject.toml:13-18`): DOLFINx 0.10-era `assemble_matrix/vector` over FFCx
kernels with UFL's automatic quadrature degree, and
numbalsoda's LSODA (here: `scipy.integrate.odeint`, the ODEPACK original).

Inputs are plain arrays (parent mesh vertices/cells/cell tags and the tagged
facets as vertex tuples) so that the oracle shares no code with the product's
host layer.
"""
from __future__ import annotations

import itertools
import math

import numpy as np
import scipy.sparse as sp
from scipy.integrate import odeint

# ---------------------------------------------------------------------------
# reference elements and quadrature (what Basix would tabulate)
# ---------------------------------------------------------------------------
_TENSOR = {"interval": 1, "quadrilateral": 2, "hexahedron": 3}
FACET_OF = {"triangle": "interval", "tetrahedron": "triangle", "hexahedron": "quadrilateral",
            "quadrilateral": "interval"}

# 12-point degree-6 fully symmetric triangle rule (Dunavant 1985; polished to
# full precision by Newton iteration on the moment equations).  SURVEY.md
# appendix D: FFCx would pick the Xiao-Gimbutas degree-6 rule (12 points) for
# the rational KNP membrane integrand; weights here sum to 1.
_T6 = (
    (0.1167862757263793660252896, 0.2492867451709104212916386),
    (0.05084490637020681692093681, 0.0630890144915022283403316),
    (0.08285107561837357519355346, 0.05314504984481694735324967, 0.3103524510337844054166077),
)


def _tri_deg6():
    pts, wts = [], []
    for w, b in _T6[:2]:
        a = 1.0 - 2.0 * b
        for p in ((a, b, b), (b, a, b), (b, b, a)):
            pts.append(p)
            wts.append(w)
    w, b, c = _T6[2]
    a = 1.0 - b - c
    for p in sorted(set(itertools.permutations((a, b, c)))):
        pts.append(p)
        wts.append(w)
    bary = np.array(pts)
    return bary[:, 1:3].copy(), 0.5 * np.array(wts)


def _gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def _gauss_jacobi01(n, alpha):
    """Gauss rule on [0,1] for the weight (1-x)^alpha, alpha in {0,1,2}."""
    from scipy.special import roots_jacobi
    x, w = roots_jacobi(n, alpha, 0)
    return 0.5 * (x + 1.0), w / 2.0 ** (alpha + 1)


def quadrature(cell_type, degree):
    """Points (nq, tdim) / weights on the reference cell, exact to `degree`."""
    if cell_type in _TENSOR:
        d = _TENSOR[cell_type]
        m = (degree + 2) // 2  # Basix Gauss-Jacobi point count (appendix D)
        x, w = _gauss01(m)
        grids = np.meshgrid(*([x] * d), indexing="ij")
        pts = np.stack([g.ravel() for g in grids], axis=1)
        wg = np.meshgrid(*([w] * d), indexing="ij")
        wts = np.prod(np.stack([g.ravel() for g in wg], axis=1), axis=1)
        return pts, wts
    if cell_type == "triangle":
        if degree <= 1:
            return np.array([[1 / 3, 1 / 3]]), np.array([0.5])
        if degree == 2:
            return (np.array([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]]),
                    np.full(3, 1 / 6))
        if degree == 6:
            return _tri_deg6()
        m = (degree + 2) // 2  # collapsed Gauss-Jacobi (Duffy) rule
        x0, w0 = _gauss_jacobi01(m, 1)
        x1, w1 = _gauss01(m)
        pts = np.array([[a, b * (1 - a)] for a in x0 for b in x1])
        wts = np.array([wa * wb for wa in w0 for wb in w1])
        return pts, wts
    if cell_type == "tetrahedron":
        if degree <= 1:
            return np.array([[0.25, 0.25, 0.25]]), np.array([1 / 6])
        if degree == 2:
            a, b = 0.5854101966249685, 0.1381966011250105
            pts = np.array([[b, b, b], [a, b, b], [b, a, b], [b, b, a]])
            return pts, np.full(4, 1 / 24)
        m = (degree + 2) // 2
        x0, w0 = _gauss_jacobi01(m, 2)
        x1, w1 = _gauss_jacobi01(m, 1)
        x2, w2 = _gauss01(m)
        pts = np.array([[a, b * (1 - a), c * (1 - a) * (1 - b)]
                        for a in x0 for b in x1 for c in x2])
        wts = np.array([wa * wb * wc for wa in w0 for wb in w1 for wc in w2])
        return pts, wts
    raise ValueError(cell_type)


def tabulate(cell_type, pts):
    """CG-1 basis values (nq, nv) and reference gradients (nq, nv, tdim)."""
    pts = np.atleast_2d(pts)
    nq = pts.shape[0]
    if cell_type in ("triangle", "tetrahedron"):
        d = pts.shape[1]
        phi = np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)
        dphi = np.zeros((nq, d + 1, d))
        dphi[:, 0, :] = -1.0
        for i in range(d):
            dphi[:, i + 1, i] = 1.0
        return phi, dphi
    d = _TENSOR[cell_type]
    nv = 2 ** d
    phi = np.ones((nq, nv))
    dphi = np.ones((nq, nv, d))
    for v in range(nv):
        for ax in range(d):
            bit = (v >> ax) & 1
            f = pts[:, ax] if bit else 1.0 - pts[:, ax]
            df = np.ones(nq) if bit else -np.ones(nq)
            phi[:, v] *= f
            for t in range(d):
                dphi[:, v, t] *= df if t == ax else f
    return phi, dphi


def _geometry(X, dphi):
    """X (nc, nv, gdim), dphi (nq, nv, tdim) -> detJ (nc, nq), physical grads
    G (nc, nq, nv, gdim) (pseudo-inverse for embedded facets)."""
    J = np.einsum("cag,qat->cqgt", X, dphi)
    gdim, tdim = J.shape[2], J.shape[3]
    if gdim == tdim:
        det = np.abs(np.linalg.det(J))
        Jinv = np.linalg.inv(J)
    else:
        JtJ = np.einsum("cqgt,cqgs->cqts", J, J)
        det = np.sqrt(np.abs(np.linalg.det(JtJ)))
        Jinv = np.einsum("cqts,cqgs->cqtg", np.linalg.inv(JtJ), J)
    G = np.einsum("qat,cqtg->cqag", dphi, Jinv)
    return det, G


# ---------------------------------------------------------------------------
# problem description
# ---------------------------------------------------------------------------
class OracleProblem:
    """Sub-meshes, membrane facets and dof numbering built from raw arrays.

    parent mesh: `x` (nv, gdim), `cells` (nc, nvpc), `cell_tags` (nc,);
    tagged facets: `facets` (nF, nvpf) parent vertex ids and `facet_tags` (nF,);
    `subdomains`: ordered {cell_tag: [membrane facet tags]} with ECS (= tag 0,
    empty list) first -- the `subdomain_list` of `run_3D.py:147-171`.

    Sub-mesh vertices are numbered by increasing parent vertex id (the rule the
    product's host layer follows independently).
    """

    def __init__(self, x, cells, cell_type, cell_tags, facets, facet_tags, subdomains):
        self.cell_type = cell_type
        self.facet_type = FACET_OF[cell_type]
        self.gdim = x.shape[1]
        self.tags = list(subdomains.keys())
        assert self.tags[0] == 0, "ECS tag must be zero (run_3D.py:146)"
        self.sub = {}
        for tag in self.tags:
            sel = cells[cell_tags == tag]
            pv = np.unique(sel)
            self.sub[tag] = dict(pv=pv, x=x[pv], cells=np.searchsorted(pv, sel))
        self.N = {tag: self.sub[tag]["pv"].shape[0] for tag in self.tags}
        off, acc = {}, 0
        for tag in self.tags:
            off[tag] = acc
            acc += self.N[tag]
        self.off, self.Ntot = off, acc
        # membrane sub-mesh per cellular sub-domain: all facets whose tag is in
        # membrane_tags (run_3D.py:158); Q dofs = their vertices.
        self.mem = {}
        for tag in self.tags[1:]:
            mtags = list(subdomains[tag])
            sel = np.isin(facet_tags, mtags)
            fv = facets[sel]
            ftag = facet_tags[sel]
            qv = np.unique(fv)
            for side in (0, tag):  # every membrane vertex exists on both sides
                assert np.all(np.isin(qv, self.sub[side]["pv"]))
            self.mem[tag] = dict(
                tags=mtags, fv=fv, ftag=ftag, qv=qv, x=x[qv],
                q=np.searchsorted(qv, fv),
                e=np.searchsorted(self.sub[0]["pv"], fv),
                i=np.searchsorted(self.sub[tag]["pv"], fv),
                q2e=np.searchsorted(self.sub[0]["pv"], qv),
                q2i=np.searchsorted(self.sub[tag]["pv"], qv))
        self.NQ = {tag: self.mem[tag]["qv"].shape[0] for tag in self.tags[1:]}

    # -- nodal trace: utils.interpolate_to_membrane :150-207 ------------------
    def trace(self, tag, ue, ui):
        m = self.mem[tag]
        return ue[m["q2e"]], ui[m["q2i"]]


# ---------------------------------------------------------------------------
# element integrals (restating the UFL forms of emiWeakForm.py / knpWeakForm.py)
# ---------------------------------------------------------------------------
def _scatter(rows, cols, vals, n):
    A = sp.coo_matrix((vals.ravel(), (rows.ravel(), cols.ravel())), shape=(n, n))
    return A.tocsr()


def _facet_tables(P, tag, degree):
    m = P.mem[tag]
    pts, wts = quadrature(P.facet_type, degree)
    phi, dphi = tabulate(P.facet_type, pts)
    X = m["x"][m["q"]]  # (nF, nf, gdim)
    det, _ = _geometry(X, dphi)
    return phi, wts, det


def kappa_nodal(P, params, ions, tag, c_all):
    """kappa_r = F psi sum_k z_k^2 D_k^r c_k^r  (emiWeakForm.py:97-103)."""
    F, psi = params["F"], params["psi"]
    k = 0.0
    for ion, c in zip(ions, c_all[tag]):
        k = k + F * ion["z"] * ion["z"] * ion["D"][tag] * psi * c
    return k


def assemble_emi(P, params, ions, c_all, phi_M_prev, mem_models, splitting_scheme=True):
    """A_emi, P_emi, b_emi (emiWeakForm.py:138-241).

    `c_all[tag]` = list of the K nodal concentration arrays (the eliminated ion
    last: `ion_list[-1]['c_tag']`, emiWeakForm.py:100); `phi_M_prev[tag]` on Q;
    `mem_models[tag]` = list of dicts {'tag': facet tag, 'I_ch_k': {name: Q array}}.
    Unknown order [phi_tag for tag in subdomain_list] (pdeSolver.py:42).
    """
    n = P.Ntot
    C_phi, F = params["C_phi"], params["F"]
    rows, cols, vals, prow, pcol, pval = [], [], [], [], [], []
    b = np.zeros(n)
    ct = P.cell_type
    deg = 1 if ct in ("triangle", "tetrahedron") else 3  # appendix D
    pts, wts = quadrature(ct, deg)
    phi, dphi = tabulate(ct, pts)
    ptsm, wtsm = quadrature(ct, 2)
    phim, dphim = tabulate(ct, ptsm)
    for tag in P.tags:
        s = P.sub[tag]
        cells, X = s["cells"], s["x"][s["cells"]]
        det, G = _geometry(X, dphi)
        dofs = cells + P.off[tag]
        kap = kappa_nodal(P, params, ions, tag, c_all)[cells]  # (nc, nv)
        kq = np.einsum("qa,ca->cq", phi, kap)
        Aloc = np.einsum("q,cq,cq,cqag,cqbg->cab", wts, det, kq, G, G)
        nv = cells.shape[1]
        rows.append(np.repeat(dofs, nv, axis=1))
        cols.append(np.tile(dofs, (1, nv)))
        vals.append(Aloc.reshape(len(cells), nv * nv))
        # RHS: - F z_k D_k grad(c_k) . grad(v) for all K ions (emiWeakForm.py:211-217)
        for ion, c in zip(ions, c_all[tag]):
            gc = np.einsum("ca,cqag->cqg", c[cells], G)
            bl = -F * ion["z"] * ion["D"][tag] * np.einsum("q,cq,cqg,cqag->ca", wts, det, gc, G)
            np.add.at(b, dofs, bl)
        if tag > 0:  # preconditioner mass term (emiWeakForm.py:196)
            detm, _ = _geometry(X, dphim)
            Mloc = np.einsum("q,cq,qa,qb->cab", wtsm, detm, phim, phim)
            prow.append(np.repeat(dofs, nv, axis=1))
            pcol.append(np.tile(dofs, (1, nv)))
            pval.append(Mloc.reshape(len(cells), nv * nv))
    # membrane coupling (emiWeakForm.py:160-165, 228-239)
    for tag in P.tags[1:]:
        m = P.mem[tag]
        fphi, fw, fdet = _facet_tables(P, tag, 2)
        M = np.einsum("q,fq,qa,qb->fab", fw, fdet, fphi, fphi)
        for mm in mem_models[tag]:
            sel = m["ftag"] == mm["tag"]
            Ms = C_phi * M[sel]
            E = m["e"][sel] + P.off[0]
            I = m["i"][sel] + P.off[tag]
            nf = E.shape[1]
            for R, Cc, sgn in ((I, I, 1.0), (I, E, -1.0), (E, I, -1.0), (E, E, 1.0)):
                rows.append(np.repeat(R, nf, axis=1))
                cols.append(np.tile(Cc, (1, nf)))
                vals.append(sgn * Ms.reshape(len(E), nf * nf))
            g = phi_M_prev[tag].copy()
            if not splitting_scheme:
                g = g - sum(mm["I_ch_k"].values()) / C_phi
            bl = np.einsum("fab,fb->fa", Ms, g[m["q"][sel]])
            np.add.at(b, I, bl)
            np.add.at(b, E, -bl)
    cat = lambda L: np.concatenate([a.ravel() for a in L])
    A = _scatter(cat(rows), cat(cols), cat(vals), n)
    if prow:
        Pm = A + _scatter(cat(prow), cat(pcol), cat(pval), n)
    else:
        Pm = A.copy()
    return A, Pm, b


def knp_block_offsets(P, n_solved):
    """Block order [c[tag][k] for tag for k] (pdeSolver.py:117)."""
    off, acc = {}, 0
    for tag in P.tags:
        for k in range(n_solved):
            off[(tag, k)] = acc
            acc += P.N[tag]
    return off, acc


def assemble_knp(P, params, ions, c_all, phi, phi_M_prev, mem_models, dt, splitting_scheme=True,
                 f_source=None):
    """A_knp and b_knp (knpWeakForm.py:123-216).  `c_all` holds c_prev for the
    solved ions and the eliminated ion last; `phi[tag]` is the potential just
    solved for.  `f_source` = optional {k: nodal ECS array} (knpWeakForm.py:164-166)."""
    K = len(ions)
    ns = K - 1
    boff, n = knp_block_offsets(P, ns)
    psi, C_M, F = params["psi"], params["C_M"], params["F"]
    rows, cols, vals = [], [], []
    b = np.zeros(n)
    ct = P.cell_type
    deg = 2 if ct in ("triangle", "tetrahedron") else 3
    pts, wts = quadrature(ct, deg)
    phi_t, dphi_t = tabulate(ct, pts)
    for tag in P.tags:
        s = P.sub[tag]
        cells, X = s["cells"], s["x"][s["cells"]]
        det, G = _geometry(X, dphi_t)
        nv = cells.shape[1]
        gphi = np.einsum("ca,cqag->cqg", phi[tag][cells], G)
        mass = np.einsum("q,cq,qa,qb->cab", wts, det, phi_t, phi_t)
        stiff = np.einsum("q,cq,cqag,cqbg->cab", wts, det, G, G)
        # drift[a (test), b (trial)] = int u_b grad(phi).grad(v_a)
        drift = np.einsum("q,cq,qb,cqg,cqag->cab", wts, det, phi_t, gphi, G)
        for k in range(ns):
            ion = ions[k]
            D, z = ion["D"][tag], ion["z"]
            dofs = cells + boff[(tag, k)]
            Aloc = mass / dt + D * stiff + z * psi * D * drift
            rows.append(np.repeat(dofs, nv, axis=1))
            cols.append(np.tile(dofs, (1, nv)))
            vals.append(Aloc.reshape(len(cells), nv * nv))
            rhs = c_all[tag][k][cells] / dt
            if tag == 0 and f_source is not None and k in f_source:
                rhs = rhs + f_source[k][cells]
            np.add.at(b, dofs, np.einsum("cab,cb->ca", mass, rhs))
    # membrane terms (knpWeakForm.py:168-214), quadrature degree 6 (appendix D)
    for tag in P.tags[1:]:
        m = P.mem[tag]
        fphi, fw, fdet = _facet_tables(P, tag, 6)
        asum_e = sum(ion["D"][0] * ion["z"] ** 2 * c for ion, c in zip(ions, c_all[0]))
        asum_i = sum(ion["D"][tag] * ion["z"] ** 2 * c for ion, c in zip(ions, c_all[tag]))
        at = lambda nodal, idx: np.einsum("qa,fa->fq", fphi, nodal[idx])
        for mm in mem_models[tag]:
            sel = m["ftag"] == mm["tag"]
            E, I, Q = m["e"][sel], m["i"][sel], m["q"][sel]
            wd = fw[None, :] * fdet[sel]
            I_ch = sum(mm["I_ch_k"].values()) + np.zeros(P.NQ[tag])
            for k in range(ns):
                ion = ions[k]
                z, name = ion["z"], ion["name"]
                # alpha = D z^2 c / alpha_sum is a ratio of interpolated P1 fields
                a_e = ion["D"][0] * z * z * at(c_all[0][k], E) / at(asum_e, E)
                a_i = ion["D"][tag] * z * z * at(c_all[tag][k], I) / at(asum_i, I)
                C_e = a_e * C_M / (F * z * dt)
                C_i = a_i * C_M / (F * z * dt)
                pm = at(phi_M_prev[tag], Q)
                Ik = at(mm["I_ch_k"][name] + np.zeros(P.NQ[tag]), Q)
                g_e = pm - dt / (C_M * a_e) * Ik
                g_i = pm - dt / (C_M * a_i) * Ik
                if splitting_scheme:
                    It = at(I_ch, Q)
                    g_e = g_e + (dt / C_M) * It
                    g_i = g_i + (dt / C_M) * It
                jump = at(phi[tag], I) - at(phi[0], E)
                fe = -C_e * g_e + C_e * jump   # tested with v_e(+)
                fi = C_i * g_i - C_i * jump    # tested with v_i(-)
                np.add.at(b, E + boff[(0, k)], np.einsum("fq,fq,qa->fa", wd, fe, fphi))
                np.add.at(b, I + boff[(tag, k)], np.einsum("fq,fq,qa->fa", wd, fi, fphi))
    cat = lambda L: np.concatenate([a.ravel() for a in L])
    return _scatter(cat(rows), cat(cols), cat(vals), n), b


# ---------------------------------------------------------------------------
# end-of-step update (utils.update_pde_variables :238-295)
# ---------------------------------------------------------------------------
def update_pde_variables(P, ions, rho, c, c_all, phi, phi_M_prev):
    """c_prev <- c; eliminated ion from electroneutrality; phi_M <- tr(phi_i) - tr(phi_e).
    `c[tag]` = the K-1 freshly solved fields; `c_all[tag]` is updated in place."""
    zK = ions[-1]["z"]
    for tag in P.tags:
        elim = -(1.0 / zK) * rho["z"] * rho[tag] + np.zeros(P.N[tag])
        for k in range(len(ions) - 1):
            c_all[tag][k][:] = c[tag][k]
            elim += -(1.0 / zK) * ions[k]["z"] * c_all[tag][k]
        c_all[tag][-1][:] = elim
        if tag != 0:
            te, ti = P.trace(tag, phi[0], phi[tag])
            phi_M_prev[tag][:] = ti - te


# ---------------------------------------------------------------------------
# membrane models (restated from the reference's Gotran modules) and LSODA sweep
# ---------------------------------------------------------------------------
def rhs_hh_si(y, t, p):
    """examples/idealized_geometries/mm_hh.py:139-227 (V, s, S/m^2)."""
    m, h, n, V = y
    psi, zK = p[21], p[19]
    E_Na = 1 / psi * 1 / zK * math.log(p[11] / p[12])  # z_K on purpose (:169)
    E_K = 1 / psi * 1 / zK * math.log(p[9] / p[10])
    am = 0.1e3 * (25. - 1.0e3 * (V + 65.0e-3)) / (math.exp((25. - 1.0e3 * (V + 65.0e-3)) / 10.) - 1)
    bm = 4.e3 * math.exp(-1.0e3 * (V + 65.0e-3) / 18.)
    ah = 0.07e3 * math.exp(-1.0e3 * (V + 65.0e-3) / 20.)
    bh = 1.e3 / (math.exp((30. - 1.0e3 * (V + 65.0e-3)) / 10.) + 1)
    an = 0.01e3 * (10. - 1.0e3 * (V + 65.0e-3)) / (math.exp((10. - 1.0e3 * (V + 65.0e-3)) / 10.) - 1.)
    bn = 0.125e3 * math.exp(-1.0e3 * (V + 65.0e-3) / 80.)
    i_stim = p[8] * math.exp(-math.fmod(t, 0.03) / 0.002) * (t < 125e-3)
    i_pump = p[6] / ((1 + p[4] / p[9]) ** 2 * (1 + p[5] / p[12]) ** 3)
    i_Na = (p[2] + p[0] * h * m ** 3 + i_stim) * (V - E_Na) + 3 * i_pump
    i_K = (p[3] + p[1] * n ** 4) * (V - E_K) - 2 * i_pump
    p[15], p[16], p[17] = i_Na, i_K, 0.0
    return [(1 - m) * am - m * bm, (1 - h) * ah - h * bh, (1 - n) * an - n * bn,
            (-i_K - i_Na) / p[7]]


def rhs_hh_mv(y, t, p):
    """examples/local_astrocyte_depolarization/mm_hh.py:130-201 (mV, ms)."""
    m, h, n, V = y
    psi, zK = p[21], p[19]
    E_Na = 1 / psi * 1 / zK * math.log(p[11] / p[12])
    E_K = 1 / psi * 1 / zK * math.log(p[9] / p[10])
    am = 0.1 * (25. - 1.0 * (V + 65.0)) / (math.exp((25. - 1.0 * (V + 65.0)) / 10.) - 1)
    bm = 4. * math.exp(-1.0 * (V + 65.0) / 18.)
    ah = 0.07 * math.exp(-1.0 * (V + 65.0) / 20.)
    bh = 1. / (math.exp((30. - 1.0 * (V + 65.0)) / 10.) + 1)
    an = 0.01 * (10. - 1.0 * (V + 65.0)) / (math.exp((10. - 1.0 * (V + 65.0)) / 10.) - 1.)
    bn = 0.125 * math.exp(-1.0 * (V + 65.0) / 80.)
    i_stim = p[8] * math.exp(-math.fmod(t, 30.0) / 2.0) * (t < 125)
    i_pump = p[6] / ((1 + p[4] / p[9]) ** 2 * (1 + p[5] / p[12]) ** 3)
    i_Na = (p[2] + p[0] * h * m ** 3 + i_stim) * (V - E_Na) + 3 * i_pump
    i_K = (p[3] + p[1] * n ** 4) * (V - E_K) - 2 * i_pump
    p[15], p[16], p[17] = i_Na, i_K, 0.0
    return [(1 - m) * am - m * bm, (1 - h) * ah - h * bh, (1 - n) * an - n * bn,
            (-i_K - i_Na) / p[7]]


def rhs_glial(y, t, p):
    """examples/local_astrocyte_depolarization/mm_glial.py:133-205 (mV, ms)."""
    V = y[0]
    psi, zK, zCl = p[22], p[20], p[21]
    E_Na = 1 / psi * 1 / zK * math.log(p[15] / p[16])
    E_K = 1 / psi * 1 / zK * math.log(p[13] / p[14])
    E_Cl = 1 / psi * 1 / zCl * math.log(p[17] / p[18])
    temperature, R, F = 307e3, 8.315e3, 96500e3
    i_pump = p[10] * (p[13] / (p[13] + p[8])) * (p[16] ** 1.5 / (p[16] ** 1.5 + p[9] ** 1.5))
    E_K_init = R * temperature / F * math.log(p[11] / p[12])
    dphi = V - E_K
    A = 1 + math.exp(18.5 / 42.4)
    B = 1 + math.exp(-(118.6 + E_K_init) / 44.1)
    C = 1 + math.exp((dphi + 18.5) / 42.4)
    D = 1 + math.exp(-(118.6 + V) / 44.1)
    g_Kir = math.sqrt(p[13] / p[11]) * (A * B) / (C * D)
    i_Kir = p[2] * g_Kir * (V - E_K)
    i_Na = p[1] * (V - E_Na) + 3 * i_pump
    i_K = i_Kir - 2 * i_pump
    i_Cl = p[0] * (V - E_Cl)
    p[5], p[6], p[7] = i_Na, i_K, i_Cl
    return [(-i_K - i_Na - i_Cl) / p[3]]


MODELS = {
    "hh_si": dict(
        rhs=rhs_hh_si, n_states=4, n_params=22, V=3,
        states=[0.016648440745822956, 0.8542015627820805, 0.1882020248041632, -0.07438609374462003],
        params=[1200, 360, 1.0, 4.0, 2, 7.7, 0.449] + [0.0] * 15,
        pidx={"Cm": 7, "stim_amplitude": 8, "K_e": 9, "K_i": 10, "Na_e": 11, "Na_i": 12,
              "Cl_e": 13, "Cl_i": 14, "I_ch_Na": 15, "I_ch_K": 16, "I_ch_Cl": 17,
              "z_Na": 18, "z_K": 19, "z_Cl": 20, "psi": 21}),
    "hh_mv": dict(
        rhs=rhs_hh_mv, n_states=4, n_params=22, V=3,
        states=[0.015211986965658385, 0.8667432624969533, 0.17994146133363148, -75.09159534786934],
        params=[120, 36, 0.1, 0.4, 1.5, 10, 58.0] + [0.0] * 15,
        pidx={"Cm": 7, "stim_amplitude": 8, "K_e": 9, "K_i": 10, "Na_e": 11, "Na_i": 12,
              "Cl_e": 13, "Cl_i": 14, "I_ch_Na": 15, "I_ch_K": 16, "I_ch_Cl": 17,
              "z_Na": 18, "z_K": 19, "z_Cl": 20, "psi": 21}),
    "glial": dict(
        rhs=rhs_glial, n_states=1, n_params=23, V=0,
        states=[-85.84503411546689],
        params=[0.05, 0.1, 1.696, 0, 0, 0, 0, 0, 1.5, 10, 10.75975,
                3.092970607490389, 99.3100014897692] + [0.0] * 10,
        pidx={"Cm": 3, "stim_amplitude": 4, "I_ch_Na": 5, "I_ch_K": 6, "I_ch_Cl": 7,
              "K_e": 13, "K_i": 14, "Na_e": 15, "Na_i": 16, "Cl_e": 17, "Cl_i": 18,
              "z_Na": 19, "z_K": 20, "z_Cl": 21, "psi": 22}),
}


def ode_sweep(model, states, params, t0, dt, stim_mask=None, stimulus=None,
              rtol=1e-8, atol=1e-10, rows=None):
    """MembraneModel.step_lsoda (odeSolver.py:92-127): per row, optional
    stimulus write, LSODA over [t0, t0+dt], new state = last row of the output;
    the RHS stores I_ch_* into the parameter row as a side effect.

    Returns `I_end` (n, 3): currents re-evaluated at the returned state at
    t0+dt -- the product's documented semantic -- next
    to the side-effect values left in `params` by LSODA's last RHS call.
    """
    rhs = MODELS[model]["rhs"] if isinstance(model, str) else model
    n = states.shape[0]
    rows = range(n) if rows is None else rows
    I_end = np.zeros((n, 3))
    for r in rows:
        p = params[r]
        if stimulus and (stim_mask is None or stim_mask[r]):
            for idx, val in stimulus.items():
                p[idx] = val
        sol = odeint(rhs, states[r].copy(), [t0, t0 + dt], args=(p,), rtol=rtol, atol=atol)
        states[r, :] = sol[-1]
        side = p.copy()
        rhs(states[r], t0 + dt, p)
        I_end[r] = p[_ich_slice(len(p))]
        p[:] = side
    return I_end


def _ich_slice(n_params):
    # I_ch_Na, I_ch_K, I_ch_Cl sit at 15..17 in the 22-parameter HH modules and
    # at 5..7 in the 23-parameter glial module.
    return slice(15, 18) if n_params == 22 else slice(5, 8)
