"""CPU restatement of the DG(P1) + symmetric-interior-penalty variant of the KNP-EMI step.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product path (knpemi.dg drives the HIP kernels and fails loudly without them).

PARITY UNPINNED, and differently from the CG oracle: `/root/reference` holds no DG code at all.  Its README
(`README.md:5-7`) points at the legacy KNP-EMI-DG solver of Ellingsrud, Benedusi & Kuchta (SISC 47(2), 2025), which is
not in the container, and `examples/idealized_geometries/make_mesh_2D.py:88-90` keeps that solver's "interior facets
tagged 0" marker convention.  What is restated here is therefore the textbook construction applied to the reference's
own equations (Appendix A of SURVEY.md; emiWeakForm.py:138-241, knpWeakForm.py:123-216): broken P1 on ONE mesh,

  * volume terms of the CG forms cell by cell,
  * on every interior facet inside a sub-domain the symmetric interior penalty treatment of the two diffusion
    operators (Arnold 1982; Riviere 2008, ch. 2):   - {k grad u}.n [v] - {k grad v}.n [u] + (gamma / h_F) {k} [u] [v],
    k = kappa (EMI) or D_k (KNP), 1 / h_F = mean of the two inverse cell heights over the facet,
  * first-order upwinding of the drift flux z_k psi D_k c grad(phi) (the advective part of the Nernst-Planck flux):
    beta_F = -z_k psi {D_k grad phi}.n, flux beta_F c^upwind [v],
  * the consistency term of the known diffusive current on the right-hand side of the potential equation,
    + F z_k {D_k grad c_k}.n [v],
  * on membrane facets exactly the reference's Robin / flux terms with the traces taken from the two adjacent cells
    (emiWeakForm.py:160-165,228-239; knpWeakForm.py:168-214), phi_M and I_ch living at the facet's own vertices.

It is verified by manufactured solutions (second order in L2, tests/test_dg_oracle.py), by its algebraic properties
(symmetry of A_emi, constants in its kernel, mass conservation of A_knp) and against the CG oracle on the same problem;
the HIP kernels are then held to it at 1e-10.

This file integrates every facet term by quadrature with the basis functions evaluated through each cell's affine
map -- on purpose a different route from the kernels, which use closed-form facet moments.
"""
import itertools

import numpy as np
import scipy.sparse as sp

from knpemi_oracle import quadrature, tabulate

_FACET_OF = {"triangle": "interval", "tetrahedron": "triangle"}


def _affine(X):
    """lambda_j(x) = a0[c, j] + G[c, j] . x for every cell; X (nc, nv, d).  Returns a0, G, |T|."""
    nc, nv, d = X.shape
    M = np.concatenate([np.ones((nc, nv, 1)), X], axis=2)          # rows: [1, x_j]
    Minv = np.linalg.inv(M)                                        # columns j: coefficients of lambda_j
    a0 = Minv[:, 0, :]
    G = np.transpose(Minv[:, 1:, :], (0, 2, 1))
    vol = np.abs(np.linalg.det(M)) / {2: 2.0, 3: 6.0}[d]
    return a0, G, vol


class DGOracle:
    """Topology of the broken space: dof (cell c, local vertex j) = c * nv + j.

    x (nvert, d), cells (nc, nv), cell_sub (nc,) sub-domain index (0 = ECS), mem_facets (nmf, nf) global vertex ids of
    the membrane facets and mem_tags (nmf,) their facet tags.  Membrane node (f, a) = f * nf + a is vertex
    mem_facets[f, a]."""

    def __init__(self, x, cells, cell_type, cell_sub, mem_facets, mem_tags):
        self.x = np.asarray(x, float)
        self.cells = np.asarray(cells, np.int64)
        self.cell_type = cell_type
        self.facet_type = _FACET_OF[cell_type]
        self.cell_sub = np.asarray(cell_sub, np.int64)
        self.nc, self.nv = self.cells.shape
        self.d = self.x.shape[1]
        self.n = self.nc * self.nv
        self.a0, self.G, self.vol = _affine(self.x[self.cells])
        # all facets as sorted vertex tuples -> the cells that own them
        owners = {}
        for c in range(self.nc):
            for f in range(self.nv):
                key = tuple(sorted(np.delete(self.cells[c], f)))
                owners.setdefault(key, []).append((c, f))
        mem = {tuple(sorted(v)): (i, t) for i, (v, t) in enumerate(zip(np.asarray(mem_facets), mem_tags))}
        self.interior, self.membrane = [], [None] * len(mem)
        for key, own in owners.items():
            if len(own) == 1:
                continue
            (c0, f0), (c1, f1) = own
            if key in mem:
                s0, s1 = self.cell_sub[c0], self.cell_sub[c1]
                assert min(s0, s1) == 0 and max(s0, s1) > 0, "a membrane separates the ECS from one cell"
                e, i = ((c0, f0), (c1, f1)) if s0 == 0 else ((c1, f1), (c0, f0))
                idx, tag = mem[key]
                self.membrane[idx] = (e[0], i[0], tag)
            else:
                assert self.cell_sub[c0] == self.cell_sub[c1], "untagged facet between two sub-domains"
                self.interior.append((c0, c1, f0, f1))
        self.mem_facets = np.asarray(mem_facets, np.int64).reshape(len(mem), self.nv - 1)
        self.mem_tags = np.asarray(mem_tags)
        self.nmf, self.nf = self.mem_facets.shape

    # -- helpers -----------------------------------------------------------------------------------------------
    def basis_at(self, c, X):
        """lambda_j of cells c (m,) at points X (m, q, d) -> (m, q, nv)."""
        return self.a0[c][:, None, :] + np.einsum("mjd,mqd->mqj", self.G[c], X)

    def _facet_frame(self, verts, degree):
        """Quadrature points (m, q, d), weights * |F| (m, q) and facet basis (q, nf) on facets verts (m, nf)."""
        pts, wts = quadrature(self.facet_type, degree)
        fphi, _ = tabulate(self.facet_type, pts)
        XF = self.x[verts]                                          # (m, nf, d)
        Xq = np.einsum("qa,mad->mqd", fphi, XF)
        if self.d == 2:
            area = np.linalg.norm(XF[:, 1] - XF[:, 0], axis=1)
            ref = 1.0
        else:
            area = 0.5 * np.linalg.norm(np.cross(XF[:, 1] - XF[:, 0], XF[:, 2] - XF[:, 0]), axis=1)
            ref = 0.5
        return Xq, wts[None, :] / ref * area[:, None], fphi

    def _interior_arrays(self):
        I = np.array(self.interior, np.int64).reshape(-1, 4)
        cT, cN, fT, fN = I.T
        verts = np.array([np.delete(self.cells[c], f) for c, f in zip(cT, fT)], np.int64).reshape(-1, self.nv - 1)
        # outward normal of T: away from T's vertex opposite the facet
        gT = self.G[cT, fT]                                         # grad lambda_f points into T
        n = -gT / np.linalg.norm(gT, axis=1)[:, None]
        inv_h = 0.5 * (np.linalg.norm(gT, axis=1) + np.linalg.norm(self.G[cN, fN], axis=1))
        return cT, cN, verts, n, inv_h

    def mem_local(self):
        """Local vertex index, in the ECS cell and in the intracellular cell, of every membrane node: (nmf, nf) each."""
        le = np.zeros((self.nmf, self.nf), np.int64)
        li = np.zeros_like(le)
        for f, (ce, ci, _) in enumerate(self.membrane):
            for a, v in enumerate(self.mem_facets[f]):
                le[f, a] = np.nonzero(self.cells[ce] == v)[0][0]
                li[f, a] = np.nonzero(self.cells[ci] == v)[0][0]
        return le, li

    def mem_cells(self):
        ce = np.array([m[0] for m in self.membrane], np.int64)
        ci = np.array([m[1] for m in self.membrane], np.int64)
        return ce, ci

    def kappa(self, params, ions, c_all):
        """kappa = F psi sum_k z_k^2 D_k c_k per dof (emiWeakForm.py:97-103); c_all[k] (nc, nv)."""
        k = 0.0
        for ion, c in zip(ions, c_all):
            k = k + params["F"] * params["psi"] * ion["z"] ** 2 * np.asarray(ion["D"])[self.cell_sub][:, None] * c
        return k

    def _dofs(self, c):
        return c[:, None] * self.nv + np.arange(self.nv)[None, :]

    def _add_blocks(self, acc, cr, cc, blk):
        """blk (m, nv, nv): rows of cells cr, columns of cells cc."""
        R = np.repeat(self._dofs(cr), self.nv, axis=1)
        C = np.tile(self._dofs(cc), (1, self.nv))
        acc.append((R.ravel(), C.ravel(), blk.reshape(-1)))

    def _csr(self, acc):
        r = np.concatenate([a[0] for a in acc])
        c = np.concatenate([a[1] for a in acc])
        v = np.concatenate([a[2] for a in acc])
        return sp.coo_matrix((v, (r, c)), shape=(self.n, self.n)).tocsr()

    # -- potential equation -------------------------------------------------------------------------------------
    def assemble_emi(self, params, ions, c_all, phi_M, I_ch, mem_tags_used=None, splitting_scheme=True, gamma=10.0):
        """A_emi (n x n CSR) and b_emi.  c_all: the K concentration fields (nc, nv), eliminated ion last; phi_M
        (nmf, nf); I_ch: list of K arrays (nmf, nf) (only read without the splitting scheme)."""
        F, C_phi = params["F"], params["C_M"] / params["dt"]
        kap = self.kappa(params, ions, c_all)
        acc = []
        b = np.zeros((self.nc, self.nv))
        G, vol = self.G, self.vol
        # volume: kappa grad u . grad v (degree-1 rule = mean of kappa) and -F z D grad c . grad v
        self._add_blocks(acc, np.arange(self.nc), np.arange(self.nc),
                         (vol * kap.mean(axis=1))[:, None, None] * np.einsum("cid,cjd->cij", G, G))
        for ion, c in zip(ions, c_all):
            Dc = np.asarray(ion["D"])[self.cell_sub]
            gc = np.einsum("cj,cjd->cd", c, G)
            b -= F * ion["z"] * (Dc * vol)[:, None] * np.einsum("cid,cd->ci", G, gc)
        if self.interior:
            cT, cN, verts, n, inv_h = self._interior_arrays()
            Xq, w, _ = self._facet_frame(verts, 3)
            side = ((cT, 1.0), (cN, -1.0))
            lam = {0: self.basis_at(cT, Xq), 1: self.basis_at(cN, Xq)}
            kq = {s: np.einsum("mqj,mj->mq", lam[s], kap[c]) for s, (c, _) in enumerate(side)}
            gn = {s: np.einsum("mjd,md->mj", G[c], n) for s, (c, _) in enumerate(side)}
            kavg = 0.5 * (kq[0] + kq[1])
            for s, (cs, sg_s) in enumerate(side):
                for t, (ct_, sg_t) in enumerate(side):
                    blk = (-0.5 * sg_s * np.einsum("mq,mqi,mq,mj->mij", w, lam[s], kq[t], gn[t])
                           - 0.5 * sg_t * np.einsum("mq,mqj,mq,mi->mij", w, lam[t], kq[s], gn[s])
                           + gamma * sg_s * sg_t * np.einsum("m,mq,mq,mqi,mqj->mij", inv_h, w, kavg, lam[s], lam[t]))
                    self._add_blocks(acc, cs, ct_, blk)
            for ion, c in zip(ions, c_all):
                Dsub = np.asarray(ion["D"])
                flux = 0.5 * sum(Dsub[self.cell_sub[cc]][:, None] * np.einsum("mj,mjd->md", c[cc], G[cc])
                                 for cc in (cT, cN))
                fn = F * ion["z"] * np.einsum("md,md->m", flux, n)
                for s, (cs, sg_s) in enumerate(side):
                    np.add.at(b, cs, sg_s * fn[:, None] * np.einsum("mq,mqi->mi", w, lam[s]))
        if self.nmf:
            ce, ci = self.mem_cells()
            Xq, w, fphi = self._facet_frame(self.mem_facets, 2)
            le, li = self.mem_local()
            lam_e, lam_i = self.basis_at(ce, Xq), self.basis_at(ci, Xq)
            for (cr, lr, sr), (cc, lc, sc) in itertools.product(((ci, lam_i, 1.0), (ce, lam_e, -1.0)), repeat=2):
                self._add_blocks(acc, cr, cc, C_phi * sr * sc * np.einsum("mq,mqi,mqj->mij", w, lr, lc))
            g = np.array(phi_M, float)
            if not splitting_scheme:
                g = g - sum(I_ch) / C_phi
            gq = np.einsum("qa,ma->mq", fphi, g)
            np.add.at(b, ci, C_phi * np.einsum("mq,mq,mqi->mi", w, gq, lam_i))
            np.add.at(b, ce, -C_phi * np.einsum("mq,mq,mqi->mi", w, gq, lam_e))
        return self._csr(acc), b.ravel()

    # -- concentration equations --------------------------------------------------------------------------------
    def assemble_knp(self, params, ions, c_all, phi, phi_M, I_ch, splitting_scheme=True, gamma=10.0, f_source=None):
        """[A_k for the K-1 solved ions] and b (K-1, n).  c_all: previous-step fields (eliminated ion last); phi (nc, nv)
        the potential just solved for; I_ch: list of K arrays (nmf, nf); f_source: optional {k: (nc, nv)} (ECS cells)."""
        K = len(ions)
        ns = K - 1
        psi, C_M, F, dt = params["psi"], params["C_M"], params["F"], params["dt"]
        G, vol, nv, d = self.G, self.vol, self.nv, self.d
        mass = vol[:, None, None] * (1.0 + np.eye(nv))[None] / ((d + 1) * (d + 2))
        stiff = vol[:, None, None] * np.einsum("cid,cjd->cij", G, G)
        gphi = np.einsum("cj,cjd->cd", phi, G)
        drift = (vol / (d + 1))[:, None, None] * np.einsum("cid,cd->ci", G, gphi)[:, :, None] * np.ones((1, 1, nv))
        accs = [[] for _ in range(ns)]
        b = np.zeros((ns, self.nc, nv))
        allc = np.arange(self.nc)
        for k in range(ns):
            Dc = np.asarray(ions[k]["D"])[self.cell_sub][:, None, None]
            self._add_blocks(accs[k], allc, allc, mass / dt + Dc * stiff + ions[k]["z"] * psi * Dc * drift)
            rhs = c_all[k] / dt
            if f_source is not None and k in f_source:
                rhs = rhs + np.where(self.cell_sub[:, None] == 0, f_source[k], 0.0)
            b[k] += np.einsum("cij,cj->ci", mass, rhs)
        if self.interior:
            cT, cN, verts, n, inv_h = self._interior_arrays()
            Xq, w, _ = self._facet_frame(verts, 2)
            side = ((cT, 1.0), (cN, -1.0))
            lam = {0: self.basis_at(cT, Xq), 1: self.basis_at(cN, Xq)}
            gn = {s: np.einsum("mjd,md->mj", G[c], n) for s, (c, _) in enumerate(side)}
            for k in range(ns):
                Dsub, z = np.asarray(ions[k]["D"]), ions[k]["z"]
                Ds = {0: Dsub[self.cell_sub[cT]], 1: Dsub[self.cell_sub[cN]]}
                Davg = 0.5 * (Ds[0] + Ds[1])
                beta = -z * psi * 0.5 * np.einsum("md,md->m", Ds[0][:, None] * gphi[cT] + Ds[1][:, None] * gphi[cN], n)
                up = np.where(beta > 0, 0, 1)                       # upwind side: T if the drift leaves T
                for s, (cs, sg_s) in enumerate(side):
                    for t, (ct_, sg_t) in enumerate(side):
                        blk = (-0.5 * sg_s * np.einsum("mq,mqi,m,mj->mij", w, lam[s], Ds[t], gn[t])
                               - 0.5 * sg_t * np.einsum("mq,mqj,m,mi->mij", w, lam[t], Ds[s], gn[s])
                               + gamma * sg_s * sg_t * np.einsum("m,m,mq,mqi,mqj->mij", inv_h, Davg, w, lam[s], lam[t])
                               + sg_s * np.einsum("m,mq,mqi,mqj->mij", np.where(up == t, beta, 0.0), w, lam[s], lam[t]))
                        self._add_blocks(accs[k], cs, ct_, blk)
        if self.nmf:
            ce, ci = self.mem_cells()
            Xq, w, fphi = self._facet_frame(self.mem_facets, 6)
            lam_e, lam_i = self.basis_at(ce, Xq), self.basis_at(ci, Xq)
            at = lambda field, c, lam_: np.einsum("mqj,mj->mq", lam_, field[c])
            asum = sum(np.asarray(ion["D"])[self.cell_sub][:, None] * ion["z"] ** 2 * c for ion, c in zip(ions, c_all))
            pm = np.einsum("qa,ma->mq", fphi, np.asarray(phi_M, float))
            It = np.einsum("qa,ma->mq", fphi, sum(I_ch))
            jump = at(phi, ci, lam_i) - at(phi, ce, lam_e)
            for k in range(ns):
                Dsub, z = np.asarray(ions[k]["D"]), ions[k]["z"]
                a_e = Dsub[self.cell_sub[ce]][:, None] * z * z * at(c_all[k], ce, lam_e) / at(asum, ce, lam_e)
                a_i = Dsub[self.cell_sub[ci]][:, None] * z * z * at(c_all[k], ci, lam_i) / at(asum, ci, lam_i)
                C_e, C_i = a_e * C_M / (F * z * dt), a_i * C_M / (F * z * dt)
                Ik = np.einsum("qa,ma->mq", fphi, I_ch[k])
                g_e = pm - dt / (C_M * a_e) * Ik
                g_i = pm - dt / (C_M * a_i) * Ik
                if splitting_scheme:
                    g_e, g_i = g_e + (dt / C_M) * It, g_i + (dt / C_M) * It
                np.add.at(b[k], ce, np.einsum("mq,mq,mqi->mi", w, -C_e * g_e + C_e * jump, lam_e))
                np.add.at(b[k], ci, np.einsum("mq,mq,mqi->mi", w, C_i * g_i - C_i * jump, lam_i))
        return [self._csr(a) for a in accs], b.reshape(ns, self.n)

    # -- end of step (utils.py:238-295) -------------------------------------------------------------------------
    def update(self, ions, rho, c_new, phi):
        """Returns (c_all for the next step, phi_M): eliminated ion from electroneutrality per dof, phi_M = phi_i - phi_e
        at the membrane nodes.  rho: per-sub-domain immobile charge density (z_rho * rho) or None."""
        zK = ions[-1]["z"]
        s = sum(ion["z"] * c for ion, c in zip(ions[:-1], c_new))
        if rho is not None:
            s = s + np.asarray(rho)[self.cell_sub][:, None]
        c_all = [np.array(c) for c in c_new] + [-s / zK]
        phi_M = np.zeros((self.nmf, self.nf))
        if self.nmf:
            ce, ci = self.mem_cells()
            le, li = self.mem_local()
            phi_M = phi[ci[:, None], li] - phi[ce[:, None], le]
        return c_all, phi_M

    def traces(self, field):
        """(ECS-side, cell-side) values of a broken field at the membrane nodes, (nmf, nf) each."""
        ce, ci = self.mem_cells()
        le, li = self.mem_local()
        return field[ce[:, None], le], field[ci[:, None], li]


# ----------------------------------------------------------------------------------------------------------------------
# Broken Q1 on hexahedra (the reference's own 3-D idealized mesh is hexahedral, make_mesh_3D.py:100-102)
# ----------------------------------------------------------------------------------------------------------------------
def _hex_ref(j):
    """Reference coordinates of local vertex j of a hexahedron (tensor ordering: bit a of j is the coordinate along axis a)."""
    return np.array([(j >> a) & 1 for a in range(3)], float)


_HEX_FACETS = [[j for j in range(8) if ((j >> a) & 1) == b] for a in range(3) for b in range(2)]   # facet f = 2 a + b


class DGOracleQ1(DGOracle):
    """The same discrete problem on broken Q1 over hexahedra.  Local vertex j of a cell sits at the reference point whose
    coordinate along axis a is bit a of j; dof (cell c, local vertex j) = 8 c + j; local facet f = 2 a + b is xi_a = b.

    What differs from the simplicial statement above is only what a non-constant Jacobian forces:
      * volume terms: the 2 x 2 x 2 Gauss rule, Jacobian inverted at every point (as the CG forms on hexahedra,
        oracle/knpemi_oracle.py) -- exact on parallelepipeds;
      * interior facets: the 2 x 2 Gauss rule on the facet's bilinear parametrisation, every quantity (normal, the two
        cells' gradients, kappa, the drift velocity and with it the upwind side, and 1 / h_F = mean of the two cells'
        |grad xi_normal|) taken at the point;
      * membrane facets: the rules of the CG forms on quadrilaterals (degree 2 -> 2 x 2 for the potential coupling,
        degree 6 -> 4 x 4 for the rational concentration terms).
    Every point is mapped into each adjacent cell's own reference coordinates and the basis is evaluated through that
    cell's own Jacobian -- the kernels instead work in a frame aligned with the facet."""

    def __init__(self, x, cells, cell_type, cell_sub, mem_facets, mem_tags):
        assert cell_type == "hexahedron"
        self.x = np.asarray(x, float)
        self.cells = np.asarray(cells, np.int64)
        self.cell_type, self.facet_type = cell_type, "quadrilateral"
        self.cell_sub = np.asarray(cell_sub, np.int64)
        self.nc, self.nv = self.cells.shape
        self.d = 3
        self.n = self.nc * self.nv
        pts, wts = quadrature("hexahedron", 2)
        det, _ = self._geom(np.arange(self.nc), np.broadcast_to(pts, (self.nc,) + pts.shape))
        self.vol = det @ wts
        owners = {}
        for c in range(self.nc):
            for f, lv in enumerate(_HEX_FACETS):
                owners.setdefault(tuple(sorted(self.cells[c, lv])), []).append((c, f))
        self.mem_facets = np.asarray(mem_facets, np.int64).reshape(-1, 4)
        self.mem_tags = np.asarray(mem_tags)
        self.nmf, self.nf = self.mem_facets.shape
        mem = {tuple(sorted(v)): i for i, v in enumerate(self.mem_facets)}
        self.interior, self.membrane, self._memf = [], [None] * len(mem), [None] * len(mem)
        for key, own in owners.items():
            if len(own) == 1:
                continue
            (c0, f0), (c1, f1) = own
            if key in mem:
                s0, s1 = self.cell_sub[c0], self.cell_sub[c1]
                assert min(s0, s1) == 0 and max(s0, s1) > 0, "a membrane separates the ECS from one cell"
                e, i = ((c0, f0), (c1, f1)) if s0 == 0 else ((c1, f1), (c0, f0))
                self.membrane[mem[key]] = (e[0], i[0], self.mem_tags[mem[key]])
                self._memf[mem[key]] = (e[1], i[1])
            else:
                assert self.cell_sub[c0] == self.cell_sub[c1], "untagged facet between two sub-domains"
                self.interior.append((c0, c1, f0, f1))

    # -- geometry ----------------------------------------------------------------------------------------------
    def _geom(self, c, xi):
        """Cells c (m,), reference points xi (m, q, 3) -> |det J| (m, q), physical gradients (m, q, 8, 3), basis (m, q, 8)."""
        m, q = xi.shape[:2]
        phi = np.ones((m, q, 8))
        dphi = np.ones((m, q, 8, 3))
        for v in range(8):
            for a in range(3):
                bit = (v >> a) & 1
                f = xi[:, :, a] if bit else 1.0 - xi[:, :, a]
                phi[:, :, v] *= f
                for t in range(3):
                    dphi[:, :, v, t] *= (1.0 if bit else -1.0) if t == a else f
        J = np.einsum("mag,mqat->mqgt", self.x[self.cells[c]], dphi)
        G = np.einsum("mqat,mqtg->mqag", dphi, np.linalg.inv(J))
        self._last_phi, self._last_J = phi, J
        return np.abs(np.linalg.det(J)), G

    def _on_facet(self, c, f, verts, st):
        """Reference coordinates in cells c (m,) of the points st (q, 2) of the facets whose vertices (global ids, in the
        order that defines the parametrisation) are verts (m, 4): bilinear interpolation of the vertices' own reference
        coordinates (the two parametrisations of a shared facet differ by a symmetry of the square)."""
        N = np.stack([(1 - st[:, 0]) * (1 - st[:, 1]), st[:, 0] * (1 - st[:, 1]), (1 - st[:, 0]) * st[:, 1],
                      st[:, 0] * st[:, 1]], axis=1)                       # (q, 4)
        ref = np.zeros((len(c), 4, 3))
        for k in range(len(c)):
            for a in range(4):
                j = int(np.nonzero(self.cells[c[k]] == verts[k, a])[0][0])
                ref[k, a] = _hex_ref(j)
        return np.einsum("qa,mad->mqd", N, ref), N

    def _facet_points(self, cT, fT, degree):
        """Gauss points of the facets fT of cells cT in T's parametrisation: vertices (m, 4), points st, weights,
        reference points in T, surface measure * weight (m, q) and the outward unit normal of T (m, q, 3)."""
        st, w = quadrature("quadrilateral", degree)
        verts = np.array([self.cells[c, _HEX_FACETS[f]] for c, f in zip(cT, fT)], np.int64).reshape(-1, 4)
        xiT, N = self._on_facet(cT, fT, verts, st)
        _, GT = self._geom(cT, xiT)
        JT = self._last_J
        ax = np.array([f // 2 for f in fT])
        sd = np.array([1.0 if f % 2 else -1.0 for f in fT])
        m = len(cT)
        t1 = np.stack([JT[k, :, :, [a for a in range(3) if a != ax[k]][0]] for k in range(m)])
        t2 = np.stack([JT[k, :, :, [a for a in range(3) if a != ax[k]][1]] for k in range(m)])
        cr = np.cross(t1, t2)
        area = np.linalg.norm(cr, axis=2)
        # outward normal: sign * grad xi_axis / |grad xi_axis|; grad xi_a = sum_j ref_a(j) grad phi_j
        refa = np.array([[(j >> a) & 1 for j in range(8)] for a in range(3)], float)
        gxi = np.einsum("mj,mqjd->mqd", refa[ax], GT)
        n = sd[:, None, None] * gxi / np.linalg.norm(gxi, axis=2)[:, :, None]
        return verts, st, w[None, :] * area, xiT, n, np.linalg.norm(gxi, axis=2)

    def _interior_q1(self, degree=3):
        I = np.array(self.interior, np.int64).reshape(-1, 4)
        cT, cN, fT, fN = I.T
        verts, st, wA, xiT, n, ghT = self._facet_points(cT, fT, degree)
        _, GT = self._geom(cT, xiT)
        lamT = self._last_phi
        xiN, _ = self._on_facet(cN, fN, verts, st)
        _, GN = self._geom(cN, xiN)
        lamN = self._last_phi
        refa = np.array([[(j >> a) & 1 for j in range(8)] for a in range(3)], float)
        gxiN = np.einsum("mj,mqjd->mqd", refa[fN // 2], GN)
        inv_h = 0.5 * (ghT + np.linalg.norm(gxiN, axis=2))
        return cT, cN, wA, n, inv_h, (lamT, lamN), (GT, GN)

    def mem_local(self):
        le = np.zeros((self.nmf, 4), np.int64)
        li = np.zeros_like(le)
        for f, (ce, ci, _) in enumerate(self.membrane):
            for a, v in enumerate(self.mem_facets[f]):
                le[f, a] = np.nonzero(self.cells[ce] == v)[0][0]
                li[f, a] = np.nonzero(self.cells[ci] == v)[0][0]
        return le, li

    def _membrane_q1(self, degree):
        """Quadrature on the membrane facets in their own vertex order: weights * measure (m, q), facet basis (q, 4),
        basis of the ECS cell and of the intracellular cell at the points (m, q, 8)."""
        ce, ci = self.mem_cells()
        st, w = quadrature("quadrilateral", degree)
        fe = np.array([f[0] for f in self._memf])
        fi = np.array([f[1] for f in self._memf])
        xiE, N = self._on_facet(ce, fe, self.mem_facets, st)
        xiI, _ = self._on_facet(ci, fi, self.mem_facets, st)
        XF = self.x[self.mem_facets]
        dN = np.stack([np.stack([-(1 - st[:, 1]), (1 - st[:, 1]), -st[:, 1], st[:, 1]], 1),
                       np.stack([-(1 - st[:, 0]), -st[:, 0], (1 - st[:, 0]), st[:, 0]], 1)], 2)   # (q, 4, 2)
        T = np.einsum("mad,qat->mqdt", XF, dN)
        area = np.linalg.norm(np.cross(T[..., 0], T[..., 1]), axis=2)
        self._geom(ce, xiE)
        lam_e = self._last_phi
        self._geom(ci, xiI)
        lam_i = self._last_phi
        return ce, ci, w[None, :] * area, N, lam_e, lam_i

    # -- potential equation -------------------------------------------------------------------------------------
    def assemble_emi(self, params, ions, c_all, phi_M, I_ch, mem_tags_used=None, splitting_scheme=True, gamma=10.0):
        F, C_phi = params["F"], params["C_M"] / params["dt"]
        kap = self.kappa(params, ions, c_all)
        acc = []
        b = np.zeros((self.nc, 8))
        allc = np.arange(self.nc)
        pts, wts = quadrature("hexahedron", 2)
        det, G = self._geom(allc, np.broadcast_to(pts, (self.nc,) + pts.shape))
        lam = self._last_phi
        wq = det * wts[None, :]
        kq = np.einsum("cqj,cj->cq", lam, kap)
        self._add_blocks(acc, allc, allc, np.einsum("cq,cq,cqid,cqjd->cij", wq, kq, G, G))
        for ion, c in zip(ions, c_all):
            Dc = np.asarray(ion["D"])[self.cell_sub]
            gc = np.einsum("cj,cqjd->cqd", c, G)
            b -= F * ion["z"] * Dc[:, None] * np.einsum("cq,cqid,cqd->ci", wq, G, gc)
        if self.interior:
            cT, cN, wA, n, inv_h, lams, Gs = self._interior_q1()
            side = ((cT, 1.0), (cN, -1.0))
            kqf = [np.einsum("mqj,mj->mq", lams[s], kap[c]) for s, (c, _) in enumerate(side)]
            gn = [np.einsum("mqjd,mqd->mqj", Gs[s], n) for s in range(2)]
            kavg = 0.5 * (kqf[0] + kqf[1])
            for s, (cs, sg_s) in enumerate(side):
                for t, (ct_, sg_t) in enumerate(side):
                    blk = (-0.5 * sg_s * np.einsum("mq,mqi,mq,mqj->mij", wA, lams[s], kqf[t], gn[t])
                           - 0.5 * sg_t * np.einsum("mq,mqj,mq,mqi->mij", wA, lams[t], kqf[s], gn[s])
                           + gamma * sg_s * sg_t * np.einsum("mq,mq,mq,mqi,mqj->mij", inv_h, wA, kavg, lams[s], lams[t]))
                    self._add_blocks(acc, cs, ct_, blk)
            for ion, c in zip(ions, c_all):
                Dsub = np.asarray(ion["D"])
                fl = 0.5 * sum(Dsub[self.cell_sub[cc]][:, None] * np.einsum("mj,mqj->mq", c[cc], gn[s])
                               for s, cc in enumerate((cT, cN)))
                for s, (cs, sg_s) in enumerate(side):
                    np.add.at(b, cs, sg_s * F * ion["z"] * np.einsum("mq,mq,mqi->mi", wA, fl, lams[s]))
        if self.nmf:
            ce, ci, wA, N, lam_e, lam_i = self._membrane_q1(2)
            for (cr, lr, sr), (cc, lc, sc) in itertools.product(((ci, lam_i, 1.0), (ce, lam_e, -1.0)), repeat=2):
                self._add_blocks(acc, cr, cc, C_phi * sr * sc * np.einsum("mq,mqi,mqj->mij", wA, lr, lc))
            g = np.array(phi_M, float)
            if not splitting_scheme:
                g = g - sum(I_ch) / C_phi
            gq = np.einsum("qa,ma->mq", N, g)
            np.add.at(b, ci, C_phi * np.einsum("mq,mq,mqi->mi", wA, gq, lam_i))
            np.add.at(b, ce, -C_phi * np.einsum("mq,mq,mqi->mi", wA, gq, lam_e))
        return self._csr(acc), b.ravel()

    # -- concentration equations --------------------------------------------------------------------------------
    def assemble_knp(self, params, ions, c_all, phi, phi_M, I_ch, splitting_scheme=True, gamma=10.0, f_source=None):
        K = len(ions)
        ns = K - 1
        psi, C_M, F, dt = params["psi"], params["C_M"], params["F"], params["dt"]
        allc = np.arange(self.nc)
        pts, wts = quadrature("hexahedron", 2)
        det, G = self._geom(allc, np.broadcast_to(pts, (self.nc,) + pts.shape))
        lam = self._last_phi
        wq = det * wts[None, :]
        mass = np.einsum("cq,cqi,cqj->cij", wq, lam, lam)
        stiff = np.einsum("cq,cqid,cqjd->cij", wq, G, G)
        gphi = np.einsum("cj,cqjd->cqd", phi, G)
        drift = np.einsum("cq,cqid,cqd,cqj->cij", wq, G, gphi, lam)
        accs = [[] for _ in range(ns)]
        b = np.zeros((ns, self.nc, 8))
        for k in range(ns):
            Dc = np.asarray(ions[k]["D"])[self.cell_sub][:, None, None]
            self._add_blocks(accs[k], allc, allc, mass / dt + Dc * stiff + ions[k]["z"] * psi * Dc * drift)
            rhs = c_all[k] / dt
            if f_source is not None and k in f_source:
                rhs = rhs + np.where(self.cell_sub[:, None] == 0, f_source[k], 0.0)
            b[k] += np.einsum("cij,cj->ci", mass, rhs)
        if self.interior:
            cT, cN, wA, n, inv_h, lams, Gs = self._interior_q1()
            side = ((cT, 1.0), (cN, -1.0))
            gn = [np.einsum("mqjd,mqd->mqj", Gs[s], n) for s in range(2)]
            gpn = [np.einsum("mj,mqj->mq", phi[c], gn[s]) for s, (c, _) in enumerate(side)]
            for k in range(ns):
                Dsub, z = np.asarray(ions[k]["D"]), ions[k]["z"]
                Ds = [Dsub[self.cell_sub[cT]], Dsub[self.cell_sub[cN]]]
                Davg = 0.5 * (Ds[0] + Ds[1])
                beta = -z * psi * 0.5 * (Ds[0][:, None] * gpn[0] + Ds[1][:, None] * gpn[1])     # (m, q)
                up = np.where(beta > 0, 0, 1)
                for s, (cs, sg_s) in enumerate(side):
                    for t, (ct_, sg_t) in enumerate(side):
                        blk = (-0.5 * sg_s * np.einsum("mq,mqi,m,mqj->mij", wA, lams[s], Ds[t], gn[t])
                               - 0.5 * sg_t * np.einsum("mq,mqj,m,mqi->mij", wA, lams[t], Ds[s], gn[s])
                               + gamma * sg_s * sg_t * np.einsum("mq,m,mq,mqi,mqj->mij", inv_h, Davg, wA, lams[s], lams[t])
                               + sg_s * np.einsum("mq,mq,mqi,mqj->mij", np.where(up == t, beta, 0.0), wA, lams[s], lams[t]))
                        self._add_blocks(accs[k], cs, ct_, blk)
        if self.nmf:
            ce, ci, wA, N, lam_e, lam_i = self._membrane_q1(6)
            at = lambda field, c, lam_: np.einsum("mqj,mj->mq", lam_, field[c])
            asum = sum(np.asarray(ion["D"])[self.cell_sub][:, None] * ion["z"] ** 2 * c for ion, c in zip(ions, c_all))
            pm = np.einsum("qa,ma->mq", N, np.asarray(phi_M, float))
            It = np.einsum("qa,ma->mq", N, sum(I_ch))
            jump = at(phi, ci, lam_i) - at(phi, ce, lam_e)
            for k in range(ns):
                Dsub, z = np.asarray(ions[k]["D"]), ions[k]["z"]
                a_e = Dsub[self.cell_sub[ce]][:, None] * z * z * at(c_all[k], ce, lam_e) / at(asum, ce, lam_e)
                a_i = Dsub[self.cell_sub[ci]][:, None] * z * z * at(c_all[k], ci, lam_i) / at(asum, ci, lam_i)
                C_e, C_i = a_e * C_M / (F * z * dt), a_i * C_M / (F * z * dt)
                Ik = np.einsum("qa,ma->mq", N, I_ch[k])
                g_e = pm - dt / (C_M * a_e) * Ik
                g_i = pm - dt / (C_M * a_i) * Ik
                if splitting_scheme:
                    g_e, g_i = g_e + (dt / C_M) * It, g_i + (dt / C_M) * It
                np.add.at(b[k], ce, np.einsum("mq,mq,mqi->mi", wA, -C_e * g_e + C_e * jump, lam_e))
                np.add.at(b[k], ci, np.einsum("mq,mq,mqi->mi", wA, C_i * g_i - C_i * jump, lam_i))
        return [self._csr(a) for a in accs], b.reshape(ns, self.n)


def make_dg_oracle(x, cells, cell_type, cell_sub, mem_facets, mem_tags):
    """The restatement for the mesh's cell type."""
    cls = DGOracleQ1 if cell_type == "hexahedron" else DGOracle
    return cls(x, cells, cell_type, cell_sub, mem_facets, mem_tags)
