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
