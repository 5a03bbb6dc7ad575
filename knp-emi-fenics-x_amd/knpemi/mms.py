"""Method-of-manufactured-solutions scaffolding for the EMI sub-problem (host side).

The reference's MMS drivers (`tests/run_mms_emi.py`, `emiWeakForm.create_rhs_mms` :244-285) add
analytic source terms to the EMI right-hand side and a Dirichlet condition on the outer boundary
(facet tag 5).  The operator A and the diffusive part of L are the hot path and are assembled on the
GPU as in production runs; the manufactured sources are test scaffolding and are integrated here on
the host with a degree-8 rule.

`mms` is a dictionary of callables instead of UFL expressions:
    'f_phi_e'(x), 'f_phi_i'(x)           volume sources on the ECS / the cell            (:265-266)
    'f_phi_m'(x, n), 'f_I_M'(x, n)        membrane sources; n = unit normal pointing out of the cell (:281-283)
    'phi_e_exact'(x), 'phi_i_exact'(x)    exact potentials
with x of shape (gdim, npoints).  Deviation from the reference: the Dirichlet data is taken from
`phi_e_exact`; the reference hard-codes sin(2 pi x) cos(2 pi y) (`emiWeakForm.py:359`), which does not
match the exact solution its own driver uses (`tests/run_mms_emi.py:172`).  The ECS Neumann term of
`create_rhs_mms` (:262) only touches rows that the Dirichlet condition replaces and is omitted.
"""
from __future__ import annotations

import numpy as np

from .fem import Function, transfer_meshtags_to_submesh
from .fem.mesh import compute_interface_data
from .pdeSolver import DirichletBC


def _gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def _tri_rule(m=5):
    """Collapsed (Duffy) Gauss rule on the reference triangle, exact to degree 2m - 2."""
    xa, wa = _gauss01(m)
    pts = np.array([[a, b * (1 - a)] for a in xa for b in xa])
    wts = np.array([wi * wj * (1 - a) for a, wi in zip(xa, wa) for wj in wa])
    return pts, wts


def _volume_load(mesh_sub, f):
    """int f v dx on a P1 triangle sub-mesh."""
    pts, wts = _tri_rule()
    N = np.stack([1 - pts[:, 0] - pts[:, 1], pts[:, 0], pts[:, 1]], axis=1)      # (nq, 3)
    X = mesh_sub.x[mesh_sub.cells]                                               # (nc, 3, 2)
    e1, e2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
    det = np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])
    xq = np.einsum("qa,cag->cqg", N, X)                                          # (nc, nq, 2)
    fq = f(xq.reshape(-1, 2).T).reshape(xq.shape[:2])
    loc = np.einsum("q,c,cq,qa->ca", wts, det, fq, N)
    b = np.zeros(mesh_sub.num_vertices)
    np.add.at(b, mesh_sub.cells, loc)
    return b


def emi_mms_rhs(form):
    """Vector of the manufactured source terms in the block order [phi_0, phi_1] (create_rhs_mms)."""
    mms, subs, mesh, ct, ft = form.mms, form.subdomain_list, form.mesh, form.ct, form.ft
    C_phi = float(form.physical_params["C_phi"])
    tags = list(subs)
    if mesh.cell_type != "triangle" or tags != [0, 1]:
        raise NotImplementedError("MMS scaffolding covers the reference's 2D two-domain test")
    ecs, ics = subs[0]["mesh_sub"], subs[1]["mesh_sub"]
    b_e = _volume_load(ecs, mms["f_phi_e"])
    b_i = _volume_load(ics, mms["f_phi_i"])
    # membrane terms: C_phi f_phi_m (v_i - v_e) - f_I_M v_e over every facet of every membrane model
    xq1, wq1 = _gauss01(5)
    for mm in subs[1]["mem_models"]:
        facets = ft.find(int(mm["ode"].tag))
        idata = compute_interface_data(ct, facets)
        fv = mesh.facets[facets]                                               # parent vertices (nF, 2)
        p0, p1 = mesh.x[fv[:, 0]], mesh.x[fv[:, 1]]
        length = np.linalg.norm(p1 - p0, axis=1)
        # normal pointing out of the cell: away from the cell-side vertex opposite to the facet
        opp = mesh.cells[idata[:, 2], idata[:, 3]]
        t = (p1 - p0) / length[:, None]
        n = np.stack([t[:, 1], -t[:, 0]], axis=1)
        flip = np.einsum("fg,fg->f", n, mesh.x[opp] - p0) > 0
        n[flip] *= -1.0
        N = np.stack([1 - xq1, xq1], axis=1)                                    # (nq, 2)
        xq = p0[:, None, :] + xq1[None, :, None] * (p1 - p0)[:, None, :]        # (nF, nq, 2)
        nq = np.repeat(n[:, None, :], len(xq1), axis=1)
        fm = mms["f_phi_m"](xq.reshape(-1, 2).T, nq.reshape(-1, 2).T).reshape(xq.shape[:2])
        fI = mms["f_I_M"](xq.reshape(-1, 2).T, nq.reshape(-1, 2).T).reshape(xq.shape[:2])
        loc_m = np.einsum("q,f,fq,qa->fa", wq1, length, C_phi * fm, N)
        loc_I = np.einsum("q,f,fq,qa->fa", wq1, length, fI, N)
        e = np.searchsorted(ecs.parent_vertices, fv)
        i = np.searchsorted(ics.parent_vertices, fv)
        np.add.at(b_i, i, loc_m)
        np.add.at(b_e, e, -loc_m - loc_I)
    return np.concatenate([b_e, b_i])


def emi_dirichlet_bc(mesh, ft, subdomain_list, phi, mms, boundary_marker=5):
    """Dirichlet condition for phi_e on the outer boundary (emiWeakForm.py:344-360)."""
    ecs = subdomain_list[0]
    sub_tag, _ = transfer_meshtags_to_submesh(ft, ecs["mesh_sub"], ecs["sub_vertex_to_parent"],
                                              ecs["sub_to_parent"])
    dofs = np.unique(ecs["mesh_sub"].facets[sub_tag.find(boundary_marker)])
    u_bc = Function(phi[0].function_space, name="u_bc")
    u_bc.interpolate(lambda x: mms["phi_e_exact"](x[:ecs["mesh_sub"].gdim]))
    return DirichletBC(u_bc, dofs, block=0)


def l2_error(u, exact):
    """|| u - exact ||_L2 over the sub-mesh of the P1 Function u (degree-8 rule)."""
    m = u.function_space.mesh
    pts, wts = _tri_rule()
    N = np.stack([1 - pts[:, 0] - pts[:, 1], pts[:, 0], pts[:, 1]], axis=1)
    X = m.x[m.cells]
    e1, e2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
    det = np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])
    xq = np.einsum("qa,cag->cqg", N, X)
    uq = np.einsum("qa,ca->cq", N, u.x._a[m.cells])
    ex = exact(xq.reshape(-1, 2).T).reshape(uq.shape)
    return float(np.sqrt(np.einsum("q,c,cq->", wts, det, (uq - ex) ** 2)))
