"""CPU tests of the oracle: known answers, invariants and the committed golden vectors."""
import contextlib
import io
import os

import numpy as np
import pytest

import knpemi_oracle as o
from helpers import Setup, csr_rel_err, rel_err

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def quiet_setup(*a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return Setup(*a, build_forms=False, **k)


@pytest.mark.parametrize("cell,deg,npts", [("triangle", 2, 3), ("triangle", 6, 12), ("tetrahedron", 2, 4),
                                          ("interval", 6, 4), ("quadrilateral", 6, 16), ("hexahedron", 3, 8)])
def test_quadrature_exactness(cell, deg, npts):
    """Every rule integrates all monomials up to its degree exactly on the reference cell."""
    from math import factorial
    pts, wts = o.quadrature(cell, deg)
    assert len(wts) == npts
    d = pts.shape[1]
    simplex = cell in ("triangle", "tetrahedron")
    for powers in np.ndindex(*([deg + 1] * d)):
        if simplex and sum(powers) > deg:
            continue
        num = np.sum(wts * np.prod(pts ** np.array(powers), axis=1))
        if simplex:
            exact = np.prod([factorial(p) for p in powers]) / factorial(sum(powers) + d)
        else:
            exact = np.prod([1.0 / (p + 1) for p in powers])
        assert abs(num - exact) < 1e-14


def test_p1_element_known_answers():
    """Stiffness of the reference triangle / tetrahedron and Q1 mass of the unit cube."""
    for cell, X in (("triangle", np.array([[0, 0], [1, 0], [0, 1.]])),
                    ("tetrahedron", np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.]]))):
        pts, w = o.quadrature(cell, 1)
        phi, dphi = o.tabulate(cell, pts)
        det, G = o._geometry(X[None], dphi)
        K = np.einsum("q,cq,cqag,cqbg->ab", w, det, G, G)
        vol = 0.5 if cell == "triangle" else 1 / 6
        d = X.shape[1]
        ref = np.zeros((d + 1, d + 1))
        ref[0, 0] = d
        ref[0, 1:] = ref[1:, 0] = -1
        ref[1:, 1:] = np.eye(d)
        assert np.allclose(K, vol * ref, atol=1e-15)
    pts, w = o.quadrature("hexahedron", 2)
    phi, dphi = o.tabulate("hexahedron", pts)
    M = np.einsum("q,qa,qb->ab", w, phi, phi)
    assert abs(M.sum() - 1.0) < 1e-14 and abs(M[0, 0] - 1 / 27) < 1e-15 and abs(M[0, 7] - 1 / 216) < 1e-15


@pytest.mark.parametrize("kind,r", [("2d", 1), ("tet", 0), ("hex", 0)])
def test_assembly_invariants(kind, r):
    """Form-independent invariants: constant null space and symmetry of A_emi,
    zero-sum membrane RHS, A_knp row sums = lumped mass / dt when phi = 0."""
    s = quiet_setup(kind, r)
    s.perturb()
    _, P, params, ions = s.oracle()
    c_all, phi, phiM, mm = s.oracle_fields()
    A, Pm, b = o.assemble_emi(P, params, ions, c_all, phiM, mm)
    scale = np.abs(A.data).max()
    assert np.abs(A @ np.ones(A.shape[0])).max() < 1e-13 * scale
    assert csr_rel_err(A, A.T.tocsr()) < 1e-14
    c_flat = {t: [np.full_like(c, c.mean()) for c in c_all[t]] for t in c_all}   # grad c = 0
    _, _, b_gamma = o.assemble_emi(P, params, ions, c_flat, phiM, mm)
    assert abs(b_gamma.sum()) < 1e-12 * np.abs(b_gamma).max()
    zero_phi = {t: np.zeros_like(phi[t]) for t in phi}
    Ak, _ = o.assemble_knp(P, params, ions, c_all, zero_phi, phiM, mm, s.dt)
    measure = (Ak @ np.ones(Ak.shape[0])).sum() * s.dt / 2
    box = s.mesh.x.max(axis=0) - s.mesh.x.min(axis=0)
    assert abs(measure - np.prod(box)) < 1e-12 * np.prod(box)
    # P = A + ICS mass: rows of the ECS block are untouched
    D = (Pm - A).tocsr()
    assert abs(D[:P.N[0]]).sum() == 0.0 and abs(D.sum() - (np.prod(box) - 0) * 0) >= 0


@pytest.mark.parametrize("kind,r", [("2d", 1), ("tet", 0), ("hex", 0)])
def test_oracle_reproduces_golden(kind, r):
    g = np.load(os.path.join(GOLDEN, f"assembly_{kind}_r{r}.npz"))
    s = quiet_setup(kind, r)
    s.perturb(12345)
    _, P, params, ions = s.oracle()
    c_all, phi, phiM, mm = s.oracle_fields()
    for split in (True, False):
        tag = "split" if split else "nosplit"
        A, Pm, b = o.assemble_emi(P, params, ions, c_all, phiM, mm, splitting_scheme=split)
        Ak, bk = o.assemble_knp(P, params, ions, c_all, phi, phiM, mm, s.dt, splitting_scheme=split)
        v = np.random.default_rng(1).uniform(-1, 1, A.shape[0])
        vk = np.random.default_rng(2).uniform(-1, 1, Ak.shape[0])
        assert rel_err(b, g[f"{tag}_b_emi"]) < 1e-13 and rel_err(bk, g[f"{tag}_b_knp"]) < 1e-13
        assert rel_err(A @ v, g[f"{tag}_A_emi_v"]) < 1e-13 and rel_err(Ak @ vk, g[f"{tag}_A_knp_v"]) < 1e-13
        assert rel_err(Pm @ v, g[f"{tag}_P_emi_v"]) < 1e-13


def test_hh_initial_state_is_calibrated():
    """The reference's HH initial state is a steady state of its own RHS with its own initial
    concentrations (mm_hh.py:12-16, run_3D.py:191-197): |dV/dt| and gate rates are ~0."""
    g = np.load(os.path.join(GOLDEN, "ode_models.npz"))
    rhs0 = g["hh_si_stim0_rhs0"]
    y0 = g["hh_si_stim0_y0"]
    assert np.abs(rhs0[:3]).max() < 1e-6          # gates: 1/s
    assert abs(rhs0[3]) < 1e-6 * abs(y0[3]) / 1e-4  # V changes by < 1e-6 relative per time step


def test_ode_golden_reproduced_by_oracle():
    g = np.load(os.path.join(GOLDEN, "ode_models.npz"))
    for key in ("hh_si_stim10", "hh_mv_stim1", "glial_stim0"):
        model = key.rsplit("_stim", 1)[0]
        st, pa = g[f"{key}_y0"][None, :].copy(), g[f"{key}_p0"][None, :].copy()
        dt = float(g[f"{key}_dt"])
        for k in range(3):
            o.ode_sweep(model, st, pa, k * dt, dt)
            ref = g[f"{key}_traj"][k]
            assert rel_err(st[0], ref[:st.shape[1]]) < 1e-12


def test_oracle_mms_emi_convergence():
    """Analytic known answer for the oracle's EMI forms (volume + membrane coupling): the manufactured
    potentials of the reference's tests/run_mms_emi.py:166-179 are recovered at second order in L2."""
    import types
    import scipy.sparse.linalg as spla
    from knpemi import mms as M
    from knpemi.fem import Function, extract_submesh, functionspace, make_mesh_mms
    from knpemi.pdeSolver import _apply_bcs
    from mms_problem import CONC, MMS
    errs = []
    for Mx in (16, 32, 64):
        mesh, ct, ft = make_mesh_mms(Mx)
        s1, i2p, iv2p, _, _ = extract_submesh(mesh, ct, 1)
        s0, e2p, ev2p, _, _ = extract_submesh(mesh, ct, 0)
        g, g2p, _, _, _ = extract_submesh(mesh, ft, 1)
        subs = {0: dict(mesh_sub=s0, sub_to_parent=e2p, sub_vertex_to_parent=ev2p),
                1: dict(mesh_sub=s1, sub_to_parent=i2p, sub_vertex_to_parent=iv2p, mesh_mem=g, mem_to_parent=g2p,
                        mem_models=[{'ode': types.SimpleNamespace(tag=1), 'I_ch_k': {}}])}
        P = o.OracleProblem(mesh.x, mesh.cells, mesh.cell_type, ct.dense(), mesh.facets[ft.indices], ft.values,
                            {0: [], 1: [1]})
        params = dict(dt=1.0, F=1.0, psi=1.0, C_M=1.0, C_phi=1.0)
        ions = [dict(name=n, z=z, D={0: 1.0, 1: 1.0}) for n, z in (("a", 1.0), ("b", -1.0), ("c", 1.0))]
        c_all = {t: [CONC[n][min(t, 1)](P.sub[t]["x"].T) for n in "abc"] for t in (0, 1)}
        zq = np.zeros(P.NQ[1])
        A, _, b = o.assemble_emi(P, params, ions, c_all, {1: zq}, {1: [dict(tag=1, I_ch_k={n: zq for n in "abc"})]},
                                 splitting_scheme=False)
        form = types.SimpleNamespace(mms=MMS, subdomain_list=subs, mesh=mesh, ct=ct, ft=ft,
                                     physical_params={'C_phi': 1.0})
        b = b + M.emi_mms_rhs(form)
        phi = {0: Function(functionspace(s0)), 1: Function(functionspace(s1))}
        bc = M.emi_dirichlet_bc(mesh, ft, subs, phi, MMS)
        A2, b2 = _apply_bcs(A, b, [bc], [0, P.N[0]])
        xs = spla.splu(A2.tocsc()).solve(b2)
        phi[0].x.array[:] = xs[:P.N[0]]
        phi[1].x.array[:] = xs[P.N[0]:]
        errs.append((M.l2_error(phi[1], MMS["phi_i_exact"]), M.l2_error(phi[0], MMS["phi_e_exact"])))
    errs = np.array(errs)
    rates = np.log2(errs[:-1] / errs[1:])
    assert np.all(rates > 1.85) and np.all(errs[-1] < 2e-3), (errs, rates)


def test_oracle_mms_knp_convergence():
    """Analytic known answer for the oracle's KNP volume forms (mass / dt, diffusion, drift, source): the
    manufactured steady state of tests/mms_knp_problem.py is recovered at second order in L2."""
    import scipy.sparse.linalg as spla
    from knpemi.fem import create_unit_square
    import mms_knp_problem as K
    errs = []
    for M in (8, 16, 32):
        mesh = create_unit_square(None, M, M)
        ct = np.zeros(mesh.num_cells, np.int32)
        P = o.OracleProblem(mesh.x, mesh.cells, mesh.cell_type, ct, np.zeros((0, 2), np.int32), np.zeros(0, np.int32),
                            {0: [], 1: [1]})
        params = dict(dt=K.DT, F=1.0, psi=K.PSI, C_M=1.0, C_phi=1.0 / K.DT)
        ions = [dict(name=n, z=z, D={0: K.D, 1: K.D}) for n, z in zip("abc", K.Z)]
        X0 = P.sub[0]["x"].T
        empty = np.zeros(0)
        c_all = {0: [K.C_EXACT[0](X0), K.C_EXACT[1](X0), K.C_ELIM(X0)], 1: [empty, empty, empty]}
        phi = {0: K.PHI(X0), 1: empty}
        src = {k: K.F_SOURCE[k](X0) for k in range(2)}
        Ak, bk = o.assemble_knp(P, params, ions, c_all, phi, {1: empty}, {1: []}, K.DT, f_source=src)
        xs = spla.splu(Ak.tocsc()).solve(bk)
        n0 = P.N[0]
        sub = type("S", (), dict(x=P.sub[0]["x"], cells=P.sub[0]["cells"]))
        errs.append([K.l2_error_p1(sub, xs[k * n0:(k + 1) * n0], K.C_EXACT[k]) for k in range(2)])
    errs = np.array(errs)
    rates = np.log2(errs[:-1] / errs[1:])
    assert np.all(rates > 1.9) and np.all(errs[-1] < 2e-2), (errs, rates)


@pytest.mark.parametrize("splitting", [False, True])
def test_oracle_mms_knp_membrane_convergence(splitting):
    """Analytic known answer for the oracle's membrane terms of b_knp (alpha fractions, C, g, signs on both sides,
    with and without the splitting correction): tests/mms_knp_problem.py, membrane variant; second order in L2."""
    import scipy.sparse.linalg as spla
    from knpemi.fem import make_mesh_mms
    import mms_knp_problem as K
    errs = []
    for M in (16, 32, 64):
        mesh, ct, ft = make_mesh_mms(M)
        P = o.OracleProblem(mesh.x, mesh.cells, mesh.cell_type, ct.dense(), mesh.facets[ft.indices], ft.values,
                            {0: [], 1: [1]})
        params = dict(dt=K.DT_M, F=K.F_CONST, psi=K.PSI, C_M=K.C_M, C_phi=K.C_M / K.DT_M)
        ions = [dict(name=n, z=z, D={0: K.D, 1: K.D}) for n, z in zip("abc", K.Z)]
        X = {t: P.sub[t]["x"].T for t in (0, 1)}
        c_all = {t: [K.M_CPREV[0](X[t]), K.M_CPREV[1](X[t]), K.M_C[2](X[t])] for t in (0, 1)}
        phi = {0: K.M_PHI(X[0]) - K.PHI0, 1: K.M_PHI(X[1])}
        XQ = P.mem[1]["x"].T
        I = K.channel_currents(XQ)
        mm = {1: [dict(tag=1, I_ch_k={n: I[k] for k, n in enumerate("abc")})]}
        phiM = {1: K.membrane_potential_prev(XQ, splitting)}
        Ak, bk = o.assemble_knp(P, params, ions, c_all, phi, phiM, mm, K.DT_M, splitting_scheme=splitting)
        xs = spla.splu(Ak.tocsc()).solve(bk)
        boff, _ = o.knp_block_offsets(P, 2)
        e = []
        for t in (0, 1):
            sub = type("S", (), dict(x=P.sub[t]["x"], cells=P.sub[t]["cells"]))
            for k in range(2):
                e.append(K.l2_error_p1(sub, xs[boff[(t, k)]:boff[(t, k)] + P.N[t]], K.M_C[k]))
        errs.append(e)
    errs = np.array(errs)
    rates = np.log2(errs[:-1] / errs[1:])
    print("oracle KNP membrane MMS: errors", errs[-1], "rates", rates[-1])
    assert np.all(rates > 1.85) and np.all(errs[-1] < 1e-2), (errs, rates)


@pytest.mark.parametrize("cell_type", ["triangle", "tetrahedron", "hexahedron"])
def test_oracle_emi_volume_terms_recover_boltzmann_potential(cell_type):
    """Analytic known answer for the oracle's EMI volume forms in 2D and 3D (tests/mms_knp_problem.py, `emi_exact`):
    concentrations in Boltzmann equilibrium with a potential make that potential the solution; second order."""
    from knpemi.fem import create_box, create_unit_square
    import driver
    import mms_knp_problem as K
    errs = []
    for M in ((8, 16, 32) if cell_type == "triangle" else (4, 8)):
        mesh = create_unit_square(None, M, M) if cell_type == "triangle" else \
            create_box(None, [np.zeros(3), np.ones(3)], (M, M, M), cell_type)
        nvf = {"triangle": 2, "tetrahedron": 3, "hexahedron": 4}[cell_type]
        P = o.OracleProblem(mesh.x, mesh.cells, mesh.cell_type, np.zeros(mesh.num_cells, np.int32),
                            np.zeros((0, nvf), np.int32), np.zeros(0, np.int32), {0: [], 1: [1]})
        params = dict(dt=1.0, F=1.0, psi=K.PSI, C_M=1.0, C_phi=1.0)
        ions = [dict(name=n, z=z, D={0: K.D, 1: K.D}) for n, z in zip("abc", K.Z)]
        ph, cs = K.emi_exact(P.sub[0]["x"].T)
        empty = np.zeros(0)
        A, _, b = o.assemble_emi(P, params, ions, {0: cs, 1: [empty] * 3}, {1: empty}, {1: []})
        x = driver.solve_singular(A, b)
        errs.append(K.nodal_rms_error(x - x.mean(), ph - ph.mean()))
    rates = np.log2(np.array(errs[:-1]) / np.array(errs[1:]))
    assert rates[-1] > 1.8 and errs[-1] < 2e-2, (errs, rates)
