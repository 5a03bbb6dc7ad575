"""CPU tests of the DG(P1)+SIP restatement (oracle/knpemi_dg_oracle.py).  There is no reference code
for this variant (see the oracle's header), so it is pinned by what the discretisation must satisfy: manufactured
solutions at second order in L2 in 2D and 3D, algebraic invariants, and agreement with the CG oracle's solution of the
same problem up to the discretisation error."""
import numpy as np
import pytest

import dg_cases as C


def _rates(errs):
    errs = np.atleast_2d(np.array(errs, float).T).T
    return errs, np.log2(errs[:-1] / errs[1:])


@pytest.mark.parametrize("dim,sizes", [(2, (8, 16, 32)), (3, (4, 8))])
def test_dg_emi_volume_and_facet_terms_recover_boltzmann_potential(dim, sizes):
    out = [C.emi_boltzmann(C.OracleBackend(dim, M, False)) for M in sizes]
    errs, rates = _rates([o[0] for o in out])
    assert rates[-1] > 1.8 and errs[-1] < 2e-2, (errs, rates)
    A, b = out[0][1], out[0][2]
    assert abs(A - A.T).max() < 1e-12 * abs(A).max()                 # SIP is symmetric
    assert np.abs(A @ np.ones(A.shape[0])).max() < 1e-10 * abs(A).max()   # constants are in the kernel
    assert abs(b.sum()) < 1e-10 * np.abs(b).sum()                    # compatible right-hand side


def test_dg_emi_matrix_is_positive_semidefinite_with_the_default_penalty():
    be = C.OracleBackend(2, 6, True)
    _, _, A, _ = C.emi_membrane(be, True)
    w = np.linalg.eigvalsh(A.toarray())
    assert w[0] > -1e-10 * w[-1] and w[1] > 1e-8 * w[-1], w[:3]      # one zero eigenvalue (the constant), no more


@pytest.mark.parametrize("splitting", [True, False])
@pytest.mark.parametrize("dim,sizes", [(2, (16, 32, 64)), (3, (8, 12))])
def test_dg_emi_membrane_coupling_recovers_jump(dim, sizes, splitting):
    if dim == 3 and not splitting:
        sizes = (4, 8)
    out = [C.emi_membrane(C.OracleBackend(dim, M, True), splitting) for M in sizes]
    errs, rates = _rates([o[0] for o in out])
    rates = rates / np.log2(sizes[-1] / sizes[-2])
    assert rates[-1] > 1.7 and errs[-1] < 2e-2 and out[-1][1] < 5e-3, (errs, rates, out[-1][1])


@pytest.mark.parametrize("dim,sizes", [(2, (8, 16, 32)), (3, (4, 8))])
def test_dg_knp_volume_facet_and_upwind_terms(dim, sizes):
    out = [C.knp_volume(C.OracleBackend(dim, M, False)) for M in sizes]
    errs, rates = _rates([o[0] for o in out])
    # 3D: 4 -> 8 cells per edge is still pre-asymptotic (1.71; 8 -> 16 gives 1.90 but takes minutes of sparse LU)
    assert np.all(rates[-1] > (1.8 if dim == 2 else 1.6)) and np.all(errs[-1] < 7e-2), (errs, rates)
    # conservation: every column of A_k - M/dt sums to zero (the diffusive and the upwinded drift flux only move mass)
    As, _ = out[0][1], out[0][2]
    be = C.OracleBackend(dim, sizes[0], False)
    nv = dim + 1
    mass_col = np.repeat(be.vol / nv, nv) / C.K.DT
    for A in As:
        assert np.abs(np.asarray(A.sum(axis=0)).ravel() - mass_col).max() < 1e-10 * abs(A).max()


@pytest.mark.parametrize("splitting", [True, False])
@pytest.mark.parametrize("dim,sizes", [(2, (16, 32, 64)), (3, (8, 12))])
def test_dg_knp_membrane_terms(dim, sizes, splitting):
    if dim == 3 and not splitting:
        pytest.skip("covered in 2D; the 3D case runs with the default splitting scheme")
    out = [C.knp_membrane(C.OracleBackend(dim, M, True), splitting) for M in sizes]
    errs, rates = _rates([o[0] for o in out])
    rates = rates / np.log2(sizes[-1] / sizes[-2])
    assert np.all(rates[-1] > (1.7 if dim == 2 else 1.6)) and np.all(errs[-1] < (2e-2 if dim == 2 else 8e-2)), (errs, rates)


def test_dg_q1_restatement_on_hexahedra_is_second_order_and_keeps_the_invariants():
    """Broken Q1 on hexahedra (DGOracleQ1; the reference's own 3-D cell type, make_mesh_3D.py:100-102): the manufactured
    problems at second order, symmetry of the SIP potential matrix with the constants in its kernel, a compatible
    right-hand side, and mass conservation of the concentration matrices."""
    out = [C.emi_boltzmann(C.OracleBackend(3, M, False, cell="hexahedron")) for M in (4, 8)]
    assert np.log2(out[0][0] / out[1][0]) > 1.8 and out[1][0] < 2e-2, [o[0] for o in out]
    A, b = out[0][1], out[0][2]
    assert abs(A - A.T).max() < 1e-12 * abs(A).max()
    assert np.abs(A @ np.ones(A.shape[0])).max() < 1e-10 * abs(A).max()
    assert abs(b.sum()) < 1e-10 * np.abs(b).sum()
    vol = [C.knp_volume(C.OracleBackend(3, M, False, cell="hexahedron")) for M in (4, 8)]
    assert np.all(np.log2(np.array(vol[0][0]) / np.array(vol[1][0])) > 1.7), [v[0] for v in vol]
    be = C.OracleBackend(3, 4, False, cell="hexahedron")
    mass_col = np.repeat(be.vol / 8, 8) / C.K.DT
    for Ak in vol[0][1]:
        assert np.abs(np.asarray(Ak.sum(axis=0)).ravel() - mass_col).max() < 1e-10 * abs(Ak).max()
    mem = [C.emi_membrane(C.OracleBackend(3, M, True, cell="hexahedron"), True) for M in (4, 8)]
    assert np.log2(mem[0][0] / mem[1][0]) > 1.7 and mem[1][1] < 5e-3, [m[:2] for m in mem]
    _, _, A, _ = mem[0]
    w = np.linalg.eigvalsh(A.toarray())
    assert w[0] > -1e-10 * w[-1] and w[1] > 1e-8 * w[-1], w[:3]      # one zero eigenvalue (the constant), no more
    km = [C.knp_membrane(C.OracleBackend(3, M, True, cell="hexahedron"), True)[0] for M in (8, 12)]
    assert np.all(np.log2(np.array(km[0]) / np.array(km[1])) / np.log2(1.5) > 1.6), km


def test_dg_penalty_parameter_only_changes_the_error_constant():
    e10 = C.emi_boltzmann(C.OracleBackend(2, 16, False, gamma=10.0))[0]
    e40 = C.emi_boltzmann(C.OracleBackend(2, 16, False, gamma=40.0))[0]
    assert 0.3 < e10 / e40 < 3.0, (e10, e40)


def test_dg_update_and_traces():
    import knpemi_dg_oracle as dg
    be = C.OracleBackend(2, 8, True)
    o = be.o
    rng = np.random.default_rng(0)
    c_new = [rng.random((o.nc, o.nv)) + 1 for _ in range(2)]
    phi = rng.random((o.nc, o.nv))
    ions = C.ions_unit()
    c_all, phi_M = o.update(ions, [0.0, 0.3], c_new, phi)
    assert np.allclose(sum(i["z"] * c for i, c in zip(ions, c_all)) + np.array([0.0, 0.3])[o.cell_sub][:, None], 0.0)
    pe, pi = o.traces(phi)
    assert np.array_equal(phi_M, pi - pe)
    # the membrane nodes see the ECS cell on one side and the intracellular cell on the other
    ce, ci = o.mem_cells()
    assert np.all(o.cell_sub[ce] == 0) and np.all(o.cell_sub[ci] == 1)
    Xe, _ = o.traces(be.X[:, :, 0])
    assert np.array_equal(Xe, be.XM[:, :, 0])


def test_dg_and_cg_restatements_agree_up_to_the_discretisation_error():
    """The same (not manufactured) problem through both oracles: unit square with an inner cell, smooth positive
    concentrations that differ across the membrane (non-zero diffusive currents), a smooth phi_M and channel currents.
    The two discretisations solve the same equations, so their potentials (up to the constant) and their updated
    concentrations must approach each other at second order under refinement."""
    import scipy.sparse.linalg as spla
    import driver
    import knpemi_oracle as o
    import knpemi_dg_oracle as dg
    from knpemi.fem import make_mesh_mms
    diffs = []
    for M in (16, 32):
        mesh, ct, ft = make_mesh_mms(M)
        cell_sub = ct.dense()
        sel = ft.values == 1
        mfac, mtag = mesh.facets[ft.indices[sel]], ft.values[sel]
        P = o.OracleProblem(mesh.x, mesh.cells, mesh.cell_type, cell_sub, mfac, mtag, {0: [], 1: [1]})
        D = dg.DGOracle(mesh.x, mesh.cells, mesh.cell_type, cell_sub, mfac, mtag)
        dt = 0.02
        params = dict(dt=dt, F=1.0, psi=1.0, C_M=1.0, C_phi=1.0 / dt)
        zs, Ds = (1.0, -1.0, 1.0), ((1.0, 0.8), (1.3, 1.1), (0.9, 0.7))
        ions_cg = [dict(name=n, z=z, D={0: d[0], 1: d[1]}) for n, z, d in zip("abc", zs, Ds)]
        ions_dg = [dict(name=n, z=z, D=list(d)) for n, z, d in zip("abc", zs, Ds)]
        u = lambda X: np.cos(np.pi * X[0]) * np.cos(2 * np.pi * X[1])
        conc = lambda X, t: [2.0 + 0.3 * u(X) + 0.5 * t, 3.0 - 0.4 * u(X) + 0.2 * t]
        pm_f = lambda X: 0.3 + 0.1 * np.cos(2 * np.pi * X[0])
        phi_f = lambda X, t: 0.2 * np.sin(np.pi * X[0]) * np.cos(np.pi * X[1]) + 0.3 * t
        # CG fields on the sub-meshes
        c_all, phi_cg = {}, {}
        for t in (0, 1):
            X = P.sub[t]["x"].T
            c0, c1 = conc(X, t)
            c_all[t] = [c0, c1, -(zs[0] * c0 + zs[1] * c1) / zs[2]]
            phi_cg[t] = phi_f(X, t)
        XQ = P.mem[1]["x"].T
        phiM = {1: pm_f(XQ)}
        Ik = [0.2 * np.sin(2 * np.pi * XQ[1]), -0.1 * np.cos(2 * np.pi * XQ[0]), 0.05 + 0 * XQ[0]]
        mm = {1: [dict(tag=1, I_ch_k={n: Ik[k] for k, n in enumerate("abc")})]}
        A, _, b = o.assemble_emi(P, params, ions_cg, c_all, phiM, mm)
        x_cg = driver.solve_singular(A, b)
        Ak, bk = o.assemble_knp(P, params, ions_cg, c_all, phi_cg, phiM, mm, dt)
        c_cg = spla.splu(Ak.tocsc()).solve(bk)
        boff, _ = o.knp_block_offsets(P, 2)
        # the same fields as broken functions
        Xd = mesh.x[mesh.cells]                                   # (nc, nv, 2)
        shape = Xd.shape[:2]
        tt = np.repeat(cell_sub[:, None], shape[1], axis=1).ravel().astype(float)
        Xf = Xd.reshape(-1, 2).T
        c0, c1 = conc(Xf, tt)
        cd = [c0.reshape(shape), c1.reshape(shape), (-(zs[0] * c0 + zs[1] * c1) / zs[2]).reshape(shape)]
        XM = mesh.x[mfac]
        pm_d = pm_f(XM.reshape(-1, 2).T).reshape(len(mfac), 2)
        XMf = XM.reshape(-1, 2).T
        Ik_d = [(0.2 * np.sin(2 * np.pi * XMf[1])).reshape(len(mfac), 2), (-0.1 * np.cos(2 * np.pi * XMf[0])).reshape(len(mfac), 2),
                np.full((len(mfac), 2), 0.05)]
        Ad, bd = D.assemble_emi(params, ions_dg, cd, pm_d, Ik_d)
        x_dg = C.solve_pinned(Ad, bd).reshape(shape)
        Akd, bkd = D.assemble_knp(params, ions_dg, cd, phi_f(Xf, tt).reshape(shape), pm_d, Ik_d)
        c_dg = [C.solve(Akd[k], bkd[k]).reshape(shape) for k in range(2)]
        # CG values at the broken dofs: vertex of the cell, on the cell's own sub-mesh
        def at_dofs(vec_by_sub):
            out = np.zeros(shape)
            for t in (0, 1):
                rows = cell_sub == t
                out[rows] = vec_by_sub[t][np.searchsorted(P.sub[t]["pv"], mesh.cells[rows])]
            return out
        w = D.vol[:, None] / shape[1] * np.ones(shape)
        xc = at_dofs({t: x_cg[P.off[t]:P.off[t] + P.N[t]] for t in (0, 1)})
        e_phi = (x_dg - xc) - np.sum(w * (x_dg - xc)) / np.sum(w)
        d_phi = np.sqrt(np.sum(w * e_phi ** 2)) / np.sqrt(np.sum(w * (xc - np.sum(w * xc) / np.sum(w)) ** 2))
        d_c = []
        for k in range(2):
            ck = at_dofs({t: c_cg[boff[(t, k)]:boff[(t, k)] + P.N[t]] for t in (0, 1)})
            # compare the step increments: the fields themselves agree trivially to the size of c
            ref = cd[k]
            d_c.append(np.sqrt(np.sum(w * (c_dg[k] - ck) ** 2)) / np.sqrt(np.sum(w * (ck - ref) ** 2)))
        diffs.append([d_phi] + d_c)
    diffs = np.array(diffs)
    print("DG vs CG relative differences (phi, c_0 increment, c_1 increment):", diffs)
    assert np.all(diffs[1] < 0.1) and np.all(diffs[0] / diffs[1] > 2.5), diffs


def _dg_2d_run(r=1, g_syn=10.0):
    """BASELINE configs[0] on the DG restatement: 2D idealized mesh, K / Cl / Na with the reference's initial
    concentrations (run_2D.py:190-251), HH on the membrane facets' nodes, synaptic stimulus on x < 20 um."""
    import dg_driver
    import knpemi_oracle as o
    import knpemi_dg_oracle as dg
    import setup_problem as sp_
    from knpemi.fem.idealized import make_mesh_2D
    mesh, ct, ft = make_mesh_2D(r)
    sel = ft.values == 1
    mfac = mesh.facets[ft.indices[sel]]
    D = dg.DGOracle(mesh.x, mesh.cells, mesh.cell_type, ct.dense(), mfac, ft.values[sel])
    ions = [dict(name="K", z=1.0, D=[sp_.D_K] * 2), dict(name="Cl", z=-1.0, D=[sp_.D_CL] * 2), dict(name="Na", z=1.0, D=[sp_.D_NA] * 2)]
    params = dict(dt=sp_.DT, F=sp_.FARADAY, psi=sp_.PSI, C_M=sp_.C_M)
    ins = (D.cell_sub > 0)[:, None] * np.ones((1, D.nv), bool)
    c_all = [np.where(ins, i, e) for e, i in ((sp_.K_E, sp_.K_I), (sp_.CL_E, sp_.CL_I), (sp_.NA_E, sp_.NA_I))]
    m = o.MODELS["hh_si"]
    ix = m["pidx"]
    prow = np.array(m["params"], float)
    prow[ix["Cm"]] = sp_.C_M
    prow[ix["z_Na"]], prow[ix["z_K"]], prow[ix["z_Cl"]], prow[ix["psi"]] = 1.0, 1.0, -1.0, sp_.PSI
    nq = D.nmf * D.nf
    XM = mesh.x[mfac].reshape(nq, 2)
    mask = XM[:, 0] < 20e-6
    run = dg_driver.DGOracleRun(D, params, ions, "hh_si", c_all, np.tile(np.array(m["states"]), (nq, 1)), np.tile(prow, (nq, 1)),
                                mask, {ix["stim_amplitude"]: g_syn})
    return run, XM, mask


def test_dg_time_loop_follows_the_cg_time_loop():
    """Ten time steps of BASELINE configs[0] with the DG restatement against the CG oracle loop (oracle/driver.py, the
    path that is parity-tested against the GPU): the stimulated cell depolarises by the same amount -- the two
    discretisations of one model differ by their (second-order) errors only, everything else (ODE coupling at the facet
    nodes, splitting scheme, update, units) has to be the same."""
    import driver
    from helpers import Setup
    import contextlib, io
    run, XM, mask = _dg_2d_run()
    with contextlib.redirect_stdout(io.StringIO()):
        s = Setup("2d", 1, g_syn=10.0, build_forms=False)
    o, P, params, ions = s.oracle()
    c_all, _, _, _ = s.oracle_fields()
    ode = s.mem_models[0]['ode']
    cmask = np.array([x[0] < 20e-6 for x in ode.dof_locations])
    cg = driver.OracleRun(P, params, ions, "hh_si", c_all, ode.states.copy(), ode.parameters.copy(), ode.dof_locations, cmask,
                          {o.MODELS["hh_si"]["pidx"]["stim_amplitude"]: 10.0}, {'z': -1, 0: 0.0, 1: 0.0})
    for _ in range(10):
        run.step()
        cg.step()
    v_dg, v_cg = run.phiM.ravel(), cg.phiM[1]
    rest = -0.07438609374462003
    dep_dg, dep_cg = v_dg.mean() - rest, v_cg.mean() - rest
    print("depolarisation after 10 steps: DG", dep_dg, "CG", dep_cg)
    assert dep_cg > 0.010 and abs(dep_dg - dep_cg) < 0.03 * dep_cg
    # pointwise: every DG membrane node against the CG membrane dof at the same point
    key = lambda X: tuple(np.round(np.asarray(X)[:2] * 1e12).astype(np.int64))
    at = {key(x): v for x, v in zip(ode.dof_locations, v_cg)}
    ref = np.array([at[key(x)] for x in XM])
    assert np.abs(v_dg - ref).max() < 0.05 * dep_cg, np.abs(v_dg - ref).max()
    # the concentrations moved the same way (ECS potassium next to the stimulated membrane rises in both)
    ke_dg, _ = run.D.traces(run.c_all[0])
    ke_cg, _ = P.trace(1, cg.c_all[0][0], cg.c_all[1][0])
    assert abs((ke_dg.mean() - ke_cg.mean()) / (ke_cg.mean() - c_all[0][0].mean() + 1e-300)) < 0.2 or abs(ke_dg.mean() - ke_cg.mean()) < 1e-6
