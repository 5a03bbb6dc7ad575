"""GPU tests of the DG(P1)+SIP variant: the HIP kernels (csrc/kernels_dg.hip) through the C ABI against
the CPU restatement (oracle/knpemi_dg_oracle.py) at 1e-10, the manufactured problems of tests/dg_cases.py through the
device assembly, the end-of-step update and the membrane ODE sweep over the membrane nodes."""
import numpy as np
import pytest

import dg_cases as C
from helpers import csr_rel_err, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _problem(dim, M, membrane, n_ions=3, gamma=10.0):
    from knpemi.dg import DGProblem
    from knpemi.fem import create_box, create_unit_square
    from knpemi.fem.idealized import _tag
    import knpemi_dg_oracle as dg
    mesh = create_unit_square(None, M, M) if dim == 2 else create_box(None, [np.zeros(3), np.ones(3)], (M, M, M),
                                                                       "tetrahedron")
    if membrane:
        ct, ft = _tag(mesh, [([0.25] * dim, [0.75] * dim)], [1], full_facet_tags=False)
    else:
        ct, ft = _tag(mesh, [], [], full_facet_tags=False)
    dp = DGProblem(mesh, ct, ft, [0, 1], [1], n_ions=n_ions)
    dp.gamma = gamma
    o = dg.DGOracle(mesh.x, mesh.cells, mesh.cell_type, dp.cell_sub, dp.mem_facets, dp.mem_tags)
    return dp, o


def _random_state(dp, K, seed):
    rng = np.random.default_rng(seed)
    shape, mshape = (dp.n_cells, dp.nv), (dp.nmf, dp.nf)
    smooth = lambda a, b: a + b * np.cos(3.0 * dp.X[:, :, 0] + 1.0) * np.sin(2.0 * dp.X[:, :, 1] + 0.5)
    c_all = [smooth(2.0 + k, 0.4) + 0.05 * rng.random(shape) for k in range(K)]
    phi = smooth(0.1, 0.3) + 0.02 * rng.random(shape)
    phi_M = 0.2 + 0.1 * rng.random(mshape)
    I_ch = [0.3 * rng.standard_normal(mshape) for _ in range(K)]
    src = {k: rng.standard_normal(shape) for k in range(K - 1)}
    return c_all, phi, phi_M, I_ch, src


def _push(dp, params, ions, c_all, phi, phi_M, I_ch, src=None, rho=None):
    dp.set_params(params, ions, rho=rho)
    for k, c in enumerate(c_all):
        dp.set_concentration(k, c)
    dp.set_potential(phi)
    dp.set_membrane_potential(phi_M)
    for k, I in enumerate(I_ch):
        dp.set_channel_current(k, I)
    if src is not None:
        for k, f in src.items():
            dp.set_source(k, f)


@pytest.mark.parametrize("splitting", [True, False])
@pytest.mark.parametrize("dim,M,K", [(2, 8, 3), (2, 12, 2), (2, 8, 4), (3, 4, 3), (3, 8, 3), (3, 4, 4)])
def test_dg_assembly_matches_oracle(hip_lib, dim, M, K, splitting):
    """A_emi, b_emi and the K-1 concentration systems, membrane terms and source included, different diffusivities on
    the two sides, non-default penalty: 1e-10 of the largest entry."""
    dp, o = _problem(dim, M, True, n_ions=K, gamma=7.5)
    zs = [1.0, -1.0, 2.0, -1.0][:K]
    ions = [dict(name=f"i{k}", z=zs[k], D=[1.0 + 0.3 * k, 0.6 + 0.2 * k]) for k in range(K)]
    params = dict(dt=0.05, F=1.3, psi=0.8, C_M=0.7)
    c_all, phi, phi_M, I_ch, src = _random_state(dp, K, 1)
    _push(dp, params, ions, c_all, phi, phi_M, I_ch, src)
    dp.assemble_emi(splitting)
    dp.assemble_knp(splitting)
    A, b = o.assemble_emi(params, ions, c_all, phi_M, I_ch, splitting_scheme=splitting, gamma=7.5)
    assert csr_rel_err(dp.matrix(0), A) < TOL and rel_err(dp.rhs(0), b) < TOL
    As, bs = o.assemble_knp(params, ions, c_all, phi, phi_M, I_ch, splitting_scheme=splitting, gamma=7.5, f_source=src)
    for k in range(K - 1):
        assert csr_rel_err(dp.matrix(1 + k), As[k]) < TOL, k
        assert rel_err(dp.rhs(1 + k), bs[k]) < TOL, k
    # the pattern is the one the header promises: one nv-wide block per cell and facet neighbour, sorted
    assert np.all(np.diff(dp.indptr) % dp.nv == 0)
    for r in (0, dp.n // 2, dp.n - 1):
        cols = dp.indices[dp.indptr[r]:dp.indptr[r + 1]]
        assert np.all(np.diff(cols) > 0) and r in cols


def _hex_problem(M, membrane, n_ions=3, gamma=10.0, distort=None, seed=5):
    """Unit cube of M^3 hexahedra, a cube of cells in the middle as the intracellular sub-domain; distort: None (boxes),
    "shear" (parallelepipeds with a full metric tensor), "random" (general trilinear cells: interior vertices moved)."""
    from knpemi.dg import DGProblem
    from knpemi.fem import create_box
    from knpemi.fem.idealized import _tag
    import knpemi_dg_oracle as dg
    mesh = create_box(None, [np.zeros(3), np.ones(3)], (M, M, M), "hexahedron")
    if membrane:
        ct, ft = _tag(mesh, [([0.25] * 3, [0.75] * 3)], [1], full_facet_tags=False)
    else:
        ct, ft = _tag(mesh, [], [], full_facet_tags=False)
    x = mesh.x
    if distort == "shear":
        S = np.array([[1.0, 0.25, -0.15], [0.1, 0.9, 0.2], [-0.2, 0.05, 1.1]])
        x[:] = x @ S.T
    elif distort == "rotate":
        c, s_ = np.cos(0.7), np.sin(0.7)
        R = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]]) @ np.array([[1.0, 0.0, 0.0], [0.0, np.cos(0.4), -np.sin(0.4)],
                                                                              [0.0, np.sin(0.4), np.cos(0.4)]])
        x[:] = (x * np.array([1.0, 0.6, 1.7])) @ R.T
    elif distort == "random":
        rng = np.random.default_rng(seed)
        inner = np.all((x > 1e-9) & (x < 1 - 1e-9), axis=1)
        x[inner] += (0.18 / M) * (rng.random((inner.sum(), 3)) - 0.5)
    dp = DGProblem(mesh, ct, ft, [0, 1], [1], n_ions=n_ions)
    dp.gamma = gamma
    o = dg.make_dg_oracle(mesh.x, mesh.cells, mesh.cell_type, dp.cell_sub, dp.mem_facets, dp.mem_tags)
    return dp, o


@pytest.mark.parametrize("splitting", [True, False])
@pytest.mark.parametrize("M,K,distort", [(4, 3, None), (4, 3, "shear"), (4, 3, "random"), (6, 2, "random"), (4, 4, "shear")])
def test_dg_q1_assembly_on_hexahedra_matches_oracle(hip_lib, M, K, distort, splitting):
    """Broken Q1 on hexahedra (csrc/kernels_dg_hex.hip: facet-aligned frames, shared Jacobian inverses) against the
    restatement that maps every point into each cell's own reference coordinates (DGOracleQ1): boxes, parallelepipeds
    with a full metric tensor and general trilinear cells, membrane terms and source included, different diffusivities
    on the two sides, 1e-10 of the largest entry."""
    dp, o = _hex_problem(M, True, n_ions=K, gamma=7.5, distort=distort)
    zs = [1.0, -1.0, 2.0, -1.0][:K]
    ions = [dict(name=f"i{k}", z=zs[k], D=[1.0 + 0.3 * k, 0.6 + 0.2 * k]) for k in range(K)]
    params = dict(dt=0.05, F=1.3, psi=0.8, C_M=0.7)
    c_all, phi, phi_M, I_ch, src = _random_state(dp, K, 1)
    _push(dp, params, ions, c_all, phi, phi_M, I_ch, src)
    dp.assemble_emi(splitting)
    dp.assemble_knp(splitting)
    A, b = o.assemble_emi(params, ions, c_all, phi_M, I_ch, splitting_scheme=splitting, gamma=7.5)
    assert csr_rel_err(dp.matrix(0), A) < TOL and rel_err(dp.rhs(0), b) < TOL
    As, bs = o.assemble_knp(params, ions, c_all, phi, phi_M, I_ch, splitting_scheme=splitting, gamma=7.5, f_source=src)
    for k in range(K - 1):
        assert csr_rel_err(dp.matrix(1 + k), As[k]) < TOL, k
        assert rel_err(dp.rhs(1 + k), bs[k]) < TOL, k
    assert dp.nv == 8 and dp.nf == 4 and np.all(np.diff(dp.indptr) % 8 == 0)
    for r in (0, dp.n // 2, dp.n - 1):
        cols = dp.indices[dp.indptr[r]:dp.indptr[r + 1]]
        assert np.all(np.diff(cols) > 0) and r in cols


@pytest.mark.parametrize("splitting", [True, False])
def test_dg_q1_box_mesh_kernels_agree_with_the_general_kernels(hip_lib, splitting, monkeypatch):
    """On a mesh of orthogonal parallelepipeds (every mesh of the reference's 3-D driver) knpemi_dg_create selects the
    box-mesh kernels (constant facet frames, no gradient tables, csrc/kernels_dg_hex.hip); KNPEMI_DG_HEX_GENERAL=1 keeps
    the general ones.  Both against the restatement, and against each other to rounding; a rotated box mesh (orthogonal
    cells that are not axis-aligned) takes the box kernels too."""
    K = 3
    ions = [dict(name=f"i{k}", z=z, D=[1.0 + 0.3 * k, 0.6 + 0.2 * k]) for k, z in enumerate((1.0, -1.0, 2.0))]
    params = dict(dt=0.05, F=1.3, psi=0.8, C_M=0.7)
    out = {}
    for name in ("box", "general", "rotated"):
        if name == "general":
            monkeypatch.setenv("KNPEMI_DG_HEX_GENERAL", "1")
        else:
            monkeypatch.delenv("KNPEMI_DG_HEX_GENERAL", raising=False)
        dp, o = _hex_problem(5, True, n_ions=K, gamma=7.5, distort="rotate" if name == "rotated" else None)
        c_all, phi, phi_M, I_ch, src = _random_state(dp, K, 1)
        _push(dp, params, ions, c_all, phi, phi_M, I_ch, src)
        dp.assemble_emi(splitting)
        dp.assemble_knp(splitting)
        A, b = o.assemble_emi(params, ions, c_all, phi_M, I_ch, splitting_scheme=splitting, gamma=7.5)
        assert csr_rel_err(dp.matrix(0), A) < TOL and rel_err(dp.rhs(0), b) < TOL, name
        As, bs = o.assemble_knp(params, ions, c_all, phi, phi_M, I_ch, splitting_scheme=splitting, gamma=7.5, f_source=src)
        for k in range(K - 1):
            assert csr_rel_err(dp.matrix(1 + k), As[k]) < TOL and rel_err(dp.rhs(1 + k), bs[k]) < TOL, (name, k)
        out[name] = [dp.matrix(w) for w in range(K)] + [dp.rhs(w) for w in range(K)]
    for w in range(K):
        assert csr_rel_err(out["box"][w], out["general"][w]) < 1e-13
        assert rel_err(out["box"][K + w], out["general"][K + w]) < 1e-12
    assert abs(out["box"][0] - out["general"][0]).max() > 0          # (two different kernels did run)


@pytest.mark.parametrize("dim", [2, 3, "hex"])
def test_dg_assembly_matches_oracle_at_physical_scales(hip_lib, dim):
    """The reference's idealized geometries and SI parameters (micrometre cells, D ~ 1e-9 m^2/s, dt = 0.1 ms,
    run_2D.py:174-251): determinants of 1e-14 .. 1e-21 and matrix entries spread over many decades must not cost the
    kernels' closed forms and fast reciprocals any accuracy against the quadrature-based restatement."""
    from knpemi.dg import DGProblem
    from knpemi.fem.idealized import make_mesh_2D, make_mesh_3D
    import knpemi_dg_oracle as dg
    mesh, ct, ft = make_mesh_2D(1) if dim == 2 else make_mesh_3D(0, "tetrahedron" if dim == 3 else "hexahedron")
    dp = DGProblem(mesh, ct, ft, [0, 1], [1])
    o = dg.make_dg_oracle(mesh.x, mesh.cells, mesh.cell_type, dp.cell_sub, dp.mem_facets, dp.mem_tags)
    ions = [dict(name="Na", z=1.0, D=[1.33e-9] * 2), dict(name="K", z=1.0, D=[1.96e-9] * 2), dict(name="Cl", z=-1.0, D=[2.03e-9] * 2)]
    params = dict(dt=1e-4, F=96485.0, psi=96485.0 / (8.314 * 300.0), C_M=0.02)
    ins = (dp.cell_sub > 0)[:, None] * np.ones((1, dp.nv), bool)
    w = np.sin(2e5 * dp.X[:, :, 0]) * np.cos(2e6 * dp.X[:, :, 1])
    c_all = [np.where(ins, i, e) * (1.0 + 1e-2 * w * (1 + k)) for k, (e, i) in enumerate(((100.0, 12.0), (4.0, 125.0), (104.0, 137.0)))]
    phi = np.where(ins, -0.0744, 0.0) + 1e-3 * w
    rng = np.random.default_rng(7)
    phi_M = -0.0744 + 1e-3 * rng.standard_normal((dp.nmf, dp.nf))
    I_ch = [1e-2 * rng.standard_normal((dp.nmf, dp.nf)) for _ in range(3)]
    _push(dp, params, ions, c_all, phi, phi_M, I_ch)
    for splitting in (True, False):
        dp.assemble_emi(splitting)
        dp.assemble_knp(splitting)
        A, b = o.assemble_emi(params, ions, c_all, phi_M, I_ch, splitting_scheme=splitting)
        assert csr_rel_err(dp.matrix(0), A) < TOL and rel_err(dp.rhs(0), b) < TOL
        As, bs = o.assemble_knp(params, ions, c_all, phi, phi_M, I_ch, splitting_scheme=splitting)
        for k in range(2):
            assert csr_rel_err(dp.matrix(1 + k), As[k]) < TOL and rel_err(dp.rhs(1 + k), bs[k]) < TOL, k


def test_dg_assembly_without_membrane_and_bit_reproducible(hip_lib):
    dp, o = _problem(3, 6, False)
    ions = C.ions_unit()
    params = dict(dt=0.1, F=1.0, psi=1.0, C_M=1.0)
    c_all, phi, phi_M, I_ch, _ = _random_state(dp, 3, 2)
    _push(dp, params, ions, c_all, phi, phi_M, I_ch)
    dp.assemble_emi()
    dp.assemble_knp()
    A0, A1 = dp.matrix(0), dp.matrix(1)
    A, b = o.assemble_emi(params, ions, c_all, phi_M, I_ch)
    assert csr_rel_err(A0, A) < TOL
    assert abs(A0 - A0.T).max() < 1e-12 * abs(A0).max()
    dp.assemble_emi()
    dp.assemble_knp()
    assert np.array_equal(dp.matrix(0).data, A0.data) and np.array_equal(dp.matrix(1).data, A1.data)


@pytest.mark.parametrize("cell", ["triangle", "hexahedron"])
def test_dg_update_matches_oracle(hip_lib, cell):
    dp, o = _problem(2, 8, True) if cell == "triangle" else _hex_problem(4, True, distort="random")
    d = dp.X.shape[2]
    ions = C.ions_unit()
    params = dict(dt=0.1, F=1.0, psi=1.0, C_M=1.0)
    c_all, phi, phi_M, I_ch, _ = _random_state(dp, 3, 3)
    rho = (-1.0, [0.0, 0.4])
    _push(dp, params, ions, c_all, phi, phi_M, I_ch, rho=rho)
    rng = np.random.default_rng(5)
    c_new = [rng.random((dp.n_cells, dp.nv)) + 1.0 for _ in range(2)]
    dp.update(np.array(c_new))
    want, pm = o.update(ions, [rho[0] * r for r in rho[1]], c_new, phi)
    for k in range(3):
        assert rel_err(dp.get_concentration(k), want[k]) < 1e-14
    assert rel_err(dp.get_membrane_potential(), pm) < 1e-14
    e, i = dp.membrane_dofs()
    assert np.all(dp.cell_sub[e // dp.nv] == 0) and np.all(dp.cell_sub[i // dp.nv] == 1)
    assert np.array_equal(dp.X.reshape(-1, d)[e.ravel()], dp.XM.reshape(-1, d))


class DeviceBackend:
    """tests/dg_cases.py backend on the HIP kernels."""

    def __init__(self, dim, M, membrane, gamma=10.0, cell="simplex"):
        self.dp, self.o = (_hex_problem(M, membrane, gamma=gamma) if cell == "hexahedron"
                           else _problem(dim, M, membrane, gamma=gamma))
        self.X, self.XM, self.cell_sub, self.vol = self.dp.X, self.dp.XM, self.dp.cell_sub, self.o.vol

    def emi(self, params, ions, c_all, phi_M, I_ch, splitting):
        dp = self.dp
        _push(dp, params, ions, c_all, np.zeros((dp.n_cells, dp.nv)), np.zeros((dp.nmf, dp.nf)) + phi_M,
              [np.zeros((dp.nmf, dp.nf)) + I for I in I_ch])
        dp.assemble_emi(splitting)
        return dp.matrix(0), dp.rhs(0)

    def knp(self, params, ions, c_all, phi, phi_M, I_ch, splitting, f_source):
        dp = self.dp
        _push(dp, params, ions, c_all, phi, np.zeros((dp.nmf, dp.nf)) + phi_M,
              [np.zeros((dp.nmf, dp.nf)) + I for I in I_ch], f_source)
        dp.assemble_knp(splitting)
        return [dp.matrix(1 + k) for k in range(len(ions) - 1)], np.array([dp.rhs(1 + k) for k in range(len(ions) - 1)])


@pytest.mark.parametrize("dim,sizes", [(2, (16, 32, 64)), (3, (8, 12))])
def test_dg_mms_rates_through_the_device_assembly(hip_lib, dim, sizes):
    """Second order in L2 for the potential (Boltzmann problem with a membrane jump) and the concentrations (membrane
    flux problem) with the matrices the HIP kernels produce."""
    e_emi, e_knp = [], []
    for M in sizes:
        be = DeviceBackend(dim, M, True)
        e_emi.append(C.emi_membrane(be, True)[0])
        e_knp.append(C.knp_membrane(be, True)[0])
    h = np.log2(sizes[-1] / sizes[-2])
    r_emi = np.log2(e_emi[-2] / e_emi[-1]) / h
    r_knp = np.log2(np.array(e_knp[-2]) / np.array(e_knp[-1])) / h
    print("DG MMS on the device: EMI", e_emi, r_emi, "KNP", e_knp, r_knp)
    assert r_emi > 1.7 and np.all(r_knp > 1.6), (e_emi, e_knp)


def test_dg_q1_mms_rates_through_the_device_assembly(hip_lib):
    """Broken Q1 on hexahedra: second order in L2 for the potential with a membrane jump and for the concentrations
    (volume problem with source; membrane flux problem) with the matrices the HIP kernels produce."""
    e_emi, e_vol, e_mem = [], [], []
    for M in (4, 8):
        e_emi.append(C.emi_membrane(DeviceBackend(3, M, True, cell="hexahedron"), True)[0])
        e_vol.append(C.knp_volume(DeviceBackend(3, M, False, cell="hexahedron"))[0])
    for M in (8, 12):
        e_mem.append(C.knp_membrane(DeviceBackend(3, M, True, cell="hexahedron"), True)[0])
    r_emi = np.log2(e_emi[0] / e_emi[1])
    r_vol = np.log2(np.array(e_vol[0]) / np.array(e_vol[1]))
    r_mem = np.log2(np.array(e_mem[0]) / np.array(e_mem[1])) / np.log2(1.5)
    print("DG(Q1) MMS on the device: EMI", e_emi, r_emi, "KNP volume", e_vol, r_vol, "KNP membrane", e_mem, r_mem)
    assert r_emi > 1.7 and np.all(r_vol > 1.6) and np.all(r_mem > 1.6), (e_emi, e_vol, e_mem)


def test_dg_volume_mms_with_source_on_device(hip_lib):
    errs = [C.knp_volume(DeviceBackend(2, M, False))[0] for M in (8, 16, 32)]
    rates = np.log2(np.array(errs[:-1]) / np.array(errs[1:]))
    assert np.all(rates[-1] > 1.8), (errs, rates)


@pytest.mark.parametrize("cell", ["triangle", "hexahedron"])
def test_dg_membrane_ode_sweep_matches_scipy_lsoda(hip_lib, cell):
    """Hodgkin-Huxley sweep over the membrane nodes of the broken space: traces of the concentrations taken from the
    two cells of every membrane facet, V <- phi_M, LSODA, phi_M <- V, I_ch_k <- parameter columns -- against ODEPACK
    (scipy) on the same tables, tolerances of the CG-path test (tests/test_gpu_parity.py)."""
    import knpemi_oracle as ko
    from knpemi import _lib as L
    from setup_problem import C_M, PSI
    dp, o = _problem(2, 8, True) if cell == "triangle" else _hex_problem(4, True)
    m = ko.MODELS["hh_si"]
    ix = m["pidx"]
    names = ["Na", "K", "Cl"]
    ions = [dict(name=n, z=z, D=[1e-9, 1e-9]) for n, z in zip(names, (1.0, 1.0, -1.0))]
    dp.set_params(dict(dt=1e-4, F=96485.0, psi=PSI, C_M=C_M), ions)
    ins = (dp.cell_sub > 0)[:, None]
    vary = 1.0 + 0.05 * np.cos(6.0 * dp.X[:, :, 0])
    conc = [np.where(ins, 12.0, 100.0) * vary, np.where(ins, 125.0, 4.0) * vary, np.where(ins, 137.0, 104.0) * vary]
    for k in range(3):
        dp.set_concentration(k, conc[k])
    prow = np.array(m["params"], float)
    prow[ix["Cm"]] = C_M
    prow[ix["z_Na"]], prow[ix["z_K"]], prow[ix["z_Cl"]], prow[ix["psi"]] = 1.0, 1.0, -1.0, PSI
    ion_param = [ix[f"{n}{s}"] for n in names for s in ("_e", "_i")]
    ion_param = sum(([ix[f"{n}_e"], ix[f"{n}_i"], ix[f"I_ch_{n}"]] for n in names), [])
    dp.ode_bind(L.MODEL_HH_SI, m["states"], prow, ion_param, m["V"])
    nq = dp.nmf * dp.nf
    st_o, p_o = np.tile(np.array(m["states"]), (nq, 1)), np.tile(prow, (nq, 1))
    t = 0.0
    for step in range(3):
        if step:
            pm = dp.get_membrane_potential() + 1e-3 * np.cos(5.0 * dp.XM[:, :, 1])
            dp.set_membrane_potential(pm)
            st_o[:, m["V"]] = pm.ravel()
        dp.ode_step(t, 1e-4, set_v=step > 0)
        for k, n in enumerate(names):
            te, ti = o.traces(conc[k])
            p_o[:, ix[f"{n}_e"]], p_o[:, ix[f"{n}_i"]] = te.ravel(), ti.ravel()
        ko.ode_sweep("hh_si", st_o, p_o, t, 1e-4)
        t += 1e-4
        st, pr = dp.ode_tables()
        assert rel_err(st, st_o) < 1e-6
        ich = [ix["I_ch_Na"], ix["I_ch_K"], ix["I_ch_Cl"]]
        assert np.abs(pr[:, ich] - p_o[:, ich]).max() / max(np.abs(p_o[:, ich]).max(), 1e-3) < 1e-5
        assert rel_err(dp.get_membrane_potential().ravel(), st[:, m["V"]]) == 0.0
        for k, n in enumerate(names):
            assert np.array_equal(dp.get_channel_current(k).ravel(), pr[:, ix[f"I_ch_{n}"]])
    n_rhs, n_steps, n_failed = dp.ode_stats()
    assert n_failed == 0 and n_rhs > 0


def test_dg_error_paths(hip_lib):
    from knpemi.dg import DGProblem
    from knpemi.fem import create_box, create_unit_square
    from knpemi.fem.idealized import _tag
    from knpemi._lib import KnpemiError
    mesh = create_unit_square(None, 8, 8)
    ct, ft = _tag(mesh, [([0.25] * 2, [0.75] * 2)], [1], full_facet_tags=False)
    with pytest.raises(KnpemiError, match="not in mem_facets"):
        DGProblem(mesh, ct, ft, [0, 1], [7])                # the interface is not declared a membrane
    hexm = create_box(None, [np.zeros(3), np.ones(3)], (2, 2, 2), "hexahedron")
    hexm.cells[:] = hexm.cells[:, [0, 1, 3, 2, 4, 5, 7, 6]]      # counter-clockwise faces instead of the tensor order
    with pytest.raises(KnpemiError, match="tensor-product vertex order"):
        DGProblem(hexm, np.zeros(hexm.num_cells, np.int32), np.zeros(len(hexm.facets), np.int32), [0], [1])
    dp = DGProblem(mesh, ct, ft, [0, 1], [1])
    with pytest.raises(KnpemiError, match="set_params"):
        dp.assemble_emi()
    with pytest.raises(ValueError):
        dp.set_potential(np.zeros(3))
    with pytest.raises(KnpemiError, match="solve_knp has not been called"):
        dp.solution()
    with pytest.raises(KnpemiError, match="set_params"):
        dp.solve_knp(update=True)
    # a solve that cannot reach its tolerance in maxit iterations reports it (ksp_error_if_not_converged)
    params = dict(dt=0.05, F=1.0, psi=1.0, C_M=1.0)
    ions = [dict(z=z, D=[1.0, 0.7]) for z in (1.0, -1.0, 1.0)]
    c_all, phi, phi_M, I_ch, _ = _random_state(dp, 3, 5)
    _push(dp, params, ions, c_all, phi, phi_M, I_ch)
    big = DGProblem(create_unit_square(None, 24, 24), *_tag(create_unit_square(None, 24, 24), [([0.25] * 2, [0.75] * 2)], [1],
                                                            full_facet_tags=False), [0, 1], [1])
    c_all, phi, phi_M, I_ch, _ = _random_state(big, 3, 5)
    _push(big, params, ions, c_all, phi, phi_M, I_ch)
    big.assemble_emi()
    with pytest.raises(KnpemiError, match="did not converge"):
        big.solve_emi(rtol=1e-14, maxit=2)
    its, rr = big.solve_emi(rtol=1e-9)
    assert 0 < its < 100 and rr < 1e-9


def test_dg_time_loop_on_the_device_matches_the_restatement(hip_lib):
    """BASELINE configs[0] with the DG variant, as examples/idealized_geometries/run_2D_dg.py runs it (facet-node ODE
    sweep, both assemblies and the update on the GPU, SciPy solves of the device-assembled systems): ten steps against
    the loop of the CPU restatement (oracle/dg_driver.py, which tests/test_dg_oracle.py ties to the CG loop)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "idealized_geometries"))
    import run_2D_dg
    from test_dg_oracle import _dg_2d_run
    ref, XM, mask = _dg_2d_run()
    run = run_2D_dg.DGRun(1, device_solves=False)
    assert np.array_equal(run.dp.XM.reshape(-1, 2), XM) and np.array_equal(run.stimulated, mask)
    for _ in range(10):
        run.step()
        ref.step()
    dp = run.dp
    assert rel_err(dp.get_membrane_potential(), ref.phiM) < 1e-8
    st, pr = dp.ode_tables()
    assert rel_err(st, ref.states) < 1e-8
    p_dev, p_ref = dp.get_potential(), ref.phi
    assert rel_err(p_dev - p_dev.mean(), p_ref - p_ref.mean()) < 1e-8
    for k in range(3):
        assert rel_err(dp.get_concentration(k), ref.c_all[k]) < 1e-10
    v = dp.get_membrane_potential()
    assert v.mean() > -0.0744 + 0.010


@pytest.mark.parametrize("dim", [2, 3, "hex"])
def test_dg_device_solves_match_direct_solves(hip_lib, dim):
    """knpemi_dg_solve_emi / knpemi_dg_solve_knp (CG with the constants projected out / BiCGStab, AMG over the continuous
    P1 auxiliary space; pdeSolver.py:24-35,74-78,99-110) on the systems of the idealized geometries at SI scales against
    SciPy's sparse LU of the same device-assembled matrices."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "idealized_geometries"))
    import scipy.sparse.linalg as spla
    from run_2D_dg import solve_singular
    from knpemi.dg import DGProblem
    from knpemi.fem.idealized import make_mesh_2D, make_mesh_3D
    mesh, ct, ft = make_mesh_2D(2) if dim == 2 else make_mesh_3D(0, "tetrahedron" if dim == 3 else "hexahedron")
    dp = DGProblem(mesh, ct, ft, [0, 1], [1])
    assert dp.n > 640                                               # beyond the size the AMG inverts directly
    ions = [dict(name="Na", z=1.0, D=[1.33e-9] * 2), dict(name="K", z=1.0, D=[1.96e-9] * 2), dict(name="Cl", z=-1.0, D=[2.03e-9] * 2)]
    params = dict(dt=1e-4, F=96485.0, psi=96485.0 / (8.314 * 300.0), C_M=0.02)
    ins = (dp.cell_sub > 0)[:, None] * np.ones((1, dp.nv), bool)
    w = np.sin(2e5 * dp.X[:, :, 0]) * np.cos(2e6 * dp.X[:, :, 1])
    c_all = [np.where(ins, i, e) * (1.0 + 1e-2 * w * (1 + k)) for k, (e, i) in enumerate(((100.0, 12.0), (4.0, 125.0), (104.0, 137.0)))]
    rng = np.random.default_rng(11)
    phi_M = -0.0744 + 1e-3 * rng.standard_normal((dp.nmf, dp.nf))
    I_ch = [1e-2 * rng.standard_normal((dp.nmf, dp.nf)) for _ in range(3)]
    _push(dp, params, ions, c_all, np.where(ins, -0.0744, 0.0), phi_M, I_ch)
    dp.assemble_emi()
    A, b = dp.matrix(0), dp.rhs(0)
    rtol = 1e-11 if dim == 2 else 1e-8          # (the 3D residual stalls near 4e-10 ||b||: entries spread over decades)
    its, rr = dp.solve_emi(rtol=rtol)
    phi = dp.get_potential().reshape(-1)
    assert 0 < its < 150 and rr < rtol, (its, rr)
    assert abs(phi.mean()) < 1e-12 * np.abs(phi).max()
    bp = b - b.mean()                                               # the solver projects b (compatible to rounding)
    assert np.linalg.norm(A @ phi - bp) < 2 * rtol * np.linalg.norm(bp)
    if dim == 2:                                                    # (a bordered sparse LU of the 3D system fills in)
        ref = solve_singular(A, bp)
        assert rel_err(phi, ref - ref.mean()) < 1e-7
    dp.assemble_knp()                                               # with the potential just solved
    its, rr = dp.solve_knp(rtol=1e-12)
    assert 0 < its < 150 and rr < 1e-12, (its, rr)
    c = dp.solution()
    for k in range(2):
        Ak, bk = dp.matrix(1 + k), dp.rhs(1 + k)
        assert np.linalg.norm(Ak @ c[k] - bk) < 1e-11 * np.linalg.norm(bk), k
        if dim == 2:
            assert rel_err(c[k], spla.splu(Ak.tocsc()).solve(bk)) < 1e-9, k


@pytest.mark.parametrize("cell", ["tetrahedron", "hexahedron"])
def test_dg_split_auxiliary_space_lowers_the_iteration_count(hip_lib, cell, monkeypatch):
    """The first coarse level of the DG hierarchies: the broken dofs at a (sub-domain, vertex) are split into the connected
    components of their strong couplings (KnAmg::split_first) and strongly positively coupled unknowns stay in different
    aggregates below (aggregate_apart).  On the 10:1 cells of the idealized geometries the continuous auxiliary space
    (KNPEMI_DG_AUX_UNSPLIT=1) needs at least twice the iterations; both converge to the same solution."""
    from knpemi.dg import DGProblem
    from knpemi.fem.idealized import make_mesh_3D
    mesh, ct, ft = make_mesh_3D(0, cell)
    ions = [dict(name="Na", z=1.0, D=[1.33e-9] * 2), dict(name="K", z=1.0, D=[1.96e-9] * 2), dict(name="Cl", z=-1.0, D=[2.03e-9] * 2)]
    params = dict(dt=1e-4, F=96485.0, psi=96485.0 / (8.314 * 300.0), C_M=0.02)
    rng = np.random.default_rng(3)
    out = {}
    for mode in ("split", "unsplit"):
        if mode == "unsplit":
            monkeypatch.setenv("KNPEMI_DG_AUX_UNSPLIT", "1")
            monkeypatch.setenv("KNPEMI_DG_PLAIN_AGGREGATION", "1")
        dp = DGProblem(mesh, ct, ft, [0, 1], [1])
        ins = (dp.cell_sub > 0)[:, None] * np.ones((1, dp.nv), bool)
        if mode == "split":
            w = rng.standard_normal((dp.n_cells, dp.nv))
            phi_M = -0.0744 + 1e-3 * rng.standard_normal((dp.nmf, dp.nf))
            I_ch = [1e-2 * rng.standard_normal((dp.nmf, dp.nf)) for _ in range(3)]
        c_all = [np.where(ins, i, e) * (1.0 + 1e-3 * w) for (e, i) in ((100.0, 12.0), (4.0, 125.0), (104.0, 137.0))]
        _push(dp, params, ions, c_all, np.where(ins, -0.0744, 0.0), phi_M, I_ch)
        dp.assemble_emi()
        its_e, rr = dp.solve_emi(rtol=1e-8)
        phi = dp.get_potential().copy()
        dp.assemble_knp()
        its_k, rr = dp.solve_knp(rtol=1e-10)
        out[mode] = (its_e, its_k, phi, dp.solution().copy())
    (e1, k1, p1, c1), (e0, k0, p0, c0) = out["split"], out["unsplit"]
    assert 2 * e1 <= e0 and 2 * k1 <= k0, (e1, e0, k1, k0)
    assert rel_err(p1, p0) < 1e-5 and rel_err(c1, c0) < 1e-8


def test_dg_time_loop_with_device_solves_follows_the_direct_solves(hip_lib):
    """The whole DG time step on the device (ODE sweep, assemblies, CG / BiCGStab + AMG at the reference's rtol
    1e-5 / 1e-7, update without leaving the device) against the same loop with SciPy's direct solves: ten steps of
    BASELINE configs[0] while the stimulated membrane depolarises."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "idealized_geometries"))
    import run_2D_dg
    dev, host = run_2D_dg.DGRun(1), run_2D_dg.DGRun(1, device_solves=False)
    for _ in range(10):
        dev.step()
        host.step()
    assert all(0 <= a < 60 and 0 <= b < 30 for a, b in dev.iterations), dev.iterations
    v_dev, v_host = dev.dp.get_membrane_potential(), host.dp.get_membrane_potential()
    assert v_host.mean() > -0.0744 + 0.010
    assert rel_err(v_dev, v_host) < 1e-4
    for k in range(3):
        assert rel_err(dev.dp.get_concentration(k), host.dp.get_concentration(k)) < 1e-6
    tight = run_2D_dg.DGRun(1, rtol=(1e-10, 1e-12))
    for _ in range(10):
        tight.step()
    assert rel_err(tight.dp.get_membrane_potential(), v_host) < 1e-7


@pytest.mark.parametrize("r,label", [(1, "config2: 124 416 tetrahedra"), (2, "config3: 995 328 tetrahedra")])
def test_dg_full_size_properties(hip_lib, r, label):
    """BASELINE sizes, checked through what needs no oracle: the potential matrix is symmetric with the constants in
    its kernel and a compatible right-hand side; every column of A_k - M/dt sums to zero (the SIP and the upwinded drift
    fluxes only move mass between cells; membrane fluxes are right-hand-side terms); rows are sorted and hold one
    nv-wide block per cell and neighbour; a second assembly reproduces every bit."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dg_time
    dp = dg_time.build(r)
    n = dp.n
    assert dp.n_cells == 124416 * 8 ** (r - 1)
    dp.assemble_emi()
    dp.assemble_knp()
    A = dp.matrix(0)
    scale = abs(A.data).max()
    assert abs(A - A.T).max() < 1e-12 * scale
    assert np.abs(A @ np.ones(n)).max() < 1e-10 * scale
    b = dp.rhs(0)
    assert abs(b.sum()) < 1e-9 * np.abs(b).sum()
    assert np.all(np.diff(dp.indptr) % dp.nv == 0) and int(np.diff(dp.indptr).max()) == dp.nv * (dp.nv + 1)
    rows = np.random.default_rng(0).integers(0, n, 2000)
    for i in rows:
        cols = dp.indices[dp.indptr[i]:dp.indptr[i + 1]]
        assert np.all(np.diff(cols) > 0) and i in cols
    # cell volumes from the coordinates
    X = dp.X
    e = X[:, 1:] - X[:, :1]
    vol = np.abs(np.einsum("ci,ci->c", e[:, 0], np.cross(e[:, 1], e[:, 2]))) / 6.0
    mass_col = np.repeat(vol / dp.nv, dp.nv) / 1e-4
    sums, datas = [], []
    for k in range(2):
        Ak = dp.matrix(1 + k)
        col = np.asarray(Ak.sum(axis=0)).ravel()
        assert np.abs(col - mass_col).max() < 1e-9 * abs(Ak.data).max(), k
        datas.append(Ak.data.copy())
    dp.assemble_emi()
    dp.assemble_knp()
    assert np.array_equal(dp.matrix(0).data, A.data)
    for k in range(2):
        assert np.array_equal(dp.matrix(1 + k).data, datas[k])


def test_dg_q1_assembly_matches_oracle_on_20k_hexahedra(hip_lib):
    """The box-mesh kernels on the reference's 3-D geometry at r = 1 (20 736 hexahedra, 2 592 workgroups: the XCD-aware
    workgroup order with 324 workgroups per XCD, neighbours in other workgroups and other XCD ranges, the streaming block
    stores and the wave-wide neighbour reads of round 4) against the restatement, SI parameters, 1e-10 of the largest entry.
    The parity tests above stop at 216 cells; the restatement needs ~30 s here."""
    from knpemi.dg import DGProblem
    from knpemi.fem.idealized import make_mesh_3D
    import knpemi_dg_oracle as dg
    mesh, ct, ft = make_mesh_3D(1, "hexahedron")
    dp = DGProblem(mesh, ct, ft, [0, 1], [1])
    assert dp.n_cells == 20736
    o = dg.make_dg_oracle(mesh.x, mesh.cells, mesh.cell_type, dp.cell_sub, dp.mem_facets, dp.mem_tags)
    ions = [dict(name="Na", z=1.0, D=[1.33e-9] * 2), dict(name="K", z=1.0, D=[1.96e-9] * 2), dict(name="Cl", z=-1.0, D=[2.03e-9] * 2)]
    params = dict(dt=1e-4, F=96485.0, psi=96485.0 / (8.314 * 300.0), C_M=0.02)
    ins = (dp.cell_sub > 0)[:, None] * np.ones((1, dp.nv), bool)
    w = np.sin(2e5 * dp.X[:, :, 0]) * np.cos(2e6 * dp.X[:, :, 1])
    c_all = [np.where(ins, i, e) * (1.0 + 1e-2 * w * (1 + k)) for k, (e, i) in enumerate(((100.0, 12.0), (4.0, 125.0), (104.0, 137.0)))]
    phi = np.where(ins, -0.0744, 0.0) + 1e-3 * w
    rng = np.random.default_rng(7)
    phi_M = -0.0744 + 1e-3 * rng.standard_normal((dp.nmf, dp.nf))
    I_ch = [1e-2 * rng.standard_normal((dp.nmf, dp.nf)) for _ in range(3)]
    _push(dp, params, ions, c_all, phi, phi_M, I_ch)
    dp.assemble_emi(True)
    dp.assemble_knp(True)
    A, b = o.assemble_emi(params, ions, c_all, phi_M, I_ch, splitting_scheme=True)
    assert csr_rel_err(dp.matrix(0), A) < TOL and rel_err(dp.rhs(0), b) < TOL
    As, bs = o.assemble_knp(params, ions, c_all, phi, phi_M, I_ch, splitting_scheme=True)
    for k in range(2):
        assert csr_rel_err(dp.matrix(1 + k), As[k]) < TOL and rel_err(dp.rhs(1 + k), bs[k]) < TOL, k


def test_dg_q1_full_size_properties(hip_lib, monkeypatch):
    """config 2h (165 888 hexahedra, the reference's own 3-D cell type at r = 2) through what needs no oracle, as
    test_dg_full_size_properties does for tetrahedra: symmetric potential matrix with the constants in its kernel, a
    compatible right-hand side, every column of A_k - M / dt summing to zero, sorted rows of one 8-wide block per cell and
    neighbour, a second assembly reproducing every bit -- and the general kernels (KNPEMI_DG_HEX_GENERAL=1) agreeing with
    the box-mesh kernels to rounding on the whole mesh."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dg_time
    dp = dg_time.build(2, cell="hexahedron")
    n = dp.n
    assert dp.n_cells == 165888 and dp.nv == 8
    dp.assemble_emi()
    dp.assemble_knp()
    A = dp.matrix(0)
    scale = abs(A.data).max()
    assert abs(A - A.T).max() < 1e-12 * scale
    assert np.abs(A @ np.ones(n)).max() < 1e-10 * scale
    b = dp.rhs(0)
    assert abs(b.sum()) < 1e-9 * np.abs(b).sum()
    assert np.all(np.diff(dp.indptr) % 8 == 0) and int(np.diff(dp.indptr).max()) == 8 * 7
    for i in np.random.default_rng(0).integers(0, n, 2000):
        cols = dp.indices[dp.indptr[i]:dp.indptr[i + 1]]
        assert np.all(np.diff(cols) > 0) and i in cols
    X = dp.X                                                   # box cells: volume = product of the three edges at vertex 0
    vol = np.prod([np.linalg.norm(X[:, 1 << t] - X[:, 0], axis=1) for t in range(3)], axis=0)
    mass_col = np.repeat(vol / 8.0, 8) / 1e-4
    datas, rhs = [A.data.copy()], [b.copy()]
    for k in range(2):
        Ak = dp.matrix(1 + k)
        col = np.asarray(Ak.sum(axis=0)).ravel()
        assert np.abs(col - mass_col).max() < 1e-9 * abs(Ak.data).max(), k
        datas.append(Ak.data.copy())
        rhs.append(dp.rhs(1 + k).copy())
        del Ak
    dp.assemble_emi()
    dp.assemble_knp()
    for w in range(3):
        assert np.array_equal(dp.matrix(w).data, datas[w]), w
    del dp, A
    monkeypatch.setenv("KNPEMI_DG_HEX_GENERAL", "1")
    dg2 = dg_time.build(2, cell="hexahedron")
    dg2.assemble_emi()
    dg2.assemble_knp()
    for w in range(3):
        d = dg2.matrix(w).data
        assert np.abs(d - datas[w]).max() < 1e-11 * np.abs(datas[w]).max(), w
        assert np.abs(dg2.rhs(w) - rhs[w]).max() < 1e-11 * np.abs(rhs[w]).max(), w
