"""The C++ CPU port (oracle/knpemi_cpu.cpp, the timed CPU baseline) agrees with the numpy oracle."""
import contextlib
import io

import numpy as np
import pytest
import scipy.sparse as sp

import cpu_port
import knpemi_oracle as o
from helpers import Setup, rel_err


@pytest.mark.parametrize("kind,r", [("2d", 1), ("tet", 0), ("hex", 0)])
@pytest.mark.parametrize("splitting", [True, False])
def test_cpu_port_matches_numpy_oracle(kind, r, splitting):
    with contextlib.redirect_stdout(io.StringIO()):
        s = Setup(kind, r, build_forms=False)
    s.perturb()
    _, P, params, ions = s.oracle()
    c_all, phi, phiM, mm = s.oracle_fields()
    A, Pm, b = o.assemble_emi(P, params, ions, c_all, phiM, mm, splitting_scheme=splitting)
    Ak, bk = o.assemble_knp(P, params, ions, c_all, phi, phiM, mm, s.dt, splitting_scheme=splitting)
    port = cpu_port.CpuPort(P, params, ions, A, Ak)
    Ich = np.stack([mm[1][0]["I_ch_k"][n] for n in ("K", "Cl", "Na")])
    a, p, bb = port.assemble_emi(c_all, phiM, Ich, splitting)
    ak, bbk = port.assemble_knp(c_all, phi, phiM, Ich, splitting)
    As = A.tocsr(); As.sort_indices()
    Ps = (Pm + 0 * A).tocsr(); Ps.sort_indices()
    Aks = Ak.tocsr(); Aks.sort_indices()
    assert rel_err(a, As.data) < 1e-12 and rel_err(bb, b) < 1e-12
    Pp = sp.csr_matrix((p, port.ci, port.rp), shape=A.shape)
    assert abs(Pp - Pm).max() < 1e-12 * abs(Pm).max()
    assert rel_err(ak, Aks.data) < 1e-12 and rel_err(bbk, bk) < 1e-12


def test_cpu_port_ode_sweep_matches_scipy():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ode_models.npz"))
    st = np.tile(g["hh_si_stim10_y0"], (5, 1))
    pa = np.tile(g["hh_si_stim10_p0"], (5, 1))
    port_lib = cpu_port.lib()
    port = cpu_port.CpuPort.__new__(cpu_port.CpuPort)
    for k in range(3):
        failed, nrhs = cpu_port.CpuPort.ode_sweep(port, 0, st, pa, k * 1e-4, 1e-4, np.ones(5), [8], [10.0])
        assert failed == 0 and nrhs > 0
        assert rel_err(st[0], g["hh_si_stim10_traj"][k][:4]) < 1e-9
