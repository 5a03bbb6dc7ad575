"""The 2D idealized run (`run_2D.py:137-372` of the reference: one cell in the ECS, K / Cl / Na, Hodgkin-Huxley membrane,
synaptic stimulus on x < 20 um) with the DG(P1) + interior-penalty variant (`knpemi.dg.DGProblem`):
membrane ODEs at the facet nodes, both assemblies, both linear solves (CG / BiCGStab + auxiliary-space AMG,
`knpemi_dg_solve_emi/knp`) and the end-of-step update on the GPU; `--host-solves` solves the two systems with SciPy
from the CSR the kernels fill instead.

    python run_2D_dg.py [--resolution 1] [--steps 100] [--host-solves]
"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "knp-emi-fenics-x_amd"))
sys.path.insert(0, HERE)

import mm_hh                                                       # noqa: E402
from setup_problem import (C_M, CL_E, CL_I, D_CL, D_K, D_NA, DT, FARADAY, K_E, K_I, NA_E, NA_I, PSI)  # noqa: E402
from knpemi import _lib as L                                       # noqa: E402
from knpemi.dg import DGProblem                                    # noqa: E402
from knpemi.fem.idealized import make_mesh_2D                      # noqa: E402


def solve_singular(A, b):
    """Pure Neumann potential system: bordered solve orthogonal to the constants (pdeSolver.py:74-78)."""
    n = A.shape[0]
    e = np.full((n, 1), 1.0 / np.sqrt(n))
    K = sp.bmat([[A, sp.csr_matrix(e)], [sp.csr_matrix(e.T), None]], format="csc")
    return spla.splu(K).solve(np.concatenate([b, [0.0]]))[:n]


class DGRun:
    def __init__(self, resolution=1, g_syn=10.0, dt=DT, device_solves=True, rtol=(1e-5, 1e-7)):
        self.device_solves, self.rtol, self.iterations = device_solves, rtol, []
        mesh, ct, ft = make_mesh_2D(resolution)
        self.dp = dp = DGProblem(mesh, ct, ft, [0, 1], [1])
        self.ions = [dict(name="K", z=1.0, D=[D_K] * 2), dict(name="Cl", z=-1.0, D=[D_CL] * 2), dict(name="Na", z=1.0, D=[D_NA] * 2)]
        dp.set_params(dict(dt=dt, F=FARADAY, psi=PSI, C_M=C_M), self.ions)
        ins = (dp.cell_sub > 0)[:, None] * np.ones((1, dp.nv), bool)
        for k, (e, i) in enumerate(((K_E, K_I), (CL_E, CL_I), (NA_E, NA_I))):
            dp.set_concentration(k, np.where(ins, i, e))
        nq = dp.nmf * dp.nf
        ode = mm_hh
        states = np.tile(np.asarray(ode.init_state_values(), float), (nq, 1))
        params = np.tile(np.asarray(ode.init_parameter_values(), float), (nq, 1))
        pi = ode.parameter_indices
        params[:, pi("Cm")] = C_M
        params[:, pi("psi")] = PSI
        for n, z in (("Na", 1.0), ("K", 1.0), ("Cl", -1.0)):
            params[:, pi("z_" + n)] = z
        self.stimulated = dp.XM.reshape(nq, 2)[:, 0] < 20e-6       # stimulus_locator of run_2D.py:268-270
        params[self.stimulated, pi("stim_amplitude")] = g_syn
        ion_param = sum(([pi(f"{i['name']}_e"), pi(f"{i['name']}_i"), pi(f"I_ch_{i['name']}")] for i in self.ions), [])
        self.v_index = ode.state_indices("V")
        dp.ode_bind(L.MODEL_HH_SI, states, params, ion_param, self.v_index)
        if device_solves:
            dp.set_extrapolation(True)
        self.dt, self.time, self.k = dt, 0.0, 0

    def step(self):
        dp = self.dp
        dp.ode_step(self.time, self.dt, set_v=self.k > 0)           # traces -> LSODA -> phi_M, I_ch
        dp.assemble_emi()
        if self.device_solves:                                       # pdeSolver.py:24-35,99-110 (rtol 1e-5 / 1e-7)
            it_emi = dp.solve_emi(rtol=self.rtol[0])[0]
            dp.assemble_knp()
            it_knp = dp.solve_knp(rtol=self.rtol[1], update=True)[0]
            self.iterations.append((it_emi, it_knp))
        else:
            phi = solve_singular(dp.matrix(0), dp.rhs(0))
            dp.set_potential(phi)
            dp.assemble_knp()
            c_new = np.stack([spla.splu(dp.matrix(1 + k).tocsc()).solve(dp.rhs(1 + k)) for k in range(2)])
            dp.update(c_new)
        n_rhs, n_steps, n_failed = dp.ode_stats()
        assert n_failed == 0                                         # odeSolver.py:121
        self.time += self.dt
        self.k += 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--resolution", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--host-solves", action="store_true")
    a = ap.parse_args()
    run = DGRun(a.resolution, device_solves=not a.host_solves)
    for k in range(a.steps):
        run.step()
        if (k + 1) % 10 == 0:
            v = run.dp.get_membrane_potential()
            its = f"   iterations {run.iterations[-1]}" if run.iterations else ""
            print(f"t = {run.time * 1e3:5.1f} ms   phi_M: mean {v.mean() * 1e3:8.3f} mV, max {v.max() * 1e3:8.3f} mV{its}")
