"""Time loop of the DG(P1)+SIP variant on the CPU restatement (TEST INFRASTRUCTURE, like oracle/driver.py for the CG
path): per step the membrane ODEs at the facet nodes (scipy's ODEPACK LSODA with the reference's side-effect currents,
knpemi_oracle.ode_sweep), the potential system, the concentration systems, the end-of-step update -- the sequence of
examples/idealized_geometries/run_2D.py:341-372 with the DG discretisation in place of the CG one."""
import numpy as np
import scipy.sparse.linalg as spla

import knpemi_oracle as o
from driver import solve_singular


class DGOracleRun:
    def __init__(self, D, params, ions, model, c_all, states, parameters, stim_mask=None, stimulus=None, rho=None, gamma=10.0):
        self.D, self.params, self.ions, self.model = D, params, ions, model
        self.c_all = [np.array(c, float) for c in c_all]          # K fields (nc, nv), eliminated ion last
        self.states, self.parameters = states, parameters         # one row per membrane node (facet-major)
        self.stim_mask, self.stimulus, self.rho, self.gamma = stim_mask, stimulus, rho, gamma
        self.phi = np.zeros((D.nc, D.nv))
        self.phiM = np.zeros((D.nmf, D.nf))
        ix = o.MODELS[model]["pidx"]
        self.I_ch = [parameters[:, ix[f"I_ch_{ion['name']}"]].reshape(D.nmf, D.nf).copy() for ion in ions]
        self.time, self.k = 0.0, 0

    def step(self):
        D, prm, ions = self.D, self.params, self.ions
        ix, vi = o.MODELS[self.model]["pidx"], o.MODELS[self.model]["V"]
        dt = prm["dt"]
        for k, ion in enumerate(ions):
            te, ti = D.traces(self.c_all[k])
            self.parameters[:, ix[f"{ion['name']}_e"]] = te.ravel()
            self.parameters[:, ix[f"{ion['name']}_i"]] = ti.ravel()
        if self.k > 0:
            self.states[:, vi] = self.phiM.ravel()
        o.ode_sweep(self.model, self.states, self.parameters, self.time, dt, self.stim_mask, self.stimulus)
        self.phiM = self.states[:, vi].reshape(D.nmf, D.nf).copy()
        self.I_ch = [self.parameters[:, ix[f"I_ch_{ion['name']}"]].reshape(D.nmf, D.nf).copy() for ion in ions]
        A, b = D.assemble_emi(prm, ions, self.c_all, self.phiM, self.I_ch, gamma=self.gamma)
        self.phi = solve_singular(A, b).reshape(D.nc, D.nv)
        As, bs = D.assemble_knp(prm, ions, self.c_all, self.phi, self.phiM, self.I_ch, gamma=self.gamma)
        c_new = [spla.splu(As[k].tocsc()).solve(bs[k]).reshape(D.nc, D.nv) for k in range(len(ions) - 1)]
        self.c_all, self.phiM = D.update(ions, self.rho, c_new, self.phi)
        self.time += dt
        self.k += 1
