"""Oracle time loop (TEST INFRASTRUCTURE): restates the driver loop of the reference
(`examples/idealized_geometries/run_2D.py:341-372`: solve_odes -> EMI solve -> KNP solve ->
update_pde_variables) on top of knpemi_oracle, with direct sparse solves.  Used by tests/ to check
whole trajectories produced through the knpemi API on the GPU."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import knpemi_oracle as o


def solve_singular(A, b):
    """Pure-Neumann EMI system: constant null space (pdeSolver.py:74-78).  Solve the bordered system
    [[A, e], [e^T, 0]] so that the solution is orthogonal to constants."""
    n = A.shape[0]
    e = np.full((n, 1), 1.0 / np.sqrt(n))
    K = sp.bmat([[A, sp.csr_matrix(e)], [sp.csr_matrix(e.T), None]], format="csc")
    return spla.splu(K).solve(np.concatenate([b, [0.0]]))[:n]


class OracleRun:
    def __init__(self, P, params, ions, model, c_all, states, parameters, dof_x, stim_mask, stimulus, rho):
        self.P, self.params, self.ions, self.model = P, params, ions, model
        self.c_all = c_all                       # {tag: [c_prev_0, c_prev_1, c_elim]}
        self.states, self.parameters = states, parameters
        self.stim_mask, self.stimulus, self.rho = stim_mask, stimulus, rho
        self.phi = {t: np.zeros(P.N[t]) for t in P.tags}
        self.phiM = {t: np.zeros(P.NQ[t]) for t in P.tags[1:]}
        self.I_ch = {t: {n: parameters[:, o.MODELS[model]["pidx"][f"I_ch_{n}"]].copy() for n in ("K", "Cl", "Na")}
                     for t in P.tags[1:]}
        self.time = 0.0
        self.k = 0

    def step(self):
        P, prm, ions = self.P, self.params, self.ions
        ix = o.MODELS[self.model]["pidx"]
        vi = o.MODELS[self.model]["V"]
        dt = prm["dt"]
        for tag in P.tags[1:]:
            # update_ode_variables (utils.py:210-235)
            for name, kk in (("K", 0), ("Cl", 1), ("Na", 2)):
                te, ti = P.trace(tag, self.c_all[0][kk], self.c_all[tag][kk])
                self.parameters[:, ix[f"{name}_e"]] = te
                self.parameters[:, ix[f"{name}_i"]] = ti
            if self.k > 0:
                self.states[:, vi] = self.phiM[tag]
            o.ode_sweep(self.model, self.states, self.parameters, self.time, dt, self.stim_mask, self.stimulus)
            self.phiM[tag][:] = self.states[:, vi]
            for n in ("K", "Cl", "Na"):
                self.I_ch[tag][n] = self.parameters[:, ix[f"I_ch_{n}"]].copy()
        mm = {t: [dict(tag=1, I_ch_k=self.I_ch[t])] for t in P.tags[1:]}
        A, _, b = o.assemble_emi(P, prm, ions, self.c_all, self.phiM, mm)
        x = solve_singular(A, b)
        for t in P.tags:
            self.phi[t] = x[P.off[t]:P.off[t] + P.N[t]].copy()
        Ak, bk = o.assemble_knp(P, prm, ions, self.c_all, self.phi, self.phiM, mm, dt)
        xk = spla.splu(Ak.tocsc()).solve(bk)
        boff, _ = o.knp_block_offsets(P, 2)
        c_new = {t: [xk[boff[(t, k)]:boff[(t, k)] + P.N[t]].copy() for k in range(2)] for t in P.tags}
        o.update_pde_variables(P, ions, self.rho, c_new, self.c_all, self.phi, self.phiM)
        self.time += dt
        self.k += 1
