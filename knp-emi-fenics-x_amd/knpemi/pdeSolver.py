"""Solver factories for the two sub-problems (drop-in for `src/knpemi/pdeSolver.py`).

`create_solver_emi/knp` return an object with the surface the drivers use of
`dolfinx.fem.petsc.LinearProblem`: `.solve()` (assembles, solves, updates the
Functions passed as unknowns in place and returns them), `.solver.getIterationNumber()`,
`.A`, `.u` (`run_3D.py:355-360`, `tests/run_mms_emi.py:321-323`).

`.solve()` = GPU assembly (the hot path: `knpemi_assemble_emi/knp`) followed by a
linear solve (adjacent to the hot path):

* `direct=False` (the reference's CG / GMRES + hypre options, `pdeSolver.py:24-35,99-110`):
  device-resident Krylov solve on the assembled CSR (`knpemi_solve_emi`: PCG with the
  constant null space projected out, convergence on the true residual; `knpemi_solve_knp`: GMRES(30) with left
  preconditioning and the preconditioned-norm test as PETSc runs `ksp_type gmres`, `ksp_min_it = 5`; smoothed-aggregation
  AMG for both), same `rtol` / `atol` / `ksp_max_it = 1000`, non-zero initial guess;
* `direct=True` (MUMPS LU, `pdeSolver.py:15-21`) and systems with Dirichlet conditions
  (MMS): sparse LU on the host with SciPy as a stand-in.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from . import _lib as L
from .fem.function import Function, as_float


# The reference's iterative options for the concentration solve (pdeSolver.py:99-110): `ksp_type gmres`, `ksp_min_it 5`.
# The drop-in runs them as PETSc would (KNPEMI_OPT_KNP_METHOD = 1: GMRES(30), left preconditioning, preconditioned-norm
# test; KNPEMI_OPT_KNP_MIN_IT = 5 GMRES iterations), so `solver.getIterationNumber()` counts what the reference's does.
# The faster device path (bench.py `with_solves`, DeviceStepper(knp_method="bicgstab")) is BiCGStab on the true residual,
# whose iteration applies operator and preconditioner twice: three of them are the fewest that do at least the work of
# five GMRES iterations.
KSP_MIN_IT_KNP = 5
KNP_MIN_BICGSTAB_ITERATIONS = (KSP_MIN_IT_KNP + 1) // 2


def set_knp_solver_options(dp, method="gmres", min_it=None):
    """Select the concentration solve of a device problem: "gmres" (the reference's options) or "bicgstab"; `min_it` defaults
    to the reference's ksp_min_it in the units of the method."""
    gm = method == "gmres"
    if not gm and method != "bicgstab":
        raise ValueError(f"unknown KNP method {method!r}")
    if min_it is None:
        min_it = KSP_MIN_IT_KNP if gm else KNP_MIN_BICGSTAB_ITERATIONS
    L.check(dp.lib.knpemi_set_option(dp.h, L.OPT_KNP_METHOD, 1 if gm else 0))
    L.check(dp.lib.knpemi_set_option(dp.h, L.OPT_KNP_MIN_IT, int(min_it)))


def set_emi_solver_options(dp, norm="preconditioned"):
    """Convergence test of the potential solve's CG: "preconditioned" = what PETSc's KSPCG defaults make of the reference's
    options (pdeSolver.py:60-72: |M^-1 r| against |M^-1 b|), "true" = the true residual (the faster device path's choice:
    its norm comes for free with the residual update)."""
    if norm not in ("preconditioned", "true"):
        raise ValueError(f"unknown norm {norm!r}")
    L.check(dp.lib.knpemi_set_option(dp.h, L.OPT_EMI_NORM, 1 if norm == "preconditioned" else 0))


class _KSPInfo:
    """`problem.solver` stand-in: only what the drivers query."""

    def __init__(self):
        self.iterations = 0
        self.residual_norm = 0.0

    def getIterationNumber(self):
        return self.iterations


class DirichletBC:
    """`dolfinx.fem.dirichletbc(u_bc, dofs)` for block `block` of the system."""

    def __init__(self, u_bc, dofs, block=0):
        self.g = u_bc
        self.dofs = np.asarray(dofs, np.int64)
        self.block = block


def _push_common(dp, form):
    """Upload the coefficient Functions both systems read (version-tracked)."""
    ion_list, subs = form.ion_list, form.subdomain_list
    n_solved = len(ion_list) - 1
    for tag in subs:
        s = dp.sub_index[tag]
        for idx in range(n_solved):
            dp.push(L.F_C_PREV, s, idx, form.c_prev[tag][idx])
        dp.push(L.F_C_ELIM, s, 0, ion_list[-1][f'c_{tag}'])
        if tag > 0:
            dp.push(L.F_PHI_M, s, 0, form.phi_M_prev[tag])


def _push_currents(dp, form):
    for tag, sd in form.subdomain_list.items():
        if tag == 0:
            continue
        s = dp.sub_index[tag]
        for j, mm in enumerate(sd.get('mem_models', [])):
            for k, ion in enumerate(form.ion_list):
                src = mm['I_ch_k'][ion['name']]
                idx = j * L.MAX_IONS + k
                if isinstance(src, Function):
                    dp.push(L.F_I_CH, s, idx, src)
                else:
                    dp.push_array(L.F_I_CH, s, idx, np.full(int(dp.n_q[s]), as_float(src)))


def _apply_bcs(A, b, bcs, offsets):
    """Row/column elimination of Dirichlet dofs (MMS runs only)."""
    A = A.tocsr(copy=True)
    rows = np.concatenate([bc.dofs + offsets[bc.block] for bc in bcs])
    vals = np.concatenate([bc.g.x._a[bc.dofs] for bc in bcs])
    x_bc = np.zeros(A.shape[0])
    x_bc[rows] = vals
    b = b - A @ x_bc
    keep = np.ones(A.shape[0])
    keep[rows] = 0.0
    Dk = sp.diags(keep)
    A = Dk @ A @ Dk + sp.diags(1.0 - keep)
    b[rows] = vals
    return A.tocsr(), b


def _push_sources(dp, f):
    """ECS source terms of the solved ions (knpWeakForm.py:164-166) as nodal fields."""
    for k, ion in enumerate(f.ion_list[:-1]):
        if 'f_source' in ion:
            src = ion['f_source']
            arr = src.x._a if isinstance(src, Function) else np.full(int(dp.n_vert[0]), as_float(src))
            dp.push_array(L.F_SOURCE, 0, k, arr)


class LinearProblem:
    def __init__(self, system, a, Lf, u, subdomain_list, direct, p, bcs, atol, rtol, threshold, prefix):
        if a.system != system or Lf.system != system:
            raise ValueError(f"forms do not belong to the {system} system")
        self.system, self.a, self.L, self.p = system, a, Lf, p
        self.u = u
        self.dp = a.dp
        self.direct, self.bcs = direct, bcs
        self.atol, self.rtol, self.threshold = atol, rtol, threshold
        self.petsc_options_prefix = prefix
        self.solver = _KSPInfo()
        self.A = None
        self.P = None
        self.b = None
        self.nullspace = None
        self._b_mms = None

    # -- assembly on the GPU ---------------------------------------------------------
    def assemble(self):
        f, dp = self.a, self.dp
        dp.set_params(f.physical_params, f.ion_list, f.dt)
        _push_common(dp, f)
        if self.system == "emi":
            if not f.splitting_scheme:
                _push_currents(dp, f)
            want_p = (self.p is not None) and not self.direct
            dp.assemble_emi(want_p=want_p, splitting_scheme=f.splitting_scheme)
            self.A = dp.csr(L.A_EMI)
            self.P = dp.csr(L.P_EMI) if want_p else None
            self.b = dp.rhs(L.B_EMI)
            if f.mms is not None:   # manufactured sources (host scaffolding, knpemi/mms.py)
                if self._b_mms is None:
                    from .mms import emi_mms_rhs
                    self._b_mms = emi_mms_rhs(f)
                self.b = self.b + self._b_mms
        else:
            for tag in f.subdomain_list:
                dp.push(L.F_PHI, dp.sub_index[tag], 0, f.phi[tag])
            _push_currents(dp, f)
            _push_sources(dp, f)
            dp.assemble_knp(splitting_scheme=f.splitting_scheme)
            self.A = dp.csr(L.A_KNP)
            self.P = self.A
            self.b = dp.rhs(L.B_KNP)
        return self.A, self.b

    # -- linear solve (host stand-in for PETSc KSP) -----------------------------------
    def _solve_linear(self, A, b, x0):
        n = A.shape[0]
        if self.direct:
            if self.nullspace is not None:
                e = np.full((n, 1), 1.0 / np.sqrt(n))
                K = sp.bmat([[A, sp.csr_matrix(e)], [sp.csr_matrix(e.T), None]], format="csc")
                x = spla.splu(K).solve(np.concatenate([b, [0.0]]))[:n]
            else:
                x = spla.splu(A.tocsc()).solve(b)
            self.solver.iterations = 1
            return x
        Pm = self.P if self.P is not None else A
        ilu = spla.spilu(Pm.tocsc(), drop_tol=1e-5, fill_factor=20)
        M = spla.LinearOperator(A.shape, ilu.solve)
        count = [0]

        def cb(_):
            count[0] += 1
        if self.system == "emi":
            if self.nullspace is not None:
                b = b - b.mean()
            x, info = spla.cg(A, b, x0=x0, rtol=self.rtol, atol=self.atol, maxiter=1000, M=M, callback=cb)
            if self.nullspace is not None:
                x = x - x.mean()
        else:
            x, info = spla.gmres(A, b, x0=x0, rtol=self.rtol, atol=self.atol, maxiter=1000, M=M,
                                 callback=cb, callback_type="pr_norm")
        if info != 0:
            raise RuntimeError(f"{self.petsc_options_prefix}: Krylov solver did not converge (info={info})")
        self.solver.iterations = count[0]
        return x

    def _solve_on_device(self):
        """direct=False: assemble and solve on the GPU, then mirror the solution into the Functions."""
        f, dp = self.a, self.dp
        dp.set_params(f.physical_params, f.ion_list, f.dt)
        _push_common(dp, f)
        which = L.B_EMI if self.system == "emi" else L.B_KNP
        for fn, (field, sub, idx) in zip(self.u, self._unknown_fields()):
            dp.push(field, sub, idx, fn)                     # initial guess = current unknowns
        if self.system == "emi":
            if not f.splitting_scheme:
                _push_currents(dp, f)
            dp.assemble_emi(want_p=False, splitting_scheme=f.splitting_scheme)
        else:
            for tag in f.subdomain_list:
                dp.push(L.F_PHI, dp.sub_index[tag], 0, f.phi[tag])
            _push_currents(dp, f)
            _push_sources(dp, f)
            dp.assemble_knp(splitting_scheme=f.splitting_scheme)
        if self.system == "knp":
            set_knp_solver_options(dp, "gmres")
        else:
            set_emi_solver_options(dp, "preconditioned")
        its, relres = dp.solve(which, self.rtol, self.atol, maxit=1000)
        self.solver.iterations, self.solver.residual_norm = its, relres
        for fn, (field, sub, idx) in zip(self.u, self._unknown_fields()):
            dp.pull(field, sub, idx, fn)
            fn.x.scatter_forward()
        return self.u

    def _unknown_fields(self):
        dp = self.dp
        if self.system == "emi":
            return [(L.F_PHI, dp.sub_index[tag], 0) for tag in self.a.subdomain_list]
        n_solved = len(self.a.ion_list) - 1
        return [(L.F_C, dp.sub_index[tag], k) for tag in self.a.subdomain_list for k in range(n_solved)]

    def solve(self):
        if not self.direct and not self.bcs and self.a.mms is None:
            return self._solve_on_device()
        A, b = self.assemble()
        sizes = [f.x._a.shape[0] for f in self.u]
        offsets = np.concatenate([[0], np.cumsum(sizes)])
        if self.bcs:
            A, b = _apply_bcs(A, b, self.bcs, offsets)
        x0 = np.concatenate([f.x._a for f in self.u])
        x = self._solve_linear(A, b, x0)
        for f, o, n in zip(self.u, offsets[:-1], sizes):
            f.x.array[:] = x[o:o + n]
            f.x.scatter_forward()
        self.dp.set_solution(L.B_EMI if self.system == "emi" else L.B_KNP, x)
        return self.u


def create_solver_emi(a, L, phi, entity_maps, subdomain_list, comm,
                      direct=True, p=None, bcs=None, atol=1E-40, rtol=1E-5, threshold=None):
    """ EMI solver: direct (LU) or CG preconditioned with p (pdeSolver.py:8-80) """
    u = [phi[tag] for tag in subdomain_list]                     # [phi_e, phi_i, ...] (:42)
    prefix = "emi_direct_" if direct else "emi_iterative_"
    problem = LinearProblem("emi", a, L, u, subdomain_list, direct, p, bcs, atol, rtol, threshold, prefix)
    if bcs is None:
        problem.nullspace = "constant"                            # pure Neumann problem (:74-78)
    return problem


def create_solver_knp(a, L, c, entity_maps, subdomain_list, comm,
                      direct=True, p=None, bcs=None, atol=1E-40, rtol=1E-5, threshold=None):
    """ KNP solver: direct (LU) or GMRES preconditioned with p = a (pdeSolver.py:83-141).

    `bcs` is accepted and dropped, exactly as in the reference: its `create_solver_knp` takes the argument (:84) and passes
    it to neither `LinearProblem` call (:119-139) -- the concentration systems are pure Neumann problems made regular by
    the mass term.  (`create_solver_emi` does honour `bcs`, :56,66, and so does this module.) """
    u = [val for tag in subdomain_list for val in c[tag]]       # [c[0][0], c[0][1], c[1][0], ...] (:117)
    prefix = "knp_direct_" if direct else "knp_iterative_"
    return LinearProblem("knp", a, L, u, subdomain_list, direct, p, None, atol, rtol, threshold, prefix)
