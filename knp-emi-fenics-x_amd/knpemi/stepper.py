"""Device-resident time stepping of the hot path.

The reference API keeps every field in host numpy arrays (`Function.x.array`), so its drop-in
(`pdeSolver.LinearProblem.solve`, `MembraneModel.step_lsoda`) mirrors data across PCIe at every
call.  `DeviceStepper` runs the same sequence of one time step of `run_3D.py:345-368`

    solve_odes  ->  assemble EMI (A, P, b)  ->  [solve]  ->  assemble KNP (A, b)  ->  [solve]
                ->  update_pde_variables

with all fields, tables and operators resident in HBM, calling the C ABI directly.  The linear
solves are not part of the hot path; a caller plugs them in through
`solve_emi` / `solve_knp` callbacks that receive the device problem (device CSR pointers are
available from `knpemi_device_csr`).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class DeviceStepper:
    def __init__(self, forms_emi, forms_knp, c, c_prev, phi, phi_M_prev, solve_emi=None, solve_knp=None,
                 assemble_knp_twice=False, overlap=True, device_solves=None, extrapolate_guess=True,
                 fuse_update=None, fuse_membrane=False, early_membrane=False, knp_method="gmres"):
        a = forms_emi[0]
        self.dp = a.dp
        self.a = a
        self.lib = self.dp.lib
        self.c, self.c_prev, self.phi, self.phi_M_prev = c, c_prev, phi, phi_M_prev
        if device_solves is not None:
            # (rtol_emi, rtol_knp): Krylov solves on the device between the assemblies (knpemi_solve_emi/knp)
            # extrapolate_guess: start each solve from 2 x_n - x_(n-1) instead of x_n (knpemi_extrapolate_guess)
            rtol_emi, rtol_knp = device_solves
            self.iterations = []
            # the concentration solve: "gmres" = the reference's options as PETSc runs them (pdeSolver.py:99-110),
            # "bicgstab" = the faster device path (knpemi.pdeSolver.set_knp_solver_options)
            from .pdeSolver import set_emi_solver_options, set_knp_solver_options
            set_knp_solver_options(a.dp, knp_method)
            set_emi_solver_options(a.dp, "preconditioned" if knp_method == "gmres" else "true")

            def _solve(dp, which, name, rtol, atol):
                if extrapolate_guess:
                    L.check(dp.lib.knpemi_extrapolate_guess(dp.h, which))
                self.iterations.append((name,) + dp.solve(which, rtol, atol))
            solve_emi = lambda dp: _solve(dp, L.B_EMI, "emi", rtol_emi, 1e-40)
            solve_knp = lambda dp: _solve(dp, L.B_KNP, "knp", rtol_knp, 2e-40)
        self.solve_emi, self.solve_knp = solve_emi, solve_knp
        # fuse_update: the write-back kernel of the device KNP solve (or of a device-side set_solution) also performs
        # update_pde_variables, which follows the solve directly in the reference's loop (run_3D.py:356,362): one
        # launch fewer per step.  Default: on with the device solves, off with caller-supplied callbacks.
        self.fuse_update = bool(device_solves is not None) if fuse_update is None else bool(fuse_update)
        self.assemble_knp_twice = assemble_knp_twice
        self.overlap = overlap
        # Which of the two overlapped kernels runs on the auxiliary stream: the one that finishes first, so that the
        # kernels after the join follow the longer one on the same stream without a cross-stream signal (~15 us).
        # Decided from their measured durations at the first step (None = not yet known).
        self.ode_on_aux = None
        import os
        self.overlap_threshold_ms = float(os.environ.get("KNPEMI_OVERLAP_THRESHOLD_MS", "0.025"))     # (0: always side by side)
        self.k = 0
        self.models = []   # MembraneModel objects, in registration order
        self._model_setup = []   # (MembraneModel, stimulus, locator, initial time) for reset()
        dp = self.dp
        dp.set_params(a.physical_params, a.ion_list, a.dt)
        L.check(self.lib.knpemi_set_option(dp.h, L.OPT_FUSE_UPDATE, 1 if self.fuse_update else 0))
        # membrane-facet integrals of b_knp inside the KNP row kernel (default) or as a launch of their own
        self.fuse_membrane = bool(fuse_membrane)
        # membrane-facet integrals of b_knp prepared beside the EMI solve (knpemi_assemble_knp_membrane_early): takes the
        # facet kernel off the chain between the two solves but lengthens the membrane rows of the KNP row kernel by
        # more than it saves (config 2: 0.187 -> 0.193 ms per step), hence off by default
        self.early_membrane = bool(early_membrane) and not self.fuse_membrane
        L.check(self.lib.knpemi_set_option(dp.h, L.OPT_FUSE_MEMBRANE, 1 if self.fuse_membrane else 0))
        self.dt = float(a.dt)
        self.flags_emi = L.WANT_P | (0 if a.splitting_scheme else L.NO_SPLITTING)
        self.flags_knp = 0 if a.splitting_scheme else L.NO_SPLITTING
        self.upload()

    # -- host <-> device ------------------------------------------------------------------
    def upload(self):
        """Push every Function and ODE table to the device (start of a run)."""
        dp, a = self.dp, self.a
        n_solved = len(a.ion_list) - 1
        for tag, sd in a.subdomain_list.items():
            s = dp.sub_index[tag]
            dp.push(L.F_PHI, s, 0, self.phi[tag])
            for k in range(n_solved):
                dp.push(L.F_C_PREV, s, k, self.c_prev[tag][k])
                dp.push(L.F_C, s, k, self.c[tag][k])
            dp.push(L.F_C_ELIM, s, 0, a.ion_list[-1][f'c_{tag}'])
            if tag > 0:
                dp.push(L.F_PHI_M, s, 0, self.phi_M_prev[tag])
                for j, mm in enumerate(sd.get('mem_models', [])):
                    for k, ion in enumerate(a.ion_list):
                        dp.push(L.F_I_CH, s, j * L.MAX_IONS + k, mm['I_ch_k'][ion['name']])

    def set_source(self, ion_index, values):
        """Nodal ECS source term of solved ion `ion_index` (`ion['f_source']`, knpWeakForm.py:164-166)."""
        self.dp.push_array(L.F_SOURCE, 0, int(ion_index), np.ascontiguousarray(values, np.float64))

    def add_membrane_model(self, ode_model, stimulus=None, stimulus_locator=None):
        """Register a bound MembraneModel: uploads its tables and stimulus once."""
        dp = self.dp
        stimulus = stimulus or {}
        if stimulus_locator is None:
            mask = np.ones(ode_model.nodes, np.uint8)
        else:
            mask = np.fromiter(map(stimulus_locator, ode_model.dof_locations), dtype=bool).astype(np.uint8)
        sidx = np.array([ode_model.ode.parameter_indices(k) for k in stimulus], np.int32)
        sval = np.array([float(v) for v in stimulus.values()], np.float64)
        L.check(self.lib.knpemi_ode_set_stimulus(
            dp.h, ode_model._sub, ode_model._model, mask.ctypes.data_as(L.c_u8_p), len(sidx),
            L.iptr(sidx) if len(sidx) else None, L.dptr(sval) if len(sval) else None))
        st = np.ascontiguousarray(ode_model.states)
        pa = np.ascontiguousarray(ode_model.parameters)
        L.check(self.lib.knpemi_ode_set_tables(dp.h, ode_model._sub, ode_model._model, L.dptr(st), L.dptr(pa)))
        if ode_model not in self.models:
            self.models.append(ode_model)
            self._model_setup.append((ode_model, dict(stimulus), stimulus_locator, float(ode_model.time)))

    def reset(self):
        """Back to the state of the host objects (fields, ODE tables, times): a second run from the same start."""
        self.dp._uploaded.clear()      # the device copies have moved on although the host versions have not
        self.upload()
        for m, stim, loc, t0 in self._model_setup:
            m.time = t0
            self.add_membrane_model(m, stim, loc)
        self.k = 0
        self.ode_failures()            # clears the counters of the previous run

    def check_ode_failures(self):
        """`assert success` of odeSolver.py:121 for the device-resident loop: raises KnpemiError(EODE) when LSODA
        failed on any membrane dof since the last check (the counters live on the device; this synchronises)."""
        n = self.ode_failures()
        if n:
            raise L.KnpemiError(L.EODE, f"LSODA failed on {n} membrane dof(s) (odeSolver.py:121 `assert success`)")

    def download(self):
        """Pull fields and ODE tables back into the host objects (end of a run / output).  A dof on which LSODA
        failed keeps a wrong V / I_ch that has fed the PDEs: the download refuses to hand such results out."""
        self.check_ode_failures()
        dp, a = self.dp, self.a
        n_solved = len(a.ion_list) - 1
        for tag, sd in a.subdomain_list.items():
            s = dp.sub_index[tag]
            dp.pull(L.F_PHI, s, 0, self.phi[tag])
            for k in range(n_solved):
                dp.pull(L.F_C_PREV, s, k, self.c_prev[tag][k])
                dp.pull(L.F_C, s, k, self.c[tag][k])
            dp.pull(L.F_C_ELIM, s, 0, a.ion_list[-1][f'c_{tag}'])
            if tag > 0:
                dp.pull(L.F_PHI_M, s, 0, self.phi_M_prev[tag])
                for j, mm in enumerate(sd.get('mem_models', [])):
                    for k, ion in enumerate(a.ion_list):
                        dp.pull(L.F_I_CH, s, j * L.MAX_IONS + k, mm['I_ch_k'][ion['name']])
        for m in self.models:
            st = np.ascontiguousarray(m.states)
            pa = np.ascontiguousarray(m.parameters)
            L.check(self.lib.knpemi_ode_get_tables(dp.h, m._sub, m._model, L.dptr(st), L.dptr(pa)))
            m.states[...] = st
            m.parameters[...] = pa

    # -- one time step, everything enqueued on the handle's stream ---------------------------
    def step(self, halo=None):
        dp, lib = self.dp, self.lib
        if halo is not None and (self.solve_emi is not None or self.solve_knp is not None) \
                and not getattr(halo, "supports_solves", False):
            raise NotImplementedError(
                "this halo does not distribute the linear solves: knpemi_solve_emi/knp on a partitioned problem "
                "need the halo'd SpMV and the all-reduced dot products (knpemi.fem.partition.DistributedSolves)")
        flags = L.ODE_SET_TRACES | (L.ODE_SET_V if self.k > 0 else 0)
        calibrate = self.overlap and self.ode_on_aux is None and self.models
        if calibrate:
            L.check(lib.knpemi_profile(dp.h, (1 << L.KERNEL_NAMES.index("ode_step_kernel"))
                                       | (1 << L.KERNEL_NAMES.index("emi_rows_kernel"))))
        ode_aux = bool(self.overlap and self.ode_on_aux)
        if self.overlap and not ode_aux:
            # the EMI matrix (A, P, volume part of b) does not depend on the ODE output: assemble it on the
            # auxiliary stream while the ODE sweep runs on the main one
            L.check(lib.knpemi_assemble_emi(dp.h, self.flags_emi | L.SKIP_MEMBRANE_RHS | L.ON_AUX_STREAM))
        side_of_sub = {}
        for m in self.models:
            # The sweeps of different cells (sub-domains) are independent: the first cell's models share the main /
            # auxiliary stream pair with the EMI assembly, the other cells' run beside them on the second auxiliary
            # stream.  Several models of ONE cell integrate the same dofs and overwrite phi_M_prev in turn
            # (odeSolver.py:31-38, benchmark/run_stim_duration.py:163-166): they stay in order on one stream.
            if m._sub not in side_of_sub:
                side_of_sub[m._sub] = L.ODE_ON_AUX2 if side_of_sub else (L.ODE_ON_AUX if ode_aux else 0)
        # side streams first: a side launch is ordered after what the main stream holds at that moment, so it must be
        # enqueued before this step's main-stream kernels to run beside them
        for wanted in (L.ODE_ON_AUX2, L.ODE_ON_AUX, 0):
            for m in self.models:
                if side_of_sub[m._sub] != wanted:
                    continue
                L.check(lib.knpemi_ode_step(dp.h, m._sub, m._model, float(m.time), self.dt, m.rtol, m.atol,
                                            flags | wanted, L.iptr(m._ion_param), int(m.V_index)))
                m.time = m.time + self.dt
        if ode_aux:     # the assembly is the longer kernel here: it stays on the main stream
            L.check(lib.knpemi_assemble_emi(dp.h, self.flags_emi | L.SKIP_MEMBRANE_RHS))
        # Partitioned runs: the membrane dofs of the ghost cell layer are integrated redundantly on both ranks
        # (same inputs after the bulk halo, deterministic LSODA => identical bits, tools/check_partition_steps.py),
        # so phi_M / I_ch need no exchange of their own.
        if len(side_of_sub) > 1 and not self.overlap:
            L.check(lib.knpemi_join(dp.h))      # the sweeps on the second auxiliary stream
        if self.overlap:
            L.check(lib.knpemi_join(dp.h))
            L.check(lib.knpemi_assemble_emi_membrane_rhs(dp.h, self.flags_emi))
            if calibrate:
                us = {}
                for name in ("ode_step_kernel", "emi_rows_kernel"):
                    n, ms = C.c_int64(), C.c_double()
                    L.check(lib.knpemi_profile_read(dp.h, L.KERNEL_NAMES.index(name), C.byref(n), C.byref(ms)))
                    us[name] = ms.value
                L.check(lib.knpemi_profile(dp.h, 0))
                self.ode_on_aux = us["emi_rows_kernel"] > us["ode_step_kernel"]
                # Running the two side by side costs a cross-stream join and a separate launch for the membrane
                # Robin term (~25 us together on this stack): not worth it when the shorter kernel is shorter than that
                if min(us.values()) < self.overlap_threshold_ms:
                    self.overlap = False
        else:
            L.check(lib.knpemi_assemble_emi(dp.h, self.flags_emi))
        knp_flags = self.flags_knp
        if self.early_membrane:
            # everything of the membrane integrals that does not need the new potential: now, beside the EMI solve
            L.check(lib.knpemi_assemble_knp_membrane_early(dp.h, self.flags_knp | (L.ON_AUX_STREAM if self.overlap else 0)))
            knp_flags |= L.MEMBRANE_EARLY
        if self.solve_emi is not None:
            self.solve_emi(dp)
            if halo is not None:
                halo.exchange_bulk()
        L.check(lib.knpemi_assemble_knp(dp.h, knp_flags))
        if self.assemble_knp_twice:   # the reference assembles p = a a second time (knpWeakForm.py:319)
            L.check(lib.knpemi_assemble_knp(dp.h, knp_flags))
        if self.solve_knp is not None:
            self.solve_knp(dp)
        if not (self.fuse_update and self.solve_knp is not None):
            L.check(lib.knpemi_update_pde(dp.h))
        if halo is not None:
            halo.exchange_bulk()
        self.k += 1

    def ode_failures(self):
        n = 0
        for m in self.models:
            nf = C.c_int32()
            nr, ns = C.c_int64(), C.c_int64()
            self.lib.knpemi_ode_stats(self.dp.h, m._sub, m._model, C.byref(nr), C.byref(ns), C.byref(nf))
            n += nf.value
        return n
