#!/usr/bin/env python3
"""bench.py -- assembled dofs/s of the KNP-EMI per-time-step hot path on MI355X.

One "step" = one pass of the hot path over one field state, device-resident (SURVEY.md section 8d /
BASELINE.md section 2): fused ODE launch (trace refresh + LSODA sweep + copy-back), EMI assembly (A, P, b in one
pass), KNP assembly (A once, b incl. the membrane kernel), end-of-step update.  Krylov solves and file output are
excluded from the timed region -- but the state the timed steps run on is a REAL trajectory: an untimed pass first
integrates the same problem with the device Krylov solves from t = 0 and records the solution of every step; the timed
steps replay it, pasting the recorded solution where the solve would write it (one launch per system, as the solve's
own write-back).  The membrane therefore depolarises and fires as in a real run and the ODE sweep does the work a real
run gives it.

`value` = dofs of one step / (MEDIAN over --repeats windows of the time of exactly --steps steps), the window being
trajectory steps [--warmup, --warmup + --steps); the device is put back to the start of the window before every repeat.
`spike_window` is the same measurement around the step where the membrane fires (most ODE work), `with_solves` whole
time steps including the device Krylov solves from a fixed trajectory step, `config3_leg` the same measurement on the
995 328-tet mesh (where the row kernels are HBM-bound, not launch-bound), `dg_variant` a short run of the DG variant.
Any failure (a solve that does not converge, a communication hook, LSODA) exits non-zero: there is no fallback state.

At N > 1 the mesh is partitioned into x-slabs (weak: N times longer box; --scaling strong: the fixed config-3 box),
every step exchanges the ghost-dof halo of the bulk fields (stream-ordered RCCL point-to-point; ghost membrane dofs
are integrated redundantly), and the recorded trajectory comes from the distributed Krylov solves
(knpemi_set_distributed: halo'd SpMV, all-reduced dot products, per-GPU AMG).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...        (starts the N ranks itself, as a child `python -m torch.distributed.run`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import contextlib
import ctypes as C
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "knp-emi-fenics-x_amd"), os.path.join(ROOT, "examples", "idealized_geometries"),
          os.path.join(ROOT, "examples", "local_astrocyte_depolarization")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
LAUNCH_FLOOR_US = 5.0      # duration of a trivial streaming kernel in the same event brackets on this stack
PROFILE_STEPS = 5
SPIKE_SEARCH = 72        # trajectory steps recorded at least: contains the action potential of the stimulated cell end
if os.environ.get("KNPEMI_BENCH_TRAJ_MIN"):     # rehearsals of the N > 1 code path on one GPU (gloo, ranks sharing the card)
    SPIKE_SEARCH = max(8, int(os.environ["KNPEMI_BENCH_TRAJ_MIN"]))
WITH_SOLVES_START = 10   # trajectory step the with_solves pass starts from, whatever --steps / --warmup are
AGREEMENT_LIMIT = 1e-6   # cpu_baseline.agreement above this ends the bench with a non-zero exit code

WORKLOADS = {
    # name: (family, mesh kind, resolution factor) -- geometry of make_mesh_3D.py, 6 tets per hexahedron
    "config2": ("idealized", "tet", 1),     # BASELINE.json configs[1]: 124 416 tets, 79 251 dofs/step
    "config2h": ("idealized", "hex", 2),    # reference-faithful Q1 hexahedra, 165 888 cells
    "config3": ("idealized", "tet", 2),     # 995 328 tets on ONE GPU (the 8-GPU mesh of configs[2])
    "r3": ("idealized", "tet", 3),          # 7.96 M tets
    "2d": ("idealized", "2d", 3),
    "tet_r0": ("idealized", "tet", 0),      # small meshes for solver diagnostics (tools/dg_solves.py)
    "hex_r1": ("idealized", "hex", 1),
    # BASELINE.json configs[4] stand-in (the emimesh mesh needs a network fetch): three sub-domains -- ECS, neuron
    # (cells 1, 3: HH, mV / ms units), glia (cells 2, 4: Kir4.1 + pump) -- with the pulsed ECS K+ source of
    # examples/local_astrocyte_depolarization/run_stim_duration.py on the box mesh
    "config5s": ("astro", "tet", 2),        # 995 328 tets, 3 sub-domains, two membrane models
    "config5s_r3": ("astro", "tet", 3),     # 7.96 M tets
}


class Case:
    """What the bench needs from a driver set-up, whichever example builds it."""

    def __init__(self, workload, rank=0, world=1, scaling="weak"):
        family, kind, r = WORKLOADS[workload]
        self.family, self.kind, self.r = family, kind, r
        quiet = io.StringIO()
        with contextlib.redirect_stdout(quiet):
            if family == "astro":
                import run_stim_duration as rsd
                cfg = dict(rsd.DEFAULTS)
                cfg["mesh"] = dict(kind="box3d", resolution_factor=r, cell_type="tetrahedron", length=2)
                # a source box like the reference's (0.8 x 0.8 x 0.4 um there), placed in an ECS corner of this box
                # (cm), on from the first step for 1 ms
                cfg.update(delay=0.0, pulse_width=1.0, period=10.0, end_time=100.0, x_L=15e-4, x_U=17e-4, y_L=-1.0,
                           y_U=0.2e-4, z_L=-1.0, z_U=0.2e-4)
                if world > 1:      # weak scaling: a box `world` times as long, cut into x-slabs of whole cell layers
                    from knpemi.fem.distributed import make_partitioned_astro
                    cfg["mesh"]["length"] = 2 * world if scaling == "weak" else 2
                    s = make_partitioned_astro(cfg, rank, world, method="slab")
                else:
                    s = rsd.Problem(cfg)
                s.set_source(0.0)
                self.models = [(mm["ode"], s.stim_params["stimulus"], s.stim_params["stimulus_locator"])
                               for tag in (1, 2) for mm in s.subdomain_list[tag]["mem_models"]]
                self.solver_rtol = (1e-6, 1e-7)      # run_stim_duration.py:424-437
                self.source = s.f_source_K.x._a
                self.describe = ("HH (mV/ms) on cells 1,3 + glial Kir4.1/pump on cells 2,4, pulsed ECS K+ source on, "
                                 "dt=0.1 ms")
            else:
                from setup_problem import Setup
                if world > 1:
                    from knpemi.fem.partition import make_slab_problem
                    s = make_slab_problem(kind, r, rank, world, g_syn=10.0, length=2 if scaling == "strong" else None)
                else:
                    s = Setup(kind, r, g_syn=10.0)
                self.models = [(mm["ode"], s.stim_params["stimulus"], s.stim_params["stimulus_locator"])
                               for mm in s.mem_models]
                self.solver_rtol = (1e-5, 1e-7)      # run_3D.py:296-305
                self.source = None
                self.describe = "HH, g_syn=10 for x<20um, dt=1e-4"
        self.s = s
        self.dt = s.dt


def algorithmic_bytes(case, dp):
    """Compulsory bytes per launch, every array counted once: `survey` = the accounting of SURVEY.md section 8(d)
    (FEniCSx-style dofmap + element scatter-slot map), `design` = the arrays this implementation reads and writes
    (4-byte pair entries, 2-byte entry map, distinct-vertex lists, 64-byte vertex records)."""
    from knpemi import _lib as L
    s = case.s
    nv = s.mesh.cells.shape[1]
    gdim = s.mesh.gdim
    nc = int(dp.n_cell.sum())
    N = int(dp.n_vert.sum())
    nF = int(dp.n_facet.sum())
    nf = dp.flat["facet_e"][1].shape[1] if nF else 0
    NQ = int(dp.n_q.sum())
    nnz = dp._pattern(L.A_EMI)[1]
    nnzL = dp._pattern(L.A_KNP)[1] // 2
    n_slots = int(dp.n_models.sum())
    idx = nc * (4 * nv + 4 * nv * nv)          # dofmap + scatter-slot map
    geo = 8 * gdim * N
    gam_idx = nF * (16 + 4 * (2 * nf) ** 2 + 2 * nf * 4)
    survey = {
        # A and P (two matrices, one pass), b_emi, 3 coefficient fields, membrane coupling + Robin RHS
        "emi_rows_kernel": idx + geo + 8 * N * 3 + 2 * 8 * nnz + 8 * N + gam_idx + 8 * NQ + 8 * 2 * NQ,
        # two ion blocks, phi + 2 c_prev coefficients, 2 RHS vectors
        "knp_rows_kernel": idx + geo + 8 * N * 3 + 2 * 8 * nnzL + 2 * 8 * N,
        # pairs + dofs, ~12 gathered coefficient fields on the membrane, RMW of 2 RHS entries per side
        "knp_membrane_kernel": nF * (16 + 2 * nf * 4) + 8 * NQ * 12 + 2 * 8 * 2 * NQ * 2,
        "update_pde_kernel": 8 * N * (2 * 2 + 3) + 8 * NQ * 3,
    }
    # what the kernels touch: per (row, cell) pair 4 bytes (simplices) or 12 (hexahedra), 2 bytes per Laplacian
    # entry, 4 bytes per distinct vertex of a row block (~7 per row on these meshes, measured through lds_uniq),
    # one 64-byte record per distinct vertex of a block, 16-byte row descriptor
    pair_b = 4 if nv != 8 else 12
    # distinct vertices per row block, summed (rows per block: 256 threads / lanes per row, blocks do not straddle
    # sub-domains), from the Laplacian pattern (first ion block of every sub-domain of A_knp)
    import numpy as np
    _, _, krp, kci = dp._pattern(L.A_KNP)
    rpb = 128 if nv == 3 else 64
    uniq = 0
    for sd in range(len(dp.n_vert)):
        r0, nvs = 2 * int(dp.voff[sd]), int(dp.n_vert[sd])
        for a in range(0, nvs, rpb):
            b = min(nvs, a + rpb)
            uniq += np.unique(kci[krp[r0 + a]:krp[r0 + b]]).size
    design = {
        "emi_rows_kernel": nc * nv * pair_b + 2 * nnzL + 4 * uniq + 64 * uniq + 16 * N + 2 * 8 * nnz + 8 * (nnz - nnzL)
                           + 8 * N + 2 * nF * nf * (8 + 4 + 4 * nf + 8 * nf),
        "knp_rows_kernel": nc * nv * pair_b + 2 * nnzL + 4 * uniq + 64 * uniq + 16 * N + 2 * 8 * nnzL + 2 * 8 * N
                           + 2 * nF * nf * 16,
        # per (facet, side): 3 index triples, 2 nf 64-byte records, phi_M and the model's K currents on nf dofs,
        # nf x 16 bytes of partial integrals out + their positions
        "knp_membrane_kernel": 2 * nF * (3 * 4 * nf + 4 + 2 * nf * 64 + 8 * nf * (1 + 3) + nf * (16 + 4)),
        "update_pde_kernel": 8 * N * 2 + 24 * N + 8 * NQ * 3 + 8 * NQ,
    }
    ode = 0
    for m, _, _ in case.models:
        ns, npar = m.states.shape[1], m.parameters.shape[1]
        ode += 2 * 8 * m.nodes * (ns + npar) + 8 * m.nodes * 6 + 8 * m.nodes * 4
    survey["ode_step_kernel"] = design["ode_step_kernel"] = ode
    return survey, design, dict(nc=nc, N=N, nF=nF, NQ=NQ, nnz=nnz, nnzL=nnzL, n_model_slots=n_slots)


def cpu_baseline(s, n_steps, threads=1, traj=None, start=0):
    """C++ port of the reference path (oracle/knpemi_cpu.cpp: scalar element loops with CSR scatter-add, one
    LSODA integration per membrane dof; checked against the numpy oracle by tests/test_cpu_port.py), timed
    on `threads` cores of this host (OpenMP over cells / facets / membrane dofs) for `n_steps` whole steps of the
    same workload.  The CSR patterns come from one untimed oracle assembly.

    `traj` = (phi_t, c_t), the recorded solutions of the GPU's trajectory pass (host arrays): the CPU loop replays the
    same trajectory -- the recorded solution is pasted where a solve would write it, as in the GPU's timed steps -- so
    both legs integrate the same firing membrane.  Steps 0 .. start-1 bring the state to the start of the timed window
    and are not timed."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import adapters
    import cpu_port
    import knpemi_oracle as ko
    o, P, params, ions = adapters.oracle_problem(s)
    c_all, phi, phiM, mm = adapters.oracle_fields(s)
    A, _, _ = o.assemble_emi(P, params, ions, c_all, phiM, mm)
    Ak, _ = o.assemble_knp(P, params, ions, c_all, phi, phiM, mm, s.dt)
    port = cpu_port.CpuPort(P, params, ions, A, Ak)
    cpu_port.lib().cpu_set_threads(int(threads))
    ode = s.mem_models[0]['ode']
    model = ode.ode.MODEL_ID
    ix = o.MODELS[model]["pidx"]
    vi = o.MODELS[model]["V"]
    st, pa = np.ascontiguousarray(ode.states.copy()), np.ascontiguousarray(ode.parameters.copy())
    mask = np.fromiter(map(s.stim_params['stimulus_locator'], ode.dof_locations), dtype=bool)
    sidx = [ix[k] for k in s.stim_params['stimulus']]
    sval = [float(v) for v in s.stim_params['stimulus'].values()]
    names = ("K", "Cl", "Na")
    Ich = np.stack([mm[1][0]["I_ch_k"][n] for n in names]).copy()
    mid = {"hh_si": 0, "hh_mv": 1, "glial": 2}[model]
    boff, _ = ko.knp_block_offsets(P, 2)
    t_asm = t_ode = 0.0
    for k in range(start + n_steps):
        t0 = time.perf_counter()
        for name, kk in zip(names, range(3)):               # update_ode_variables (utils.py:210-235)
            te, ti = P.trace(1, c_all[0][kk], c_all[1][kk])
            pa[:, ix[f"{name}_e"]] = te
            pa[:, ix[f"{name}_i"]] = ti
        if k > 0:
            st[:, vi] = phiM[1]
        failed, _ = port.ode_sweep(mid, st, pa, k * s.dt, s.dt, mask, sidx, sval)
        assert failed == 0
        phiM[1][:] = st[:, vi]                               # run_3D.py:104-109
        for j, name in enumerate(names):
            Ich[j] = pa[:, ix[f"I_ch_{name}"]]
        t1 = time.perf_counter()
        port.assemble_emi(c_all, phiM, Ich)
        if traj is not None:                                 # where the EMI solve writes
            for t in P.tags:
                phi[t][:] = traj[0][k][P.off[t]:P.off[t] + P.N[t]]
        port.assemble_knp(c_all, phi, phiM, Ich)
        if traj is not None:                                 # where the KNP solve writes, then c_prev <- c
            for t in P.tags:
                for kk in range(2):
                    c_all[t][kk][:] = traj[1][k][boff[(t, kk)]:boff[(t, kk)] + P.N[t]]
        te, ti = P.trace(1, phi[0], phi[1])      # end-of-step update (utils.py:238-295)
        phiM[1][:] = ti - te
        for t in c_all:
            c_all[t][2][:] = -(ions[0]["z"] * c_all[t][0] + ions[1]["z"] * c_all[t][1]) / ions[2]["z"]
        t2 = time.perf_counter()
        if k >= start:
            t_ode += t1 - t0
            t_asm += t2 - t1
    # what the leg holds after its last step (trajectory step start + n_steps - 1): ODE tables, currents and the
    # systems assembled in that step, for the comparison with the GPU's replay of the same steps (`agreement`)
    import scipy.sparse as sp
    final = {"k_end": start + n_steps, "states": st.copy(), "I_ch": Ich.copy(),
             "A_emi": sp.csr_matrix((port.A.copy(), port.ci, port.rp), shape=(port.ntot, port.ntot)),
             "P_emi": sp.csr_matrix((port.Pm.copy(), port.ci, port.rp), shape=(port.ntot, port.ntot)),
             "A_knp": sp.csr_matrix((port.Ak.copy(), port.kci, port.krp), shape=(2 * port.ntot, 2 * port.ntot)),
             "b_emi": port.b.copy(), "b_knp": port.bk.copy()}
    return (t_asm + t_ode) / n_steps, t_asm / n_steps, t_ode / n_steps, ode.nodes, final


def agreement(case, replay, final):
    """GPU against the CPU port after both have replayed the same trajectory steps 0 .. k_end - 1 (same recorded solutions
    pasted where the solves write): the membrane ODE states and side-effect currents of the last sweep and the five
    objects assembled in the last step, max |gpu - cpu| / max |cpu| each.  The port is checker here, not product."""
    import numpy as np
    from knpemi import _lib as L
    stepper = replay.stepper
    dp = stepper.dp
    replay.restart(final["k_end"])
    dp.sync()
    m = case.models[0][0]
    st = np.ascontiguousarray(np.empty_like(m.states))
    pa = np.ascontiguousarray(np.empty_like(m.parameters))
    L.check(dp.lib.knpemi_ode_get_tables(dp.h, m._sub, m._model, L.dptr(st), L.dptr(pa)))
    ix = m.ode.parameter_indices
    Ich = np.stack([pa[:, ix(f"I_ch_{n}")] for n in ("K", "Cl", "Na")])

    def rel(a, b):
        scale = float(np.abs(b).max())
        return float(np.abs(a - b).max() / (scale if scale > 0 else 1.0))

    def crel(A, B):
        D = (A - B).tocoo()
        scale = float(np.abs(B.data).max())
        return float((np.abs(D.data).max() if D.nnz else 0.0) / (scale if scale > 0 else 1.0))
    out = {"ode_states": rel(st, final["states"]), "I_ch": rel(Ich, final["I_ch"]),
           "A_emi": crel(dp.csr(L.A_EMI), final["A_emi"]), "P_emi": crel(dp.csr(L.P_EMI), final["P_emi"]),
           "A_knp": crel(dp.csr(L.A_KNP), final["A_knp"]),
           "b_emi": rel(dp.rhs(L.B_EMI), final["b_emi"]), "b_knp": rel(dp.rhs(L.B_KNP), final["b_knp"])}
    out["max"] = max(out.values())
    out["after_trajectory_step"] = final["k_end"] - 1
    return out


def device_solvers(case, its, maxit=1000, dp=None, min_it=None, method="bicgstab", emi_norm=None):
    """Solve callbacks of the stepper: knpemi_solve_emi (CG + AMG) / knpemi_solve_knp (BiCGStab or GMRES + AMG) at the
    reference's tolerances, starting from the extrapolated previous solutions.  `min_it` (with `dp`): fewest iterations of
    the concentration solve in the units of `method` (KNPEMI_OPT_KNP_MIN_IT / _METHOD; None leaves the handle's settings)."""
    from knpemi import _lib as L
    rtol_emi, rtol_knp = case.solver_rtol
    if dp is not None and min_it is not None:
        from knpemi.pdeSolver import set_emi_solver_options, set_knp_solver_options
        set_knp_solver_options(dp, method, min_it)
        set_emi_solver_options(dp, emi_norm or ("preconditioned" if method == "gmres" else "true"))

    def solver(which, key, rtol, atol):
        def run(d):
            L.check(d.lib.knpemi_extrapolate_guess(d.h, which))
            its[key].append(d.solve(which, rtol, atol, maxit)[0])
        return run
    return solver(L.B_EMI, "emi", rtol_emi, 1e-40), solver(L.B_KNP, "knp", rtol_knp, 2e-40)


def ode_counters(case, dp):
    """(rhs evaluations, LSODA steps, failed dofs) since the last call, summed over the membrane models (synchronises)."""
    nr = ns = nf = 0
    for m, _, _ in case.models:
        a, b, c = C.c_int64(), C.c_int64(), C.c_int32()
        dp.lib.knpemi_ode_stats(dp.h, m._sub, m._model, C.byref(a), C.byref(b), C.byref(c))
        nr, ns, nf = nr + a.value, ns + b.value, nf + c.value
    return nr, ns, nf


def record_trajectory(case, stepper, n_steps, halo=None):
    """Untimed: n_steps whole time steps from t = 0 with the device Krylov solves.  Returns the solutions (phi, c in the
    unknown order of the two systems) of every step as host arrays [n_steps, n], the iteration counts and the RHS
    evaluations per membrane dof of every step (where the cell fires)."""
    import numpy as np
    from knpemi import _lib as L
    dp = stepper.dp
    n_emi = dp._pattern(L.A_EMI)[0]
    n_knp = dp._pattern(L.A_KNP)[0]
    phi_t = np.empty((n_steps, n_emi))
    c_t = np.empty((n_steps, n_knp))
    its = {"emi": [], "knp": []}
    solve_emi, solve_knp = device_solvers(case, its)
    k = [0]

    def emi(d):
        solve_emi(d)
        L.check(d.lib.knpemi_get_solution(d.h, L.B_EMI, L.dptr(phi_t[k[0]])))

    def knp(d):
        solve_knp(d)
        L.check(d.lib.knpemi_get_solution(d.h, L.B_KNP, L.dptr(c_t[k[0]])))
        k[0] += 1
    stepper.solve_emi, stepper.solve_knp = emi, knp
    n_dofs = max(1, sum(m.nodes for m, _, _ in case.models))
    ode_counters(case, dp)
    rhs = []
    for _ in range(n_steps):
        stepper.step(halo)
        nr, _, nf = ode_counters(case, dp)
        if nf:
            raise SystemExit("LSODA failed on the device (trajectory pass)")
        rhs.append(nr / n_dofs)
    dp.sync()
    stepper.solve_emi = stepper.solve_knp = None
    return phi_t, c_t, its, rhs


class Replay:
    """The timed steps: the stepper with the recorded solutions pasted where the solves would write them (one launch
    per system, as a solve's own write-back).  `restart(i)` puts the device back to the state before trajectory step i:
    start state re-uploaded, steps 0 .. i-1 replayed untimed."""

    def __init__(self, case, stepper, halo, phi_t, c_t, torch):
        from knpemi import _lib as L
        self.case, self.stepper, self.halo = case, stepper, halo
        dev = torch.device("cuda", stepper.dp.device)
        self.phi_t, self.c_t = torch.from_numpy(phi_t).to(dev), torch.from_numpy(c_t).to(dev)
        self.n = len(phi_t)
        self.cursor = 0
        lib = stepper.dp.lib

        def paste_emi(d):
            L.check(lib.knpemi_set_solution(d.h, L.B_EMI, self.phi_t[self.cursor].data_ptr(), 1))

        def paste_knp(d):
            L.check(lib.knpemi_set_solution(d.h, L.B_KNP, self.c_t[self.cursor].data_ptr(), 1))
            self.cursor += 1
        self.paste = (paste_emi, paste_knp)

    def restart(self, i):
        st = self.stepper
        st.solve_emi = st.solve_knp = None
        st.reset()
        if self.halo is not None:
            self.halo.exchange_bulk()
            self.halo.exchange_membrane()
        if self.case.source is not None:
            st.set_source(0, self.case.source)
        st.solve_emi, st.solve_knp = self.paste
        self.cursor = 0
        self.steps(i)

    def steps(self, n):
        if self.cursor + n > self.n:
            raise SystemExit(f"bench: trajectory of {self.n} steps is too short for step {self.cursor + n}")
        for _ in range(n):
            self.stepper.step(self.halo)


def with_solves(case, replay, start, n_steps, torch, min_it=0, method="bicgstab", emi_norm=None):
    """Whole time steps of run_3D.py:345-368 on the device, Krylov solves included: the
    same stepper with `knpemi_solve_emi` (CG + AMG) and `knpemi_solve_knp` (BiCGStab + AMG) between the assemblies.
    Starts at the FIXED trajectory step `start` (whatever --steps / --warmup are): steps start, start + 1 are untimed
    (initial guesses settle), steps start + 2 .. start + 2 + n_steps are timed.  Not part of `value`.
    `min_it`: fewest BiCGStab iterations of the concentration solve (the reference's ksp_min_it = 5 GMRES iterations are
    three of them, knpemi.pdeSolver.KNP_MIN_BICGSTAB_ITERATIONS)."""
    from knpemi import _lib as L
    stepper, halo = replay.stepper, replay.halo
    dp = stepper.dp
    replay.restart(start)
    its = {"emi": [], "knp": []}
    stepper.solve_emi, stepper.solve_knp = device_solvers(case, its, dp=dp, min_it=min_it, method=method, emi_norm=emi_norm)
    for _ in range(2):
        stepper.step(halo)
    dp.sync()
    torch.cuda.synchronize()
    its["emi"].clear()
    its["knp"].clear()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        stepper.step(halo)
    dp.sync()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n_steps * 1e3
    stepper.solve_emi = stepper.solve_knp = None
    info = {k: dp.solver_info(w) for k, w in (("emi", L.B_EMI), ("knp", L.B_KNP))}
    re, rk = case.solver_rtol
    from knpemi.pdeSolver import set_emi_solver_options, set_knp_solver_options
    set_knp_solver_options(dp, "bicgstab", 0)
    set_emi_solver_options(dp, "true")
    emi_test = ("true residual" if (emi_norm or ("true" if method == "bicgstab" else "preconditioned")) == "true"
                else "preconditioned-norm test (KSPCG's default)")
    knp_name = "BiCGStab" if method == "bicgstab" else "GMRES(30), left preconditioning, preconditioned-norm test"
    return {"ms_per_step": ms, "steps": n_steps, "trajectory_steps": [start + 2, start + 2 + n_steps],
            "knp_min_iterations": int(min_it), "knp_method": method,
            "initial_guess": "3 x_n - 3 x_(n-1) + x_(n-2) (knpemi_extrapolate_guess)",
            "emi": {"solver": f"CG + SA-AMG V(1,1), rtol {re:g}, {emi_test}", "iterations_avg": sum(its["emi"]) / n_steps,
                    "iterations_max": max(its["emi"]), **info["emi"]},
            "knp": {"solver": f"{knp_name} + SA-AMG V(1,1), rtol {rk:g}", "iterations_avg": sum(its["knp"]) / n_steps,
                    "iterations_max": max(its["knp"]), **info["knp"]}}


def run_dg(args, torch, steps=None, warmup=None, cpu=True, dist=None, rank=0, world=1, solves=None):
    """The DG(P1) + interior-penalty variant (csrc/kernels_dg.hip) on the workload's mesh: one step =
    membrane ODE sweep over the facet nodes + potential-system assembly + concentration-systems assembly + end-of-step
    update (+ the ghost-cell refresh at N > 1: x-slabs of an N times longer box, knpemi.dg.DGSlab), device-resident.
    The timed steps hold the fields at the initial state; `with_solves` then runs whole steps with the device solves."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dg_time
    from knpemi import _lib as L
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    family, kind, r = WORKLOADS[args.workload]
    if family != "idealized" or kind not in ("tet", "hex"):
        raise SystemExit("--variant dg runs on the idealized 3-D workloads (config2, config3, r3; config2h: broken Q1 on hexahedra)")
    cell_name = "tetrahedra" if kind == "tet" else "hexahedra"
    slab = None
    with contextlib.redirect_stdout(io.StringIO()):
        if world == 1:
            dp = dg_time.build(r, cell="tetrahedron" if kind == "tet" else "hexahedron")
        else:
            from knpemi.dg import DGSlab
            slab = DGSlab(r, 2 * world, rank, world, device=torch.cuda.current_device(),
                          cell="tetrahedron" if kind == "tet" else "hexahedron")
            dp = dg_time.init_fields(slab.dp)
            slab.attach()
            slab.exchange()
    import mm_hh                                   # the driver's membrane model module (examples/idealized_geometries)
    pi = mm_hh.parameter_indices
    prow = np.asarray(mm_hh.init_parameter_values(), float)
    prow[pi("Cm")] = 0.02
    prow[pi("z_Na")], prow[pi("z_K")], prow[pi("z_Cl")], prow[pi("psi")] = 1.0, 1.0, -1.0, 96485.0 / (8.314 * 300.0)
    names = ("Na", "K", "Cl")
    # the stimulus of the idealized drivers (setup_problem.py:83-84, run_3D.py): g_syn = 10 on the membrane nodes with
    # x < 20 um, written into those nodes' parameter rows -- the stimulated end of the cell fires during the run
    ptab = np.tile(prow, (dp.nmf * dp.nf, 1))
    ptab[dp.XM.reshape(-1, dp.XM.shape[2])[:, 0] < 20e-6, pi("stim_amplitude")] = 10.0
    dp.ode_bind(L.MODEL_HH_SI, np.asarray(mm_hh.init_state_values(), float), ptab,
                sum(([pi(f"{n}_e"), pi(f"{n}_i"), pi(f"I_ch_{n}")] for n in names), []), mm_hh.state_indices("V"))
    c_new = torch.tensor(np.stack([dp.get_concentration(k).ravel() for k in range(2)]), device="cuda")
    torch.cuda.synchronize()
    dt = 1e-4

    def step(k):
        dp.ode_step(k * dt, dt, set_v=k > 0)
        dp.assemble_emi()
        dp.assemble_knp()
        dp.update_device(c_new.data_ptr())
        if slab is not None:
            slab.exchange()

    def sync():
        dp.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(warmup):
        step(k)
    sync()
    dp.ode_stats()
    dp.profile(8)             # every 8th launch of the two assembly kernels carries an event pair
    t0 = time.perf_counter()
    for k in range(steps):
        step(warmup + k)
    sync()
    elapsed = time.perf_counter() - t0
    n_owned_cells = dp.n_cells if slab is None else int(slab.owned_cells.sum())
    if dist is not None:
        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([float(n_owned_cells)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot)
        n_owned_cells = int(tot.item())
    dp.profile(False)
    (_, emi_us), (_, knp_us) = dp.profile_read(0), dp.profile_read(1)
    n_rhs, _, n_failed = dp.ode_stats()
    if n_failed:
        raise SystemExit("LSODA failed on the device")
    dofs = n_owned_cells * dp.nv * dp.K
    ms = elapsed / steps * 1e3

    def roof(which, name, us):
        by = dg_time.algorithmic_bytes(dp, which)
        # PMC counters of this build, collected in their own profiler passes (tools/collect_traffic.sh)
        tfile, tdata = load_traffic(args.workload)
        tr = tdata.get(name + ("_general" if os.environ.get("KNPEMI_DG_HEX_GENERAL") and kind == "hex" else ""))
        traffic = (2.0 * tr["FETCH_SIZE_KiB"] + tr["WRITE_SIZE_KiB"]) * 1024.0 if tr else None
        return {"bound": "hbm", "kernel": name, "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tfile if tr else None,
                "frac_by_counters": traffic / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if traffic else None,
                "algorithmic_bytes_per_launch": by, "avg_launch_us": us,
                "kernel_trace": load_kernel_trace(args.workload, name, "dg")
                if not (os.environ.get("KNPEMI_DG_HEX_GENERAL") and kind == "hex") else None}
    out = {
        "metric": "assembled dofs/s (DG volume + interior-facet SIP + membrane-facet assembly + membrane ODE sweep) per "
                  "timestep; 3D idealized mesh, fp64",
        "value": dofs / (elapsed / steps), "unit": "dofs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: make_mesh_3D geometry r={r}, {dp.n_cells} {cell_name}, broken "
                               f"{'P1' if kind == 'tet' else 'Q1'}: {dp.n} dofs "
                               f"per field, {dp.nnz} CSR entries per system, {dp.nmf * dp.nf} membrane facet nodes (HH), 3 ions",
                   "variant": f"DG({'P1' if kind == 'tet' else 'Q1'}) + symmetric interior penalty (gamma = 10), upwinded drift",
                   "dofs_per_step": dofs,
                   "partition": ("none" if slab is None else
                                 f"x-slabs of a {32 * world} um box (config 2 per GPU), one ghost-cell layer per cut, "
                                 f"ghost dofs refreshed once per step: {slab.mode}"),
                   "state": "concentrations and potential held at the initial state in the timed steps, the membrane ODEs "
                            "stimulated (g_syn = 10 for x < 20 um); whole steps with the device solves in with_solves"},
        "roofline": roof(1, "dg_knp_kernel" if kind == "tet" else "dg_knp_hex_kernel", knp_us),
        "roofline_potential_kernel": roof(0, "dg_emi_kernel" if kind == "tet" else "dg_emi_hex_kernel", emi_us),
        "kernels_us_per_step": {("dg_emi_kernel" if kind == "tet" else "dg_emi_hex_kernel"): emi_us,
                                ("dg_knp_kernel" if kind == "tet" else "dg_knp_hex_kernel"): knp_us},
        "ode": {"rhs_evals_per_dof_per_step": n_rhs / max(1, dp.nmf * dp.nf) / steps},
    }
    n_solve = getattr(args, "solve_steps", 0)
    if world == 1 and (cpu if solves is None else solves) and n_solve > 0 and dp.n <= 1_400_000:
        # whole DG time steps: the two systems solved on the device (CG / BiCGStab + auxiliary-space AMG at the reference's
        # rtol 1e-5 / 1e-7, knpemi_dg_solve_emi/knp), the update taken from the solution without leaving the device
        def solved_step(k):
            dp.ode_step(k * dt, dt, set_v=k > 0)
            dp.assemble_emi()
            a = dp.solve_emi(rtol=1e-5)[0]
            dp.assemble_knp()
            b = dp.solve_knp(rtol=1e-7, update=True)[0]
            return a, b
        k0 = warmup + steps
        dp.set_extrapolation(True)
        t0 = time.perf_counter()
        solved_step(k0)                        # builds the two hierarchies on the host
        dp.sync()
        t_first = time.perf_counter() - t0
        t0 = time.perf_counter()
        its = [solved_step(k0 + 1 + k) for k in range(n_solve)]
        dp.sync()
        ts = (time.perf_counter() - t0) / n_solve
        out["with_solves"] = {"ms_per_step": ts * 1e3, "steps": n_solve, "first_step_with_amg_setup_s": t_first,
                              "cg_iterations_per_step": sum(a for a, _ in its) / n_solve,
                              "bicgstab_iterations_per_step": sum(b for _, b in its) / n_solve,
                              "rtol": [1e-5, 1e-7],
                              "state": "stimulated membrane (g_syn = 10 for x < 20 um), fields evolving: the steps follow the "
                                       "timed ones, i.e. while the stimulated end of the cell depolarises",
                              "phi_M_range_at_the_end": [float(dp.get_membrane_potential().min()),
                                                         float(dp.get_membrane_potential().max())]}
    if cpu and rank == 0 and world == 1:      # the only use of oracle/ in this function: the timed CPU restatement
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import knpemi_dg_oracle as dgo
        from knpemi.fem.idealized import make_mesh_3D
        mesh, ct, ft = make_mesh_3D(0, "tetrahedron" if kind == "tet" else "hexahedron")
        sel = ft.values == 1
        o = dgo.make_dg_oracle(mesh.x, mesh.cells, mesh.cell_type, ct.dense(), mesh.facets[ft.indices[sel]], ft.values[sel])
        shape, ms_ = (o.nc, o.nv), (o.nmf, o.nf)
        ions = [dict(z=1.0, D=[1.33e-9] * 2), dict(z=1.0, D=[1.96e-9] * 2), dict(z=-1.0, D=[2.03e-9] * 2)]
        pr = dict(dt=dt, F=96485.0, psi=96485.0 / (8.314 * 300.0), C_M=0.02)
        c_all = [np.full(shape, v) for v in (100.0, 4.0, 104.0)]
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 10.0:
            o.assemble_emi(pr, ions, c_all, np.zeros(ms_), [np.zeros(ms_)] * 3)
            o.assemble_knp(pr, ions, c_all, np.zeros(shape), np.zeros(ms_), [np.zeros(ms_)] * 3)
            reps += 1
        tc = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": o.n * 3 / tc, "unit": "dofs/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} assemblies (potential + 2 concentration systems, no ODE sweep) of the numpy "
                                         f"restatement oracle/knpemi_dg_oracle.py on the r=0 mesh ({o.nc} {cell_name}, "
                                         f"{o.n} dofs per field), {tc * 1e3:.0f} ms each; there is no reference DG code"}
    return out


def build_problem(workload, args, rank, world):
    """Case + device stepper (+ halo at N > 1) of one workload."""
    from knpemi.stepper import DeviceStepper
    case = Case(workload, rank, world, args.scaling)
    s = case.s
    stepper = DeviceStepper((s.a_emi, s.p_emi, s.L_emi), (s.a_knp, s.p_knp, s.L_knp),
                            s.c, s.c_prev, s.phi, s.phi_M_prev, assemble_knp_twice=args.knp_twice,
                            overlap=not args.no_overlap, fuse_update=not args.frozen_state,
                            early_membrane=bool(os.environ.get("KNPEMI_EARLY_MEMBRANE")))
    if os.environ.get("KNPEMI_OVERLAP_THRESHOLD_US"):      # experiment: when the stepper gives up running the EMI assembly beside the sweep
        stepper.overlap_threshold_ms = float(os.environ["KNPEMI_OVERLAP_THRESHOLD_US"]) * 1e-3
    for m, stim, loc in case.models:
        stepper.add_membrane_model(m, stim, loc)
    if case.source is not None:
        stepper.set_source(0, case.source)
    halo = getattr(s, "halo", None)
    if halo is not None:
        halo.attach(stepper.dp)
        halo.exchange_bulk()        # ghosts start from their owners' values
        halo.exchange_membrane()    # once: afterwards the ghost membrane dofs are integrated redundantly
    return case, stepper, halo


def load_traffic(workload):
    """HBM-side bytes per launch from the PMC passes of this build (tools/collect_traffic.sh -> profiles/rNN_traffic.json,
    collected and corrected as MI355X_MICROARCH.md prescribes); the newest round's file that has the workload."""
    for name in ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json"):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", name)))[workload]
            return name, tr
        except (OSError, KeyError, ValueError):
            continue
    return None, {}


def load_kernel_trace(workload, kernel, variant="cg"):
    """Average duration of `kernel` in the committed rocprofv3 kernel statistics of this build for that leg alone
    (profiles/rNN_<leg>_kernel_stats.csv, one population of launches per row).  The HIP events `avg_launch_us` comes from
    bracket the launch on its stream and so include the dispatch (~2-3 us) that a kernel trace leaves out: for the
    10-15 us kernels of config 2 the two differ by that much, for the long kernels by a few per cent."""
    import csv
    leg = {("cg", "config2"): "config2", ("cg", "config3"): "config3", ("cg", "config2h"): "config2h",
           ("dg", "config2"): "dg", ("dg", "config2h"): "dg_config2h"}.get((variant, workload))
    stem = {"emi_rows_kernel": "emi_rows", "knp_rows_kernel": "knp_rows", "dg_emi_hex_kernel": "dg_emi_hex",
            "dg_knp_hex_kernel": "dg_knp_hex", "dg_emi_kernel": "dg_emi_kernel", "dg_knp_kernel": "dg_knp_kernel"}.get(kernel)
    if leg is None or stem is None:
        return None
    for rnd in ("r04", "r03"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{leg}_kernel_stats.csv")
        try:
            for r in csv.DictReader(open(path)):
                if stem in r["Name"]:
                    return {"avg_us": float(r["AverageNs"]) / 1e3, "launches": int(r["Calls"]), "source": os.path.basename(path),
                            "kernel": r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0],
                            "note": "kernel trace (excludes the dispatch the HIP events of avg_launch_us include)"}
        except (OSError, KeyError, ValueError):
            continue
    return None


def measure(workload, args, torch, dist, rank, world, steps, warmup, repeats, spike=True, traj_min=0):
    """One workload on the recorded trajectory: median-timed window, per-kernel durations, row kernels alone.  Returns
    (line fragment for rank 0, replay, case)."""
    import numpy as np
    from knpemi import _lib as L
    case, stepper, halo = build_problem(workload, args, rank, world)
    s, dp, lib = case.s, stepper.dp, stepper.dp.lib
    frozen = args.frozen_state

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def rank_max(x):
        if dist is None:
            return x
        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if frozen:
        if case.family != "idealized":
            raise SystemExit("--frozen-state is defined for the idealized workloads")
        # synthetic stationary state: c = c_prev (the update keeps the fields), rest potential + a smooth perturbation
        s.perturb(seed=12345 + rank)
        for tag in s.subdomain_list:
            for k in range(2):
                s.c[tag][k].x.array[:] = s.c_prev[tag][k].x._a
            s.ion_list[-1][f'c_{tag}'].x.array[:] = -(1.0 / s.ion_list[-1]['z']) * sum(
                ion['z'] * f.x._a for ion, f in zip(s.ion_list[:-1], s.c_prev[tag]))
        L_x = s.mesh.x[:, 0].max() if world == 1 else s.global_length
        for tag in s.subdomain_list:
            x = s.subdomain_list[tag]['mesh_sub'].x
            s.phi[tag].x.array[:] = (-0.0744 if tag > 0 else 0.0) + 1e-3 * np.sin(2 * np.pi * x[:, 0] / L_x)
        stepper.reset()

    # ---- untimed trajectory pass: whole steps with the device Krylov solves from t = 0.  A failure here (a solve that
    # does not converge, a communication hook) ends the bench with a non-zero exit code: there is no fallback state.
    n_traj = max(warmup + steps + 8, traj_min)
    traj_its = rhs_per_step = None
    replay = None
    if not frozen:
        if halo is not None:
            halo.enable_solves()    # knpemi_solve_emi / knp solve the global systems
        phi_t, c_t, traj_its, rhs_per_step = record_trajectory(case, stepper, n_traj, halo)
        replay = Replay(case, stepper, halo, phi_t, c_t, torch)

    def restart(i):
        if replay is not None:
            replay.restart(i)
        else:
            stepper.reset()
            for _ in range(i):
                stepper.step(halo)

    def run_steps(n):
        if replay is not None:
            replay.steps(n)
        else:
            for _ in range(n):
                stepper.step(halo)

    # ---- untimed profiling pass over the first steps of the window: every kernel bracketed by HIP events
    restart(warmup)
    sync()
    L.check(lib.knpemi_profile(dp.h, (1 << len(L.KERNEL_NAMES)) - 1))
    n_prof = min(PROFILE_STEPS, steps)
    run_steps(n_prof)
    sync()
    per_kernel = {}
    for kid, name in enumerate(L.KERNEL_NAMES):
        n, ms = C.c_int64(), C.c_double()
        L.check(lib.knpemi_profile_read(dp.h, kid, C.byref(n), C.byref(ms)))
        if n.value:
            per_kernel[name] = ms.value / n.value * 1e3 * (n.value / n_prof)   # us per step (all launches)
    L.check(lib.knpemi_profile(dp.h, 0))
    rows = {k: v for k, v in per_kernel.items() if k in ("emi_rows_kernel", "knp_rows_kernel")}
    dominant = max(rows, key=rows.get) if rows else "emi_rows_kernel"
    dom_id = L.KERNEL_NAMES.index(dominant)
    stride = int(os.environ.get("KNPEMI_BENCH_PROFILE_STRIDE", "8"))

    # ---- timed windows.  One window = EXACTLY `steps` steps, trajectory steps [first, first + steps), between
    # barrier + synchronize on both sides; the state is put back to the start of the window before every repeat and
    # `value` comes from the MEDIAN window.  Only the dominant row kernel keeps an event pair,
    # on every 8th launch (an event pair on the critical path costs the step several microseconds).
    def timed_windows(first, reps):
        times, enq = [], []
        dom_n, dom_ms, n_rhs, n_lsoda = 0, 0.0, 0, 0
        for _ in range(reps):
            restart(first)
            L.check(lib.knpemi_profile(dp.h, 1 << dom_id))
            L.check(lib.knpemi_set_option(dp.h, L.OPT_PROFILE_STRIDE, stride))
            ode_counters(case, dp)                      # reset
            sync()
            t0 = time.perf_counter()
            run_steps(steps)
            t_enq = time.perf_counter() - t0           # host side: all launches of the window are enqueued
            sync()
            elapsed = time.perf_counter() - t0
            times.append(rank_max(elapsed))
            enq.append(t_enq)
            nr, ns, nf = ode_counters(case, dp)
            if nf:
                raise SystemExit("LSODA failed on the device")
            n_rhs, n_lsoda = n_rhs + nr, n_lsoda + ns
            n, ms = C.c_int64(), C.c_double()
            L.check(lib.knpemi_profile_read(dp.h, dom_id, C.byref(n), C.byref(ms)))
            dom_n, dom_ms = dom_n + n.value, dom_ms + ms.value
            L.check(lib.knpemi_profile(dp.h, 0))
            L.check(lib.knpemi_set_option(dp.h, L.OPT_PROFILE_STRIDE, 1))
        return dict(times=times, median=float(np.median(times)), enqueue=float(np.median(enq)),
                    dom_us=dom_ms / max(dom_n, 1) * 1e3, rhs=n_rhs / reps, lsoda=n_lsoda / reps)

    main_w = timed_windows(warmup, repeats)
    # the same window with A_knp assembled twice per step, as the reference does it (p = a: knpWeakForm.py:319, the second
    # assembly at pdeSolver.py:131-139; BASELINE.md section 2 and SURVEY section 8d promise both figures)
    twice_w = None
    if not args.knp_twice:
        stepper.assemble_knp_twice = True
        twice_w = timed_windows(warmup, max(3, repeats // 2))
        stepper.assemble_knp_twice = False
    # second window: the same number of steps around the step whose ODE sweep takes longest (the action potential passes
    # the wave that integrates it: the sweep is as slow as its slowest wave).  Found with an untimed replay of the whole
    # trajectory, the sweep bracketed by events and read back after every step.
    spike_w = spike_first = ode_us_per_step = None
    if spike and replay is not None and replay.n >= steps:
        kid = L.KERNEL_NAMES.index("ode_step_kernel")
        restart(0)
        L.check(lib.knpemi_profile(dp.h, 1 << kid))
        ode_us_per_step = []
        for _ in range(replay.n):
            run_steps(1)
            n, ms = C.c_int64(), C.c_double()
            L.check(lib.knpemi_profile_read(dp.h, kid, C.byref(n), C.byref(ms)))      # synchronises, clears the brackets
            ode_us_per_step.append(ms.value * 1e3)
        L.check(lib.knpemi_profile(dp.h, 0))
        if dist is not None:
            # every rank must pick the SAME window (the replay exchanges halos step by step: ranks that replayed
            # different numbers of steps wait for each other forever): the slowest rank's sweep decides, step by step
            red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            tt = torch.tensor(ode_us_per_step, dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ode_us_per_step = [float(v) for v in tt.cpu()]
        peak = int(np.argmax(ode_us_per_step))
        spike_first = int(min(max(0, peak - steps // 2), len(rhs_per_step) - steps))
        spike_w = timed_windows(spike_first, max(3, repeats // 2))
        # the sweep's duration inside that window (every launch bracketed, untimed pass)
        restart(spike_first)
        L.check(lib.knpemi_profile(dp.h, 1 << L.KERNEL_NAMES.index("ode_step_kernel")))
        run_steps(steps)
        sync()
        n, ms = C.c_int64(), C.c_double()
        L.check(lib.knpemi_profile_read(dp.h, L.KERNEL_NAMES.index("ode_step_kernel"), C.byref(n), C.byref(ms)))
        spike_w["ode_us"] = ms.value / max(n.value, 1) * 1e3 * (n.value / steps)
        L.check(lib.knpemi_profile(dp.h, 0))

    # ---- the row kernels by themselves: while the EMI assembly shares the chip with the ODE sweep its launches are
    # stretched by the sweep's waves.  A short untimed pass with everything on one stream gives the durations the
    # kernels reach alone (`roofline_row_kernels_alone`); the step times above are the overlapped schedule's.
    alone = {}
    overlapped = bool(stepper.overlap)
    restart(warmup)
    stepper.overlap = False
    # (the facet kernel as a launch of its own in this pass: in the timed steps its work rides in the launch that writes the
    # potential back, KNPEMI_OPT_FOLD_MEMBRANE)
    L.check(lib.knpemi_set_option(dp.h, L.OPT_FOLD_MEMBRANE, 0))
    L.check(lib.knpemi_profile(dp.h, (1 << L.KERNEL_NAMES.index("emi_rows_kernel")) | (1 << L.KERNEL_NAMES.index("knp_rows_kernel"))
                               | (1 << L.KERNEL_NAMES.index("knp_membrane_kernel"))))
    run_steps(min(8, steps))
    sync()
    L.check(lib.knpemi_set_option(dp.h, L.OPT_FOLD_MEMBRANE, 1))
    for name in ("emi_rows_kernel", "knp_rows_kernel", "knp_membrane_kernel"):
        n, ms = C.c_int64(), C.c_double()
        L.check(lib.knpemi_profile_read(dp.h, L.KERNEL_NAMES.index(name), C.byref(n), C.byref(ms)))
        if n.value:
            alone[name] = ms.value / n.value * 1e3
    L.check(lib.knpemi_profile(dp.h, 0))
    stepper.overlap = overlapped

    owned = getattr(s, "owned_dofs", None)
    dofs_local = 3 * (owned if owned is not None else int(dp.n_vert.sum()))
    if dist is not None:
        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tot = torch.tensor([dofs_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot)
        dofs_total = int(tot.item())
    else:
        dofs_total = dofs_local
    out = None
    if rank == 0:
        survey_b, design_b, sizes = algorithmic_bytes(case, dp)
        n_ode_dofs = sum(m.nodes for m, _, _ in case.models)
        traffic_file, traffic = load_traffic(workload) if world == 1 else (None, {})

        def roof(kernel, us):
            tr = traffic.get(kernel)
            by_counters = (2.0 * tr["FETCH_SIZE_KiB"] + tr["WRITE_SIZE_KiB"]) * 1024.0 if tr else None
            ach = survey_b[kernel] / (us * 1e-6) / 1e9
            return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": by_counters, "traffic_source": traffic_file if tr else None,
                    # the same launch priced with the bytes the PMC counters saw instead of the SURVEY accounting
                    "frac_by_counters": by_counters / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if by_counters else None,
                    "algorithmic_bytes_per_launch": survey_b[kernel], "avg_launch_us": us,
                    "kernel_trace": load_kernel_trace(workload, kernel) if world == 1 else None,
                    "bytes_this_design_touches": design_b[kernel],
                    "frac_of_design_bytes": design_b[kernel] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    # a launch cannot be shorter than the launch floor: the fraction of peak a perfect kernel would
                    # reach at this size (1.0 = large enough for HBM to be the bound)
                    "launch_floor_us": LAUNCH_FLOOR_US,
                    "frac_at_launch_floor": min(1.0, survey_b[kernel] / (LAUNCH_FLOOR_US * 1e-6) / 1e9 / HBM_PEAK_GBS)}
        med = main_w["median"]
        mem_us = alone.pop("knp_membrane_kernel", 0.0)        # the facet kernel as its own launch (untimed pass above)
        if "knp_membrane_kernel" in per_kernel:               # in the timed steps: phi write-back + facet integrals, one launch
            per_kernel["emi_writeback_membrane_kernel"] = per_kernel.pop("knp_membrane_kernel")
        out = {
            "value": dofs_total / (med / steps), "unit": "dofs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": med / steps * 1e3,
            "timing": {"what": f"median of {len(main_w['times'])} repeats of the timed window; one window = {steps} steps, "
                               f"trajectory steps [{warmup}, {warmup + steps}) (t = {warmup * case.dt:g} .. "
                               f"{(warmup + steps) * case.dt:g}), state restored before every repeat",
                       "repeats_ms_per_step": [t / steps * 1e3 for t in main_w["times"]],
                       "min_ms_per_step": min(main_w["times"]) / steps * 1e3,
                       "max_ms_per_step": max(main_w["times"]) / steps * 1e3},
            "host_enqueue_ms_per_step": main_w["enqueue"] / steps * 1e3,
            "config": {"workload": f"{workload}: make_mesh_3D geometry r={case.r}, {case.kind}, "
                                   f"{sizes['nc']} cells/GPU, {sizes['N']} sub-mesh vertices/GPU in "
                                   f"{len(s.subdomain_list)} sub-domains, {n_ode_dofs} membrane ODE dofs/GPU, 3 ions "
                                   f"(K, Cl, Na eliminated), {case.describe}",
                       "dofs_per_step": dofs_total, "A_knp_assemblies_per_step": 2 if args.knp_twice else 1,
                       "emi_matrix_beside_ode_sweep": overlapped,
                       "update_fused_into_knp_write_back": bool(stepper.fuse_update),
                       "facet_integrals_in_the_potential_write_back_launch": True,
                       "state": ("fields frozen at the initial state (phi_M reset every step)" if frozen else
                                 f"recorded trajectory of the first {n_traj} time steps from t = 0 (device Krylov solves, "
                                 f"untimed); the timed steps replay it, pasting each recorded solution where the solve "
                                 f"writes it"),
                       "partition": "x-slabs" if world > 1 else "none"},
            "roofline": roof(dominant, main_w["dom_us"]),
            "roofline_membrane_facet_kernel": roof("knp_membrane_kernel", mem_us) if mem_us > 0 else None,
            "roofline_row_kernels_alone": {k: roof(k, v) for k, v in alone.items()} or None,
            "kernels_us_per_step": per_kernel,
            "ode": {"rhs_evals_per_dof_per_step": main_w["rhs"] / max(1, n_ode_dofs) / steps,
                    "lsoda_steps_per_dof_per_step": main_w["lsoda"] / max(1, n_ode_dofs) / steps,
                    "kernel_us_per_step": per_kernel.get("ode_step_kernel"),
                    "share_of_step": per_kernel.get("ode_step_kernel", 0.0) / (med / steps * 1e6)},
        }
        if twice_w is not None:
            tm = twice_w["median"]
            out["reference_faithful_knp_assembled_twice"] = {
                "what": "the same timed window with A_knp assembled twice per step (the reference passes p = a and "
                        "LinearProblem assembles both: knpWeakForm.py:319, pdeSolver.py:131-139)",
                "A_knp_assemblies_per_step": 2, "value": dofs_total / (tm / steps), "unit": "dofs/s",
                "ms_per_step": tm / steps * 1e3, "repeats_ms_per_step": [t / steps * 1e3 for t in twice_w["times"]]}
        if spike_w is not None:
            sm = spike_w["median"]
            out["spike_window"] = {
                "what": f"the same measurement on the {steps} steps around the step with the longest ODE sweep (step "
                        f"{int(np.argmax(ode_us_per_step))}: {max(ode_us_per_step):.0f} us): trajectory steps "
                        f"[{spike_first}, {spike_first + steps})",
                "ode_kernel_us_max": max(ode_us_per_step),
                "ode_kernel_us_along_the_trajectory": [round(x, 1) for x in ode_us_per_step],
                "ms_per_step": sm / steps * 1e3, "value": dofs_total / (sm / steps),
                "repeats_ms_per_step": [t / steps * 1e3 for t in spike_w["times"]],
                "ode_kernel_us_per_step": spike_w["ode_us"],
                "rhs_evals_per_dof_per_step": spike_w["rhs"] / max(1, n_ode_dofs) / steps}
        if rhs_per_step is not None:
            out["ode"]["rhs_evals_per_dof_along_the_trajectory"] = [round(x, 1) for x in rhs_per_step]
        if halo is not None:
            out["config"]["halo"] = halo.mode
        if traj_its is not None:
            out["config"]["trajectory_iterations_avg"] = {k: sum(v) / max(1, len(v)) for k, v in traj_its.items()}
    return out, replay, case, dofs_total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=7,
                    help="repeats of the timed window (state restored before each); `value` is the median")
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = box length proportional to N (config 2 per GPU), strong = the fixed config-3 box "
                         "(995 328 tets at --workload config3) cut into N slabs")
    ap.add_argument("--cpu-steps", type=int, default=100,
                    help="CPU-port steps timed for cpu_baseline on all host cores; the 1-core leg runs 0.4 x as many "
                         "(0 = skip)")
    ap.add_argument("--solve-steps", type=int, default=20,
                    help="extra untimed-for-`value` pass: whole time steps including the device Krylov solves, reported "
                         "as `with_solves` (0 = skip)")
    ap.add_argument("--frozen-state", action="store_true",
                    help="hold the fields at their initial state instead of replaying a recorded trajectory (the "
                         "round-1 measurement: phi_M is reset every step and the cell never fires)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the EMI matrix assembly after the ODE sweep instead of beside it (aux stream)")
    ap.add_argument("--variant", default="cg", choices=["cg", "dg"],
                    help="cg: the reference's formulation (CG on sub-meshes, the parity path); dg: the DG(P1) + interior "
                         "penalty variant of SURVEY.md section 8 f4 on the same mesh")
    ap.add_argument("--no-dg", action="store_true", help="skip the short DG-variant measurement appended to the cg line")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the 995 328-tet leg appended to the default line (`config3_leg`: the size at which the "
                         "row kernels are HBM-bound rather than launch-bound)")
    ap.add_argument("--knp-twice", action="store_true",
                    help="assemble A_knp twice per step as the reference does (p = a, knpWeakForm.py:319)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher (the reference is started with plain `mpirun` on an unchanged script,
        # examples/idealized_geometries/run_3D.py:27,117-121): start one rank per GPU as a CHILD process -- before torch is
        # imported or anything touches the GPU, never by replacing this process -- and hand its exit code on.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))

    if os.environ.get("KNPEMI_BENCH_WATCHDOG"):   # diagnosis of a stuck rank: Python stacks of every thread after so many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["KNPEMI_BENCH_WATCHDOG"]), repeat=True, file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("KNPEMI_BENCH_LAUNCH_PROBE"):   # test hook of the self-launch: report the rank's environment, exit with the given code
        print(f"bench launch probe: rank {rank} of {world}, local rank {local_rank}, --gpus {args.gpus}", flush=True)
        raise SystemExit(int(os.environ["KNPEMI_BENCH_LAUNCH_PROBE"]) if rank == world - 1 else 0)
    import torch
    if args.gpus != world:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE = {world} (one rank per GPU; without a launcher "
                         f"`python bench.py --gpus N` starts torch.distributed.run itself)")
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("KNPEMI_BENCH_BACKEND", "nccl")   # "gloo": single-GPU rehearsal only
        dev = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    if args.variant == "dg":
        out = run_dg(args, torch, dist=dist, rank=rank, world=world)
        if rank == 0:
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if world > 1 and args.scaling == "strong" and args.workload == "config2":
        args.workload = "config3"       # the fixed mesh of BASELINE.json configs[2]

    # the trajectory is long enough to contain the action potential of the stimulated end of the cell (SPIKE_SEARCH)
    out, replay, case, dofs_total = measure(args.workload, args, torch, dist, rank, world, args.steps, args.warmup,
                                            args.repeats, spike=not args.frozen_state, traj_min=SPIKE_SEARCH)
    if rank == 0:
        out = {"metric": "assembled dofs/s (volume + membrane-facet assembly + membrane ODE sweep) per timestep; "
                         "3D idealized mesh, fp64",
               **{k: out[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")},
               "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               **{k: v for k, v in out.items() if k not in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")}}
    ws = ws_min = ws_ref = None
    if args.solve_steps > 0 and replay is not None:      # collective at N > 1: every rank takes part
        ws = with_solves(case, replay, WITH_SOLVES_START, args.solve_steps, torch)
        # the same with the reference's ksp_min_it = 5 of the concentration solve (pdeSolver.py:101): five GMRES iterations
        # apply operator and preconditioner five times, three BiCGStab iterations six times
        from knpemi.pdeSolver import KNP_MIN_BICGSTAB_ITERATIONS
        ws_min = with_solves(case, replay, WITH_SOLVES_START, args.solve_steps, torch, min_it=KNP_MIN_BICGSTAB_ITERATIONS)
        # ... and with the reference's options as PETSc runs them: GMRES(30), left preconditioning, preconditioned norm,
        # ksp_min_it = 5 GMRES iterations (single rank: the fused loops)
        ws_ref = with_solves(case, replay, WITH_SOLVES_START, args.solve_steps, torch, min_it=5, method="gmres") if world == 1 else None
    if rank == 0:
        if ws is not None:
            out["with_solves"] = ws
            out["with_solves_reference_ksp_min_it"] = {
                "what": "the same steps with ksp_min_it = 5 of the reference's concentration solve honoured (pdeSolver.py:101): "
                        "at least three BiCGStab iterations = six applications of operator and preconditioner",
                **{k: ws_min[k] for k in ("ms_per_step", "steps", "knp_min_iterations")},
                "emi_iterations_avg": ws_min["emi"]["iterations_avg"], "knp_iterations_avg": ws_min["knp"]["iterations_avg"]}
            if world == 1:
                # the device's BiCGStab for the concentrations, the potential solve stopping where the REFERENCE's CG stops
                # (KSPCG's default test on the preconditioned residual) instead of on the true residual
                ws_mix = with_solves(case, replay, WITH_SOLVES_START, args.solve_steps, torch, min_it=KNP_MIN_BICGSTAB_ITERATIONS,
                                     emi_norm="preconditioned")
                out["with_solves_reference_tests_fastest_methods"] = {
                    "what": "the same steps with the stopping rules of the reference (potential: KSPCG's default test |M^-1 r| <= "
                            "rtol |M^-1 b|, pdeSolver.py:60-72; concentrations: ksp_min_it = 5 honoured as three BiCGStab "
                            "iterations, pdeSolver.py:101) and the device's cheaper method for the concentrations (BiCGStab on "
                            "the true residual instead of GMRES(30))",
                    "emi_solver": ws_mix["emi"]["solver"],
                    **{k: ws_mix[k] for k in ("ms_per_step", "steps", "knp_min_iterations", "knp_method")},
                    "emi_iterations_avg": ws_mix["emi"]["iterations_avg"], "knp_iterations_avg": ws_mix["knp"]["iterations_avg"]}
            if ws_ref is not None:
                out["with_solves_reference_options"] = {
                    "what": "the same steps with both solves as the reference configures them: the concentration solve with "
                            "ksp_type gmres, ksp_min_it 5 (pdeSolver.py:99-110; PETSc's defaults: restart 30, left "
                            "preconditioning, classical Gram-Schmidt, preconditioned residual norm against |M^-1 b|), the "
                            "potential solve's CG with KSPCG's default test |M^-1 r| <= rtol |M^-1 b| (pdeSolver.py:60-72)",
                    "emi_solver": ws_ref["emi"]["solver"],
                    **{k: ws_ref[k] for k in ("ms_per_step", "steps", "knp_min_iterations", "knp_method")},
                    "emi_iterations_avg": ws_ref["emi"]["iterations_avg"], "knp_iterations_avg": ws_ref["knp"]["iterations_avg"]}
        s = case.s
        if args.cpu_steps > 0 and world == 1 and case.family == "idealized":
            avail = len(os.sched_getaffinity(0))
            # bounded sample: --cpu-steps refers to the config-2 size and shrinks with the problem size
            n_all = max(3, int(round(args.cpu_steps * min(1.0, 79251.0 / dofs_total))))
            n_all = min(n_all, replay.n - args.warmup) if replay is not None else n_all
            n1 = max(2, int(0.4 * n_all))
            traj = (replay.phi_t.cpu().numpy(), replay.c_t.cpu().numpy()) if replay is not None else None
            quiet = io.StringIO()
            with contextlib.redirect_stdout(quiet):
                t_one, a_one, o_one, _, _ = cpu_baseline(s, n1, threads=1, traj=traj, start=args.warmup)   # before any OpenMP team exists
                # thread count: a fully subscribed host can be slower than a partly subscribed one (spinning OpenMP
                # team + the Python thread, CPU shares below the visible core count): probe and time the best
                cand = sorted({min(avail, c) for c in (4, 8, 16, 32, avail)})
                probes = {c: cpu_baseline(s, 3, threads=c)[0] for c in cand}
                cores = min(probes, key=probes.get)
                agree = None
                t_all, a_all, o_all, nrows, final = cpu_baseline(s, n_all, threads=cores, traj=traj, start=args.warmup)
                agree = agreement(case, replay, final) if replay is not None else None
            what = ("whole steps of the C++ port (oracle/knpemi_cpu.cpp, sequential ODEPACK restatement "
                    "oracle/lsoda_seq.h) on the same mesh, replaying the same recorded trajectory from step {w} (the "
                    "window the GPU's value is timed on): EMI (A, P, b) + KNP (A, b) assembly and update {a:.1f} ms/step, "
                    "LSODA sweep over all {n} membrane dofs {o:.1f} ms/step; the reference itself cannot run here")
            out["cpu_baseline"] = {
                "value": dofs_total / t_all, "unit": "dofs/s", "cores": cores, "kind": "port",
                "sample": f"{n_all} " + what.format(a=a_all * 1e3, o=o_all * 1e3, n=nrows, w=args.warmup)
                          + f"; OpenMP over cells / facets / membrane dofs, {cores} threads (fastest of "
                            f"{cand} on the {avail} cores visible to this process)"}
            if agree is not None:
                # both legs replayed the same steps: their numbers must agree (ODE tolerances 1e-8 / 1e-10, libm against the
                # device's exp / log / quotients; operators of identical inputs to rounding) -- a disagreement is a failed run
                out["cpu_baseline"]["agreement"] = {
                    "what": "max |GPU - CPU port| / max |CPU port| after both legs replayed trajectory steps 0 .. "
                            f"{agree['after_trajectory_step']}: membrane ODE states and channel currents of the last sweep, "
                            "the five objects assembled in the last step", **agree, "limit": AGREEMENT_LIMIT}
                if not agree["max"] <= AGREEMENT_LIMIT:
                    print(json.dumps(out["cpu_baseline"]["agreement"]), file=sys.stderr)
                    raise SystemExit(f"bench: GPU and CPU port disagree by {agree['max']:.3e} (> {AGREEMENT_LIMIT:g}) on the "
                                     f"same trajectory")
            out["cpu_baseline_1core"] = {
                "value": dofs_total / t_one, "unit": "dofs/s", "cores": 1, "kind": "port",
                "sample": f"{n1} " + what.format(a=a_one * 1e3, o=o_one * 1e3, n=nrows, w=args.warmup)
                          + "; one thread, as the reference's serial run"}
    tet = case.family == "idealized" and case.kind == "tet"
    del replay, case
    if world == 1 and tet and args.workload == "config2" and not args.no_config3 and not args.frozen_state:
        # The 995 328-tet mesh of BASELINE configs[2] on this one GPU: config 2 is launch-bound (a perfect row kernel
        # reaches < 0.5 of the HBM peak at the launch floor), this is the size at which the roofline fraction says
        # something about the kernels.  Same measurement, shorter: no spike window, no solves pass, no CPU leg.
        leg, _, _, _ = measure("config3", args, torch, None, 0, 1, steps=10, warmup=5, repeats=5, spike=False)
        out["config3_leg"] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "timing", "config", "roofline",
                                                   "roofline_membrane_facet_kernel", "roofline_row_kernels_alone",
                                                   "kernels_us_per_step", "ode", "reference_faithful_knp_assembled_twice")
                              if k in leg}
    if rank == 0:
        if world == 1 and not args.no_dg and tet:
            dg = run_dg(args, torch, steps=10, warmup=2, cpu=False, solves=True)
            out["dg_variant"] = {k: dg[k] for k in ("value", "unit", "ms_per_step", "config", "roofline",
                                                     "roofline_potential_kernel", "kernels_us_per_step", "with_solves")}
            # the same variant on the reference's own 3-D cell type: broken Q1 on the hexahedral box (config2h, 165 888 cells)
            hargs = argparse.Namespace(**{**vars(args), "workload": "config2h"})
            dgh = run_dg(hargs, torch, steps=10, warmup=2, cpu=False)
            out["dg_variant_hexahedra"] = {k: dgh[k] for k in ("value", "unit", "ms_per_step", "config", "roofline",
                                                                "roofline_potential_kernel", "kernels_us_per_step")}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
