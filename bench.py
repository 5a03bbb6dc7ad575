#!/usr/bin/env python3
"""bench.py -- assembled dofs/s of the KNP-EMI per-time-step hot path on MI355X.

One "step" = one pass of the hot path over one field state, device-resident (SURVEY.md section 8d /
BASELINE.md section 2): fused ODE launch (trace refresh + LSODA sweep + copy-back), EMI assembly (A, P, b in one
pass), KNP assembly (A once, b incl. the membrane kernel), end-of-step update.  Krylov solves and file output are
excluded from the timed region -- but the state the timed steps run on is a REAL trajectory: an untimed pass first
integrates the same problem with the device Krylov solves and records the solution of every step; the timed steps
start again from t = 0 and paste the recorded solution where the solve would write it (one launch per system, as the
solve's own write-back).  The membrane therefore depolarises and fires as in a real run and the ODE sweep does the
work a real run gives it (`ode_rhs_evals_per_dof_per_step`).

At N > 1 the mesh is partitioned into x-slabs (weak: N times longer box; --scaling strong: the fixed config-3 box),
every step exchanges the ghost-dof halo of the bulk fields (stream-ordered RCCL point-to-point; ghost membrane dofs
are integrated redundantly), and the recorded trajectory comes from the distributed Krylov solves
(knpemi_set_distributed: halo'd SpMV, all-reduced dot products, per-GPU AMG).

    python bench.py --gpus 1 --steps 50 --warmup 5
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

WORKLOADS = {
    # name: (family, mesh kind, resolution factor) -- geometry of make_mesh_3D.py, 6 tets per hexahedron
    "config2": ("idealized", "tet", 1),     # BASELINE.json configs[1]: 124 416 tets, 79 251 dofs/step
    "config2h": ("idealized", "hex", 2),    # reference-faithful Q1 hexahedra, 165 888 cells
    "config3": ("idealized", "tet", 2),     # 995 328 tets on ONE GPU (the 8-GPU mesh of configs[2])
    "r3": ("idealized", "tet", 3),          # 7.96 M tets
    "2d": ("idealized", "2d", 3),
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
                if world > 1:
                    raise SystemExit("config5s runs on one GPU (the slab partitioner builds the idealized set-up)")
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


def cpu_baseline(s, n_steps, threads=1):
    """C++ port of the reference path (oracle/knpemi_cpu.cpp: scalar element loops with CSR scatter-add, one
    LSODA integration per membrane dof; checked against the numpy oracle by tests/test_cpu_port.py), timed
    on `threads` cores of this host (OpenMP over cells / facets / membrane dofs) for `n_steps` whole steps of the
    same workload.  The CSR patterns come from one untimed oracle assembly."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import adapters
    import cpu_port
    o, P, params, ions = adapters.oracle_problem(s)
    c_all, phi, phiM, mm = adapters.oracle_fields(s)
    A, _, _ = o.assemble_emi(P, params, ions, c_all, phiM, mm)
    Ak, _ = o.assemble_knp(P, params, ions, c_all, phi, phiM, mm, s.dt)
    port = cpu_port.CpuPort(P, params, ions, A, Ak)
    cpu_port.lib().cpu_set_threads(int(threads))
    ode = s.mem_models[0]['ode']
    model = ode.ode.MODEL_ID
    ix = o.MODELS[model]["pidx"]
    st, pa = np.ascontiguousarray(ode.states.copy()), np.ascontiguousarray(ode.parameters.copy())
    mask = np.fromiter(map(s.stim_params['stimulus_locator'], ode.dof_locations), dtype=bool)
    sidx = [ix[k] for k in s.stim_params['stimulus']]
    sval = [float(v) for v in s.stim_params['stimulus'].values()]
    Ich = np.stack([mm[1][0]["I_ch_k"][n] for n in ("K", "Cl", "Na")])
    mid = {"hh_si": 0, "hh_mv": 1, "glial": 2}[model]
    t_asm = t_ode = 0.0
    for k in range(n_steps):
        t0 = time.perf_counter()
        for name, kk in (("K", 0), ("Cl", 1), ("Na", 2)):
            te, ti = P.trace(1, c_all[0][kk], c_all[1][kk])
            pa[:, ix[f"{name}_e"]] = te
            pa[:, ix[f"{name}_i"]] = ti
        failed, _ = port.ode_sweep(mid, st, pa, k * s.dt, s.dt, mask, sidx, sval)
        assert failed == 0
        t1 = time.perf_counter()
        port.assemble_emi(c_all, phiM, Ich)
        port.assemble_knp(c_all, phi, phiM, Ich)
        te, ti = P.trace(1, phi[0], phi[1])      # end-of-step update (vector copies + trace)
        phiM[1][:] = ti - te
        for t in c_all:
            c_all[t][2][:] = -(ions[0]["z"] * c_all[t][0] + ions[1]["z"] * c_all[t][1]) / ions[2]["z"]
        t2 = time.perf_counter()
        t_ode += t1 - t0
        t_asm += t2 - t1
    return (t_asm + t_ode) / n_steps, t_asm / n_steps, t_ode / n_steps, ode.nodes


def device_solvers(case, its, maxit=1000):
    """Solve callbacks of the stepper: knpemi_solve_emi (CG + AMG) / knpemi_solve_knp (BiCGStab + AMG) at the
    reference's tolerances, starting from 2 x_n - x_(n-1)."""
    from knpemi import _lib as L
    rtol_emi, rtol_knp = case.solver_rtol

    def solver(which, key, rtol, atol):
        def run(d):
            L.check(d.lib.knpemi_extrapolate_guess(d.h, which))
            its[key].append(d.solve(which, rtol, atol, maxit)[0])
        return run
    return solver(L.B_EMI, "emi", rtol_emi, 1e-40), solver(L.B_KNP, "knp", rtol_knp, 2e-40)


def record_trajectory(case, stepper, n_steps, torch, halo=None):
    """Untimed: n_steps whole time steps with the device Krylov solves; returns the solutions (phi, c in the unknown
    order of the two systems) of every step as device tensors [n_steps, n]."""
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
    for _ in range(n_steps):
        stepper.step(halo)
    dp.sync()
    stepper.solve_emi = stepper.solve_knp = None
    dev = torch.device("cuda", dp.device)
    return torch.from_numpy(phi_t).to(dev), torch.from_numpy(c_t).to(dev), its


def with_solves(case, stepper, n_steps, torch, halo=None):
    """Whole time steps of run_3D.py:345-368 on the device, Krylov solves included: the
    same stepper with `knpemi_solve_emi` (CG + AMG) and `knpemi_solve_knp` (BiCGStab + AMG) between the assemblies,
    continuing from the end of the timed trajectory.  Not part of `value`."""
    from knpemi import _lib as L
    dp = stepper.dp
    its = {"emi": [], "knp": []}
    stepper.solve_emi, stepper.solve_knp = device_solvers(case, its)
    for _ in range(2):
        stepper.step(halo)
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
    return {"ms_per_step": ms, "steps": n_steps, "initial_guess": "2 x_n - x_(n-1) (knpemi_extrapolate_guess)",
            "emi": {"solver": f"CG + SA-AMG V(1,1), rtol {re:g}", "iterations_avg": sum(its["emi"]) / n_steps, **info["emi"]},
            "knp": {"solver": f"BiCGStab + SA-AMG V(1,1), rtol {rk:g}", "iterations_avg": sum(its["knp"]) / n_steps,
                    **info["knp"]}}


def run_dg(args, torch, steps=None, warmup=None, cpu=True, dist=None, rank=0, world=1):
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
    if family != "idealized" or kind != "tet":
        raise SystemExit("--variant dg runs on the idealized tetrahedral workloads (config2, config3, r3)")
    slab = None
    with contextlib.redirect_stdout(io.StringIO()):
        if world == 1:
            dp = dg_time.build(r)
        else:
            from knpemi.dg import DGSlab
            slab = DGSlab(r, 2 * world, rank, world, device=torch.cuda.current_device())
            dp = dg_time.init_fields(slab.dp)
            slab.attach()
            slab.exchange()
    import mm_hh                                   # the driver's membrane model module (examples/idealized_geometries)
    pi = mm_hh.parameter_indices
    prow = np.asarray(mm_hh.init_parameter_values(), float)
    prow[pi("Cm")] = 0.02
    prow[pi("z_Na")], prow[pi("z_K")], prow[pi("z_Cl")], prow[pi("psi")] = 1.0, 1.0, -1.0, 96485.0 / (8.314 * 300.0)
    names = ("Na", "K", "Cl")
    dp.ode_bind(L.MODEL_HH_SI, np.asarray(mm_hh.init_state_values(), float), prow,
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
        traffic = None      # PMC counters of this build, collected in their own profiler passes (tools/collect_traffic.sh)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))[args.workload][name]
            traffic = (2.0 * tr["FETCH_SIZE_KiB"] + tr["WRITE_SIZE_KiB"]) * 1024.0
        except (OSError, KeyError, ValueError):
            pass
        return {"bound": "hbm", "kernel": name, "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": by,
                "avg_launch_us": us}
    out = {
        "metric": "assembled dofs/s (DG volume + interior-facet SIP + membrane-facet assembly + membrane ODE sweep) per "
                  "timestep; 3D idealized mesh, fp64",
        "value": dofs / (elapsed / steps), "unit": "dofs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: make_mesh_3D geometry r={r}, {dp.n_cells} tetrahedra, broken P1: {dp.n} dofs "
                               f"per field, {dp.nnz} CSR entries per system, {dp.nmf * dp.nf} membrane facet nodes (HH), 3 ions",
                   "variant": "DG(P1) + symmetric interior penalty (gamma = 10), upwinded drift", "dofs_per_step": dofs,
                   "partition": ("none" if slab is None else
                                 f"x-slabs of a {32 * world} um box (config 2 per GPU), one ghost-cell layer per cut, "
                                 f"ghost dofs refreshed once per step: {slab.mode}"),
                   "state": "fields held at the initial state in the timed steps; whole steps with the device solves in with_solves"},
        "roofline": roof(1, "dg_knp_kernel", knp_us),
        "roofline_potential_kernel": roof(0, "dg_emi_kernel", emi_us),
        "kernels_us_per_step": {"dg_emi_kernel": emi_us, "dg_knp_kernel": knp_us},
        "ode": {"rhs_evals_per_dof_per_step": n_rhs / max(1, dp.nmf * dp.nf) / steps},
    }
    n_solve = getattr(args, "solve_steps", 0)
    if world == 1 and cpu and n_solve > 0 and dp.n <= 1_000_000:
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
                              "rtol": [1e-5, 1e-7], "state": "membrane at rest (no stimulus), fields evolving"}
    if cpu and rank == 0 and world == 1:      # the only use of oracle/ in this function: the timed CPU restatement
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import knpemi_dg_oracle as dgo
        from knpemi.fem.idealized import make_mesh_3D
        mesh, ct, ft = make_mesh_3D(0, "tetrahedron")
        sel = ft.values == 1
        o = dgo.DGOracle(mesh.x, mesh.cells, mesh.cell_type, ct.dense(), mesh.facets[ft.indices[sel]], ft.values[sel])
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
                                         f"restatement oracle/knpemi_dg_oracle.py on the r=0 mesh ({o.nc} tetrahedra, "
                                         f"{o.n} dofs per field), {tc * 1e3:.0f} ms each; there is no reference DG code"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = box length proportional to N (config 2 per GPU), strong = the fixed config-3 box "
                         "(995 328 tets at --workload config3) cut into N slabs")
    ap.add_argument("--cpu-steps", type=int, default=100,
                    help="CPU-port steps timed for cpu_baseline on all host cores; the 1-core leg runs 0.4 x as many "
                         "(0 = skip)")
    ap.add_argument("--solve-steps", type=int, default=20,
                    help="extra untimed-for-`value` pass: whole time steps including the device Krylov solves, reported "
                         "as `with_solves` (0 = skip; N = 1 only)")
    ap.add_argument("--frozen-state", action="store_true",
                    help="hold the fields at their initial state instead of replaying a recorded trajectory (the "
                         "round-1 measurement: phi_M is reset every step and the cell never fires)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the EMI matrix assembly after the ODE sweep instead of beside it (aux stream)")
    ap.add_argument("--variant", default="cg", choices=["cg", "dg"],
                    help="cg: the reference's formulation (CG on sub-meshes, the parity path); dg: the DG(P1) + interior "
                         "penalty variant of SURVEY.md section 8 f4 on the same mesh (assembly + ODE sweep + update, fields "
                         "held at the initial state: its linear solves are not on the device)")
    ap.add_argument("--no-dg", action="store_true", help="skip the short DG-variant measurement appended to the cg line")
    ap.add_argument("--knp-twice", action="store_true",
                    help="assemble A_knp twice per step as the reference does (p = a, knpWeakForm.py:319)")
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one rank per GPU)")
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

    from knpemi import _lib as L
    from knpemi.stepper import DeviceStepper

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
    case = Case(args.workload, rank, world, args.scaling)
    s = case.s
    frozen = args.frozen_state
    def frozen_state():
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

    if frozen and case.family == "idealized":
        frozen_state()

    stepper = DeviceStepper((s.a_emi, s.p_emi, s.L_emi), (s.a_knp, s.p_knp, s.L_knp),
                            s.c, s.c_prev, s.phi, s.phi_M_prev, assemble_knp_twice=args.knp_twice,
                            overlap=not args.no_overlap, fuse_update=not args.frozen_state,
                            early_membrane=bool(os.environ.get("KNPEMI_EARLY_MEMBRANE")))
    if os.environ.get("KNPEMI_OVERLAP_THRESHOLD_US"):      # experiment: when the stepper gives up running the EMI assembly beside the sweep
        stepper.overlap_threshold_ms = float(os.environ["KNPEMI_OVERLAP_THRESHOLD_US"]) * 1e-3
    dp = stepper.dp
    for m, stim, loc in case.models:
        stepper.add_membrane_model(m, stim, loc)
    if case.source is not None:
        stepper.set_source(0, case.source)
    halo = getattr(s, "halo", None)
    if halo is not None:
        halo.attach(dp)
        halo.exchange_bulk()        # ghosts start from their owners' values
        halo.exchange_membrane()    # once: afterwards the ghost membrane dofs are integrated redundantly

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    lib = dp.lib
    n_traj = args.warmup + PROFILE_STEPS + args.steps + 8      # + the short pass that times the row kernels alone
    traj_its = None
    traj_fallback = None
    if not frozen and halo is not None:
        # The distributed Krylov solves (RCCL all-reduce and halo from inside the solver loop) are the one part of the
        # N > 1 path that could only be rehearsed over gloo here.  If they fail -- the same way on every rank, i.e.
        # before or outside a collective -- the measurement falls back to the frozen state instead of being lost, and
        # says so in `config.state`.
        try:
            halo.enable_solves()    # knpemi_solve_emi / knp solve the global systems
            if os.environ.get("KNPEMI_BENCH_TEST_FALLBACK"):      # rehearsal hook for the branch below
                raise RuntimeError("KNPEMI_BENCH_TEST_FALLBACK is set")
            phi_t, c_t, traj_its = record_trajectory(case, stepper, n_traj, torch, halo)
        except Exception as e:      # noqa: BLE001
            if case.family != "idealized":
                raise
            traj_fallback = f"{type(e).__name__}: {e}"[:300]
            frozen = True
            stepper.solve_emi = stepper.solve_knp = None
            frozen_state()
            stepper.reset()
            halo.exchange_bulk()
            halo.exchange_membrane()
    elif not frozen:
        phi_t, c_t, traj_its = record_trajectory(case, stepper, n_traj, torch, halo)
    if not frozen:
        stepper.reset()
        if halo is not None:
            halo.exchange_bulk()
            halo.exchange_membrane()
        if case.source is not None:
            stepper.set_source(0, case.source)
        cursor = [0]

        def paste_emi(d):
            L.check(d.lib.knpemi_set_solution(d.h, L.B_EMI, phi_t[cursor[0]].data_ptr(), 1))

        def paste_knp(d):
            L.check(d.lib.knpemi_set_solution(d.h, L.B_KNP, c_t[cursor[0]].data_ptr(), 1))
            cursor[0] += 1
        stepper.solve_emi, stepper.solve_knp = paste_emi, paste_knp

    def ode_stats():
        nr = ns = nf = 0
        for m, _, _ in case.models:
            a, b, c = C.c_int64(), C.c_int64(), C.c_int32()
            lib.knpemi_ode_stats(dp.h, m._sub, m._model, C.byref(a), C.byref(b), C.byref(c))
            nr, ns, nf = nr + a.value, ns + b.value, nf + c.value
        return nr, ns, nf

    for _ in range(args.warmup):
        stepper.step(halo)
    sync()
    # untimed profiling pass: every kernel bracketed by HIP events -> per-kernel averages, dominant kernel
    L.check(lib.knpemi_profile(dp.h, (1 << len(L.KERNEL_NAMES)) - 1))
    for _ in range(PROFILE_STEPS):
        stepper.step(halo)
    sync()
    per_kernel = {}
    for kid, name in enumerate(L.KERNEL_NAMES):
        n, ms = C.c_int64(), C.c_double()
        L.check(lib.knpemi_profile_read(dp.h, kid, C.byref(n), C.byref(ms)))
        if n.value:
            per_kernel[name] = ms.value / n.value * 1e3 * (n.value / PROFILE_STEPS)   # us per step (all launches)
    rows = {k: v for k, v in per_kernel.items() if k in ("emi_rows_kernel", "knp_rows_kernel")}
    dominant = max(rows, key=rows.get) if rows else "emi_rows_kernel"
    dom_id = L.KERNEL_NAMES.index(dominant)
    # timed region: only the dominant row kernel stays bracketed by HIP events (it runs on the side stream).  The
    # membrane-facet kernel sits on the critical path, where an event pair costs the step ~10 us: its duration for
    # `roofline_membrane_facet_kernel` is the one of the profiling pass above.
    L.check(lib.knpemi_profile(dp.h, 1 << dom_id))
    # ... and only every 8th of its launches: at the small sizes the stepper runs everything on one stream and the
    # event pair would sit on the critical path of every step
    stride = int(os.environ.get("KNPEMI_BENCH_PROFILE_STRIDE", "8"))
    L.check(lib.knpemi_set_option(dp.h, L.OPT_PROFILE_STRIDE, stride))
    ode_stats()                                 # reset the counters
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stepper.step(halo)
    t_enqueue = time.perf_counter() - t0      # host side: all launches of the timed steps are enqueued
    sync()
    elapsed = time.perf_counter() - t0
    n_rhs, n_lsoda_steps, n_failed = ode_stats()
    n, ms = C.c_int64(), C.c_double()
    L.check(lib.knpemi_profile_read(dp.h, dom_id, C.byref(n), C.byref(ms)))
    dom_us = ms.value / max(n.value, 1) * 1e3
    mem_us = per_kernel.get("knp_membrane_kernel", 0.0)
    L.check(lib.knpemi_profile(dp.h, 0))
    L.check(lib.knpemi_set_option(dp.h, L.OPT_PROFILE_STRIDE, 1))
    if n_failed:
        raise SystemExit("LSODA failed on the device")
    # The row kernels by themselves: while the EMI assembly shares the chip with the ODE sweep its launches are
    # stretched by the sweep's waves.  A short untimed pass with everything on one stream gives the durations the
    # kernels reach alone (`roofline_row_kernels_alone`); the step time above is the overlapped schedule's.
    alone = {}
    overlapped = bool(stepper.overlap)
    if overlapped and not frozen and len(phi_t) >= args.warmup + PROFILE_STEPS + args.steps + 8:
        stepper.overlap = False
        L.check(lib.knpemi_profile(dp.h, (1 << L.KERNEL_NAMES.index("emi_rows_kernel")) | (1 << L.KERNEL_NAMES.index("knp_rows_kernel"))))
        for _ in range(8):
            stepper.step(halo)
        sync()
        for name in ("emi_rows_kernel", "knp_rows_kernel"):
            n, ms = C.c_int64(), C.c_double()
            L.check(lib.knpemi_profile_read(dp.h, L.KERNEL_NAMES.index(name), C.byref(n), C.byref(ms)))
            if n.value:
                alone[name] = ms.value / n.value * 1e3
        L.check(lib.knpemi_profile(dp.h, 0))
        stepper.overlap = True
    if dist is not None:
        red_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    owned = getattr(s, "owned_dofs", None)
    dofs_local = 3 * (owned if owned is not None else int(dp.n_vert.sum()))
    if dist is not None:
        tot = torch.tensor([dofs_local], dtype=torch.int64, device=red_dev)
        dist.all_reduce(tot)
        dofs_total = int(tot.item())
    else:
        dofs_total = dofs_local
    ms_per_step = elapsed / args.steps * 1e3
    value = dofs_total / (elapsed / args.steps)

    if rank == 0:
        survey_b, design_b, sizes = algorithmic_bytes(case, dp)
        n_ode_dofs = sum(m.nodes for m, _, _ in case.models)

        def roof(kernel, us):
            # HBM bytes per launch measured with rocprofv3 PMC counters on THIS build, if they were collected
            # (tools/collect_traffic.sh -> profiles/r02_traffic.json); PMC collection needs its own profiler passes
            traffic = None
            try:
                tr = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))[args.workload][kernel]
                if world == 1:
                    traffic = (2.0 * tr["FETCH_SIZE_KiB"] + tr["WRITE_SIZE_KiB"]) * 1024.0
            except (OSError, KeyError, ValueError):
                pass
            ach = survey_b[kernel] / (us * 1e-6) / 1e9
            return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": survey_b[kernel], "avg_launch_us": us,
                    "bytes_this_design_touches": design_b[kernel],
                    "frac_of_design_bytes": design_b[kernel] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    # a launch cannot be shorter than the launch floor: the fraction of peak a perfect kernel would
                    # reach at this size (1.0 = large enough for HBM to be the bound)
                    "launch_floor_us": LAUNCH_FLOOR_US,
                    "frac_at_launch_floor": min(1.0, survey_b[kernel] / (LAUNCH_FLOOR_US * 1e-6) / 1e9 / HBM_PEAK_GBS)}
        out = {
            "metric": "assembled dofs/s (volume + membrane-facet assembly + membrane ODE sweep) per timestep; "
                      "3D idealized mesh, fp64",
            "value": value, "unit": "dofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "host_enqueue_ms_per_step": t_enqueue / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: make_mesh_3D geometry r={case.r}, {case.kind}, "
                                   f"{sizes['nc']} cells/GPU, {sizes['N']} sub-mesh vertices/GPU in "
                                   f"{len(s.subdomain_list)} sub-domains, {n_ode_dofs} membrane ODE dofs/GPU, 3 ions "
                                   f"(K, Cl, Na eliminated), {case.describe}",
                       "dofs_per_step": dofs_total, "A_knp_assemblies_per_step": 2 if args.knp_twice else 1,
                       "emi_matrix_beside_ode_sweep": overlapped,
                       "update_fused_into_knp_write_back": bool(stepper.fuse_update),
                       "state": (("fields frozen at the initial state (phi_M reset every step)"
                                  + (f"; FALLBACK: the trajectory pass with distributed solves raised {traj_fallback}"
                                     if traj_fallback else "")) if frozen else
                                 f"recorded trajectory of the first {n_traj} time steps from t = 0 (device Krylov solves, "
                                 f"untimed); the timed steps replay it, pasting each recorded solution where the solve "
                                 f"writes it"),
                       "partition": "x-slabs" if world > 1 else "none"},
            "roofline": roof(dominant, dom_us),
            "roofline_membrane_facet_kernel": roof("knp_membrane_kernel", mem_us) if mem_us > 0 else None,
            "roofline_row_kernels_alone": {k: roof(k, v) for k, v in alone.items()} or None,
            "kernels_us_per_step": per_kernel,
            "ode": {"rhs_evals_per_dof_per_step": n_rhs / max(1, n_ode_dofs) / args.steps,
                    "lsoda_steps_per_dof_per_step": n_lsoda_steps / max(1, n_ode_dofs) / args.steps,
                    "kernel_us_per_step": per_kernel.get("ode_step_kernel"),
                    "share_of_step": per_kernel.get("ode_step_kernel", 0.0) / (ms_per_step * 1e3)},
        }
        if halo is not None:
            out["config"]["halo"] = halo.mode
        if traj_its is not None:
            out["config"]["trajectory_iterations_avg"] = {k: sum(v) / max(1, len(v)) for k, v in traj_its.items()}
    ws = None
    if args.solve_steps > 0 and not frozen:      # collective at N > 1: every rank takes part
        stepper.solve_emi = stepper.solve_knp = None
        ws = with_solves(case, stepper, args.solve_steps, torch, halo)
    if rank == 0:
        if ws is not None:
            out["with_solves"] = ws
        if args.cpu_steps > 0 and world == 1 and case.family == "idealized":
            avail = len(os.sched_getaffinity(0))
            # bounded sample: --cpu-steps refers to the config-2 size and shrinks with the problem size
            n_all = max(3, int(round(args.cpu_steps * min(1.0, 79251.0 / dofs_total))))
            n1 = max(2, int(0.4 * n_all))
            quiet = io.StringIO()
            with contextlib.redirect_stdout(quiet):
                t_one, a_one, o_one, _ = cpu_baseline(s, n1, threads=1)     # before any OpenMP team exists
                # thread count: a fully subscribed host can be slower than a partly subscribed one (spinning OpenMP
                # team + the Python thread, CPU shares below the visible core count): probe and time the best
                cand = sorted({min(avail, c) for c in (4, 8, 16, 32, avail)})
                probes = {c: cpu_baseline(s, 3, threads=c)[0] for c in cand}
                cores = min(probes, key=probes.get)
                t_all, a_all, o_all, nrows = cpu_baseline(s, n_all, threads=cores)
            what = ("whole steps of the C++ port (oracle/knpemi_cpu.cpp, sequential ODEPACK restatement "
                    "oracle/lsoda_seq.h) on the same mesh from the initial state: EMI (A, P, b) + KNP (A, b) assembly "
                    "and update {a:.0f} ms/step, LSODA sweep over all {n} membrane dofs {o:.0f} ms/step; "
                    "the reference itself cannot run here")
            out["cpu_baseline"] = {
                "value": dofs_total / t_all, "unit": "dofs/s", "cores": cores, "kind": "port",
                "sample": f"{n_all} " + what.format(a=a_all * 1e3, o=o_all * 1e3, n=nrows)
                          + f"; OpenMP over cells / facets / membrane dofs, {cores} threads (fastest of "
                            f"{cand} on the {avail} cores visible to this process)"}
            out["cpu_baseline_1core"] = {
                "value": dofs_total / t_one, "unit": "dofs/s", "cores": 1, "kind": "port",
                "sample": f"{n1} " + what.format(a=a_one * 1e3, o=o_one * 1e3, n=nrows)
                          + "; one thread, as the reference's serial run"}
        if world == 1 and not args.no_dg and case.family == "idealized" and case.kind == "tet":
            del stepper, s, case          # free the CG problem first
            dg = run_dg(args, torch, steps=10, warmup=2, cpu=False)
            out["dg_variant"] = {k: dg[k] for k in ("value", "unit", "ms_per_step", "config", "roofline",
                                                     "roofline_potential_kernel", "kernels_us_per_step")}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
