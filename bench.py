#!/usr/bin/env python3
"""bench.py -- assembled dofs/s of the KNP-EMI per-time-step hot path on MI355X.

One "step" = one pass of the hot path over one synthetic field state, device-resident
(SURVEY.md section 8d / BASELINE.md section 2): fused ODE launch (trace refresh + LSODA sweep +
copy-back), EMI assembly (A, P, b in one pass), KNP assembly (A once, b incl. the membrane
kernel), end-of-step update.  Krylov solves and file output are excluded.  At N > 1 the mesh is
N times longer (weak scaling, x-slabs) and every step also exchanges the ghost-dof halo of the bulk fields
(stream-ordered RCCL point-to-point, no host synchronisation; ghost membrane dofs are integrated redundantly).

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
for p in (os.path.join(ROOT, "knp-emi-fenics-x_amd"), os.path.join(ROOT, "examples", "idealized_geometries")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)

WORKLOADS = {
    # name: (mesh kind, resolution factor) -- geometry of make_mesh_3D.py, 6 tets per hexahedron
    "config2": ("tet", 1),     # BASELINE.json configs[1]: 124 416 tets, 79 251 dofs/step
    "config2h": ("hex", 2),    # reference-faithful Q1 hexahedra, 165 888 cells
    "config3": ("tet", 2),     # 995 328 tets on ONE GPU (the 8-GPU mesh of configs[2])
    "r3": ("tet", 3),          # 7.96 M tets
    "2d": ("2d", 3),
}


def algorithmic_bytes(s, dp):
    """compulsory bytes per launch, every array counted once."""
    import numpy as np
    from knpemi import _lib as L
    nv = s.mesh.cells.shape[1]
    gdim = s.mesh.gdim
    nc = int(dp.n_cell.sum())
    N = int(dp.n_vert.sum())
    nF = int(dp.n_facet.sum())
    nf = dp.flat["facet_e"][1].shape[1] if nF else 0
    NQ = int(dp.n_q.sum())
    nnz = dp._pattern(L.A_EMI)[1]
    nnzL = dp._pattern(L.A_KNP)[1] // 2
    idx = nc * (4 * nv + 4 * nv * nv)          # dofmap + scatter-slot map
    geo = 8 * gdim * N
    gam_idx = nF * (16 + 4 * (2 * nf) ** 2 + 2 * nf * 4)
    out = {
        # A and P (two matrices, one pass), b_emi, 3 coefficient fields, membrane coupling + Robin RHS
        "emi_rows_kernel": idx + geo + 8 * N * 3 + 2 * 8 * nnz + 8 * N + gam_idx + 8 * NQ + 8 * 2 * NQ,
        # two ion blocks, phi + 2 c_prev coefficients, 2 RHS vectors
        "knp_rows_kernel": idx + geo + 8 * N * 3 + 2 * 8 * nnzL + 2 * 8 * N,
        # pairs + dofs, ~12 gathered coefficient fields on the membrane, RMW of 2 RHS entries per side
        "knp_membrane_kernel": nF * (16 + 2 * nf * 4) + 8 * NQ * 12 + 2 * 8 * 2 * NQ * 2,
        "update_pde_kernel": 8 * N * (2 * 2 + 3) + 8 * NQ * 3,
    }
    for m in s.mem_models:
        ns, npar = m['ode'].states.shape[1], m['ode'].parameters.shape[1]
        out["ode_step_kernel"] = out.get("ode_step_kernel", 0) + 2 * 8 * NQ * (ns + npar) + 8 * NQ * 6 + 8 * NQ * 4
    return out


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


def with_solves(s, stepper, n_steps, torch):
    """Whole time steps of run_3D.py:345-368 on the device, Krylov solves included: the
    same stepper with `knpemi_solve_emi` (CG + AMG, rtol 1e-5) and `knpemi_solve_knp` (BiCGStab + AMG, rtol 1e-7)
    between the assemblies.  Not part of `value`."""
    from knpemi import _lib as L
    dp = stepper.dp
    its = {"emi": [], "knp": []}

    def solver(which, key, rtol, atol):
        def run(d):
            L.check(d.lib.knpemi_extrapolate_guess(d.h, which))     # start from 2 x_n - x_(n-1)
            its[key].append(d.solve(which, rtol, atol, 1000)[0])
        return run
    stepper.solve_emi = solver(L.B_EMI, "emi", 1e-5, 1e-40)
    stepper.solve_knp = solver(L.B_KNP, "knp", 1e-7, 2e-40)
    for _ in range(3):
        stepper.step()           # builds the AMG hierarchies
    torch.cuda.synchronize()
    its["emi"].clear()
    its["knp"].clear()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        stepper.step()
    dp.sync()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n_steps * 1e3
    stepper.solve_emi = stepper.solve_knp = None
    info = {k: dp.solver_info(w) for k, w in (("emi", L.B_EMI), ("knp", L.B_KNP))}
    return {"ms_per_step": ms, "steps": n_steps, "initial_guess": "2 x_n - x_(n-1) (knpemi_extrapolate_guess)",
            "emi": {"solver": "CG + SA-AMG V(1,1), rtol 1e-5", "iterations_avg": sum(its["emi"]) / n_steps, **info["emi"]},
            "knp": {"solver": "BiCGStab + SA-AMG V(1,1), rtol 1e-7", "iterations_avg": sum(its["knp"]) / n_steps,
                    **info["knp"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-steps", type=int, default=100,
                    help="CPU-port steps timed for cpu_baseline on all host cores; the 1-core leg runs 0.4 x as many "
                         "(0 = skip)")
    ap.add_argument("--solve-steps", type=int, default=20,
                    help="extra untimed-for-`value` pass: whole time steps including the device Krylov solves "
                         "(rtol 1e-5 / 1e-7 as run_3D.py:296-305), reported as `with_solves` (0 = skip; N = 1 only)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the EMI matrix assembly after the ODE sweep instead of beside it (aux stream)")
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
    from setup_problem import Setup

    kind, r = WORKLOADS[args.workload]
    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet):
        if world > 1:
            from knpemi.fem.partition import make_slab_problem
            s = make_slab_problem(kind, r, rank, world, g_syn=10.0)
        else:
            s = Setup(kind, r, g_syn=10.0)
    s.perturb(seed=12345 + rank)
    # synthetic "solution" state: c = c_prev, so the end-of-step update keeps the fields stationary
    for tag in s.subdomain_list:
        for k in range(2):
            s.c[tag][k].x.array[:] = s.c_prev[tag][k].x._a
        s.ion_list[-1][f'c_{tag}'].x.array[:] = -(1.0 / s.ion_list[-1]['z']) * sum(
            ion['z'] * f.x._a for ion, f in zip(s.ion_list[:-1], s.c_prev[tag]))
    # potentials: rest potential across the membrane plus a smooth perturbation
    L_x = s.mesh.x[:, 0].max() if world == 1 else s.global_length
    for tag in s.subdomain_list:
        x = s.subdomain_list[tag]['mesh_sub'].x
        s.phi[tag].x.array[:] = (-0.0744 if tag > 0 else 0.0) + 1e-3 * np.sin(2 * np.pi * x[:, 0] / L_x)

    stepper = DeviceStepper((s.a_emi, s.p_emi, s.L_emi), (s.a_knp, s.p_knp, s.L_knp),
                            s.c, s.c_prev, s.phi, s.phi_M_prev, assemble_knp_twice=args.knp_twice,
                            overlap=not args.no_overlap)
    dp = stepper.dp
    for mm in s.mem_models:
        stepper.add_membrane_model(mm['ode'], s.stim_params['stimulus'], s.stim_params['stimulus_locator'])
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
    for _ in range(args.warmup):
        stepper.step(halo)
    sync()
    # untimed profiling pass: every kernel bracketed by HIP events -> per-kernel averages, dominant kernel
    L.check(lib.knpemi_profile(dp.h, 0x1F))
    for _ in range(5):
        stepper.step(halo)
    sync()
    per_kernel = {}
    for kid, name in enumerate(L.KERNEL_NAMES):
        n, ms = C.c_int64(), C.c_double()
        L.check(lib.knpemi_profile_read(dp.h, kid, C.byref(n), C.byref(ms)))
        if n.value:
            per_kernel[name] = ms.value / n.value * 1e3   # us per launch
    asm = {k: v for k, v in per_kernel.items() if k != "ode_step_kernel"}
    dominant = max(asm, key=asm.get) if asm else "emi_rows_kernel"
    dom_id = L.KERNEL_NAMES.index(dominant)
    # timed region: only the dominant assembly kernel stays bracketed by HIP events
    L.check(lib.knpemi_profile(dp.h, 1 << dom_id))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stepper.step(halo)
    sync()
    elapsed = time.perf_counter() - t0
    n, ms = C.c_int64(), C.c_double()
    L.check(lib.knpemi_profile_read(dp.h, dom_id, C.byref(n), C.byref(ms)))
    dom_us = ms.value / max(n.value, 1) * 1e3
    L.check(lib.knpemi_profile(dp.h, 0))
    if stepper.ode_failures():
        raise SystemExit("LSODA failed on the device")
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
        bytes_alg = algorithmic_bytes(s, dp)
        achieved = bytes_alg[dominant] / (dom_us * 1e-6) / 1e9
        # HBM bytes per launch measured with rocprofv3 PMC counters (committed, profiles/r01_traffic.json):
        # PMC collection needs its own profiler passes and cannot run inside this process
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))[args.workload][dominant]
            if world == 1:
                traffic = (2.0 * tr["FETCH_SIZE_KiB"] + tr["WRITE_SIZE_KiB"]) * 1024.0
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "assembled dofs/s (volume + membrane-facet assembly + membrane ODE sweep) per timestep; "
                      "3D idealized mesh, fp64",
            "value": value, "unit": "dofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: make_mesh_3D geometry r={r}, {kind}, "
                                   f"{int(dp.n_cell.sum())} cells/GPU, {int(dp.n_vert.sum())} sub-mesh vertices/GPU, "
                                   f"{int(dp.n_q.sum())} membrane ODE dofs/GPU, 3 ions (K, Cl, Na eliminated), HH, "
                                   f"g_syn=10 for x<20um, dt=1e-4",
                       "dofs_per_step": dofs_total, "A_knp_assemblies_per_step": 2 if args.knp_twice else 1,
                       "emi_matrix_beside_ode_sweep": not args.no_overlap,
                       "partition": "x-slabs" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_alg[dominant], "avg_launch_us": dom_us},
            "kernels_us": per_kernel,
        }
        if args.solve_steps > 0 and world == 1:
            out["with_solves"] = with_solves(s, stepper, args.solve_steps, torch)
        if args.cpu_steps > 0 and world == 1:
            avail = len(os.sched_getaffinity(0))
            # bounded sample: --cpu-steps refers to the config-2 size and shrinks with the problem size
            n_all = max(3, int(round(args.cpu_steps * min(1.0, 79251.0 / dofs_total))))
            n1 = max(2, int(0.4 * n_all))
            with contextlib.redirect_stdout(quiet):
                t_one, a_one, o_one, _ = cpu_baseline(s, n1, threads=1)     # before any OpenMP team exists
                # thread count: a fully subscribed host can be slower than a partly subscribed one (spinning OpenMP
                # team + the Python thread, CPU shares below the visible core count): probe and time the best
                cand = sorted({min(avail, c) for c in (4, 8, 16, 32, avail)})
                probes = {c: cpu_baseline(s, 3, threads=c)[0] for c in cand}
                cores = min(probes, key=probes.get)
                t_all, a_all, o_all, nrows = cpu_baseline(s, n_all, threads=cores)
            what = ("whole steps of the C++ port (oracle/knpemi_cpu.cpp) on the same mesh: EMI (A, P, b) + KNP (A, b) "
                    "assembly and update {a:.0f} ms/step, LSODA sweep over all {n} membrane dofs {o:.0f} ms/step; "
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
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
