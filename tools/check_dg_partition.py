"""Cell-partitioned DG variant on the GPU (rehearsal with gloo, 2+ ranks on one card, or RCCL on several):

    torchrun --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 tools/check_dg_partition.py

Every rank assembles its slab (own hexahedron layers + one ghost layer) for K steps -- membrane ODE sweep over the facet
nodes, potential system, concentration systems, update, ghost refresh -- and compares the rows of its OWNED cells
(matrix values, right-hand sides) and its owned membrane nodes with a single-rank run of the whole box, bit for bit.
The fields evolve between the steps through a cheap stand-in for the solves that every rank can apply locally:
phi <- phi + eps * b_emi / max|b_emi| is not partition-independent, so instead c_new = c * (1 + 1e-3 sin(step + x)).
"""
import argparse, contextlib, io, os, sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-fenics-x_amd"))
sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))


def setup(dp):
    from knpemi import _lib as L
    ions = [dict(name="Na", z=1.0, D=[1.33e-9] * 2), dict(name="K", z=1.0, D=[1.96e-9] * 2), dict(name="Cl", z=-1.0, D=[2.03e-9] * 2)]
    dp.set_params(dict(dt=1e-4, F=96485.0, psi=96485.0 / (8.314 * 300.0), C_M=0.02), ions)
    ins = (dp.cell_sub > 0)[:, None] * np.ones((1, dp.nv), bool)
    w = np.sin(2e5 * dp.X[:, :, 0]) * np.cos(3e6 * dp.X[:, :, 1]) * np.cos(2e6 * dp.X[:, :, 2])
    for k, (e, i) in enumerate(((100.0, 12.0), (4.0, 125.0), (104.0, 137.0))):
        dp.set_concentration(k, np.where(ins, i, e) * (1.0 + 1e-3 * w * (1 + k)))
    dp.set_potential(np.where(ins, -0.0744, 0.0) + 1e-3 * w)
    dp.set_membrane_potential(np.full((dp.nmf, dp.nf), -0.0744))
    import mm_hh
    pi = mm_hh.parameter_indices
    prow = np.asarray(mm_hh.init_parameter_values(), float)
    prow[pi("Cm")] = 0.02
    prow[pi("z_Na")], prow[pi("z_K")], prow[pi("z_Cl")], prow[pi("psi")] = 1.0, 1.0, -1.0, 96485.0 / (8.314 * 300.0)
    names = ("Na", "K", "Cl")
    dp.ode_bind(L.MODEL_HH_SI, np.asarray(mm_hh.init_state_values(), float), prow,
                sum(([pi(f"{n}_e"), pi(f"{n}_i"), pi(f"I_ch_{n}")] for n in names), []), mm_hh.state_indices("V"))


def run(dp, K, slab=None):
    for k in range(K):
        dp.ode_step(k * 1e-4, 1e-4, set_v=k > 0)
        dp.assemble_emi()
        dp.assemble_knp()
        c_new = np.stack([dp.get_concentration(j) * (1.0 + 1e-3 * np.sin(k + 1e5 * dp.X[:, :, 0])) for j in range(2)])
        dp.set_potential(dp.get_potential() * (1.0 + 1e-3 * np.cos(k + 1e5 * dp.X[:, :, 0])))
        dp.update(c_new)
        if slab is not None:
            # poison the ghost cells first: only a correct halo brings back what the owned rows next to the cut need
            ghost = ~slab.owned_cells
            for j in range(3):
                c = dp.get_concentration(j)
                c[ghost] = 1e3 + j
                dp.set_concentration(j, c)
            p = dp.get_potential()
            p[ghost] = 7.0
            dp.set_potential(p)
            slab.exchange()
    dp.ode_stats()
    return dict(A=[dp.matrix(w).data.copy() for w in range(3)], b=[dp.rhs(w) for w in range(3)], indptr=dp.indptr.copy(),
                phiM=dp.get_membrane_potential(), c=[dp.get_concentration(j) for j in range(3)], ode=dp.ode_tables()[0],
                Ich=[dp.get_channel_current(j) for j in range(3)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-r", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cell", default="tetrahedron", choices=["tetrahedron", "hexahedron"])
    ap.add_argument("--solves", action="store_true",
                    help="after the steps: the two systems solved on the partition (knpemi_dg_set_distributed) against the "
                         "single-rank solves of the whole box")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("KNPEMI_BENCH_BACKEND", "gloo")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend)
    from knpemi.dg import DGProblem, DGSlab
    from knpemi.fem.idealized import make_mesh_3D
    dev = torch.cuda.current_device()
    slab = DGSlab(a.r, 2, rank, world, device=dev, cell=a.cell)
    setup(slab.dp)
    slab.attach()
    slab.exchange()
    mine = run(slab.dp, a.steps, slab)
    ok = True
    if True:   # every rank checks its own part against the whole box (each builds it: small mesh)
        with contextlib.redirect_stdout(io.StringIO()):
            mesh, ct, ft = make_mesh_3D(a.r, a.cell, l=2)
        g = DGProblem(mesh, ct, ft, [0, 1], [1], device=dev)
        setup(g)
        ref = run(g, a.steps)
        dp = slab.dp
        nxg = slab.nx
        lo = max(slab.a - 1, 0)
        nxl = min(slab.b + 1, nxg) - lo
        # local cell -> global cell (x fastest; 6 tetrahedra per hexahedron on the simplicial mesh)
        lc = np.arange(dp.n_cells)
        cph = slab.cells_per_hex
        hx, t = lc // cph, lc % cph
        gcell = ((hx // nxl) * nxg + lo + hx % nxl) * cph + t
        own = np.flatnonzero(slab.owned_cells)
        for w in range(3):
            for c in own[:: max(1, len(own) // 4000)]:
                for i in range(dp.nv):
                    rl, rg = c * dp.nv + i, gcell[c] * dp.nv + i
                    vl = mine["A"][w][mine["indptr"][rl]:mine["indptr"][rl + 1]]
                    vg = ref["A"][w][ref["indptr"][rg]:ref["indptr"][rg + 1]]
                    if not np.array_equal(vl, vg):
                        ok = False
            gl = (gcell[own][:, None] * dp.nv + np.arange(dp.nv)).ravel()
            ll = (own[:, None] * dp.nv + np.arange(dp.nv)).ravel()
            if not np.array_equal(mine["b"][w][ll], ref["b"][w][gl]):
                ok = False
        for j in range(3):
            if not np.array_equal(mine["c"][j][own], ref["c"][j][gcell[own]]):
                ok = False
        # membrane nodes of owned ECS cells: match by coordinates
        key = lambda X: np.round(X.reshape(len(X), -1) * 1e12).astype(np.int64)
        gmap = {tuple(k): i for i, k in enumerate(key(g.XM))}
        e, _ = dp.membrane_dofs()
        mown = slab.owned_cells[e[:, 0] // dp.nv]
        gi = np.array([gmap[tuple(k)] for k in key(dp.XM[mown])], np.int64)
        for name in ("phiM",):
            if not np.array_equal(mine[name][mown], ref[name][gi]):
                ok = False
        for j in range(3):
            if not np.array_equal(mine["Ich"][j][mown], ref["Ich"][j][gi]):
                ok = False
        nfl, nfg = dp.nf, g.nf
        st_l = mine["ode"].reshape(dp.nmf, nfl, -1)[mown]
        st_g = ref["ode"].reshape(g.nmf, nfg, -1)[gi]
        if not np.array_equal(st_l, st_g):
            ok = False
    sol_msg = ""
    if a.solves:
        # the systems of the state both runs have reached: solved GLOBALLY on the partition (every rank takes part); the
        # owned cells' parts, put together, must satisfy the single-rank systems of the whole box to the solver's tolerance
        rt_e, rt_k = 1e-8, 1e-11
        slab.enable_solves()
        dp = slab.dp
        dp.assemble_emi()
        it_e, _ = dp.solve_emi(rtol=rt_e)
        phi_l = dp.get_potential()
        dp.assemble_knp()
        it_k, _ = dp.solve_knp(rtol=rt_k)
        c_l = dp.solution().reshape(2, dp.n_cells, dp.nv)
        if slab._hook_error is not None:
            raise slab._hook_error
        own = np.flatnonzero(slab.owned_cells)
        parts = [None] * world
        dist.all_gather_object(parts, (gcell[own], phi_l[own], c_l[:, own]))
        phi_x = np.zeros((g.n_cells, g.nv))
        c_x = np.zeros((2, g.n_cells, g.nv))
        for gc, ph, cc in parts:
            phi_x[gc] = ph
            c_x[:, gc] = cc
        g.assemble_emi()
        A, bb = g.matrix(0), g.rhs(0)
        bb = bb - bb.mean()
        ge, _ = g.solve_emi(rtol=rt_e)              # (the single-rank iteration count, for the record)
        r_e = np.linalg.norm(A @ phi_x.ravel() - bb) / np.linalg.norm(bb)
        g.set_potential(phi_x)                      # the concentration systems are assembled with the potential just solved
        g.assemble_knp()
        r_k = max(np.linalg.norm(g.matrix(1 + k) @ c_x[k].ravel() - g.rhs(1 + k)) / np.linalg.norm(g.rhs(1 + k)) for k in range(2))
        sol_msg = (f"distributed solves: residuals of the assembled solution in the single-rank systems {r_e:.2e} (potential), "
                   f"{r_k:.2e} (concentrations); {it_e} CG / {it_k} BiCGStab iterations on the partition, {ge} CG on one rank")
        if not (r_e < 20 * rt_e and r_k < 100 * rt_k and abs(phi_x.mean()) < 1e-10 * np.abs(phi_x).max()):
            ok = False
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"mode {slab.mode}; owned cells {int(slab.owned_cells.sum())} of {slab.dp.n_cells} local")
        if sol_msg:
            print(sol_msg)
        print("DG PARTITION OK" if int(flag.item()) else "DG PARTITION MISMATCH")
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
