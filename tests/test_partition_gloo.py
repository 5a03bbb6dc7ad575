"""world_size-2 CPU test (gloo) of the multi-GPU decomposition: halo plan and owner-computes rows."""
import contextlib
import io
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _field(x, k):
    return 50.0 + 10.0 * np.sin(2e5 * x[:, 0] + k) + 3e6 * x[:, 1] - 2e6 * x[:, 2] * (k + 1)


def _worker(rank, world, port, kind, out_dir):
    for p in ("knp-emi-fenics-x_amd", "oracle", "examples/idealized_geometries", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p))
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import knpemi_oracle as o
    from knpemi.fem import extract_submesh, make_mesh_3D
    from knpemi.fem.partition import build_halo, make_slab_layout_and_mesh
    cell = {"tet": "tetrahedron", "hex": "hexahedron"}[kind]
    lay, (mesh, ct, ft) = make_slab_layout_and_mesh(kind, 0, rank, world)
    s0, *_ = extract_submesh(mesh, ct, 0)
    s1, *_ = extract_submesh(mesh, ct, 1)
    g1, *_ = extract_submesh(mesh, ft, [1])
    subs = {0: dict(mesh_sub=s0), 1: dict(mesh_sub=s1, mesh_mem=g1)}

    def gather(obj):
        res = [None] * world
        dist.all_gather_object(res, obj)
        return res
    halo, owned = build_halo(lay, subs, gather)
    tot = gather(owned)
    # 1. ownership partitions the global dof set
    gm, gct, gft = make_mesh_3D(0, cell, l=2 * world)
    G0, *_ = extract_submesh(gm, gct, 0)
    G1, *_ = extract_submesh(gm, gct, 1)
    assert sum(tot) == G0.num_vertices + G1.num_vertices
    # 2. forward halo delivers owner values to ghosts (bulk: 4 fields, membrane: 4 fields)
    xs = np.concatenate([s0.x, s1.x])
    planes = np.concatenate([halo.sub_keys[0][0], halo.sub_keys[1][0]])
    is_owned = (planes >= lay.own_lo) & (planes <= lay.own_hi)
    ref = np.stack([_field(xs, k) for k in range(4)], axis=1)
    arr = np.where(is_owned[:, None], ref, np.nan)
    halo.forward_host_array("bulk", arr, dist)
    assert np.array_equal(arr, ref)
    qplanes = halo.q_keys[1][0]
    qref = np.stack([_field(g1.x, k) for k in range(4)], axis=1)
    qarr = np.where(((qplanes >= lay.own_lo) & (qplanes <= lay.own_hi))[:, None], qref, np.nan)
    halo.forward_host_array("mem", qarr, dist)
    assert np.array_equal(qarr, qref)
    # 3. owner-computes: rows of owned vertices assembled on the local mesh (one ghost cell layer)
    #    equal the rows of the global assembly
    params = dict(dt=1e-4, F=96485.0, psi=96485.0 / (8.314 * 300), C_M=0.02, C_phi=200.0)
    ions = [dict(name=n, z=z, D={0: D, 1: D}) for n, z, D in (("K", 1.0, 1.96e-9), ("Cl", -1.0, 2.03e-9), ("Na", 1.0, 1.33e-9))]

    def assemble(m, c, f):
        P = o.OracleProblem(m.x, m.cells, m.cell_type, c.dense(), m.facets[f.indices], f.values, {0: [], 1: [1]})
        xsub = {t: P.sub[t]["x"] for t in (0, 1)}
        c_all = {t: [_field(xsub[t], k) for k in range(3)] for t in (0, 1)}
        phi = {t: 1e-3 * _field(xsub[t], 5) for t in (0, 1)}
        phiM = {1: -0.07 + 1e-5 * _field(P.mem[1]["x"], 6)}
        mm = {1: [dict(tag=1, I_ch_k={n: 1e-3 * _field(P.mem[1]["x"], 7 + i) for i, n in enumerate(("K", "Cl", "Na"))})]}
        A, Pm, b = o.assemble_emi(P, params, ions, c_all, phiM, mm)
        Ak, bk = o.assemble_knp(P, params, ions, c_all, phi, phiM, mm, 1e-4)
        return P, A, b, Ak, bk
    Pl, Al, bl, Akl, bkl = assemble(mesh, ct, ft)
    Pg, Ag, bg, Akg, bkg = assemble(gm, gct, gft)
    # local sub-mesh vertex -> global sub-mesh vertex through the (plane, yz) key
    nxg = lay.nx
    l2g = {}
    for t in (0, 1):
        pl, yz = lay.key_of_local_vertex(Pl.sub[t]["pv"])
        gparent = pl + (nxg + 1) * yz
        l2g[t] = np.searchsorted(Pg.sub[t]["pv"], gparent)
        assert np.array_equal(Pg.sub[t]["pv"][l2g[t]], gparent)
    lmap = np.concatenate([l2g[0] + Pg.off[0], l2g[1] + Pg.off[1]])
    own_rows = np.flatnonzero(is_owned)
    Ag_rows = Ag[lmap[own_rows]]
    Al_rows = Al[own_rows]
    # compare entries: move local columns to global numbering
    import scipy.sparse as sp
    Tcol = sp.csr_matrix((np.ones(len(lmap)), (np.arange(len(lmap)), lmap)), shape=(len(lmap), Ag.shape[0]))
    diff = (Al_rows @ Tcol - Ag_rows).tocoo()
    scale = np.abs(Ag.data).max()
    assert (np.abs(diff.data).max() if diff.nnz else 0.0) < 1e-12 * scale
    assert np.abs(bl[own_rows] - bg[lmap[own_rows]]).max() < 1e-12 * np.abs(bg).max()
    # KNP block order (sub, ion): rows of ion k of sub t
    for t in (0, 1):
        for k in range(2):
            lo = 2 * Pl.off[t] + k * Pl.N[t]
            go = 2 * Pg.off[t] + k * Pg.N[t]
            pl_t = halo.sub_keys[t][0]
            ow = np.flatnonzero((pl_t >= lay.own_lo) & (pl_t <= lay.own_hi))
            assert np.abs(bkl[lo + ow] - bkg[go + l2g[t][ow]]).max() < 1e-12 * np.abs(bkg).max()
            d = Akl.diagonal()[lo + ow] - Akg.diagonal()[go + l2g[t][ow]]
            assert np.abs(d).max() < 1e-12 * np.abs(Akg.diagonal()).max()
    open(os.path.join(out_dir, f"ok_{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["tet", "hex"])
def test_slab_partition_world2_gloo(tmp_path, kind):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, kind, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_slab_partition_world8_gloo(tmp_path):
    """The driver's 8-GPU run cuts a box eight times as long into eight slabs (bench.py --gpus 8, weak scaling): the same
    checks -- ownership partitions the dofs, the forward halo delivers the owners' values, owned rows equal the rows of the
    global assembly -- with eight ranks, interior ranks having two neighbours."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(8, port, "tet", str(tmp_path)), nprocs=8, join=True)
    assert all((tmp_path / f"ok_{r}").exists() for r in range(8))


def _general_worker(rank, world, port, method, out_dir):
    """General cell partitioner on a vertex- and cell-shuffled tetrahedral mesh (no structure left for the partition
    or the halo to lean on): ownership partitions the dofs, the id-keyed halo delivers owner values, and the owned rows
    of the oracle assembly on the local mesh (owned cells + ghost layer) equal the rows of the global assembly."""
    for p in ("knp-emi-fenics-x_amd", "oracle", "examples/idealized_geometries", "tests"):
        sys.path.insert(0, os.path.join(ROOT, p))
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    import scipy.sparse as sp
    import knpemi_oracle as o
    from knpemi.fem import Mesh, MeshTags, extract_submesh, make_mesh_3D, match_facets
    from knpemi.fem.distributed import LocalPart, VertexHalo, rcb_partition, slab_partition
    m0, ct0, ft0 = make_mesh_3D(0, "tetrahedron", l=2)
    rng = np.random.default_rng(7)            # same shuffle on every rank
    vperm = rng.permutation(m0.num_vertices)   # new id of old vertex v
    cperm = rng.permutation(m0.num_cells)
    x = np.empty_like(m0.x)
    x[vperm] = m0.x
    gm = Mesh(x, vperm[m0.cells][cperm], m0.cell_type)
    gct = MeshTags(gm, gm.tdim, np.arange(gm.num_cells, dtype=np.int32), ct0.dense()[cperm])
    # facet tags by matching the shuffled facets with the original ones
    class _V:                                  # sub_to_parent of the identity "sub-mesh" gm -> m0 vertex ids
        sub_to_parent = np.argsort(vperm)
    pf = match_facets(m0, gm, _V)
    assert (pf >= 0).all()
    gft = MeshTags(gm, gm.tdim - 1, np.arange(gm.num_facets, dtype=np.int32), ft0.dense(fill=0)[pf])
    cent = gm.x[gm.cells].mean(axis=1)
    part = rcb_partition(cent, world) if method == "rcb" else slab_partition(cent, world)
    assert len(np.unique(part)) == world
    local = LocalPart(gm, gct, gft, part, rank, world)
    mesh, ct, ft = local.mesh, local.ct, local.ft
    s0, *_ = extract_submesh(mesh, ct, 0)
    s1, *_ = extract_submesh(mesh, ct, 1)
    g1, *_ = extract_submesh(mesh, ft, [1])
    subs = {0: dict(mesh_sub=s0), 1: dict(mesh_sub=s1, mesh_mem=g1)}

    def gather(obj):
        res = [None] * world
        dist.all_gather_object(res, obj)
        return res
    halo = VertexHalo(local, subs)
    halo.build(gather)
    G0, *_ = extract_submesh(gm, gct, 0)
    G1, *_ = extract_submesh(gm, gct, 1)
    GQ, *_ = extract_submesh(gm, gft, [1])
    # 1. ownership partitions the global dof sets (bulk and membrane)
    assert sum(gather(halo.owned_dofs)) == G0.num_vertices + G1.num_vertices
    own_q = int((halo.keys["mem"][1][1] == rank).sum())
    assert sum(gather(own_q)) == GQ.num_vertices
    # 2. the forward halo delivers owner values to every ghost
    xs = np.concatenate([s0.x, s1.x])
    is_owned = np.concatenate([halo.keys["bulk"][s][1] == rank for s in (0, 1)])
    ref = np.stack([_field(xs, k) for k in range(4)], axis=1)
    arr = np.where(is_owned[:, None], ref, np.nan)
    halo.forward_host_array("bulk", arr, dist)
    assert np.array_equal(arr, ref)
    qref = np.stack([_field(g1.x, k) for k in range(4)], axis=1)
    qarr = np.where((halo.keys["mem"][1][1] == rank)[:, None], qref, np.nan)
    halo.forward_host_array("mem", qarr, dist)
    assert np.array_equal(qarr, qref)
    # 3. owner-computes: owned rows assembled on the local mesh equal the rows of the global assembly
    params = dict(dt=1e-4, F=96485.0, psi=96485.0 / (8.314 * 300), C_M=0.02, C_phi=200.0)
    ions = [dict(name=n, z=z, D={0: D, 1: D}) for n, z, D in (("K", 1.0, 1.96e-9), ("Cl", -1.0, 2.03e-9), ("Na", 1.0, 1.33e-9))]

    def assemble(m, c, f):
        P = o.OracleProblem(m.x, m.cells, m.cell_type, c.dense(), m.facets[f.indices], f.values, {0: [], 1: [1]})
        xsub = {t: P.sub[t]["x"] for t in (0, 1)}
        c_all = {t: [_field(xsub[t], k) for k in range(3)] for t in (0, 1)}
        phi = {t: 1e-3 * _field(xsub[t], 5) for t in (0, 1)}
        phiM = {1: -0.07 + 1e-5 * _field(P.mem[1]["x"], 6)}
        mm = {1: [dict(tag=1, I_ch_k={n: 1e-3 * _field(P.mem[1]["x"], 7 + i) for i, n in enumerate(("K", "Cl", "Na"))})]}
        A, Pm, b = o.assemble_emi(P, params, ions, c_all, phiM, mm)
        Ak, bk = o.assemble_knp(P, params, ions, c_all, phi, phiM, mm, 1e-4)
        return P, A, b, Ak, bk
    Pl, Al, bl, Akl, bkl = assemble(mesh, ct, ft)
    Pg, Ag, bg, Akg, bkg = assemble(gm, gct, gft)
    l2g = {}
    for t in (0, 1):
        gid = local.vert_global[Pl.sub[t]["pv"]]
        l2g[t] = np.searchsorted(Pg.sub[t]["pv"], gid)
        assert np.array_equal(Pg.sub[t]["pv"][l2g[t]], gid)
    lmap = np.concatenate([l2g[0] + Pg.off[0], l2g[1] + Pg.off[1]])
    own_rows = np.flatnonzero(is_owned)
    Tcol = sp.csr_matrix((np.ones(len(lmap)), (np.arange(len(lmap)), lmap)), shape=(len(lmap), Ag.shape[0]))
    diff = (Al[own_rows] @ Tcol - Ag[lmap[own_rows]]).tocoo()
    assert (np.abs(diff.data).max() if diff.nnz else 0.0) < 1e-12 * np.abs(Ag.data).max()
    assert np.abs(bl[own_rows] - bg[lmap[own_rows]]).max() < 1e-12 * np.abs(bg).max()
    for t in (0, 1):
        ow = np.flatnonzero(halo.keys["bulk"][t][1] == rank)
        for k in range(2):
            lo = 2 * Pl.off[t] + k * Pl.N[t]
            go = 2 * Pg.off[t] + k * Pg.N[t]
            assert np.abs(bkl[lo + ow] - bkg[go + l2g[t][ow]]).max() < 1e-12 * np.abs(bkg).max()
            d = Akl.diagonal()[lo + ow] - Akg.diagonal()[go + l2g[t][ow]]
            assert np.abs(d).max() < 1e-12 * np.abs(Akg.diagonal()).max()
    open(os.path.join(out_dir, f"ok_{rank}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,method", [(2, "rcb"), (4, "rcb"), (3, "slab")])
def test_general_partition_on_shuffled_mesh_gloo(tmp_path, world, method):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_general_worker, args=(world, port, method, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok_{r}").exists() for r in range(world))
