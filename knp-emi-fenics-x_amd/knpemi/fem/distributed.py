"""Cell partition of an arbitrary mesh over the GPUs of a node, id-keyed halos and distributed linear solves.

The reference lets DOLFINx partition whatever mesh it reads -- the structured boxes as well as the unstructured emimesh
tetrahedra of `examples/local_astrocyte_depolarization/run_stim_duration.py:127-134` -- with
`GhostMode.shared_facet`, updates ghosts with `Function.x.scatter_forward()` and runs its KSP solves on the mesh's
communicator (`src/knpemi/pdeSolver.py:24-35,74-78,99-110`).  Here:

* cells are partitioned by recursive coordinate bisection of their centroids (`rcb_partition`; `slab_partition`
  cuts whole layers along one axis), a vertex belongs to the lowest rank among its cells, and a rank keeps every cell
  that touches a vertex it owns -- so all rows of owned vertices are assembled locally (owner-computes, no matrix
  communication) with the cells in their global order (bit-identical rows);
* `VertexHalo` is the forward halo (owner -> ghost) of the dof fields, keyed by (sub-domain, global vertex id);
* `Halo.enable_solves` hands the library the ownership mask and the two communication steps of a distributed Krylov
  solve (all-reduce of the dot products, halo of a solver vector), both stream-ordered `torch.distributed` calls
  (backend "nccl" = RCCL over xGMI; "gloo" for CPU tests and one-card rehearsals).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .mesh import MeshTags, extract_submesh, match_facets


# ---------------------------------------------------------------------------------------------------------------
# cell partitions
# ---------------------------------------------------------------------------------------------------------------
def rcb_partition(centroids, world):
    """Recursive coordinate bisection: split along the longest axis of the bounding box, proportionally to the
    number of ranks on either side.  Deterministic (ties by cell index)."""
    part = np.zeros(centroids.shape[0], np.int32)

    def split(ids, r0, n):
        if n == 1:
            part[ids] = r0
            return
        x = centroids[ids]
        axis = int(np.argmax(x.max(axis=0) - x.min(axis=0)))
        order = ids[np.lexsort((ids, x[:, axis]))]
        nl = n // 2
        cut = (len(order) * nl) // n
        split(order[:cut], r0, nl)
        split(order[cut:], r0 + nl, n - nl)
    split(np.arange(centroids.shape[0]), 0, world)
    return part


def slab_partition(centroids, world, axis=0):
    """Whole layers of cells along `axis`, as equal as the layer structure allows (the x-slabs of the idealized
    boxes: 2 neighbours per rank, every axon cross-section on one GPU)."""
    x = np.round(centroids[:, axis] / (np.ptp(centroids[:, axis]) + 1e-300) * 1e9).astype(np.int64)
    layers, inv = np.unique(x, return_inverse=True)
    if len(layers) < world:
        raise ValueError("fewer cell layers than ranks")
    cuts = [(len(layers) * r) // world for r in range(world + 1)]
    layer_rank = np.zeros(len(layers), np.int32)
    for r in range(world):
        layer_rank[cuts[r]:cuts[r + 1]] = r
    return layer_rank[inv]


class LocalPart:
    """One rank's share of a partitioned mesh."""

    def __init__(self, mesh, ct, ft, part, rank, world):
        nvpc = mesh.cells.shape[1]
        owner = np.full(mesh.num_vertices, world, np.int32)
        np.minimum.at(owner, mesh.cells.ravel(), np.repeat(part.astype(np.int32), nvpc))
        own_v = owner == rank
        local_cells = np.flatnonzero(own_v[mesh.cells].any(axis=1)).astype(np.int32)
        marker = MeshTags(mesh, mesh.tdim, local_cells, np.ones(len(local_cells), np.int32))
        sub, emap, vmap, _, _ = extract_submesh(mesh, marker, 1)
        self.mesh = sub
        if getattr(mesh, "uniform_cell", None) is not None:
            sub.uniform_cell = mesh.uniform_cell          # cells of a uniform grid stay cells of that grid: the same bits on every rank
        self.rank, self.world = rank, world
        self.cell_global = local_cells
        self.vert_global = vmap.sub_to_parent.astype(np.int64)     # ascending
        self.vert_owner = owner[self.vert_global]
        self.ct = MeshTags(sub, sub.tdim, np.arange(sub.num_cells, dtype=np.int32), ct.dense()[local_cells])
        self.ct.name = getattr(ct, "name", "cell_marker")
        # facet tags: those of the matching global facets, except on facets that are interior globally but lie on
        # the boundary of the local mesh (no vertex of such a facet is owned here, so no owned row needs it -- and a
        # membrane facet must see both of its cells)
        pf = match_facets(mesh, sub, vmap)
        dense = ft.dense(fill=-1)
        tag = np.where(pf >= 0, dense[np.clip(pf, 0, None)], -1)
        sub._build_facets()
        mesh._build_facets()
        cut = (sub._facets["counts"] == 1) & (pf >= 0) & (mesh._facets["counts"][np.clip(pf, 0, None)] == 2)
        tag[cut] = -1
        keep = np.flatnonzero(tag >= 0).astype(np.int32)
        self.ft = MeshTags(sub, sub.tdim - 1, keep, tag[keep])
        self.ft.name = getattr(ft, "name", "facet_marker")


# ---------------------------------------------------------------------------------------------------------------
# halos
# ---------------------------------------------------------------------------------------------------------------
class Halo:
    """Forward (owner -> ghost) halo of one rank.  `plans[kind]` (kind in {'bulk', 'mem'}) maps a label to
    dict(nb=neighbour rank, send=index array, recv=index array) in the global device numbering of the DeviceProblem
    (vertex ids across sub-meshes, Q-dof ids)."""

    supports_solves = False

    def __init__(self):
        self.plans = None
        self.dp = None
        self._dev = None
        self.mode = "not attached"

    # -- host exchange (CPU tests, Vector.scatter_forward): `arr` = [n_global_ids, width] array ------------------
    def forward_host_array(self, kind, arr, dist):
        import torch
        ops, bufs = [], []
        for pl in self.plans[kind].values():
            sb = torch.from_numpy(np.ascontiguousarray(arr[pl["send"]]))
            rb = torch.empty((len(pl["recv"]),) + arr.shape[1:], dtype=torch.float64)
            ops += [dist.P2POp(dist.isend, sb, pl["nb"]), dist.P2POp(dist.irecv, rb, pl["nb"])]
            bufs.append((pl, rb, sb))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for pl, rb, _ in bufs:
            arr[pl["recv"]] = rb.numpy()

    # -- device exchange ---------------------------------------------------------------------------------------
    def attach(self, dp):
        import torch
        import torch.distributed as dist
        from .. import _lib as L
        self.dp, self.L, self.torch, self.dist = dp, L, torch, dist
        dev = torch.device("cuda", dp.device)
        self._device = dev
        n_slots = int(dp.n_models.sum())
        self.width = {"bulk": int(dp.lib.knpemi_halo_width(dp.h, 0)), "mem": int(dp.lib.knpemi_halo_width(dp.h, 1))}
        # The exchange mode is chosen ONCE, here, and agreed on by all ranks; a communication error during a run
        # propagates (the rank exits non-zero) instead of switching modes under a half-posted batch.
        #   "stream-ordered RCCL": pack kernel -> send/recv -> unpack kernel on the library's stream, no host sync
        #   "host-synchronised RCCL": KNPEMI_HALO_SYNC=1, or torch cannot wrap the library's stream
        #   "gloo host-staged": single-GPU rehearsals and CPU tests
        stream_ordered = os.environ.get("KNPEMI_HALO_SYNC") is None
        try:
            self._ext = torch.cuda.ExternalStream(dp.lib.knpemi_stream(dp.h), device=dev)
        except (RuntimeError, TypeError):
            self._ext, stream_ordered = None, False
        self._native = False
        if dist.get_backend() == "gloo":
            self.mode = "gloo host-staged"
        else:
            # first choice: RCCL called by the library itself (knpemi_comm_*: pack -> ncclSend/ncclRecv group -> unpack
            # on the library's stream, a few microseconds of host time per exchange instead of ~50 through
            # torch.distributed's Python layer).  All ranks must agree, so every step of the set-up is voted on.
            # KNPEMI_HALO_SYNC=1 asks for the host-synchronised exchange, which only the torch transport has
            native = os.environ.get("KNPEMI_HALO_TORCH") is None and os.environ.get("KNPEMI_HALO_SYNC") is None
            if native:
                native = self._native_comm_init(dp, dev)
            if native:
                self.mode = "library RCCL (knpemi_comm_sendrecv)"
                self._native = True
            else:
                flag = torch.tensor([1 if stream_ordered else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                self.mode = "stream-ordered RCCL" if int(flag.item()) else "host-synchronised RCCL"
        self._stream_ordered = self.mode == "stream-ordered RCCL"
        self._dev = {}
        for kind in ("bulk", "mem"):
            self._dev[kind] = self._device_plan(self.plans[kind], self.width[kind])

    def _vote(self, ok, dev):
        flag = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def _native_comm_init(self, dp, dev):
        """Create the library's own RCCL communicator: rank 0 draws the unique id, torch.distributed carries it."""
        dist, L = self.dist, self.L
        rank, world = dist.get_rank(), dist.get_world_size()
        buf = C.create_string_buffer(128)
        ok = True
        if rank == 0:
            ok = dp.lib.knpemi_comm_unique_id(buf, 128) == 0
        box = [buf.raw if ok else None]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            return False
        rc = dp.lib.knpemi_comm_init(dp.h, rank, world, box[0], 128)
        return self._vote(rc == 0, dev)

    def _device_plan(self, plan, w, index_map=None):
        """One packed buffer per direction: all neighbours' entries are packed / unpacked by a single kernel launch,
        the point-to-point operations work on slices of it."""
        torch, dev = self.torch, self._device
        sides = list(plan.values())
        if not sides:
            return None
        f = (lambda a: a) if index_map is None else index_map
        send_idx = np.concatenate([f(pl["send"]) for pl in sides]).astype(np.int32)
        recv_idx = np.concatenate([f(pl["recv"]) for pl in sides]).astype(np.int32)
        m = 1 if index_map is None else len(f(np.zeros(1, np.int64)))
        d = dict(send_idx=torch.from_numpy(send_idx).to(dev), recv_idx=torch.from_numpy(recv_idx).to(dev),
                 send_buf=torch.empty(len(send_idx) * w, dtype=torch.float64, device=dev),
                 recv_buf=torch.empty(len(recv_idx) * w, dtype=torch.float64, device=dev), parts=[])
        so = ro = 0
        for pl in sides:
            ns, nr = len(pl["send"]) * w * m, len(pl["recv"]) * w * m
            d["parts"].append((pl["nb"], slice(so, so + ns), slice(ro, ro + nr)))
            so, ro = so + ns, ro + nr
        # the same parts as flat arrays for knpemi_comm_sendrecv
        d["peer"] = np.array([p[0] for p in d["parts"]], np.int32)
        for key, idx, attr in (("send_off", 1, "start"), ("recv_off", 2, "start")):
            d[key] = np.array([getattr(p[idx], attr) for p in d["parts"]], np.int64)
        d["send_cnt"] = np.array([p[1].stop - p[1].start for p in d["parts"]], np.int64)
        d["recv_cnt"] = np.array([p[2].stop - p[2].start for p in d["parts"]], np.int64)
        return d

    def _transfer(self, d):
        """send_buf -> neighbours -> recv_buf, ordered on the library's stream (see `mode`)."""
        dp, dist, torch = self.dp, self.dist, self.torch
        if getattr(self, "_native", False):
            i64 = C.POINTER(C.c_int64)
            self.L.check(dp.lib.knpemi_comm_sendrecv(
                dp.h, d["send_buf"].data_ptr(), d["recv_buf"].data_ptr(), len(d["peer"]), self.L.iptr(d["peer"]),
                d["send_off"].ctypes.data_as(i64), d["send_cnt"].ctypes.data_as(i64),
                d["recv_off"].ctypes.data_as(i64), d["recv_cnt"].ctypes.data_as(i64)))
        elif dist.get_backend() != "gloo":
            ops = []
            for nb, ss, rs in d["parts"]:
                ops += [dist.P2POp(dist.isend, d["send_buf"][ss], nb), dist.P2POp(dist.irecv, d["recv_buf"][rs], nb)]
            if self._stream_ordered:
                # RCCL: torch's current stream is the library's stream (ExternalStream), so the send/recv kernels are
                # ordered after the pack kernel and `wait()` orders the unpack kernel after them: no host
                # synchronisation anywhere in the exchange (tools/check_async_halo.py rehearses this with real RCCL)
                with torch.cuda.stream(self._ext):
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
            else:
                dp.sync()                        # packed data is complete before RCCL reads it
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                torch.cuda.current_stream().synchronize()
        else:                           # single-GPU rehearsal: stage through host memory
            dp.sync()
            sb = d["send_buf"].cpu()
            rb = torch.empty(d["recv_buf"].shape, dtype=torch.float64)
            ops = []
            for nb, ss, rs in d["parts"]:
                ops += [dist.P2POp(dist.isend, sb[ss], nb), dist.P2POp(dist.irecv, rb[rs], nb)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            d["recv_buf"].copy_(rb)
            torch.cuda.current_stream().synchronize()

    def _exchange(self, kind):
        L, dp = self.L, self.dp
        d = self._dev.get(kind)
        if d is None:
            return
        k = 0 if kind == "bulk" else 1
        L.check(dp.lib.knpemi_halo_pack(dp.h, k, d["send_idx"].data_ptr(), d["send_idx"].numel(),
                                        d["send_buf"].data_ptr()))
        self._transfer(d)
        L.check(dp.lib.knpemi_halo_unpack(dp.h, k, d["recv_idx"].data_ptr(), d["recv_idx"].numel(),
                                          d["recv_buf"].data_ptr()))

    def exchange_bulk(self):
        self._exchange("bulk")

    def exchange_membrane(self):
        self._exchange("mem")

    # -- distributed linear solves ------------------------------------------------------------------------------
    def owned_mask(self):
        """One byte per local vertex (device numbering): 1 = owned.  Every ghost is received from its owner."""
        n = int(self.dp.n_vert.sum())
        own = np.ones(n, np.uint8)
        for pl in self.plans["bulk"].values():
            own[pl["recv"]] = 0
        return own

    def enable_solves(self):
        """knpemi_solve_emi / knpemi_solve_knp on this handle become solves of the GLOBAL systems
        (knpemi_set_distributed): halo'd SpMV, all-reduced dot products, per-rank AMG (block Jacobi)."""
        L, dp, torch = self.L, self.dp, self.torch
        KS = dp.K - 1
        voff, nvs = dp.voff, dp.n_vert

        def knp_index(g):          # (vertex id) -> its KS entries in the block order [sub][ion][vertex]
            g = np.asarray(g, np.int64)
            s = np.searchsorted(voff[1:], g, side="right")
            base = KS * voff[s] + (g - voff[s])
            return np.concatenate([base + k * nvs[s] for k in range(KS)])
        self._vec = {L.B_EMI: self._device_plan(self.plans["bulk"], 1),
                     L.B_KNP: self._device_plan_knp(knp_index)}
        self._red = torch.zeros(8 + 64, dtype=torch.float64, device=self._device)    # 8 scalars + the coarse vector
        self._hook_error = None

        def allreduce(ctx, n):
            try:
                self._allreduce(n)
                return 0
            except Exception as exc:       # noqa: BLE001 -- must not propagate through the C frame
                self._hook_error = exc
                return -1

        def halo(ctx, vec, which):
            try:
                self._exchange_vector(vec, which)
                return 0
            except Exception as exc:       # noqa: BLE001
                self._hook_error = exc
                return -1
        own = np.ascontiguousarray(self.owned_mask())
        if getattr(self, "_native", False):
            # the solves stay inside the library: its own all-reduce and vector halo are the hooks, the handle the context
            i64 = C.POINTER(C.c_int64)
            for which, d in self._vec.items():
                if d is None:
                    d = dict(send_idx=torch.zeros(0, dtype=torch.int32, device=self._device),
                             recv_idx=torch.zeros(0, dtype=torch.int32, device=self._device),
                             send_buf=torch.zeros(1, dtype=torch.float64, device=self._device),
                             recv_buf=torch.zeros(1, dtype=torch.float64, device=self._device),
                             peer=np.zeros(0, np.int32), send_off=np.zeros(0, np.int64), send_cnt=np.zeros(0, np.int64),
                             recv_off=np.zeros(0, np.int64), recv_cnt=np.zeros(0, np.int64))
                    self._vec[which] = d
                L.check(dp.lib.knpemi_comm_set_vector_plan(
                    dp.h, which, d["send_idx"].data_ptr(), d["send_idx"].numel(), d["recv_idx"].data_ptr(),
                    d["recv_idx"].numel(), d["send_buf"].data_ptr(), d["recv_buf"].data_ptr(), len(d["peer"]),
                    L.iptr(d["peer"]), d["send_off"].ctypes.data_as(i64), d["send_cnt"].ctypes.data_as(i64),
                    d["recv_off"].ctypes.data_as(i64), d["recv_cnt"].ctypes.data_as(i64)))
            hooks = (C.cast(dp.lib.knpemi_comm_allreduce_hook, C.c_void_p), C.cast(dp.lib.knpemi_comm_halo_hook, C.c_void_p))
            L.check(dp.lib.knpemi_set_distributed(dp.h, own.ctypes.data_as(L.c_u8_p), self._red.data_ptr(),
                                                  hooks[0], hooks[1], dp.h))
        else:
            self._cb = (L.ALLREDUCE_FN(allreduce), L.HALO_FN(halo))    # keep the callbacks alive
            L.check(dp.lib.knpemi_set_distributed(dp.h, own.ctypes.data_as(L.c_u8_p), self._red.data_ptr(),
                                                  C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None))
        if os.environ.get("KNPEMI_NO_COARSE") is None and self.dist.get_world_size() * len(dp.n_vert) <= 64:
            L.check(dp.lib.knpemi_set_distributed_coarse(dp.h, self.dist.get_rank(), self.dist.get_world_size()))
        self.supports_solves = True

    def _device_plan_knp(self, knp_index):
        # per neighbour the KS blocks must stay together in the packed buffer: build the plan from mapped indices
        mapped = {key: dict(nb=pl["nb"], send=knp_index(pl["send"]), recv=knp_index(pl["recv"]))
                  for key, pl in self.plans["bulk"].items()}
        return self._device_plan(mapped, 1)

    def _allreduce(self, n):
        dist, torch = self.dist, self.torch
        if dist.get_backend() == "gloo":
            self.dp.sync()
            host = self._red[:n].cpu()
            dist.all_reduce(host)
            self._red[:n].copy_(host)
            torch.cuda.current_stream().synchronize()
        elif self._stream_ordered:
            with torch.cuda.stream(self._ext):
                dist.all_reduce(self._red[:n])
        else:
            self.dp.sync()
            dist.all_reduce(self._red[:n])
            torch.cuda.current_stream().synchronize()

    def _exchange_vector(self, vec, which):
        L, dp = self.L, self.dp
        d = self._vec[which]
        if d is None:
            return
        L.check(dp.lib.knpemi_vec_gather(dp.h, vec, d["send_idx"].data_ptr(), d["send_idx"].numel(),
                                         d["send_buf"].data_ptr()))
        self._transfer(d)
        L.check(dp.lib.knpemi_vec_scatter(dp.h, vec, d["recv_idx"].data_ptr(), d["recv_idx"].numel(),
                                          d["recv_buf"].data_ptr()))


class VertexHalo(Halo):
    """Halo of a general cell partition, keyed by (sub-domain index, global vertex id)."""

    def __init__(self, local: LocalPart, subdomain_list):
        super().__init__()
        self.local = local
        self.keys = {"bulk": {}, "mem": {}}       # kind -> {s: (global ids ascending, owners, device offset)}
        off = qoff = 0
        self.owned_dofs = 0
        for s, (tag, sd) in enumerate(subdomain_list.items()):
            pv = sd["mesh_sub"].parent_vertices
            self.keys["bulk"][s] = (local.vert_global[pv], local.vert_owner[pv], off)
            off += len(pv)
            self.owned_dofs += int((local.vert_owner[pv] == local.rank).sum())
            if tag > 0:
                qv = sd["mesh_mem"].parent_vertices
                self.keys["mem"][s] = (local.vert_global[qv], local.vert_owner[qv], qoff)
                qoff += len(qv)

    def build(self, gather_objects):
        """`gather_objects(obj) -> [obj of rank 0, ..., obj of rank world-1]`."""
        rank = self.local.rank
        needs = {}
        for kind, subs in self.keys.items():
            for s, (gid, owner, _) in subs.items():
                for r in np.unique(owner[owner != rank]):
                    needs.setdefault(int(r), {}).setdefault(kind, {})[s] = gid[owner == r]
        all_needs = gather_objects(needs)
        plans = {"bulk": {}, "mem": {}}
        for kind, subs in self.keys.items():
            nbs = set(r for r, nd in enumerate(all_needs) if rank in nd and kind in nd[rank])
            nbs |= set(r for r in needs if kind in needs[r])
            for nb in sorted(nbs):
                send, recv = [], []
                for s, (gid, owner, off) in sorted(subs.items()):
                    want = all_needs[nb].get(rank, {}).get(kind, {}).get(s)
                    if want is not None and len(want):
                        pos = np.searchsorted(gid, want)
                        if not (np.all(pos < len(gid)) and np.array_equal(gid[pos], want)
                                and np.all(owner[pos] == rank)):
                            raise RuntimeError("halo: a neighbour's ghost is not owned here")
                        send.append(pos + off)
                    mine = needs.get(nb, {}).get(kind, {}).get(s)
                    if mine is not None and len(mine):
                        recv.append(np.searchsorted(gid, mine) + off)
                cat = lambda v: np.concatenate(v).astype(np.int32) if v else np.zeros(0, np.int32)
                plans[kind][nb] = dict(nb=nb, send=cat(send), recv=cat(recv))
        self.plans = plans
        return plans


def make_partitioned_problem(kind, r, rank, world, g_syn=10.0, length=None, method="rcb", gather=None):
    """The rank-local driver set-up (examples/idealized_geometries/setup_problem.Setup) of a partitioned idealized
    box: global mesh `make_mesh_3D(r, kind, l=length)` (default length 2 * world: weak scaling), cells partitioned
    with `method` in {"rcb", "slab"}."""
    from setup_problem import Setup
    from .idealized import make_mesh_3D
    cell_type = {"tet": "tetrahedron", "hex": "hexahedron"}[kind]
    l = 2 * world if length is None else int(length)
    mesh, ct, ft = make_mesh_3D(r, cell_type, l=l)
    cent = mesh.x[mesh.cells].mean(axis=1)
    part = rcb_partition(cent, world) if method == "rcb" else slab_partition(cent, world)
    local = LocalPart(mesh, ct, ft, part, rank, world)
    s = Setup(kind, r, g_syn=g_syn, mesh_data=(local.mesh, local.ct, local.ft), build_forms=True)
    if gather is None:
        import torch.distributed as dist

        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
    s.halo = VertexHalo(local, s.subdomain_list)
    s.halo.build(gather)
    s.owned_dofs = s.halo.owned_dofs
    s.local = local
    s.global_length = l * 16e-6
    s.global_mesh = (mesh, ct, ft)
    return s


def make_partitioned_astro(cfg, rank, world, method="rcb", gather=None):
    """The rank-local set-up of the three-sub-domain driver (ECS + neuron with the HH model + glia with the Kir4.1 / pump
    model, pulsed ECS source: `examples/local_astrocyte_depolarization/run_stim_duration.Problem`) on a cell partition
    of its mesh -- what DOLFINx's MPI partition gives the reference when that driver runs under `mpirun`
    (run_stim_duration.py:127-134,168-211).  Three sub-domain halos, two membrane models (their ghost dofs integrated
    redundantly), the source term evaluated on the local ECS vertices."""
    import run_stim_duration as rsd
    mesh, ct, ft = rsd.read_mesh(cfg)
    cent = mesh.x[mesh.cells].mean(axis=1)
    part = rcb_partition(cent, world) if method == "rcb" else slab_partition(cent, world)
    local = LocalPart(mesh, ct, ft, part, rank, world)
    s = rsd.Problem(cfg, mesh_data=(local.mesh, local.ct, local.ft))
    if gather is None:
        import torch.distributed as dist

        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out
    s.halo = VertexHalo(local, s.subdomain_list)
    s.halo.build(gather)
    s.owned_dofs = s.halo.owned_dofs
    s.local = local
    s.global_length = float(mesh.x[:, 0].max())
    s.global_mesh = (mesh, ct, ft)
    return s
