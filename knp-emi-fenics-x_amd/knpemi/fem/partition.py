"""Multi-GPU decomposition of the idealized 3D box: x-slabs, one rank per GPU.

The reference is parallelised only through DOLFINx's MPI cell partition with
`GhostMode.shared_facet` and `Function.x.scatter_forward()` (`run_3D.py:117-121`,
`utils.py:100,199,204,254,293`).  Here every rank owns the vertex
planes of its slab, keeps one layer of ghost cells on its low-x side plus the ghost vertex
plane on its high-x side, and assembles all rows of its owned vertices locally
(owner-computes: no matrix communication).  The only exchange is the forward halo of the
bulk dof fields (c_prev, c_eliminated, phi) after each update, sent point-to-point to the two
x-neighbours with `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  The membrane dofs of the ghost layer are integrated redundantly (identical
bits on both ranks), so phi_M / I_ch need no halo; `exchange_membrane` remains for drivers that
skip ghost dofs.
"""
from __future__ import annotations

import os

import numpy as np

from .idealized import make_mesh_3D_slab


def slab_ranges(nx, world):
    """Cell x-index range [a, b) of every rank."""
    cuts = [(nx * r) // world for r in range(world + 1)]
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class SlabLayout:
    """Ownership bookkeeping of one rank's local mesh (pure numpy, no communication)."""

    def __init__(self, r, rank, world, l):
        self.rank, self.world, self.l = rank, world, l
        self.nx = l * 16 * 2 ** r
        self.ny = self.nz = 9 * 2 ** r
        self.a, self.b = slab_ranges(self.nx, world)[rank]
        self.lo = self.a - 1 if rank > 0 else self.a         # first local cell (ghost layer on the low side)
        self.nxl = self.b - self.lo                           # local cells in x
        # owned vertex planes (global plane index): a .. b-1, the last rank also owns plane nx
        self.own_lo, self.own_hi = self.a, (self.b if rank == world - 1 else self.b - 1)

    def plane_of_local_vertex(self, v):
        return self.lo + (v % (self.nxl + 1))

    def key_of_local_vertex(self, v):
        """(global plane, y-z index) -- identical on every rank that holds the vertex."""
        return self.plane_of_local_vertex(v), v // (self.nxl + 1)


class SlabHalo:
    """Forward (owner -> ghost) halo of one rank.  `plans[kind][side]` hold, for kind in
    {'bulk', 'mem'} and side in {'lo', 'hi'}: `send` / `recv` index arrays (global device
    numbering of the DeviceProblem: vertex ids across sub-meshes, Q-dof ids)."""

    def __init__(self, layout, sub_keys, sub_offsets, q_keys, q_offsets):
        self.layout = layout
        self.sub_keys, self.sub_offsets = sub_keys, sub_offsets
        self.q_keys, self.q_offsets = q_keys, q_offsets
        self.plans = None
        self.dp = None
        self._dev = None

    # -- set-up handshake: tell each neighbour which of its owned plane entries we hold as ghosts ----
    def _needs(self):
        lay = self.layout
        out = {}
        for kind, keysets in (("bulk", self.sub_keys), ("mem", self.q_keys)):
            for side, plane in (("lo", lay.own_lo - 1), ("hi", lay.own_hi + 1)):
                need = []
                for s, (planes, yz) in keysets.items():
                    sel = np.flatnonzero(planes == plane)
                    need.append((s, yz[sel]))
                out[(kind, side)] = need
        return out

    def build(self, gather_objects):
        """`gather_objects(obj) -> [obj of rank 0, ..., obj of rank world-1]`."""
        lay = self.layout
        all_needs = gather_objects(self._needs())
        plans = {}
        for kind, keysets, offsets in (("bulk", self.sub_keys, self.sub_offsets),
                                       ("mem", self.q_keys, self.q_offsets)):
            plans[kind] = {}
            for side, nb, their_side, my_plane, ghost_plane in (
                    ("lo", lay.rank - 1, "hi", lay.own_lo, lay.own_lo - 1),
                    ("hi", lay.rank + 1, "lo", lay.own_hi, lay.own_hi + 1)):
                if nb < 0 or nb >= lay.world:
                    continue
                send, recv = [], []
                for (s, yz_needed) in all_needs[nb][(kind, their_side)]:
                    planes, yz = keysets[s]
                    mine = np.flatnonzero(planes == my_plane)
                    pos = np.searchsorted(yz[mine], yz_needed)
                    if not np.array_equal(yz[mine][np.clip(pos, 0, len(mine) - 1)], yz_needed):
                        raise RuntimeError("halo: a neighbour's ghost is not owned here")
                    send.append(mine[pos] + offsets[s])
                for s, (planes, yz) in keysets.items():
                    recv.append(np.flatnonzero(planes == ghost_plane) + offsets[s])
                plans[kind][side] = dict(nb=nb, send=np.concatenate(send).astype(np.int32),
                                         recv=np.concatenate(recv).astype(np.int32))
        self.plans = plans
        return plans

    # -- host exchange (CPU tests, Vector.scatter_forward): `fields` = [n_global_ids, width] array --
    def forward_host_array(self, kind, arr, dist):
        import torch
        ops, bufs = [], []
        for side, pl in self.plans[kind].items():
            sb = torch.from_numpy(np.ascontiguousarray(arr[pl["send"]]))
            rb = torch.empty((len(pl["recv"]),) + arr.shape[1:], dtype=torch.float64)
            ops += [dist.P2POp(dist.isend, sb, pl["nb"]), dist.P2POp(dist.irecv, rb, pl["nb"])]
            bufs.append((pl, rb, sb))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for pl, rb, _ in bufs:
            arr[pl["recv"]] = rb.numpy()

    # -- device exchange (bench / production) ------------------------------------------------------------
    def attach(self, dp):
        import torch
        from .. import _lib as L
        self.dp, self.L, self.torch = dp, L, torch
        import torch.distributed as dist
        self.dist = dist
        dev = torch.device("cuda", dp.device)
        n_slots = int(dp.n_models.sum())
        self.width = {"bulk": 4, "mem": 1 + n_slots * L.MAX_IONS}
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
        if dist.get_backend() == "gloo":
            self.mode = "gloo host-staged"
        else:
            flag = torch.tensor([1 if stream_ordered else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            self.mode = "stream-ordered RCCL" if int(flag.item()) else "host-synchronised RCCL"
        self._stream_ordered = self.mode == "stream-ordered RCCL"
        self._dev = {}
        for kind in ("bulk", "mem"):
            # one packed buffer per direction and kind: both neighbours' entries are packed / unpacked by a single
            # kernel launch, the point-to-point operations work on slices of it
            w = self.width[kind]
            sides = list(self.plans[kind].values())
            if not sides:
                continue
            send_idx = np.concatenate([pl["send"] for pl in sides]).astype(np.int32)
            recv_idx = np.concatenate([pl["recv"] for pl in sides]).astype(np.int32)
            d = dict(send_idx=torch.from_numpy(send_idx).to(dev), recv_idx=torch.from_numpy(recv_idx).to(dev),
                     send_buf=torch.empty(len(send_idx) * w, dtype=torch.float64, device=dev),
                     recv_buf=torch.empty(len(recv_idx) * w, dtype=torch.float64, device=dev), parts=[])
            so = ro = 0
            for pl in sides:
                ns, nr = len(pl["send"]) * w, len(pl["recv"]) * w
                d["parts"].append((pl["nb"], slice(so, so + ns), slice(ro, ro + nr)))
                so, ro = so + ns, ro + nr
            self._dev[kind] = d

    def _exchange(self, kind):
        L, dp, dist = self.L, self.dp, self.dist
        d = self._dev.get(kind)
        if d is None:
            return
        k = 0 if kind == "bulk" else 1
        L.check(dp.lib.knpemi_halo_pack(dp.h, k, d["send_idx"].data_ptr(), d["send_idx"].numel(),
                                        d["send_buf"].data_ptr()))
        if dist.get_backend() != "gloo":
            ops = []
            for nb, ss, rs in d["parts"]:
                ops += [dist.P2POp(dist.isend, d["send_buf"][ss], nb), dist.P2POp(dist.irecv, d["recv_buf"][rs], nb)]
            if self._stream_ordered:
                # RCCL: torch's current stream is the library's stream (ExternalStream), so the send/recv kernels
                # are ordered after the pack kernel and `wait()` orders the unpack kernel after them: no host
                # synchronisation anywhere in the exchange (tools/check_async_halo.py rehearses this with real RCCL)
                with self.torch.cuda.stream(self._ext):
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
            else:
                dp.sync()                        # packed data is complete before RCCL reads it
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                self.torch.cuda.current_stream().synchronize()
        else:                           # single-GPU rehearsal: stage through host memory
            dp.sync()
            sb = d["send_buf"].cpu()
            rb = self.torch.empty(d["recv_buf"].shape, dtype=self.torch.float64)
            ops = []
            for nb, ss, rs in d["parts"]:
                ops += [dist.P2POp(dist.isend, sb[ss], nb), dist.P2POp(dist.irecv, rb[rs], nb)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            d["recv_buf"].copy_(rb)
            self.torch.cuda.current_stream().synchronize()
        L.check(dp.lib.knpemi_halo_unpack(dp.h, k, d["recv_idx"].data_ptr(), d["recv_idx"].numel(),
                                          d["recv_buf"].data_ptr()))

    def exchange_bulk(self):
        self._exchange("bulk")

    def exchange_membrane(self):
        self._exchange("mem")


def make_slab_layout_and_mesh(kind, r, rank, world, length=None):
    """length: box length in units of 16 um; default 2 * world (weak scaling: config 2 per rank at r = 1)."""
    cell_type = {"tet": "tetrahedron", "hex": "hexahedron"}[kind]
    l = 2 * world if length is None else int(length)
    lay = SlabLayout(r, rank, world, l)
    mesh, ct, ft = make_mesh_3D_slab(r, cell_type, l, (lay.lo, lay.b))
    return lay, (mesh, ct, ft)


def build_halo(lay, subdomain_list, gather_objects):
    """Keys, offsets and the exchanged plan for the sub-meshes and membrane meshes of one rank."""
    sub_keys, sub_off, q_keys, q_off = {}, {}, {}, {}
    off = qoff = 0
    owned = 0
    for s, (tag, sd) in enumerate(subdomain_list.items()):
        pv = sd["mesh_sub"].parent_vertices
        planes, yz = lay.key_of_local_vertex(pv)
        sub_keys[s], sub_off[s] = (planes, yz), off
        off += len(pv)
        owned += int(((planes >= lay.own_lo) & (planes <= lay.own_hi)).sum())
        if tag > 0:
            qv = sd["mesh_mem"].parent_vertices
            q_keys[s], q_off[s] = lay.key_of_local_vertex(qv), qoff
            qoff += len(qv)
    halo = SlabHalo(lay, sub_keys, sub_off, q_keys, q_off)
    halo.build(gather_objects)
    return halo, owned


def make_slab_problem(kind, r, rank, world, g_syn=10.0, length=None):
    """The rank-local driver set-up (examples/idealized_geometries/setup_problem.Setup): x-slab `rank` of the box of
    length 16 * length um at resolution r (default length 2 * world: the weak-scaling family; length = 2 is the
    reference's own box, cut into `world` slabs: strong scaling)."""
    import torch.distributed as dist
    from setup_problem import Setup
    lay, mesh_data = make_slab_layout_and_mesh(kind, r, rank, world, length)
    s = Setup(kind, r, g_syn=g_syn, mesh_data=mesh_data, build_forms=True)

    def gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    s.halo, owned = build_halo(lay, s.subdomain_list, gather)
    s.owned_dofs = owned
    s.layout = lay
    s.global_length = lay.l * 16e-6
    return s
