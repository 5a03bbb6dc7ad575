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

from .distributed import Halo
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


class SlabHalo(Halo):
    """Halo of the structured x-slab partition (the slab meshes are generated directly, without the global mesh):
    `plans[kind][side]`, side in {'lo', 'hi'}, keyed by (vertex plane, y-z index).  Exchange, device plumbing and the
    hooks of the distributed solves are those of `knpemi.fem.distributed.Halo`."""

    def __init__(self, layout, sub_keys, sub_offsets, q_keys, q_offsets):
        super().__init__()
        self.layout = layout
        self.sub_keys, self.sub_offsets = sub_keys, sub_offsets
        self.q_keys, self.q_offsets = q_keys, q_offsets

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
