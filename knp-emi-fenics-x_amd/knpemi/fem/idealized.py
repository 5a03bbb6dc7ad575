"""In-memory versions of the reference's idealized mesh scripts.

Geometry, tags and marker conventions follow
`examples/idealized_geometries/make_mesh_2D.py:21-120`,
`examples/idealized_geometries/make_mesh_3D.py:12-198` and
`tests/make_mesh_mms.py:21-100`: cell tag 0 = ECS, 1 = ICS; facet tag
0 = interior, 1 = membrane, 5 = exterior boundary (the MMS mesh leaves all
other facets untagged).
"""
from __future__ import annotations

import numpy as np

from .mesh import (CellType, create_box, create_rectangle, create_unit_square,
                   exterior_facet_indices, find_interface, locate_entities, meshtags)


def _box_marker(lo, hi, tol=1e-12):
    lo = np.asarray(lo, float)
    hi = np.asarray(hi, float)

    def marker(x):
        ok = np.ones(x.shape[1], bool)
        for d in range(lo.shape[0]):
            ok &= (x[d] >= lo[d] - tol) & (x[d] <= hi[d] + tol)
        return ok
    return marker


def _tag(mesh, interior_boxes, interior_tags, full_facet_tags=True):
    tdim = mesh.tdim
    cell_marker = np.zeros(mesh.num_cells, np.int32)
    for (lo, hi), tag in zip(interior_boxes, interior_tags):
        cell_marker[locate_entities(mesh, tdim, _box_marker(lo, hi))] = tag
    ct = meshtags(mesh, tdim, np.arange(mesh.num_cells, dtype=np.int32), cell_marker)
    ct.name = "cell_marker"
    marker = np.full(mesh.num_facets, 0 if full_facet_tags else -1, np.int32)
    for tag in sorted(set(interior_tags)):
        marker[find_interface(ct, tag, 0)] = tag
    marker[exterior_facet_indices(mesh)] = 5
    keep = np.flatnonzero(marker != -1).astype(np.int32)
    ft = meshtags(mesh, tdim - 1, keep, marker[keep])
    ft.name = "facet_marker"
    return ct, ft


def make_mesh_2D(resolution_factor, comm=None):
    """62 x 4 um rectangle, ICS = [1,61] x [1,3] um, triangles
    (`make_mesh_2D.py:21-22,45-55`).  r = 0 has an empty ICS."""
    n = (31 * 2 ** resolution_factor, 2 * 2 ** resolution_factor)
    mesh = create_rectangle(comm, [np.array([0.0, 0.0]), np.array([62.0e-6, 4.0e-6])], n,
                            CellType.triangle)
    ct, ft = _tag(mesh, [([1e-6, 1e-6], [61e-6, 3e-6])], [1])
    return mesh, ct, ft


def axon_boxes(l=2):
    """The four axon boxes of `make_mesh_3D.py:12-24`."""
    xs = (5e-6, l * 16e-6 - 5e-6)
    yz = [(0.2e-6, 0.4e-6), (0.5e-6, 0.7e-6)]
    order = [(0, 0), (1, 1), (1, 0), (0, 1)]  # (y range, z range) of axon 1..4
    return [([xs[0], yz[a][0], yz[b][0]], [xs[1], yz[a][1], yz[b][1]]) for a, b in order]


def make_mesh_3D(resolution_factor, cell_type=CellType.hexahedron, l=2, axon_tags=(1, 1, 1, 1),
                 comm=None):
    """32 x 0.9 x 0.9 um box with four axons (`make_mesh_3D.py:89-198`).

    `cell_type=tetrahedron` gives the 6-tet split used by BASELINE configs 2/3;
    `l` stretches the box (and the axons) along x in units of 16 um, which the
    multi-GPU weak-scaling runs use (`l = 2 * n_gpus`).  `axon_tags` lets the
    config-5 style runs tag axons 3-4 as a second cell type.
    """
    n = (l * 16 * 2 ** resolution_factor, 9 * 2 ** resolution_factor, 9 * 2 ** resolution_factor)
    mesh = create_box(comm, [np.zeros(3), np.array([l * 16e-6, 0.9e-6, 0.9e-6])], n, cell_type)
    ct, ft = _tag(mesh, axon_boxes(l), list(axon_tags))
    return mesh, ct, ft


def make_mesh_3D_slab(resolution_factor, cell_type, l, x_cells, comm=None):
    """Cells with x-index in [x_cells[0], x_cells[1]) of `make_mesh_3D(r, cell_type, l)`: the local
    mesh of one rank of the x-slab partition (coordinates identical to the global mesh)."""
    n = (l * 16 * 2 ** resolution_factor, 9 * 2 ** resolution_factor, 9 * 2 ** resolution_factor)
    hi = (l * 16e-6, 0.9e-6, 0.9e-6)
    axes = [np.linspace(0.0, hi[d], n[d] + 1) for d in range(3)]
    axes[0] = axes[0][x_cells[0]:x_cells[1] + 1]
    mesh = create_box(comm, None, None, cell_type, axes=axes)
    mesh.uniform_cell = np.diag(np.array(hi) / np.array(n, float))      # the global grid's cell: the same bits on every rank
    ct, ft = _tag(mesh, axon_boxes(l), [1, 1, 1, 1])
    return mesh, ct, ft


def make_mesh_mms(M, comm=None):
    """Unit square, ICS = [0.25, 0.75]^2 (`tests/make_mesh_mms.py:21-24,43-83`)."""
    mesh = create_unit_square(comm, M, M)
    ct, ft = _tag(mesh, [([0.25, 0.25], [0.75, 0.75])], [1], full_facet_tags=False)
    return mesh, ct, ft
