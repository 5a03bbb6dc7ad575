"""Host-side mesh / tag / sub-mesh layer ("FEniCSx mesh I/O" for the hot path).

The reference leans on DOLFINx + scifem for everything in this file
(`examples/idealized_geometries/make_mesh_2D.py:43-120`, `make_mesh_3D.py:89-198`,
`tests/make_mesh_mms.py:43-100`, `run_3D.py:114-171`).  Those packages are not
available offline, so this module provides the small subset of their behaviour
that the knpemi hot path consumes: P1/Q1 meshes as flat numpy arrays, entity tags,
facet connectivity, sub-mesh extraction with parent maps, and the oriented
interface data (`scifem.compute_interface_data`, `emiWeakForm.py:40`).

Everything here is plain numpy; it runs once at set-up and its output is what
`knpemi.fem.topology.flatten_problem` uploads to the GPU.
"""
from __future__ import annotations

import numpy as np

# Local facet -> local vertices (Basix/DOLFINx reference-cell convention: simplex
# facet i is opposite vertex i; tensor cells use lexicographic vertex order).
CELL_INFO = {
    "point": dict(tdim=0, nv=1, facet_type=None, facets=np.zeros((0, 0), np.int32)),
    "interval": dict(tdim=1, nv=2, facet_type="point",
                     facets=np.array([[0], [1]], np.int32)),
    "triangle": dict(tdim=2, nv=3, facet_type="interval",
                     facets=np.array([[1, 2], [0, 2], [0, 1]], np.int32)),
    "tetrahedron": dict(tdim=3, nv=4, facet_type="triangle",
                        facets=np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], np.int32)),
    "quadrilateral": dict(tdim=2, nv=4, facet_type="interval",
                          facets=np.array([[0, 1], [0, 2], [1, 3], [2, 3]], np.int32)),
    "hexahedron": dict(tdim=3, nv=8, facet_type="quadrilateral",
                       facets=np.array([[0, 1, 2, 3], [0, 1, 4, 5], [0, 2, 4, 6],
                                        [1, 3, 5, 7], [2, 3, 6, 7], [4, 5, 6, 7]], np.int32)),
}


class CellType:
    """Names mirror `dolfinx.mesh.CellType` members used by the mesh scripts."""
    interval = "interval"
    triangle = "triangle"
    tetrahedron = "tetrahedron"
    quadrilateral = "quadrilateral"
    hexahedron = "hexahedron"


def _row_keys(sorted_rows: np.ndarray, nvert: int) -> np.ndarray:
    """One uint64 key per row of (already sorted) vertex ids."""
    k = sorted_rows.shape[1]
    use = min(k, 3)  # three vertices identify a facet of a valid mesh
    if nvert ** use < 2 ** 63:
        key = np.zeros(sorted_rows.shape[0], np.uint64)
        for c in range(use):
            key = key * np.uint64(nvert) + sorted_rows[:, c].astype(np.uint64)
        return key
    # fall back to a lexicographic rank (slow path, huge meshes only)
    _, inv = np.unique(sorted_rows[:, :use], axis=0, return_inverse=True)
    return inv.astype(np.uint64)


class Mesh:
    """Flat P1/Q1 mesh: `x` (nvert, gdim) f64 and `cells` (ncell, nv) int32."""

    def __init__(self, x, cells, cell_type, comm=None):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.cell_type = cell_type
        info = CELL_INFO[cell_type]
        self.tdim = info["tdim"]
        self.gdim = self.x.shape[1]
        self.comm = comm
        self._facets = None
        # multi-GPU bookkeeping (filled by knpemi.fem.partition): number of
        # owned vertices (ghosts are numbered after them) and the halo plan.
        self.num_owned_vertices = self.x.shape[0]
        self.halo = None

    # -- sizes -------------------------------------------------------------
    @property
    def num_vertices(self):
        return self.x.shape[0]

    @property
    def num_cells(self):
        return self.cells.shape[0]

    @property
    def facet_type(self):
        return CELL_INFO[self.cell_type]["facet_type"]

    # -- facet connectivity --------------------------------------------------
    def _build_facets(self):
        if self._facets is not None:
            return
        info = CELL_INFO[self.cell_type]
        ft = info["facets"]
        nfpc, nvpf = ft.shape
        nc = self.num_cells
        allf = self.cells[:, ft].reshape(nc * nfpc, nvpf)
        keys = _row_keys(np.sort(allf, axis=1), self.num_vertices)
        ukeys, first, inv = np.unique(keys, return_index=True, return_inverse=True)
        nF = ukeys.shape[0]
        facets = allf[first].astype(np.int32)
        cell_facets = inv.reshape(nc, nfpc).astype(np.int32)
        # facet -> cells (CSR), cells in increasing order
        order = np.argsort(inv, kind="stable")
        counts = np.bincount(inv, minlength=nF)
        ptr = np.zeros(nF + 1, np.int64)
        np.cumsum(counts, out=ptr[1:])
        f2c = (order // nfpc).astype(np.int32)
        f2lf = (order % nfpc).astype(np.int32)
        self._facets = dict(facets=facets, cell_facets=cell_facets, ptr=ptr,
                            f2c=f2c, f2lf=f2lf, counts=counts)

    @property
    def facets(self):
        self._build_facets()
        return self._facets["facets"]

    @property
    def num_facets(self):
        return self.facets.shape[0]

    @property
    def cell_facets(self):
        self._build_facets()
        return self._facets["cell_facets"]

    def facet_cells(self):
        """CSR (ptr, cells, local_facet) of the facet->cell connectivity."""
        self._build_facets()
        f = self._facets
        return f["ptr"], f["f2c"], f["f2lf"]

    def num_entities(self, dim):
        if dim == self.tdim:
            return self.num_cells
        if dim == self.tdim - 1:
            return self.num_facets
        if dim == 0:
            return self.num_vertices
        raise ValueError("only cells, facets and vertices are tabulated")

    def entity_vertices(self, dim):
        if dim == self.tdim:
            return self.cells
        if dim == self.tdim - 1:
            return self.facets
        raise ValueError("only cells and facets are tabulated")


class MeshTags:
    """Subset of `dolfinx.mesh.MeshTags`: `.indices`, `.values`, `.dim`, `.find`."""

    def __init__(self, mesh, dim, indices, values, name="tags"):
        indices = np.asarray(indices, np.int32)
        values = np.asarray(values, np.int32)
        order = np.argsort(indices, kind="stable")
        self.mesh = mesh
        self.dim = dim
        self.indices = indices[order]
        self.values = values[order]
        self.name = name

    def find(self, value):
        return self.indices[self.values == value]

    def dense(self, fill=-1):
        out = np.full(self.mesh.num_entities(self.dim), fill, np.int32)
        out[self.indices] = self.values
        return out


def meshtags(mesh, dim, indices, values):
    return MeshTags(mesh, dim, indices, values)


class EntityMap:
    """Sub-entity -> parent-entity index map (`dolfinx.mesh.EntityMap` subset)."""

    def __init__(self, sub_to_parent, num_parent):
        self.sub_to_parent = np.asarray(sub_to_parent, np.int32)
        self.num_parent = int(num_parent)
        self._inv = None

    def sub_topology_to_topology(self, entities, inverse=False):
        entities = np.asarray(entities, np.int32)
        if not inverse:
            return self.sub_to_parent[entities]
        if self._inv is None:
            inv = np.full(self.num_parent, -1, np.int32)
            inv[self.sub_to_parent] = np.arange(self.sub_to_parent.shape[0], dtype=np.int32)
            self._inv = inv
        return self._inv[entities]


# ---------------------------------------------------------------------------
# structured generators (geometry of the reference's mesh scripts)
# ---------------------------------------------------------------------------
def _grid_points(p0, p1, n, axes=None):
    if axes is None:
        axes = [np.linspace(p0[d], p1[d], n[d] + 1) for d in range(len(n))]
    if len(n) == 2:
        Y, X = np.meshgrid(axes[1], axes[0], indexing="ij")
        return np.stack([X.ravel(), Y.ravel()], axis=1)
    Z, Y, X = np.meshgrid(axes[2], axes[1], axes[0], indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)


def create_rectangle(comm, points, n, cell_type=CellType.triangle):
    """`dolfinx.mesh.create_rectangle` geometry (`make_mesh_2D.py:53-55`).

    Vertices are numbered lexicographically (x fastest); each grid square is
    split by its lower-left -> upper-right diagonal ("right" diagonal).
    """
    nx, ny = int(n[0]), int(n[1])
    x = _grid_points(points[0], points[1], (nx, ny))
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    v00 = (i + (nx + 1) * j).ravel()
    v10, v01, v11 = v00 + 1, v00 + nx + 1, v00 + nx + 2
    if cell_type == CellType.triangle:
        cells = np.stack([np.stack([v00, v10, v11], 1), np.stack([v00, v01, v11], 1)], 1).reshape(-1, 3)
    elif cell_type == CellType.quadrilateral:
        cells = np.stack([v00, v10, v01, v11], 1)
    else:
        raise ValueError(cell_type)
    return Mesh(x, cells, cell_type, comm)


def create_unit_square(comm, nx, ny, cell_type=CellType.triangle, ghost_mode=None):
    """`dolfinx.mesh.create_unit_square` (`tests/make_mesh_mms.py:46-48`)."""
    return create_rectangle(comm, [np.zeros(2), np.ones(2)], (nx, ny), cell_type)


# Kuhn split of the unit cube into 6 tetrahedra around the body diagonal 0-7
# (local hex vertices in lexicographic order, bit0 = x, bit1 = y, bit2 = z).
_KUHN = np.array([[0, 1, 3, 7], [0, 1, 5, 7], [0, 2, 3, 7],
                  [0, 2, 6, 7], [0, 4, 5, 7], [0, 4, 6, 7]], np.int32)


def create_box(comm, points, n, cell_type=CellType.hexahedron, axes=None):
    """`dolfinx.mesh.create_box` geometry (`make_mesh_3D.py:100-102`).

    `hexahedron` reproduces the reference's Q1 mesh; `tetrahedron` splits every
    hexahedron into 6 Kuhn tetrahedra (BASELINE configs 2, 3 and 5).  `axes`
    (three coordinate arrays) overrides the uniform grid; the multi-GPU slab
    meshes pass slices of the global axes so that coordinates match bit for bit.
    """
    if axes is not None:
        n = [len(a) - 1 for a in axes]
    nx, ny, nz = (int(v) for v in n)
    x = _grid_points(points[0] if points else None, points[1] if points else None, (nx, ny, nz), axes)
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    sx, sy = 1, nx + 1
    sz = (nx + 1) * (ny + 1)
    v0 = (i * sx + j * sy + k * sz).ravel()
    hexes = np.stack([v0 + (b & 1) * sx + ((b >> 1) & 1) * sy + ((b >> 2) & 1) * sz
                      for b in range(8)], axis=1)
    if cell_type == CellType.hexahedron:
        cells = hexes
    elif cell_type == CellType.tetrahedron:
        cells = hexes[:, _KUHN].reshape(-1, 4)
    else:
        raise ValueError(cell_type)
    mesh = Mesh(x, cells, cell_type, comm)
    if axes is None and points is not None:
        # a uniform grid: every cell has the edge vectors (L_d / n_d) e_d -- known exactly here, handed to the device
        # library so that it need not derive the cell from rounded coordinate differences (knpemi_problem_desc.uniform_cell)
        lo, hi = np.asarray(points[0], float), np.asarray(points[1], float)
        mesh.uniform_cell = np.diag((hi - lo) / np.array([nx, ny, nz], float))
    return mesh


# ---------------------------------------------------------------------------
# entity queries
# ---------------------------------------------------------------------------
def locate_entities(mesh, dim, marker):
    """Entities whose vertices ALL satisfy `marker(x)` with x of shape (3, n)
    (`dolfinx.mesh.locate_entities`, used at `make_mesh_3D.py:121-142`)."""
    xt = np.zeros((3, mesh.num_vertices))
    xt[:mesh.gdim] = mesh.x.T
    ok = np.asarray(marker(xt), bool)
    ev = mesh.entity_vertices(dim)
    return np.flatnonzero(ok[ev].all(axis=1)).astype(np.int32)


def exterior_facet_indices(mesh):
    mesh._build_facets()
    return np.flatnonzero(mesh._facets["counts"] == 1).astype(np.int32)


def find_interface(ct, tag_a, tag_b):
    """Facets shared by a cell tagged `tag_a` and a cell tagged `tag_b`
    (`scifem.find_interface`, `make_mesh_3D.py:165-168`)."""
    mesh = ct.mesh
    ptr, f2c, _ = mesh.facet_cells()
    dense = ct.dense()
    two = np.flatnonzero(np.diff(ptr) == 2)
    c0 = dense[f2c[ptr[two]]]
    c1 = dense[f2c[ptr[two] + 1]]
    hit = ((c0 == tag_a) & (c1 == tag_b)) | ((c0 == tag_b) & (c1 == tag_a))
    return two[hit].astype(np.int32)


def compute_interface_data(ct, facet_indices):
    """Oriented integration data of interior facets: one row
    `(cell+, local_facet+, cell-, local_facet-)` per facet, "+" being the cell
    with the lower cell tag (the ECS side; `emiWeakForm.py:25-26,40`,
    `README.md:69-72`)."""
    mesh = ct.mesh
    facet_indices = np.asarray(facet_indices, np.int64)
    ptr, f2c, f2lf = mesh.facet_cells()
    if np.any(ptr[facet_indices + 1] - ptr[facet_indices] != 2):
        raise RuntimeError("Facet is assumed to be an interior facet")
    dense = ct.dense()
    a = ptr[facet_indices]
    ca, cb = f2c[a], f2c[a + 1]
    la, lb = f2lf[a], f2lf[a + 1]
    swap = dense[ca] > dense[cb]
    plus_c = np.where(swap, cb, ca)
    plus_l = np.where(swap, lb, la)
    minus_c = np.where(swap, ca, cb)
    minus_l = np.where(swap, la, lb)
    return np.stack([plus_c, plus_l, minus_c, minus_l], axis=1).astype(np.int32)


def extract_submesh(mesh, tags, values):
    """`scifem.extract_submesh` (`run_3D.py:156-158`): returns
    `(submesh, sub_to_parent EntityMap, sub_vertex_to_parent EntityMap, None, None)`.

    Sub-mesh vertices are numbered in increasing parent-vertex order, so
    membrane vertices are duplicated between the ECS and ICS sub-meshes.
    """
    if np.isscalar(values):
        values = (values,)
    values = np.asarray(list(values), np.int32)
    ents = tags.indices[np.isin(tags.values, values)]
    ev = mesh.entity_vertices(tags.dim)[ents]
    pverts = np.unique(ev)
    lookup = np.full(mesh.num_vertices, -1, np.int32)
    lookup[pverts] = np.arange(pverts.shape[0], dtype=np.int32)
    ctype = mesh.cell_type if tags.dim == mesh.tdim else mesh.facet_type
    sub = Mesh(mesh.x[pverts], lookup[ev], ctype, mesh.comm)
    if tags.dim == mesh.tdim and getattr(mesh, "uniform_cell", None) is not None:
        sub.uniform_cell = mesh.uniform_cell          # cells of a uniform grid stay cells of that grid
    sub.parent_vertices = pverts      # sub-mesh vertex -> parent vertex
    sub.parent_entities = ents        # sub-mesh cell -> parent entity (cell or facet)
    sub.parent = mesh
    emap = EntityMap(ents, mesh.num_entities(tags.dim))
    vmap = EntityMap(pverts, mesh.num_vertices)
    return sub, emap, vmap, None, None


def match_facets(parent, submesh, sub_vertex_to_parent):
    """Parent facet index of every facet of `submesh` (same vertices), -1 where the parent has no such facet."""
    pv = sub_vertex_to_parent.sub_to_parent
    key_sub = _row_keys(np.sort(pv[submesh.facets], axis=1), parent.num_vertices)
    key_parent = _row_keys(np.sort(parent.facets, axis=1), parent.num_vertices)
    order = np.argsort(key_parent)
    pos = np.clip(np.searchsorted(key_parent[order], key_sub), 0, max(order.shape[0] - 1, 0))
    hit = key_parent[order][pos] == key_sub
    return np.where(hit, order[pos], -1).astype(np.int64)


def transfer_meshtags_to_submesh(ft, submesh, sub_vertex_to_parent, sub_cell_to_parent):
    """`scifem.transfer_meshtags_to_submesh` for facet tags (`emiWeakForm.py:349-351`)."""
    parent = ft.mesh
    key_parent = _row_keys(np.sort(parent.facets[ft.indices], axis=1), parent.num_vertices)
    pv = sub_vertex_to_parent.sub_to_parent
    sub_f_parent_verts = pv[submesh.facets]
    key_sub = _row_keys(np.sort(sub_f_parent_verts, axis=1), parent.num_vertices)
    order = np.argsort(key_parent)
    pos = np.searchsorted(key_parent[order], key_sub)
    pos = np.clip(pos, 0, max(order.shape[0] - 1, 0))
    hit = key_parent[order][pos] == key_sub if order.shape[0] else np.zeros_like(key_sub, bool)
    idx = np.flatnonzero(hit).astype(np.int32)
    vals = ft.values[order][pos[hit]]
    return MeshTags(submesh, submesh.tdim - 1, idx, vals), None
