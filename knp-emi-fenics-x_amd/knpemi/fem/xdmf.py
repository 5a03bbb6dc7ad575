"""XDMF + HDF5 mesh / tag / result files in the layout DOLFINx's `dolfinx.io.XDMFFile` uses.

The reference's drivers read their meshes with

    with dolfinx.io.XDMFFile(comm, mesh_file, 'r') as xdmf:          # run_2D.py:114-134
        mesh = xdmf.read_mesh(ghost_mode=...)
        ct = xdmf.read_meshtags(mesh, name='cell_marker')
        ft = xdmf.read_meshtags(mesh, name='facet_marker')

and the mesh scripts write them with `write_mesh` / `write_meshtags` (make_mesh_2D.py:110-120, remark_mesh.py).
`XDMFFile` here keeps those method names and the on-disk conventions of that writer **[3P-knowledge of the DOLFINx
XDMF layout]**: one `Grid` "mesh" with `/Mesh/mesh/topology` + `/Mesh/mesh/geometry`, one `Grid` per tag set with
`/MeshTags/<name>/topology` (entity -> vertices) + `/MeshTags/<name>/Values`, heavy data in `<file>.h5` (inline
`Format="XML"` items are read too), VTK vertex order for quadrilaterals / hexahedra, `write_function` as a temporal
collection of node-centred attributes.  Heavy data goes through `hdf5.py` (ctypes on libhdf5).
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

import numpy as np

from . import hdf5
from .mesh import CELL_INFO, Mesh, MeshTags

_XDMF_NAME = {"interval": "PolyLine", "triangle": "Triangle", "quadrilateral": "Quadrilateral",
              "tetrahedron": "Tetrahedron", "hexahedron": "Hexahedron"}
_CELL_OF = {v.lower(): k for k, v in _XDMF_NAME.items()}
_CELL_OF["polyline"] = "interval"
# XDMF (VTK) vertex order <-> tensor-product order (both maps are involutions)
_PERM = {"quadrilateral": [0, 1, 3, 2], "hexahedron": [0, 1, 3, 2, 4, 5, 7, 6]}
_XI_NS = "https://www.w3.org/2001/XInclude"
_XI = "{" + _XI_NS + "}"
ET.register_namespace("xi", _XI_NS)


def _void_rows(a):
    a = np.ascontiguousarray(a)
    return a.view(np.dtype((np.void, a.dtype.itemsize * a.shape[1]))).ravel()


def _match_rows(table, query, what):
    """Index in `table` of every row of `query` (rows are vertex tuples, compared as sets)."""
    t = _void_rows(np.sort(table.astype(np.int64), axis=1))
    q = _void_rows(np.sort(query.astype(np.int64), axis=1))
    order = np.argsort(t, kind="stable")
    pos = np.searchsorted(t[order], q)
    pos = np.clip(pos, 0, len(order) - 1)
    idx = order[pos]
    if not np.array_equal(t[idx], q):
        raise ValueError(f"XDMF meshtags: {int((t[idx] != q).sum())} tagged {what} are not entities of the mesh")
    return idx


class XDMFFile:
    def __init__(self, comm, filename, file_mode="r"):
        self.comm = comm
        self.filename = str(filename)
        self.mode = file_mode
        self.h5name = os.path.splitext(self.filename)[0] + ".h5"
        self._h5 = None
        self._tree = None
        self._nsteps = {}
        if file_mode == "r":
            if not os.path.isfile(self.filename):
                raise FileNotFoundError(self.filename)
            self._tree = ET.parse(self.filename)
        elif file_mode == "w":
            os.makedirs(os.path.dirname(os.path.abspath(self.filename)), exist_ok=True)
            self._root = ET.Element("Xdmf", {"Version": "3.0"})
            self._domain = ET.SubElement(self._root, "Domain")
            self._h5 = hdf5.File(self.h5name, "w")
        else:
            raise ValueError("file_mode must be 'r' or 'w'")

    # -- context manager / close (the reference calls close() again after the with block) -----------------
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self.mode == "w" and self._h5 is not None:
            self._flush()
        if self._h5 is not None:
            self._h5.close()
            self._h5 = None

    def _flush(self):
        ET.indent(self._root)
        with open(self.filename, "w") as fh:
            fh.write('<?xml version="1.0"?>\n<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>\n')
            fh.write(ET.tostring(self._root, encoding="unicode"))
            fh.write("\n")

    # -- reading -------------------------------------------------------------------------------------------------
    def _grid(self, name):
        for g in self._tree.getroot().iter("Grid"):
            if g.get("Name") == name and g.get("GridType", "Uniform") == "Uniform":
                return g
        raise KeyError(f"{self.filename}: no Grid named {name!r}")

    def _item(self, node):
        item = node.find("DataItem")
        if item is None:
            raise ValueError(f"{self.filename}: <{node.tag}> without DataItem")
        dims = [int(d) for d in item.get("Dimensions", "").split()]
        text = (item.text or "").strip()
        if item.get("Format", "XML").upper() == "HDF":
            fname, path = text.split(":", 1)
            fname = os.path.join(os.path.dirname(os.path.abspath(self.filename)), fname)
            if self._h5 is None or self._h5.path != fname:
                if self._h5 is not None:
                    self._h5.close()
                self._h5 = hdf5.File(fname, "r")
            a = self._h5.read(path)
        else:
            a = np.array(text.split(), dtype=np.float64)
            if item.get("NumberType", "Float") in ("Int", "UInt"):
                a = a.astype(np.int64)
        return a.reshape(dims) if dims and int(np.prod(dims)) == a.size else a

    def _topology(self, grid):
        topo = grid.find("Topology")
        cell = _CELL_OF.get(topo.get("TopologyType", "").lower())
        if cell is None:
            raise ValueError(f"{self.filename}: unsupported TopologyType {topo.get('TopologyType')!r} (P1/Q1 cells only)")
        conn = np.asarray(self._item(topo), np.int64).reshape(-1, CELL_INFO[cell]["nv"])
        if cell in _PERM:
            conn = conn[:, _PERM[cell]]
        return cell, conn

    def read_mesh(self, ghost_mode=None, name="mesh"):
        grid = self._grid(name)
        cell, conn = self._topology(grid)
        geo = grid.find("Geometry")
        x = np.asarray(self._item(geo), np.float64)
        gdim = 2 if geo.get("GeometryType", "XYZ").upper() == "XY" else 3
        x = x.reshape(-1, x.shape[-1] if x.ndim == 2 else gdim)
        tdim = CELL_INFO[cell]["tdim"]
        if tdim == 2 and x.shape[1] == 3 and not np.any(x[:, 2]):
            x = x[:, :2]            # planar meshes are often stored with a zero z column
        return Mesh(x, conn.astype(np.int32), cell, comm=self.comm)

    def read_meshtags(self, mesh, name):
        grid = self._grid(name)
        cell, conn = self._topology(grid)
        att = grid.find("Attribute")
        values = np.asarray(self._item(att)).reshape(-1)
        dim = CELL_INFO[cell]["tdim"]
        if dim == mesh.tdim:
            idx = _match_rows(mesh.cells, conn, "cells")
        elif dim == mesh.tdim - 1:
            idx = _match_rows(mesh.facets, conn, "facets")
        else:
            raise ValueError(f"meshtags {name!r}: entities of dimension {dim} are not tabulated")
        tags = MeshTags(mesh, dim, idx, values.astype(np.int32), name=name)
        return tags

    # -- writing -------------------------------------------------------------------------------------------------
    def _data_item(self, parent, path, array, number_type=None):
        self._h5.write(path, array)
        attrs = {"Dimensions": " ".join(str(d) for d in array.shape), "Format": "HDF"}
        if number_type:
            attrs["NumberType"] = number_type
        item = ET.SubElement(parent, "DataItem", attrs)
        item.text = f"{os.path.basename(self.h5name)}:{path}"

    def _write_topology(self, grid, cell, conn, path):
        out = conn[:, _PERM[cell]] if cell in _PERM else conn
        topo = ET.SubElement(grid, "Topology", {"TopologyType": _XDMF_NAME[cell], "NumberOfElements": str(len(conn)),
                                               "NodesPerElement": str(conn.shape[1])})
        self._data_item(topo, path, np.ascontiguousarray(out, np.int64), "Int")

    def write_mesh(self, mesh, name="mesh"):
        grid = ET.SubElement(self._domain, "Grid", {"Name": name, "GridType": "Uniform"})
        self._write_topology(grid, mesh.cell_type, mesh.cells, f"/Mesh/{name}/topology")
        geo = ET.SubElement(grid, "Geometry", {"GeometryType": "XY" if mesh.gdim == 2 else "XYZ"})
        self._data_item(geo, f"/Mesh/{name}/geometry", mesh.x)
        self._mesh_name = name
        self._flush()

    def write_meshtags(self, tags, geometry=None, name=None):
        mesh = tags.mesh
        name = name or tags.name
        cell = mesh.cell_type if tags.dim == mesh.tdim else mesh.facet_type
        ent = mesh.entity_vertices(tags.dim)[tags.indices]
        grid = ET.SubElement(self._domain, "Grid", {"Name": name, "GridType": "Uniform"})
        ET.SubElement(grid, _XI + "include", {"xpointer": "xpointer(/Xdmf/Domain/Grid/Geometry)"})
        self._write_topology(grid, cell, ent, f"/MeshTags/{name}/topology")
        att = ET.SubElement(grid, "Attribute", {"Name": name, "AttributeType": "Scalar", "Center": "Cell"})
        self._data_item(att, f"/MeshTags/{name}/Values", np.asarray(tags.values, np.int32).reshape(-1, 1))
        self._flush()

    def write_function(self, f, t=0.0, mesh_name=None):
        """Append the nodal values of a P1/Q1 Function at time t (temporal collection named after the Function)."""
        name = f.name
        coll = None
        for g in self._domain.findall("Grid"):
            if g.get("Name") == name and g.get("GridType") == "Collection":
                coll = g
        if coll is None:
            coll = ET.SubElement(self._domain, "Grid", {"Name": name, "GridType": "Collection",
                                                       "CollectionType": "Temporal"})
        step = self._nsteps.get(name, 0)
        self._nsteps[name] = step + 1
        grid = ET.SubElement(coll, "Grid", {"Name": name, "GridType": "Uniform"})
        ET.SubElement(grid, _XI + "include", {"xpointer": "xpointer(/Xdmf/Domain/Grid[@GridType='Uniform'][1]/"
                                                         "*[self::Topology or self::Geometry])"})
        ET.SubElement(grid, "Time", {"Value": repr(float(t))})
        att = ET.SubElement(grid, "Attribute", {"Name": name, "AttributeType": "Scalar", "Center": "Node"})
        self._data_item(att, f"/Function/{name}/{step}", np.asarray(f.x._a, np.float64).reshape(-1, 1))
        self._flush()
