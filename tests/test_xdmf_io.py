"""CPU tests of the XDMF + HDF5 mesh / tag / result files: round trips through the layout
DOLFINx's XDMFFile writes, the reference's `read_mesh` sequence (run_2D.py:114-134), inline XML data items and the
error paths."""
import os

import numpy as np
import pytest

from knpemi.fem import (Function, XDMFFile, extract_submesh, functionspace, make_mesh_2D, make_mesh_3D)
from knpemi.fem import hdf5


def _same_tags(a, b):
    return np.array_equal(a.indices, b.indices) and np.array_equal(a.values, b.values)


@pytest.mark.parametrize("maker,args", [(make_mesh_2D, (1,)), (make_mesh_3D, (0, "tetrahedron")),
                                        (make_mesh_3D, (0, "hexahedron"))])
def test_mesh_and_tags_round_trip(tmp_path, maker, args):
    mesh, ct, ft = maker(*args)
    ct.name, ft.name = "cell_marker", "facet_marker"
    path = tmp_path / "meshes" / "mesh.xdmf"
    with XDMFFile(None, path, "w") as xdmf:            # make_mesh_2D.py:110-120
        xdmf.write_mesh(mesh)
        xdmf.write_meshtags(ct, None)
        xdmf.write_meshtags(ft, None)
    xdmf.close()                                        # the reference closes twice
    assert os.path.isfile(path) and os.path.isfile(tmp_path / "meshes" / "mesh.h5")
    with XDMFFile(None, path, "r") as xdmf:            # run_2D.py:114-134
        m2 = xdmf.read_mesh(ghost_mode=None)
        ct2 = xdmf.read_meshtags(m2, name="cell_marker")
        ft2 = xdmf.read_meshtags(m2, name="facet_marker")
    assert m2.cell_type == mesh.cell_type and np.array_equal(m2.cells, mesh.cells)
    assert np.array_equal(m2.x, mesh.x)               # bit-exact geometry
    assert _same_tags(ct, ct2) and _same_tags(ft, ft2)
    # the sub-mesh builder gives the same sub-domains from the file as from the generator
    for tag in (0, 1):
        a = extract_submesh(mesh, ct, tag)[0]
        b = extract_submesh(m2, ct2, tag)[0]
        assert np.array_equal(a.cells, b.cells) and np.array_equal(a.x, b.x)


def test_tags_in_arbitrary_order_and_vertex_order(tmp_path):
    """Tagged entities are matched by their vertex sets: a shuffled, vertex-permuted facet list gives the same tags."""
    mesh, ct, ft = make_mesh_3D(0, "tetrahedron")
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(ft.indices))
    ent = mesh.facets[ft.indices][perm][:, ::-1]
    with XDMFFile(None, tmp_path / "m.xdmf", "w") as xdmf:
        xdmf.write_mesh(mesh)
    h5 = hdf5.File(str(tmp_path / "t.h5"), "w")
    h5.write("/MeshTags/facet_marker/topology", ent.astype(np.int64))
    h5.write("/MeshTags/facet_marker/Values", ft.values[perm].astype(np.int32).reshape(-1, 1))
    h5.close()
    xml = open(tmp_path / "m.xdmf").read().replace("</Domain>", f"""<Grid Name="facet_marker" GridType="Uniform">
      <Topology TopologyType="Triangle" NumberOfElements="{len(ent)}" NodesPerElement="3">
        <DataItem Dimensions="{len(ent)} 3" NumberType="Int" Format="HDF">t.h5:/MeshTags/facet_marker/topology</DataItem>
      </Topology>
      <Attribute Name="facet_marker" AttributeType="Scalar" Center="Cell">
        <DataItem Dimensions="{len(ent)} 1" Format="HDF">t.h5:/MeshTags/facet_marker/Values</DataItem>
      </Attribute></Grid></Domain>""")
    open(tmp_path / "m.xdmf", "w").write(xml)
    with XDMFFile(None, tmp_path / "m.xdmf", "r") as xdmf:
        m2 = xdmf.read_mesh()
        ft2 = xdmf.read_meshtags(m2, "facet_marker")
    assert _same_tags(ft, ft2)


def test_inline_xml_items_and_errors(tmp_path):
    xml = """<?xml version="1.0"?><Xdmf Version="3.0"><Domain>
      <Grid Name="mesh" GridType="Uniform">
        <Topology TopologyType="Triangle" NumberOfElements="2" NodesPerElement="3">
          <DataItem Dimensions="2 3" NumberType="Int" Format="XML">0 1 2 1 3 2</DataItem></Topology>
        <Geometry GeometryType="XY"><DataItem Dimensions="4 2" Format="XML">0 0 1 0 0 1 1 1</DataItem></Geometry>
      </Grid>
      <Grid Name="cell_marker" GridType="Uniform">
        <Topology TopologyType="Triangle" NumberOfElements="1" NodesPerElement="3">
          <DataItem Dimensions="1 3" NumberType="Int" Format="XML">2 3 1</DataItem></Topology>
        <Attribute Name="cell_marker" AttributeType="Scalar" Center="Cell">
          <DataItem Dimensions="1 1" Format="XML">7</DataItem></Attribute>
      </Grid>
      <Grid Name="bad" GridType="Uniform">
        <Topology TopologyType="Triangle" NumberOfElements="1" NodesPerElement="3">
          <DataItem Dimensions="1 3" NumberType="Int" Format="XML">0 1 3</DataItem></Topology>
        <Attribute Name="bad" AttributeType="Scalar" Center="Cell"><DataItem Dimensions="1 1" Format="XML">1</DataItem></Attribute>
      </Grid>
      <Grid Name="p2" GridType="Uniform"><Topology TopologyType="Triangle_6" NumberOfElements="0" NodesPerElement="6">
          <DataItem Dimensions="0 6" Format="XML"></DataItem></Topology></Grid>
    </Domain></Xdmf>"""
    path = tmp_path / "inline.xdmf"
    path.write_text(xml)
    with XDMFFile(None, path, "r") as xdmf:
        mesh = xdmf.read_mesh()
        assert mesh.cells.tolist() == [[0, 1, 2], [1, 3, 2]] and mesh.gdim == 2
        ct = xdmf.read_meshtags(mesh, "cell_marker")
        assert ct.indices.tolist() == [1] and ct.values.tolist() == [7]
        with pytest.raises(ValueError, match="not entities of the mesh"):
            xdmf.read_meshtags(mesh, "bad")
        with pytest.raises(ValueError, match="unsupported TopologyType"):
            xdmf.read_meshtags(mesh, "p2")
        with pytest.raises(KeyError):
            xdmf.read_meshtags(mesh, "missing")
    with pytest.raises(FileNotFoundError):
        XDMFFile(None, tmp_path / "nope.xdmf", "r")


def test_function_time_series(tmp_path):
    mesh, ct, ft = make_mesh_2D(1)
    sub = extract_submesh(mesh, ct, 1)[0]
    f = Function(functionspace(sub), name="phi_1")
    with XDMFFile(None, tmp_path / "results_sub_1.xdmf", "w") as xdmf:      # run_2D.py:318-330
        xdmf.write_mesh(sub)
        for k in range(3):
            f.x.array[:] = k + sub.x[:, 0]
            xdmf.write_function(f, 0.1 * k)
    h5 = hdf5.File(str(tmp_path / "results_sub_1.h5"), "r")
    assert np.array_equal(h5.read("/Function/phi_1/2").ravel(), 2 + sub.x[:, 0])
    assert h5.exists("/Mesh/mesh/geometry") and not h5.exists("/Function/phi_1/3")
    h5.close()
    text = open(tmp_path / "results_sub_1.xdmf").read()
    assert text.count("<Time ") == 3 and 'CollectionType="Temporal"' in text


def test_hdf5_types(tmp_path):
    with hdf5.File(str(tmp_path / "t.h5"), "w") as h:
        h.write("/a/b/i32", np.arange(6, dtype=np.int32).reshape(2, 3))
        h.write("/a/f32", np.linspace(0, 1, 5, dtype=np.float32))
        h.write("/empty", np.zeros((0, 3)))
    with hdf5.File(str(tmp_path / "t.h5"), "r") as h:
        a = h.read("/a/b/i32")
        assert a.dtype == np.int32 and a.tolist() == [[0, 1, 2], [3, 4, 5]]
        assert h.read("/a/f32").dtype == np.float32 and h.read("/empty").shape == (0, 3)
        with pytest.raises(KeyError):
            h.read("/nothing")
    with pytest.raises(OSError):
        hdf5.File(str(tmp_path / "missing.h5"), "r")


def test_mesh_scripts_and_driver_setup_from_file(tmp_path):
    """make_mesh_2D.py writes what the driver's read_mesh reads; the problem set up from the file has the same
    sub-meshes, membrane space and ODE table as the one set up from the generator (no GPU needed: forms not built)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ex = os.path.join(root, "examples", "idealized_geometries")

    def load(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ex, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    path = load("make_mesh_2D").main(str(tmp_path / "meshes" / "2D"), 1)
    mesh, ct, ft = load("run_2D").read_mesh(path)
    from helpers import Setup
    a = Setup("2d", 1, build_forms=False)
    b = Setup("2d", 1, build_forms=False, mesh_data=(mesh, ct, ft))
    for tag in (0, 1):
        assert np.array_equal(a.subdomain_list[tag]["mesh_sub"].cells, b.subdomain_list[tag]["mesh_sub"].cells)
    assert np.array_equal(a.mem_models[0]["ode"].dof_locations, b.mem_models[0]["ode"].dof_locations)
    assert np.array_equal(a.c_prev[1][0].x._a, b.c_prev[1][0].x._a)
