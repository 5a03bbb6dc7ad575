#!/usr/bin/env python3
"""Writes the idealized 2D mesh (geometry and tags of the reference's make_mesh_2D.py) as XDMF + HDF5 with the
grids `mesh`, `cell_marker` and `facet_marker`, as the reference's script does (make_mesh_2D.py:110-120)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "knp-emi-fenics-x_amd"))
from knpemi.fem import XDMFFile, make_mesh_2D  # noqa: E402


def main(output_path, resolution_factor):
    mesh, ct, ft = make_mesh_2D(resolution_factor)
    ct.name, ft.name = "cell_marker", "facet_marker"
    xdmf_filename = os.path.join(output_path, f"mesh_{resolution_factor}.xdmf")
    with XDMFFile(None, xdmf_filename, "w") as xdmf:
        xdmf.write_mesh(mesh)
        xdmf.write_meshtags(ct, None)
        xdmf.write_meshtags(ft, None)
    xdmf.close()
    return xdmf_filename


if __name__ == "__main__":
    for r in (0, 1, 2, 3):
        print(main(os.path.join(HERE, "meshes", "2D"), r))
