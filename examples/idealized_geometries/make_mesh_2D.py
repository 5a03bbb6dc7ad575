#!/usr/bin/env python3
"""Writes the idealized 2D meshes (geometry and tags of the reference's make_mesh_2D.py) as `.npz`."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "knp-emi-fenics-x_amd"))
from knpemi.fem import make_mesh_2D  # noqa: E402


def main(output_path, resolution_factor):
    mesh, ct, ft = make_mesh_2D(resolution_factor)
    os.makedirs(output_path, exist_ok=True)
    np.savez_compressed(os.path.join(output_path, f"mesh_{resolution_factor}.npz"), x=mesh.x, cells=mesh.cells,
                        cell_marker=ct.dense(), facets=mesh.facets[ft.indices], facet_marker=ft.values)


if __name__ == "__main__":
    for r in (0, 1, 2, 3):
        main(os.path.join(HERE, "meshes", "2D"), r)
