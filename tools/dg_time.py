"""Times the two DG assembly kernels on an idealized 3D mesh (HIP events, knpemi_dg_time_kernel) and prints the achieved
fraction of the HBM roofline from the algorithmic bytes (DG section)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("knp-emi-fenics-x_amd", "examples/idealized_geometries"):
    sys.path.insert(0, os.path.join(ROOT, p))


def init_fields(dp):
    """Physical parameters and a smooth initial state (functions of the coordinates: partition-independent)."""
    ions = [dict(name="Na", z=1.0, D=[1.33e-9] * 2), dict(name="K", z=1.0, D=[1.96e-9] * 2), dict(name="Cl", z=-1.0, D=[2.03e-9] * 2)]
    dp.set_params(dict(dt=1e-4, F=96485.0, psi=96485.0 / (8.314 * 300.0), C_M=0.02), ions)
    ins = (dp.cell_sub > 0)[:, None] * np.ones((1, dp.nv), bool)
    for k, (e, i) in enumerate(((100.0, 12.0), (4.0, 125.0), (104.0, 137.0))):
        dp.set_concentration(k, np.where(ins, i, e) * (1.0 + 0.01 * np.cos(1e6 * dp.X[:, :, 0])))
    dp.set_potential(np.where(ins, -0.07, 0.0) + 1e-3 * np.sin(1e6 * dp.X[:, :, 0]))
    dp.set_membrane_potential(np.full((dp.nmf, dp.nf), -0.07))
    return dp


def build(r, l=2, cell="tetrahedron"):
    from knpemi.dg import DGProblem
    from knpemi.fem.idealized import make_mesh_3D
    mesh, ct, ft = make_mesh_3D(r, cell, l=l)
    return init_fields(DGProblem(mesh, ct, ft, [0, 1], [1]))


def algorithmic_bytes(dp, which):
    """HBM bytes one launch must move: the CSR values and right-hand sides it writes, the dof records, connectivity and
    row pointers it reads once (neighbour records are re-reads of the same array: cache traffic, not counted)."""
    n, nnz, KS = dp.n, dp.nnz, dp.K - 1
    nfc = 6 if dp.nv == 8 else dp.nv                    # facets per cell: nbr, finfo, mfid entries
    reads = n * 64 + dp.n_cells * nfc * 12 + n * 4 + dp.n_cells
    if which == 0:
        return reads + nnz * 8 + n * 8
    return reads + KS * (nnz * 8 + n * 8)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-r", type=int, default=1)
    ap.add_argument("-l", type=int, default=2)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cell", default="tetrahedron", choices=["tetrahedron", "hexahedron"])
    a = ap.parse_args()
    dp = build(a.r, a.l, a.cell)
    print(f"cells {dp.n_cells} dofs {dp.n} nnz {dp.nnz} membrane nodes {dp.nmf * dp.nf}")
    for which, name in ((0, "dg_emi_kernel"), (1, "dg_knp_kernel")):
        ms = dp.time_kernel(which, a.reps)
        by = algorithmic_bytes(dp, which)
        print(f"{name}: {ms * 1e3:.1f} us  {by / 1e6:.1f} MB  {by / ms / 1e6:.0f} GB/s  frac {by / ms / 1e6 / 8000:.3f}")
