#!/usr/bin/env python3
"""3D idealized run (four axons in a box) on the MI355X hot path: the reference's
`examples/idealized_geometries/run_3D.py` (hexahedral mesh, `make_mesh_3D.py:100-102`; g_syn = 0,
Tstop = 2e-3, `run_3D.py:176-177,265`), or the 6-tet split of BASELINE configs 2/3 with `--tets`.

    python run_3D.py [--res 0] [--steps 20] [--tets] [--iterative]
"""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from run_2D import solve_system  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=0)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--tets", action="store_true")
    ap.add_argument("--iterative", action="store_true")
    ap.add_argument("--mesh-file", default=None, help="XDMF mesh written by make_mesh_3D.py (default: generate)")
    a = ap.parse_args()
    s, it_emi, it_knp = solve_system("tet" if a.tets else "hex", a.res, a.steps, direct=not a.iterative,
                                     g_syn=0.0, mesh_file=a.mesh_file, out=os.path.join(HERE, "results", f"3D_{a.res}.npz"))
    v = s.phi_M_prev[1].x._a
    print(f"phi_M after {a.steps} steps: min {v.min():.6f} V, max {v.max():.6f} V")
    print(it_emi)
    print(it_knp)
