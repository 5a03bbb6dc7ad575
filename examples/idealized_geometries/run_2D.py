#!/usr/bin/env python3
"""2D idealized neuron-in-ECS run on the MI355X hot path.

Same structure as the reference's `examples/idealized_geometries/run_2D.py` (`solve_odes` :80-111,
time loop :341-372): per step the membrane ODEs, the EMI solve, the KNP solve and the end-of-step
update, through the knpemi API.  The mesh comes from the in-memory generator or from the XDMF file `make_mesh_2D.py` writes (`--mesh-file`);
ADIOS2 checkpoints are replaced by a compressed `.npz` of the final fields (I/O is outside the hot path).

    python run_2D.py [--res 1] [--steps 10] [--iterative]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "knp-emi-fenics-x_amd"))
sys.path.insert(0, HERE)

from knpemi import (create_solver_emi, create_solver_knp, update_ode_variables,  # noqa: E402
                    update_pde_variables)
from setup_problem import Setup  # noqa: E402


def solve_odes(s, k):
    """ Solve ODEs (membrane models) for each membrane tag in each subdomain """
    for tag, subdomain in s.subdomain_list.items():
        if tag == 0:
            continue
        phi_M_prev_sub = s.phi_M_prev[tag]
        for mem_model in subdomain['mem_models']:
            ode_model = mem_model['ode']
            update_ode_variables(ode_model, s.c_prev, phi_M_prev_sub, s.ion_list, s.subdomain_list,
                                 s.mesh, s.ct, tag, k)
            ode_model.step_lsoda(dt=s.dt, stimulus=s.stim_params['stimulus'],
                                 stimulus_locator=s.stim_params['stimulus_locator'])
            ode_model.get_membrane_potential(phi_M_prev_sub)
            for ion, I_ch_k in mem_model['I_ch_k'].items():
                ode_model.get_parameter("I_ch_" + ion, I_ch_k)


def read_mesh(mesh_file):
    """run_2D.py:114-134 of the reference, through knpemi.fem.XDMFFile."""
    from knpemi.fem import XDMFFile
    with XDMFFile(None, mesh_file, 'r') as xdmf:
        mesh = xdmf.read_mesh(ghost_mode=None)
        ct = xdmf.read_meshtags(mesh, name='cell_marker')
        ft = xdmf.read_meshtags(mesh, name='facet_marker')
    xdmf.close()
    return mesh, ct, ft


def solve_system(kind, res, n_steps, direct=True, g_syn=10.0, out=None, mesh_file=None):
    s = Setup(kind, res, g_syn=g_syn, mesh_data=read_mesh(mesh_file) if mesh_file else None)
    problem_emi = create_solver_emi(s.a_emi, s.L_emi, s.phi, s.entity_maps, s.subdomain_list, None,
                                    direct=direct, p=s.p_emi, atol=1e-40, rtol=1e-5)
    problem_knp = create_solver_knp(s.a_knp, s.L_knp, s.c, s.entity_maps, s.subdomain_list, None,
                                    direct=direct, p=s.p_knp, atol=2e-40, rtol=1e-7)
    num_it_emi, num_it_knp = [], []
    t = 0.0
    for k in range(n_steps):
        print(f'Solving for t = {t:.4f} s')
        solve_odes(s, k)
        problem_emi.solve()
        problem_knp.solve()
        num_it_emi.append(problem_emi.solver.getIterationNumber())
        num_it_knp.append(problem_knp.solver.getIterationNumber())
        update_pde_variables(s.c, s.c_prev, s.phi, s.phi_M_prev, s.physical_parameters, s.ion_list,
                             s.subdomain_list, s.mesh, s.ct)
        t += s.dt
    if out:
        os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
        fields = {f.name: f.x._a for tag in s.subdomain_list for f in [s.phi[tag]] + s.c[tag]}
        fields.update({f.name: f.x._a for f in s.phi_M_prev.values()})
        np.savez_compressed(out, t=t, **fields)
    return s, num_it_emi, num_it_knp


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--iterative", action="store_true")
    ap.add_argument("--mesh-file", default=None, help="XDMF mesh written by make_mesh_2D.py (default: generate)")
    a = ap.parse_args()
    s, it_emi, it_knp = solve_system("2d", a.res, a.steps, direct=not a.iterative, mesh_file=a.mesh_file,
                                     out=os.path.join(HERE, "results", f"2D_{a.res}.npz"))
    v = s.phi_M_prev[1].x._a
    print(f"phi_M after {a.steps} steps: min {v.min():.6f} V, max {v.max():.6f} V")
    print(f"average number of iterations emi solver: {sum(it_emi) / len(it_emi)}")
    print(f"average number of iterations knp solver: {sum(it_knp) / len(it_knp)}")
