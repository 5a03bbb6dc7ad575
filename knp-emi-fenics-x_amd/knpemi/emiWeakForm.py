"""EMI (potential) sub-problem: function creation and form descriptors.

Drop-in for `src/knpemi/emiWeakForm.py`.  The reference writes the weak form in
UFL and lets FFCx/DOLFINx generate and run the element kernels; here
`emi_system` returns *descriptors* of the same integrals

    a = sum_r int kappa_r grad(u_r).grad(v_r) dx
        + sum_{r>0,m} int C_phi (u_r(-) - u_0(+)) (v_r(-) - v_0(+)) dS(m)     (:138-167)
    p = a + sum_{r>0} int u_r v_r dx                                          (:169-198)
    L = sum_{r,k} -F z_k int D_k grad(c_k).grad(v_r) dx
        + sum_{r>0,m} int C_phi g (v_r(-) - v_0(+)) dS(m)                      (:201-241)

which `create_solver_emi` hands to the HIP kernel `emi_rows_kernel`
(csrc/kernels_assemble.hip).  The tuples are opaque to the drivers, exactly as
the UFL forms are (`run_3D.py:281-312`).
"""
from __future__ import annotations

import numpy as np

from .device import DeviceProblem
from .fem import Function, functionspace
from .forms import FormDescriptor, Measures, bind_membrane_models

i_res = "-"
e_res = "+"


def create_measures(mesh, ct, ft):
    """ dx, dS, ds descriptors on the parent mesh; dS carries the oriented
    interface data of every facet tag (emiWeakForm.py:28-51) """
    return Measures(mesh, ct, ft).as_tuple()


def create_functions_emi(subdomain_list, degree=1):
    """ Potentials phi = {tag: Function(V_tag)} and previous membrane
    potentials phi_M_prev = {tag > 0: Function(Q_tag)} (emiWeakForm.py:54-81) """
    phi, phi_M_prev = {}, {}
    for tag, subdomain in subdomain_list.items():
        V = functionspace(subdomain["mesh_sub"], ("CG", degree))
        phi[tag] = Function(V, name=f"phi_{tag}")
        if tag > 0:
            Q = functionspace(subdomain["mesh_mem"], ("CG", degree))
            phi_M_prev[tag] = Function(Q, name=f"phi_M_{tag}")
    return phi, phi_M_prev


def emi_system(mesh, ct, ft, physical_params, ion_list, subdomain_list,
               phi, phi_M_prev, c_prev, dt, degree=1, splitting_scheme=True, mms=None):
    """ Create and return the EMI forms (a, p, L); with `mms` also (dx, bc) """
    if degree != 1:
        raise NotImplementedError("the MI355X hot path implements CG-1 (degree=1) only")
    if mms is not None:
        splitting_scheme = False   # no ODEs in the MMS runs (emiWeakForm.py:294)
    dp = DeviceProblem.get(mesh, ct, ft, subdomain_list, ion_list)
    bind_membrane_models(dp, subdomain_list, ion_list)
    shared = dict(dp=dp, physical_params=physical_params, ion_list=ion_list,
                  subdomain_list=subdomain_list, phi=phi, phi_M_prev=phi_M_prev, c_prev=c_prev,
                  dt=dt, splitting_scheme=splitting_scheme, mms=mms, mesh=mesh, ct=ct, ft=ft)
    a = FormDescriptor("emi", "a", shared)
    p = FormDescriptor("emi", "p", shared)
    Lf = FormDescriptor("emi", "L", shared)
    if mms is None:
        return a, p, Lf
    from .mms import emi_dirichlet_bc
    dx, _, _ = create_measures(mesh, ct, ft)
    return a, p, Lf, dx, emi_dirichlet_bc(mesh, ft, subdomain_list, phi, mms)
