"""KNP (concentration) sub-problem: function creation and form descriptors.

Drop-in for `src/knpemi/knpWeakForm.py`.  `knp_system` returns descriptors of

    a = sum_{r,k<K} int (1/dt) u v + D grad(u).grad(v) + z psi D u grad(phi_r).grad(v) dx   (:123-143)
    p = a                                                                                   (:319)
    L = sum_{r,k<K} int (1/dt) c_prev v dx [+ int f v dx(0)]
        + membrane Robin/coupling terms with alpha = D z^2 c / sum_j D_j z_j^2 c_j           (:146-216)

assembled by `knp_rows_kernel` and `knp_membrane_kernel` (csrc/kernels_assemble.hip).
"""
from __future__ import annotations

from .device import DeviceProblem
from .emiWeakForm import create_measures  # same measures (knpWeakForm.py:21-44)
from .fem import Function, functionspace
from .forms import FormDescriptor, bind_membrane_models

i_res = "-"
e_res = "+"


def create_functions_knp(subdomain_list, ion_list, degree=1):
    """ c, c_prev = {tag: [Function] * (K-1)}; side effect: the eliminated ion's
    concentration is stored as ion_list[-1]['c_tag'] (knpWeakForm.py:47-80) """
    n_solved = len(ion_list) - 1
    c, c_prev = {}, {}
    for tag, subdomain in subdomain_list.items():
        V = functionspace(subdomain['mesh_sub'], ("CG", degree))
        spaces = [V.clone() for _ in range(n_solved)]
        c[tag] = [Function(W, name=f"c_{ion['name']}_{tag}") for W, ion in zip(spaces, ion_list)]
        c_prev[tag] = [Function(W) for W in spaces]
        ion_list[-1][f'c_{tag}'] = Function(V, name=f"c_{ion_list[-1]['name']}_{tag}")
    return c, c_prev


def knp_system(mesh, ct, ft, physical_params, ion_list, subdomain_list,
               phi, phi_M_prev, c, c_prev, dt, degree=1, splitting_scheme=True, mms=None):
    """ Create and return the KNP forms (a, p, L); with `mms` also dx """
    if degree != 1:
        raise NotImplementedError("the MI355X hot path implements CG-1 (degree=1) only")
    if mms is not None:
        raise NotImplementedError("the KNP MMS right-hand side (tests/run_mms.py) is unfinished in "
                                  "the reference (SURVEY.md M5) and is not reproduced")
    dp = DeviceProblem.get(mesh, ct, ft, subdomain_list, ion_list)
    bind_membrane_models(dp, subdomain_list, ion_list)
    shared = dict(dp=dp, physical_params=physical_params, ion_list=ion_list,
                  subdomain_list=subdomain_list, phi=phi, phi_M_prev=phi_M_prev, c=c, c_prev=c_prev,
                  dt=dt, splitting_scheme=splitting_scheme, mms=None, mesh=mesh, ct=ct, ft=ft)
    a = FormDescriptor("knp", "a", shared)
    Lf = FormDescriptor("knp", "L", shared)
    return a, a, Lf
