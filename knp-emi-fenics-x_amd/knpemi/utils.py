"""Glue between the PDE fields and the membrane ODEs (drop-in for `src/knpemi/utils.py`).

Function names, argument order and in-place mutation semantics follow the
reference; the arithmetic runs on the GPU:

* `update_ode_variables` (utils.py:210-235) uploads the concentration fields and
  arms the fused ODE launch, which gathers the membrane traces itself;
* `update_pde_variables` (utils.py:238-295) is one launch of `update_pde_kernel`;
* `interpolate_to_membrane` (utils.py:150-207) is the gather kernel `trace_kernel`.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L
from .device import DeviceProblem
from .fem.function import Function, as_float
from .odeSolver import MembraneModel

i_res = "-"
e_res = "+"


def _is_last(idx, ion_list):
    return idx == len(ion_list) - 1


def _conc(ion_list, c, tag, idx):
    """c_prev[tag][idx] for solved ions, the eliminated-ion Function for the last one."""
    return ion_list[-1][f'c_{tag}'] if _is_last(idx, ion_list) else c[tag][idx]


def set_initial_conditions(ion_list, subdomain_list, c_prev):
    """ Set initial conditions given by constants or nodal arrays (utils.py:90-102) """
    for tag in subdomain_list:
        for idx, ion in enumerate(ion_list):
            target = _conc(ion_list, c_prev, tag, idx)
            init = ion['c_init'][tag]
            target.x.array[:] = init if isinstance(init, np.ndarray) else as_float(init)
            target.x.scatter_forward()


def setup_membrane_model(stim_params, physical_params, ode_models, ct, Q, ion_list):
    """ Initiate membrane model(s): ODE tables plus the source-term Functions
        I_ch_k on Q handed to the PDE forms (utils.py:105-148) """
    mem_models = []
    for tag_sub, ode in ode_models.items():
        ode_model = MembraneModel(ode, ct, tag_sub, Q)
        # constants shared with the PDE side
        C_M, psi = as_float(physical_params["C_M"]), as_float(physical_params["psi"])
        ode_model.set_parameter_values({'Cm': lambda x: C_M})
        ode_model.set_parameter_values({'psi': lambda x: psi})
        for ion in ion_list:
            z = ion['z']
            ode_model.set_parameter_values({f"z_{ion['name']}": lambda x, z=z: z})
        I_ch_k = {}
        for ion in ion_list:
            f = Function(Q, name=f"I_ch_{ion['name']}")
            ode_model.get_parameter("I_ch_" + ion['name'], f)
            I_ch_k[ion['name']] = f
        mem_models.append({'ode': ode_model, 'I_ch_k': I_ch_k})
    return mem_models


def _device_problem(mesh):
    dp = getattr(mesh, "_knpemi_device_problem", None)
    if dp is None:
        raise RuntimeError("no device problem for this mesh: call emi_system()/knp_system() first")
    return dp


def interpolate_to_membrane(ue, ui, Q, mesh, ct, subdomain_list, tag):
    """Nodal traces of an (ECS, cell) function pair on the membrane space Q
    (utils.py:150-207); returns new Functions named after the inputs."""
    dp = _device_problem(mesh)
    qe_a, qi_a = dp.trace(dp.sub_index[tag], ue.x._a, ui.x._a)
    qe = Function(Q, name=ue.name)
    qi = Function(Q, name=ui.name)
    qe.x.array[:] = qe_a
    qi.x.array[:] = qi_a
    qe.x.scatter_forward()
    qi.x.scatter_forward()
    return qe, qi


def update_ode_variables(ode_model, c_prev, phi_M_prev, ion_list, subdomain_list, mesh, ct, tag, k):
    """ Update parameters in ODE solver (based on previous PDEs step).

    The reference writes the six concentration traces into the parameter table
    and, for k > 0, phi_M_prev into the V column on the host (utils.py:217-233).
    Here the fields go to the device and the writes happen at the start of the
    fused ODE launch (`MembraneModel.step_lsoda`); the host tables show them
    after that launch. """
    dp = _device_problem(mesh)
    s = dp.sub_index[tag]
    K = len(ion_list)
    for sub, t in ((0, 0), (s, tag)):
        for idx in range(K - 1):
            dp.push(L.F_C_PREV, sub, idx, c_prev[t][idx])
        dp.push(L.F_C_ELIM, sub, 0, ion_list[-1][f'c_{t}'])
    flags = L.ODE_SET_TRACES
    if k > 0:
        dp.push(L.F_PHI_M, s, 0, phi_M_prev)
        flags |= L.ODE_SET_V
    ode_model._pending_flags = flags


def update_pde_variables(c, c_prev, phi, phi_M_prev, physical_parameters, ion_list, subdomain_list,
                         mesh, ct):
    """ End-of-step update (utils.py:238-295): c_prev <- c, eliminated ion from
    electroneutrality, phi_M_prev <- tr(phi_i) - tr(phi_e); all Functions are
    updated in place. (The UFL Nernst potentials the reference rebuilds at
    utils.py:270-282 are never used by any form and are not reproduced.) """
    dp = _device_problem(mesh)
    dp.set_params(physical_parameters, ion_list, physical_parameters['dt'])
    n_solved = len(ion_list) - 1
    for tag in subdomain_list:
        s = dp.sub_index[tag]
        dp.push(L.F_PHI, s, 0, phi[tag])
        for idx in range(n_solved):
            dp.push(L.F_C, s, idx, c[tag][idx])
    dp.update_pde()
    for tag in subdomain_list:
        s = dp.sub_index[tag]
        for idx in range(n_solved):
            dp.pull(L.F_C_PREV, s, idx, c_prev[tag][idx])
            c_prev[tag][idx].x.scatter_forward()
        dp.pull(L.F_C_ELIM, s, 0, ion_list[-1][f'c_{tag}'])
        if tag != 0:
            dp.pull(L.F_PHI_M, s, 0, phi_M_prev[tag])
            phi_M_prev[tag].x.scatter_forward()
