"""Problem set-up of the idealized drivers, shared by run_2D.py / run_3D.py, bench.py and the tests.

Mirrors everything `solve_system` does before its time loop in the reference's
`examples/idealized_geometries/run_3D.py:137-319` (sub-mesh extraction, parameter and ion
dictionaries, function creation, initial conditions, membrane model, forms), through the
knpemi API, so that a reference user finds the same objects under the same names.
"""
import importlib.util
import os

import numpy as np

from knpemi import (create_functions_emi, create_functions_knp, emi_system, knp_system,
                    set_initial_conditions, setup_membrane_model)
from knpemi.fem import Constant, extract_submesh, make_mesh_2D, make_mesh_3D

HERE = os.path.dirname(os.path.abspath(__file__))
EXAMPLES = os.path.dirname(HERE)


def load_model(name):
    """Membrane plug-in modules of the examples: 'hh_si', 'hh_mv', 'glial'."""
    path = {"hh_si": ("idealized_geometries", "mm_hh"),
            "hh_mv": ("local_astrocyte_depolarization", "mm_hh"),
            "glial": ("local_astrocyte_depolarization", "mm_glial")}[name]
    spec = importlib.util.spec_from_file_location(f"mm_{name}", os.path.join(EXAMPLES, path[0], path[1] + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# constants of run_3D.py:176-197
DT = 1.0e-4
C_M, TEMP, FARADAY, RGAS = 0.02, 300.0, 96485.0, 8.314
PSI = FARADAY / (RGAS * TEMP)
D_NA, D_K, D_CL = 1.33e-9, 1.96e-9, 2.03e-9
NA_I, NA_E = 12.838513108648856, 100.71925900027354
K_I, K_E = 124.15397583491901, 3.3236967382705265
CL_E, CL_I = NA_E + K_E, NA_I + K_I


def make_mesh(kind, r):
    if kind == "2d":
        return make_mesh_2D(r)
    if kind == "hex":
        return make_mesh_3D(r, "hexahedron")
    if kind == "tet":
        return make_mesh_3D(r, "tetrahedron")
    raise ValueError(kind)


class Setup:
    """Everything `solve_system` builds before the time loop (run_3D.py:137-290)."""

    def __init__(self, kind="2d", r=1, g_syn=10.0, model="hh_si", dt=DT, mesh_data=None, build_forms=True):
        self.mesh, self.ct, self.ft = mesh_data if mesh_data is not None else make_mesh(kind, r)
        mesh, ct, ft = self.mesh, self.ct, self.ft
        ECS = {"tag": 0, "name": "ECS"}
        mm = load_model(model)
        neuron = {"tag": 1, "membrane_tags": [1], "name": "neuron", "ode_models": {1: mm}}
        s0, e0, v0, _, _ = extract_submesh(mesh, ct, 0)
        s1, e1, v1, _, _ = extract_submesh(mesh, ct, 1)
        g1, ge1, gv1, _, _ = extract_submesh(mesh, ft, [1])
        ECS.update(mesh_sub=s0, sub_to_parent=e0, sub_vertex_to_parent=v0)
        neuron.update(mesh_sub=s1, sub_to_parent=e1, sub_vertex_to_parent=v1, mesh_mem=g1, mem_to_parent=ge1)
        self.subdomain_list = {0: ECS, 1: neuron}
        self.dt = dt
        rho = {'z': -1, 0: Constant(s0, 0.0), 1: Constant(s1, 0.0)}
        self.physical_parameters = {
            'dt': Constant(mesh, dt), 'n_steps_ODE': Constant(mesh, dt), 'F': Constant(mesh, FARADAY),
            'psi': Constant(mesh, PSI), 'C_phi': Constant(mesh, C_M / dt), 'C_M': Constant(mesh, C_M),
            'R': Constant(mesh, RGAS), 'temperature': Constant(mesh, TEMP), 'rho': rho}

        def consts(a, b):
            return {0: Constant(s0, a), 1: Constant(s1, b)}
        Na = {'c_init': consts(NA_E, NA_I), 'z': 1.0, 'name': 'Na', 'D': consts(D_NA, D_NA)}
        K = {'c_init': consts(K_E, K_I), 'z': 1.0, 'name': 'K', 'D': consts(D_K, D_K)}
        Cl = {'c_init': consts(CL_E, CL_I), 'z': -1.0, 'name': 'Cl', 'D': consts(D_CL, D_CL)}
        self.ion_list = [K, Cl, Na]   # the last ion is eliminated (run_3D.py:256)
        self.phi, self.phi_M_prev = create_functions_emi(self.subdomain_list, degree=1)
        self.c, self.c_prev = create_functions_knp(self.subdomain_list, self.ion_list, degree=1)
        set_initial_conditions(self.ion_list, self.subdomain_list, self.c_prev)
        self.stim_params = {'stimulus': {'stim_amplitude': g_syn},
                            'stimulus_locator': lambda x: (x[0] < 20e-6)}
        self.mem_models = setup_membrane_model(self.stim_params, self.physical_parameters,
                                               neuron['ode_models'], ft, self.phi_M_prev[1].function_space,
                                               self.ion_list)
        self.subdomain_list[1]['mem_models'] = self.mem_models
        self.entity_maps = [ge1, e0, e1]
        if build_forms:
            self.build_forms()

    def build_forms(self):
        self.a_emi, self.p_emi, self.L_emi = emi_system(
            self.mesh, self.ct, self.ft, self.physical_parameters, self.ion_list, self.subdomain_list,
            self.phi, self.phi_M_prev, self.c_prev, self.dt)
        self.a_knp, self.p_knp, self.L_knp = knp_system(
            self.mesh, self.ct, self.ft, self.physical_parameters, self.ion_list, self.subdomain_list,
            self.phi, self.phi_M_prev, self.c, self.c_prev, self.dt)

    # -- seeded perturbation so that parity is tested on non-trivial fields ----------
    def perturb(self, seed=12345, rel=1e-3):
        rng = np.random.default_rng(seed)
        for tag in self.subdomain_list:
            for f in self.c_prev[tag] + [self.ion_list[-1][f'c_{tag}']]:
                f.x.array[:] *= 1.0 + rel * rng.uniform(-1, 1, f.x.array.shape[0])
            self.phi[tag].x.array[:] = 1e-3 * rng.uniform(-1, 1, self.phi[tag].x.array.shape[0])
            for f in self.c[tag]:
                f.x.array[:] = rng.uniform(1.0, 100.0, f.x.array.shape[0])
        self.phi_M_prev[1].x.array[:] = -0.07 + 1e-3 * rng.uniform(-1, 1, self.phi_M_prev[1].x.array.shape[0])
        for mm in self.mem_models:
            for f in mm['I_ch_k'].values():
                f.x.array[:] = 1e-2 * rng.uniform(-1, 1, f.x.array.shape[0])
