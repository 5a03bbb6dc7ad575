import sys, io, contextlib
sys.path[:0]=['knp-emi-fenics-x_amd','oracle','examples/idealized_geometries','tests']
import numpy as np
from helpers import Setup
from knpemi import update_ode_variables, update_pde_variables
from knpemi.pdeSolver import create_solver_emi, create_solver_knp
from knpemi.stepper import DeviceStepper
def mk():
    with contextlib.redirect_stdout(io.StringIO()):
        s = Setup("tet", 0, g_syn=10.0)
    s.perturb()
    for t in s.subdomain_list:
        for k in range(2):
            s.c[t][k].x.array[:] = s.c_prev[t][k].x._a * 1.001
    s.phi[1].x.array[:] += -0.0744
    return s
s1=mk(); s2=mk()
ode1=s1.mem_models[0]['ode']; ode2=s2.mem_models[0]['ode']
st = DeviceStepper((s2.a_emi, s2.p_emi, s2.L_emi), (s2.a_knp, s2.p_knp, s2.L_knp), s2.c, s2.c_prev, s2.phi, s2.phi_M_prev)
st.add_membrane_model(ode2, s2.stim_params['stimulus'], s2.stim_params['stimulus_locator'])
emi = create_solver_emi(s1.a_emi, s1.L_emi, s1.phi, s1.entity_maps, s1.subdomain_list, None, p=s1.p_emi, direct=False)
knp = create_solver_knp(s1.a_knp, s1.L_knp, s1.c, s1.entity_maps, s1.subdomain_list, None, p=s1.p_knp)
for k in range(3):
    with contextlib.redirect_stdout(io.StringIO()):
        update_ode_variables(ode1, s1.c_prev, s1.phi_M_prev[1], s1.ion_list, s1.subdomain_list, s1.mesh, s1.ct, 1, k)
        ode1.step_lsoda(s1.dt, s1.stim_params['stimulus'], s1.stim_params['stimulus_locator'])
        ode1.get_membrane_potential(s1.phi_M_prev[1])
        for ion, f in s1.mem_models[0]['I_ch_k'].items():
            ode1.get_parameter("I_ch_" + ion, f)
        emi.assemble(); knp.assemble()
        update_pde_variables(s1.c, s1.c_prev, s1.phi, s1.phi_M_prev, s1.physical_parameters, s1.ion_list, s1.subdomain_list, s1.mesh, s1.ct)
    st.step(); st.download()
    d = np.abs(ode1.states-ode2.states).max(axis=0); dp_ = np.abs(ode1.parameters-ode2.parameters).max(axis=0)
    print(k, "state diff", d, "param diff cols", np.flatnonzero(dp_>0), dp_[dp_>0], "time", ode1.time, ode2.time)
