"""LSODA work per membrane dof and PDE step at the bench state (diagnostic)."""
import sys, os, contextlib, io, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import conftest  # noqa
from helpers import Setup
from knpemi.stepper import DeviceStepper
with contextlib.redirect_stdout(io.StringIO()):
    s = Setup("tet", 1, g_syn=10.0)
for tag in s.subdomain_list:
    for k in range(2):
        s.c[tag][k].x.array[:] = s.c_prev[tag][k].x._a
    s.phi[tag].x.array[:] = -0.0744 if tag > 0 else 0.0
st = DeviceStepper((s.a_emi, s.p_emi, s.L_emi), (s.a_knp, s.p_knp, s.L_knp), s.c, s.c_prev, s.phi, s.phi_M_prev)
m = s.mem_models[0]['ode']
st.add_membrane_model(m, s.stim_params['stimulus'], s.stim_params['stimulus_locator'])
lib = st.dp.lib
for k in range(12):
    st.step()
    st.dp.sync()
    nr, ns, nf = C.c_int64(), C.c_int64(), C.c_int32()
    lib.knpemi_ode_stats(st.dp.h, m._sub, m._model, C.byref(nr), C.byref(ns), C.byref(nf))
    print(f"step {k}: rhs evals/dof {nr.value / m.nodes:.1f}  lsoda steps/dof {ns.value / m.nodes:.1f}  failures {nf.value}")
