"""Where a trip of the flattened LSODA spends its cycles (diagnostic build, KNPEMI_ODE_STAMPS=1): s_memtime sums per
phase and workgroup of the ODE sweep at the bench state.  Shares, not durations (the stamps serialise the phases)."""
import os, sys, contextlib, io, ctypes as C
os.environ["KNPEMI_ODE_STAMPS"] = "1"
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
names = ["loop head", "TOP", "PRED", "RHS", "CORR", "ERR: test+update", "prologue", "-", "ERR: orderswitch", "ERR: coef+scaleh", "ERR: rest", "-"]
for k in range(2):
    st.step()
    st.dp.sync()
    buf = np.zeros(24 * 4096, np.uint64)
    nb = lib.knpemi_debug_ode_stamps(st.dp.h, m._sub, m._model, buf.ctypes.data_as(C.POINTER(C.c_uint64)), 4096)
    a = buf[:24 * nb].reshape(nb, 24).astype(np.float64)
    tot = a[:, :12].sum(1)
    slow = int(np.argmax(tot))
    print(f"step {k}: {nb} workgroups; cycles per workgroup mean {tot.mean():.0f} max {tot.max():.0f} (x 10 ns at 100 MHz)")
    for i, n in enumerate(names):
        print(f"   {n:20s} mean cycles {a[:, i].mean():9.0f}  share {a[:, i].sum() / tot.sum():6.3f}  trips that ran it {a[:, 12 + i].mean():6.1f}"
              f"   | slowest wg: {a[slow, i]:9.0f} / {a[slow, 12 + i]:4.0f}")
