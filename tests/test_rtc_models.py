"""Membrane models that bring their own right-hand side as HIP source (knpemi_ode_bind_source, csrc/kernels_rtc.hip)."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HH_SI_SOURCE = r"""
// examples/idealized_geometries/mm_hh.py:139-227 of the reference, as a user would write it for the device
__device__ inline void rhs(double t, const double* states, double* values, double* parameters) {
  const double m = states[0], h = states[1], n = states[2], V = states[3];
  const double g_Na_bar = parameters[0], g_K_bar = parameters[1], g_leak_Na = parameters[2], g_leak_K = parameters[3];
  const double m_K = parameters[4], m_Na = parameters[5], I_max = parameters[6], Cm = parameters[7];
  const double stim_amplitude = parameters[8], K_e = parameters[9], K_i = parameters[10], Na_e = parameters[11];
  const double Na_i = parameters[12], z_K = parameters[19], psi = parameters[21];
  const double E_Na = 1.0 / psi * 1.0 / z_K * log(Na_e / Na_i);
  const double E_K = 1.0 / psi * 1.0 / z_K * log(K_e / K_i);
  const double u = 1.0e3 * (V + 65.0e-3);
  const double am = 0.1e3 * (25. - u) / (exp((25. - u) / 10.) - 1);
  const double bm = 4.e3 * exp(-u / 18.);
  const double ah = 0.07e3 * exp(-u / 20.);
  const double bh = 1.e3 / (exp((30. - u) / 10.) + 1);
  const double an = 0.01e3 * (10. - u) / (exp((10. - u) / 10.) - 1.);
  const double bn = 0.125e3 * exp(-u / 80.);
  values[0] = (1 - m) * am - m * bm;
  values[1] = (1 - h) * ah - h * bh;
  values[2] = (1 - n) * an - n * bn;
  const double i_stim = stim_amplitude * exp(-fmod(t, 0.03) / 0.002) * (t < 125e-3 ? 1.0 : 0.0);
  const double a1 = 1 + m_K / K_e, a2 = 1 + m_Na / Na_i;
  const double i_pump = I_max / ((a1 * a1) * (a2 * a2 * a2));
  const double i_Na = (g_leak_Na + g_Na_bar * h * (m * m * m) + i_stim) * (V - E_Na) + 3 * i_pump;
  const double n2 = n * n;
  const double i_K = (g_leak_K + g_K_bar * (n2 * n2)) * (V - E_K) - 2 * i_pump;
  parameters[15] = i_Na;
  parameters[16] = i_K;
  parameters[17] = 0.0;
  values[3] = (-i_K - i_Na) / Cm;
}
"""


def load_benchmark_glial():
    spec = importlib.util.spec_from_file_location("mm_glial_benchmark",
                                                  os.path.join(ROOT, "examples", "benchmark", "mm_glial.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plugin_sources_compile_for_gfx950_without_a_device(hip_lib):
    """hipRTC cross-compiles the plug-in under the sweep kernel (no GPU needed); a broken source is refused with the
    compiler's message."""
    log = C.create_string_buffer(1 << 16)
    mm = load_benchmark_glial()
    assert hip_lib.knpemi_ode_compile_source(1, 21, mm.RHS_HIP.encode(), log, len(log)) == 0, log.value.decode()
    assert hip_lib.knpemi_ode_compile_source(4, 22, HH_SI_SOURCE.encode(), log, len(log)) == 0, log.value.decode()
    bad = mm.RHS_HIP.replace("values[0] =", "values[0] = undeclared_symbol +")
    assert hip_lib.knpemi_ode_compile_source(1, 21, bad.encode(), log, len(log)) != 0
    assert b"undeclared_symbol" in log.value and b"undeclared_symbol" in hip_lib.knpemi_last_error()
    assert hip_lib.knpemi_ode_compile_source(0, 21, mm.RHS_HIP.encode(), None, 0) != 0


def test_benchmark_glial_host_rhs_is_near_rest_at_its_initial_state():
    """The host restatement used as the checker: with the calibrated concentrations of the astrocyte examples
    (run_stim_duration.py:205-213) the tabulated initial potential is within 0.03 mV/ms of rest, the three currents
    nearly cancel, and the Kir4.1 / pump current is there."""
    mm = load_benchmark_glial()
    p = mm.init_parameter_values()
    ix = mm.parameter_indices
    for name, val in (("psi", 96500e3 / (8.315e3 * 307e3)), ("z_Na", 1.0), ("z_K", 1.0), ("z_Cl", -1.0), ("Cm", 1.0),
                      ("K_e", 3.092970607490389), ("K_i", 99.3100014897692), ("Na_e", 144.60625137617149),
                      ("Na_i", 15.775818906083778), ("Cl_e", 133.62525154406637), ("Cl_i", 5.203660274163705)):
        p[ix(name)] = val
    y = mm.init_state_values()
    dy = mm.rhs(0.0, y, np.zeros(1), p)
    assert abs(dy[0]) < 0.03 and abs(p[ix("I_ch_Na")] + p[ix("I_ch_K")] + p[ix("I_ch_Cl")]) < 0.03
    # the sodium leak and three pump cycles cancel at the tabulated state (that is how it was calibrated)
    assert abs(p[ix("I_ch_Na")]) < 1e-4 and 0.0 < abs(p[ix("I_ch_K")]) < 0.03


@pytest.mark.gpu
def test_user_source_models_match_odepack_and_the_shipped_kernel(hip_lib):
    """(1) The benchmark glial plug-in (own RHS_HIP, 21 parameters in another order) integrated on the GPU vs scipy's
    ODEPACK LSODA on its host restatement: states 1e-8, side-effect currents 1e-5.  (2) HH written as plug-in source
    (four lanes per dof, every lane evaluating the whole right-hand side) vs the shipped ModelHHSI kernel."""
    from scipy.integrate import odeint
    from helpers import Setup
    from knpemi import _lib as L
    # -- (1) ----------------------------------------------------------------------------------------------------
    mm = load_benchmark_glial()
    assert not hasattr(mm, "MODEL_ID")
    s = Setup("2d", 1, model="glial")
    ode = s.mem_models[0]['ode']
    ix = mm.parameter_indices
    p0 = mm.init_parameter_values()
    for name, val in (("psi", 96500e3 / (8.315e3 * 307e3)), ("z_Na", 1.0), ("z_K", 1.0), ("z_Cl", -1.0), ("Cm", 1.0),
                      ("K_e", 3.4), ("K_i", 99.3100014897692), ("Na_e", 144.60625137617149),
                      ("Na_i", 15.775818906083778), ("Cl_e", 133.62525154406637), ("Cl_i", 5.203660274163705)):
        p0[ix(name)] = val
    n = ode.nodes
    states = np.tile(mm.init_state_values(), (n, 1))
    params = np.tile(p0, (n, 1))
    params[:, ix("K_e")] = np.linspace(3.0, 12.0, n)         # a different ODE on every dof
    dp = ode._dp
    # a second model slot does not exist on this problem: bind the source to a fresh handle of the same mesh instead
    from knpemi.odeSolver import MembraneModel
    s2 = Setup("2d", 1, model="glial", build_forms=False)
    mem = s2.subdomain_list[1]
    user = MembraneModel(mm, s2.ft, 1, s2.phi_M_prev[1].function_space)
    s2.subdomain_list[1]['mem_models'] = [{'ode': user, 'I_ch_k': s2.mem_models[0]['I_ch_k']}]
    s2.mem_models = s2.subdomain_list[1]['mem_models']
    s2.build_forms()
    assert user._dp is not None and user._dp is not dp
    user.states[:], user.parameters[:] = states, params
    dt = 0.1
    ref_y, ref_p = states.copy(), params.copy()

    def f(y, t, p):
        return mm.rhs(t, y, np.zeros(1), p)
    ich = [ix("I_ch_Na"), ix("I_ch_K"), ix("I_ch_Cl")]
    for k in range(5):
        user._pending_flags = 0
        user.step_lsoda(dt, None)
        for i in range(0, n, 7):
            sol = odeint(f, ref_y[i], [k * dt, (k + 1) * dt], args=(ref_p[i],), rtol=1e-8, atol=1e-10)
            ref_y[i] = sol[-1]
        rows = np.arange(0, n, 7)
        assert np.abs(user.states[rows] - ref_y[rows]).max() <= 1e-8 * np.abs(ref_y[rows]).max()
        assert np.abs(user.parameters[rows][:, ich] - ref_p[rows][:, ich]).max() <= 1e-5 * np.abs(ref_p[rows][:, ich]).max()
    assert user.last_stats["n_failed"] == 0 and user.last_stats["n_rhs"] > 5 * n
    assert np.ptp(user.states[:, 0]) > 1.0          # the dofs really differ
    # -- (2) ----------------------------------------------------------------------------------------------------
    out = []
    for variant in ("shipped", "source"):
        s3 = Setup("2d", 1, g_syn=10.0, build_forms=False)
        ode3 = s3.mem_models[0]['ode']
        if variant == "source":
            class _Plug:            # same tables, own right-hand side
                __name__ = "mm_hh_user"
                RHS_HIP = HH_SI_SOURCE
                init_state_values = staticmethod(ode3.ode.init_state_values)
                init_parameter_values = staticmethod(ode3.ode.init_parameter_values)
                state_indices = staticmethod(ode3.ode.state_indices)
                parameter_indices = staticmethod(ode3.ode.parameter_indices)
            ode3.ode = _Plug
        s3.build_forms()
        from knpemi.utils import update_ode_variables
        for k in range(3):
            update_ode_variables(ode3, s3.c_prev, s3.phi_M_prev[1], s3.ion_list, s3.subdomain_list, s3.mesh, s3.ct, 1, k)
            ode3.step_lsoda(s3.dt, s3.stim_params['stimulus'], s3.stim_params['stimulus_locator'])
            ode3.get_membrane_potential(s3.phi_M_prev[1])
        out.append((ode3.states.copy(), ode3.parameters.copy(), dict(ode3.last_stats)))
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-10 * np.abs(out[0][0]).max()
    assert np.abs(out[0][1][:, 15:18] - out[1][1][:, 15:18]).max() <= 1e-5 * np.abs(out[0][1][:, 15:18]).max()
    assert abs(out[0][2]["n_rhs"] - out[1][2]["n_rhs"]) <= 0.01 * out[0][2]["n_rhs"]
