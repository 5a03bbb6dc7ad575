"""knpemi.rhs_codegen: a membrane module's Python right-hand side (the reference's `rhs_numba` protocol,
/root/reference/src/knpemi/odeSolver.py:96) translated into the device function hipRTC compiles."""
import ctypes as C
import importlib.util
import math
import os
import subprocess
import tempfile

import numpy as np
import pytest

from knpemi.rhs_codegen import hip_source_from_module, hip_source_from_python

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# A right-hand side in the style of the Gotran-generated modules the reference ships (decorated numba cfunc, parameter
# unpacking, math. / np. calls, a comparison used as a factor, `**` and math.pow, a commented-out block as a string
# statement, stores into values[] and -- the side effect -- parameters[]): Hodgkin-Huxley in SI units, SURVEY appendix C.1.
HH_PY = '''
@cfunc(lsoda_sig, nopython=True)
def rhs_numba(t, states, values, parameters):
    """
    Compute the right hand side
    """
    g_Na_bar = parameters[0]; g_K_bar = parameters[1]
    g_leak_Na = parameters[2]
    g_leak_K = parameters[3]
    m_K = parameters[4]
    m_Na = parameters[5]
    I_max = parameters[6]
    Cm = parameters[7]
    stim_amplitude = parameters[8]
    K_e, K_i, Na_e, Na_i = parameters[9], parameters[10], parameters[11], parameters[12]
    z_K = parameters[19]
    psi = parameters[21]
    E_Na = 1/psi * 1/z_K * math.log(Na_e/Na_i)
    E_K = 1/psi * 1/z_K * math.log(K_e/K_i)
    """
    alpha_m = an older variant kept as a comment
    """
    alpha_m = 0.1e3 * (25. - 1.0e3*(states[3] + 65.0e-3))/(math.exp((25. - 1.0e3*(states[3] + 65.0e-3))/10.) - 1)
    beta_m = 4.e3*math.exp(- 1.0e3*(states[3] + 65.0e-3)/18.)
    values[0] = (1 - states[0])*alpha_m - states[0]*beta_m
    alpha_h = 0.07e3*math.exp(- 1.0e3*(states[3] + 65.0e-3)/20.)
    beta_h = 1.e3/(math.exp((30.- 1.0e3*(states[3] + 65.0e-3))/10.) + 1)
    values[1] = (1 - states[1])*alpha_h - states[1]*beta_h
    alpha_n = 0.01e3*(10.- 1.0e3*(states[3] + 65.0e-3))/(math.exp((10.- 1.0e3*(states[3] + 65.0e-3))/10.) - 1.)
    beta_n = 0.125e3*math.exp(- 1.0e3*(states[3] + 65.0e-3) /80.)
    values[2] = (1 - states[2])*alpha_n - states[2]*beta_n
    i_Stim = stim_amplitude * np.exp(-np.mod(t, 0.03)/0.002)*(t < 125e-3)
    i_pump = I_max / ((1 + m_K / K_e) ** 2 * (1 + m_Na / Na_i) ** 3)
    i_Na = (g_leak_Na + g_Na_bar * states[1] * math.pow(states[0], 3) + i_Stim) * \\
           (states[3] - E_Na) + 3 * i_pump
    i_K = (g_leak_K + g_K_bar * math.pow(states[2], 4)) * \\
          (states[3] - E_K) - 2 * i_pump
    parameters[15] = i_Na
    parameters[16] = i_K
    parameters[17] = 0.0
    values[3] = (- i_K - i_Na) / Cm
'''


def _python_function(source, name):
    """The same text as a plain Python function (decorator dropped), for the comparison."""
    ns = {"math": math, "np": np}
    exec(source.replace("@cfunc(lsoda_sig, nopython=True)", ""), ns)
    return ns[name]


def _host_build(hip_source):
    """The generated function compiled as host C++ (the source is plain C++ but for `__device__`)."""
    d = tempfile.mkdtemp()
    src = os.path.join(d, "rhs.cpp")
    with open(src, "w") as f:
        # (kn_exp / kn_log are the sweep's device versions of exp / log, csrc/lsoda_core.h: the C library's on the host)
        f.write("#include <cmath>\nusing namespace std;\n#define __device__\n#define kn_exp exp\n#define kn_log log\n" + hip_source +
                '\nextern "C" void call(double t, const double* s, double* v, double* p) { rhs(t, s, v, p); }\n')
    so = os.path.join(d, "rhs.so")
    subprocess.check_call(["g++", "-O0", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    lib.call.argtypes = [C.c_double] + [C.POINTER(C.c_double)] * 3
    return lib


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gotran_style_rhs_translates_and_computes_the_same_numbers():
    hip = hip_source_from_python(HH_PY, "rhs_numba")
    assert "__device__ inline void rhs(double t, const double* states, double* values, double* parameters)" in hip
    lib = _host_build(hip)
    f = _python_function(HH_PY, "rhs_numba")
    mm = _load(os.path.join(ROOT, "examples", "idealized_geometries", "mm_hh.py"), "mm_hh_tables")
    rng = np.random.default_rng(1)
    for t in (0.0, 0.0123, 0.031, 0.2):
        s = np.asarray(mm.init_state_values(), float) * (1.0 + 0.05 * rng.standard_normal(4))
        p = np.asarray(mm.init_parameter_values(), float)
        p[mm.parameter_indices("Cm")] = 0.02
        p[mm.parameter_indices("stim_amplitude")] = 10.0
        for n_, v in (("K_e", 3.3), ("K_i", 124.0), ("Na_e", 100.7), ("Na_i", 12.8), ("z_K", 1.0), ("psi", 38.7)):
            p[mm.parameter_indices(n_)] = v
        v_py, p_py = np.zeros(4), p.copy()
        f(t, s, v_py, p_py)
        v_c, p_c = np.zeros(4), p.copy()
        lib.call(t, s.ctypes.data_as(C.POINTER(C.c_double)), v_c.ctypes.data_as(C.POINTER(C.c_double)),
                 p_c.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.allclose(v_c, v_py, rtol=1e-14, atol=0) and np.allclose(p_c, p_py, rtol=1e-14, atol=0)
        assert p_c[15] != p[15] and p_c[17] == 0.0              # the side-effect currents were stored


def test_module_rhs_matches_the_hand_written_device_source():
    """examples/benchmark/mm_glial.py carries both a Python `rhs` and a hand-written RHS_HIP: the translation of the
    former computes what the latter computes."""
    mm = _load(os.path.join(ROOT, "examples", "benchmark", "mm_glial.py"), "mm_glial_bench")
    gen = _host_build(hip_source_from_module(mm))
    hand = _host_build(mm.RHS_HIP)
    p = np.asarray(mm.init_parameter_values(), float)
    ix = mm.parameter_indices
    for name, val in (("psi", 0.0378), ("z_Na", 1.0), ("z_K", 1.0), ("z_Cl", -1.0), ("Cm", 1.0), ("K_e", 3.4),
                      ("K_i", 99.3), ("Na_e", 144.6), ("Na_i", 15.8), ("Cl_e", 133.6), ("Cl_i", 5.2)):
        p[ix(name)] = val
    dp = C.POINTER(C.c_double)
    for V in (-85.0, -60.0, -20.0):
        s = np.array([V])
        out = []
        for lib in (gen, hand):
            v, q = np.zeros(1), p.copy()
            lib.call(0.0, s.ctypes.data_as(dp), v.ctypes.data_as(dp), q.ctypes.data_as(dp))
            out.append((v, q))
        assert np.allclose(out[0][0], out[1][0], rtol=1e-13) and np.allclose(out[0][1], out[1][1], rtol=1e-13)


def test_unsupported_constructs_are_named():
    with pytest.raises(NotImplementedError, match="For"):
        hip_source_from_python("def rhs(t, states, values, parameters):\n    for i in range(3):\n        values[i] = 0.0\n")
    with pytest.raises(NotImplementedError, match="read before"):
        hip_source_from_python("def rhs(t, states, values, parameters):\n    values[0] = undefined_name\n")
    with pytest.raises(NotImplementedError, match="signature"):
        hip_source_from_python("def rhs(t, states):\n    return 0\n")


def test_generated_source_cross_compiles_for_gfx950(hip_lib):
    log = C.create_string_buffer(1 << 16)
    hip = hip_source_from_python(HH_PY, "rhs_numba")
    assert hip_lib.knpemi_ode_compile_source(4, 22, hip.encode(), log, len(log)) == 0, log.value.decode()


@pytest.mark.gpu
def test_module_with_only_a_python_rhs_runs_on_the_device(hip_lib):
    """A plug-in with the reference's protocol and nothing else (no MODEL_ID, no RHS_HIP): bound through the generated
    source, integrated on the GPU, and equal to the same model bound through its hand-written device source."""
    from helpers import Setup
    from knpemi.odeSolver import MembraneModel
    mm = _load(os.path.join(ROOT, "examples", "benchmark", "mm_glial.py"), "mm_glial_bench2")

    class PythonOnly:
        __name__ = "mm_glial_python_only"
        rhs = staticmethod(mm.rhs)
        init_state_values = staticmethod(mm.init_state_values)
        init_parameter_values = staticmethod(mm.init_parameter_values)
        state_indices = staticmethod(mm.state_indices)
        parameter_indices = staticmethod(mm.parameter_indices)
    out = []
    for module in (mm, PythonOnly):
        s = Setup("2d", 1, model="glial", build_forms=False)
        user = MembraneModel(module, s.ft, 1, s.phi_M_prev[1].function_space)
        s.subdomain_list[1]['mem_models'] = [{'ode': user, 'I_ch_k': s.mem_models[0]['I_ch_k']}]
        s.mem_models = s.subdomain_list[1]['mem_models']
        s.build_forms()
        ix = mm.parameter_indices
        for name, val in (("psi", 96500e3 / (8.315e3 * 307e3)), ("z_Na", 1.0), ("z_K", 1.0), ("z_Cl", -1.0), ("Cm", 1.0),
                          ("K_i", 99.3100014897692), ("Na_e", 144.60625137617149), ("Na_i", 15.775818906083778),
                          ("Cl_e", 133.62525154406637), ("Cl_i", 5.203660274163705)):
            user.parameters[:, ix(name)] = val
        user.parameters[:, ix("K_e")] = np.linspace(3.0, 12.0, user.nodes)
        for _ in range(3):
            user._pending_flags = 0
            user.step_lsoda(0.1, None)
        out.append((user.states.copy(), user.parameters.copy(), dict(user.last_stats)))
    assert out[1][2]["n_failed"] == 0 and out[1][2]["n_rhs"] > 5 * len(out[1][0])
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-9 * np.abs(out[0][0]).max()
    ich = [mm.parameter_indices(f"I_ch_{n}") for n in ("Na", "K", "Cl")]
    assert np.abs(out[0][1][:, ich] - out[1][1][:, ich]).max() <= 1e-6 * np.abs(out[0][1][:, ich]).max()
