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

# A SYNTHETIC right-hand side written for this test (not any model of the reference): three states, nine parameters, and
# every construct the Gotran-generated modules of the reference use -- a decorated numba cfunc, parameter unpacking one by
# one, several on a line and as a tuple, math. / np. calls, a comparison used as a factor, `**` and math.pow, `%` and np.mod
# on a possibly NEGATIVE operand, a two-argument math.log, a conditional expression, an augmented assignment, a line
# continuation, a commented-out block as a string statement, stores into values[] and -- the side effect -- parameters[].
SYNTH_PY = '''
@cfunc(lsoda_sig, nopython=True)
def rhs_numba(t, states, values, parameters):
    """
    Compute the right hand side
    """
    gain = parameters[0]; leak = parameters[1]
    tau = parameters[2]
    a_in, a_out, shift = parameters[3], parameters[4], parameters[5]
    period = parameters[6]
    (u, w, q) = states
    rev = 1/gain * math.log(a_out/a_in)
    """
    rate = an older variant kept as a comment
    """
    rate = 0.5e2 * (3. - 2.0e1*(u + shift))/(math.exp((3. - 2.0e1*(u + shift))/4.) - 1)
    decay = 7.e1*math.exp(- 2.0e1*(u + shift)/9.)
    values[1] = (1 - w)*rate - w*decay
    phase = np.mod(t - 0.4*period, period)
    wrapped = (t - 0.4*period) % period
    trunc = math.fmod(t - 0.4*period, period)
    drive = parameters[7] * np.exp(-phase/0.3)*(t < 2.5) + 1e-3*(wrapped - phase) + 1e-2*trunc
    octave = math.log(1.0 + q*q, 2.0)
    pump = leak / ((1 + tau / a_out) ** 2 * (1 + tau / a_in) ** 3)
    i_fast = (leak + gain * w * math.pow(u, 2) + drive) * \\
             (u - rev) + 3 * pump
    i_slow = (0.5*leak if u > rev else 2.0*leak) * (u - rev) - 2 * pump
    i_slow += octave
    parameters[7 + 1] = i_fast - i_slow
    values[0] = (- i_slow - i_fast) / tau
    values[2] = -q/tau + np.sqrt(np.absolute(u))
'''


def _synth_tables(rng):
    s = np.array([-0.3, 0.4, 0.7]) * (1.0 + 0.05 * rng.standard_normal(3))
    p = np.array([1.7, 0.3, 2.5, 11.0, 140.0, 0.12, 0.9, 4.0, 0.0])
    return s, p


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
    hip = hip_source_from_python(SYNTH_PY, "rhs_numba", n_states=3, n_params=9)
    assert "__device__ inline void rhs(double t, const double* states, double* values, double* parameters)" in hip
    lib = _host_build(hip)
    f = _python_function(SYNTH_PY, "rhs_numba")
    rng = np.random.default_rng(1)
    dp = C.POINTER(C.c_double)
    for t in (0.0, 0.123, 0.31, 0.37, 2.0, 3.1):          # t - 0.4 period < 0 for the first ones: the sign of `%` matters
        s, p = _synth_tables(rng)
        v_py, p_py = np.zeros(3), p.copy()
        f(t, s, v_py, p_py)
        v_c, p_c = np.zeros(3), p.copy()
        lib.call(t, s.ctypes.data_as(dp), v_c.ctypes.data_as(dp), p_c.ctypes.data_as(dp))
        assert np.allclose(v_c, v_py, rtol=1e-13, atol=0) and np.allclose(p_c, p_py, rtol=1e-13, atol=0), t
        assert p_c[8] != p[8]                                    # the side-effect store happened


def test_modulo_takes_the_sign_of_the_divisor_as_in_python():
    """`%` and np.mod follow the divisor's sign, math.fmod / np.fmod the dividend's (C's fmod): a time-modulo stimulus with
    a negative operand must not change value in translation (round-3 advisor finding: all three were emitted as fmod)."""
    src = ("def rhs(t, states, values, parameters):\n"
           "    values[0] = (t - 5.0) % 3.0\n    values[1] = np.mod(t - 5.0, 3.0)\n    values[2] = math.fmod(t - 5.0, 3.0)\n"
           "    values[3] = np.fmod(t - 5.0, -3.0)\n    values[4] = (t - 5.0) % -3.0\n")
    lib = _host_build(hip_source_from_python(src))
    dp = C.POINTER(C.c_double)
    for t in (0.5, 4.0, 7.25, -2.0):
        v = np.zeros(5)
        lib.call(t, np.zeros(1).ctypes.data_as(dp), v.ctypes.data_as(dp), np.zeros(1).ctypes.data_as(dp))
        x = t - 5.0
        assert np.allclose(v, [x % 3.0, np.mod(x, 3.0), math.fmod(x, 3.0), np.fmod(x, -3.0), x % -3.0], rtol=1e-15, atol=1e-15), (t, v)


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
    head = "def rhs(t, states, values, parameters):\n"
    with pytest.raises(NotImplementedError, match=r"math.exp.*2 argument"):        # wrong arity: named, not a hipRTC error
        hip_source_from_python(head + "    values[0] = math.exp(t, 2.0)\n")
    with pytest.raises(NotImplementedError, match="3 names unpacked from `parameters`, which has 4"):
        hip_source_from_python(head + "    (a, b, c) = parameters\n    values[0] = a\n", n_params=4)
    with pytest.raises(NotImplementedError, match="out of range"):
        hip_source_from_python(head + "    values[0] = parameters[4]\n", n_params=4)
    with pytest.raises(NotImplementedError, match="keyword"):
        hip_source_from_python(head + "    values[0] = np.power(t, x2=2.0)\n")
    assert "(kn_log(t) / kn_log(10.0))" in hip_source_from_python(head + "    values[0] = math.log(t, 10)\n")


def test_generated_source_cross_compiles_for_gfx950(hip_lib):
    log = C.create_string_buffer(1 << 16)
    hip = hip_source_from_python(SYNTH_PY, "rhs_numba", n_states=3, n_params=9)
    assert hip_lib.knpemi_ode_compile_source(3, 9, hip.encode(), log, len(log)) == 0, log.value.decode()


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
