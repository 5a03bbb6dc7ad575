"""Passive glial membrane model of the reference's `benchmark` example (mV / ms) -- a plug-in that brings its OWN
right-hand side to the GPU.

Protocol, parameter order (psi first, 21 parameters), constants (18.4 / 42.5 in the Kir4.1 factor, E_K_init without
z_K) and initial values follow the reference's `examples/benchmark/mm_glial.py:6-125,127-215`; they differ from
the glial model of the astrocyte example, so none of the right-hand sides shipped inside libknpemi_hip.so fits.  Where
the reference's module defines `rhs_numba`, a numba cfunc whose address numbalsoda calls (`odeSolver.py:96`), this one
defines `RHS_HIP`: the same function as HIP source, compiled for gfx950 with hipRTC when `MembraneModel` binds the
model (knpemi_ode_bind_source).  `rhs` is the same function in Python, for host-side checks (the tests integrate it
with scipy's ODEPACK LSODA and compare with the GPU).
"""
import math

import numpy as np

_STATES = ("V",)
_STATE_INIT = dict(V=-85.85765274084892)
_PARAMS = ("psi", "g_leak_Cl", "g_leak_Na", "g_leak_K", "z_Na", "z_K", "z_Cl", "Cm", "stim_amplitude",
           "I_ch_Na", "I_ch_K", "I_ch_Cl", "K_e", "K_i", "Na_e", "Na_i", "Cl_e", "Cl_i", "m_K", "m_Na", "I_max")
_PARAM_INIT = dict(g_leak_Cl=0.05, g_leak_Na=0.1, g_leak_K=1.696, m_K=1.5, m_Na=10.0, I_max=10.75975)

RHS_HIP = r"""
// rhs(t, states, values, parameters): numbalsoda's signature.  parameters is the dof's in/out row.
__device__ inline void rhs(double t, const double* states, double* values, double* parameters) {
  (void)t;
  const double psi = parameters[0], g_leak_Cl = parameters[1], g_leak_Na = parameters[2], g_leak_K = parameters[3];
  const double z_K = parameters[5], z_Cl = parameters[6], Cm = parameters[7];
  const double K_e = parameters[12], K_i = parameters[13], Na_e = parameters[14], Na_i = parameters[15];
  const double Cl_e = parameters[16], Cl_i = parameters[17];
  const double m_K = parameters[18], m_Na = parameters[19], I_max = parameters[20];
  const double V = states[0];
  const double E_Na = 1 / psi * 1 / z_K * log(Na_e / Na_i);
  const double E_K = 1 / psi * 1 / z_K * log(K_e / K_i);
  const double E_Cl = 1 / psi * 1 / z_Cl * log(Cl_e / Cl_i);
  const double K_e_init = 3.092970607490389, K_i_init = 99.3100014897692;
  const double i_pump = I_max * (K_e / (K_e + m_K)) * (pow(Na_i, 1.5) / (pow(Na_i, 1.5) + pow(m_Na, 1.5)));
  const double E_K_init = 1 / psi * log(K_e_init / K_i_init);
  const double dphi = V - E_K;
  const double A = 1 + exp(18.4 / 42.4);
  const double B = 1 + exp(-(0.1186e3 + E_K_init) / 0.0441e3);
  const double C = 1 + exp((dphi + 0.0185e3) / 0.0425e3);
  const double D = 1 + exp(-(0.1186e3 + V) / 0.0441e3);
  const double g_Kir = sqrt(K_e / K_e_init) * (A * B) / (C * D);
  const double i_Kir = g_leak_K * g_Kir * (V - E_K);
  const double i_Na = g_leak_Na * (V - E_Na) + 3 * i_pump;
  const double i_K = i_Kir - 2 * i_pump;
  const double i_Cl = g_leak_Cl * (V - E_Cl);
  parameters[9] = i_Na;
  parameters[10] = i_K;
  parameters[11] = i_Cl;
  values[0] = (-i_K - i_Na - i_Cl) / Cm;
}
"""


def rhs(t, states, values, parameters):
    """The same right-hand side on the host (in/out `parameters`, result in `values`)."""
    (psi, g_leak_Cl, g_leak_Na, g_leak_K, _z_Na, z_K, z_Cl, Cm, _stim, _iNa, _iK, _iCl, K_e, K_i, Na_e, Na_i, Cl_e,
     Cl_i, m_K, m_Na, I_max) = parameters
    V = states[0]
    E_Na = 1 / psi * 1 / z_K * math.log(Na_e / Na_i)
    E_K = 1 / psi * 1 / z_K * math.log(K_e / K_i)
    E_Cl = 1 / psi * 1 / z_Cl * math.log(Cl_e / Cl_i)
    K_e_init, K_i_init = 3.092970607490389, 99.3100014897692
    i_pump = I_max * (K_e / (K_e + m_K)) * (Na_i ** 1.5 / (Na_i ** 1.5 + m_Na ** 1.5))
    E_K_init = 1 / psi * math.log(K_e_init / K_i_init)
    dphi = V - E_K
    A = 1 + math.exp(18.4 / 42.4)
    B = 1 + math.exp(-(0.1186e3 + E_K_init) / 0.0441e3)
    Cc = 1 + math.exp((dphi + 0.0185e3) / 0.0425e3)
    D = 1 + math.exp(-(0.1186e3 + V) / 0.0441e3)
    g_Kir = math.sqrt(K_e / K_e_init) * (A * B) / (Cc * D)
    i_Kir = g_leak_K * g_Kir * (V - E_K)
    i_Na = g_leak_Na * (V - E_Na) + 3 * i_pump
    i_K = i_Kir - 2 * i_pump
    i_Cl = g_leak_Cl * (V - E_Cl)
    parameters[9], parameters[10], parameters[11] = i_Na, i_K, i_Cl
    values[0] = (-i_K - i_Na - i_Cl) / Cm
    return values


def _init(names, defaults, overrides, what):
    out = np.array([defaults.get(n, 0.0) for n in names], dtype=np.float64)
    for name, value in overrides.items():
        if name not in names:
            raise ValueError(f"{name} is not a {what}.")
        out[names.index(name)] = value
    return out


def _indices(names, wanted, what):
    for w in wanted:
        if w not in names:
            raise ValueError(f"Unknown {what}: '{w}'")
    idx = [names.index(w) for w in wanted]
    return idx if len(idx) > 1 else idx[0]


def init_state_values(**values):
    return _init(_STATES, _STATE_INIT, values, "state")


def init_parameter_values(**values):
    return _init(_PARAMS, _PARAM_INIT, values, "parameter")


def state_indices(*states):
    return _indices(_STATES, states, "state")


def parameter_indices(*params):
    return _indices(_PARAMS, params, "param")
