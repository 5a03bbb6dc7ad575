"""Hodgkin-Huxley membrane model in SI units (V, s, S/m^2) -- plug-in module.

Same module protocol, state/parameter order, initial values and parameter names as
the reference's Gotran module `examples/idealized_geometries/mm_hh.py:7-131`.  The
right-hand side itself lives in the HIP library (csrc/membrane_models.h,
`ModelHHSI`, restating `mm_hh.py:139-227`); `MODEL_ID` selects it.
"""
import numpy as np

MODEL_ID = "hh_si"

_STATES = ("m", "h", "n", "V")
_STATE_INIT = dict(m=0.016648440745822956, h=0.8542015627820805, n=0.1882020248041632,
                   V=-0.07438609374462003)

_PARAMS = ("g_Na_bar", "g_K_bar", "g_leak_Na", "g_leak_K", "m_K", "m_Na", "I_max", "Cm",
           "stim_amplitude", "K_e", "K_i", "Na_e", "Na_i", "Cl_e", "Cl_i",
           "I_ch_Na", "I_ch_K", "I_ch_Cl", "z_Na", "z_K", "z_Cl", "psi")
_PARAM_INIT = dict(g_Na_bar=1200.0, g_K_bar=360.0, g_leak_Na=1.0, g_leak_K=4.0,
                   m_K=2.0, m_Na=7.7, I_max=0.449)


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
