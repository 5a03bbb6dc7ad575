"""Hodgkin-Huxley membrane model in mV / ms units -- plug-in module.

Protocol, ordering and initial values of the reference's
`examples/local_astrocyte_depolarization/mm_hh.py:7-128`; device RHS `ModelHHMV`
(csrc/membrane_models.h, restating `mm_hh.py:130-201`).
"""
import numpy as np

MODEL_ID = "hh_mv"

_STATES = ("m", "h", "n", "V")
_STATE_INIT = dict(m=0.015211986965658385, h=0.8667432624969533, n=0.17994146133363148,
                   V=-75.09159534786934)
_PARAMS = ("g_Na_bar", "g_K_bar", "g_leak_Na", "g_leak_K", "m_K", "m_Na", "I_max", "Cm",
           "stim_amplitude", "K_e", "K_i", "Na_e", "Na_i", "Cl_e", "Cl_i",
           "I_ch_Na", "I_ch_K", "I_ch_Cl", "z_Na", "z_K", "z_Cl", "psi")
_PARAM_INIT = dict(g_Na_bar=120.0, g_K_bar=36.0, g_leak_Na=0.1, g_leak_K=0.4,
                   m_K=1.5, m_Na=10.0, I_max=58.0)


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
