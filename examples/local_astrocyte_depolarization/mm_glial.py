"""Passive glial membrane model with Kir4.1 and Na/K pump (mV / ms) -- plug-in module.

Protocol, ordering and initial values of the reference's
`examples/local_astrocyte_depolarization/mm_glial.py:7-131`; device RHS `ModelGlial`
(csrc/membrane_models.h, restating `mm_glial.py:133-205`).
"""
import numpy as np

MODEL_ID = "glial"

_STATES = ("V",)
_STATE_INIT = dict(V=-85.84503411546689)
_PARAMS = ("g_leak_Cl", "g_leak_Na", "g_leak_K", "Cm", "stim_amplitude",
           "I_ch_Na", "I_ch_K", "I_ch_Cl", "m_K", "m_Na", "I_max", "K_e_init", "K_i_init",
           "K_e", "K_i", "Na_e", "Na_i", "Cl_e", "Cl_i", "z_Na", "z_K", "z_Cl", "psi")
_PARAM_INIT = dict(g_leak_Cl=0.05, g_leak_Na=0.1, g_leak_K=1.696, m_K=1.5, m_Na=10.0,
                   I_max=10.75975, K_e_init=3.092970607490389, K_i_init=99.3100014897692)


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
