"""`MembraneModel`: the membrane ODE systems, integrated on the GPU.

Drop-in for `src/knpemi/odeSolver.py` of the reference: same constructor,
attributes (`states`, `parameters`, `dof_locations`, `indices`, `nodes`, `tag`,
`ode`, `prefix`, `time`) and methods.  `step_lsoda` replaces the reference's
serial Python loop over numbalsoda calls (`odeSolver.py:107-122`) by one launch
of the HIP kernel `ode_step_kernel` (csrc/kernels_ode.hip): one thread per
membrane dof running LSODA.

Membrane-model plug-ins keep the reference's module protocol
(`init_state_values`, `init_parameter_values`, `state_indices`,
`parameter_indices`, `__name__`).  Where the reference takes the address of a
numba `cfunc` (`rhs_numba.address`, `odeSolver.py:96`) -- a host function a GPU
cannot call -- a plug-in here either names one of the right-hand sides compiled
into the library (`MODEL_ID` = "hh_si", "hh_mv" or "glial") or brings its own
as HIP source in `RHS_HIP`: a `__device__ void rhs(double t, const double*
states, double* values, double* parameters)` with the cfunc's semantics, compiled
for gfx950 with hipRTC when the model is bound (csrc/kernels_rtc.hip) -- or has
only what the reference's modules have, a Python `rhs_numba` / `rhs` made of
assignments: `knpemi.rhs_codegen` translates its source text into that function.
"""
from __future__ import annotations

import numpy as np

from . import _lib as L

_MODEL_IDS = {"hh_si": L.MODEL_HH_SI, "hh_mv": L.MODEL_HH_MV, "glial": L.MODEL_GLIAL}


class MembraneModel:
    '''ODE on membrane defined by tagged facet function'''

    def __init__(self, ode, ft, tag, Q):
        assert isinstance(tag, int)
        # All dofs of Q carry an ODE system (odeSolver.py:31-38: the restriction
        # to ft.find(tag) is commented out in the reference).
        self.dof_locations = Q.tabulate_dof_coordinates()
        self.indices = np.arange(len(self.dof_locations))
        nodes = len(self.indices)
        self.nodes = nodes
        s0 = np.asarray(ode.init_state_values(), dtype=np.float64)
        p0 = np.asarray(ode.init_parameter_values(), dtype=np.float64)
        self.states = np.tile(s0, (nodes, 1))
        self.parameters = np.tile(p0, (nodes, 1))
        self.tag = tag
        self.ode = ode
        self.prefix = ode.__name__
        self.time = 0
        self.rtol, self.atol = 1.0e-8, 1.0e-10   # odeSolver.py:120
        self.last_stats = None
        # device binding (set by DeviceProblem through emi_system / knp_system)
        self._dp = None
        self._sub = None
        self._model = None
        self._pending_flags = 0
        self._ion_param = None
        self._mask_cache = {}
        print(f'\t{self.prefix} Number of ODE points on the membrane {nodes}')

    # --- device binding ------------------------------------------------------
    def _bind(self, dp, sub, model, ion_names):
        if self._dp is dp:
            return
        model_id = getattr(self.ode, "MODEL_ID", None)
        source = getattr(self.ode, "RHS_HIP", None)
        if model_id not in _MODEL_IDS and source is None:
            # a module written for the reference: its Python right-hand side (rhs_numba / rhs, odeSolver.py:96 takes the
            # cfunc's address) is plain arithmetic -- translate the source text into the device function
            from .rhs_codegen import hip_source_from_module
            source = hip_source_from_module(self.ode)
        if model_id in _MODEL_IDS:
            L.check(dp.lib.knpemi_ode_bind(dp.h, sub, model, _MODEL_IDS[model_id],
                                           self.states.shape[1], self.parameters.shape[1]))
        elif source is not None:
            # the plug-in's own right-hand side, compiled for gfx950 now (a few seconds, cached per process)
            L.check(dp.lib.knpemi_ode_bind_source(dp.h, sub, model, self.states.shape[1], self.parameters.shape[1],
                                                  source.encode()))
        else:
            raise NotImplementedError(
                f"membrane model module '{self.prefix}' has neither a MODEL_ID naming a shipped device RHS "
                f"(one of {sorted(_MODEL_IDS)}), nor its own RHS_HIP source, nor a Python right-hand side (rhs_numba / "
                f"rhs) whose source knpemi.rhs_codegen could translate (see examples/benchmark/mm_glial.py)")
        self._dp, self._sub, self._model = dp, sub, model
        idx = []
        for name in ion_names:
            idx += [self.ode.parameter_indices(f"{name}_e"), self.ode.parameter_indices(f"{name}_i"),
                    self.ode.parameter_indices(f"I_ch_{name}")]
        self._ion_param = np.array(idx, np.int32)

    # --- PDE <-> ODE column copies (odeSolver.py:52-85, 130-188) -------------------
    def _table(self, what):
        if what == 'state':
            return self.states, self.ode.state_indices
        return self.parameters, self.ode.parameter_indices

    def _rows(self, locator):
        if locator is None:
            return slice(None)
        return np.flatnonzero(np.fromiter(map(locator, self.dof_locations), dtype=bool))

    def _from_function(self, what, which, u, locator):
        table, col_of = self._table(what)
        rows = self._rows(locator)
        table[rows, col_of(which)] = u.x.array[self.indices[rows]]
        return self.states

    def _to_function(self, what, which, u, locator):
        table, col_of = self._table(what)
        rows = self._rows(locator)
        u.x.array[self.indices[rows]] = table[rows, col_of(which)]
        return u

    def _from_callables(self, what, value_dict, locator):
        table, col_of = self._table(what)
        rows = np.arange(self.nodes)[self._rows(locator)]
        print(f'\t{self.prefix} Set {what} for {len(rows)} ODES')
        for name, fn in value_dict.items():
            if len(rows):
                table[rows, col_of(name)] = [fn(x) for x in self.dof_locations[rows]]
        return table

    def set_state(self, which, u, locator=None):
        return self._from_function('state', which, u, locator)

    def set_parameter(self, which, u, locator=None):
        return self._from_function('parameter', which, u, locator)

    def get_state(self, which, u, locator=None):
        return self._to_function('state', which, u, locator)

    def get_parameter(self, which, u, locator=None):
        return self._to_function('parameter', which, u, locator)

    def set_state_values(self, value_dict, locator=None):
        return self._from_callables('state', value_dict, locator)

    def set_parameter_values(self, value_dict, locator=None):
        return self._from_callables('parameter', value_dict, locator)

    def set_membrane_potential(self, u, locator=None):
        return self.set_state('V', u, locator=locator)

    def get_membrane_potential(self, u, locator=None):
        return self.get_state('V', u, locator=locator)

    @property
    def V_index(self):
        return self.ode.state_indices('V')

    # ---- ODE integration ------
    def step_lsoda(self, dt, stimulus, stimulus_locator=None):
        '''Solve the ODEs forward by dt with optional stimulus (on the GPU)'''
        if self._dp is None:
            raise RuntimeError(
                "MembraneModel is not attached to a device problem: build the forms with "
                "emi_system()/knp_system() first (the ODE sweep has no CPU fallback)")
        if stimulus is None:
            stimulus = {}
        dp, lib = self._dp, self._dp.lib
        # keyed on the locator object itself (the cache holds a reference, so a fresh lambda can never reuse the id
        # of a dead one and pick up its mask)
        key = stimulus_locator
        if key not in self._mask_cache:
            if stimulus_locator is None:
                mask = np.ones(self.nodes, np.uint8)
            else:
                mask = np.fromiter(map(stimulus_locator, self.dof_locations), dtype=bool).astype(np.uint8)
            self._mask_cache = {key: np.ascontiguousarray(mask)}
        mask = self._mask_cache[key]
        sidx = np.array([self.ode.parameter_indices(k) for k in stimulus], np.int32)
        sval = np.array([float(v) for v in stimulus.values()], np.float64)
        L.check(lib.knpemi_ode_set_stimulus(dp.h, self._sub, self._model, mask.ctypes.data_as(L.c_u8_p),
                                            len(sidx), L.iptr(sidx) if len(sidx) else None,
                                            L.dptr(sval) if len(sval) else None))
        print(f'\t{self.prefix} Stepping {self.nodes} ODEs')
        states = np.ascontiguousarray(self.states, np.float64)
        params = np.ascontiguousarray(self.parameters, np.float64)
        L.check(lib.knpemi_ode_set_tables(dp.h, self._sub, self._model, L.dptr(states), L.dptr(params)))
        dp.timer_start()
        L.check(lib.knpemi_ode_step(dp.h, self._sub, self._model, float(self.time), float(dt),
                                    self.rtol, self.atol, int(self._pending_flags),
                                    L.iptr(self._ion_param), int(self.V_index)))
        ms = dp.timer_stop_ms()
        self._pending_flags = 0
        import ctypes as C
        nrhs, nst, nfail = C.c_int64(), C.c_int64(), C.c_int32()
        rc = lib.knpemi_ode_stats(dp.h, self._sub, self._model, C.byref(nrhs), C.byref(nst), C.byref(nfail))
        self.last_stats = dict(n_rhs=nrhs.value, n_steps=nst.value, n_failed=nfail.value, ms=ms)
        L.check(lib.knpemi_ode_get_tables(dp.h, self._sub, self._model, L.dptr(states), L.dptr(params)))
        self.states[...] = states
        self.parameters[...] = params
        assert rc == L.OK, "LSODA failed on at least one membrane dof"   # odeSolver.py:121
        self.time = self.time + dt
        print(f'\t{self.prefix} Stepped {self.nodes} ODES in {ms * 1e-3}s')
        return self.states
