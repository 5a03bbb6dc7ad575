"""ctypes binding of libknpemi_hip.so (C ABI: include/knpemi_hip.h).

There is no CPU fallback: if the shared library is missing, or no HIP device is
visible when a device problem is created, the hot path raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KNPEMI_HIP_LIB") or os.path.join(_HERE, "libknpemi_hip.so")   # override: build experiments

OK, EINVAL, EHIP, ENOMEM, EODE, ESOLVE = 0, -1, -2, -3, -4, -5
PC_JACOBI, PC_AMG = 0, 1
TRIANGLE, TETRAHEDRON, HEXAHEDRON = 0, 1, 2
MODEL_HH_SI, MODEL_HH_MV, MODEL_GLIAL = 0, 1, 2
MAX_IONS, MAX_SUB, MAX_MODELS = 4, 8, 4
F_PHI, F_C, F_C_PREV, F_C_ELIM, F_PHI_M, F_I_CH, F_SOURCE = range(7)
A_EMI, P_EMI, A_KNP = 0, 1, 2
B_EMI, B_KNP = 0, 1
WANT_P, NO_SPLITTING, SKIP_MEMBRANE_RHS, ON_AUX_STREAM, MEMBRANE_EARLY = 1, 2, 4, 8, 16
ODE_SET_V, ODE_SET_TRACES, ODE_ON_AUX, ODE_ON_AUX2 = 1, 2, 4, 8
OPT_FUSE_UPDATE, OPT_FUSE_MEMBRANE, OPT_PROFILE_STRIDE, OPT_KNP_MIN_IT, OPT_FOLD_MEMBRANE, OPT_KNP_METHOD = 1, 2, 3, 4, 5, 6
OPT_EMI_NORM = 7
K_ODE, K_EMI_ROWS, K_KNP_ROWS, K_KNP_MEMBRANE, K_UPDATE, K_EMI_MEMBRANE = range(6)
KERNEL_NAMES = ["ode_step_kernel", "emi_rows_kernel", "knp_rows_kernel", "knp_membrane_kernel", "update_pde_kernel",
                "emi_membrane_rhs_kernel"]

c_int_p = C.POINTER(C.c_int32)
c_dbl_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)


class ProblemDesc(C.Structure):
    _fields_ = [
        ("gdim", C.c_int32), ("cell_kind", C.c_int32), ("n_sub", C.c_int32), ("n_ions", C.c_int32),
        ("n_vert", c_int_p), ("n_cell", c_int_p),
        ("x", C.POINTER(c_dbl_p)), ("cells", C.POINTER(c_int_p)),
        ("n_q", c_int_p), ("n_facet", c_int_p),
        ("facet_e", C.POINTER(c_int_p)), ("facet_i", C.POINTER(c_int_p)),
        ("facet_q", C.POINTER(c_int_p)), ("facet_model", C.POINTER(c_int_p)),
        ("q_to_e", C.POINTER(c_int_p)), ("q_to_i", C.POINTER(c_int_p)),
        ("n_models", c_int_p),
        ("uniform_cell", C.c_double * 9),
    ]


class Params(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("F", C.c_double), ("psi", C.c_double), ("C_M", C.c_double),
        ("z", C.c_double * MAX_IONS),
        ("D", (C.c_double * MAX_IONS) * MAX_SUB),
        ("rho_z", C.c_double),
        ("rho", C.c_double * MAX_SUB),
        ("C_phi", C.c_double),
    ]


class DGDesc(C.Structure):
    _fields_ = [
        ("cell_kind", C.c_int32), ("n_sub", C.c_int32), ("n_ions", C.c_int32),
        ("n_cells", C.c_int64), ("n_vertices", C.c_int64), ("n_mem_facets", C.c_int64),
        ("x", c_dbl_p), ("cells", c_int_p), ("cell_sub", c_int_p), ("mem_facets", c_int_p),
    ]


class DGParams(C.Structure):
    _fields_ = [
        ("dt", C.c_double), ("F", C.c_double), ("psi", C.c_double), ("C_M", C.c_double), ("gamma", C.c_double),
        ("z", C.c_double * MAX_IONS),
        ("D", (C.c_double * MAX_IONS) * MAX_SUB),
        ("rho_z", C.c_double),
        ("rho", C.c_double * MAX_SUB),
    ]


DG_C, DG_PHI, DG_PHI_M, DG_I_CH, DG_SOURCE = range(5)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)


class KnpemiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libknpemi_hip error {code}: {msg}")
        self.code = code


_lib = None

# name -> (restype, argtypes); every symbol include/knpemi_hip.h declares
SIGNATURES = {
    "knpemi_last_error": (C.c_char_p, []),
    "knpemi_device_count": (C.c_int, []),
    "knpemi_create": (C.c_int, [C.POINTER(ProblemDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "knpemi_destroy": (None, [C.c_void_p]),
    "knpemi_set_params": (C.c_int, [C.c_void_p, C.POINTER(Params)]),
    "knpemi_sync": (C.c_int, [C.c_void_p]),
    "knpemi_set_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_dbl_p, C.c_size_t]),
    "knpemi_get_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_dbl_p, C.c_size_t]),
    "knpemi_assemble_emi": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_assemble_knp": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_assemble_emi_membrane_rhs": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_assemble_knp_membrane_early": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_join": (C.c_int, [C.c_void_p]),
    "knpemi_solve_emi": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), c_dbl_p]),
    "knpemi_solve_knp": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), c_dbl_p]),
    "knpemi_extrapolate_guess": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_solver_setup": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double]),
    "knpemi_solver_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), c_dbl_p, C.POINTER(C.c_int)]),
    "knpemi_csr_dims": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "knpemi_get_csr_pattern": (C.c_int, [C.c_void_p, C.c_int, c_int_p, c_int_p]),
    "knpemi_get_csr_values": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p]),
    "knpemi_get_rhs": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p]),
    "knpemi_set_csr_values": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p]),
    "knpemi_set_rhs": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p]),
    "knpemi_device_csr": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_void_p)]),
    "knpemi_device_rhs": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "knpemi_set_solution": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "knpemi_get_solution": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p]),
    "knpemi_ode_bind": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "knpemi_ode_bind_source": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "knpemi_ode_compile_source": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]),
    "knpemi_ode_set_tables": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dbl_p, c_dbl_p]),
    "knpemi_ode_get_tables": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dbl_p, c_dbl_p]),
    "knpemi_ode_set_stimulus": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_u8_p, C.c_int, c_int_p, c_dbl_p]),
    "knpemi_ode_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_int, c_int_p, C.c_int]),
    "knpemi_ode_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "knpemi_debug_ode_stamps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_int]),
    "knpemi_debug_math": (C.c_int, [C.c_int, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p]),
    "knpemi_debug_launch_chain": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_dbl_p]),
    "knpemi_debug_geometry": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "knpemi_update_pde": (C.c_int, [C.c_void_p]),
    "knpemi_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "knpemi_trace": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]),
    "knpemi_halo_width": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_halo_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "knpemi_halo_unpack": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "knpemi_set_distributed": (C.c_int, [C.c_void_p, c_u8_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "knpemi_set_distributed_coarse": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "knpemi_vec_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "knpemi_vec_scatter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "knpemi_comm_unique_id": (C.c_int, [C.c_char_p, C.c_size_t]),
    "knpemi_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "knpemi_comm_sendrecv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_int_p, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "knpemi_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "knpemi_comm_set_vector_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_int, c_int_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                              C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "knpemi_comm_allreduce_hook": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_comm_halo_hook": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "knpemi_profile": (C.c_int, [C.c_void_p, C.c_uint32]),
    "knpemi_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), c_dbl_p]),
    "knpemi_timer_start": (C.c_int, [C.c_void_p]),
    "knpemi_timer_stop_ms": (C.c_int, [C.c_void_p, c_dbl_p]),
    "knpemi_stream": (C.c_void_p, [C.c_void_p]),
    # DG(P1) + interior penalty variant
    "knpemi_dg_create": (C.c_int, [C.POINTER(DGDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "knpemi_dg_destroy": (None, [C.c_void_p]),
    "knpemi_dg_set_params": (C.c_int, [C.c_void_p, C.POINTER(DGParams)]),
    "knpemi_dg_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "knpemi_dg_get_pattern": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "knpemi_dg_get_membrane_dofs": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "knpemi_dg_set_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dbl_p, C.c_size_t]),
    "knpemi_dg_get_field": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_dbl_p, C.c_size_t]),
    "knpemi_dg_assemble_emi": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_dg_assemble_knp": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_dg_get_values": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p]),
    "knpemi_dg_get_rhs": (C.c_int, [C.c_void_p, C.c_int, c_dbl_p]),
    "knpemi_dg_device_system": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "knpemi_dg_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "knpemi_dg_solve_emi": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), c_dbl_p]),
    "knpemi_dg_solve_knp": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), c_dbl_p, C.c_int]),
    "knpemi_dg_get_solution": (C.c_int, [C.c_void_p, c_dbl_p]),
    "knpemi_dg_set_extrapolation": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_dg_ode_bind": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_dbl_p, c_dbl_p, c_int_p, C.c_int]),
    "knpemi_dg_ode_step": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    "knpemi_dg_ode_get_tables": (C.c_int, [C.c_void_p, c_dbl_p, c_dbl_p]),
    "knpemi_dg_ode_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "knpemi_dg_halo_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "knpemi_dg_halo_unpack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "knpemi_dg_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "knpemi_dg_comm_sendrecv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_int_p, C.POINTER(C.c_int64),
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "knpemi_dg_sync": (C.c_int, [C.c_void_p]),
    "knpemi_dg_time_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_dbl_p]),
    "knpemi_dg_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "knpemi_dg_profile_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), c_dbl_p]),
    "knpemi_dg_stream": (C.c_void_p, [C.c_void_p]),
    "knpemi_dg_set_distributed": (C.c_int, [C.c_void_p, c_u8_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "knpemi_dg_solver_handle": (C.c_void_p, [C.c_void_p]),
}


def load():
    """Load libknpemi_hip.so (built in-tree by `__graft_entry__.build()` / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C knp-emi-fenics-x_amd/csrc). "
            "The knpemi hot path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        msg = load().knpemi_last_error()
        raise KnpemiError(rc, msg.decode() if msg else "unknown error")


def dptr(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(c_dbl_p)


def iptr(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_int_p)
