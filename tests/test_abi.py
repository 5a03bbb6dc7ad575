"""CPU tests of the drop-in boundary: the library loads, exports every declared symbol and
refuses to run without a device (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hip_lib):
    from knpemi import _lib
    header = open(os.path.join(ROOT, "include", "knpemi_hip.h")).read()
    declared = set(re.findall(r"\b(knpemi_[a-z_]+)\s*\(", header))
    declared.discard("knpemi_handle")
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(hip_lib, name), f"{name} declared in knpemi_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"


def test_constants_match_header():
    from knpemi import _lib as L
    header = open(os.path.join(ROOT, "include", "knpemi_hip.h")).read()
    consts = dict(re.findall(r"#define\s+KNPEMI_([A-Z_0-9]+)\s+\(?(-?\d+)\)?", header))
    for name in ("F_PHI", "F_C", "F_C_PREV", "F_C_ELIM", "F_PHI_M", "F_I_CH", "F_SOURCE", "A_EMI", "P_EMI",
                 "A_KNP", "B_EMI", "B_KNP", "WANT_P", "NO_SPLITTING", "ODE_SET_V", "ODE_SET_TRACES",
                 "TRIANGLE", "TETRAHEDRON", "HEXAHEDRON", "MODEL_HH_SI", "MODEL_HH_MV", "MODEL_GLIAL",
                 "EINVAL", "EHIP", "EODE", "MAX_IONS", "MAX_SUB", "K_ODE", "K_EMI_ROWS", "K_UPDATE"):
        assert int(consts[name]) == getattr(L, name), name
    assert C.sizeof(L.Params) == 8 * (4 + L.MAX_IONS + 8 * L.MAX_IONS + 1 + 8 + 1)      # ... + C_phi


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_device_fails_loudly(hip_lib):
    """Without a HIP device nothing is computed anywhere: the ABI returns EHIP and the Python layer raises."""
    from helpers import Setup
    from knpemi import _lib as L
    assert hip_lib.knpemi_device_count() == 0
    n1 = np.array([3], np.int32)
    desc = L.ProblemDesc(gdim=2, cell_kind=L.TRIANGLE, n_sub=1, n_ions=3, n_vert=L.iptr(n1), n_cell=L.iptr(n1))
    h = C.c_void_p()
    assert hip_lib.knpemi_create(C.byref(desc), 0, C.byref(h)) == L.EHIP
    assert b"no HIP device" in hip_lib.knpemi_last_error() and not h.value
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Setup("2d", 1)


def test_null_and_bad_arguments(hip_lib):
    from knpemi import _lib as L
    assert hip_lib.knpemi_create(None, 0, None) == L.EINVAL
    assert hip_lib.knpemi_sync(None) == L.EINVAL
    assert hip_lib.knpemi_update_pde(None) == L.EINVAL
    hip_lib.knpemi_destroy(None)   # harmless
    n1 = np.array([3], np.int32)
    desc = L.ProblemDesc(gdim=2, cell_kind=L.TRIANGLE, n_sub=1, n_ions=5, n_vert=L.iptr(n1), n_cell=L.iptr(n1))
    h = C.c_void_p()
    assert hip_lib.knpemi_create(C.byref(desc), 0, C.byref(h)) == L.EINVAL
    assert b"2 to 4 ionic species" in hip_lib.knpemi_last_error()
    desc.n_ions, desc.gdim = 3, 3
    assert hip_lib.knpemi_create(C.byref(desc), 0, C.byref(h)) == L.EINVAL
