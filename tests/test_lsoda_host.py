"""CPU tests of the PRODUCT integrator (csrc/lsoda_core.h + membrane_models.h, host build in
tests/native) against ODEPACK's LSODA (scipy.integrate.odeint) and the golden trajectories."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from scipy.integrate import odeint

import knpemi_oracle as o

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL_ID = {"hh_si": 0, "hh_mv": 1, "glial": 2}


@pytest.fixture(scope="module")
def host():
    so = os.path.join(HERE, "native", "_build", "liblsoda_host.so")
    src = os.path.join(HERE, "native", "lsoda_host_check.cpp")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    lib.lsoda_host.argtypes = [C.c_int, dp, dp, C.c_double, C.c_double, C.c_double, C.c_double, ip]
    lib.lsoda_host_rhs.argtypes = [C.c_int, C.c_double, dp, dp, dp]
    lib.lsoda_host_coef.argtypes = [dp, dp]
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_cfode_tables(host):
    elco, tesco = np.zeros(2 * 13 * 14), np.zeros(2 * 13 * 4)
    host.lsoda_host_coef(_ptr(elco), _ptr(tesco))
    elco, tesco = elco.reshape(2, 13, 14), tesco.reshape(2, 13, 4)
    # Adams-Moulton order 1 (trapezoid family) and 2; BDF1 / BDF2 leading coefficients
    assert np.allclose(elco[0, 1, 1:3], [1, 1]) and np.allclose(elco[0, 2, 1:4], [0.5, 1, 0.5])
    assert np.allclose(elco[1, 1, 1:3], [1, 1]) and np.allclose(elco[1, 2, 1:4], [2 / 3, 1, 1 / 3])
    assert np.isclose(tesco[0, 1, 2], 2.0) and np.isclose(tesco[1, 1, 2], 2.0) and np.isclose(tesco[1, 2, 2], 4.5)
    assert np.allclose(elco[1, 5, 1], 60 / 137)


@pytest.mark.parametrize("key", ["hh_si_stim0", "hh_si_stim10", "hh_mv_stim0", "hh_mv_stim1", "glial_stim0"])
def test_rhs_and_trajectory_match_odepack(host, key):
    g = np.load(os.path.join(HERE, "golden", "ode_models.npz"))
    model = key.rsplit("_stim", 1)[0]
    mid = MODEL_ID[model]
    M = o.MODELS[model]
    y, p = g[f"{key}_y0"].copy(), g[f"{key}_p0"].copy()
    dy, p2 = np.zeros_like(y), p.copy()
    host.lsoda_host_rhs(mid, 0.0, _ptr(y), _ptr(dy), _ptr(p2))
    ref = g[f"{key}_rhs0"]
    assert np.abs(dy - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1.0)
    dt = float(g[f"{key}_dt"])
    ys, ps = y.copy(), p.copy()
    ns = len(y)
    for k in range(10):
        stats = (C.c_int * 5)()
        assert host.lsoda_host(mid, _ptr(y), _ptr(p), k * dt, (k + 1) * dt, 1e-8, 1e-10, stats) == 0
        sol, info = odeint(M["rhs"], ys, [k * dt, (k + 1) * dt], args=(ps,), rtol=1e-8, atol=1e-10, full_output=True)
        ys = sol[-1].copy()
        # same algorithm: identical step / RHS-evaluation / Jacobian counts as ODEPACK
        assert (stats[1], stats[0], stats[2]) == (info["nst"][-1], info["nfe"][-1], info["nje"][-1])
        gold = g[f"{key}_traj"][k]
        assert np.abs(y - gold[:ns]).max() <= 1e-9 * np.abs(gold[:ns]).max()
        sl = o._ich_slice(len(p))
        assert np.abs(p[sl] - gold[ns:]).max() <= 1e-6 * max(np.abs(gold[ns:]).max(), 1e-3)


def test_stiff_switch_to_bdf(host):
    """A stiff parameter set (tiny membrane capacitance) makes LSODA switch to BDF; the host build of
    the product integrator follows ODEPACK through the switch."""
    g = np.load(os.path.join(HERE, "golden", "ode_models.npz"))
    y, p = g["hh_si_stim10_y0"].copy(), g["hh_si_stim10_p0"].copy()
    p[7] = 2e-7     # Cm: V relaxes 1e5 times faster than the gates
    ys, ps = y.copy(), p.copy()
    stats = (C.c_int * 5)()
    assert host.lsoda_host(0, _ptr(y), _ptr(p), 0.0, 5e-3, 1e-8, 1e-10, stats) == 0
    sol, info = odeint(o.rhs_hh_si, ys, [0.0, 5e-3], args=(ps,), rtol=1e-8, atol=1e-10, full_output=True,
                       mxstep=10000)
    assert info["mused"][-1] == 2 and stats[3] == 2 and stats[2] > 0       # BDF with Jacobians
    assert np.abs(y - sol[-1]).max() <= 1e-6 * np.abs(sol[-1]).max()
