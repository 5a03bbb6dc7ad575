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
    lib.lsoda_seq_host.argtypes = [C.c_int, dp, dp, C.c_double, C.c_double, C.c_double, C.c_double, ip]
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
        # The side-effect currents are those of LSODA's LAST internal right-hand-side call, at a time beyond
        # t + dt that the step-size controller chose.  Its error estimates are differences of nearly equal numbers,
        # so two faithful implementations (another libm is enough) place that point ~1e-7 dt apart while taking
        # identical decisions; during the upstroke of an action potential (dI/dt dt ~ 10 I) that is ~1e-6 of the
        # current.  States (interpolated back to t + dt) agree to 1e-9 above.
        assert np.abs(p[sl] - gold[ns:]).max() <= 1e-5 * max(np.abs(gold[ns:]).max(), 1e-3)


@pytest.mark.parametrize("key", ["hh_si_stim0", "hh_si_stim10", "hh_mv_stim0", "hh_mv_stim1", "glial_stim0"])
def test_flat_integrator_equals_sequential_restatement_bit_for_bit(host, key):
    """The product integrator is a flattened phase machine (one right-hand-side evaluation per trip, Nordsieck array in
    registers with zero rows above the order); oracle/lsoda_seq.h keeps ODEPACK's loop nest.  Same arithmetic, so the
    host builds agree in every bit of the states and of the parameter row (side-effect currents), and in all counters."""
    g = np.load(os.path.join(HERE, "golden", "ode_models.npz"))
    mid = MODEL_ID[key.rsplit("_stim", 1)[0]]
    y, p = g[f"{key}_y0"].copy(), g[f"{key}_p0"].copy()
    y2, p2 = y.copy(), p.copy()
    dt = float(g[f"{key}_dt"])
    for k in range(40):
        s1, s2 = (C.c_int * 5)(), (C.c_int * 5)()
        r1 = host.lsoda_host(mid, _ptr(y), _ptr(p), k * dt, (k + 1) * dt, 1e-8, 1e-10, s1)
        r2 = host.lsoda_seq_host(mid, _ptr(y2), _ptr(p2), k * dt, (k + 1) * dt, 1e-8, 1e-10, s2)
        assert r1 == r2 == 0 and list(s1) == list(s2)
        assert np.array_equal(y.view(np.uint64), y2.view(np.uint64))
        assert np.array_equal(p.view(np.uint64), p2.view(np.uint64))


@pytest.mark.parametrize("case", ["stiff", "loose", "tight", "long", "nan", "zero_interval"])
def test_flat_integrator_equals_sequential_on_hard_cases(host, case):
    """Method switch to BDF with Jacobians, loose / tight tolerances (orders 1..12, failed error tests, corrector
    failures), one long interval, and the failure returns."""
    g = np.load(os.path.join(HERE, "golden", "ode_models.npz"))
    y, p = g["hh_si_stim10_y0"].copy(), g["hh_si_stim10_p0"].copy()
    t0, t1, rtol, atol = 0.0, 5e-3, 1e-8, 1e-10
    if case == "stiff":
        p[7] = 2e-7
    elif case == "loose":
        rtol, atol = 1e-3, 1e-5
    elif case == "tight":
        rtol, atol, t1 = 1e-13, 1e-15, 2e-2
    elif case == "long":
        t1 = 0.2
    elif case == "nan":
        y[3] = np.nan
    elif case == "zero_interval":
        t0 = t1 = 1.0
    y2, p2 = y.copy(), p.copy()
    s1, s2 = (C.c_int * 5)(), (C.c_int * 5)()
    r1 = host.lsoda_host(0, _ptr(y), _ptr(p), t0, t1, rtol, atol, s1)
    r2 = host.lsoda_seq_host(0, _ptr(y2), _ptr(p2), t0, t1, rtol, atol, s2)
    assert r1 == r2
    if case in ("nan", "zero_interval"):
        assert r1 != 0
        return
    assert r1 == 0 and list(s1) == list(s2)
    assert np.array_equal(y.view(np.uint64), y2.view(np.uint64)) and np.array_equal(p.view(np.uint64), p2.view(np.uint64))
    if case == "stiff":
        assert s1[3] == 2 and s1[2] > 0
    if case == "tight":
        assert s1[1] > 200


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
