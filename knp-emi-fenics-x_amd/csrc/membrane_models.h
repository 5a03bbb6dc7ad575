// Membrane-model right-hand sides, ahead-of-time compiled for the device.
//
// The reference ships these as Gotran-generated Python modules whose `rhs_numba` is a numba
// `cfunc(lsoda_sig)` handed to numbalsoda by address (src/knpemi/odeSolver.py:96).  A GPU needs
// device code, so the three in-tree models are restated here with the parameter-row layout,
// units and quirks of the originals:
//   HHSI   examples/idealized_geometries/mm_hh.py:139-227              (V, s, S/m^2)
//   HHMV   examples/local_astrocyte_depolarization/mm_hh.py:130-201     (mV, ms, mS/cm^2)
//   Glial  examples/local_astrocyte_depolarization/mm_glial.py:133-205  (mV, ms)
// Every RHS stores the ionic currents I_ch_Na/K/Cl into the parameter row as a side effect,
// like the originals (mm_hh.py:220-225).
#pragma once

#include "lsoda_core.h"

struct ModelHHSI {
  static constexpr int NS = 4, NP = 22;
  KN_HD static void rhs(double t, const double* y, double* dy, double* p) {
    const double m = y[0], h = y[1], n = y[2], V = y[3];
    const double psi = p[21], zK = p[19];
    // both Nernst potentials use z_K, as the reference does (mm_hh.py:169-170)
    const double E_Na = 1.0 / psi * 1.0 / zK * log(p[11] / p[12]);
    const double E_K = 1.0 / psi * 1.0 / zK * log(p[9] / p[10]);
    const double u = 1.0e3 * (V + 65.0e-3);
    const double am = 0.1e3 * (25. - u) / (exp((25. - u) / 10.) - 1);
    const double bm = 4.e3 * exp(-u / 18.);
    const double ah = 0.07e3 * exp(-u / 20.);
    const double bh = 1.e3 / (exp((30. - u) / 10.) + 1);
    const double an = 0.01e3 * (10. - u) / (exp((10. - u) / 10.) - 1.);
    const double bn = 0.125e3 * exp(-u / 80.);
    dy[0] = (1 - m) * am - m * bm;
    dy[1] = (1 - h) * ah - h * bh;
    dy[2] = (1 - n) * an - n * bn;
    const double i_stim = p[8] * exp(-fmod(t, 0.03) / 0.002) * (t < 125e-3 ? 1.0 : 0.0);
    const double a1 = 1 + p[4] / p[9], a2 = 1 + p[5] / p[12];
    const double i_pump = p[6] / ((a1 * a1) * (a2 * a2 * a2));
    const double i_Na = (p[2] + p[0] * h * (m * m * m) + i_stim) * (V - E_Na) + 3 * i_pump;
    const double n2 = n * n;
    const double i_K = (p[3] + p[1] * (n2 * n2)) * (V - E_K) - 2 * i_pump;
    p[15] = i_Na;
    p[16] = i_K;
    p[17] = 0.0;
    dy[3] = (-i_K - i_Na) / p[7];
  }
};

struct ModelHHMV {
  static constexpr int NS = 4, NP = 22;
  KN_HD static void rhs(double t, const double* y, double* dy, double* p) {
    const double m = y[0], h = y[1], n = y[2], V = y[3];
    const double psi = p[21], zK = p[19];
    const double E_Na = 1.0 / psi * 1.0 / zK * log(p[11] / p[12]);
    const double E_K = 1.0 / psi * 1.0 / zK * log(p[9] / p[10]);
    const double u = V + 65.0;
    const double am = 0.1 * (25. - u) / (exp((25. - u) / 10.) - 1);
    const double bm = 4. * exp(-u / 18.);
    const double ah = 0.07 * exp(-u / 20.);
    const double bh = 1. / (exp((30. - u) / 10.) + 1);
    const double an = 0.01 * (10. - u) / (exp((10. - u) / 10.) - 1.);
    const double bn = 0.125 * exp(-u / 80.);
    dy[0] = (1 - m) * am - m * bm;
    dy[1] = (1 - h) * ah - h * bh;
    dy[2] = (1 - n) * an - n * bn;
    const double i_stim = p[8] * exp(-fmod(t, 30.0) / 2.0) * (t < 125 ? 1.0 : 0.0);
    const double a1 = 1 + p[4] / p[9], a2 = 1 + p[5] / p[12];
    const double i_pump = p[6] / ((a1 * a1) * (a2 * a2 * a2));
    const double i_Na = (p[2] + p[0] * h * (m * m * m) + i_stim) * (V - E_Na) + 3 * i_pump;
    const double n2 = n * n;
    const double i_K = (p[3] + p[1] * (n2 * n2)) * (V - E_K) - 2 * i_pump;
    p[15] = i_Na;
    p[16] = i_K;
    p[17] = 0.0;
    dy[3] = (-i_K - i_Na) / p[7];
  }
};

struct ModelGlial {
  static constexpr int NS = 1, NP = 23;
  KN_HD static void rhs(double t, const double* y, double* dy, double* p) {
    (void)t;
    const double V = y[0];
    const double psi = p[22], zK = p[20], zCl = p[21];
    const double E_Na = 1.0 / psi * 1.0 / zK * log(p[15] / p[16]);
    const double E_K = 1.0 / psi * 1.0 / zK * log(p[13] / p[14]);
    const double E_Cl = 1.0 / psi * 1.0 / zCl * log(p[17] / p[18]);
    const double temperature = 307e3, R = 8.315e3, F = 96500e3;  // hard-coded in mm_glial.py:168-170
    const double na15 = p[16] * sqrt(p[16]), mna15 = p[9] * sqrt(p[9]);
    const double i_pump = p[10] * (p[13] / (p[13] + p[8])) * (na15 / (na15 + mna15));
    const double E_K_init = R * temperature / F * log(p[11] / p[12]);
    const double dphi = V - E_K;
    const double A = 1 + exp(18.5 / 42.4);
    const double B = 1 + exp(-(118.6 + E_K_init) / 44.1);
    const double C = 1 + exp((dphi + 18.5) / 42.4);
    const double D = 1 + exp(-(118.6 + V) / 44.1);
    const double g_Kir = sqrt(p[13] / p[11]) * (A * B) / (C * D);
    const double i_Kir = p[2] * g_Kir * (V - E_K);
    const double i_Na = p[1] * (V - E_Na) + 3 * i_pump;
    const double i_K = i_Kir - 2 * i_pump;
    const double i_Cl = p[0] * (V - E_Cl);
    p[5] = i_Na;
    p[6] = i_K;
    p[7] = i_Cl;
    dy[0] = (-i_K - i_Na - i_Cl) / p[3];
  }
};
