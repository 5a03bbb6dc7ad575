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

#include "ode_kernel.h"   // lsoda_core.h, StridedRow

// Each model is a functor.  `prepare(p)` reads the parameter row once (any indexable row: a plain
// array on the host, a strided view of the transposed device table in the kernel) and caches what the
// RHS needs, including the sub-expressions that depend on parameters only (Nernst potentials, pump
// current); `rhs` is the cfunc body and keeps the side-effect currents in members; `finish(p)` stores
// them into the parameter row, which is what the reference's in-place writes amount to after the
// last call.
// fmod(t, period) for the periodic stimulus: inside the first period the remainder is t itself (fmod is exact, so
// this returns the same bits) and the ~70-instruction reduction loop is skipped -- by the whole wavefront, since all
// membrane dofs of a sweep integrate the same time interval.
// Both exponentials of a lane are evaluated here, side by side: the second one is only used by the gate lanes, and left
// alone the compiler sinks its whole evaluation into that divergent region -- after the first one, so the two dependent
// chains no longer overlap and the polynomial's constants are materialised twice.
#if defined(__HIP_DEVICE_COMPILE__)
#define KN_KEEP_TOGETHER(a, b) asm volatile("" : "+v"(a), "+v"(b))
#else
#define KN_KEEP_TOGETHER(a, b) ((void)0)
#endif

KN_HD double kn_fmod_period(double t, double period) {
  if (t >= 0.0 && t < period) return t;
  return fmod(t, period);
}

struct ModelHHSI {
  static constexpr int NS = 4, NP = 22;
  double E_Na, E_K, i_pump, gNa, gK, glNa, glK, Cm, stim;
  mutable double I_Na, I_K;
  template <class Row>
  KN_HD void prepare(const Row& p) {
    gNa = p[0]; gK = p[1]; glNa = p[2]; glK = p[3]; Cm = p[7]; stim = p[8];
    const double psi = p[21], zK = p[19];
    // both Nernst potentials use z_K, as the reference does (mm_hh.py:169-170)
    E_Na = 1.0 / psi * 1.0 / zK * log(p[11] / p[12]);
    E_K = 1.0 / psi * 1.0 / zK * log(p[9] / p[10]);
    const double a1 = 1 + p[4] / p[9], a2 = 1 + p[5] / p[12];
    i_pump = p[6] / ((a1 * a1) * (a2 * a2 * a2));
  }
  KN_HD void rhs(double t, const double* y, double* dy) const {
    const double m = y[0], h = y[1], n = y[2], V = y[3];
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
    const double i_stim = stim * exp(-kn_fmod_period(t, 0.03) / 0.002) * (t < 125e-3 ? 1.0 : 0.0);
    const double i_Na = (glNa + gNa * h * (m * m * m) + i_stim) * (V - E_Na) + 3 * i_pump;
    const double n2 = n * n;
    const double i_K = (glK + gK * (n2 * n2)) * (V - E_K) - 2 * i_pump;
    I_Na = i_Na;
    I_K = i_K;
    dy[3] = (-i_K - i_Na) / Cm;
  }
  // Component-wise evaluation for the one-lane-per-component integrator: lane c computes dy[c].  The
  // four lanes of a system run the same straight-line instruction stream; what differs per lane (offsets and scales
  // of the two exponent arguments, rate-law variant) is a handful of per-lane constants fixed by `set_lane` before the
  // sweep and selects on c -- no divergence, no branch.  Every lane needs two exponentials and ONE quotient (lanes m
  // and n: alpha = k n1 / (e1 - 1); lane h: beta = 1 / (e2 + 1); lane V divides by the constant C_m, i.e. multiplies by
  // its reciprocal), and the exponent arguments (off - u) / d and the numerator k (off - u), u = s (V + 65 mV), are
  // each ONE fused multiply-add on V with per-lane coefficients: the right-hand side is the longest stretch of
  // every trip of the integrator and its depth in dependent fp64 instructions is what a sweep costs (lsoda_core.h,
  // kn_div).  The expressions are those of `rhs` up to these roundings.  The side-effect currents are those of the
  // lane that owns V (CURRENT_LANE).
  static constexpr int CURRENT_LANE = 3;
  double l_a1, l_b1, l_a2, l_b2, l_ka, l_kb, l_k2, l_rCm;
  KN_HD void set_lane(int c) {
    // exponent arguments and the numerator of alpha as linear functions of V: (off - u) / d with u = s (V + 65 mV)
    const double sv = 1.0e3, off1 = c == 0 ? 25. : (c == 2 ? 10. : 0.), off2 = c == 1 ? 30. : 0.;
    const double rd1 = c == 1 ? 1.0 / 20. : (c == 3 ? 1.0 / 0.002 : 1.0 / 10.);
    const double rd2 = c == 0 ? 1.0 / 18. : (c == 1 ? 1.0 / 10. : 1.0 / 80.);
    const double k1 = c == 0 ? 0.1e3 : 0.01e3;
    l_a1 = c == 3 ? 0.0 : (off1 - 65.) * rd1;
    l_b1 = c == 3 ? -rd1 : -sv * rd1;          // lane V: the argument is -fmod(t, period) / tau
    l_a2 = (off2 - 65.) * rd2;
    l_b2 = -sv * rd2;
    l_ka = k1 * (off1 - 65.);
    l_kb = -k1 * sv;
    l_k2 = c == 0 ? 4.e3 : 0.125e3;
    l_rCm = kn_div(1.0, Cm);
  }
  KN_HD double rhs_lane(int c, double t, const double* y) const {
    const double m = y[0], h = y[1], n = y[2], V = y[3];
    const double tp = kn_fmod_period(t, 0.03);
    double e1 = kn_exp(fma(c == 3 ? tp : V, l_b1, l_a1)), e2 = kn_exp(fma(V, l_b2, l_a2));
    KN_KEEP_TOGETHER(e1, e2);
    const double q = kn_div(c == 1 ? 1.e3 : fma(V, l_kb, l_ka), c == 1 ? e2 + 1 : e1 - 1);
    const double alpha = c == 1 ? 0.07e3 * e1 : q;
    const double beta = c == 1 ? q : l_k2 * e2;
    const double gate = c == 0 ? m : (c == 1 ? h : n);
    const double dgate = (1 - gate) * alpha - gate * beta;
    const double i_stim = stim * e1 * (t < 125e-3 ? 1.0 : 0.0);
    const double i_Na = (glNa + gNa * h * (m * m * m) + i_stim) * (V - E_Na) + 3 * i_pump;
    const double n2q = n * n;
    const double i_K = (glK + gK * (n2q * n2q)) * (V - E_K) - 2 * i_pump;
    I_Na = i_Na;
    I_K = i_K;
    return c == 3 ? (-i_K - i_Na) * l_rCm : dgate;
  }
  template <class Row>
  KN_HD void finish(const Row& p) const { p[15] = I_Na; p[16] = I_K; p[17] = 0.0; }
};

struct ModelHHMV {
  static constexpr int NS = 4, NP = 22;
  double E_Na, E_K, i_pump, gNa, gK, glNa, glK, Cm, stim;
  mutable double I_Na, I_K;
  template <class Row>
  KN_HD void prepare(const Row& p) {
    gNa = p[0]; gK = p[1]; glNa = p[2]; glK = p[3]; Cm = p[7]; stim = p[8];
    const double psi = p[21], zK = p[19];
    E_Na = 1.0 / psi * 1.0 / zK * log(p[11] / p[12]);
    E_K = 1.0 / psi * 1.0 / zK * log(p[9] / p[10]);
    const double a1 = 1 + p[4] / p[9], a2 = 1 + p[5] / p[12];
    i_pump = p[6] / ((a1 * a1) * (a2 * a2 * a2));
  }
  KN_HD void rhs(double t, const double* y, double* dy) const {
    const double m = y[0], h = y[1], n = y[2], V = y[3];
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
    const double i_stim = stim * exp(-kn_fmod_period(t, 30.0) / 2.0) * (t < 125 ? 1.0 : 0.0);
    const double i_Na = (glNa + gNa * h * (m * m * m) + i_stim) * (V - E_Na) + 3 * i_pump;
    const double n2 = n * n;
    const double i_K = (glK + gK * (n2 * n2)) * (V - E_K) - 2 * i_pump;
    I_Na = i_Na;
    I_K = i_K;
    dy[3] = (-i_K - i_Na) / Cm;
  }
  // component-wise evaluation, see ModelHHSI::rhs_lane
  static constexpr int CURRENT_LANE = 3;
  double l_a1, l_b1, l_a2, l_b2, l_ka, l_kb, l_k2, l_rCm;
  KN_HD void set_lane(int c) {
    // exponent arguments and the numerator of alpha as linear functions of V: (off - u) / d with u = s (V + 65 mV)
    const double sv = 1.0, off1 = c == 0 ? 25. : (c == 2 ? 10. : 0.), off2 = c == 1 ? 30. : 0.;
    const double rd1 = c == 1 ? 1.0 / 20. : (c == 3 ? 1.0 / 2.0 : 1.0 / 10.);
    const double rd2 = c == 0 ? 1.0 / 18. : (c == 1 ? 1.0 / 10. : 1.0 / 80.);
    const double k1 = c == 0 ? 0.1 : 0.01;
    l_a1 = c == 3 ? 0.0 : (off1 - 65.) * rd1;
    l_b1 = c == 3 ? -rd1 : -sv * rd1;          // lane V: the argument is -fmod(t, period) / tau
    l_a2 = (off2 - 65.) * rd2;
    l_b2 = -sv * rd2;
    l_ka = k1 * (off1 - 65.);
    l_kb = -k1 * sv;
    l_k2 = c == 0 ? 4. : 0.125;
    l_rCm = kn_div(1.0, Cm);
  }
  KN_HD double rhs_lane(int c, double t, const double* y) const {
    const double m = y[0], h = y[1], n = y[2], V = y[3];
    const double tp = kn_fmod_period(t, 30.0);
    double e1 = kn_exp(fma(c == 3 ? tp : V, l_b1, l_a1)), e2 = kn_exp(fma(V, l_b2, l_a2));
    KN_KEEP_TOGETHER(e1, e2);
    const double q = kn_div(c == 1 ? 1. : fma(V, l_kb, l_ka), c == 1 ? e2 + 1 : e1 - 1);
    const double alpha = c == 1 ? 0.07 * e1 : q;
    const double beta = c == 1 ? q : l_k2 * e2;
    const double gate = c == 0 ? m : (c == 1 ? h : n);
    const double dgate = (1 - gate) * alpha - gate * beta;
    const double i_stim = stim * e1 * (t < 125 ? 1.0 : 0.0);
    const double i_Na = (glNa + gNa * h * (m * m * m) + i_stim) * (V - E_Na) + 3 * i_pump;
    const double n2q = n * n;
    const double i_K = (glK + gK * (n2q * n2q)) * (V - E_K) - 2 * i_pump;
    I_Na = i_Na;
    I_K = i_K;
    return c == 3 ? (-i_K - i_Na) * l_rCm : dgate;
  }
  template <class Row>
  KN_HD void finish(const Row& p) const { p[15] = I_Na; p[16] = I_K; p[17] = 0.0; }
};

struct ModelGlial {
  static constexpr int NS = 1, NP = 23;
  double E_Na, E_K, E_Cl, i_pump, gfac, glCl, glNa, glK, Cm;   // gfac = sqrt(K_e/K_e_init) * A * B
  mutable double I_Na, I_K, I_Cl;
  template <class Row>
  KN_HD void prepare(const Row& p) {
    glCl = p[0]; glNa = p[1]; glK = p[2]; Cm = p[3];
    const double psi = p[22], zK = p[20], zCl = p[21];
    E_Na = 1.0 / psi * 1.0 / zK * log(p[15] / p[16]);
    E_K = 1.0 / psi * 1.0 / zK * log(p[13] / p[14]);
    E_Cl = 1.0 / psi * 1.0 / zCl * log(p[17] / p[18]);
    const double temperature = 307e3, R = 8.315e3, F = 96500e3;  // hard-coded in mm_glial.py:168-170
    const double na15 = p[16] * sqrt(p[16]), mna15 = p[9] * sqrt(p[9]);
    i_pump = p[10] * (p[13] / (p[13] + p[8])) * (na15 / (na15 + mna15));
    const double E_K_init = R * temperature / F * log(p[11] / p[12]);
    const double A = 1 + exp(18.5 / 42.4);
    const double B = 1 + exp(-(118.6 + E_K_init) / 44.1);
    gfac = sqrt(p[13] / p[11]) * (A * B);
  }
  KN_HD void rhs(double t, const double* y, double* dy) const {
    (void)t;
    const double V = y[0];
    const double dphi = V - E_K;
    const double C = 1 + kn_exp(kn_div(dphi + 18.5, 42.4));
    const double D = 1 + kn_exp(kn_div(-(118.6 + V), 44.1));
    const double g_Kir = kn_div(gfac, C * D);
    const double i_Kir = glK * g_Kir * (V - E_K);
    const double i_Na = glNa * (V - E_Na) + 3 * i_pump;
    const double i_K = i_Kir - 2 * i_pump;
    const double i_Cl = glCl * (V - E_Cl);
    I_Na = i_Na;
    I_K = i_K;
    I_Cl = i_Cl;
    dy[0] = kn_div(-i_K - i_Na - i_Cl, Cm);
  }
  static constexpr int CURRENT_LANE = 0;
  KN_HD double rhs_lane(int, double t, const double* y) const {
    double dy[1];
    rhs(t, y, dy);
    return dy[0];
  }
  template <class Row>
  KN_HD void finish(const Row& p) const { p[5] = I_Na; p[6] = I_K; p[7] = I_Cl; }
};
