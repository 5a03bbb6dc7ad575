// Test-only host build of the PRODUCT integrator (csrc/lsoda_core.h + membrane_models.h) so that
// `pytest -m "not gpu"` can check it against scipy's ODEPACK LSODA on the CPU.  Not shipped, not a
// fallback: the product's ODE sweep only exists as the HIP kernel in csrc/kernels_ode.hip.
#include "../../knp-emi-fenics-x_amd/csrc/membrane_models.h"
// the sequential restatement of ODEPACK (oracle/, test infrastructure), here with the product's root function so
// that the two integrators can be compared bit for bit
#define KN_SEQ_POW kn_powr
#include "../../oracle/lsoda_seq.h"

static LsodaCoef g_cf;
static bool g_init = false;

template <class M>
static int run(double* y, double* p, double t0, double t1, double rtol, double atol, int* stats) {
  Lsoda<M::NS, M> s;
  double work[Lsoda<M::NS, M>::WORK];
  s.f.prepare(p);
  int rc = s.integrate(&g_cf, work, y, t0, t1, rtol, atol, 10000);
  s.f.finish(p);   // currents: whatever the last RHS call inside LSODA computed (the reference's semantic)
  if (stats) { stats[0] = s.nfe; stats[1] = s.nst; stats[2] = s.nje; stats[3] = s.mused; stats[4] = s.nqu; }
  return rc;
}

extern "C" int lsoda_host(int model, double* y, double* p, double t0, double t1, double rtol,
                          double atol, int* stats) {
  if (!g_init) { lsoda_fill_coef(&g_cf); g_init = true; }
  if (model == 0) return run<ModelHHSI>(y, p, t0, t1, rtol, atol, stats);
  if (model == 1) return run<ModelHHMV>(y, p, t0, t1, rtol, atol, stats);
  if (model == 2) return run<ModelGlial>(y, p, t0, t1, rtol, atol, stats);
  return -100;
}

template <class M>
static int run_seq(double* y, double* p, double t0, double t1, double rtol, double atol, int* stats) {
  LsodaSeq<M::NS, M> s;
  double work[LsodaSeq<M::NS, M>::WORK];
  s.f.prepare(p);
  int rc = s.integrate(&g_cf, work, y, t0, t1, rtol, atol, 10000);
  s.f.finish(p);
  if (stats) { stats[0] = s.nfe; stats[1] = s.nst; stats[2] = s.nje; stats[3] = s.mused; stats[4] = s.nqu; }
  return rc;
}

extern "C" int lsoda_seq_host(int model, double* y, double* p, double t0, double t1, double rtol,
                              double atol, int* stats) {
  if (!g_init) { lsoda_fill_coef(&g_cf); g_init = true; }
  if (model == 0) return run_seq<ModelHHSI>(y, p, t0, t1, rtol, atol, stats);
  if (model == 1) return run_seq<ModelHHMV>(y, p, t0, t1, rtol, atol, stats);
  if (model == 2) return run_seq<ModelGlial>(y, p, t0, t1, rtol, atol, stats);
  return -100;
}

extern "C" void lsoda_host_rhs(int model, double t, const double* y, double* dy, double* p) {
  if (model == 0) { ModelHHSI m; m.prepare(p); m.rhs(t, y, dy); m.finish(p); }
  else if (model == 1) { ModelHHMV m; m.prepare(p); m.rhs(t, y, dy); m.finish(p); }
  else { ModelGlial m; m.prepare(p); m.rhs(t, y, dy); m.finish(p); }
}

extern "C" void lsoda_host_coef(double* elco, double* tesco) {
  LsodaCoef c; lsoda_fill_coef(&c);
  for (int m = 0; m < 2; ++m) for (int q = 0; q < 13; ++q) {
    for (int i = 0; i < 14; ++i) elco[(m * 13 + q) * 14 + i] = c.elco[m][q][i];
    for (int i = 0; i < 4; ++i) tesco[(m * 13 + q) * 4 + i] = c.tesco[m][q][i];
  }
}
