// Membrane ODE sweep on gfx950: one lane per state component, LSODA entirely in registers.
//
// One launch fuses what the reference does in three Python stages per membrane model and step
// (examples/idealized_geometries/run_3D.py:80-111):
//   1. update_ode_variables (src/knpemi/utils.py:210-235): nodal traces of the K concentrations on
//      both sides of the membrane -> parameter columns "<ion>_e"/"<ion>_i"; V <- phi_M_prev (k > 0);
//   2. MembraneModel.step_lsoda (src/knpemi/odeSolver.py:92-127): optional stimulus write, LSODA over
//      [t, t + dt] with rtol 1e-8 / atol 1e-10, state row <- solution at t + dt;
//   3. copy-back (run_3D.py:104-109): phi_M_prev <- V, I_ch_k <- parameter columns "I_ch_<ion>".
// The currents handed to the PDEs are, as in the reference, whatever the last RHS call made by LSODA
// stored in the parameter row; lsoda_core.h reproduces ODEPACK's call
// sequence, so this is the same evaluation point.
//
// Tables are stored transposed on the device ([column][dof]) so that neighbouring threads read
// neighbouring addresses; the ODE work itself is latency/compute bound (fp64 exp/log, divergent
// step control), not HBM bound.
#include <algorithm>
#include <cstdlib>

#include "knpemi_internal.h"
#include "membrane_models.h"
#include "ode_kernel.h"

namespace {

OdeDev ode_dev(const knpemi_handle* h) {
  const KnDev& D = h->dev;
  return OdeDev{D.VR, D.q2e, D.q2i, D.phiM, D.Ich};
}

// Dofs per wavefront (ode_step_body): as few as keep the sweep at one wave per SIMD (1 024 SIMDs; the stats slots of the
// handle's sweeps are sized for that), all 64 / LANES of them otherwise.  KNPEMI_ODE_DPW forces a value (0: full waves).
int dofs_per_wave(int nq, int lanes, int max_blocks) {
  static const int forced = [] { const char* e = getenv("KNPEMI_ODE_DPW"); return e ? atoi(e) : -1; }();
  const int full = ODE_BLOCK / lanes;
  if (forced >= 0) return forced > 0 && forced < full && (nq + forced - 1) / forced <= max_blocks ? forced : 0;
  // Measured at config 2 (2 952 dofs; round 4, bench.py): 16 dofs per wave 77.5-81 us per sweep, 8: 83.1, 4: 83.1 (spike
  // window 106.7 / 110.8 / 104.6): fewer dofs per wave do NOT shorten the wave's stream measurably -- the trips of the 16
  // phase machines of a wave overlap almost completely -- and four times the waves cost more than they save.  Full waves
  // unless forced.
  (void)max_blocks;
  return 0;
}

template <class M, int LANES>
void launch_model(hipStream_t st, const OdeDev& dv, const OdeArgs& a, const LsodaCoef* cf, int force_waves) {
  const int per_wave = a.dpw > 0 ? a.dpw : ODE_BLOCK / LANES;
  dim3 grid(((size_t)a.nq + per_wave - 1) / per_wave), block(ODE_BLOCK);
  // more waves than 1.5 x the chip's 1024 SIMDs: trade registers for a second resident wave per SIMD
  const bool dense = force_waves ? force_waves == 2 : (size_t)grid.x > 1536;
  if (a.stamps) hipLaunchKernelGGL((ode_step_kernel<M, LANES, 1, true>), grid, block, 0, st, dv, a, cf);
  else if (dense) hipLaunchKernelGGL((ode_step_kernel<M, LANES, 2>), grid, block, 0, st, dv, a, cf);
  else hipLaunchKernelGGL((ode_step_kernel<M, LANES, 1>), grid, block, 0, st, dv, a, cf);
}

int launch_builtin(hipStream_t st, int model_id, const OdeDev& dv, const OdeArgs& a, const LsodaCoef* cf, int force_waves) {
  switch (model_id) {
    case KNPEMI_MODEL_HH_SI: launch_model<ModelHHSI, 4>(st, dv, a, cf, force_waves); break;
    case KNPEMI_MODEL_HH_MV: launch_model<ModelHHMV, 4>(st, dv, a, cf, force_waves); break;
    default: launch_model<ModelGlial, 1>(st, dv, a, cf, force_waves); break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    kn_set_error(std::string("ode_step_kernel: ") + hipGetErrorString(e));
    return KNPEMI_EHIP;
  }
  return KNPEMI_OK;
}

}  // namespace

static int ensure_coef(knpemi_handle* h, const LsodaCoef** out) {
  if (!h->d_lsoda_coef) {
    LsodaCoef c;
    lsoda_fill_coef(&c);
    void* d = nullptr;
    KN_HIP(hipMalloc(&d, sizeof(LsodaCoef)));
    h->allocs.push_back(d);
    KN_HIP(hipMemcpy(d, &c, sizeof(LsodaCoef), hipMemcpyHostToDevice));
    h->d_lsoda_coef = d;
  }
  *out = static_cast<const LsodaCoef*>(h->d_lsoda_coef);
  return KNPEMI_OK;
}

int kn_launch_ode_step(knpemi_handle* h, int slot, double t0, double dt, double rtol, double atol,
                       int flags, const int32_t* ion_param, int v_index) {
  KnOdeModel& m = h->ode[slot];
  if (m.nq == 0) return KNPEMI_OK;
  const LsodaCoef* cf = nullptr;
  int rc = ensure_coef(h, &cf);
  if (rc) return rc;
  OdeArgs a;
  a.nq = m.nq; a.q0 = h->qoff[m.sub]; a.n_stim = m.n_stim; a.flags = flags; a.v_index = v_index;
  a.model_slot = slot; a.NQtot = h->dev.NQtot; a.n_ions = h->K;
  for (int i = 0; i < 3 * KN_MAXK; ++i) a.ion_param[i] = i < 3 * h->K ? ion_param[i] : 0;
  for (int i = 0; i < 8; ++i) { a.stim_idx[i] = m.stim_idx[i]; a.stim_val[i] = m.stim_val[i]; }
  a.t0 = t0; a.dt = dt; a.rtol = rtol; a.atol = atol;
  a.states = m.d_states; a.params = m.d_params; a.mask = m.d_mask; a.stats = m.d_stats;
  a.dpw = dofs_per_wave(m.nq, m.rtc_function ? m.rtc_lanes : (m.n_states == 4 ? 4 : 1), m.n_stat_blocks - 1);
  if (m.rtc_function) a.dpw = 0;      // (the run-time compiled sweep is launched with full waves, kernels_rtc.hip)
  // KNPEMI_ODE_STAMPS=1: diagnostic build of the sweep with s_memtime stamps between the phases (tools/ode_stamps.py)
  static const bool want_stamps = getenv("KNPEMI_ODE_STAMPS") != nullptr;
  a.stamps = nullptr;
  if (want_stamps) {
    if (!m.d_stamps) {
      void* d = nullptr;
      KN_HIP(hipMalloc(&d, 24 * sizeof(unsigned long long) * (size_t)m.n_stat_blocks));
      h->allocs.push_back(d);
      m.d_stamps = static_cast<unsigned long long*>(d);
    }
    a.stamps = m.d_stamps;
  }
  // counters accumulate over launches; knpemi_ode_stats() reads and resets them
  // 64-thread workgroups: the sweep has only n_q (10^3..10^5) threads, so spread the waves over as
  // many CUs as possible instead of stacking four of them on one.
  static const int force_waves = [] { const char* e = getenv("KNPEMI_ODE_WAVES"); return e ? atoi(e) : 0; }();
  KnProfScope prof(h, KNPEMI_K_ODE);
  if (m.rtc_function) {   // plug-in compiled at bind time (kernels_rtc.hip)
    const OdeDev dv = ode_dev(h);
    a.stamps = nullptr;
    return kn_rtc_launch(h, m, &dv, sizeof(dv), &a, sizeof(a), cf);
  }
  return launch_builtin(h->cur, m.model_id, ode_dev(h), a, cf, force_waves);
}

// The same sweep over a caller-described table (the DG variant, kernels_dg.hip: membrane nodes of the broken space).
int kn_launch_ode_raw(hipStream_t st, int model_id, const OdeDev& dv, const OdeArgs& a, const void* coef) {
  return launch_builtin(st, model_id, dv, a, static_cast<const LsodaCoef*>(coef), 0);
}

int kn_lsoda_coef_upload(void** out) {
  LsodaCoef c;
  lsoda_fill_coef(&c);
  void* d = nullptr;
  KN_HIP(hipMalloc(&d, sizeof(LsodaCoef)));
  KN_HIP(hipMemcpy(d, &c, sizeof(LsodaCoef), hipMemcpyHostToDevice));
  *out = d;
  return KNPEMI_OK;
}

// ---- diagnostics: the sweep's math helpers over an array (knpemi_debug_math) ------------------------------------
__global__ void debug_math_kernel(int op, int n, const double* __restrict__ a, const double* __restrict__ b,
                                  double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = op == 0 ? kn_div(a[i], b[i]) : (op == 1 ? kn_exp(a[i]) : (op == 2 ? kn_powr(a[i], b[i]) : kn_log(a[i])));
}

extern "C" int knpemi_debug_math(int op, int n, const double* a, const double* b, double* out) {
  if (op < 0 || op > 3 || n < 0 || !a || !out || ((op == 0 || op == 2) && !b)) {
    kn_set_error("knpemi_debug_math: bad arguments");
    return KNPEMI_EINVAL;
  }
  if (n == 0) return KNPEMI_OK;
  double* d = nullptr;
  KN_HIP(hipMalloc(reinterpret_cast<void**>(&d), 3 * sizeof(double) * (size_t)n));
  int rc = KNPEMI_OK;
  auto check = [&](hipError_t e) { if (e != hipSuccess && rc == KNPEMI_OK) { kn_set_error(hipGetErrorString(e)); rc = KNPEMI_EHIP; } };
  check(hipMemcpy(d, a, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  check(hipMemcpy(d + n, (op == 1 || op == 3) ? a : b, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  if (rc == KNPEMI_OK) {
    hipLaunchKernelGGL(debug_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, d, d + n, d + 2 * (size_t)n);
    check(hipGetLastError());
    check(hipMemcpy(out, d + 2 * (size_t)n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  }
  (void)hipFree(d);
  return rc;
}
