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
#include <cstdlib>

#include "knpemi_internal.h"
#include "membrane_models.h"

namespace {

struct OdeArgs {
  int nq, q0, n_stim, flags, v_index, model_slot, NQtot;
  int ion_param[3 * KN_MAXK];
  int stim_idx[8];
  double stim_val[8];
  double t0, dt, rtol, atol;
  double* states;
  double* params;
  const uint8_t* mask;
  unsigned long long* stats;
  unsigned long long* stamps;   // diagnostic build: [workgroup][24] phase cycle sums and counts
};

constexpr int ODE_BLOCK = 64;

// LANES = M::NS: lane c of every group of NS adjacent lanes integrates component c of one membrane dof
// (lsoda_core.h); LANES = 1: one thread per dof (one-state models).
// WAVES = 2 caps the register budget at 256 per lane so that two waves share a SIMD.  It pays once there are more
// waves than SIMDs (large membranes); small sweeps run one wave per SIMD with the full register file.
template <class M, int LANES, int WAVES = 1, bool STAMPS = false>
__global__ __launch_bounds__(ODE_BLOCK, WAVES) void ode_step_kernel(KnDev D, OdeArgs a, const LsodaCoef* __restrict__ cf) {
  using Integrator = Lsoda<M::NS, M, LANES, STAMPS, ODE_BLOCK>;
  constexpr int NI = Integrator::NI;
  // factorised iteration matrix + pivots of the BDF method, one column per lane (touched by stiff dofs only)
  __shared__ double work[Integrator::WORK * ODE_BLOCK];
  // LSODA's coefficient tables (4 kB) are consulted with a per-lane order index whenever an order changes: keep the
  // workgroup's copy in LDS.  Everything else the non-stiff integrator touches lives in registers.
  __shared__ LsodaCoef scf;
  {
    const double* src = reinterpret_cast<const double*>(cf);
    double* dst = reinterpret_cast<double*>(&scf);
    for (int i = threadIdx.x; i < (int)(sizeof(LsodaCoef) / sizeof(double)); i += ODE_BLOCK) dst[i] = src[i];
    __syncthreads();
  }
  const int gt = blockIdx.x * blockDim.x + threadIdx.x;
  // the lanes past the last dof repeat the last dof and drop their results: every lane of the wave stays active,
  // so the wave-level sums below see all 64 lanes
  const bool live = gt / LANES < a.nq;
  const int q = live ? gt / LANES : a.nq - 1, comp = gt % LANES;
  const int qg = a.q0 + q;
  const StridedRow<0> p{a.params + q, (size_t)a.nq};   // this dof's parameter row in the transposed table
  double y[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) y[j] = a.states[(size_t)(comp + j) * a.nq + q];
  // 1. concentration traces (record components 4..6 hold c_0, c_1, c_eliminated) -> parameter columns.
  //    With several lanes per dof every lane writes the same values and later reads only its own stores.
  if (a.flags & KNPEMI_ODE_SET_TRACES) {
    const double* re = D.VR + (size_t)D.q2e[qg] * KN_REC + 4;
    const double* ri = D.VR + (size_t)D.q2i[qg] * KN_REC + 4;
    for (int k = 0; k < KN_MAXK; ++k) {
      p[a.ion_param[3 * k]] = re[k];
      p[a.ion_param[3 * k + 1]] = ri[k];
    }
  }
  if (a.flags & KNPEMI_ODE_SET_V) {
    const double v = D.phiM[qg];
#pragma unroll
    for (int j = 0; j < NI; ++j) y[j] = (comp + j == a.v_index) ? v : y[j];
  }
  // 2. stimulus + LSODA (the parameter row is read once by prepare(); only the currents change)
  if (a.n_stim > 0 && (!a.mask || a.mask[q]))
    for (int i = 0; i < a.n_stim; ++i) p[a.stim_idx[i]] = a.stim_val[i];
  Integrator s;
  if constexpr (STAMPS) s.st_last = __builtin_amdgcn_s_memtime();
  s.f.prepare(p);
  const int rc = s.integrate(&scf, work + threadIdx.x, y, a.t0, a.t0 + a.dt, a.rtol, a.atol, 10000, comp);
  // 3. write back: state row, phi_M_prev <- V; the lane that owns V stores the currents (the reference's
  //    RHS side effect) into the parameter row and the I_ch_k fields
  const bool owner = live && (LANES == 1 || comp == M::CURRENT_LANE);
  if (live) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      a.states[(size_t)(comp + j) * a.nq + q] = y[j];
      if (comp + j == a.v_index) D.phiM[qg] = y[j];
    }
  }
  if (owner) {
    s.f.finish(p);
    for (int k = 0; k < KN_MAXK; ++k)
      D.Ich[((size_t)a.model_slot * KN_MAXK + k) * a.NQtot + qg] = p[a.ion_param[3 * k + 2]];
  }
  // counters: summed over the wave, then added to this workgroup's own slot -- no atomics (thousands of atomic adds
  // to one word serialise at ~90 per microsecond: the tail of the sweep); knpemi_ode_stats() adds the slots up
  unsigned n_rhs = owner ? (unsigned)s.nfe : 0u, n_st = owner ? (unsigned)s.nst : 0u, n_bad = (owner && rc != 0) ? 1u : 0u;
#pragma unroll
  for (int msk = 32; msk >= 1; msk >>= 1) {
    n_rhs += __shfl_xor(n_rhs, msk);
    n_st += __shfl_xor(n_st, msk);
    n_bad += __shfl_xor(n_bad, msk);
  }
  if constexpr (STAMPS) {
    if (threadIdx.x == 0 && a.stamps)
      for (int i = 0; i < 12; ++i) {
        a.stamps[24 * (size_t)blockIdx.x + i] = s.st_acc[i];
        a.stamps[24 * (size_t)blockIdx.x + 12 + i] = s.st_cnt[i];
      }
  }
  if (threadIdx.x == 0) {
    unsigned long long* st = a.stats + 3 * (size_t)blockIdx.x;
    st[0] += n_rhs;
    st[1] += n_st;
    st[2] += n_bad;
  }
}

template <class M, int LANES>
void launch_model(knpemi_handle* h, const OdeArgs& a, const LsodaCoef* cf, int force_waves) {
  dim3 grid(((size_t)a.nq * LANES + ODE_BLOCK - 1) / ODE_BLOCK), block(ODE_BLOCK);
  // more waves than 1.5 x the chip's 1024 SIMDs: trade registers for a second resident wave per SIMD
  const bool dense = force_waves ? force_waves == 2 : (size_t)grid.x > 1536;
  if (a.stamps) hipLaunchKernelGGL((ode_step_kernel<M, LANES, 1, true>), grid, block, 0, h->cur, h->dev, a, cf);
  else if (dense) hipLaunchKernelGGL((ode_step_kernel<M, LANES, 2>), grid, block, 0, h->cur, h->dev, a, cf);
  else hipLaunchKernelGGL((ode_step_kernel<M, LANES, 1>), grid, block, 0, h->cur, h->dev, a, cf);
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
  a.model_slot = slot; a.NQtot = h->dev.NQtot;
  for (int i = 0; i < 3 * KN_MAXK; ++i) a.ion_param[i] = ion_param[i];
  for (int i = 0; i < 8; ++i) { a.stim_idx[i] = m.stim_idx[i]; a.stim_val[i] = m.stim_val[i]; }
  a.t0 = t0; a.dt = dt; a.rtol = rtol; a.atol = atol;
  a.states = m.d_states; a.params = m.d_params; a.mask = m.d_mask; a.stats = m.d_stats;
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
  switch (m.model_id) {
    case KNPEMI_MODEL_HH_SI: launch_model<ModelHHSI, 4>(h, a, cf, force_waves); break;
    case KNPEMI_MODEL_HH_MV: launch_model<ModelHHMV, 4>(h, a, cf, force_waves); break;
    default: launch_model<ModelGlial, 1>(h, a, cf, force_waves); break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    kn_set_error(std::string("ode_step_kernel: ") + hipGetErrorString(e));
    return KNPEMI_EHIP;
  }
  return KNPEMI_OK;
}
