// The membrane ODE sweep kernel (see kernels_ode.hip for what it fuses).  This header is compiled twice: by hipcc into
// libknpemi_hip.so for the membrane models that ship with the library (membrane_models.h), and at run time by hipRTC
// for a plug-in that brings its own right-hand side as HIP source (knpemi_ode_bind_source): it therefore includes
// nothing but lsoda_core.h and uses plain types only.
#pragma once

#include "lsoda_core.h"

#define KN_ODE_MAXK 4     // == KNPEMI_MAX_IONS
#define KN_ODE_REC 8      // == KN_REC: doubles per vertex record, ion k in slot KN_ODE_CSLOT(k)
#define KN_ODE_CSLOT(k) ((k) < 3 ? 4 + (k) : 3)

// the slice of the device problem the sweep touches
struct OdeDev {
  const double* VR;      // [Ntot][8] vertex records
  const int* q2e;        // [NQtot] ECS / cell vertex of every membrane dof
  const int* q2i;
  double* phiM;          // [NQtot]
  double* Ich;           // [model slot][ion][NQtot]
};

struct OdeArgs {
  int nq, q0, n_stim, flags, v_index, model_slot, NQtot, n_ions;
  int dpw;               // membrane dofs per wavefront (<= 64 / LANES; 0 = all of them), see ode_step_body
  int ion_param[3 * KN_ODE_MAXK];
  int stim_idx[8];
  double stim_val[8];
  double t0, dt, rtol, atol;
  double* states;
  double* params;
  const unsigned char* mask;
  unsigned long long* stats;
  unsigned long long* stamps;   // diagnostic build: [workgroup][24] phase cycle sums and counts
};

#define KN_ODE_SET_V 1        // == KNPEMI_ODE_SET_V
#define KN_ODE_SET_TRACES 2   // == KNPEMI_ODE_SET_TRACES

constexpr int ODE_BLOCK = 64;

// a model may fix per-lane constants of its component-wise right-hand side once per sweep (membrane_models.h)
template <class M>
KN_HD auto kn_model_set_lane(M& m, int c, int) -> decltype(m.set_lane(c), void()) { m.set_lane(c); }
template <class M>
KN_HD void kn_model_set_lane(M&, int, long) {}

template <int S>
struct StridedRow {   // p[j] of dof q in a [column][dof] table: base + j * S
  double* b;
  size_t stride;
  KN_HD double& operator[](int j) const { return b[(size_t)j * stride]; }
};

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
// LANES = M::NS: lane c of every group of NS adjacent lanes integrates component c of one membrane dof
// (lsoda_core.h); LANES = 1: one thread per dof (one-state models).
// WAVES = 2 caps the register budget at 256 per lane so that two waves share a SIMD.  It pays once there are more
// waves than SIMDs (large membranes); small sweeps run one wave per SIMD with the full register file.
template <class M, int LANES, int WAVES, bool STAMPS>
__device__ __forceinline__ void ode_step_body(const OdeDev& D, const OdeArgs& a, const LsodaCoef* __restrict__ cf) {
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
  // Dofs per wavefront.  A wave runs the UNION of the trips of its dofs' phase machines, and an instruction costs the same
  // whatever the exec mask: with 64 / LANES dofs per wave ~10 % of the stream is other dofs' trips.  While the sweep has fewer
  // waves than the chip has SIMDs (config 2: 185 on 1 024) the idle SIMDs buy that back: `dpw` dofs per wave, the other
  // lanes MIRROR them (lane l integrates what lane l mod (dpw LANES) integrates and drops the result: identical control
  // flow, no divergence added, every lane active for the wave-level sums).
  constexpr int FULL = ODE_BLOCK / LANES;
  const int dpw = (a.dpw > 0 && a.dpw < FULL) ? a.dpw : FULL;
  const int lane_in = threadIdx.x % (dpw * LANES);
  const bool primary = threadIdx.x < dpw * LANES;
  const int qw = blockIdx.x * dpw + lane_in / LANES;
  // the lanes past the last dof repeat the last dof and drop their results: every lane of the wave stays active,
  // so the wave-level sums below see all 64 lanes
  const bool live = primary && qw < a.nq;
  const int q = qw < a.nq ? qw : a.nq - 1, comp = threadIdx.x % LANES;
  const int qg = a.q0 + q;
  const StridedRow<0> p{a.params + q, (size_t)a.nq};   // this dof's parameter row in the transposed table
  double y[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) y[j] = a.states[(size_t)(comp + j) * a.nq + q];
  // 1. concentration traces (record components 4..6 hold c_0, c_1, c_eliminated) -> parameter columns.
  //    With several lanes per dof every lane writes the same values and later reads only its own stores.
  if (a.flags & KN_ODE_SET_TRACES) {
    const double* re = D.VR + (size_t)D.q2e[qg] * KN_ODE_REC;
    const double* ri = D.VR + (size_t)D.q2i[qg] * KN_ODE_REC;
    for (int k = 0; k < a.n_ions; ++k) {
      p[a.ion_param[3 * k]] = re[KN_ODE_CSLOT(k)];
      p[a.ion_param[3 * k + 1]] = ri[KN_ODE_CSLOT(k)];
    }
  }
  if (a.flags & KN_ODE_SET_V) {
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
  if constexpr (LANES > 1) kn_model_set_lane(s.f, comp, 0);
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
    for (int k = 0; k < a.n_ions; ++k)
      D.Ich[((size_t)a.model_slot * KN_ODE_MAXK + k) * a.NQtot + qg] = p[a.ion_param[3 * k + 2]];
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


template <class M, int LANES, int WAVES = 1, bool STAMPS = false>
__global__ __launch_bounds__(ODE_BLOCK, WAVES) void ode_step_kernel(OdeDev D, OdeArgs a, const LsodaCoef* __restrict__ cf) {
  ode_step_body<M, LANES, WAVES, STAMPS>(D, a, cf);
}
#endif
