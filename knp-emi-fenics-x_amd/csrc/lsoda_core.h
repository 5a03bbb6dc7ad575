// LSODA (Livermore Solver for ODEs with Automatic method switching) for tiny systems on gfx950,
// written as a flat phase machine so that the lanes of a wavefront never serialise each other.
//
// The reference calls numbalsoda's C++ LSODA once per membrane dof and step
// (src/knpemi/odeSolver.py:116-120: `lsoda(addr, state, [t, t+dt], data=params, rtol=1e-8,
// atol=1e-10)`).  numbalsoda is un-vendored and un-pinned (pyproject.toml:13-18); what is
// restated here is the published ODEPACK algorithm it implements (Hindmarsh 1983; Petzold 1983):
// variable-order, variable-step Adams (orders 1..12, functional iteration) and BDF (orders 1..5,
// chord iteration with a finite-difference Jacobian) in Nordsieck form with automatic stiffness
// detection, istate = 1 / itask = 1 semantics (fresh start at every call, overshoot tout and
// interpolate back).
//
// Shape of the code.  ODEPACK's DLSODA -> DSTODA -> corrector nest is four loops deep (time loop, retry after
// a failed error test, retry after a corrector failure, corrector iterations).  On a GPU every lane of a
// wavefront would wait for the slowest lane at each level: a wave pays the SUM over steps of the MAXIMUM
// number of retries / iterations, and the compiler keeps a copy of the right-hand side per call site.  Here one
// trip of a single loop is one right-hand-side evaluation for every lane, whatever that lane is doing:
//
//     TOP (checks before a step, DSTODA prologue) -> PRED (predict, start the corrector) -> [RHS] ->
//     CORR (one corrector iteration, convergence test) -> ERR (error test, order / step / method selection)
//
// and each lane carries its phase.  A wave therefore runs max-over-lanes(RHS evaluations) trips, there is ONE
// right-hand-side call site in the loop, and all integrator state -- the Nordsieck array included -- is held in
// registers: loops over the rows of the Nordsieck array are unrolled over the 13 possible rows and cut at a
// wave-uniform bound `lhi` >= every lane's order + 1, rows above a lane's own order are kept at zero so that
// these loops need no per-lane predicate (x + 0 = x exactly).  The arithmetic of every formula is ODEPACK's;
// tests/ check the host build of this file against scipy's ODEPACK LSODA (same step, evaluation and Jacobian
// counts) and, bit for bit, against the plain sequential restatement in oracle/lsoda_seq.h.
//
// Compiles for the device (hipcc) and for the host (g++, used only by tests/ to check this very
// code on the CPU -- it is not a CPU fallback of the product).
#pragma once

#if !defined(__HIPCC_RTC__)   // (hipRTC provides the device math functions and size_t itself)
#include <math.h>
#include <stddef.h>
#endif

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define KN_HD __host__ __device__ __forceinline__
#define KN_HDN __host__ __device__ __forceinline__
#else
#define KN_HD inline
#define KN_HDN
#endif

#define KN_ETA 2.2204460492503131e-16

// true when the predicate holds on any active lane of the wavefront (a scalar branch condition on the device)
#if defined(__HIP_DEVICE_COMPILE__)
#define KN_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
// a && b on any lane, as the AND of the two lane masks (the ballot of a conjunction is materialised as a 0 / 1 value per
// lane and compared again)
#define KN_ANY2(a, b) ((__builtin_amdgcn_ballot_w64(a) & __builtin_amdgcn_ballot_w64(b)) != 0ull)
#else
#define KN_ANY(c) (c)
#define KN_ANY2(a, b) ((a) && (b))
#endif

// Method coefficients (ODEPACK CFODE), 1-based like the original: elco[meth-1][nq][i],
// tesco[meth-1][nq][k]; cm1/cm2 = tesco[.][2] * elco[.][nq+1] used by the method switch; sm1 = the
// stability-region bounds of the Adams methods (DSTODA's data statement).
struct LsodaCoef {
  double elco[2][13][14];
  double tesco[2][13][4];
  double cm1[13];
  double cm2[6];
  double sm1[13];
  double rtesco[2][13][4];   // 1 / tesco and 1 / i (i = 1..15), correctly rounded on the host: the device multiplies
  double rint_[16];          // where DSTODA divides by these constants (the reciprocals themselves are bit-identical)
};

#if !defined(__HIPCC_RTC__)
// Host-side construction of the coefficient tables (exact restatement of CFODE).
inline void lsoda_fill_coef(LsodaCoef* c) {
  for (int m = 0; m < 2; ++m)
    for (int q = 0; q < 13; ++q) {
      for (int i = 0; i < 14; ++i) c->elco[m][q][i] = 0.0;
      for (int i = 0; i < 4; ++i) c->tesco[m][q][i] = 0.0;
    }
  double pc[13];
  {  // Adams
    double(*elco)[14] = c->elco[0];
    double(*tesco)[4] = c->tesco[0];
    elco[1][1] = 1.0; elco[1][2] = 1.0;
    tesco[1][1] = 0.0; tesco[1][2] = 2.0; tesco[2][1] = 1.0; tesco[12][3] = 0.0;
    pc[1] = 1.0;
    double rqfac = 1.0;
    for (int nq = 2; nq <= 12; ++nq) {
      const double rq1fac = rqfac;
      rqfac = rqfac / (double)nq;
      const int nqm1 = nq - 1, nqp1 = nq + 1;
      const double fnqm1 = (double)nqm1;
      pc[nq] = 0.0;
      for (int i = nq; i >= 2; --i) pc[i] = pc[i - 1] + fnqm1 * pc[i];
      pc[1] = fnqm1 * pc[1];
      double pint = pc[1], xpin = pc[1] / 2.0, tsign = 1.0;
      for (int i = 2; i <= nq; ++i) {
        tsign = -tsign;
        pint += tsign * pc[i] / (double)i;
        xpin += tsign * pc[i] / (double)(i + 1);
      }
      elco[nq][1] = pint * rq1fac;
      elco[nq][2] = 1.0;
      for (int i = 2; i <= nq; ++i) elco[nq][i + 1] = rq1fac * pc[i] / (double)i;
      const double agamq = rqfac * xpin, ragq = 1.0 / agamq;
      tesco[nq][2] = ragq;
      if (nq < 12) tesco[nqp1][1] = ragq * rqfac / (double)nqp1;
      tesco[nqm1][3] = ragq;
    }
  }
  {  // BDF
    double(*elco)[14] = c->elco[1];
    double(*tesco)[4] = c->tesco[1];
    pc[1] = 1.0;
    double rq1fac = 1.0;
    for (int nq = 1; nq <= 5; ++nq) {
      const double fnq = (double)nq;
      const int nqp1 = nq + 1;
      pc[nqp1] = 0.0;
      for (int i = nq + 1; i >= 2; --i) pc[i] = pc[i - 1] + fnq * pc[i];
      pc[1] *= fnq;
      for (int i = 1; i <= nqp1; ++i) elco[nq][i] = pc[i] / pc[2];
      elco[nq][2] = 1.0;
      tesco[nq][1] = rq1fac;
      tesco[nq][2] = (double)nqp1 / elco[nq][1];
      tesco[nq][3] = (double)(nq + 2) / elco[nq][1];
      rq1fac /= fnq;
    }
  }
  c->cm1[0] = 0.0;
  c->cm2[0] = 0.0;
  for (int i = 1; i <= 12; ++i) c->cm1[i] = c->tesco[0][i][2] * c->elco[0][i][i + 1];
  for (int i = 1; i <= 5; ++i) c->cm2[i] = c->tesco[1][i][2] * c->elco[1][i][i + 1];
  static const double sm1[13] = {0.0, 0.5, 0.575, 0.55, 0.45, 0.35, 0.25, 0.2, 0.15, 0.1, 0.075, 0.05, 0.025};
  for (int i = 0; i < 13; ++i) c->sm1[i] = sm1[i];
  for (int m = 0; m < 2; ++m)
    for (int q = 0; q < 13; ++q)
      for (int k = 0; k < 4; ++k) c->rtesco[m][q][k] = c->tesco[m][q][k] != 0.0 ? 1.0 / c->tesco[m][q][k] : 0.0;
  c->rint_[0] = 0.0;
  for (int i = 1; i < 16; ++i) c->rint_[i] = 1.0 / (double)i;
}
#endif

// LANES: number of GPU lanes that share one ODE system.  LANES = 1: one thread integrates all N
// components (host build, and N = 1 models).  LANES = N (device only): lane c of a group of N adjacent
// lanes owns component c -- every vector operation of the algorithm becomes one scalar operation per
// lane, norms become a max over the group (exact, so all lanes of a group take identical decisions),
// the right-hand side is evaluated component-wise by `F::rhs_lane` after an all-gather of the state,
// and the N x N iteration matrix is gathered and factorised redundantly by every lane.
#if defined(__HIP_DEVICE_COMPILE__)
// Exchange inside an aligned group of four lanes as a DPP quad permute on the two halves of the double: a
// register-to-register move (two v_mov_b32_dpp), where __shfl / __shfl_xor go through the LDS crossbar
// (ds_bpermute, ~100 cycles).  CTRL = quad_perm[0] | [1] << 2 | [2] << 4 | [3] << 6.  All lanes of a group are
// active together (they take identical decisions), so no lane ever reads a disabled neighbour.
template <int CTRL>
__device__ __forceinline__ double kn_dpp_quad(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// max(a, b) as the one instruction it is.  fmax() on a value that arrived through a DPP move of its two halves is
// preceded by a canonicalising v_max_f64 x, x, x (the compiler cannot know that the bits are a quiet number): one more
// instruction per exchange, and every norm of the integrator is two exchanges deep.
__device__ __forceinline__ double kn_max_raw(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#endif

template <int L>
KN_HD double kn_group_max(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (L == 2 || L == 4) {
    v = kn_max_raw(v, kn_dpp_quad<0xB1>(v));                      // lanes 1 0 3 2
    if constexpr (L == 4) v = kn_max_raw(v, kn_dpp_quad<0x4E>(v));   // lanes 2 3 0 1
  } else {
#pragma unroll
    for (int m = 1; m < L; m <<= 1) v = fmax(v, __shfl_xor(v, m));
  }
#endif
  return v;
}

template <int L>
KN_HD double kn_group_get(double v, int k) {   // value held by lane k of this lane's group
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (L == 4) {
    switch (k) {   // k is a compile-time constant at every call site (unrolled loops)
      case 0: return kn_dpp_quad<0x00>(v);
      case 1: return kn_dpp_quad<0x55>(v);
      case 2: return kn_dpp_quad<0xAA>(v);
      default: return kn_dpp_quad<0xFF>(v);
    }
  } else if constexpr (L > 1) {
    return __shfl(v, (int)((__lane_id() & ~(unsigned)(L - 1)) | (unsigned)k));
  }
#else
  (void)k;
#endif
  return v;
}

// a / b.  On the device: a times the hardware reciprocal estimate r0, corrected by the product form of the Newton
// series, a r0 (1 + e)(1 + e^2) with e = 1 - b r0 (exact quotient: a r0 (1 + e + e^2 + e^3 + e^4 + ...), so the
// relative error is e^4 plus rounding, a few 1e-16).  The integrator is ONE dependent chain -- a sweep is as long as the
// longest path through its fp64 instructions (18 ns per dependent instruction on gfx950, 3.5 ns per independent one,
// tools/probes/exec_skip.hip) -- and divides four to eight times per trip: the two legs {e, a r0} and {e^2, q (1 + e)}
// run side by side, so a quotient is 4 instructions deep where two Newton steps on r0 followed by the product are 6
// and the IEEE sequence 11.  The host build (the bit-for-bit comparison with the sequential restatement and the
// nst / nfe comparison with ODEPACK) keeps the exact quotient.
KN_HD double kn_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = __builtin_amdgcn_rcp(b);
  const double e = fma(-b, r, 1.0);
  double q = a * r;
  const double e2 = e * e;
  q = fma(q, e, q);
  return fma(q, e2, q);
#else
  return a / b;
#endif
}

// exp(x).  On the device: x = k ln 2 + r with |r| <= ln 2 / 2 (two-constant reduction), exp(r) from the degree-13
// Taylor polynomial (remainder r^14 / 14! < 5e-18) in Estrin's arrangement -- pairs, quads, octets side by side: 4
// dependent instructions after r where Horner's rule (the library's exp) is 12 -- then scaled by 2^k; rounding error
// below 2 ulp (tests/test_gpu_parity.py::test_device_math_helpers).  The host build calls the C library.
KN_HD double kn_exp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double k = __builtin_rint(x * 1.4426950408889634);
  double r = fma(k, -0x1.62e42fefa39efp-1, x);
  r = fma(k, -0x1.abc9e3b39803fp-56, r);
  const double r2 = r * r;
  const double p01 = 1.0 + r;
  const double p23 = fma(r, 1.0 / 6.0, 0.5);
  const double p45 = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  const double p67 = fma(r, 1.0 / 5040.0, 1.0 / 720.0);
  const double p89 = fma(r, 1.0 / 362880.0, 1.0 / 40320.0);
  const double pab = fma(r, 1.0 / 39916800.0, 1.0 / 3628800.0);
  const double pcd = fma(r, 1.0 / 6227020800.0, 1.0 / 479001600.0);
  const double r4 = r2 * r2;
  const double q0 = fma(p23, r2, p01);
  const double q1 = fma(p67, r2, p45);
  const double q2 = fma(pab, r2, p89);
  const double r8 = r4 * r4;
  const double o0 = fma(q1, r4, q0);
  const double o1 = fma(pcd, r4, q2);
  double v = __builtin_amdgcn_ldexp(fma(o1, r8, o0), (int)k);
  v = x > 709.782712893384 ? __builtin_inf() : v;     // also +inf (the reduction of an infinity is a NaN)
  v = x < -745.1332191019412 ? 0.0 : v;               // also -inf
  return v;
#else
  return exp(x);
#endif
}

// log(x), x >= 0.  On the device: x = 2^k m with m in [sqrt(1/2), sqrt(2)), s = f / (2 + f), f = m - 1, and
// log m = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-7 polynomial of the classical argument reduction (fdlibm's
// published coefficients) split into its even and odd halves: 19 dependent instructions where the library's log
// (double-double arithmetic, needed for pow, not here) is about 50; a few ulp.  Used by the step-ratio powers only.
KN_HD double kn_log(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double m = __builtin_amdgcn_frexp_mant(x);   // [1/2, 1)
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.7071067811865476;
  m = lo ? m + m : m;
  k = lo ? k - 1 : k;
  const double f = m - 1.0;
  const double s = kn_div(f, 2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                   2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  double v = dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
  v = x == 0.0 ? -__builtin_inf() : v;
  v = x == __builtin_inf() ? x : v;
  v = x < 0.0 ? __builtin_nan("") : v;
  return v;
#else
  return log(x);
#endif
}

// x^e for x >= 0, e > 0 (the step-ratio formulas of DSTODA): exp(e log x), a third of the dependent instructions of
// the general pow() on gfx950 and accurate to a few ulp, far inside what these heuristics resolve.
KN_HD double kn_powr(double x, double e) { return kn_exp(kn_log(x) * e); }

// N = number of states (1..8); F is a model functor (membrane_models.h): `rhs(t, y, dy)` evaluates all
// components, `rhs_lane(c, t, y)` component c.
// STAMPS: diagnostic build only (KNPEMI_ODE_STAMPS=1, never the timed kernel) -- s_memtime stamps between the phases of
// a trip, summed per phase.
// STRIDE: distance (in doubles) between consecutive entries of the work storage that holds the factorised iteration
// matrix of the BDF method and its pivots (only the stiff method touches it): 1 on the host (a private array), the
// workgroup size on the device, where `work` points into LDS and lane t owns column t.
template <int N, class F, int LANES = 1, bool STAMPS = false, int STRIDE = 1>
struct Lsoda {
  static constexpr int MXORDN = 12, MXORDS = 5, MAXCOR = 3, MSBP = 20, MXNCF = 10;
  static constexpr int NI = N / LANES;               // components held by one lane
  static constexpr int WORK = N * N + N;             // doubles of strided work storage (iteration matrix, pivots)
  static constexpr int ROWS = 13;                    // Nordsieck rows 1..13
  static_assert(LANES == 1 || LANES == N, "one lane per system or one lane per component");
  enum Phase { PH_TOP = 0, PH_PRED = 1, PH_CORR = 2, PH_ERR = 3, PH_RESTART = 4, PH_DONE = 5 };

  const LsodaCoef* cf;
  F f;        // model functor: caches the parameter row, keeps the side-effect currents
  double rtol, atol;
  double yh[ROWS + 1][NI];   // Nordsieck array, rows 1..13; rows above l are kept at +0
  double ysave[NI];          // ODEPACK's YH(lmax) slot: the correction saved for the order-increase test
  // The method coefficients el(1..l) of the current (meth, nq) are the row elco[meth - 1][nq][.] of the table (entries
  // above l = nq + 1 are zero there): read from the table where they are used (the workgroup's copy is in LDS on the
  // device) instead of being held in 28 registers per lane -- the integrator state did not fit the 256 architected
  // registers and lived partly in accumulator registers, every use a copy.  el(1) is used in every corrector pass.
  double el1;
  double tq2;                // tesco(nq, 2) of the current (meth, nq); tesco(nq, 1) and (nq, 3) are read where the order
  double rtq2;               // is selected; rtq2: its reciprocal (device: from the table; see over_tq)
  double ewt[NI], savf[NI], acor[NI], y[NI];
  double* work;
  KN_HD double& WM(int i, int j) const { return work[(i * N + j) * STRIDE]; }   // iteration matrix / its LU factors
  KN_HD double& PIV(int k) const { return work[(N * N + k) * STRIDE]; }         // pivot rows (stored as doubles)
  int comp = 0;                                         // LANES = N: the component this lane owns
  double h, hu, tn, hold, rcr, crate, conit, el0, rmax, pdest, pdlast, pdnorm, ratio, tsw;
  double told, delp, pdh, rh, del, pnorm, rate;         // DSTODA / corrector locals that live across phases
  double anorm;   // vmnorm(acor) of the corrector iteration just made (m > 0): the error test reads it
  int nq, l, meth, mused, miter, ialth, ipup, jcur, jstart, kflag, icount, irflag, nslp, nst, nfe,
      nje, lmax, maxord, nqu, ierpj;
  int m, ncf, ph, ret;
  int lhi;    // wave-uniform bound: l <= lhi on every running lane when a trip starts
  int lseen;  // the order + 1 this lane had when lhi was last brought up to date
  unsigned long long st_last = 0, st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned st_cnt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

  KN_HD void stamp(int i, bool ran) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (STAMPS) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      __builtin_amdgcn_sched_barrier(0);
      st_acc[i] += t - st_last;
      st_last = t;
      st_cnt[i] += ran ? 1u : 0u;
    }
#else
    (void)i; (void)ran;
#endif
  }

  KN_HD double sm1(int i) const { return cf->sm1[i]; }

  // x / tesco(nq, k) and x / i: on the device products with the tabulated reciprocals (a quotient costs 4 dependent
  // instructions, lsoda_core.h kn_div; DSTODA divides by these constants in every step), true quotients on the host
  KN_HD double over_tq(double x, double tq, double rtq) const {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)tq;
    return x * rtq;
#else
    (void)rtq;
    return x / tq;
#endif
  }
  KN_HD double over_int(double x, int i) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return x * cf->rint_[i];
#else
    return x / (double)i;
#endif
  }

  KN_HD double vmnorm(const double* v) const {
#if defined(__HIP_DEVICE_COMPILE__)
    // one component per lane: the group maximum of the weighted magnitudes is the norm (the maximum with 0 that
    // opens the host loop only matters for a NaN, which here stays a NaN unless another component is a number --
    // the corrector then fails and the sweep reports the dof, instead of accepting a step on a vanished norm)
    if constexpr (NI == 1 && LANES > 1) return kn_group_max<LANES>(fabs(v[0]) * ewt[0]);
#endif
    double vm = 0.0;
    _Pragma("unroll") for (int i = 0; i < NI; ++i) vm = fmax(vm, fabs(v[i]) * ewt[i]);
    return kn_group_max<LANES>(vm);
  }

  // ---- loops over the rows of the Nordsieck array ---------------------------------------------------------
  // fn(j) for j = FIRST .. cap, cap = the class (4, 6, 8 or 13) of the wave-uniform bound lb >= l.  Every body is
  // harmless on the rows between a lane's own l and the cap (zero rows, masked selects), so no row costs a branch.
  // The classes are nested, not alternatives: rows up to 4 unconditionally, then 5-6, 7-8, 9-13 behind wave-uniform
  // tests, so the statement that updates a row exists once and updates its registers in place -- as four alternative
  // unrolled copies every row had four definitions and the merge points were register-to-register copies of the whole
  // array (a fifth of the instructions of a trip).
  template <int FIRST, int LAST, class Fn>
  KN_HD void rows_range(Fn&& fn) {
    _Pragma("unroll") for (int j = FIRST; j <= LAST; ++j) fn(j);
  }
  template <int FIRST, class Fn>
  KN_HD void for_rows(int lb, Fn&& fn) {
    rows_range<FIRST, 4>(fn);
    if (lb > 4) {
      rows_range<5, 6>(fn);
      if (lb > 6) {
        rows_range<7, 8>(fn);
        if (lb > 8) rows_range<9, ROWS>(fn);
      }
    }
  }
  template <int FIRST, int LAST, class Fn>
  KN_HD void rows_range_down(Fn&& fn) {
    _Pragma("unroll") for (int j = FIRST; j >= LAST; --j) fn(j);
  }
  template <class Fn>
  KN_HD void for_rows_down(int lb, Fn&& fn) {
    if (lb > 8) rows_range_down<ROWS, 9>(fn);
    if (lb > 6) rows_range_down<8, 7>(fn);
    if (lb > 4) rows_range_down<6, 5>(fn);
    rows_range_down<4, 1>(fn);
  }

  // row j of the Nordsieck array, j <= jb (jb wave-uniform)
  KN_HD void get_row(int j, int jb, double* out) {
    _Pragma("unroll") for (int i = 0; i < NI; ++i) out[i] = 0.0;
    for_rows<1>(jb, [&](int jj) {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) out[i] = (jj == j) ? yh[jj][i] : out[i];
    });
  }
  KN_HD void set_row(int j, int jb, const double* in) {
    for_rows<1>(jb, [&](int jj) {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) yh[jj][i] = (jj == j) ? in[i] : yh[jj][i];
    });
  }

  // right-hand side: all components (LANES = 1) or this lane's component after an all-gather
  KN_HD void eval_rhs(double t, const double* yv, double* out) {
    if constexpr (LANES == 1) {
      f.rhs(t, yv, out);
    } else {
      double ya[N];
      _Pragma("unroll") for (int k = 0; k < N; ++k) ya[k] = kn_group_get<LANES>(yv[0], k);
      out[0] = f.rhs_lane(comp, t, ya);
    }
  }

  KN_HD bool ewset(const double* yc) {
    double bad = 0.0;
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {
      const double e = rtol * fabs(yc[i]) + atol;
      if (!(e > 0.0)) bad = 1.0;
      ewt[i] = kn_div(1.0, e);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // a non-positive weight happens on no lane of a healthy sweep: one wave-wide test instead of the exchanges of a group
    // maximum in every step
    if (!KN_ANY(bad != 0.0)) return true;
#endif
    return kn_group_max<LANES>(bad) == 0.0;
  }

  // three step-ratio powers at once; with four lanes per system each of three lanes evaluates one of them
  KN_HD void pow3(double b0, double e0, double b1, double e1, double b2, double e2, double& r0, double& r1,
                  double& r2) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (LANES >= 4) {
      const int c = comp & 3;
      const double b = c == 0 ? b0 : (c == 1 ? b1 : b2), e = c == 0 ? e0 : (c == 1 ? e1 : e2);
      const double r = kn_powr(b, e);
      if constexpr (LANES == 4) {
        r0 = kn_group_get<4>(r, 0); r1 = kn_group_get<4>(r, 1); r2 = kn_group_get<4>(r, 2);
      } else {
        const int base = (int)(__lane_id() & ~(unsigned)(LANES - 1));
        r0 = __shfl(r, base); r1 = __shfl(r, base + 1); r2 = __shfl(r, base + 2);
      }
      return;
    }
#endif
    r0 = kn_powr(b0, e0); r1 = kn_powr(b1, e1); r2 = kn_powr(b2, e2);
  }

  // coefficients of the current (meth, nq): el(1..l), tesco(nq, 1..3)   [DSTODA label 150]
  KN_HD void resetcoeff(int lb) {
    (void)lb;
    el1 = cf->elco[meth - 1][nq][1];
    tq2 = cf->tesco[meth - 1][nq][2];
    rtq2 = cf->rtesco[meth - 1][nq][2];
    rcr = kn_div(rcr * el1, el0);
    el0 = el1;
    conit = over_int(0.5, nq + 2);
  }

  // new order: rows above the new l are dead in ODEPACK; here they return to zero
  KN_HD void set_order(int newq, int lb) {
    nq = newq;
    l = nq + 1;
    for_rows<2>(lb, [&](int j) {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) yh[j][i] = (j > l) ? 0.0 : yh[j][i];
    });
  }

  KN_HD void scaleh(int lb) {
    rh = fmin(rh, rmax);
    // hmxi = 0 (no maximum step): rh / max(1, |h| * hmxi * rh) == rh
    if (meth == 1) {
      irflag = 0;
      pdh = fmax(fabs(h) * pdlast, 0.000001);
      const double s = sm1(nq);
      if ((rh * pdh * 1.00001) >= s) {
        rh = kn_div(s, pdh);
        irflag = 1;
      }
    }
    double r = 1.0;
    for_rows<2>(lb, [&](int j) {
      r *= rh;
      _Pragma("unroll") for (int i = 0; i < NI; ++i) yh[j][i] *= r;
    });
    h *= rh;
    rcr *= rh;
    ialth = l;
  }

  // Pascal-triangle products of the Nordsieck array: SIGN = +1 predicts, -1 retracts.  qb >= nq is wave-uniform;
  // the triangle runs to the class of qb (rows above a lane's order are zero: x +- 0 = x).
  template <int SIGN, int Q>
  KN_HD void triangle_to() {
    _Pragma("unroll") for (int j = Q; j >= 1; --j)
      _Pragma("unroll") for (int i1 = j; i1 <= Q; ++i1)
        _Pragma("unroll") for (int i = 0; i < NI; ++i) {
          if constexpr (SIGN > 0) yh[i1][i] += yh[i1 + 1][i];
          else yh[i1][i] -= yh[i1 + 1][i];
        }
  }
  template <int SIGN>
  KN_HD void triangle(int qb) {
    if (qb <= 3) triangle_to<SIGN, 3>();
    else if (qb <= 5) triangle_to<SIGN, 5>();
    else if (qb <= 7) triangle_to<SIGN, 7>();
    else triangle_to<SIGN, ROWS - 1>();
  }

  KN_HD void retract(int lb) {
    tn = told;
    triangle<-1>(lb - 1);
  }

  // corrector failure [DSTODA label 410]: returns 1 (retry with a quarter of the step) or 2 (give up)
  KN_HD int corfailure(int lb) {
    ncf++;
    rmax = 2.0;
    retract(lb);
    if (fabs(h) <= 0.0 || ncf == MXNCF) return 2;   // hmin = 0
    rh = 0.25;
    ipup = miter;
    return 1;
  }

  // Finite-difference Jacobian, P = I - h*el0*J, LU factorisation (PRJA with miter = 2).  The factors and the pivots
  // go to the work storage (LDS on the device): only the stiff method ever reads them.
  KN_HDN void prja(double t) {
    nje++;
    ierpj = 0;
    jcur = 1;
    const double hl0 = h * el0;
    double fac = vmnorm(savf);
    double r0 = 1000.0 * fabs(h) * KN_ETA * (double)N * fac;
    if (r0 == 0.0) r0 = 1.0;
    const double sqrteta = 1.4901161193847656e-08;
    if constexpr (LANES == 1) {
  #pragma unroll
      for (int j = 0; j < N; ++j) {
        const double yj = y[j];
        const double r = fmax(sqrteta * fabs(yj), r0 / ewt[j]);
        y[j] += r;
        fac = -hl0 / r;
        f.rhs(t, y, acor);
        _Pragma("unroll") for (int i = 0; i < NI; ++i) WM(i, j) = (acor[i] - savf[i]) * fac;
        y[j] = yj;
      }
      nfe += N;
      double an = 0.0;  // fnorm: weighted max-row-sum norm of (-h*el0*J)
      _Pragma("unroll") for (int i = 0; i < NI; ++i) {
        double sum = 0.0;
        for (int j = 0; j < N; ++j) sum += fabs(WM(i, j)) / ewt[j];
        an = fmax(an, sum * ewt[i]);
      }
      pdnorm = an / fabs(hl0);
      _Pragma("unroll") for (int i = 0; i < NI; ++i) WM(i, i) += 1.0;
      // dgefa: Gaussian elimination with partial pivoting (LINPACK column-oriented variant)
  #pragma unroll
      for (int k = 0; k < N - 1; ++k) {
        int piv = k;
        double mx = fabs(WM(k, k));
        for (int i = k + 1; i < N; ++i)
          if (fabs(WM(i, k)) > mx) { mx = fabs(WM(i, k)); piv = i; }
        PIV(k) = (double)piv;
        if (WM(piv, k) == 0.0) { ierpj = 1; continue; }
        if (piv != k) { const double t2 = WM(piv, k); WM(piv, k) = WM(k, k); WM(k, k) = t2; }
        const double tinv = -1.0 / WM(k, k);
        for (int i = k + 1; i < N; ++i) WM(i, k) *= tinv;
        for (int j = k + 1; j < N; ++j) {
          double t2 = WM(piv, j);
          if (piv != k) { WM(piv, j) = WM(k, j); WM(k, j) = t2; }
          for (int i = k + 1; i < N; ++i) WM(i, j) += t2 * WM(i, k);
        }
      }
      PIV(N - 1) = (double)(N - 1);
      if (WM(N - 1, N - 1) == 0.0) ierpj = 1;
    }
    else {
      double ya[N], ea[N], row[N];
#pragma unroll
      for (int k = 0; k < N; ++k) { ya[k] = kn_group_get<LANES>(y[0], k); ea[k] = kn_group_get<LANES>(ewt[0], k); }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const double yj = ya[j];
        const double r = fmax(sqrteta * fabs(yj), r0 / ea[j]);
        ya[j] += r;
        fac = -hl0 / r;
        const double aj = f.rhs_lane(comp, t, ya);
        row[j] = (aj - savf[0]) * fac;
        ya[j] = yj;
      }
      nfe += N;
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < N; ++j) sum += fabs(row[j]) / ea[j];
      pdnorm = kn_group_max<LANES>(sum * ewt[0]) / fabs(hl0);
#pragma unroll
      for (int j = 0; j < N; ++j) row[j] += (j == comp) ? 1.0 : 0.0;
      // every lane gathers the whole matrix and factorises it in registers (dgefa, selects instead of dynamic
      // indices), then parks the factors in its column of the work storage
      double lu[N][N];
      int ipvt[N];
#pragma unroll
      for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) lu[i][j] = kn_group_get<LANES>(row[j], i);
#pragma unroll
      for (int k = 0; k < N - 1; ++k) {
        int piv = k;
        double mx = fabs(lu[k][k]);
#pragma unroll
        for (int i = k + 1; i < N; ++i)
          if (fabs(lu[i][k]) > mx) { mx = fabs(lu[i][k]); piv = i; }
        ipvt[k] = piv;
        if (mx == 0.0) { ierpj = 1; continue; }
        // swap rows' entries of column k.. between piv and k as LINPACK does (column by column)
#pragma unroll
        for (int j = k; j < N; ++j) {
          double tp = lu[k][j];
#pragma unroll
          for (int i = k + 1; i < N; ++i) tp = (i == piv) ? lu[i][j] : tp;
          const double tk = lu[k][j];
#pragma unroll
          for (int i = k + 1; i < N; ++i) lu[i][j] = (i == piv) ? tk : lu[i][j];
          lu[k][j] = tp;
        }
        const double tinv = -1.0 / lu[k][k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) lu[i][k] *= tinv;
#pragma unroll
        for (int j = k + 1; j < N; ++j) {
          const double t2 = lu[k][j];
#pragma unroll
          for (int i = k + 1; i < N; ++i) lu[i][j] += t2 * lu[i][k];
        }
      }
      ipvt[N - 1] = N - 1;
      if (lu[N - 1][N - 1] == 0.0) ierpj = 1;
#pragma unroll
      for (int i = 0; i < N; ++i) {
        PIV(i) = (double)ipvt[i];
#pragma unroll
        for (int j = 0; j < N; ++j) WM(i, j) = lu[i][j];
      }
    }
  }

  KN_HD void solsy(double* b) {  // dgesl, job = 0 (selects instead of b[piv]: b stays in registers)
    if constexpr (LANES > 1) {
      double ba[N], lu[N][N];
      int ipvt[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        ipvt[i] = (int)PIV(i);
#pragma unroll
        for (int j = 0; j < N; ++j) lu[i][j] = WM(i, j);
      }
#pragma unroll
      for (int k = 0; k < N; ++k) ba[k] = kn_group_get<LANES>(b[0], k);
#pragma unroll
      for (int k = 0; k < N - 1; ++k) {
        const int piv = ipvt[k];
        double t2 = ba[k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) t2 = (i == piv) ? ba[i] : t2;
        const double bk = ba[k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) ba[i] = (i == piv) ? bk : ba[i];
        ba[k] = t2;
#pragma unroll
        for (int i = k + 1; i < N; ++i) ba[i] += t2 * lu[i][k];
      }
#pragma unroll
      for (int k = N - 1; k >= 0; --k) {
        ba[k] /= lu[k][k];
        const double t2 = -ba[k];
#pragma unroll
        for (int i = 0; i < k; ++i) ba[i] += t2 * lu[i][k];
      }
      double own = ba[0];
#pragma unroll
      for (int k = 1; k < N; ++k) own = (k == comp) ? ba[k] : own;
      b[0] = own;
      return;
    } else {
#pragma unroll
      for (int k = 0; k < N - 1; ++k) {
        const int piv = (int)PIV(k);
        double t2 = b[k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) t2 = (i == piv) ? b[i] : t2;
        const double bk = b[k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) b[i] = (i == piv) ? bk : b[i];
        b[k] = t2;
#pragma unroll
        for (int i = k + 1; i < N; ++i) b[i] += t2 * WM(i, k);
      }
#pragma unroll
      for (int k = N - 1; k >= 0; --k) {
        b[k] /= WM(k, k);
        const double t2 = -b[k];
#pragma unroll
        for (int i = 0; i < k; ++i) b[i] += t2 * WM(i, k);
      }
    }
  }

  KN_HDN void methodswitch(double dsm, double* rhp) {
    int nqm1, nqm2;
    double rh1, rh2, rh1it, exm2, dm2, exm1, dm1, alpha, exsm;
    double row[NI];
    if (meth == 1) {
      if (nq > 5) return;
      if (dsm <= (100.0 * pnorm * KN_ETA) || pdest == 0.0) {
        if (irflag == 0) return;
        rh2 = 2.0;
        nqm2 = nq < MXORDS ? nq : MXORDS;
      } else {
        exsm = over_int(1.0, l);
        rh1 = kn_div(1.0, 1.2 * kn_powr(dsm, exsm) + 0.0000012);
        rh1it = 2.0 * rh1;
        pdh = pdlast * fabs(h);
        if ((pdh * rh1) > 0.00001) rh1it = kn_div(sm1(nq), pdh);
        rh1 = fmin(rh1, rh1it);
        if (nq > MXORDS) {   // (unreachable behind `nq > 5` above; kept as in DSTODA)
          nqm2 = MXORDS;
          const int lm2 = MXORDS + 1;
          exm2 = over_int(1.0, lm2);
          get_row(lm2 + 1, ROWS, row);
          dm2 = kn_div(vmnorm(row), cf->cm2[MXORDS]);
          rh2 = kn_div(1.0, 1.2 * kn_powr(dm2, exm2) + 0.0000012);
        } else {
          dm2 = dsm * kn_div(cf->cm1[nq], cf->cm2[nq]);
          rh2 = kn_div(1.0, 1.2 * kn_powr(dm2, exsm) + 0.0000012);
          nqm2 = nq;
        }
        if (rh2 < ratio * rh1) return;
      }
      *rhp = rh2;
      icount = 20;
      meth = 2;
      miter = 2;
      pdlast = 0.0;
      set_order(nqm2, ROWS);
      return;
    }
    exsm = over_int(1.0, l);
    if (MXORDN < nq) {
      nqm1 = MXORDN;
      const int lm1 = MXORDN + 1;
      exm1 = over_int(1.0, lm1);
      get_row(lm1 + 1 > ROWS ? ROWS : lm1 + 1, ROWS, row);
      dm1 = kn_div(vmnorm(row), cf->cm1[MXORDN]);
      rh1 = kn_div(1.0, 1.2 * kn_powr(dm1, exm1) + 0.0000012);
    } else {
      dm1 = dsm * kn_div(cf->cm2[nq], cf->cm1[nq]);
      rh1 = kn_div(1.0, 1.2 * kn_powr(dm1, exsm) + 0.0000012);
      nqm1 = nq;
      exm1 = exsm;
    }
    rh1it = 2.0 * rh1;
    pdh = pdnorm * fabs(h);
    if ((pdh * rh1) > 0.00001) rh1it = kn_div(sm1(nqm1), pdh);
    rh1 = fmin(rh1, rh1it);
    rh2 = kn_div(1.0, 1.2 * kn_powr(dsm, exsm) + 0.0000012);
    if ((rh1 * ratio) < (5.0 * rh2)) return;
    alpha = fmax(0.001, rh1);
    dm1 *= kn_powr(alpha, exm1);
    if (dm1 <= 1000.0 * KN_ETA * pnorm) return;
    *rhp = rh1;
    icount = 20;
    meth = 1;
    miter = 0;
    pdlast = 0.0;
    set_order(nqm1, ROWS);
  }

  // Order selection [DSTODA labels 520-620].  want_up: the step was accepted and the order may rise (rhup from the
  // saved correction); after a failed error test rhup = 0.  Returns ODEPACK's orderflag.
  KN_HDN int orderswitch(bool want_up, double dsm, int lb) {
    int newq;
    const double exsm = over_int(1.0, l);
    // the three candidates' error norms, then their roots in one go
    const bool up = want_up && l != lmax, dn = nq != 1;
    double dup = 1.0, ddn = 1.0;
    if (up) {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) savf[i] = acor[i] - ysave[i];
      dup = over_tq(vmnorm(savf), cf->tesco[meth - 1][nq][3], cf->rtesco[meth - 1][nq][3]);   // (order selection only)
    }
    if (dn) {
      double row[NI];
      get_row(l, lb, row);
      ddn = over_tq(vmnorm(row), cf->tesco[meth - 1][nq][1], cf->rtesco[meth - 1][nq][1]);
    }
    double psm, pdn, pup;
    pow3(dsm, exsm, ddn, over_int(1.0, nq), dup, over_int(1.0, l + 1), psm, pdn, pup);
    double rhup = up ? kn_div(1.0, 1.4 * pup + 0.0000014) : 0.0;
    double rhsm = kn_div(1.0, 1.2 * psm + 0.0000012);
    double rhdn = dn ? kn_div(1.0, 1.3 * pdn + 0.0000013) : 0.0;
    if (meth == 1) {
      pdh = fmax(fabs(h) * pdlast, 0.000001);
      if (l < lmax) rhup = fmin(rhup, kn_div(sm1(l), pdh));
      rhsm = fmin(rhsm, kn_div(sm1(nq), pdh));
      if (nq > 1) rhdn = fmin(rhdn, kn_div(sm1(nq - 1), pdh));
      pdest = 0.0;
    }
    if (rhsm >= rhup) {
      if (rhsm >= rhdn) {
        newq = nq;
        rh = rhsm;
      } else {
        newq = nq - 1;
        rh = rhdn;
        if (kflag < 0 && rh > 1.0) rh = 1.0;
      }
    } else {
      if (rhup <= rhdn) {
        newq = nq - 1;
        rh = rhdn;
        if (kflag < 0 && rh > 1.0) rh = 1.0;
      } else {
        rh = rhup;
        if (rh >= 1.1) {
          const double r = over_int(cf->elco[meth - 1][nq][l], l);   // el(l) of the order that took the step
          nq = l;
          l = nq + 1;
          double row[NI];
          _Pragma("unroll") for (int i = 0; i < NI; ++i) row[i] = acor[i] * r;
          set_row(l, lb, row);
          return 2;
        }
        ialth = 3;
        return 0;
      }
    }
    if (meth == 1) {
      if ((rh * pdh * 1.00001) < sm1(newq))
        if (kflag == 0 && rh < 1.1) { ialth = 3; return 0; }
    } else {
      if (kflag == 0 && rh < 1.1) { ialth = 3; return 0; }
    }
    if (kflag <= -2) rh = fmin(rh, 0.2);
    if (newq == nq) return 1;
    set_order(newq, lb);
    return 2;
  }

  // ---- phases -----------------------------------------------------------------------------------
  // before a step: DLSODA's checks at the top of its loop (label 250) and DSTODA's prologue
  KN_HDN void phase_top(int mxstep) {
    if (nst > 0) {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) y[i] = yh[1][i];
      if (!ewset(y)) { ret = -6; ph = PH_DONE; return; }
    }
    if (nst >= mxstep) { ret = -1; ph = PH_DONE; return; }
    const double tolsf = KN_ETA * vmnorm(yh[1]);
    if (tolsf > 0.01) { ret = -2; ph = PH_DONE; return; }
    kflag = 0;
    told = tn;
    ncf = 0;
    ierpj = 0;
    jcur = 0;
    delp = 0.0;
    pdh = 0.0;
    rh = 1.0;
    if (jstart == 0) {
      lmax = maxord + 1;
      nq = 1;
      l = 2;
      ialth = 2;
      rmax = 10000.0;
      rcr = 0.0;
      el0 = 1.0;
      crate = 0.7;
      hold = h;
      nslp = 0;
      ipup = miter;
      icount = 20;
      irflag = 0;
      pdest = 0.0;
      pdlast = 0.0;
      ratio = 5.0;
      resetcoeff(lhi);
    }
    if (jstart == -1) {
      ipup = miter;
      lmax = maxord + 1;
      if (ialth == 1) ialth = 2;
      if (meth != mused) {
        ialth = l;
        resetcoeff(lhi);
      }
      if (h != hold) {
        rh = kn_div(h, hold);
        h = hold;
        scaleh(lhi);
      }
    }
    if (jstart > 0 && h != hold) {
      rh = kn_div(h, hold);
      h = hold;
      scaleh(lhi);
    }
    ph = PH_PRED;
  }

  // predict, start the corrector [DSTODA label 200 and the head of the corrector loop]
  KN_HD void phase_pred() {
    if (fabs(rcr - 1.0) > 0.3) ipup = miter;
    if (nst >= nslp + MSBP) ipup = miter;
    tn += h;
    triangle<1>(lhi - 1);
    pnorm = vmnorm(yh[1]);
    rate = 0.0;
    m = 0;
    del = 0.0;
    _Pragma("unroll") for (int i = 0; i < NI; ++i) y[i] = yh[1][i];
    ph = PH_CORR;
  }

  // one corrector iteration on the right-hand side just evaluated at (tn, y) [DSTODA labels 220-430]
  KN_HDN void phase_corr() {
    if (m == 0) {
      if (ipup > 0) {
        prja(tn);
        ipup = 0;
        rcr = 1.0;
        nslp = nst;
        crate = 0.7;
        if (ierpj != 0) { corr_failed(); return; }
      }
      _Pragma("unroll") for (int i = 0; i < NI; ++i) acor[i] = 0.0;
    }
    if (miter == 0) {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) {
        savf[i] = h * savf[i] - yh[2][i];
        y[i] = savf[i] - acor[i];
      }
      del = vmnorm(y);
      // the norm of the accumulated correction (what the error test looks at when m > 0) does not depend on del:
      // evaluated here its exchanges overlap those of del instead of following the convergence test
      if (m != 0) anorm = vmnorm(savf);
      _Pragma("unroll") for (int i = 0; i < NI; ++i) {
        y[i] = yh[1][i] + el1 * savf[i];
        acor[i] = savf[i];
      }
    } else {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) y[i] = h * savf[i] - (yh[2][i] + acor[i]);
      solsy(y);
      del = vmnorm(y);
      _Pragma("unroll") for (int i = 0; i < NI; ++i) {
        acor[i] += y[i];
        y[i] = yh[1][i] + el1 * acor[i];
      }
      if (m != 0) anorm = vmnorm(acor);
    }
    if (del <= 100.0 * pnorm * KN_ETA) { ph = PH_ERR; return; }
    if (m != 0 || meth != 1) {
      if (m != 0) {
        double rm = 1024.0;
        if (del <= (1024.0 * delp)) rm = kn_div(del, delp);
        rate = fmax(rate, rm);
        crate = fmax(0.2 * crate, rm);
      }
#if defined(__HIP_DEVICE_COMPILE__)
      const double dcon = del * fmin(1.0, 1.5 * crate) * (rtq2 * (double)(2 * (nq + 2)));   // 1 / conit = 2 (nq + 2)
#else
      const double dcon = del * fmin(1.0, 1.5 * crate) / (tq2 * conit);
#endif
      if (dcon <= 1.0) {
        pdest = fmax(pdest, kn_div(rate, fabs(h * el1)));
        if (pdest != 0.0) pdlast = pdest;
        ph = PH_ERR;
        return;
      }
    }
    m++;
    if (m == MAXCOR || (m >= 2 && del > 2.0 * delp)) {
      if (miter == 0 || jcur == 1) { corr_failed(); return; }
      // the Jacobian is out of date: restart the corrector with a fresh one
      ipup = miter;
      m = 0;
      rate = 0.0;
      del = 0.0;
      _Pragma("unroll") for (int i = 0; i < NI; ++i) y[i] = yh[1][i];
    } else {
      delp = del;
    }
  }

  KN_HD void corr_failed() {
    const int corflag = corfailure(lhi);
    if (corflag == 2) {   // kflag = -2: repeated corrector failures
      ret = -5;
      ph = PH_DONE;
      return;
    }
    rh = fmax(rh, 0.0);
    scaleh(lhi);
    ph = PH_PRED;
  }

  // after a converged corrector: error test, then order / step / method selection [labels 450-700], and back in
  // DLSODA the method-switch bookkeeping and the test for tout
  KN_HDN void phase_err(double tout, double* y0) {
    const int lb = lhi + 1 > ROWS ? ROWS : lhi + 1;   // an order increase in this phase can reach lhi + 1
    jcur = 0;
    double dsm;
    if (m == 0) dsm = over_tq(del, tq2, rtq2);
    else dsm = over_tq(anorm, tq2, rtq2);
    if (dsm <= 1.0) {
      kflag = 0;
      nst++;
      hu = h;
      nqu = nq;
      mused = meth;
      for_rows<1>(lb, [&](int j) {
        const double elj = cf->elco[meth - 1][nq][j];
        _Pragma("unroll") for (int i = 0; i < NI; ++i) yh[j][i] += elj * acor[i];
      });
      const double tq2u = tq2, rtq2u = rtq2;   // tesco(nqu, 2) of the method that took the step (used by "endstoda")
      icount--;
      bool ended = false;
      if (icount < 0) {
        methodswitch(dsm, &rh);
        if (meth != mused) {
          rh = fmax(rh, 0.0);
          scaleh(ROWS);
          rmax = 10.0;
          ended = true;
        }
      }
      if (!ended) {
        ialth--;
        if (ialth == 0) {
          stamp(5, true);
          const int orderflag = orderswitch(true, dsm, lb);
          stamp(8, true);
          if (orderflag != 0) {
            if (orderflag == 2) resetcoeff(lb);
            rh = fmax(rh, 0.0);
            scaleh(lb);
            rmax = 10.0;
          }
          stamp(9, true);
        } else if (!(ialth > 1 || l == lmax)) {
          _Pragma("unroll") for (int i = 0; i < NI; ++i) ysave[i] = acor[i];
        }
      }
      {   // "endstoda"
        const double r = over_tq(1.0, tq2u, rtq2u);
        _Pragma("unroll") for (int i = 0; i < NI; ++i) acor[i] *= r;
        hold = h;
        jstart = 1;
      }
      // DLSODA after a successful return of DSTODA
      if (meth != mused) {
        tsw = tn;
        maxord = meth == 2 ? MXORDS : MXORDN;
        jstart = -1;
      }
      if ((tn - tout) * h < 0.0) { ph = PH_TOP; return; }
      // intdy, k = 0: Horner from row l down
      const double s = kn_div(tout - tn, h);
      _Pragma("unroll") for (int i = 0; i < NI; ++i) y0[i] = 0.0;
      for_rows_down(lb, [&](int j) {
        _Pragma("unroll") for (int i = 0; i < NI; ++i)
          y0[i] = (j < l) ? yh[j][i] + s * y0[i] : ((j == l) ? yh[j][i] : y0[i]);
      });
      ret = 0;
      ph = PH_DONE;
      return;
    }
    // error test failed
    kflag--;
    retract(lb);
    rmax = 2.0;
    if (fabs(h) <= 0.0) {   // hmin = 0
      ret = -4;
      ph = PH_DONE;
      return;
    }
    if (kflag > -3) {
      const int orderflag = orderswitch(false, dsm, lb);
      if (orderflag == 2) resetcoeff(lb);
      if (orderflag == 0) rh = fmin(rh, 0.2);
      rh = fmax(rh, 0.0);
      scaleh(lb);
      ph = PH_PRED;
      return;
    }
    if (kflag == -10) {
      ret = -4;
      ph = PH_DONE;
      return;
    }
    // three failures in a row: drop to order 1 with a tenth of the step and a fresh derivative
    rh = 0.1;
    h *= rh;
    _Pragma("unroll") for (int i = 0; i < NI; ++i) y[i] = yh[1][i];
    ph = PH_RESTART;
  }

  KN_HD void phase_restart() {
    _Pragma("unroll") for (int i = 0; i < NI; ++i) yh[2][i] = h * savf[i];
    ipup = miter;
    ialth = 5;
    if (nq != 1) {
      set_order(1, lhi);
      resetcoeff(lhi);
    }
    ph = PH_PRED;
  }

  // Integrate y0 from t0 to tout (istate = 1, itask = 1).  Returns 0 on success, a negative
  // ODEPACK-style code otherwise.  On success y0 holds y(tout).
  KN_HDN int integrate(const LsodaCoef* coef, double* work_, double* y0, double t0, double tout,
                       double rtol_, double atol_, int mxstep, int comp_ = 0) {
    cf = coef;
    work = work_;
    comp = comp_;
    rtol = rtol_;
    atol = atol_;
    _Pragma("unroll") for (int j = 0; j <= ROWS; ++j) {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) yh[j][i] = 0.0;
    }
    _Pragma("unroll") for (int i = 0; i < NI; ++i) ysave[i] = 0.0;
    tq2 = 0.0;
    rtq2 = 0.0;
    tn = t0;
    tsw = t0;
    maxord = MXORDN;
    jstart = 0;
    nst = 0; nje = 0; nslp = 0;
    hu = 0.0; nqu = 0; mused = 0; miter = 0; meth = 1;
    nq = 1; l = 2; lhi = 2; lseen = 2;
    ret = 0;
    ph = PH_TOP;
    _Pragma("unroll") for (int i = 0; i < NI; ++i) y[i] = y0[i];
    eval_rhs(t0, y, savf);
    nfe = 1;
    _Pragma("unroll") for (int i = 0; i < NI; ++i) { yh[1][i] = y[i]; yh[2][i] = savf[i]; }
    if (!ewset(y)) { ret = -6; ph = PH_DONE; }
    // initial step size (DLSODA block c)
    const double tdist = fabs(tout - t0);
    const double w0 = fmax(fabs(t0), fabs(tout));
    if (ph != PH_DONE && tdist < 2.0 * KN_ETA * w0) { ret = -3; ph = PH_DONE; }
    double tol = rtol;
    if (tol <= 0.0) {
      _Pragma("unroll") for (int i = 0; i < NI; ++i) {
        const double ayi = fabs(y[i]);
        if (ayi != 0.0) tol = fmax(tol, kn_div(atol, ayi));
      }
      tol = kn_group_max<LANES>(tol);
    }
    tol = fmax(tol, 100.0 * KN_ETA);
    tol = fmin(tol, 0.001);
    double sum = vmnorm(yh[2]);
    sum = kn_div(1.0, tol * w0 * w0) + tol * sum * sum;
    double h0 = kn_div(1.0, sqrt(sum));
    h0 = fmin(h0, tdist);
    h0 = (tout - t0) >= 0.0 ? h0 : -h0;
    h = h0;
    _Pragma("unroll") for (int i = 0; i < NI; ++i) yh[2][i] *= h0;

    // one trip = one right-hand-side evaluation for every lane that is still integrating
    const int trips_max = 16 * mxstep + 64;
    stamp(6, true);
    for (int trip = 0; KN_ANY(ph != PH_DONE); ++trip) {
      if (trip >= trips_max) {   // cannot happen (mxstep bounds every path); never leave a wave spinning
        if (ph != PH_DONE) { ret = -7; ph = PH_DONE; }
        break;
      }
      // keep l <= lhi on every running lane (an order rises by one per trip at most) and let lhi fall when it can;
      // orders change a few times per sweep, so the common trip pays for one wave-wide test only
      if (KN_ANY2(ph != PH_DONE, l != lseen)) {
        lseen = l;
        if (KN_ANY2(ph != PH_DONE, l > lhi)) ++lhi;
        else
          while (lhi > 2 && !KN_ANY2(ph != PH_DONE, l >= lhi)) --lhi;
      }
      bool ran = false;
      if constexpr (STAMPS) { stamp(0, true); ran = KN_ANY(ph == PH_TOP); }
      if (ph == PH_TOP) phase_top(mxstep);
      if constexpr (STAMPS) { stamp(1, ran); ran = KN_ANY(ph == PH_PRED); }
      if (ph == PH_PRED) phase_pred();
      if constexpr (STAMPS) { stamp(2, ran); ran = KN_ANY(ph == PH_CORR || ph == PH_RESTART); }
      if (ph == PH_CORR || ph == PH_RESTART) {
        eval_rhs(tn, y, savf);
        nfe++;
      }
      if constexpr (STAMPS) { stamp(3, ran); ran = KN_ANY(ph == PH_CORR || ph == PH_RESTART); }
      if (ph == PH_RESTART) phase_restart();
      else if (ph == PH_CORR) phase_corr();
      if constexpr (STAMPS) { stamp(4, ran); ran = KN_ANY(ph == PH_ERR); }
      if (ph == PH_ERR) phase_err(tout, y0);
      if constexpr (STAMPS) stamp(10, ran);
    }
    return ret;
  }
};
