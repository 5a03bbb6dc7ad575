// Fused iteration loops of the device Krylov solves (the reference's KSP solves,
// src/knpemi/pdeSolver.py:24-35,99-110).
//
// At the sizes of the idealized runs (2.6e4 .. 1.9e5 unknowns) every kernel of a solve is at the launch floor, so the
// cost of an iteration is its NUMBER of dependent launches (~3 us of gap + ~2-3 us of kernel each): the plain loops of
// kernels_krylov.hip spend ~25 (CG) / ~45 (BiCGStab) per iteration -- 4 per multigrid level and cycle, one per dot
// product, one per vector update.  Here:
//   * the V(1,1) cycle uses merged transfer operators (kernels_amg.hip: Rm = R (I - w A D^-1), Pm = (I - w D^-1 A) P):
//     from a zero guess  x = w D^-1 (r + t) + Pm e_c,  t = (I - w A D^-1) r,  e_c = cycle(Rm r)  -- restriction + residual
//     are ONE launch per level on the way down, prolongation + post-smoothing ONE on the way up, the coarsest level a dense
//     inverse: 2 (levels - 1) + 1 launches per cycle instead of 4 (levels - 1) + 1.  The cycle works on the operator the
//     hierarchy was built from (a fixed symmetric preconditioner for CG); residuals use the current operator;
//   * vector updates are folded into the SpMV that consumes them: the finest `down` kernel forms its input on the fly
//     (r - alpha q / r + beta (p - omega v) / r - alpha v), stores it, and for CG also advances x; the search direction
//     p = z + beta p is formed inside the SpMV q = A p (ping-pong buffers);
//   * dot products are block partials + a last-block-done ticket inside the kernel that produces the vector, and the
//     scalar algebra (alpha, beta, omega, convergence) runs in that last block: fixed summation order, bit-reproducible;
//   * convergence is decided on the device: once |r| <= target the remaining kernels of a chunk return at once, the
//     host reads the scalars once per chunk of iterations, and the iteration count is the device's.
// CG iteration: 2 + cycle launches (6 with three levels), BiCGStab: 3 + 2 cycles (13).
#include <cmath>
#include <cstdio>

#include "knpemi_internal.h"

namespace {

// device scalars (the first 13 in the layout of kernels_krylov.hip)
enum { S_RHO0 = 0, S_RHO1, S_PAP, S_ALPHA, S_BETA, S_OMEGA, S_RR, S_BB, S_TS, S_TT, S_MEAN, S_FLAG, S_RV,
       S_DONE, S_IT, S_TARGET2, S_NF };
enum { F_RHO_ZERO = 1, F_RV_ZERO = 2, F_OMEGA_ZERO = 4, F_PAP_ZERO = 8 };
enum { DONE_NO = 0, DONE_CONVERGED = 1, DONE_BAD_RHS = 2, DONE_BAD_START = 3 };
constexpr int FT = 256, LPR = 16;
constexpr int KN_PB = 2048;  // most blocks of a kernel that produces a dot product (they walk the rows with a grid stride)
// partial-sum arrays (KN_PB doubles each)
enum { P_PQ = 0, P_RR, P_RZ, P_RV, P_TS, P_TT, P_RHR, P_N };

// sum_j vals[j] f(colind[j]) over row `row` with L lanes per row (all L lanes return the sum)
template <int L, class F>
__device__ __forceinline__ double row_sum(const int* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ v,
                                          int row, int lane, F&& f) {
  double acc = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  const int a = rp[row], b = rp[row + 1];
  int j = a + lane;
  for (; j + 3 * L < b; j += 4 * L) {     // the merged transfer operators have 50-100 entries per row: four gathers in flight
    const int c0 = ci[j], c1 = ci[j + L], c2 = ci[j + 2 * L], c3 = ci[j + 3 * L];
    const double v0 = v[j], v1 = v[j + L], v2 = v[j + 2 * L], v3 = v[j + 3 * L];
    acc += v0 * f(c0); acc1 += v1 * f(c1); acc2 += v2 * f(c2); acc3 += v3 * f(c3);
  }
  for (; j < b; j += L) acc += v[j] * f(ci[j]);
  acc = (acc + acc1) + (acc2 + acc3);
#pragma unroll
  for (int m = L / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  return acc;
}

// Dot products without atomics and without a kernel of their own.  A device-scope atomic on one address costs ~50 ns per
// arriving block on this chip and a "last block done" ticket adds a store -> fence -> atomic -> fence -> load chain of
// ~7-15 us to the kernel that carries it (measured: the whole cost of the separate dot-product kernels of
// kernels_krylov.hip).  Instead the PRODUCER of a vector leaves one partial sum per block (plain stores; the kernel
// boundary publishes them) and every block of the CONSUMER adds the <= 2048 partials up again, in the same fixed order:
// one <= 16 KiB read from L2 and an LDS tree per block, identical bits in every block, bit-reproducible.
// sum over the 256 threads of a block: butterflies inside the waves (no barrier), the four wave sums through LDS -- one
// barrier instead of the eight of an LDS tree; the order is fixed, every thread returns the same bits
template <int ND>
__device__ __forceinline__ void block_sum(double (&v)[ND]) {
  __shared__ double sh[ND][FT / 64];
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < ND; ++k) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m);
  }
  if ((t & 63) == 0)
#pragma unroll
    for (int k = 0; k < ND; ++k) sh[k][t >> 6] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ND; ++k) v[k] = (sh[k][0] + sh[k][1]) + (sh[k][2] + sh[k][3]);
  __syncthreads();      // the buffer is reused by the next call
}

template <int ND>
__device__ __forceinline__ void block_partials(double (&v)[ND], double* const (&dst)[ND]) {
  block_sum<ND>(v);
  if (threadIdx.x == 0)
#pragma unroll
    for (int k = 0; k < ND; ++k) dst[k][blockIdx.x] = v[k];
}

template <int ND>
__device__ __forceinline__ void totals(const double* const (&src)[ND], int np, double (&out)[ND]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < ND; ++k) {
    double s = 0.0;
    for (int i = t; i < np; i += FT) s += src[k][i];
    out[k] = s;
  }
  block_sum<ND>(out);
}

__device__ __forceinline__ void raise_flag(double* sc, int bit) { sc[S_FLAG] = (double)((int)sc[S_FLAG] | bit); }

struct Red {                 // where the dot products of a solve live
  double* sc;                // device scalars
  double* part;              // P_N arrays of KN_PB partial sums
  __device__ __forceinline__ double* arr(int which) const { return part + (size_t)which * KN_PB; }
};

// what the finest `down` kernel takes as its input vector
enum { IN_PLAIN = 0, IN_CG, IN_BI_P, IN_BI_S };

struct DownArgs {
  int n, nc;                                   // rows of A (this level), rows of Rm (next level)
  const int *arp, *aci; const double* av;      // frozen operator of this level
  const int *rrp, *rci; const double* rv;      // merged restriction
  const double* dinv; double omega;
  const double* r;                             // input (IN_PLAIN) / r of the Krylov loop
  const double *u, *w;                         // IN_CG: q, p   IN_BI_P: p, v   IN_BI_S: v, -
  double* out;                                 // the formed input is stored here (NULL: IN_PLAIN)
  double* x;                                   // IN_CG: x += alpha p
  double* t; double* rc;                       // t = (I - w A D^-1) in, rc = Rm in
  Red red;
  int np;                                      // partial sums the previous producer left
  int k;                                       // iteration (parity of the rho slots; BiCGStab: completed iterations)
};

// The scalars the finest `down` kernel needs before it can form its input (every block computes them from the producer's
// partial sums, block 0 records them); false: the iteration has converged, nothing is left to do.
template <int IN>
__device__ __forceinline__ bool down_head(const DownArgs& a, double& alpha, double& beta, double& omb) {
  double* sc = a.red.sc;
  if constexpr (IN == IN_CG) {              // alpha = rho / p.Ap
    const double* const src[1] = {a.red.arr(P_PQ)};
    double pq[1];
    totals<1>(src, a.np, pq);
    const double rho = sc[S_RHO0 + (a.k & 1)];
    alpha = pq[0] != 0.0 ? rho / pq[0] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc[S_PAP] = pq[0]; sc[S_ALPHA] = alpha;
      if (pq[0] == 0.0) raise_flag(sc, F_PAP_ZERO);
    }
  } else if constexpr (IN == IN_BI_P) {     // convergence of the previous iteration, rho, beta
    const double* const src[2] = {a.red.arr(P_RHR), a.red.arr(P_RR)};
    double d[2];
    totals<2>(src, a.np, d);
    const bool done = !(d[1] > sc[S_TARGET2]);
    const double rho_old = sc[S_RHO0 + ((a.k + 1) & 1)], al = sc[S_ALPHA];
    omb = sc[S_OMEGA];
    // rho_old = 0: the first direction (or a restart), p = r.  omega = 0 with a residual left is a breakdown: beta = 0
    // restarts the recurrence from p = r as well (the host re-bases rhat when it sees the flag)
    const bool first = rho_old == 0.0;
    beta = (!first && omb != 0.0) ? (d[0] / rho_old) * (al / omb) : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc[S_RR] = d[1]; sc[S_IT] = (double)a.k;
      if (done) sc[S_DONE] = DONE_CONVERGED;
      else {
        sc[S_RHO0 + (a.k & 1)] = d[0]; sc[S_BETA] = beta;
        if (!first && omb == 0.0) raise_flag(sc, F_OMEGA_ZERO);
        if (d[0] == 0.0) raise_flag(sc, F_RHO_ZERO);
      }
    }
    if (done) return false;
  } else if constexpr (IN == IN_BI_S) {     // alpha = rho / rhat.v
    const double* const src[1] = {a.red.arr(P_RV)};
    double rv[1];
    totals<1>(src, a.np, rv);
    const double rho = sc[S_RHO0 + (a.k & 1)];
    alpha = rv[0] != 0.0 ? rho / rv[0] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc[S_RV] = rv[0]; sc[S_ALPHA] = alpha;
      if (rv[0] == 0.0) raise_flag(sc, F_RV_ZERO);
    }
  }
  return true;
}

// Rows of A with 16 lanes each, rows of the merged restriction (hundreds of entries) with a wavefront each.
template <int IN>
__global__ __launch_bounds__(FT) void down_kernel(DownArgs a) {
  double* sc = a.red.sc;
  if (sc[S_DONE] != 0.0) return;
  double alpha = 0.0, beta = 0.0, omb = 0.0;
  if (!down_head<IN>(a, alpha, beta, omb)) return;
  auto in = [&](int j) -> double {
    if constexpr (IN == IN_PLAIN) return a.r[j];
    else if constexpr (IN == IN_CG) return a.r[j] - alpha * a.u[j];
    else if constexpr (IN == IN_BI_P) return beta != 0.0 ? a.r[j] + beta * (a.u[j] - omb * a.w[j]) : a.r[j];
    else return a.r[j] - alpha * a.u[j];
  };
  double rr[1] = {0.0};
  // A rows: FT / 16 per block and pass; restriction rows: FT / 64 per block and pass
  const int nra = (a.n + FT / LPR - 1) / (FT / LPR);      // block-passes over A
  const int nrr = (a.nc + FT / 64 - 1) / (FT / 64);       // block-passes over Rm
  for (int pass = blockIdx.x; pass < nra + nrr; pass += gridDim.x) {
    if (pass < nra) {
      const int row = pass * (FT / LPR) + threadIdx.x / LPR, lane = threadIdx.x % LPR;
      if (row < a.n) {
        const double mine = in(row);
        const double acc = row_sum<LPR>(a.arp, a.aci, a.av, row, lane, [&](int c) { return a.dinv[c] * in(c); });
        if (lane == 0) {
          a.t[row] = mine - a.omega * acc;
          if constexpr (IN != IN_PLAIN) a.out[row] = mine;
          if constexpr (IN == IN_CG) { a.x[row] += alpha * a.w[row]; rr[0] += mine * mine; }
        }
      }
    } else {
      const int k = (pass - nra) * (FT / 64) + threadIdx.x / 64, lane = threadIdx.x % 64;
      if (k < a.nc) {
        const double acc = row_sum<64>(a.rrp, a.rci, a.rv, k, lane, in);
        if (lane == 0) a.rc[k] = acc;
      }
    }
  }
  if constexpr (IN == IN_CG) {     // |r_new|^2 for the convergence test (dense_kernel)
    double* const dst[1] = {a.red.arr(P_RR)};
    block_partials<1>(rr, dst);
  }
}

// BiCGStab's two inputs of the cycle, p = r + beta (p - omega v) and s = r - alpha v, formed ONCE: folded into the finest
// `down` kernel every matrix entry of A and of the merged restriction gathers three (two) vectors instead of one --
// 50 us for the 52 848-row concentration system of config 2 where this kernel (6 us) + the plain `down` kernel take 25.
// (CG's r - alpha q stays folded: two gathers cost about what the extra launch would.)
template <int IN>
__global__ __launch_bounds__(FT) void form_kernel(DownArgs a) {
  if (a.red.sc[S_DONE] != 0.0) return;
  double alpha = 0.0, beta = 0.0, omb = 0.0;
  if (!down_head<IN>(a, alpha, beta, omb)) return;
  for (int j = blockIdx.x * FT + threadIdx.x; j < a.n; j += gridDim.x * FT) {
    if constexpr (IN == IN_BI_P) a.out[j] = beta != 0.0 ? a.r[j] + beta * (a.u[j] - omb * a.w[j]) : a.r[j];
    else a.out[j] = a.r[j] - alpha * a.u[j];
  }
}

struct UpArgs {
  int n;
  const int *prp, *pci; const double* pv;      // merged prolongation
  const double* dinv; double omega;
  const double *r, *t, *ec;                    // this level's input and residual, the coarse solution
  double* x;                                   // w D^-1 (r + t) + Pm ec
  Red red;
};

// DOTS: CG's r.z (the finest level of the cycle)
template <bool DOTS>
__global__ __launch_bounds__(FT) void up_kernel(UpArgs a) {
  if (a.red.sc[S_DONE] != 0.0) return;
  const int lane = threadIdx.x % LPR;
  double rz[1] = {0.0};
  for (int row = (blockIdx.x * FT + threadIdx.x) / LPR; row < a.n; row += gridDim.x * (FT / LPR)) {
    const double acc = row_sum<LPR>(a.prp, a.pci, a.pv, row, lane, [&](int c) { return a.ec[c]; });
    if (lane == 0) {
      const double z = a.omega * a.dinv[row] * (a.r[row] + a.t[row]) + acc;
      a.x[row] = z;
      if constexpr (DOTS) rz[0] += a.r[row] * z;
    }
  }
  if constexpr (DOTS) {
    double* const dst[1] = {a.red.arr(P_RZ)};
    block_partials<1>(rz, dst);
  }
}

// coarsest level: e = Minv r, one wavefront per row.  CHECK (CG): the convergence test of the iteration whose finest
// down kernel left |r|^2 -- the rest of the cycle and of the chunk returns at once when it is met.
template <bool CHECK>
__global__ __launch_bounds__(FT) void dense_kernel(int n, const double* __restrict__ Minv, const double* __restrict__ r,
                                                   double* __restrict__ x, Red red, int np, int k) {
  double* sc = red.sc;
  if (sc[S_DONE] != 0.0) return;
  if constexpr (CHECK) {
    const double* const src[1] = {red.arr(P_RR)};
    double rr[1];
    totals<1>(src, np, rr);
    const bool done = !(rr[0] > sc[S_TARGET2]);      // also stops on a NaN: the host reports it
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc[S_RR] = rr[0]; sc[S_IT] = (double)(k + 1);
      if (done) sc[S_DONE] = DONE_CONVERGED;
    }
    if (done) return;
  }
  const int row = (blockIdx.x * FT + threadIdx.x) >> 6, l = threadIdx.x & 63;
  double acc = 0.0;
  if (row < n)
    for (int j = l; j < n; j += 64) acc += Minv[(size_t)row * n + j] * r[j];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (row < n && l == 0) x[row] = acc;
}

// CG: beta = rho / rho_old, p_new = z + beta p (stored), q = A p_new, p.q
__global__ __launch_bounds__(FT) void cg_dir_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                    const double* __restrict__ v, const double* __restrict__ z,
                                                    const double* __restrict__ p, double* __restrict__ pn, double* __restrict__ q,
                                                    Red red, int np, int k) {
  double* sc = red.sc;
  if (sc[S_DONE] != 0.0) return;
  const double* const src[1] = {red.arr(P_RZ)};
  double rho[1];
  totals<1>(src, np, rho);
  const double rho_old = sc[S_RHO0 + ((k + 1) & 1)];
  const double beta = rho_old != 0.0 ? rho[0] / rho_old : 0.0;     // the first direction is z (whatever the buffer of p holds)
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc[S_RHO0 + (k & 1)] = rho[0]; sc[S_BETA] = beta; }
  const int lane = threadIdx.x % LPR;
  double pq[1] = {0.0};
  for (int row = (blockIdx.x * FT + threadIdx.x) / LPR; row < n; row += gridDim.x * (FT / LPR)) {
    const double mine = beta != 0.0 ? z[row] + beta * p[row] : z[row];
    const double acc = row_sum<LPR>(rp, ci, v, row, lane, [&](int c) { return beta != 0.0 ? z[c] + beta * p[c] : z[c]; });
    if (lane == 0) { pn[row] = mine; q[row] = acc; pq[0] += mine * acc; }
  }
  double* const dst[1] = {red.arr(P_PQ)};
  block_partials<1>(pq, dst);
}

// BiCGStab: y = A x with the dot products against y: MODE 0: rhat.y;  MODE 1: y.s, y.y
template <int MODE>
__global__ __launch_bounds__(FT) void bi_spmv_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                     const double* __restrict__ v, const double* __restrict__ x,
                                                     double* __restrict__ y, const double* __restrict__ other, Red red) {
  if (red.sc[S_DONE] != 0.0) return;
  const int lane = threadIdx.x % LPR;
  double d[2] = {0.0, 0.0};
  for (int row = (blockIdx.x * FT + threadIdx.x) / LPR; row < n; row += gridDim.x * (FT / LPR)) {
    const double acc = row_sum<LPR>(rp, ci, v, row, lane, [&](int c) { return x[c]; });
    if (lane == 0) { y[row] = acc; d[0] += acc * other[row]; d[1] += acc * acc; }
  }
  if (MODE == 0) {
    double one[1] = {d[0]};
    double* const dst[1] = {red.arr(P_RV)};
    block_partials<1>(one, dst);
  } else {
    double* const dst[2] = {red.arr(P_TS), red.arr(P_TT)};
    block_partials<2>(d, dst);
  }
}

// BiCGStab: omega = t.s / t.t, x += alpha phat + omega shat, r = s - omega t; rhat.r, r.r
__global__ __launch_bounds__(FT) void bi_update_kernel(int n, double* __restrict__ x, double* __restrict__ r,
                                                       const double* __restrict__ phat, const double* __restrict__ shat,
                                                       const double* __restrict__ s, const double* __restrict__ t,
                                                       const double* __restrict__ rhat, Red red, int np) {
  double* sc = red.sc;
  if (sc[S_DONE] != 0.0) return;
  const double* const src[2] = {red.arr(P_TS), red.arr(P_TT)};
  double ts[2];
  totals<2>(src, np, ts);
  const double om = ts[1] != 0.0 ? ts[0] / ts[1] : 0.0;      // t = 0 <=> s = 0: the half step was exact, r = s
  const double alpha = sc[S_ALPHA];
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc[S_TS] = ts[0]; sc[S_TT] = ts[1]; sc[S_OMEGA] = om; }
  double d[2] = {0.0, 0.0};
  for (int i = blockIdx.x * FT + threadIdx.x; i < n; i += gridDim.x * FT) {
    x[i] += alpha * phat[i] + om * shat[i];
    const double rn = s[i] - om * t[i];
    r[i] = rn;
    d[0] += rhat[i] * rn;
    d[1] += rn * rn;
  }
  double* const dst[2] = {red.arr(P_RHR), red.arr(P_RR)};
  block_partials<2>(d, dst);
}

// BiCGStab, end of a chunk: the convergence test the next iteration's first kernel would make (one block)
__global__ __launch_bounds__(FT) void bi_check_kernel(Red red, int np, int k) {
  double* sc = red.sc;
  if (sc[S_DONE] != 0.0) return;
  const double* const src[1] = {red.arr(P_RR)};
  double rr[1];
  totals<1>(src, np, rr);
  if (threadIdx.x == 0) {
    sc[S_RR] = rr[0]; sc[S_IT] = (double)k;
    if (!(rr[0] > sc[S_TARGET2])) sc[S_DONE] = DONE_CONVERGED;
  }
}

// r = b - A x (and rhat = r for BiCGStab), |r|^2 and |b|^2 as block partial sums
__global__ __launch_bounds__(FT) void residual_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      const double* __restrict__ v, const double* __restrict__ x,
                                                      const double* __restrict__ b, double* __restrict__ r,
                                                      double* __restrict__ rhat, Red red) {
  const int lane = threadIdx.x % LPR;
  double d[2] = {0.0, 0.0};
  for (int row = (blockIdx.x * FT + threadIdx.x) / LPR; row < n; row += gridDim.x * (FT / LPR)) {
    const double acc = row_sum<LPR>(rp, ci, v, row, lane, [&](int c) { return x[c]; });
    if (lane == 0) {
      const double bi = b[row], ri = bi - acc;
      r[row] = ri;
      if (rhat) rhat[row] = ri;
      d[0] += ri * ri; d[1] += bi * bi;
    }
  }
  double* const dst[2] = {red.arr(P_RR), red.arr(P_PQ)};
  block_partials<2>(d, dst);
}

// Loop state at the start of a solve (one block): |r0|^2, |b|^2 from the residual kernel's partial sums, the target
// max(atol, rtol |b|) -- the host does not have to read anything before the first chunk.
__global__ __launch_bounds__(FT) void start_kernel(Red red, int np, double rtol, double atol, int bicg) {
  double* sc = red.sc;
  const double* const src[2] = {red.arr(P_RR), red.arr(P_PQ)};
  double d[2];
  totals<2>(src, np, d);
  if (threadIdx.x != 0) return;
  const double rr = d[0], bb = d[1], bnorm = sqrt(bb);
  const double target = fmax(atol, rtol * (bnorm > 0.0 ? bnorm : 1.0));
  sc[S_RR] = rr; sc[S_BB] = bb;
  sc[S_TARGET2] = target * target;
  sc[S_IT] = 0.0; sc[S_FLAG] = 0.0;
  sc[S_RHO0] = 0.0; sc[S_RHO1] = 0.0; sc[S_BETA] = 0.0; sc[S_ALPHA] = bicg ? 1.0 : 0.0; sc[S_OMEGA] = 1.0;
  // non-finite data is an input error, not a breakdown: no kernel of the loop touches the iterate
  sc[S_DONE] = !(bb - bb == 0.0) ? DONE_BAD_RHS : (!(rr - rr == 0.0) ? DONE_BAD_START : (rr > target * target ? DONE_NO : DONE_CONVERGED));
  if (bicg) { red.arr(P_RHR)[0] = rr; red.arr(P_RR)[0] = rr; }      // rhat = r: rho_0 = r.r, one "partial sum" each
}

// BiCGStab restart after a breakdown: rhat = r (rho = r.r: the r.r partial sums become the rhat.r ones), p = r next
__global__ void bi_restart_kernel(int n, double* __restrict__ rhat, const double* __restrict__ r, Red red, int np) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rhat[i] = r[i];
  if (i < np) red.arr(P_RHR)[i] = red.arr(P_RR)[i];
  if (i == 0) { red.sc[S_RHO0] = 0.0; red.sc[S_RHO1] = 0.0; red.sc[S_FLAG] = 0.0; }
}

struct Loop {
  knpemi_handle* h;
  KnAmg& G;
  const KnFusedSys& S;
  Red red;
  hipStream_t st;

  static int blocks16(int rows) { return (int)(((size_t)rows * LPR + FT - 1) / FT); }
  static int capped(int b) { return std::max(1, std::min(KN_PB, b)); }
  int down_blocks(const KnAmgLevel& L) const { return (L.n + FT / LPR - 1) / (FT / LPR) + (L.nc + FT / 64 - 1) / (FT / 64); }

  // out = V-cycle(in) on the frozen hierarchy.  IN: how the finest level forms its input (down_kernel); np: partial sums
  // its scalar needs; k: iteration.  CHECK: CG's convergence test in the coarsest kernel.  Returns the number of partial
  // sums the finest kernels leave (IN_CG: r.r by the first, DOTS: r.z by the last).
  template <int IN, bool DOTS, bool CHECK>
  void cycle(const double* r, const double* u, const double* w, double* formed, double* x, double* out, int np, int k,
             int* np_rr, int* np_rz) {
    const int nl = (int)G.lev.size();
    for (int l = 0; l + 1 < nl; ++l) {
      KnAmgLevel& L = G.lev[l];
      DownArgs a{};
      a.n = L.n; a.nc = L.nc;
      a.arp = L.A.rp; a.aci = L.A.ci; a.av = l == 0 ? L.frozen_v : L.A.v;
      a.rrp = L.Rm.rp; a.rci = L.Rm.ci; a.rv = L.Rm.v;
      a.dinv = L.dinv; a.omega = L.omega;
      a.t = L.t; a.rc = G.lev[l + 1].r;
      a.red = red; a.np = np; a.k = k;
      if (l == 0) {
        a.r = r; a.u = u; a.w = w; a.out = formed; a.x = x;
        const int nb = IN == IN_CG ? capped(down_blocks(L)) : down_blocks(L);
        if (np_rr) *np_rr = nb;
        static const bool split = getenv("KNPEMI_FUSED_NO_SPLIT") == nullptr;
        if ((IN == IN_BI_P || IN == IN_BI_S) && split) {
          hipLaunchKernelGGL((form_kernel<IN>), dim3(capped((L.n + FT - 1) / FT)), dim3(FT), 0, st, a);
          a.r = formed; a.out = nullptr;
          hipLaunchKernelGGL((down_kernel<IN_PLAIN>), dim3(nb), dim3(FT), 0, st, a);
        } else {
          hipLaunchKernelGGL((down_kernel<IN>), dim3(nb), dim3(FT), 0, st, a);
        }
      } else {
        a.r = L.r;
        hipLaunchKernelGGL((down_kernel<IN_PLAIN>), dim3(down_blocks(L)), dim3(FT), 0, st, a);
      }
    }
    KnAmgLevel& C = G.lev[nl - 1];
    hipLaunchKernelGGL((dense_kernel<CHECK>), dim3(((size_t)C.n * 64 + FT - 1) / FT), dim3(FT), 0, st, C.n, C.dense_inv, C.r, C.x,
                       red, np_rr ? *np_rr : 0, k);
    for (int l = nl - 2; l >= 0; --l) {
      KnAmgLevel& L = G.lev[l];
      UpArgs a{};
      a.n = L.n;
      a.prp = L.Pm.rp; a.pci = L.Pm.ci; a.pv = L.Pm.v;
      a.dinv = L.dinv; a.omega = L.omega;
      a.t = L.t; a.ec = G.lev[l + 1].x;
      a.red = red;
      if (l == 0) {
        a.r = IN == IN_PLAIN ? r : formed; a.x = out;
        const int nb = DOTS ? capped(blocks16(L.n)) : blocks16(L.n);
        if (np_rz) *np_rz = nb;
        hipLaunchKernelGGL((up_kernel<DOTS>), dim3(nb), dim3(FT), 0, st, a);
      } else {
        a.r = L.r; a.x = L.x;
        hipLaunchKernelGGL((up_kernel<false>), dim3(blocks16(L.n)), dim3(FT), 0, st, a);
      }
    }
  }
};

}  // namespace

// Levels l0 .. coarsest of the cycle with the merged transfer operators (two launches per level instead of four), for a
// hierarchy whose FINEST level is handled by the caller (the block-smoothed DG systems, kn_amg_apply): input in
// G.lev[l0].r, result in G.lev[l0].x.  The kernels are those of the fused loops; their early-out flag reads a zeroed
// scalar block (G.zero_sc), no dot products are involved.  l0 = 0 (the rank-local cycle of a partitioned problem): input
// r0, result out0, the finest operator is the frozen one the hierarchy was built from.
int kn_fused_subcycle(knpemi_handle* h, KnAmg& G, int l0, const double* r0, double* out0) {
  hipStream_t st = h->stream;
  const Red red{G.zero_sc, nullptr};
  const int nl = (int)G.lev.size();
  for (int l = l0; l + 1 < nl; ++l) {
    KnAmgLevel& L = G.lev[l];
    DownArgs a{};
    a.n = L.n; a.nc = L.nc;
    a.arp = L.A.rp; a.aci = L.A.ci; a.av = l == 0 ? L.frozen_v : L.A.v;
    a.rrp = L.Rm.rp; a.rci = L.Rm.ci; a.rv = L.Rm.v;
    a.dinv = L.dinv; a.omega = L.omega;
    a.r = (l == l0 && r0) ? r0 : L.r; a.t = L.t; a.rc = G.lev[l + 1].r;
    a.red = red;
    const int nb = (L.n + FT / LPR - 1) / (FT / LPR) + (L.nc + FT / 64 - 1) / (FT / 64);
    hipLaunchKernelGGL((down_kernel<IN_PLAIN>), dim3(nb), dim3(FT), 0, st, a);
  }
  KnAmgLevel& C = G.lev[nl - 1];
  hipLaunchKernelGGL((dense_kernel<false>), dim3(((size_t)C.n * 64 + FT - 1) / FT), dim3(FT), 0, st, C.n, C.dense_inv, C.r, C.x, red, 0, 0);
  for (int l = nl - 2; l >= l0; --l) {
    KnAmgLevel& L = G.lev[l];
    UpArgs a{};
    a.n = L.n;
    a.prp = L.Pm.rp; a.pci = L.Pm.ci; a.pv = L.Pm.v;
    a.dinv = L.dinv; a.omega = L.omega;
    a.r = (l == l0 && r0) ? r0 : L.r; a.t = L.t; a.ec = G.lev[l + 1].x; a.x = (l == l0 && out0) ? out0 : L.x;
    a.red = red;
    hipLaunchKernelGGL((up_kernel<false>), dim3((int)(((size_t)L.n * LPR + FT - 1) / FT)), dim3(FT), 0, st, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("fused sub-cycle: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  return KNPEMI_OK;
}

namespace {

// scalars + partial-sum arrays of the fused loops
int ensure_partials(knpemi_handle* h) {
  const size_t need = (size_t)P_N * KN_PB;
  if (h->fused_part_n >= need) return KNPEMI_OK;
  void* p = nullptr;
  KN_HIP(hipMalloc(&p, need * sizeof(double)));
  h->allocs.push_back(p);
  KN_HIP(hipMemsetAsync(p, 0, need * sizeof(double), h->stream));
  h->fused_part = static_cast<double*>(p);
  h->fused_part_n = need;
  return KNPEMI_OK;
}

int read_state(knpemi_handle* h, const double* sc_dev, double* host) {
  if (!h->kry_pinned) KN_HIP(hipHostMalloc(&h->kry_pinned, 64 * sizeof(double), hipHostMallocDefault));
  KN_HIP(hipMemcpyAsync(h->kry_pinned, sc_dev, S_NF * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < S_NF; ++i) host[i] = static_cast<double*>(h->kry_pinned)[i];
  return KNPEMI_OK;
}

std::string describe(const double* sc) {
  char buf[320];
  snprintf(buf, sizeof buf,
           "[iteration %d: |r|^2 = %.6e, |b|^2 = %.6e, rho = %.6e / %.6e, alpha = %.6e, beta = %.6e, omega = %.6e, rhat.v = %.6e, "
           "t.s = %.6e, t.t = %.6e, p.Ap = %.6e, breakdown flags = %d]",
           (int)sc[S_IT], sc[S_RR], sc[S_BB], sc[S_RHO0], sc[S_RHO1], sc[S_ALPHA], sc[S_BETA], sc[S_OMEGA], sc[S_RV], sc[S_TS],
           sc[S_TT], sc[S_PAP], (int)sc[S_FLAG]);
  return buf;
}

int bad_input(const char* who, const double* sc) {
  if ((int)sc[S_DONE] == DONE_BAD_RHS) { kn_set_error(std::string(who) + ": the right-hand side contains non-finite values"); return KNPEMI_EINVAL; }
  if ((int)sc[S_DONE] == DONE_BAD_START) {
    kn_set_error(std::string(who) + ": the initial residual b - A x0 is not finite (matrix values or initial guess)");
    return KNPEMI_EINVAL;
  }
  return KNPEMI_OK;
}

// Iterations enqueued before the host looks at the device state: as many as the previous solve of this system took
// (the counts change slowly along a trajectory), then one at a time (a further read costs ~25 us, an enqueued iteration
// that turns out not to be needed ~35 us of kernels that return at once).
int first_chunk(int last_its, int maxit) {
  static const int forced = getenv("KNPEMI_FUSED_CHUNK") ? std::max(1, atoi(getenv("KNPEMI_FUSED_CHUNK"))) : 0;
  const int c = forced ? forced : std::min(16, std::max(1, last_its));
  return std::max(1, std::min(c, maxit));
}

}  // namespace

// CG on S x = b from the x in the workspace; workspace vectors as in kn_solve_emi.
int kn_fused_cg(knpemi_handle* h, KnAmg& G, const KnFusedSys& S, const double* b, double rtol, double atol, int maxit, int* iters,
                double* rr_out, double* bb_out) {
  const int n = S.n;
  const size_t N = S.N;
  double *x = S.work, *r = x + N, *z = r + N, *p = z + N, *q = p + N, *r2 = S.work + 7 * N, *p2 = S.work + 8 * N;
  int rc = ensure_partials(h);
  if (rc) return rc;
  Loop L{h, G, S, Red{S.sc, h->fused_part}, h->stream};
  const int nb_res = Loop::capped(Loop::blocks16(n));
  hipLaunchKernelGGL(residual_kernel, dim3(nb_res), dim3(FT), 0, h->stream, n, S.rowptr, S.colind, S.vals, x, b, r,
                     (double*)nullptr, L.red);
  hipLaunchKernelGGL(start_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, nb_res, rtol, atol, 0);
  // z_0 = M^-1 r_0, r_0.z_0 (rho_old = 0: the first direction is z_0)
  int np_rr = 0, np_rz = 0;
  L.cycle<IN_PLAIN, true, false>(r, nullptr, nullptr, nullptr, nullptr, z, 0, 0, nullptr, &np_rz);
  const int nb_dir = Loop::capped(Loop::blocks16(n));
  double sc[S_NF];
  int it = 0, k = 0, todo = first_chunk(G.its_last, maxit);
  for (;;) {
    for (int j = 0; j < todo; ++j, ++k) {
      hipLaunchKernelGGL(cg_dir_kernel, dim3(nb_dir), dim3(FT), 0, h->stream, n, S.rowptr, S.colind, S.vals, z, p, p2, q, L.red,
                         np_rz, k);
      // r_new = r - alpha q (stored in r2), x += alpha p_new, |r_new|^2 -> convergence; then z = M^-1 r_new, r_new.z
      L.cycle<IN_CG, true, true>(r, q, p2, r2, x, z, nb_dir, k, &np_rr, &np_rz);
      std::swap(r, r2);
      std::swap(p, p2);
    }
    if ((rc = read_state(h, S.sc, sc))) return rc;
    if ((rc = bad_input("EMI CG", sc))) return rc;
    it = (int)sc[S_IT];
    if (!std::isfinite(sc[S_RR])) { kn_set_error("EMI CG broke down (non-finite residual) " + describe(sc)); return KNPEMI_ESOLVE; }
    if (sc[S_DONE] != 0.0 || it >= maxit) break;
    todo = 1;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("fused CG: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  G.its_last = it;
  *iters = it;
  *rr_out = sc[S_RR];
  *bb_out = sc[S_BB];
  return KNPEMI_OK;
}

// Right-preconditioned BiCGStab on S x = b from the x in the workspace; vectors as in kn_solve_knp.
int kn_fused_bicgstab(knpemi_handle* h, KnAmg& G, const KnFusedSys& S, const double* b, double rtol, double atol, int maxit,
                      int* iters, double* rr_out, double* bb_out) {
  const int n = S.n;
  const size_t N = S.N;
  double *x = S.work, *r = x + N, *rhat = r + N, *p = rhat + N, *v = p + N, *s = v + N, *t = s + N;
  double *phat = S.work + 8 * N, *shat = S.work + 9 * N;
  int rc = ensure_partials(h);
  if (rc) return rc;
  Loop L{h, G, S, Red{S.sc, h->fused_part}, h->stream};
  const int nb_res = Loop::capped(Loop::blocks16(n));
  hipLaunchKernelGGL(residual_kernel, dim3(nb_res), dim3(FT), 0, h->stream, n, S.rowptr, S.colind, S.vals, x, b, r, rhat, L.red);
  hipLaunchKernelGGL(start_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, nb_res, rtol, atol, 1);
  const int nb_spmv = Loop::capped(Loop::blocks16(n)), nb_update = Loop::capped((n + FT - 1) / FT);
  double sc[S_NF];
  int it = 0, k = 0, restarts = 0, np_upd = 1, todo = first_chunk(G.its_last, maxit);
  for (;;) {
    for (int j = 0; j < todo; ++j, ++k) {
      // p = r + beta (p - omega v) is formed by the finest down kernel, which reads p and v at neighbouring rows while
      // it stores its own: the new direction goes to the buffer of t (free until the second SpMV) and the two names swap
      L.cycle<IN_BI_P, false, false>(r, p, v, t, nullptr, phat, np_upd, k, nullptr, nullptr);
      std::swap(p, t);
      hipLaunchKernelGGL((bi_spmv_kernel<0>), dim3(nb_spmv), dim3(FT), 0, h->stream, n, S.rowptr, S.colind, S.vals, phat, v, rhat,
                         L.red);
      L.cycle<IN_BI_S, false, false>(r, v, nullptr, s, nullptr, shat, nb_spmv, k, nullptr, nullptr);
      hipLaunchKernelGGL((bi_spmv_kernel<1>), dim3(nb_spmv), dim3(FT), 0, h->stream, n, S.rowptr, S.colind, S.vals, shat, t, s,
                         L.red);
      hipLaunchKernelGGL(bi_update_kernel, dim3(nb_update), dim3(FT), 0, h->stream, n, x, r, phat, shat, s, t, rhat, L.red, nb_spmv);
      np_upd = nb_update;
    }
    hipLaunchKernelGGL(bi_check_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, np_upd, k);
    if ((rc = read_state(h, S.sc, sc))) return rc;
    if ((rc = bad_input("KNP BiCGStab", sc))) return rc;
    it = (int)sc[S_IT];
    if (!std::isfinite(sc[S_RR])) {
      kn_set_error("KNP BiCGStab broke down (non-finite residual) " + describe(sc));
      return KNPEMI_ESOLVE;
    }
    if (sc[S_DONE] != 0.0 || it >= maxit) break;
    if ((int)sc[S_FLAG] & (F_RHO_ZERO | F_OMEGA_ZERO | F_RV_ZERO)) {
      // a true breakdown with a residual left: restart from the current iterate with rhat = r, as PETSc's KSPBCGS does
      if (++restarts > 3) { kn_set_error("KNP BiCGStab broke down repeatedly " + describe(sc)); return KNPEMI_ESOLVE; }
      hipLaunchKernelGGL(bi_restart_kernel, dim3((std::max(n, KN_PB) + 255) / 256), dim3(256), 0, h->stream, n, rhat, r, L.red, np_upd);
    }
    todo = 1;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("fused BiCGStab: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  G.its_last = it;
  *iters = it;
  *rr_out = sc[S_RR];
  *bb_out = sc[S_BB];
  return KNPEMI_OK;
}
