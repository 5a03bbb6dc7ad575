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
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "knpemi_internal.h"

namespace {

// device scalars (the first 13 in the layout of kernels_krylov.hip)
enum { S_RHO0 = 0, S_RHO1, S_PAP, S_ALPHA, S_BETA, S_OMEGA, S_RR, S_BB, S_TS, S_TT, S_MEAN, S_FLAG, S_RV,
       S_DONE, S_IT, S_TARGET2, S_MINIT, S_NF };     // S_MINIT: fewest iterations before convergence counts (ksp_min_it)
enum { F_RHO_ZERO = 1, F_RV_ZERO = 2, F_OMEGA_ZERO = 4, F_PAP_ZERO = 8 };
enum { DONE_NO = 0, DONE_CONVERGED = 1, DONE_BAD_RHS = 2, DONE_BAD_START = 3 };
constexpr int FT = 256, LPR = 16;
// Most blocks of a kernel that produces a dot product (they walk the rows with a grid stride).  Every block of the CONSUMER
// adds these partial sums up again at its head: tools/probes/kernel_head.hip prices that at +2.6 us per kernel for 1 651
// partial sums read one after the other (8 dependent loads per thread), +1.9 with the 8 loads in flight together and +0.7 for
// 256 (one load): at most 1 024 partial sums, all of a thread's (<= 4) loads issued before the first is waited for.
constexpr int KN_PB = 1024;
// partial-sum arrays (KN_PB doubles each)
enum { P_PQ = 0, P_RR, P_RZ, P_RV, P_TS, P_TT, P_RHR, P_BS, P_XS, P_ZZ, P_N };     // P_BS: sum of b_emi, P_XS: sum of x (means)

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

// the same with the row's range [a, b) already in registers (requested at the top of the kernel, before the early-out flag
// and the producer's partial sums have arrived: one dependent memory round trip fewer per kernel)
template <int L, class F>
__device__ __forceinline__ double row_sum_ab(const int* __restrict__ ci, const double* __restrict__ v, int a, int b, int lane, F&& f) {
  double acc = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  int j = a + lane;
  for (; j + 3 * L < b; j += 4 * L) {
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

// The producer's partial sums of ND dot products: load() issues every load of the thread (KN_PB / FT per array, predicated)
// without waiting -- the caller reads its scalars and the early-out flag while they are in flight --, finish() adds them
// up in a fixed order (the same bits in every block and on every run).
template <int ND>
struct Totals {
  static constexpr int PER = KN_PB / FT;
  double v[ND][PER];
  __device__ __forceinline__ void load(const double* const (&src)[ND], int np) {
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < ND; ++k)
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int i = t + j * FT;
        v[k][j] = i < np ? src[k][i] : 0.0;
      }
  }
  __device__ __forceinline__ void finish(double (&out)[ND]) {
#pragma unroll
    for (int k = 0; k < ND; ++k) {
      double s = v[k][0];
#pragma unroll
      for (int j = 1; j < PER; ++j) s += v[k][j];
      out[k] = s;
    }
    block_sum<ND>(out);
  }
};

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
  int par;                                     // parity of the iteration (which of the two rho slots is the current one)
};

// The scalars the finest `down` kernel needs before it can form its input (every block computes them from the producer's
// partial sums, block 0 records them); false: the solve has converged (or the early-out flag was up), nothing is left to
// do.  The partial sums, the flag and the scalars are requested together: one memory round trip at the head of the kernel.
// The iteration count lives on the device (S_IT, advanced by the kernel that ends an iteration), so a captured graph of
// iterations can be replayed whatever the number of iterations before it.
template <int IN>
__device__ __forceinline__ bool down_head(const DownArgs& a, double& alpha, double& beta, double& omb) {
  double* sc = a.red.sc;
  if constexpr (IN == IN_PLAIN) {
    return sc[S_DONE] == 0.0;
  } else if constexpr (IN == IN_CG) {       // alpha = rho / p.Ap
    const double* const src[1] = {a.red.arr(P_PQ)};
    Totals<1> T;
    T.load(src, a.np);
    const double done = sc[S_DONE], rho = sc[S_RHO0 + a.par];
    if (done != 0.0) return false;
    double pq[1];
    T.finish(pq);
    alpha = pq[0] != 0.0 ? rho / pq[0] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc[S_PAP] = pq[0]; sc[S_ALPHA] = alpha;
      if (pq[0] == 0.0) raise_flag(sc, F_PAP_ZERO);
    }
    return true;
  } else if constexpr (IN == IN_BI_P) {     // convergence of the previous iteration, rho, beta
    const double* const src[2] = {a.red.arr(P_RHR), a.red.arr(P_RR)};
    Totals<2> T;
    T.load(src, a.np);
    const double done0 = sc[S_DONE], target2 = sc[S_TARGET2], rho_old = sc[S_RHO0 + (a.par ^ 1)], al = sc[S_ALPHA];
    const double its = sc[S_IT], min_it = sc[S_MINIT];
    omb = sc[S_OMEGA];
    if (done0 != 0.0) return false;
    double d[2];
    T.finish(d);
    // ksp_min_it: a residual below the target does not end the solve before min_it iterations (a vanished one does)
    const bool done = !(d[1] > target2) && (its >= min_it || d[1] == 0.0);
    // rho_old = 0: the first direction (or a restart), p = r.  omega = 0 with a residual left is a breakdown: beta = 0
    // restarts the recurrence from p = r as well (the host re-bases rhat when it sees the flag)
    const bool first = rho_old == 0.0;
    beta = (!first && omb != 0.0) ? (d[0] / rho_old) * (al / omb) : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc[S_RR] = d[1];
      if (done) sc[S_DONE] = DONE_CONVERGED;
      else {
        sc[S_RHO0 + a.par] = d[0]; sc[S_BETA] = beta;
        if (!first && omb == 0.0) raise_flag(sc, F_OMEGA_ZERO);
        if (d[0] == 0.0) raise_flag(sc, F_RHO_ZERO);
      }
    }
    return !done;
  } else {                                  // IN_BI_S: alpha = rho / rhat.v
    const double* const src[1] = {a.red.arr(P_RV)};
    Totals<1> T;
    T.load(src, a.np);
    const double done = sc[S_DONE], rho = sc[S_RHO0 + a.par];
    if (done != 0.0) return false;
    double rv[1];
    T.finish(rv);
    alpha = rv[0] != 0.0 ? rho / rv[0] : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc[S_RV] = rv[0]; sc[S_ALPHA] = alpha;
      if (rv[0] == 0.0) raise_flag(sc, F_RV_ZERO);
    }
    return true;
  }
}

// Rows of A with LA lanes each (4 where the rows are short -- a quarter of the waves to dispatch --, else 16), rows of the
// merged restriction (hundreds of entries) with a wavefront each.  The range of the block's first row is requested before
// the head of the kernel waits for anything.
template <int IN, int LA>
__global__ __launch_bounds__(FT) void down_kernel(DownArgs a) {
  const int nra = (a.n + FT / LA - 1) / (FT / LA);        // block-passes over A
  const int nrr = (a.nc + FT / 64 - 1) / (FT / 64);       // block-passes over Rm
  auto range = [&](int pass, int& ra, int& rb) {
    ra = rb = 0;
    if (pass < nra) {
      const int row = pass * (FT / LA) + threadIdx.x / LA;
      if (row < a.n) { ra = a.arp[row]; rb = a.arp[row + 1]; }
    } else if (pass < nra + nrr) {
      const int k = (pass - nra) * (FT / 64) + threadIdx.x / 64;
      if (k < a.nc) { ra = a.rrp[k]; rb = a.rrp[k + 1]; }
    }
  };
  int pass = blockIdx.x, ra, rb;
  range(pass, ra, rb);
  double alpha = 0.0, beta = 0.0, omb = 0.0;
  if (!down_head<IN>(a, alpha, beta, omb)) return;
  auto in = [&](int j) -> double {
    if constexpr (IN == IN_PLAIN) return a.r[j];
    else if constexpr (IN == IN_CG) return a.r[j] - alpha * a.u[j];
    else if constexpr (IN == IN_BI_P) return beta != 0.0 ? a.r[j] + beta * (a.u[j] - omb * a.w[j]) : a.r[j];
    else return a.r[j] - alpha * a.u[j];
  };
  double rr[1] = {0.0};
  for (; pass < nra + nrr;) {
    if (pass < nra) {
      const int row = pass * (FT / LA) + threadIdx.x / LA, lane = threadIdx.x % LA;
      if (row < a.n) {
        const double mine = in(row);
        const double acc = row_sum_ab<LA>(a.aci, a.av, ra, rb, lane, [&](int c) { return a.dinv[c] * in(c); });
        if (lane == 0) {
          a.t[row] = mine - a.omega * acc;
          if constexpr (IN != IN_PLAIN) a.out[row] = mine;
          if constexpr (IN == IN_CG) { a.x[row] += alpha * a.w[row]; rr[0] += mine * mine; }
        }
      }
    } else {
      const int k = (pass - nra) * (FT / 64) + threadIdx.x / 64, lane = threadIdx.x % 64;
      if (k < a.nc) {
        const double acc = row_sum_ab<64>(a.rci, a.rv, ra, rb, lane, in);
        if (lane == 0) a.rc[k] = acc;
      }
    }
    pass += gridDim.x;
    if (pass < nra + nrr) range(pass, ra, rb);
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
  // the thread's first operands are requested before the head waits for the scalars
  int j = blockIdx.x * FT + threadIdx.x;
  double rj = 0.0, uj = 0.0, wj = 0.0;
  if (j < a.n) { rj = a.r[j]; uj = a.u[j]; if constexpr (IN == IN_BI_P) wj = a.w[j]; }
  double alpha = 0.0, beta = 0.0, omb = 0.0;
  if (!down_head<IN>(a, alpha, beta, omb)) return;
  while (j < a.n) {
    if constexpr (IN == IN_BI_P) a.out[j] = beta != 0.0 ? rj + beta * (uj - omb * wj) : rj;
    else a.out[j] = rj - alpha * uj;
    j += gridDim.x * FT;
    if (j < a.n) { rj = a.r[j]; uj = a.u[j]; if constexpr (IN == IN_BI_P) wj = a.w[j]; }
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

// DOTS (the finest level of the cycle in CG): bit 0 r.z, bit 1 also z.z (the preconditioned residual norm KSPCG tests by
// default), bit 2 the iteration ends here (count)
template <int DOTS>
__global__ __launch_bounds__(FT) void up_kernel(UpArgs a) {
  const int lane = threadIdx.x % LPR, stride = gridDim.x * (FT / LPR);
  int row = (blockIdx.x * FT + threadIdx.x) / LPR, ra = 0, rb = 0;
  double rrow = 0.0, trow = 0.0, drow = 0.0;
  auto fetch = [&]() { if (row < a.n) { ra = a.prp[row]; rb = a.prp[row + 1]; rrow = a.r[row]; trow = a.t[row]; drow = a.dinv[row]; } };
  fetch();
  if (a.red.sc[S_DONE] != 0.0) return;
  double rz[2] = {0.0, 0.0};
  double it0 = 0.0;
  if constexpr ((DOTS & 4) != 0) it0 = a.red.sc[S_IT];
  while (row < a.n) {
    const double acc = row_sum_ab<LPR>(a.pci, a.pv, ra, rb, lane, [&](int c) { return a.ec[c]; });
    if (lane == 0) {
      const double z = a.omega * drow * (rrow + trow) + acc;
      a.x[row] = z;
      if constexpr ((DOTS & 1) != 0) rz[0] += rrow * z;
      if constexpr ((DOTS & 2) != 0) rz[1] += z * z;
    }
    row += stride;
    fetch();
  }
  if constexpr ((DOTS & 2) != 0) {
    double* const dst[2] = {a.red.arr(P_RZ), a.red.arr(P_ZZ)};
    block_partials<2>(rz, dst);
  } else if constexpr ((DOTS & 1) != 0) {
    double one[1] = {rz[0]};
    double* const dst[1] = {a.red.arr(P_RZ)};
    block_partials<1>(one, dst);
  }
  if constexpr ((DOTS & 4) != 0) { if (blockIdx.x == 0 && threadIdx.x == 0) a.red.sc[S_IT] = it0 + 1.0; }
}

// Coarsest level: e = Minv r, one wavefront per row; the inverse is stored as up to 8 dense diagonal blocks (the K - 1
// ion systems of the concentration matrix are independent, and so is every coarse operator of their hierarchy: a third of
// the values of the full n x n array at config 2), the blocks are contiguous ranges of unknowns (kernels_amg.hip numbers the
// aggregates component by component), so a row's values and its part of r are addressed from the kernel arguments alone and
// the first of them are requested before the head waits.  CHECK (CG): the convergence test of the iteration whose finest
// down kernel left |r|^2 -- the rest of the cycle and of the chunk returns at once when it is met -- and the iteration count.
template <bool CHECK>
__global__ __launch_bounds__(FT) void dense_kernel(int n, const double* __restrict__ Minv, KnDenseBlocks B,
                                                   const double* __restrict__ r, double* __restrict__ x, Red red, int np) {
  double* sc = red.sc;
  const int row = (blockIdx.x * FT + threadIdx.x) >> 6, l = threadIdx.x & 63;
  const double* m = Minv;
  const double* rb = r;
  int mc = 0;
  if (row < n) {
    int b = 0;
    for (int k = 1; k < B.nb; ++k) b = row >= B.start[k] ? k : b;
    m = Minv + (size_t)B.off[b] + (size_t)(row - B.start[b]) * B.size[b];
    rb = r + B.start[b];
    mc = B.size[b];
  }
  // first four products' operands in flight before anything is waited for
  double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0, r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
  if (l < mc) { m0 = m[l]; r0 = rb[l]; }
  if (l + 64 < mc) { m1 = m[l + 64]; r1 = rb[l + 64]; }
  if (l + 128 < mc) { m2 = m[l + 128]; r2 = rb[l + 128]; }
  if (l + 192 < mc) { m3 = m[l + 192]; r3 = rb[l + 192]; }
  if constexpr (CHECK) {
    const double* const src[1] = {red.arr(P_RR)};
    Totals<1> T;
    T.load(src, np);
    const double done0 = sc[S_DONE], target2 = sc[S_TARGET2], it = sc[S_IT];
    if (done0 != 0.0) return;
    double rr[1];
    T.finish(rr);
    const bool done = !(rr[0] > target2);      // also stops on a NaN: the host reports it
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc[S_RR] = rr[0]; sc[S_IT] = it + 1.0;
      if (done) sc[S_DONE] = DONE_CONVERGED;
    }
    if (done) return;
  } else {
    if (sc[S_DONE] != 0.0) return;
  }
  double acc = (m0 * r0 + m1 * r1) + (m2 * r2 + m3 * r3), acc1 = 0.0;
  int j = l + 256;
  for (; j + 64 < mc; j += 128) { acc += m[j] * rb[j]; acc1 += m[j + 64] * rb[j + 64]; }
  if (j < mc) acc += m[j] * rb[j];
  acc += acc1;
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) acc += __shfl_xor(acc, k);
  if (row < n && l == 0) x[row] = acc;
}

// CG: beta = rho / rho_old, p_new = z + beta p (stored), q = A p_new, p.q
// PRE: convergence on |z| = |M^-1 r| (the producer of z left z.z beside r.z), tested here for the iterate the previous
// iteration produced -- at the first iteration that is the initial residual, as KSPCG tests it
template <int L0, bool PRE>
__global__ __launch_bounds__(FT) void cg_dir_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                    const double* __restrict__ v, const double* __restrict__ z,
                                                    const double* __restrict__ p, double* __restrict__ pn, double* __restrict__ q,
                                                    Red red, int np, int par) {
  double* sc = red.sc;
  const int lane = threadIdx.x % L0, stride = gridDim.x * (FT / L0);
  int row = (blockIdx.x * FT + threadIdx.x) / L0, ra = 0, rb = 0;
  double zr = 0.0, pr = 0.0;
  auto fetch = [&]() { if (row < n) { ra = rp[row]; rb = rp[row + 1]; zr = z[row]; pr = p[row]; } };
  fetch();
  double rho[1];
  double done = 0.0, rho_old = 0.0;
  if constexpr (PRE) {
    const double* const src[2] = {red.arr(P_RZ), red.arr(P_ZZ)};
    Totals<2> T;
    T.load(src, np);
    done = sc[S_DONE]; rho_old = sc[S_RHO0 + (par ^ 1)];
    const double target2 = sc[S_TARGET2];
    if (done != 0.0) return;
    double d[2];
    T.finish(d);
    rho[0] = d[0];
    const bool conv = !(d[1] > target2);       // also stops on a NaN: the host reports it
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[S_RR] = d[1]; if (conv) sc[S_DONE] = DONE_CONVERGED; }
    if (conv) return;
  } else {
    const double* const src[1] = {red.arr(P_RZ)};
    Totals<1> T;
    T.load(src, np);
    done = sc[S_DONE]; rho_old = sc[S_RHO0 + (par ^ 1)];
    if (done != 0.0) return;
    T.finish(rho);
  }
  const double beta = rho_old != 0.0 ? rho[0] / rho_old : 0.0;     // the first direction is z (whatever the buffer of p holds)
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc[S_RHO0 + par] = rho[0]; sc[S_BETA] = beta; }
  double pq[1] = {0.0};
  while (row < n) {
    const double mine = beta != 0.0 ? zr + beta * pr : zr;
    const double acc = row_sum_ab<L0>(ci, v, ra, rb, lane, [&](int c) { return beta != 0.0 ? z[c] + beta * p[c] : z[c]; });
    if (lane == 0) { pn[row] = mine; q[row] = acc; pq[0] += mine * acc; }
    row += stride;
    fetch();
  }
  double* const dst[1] = {red.arr(P_PQ)};
  block_partials<1>(pq, dst);
}

// BiCGStab: y = A x with the dot products against y: MODE 0: rhat.y;  MODE 1: y.s, y.y
template <int MODE, int L0>
__global__ __launch_bounds__(FT) void bi_spmv_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                     const double* __restrict__ v, const double* __restrict__ x,
                                                     double* __restrict__ y, const double* __restrict__ other, Red red) {
  const int lane = threadIdx.x % L0, stride = gridDim.x * (FT / L0);
  int row = (blockIdx.x * FT + threadIdx.x) / L0, ra = 0, rb = 0;
  double orow = 0.0;
  auto fetch = [&]() { if (row < n) { ra = rp[row]; rb = rp[row + 1]; orow = other[row]; } };
  fetch();
  if (red.sc[S_DONE] != 0.0) return;
  double d[2] = {0.0, 0.0};
  while (row < n) {
    const double acc = row_sum_ab<L0>(ci, v, ra, rb, lane, [&](int c) { return x[c]; });
    if (lane == 0) { y[row] = acc; d[0] += acc * orow; d[1] += acc * acc; }
    row += stride;
    fetch();
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

// BiCGStab: omega = t.s / t.t, x += alpha phat + omega shat, r = s - omega t; rhat.r, r.r; ends the iteration (count)
__global__ __launch_bounds__(FT) void bi_update_kernel(int n, double* __restrict__ x, double* __restrict__ r,
                                                       const double* __restrict__ phat, const double* __restrict__ shat,
                                                       const double* __restrict__ s, const double* __restrict__ t,
                                                       const double* __restrict__ rhat, Red red, int np) {
  double* sc = red.sc;
  int i = blockIdx.x * FT + threadIdx.x;
  double xi = 0.0, ph = 0.0, sh = 0.0, si = 0.0, ti = 0.0, rh = 0.0;
  auto fetch = [&]() { if (i < n) { xi = x[i]; ph = phat[i]; sh = shat[i]; si = s[i]; ti = t[i]; rh = rhat[i]; } };
  fetch();
  const double* const src[2] = {red.arr(P_TS), red.arr(P_TT)};
  Totals<2> T;
  T.load(src, np);
  const double done = sc[S_DONE], alpha = sc[S_ALPHA], it = sc[S_IT];
  if (done != 0.0) return;
  double ts[2];
  T.finish(ts);
  const double om = ts[1] != 0.0 ? ts[0] / ts[1] : 0.0;      // t = 0 <=> s = 0: the half step was exact, r = s
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc[S_TS] = ts[0]; sc[S_TT] = ts[1]; sc[S_OMEGA] = om; sc[S_IT] = it + 1.0; }
  double d[2] = {0.0, 0.0};
  while (i < n) {
    x[i] = xi + (alpha * ph + om * sh);
    const double rn = si - om * ti;
    r[i] = rn;
    d[0] += rh * rn;
    d[1] += rn * rn;
    i += gridDim.x * FT;
    fetch();
  }
  double* const dst[2] = {red.arr(P_RHR), red.arr(P_RR)};
  block_partials<2>(d, dst);
}

// |a|^2 as block partial sums into array `which`
__global__ __launch_bounds__(FT) void gm_sqnorm_kernel(int n, const double* __restrict__ a, Red red, int which) {
  double d[1] = {0.0};
  for (int r = blockIdx.x * FT + threadIdx.x; r < n; r += gridDim.x * FT) d[0] += a[r] * a[r];
  double* const dst[1] = {red.arr(which)};
  block_partials<1>(d, dst);
}

// loop state at the start of a restart cycle (one block): target from |M^-1 b| (first cycle only), flags
__global__ __launch_bounds__(FT) void gm_start_kernel(Red red, int np_b, double rtol, double atol, int min_it, int first, int np_r) {
  double* sc = red.sc;
  double d[2], one[1];
  {
    const double* const src[1] = {red.arr(P_BS)};
    Totals<1> T;
    T.load(src, np_b);
    T.finish(one);
    d[0] = one[0];
  }
  {
    const double* const src[1] = {red.arr(P_RR)};
    Totals<1> T;
    T.load(src, np_r);
    T.finish(one);
    d[1] = one[0];
  }
  if (threadIdx.x != 0) return;
  if (first) {
    const double bb = d[0], bnorm = sqrt(bb);
    const double target = fmax(atol, rtol * (bnorm > 0.0 ? bnorm : 1.0));
    sc[S_BB] = bb; sc[S_TARGET2] = target * target;
    sc[S_IT] = 0.0; sc[S_FLAG] = 0.0; sc[S_MINIT] = (double)min_it;
    sc[S_RHO0] = 0.0; sc[S_RHO1] = 0.0; sc[S_BETA] = 0.0; sc[S_ALPHA] = 0.0; sc[S_OMEGA] = 1.0;      // (CG's recurrence, when it starts here)
    const double rr = d[1];                      // |b - A x0|^2: only to tell non-finite input apart
    sc[S_DONE] = !(bb - bb == 0.0) ? DONE_BAD_RHS : (!(rr - rr == 0.0) ? DONE_BAD_START : DONE_NO);
  }
}

// CG with the preconditioned-norm test, end of a chunk: the test the next iteration's first kernel would make (one block)
__global__ __launch_bounds__(FT) void cg_check_kernel(Red red, int np) {
  double* sc = red.sc;
  const double* const src[1] = {red.arr(P_ZZ)};
  Totals<1> T;
  T.load(src, np);
  const double done = sc[S_DONE], target2 = sc[S_TARGET2];
  if (done != 0.0) return;
  double zz[1];
  T.finish(zz);
  if (threadIdx.x == 0) {
    sc[S_RR] = zz[0];
    if (!(zz[0] > target2)) sc[S_DONE] = DONE_CONVERGED;
  }
}

// BiCGStab, end of a chunk: the convergence test the next iteration's first kernel would make (one block)
__global__ __launch_bounds__(FT) void bi_check_kernel(Red red, int np) {
  double* sc = red.sc;
  const double* const src[1] = {red.arr(P_RR)};
  Totals<1> T;
  T.load(src, np);
  const double done = sc[S_DONE], target2 = sc[S_TARGET2], its = sc[S_IT], min_it = sc[S_MINIT];
  if (done != 0.0) return;
  double rr[1];
  T.finish(rr);
  if (threadIdx.x == 0) {
    sc[S_RR] = rr[0];
    if (!(rr[0] > target2) && (its >= min_it || rr[0] == 0.0)) sc[S_DONE] = DONE_CONVERGED;
  }
}

// EMI, before the first residual: the initial guess out of the vertex records (phi is component 7 of the 64-byte records)
// and the block partial sums of b_emi, whose mean the residual kernel removes (constant null space: pdeSolver.py:74-78
// attaches it to the matrix, the right-hand side is projected here) -- one launch where the plain loop spends three, one of
// them a dot product with a ticket counter (8.6 us in the timeline of a step)
__global__ __launch_bounds__(FT) void emi_pre_kernel(int n, const double* __restrict__ phi, int stride, double* __restrict__ x,
                                                     const double* __restrict__ b, Red red) {
  double bs[1] = {0.0};
  for (int i = blockIdx.x * FT + threadIdx.x; i < n; i += gridDim.x * FT) {
    x[i] = phi[(size_t)i * stride];
    bs[0] += b[i];
  }
  double* const dst[1] = {red.arr(P_BS)};
  block_partials<1>(bs, dst);
  // a new solve: the cycles that run before the loop state is written (M^-1 b of the preconditioned-norm test) must not
  // see the previous solve's flag
  if (blockIdx.x == 0 && threadIdx.x == 0) red.sc[S_DONE] = DONE_NO;
}

// r = b - A x (and rhat = r for BiCGStab), |r|^2 and |b|^2 as block partial sums.  SHIFT (EMI): b is b_emi minus its mean
// (from emi_pre_kernel's np_b partial sums; inv_n = 1 / n)
template <int L0, bool SHIFT>
__global__ __launch_bounds__(FT) void residual_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                      const double* __restrict__ v, const double* __restrict__ x,
                                                      const double* __restrict__ b, double* __restrict__ r,
                                                      double* __restrict__ rhat, Red red, int np_b, double inv_n,
                                                      double* __restrict__ b_out) {
  const int lane = threadIdx.x % L0, stride = gridDim.x * (FT / L0);
  int row = (blockIdx.x * FT + threadIdx.x) / L0, ra = 0, rb = 0;
  double brow = 0.0;
  auto fetch = [&]() { if (row < n) { ra = rp[row]; rb = rp[row + 1]; brow = b[row]; } };
  fetch();
  double mean = 0.0;
  if constexpr (SHIFT) {
    const double* const src[1] = {red.arr(P_BS)};
    Totals<1> T;
    T.load(src, np_b);
    double tot[1];
    T.finish(tot);
    mean = tot[0] * inv_n;
    if (blockIdx.x == 0 && threadIdx.x == 0) red.sc[S_MEAN] = mean;
  }
  double d[2] = {0.0, 0.0};
  while (row < n) {
    const double acc = row_sum_ab<L0>(ci, v, ra, rb, lane, [&](int c) { return x[c]; });
    if (lane == 0) {
      const double bi = brow - mean, ri = bi - acc;
      r[row] = ri;
      if (rhat) rhat[row] = ri;
      if (b_out) b_out[row] = bi;       // the projected right-hand side, for |M^-1 b|
      d[0] += ri * ri; d[1] += bi * bi;
    }
    row += stride;
    fetch();
  }
  double* const dst[2] = {red.arr(P_RR), red.arr(P_PQ)};
  block_partials<2>(d, dst);
}

// EMI, after the last iteration: the solution made orthogonal to the constants and written into the vertex records.  Two
// launches without atomics: block partial sums of x, then every block re-sums them (<= 1 024) at its head.  Idempotent.
__global__ __launch_bounds__(FT) void x_sum_kernel(int n, const double* __restrict__ x, Red red) {
  double xs[1] = {0.0};
  for (int i = blockIdx.x * FT + threadIdx.x; i < n; i += gridDim.x * FT) xs[0] += x[i];
  double* const dst[1] = {red.arr(P_XS)};
  block_partials<1>(xs, dst);
}
__global__ __launch_bounds__(FT) void emi_post_kernel(int n, const double* __restrict__ x, double* __restrict__ phi, int stride,
                                                      Red red, int np, double inv_n) {
  int i = blockIdx.x * FT + threadIdx.x;
  double xi = i < n ? x[i] : 0.0;
  const double* const src[1] = {red.arr(P_XS)};
  Totals<1> T;
  T.load(src, np);
  double tot[1];
  T.finish(tot);
  const double mean = tot[0] * inv_n;
  if (blockIdx.x == 0 && threadIdx.x == 0) red.sc[S_MEAN] = mean;
  while (i < n) {
    phi[(size_t)i * stride] = xi - mean;
    i += gridDim.x * FT;
    if (i < n) xi = x[i];
  }
}

// Loop state at the start of a solve (one block): |r0|^2, |b|^2 from the residual kernel's partial sums, the target
// max(atol, rtol |b|) -- the host does not have to read anything before the first chunk.  BiCGStab: rhat = r, so
// rho_0 = r.r: ONE "partial sum" each in the arrays the first iteration reads, the other np_fill - 1 slots zeroed (every
// iteration then reads the same number of partial sums, whichever it is: the launches of an iteration are the same for all)
__global__ __launch_bounds__(FT) void start_kernel(Red red, int np, double rtol, double atol, int bicg, int np_fill, int min_it) {
  double* sc = red.sc;
  const double* const src[2] = {red.arr(P_RR), red.arr(P_PQ)};
  Totals<2> T;
  T.load(src, np);
  double d[2];
  T.finish(d);
  const double rr = d[0], bb = d[1];
  if (bicg)
    for (int i = threadIdx.x; i < np_fill; i += FT) { red.arr(P_RHR)[i] = i == 0 ? rr : 0.0; red.arr(P_RR)[i] = i == 0 ? rr : 0.0; }
  if (threadIdx.x != 0) return;
  const double bnorm = sqrt(bb);
  const double target = fmax(atol, rtol * (bnorm > 0.0 ? bnorm : 1.0));
  sc[S_RR] = rr; sc[S_BB] = bb;
  sc[S_TARGET2] = target * target;
  sc[S_IT] = 0.0; sc[S_FLAG] = 0.0; sc[S_MINIT] = (double)min_it;
  sc[S_RHO0] = 0.0; sc[S_RHO1] = 0.0; sc[S_BETA] = 0.0; sc[S_ALPHA] = bicg ? 1.0 : 0.0; sc[S_OMEGA] = 1.0;
  // non-finite data is an input error, not a breakdown: no kernel of the loop touches the iterate
  sc[S_DONE] = !(bb - bb == 0.0) ? DONE_BAD_RHS : (!(rr - rr == 0.0) ? DONE_BAD_START :
               ((rr > target * target || (min_it > 0 && rr != 0.0)) ? DONE_NO : DONE_CONVERGED));
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
  int l0;        // lanes per row of the finest operator's rows: 4 for short rows (<= 24 entries on average), else 16

  static int capped(int b) { return std::max(1, std::min(KN_PB, b)); }
  static int blocks_l(int rows, int lanes) { return (int)(((size_t)rows * lanes + FT - 1) / FT); }
  static int blocks16(int rows) { return blocks_l(rows, LPR); }
  int blocks0(int rows) const { return blocks_l(rows, l0); }
  int down_blocks(const KnAmgLevel& L, int la) const { return (L.n + FT / la - 1) / (FT / la) + (L.nc + FT / 64 - 1) / (FT / 64); }

  template <int IN>
  void launch_down(int la, int nb, const DownArgs& a) const {
    if (la == 4) hipLaunchKernelGGL((down_kernel<IN, 4>), dim3(nb), dim3(FT), 0, st, a);
    else hipLaunchKernelGGL((down_kernel<IN, 16>), dim3(nb), dim3(FT), 0, st, a);
  }

  template <bool CHECK>
  void launch_dense(const KnAmgLevel& C, int np) const {
    hipLaunchKernelGGL((dense_kernel<CHECK>), dim3(((size_t)C.n * 64 + FT - 1) / FT), dim3(FT), 0, st, C.n, C.dense_inv,
                       C.dense_blk, C.r, C.x, red, np);
  }

  // out = V-cycle(in) on the frozen hierarchy.  IN: how the finest level forms its input (down_kernel); np: partial sums
  // its scalar needs; par: parity of the iteration.  CHECK: CG's convergence test in the coarsest kernel.  Returns the
  // number of partial sums the finest kernels leave (IN_CG: r.r by the first, DOTS: r.z by the last).
  template <int IN, int DOTS, bool CHECK>
  void cycle(const double* r, const double* u, const double* w, double* formed, double* x, double* out, int np, int par,
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
      a.red = red; a.np = np; a.par = par;
      if (l == 0) {
        a.r = r; a.u = u; a.w = w; a.out = formed; a.x = x;
        const int nb = IN == IN_CG ? capped(down_blocks(L, l0)) : down_blocks(L, l0);
        if (np_rr) *np_rr = nb;
        static const bool split = getenv("KNPEMI_FUSED_NO_SPLIT") == nullptr;
        if ((IN == IN_BI_P || IN == IN_BI_S) && split) {
          hipLaunchKernelGGL((form_kernel<IN>), dim3(capped((L.n + FT - 1) / FT)), dim3(FT), 0, st, a);
          a.r = formed; a.out = nullptr;
          launch_down<IN_PLAIN>(l0, nb, a);
        } else {
          launch_down<IN>(l0, nb, a);
        }
      } else {
        a.r = L.r;
        launch_down<IN_PLAIN>(LPR, down_blocks(L, LPR), a);
      }
    }
    launch_dense<CHECK>(G.lev[nl - 1], np_rr ? *np_rr : 0);
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
        const int nb = DOTS != 0 ? capped(blocks16(L.n)) : blocks16(L.n);
        if (np_rz) *np_rz = nb;
        hipLaunchKernelGGL((up_kernel<DOTS>), dim3(nb), dim3(FT), 0, st, a);
      } else {
        a.r = L.r; a.x = L.x;
        hipLaunchKernelGGL((up_kernel<0>), dim3(blocks16(L.n)), dim3(FT), 0, st, a);
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
    hipLaunchKernelGGL((down_kernel<IN_PLAIN, LPR>), dim3(nb), dim3(FT), 0, st, a);
  }
  KnAmgLevel& C = G.lev[nl - 1];
  hipLaunchKernelGGL((dense_kernel<false>), dim3(((size_t)C.n * 64 + FT - 1) / FT), dim3(FT), 0, st, C.n, C.dense_inv, C.dense_blk,
                     C.r, C.x, red, 0);
  for (int l = nl - 2; l >= l0; --l) {
    KnAmgLevel& L = G.lev[l];
    UpArgs a{};
    a.n = L.n;
    a.prp = L.Pm.rp; a.pci = L.Pm.ci; a.pv = L.Pm.v;
    a.dinv = L.dinv; a.omega = L.omega;
    a.r = (l == l0 && r0) ? r0 : L.r; a.t = L.t; a.ec = G.lev[l + 1].x; a.x = (l == l0 && out0) ? out0 : L.x;
    a.red = red;
    hipLaunchKernelGGL((up_kernel<0>), dim3((int)(((size_t)L.n * LPR + FT - 1) / FT)), dim3(FT), 0, st, a);
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

// The loop state on the host.  A copy + stream synchronisation costs 25-40 us of idle device per read (the blit kernel, its
// completion signal, the host thread's wake-up; tools/timeline.py: the next kernel starts ~35 us after the last one of the
// solve).  Instead the last kernel of a chunk PUBLISHES the state: one lane copies the scalars into host memory mapped into
// the device's address space, fences at system scope and stores a sequence number behind them; the host spins on that
// number (it knows how many chunks it has enqueued on this handle).  After 2 ms without it -- an error on the stream -- the
// host falls back to the synchronising copy, which reports the error.
struct Publish {
  double* host_dev;            // device address of the mapped host buffer: [0, S_NF) scalars, [S_NF] sequence number
  unsigned long long* count;   // device counter of publications on this handle
};

__device__ __forceinline__ void publish_state(const Publish& p, const double* sc) {
  // (called by ONE lane of a kernel that follows every kernel that writes the state in stream order)
  for (int i = 0; i < S_NF; ++i) __hip_atomic_store(p.host_dev + i, sc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long c = *p.count + 1ull;
  *p.count = c;
  __atomic_thread_fence(__ATOMIC_RELEASE);      // system scope: the scalars are visible to the host before the number is
  __hip_atomic_store(p.host_dev + S_NF, (double)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void publish_kernel(Publish p, const double* __restrict__ sc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) publish_state(p, sc);
}

int ensure_publish(knpemi_handle* h, Publish* out) {
  if (!h->pub_host) {
    void* p = nullptr;
    KN_HIP(hipHostMalloc(&p, 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(p, 0, 64 * sizeof(double));
    void* d = nullptr;
    KN_HIP(hipHostGetDevicePointer(&d, p, 0));
    void* c = nullptr;
    KN_HIP(hipMalloc(&c, sizeof(unsigned long long)));
    h->allocs.push_back(c);
    KN_HIP(hipMemsetAsync(c, 0, sizeof(unsigned long long), h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
    h->pub_host = static_cast<double*>(p);
    h->pub_host_dev = static_cast<double*>(d);
    h->pub_count = static_cast<unsigned long long*>(c);
    h->pub_expected = 0;
  }
  out->host_dev = h->pub_host_dev;
  out->count = h->pub_count;
  return KNPEMI_OK;
}

bool publish_usable() {
  static const bool off = getenv("KNPEMI_NO_PUBLISH") != nullptr;
  return !off;
}

// enqueue the publication of the state (the caller counts it) ...
void enqueue_publish(knpemi_handle* h, const Publish& p, const double* sc_dev) {
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, h->stream, p, sc_dev);
}

// ... and wait for publication number h->pub_expected
int read_state(knpemi_handle* h, const double* sc_dev, double* host, bool published) {
  if (published) {
    const double want = (double)h->pub_expected;
    volatile double* seq = h->pub_host + S_NF;
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    bool seen = false;
    for (;;) {
      if (*seq >= want) { seen = true; break; }
      if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
        // long solves are legitimate (large meshes): keep polling, but let the runtime report a dead stream
        if (hipStreamQuery(h->stream) != hipErrorNotReady) break;
      }
    }
    if (seen) {
      std::atomic_thread_fence(std::memory_order_acquire);
      for (int i = 0; i < S_NF; ++i) host[i] = const_cast<const volatile double*>(h->pub_host)[i];
      return KNPEMI_OK;
    }
  }
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
// that turns out not to be needed ~15-40 us of kernels that return at once).
int first_chunk(int last_its, int maxit) {
  static const int forced = getenv("KNPEMI_FUSED_CHUNK") ? std::max(1, atoi(getenv("KNPEMI_FUSED_CHUNK"))) : 0;
  const int c = forced ? forced : std::min(16, std::max(1, last_its));
  return std::max(1, std::min(c, maxit));
}

// A chunk of a solve as ONE hipGraph launch.  On this stack the host needs 4.4-5 us to enqueue a kernel on a stream
// (tools/probes/kernel_head.hip: "stream" against "graph" columns; bench.py: host_enqueue_ms_per_step) while the device
// side of a dependent trivial kernel is 2.4-3.9 us: launched one by one the ~45 kernels of a solve are host-bound.  The
// kernels of a chunk depend on nothing the host knows -- the iteration count and the convergence flag live on the device,
// vector buffers and rho slots alternate with the parity of the iteration --, so a chunk is captured once per
// (system, hierarchy build, first / later chunk, number of iterations, parity, tolerances) and replayed afterwards.
// KNPEMI_NO_GRAPH=1, an active event profile (knpemi_profile brackets record events on the stream) or a partitioned
// problem (communication hooks): direct launches.
bool graphs_usable(const knpemi_handle* h) {
  static const bool off = getenv("KNPEMI_NO_GRAPH") != nullptr || getenv("KNPEMI_FUSED_NO_GRAPH") != nullptr;
  return !off && h->prof_mask == 0 && !h->dist.on;
}

// Graph replay or direct launches?  Which is faster depends on the host: where a launch costs the host 4.4-5 us the ~45
// kernels of a solve are host-bound and one graph launch wins; where it costs 2.5 us the stream keeps ahead of the device
// (5 us per kernel) and the graph only adds its launch latency (10-30 us on an idle stream) to every chunk.  Both were
// measured on boxes of this pool, so each system decides for itself: solves 2 .. 9 of a handle alternate between the two and
// time the first chunk (enqueue -> state on the host) per enqueued iteration, afterwards the faster one stays.
// KNPEMI_FUSED_GRAPH=1 / 0 forces the choice.
struct ModeChoice { bool graph; bool timed; };
ModeChoice choose_mode(knpemi_handle* h, int sys) {
  if (!graphs_usable(h)) return {false, false};
  static const char* forced = getenv("KNPEMI_FUSED_GRAPH");
  if (forced) return {atoi(forced) != 0, false};
  knpemi_handle::FusedMode& m = h->fused_mode[sys];
  if (m.decided) return {m.graph, false};
  const int s = m.solves++;
  if (s < 2) return {true, false};            // the graphs get captured and instantiated
  return {(s & 1) == 0, true};
}
void record_mode(knpemi_handle* h, int sys, bool graph, double us_per_iteration) {
  knpemi_handle::FusedMode& m = h->fused_mode[sys];
  m.t[graph ? 0 : 1] += us_per_iteration;
  ++m.n[graph ? 0 : 1];
  if (m.n[0] >= 4 && m.n[1] >= 4) {
    m.graph = m.t[0] / m.n[0] < m.t[1] / m.n[1];
    m.decided = true;
    if (getenv("KNPEMI_AMG_VERBOSE"))
      fprintf(stderr, "[knpemi fused] system %d: %.1f us per iteration from a graph, %.1f from the stream -> %s\n", sys, m.t[0] / m.n[0],
              m.t[1] / m.n[1], m.graph ? "graph" : "stream");
  }
}

uint64_t mix(uint64_t k, uint64_t v) { return (k ^ v) * 0x9E3779B97F4A7C15ull + (k << 6) + (k >> 2); }
uint64_t bits(double d) { uint64_t u; memcpy(&u, &d, sizeof u); return u; }

template <class Enqueue>
int run_chunk_graph(knpemi_handle* h, uint64_t key, Enqueue&& enqueue, bool use_graph) {
  if (!use_graph) return enqueue();
  auto it = h->fused_graphs.find(key);
  if (it == h->fused_graphs.end()) {
    if (h->fused_graphs.size() >= 96) {      // tolerances or hierarchies that keep changing: start over rather than grow
      for (auto& kv : h->fused_graphs) (void)hipGraphExecDestroy(kv.second);
      h->fused_graphs.clear();
    }
    hipGraph_t graph = nullptr;
    KN_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue();
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { kn_set_error(std::string("fused loop, hipStreamEndCapture: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { kn_set_error(std::string("fused loop, hipGraphInstantiate: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
    it = h->fused_graphs.emplace(key, exec).first;
  }
  KN_HIP(hipGraphLaunch(it->second, h->stream));
  return KNPEMI_OK;
}

int lanes0(const KnAmg& G) { return G.lev[0].avg_row <= 24 ? 4 : LPR; }

// launch of a kernel templated on the lanes per row of the finest operator (4 or 16)
#define KN_LAUNCH_L0(l0, grid, st, KERN4, KERN16, ...)                                  \
  do {                                                                                  \
    if ((l0) == 4) hipLaunchKernelGGL(KERN4, grid, dim3(FT), 0, st, __VA_ARGS__);       \
    else hipLaunchKernelGGL(KERN16, grid, dim3(FT), 0, st, __VA_ARGS__);                \
  } while (0)

}  // namespace

void kn_fused_graphs_free(knpemi_handle* h) {
  for (auto& kv : h->fused_graphs) (void)hipGraphExecDestroy(kv.second);
  h->fused_graphs.clear();
}

// CG on A_emi phi = b_emi - mean(b_emi); workspace vectors as in kn_solve_emi.  The initial guess is gathered from `phi`
// (component of the vertex records, stride doubles apart), the solution minus its mean is written back there: both are
// part of the captured graph of a chunk (the write-back is idempotent and follows every chunk).
int kn_fused_cg(knpemi_handle* h, KnAmg& G, const KnFusedSys& S, const double* b, double rtol, double atol, int maxit, int* iters,
                double* rr_out, double* bb_out, double* phi, int phi_stride) {
  const int n = S.n;
  const size_t N = S.N;
  double *x = S.work, *r = x + N, *z = r + N, *p = z + N, *q = p + N, *r2 = S.work + 7 * N, *p2 = S.work + 8 * N;
  int rc = ensure_partials(h);
  if (rc) return rc;
  Loop L{h, G, S, Red{S.sc, h->fused_part}, h->stream, lanes0(G)};
  const int l0 = L.l0;
  const int nb_res = Loop::capped(L.blocks0(n)), nb_dir = Loop::capped(L.blocks0(n));
  int np_rr = 0, np_rz = 0;
  const int nb_vec = Loop::capped((n + FT - 1) / FT);
  const double inv_n = 1.0 / (double)n;
  // KNPEMI_OPT_FOLD_MEMBRANE: the write-back launch also integrates the membrane facets (the potential system of the CG
  // path only: its unknowns are the vertex records' phi)
  const bool fold = h->fold_membrane && !h->fuse_membrane && h->have_params && !h->plain_knp && phi == h->dev.VR + 7;
  const bool pre_norm = h->emi_norm_pre;
  auto head = [&]() -> int {
    // x0 = current phi (ksp_initial_guess_nonzero), b = b_emi projected onto zero mean (constant null space)
    hipLaunchKernelGGL(emi_pre_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, (const double*)phi, phi_stride, x, b, L.red);
    if (pre_norm) {
      // KSPCG's default test: |M^-1 r| <= max(atol, rtol |M^-1 b|).  The projected b goes to q (free until the first
      // direction), M^-1 b to p2, its square norm replaces the sums of b in P_BS
      KN_LAUNCH_L0(l0, dim3(nb_res), h->stream, (residual_kernel<4, true>), (residual_kernel<16, true>), n, S.rowptr, S.colind,
                   S.vals, (const double*)x, b, r, (double*)nullptr, L.red, nb_vec, inv_n, q);
      L.cycle<IN_PLAIN, 0, false>(q, nullptr, nullptr, nullptr, nullptr, p2, 0, 0, nullptr, nullptr);
      hipLaunchKernelGGL(gm_sqnorm_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, (const double*)p2, L.red, (int)P_BS);
      hipLaunchKernelGGL(gm_start_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, nb_vec, rtol, atol, 0, 1, nb_res);
      // z_0 = M^-1 r_0 with r_0.z_0 and |z_0|^2: the first direction kernel tests the initial residual
      L.cycle<IN_PLAIN, 3, false>(r, nullptr, nullptr, nullptr, nullptr, z, 0, 0, nullptr, &np_rz);
      return KNPEMI_OK;
    }
    KN_LAUNCH_L0(l0, dim3(nb_res), h->stream, (residual_kernel<4, true>), (residual_kernel<16, true>), n, S.rowptr, S.colind,
                 S.vals, (const double*)x, b, r, (double*)nullptr, L.red, nb_vec, inv_n, (double*)nullptr);
    hipLaunchKernelGGL(start_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, nb_res, rtol, atol, 0, 0, 0);
    // z_0 = M^-1 r_0, r_0.z_0 (rho_old = 0: the first direction is z_0)
    L.cycle<IN_PLAIN, 1, false>(r, nullptr, nullptr, nullptr, nullptr, z, 0, 0, nullptr, &np_rz);
    return KNPEMI_OK;
  };
  int k = 0;     // iterations enqueued so far: their parity decides which of the alternating buffers is which
  auto iterations = [&](int count) -> int {
    for (int j = 0; j < count; ++j, ++k) {
      if (pre_norm) {
        KN_LAUNCH_L0(l0, dim3(nb_dir), h->stream, (cg_dir_kernel<4, true>), (cg_dir_kernel<16, true>), n, S.rowptr, S.colind, S.vals,
                     (const double*)z, (const double*)p, p2, q, L.red, np_rz, k & 1);
        // r_new = r - alpha q (stored in r2), x += alpha p_new; z = M^-1 r_new with r_new.z and |z|^2; the iteration is counted
        L.cycle<IN_CG, 7, false>(r, q, p2, r2, x, z, nb_dir, k & 1, &np_rr, &np_rz);
      } else {
        KN_LAUNCH_L0(l0, dim3(nb_dir), h->stream, (cg_dir_kernel<4, false>), (cg_dir_kernel<16, false>), n, S.rowptr, S.colind, S.vals,
                     (const double*)z, (const double*)p, p2, q, L.red, np_rz, k & 1);
        // r_new = r - alpha q (stored in r2), x += alpha p_new, |r_new|^2 -> convergence; then z = M^-1 r_new, r_new.z
        L.cycle<IN_CG, 1, true>(r, q, p2, r2, x, z, nb_dir, k & 1, &np_rr, &np_rz);
      }
      std::swap(r, r2);
      std::swap(p, p2);
    }
    // (the test the next direction kernel would make, for the host that reads the state after this chunk)
    if (pre_norm) hipLaunchKernelGGL(cg_check_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, np_rz);
    return KNPEMI_OK;
  };
  uint64_t base = mix(mix(mix(mix(0xC6ull, (uint64_t)(uintptr_t)S.work), (uint64_t)G.builds), bits(rtol)), bits(atol));
  base = mix(mix(mix(base, (uint64_t)n), (uint64_t)(uintptr_t)b), (uint64_t)(uintptr_t)phi);
  base = mix(mix(base, (uint64_t)(h->fold_membrane && !h->fuse_membrane)), (uint64_t)(h->emi_flags & KNPEMI_NO_SPLITTING));
  base = mix(base, pre_norm ? 2 : 0);
  double sc[S_NF];
  int it = 0, todo = first_chunk(G.its_last, maxit);
  bool first = true;
  Publish pub{};
  const bool use_pub = publish_usable();
  if (use_pub && (rc = ensure_publish(h, &pub))) return rc;
  base = mix(base, use_pub ? 1 : 0);
  const ModeChoice mode = choose_mode(h, 0);
  const auto t_start = std::chrono::steady_clock::now();
  for (;;) {
    const int k0 = k;
    const uint64_t key = mix(mix(mix(base, first ? 1 : 0), (uint64_t)todo), (uint64_t)(k0 & 1));
    auto chunk = [&]() -> int {
      if (first) if (int e = head()) return e;
      if (int e = iterations(todo)) return e;
      // the solution orthogonal to the constants, into the phi component of the vertex records (idempotent)
      hipLaunchKernelGGL(x_sum_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, (const double*)x, L.red);
      if (fold) {   // ... and, in the same launch, the membrane-facet integrals of b_knp for that potential
        if (int e = kn_launch_emi_writeback_membrane(h, x, L.red.part + (size_t)P_XS * KN_PB, nb_vec, inv_n, S.sc + S_MEAN)) return e;
      } else
      hipLaunchKernelGGL(emi_post_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, (const double*)x, phi, phi_stride, L.red, nb_vec, inv_n);
      if (use_pub) enqueue_publish(h, pub, S.sc);
      return KNPEMI_OK;
    };
    if (mode.graph && h->fused_graphs.count(key)) {
      // replay: the host-side state advances as if the chunk had been enqueued
      if (first) np_rz = Loop::capped(Loop::blocks16(G.lev[0].n));
      if ((todo & 1)) { std::swap(r, r2); std::swap(p, p2); }
      k += todo;
      KN_HIP(hipGraphLaunch(h->fused_graphs[key], h->stream));
      if (fold) { h->gam_valid = h->dev.nftot > 0; h->gam_split = (h->emi_flags & KNPEMI_NO_SPLITTING) ? 0 : 1; }
    } else if ((rc = run_chunk_graph(h, key, chunk, mode.graph))) return rc;
    if (use_pub) ++h->pub_expected;
    if ((rc = read_state(h, S.sc, sc, use_pub))) return rc;
    if (first && mode.timed)
      record_mode(h, 0, mode.graph, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count() / (todo + 1.5));
    first = false;
    if ((rc = bad_input("EMI CG", sc))) return rc;
    it = (int)sc[S_IT];
    if (!std::isfinite(sc[S_RR])) { kn_set_error("EMI CG broke down (non-finite residual) " + describe(sc)); return KNPEMI_ESOLVE; }
    if (sc[S_DONE] != 0.0 || it >= maxit) break;
    todo = std::min(1, maxit - it);
    if (todo <= 0) break;
    (void)k0;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("fused CG: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  G.its_last = it;
  *iters = it;
  *rr_out = sc[S_RR];
  *bb_out = sc[S_BB];
  return KNPEMI_OK;
}

// Right-preconditioned BiCGStab on S x = b from the x in the workspace; vectors as in kn_solve_knp; pre / post as above.
int kn_fused_bicgstab(knpemi_handle* h, KnAmg& G, const KnFusedSys& S, const double* b, double rtol, double atol, int maxit,
                      int* iters, double* rr_out, double* bb_out, const std::function<int()>& pre,
                      const std::function<int()>& post) {
  const int n = S.n;
  const int min_it = std::max(0, std::min(h->knp_min_it, maxit));
  const size_t N = S.N;
  double *x = S.work, *r = x + N, *rhat = r + N, *p = rhat + N, *v = p + N, *s = v + N, *t = s + N;
  double *phat = S.work + 8 * N, *shat = S.work + 9 * N;
  int rc = ensure_partials(h);
  if (rc) return rc;
  Loop L{h, G, S, Red{S.sc, h->fused_part}, h->stream, lanes0(G)};
  const int l0 = L.l0;
  const int nb_res = Loop::capped(L.blocks0(n)), nb_spmv = Loop::capped(L.blocks0(n)), nb_update = Loop::capped((n + FT - 1) / FT);
  auto head = [&]() -> int {
    if (int e = pre()) return e;
    KN_LAUNCH_L0(l0, dim3(nb_res), h->stream, (residual_kernel<4, false>), (residual_kernel<16, false>), n, S.rowptr, S.colind,
                 S.vals, (const double*)x, b, r, rhat, L.red, 0, 0.0, (double*)nullptr);
    hipLaunchKernelGGL(start_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, nb_res, rtol, atol, 1, nb_update, min_it);
    return KNPEMI_OK;
  };
  int k = 0;
  auto iterations = [&](int count) -> int {
    for (int j = 0; j < count; ++j, ++k) {
      // p = r + beta (p - omega v) is formed by the finest down kernel, which reads p and v at neighbouring rows while
      // it stores its own: the new direction goes to the buffer of t (free until the second SpMV) and the two names swap
      L.cycle<IN_BI_P, 0, false>(r, p, v, t, nullptr, phat, nb_update, k & 1, nullptr, nullptr);
      std::swap(p, t);
      KN_LAUNCH_L0(l0, dim3(nb_spmv), h->stream, (bi_spmv_kernel<0, 4>), (bi_spmv_kernel<0, 16>), n, S.rowptr, S.colind, S.vals,
                   (const double*)phat, v, (const double*)rhat, L.red);
      L.cycle<IN_BI_S, 0, false>(r, v, nullptr, s, nullptr, shat, nb_spmv, k & 1, nullptr, nullptr);
      KN_LAUNCH_L0(l0, dim3(nb_spmv), h->stream, (bi_spmv_kernel<1, 4>), (bi_spmv_kernel<1, 16>), n, S.rowptr, S.colind, S.vals,
                   (const double*)shat, t, (const double*)s, L.red);
      hipLaunchKernelGGL(bi_update_kernel, dim3(nb_update), dim3(FT), 0, h->stream, n, x, r, (const double*)phat,
                         (const double*)shat, (const double*)s, (const double*)t, (const double*)rhat, L.red, nb_spmv);
    }
    hipLaunchKernelGGL(bi_check_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, nb_update);
    return KNPEMI_OK;
  };
  Publish pub{};
  const bool use_pub = publish_usable();
  if (use_pub && (rc = ensure_publish(h, &pub))) return rc;
  const ModeChoice mode = choose_mode(h, 1);
  const auto t_start = std::chrono::steady_clock::now();
  uint64_t base = mix(mix(mix(mix(0xB1ull, (uint64_t)(uintptr_t)S.work), (uint64_t)G.builds), bits(rtol)), bits(atol));
  base = mix(mix(mix(mix(base, (uint64_t)n), (uint64_t)(uintptr_t)b), (uint64_t)h->fuse_update), publish_usable() ? 1 : 0);
  double sc[S_NF];
  base = mix(base, (uint64_t)min_it);
  int it = 0, restarts = 0, todo = std::max(first_chunk(G.its_last, maxit), min_it);
  bool first = true;
  for (;;) {
    const uint64_t key = mix(mix(mix(base, first ? 1 : 0), (uint64_t)todo), (uint64_t)(k & 1));
    auto chunk = [&]() -> int {
      if (first) if (int e = head()) return e;
      if (int e = iterations(todo)) return e;
      if (int e = post()) return e;
      if (use_pub) enqueue_publish(h, pub, S.sc);
      return KNPEMI_OK;
    };
    if (mode.graph && h->fused_graphs.count(key)) {
      if ((todo & 1)) std::swap(p, t);
      k += todo;
      KN_HIP(hipGraphLaunch(h->fused_graphs[key], h->stream));
    } else if ((rc = run_chunk_graph(h, key, chunk, mode.graph))) return rc;
    if (use_pub) ++h->pub_expected;
    if ((rc = read_state(h, S.sc, sc, use_pub))) return rc;
    if (first && mode.timed)
      record_mode(h, 1, mode.graph, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count() / (todo + 0.5));
    first = false;
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
      hipLaunchKernelGGL(bi_restart_kernel, dim3((std::max(n, KN_PB) + 255) / 256), dim3(256), 0, h->stream, n, rhat, r, L.red, nb_update);
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

// =====================================================================================================================
// GMRES(30), left-preconditioned, classical Gram-Schmidt, convergence on the PRECONDITIONED residual norm relative to
// |M^-1 b| -- what PETSc's defaults make of the reference's options for the concentration solve (`ksp_type gmres`,
// pdeSolver.py:100; KSPGMRES: restart 30, left preconditioning, classical Gram-Schmidt without refinement, preconditioned
// norm; KSPConvergedDefault with a non-zero initial guess measures against the preconditioned right-hand side).
// KNPEMI_OPT_KNP_METHOD = 1 selects it; the default stays BiCGStab on the true residual (fewer launches per V-cycle).
// Per Arnoldi step: normalise + SpMV (the head of that kernel applies the Givens rotations to the previous column and
// decides convergence, every block for itself from the same numbers), the V-cycle (5 launches with three levels), the
// j + 1 dot products as block partial sums, the orthogonalisation (its head re-sums them) = 8 launches.
// =====================================================================================================================
namespace {
constexpr int GM_M = 30;                      // restart length (KSPGMRES default)
// device scalars of a GMRES solve behind the shared ones: beta, the rotations, the rotated (upper triangular) columns
enum { GM_BETA = 0, GM_CS = 8, GM_SN = GM_CS + GM_M, GM_R = GM_SN + GM_M, GM_H = GM_R + GM_M * GM_M, GM_N = GM_H + GM_M + 2 };

struct GmState { double* gm; double* dots; };   // dots: (GM_M + 1) arrays of KN_PB partial sums

// The scalar work of step j (j >= 1: column j - 1 of the Hessenberg matrix is complete once |w| = h_{j,j-1} is known): every
// thread of every block computes it from the same inputs; block 0 records.  Returns false when the solve is over.
__device__ __forceinline__ bool gm_column(const Red& red, const GmState& g, int j, int base, double ww, double& inv_h) {
  double* sc = red.sc;
  const double hj = sqrt(ww);
  inv_h = hj != 0.0 ? 1.0 / hj : 0.0;
  const double target2 = sc[S_TARGET2], min_it = sc[S_MINIT];
  const bool rec = blockIdx.x == 0 && threadIdx.x == 0;
  double res;
  if (j == 0) {                                  // hj = beta = |M^-1 (b - A x0)|
    res = hj;
    if (rec) g.gm[GM_BETA] = hj;
  } else {
    // rotate column j - 1 by the earlier rotations, then annihilate its subdiagonal entry hj
    const double beta = g.gm[GM_BETA];
    double gj = beta;                            // g_{j-1} after the earlier rotations
    double col_prev = g.gm[GM_H + 0];
    for (int i = 0; i < j - 1; ++i) {
      const double c = g.gm[GM_CS + i], s = g.gm[GM_SN + i];
      const double hn = g.gm[GM_H + i + 1];
      const double a = c * col_prev + s * hn, bnew = -s * col_prev + c * hn;
      if (rec) g.gm[GM_R + (j - 1) * GM_M + i] = a;
      col_prev = bnew;
      gj = -s * gj;
    }
    const double d = sqrt(col_prev * col_prev + hj * hj);
    const double c = d != 0.0 ? col_prev / d : 1.0, s = d != 0.0 ? hj / d : 0.0;
    if (rec) { g.gm[GM_R + (j - 1) * GM_M + (j - 1)] = d; g.gm[GM_CS + j - 1] = c; g.gm[GM_SN + j - 1] = s; }
    res = fabs(s * gj);                          // |g_j|
  }
  const double it = (double)(base + j);
  const bool done = (!(res * res > target2) && it >= min_it) || hj == 0.0;
  if (rec) {
    sc[S_RR] = res * res; sc[S_IT] = it;
    if (done) sc[S_DONE] = DONE_CONVERGED;
  }
  return !done;
}

// step j: v_j = w / |w| (stored), t = A v_j
template <int L0>
__global__ __launch_bounds__(FT) void gm_spmv_kernel(int n, const int* __restrict__ rp, const int* __restrict__ ci,
                                                     const double* __restrict__ av, const double* __restrict__ w,
                                                     double* __restrict__ vj, double* __restrict__ t, Red red, GmState g, int np,
                                                     int j, int base) {
  const int lane = threadIdx.x % L0, stride = gridDim.x * (FT / L0);
  int row = (blockIdx.x * FT + threadIdx.x) / L0, ra = 0, rb = 0;
  double wr = 0.0;
  auto fetch = [&]() { if (row < n) { ra = rp[row]; rb = rp[row + 1]; wr = w[row]; } };
  fetch();
  const double* const src[1] = {red.arr(P_PQ)};
  Totals<1> T;
  T.load(src, np);
  if (red.sc[S_DONE] != 0.0) return;
  double ww[1];
  T.finish(ww);
  double inv = 0.0;
  if (!gm_column(red, g, j, base, ww[0], inv)) return;
  while (row < n) {
    const double acc = row_sum_ab<L0>(ci, av, ra, rb, lane, [&](int c) { return w[c]; });
    if (lane == 0) { vj[row] = wr * inv; t[row] = acc * inv; }
    row += stride;
    fetch();
  }
}

// end of a chunk / of a restart cycle: the scalar work of step j alone (one block)
__global__ __launch_bounds__(FT) void gm_check_kernel(Red red, GmState g, int np, int j, int base) {
  const double* const src[1] = {red.arr(P_PQ)};
  Totals<1> T;
  T.load(src, np);
  if (red.sc[S_DONE] != 0.0) return;
  double ww[1];
  T.finish(ww);
  double inv;
  (void)gm_column(red, g, j, base, ww[0], inv);
}

// partial sums of w . v_i, i = 0 .. j (one array of partial sums per i)
__global__ __launch_bounds__(FT) void gm_dots_kernel(int n, const double* __restrict__ w, const double* __restrict__ V, size_t ldv,
                                                     int j, Red red, GmState g) {
  if (red.sc[S_DONE] != 0.0) return;
  for (int i = 0; i <= j; ++i) {
    const double* v = V + (size_t)i * ldv;
    double d[1] = {0.0};
    for (int r = blockIdx.x * FT + threadIdx.x; r < n; r += gridDim.x * FT) d[0] += w[r] * v[r];
    double* const dst[1] = {g.dots + (size_t)i * KN_PB};
    block_partials<1>(d, dst);
  }
}

// h_i = w . v_i from the partial sums (block 0 records column j), w -= sum_i h_i v_i, partial sums of |w|^2
__global__ __launch_bounds__(FT) void gm_orth_kernel(int n, double* __restrict__ w, const double* __restrict__ V, size_t ldv, int j,
                                                     Red red, GmState g, int np) {
  if (red.sc[S_DONE] != 0.0) return;
  __shared__ double hs[GM_M + 1];
  for (int i = 0; i <= j; ++i) {
    const double* const src[1] = {g.dots + (size_t)i * KN_PB};
    Totals<1> T;
    T.load(src, np);
    double h[1];
    T.finish(h);
    if (threadIdx.x == 0) { hs[i] = h[0]; if (blockIdx.x == 0) g.gm[GM_H + i] = h[0]; }
  }
  __syncthreads();
  double ww[1] = {0.0};
  for (int r = blockIdx.x * FT + threadIdx.x; r < n; r += gridDim.x * FT) {
    double x = w[r];
    for (int i = 0; i <= j; ++i) x -= hs[i] * V[(size_t)i * ldv + r];
    w[r] = x;
    ww[0] += x * x;
  }
  double* const dst[1] = {red.arr(P_PQ)};
  block_partials<1>(ww, dst);
}

// x += sum_{i < k} y_i v_i, R y = g (k = columns completed in this restart cycle = S_IT - base): the triangular solve is a few
// hundred operations, every thread does it for itself
__global__ __launch_bounds__(FT) void gm_update_kernel(int n, double* __restrict__ x, const double* __restrict__ V, size_t ldv,
                                                       Red red, GmState g, int base) {
  const int k = (int)red.sc[S_IT] - base;
  if (k <= 0) return;
  __shared__ double ys[GM_M];
  if (threadIdx.x == 0) {
    double gv[GM_M];
    double gj = g.gm[GM_BETA];
    for (int i = 0; i < k; ++i) { const double c = g.gm[GM_CS + i], s = g.gm[GM_SN + i]; gv[i] = c * gj; gj = -s * gj; }
    for (int i = k - 1; i >= 0; --i) {
      double a = gv[i];
      for (int l = i + 1; l < k; ++l) a -= g.gm[GM_R + l * GM_M + i] * ys[l];
      ys[i] = a / g.gm[GM_R + i * GM_M + i];
    }
  }
  __syncthreads();
  for (int r = blockIdx.x * FT + threadIdx.x; r < n; r += gridDim.x * FT) {
    double a = x[r];
    for (int i = 0; i < k; ++i) a += ys[i] * V[(size_t)i * ldv + r];
    x[r] = a;
  }
}

int ensure_gmres(knpemi_handle* h, size_t n, GmState* out) {
  const size_t need = (size_t)GM_M * n;
  if (h->gm_n < need) {
    void* p = nullptr;
    KN_HIP(hipMalloc(&p, need * sizeof(double)));
    h->allocs.push_back(p);
    h->gm_V = static_cast<double*>(p);
    h->gm_n = need;
  }
  if (!h->gm_state) {
    void* p = nullptr;
    const size_t doubles = GM_N + (size_t)(GM_M + 1) * KN_PB;
    KN_HIP(hipMalloc(&p, doubles * sizeof(double)));
    h->allocs.push_back(p);
    KN_HIP(hipMemsetAsync(p, 0, doubles * sizeof(double), h->stream));
    h->gm_state = static_cast<double*>(p);
  }
  out->gm = h->gm_state;
  out->dots = h->gm_state + GM_N;
  return KNPEMI_OK;
}

}  // namespace

int kn_fused_gmres(knpemi_handle* h, KnAmg& G, const KnFusedSys& S, const double* b, double rtol, double atol, int maxit,
                   int* iters, double* rr_out, double* bb_out, const std::function<int()>& pre, const std::function<int()>& post) {
  const int n = S.n;
  const size_t N = S.N;
  double *x = S.work, *r = x + N, *w = r + N, *t = w + N, *tmp = t + N;
  int rc = ensure_partials(h);
  if (rc) return rc;
  GmState g{};
  if ((rc = ensure_gmres(h, (size_t)n, &g))) return rc;
  double* V = h->gm_V;
  const size_t ldv = (size_t)n;
  Loop L{h, G, S, Red{S.sc, h->fused_part}, h->stream, lanes0(G)};
  const int l0 = L.l0;
  const int nb_row = Loop::capped(L.blocks0(n)), nb_vec = Loop::capped((n + FT - 1) / FT);
  const int min_it = std::max(0, std::min(h->knp_min_it, maxit));
  Publish pub{};
  const bool use_pub = publish_usable();
  if (use_pub && (rc = ensure_publish(h, &pub))) return rc;
  double sc[S_NF];
  int it = 0, base = 0;
  bool first = true;
  // restart length: KSPGMRES's default 30; KNPEMI_GMRES_RESTART (2 .. 30) shortens it (the restart path in the tests)
  static const int m_restart = [] { const char* e = getenv("KNPEMI_GMRES_RESTART"); const int v = e ? atoi(e) : GM_M; return v >= 2 && v <= GM_M ? v : GM_M; }();
  if ((rc = pre())) return rc;
  // (the kernels of the cycle look at the early-out flag: clear what the previous solve left before the first of them runs)
  KN_HIP(hipMemsetAsync(S.sc + S_DONE, 0, sizeof(double), h->stream));
  for (;;) {      // restart cycles
    // r = b - A x; w = M^-1 r; first cycle: also |M^-1 b| for the target
    KN_LAUNCH_L0(l0, dim3(nb_row), h->stream, (residual_kernel<4, false>), (residual_kernel<16, false>), n, S.rowptr, S.colind,
                 S.vals, (const double*)x, b, r, (double*)nullptr, L.red, 0, 0.0, (double*)nullptr);
    if (first) {
      L.cycle<IN_PLAIN, 0, false>(b, nullptr, nullptr, nullptr, nullptr, tmp, 0, 0, nullptr, nullptr);
      hipLaunchKernelGGL(gm_sqnorm_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, (const double*)tmp, L.red, (int)P_BS);
    }
    hipLaunchKernelGGL(gm_start_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, nb_vec, rtol, atol, min_it, first ? 1 : 0, nb_row);
    L.cycle<IN_PLAIN, 0, false>(r, nullptr, nullptr, nullptr, nullptr, w, 0, 0, nullptr, nullptr);
    hipLaunchKernelGGL(gm_sqnorm_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, (const double*)w, L.red, (int)P_PQ);
    first = false;
    int j = 0;
    int todo = std::max(1, std::min({std::max(first_chunk(G.its_last, maxit), min_it) - base, m_restart, maxit - base}));
    bool done = false;
    while (!done) {
      for (int k = 0; k < todo && j < m_restart; ++k, ++j) {
        KN_LAUNCH_L0(l0, dim3(nb_row), h->stream, gm_spmv_kernel<4>, gm_spmv_kernel<16>, n, S.rowptr, S.colind, S.vals,
                     (const double*)w, V + (size_t)j * ldv, t, L.red, g, nb_vec, j, base);
        L.cycle<IN_PLAIN, 0, false>(t, nullptr, nullptr, nullptr, nullptr, w, 0, 0, nullptr, nullptr);
        hipLaunchKernelGGL(gm_dots_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, (const double*)w, (const double*)V, ldv, j, L.red, g);
        hipLaunchKernelGGL(gm_orth_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, w, (const double*)V, ldv, j, L.red, g, nb_vec);
      }
      hipLaunchKernelGGL(gm_check_kernel, dim3(1), dim3(FT), 0, h->stream, L.red, g, nb_vec, j, base);
      if (use_pub) { enqueue_publish(h, pub, S.sc); ++h->pub_expected; }
      if ((rc = read_state(h, S.sc, sc, use_pub))) return rc;
      if ((rc = bad_input("KNP GMRES", sc))) return rc;
      it = (int)sc[S_IT];
      if (!std::isfinite(sc[S_RR])) { kn_set_error("KNP GMRES broke down (non-finite residual) " + describe(sc)); return KNPEMI_ESOLVE; }
      done = sc[S_DONE] != 0.0 || it >= maxit || j >= m_restart;
      todo = 1;
    }
    hipLaunchKernelGGL(gm_update_kernel, dim3(nb_vec), dim3(FT), 0, h->stream, n, x, (const double*)V, ldv, L.red, g, base);
    if (sc[S_DONE] != 0.0 || it >= maxit) break;
    base = it;       // restart from the updated iterate
  }
  if ((rc = post())) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("fused GMRES: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  G.its_last = it;
  *iters = it;
  *rr_out = sc[S_RR];
  *bb_out = sc[S_BB];
  return KNPEMI_OK;
}

// ---- diagnostics: a chain of dependent trivial kernels on the handle's stream, timed on the host (knpemi_debug_launch_chain)
namespace {
__global__ void chain_empty_kernel(double* a) { if (a == nullptr) a[0] = 0.0; }
__global__ __launch_bounds__(256) void chain_copy_kernel(const double* __restrict__ x, double* __restrict__ y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = x[i] + 1.0;
}
}  // namespace

extern "C" int knpemi_debug_launch_chain(knpemi_handle* h, int kind, int n, int links, int reps, int use_graph, double* us_per_kernel) {
  if (!h || !us_per_kernel || links < 1 || reps < 1 || n < 1) { kn_set_error("knpemi_debug_launch_chain: bad arguments"); return KNPEMI_EINVAL; }
  KN_HIP(hipSetDevice(h->device));
  int rc = ensure_partials(h);
  if (rc) return rc;
  void* buf = nullptr;
  KN_HIP(hipMalloc(&buf, (size_t)(2 * n + 64) * sizeof(double)));
  KN_HIP(hipMemsetAsync(buf, 0, (size_t)(2 * n + 64) * sizeof(double), h->stream));
  double* a = static_cast<double*>(buf);
  double* b = a + n;
  const Red red{a + 2 * n, h->fused_part};
  auto launch = [&](int k) {
    if (kind == 0) hipLaunchKernelGGL(chain_empty_kernel, dim3(1), dim3(64), 0, h->stream, a);
    else if (kind == 1) hipLaunchKernelGGL(chain_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, (k & 1) ? b : a, (k & 1) ? a : b, n);
    else hipLaunchKernelGGL(start_kernel, dim3(1), dim3(FT), 0, h->stream, red, 512, 1e-5, 1e-40, 0, 0, 0);
  };
  for (int k = 0; k < links; ++k) launch(k);
  KN_HIP(hipStreamSynchronize(h->stream));
  hipGraphExec_t exec = nullptr;
  if (use_graph) {
    hipGraph_t g = nullptr;
    KN_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < links; ++k) launch(k);
    KN_HIP(hipStreamEndCapture(h->stream, &g));
    KN_HIP(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    for (int w = 0; w < 5; ++w) KN_HIP(hipGraphLaunch(exec, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) {
    if (exec) KN_HIP(hipGraphLaunch(exec, h->stream));
    else for (int k = 0; k < links; ++k) launch(k);
  }
  KN_HIP(hipStreamSynchronize(h->stream));
  *us_per_kernel = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ((double)reps * links);
  if (exec) (void)hipGraphExecDestroy(exec);
  (void)hipFree(buf);
  return KNPEMI_OK;
}
