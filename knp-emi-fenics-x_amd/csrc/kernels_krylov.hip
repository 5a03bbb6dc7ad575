// Device-resident Krylov solves for the two sub-problems (the row next to the
// hot path).  The reference hands its systems to PETSc (src/knpemi/pdeSolver.py:13-38,88-113): CG +
// hypre BoomerAMG for the symmetric positive semi-definite EMI system with the constant null space
// attached (:74-78), GMRES + BoomerAMG for the non-symmetric block-diagonal KNP system, both with a
// non-zero initial guess (the previous solution).  Here: CG with the right-hand side and the solution
// projected onto zero mean, and right-preconditioned BiCGStab, both preconditioned with a smoothed-aggregation
// V-cycle (kernels_amg.hip; Jacobi selectable), working in place on the assembled device CSR.
//
// All vectors and scalars stay on the device; the host reads back one residual norm per chunk of iterations
// (one iteration with AMG) to test convergence.  A dot product, its block partials and the scalar update it
// feeds are one launch with a fixed summation order, so the solves are bit-reproducible.
#include <cmath>
#include <cstdio>

#include "knpemi_internal.h"

namespace {

constexpr int RED_BLOCKS = 512, RED_THREADS = 256;

// sc[] layout (device scalars)
enum { S_RHO = 0, S_RHO_OLD, S_PAP, S_ALPHA, S_BETA, S_OMEGA, S_RR, S_BB, S_TS, S_TT, S_MEAN, S_FLAG, S_RV, S_N };
// S_FLAG: breakdown events of the current chunk of iterations (bit set, stored as a double): a vanishing denominator never
// reaches a division -- the quotient is replaced by 0 and the event is recorded, so a breakdown surfaces as a flag the host
// acts on (converged / restart / error with the scalars in the message), not as a NaN that has already overwritten x.
enum { F_RHO_ZERO = 1, F_RV_ZERO = 2, F_OMEGA_ZERO = 4, F_PAP_ZERO = 8 };

// y = A * (x .* dinv?), or (dinv == NULL, b != NULL) the residual y = b - A x : LPR lanes per row, four independent
// index -> value -> gather chains per lane and trip
template <bool SCALED, int LPR>
__global__ __launch_bounds__(256) void spmv_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ colind,
                                                   const double* __restrict__ vals, const double* __restrict__ x,
                                                   const double* __restrict__ dinv, double* __restrict__ y,
                                                   const double* __restrict__ b = nullptr,
                                                   const uint8_t* __restrict__ owned = nullptr,
                                                   double* __restrict__ dinv_out = nullptr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = t / LPR, l = t % LPR;
  double acc = 0.0, diag = 0.0;
  // partitioned problem: only the rows of owned unknowns are assembled completely; ghost rows give 0
  const bool live = row < n && (!owned || owned[row]);
  if (live) {
    const int a = rowptr[row], e = rowptr[row + 1];
    double acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int j = a + l;
    for (; j + 3 * LPR < e; j += 4 * LPR) {
      const int c0 = colind[j], c1 = colind[j + LPR], c2 = colind[j + 2 * LPR], c3 = colind[j + 3 * LPR];
      const double v0 = vals[j], v1 = vals[j + LPR], v2 = vals[j + 2 * LPR], v3 = vals[j + 3 * LPR];
      if (SCALED) {
        acc += v0 * (x[c0] * dinv[c0]); acc1 += v1 * (x[c1] * dinv[c1]);
        acc2 += v2 * (x[c2] * dinv[c2]); acc3 += v3 * (x[c3] * dinv[c3]);
      } else {
        acc += v0 * x[c0]; acc1 += v1 * x[c1]; acc2 += v2 * x[c2]; acc3 += v3 * x[c3];
        diag = c0 == row ? v0 : c1 == row ? v1 : c2 == row ? v2 : c3 == row ? v3 : diag;
      }
    }
    for (; j < e; j += LPR) {
      const int c = colind[j];
      acc += vals[j] * (SCALED ? x[c] * dinv[c] : x[c]);
      if (!SCALED && c == row) diag = vals[j];
    }
    acc = (acc + acc1) + (acc2 + acc3);
  }
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (!SCALED && dinv_out) {   // the inverse diagonal as a by-product of the first residual (diag_inv_kernel's rule)
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) diag += __shfl_xor(diag, m);
    if (row < n && l == 0) dinv_out[row] = diag != 0.0 ? 1.0 / diag : 1.0;
  }
  if (row < n && l == 0) y[row] = live ? ((!SCALED && b) ? b[row] - acc : acc) : 0.0;
}

// a[i] = 0 on ghost entries (partitioned problems)
__global__ void mask_kernel(int n, const uint8_t* __restrict__ owned, double* __restrict__ a, int as_ones) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (as_ones) a[i] = owned[i] ? 1.0 : 0.0;
  else if (!owned[i]) a[i] = 0.0;
}

// ghost rows of the operator become identity rows: the rank-local V-cycle then leaves ghost entries at zero
__global__ void ghost_identity_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ colind,
                                      const uint8_t* __restrict__ owned, double* __restrict__ vals) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n || owned[row]) return;
  for (int j = rowptr[row]; j < rowptr[row + 1]; ++j) vals[j] = colind[j] == row ? 1.0 : 0.0;
}

__global__ void diag_inv_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ colind,
                                const double* __restrict__ vals, double* __restrict__ dinv) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double d = 1.0;
  for (int j = rowptr[row]; j < rowptr[row + 1]; ++j)
    if (colind[j] == row) d = vals[j];
  dinv[row] = d != 0.0 ? 1.0 / d : 1.0;
}

// second-stage scalar algebra of the algorithm step `op`
enum { OP_STORE3 = 0, OP_CG_INIT, OP_CG_PAP, OP_CG_RHO, OP_BI_RHO, OP_BI_ALPHA, OP_BI_OMEGA, OP_MEAN, OP_START };

__device__ __forceinline__ void raise_flag(double* __restrict__ sc, int bit) { sc[S_FLAG] = (double)((int)sc[S_FLAG] | bit); }

__device__ __forceinline__ void apply_dot_op(double* __restrict__ sc, int op, int nd, double v0, double v1, double v2, int d0,
                                             int d1, int d2, double scale) {
  switch (op) {
    case OP_STORE3: sc[d0] = v0; if (nd > 1) sc[d1] = v1; if (nd > 2) sc[d2] = v2; break;
    case OP_MEAN: sc[S_MEAN] = v0 * scale; break;
    case OP_START: sc[S_RR] = v0; sc[S_BB] = v1; sc[S_FLAG] = 0.0; break;                        // r.r, b.b of a new solve
    case OP_CG_INIT: sc[S_RHO] = v0; sc[S_RR] = v1; if (nd > 2) sc[S_BB] = v2; break;          // r.z, r.r[, b.b]
    case OP_CG_PAP:                                                                          // p.Ap
      sc[S_PAP] = v0;
      if (v0 != 0.0) sc[S_ALPHA] = sc[S_RHO] / v0;
      else { sc[S_ALPHA] = 0.0; raise_flag(sc, F_PAP_ZERO); }                                // p = 0: nothing left to correct
      break;
    case OP_CG_RHO:                                                                          // r.z, r.r
      sc[S_BETA] = sc[S_RHO] != 0.0 ? v0 / sc[S_RHO] : 0.0;
      sc[S_RHO] = v0; sc[S_RR] = v1;
      break;
    case OP_BI_RHO:                                                                          // rhat.r
      // rho_(i-1) = 0 or omega_(i-1) = 0 with a residual left is BiCGStab's breakdown: beta = 0 restarts the recurrence
      // from p = r (the host re-bases rhat as well when it sees the flag)
      if (sc[S_RHO] != 0.0 && sc[S_OMEGA] != 0.0) sc[S_BETA] = (v0 / sc[S_RHO]) * (sc[S_ALPHA] / sc[S_OMEGA]);
      else { sc[S_BETA] = 0.0; raise_flag(sc, sc[S_RHO] != 0.0 ? F_OMEGA_ZERO : F_RHO_ZERO); }
      sc[S_RHO] = v0;
      break;
    case OP_BI_ALPHA:                                                                        // rhat.v
      sc[S_RV] = v0;
      if (v0 != 0.0) sc[S_ALPHA] = sc[S_RHO] / v0;
      else { sc[S_ALPHA] = 0.0; raise_flag(sc, F_RV_ZERO); }
      break;
    case OP_BI_OMEGA:                                                                        // t.s, t.t
      sc[S_TS] = v0; sc[S_TT] = v1;
      sc[S_OMEGA] = v1 != 0.0 ? v0 / v1 : 0.0;      // t = 0 <=> s = 0: the half step x += alpha p was exact, r = s
      break;
  }
}

// Up to three dot products and the scalar update that consumes them in ONE launch: every block writes its
// partial sums, the last block to finish (ticket counter) adds the partials in a fixed order and applies
// `op` to the device scalars.  The summation order does not depend on which block is last, so the solves stay
// bit-reproducible.
__global__ __launch_bounds__(RED_THREADS) void dots_kernel(int n, int nd, const double* a0, const double* b0,
                                                           const double* a1, const double* b1, const double* a2,
                                                           const double* b2, double* partial, unsigned* ticket,
                                                           double* __restrict__ sc, int op, int d0, int d1, int d2,
                                                           double scale, double* __restrict__ red = nullptr) {
  __shared__ double sh[3][RED_THREADS];
  __shared__ bool last;
  double s0 = 0, s1 = 0, s2 = 0;
  const int stride = gridDim.x * blockDim.x;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {     // four independent loads per vector in flight
    const double x0 = a0[i], x1 = a0[i + stride], x2 = a0[i + 2 * stride], x3 = a0[i + 3 * stride];
    const double y0 = b0[i], y1 = b0[i + stride], y2 = b0[i + 2 * stride], y3 = b0[i + 3 * stride];
    s0 += x0 * y0; s0 += x1 * y1; s0 += x2 * y2; s0 += x3 * y3;
    if (nd > 1) { s1 += a1[i] * b1[i]; s1 += a1[i + stride] * b1[i + stride]; s1 += a1[i + 2 * stride] * b1[i + 2 * stride]; s1 += a1[i + 3 * stride] * b1[i + 3 * stride]; }
    if (nd > 2) { s2 += a2[i] * b2[i]; s2 += a2[i + stride] * b2[i + stride]; s2 += a2[i + 2 * stride] * b2[i + 2 * stride]; s2 += a2[i + 3 * stride] * b2[i + 3 * stride]; }
  }
  for (; i < n; i += stride) {
    s0 += a0[i] * b0[i];
    if (nd > 1) s1 += a1[i] * b1[i];
    if (nd > 2) s2 += a2[i] * b2[i];
  }
  sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2;
  __syncthreads();
  for (int w = RED_THREADS / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w)
      for (int k = 0; k < nd; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int k = 0; k < nd; ++k)
      __hip_atomic_store(&partial[k * RED_BLOCKS + blockIdx.x], sh[k][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  const int nb = gridDim.x;
  for (int k = 0; k < nd; ++k) {
    double s = 0;
    for (int i = threadIdx.x; i < nb; i += RED_THREADS)
      s += __hip_atomic_load(&partial[k * RED_BLOCKS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[k][threadIdx.x] = s;
  }
  __syncthreads();
  for (int w = RED_THREADS / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w)
      for (int k = 0; k < nd; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  *ticket = 0;
  const double v0 = sh[0][0], v1 = nd > 1 ? sh[1][0] : 0.0, v2 = nd > 2 ? sh[2][0] : 0.0;
  if (red) {   // partitioned problem: the sums of this rank; the scalar algebra follows the all-reduce (dots_apply_kernel)
    red[0] = v0; red[1] = v1; red[2] = v2;
    return;
  }
  apply_dot_op(sc, op, nd, v0, v1, v2, d0, d1, d2, scale);
}

__global__ void dots_apply_kernel(double* __restrict__ sc, const double* __restrict__ red, int op, int nd, int d0, int d1,
                                  int d2, double scale) {
  apply_dot_op(sc, op, nd, red[0], nd > 1 ? red[1] : 0.0, nd > 2 ? red[2] : 0.0, d0, d1, d2, scale);
}


enum { V_RESID = 0, V_SHIFT, V_CG_XR, V_CG_P, V_JACOBI, V_BI_P, V_BI_S, V_COPY, V_COPY_SHIFT };

// small fused vector updates; scalars are read from device memory
__global__ void vec_kernel(int n, int op, const double* __restrict__ sc, double* __restrict__ a, double* __restrict__ b,
                           const double* __restrict__ c, const double* __restrict__ d, const double* __restrict__ e) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (op) {
    case V_RESID: a[i] = c[i] - d[i]; break;                                   // r = b - Ax
    case V_SHIFT: a[i] -= sc[S_MEAN]; break;                                   // a -= mean
    case V_CG_XR: a[i] += sc[S_ALPHA] * c[i]; b[i] -= sc[S_ALPHA] * d[i]; break;   // x += alpha p; r -= alpha q
    case V_JACOBI: a[i] = c[i] * d[i]; break;                                  // z = r .* dinv
    case V_CG_P: a[i] = c[i] + sc[S_BETA] * a[i]; break;                       // p = z + beta p
    case V_BI_P: a[i] = c[i] + sc[S_BETA] * (a[i] - sc[S_OMEGA] * d[i]); break;  // p = r + beta (p - omega v)
    case V_BI_S: a[i] = c[i] - sc[S_ALPHA] * d[i]; break;                      // s = r - alpha v
    case V_COPY: a[i] = c[i]; break;
    case V_COPY_SHIFT: a[i] = c[i] - sc[S_MEAN]; break;                        // copy and shift in one pass
  }
}

// dst[i * stride] = src[i] - mean: the solution made orthogonal to the constants on its way into the vertex records
__global__ void scatter_shift_kernel(const double* __restrict__ src, double* __restrict__ dst, int n, int stride,
                                     const double* __restrict__ sc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(size_t)i * stride] = src[i] - sc[S_MEAN];
}

// dinv != NULL: p, s are the unpreconditioned directions (Jacobi applied here); else they are M^-1 p, M^-1 s
__global__ void bicg_update_kernel(int n, const double* __restrict__ sc, double* __restrict__ x, double* __restrict__ r,
                                   const double* __restrict__ p, const double* __restrict__ s, const double* __restrict__ sres,
                                   const double* __restrict__ t, const double* __restrict__ dinv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = dinv ? dinv[i] : 1.0;
  x[i] += sc[S_ALPHA] * (p[i] * d) + sc[S_OMEGA] * (s[i] * d);
  r[i] = sres[i] - sc[S_OMEGA] * t[i];
}

// KNP unknowns: `csol` is ion-major over the global vertex numbering ([k][g]); the solver works in the
// reference's block order [sub-domain][ion][vertex] (pdeSolver.py:117).  to_blocks: x = csol, else csol = x.
__global__ void knp_order_kernel(int ntot, int ks, int n_sub, const KnConsts* __restrict__ Cp, double* __restrict__ x,
                                 double* __restrict__ csol, int to_blocks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ks * ntot) return;
  const int k = i / ntot, g = i - k * ntot;
  int sd = 0;
  for (int t = 1; t < n_sub; ++t) sd += g >= Cp->voff[t];
  const int v0 = Cp->voff[sd], nv = Cp->voff[sd + 1] - v0;
  const size_t xb = (size_t)ks * v0 + (size_t)k * nv + (g - v0);
  if (to_blocks) x[xb] = csol[i];
  else csol[i] = x[xb];
}

// cur <- 2 cur - old (linear extrapolation of the last two solutions), old <- the value cur had; `have` = 0: only store.
// have >= 2 with old2 != NULL: quadratic extrapolation 3 cur - 3 old + old2 through the last three solutions.
__global__ void extrapolate_kernel(int n, double* __restrict__ cur, int stride, double* __restrict__ old, double* __restrict__ old2,
                                   int have) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double c = cur[(size_t)i * stride];
  if (have >= 2 && old2) cur[(size_t)i * stride] = 3.0 * (c - old[i]) + old2[i];
  else if (have >= 1) cur[(size_t)i * stride] = 2.0 * c - old[i];
  if (old2) old2[i] = old[i];
  old[i] = c;
}

// ---- coarse space of the distributed EMI preconditioner ---------------------------------------------------------
// Coarse functions (knpemi_set_distributed_coarse): per rank and sub-domain the k + 1 hat functions over k slices of the
// owned vertices' extent along its longest axis (piecewise linear along the axis, a partition of unity on the owned
// vertices; the two half hats that meet at a cut between ranks are separate functions, so the space contains the
// continuous piecewise-linear functions across the cuts).  agg_of[i] = the lower node of unknown i (-1: ghost), agg_w[i]
// its weight in the upper node; per node the list of its unknowns and their weights (agg_ptr / agg_idx / agg_wt).
// restriction: one workgroup per local node sums the weighted r over its list in a fixed order (reproducible) and writes
// its entry of the coarse vector in the reduction buffer; the all-reduce fills in the other ranks' entries.
__global__ __launch_bounds__(256) void coarse_restrict_kernel(const double* __restrict__ r, const int* __restrict__ agg_ptr,
                                                              const int* __restrict__ agg_idx, const double* __restrict__ agg_wt,
                                                              int nl, int rank, int world, double* __restrict__ red) {
  __shared__ double sh[256];
  const int a = blockIdx.x;
  double acc = 0.0;
  for (int t = agg_ptr[a] + threadIdx.x; t < agg_ptr[a + 1]; t += 256) acc += agg_wt[t] * r[agg_idx[t]];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) sh[threadIdx.x] += sh[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) red[KN_COARSE_OFF + rank * nl + a] = sh[0];
  if (a == 0)   // the other ranks' entries start from zero
    for (int j = threadIdx.x; j < world * nl; j += 256)
      if (j / nl != rank) red[KN_COARSE_OFF + j] = 0.0;
}

// z_c = A_c^+ r_c, this rank's rows only
__global__ void coarse_solve_kernel(const double* __restrict__ inv, const double* __restrict__ red, int nl, int rank, int nc,
                                    double* __restrict__ zc) {
  const int a = threadIdx.x;
  if (a >= nl) return;
  const double* row = inv + (size_t)(rank * nl + a) * nc;
  double v = 0.0;
  for (int j = 0; j < nc; ++j) v += row[j] * red[KN_COARSE_OFF + j];
  zc[a] = v;
}

// prolongation: z += (1 - w) z_c[node] + w z_c[node + 1] on the owned unknowns (mode 0); mode 1: z = coarse function `pick`
__global__ void coarse_prolong_kernel(int n, const int* __restrict__ agg_of, const double* __restrict__ agg_w,
                                      const double* __restrict__ zc, double* __restrict__ z, int mode, int pick) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = agg_of[i];
  const double w = a >= 0 ? agg_w[i] : 0.0;
  // w == 0 (one node per rank and sub-domain, or a vertex on the lower node itself): zc[a + 1] may lie past the nl entries
  // coarse_solve_kernel writes -- it must not be read, 0 * (stale NaN) would poison z
  if (mode == 0) { if (a >= 0) z[i] += w > 0.0 ? (1.0 - w) * zc[a] + w * zc[a + 1] : zc[a]; }
  else z[i] = a < 0 ? 0.0 : (a == pick ? 1.0 - w : (a + 1 == pick ? w : 0.0));
}

struct Ctx {
  knpemi_handle* h;
  int n;
  const int* rowptr; const int* colind; const double* vals;
  double* sc; double* partial;
  // partitioned problem (knpemi_set_distributed): mask of owned unknowns of this system, which system
  const uint8_t* owned = nullptr;
  int which = 0;
  int comm_rc = 0;     // first failure of a communication hook
  int lpr = 16;        // lanes per row of the SpMV (spmv_lanes)
};

// lanes per row from the average row length (read once per system and handle)
int spmv_lanes(knpemi_handle* h, int which, const int* rowptr, int n) {
  int& l = h->spmv_lpr[which == KNPEMI_B_KNP ? 1 : 0];
  if (l == 0) {
    int nnz = 0;
    if (hipMemcpy(&nnz, rowptr + n, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return 16;
    l = n > 0 && nnz / n <= 24 ? 4 : 16;
  }
  return l;
}

inline dim3 grid1(int n) { return dim3((n + 255) / 256); }

void spmv(Ctx& c, const double* x, double* y, const double* dinv, const double* b = nullptr, double* dinv_out = nullptr) {
  dim3 g(((size_t)c.n * c.lpr + 255) / 256);
  if (c.owned) {   // the argument's ghost entries take their owners' values first
    const KnDist& d = c.h->dist;
    if (int e = d.halo(d.ctx, const_cast<double*>(x), c.which)) c.comm_rc = c.comm_rc ? c.comm_rc : e;
  }
  if (c.h->bcols.bcol && !dinv && !dinv_out) {   // DG systems: block columns instead of one column index per entry
    launch_block_spmv<double>(c.h->stream, c.h->bcols, c.n, c.rowptr, c.vals, x, b, y, c.owned);
    return;
  }
  if (c.lpr == 4) {
    if (dinv) hipLaunchKernelGGL((spmv_kernel<true, 4>), g, dim3(256), 0, c.h->stream, c.n, c.rowptr, c.colind, c.vals, x, dinv, y,
                                 (const double*)nullptr, c.owned);
    else hipLaunchKernelGGL((spmv_kernel<false, 4>), g, dim3(256), 0, c.h->stream, c.n, c.rowptr, c.colind, c.vals, x, dinv, y, b,
                            c.owned, dinv_out);
    return;
  }
  if (dinv) hipLaunchKernelGGL((spmv_kernel<true, 16>), g, dim3(256), 0, c.h->stream, c.n, c.rowptr, c.colind, c.vals, x, dinv, y,
                               (const double*)nullptr, c.owned);
  else hipLaunchKernelGGL((spmv_kernel<false, 16>), g, dim3(256), 0, c.h->stream, c.n, c.rowptr, c.colind, c.vals, x, dinv, y, b,
                          c.owned, dinv_out);
}

void dots(Ctx& c, int nd, const double* a0, const double* b0, const double* a1, const double* b1, const double* a2,
          const double* b2, int op, int d0 = 0, int d1 = 0, int d2 = 0, double scale = 1.0) {
  // 16 entries per thread: every block costs ~50 ns at the ticket counter, whatever it sums
  const int nb = std::min(RED_BLOCKS, (c.n + 16 * RED_THREADS - 1) / (16 * RED_THREADS));
  double* red = c.owned ? c.h->dist.d_red : nullptr;
  hipLaunchKernelGGL(dots_kernel, dim3(nb), dim3(RED_THREADS), 0, c.h->stream, c.n, nd, a0, b0, a1, b1, a2, b2, c.partial,
                     reinterpret_cast<unsigned*>(c.partial + 3 * RED_BLOCKS), c.sc, op, d0, d1, d2, scale, red);
  if (red) {   // every vector is zero on its ghost entries, so the local sums run over the owned unknowns only
    const KnDist& d = c.h->dist;
    if (int e = d.allreduce(d.ctx, nd)) c.comm_rc = c.comm_rc ? c.comm_rc : e;
    hipLaunchKernelGGL(dots_apply_kernel, dim3(1), dim3(1), 0, c.h->stream, c.sc, red, op, nd, d0, d1, d2, scale);
  }
}

void mask(const Ctx& c, double* a, int as_ones = 0) {
  if (c.owned) hipLaunchKernelGGL(mask_kernel, grid1(c.n), dim3(256), 0, c.h->stream, c.n, c.owned, a, as_ones);
}

void vec(const Ctx& c, int op, double* a, double* b, const double* cc, const double* d) {
  hipLaunchKernelGGL(vec_kernel, grid1(c.n), dim3(256), 0, c.h->stream, c.n, op, c.sc, a, b, cc, d, (const double*)nullptr);
}

// the solver's scalars on the host: through a pinned buffer (a copy into pageable memory is staged by the runtime)
int read_scalars(const Ctx& c, double* host, int count) {
  knpemi_handle* h = c.h;
  if (!h->kry_pinned) KN_HIP(hipHostMalloc(&h->kry_pinned, 64 * sizeof(double), hipHostMallocDefault));
  KN_HIP(hipMemcpyAsync(h->kry_pinned, c.sc, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < count; ++i) host[i] = static_cast<double*>(h->kry_pinned)[i];
  return KNPEMI_OK;
}

// BiCGStab start: p = v = 0, rho = alpha = omega = 1, everything else 0 (one launch instead of two fills and a copy)
__global__ void bicg_init_kernel(int n, double* __restrict__ p, double* __restrict__ v, double* __restrict__ sc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { p[i] = 0.0; v[i] = 0.0; }
  if (i < S_N) sc[i] = (i == S_RHO || i == S_ALPHA || i == S_OMEGA) ? 1.0 : 0.0;
}

// Run `chunk` iterations of `body`: captured once into a hipGraph (the launch-bound inner loop of small
// systems: ~20 launches of a few microseconds per iteration) and replayed; the graph is keyed on everything its
// kernel arguments depend on.  KNPEMI_NO_GRAPH=1 launches the kernels directly.
template <class Body>
int run_chunk(knpemi_handle* h, knpemi_handle::KnGraph& g, uint64_t key, int chunk, Body&& body) {
  static const bool no_graph_env = getenv("KNPEMI_NO_GRAPH") != nullptr;
  const bool no_graph = no_graph_env || h->dist.on;   // the communication hooks cannot be captured
  int rc = KNPEMI_OK;
  if (no_graph) {
    for (int k = 0; k < chunk && !rc; ++k) rc = body();
    return rc;
  }
  if (!g.exec || g.key != key) {
    if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
    hipGraph_t graph = nullptr;
    KN_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < chunk && !rc; ++k) rc = body();
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) { kn_set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
    e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { g.exec = nullptr; kn_set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
    g.key = key;
  }
  KN_HIP(hipGraphLaunch(g.exec, h->stream));
  return KNPEMI_OK;
}

// KNPEMI_NO_FUSED=1: the plain loops (one launch per operation) instead of kernels_fused.hip
inline bool use_fused() {
  static const bool off = getenv("KNPEMI_NO_FUSED") != nullptr;
  return !off;
}

// KNPEMI_AMG_REBUILD_EVERY=k (test hook): the hierarchy is declared aged at every k-th solve of its system
inline int rebuild_every() {
  static const int k = getenv("KNPEMI_AMG_REBUILD_EVERY") ? atoi(getenv("KNPEMI_AMG_REBUILD_EVERY")) : 0;
  return k;
}

inline bool debug_krylov() {
  static const bool on = getenv("KNPEMI_DEBUG_KRYLOV") != nullptr;
  return on;
}

inline uint64_t graph_key(const knpemi_handle* h, const KnAmg& G, bool amg, int chunk, int n) {
  uint64_t k = reinterpret_cast<uint64_t>(h->kry);
  k = k * 1000003u + (uint64_t)G.builds;
  k = k * 1000003u + (amg ? 1u : 0u);
  k = k * 1000003u + (uint64_t)chunk;
  k = k * 1000003u + (uint64_t)n;
  return k | 1u;
}

// the solver's scalars for an error message
std::string describe_scalars(const double* sc, int it) {
  char buf[320];
  snprintf(buf, sizeof buf,
           "[iteration %d: |r|^2 = %.6e, |b|^2 = %.6e, rho = %.6e, alpha = %.6e, beta = %.6e, omega = %.6e, rhat.v = %.6e, "
           "t.s = %.6e, t.t = %.6e, p.Ap = %.6e, breakdown flags = %d]",
           it, sc[S_RR], sc[S_BB], sc[S_RHO], sc[S_ALPHA], sc[S_BETA], sc[S_OMEGA], sc[S_RV], sc[S_TS], sc[S_TT], sc[S_PAP],
           (int)sc[S_FLAG]);
  return buf;
}

// A solve that starts from non-finite data is an input error, not a breakdown: say which operand it is before the first
// iteration has overwritten the iterate (|b| from the right-hand side alone; |r0| = |b - A x0| brings in the matrix and
// the initial guess).
int check_start(const char* who, const double* sc) {
  if (!std::isfinite(sc[S_BB])) {
    kn_set_error(std::string(who) + ": the right-hand side contains non-finite values");
    return KNPEMI_EINVAL;
  }
  if (!std::isfinite(sc[S_RR])) {
    kn_set_error(std::string(who) + ": the initial residual b - A x0 is not finite (matrix values or initial guess)");
    return KNPEMI_EINVAL;
  }
  return KNPEMI_OK;
}

}  // namespace

// Workspace: 10 vectors of the larger system + ones + scalars + partials (allocated on first use).
static int ensure_work(knpemi_handle* h, size_t n) {
  if (h->kry_n >= n) return KNPEMI_OK;
  void* p = nullptr;
  const size_t doubles = 11 * n + 64 + 3 * RED_BLOCKS + 2;   // + ticket counter of dots_kernel
  KN_HIP(hipMalloc(&p, doubles * sizeof(double)));
  h->allocs.push_back(p);
  KN_HIP(hipMemsetAsync(p, 0, doubles * sizeof(double), h->stream));
  h->kry = static_cast<double*>(p);
  h->kry_n = n;
  std::vector<double> ones(n, 1.0);
  KN_HIP(hipMemcpyAsync(h->kry + 10 * n, ones.data(), n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

// r_c = Phi^T r into the reduction buffer (all-reduced): every rank ends up with the whole coarse vector
static void coarse_restrict(Ctx& c, const double* r) {
  knpemi_handle* h = c.h;
  KnDist& d = h->dist;
  hipLaunchKernelGGL(coarse_restrict_kernel, dim3(d.nl), dim3(256), 0, h->stream, r, d.d_agg_ptr, d.d_agg_idx, d.d_agg_wt, d.nl,
                     d.rank, d.world, d.d_red);
  if (int e = d.allreduce(d.ctx, KN_COARSE_OFF + d.nc)) c.comm_rc = c.comm_rc ? c.comm_rc : e;
}

// Coarse operator A_c = Phi^T A Phi of the partitioned EMI system (one column per coarse function: an SpMV of it and a
// restriction), then on the host the inverse of A_c + alpha 1 1^T (A_c has the constants in its kernel;
// the residuals it is applied to have zero mean) with empty aggregates decoupled.
static int coarse_setup(Ctx& c, double* work_phi, double* work_y) {
  knpemi_handle* h = c.h;
  KnDist& d = h->dist;
  const int nc = d.nc, nl = d.nl;
  if (!d.d_coarse_inv) {
    void* p = nullptr;
    KN_HIP(hipMalloc(&p, (size_t)KN_COARSE_MAX * KN_COARSE_MAX * sizeof(double))); h->allocs.push_back(p); d.d_coarse_inv = static_cast<double*>(p);
    KN_HIP(hipMalloc(&p, (KN_COARSE_MAX + 1) * sizeof(double))); h->allocs.push_back(p); d.d_coarse_z = static_cast<double*>(p);
    // (KNPEMI_DEBUG_POISON_COARSE: all-ones bytes = NaN, for the test that no entry past the nl written ones is read)
    KN_HIP(hipMemset(d.d_coarse_z, getenv("KNPEMI_DEBUG_POISON_COARSE") ? 0xFF : 0, (KN_COARSE_MAX + 1) * sizeof(double)));
  }
  std::vector<double> Ac((size_t)nc * nc, 0.0), col(nc);
  for (int j = 0; j < nc; ++j) {
    const int rj = j / nl, aj = j % nl;
    // phi_j: 1 on the owned unknowns of aggregate aj of rank rj (ghost copies are filled by the SpMV's halo), 0 elsewhere
    hipLaunchKernelGGL(coarse_prolong_kernel, grid1(c.n), dim3(256), 0, h->stream, c.n, d.d_agg_of, d.d_agg_w,
                       (const double*)nullptr, work_phi, 1, rj == d.rank ? aj : -2);
    spmv(c, work_phi, work_y, nullptr);
    coarse_restrict(c, work_y);
    KN_HIP(hipMemcpyAsync(col.data(), d.d_red + KN_COARSE_OFF, nc * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
    if (c.comm_rc) { kn_set_error("coarse space set-up: a communication hook failed"); return KNPEMI_EHIP; }
    for (int i = 0; i < nc; ++i) Ac[(size_t)i * nc + j] = col[i];
  }
  // symmetrise (rounding), decouple empty aggregates, shift the constants' kernel, invert (Gauss-Jordan, nc <= 64)
  double tr = 0.0;
  for (int i = 0; i < nc; ++i) tr += Ac[(size_t)i * nc + i];
  const double alpha = tr > 0 ? tr / ((double)nc * nc) : 1.0;
  std::vector<double> M((size_t)nc * nc), inv((size_t)nc * nc, 0.0);
  std::vector<char> empty(nc, 0);
  for (int i = 0; i < nc; ++i) empty[i] = !(Ac[(size_t)i * nc + i] > 0.0);
  for (int i = 0; i < nc; ++i)
    for (int j = 0; j < nc; ++j) {
      double v = 0.5 * (Ac[(size_t)i * nc + j] + Ac[(size_t)j * nc + i]);
      if (empty[i] || empty[j]) v = i == j ? 1.0 : 0.0;
      else v += alpha;
      M[(size_t)i * nc + j] = v;
    }
  for (int i = 0; i < nc; ++i) inv[(size_t)i * nc + i] = 1.0;
  for (int k = 0; k < nc; ++k) {
    int piv = k;
    for (int i = k + 1; i < nc; ++i) if (std::fabs(M[(size_t)i * nc + k]) > std::fabs(M[(size_t)piv * nc + k])) piv = i;
    if (!(std::fabs(M[(size_t)piv * nc + k]) > 0.0)) { kn_set_error("coarse space set-up: singular coarse operator"); return KNPEMI_EINVAL; }
    if (piv != k)
      for (int j = 0; j < nc; ++j) { std::swap(M[(size_t)k * nc + j], M[(size_t)piv * nc + j]); std::swap(inv[(size_t)k * nc + j], inv[(size_t)piv * nc + j]); }
    const double dk = 1.0 / M[(size_t)k * nc + k];
    for (int j = 0; j < nc; ++j) { M[(size_t)k * nc + j] *= dk; inv[(size_t)k * nc + j] *= dk; }
    for (int i = 0; i < nc; ++i) {
      if (i == k) continue;
      const double f = M[(size_t)i * nc + k];
      if (f == 0.0) continue;
      for (int j = 0; j < nc; ++j) { M[(size_t)i * nc + j] -= f * M[(size_t)k * nc + j]; inv[(size_t)i * nc + j] -= f * inv[(size_t)k * nc + j]; }
    }
  }
  for (int i = 0; i < nc; ++i)
    if (empty[i]) for (int j = 0; j < nc; ++j) inv[(size_t)i * nc + j] = inv[(size_t)j * nc + i] = 0.0;   // nothing to correct there
  KN_HIP(hipMemcpy(d.d_coarse_inv, inv.data(), inv.size() * sizeof(double), hipMemcpyHostToDevice));
  d.coarse_built = true;
  return KNPEMI_OK;
}

// z += Phi A_c^+ Phi^T r  (additive two-level correction next to the per-rank AMG)
static void coarse_correct(Ctx& c, const double* r, double* z) {
  knpemi_handle* h = c.h;
  KnDist& d = h->dist;
  coarse_restrict(c, r);
  hipLaunchKernelGGL(coarse_solve_kernel, dim3(1), dim3(KN_COARSE_MAX), 0, h->stream, d.d_coarse_inv, d.d_red, d.nl, d.rank, d.nc,
                     d.d_coarse_z);
  hipLaunchKernelGGL(coarse_prolong_kernel, grid1(c.n), dim3(256), 0, h->stream, c.n, d.d_agg_of, d.d_agg_w, d.d_coarse_z, z, 0, 0);
}

// Jacobi-PCG on A_emi x = b_emi, x = phi (record component 7, gathered into a contiguous vector first).
int kn_solve_emi(knpemi_handle* h, double rtol, double atol, int maxit, int* iters, double* relres) {
  KnDev& D = h->dev;
  const int n = D.Ntot;
  int rc = ensure_work(h, (size_t)std::max(1, h->K - 1) * D.Ntot);
  if (rc) return rc;
  const size_t N = h->kry_n;
  double *x = h->kry, *r = x + N, *z = r + N, *p = z + N, *q = p + N, *b = q + N, *dinv = b + N, *ones = h->kry + 10 * N;
  Ctx c{h, n, D.rowptr, D.colind, D.A_emi, h->kry + 11 * N, h->kry + 11 * N + 64};
  c.lpr = spmv_lanes(h, KNPEMI_B_EMI, D.rowptr, n);
  const KnDist& dist = h->dist;
  bool has_ghosts = false;
  double n_mean = (double)n;
  if (dist.on) {   // partitioned problem: this rank's rows of the global system
    c.owned = dist.d_owned_emi;
    c.which = KNPEMI_B_EMI;
    n_mean = dist.n_owned_global;
    for (uint8_t o : dist.h_owned_emi) has_ghosts |= !o;
    hipLaunchKernelGGL(ghost_identity_kernel, grid1(n), dim3(256), 0, h->stream, n, c.rowptr, c.colind, c.owned, D.A_emi);
  }
  if (h->kry_ones_masked != (dist.on ? 1 : 0)) {   // `ones` counts the owned unknowns
    if (dist.on) mask(c, ones, 1);
    else {
      std::vector<double> one(n, 1.0);
      KN_HIP(hipMemcpy(ones, one.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    }
    h->kry_ones_masked = dist.on ? 1 : 0;
  }
  // x0 = current phi (ksp_initial_guess_nonzero), b = b_emi projected onto zero mean (constant null space).  With the
  // fused loop these launches open the captured graph of the solve's first chunk (kernels_fused.hip), hence the lambda.
  auto pre = [&]() -> int {
    if (int e = kn_launch_field_gather(h, D.VR + 7, KN_REC, x, n)) return e;
    if (dist.on) {
      vec(c, V_COPY, b, nullptr, D.b_emi, nullptr);
      mask(c, b);
      dots(c, 1, b, ones, nullptr, nullptr, nullptr, nullptr, OP_MEAN, 0, 0, 0, 1.0 / n_mean);
      vec(c, V_SHIFT, b, nullptr, nullptr, nullptr);
      mask(c, b);
    } else {   // single rank: the mean of b_emi itself, then copy and shift in one pass (same arithmetic)
      dots(c, 1, D.b_emi, ones, nullptr, nullptr, nullptr, nullptr, OP_MEAN, 0, 0, 0, 1.0 / n_mean);
      vec(c, V_COPY_SHIFT, b, nullptr, D.b_emi, nullptr);
    }
    // (single rank: the inverse diagonal comes out of the first residual below)
    if (dist.on) hipLaunchKernelGGL(diag_inv_kernel, grid1(n), dim3(256), 0, h->stream, n, c.rowptr, c.colind, c.vals, dinv);
    return KNPEMI_OK;
  };
  // solution orthogonal to constants, then into the phi component of the vertex records (ghosts included: they take
  // their owners' values first); idempotent
  auto post = [&]() -> int {
    h->gam_valid = false;      // the potential changes (knpemi_handle::gam_valid)
    dots(c, 1, x, ones, nullptr, nullptr, nullptr, nullptr, OP_MEAN, 0, 0, 0, 1.0 / n_mean);
    if (dist.on) {
      if (int e = dist.halo(dist.ctx, x, KNPEMI_B_EMI)) c.comm_rc = c.comm_rc ? c.comm_rc : e;
      vec(c, V_SHIFT, x, nullptr, nullptr, nullptr);
      if (int e = kn_launch_field_scatter(h, x, D.VR + 7, n, KN_REC)) return e;
    } else if (n > 0) {
      hipLaunchKernelGGL(scatter_shift_kernel, grid1(n), dim3(256), 0, h->stream, x, D.VR + 7, n, KN_REC, c.sc);
    }
    return KNPEMI_OK;
  };
  KnAmg& G = h->amg_emi;
  const bool amg = h->pc_emi == KNPEMI_PC_AMG;
  if (amg && (!G.built || G.n != n)) {
    G.negative_strength = true;
    // stretched Q1 cells couple the two ends of an edge along the long direction positively: aggregate_apart keeps them
    // in different aggregates (config 2h: 9.35 / 3.95 -> 7.4 / 2.9 iterations per solve; on simplices the plain greedy
    // pass is the better one, 5.3 / 2.35 against 5.75 / 2.6 at config 2).  The DG solver handle sets the flag itself.
    // (plain_knp marks the solver handle of a DG problem, which has made its own choices: kernels_dg.hip)
    if (h->plain_knp) {}
    else if (h->NV == 8) G.positive_conflict = true;
    // prolongator smoothing along the large couplings only (KnAmg::filter_theta): the operator complexity falls from
    // 2.5-2.8 to 1.5 on simplices, the iteration counts stay (995 k tets: 6.65 / 2.8 -> 6.75 / 2.85, 3.34 -> 2.37 ms per
    // step with solves).  "Large" is by magnitude: the big positive entries of stretched Q1 cells stay in the smoothing
    // (lumping them into the diagonal cost config 2h 7.4 -> 12.6 iterations).
    if (!h->plain_knp) G.filter_theta = 0.02;
    if (getenv("KNPEMI_AMG_APART")) G.positive_conflict = atoi(getenv("KNPEMI_AMG_APART")) != 0;
    if (const char* ft = getenv("KNPEMI_AMG_FILTER")) G.filter_theta = atof(ft);
    G.want_fused = !dist.on && use_fused();
    G.want_cycle = dist.on && use_fused() && !getenv("KNPEMI_DIST_PLAIN_CYCLE");
    // the diagonal block of a rank that has ghosts has lost couplings: it is non-singular
    if ((rc = kn_amg_setup(h, G, n, D.rowptr, D.colind, D.A_emi, !has_ghosts,
                           dist.on ? dist.h_owned_emi.data() : nullptr))) return rc;
    G.its_ref = -1;
    ++G.builds;
  } else if (amg && !dist.on) {
    if (rebuild_every() > 0 && ++G.solves % rebuild_every() == 0) G.rebuild_wanted = true;
    if (G.rebuild_wanted && (rc = kn_amg_rebuild_step(h, G, n, D.rowptr, D.colind, D.A_emi, !has_ghosts))) return rc;
  }
  if (amg && (rc = kn_amg_refresh(h, G, D.A_emi))) return rc;
  // two-level variant on a partitioned mesh: the ranks' AMG cycles do not see each other, a coarse space of one
  // constant per (rank, sub-domain) carries the error across the cuts (knpemi_set_distributed_coarse)
  const bool coarse = dist.on && amg && h->dist.nc > 0;
  if (coarse && (!h->dist.coarse_built || G.its_ref < 0)) {
    if ((rc = coarse_setup(c, p, q))) return rc;               // p, q are free before the first search direction
  }
  auto precond = [&]() -> int {
    if (amg) {
      int e = kn_amg_apply(h, G, D.A_emi, dinv, r, q, z);      // q = A p is free here
      if (e) return e;
      if (coarse) coarse_correct(c, r, z);
      return KNPEMI_OK;
    }
    vec(c, V_JACOBI, z, nullptr, r, dinv);
    return KNPEMI_OK;
  };
  const int chunk = amg ? 1 : 8;   // a V-cycle costs ~15 launches: test convergence after every iteration
  double sc[S_N];
  const bool fused = amg && G.fused_ok && !dist.on && use_fused();
  int it = 0;
  if (fused) {   // residual, target and the loop in 2 + (2 levels - 1) launches per iteration (kernels_fused.hip)
    const KnFusedSys S{n, c.rowptr, c.colind, c.vals, c.sc, h->kry, N};
    if ((rc = kn_fused_cg(h, G, S, D.b_emi, rtol, atol, maxit, &it, &sc[S_RR], &sc[S_BB], D.VR + 7, KN_REC))) return rc;
  } else {
    if ((rc = pre())) return rc;
    spmv(c, x, r, nullptr, b, dist.on ? nullptr : dinv);         // r = b - A x
    dots(c, 2, r, r, b, b, nullptr, nullptr, OP_START);
    if ((rc = read_scalars(c, sc, S_N))) return rc;
    if ((rc = check_start("EMI CG", sc))) return rc;
  }
  const double bnorm = std::sqrt(sc[S_BB]);
  const double target = std::max(atol, rtol * (bnorm > 0 ? bnorm : 1.0));
  double rn = std::sqrt(sc[S_RR]);
  if (!fused && rn > target) {   // the (extrapolated) initial guess is not good enough: first search direction
    if ((rc = precond())) return rc;
    vec(c, V_COPY, p, nullptr, z, nullptr);
    dots(c, 2, r, z, r, r, nullptr, nullptr, OP_CG_INIT);      // r.z, r.r
  }
  auto iteration = [&]() -> int {
    spmv(c, p, q, nullptr);
    dots(c, 1, p, q, nullptr, nullptr, nullptr, nullptr, OP_CG_PAP);
    vec(c, V_CG_XR, x, r, p, q);
    if (int e = precond()) return e;
    dots(c, 2, r, z, r, r, nullptr, nullptr, OP_CG_RHO);
    vec(c, V_CG_P, p, nullptr, z, nullptr);
    return KNPEMI_OK;
  };
  const uint64_t gkey = graph_key(h, G, amg, chunk, n);
  while (!fused && rn > target && it < maxit) {
    const int todo = std::min(chunk, maxit - it);
    if (todo == chunk) { if ((rc = run_chunk(h, h->graph_emi, gkey, chunk, iteration))) return rc; }
    else for (int k = 0; k < todo; ++k) if ((rc = iteration())) return rc;
    it += todo;
    if ((rc = read_scalars(c, sc, S_N))) return rc;
    rn = std::sqrt(sc[S_RR]);
    if (debug_krylov()) fprintf(stderr, "[knpemi] emi cg it %d |r| %.3e target %.3e alpha %.3e beta %.3e\n", it, rn, target, sc[S_ALPHA], sc[S_BETA]);
    if (!std::isfinite(rn)) { kn_set_error("EMI CG broke down (non-finite residual) " + describe_scalars(sc, it)); return KNPEMI_ESOLVE; }
  }
  if (!fused && (rc = post())) return rc;
  if (c.comm_rc) { kn_set_error("EMI solve: a communication hook failed"); return KNPEMI_EHIP; }
  if (iters) *iters = it;
  if (relres) *relres = bnorm > 0 ? rn / bnorm : rn;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("krylov (emi): ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  if (amg) {   // frozen hierarchy: rebuild at the next solve once it has visibly aged
    if (G.its_ref < 0) G.its_ref = it;
    else if (it > 2 * G.its_ref + 4) { if (dist.on) G.built = false; else G.rebuild_wanted = true; }
  }
  if (rn > target) { kn_set_error("EMI CG did not converge (ksp_error_if_not_converged)"); return KNPEMI_ESOLVE; }
  return KNPEMI_OK;
}

// Jacobi-BiCGStab on the block-diagonal KNP system; x = c (csol, converted to the block order).
int kn_solve_knp(knpemi_handle* h, double rtol, double atol, int maxit, int* iters, double* relres) {
  KnDev& D = h->dev;
  const int KS = h->K - 1;
  const int n = KS * D.Ntot;
  int rc = ensure_work(h, (size_t)std::max(1, h->K - 1) * D.Ntot);
  if (rc) return rc;
  if (!D.krowptr) { kn_set_error("KNP monolithic pattern missing"); return KNPEMI_EINVAL; }
  const size_t N = h->kry_n;
  double *x = h->kry, *r = x + N, *rhat = r + N, *p = rhat + N, *v = p + N, *s = v + N, *t = s + N, *dinv = t + N;
  double *phat = h->kry + 8 * N, *shat = h->kry + 9 * N;
  Ctx c{h, n, D.krowptr, D.kcolind, D.A_knp, h->kry + 11 * N, h->kry + 11 * N + 64};
  c.lpr = spmv_lanes(h, KNPEMI_B_KNP, D.krowptr, n);
  const KnDist& dist = h->dist;
  if (dist.on) {
    c.owned = dist.d_owned_knp;
    c.which = KNPEMI_B_KNP;
    hipLaunchKernelGGL(ghost_identity_kernel, grid1(n), dim3(256), 0, h->stream, n, c.rowptr, c.colind, c.owned, D.A_knp);
    mask(c, D.b_knp);      // the ghost rows of the right-hand side are not assembled
  }
  // x0 = previous concentrations in the block order [c[0][0], c[0][1], c[1][0], ...] (first launch of the solve: with the
  // fused loop it opens the captured graph of the first chunk)
  auto pre = [&]() -> int {
    if (h->plain_knp) vec(c, V_COPY, x, nullptr, D.csol, nullptr);
    else hipLaunchKernelGGL(knp_order_kernel, grid1(n), dim3(256), 0, h->stream, D.Ntot, KS, h->n_sub, h->d_consts, x, D.csol, 1);
    return KNPEMI_OK;
  };
  // the solution back into csol / the vertex records (+ the end-of-step update when it is fused); idempotent
  auto post = [&]() -> int {
    if (h->plain_knp) vec(c, V_COPY, D.csol, nullptr, x, nullptr);
    else if (h->fuse_update) { if (int e = kn_launch_knp_writeback_update(h, x)) return e; }
    else hipLaunchKernelGGL(knp_order_kernel, grid1(n), dim3(256), 0, h->stream, D.Ntot, KS, h->n_sub, h->d_consts, x, D.csol, 0);
    return KNPEMI_OK;
  };
  const bool fused_planned = h->pc_knp == KNPEMI_PC_AMG && !dist.on && use_fused();
  if (!fused_planned && (rc = pre())) return rc;
  KnAmg& G = h->amg_knp;
  const bool amg = h->pc_knp == KNPEMI_PC_AMG;
  if (dist.on) {
    hipLaunchKernelGGL(diag_inv_kernel, grid1(n), dim3(256), 0, h->stream, n, c.rowptr, c.colind, c.vals, dinv);
    // Jacobi: the scaled SpMV A (dinv .* p) reads dinv on ghost columns, where the identity rows above left 1: take the
    // owners' 1 / a_ii instead, so that the recurrence residual stays b - A x (x is updated with the owners' dinv)
    if (!amg) if (int e = dist.halo(dist.ctx, dinv, KNPEMI_B_KNP)) c.comm_rc = c.comm_rc ? c.comm_rc : e;
  }
  if (amg && (!G.built || G.n != n)) {
    G.negative_strength = true;
    // stretched Q1 cells couple the two ends of an edge along the long direction positively: aggregate_apart keeps them
    // in different aggregates (config 2h: 9.35 / 3.95 -> 7.4 / 2.9 iterations per solve; on simplices the plain greedy
    // pass is the better one, 5.3 / 2.35 against 5.75 / 2.6 at config 2).  The DG solver handle sets the flag itself.
    // (plain_knp marks the solver handle of a DG problem, which has made its own choices: kernels_dg.hip)
    if (h->plain_knp) {}
    else if (h->NV == 8) G.positive_conflict = true;
    // prolongator smoothing along the large couplings only (KnAmg::filter_theta): the operator complexity falls from
    // 2.5-2.8 to 1.5 on simplices, the iteration counts stay (995 k tets: 6.65 / 2.8 -> 6.75 / 2.85, 3.34 -> 2.37 ms per
    // step with solves).  "Large" is by magnitude: the big positive entries of stretched Q1 cells stay in the smoothing
    // (lumping them into the diagonal cost config 2h 7.4 -> 12.6 iterations).
    if (!h->plain_knp) G.filter_theta = 0.02;
    if (getenv("KNPEMI_AMG_APART")) G.positive_conflict = atoi(getenv("KNPEMI_AMG_APART")) != 0;
    if (const char* ft = getenv("KNPEMI_AMG_FILTER")) G.filter_theta = atof(ft);
    G.want_fused = !dist.on && use_fused();
    G.want_cycle = dist.on && use_fused() && !getenv("KNPEMI_DIST_PLAIN_CYCLE");
    if ((rc = kn_amg_setup(h, G, n, D.krowptr, D.kcolind, D.A_knp, false,
                           dist.on ? dist.h_owned_knp.data() : nullptr))) return rc;
    G.its_ref = -1;
    ++G.builds;
  } else if (amg && !dist.on) {
    if (rebuild_every() > 0 && ++G.solves % rebuild_every() == 0) G.rebuild_wanted = true;
    if (G.rebuild_wanted && (rc = kn_amg_rebuild_step(h, G, n, D.krowptr, D.kcolind, D.A_knp, false))) return rc;
  }
  if (amg && (rc = kn_amg_refresh(h, G, D.A_knp))) return rc;
  double sc[S_N];
  const bool fused = amg && G.fused_ok && !dist.on && use_fused();
  int it = 0, restarts = 0;
  if (fused) {   // residual, target and 3 + 2 cycles launches per iteration (kernels_fused.hip)
    const KnFusedSys S{n, c.rowptr, c.colind, c.vals, c.sc, h->kry, N};
    if (h->knp_method == 1) {      // GMRES(30) as PETSc runs the reference's options (KNPEMI_OPT_KNP_METHOD)
      if ((rc = kn_fused_gmres(h, G, S, D.b_knp, rtol, atol, maxit, &it, &sc[S_RR], &sc[S_BB], pre, post))) return rc;
    } else
    if ((rc = kn_fused_bicgstab(h, G, S, D.b_knp, rtol, atol, maxit, &it, &sc[S_RR], &sc[S_BB], pre, post))) return rc;
  } else {
    if (fused_planned && (rc = pre())) return rc;      // (the hierarchy did not qualify for the fused loop after all)
    spmv(c, x, r, nullptr, D.b_knp, dist.on ? nullptr : dinv);   // r = b - A x
    vec(c, V_COPY, rhat, nullptr, r, nullptr);
    hipLaunchKernelGGL(bicg_init_kernel, grid1(std::max(n, (int)S_N)), dim3(256), 0, h->stream, n, p, v, c.sc);
    dots(c, 2, r, r, D.b_knp, D.b_knp, nullptr, nullptr, OP_START);
    if ((rc = read_scalars(c, sc, S_N))) return rc;
    if ((rc = check_start("KNP BiCGStab", sc))) return rc;
  }
  const double bnorm = std::sqrt(sc[S_BB]);
  const double target = std::max(atol, rtol * (bnorm > 0 ? bnorm : 1.0));
  double rn = std::sqrt(sc[S_RR]);
  const int chunk = amg ? 1 : 4;
  auto iteration = [&]() -> int {
    int rc = KNPEMI_OK;

      dots(c, 1, rhat, r, nullptr, nullptr, nullptr, nullptr, OP_BI_RHO);
      vec(c, V_BI_P, p, nullptr, r, v);
      if (amg) {
        if ((rc = kn_amg_apply(h, G, D.A_knp, dinv, p, t, phat))) return rc;   // t is free here
        spmv(c, phat, v, nullptr);
      } else spmv(c, p, v, dinv);                           // v = A M^-1 p
      dots(c, 1, rhat, v, nullptr, nullptr, nullptr, nullptr, OP_BI_ALPHA);
      vec(c, V_BI_S, s, nullptr, r, v);
      if (amg) {
        if ((rc = kn_amg_apply(h, G, D.A_knp, dinv, s, t, shat))) return rc;
        spmv(c, shat, t, nullptr);
      } else spmv(c, s, t, dinv);                           // t = A M^-1 s
      dots(c, 2, t, s, t, t, nullptr, nullptr, OP_BI_OMEGA);
      if (amg)
        hipLaunchKernelGGL(bicg_update_kernel, grid1(n), dim3(256), 0, h->stream, n, c.sc, x, r, phat, shat, s, t,
                           (const double*)nullptr);
      else
        hipLaunchKernelGGL(bicg_update_kernel, grid1(n), dim3(256), 0, h->stream, n, c.sc, x, r, p, s, s, t, dinv);
      dots(c, 1, r, r, nullptr, nullptr, nullptr, nullptr, OP_STORE3, S_RR);
    return rc;
  };
  const uint64_t gkey = graph_key(h, G, amg, chunk, n);
  // ksp_min_it (pdeSolver.py:101).  With KNPEMI_OPT_KNP_METHOD = 1 it counts GMRES iterations; where this BiCGStab loop runs
  // instead (partitioned problems, Jacobi preconditioning) an iteration applies operator and preconditioner twice
  const int min_it = std::max(0, std::min(h->knp_method == 1 ? (h->knp_min_it + 1) / 2 : h->knp_min_it, maxit));
  while (!fused && (rn > target || (it < min_it && rn != 0.0)) && it < maxit) {
    const int todo = std::min(chunk, maxit - it);
    if (todo == chunk) { if ((rc = run_chunk(h, h->graph_knp, gkey, chunk, iteration))) return rc; }
    else for (int k = 0; k < todo; ++k) if ((rc = iteration())) return rc;
    it += todo;
    if ((rc = read_scalars(c, sc, S_N))) return rc;
    rn = std::sqrt(sc[S_RR]);
    if (!std::isfinite(rn)) {
      kn_set_error("KNP BiCGStab broke down (non-finite residual) " + describe_scalars(sc, it));
      return KNPEMI_ESOLVE;
    }
    const int flag = (int)sc[S_FLAG];
    if ((flag & (F_RHO_ZERO | F_OMEGA_ZERO | F_RV_ZERO)) && rn > target) {
      // a true breakdown (rhat has become orthogonal to the recurrence, or the stabilising step vanished) with a
      // residual left: restart from the current iterate with rhat = r, as PETSc's KSPBCGS does
      if (++restarts > 3) {
        kn_set_error("KNP BiCGStab broke down repeatedly " + describe_scalars(sc, it));
        return KNPEMI_ESOLVE;
      }
      vec(c, V_COPY, rhat, nullptr, r, nullptr);
      hipLaunchKernelGGL(bicg_init_kernel, grid1(std::max(n, (int)S_N)), dim3(256), 0, h->stream, n, p, v, c.sc);
    }
  }
  if (dist.on) if (int e = dist.halo(dist.ctx, x, KNPEMI_B_KNP)) c.comm_rc = c.comm_rc ? c.comm_rc : e;
  if (c.comm_rc) { kn_set_error("KNP solve: a communication hook failed"); return KNPEMI_EHIP; }
  if (!fused && (rc = post())) return rc;
  if (iters) *iters = it;
  if (relres) *relres = bnorm > 0 ? rn / bnorm : rn;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("krylov (knp): ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  if (amg) {
    if (G.its_ref < 0) G.its_ref = it;
    else if (it > 2 * G.its_ref + 4) { if (dist.on) G.built = false; else G.rebuild_wanted = true; }
  }
  if (rn > target) { kn_set_error("KNP BiCGStab did not converge (ksp_error_if_not_converged)"); return KNPEMI_ESOLVE; }
  return KNPEMI_OK;
}

// x in the reference's block order (pdeSolver.py:117) <-> csol (ion-major over the global vertex numbering)
int kn_launch_knp_order(knpemi_handle* h, double* x, int to_blocks) {
  const KnDev& D = h->dev;
  const int n = (h->K - 1) * D.Ntot;
  if (n == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(knp_order_kernel, grid1(n), dim3(256), 0, h->stream, D.Ntot, h->K - 1, h->n_sub, h->d_consts, x, D.csol,
                     to_blocks);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("knp_order_kernel: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  return KNPEMI_OK;
}

// Initial guess of the next solve: instead of the previous solution (ksp_initial_guess_nonzero) the extrapolation of the
// last solutions -- 3 x_n - 3 x_(n-1) + x_(n-2) once three are known, 2 x_n - x_(n-1) before -- written into phi (record
// component 7) / csol in place.  Only the starting point changes; the solves still stop on the same residual criterion.
int kn_extrapolate_guess(knpemi_handle* h, int which) {
  KnDev& D = h->dev;
  const int slot = which == KNPEMI_B_EMI ? 0 : 1;
  const int n = slot == 0 ? D.Ntot : (h->K - 1) * D.Ntot;
  if (n == 0) return KNPEMI_OK;
  // order 2 (default) extrapolates through the last three solutions: 5.3 instead of 5.85 CG iterations per step while the
  // cell fires at config 2, 3.5 instead of 3.8 over the first 72 steps; KNPEMI_EXTRAPOLATE_ORDER=1: 2 x_n - x_(n-1)
  static const int order = getenv("KNPEMI_EXTRAPOLATE_ORDER") ? atoi(getenv("KNPEMI_EXTRAPOLATE_ORDER")) : 2;
  if (!h->guess_old[slot]) {
    void* p = nullptr;
    KN_HIP(hipMalloc(&p, 2 * (size_t)n * sizeof(double)));
    h->allocs.push_back(p);
    h->guess_old[slot] = static_cast<double*>(p);
    h->guess_have[slot] = slot == 0 ? -1 : 0;
  }
  // the potential a run starts with is a guess, not a solution of the system (the concentrations it starts with are the
  // initial condition, i.e. the first point of the trajectory): no history from it -- 2 x_1 - x_0 with that x_0 cost the
  // second and third solves of a run twice their iterations
  if (h->guess_have[slot] < 0) { h->guess_have[slot] = 0; return KNPEMI_OK; }
  h->gam_valid = false;
  double* cur = slot == 0 ? D.VR + 7 : D.csol;
  hipLaunchKernelGGL(extrapolate_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, n, cur, slot == 0 ? KN_REC : 1,
                     h->guess_old[slot], order >= 2 ? h->guess_old[slot] + n : (double*)nullptr, h->guess_have[slot]);
  h->guess_have[slot] = std::min(2, h->guess_have[slot] + 1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("extrapolate_kernel: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  return KNPEMI_OK;
}
