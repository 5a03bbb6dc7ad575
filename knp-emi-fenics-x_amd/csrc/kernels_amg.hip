// Smoothed-aggregation algebraic multigrid used as the preconditioner of the device Krylov solves
// (SURVEY.md section 8 f1).  The reference preconditions with hypre BoomerAMG through PETSc
// (src/knpemi/pdeSolver.py:24-35,99-110: pc_type hypre, boomeramg, one V-cycle per application); this is
// the MI355X-side counterpart: a V(1,1) cycle with damped-Jacobi smoothing whose levels are CSR SpMVs.
//
// Set-up (host, once per hierarchy): strength graph -a_ij >= theta sqrt(a_ii a_jj), greedy
// aggregation, tentative piecewise-constant prolongator smoothed once with damped Jacobi
// (P = (I - 4/(3 rho) D^-1 A) T, rho from a few power iterations on D^-1 A), Galerkin products R A P, until the
// level is small enough for an explicit dense inverse.  The idealized EMI geometries are 35:1 cables meshed with
// 10:1 elements and sub-domains coupled only through the weak membrane capacitance: the strength graph
// semi-coarsens the cross-sections and never aggregates across the membrane, which is what brings CG from
// ~1000 Jacobi iterations to ~15.
//
// The hierarchy is frozen: the matrix of the current time step is used on the finest level (smoothing,
// residuals), the coarse operators are those of the step the hierarchy was built at.  It is a preconditioner
// only, so the solve still converges to the current system; the Krylov driver rebuilds when the iteration count
// has doubled (the conductivities drift slowly with the concentrations).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <atomic>
#include <numeric>
#include <thread>

#include "knpemi_internal.h"
#include "amg_host.h"

namespace {

using namespace kn_amg_host;

// ---- device kernels -------------------------------------------------------------------------------

enum { M_AX = 0, M_PRE, M_ADD, M_JAC, M_RES };

// LPR lanes per row.
//   M_AX : y = A x
//   M_PRE: x = w dinv r (this row), y = r - A (w dinv r)      pre-smoothing from a zero guess + residual
//   M_ADD: y += A x                                             prolongation
//   M_JAC: y = x + w dinv (r - A x)                            post-smoothing (y != x)
//   M_RES: y = r - A x                                          residual (block-Jacobi smoothing applies B^-1 apart)
template <int MODE, int LPR>
__global__ __launch_bounds__(256) void amg_spmv_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ colind,
                                                       const double* __restrict__ vals, const double* x,
                                                       const double* __restrict__ r, const double* __restrict__ dinv,
                                                       double w, double* y, double* xout) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = t / LPR, l = t % LPR;
  double acc = 0.0;
  if (row < n) {
    const int a = rowptr[row], b = rowptr[row + 1];
    // four independent index -> value -> gather chains per lane and trip: a lane of these kernels otherwise has one load
    // in flight at a time, and the DG systems (20 / 56 entries per row) ran at 1.7 TB/s
    double acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int j = a + l;
    for (; j + 3 * LPR < b; j += 4 * LPR) {
      const int c0 = colind[j], c1 = colind[j + LPR], c2 = colind[j + 2 * LPR], c3 = colind[j + 3 * LPR];
      const double v0 = vals[j], v1 = vals[j + LPR], v2 = vals[j + 2 * LPR], v3 = vals[j + 3 * LPR];
      if (MODE == M_PRE) {
        acc += v0 * (dinv[c0] * r[c0]); acc1 += v1 * (dinv[c1] * r[c1]);
        acc2 += v2 * (dinv[c2] * r[c2]); acc3 += v3 * (dinv[c3] * r[c3]);
      } else {
        acc += v0 * x[c0]; acc1 += v1 * x[c1]; acc2 += v2 * x[c2]; acc3 += v3 * x[c3];
      }
    }
    for (; j < b; j += LPR) {
      const int c = colind[j];
      acc += vals[j] * (MODE == M_PRE ? dinv[c] * r[c] : x[c]);
    }
    acc = (acc + acc1) + (acc2 + acc3);
  }
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (row < n && l == 0) {
    if (MODE == M_AX) y[row] = acc;
    else if (MODE == M_PRE) { y[row] = r[row] - w * acc; xout[row] = w * dinv[row] * r[row]; }
    else if (MODE == M_ADD) y[row] += acc;
    else if (MODE == M_RES) y[row] = r[row] - acc;
    else y[row] = x[row] + w * dinv[row] * (r[row] - acc);
  }
}

// binv[c] = inverse of the BS x BS diagonal block of cell c (rows c BS .. c BS + BS - 1).  W lanes per cell (W = 4 or 8 >=
// BS): lane a reads row a of the block, the rows are exchanged through the wave, every lane eliminates the whole block and
// stores row a of the inverse (one thread per cell read 4 or 8 rows at a stride of a row: 148 us for 124 416 cells).
template <int BS>
__global__ void amg_block_inv_kernel(int nb, const int* __restrict__ rowptr, const int* __restrict__ colind,
                                     const double* __restrict__ vals, double* __restrict__ binv,
                                     const int* __restrict__ bcol, int nbmax, int cells_per_system) {
  constexpr int W = BS > 4 ? 8 : 4;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = t / W, a = t % W;
  const bool live = c < nb && a < BS;
  double mine[BS];
#pragma unroll
  for (int bb = 0; bb < BS; ++bb) mine[bb] = 0.0;
  if (live && bcol) {      // block structure known (DG systems): the diagonal block's position from the cell's block list
    const int cl = c % cells_per_system;
    int pos = 0;
    for (int b = 0; b < nbmax; ++b) if (bcol[(size_t)cl * nbmax + b] == cl) pos = b;
    const double* v = vals + rowptr[c * BS + a] + pos * BS;
#pragma unroll
    for (int bb = 0; bb < BS; ++bb) mine[bb] = v[bb];
  } else if (live)
    for (int j = rowptr[c * BS + a]; j < rowptr[c * BS + a + 1]; ++j) {
      const int b = colind[j] - c * BS;
#pragma unroll
      for (int bb = 0; bb < BS; ++bb) if (b == bb) mine[bb] = vals[j];
    }
  double m[BS * BS], inv[BS * BS];
  const int base = (threadIdx.x & 63) - a;       // first lane of this cell's group
#pragma unroll
  for (int i = 0; i < BS; ++i)
#pragma unroll
    for (int bb = 0; bb < BS; ++bb) m[i * BS + bb] = __shfl(mine[bb], base + i);
  if (c >= nb) return;
  // Gauss-Jordan without pivoting, fully unrolled (registers only): the diagonal blocks of the SIP / mass-dominated
  // systems are positive definite
#pragma unroll
  for (int i = 0; i < BS * BS; ++i) inv[i] = (i / BS == i % BS) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < BS; ++k) {
    const double d = 1.0 / m[k * BS + k];
#pragma unroll
    for (int j = 0; j < BS; ++j) { m[k * BS + j] *= d; inv[k * BS + j] *= d; }
#pragma unroll
    for (int i = 0; i < BS; ++i) {
      if (i == k) continue;
      const double f = m[i * BS + k];
#pragma unroll
      for (int j = 0; j < BS; ++j) { m[i * BS + j] -= f * m[k * BS + j]; inv[i * BS + j] -= f * inv[k * BS + j]; }
    }
  }
  if (!live) return;
#pragma unroll
  for (int i = 0; i < BS; ++i)
    if (i == a)
#pragma unroll
      for (int bb = 0; bb < BS; ++bb) binv[((size_t)c * BS + i) * BS + bb] = inv[i * BS + bb];
}

// y = (x ? x : 0) + w B^-1 v, block by block (one thread per unknown)
template <int BS>
__global__ void amg_block_apply_kernel(int n, const double* __restrict__ binv, const double* __restrict__ v, const double* x,
                                       double w, double* y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = i / BS, a = i % BS;
  double s = 0.0;
#pragma unroll
  for (int b = 0; b < BS; ++b) s += binv[((size_t)c * BS + a) * BS + b] * v[c * BS + b];
  y[i] = (x ? x[i] : 0.0) + w * s;
}

__global__ void amg_diag_inv_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ colind,
                                    const double* __restrict__ vals, double* __restrict__ dinv) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double d = 1.0;
  for (int j = rowptr[row]; j < rowptr[row + 1]; ++j)
    if (colind[j] == row) d = vals[j];
  dinv[row] = d != 0.0 ? 1.0 / d : 1.0;
}

// x = Minv r for the dense coarsest level: one wavefront per row
// (the inverse is stored as dense diagonal blocks, KnAmgLevel::dense_blk)
__global__ __launch_bounds__(256) void amg_dense_kernel(int n, const double* __restrict__ Minv, KnDenseBlocks B,
                                                        const double* __restrict__ r, double* __restrict__ x) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, l = threadIdx.x & 63;
  double acc = 0.0;
  if (row < n) {
    int b = 0;
    for (int k = 1; k < B.nb; ++k) b = row >= B.start[k] ? k : b;
    const double* m = Minv + (size_t)B.off[b] + (size_t)(row - B.start[b]) * B.size[b];
    const double* rb = r + B.start[b];
    for (int j = l; j < B.size[b]; j += 64) acc += m[j] * rb[j];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (row < n && l == 0) x[row] = acc;
}

template <int MODE>
void launch_spmv(hipStream_t st, int n, int avg_row, const int* rp, const int* ci, const double* v, const double* x,
                 const double* r, const double* dinv, double w, double* y, double* xout = nullptr) {
  if (avg_row > 128) {
    dim3 g(((size_t)n * 64 + 255) / 256);
    hipLaunchKernelGGL((amg_spmv_kernel<MODE, 64>), g, dim3(256), 0, st, n, rp, ci, v, x, r, dinv, w, y, xout);
  } else if (avg_row > 24) {
    dim3 g(((size_t)n * 16 + 255) / 256);
    hipLaunchKernelGGL((amg_spmv_kernel<MODE, 16>), g, dim3(256), 0, st, n, rp, ci, v, x, r, dinv, w, y, xout);
  } else {
    dim3 g(((size_t)n * 4 + 255) / 256);
    hipLaunchKernelGGL((amg_spmv_kernel<MODE, 4>), g, dim3(256), 0, st, n, rp, ci, v, x, r, dinv, w, y, xout);
  }
}

// y = (x ? x : 0) + omega_block B^-1 v on the finest level
void block_apply(hipStream_t st, const KnAmg& G, int n, const double* v, const double* x, double* y) {
  dim3 g((n + 255) / 256);
  if (G.block == 3) hipLaunchKernelGGL(amg_block_apply_kernel<3>, g, dim3(256), 0, st, n, G.binv, v, x, G.omega_block, y);
  else if (G.block == 4) hipLaunchKernelGGL(amg_block_apply_kernel<4>, g, dim3(256), 0, st, n, G.binv, v, x, G.omega_block, y);
  else hipLaunchKernelGGL(amg_block_apply_kernel<8>, g, dim3(256), 0, st, n, G.binv, v, x, G.omega_block, y);
}

template <class T>
int upload(KnAmg& G, const std::vector<T>& src, T** dst, hipStream_t st) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
  KN_HIP(hipMalloc(&p, bytes));
  G.allocs.push_back(p);
  // blocking copy: the sources are temporaries of the set-up (which is host-bound anyway)
  (void)st;
  if (!src.empty()) KN_HIP(hipMemcpy(p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = static_cast<T*>(p);
  return KNPEMI_OK;
}

int upload_csr(KnAmg& G, const HostCsr& A, KnAmgCsr& D, hipStream_t st) {
  int rc;
  D.n = A.n; D.m = A.m; D.nnz = (int)A.ci.size();
  if ((rc = upload(G, A.rp, &D.rp, st))) return rc;
  if ((rc = upload(G, A.ci, &D.ci, st))) return rc;
  return upload(G, A.v, &D.v, st);
}

}  // namespace

void kn_amg_free(KnAmg& G) {
  // (a background rebuild of this hierarchy, if any, keeps running: it owns its own KnAmg; kn_amg_async_join ends it)
  for (void* p : G.allocs) (void)hipFree(p);
  G.allocs.clear();
  G.zero_sc = nullptr; G.sub_fused_ok = false; G.cycle_ok = false;
  G.lev.clear();
  G.built = false;
}

// Build the hierarchy for the n x n device CSR (rowptr, colind, vals).  `singular`: the operator has the
// constant null space (EMI).
// host copy of the device CSR, ordered after everything the handle's stream holds
static int amg_fetch(knpemi_handle* h, int n, const int* d_rowptr, const int* d_colind, const double* d_vals, HostCsr& A) {
  hipStream_t st = h->stream;
  A.n = A.m = n;
  A.rp.resize(n + 1);
  KN_HIP(hipMemcpyAsync(A.rp.data(), d_rowptr, (n + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
  KN_HIP(hipStreamSynchronize(st));
  const int nnz = A.rp[n];
  A.ci.resize(nnz); A.v.resize(nnz);
  KN_HIP(hipMemcpyAsync(A.ci.data(), d_colind, nnz * sizeof(int), hipMemcpyDeviceToHost, st));
  KN_HIP(hipMemcpyAsync(A.v.data(), d_vals, nnz * sizeof(double), hipMemcpyDeviceToHost, st));
  KN_HIP(hipStreamSynchronize(st));
  return KNPEMI_OK;
}

static int amg_build(knpemi_handle* h, KnAmg& G, HostCsr&& A, int n, const int* d_rowptr, const int* d_colind, const double* d_vals,
                     bool singular, const uint8_t* h_owned, bool background);

void kn_amg_async_join(KnAmg& G);
int kn_amg_setup(knpemi_handle* h, KnAmg& G, int n, const int* d_rowptr, const int* d_colind, const double* d_vals,
                 bool singular, const uint8_t* h_owned) {
  kn_amg_async_join(G);       // a background rebuild of the hierarchy this call replaces ends first
  G.rebuild_wanted = false;
  kn_amg_free(G);
  HostCsr A;
  if (int rc = amg_fetch(h, n, d_rowptr, d_colind, d_vals, A)) return rc;
  return amg_build(h, G, std::move(A), n, d_rowptr, d_colind, d_vals, singular, h_owned, false);
}

// ---- rebuild of an aged hierarchy in the background ---------------------------------------------------------------
// The set-up is sequential host code (0.1 s at config 2, 1.4 s for the DG systems there, 14 s for the DG systems on 165 888
// hexahedra): a rebuild in the middle of a run used to stall it for that long.  The frozen hierarchy is a preconditioner --
// an old one costs iterations, not correctness -- so the rebuild runs on a host thread from a snapshot of the operator while
// the solves go on with the old hierarchy, and the next solve after it has finished swaps the new one in.
struct KnAmgAsync {
  std::thread th;
  std::atomic<int> state{0};      // 0 idle, 1 running, 2 ready, 3 failed
  KnAmg next;
  std::string error;
};

void kn_amg_async_join(KnAmg& G) {
  if (!G.async) return;
  if (G.async->th.joinable()) G.async->th.join();
  if (G.async->state.load() == 2) kn_amg_free(G.async->next);
  delete G.async;
  G.async = nullptr;
}

// Called at the head of a solve when G.rebuild_wanted: starts the background build, or swaps a finished one in.  Returns
// KNPEMI_OK and leaves G usable in every case.
int kn_amg_rebuild_step(knpemi_handle* h, KnAmg& G, int n, const int* d_rowptr, const int* d_colind, const double* d_vals,
                        bool singular) {
  if (!G.async) G.async = new KnAmgAsync();
  KnAmgAsync& a = *G.async;
  const int st = a.state.load();
  if (st == 0) {
    HostCsr A;
    if (int rc = amg_fetch(h, n, d_rowptr, d_colind, d_vals, A)) return rc;
    // the new hierarchy takes the configuration of the old one (copy), none of its levels
    a.next = KnAmg();
    a.next.negative_strength = G.negative_strength; a.next.theta = G.theta; a.next.want_fused = G.want_fused;
    a.next.first_agg = G.first_agg; a.next.first_na = G.first_na; a.next.split_first = G.split_first;
    a.next.positive_conflict = G.positive_conflict; a.next.sub_fused = G.sub_fused; a.next.want_cycle = G.want_cycle;
    a.next.filter_theta = G.filter_theta; a.next.first_tentative = G.first_tentative; a.next.split_theta = G.split_theta;
    a.next.block = G.block; a.next.its_last = G.its_last; a.next.builds = G.builds;
    a.state.store(1);
    if (a.th.joinable()) a.th.join();
    const int device = h->device;
    a.th = std::thread([h, &a, device, n, d_rowptr, d_colind, d_vals, singular](HostCsr Ah) {
      int rc = hipSetDevice(device) == hipSuccess ? KNPEMI_OK : KNPEMI_EHIP;
      if (!rc) rc = amg_build(h, a.next, std::move(Ah), n, d_rowptr, d_colind, d_vals, singular, nullptr, true);
      if (rc) { a.error = knpemi_last_error(); kn_amg_free(a.next); }
      a.state.store(rc ? 3 : 2);
    }, std::move(A));
    return KNPEMI_OK;
  }
  if (st == 1) return KNPEMI_OK;                 // still building: the old hierarchy serves
  a.th.join();
  if (st == 2) {
    const int builds = G.builds + 1, its_last = G.its_last;
    KnAmgAsync* keep = G.async;
    kn_amg_free(G);
    G = std::move(a.next);
    G.async = keep;
    G.builds = builds; G.its_last = its_last; G.its_ref = -1;
  } else if (getenv("KNPEMI_AMG_VERBOSE")) {
    fprintf(stderr, "[knpemi amg] background rebuild failed (%s): the old hierarchy stays\n", a.error.c_str());
  }
  G.rebuild_wanted = false;
  a.next = KnAmg();
  a.state.store(0);
  return KNPEMI_OK;
}

static int amg_build(knpemi_handle* h, KnAmg& G, HostCsr&& A_in, int n, const int* d_rowptr, const int* d_colind, const double* d_vals,
                     bool singular, const uint8_t* h_owned, bool background) {
  hipStream_t st = h->stream;
  HostCsr A = std::move(A_in);
  if (h_owned) {
    // partitioned problem: the hierarchy is built for this rank's diagonal block -- couplings between owned and
    // ghost unknowns are dropped, ghost rows are identity rows (block Jacobi over the ranks; the finest level of the
    // cycle works on the caller's matrix, whose ghost rows the solver has turned into identity rows too)
    for (int i = 0; i < n; ++i)
      for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
        const int c = A.ci[j];
        if (!h_owned[i]) A.v[j] = c == i ? 1.0 : 0.0;
        else if (!h_owned[c]) A.v[j] = 0.0;
      }
  }
  // The hierarchy is only as good as the values it is built from: one non-finite (or absurdly large) entry does not stop the
  // set-up -- it collapses the damping of a level (rho -> inf), changes the aggregation of every coarser level and reaches
  // the Krylov loop as a NaN several launches later.  Refuse it here, naming the entry.
  for (int i = 0; i < n; ++i)
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j)
      if (!std::isfinite(A.v[j])) {
        kn_set_error("AMG set-up: operator entry (" + std::to_string(i) + ", " + std::to_string(A.ci[j]) + ") is not finite");
        return KNPEMI_EINVAL;
      }

  const double theta = G.theta;
  const bool verbose = getenv("KNPEMI_AMG_VERBOSE") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  // fused cycle: one level fewer is worth more than a cheaper coarsest solve (every level costs two launches per cycle)
  const bool fused_loops = G.want_fused && G.block == 0 && !h_owned && G.first_na == 0;
  // (want_cycle: the merged operators for kn_fused_subcycle from level 0 on a partitioned problem, where the Krylov loop
  // itself stays the plain one)
  const bool fused = fused_loops || (G.want_cycle && G.block == 0 && G.first_na == 0);
  // (a non-singular system of several independent blocks may end on a larger dense level: dense_inverse works block by block)
  static const int nd_env = getenv("KNPEMI_AMG_NDENSE") ? atoi(getenv("KNPEMI_AMG_NDENSE")) : 0;
  // (block-smoothed hierarchies take the same limits: 5.20 -> 5.06 ms per DG step at config 2, 9.7 -> 9.0 CG iterations)
  const int n_dense = nd_env > 0 ? nd_env : (fused || G.sub_fused) ? (singular ? 1024 : 2048) : 640, max_levels = 12;
  G.fused_ok = false;
  int rc;
  G.singular = singular;
  size_t work = 0;
  HostCsr cur = std::move(A);
  for (int l = 0; l < max_levels; ++l) {
    KnAmgLevel L;
    L.n = cur.n;
    std::vector<double> d = diagonal(cur);
    for (int i = 0; i < cur.n; ++i)
      if (!std::isfinite(d[i]) || d[i] == 0.0) {
        kn_set_error("AMG set-up: level " + std::to_string(l) + " has a zero or non-finite diagonal entry in row " + std::to_string(i));
        return KNPEMI_ESOLVE;
      }
    if (l == 0) {   // finite but absurd: |a_ij| a thousand times beyond sqrt(a_ii a_jj) (amg_host.h: find_outlier)
      int oi = -1, oj = -1;
      if (find_outlier(cur, d, 1e3, &oi, &oj)) {
        char buf[256];
        double vij = 0.0;
        for (int j = cur.rp[oi]; j < cur.rp[oi + 1]; ++j) if (cur.ci[j] == oj) vij = cur.v[j];
        snprintf(buf, sizeof buf, "AMG set-up: operator entry (%d, %d) = %.6e is out of scale (a_ii = %.6e, a_jj = %.6e): "
                 "a corrupted value, refused", oi, oj, vij, d[oi], d[oj]);
        kn_set_error(buf);
        return KNPEMI_EINVAL;
      }
    }
    const double rho = estimate_rho(cur, d);
    if (!std::isfinite(rho) || !(rho > 0.0)) {
      kn_set_error("AMG set-up: spectral radius estimate of level " + std::to_string(l) + " is not a positive finite number");
      return KNPEMI_ESOLVE;
    }
    L.omega = 4.0 / (3.0 * rho);
    L.avg_row = cur.n ? (int)(cur.ci.size() / (size_t)cur.n) : 0;
    if (l == 0 && G.block > 0) {
      if (cur.n % G.block || (G.block != 3 && G.block != 4 && G.block != 8)) {
        kn_set_error("AMG set-up: bad smoother block size");
        return KNPEMI_EINVAL;
      }
      const double rb = estimate_rho_block(cur, G.block);
      if (!(rb > 0)) { kn_set_error("AMG set-up: singular diagonal block"); return KNPEMI_ESOLVE; }
      G.omega_block = 4.0 / (3.0 * rb);
      void* pb = nullptr;
      KN_HIP(hipMalloc(&pb, (size_t)cur.n * G.block * sizeof(double)));
      G.allocs.push_back(pb);
      G.binv = static_cast<double*>(pb);
    }
    if (l == 0) {   // the finest operator is the caller's CSR of the current step
      L.A.n = L.A.m = cur.n; L.A.nnz = (int)cur.ci.size();
      L.A.rp = const_cast<int*>(d_rowptr); L.A.ci = const_cast<int*>(d_colind); L.A.v = const_cast<double*>(d_vals);
    } else if ((rc = upload_csr(G, cur, L.A, st))) return rc;
    std::vector<double> dinv(cur.n);
    for (int i = 0; i < cur.n; ++i) dinv[i] = d[i] != 0.0 ? 1.0 / d[i] : 1.0;
    if ((rc = upload(G, dinv, &L.dinv, st))) return rc;
    std::vector<int> agg;
    int na = 0;
    if (l == 0 && G.first_na > 0 && (int)G.first_agg.size() == cur.n) {
      agg = G.first_agg;
      na = G.first_na;
      if (G.split_first) na = split_aggregates(cur, d, G.split_theta, h_owned, agg, na);
    } else if (cur.n > n_dense) {
      // a threshold that leaves (almost) no strong connections stalls the coarsening: relax it for this level
      double th = theta;
      for (int attempt = 0; attempt < 6; ++attempt, th = attempt == 5 ? 0.0 : 0.5 * th) {
        na = G.positive_conflict ? aggregate_apart(cur, d, th, 0.2, agg) : aggregate(cur, d, th, G.negative_strength, agg);
        if (na < cur.n * 0.7) break;
      }
    }
    // independent systems (the ion blocks of the concentration matrix) stay contiguous ranges on every level
    if (na > 0 && !h_owned) renumber_by_component(cur, agg, na);
    if (cur.n <= n_dense || na >= cur.n * 0.9 || l == max_levels - 1) {
      // coarsening stalled on a large level (e.g. a mass-dominated operator with no strong connections):
      // that level is well conditioned for Jacobi, which then stands in for the coarsest solve
      if (cur.n <= std::max(1024, n_dense)) {
        DenseInvBlocks inv;
        if (!dense_inverse_blocks(cur, singular, inv)) { kn_set_error("AMG set-up: singular coarsest operator"); return KNPEMI_ESOLVE; }
        if ((rc = upload(G, inv.v, &L.dense_inv, st))) return rc;
        L.dense_blk.nb = inv.nb;
        for (int b = 0; b < inv.nb; ++b) { L.dense_blk.start[b] = inv.start[b]; L.dense_blk.size[b] = inv.size[b]; L.dense_blk.off[b] = inv.off[b]; }
      }
      L.nc = 0;
      G.lev.push_back(L);
      work += 3 * (size_t)cur.n;
      if (verbose) fprintf(stderr, "[knpemi amg] level %d (last): n %d nnz %zu, %s (%.2f s)\n", l, cur.n, cur.ci.size(),
                           L.dense_inv ? "dense inverse" : "Jacobi", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
      break;
    }
    // given aggregates (auxiliary space): the piecewise-constant prolongator IS the embedding of that space (every broken
    // dof takes the value of its vertex); smoothing it would only widen the stencil of every coarser operator
    const bool given = l == 0 && G.first_na > 0 && (int)G.first_agg.size() == cur.n;
    HostCsr P = smoothed_prolongator(cur, d, agg, na, given && G.first_tentative ? 0.0 : L.omega, G.filter_theta);
    HostCsr R = transpose(P);
    L.nc = na;
    L.p_row = std::max(1, (int)(P.ci.size() / (size_t)P.n));
    L.r_row = std::max(1, (int)(R.ci.size() / (size_t)R.n));
    if ((rc = upload_csr(G, P, L.P, st))) return rc;
    if ((rc = upload_csr(G, R, L.R, st))) return rc;
    if (fused || (G.sub_fused && l >= 1)) {
      // V(1,1) from a zero guess is  x = w D^-1 (r + t) + Pm e_c,  t = (I - w A D^-1) r,  e_c = cycle(Rm r)  with the
      // smoother folded into the transfer operators: restriction and residual, prolongation and post-smoothing become
      // one SpMV each (kernels_fused.hip)
      HostCsr Sl = cur, Sr = cur;     // I - w D^-1 A (rows scaled), I - w A D^-1 (columns scaled)
      for_chunks(cur.n, 4 * host_threads(), [&](int, int r0, int r1) {
        for (int i = r0; i < r1; ++i)
          for (int j = cur.rp[i]; j < cur.rp[i + 1]; ++j) {
            const int c = cur.ci[j];
            Sl.v[j] = (c == i ? 1.0 : 0.0) - L.omega * cur.v[j] / d[i];
            Sr.v[j] = (c == i ? 1.0 : 0.0) - L.omega * cur.v[j] / d[c];
          }
      });
      const HostCsr Pm = spgemm(Sl, P), Rm = spgemm(R, Sr);
      if ((rc = upload_csr(G, Pm, L.Pm, st))) return rc;
      if ((rc = upload_csr(G, Rm, L.Rm, st))) return rc;
      if (l == 0 && (rc = upload(G, cur.v, &L.frozen_v, st))) return rc;
    }
    G.lev.push_back(L);
    work += 3 * (size_t)cur.n;
    if (verbose) fprintf(stderr, "[knpemi amg] level %d: n %d nnz %zu -> %d aggregates, P nnz %zu, omega %.3g (%.2f s)\n", l, cur.n,
                         cur.ci.size(), na, P.ci.size(), L.omega, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    cur = spgemm(R, spgemm(cur, P));
  }
  // per-level work vectors x, r, t (level 0 uses the caller's r and z for r and x)
  void* p = nullptr;
  KN_HIP(hipMalloc(&p, std::max<size_t>(work, 1) * sizeof(double)));
  G.allocs.push_back(p);
  double* w = static_cast<double*>(p);
  for (auto& L : G.lev) { L.x = w; L.r = w + L.n; L.t = w + 2 * (size_t)L.n; w += 3 * (size_t)L.n; }
  if (!background) KN_HIP(hipStreamSynchronize(st));      // (every upload of the build is a blocking copy)
  G.built = true;
  G.n = n;
  size_t tot = 0;
  for (auto& L : G.lev) tot += L.A.nnz;
  G.op_complexity = G.lev[0].A.nnz ? (double)tot / G.lev[0].A.nnz : 1.0;
  G.fused_ok = fused_loops && G.lev.size() >= 2 && G.lev.back().dense_inv != nullptr;
  G.cycle_ok = fused && !fused_loops && G.lev.size() >= 2 && G.lev.back().dense_inv != nullptr;
  G.sub_fused_ok = G.sub_fused && G.block > 0 && G.lev.size() >= 3 && G.lev.back().dense_inv != nullptr;
  if (G.sub_fused_ok || G.cycle_ok) {
    void* z = nullptr;
    KN_HIP(hipMalloc(&z, 32 * sizeof(double)));
    G.allocs.push_back(z);
    KN_HIP(hipMemset(z, 0, 32 * sizeof(double)));
    G.zero_sc = static_cast<double*>(z);
  }
  return KNPEMI_OK;
}

// out = V(1,1)-cycle applied to r (all of the finest size; `scratch` is a work vector, out != scratch != r).
// The finest operator/diagonal are those of the current step (`vals`, `dinv0`).  Per level: x holds the
// pre-smoothed iterate and receives the coarse correction, t the residual on the way down and the
// post-smoothed result on the way up (read by the parent's prolongation): 4 launches per level, no copies,
// fixed buffers (the sequence can be captured in a hipGraph).
int kn_amg_apply(knpemi_handle* h, KnAmg& G, const double* vals, const double* dinv0, const double* r, double* scratch,
                 double* out) {
  hipStream_t st = h->stream;
  const int nl = (int)G.lev.size();
  if (G.cycle_ok) return kn_fused_subcycle(h, G, 0, r, out);
  if (G.sub_fused_ok && G.block > 0) {
    // finest level: block-Jacobi sweeps and residuals here, everything below through the merged transfer operators
    KnAmgLevel& L = G.lev[0];
    auto residual = [&](const double* x, double* y) {
      if (h->bcols.bcol) launch_block_spmv<double>(st, h->bcols, L.n, L.A.rp, vals, x, r, y, nullptr);
      else launch_spmv<M_RES>(st, L.n, L.avg_row, L.A.rp, L.A.ci, vals, x, r, nullptr, 0.0, y);
    };
    block_apply(st, G, L.n, r, nullptr, scratch);                       // x = w B^-1 r
    residual(scratch, L.t);                                             // t = r - A x
    launch_spmv<M_AX>(st, L.nc, L.r_row, L.R.rp, L.R.ci, L.R.v, L.t, nullptr, nullptr, 0.0, G.lev[1].r);
    if (int e = kn_fused_subcycle(h, G, 1)) return e;
    launch_spmv<M_ADD>(st, L.n, L.p_row, L.P.rp, L.P.ci, L.P.v, G.lev[1].x, nullptr, nullptr, 0.0, scratch);
    residual(scratch, L.t);
    block_apply(st, G, L.n, L.t, scratch, out);                         // out = x + w B^-1 (r - A x)
    return KNPEMI_OK;
  }
  for (int l = 0; l < nl; ++l) {
    KnAmgLevel& L = G.lev[l];
    const double* rl = l == 0 ? r : L.r;
    double* xl = l == 0 ? scratch : L.x;
    const double* Av = l == 0 ? vals : L.A.v;
    const double* dinv = l == 0 ? dinv0 : L.dinv;
    if (L.nc == 0) {
      double* dst = nl == 1 ? out : L.t;
      if (L.dense_inv) {
        dim3 g(((size_t)L.n * 64 + 255) / 256);
        hipLaunchKernelGGL(amg_dense_kernel, g, dim3(256), 0, st, L.n, L.dense_inv, L.dense_blk, rl, dst);
      } else {   // Jacobi level: two damped sweeps from a zero guess
        launch_spmv<M_PRE>(st, L.n, L.avg_row, L.A.rp, L.A.ci, Av, nullptr, rl, dinv, L.omega, xl == dst ? L.x : dst, xl);
        launch_spmv<M_JAC>(st, L.n, L.avg_row, L.A.rp, L.A.ci, Av, xl, rl, dinv, L.omega, dst);
      }
      break;
    }
    if (l == 0 && G.block > 0) {   // x = w B^-1 r, t = r - A x
      block_apply(st, G, L.n, rl, nullptr, xl);
      if (h->bcols.bcol) launch_block_spmv<double>(st, h->bcols, L.n, L.A.rp, Av, xl, rl, L.t, nullptr);
      else launch_spmv<M_RES>(st, L.n, L.avg_row, L.A.rp, L.A.ci, Av, xl, rl, nullptr, 0.0, L.t);
    } else launch_spmv<M_PRE>(st, L.n, L.avg_row, L.A.rp, L.A.ci, Av, nullptr, rl, dinv, L.omega, L.t, xl);
    launch_spmv<M_AX>(st, L.nc, L.r_row, L.R.rp, L.R.ci, L.R.v, L.t, nullptr, nullptr, 0.0, G.lev[l + 1].r);
  }
  for (int l = nl - 2; l >= 0; --l) {
    KnAmgLevel& L = G.lev[l];
    const double* rl = l == 0 ? r : L.r;
    double* xl = l == 0 ? scratch : L.x;
    const double* Av = l == 0 ? vals : L.A.v;
    const double* dinv = l == 0 ? dinv0 : L.dinv;
    launch_spmv<M_ADD>(st, L.n, L.p_row, L.P.rp, L.P.ci, L.P.v, G.lev[l + 1].t, nullptr, nullptr, 0.0, xl);
    if (l == 0 && G.block > 0) {   // out = x + w B^-1 (r - A x); the level's t is free again
      if (h->bcols.bcol) launch_block_spmv<double>(st, h->bcols, L.n, L.A.rp, Av, xl, rl, L.t, nullptr);
      else launch_spmv<M_RES>(st, L.n, L.avg_row, L.A.rp, L.A.ci, Av, xl, rl, nullptr, 0.0, L.t);
      block_apply(st, G, L.n, L.t, xl, out);
    } else launch_spmv<M_JAC>(st, L.n, L.avg_row, L.A.rp, L.A.ci, Av, xl, rl, dinv, L.omega, l == 0 ? out : L.t);
  }
  return KNPEMI_OK;
}

int kn_amg_refresh(knpemi_handle* h, KnAmg& G, const double* vals) {
  if (!G.built || G.block <= 0 || G.lev.empty() || G.lev[0].nc == 0) return KNPEMI_OK;
  const KnAmgLevel& L = G.lev[0];
  const int nb = L.n / G.block;
  dim3 g(((size_t)nb * 4 + 255) / 256), g8(((size_t)nb * 8 + 255) / 256);
  const KnBlockCols& B = h->bcols;
  const int* bc = (B.bcol && B.nv == G.block) ? B.bcol : nullptr;
  const int cps = bc ? B.n / B.nv : 1;
  if (G.block == 3) hipLaunchKernelGGL(amg_block_inv_kernel<3>, g, dim3(256), 0, h->stream, nb, L.A.rp, L.A.ci, vals, G.binv, bc, B.nbmax, cps);
  else if (G.block == 4) hipLaunchKernelGGL(amg_block_inv_kernel<4>, g, dim3(256), 0, h->stream, nb, L.A.rp, L.A.ci, vals, G.binv, bc, B.nbmax, cps);
  else if (G.block == 8) hipLaunchKernelGGL(amg_block_inv_kernel<8>, g8, dim3(256), 0, h->stream, nb, L.A.rp, L.A.ci, vals, G.binv, bc, B.nbmax, cps);
  else { kn_set_error("AMG: smoother blocks of 3, 4 or 8 unknowns only"); return KNPEMI_EINVAL; }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { kn_set_error(std::string("amg_block_inv_kernel: ") + hipGetErrorString(e)); return KNPEMI_EHIP; }
  return KNPEMI_OK;
}
