// SpMV of the DG systems through their block structure: a row of cell c holds one NV-wide block per entry of
// bcol[c * nbmax ..] (the cell itself and its facet neighbours, increasing), so the column of entry idx of a row is
// bcol[idx / NV] * NV + idx % NV -- 4 bytes of index per BLOCK instead of per entry (a third of the bytes of the CSR
// kernel on broken P1).  (An fp32 copy of the values for the preconditioner's residuals was measured too: 5.58 against 5.57 ms
// per DG step at config 2 -- the kernel is not bound by its bytes; not kept.  Measured again in round 4 with the cell-wise
// kernel below and the copy refreshed once per solve: 25.4 against 25.2 ms on 166 k hexahedra, 5.74 against 5.46 ms at
// config 2, same iteration counts -- still not kept.  VT stays a template parameter.)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

struct KnBlockCols {
  const int* bcol = nullptr;   // [cells][nbmax] block columns (cell numbers), -1 beyond a cell's last block
  int nv = 0, nbmax = 0;
  int n = 0;                   // unknowns of one system (the concentration matrix is KS such systems on its diagonal)
};

// y = b ? b - A x : A x, rows of unknowns that are not owned give 0 (partitioned problems)
template <int NV, int LPR, typename VT>
__global__ __launch_bounds__(256) void block_spmv_kernel(int rows, int n, const int* __restrict__ rowptr,
                                                         const int* __restrict__ bcol, int nbmax, const VT* __restrict__ vals,
                                                         const double* __restrict__ x, const double* b, double* y,
                                                         const uint8_t* __restrict__ owned) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = t / LPR, l = t % LPR;
  const bool live = row < rows && (!owned || owned[row]);
  double acc = 0.0;
  if (live) {
    const int a = rowptr[row], e = rowptr[row + 1];
    const int sys = row / n, off = sys * n;
    const int* __restrict__ bc = bcol + (size_t)((row - off) / NV) * nbmax;
    auto col = [&](int j) { const int idx = j - a; return off + bc[idx / NV] * NV + idx % NV; };
    double acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int j = a + l;
    for (; j + 3 * LPR < e; j += 4 * LPR) {
      const int c0 = col(j), c1 = col(j + LPR), c2 = col(j + 2 * LPR), c3 = col(j + 3 * LPR);
      const double v0 = (double)vals[j], v1 = (double)vals[j + LPR], v2 = (double)vals[j + 2 * LPR], v3 = (double)vals[j + 3 * LPR];
      acc += v0 * x[c0]; acc1 += v1 * x[c1]; acc2 += v2 * x[c2]; acc3 += v3 * x[c3];
    }
    for (; j < e; j += LPR) acc += (double)vals[j] * x[col(j)];
    acc = (acc + acc1) + (acc2 + acc3);
  }
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (row < rows && l == 0) y[row] = live ? (b ? b[row] - acc : acc) : 0.0;
}

// The same product cell by cell (round 4; NV = 8, 4): the NV rows of a cell are consecutive in the value array and share their
// blocks, so NV x NV lanes take one cell -- lane (r, j) walks the blocks of row r at column j.  A load of the values then
// covers NV segments of NV doubles (hexahedra: eight 64-byte segments, the whole wave one cell), the x values of a block are
// NV consecutive doubles read once per cell instead of once per row, and the block column is the same word for all lanes of
// the cell.  The row kernel above issued ~30 loads per row of a hexahedral cell in two waves, this one 21 per cell in one.
// All loads are requested before the first product (NBMAX blocks, predicated).
template <int NV, int NBMAX, typename VT>
__global__ __launch_bounds__(256) void block_spmv_cell_kernel(int rows, int n, const int* __restrict__ rowptr,
                                                              const int* __restrict__ bcol, int nbmax, const VT* __restrict__ vals,
                                                              const double* __restrict__ x, const double* b, double* y,
                                                              const uint8_t* __restrict__ owned) {
  constexpr int LC = NV * NV;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int gcell = t / LC, lane = t % LC, r = lane / NV, j = lane % NV;
  const int row = gcell * NV + r;
  const bool in = row < rows;
  double acc = 0.0;
  if (in) {
    const int row0 = gcell * NV;
    const int a = rowptr[row0], nbk = (rowptr[row0 + 1] - a) / NV;
    const int sys = row0 / n, off = sys * n;
    const int* __restrict__ bc = bcol + (size_t)((row0 - off) / NV) * nbmax;
    int c[NBMAX];
    double v[NBMAX], xv[NBMAX];
#pragma unroll
    for (int k = 0; k < NBMAX; ++k) c[k] = k < nbk ? bc[k] : -1;
#pragma unroll
    for (int k = 0; k < NBMAX; ++k) v[k] = k < nbk ? (double)vals[a + (r * nbk + k) * NV + j] : 0.0;
#pragma unroll
    for (int k = 0; k < NBMAX; ++k) xv[k] = c[k] >= 0 ? x[off + c[k] * NV + j] : 0.0;
#pragma unroll
    for (int k = 0; k < NBMAX; ++k) acc += v[k] * xv[k];
    for (int k = NBMAX; k < nbk; ++k) acc += (double)vals[a + (r * nbk + k) * NV + j] * x[off + bc[k] * NV + j];   // (never, for conforming meshes)
  }
#pragma unroll
  for (int m = NV / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if (in && j == 0) {
    const bool live = !owned || owned[row];
    y[row] = live ? (b ? b[row] - acc : acc) : 0.0;
  }
}

inline bool block_spmv_by_cell() {
  static const bool on = getenv("KNPEMI_BLOCK_SPMV_ROWS") == nullptr;     // (the row kernel, for comparison)
  return on;
}

template <typename VT>
inline void launch_block_spmv(hipStream_t st, const KnBlockCols& B, int rows, const int* rowptr, const VT* vals, const double* x,
                              const double* b, double* y, const uint8_t* owned) {
  if (B.nv == 8 && B.nbmax <= 7 && block_spmv_by_cell()) {
    dim3 g(((size_t)rows * 8 + 255) / 256);
    hipLaunchKernelGGL((block_spmv_cell_kernel<8, 7, VT>), g, dim3(256), 0, st, rows, B.n, rowptr, B.bcol, B.nbmax, vals, x, b, y, owned);
  } else if (B.nv == 4 && B.nbmax <= 5 && block_spmv_by_cell()) {
    dim3 g(((size_t)rows * 4 + 255) / 256);
    hipLaunchKernelGGL((block_spmv_cell_kernel<4, 5, VT>), g, dim3(256), 0, st, rows, B.n, rowptr, B.bcol, B.nbmax, vals, x, b, y, owned);
  } else if (B.nv == 8) {
    dim3 g(((size_t)rows * 16 + 255) / 256);
    hipLaunchKernelGGL((block_spmv_kernel<8, 16, VT>), g, dim3(256), 0, st, rows, B.n, rowptr, B.bcol, B.nbmax, vals, x, b, y, owned);
  } else if (B.nv == 4) {
    dim3 g(((size_t)rows * 4 + 255) / 256);
    hipLaunchKernelGGL((block_spmv_kernel<4, 4, VT>), g, dim3(256), 0, st, rows, B.n, rowptr, B.bcol, B.nbmax, vals, x, b, y, owned);
  } else {
    dim3 g(((size_t)rows * 4 + 255) / 256);
    hipLaunchKernelGGL((block_spmv_kernel<3, 4, VT>), g, dim3(256), 0, st, rows, B.n, rowptr, B.bcol, B.nbmax, vals, x, b, y, owned);
  }
}
