// DG(Q1) + symmetric interior penalty on hexahedra for gfx950: the broken-space variant on
// the reference's own 3-D idealized mesh (examples/idealized_geometries/make_mesh_3D.py:100-102 builds Q1 hexahedra).
//
// As for the simplicial kernels (kernels_dg.hip) no reference file is restated: the discrete problem is the one
// oracle/knpemi_dg_oracle.py (DGOracleQ1) spells out -- the reference's volume terms cell by cell with the 2 x 2 x 2 Gauss
// rule, SIP + upwind terms with the 2 x 2 rule on interior facets (everything taken at the point), the reference's
// membrane terms (emiWeakForm.py:160-165,228-239, knpWeakForm.py:168-214) with its quadrilateral rules on tagged facets.
//
// Layout.  Local vertex j of a cell sits at the reference point whose coordinate along axis a is bit a of j, dof
// (cell c, vertex j) = 8 c + j in the 64-byte record layout of the other paths; local facet f = 2 a + b is xi_a = b, its
// vertex at facet position m (bit 0 along the lower remaining axis, bit 1 along the higher) is facet_vertex(a, b, m).
// nbr / finfo / mfid are [n_cell][6]; finfo = kind | neighbour's local facet << 2 | the neighbour's local vertex of my
// facet vertex m << (5 + 3 m) | membrane node of facet vertex m << (17 + 2 m).  A row holds one 8-wide block per cell of
// {the cell} + its facet neighbours in increasing cell order: every block of a row is one 64-byte line.
//
// One lane per row, one wave (8 cells) per workgroup, three phases:
//   1. lane (cell, facet) -- 48 of the 64 lanes -- prepares its facet IN A FRAME ALIGNED WITH THE FACET: both adjacent
//      cells are trilinear over (s, t) on the facet and a depth coordinate w, x = sum_m N_m(s, t) (x_m + w d_m) with d_m the
//      edge from facet vertex m to the vertex behind it, so at a facet point the Jacobian of either cell is
//      [t_s t_t d(s, t)] with the SHARED tangents t_s, t_t and the cell's own depth vector d.  The normal derivative of
//      every basis function of either cell then needs three numbers per cell and point:
//        grad phi_(m, w=0) . n = d_s N_m a_s + d_t N_m a_t - N_m c,   grad phi_(m, w=1) . n = N_m c,
//        c = 1 / (d . n),  a_s = ((t_t x d) . n) / det,  a_t = ((d x t_s) . n) / det,  det = d . (t_s x t_t)
//      -- no 3 x 3 inverse of the neighbour, no reference coordinates of the neighbour.  The lane leaves per point the
//      surface weight, 1 / h_F = (|c_T| + |c_N|) / 2, (a_s, a_t, c) of both cells and the traces the rows need;
//   1b. lane (cell, i) inverts the Jacobian at volume Gauss point i of its cell and leaves the physical gradients of the
//      cell's 8 basis functions at that point in LDS;
//   2. every lane builds its row: volume terms from the eight shared gradient tables, facet terms from the six summaries.
//      Loops over facets, points and columns are unrolled, so every index into a register array is a compile-time
//      constant; what depends on the lane's own vertex i are a few scalars per point (its trace and normal derivative).
//      Every block of a row is one 64-byte line: the lanes put their blocks (a neighbour block in the neighbour's local
//      column order) into LDS scratch lines and four lanes store one line together, 16 whole lines per instruction.
// Box meshes (all cells orthogonal parallelepipeds) take the kernels at the end of this file.
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "dg_internal.h"

namespace {

using namespace kn_dg;

constexpr int HX_NV = 8, HX_NFC = 6;
constexpr int HX_CELLS = DG_BLOCK / HX_NV;          // cells per workgroup
constexpr int HX_TASKS = HX_CELLS * HX_NFC;         // (cell, facet) pairs per workgroup
constexpr int HX_FS_EMI = 49;                       // doubles per facet summary (12 per point, odd pitch)
constexpr int HX_VG = 25;                           // per volume point: the 8 physical gradients (24) + weight * |det|
constexpr int HX_SCR = 10;                          // scratch pitch of the block transposition (even: 16-byte reads)
constexpr double HX_G0 = 0.21132486540518711775, HX_G1 = 0.78867513459481288225;   // 2-point Gauss on [0, 1]

__host__ __device__ constexpr int hx_ax1(int a) { return a == 0 ? 1 : 0; }
__host__ __device__ constexpr int hx_ax2(int a) { return a == 2 ? 1 : 2; }
__host__ __device__ constexpr int hx_facet_vertex(int a, int b, int m) {
  return (b << a) | ((m & 1) << hx_ax1(a)) | ((m >> 1) << hx_ax2(a));
}
// value at a Gauss point of the 1-D shape function of a vertex: the far one (G0) when the bits differ
__host__ __device__ constexpr double hx_w(int bit) { return bit ? HX_G0 : HX_G1; }

struct Vec3 {
  double x, y, z;
};
__device__ __forceinline__ Vec3 cross(const Vec3& a, const Vec3& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double dot3(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// position of the row's blocks: the cell itself and its neighbours in increasing cell order
__device__ __forceinline__ void hx_slots(int T, const int (&nb)[HX_NFC], int& slot_self, int (&slot)[HX_NFC]) {
  slot_self = 0;
#pragma unroll
  for (int f = 0; f < HX_NFC; ++f) {
    slot_self += nb[f] >= 0 && nb[f] < T;
    int s = T < nb[f];
#pragma unroll
    for (int g = 0; g < HX_NFC; ++g) s += nb[g] >= 0 && nb[g] < nb[f];
    slot[f] = s;
  }
}

// tangents, surface element, outward unit normal of T and the depth-frame numbers of T and N at facet point p
struct FacetFrame {
  double A, aTs, aTt, cT, aNs, aNt, cN, N[4], ds[4], dt[4];
};
__device__ __forceinline__ void hx_shapes(int p, double (&N)[4], double (&ds)[4], double (&dt)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const double ws = ((m ^ p) & 1) ? HX_G0 : HX_G1, wt = (((m ^ p) >> 1) & 1) ? HX_G0 : HX_G1;
    N[m] = ws * wt;
    ds[m] = (m & 1) ? wt : -wt;
    dt[m] = (m >> 1) ? ws : -ws;
  }
}
__device__ __forceinline__ void hx_frame(int p, const double (&xm)[4][3], const double (&dT)[4][3], const double (&dN)[4][3],
                                         bool have_n, FacetFrame& F) {
  hx_shapes(p, F.N, F.ds, F.dt);
  Vec3 ts{0, 0, 0}, tt{0, 0, 0}, eT{0, 0, 0}, eN{0, 0, 0};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    ts.x += F.ds[m] * xm[m][0]; ts.y += F.ds[m] * xm[m][1]; ts.z += F.ds[m] * xm[m][2];
    tt.x += F.dt[m] * xm[m][0]; tt.y += F.dt[m] * xm[m][1]; tt.z += F.dt[m] * xm[m][2];
    eT.x += F.N[m] * dT[m][0]; eT.y += F.N[m] * dT[m][1]; eT.z += F.N[m] * dT[m][2];
    eN.x += F.N[m] * dN[m][0]; eN.y += F.N[m] * dN[m][1]; eN.z += F.N[m] * dN[m][2];
  }
  const Vec3 cr = cross(ts, tt);
  const double a2 = dot3(cr, cr), ra = fast_rsqrt(a2);
  F.A = a2 * ra;
  const double detT = dot3(cr, eT);
  const double nu = detT > 0.0 ? -ra : ra;            // n = nu cr points away from T
  const Vec3 n{nu * cr.x, nu * cr.y, nu * cr.z};
  const double riT = fast_rcp(detT);
  F.cT = dot3(cr, n) * riT;
  F.aTs = dot3(cross(tt, eT), n) * riT;
  F.aTt = dot3(cross(eT, ts), n) * riT;
  F.cN = F.aNs = F.aNt = 0.0;
  if (have_n) {
    const double riN = fast_rcp(dot3(cr, eN));
    F.cN = dot3(cr, n) * riN;
    F.aNs = dot3(cross(tt, eN), n) * riN;
    F.aNt = dot3(cross(eN, ts), n) * riN;
  }
}

// the physical gradients of the 8 basis functions and weight * |det J| of the cell at volume Gauss point p (runtime):
// out[3 j + d] = d phi_j / d x_d, out[24] = |det| / 8.  Computed once per (cell, point) by the lane whose vertex has the
// point's number and read by the 8 rows of the cell.
__device__ __forceinline__ void hx_volume_point(const double (&X)[8][3], int p, double* out) {
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};   // J[d][t] = d x_d / d xi_t
  const double w0[2] = {(p & 1) ? HX_G0 : HX_G1, (p & 1) ? HX_G1 : HX_G0};          // shape of bit 0 = 0 / 1 along axis 0
  const double w1[2] = {(p & 2) ? HX_G0 : HX_G1, (p & 2) ? HX_G1 : HX_G0};
  const double w2[2] = {(p & 4) ? HX_G0 : HX_G1, (p & 4) ? HX_G1 : HX_G0};
  double dr[8][3];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double a = w0[j & 1], b = w1[(j >> 1) & 1], c = w2[(j >> 2) & 1];
    dr[j][0] = ((j & 1) ? 1.0 : -1.0) * b * c; dr[j][1] = ((j & 2) ? 1.0 : -1.0) * a * c; dr[j][2] = ((j & 4) ? 1.0 : -1.0) * a * b;
#pragma unroll
    for (int d = 0; d < 3; ++d) { J[d][0] += X[j][d] * dr[j][0]; J[d][1] += X[j][d] * dr[j][1]; J[d][2] += X[j][d] * dr[j][2]; }
  }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
               c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, ri = fast_rcp(det);
  // inverse: Ji[t][d] = cofactor(J)[d][t] / det  (row t = grad xi_t)
  double Ji[9];
  Ji[0] = c00 * ri; Ji[1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * ri; Ji[2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * ri;
  Ji[3] = c01 * ri; Ji[4] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * ri; Ji[5] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * ri;
  Ji[6] = c02 * ri; Ji[7] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * ri; Ji[8] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * ri;
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int d = 0; d < 3; ++d) out[3 * j + d] = dr[j][0] * Ji[d] + dr[j][1] * Ji[3 + d] + dr[j][2] * Ji[6 + d];
  out[24] = 0.125 * fabs(det);
}

// values of the 8 basis functions at volume point p (runtime)
__device__ __forceinline__ void hx_values(int p, double (&ph)[8]) {
  const double w0[2] = {(p & 1) ? HX_G0 : HX_G1, (p & 1) ? HX_G1 : HX_G0};
  const double w1[2] = {(p & 2) ? HX_G0 : HX_G1, (p & 2) ? HX_G1 : HX_G0};
  const double w2[2] = {(p & 4) ? HX_G0 : HX_G1, (p & 4) ? HX_G1 : HX_G0};
#pragma unroll
  for (int j = 0; j < 8; ++j) ph[j] = w0[j & 1] * w1[(j >> 1) & 1] * w2[(j >> 2) & 1];
}

// A block of a row (8 values = one 64-byte line of the CSR value array) leaves through LDS: every lane puts its block
// into its scratch line -- a neighbour block in the NEIGHBOUR's local column order (values arrive in facet order: on-facet
// vertices m, then the vertices behind them) -- and the offset of the line into dsts; then four lanes store one line
// together (16 bytes each, 16 whole lines per store instruction) instead of every lane storing quarter lines of its own
// (64 partial lines per instruction: the write path then runs at a quarter of its request rate).  Called by ALL lanes of
// the workgroup: `have` = this lane's row has the block, at element offset `off` of `base`.
__device__ __forceinline__ void hx_put_nbr(double* scr, unsigned fi, const double (&B0)[4], const double (&B1)[4]) {
  const int ao = ((fi >> 2) & 7) >> 1;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int jn = (fi >> (5 + 3 * m)) & 7;
    scr[jn] = B0[m];
    scr[jn ^ (1 << ao)] = B1[m];
  }
}
__device__ __forceinline__ void hx_put_own(double* scr, const double (&A)[8]) {
  double2* q = reinterpret_cast<double2*>(scr);
#pragma unroll
  for (int e = 0; e < 4; ++e) q[e] = make_double2(A[2 * e], A[2 * e + 1]);
}
// The workgroup is ONE wavefront: its LDS operations execute in program order, so the exchange between the lanes needs
// no barrier -- only the compiler must keep the order (and the counter wait makes the written data visible to the
// reads that follow).  __syncthreads() would also wait for every global store still in flight (its fence covers global
// memory): seven to fourteen times per row, each time the full latency of the write path.
static_assert(DG_BLOCK == 64, "the block stores of the hexahedral kernels assume single-wave workgroups");
__device__ __forceinline__ void hx_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void hx_flush(const double* scrs, int* dsts, int tid, bool have, int off, double* base) {
  dsts[tid] = have ? off : -1;
  hx_lds_sync();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = 16 * k + (tid >> 2), part = tid & 3;
    const int d = dsts[r];
    if (d >= 0) {
      // written once, read by a later kernel: a streaming store, so that the value arrays (600 MB per system at 166 k cells)
      // do not push the records the neighbours are about to read out of the L2
      typedef double hx_d2 __attribute__((ext_vector_type(2)));
      const hx_d2 v = *reinterpret_cast<const hx_d2*>(scrs + r * HX_SCR + 2 * part);
      __builtin_nontemporal_store(v, reinterpret_cast<hx_d2*>(base + d + 2 * part));
    }
  }
  hx_lds_sync();
}

// ---------------------------------------------------------------------------------------------------------------------
// potential system
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DG_BLOCK) void dg_emi_hex_kernel(DgDev D, const DgConsts* __restrict__ Cp, int chunk, int splitting) {
  __shared__ double lrec[DG_BLOCK * DG_RPITCH];
  __shared__ double fsum[HX_TASKS * HX_FS_EMI];
  __shared__ double vgeo[DG_BLOCK * HX_VG];       // later the scratch of the block transposition (HX_SCR <= HX_VG)
  __shared__ int dsts[DG_BLOCK];
  const DgConsts& C = *Cp;
  const int cell0 = dg_block_index(blockIdx.x, chunk) * HX_CELLS;
  if (cell0 >= D.n_cell) return;
  const int ncell = min(HX_CELLS, D.n_cell - cell0);
  const int tid = threadIdx.x;
  const int lc = tid >> 3, i = tid & 7, T = cell0 + lc, row = T * HX_NV + i;
  const bool valid = lc < ncell;
  // phase 0: records of the workgroup's cells; facet tables of this lane's (cell, facet) task
  const int tc = tid / HX_NFC, tf = tid - tc * HX_NFC;
  const bool task = tid < HX_TASKS && tc < ncell;
  int tnb = -1, tmf = -1;
  unsigned tfi = 0;
  if (valid) {
    const double2* p = reinterpret_cast<const double2*>(D.rec + (size_t)row * KN_REC);
    double2* q = reinterpret_cast<double2*>(lrec + tid * DG_RPITCH);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3];
    q[0] = a; q[1] = b; q[2] = c; q[3] = d;
  }
  if (task) {
    tnb = D.nbr[(cell0 + tc) * HX_NFC + tf];
    tfi = D.finfo[(cell0 + tc) * HX_NFC + tf];
    tmf = D.mfid[(cell0 + tc) * HX_NFC + tf];
  }
  int nb[HX_NFC] = {-1, -1, -1, -1, -1, -1};
  unsigned fi[HX_NFC] = {0, 0, 0, 0, 0, 0};
  int s = 0, rp = 0;
  if (valid) {
    s = D.cell_sub[T];
    rp = D.rowptr[row];
#pragma unroll
    for (int f = 0; f < HX_NFC; ++f) { nb[f] = D.nbr[T * HX_NFC + f]; fi[f] = D.finfo[T * HX_NFC + f]; }
  }
  __syncthreads();
  // phase 1: facet summaries
  if (task && tnb >= 0) {
    const int a = tf >> 1, b = tf & 1, kind = tfi & 3, ts_ = D.cell_sub[cell0 + tc];
    double xm[4][3], dT[4][3], dN[4][3], kT[4], kN[4], sT0[4], sT1[4], sN0[4], sN1[4];
    const int ao = ((tfi >> 2) & 7) >> 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int jv = hx_facet_vertex(a, b, m), jo = jv ^ (1 << a);
      const double* r0 = lrec + (tc * HX_NV + jv) * DG_RPITCH;
      const double* r1 = lrec + (tc * HX_NV + jo) * DG_RPITCH;
      double k0 = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
      for (int q = 0; q < KN_MAXK; ++q) {
        k0 += C.kap[ts_][q] * r0[KN_CSLOT(q)]; g0 += C.sig[ts_][q] * r0[KN_CSLOT(q)]; g1 += C.sig[ts_][q] * r1[KN_CSLOT(q)];
      }
      kT[m] = k0; sT0[m] = g0; sT1[m] = g1;
#pragma unroll
      for (int d = 0; d < 3; ++d) { xm[m][d] = r0[d]; dT[m][d] = r1[d] - r0[d]; dN[m][d] = 0.0; }
      kN[m] = sN0[m] = sN1[m] = 0.0;
    }
    double* my = fsum + tid * HX_FS_EMI;
    if (kind == 1) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int jn = (tfi >> (5 + 3 * m)) & 7, jt = jn ^ (1 << ao);
        const double* r0 = D.rec + ((size_t)tnb * HX_NV + jn) * KN_REC;
        const double* r1 = D.rec + ((size_t)tnb * HX_NV + jt) * KN_REC;
        double k0 = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
        for (int q = 0; q < KN_MAXK; ++q) {
          k0 += C.kap[ts_][q] * r0[KN_CSLOT(q)]; g0 += C.sig[ts_][q] * r0[KN_CSLOT(q)]; g1 += C.sig[ts_][q] * r1[KN_CSLOT(q)];
        }
        kN[m] = k0; sN0[m] = g0; sN1[m] = g1;
#pragma unroll
        for (int d = 0; d < 3; ++d) dN[m][d] = r1[d] - xm[m][d];
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        FacetFrame F;
        hx_frame(p, xm, dT, dN, true, F);
        double kTq = 0.0, kNq = 0.0, JT = 0.0, JN = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          kTq += F.N[m] * kT[m]; kNq += F.N[m] * kN[m];
          JT += sT0[m] * (F.ds[m] * F.aTs + F.dt[m] * F.aTt - F.N[m] * F.cT) + sT1[m] * F.N[m] * F.cT;
          JN += sN0[m] * (F.ds[m] * F.aNs + F.dt[m] * F.aNt - F.N[m] * F.cN) + sN1[m] * F.N[m] * F.cN;
        }
        double* o = my + 12 * p;
        o[0] = 0.25 * F.A; o[1] = 0.5 * (fabs(F.cT) + fabs(F.cN));
        o[2] = F.aTs; o[3] = F.aTt; o[4] = F.cT; o[5] = kTq; o[6] = JT;
        o[7] = F.aNs; o[8] = F.aNt; o[9] = F.cN; o[10] = kNq; o[11] = JN;
      }
    } else {
      // membrane facet: C_phi [u][v] and C_phi g [v] with the 2 x 2 rule; g = phi_M (- I_ch / C_phi without the
      // splitting scheme) at the facet's own nodes (emiWeakForm.py:160-165, 228-239)
      double g[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int q = tmf * 4 + ((tfi >> (17 + 2 * m)) & 3);
        double v = D.phiM[q];
        if (!splitting) {
          double it = 0.0;
          for (int k = 0; k < C.K; ++k) it += D.Ich[(size_t)k * D.nq + q];
          v -= it / C.C_phi;
        }
        g[m] = v;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        FacetFrame F;
        hx_frame(p, xm, dT, dN, false, F);
        double gq = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) gq += F.N[m] * g[m];
        my[12 * p] = 0.25 * F.A * C.C_phi;
        my[12 * p + 1] = gq;
      }
    }
  }
  // phase 1b: the cell's Jacobian at volume point i
  double X[8][3], kap[8], sg[8];
  if (valid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double* r = lrec + (lc * HX_NV + j) * DG_RPITCH;
      double k0 = 0.0, g0 = 0.0;
#pragma unroll
      for (int q = 0; q < KN_MAXK; ++q) { k0 += C.kap[s][q] * r[KN_CSLOT(q)]; g0 += C.sig[s][q] * r[KN_CSLOT(q)]; }
      kap[j] = k0; sg[j] = g0;
#pragma unroll
      for (int d = 0; d < 3; ++d) X[j][d] = r[d];
    }
    hx_volume_point(X, i, vgeo + tid * HX_VG);
  }
  __syncthreads();
  // phase 2: the row (every lane stays for the block stores below)
  double A[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double rhs = 0.0;
  if (valid) {
    const double* vg = vgeo + lc * HX_NV * HX_VG;
    const int x = i;
#pragma unroll 1
    for (int P = 0; P < 8; ++P) {
      const double* gt = vg + P * HX_VG;
      const double wd = gt[24];
      const double gi[3] = {gt[3 * x], gt[3 * x + 1], gt[3 * x + 2]};   // the row's own gradient: runtime vertex
      double g[8][3], ph[8];
      hx_values(P, ph);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int d = 0; d < 3; ++d) g[j][d] = gt[3 * j + d];
      double kq = 0.0, js[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        kq += kap[j] * ph[j];
#pragma unroll
        for (int d = 0; d < 3; ++d) js[d] += sg[j] * g[j][d];
      }
      const double wk = wd * kq;
#pragma unroll
      for (int j = 0; j < 8; ++j) A[j] += wk * (gi[0] * g[j][0] + gi[1] * g[j][1] + gi[2] * g[j][2]);
      rhs -= wd * (gi[0] * js[0] + gi[1] * js[1] + gi[2] * js[2]);
    }
  }
  int slot_self, slot[HX_NFC];
  hx_slots(T, nb, slot_self, slot);
  __syncthreads();   // every lane of the wave is done with the shared Jacobians: their space becomes the scratch lines
  double* scr = vgeo + tid * HX_SCR;
  auto facet = [&](auto F_) {
    constexpr int f = decltype(F_)::value, a = f >> 1, b = f & 1;
    const bool have = valid && nb[f] >= 0;
    if (have) {
    const int kind = fi[f] & 3;
    const bool on = ((i >> a) & 1) == b;
    const int mi = ((i >> hx_ax1(a)) & 1) | (((i >> hx_ax2(a)) & 1) << 1);
    const double* fd = fsum + (lc * HX_NFC + f) * HX_FS_EMI;
    double B0[4] = {0, 0, 0, 0}, B1[4] = {0, 0, 0, 0};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      double N[4], ds[4], dt[4];
      hx_shapes(p, N, ds, dt);
      const int xm_ = mi ^ p;
      const double wa = (xm_ & 1) ? HX_G0 : HX_G1, wb = (xm_ & 2) ? HX_G0 : HX_G1;
      const double Ni = wa * wb, dsi = (mi & 1) ? wb : -wb, dti = (mi & 2) ? wa : -wa;
      const double* o = fd + 12 * p;
      if (kind == 1) {
        const double W = o[0], ih = o[1], aTs = o[2], aTt = o[3], cT = o[4], kT = o[5], JT = o[6];
        const double aNs = o[7], aNt = o[8], cN = o[9], kN = o[10], JN = o[11];
        const double phi_i = on ? Ni : 0.0;
        const double gni = on ? dsi * aTs + dti * aTt - Ni * cT : Ni * cT;
        const double pen = C.gamma * ih * (0.5 * (kT + kN)) * W * phi_i;
        const double c1 = -0.5 * W * kT * phi_i, c2 = -0.5 * W * kT * gni, c1n = -0.5 * W * kN * phi_i;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const double gT = ds[m] * aTs + dt[m] * aTt - N[m] * cT;
          const double gN = ds[m] * aNs + dt[m] * aNt - N[m] * cN;
          A[hx_facet_vertex(a, b, m)] += c1 * gT + (c2 + pen) * N[m];
          A[hx_facet_vertex(a, b, m) ^ (1 << a)] += c1 * (N[m] * cT);
          B0[m] += c1n * gN - (c2 + pen) * N[m];
          B1[m] += c1n * (N[m] * cN);
        }
        rhs += W * 0.5 * (JT + JN) * phi_i;
      } else if (on) {
        const double W = o[0] * Ni;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          A[hx_facet_vertex(a, b, m)] += W * N[m];
          B0[m] -= W * N[m];
        }
        rhs += (kind == 3 ? W : -W) * o[1];
      }
    }
    hx_put_nbr(scr, fi[f], B0, B1);
    }
    hx_flush(vgeo, dsts, tid, have, rp + slot[f] * HX_NV, D.A_emi);
  };
  facet(std::integral_constant<int, 0>{}); facet(std::integral_constant<int, 1>{});
  facet(std::integral_constant<int, 2>{}); facet(std::integral_constant<int, 3>{});
  facet(std::integral_constant<int, 4>{}); facet(std::integral_constant<int, 5>{});
  if (valid) hx_put_own(scr, A);
  hx_flush(vgeo, dsts, tid, valid, rp + slot_self * HX_NV, D.A_emi);
  if (valid) D.b_emi[row] = rhs;
}

// ---------------------------------------------------------------------------------------------------------------------
// concentration systems: A_knp[k], b_knp[k], k < K - 1.  The unit-diffusivity SIP entries and the mass matrix are the
// same for every ion (D_k is constant inside a sub-domain); the upwinded drift differs by the sign of z_k only, so the
// row keeps the positive and the negative part of the facet's drift term apart and every ion picks its side.
// ---------------------------------------------------------------------------------------------------------------------
template <int KS> constexpr int hx_fs_knp() { return 41; }   // 10 per point | membrane: 4 x KS integrals

template <int KS>
__global__ __launch_bounds__(DG_BLOCK) void dg_knp_hex_kernel(DgDev D, const DgConsts* __restrict__ Cp, int chunk, int splitting) {
  constexpr int FS = hx_fs_knp<KS>();
  __shared__ double lrec[DG_BLOCK * DG_RPITCH];
  __shared__ double fsum[HX_TASKS * FS];
  __shared__ double vgeo[DG_BLOCK * HX_VG];
  __shared__ int dsts[DG_BLOCK];
  const DgConsts& C = *Cp;
  const int cell0 = dg_block_index(blockIdx.x, chunk) * HX_CELLS;
  if (cell0 >= D.n_cell) return;
  const int ncell = min(HX_CELLS, D.n_cell - cell0);
  const int tid = threadIdx.x;
  const int lc = tid >> 3, i = tid & 7, T = cell0 + lc, row = T * HX_NV + i;
  const bool valid = lc < ncell;
  const int tc = tid / HX_NFC, tf = tid - tc * HX_NFC;
  const bool task = tid < HX_TASKS && tc < ncell;
  int tnb = -1, tmf = -1;
  unsigned tfi = 0;
  if (valid) {
    const double2* p = reinterpret_cast<const double2*>(D.rec + (size_t)row * KN_REC);
    double2* q = reinterpret_cast<double2*>(lrec + tid * DG_RPITCH);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3];
    q[0] = a; q[1] = b; q[2] = c; q[3] = d;
  }
  if (task) {
    tnb = D.nbr[(cell0 + tc) * HX_NFC + tf];
    tfi = D.finfo[(cell0 + tc) * HX_NFC + tf];
    tmf = D.mfid[(cell0 + tc) * HX_NFC + tf];
  }
  int nb[HX_NFC] = {-1, -1, -1, -1, -1, -1};
  unsigned fi[HX_NFC] = {0, 0, 0, 0, 0, 0};
  int s = 0, rp = 0;
  if (valid) {
    s = D.cell_sub[T];
    rp = D.rowptr[row];
#pragma unroll
    for (int f = 0; f < HX_NFC; ++f) { nb[f] = D.nbr[T * HX_NFC + f]; fi[f] = D.finfo[T * HX_NFC + f]; }
  }
  __syncthreads();
  if (task && tnb >= 0) {
    const int a = tf >> 1, b = tf & 1, kind = tfi & 3, ts_ = D.cell_sub[cell0 + tc];
    const int ao = ((tfi >> 2) & 7) >> 1;
    double xm[4][3], dT[4][3], dN[4][3], pT0[4], pT1[4], pN0[4], pN1[4];
    const double* rT[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int jv = hx_facet_vertex(a, b, m), jo = jv ^ (1 << a);
      const double* r0 = lrec + (tc * HX_NV + jv) * DG_RPITCH;
      const double* r1 = lrec + (tc * HX_NV + jo) * DG_RPITCH;
      rT[m] = r0;
      pT0[m] = r0[7]; pT1[m] = r1[7];
#pragma unroll
      for (int d = 0; d < 3; ++d) { xm[m][d] = r0[d]; dT[m][d] = r1[d] - r0[d]; dN[m][d] = 0.0; }
      const int jn = (tfi >> (5 + 3 * m)) & 7, jt = jn ^ (1 << ao);
      const double* n0 = D.rec + ((size_t)tnb * HX_NV + jn) * KN_REC;
      pN0[m] = n0[7];
      pN1[m] = 0.0;
      if (kind == 1) {
        const double* n1 = D.rec + ((size_t)tnb * HX_NV + jt) * KN_REC;
        pN1[m] = n1[7];
#pragma unroll
        for (int d = 0; d < 3; ++d) dN[m][d] = n1[d] - xm[m][d];
      }
    }
    double* my = fsum + tid * FS;
    if (kind == 1) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        FacetFrame F;
        hx_frame(p, xm, dT, dN, true, F);
        double gT = 0.0, gN = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          gT += pT0[m] * (F.ds[m] * F.aTs + F.dt[m] * F.aTt - F.N[m] * F.cT) + pT1[m] * F.N[m] * F.cT;
          gN += pN0[m] * (F.ds[m] * F.aNs + F.dt[m] * F.aNt - F.N[m] * F.cN) + pN1[m] * F.N[m] * F.cN;
        }
        double* o = my + 10 * p;
        o[0] = 0.25 * F.A; o[1] = 0.5 * (fabs(F.cT) + fabs(F.cN));
        o[2] = F.aTs; o[3] = F.aTt; o[4] = F.cT; o[5] = F.aNs; o[6] = F.aNt; o[7] = F.cN; o[8] = gT; o[9] = gN;
      }
    } else {
      // membrane facet (knpWeakForm.py:168-214), 4 x 4 rule, once per (cell, facet).  With alpha_k = D_k z_k^2 c_k /
      // sum_j D_j z_j^2 c_j of this side the integrand is
      //   sgn [ alpha_k C_M / (F z_k dt) ([phi] - phi_M - (dt / C_M) I_ch) + I_ch_k / (F z_k) ],  sgn = +1 on the ECS side
      double jm[4], pm[4], It[4], Ik[4][KN_MAXK], Gk[4][KS];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        jm[m] = kind == 2 ? pN0[m] - pT0[m] : pT0[m] - pN0[m];
        const int q = tmf * 4 + ((tfi >> (17 + 2 * m)) & 3);
        pm[m] = D.phiM[q];
        double it = 0.0;
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) {
          Ik[m][k] = k < C.K ? D.Ich[(size_t)k * D.nq + q] : 0.0;
          it += Ik[m][k];
        }
        It[m] = it;
#pragma unroll
        for (int k = 0; k < KS; ++k) Gk[m][k] = 0.0;
      }
      const double sgn = kind == 2 ? 1.0 : -1.0;
      const double* qw = D.qtab;
      const double* qN = D.qtab + D.nquad;
      const double* qd = D.qtab + D.nquad * 5;
      for (int q = 0; q < D.nquad; ++q) {
        double cq[KN_MAXK], iq[KN_MAXK], jq = 0.0, pq = 0.0, itq = 0.0, Nq[4];
        Vec3 ts{0, 0, 0}, tt{0, 0, 0};
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) { cq[k] = 0.0; iq[k] = 0.0; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          Nq[m] = qN[q * 4 + m];
          const double d0 = qd[(q * 4 + m) * 2], d1 = qd[(q * 4 + m) * 2 + 1];
          ts.x += d0 * xm[m][0]; ts.y += d0 * xm[m][1]; ts.z += d0 * xm[m][2];
          tt.x += d1 * xm[m][0]; tt.y += d1 * xm[m][1]; tt.z += d1 * xm[m][2];
          // ion k of the record sits in slot KN_CSLOT(k) = 4 + k (k < 3), 3 (k = 3)
#pragma unroll
          for (int k = 0; k < KN_MAXK; ++k) { cq[k] += Nq[m] * rT[m][KN_CSLOT(k)]; iq[k] += Nq[m] * Ik[m][k]; }
          jq += Nq[m] * jm[m]; pq += Nq[m] * pm[m]; itq += Nq[m] * It[m];
        }
        const Vec3 cr = cross(ts, tt);
        const double a2 = dot3(cr, cr), area = a2 * fast_rsqrt(a2);
        double asum = 0.0;
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) asum += C.az2D[ts_][k] * cq[k];
        const double w = sgn * qw[q] * area;
        double drive = jq - pq;
        if (splitting) drive -= (C.dt / C.C_M) * itq;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const double al = C.az2D[ts_][k] * cq[k] / asum;
          const double fz = 1.0 / (C.F * C.z[k]);
          const double fq = w * (al * C.C_M * fz * C.inv_dt * drive + iq[k] * fz);
#pragma unroll
          for (int m = 0; m < 4; ++m) Gk[m][k] += Nq[m] * fq;
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < KS; ++k) my[m * KS + k] = Gk[m][k];
    }
  }
  double X[8][3], ph[8];
  if (valid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double* r = lrec + (lc * HX_NV + j) * DG_RPITCH;
      ph[j] = r[7];
#pragma unroll
      for (int d = 0; d < 3; ++d) X[j][d] = r[d];
    }
    hx_volume_point(X, i, vgeo + tid * HX_VG);
  }
  __syncthreads();
  // volume: mass, unit stiffness, unit drift (grad phi_i . grad phi) phi_j   (every lane stays for the block stores)
  double M[8] = {0, 0, 0, 0, 0, 0, 0, 0}, S[8] = {0, 0, 0, 0, 0, 0, 0, 0}, Dr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (valid) {
    const double* vg = vgeo + lc * HX_NV * HX_VG;
    const int x = i;
#pragma unroll 1
    for (int P = 0; P < 8; ++P) {
      const double* gt = vg + P * HX_VG;
      const double wd = gt[24];
      const double gi[3] = {gt[3 * x], gt[3 * x + 1], gt[3 * x + 2]};
      const int xi = x ^ P;
      const double phi_i = ((xi & 1) ? HX_G0 : HX_G1) * ((xi & 2) ? HX_G0 : HX_G1) * ((xi & 4) ? HX_G0 : HX_G1);
      double g[8][3], pj[8];
      hx_values(P, pj);
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int d = 0; d < 3; ++d) g[j][d] = gt[3 * j + d];
      double gp[3] = {0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int d = 0; d < 3; ++d) gp[d] += ph[j] * g[j][d];
      const double dr = wd * (gi[0] * gp[0] + gi[1] * gp[1] + gi[2] * gp[2]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        M[j] += wd * phi_i * pj[j];
        S[j] += wd * (gi[0] * g[j][0] + gi[1] * g[j][1] + gi[2] * g[j][2]);
        Dr[j] += dr * pj[j];
      }
    }
  }
  int slot_self, slot[HX_NFC];
  hx_slots(T, nb, slot_self, slot);
  __syncthreads();
  double* scr = vgeo + tid * HX_SCR;
  // right-hand sides: mass times (c_k / dt + f_k), then the membrane integrals of the row's function
  double rhs[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double v = lrec[(lc * HX_NV + j) * DG_RPITCH + KN_CSLOT(k)] * C.inv_dt;
      if (D.fsrc && s == 0 && valid) v += D.fsrc[(size_t)k * D.n_dof + T * HX_NV + j];
      acc += M[j] * v;
    }
    rhs[k] = acc;
  }
  double UP[8] = {0, 0, 0, 0, 0, 0, 0, 0}, UM[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // own block: drift leaving / entering
  auto facet = [&](auto F_) {
    constexpr int f = decltype(F_)::value, a = f >> 1, b = f & 1;
    const bool have = valid && nb[f] >= 0;
    const int kind = fi[f] & 3;
    const bool on = ((i >> a) & 1) == b;
    const int mi = ((i >> hx_ax1(a)) & 1) | (((i >> hx_ax2(a)) & 1) << 1);
    const double* fd = fsum + (lc * HX_NFC + f) * FS;
    double B0[4] = {0, 0, 0, 0}, B1[4] = {0, 0, 0, 0}, VP[4] = {0, 0, 0, 0}, VM[4] = {0, 0, 0, 0};
    if (have && kind != 1 && on) {   // the membrane couples the two sides through the right-hand side only (zero block)
#pragma unroll
      for (int k = 0; k < KS; ++k) rhs[k] += fd[mi * KS + k];
    }
    if (have && kind == 1) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      double N[4], ds[4], dt[4];
      hx_shapes(p, N, ds, dt);
      const int xm_ = mi ^ p;
      const double wa = (xm_ & 1) ? HX_G0 : HX_G1, wb = (xm_ & 2) ? HX_G0 : HX_G1;
      const double Ni = wa * wb, dsi = (mi & 1) ? wb : -wb, dti = (mi & 2) ? wa : -wa;
      const double* o = fd + 10 * p;
      const double W = o[0], ih = o[1], aTs = o[2], aTt = o[3], cT = o[4], aNs = o[5], aNt = o[6], cN = o[7];
      const double phi_i = on ? Ni : 0.0;
      const double gni = on ? dsi * aTs + dti * aTt - Ni * cT : Ni * cT;
      const double pen = C.gamma * ih * W * phi_i;
      const double c1 = -0.5 * W * phi_i, c2 = -0.5 * W * gni;
      const double b0 = -C.psi * 0.5 * (o[8] + o[9]) * W * phi_i;   // beta_k W phi_i = z_k D_k b0
      const double bp = fmax(b0, 0.0), bm = fmin(b0, 0.0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double gT = ds[m] * aTs + dt[m] * aTt - N[m] * cT;
        const double gN = ds[m] * aNs + dt[m] * aNt - N[m] * cN;
        S[hx_facet_vertex(a, b, m)] += c1 * gT + (c2 + pen) * N[m];
        S[hx_facet_vertex(a, b, m) ^ (1 << a)] += c1 * (N[m] * cT);
        B0[m] += c1 * gN - (c2 + pen) * N[m];
        B1[m] += c1 * (N[m] * cN);
        UP[hx_facet_vertex(a, b, m)] += bp * N[m];
        UM[hx_facet_vertex(a, b, m)] += bm * N[m];
        VP[m] += bp * N[m];
        VM[m] += bm * N[m];
      }
    }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (have) {
        const double Dk = C.D[s][k], zD = C.z[k] * Dk;
        double b0k[4], b1k[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          // upwind: the neighbour's trace carries the flux where the drift enters T (beta < 0)
          b0k[m] = Dk * B0[m] + zD * (C.z[k] > 0.0 ? VM[m] : VP[m]);
          b1k[m] = Dk * B1[m];
        }
        hx_put_nbr(scr, fi[f], b0k, b1k);
      }
      hx_flush(vgeo, dsts, tid, have, rp + slot[f] * HX_NV, D.A_knp + (size_t)k * D.nnz);
    }
  };
  facet(std::integral_constant<int, 0>{}); facet(std::integral_constant<int, 1>{});
  facet(std::integral_constant<int, 2>{}); facet(std::integral_constant<int, 3>{});
  facet(std::integral_constant<int, 4>{}); facet(std::integral_constant<int, 5>{});
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const double Dk = C.D[s][k], zD = C.z[k] * Dk, zpD = zD * C.psi;
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = M[j] * C.inv_dt + Dk * S[j] + zpD * Dr[j] + zD * (C.z[k] > 0.0 ? UP[j] : UM[j]);
    if (valid) hx_put_own(scr, v);
    hx_flush(vgeo, dsts, tid, valid, rp + slot_self * HX_NV, D.A_knp + (size_t)k * D.nnz);
    if (valid) D.b_knp[(size_t)k * D.n_dof + row] = rhs[k];
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Box meshes: every cell an orthogonal parallelepiped (checked at knpemi_dg_create; every mesh of the reference's 3-D
// driver is one, make_mesh_3D.py:100-102).  Then the Jacobian is constant and diagonal in the cell's own frame,
//   grad phi_i . grad phi_j = sum_t dref_i[t] dref_j[t] / |e_t|^2        (e_t: the cell's edge along axis t),
// the depth vector of either cell at a facet is parallel to the normal, so a_s = a_t = 0 and c = -/+ 1 / |e_a| is the
// same at the four facet points: grad phi_(m, 0) . n = -N_m c, grad phi_(m, 1) . n = N_m c.  Every entry of a row that a
// facet contributes to is then  sum_p e(p) N_m(p)  with FOUR row scalars e per point (own block on / behind the facet,
// neighbour block on / behind the facet) -- a 4 x 4 product per kind instead of sixteen three-term entries per point --
// the facet summary shrinks to 4 + 4 (2) numbers per point, the gradient tables disappear, and with 18 (15) kB of LDS
// and half the registers two workgroups share a SIMD.  Same numbers as the general kernels up to rounding
// (tests/test_dg_gpu.py runs both on the same box mesh).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int HX_FS_BOX_EMI = 21;    // W ih cT cN | (kT kN JT JN) x 4 points
constexpr int HX_FS_BOX_KNP = 13;    // W ih cT cN | (gT gN) x 4 points     (membrane: 4 x KS integrals)

struct BoxCell {
  double h2inv[3];   // 1 / |e_t|^2
  double wd;         // |det J| / 8
};
__device__ __forceinline__ BoxCell hx_box_cell(const double* r0, const double* r1, const double* r2, const double* r4) {
  BoxCell B;
  double vol = 1.0;
  const double* rt[3] = {r1, r2, r4};
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const double ex = rt[t][0] - r0[0], ey = rt[t][1] - r0[1], ez = rt[t][2] - r0[2];
    const double l2 = ex * ex + ey * ey + ez * ez, ri = fast_rsqrt(l2);
    B.h2inv[t] = ri * ri;
    vol *= l2 * ri;
  }
  B.wd = 0.125 * vol;
  return B;
}

// |edge| between two records
__device__ __forceinline__ double hx_edge_len(const double* a, const double* b) {
  const double ex = b[0] - a[0], ey = b[1] - a[1], ez = b[2] - a[2];
  const double l2 = ex * ex + ey * ey + ez * ez;
  return l2 * fast_rsqrt(l2);
}

// acc[m] += sum over the four facet points of e[p] N_m(p)
__device__ __forceinline__ void hx_fold(const double (&e)[4], double (&acc)[4]) {
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const double N = (((m ^ p) & 1) ? HX_G0 : HX_G1) * ((((m ^ p) >> 1) & 1) ? HX_G0 : HX_G1);
      acc[m] += e[p] * N;
    }
}

// The workgroup's own records (64 rows x 64 bytes, contiguous) into LDS: lane l of pass k moves bytes 16 (64 k + l) .. + 16,
// 16 whole lines per instruction (a lane reading its own record alone touches 64 lines per instruction).
__device__ __forceinline__ void hx_stage_own(double* lrec, const double* __restrict__ rec, int cell0, int ncell, int tid) {
  const double2* blk = reinterpret_cast<const double2*>(rec + (size_t)cell0 * HX_NV * KN_REC);
  double2 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = k * DG_BLOCK + tid;
    v[k] = make_double2(0.0, 0.0);
    if ((e >> 2) < ncell * HX_NV) v[k] = blk[e];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = k * DG_BLOCK + tid;
    *reinterpret_cast<double2*>(lrec + (e >> 2) * DG_RPITCH + 2 * (e & 3)) = v[k];
  }
}

// The neighbour cells of the workgroup's (cell, facet) tasks, read WHOLE.  A task needs the four vertices of the
// neighbour on the facet and the four behind them -- the neighbour's eight records, 512 contiguous bytes.  One
// instruction reads the neighbours of two tasks (32 lanes x 16 bytes each, 16 whole lines); a task lane gathering the
// same values for itself issues ~70 instructions of 48 lines each, and the CU's address unit, not HBM, is what the phase
// waited for (s_memtime stamps, round 4: 37 % of the kernel).  Tasks 2 b and 2 b + 1 belong to cell b / 3 of the
// workgroup.  `tgo`: on lane t < 48 the neighbour cell of task t, or -1.  consume(cell, task, record, part, v, sub):
// v = doubles 2 part, 2 part + 1 of the record; called by all 64 lanes (zeros for an absent neighbour), `sub` = the
// cell's sub-domain, a scalar.
template <int GROUPS, class F>
__device__ __forceinline__ void hx_neighbour_cells(const double* __restrict__ rec, int tgo, int tsub, int tid, F&& consume) {
  constexpr int NB = HX_TASKS / 2, PER = NB / GROUPS;      // instructions in flight together: 4 registers each
  static_assert(NB % GROUPS == 0, "groups of equal size");
  const int half = tid >> 5, within = tid & 31;
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    double2 v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int b = PER * g + u;
      const int cA = __builtin_amdgcn_readlane(tgo, 2 * b), cB = __builtin_amdgcn_readlane(tgo, 2 * b + 1);
      const int c = half ? cB : cA;
      v[u] = make_double2(0.0, 0.0);
      if (c >= 0) v[u] = reinterpret_cast<const double2*>(rec + (size_t)c * (HX_NV * KN_REC))[within];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int b = PER * g + u;
      const int sub = __builtin_amdgcn_readlane(tsub, 6 * (b / 3));
      consume(2 * b + half, within >> 2, within & 3, v[u], sub);
    }
  }
}

// lane p of every quad of lanes gets x of the quad's lane SRC
template <int SRC>
__device__ __forceinline__ double hx_quad_bcast(double x) {
  constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);     // quad_perm:[SRC, SRC, SRC, SRC]
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// Diagnostic build (make CXXFLAGS+=-DKN_HX_STAMPS): the box kernels add up s_memtime differences between their phases
// (one atomic per phase and wave); the launcher prints the averages every 32 launches.
#ifdef KN_HX_STAMPS
__device__ unsigned long long hx_stamp_acc[1024 * 32];     // [slot = workgroup % 1024][phase]
#define HX_STAMP_BEGIN unsigned long long hx_t_ = __builtin_amdgcn_s_memtime();
#define HX_STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                         if (threadIdx.x == 0) atomicAdd(&hx_stamp_acc[(blockIdx.x & 1023) * 32 + (i)], n_ - hx_t_); hx_t_ = n_; } while (0)
#else
#define HX_STAMP_BEGIN
#define HX_STAMP(i) do {} while (0)
#endif

__global__ __launch_bounds__(DG_BLOCK, 3) void dg_emi_hex_box_kernel(DgDev D, const DgConsts* __restrict__ Cp, int chunk, int splitting) {
  __shared__ double lrec[DG_BLOCK * DG_RPITCH];   // the records; after the second barrier the scratch lines of the block stores
  __shared__ __attribute__((aligned(16))) double fsum[HX_TASKS * HX_FS_BOX_EMI];   // first: (kappa, sigma) sums of the neighbours' vertices
  __shared__ int dsts[DG_BLOCK];
  double* const scrs = lrec;                      // (HX_SCR == DG_RPITCH; the rows are done with the records by then)
  const DgConsts& C = *Cp;
  const int cell0 = dg_block_index(blockIdx.x, chunk) * HX_CELLS;
  if (cell0 >= D.n_cell) return;
  const int ncell = min(HX_CELLS, D.n_cell - cell0);
  const int tid = threadIdx.x;
  HX_STAMP_BEGIN
  const int lc = tid >> 3, i = tid & 7, T = cell0 + lc, row = T * HX_NV + i;
  const bool valid = lc < ncell;
  const int tc = tid / HX_NFC, tf = tid - tc * HX_NFC;
  const bool task = tid < HX_TASKS && tc < ncell;
  int tnb = -1, tmf = -1, tsub = 0;
  unsigned tfi = 0;
  if (task) {
    tnb = D.nbr[(cell0 + tc) * HX_NFC + tf];
    tfi = D.finfo[(cell0 + tc) * HX_NFC + tf];
    tmf = D.mfid[(cell0 + tc) * HX_NFC + tf];
    tsub = D.cell_sub[cell0 + tc];
  }
  hx_stage_own(lrec, D.rec, cell0, ncell, tid);
  int nb[HX_NFC] = {-1, -1, -1, -1, -1, -1};
  unsigned fi[HX_NFC] = {0, 0, 0, 0, 0, 0};
  int s = 0, rp = 0;
  if (valid) {
    s = D.cell_sub[T];
    rp = D.rowptr[row];
#pragma unroll
    for (int f = 0; f < HX_NFC; ++f) { nb[f] = D.nbr[T * HX_NFC + f]; fi[f] = D.finfo[T * HX_NFC + f]; }
  }
  HX_STAMP(0);      // own records, topology
  // interior facets: kappa- and sigma-weighted concentration sums of the neighbour's eight vertices, summed in the order the
  // rows of the neighbour sum their own (ion 0, 1, 2, 3: doubles 4, 5, 6, 3 of the record, i.e. lanes 2, 2, 3, 1 of the
  // record's quad); lane 1 of the quad ends up with both and stores them
  const bool interior = task && tnb >= 0 && (tfi & 3) == 1;
  double hN = 1.0;
  if (interior) hN = D.box_h[(size_t)tnb * 3 + (((tfi >> 2) & 7) >> 1)];
  hx_neighbour_cells<1>(D.rec, interior ? tnb : -1, tsub, tid, [&](int t, int j, int part, double2 v, int sub) {
    const double* kp = C.kap[sub];
    const double* sp = C.sig[sub];
    double ks = fma(kp[1], v.y, kp[0] * v.x), gs = fma(sp[1], v.y, sp[0] * v.x);      // lane 2: ions 0, 1
    ks = fma(kp[2], v.x, hx_quad_bcast<2>(ks)); gs = fma(sp[2], v.x, hx_quad_bcast<2>(gs));      // lane 3: + ion 2
    ks = fma(kp[3], v.y, hx_quad_bcast<3>(ks)); gs = fma(sp[3], v.y, hx_quad_bcast<3>(gs));      // lane 1: + ion 3
    if (part == 1) *reinterpret_cast<double2*>(fsum + (t * HX_NV + j) * 2) = make_double2(ks, gs);
  });
  __syncthreads();
  HX_STAMP(1);      // neighbour cells
  double kN[4] = {0, 0, 0, 0}, dsN[4] = {0, 0, 0, 0};
  if (interior) {
    const int ao = ((tfi >> 2) & 7) >> 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int jn = (tfi >> (5 + 3 * m)) & 7, jt = jn ^ (1 << ao);
      const double2 n0 = *reinterpret_cast<const double2*>(fsum + (tid * HX_NV + jn) * 2);
      kN[m] = n0.x;
      dsN[m] = fsum[(tid * HX_NV + jt) * 2 + 1] - n0.y;
    }
  }
  hx_lds_sync();     // (the sums are in registers before the summaries overwrite them)
  if (task && tnb >= 0) {
    const int a = tf >> 1, b = tf & 1, kind = tfi & 3, ts_ = tsub;
    const double* r0[4];
    double kT[4], dsT[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int jv = hx_facet_vertex(a, b, m), jo = jv ^ (1 << a);
      r0[m] = lrec + (tc * HX_NV + jv) * DG_RPITCH;
      const double* r1 = lrec + (tc * HX_NV + jo) * DG_RPITCH;
      double k0 = 0.0, g0 = 0.0, g1 = 0.0;
#pragma unroll
      for (int q = 0; q < KN_MAXK; ++q) {
        k0 += C.kap[ts_][q] * r0[m][KN_CSLOT(q)]; g0 += C.sig[ts_][q] * r0[m][KN_CSLOT(q)]; g1 += C.sig[ts_][q] * r1[KN_CSLOT(q)];
      }
      kT[m] = k0; dsT[m] = g1 - g0;
    }
    const double* cb = lrec + tc * HX_NV * DG_RPITCH;
    const double hT = hx_edge_len(cb, cb + (1 << a) * DG_RPITCH);
    const double A = hx_edge_len(cb, cb + (1 << hx_ax1(a)) * DG_RPITCH) * hx_edge_len(cb, cb + (1 << hx_ax2(a)) * DG_RPITCH);
    double* my = fsum + tid * HX_FS_BOX_EMI;
    if (kind == 1) {
      const double cT = -fast_rcp(hT), cN = fast_rcp(hN);
      my[0] = 0.25 * A; my[1] = 0.5 * (cN - cT); my[2] = cT; my[3] = cN;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        double N[4], ds[4], dt[4];
        hx_shapes(p, N, ds, dt);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) { a0 += N[m] * kT[m]; a1 += N[m] * kN[m]; a2 += N[m] * dsT[m]; a3 += N[m] * dsN[m]; }
        my[4 + 4 * p] = a0; my[5 + 4 * p] = a1; my[6 + 4 * p] = a2 * cT; my[7 + 4 * p] = a3 * cN;
      }
    } else {
      double g[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int q = tmf * 4 + ((tfi >> (17 + 2 * m)) & 3);
        double v = D.phiM[q];
        if (!splitting) {
          double it = 0.0;
          for (int k = 0; k < C.K; ++k) it += D.Ich[(size_t)k * D.nq + q];
          v -= it / C.C_phi;
        }
        g[m] = v;
      }
      my[0] = 0.25 * A * C.C_phi;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        double N[4], ds[4], dt[4];
        hx_shapes(p, N, ds, dt);
        my[4 + 4 * p] = N[0] * g[0] + N[1] * g[1] + N[2] * g[2] + N[3] * g[3];
      }
    }
  }
  HX_STAMP(2);      // facet summaries
  double kap[8], sg[8];
  BoxCell Bc{};
  if (valid) {
    const double* cb = lrec + lc * HX_NV * DG_RPITCH;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double* r = cb + j * DG_RPITCH;
      double k0 = 0.0, g0 = 0.0;
#pragma unroll
      for (int q = 0; q < KN_MAXK; ++q) { k0 += C.kap[s][q] * r[KN_CSLOT(q)]; g0 += C.sig[s][q] * r[KN_CSLOT(q)]; }
      kap[j] = k0; sg[j] = g0;
    }
    Bc = hx_box_cell(cb, cb + DG_RPITCH, cb + 2 * DG_RPITCH, cb + 4 * DG_RPITCH);
  }
  __syncthreads();
  HX_STAMP(17);     // coefficients, barrier
  double A[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double rhs = 0.0;
  auto point = [&](auto P_) {
    constexpr int P = decltype(P_)::value;
    const int xi = i ^ P;
    const double a = (xi & 1) ? HX_G0 : HX_G1, b = (xi & 2) ? HX_G0 : HX_G1, c = (xi & 4) ? HX_G0 : HX_G1;
    // u[t] = dref_i[t] / |e_t|^2
    const double u0 = ((i & 1) ? b : -b) * c * Bc.h2inv[0], u1 = ((i & 2) ? a : -a) * c * Bc.h2inv[1],
                 u2 = ((i & 4) ? a : -a) * b * Bc.h2inv[2];
    double kq = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0, dd[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      constexpr double G[2] = {HX_G1, HX_G0};
      const int x = j ^ P;
      const double wa = G[x & 1], wb = G[(x >> 1) & 1], wc = G[(x >> 2) & 1];
      const double d0 = ((j & 1) ? 1.0 : -1.0) * wb * wc, d1 = ((j & 2) ? 1.0 : -1.0) * wa * wc, d2 = ((j & 4) ? 1.0 : -1.0) * wa * wb;
      kq += kap[j] * (wa * wb * wc);
      s0 += sg[j] * d0; s1 += sg[j] * d1; s2 += sg[j] * d2;
      dd[j] = u0 * d0 + u1 * d1 + u2 * d2;
    }
    const double wk = Bc.wd * kq;
#pragma unroll
    for (int j = 0; j < 8; ++j) A[j] += wk * dd[j];
    rhs -= Bc.wd * (u0 * s0 + u1 * s1 + u2 * s2);
  };
  point(std::integral_constant<int, 0>{}); point(std::integral_constant<int, 1>{});
  point(std::integral_constant<int, 2>{}); point(std::integral_constant<int, 3>{});
  point(std::integral_constant<int, 4>{}); point(std::integral_constant<int, 5>{});
  point(std::integral_constant<int, 6>{}); point(std::integral_constant<int, 7>{});
  HX_STAMP(3);      // volume terms
  int slot_self, slot[HX_NFC];
  hx_slots(T, nb, slot_self, slot);
  double* scr = scrs + tid * HX_SCR;
  auto facet = [&](auto F_) {
    constexpr int f = decltype(F_)::value, a = f >> 1, b = f & 1;
    const bool have = valid && nb[f] >= 0;
    const int kind = fi[f] & 3;
    const bool on = ((i >> a) & 1) == b;
    const int mi = ((i >> hx_ax1(a)) & 1) | (((i >> hx_ax2(a)) & 1) << 1);
    const double* fd = fsum + (lc * HX_NFC + f) * HX_FS_BOX_EMI;
    double B0[4] = {0, 0, 0, 0}, B1[4] = {0, 0, 0, 0};
    const double W = fd[0];
    if (have && kind == 1) {
      const double ih = fd[1], cT = fd[2], cN = fd[3];
      const double sgi = on ? -cT : cT, onf = on ? 1.0 : 0.0;
      double eo[4], et[4], e0[4], e1[4], Ao[4] = {0, 0, 0, 0}, At[4] = {0, 0, 0, 0};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int xm_ = mi ^ p;
        const double Ni = ((xm_ & 1) ? HX_G0 : HX_G1) * ((xm_ & 2) ? HX_G0 : HX_G1);
        const double kT = fd[4 + 4 * p], kN = fd[5 + 4 * p];
        const double phi_i = onf * Ni, gni = Ni * sgi;
        const double pen = C.gamma * ih * (0.5 * (kT + kN)) * W * phi_i;
        const double c1 = -0.5 * W * kT * phi_i, c2 = -0.5 * W * kT * gni, c1n = -0.5 * W * kN * phi_i;
        eo[p] = c2 + pen - c1 * cT;
        et[p] = c1 * cT;
        e0[p] = -(c2 + pen) - c1n * cN;
        e1[p] = c1n * cN;
        rhs += W * 0.5 * (fd[6 + 4 * p] + fd[7 + 4 * p]) * phi_i;
      }
      hx_fold(eo, Ao); hx_fold(et, At); hx_fold(e0, B0); hx_fold(e1, B1);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        A[hx_facet_vertex(a, b, m)] += Ao[m];
        A[hx_facet_vertex(a, b, m) ^ (1 << a)] += At[m];
      }
    } else if (have && on) {
      double e[4], Am[4] = {0, 0, 0, 0}, g = 0.0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int xm_ = mi ^ p;
        e[p] = W * (((xm_ & 1) ? HX_G0 : HX_G1) * ((xm_ & 2) ? HX_G0 : HX_G1));
        g += e[p] * fd[4 + 4 * p];
      }
      hx_fold(e, Am);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        A[hx_facet_vertex(a, b, m)] += Am[m];
        B0[m] = -Am[m];
      }
      rhs += kind == 3 ? g : -g;
    }
    HX_STAMP(4 + 2 * f);      // facet f: arithmetic
    if (have) hx_put_nbr(scr, fi[f], B0, B1);
    hx_flush(scrs, dsts, tid, have, rp + slot[f] * HX_NV, D.A_emi);
    HX_STAMP(5 + 2 * f);      // facet f: exchange + stores issued
  };
  facet(std::integral_constant<int, 0>{}); facet(std::integral_constant<int, 1>{});
  facet(std::integral_constant<int, 2>{}); facet(std::integral_constant<int, 3>{});
  facet(std::integral_constant<int, 4>{}); facet(std::integral_constant<int, 5>{});
  if (valid) hx_put_own(scr, A);
  hx_flush(scrs, dsts, tid, valid, rp + slot_self * HX_NV, D.A_emi);
  if (valid) D.b_emi[row] = rhs;
  HX_STAMP(16);     // own block
}

template <int KS>
__global__ __launch_bounds__(DG_BLOCK, 2) void dg_knp_hex_box_kernel(DgDev D, const DgConsts* __restrict__ Cp, int chunk, int splitting) {
  constexpr int FS = HX_FS_BOX_KNP;
  __shared__ double lrec[DG_BLOCK * DG_RPITCH];
  __shared__ __attribute__((aligned(16))) double fsum[HX_TASKS * FS];   // first: the potentials of the neighbours' vertices
  __shared__ double scrs[DG_BLOCK * HX_SCR];
  __shared__ int dsts[DG_BLOCK];
  const DgConsts& C = *Cp;
  const int cell0 = dg_block_index(blockIdx.x, chunk) * HX_CELLS;
  if (cell0 >= D.n_cell) return;
  const int ncell = min(HX_CELLS, D.n_cell - cell0);
  const int tid = threadIdx.x;
  const int lc = tid >> 3, i = tid & 7, T = cell0 + lc, row = T * HX_NV + i;
  const bool valid = lc < ncell;
  const int tc = tid / HX_NFC, tf = tid - tc * HX_NFC;
  const bool task = tid < HX_TASKS && tc < ncell;
  int tnb = -1, tmf = -1, tsub = 0;
  unsigned tfi = 0;
  if (task) {
    tnb = D.nbr[(cell0 + tc) * HX_NFC + tf];
    tfi = D.finfo[(cell0 + tc) * HX_NFC + tf];
    tmf = D.mfid[(cell0 + tc) * HX_NFC + tf];
    tsub = D.cell_sub[cell0 + tc];
  }
  hx_stage_own(lrec, D.rec, cell0, ncell, tid);
  int nb[HX_NFC] = {-1, -1, -1, -1, -1, -1};
  unsigned fi[HX_NFC] = {0, 0, 0, 0, 0, 0};
  int s = 0, rp = 0;
  if (valid) {
    s = D.cell_sub[T];
    rp = D.rowptr[row];
#pragma unroll
    for (int f = 0; f < HX_NFC; ++f) { nb[f] = D.nbr[T * HX_NFC + f]; fi[f] = D.finfo[T * HX_NFC + f]; }
  }
  // the potential at the eight vertices of every task's neighbour (hx_neighbour_cells): double 7 of a record sits in lane 3 of
  // the record's quad
  double hN = 1.0;
  if (task && tnb >= 0 && (tfi & 3) == 1) hN = D.box_h[(size_t)tnb * 3 + (((tfi >> 2) & 7) >> 1)];
  hx_neighbour_cells<1>(D.rec, task ? tnb : -1, tsub, tid, [&](int t, int j, int part, double2 v, int) {
    if (part == 3) fsum[t * HX_NV + j] = v.y;
  });
  __syncthreads();
  double pN0[4] = {0, 0, 0, 0}, dpN[4] = {0, 0, 0, 0};
  if (task && tnb >= 0) {
    const int ao = ((tfi >> 2) & 7) >> 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int jn = (tfi >> (5 + 3 * m)) & 7, jt = jn ^ (1 << ao);
      pN0[m] = fsum[tid * HX_NV + jn];
      if ((tfi & 3) == 1) dpN[m] = fsum[tid * HX_NV + jt] - pN0[m];
    }
  }
  hx_lds_sync();     // (the potentials are in registers before the summaries overwrite them)
  if (task && tnb >= 0) {
    const int a = tf >> 1, b = tf & 1, kind = tfi & 3, ts_ = tsub;
    const double* rT[4];
    double pT0[4], dpT[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int jv = hx_facet_vertex(a, b, m), jo = jv ^ (1 << a);
      rT[m] = lrec + (tc * HX_NV + jv) * DG_RPITCH;
      const double* r1 = lrec + (tc * HX_NV + jo) * DG_RPITCH;
      pT0[m] = rT[m][7];
      dpT[m] = r1[7] - rT[m][7];
    }
    const double* cb = lrec + tc * HX_NV * DG_RPITCH;
    const double hT = hx_edge_len(cb, cb + (1 << a) * DG_RPITCH);
    const double A = hx_edge_len(cb, cb + (1 << hx_ax1(a)) * DG_RPITCH) * hx_edge_len(cb, cb + (1 << hx_ax2(a)) * DG_RPITCH);
    double* my = fsum + tid * FS;
    if (kind == 1) {
      const double cT = -fast_rcp(hT), cN = fast_rcp(hN);
      my[0] = 0.25 * A; my[1] = 0.5 * (cN - cT); my[2] = cT; my[3] = cN;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        double N[4], ds[4], dt[4];
        hx_shapes(p, N, ds, dt);
        double gT = 0.0, gN = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) { gT += N[m] * dpT[m]; gN += N[m] * dpN[m]; }
        my[4 + 2 * p] = gT * cT; my[5 + 2 * p] = gN * cN;
      }
    } else {
      // membrane facet (knpWeakForm.py:168-214), 4 x 4 rule, once per (cell, facet); the surface element is constant
      double jm[4], pm[4], It[4], Ik[4][KN_MAXK], Gk[4][KS];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        jm[m] = kind == 2 ? pN0[m] - pT0[m] : pT0[m] - pN0[m];
        const int q = tmf * 4 + ((tfi >> (17 + 2 * m)) & 3);
        pm[m] = D.phiM[q];
        double it = 0.0;
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) {
          Ik[m][k] = k < C.K ? D.Ich[(size_t)k * D.nq + q] : 0.0;
          it += Ik[m][k];
        }
        It[m] = it;
#pragma unroll
        for (int k = 0; k < KS; ++k) Gk[m][k] = 0.0;
      }
      const double sgn = kind == 2 ? 1.0 : -1.0;
      const double* qw = D.qtab;
      const double* qN = D.qtab + D.nquad;
      for (int q = 0; q < D.nquad; ++q) {
        double cq[KN_MAXK], iq[KN_MAXK], jq = 0.0, pq = 0.0, itq = 0.0, Nq[4];
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) { cq[k] = 0.0; iq[k] = 0.0; }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          Nq[m] = qN[q * 4 + m];
#pragma unroll
          for (int k = 0; k < KN_MAXK; ++k) { cq[k] += Nq[m] * rT[m][KN_CSLOT(k)]; iq[k] += Nq[m] * Ik[m][k]; }
          jq += Nq[m] * jm[m]; pq += Nq[m] * pm[m]; itq += Nq[m] * It[m];
        }
        double asum = 0.0;
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) asum += C.az2D[ts_][k] * cq[k];
        const double w = sgn * qw[q] * A;
        double drive = jq - pq;
        if (splitting) drive -= (C.dt / C.C_M) * itq;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const double al = C.az2D[ts_][k] * cq[k] / asum;
          const double fz = 1.0 / (C.F * C.z[k]);
          const double fq = w * (al * C.C_M * fz * C.inv_dt * drive + iq[k] * fz);
#pragma unroll
          for (int m = 0; m < 4; ++m) Gk[m][k] += Nq[m] * fq;
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int k = 0; k < KS; ++k) my[m * KS + k] = Gk[m][k];
    }
  }
  double ph[8];
  BoxCell Bc{};
  if (valid) {
    const double* cb = lrec + lc * HX_NV * DG_RPITCH;
#pragma unroll
    for (int j = 0; j < 8; ++j) ph[j] = cb[j * DG_RPITCH + 7];
    Bc = hx_box_cell(cb, cb + DG_RPITCH, cb + 2 * DG_RPITCH, cb + 4 * DG_RPITCH);
  }
  __syncthreads();
  double M[8] = {0, 0, 0, 0, 0, 0, 0, 0}, S[8] = {0, 0, 0, 0, 0, 0, 0, 0}, Dr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto point = [&](auto P_) {
    constexpr int P = decltype(P_)::value;
    const int xi = i ^ P;
    const double a = (xi & 1) ? HX_G0 : HX_G1, b = (xi & 2) ? HX_G0 : HX_G1, c = (xi & 4) ? HX_G0 : HX_G1;
    const double u0 = ((i & 1) ? b : -b) * c * Bc.h2inv[0], u1 = ((i & 2) ? a : -a) * c * Bc.h2inv[1],
                 u2 = ((i & 4) ? a : -a) * b * Bc.h2inv[2];
    const double phi_i = a * b * c;
    double g0 = 0.0, g1 = 0.0, g2 = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      constexpr double G[2] = {HX_G1, HX_G0};
      const int x = j ^ P;
      const double wa = G[x & 1], wb = G[(x >> 1) & 1], wc = G[(x >> 2) & 1];
      g0 += ph[j] * (((j & 1) ? 1.0 : -1.0) * wb * wc);
      g1 += ph[j] * (((j & 2) ? 1.0 : -1.0) * wa * wc);
      g2 += ph[j] * (((j & 4) ? 1.0 : -1.0) * wa * wb);
    }
    const double dr = Bc.wd * (u0 * g0 + u1 * g1 + u2 * g2), wm = Bc.wd * phi_i;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      constexpr double G[2] = {HX_G1, HX_G0};
      const int x = j ^ P;
      const double wa = G[x & 1], wb = G[(x >> 1) & 1], wc = G[(x >> 2) & 1];
      const double d0 = ((j & 1) ? 1.0 : -1.0) * wb * wc, d1 = ((j & 2) ? 1.0 : -1.0) * wa * wc, d2 = ((j & 4) ? 1.0 : -1.0) * wa * wb;
      const double pj = wa * wb * wc;
      M[j] += wm * pj;
      S[j] += Bc.wd * (u0 * d0 + u1 * d1 + u2 * d2);
      Dr[j] += dr * pj;
    }
  };
  point(std::integral_constant<int, 0>{}); point(std::integral_constant<int, 1>{});
  point(std::integral_constant<int, 2>{}); point(std::integral_constant<int, 3>{});
  point(std::integral_constant<int, 4>{}); point(std::integral_constant<int, 5>{});
  point(std::integral_constant<int, 6>{}); point(std::integral_constant<int, 7>{});
  int slot_self, slot[HX_NFC];
  hx_slots(T, nb, slot_self, slot);
  double* scr = scrs + tid * HX_SCR;
  double rhs[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double v = lrec[(lc * HX_NV + j) * DG_RPITCH + KN_CSLOT(k)] * C.inv_dt;
      if (D.fsrc && s == 0 && valid) v += D.fsrc[(size_t)k * D.n_dof + T * HX_NV + j];
      acc += M[j] * v;
    }
    rhs[k] = acc;
  }
  double UP[8] = {0, 0, 0, 0, 0, 0, 0, 0}, UM[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto facet = [&](auto F_) {
    constexpr int f = decltype(F_)::value, a = f >> 1, b = f & 1;
    const bool have = valid && nb[f] >= 0;
    const int kind = fi[f] & 3;
    const bool on = ((i >> a) & 1) == b;
    const int mi = ((i >> hx_ax1(a)) & 1) | (((i >> hx_ax2(a)) & 1) << 1);
    const double* fd = fsum + (lc * HX_NFC + f) * FS;
    double B0[4] = {0, 0, 0, 0}, B1[4] = {0, 0, 0, 0}, VP[4] = {0, 0, 0, 0}, VM[4] = {0, 0, 0, 0};
    if (have && kind != 1 && on) {   // the membrane couples the two sides through the right-hand side only (zero block)
#pragma unroll
      for (int k = 0; k < KS; ++k) rhs[k] += fd[mi * KS + k];
    }
    if (have && kind == 1) {
    const double W = fd[0], ih = fd[1], cT = fd[2], cN = fd[3];
    const double sgi = on ? -cT : cT, onf = on ? 1.0 : 0.0;
    double eo[4], et[4], e0[4], e1[4], ep[4], em[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int xm_ = mi ^ p;
      const double Ni = ((xm_ & 1) ? HX_G0 : HX_G1) * ((xm_ & 2) ? HX_G0 : HX_G1);
      const double phi_i = onf * Ni, gni = Ni * sgi;
      const double pen = C.gamma * ih * W * phi_i;
      const double c1 = -0.5 * W * phi_i, c2 = -0.5 * W * gni;
      eo[p] = c2 + pen - c1 * cT;
      et[p] = c1 * cT;
      e0[p] = -(c2 + pen) - c1 * cN;
      e1[p] = c1 * cN;
      const double b0 = -C.psi * 0.5 * (fd[4 + 2 * p] + fd[5 + 2 * p]) * W * phi_i;
      ep[p] = fmax(b0, 0.0);
      em[p] = fmin(b0, 0.0);
    }
    double So[4] = {0, 0, 0, 0}, St[4] = {0, 0, 0, 0};
    hx_fold(eo, So); hx_fold(et, St); hx_fold(e0, B0); hx_fold(e1, B1); hx_fold(ep, VP); hx_fold(em, VM);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      S[hx_facet_vertex(a, b, m)] += So[m];
      S[hx_facet_vertex(a, b, m) ^ (1 << a)] += St[m];
      UP[hx_facet_vertex(a, b, m)] += VP[m];
      UM[hx_facet_vertex(a, b, m)] += VM[m];
    }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      if (have) {
        const double Dk = C.D[s][k], zD = C.z[k] * Dk;
        double b0k[4], b1k[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          b0k[m] = Dk * B0[m] + zD * (C.z[k] > 0.0 ? VM[m] : VP[m]);
          b1k[m] = Dk * B1[m];
        }
        hx_put_nbr(scr, fi[f], b0k, b1k);
      }
      hx_flush(scrs, dsts, tid, have, rp + slot[f] * HX_NV, D.A_knp + (size_t)k * D.nnz);
    }
  };
  facet(std::integral_constant<int, 0>{}); facet(std::integral_constant<int, 1>{});
  facet(std::integral_constant<int, 2>{}); facet(std::integral_constant<int, 3>{});
  facet(std::integral_constant<int, 4>{}); facet(std::integral_constant<int, 5>{});
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const double Dk = C.D[s][k], zD = C.z[k] * Dk, zpD = zD * C.psi;
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = M[j] * C.inv_dt + Dk * S[j] + zpD * Dr[j] + zD * (C.z[k] > 0.0 ? UP[j] : UM[j]);
    if (valid) hx_put_own(scr, v);
    hx_flush(scrs, dsts, tid, valid, rp + slot_self * HX_NV, D.A_knp + (size_t)k * D.nnz);
    if (valid) D.b_knp[(size_t)k * D.n_dof + row] = rhs[k];
  }
}

int hx_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    kn_set_error(std::string(what) + ": " + hipGetErrorString(e));
    return KNPEMI_EHIP;
  }
  return KNPEMI_OK;
}

}  // namespace

int kn_dg_hex_launch_emi(hipStream_t st, const kn_dg::DgDev& D, const kn_dg::DgConsts* d_consts, int splitting, int box) {
  const int nblocks = (D.n_cell + HX_CELLS - 1) / HX_CELLS, chunk = (nblocks + 7) / 8;
  if (box) hipLaunchKernelGGL(dg_emi_hex_box_kernel, dim3(8 * chunk), dim3(DG_BLOCK), 0, st, D, d_consts, chunk, splitting);
  else hipLaunchKernelGGL(dg_emi_hex_kernel, dim3(8 * chunk), dim3(DG_BLOCK), 0, st, D, d_consts, chunk, splitting);
#ifdef KN_HX_STAMPS
  static int launches = 0;
  if (box && ++launches % 8 == 0) {
    static std::vector<unsigned long long> acc(1024 * 32);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(acc.data(), HIP_SYMBOL(hx_stamp_acc), acc.size() * sizeof(unsigned long long));
    const double waves = 8.0 * nblocks;
    double ph[18] = {0}, total = 0.0;
    for (int sl = 0; sl < 1024; ++sl) for (int i = 0; i < 18; ++i) ph[i] += (double)acc[sl * 32 + i];
    for (int i = 0; i < 18; ++i) total += ph[i];
    fprintf(stderr, "[hx stamps] dg_emi_hex_box_kernel, s_memtime ticks (100 MHz) per wave, 8 launches of %d waves: total %.1f\n", nblocks, total / waves);
    for (int i = 0; i < 18; ++i) fprintf(stderr, "[hx stamps]   phase %2d: %8.1f\n", i, ph[i] / waves);
    std::fill(acc.begin(), acc.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(hx_stamp_acc), acc.data(), acc.size() * sizeof(unsigned long long));
  }
#endif
  return hx_check("dg_emi_hex_kernel");
}

int kn_dg_hex_launch_knp(hipStream_t st, const kn_dg::DgDev& D, const kn_dg::DgConsts* d_consts, int KS, int splitting, int box) {
  const int nblocks = (D.n_cell + HX_CELLS - 1) / HX_CELLS, chunk = (nblocks + 7) / 8;
  const dim3 grid(8 * chunk), block(DG_BLOCK);
  if (box) {
    switch (KS) {
      case 1: hipLaunchKernelGGL(dg_knp_hex_box_kernel<1>, grid, block, 0, st, D, d_consts, chunk, splitting); break;
      case 2: hipLaunchKernelGGL(dg_knp_hex_box_kernel<2>, grid, block, 0, st, D, d_consts, chunk, splitting); break;
      default: hipLaunchKernelGGL(dg_knp_hex_box_kernel<3>, grid, block, 0, st, D, d_consts, chunk, splitting); break;
    }
    return hx_check("dg_knp_hex_box_kernel");
  }
  switch (KS) {
    case 1: hipLaunchKernelGGL(dg_knp_hex_kernel<1>, grid, block, 0, st, D, d_consts, chunk, splitting); break;
    case 2: hipLaunchKernelGGL(dg_knp_hex_kernel<2>, grid, block, 0, st, D, d_consts, chunk, splitting); break;
    default: hipLaunchKernelGGL(dg_knp_hex_kernel<3>, grid, block, 0, st, D, d_consts, chunk, splitting); break;
  }
  return hx_check("dg_knp_hex_kernel");
}
