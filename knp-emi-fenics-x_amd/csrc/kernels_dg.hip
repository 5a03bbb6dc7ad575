// DG(P1) + symmetric interior penalty variant of the per-step assembly on gfx950.
//
// No reference file is restated here: /root/reference is continuous Galerkin on sub-meshes; only its README
// (README.md:5-7) and the "interior facets tagged 0" convention of make_mesh_2D.py:88-90 point at the DG method.  The
// discrete problem is the one oracle/knpemi_dg_oracle.py spells out (volume terms of emiWeakForm.py:138-241 and
// knpWeakForm.py:123-166 cell by cell, SIP + upwind terms on interior facets, the reference's membrane terms on tagged
// facets) and these kernels are held to that restatement at 1e-10.
//
// Layout in HBM
//   rec      [n_dof][8]   one 64-byte record per broken dof (cell c, local vertex j) = c * nv + j:
//                         x y z c3 | c0 c1 c2 phi -- the vertex record of the CG path (KN_REC, KN_CSLOT), so the membrane
//                         ODE sweep (ode_kernel.h) reads its concentration traces from it unchanged
//   nbr      [n_cell][nv] cell across local facet f (-1: outer boundary)
//   finfo    [n_cell][nv] packed: kind (0 boundary, 1 interior, 2 membrane seen from the ECS cell, 3 from the other
//                         side) | the neighbour's local vertex opposite the facet | for every local vertex a of this
//                         cell the neighbour's local index of the same vertex | its node index on the membrane facet
//   A_*      CSR values: every row holds one nv-wide block per cell (itself and its facet neighbours, sorted by cell),
//            so the rows of one cell are contiguous and the rows of consecutive cells follow each other
//
// One lane per row.  All lanes of a cell compute the cell geometry redundantly (a few dozen flops) rather than share it:
// the kernels write 20 (tetrahedra) or 12 (triangles) doubles per row and ion and read about a quarter of that, so
// they are bound by the HBM write stream.  To make that stream coalesced the workgroup builds the image of its rows in
// LDS exactly as it sits in the CSR value array and then copies it out linearly (each wave instruction stores 1 KiB
// of consecutive addresses), instead of 64 lanes storing to 64 different rows.
//
// The lane rotates the cell's local numbering so that its own row vertex is local vertex 0 and reads a neighbour's dofs
// in the order of the vertices they share, the neighbour's far vertex taking the place of the vertex opposite the
// facet: every index into a register array is then a compile-time constant and only memory addresses are computed.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

#include "dg_internal.h"
#include "ode_kernel.h"

namespace {

using namespace kn_dg;

template <int NV>
struct Geo {
  double g[NV][NV - 1];   // gradients of the barycentric coordinates
  double vol;
};

template <int GD>
__device__ __forceinline__ double dot(const double (&a)[GD], const double (&b)[GD]) {
  double s = a[0] * b[0];
#pragma unroll
  for (int d = 1; d < GD; ++d) s += a[d] * b[d];
  return s;
}

template <int NV>
__device__ __forceinline__ void geometry(const double (&X)[NV][NV - 1], Geo<NV>& G) {
  if constexpr (NV == 3) {
    const double e1x = X[1][0] - X[0][0], e1y = X[1][1] - X[0][1];
    const double e2x = X[2][0] - X[0][0], e2y = X[2][1] - X[0][1];
    const double det = e1x * e2y - e1y * e2x, inv = fast_rcp(det);
    G.g[1][0] = e2y * inv; G.g[1][1] = -e2x * inv;
    G.g[2][0] = -e1y * inv; G.g[2][1] = e1x * inv;
    G.g[0][0] = -(G.g[1][0] + G.g[2][0]); G.g[0][1] = -(G.g[1][1] + G.g[2][1]);
    G.vol = 0.5 * fabs(det);
  } else {
    double e[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int d = 0; d < 3; ++d) e[a][d] = X[a + 1][d] - X[0][d];
    double cr[3][3];   // cr[a] = e[a+1] x e[a+2]
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int b = (a + 1) % 3, c = (a + 2) % 3;
      cr[a][0] = e[b][1] * e[c][2] - e[b][2] * e[c][1];
      cr[a][1] = e[b][2] * e[c][0] - e[b][0] * e[c][2];
      cr[a][2] = e[b][0] * e[c][1] - e[b][1] * e[c][0];
    }
    const double det = e[0][0] * cr[0][0] + e[0][1] * cr[0][1] + e[0][2] * cr[0][2], inv = fast_rcp(det);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      G.g[1][d] = cr[0][d] * inv; G.g[2][d] = cr[1][d] * inv; G.g[3][d] = cr[2][d] * inv;
      G.g[0][d] = -(G.g[1][d] + G.g[2][d] + G.g[3][d]);
    }
    G.vol = fabs(det) * (1.0 / 6.0);
  }
}

// position of every block of the row: the cell itself and the neighbours in increasing cell order
template <int NV>
__device__ __forceinline__ void block_slots(int T, const int (&nb)[NV], int& slot_self, int (&slot)[NV]) {
  slot_self = 0;
#pragma unroll
  for (int f = 0; f < NV; ++f) {
    slot_self += nb[f] >= 0 && nb[f] < T;
    int s = T < nb[f];
#pragma unroll
    for (int g = 0; g < NV; ++g) s += nb[g] >= 0 && nb[g] < nb[f];
    slot[f] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Both kernels work in two phases.  Phase 1: lane (cell, i) prepares facet i of its cell -- the facet opposite its
// own vertex -- i.e. the neighbour's geometry and traces, and leaves what the rows need of it in LDS (FS doubles per
// facet, indexed by the cell's ORIGINAL local numbering).  Phase 2: every lane builds its row from the four facet
// summaries of its cell.  Each neighbour geometry is thus computed once per cell instead of once per row.
// ---------------------------------------------------------------------------------------------------------------------
// The LDS row image holds the block's rows in CSR order, so that it can be copied out linearly.  Rows are 20 (12)
// doubles apart: lanes 8 rows apart would hit the same banks (8 x 40 dwords = 5 x 64), an 8-way conflict on every one of
// the 20 stores of a row.  Two doubles of padding after every 8 interior rows' worth of entries (DG_PADQ) shift those
// lanes onto different banks and keep the 16-byte alignment of the copy-out.
template <int NV> constexpr int dg_padq() { return 8 * NV * (NV + 1); }
template <int NV> constexpr int dg_img_doubles() { return DG_ROUND * NV * (NV + 1) + 2 * (DG_ROUND / 8 + 1); }
// LDS scratch behind the facet summaries: first the staged records of the whole workgroup, later NIMG row images
template <int NV, int NIMG> constexpr int dg_scratch_doubles() {
  constexpr int a = NIMG * dg_img_doubles<NV>(), b = (DG_BLOCK / NV) * NV * DG_RPITCH;
  return a > b ? a : b;
}

template <int NV>
struct ImgRow {
  double* base;    // image address of the row's first entry
  int to_pad;      // entries of this row before the next padding point
  __device__ __forceinline__ ImgRow(double* img, int off) {
    const int q = off / dg_padq<NV>();
    base = img + off + 2 * q;
    to_pad = (q + 1) * dg_padq<NV>() - off;
  }
  __device__ __forceinline__ void put(int k, double v) const { base[k + (k >= to_pad ? 2 : 0)] = v; }
};

// LDS row image -> CSR value array.  Rows of tetrahedra are multiples of 4 doubles long, so the block's span starts
// 32-byte aligned: two doubles per lane and store (1 KiB of consecutive addresses per wave instruction).
template <int NV>
__device__ __forceinline__ void copy_out(double* __restrict__ dst, const double* img, int span, int tid) {
  constexpr int PQ = dg_padq<NV>();
  // (plain stores: the whole-line streaming stores that help the hexahedral kernels -- kernels_dg_hex.hip, hx_flush -- cost these
  // contiguous spans 6-8 % at 995 k tetrahedra and gain 3-4 % at 124 k; round 4, A/B on one box)
  if constexpr (NV == 4) {
    double2* d2 = reinterpret_cast<double2*>(dst);
    for (int e = tid; e < span / 2; e += DG_BLOCK)
      d2[e] = *reinterpret_cast<const double2*>(img + 2 * e + 2 * ((2 * e) / PQ));
  } else {
    for (int e = tid; e < span; e += DG_BLOCK) dst[e] = img[e + 2 * (e / PQ)];
  }
}

template <int NV> constexpr int dg_fs_emi() { return (6 + 2 * NV) | 1; }   // n[3] area inv_h JNn | gNn[nv] | kN[nv]
// n[3] area inv_h gphiNn | interior facet: gNn[nv]; membrane facet: G[facet vertex][k], the membrane integrals
template <int NV, int KS> constexpr int dg_fs_knp() { return (6 + ((NV - 1) * KS > NV ? (NV - 1) * KS : NV)) | 1; }

// Diagnostic build (make CXXFLAGS+=-DKN_DG_STAMPS): dg_emi_kernel adds up s_memtime differences between its phases (one
// atomic per phase and wave, spread over 1024 slots); the launcher prints the averages every 8 launches.
#ifdef KN_DG_STAMPS
__device__ unsigned long long dg_stamp_acc[1024 * 16];
#define DG_STAMP_BEGIN unsigned long long dg_t_ = __builtin_amdgcn_s_memtime();
#define DG_STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                         if (threadIdx.x == 0) atomicAdd(&dg_stamp_acc[(blockIdx.x & 1023) * 16 + (i)], n_ - dg_t_); dg_t_ = n_; } while (0)
#else
#define DG_STAMP_BEGIN
#define DG_STAMP(i) do {} while (0)
#endif

template <int NV>
__global__ __launch_bounds__(DG_BLOCK) void dg_emi_kernel(DgDev D, const DgConsts* __restrict__ Cp, int chunk, int splitting) {
  constexpr int GD = NV - 1, NF = NV - 1;
  constexpr int RPB = (DG_BLOCK / NV) * NV;
  constexpr int FS = dg_fs_emi<NV>();
  extern __shared__ double lds[];
  double* fs = lds;
  double* img = lds + RPB * FS;
  const DgConsts& C = *Cp;
  const int row0 = dg_block_index(blockIdx.x, chunk) * RPB;
  if (row0 >= D.n_dof) return;
  const int nrows = min(RPB, D.n_dof - row0);
  const int tid = threadIdx.x;
  DG_STAMP_BEGIN
  const bool valid = tid < nrows;
  const int base = D.rowptr[row0];
  // (the end of the workgroup's span of the value array: needed only when the rows leave, requested with everything else --
  // loaded there it was one more round trip to memory per wave, an eighth of its life; s_memtime stamps, round 4)
  const int span_end = D.rowptr[row0 + nrows];
  const int row = row0 + tid, T = row / NV, i = row - T * NV;
  int p[NV], nb[NV];
  unsigned fi[NV];
  double X[NV][GD], kap[NV], J[GD];
  Geo<NV> G;
  int s = 0;
#pragma unroll
  for (int j = 0; j < NV; ++j) p[j] = i + j >= NV ? i + j - NV : i + j;
  // everything that does not depend on other loads is requested now, before the first barrier: one round trip to
  // memory instead of three (the waves spend most of their life waiting for loads, not computing)
  int rowoff = 0;
  if (valid) {
    stage_rec(img, D.rec, row, tid);
    s = D.cell_sub[T];
    rowoff = D.rowptr[row] - base;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      nb[j] = D.nbr[T * NV + p[j]];
      fi[j] = D.finfo[T * NV + p[j]];
    }
  }
  __syncthreads();
  DG_STAMP(0);      // own records and topology
  if (valid) {
    double sg[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const DofRec r = lds_rec(img, tid - i + p[j]);
#pragma unroll
      for (int d = 0; d < GD; ++d) X[j][d] = r.x[d];
      double k = 0.0, g = 0.0;
#pragma unroll
      for (int q = 0; q < KN_MAXK; ++q) { k += C.kap[s][q] * r.c[q]; g += C.sig[s][q] * r.c[q]; }
      kap[j] = k; sg[j] = g;
    }
    geometry<NV>(X, G);
#pragma unroll
    for (int d = 0; d < GD; ++d) {   // sum_k F z_k D_k grad c_k
      double a = 0.0;
#pragma unroll
      for (int j = 0; j < NV; ++j) a += sg[j] * G.g[j][d];
      J[d] = a;
    }
    // phase 1: the facet opposite my vertex
    double* my = fs + tid * FS;
    const double g2 = dot<GD>(G.g[0], G.g[0]), rgf = fast_rsqrt(g2), gf = g2 * rgf;
    double n[GD];
#pragma unroll
    for (int d = 0; d < GD; ++d) { n[d] = -G.g[0][d] * rgf; my[d] = n[d]; }
    my[3] = GD * G.vol * gf;
    if (nb[0] >= 0 && (fi[0] & 3) == 1) {
      const int N = nb[0];
      // The neighbour's gradients are needed along n only, and those follow from this cell's own gradients and the
      // neighbour's far vertex x_far: with h_N = (x_far - x_1) . n its height over the facet,
      //   grad lambda^N_far . n = 1 / h_N,     grad lambda^N_b . n = grad lambda_b . n - lambda_b(x_far) / h_N  (b on the facet).
      double kN[NV], sN[NV], dfar[GD];
#pragma unroll
      for (int b = 0; b < NV; ++b) {
        const int jn = b == 0 ? (fi[0] >> 2) & 3 : (fi[0] >> (4 + 2 * p[b])) & 3;
        // a neighbour inside this workgroup's cells is already staged in LDS (half of them are, for Kuhn-split meshes:
        // the tetrahedra of one hexahedron follow each other): only the others are gathered from memory
        const int ln = N * NV + jn - row0;
        const double* rp = (ln >= 0 && ln < nrows) ? img + ln * DG_RPITCH : D.rec + (size_t)(N * NV + jn) * KN_REC;
        const DofRec r = load_rec(rp, 0);      // one address, LDS or global: generic loads
        if (b == 0) {
#pragma unroll
          for (int d = 0; d < GD; ++d) dfar[d] = r.x[d] - X[1][d];
        }
        double k = 0.0, g = 0.0;
#pragma unroll
        for (int q = 0; q < KN_MAXK; ++q) { k += C.kap[s][q] * r.c[q]; g += C.sig[s][q] * r.c[q]; }
        kN[b] = k; sN[b] = g;
      }
      const double rh = fast_rcp(dot<GD>(dfar, n));
      my[4] = 0.5 * (gf + rh);
      double JNn = 0.0;
#pragma unroll
      for (int b = 0; b < NV; ++b) {
        const double gn = b == 0 ? rh : dot<GD>(G.g[b], n) - ((b == 1 ? 1.0 : 0.0) + dot<GD>(G.g[b], dfar)) * rh;
        JNn += sN[b] * gn;
        my[6 + p[b]] = gn;
        my[6 + NV + p[b]] = kN[b];
      }
      my[5] = JNn;
    } else if (nb[0] >= 0) {
      // membrane facet: the Robin datum g = phi_M (- I_ch / C_phi without the splitting scheme) at the facet's nodes, read
      // once by this lane (two dependent loads) for the three rows that need it
      const int mf = D.mfid[T * NV + p[0]];
#pragma unroll
      for (int a = 1; a < NV; ++a) {
        const int q = mf * NF + ((fi[0] >> (12 + 2 * p[a])) & 3);
        double g = D.phiM[q];
        if (!splitting) {
          double it = 0.0;
          for (int k = 0; k < C.K; ++k) it += D.Ich[(size_t)k * D.nq + q];
          g -= it / C.C_phi;
        }
        my[6 + (p[a] < i ? p[a] : p[a] - 1)] = g;
      }
    }
  }
  __syncthreads();
  DG_STAMP(1);      // geometry, the facet opposite my vertex (neighbour gather)
  double self[NV], nbv[NV][NV];   // nbv[f][b]: column of the neighbour's dof at my vertex b; b == f: its far vertex
  if (valid) {
    double kbar = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) kbar += kap[j];
    kbar *= 1.0 / NV;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      self[j] = G.vol * kbar * dot<GD>(G.g[0], G.g[j]);
#pragma unroll
      for (int b = 0; b < NV; ++b) nbv[j][b] = 0.0;
    }
    double rhs = -G.vol * dot<GD>(G.g[0], J);
#pragma unroll
    for (int f = 0; f < NV; ++f) {
      if (nb[f] < 0) continue;
      const bool on0 = f != 0;   // the row's basis function does not vanish on this facet
      const int kind = fi[f] & 3;
      const double* fd = fs + (tid - i + p[f]) * FS;
      double n[GD];
#pragma unroll
      for (int d = 0; d < GD; ++d) n[d] = fd[d];
      const double area = fd[3];
      const double c2 = area * (1.0 / (GD * (GD + 1)));
      if (kind == 1) {
        const double inv_h = fd[4], JNn = fd[5];
        double gNn[NV], kN[NV];
#pragma unroll
        for (int b = 0; b < NV; ++b) { gNn[b] = fd[6 + p[b]]; kN[b] = fd[6 + NV + p[b]]; }
        double ST = 0.0, SN = 0.0, St = 0.0, kt[NV];
#pragma unroll
        for (int b = 0; b < NV; ++b) {
          kt[b] = 0.5 * (kap[b] + kN[b]);
          if (b != f) { ST += kap[b]; SN += kN[b]; St += kt[b]; }
        }
        const double c3 = area * (GD == 2 ? 1.0 / 24.0 : 1.0 / 60.0);
        const double gin = dot<GD>(G.g[0], n);
        const double pen = C.gamma * inv_h;
        const double K1T0 = c2 * (ST + kap[0]), K1N0 = c2 * (SN + kN[0]);   // int kappa lambda_0 over the facet (on0 only)
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          if (on0) {   // consistency: -1/2 (grad lambda_j . n) int kappa lambda_0, both sides, every j
            self[j] -= 0.5 * dot<GD>(G.g[j], n) * K1T0;
            nbv[f][j] -= 0.5 * gNn[j] * K1N0;
          }
          if (j != f) {   // symmetry and penalty: columns of the dofs on the facet
            const double K1Tj = c2 * (ST + kap[j]);
            self[j] -= 0.5 * gin * K1Tj;
            nbv[f][j] += 0.5 * gin * K1Tj;
            if (on0) {
              const double P3 = j == 0 ? c3 * (2.0 * St + 4.0 * kt[0]) : c3 * (St + kt[0] + kt[j]);
              self[j] += pen * P3;
              nbv[f][j] -= pen * P3;
            }
          }
        }
        if (on0) rhs += 0.5 * (dot<GD>(J, n) + JNn) * (area * (1.0 / GD));
      } else if (on0) {   // membrane: C_phi [u][v] and C_phi g [v]  (emiWeakForm.py:160-165, 228-239)
        const int fo = p[f];           // original local vertex opposite the facet: g sits by facet-local position
        double gsum = 0.0;
#pragma unroll
        for (int a = 0; a < NV; ++a) {
          if (a == f) continue;
          const double m2 = C.C_phi * c2 * (a == 0 ? 2.0 : 1.0);
          self[a] += m2;
          nbv[f][a] -= m2;
          gsum += fd[6 + (p[a] < fo ? p[a] : p[a] - 1)] * m2;
        }
        rhs += kind == 3 ? gsum : -gsum;
      }
    }
    D.b_emi[row] = rhs;
  }
  DG_STAMP(2);      // rows
  // The rows go out through an LDS image of DG_ROUND rows at a time, laid out as they sit in the CSR value array: the
  // wave that owns them fills it, the whole workgroup copies it out linearly.  (One image for all rows of the
  // workgroup would cost 2.5 x the LDS and a third of the resident waves.)
  int slot_self = 0, slot[NV];
  if (valid) block_slots<NV>(T, nb, slot_self, slot);
  for (int r0 = 0; r0 < nrows; r0 += DG_ROUND) {
    const int r1 = min(r0 + DG_ROUND, nrows);
    const int wbase = r0 == 0 ? 0 : D.rowptr[row0 + r0] - base, wspan = (r1 == nrows ? span_end : D.rowptr[row0 + r1]) - base - wbase;
    if (tid >= r0 && tid < r1) {
      const ImgRow<NV> out(img, rowoff - wbase);
#pragma unroll
      for (int j = 0; j < NV; ++j) out.put(slot_self * NV + p[j], self[j]);
#pragma unroll
      for (int f = 0; f < NV; ++f) {
        if (nb[f] < 0) continue;
#pragma unroll
        for (int b = 0; b < NV; ++b) {
          const int jn = b == f ? (fi[f] >> 2) & 3 : (fi[f] >> (4 + 2 * p[b])) & 3;
          out.put(slot[f] * NV + jn, nbv[f][b]);
        }
      }
    }
    __syncthreads();
    DG_STAMP(3);    // row image filled
    copy_out<NV>(D.A_emi + base + wbase, img, wspan, tid);
    if (r1 < nrows) __syncthreads();
    DG_STAMP(4);    // copied out
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// concentration systems: A_knp[k], b_knp[k], k < K - 1.  The unit-diffusivity SIP entries are the same for every ion
// (D_k is constant inside a sub-domain); the K - 1 matrices go through one LDS row image, one after the other.
// ---------------------------------------------------------------------------------------------------------------------
template <int NV, int KS>
__global__ __launch_bounds__(DG_BLOCK) void dg_knp_kernel(DgDev D, const DgConsts* __restrict__ Cp, int chunk, int splitting) {
  constexpr int GD = NV - 1, NF = NV - 1;
  constexpr int RPB = (DG_BLOCK / NV) * NV;
  constexpr int FS = dg_fs_knp<NV, KS>();
  extern __shared__ double lds[];
  double* fs = lds;
  double* img = lds + RPB * FS;
  const DgConsts& C = *Cp;
  const int row0 = dg_block_index(blockIdx.x, chunk) * RPB;
  if (row0 >= D.n_dof) return;
  const int nrows = min(RPB, D.n_dof - row0);
  const int tid = threadIdx.x;
  const bool valid = tid < nrows;
  const int base = D.rowptr[row0];
  // (the end of the workgroup's span of the value array: needed only when the rows leave, requested with everything else --
  // loaded there it was one more round trip to memory per wave, an eighth of its life; s_memtime stamps, round 4)
  const int span_end = D.rowptr[row0 + nrows];
  const int row = row0 + tid, T = row / NV, i = row - T * NV;
  int p[NV], nb[NV];
  unsigned fi[NV];
  double X[NV][GD], ph[NV], cc[NV][KN_MAXK], gphi[GD];
  Geo<NV> G;
  int s = 0;
#pragma unroll
  for (int j = 0; j < NV; ++j) p[j] = i + j >= NV ? i + j - NV : i + j;
  // everything that does not depend on other loads is requested now, before the first barrier: one round trip to
  // memory instead of three (the waves spend most of their life waiting for loads, not computing)
  int rowoff = 0;
  if (valid) {
    stage_rec(img, D.rec, row, tid);
    s = D.cell_sub[T];
    rowoff = D.rowptr[row] - base;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      nb[j] = D.nbr[T * NV + p[j]];
      fi[j] = D.finfo[T * NV + p[j]];
    }
  }
  __syncthreads();
  if (valid) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const DofRec r = lds_rec(img, tid - i + p[j]);
#pragma unroll
      for (int d = 0; d < GD; ++d) X[j][d] = r.x[d];
#pragma unroll
      for (int q = 0; q < KN_MAXK; ++q) cc[j][q] = r.c[q];
      ph[j] = r.phi;
    }
    geometry<NV>(X, G);
#pragma unroll
    for (int d = 0; d < GD; ++d) {
      double a = 0.0;
#pragma unroll
      for (int j = 0; j < NV; ++j) a += ph[j] * G.g[j][d];
      gphi[d] = a;
    }
    double* my = fs + tid * FS;
    const double g2 = dot<GD>(G.g[0], G.g[0]), rgf = fast_rsqrt(g2), gf = g2 * rgf;
    double n[GD];
#pragma unroll
    for (int d = 0; d < GD; ++d) { n[d] = -G.g[0][d] * rgf; my[d] = n[d]; }
    my[3] = GD * G.vol * gf;
    if (nb[0] >= 0 && (fi[0] & 3) == 1) {
      const int N = nb[0];
      double phN[NV], dfar[GD];   // normal components of the neighbour's gradients: see dg_emi_kernel
#pragma unroll
      for (int b = 0; b < NV; ++b) {
        const int jn = b == 0 ? (fi[0] >> 2) & 3 : (fi[0] >> (4 + 2 * p[b])) & 3;
        const int ln = N * NV + jn - row0;      // staged in LDS if the neighbour belongs to this workgroup
        const bool local = ln >= 0 && ln < nrows;
        const double* r = local ? img + ln * DG_RPITCH : D.rec + (size_t)(N * NV + jn) * KN_REC;
        phN[b] = r[7];
        if (b == 0) {
#pragma unroll
          for (int d = 0; d < GD; ++d) dfar[d] = r[d] - X[1][d];
        }
      }
      const double rh = fast_rcp(dot<GD>(dfar, n));
      my[4] = 0.5 * (gf + rh);
      double gphiNn = 0.0;
#pragma unroll
      for (int b = 0; b < NV; ++b) {
        const double gn = b == 0 ? rh : dot<GD>(G.g[b], n) - ((b == 1 ? 1.0 : 0.0) + dot<GD>(G.g[b], dfar)) * rh;
        gphiNn += phN[b] * gn;
        my[6 + p[b]] = gn;
      }
      my[5] = gphiNn;
    } else if (nb[0] >= 0) {
      // membrane facet (knpWeakForm.py:168-214), degree-6 rule, evaluated ONCE per (cell, facet) by this lane for the
      // three rows that see the facet.  With alpha_k = D_k z_k^2 c_k / sum_j D_j z_j^2 c_j of this side the integrand
      //   -/+ (C_k g_k - C_k [phi])  is
      //   sgn * [ alpha_k C_M / (F z_k dt) ([phi] - phi_M - (dt / C_M) I_ch) + I_ch_k / (F z_k) ],  sgn = +1 on the ECS side
      const int N = nb[0], kind = fi[0] & 3;
      const int mf = D.mfid[T * NV + p[0]];
      const double area = GD * G.vol * gf;
      double jm[NV], pm[NV], It[NV], Ik[NV][KN_MAXK], Gk[NV][KS];
#pragma unroll
      for (int a = 1; a < NV; ++a) {
        const int jn = (fi[0] >> (4 + 2 * p[a])) & 3;
        const int ln = N * NV + jn - row0;
        const double phn = (ln >= 0 && ln < nrows) ? img[ln * DG_RPITCH + 7] : D.rec[(size_t)(N * NV + jn) * KN_REC + 7];
        jm[a] = kind == 2 ? phn - ph[a] : ph[a] - phn;
        const int q = mf * NF + ((fi[0] >> (12 + 2 * p[a])) & 3);
        pm[a] = D.phiM[q];
        double it = 0.0;
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) {
          Ik[a][k] = k < C.K ? D.Ich[(size_t)k * D.nq + q] : 0.0;
          it += Ik[a][k];
        }
        It[a] = it;
#pragma unroll
        for (int k = 0; k < KS; ++k) Gk[a][k] = 0.0;
      }
      const double sgn = kind == 2 ? 1.0 : -1.0;
      const double* qw = D.qtab;
      const double* qN = D.qtab + D.nquad;
      for (int q = 0; q < D.nquad; ++q) {
        double cq[KN_MAXK], iq[KN_MAXK], jq = 0.0, pq = 0.0, itq = 0.0, Nq[NV];
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) { cq[k] = 0.0; iq[k] = 0.0; }
#pragma unroll
        for (int a = 1; a < NV; ++a) {
          Nq[a] = qN[q * NF + (a - 1)];
#pragma unroll
          for (int k = 0; k < KN_MAXK; ++k) { cq[k] += Nq[a] * cc[a][k]; iq[k] += Nq[a] * Ik[a][k]; }
          jq += Nq[a] * jm[a]; pq += Nq[a] * pm[a]; itq += Nq[a] * It[a];
        }
        double asum = 0.0;
#pragma unroll
        for (int k = 0; k < KN_MAXK; ++k) asum += C.az2D[s][k] * cq[k];
        const double w = sgn * qw[q] * area * (NF == 2 ? 1.0 : 2.0);
        double drive = jq - pq;
        if (splitting) drive -= (C.dt / C.C_M) * itq;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const double al = C.az2D[s][k] * cq[k] / asum;
          const double fz = 1.0 / (C.F * C.z[k]);
          const double fq = w * (al * C.C_M * fz * C.inv_dt * drive + iq[k] * fz);
#pragma unroll
          for (int a = 1; a < NV; ++a) Gk[a][k] += Nq[a] * fq;
        }
      }
      // by the vertex's position among the facet's vertices in the cell's ORIGINAL numbering (the facet is opposite i)
#pragma unroll
      for (int a = 1; a < NV; ++a) {
        const int loc = p[a] < i ? p[a] : p[a] - 1;
#pragma unroll
        for (int k = 0; k < KS; ++k) my[6 + loc * KS + k] = Gk[a][k];
      }
    }
  }
  __syncthreads();
  double P[NV], Pn[NV][NV], bf[NV], c2f[NV], rhs[KS];
  double drift = 0.0, m0 = 0.0;
  int slot_self = 0, slot[NV], off = 0;
  if (valid) {
    drift = dot<GD>(G.g[0], gphi) * G.vol * (1.0 / (GD + 1));
    m0 = G.vol * (1.0 / ((GD + 1) * (GD + 2)));
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      double a = 0.0;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        double v = cc[j][k] * C.inv_dt;
        if (s == 0 && D.fsrc) v += D.fsrc[(size_t)k * D.n_dof + T * NV + p[j]];
        a += (j == 0 ? 2.0 : 1.0) * m0 * v;
      }
      rhs[k] = a;
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      P[j] = G.vol * dot<GD>(G.g[0], G.g[j]);
      bf[j] = 0.0; c2f[j] = 0.0;
#pragma unroll
      for (int b = 0; b < NV; ++b) Pn[j][b] = 0.0;
    }
#pragma unroll
    for (int f = 0; f < NV; ++f) {
      if (nb[f] < 0) continue;
      const bool on0 = f != 0;
      const int kind = fi[f] & 3;
      const double* fd = fs + (tid - i + p[f]) * FS;
      double n[GD];
#pragma unroll
      for (int d = 0; d < GD; ++d) n[d] = fd[d];
      const double area = fd[3];
      const double c2 = area * (1.0 / (GD * (GD + 1)));
      if (kind == 1) {
        const double inv_h = fd[4], gphiNn = fd[5];
        const double m1 = area * (1.0 / GD);
        const double gin = dot<GD>(G.g[0], n);
        const double pen = C.gamma * inv_h;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          if (on0) {
            P[j] -= 0.5 * dot<GD>(G.g[j], n) * m1;
            Pn[f][j] -= 0.5 * fd[6 + p[j]] * m1;
          }
          if (j != f) {
            P[j] -= 0.5 * gin * m1;
            Pn[f][j] += 0.5 * gin * m1;
            if (on0) {
              const double m2 = pen * c2 * (j == 0 ? 2.0 : 1.0);
              P[j] += m2;
              Pn[f][j] -= m2;
            }
          }
        }
        bf[f] = -0.5 * (dot<GD>(gphi, n) + gphiNn);   // beta_k = z_k psi D_k bf: drift speed out of this cell
        c2f[f] = on0 ? c2 : 0.0;
      } else if (on0) {   // membrane facet: the integrals of this row's function, prepared by the facet's lane
        const int fo = p[f];                          // original local vertex opposite the facet
        const int loc = i < fo ? i : i - 1;
#pragma unroll
        for (int k = 0; k < KS; ++k) rhs[k] += fd[6 + loc * KS + k];
      }
    }
    block_slots<NV>(T, nb, slot_self, slot);
    off = rowoff;
#pragma unroll
    for (int k = 0; k < KS; ++k) D.b_knp[(size_t)k * D.n_dof + row] = rhs[k];
  }
  // the K - 1 matrices go through the one LDS image in turn (DG_ROUND rows at a time, as in dg_emi_kernel)
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const double Dk = C.D[s][k], zpD = C.z[k] * C.psi * Dk;
    for (int r0 = 0; r0 < nrows; r0 += DG_ROUND) {
      const int r1 = min(r0 + DG_ROUND, nrows);
      const int wbase = r0 == 0 ? 0 : D.rowptr[row0 + r0] - base, wspan = (r1 == nrows ? span_end : D.rowptr[row0 + r1]) - base - wbase;
      if (k > 0 || r0 > 0) __syncthreads();   // the previous image has been copied out
      if (tid >= r0 && tid < r1) {
        const ImgRow<NV> out(img, off - wbase);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          double v = (j == 0 ? 2.0 : 1.0) * m0 * C.inv_dt + Dk * P[j] + zpD * drift;
#pragma unroll
          for (int f = 0; f < NV; ++f) {   // upwinding: the drift leaves through facet f -> this cell's own value
            const double beta = zpD * bf[f];
            if (j != f && beta > 0.0) v += beta * c2f[f] * (j == 0 ? 2.0 : 1.0);
          }
          out.put(slot_self * NV + p[j], v);
        }
#pragma unroll
        for (int f = 0; f < NV; ++f) {
          if (nb[f] < 0) continue;
          const double beta = zpD * bf[f];
#pragma unroll
          for (int b = 0; b < NV; ++b) {
            const int jn = b == f ? (fi[f] >> 2) & 3 : (fi[f] >> (4 + 2 * p[b])) & 3;
            double v = Dk * Pn[f][b];
            if (b != f && !(beta > 0.0)) v += beta * c2f[f] * (b == 0 ? 2.0 : 1.0);   // ... enters: the neighbour's value
            out.put(slot[f] * NV + jn, v);
          }
        }
      }
      __syncthreads();
      copy_out<NV>(D.A_knp + (size_t)k * D.nnz + base + wbase, img, wspan, tid);
    }
  }
}

// end of step (utils.py:238-295): c_prev <- c, eliminated ion from electroneutrality, phi_M <- phi_i - phi_e
__global__ void dg_update_kernel(DgDev D, const DgConsts* __restrict__ Cp, const double* __restrict__ cnew, int nv) {
  const DgConsts& C = *Cp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < D.n_dof) {
    const int s = D.cell_sub[t / nv];
    double* r = D.rec + (size_t)t * KN_REC;
    double acc = C.rho_term[s];
    for (int k = 0; k < C.K - 1; ++k) {
      const double v = cnew[(size_t)k * D.n_dof + t];
      r[KN_CSLOT(k)] = v;
      acc += C.elim[k] * v;
    }
    r[KN_CSLOT(C.K - 1)] = acc;
  }
  if (t < D.nq) D.phiM[t] = D.rec[(size_t)D.q2i[t] * KN_REC + 7] - D.rec[(size_t)D.q2e[t] * KN_REC + 7];
}

// ghost-cell halo: the five field slots (c3 c0 c1 c2 phi) of the listed dofs <-> a packed buffer
__global__ void dg_halo_kernel(double* rec, const int* __restrict__ idx, int n, double* buf, int unpack) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 5) return;
  const int e = t / 5, c = t - e * 5;
  double* r = rec + (size_t)idx[e] * KN_REC + 3 + c;
  if (unpack) *r = buf[t];
  else buf[t] = *r;
}

__global__ void dg_slot_kernel(double* rec, int slot, double* buf, int n, int to_records) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  if (to_records) rec[(size_t)t * KN_REC + slot] = buf[t];
  else buf[t] = rec[(size_t)t * KN_REC + slot];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
struct knpemi_dg {
  int device = 0, NV = 0, K = 0, n_sub = 0;
  int NFC = 0, NFV = 0;            // facets per cell, vertices per facet (hexahedra: 6 and 4)
  int hex_box = 0;                 // hexahedra: every cell is an orthogonal parallelepiped (box-mesh kernels)
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DgDev dev{};
  DgConsts consts{};
  DgConsts* d_consts = nullptr;
  int have_params = 0;
  std::vector<void*> allocs;
  std::vector<int> h_rowptr, h_colind, h_q2e, h_q2i;
  double* d_stage = nullptr;
  size_t stage_len = 0;
  double* d_fsrc = nullptr;
  const int* d_colind = nullptr;
  void* comm = nullptr;            // the library's RCCL communicator for the ghost-cell halo (comm_rccl.hip)
  int comm_world = 1;
  // per-launch event brackets of the two assembly kernels (knpemi_dg_profile)
  int prof_on = 0;                 // 0 off, n >= 1: every n-th launch is bracketed
  unsigned prof_count[2] = {0, 0};
  std::vector<hipEvent_t> prof_ev[2];
  size_t prof_used[2] = {0, 0};
  // membrane ODE sweep
  int ode_model = -1, ode_ns = 0, ode_np = 0, ode_v = 0, ode_blocks = 0;
  int ode_ion_param[3 * KN_MAXK] = {0};
  double* d_states = nullptr;
  double* d_params = nullptr;
  unsigned long long* d_stats = nullptr;
  void* d_coef = nullptr;
  // device solves (knpemi_dg_solve_emi / knpemi_dg_solve_knp): the Krylov + AMG code of the CG path run on the DG
  // systems through a handle that only carries what the solvers read
  std::vector<int> aux_of;         // continuous P1 dof (sub-domain, mesh vertex) of every broken dof
  int n_aux = 0;
  knpemi_handle* sol = nullptr;
  double* d_csol = nullptr;        // [K-1][n_dofs] solved concentrations
  int extrapolate = 0;             // knpemi_dg_set_extrapolation
};

namespace {

int dg_fail(int code, const std::string& msg) {
  kn_set_error(msg);
  return code;
}

template <class T>
int dg_alloc(knpemi_dg* h, size_t n, T** out) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  KN_HIP(hipMalloc(&p, bytes));
  h->allocs.push_back(p);
  // zero on the handle's own (non-blocking) stream: a null-stream hipMemset is not ordered with it
  KN_HIP(hipMemsetAsync(p, 0, bytes, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  *out = static_cast<T*>(p);
  return 0;
}

template <class T>
int dg_upload(knpemi_dg* h, const std::vector<T>& v, const T** out) {
  T* p = nullptr;
  int rc = dg_alloc(h, v.size(), &p);
  if (rc) return rc;
  if (!v.empty()) KN_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return 0;
}

int dg_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dg_fail(KNPEMI_EHIP, std::string(what) + ": " + hipGetErrorString(e));
  return KNPEMI_OK;
}

struct DgProf {   // brackets one launch with an event pair on the handle's stream
  knpemi_dg* h; int k; bool on;
  DgProf(knpemi_dg* h_, int k_) : h(h_), k(k_), on(h_->prof_on != 0) {
    if (on && h->prof_on > 1 && (h->prof_count[k]++ % (unsigned)h->prof_on) != 0) on = false;   // every n-th launch
    if (!on) return;
    auto& v = h->prof_ev[k];
    if (h->prof_used[k] + 2 > v.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
      v.push_back(a); v.push_back(b);
    }
    (void)hipEventRecord(v[h->prof_used[k]], h->stream);
  }
  ~DgProf() {
    if (!on) return;
    (void)hipEventRecord(h->prof_ev[k][h->prof_used[k] + 1], h->stream);
    h->prof_used[k] += 2;
  }
};

int launch_emi(knpemi_dg* h, int flags) {
  DgProf prof(h, 0);
  if (h->NV == 8) return kn_dg_hex_launch_emi(h->stream, h->dev, h->d_consts, !(flags & KNPEMI_NO_SPLITTING), h->hex_box);
  const int NV = h->NV, rpb = (DG_BLOCK / NV) * NV;
  const int nblocks = (h->dev.n_dof + rpb - 1) / rpb, chunk = (nblocks + 7) / 8;
  const size_t lds = ((size_t)rpb * (NV == 3 ? dg_fs_emi<3>() : dg_fs_emi<4>()) +
                      (NV == 3 ? dg_scratch_doubles<3, 1>() : dg_scratch_doubles<4, 1>())) * sizeof(double);
  const int split = !(flags & KNPEMI_NO_SPLITTING);
  if (NV == 3) hipLaunchKernelGGL(dg_emi_kernel<3>, dim3(8 * chunk), dim3(DG_BLOCK), lds, h->stream, h->dev, h->d_consts, chunk, split);
  else hipLaunchKernelGGL(dg_emi_kernel<4>, dim3(8 * chunk), dim3(DG_BLOCK), lds, h->stream, h->dev, h->d_consts, chunk, split);
#ifdef KN_DG_STAMPS
  static int launches = 0;
  if (++launches % 8 == 0) {
    static std::vector<unsigned long long> acc(1024 * 16);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(acc.data(), HIP_SYMBOL(dg_stamp_acc), acc.size() * sizeof(unsigned long long));
    const double waves = 8.0 * nblocks;
    double ph[5] = {0}, total = 0.0;
    for (int sl = 0; sl < 1024; ++sl) for (int i = 0; i < 5; ++i) ph[i] += (double)acc[sl * 16 + i];
    for (int i = 0; i < 5; ++i) total += ph[i];
    fprintf(stderr, "[dg stamps] dg_emi_kernel, cycles per wave, 8 launches of %d waves: total %.1f\n", nblocks, total / waves);
    for (int i = 0; i < 5; ++i) fprintf(stderr, "[dg stamps]   phase %d: %8.1f\n", i, ph[i] / waves);
    std::fill(acc.begin(), acc.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dg_stamp_acc), acc.data(), acc.size() * sizeof(unsigned long long));
  }
#endif
  return dg_check_launch("dg_emi_kernel");
}

template <int NV>
int launch_knp_nv(knpemi_dg* h, int chunk, int split) {
  constexpr int rpb = (DG_BLOCK / NV) * NV;
  const dim3 grid(8 * chunk), block(DG_BLOCK);
  switch (h->K - 1) {
    case 1: hipLaunchKernelGGL((dg_knp_kernel<NV, 1>), grid, block, ((size_t)rpb * dg_fs_knp<NV, 1>() + dg_scratch_doubles<NV, 1>()) * sizeof(double), h->stream, h->dev, h->d_consts, chunk, split); break;
    case 2: hipLaunchKernelGGL((dg_knp_kernel<NV, 2>), grid, block, ((size_t)rpb * dg_fs_knp<NV, 2>() + dg_scratch_doubles<NV, 1>()) * sizeof(double), h->stream, h->dev, h->d_consts, chunk, split); break;
    default: hipLaunchKernelGGL((dg_knp_kernel<NV, 3>), grid, block, ((size_t)rpb * dg_fs_knp<NV, 3>() + dg_scratch_doubles<NV, 1>()) * sizeof(double), h->stream, h->dev, h->d_consts, chunk, split); break;
  }
  return dg_check_launch("dg_knp_kernel");
}

int launch_knp(knpemi_dg* h, int flags) {
  DgProf prof(h, 1);
  if (h->NV == 8) return kn_dg_hex_launch_knp(h->stream, h->dev, h->d_consts, h->K - 1, !(flags & KNPEMI_NO_SPLITTING), h->hex_box);
  const int NV = h->NV, rpb = (DG_BLOCK / NV) * NV;
  const int nblocks = (h->dev.n_dof + rpb - 1) / rpb, chunk = (nblocks + 7) / 8;
  const int split = !(flags & KNPEMI_NO_SPLITTING);
  return NV == 3 ? launch_knp_nv<3>(h, chunk, split) : launch_knp_nv<4>(h, chunk, split);
}

}  // namespace

extern "C" void knpemi_dg_destroy(knpemi_dg* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  kn_comm_free(h->comm);
  if (h->sol) {
    kn_amg_async_join(h->sol->amg_emi);
    kn_amg_async_join(h->sol->amg_knp);
    kn_amg_free(h->sol->amg_emi);
    kn_amg_free(h->sol->amg_knp);
    if (h->sol->kry_pinned) (void)hipHostFree(h->sol->kry_pinned);
    if (h->sol->pub_host) (void)hipHostFree(h->sol->pub_host);
    kn_fused_graphs_free(h->sol);
    if (h->sol->graph_emi.exec) (void)hipGraphExecDestroy(h->sol->graph_emi.exec);
    if (h->sol->graph_knp.exec) (void)hipGraphExecDestroy(h->sol->graph_knp.exec);
    for (void* p : h->sol->allocs) (void)hipFree(p);
    delete h->sol;
  }
  if (h->d_coef) (void)hipFree(h->d_coef);
  for (auto& v : h->prof_ev) for (hipEvent_t e : v) (void)hipEventDestroy(e);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int knpemi_dg_create(const knpemi_dg_desc* d, int device, knpemi_dg** out) {
  if (!d || !out) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: null argument");
  *out = nullptr;
  int NV, NFC, NF, GD;   // vertices and facets per cell, vertices per facet, dimension
  if (d->cell_kind == KNPEMI_TRIANGLE) { NV = 3; NFC = 3; NF = 2; GD = 2; }
  else if (d->cell_kind == KNPEMI_TETRAHEDRON) { NV = 4; NFC = 4; NF = 3; GD = 3; }
  else if (d->cell_kind == KNPEMI_HEXAHEDRON) { NV = 8; NFC = 6; NF = 4; GD = 3; }
  else return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: unknown cell kind");
  const bool hex = NV == 8;
  // local vertex of local facet f at facet position m.  Simplices: facet f is opposite vertex f, its vertices in
  // increasing local order.  Hexahedra (tensor numbering: bit a of a local vertex is its coordinate along axis a):
  // facet f = 2 a + b is xi_a = b, position m has bit 0 along the lower and bit 1 along the higher remaining axis.
  auto facet_vertex = [&](int f, int m) {
    if (!hex) return m < f ? m : m + 1;
    const int a = f >> 1, b = f & 1, a1 = a == 0 ? 1 : 0, a2 = a == 2 ? 1 : 2;
    return (b << a) | ((m & 1) << a1) | ((m >> 1) << a2);
  };
  if (d->n_ions < 2 || d->n_ions > KN_MAXK) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: 2 to 4 ionic species are supported");
  if (d->n_sub < 1 || d->n_sub > KN_MAXSUB) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: bad n_sub");
  if (d->n_cells < 1 || d->n_cells * NV >= (int64_t)1 << 31) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: bad n_cells");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return dg_fail(KNPEMI_EHIP, "knpemi_dg_create: no HIP device visible (the hot path has no CPU fallback)");
  if (device < 0 || device >= ndev) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: bad device index");
  KN_HIP(hipSetDevice(device));
  auto* h = new knpemi_dg();
  std::unique_ptr<knpemi_dg, void (*)(knpemi_dg*)> guard(h, knpemi_dg_destroy);
  h->device = device; h->NV = NV; h->K = d->n_ions; h->n_sub = d->n_sub;
  h->NFC = NFC; h->NFV = NF;
  KN_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  KN_HIP(hipEventCreate(&h->ev0));
  KN_HIP(hipEventCreate(&h->ev1));

  const int nc = (int)d->n_cells, nmf = (int)d->n_mem_facets, n = nc * NV;
  std::vector<double> box_h;      // hexahedral box meshes: the cells' edge lengths along their local axes
  for (int c = 0; c < nc; ++c)
    if (d->cell_sub[c] < 0 || d->cell_sub[c] >= d->n_sub) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: cell_sub out of range");
  if (hex) {
    // tensor-product vertex order (bit a of a local vertex = its coordinate along axis a): the trilinear map of a cell
    // given in another order (e.g. the counter-clockwise order of other formats) folds over, i.e. its Jacobian
    // determinant changes sign between the corners
    for (int c = 0; c < nc; ++c) {
      int pos = 0, neg = 0;
      for (int j = 0; j < 8; ++j) {
        double e[3][3];
        for (int a = 0; a < 3; ++a) {
          const int v0 = d->cells[(size_t)c * 8 + j], v1 = d->cells[(size_t)c * 8 + (j ^ (1 << a))];
          if (v0 < 0 || v0 >= d->n_vertices || v1 < 0 || v1 >= d->n_vertices)
            return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: vertex id out of range");
          const double sgn = ((j >> a) & 1) ? -1.0 : 1.0;
          for (int k = 0; k < 3; ++k) e[a][k] = sgn * (d->x[(size_t)v1 * 3 + k] - d->x[(size_t)v0 * 3 + k]);
        }
        const double det = e[0][0] * (e[1][1] * e[2][2] - e[1][2] * e[2][1]) - e[0][1] * (e[1][0] * e[2][2] - e[1][2] * e[2][0]) +
                           e[0][2] * (e[1][0] * e[2][1] - e[1][1] * e[2][0]);
        pos += det > 0.0;
        neg += det < 0.0;
      }
      if (pos != 8 && neg != 8)
        return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: hexahedron " + std::to_string(c) +
                                          " is degenerate or not in tensor-product vertex order");
    }
    // box mesh (every mesh of the reference's 3-D driver, make_mesh_3D.py:100-102): all cells orthogonal parallelepipeds to
    // 1e-12 of their size -- the kernels then use constant facet frames (kernels_dg_hex.hip); KNPEMI_DG_HEX_GENERAL=1
    // keeps the general kernels (tests run both on the same mesh)
    bool box = getenv("KNPEMI_DG_HEX_GENERAL") == nullptr;
    box_h.assign((size_t)nc * 3, 0.0);
    for (int c = 0; c < nc && box; ++c) {
      const int32_t* cv = d->cells + (size_t)c * 8;
      const double* x0 = d->x + (size_t)cv[0] * 3;
      double e[3][3], len[3];
      for (int t = 0; t < 3; ++t) {
        const double* xt = d->x + (size_t)cv[1 << t] * 3;
        for (int k = 0; k < 3; ++k) e[t][k] = xt[k] - x0[k];
        len[t] = std::sqrt(e[t][0] * e[t][0] + e[t][1] * e[t][1] + e[t][2] * e[t][2]);
      }
      for (int t = 0; t < 3; ++t) box_h[(size_t)c * 3 + t] = len[t];
      const double size = std::max(len[0], std::max(len[1], len[2]));
      for (int a = 0; a < 3 && box; ++a)
        for (int b = a + 1; b < 3; ++b)
          if (std::fabs(e[a][0] * e[b][0] + e[a][1] * e[b][1] + e[a][2] * e[b][2]) > 1e-12 * len[a] * len[b]) box = false;
      for (int j = 0; j < 8 && box; ++j) {
        const double* xj = d->x + (size_t)cv[j] * 3;
        for (int k = 0; k < 3; ++k) {
          const double want = x0[k] + ((j & 1) ? e[0][k] : 0.0) + ((j & 2) ? e[1][k] : 0.0) + ((j & 4) ? e[2][k] : 0.0);
          if (std::fabs(xj[k] - want) > 1e-12 * size) box = false;
        }
      }
    }
    h->hex_box = box ? 1 : 0;
  }
  // facet -> (cell, local facet) by sorting the facets' sorted vertex tuples; membrane facets ride along with id < 0
  struct Ent { std::array<int, 4> key; int id; };
  std::vector<Ent> ents;
  const int nfac = nc * NFC;
  ents.reserve((size_t)nfac + nmf);
  auto sorted_key = [&](std::array<int, 4> k) {
    std::sort(k.begin(), k.begin() + NF);
    return k;
  };
  for (int c = 0; c < nc; ++c) {
    for (int a = 0; a < NV; ++a)
      if (d->cells[(size_t)c * NV + a] < 0 || d->cells[(size_t)c * NV + a] >= d->n_vertices)
        return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: vertex id out of range");
    for (int f = 0; f < NFC; ++f) {
      std::array<int, 4> k = {-1, -1, -1, -1};
      for (int m = 0; m < NF; ++m) k[m] = d->cells[(size_t)c * NV + facet_vertex(f, m)];
      ents.push_back({sorted_key(k), c * NFC + f});
    }
  }
  for (int m = 0; m < nmf; ++m) {
    std::array<int, 4> k = {-1, -1, -1, -1};
    for (int a = 0; a < NF; ++a) k[a] = d->mem_facets[(size_t)m * NF + a];
    ents.push_back({sorted_key(k), -1 - m});
  }
  std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.key != b.key ? a.key < b.key : a.id < b.id; });

  std::vector<int> nbr((size_t)nfac, -1), mfid((size_t)nfac, -1);
  std::vector<unsigned> finfo((size_t)nfac, 0u);
  std::vector<unsigned char> csub((size_t)nc);
  for (int c = 0; c < nc; ++c) csub[c] = (unsigned char)d->cell_sub[c];
  h->h_q2e.assign((size_t)nmf * NF, -1);
  h->h_q2i.assign((size_t)nmf * NF, -1);
  {   // continuous P1 numbering per sub-domain (vertices on the membrane are doubled): the solver's auxiliary space
    std::vector<int64_t> key((size_t)n);
    for (int c = 0; c < nc; ++c)
      for (int a = 0; a < NV; ++a) key[(size_t)c * NV + a] = (int64_t)d->cell_sub[c] * d->n_vertices + d->cells[(size_t)c * NV + a];
    std::vector<int64_t> uniq(key);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    h->n_aux = (int)uniq.size();
    h->aux_of.resize((size_t)n);
    for (int i = 0; i < n; ++i) h->aux_of[i] = (int)(std::lower_bound(uniq.begin(), uniq.end(), key[i]) - uniq.begin());
  }
  auto local_of = [&](int c, int v) {
    for (int a = 0; a < NV; ++a) if (d->cells[(size_t)c * NV + a] == v) return a;
    return -1;
  };
  for (size_t e = 0; e < ents.size();) {
    size_t e1 = e;
    while (e1 < ents.size() && ents[e1].key == ents[e].key) ++e1;
    int mem = -1, cf[2], ncf = 0;
    for (size_t t = e; t < e1; ++t) {
      if (ents[t].id < 0) {
        if (mem >= 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: membrane facet listed twice");
        mem = -1 - ents[t].id;
      } else {
        if (ncf == 2) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: a facet is shared by more than two cells");
        cf[ncf++] = ents[t].id;
      }
    }
    e = e1;
    if (ncf < 2) {
      if (mem >= 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: membrane facet is not shared by two cells");
      continue;
    }
    const int c0 = cf[0] / NFC, f0 = cf[0] % NFC, c1 = cf[1] / NFC, f1 = cf[1] % NFC;
    const int s0 = d->cell_sub[c0], s1 = d->cell_sub[c1];
    int kind0 = 1, kind1 = 1;
    if (mem >= 0) {
      if (!((s0 == 0) != (s1 == 0)))
        return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: a membrane facet must separate an ECS cell from a cell of a sub-domain > 0");
      kind0 = s0 == 0 ? 2 : 3;
      kind1 = s1 == 0 ? 2 : 3;
    } else if (s0 != s1) {
      return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: facet between two sub-domains is not in mem_facets");
    }
    for (int side = 0; side < 2; ++side) {
      const int c = side ? c1 : c0, f = side ? f1 : f0, o = side ? c0 : c1, fo = side ? f0 : f1;
      unsigned w = (unsigned)(side ? kind1 : kind0) | (unsigned)fo << 2;
      for (int m = 0; m < NF; ++m) {
        const int a = facet_vertex(f, m);
        const int v = d->cells[(size_t)c * NV + a];
        const int lo_ = local_of(o, v);
        if (lo_ < 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: two cells share a facet key but not its vertices");
        // simplices: 2-bit fields indexed by the cell's local vertex; hexahedra: 3-bit / 2-bit fields by facet position
        w |= hex ? (unsigned)lo_ << (5 + 3 * m) : (unsigned)lo_ << (4 + 2 * a);
        if (mem >= 0) {
          int node = -1;
          for (int t = 0; t < NF; ++t) if (d->mem_facets[(size_t)mem * NF + t] == v) node = t;
          if (node < 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: membrane facet vertices do not match the cells'");
          w |= hex ? (unsigned)node << (17 + 2 * m) : (unsigned)node << (12 + 2 * a);
          (d->cell_sub[c] == 0 ? h->h_q2e : h->h_q2i)[(size_t)mem * NF + node] = c * NV + a;
        }
      }
      if (hex) {   // the facet of the neighbour must be one of ITS facets with the four vertices at matching positions
        const int ao = fo >> 1;
        for (int m = 0; m < NF; ++m)
          if ((int)((((w >> (5 + 3 * m)) & 7) >> ao) & 1) != (fo & 1))
            return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: hexahedra are not in tensor-product vertex order");
      }
      nbr[(size_t)c * NFC + f] = o;
      finfo[(size_t)c * NFC + f] = w;
      mfid[(size_t)c * NFC + f] = mem;
    }
  }
  // CSR pattern: per row one nv-wide block per cell of {self} + neighbours, increasing cell order
  h->h_rowptr.assign((size_t)n + 1, 0);
  int64_t nnz = 0;
  for (int c = 0; c < nc; ++c) {
    int cnt = 1;
    for (int f = 0; f < NFC; ++f) cnt += nbr[(size_t)c * NFC + f] >= 0;
    for (int i = 0; i < NV; ++i) { nnz += (int64_t)cnt * NV; h->h_rowptr[(size_t)c * NV + i + 1] = (int)nnz; }
    if (nnz >= ((int64_t)1 << 31) - 64) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: more than 2^31 matrix entries");
  }
  h->h_colind.resize((size_t)nnz);
  for (int c = 0; c < nc; ++c) {
    int blk[7], cnt = 0;
    blk[cnt++] = c;
    for (int f = 0; f < NFC; ++f) if (nbr[(size_t)c * NFC + f] >= 0) blk[cnt++] = nbr[(size_t)c * NFC + f];
    std::sort(blk, blk + cnt);
    for (int i = 0; i < NV; ++i) {
      int* ci = h->h_colind.data() + h->h_rowptr[(size_t)c * NV + i];
      for (int b = 0; b < cnt; ++b) for (int j = 0; j < NV; ++j) *ci++ = blk[b] * NV + j;
    }
  }
  std::vector<double> rec((size_t)n * KN_REC, 0.0);
  for (int c = 0; c < nc; ++c)
    for (int j = 0; j < NV; ++j)
      for (int dd = 0; dd < GD; ++dd) rec[((size_t)c * NV + j) * KN_REC + dd] = d->x[(size_t)d->cells[(size_t)c * NV + j] * GD + dd];

  DgDev& D = h->dev;
  D.n_cell = nc; D.n_dof = n; D.nq = nmf * NF; D.nnz = nnz;
  int rc;
  const double* crec = nullptr;
  if ((rc = dg_upload(h, rec, &crec))) return rc;
  D.rec = const_cast<double*>(crec);
  if ((rc = dg_upload(h, nbr, &D.nbr))) return rc;
  if ((rc = dg_upload(h, finfo, &D.finfo))) return rc;
  if ((rc = dg_upload(h, mfid, &D.mfid))) return rc;
  D.box_h = nullptr;
  if (h->hex_box && (rc = dg_upload(h, box_h, &D.box_h))) return rc;
  if ((rc = dg_upload(h, csub, &D.cell_sub))) return rc;
  if ((rc = dg_upload(h, h->h_rowptr, &D.rowptr))) return rc;
  if ((rc = dg_upload(h, h->h_q2e, &D.q2e))) return rc;
  if ((rc = dg_upload(h, h->h_q2i, &D.q2i))) return rc;
  if ((rc = dg_alloc(h, (size_t)nnz, &D.A_emi))) return rc;
  if ((rc = dg_alloc(h, (size_t)n, &D.b_emi))) return rc;
  if ((rc = dg_alloc(h, (size_t)(h->K - 1) * nnz, &D.A_knp))) return rc;
  if ((rc = dg_alloc(h, (size_t)(h->K - 1) * n, &D.b_knp))) return rc;
  if ((rc = dg_alloc(h, (size_t)D.nq, &D.phiM))) return rc;
  if ((rc = dg_alloc(h, (size_t)KN_MAXK * std::max(D.nq, 1), &D.Ich))) return rc;
  D.fsrc = nullptr;
  {
    std::vector<double> qt;
    D.nquad = kn_gamma_quadrature(NF, &qt);
    if ((rc = dg_upload(h, qt, &D.qtab))) return rc;
  }
  h->stage_len = (size_t)std::max(n, 1) * (size_t)std::max(1, h->K - 1);
  if ((rc = dg_alloc(h, h->stage_len, &h->d_stage))) return rc;
  if ((rc = dg_alloc(h, 1, &h->d_consts))) return rc;
  for (size_t t = 0; t < h->h_q2e.size(); ++t)
    if (h->h_q2e[t] < 0 || h->h_q2i[t] < 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_create: membrane facet without two cells");
  guard.release();
  *out = h;
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_set_params(knpemi_dg* h, const knpemi_dg_params* p) {
  if (!h || !p) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_params: null argument");
  if (!(p->dt > 0) || !(p->gamma > 0)) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_params: dt and gamma must be positive");
  KN_HIP(hipSetDevice(h->device));
  DgConsts& C = h->consts;
  std::memset(&C, 0, sizeof(C));
  C.n_sub = h->n_sub; C.K = h->K;
  C.F = p->F; C.psi = p->psi; C.C_M = p->C_M; C.dt = p->dt; C.inv_dt = 1.0 / p->dt; C.C_phi = p->C_M / p->dt; C.gamma = p->gamma;
  const double zK = p->z[h->K - 1];
  if (zK == 0.0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_params: the eliminated ion needs a non-zero valence");
  for (int k = 0; k < h->K; ++k) {
    C.z[k] = p->z[k];
    C.elim[k] = -(p->z[k] / zK);
    if (k < h->K - 1 && p->z[k] == 0.0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_params: zero valence");
  }
  for (int s = 0; s < h->n_sub; ++s) {
    for (int k = 0; k < h->K; ++k) {
      const double Dk = p->D[s][k], z = p->z[k];
      C.D[s][k] = Dk;
      C.kap[s][k] = p->F * z * z * Dk * p->psi;
      C.sig[s][k] = p->F * z * Dk;
      C.az2D[s][k] = Dk * z * z;
    }
    C.rho_term[s] = -(1.0 / zK) * p->rho_z * p->rho[s];
  }
  KN_HIP(hipMemcpyAsync(h->d_consts, &C, sizeof(C), hipMemcpyHostToDevice, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  h->have_params = 1;
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_dims(knpemi_dg* h, int64_t* n_dofs, int64_t* nnz, int64_t* n_mem_nodes) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_dims: null handle");
  if (n_dofs) *n_dofs = h->dev.n_dof;
  if (nnz) *nnz = h->dev.nnz;
  if (n_mem_nodes) *n_mem_nodes = h->dev.nq;
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_get_pattern(knpemi_dg* h, int32_t* rowptr, int32_t* colind) {
  if (!h || !rowptr || !colind) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_get_pattern: null argument");
  std::memcpy(rowptr, h->h_rowptr.data(), h->h_rowptr.size() * sizeof(int));
  std::memcpy(colind, h->h_colind.data(), h->h_colind.size() * sizeof(int));
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_get_membrane_dofs(knpemi_dg* h, int32_t* dof_e, int32_t* dof_i) {
  if (!h || !dof_e || !dof_i) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_get_membrane_dofs: null argument");
  std::memcpy(dof_e, h->h_q2e.data(), h->h_q2e.size() * sizeof(int));
  std::memcpy(dof_i, h->h_q2i.data(), h->h_q2i.size() * sizeof(int));
  return KNPEMI_OK;
}

namespace {
// (device pointer, length, record slot or -1) of a field
int dg_field(knpemi_dg* h, int field, int idx, double** ptr, size_t* len, int* slot) {
  const DgDev& D = h->dev;
  *slot = -1;
  switch (field) {
    case KNPEMI_DG_C:
      if (idx < 0 || idx >= h->K) return dg_fail(KNPEMI_EINVAL, "knpemi_dg field: ion index out of range");
      *slot = KN_CSLOT(idx); *ptr = D.rec; *len = D.n_dof; return 0;
    case KNPEMI_DG_PHI: *slot = 7; *ptr = D.rec; *len = D.n_dof; return 0;
    case KNPEMI_DG_PHI_M: *ptr = D.phiM; *len = D.nq; return 0;
    case KNPEMI_DG_I_CH:
      if (idx < 0 || idx >= h->K) return dg_fail(KNPEMI_EINVAL, "knpemi_dg field: ion index out of range");
      *ptr = D.Ich + (size_t)idx * D.nq; *len = D.nq; return 0;
    case KNPEMI_DG_SOURCE:
      if (idx < 0 || idx >= h->K - 1) return dg_fail(KNPEMI_EINVAL, "knpemi_dg field: ion index out of range");
      if (!h->d_fsrc) {
        int rc = dg_alloc(h, (size_t)(h->K - 1) * D.n_dof, &h->d_fsrc);
        if (rc) return rc;
        h->dev.fsrc = h->d_fsrc;
      }
      *ptr = h->d_fsrc + (size_t)idx * D.n_dof; *len = D.n_dof; return 0;
  }
  return dg_fail(KNPEMI_EINVAL, "knpemi_dg field: unknown field");
}
}  // namespace

extern "C" int knpemi_dg_set_field(knpemi_dg* h, int field, int idx, const double* host, size_t n) {
  if (!h || !host) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_field: null argument");
  KN_HIP(hipSetDevice(h->device));
  double* ptr; size_t len; int slot;
  int rc = dg_field(h, field, idx, &ptr, &len, &slot);
  if (rc) return rc;
  if (n != len) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_field: length mismatch");
  if (len == 0) return KNPEMI_OK;
  if (slot < 0) {
    KN_HIP(hipMemcpyAsync(ptr, host, len * sizeof(double), hipMemcpyHostToDevice, h->stream));
  } else {
    KN_HIP(hipMemcpyAsync(h->d_stage, host, len * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(dg_slot_kernel, dim3((len + 255) / 256), dim3(256), 0, h->stream, ptr, slot, h->d_stage, (int)len, 1);
    if ((rc = dg_check_launch("dg_slot_kernel"))) return rc;
  }
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_get_field(knpemi_dg* h, int field, int idx, double* host, size_t n) {
  if (!h || !host) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_get_field: null argument");
  KN_HIP(hipSetDevice(h->device));
  double* ptr; size_t len; int slot;
  int rc = dg_field(h, field, idx, &ptr, &len, &slot);
  if (rc) return rc;
  if (n != len) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_get_field: length mismatch");
  if (len == 0) return KNPEMI_OK;
  if (slot < 0) {
    KN_HIP(hipMemcpyAsync(host, ptr, len * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  } else {
    hipLaunchKernelGGL(dg_slot_kernel, dim3((len + 255) / 256), dim3(256), 0, h->stream, ptr, slot, h->d_stage, (int)len, 0);
    if ((rc = dg_check_launch("dg_slot_kernel"))) return rc;
    KN_HIP(hipMemcpyAsync(host, h->d_stage, len * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_assemble_emi(knpemi_dg* h, int flags) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_assemble_emi: null handle");
  if (!h->have_params) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_assemble_emi: knpemi_dg_set_params has not been called");
  KN_HIP(hipSetDevice(h->device));
  return launch_emi(h, flags);
}

extern "C" int knpemi_dg_assemble_knp(knpemi_dg* h, int flags) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_assemble_knp: null handle");
  if (!h->have_params) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_assemble_knp: knpemi_dg_set_params has not been called");
  KN_HIP(hipSetDevice(h->device));
  return launch_knp(h, flags);
}

extern "C" int knpemi_dg_get_values(knpemi_dg* h, int which, double* vals) {
  if (!h || !vals || which < 0 || which > h->K - 1) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_get_values: bad argument");
  KN_HIP(hipSetDevice(h->device));
  const double* src = which == 0 ? h->dev.A_emi : h->dev.A_knp + (size_t)(which - 1) * h->dev.nnz;
  KN_HIP(hipMemcpyAsync(vals, src, (size_t)h->dev.nnz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_get_rhs(knpemi_dg* h, int which, double* b) {
  if (!h || !b || which < 0 || which > h->K - 1) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_get_rhs: bad argument");
  KN_HIP(hipSetDevice(h->device));
  const double* src = which == 0 ? h->dev.b_emi : h->dev.b_knp + (size_t)(which - 1) * h->dev.n_dof;
  KN_HIP(hipMemcpyAsync(b, src, (size_t)h->dev.n_dof * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_device_system(knpemi_dg* h, int which, const int32_t** rowptr, const int32_t** colind,
                                       const double** vals, const double** b) {
  if (!h || which < 0 || which > h->K - 1) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_device_system: bad argument");
  if (colind) {   // uploaded on first request: the assembly kernels never read the column indices
    if (!h->d_colind) {
      KN_HIP(hipSetDevice(h->device));
      int rc = dg_upload(h, h->h_colind, &h->d_colind);
      if (rc) return rc;
    }
    *colind = h->d_colind;
  }
  if (rowptr) *rowptr = h->dev.rowptr;
  if (vals) *vals = which == 0 ? h->dev.A_emi : h->dev.A_knp + (size_t)(which - 1) * h->dev.nnz;
  if (b) *b = which == 0 ? h->dev.b_emi : h->dev.b_knp + (size_t)(which - 1) * h->dev.n_dof;
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_update(knpemi_dg* h, const double* c_new, int on_device) {
  if (!h || !c_new) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_update: null argument");
  if (!h->have_params) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_update: knpemi_dg_set_params has not been called");
  KN_HIP(hipSetDevice(h->device));
  const size_t len = (size_t)(h->K - 1) * h->dev.n_dof;
  const double* src = c_new;
  if (!on_device) {
    KN_HIP(hipMemcpyAsync(h->d_stage, c_new, len * sizeof(double), hipMemcpyHostToDevice, h->stream));
    src = h->d_stage;
  }
  const int nt = std::max(h->dev.n_dof, h->dev.nq);
  hipLaunchKernelGGL(dg_update_kernel, dim3((nt + 255) / 256), dim3(256), 0, h->stream, h->dev, h->d_consts, src, h->NV);
  int rc = dg_check_launch("dg_update_kernel");
  if (rc) return rc;
  if (!on_device) KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

namespace {

// The solver handle of a DG problem: stream, systems and the potential component of the dof records (same 64-byte
// layout as the CG vertex records), nothing else.
int dg_solver(knpemi_dg* h, knpemi_handle** out) {
  if (h->sol) { *out = h->sol; return KNPEMI_OK; }
  const DgDev& D = h->dev;
  const int n = D.n_dof, KS = h->K - 1;
  if ((int64_t)KS * D.nnz >= ((int64_t)1 << 31) - 64)
    return dg_fail(KNPEMI_EINVAL, "knpemi_dg_solve: the block-diagonal concentration system has more than 2^31 entries");
  int rc;
  if (!h->d_colind && (rc = dg_upload(h, h->h_colind, &h->d_colind))) return rc;
  // block-diagonal pattern of the K - 1 concentration systems
  std::vector<int> krp((size_t)KS * n + 1), kci((size_t)KS * D.nnz);
  for (int k = 0; k < KS; ++k) {
    for (int i = 0; i < n; ++i) krp[(size_t)k * n + i] = (int)(k * D.nnz) + h->h_rowptr[i];
    for (int64_t j = 0; j < D.nnz; ++j) kci[(size_t)k * D.nnz + j] = k * n + h->h_colind[j];
  }
  krp[(size_t)KS * n] = (int)(KS * D.nnz);
  const int *d_krp = nullptr, *d_kci = nullptr;
  if ((rc = dg_upload(h, krp, &d_krp))) return rc;
  if ((rc = dg_upload(h, kci, &d_kci))) return rc;
  if ((rc = dg_alloc(h, (size_t)KS * n, &h->d_csol))) return rc;
  auto* s = new knpemi_handle();
  s->device = h->device;
  s->stream = s->cur = h->stream;
  s->K = h->K; s->n_sub = h->n_sub;
  s->dev.Ntot = n;
  s->dev.VR = D.rec;
  s->dev.rowptr = D.rowptr; s->dev.colind = h->d_colind;
  s->dev.A_emi = D.A_emi; s->dev.b_emi = D.b_emi;
  s->dev.krowptr = d_krp; s->dev.kcolind = d_kci;
  s->dev.A_knp = D.A_knp; s->dev.b_knp = D.b_knp;
  s->dev.csol = h->d_csol;
  s->plain_knp = true;
  // block columns of the systems (the cell and its facet neighbours, increasing, as the CSR pattern was built): the solver's
  // SpMVs read these instead of one column index per entry (block_spmv.h); KNPEMI_DG_NO_BLOCK_SPMV=1 keeps the CSR kernels
  if (!getenv("KNPEMI_DG_NO_BLOCK_SPMV")) {
    const int NV = h->NV, nc = n / NV;
    int nbmax = 1;
    for (int c = 0; c < nc; ++c) nbmax = std::max(nbmax, (h->h_rowptr[(size_t)c * NV + 1] - h->h_rowptr[(size_t)c * NV]) / NV);
    std::vector<int> bcol((size_t)nc * nbmax, -1);
    for (int c = 0; c < nc; ++c) {
      const int a = h->h_rowptr[(size_t)c * NV], nb = (h->h_rowptr[(size_t)c * NV + 1] - a) / NV;
      for (int b = 0; b < nb; ++b) bcol[(size_t)c * nbmax + b] = h->h_colind[(size_t)a + (size_t)b * NV] / NV;
    }
    const int* d_bcol = nullptr;
    if ((rc = dg_upload(h, bcol, &d_bcol))) return rc;
    s->bcols.bcol = d_bcol; s->bcols.nv = NV; s->bcols.nbmax = nbmax; s->bcols.n = n;
  }
  s->amg_emi.first_agg = h->aux_of;
  s->amg_emi.first_na = h->n_aux;
  s->amg_knp.first_agg.resize((size_t)KS * n);
  for (int k = 0; k < KS; ++k)
    for (int i = 0; i < n; ++i) s->amg_knp.first_agg[(size_t)k * n + i] = k * h->n_aux + h->aux_of[i];
  s->amg_knp.first_na = KS * h->n_aux;
  s->amg_emi.block = s->amg_knp.block = h->NV;       // block-Jacobi smoothing over the dofs of a cell
  // the vertex aggregates are split along the weakly penalised facets (KnAmg::split_first; KNPEMI_DG_AUX_UNSPLIT=1 keeps
  // the continuous P1 space as the first coarse level)
  s->amg_emi.split_first = s->amg_knp.split_first = !getenv("KNPEMI_DG_AUX_UNSPLIT");
  s->amg_emi.first_tentative = s->amg_knp.first_tentative = !getenv("KNPEMI_DG_AUX_SMOOTHED");
  s->amg_emi.positive_conflict = s->amg_knp.positive_conflict = !getenv("KNPEMI_DG_PLAIN_AGGREGATION");
  s->amg_emi.filter_theta = s->amg_knp.filter_theta = 0.02;   // (as kernels_krylov.hip)
  s->amg_emi.sub_fused = s->amg_knp.sub_fused = !getenv("KNPEMI_DG_NO_SUBCYCLE");   // merged transfer operators below the finest level
  if (const char* ft = getenv("KNPEMI_AMG_FILTER")) s->amg_emi.filter_theta = s->amg_knp.filter_theta = atof(ft);
  if (const char* th = getenv("KNPEMI_DG_THETA")) s->amg_emi.theta = s->amg_knp.theta = atof(th);
  if (getenv("KNPEMI_DG_PLAIN_AMG")) { s->amg_emi.first_na = s->amg_knp.first_na = 0; }
  if (getenv("KNPEMI_DG_POINT_JACOBI")) { s->amg_emi.block = s->amg_knp.block = 0; }
  h->sol = s;
  *out = s;
  return KNPEMI_OK;
}

}  // namespace

// The solver handle of a DG problem (a knpemi_handle that carries the stream, the two CSR systems and the dof records):
// knpemi_vec_gather / knpemi_vec_scatter take it for the ghost refresh of a solver vector on a cell partition.
extern "C" void* knpemi_dg_solver_handle(knpemi_dg* h) {
  if (!h) { kn_set_error("knpemi_dg_solver_handle: null handle"); return nullptr; }
  if (hipSetDevice(h->device) != hipSuccess) return nullptr;
  knpemi_handle* s = nullptr;
  return dg_solver(h, &s) == KNPEMI_OK ? s : nullptr;
}

// Cell partition (knpemi.dg.DGSlab): knpemi_dg_solve_emi / knpemi_dg_solve_knp become solves of the GLOBAL systems, as
// knpemi_set_distributed does for the CG path -- rows of owned cells only (ghost rows become identity rows), every SpMV
// after a ghost refresh of its argument (`halo`), dot products over owned dofs summed with `allreduce`, each rank's
// auxiliary-space AMG on its own diagonal block.  owned: one byte per local broken dof (1 = dof of an owned cell), NULL
// switches back to single-rank solves.  Vector orders: KNPEMI_B_EMI one value per dof, KNPEMI_B_KNP [solved ion][dof].
extern "C" int knpemi_dg_set_distributed(knpemi_dg* h, const uint8_t* owned, void* reduce_buf_dev, knpemi_allreduce_fn allreduce,
                                         knpemi_halo_fn halo, void* ctx) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_distributed: null handle");
  KN_HIP(hipSetDevice(h->device));
  knpemi_handle* s = nullptr;
  int rc = dg_solver(h, &s);
  if (rc) return rc;
  KnDist& d = s->dist;
  const int n = h->dev.n_dof, KS = h->K - 1;
  s->amg_emi.built = false;      // the preconditioner changes with the ownership
  s->amg_knp.built = false;
  // first aggregates of the auxiliary space: all broken dofs at one (sub-domain, mesh vertex) -- on a partition the dofs of
  // ghost cells (identity rows of the rank's diagonal block) get aggregates of their own
  std::vector<int> agg((size_t)n);
  int na = h->n_aux;
  if (owned) {
    std::vector<int> ghost_of((size_t)h->n_aux, -1);
    for (int i = 0; i < n; ++i) {
      if (owned[i]) { agg[i] = h->aux_of[i]; continue; }
      int& g = ghost_of[h->aux_of[i]];
      if (g < 0) g = na++;
      agg[i] = g;
    }
    // an auxiliary vertex all of whose dofs are ghosts leaves an empty aggregate behind: renumber the used ones
    std::vector<int> used((size_t)na, 0), renum((size_t)na, -1);
    for (int i = 0; i < n; ++i) used[agg[i]] = 1;
    int cnt = 0;
    for (int a = 0; a < na; ++a) if (used[a]) renum[a] = cnt++;
    for (int i = 0; i < n; ++i) agg[i] = renum[agg[i]];
    na = cnt;
  } else {
    agg = h->aux_of;
  }
  if (!getenv("KNPEMI_DG_PLAIN_AMG")) {
    s->amg_emi.first_agg = agg;
    s->amg_emi.first_na = na;
    s->amg_knp.first_agg.resize((size_t)KS * n);
    for (int k = 0; k < KS; ++k)
      for (int i = 0; i < n; ++i) s->amg_knp.first_agg[(size_t)k * n + i] = k * na + agg[i];
    s->amg_knp.first_na = KS * na;
  }
  if (!owned) { d.on = false; return KNPEMI_OK; }
  if (!reduce_buf_dev || !allreduce || !halo)
    return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_distributed: reduction buffer and both communication hooks are required");
  d.h_owned_emi.assign(owned, owned + n);
  d.h_owned_knp.resize((size_t)KS * n);
  for (int k = 0; k < KS; ++k) std::copy(owned, owned + n, d.h_owned_knp.begin() + (size_t)k * n);
  const uint8_t* p = nullptr;
  if ((rc = dg_upload(h, d.h_owned_emi, &p))) return rc;
  d.d_owned_emi = const_cast<uint8_t*>(p);
  if ((rc = dg_upload(h, d.h_owned_knp, &p))) return rc;
  d.d_owned_knp = const_cast<uint8_t*>(p);
  d.d_red = static_cast<double*>(reduce_buf_dev);
  d.allreduce = allreduce; d.halo = halo; d.ctx = ctx;
  double cnt = 0.0;   // owned dofs over all ranks (mean of the constant null space)
  for (int i = 0; i < n; ++i) cnt += owned[i] ? 1.0 : 0.0;
  KN_HIP(hipMemcpyAsync(d.d_red, &cnt, sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (allreduce(ctx, 1)) return dg_fail(KNPEMI_EHIP, "knpemi_dg_set_distributed: allreduce hook failed");
  KN_HIP(hipMemcpyAsync(&cnt, d.d_red, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  d.n_owned_global = cnt;
  d.nc = 0;            // (no cross-rank coarse space for the DG systems)
  d.on = true;
  return KNPEMI_OK;
}

// Device solves of the two DG systems (pdeSolver.py:24-35,74-78,99-110 with the iterative options): CG on the
// symmetric interior-penalty potential system with the constants projected out, BiCGStab on the K - 1 concentration
// systems taken as one block-diagonal system; preconditioner = the smoothed-aggregation AMG of the CG path whose first
// coarse level is the continuous P1 space of every sub-domain (auxiliary-space correction) under a damped-Jacobi
// smoother on the broken dofs.  The potential goes into the dof records, the concentrations stay in the solver's
// buffer until knpemi_dg_update (update != 0 runs it right away).
extern "C" int knpemi_dg_solve_emi(knpemi_dg* h, double rtol, double atol, int maxit, int* iters, double* relres) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_solve_emi: null handle");
  KN_HIP(hipSetDevice(h->device));
  knpemi_handle* s = nullptr;
  int rc = dg_solver(h, &s);
  if (rc) return rc;
  if (h->extrapolate && (rc = kn_extrapolate_guess(s, KNPEMI_B_EMI))) return rc;   // phi <- 2 phi_n - phi_(n-1)
  return kn_solve_emi(s, rtol, atol, maxit, iters, relres);
}

// Initial guesses of the two solves: the linear extrapolation 2 x_n - x_(n-1) of the last two solutions instead of the
// last one (knpemi_extrapolate_guess of the CG path); only the starting point changes.
extern "C" int knpemi_dg_set_extrapolation(knpemi_dg* h, int on) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_set_extrapolation: null handle");
  h->extrapolate = on ? 1 : 0;
  if (h->sol) { h->sol->guess_have[0] = -1; h->sol->guess_have[1] = 0; }
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_solve_knp(knpemi_dg* h, double rtol, double atol, int maxit, int* iters, double* relres, int update) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_solve_knp: null handle");
  if (update && !h->have_params) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_solve_knp: knpemi_dg_set_params has not been called");
  KN_HIP(hipSetDevice(h->device));
  knpemi_handle* s = nullptr;
  int rc = dg_solver(h, &s);
  if (rc) return rc;
  const int n = h->dev.n_dof;
  for (int k = 0; k < h->K - 1; ++k)   // initial guess: the previous concentrations (ksp_initial_guess_nonzero)
    if ((rc = kn_launch_field_gather(s, h->dev.rec + KN_CSLOT(k), KN_REC, h->d_csol + (size_t)k * n, n))) return rc;
  if (h->extrapolate && (rc = kn_extrapolate_guess(s, KNPEMI_B_KNP))) return rc;
  if ((rc = kn_solve_knp(s, rtol, atol, maxit, iters, relres))) return rc;
  if (update) return knpemi_dg_update(h, h->d_csol, 1);
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_get_solution(knpemi_dg* h, double* c_host) {
  if (!h || !c_host) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_get_solution: null argument");
  if (!h->d_csol) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_get_solution: knpemi_dg_solve_knp has not been called");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipMemcpyAsync(c_host, h->d_csol, (size_t)(h->K - 1) * h->dev.n_dof * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_halo_pack(knpemi_dg* h, const int32_t* idx_dev, int n, double* buf_dev) {
  if (!h || n < 0 || (n > 0 && (!idx_dev || !buf_dev))) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_halo_pack: bad argument");
  if (n == 0) return KNPEMI_OK;
  KN_HIP(hipSetDevice(h->device));
  hipLaunchKernelGGL(dg_halo_kernel, dim3((5 * (size_t)n + 255) / 256), dim3(256), 0, h->stream, h->dev.rec, idx_dev, n, buf_dev, 0);
  return dg_check_launch("dg_halo_kernel");
}

extern "C" int knpemi_dg_halo_unpack(knpemi_dg* h, const int32_t* idx_dev, int n, const double* buf_dev) {
  if (!h || n < 0 || (n > 0 && (!idx_dev || !buf_dev))) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_halo_unpack: bad argument");
  if (n == 0) return KNPEMI_OK;
  KN_HIP(hipSetDevice(h->device));
  hipLaunchKernelGGL(dg_halo_kernel, dim3((5 * (size_t)n + 255) / 256), dim3(256), 0, h->stream, h->dev.rec, idx_dev, n,
                     const_cast<double*>(buf_dev), 1);
  return dg_check_launch("dg_halo_kernel");
}

extern "C" int knpemi_dg_comm_init(knpemi_dg* h, int rank, int world, const char* id_bytes, size_t len) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_comm_init: null handle");
  if (h->comm) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_comm_init: communicator already created");
  int rc = kn_comm_create(h->device, rank, world, id_bytes, len, &h->comm);
  if (rc) return rc;
  h->comm_world = world;
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_comm_sendrecv(knpemi_dg* h, const double* send_buf_dev, double* recv_buf_dev, int n_parts,
                                       const int32_t* peer, const int64_t* send_off, const int64_t* send_cnt,
                                       const int64_t* recv_off, const int64_t* recv_cnt) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_comm_sendrecv: null handle");
  return kn_comm_sendrecv(h->comm, h->comm_world, h->device, h->stream, send_buf_dev, recv_buf_dev, n_parts, peer, send_off,
                          send_cnt, recv_off, recv_cnt);
}

extern "C" int knpemi_dg_sync(knpemi_dg* h) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_sync: null handle");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_time_kernel(knpemi_dg* h, int which, int flags, int reps, double* avg_ms) {
  if (!h || !avg_ms || reps < 1 || which < 0 || which > 1) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_time_kernel: bad argument");
  if (!h->have_params) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_time_kernel: knpemi_dg_set_params has not been called");
  KN_HIP(hipSetDevice(h->device));
  int rc = which == 0 ? launch_emi(h, flags) : launch_knp(h, flags);   // warm-up
  if (rc) return rc;
  KN_HIP(hipEventRecord(h->ev0, h->stream));
  for (int r = 0; r < reps; ++r)
    if ((rc = which == 0 ? launch_emi(h, flags) : launch_knp(h, flags))) return rc;
  KN_HIP(hipEventRecord(h->ev1, h->stream));
  KN_HIP(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  KN_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *avg_ms = (double)ms / reps;
  return KNPEMI_OK;
}

extern "C" void* knpemi_dg_stream(knpemi_dg* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int knpemi_dg_profile(knpemi_dg* h, int on) {
  if (!h) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_profile: null handle");
  h->prof_on = on > 0 ? on : 0;
  h->prof_count[0] = h->prof_count[1] = 0;
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_profile_read(knpemi_dg* h, int which, int64_t* launches, double* total_ms) {
  if (!h || which < 0 || which > 1) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_profile_read: bad argument");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamSynchronize(h->stream));
  double tot = 0.0;
  for (size_t i = 0; i + 1 < h->prof_used[which]; i += 2) {
    float ms = 0.f;
    KN_HIP(hipEventElapsedTime(&ms, h->prof_ev[which][i], h->prof_ev[which][i + 1]));
    tot += ms;
  }
  if (launches) *launches = (int64_t)(h->prof_used[which] / 2);
  if (total_ms) *total_ms = tot;
  h->prof_used[which] = 0;
  return KNPEMI_OK;
}

// ---- membrane ODE sweep over the membrane nodes ------------------------------------------------------------------
extern "C" int knpemi_dg_ode_bind(knpemi_dg* h, int model_id, int n_states, int n_params, const double* states,
                                  const double* params, const int32_t* ion_param, int v_index) {
  if (!h || !states || !params || !ion_param) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_bind: null argument");
  static const int ns_of[3] = {4, 4, 1}, np_of[3] = {22, 22, 23};
  if (model_id < 0 || model_id > 2) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_bind: unknown model id");
  if (n_states != ns_of[model_id] || n_params != np_of[model_id])
    return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_bind: state/parameter count does not match the model");
  if (h->ode_model >= 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_bind: model already bound");
  if (v_index < 0 || v_index >= n_states) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_bind: bad v_index");
  for (int i = 0; i < 3 * h->K; ++i)
    if (ion_param[i] < 0 || ion_param[i] >= n_params) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_bind: parameter index out of range");
  KN_HIP(hipSetDevice(h->device));
  const int nq = h->dev.nq;
  int rc;
  if ((rc = dg_alloc(h, (size_t)n_states * nq, &h->d_states))) return rc;
  if ((rc = dg_alloc(h, (size_t)n_params * nq, &h->d_params))) return rc;
  h->ode_blocks = (int)(((size_t)nq * n_states + 63) / 64) + 1;
  if ((rc = dg_alloc(h, 3 * (size_t)h->ode_blocks, &h->d_stats))) return rc;
  std::vector<double> t((size_t)std::max(n_states, n_params) * std::max(nq, 1));
  for (int pass = 0; pass < 2; ++pass) {
    const int cols = pass ? n_params : n_states;
    const double* src = pass ? params : states;
    for (int r = 0; r < nq; ++r) for (int c = 0; c < cols; ++c) t[(size_t)c * nq + r] = src[(size_t)r * cols + c];
    if (nq) KN_HIP(hipMemcpy(pass ? h->d_params : h->d_states, t.data(), (size_t)cols * nq * sizeof(double), hipMemcpyHostToDevice));
  }
  if ((rc = kn_lsoda_coef_upload(&h->d_coef))) return rc;
  for (int i = 0; i < 3 * KN_MAXK; ++i) h->ode_ion_param[i] = i < 3 * h->K ? ion_param[i] : 0;
  h->ode_model = model_id; h->ode_ns = n_states; h->ode_np = n_params; h->ode_v = v_index;
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_ode_step(knpemi_dg* h, double t0, double dt, double rtol, double atol, int flags) {
  if (!h || h->ode_model < 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_step: no membrane model bound");
  if (h->dev.nq == 0) return KNPEMI_OK;
  KN_HIP(hipSetDevice(h->device));
  const DgDev& D = h->dev;
  OdeDev dv{D.rec, D.q2e, D.q2i, D.phiM, D.Ich};
  OdeArgs a;
  std::memset(&a, 0, sizeof(a));
  a.nq = D.nq; a.q0 = 0; a.n_stim = 0; a.flags = flags & (KNPEMI_ODE_SET_V | KNPEMI_ODE_SET_TRACES); a.v_index = h->ode_v;
  a.model_slot = 0; a.NQtot = D.nq; a.n_ions = h->K;
  for (int i = 0; i < 3 * KN_MAXK; ++i) a.ion_param[i] = h->ode_ion_param[i];
  a.t0 = t0; a.dt = dt; a.rtol = rtol; a.atol = atol;
  a.states = h->d_states; a.params = h->d_params; a.mask = nullptr; a.stats = h->d_stats; a.stamps = nullptr;
  return kn_launch_ode_raw(h->stream, h->ode_model, dv, a, h->d_coef);
}

extern "C" int knpemi_dg_ode_get_tables(knpemi_dg* h, double* states, double* params) {
  if (!h || h->ode_model < 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_get_tables: no membrane model bound");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamSynchronize(h->stream));
  const int nq = h->dev.nq;
  std::vector<double> t((size_t)std::max(h->ode_ns, h->ode_np) * std::max(nq, 1));
  for (int pass = 0; pass < 2; ++pass) {
    double* dst = pass ? params : states;
    if (!dst || !nq) continue;
    const int cols = pass ? h->ode_np : h->ode_ns;
    KN_HIP(hipMemcpy(t.data(), pass ? h->d_params : h->d_states, (size_t)cols * nq * sizeof(double), hipMemcpyDeviceToHost));
    for (int r = 0; r < nq; ++r) for (int c = 0; c < cols; ++c) dst[(size_t)r * cols + c] = t[(size_t)c * nq + r];
  }
  return KNPEMI_OK;
}

extern "C" int knpemi_dg_ode_stats(knpemi_dg* h, int64_t* n_rhs, int64_t* n_steps, int64_t* n_failed) {
  if (!h || h->ode_model < 0) return dg_fail(KNPEMI_EINVAL, "knpemi_dg_ode_stats: no membrane model bound");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamSynchronize(h->stream));
  std::vector<unsigned long long> part(3 * (size_t)h->ode_blocks);
  KN_HIP(hipMemcpyAsync(part.data(), h->d_stats, part.size() * sizeof(part[0]), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipMemsetAsync(h->d_stats, 0, part.size() * sizeof(part[0]), h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  unsigned long long st[3] = {0, 0, 0};
  for (size_t i = 0; i < part.size(); ++i) st[i % 3] += part[i];
  if (n_rhs) *n_rhs = (int64_t)st[0];
  if (n_steps) *n_steps = (int64_t)st[1];
  if (n_failed) *n_failed = (int64_t)st[2];
  if (st[2]) return dg_fail(KNPEMI_EODE, "LSODA failed on at least one membrane node (odeSolver.py:121 `assert success`)");
  return KNPEMI_OK;
}
