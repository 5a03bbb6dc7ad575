// gfx950 assembly kernels of the knpemi hot path.
//
// Design: "owner-computes rows".  One thread owns one matrix row
// (= one sub-mesh vertex); it walks the cells incident to its vertex (sliced-ELL list, coalesced
// index reads), recomputes the closed-form P1 element row (or the Gauss-quadrature Q1 row) from
// 64-byte vertex records gathered through L2, and accumulates into the block's slice of the CSR
// value array held in LDS.  A block of 256 consecutive rows owns a contiguous CSR range, so the
// final write to HBM is a plain coalesced stream: no atomics, bit-reproducible results, and the
// matrix, the preconditioner and the RHS come out of one pass over the mesh.
//
// Forms restated (paths relative to the reference repository):
//   EMI  a, p, L : src/knpemi/emiWeakForm.py:138-241
//   KNP  a, L    : src/knpemi/knpWeakForm.py:123-216
//   update       : src/knpemi/utils.py:238-295
#include "knpemi_internal.h"

namespace {

__device__ __forceinline__ int logical_block(int bid, int nb) {
  // Workgroups are dealt round-robin over the 8 XCDs; give each XCD one contiguous chunk of row
  // blocks so neighbouring rows (which share vertices and cells) meet in the same L2.
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

struct Rec {
  double x, y, z, c0, c1, c2, phi;
};

__device__ __forceinline__ Rec load_rec(const double* __restrict__ VR, int v) {
  const double4* p = reinterpret_cast<const double4*>(VR) + 2 * (size_t)v;
  const double4 a = p[0], b = p[1];
  Rec r;
  r.x = a.x; r.y = a.y; r.z = a.z; r.c0 = b.x; r.c1 = b.y; r.c2 = b.z; r.phi = b.w;
  return r;
}

// Gradient dot products d[j] = grad(lambda_li) . grad(lambda_j) and the cell measure of a P1
// simplex (closed form; FFCx reaches the same numbers with a 1-point rule).
template <int GDIM>
__device__ __forceinline__ double simplex_row(const Rec (&r)[GDIM + 1], int li, double (&d)[GDIM + 1]);

template <>
__device__ __forceinline__ double simplex_row<2>(const Rec (&r)[3], int li, double (&d)[3]) {
  const double e1x = r[1].x - r[0].x, e1y = r[1].y - r[0].y;
  const double e2x = r[2].x - r[0].x, e2y = r[2].y - r[0].y;
  const double det = e1x * e2y - e1y * e2x, inv = 1.0 / det;
  const double g1x = e2y * inv, g1y = -e2x * inv;
  const double g2x = -e1y * inv, g2y = e1x * inv;
  const double g0x = -(g1x + g2x), g0y = -(g1y + g2y);
  const double lx = li == 0 ? g0x : (li == 1 ? g1x : g2x);
  const double ly = li == 0 ? g0y : (li == 1 ? g1y : g2y);
  d[0] = lx * g0x + ly * g0y;
  d[1] = lx * g1x + ly * g1y;
  d[2] = lx * g2x + ly * g2y;
  return 0.5 * fabs(det);
}

template <>
__device__ __forceinline__ double simplex_row<3>(const Rec (&r)[4], int li, double (&d)[4]) {
  const double ax = r[1].x - r[0].x, ay = r[1].y - r[0].y, az = r[1].z - r[0].z;
  const double bx = r[2].x - r[0].x, by = r[2].y - r[0].y, bz = r[2].z - r[0].z;
  const double cx = r[3].x - r[0].x, cy = r[3].y - r[0].y, cz = r[3].z - r[0].z;
  // cofactors: grad l1 = (b x c)/det, grad l2 = (c x a)/det, grad l3 = (a x b)/det
  double g1x = by * cz - bz * cy, g1y = bz * cx - bx * cz, g1z = bx * cy - by * cx;
  double g2x = cy * az - cz * ay, g2y = cz * ax - cx * az, g2z = cx * ay - cy * ax;
  double g3x = ay * bz - az * by, g3y = az * bx - ax * bz, g3z = ax * by - ay * bx;
  const double det = ax * g1x + ay * g1y + az * g1z, inv = 1.0 / det;
  g1x *= inv; g1y *= inv; g1z *= inv;
  g2x *= inv; g2y *= inv; g2z *= inv;
  g3x *= inv; g3y *= inv; g3z *= inv;
  const double g0x = -(g1x + g2x + g3x), g0y = -(g1y + g2y + g3y), g0z = -(g1z + g2z + g3z);
  const double lx = li == 0 ? g0x : (li == 1 ? g1x : (li == 2 ? g2x : g3x));
  const double ly = li == 0 ? g0y : (li == 1 ? g1y : (li == 2 ? g2y : g3y));
  const double lz = li == 0 ? g0z : (li == 1 ? g1z : (li == 2 ? g2z : g3z));
  d[0] = lx * g0x + ly * g0y + lz * g0z;
  d[1] = lx * g1x + ly * g1y + lz * g1z;
  d[2] = lx * g2x + ly * g2y + lz * g2z;
  d[3] = lx * g3x + ly * g3y + lz * g3z;
  return fabs(det) * (1.0 / 6.0);
}

// measure of a membrane facet from its own-side vertex records
template <int NF>
__device__ __forceinline__ double facet_measure(const Rec (&p)[NF]) {
  if constexpr (NF == 2) {
    const double dx = p[1].x - p[0].x, dy = p[1].y - p[0].y;
    return sqrt(dx * dx + dy * dy);
  } else {
    const double ax = p[1].x - p[0].x, ay = p[1].y - p[0].y, az = p[1].z - p[0].z;
    const double bx = p[2].x - p[0].x, by = p[2].y - p[0].y, bz = p[2].z - p[0].z;
    const double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    return 0.5 * sqrt(nx * nx + ny * ny + nz * nz);
  }
}

// surface Jacobian of a bilinear quadrilateral facet (lexicographic vertices) at (xi, eta)
__device__ __forceinline__ double quad_jac(const Rec (&p)[4], double xi, double eta) {
  const double ux = (1 - eta) * (p[1].x - p[0].x) + eta * (p[3].x - p[2].x);
  const double uy = (1 - eta) * (p[1].y - p[0].y) + eta * (p[3].y - p[2].y);
  const double uz = (1 - eta) * (p[1].z - p[0].z) + eta * (p[3].z - p[2].z);
  const double vx = (1 - xi) * (p[2].x - p[0].x) + xi * (p[3].x - p[1].x);
  const double vy = (1 - xi) * (p[2].y - p[0].y) + xi * (p[3].y - p[1].y);
  const double vz = (1 - xi) * (p[2].z - p[0].z) + xi * (p[3].z - p[1].z);
  const double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
  return sqrt(nx * nx + ny * ny + nz * nz);
}

// Facet mass matrix row M[a][b], b = 0..NF-1 (degree-2 integrand: closed form on simplices,
// 2x2 Gauss on quadrilaterals as FFCx would choose).
template <int NF>
__device__ __forceinline__ void facet_mass_row(const Rec (&p)[NF], int a, double (&M)[NF]) {
  if constexpr (NF == 4) {
    const double g0 = 0.5 - 0.28867513459481287, g1 = 0.5 + 0.28867513459481287;
#pragma unroll
    for (int b = 0; b < 4; ++b) M[b] = 0.0;
#pragma unroll
    for (int qj = 0; qj < 2; ++qj)
#pragma unroll
      for (int qi = 0; qi < 2; ++qi) {
        const double xi = qi ? g1 : g0, eta = qj ? g1 : g0;
        const double N[4] = {(1 - xi) * (1 - eta), xi * (1 - eta), (1 - xi) * eta, xi * eta};
        const double Na = a == 0 ? N[0] : (a == 1 ? N[1] : (a == 2 ? N[2] : N[3]));
        const double w = 0.25 * quad_jac(p, xi, eta) * Na;
#pragma unroll
        for (int b = 0; b < 4; ++b) M[b] += w * N[b];
      }
  } else {
    const double m = facet_measure<NF>(p) * (1.0 / (NF * (NF + 1)));
#pragma unroll
    for (int b = 0; b < NF; ++b) M[b] = (b == a) ? 2.0 * m : m;
  }
}

// ---------------------------------------------------------------------------------------------
// Q1 hexahedron: 2x2x2 Gauss tables staged in LDS (basis values and reference gradients)
// ---------------------------------------------------------------------------------------------
struct HexTab {
  double N[8][8];      // [q][v]
  double dN[8][8][3];  // [q][v][t]
};

__device__ __forceinline__ void stage_hex_tables(HexTab* T) {
  const double g0 = 0.5 - 0.28867513459481287, g1 = 0.5 + 0.28867513459481287;
  for (int i = threadIdx.x; i < 64; i += blockDim.x) {
    const int q = i >> 3, v = i & 7;
    double f[3], df[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      const double x = ((q >> ax) & 1) ? g1 : g0;
      const bool hi = (v >> ax) & 1;
      f[ax] = hi ? x : 1.0 - x;
      df[ax] = hi ? 1.0 : -1.0;
    }
    T->N[q][v] = f[0] * f[1] * f[2];
    T->dN[q][v][0] = df[0] * f[1] * f[2];
    T->dN[q][v][1] = f[0] * df[1] * f[2];
    T->dN[q][v][2] = f[0] * f[1] * df[2];
  }
}

// Physical gradients of the 8 basis functions at Gauss point q; returns w*|det J| (w = 1/8).
__device__ __forceinline__ double hex_point(const HexTab* T, const Rec (&r)[8], int q, double (&G)[8][3]) {
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
  for (int v = 0; v < 8; ++v)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const double dn = T->dN[q][v][t];
      J[0][t] += r[v].x * dn; J[1][t] += r[v].y * dn; J[2][t] += r[v].z * dn;
    }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, inv = 1.0 / det;
  // Jinv[t][g]
  const double i00 = c00 * inv, i01 = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * inv,
               i02 = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * inv;
  const double i10 = c01 * inv, i11 = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * inv,
               i12 = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * inv;
  const double i20 = c02 * inv, i21 = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * inv,
               i22 = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * inv;
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const double a = T->dN[q][v][0], b = T->dN[q][v][1], c = T->dN[q][v][2];
    G[v][0] = a * i00 + b * i10 + c * i20;
    G[v][1] = a * i01 + b * i11 + c * i21;
    G[v][2] = a * i02 + b * i12 + c * i22;
  }
  return 0.125 * fabs(det);
}

template <int NV>
__device__ __forceinline__ void load_cell(const KnDev& D, int cell, Rec (&r)[NV]) {
  const int* cv = D.cells + (size_t)cell * NV;
  if constexpr (NV == 4 || NV == 8) {
    const int4 a = *reinterpret_cast<const int4*>(cv);
    r[0] = load_rec(D.VR, a.x); r[1] = load_rec(D.VR, a.y);
    r[2] = load_rec(D.VR, a.z); r[3] = load_rec(D.VR, a.w);
    if constexpr (NV == 8) {
      const int4 b = *reinterpret_cast<const int4*>(cv + 4);
      r[4] = load_rec(D.VR, b.x); r[5] = load_rec(D.VR, b.y);
      r[6] = load_rec(D.VR, b.z); r[7] = load_rec(D.VR, b.w);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NV; ++j) r[j] = load_rec(D.VR, cv[j]);
  }
}

template <int NV>
__device__ __forceinline__ int slot_of(const uint32_t* sl, int j) {
  return (sl[j >> 2] >> (8 * (j & 3))) & 255;
}

// ---------------------------------------------------------------------------------------------
// EMI rows: A_emi, P_emi (= A + ICS mass) and b_emi in one pass, membrane coupling included.
// ---------------------------------------------------------------------------------------------
// LPR = lanes per row: the pairs of one row are dealt round-robin to LPR adjacent lanes, each lane
// accumulating into its own LDS copy of the block's CSR segment; the copies are summed in a fixed
// order in the epilogue (bit-reproducible, no atomics).  LPR > 1 shortens the per-thread dependent
// gather chain and multiplies the number of waves on small meshes.
template <int GDIM, int NV, int LPR>
__global__ __launch_bounds__(KN_BLOCK) void emi_rows_kernel(KnDev D, const KnConsts* __restrict__ Cp, int lds_n,
                                                            int want_p, int splitting) {
  constexpr int NF = (NV == 8) ? 4 : GDIM;
  constexpr int SW = (NV == 8) ? 2 : 1;
  const KnConsts& C = *Cp;
  extern __shared__ double lds[];
  double* segA = lds;
  double* segP = lds + (size_t)LPR * lds_n;
  HexTab* T = reinterpret_cast<HexTab*>(lds + 2 * (size_t)LPR * lds_n);
  const int tid = threadIdx.x;
  const int b = logical_block(blockIdx.x, D.nblocks);
  const int row0 = D.blk_row0[b], nrows = D.blk_nrows[b], s = D.blk_sub[b];
  const int seg0 = D.rowptr[row0], seglen = D.rowptr[row0 + nrows] - seg0;
  for (int c = 0; c < LPR; ++c)
    for (int i = tid; i < seglen; i += KN_BLOCK) { segA[c * lds_n + i] = 0.0; segP[c * lds_n + i] = 0.0; }
  if constexpr (NV == 8) stage_hex_tables(T);
  __syncthreads();

  const KnSubConst& sc = C.sc[s];
  const bool cell_side = s > 0;
  const int rloc = tid / LPR, sub = tid % LPR;
  const bool valid = rloc < nrows;
  const int g = row0 + (valid ? rloc : 0);
  double bacc = 0.0, gam = 0.0;   // volume part / membrane Robin part of b_emi
  if (valid) {
    double* segA_own = segA;   // membrane terms go to copy 0
    double* segP_own = segP;
    (void)segA_own; (void)segP_own;
    double* segA = lds + (size_t)sub * lds_n;
    double* segP = lds + (size_t)(LPR + sub) * lds_n;
    const int rowbase = D.rowptr[g] - seg0;
    const int lap = rowbase + D.lapoff[g];
    const int w = tid >> 6, lane = tid & 63;
    const int64_t base = D.sl_ptr[(size_t)b * 4 + w];
    const int np = (int)((D.sl_ptr[(size_t)b * 4 + w + 1] - base) >> 6);
    Rec r[NV];
    for (int p = 0; p < np; ++p) {
      const int64_t ent = base + (int64_t)p * KN_SLICE + lane;
      uint32_t sl[SW];
      int li = 0;
      {
        const int pc = D.pair_cell[ent];
        if (pc < 0) continue;
#pragma unroll
        for (int k = 0; k < SW; ++k) sl[k] = D.pair_slots[ent * SW + k];
        li = pc & 7;
        load_cell<NV>(D, pc >> 3, r);
      }
      if constexpr (NV != 8) {
        double d[NV];
        const double vol = simplex_row<GDIM>(r, 0, d);
        double cb0 = 0, cb1 = 0, cb2 = 0, sd = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          cb0 += r[j].c0; cb1 += r[j].c1; cb2 += r[j].c2;
          sd += (sc.sig[0] * r[j].c0 + sc.sig[1] * r[j].c1 + sc.sig[2] * r[j].c2) * d[j];
        }
        const double kbar = (sc.kap[0] * cb0 + sc.kap[1] * cb1 + sc.kap[2] * cb2) * (1.0 / NV);
        const double m = vol * (1.0 / ((GDIM + 1) * (GDIM + 2)));
        bacc -= vol * sd;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const double a = vol * kbar * d[j];
          const int idx = lap + slot_of<NV>(sl, j);
          segA[idx] += a;
          segP[idx] += cell_side ? a + (j == li ? 2.0 * m : m) : a;
        }
      } else {
        double kv[8], sv[8], ra[8], rm[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          kv[j] = sc.kap[0] * r[j].c0 + sc.kap[1] * r[j].c1 + sc.kap[2] * r[j].c2;
          sv[j] = sc.sig[0] * r[j].c0 + sc.sig[1] * r[j].c1 + sc.sig[2] * r[j].c2;
          ra[j] = 0.0; rm[j] = 0.0;
        }
        for (int q = 0; q < 8; ++q) {
          double G[8][3];
          const double wd = hex_point(T, r, q, G);
          double lx = 0, ly = 0, lz = 0, kq = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool me = j == li;
            lx = me ? G[j][0] : lx; ly = me ? G[j][1] : ly; lz = me ? G[j][2] : lz;
            kq += T->N[q][j] * kv[j];
          }
          const double Nl = T->N[q][li];
          double sd = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const double dj = lx * G[j][0] + ly * G[j][1] + lz * G[j][2];
            ra[j] += wd * kq * dj;
            rm[j] += wd * Nl * T->N[q][j];
            sd += sv[j] * dj;
          }
          bacc -= wd * sd;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int idx = lap + slot_of<NV>(sl, j);
          segA[idx] += ra[j];
          segP[idx] += cell_side ? ra[j] + rm[j] : ra[j];
        }
      }
    }
    // membrane coupling C_phi (u_i - u_e)(v_i - v_e) and Robin RHS (emiWeakForm.py:160-165,228-239)
    const int m = sub == 0 ? D.gam_idx[g] : -1;
    if (m >= 0) {
      const int* fown = cell_side ? D.fi : D.fe;
      for (int e = D.mptr[m]; e < D.mptr[m + 1]; ++e) {
        const int ent = D.mentry[e];
        const int fg = ent >> 3, a = ent & 7;
        const int ms = D.fmodel[fg];
        if (ms < 0) continue;
        const uint64_t sl = D.mslots[e];
        Rec p[NF];
#pragma unroll
        for (int bb = 0; bb < NF; ++bb) p[bb] = load_rec(D.VR, fown[(size_t)fg * NF + bb]);
        double Mr[NF];
        facet_mass_row<NF>(p, a, Mr);
        double gs = 0.0;
#pragma unroll
        for (int bb = 0; bb < NF; ++bb) {
          const int q = D.fq[(size_t)fg * NF + bb];
          double gq = D.phiM[q];
          if (!(splitting & 1)) {
            double it = 0.0;
            for (int k = 0; k < KN_MAXK; ++k) it += D.Ich[((size_t)ms * KN_MAXK + k) * D.NQtot + q];
            gq -= it / C.C_phi;
          }
          gs += Mr[bb] * gq;
          const double val = C.C_phi * Mr[bb];
          const int io = rowbase + (int)((sl >> (8 * bb)) & 255);
          const int it2 = rowbase + (int)((sl >> (8 * (4 + bb))) & 255);
          segA[io] += val; segP[io] += val;
          segA[it2] -= val; segP[it2] -= val;
        }
        if (!(splitting & 2)) gam += (cell_side ? 1.0 : -1.0) * C.C_phi * gs;
      }
    }
  }
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) bacc += __shfl_xor(bacc, m);   // fixed-shape tree over the row's lanes
  if (valid && sub == 0) D.b_emi[g] = bacc + gam;
  __syncthreads();
  for (int i = tid; i < seglen; i += KN_BLOCK) {
    double a = segA[i], p = segP[i];
#pragma unroll
    for (int c = 1; c < LPR; ++c) { a += segA[c * lds_n + i]; p += segP[c * lds_n + i]; }
    D.A_emi[seg0 + i] = a;
    if (want_p) D.P_emi[seg0 + i] = p;
  }
}

// ---------------------------------------------------------------------------------------------
// KNP rows: the K-1 diagonal blocks (mass/dt + diffusion + drift) and the volume part of b_knp.
// ---------------------------------------------------------------------------------------------
template <int GDIM, int NV, int LPR>
__global__ __launch_bounds__(KN_BLOCK) void knp_rows_kernel(KnDev D, const KnConsts* __restrict__ Cp, int lds_n) {
  const KnConsts& C = *Cp;
  constexpr int NF = (NV == 8) ? 4 : GDIM;
  constexpr int SW = (NV == 8) ? 2 : 1;
  extern __shared__ double lds[];
  double* seg0k = lds;
  double* seg1k = lds + (size_t)LPR * lds_n;
  HexTab* T = reinterpret_cast<HexTab*>(lds + 2 * (size_t)LPR * lds_n);
  const int tid = threadIdx.x;
  const int b = logical_block(blockIdx.x, D.nblocks);
  const int row0 = D.blk_row0[b], nrows = D.blk_nrows[b], s = D.blk_sub[b];
  const int seg0 = D.rowptrL[row0], seglen = D.rowptrL[row0 + nrows] - seg0;
  for (int c = 0; c < LPR; ++c)
    for (int i = tid; i < seglen; i += KN_BLOCK) { seg0k[c * lds_n + i] = 0.0; seg1k[c * lds_n + i] = 0.0; }
  if constexpr (NV == 8) stage_hex_tables(T);
  __syncthreads();

  const KnSubConst& sc = C.sc[s];
  const int v0 = C.voff[s], nvs = C.voff[s + 1] - v0;
  const double* fs0 = (D.fsrc && s == 0) ? D.fsrc : nullptr;
  const int rloc = tid / LPR, sub = tid % LPR;
  const bool valid = rloc < nrows;
  const int g = row0 + (valid ? rloc : 0);
  double b0 = 0.0, b1 = 0.0;
  if (valid) {
    double* seg0k = lds + (size_t)sub * lds_n;
    double* seg1k = lds + (size_t)(LPR + sub) * lds_n;
    const int lap = D.rowptrL[g] - seg0;
    const int w = tid >> 6, lane = tid & 63;
    const int64_t base = D.sl_ptr[(size_t)b * 4 + w];
    const int np = (int)((D.sl_ptr[(size_t)b * 4 + w + 1] - base) >> 6);
    Rec r[NV];
    for (int p = 0; p < np; ++p) {
      const int64_t ent = base + (int64_t)p * KN_SLICE + lane;
      uint32_t sl[SW];
      int li = 0;
      int cv[NV];
      {
        const int pc = D.pair_cell[ent];
        if (pc < 0) continue;
#pragma unroll
        for (int k = 0; k < SW; ++k) sl[k] = D.pair_slots[ent * SW + k];
        li = pc & 7;
        load_cell<NV>(D, pc >> 3, r);
#pragma unroll
        for (int j = 0; j < NV; ++j) cv[j] = D.cells[(size_t)(pc >> 3) * NV + j];
      }
      double f0[NV], f1[NV];  // (1/dt) c_prev + f_source at the cell vertices
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        f0[j] = r[j].c0 * C.inv_dt;
        f1[j] = r[j].c1 * C.inv_dt;
        if (fs0) { f0[j] += fs0[cv[j]]; f1[j] += fs0[nvs + cv[j]]; }
      }
      if constexpr (NV != 8) {
        double d[NV];
        const double vol = simplex_row<GDIM>(r, 0, d);
        double gp = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) gp += r[j].phi * d[j];
        const double m = vol * (1.0 / ((GDIM + 1) * (GDIM + 2)));
        const double drift = gp * vol * (1.0 / (GDIM + 1));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const double mm = (j == li) ? 2.0 * m : m;
          const int idx = lap + slot_of<NV>(sl, j);
          seg0k[idx] += mm * C.inv_dt + sc.D[0] * vol * d[j] + sc.zpsiD[0] * drift;
          seg1k[idx] += mm * C.inv_dt + sc.D[1] * vol * d[j] + sc.zpsiD[1] * drift;
          b0 += mm * f0[j];
          b1 += mm * f1[j];
        }
      } else {
        double r0[8], r1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { r0[j] = 0.0; r1[j] = 0.0; }
        for (int q = 0; q < 8; ++q) {
          double G[8][3];
          const double wd = hex_point(T, r, q, G);
          double lx = 0, ly = 0, lz = 0, px = 0, py = 0, pz = 0, fq0 = 0, fq1 = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool me = j == li;
            lx = me ? G[j][0] : lx; ly = me ? G[j][1] : ly; lz = me ? G[j][2] : lz;
            px += r[j].phi * G[j][0]; py += r[j].phi * G[j][1]; pz += r[j].phi * G[j][2];
            fq0 += T->N[q][j] * f0[j]; fq1 += T->N[q][j] * f1[j];
          }
          const double Nl = T->N[q][li];
          const double gp = px * lx + py * ly + pz * lz;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const double dj = lx * G[j][0] + ly * G[j][1] + lz * G[j][2];
            const double Nj = T->N[q][j];
            r0[j] += wd * (Nl * Nj * C.inv_dt + sc.D[0] * dj + sc.zpsiD[0] * Nj * gp);
            r1[j] += wd * (Nl * Nj * C.inv_dt + sc.D[1] * dj + sc.zpsiD[1] * Nj * gp);
          }
          b0 += wd * Nl * fq0;
          b1 += wd * Nl * fq1;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int idx = lap + slot_of<NV>(sl, j);
          seg0k[idx] += r0[j];
          seg1k[idx] += r1[j];
        }
      }
    }
    // membrane Robin/coupling contributions, precomputed per (facet, side) by knp_membrane_kernel
    const int m = sub == 0 ? D.gam_idx[g] : -1;
    if (m >= 0) {
      const int side = s > 0 ? 1 : 0;
      for (int e = D.mptr[m]; e < D.mptr[m + 1]; ++e) {
        const int ent = D.mentry[e];
        const int fg = ent >> 3, a = ent & 7;
        if (D.fmodel[fg] < 0) continue;
        const double* cg = D.gam_contrib + ((size_t)(fg * 2 + side) * NF + a) * 2;
        b0 += cg[0];
        b1 += cg[1];
      }
    }
  }
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) { b0 += __shfl_xor(b0, m); b1 += __shfl_xor(b1, m); }
  if (valid && sub == 0) {
    const size_t bb = (size_t)(KN_MAXK - 1) * v0 + (size_t)(g - v0);
    D.b_knp[bb] = b0;
    D.b_knp[bb + nvs] = b1;
  }
  __syncthreads();
  const int64_t subnnz0 = D.rowptrL[v0], subnnz = D.rowptrL[v0 + nvs] - subnnz0;
  double* out0 = D.A_knp + (size_t)(KN_MAXK - 1) * subnnz0 + (seg0 - subnnz0);
  double* out1 = out0 + subnnz;
  for (int i = tid; i < seglen; i += KN_BLOCK) {
    double a0 = seg0k[i], a1 = seg1k[i];
#pragma unroll
    for (int c = 1; c < LPR; ++c) { a0 += seg0k[c * lds_n + i]; a1 += seg1k[c * lds_n + i]; }
    out0[i] = a0;
    out1[i] = a1;
  }
}


// =============================================================================================
// Simplex row kernels, version 2: neighbour records staged in LDS.
//
// The Laplacian row of vertex g lists exactly the vertices of its incident cells, so a (row, cell)
// pair needs nothing but NV row-relative slot bytes.  Phase A gathers, once per row block, the 48-byte
// record of every entry of the block's Laplacian segment into LDS (~15 gathers per row instead of
// 3 per incident cell, i.e. 72 for a Kuhn tetrahedral mesh) and zeroes the accumulators; phase B
// walks the pairs (4 bytes each, coalesced), reads the NV records from LDS and accumulates with
// LDS fp64 adds (ds_add_f64).  All lanes that add into a given row belong to one wavefront, so the
// order of the adds is fixed by program order and lane order: results are bit-reproducible.
// =============================================================================================
#define KN_PREFETCH 8   // pair entries per lane fetched ahead into registers

struct Rec6 {
  double x, y, z, a, b, c;   // EMI: c_prev0, c_prev1, c_elim;  KNP: f0, f1 (= c_prev/dt + f_source), phi
};

__device__ __forceinline__ Rec6 lds_rec(const double* recs, int i) {
  const double2* p = reinterpret_cast<const double2*>(recs + (size_t)i * 6);
  const double2 u = p[0], v = p[1], w = p[2];
  return Rec6{u.x, u.y, v.x, v.y, w.x, w.y};
}

template <int GDIM>
__device__ __forceinline__ double simplex_row0(const Rec6 (&r)[GDIM + 1], double (&d)[GDIM + 1]) {
  // gradient dot products of lambda_0 with all lambda_j, and the cell measure
  if constexpr (GDIM == 2) {
    const double e1x = r[1].x - r[0].x, e1y = r[1].y - r[0].y;
    const double e2x = r[2].x - r[0].x, e2y = r[2].y - r[0].y;
    const double det = e1x * e2y - e1y * e2x, inv = 1.0 / det;
    const double g1x = e2y * inv, g1y = -e2x * inv;
    const double g2x = -e1y * inv, g2y = e1x * inv;
    const double g0x = -(g1x + g2x), g0y = -(g1y + g2y);
    d[0] = g0x * g0x + g0y * g0y;
    d[1] = g0x * g1x + g0y * g1y;
    d[2] = g0x * g2x + g0y * g2y;
    return 0.5 * fabs(det);
  } else {
    const double ax = r[1].x - r[0].x, ay = r[1].y - r[0].y, az = r[1].z - r[0].z;
    const double bx = r[2].x - r[0].x, by = r[2].y - r[0].y, bz = r[2].z - r[0].z;
    const double cx = r[3].x - r[0].x, cy = r[3].y - r[0].y, cz = r[3].z - r[0].z;
    double g1x = by * cz - bz * cy, g1y = bz * cx - bx * cz, g1z = bx * cy - by * cx;
    double g2x = cy * az - cz * ay, g2y = cz * ax - cx * az, g2z = cx * ay - cy * ax;
    double g3x = ay * bz - az * by, g3y = az * bx - ax * bz, g3z = ax * by - ay * bx;
    const double det = ax * g1x + ay * g1y + az * g1z, inv = 1.0 / det;
    g1x *= inv; g1y *= inv; g1z *= inv;
    g2x *= inv; g2y *= inv; g2z *= inv;
    g3x *= inv; g3y *= inv; g3z *= inv;
    const double g0x = -(g1x + g2x + g3x), g0y = -(g1y + g2y + g3y), g0z = -(g1z + g2z + g3z);
    d[0] = g0x * g0x + g0y * g0y + g0z * g0z;
    d[1] = g0x * g1x + g0y * g1y + g0z * g1z;
    d[2] = g0x * g2x + g0y * g2y + g0z * g2z;
    d[3] = g0x * g3x + g0y * g3y + g0z * g3z;
    return fabs(det) * (1.0 / 6.0);
  }
}


// Block descriptor (uniform) and the staging of the block's Laplacian-entry records: all index loads
// are issued together, then all record loads, so a wave waits for memory twice instead of 2 x trips.
struct BlkInfo {
  int row0, nrows, sub, seg0, seglen, segL0, nnzLb, uoff, nuniq, slbase, np;
};

__device__ __forceinline__ BlkInfo load_blk(const KnDev& D, int b, int wave) {
  const int4* p = D.blk_info + 4 * (size_t)b;
  const int4 i0 = p[0], i1 = p[1], i2 = p[2], i3 = p[3];
  BlkInfo B;
  B.row0 = i0.x; B.nrows = i0.y; B.sub = i0.z; B.seg0 = i0.w;
  B.seglen = i1.x; B.segL0 = i1.y; B.nnzLb = i1.z; B.uoff = i1.w;
  B.slbase = wave == 0 ? i2.x : (wave == 1 ? i2.y : (wave == 2 ? i2.z : i2.w));
  B.np = ((unsigned)i3.x >> (8 * wave)) & 255;
  B.nuniq = i3.y;
  return B;
}

#define KN_STAGE 3   // distinct vertices per thread staged with batched loads (256 threads x 3 = 768)

// Phase A of the v2 kernels: the 48-byte records of the block's distinct vertices go to `recs`, the
// 2-byte local index of every Laplacian entry to `eloc`.
template <bool KNP>
__device__ __forceinline__ void stage_records(const KnDev& D, const BlkInfo& B, double* recs, uint16_t* eloc, int tid,
                                              double inv_dt, const double* fs0, int nvs) {
  int vv[KN_STAGE];
#pragma unroll
  for (int k = 0; k < KN_STAGE; ++k) {
    const int i = tid + k * KN_BLOCK;
    vv[k] = i < B.nuniq ? D.blk_uverts[B.uoff + i] : -1;
  }
  for (int i = tid; i < B.nnzLb; i += KN_BLOCK) eloc[i] = D.ent_loc[B.segL0 + i];
  double2 u[KN_STAGE][4];
#pragma unroll
  for (int k = 0; k < KN_STAGE; ++k)
    if (vv[k] >= 0) {
      const double2* src = reinterpret_cast<const double2*>(D.VR + (size_t)vv[k] * KN_REC);
      u[k][0] = src[0]; u[k][1] = src[1]; u[k][2] = src[2]; u[k][3] = src[3];   // x y | z _ | c0 c1 | c2 phi
    }
#pragma unroll
  for (int k = 0; k < KN_STAGE; ++k)
    if (vv[k] >= 0) {
      double2* dst = reinterpret_cast<double2*>(recs + (size_t)(tid + k * KN_BLOCK) * 6);
      if constexpr (KNP) {
        double f0 = u[k][2].x * inv_dt, f1 = u[k][2].y * inv_dt;   // (1/dt) c_prev (+ f_source on the ECS)
        if (fs0) { f0 += fs0[vv[k]]; f1 += fs0[nvs + vv[k]]; }
        dst[0] = u[k][0]; dst[1] = double2{u[k][1].x, f0}; dst[2] = double2{f1, u[k][3].y};
      } else {
        dst[0] = u[k][0]; dst[1] = double2{u[k][1].x, u[k][2].x}; dst[2] = double2{u[k][2].y, u[k][3].x};
      }
    }
  for (int i = tid + KN_STAGE * KN_BLOCK; i < B.nuniq; i += KN_BLOCK) {   // oversized blocks only
    const int v = D.blk_uverts[B.uoff + i];
    const double2* src = reinterpret_cast<const double2*>(D.VR + (size_t)v * KN_REC);
    const double2 u0 = src[0], u1 = src[1], u2 = src[2], u3 = src[3];
    double2* dst = reinterpret_cast<double2*>(recs + (size_t)i * 6);
    if constexpr (KNP) {
      double f0 = u2.x * inv_dt, f1 = u2.y * inv_dt;
      if (fs0) { f0 += fs0[v]; f1 += fs0[nvs + v]; }
      dst[0] = u0; dst[1] = double2{u1.x, f0}; dst[2] = double2{f1, u3.y};
    } else {
      dst[0] = u0; dst[1] = double2{u1.x, u2.x}; dst[2] = double2{u2.y, u3.x};
    }
  }
}

template <int GDIM, int LPR>
__global__ __launch_bounds__(KN_BLOCK) void emi_rows_v2(KnDev D, const KnConsts* __restrict__ Cp, int acc_n,
                                                        int rec_n, int want_p, int splitting) {
  constexpr int NV = GDIM + 1, NF = GDIM;
  const KnConsts& C = *Cp;
  extern __shared__ __align__(16) double lds[];
  double* accA = lds;
  double* accP = lds + acc_n;
  double* recs = lds + 2 * (size_t)acc_n;
  uint16_t* eloc = reinterpret_cast<uint16_t*>(recs + 6 * (size_t)rec_n);
  const int tid = threadIdx.x;
  const int b = logical_block(blockIdx.x, D.nblocks);
  const BlkInfo B = load_blk(D, b, tid >> 6);
  const int row0 = B.row0, nrows = B.nrows, s = B.sub, seg0 = B.seg0, seglen = B.seglen;
  // this lane's pair entries and row descriptor: issued first, their latency overlaps phase A
  const int rloc = tid / LPR, sub = tid % LPR;
  const bool valid = rloc < nrows;
  const int g = row0 + (valid ? rloc : 0);
  const int64_t base = (int64_t)B.slbase * KN_SLICE + (tid & 63);
  const int np = B.np;
  uint32_t slr[KN_PREFETCH];
#pragma unroll
  for (int p = 0; p < KN_PREFETCH; ++p)
    slr[p] = (valid && p < np) ? D.pair_sl[base + (int64_t)p * KN_SLICE] : 0xFFFFFFFFu;
  const int4 ri = D.row_info[g];
  // phase A: zero accumulators, stage the records of the block's Laplacian entries
  for (int i = tid; i < seglen; i += KN_BLOCK) { accA[i] = 0.0; accP[i] = 0.0; }
  stage_records<false>(D, B, recs, eloc, tid, 0.0, nullptr, 0);
  __syncthreads();

  const KnSubConst& sc = C.sc[s];
  const bool cell_side = s > 0;
  double bacc = 0.0, gam = 0.0;   // volume part / membrane Robin part of b_emi
  if (valid) {
    const int rowbase = ri.x, lap = ri.y, rL = ri.z;
    // The diagonal entry receives a term from every pair: keep it in registers and add it once.
    // P differs from A only on cell-side rows (ICS mass), so ECS rows accumulate A alone.
    const bool acc_p = want_p && cell_side;
    int diag = -1;
    double dA = 0.0, dP = 0.0;
    Rec6 r[NV];
    auto do_pair = [&](uint32_t sl) {
      int slot[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) slot[j] = (sl >> (8 * j)) & 255;
      if (diag < 0) { diag = slot[0]; r[0] = lds_rec(recs, eloc[rL + diag]); }   // the row's own vertex, once
#pragma unroll
      for (int j = 1; j < NV; ++j) r[j] = lds_rec(recs, eloc[rL + slot[j]]);
      double d[NV];
      const double vol = simplex_row0<GDIM>(r, d);
      double cb0 = 0, cb1 = 0, cb2 = 0, sd = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        cb0 += r[j].a; cb1 += r[j].b; cb2 += r[j].c;
        sd += (sc.sig[0] * r[j].a + sc.sig[1] * r[j].b + sc.sig[2] * r[j].c) * d[j];
      }
      const double kbar = (sc.kap[0] * cb0 + sc.kap[1] * cb1 + sc.kap[2] * cb2) * (1.0 / NV);
      const double m = vol * (1.0 / ((GDIM + 1) * (GDIM + 2)));
      bacc -= vol * sd;
      dA += vol * kbar * d[0];
      dP += vol * kbar * d[0] + 2.0 * m;
#pragma unroll
      for (int j = 1; j < NV; ++j) {
        const double a = vol * kbar * d[j];
        unsafeAtomicAdd(&accA[lap + slot[j]], a);
        if (acc_p) unsafeAtomicAdd(&accP[lap + slot[j]], a + m);
      }
    };
#pragma unroll
    for (int p = 0; p < KN_PREFETCH; ++p)
      if (slr[p] != 0xFFFFFFFFu) do_pair(slr[p]);
    for (int p = KN_PREFETCH; p < np; ++p) {
      const uint32_t sl = D.pair_sl[base + (int64_t)p * KN_SLICE];
      if (sl != 0xFFFFFFFFu) do_pair(sl);
    }
    if (diag >= 0) {
      unsafeAtomicAdd(&accA[lap + diag], dA);
      if (acc_p) unsafeAtomicAdd(&accP[lap + diag], dP);
    }
    // membrane coupling C_phi (u_i - u_e)(v_i - v_e) and Robin RHS (emiWeakForm.py:160-165,228-239)
    const int m = sub == 0 ? ri.w : -1;
    if (m >= 0) {
      const int* fown = cell_side ? D.fi : D.fe;
      for (int e = D.mptr[m]; e < D.mptr[m + 1]; ++e) {
        const int ent = D.mentry[e];
        const int fg = ent >> 3, a = ent & 7;
        const int ms = D.fmodel[fg];
        if (ms < 0) continue;
        const uint64_t sl = D.mslots[e];
        Rec p[NF];
#pragma unroll
        for (int bb = 0; bb < NF; ++bb) p[bb] = load_rec(D.VR, fown[(size_t)fg * NF + bb]);
        double Mr[NF];
        facet_mass_row<NF>(p, a, Mr);
        double gs = 0.0;
#pragma unroll
        for (int bb = 0; bb < NF; ++bb) {
          const int q = D.fq[(size_t)fg * NF + bb];
          double gq = D.phiM[q];
          if (!(splitting & 1)) {
            double it = 0.0;
            for (int k = 0; k < KN_MAXK; ++k) it += D.Ich[((size_t)ms * KN_MAXK + k) * D.NQtot + q];
            gq -= it / C.C_phi;
          }
          gs += Mr[bb] * gq;
          const double val = C.C_phi * Mr[bb];
          const int io = rowbase + (int)((sl >> (8 * bb)) & 255);
          const int it2 = rowbase + (int)((sl >> (8 * (4 + bb))) & 255);
          unsafeAtomicAdd(&accA[io], val);
          unsafeAtomicAdd(&accA[it2], -val);
          if (acc_p) { unsafeAtomicAdd(&accP[io], val); unsafeAtomicAdd(&accP[it2], -val); }
        }
        if (!(splitting & 2)) gam += (cell_side ? 1.0 : -1.0) * C.C_phi * gs;
      }
    }
  }
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) bacc += __shfl_xor(bacc, m);
  if (valid && sub == 0) D.b_emi[g] = bacc + gam;
  __syncthreads();
  for (int i = tid; i < seglen; i += KN_BLOCK) {
    const double a = accA[i];
    D.A_emi[seg0 + i] = a;
    if (want_p) D.P_emi[seg0 + i] = cell_side ? accP[i] : a;
  }
}

template <int GDIM, int LPR>
__global__ __launch_bounds__(KN_BLOCK) void knp_rows_v2(KnDev D, const KnConsts* __restrict__ Cp, int acc_n, int rec_n) {
  constexpr int NV = GDIM + 1, NF = GDIM;
  const KnConsts& C = *Cp;
  extern __shared__ __align__(16) double lds[];
  double* acc0 = lds;
  double* acc1 = lds + acc_n;
  double* recs = lds + 2 * (size_t)acc_n;
  uint16_t* eloc = reinterpret_cast<uint16_t*>(recs + 6 * (size_t)rec_n);
  const int tid = threadIdx.x;
  const int b = logical_block(blockIdx.x, D.nblocks);
  const BlkInfo B = load_blk(D, b, tid >> 6);
  const int row0 = B.row0, nrows = B.nrows, s = B.sub, segL0 = B.segL0, nnzLb = B.nnzLb;
  const int v0 = C.voff[s], nvs = C.voff[s + 1] - v0;
  const double* fs0 = (D.fsrc && s == 0) ? D.fsrc : nullptr;
  const int rloc = tid / LPR, sub = tid % LPR;
  const bool valid = rloc < nrows;
  const int g = row0 + (valid ? rloc : 0);
  const int64_t base = (int64_t)B.slbase * KN_SLICE + (tid & 63);
  const int np = B.np;
  uint32_t slr[KN_PREFETCH];
#pragma unroll
  for (int p = 0; p < KN_PREFETCH; ++p)
    slr[p] = (valid && p < np) ? D.pair_sl[base + (int64_t)p * KN_SLICE] : 0xFFFFFFFFu;
  const int4 ri = D.row_info[g];
  for (int i = tid; i < nnzLb; i += KN_BLOCK) { acc0[i] = 0.0; acc1[i] = 0.0; }
  stage_records<true>(D, B, recs, eloc, tid, C.inv_dt, fs0, nvs);
  __syncthreads();

  const KnSubConst& sc = C.sc[s];
  double b0 = 0.0, b1 = 0.0;
  if (valid) {
    const int rL = ri.z;
    int diag = -1;
    double d0 = 0.0, d1 = 0.0;   // diagonal entries of the two ion blocks, added once after the loop
    Rec6 r[NV];
    auto do_pair = [&](uint32_t sl) {
      int slot[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) slot[j] = (sl >> (8 * j)) & 255;
      if (diag < 0) { diag = slot[0]; r[0] = lds_rec(recs, eloc[rL + diag]); }
#pragma unroll
      for (int j = 1; j < NV; ++j) r[j] = lds_rec(recs, eloc[rL + slot[j]]);
      double d[NV];
      const double vol = simplex_row0<GDIM>(r, d);
      double gp = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) gp += r[j].c * d[j];
      const double m = vol * (1.0 / ((GDIM + 1) * (GDIM + 2)));
      const double drift = gp * vol * (1.0 / (GDIM + 1));
      d0 += 2.0 * m * C.inv_dt + sc.D[0] * vol * d[0] + sc.zpsiD[0] * drift;
      d1 += 2.0 * m * C.inv_dt + sc.D[1] * vol * d[0] + sc.zpsiD[1] * drift;
      b0 += 2.0 * m * r[0].a;
      b1 += 2.0 * m * r[0].b;
#pragma unroll
      for (int j = 1; j < NV; ++j) {
        unsafeAtomicAdd(&acc0[rL + slot[j]], m * C.inv_dt + sc.D[0] * vol * d[j] + sc.zpsiD[0] * drift);
        unsafeAtomicAdd(&acc1[rL + slot[j]], m * C.inv_dt + sc.D[1] * vol * d[j] + sc.zpsiD[1] * drift);
        b0 += m * r[j].a;
        b1 += m * r[j].b;
      }
    };
#pragma unroll
    for (int p = 0; p < KN_PREFETCH; ++p)
      if (slr[p] != 0xFFFFFFFFu) do_pair(slr[p]);
    for (int p = KN_PREFETCH; p < np; ++p) {
      const uint32_t sl = D.pair_sl[base + (int64_t)p * KN_SLICE];
      if (sl != 0xFFFFFFFFu) do_pair(sl);
    }
    if (diag >= 0) {
      unsafeAtomicAdd(&acc0[rL + diag], d0);
      unsafeAtomicAdd(&acc1[rL + diag], d1);
    }
    // membrane Robin/coupling contributions, precomputed per (facet, side) by knp_membrane_kernel
    const int m = sub == 0 ? ri.w : -1;
    if (m >= 0) {
      const int side = s > 0 ? 1 : 0;
      for (int e = D.mptr[m]; e < D.mptr[m + 1]; ++e) {
        const int ent = D.mentry[e];
        const int fg = ent >> 3, a = ent & 7;
        if (D.fmodel[fg] < 0) continue;
        const double* cg = D.gam_contrib + ((size_t)(fg * 2 + side) * NF + a) * 2;
        b0 += cg[0];
        b1 += cg[1];
      }
    }
  }
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) { b0 += __shfl_xor(b0, m); b1 += __shfl_xor(b1, m); }
  if (valid && sub == 0) {
    const size_t bb = (size_t)(KN_MAXK - 1) * v0 + (size_t)(g - v0);
    D.b_knp[bb] = b0;
    D.b_knp[bb + nvs] = b1;
  }
  __syncthreads();
  const int64_t subnnz0 = D.rowptrL[v0], subnnz = D.rowptrL[v0 + nvs] - subnnz0;
  double* out0 = D.A_knp + (size_t)(KN_MAXK - 1) * subnnz0 + (segL0 - subnnz0);
  double* out1 = out0 + subnnz;
  for (int i = tid; i < nnzLb; i += KN_BLOCK) { out0[i] = acc0[i]; out1[i] = acc1[i]; }
}

// ---------------------------------------------------------------------------------------------
// KNP membrane-facet kernel: the rational Robin/coupling integrand of knpWeakForm.py:168-214 with
// degree-6 quadrature (tables staged in LDS).  One thread per (membrane facet, side): it evaluates the
// integrand once per quadrature point and tests it against all NF facet basis functions, writing
// NF x 2 (ions) partial integrals to `gam_contrib`; the row kernel adds them into b_knp in a fixed
// order, so nothing is accumulated atomically.
// ---------------------------------------------------------------------------------------------
template <int NF>
__global__ __launch_bounds__(256) void knp_membrane_kernel(KnDev D, const KnConsts* __restrict__ Cp, int splitting) {
  const KnConsts& C = *Cp;
  extern __shared__ double qt[];
  const int nq = D.nq_gamma;
  const int ntab = nq * (1 + NF + (NF == 4 ? 2 * NF : 0));
  for (int i = threadIdx.x; i < ntab; i += blockDim.x) qt[i] = D.qtab[i];
  __syncthreads();
  const double* qw = qt;
  const double* qN = qt + nq;
  const double* qdN = qt + nq * (1 + NF);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * D.nftot) return;
  const int fg = t >> 1;
  const bool cell_side = t & 1;
  const int ms = D.fmodel[fg];
  double* out = D.gam_contrib + (size_t)t * NF * 2;
  if (ms < 0) {
#pragma unroll
    for (int i = 0; i < NF * 2; ++i) out[i] = 0.0;
    return;
  }
  Rec pe[NF], pi[NF];
  double pm[NF], I0[NF], I1[NF], It[NF];
  int si = 0;  // sub-domain of the cell side of this facet
#pragma unroll
  for (int bb = 0; bb < NF; ++bb) {
    const int vi = D.fi[(size_t)fg * NF + bb];
    pe[bb] = load_rec(D.VR, D.fe[(size_t)fg * NF + bb]);
    pi[bb] = load_rec(D.VR, vi);
    const int q = D.fq[(size_t)fg * NF + bb];
    pm[bb] = D.phiM[q];
    const double* ich = D.Ich + (size_t)ms * KN_MAXK * D.NQtot + q;
    I0[bb] = ich[0];
    I1[bb] = ich[D.NQtot];
    It[bb] = I0[bb] + I1[bb] + ich[2 * (size_t)D.NQtot];
    if (bb == 0) for (int tt = 1; tt < C.n_sub; ++tt) si += vi >= C.voff[tt];
  }
  const KnSubConst& so = C.sc[cell_side ? si : 0];   // own-side constants
  double meas = 0.0;
  if constexpr (NF != 4) meas = facet_measure<NF>(pe);
  double acc0[NF], acc1[NF];
#pragma unroll
  for (int a = 0; a < NF; ++a) { acc0[a] = 0.0; acc1[a] = 0.0; }
  const double sgn = cell_side ? 1.0 : -1.0;
  for (int q = 0; q < nq; ++q) {
    double c0 = 0, c1 = 0, c2 = 0, ph_e = 0, ph_i = 0, pmq = 0, i0 = 0, i1 = 0, it = 0;
#pragma unroll
    for (int bb = 0; bb < NF; ++bb) {
      const double N = qN[q * NF + bb];
      const Rec& o = cell_side ? pi[bb] : pe[bb];
      c0 += N * o.c0; c1 += N * o.c1; c2 += N * o.c2;
      ph_e += N * pe[bb].phi; ph_i += N * pi[bb].phi;
      pmq += N * pm[bb]; i0 += N * I0[bb]; i1 += N * I1[bb]; it += N * It[bb];
    }
    double wq;
    if constexpr (NF == 4) {
      // surface Jacobian of the bilinear facet at this point
      double ux = 0, uy = 0, uz = 0, vx = 0, vy = 0, vz = 0;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const double da = qdN[(q * 4 + bb) * 2], db = qdN[(q * 4 + bb) * 2 + 1];
        ux += da * pe[bb].x; uy += da * pe[bb].y; uz += da * pe[bb].z;
        vx += db * pe[bb].x; vy += db * pe[bb].y; vz += db * pe[bb].z;
      }
      const double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
      wq = qw[q] * sqrt(nx * nx + ny * ny + nz * nz);
    } else {
      wq = qw[q] * meas * (NF == 2 ? 1.0 : 2.0);  // reference measure 1 (interval), 1/2 (triangle)
    }
    const double asum = so.az2D[0] * c0 + so.az2D[1] * c1 + so.az2D[2] * c2;
    const double jump = ph_i - ph_e;
    double f0, f1;
    {
      const double al = so.az2D[0] * c0 / asum;
      const double Cc = al * C.C_M / (C.F * C.z[0] * C.dt);
      double gr = pmq - C.dt / (C.C_M * al) * i0;
      if (splitting) gr += (C.dt / C.C_M) * it;
      f0 = wq * sgn * (Cc * gr - Cc * jump);
    }
    {
      const double al = so.az2D[1] * c1 / asum;
      const double Cc = al * C.C_M / (C.F * C.z[1] * C.dt);
      double gr = pmq - C.dt / (C.C_M * al) * i1;
      if (splitting) gr += (C.dt / C.C_M) * it;
      f1 = wq * sgn * (Cc * gr - Cc * jump);
    }
#pragma unroll
    for (int a = 0; a < NF; ++a) {
      const double Na = qN[q * NF + a];
      acc0[a] += Na * f0;
      acc1[a] += Na * f1;
    }
  }
#pragma unroll
  for (int a = 0; a < NF; ++a) { out[2 * a] = acc0[a]; out[2 * a + 1] = acc1[a]; }
}

// ---------------------------------------------------------------------------------------------
// small data-movement kernels
// ---------------------------------------------------------------------------------------------
__global__ void scatter_kernel(const double* __restrict__ src, double* __restrict__ dst, int n, int stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(size_t)i * stride] = src[i];
}

__global__ void gather_kernel(const double* __restrict__ src, int stride, double* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[(size_t)i * stride];
}

// interpolate_to_membrane (utils.py:150-207): for CG-1 on matching meshes the trace is a gather.
// ue / ui are sub-mesh-local arrays, q2e / q2i hold global vertex ids (offsets v0e = 0, v0i).
__global__ void trace_kernel(const double* __restrict__ ue, const double* __restrict__ ui,
                             const int* __restrict__ q2e, const int* __restrict__ q2i, int q0, int nq,
                             int v0i, double* __restrict__ qe, double* __restrict__ qi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) {
    qe[i] = ue[q2e[q0 + i]];
    qi[i] = ui[q2i[q0 + i] - v0i];
  }
}

// update_pde_variables (utils.py:238-295): c_prev <- c, eliminated ion from electroneutrality,
// phi_M_prev <- tr(phi_i) - tr(phi_e).
__global__ void update_pde_kernel(KnDev D, const KnConsts* __restrict__ Cp) {
  const KnConsts& C = *Cp;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.Ntot) {
    int s = 0;
    for (int t = 1; t < C.n_sub; ++t) s += i >= C.voff[t];
    const double c0 = D.csol[i], c1 = D.csol[(size_t)D.Ntot + i];
    double el = C.sc[s].rho_term;
    el += C.elim_coef[0] * c0;
    el += C.elim_coef[1] * c1;
    double* rec = D.VR + (size_t)i * KN_REC + 4;  // the phi component is left untouched
    rec[0] = c0; rec[1] = c1; rec[2] = el;
  }
  if (i < D.NQtot) D.phiM[i] = D.VR[(size_t)D.q2i[i] * KN_REC + 7] - D.VR[(size_t)D.q2e[i] * KN_REC + 7];
}

// Forward-halo pack / unpack (owner -> ghost copies of dof fields between GPUs).
__global__ void halo_kernel(KnDev D, int kind, int pack, const int* __restrict__ idx, int n, int n_slots,
                            double* __restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = idx[i];
  if (kind == 0) {
    double* rec = D.VR + (size_t)g * KN_REC + 4;
    double* b = buf + (size_t)i * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (pack) b[c] = rec[c];
      else rec[c] = b[c];
    }
  } else {
    const int w = 1 + KN_MAXK * n_slots;
    double* b = buf + (size_t)i * w;
    if (pack) b[0] = D.phiM[g];
    else D.phiM[g] = b[0];
    for (int c = 0; c < KN_MAXK * n_slots; ++c) {
      double* f = D.Ich + (size_t)c * D.NQtot + g;
      if (pack) b[1 + c] = *f;
      else *f = b[1 + c];
    }
  }
}

// Membrane Robin term of b_emi alone (emiWeakForm.py:228-239), for the runs that assemble the EMI
// matrix beside the ODE sweep: one thread per membrane row, same arithmetic and entry order as the
// row kernels, so b_emi is bit-identical to the fused path.
template <int NF>
__global__ __launch_bounds__(256) void emi_membrane_rhs_kernel(KnDev D, const KnConsts* __restrict__ Cp, int splitting) {
  const KnConsts& C = *Cp;
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= D.M) return;
  const int g = D.mrow[m];
  const bool cell_side = g >= C.voff[1];
  const int* fown = cell_side ? D.fi : D.fe;
  double gam = 0.0;
  for (int e = D.mptr[m]; e < D.mptr[m + 1]; ++e) {
    const int ent = D.mentry[e];
    const int fg = ent >> 3, a = ent & 7;
    const int ms = D.fmodel[fg];
    if (ms < 0) continue;
    Rec p[NF];
#pragma unroll
    for (int bb = 0; bb < NF; ++bb) p[bb] = load_rec(D.VR, fown[(size_t)fg * NF + bb]);
    double Mr[NF];
    facet_mass_row<NF>(p, a, Mr);
    double gs = 0.0;
#pragma unroll
    for (int bb = 0; bb < NF; ++bb) {
      const int q = D.fq[(size_t)fg * NF + bb];
      double gq = D.phiM[q];
      if (!splitting) {
        double it = 0.0;
        for (int k = 0; k < KN_MAXK; ++k) it += D.Ich[((size_t)ms * KN_MAXK + k) * D.NQtot + q];
        gq -= it / C.C_phi;
      }
      gs += Mr[bb] * gq;
    }
    gam += (cell_side ? 1.0 : -1.0) * C.C_phi * gs;
  }
  D.b_emi[g] = D.b_emi[g] + gam;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    kn_set_error(std::string(what) + ": " + hipGetErrorString(e));
    return KNPEMI_EHIP;
  }
  return KNPEMI_OK;
}

template <class K>
int set_lds_limit(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      kn_set_error(std::string("hipFuncSetAttribute(max dynamic LDS): ") + hipGetErrorString(e));
      return KNPEMI_EHIP;
    }
  }
  return KNPEMI_OK;
}

}  // namespace

template <int GDIM, int NV>
static int launch_emi(knpemi_handle* h, size_t lds, int lds_n, int want_p, int split) {
  const KnDev& D = h->dev;
  dim3 grid(D.nblocks), block(KN_BLOCK);
  int rc = 0;
#define KN_CASE(L)                                                                                   \
  case L:                                                                                            \
    if ((rc = set_lds_limit(emi_rows_kernel<GDIM, NV, L>, lds))) return rc;                          \
    {                                                                                                \
      KnProfScope prof(h, KNPEMI_K_EMI_ROWS);                                                        \
      hipLaunchKernelGGL((emi_rows_kernel<GDIM, NV, L>), grid, block, lds, h->cur, D, h->d_consts, \
                         lds_n, want_p, split);                                                      \
    }                                                                                                \
    break;
  switch (h->lpr) { KN_CASE(1) KN_CASE(2) KN_CASE(4) KN_CASE(8) default: kn_set_error("bad lanes-per-row"); return KNPEMI_EINVAL; }
#undef KN_CASE
  return check_launch("emi_rows_kernel");
}

template <int GDIM, int NV>
static int launch_knp(knpemi_handle* h, size_t lds, int lds_n) {
  const KnDev& D = h->dev;
  dim3 grid(D.nblocks), block(KN_BLOCK);
  int rc = 0;
#define KN_CASE(L)                                                                                   \
  case L:                                                                                            \
    if ((rc = set_lds_limit(knp_rows_kernel<GDIM, NV, L>, lds))) return rc;                          \
    {                                                                                                \
      KnProfScope prof(h, KNPEMI_K_KNP_ROWS);                                                        \
      hipLaunchKernelGGL((knp_rows_kernel<GDIM, NV, L>), grid, block, lds, h->cur, D, h->d_consts, \
                         lds_n);                                                                     \
    }                                                                                                \
    break;
  switch (h->lpr) { KN_CASE(1) KN_CASE(2) KN_CASE(4) KN_CASE(8) default: kn_set_error("bad lanes-per-row"); return KNPEMI_EINVAL; }
#undef KN_CASE
  return check_launch("knp_rows_kernel");
}

template <int GDIM>
static int launch_emi_v2(knpemi_handle* h, int want_p, int split) {
  const KnDev& D = h->dev;
  const int acc_n = (h->lds_doubles_emi + 1) & ~1, rec_n = h->lds_uniq_max;
  const size_t lds = ((size_t)2 * acc_n + 6 * (size_t)rec_n) * sizeof(double) + 2 * (size_t)h->lds_doubles_knp + 16;
  if (lds > 160 * 1024) { kn_set_error("EMI row block does not fit in 160 KiB of LDS"); return KNPEMI_EINVAL; }
  dim3 grid(D.nblocks), block(KN_BLOCK);
  int rc = 0;
#define KN_CASE(L)                                                                                  \
  case L:                                                                                           \
    if ((rc = set_lds_limit(emi_rows_v2<GDIM, L>, lds))) return rc;                                 \
    {                                                                                               \
      KnProfScope prof(h, KNPEMI_K_EMI_ROWS);                                                       \
      hipLaunchKernelGGL((emi_rows_v2<GDIM, L>), grid, block, lds, h->cur, D, h->d_consts, acc_n, \
                         rec_n, want_p, split);                                                            \
    }                                                                                               \
    break;
  switch (h->lpr) { KN_CASE(1) KN_CASE(2) KN_CASE(4) KN_CASE(8) default: kn_set_error("bad lanes-per-row"); return KNPEMI_EINVAL; }
#undef KN_CASE
  return check_launch("emi_rows_v2");
}

template <int GDIM>
static int launch_knp_v2(knpemi_handle* h) {
  const KnDev& D = h->dev;
  const int acc_n = (h->lds_doubles_knp + 1) & ~1, rec_n = h->lds_uniq_max;
  const size_t lds = ((size_t)2 * acc_n + 6 * (size_t)rec_n) * sizeof(double) + 2 * (size_t)h->lds_doubles_knp + 16;
  if (lds > 160 * 1024) { kn_set_error("KNP row block does not fit in 160 KiB of LDS"); return KNPEMI_EINVAL; }
  dim3 grid(D.nblocks), block(KN_BLOCK);
  int rc = 0;
#define KN_CASE(L)                                                                                  \
  case L:                                                                                           \
    if ((rc = set_lds_limit(knp_rows_v2<GDIM, L>, lds))) return rc;                                 \
    {                                                                                               \
      KnProfScope prof(h, KNPEMI_K_KNP_ROWS);                                                       \
      hipLaunchKernelGGL((knp_rows_v2<GDIM, L>), grid, block, lds, h->cur, D, h->d_consts, acc_n, rec_n); \
    }                                                                                               \
    break;
  switch (h->lpr) { KN_CASE(1) KN_CASE(2) KN_CASE(4) KN_CASE(8) default: kn_set_error("bad lanes-per-row"); return KNPEMI_EINVAL; }
#undef KN_CASE
  return check_launch("knp_rows_v2");
}

int kn_launch_emi_rows(knpemi_handle* h, int flags) {
  const KnDev& D = h->dev;
  if (D.nblocks == 0) return KNPEMI_OK;
  if (h->NV != 8) {
    const int want_p = (flags & KNPEMI_WANT_P) ? 1 : 0;
    const int split = ((flags & KNPEMI_NO_SPLITTING) ? 0 : 1) | ((flags & KNPEMI_SKIP_MEMBRANE_RHS) ? 2 : 0);
    return h->gdim == 2 ? launch_emi_v2<2>(h, want_p, split) : launch_emi_v2<3>(h, want_p, split);
  }
  const int lds_n = h->lds_doubles_emi;
  const size_t lds = (size_t)2 * h->lpr * lds_n * sizeof(double) + (h->NV == 8 ? sizeof(HexTab) : 0);
  if (lds > 160 * 1024) { kn_set_error("EMI row block does not fit in 160 KiB of LDS"); return KNPEMI_EINVAL; }
  const int want_p = (flags & KNPEMI_WANT_P) ? 1 : 0;
    const int split = ((flags & KNPEMI_NO_SPLITTING) ? 0 : 1) | ((flags & KNPEMI_SKIP_MEMBRANE_RHS) ? 2 : 0);
  return launch_emi<3, 8>(h, lds, lds_n, want_p, split);
}

int kn_launch_knp_rows(knpemi_handle* h, int flags) {
  (void)flags;
  const KnDev& D = h->dev;
  if (D.nblocks == 0) return KNPEMI_OK;
  if (h->NV != 8) return h->gdim == 2 ? launch_knp_v2<2>(h) : launch_knp_v2<3>(h);
  const int lds_n = h->lds_doubles_knp;
  const size_t lds = (size_t)2 * h->lpr * lds_n * sizeof(double) + (h->NV == 8 ? sizeof(HexTab) : 0);
  if (lds > 160 * 1024) { kn_set_error("KNP row block does not fit in 160 KiB of LDS"); return KNPEMI_EINVAL; }
  return launch_knp<3, 8>(h, lds, lds_n);
}

int kn_launch_knp_membrane(knpemi_handle* h, int flags) {
  const KnDev& D = h->dev;
  if (D.nftot == 0) return KNPEMI_OK;
  const int split = (flags & KNPEMI_NO_SPLITTING) ? 0 : 1;
  const int NF = h->NF;
  const size_t lds = (size_t)D.nq_gamma * (1 + NF + (NF == 4 ? 2 * NF : 0)) * sizeof(double);
  dim3 grid((2 * D.nftot + 255) / 256), block(256);
  KnProfScope prof(h, KNPEMI_K_KNP_MEMBRANE);
  if (NF == 2) hipLaunchKernelGGL((knp_membrane_kernel<2>), grid, block, lds, h->cur, D, h->d_consts, split);
  else if (NF == 3) hipLaunchKernelGGL((knp_membrane_kernel<3>), grid, block, lds, h->cur, D, h->d_consts, split);
  else hipLaunchKernelGGL((knp_membrane_kernel<4>), grid, block, lds, h->cur, D, h->d_consts, split);
  return check_launch("knp_membrane_kernel");
}

int kn_launch_emi_membrane_rhs(knpemi_handle* h, int flags) {
  const KnDev& D = h->dev;
  if (D.M == 0) return KNPEMI_OK;
  const int split = (flags & KNPEMI_NO_SPLITTING) ? 0 : 1;
  dim3 grid((D.M + 255) / 256), block(256);
  if (h->NF == 2) hipLaunchKernelGGL((emi_membrane_rhs_kernel<2>), grid, block, 0, h->stream, D, h->d_consts, split);
  else if (h->NF == 3) hipLaunchKernelGGL((emi_membrane_rhs_kernel<3>), grid, block, 0, h->stream, D, h->d_consts, split);
  else hipLaunchKernelGGL((emi_membrane_rhs_kernel<4>), grid, block, 0, h->stream, D, h->d_consts, split);
  return check_launch("emi_membrane_rhs_kernel");
}

int kn_launch_update_pde(knpemi_handle* h) {
  const KnDev& D = h->dev;
  const int n = std::max(D.Ntot, D.NQtot);
  if (n == 0) return KNPEMI_OK;
  KnProfScope prof(h, KNPEMI_K_UPDATE);
  hipLaunchKernelGGL(update_pde_kernel, dim3((n + 255) / 256), dim3(256), 0, h->cur, D, h->d_consts);
  return check_launch("update_pde_kernel");
}

int kn_launch_halo(knpemi_handle* h, int kind, int pack, const int32_t* idx, int n, double* buf) {
  if (n == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(halo_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->dev, kind, pack, idx, n,
                     h->moff[h->n_sub], buf);
  return check_launch("halo_kernel");
}

int kn_launch_field_scatter(knpemi_handle* h, const double* src, double* dst, int n, int dst_stride) {
  if (n == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, src, dst, n, dst_stride);
  return check_launch("scatter_kernel");
}

int kn_launch_field_gather(knpemi_handle* h, const double* src, int src_stride, double* dst, int n) {
  if (n == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, src, src_stride, dst, n);
  return check_launch("gather_kernel");
}

int kn_launch_trace(knpemi_handle* h, const double* ue, const double* ui, int sub, double* qe, double* qi) {
  const int nq = h->n_q[sub];
  if (nq == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(trace_kernel, dim3((nq + 255) / 256), dim3(256), 0, h->stream, ue, ui,
                     h->dev.q2e, h->dev.q2i, h->qoff[sub], nq, h->voff[sub], qe, qi);
  return check_launch("trace_kernel");
}
