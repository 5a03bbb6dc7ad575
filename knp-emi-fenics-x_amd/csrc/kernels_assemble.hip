// gfx950 assembly kernels of the knpemi hot path.
//
// Design: "owner-computes rows".  A block of 256 / LPR consecutive matrix rows
// (= sub-mesh vertices) owns a contiguous CSR range.  It stages the 48-byte records of the distinct
// vertices its rows touch in LDS once, then LPR lanes per row walk the cells incident to the row's
// vertex (sliced-ELL list, coalesced index reads), recompute the closed-form P1 element row (or the
// Gauss-quadrature Q1 row) and accumulate into the block's slice of the CSR value array held in LDS.
// The final write to HBM is a plain coalesced stream: no global atomics, bit-reproducible results,
// and the matrix, the preconditioner and the RHS come out of one pass over the mesh.
//
// Forms restated (paths relative to the reference repository):
//   EMI  a, p, L : src/knpemi/emiWeakForm.py:138-241
//   KNP  a, L    : src/knpemi/knpWeakForm.py:123-216
//   update       : src/knpemi/utils.py:238-295
#include "knpemi_internal.h"

// The operators leave the row kernels as streaming stores: they are read next by another kernel and would only push the vertex
// records, which neighbouring row blocks re-read, out of the L2 (round 4, A/B on one box: 2-4 % per row kernel at 995 k tets and
// on the 166 k-hexahedron mesh).
#define KN_ROW_STORE(v, p) __builtin_nontemporal_store((v), (p))
// (the pair entries are also read once per kernel, but loading them non-temporally costs 3-10 %: measured, not kept)

namespace {

__device__ __forceinline__ int logical_block(int bid, int nb) {
  // Workgroups are dealt round-robin over the 8 XCDs; give each XCD one contiguous chunk of row
  // blocks so neighbouring rows (which share vertices and cells) meet in the same L2.
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

struct Rec {
  double x, y, z, c[KN_MAXK], phi;   // c[k]: ion k (record slot KN_CSLOT(k))
};

__device__ __forceinline__ Rec load_rec(const double* __restrict__ VR, int v) {
  const double4* p = reinterpret_cast<const double4*>(VR) + 2 * (size_t)v;
  const double4 a = p[0], b = p[1];
  Rec r;
  r.x = a.x; r.y = a.y; r.z = a.z; r.c[0] = b.x; r.c[1] = b.y; r.c[2] = b.z; r.c[3] = a.w; r.phi = b.w;
  return r;
}

// Gradient dot products d[j] = grad(lambda_li) . grad(lambda_j) and the cell measure of a P1
// simplex (closed form; FFCx reaches the same numbers with a 1-point rule).
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

// 1 / a from the hardware estimate and two Newton steps (relative error of a few 1e-16): five instructions instead of the
// eleven of the IEEE division sequence, once per (row, cell) pair and per quadrature point of the Q1 kernels.
__device__ __forceinline__ double kn_rcp(double a) {
  double r = __builtin_amdgcn_rcp(a);
  r = fma(fma(-a, r, 1.0), r, r);
  return fma(fma(-a, r, 1.0), r, r);
}

// ---------------------------------------------------------------------------------------------
// Membrane part of an EMI row: coupling C_phi (u_i - u_e)(v_i - v_e) and Robin RHS
// (emiWeakForm.py:160-165,228-239).  Everything static about the (row, facet) entries e0 .. e0 + ne - 1 was
// flattened at set-up -- model slot, the facet-mass row (computed once on the device by
// `membrane_mass_kernel`, same `facet_mass_row` as before), the Q dofs and the byte-packed CSR slots -- so the
// row needs one level of loads for the matrix and one more (phi_M) for the right-hand side, instead of the
// former chain row -> list pointer -> entry -> facet -> vertex ids -> vertex records.
// ---------------------------------------------------------------------------------------------
template <int NF>
__device__ __forceinline__ double emi_membrane_entry_rhs(const KnDev& D, const KnConsts& C, int e, int ms, int splitting) {
  double gs = 0.0;
#pragma unroll
  for (int bb = 0; bb < NF; ++bb) {
    const int q = D.me_q[(size_t)e * NF + bb];
    double gq = D.phiM[q];
    if (!(splitting & 1)) {
      double it = 0.0;
      for (int k = 0; k < C.K; ++k) it += D.Ich[((size_t)ms * KN_MAXK + k) * D.NQtot + q];
      gq -= it / C.C_phi;
    }
    gs += D.me_mass[(size_t)e * NF + bb] * gq;
  }
  return gs;
}

template <int NF>
__device__ __forceinline__ void emi_membrane_row(const KnDev& D, const KnConsts& C, int e0, int ne, int first, int stride,
                                                 bool cell_side, int rowbase, double* accA, int splitting, double& gam) {
  // the lanes of a row share its membrane entries: lane `first` of `stride` takes every stride-th one; the partial
  // Robin sums meet in the caller's shuffle reduction (emi_membrane_rhs_kernel forms the same partial sums)
  for (int e = e0 + first; e < e0 + ne; e += stride) {
    const int ms = D.me_model[e];
    if (ms < 0) continue;
    const uint64_t sl = D.mslots[e];
#pragma unroll
    for (int bb = 0; bb < NF; ++bb) {
      const double val = C.C_phi * D.me_mass[(size_t)e * NF + bb];
      const int io = rowbase + (int)((sl >> (8 * bb)) & 255);
      const int it2 = rowbase + (int)((sl >> (8 * (4 + bb))) & 255);
      unsafeAtomicAdd(&accA[io], val);
      unsafeAtomicAdd(&accA[it2], -val);
    }
    if (!(splitting & 2))
      gam += (cell_side ? 1.0 : -1.0) * C.C_phi * emi_membrane_entry_rhs<NF>(D, C, e, ms, splitting);
  }
}

// one-time fill of the facet-mass rows of the membrane entries (set-up; geometry is static)
template <int NF>
__global__ void membrane_mass_kernel(KnDev D, int n_entries, int v_cells, const int* __restrict__ entry_row,
                                     double* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_entries) return;
  const int ent = D.mentry[e];
  const int fg = ent >> 3, a = ent & 7;
  const int* fown = entry_row[e] >= v_cells ? D.fi : D.fe;
  Rec p[NF];
#pragma unroll
  for (int bb = 0; bb < NF; ++bb) p[bb] = load_rec(D.VR, fown[(size_t)fg * NF + bb]);
  double Mr[NF];
  facet_mass_row<NF>(p, a, Mr);
#pragma unroll
  for (int bb = 0; bb < NF; ++bb) out[(size_t)e * NF + bb] = Mr[bb];
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

// (defined with the membrane-facet code further down)
template <int NF, int KS>
__device__ __forceinline__ void membrane_entries_to_lds(const KnDev& D, const KnConsts& C, int e0, int ne, bool cell_side,
                                                        int splitting, int tid, double* gam);

// Early membrane integrals (knp_membrane_pre_kernel): b_knp contribution of membrane entry e from the potential that
// now sits in the vertex records.
template <int NF, int KS>
__device__ __forceinline__ void membrane_entry_early(const KnDev& D, int e, double (&bk)[KS]) {
  constexpr int GS = (1 + NF) * KS;
  const double* g = D.gpre + (size_t)GS * e;
  const int fg = D.mentry[e] >> 3;
  double jump[NF];
#pragma unroll
  for (int b = 0; b < NF; ++b)
    jump[b] = D.VR[(size_t)D.fi[(size_t)fg * NF + b] * KN_REC + 7] - D.VR[(size_t)D.fe[(size_t)fg * NF + b] * KN_REC + 7];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    double v = g[k];
#pragma unroll
    for (int b = 0; b < NF; ++b) v -= g[(1 + b) * KS + k] * jump[b];
    bk[k] += v;
  }
}

// staged record of a KNP row block: coordinates, f_k = c_prev_k / dt (+ f_source_k) of the KS solved ions, phi
template <int KS>
struct RecK {
  double x, y, z, f[KS], c;   // c = phi
};

template <int KS>
__device__ __forceinline__ RecK<KS> lds_rec(const double* recs, int i) {
  RecK<KS> r;
  if constexpr (KS == 2) {   // 48-byte records: three 16-byte LDS reads
    const double2* p = reinterpret_cast<const double2*>(recs + (size_t)i * 6);
    const double2 u = p[0], v = p[1], w = p[2];
    r.x = u.x; r.y = u.y; r.z = v.x; r.f[0] = v.y; r.f[1] = w.x; r.c = w.y;
  } else {
    const double* p = recs + (size_t)i * (4 + KS);
    r.x = p[0]; r.y = p[1]; r.z = p[2];
#pragma unroll
    for (int k = 0; k < KS; ++k) r.f[k] = p[3 + k];
    r.c = p[3 + KS];
  }
  return r;
}

// Staged records of a UNIFORM hexahedral mesh (every cell the same parallelepiped: the geometry is a constant of the
// mesh, knpemi_handle::hex_geo, and no coordinates are staged): EMI 16 bytes (kappa, sigma), KNP 8 (KS + 1) bytes
// (f_0 .. f_{KS-1}, phi) -- 40 % / 50 % of the 40- / 48-byte records, in LDS bytes staged, LDS bytes read per (row, cell)
// pair and registers held per pair (8 records at once).
struct Rec2 { double k, s; };
__device__ __forceinline__ Rec2 lds_rec2(const double* recs, int i) {
  const double2 u = *reinterpret_cast<const double2*>(recs + (size_t)i * 2);
  return Rec2{u.x, u.y};
}
template <int KS>
struct RecU { double f[KS], c; };   // c = phi
template <int KS>
__device__ __forceinline__ RecU<KS> lds_recu(const double* recs, int i) {
  RecU<KS> r;
  const double* p = recs + (size_t)i * (KS + 1);
#pragma unroll
  for (int k = 0; k < KS; ++k) r.f[k] = p[k];
  r.c = p[KS];
  return r;
}

template <int GDIM, class R>
__device__ __forceinline__ double simplex_row0(const R (&r)[GDIM + 1], double (&d)[GDIM + 1]) {
  // gradient dot products of lambda_0 with all lambda_j, and the cell measure
  if constexpr (GDIM == 2) {
    const double e1x = r[1].x - r[0].x, e1y = r[1].y - r[0].y;
    const double e2x = r[2].x - r[0].x, e2y = r[2].y - r[0].y;
    const double det = e1x * e2y - e1y * e2x, inv = kn_rcp(det);
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
    const double det = ax * g1x + ay * g1y + az * g1z, inv = kn_rcp(det);
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
  int row0, nrows, sub, seg0, seglen, segL0, nnzLb, uoff, nuniq, slbase, np, me0, mne;
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
  B.me0 = i3.z; B.mne = i3.w;      // membrane entries of the block's rows: [me0, me0 + mne)
  return B;
}

// Rows of a block: KN_BLOCK / LPR / KN_CHUNK chunks of (up to) KN_CHUNK consecutive rows each (KN_CHUNK_SIMPLEX or
// KN_CHUNK_HEX; D.blk_rng, six ints per
// chunk: first row, rows, end of the chunk's entries in the block's concatenated EMI segment, global EMI position minus
// concatenated position, the same two for the Laplacian pattern).  A block of consecutive rows is the special case of
// consecutive chunks; clustered chunks (knpemi_create) touch fewer distinct vertices.
template <int LPR, int KN_CHUNK>
struct BlkRows {
  static constexpr int NR = KN_BLOCK / LPR / KN_CHUNK;
  const int* rng;
  __device__ __forceinline__ BlkRows(const KnDev& D, int b) : rng(D.blk_rng + (size_t)b * (6 * NR)) {}
  // global row of block-local row rloc (valid = false: a lane without a row; it takes the block's first row)
  __device__ __forceinline__ int row(int rloc, bool& valid) const {
    const int2 c = *reinterpret_cast<const int2*>(rng + 6 * (rloc / KN_CHUNK));
    valid = (rloc % KN_CHUNK) < c.y;
    return valid ? c.x + (rloc % KN_CHUNK) : rng[0];
  }
  // f(i, global CSR position) for every position i of the block's concatenated segment (which = 0: EMI pattern,
  // 1: Laplacian pattern), two chunks per pass: bounds and offsets are wave-uniform, the stores of a pass cover two
  // contiguous pieces of the global array
  template <class F>
  __device__ __forceinline__ void for_each_entry(int tid, int which, F&& f) const {
    static_assert(NR % 2 == 0, "chunks are walked in pairs");
    int lo = 0;
#pragma unroll
    for (int r = 0; r < NR; r += 2) {
      const int mid = rng[6 * r + 2 + 2 * which], hi = rng[6 * (r + 1) + 2 + 2 * which];
      const int d0 = rng[6 * r + 3 + 2 * which], d1 = rng[6 * (r + 1) + 3 + 2 * which];
      for (int i = lo + tid; i < hi; i += KN_BLOCK) f(i, (int64_t)i + (i >= mid ? d1 : d0));
      lo = hi;
    }
  }
};

#define KN_STAGE 3   // distinct vertices per thread staged with batched loads (256 threads x 3 = 768)

// Phase A of the v2 kernels: the 48-byte records of the block's distinct vertices go to `recs`, the
// 2-byte local index of every Laplacian entry to `eloc`.
// EMI records are 5 doubles: x y z | kappa = sum_k kap_k c_k | sigma = sum_k sig_k c_k (the two combinations of
// the concentrations the forms need; every staged vertex belongs to the block's sub-domain, so its constants
// apply).  KNP records are 6: x y z | f0 f1 | phi.
struct Rec5 { double x, y, z, k, s; };

__device__ __forceinline__ Rec5 lds_rec5(const double* recs, int i) {
  const double* p = recs + (size_t)i * 5;
  return Rec5{p[0], p[1], p[2], p[3], p[4]};
}

// one staged record from the 64-byte vertex record u (x y | z c3 | c0 c1 | c2 phi).  KS = 0: EMI (5 doubles),
// KS >= 1: KNP with KS solved ions (4 + KS doubles).  fs / nvs: ECS source term (NULL when unused), v the vertex.
template <int KS, bool UNI = false>
__device__ __forceinline__ void write_staged(double* recs, int i, const double2 (&u)[4], double inv_dt, const double* fs0,
                                             int nvs, int v, const KnSubConst* scp) {
  if constexpr (KS == 0) {
    const double c0 = u[2].x, c1 = u[2].y, c2 = u[3].x, c3 = u[1].y;   // ions beyond K have kap = sig = 0 (and c = 0)
    const double kap = scp->kap[0] * c0 + scp->kap[1] * c1 + scp->kap[2] * c2 + scp->kap[3] * c3;
    const double sig = scp->sig[0] * c0 + scp->sig[1] * c1 + scp->sig[2] * c2 + scp->sig[3] * c3;
    if constexpr (UNI) {
      *reinterpret_cast<double2*>(recs + (size_t)i * 2) = double2{kap, sig};
    } else {
      double* d5 = recs + (size_t)i * 5;
      d5[0] = u[0].x; d5[1] = u[0].y; d5[2] = u[1].x;
      d5[3] = kap;
      d5[4] = sig;
    }
  } else if constexpr (UNI) {
    const double cp[3] = {u[2].x, u[2].y, u[3].x};
    double* d = recs + (size_t)i * (KS + 1);
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      double f = cp[k] * inv_dt;
      if (fs0) f += fs0[(size_t)k * nvs + v];
      d[k] = f;
    }
    d[KS] = u[3].y;
  } else {
    const double cp[3] = {u[2].x, u[2].y, u[3].x};     // the solved ions live in slots 4..6
    double f[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      f[k] = cp[k] * inv_dt;                            // (1/dt) c_prev (+ f_source on the ECS)
      if (fs0) f[k] += fs0[(size_t)k * nvs + v];
    }
    if constexpr (KS == 2) {
      double2* dst = reinterpret_cast<double2*>(recs + (size_t)i * 6);
      dst[0] = u[0]; dst[1] = double2{u[1].x, f[0]}; dst[2] = double2{f[1], u[3].y};
    } else {
      double* d = recs + (size_t)i * (4 + KS);
      d[0] = u[0].x; d[1] = u[0].y; d[2] = u[1].x;
#pragma unroll
      for (int k = 0; k < KS; ++k) d[3 + k] = f[k];
      d[3 + KS] = u[3].y;
    }
  }
}

template <int KS, bool UNI = false>
__device__ __forceinline__ void stage_records(const KnDev& D, const BlkInfo& B, double* recs, uint16_t* eloc, int tid,
                                              double inv_dt, const double* fs0, int nvs,
                                              const KnSubConst* scp = nullptr) {
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
      u[k][0] = src[0]; u[k][1] = src[1]; u[k][2] = src[2]; u[k][3] = src[3];   // x y | z c3 | c0 c1 | c2 phi
    }
#pragma unroll
  for (int k = 0; k < KN_STAGE; ++k)
    if (vv[k] >= 0) write_staged<KS, UNI>(recs, tid + k * KN_BLOCK, u[k], inv_dt, fs0, nvs, vv[k], scp);
  for (int i = tid + KN_STAGE * KN_BLOCK; i < B.nuniq; i += KN_BLOCK) {   // oversized blocks only
    const int v = D.blk_uverts[B.uoff + i];
    const double2* src = reinterpret_cast<const double2*>(D.VR + (size_t)v * KN_REC);
    const double2 uu[4] = {src[0], src[1], src[2], src[3]};
    write_staged<KS, UNI>(recs, i, uu, inv_dt, fs0, nvs, v, scp);
  }
}

// Lattice tetrahedra of a uniform grid (knpemi_create, KnDev::tet_tab): a cell is one of at most eight shapes whose gradient
// dot products and volume are constants of the mesh, so the row kernels stage no coordinates and do no geometry: the pair
// entry names the shape and the canonical number of each of its vertices, the 4 + 1 numbers come from a 136-double table in
// LDS.  KN_TET_TAB doubles: [shape][a][b] g_a . g_b, then [shape] |T|.
constexpr int KN_TET_TAB = 8 * 16 + 8;
template <int NV>
__device__ __forceinline__ double tet_table_row0(const double* tab, uint32_t sl, double (&d)[NV]) {
  static_assert(NV == 4, "lattice tetrahedra");
  const int shape = (sl >> 5) & 7;
  const int c1 = (sl >> 13) & 3, c2 = (sl >> 21) & 3, c3 = (sl >> 29) & 3, c0 = 6 - c1 - c2 - c3;
  const double* row = tab + (shape * 4 + c0) * 4;
  d[0] = row[c0]; d[1] = row[c1]; d[2] = row[c2]; d[3] = row[c3];
  return tab[128 + shape];
}

// Diagnostic build (make CXXFLAGS+=-DKN_ROW_STAMPS): emi_rows_v2 adds up s_memtime differences between its phases (first wave
// of every workgroup); the launcher prints the averages every 16 launches.
#ifdef KN_ROW_STAMPS
__device__ unsigned long long row_stamp_acc[1024 * 8];
#define ROW_STAMP_BEGIN unsigned long long row_t_ = __builtin_amdgcn_s_memtime();
#define ROW_STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
                          if (threadIdx.x == 0) atomicAdd(&row_stamp_acc[(blockIdx.x & 1023) * 8 + (i)], n_ - row_t_); row_t_ = n_; } while (0)
#else
#define ROW_STAMP_BEGIN
#define ROW_STAMP(i) do {} while (0)
#endif

// UT: lattice tetrahedra of a uniform grid (tet_table_row0)
template <int GDIM, int LPR, bool UT = false>
__global__ __launch_bounds__(KN_BLOCK) void emi_rows_v2(KnDev D, const KnConsts* __restrict__ Cp, int acc_n,
                                                        int rec_n, int want_p, int splitting) {
  constexpr int NV = GDIM + 1, NF = GDIM;
  const KnConsts& C = *Cp;
  extern __shared__ __align__(16) double lds[];
  constexpr int RD = UT ? 2 : 5;                       // doubles per staged record
  constexpr int SM = UT ? 31 : 255;                    // slot mask of a pair-entry byte (= the "no slot" value)
  using RecT = std::conditional_t<UT, Rec2, Rec5>;
  double* accA = lds;
  double* tab = lds + (size_t)acc_n;                   // UT: the shape table
  double* recs = tab + (UT ? KN_TET_TAB : 0);
  uint16_t* eloc = reinterpret_cast<uint16_t*>(recs + RD * (size_t)rec_n + ((RD * rec_n) & 1));
  const int tid = threadIdx.x;
  ROW_STAMP_BEGIN
  const int b = logical_block(blockIdx.x, D.nblocks);
  const BlkInfo B = load_blk(D, b, tid >> 6);
  const int s = B.sub, seglen = B.seglen;
  // this lane's pair entries and row descriptor: issued first, their latency overlaps phase A
  const int rloc = tid / LPR, sub = tid % LPR;
  const BlkRows<LPR, KN_CHUNK_SIMPLEX> rows(D, b);
  bool valid;
  const int g = rows.row(rloc, valid);
  const int64_t base = (int64_t)B.slbase * KN_SLICE + (tid & 63);
  const int np = B.np;
  uint32_t slr[KN_PREFETCH];
#pragma unroll
  for (int p = 0; p < KN_PREFETCH; ++p)
    slr[p] = (valid && p < np) ? D.pair_sl[base + (int64_t)p * KN_SLICE] : 0xFFFFFFFFu;
  const int4 ri = D.row_info[g];
  // phase A: zero accumulators, stage the records of the block's Laplacian entries
  for (int i = tid; i < seglen; i += KN_BLOCK) accA[i] = 0.0;
  const KnSubConst& sc = C.sc[s];
  if constexpr (UT) { if (tid < KN_TET_TAB) tab[tid] = D.tet_tab[tid]; }
  stage_records<0, UT>(D, B, recs, eloc, tid, 0.0, nullptr, 0, &sc);
  ROW_STAMP(0);     // descriptors -> vertex ids -> records -> LDS
  __syncthreads();
  ROW_STAMP(1);     // barrier

  const bool cell_side = s > 0;
  double bacc = 0.0, gam = 0.0;   // volume part / membrane Robin part of b_emi
  if (valid) {
    const int rowbase = ri.x, lap = ri.y & 0xFFFF, ne = (unsigned)ri.y >> 16, rL = ri.z;
    // The diagonal entry receives a term from every pair: keep it in registers and add it once.
    int diag = -1;
    double dA = 0.0;
    RecT r[NV];
    auto rec = [&](int i) { if constexpr (UT) return lds_rec2(recs, i); else return lds_rec5(recs, i); };
    uint32_t have = 0xFFFFFFFFu;     // slots whose contributions pend[1..] currently hold (SM: none)
    double pend[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) pend[j] = 0.0;
    auto do_pair = [&](uint32_t sl) {
      int slot[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) slot[j] = (sl >> (8 * j)) & SM;
      if (diag < 0) { diag = slot[0]; r[0] = rec(eloc[rL + diag]); }   // the row's own vertex, once
      // (The host orders a lane's pairs in strips that share all but one vertex; knp_rows_v2 keeps the shared records
      // in registers.  Here that costs the fifth block per CU -- 104 instead of 84 registers -- and the LDS pipe is
      // less loaded, 36 % against 51 %: measured 49.4 against 47.7 us at 995 k tets, so every record is read again.)
#pragma unroll
      for (int j = 1; j < NV; ++j) r[j] = rec(eloc[rL + slot[j]]);
      double d[NV];
      double vol;
      if constexpr (UT) vol = tet_table_row0<NV>(tab, sl, d);
      else vol = simplex_row0<GDIM>(r, d);
      double ksum = 0, sd = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        ksum += r[j].k;
        sd += r[j].s * d[j];
      }
      const double kbar = ksum * (1.0 / NV);
      bacc -= vol * sd;
      dA += vol * kbar * d[0];
      // Off-diagonal entries: in the strip order a vertex keeps its position over consecutive pairs, so its
      // contributions are summed in a register and go to LDS when another vertex takes the position.
#pragma unroll
      for (int j = 1; j < NV; ++j) {
        const int hs = (int)((have >> (8 * j)) & SM);
        if (slot[j] != hs) {
          if (hs != SM) unsafeAtomicAdd(&accA[lap + hs], pend[j]);
          pend[j] = 0.0;
        }
        pend[j] += vol * kbar * d[j];
      }
      have = sl;
    };
#pragma unroll
    for (int p = 0; p < KN_PREFETCH; ++p)
      if (slr[p] != 0xFFFFFFFFu) do_pair(slr[p]);
    for (int p = KN_PREFETCH; p < np; ++p) {
      const uint32_t sl = D.pair_sl[base + (int64_t)p * KN_SLICE];
      if (sl != 0xFFFFFFFFu) do_pair(sl);
    }
#pragma unroll
    for (int j = 1; j < NV; ++j) {
      const int hs = (int)((have >> (8 * j)) & SM);
      if (hs != SM) unsafeAtomicAdd(&accA[lap + hs], pend[j]);
    }
    if (diag >= 0) unsafeAtomicAdd(&accA[lap + diag], dA);
    if (ne > 0) emi_membrane_row<NF>(D, C, ri.w, ne, sub, LPR, cell_side, rowbase, accA, splitting, gam);
  }
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) { bacc += __shfl_xor(bacc, m); gam += __shfl_xor(gam, m); }
  if (valid && sub == 0) D.b_emi[g] = bacc + gam;
  ROW_STAMP(2);     // pair loop
  __syncthreads();
  ROW_STAMP(3);     // barrier
  rows.for_each_entry(tid, 0, [&](int i, int64_t gp) {
    const double a = accA[i];
    KN_ROW_STORE(a, &D.A_emi[gp]);
    if (want_p) KN_ROW_STORE(cell_side ? a + D.P_mass[gp - D.pmass0] : a, &D.P_emi[gp]);
  });
  ROW_STAMP(4);     // write-out issued
}

// MEM: where the membrane integrals of b_knp come from -- 0: the partial integrals knp_membrane_kernel left in gam_e
// (default), 1: evaluated here into LDS (KNPEMI_OPT_FUSE_MEMBRANE), 2: the early form (KNPEMI_MEMBRANE_EARLY).  A
// template parameter, not a run-time switch: the two optional paths cost the default one 46 registers per lane
// (140 instead of 94: three instead of five waves per SIMD).
template <int GDIM, int LPR, int KS, int MEM, bool UT = false>
__global__ __launch_bounds__(KN_BLOCK) void knp_rows_v2(KnDev D, const KnConsts* __restrict__ Cp, int acc_n, int rec_n,
                                                        int gam_n, int splitting) {
  constexpr int NV = GDIM + 1;
  const KnConsts& C = *Cp;
  extern __shared__ __align__(16) double lds[];
  constexpr int RD = UT ? KS + 1 : 4 + KS;             // doubles per staged record
  constexpr int SM = UT ? 31 : 255;                    // slot mask of a pair-entry byte (= the "no slot" value)
  using RecT = std::conditional_t<UT, RecU<KS>, RecK<KS>>;
  double* acc = lds;                                   // KS accumulator arrays of acc_n doubles, one per solved ion
  double* tab = lds + (size_t)KS * acc_n;              // UT: the shape table of the lattice tetrahedra
  double* recs = tab + (UT ? KN_TET_TAB : 0);
  double* gam = recs + (size_t)RD * rec_n + ((RD * rec_n) & 1);   // gam_n > 0: fused membrane integrals
  uint16_t* eloc = reinterpret_cast<uint16_t*>(gam + (size_t)KS * gam_n);
  const int tid = threadIdx.x;
  const int b = logical_block(blockIdx.x, D.nblocks);
  const BlkInfo B = load_blk(D, b, tid >> 6);
  const int s = B.sub, nnzLb = B.nnzLb;
  const int v0 = C.voff[s], nvs = C.voff[s + 1] - v0;
  const double* fs0 = (D.fsrc && s == 0) ? D.fsrc : nullptr;
  const int rloc = tid / LPR, sub = tid % LPR;
  const BlkRows<LPR, KN_CHUNK_SIMPLEX> rows(D, b);
  bool valid;
  const int g = rows.row(rloc, valid);
  const int64_t base = (int64_t)B.slbase * KN_SLICE + (tid & 63);
  const int np = B.np;
  uint32_t slr[KN_PREFETCH];
#pragma unroll
  for (int p = 0; p < KN_PREFETCH; ++p)
    slr[p] = (valid && p < np) ? D.pair_sl[base + (int64_t)p * KN_SLICE] : 0xFFFFFFFFu;
  const int4 ri = D.row_info[g];
  for (int i = tid; i < nnzLb; i += KN_BLOCK) {
#pragma unroll
    for (int k = 0; k < KS; ++k) acc[(size_t)k * acc_n + i] = 0.0;
  }
  if constexpr (UT) { if (tid < KN_TET_TAB) tab[tid] = D.tet_tab[tid]; }
  stage_records<KS, UT>(D, B, recs, eloc, tid, C.inv_dt, fs0, nvs);
  if constexpr (MEM == 1) membrane_entries_to_lds<GDIM, KS>(D, C, B.me0, B.mne, s > 0, splitting, tid, gam);
  __syncthreads();

  const KnSubConst& sc = C.sc[s];
  double bk[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) bk[k] = 0.0;
  if (valid) {
    const int rL = ri.z;
    int diag = -1;
    double dk[KS];   // diagonal entries of the ion blocks, added once after the loop
#pragma unroll
    for (int k = 0; k < KS; ++k) dk[k] = 0.0;
    RecT r[NV];
    auto rec = [&](int i) { if constexpr (UT) return lds_recu<KS>(recs, i); else return lds_rec<KS>(recs, i); };
    // The host orders a lane's pairs in strips: consecutive cells share all but one vertex, at the same byte positions
    // of the pair entry.  The records of the shared vertices stay in registers; only a slot that changed is read from
    // LDS again (one 48-byte record per pair instead of three: knp_rows 53.6 -> 46.9 us at 995 k tets, with four
    // instead of five blocks per CU -- 112 registers; capping them at 96 spills and gives the gain back).
    uint32_t have = 0xFFFFFFFFu;     // slots of the records r[1..] (and of the pending sums) currently held; 255: none
    double pend[NV][KS];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int k = 0; k < KS; ++k) pend[j][k] = 0.0;
    auto do_pair = [&](uint32_t sl) {
      int slot[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) slot[j] = (sl >> (8 * j)) & SM;
      if (diag < 0) { diag = slot[0]; r[0] = rec(eloc[rL + diag]); }
#pragma unroll
      for (int j = 1; j < NV; ++j)
        if (slot[j] != (int)((have >> (8 * j)) & SM)) r[j] = rec(eloc[rL + slot[j]]);
      double d[NV];
      double vol;
      if constexpr (UT) vol = tet_table_row0<NV>(tab, sl, d);
      else vol = simplex_row0<GDIM>(r, d);
      double gp = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) gp += r[j].c * d[j];
      const double m = vol * (1.0 / ((GDIM + 1) * (GDIM + 2)));
      const double drift = gp * vol * (1.0 / (GDIM + 1));
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        dk[k] += 2.0 * m * C.inv_dt + sc.D[k] * vol * d[0] + sc.zpsiD[k] * drift;
        bk[k] += 2.0 * m * r[0].f[k];
      }
      // off-diagonal entries: summed in registers while a vertex keeps its position, to LDS when it leaves (emi_rows_v2)
#pragma unroll
      for (int j = 1; j < NV; ++j) {
        const int hs = (int)((have >> (8 * j)) & SM);
        const bool moved = slot[j] != hs;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          if (moved) {
            if (hs != SM) unsafeAtomicAdd(&acc[(size_t)k * acc_n + rL + hs], pend[j][k]);
            pend[j][k] = 0.0;
          }
          pend[j][k] += m * C.inv_dt + sc.D[k] * vol * d[j] + sc.zpsiD[k] * drift;
          bk[k] += m * r[j].f[k];
        }
      }
      have = sl;
    };
#pragma unroll
    for (int p = 0; p < KN_PREFETCH; ++p)
      if (slr[p] != 0xFFFFFFFFu) do_pair(slr[p]);
    for (int p = KN_PREFETCH; p < np; ++p) {
      const uint32_t sl = D.pair_sl[base + (int64_t)p * KN_SLICE];
      if (sl != 0xFFFFFFFFu) do_pair(sl);
    }
#pragma unroll
    for (int j = 1; j < NV; ++j) {
      const int hs = (int)((have >> (8 * j)) & SM);
#pragma unroll
      for (int k = 0; k < KS; ++k)
        if (hs != SM) unsafeAtomicAdd(&acc[(size_t)k * acc_n + rL + hs], pend[j][k]);
    }
    if (diag >= 0) {
#pragma unroll
      for (int k = 0; k < KS; ++k) unsafeAtomicAdd(&acc[(size_t)k * acc_n + rL + diag], dk[k]);
    }
    // membrane Robin/coupling contributions, written per (row, facet) entry by knp_membrane_kernel
    {   // the row's LPR lanes share its membrane entries (the sums meet in the reduction below)
      const int ne = (unsigned)ri.y >> 16;
      for (int e = ri.w + sub; e < ri.w + ne; e += LPR) {
        if constexpr (MEM == 2) membrane_entry_early<GDIM, KS>(D, e, bk);
        else {
#pragma unroll
          for (int k = 0; k < KS; ++k) bk[k] += MEM == 1 ? gam[(size_t)(e - B.me0) * KS + k] : D.gam_e[(size_t)KS * e + k];
        }
      }
    }
  }
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) {
#pragma unroll
    for (int k = 0; k < KS; ++k) bk[k] += __shfl_xor(bk[k], m);
  }
  if (valid && sub == 0) {
    const size_t bb = (size_t)KS * v0 + (size_t)(g - v0);
#pragma unroll
    for (int k = 0; k < KS; ++k) D.b_knp[bb + (size_t)k * nvs] = bk[k];
  }
  __syncthreads();
  const int64_t subnnz0 = D.rowptrL[v0], subnnz = D.rowptrL[v0 + nvs] - subnnz0;
  double* out0 = D.A_knp + (size_t)KS * subnnz0 - subnnz0;
  rows.for_each_entry(tid, 1, [&](int i, int64_t gp) {
#pragma unroll
    for (int k = 0; k < KS; ++k) KN_ROW_STORE(acc[(size_t)k * acc_n + i], &out0[(size_t)k * subnnz + gp]);
  });
}

// =============================================================================================
// Q1 hexahedra, version 2: the same row-block scheme as the simplex kernels (distinct vertex records
// staged in LDS, LDS fp64 adds, one launch descriptor per block) with the 2x2x2 Gauss tables folded
// into the instruction stream: every loop over quadrature points and vertices is unrolled with constexpr
// weights, so no table is read at run time.  Per quadrature point the row of vertex l needs
//   d_j = grad N_l . grad N_j = dN_j . w,   w = J^-1 (J^-T dN_l)
// i.e. the Jacobian (from the 12 edge vectors), its inverse and one 3-vector; the physical gradients
// of the other seven basis functions are never formed.  Nodal fields enter through their reference
// gradients (edge differences), the mass row M_j = sum_q w_q N_l N_j is shared by the time-derivative,
// drift and right-hand-side terms.
// =============================================================================================
namespace hexq {
constexpr double G0 = 0.5 - 0.28867513459481287, G1 = 0.5 + 0.28867513459481287;
__host__ __device__ constexpr double f(int q, int v, int ax) {
  const double x = ((q >> ax) & 1) ? G1 : G0;
  return ((v >> ax) & 1) ? x : 1.0 - x;
}
__host__ __device__ constexpr double N(int q, int v) { return f(q, v, 0) * f(q, v, 1) * f(q, v, 2); }
// |dN_v/dxi_t| at point q (independent of bit t of v); the sign is + when bit t of v is set
__host__ __device__ constexpr double E(int q, int v, int t) { return f(q, v, (t + 1) % 3) * f(q, v, (t + 2) % 3); }
// k-th vertex (k = 0..3) whose bit t is clear, and the inverse map
__host__ __device__ constexpr int lo(int t, int k) { return ((k >> t) << (t + 1)) | (k & ((1 << t) - 1)); }
__host__ __device__ constexpr int kof(int t, int v) { return ((v >> (t + 1)) << t) | (v & ((1 << t) - 1)); }
struct Point { double E[3][4]; double N[8]; double x[3]; };   // weights of one Gauss point
struct Table { Point p[8]; };
constexpr Table make_table() {
  Table T{};
  for (int q = 0; q < 8; ++q) {
    for (int t = 0; t < 3; ++t) {
      for (int k = 0; k < 4; ++k) T.p[q].E[t][k] = E(q, lo(t, k), t);
      T.p[q].x[t] = ((q >> t) & 1) ? G1 : G0;
    }
    for (int v = 0; v < 8; ++v) T.p[q].N[v] = N(q, v);
  }
  return T;
}
}  // namespace hexq

// read with a wave-uniform index: the weights arrive in scalar registers
__constant__ hexq::Table c_hex = hexq::make_table();

struct HexGeo {   // geometry of the row vertex l at one Gauss point
  double w[3];    // J^-1 J^-T dN_l
  double wd;      // weight * |det J|
  double Nl;      // N_l
};

// edge vectors of the cell: e[t][k] = x[v | 1 << t] - x[v], v = lo(t, k)
struct HexEdges { double x[3][4], y[3][4], z[3][4]; };

template <class R>
__device__ __forceinline__ HexEdges hex_edges(const R (&r)[8]) {
  HexEdges e;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = hexq::lo(t, k), u = v | (1 << t);
      e.x[t][k] = r[u].x - r[v].x; e.y[t][k] = r[u].y - r[v].y; e.z[t][k] = r[u].z - r[v].z;
    }
  return e;
}

struct HexInv { double i[3][3]; double wd; };   // J^-1[t][g] and weight * |det J|

// Jacobian (from the edge vectors) and its inverse at the Gauss point with weights P
__device__ __forceinline__ HexInv hex_inverse(const HexEdges& e, const hexq::Point& P) {
  double J[3][3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    double jx = 0, jy = 0, jz = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { jx += P.E[t][k] * e.x[t][k]; jy += P.E[t][k] * e.y[t][k]; jz += P.E[t][k] * e.z[t][k]; }
    J[0][t] = jx; J[1][t] = jy; J[2][t] = jz;
  }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, inv = kn_rcp(det);
  HexInv I;
  I.i[0][0] = c00 * inv; I.i[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * inv;
  I.i[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * inv;
  I.i[1][0] = c01 * inv; I.i[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * inv;
  I.i[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * inv;
  I.i[2][0] = c02 * inv; I.i[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * inv;
  I.i[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * inv;
  I.wd = 0.125 * fabs(det);
  return I;
}

// Parallelepipeds (every box mesh of the reference, make_mesh_3D.py:101) have a constant Jacobian: the four
// edge vectors of each axis coincide.  The row kernels then invert it once per cell instead of per point.
__device__ __forceinline__ bool hex_is_affine(const HexEdges& e) {
  double dev = 0.0, len = 0.0;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    len += fabs(e.x[t][0]) + fabs(e.y[t][0]) + fabs(e.z[t][0]);
#pragma unroll
    for (int k = 1; k < 4; ++k)
      dev += fabs(e.x[t][k] - e.x[t][0]) + fabs(e.y[t][k] - e.y[t][0]) + fabs(e.z[t][k] - e.z[t][0]);
  }
  return dev <= 1e-13 * len;
}

__device__ __forceinline__ HexGeo hex_geo(const HexInv& I, const hexq::Point& P, int l0, int l1, int l2) {
  const double f0 = l0 ? P.x[0] : 1.0 - P.x[0], f1 = l1 ? P.x[1] : 1.0 - P.x[1], f2 = l2 ? P.x[2] : 1.0 - P.x[2];
  const double a = (l0 ? f1 : -f1) * f2, b = (l1 ? f0 : -f0) * f2, c = (l2 ? f0 : -f0) * f1;   // dN_l
  const double g0 = a * I.i[0][0] + b * I.i[1][0] + c * I.i[2][0];
  const double g1 = a * I.i[0][1] + b * I.i[1][1] + c * I.i[2][1];
  const double g2 = a * I.i[0][2] + b * I.i[1][2] + c * I.i[2][2];
  HexGeo G;
  G.w[0] = I.i[0][0] * g0 + I.i[0][1] * g1 + I.i[0][2] * g2;
  G.w[1] = I.i[1][0] * g0 + I.i[1][1] * g1 + I.i[1][2] * g2;
  G.w[2] = I.i[2][0] * g0 + I.i[2][1] * g1 + I.i[2][2] * g2;
  G.wd = I.wd;
  G.Nl = f0 * f1 * f2;
  return G;
}

// reference gradient of a nodal field given by its edge differences d[t][k]
__device__ __forceinline__ void hex_ref_grad(const hexq::Point& P, const double (&d)[3][4], double (&g)[3]) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += P.E[t][k] * d[t][k];
    g[t] = s;
  }
}

__device__ __forceinline__ void hex_edge_diffs(const double (&u)[8], double (&d)[3][4]) {
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) d[t][k] = u[hexq::lo(t, k) | (1 << t)] - u[hexq::lo(t, k)];
}

// acc[j] += dN_j . v   (dN_j[t] = +-E[t][k(j)], the sign is that of bit t of j)
__device__ __forceinline__ void hex_add_grad_dot(const hexq::Point& P, double (&acc)[8], const double (&v)[3]) {
  double pv[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) pv[t][k] = P.E[t][k] * v[t];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const double a0 = pv[0][hexq::kof(0, j)], a1 = pv[1][hexq::kof(1, j)], a2 = pv[2][hexq::kof(2, j)];
    acc[j] += ((j & 1) ? a0 : -a0) + ((j & 2) ? a1 : -a1) + ((j & 4) ? a2 : -a2);
  }
}

__device__ __forceinline__ void hex_add_shape(const hexq::Point& P, double (&acc)[8], double a) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] += P.N[j] * a;
}

__device__ __forceinline__ double hex_interp(const hexq::Point& P, const double (&u)[8]) {
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += P.N[j] * u[j];
  return s;
}

// Row l of the EMI element matrix of one hexahedron: ra = kappa-stiffness, and the volume right-hand side
// (emiWeakForm.py:138-241).  The ICS mass of P_emi is static and added at write-out (D.P_mass).
template <bool AFFINE>
__device__ __forceinline__ void hex_emi_row(const Rec5 (&r)[8], int li, double (&ra)[8], double& bvol) {
  double kv[8], sv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { kv[j] = r[j].k; sv[j] = r[j].s; ra[j] = 0.0; }
  double ds[3][4];
  hex_edge_diffs(sv, ds);
  const HexEdges e = hex_edges(r);
  const bool aff = AFFINE || hex_is_affine(e);
  HexInv I = hex_inverse(e, c_hex.p[0]);
  const int l0 = li & 1, l1 = (li >> 1) & 1, l2 = (li >> 2) & 1;
#pragma unroll 1
  for (int q = 0; q < 8; ++q) {
    const hexq::Point& P = c_hex.p[q];
    if (!AFFINE && !aff && q > 0) I = hex_inverse(e, P);
    const HexGeo G = hex_geo(I, P, l0, l1, l2);
    const double kq = hex_interp(P, kv);
    double gs[3];
    hex_ref_grad(P, ds, gs);
    bvol -= G.wd * (G.w[0] * gs[0] + G.w[1] * gs[1] + G.w[2] * gs[2]);
    const double wk = G.wd * kq;
    const double v[3] = {G.w[0] * wk, G.w[1] * wk, G.w[2] * wk};
    hex_add_grad_dot(P, ra, v);
  }
}

// Row l of the KNP element matrices: M = mass, S = stiffness, Cc = drift (grad phi . grad N_l) N_j
template <bool AFFINE, class R8>
__device__ __forceinline__ void hex_knp_row(const R8 (&r)[8], int li, double (&M)[8], double (&S)[8], double (&Cc)[8]) {
  double phi[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { phi[j] = r[j].c; M[j] = 0.0; S[j] = 0.0; Cc[j] = 0.0; }
  double dphi[3][4];
  hex_edge_diffs(phi, dphi);
  const HexEdges e = hex_edges(r);
  const bool aff = AFFINE || hex_is_affine(e);
  HexInv I = hex_inverse(e, c_hex.p[0]);
  const int l0 = li & 1, l1 = (li >> 1) & 1, l2 = (li >> 2) & 1;
#pragma unroll 1
  for (int q = 0; q < 8; ++q) {
    const hexq::Point& P = c_hex.p[q];
    if (!AFFINE && !aff && q > 0) I = hex_inverse(e, P);
    const HexGeo G = hex_geo(I, P, l0, l1, l2);
    double gp[3];
    hex_ref_grad(P, dphi, gp);
    const double drift = G.w[0] * gp[0] + G.w[1] * gp[1] + G.w[2] * gp[2];   // grad phi . grad N_l
    const double v[3] = {G.w[0] * G.wd, G.w[1] * G.wd, G.w[2] * G.wd};
    hex_add_grad_dot(P, S, v);
    hex_add_shape(P, M, G.wd * G.Nl);
    hex_add_shape(P, Cc, G.wd * drift);
  }
}

// ---------------------------------------------------------------------------------------------
// Parallelepipeds: closed forms instead of the 2 x 2 x 2 rule (which is exact for these integrands when the Jacobian
// is constant, so both give the same numbers up to rounding).  The row vertex l is moved to local vertex 0 by
// reflecting the reference cell along the axes where l has a bit set: the caller loads the records in the order
// j' = j ^ l, the edge vectors taken from the permuted records carry the reflection, and every result of the frame
// belongs to the cell's local vertex j' ^ l.  With l = 0 the integrals over the unit cube factor into 1D integrals of
// two or three linear functions (1/3, 1/6; 1/4, 1/12), contracted axis by axis (sum factorisation): ~1/5 of the
// operations of the quadrature loop on a box mesh, whose metric tensor is diagonal.
// ---------------------------------------------------------------------------------------------
namespace hexcf {
constexpr double T4 = 0.25, T12 = 1.0 / 12.0, A3 = 1.0 / 3.0, A6 = 1.0 / 6.0;
__host__ __device__ constexpr int ix(int s, int ms, int a, int ma, int b, int mb) { return (ms << s) | (ma << a) | (mb << b); }
__host__ __device__ constexpr double a0(int j) { return j ? A6 : A3; }
// P[ja][jb] = sum_{ma, mb} b(ma, 0, ja) b(mb, 0, jb) k[ma][mb],  b(m, 0, j) = 1/4 if m = j = 0 else 1/12
__device__ __forceinline__ void contract_bb(const double (&k)[2][2], double (&P)[2][2]) {
  const double r00 = T4 * k[0][0] + T12 * k[1][0], r01 = T4 * k[0][1] + T12 * k[1][1];
  const double r10 = T12 * (k[0][0] + k[1][0]), r11 = T12 * (k[0][1] + k[1][1]);
  P[0][0] = T4 * r00 + T12 * r01; P[0][1] = T12 * (r00 + r01);
  P[1][0] = T4 * r10 + T12 * r11; P[1][1] = T12 * (r10 + r11);
}
// Q[js][ju] = sum_{ms, mu} a(ms, js) b(mu, 0, ju) k[ms][mu],  a(m, j) = 1/3 if m = j else 1/6
__device__ __forceinline__ void contract_ab(const double (&k)[2][2], double (&Q)[2][2]) {
  const double v00 = T4 * k[0][0] + T12 * k[0][1], v01 = T12 * (k[0][0] + k[0][1]);
  const double v10 = T4 * k[1][0] + T12 * k[1][1], v11 = T12 * (k[1][0] + k[1][1]);
  Q[0][0] = A3 * v00 + A6 * v10; Q[0][1] = A3 * v01 + A6 * v11;
  Q[1][0] = A6 * v00 + A3 * v10; Q[1][1] = A6 * v01 + A3 * v11;
}
struct Geo { double g[3][3]; double det; bool skew; };   // g = |det J| J^-1 J^-T
template <class R>
__device__ __forceinline__ Geo geometry(const R (&r)[8]) {
  double J[3][3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    J[0][t] = r[1 << t].x - r[0].x; J[1][t] = r[1 << t].y - r[0].y; J[2][t] = r[1 << t].z - r[0].z;
  }
  const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
  const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
  const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02, inv = kn_rcp(det);
  double I[3][3];
  I[0][0] = c00 * inv; I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * inv; I[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * inv;
  I[1][0] = c01 * inv; I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * inv; I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * inv;
  I[2][0] = c02 * inv; I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * inv; I[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * inv;
  Geo G;
  G.det = fabs(det);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = a; b < 3; ++b) {
      const double v = G.det * (I[a][0] * I[b][0] + I[a][1] * I[b][1] + I[a][2] * I[b][2]);
      G.g[a][b] = v; G.g[b][a] = v;
    }
  G.skew = G.g[0][1] != 0.0 || G.g[0][2] != 0.0 || G.g[1][2] != 0.0;   // false on every cell of a box mesh
  return G;
}
}  // namespace hexcf

namespace hexcf {
// The geometry of a cell of a uniform mesh in the frame of the row vertex: local vertex j ^ li takes position j, i.e. the
// edge vectors of the axes whose bit is set in li change sign: g[s][t] -> d_s d_t g[s][t], |det| unchanged.
__device__ __forceinline__ Geo reflected(const KnHexGeo& U, int li) {
  Geo G;
  G.det = U.det;
  G.skew = U.skew != 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) G.g[a][b] = (a != b && (((li >> a) ^ (li >> b)) & 1)) ? -U.g[a][b] : U.g[a][b];
  return G;
}
}  // namespace hexcf

// Row of local vertex 0 (records in the reflected order, see above): kappa-stiffness ra, volume right-hand side.
// BOX: the metric tensor is known to be diagonal (uniform box meshes): the mixed terms are not even compiled
template <bool BOX = false, class R>
__device__ __forceinline__ void hex_emi_row0_g(const R (&r)[8], const hexcf::Geo& G, double (&ra)[8], double& bvol);
__device__ __forceinline__ void hex_emi_row0(const Rec5 (&r)[8], double (&ra)[8], double& bvol) {
  hex_emi_row0_g(r, hexcf::geometry(r), ra, bvol);
}
template <bool BOX, class R>
__device__ __forceinline__ void hex_emi_row0_g(const R (&r)[8], const hexcf::Geo& G, double (&ra)[8], double& bvol) {
  using namespace hexcf;
#pragma unroll
  for (int j = 0; j < 8; ++j) ra[j] = 0.0;
  double bv = 0.0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int a = (s + 1) % 3, b = (s + 2) % 3;
    double kb[2][2], P[2][2], ds = 0.0;
#pragma unroll
    for (int ma = 0; ma < 2; ++ma)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        kb[ma][mb] = r[ix(s, 0, a, ma, b, mb)].k + r[ix(s, 1, a, ma, b, mb)].k;
        ds += a0(ma) * a0(mb) * (r[ix(s, 1, a, ma, b, mb)].s - r[ix(s, 0, a, ma, b, mb)].s);
      }
    contract_bb(kb, P);
    const double c = 0.5 * G.g[s][s];
#pragma unroll
    for (int j = 0; j < 8; ++j) ra[j] += (((j >> s) & 1) ? -c : c) * P[(j >> a) & 1][(j >> b) & 1];
    bv += G.g[s][s] * ds;
  }
  if (!BOX && G.skew) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (s == t) continue;
        const int u = 3 - s - t;
        double kt[2][2], Q[2][2], ds = 0.0;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int mu = 0; mu < 2; ++mu) {
            kt[ms][mu] = A3 * r[ix(s, ms, u, mu, t, 0)].k + A6 * r[ix(s, ms, u, mu, t, 1)].k;
            ds += a0(mu) * (r[ix(s, ms, u, mu, t, 1)].s - r[ix(s, ms, u, mu, t, 0)].s);
          }
        contract_ab(kt, Q);
        const double g = G.g[s][t];
#pragma unroll
        for (int j = 0; j < 8; ++j) ra[j] += (((j >> t) & 1) ? -g : g) * Q[(j >> s) & 1][(j >> u) & 1];
        bv += 0.25 * g * ds;
      }
  }
  bvol += bv;
}

// Row of local vertex 0 of the KNP element matrices (records in the reflected order): mass, stiffness, drift.
template <bool BOX = false, class R8>
__device__ __forceinline__ void hex_knp_row0_g(const R8 (&r)[8], const hexcf::Geo& G, double (&M)[8], double (&S)[8], double (&Cc)[8]);
template <class R8>
__device__ __forceinline__ void hex_knp_row0(const R8 (&r)[8], double (&M)[8], double (&S)[8], double (&Cc)[8]) {
  hex_knp_row0_g(r, hexcf::geometry(r), M, S, Cc);
}
template <bool BOX, class R8>
__device__ __forceinline__ void hex_knp_row0_g(const R8 (&r)[8], const hexcf::Geo& G, double (&M)[8], double (&S)[8], double (&Cc)[8]) {
  using namespace hexcf;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    M[j] = G.det * (a0(j & 1) * a0((j >> 1) & 1) * a0((j >> 2) & 1));
    S[j] = 0.0; Cc[j] = 0.0;
  }
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int a = (s + 1) % 3, b = (s + 2) % 3;
    double dp[2][2], P[2][2];
#pragma unroll
    for (int ma = 0; ma < 2; ++ma)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) dp[ma][mb] = r[ix(s, 1, a, ma, b, mb)].c - r[ix(s, 0, a, ma, b, mb)].c;
    contract_bb(dp, P);
    const double g = G.g[s][s];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double w = a0((j >> a) & 1) * a0((j >> b) & 1);
      S[j] += (((j >> s) & 1) ? -g : g) * w;
      Cc[j] -= 0.5 * g * P[(j >> a) & 1][(j >> b) & 1];
    }
  }
  if (!BOX && G.skew) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (s == t) continue;
        const int u = 3 - s - t;
        double dp[2][2], Q[2][2];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int mu = 0; mu < 2; ++mu) dp[ms][mu] = r[ix(s, ms, u, mu, t, 1)].c - r[ix(s, ms, u, mu, t, 0)].c;
        contract_ab(dp, Q);
        const double g = G.g[s][t];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          S[j] += (((j >> t) & 1) ? -0.25 * g : 0.25 * g) * a0((j >> u) & 1);
          Cc[j] -= g * a0((j >> t) & 1) * Q[(j >> s) & 1][(j >> u) & 1];
        }
      }
  }
}

// GEO: 0 general trilinear cells (2 x 2 x 2 quadrature), 1 every cell a parallelepiped (closed forms, geometry from the
// staged coordinates), 2 every cell the SAME box (closed forms, the diagonal metric U a constant of the mesh, records without
// coordinates: 16 instead of 40 bytes).  Measured at 165 888 hexahedra (config 2h, round 4): GEO 1 -> 2 takes emi_rows from
// 134 to 99 registers (3 -> 4 waves per SIMD) and 59.9 -> 45.0 us alone (0.29 -> 0.39 of 8 TB/s), 99 -> 64.5 us beside the
// ODE sweep; knp_rows 168 registers + spills -> 123, 59.1 -> 46.1 us (0.285 -> 0.365).  Capping emi_rows at 96 registers
// for a fifth wave per SIMD is slower (53.8 us): fewer registers serialise the eight record reads of a pair.
template <int LPR, int GEO>
__global__ __launch_bounds__(KN_BLOCK, GEO == 2 ? 4 : (GEO == 1 ? 3 : 2)) void emi_rows_hex_v2(
    KnDev D, const KnConsts* __restrict__ Cp, int acc_n, int rec_n, int want_p, int splitting, KnHexGeo U) {
  constexpr int NF = 4;
  constexpr bool AFFINE = GEO >= 1, UNI = GEO == 2;
  constexpr int RW = UNI ? 2 : 5;
  const KnConsts& C = *Cp;
  extern __shared__ __align__(16) double lds[];
  double* accA = lds;
  double* recs = lds + (size_t)acc_n;
  uint16_t* eloc = reinterpret_cast<uint16_t*>(recs + RW * (size_t)rec_n + ((RW * rec_n) & 1));
  const int tid = threadIdx.x;
  const int b = logical_block(blockIdx.x, D.nblocks);
  const BlkInfo B = load_blk(D, b, tid >> 6);
  const int s = B.sub, seglen = B.seglen;
  const int rloc = tid / LPR, sub = tid % LPR;
  const BlkRows<LPR, KN_CHUNK_HEX> rows(D, b);
  bool valid;
  const int g = rows.row(rloc, valid);
  const int64_t base = (int64_t)B.slbase * KN_SLICE + (tid & 63);
  const int np = B.np;
  // first pair entry of this lane: issued before phase A so that its latency overlaps the staging
  int pc0 = -1;
  uint2 sl0 = {0u, 0u};
  if (valid && np > 0) {
    pc0 = D.pair_cell[base];
    sl0 = *reinterpret_cast<const uint2*>(D.pair_slots + 2 * base);
  }
  const int4 ri = D.row_info[g];
  for (int i = tid; i < seglen; i += KN_BLOCK) accA[i] = 0.0;
  const KnSubConst& sc = C.sc[s];
  stage_records<0, UNI>(D, B, recs, eloc, tid, 0.0, nullptr, 0, &sc);
  __syncthreads();

  const bool cell_side = s > 0;
  double bacc = 0.0, gam = 0.0;
  if (valid) {
    const int rowbase = ri.x, lap = ri.y & 0xFFFF, ne = (unsigned)ri.y >> 16, rL = ri.z;
    auto do_pair = [&](int pc, uint2 sl) {
      if constexpr (UNI) {
        const int li = pc & 7;
        const uint64_t s64 = ((uint64_t)sl.y << 32) | sl.x;
        int slot[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) slot[j] = (int)((s64 >> (8 * (j ^ li))) & 255);
        Rec2 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = lds_rec2(recs, eloc[rL + slot[j]]);
        double ra[8];
        hexcf::Geo G;          // box: diagonal metric, invariant under the reflections of the row-vertex frame
        G.det = U.det; G.skew = false;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) G.g[a][b2] = a == b2 ? U.g[a][a] : 0.0;
        hex_emi_row0_g<true>(r, G, ra, bacc);
#pragma unroll
        for (int j = 0; j < 8; ++j) unsafeAtomicAdd(&accA[lap + slot[j]], ra[j]);
        return;
      }
      const int li = pc & 7;
      int slot[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { slot[j] = (sl.x >> (8 * j)) & 255; slot[4 + j] = (sl.y >> (8 * j)) & 255; }
      if constexpr (AFFINE) {   // frame of the row vertex: local vertex j ^ li takes position j
        const uint64_t s64 = ((uint64_t)sl.y << 32) | sl.x;
#pragma unroll
        for (int j = 0; j < 8; ++j) slot[j] = (int)((s64 >> (8 * (j ^ li))) & 255);
      }
      Rec5 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = lds_rec5(recs, eloc[rL + slot[j]]);
      double ra[8];
      if constexpr (AFFINE) hex_emi_row0(r, ra, bacc);
      else hex_emi_row<false>(r, li, ra, bacc);
#pragma unroll
      for (int j = 0; j < 8; ++j) unsafeAtomicAdd(&accA[lap + slot[j]], ra[j]);
    };
    if (pc0 >= 0) do_pair(pc0, sl0);
    for (int p = 1; p < np; ++p) {
      const int64_t ent = base + (int64_t)p * KN_SLICE;
      const int pc = D.pair_cell[ent];
      if (pc >= 0) do_pair(pc, *reinterpret_cast<const uint2*>(D.pair_slots + 2 * ent));
    }
    if (ne > 0) emi_membrane_row<NF>(D, C, ri.w, ne, sub, LPR, cell_side, rowbase, accA, splitting, gam);
  }
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) { bacc += __shfl_xor(bacc, m); gam += __shfl_xor(gam, m); }
  if (valid && sub == 0) D.b_emi[g] = bacc + gam;
  __syncthreads();
  rows.for_each_entry(tid, 0, [&](int i, int64_t gp) {
    const double a = accA[i];
    KN_ROW_STORE(a, &D.A_emi[gp]);
    if (want_p) KN_ROW_STORE(cell_side ? a + D.P_mass[gp - D.pmass0] : a, &D.P_emi[gp]);
  });
}

template <int LPR, int GEO, int KS, int MEM>
__global__ __launch_bounds__(KN_BLOCK, GEO == 2 ? 4 : (GEO == 1 ? 3 : 2)) void knp_rows_hex_v2(
    KnDev D, const KnConsts* __restrict__ Cp, int acc_n, int rec_n, int gam_n, int splitting, KnHexGeo U) {
  constexpr bool AFFINE = GEO >= 1, UNI = GEO == 2;
  constexpr int RW = UNI ? KS + 1 : 4 + KS;
  const KnConsts& C = *Cp;
  extern __shared__ __align__(16) double lds[];
  double* acc = lds;
  double* recs = lds + (size_t)KS * acc_n;
  double* gam = recs + (size_t)RW * rec_n + ((RW * rec_n) & 1);
  uint16_t* eloc = reinterpret_cast<uint16_t*>(gam + (size_t)KS * gam_n);
  const int tid = threadIdx.x;
  const int b = logical_block(blockIdx.x, D.nblocks);
  const BlkInfo B = load_blk(D, b, tid >> 6);
  const int s = B.sub, nnzLb = B.nnzLb;
  const int v0 = C.voff[s], nvs = C.voff[s + 1] - v0;
  const double* fs0 = (D.fsrc && s == 0) ? D.fsrc : nullptr;
  const int rloc = tid / LPR, sub = tid % LPR;
  const BlkRows<LPR, KN_CHUNK_HEX> rows(D, b);
  bool valid;
  const int g = rows.row(rloc, valid);
  const int64_t base = (int64_t)B.slbase * KN_SLICE + (tid & 63);
  const int np = B.np;
  int pc0 = -1;
  uint2 sl0 = {0u, 0u};
  if (valid && np > 0) {
    pc0 = D.pair_cell[base];
    sl0 = *reinterpret_cast<const uint2*>(D.pair_slots + 2 * base);
  }
  const int4 ri = D.row_info[g];
  for (int i = tid; i < nnzLb; i += KN_BLOCK) {
#pragma unroll
    for (int k = 0; k < KS; ++k) acc[(size_t)k * acc_n + i] = 0.0;
  }
  stage_records<KS, UNI>(D, B, recs, eloc, tid, C.inv_dt, fs0, nvs);
  if constexpr (MEM == 1) membrane_entries_to_lds<4, KS>(D, C, B.me0, B.mne, s > 0, splitting, tid, gam);
  __syncthreads();

  const KnSubConst& sc = C.sc[s];
  double bk[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) bk[k] = 0.0;
  if (valid) {
    const int rL = ri.z;
    auto do_pair = [&](int pc, uint2 sl) {
      const int li = pc & 7;
      if constexpr (UNI) {
        const uint64_t s64 = ((uint64_t)sl.y << 32) | sl.x;
        int slot[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) slot[j] = (int)((s64 >> (8 * (j ^ li))) & 255);
        RecU<KS> r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = lds_recu<KS>(recs, eloc[rL + slot[j]]);
        double M[8], S[8], Cc[8];
        hexcf::Geo G;
        G.det = U.det; G.skew = false;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b2 = 0; b2 < 3; ++b2) G.g[a][b2] = a == b2 ? U.g[a][a] : 0.0;
        hex_knp_row0_g<true>(r, G, M, S, Cc);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            unsafeAtomicAdd(&acc[(size_t)k * acc_n + rL + slot[j]], M[j] * C.inv_dt + sc.D[k] * S[j] + sc.zpsiD[k] * Cc[j]);
            bk[k] += M[j] * r[j].f[k];
          }
        }
        return;
      }
      int slot[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { slot[j] = (sl.x >> (8 * j)) & 255; slot[4 + j] = (sl.y >> (8 * j)) & 255; }
      if constexpr (AFFINE) {   // frame of the row vertex: local vertex j ^ li takes position j
        const uint64_t s64 = ((uint64_t)sl.y << 32) | sl.x;
#pragma unroll
        for (int j = 0; j < 8; ++j) slot[j] = (int)((s64 >> (8 * (j ^ li))) & 255);
      }
      RecK<KS> r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = lds_rec<KS>(recs, eloc[rL + slot[j]]);
      double M[8], S[8], Cc[8];
      if constexpr (AFFINE) hex_knp_row0(r, M, S, Cc);
      else hex_knp_row<false>(r, li, M, S, Cc);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          unsafeAtomicAdd(&acc[(size_t)k * acc_n + rL + slot[j]], M[j] * C.inv_dt + sc.D[k] * S[j] + sc.zpsiD[k] * Cc[j]);
          bk[k] += M[j] * r[j].f[k];
        }
      }
    };
    if (pc0 >= 0) do_pair(pc0, sl0);
    for (int p = 1; p < np; ++p) {
      const int64_t ent = base + (int64_t)p * KN_SLICE;
      const int pc = D.pair_cell[ent];
      if (pc >= 0) do_pair(pc, *reinterpret_cast<const uint2*>(D.pair_slots + 2 * ent));
    }
    {   // the row's LPR lanes share its membrane entries (the sums meet in the reduction below)
      const int ne = (unsigned)ri.y >> 16;
      for (int e = ri.w + sub; e < ri.w + ne; e += LPR) {
        if constexpr (MEM == 2) membrane_entry_early<4, KS>(D, e, bk);
        else {
#pragma unroll
          for (int k = 0; k < KS; ++k) bk[k] += MEM == 1 ? gam[(size_t)(e - B.me0) * KS + k] : D.gam_e[(size_t)KS * e + k];
        }
      }
    }
  }
#pragma unroll
  for (int m = 1; m < LPR; m <<= 1) {
#pragma unroll
    for (int k = 0; k < KS; ++k) bk[k] += __shfl_xor(bk[k], m);
  }
  if (valid && sub == 0) {
    const size_t bb = (size_t)KS * v0 + (size_t)(g - v0);
#pragma unroll
    for (int k = 0; k < KS; ++k) D.b_knp[bb + (size_t)k * nvs] = bk[k];
  }
  __syncthreads();
  const int64_t subnnz0 = D.rowptrL[v0], subnnz = D.rowptrL[v0 + nvs] - subnnz0;
  double* out0 = D.A_knp + (size_t)KS * subnnz0 - subnnz0;
  rows.for_each_entry(tid, 1, [&](int i, int64_t gp) {
#pragma unroll
    for (int k = 0; k < KS; ++k) KN_ROW_STORE(acc[(size_t)k * acc_n + i], &out0[(size_t)k * subnnz + gp]);
  });
}

// ---------------------------------------------------------------------------------------------
// KNP membrane-facet kernel: the rational Robin/coupling integrand of knpWeakForm.py:168-214 with
// degree-6 quadrature (tables staged in LDS).  One thread per (membrane facet, side): it evaluates the
// integrand once per quadrature point and tests it against all NF facet basis functions, writing
// NF x 2 (ions) partial integrals to `gam_contrib`; the row kernel adds them into b_knp in a fixed
// order, so nothing is accumulated atomically.
// ---------------------------------------------------------------------------------------------
// everything one side of a membrane facet contributes to the integrand: loaded once per (facet, side)
template <int NF>
struct FacetData {
  Rec pe[NF], pi[NF];
  double pm[NF], Ik[NF][KN_MAXK], It[NF];
  double meas, sgn;
  const KnSubConst* so;    // own-side constants
  bool cell_side;
};

// phi_x != NULL: the potential is taken from that vector (one value per vertex, e.g. the solver's solution before it is
// written into the records -- a constant shift of it cancels in the jump) instead of the records' component 7
template <int NF>
__device__ __forceinline__ void load_facet(const KnDev& D, const KnConsts& C, int fg, bool cell_side, int ms,
                                           FacetData<NF>& f, const double* __restrict__ phi_x = nullptr) {
  const int K = C.K;
  int si = 0;  // sub-domain of the cell side of this facet
#pragma unroll
  for (int bb = 0; bb < NF; ++bb) {
    const int vi = D.fi[(size_t)fg * NF + bb], ve = D.fe[(size_t)fg * NF + bb];
    f.pe[bb] = load_rec(D.VR, ve);
    f.pi[bb] = load_rec(D.VR, vi);
    if (phi_x) { f.pe[bb].phi = phi_x[ve]; f.pi[bb].phi = phi_x[vi]; }
    const int q = D.fq[(size_t)fg * NF + bb];
    f.pm[bb] = D.phiM[q];
    const double* ich = D.Ich + (size_t)ms * KN_MAXK * D.NQtot + q;
    double it = 0.0;
#pragma unroll
    for (int k = 0; k < KN_MAXK; ++k) {
      f.Ik[bb][k] = k < K ? ich[(size_t)k * D.NQtot] : 0.0;
      it += f.Ik[bb][k];
    }
    f.It[bb] = it;
    if (bb == 0) for (int tt = 1; tt < C.n_sub; ++tt) si += vi >= C.voff[tt];
  }
  f.so = &C.sc[cell_side ? si : 0];
  f.meas = 0.0;
  if constexpr (NF != 4) f.meas = facet_measure<NF>(f.pe);
  f.sgn = cell_side ? 1.0 : -1.0;
  f.cell_side = cell_side;
}

// weight x integrand of the K - 1 solved ions at quadrature point q (knpWeakForm.py:178-214):
//   fk[k] = w_q * sgn * (C_k g_k - C_k [phi]),  C_k = alpha_k C_M / (F z_k dt),  alpha_k = D_k z_k^2 c_k / sum_j D_j z_j^2 c_j
template <int NF>
__device__ __forceinline__ void facet_point(const FacetData<NF>& f, const KnConsts& C, int q, const double* qw,
                                            const double* qN, const double* qdN, int splitting, double (&fk)[KN_MAXK - 1]) {
  const int KS = C.K - 1;
  double cq[KN_MAXK], iq[KN_MAXK], ph_e = 0, ph_i = 0, pmq = 0, it = 0;
#pragma unroll
  for (int k = 0; k < KN_MAXK; ++k) { cq[k] = 0.0; iq[k] = 0.0; }
#pragma unroll
  for (int bb = 0; bb < NF; ++bb) {
    const double N = qN[q * NF + bb];
    const Rec& o = f.cell_side ? f.pi[bb] : f.pe[bb];
#pragma unroll
    for (int k = 0; k < KN_MAXK; ++k) { cq[k] += N * o.c[k]; iq[k] += N * f.Ik[bb][k]; }
    ph_e += N * f.pe[bb].phi; ph_i += N * f.pi[bb].phi;
    pmq += N * f.pm[bb]; it += N * f.It[bb];
  }
  double wq;
  if constexpr (NF == 4) {
    // surface Jacobian of the bilinear facet at this point
    double ux = 0, uy = 0, uz = 0, vx = 0, vy = 0, vz = 0;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const double da = qdN[(q * 4 + bb) * 2], db = qdN[(q * 4 + bb) * 2 + 1];
      ux += da * f.pe[bb].x; uy += da * f.pe[bb].y; uz += da * f.pe[bb].z;
      vx += db * f.pe[bb].x; vy += db * f.pe[bb].y; vz += db * f.pe[bb].z;
    }
    const double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    wq = qw[q] * sqrt(nx * nx + ny * ny + nz * nz);
  } else {
    wq = qw[q] * f.meas * (NF == 2 ? 1.0 : 2.0);  // reference measure 1 (interval), 1/2 (triangle)
  }
  double asum = 0.0;      // sum over ALL K ions (knpWeakForm.py:97); az2D = 0 beyond K
#pragma unroll
  for (int k = 0; k < KN_MAXK; ++k) asum += f.so->az2D[k] * cq[k];
  const double jump = ph_i - ph_e;
  const double rasum = kn_rcp(asum);
#pragma unroll
  for (int k = 0; k < KN_MAXK - 1; ++k) {
    fk[k] = 0.0;
    if (k < KS) {
      const double al = f.so->az2D[k] * cq[k] * rasum;
      const double Cc = al * C.C_M * kn_rcp(C.F * C.z[k] * C.dt);
      double gr = pmq - C.dt * kn_rcp(C.C_M * al) * iq[k];
      if (splitting) gr += (C.dt / C.C_M) * it;
      fk[k] = wq * f.sgn * (Cc * gr - Cc * jump);
    }
  }
}

// The same point for the early form: cw[k] = w sgn C_k (the factor of -[phi]) and pg[k] = w sgn C_k g_k.
template <int NF>
__device__ __forceinline__ void facet_point_split(const FacetData<NF>& f, const KnConsts& C, int q, const double* qw,
                                                  const double* qN, const double* qdN, int splitting,
                                                  double (&cw)[KN_MAXK - 1], double (&pg)[KN_MAXK - 1]) {
  const int KS = C.K - 1;
  double cq[KN_MAXK], iq[KN_MAXK], pmq = 0, it = 0;
#pragma unroll
  for (int k = 0; k < KN_MAXK; ++k) { cq[k] = 0.0; iq[k] = 0.0; }
#pragma unroll
  for (int bb = 0; bb < NF; ++bb) {
    const double N = qN[q * NF + bb];
    const Rec& o = f.cell_side ? f.pi[bb] : f.pe[bb];
#pragma unroll
    for (int k = 0; k < KN_MAXK; ++k) { cq[k] += N * o.c[k]; iq[k] += N * f.Ik[bb][k]; }
    pmq += N * f.pm[bb]; it += N * f.It[bb];
  }
  double wq;
  if constexpr (NF == 4) {
    double ux = 0, uy = 0, uz = 0, vx = 0, vy = 0, vz = 0;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) {
      const double da = qdN[(q * 4 + bb) * 2], db = qdN[(q * 4 + bb) * 2 + 1];
      ux += da * f.pe[bb].x; uy += da * f.pe[bb].y; uz += da * f.pe[bb].z;
      vx += db * f.pe[bb].x; vy += db * f.pe[bb].y; vz += db * f.pe[bb].z;
    }
    const double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    wq = qw[q] * sqrt(nx * nx + ny * ny + nz * nz);
  } else {
    wq = qw[q] * f.meas * (NF == 2 ? 1.0 : 2.0);
  }
  double asum = 0.0;
#pragma unroll
  for (int k = 0; k < KN_MAXK; ++k) asum += f.so->az2D[k] * cq[k];
#pragma unroll
  for (int k = 0; k < KN_MAXK - 1; ++k) {
    cw[k] = 0.0; pg[k] = 0.0;
    if (k < KS) {
      const double al = f.so->az2D[k] * cq[k] / asum;
      const double Cc = al * C.C_M / (C.F * C.z[k] * C.dt);
      double gr = pmq - C.dt / (C.C_M * al) * iq[k];
      if (splitting) gr += (C.dt / C.C_M) * it;
      cw[k] = wq * f.sgn * Cc;
      pg[k] = cw[k] * gr;
    }
  }
}

// Stand-alone form (diagnostics, KNPEMI_OPT_FUSE_MEMBRANE = 0): one thread per (facet, side) tests the integrand
// against all NF facet functions and writes NF x (K - 1) partial integrals to gam_e.
// KN_MEM_LQ adjacent lanes share one (facet, side): each takes every KN_MEM_LQ-th quadrature point and the partial
// integrals meet in a shuffle reduction.  The membrane is a 2-D set: with one thread per (facet, side) a config-2 launch
// has 46 workgroups whose threads walk 12 points of ~150 instructions one after the other; spreading the points
// turns that latency chain into four times as many, four times shorter waves.
#ifndef KN_MEM_LQ
#define KN_MEM_LQ 4
#endif
// One workgroup's share of the facet integrals; `bid` = its index among the `nb` workgroups that do this work (after the
// XCD-aware remap: consecutive facets share vertex records, so each XCD takes one contiguous run of facets and fetches a
// record once instead of once per XCD that meets it: 2.5 MB instead of ... of HBM-side traffic at config 2).
template <int NF>
__device__ __forceinline__ void knp_membrane_block(const KnDev& D, const KnConsts& C, int splitting, int bid,
                                                   const double* __restrict__ phi_x, double* qt) {
  const int nq = D.nq_gamma;
  const int ntab = nq * (1 + NF + (NF == 4 ? 2 * NF : 0));
  for (int i = threadIdx.x; i < ntab; i += blockDim.x) qt[i] = D.qtab[i];
  __syncthreads();
  const double* qw = qt;
  const double* qN = qt + nq;
  const double* qdN = qt + nq * (1 + NF);
  const int gt = bid * blockDim.x + threadIdx.x;
  const int t = gt / KN_MEM_LQ, lq = gt % KN_MEM_LQ;
  // lanes past the last (facet, side) repeat the last one and drop their results: the shuffles below see whole groups
  const bool live = t < 2 * D.nftot;
  const int tt = live ? t : 2 * D.nftot - 1;
  const int KS = C.K - 1;
  const int fg = tt >> 1;
  const bool cell_side = tt & 1;
  const int ms = D.fmodel[fg];
  const int* pos = D.gam_pos + (size_t)tt * NF;   // entry of (facet, side, local vertex) in the membrane row lists
  double acc[NF][KN_MAXK - 1];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int k = 0; k < KN_MAXK - 1; ++k) acc[a][k] = 0.0;
  if (ms >= 0) {
    FacetData<NF> f;
    load_facet<NF>(D, C, fg, cell_side, ms, f, phi_x);
    for (int q = lq; q < nq; q += KN_MEM_LQ) {
      double fk[KN_MAXK - 1];
      facet_point<NF>(f, C, q, qw, qN, qdN, splitting, fk);
#pragma unroll
      for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int k = 0; k < KN_MAXK - 1; ++k) acc[a][k] += qN[q * NF + a] * fk[k];
    }
  }
#pragma unroll
  for (int m = 1; m < KN_MEM_LQ; m <<= 1)
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int k = 0; k < KN_MAXK - 1; ++k) acc[a][k] += __shfl_xor(acc[a][k], m);
  if (live && lq == 0) {
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int k = 0; k < KN_MAXK - 1; ++k)
        if (k < KS) D.gam_e[(size_t)KS * pos[a] + k] = acc[a][k];
  }
}

template <int NF>
__global__ __launch_bounds__(256) void knp_membrane_kernel(KnDev D, const KnConsts* __restrict__ Cp, int splitting) {
  extern __shared__ double qt[];
  knp_membrane_block<NF>(D, *Cp, splitting, logical_block(blockIdx.x, gridDim.x), nullptr, qt);
}

// The write-back of the potential and the facet integrals of b_knp in ONE launch: the first `nbm` workgroups integrate the
// membrane facets with the potential taken from the solution vector x itself (the mean that the write-back removes cancels
// in the jump phi_i - phi_e), the others write phi = x - mean into the vertex records (mean = inv_n x the sum of the np
// partial sums in `part`; np = 0: no shift, e.g. a pasted solution).  This takes the facet kernel out of the chain
// EMI solve -> KNP assembly: knpemi_assemble_knp finds the integrals in gam_e (knpemi_handle::gam_valid).
template <int NF>
__global__ __launch_bounds__(256) void emi_writeback_membrane_kernel(KnDev D, const KnConsts* __restrict__ Cp, int splitting,
                                                                     int nbm, const double* __restrict__ x, int n,
                                                                     const double* __restrict__ part, int np, double inv_n,
                                                                     double* __restrict__ mean_out) {
  extern __shared__ double qt[];
  if ((int)blockIdx.x < nbm) {
    knp_membrane_block<NF>(D, *Cp, splitting, logical_block(blockIdx.x, nbm), x, qt);
    return;
  }
  const int vb = blockIdx.x - nbm, nvb = gridDim.x - nbm;
  int i = vb * 256 + threadIdx.x;
  double xi = i < n ? x[i] : 0.0;
  double mean = 0.0;
  if (np > 0) {          // the <= 1 024 partial sums, all of a thread's loads in flight; fixed order: the same bits in every block
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    const int t = threadIdx.x;
    if (t < np) v0 = part[t];
    if (t + 256 < np) v1 = part[t + 256];
    if (t + 512 < np) v2 = part[t + 512];
    if (t + 768 < np) v3 = part[t + 768];
    double sum = ((v0 + v1) + v2) + v3;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    __shared__ double sh[4];
    if ((t & 63) == 0) sh[t >> 6] = sum;
    __syncthreads();
    mean = ((sh[0] + sh[1]) + (sh[2] + sh[3])) * inv_n;
    if (vb == 0 && t == 0 && mean_out) *mean_out = mean;
  }
  while (i < n) {
    D.VR[(size_t)i * KN_REC + 7] = xi - mean;
    i += nvb * 256;
    if (i < n) xi = x[i];
  }
}

// Early form (KNPEMI_MEMBRANE_EARLY): everything of the integrand that does not depend on the potential is integrated
// BEFORE the EMI solve, beside it on the auxiliary stream.  With  [phi](q) = sum_b N_b(q) [phi]_b  the integrand
//   w sgn (C_k g_k - C_k [phi])  tested with N_a  splits into
//   P[a][k] = sum_q N_a w sgn C_k g_k      and      W[a][b][k] = sum_q N_a N_b w sgn C_k,
// so that the KNP row kernel only has to form  P[a][k] - sum_b W[a][b][k] [phi]_b  from the potential just solved for:
// the degree-6 quadrature leaves the chain  EMI solve -> KNP assembly.  (1 + NF) (K - 1) doubles per entry in gpre.
template <int NF>
__global__ __launch_bounds__(256) void knp_membrane_pre_kernel(KnDev D, const KnConsts* __restrict__ Cp, int splitting) {
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
  const int KS = C.K - 1, GS = (1 + NF) * KS;
  const int fg = t >> 1;
  const bool cell_side = t & 1;
  const int ms = D.fmodel[fg];
  const int* pos = D.gam_pos + (size_t)t * NF;
  if (ms < 0) {
#pragma unroll
    for (int a = 0; a < NF; ++a)
      for (int j = 0; j < GS; ++j) D.gpre[(size_t)GS * pos[a] + j] = 0.0;
    return;
  }
  FacetData<NF> f;
  load_facet<NF>(D, C, fg, cell_side, ms, f);
  double P[NF][KN_MAXK - 1], W[NF][NF][KN_MAXK - 1];
#pragma unroll
  for (int a = 0; a < NF; ++a)
#pragma unroll
    for (int k = 0; k < KN_MAXK - 1; ++k) {
      P[a][k] = 0.0;
#pragma unroll
      for (int b = 0; b < NF; ++b) W[a][b][k] = 0.0;
    }
  for (int q = 0; q < nq; ++q) {
    double cw[KN_MAXK - 1], pg[KN_MAXK - 1];
    facet_point_split<NF>(f, C, q, qw, qN, qdN, splitting, cw, pg);
#pragma unroll
    for (int a = 0; a < NF; ++a) {
      const double Na = qN[q * NF + a];
#pragma unroll
      for (int k = 0; k < KN_MAXK - 1; ++k) {
        P[a][k] += Na * pg[k];
#pragma unroll
        for (int b = 0; b < NF; ++b) W[a][b][k] += Na * qN[q * NF + b] * cw[k];
      }
    }
  }
#pragma unroll
  for (int a = 0; a < NF; ++a) {
    double* g = D.gpre + (size_t)GS * pos[a];
#pragma unroll
    for (int k = 0; k < KN_MAXK - 1; ++k)
      if (k < KS) {
        g[k] = P[a][k];
#pragma unroll
        for (int b = 0; b < NF; ++b) g[(1 + b) * KS + k] = W[a][b][k];
      }
  }
}

// Fused form: the KNP row kernels call this before their barrier.  The membrane entries (row, facet, local vertex a)
// of a row block are contiguous [e0, e0 + ne); each thread integrates the entries i = tid, tid + 256, ... against the
// ONE facet function of its entry -- the same sums, in the same order, as the stand-alone kernel forms for that
// function -- and leaves the K - 1 integrals in LDS for the row's lane.  No separate launch, nothing through HBM.
template <int NF, int KS>
__device__ __forceinline__ void membrane_entries_to_lds(const KnDev& D, const KnConsts& C, int e0, int ne, bool cell_side,
                                                        int splitting, int tid, double* gam) {
  const int nq = D.nq_gamma;
  const double* qw = D.qtab;
  const double* qN = D.qtab + nq;
  const double* qdN = D.qtab + nq * (1 + NF);
  for (int i = tid; i < ne; i += KN_BLOCK) {
    const int e = e0 + i;
    const int ms = D.me_model[e];
    double acc[KN_MAXK - 1];
#pragma unroll
    for (int k = 0; k < KN_MAXK - 1; ++k) acc[k] = 0.0;
    if (ms >= 0) {
      const int ent = D.mentry[e];
      const int fg = ent >> 3, a = ent & 7;
      FacetData<NF> f;
      load_facet<NF>(D, C, fg, cell_side, ms, f);
      for (int q = 0; q < nq; ++q) {
        double fk[KN_MAXK - 1];
        facet_point<NF>(f, C, q, qw, qN, qdN, splitting, fk);
        const double Na = qN[q * NF + a];
#pragma unroll
        for (int k = 0; k < KN_MAXK - 1; ++k) acc[k] += Na * fk[k];
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) gam[(size_t)i * KS + k] = acc[k];
  }
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

// buf[i] = vec[idx[i]] (gather) or vec[idx[i]] = buf[i] (scatter): pack / unpack of the halo of a solver vector
__global__ void vec_index_kernel(double* __restrict__ vec, const int* __restrict__ idx, int n, double* __restrict__ buf,
                                 int gather) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (gather) buf[i] = vec[idx[i]];
  else vec[idx[i]] = buf[i];
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
    const int KS = C.K - 1;
    double* rec = D.VR + (size_t)i * KN_REC;  // the phi component is left untouched
    double el = C.sc[s].rho_term;
    for (int k = 0; k < KS; ++k) {
      const double ck = D.csol[(size_t)k * D.Ntot + i];
      el += C.elim_coef[k] * ck;
      rec[KN_CSLOT(k)] = ck;
    }
    rec[KN_CSLOT(KS)] = el;
  }
  if (i < D.NQtot) D.phiM[i] = D.VR[(size_t)D.q2i[i] * KN_REC + 7] - D.VR[(size_t)D.q2e[i] * KN_REC + 7];
}

// Write-back of the KNP solve fused with update_pde_variables (KNPEMI_OPT_FUSE_UPDATE): x holds the solution in the
// reference's block order [sub-domain][ion][vertex] (pdeSolver.py:117); same arithmetic as update_pde_kernel.
__global__ void knp_writeback_update_kernel(KnDev D, const KnConsts* __restrict__ Cp, const double* __restrict__ x) {
  const KnConsts& C = *Cp;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.Ntot) {
    int s = 0;
    for (int t = 1; t < C.n_sub; ++t) s += i >= C.voff[t];
    const int v0 = C.voff[s], nv = C.voff[s + 1] - v0;
    const int KS = C.K - 1;
    const size_t xb = (size_t)KS * v0 + (size_t)(i - v0);
    double* rec = D.VR + (size_t)i * KN_REC;
    double el = C.sc[s].rho_term;
    for (int k = 0; k < KS; ++k) {
      const double ck = x[xb + (size_t)k * nv];
      D.csol[(size_t)k * D.Ntot + i] = ck;
      el += C.elim_coef[k] * ck;
      rec[KN_CSLOT(k)] = ck;
    }
    rec[KN_CSLOT(KS)] = el;
  }
  if (i < D.NQtot) D.phiM[i] = D.VR[(size_t)D.q2i[i] * KN_REC + 7] - D.VR[(size_t)D.q2e[i] * KN_REC + 7];
}

// Forward-halo pack / unpack (owner -> ghost copies of dof fields between GPUs).
__global__ void halo_kernel(KnDev D, int kind, int pack, const int* __restrict__ idx, int n, int n_slots,
                            double* __restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int g = idx[i];
  if (kind == 0) {   // record slots 3..7: every concentration (K <= 4) and phi
    double* rec = D.VR + (size_t)g * KN_REC + 3;
    double* b = buf + (size_t)i * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) {
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
template <int NF, int LPR>
__global__ __launch_bounds__(256) void emi_membrane_rhs_kernel(KnDev D, const KnConsts* __restrict__ Cp, int splitting) {
  const KnConsts& C = *Cp;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int mm = t / LPR, sub = t % LPR;
  const bool live = mm < D.M;                  // lanes past the last row repeat it: the shuffles see whole groups
  const int m = live ? mm : D.M - 1;
  const int g = D.mrow[m];
  const bool cell_side = g >= C.voff[1];
  double gam = 0.0;
  for (int e = D.mptr[m] + sub; e < D.mptr[m + 1]; e += LPR) {
    const int ms = D.me_model[e];
    if (ms < 0) continue;
    gam += (cell_side ? 1.0 : -1.0) * C.C_phi * emi_membrane_entry_rhs<NF>(D, C, e, ms, splitting);
  }
#pragma unroll
  for (int k = 1; k < LPR; k <<= 1) gam += __shfl_xor(gam, k);
  if (live && sub == 0) D.b_emi[g] = D.b_emi[g] + gam;
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

template <int GDIM>
static int launch_emi_v2(knpemi_handle* h, int want_p, int split) {
  const KnDev& D = h->dev;
  const int acc_n = (h->lds_doubles_emi + 1) & ~1, rec_n = h->lds_uniq_max;
  const bool ut = GDIM == 3 && h->tet_uniform && D.tet_tab;
  const size_t lds = ((size_t)acc_n + (ut ? 2 : 5) * (size_t)rec_n + 1 + (ut ? KN_TET_TAB : 0)) * sizeof(double) + 2 * (size_t)h->lds_doubles_knp + 16;
  if (lds > 160 * 1024) { kn_set_error("EMI row block does not fit in 160 KiB of LDS"); return KNPEMI_EINVAL; }
  dim3 grid(D.nblocks), block(KN_BLOCK);
  int rc = 0;
#define KN_CASE2(L, U)                                                                              \
    if ((rc = set_lds_limit(emi_rows_v2<GDIM, L, U>, lds))) return rc;                              \
    {                                                                                               \
      KnProfScope prof(h, KNPEMI_K_EMI_ROWS);                                                       \
      hipLaunchKernelGGL((emi_rows_v2<GDIM, L, U>), grid, block, lds, h->cur, D, h->d_consts, acc_n, \
                         rec_n, want_p, split);                                                     \
    }
#define KN_CASE(L)                                                                                  \
  case L:                                                                                           \
    if constexpr (GDIM == 3) { if (ut) { KN_CASE2(L, true) } else { KN_CASE2(L, false) } }          \
    else { KN_CASE2(L, false) }                                                                     \
    break;
  switch (h->lpr) { KN_CASE(1) KN_CASE(2) KN_CASE(4) KN_CASE(8) default: kn_set_error("bad lanes-per-row"); return KNPEMI_EINVAL; }
#undef KN_CASE
#undef KN_CASE2
#ifdef KN_ROW_STAMPS
  static int launches = 0;
  if (++launches % 16 == 0) {
    static std::vector<unsigned long long> acc(1024 * 8);
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(acc.data(), HIP_SYMBOL(row_stamp_acc), acc.size() * sizeof(unsigned long long));
    const double blocks = 16.0 * D.nblocks;
    double ph[5] = {0}, total = 0.0;
    for (int sl = 0; sl < 1024; ++sl) for (int i = 0; i < 5; ++i) ph[i] += (double)acc[sl * 8 + i];
    for (int i = 0; i < 5; ++i) total += ph[i];
    fprintf(stderr, "[row stamps] emi_rows_v2, cycles per workgroup, 16 launches of %d workgroups: total %.1f\n", D.nblocks, total / blocks);
    for (int i = 0; i < 5; ++i) fprintf(stderr, "[row stamps]   phase %d: %8.1f\n", i, ph[i] / blocks);
    std::fill(acc.begin(), acc.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(row_stamp_acc), acc.data(), acc.size() * sizeof(unsigned long long));
  }
#endif
  return check_launch("emi_rows_v2");
}

template <int GDIM>
static int launch_knp_v2(knpemi_handle* h, int split, int pre) {
  const KnDev& D = h->dev;
  const int acc_n = (h->lds_doubles_knp + 1) & ~1, rec_n = h->lds_uniq_max;
  const int KS = h->K - 1;
  const int gam_n = h->fuse_membrane ? std::max(1, h->lds_gam_max) : 0;
  const bool ut = GDIM == 3 && h->tet_uniform && D.tet_tab;
  const size_t lds = ((size_t)KS * acc_n + (ut ? KS + 1 : 4 + KS) * (size_t)rec_n + 1 + (ut ? KN_TET_TAB : 0) + (size_t)KS * gam_n) * sizeof(double)
                     + 2 * (size_t)h->lds_doubles_knp + 16;
  if (lds > 160 * 1024) { kn_set_error("KNP row block does not fit in 160 KiB of LDS"); return KNPEMI_EINVAL; }
  dim3 grid(D.nblocks), block(KN_BLOCK);
  int rc = 0;
  const int mem = pre ? 2 : (gam_n > 0 ? 1 : 0);
#define KN_CASE2(L, S, M, U)                                                                        \
    if ((rc = set_lds_limit(knp_rows_v2<GDIM, L, S, M, U>, lds))) return rc;                        \
    KnProfScope prof(h, KNPEMI_K_KNP_ROWS);                                                         \
    hipLaunchKernelGGL((knp_rows_v2<GDIM, L, S, M, U>), grid, block, lds, h->cur, D, h->d_consts, acc_n, rec_n, gam_n, split); \
    return check_launch("knp_rows_v2");
#define KN_CASE(L, S, M)                                                                            \
  if (h->lpr == L && KS == S && mem == M) {                                                         \
    if constexpr (GDIM == 3) { if (ut) { KN_CASE2(L, S, M, true) } else { KN_CASE2(L, S, M, false) } } \
    else { KN_CASE2(L, S, M, false) }                                                               \
  }
  // lanes per row: 2 (triangles) or 4 by default, KNPEMI_LPR for experiments; K - 1 = 1..3 solved ions; the optional
  // membrane paths (fused, early) for the default lanes-per-row only
  KN_CASE(2, 2, 0) KN_CASE(4, 2, 0) KN_CASE(1, 2, 0) KN_CASE(8, 2, 0) KN_CASE(2, 1, 0) KN_CASE(4, 1, 0) KN_CASE(2, 3, 0) KN_CASE(4, 3, 0)
  KN_CASE(2, 2, 1) KN_CASE(4, 2, 1) KN_CASE(2, 1, 1) KN_CASE(4, 1, 1) KN_CASE(2, 3, 1) KN_CASE(4, 3, 1)
  KN_CASE(2, 2, 2) KN_CASE(4, 2, 2) KN_CASE(2, 1, 2) KN_CASE(4, 1, 2) KN_CASE(2, 3, 2) KN_CASE(4, 3, 2)
#undef KN_CASE
#undef KN_CASE2
  kn_set_error("knp_rows: unsupported lanes-per-row / ion-count / membrane-option combination");
  return KNPEMI_EINVAL;
}

static int launch_emi_hex_v2(knpemi_handle* h, int want_p, int split) {
  const KnDev& D = h->dev;
  const int acc_n = (h->lds_doubles_emi + 1) & ~1, rec_n = h->lds_uniq_max;
  const int geo = h->hex_uniform ? 2 : (h->hex_affine ? 1 : 0);
  const size_t lds = ((size_t)acc_n + (geo == 2 ? 2 : 5) * (size_t)rec_n + 1) * sizeof(double) + 2 * (size_t)h->lds_doubles_knp + 16;
  if (lds > 160 * 1024) { kn_set_error("EMI row block does not fit in 160 KiB of LDS"); return KNPEMI_EINVAL; }
  dim3 grid(D.nblocks), block(KN_BLOCK);
  int rc = 0;
#define KN_CASE2(L, GEO)                                                                            \
    if ((rc = set_lds_limit(emi_rows_hex_v2<L, GEO>, lds))) return rc;                              \
    {                                                                                               \
      KnProfScope prof(h, KNPEMI_K_EMI_ROWS);                                                       \
      hipLaunchKernelGGL((emi_rows_hex_v2<L, GEO>), grid, block, lds, h->cur, D, h->d_consts, acc_n, \
                         rec_n, want_p, split, h->hex_geo);                                         \
    }
#define KN_CASE(L)                                                                                  \
  case L:                                                                                           \
    if (geo == 2) { KN_CASE2(L, 2) } else if (geo == 1) { KN_CASE2(L, 1) } else { KN_CASE2(L, 0) }  \
    break;
  switch (h->lpr) { KN_CASE(1) KN_CASE(2) KN_CASE(4) KN_CASE(8) default: kn_set_error("bad lanes-per-row"); return KNPEMI_EINVAL; }
#undef KN_CASE
#undef KN_CASE2
  return check_launch("emi_rows_hex_v2");
}

static int launch_knp_hex_v2(knpemi_handle* h, int split, int pre) {
  const KnDev& D = h->dev;
  const int acc_n = (h->lds_doubles_knp + 1) & ~1, rec_n = h->lds_uniq_max;
  const int KS = h->K - 1;
  const int gam_n = h->fuse_membrane ? std::max(1, h->lds_gam_max) : 0;
  const int geo = h->hex_uniform ? 2 : (h->hex_affine ? 1 : 0);
  const size_t lds = ((size_t)KS * acc_n + (geo == 2 ? KS + 1 : 4 + KS) * (size_t)rec_n + 1 + (size_t)KS * gam_n) * sizeof(double)
                     + 2 * (size_t)h->lds_doubles_knp + 16;
  if (lds > 160 * 1024) { kn_set_error("KNP row block does not fit in 160 KiB of LDS"); return KNPEMI_EINVAL; }
  dim3 grid(D.nblocks), block(KN_BLOCK);
  int rc = 0;
  const int mem = pre ? 2 : (gam_n > 0 ? 1 : 0);
#define KN_CASE(L, GEO, S, M)                                                                       \
  if (h->lpr == L && geo == GEO && KS == S && mem == M) {                                           \
    if ((rc = set_lds_limit(knp_rows_hex_v2<L, GEO, S, M>, lds))) return rc;                        \
    KnProfScope prof(h, KNPEMI_K_KNP_ROWS);                                                         \
    hipLaunchKernelGGL((knp_rows_hex_v2<L, GEO, S, M>), grid, block, lds, h->cur, D, h->d_consts, acc_n, rec_n, gam_n, split, h->hex_geo); \
    return check_launch("knp_rows_hex_v2");                                                         \
  }
#define KN_GEO(L, S, M) KN_CASE(L, 2, S, M) KN_CASE(L, 1, S, M) KN_CASE(L, 0, S, M)
  KN_GEO(4, 2, 0) KN_GEO(2, 2, 0) KN_GEO(8, 2, 0) KN_GEO(1, 2, 0) KN_GEO(4, 1, 0) KN_GEO(4, 3, 0)
  KN_GEO(4, 2, 1) KN_GEO(4, 1, 1) KN_GEO(4, 3, 1)
  KN_GEO(4, 2, 2) KN_GEO(4, 1, 2) KN_GEO(4, 3, 2)
#undef KN_GEO
#undef KN_CASE
  kn_set_error("knp_rows (hexahedra): unsupported lanes-per-row / ion-count / membrane-option combination");
  return KNPEMI_EINVAL;
}

int kn_launch_emi_rows(knpemi_handle* h, int flags) {
  const KnDev& D = h->dev;
  if (D.nblocks == 0) return KNPEMI_OK;
  const int want_p = (flags & KNPEMI_WANT_P) ? 1 : 0;
  const int split = ((flags & KNPEMI_NO_SPLITTING) ? 0 : 1) | ((flags & KNPEMI_SKIP_MEMBRANE_RHS) ? 2 : 0);
  if (h->NV == 8) return launch_emi_hex_v2(h, want_p, split);
  return h->gdim == 2 ? launch_emi_v2<2>(h, want_p, split) : launch_emi_v2<3>(h, want_p, split);
}

int kn_launch_knp_rows(knpemi_handle* h, int flags) {
  const KnDev& D = h->dev;
  if (D.nblocks == 0) return KNPEMI_OK;
  const int split = (flags & KNPEMI_NO_SPLITTING) ? 0 : 1;
  const int pre = (flags & KNPEMI_MEMBRANE_EARLY) ? 1 : 0;   // membrane integrals from knp_membrane_pre_kernel
  if (h->NV == 8) return launch_knp_hex_v2(h, split, pre);
  return h->gdim == 2 ? launch_knp_v2<2>(h, split, pre) : launch_knp_v2<3>(h, split, pre);
}

int kn_launch_membrane_mass(knpemi_handle* h, int n_entries, const int* d_entry_row, double* d_out) {
  if (n_entries == 0) return KNPEMI_OK;
  const KnDev& D = h->dev;
  dim3 grid((n_entries + 255) / 256), block(256);
  const int v_cells = h->voff[1];
  if (h->NF == 2) hipLaunchKernelGGL(membrane_mass_kernel<2>, grid, block, 0, h->stream, D, n_entries, v_cells, d_entry_row, d_out);
  else if (h->NF == 3) hipLaunchKernelGGL(membrane_mass_kernel<3>, grid, block, 0, h->stream, D, n_entries, v_cells, d_entry_row, d_out);
  else hipLaunchKernelGGL(membrane_mass_kernel<4>, grid, block, 0, h->stream, D, n_entries, v_cells, d_entry_row, d_out);
  return check_launch("membrane_mass_kernel");
}

int kn_launch_knp_membrane(knpemi_handle* h, int flags) {
  const KnDev& D = h->dev;
  if (D.nftot == 0) return KNPEMI_OK;
  const int split = (flags & KNPEMI_NO_SPLITTING) ? 0 : 1;
  const int NF = h->NF;
  const size_t lds = (size_t)D.nq_gamma * (1 + NF + (NF == 4 ? 2 * NF : 0)) * sizeof(double);
  dim3 grid((2 * (size_t)D.nftot * KN_MEM_LQ + 255) / 256), block(256);
  KnProfScope prof(h, KNPEMI_K_KNP_MEMBRANE);
  if (NF == 2) hipLaunchKernelGGL((knp_membrane_kernel<2>), grid, block, lds, h->cur, D, h->d_consts, split);
  else if (NF == 3) hipLaunchKernelGGL((knp_membrane_kernel<3>), grid, block, lds, h->cur, D, h->d_consts, split);
  else hipLaunchKernelGGL((knp_membrane_kernel<4>), grid, block, lds, h->cur, D, h->d_consts, split);
  return check_launch("knp_membrane_kernel");
}

// phi <- x - mean and the membrane-facet integrals of b_knp for that potential, one launch (see the kernel).  np = 0: x is
// written as it is.  Marks the integrals in gam_e as current.
int kn_launch_emi_writeback_membrane(knpemi_handle* h, const double* x, const double* part, int np, double inv_n, double* mean_out) {
  const KnDev& D = h->dev;
  const int n = D.Ntot;
  if (n == 0) return KNPEMI_OK;
  const int split = (h->emi_flags & KNPEMI_NO_SPLITTING) ? 0 : 1;
  const int NF = h->NF;
  const size_t lds = (size_t)D.nq_gamma * (1 + NF + (NF == 4 ? 2 * NF : 0)) * sizeof(double);
  const int nbm = (int)((2 * (size_t)D.nftot * KN_MEM_LQ + 255) / 256);
  const int nbv = std::max(1, std::min(1024, (n + 255) / 256));
  dim3 grid(nbm + nbv), block(256);
  KnProfScope prof(h, KNPEMI_K_KNP_MEMBRANE);
  if (NF == 2) hipLaunchKernelGGL((emi_writeback_membrane_kernel<2>), grid, block, lds, h->cur, D, h->d_consts, split, nbm, x, n, part, np, inv_n, mean_out);
  else if (NF == 3) hipLaunchKernelGGL((emi_writeback_membrane_kernel<3>), grid, block, lds, h->cur, D, h->d_consts, split, nbm, x, n, part, np, inv_n, mean_out);
  else hipLaunchKernelGGL((emi_writeback_membrane_kernel<4>), grid, block, lds, h->cur, D, h->d_consts, split, nbm, x, n, part, np, inv_n, mean_out);
  h->gam_valid = D.nftot > 0;
  h->gam_split = split;
  return check_launch("emi_writeback_membrane_kernel");
}

int kn_launch_knp_membrane_pre(knpemi_handle* h, int flags) {
  const KnDev& D = h->dev;
  if (D.nftot == 0) return KNPEMI_OK;
  const int split = (flags & KNPEMI_NO_SPLITTING) ? 0 : 1;
  const int NF = h->NF;
  const size_t lds = (size_t)D.nq_gamma * (1 + NF + (NF == 4 ? 2 * NF : 0)) * sizeof(double);
  dim3 grid((2 * D.nftot + 255) / 256), block(256);
  KnProfScope prof(h, KNPEMI_K_KNP_MEMBRANE);
  if (NF == 2) hipLaunchKernelGGL((knp_membrane_pre_kernel<2>), grid, block, lds, h->cur, D, h->d_consts, split);
  else if (NF == 3) hipLaunchKernelGGL((knp_membrane_pre_kernel<3>), grid, block, lds, h->cur, D, h->d_consts, split);
  else hipLaunchKernelGGL((knp_membrane_pre_kernel<4>), grid, block, lds, h->cur, D, h->d_consts, split);
  return check_launch("knp_membrane_pre_kernel");
}

int kn_launch_emi_membrane_rhs(knpemi_handle* h, int flags) {
  const KnDev& D = h->dev;
  if (D.M == 0) return KNPEMI_OK;
  const int split = (flags & KNPEMI_NO_SPLITTING) ? 0 : 1;
  const int lpr = h->lpr;      // the same lane groups as the row kernels, so that both forms produce the same bits
  dim3 grid(((size_t)D.M * lpr + 255) / 256), block(256);
  KnProfScope prof(h, KNPEMI_K_EMI_MEMBRANE);
#define KN_ROBIN(NFV, L) hipLaunchKernelGGL((emi_membrane_rhs_kernel<NFV, L>), grid, block, 0, h->stream, D, h->d_consts, split)
#define KN_ROBIN_L(NFV) \
  switch (lpr) { case 1: KN_ROBIN(NFV, 1); break; case 2: KN_ROBIN(NFV, 2); break; case 8: KN_ROBIN(NFV, 8); break; \
                 default: KN_ROBIN(NFV, 4); }
  if (h->NF == 2) { KN_ROBIN_L(2) } else if (h->NF == 3) { KN_ROBIN_L(3) } else { KN_ROBIN_L(4) }
#undef KN_ROBIN_L
#undef KN_ROBIN
  return check_launch("emi_membrane_rhs_kernel");
}

int kn_launch_update_pde(knpemi_handle* h) {
  h->gam_valid = false;      // the concentrations (and phi_M) the facet integrals were formed with change
  const KnDev& D = h->dev;
  const int n = std::max(D.Ntot, D.NQtot);
  if (n == 0) return KNPEMI_OK;
  KnProfScope prof(h, KNPEMI_K_UPDATE);
  hipLaunchKernelGGL(update_pde_kernel, dim3((n + 255) / 256), dim3(256), 0, h->cur, D, h->d_consts);
  return check_launch("update_pde_kernel");
}

int kn_launch_knp_writeback_update(knpemi_handle* h, const double* x) {
  h->gam_valid = false;      // the concentrations (and phi_M) the facet integrals were formed with change
  const KnDev& D = h->dev;
  const int n = std::max(D.Ntot, D.NQtot);
  if (n == 0) return KNPEMI_OK;
  KnProfScope prof(h, KNPEMI_K_UPDATE);
  hipLaunchKernelGGL(knp_writeback_update_kernel, dim3((n + 255) / 256), dim3(256), 0, h->cur, D, h->d_consts, x);
  return check_launch("knp_writeback_update_kernel");
}

int kn_launch_halo(knpemi_handle* h, int kind, int pack, const int32_t* idx, int n, double* buf) {
  if (!pack) h->gam_valid = false;
  if (n == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(halo_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->dev, kind, pack, idx, n,
                     h->moff[h->n_sub], buf);
  return check_launch("halo_kernel");
}

int kn_launch_field_scatter(knpemi_handle* h, const double* src, double* dst, int n, int dst_stride) {
  if (n == 0) return KNPEMI_OK;
  h->gam_valid = false;
  hipLaunchKernelGGL(scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, src, dst, n, dst_stride);
  return check_launch("scatter_kernel");
}

int kn_launch_field_gather(knpemi_handle* h, const double* src, int src_stride, double* dst, int n) {
  if (n == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(gather_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, src, src_stride, dst, n);
  return check_launch("gather_kernel");
}

int kn_launch_vec_index(knpemi_handle* h, double* vec, const int32_t* idx, int n, double* buf, int gather) {
  if (n == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(vec_index_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, vec, idx, n, buf, gather);
  return check_launch("vec_index_kernel");
}

int kn_launch_trace(knpemi_handle* h, const double* ue, const double* ui, int sub, double* qe, double* qi) {
  const int nq = h->n_q[sub];
  if (nq == 0) return KNPEMI_OK;
  hipLaunchKernelGGL(trace_kernel, dim3((nq + 255) / 256), dim3(256), 0, h->stream, ue, ui,
                     h->dev.q2e, h->dev.q2i, h->qoff[sub], nq, h->voff[sub], qe, qi);
  return check_launch("trace_kernel");
}
