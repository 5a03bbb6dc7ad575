// Shared declarations of the DG variant's kernels: kernels_dg.hip (broken P1 on simplices, the C ABI) and
// kernels_dg_hex.hip (broken Q1 on hexahedra).
#pragma once

#include "knpemi_internal.h"

namespace kn_dg {

#ifndef KN_DG_BLOCK
#define KN_DG_BLOCK 64
#endif
#ifndef KN_DG_ROUND
#define KN_DG_ROUND KN_DG_BLOCK
#endif
constexpr int DG_BLOCK = KN_DG_BLOCK;   // threads (= rows) per workgroup
constexpr int DG_RPITCH = 10;           // doubles per staged record in LDS
constexpr int DG_ROUND = KN_DG_ROUND;   // rows whose image is in LDS at a time

struct DgConsts {
  int n_sub, K;
  double F, psi, C_M, dt, inv_dt, C_phi, gamma;
  double z[KN_MAXK];
  double elim[KN_MAXK];              // -(z_k / z_K)
  double D[KN_MAXSUB][KN_MAXK];
  double kap[KN_MAXSUB][KN_MAXK];    // F psi z_k^2 D_k
  double sig[KN_MAXSUB][KN_MAXK];    // F z_k D_k
  double az2D[KN_MAXSUB][KN_MAXK];   // D_k z_k^2
  double rho_term[KN_MAXSUB];        // -(1 / z_K) rho_z rho^s
};

struct DgDev {
  int n_cell, n_dof, nq;             // nq: membrane nodes
  int nquad;                         // points of the degree-6 membrane rule
  long long nnz;
  double* rec;
  const int* nbr;
  const unsigned* finfo;
  const int* mfid;                   // [n_cell][nv] membrane facet of local facet f (-1)
  const double* box_h;               // box meshes of hexahedra: [n_cell][3] edge lengths along the local axes, else NULL
  const unsigned char* cell_sub;
  const int* rowptr;
  double* A_emi;
  double* b_emi;
  double* A_knp;                     // [K-1][nnz]
  double* b_knp;                     // [K-1][n_dof]
  double* phiM;                      // [nq]
  double* Ich;                       // [KN_MAXK][nq]
  const double* fsrc;                // [K-1][n_dof] or NULL
  const double* qtab;                // weights, then shape values [nquad][nf]
  const int* q2e;
  const int* q2i;
};

struct DofRec {
  double x[3], c[KN_MAXK], phi;
};

__device__ __forceinline__ DofRec load_rec(const double* rec, int dof) {
  const double2* p = reinterpret_cast<const double2*>(rec + (size_t)dof * KN_REC);
  const double2 a = p[0], b = p[1], c = p[2], d = p[3];
  DofRec r;
  r.x[0] = a.x; r.x[1] = a.y; r.x[2] = b.x;
  r.c[3] = b.y; r.c[0] = c.x; r.c[1] = c.y; r.c[2] = d.x; r.phi = d.y;
  return r;
}

// The workgroup's own records are staged in LDS by one coalesced pass (lane t loads record row0 + t) and read from there
// by the NV lanes of each cell; 10-double pitch keeps the 16-byte reads of a wave on distinct banks.

__device__ __forceinline__ void stage_rec(double* lrec, const double* rec, int dof, int t) {
  const double2* p = reinterpret_cast<const double2*>(rec + (size_t)dof * KN_REC);
  double2* q = reinterpret_cast<double2*>(lrec + t * DG_RPITCH);
  const double2 a = p[0], b = p[1], c = p[2], d = p[3];
  q[0] = a; q[1] = b; q[2] = c; q[3] = d;
}

__device__ __forceinline__ DofRec lds_rec(const double* lrec, int t) {
  const double2* p = reinterpret_cast<const double2*>(lrec + t * DG_RPITCH);
  const double2 a = p[0], b = p[1], c = p[2], d = p[3];
  DofRec r;
  r.x[0] = a.x; r.x[1] = a.y; r.x[2] = b.x;
  r.c[3] = b.y; r.c[0] = c.x; r.c[1] = c.y; r.c[2] = d.x; r.phi = d.y;
  return r;
}


// 1 / a and 1 / sqrt(a) from the hardware estimates plus two Newton steps (relative error ~1e-16): a fraction of the
// instruction count of the IEEE division / square-root sequences, and the results are only compared at 1e-10.
__device__ __forceinline__ double fast_rcp(double a) {
  double r = __builtin_amdgcn_rcp(a);
  r = fma(fma(-a, r, 1.0), r, r);
  return fma(fma(-a, r, 1.0), r, r);
}
__device__ __forceinline__ double fast_rsqrt(double a) {
  double y = __builtin_amdgcn_rsq(a);
  y = y * fma(-0.5 * a * y, y, 1.5);
  return y * fma(-0.5 * a * y, y, 1.5);
}

// Workgroups are dealt to the 8 XCDs round robin; give every XCD a contiguous run of cells so that the neighbour
// records a workgroup reads are mostly the ones its XCD's L2 already holds.  Bijection on [0, 8 * chunk).
__device__ __forceinline__ int dg_block_index(int b, int chunk) { return (b & 7) * chunk + (b >> 3); }

}  // namespace kn_dg

// Q1 hexahedra (kernels_dg_hex.hip): one launch each, on `st`; KS = K - 1 solved ions; box != 0: every cell is an
// orthogonal parallelepiped (the closed-frame kernels), 0: general trilinear cells
int kn_dg_hex_launch_emi(hipStream_t st, const kn_dg::DgDev& D, const kn_dg::DgConsts* d_consts, int splitting, int box);
int kn_dg_hex_launch_knp(hipStream_t st, const kn_dg::DgDev& D, const kn_dg::DgConsts* d_consts, int KS, int splitting, int box);
