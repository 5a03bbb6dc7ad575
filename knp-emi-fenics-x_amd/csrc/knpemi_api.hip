// Host side of libknpemi_hip.so: problem builder (adjacency, CSR patterns, scatter slots),
// device memory management, field I/O and the exported C ABI (include/knpemi_hip.h).
//
// The reference builds the equivalent data inside DOLFINx when `LinearProblem` is constructed
// (src/knpemi/pdeSolver.py:46-66,121-139: sparsity patterns, dof maps, entity maps) and inside
// scifem.compute_interface_data (src/knpemi/emiWeakForm.py:39-42).
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "knpemi_internal.h"

static thread_local std::string g_err;
void kn_set_error(const std::string& msg) { g_err = msg; }

extern "C" const char* knpemi_last_error(void) { return g_err.c_str(); }

extern "C" int knpemi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

namespace {

template <class T>
int dev_upload(knpemi_handle* h, const std::vector<T>& v, const T** out) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  KN_HIP(hipMalloc(&p, bytes));
  h->allocs.push_back(p);
  if (!v.empty()) KN_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(p);
  return 0;
}

template <class T>
int dev_zeros(knpemi_handle* h, size_t n, T** out) {
  void* p = nullptr;
  size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  KN_HIP(hipMalloc(&p, bytes));
  h->allocs.push_back(p);
  // zero on the handle's own (non-blocking) stream: a null-stream hipMemset is not ordered with it
  KN_HIP(hipMemsetAsync(p, 0, bytes, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  *out = static_cast<T*>(p);
  return 0;
}

int fail(int code, const std::string& msg) {
  kn_set_error(msg);
  return code;
}

}  // namespace

// Degree-6 rules on the membrane facet (UFL estimates degree 6 for the
// rational KNP coupling integrand; Basix would pick Gauss-Jacobi with 4 points per direction on
// intervals/quadrilaterals and the 12-point Xiao-Gimbutas rule on triangles).  Layout: nq weights
// (reference-cell measure included), nq*NF shape values, and for quadrilaterals nq*NF*2 shape
// derivatives.  Returns nq.
int kn_gamma_quadrature(int NF, std::vector<double>* out) {
  static const double gx[4] = {0.5 - 0.5 * 0.8611363115940526, 0.5 - 0.5 * 0.3399810435848563,
                               0.5 + 0.5 * 0.3399810435848563, 0.5 + 0.5 * 0.8611363115940526};
  static const double gw[4] = {0.5 * 0.3478548451374538, 0.5 * 0.6521451548625461,
                               0.5 * 0.6521451548625461, 0.5 * 0.3478548451374538};
  std::vector<double>& q = *out;
  q.clear();
  if (NF == 2) {
    for (int i = 0; i < 4; ++i) q.push_back(gw[i]);
    for (int i = 0; i < 4; ++i) { q.push_back(1.0 - gx[i]); q.push_back(gx[i]); }
    return 4;
  }
  if (NF == 3) {
    // 12-point degree-6 symmetric rule (Dunavant), polished to full double precision
    const double w1 = 0.1167862757263793660252896, b1 = 0.2492867451709104212916386;
    const double w2 = 0.05084490637020681692093681, b2 = 0.0630890144915022283403316;
    const double w3 = 0.08285107561837357519355346, b3 = 0.05314504984481694735324967,
                 c3 = 0.3103524510337844054166077;
    const double a1 = 1.0 - 2.0 * b1, a2 = 1.0 - 2.0 * b2, a3 = 1.0 - b3 - c3;
    const double P[12][4] = {
        {w1, a1, b1, b1}, {w1, b1, a1, b1}, {w1, b1, b1, a1},
        {w2, a2, b2, b2}, {w2, b2, a2, b2}, {w2, b2, b2, a2},
        {w3, a3, b3, c3}, {w3, a3, c3, b3}, {w3, b3, a3, c3},
        {w3, b3, c3, a3}, {w3, c3, a3, b3}, {w3, c3, b3, a3}};
    for (int i = 0; i < 12; ++i) q.push_back(0.5 * P[i][0]);
    for (int i = 0; i < 12; ++i) for (int b = 0; b < 3; ++b) q.push_back(P[i][1 + b]);
    return 12;
  }
  // quadrilateral: 4 x 4 Gauss, vertices in lexicographic order
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) q.push_back(gw[i] * gw[j]);
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) {
    const double x = gx[i], y = gx[j];
    q.push_back((1 - x) * (1 - y)); q.push_back(x * (1 - y)); q.push_back((1 - x) * y); q.push_back(x * y);
  }
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) {
    const double x = gx[i], y = gx[j];
    const double d[4][2] = {{-(1 - y), -(1 - x)}, {(1 - y), -x}, {-y, (1 - x)}, {y, x}};
    for (int b = 0; b < 4; ++b) { q.push_back(d[b][0]); q.push_back(d[b][1]); }
  }
  return 16;
}

extern "C" int knpemi_create(const knpemi_problem_desc* d, int device, knpemi_handle** out) {
  if (!d || !out) return fail(KNPEMI_EINVAL, "knpemi_create: null argument");
  *out = nullptr;
  if (d->n_ions < 2 || d->n_ions > KN_MAXK)
    return fail(KNPEMI_EINVAL, "knpemi_create: 2 to 4 ionic species (the last one eliminated) are supported; the "
                               "reference drivers use 3 (run_3D.py:256)");
  if (d->n_sub < 1 || d->n_sub > KN_MAXSUB) return fail(KNPEMI_EINVAL, "knpemi_create: bad n_sub");
  int NV, NF;
  if (d->cell_kind == KNPEMI_TRIANGLE && d->gdim == 2) { NV = 3; NF = 2; }
  else if (d->cell_kind == KNPEMI_TETRAHEDRON && d->gdim == 3) { NV = 4; NF = 3; }
  else if (d->cell_kind == KNPEMI_HEXAHEDRON && d->gdim == 3) { NV = 8; NF = 4; }
  else return fail(KNPEMI_EINVAL, "knpemi_create: cell_kind/gdim combination not supported");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(KNPEMI_EHIP, "knpemi_create: no HIP device visible (the hot path has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(KNPEMI_EINVAL, "knpemi_create: bad device index");
  KN_HIP(hipSetDevice(device));

  auto* h = new knpemi_handle();
  std::unique_ptr<knpemi_handle, void (*)(knpemi_handle*)> guard(h, knpemi_destroy);
  h->device = device;
  KN_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  KN_HIP(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
  KN_HIP(hipStreamCreateWithFlags(&h->aux2, hipStreamNonBlocking));
  KN_HIP(hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming));
  KN_HIP(hipEventCreateWithFlags(&h->ev_pre_fork, hipEventDisableTiming));
  KN_HIP(hipEventCreateWithFlags(&h->ev_pre, hipEventDisableTiming));
  h->cur = h->stream;
  KN_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  KN_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  KN_HIP(hipEventCreate(&h->ev0));
  KN_HIP(hipEventCreate(&h->ev1));
  h->gdim = d->gdim; h->cell_kind = d->cell_kind; h->NV = NV; h->NF = NF;
  h->n_sub = d->n_sub; h->K = d->n_ions;
  const int S = d->n_sub, K = d->n_ions;
  h->n_vert.assign(d->n_vert, d->n_vert + S);
  h->n_cell.assign(d->n_cell, d->n_cell + S);
  h->n_q.assign(S, 0); h->n_facet.assign(S, 0); h->n_models.assign(S, 0);
  for (int s = 1; s < S; ++s) {
    h->n_q[s] = d->n_q ? d->n_q[s] : 0;
    h->n_facet[s] = d->n_facet ? d->n_facet[s] : 0;
    h->n_models[s] = d->n_models ? d->n_models[s] : 0;
    if (h->n_models[s] > KNPEMI_MAX_MODELS) return fail(KNPEMI_EINVAL, "too many membrane models");
  }
  auto prefix = [&](const std::vector<int>& n) {
    std::vector<int> o(S + 1, 0);
    for (int s = 0; s < S; ++s) o[s + 1] = o[s] + n[s];
    return o;
  };
  h->voff = prefix(h->n_vert); h->coff = prefix(h->n_cell); h->qoff = prefix(h->n_q);
  h->foff = prefix(h->n_facet); h->moff = prefix(h->n_models);
  const int Ntot = h->voff[S], nctot = h->coff[S], NQtot = h->qoff[S], nftot = h->foff[S];
  h->ode.resize(h->moff[S]);

  // ---- global cells, vertex records -----------------------------------------------------------
  std::vector<int> cells((size_t)nctot * NV);
  std::vector<double> VR((size_t)Ntot * KN_REC, 0.0);
  for (int s = 0; s < S; ++s) {
    for (int64_t i = 0; i < (int64_t)h->n_cell[s] * NV; ++i) {
      int v = d->cells[s][i];
      if (v < 0 || v >= h->n_vert[s]) return fail(KNPEMI_EINVAL, "cell vertex id out of range");
      cells[(size_t)h->coff[s] * NV + i] = v + h->voff[s];
    }
    for (int v = 0; v < h->n_vert[s]; ++v)
      for (int c = 0; c < d->gdim; ++c)
        VR[(size_t)(h->voff[s] + v) * KN_REC + c] = d->x[s][(size_t)v * d->gdim + c];
  }

  // ---- vertex -> cell adjacency (counting sort) ---------------------------------------------------
  std::vector<int64_t> v2c_ptr(Ntot + 1, 0);
  for (size_t i = 0; i < cells.size(); ++i) v2c_ptr[cells[i] + 1]++;
  for (int g = 0; g < Ntot; ++g) v2c_ptr[g + 1] += v2c_ptr[g];
  std::vector<int> v2c(cells.size());
  {
    std::vector<int64_t> fill(v2c_ptr.begin(), v2c_ptr.end() - 1);
    for (int c = 0; c < nctot; ++c)
      for (int li = 0; li < NV; ++li) v2c[fill[cells[(size_t)c * NV + li]]++] = c * 8 + li;
  }

  // ---- Laplacian pattern -------------------------------------------------------------------------
  std::vector<int>& rowptrL = h->h_rowptrL; std::vector<int>& colindL = h->h_colindL;
  rowptrL.assign(Ntot + 1, 0);
  colindL.reserve((size_t)Ntot * (NV == 8 ? 27 : (NV == 4 ? 15 : 7)));
  {
    std::vector<int> tmp;
    for (int g = 0; g < Ntot; ++g) {
      tmp.clear();
      tmp.push_back(g);
      for (int64_t p = v2c_ptr[g]; p < v2c_ptr[g + 1]; ++p) {
        const int* cv = &cells[(size_t)(v2c[p] >> 3) * NV];
        tmp.insert(tmp.end(), cv, cv + NV);
      }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      colindL.insert(colindL.end(), tmp.begin(), tmp.end());
      if (colindL.size() > (size_t)INT32_MAX) return fail(KNPEMI_EINVAL, "matrix too large for int32 CSR");
      rowptrL[g + 1] = (int)colindL.size();
    }
  }

  // ---- membrane facets in global ids, coupling columns ------------------------------------------
  std::vector<int> fe((size_t)nftot * NF), fi((size_t)nftot * NF), fq((size_t)nftot * NF), fmodel(nftot, -1);
  std::vector<int> q2e(NQtot), q2i(NQtot);
  std::vector<std::pair<int, int>> coup;  // (row, col)
  std::vector<std::pair<int, int>> ment;  // (row, facet*8 + a)
  for (int s = 1; s < S; ++s) {
    for (int q = 0; q < h->n_q[s]; ++q) {
      q2e[h->qoff[s] + q] = d->q_to_e[s][q] + h->voff[0];
      q2i[h->qoff[s] + q] = d->q_to_i[s][q] + h->voff[s];
    }
    for (int f = 0; f < h->n_facet[s]; ++f) {
      int fg = h->foff[s] + f;
      int m = d->facet_model ? d->facet_model[s][f] : -1;
      if (m >= h->n_models[s]) return fail(KNPEMI_EINVAL, "facet_model out of range");
      fmodel[fg] = m < 0 ? -1 : h->moff[s] + m;
      for (int a = 0; a < NF; ++a) {
        int e = d->facet_e[s][(size_t)f * NF + a], i = d->facet_i[s][(size_t)f * NF + a];
        int q = d->facet_q[s][(size_t)f * NF + a];
        if (e < 0 || e >= h->n_vert[0] || i < 0 || i >= h->n_vert[s] || q < 0 || q >= h->n_q[s])
          return fail(KNPEMI_EINVAL, "membrane facet index out of range");
        fe[(size_t)fg * NF + a] = e + h->voff[0];
        fi[(size_t)fg * NF + a] = i + h->voff[s];
        fq[(size_t)fg * NF + a] = q + h->qoff[s];
      }
      for (int a = 0; a < NF; ++a) {
        ment.emplace_back(fe[(size_t)fg * NF + a], fg * 8 + a);
        ment.emplace_back(fi[(size_t)fg * NF + a], fg * 8 + a);
        for (int b = 0; b < NF; ++b) {
          coup.emplace_back(fe[(size_t)fg * NF + a], fi[(size_t)fg * NF + b]);
          coup.emplace_back(fi[(size_t)fg * NF + a], fe[(size_t)fg * NF + b]);
        }
      }
    }
  }
  std::sort(coup.begin(), coup.end());
  coup.erase(std::unique(coup.begin(), coup.end()), coup.end());
  std::stable_sort(ment.begin(), ment.end(),
                   [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });

  // ---- EMI pattern: ECS rows [laplacian | coupling], cell rows [coupling | laplacian] --------------
  std::vector<int>& rowptr = h->h_rowptr; std::vector<int>& colind = h->h_colind;
  rowptr.assign(Ntot + 1, 0);
  colind.reserve(colindL.size() + coup.size());
  std::vector<uint8_t> lapoff(Ntot, 0);
  {
    size_t cp = 0;
    for (int g = 0; g < Ntot; ++g) {
      size_t c0 = cp;
      while (cp < coup.size() && coup[cp].first == g) ++cp;
      int nco = (int)(cp - c0);
      const bool ecs = g < h->voff[1];
      if (!ecs) for (size_t c = c0; c < cp; ++c) colind.push_back(coup[c].second);
      colind.insert(colind.end(), colindL.begin() + rowptrL[g], colindL.begin() + rowptrL[g + 1]);
      if (ecs) for (size_t c = c0; c < cp; ++c) colind.push_back(coup[c].second);
      if (colind.size() > (size_t)INT32_MAX) return fail(KNPEMI_EINVAL, "matrix too large for int32 CSR");
      rowptr[g + 1] = (int)colind.size();
      if (rowptr[g + 1] - rowptr[g] > 255) return fail(KNPEMI_EINVAL, "matrix row longer than 255 entries");
      lapoff[g] = (uint8_t)(ecs ? 0 : nco);
    }
  }

  // ---- hexahedra: is every cell a parallelepiped? (constant Jacobian -> specialised row kernels) -----
  if (NV == 8) {
    bool affine = true;
    for (size_t c = 0; c < (size_t)nctot && affine; ++c) {
      const int* cv = &cells[c * 8];
      double dev = 0.0, len = 0.0;
      for (int t = 0; t < 3; ++t) {
        double e0[3] = {0, 0, 0};
        for (int k = 0; k < 4; ++k) {
          const int v = ((k >> t) << (t + 1)) | (k & ((1 << t) - 1)), u = v | (1 << t);
          for (int a = 0; a < 3; ++a) {
            const double e = VR[(size_t)cv[u] * KN_REC + a] - VR[(size_t)cv[v] * KN_REC + a];
            if (k == 0) { e0[a] = e; len += std::fabs(e); }
            else dev += std::fabs(e - e0[a]);
          }
        }
      }
      affine = dev <= 1e-13 * len;
    }
    h->hex_affine = affine && getenv("KNPEMI_HEX_GENERAL") == nullptr;
    // ... and is every cell the SAME parallelepiped (every box mesh of the reference's 3-D driver, make_mesh_3D.py:100-102:
    // create_box with a uniform grid)?  Then the geometry is a constant of the mesh and the row kernels stage no
    // coordinates.  The constant comes from the descriptor when the mesh generator supplies its exact cell
    // (knpemi_problem_desc::uniform_cell: the same bits on every rank of a partitioned run), else from the first cell;
    // every cell is checked against it.
    if (h->hex_affine && nctot > 0 && getenv("KNPEMI_HEX_NOT_UNIFORM") == nullptr) {
      double E[3][3];      // E[t][a]: component a of edge vector t
      bool given = false, uniform = false;
      for (int i = 0; i < 9; ++i) given |= d->uniform_cell[i] != 0.0;
      // (a supplied cell that the coordinates do not bear out -- a generated mesh transformed afterwards -- is ignored)
      for (int attempt = given ? 0 : 1; attempt < 2 && !uniform; ++attempt) {
        for (int t = 0; t < 3; ++t)
          for (int a = 0; a < 3; ++a)
            E[t][a] = attempt == 0 ? d->uniform_cell[3 * t + a]
                                   : VR[(size_t)cells[1 << t] * KN_REC + a] - VR[(size_t)cells[0] * KN_REC + a];
        double scale = 0.0;
        for (int t = 0; t < 3; ++t) for (int a = 0; a < 3; ++a) scale = std::max(scale, std::fabs(E[t][a]));
        uniform = scale > 0.0;
        for (size_t c = 0; c < (size_t)nctot && uniform; ++c) {
          const int* cv = &cells[c * 8];
          for (int t = 0; t < 3 && uniform; ++t)
            for (int a = 0; a < 3; ++a) {
              const double e = VR[(size_t)cv[1 << t] * KN_REC + a] - VR[(size_t)cv[0] * KN_REC + a];
              if (std::fabs(e - E[t][a]) > 1e-9 * scale) { uniform = false; break; }
            }
        }
      }
      if (uniform) {
        // the arithmetic of hexcf::geometry (kernels_assemble.hip) on the edge vectors
        double J[3][3];
        for (int t = 0; t < 3; ++t) for (int a = 0; a < 3; ++a) J[a][t] = E[t][a];
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1], c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2],
                     c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
        if (det != 0.0) {
          const double inv = 1.0 / det;
          double I[3][3];
          I[0][0] = c00 * inv; I[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * inv; I[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * inv;
          I[1][0] = c01 * inv; I[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * inv; I[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * inv;
          I[2][0] = c02 * inv; I[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * inv; I[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * inv;
          KnHexGeo& G = h->hex_geo;
          G.det = std::fabs(det);
          for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) G.g[a][b] = G.det * (I[a][0] * I[b][0] + I[a][1] * I[b][1] + I[a][2] * I[b][2]);
          G.skew = (G.g[0][1] != 0.0 || G.g[0][2] != 0.0 || G.g[1][2] != 0.0) ? 1 : 0;
          // the uniform kernels are built for BOX cells (diagonal metric: what the reference's meshes are); a uniform
          // sheared mesh keeps the general parallelepiped kernels
          h->hex_uniform = G.skew == 0;
        }
      }
    }
  }

  // ---- static ICS mass of the preconditioner: P_emi = A_emi + int u v dx on the cell sub-domains ---------
  // (emiWeakForm.py:169-198).  The geometry never changes, so the mass entries are tabulated once (P1: closed
  // form; Q1: the 2x2x2 Gauss rule) in the layout of the cell-side CSR rows and added when P is written.
  const int64_t pmass0 = rowptr[h->voff[1]];
  std::vector<double> pmass((size_t)(rowptr[Ntot] - pmass0), 0.0);
  for (int sdm = 1; sdm < S; ++sdm)
    for (int c = h->coff[sdm]; c < h->coff[sdm + 1]; ++c) {
      const int* cv = &cells[(size_t)c * NV];
      double X[8][3] = {};
      for (int a = 0; a < NV; ++a)
        for (int t = 0; t < d->gdim; ++t) X[a][t] = VR[(size_t)cv[a] * KN_REC + t];
      double Me[8][8] = {};
      if (NV == 3 || NV == 4) {
        double det;
        if (NV == 3) {
          det = (X[1][0] - X[0][0]) * (X[2][1] - X[0][1]) - (X[1][1] - X[0][1]) * (X[2][0] - X[0][0]);
        } else {
          const double ax = X[1][0] - X[0][0], ay = X[1][1] - X[0][1], az = X[1][2] - X[0][2];
          const double bx = X[2][0] - X[0][0], by = X[2][1] - X[0][1], bz = X[2][2] - X[0][2];
          const double cx = X[3][0] - X[0][0], cy = X[3][1] - X[0][1], cz = X[3][2] - X[0][2];
          det = ax * (by * cz - bz * cy) + ay * (bz * cx - bx * cz) + az * (bx * cy - by * cx);
        }
        const double vol = std::fabs(det) / (NV == 3 ? 2.0 : 6.0), m = vol / ((d->gdim + 1) * (d->gdim + 2));
        for (int i = 0; i < NV; ++i)
          for (int j = 0; j < NV; ++j) Me[i][j] = i == j ? 2.0 * m : m;
      } else {
        const double g0 = 0.5 - 0.28867513459481287, g1 = 0.5 + 0.28867513459481287;
        for (int q = 0; q < 8; ++q) {
          double N[8], dN[8][3], J[3][3] = {};
          for (int v = 0; v < 8; ++v) {
            double f[3], df[3];
            for (int ax = 0; ax < 3; ++ax) {
              const double xq = ((q >> ax) & 1) ? g1 : g0;
              const bool hi = (v >> ax) & 1;
              f[ax] = hi ? xq : 1.0 - xq;
              df[ax] = hi ? 1.0 : -1.0;
            }
            N[v] = f[0] * f[1] * f[2];
            dN[v][0] = df[0] * f[1] * f[2]; dN[v][1] = f[0] * df[1] * f[2]; dN[v][2] = f[0] * f[1] * df[2];
            for (int a = 0; a < 3; ++a)
              for (int t = 0; t < 3; ++t) J[a][t] += X[v][a] * dN[v][t];
          }
          const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) + J[0][1] * (J[1][2] * J[2][0] - J[1][0] * J[2][2]) +
                             J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
          const double wd = 0.125 * std::fabs(det);
          for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 8; ++j) Me[i][j] += wd * N[i] * N[j];
        }
      }
      for (int i = 0; i < NV; ++i) {
        const int g = cv[i];
        const int* rb = &colindL[rowptrL[g]];
        const int* re = &colindL[rowptrL[g + 1]];
        const int64_t base = (int64_t)rowptr[g] + lapoff[g] - pmass0;
        for (int j = 0; j < NV; ++j) pmass[(size_t)(base + (std::lower_bound(rb, re, cv[j]) - rb))] += Me[i][j];
      }
    }

  // ---- lanes per row: enough workgroups to fill 256 CUs several times over on small meshes --------
  {
    // simplices (v2 kernels stage 48 B per Laplacian entry in LDS): 64-row blocks for tetrahedra,
    // 128-row blocks for triangles; hexahedra: enough workgroups to fill the chip on small meshes
    int lpr = NV == 3 ? 2 : 4;   // hexahedra: 8 heavy pairs per row, two per lane
    if (const char* env = getenv("KNPEMI_LPR")) {
      int v = atoi(env);
      if (v == 1 || v == 2 || v == 4 || v == 8) lpr = v;
    }
    h->lpr = lpr;
  }
  const int LPR = h->lpr, RPB = KN_BLOCK / LPR, RPS = KN_SLICE / LPR;

  // ---- row blocks (never straddle a sub-domain) and sliced ELL of (row, cell) pairs ----------------
  // A block holds NR chunks of up to KN_CHUNK consecutive rows (kernels_assemble.hip: BlkRows).  Chunks are clustered
  // greedily by the number of Laplacian entries that connect them, so that the rows of a block share neighbours: on
  // the x-fastest numbering of the box meshes 64 consecutive rows are a line of vertices that touches 456 distinct
  // vertices, a bundle of pieces of neighbouring lines 230-270.  KNPEMI_BLOCK_CLASSIC=1: consecutive chunks.
  const int CH = NV == 8 ? KN_CHUNK_HEX : KN_CHUNK_SIMPLEX, NR = RPB / CH;
  struct Chunk { int start, len, sub; };
  std::vector<Chunk> chunks;
  std::vector<int> chunk_of(Ntot, -1);
  for (int s = 0; s < S; ++s)
    for (int r0 = h->voff[s]; r0 < h->voff[s + 1]; r0 += CH) {
      const int len = std::min(CH, h->voff[s + 1] - r0);
      for (int g = r0; g < r0 + len; ++g) chunk_of[g] = (int)chunks.size();
      chunks.push_back({r0, len, s});
    }
  const int nchunks = (int)chunks.size();
  std::vector<std::vector<int>> blocks;
  h->blocks_clustered = getenv("KNPEMI_BLOCK_CLASSIC") == nullptr;
  if (!h->blocks_clustered) {
    for (int c = 0; c < nchunks;) {
      std::vector<int> blk;
      const int s = chunks[c].sub;
      while (c < nchunks && chunks[c].sub == s && (int)blk.size() < NR) blk.push_back(c++);
      blocks.push_back(blk);
    }
  } else {
    // greedy clustering; `cap` bounds the distinct vertices of a block (the LDS of a launch is sized by the largest
    // block): a first pass without a bound gives the distribution, the second pass cuts its tail at 9/8 of the 90th
    // percentile by closing a block early
    std::vector<int> seen(Ntot, -1);
    auto cluster = [&](int cap, std::vector<int>* uniq_out) {
      blocks.clear();
      std::fill(seen.begin(), seen.end(), -1);
      std::vector<char> taken(nchunks, 0);
      std::vector<int> weight(nchunks, 0), cand;
      int next_free = 0;
      for (int c0 = 0; c0 < nchunks; ++c0) {
        if (taken[c0]) continue;
        const int s = chunks[c0].sub, id = (int)blocks.size();
        std::vector<int> blk;
        cand.clear();
        int uniq = 0;
        auto fresh = [&](int c, bool mark) {   // distinct vertices chunk c would add to the block
          int n = 0;
          for (int p = rowptrL[chunks[c].start]; p < rowptrL[chunks[c].start + chunks[c].len]; ++p) {
            const int v = colindL[p];
            if (seen[v] == id || (!mark && seen[v] == -2 - id)) continue;     // in the block already / counted in this probe
            if (mark) seen[v] = id;
            else seen[v] = -2 - id;             // undone below
            ++n;
          }
          if (!mark)
            for (int p = rowptrL[chunks[c].start]; p < rowptrL[chunks[c].start + chunks[c].len]; ++p)
              if (seen[colindL[p]] == -2 - id) seen[colindL[p]] = -1;
          return n;
        };
        auto take = [&](int c) {
          taken[c] = 1;
          blk.push_back(c);
          uniq += fresh(c, true);
          for (int g = chunks[c].start; g < chunks[c].start + chunks[c].len; ++g)
            for (int p = rowptrL[g]; p < rowptrL[g + 1]; ++p) {
              const int d = chunk_of[colindL[p]];
              if (taken[d] || chunks[d].sub != s) continue;
              if (weight[d]++ == 0) cand.push_back(d);
            }
        };
        take(c0);
        while ((int)blk.size() < NR) {
          int best = -1;
          for (int d : cand)
            if (!taken[d] && (best < 0 || weight[d] > weight[best] || (weight[d] == weight[best] && d < best))) best = d;
          if (best < 0) {   // no free neighbour left: continue with the next free chunk of the sub-domain
            int f = std::max(next_free, c0 + 1);      // every chunk below next_free is taken
            while (f < nchunks && chunks[f].sub == s && taken[f]) ++f;
            next_free = f;
            if (f >= nchunks || chunks[f].sub != s) break;
            best = f;
          }
          if (uniq + fresh(best, false) > cap) break;
          take(best);
        }
        for (int d : cand) weight[d] = 0;
        std::sort(blk.begin(), blk.end());
        blocks.push_back(blk);
        if (uniq_out) uniq_out->push_back(uniq);
      }
    };
    std::vector<int> uq;
    cluster(INT32_MAX, &uq);
    if (!uq.empty()) {
      std::sort(uq.begin(), uq.end());
      const int p90 = uq[(size_t)(0.9 * (uq.size() - 1))];
      if (uq.back() > p90 + p90 / 8) cluster(p90 + p90 / 8, nullptr);
    }
  }
  const int nblocks = (int)blocks.size();
  std::vector<int> blk_sub(nblocks);
  // row of slot t of block b (-1: none), slots t = r * CH + o
  auto slot_row = [&](int b, int t) {
    const int r = t / CH, o = t % CH;
    if (r >= (int)blocks[b].size()) return -1;
    const Chunk& c = chunks[blocks[b][r]];
    return o < c.len ? c.start + o : -1;
  };
  for (int b = 0; b < nblocks; ++b) blk_sub[b] = chunks[blocks[b][0]].sub;
  const int SPB = KN_BLOCK / KN_SLICE;  // slices (wavefronts) per block
  const int SW = NV == 8 ? 2 : 1;
  std::vector<int64_t> sl_ptr((size_t)nblocks * SPB + 1, 0);
  for (int b = 0; b < nblocks; ++b)
    for (int w = 0; w < SPB; ++w) {
      int64_t mx = 0;
      for (int t = w * RPS; t < (w + 1) * RPS; ++t) {
        const int g = slot_row(b, t);
        if (g >= 0) mx = std::max(mx, (v2c_ptr[g + 1] - v2c_ptr[g] + LPR - 1) / LPR);
      }
      sl_ptr[(size_t)b * SPB + w + 1] = sl_ptr[(size_t)b * SPB + w] + mx * KN_SLICE;
    }
  const bool simplex = NV != 8;
  // ---- tetrahedra: is every cell a lattice tetrahedron of a uniform grid? ------------------------------------------
  // (the box meshes of the reference's 3-D driver split into tetrahedra, knpemi_problem_desc::uniform_cell: the vertices of a
  // cell are corners of ONE grid cell.)  Then a cell is one of a handful of shapes -- six for the Kuhn split -- whose gradient
  // dot products and volume follow from the grid's edge vectors alone: the same bits on every rank, no coordinates staged,
  // no geometry in the row kernels (kernels_assemble.hip: tet_table_row0).  The pair entries carry the shape (3 bits) and
  // the canonical number of each vertex (its rank among the cell's corner codes, 2 bits), which leaves 5 bits per slot.
  std::vector<uint8_t> tet_shape, tet_canon;      // per cell: shape; canonical numbers of the four vertices, 2 bits each
  std::vector<double> tet_tab;
  if (NV == 4 && d->gdim == 3 && nctot > 0 && getenv("KNPEMI_TET_NOT_UNIFORM") == nullptr) {
    bool ok = false;
    for (int i = 0; i < 9; ++i) ok |= d->uniform_cell[i] != 0.0;
    int longest = 0;
    for (int g = 0; g < Ntot; ++g) longest = std::max(longest, rowptrL[g + 1] - rowptrL[g]);
    ok = ok && longest <= 31;
    double E[3][3], Ei[3][3], scale = 0.0;      // E[t][a]: component a of grid edge t; Ei: x = sum_t l_t E[t]  =>  l = Ei (x)
    for (int t = 0; t < 3; ++t) for (int a = 0; a < 3; ++a) { E[t][a] = d->uniform_cell[3 * t + a]; scale = std::max(scale, std::fabs(E[t][a])); }
    if (ok) {
      const double det = E[0][0] * (E[1][1] * E[2][2] - E[1][2] * E[2][1]) - E[0][1] * (E[1][0] * E[2][2] - E[1][2] * E[2][0]) +
                         E[0][2] * (E[1][0] * E[2][1] - E[1][1] * E[2][0]);
      ok = det != 0.0;
      if (ok) {
        // rows of the inverse of the matrix whose COLUMNS are the edges
        const double c[3][3] = {{E[1][1] * E[2][2] - E[1][2] * E[2][1], E[1][2] * E[2][0] - E[1][0] * E[2][2], E[1][0] * E[2][1] - E[1][1] * E[2][0]},
                                {E[2][1] * E[0][2] - E[2][2] * E[0][1], E[2][2] * E[0][0] - E[2][0] * E[0][2], E[2][0] * E[0][1] - E[2][1] * E[0][0]},
                                {E[0][1] * E[1][2] - E[0][2] * E[1][1], E[0][2] * E[1][0] - E[0][0] * E[1][2], E[0][0] * E[1][1] - E[0][1] * E[1][0]}};
        for (int t = 0; t < 3; ++t) for (int a = 0; a < 3; ++a) Ei[t][a] = c[t][a] / det;
      }
    }
    std::vector<int> keys;                      // shape -> sorted corner codes, 3 bits each
    if (ok) {
      tet_shape.assign((size_t)nctot, 0);
      tet_canon.assign((size_t)nctot, 0);
      for (size_t cc = 0; cc < (size_t)nctot && ok; ++cc) {
        const int* cv = &cells[cc * 4];
        int l[4][3], code[4];
        for (int i = 0; i < 4 && ok; ++i)
          for (int t = 0; t < 3; ++t) {
            double x = 0.0;
            for (int a = 0; a < 3; ++a) x += Ei[t][a] * (VR[(size_t)cv[i] * KN_REC + a] - VR[(size_t)cv[0] * KN_REC + a]);
            const double r = std::nearbyint(x);
            if (std::fabs(x - r) > 1e-6 || std::fabs(r) > 1.0) { ok = false; break; }
            l[i][t] = (int)r;
          }
        if (!ok) break;
        for (int t = 0; t < 3; ++t) {
          int m = 0;
          for (int i = 0; i < 4; ++i) m = std::min(m, l[i][t]);
          for (int i = 0; i < 4; ++i) { l[i][t] -= m; if (l[i][t] > 1) ok = false; }
        }
        if (!ok) break;
        for (int i = 0; i < 4; ++i) code[i] = l[i][0] | (l[i][1] << 1) | (l[i][2] << 2);
        int key = 0, canon = 0, sorted[4] = {code[0], code[1], code[2], code[3]};
        std::sort(sorted, sorted + 4);
        for (int i = 0; i < 3; ++i) if (sorted[i] == sorted[i + 1]) ok = false;
        if (!ok) break;
        for (int i = 0; i < 4; ++i) {
          key |= sorted[i] << (3 * i);
          canon |= (int)(std::lower_bound(sorted, sorted + 4, code[i]) - sorted) << (2 * i);
        }
        int sh = (int)(std::find(keys.begin(), keys.end(), key) - keys.begin());
        if (sh == (int)keys.size()) { keys.push_back(key); if (keys.size() > 8) { ok = false; break; } }
        tet_shape[cc] = (uint8_t)sh;
        tet_canon[cc] = (uint8_t)canon;
      }
    }
    if (ok) {
      tet_tab.assign(8 * 16 + 8, 0.0);
      for (size_t sh = 0; sh < keys.size() && ok; ++sh) {
        double P[4][3], e[3][3];
        for (int i = 0; i < 4; ++i) {
          const int code = (keys[sh] >> (3 * i)) & 7;
          for (int a = 0; a < 3; ++a) P[i][a] = ((code & 1) ? E[0][a] : 0.0) + ((code & 2) ? E[1][a] : 0.0) + ((code & 4) ? E[2][a] : 0.0);
        }
        for (int j = 0; j < 3; ++j) for (int a = 0; a < 3; ++a) e[j][a] = P[j + 1][a] - P[0][a];
        auto cross = [](const double* u, const double* v, double* w) {
          w[0] = u[1] * v[2] - u[2] * v[1]; w[1] = u[2] * v[0] - u[0] * v[2]; w[2] = u[0] * v[1] - u[1] * v[0];
        };
        double g[4][3];
        cross(e[1], e[2], g[1]); cross(e[2], e[0], g[2]); cross(e[0], e[1], g[3]);
        const double det = e[0][0] * g[1][0] + e[0][1] * g[1][1] + e[0][2] * g[1][2];
        if (det == 0.0) { ok = false; break; }
        for (int j = 1; j < 4; ++j) for (int a = 0; a < 3; ++a) g[j][a] /= det;
        for (int a = 0; a < 3; ++a) g[0][a] = -(g[1][a] + g[2][a] + g[3][a]);
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) tet_tab[(sh * 4 + i) * 4 + j] = g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2];
        tet_tab[128 + sh] = std::fabs(det) * (1.0 / 6.0);
      }
    }
    h->tet_uniform = ok;
    if (!ok) { tet_shape.clear(); tet_canon.clear(); tet_tab.clear(); }
  }
  std::vector<int> pair_cell(simplex ? 0 : (size_t)sl_ptr.back(), -1);
  std::vector<uint32_t> pair_slots(simplex ? 0 : (size_t)sl_ptr.back() * SW, 0);
  std::vector<uint32_t> pair_sl(simplex ? (size_t)sl_ptr.back() : 0, 0xFFFFFFFFu);
  for (int b = 0; b < nblocks; ++b)
    for (int t = 0; t < RPB; ++t) {
      const int g = slot_row(b, t), w = t / RPS, rs = t % RPS;
      if (g < 0) continue;
      const int* rb = &colindL[rowptrL[g]];
      const int* re = &colindL[rowptrL[g + 1]];
      if (simplex) {
        // The LPR lanes of the row take strips of its incident cells: consecutive cells of a lane share a face (an
        // edge in 2D) that contains the row vertex, and the shared vertices keep their byte position in the pair
        // entry, so the kernel re-reads from LDS only the record whose slot changed (one per pair instead of NV - 1).
        // Ties are broken by the cell order, vertex positions start from the cell's own vertex order: the same on
        // every partition of the mesh.
        const int n = (int)(v2c_ptr[g + 1] - v2c_ptr[g]), NO = NV - 1, Lmax = (n + LPR - 1) / LPR;
        std::vector<std::array<int, 3>> ov(n);
        for (int q = 0; q < n; ++q) {
          const int vc = v2c[v2c_ptr[g] + q], li = vc & 7;
          const int* cv = &cells[(size_t)(vc >> 3) * NV];
          for (int j = 1; j < NV; ++j) ov[q][j - 1] = cv[(li + j) % NV];
        }
        std::vector<char> used(n, 0);
        int remaining = n;
        for (int sub = 0; sub < LPR && remaining > 0; ++sub) {
          std::array<int, 3> prev{};
          bool have_prev = false;
          for (int len = 0; len < Lmax && remaining > 0; ++len) {
            auto shared = [&](const std::array<int, 3>& a) {
              int c = 0;
              for (int x = 0; x < NO; ++x) for (int y = 0; y < NO; ++y) c += a[x] == prev[y];
              return c;
            };
            int pick = -1;
            if (have_prev)
              for (int q = 0; q < n && pick < 0; ++q) if (!used[q] && shared(ov[q]) == NO - 1) pick = q;
            const bool strip = pick >= 0;
            for (int q = 0; q < n && pick < 0; ++q) if (!used[q]) pick = q;
            std::array<int, 3> cur = ov[pick];
            if (strip) {   // the vertices shared with the previous pair stay where they are, the new one takes the free place
              cur = prev;
              int fresh = -1;
              for (int x = 0; x < NO; ++x) {
                bool in_prev = false;
                for (int y = 0; y < NO; ++y) in_prev |= ov[pick][x] == prev[y];
                if (!in_prev) fresh = ov[pick][x];
              }
              for (int y = 0; y < NO; ++y) {
                bool kept = false;
                for (int x = 0; x < NO; ++x) kept |= ov[pick][x] == prev[y];
                if (!kept) cur[y] = fresh;
              }
            }
            uint32_t slots = (uint32_t)(std::lower_bound(rb, re, g) - rb);
            for (int j = 1; j < NV; ++j) slots |= (uint32_t)(std::lower_bound(rb, re, cur[j - 1]) - rb) << (8 * j);
            if (h->tet_uniform) {   // lattice tetrahedra: the cell's shape and the canonical numbers of the vertices in bytes 1-3
              const size_t cc = (size_t)(v2c[v2c_ptr[g] + pick] >> 3);
              const int* cv = &cells[cc * NV];
              slots |= (uint32_t)tet_shape[cc] << 5;
              for (int j = 1; j < NV; ++j) {
                int li = 0;
                while (li < NV && cv[li] != cur[j - 1]) ++li;
                slots |= (uint32_t)((tet_canon[cc] >> (2 * li)) & 3) << (8 * j + 5);
              }
            }
            const size_t ent = (size_t)sl_ptr[(size_t)b * SPB + w] + (size_t)len * KN_SLICE + (rs * LPR + sub);
            pair_sl[ent] = slots;
            used[pick] = 1;
            prev = cur;
            have_prev = true;
            --remaining;
          }
        }
        continue;
      }
      for (int64_t p = v2c_ptr[g]; p < v2c_ptr[g + 1]; ++p) {
        const int64_t pi = p - v2c_ptr[g];
        const int lane = rs * LPR + (int)(pi % LPR);
        size_t ent = (size_t)sl_ptr[(size_t)b * SPB + w] + (size_t)(pi / LPR) * KN_SLICE + lane;
        const int* cv = &cells[(size_t)(v2c[p] >> 3) * NV];
        if (simplex) {
          const int li = v2c[p] & 7;
          uint32_t slots = 0;
          for (int j = 0; j < NV; ++j) {
            const int v = cv[(li + j) % NV];   // j = 0 is the row's own vertex
            slots |= (uint32_t)(std::lower_bound(rb, re, v) - rb) << (8 * j);
          }
          pair_sl[ent] = slots;
          continue;
        }
        pair_cell[ent] = v2c[p];
        for (int j = 0; j < NV; ++j) {
          uint32_t slot = (uint32_t)(std::lower_bound(rb, re, cv[j]) - rb);
          pair_slots[ent * SW + (j >> 2)] |= slot << (8 * (j & 3));
        }
      }
    }

  // ---- membrane rows ----------------------------------------------------------------------------------
  std::vector<int> gam_idx(Ntot, -1), mptr(1, 0), mentry, mrow;
  std::vector<int> me_model, me_q, me_row, gam_pos((size_t)std::max(1, nftot) * 2 * NF, 0);
  std::vector<uint64_t> mslots;
  for (size_t i = 0; i < ment.size();) {
    int g = ment[i].first;
    gam_idx[g] = (int)mrow.size();
    mrow.push_back(g);
    bool ecs = g < h->voff[1];
    const int* rb = &colind[rowptr[g]];
    const int* re = &colind[rowptr[g + 1]];
    for (; i < ment.size() && ment[i].first == g; ++i) {
      int fg = ment[i].second >> 3;
      uint64_t sl = 0;
      for (int b = 0; b < NF; ++b) {
        int own = ecs ? fe[(size_t)fg * NF + b] : fi[(size_t)fg * NF + b];
        int oth = ecs ? fi[(size_t)fg * NF + b] : fe[(size_t)fg * NF + b];
        uint64_t so = (uint64_t)(std::lower_bound(rb, re, own) - rb);
        uint64_t st = (uint64_t)(std::lower_bound(rb, re, oth) - rb);
        sl |= so << (8 * b);
        sl |= st << (8 * (4 + b));
      }
      gam_pos[((size_t)fg * 2 + (ecs ? 0 : 1)) * NF + (ment[i].second & 7)] = (int)mentry.size();
      mentry.push_back(ment[i].second);
      mslots.push_back(sl);
      me_model.push_back(fmodel[fg]);
      me_row.push_back(g);
      for (int b = 0; b < NF; ++b) me_q.push_back(fq[(size_t)fg * NF + b]);
    }
    mptr.push_back((int)mentry.size());
  }

  // ---- per-block / per-row descriptors: one scalar load per block, one 16-byte load per row ----------
  std::vector<int> blk_info((size_t)nblocks * 16, 0), row_info((size_t)Ntot * 4, 0), blk_uverts;
  std::vector<int> blk_rng((size_t)nblocks * NR * 6, 0);
  std::vector<uint16_t> ent_loc(colindL.size(), 0);
  size_t ent_base = 0;
  for (int b = 0; b < nblocks; ++b) {
    int* bi = &blk_info[(size_t)b * 16];
    int* rg = &blk_rng[(size_t)b * NR * 6];
    int offE = 0, offL = 0, nrows = 0;
    for (int r = 0; r < NR; ++r) {
      int* q = rg + 6 * r;
      if (r < (int)blocks[b].size()) {
        const Chunk& c = chunks[blocks[b][r]];
        const int g0 = c.start, g1 = c.start + c.len;
        q[0] = g0; q[1] = c.len;
        q[3] = rowptr[g0] - offE; q[5] = rowptrL[g0] - offL;
        for (int g = g0; g < g1; ++g) {
          int* ri = &row_info[(size_t)g * 4];
          const int m = gam_idx[g];
          const int ne = m < 0 ? 0 : mptr[m + 1] - mptr[m];
          ri[0] = offE + (rowptr[g] - rowptr[g0]);
          ri[1] = ri[0] + lapoff[g];
          if (ri[1] > 0xFFFF || ne > 0x7FFF) return fail(KNPEMI_EINVAL, "row block too large for the packed row descriptor");
          ri[1] |= ne << 16;
          ri[2] = offL + (rowptrL[g] - rowptrL[g0]);
          ri[3] = m < 0 ? 0 : mptr[m];
        }
        offE += rowptr[g1] - rowptr[g0];
        offL += rowptrL[g1] - rowptrL[g0];
        nrows += c.len;
      } else {
        q[0] = chunks[blocks[b][0]].start; q[1] = 0;
        q[3] = rg[6 * (r - 1) + 3]; q[5] = rg[6 * (r - 1) + 5];
      }
      q[2] = offE; q[4] = offL;
    }
    bi[0] = chunks[blocks[b][0]].start; bi[1] = nrows; bi[2] = blk_sub[b]; bi[3] = 0;
    bi[4] = offE; bi[5] = (int)ent_base; bi[6] = offL;
    uint32_t steps = 0;
    for (int w = 0; w < SPB; ++w) {
      bi[8 + w] = (int)(sl_ptr[(size_t)b * SPB + w] / KN_SLICE);
      const int64_t np = (sl_ptr[(size_t)b * SPB + w + 1] - sl_ptr[(size_t)b * SPB + w]) / KN_SLICE;
      if (np > 255) return fail(KNPEMI_EINVAL, "a vertex has too many incident cells for the pair layout");
      steps |= (uint32_t)np << (8 * w);
    }
    bi[12] = (int)steps;
    if (!h->blocks_clustered) {   // membrane entries of the block's rows (entries are sorted by row: one range)
      int me0 = 0, mne = 0;
      for (int t = 0; t < RPB; ++t) {
        const int g = slot_row(b, t);
        const int m = g < 0 ? -1 : gam_idx[g];
        if (m < 0) continue;
        if (mne == 0) me0 = mptr[m];
        mne = mptr[m + 1] - me0;
      }
      bi[14] = me0; bi[15] = mne;
      h->lds_gam_max = std::max(h->lds_gam_max, mne);
    }
    // distinct vertices touched by the block's rows (sorted: consecutive ids = contiguous records); the local index
    // of every Laplacian entry, in the order of the block's concatenated segment
    {
      std::vector<int> u;
      for (int c : blocks[b])
        u.insert(u.end(), colindL.begin() + rowptrL[chunks[c].start], colindL.begin() + rowptrL[chunks[c].start + chunks[c].len]);
      std::vector<int> ordered(u);
      std::sort(u.begin(), u.end());
      u.erase(std::unique(u.begin(), u.end()), u.end());
      if (u.size() > 65535) return fail(KNPEMI_EINVAL, "row block touches more than 65535 vertices");
      bi[7] = (int)blk_uverts.size();
      bi[13] = (int)u.size();
      h->lds_uniq_max = std::max(h->lds_uniq_max, (int)u.size());
      for (size_t i = 0; i < ordered.size(); ++i)
        ent_loc[ent_base + i] = (uint16_t)(std::lower_bound(u.begin(), u.end(), ordered[i]) - u.begin());
      ent_base += ordered.size();
      if (ent_base > (size_t)INT32_MAX) return fail(KNPEMI_EINVAL, "mesh too large for int32 block lists");
      blk_uverts.insert(blk_uverts.end(), u.begin(), u.end());
      if (blk_uverts.size() > (size_t)INT32_MAX) return fail(KNPEMI_EINVAL, "mesh too large for int32 block lists");
    }
    h->lds_doubles_emi = std::max(h->lds_doubles_emi, offE);
    h->lds_doubles_knp = std::max(h->lds_doubles_knp, offL);
  }

  if (getenv("KNPEMI_DEBUG_LDS")) {
    // distribution of the per-block LDS needs (the launch uses the maxima)
    std::vector<int> segs, uniq, lsegs;
    for (int b = 0; b < nblocks; ++b) {
      segs.push_back(blk_info[(size_t)b * 16 + 4]);
      lsegs.push_back(blk_info[(size_t)b * 16 + 6]);
      uniq.push_back(blk_info[(size_t)b * 16 + 13]);
    }
    std::sort(segs.begin(), segs.end());
    std::sort(lsegs.begin(), lsegs.end());
    std::sort(uniq.begin(), uniq.end());
    fprintf(stderr, "[knpemi] Laplacian segment doubles: p50 %d p90 %d p99 %d max %d\n", lsegs[lsegs.size() / 2],
            lsegs[(size_t)(0.9 * (lsegs.size() - 1))], lsegs[(size_t)(0.99 * (lsegs.size() - 1))], lsegs.back());
    auto pct = [&](const std::vector<int>& v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
    fprintf(stderr, "[knpemi] blocks %d (%s) lpr %d | EMI segment doubles: p50 %d p90 %d p99 %d max %d | distinct vertices: p50 %d p90 %d p99 %d max %d\n",
            nblocks, h->blocks_clustered ? "clustered chunks" : "consecutive rows", LPR, pct(segs, .5), pct(segs, .9), pct(segs, .99),
            segs.back(), pct(uniq, .5), pct(uniq, .9), pct(uniq, .99), uniq.back());
  }

  // ---- upload --------------------------------------------------------------------------------------------
  KnDev& D = h->dev;
  D.Ntot = Ntot; D.nctot = nctot; D.NQtot = NQtot; D.nftot = nftot; D.nblocks = nblocks;
  D.M = (int)mrow.size();
  D.nnz = (int64_t)colind.size(); D.nnzL = (int64_t)colindL.size();
  int rc;
  const double* vr_c = nullptr;
  if ((rc = dev_upload(h, VR, &vr_c))) return rc;
  D.VR = const_cast<double*>(vr_c);
  if ((rc = dev_upload(h, cells, &D.cells))) return rc;
  if ((rc = dev_upload(h, blk_rng, &D.blk_rng))) return rc;
  if ((rc = dev_upload(h, blk_sub, &D.blk_sub))) return rc;
  {
    const int* bi = nullptr; const int* ri = nullptr;
    if ((rc = dev_upload(h, blk_info, &bi))) return rc;
    if ((rc = dev_upload(h, row_info, &ri))) return rc;
    D.blk_info = reinterpret_cast<const int4*>(bi);
    if ((rc = dev_upload(h, blk_uverts, &D.blk_uverts))) return rc;
    if ((rc = dev_upload(h, ent_loc, &D.ent_loc))) return rc;
    D.row_info = reinterpret_cast<const int4*>(ri);
  }
  if ((rc = dev_upload(h, sl_ptr, &D.sl_ptr))) return rc;
  if ((rc = dev_upload(h, pair_sl, &D.pair_sl))) return rc;
  D.tet_tab = nullptr;
  if (h->tet_uniform && (rc = dev_upload(h, tet_tab, &D.tet_tab))) return rc;
  if ((rc = dev_upload(h, pair_cell, &D.pair_cell))) return rc;
  if ((rc = dev_upload(h, pair_slots, &D.pair_slots))) return rc;
  if ((rc = dev_upload(h, rowptr, &D.rowptr))) return rc;
  if ((rc = dev_upload(h, colind, &D.colind))) return rc;
  if ((rc = dev_upload(h, lapoff, &D.lapoff))) return rc;
  if ((rc = dev_upload(h, rowptrL, &D.rowptrL))) return rc;
  if ((rc = dev_upload(h, colindL, &D.colindL))) return rc;
  if ((rc = dev_upload(h, mptr, &D.mptr))) return rc;
  if ((rc = dev_upload(h, mentry, &D.mentry))) return rc;
  if ((rc = dev_upload(h, mslots, &D.mslots))) return rc;
  if ((rc = dev_upload(h, mrow, &D.mrow))) return rc;
  if ((rc = dev_upload(h, me_model, &D.me_model))) return rc;
  if ((rc = dev_upload(h, me_q, &D.me_q))) return rc;
  if ((rc = dev_upload(h, gam_pos, &D.gam_pos))) return rc;
  if ((rc = dev_upload(h, pmass, &D.P_mass))) return rc;
  D.pmass0 = pmass0;
  if ((rc = dev_upload(h, fe, &D.fe))) return rc;
  if ((rc = dev_upload(h, fi, &D.fi))) return rc;
  if ((rc = dev_upload(h, fq, &D.fq))) return rc;
  if ((rc = dev_upload(h, fmodel, &D.fmodel))) return rc;
  if ((rc = dev_upload(h, q2e, &D.q2e))) return rc;
  if ((rc = dev_upload(h, q2i, &D.q2i))) return rc;
  if ((rc = dev_zeros(h, (size_t)(K - 1) * Ntot, &D.csol))) return rc;
  D.fsrc = nullptr;
  if ((rc = dev_zeros(h, (size_t)D.nnz, &D.A_emi))) return rc;
  if ((rc = dev_zeros(h, (size_t)D.nnz, &D.P_emi))) return rc;
  if ((rc = dev_zeros(h, (size_t)Ntot, &D.b_emi))) return rc;
  if ((rc = dev_zeros(h, (size_t)(K - 1) * D.nnzL, &D.A_knp))) return rc;
  if ((rc = dev_zeros(h, (size_t)(K - 1) * Ntot, &D.b_knp))) return rc;
  if ((rc = dev_zeros(h, (size_t)NQtot, &D.phiM))) return rc;
  if ((rc = dev_zeros(h, std::max<size_t>(1, mentry.size()) * (size_t)(K - 1), &D.gam_e))) return rc;
  if ((rc = dev_zeros(h, std::max<size_t>(1, mentry.size()) * (size_t)(K - 1) * (1 + NF), &D.gpre))) return rc;
  if ((rc = dev_zeros(h, (size_t)std::max(1, h->moff[S]) * KN_MAXK * std::max(1, NQtot), &D.Ich))) return rc;
  {
    std::vector<int> krp((size_t)(K - 1) * Ntot + 1, 0), kci((size_t)(K - 1) * colindL.size());
    int64_t row = 0, pos = 0;
    for (int s = 0; s < S; ++s) {
      const int v0 = h->voff[s], nv = h->n_vert[s];
      for (int k = 0; k < K - 1; ++k) {
        const int64_t shift = (int64_t)(K - 1) * v0 + (int64_t)k * nv - v0;
        for (int v = 0; v < nv; ++v) {
          for (int p = rowptrL[v0 + v]; p < rowptrL[v0 + v + 1]; ++p) kci[pos++] = (int)(colindL[p] + shift);
          krp[++row] = (int)pos;
        }
      }
    }
    if ((rc = dev_upload(h, krp, &D.krowptr))) return rc;
    if ((rc = dev_upload(h, kci, &D.kcolind))) return rc;
  }
  h->stage_len = (size_t)std::max(Ntot, 1) * (K - 1) + 2 * (size_t)std::max(NQtot, 1);
  if ((rc = dev_zeros(h, h->stage_len, &h->d_stage))) return rc;

  {
    std::vector<double> qt;
    D.nq_gamma = kn_gamma_quadrature(NF, &qt);
    if ((rc = dev_upload(h, qt, &D.qtab))) return rc;
  }

  {   // facet-mass rows of the membrane entries: static geometry, computed once by the device code the kernels share
    const int n_entries = (int)mentry.size();
    double* mass = nullptr;
    if ((rc = dev_zeros(h, std::max<size_t>(1, (size_t)n_entries * NF), &mass))) return rc;
    D.me_mass = mass;
    const int* d_rows = nullptr;
    if ((rc = dev_upload(h, me_row, &d_rows))) return rc;
    if ((rc = kn_launch_membrane_mass(h, n_entries, d_rows, mass))) return rc;
    KN_HIP(hipStreamSynchronize(h->stream));
  }

  KnConsts& C = h->consts;
  C.n_sub = S; C.K = K;
  for (int s = 0; s <= S; ++s) { C.voff[s] = h->voff[s]; C.qoff[s] = h->qoff[s]; }
  for (int s = S + 1; s <= KN_MAXSUB; ++s) { C.voff[s] = h->voff[S]; C.qoff[s] = h->qoff[S]; }
  C.splitting = 1;

  *out = guard.release();
  return KNPEMI_OK;
}

extern "C" void knpemi_destroy(knpemi_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->aux) (void)hipStreamSynchronize(h->aux);
  if (h->aux2) (void)hipStreamSynchronize(h->aux2);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->ev_join2) (void)hipEventDestroy(h->ev_join2);
  if (h->ev_pre_fork) (void)hipEventDestroy(h->ev_pre_fork);
  if (h->ev_pre) (void)hipEventDestroy(h->ev_pre);
  if (h->aux2) (void)hipStreamDestroy(h->aux2);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->aux) (void)hipStreamDestroy(h->aux);
  kn_comm_destroy(h);
  kn_amg_async_join(h->amg_emi);
  kn_amg_async_join(h->amg_knp);
  kn_amg_free(h->amg_emi);
  kn_amg_free(h->amg_knp);
  if (h->kry_pinned) (void)hipHostFree(h->kry_pinned);
  if (h->pub_host) (void)hipHostFree(h->pub_host);
  kn_fused_graphs_free(h);
  if (h->graph_emi.exec) (void)hipGraphExecDestroy(h->graph_emi.exec);
  if (h->graph_knp.exec) (void)hipGraphExecDestroy(h->graph_knp.exec);
  for (void* p : h->allocs) (void)hipFree(p);
  for (void* m : h->rtc_modules) (void)hipModuleUnload(static_cast<hipModule_t>(m));
  for (auto& v : h->prof_ev) for (hipEvent_t e : v) (void)hipEventDestroy(e);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" int knpemi_set_params(knpemi_handle* h, const knpemi_params* p) {
  if (h) h->gam_valid = false;
  if (!h || !p) return fail(KNPEMI_EINVAL, "knpemi_set_params: null argument");
  if (!(p->dt > 0) || !(p->C_M > 0) || p->z[h->K - 1] == 0.0)
    return fail(KNPEMI_EINVAL, "knpemi_set_params: dt, C_M must be positive and z_K non-zero");
  KnConsts& C = h->consts;
  C.dt = p->dt; C.inv_dt = 1.0 / p->dt; C.F = p->F; C.psi = p->psi; C.C_M = p->C_M;
  C.C_phi = p->C_phi > 0.0 ? p->C_phi : p->C_M / p->dt;
  const int K = h->K;
  for (int k = 0; k < K; ++k) {
    C.z[k] = p->z[k];
    C.elim_coef[k] = -(1.0 / p->z[K - 1]) * p->z[k];
  }
  for (int s = 0; s < h->n_sub; ++s) {
    KnSubConst& sc = C.sc[s];
    for (int k = 0; k < K; ++k) {
      const double z = p->z[k], Dk = p->D[s][k];
      sc.kap[k] = p->F * z * z * Dk * p->psi;
      sc.sig[k] = p->F * z * Dk;
      sc.D[k] = Dk;
      sc.zpsiD[k] = z * p->psi * Dk;
      sc.az2D[k] = Dk * z * z;
    }
    sc.rho_term = -(1.0 / p->z[K - 1]) * p->rho_z * p->rho[s];
  }
  KN_HIP(hipSetDevice(h->device));
  if (!h->d_consts) {
    void* p2 = nullptr;
    KN_HIP(hipMalloc(&p2, sizeof(KnConsts)));
    h->allocs.push_back(p2);
    h->d_consts = static_cast<KnConsts*>(p2);
  }
  KN_HIP(hipStreamSynchronize(h->stream));
  KN_HIP(hipMemcpy(h->d_consts, &h->consts, sizeof(KnConsts), hipMemcpyHostToDevice));
  h->have_params = 1;
  return KNPEMI_OK;
}

extern "C" int knpemi_sync(knpemi_handle* h) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  KN_HIP(hipStreamSynchronize(h->aux));
  KN_HIP(hipStreamSynchronize(h->aux2));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" void* knpemi_stream(knpemi_handle* h) { return h ? (void*)h->stream : nullptr; }

// ---------------------------------------------------------------------------------------------------
// field I/O
// ---------------------------------------------------------------------------------------------------
namespace {
struct FieldLoc { double* base; int stride; size_t n; };

int locate(knpemi_handle* h, int field, int sub, int idx, FieldLoc* loc) {
  if (sub < 0 || sub >= h->n_sub) return fail(KNPEMI_EINVAL, "field: bad sub-domain index");
  const int K = h->K;
  KnDev& D = h->dev;
  const size_t v0 = h->voff[sub], nv = h->n_vert[sub], q0 = h->qoff[sub], nq = h->n_q[sub];
  switch (field) {
    case KNPEMI_F_PHI: *loc = {D.VR + v0 * KN_REC + 7, KN_REC, nv}; return 0;
    case KNPEMI_F_C_PREV:
      if (idx < 0 || idx >= K - 1) return fail(KNPEMI_EINVAL, "field: bad ion index");
      *loc = {D.VR + v0 * KN_REC + KN_CSLOT(idx), KN_REC, nv}; return 0;
    case KNPEMI_F_C_ELIM: *loc = {D.VR + v0 * KN_REC + KN_CSLOT(K - 1), KN_REC, nv}; return 0;
    case KNPEMI_F_C:
      if (idx < 0 || idx >= K - 1) return fail(KNPEMI_EINVAL, "field: bad ion index");
      *loc = {D.csol + (size_t)idx * D.Ntot + v0, 1, nv}; return 0;
    case KNPEMI_F_PHI_M:
      if (sub == 0) return fail(KNPEMI_EINVAL, "field: the ECS has no membrane space");
      *loc = {D.phiM + q0, 1, nq}; return 0;
    case KNPEMI_F_I_CH: {
      int m = idx / KN_MAXK, k = idx % KN_MAXK;
      if (sub == 0 || m < 0 || m >= h->n_models[sub] || k >= K)
        return fail(KNPEMI_EINVAL, "field: bad I_ch index");
      *loc = {D.Ich + ((size_t)(h->moff[sub] + m) * KN_MAXK + k) * std::max(1, D.NQtot) + q0, 1, nq};
      return 0;
    }
    case KNPEMI_F_SOURCE:
      if (sub != 0 || idx < 0 || idx >= K - 1) return fail(KNPEMI_EINVAL, "field: f_source lives on the ECS");
      if (!D.fsrc) {
        int rc = dev_zeros(h, (size_t)(K - 1) * h->n_vert[0], &D.fsrc);
        if (rc) return rc;
      }
      *loc = {D.fsrc + (size_t)idx * h->n_vert[0], 1, nv}; return 0;
  }
  return fail(KNPEMI_EINVAL, "field: unknown field id");
}
}  // namespace

extern "C" int knpemi_set_field(knpemi_handle* h, int field, int sub, int idx, const double* host, size_t n) {
  if (h) h->gam_valid = false;
  if (!h || !host) return fail(KNPEMI_EINVAL, "knpemi_set_field: null argument");
  FieldLoc L;
  int rc = locate(h, field, sub, idx, &L);
  if (rc) return rc;
  if (n != L.n) return fail(KNPEMI_EINVAL, "knpemi_set_field: length does not match the function space");
  if (n == 0) return KNPEMI_OK;
  KN_HIP(hipSetDevice(h->device));
  if (L.stride == 1) {
    KN_HIP(hipMemcpyAsync(L.base, host, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
    return KNPEMI_OK;
  }
  KN_HIP(hipMemcpyAsync(h->d_stage, host, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  rc = kn_launch_field_scatter(h, h->d_stage, L.base, (int)n, L.stride);
  if (rc) return rc;
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_get_field(knpemi_handle* h, int field, int sub, int idx, double* host, size_t n) {
  if (!h || !host) return fail(KNPEMI_EINVAL, "knpemi_get_field: null argument");
  FieldLoc L;
  int rc = locate(h, field, sub, idx, &L);
  if (rc) return rc;
  if (n != L.n) return fail(KNPEMI_EINVAL, "knpemi_get_field: length does not match the function space");
  if (n == 0) return KNPEMI_OK;
  KN_HIP(hipSetDevice(h->device));
  const double* src = L.base;
  if (L.stride != 1) {
    rc = kn_launch_field_gather(h, L.base, L.stride, h->d_stage, (int)n);
    if (rc) return rc;
    src = h->d_stage;
  }
  KN_HIP(hipMemcpyAsync(host, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_trace(knpemi_handle* h, int sub, const double* u_e, const double* u_i, double* q_e,
                            double* q_i) {
  if (!h || !u_e || !u_i || !q_e || !q_i) return fail(KNPEMI_EINVAL, "knpemi_trace: null argument");
  if (sub < 1 || sub >= h->n_sub) return fail(KNPEMI_EINVAL, "knpemi_trace: sub must be a cellular sub-domain");
  const int nq = h->n_q[sub], ne = h->n_vert[0], ni = h->n_vert[sub];
  if (nq == 0) return KNPEMI_OK;
  if ((size_t)ne + ni + 2 * (size_t)nq > h->stage_len) return fail(KNPEMI_EINVAL, "knpemi_trace: staging buffer too small");
  KN_HIP(hipSetDevice(h->device));
  double* de = h->d_stage;
  double* di = de + ne;
  double* qe = di + ni;
  double* qi = qe + nq;
  KN_HIP(hipMemcpyAsync(de, u_e, (size_t)ne * sizeof(double), hipMemcpyHostToDevice, h->stream));
  KN_HIP(hipMemcpyAsync(di, u_i, (size_t)ni * sizeof(double), hipMemcpyHostToDevice, h->stream));
  int rc = kn_launch_trace(h, de, di, sub, qe, qi);
  if (rc) return rc;
  KN_HIP(hipMemcpyAsync(q_e, qe, (size_t)nq * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipMemcpyAsync(q_i, qi, (size_t)nq * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

// ---------------------------------------------------------------------------------------------------
// assembly + CSR access
// ---------------------------------------------------------------------------------------------------
extern "C" int knpemi_assemble_emi(knpemi_handle* h, int flags) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (!h->have_params) return fail(KNPEMI_EINVAL, "knpemi_assemble_emi: knpemi_set_params not called");
  KN_HIP(hipSetDevice(h->device));
  h->emi_flags = flags;
  if (flags & KNPEMI_ON_AUX_STREAM) {
    // fork: the auxiliary stream starts after everything enqueued on the main stream so far
    KN_HIP(hipEventRecord(h->ev_fork, h->stream));
    KN_HIP(hipStreamWaitEvent(h->aux, h->ev_fork, 0));
    h->cur = h->aux;
    int rc = kn_launch_emi_rows(h, flags);
    h->cur = h->stream;
    if (rc) return rc;
    KN_HIP(hipEventRecord(h->ev_join, h->aux));
    return KNPEMI_OK;
  }
  return kn_launch_emi_rows(h, flags);
}

extern "C" int knpemi_join(knpemi_handle* h) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join2, 0));
  return KNPEMI_OK;
}

extern "C" int knpemi_assemble_emi_membrane_rhs(knpemi_handle* h, int flags) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (!h->have_params) return fail(KNPEMI_EINVAL, "knpemi_assemble_emi_membrane_rhs: knpemi_set_params not called");
  KN_HIP(hipSetDevice(h->device));
  return kn_launch_emi_membrane_rhs(h, flags);
}

extern "C" int knpemi_assemble_knp_membrane_early(knpemi_handle* h, int flags) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (!h->have_params) return fail(KNPEMI_EINVAL, "knpemi_assemble_knp_membrane_early: knpemi_set_params not called");
  KN_HIP(hipSetDevice(h->device));
  int rc;
  if (flags & KNPEMI_ON_AUX_STREAM) {
    // after everything the main stream holds (the ODE output, the previous update), beside what comes next on it
    KN_HIP(hipEventRecord(h->ev_pre_fork, h->stream));
    KN_HIP(hipStreamWaitEvent(h->aux, h->ev_pre_fork, 0));
    h->cur = h->aux;
    rc = kn_launch_knp_membrane_pre(h, flags);
    h->cur = h->stream;
    if (rc) return rc;
    KN_HIP(hipEventRecord(h->ev_pre, h->aux));
    h->pre_pending = 1;
    return KNPEMI_OK;
  }
  return kn_launch_knp_membrane_pre(h, flags);
}

extern "C" int knpemi_assemble_knp(knpemi_handle* h, int flags) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (!h->have_params) return fail(KNPEMI_EINVAL, "knpemi_assemble_knp: knpemi_set_params not called");
  KN_HIP(hipSetDevice(h->device));
  // phi_M and I_ch come from the ODE sweeps, which may run on the auxiliary streams: the assembly is ordered after them
  // whether or not the caller has called knpemi_join (a wait on a completed or never-recorded event costs nothing)
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join2, 0));
  if (flags & KNPEMI_MEMBRANE_EARLY) {   // the integrals were prepared by knpemi_assemble_knp_membrane_early
    if (h->pre_pending) {
      KN_HIP(hipStreamWaitEvent(h->stream, h->ev_pre, 0));
      h->pre_pending = 0;
    }
    return kn_launch_knp_rows(h, flags);
  }
  if (!h->fuse_membrane) {   // stand-alone facet kernel: partial integrals through gam_e, the row kernel adds them
    // ... unless the launch that wrote the potential back has formed them already, for these fields and this scheme
    const int split = (flags & KNPEMI_NO_SPLITTING) ? 0 : 1;
    if (!(h->gam_valid && h->gam_split == split)) {
      int rc = kn_launch_knp_membrane(h, flags);
      if (rc) return rc;
    }
  }
  return kn_launch_knp_rows(h, flags);
}

extern "C" int knpemi_solve_emi(knpemi_handle* h, double rtol, double atol, int maxit, int* iters, double* relres) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (!(rtol >= 0) || !(atol >= 0) || maxit < 0) return fail(KNPEMI_EINVAL, "knpemi_solve_emi: bad tolerances");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));   // a matrix assembled on the auxiliary stream is complete
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join2, 0));  // ... and so is every sweep whose output the right-hand side holds
  return kn_solve_emi(h, rtol, atol, maxit, iters, relres);
}

extern "C" int knpemi_solve_knp(knpemi_handle* h, double rtol, double atol, int maxit, int* iters, double* relres) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (!(rtol >= 0) || !(atol >= 0) || maxit < 0) return fail(KNPEMI_EINVAL, "knpemi_solve_knp: bad tolerances");
  KN_HIP(hipSetDevice(h->device));
  // every writer of A_knp, b_knp and their inputs (the potential, phi_M, I_ch) is on the main stream or joined into it
  // before the set-up's device-to-host copy and the first iteration read them (round-3 advisor finding: this entry point
  // waited on neither join event)
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join2, 0));
  return kn_solve_knp(h, rtol, atol, maxit, iters, relres);
}

extern "C" int knpemi_extrapolate_guess(knpemi_handle* h, int which) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (which != KNPEMI_B_EMI && which != KNPEMI_B_KNP) return fail(KNPEMI_EINVAL, "knpemi_extrapolate_guess: unknown system");
  KN_HIP(hipSetDevice(h->device));
  return kn_extrapolate_guess(h, which);
}

extern "C" int knpemi_solver_setup(knpemi_handle* h, int which, int precond, double theta) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (which != KNPEMI_B_EMI && which != KNPEMI_B_KNP) return fail(KNPEMI_EINVAL, "knpemi_solver_setup: unknown system");
  if (precond != KNPEMI_PC_JACOBI && precond != KNPEMI_PC_AMG)
    return fail(KNPEMI_EINVAL, "knpemi_solver_setup: unknown preconditioner");
  KnAmg& G = which == KNPEMI_B_EMI ? h->amg_emi : h->amg_knp;
  (which == KNPEMI_B_EMI ? h->pc_emi : h->pc_knp) = precond;
  G.theta = theta > 0 ? theta : 0.08;
  G.built = false;
  return KNPEMI_OK;
}

extern "C" int knpemi_solver_info(knpemi_handle* h, int which, int* levels, double* op_complexity, int* builds) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (which != KNPEMI_B_EMI && which != KNPEMI_B_KNP) return fail(KNPEMI_EINVAL, "knpemi_solver_info: unknown system");
  const KnAmg& G = which == KNPEMI_B_EMI ? h->amg_emi : h->amg_knp;
  if (levels) *levels = (int)G.lev.size();
  if (op_complexity) *op_complexity = G.op_complexity;
  if (builds) *builds = G.builds;
  return KNPEMI_OK;
}

extern "C" int knpemi_csr_dims(knpemi_handle* h, int which, int64_t* n_rows, int64_t* nnz) {
  if (!h || !n_rows || !nnz) return fail(KNPEMI_EINVAL, "knpemi_csr_dims: null argument");
  if (which == KNPEMI_A_EMI || which == KNPEMI_P_EMI) { *n_rows = h->dev.Ntot; *nnz = h->dev.nnz; }
  else if (which == KNPEMI_A_KNP) { *n_rows = (int64_t)(h->K - 1) * h->dev.Ntot; *nnz = (h->K - 1) * h->dev.nnzL; }
  else return fail(KNPEMI_EINVAL, "knpemi_csr_dims: unknown matrix");
  return KNPEMI_OK;
}

extern "C" int knpemi_get_csr_pattern(knpemi_handle* h, int which, int32_t* rowptr, int32_t* colind) {
  if (!h || !rowptr || !colind) return fail(KNPEMI_EINVAL, "knpemi_get_csr_pattern: null argument");
  if (which == KNPEMI_A_EMI || which == KNPEMI_P_EMI) {
    std::memcpy(rowptr, h->h_rowptr.data(), h->h_rowptr.size() * sizeof(int));
    std::memcpy(colind, h->h_colind.data(), h->h_colind.size() * sizeof(int));
    return KNPEMI_OK;
  }
  if (which != KNPEMI_A_KNP) return fail(KNPEMI_EINVAL, "knpemi_get_csr_pattern: unknown matrix");
  // block (sub, ion): rows/cols shifted to [c[0][0], c[0][1], c[1][0], ...] (pdeSolver.py:117)
  const int KS = h->K - 1;
  int64_t row = 0, pos = 0;
  rowptr[0] = 0;
  for (int s = 0; s < h->n_sub; ++s) {
    const int v0 = h->voff[s], nv = h->n_vert[s];
    for (int k = 0; k < KS; ++k) {
      const int64_t shift = (int64_t)KS * v0 + (int64_t)k * nv - v0;
      for (int v = 0; v < nv; ++v) {
        for (int p = h->h_rowptrL[v0 + v]; p < h->h_rowptrL[v0 + v + 1]; ++p)
          colind[pos++] = (int32_t)(h->h_colindL[p] + shift);
        rowptr[++row] = (int32_t)pos;
      }
    }
  }
  return KNPEMI_OK;
}

extern "C" int knpemi_device_csr(knpemi_handle* h, int which, const int32_t** rowptr,
                                 const int32_t** colind, const double** vals) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (which == KNPEMI_A_EMI || which == KNPEMI_P_EMI) {
    if (rowptr) *rowptr = h->dev.rowptr;
    if (colind) *colind = h->dev.colind;
    if (vals) *vals = which == KNPEMI_A_EMI ? h->dev.A_emi : h->dev.P_emi;
    return KNPEMI_OK;
  }
  if (which == KNPEMI_A_KNP) {  // values only: block b = (sub, ion) occupies a contiguous range
    if (rowptr) *rowptr = h->dev.rowptrL;
    if (colind) *colind = h->dev.colindL;
    if (vals) *vals = h->dev.A_knp;
    return KNPEMI_OK;
  }
  return fail(KNPEMI_EINVAL, "knpemi_device_csr: unknown matrix");
}

extern "C" int knpemi_get_csr_values(knpemi_handle* h, int which, double* vals) {
  if (!h || !vals) return fail(KNPEMI_EINVAL, "knpemi_get_csr_values: null argument");
  const double* src; size_t n;
  if (which == KNPEMI_A_EMI) { src = h->dev.A_emi; n = h->dev.nnz; }
  else if (which == KNPEMI_P_EMI) { src = h->dev.P_emi; n = h->dev.nnz; }
  else if (which == KNPEMI_A_KNP) { src = h->dev.A_knp; n = (size_t)(h->K - 1) * h->dev.nnzL; }
  else return fail(KNPEMI_EINVAL, "knpemi_get_csr_values: unknown matrix");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipMemcpyAsync(vals, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

// Values of an operator / right-hand side supplied by the caller (host arrays in the layout knpemi_get_csr_values /
// knpemi_get_rhs return): the counterpart of PETSc's MatSetValues / VecSetValues for a caller that brings its own system
// to the device solves of this library, and what the solver tests use to put the Krylov loops on prescribed systems.
extern "C" int knpemi_set_csr_values(knpemi_handle* h, int which, const double* vals) {
  if (!h || !vals) return fail(KNPEMI_EINVAL, "knpemi_set_csr_values: null argument");
  double* dst; size_t n;
  if (which == KNPEMI_A_EMI) { dst = h->dev.A_emi; n = h->dev.nnz; }
  else if (which == KNPEMI_P_EMI) { dst = h->dev.P_emi; n = h->dev.nnz; }
  else if (which == KNPEMI_A_KNP) { dst = h->dev.A_knp; n = (size_t)(h->K - 1) * h->dev.nnzL; }
  else return fail(KNPEMI_EINVAL, "knpemi_set_csr_values: unknown matrix");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));   // an assembly on the auxiliary stream writes the same array
  KN_HIP(hipMemcpyAsync(dst, vals, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_set_rhs(knpemi_handle* h, int which, const double* b) {
  if (!h || !b) return fail(KNPEMI_EINVAL, "knpemi_set_rhs: null argument");
  double* dst; size_t n;
  if (which == KNPEMI_B_EMI) { dst = h->dev.b_emi; n = h->dev.Ntot; }
  else if (which == KNPEMI_B_KNP) { dst = h->dev.b_knp; n = (size_t)(h->K - 1) * h->dev.Ntot; }
  else return fail(KNPEMI_EINVAL, "knpemi_set_rhs: unknown vector");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  KN_HIP(hipMemcpyAsync(dst, b, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_device_rhs(knpemi_handle* h, int which, const double** b) {
  if (!h || !b) return fail(KNPEMI_EINVAL, "null argument");
  if (which == KNPEMI_B_EMI) *b = h->dev.b_emi;
  else if (which == KNPEMI_B_KNP) *b = h->dev.b_knp;
  else return fail(KNPEMI_EINVAL, "knpemi_device_rhs: unknown vector");
  return KNPEMI_OK;
}

extern "C" int knpemi_get_rhs(knpemi_handle* h, int which, double* b) {
  if (!h || !b) return fail(KNPEMI_EINVAL, "knpemi_get_rhs: null argument");
  const double* src; size_t n;
  if (which == KNPEMI_B_EMI) { src = h->dev.b_emi; n = h->dev.Ntot; }
  else if (which == KNPEMI_B_KNP) { src = h->dev.b_knp; n = (size_t)(h->K - 1) * h->dev.Ntot; }
  else return fail(KNPEMI_EINVAL, "knpemi_get_rhs: unknown vector");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipMemcpyAsync(b, src, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_set_solution(knpemi_handle* h, int which, const double* x, int on_device) {
  if (!h || !x) return fail(KNPEMI_EINVAL, "knpemi_set_solution: null argument");
  KN_HIP(hipSetDevice(h->device));
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  KnDev& D = h->dev;
  if (which == KNPEMI_B_EMI) {
    const double* src = x;
    if (!on_device) {
      KN_HIP(hipMemcpyAsync(h->d_stage, x, (size_t)D.Ntot * sizeof(double), kind, h->stream));
      src = h->d_stage;
    }
    // a solution pasted on the device stands for the write-back of knpemi_solve_emi: the same launch, facet integrals of
    // b_knp included (KNPEMI_OPT_FOLD_MEMBRANE)
    int rc = (on_device && h->fold_membrane && h->have_params && !h->fuse_membrane && !h->dist.on)
                 ? kn_launch_emi_writeback_membrane(h, src, nullptr, 0, 0.0, nullptr)
                 : kn_launch_field_scatter(h, src, D.VR + 7, D.Ntot, KN_REC);
    if (rc) return rc;
  } else if (which == KNPEMI_B_KNP) {
    const int KS = h->K - 1;
    if (on_device) {   // one launch, as the write-back of knpemi_solve_knp
      if (!h->have_params) return fail(KNPEMI_EINVAL, "knpemi_set_solution: knpemi_set_params not called");
      int rc = h->fuse_update ? kn_launch_knp_writeback_update(h, x) : kn_launch_knp_order(h, const_cast<double*>(x), 0);
      if (rc) return rc;
    } else
    for (int s = 0; s < h->n_sub; ++s)
      for (int k = 0; k < KS; ++k) {
        const size_t nv = h->n_vert[s];
        if (!nv) continue;
        KN_HIP(hipMemcpyAsync(D.csol + (size_t)k * D.Ntot + h->voff[s],
                              x + (size_t)KS * h->voff[s] + (size_t)k * nv, nv * sizeof(double), kind, h->stream));
      }
  } else return fail(KNPEMI_EINVAL, "knpemi_set_solution: unknown system");
  // KNPEMI_OPT_FUSE_UPDATE: whoever writes the KNP solution also performs the end-of-step update (the device write-back
  // above does it in the same launch); the host path launches it after its copies, so that a caller's own solver whose
  // result arrives through this entry point leaves c_prev, the eliminated ion and phi_M updated as well
  if (!on_device && which == KNPEMI_B_KNP && h->fuse_update) {
    if (!h->have_params) return fail(KNPEMI_EINVAL, "knpemi_set_solution: knpemi_set_params not called");
    int rc = kn_launch_update_pde(h);
    if (rc) return rc;
  }
  if (!on_device) KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_get_solution(knpemi_handle* h, int which, double* x) {
  if (!h || !x) return fail(KNPEMI_EINVAL, "knpemi_get_solution: null argument");
  KN_HIP(hipSetDevice(h->device));
  KnDev& D = h->dev;
  if (which == KNPEMI_B_EMI) {
    int rc = kn_launch_field_gather(h, D.VR + 7, KN_REC, h->d_stage, D.Ntot);
    if (rc) return rc;
    KN_HIP(hipMemcpyAsync(x, h->d_stage, (size_t)D.Ntot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  } else if (which == KNPEMI_B_KNP) {
    const int KS = h->K - 1;
    for (int s = 0; s < h->n_sub; ++s)
      for (int k = 0; k < KS; ++k) {
        const size_t nv = h->n_vert[s];
        if (!nv) continue;
        KN_HIP(hipMemcpyAsync(x + (size_t)KS * h->voff[s] + (size_t)k * nv,
                              D.csol + (size_t)k * D.Ntot + h->voff[s], nv * sizeof(double),
                              hipMemcpyDeviceToHost, h->stream));
      }
  } else return fail(KNPEMI_EINVAL, "knpemi_get_solution: unknown system");
  KN_HIP(hipStreamSynchronize(h->stream));
  return KNPEMI_OK;
}

// ---------------------------------------------------------------------------------------------------
// membrane ODE tables
// ---------------------------------------------------------------------------------------------------
namespace {
int ode_slot(knpemi_handle* h, int sub, int model, int need_bound) {
  if (!h || sub < 1 || sub >= h->n_sub || model < 0 || model >= h->n_models[sub]) {
    kn_set_error("membrane model index out of range");
    return -1;
  }
  int slot = h->moff[sub] + model;
  if (need_bound && !h->ode[slot].bound) {
    kn_set_error("membrane model not bound (knpemi_ode_bind)");
    return -1;
  }
  return slot;
}
}  // namespace

extern "C" int knpemi_ode_bind(knpemi_handle* h, int sub, int model, int model_id, int n_states, int n_params) {
  int slot = ode_slot(h, sub, model, 0);
  if (slot < 0) return KNPEMI_EINVAL;
  static const int ns_of[3] = {4, 4, 1}, np_of[3] = {22, 22, 23};
  if (model_id < 0 || model_id > 2) return fail(KNPEMI_EINVAL, "knpemi_ode_bind: unknown model id");
  if (n_states != ns_of[model_id] || n_params != np_of[model_id])
    return fail(KNPEMI_EINVAL, "knpemi_ode_bind: state/parameter count does not match the model");
  KN_HIP(hipSetDevice(h->device));
  KnOdeModel& m = h->ode[slot];
  if (m.bound) return fail(KNPEMI_EINVAL, "knpemi_ode_bind: model already bound");
  m.sub = sub; m.model_id = model_id; m.n_states = n_states; m.n_params = n_params;
  m.nq = h->n_q[sub];
  int rc;
  if ((rc = dev_zeros(h, (size_t)n_states * m.nq, &m.d_states))) return rc;
  if ((rc = dev_zeros(h, (size_t)n_params * m.nq, &m.d_params))) return rc;
  // one slot per workgroup of the sweep (small sweeps run with fewer dofs per wave, up to one wave per SIMD: kernels_ode.hip)
  m.n_stat_blocks = std::max((int)(((size_t)m.nq * n_states + 63) / 64), std::min(m.nq, 1024)) + 1;
  if ((rc = dev_zeros(h, 3 * (size_t)m.n_stat_blocks, &m.d_stats))) return rc;
  m.bound = 1;
  return KNPEMI_OK;
}

extern "C" int knpemi_ode_bind_source(knpemi_handle* h, int sub, int model, int n_states, int n_params,
                                      const char* rhs_source) {
  int slot = ode_slot(h, sub, model, 0);
  if (slot < 0) return KNPEMI_EINVAL;
  if (!rhs_source || n_states < 1 || n_states > 16 || n_params < 1 || n_params > 128)
    return fail(KNPEMI_EINVAL, "knpemi_ode_bind_source: bad argument (1..16 states, 1..128 parameters)");
  KN_HIP(hipSetDevice(h->device));
  KnOdeModel& m = h->ode[slot];
  if (m.bound) return fail(KNPEMI_EINVAL, "knpemi_ode_bind_source: model already bound");
  int rc = kn_rtc_bind(h, m, n_states, n_params, rhs_source);
  if (rc) return rc;
  m.sub = sub; m.model_id = -1; m.n_states = n_states; m.n_params = n_params;
  m.nq = h->n_q[sub];
  if ((rc = dev_zeros(h, (size_t)n_states * m.nq, &m.d_states))) return rc;
  if ((rc = dev_zeros(h, (size_t)n_params * m.nq, &m.d_params))) return rc;
  m.n_stat_blocks = std::max((int)(((size_t)m.nq * n_states + 63) / 64), std::min(m.nq, 1024)) + 1;
  if ((rc = dev_zeros(h, 3 * (size_t)m.n_stat_blocks, &m.d_stats))) return rc;
  m.bound = 1;
  return KNPEMI_OK;
}

static void transpose(const double* src, double* dst, int rows, int cols) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) dst[(size_t)c * rows + r] = src[(size_t)r * cols + c];
}

extern "C" int knpemi_ode_set_tables(knpemi_handle* h, int sub, int model, const double* states, const double* params) {
  int slot = ode_slot(h, sub, model, 1);
  if (slot < 0) return KNPEMI_EINVAL;
  KnOdeModel& m = h->ode[slot];
  if (m.nq == 0) return KNPEMI_OK;
  KN_HIP(hipSetDevice(h->device));
  std::vector<double> t;
  if (states) {
    t.resize((size_t)m.nq * m.n_states);
    transpose(states, t.data(), m.nq, m.n_states);
    KN_HIP(hipMemcpyAsync(m.d_states, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
  }
  if (params) {
    t.resize((size_t)m.nq * m.n_params);
    transpose(params, t.data(), m.nq, m.n_params);
    KN_HIP(hipMemcpyAsync(m.d_params, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
  }
  return KNPEMI_OK;
}

extern "C" int knpemi_ode_get_tables(knpemi_handle* h, int sub, int model, double* states, double* params) {
  int slot = ode_slot(h, sub, model, 1);
  if (slot < 0) return KNPEMI_EINVAL;
  KnOdeModel& m = h->ode[slot];
  if (m.nq == 0) return KNPEMI_OK;
  KN_HIP(hipSetDevice(h->device));
  std::vector<double> t;
  if (states) {
    t.resize((size_t)m.nq * m.n_states);
    KN_HIP(hipMemcpyAsync(t.data(), m.d_states, t.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
    transpose(t.data(), states, m.n_states, m.nq);
  }
  if (params) {
    t.resize((size_t)m.nq * m.n_params);
    KN_HIP(hipMemcpyAsync(t.data(), m.d_params, t.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
    transpose(t.data(), params, m.n_params, m.nq);
  }
  return KNPEMI_OK;
}

extern "C" int knpemi_ode_set_stimulus(knpemi_handle* h, int sub, int model, const uint8_t* mask,
                                       int n_pairs, const int32_t* param_idx, const double* values) {
  int slot = ode_slot(h, sub, model, 1);
  if (slot < 0) return KNPEMI_EINVAL;
  KnOdeModel& m = h->ode[slot];
  if (n_pairs < 0 || n_pairs > 8) return fail(KNPEMI_EINVAL, "knpemi_ode_set_stimulus: at most 8 stimulus entries");
  KN_HIP(hipSetDevice(h->device));
  for (int i = 0; i < n_pairs; ++i) {
    if (param_idx[i] < 0 || param_idx[i] >= m.n_params)
      return fail(KNPEMI_EINVAL, "knpemi_ode_set_stimulus: parameter index out of range");
    m.stim_idx[i] = param_idx[i];
    m.stim_val[i] = values[i];
  }
  m.n_stim = n_pairs;
  if (mask && m.nq > 0) {
    if (!m.d_mask) {
      int rc = dev_zeros(h, (size_t)m.nq, &m.d_mask);
      if (rc) return rc;
    }
    KN_HIP(hipMemcpyAsync(m.d_mask, mask, (size_t)m.nq, hipMemcpyHostToDevice, h->stream));
    KN_HIP(hipStreamSynchronize(h->stream));
  } else if (m.d_mask) {
    KN_HIP(hipMemsetAsync(m.d_mask, 1, (size_t)m.nq, h->stream));
  }
  return KNPEMI_OK;
}

extern "C" int knpemi_ode_step(knpemi_handle* h, int sub, int model, double t0, double dt, double rtol,
                               double atol, int flags, const int32_t* ion_param, int v_index) {
  int slot = ode_slot(h, sub, model, 1);
  if (slot < 0) return KNPEMI_EINVAL;
  if (!ion_param) return fail(KNPEMI_EINVAL, "knpemi_ode_step: ion_param is required");
  KnOdeModel& m = h->ode[slot];
  for (int i = 0; i < 3 * h->K; ++i)
    if (ion_param[i] < 0 || ion_param[i] >= m.n_params)
      return fail(KNPEMI_EINVAL, "knpemi_ode_step: parameter index out of range");
  if (v_index < 0 || v_index >= m.n_states) return fail(KNPEMI_EINVAL, "knpemi_ode_step: bad V index");
  if (!(dt > 0) || !(rtol >= 0) || !(atol >= 0) || (rtol == 0 && atol == 0))
    return fail(KNPEMI_EINVAL, "knpemi_ode_step: bad dt / tolerances");
  KN_HIP(hipSetDevice(h->device));
  h->gam_valid = false;      // phi_M and the channel currents change
  if (flags & (KNPEMI_ODE_ON_AUX_STREAM | KNPEMI_ODE_ON_AUX2_STREAM)) {
    const bool second = (flags & KNPEMI_ODE_ON_AUX2_STREAM) != 0;
    hipStream_t side = second ? h->aux2 : h->aux;
    KN_HIP(hipEventRecord(h->ev_fork, h->stream));
    KN_HIP(hipStreamWaitEvent(side, h->ev_fork, 0));
    h->cur = side;
    int rc = kn_launch_ode_step(h, slot, t0, dt, rtol, atol, flags, ion_param, v_index);
    h->cur = h->stream;
    if (rc) return rc;
    KN_HIP(hipEventRecord(second ? h->ev_join2 : h->ev_join, side));
    return KNPEMI_OK;
  }
  return kn_launch_ode_step(h, slot, t0, dt, rtol, atol, flags, ion_param, v_index);
}

extern "C" int knpemi_ode_stats(knpemi_handle* h, int sub, int model, int64_t* n_rhs, int64_t* n_steps, int32_t* n_failed) {
  int slot = ode_slot(h, sub, model, 1);
  if (slot < 0) return KNPEMI_EINVAL;
  unsigned long long st[3] = {0, 0, 0};
  KN_HIP(hipSetDevice(h->device));
  KnOdeModel& mo = h->ode[slot];
  std::vector<unsigned long long> part(3 * (size_t)mo.n_stat_blocks);
  KN_HIP(hipStreamSynchronize(h->aux));   // the sweep may run on an auxiliary stream
  KN_HIP(hipStreamSynchronize(h->aux2));
  KN_HIP(hipMemcpyAsync(part.data(), mo.d_stats, part.size() * sizeof(part[0]), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipMemsetAsync(mo.d_stats, 0, part.size() * sizeof(part[0]), h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  for (size_t i = 0; i < part.size(); ++i) st[i % 3] += part[i];
  if (n_rhs) *n_rhs = (int64_t)st[0];
  if (n_steps) *n_steps = (int64_t)st[1];
  if (n_failed) *n_failed = (int32_t)st[2];
  if (st[2]) return fail(KNPEMI_EODE, "LSODA failed on at least one membrane dof (odeSolver.py:121 `assert success`)");
  return KNPEMI_OK;
}

extern "C" int knpemi_debug_ode_stamps(knpemi_handle* h, int sub, int model, uint64_t* out, int max_blocks) {
  int slot = ode_slot(h, sub, model, 1);
  if (slot < 0) return KNPEMI_EINVAL;
  KnOdeModel& m = h->ode[slot];
  if (!m.d_stamps || !out) return fail(KNPEMI_EINVAL, "knpemi_debug_ode_stamps: run with KNPEMI_ODE_STAMPS=1");
  const int nb = std::min(max_blocks, m.n_stat_blocks - 1);
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamSynchronize(h->aux));
  KN_HIP(hipStreamSynchronize(h->stream));
  KN_HIP(hipMemcpy(out, m.d_stamps, (size_t)nb * 24 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return nb;
}

extern "C" int knpemi_update_pde(knpemi_handle* h) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (!h->have_params) return fail(KNPEMI_EINVAL, "knpemi_update_pde: knpemi_set_params not called");
  KN_HIP(hipSetDevice(h->device));
  return kn_launch_update_pde(h);
}

extern "C" int knpemi_set_distributed(knpemi_handle* h, const uint8_t* owned, void* reduce_buf_dev,
                                      knpemi_allreduce_fn allreduce, knpemi_halo_fn halo, void* ctx) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  KnDist& d = h->dist;
  h->amg_emi.built = false;      // the preconditioner changes with the ownership
  h->amg_knp.built = false;
  if (!owned) { d.on = false; return KNPEMI_OK; }
  if (!reduce_buf_dev || !allreduce || !halo)
    return fail(KNPEMI_EINVAL, "knpemi_set_distributed: reduction buffer and both communication hooks are required");
  KN_HIP(hipSetDevice(h->device));
  const int Ntot = h->dev.Ntot, KS = h->K - 1;
  d.h_owned_emi.assign(owned, owned + Ntot);
  d.h_owned_knp.assign((size_t)KS * Ntot, 0);
  for (int s = 0; s < h->n_sub; ++s) {
    const int v0 = h->voff[s], nv = h->n_vert[s];
    for (int k = 0; k < KS; ++k)
      for (int v = 0; v < nv; ++v) d.h_owned_knp[(size_t)KS * v0 + (size_t)k * nv + v] = owned[v0 + v];
  }
  int rc;
  const uint8_t* p = nullptr;
  if ((rc = dev_upload(h, d.h_owned_emi, &p))) return rc;
  d.d_owned_emi = const_cast<uint8_t*>(p);
  if ((rc = dev_upload(h, d.h_owned_knp, &p))) return rc;
  d.d_owned_knp = const_cast<uint8_t*>(p);
  d.d_red = static_cast<double*>(reduce_buf_dev);
  d.allreduce = allreduce; d.halo = halo; d.ctx = ctx;
  // number of owned unknowns of the EMI system over all ranks (mean of the constant null space)
  double cnt = 0.0;
  for (int i = 0; i < Ntot; ++i) cnt += owned[i] ? 1.0 : 0.0;
  KN_HIP(hipMemcpyAsync(d.d_red, &cnt, sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (allreduce(ctx, 1)) return fail(KNPEMI_EHIP, "knpemi_set_distributed: allreduce hook failed");
  KN_HIP(hipMemcpyAsync(&cnt, d.d_red, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  KN_HIP(hipStreamSynchronize(h->stream));
  d.n_owned_global = cnt;
  d.on = true;
  return KNPEMI_OK;
}

extern "C" int knpemi_set_distributed_coarse(knpemi_handle* h, int rank, int world) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  KnDist& d = h->dist;
  if (world <= 1) { d.nc = 0; d.coarse_built = false; return KNPEMI_OK; }
  if (rank < 0 || rank >= world) return fail(KNPEMI_EINVAL, "knpemi_set_distributed_coarse: bad rank");
  if (world * h->n_sub > KN_COARSE_MAX)
    return fail(KNPEMI_EINVAL, "knpemi_set_distributed_coarse: more than 64 (rank, sub-domain) aggregates");
  if (!d.on) return fail(KNPEMI_EINVAL, "knpemi_set_distributed_coarse: call knpemi_set_distributed first");
  KN_HIP(hipSetDevice(h->device));
  d.rank = rank; d.world = world;
  // as many nodes per (rank, sub-domain) as the coarse size allows: k slices, k + 1 hat functions (k = 0: one constant)
  const int nodes = std::max(1, KN_COARSE_MAX / (world * h->n_sub));
  d.k = nodes - 1;
  d.nl = h->n_sub * nodes;
  d.nc = world * d.nl;
  d.coarse_built = false;
  // the owned vertices of a sub-domain along the longest axis of their bounding box: slice b, position t in it ->
  // weight 1 - t in node b, t in node b + 1
  const int Ntot = h->dev.Ntot;
  std::vector<double> rec((size_t)Ntot * KN_REC);
  KN_HIP(hipMemcpy(rec.data(), h->dev.VR, rec.size() * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<int> agg_of(Ntot, -1);
  std::vector<double> agg_w(Ntot, 0.0);
  for (int s = 0; s < h->n_sub; ++s) {
    const int v0 = h->voff[s], v1 = v0 + h->n_vert[s];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = v0; v < v1; ++v) {
      if (!d.h_owned_emi[v]) continue;
      for (int a = 0; a < h->gdim; ++a) { lo[a] = std::min(lo[a], rec[(size_t)v * KN_REC + a]); hi[a] = std::max(hi[a], rec[(size_t)v * KN_REC + a]); }
    }
    int ax = 0;
    for (int a = 1; a < h->gdim; ++a) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
    const double len = hi[ax] - lo[ax];
    for (int v = v0; v < v1; ++v) {
      if (!d.h_owned_emi[v]) continue;
      int bin = 0;
      double t = 0.0;
      if (d.k > 0 && len > 0) {
        const double u = (rec[(size_t)v * KN_REC + ax] - lo[ax]) / len * d.k;
        bin = std::min(d.k - 1, std::max(0, (int)u));
        t = std::min(1.0, std::max(0.0, u - bin));
      }
      agg_of[v] = s * nodes + bin;
      agg_w[v] = t;
    }
  }
  std::vector<int> ptr(d.nl + 1, 0), idx;
  std::vector<double> wts;
  for (int v = 0; v < Ntot; ++v)
    if (agg_of[v] >= 0) {
      ++ptr[agg_of[v] + 1];
      if (agg_w[v] > 0.0) ++ptr[agg_of[v] + 2];
    }
  for (int a = 0; a < d.nl; ++a) ptr[a + 1] += ptr[a];
  idx.resize(ptr[d.nl]);
  wts.resize(ptr[d.nl]);
  {
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int v = 0; v < Ntot; ++v)
      if (agg_of[v] >= 0) {
        const int a = agg_of[v];
        idx[fill[a]] = v; wts[fill[a]++] = 1.0 - agg_w[v];
        if (agg_w[v] > 0.0) { idx[fill[a + 1]] = v; wts[fill[a + 1]++] = agg_w[v]; }
      }
  }
  int rc;
  const int* p = nullptr;
  const double* pd = nullptr;
  if ((rc = dev_upload(h, agg_of, &p))) return rc;
  d.d_agg_of = const_cast<int*>(p);
  if ((rc = dev_upload(h, agg_w, &pd))) return rc;
  d.d_agg_w = const_cast<double*>(pd);
  if ((rc = dev_upload(h, ptr, &p))) return rc;
  d.d_agg_ptr = const_cast<int*>(p);
  if ((rc = dev_upload(h, idx, &p))) return rc;
  d.d_agg_idx = const_cast<int*>(p);
  if ((rc = dev_upload(h, wts, &pd))) return rc;
  d.d_agg_wt = const_cast<double*>(pd);
  return KNPEMI_OK;
}

extern "C" int knpemi_vec_gather(knpemi_handle* h, const void* vec_dev, const int32_t* idx_dev, int n, void* buf_dev) {
  if (!h || n < 0 || (n > 0 && (!vec_dev || !idx_dev || !buf_dev))) return fail(KNPEMI_EINVAL, "knpemi_vec_gather: bad argument");
  KN_HIP(hipSetDevice(h->device));
  return kn_launch_vec_index(h, static_cast<double*>(const_cast<void*>(vec_dev)), idx_dev, n, static_cast<double*>(buf_dev), 1);
}

extern "C" int knpemi_vec_scatter(knpemi_handle* h, void* vec_dev, const int32_t* idx_dev, int n, const void* buf_dev) {
  if (!h || n < 0 || (n > 0 && (!vec_dev || !idx_dev || !buf_dev))) return fail(KNPEMI_EINVAL, "knpemi_vec_scatter: bad argument");
  KN_HIP(hipSetDevice(h->device));
  return kn_launch_vec_index(h, static_cast<double*>(vec_dev), idx_dev, n, static_cast<double*>(const_cast<void*>(buf_dev)), 0);
}

extern "C" int knpemi_set_option(knpemi_handle* h, int option, int value) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  if (option == KNPEMI_OPT_FUSE_UPDATE) { h->fuse_update = value ? 1 : 0; return KNPEMI_OK; }
  if (option == KNPEMI_OPT_FUSE_MEMBRANE) {
    if (value && h->blocks_clustered)
      return fail(KNPEMI_EINVAL, "KNPEMI_OPT_FUSE_MEMBRANE needs row blocks of consecutive rows (create the handle with KNPEMI_BLOCK_CLASSIC=1)");
    h->fuse_membrane = value ? 1 : 0;
    return KNPEMI_OK;
  }
  if (option == KNPEMI_OPT_PROFILE_STRIDE) {   // the next launch of every kernel is a bracketed one
    h->prof_stride = value > 1 ? value : 1;
    for (unsigned& c : h->prof_count) c = 0;
    return KNPEMI_OK;
  }
  if (option == KNPEMI_OPT_KNP_METHOD) {
    if (value != 0 && value != 1) return fail(KNPEMI_EINVAL, "KNPEMI_OPT_KNP_METHOD: 0 (BiCGStab) or 1 (GMRES)");
    h->knp_method = value;
    return KNPEMI_OK;
  }
  if (option == KNPEMI_OPT_EMI_NORM) {
    if (value != 0 && value != 1) return fail(KNPEMI_EINVAL, "KNPEMI_OPT_EMI_NORM: 0 (true residual) or 1 (preconditioned)");
    h->emi_norm_pre = value;
    return KNPEMI_OK;
  }
  if (option == KNPEMI_OPT_FOLD_MEMBRANE) { h->fold_membrane = value ? 1 : 0; h->gam_valid = false; return KNPEMI_OK; }
  if (option == KNPEMI_OPT_KNP_MIN_IT) {
    if (value < 0) return fail(KNPEMI_EINVAL, "KNPEMI_OPT_KNP_MIN_IT: negative");
    h->knp_min_it = value;
    return KNPEMI_OK;
  }
  return fail(KNPEMI_EINVAL, "knpemi_set_option: unknown option");
}

extern "C" int knpemi_halo_width(knpemi_handle* h, int kind) {
  if (!h) return KNPEMI_EINVAL;
  return kind == 0 ? 5 : 1 + KN_MAXK * h->moff[h->n_sub];
}

extern "C" int knpemi_halo_pack(knpemi_handle* h, int kind, const int32_t* idx_dev, int n, double* buf_dev) {
  if (!h || (n > 0 && (!idx_dev || !buf_dev)) || n < 0 || kind < 0 || kind > 1)
    return fail(KNPEMI_EINVAL, "knpemi_halo_pack: bad argument");
  KN_HIP(hipSetDevice(h->device));
  return kn_launch_halo(h, kind, 1, idx_dev, n, buf_dev);
}

extern "C" int knpemi_halo_unpack(knpemi_handle* h, int kind, const int32_t* idx_dev, int n, const double* buf_dev) {
  if (!h || (n > 0 && (!idx_dev || !buf_dev)) || n < 0 || kind < 0 || kind > 1)
    return fail(KNPEMI_EINVAL, "knpemi_halo_unpack: bad argument");
  KN_HIP(hipSetDevice(h->device));
  return kn_launch_halo(h, kind, 0, idx_dev, n, const_cast<double*>(buf_dev));
}

extern "C" int knpemi_profile(knpemi_handle* h, uint32_t kernel_mask) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamSynchronize(h->stream));
  h->prof_mask = kernel_mask;
  for (int k = 0; k < KNPEMI_N_KERNELS; ++k) h->prof_used[k] = 0;
  return KNPEMI_OK;
}

extern "C" int knpemi_profile_read(knpemi_handle* h, int kernel, int64_t* launches, double* total_ms) {
  if (!h || kernel < 0 || kernel >= KNPEMI_N_KERNELS || !launches || !total_ms)
    return fail(KNPEMI_EINVAL, "knpemi_profile_read: bad argument");
  KN_HIP(hipSetDevice(h->device));
  KN_HIP(hipStreamSynchronize(h->stream));
  KN_HIP(hipStreamSynchronize(h->aux));
  KN_HIP(hipStreamSynchronize(h->aux2));
  double sum = 0.0;
  for (size_t i = 0; i + 1 < h->prof_used[kernel]; i += 2) {
    float f = 0.f;
    KN_HIP(hipEventElapsedTime(&f, h->prof_ev[kernel][i], h->prof_ev[kernel][i + 1]));
    sum += f;
  }
  *launches = (int64_t)(h->prof_used[kernel] / 2);
  *total_ms = sum;
  h->prof_used[kernel] = 0;
  return KNPEMI_OK;
}

extern "C" int knpemi_timer_start(knpemi_handle* h) {
  if (!h) return fail(KNPEMI_EINVAL, "null handle");
  KN_HIP(hipEventRecord(h->ev0, h->stream));
  return KNPEMI_OK;
}

extern "C" int knpemi_timer_stop_ms(knpemi_handle* h, double* ms) {
  if (!h || !ms) return fail(KNPEMI_EINVAL, "null argument");
  KN_HIP(hipEventRecord(h->ev1, h->stream));
  KN_HIP(hipEventSynchronize(h->ev1));
  float f = 0.f;
  KN_HIP(hipEventElapsedTime(&f, h->ev0, h->ev1));
  *ms = f;
  return KNPEMI_OK;
}

extern "C" int knpemi_debug_geometry(knpemi_handle* h, int* flags) {
  if (!h || !flags) return fail(KNPEMI_EINVAL, "knpemi_debug_geometry: null argument");
  *flags = (h->tet_uniform ? 1 : 0) | (h->hex_affine ? 2 : 0) | (h->hex_uniform ? 4 : 0);
  return KNPEMI_OK;
}
