// Host build of the PRODUCT's AMG set-up code (knp-emi-fenics-x_amd/csrc/amg_host.h) for the CPU tests: C entry points
// over the aggregation, prolongator and coarsest-level routines (tests/test_amg_host.py).
#include "../../knp-emi-fenics-x_amd/csrc/amg_host.h"

#include <cstring>

using namespace kn_amg_host;

static HostCsr make(int n, const int* rp, const int* ci, const double* v) {
  HostCsr A;
  A.n = A.m = n;
  A.rp.assign(rp, rp + n + 1);
  A.ci.assign(ci, ci + rp[n]);
  A.v.assign(v, v + rp[n]);
  return A;
}

extern "C" {

// mode 0: greedy aggregation on -a_ij; 1: |a_ij|; 2: aggregate_apart (theta_pos = 0.2)
int amg_host_aggregate(int n, const int* rp, const int* ci, const double* v, double theta, int mode, int* agg_out) {
  const HostCsr A = make(n, rp, ci, v);
  const std::vector<double> d = diagonal(A);
  std::vector<int> agg;
  const int na = mode == 2 ? aggregate_apart(A, d, theta, 0.2, agg) : aggregate(A, d, theta, mode == 0, agg);
  std::memcpy(agg_out, agg.data(), sizeof(int) * n);
  return na;
}

int amg_host_split(int n, const int* rp, const int* ci, const double* v, double theta, const unsigned char* owned, int* agg, int na) {
  const HostCsr A = make(n, rp, ci, v);
  const std::vector<double> d = diagonal(A);
  std::vector<int> a(agg, agg + n);
  const int cnt = split_aggregates(A, d, theta, owned, a, na);
  std::memcpy(agg, a.data(), sizeof(int) * n);
  return cnt;
}

// dense n x na prolongator (row-major) of the given aggregates: w = 0 the embedding, filter_theta > 0 the filtered smoothing
void amg_host_prolongator(int n, const int* rp, const int* ci, const double* v, const int* agg, int na, double w, double filter_theta,
                          double* P_out) {
  const HostCsr A = make(n, rp, ci, v);
  const std::vector<double> d = diagonal(A);
  const HostCsr P = smoothed_prolongator(A, d, std::vector<int>(agg, agg + n), na, w, filter_theta);
  std::memset(P_out, 0, sizeof(double) * (size_t)n * na);
  for (int i = 0; i < n; ++i)
    for (int j = P.rp[i]; j < P.rp[i + 1]; ++j) P_out[(size_t)i * na + P.ci[j]] = P.v[j];
}

// Galerkin product P^T A P as a dense na x na matrix (spgemm + transpose, as the set-up forms it)
void amg_host_galerkin(int n, const int* rp, const int* ci, const double* v, const int* agg, int na, double w, double filter_theta,
                       double* Ac_out) {
  const HostCsr A = make(n, rp, ci, v);
  const std::vector<double> d = diagonal(A);
  const HostCsr P = smoothed_prolongator(A, d, std::vector<int>(agg, agg + n), na, w, filter_theta);
  const HostCsr Ac = spgemm(transpose(P), spgemm(A, P));
  std::memset(Ac_out, 0, sizeof(double) * (size_t)na * na);
  for (int i = 0; i < na; ++i)
    for (int j = Ac.rp[i]; j < Ac.rp[i + 1]; ++j) Ac_out[(size_t)i * na + Ac.ci[j]] = Ac.v[j];
}

int amg_host_dense_inverse(int n, const int* rp, const int* ci, const double* v, int singular, double* inv_out) {
  const HostCsr A = make(n, rp, ci, v);
  std::vector<double> inv;
  if (!dense_inverse(A, singular != 0, inv)) return 0;
  std::memcpy(inv_out, inv.data(), sizeof(double) * (size_t)n * n);
  return 1;
}

// the block-wise storage of the same inverse, expanded to a dense n x n array again; returns the number of stored values
// (negative: number of blocks, when asked through nb_out)
int amg_host_dense_inverse_blocks(int n, const int* rp, const int* ci, const double* v, int singular, double* inv_out, int* nb_out) {
  const HostCsr A = make(n, rp, ci, v);
  DenseInvBlocks c;
  if (!dense_inverse_blocks(A, singular != 0, c)) return 0;
  std::memset(inv_out, 0, sizeof(double) * (size_t)n * n);
  for (int b = 0; b < c.nb; ++b)
    for (int i = 0; i < c.size[b]; ++i)
      for (int j = 0; j < c.size[b]; ++j)
        inv_out[(size_t)(c.start[b] + i) * n + c.start[b] + j] = c.v[(size_t)c.off[b] + (size_t)i * c.size[b] + j];
  if (nb_out) *nb_out = c.nb;
  return (int)c.v.size();
}

// aggregates renumbered component by component (renumber_by_component)
void amg_host_renumber(int n, const int* rp, const int* ci, const double* v, int* agg, int na) {
  const HostCsr A = make(n, rp, ci, v);
  std::vector<int> a(agg, agg + n);
  renumber_by_component(A, a, na);
  std::memcpy(agg, a.data(), sizeof(int) * n);
}

int amg_host_outlier(int n, const int* rp, const int* ci, const double* v, double factor, int* row, int* col) {
  const HostCsr A = make(n, rp, ci, v);
  return find_outlier(A, diagonal(A), factor, row, col) ? 1 : 0;
}

double amg_host_rho(int n, const int* rp, const int* ci, const double* v) {
  const HostCsr A = make(n, rp, ci, v);
  return estimate_rho(A, diagonal(A));
}
}
