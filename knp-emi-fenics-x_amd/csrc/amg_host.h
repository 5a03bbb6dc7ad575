// Host side of the AMG set-up (kernels_amg.hip): strength graphs and aggregation, prolongators, sparse products, the dense
// inverse of the coarsest level.  Plain C++ without any device dependency, so that the CPU tests can exercise it
// (tests/native/amg_host_check.cpp, tests/test_amg_host.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <utility>
#include <vector>

namespace kn_amg_host {

struct HostCsr {
  int n = 0, m = 0;
  std::vector<int> rp, ci;
  std::vector<double> v;
};

// ---- row-parallel loops ---------------------------------------------------------------------------
// The products and row-wise constructions below treat every row independently, so they run on the host's threads
// (KNPEMI_AMG_THREADS, default: the hardware's, at most 16; small inputs stay on the calling thread).  Results do not depend
// on the number of threads: rows are computed exactly as the sequential loop computes them, and sums over rows are taken
// over KN_SUM_CHUNKS fixed row ranges in range order (a hierarchy must be reproducible from one machine to the next).
constexpr int KN_SUM_CHUNKS = 64;
constexpr int KN_PAR_MIN_ROWS = 20000;

inline int host_threads() {
  static const int n = [] {
    const char* e = getenv("KNPEMI_AMG_THREADS");
    int t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(t, 16));
  }();
  return n;
}

// fn(chunk, begin, end) for `chunks` contiguous ranges of [0, n), dealt to the threads round robin
template <class F>
void for_chunks(int n, int chunks, F&& fn) {
  chunks = std::max(1, std::min(chunks, n));
  auto range = [&](int c, int& b, int& e) { b = (int)((int64_t)n * c / chunks); e = (int)((int64_t)n * (c + 1) / chunks); };
  const int nt = n >= KN_PAR_MIN_ROWS ? std::min(host_threads(), chunks) : 1;
  if (nt <= 1) {
    for (int c = 0; c < chunks; ++c) { int b, e; range(c, b, e); fn(c, b, e); }
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] { for (int c = t; c < chunks; c += nt) { int b, e; range(c, b, e); fn(c, b, e); } });
  for (auto& x : th) x.join();
}

// A matrix whose rows are produced independently: row(i, scratch, cols, vals) appends row i's sorted columns and values; every
// thread owns one Scratch (constructed by make_scratch()).  The pieces are stitched together in row order.
template <class MakeScratch, class Row>
HostCsr build_rows(int n, int m, MakeScratch&& make_scratch, Row&& row) {
  HostCsr C;
  C.n = n; C.m = m;
  C.rp.assign(n + 1, 0);
  const int chunks = n >= KN_PAR_MIN_ROWS ? 4 * host_threads() : 1;
  std::vector<std::vector<int>> pc(chunks);
  std::vector<std::vector<double>> pv(chunks);
  for_chunks(n, chunks, [&](int c, int b, int e) {
    auto scratch = make_scratch();
    for (int i = b; i < e; ++i) {
      row(i, scratch, pc[c], pv[c]);
      C.rp[i + 1] = (int)pc[c].size();       // (within the chunk; shifted below)
    }
  });
  // (chunk c covers rows [n c / chunks, n (c + 1) / chunks): same ranges as for_chunks dealt out)
  const int used = std::max(1, std::min(chunks, n));
  size_t off = 0;
  for (int c = 0; c < used; ++c) {
    const int b = (int)((int64_t)n * c / used), e = (int)((int64_t)n * (c + 1) / used);
    for (int i = b; i < e; ++i) C.rp[i + 1] += (int)off;
    off += pc[c].size();
  }
  C.ci.resize(off); C.v.resize(off);
  for_chunks(used, used, [&](int c, int, int) {
    const int b = (int)((int64_t)n * c / used);
    const size_t at = (size_t)C.rp[b];
    std::copy(pc[c].begin(), pc[c].end(), C.ci.begin() + at);
    std::copy(pv[c].begin(), pv[c].end(), C.v.begin() + at);
  });
  return C;
}

// sum over rows of f(i), in KN_SUM_CHUNKS fixed ranges added up in range order
template <class F>
double sum_rows(int n, F&& f) {
  double part[KN_SUM_CHUNKS] = {0.0};
  for_chunks(n, KN_SUM_CHUNKS, [&](int c, int b, int e) {
    double s = 0.0;
    for (int i = b; i < e; ++i) s += f(i);
    part[c] = s;
  });
  double s = 0.0;
  for (int c = 0; c < KN_SUM_CHUNKS; ++c) s += part[c];
  return s;
}

// ---- host set-up ---------------------------------------------------------------------------------

inline std::vector<double> diagonal(const HostCsr& A) {
  std::vector<double> d(A.n, 1.0);
  for (int i = 0; i < A.n; ++i)
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j)
      if (A.ci[j] == i) d[i] = A.v[j];
  return d;
}

// First entry with |a_ij| > factor * sqrt(|a_ii a_jj|): for the operators this set-up sees (SPD potential systems, mass- and
// diffusion-dominated concentration systems with a drift perturbation) every off-diagonal entry is bounded by the geometric
// mean of its two diagonal entries up to a small factor, so an entry a thousand times beyond it is not physics but a
// corrupted value -- finite, so the non-finite test lets it pass, and large enough to collapse the damping of a level.
inline bool find_outlier(const HostCsr& A, const std::vector<double>& d, double factor, int* row, int* col) {
  for (int i = 0; i < A.n; ++i)
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
      const int c = A.ci[j];
      if (c == i || c < 0 || c >= A.n) continue;
      if (std::fabs(A.v[j]) > factor * std::sqrt(std::fabs(d[i] * d[c]))) { *row = i; *col = c; return true; }
    }
  return false;
}

// greedy aggregation on the strength graph (three passes: roots with free neighbourhoods, attach
// leftovers to a neighbouring aggregate, remaining isolated points become their own aggregates)
inline int aggregate(const HostCsr& A, const std::vector<double>& d, double theta, bool negative_only, std::vector<int>& agg) {
  const int n = A.n;
  std::vector<int> srp(n + 1, 0), sci;
  sci.reserve(A.ci.size());
  for (int i = 0; i < n; ++i) {
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
      const int c = A.ci[j];
      // negative_only: classical strength (-a_ij): the positive couplings of stretched Q1 cells do not carry
      // smooth error and must not glue aggregates together along the weak direction
      const double a = negative_only ? -A.v[j] : std::fabs(A.v[j]);
      if (c != i && a > 0.0 && a >= theta * std::sqrt(std::fabs(d[i] * d[c]))) sci.push_back(c);
    }
    srp[i + 1] = (int)sci.size();
  }
  agg.assign(n, -1);
  int na = 0;
  for (int i = 0; i < n; ++i) {
    if (agg[i] >= 0 || srp[i] == srp[i + 1]) continue;
    bool free_nb = true;
    for (int j = srp[i]; j < srp[i + 1] && free_nb; ++j) free_nb = agg[sci[j]] < 0;
    if (!free_nb) continue;
    agg[i] = na;
    for (int j = srp[i]; j < srp[i + 1]; ++j) agg[sci[j]] = na;
    ++na;
  }
  std::vector<int> pass1(agg);
  for (int i = 0; i < n; ++i) {
    if (pass1[i] >= 0) continue;
    for (int j = srp[i]; j < srp[i + 1]; ++j)
      if (pass1[sci[j]] >= 0) { agg[i] = pass1[sci[j]]; break; }
  }
  for (int i = 0; i < n; ++i) {
    if (agg[i] >= 0) continue;
    agg[i] = na;
    for (int j = srp[i]; j < srp[i + 1]; ++j)
      if (agg[sci[j]] < 0) agg[sci[j]] = na;
    ++na;
  }
  return na;
}

// Splits given aggregates into the connected components of the strong couplings inside them (union-find; the numbering
// of the components follows their lowest member, so the result does not depend on the order of the unions).  Unknowns
// that are not owned (identity rows of a rank's diagonal block) have no couplings: those of one given aggregate stay together.
inline int split_aggregates(const HostCsr& A, const std::vector<double>& d, double theta, const uint8_t* owned, std::vector<int>& agg, int na) {
  const int n = A.n;
  std::vector<int> parent(n);
  for (int i = 0; i < n; ++i) parent[i] = i;
  auto find = [&](int i) {
    while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; }
    return i;
  };
  auto unite = [&](int a, int b) {
    a = find(a); b = find(b);
    if (a != b) parent[std::max(a, b)] = std::min(a, b);
  };
  std::vector<int> first_ghost(na, -1);
  for (int i = 0; i < n; ++i) {
    if (owned && !owned[i]) {
      int& g = first_ghost[agg[i]];
      if (g < 0) g = i; else unite(g, i);
      continue;
    }
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
      const int c = A.ci[j];
      if (c == i || agg[c] != agg[i] || (owned && !owned[c])) continue;
      if (-A.v[j] >= theta * std::sqrt(std::fabs(d[i] * d[c]))) unite(i, c);
    }
  }
  std::vector<int> id(n, -1);
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    const int r = find(i);
    if (id[r] < 0) id[r] = cnt++;
    agg[i] = id[r];
  }
  return cnt;
}

// Aggregation that keeps strongly POSITIVELY coupled unknowns apart (KnAmg::positive_conflict).  On the first coarse level of
// the DG systems on stretched hexahedra the two ends of a cell are coupled by +0.5 sqrt(a_ii a_jj) (the mass-like factor
// of the long direction): the smooth error takes independent values there, but each end is negatively coupled to the
// in-plane neighbours of the other one just above the strength threshold, and the plain greedy pass glues the two
// cross-sections together -- every variation along the long direction is then lost to the coarse space (convergence
// factor 0.97 of the two-level cycle on that operator).  Rule: an unknown does not join an aggregate that holds a
// strongly positive partner of it; neighbours are taken in the order of their strength, so the in-plane ones come first.
inline int aggregate_apart(const HostCsr& A, const std::vector<double>& d, double theta, double theta_pos, std::vector<int>& agg) {
  const int n = A.n;
  std::vector<int> srp(n + 1, 0), sci, prp(n + 1, 0), pci;
  std::vector<std::pair<double, int>> row;
  for (int i = 0; i < n; ++i) {
    row.clear();
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
      const int c = A.ci[j];
      if (c == i) continue;
      const double s = std::sqrt(std::fabs(d[i] * d[c]));
      if (-A.v[j] > 0.0 && -A.v[j] >= theta * s) row.emplace_back(A.v[j], c);
      else if (A.v[j] >= theta_pos * s) pci.push_back(c);
    }
    std::sort(row.begin(), row.end());     // most negative first; ties by column: reproducible
    for (auto& e : row) sci.push_back(e.second);
    srp[i + 1] = (int)sci.size();
    prp[i + 1] = (int)pci.size();
  }
  agg.assign(n, -1);
  std::vector<std::vector<int>> members;
  auto conflicts = [&](int k, const std::vector<int>& mem) {
    for (int j = prp[k]; j < prp[k + 1]; ++j)
      for (int m : mem) if (m == pci[j]) return true;
    return false;
  };
  for (int i = 0; i < n; ++i) {
    if (agg[i] >= 0 || srp[i] == srp[i + 1]) continue;
    // a root needs the neighbours it would TAKE to be free, not all its strong neighbours: the ones it leaves out (the
    // in-plane neighbours of its positive partner) belong to the other cross-section, and waiting for them kept every
    // vertex of the second cross-section from becoming a root (its unknowns then joined the first one's aggregates as
    // leftovers; prototype on the r = 0 hexahedral box: 16 -> 11 CG iterations)
    std::vector<int> mem{i};
    for (int j = srp[i]; j < srp[i + 1]; ++j) if (!conflicts(sci[j], mem)) mem.push_back(sci[j]);
    bool free_nb = true;
    for (size_t k = 1; k < mem.size() && free_nb; ++k) free_nb = agg[mem[k]] < 0;
    if (!free_nb) continue;
    for (int k : mem) agg[k] = (int)members.size();
    members.push_back(std::move(mem));
  }
  std::vector<int> pass1(agg);
  for (int i = 0; i < n; ++i) {
    if (pass1[i] >= 0) continue;
    for (int j = srp[i]; j < srp[i + 1]; ++j) {
      const int a = pass1[sci[j]];
      if (a >= 0 && !conflicts(i, members[a])) { agg[i] = a; members[a].push_back(i); break; }
    }
  }
  for (int i = 0; i < n; ++i) {
    if (agg[i] >= 0) continue;
    std::vector<int> mem{i};
    agg[i] = (int)members.size();
    for (int j = srp[i]; j < srp[i + 1]; ++j) {
      const int c = sci[j];
      if (agg[c] < 0 && !conflicts(c, mem)) { agg[c] = agg[i]; mem.push_back(c); }
    }
    members.push_back(std::move(mem));
  }
  return (int)members.size();
}

// C = A * B (Gustavson, columns of each row sorted)
struct GustavsonScratch {
  std::vector<int> mark, cols;
  std::vector<double> acc;
  explicit GustavsonScratch(int m) : mark(m, -1), acc(m, 0.0) {}
};

inline HostCsr spgemm(const HostCsr& A, const HostCsr& B) {
  return build_rows(A.n, B.m, [&] { return GustavsonScratch(B.m); },
                    [&](int i, GustavsonScratch& S, std::vector<int>& ci, std::vector<double>& v) {
    S.cols.clear();
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
      const int k = A.ci[j];
      const double a = A.v[j];
      for (int l = B.rp[k]; l < B.rp[k + 1]; ++l) {
        const int c = B.ci[l];
        if (S.mark[c] != i) { S.mark[c] = i; S.acc[c] = 0.0; S.cols.push_back(c); }
        S.acc[c] += a * B.v[l];
      }
    }
    std::sort(S.cols.begin(), S.cols.end());
    for (int c : S.cols) { ci.push_back(c); v.push_back(S.acc[c]); }
  });
}

inline HostCsr transpose(const HostCsr& A) {
  HostCsr T;
  T.n = A.m; T.m = A.n;
  T.rp.assign(A.m + 1, 0);
  for (int c : A.ci) ++T.rp[c + 1];
  for (int i = 0; i < A.m; ++i) T.rp[i + 1] += T.rp[i];
  T.ci.resize(A.ci.size()); T.v.resize(A.v.size());
  std::vector<int> pos(T.rp.begin(), T.rp.end() - 1);
  for (int i = 0; i < A.n; ++i)
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
      const int p = pos[A.ci[j]]++;
      T.ci[p] = i; T.v[p] = A.v[j];
    }
  return T;
}

// spectral radius of D^-1 A: 15 power iterations (x 1.1), capped by the Gershgorin bound
inline double estimate_rho(const HostCsr& A, const std::vector<double>& d) {
  double bound = 0.0;
  for (int i = 0; i < A.n; ++i) {
    double s = 0.0;
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) s += std::fabs(A.v[j]);
    bound = std::max(bound, s / std::fabs(d[i]));
  }
  if (!(bound > 0)) return 1.0;
  std::vector<double> v(A.n), w(A.n);
  uint64_t state = 0x9E3779B97F4A7C15ull;   // fixed seed: the hierarchy must be reproducible
  for (int i = 0; i < A.n; ++i) {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    v[i] = (double)(state >> 11) / 9007199254740992.0 - 0.5;
  }
  double lam = bound;
  for (int it = 0; it < 15; ++it) {
    double nrm = sum_rows(A.n, [&](int i) {
      double s = 0.0;
      for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) s += A.v[j] * v[A.ci[j]];
      w[i] = s / d[i];
      return w[i] * w[i];
    });
    nrm = std::sqrt(nrm);
    if (!(nrm > 0)) return bound;
    const double dot = sum_rows(A.n, [&](int i) { return w[i] * v[i]; }), vv = sum_rows(A.n, [&](int i) { return v[i] * v[i]; });
    lam = std::fabs(dot / vv);
    for (int i = 0; i < A.n; ++i) v[i] = w[i] / nrm;
  }
  return std::min(bound, 1.1 * lam);
}

// inverse of the bs x bs matrix m (row-major, destroyed); false if a pivot vanishes
template <class T>
bool small_inverse(T* m, T* inv, int bs) {
  for (int i = 0; i < bs; ++i) for (int j = 0; j < bs; ++j) inv[i * bs + j] = i == j ? 1.0 : 0.0;
  for (int k = 0; k < bs; ++k) {
    int piv = k;
    for (int i = k + 1; i < bs; ++i) if (fabs(m[i * bs + k]) > fabs(m[piv * bs + k])) piv = i;
    if (m[piv * bs + k] == 0.0) return false;
    if (piv != k)
      for (int j = 0; j < bs; ++j) {
        T t = m[k * bs + j]; m[k * bs + j] = m[piv * bs + j]; m[piv * bs + j] = t;
        t = inv[k * bs + j]; inv[k * bs + j] = inv[piv * bs + j]; inv[piv * bs + j] = t;
      }
    const T d = 1.0 / m[k * bs + k];
    for (int j = 0; j < bs; ++j) { m[k * bs + j] *= d; inv[k * bs + j] *= d; }
    for (int i = 0; i < bs; ++i) {
      if (i == k) continue;
      const T f = m[i * bs + k];
      for (int j = 0; j < bs; ++j) { m[i * bs + j] -= f * m[k * bs + j]; inv[i * bs + j] -= f * inv[k * bs + j]; }
    }
  }
  return true;
}

// spectral radius of B^-1 A, B = the bs x bs diagonal blocks of A (power iteration, fixed seed)
inline double estimate_rho_block(const HostCsr& A, int bs) {
  const int nb = A.n / bs;
  std::vector<double> binv((size_t)nb * bs * bs);
  for (int c = 0; c < nb; ++c) {
    double m[64] = {0};
    for (int a = 0; a < bs; ++a)
      for (int j = A.rp[c * bs + a]; j < A.rp[c * bs + a + 1]; ++j) {
        const int b = A.ci[j] - c * bs;
        if (b >= 0 && b < bs) m[a * bs + b] = A.v[j];
      }
    if (!small_inverse(m, &binv[(size_t)c * bs * bs], bs)) return -1.0;
  }
  std::vector<double> v(A.n), w(A.n), u(A.n);
  uint64_t state = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < A.n; ++i) {
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    v[i] = (double)(state >> 11) / 9007199254740992.0 - 0.5;
  }
  double lam = 1.0;
  for (int it = 0; it < 20; ++it) {
    for_chunks(A.n, 4 * host_threads(), [&](int, int r0, int r1) {
      for (int i = r0; i < r1; ++i) {
        double s = 0.0;
        for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) s += A.v[j] * v[A.ci[j]];
        u[i] = s;
      }
    });
    double nrm = sum_rows(nb, [&](int c) {
      double q = 0.0;
      for (int a = 0; a < bs; ++a) {
        double s = 0.0;
        for (int b = 0; b < bs; ++b) s += binv[((size_t)c * bs + a) * bs + b] * u[c * bs + b];
        w[c * bs + a] = s;
        q += s * s;
      }
      return q;
    });
    nrm = std::sqrt(nrm);
    if (!(nrm > 0)) break;
    const double dot = sum_rows(A.n, [&](int i) { return w[i] * v[i]; }), vv = sum_rows(A.n, [&](int i) { return v[i] * v[i]; });
    lam = std::fabs(dot / vv);
    for (int i = 0; i < A.n; ++i) v[i] = w[i] / nrm;
  }
  return 1.1 * lam;     // the iteration approaches rho from below
}

// P = (I - w D^-1 A) T for the piecewise-constant T of `agg`.  filter_theta > 0: the smoothing uses the FILTERED operator
// -- off-diagonal entries that are not large (|a_ij| >= filter_theta sqrt(a_ii a_jj)) are dropped and added to the
// diagonal, so the basis functions spread along the large couplings only and the coarse stencils stay narrow (on the
// stretched cells the small entries are the majority: 977 entries per row on the third level of the hexahedral DG
// hierarchy without it).  By magnitude, not by sign: the big positive entries of stretched Q1 cells must stay in the
// smoothing (lumped into the diagonal they inflate it and the smoothing is lost).
inline HostCsr smoothed_prolongator(const HostCsr& A, const std::vector<double>& d, const std::vector<int>& agg, int na, double w,
                             double filter_theta = 0.0) {
  return build_rows(A.n, na, [&] { return GustavsonScratch(na); },
                    [&](int i, GustavsonScratch& S, std::vector<int>& ci, std::vector<double>& v) {
    S.cols.clear();
    auto add = [&](int c, double x) {
      if (S.mark[c] != i) { S.mark[c] = i; S.acc[c] = 0.0; S.cols.push_back(c); }
      S.acc[c] += x;
    };
    add(agg[i], 1.0);
    if (w != 0.0 && filter_theta > 0.0) {
      double dF = d[i];
      for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
        const int c = A.ci[j];
        if (c != i && !(std::fabs(A.v[j]) >= filter_theta * std::sqrt(std::fabs(d[i] * d[c])))) dF += A.v[j];
      }
      if (!(dF >= 0.25 * d[i])) dF = d[i];        // (lumping must not empty the diagonal)
      add(agg[i], -w);                             // the diagonal of the filtered row: -w dF / dF
      for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
        const int c = A.ci[j];
        if (c != i && std::fabs(A.v[j]) >= filter_theta * std::sqrt(std::fabs(d[i] * d[c]))) add(agg[c], -w * A.v[j] / dF);
      }
    } else if (w != 0.0) {
      for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) add(agg[A.ci[j]], -w * A.v[j] / d[i]);
    }
    std::sort(S.cols.begin(), S.cols.end());
    for (int c : S.cols) { ci.push_back(c); v.push_back(S.acc[c]); }
  });
}

// explicit inverse of the dense coarsest operator (+ shift * 1 1^T / n when it carries the constant null space)
// Gauss-Jordan with partial pivoting: inv = M^-1 (M is destroyed)
inline bool invert_dense(std::vector<double>& M, std::vector<double>& inv, int n) {
  inv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(M[(size_t)i * n + k]) > std::fabs(M[(size_t)piv * n + k])) piv = i;
    if (M[(size_t)piv * n + k] == 0.0) return false;
    if (piv != k)
      for (int j = 0; j < n; ++j) {
        std::swap(M[(size_t)k * n + j], M[(size_t)piv * n + j]);
        std::swap(inv[(size_t)k * n + j], inv[(size_t)piv * n + j]);
      }
    const double ip = 1.0 / M[(size_t)k * n + k];
    for (int j = 0; j < n; ++j) { M[(size_t)k * n + j] *= ip; inv[(size_t)k * n + j] *= ip; }
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const double f = M[(size_t)i * n + k];
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) {
        M[(size_t)i * n + j] -= f * M[(size_t)k * n + j];
        inv[(size_t)i * n + j] -= f * inv[(size_t)k * n + j];
      }
    }
  }
  return true;
}

inline bool dense_inverse(const HostCsr& A, bool singular, std::vector<double>& inv) {
  const int n = A.n;
  if (!singular) {
    // The concentration system is K - 1 independent ion blocks and so is every coarse operator of its hierarchy: the
    // connected components are inverted one by one (half the work of the elimination for two blocks, and the reason a
    // 1 760-row coarsest level is affordable)
    std::vector<int> parent(n);
    for (int i = 0; i < n; ++i) parent[i] = i;
    auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
    for (int i = 0; i < n; ++i)
      for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
        const int a = find(i), b = find(A.ci[j]);
        if (a != b) parent[std::max(a, b)] = std::min(a, b);
      }
    int ncomp = 0;
    for (int i = 0; i < n; ++i) ncomp += find(i) == i;
    if (ncomp > 1 && ncomp <= 8) {
      inv.assign((size_t)n * n, 0.0);
      std::vector<int> loc(n, -1), idx;
      std::vector<double> M, Mi;
      for (int root = 0; root < n; ++root) {
        if (find(root) != root) continue;
        idx.clear();
        for (int i = 0; i < n; ++i) if (find(i) == root) { loc[i] = (int)idx.size(); idx.push_back(i); }
        const int m = (int)idx.size();
        M.assign((size_t)m * m, 0.0);
        for (int a = 0; a < m; ++a)
          for (int j = A.rp[idx[a]]; j < A.rp[idx[a] + 1]; ++j) M[(size_t)a * m + loc[A.ci[j]]] = A.v[j];
        if (!invert_dense(M, Mi, m)) return false;
        for (int a = 0; a < m; ++a)
          for (int b = 0; b < m; ++b) inv[(size_t)idx[a] * n + idx[b]] = Mi[(size_t)a * m + b];
      }
      return true;
    }
  }
  std::vector<double> M((size_t)n * n, 0.0);
  double tr = 0.0;
  for (int i = 0; i < n; ++i)
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
      M[(size_t)i * n + A.ci[j]] = A.v[j];
      if (A.ci[j] == i) tr += A.v[j];
    }
  if (singular) {
    const double s = tr / n / n;
    for (auto& x : M) x += s;
  }
  return invert_dense(M, inv, n);
}

// Connected components of the graph of A (numbered by their lowest row); returns their number.
inline int components(const HostCsr& A, std::vector<int>& comp) {
  const int n = A.n;
  std::vector<int> parent(n);
  for (int i = 0; i < n; ++i) parent[i] = i;
  auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
  for (int i = 0; i < n; ++i)
    for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) {
      const int a = find(i), b = find(A.ci[j]);
      if (a != b) parent[std::max(a, b)] = std::min(a, b);
    }
  comp.assign(n, -1);
  std::vector<int> id(n, -1);
  int nc = 0;
  for (int i = 0; i < n; ++i) { const int r = find(i); if (id[r] < 0) id[r] = nc++; comp[i] = id[r]; }
  return nc;
}

// Aggregates renumbered so that those of one connected component of A are consecutive (order inside a component kept):
// the K - 1 ion systems of every sub-domain of the concentration matrix are independent, and with this numbering every
// coarse operator of their hierarchy is block diagonal with CONTIGUOUS blocks -- the dense inverse of the coarsest level is
// then stored and applied block by block without any index list (dense_inverse_blocks).  2 .. 8 components only.
inline void renumber_by_component(const HostCsr& A, std::vector<int>& agg, int na) {
  std::vector<int> comp;
  const int nc = components(A, comp);
  if (nc < 2 || nc > 8) return;
  std::vector<int> agg_comp(na, nc);
  for (int i = 0; i < A.n; ++i) if (agg[i] >= 0 && agg[i] < na) agg_comp[agg[i]] = std::min(agg_comp[agg[i]], comp[i]);
  std::vector<int> order(na);
  for (int a = 0; a < na; ++a) order[a] = a;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return agg_comp[x] < agg_comp[y]; });
  std::vector<int> new_id(na);
  for (int k = 0; k < na; ++k) new_id[order[k]] = k;
  for (int i = 0; i < A.n; ++i) if (agg[i] >= 0 && agg[i] < na) agg[i] = new_id[agg[i]];
}

// The inverse of the coarsest operator as up to 8 dense diagonal blocks: block b covers the unknowns start[b] ..
// start[b] + size[b] - 1 (a connected component, or several), its size[b] x size[b] inverse sits at v[off[b]] row-major.
// One block of all unknowns when the components are not contiguous ranges, more than 8, or the operator is singular (the
// shift of the constant null space couples everything).  Two ion systems in two sub-domains at config 2: 1 760 unknowns in
// blocks of 730 + 730 + 150 + 150, 1.1 M instead of 3.1 M values streamed at every application.
struct DenseInvBlocks {
  int nb = 0;
  int start[8], size[8], off[8];
  std::vector<double> v;
};
inline bool dense_inverse_blocks(const HostCsr& A, bool singular, DenseInvBlocks& out) {
  const int n = A.n;
  std::vector<int> comp;
  int nc = singular ? 1 : components(A, comp);
  bool contiguous = nc >= 2 && nc <= 8;
  if (contiguous)
    for (int i = 1; i < n && contiguous; ++i) contiguous = comp[i] == comp[i - 1] || comp[i] == comp[i - 1] + 1;
  out.v.clear();
  if (!contiguous) {
    std::vector<double> inv;
    if (!dense_inverse(A, singular, inv)) return false;
    out.nb = 1; out.start[0] = 0; out.size[0] = n; out.off[0] = 0;
    out.v = std::move(inv);
    return true;
  }
  out.nb = nc;
  int s = 0;
  std::vector<double> M, Mi;
  for (int c = 0; c < nc; ++c) {
    int e = s;
    while (e < n && comp[e] == c) ++e;
    const int m = e - s;
    M.assign((size_t)m * m, 0.0);
    for (int i = s; i < e; ++i)
      for (int j = A.rp[i]; j < A.rp[i + 1]; ++j) M[(size_t)(i - s) * m + (A.ci[j] - s)] = A.v[j];
    if (!invert_dense(M, Mi, m)) return false;
    out.start[c] = s; out.size[c] = m; out.off[c] = (int)out.v.size();
    out.v.insert(out.v.end(), Mi.begin(), Mi.end());
    s = e;
  }
  return true;
}

}  // namespace kn_amg_host
