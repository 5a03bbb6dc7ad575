// Host-side AMG set-up pieces (csrc/amg_host.h) on a 27-point operator of a cubic grid: wall time with 1 and with the
// default number of threads, and a check that the results are the same bits either way.
//   g++ -O2 -std=c++17 -pthread -I knp-emi-fenics-x_amd/csrc tools/probes/amg_host_bench.cpp -o /tmp/amg_host_bench && /tmp/amg_host_bench 72
// (KNPEMI_AMG_THREADS is read once per process: the program re-runs itself with KNPEMI_AMG_THREADS=1 for the comparison.)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "amg_host.h"
using namespace kn_amg_host;

static HostCsr grid27(int m) {
  HostCsr A;
  A.n = A.m = m * m * m;
  A.rp.assign(A.n + 1, 0);
  for (int z = 0; z < m; ++z) for (int y = 0; y < m; ++y) for (int x = 0; x < m; ++x) {
    const int i = (z * m + y) * m + x;
    double diag = 0.0;
    const size_t at = A.ci.size();
    int dpos = -1;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      const int X = x + dx, Y = y + dy, Z = z + dz;
      if (X < 0 || Y < 0 || Z < 0 || X >= m || Y >= m || Z >= m) continue;
      const int j = (Z * m + Y) * m + X;
      if (j == i) { dpos = (int)A.ci.size(); A.ci.push_back(j); A.v.push_back(0.0); continue; }
      const double w = -1.0 / (1 + abs(dx) + abs(dy) + 4 * abs(dz));     // anisotropic: weaker along z
      A.ci.push_back(j); A.v.push_back(w);
      diag -= w;
    }
    (void)at;
    A.v[dpos] = diag + 1e-3;
    A.rp[i + 1] = (int)A.ci.size();
  }
  return A;
}

static uint64_t digest(const HostCsr& A) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
  mix(A.rp.data(), A.rp.size() * sizeof(int)); mix(A.ci.data(), A.ci.size() * sizeof(int)); mix(A.v.data(), A.v.size() * sizeof(double));
  return h;
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 48;
  const bool child = argc > 2;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
  HostCsr A = grid27(m);
  const auto d = diagonal(A);
  std::vector<int> agg;
  auto t0 = now();
  const int na = aggregate(A, d, 0.08, false, agg);
  auto t1 = now();
  const double rho = estimate_rho(A, d);
  auto t2 = now();
  HostCsr P = smoothed_prolongator(A, d, agg, na, 4.0 / (3.0 * rho));
  auto t3 = now();
  HostCsr R = transpose(P);
  auto t4 = now();
  HostCsr AP = spgemm(A, P);
  auto t5 = now();
  HostCsr C = spgemm(R, AP);
  auto t6 = now();
  printf("%s threads %2d | n %d nnz %zu -> %d aggregates | aggregate %.3f  rho %.3f  prolongator %.3f  transpose %.3f  A P %.3f  R (A P) %.3f s | "
         "rho %.17g  digests P %016llx  C %016llx\n", child ? "  " : "", host_threads(), A.n, A.ci.size(), na, secs(t0, t1), secs(t1, t2), secs(t2, t3),
         secs(t3, t4), secs(t4, t5), secs(t5, t6), rho, (unsigned long long)digest(P), (unsigned long long)digest(C));
  if (!child) {
    setenv("KNPEMI_AMG_THREADS", "1", 1);
    std::string cmd = std::string(argv[0]) + " " + std::to_string(m) + " child";
    return system(cmd.c_str());
  }
  return 0;
}
