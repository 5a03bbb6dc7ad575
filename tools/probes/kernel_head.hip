// Probe: where the 4-5 us of a "trivial" solver kernel go (round-3 review: bi_check_kernel 4.97, start_kernel 5.48,
// dense_kernel<false> 5.09, vec_kernel 5.26 us under rocprofv3 against the guide's 1.45-1.9 us dependent boundary).
// A chain of dependent kernels on one stream in which kernel k leaves one partial sum per block and kernel k + 1 starts
// with the head of the fused Krylov kernels (kernels_fused.hip): early-out flag -> re-summation of the producer's partial
// sums -> scalars -> branch.  Variants isolate each ingredient.  Reported: host wall time per kernel of the chain (kernel
// duration + boundary), chain of 12, 400 repetitions.
//   hipcc --offload-arch=gfx950 -O3 kernel_head.hip -o kernel_head && ./kernel_head
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int FT = 256;

__device__ __forceinline__ double block_sum(double v) {
  __shared__ double sh[FT / 64];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return v;
}

struct Big { double pad[60]; };   // 480 bytes of by-value kernel arguments (DownArgs is ~160)

// HEAD 0: nothing.  1: flag load + branch.  2: + serial re-summation (as kernels_fused.hip: totals).  3: + re-summation with
// all loads issued before the first wait (fixed 8 per thread, predicated).  4: as 3, and the flag / scalar loads issued
// together with them (before the branch).  5: as 2 plus two more scalar loads AFTER the barrier (rho, omega: as now).
template <int HEAD, bool BIG>
__global__ __launch_bounds__(FT) void link(const double* __restrict__ part_in, int np, double* __restrict__ part_out,
                                          double* __restrict__ sc, double* __restrict__ data, int n, Big big) {
  double total = 0.0, extra = 0.0;
  if constexpr (HEAD == 1) { if (sc[0] != 0.0) return; }
  if constexpr (HEAD == 2 || HEAD == 5) {
    if (sc[0] != 0.0) return;
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += FT) s += part_in[i];
    total = block_sum(s);
    if constexpr (HEAD == 5) extra = sc[1] + sc[2];
  }
  if constexpr (HEAD == 3) {
    if (sc[0] != 0.0) return;
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = threadIdx.x + j * FT; v[j] = i < np ? part_in[i] : 0.0; }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    total = block_sum(s);
  }
  if constexpr (HEAD == 4) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int i = threadIdx.x + j * FT; v[j] = i < np ? part_in[i] : 0.0; }
    const double done = sc[0], s1 = sc[1], s2 = sc[2];
    if (done != 0.0) return;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v[j];
    total = block_sum(s);
    extra = s1 + s2;
  }
  if constexpr (BIG) extra += big.pad[threadIdx.x % 60];
  // the body: a little streaming work and one partial sum per block
  double acc = 0.0;
  for (int i = blockIdx.x * FT + threadIdx.x; i < n; i += gridDim.x * FT) {
    const double x = data[i] * 0.999 + 1e-12 * (total + extra);
    data[i] = x;
    acc += x;
  }
  acc = block_sum(acc);
  if (threadIdx.x == 0) part_out[blockIdx.x] = acc;
}

template <int HEAD, bool BIG>
static double run(hipStream_t st, double* part, double* sc, double* data, int n, int nb, int chain, int reps, bool graph) {
  Big big{};
  auto launch = [&](int k) {
    hipLaunchKernelGGL((link<HEAD, BIG>), dim3(nb), dim3(FT), 0, st, part + (k & 1) * 2048, nb, part + ((k + 1) & 1) * 2048, sc, data, n, big);
  };
  for (int w = 0; w < 50; ++w) launch(w);
  (void)hipStreamSynchronize(st);
  hipGraphExec_t ge = nullptr;
  if (graph) {
    hipGraph_t g;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < chain; ++k) launch(k);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    for (int w = 0; w < 10; ++w) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
  }
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) {
    if (graph) (void)hipGraphLaunch(ge, st);
    else for (int k = 0; k < chain; ++k) launch(k);
  }
  (void)hipStreamSynchronize(st);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  if (ge) (void)hipGraphExecDestroy(ge);
  return us / ((double)reps * chain);
}

int main(int argc, char** argv) {
  const int chain = 12, reps = 400;
  double *part, *sc, *data;
  const int nmax = 1 << 20;
  CK(hipMalloc(&part, 2 * 2048 * sizeof(double)));
  CK(hipMalloc(&sc, 64 * sizeof(double)));
  CK(hipMalloc(&data, nmax * sizeof(double)));
  CK(hipMemset(part, 0, 2 * 2048 * sizeof(double)));
  CK(hipMemset(sc, 0, 64 * sizeof(double)));
  CK(hipMemset(data, 0, nmax * sizeof(double)));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  printf("us per kernel of a chain of %d dependent kernels (host wall; duration + boundary)\n", chain);
  printf("%-78s %8s %8s\n", "variant", "stream", "graph");
  for (int nb : {1, 256, 1651, 2048}) {
    const int n = nb * FT;   // one element per thread: the body is as short as a body gets
    printf("-- %d blocks (%d partial sums for the next kernel)\n", nb, nb);
#define ROW(H, B, label) printf("%-78s %8.2f %8.2f\n", label, run<H, B>(st, part, sc, data, n, nb, chain, reps, false), \
                                run<H, B>(st, part, sc, data, n, nb, chain, reps, true))
    ROW(0, false, "0 no head");
    ROW(0, true, "0 no head, 480 B of by-value kernel arguments read");
    ROW(1, false, "1 flag load + branch");
    ROW(2, false, "2 flag, then serial re-summation of the partial sums (kernels_fused.hip: totals)");
    ROW(5, false, "5 as 2 + two scalar loads after the barrier (rho, omega: as the loops do now)");
    ROW(3, false, "3 flag, then re-summation with all 8 loads per thread in flight");
    ROW(4, false, "4 flag, scalars and the 8 loads issued together, then the branch");
  }
  return 0;
}
