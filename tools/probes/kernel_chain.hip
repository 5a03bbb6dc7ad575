// Probe 2 of the per-kernel cost of the solver loops: the timeline of a whole step (tools/timeline.py on a rocprofv3 kernel
// trace, gpurun_out/r04/timeline_c2.txt) shows ~5 us for EVERY trivial kernel (extrapolate 5.3, gather 5.0, vec 5.2, start
// 4.8 us) although a chain of one repeated trivial kernel costs 2.4-2.7 us per link (kernel_head.hip).  What differs in the
// real chain: (a) every link is a DIFFERENT kernel (instruction cache), (b) the links read what the previous link wrote
// on other XCDs, (c) some kernels of the stream use scratch memory (the ODE sweep spills), (d) kernel argument blocks of
// up to 160 bytes, (e) kernel sizes.  Reported: us per kernel of a chain of 12, replayed from a hipGraph and launched on a
// stream.
//   hipcc --offload-arch=gfx950 -O3 kernel_chain.hip -o kernel_chain && ./kernel_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// ID makes each instantiation a different kernel; BODY pads the code with ID-dependent arithmetic (BODY fused multiply-adds)
template <int ID, int BODY>
__global__ __launch_bounds__(256) void axpy(const double* __restrict__ x, double* __restrict__ y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
#pragma unroll
  for (int k = 0; k < BODY; ++k) v = v * (1.0 + 1e-9 * (ID + k)) + 1e-12 * (k + 1);
  y[i] = v + ID;
}

// a kernel that needs scratch memory (a dynamically indexed private array)
__global__ __launch_bounds__(256) void scratchy(const double* __restrict__ x, double* __restrict__ y, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a[64];
  for (int k = 0; k < 64; ++k) a[k] = x[i] + k;
  int j = (int)(x[i] * 1e-300) & 63;
  for (int k = 0; k < 8; ++k) { a[(j + k) & 63] += 1.0; j = (j * 5 + 1) & 63; }
  y[i] = a[j];
}

using Launch = std::function<void(int, hipStream_t)>;

static double time_chain(hipStream_t st, const Launch& launch, int chain, int reps, bool graph) {
  for (int w = 0; w < 3 * chain; ++w) launch(w % chain, st);
  (void)hipStreamSynchronize(st);
  hipGraphExec_t ge = nullptr;
  if (graph) {
    hipGraph_t g;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int k = 0; k < chain; ++k) launch(k, st);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    for (int w = 0; w < 10; ++w) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
  }
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) {
    if (graph) (void)hipGraphLaunch(ge, st);
    else for (int k = 0; k < chain; ++k) launch(k, st);
  }
  (void)hipStreamSynchronize(st);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  if (ge) (void)hipGraphExecDestroy(ge);
  return us / ((double)reps * chain);
}

template <int BODY>
static void launch_distinct(int k, hipStream_t st, double* a, double* b, int n) {
  const dim3 g((n + 255) / 256), t(256);
  double* x = (k & 1) ? b : a;
  double* y = (k & 1) ? a : b;
  switch (k % 12) {
    case 0: hipLaunchKernelGGL((axpy<0, BODY>), g, t, 0, st, x, y, n); break;
    case 1: hipLaunchKernelGGL((axpy<1, BODY>), g, t, 0, st, x, y, n); break;
    case 2: hipLaunchKernelGGL((axpy<2, BODY>), g, t, 0, st, x, y, n); break;
    case 3: hipLaunchKernelGGL((axpy<3, BODY>), g, t, 0, st, x, y, n); break;
    case 4: hipLaunchKernelGGL((axpy<4, BODY>), g, t, 0, st, x, y, n); break;
    case 5: hipLaunchKernelGGL((axpy<5, BODY>), g, t, 0, st, x, y, n); break;
    case 6: hipLaunchKernelGGL((axpy<6, BODY>), g, t, 0, st, x, y, n); break;
    case 7: hipLaunchKernelGGL((axpy<7, BODY>), g, t, 0, st, x, y, n); break;
    case 8: hipLaunchKernelGGL((axpy<8, BODY>), g, t, 0, st, x, y, n); break;
    case 9: hipLaunchKernelGGL((axpy<9, BODY>), g, t, 0, st, x, y, n); break;
    case 10: hipLaunchKernelGGL((axpy<10, BODY>), g, t, 0, st, x, y, n); break;
    default: hipLaunchKernelGGL((axpy<11, BODY>), g, t, 0, st, x, y, n); break;
  }
}

int main() {
  const int chain = 12, reps = 400;
  const int nmax = 1 << 21;
  double *a, *b;
  CK(hipMalloc(&a, nmax * sizeof(double)));
  CK(hipMalloc(&b, nmax * sizeof(double)));
  CK(hipMemset(a, 0, nmax * sizeof(double)));
  CK(hipMemset(b, 0, nmax * sizeof(double)));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  printf("us per kernel, chain of %d dependent kernels (y = f(x), x and y alternate)\n", chain);
  printf("%-86s %8s %8s\n", "variant", "stream", "graph");
  for (int n : {26417, 422656, 2000000}) {
    printf("-- n = %d doubles (%d blocks)\n", n, (n + 255) / 256);
    auto row = [&](const char* label, const Launch& l) {
      printf("%-86s %8.2f %8.2f\n", label, time_chain(st, l, chain, reps, false), time_chain(st, l, chain, reps, true));
    };
    row("A the same small kernel 12 times", [&](int k, hipStream_t s) {
      hipLaunchKernelGGL((axpy<0, 4>), dim3((n + 255) / 256), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, n); });
    row("B twelve different small kernels", [&](int k, hipStream_t s) { launch_distinct<4>(k, s, a, b, n); });
    row("C twelve different kernels with ~2 KB of code each (200 fused multiply-adds)", [&](int k, hipStream_t s) { launch_distinct<200>(k, s, a, b, n); });
    row("D twelve different kernels with ~10 KB of code each (1000 fused multiply-adds)", [&](int k, hipStream_t s) { launch_distinct<1000>(k, s, a, b, n); });
    row("E as B, every 12th link a kernel that uses scratch memory", [&](int k, hipStream_t s) {
      if (k % 12 == 5) hipLaunchKernelGGL(scratchy, dim3((n + 255) / 256), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, n);
      else launch_distinct<4>(k, s, a, b, n); });
  }
  return 0;
}
