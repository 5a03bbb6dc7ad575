// Probe: time per kernel of a chain of dependent small kernels, (a) launched one by one on a stream, (b) replayed from a
// hipGraph captured from the same stream.  hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void small(double* a, int n, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    double v = a[i];
    for (int k = 0; k < spin; ++k) v = v * 1.0000001 + 1e-9;
    a[i] = v;
  }
}
int main() {
  const int n = 1 << 16, chain = 6, reps = 2000;
  double* a;
  CK(hipMalloc(&a, n * sizeof(double)));
  CK(hipMemset(a, 0, n * sizeof(double)));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int spin : {0, 200}) {
    for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, st, a, n, spin);
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r)
      for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, st, a, n, spin);
    CK(hipStreamSynchronize(st));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("spin %3d  stream launches: %.2f us per kernel\n", spin, us / (reps * chain));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < chain; ++k) hipLaunchKernelGGL(small, dim3(n / 256), dim3(256), 0, st, a, n, spin);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 20; ++w) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("spin %3d  graph of %d:      %.2f us per kernel\n", spin, chain, us / (reps * chain));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
