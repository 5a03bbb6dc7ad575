// Probe: does a wave64 fp64 VALU instruction get cheaper when only part of the wave is active?  One wave runs N dependent
// (and, second kernel, 4-way independent) fp64 FMAs with `active` live lanes.  usage: exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int ILP>
__global__ void chain(double* out, int n, int active) {
  if ((int)threadIdx.x >= active) return;
  double a[ILP];
  for (int k = 0; k < ILP; ++k) a[k] = 1.0 + threadIdx.x * 1e-3 + k;
  const double b = 1.0000001, c = 1e-9;
  for (int i = 0; i < n; ++i)
#pragma unroll
    for (int k = 0; k < ILP; ++k) a[k] = fma(a[k], b, c);
  double s = 0;
  for (int k = 0; k < ILP; ++k) s += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP>
void run(const char* what, int blocks) {
  double* d;
  hipMalloc(&d, sizeof(double) * 64 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 200000;
  for (int active : {64, 32, 16, 8, 4}) {
    chain<ILP><<<blocks, 64>>>(d, 1000, active);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chain<ILP><<<blocks, 64>>>(d, n, active);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%s blocks %d active %2d: %.3f ms, %.2f ns per FMA instruction per wave\n", what, blocks, active, ms, ms * 1e6 / ((double)n * ILP));
  }
  hipFree(d);
}

int main() {
  run<1>("dependent chain", 1);
  run<4>("4 independent chains", 1);
  run<8>("8 independent chains", 1);
  run<8>("8 independent chains", 1024);   // one wave per SIMD on the whole chip
  return 0;
}
