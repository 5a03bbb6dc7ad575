// Ceiling probe for the DG assembly kernels: how fast can 128-thread workgroups stream 20 KiB row images to HBM
// (the kernels' store pattern), alone and next to a 64 B/row record read?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(128) void wr(double2* dst, int per_block, int chunk) {
  extern __shared__ double2 img[];
  const int b = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  for (int e = threadIdx.x; e < per_block; e += 128) img[e] = make_double2(e, b);
  __syncthreads();
  double2* d = dst + (size_t)b * per_block;
  for (int e = threadIdx.x; e < per_block; e += 128) d[e] = img[e];
}
__global__ __launch_bounds__(128) void rdwr(double2* dst, const double2* rec, int per_block, int chunk) {
  extern __shared__ double2 img[];
  const int b = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
  const double2* r = rec + ((size_t)b * 128 + threadIdx.x) * 4;
  const double2 a = r[0], c = r[1], e2 = r[2], f = r[3];
  for (int e = threadIdx.x; e < per_block; e += 128) img[e] = make_double2(a.x + c.y + e, e2.x + f.y);
  __syncthreads();
  double2* d = dst + (size_t)b * per_block;
  for (int e = threadIdx.x; e < per_block; e += 128) d[e] = img[e];
}
int main() {
  const int nblocks = 31104, per_block = 128 * 20 / 2;   // double2 per block: 20 KiB; 31104 blocks = r=2 mesh
  const int chunk = (nblocks + 7) / 8;
  double2 *dst, *rec;
  hipMalloc(&dst, (size_t)8 * chunk * per_block * 16);
  hipMalloc(&rec, (size_t)8 * chunk * 128 * 64);
  hipMemset(rec, 0, (size_t)8 * chunk * 128 * 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) {
        if (mode == 0) hipLaunchKernelGGL(wr, dim3(8 * chunk), dim3(128), per_block * 16, 0, dst, per_block, chunk);
        else hipLaunchKernelGGL(rdwr, dim3(8 * chunk), dim3(128), per_block * 16, 0, dst, rec, per_block, chunk);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)8 * chunk * per_block * 16 + (mode ? (double)8 * chunk * 128 * 64 : 0);
      printf("%s: %.1f us  %.0f GB/s\n", mode ? "read 64B/row + write" : "write only", ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
    }
  }
  return 0;
}
