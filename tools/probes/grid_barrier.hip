// Probe: cost of a software grid barrier (one arrive counter + generation word, agent scope) between the workgroups of a
// persistent kernel, as a function of the number of workgroups.  usage: grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned nwg, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++gen;
    __threadfence();
    const unsigned arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == nwg * gen) __hip_atomic_store(&bar[1], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else {
      unsigned spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen && ++spins < (1u << 22))
        __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();
  }
  __syncthreads();
}

__global__ void k(unsigned* bar, double* data, int n, int rounds) {
  unsigned gen = 0;
  const int t = blockIdx.x * blockDim.x + threadIdx.x, T = gridDim.x * blockDim.x;
  for (int r = 0; r < rounds; ++r) {
    for (int i = t; i < n; i += T) data[i] = data[(i + 977) % n] * 0.5 + 1.0;   // a little dependent work per phase
    grid_sync(bar, gridDim.x, gen);
  }
}

int main() {
  unsigned* bar; double* data;
  const int n = 26417;
  (void)hipMalloc(&bar, 8); (void)hipMalloc(&data, n * 8);
  (void)hipMemset(data, 0, n * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wgs : {32, 64, 128, 256, 512}) {
    for (int threads : {256, 1024}) {
      for (int rounds : {100, 1100}) {
        (void)hipMemset(bar, 0, 8);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<<<wgs, threads>>>(bar, data, n, rounds);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        static float base;
        if (rounds == 100) base = ms;
        else printf("workgroups %4d x %4d threads: %.2f us per phase (work + barrier)\n", wgs, threads, (ms - base) * 1e3 / 1000.0);
      }
    }
  }
  return 0;
}
