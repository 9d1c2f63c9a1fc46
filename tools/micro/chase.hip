// Dependent-load latency on MI355X as a function of the footprint: one lane chases a random cycle
// through a buffer of `bytes` (one 64 B line per hop).  Tells how much of the search kernel's
// "memory round trip" is HBM and how much is address translation.   hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
__global__ void init_kernel(uint64_t *buf, size_t n_lines, uint64_t mult) {
  // next(i) = (i * mult + 1) mod n_lines with n_lines a power of two and mult = 5 (mod 8): a full-period LCG
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += (size_t)gridDim.x * blockDim.x)
    buf[i * 8] = (i * mult + 1) & (n_lines - 1);
}
__global__ void chase_kernel(const uint64_t *buf, int hops, uint64_t *out, unsigned long long *cycles) {
  uint64_t i = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int h = 0; h < hops; h++) i = __builtin_nontemporal_load(&buf[i * 8]);
  const unsigned long long t1 = __builtin_readcyclecounter();
  *out = i;
  *cycles = t1 - t0;
}
// one workgroup, every thread fetches independent random lines: the scattered-line throughput of ONE CU
__global__ __launch_bounds__(512) void gather_kernel(const uint64_t *buf, size_t n_lines, int iters, uint64_t *out, unsigned long long *cycles) {
  uint64_t acc = 0, x = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    x = x * 6364136223846793005ull + 1442695040888963407ull;
    acc += __builtin_nontemporal_load(&buf[((x >> 20) & (n_lines - 1)) * 8]);
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}
int main() {
  const size_t sizes_mb[] = {16, 256, 2048, 16384, 65536};
  for (size_t mb : sizes_mb) {
    const size_t bytes = mb << 20, n_lines = bytes / 64;
    uint64_t *buf = nullptr, *out = nullptr;
    unsigned long long *cyc = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("%zu MB: hipMalloc failed\n", mb); continue; }
    hipMalloc(&out, 8); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(init_kernel, dim3(8192), dim3(256), 0, 0, buf, n_lines, 0x9E3779B97F4A7C15ull | 5ull);
    hipDeviceSynchronize();
    const int hops = 20000;
    unsigned long long c = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(chase_kernel, dim3(1), dim3(1), 0, 0, buf, hops, out, cyc);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    }
    printf("%6zu MB footprint: %.0f ns per dependent 64 B load (HIP events), %.0f s_memtime ticks -> %.2f ticks/ns\n", mb, 1e6 * ms / hops, (double)c / hops, (double)c / (1e6 * ms));
    {
      uint64_t *out2 = nullptr;
      hipMalloc(&out2, 8 * 512);
      const int iters = 2000;
      hipLaunchKernelGGL(gather_kernel, dim3(1), dim3(512), 0, 0, buf, n_lines, iters, out2, cyc);
      hipDeviceSynchronize();
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double lines = 512.0 * iters;
      printf("          one workgroup of 512 threads, independent random lines: %.2f cycles per line -> %.1f GB/s per CU at 2.39 GHz\n", (double)c / lines, 64.0 * lines / ((double)c / 2.39));
      hipFree(out2);
    }
    hipFree(buf); hipFree(out); hipFree(cyc);
  }
  return 0;
}
