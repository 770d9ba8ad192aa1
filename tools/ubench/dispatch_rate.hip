// How fast does the GPU start single-wavefront workgroups?  An (almost) empty kernel with the sub-step kernel's
// footprint (64 threads, 19.5 KB of LDS, ~190 VGPRs: two workgroups per SIMD), N workgroups, each spinning `spin` cycles.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/dispatch_rate.hip -o tools/ubench/dispatch_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void k(float* out, int spin) {
  extern __shared__ float lds[];
  float r[180];
#pragma unroll
  for (int i = 0; i < 180; ++i) r[i] = threadIdx.x * 0.5f + i;
  asm volatile("" : "+v"(r[0]), "+v"(r[179]));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < spin) {
#pragma unroll
    for (int i = 0; i < 179; ++i) r[i] = r[i] * 1.0001f + r[i + 1];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 180; ++i) s += r[i];
  lds[threadIdx.x] = s;
  if (s == 1234.5f) out[0] = lds[(threadIdx.x + 1) & 63];
}

int main() {
  float* out; (void)hipMalloc(&out, 64);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 20 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int spin : {0, 5000, 20000}) {
    for (int n : {256, 1024, 2048, 2400, 3072, 4096, 6144, 8192, 16384}) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(n), dim3(64), 19968, 0, out, spin);
      (void)hipEventRecord(e0);
      const int reps = 20;
      for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k, dim3(n), dim3(64), 19968, 0, out, spin);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps;
      printf("spin %5d cycles, %5d workgroups: %7.2f us per launch (%.1f rounds of 2048 slots; a round of spin = %.2f us at 2.4 GHz)\n",
             spin, n, us, n / 2048.0, spin / 2.4e3);
    }
  }
  return 0;
}
