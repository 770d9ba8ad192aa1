// Cycles per sweep of greb_chain6.h's loops on a lone wavefront (and with a second chain beside it on the same SIMD).
// hipcc --offload-arch=gfx950 -O3 -I greb_climate_model_amd/csrc tools/ubench/chain_rate.hip -o tools/ubench/chain_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "greb_chain6.h"
using namespace greb;

template <int MODE>
__global__ void k(float* out, int n, unsigned long long* cyc) {
  float T[6], K[6][6];
  for (int i = 0; i < 6; ++i) T[i] = 250.f + 0.01f * ((threadIdx.x * 6 + i) % 37);
  for (int i = 0; i < 6; ++i) {
    const float w = 0.9f + 0.001f * i, cs = 0.05f;
    K[i][0] = -cs * w; K[i][1] = -3.f * cs * w; K[i][2] = -6.f * cs * w;
    K[i][3] = 6.f * cs * w; K[i][4] = 3.f * cs * w; K[i][5] = cs * w;
  }
  const ChainK c = chain_pack(K);
  n = __builtin_amdgcn_readfirstlane(n);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float mn = 1.f;
  if (MODE == 0) { int rem = chain_sweeps6<false>(T, c, n); mn = (float)rem; }
  if (MODE == 1) mn = chain_sweeps6_all<false>(T, c, n);
  if (MODE == 2) chain_sweeps6_plain(T, c, n);
  if (MODE == 3) mn = chain_sweeps6_all<true>(T, c, n);
  if (MODE == 4) { const bool pos = chain_stays_positive(T, K, n); chain_run6<false>(T, c, n, pos); mn = pos ? 1.f : 0.f; if (threadIdx.x == 0) cyc[32] = pos; }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = mn;
  for (int i = 0; i < 6; ++i) s += T[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
  for (int threads : {64, 320}) { // 320: waves 0 and 4 share SIMD 0
    for (int n : {256, 1024}) {
      hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, n, cyc);
      (void)hipDeviceSynchronize();
      hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, n, cyc);
      (void)hipDeviceSynchronize();
      unsigned long long h[8];
      (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      printf("%-34s waves=%d n=%4d: %.1f cycles per sweep (wave 0)%s\n", name, threads / 64, n, (double)h[0] / n,
             threads > 64 ? " with 4 more waves, one of them on its SIMD" : "");
    }
  }
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 64 * 8);
  run<0>("stop-before loop (2 per trip)", out, cyc);
  run<1>("carried min (8 per trip)", out, cyc);
  run<2>("plain (8 per trip)", out, cyc);
  run<3>("carried, 16-lane circles (4 per trip)", out, cyc);
  run<4>("chain_run6 behind chain_stays_positive", out, cyc);
  unsigned long long pos; (void)hipMemcpy(&pos, cyc + 32, 8, hipMemcpyDeviceToHost);
  printf("chain_stays_positive said %llu\n", pos);
  return 0;
}
