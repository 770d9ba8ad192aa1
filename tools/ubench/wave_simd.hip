// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_REG_HW_ID: wave_id[3:0], simd_id[5:4], ...)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = hw;
}
int main() {
  unsigned* d; hipMalloc(&d, 1024 * 16 * 4);
  for (int threads : {512, 640, 1024}) {
    hipMemset(d, 0, 1024 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(512), dim3(threads), 150 * 1024, 0, d);
    hipDeviceSynchronize();
    unsigned h[1024 * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b : {0, 1, 7, 300}) {
      printf("threads=%d block %d: simd per wave:", threads, b);
      for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
      printf("   cu:");
      for (int w = 0; w < threads / 64; ++w) printf(" %u", (h[b * 16 + w] >> 8) & 15);
      printf("\n");
    }
  }
  return 0;
}
