// VALU issue-rate microbenchmark for gfx950: scalar v_fma_f32 vs packed v_pk_fma_f32 vs v_pk_add/mul,
// 1..4 waves per SIMD.  Prints cycles per wave-instruction per SIMD (lower = faster).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  float b0 = 8.f, b1 = 9.f, b2 = 10.f, b3 = 11.f, b4 = 12.f, b5 = 13.f, b6 = 14.f, b7 = 15.f;
  const float c = 1.0001f, d = 0.5f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p0 = {a0, b0}, p1 = {a1, b1}, p2 = {a2, b2}, p3 = {a3, b3}, p4 = {a4, b4}, p5 = {a5, b5}, p6 = {a6, b6}, p7 = {a7, b7};
  v2 cc = {c, c}, dd = {d, d};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { // 8 independent scalar FMAs x 16
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));)
    } else if (MODE == 1) { // 8 independent packed FMAs x 16
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(cc), "v"(dd));)
    } else if (MODE == 2) { // packed add
      REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(dd));)
    } else if (MODE == 3) { // scalar add
      REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));)
    } else if (MODE == 4) { // dependent chain scalar fma
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(c), "v"(d));)
    } else if (MODE == 5) { // dependent chain packed fma
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(p0) : "v"(cc), "v"(dd));)
    } else if (MODE == 6) { // v_max_f32
      REP16(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                         "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));)
    } else if (MODE == 7) { // v_mov_b32
      REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                         "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));)
    }
  }
  long long t1 = clock64();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.y;
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out) {
  const int iters = 2000;
  for (int waves_per_simd : {1, 2, 3, 4}) {
    const int threads = 64 * 4 * waves_per_simd; // one block per CU, waves spread over 4 SIMDs
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, (char*)out + 8, 8, hipMemcpyDeviceToHost);
    const double ninstr = (double)iters * 16 * 8;                 // per wave
    const double per_simd = ninstr * waves_per_simd;              // wave-instructions through one SIMD
    printf("%-22s waves/SIMD=%d  %.2f ms  s_memtime cycles/instr/wave=%.2f  => cycles per wave-instr per SIMD=%.2f (@%.0f MHz eff)\n",
           name, waves_per_simd, ms, (double)cyc / ninstr, (double)cyc / per_simd, cyc / (ms * 1e3));
  }
}

int main() {
  float* out; hipMalloc(&out, 64);
  run<0>("v_fma_f32 indep", out);
  run<1>("v_pk_fma_f32 indep", out);
  run<3>("v_add_f32 indep", out);
  run<2>("v_pk_add_f32 indep", out);
  run<4>("v_fma_f32 dependent", out);
  run<5>("v_pk_fma_f32 dependent", out);
  run<6>("v_max_f32 indep", out);
  run<7>("v_mov_b32 indep", out);
  return 0;
}
