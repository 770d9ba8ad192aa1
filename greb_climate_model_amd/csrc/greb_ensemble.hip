// greb_ensemble.hip -- on-device ensemble statistics across members.
//
// The reference has no ensemble machinery: separate `ens_id` processes write separate files and the
// statistics are left to the R scripts (src/greb.f90:153,1064-1068; R/analyse_*.R).  Here the members of a
// GPU sit side by side in HBM (x[n_members][n], e.g. the monthly means [member][year][12][5][ny][nx] viewed as
// [member][n]), so their moments and quantiles are one streaming pass each:
//   moments   : per element the fp64 sum and sum of squares, min and max over the members -- the partials a
//               multi-GPU run all-reduces (ensemble.py); HBM-bound, 4 B per member-element read once
//   quantiles : per element the members are sorted in LDS (bitonic network, tile of points x all members
//               resident in one CU's 160 KB) and read out by linear interpolation of the order statistics
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/greb_engine.h"
#include "greb_kernels.h"

namespace {

constexpr int kMomThreads = 256;

// V consecutive elements per lane (2 or 4), U member rows in flight per lane
template <int V, int U>
__global__ __launch_bounds__(kMomThreads) void moments_kernel(const float* __restrict__ x, int nm, size_t n,
                                                              double* __restrict__ sum, double* __restrict__ sumsq,
                                                              float* __restrict__ mn, float* __restrict__ mx) {
  typedef float vec __attribute__((ext_vector_type(V)));
  const size_t g = (size_t)blockIdx.x * kMomThreads + threadIdx.x; // one group of V consecutive elements
  if (V * g >= n) return; // n % V == 0 (checked by the launcher): rows stay aligned, groups are whole
  double s[V], s2[V];
  float lo[V], hi[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { s[e] = 0; s2[e] = 0; lo[e] = INFINITY; hi[e] = -INFINITY; }
  const vec* p = reinterpret_cast<const vec*>(x) + g;
  const size_t stride = n / V;
  auto take = [&](const vec& v) {
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const double d = (double)v[e];
      s[e] += d; s2[e] = fma(d, d, s2[e]);
      lo[e] = fminf(lo[e], v[e]); hi[e] = fmaxf(hi[e], v[e]);
    }
  };
  int m = 0;
  for (; m + U <= nm; m += U) {
    vec v[U];
#pragma unroll
    for (int j = 0; j < U; ++j) v[j] = p[(size_t)(m + j) * stride];
#pragma unroll
    for (int j = 0; j < U; ++j) take(v[j]);
  }
  for (; m < nm; ++m) take(p[(size_t)m * stride]);
#pragma unroll
  for (int e = 0; e < V; ++e) {
    if (sum) sum[V * g + e] = s[e];
    if (sumsq) sumsq[V * g + e] = s2[e];
    if (mn) mn[V * g + e] = lo[e];
    if (mx) mx[V * g + e] = hi[e];
  }
}

// unaligned rows (n % 4 != 0): one element per thread
__global__ void moments_scalar_kernel(const float* __restrict__ x, int nm, size_t n, double* __restrict__ sum,
                                      double* __restrict__ sumsq, float* __restrict__ mn, float* __restrict__ mx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0, s2 = 0;
  float lo = INFINITY, hi = -INFINITY;
  for (int m = 0; m < nm; ++m) {
    const float v = x[(size_t)m * n + i];
    s += (double)v; s2 = fma((double)v, (double)v, s2);
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  if (sum) sum[i] = s;
  if (sumsq) sumsq[i] = s2;
  if (mn) mn[i] = lo;
  if (mx) mx[i] = hi;
}

// ---- quantiles -------------------------------------------------------------------------------
constexpr int kQThreads = 256;
constexpr int kQLdsFloats = 32768; // 128 KB tile: [mpad][P]
constexpr int kMaxProbs = 16;
struct Probs { int n; float p[kMaxProbs]; };

template <int P> // points per tile, a power of two <= 64
__global__ __launch_bounds__(kQThreads) void quantile_kernel(const float* __restrict__ x, int nm, int mpad, size_t n,
                                                             Probs pr, float* __restrict__ out) {
  extern __shared__ __align__(16) float tile[]; // [mpad][P]
  const size_t p0 = (size_t)blockIdx.x * P;
  const int tid = threadIdx.x;
  for (int i = tid; i < mpad * P; i += kQThreads) { // member-major: P consecutive elements per member row
    const int m = i / P, p = i % P;
    tile[i] = (m < nm && p0 + p < n) ? x[(size_t)m * n + p0 + p] : INFINITY; // padding sorts to the end
  }
  __syncthreads();
  // bitonic sort of every column (ascending along m)
  for (int k = 2; k <= mpad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (mpad / 2) * P; t += kQThreads) {
        const int p = t % P, h = t / P;                 // h-th compare-exchange of the stage
        const int i = ((h / j) * 2 * j) + (h % j);      // lower index of the pair, partner i + j
        const bool up = (i & k) == 0;
        const float a = tile[i * P + p], b = tile[(i + j) * P + p];
        const bool swap = up ? a > b : a < b;
        if (swap) { tile[i * P + p] = b; tile[(i + j) * P + p] = a; }
      }
      __syncthreads();
    }
  }
  for (int t = tid; t < pr.n * P; t += kQThreads) {
    const int p = t % P, qi = t / P;
    if (p0 + p >= n) continue;
    const float hpos = pr.p[qi] * (float)(nm - 1); // numpy's default ("linear") definition
    int lo = (int)floorf(hpos);
    lo = lo < 0 ? 0 : (lo > nm - 1 ? nm - 1 : lo);
    const int hi = lo + 1 < nm ? lo + 1 : lo;
    const float g = hpos - (float)lo;
    const float a = tile[lo * P + p], b = tile[hi * P + p];
    out[(size_t)qi * n + p0 + p] = a + g * (b - a);
  }
}

int fail_rc(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

} // namespace

extern "C" {

int greb_ensemble_moments_dev(const float* x_dev, int n_members, size_t n, double* sum_dev, double* sumsq_dev,
                              float* min_dev, float* max_dev, void* stream) {
  if (!x_dev || n_members < 1 || n < 1) return GREB_E_INVALID;
  hipStream_t s = (hipStream_t)stream;
  if (n % 4 == 0 && (reinterpret_cast<uintptr_t>(x_dev) & 15) == 0) {
    // 2 elements per lane x 16 member rows in flight: the most waves and loads in flight for the 270 K-element
    // fields of a GPU's ensemble (measured 4.4 TB/s of member data at 512 members; 4 x 8: 4.1)
    const size_t groups = n / 2;
    hipLaunchKernelGGL((moments_kernel<2, 16>), dim3((unsigned)((groups + kMomThreads - 1) / kMomThreads)),
                       dim3(kMomThreads), 0, s, x_dev, n_members, n, sum_dev, sumsq_dev, min_dev, max_dev);
  } else {
    hipLaunchKernelGGL(moments_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x_dev, n_members, n,
                       sum_dev, sumsq_dev, min_dev, max_dev);
  }
  return fail_rc(hipGetLastError());
}

int greb_ensemble_quantiles_dev(const float* x_dev, int n_members, size_t n, const float* probs, int n_probs,
                                float* out_dev, void* stream) {
  if (!x_dev || !probs || !out_dev || n_members < 1 || n < 1 || n_probs < 1 || n_probs > kMaxProbs) return GREB_E_INVALID;
  int mpad = 2;
  while (mpad < n_members) mpad <<= 1;
  if (mpad > 4096) return GREB_E_UNSUPPORTED; // a tile of 8 points x 4096 members is the LDS limit
  Probs pr; pr.n = n_probs;
  for (int i = 0; i < n_probs; ++i) {
    if (!(probs[i] >= 0.f && probs[i] <= 1.f)) return GREB_E_INVALID;
    pr.p[i] = probs[i];
  }
  int P = kQLdsFloats / mpad;
  if (P > 64) P = 64;
  const size_t lds = (size_t)mpad * P * sizeof(float);
  const unsigned grid = (unsigned)((n + P - 1) / P);
  hipStream_t s = (hipStream_t)stream;
#define GREB_QLAUNCH(PP)                                                                                              \
  {                                                                                                                   \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(quantile_kernel<PP>),                           \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, greb::kMaxDynamicLds);             \
    if (e != hipSuccess) return (int)e;                                                                               \
    hipLaunchKernelGGL(quantile_kernel<PP>, dim3(grid), dim3(kQThreads), lds, s, x_dev, n_members, mpad, n, pr, out_dev); \
  }
  switch (P) {
    case 64: GREB_QLAUNCH(64) break;
    case 32: GREB_QLAUNCH(32) break;
    case 16: GREB_QLAUNCH(16) break;
    default: GREB_QLAUNCH(8) break;
  }
#undef GREB_QLAUNCH
  return fail_rc(hipGetLastError());
}

} // extern "C"
