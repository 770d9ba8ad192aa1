// greb_kernels.hip -- HIP kernels of the GREB hot path for MI355X (gfx950, wave64).
//
//   member_kernel       the fused engine: one 1024-thread workgroup per ensemble member keeps the
//                       member's two transported tracers (Tair, q) double-buffered in LDS together
//                       with both weight fields and the step's wind slice, and runs whole model
//                       steps -- point physics, 2 x 24 circulation sub-steps, Euler update, sea ice,
//                       monthly/annual accumulation -- without leaving the CU
//                       (src/greb.f90:239-364, 528-553).  96x48 grid.
//   diffusion_kernel    standalone batched diffusion sweep (src/greb.f90:556-723): the kernel the
//   advection_kernel    HBM-roofline metric is defined on (12 B/point), plus its advection twin
//                       (src/greb.f90:726-915); any grid, latitude-banded through LDS.
//   circulation_kernel  24 fused sub-steps for a batch of single fields (test mirror of a3).
//   point_kernel        a4-a8 for one step (test mirror).
//
// No MFMA anywhere: there is no dense contraction in this model.  The work is LDS-staged
// stencils + VALU; longitudes are processed in 16-byte quads so every LDS/HBM access is a
// coalesced dwordx4.
#include "greb_kernels.h"

namespace greb {

__device__ __forceinline__ void wave_lds_sync() {
  // order this wave's LDS writes before its later LDS reads (one wave owns a chain row)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A latitude band [r0, r1) of a field staged in LDS, row-major, nx floats per row.
struct Rows {
  const float* base;
  int r0, nx;
  __device__ __forceinline__ const float* row(int k) const { return base + (k - r0) * nx; }
};

struct QuadIn {
  float T[12], w[12];
  f4 T0, w0, Tm2, Tm1, Tp1, Tp2, wm2, wm1, wp1, wp2;
};

__device__ __forceinline__ f4 zero4() { return f4{{0.f, 0.f, 0.f, 0.f}}; }

// gather the neighbourhood of quad (k,q); rows outside [0,ny) get w = 0 and a clamped T row
__device__ __forceinline__ void gather(const Rows& X, const Rows& W, int k, int q, int nq, int ny, bool lat2,
                                       QuadIn& in) {
  load_window(X.row(k), q, nq, in.T);
  load_window(W.row(k), q, nq, in.w);
#pragma unroll
  for (int i = 0; i < 4; ++i) { in.T0.v[i] = in.T[4 + i]; in.w0.v[i] = in.w[4 + i]; }
  const int km1 = k >= 1 ? k - 1 : k, kp1 = k <= ny - 2 ? k + 1 : k;
  in.Tm1 = ld4(X.row(km1) + 4 * q); in.Tp1 = ld4(X.row(kp1) + 4 * q);
  in.wm1 = k >= 1 ? ld4(W.row(km1) + 4 * q) : zero4();
  in.wp1 = k <= ny - 2 ? ld4(W.row(kp1) + 4 * q) : zero4();
  if (lat2) {
    const int km2 = k >= 2 ? k - 2 : k, kp2 = k <= ny - 3 ? k + 2 : k;
    in.Tm2 = ld4(X.row(km2) + 4 * q); in.Tp2 = ld4(X.row(kp2) + 4 * q);
    in.wm2 = k >= 2 ? ld4(W.row(km2) + 4 * q) : zero4();
    in.wp2 = k <= ny - 3 ? ld4(W.row(kp2) + 4 * q) : zero4();
  } else {
    in.Tm2 = in.T0; in.Tp2 = in.T0; in.wm2 = zero4(); in.wp2 = zero4();
  }
}

// ---- one-sweep increments for a non-chain row (full-row branch, or sub-cycled with time2 == 1)
template <bool STRICT>
__device__ __forceinline__ void dif_quad(const QuadIn& in, const RowTables& tab, int k, int ny, float out[4]) {
  float dTx[4], dTy[4];
  if (STRICT) {
#pragma clang fp contract(off)
    if (!tab.subcycled[k]) {
      dif_lon_strict(in.T, in.w, tab.dif_ccx[k], dTx);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      dif_lon_strict(in.T, in.w, tab.dif_ccx2[k], dTx);
      clamp_add(T1h, dTx);
#pragma unroll
      for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - in.T[4 + i]; // :718
    }
    dif_lat_strict(in.T0, in.Tm1, in.Tp1, in.wm1, in.wp1, tab.dif_ccy, k, ny, dTy);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = in.w0.v[i] * (dTx[i] + dTy[i]); // :721
  } else {
    Flux f;
    make_flux(in.T, in.w, f);
    if (!tab.subcycled[k]) {
      dif_lon_fast(f, tab.dif_ccx[k] * 0.05f, dTx);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      dif_lon_fast(f, tab.dif_ccx2[k] * 0.05f, dTx);
      clamp_add_fast(T1h, dTx);
#pragma unroll
      for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - in.T[4 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g = in.wm1.v[i] * (in.Tm1.v[i] - in.T0.v[i]) + in.wp1.v[i] * (in.Tp1.v[i] - in.T0.v[i]);
      out[i] = in.w0.v[i] * (dTx[i] + tab.dif_ccy * g);
    }
  }
}

__device__ __forceinline__ void adv_lat_coef(const RowTables& tab, int k, int ny, float& am, float& ap) {
  const float third = tab.adv_ccy * (1.f / 3.f);
  am = (k == 1) ? tab.adv_ccy : third;      // :766-769
  ap = (k == ny - 2) ? tab.adv_ccy : third; // :784-787
  if (k == 0) am = 0.f;
  if (k == ny - 1) ap = 0.f;
}

template <bool STRICT>
__device__ __forceinline__ void adv_quad(const QuadIn& in, const float u[4], const float v[4],
                                         const RowTables& tab, int k, int ny, bool last_quad, float out[4]) {
  float dTx[4], dTy[4];
  if (STRICT) {
#pragma clang fp contract(off)
    if (!tab.subcycled[k]) {
      adv_lon_full_strict(in.T, in.w, u, tab.adv_ccx[k], dTx);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      adv_lon_sub_strict(in.T, in.w, u, tab.adv_ccx2[k], last_quad, dTx);
      clamp_add(T1h, dTx);
#pragma unroll
      for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - in.T[4 + i]; // :910
    }
    adv_lat_strict(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, v, tab.adv_ccy, k, ny, dTy);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = dTx[i] + dTy[i]; // :913
  } else {
    Flux f;
    make_flux(in.T, in.w, f);
    if (!tab.subcycled[k]) {
      adv_lon_full_fast(f, in.T, in.w, u, tab.adv_ccx[k] * (1.f / 3.f), dTx);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      adv_lon_sub_fast(f, in.T, in.w, u, tab.adv_ccx2[k] * 0.05f, last_quad, dTx);
      clamp_add_fast(T1h, dTx);
#pragma unroll
      for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - in.T[4 + i];
    }
    float am, ap, dd[4];
    adv_lat_coef(tab, k, ny, am, ap);
    lat_fast(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, v, 0.f, am, ap, dd, dTy);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = dTx[i] + dTy[i];
  }
}

// fused sub-step for a non-chain quad: X_new = (X + dX_diffuse) + dX_advec (:549)
template <bool STRICT>
__device__ __forceinline__ f4 substep_quad(const QuadIn& in, const float u[4], const float v[4],
                                           const RowTables& tab, int k, int ny, bool last_quad) {
  f4 r;
  if (STRICT) {
    float dd[4], da[4];
    dif_quad<true>(in, tab, k, ny, dd);
    adv_quad<true>(in, u, v, tab, k, ny, last_quad, da);
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int i = 0; i < 4; ++i) r.v[i] = in.T[4 + i] + dd[i] + da[i];
    }
  } else {
    Flux f;
    make_flux(in.T, in.w, f); // shared by both operators
    float ddx[4], dax[4];
    if (!tab.subcycled[k]) {
      dif_lon_fast(f, tab.dif_ccx[k] * 0.05f, ddx);
      adv_lon_full_fast(f, in.T, in.w, u, tab.adv_ccx[k] * (1.f / 3.f), dax);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      float T2h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      dif_lon_fast(f, tab.dif_ccx2[k] * 0.05f, ddx);
      clamp_add_fast(T1h, ddx);
      adv_lon_sub_fast(f, in.T, in.w, u, tab.adv_ccx2[k] * 0.05f, last_quad, dax);
      clamp_add_fast(T2h, dax);
#pragma unroll
      for (int i = 0; i < 4; ++i) { ddx[i] = T1h[i] - in.T[4 + i]; dax[i] = T2h[i] - in.T[4 + i]; }
    }
    float am, ap, ddy[4], day[4];
    adv_lat_coef(tab, k, ny, am, ap);
    lat_fast(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, v, tab.dif_ccy, am, ap, ddy, day);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dd = in.w0.v[i] * (ddx[i] + ddy[i]);
      const float da = dax[i] + day[i];
      r.v[i] = (in.T[4 + i] + dd) + da;
    }
  }
  return r;
}

// ---- chain rows: rows whose sub-cycle count is > 1 (src/greb.f90:656-717, 842-909).  One wave
// owns the row and iterates Jacobi sweeps through two LDS row buffers.
enum ChainMode { kChainDif = 0, kChainAdv = 1, kChainFused = 2 };

template <bool STRICT>
__device__ void chain_lon(const float* Trow, const float* wrow, const float* urow, float cc, int time2,
                          bool is_adv, int nq, int lane, float* bufA, float* bufB) {
  // result (T1h after time2 sweeps) is left in bufA
  for (int q = lane; q < nq; q += 64) st4(bufA + 4 * q, ld4(Trow + 4 * q));
  wave_lds_sync();
  for (int tt = 0; tt < time2; ++tt) {
    for (int q = lane; q < nq; q += 64) {
      float T[12], w[12], d[4];
      load_window(bufA, q, nq, T);
      load_window(wrow, q, nq, w);
      float T1h[4] = {T[4], T[5], T[6], T[7]};
      if (is_adv) {
        const f4 uq = ld4(urow + 4 * q);
        if (STRICT) {
          adv_lon_sub_strict(T, w, uq.v, cc, q == nq - 1, d);
          clamp_add(T1h, d);
        } else {
          Flux f; make_flux(T, w, f);
          adv_lon_sub_fast(f, T, w, uq.v, cc * 0.05f, q == nq - 1, d);
          clamp_add_fast(T1h, d);
        }
      } else {
        if (STRICT) {
          dif_lon_strict(T, w, cc, d);
          clamp_add(T1h, d);
        } else {
          Flux f; make_flux(T, w, f);
          dif_lon_fast(f, cc * 0.05f, d);
          clamp_add_fast(T1h, d);
        }
      }
      st4(bufB + 4 * q, f4{{T1h[0], T1h[1], T1h[2], T1h[3]}});
    }
    wave_lds_sync();
    float* t = bufA; bufA = bufB; bufB = t;
  }
  if (time2 & 1) { // make the result land in the caller's bufA
    for (int q = lane; q < nq; q += 64) st4(bufB + 4 * q, ld4(bufA + 4 * q));
    wave_lds_sync();
  }
}

// complete update of one chain row by one wave.  out: X_new row (fused) or dX row (dif/adv).
template <bool STRICT>
__device__ void chain_row(const Rows& X, const Rows& W, const Rows& U, const Rows& V, const RowTables& tab, int k,
                          int nq, int ny, int lane, int mode, float* scratch /* 4*nx */, float* out_row) {
  const int nx = 4 * nq;
  float* dA = scratch;          // diffusion T1h
  float* dB = scratch + nx;
  float* aA = scratch + 2 * nx; // advection T1h
  float* aB = scratch + 3 * nx;
  const float* Trow = X.row(k);
  const float* wrow = W.row(k);
  const bool do_dif = mode != kChainAdv, do_adv = mode != kChainDif;
  if (do_dif) chain_lon<STRICT>(Trow, wrow, nullptr, tab.dif_ccx2[k], tab.dif_time2[k], false, nq, lane, dA, dB);
  if (do_adv) chain_lon<STRICT>(Trow, wrow, U.row(k), tab.adv_ccx2[k], tab.adv_time2[k], true, nq, lane, aA, aB);
  for (int q = lane; q < nq; q += 64) {
    QuadIn in;
    gather(X, W, k, q, nq, ny, do_adv, in);
    float r[4];
    float dd[4] = {0, 0, 0, 0}, da[4] = {0, 0, 0, 0};
    if (do_dif) {
      const f4 t1 = ld4(dA + 4 * q);
      if (STRICT) {
#pragma clang fp contract(off)
        float dTy[4];
        dif_lat_strict(in.T0, in.Tm1, in.Tp1, in.wm1, in.wp1, tab.dif_ccy, k, ny, dTy);
#pragma unroll
        for (int i = 0; i < 4; ++i) dd[i] = in.w0.v[i] * ((t1.v[i] - in.T0.v[i]) + dTy[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float g = in.wm1.v[i] * (in.Tm1.v[i] - in.T0.v[i]) + in.wp1.v[i] * (in.Tp1.v[i] - in.T0.v[i]);
          dd[i] = in.w0.v[i] * ((t1.v[i] - in.T0.v[i]) + tab.dif_ccy * g);
        }
      }
    }
    if (do_adv) {
      const f4 t2 = ld4(aA + 4 * q);
      const f4 vq = ld4(V.row(k) + 4 * q);
      float dTy[4];
      if (STRICT) {
#pragma clang fp contract(off)
        adv_lat_strict(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, vq.v, tab.adv_ccy, k, ny, dTy);
#pragma unroll
        for (int i = 0; i < 4; ++i) da[i] = (t2.v[i] - in.T0.v[i]) + dTy[i];
      } else {
        float am, ap, dummy[4];
        adv_lat_coef(tab, k, ny, am, ap);
        lat_fast(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, vq.v, 0.f, am, ap, dummy, dTy);
#pragma unroll
        for (int i = 0; i < 4; ++i) da[i] = (t2.v[i] - in.T0.v[i]) + dTy[i];
      }
    }
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        r[i] = mode == kChainFused ? (in.T0.v[i] + dd[i]) + da[i] : (mode == kChainDif ? dd[i] : da[i]);
    }
    st4(out_row + 4 * q, f4{{r[0], r[1], r[2], r[3]}});
  }
}

__device__ __forceinline__ bool is_chain_row(const RowTables& tab, int k, int mode) {
  return (mode != kChainAdv && tab.dif_time2[k] > 1) || (mode != kChainDif && tab.adv_time2[k] > 1);
}

// ============================================================================================
// Batched standalone diffusion / advection: grid = (batch, bands).  Band rows [k0,k1) plus a
// halo of `halo` rows are staged in LDS with coalesced dwordx4 loads; output goes straight to
// HBM as dwordx4.  Algorithmic traffic for diffusion: read T1, read wz, write dX = 12 B/point.
// ============================================================================================
template <bool STRICT, int MODE>
__global__ __launch_bounds__(256) void sweep_kernel(const float* __restrict__ T1, const float* __restrict__ wz,
                                                    const float* __restrict__ ug, const float* __restrict__ vg,
                                                    float* __restrict__ dX, const RowTables* __restrict__ tabp,
                                                    int nx, int ny, int rows_per_band) {
  extern __shared__ __align__(16) float lds[];
  const RowTables& tab = *tabp;
  const int nq = nx >> 2;
  const int b = blockIdx.x;
  const int k0 = blockIdx.y * rows_per_band;
  const int k1 = min(ny, k0 + rows_per_band);
  constexpr int halo = MODE == kChainDif ? 1 : 2;
  const int r0 = max(0, k0 - halo), r1 = min(ny, k1 + halo);
  const int nrows = r1 - r0;
  float* sT = lds;
  float* sW = sT + nrows * nx;
  float* sU = sW + nrows * nx;                       // advection only: band rows k0..k1
  float* sV = sU + (MODE == kChainAdv ? (k1 - k0) * nx : 0);
  float* scratch = sV + (MODE == kChainAdv ? (k1 - k0) * nx : 0); // [nwaves][4*nx]
  const size_t fo = (size_t)b * nx * ny;
  for (int i = threadIdx.x; i < nrows * nq; i += blockDim.x) {
    st4(sT + 4 * i, ld4(T1 + fo + (size_t)r0 * nx + 4 * i));
    st4(sW + 4 * i, ld4(wz + fo + (size_t)r0 * nx + 4 * i));
  }
  if (MODE == kChainAdv)
    for (int i = threadIdx.x; i < (k1 - k0) * nq; i += blockDim.x) {
      st4(sU + 4 * i, ld4(ug + fo + (size_t)k0 * nx + 4 * i));
      st4(sV + 4 * i, ld4(vg + fo + (size_t)k0 * nx + 4 * i));
    }
  __syncthreads();
  const Rows X{sT, r0, nx}, W{sW, r0, nx}, U{sU, k0, nx}, V{sV, k0, nx};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  // chain rows: one wave each, round-robin
  int ci = 0;
  for (int k = k0; k < k1; ++k) {
    if (!is_chain_row(tab, k, MODE)) continue;
    if ((ci++ % nwaves) == wave)
      chain_row<STRICT>(X, W, U, V, tab, k, nq, ny, lane, MODE, scratch + wave * 4 * nx, dX + fo + (size_t)k * nx);
  }
  // everything else: quads, all threads
  for (int i = threadIdx.x; i < (k1 - k0) * nq; i += blockDim.x) {
    const int k = k0 + i / nq, q = i % nq;
    if (is_chain_row(tab, k, MODE)) continue;
    QuadIn in;
    gather(X, W, k, q, nq, ny, MODE != kChainDif, in);
    float out[4];
    if (MODE == kChainDif) {
      dif_quad<STRICT>(in, tab, k, ny, out);
    } else {
      const f4 uq = ld4(U.row(k) + 4 * q), vq = ld4(V.row(k) + 4 * q);
      adv_quad<STRICT>(in, uq.v, vq.v, tab, k, ny, q == nq - 1, out);
    }
    st4(dX + fo + (size_t)k * nx + 4 * q, f4{{out[0], out[1], out[2], out[3]}});
  }
}

static size_t sweep_lds_bytes(int nx, int rows, int halo, int extra_fields) {
  return (size_t)(((rows + 2 * halo) * 2 + rows * extra_fields) * nx + 4 * 4 * nx) * sizeof(float);
}
static int pick_band_rows(int nx, int ny, int halo, int extra_fields) {
  // whole field per workgroup when it fits ~44 KB (3+ workgroups share a CU's 160 KB); wide
  // grids get latitude bands of <= 64 KB
  const size_t limit = nx <= 128 ? 44 * 1024 : 64 * 1024;
  int rows = ny;
  while (rows > 2 && sweep_lds_bytes(nx, rows, halo, extra_fields) > limit) rows = (rows + 1) / 2;
  return rows;
}

template <int MODE>
static hipError_t launch_sweep(const float* T1, const float* wz, const float* u, const float* v, float* dX,
                               const RowTables* tab_dev, int nx, int ny, int batch, bool strict, hipStream_t s) {
  constexpr int halo = MODE == kChainDif ? 1 : 2;
  constexpr int extra = MODE == kChainAdv ? 2 : 0;
  const int rows = pick_band_rows(nx, ny, halo, extra);
  const int bands = (ny + rows - 1) / rows;
  const size_t lds = sweep_lds_bytes(nx, rows, halo, extra);
  dim3 grid(batch, bands), block(256);
  auto kern = strict ? sweep_kernel<true, MODE> : sweep_kernel<false, MODE>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, block, lds, s, T1, wz, u, v, dX, tab_dev, nx, ny, rows);
  return hipGetLastError();
}

hipError_t launch_diffusion(const float* T1, const float* wz, float* dX, const RowTables* tab_dev, int nx,
                            int ny, int batch, bool strict, hipStream_t s) {
  return launch_sweep<kChainDif>(T1, wz, nullptr, nullptr, dX, tab_dev, nx, ny, batch, strict, s);
}
hipError_t launch_advection(const float* T1, const float* wz, const float* u, const float* v, float* dX,
                            const RowTables* tab_dev, int nx, int ny, int batch, bool strict, hipStream_t s) {
  return launch_sweep<kChainAdv>(T1, wz, u, v, dX, tab_dev, nx, ny, batch, strict, s);
}

// ============================================================================================
// Fused circulation on a 96x48 member held in LDS (src/greb.f90:528-553).
//   NTR tracers; bulk threads: 384 per tracer, each owns quads (k = 3*ty + r, q = tx), r = 0..2;
//   chain waves: 2 per tracer (the two polar rows at 96x48), after the bulk threads.
// ============================================================================================
constexpr int G96_NX = 96, G96_NY = 48, G96_NQ = 24, G96_NP = 4608;

struct CircLds {
  float* X[2]; // [NTR][NP] each, double buffer
  float* W;    // [NTR][NP]
  float* U;    // [NP]
  float* V;    // [NP]
  float* scratch; // [2*NTR waves][4*NX]
  const RowTables* tab; // in LDS
};

template <bool STRICT, int NTR>
__device__ __forceinline__ void circ_substep(const CircLds& L, int cur) {
  const RowTables& tab = *L.tab;
  const int tid = threadIdx.x;
  constexpr int kBulk = 384 * NTR;
  const Rows U{L.U, 0, G96_NX}, V{L.V, 0, G96_NX};
  if (tid < kBulk) {
    const int tr = tid / 384, r = tid % 384, tx = r % G96_NQ, ty = r / G96_NQ;
    const Rows X{L.X[cur] + tr * G96_NP, 0, G96_NX}, W{L.W + tr * G96_NP, 0, G96_NX};
    float* out = L.X[cur ^ 1] + tr * G96_NP;
#pragma unroll 1
    for (int rr = 0; rr < 3; ++rr) {
      const int k = 3 * ty + rr;
      if (is_chain_row(tab, k, kChainFused)) continue;
      QuadIn in;
      gather(X, W, k, tx, G96_NQ, G96_NY, true, in);
      const f4 uq = ld4(U.row(k) + 4 * tx), vq = ld4(V.row(k) + 4 * tx);
      const f4 xn = substep_quad<STRICT>(in, uq.v, vq.v, tab, k, G96_NY, tx == G96_NQ - 1);
      st4(out + k * G96_NX + 4 * tx, xn);
    }
  } else {
    const int cw = (tid - kBulk) >> 6, lane = tid & 63;
    constexpr int nchainw = 2 * NTR;
    int ci = 0;
    for (int tr = 0; tr < NTR; ++tr) {
      const Rows X{L.X[cur] + tr * G96_NP, 0, G96_NX}, W{L.W + tr * G96_NP, 0, G96_NX};
      for (int k = 0; k < G96_NY; ++k) {
        if (!is_chain_row(tab, k, kChainFused)) continue;
        if ((ci++ % nchainw) == cw)
          chain_row<STRICT>(X, W, U, V, tab, k, G96_NQ, G96_NY, lane, kChainFused,
                            L.scratch + cw * 4 * G96_NX, L.X[cur ^ 1] + tr * G96_NP + k * G96_NX);
      }
    }
  }
}

// test mirror of circulation(): one field per workgroup, 512 threads
template <bool STRICT>
__global__ __launch_bounds__(512) void circulation_kernel(const float* __restrict__ Xin, const float* __restrict__ wz,
                                                          const float* __restrict__ ug, const float* __restrict__ vg,
                                                          float* __restrict__ dX, const RowTables* __restrict__ tabp,
                                                          int nsub) {
  extern __shared__ __align__(16) float lds[];
  CircLds L;
  L.X[0] = lds; L.X[1] = lds + G96_NP; L.W = lds + 2 * G96_NP; L.U = lds + 3 * G96_NP; L.V = lds + 4 * G96_NP;
  L.scratch = lds + 5 * G96_NP;
  RowTables* tab = reinterpret_cast<RowTables*>(L.scratch + 2 * 4 * G96_NX);
  L.tab = tab;
  const size_t fo = (size_t)blockIdx.x * G96_NP;
  for (int i = threadIdx.x; i < G96_NP / 4; i += blockDim.x) {
    st4(L.X[0] + 4 * i, ld4(Xin + fo + 4 * i)); st4(L.W + 4 * i, ld4(wz + fo + 4 * i));
    st4(L.U + 4 * i, ld4(ug + fo + 4 * i));     st4(L.V + 4 * i, ld4(vg + fo + 4 * i));
  }
  for (int i = threadIdx.x; i < (int)(sizeof(RowTables) / 4); i += blockDim.x)
    reinterpret_cast<int*>(tab)[i] = reinterpret_cast<const int*>(tabp)[i];
  __syncthreads();
  int cur = 0;
  for (int tt = 0; tt < nsub; ++tt) {
    circ_substep<STRICT, 1>(L, cur);
    __syncthreads();
    cur ^= 1;
  }
  for (int i = threadIdx.x; i < G96_NP / 4; i += blockDim.x) {
    const f4 a = ld4(L.X[cur] + 4 * i), b = ld4(Xin + fo + 4 * i);
    st4(dX + fo + 4 * i, f4{{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]}}); // :551
  }
}

__global__ void axpy2_kernel(float* __restrict__ X, const float* __restrict__ a, const float* __restrict__ b, size_t n) {
#pragma clang fp contract(off)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    X[i] = X[i] + a[i] + b[i];
}
__global__ void sub_kernel(float* __restrict__ d, const float* __restrict__ a, const float* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] = a[i] - b[i];
}

hipError_t launch_circulation(const float* X, const float* wz, const float* u, const float* v, float* dX,
                              float* scratch, const RowTables* tab_dev, int nx, int ny, int batch, int nsub,
                              bool strict, hipStream_t s) {
  if (nx == G96_NX && ny == G96_NY) {
    const size_t lds = (size_t)(5 * G96_NP + 2 * 4 * G96_NX) * sizeof(float) + sizeof(RowTables);
    auto kern = strict ? circulation_kernel<true> : circulation_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(batch), dim3(512), lds, s, X, wz, u, v, dX, tab_dev, nsub);
    return hipGetLastError();
  }
  // other grids: the member does not fit one CU's LDS -> one launch per operator per sub-step
  const size_t n = (size_t)batch * nx * ny;
  float *Xc = scratch, *dd = scratch + n, *da = scratch + 2 * n;
  hipError_t e = hipMemcpyAsync(Xc, X, n * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return e;
  for (int tt = 0; tt < nsub; ++tt) {
    if ((e = launch_diffusion(Xc, wz, dd, tab_dev, nx, ny, batch, strict, s)) != hipSuccess) return e;
    if ((e = launch_advection(Xc, wz, u, v, da, tab_dev, nx, ny, batch, strict, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(axpy2_kernel, dim3(1024), dim3(256), 0, s, Xc, dd, da, n);
  }
  hipLaunchKernelGGL(sub_kernel, dim3(1024), dim3(256), 0, s, dX, Xc, X, n);
  return hipGetLastError();
}

// ============================================================================================
// Point physics of one step (test mirror of a4-a8)
// ============================================================================================
__global__ void point_kernel(PointArgs a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.np) return;
  const size_t off = (size_t)(a.ityr - 1) * a.np, offm = (size_t)(a.ityr > 1 ? a.ityr - 2 : kNT - 1) * a.np;
  const float Ts = a.in5[p], Ta = a.in5[a.np + p], To = a.in5[2 * a.np + p], q = a.in5[3 * a.np + p];
  const float cap = a.in5[4 * a.np + p];
  const float zt = a.z_topo[p], gl = a.glacier[p], ez = a.wz_air[p];
  const float cld = a.cldclim[off + p], mld = a.mldclim[off + p];
  float albedo, sw, LWsurf, LWdown, em, Qlat, Qlat_air, dq_eva, dq_rain, dT_ocean, dTo;
  sw_radiation(a.phys, Ts, zt, gl, cld, a.sw_solar[(size_t)(a.ityr - 1) * a.ny + p / a.nx], albedo, sw);
  lw_radiation(a.phys, Ts, Ta, q, a.co2, ez, cld, a.tclim[off + p], LWsurf, LWdown, em);
  hydro(a.phys, Ts, q, a.uclim[off + p], a.vclim[off + p], zt, ez, a.swetclim[off + p], Qlat, Qlat_air, dq_eva, dq_rain);
  deep_ocean(a.phys, Ts, To, zt, mld, a.mldclim[offm + p], a.z_ocean[p], dT_ocean, dTo);
  float Qsens;
  {
#pragma clang fp contract(off)
    Qsens = a.phys.ct_sens * (Ta - Ts); // :295
  }
  const float capn = seaice(a.phys, Ts, zt, gl, mld, cap);
  const float o[15] = {albedo, sw, LWsurf, LWdown, em, Qsens, Qlat, Qlat_air, dq_eva, dq_rain, dT_ocean, dTo, capn, 0.f, 0.f};
  for (int i = 0; i < 15; ++i) a.out15[(size_t)i * a.np + p] = o[i];
}

hipError_t launch_point_physics(const PointArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(point_kernel, dim3((a.np + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ============================================================================================
// The fused member engine
// ============================================================================================
__device__ __constant__ int kMonthEnd[12] = {31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334, 365};
__device__ __constant__ int kMonthDays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31}; // :42

constexpr int kMemberThreads = 1024;

// Between model steps a member's five state fields live in HBM (`state`, L2/MALL-resident: 92 KB
// per member); inside a step the two transported tracers live in LDS.  Nothing but the LDS
// pointers is carried in registers across the 24 sub-steps, so the stencil code gets the whole
// register budget.
template <bool STRICT, bool FLUX>
__global__ __launch_bounds__(kMemberThreads) void member_kernel(MemberArgs a) {
  extern __shared__ __align__(16) float lds[];
  constexpr int NP = G96_NP;
  CircLds L;
  L.X[0] = lds; L.X[1] = lds + 2 * NP; L.W = lds + 4 * NP; L.U = lds + 6 * NP; L.V = lds + 7 * NP;
  L.scratch = lds + 8 * NP;
  RowTables* tab = reinterpret_cast<RowTables*>(L.scratch + 4 * 4 * G96_NX);
  L.tab = tab;
  const int m = blockIdx.x, tid = threadIdx.x;
  float* state = a.state + (size_t)m * 5 * NP;
  float* acc = a.acc + (size_t)m * 6 * NP;
  float* corr = a.corr + (size_t)a.corr_index[m] * 3 * kNT * NP;

  // ---- resident set-up: weights, tables and the two tracers to LDS
  for (int i = tid; i < NP / 4; i += kMemberThreads) {
    st4(L.W + 4 * i, ld4(a.wz_air + 4 * i));
    st4(L.W + NP + 4 * i, ld4(a.wz_vapor + 4 * i));
    st4(L.X[0] + 4 * i, ld4(state + NP + 4 * i));          // Tair
    st4(L.X[0] + NP + 4 * i, ld4(state + 3 * NP + 4 * i)); // q
  }
  {
    const RowTables* src = a.tabs + a.tab_index[m];
    for (int i = tid; i < (int)(sizeof(RowTables) / 4); i += kMemberThreads)
      reinterpret_cast<int*>(tab)[i] = reinterpret_cast<const int*>(src)[i];
  }
  int cur = 0;
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < a.nsteps; ++s) {
    const long long it = a.it0 + s;
    const int ityr = (int)((it - 1) % kNT) + 1;                       // :252
    const int jday = (int)(((it - 1) / 2) % 365) + 1;                 // :251
    const size_t off = (size_t)(ityr - 1) * NP;
    const size_t offm = (size_t)(ityr > 1 ? ityr - 2 : kNT - 1) * NP; // :507-508
    const int yr_rel = (int)((it - 1) / kNT - (a.it0 - 1) / kNT);     // whole years since launch start

    // wind slice of this step -> LDS (advection reads it 24 x 2 times)
    for (int i = tid; i < NP / 4; i += kMemberThreads) {
      st4(L.U + 4 * i, ld4(a.uclim + off + 4 * i));
      st4(L.V + 4 * i, ld4(a.vclim + off + 4 * i));
    }
    __syncthreads();

    // ---- circulation of Tair and q: 24 sub-steps (:543-550)
#pragma unroll 1
    for (int tt = 0; tt < a.nsub; ++tt) {
      circ_substep<STRICT, 2>(L, cur);
      __syncthreads();
      cur ^= 1;
    }

    // ---- point physics on the OLD state + Euler update (:254-268 / :328-361)
    int mon = -1; // 0-based month whose last day this is, else -1
    if (!FLUX && (it % 2 == 0))
      for (int mm = 0; mm < 12; ++mm) if (jday == kMonthEnd[mm]) mon = mm; // :975-976
    const Phys P = a.phys[m];
    const float co2 = FLUX ? a.co2_flux : a.co2[(size_t)m * a.co2_stride + a.co2_year0 + yr_rel]; // :924
#pragma unroll 1
    for (int p = tid; p < NP; p += kMemberThreads) {
#pragma clang fp contract(off)
      const float Ts1 = state[p], Ta1 = state[NP + p], To1 = state[2 * NP + p], q1 = state[3 * NP + p];
      const float cap = state[4 * NP + p];
      const float zt = a.z_topo[p], gl = a.glacier[p];
      const float dTa_crcl = L.X[cur][p] - Ta1; // :551
      const float dq_crcl = L.X[cur][NP + p] - q1;
      const float ez = L.W[p];
      const float tcl = a.tclim[off + p], cld = a.cldclim[off + p], mld = a.mldclim[off + p];
      float albedo, sw, LWsurf, LWdown, em, Qlat, Qlat_air, dq_eva, dq_rain, dT_ocean, dTo;
      sw_radiation(P, Ts1, zt, gl, cld, a.sw_solar[(size_t)(ityr - 1) * G96_NY + p / G96_NX], albedo, sw);
      lw_radiation(P, Ts1, Ta1, q1, co2, ez, cld, tcl, LWsurf, LWdown, em);
      const float Qsens = P.ct_sens * (Ta1 - Ts1); // :295
      hydro(P, Ts1, q1, L.U[p], L.V[p], zt, ez, a.swetclim[off + p], Qlat, Qlat_air, dq_eva, dq_rain);
      deep_ocean(P, Ts1, To1, zt, mld, a.mldclim[offm + p], a.z_ocean[p], dT_ocean, dTo);
      const float LWup = LWdown; // :432
      float Ts0, Ta0, To0, q0;
      if (FLUX) {
        const float dTs = P.dt * (sw + LWsurf - LWdown + Qlat + Qsens) / cap;                    // :333
        Ts0 = Ts1 + dTs + dT_ocean;                                                              // :334
        const float dTa = P.dt * (LWup + LWdown - em * LWsurf + Qlat_air - Qsens) / P.cap_air;   // :336
        Ta0 = Ta1 + dTa + dTa_crcl;                                                              // :337
        To0 = To1 + dTo;                                                                         // :339
        const float dq = P.dt * (dq_eva + dq_rain);                                              // :341
        q0 = q1 + dq + dq_crcl;                                                                  // :342
        const float TF = (tcl - Ts0) * cap / P.dt;                                               // :344-345
        corr[off + p] = TF;
        Ts0 = Ts1 + dTs + dT_ocean + TF * P.dt / cap;                                            // :347
        const float ToF = a.toclim[p] - To0;                                                     // :349
        corr[(size_t)2 * kNT * NP + off + p] = ToF;
        To0 = To1 + dTo + ToF;                                                                   // :351
        const float qF = a.qclim[off + p] - q0;                                                  // :353
        corr[(size_t)kNT * NP + off + p] = qF;
        q0 = q1 + dq + dq_crcl + qF;                                                             // :355
      } else {
        const float TF = corr[off + p], qF = corr[(size_t)kNT * NP + off + p], ToF = corr[(size_t)2 * kNT * NP + off + p];
        Ts0 = Ts1 + dT_ocean + P.dt * (sw + LWsurf - LWdown + Qlat + Qsens + TF) / cap;          // :258
        Ta0 = Ta1 + dTa_crcl + P.dt * (LWup + LWdown - em * LWsurf + Qlat_air - Qsens) / P.cap_air; // :260
        To0 = To1 + dTo + ToF;                                                                   // :262
        float dq = P.dt * (dq_eva + dq_rain) + dq_crcl + qF;                                     // :264
        if (dq <= -q1) dq = -0.9f * q1;                                                          // :265
        q0 = q1 + dq;                                                                            // :266
      }
      state[p] = Ts0; state[NP + p] = Ta0; state[2 * NP + p] = To0; state[3 * NP + p] = q0;
      state[4 * NP + p] = seaice(P, Ts0, zt, gl, mld, cap);                                      // :268/:357
      L.X[cur][p] = Ta0; L.X[cur][NP + p] = q0;
      // accumulation (:945, :974)
      float tsmn = acc[5 * NP + p] + Ts0;
      if (!FLUX) {
        float a0 = acc[p] + Ts0, a1 = acc[NP + p] + Ta0, a2 = acc[2 * NP + p] + To0, a3 = acc[3 * NP + p] + q0,
              a4 = acc[4 * NP + p] + albedo;
        if (mon >= 0) { // :975-984
          const float ndm = (float)(kMonthDays[mon] * 2);
          float* rec = a.monthly + (((size_t)m * a.monthly_years + (a.year_out0 + yr_rel)) * 12 + mon) * 5 * NP;
          rec[p] = a0 / ndm; rec[NP + p] = a1 / ndm; rec[2 * NP + p] = a2 / ndm; rec[3 * NP + p] = a3 / ndm;
          rec[4 * NP + p] = a4 / ndm;
          a0 = a1 = a2 = a3 = a4 = 0.f;
        }
        acc[p] = a0; acc[NP + p] = a1; acc[2 * NP + p] = a2; acc[3 * NP + p] = a3; acc[4 * NP + p] = a4;
      }
      if (ityr == kNT) { // :948-956
        tsmn = tsmn / (float)kNT;
        L.X[cur ^ 1][p] = tsmn; // the idle tracer buffer doubles as reduction scratch
        acc[5 * NP + p] = 0.f;
      } else {
        acc[5 * NP + p] = tsmn;
      }
    }
    if (ityr == kNT) {
      __syncthreads();
      if (tid == 0 && a.yearly) {
#pragma clang fp contract(off)
        float sum = 0.f; // the reference's sum() lowers to a sequential fp32 loop; same order here
        for (int i = 0; i < NP; ++i) sum += L.X[cur ^ 1][i];
        float* y = a.yearly + ((size_t)m * a.yearly_years + (a.yearly_year0 + yr_rel)) * 2;
        y[0] = sum / (float)NP - 273.15f;                                               // :954
        y[1] = L.X[cur ^ 1][(a.ipy - 1) * G96_NX + (a.ipx - 1)] - 273.15f;
      }
    }
    __syncthreads();
  }
  // Ts/To/cap_surf are already in `state`; Tair and q were written there every step as well.
}

hipError_t launch_member_kernel(const MemberArgs& a, int n_members, bool strict, hipStream_t s) {
  if (a.nx != G96_NX || a.ny != G96_NY) return hipErrorInvalidValue;
  const size_t lds = (size_t)(8 * G96_NP + 4 * 4 * G96_NX) * sizeof(float) + sizeof(RowTables);
  void (*kern)(MemberArgs);
  if (a.flux_phase) kern = strict ? member_kernel<true, true> : member_kernel<false, true>;
  else kern = strict ? member_kernel<true, false> : member_kernel<false, false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(n_members), dim3(kMemberThreads), lds, s, a);
  return hipGetLastError();
}

} // namespace greb
