// greb_stencil.h -- quad-level assembly of the stencil pieces (greb_device.h) shared by the
// batched sweep kernels (greb_kernels.hip) and the fused member engine (greb_member.hip).
#pragma once
#include "greb_chain6.h"
#include "greb_device.h"

namespace greb {

__device__ __forceinline__ void wave_lds_sync() {
  // Order this wave's LDS writes before its later LDS reads (one wave owns a chain row).  LDS
  // instructions of one wave execute in issue order, so only the compiler must be held back.  A
  // __builtin_amdgcn_fence here would also emit s_waitcnt vmcnt(0) and make the chain wave sit
  // out the latency of every global prefetch/store it has in flight (measured: 2000 cycles per
  // Jacobi sweep instead of ~600).
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// A latitude band [r0, r1) of a field staged in LDS, row-major, nx floats per row.
struct Rows {
  const lfloat* base;
  int r0, nx;
  __device__ __forceinline__ const lfloat* row(int k) const { return base + (k - r0) * nx; }
};

struct QuadIn {
  float T[12], w[12];
  f4 T0, w0, Tm2, Tm1, Tp1, Tp2, wm2, wm1, wp1, wp2;
};


// the constants one latitude row needs (a register-resident slice of RowTables)
struct RowK {
  float dif_cc;  // dif_ccx (full-row branch) or dif_ccx2 (sub-cycled branch), :582 / :654
  float adv_cc;  // adv_ccx or adv_ccx2, :753 / :840
  float dif_ccy, adv_ccy;
  int sub;       // !(dxlat(k) > 2.5e5)
  int dif_time2, adv_time2;
};
// LDS-staged copy of the row constants: 8 dwords per row.  Kernels that keep global loads in
// flight across their compute phase (prefetch) must not read the tables from global memory there:
// s_waitcnt vmcnt is in-order, so one table load would wait for every prefetch ahead of it.
constexpr int kRowKWords = 8;
// rows k_begin .. k_end-1 to dst[0 ..]
__device__ __forceinline__ void stage_row_consts(lfloat* dst, const RowTables& tab, int k_begin, int k_end) {
  for (int k = k_begin + threadIdx.x; k < k_end; k += blockDim.x) {
    const int sub = tab.subcycled[k];
    lfloat* d = dst + (k - k_begin) * kRowKWords;
    d[0] = sub ? tab.dif_ccx2[k] : tab.dif_ccx[k];
    d[1] = sub ? tab.adv_ccx2[k] : tab.adv_ccx[k];
    d[2] = tab.dif_ccy; d[3] = tab.adv_ccy;
    d[4] = __int_as_float(sub); d[5] = __int_as_float(tab.dif_time2[k]); d[6] = __int_as_float(tab.adv_time2[k]);
    d[7] = 0.f;
  }
}
__device__ __forceinline__ RowK row_consts(const lfloat* src, int k) {
  const f4 a = ld4(src + k * kRowKWords), b = ld4(src + k * kRowKWords + 4);
  RowK r;
  r.dif_cc = a.v[0]; r.adv_cc = a.v[1]; r.dif_ccy = a.v[2]; r.adv_ccy = a.v[3];
  r.sub = __float_as_int(b.v[0]); r.dif_time2 = __float_as_int(b.v[1]); r.adv_time2 = __float_as_int(b.v[2]);
  return r;
}
__device__ __forceinline__ bool is_chain_row(const RowK& r, int mode) {
  return (mode != 1 /*kChainAdv*/ && r.dif_time2 > 1) || (mode != 0 /*kChainDif*/ && r.adv_time2 > 1);
}
__device__ __forceinline__ RowK row_consts(const RowTables& tab, int k) {
  RowK r;
  r.sub = tab.subcycled[k];
  r.dif_cc = r.sub ? tab.dif_ccx2[k] : tab.dif_ccx[k];
  r.adv_cc = r.sub ? tab.adv_ccx2[k] : tab.adv_ccx[k];
  r.dif_ccy = tab.dif_ccy; r.adv_ccy = tab.adv_ccy;
  r.dif_time2 = tab.dif_time2[k]; r.adv_time2 = tab.adv_time2[k];
  return r;
}

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
__device__ __forceinline__ void dif_quad(const QuadIn& in, const RowK& rk, int k, int ny, float out[4]) {
  float dTx[4], dTy[4];
  if (STRICT) {
#pragma clang fp contract(off)
    if (!rk.sub) {
      dif_lon_strict(in.T, in.w, rk.dif_cc, dTx);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      dif_lon_strict(in.T, in.w, rk.dif_cc, dTx);
      clamp_add(T1h, dTx);
#pragma unroll
      for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - in.T[4 + i]; // :718
    }
    dif_lat_strict(in.T0, in.Tm1, in.Tp1, in.wm1, in.wp1, rk.dif_ccy, k, ny, dTy);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = in.w0.v[i] * (dTx[i] + dTy[i]); // :721
  } else {
    Flux f;
    make_flux(in.T, in.w, f);
    if (!rk.sub) {
      dif_lon_fast(f, rk.dif_cc * 0.05f, dTx);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      dif_lon_fast(f, rk.dif_cc * 0.05f, dTx);
      clamp_add_fast(T1h, dTx);
#pragma unroll
      for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - in.T[4 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float g = in.wm1.v[i] * (in.Tm1.v[i] - in.T0.v[i]) + in.wp1.v[i] * (in.Tp1.v[i] - in.T0.v[i]);
      out[i] = in.w0.v[i] * (dTx[i] + rk.dif_ccy * g);
    }
  }
}

__device__ __forceinline__ void adv_lat_coef(float adv_ccy, int k, int ny, float& am, float& ap) {
  const float third = adv_ccy * (1.f / 3.f);
  am = (k == 1) ? adv_ccy : third;      // :766-769
  ap = (k == ny - 2) ? adv_ccy : third; // :784-787
  if (k == 0) am = 0.f;
  if (k == ny - 1) ap = 0.f;
}

template <bool STRICT>
__device__ __forceinline__ void adv_quad(const QuadIn& in, const float u[4], const float v[4],
                                         const RowK& rk, int k, int ny, bool last_quad, float out[4]) {
  float dTx[4], dTy[4];
  if (STRICT) {
#pragma clang fp contract(off)
    if (!rk.sub) {
      adv_lon_full_strict(in.T, in.w, u, rk.adv_cc, dTx);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      adv_lon_sub_strict(in.T, in.w, u, rk.adv_cc, last_quad, dTx);
      clamp_add(T1h, dTx);
#pragma unroll
      for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - in.T[4 + i]; // :910
    }
    adv_lat_strict(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, v, rk.adv_ccy, k, ny, dTy);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = dTx[i] + dTy[i]; // :913
  } else {
    Flux f;
    make_flux(in.T, in.w, f);
    if (!rk.sub) {
      adv_lon_full_fast(f, in.T, in.w, u, rk.adv_cc * (1.f / 3.f), dTx);
    } else {
      float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
      adv_lon_sub_fast(f, in.T, in.w, u, rk.adv_cc * 0.05f, last_quad, dTx);
      clamp_add_fast(T1h, dTx);
#pragma unroll
      for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - in.T[4 + i];
    }
    float am, ap, dd[4];
    adv_lat_coef(rk.adv_ccy, k, ny, am, ap);
    lat_fast(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, v, 0.f, am, ap, dd, dTy);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = dTx[i] + dTy[i];
  }
}

// ---- chain rows: rows whose sub-cycle count is > 1 (src/greb.f90:656-717, 842-909).  One wave
// owns the row and iterates Jacobi sweeps through two LDS row buffers.
enum ChainMode { kChainDif = 0, kChainAdv = 1, kChainFused = 2 };

__host__ __device__ __forceinline__ bool is_chain_row(const RowTables& tab, int k, int mode) {
  return (mode != kChainAdv && tab.dif_time2[k] > 1) || (mode != kChainDif && tab.adv_time2[k] > 1);
}

// time2 Jacobi sweeps of one row; result (T1h) is left in bufA.  urow: raw zonal wind of the row.
template <bool STRICT>
__device__ void chain_lon(const lfloat* Trow, const lfloat* wrow, const lfloat* urow, float cc, int time2,
                          bool is_adv, int nq, int lane, lfloat* bufA, lfloat* bufB) {
  for (int q = lane; q < nq; q += 64) st4(bufA + 4 * q, ld4(Trow + 4 * q));
  wave_lds_sync();
  lfloat* src = bufA;
  lfloat* dst = bufB;
  for (int tt = 0; tt < time2; ++tt) {
    for (int q = lane; q < nq; q += 64) {
      float T[12], w[12], d[4];
      load_window((const lfloat*)src, q, nq, T);
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
      st4(dst + 4 * q, f4{{T1h[0], T1h[1], T1h[2], T1h[3]}});
    }
    wave_lds_sync();
    lfloat* t = src; src = dst; dst = t;
  }
  if (src != bufA) { // odd sweep count: make the result land in the caller's bufA
    for (int q = lane; q < nq; q += 64) st4(bufA + 4 * q, ld4((const lfloat*)src + 4 * q));
    wave_lds_sync();
  }
}

// The same time2 sweeps for a row of exactly 64*P points (P = 6: 384x192) with the row held in REGISTERS:
// lane l owns points P*l .. P*l+P-1, the 3+3 halo points come from the neighbouring lanes by wave rotates
// (v_mov_b32_dpp wave_ror:1 / wave_rol:1 -- the rotate is the row's periodic boundary), the weights and the
// wind of the row are loaded once.  No LDS traffic and no LDS round trip per sweep: the long polar chains of
// the fine grid (up to 225 dependent sweeps per diffusion call) are bound by exactly that latency.
__device__ __forceinline__ float wave_from_prev(float x) { // lane l <- lane l-1, lane 0 <- lane 63
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x13C /* wave_ror:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_from_next(float x) { // lane l <- lane l+1, lane 63 <- lane 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x134 /* wave_rol:1 */, 0xf, 0xf, false));
}

// FAST: during a chain the weights, the row constant and the wind are fixed, so the increment of point c is a fixed
// linear form in the six differences around it, d(c) = sum_m K[c][m] * e[c-3+m] with e[j] = T[j+1] - T[j]; w = the lane's P
// weights with three halo points on either side, u its winds.  The coefficients depend on nothing that changes between
// the sub-steps of a circulation call (greb_circ_rows.hip builds them once per call).
template <int P>
__device__ __forceinline__ void chain_coefficients(const float (&w)[P + 6], const float (&u)[P], float cc, bool is_adv,
                                                   bool bug_lane, float (&K)[P][6]) {
  const float cs = cc * 0.05f;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int c = 3 + i;
    if (is_adv) { // -up*(10 Pp[c] + 4 Pp[c+1] + Pp[c+2]) - um*(10 Pm[c-1] + 4 Pm[c-2] + Pm[c-3]), :845-851
      const float um = cs * split_m(u[i]), up = cs * split_p(u[i]);
      K[i][0] = -um * w[c - 3]; K[i][1] = -4.f * um * w[c - 2]; K[i][2] = -10.f * um * w[c - 1];
      K[i][3] = -10.f * up * w[c + 1]; K[i][4] = -4.f * up * w[c + 2]; K[i][5] = -up * w[c + 3];
      if (i == P - 3 && bug_lane) { // :881: the 4* term vanishes, the 1* term is w(1)*(T(xdim-1)-T(1)) = w[c+3]*(e[c+1]+e[c+2])
        K[i][4] = -up * w[c + 3]; K[i][5] = -up * w[c + 3];
      }
    } else { // 6(Pp[c] - Pm[c-1]) + 3(Pp[c+1] - Pm[c-2]) + (Pp[c+2] - Pm[c-3]), :595-600 in edge-flux form
      K[i][0] = -cs * w[c - 3]; K[i][1] = -3.f * cs * w[c - 2]; K[i][2] = -6.f * cs * w[c - 1];
      K[i][3] = 6.f * cs * w[c + 1]; K[i][4] = 3.f * cs * w[c + 2]; K[i][5] = cs * w[c + 3];
    }
  }
}

constexpr int kChainPrioSweeps = 32;
// the chain on a register window: T[0..P+5] / w[0..P+5] = the lane's P points with three halo points on either side,
// u[0..P-1] the zonal wind of its points; the result (T1h) is left in T[3..P+2]
template <bool STRICT, int P>
__device__ __forceinline__ void chain_window(float (&T)[P + 6], const float (&w)[P + 6], const float (&u)[P], float cc,
                                             int time2, bool is_adv, int lane, bool prio = true) {
  const bool bug_lane = lane == 63; // its point P-3 is longitude xdim-2 (1-based), src/greb.f90:881
  // the long chains set the length of the launch: they issue ahead of whatever shares their SIMD (measured 39.0 ->
  // 37.8 us per sub-step launch for one member)
  if (prio && time2 >= kChainPrioSweeps) __builtin_amdgcn_s_setprio(3);
  float K[P][6]; // FAST: d(c) = sum_m K[c][m] * e[c-3+m]
  if (!STRICT) chain_coefficients<P>(w, u, cc, is_adv, bug_lane, K);
  if (STRICT) {
    for (int tt = 0; tt < time2; ++tt) {
      if (tt > 0) { // refresh the halo from the neighbours' new values
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          T[i] = wave_from_prev(T[P + i]);         // the previous lane's last three own points
          T[P + 3 + i] = wave_from_next(T[3 + i]); // the next lane's first three own points
        }
      }
      float Tn[P];
#pragma unroll
      for (int i = 0; i < P; ++i) {
#pragma clang fp contract(off)
        const int c = 3 + i;
        float d = is_adv ? adv_lon_sub_point_strict(T, w, u[i], cc, c, bug_lane && i == P - 3)
                         : div20(cc * dif_S_strict(T, w, c));
        if (d <= -T[c]) d = -0.9f * T[c]; // :715 / :907
        Tn[i] = T[c] + d;
      }
#pragma unroll
      for (int i = 0; i < P; ++i) T[3 + i] = Tn[i];
    }
  } else {
    // FAST: the weights, the row constant and the wind do not change during the chain, so the increment of point c
    // is a fixed linear form in the six differences e[m] = T[m+1] - T[m] around it (coefficients K, built once
    // before the loop): 6 x (1 mul + 5 fma) per sweep instead of the edge-flux form's 22 products + 6 x 6.
    static_assert(STRICT || P == 6, "chain_run6 (greb_chain6.h) is written for 6 points per lane");
    float own[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) own[i] = T[3 + i];
    const bool positive = !is_adv && chain_stays_positive(own, K, time2);
    chain_run6<false>(own, chain_pack(K), time2, positive);
#pragma unroll
    for (int i = 0; i < 6; ++i) T[3 + i] = own[i];
  }
  if (prio && time2 >= kChainPrioSweeps) __builtin_amdgcn_s_setprio(0);
}
template <bool STRICT, int P>
__device__ void chain_lon_regs(const lfloat* Trow, const lfloat* wrow, const lfloat* urow, float cc, int time2,
                               bool is_adv, int lane, lfloat* bufA) {
  constexpr int NXR = 64 * P, W = P + 6;
  float T[W], w[W], u[P];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    int j = P * lane - 3 + i;
    j = j < 0 ? j + NXR : (j >= NXR ? j - NXR : j);
    w[i] = wrow[j];
    T[i] = Trow[j];
  }
#pragma unroll
  for (int i = 0; i < P; ++i) u[i] = is_adv ? urow[P * lane + i] : 0.f;
  chain_window<STRICT, P>(T, w, u, cc, time2, is_adv, lane);
#pragma unroll
  for (int i = 0; i < P; ++i) bufA[P * lane + i] = T[3 + i];
  wave_lds_sync();
}

// floats of LDS one wave needs for chain_row's sub-cycled results
__host__ __device__ constexpr int chain_row_scratch(int nx) { return (nx == 384 ? 2 : 4) * nx; }

// complete update of one chain row by one wave.  out_row (any address space): the X_new row
// (fused) or the dX row (dif / adv only).
template <bool STRICT, typename OutP>
__device__ void chain_row(const Rows& X, const Rows& W, const Rows& U, const Rows& V, const RowK& rk, int k,
                          int nq, int ny, int lane, int mode, lfloat* scratch /* 4*nx */, OutP out_row) {
  const int nx = 4 * nq;
  // a 384-point row iterates in registers and only parks its result; other rows ping-pong between two buffers
  lfloat* dA = scratch;          // diffusion T1h
  lfloat* dB = scratch + nx;
  lfloat* aA = scratch + (nx == 384 ? 1 : 2) * nx; // advection T1h
  lfloat* aB = aA + nx;
  const bool do_dif = mode != kChainAdv, do_adv = mode != kChainDif;
  if (nx == 384) { // the row fits the wave's registers, 6 points per lane
    if (do_dif) chain_lon_regs<STRICT, 6>(X.row(k), W.row(k), nullptr, rk.dif_cc, rk.dif_time2, false, lane, dA);
    if (do_adv) chain_lon_regs<STRICT, 6>(X.row(k), W.row(k), U.row(k), rk.adv_cc, rk.adv_time2, true, lane, aA);
  } else {
    if (do_dif) chain_lon<STRICT>(X.row(k), W.row(k), nullptr, rk.dif_cc, rk.dif_time2, false, nq, lane, dA, dB);
    if (do_adv) chain_lon<STRICT>(X.row(k), W.row(k), U.row(k), rk.adv_cc, rk.adv_time2, true, nq, lane, aA, aB);
  }
  for (int q = lane; q < nq; q += 64) {
    QuadIn in;
    gather(X, W, k, q, nq, ny, do_adv, in);
    float r[4];
    float dd[4] = {0, 0, 0, 0}, da[4] = {0, 0, 0, 0};
    if (do_dif) {
      const f4 t1 = ld4((const lfloat*)dA + 4 * q);
      if (STRICT) {
#pragma clang fp contract(off)
        float dTy[4];
        dif_lat_strict(in.T0, in.Tm1, in.Tp1, in.wm1, in.wp1, rk.dif_ccy, k, ny, dTy);
#pragma unroll
        for (int i = 0; i < 4; ++i) dd[i] = in.w0.v[i] * ((t1.v[i] - in.T0.v[i]) + dTy[i]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float g = in.wm1.v[i] * (in.Tm1.v[i] - in.T0.v[i]) + in.wp1.v[i] * (in.Tp1.v[i] - in.T0.v[i]);
          dd[i] = in.w0.v[i] * ((t1.v[i] - in.T0.v[i]) + rk.dif_ccy * g);
        }
      }
    }
    if (do_adv) {
      const f4 t2 = ld4((const lfloat*)aA + 4 * q);
      const f4 vq = ld4(V.row(k) + 4 * q);
      float dTy[4];
      if (STRICT) {
#pragma clang fp contract(off)
        adv_lat_strict(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, vq.v, rk.adv_ccy, k, ny, dTy);
#pragma unroll
        for (int i = 0; i < 4; ++i) da[i] = (t2.v[i] - in.T0.v[i]) + dTy[i];
      } else {
        float am, ap, dummy[4];
        adv_lat_coef(rk.adv_ccy, k, ny, am, ap);
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

} // namespace greb
