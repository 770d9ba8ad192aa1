// greb_pair.h -- FAST stencil arithmetic on (Tair, q) PAIRS for the fused member engine.
//
// Both transported tracers see the same winds and the same stencil; only their weights differ.
// The engine therefore stores them interleaved in LDS -- X[k][j][{Tair,q}] -- so one dwordx4 load
// delivers two longitudes of both tracers and every arithmetic operation is a packed
// v_pk_{add,mul,fma}_f32 on an even-aligned register pair: no shuffles to form the packed operands
// (the scalar formulation spent ~22 % of its VALU issue slots on v_mov / v_pk_mov), and the wind
// sign split, the address arithmetic and the row constants are shared by the two tracers.
// Same formulas as greb_device.h's FAST pieces (edge-flux form, pre-scaled winds).
#pragma once
#include "greb_device.h"

namespace greb {

// v2 = (Tair, q) register pair: typedef in greb_device.h

// a quad of 4 longitudes x 2 tracers = 8 floats = two dwordx4.
// LDS row layout [half][quad][4]: half 0 holds longitudes (4q, 4q+1), half 1 (4q+2, 4q+3), each as
// (Tair,q) pairs.  Consecutive lanes (consecutive quads) then touch consecutive 16-byte slots in
// each of the two ds_read_b128, which is bank-conflict-free; a plain [quad][8] layout strides the
// lanes by 32 B and halves the LDS rate (measured: the sub-step loop became LDS-bound).
constexpr int kHalfRow = 96; // floats per half-row at nx = 96
struct q8 {
  v2 v[4];
};
__device__ __forceinline__ q8 ld8(const lfloat* row, int q) {
  const vfloat4 a = *(const __attribute__((address_space(3))) vfloat4*)(row + 4 * q);
  const vfloat4 b = *(const __attribute__((address_space(3))) vfloat4*)(row + kHalfRow + 4 * q);
  q8 r;
  r.v[0] = v2{a.x, a.y}; r.v[1] = v2{a.z, a.w}; r.v[2] = v2{b.x, b.y}; r.v[3] = v2{b.z, b.w};
  return r;
}
__device__ __forceinline__ void st8(lfloat* row, int q, const q8& x) {
  vfloat4 a, b;
  a.x = x.v[0].x; a.y = x.v[0].y; a.z = x.v[1].x; a.w = x.v[1].y;
  b.x = x.v[2].x; b.y = x.v[2].y; b.z = x.v[3].x; b.w = x.v[3].y;
  *(__attribute__((address_space(3))) vfloat4*)(row + 4 * q) = a;
  *(__attribute__((address_space(3))) vfloat4*)(row + kHalfRow + 4 * q) = b;
}
__device__ __forceinline__ q8 zero8() {
  q8 r;
  r.v[0] = r.v[1] = r.v[2] = r.v[3] = v2{0.f, 0.f};
  return r;
}

// a - b as ONE packed instruction.  Left to itself the compiler lowers the latitudinal differences of lat2()
// to two scalar v_sub_f32 each (32 per task); the sub-step loop is VALU-issue bound, so that is 16 wasted slots.
__device__ __forceinline__ v2 pk_sub(v2 a, v2 b) {
  v2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// max(u,0) / min(u,0) of a finite wind: v_med3_f32 needs no canonicalising v_max(x,x) in front (fmaxf does),
// and min(u,0) = u - max(u,0) exactly
__device__ __forceinline__ void split_sign(float u, float& pos, float& neg) {
  pos = __builtin_amdgcn_fmed3f(u, 0.f, 3.0e38f); // (an infinite bound is folded back into v_max)
  neg = u - pos;
}

struct Flux2 {
  v2 Pp[10]; // Pp[m] = w[m+1]*(T[m+1]-T[m]), m = 4..9
  v2 Pm[10]; // Pm[m] = w[m]  *(T[m+1]-T[m]), m = 1..6
};

__device__ __forceinline__ void make_flux2(const v2 T[12], const v2 w[12], Flux2& f) {
#pragma unroll
  for (int m = 1; m <= 9; ++m) {
    const v2 e = T[m + 1] - T[m];
    if (m >= 4) f.Pp[m] = w[m + 1] * e;
    if (m <= 6) f.Pm[m] = w[m] * e;
  }
}

// cs = dif_cc/20 (src/greb.f90:595-600 in edge-flux form)
__device__ __forceinline__ void dif_lon2(const Flux2& f, float cs, v2 d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    const v2 a = f.Pp[c] - f.Pm[c - 1], b = f.Pp[c + 1] - f.Pm[c - 2], g = f.Pp[c + 2] - f.Pm[c - 3];
    d[i] = cs * (6.f * a + (3.f * b + g));
  }
}

// um = (ccx/3)*max(u,0), up = (ccx/3)*min(u,0)  (src/greb.f90:802-806)
__device__ __forceinline__ void adv_lon_full2(const Flux2& f, const v2 T[12], const v2 w[12], const float um[4],
                                              const float up[4], v2 d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    const v2 em2 = w[c - 2] * (T[c] - T[c - 2]), ep2 = w[c + 2] * (T[c] - T[c + 2]);
    d[i] = up[i] * (ep2 - f.Pp[c]) - um[i] * (f.Pm[c - 1] + em2);
  }
}

// um = (ccx2/20)*max(u,0), ...  (src/greb.f90:845-851, index bug :881 for the last quad's 2nd point)
__device__ __forceinline__ void adv_lon_sub2(const Flux2& f, const v2 T[12], const v2 w[12], const float um[4],
                                             const float up[4], bool last_quad, v2 d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    const v2 am = 10.f * f.Pm[c - 1] + (4.f * f.Pm[c - 2] + f.Pm[c - 3]);
    v2 ap = 10.f * f.Pp[c] + (4.f * f.Pp[c + 1] + f.Pp[c + 2]);
    if (i == 1) {
      const v2 bug = 10.f * f.Pp[c] - w[c + 3] * (T[c + 1] - T[c + 3]);
      ap = last_quad ? bug : ap;
    }
    d[i] = -up[i] * ap - um[i] * am;
  }
}

// clamp + accumulate (src/greb.f90:715-716, 907-908), per component
__device__ __forceinline__ void clamp_add2(v2 T1h[4], v2 d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v2 dd = d[i];
    dd.x = (dd.x <= -T1h[i].x) ? -0.9f * T1h[i].x : dd.x;
    dd.y = (dd.y <= -T1h[i].y) ? -0.9f * T1h[i].y : dd.y;
    d[i] = dd;
    T1h[i] = T1h[i] + dd;
  }
}

// latitudinal diffusion + advection; vm = am*max(v,0), vp = ap*min(v,0); missing rows have w = 0
__device__ __forceinline__ void lat2(const v2 T0[4], const v2 Tm2[4], const v2 Tm1[4], const v2 Tp1[4],
                                     const v2 Tp2[4], const v2 wm2[4], const v2 wm1[4], const v2 wp1[4],
                                     const v2 wp2[4], const float vm[4], const float vp[4], float ccy_dif,
                                     v2 ddif[4], v2 dadv[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const v2 gm1 = wm1[i] * pk_sub(Tm1[i], T0[i]), gp1 = wp1[i] * pk_sub(Tp1[i], T0[i]);
    const v2 dm2 = wm2[i] * pk_sub(T0[i], Tm2[i]), dp2 = wp2[i] * pk_sub(T0[i], Tp2[i]);
    ddif[i] = ccy_dif * (gm1 + gp1);
    dadv[i] = vp[i] * (dp2 - gp1) - vm[i] * (dm2 - gm1);
  }
}

// One row of a pair-tile: X_new = (X + dX_diffuse) + dX_advec for 4 longitudes x 2 tracers.
//   T[12], w[12]: the row's window (longitudes 4q-4 .. 4q+7), lat neighbours as q8
template <bool SUB>
__device__ __forceinline__ q8 substep_pair(const v2 T[12], const v2 w[12], const q8& Tm2, const q8& Tm1,
                                           const q8& Tp1, const q8& Tp2, const q8& wm2, const q8& wm1,
                                           const q8& wp1, const q8& wp2, const float um[4], const float up[4],
                                           const float vm[4], const float vp[4], float cs_dif, float ccy_dif,
                                           bool last_quad, bool calm_q = false) {
  Flux2 f;
  make_flux2(T, w, f);
  v2 ddx[4], dax[4], ddy[4], day[4];
  dif_lon2(f, cs_dif, ddx);
  if (SUB) {
    adv_lon_sub2(f, T, w, um, up, last_quad, dax);
    // The sub-cycle clamp `where(dTxh <= -T1h) dTxh = -0.9*T1h` (:715, :907) fires only where an increment would take
    // the tracer to zero or below.  d <= -T implies fl(T + d) <= 0 (rounding is monotonic), so one min over the 16
    // updated values (v_min3 chain) decides conservatively whether any lane component needs the reference's
    // per-component select; the common path is 9 instructions instead of 16 compares + 16 selects per row.
    v2 T1h[4], T2h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { T1h[i] = T[4 + i] + ddx[i]; T2h[i] = T[4 + i] + dax[i]; }
    float mn = min3f(T1h[0].x, T1h[0].y, T1h[1].x);
    mn = min3f(mn, T1h[1].y, T1h[2].x); mn = min3f(mn, T1h[2].y, T1h[3].x); mn = min3f(mn, T1h[3].y, T2h[0].x);
    mn = min3f(mn, T2h[0].y, T2h[1].x); mn = min3f(mn, T2h[1].y, T2h[2].x); mn = min3f(mn, T2h[2].y, T2h[3].x);
    mn = min3f(mn, T2h[3].y, T2h[3].y);
    if (__builtin_expect(!(mn > 0.f), 0)) { // also taken for a NaN: the reference's own comparisons then decide
#pragma unroll
      for (int i = 0; i < 4; ++i) { T1h[i] = T[4 + i]; T2h[i] = T[4 + i]; }
      clamp_add2(T1h, ddx);
      clamp_add2(T2h, dax);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { ddx[i] = T1h[i] - T[4 + i]; dax[i] = T2h[i] - T[4 + i]; } // :718, :910
  } else {
    adv_lon_full2(f, T, w, um, up, dax);
  }
  lat2(&T[4], Tm2.v, Tm1.v, Tp1.v, Tp2.v, wm2.v, wm1.v, wp1.v, wp2.v, vm, vp, ccy_dif, ddy, day);
  q8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const v2 dd = w[4 + i] * (ddx[i] + ddy[i]); // :721
    v2 da = dax[i] + day[i];                    // :913
    if (calm_q) da.y = 0.f; // experiment: vapour is diffused but not advected (greb.original.model.f90:560-564)
    r.v[i] = (T[4 + i] + dd) + da;              // :549
  }
  return r;
}

} // namespace greb
