// greb_device.h -- device-side arithmetic of the GREB hot path for gfx950 (CDNA4), wave64.
//
// Two arithmetic flavours of every stencil piece:
//   STRICT : the reference's expression trees verbatim (src/greb.f90:585-721, 756-913), IEEE
//            division, FMA contraction off  -> bit-identical to the reference/oracle.
//   FAST   : algebraically restructured "edge-flux" form.  With e_m = T(m+1)-T(m),
//            P+_m = w(m+1)*e_m, P-_m = w(m)*e_m, the reference's 7-point diffusion sum
//            (src/greb.f90:595-600) collapses to
//                S_j = 6(P+_j - P-_{j-1}) + 3(P+_{j+1} - P-_{j-2}) + (P+_{j+2} - P-_{j-3})
//            and both advection stencils (:802-806, :845-851) are linear in the same P's;
//            divisions become multiplications by precomputed reciprocals; FMA allowed.
//            Only the increments change (relative 1e-7); the state update keeps the reference's
//            two roundings X = (X + dX_diffuse) + dX_advec (src/greb.f90:549).
//
// Data is processed in "quads": 4 consecutive longitudes (one 16-byte LDS/HBM access).  A quad
// window t[12] holds longitudes 4(q-1) .. 4(q+1)+3 with periodic wrap; the quad's own points are
// t[4..7].
#pragma once
#include <hip/hip_runtime.h>

namespace greb {

constexpr int kMaxNy = 192;

// Per-row tables of diffusion/advection (src/greb.f90:578-582, 652-654, 749-753, 838-840),
// computed on the host in fp32 exactly as the reference does; one set per distinct kappa.
struct RowTables {
  float dif_ccy, adv_ccy;
  float dif_ccx[kMaxNy], adv_ccx[kMaxNy];   // full-row branch
  float dif_ccx2[kMaxNy], adv_ccx2[kMaxNy]; // sub-cycled branch
  int dif_time2[kMaxNy], adv_time2[kMaxNy];
  int subcycled[kMaxNy];                    // !(dxlat(k) > 2.5e5)
};

struct f4 {
  float v[4];
};

// LDS pointers carry their address space explicitly so every access is a ds_read/ds_write
// (a generic pointer costs a flat_load and, measured on MI355X, ~10x the latency).
typedef __attribute__((address_space(3))) float lfloat;
typedef float vfloat4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 ld4(const float* p) {
  const vfloat4 a = *reinterpret_cast<const vfloat4*>(p);
  return f4{{a.x, a.y, a.z, a.w}};
}
__device__ __forceinline__ f4 ld4(const lfloat* p) {
  const vfloat4 a = *(const __attribute__((address_space(3))) vfloat4*)p;
  return f4{{a.x, a.y, a.z, a.w}};
}
__device__ __forceinline__ void st4(float* p, const f4& a) {
  vfloat4 v; v.x = a.v[0]; v.y = a.v[1]; v.z = a.v[2]; v.w = a.v[3];
  *reinterpret_cast<vfloat4*>(p) = v;
}
// streaming (non-temporal) forms for data touched once per launch: they do not displace the L2 / MALL lines of
// the other stream
__device__ __forceinline__ f4 ld4_nt(const float* p) {
  const vfloat4 a = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(p));
  return f4{{a.x, a.y, a.z, a.w}};
}
__device__ __forceinline__ void st4_nt(float* p, const f4& a) {
  vfloat4 v; v.x = a.v[0]; v.y = a.v[1]; v.z = a.v[2]; v.w = a.v[3];
  __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(p));
}
__device__ __forceinline__ void st4(lfloat* p, const f4& a) {
  vfloat4 v; v.x = a.v[0]; v.y = a.v[1]; v.z = a.v[2]; v.w = a.v[3];
  *(__attribute__((address_space(3))) vfloat4*)p = v;
}

__device__ __forceinline__ f4 zero4() { return f4{{0.f, 0.f, 0.f, 0.f}}; }

// window of 12 longitudes around quad q of a row (periodic, src/greb.f90:594,602,610,...)
template <typename P>
__device__ __forceinline__ void load_window(P row, int q, int nq, float t[12]) {
  const int qm = (q == 0) ? nq - 1 : q - 1;
  const int qp = (q == nq - 1) ? 0 : q + 1;
  const f4 a = ld4(row + 4 * qm), b = ld4(row + 4 * q), c = ld4(row + 4 * qp);
#pragma unroll
  for (int i = 0; i < 4; ++i) { t[i] = a.v[i]; t[4 + i] = b.v[i]; t[8 + i] = c.v[i]; }
}

// min of three finite-or-NaN values as ONE v_min3_f32 (fminf would add a canonicalising v_max per operand)
__device__ __forceinline__ float min3f(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ float split_m(float u) { return u >= 0.0f ? u : 0.0f; } // src/greb.f90:203-205
__device__ __forceinline__ float split_p(float u) { return u >= 0.0f ? 0.0f : u; } // src/greb.f90:206-208

// ============================================================================================
// STRICT pieces (reference expression order)
// ============================================================================================

// (Tair, q) pair: the STRICT pieces below are written once for an element type E = float (one field) or v2 (both
// transported tracers at once).  With contraction off every v2 operation is the same IEEE operation on each half
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 only where an fma is written), so the packed form is bit-identical.
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fma_e(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ v2 fma_e(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float splat_e(float x, float) { return x; }
__device__ __forceinline__ v2 splat_e(float x, v2) { return v2{x, x}; }
// the sub-cycle clamp `where(dTxh <= -T1h) dTxh = -0.9*T1h` (:715, :907), per element
__device__ __forceinline__ float clamp_e(float d, float T) {
#pragma clang fp contract(off)
  return (d <= -T) ? -0.9f * T : d;
}
__device__ __forceinline__ v2 clamp_e(v2 d, v2 T) { return v2{clamp_e(d.x, T.x), clamp_e(d.y, T.y)}; }

// IEEE x/20 and x/3 in three instructions instead of the ~10 of the generic division sequence:
//   q = x*r,  e = fma(-q, c, x),  result = fma(e, r, q)      with r = RN(1/c)
// Checked EXHAUSTIVELY on the CPU against x/c for all 2^32 operands: identical for every x whose quotient is a
// normal number; the only differences are in the subnormal range (|x| < 4.8e-38 for c = 20) and the sign of an
// exact zero quotient (-0 comes out +0) -- neither changes any value the model compares or prints.
template <typename E>
__device__ __forceinline__ E div_by_const(E x, float c, float r) {
#pragma clang fp contract(off)
  const E q = x * r;
  const E e = fma_e(-q, splat_e(c, x), x);
  return fma_e(e, splat_e(r, x), q);
}
template <typename E> __device__ __forceinline__ E div20(E x) { return div_by_const(x, 20.f, 1.0f / 20.f); }
template <typename E> __device__ __forceinline__ E div3(E x) { return div_by_const(x, 3.f, 1.0f / 3.f); }

// 7-point sum of src/greb.f90:595-600 for window index c (4..7)
template <typename E>
__device__ __forceinline__ E dif_S_strict(const E* T, const E* w, int c) {
#pragma clang fp contract(off)
  return 10.f * (w[c - 1] * (T[c - 1] - T[c]) + w[c + 1] * (T[c + 1] - T[c]))
         + 4.f * (w[c - 2] * (T[c - 2] - T[c - 1]) + w[c - 1] * (T[c] - T[c - 1]))
         + 4.f * (w[c + 1] * (T[c] - T[c + 1]) + w[c + 2] * (T[c + 2] - T[c + 1]))
         + (w[c - 3] * (T[c - 3] - T[c - 2]) + w[c - 2] * (T[c - 1] - T[c - 2]))
         + (w[c + 2] * (T[c + 1] - T[c + 2]) + w[c + 3] * (T[c + 3] - T[c + 2]));
}

// one longitudinal diffusion increment: cc*S/20  (:595-600 with ccx, :659-664 with ccx2)
template <typename E>
__device__ __forceinline__ void dif_lon_strict(const E T[12], const E w[12], float cc, E d[4]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = div20(cc * dif_S_strict(T, w, 4 + i));
}

// src/greb.f90:802-806: full-row advection, window index c
template <typename E>
__device__ __forceinline__ void adv_lon_full_strict(const E T[12], const E w[12], const float u[4],
                                                    float ccx, E d[4]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    d[i] = div3(ccx * (-split_m(u[i]) * (w[c - 1] * (T[c] - T[c - 1]) + w[c - 2] * (T[c] - T[c - 2]))
                  + split_p(u[i]) * (w[c + 1] * (T[c] - T[c + 1]) + w[c + 2] * (T[c] - T[c + 2]))));
  }
}

// src/greb.f90:845-851 (+ the :881 index bug for j = xdim-2): sub-cycled advection increment of
// the point at window index c; bug = this is longitude xdim-2 (1-based), whose "+2" neighbour
// aliases the "+1" one (jp2 = xdim-1)
template <typename E>
__device__ __forceinline__ E adv_lon_sub_point_strict(const E* T, const E* w, float u, float ccx2,
                                                      int c, bool bug) {
#pragma clang fp contract(off)
  const int p1 = c + 1, p2 = bug ? c + 1 : c + 2, p3 = c + 3;
  return div20(ccx2 * (-split_m(u) * (10.f * w[c - 1] * (T[c] - T[c - 1])
                                + 4.f * w[c - 2] * (T[c - 1] - T[c - 2])
                                + w[c - 3] * (T[c - 2] - T[c - 3]))
                 + split_p(u) * (10.f * w[p1] * (T[c] - T[p1])
                                 + 4.f * w[p2] * (T[p1] - T[p2])
                                 + w[p3] * (T[p2] - T[p3]))));
}
template <typename E>
__device__ __forceinline__ void adv_lon_sub_strict(const E T[12], const E w[12], const float u[4],
                                                   float ccx2, bool last_quad, E d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = adv_lon_sub_point_strict(T, w, u[i], ccx2, 4 + i, last_quad && i == 1);
}

// clamp + accumulate of the sub-cycle loops (:715-716, :907-908)
template <typename E>
__device__ __forceinline__ void clamp_add(E T1h[4], E d[4]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d[i] = clamp_e(d[i], T1h[i]);
    T1h[i] = T1h[i] + d[i];
  }
}

// latitudinal diffusion (:585-590) of one point.  tm/tp, wm/wp: rows k-1 / k+1 (ignored where absent)
template <typename E>
__device__ __forceinline__ E dif_lat_point_strict(E t0, E tm, E tp, E wm, E wp, float ccy, int k, int ny) {
#pragma clang fp contract(off)
  if (k >= 1 && k <= ny - 2) return ccy * (wm * (tm - t0) + wp * (tp - t0));
  if (k == 0) return ccy * wp * (-t0 + tp);
  return ccy * wm * (tm - t0);
}
template <typename Q, typename E>
__device__ __forceinline__ void dif_lat_strict(const Q& T0, const Q& Tm, const Q& Tp, const Q& wm,
                                               const Q& wp, float ccy, int k, int ny, E d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = dif_lat_point_strict<E>(T0.v[i], Tm.v[i], Tp.v[i], wm.v[i], wp.v[i], ccy, k, ny);
}

// latitudinal advection (:756-795) of one point.  T/w at rows k-2,k-1,k+1,k+2
template <typename E>
__device__ __forceinline__ E adv_lat_point_strict(E t0, E tm2, E tm1, E tp1, E tp2, E wm2, E wm1, E wp1, E wp2, float v,
                                                  float ccy, int k, int ny) {
#pragma clang fp contract(off)
  const float vm = split_m(v), vp = split_p(v);
  const E dm1 = wm1 * (t0 - tm1), dm2 = wm2 * (t0 - tm2);
  const E dp1 = wp1 * (t0 - tp1), dp2 = wp2 * (t0 - tp2);
  if (k == 0) return div3(ccy * (vp * (dp1 + dp2)));                              // :759-761
  if (k == 1) return ccy * (-vm * (dm1) + div3(vp * (dp1 + dp2)));               // :766-769
  if (k <= ny - 3) return div3(ccy * (-vm * (dm1 + dm2) + vp * (dp1 + dp2)));    // :774-778
  if (k == ny - 2) return ccy * (div3(-vm * (dm1 + dm2)) + vp * (dp1));          // :784-787
  return div3(ccy * (-vm * (dm1 + dm2)));                                        // :792-794
}
template <typename Q, typename E>
__device__ __forceinline__ void adv_lat_strict(const Q& T0, const Q& Tm2, const Q& Tm1, const Q& Tp1,
                                               const Q& Tp2, const Q& wm2, const Q& wm1, const Q& wp1,
                                               const Q& wp2, const float v[4], float ccy, int k, int ny,
                                               E d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    d[i] = adv_lat_point_strict<E>(T0.v[i], Tm2.v[i], Tm1.v[i], Tp1.v[i], Tp2.v[i], wm2.v[i], wm1.v[i], wp1.v[i], wp2.v[i],
                                   v[i], ccy, k, ny);
}

// ============================================================================================
// FAST pieces (edge-flux form)
// ============================================================================================
struct Flux {
  float Pp[10]; // Pp[m] = w[m+1]*(T[m+1]-T[m]), valid m = 4..9
  float Pm[10]; // Pm[m] = w[m]  *(T[m+1]-T[m]), valid m = 1..6
};

__device__ __forceinline__ void make_flux(const float T[12], const float w[12], Flux& f) {
#pragma unroll
  for (int m = 1; m <= 9; ++m) {
    const float e = T[m + 1] - T[m];
    if (m >= 4) f.Pp[m] = w[m + 1] * e;
    if (m <= 6) f.Pm[m] = w[m] * e;
  }
}

// cs = cc/20
__device__ __forceinline__ void dif_lon_fast(const Flux& f, float cs, float d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    const float a = f.Pp[c] - f.Pm[c - 1], b = f.Pp[c + 1] - f.Pm[c - 2], g = f.Pp[c + 2] - f.Pm[c - 3];
    d[i] = cs * (6.f * a + (3.f * b + g));
  }
}

// cs = ccx/3
__device__ __forceinline__ void adv_lon_full_fast(const Flux& f, const float T[12], const float w[12],
                                                  const float u[4], float cs, float d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    const float em2 = w[c - 2] * (T[c] - T[c - 2]), ep2 = w[c + 2] * (T[c] - T[c + 2]);
    d[i] = cs * (split_p(u[i]) * (ep2 - f.Pp[c]) - split_m(u[i]) * (f.Pm[c - 1] + em2));
  }
}

// cs = ccx2/20
__device__ __forceinline__ void adv_lon_sub_fast(const Flux& f, const float T[12], const float w[12],
                                                 const float u[4], float cs, bool last_quad, float d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    const float am = 10.f * f.Pm[c - 1] + (4.f * f.Pm[c - 2] + f.Pm[c - 3]);
    float ap = 10.f * f.Pp[c] + (4.f * f.Pp[c + 1] + f.Pp[c + 2]);
    if (last_quad && i == 1) // :881 index bug: 4*w(jp1)*(T(jp1)-T(jp1)) + w(jp3)*(T(jp1)-T(jp3))
      ap = 10.f * f.Pp[c] - w[c + 3] * (T[c + 1] - T[c + 3]);
    d[i] = cs * (-split_p(u[i]) * ap - split_m(u[i]) * am);
  }
}

__device__ __forceinline__ void clamp_add_fast(float T1h[4], float d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d[i] = (d[i] <= -T1h[i]) ? -0.9f * T1h[i] : d[i];
    T1h[i] = T1h[i] + d[i];
  }
}

// Latitudinal part, both operators at once.  Missing rows are passed with w = 0.
//   dif: ccy*(G-1 + G+1),  G(n) = w(k+n)*(T(k+n)-T0)
//   adv: am*(-vm)*(D-1 + D-2) + ap*vp*(D+1 + D+2),  D(n) = w(k+n)*(T0-T(k+n)),
//        am/ap = ccy/3 except am = ccy at k=1 and ap = ccy at k=ny-2 (:766-769,:784-787)
__device__ __forceinline__ void lat_fast(const f4& T0, const f4& Tm2, const f4& Tm1, const f4& Tp1,
                                         const f4& Tp2, const f4& wm2, const f4& wm1, const f4& wp1,
                                         const f4& wp2, const float v[4], float ccy_dif, float am, float ap,
                                         float ddif[4], float dadv[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t0 = T0.v[i];
    const float gm1 = wm1.v[i] * (Tm1.v[i] - t0), gp1 = wp1.v[i] * (Tp1.v[i] - t0);
    const float dm2 = wm2.v[i] * (t0 - Tm2.v[i]), dp2 = wp2.v[i] * (t0 - Tp2.v[i]);
    ddif[i] = ccy_dif * (gm1 + gp1);
    dadv[i] = ap * split_p(v[i]) * (dp2 - gp1) - am * split_m(v[i]) * (dm2 - gm1);
  }
}


// ---- FAST pieces with pre-multiplied winds (the fused engine stages them in LDS once per model
// step):  um = c*max(u,0), up = c*min(u,0) with c = ccx/3 (full rows) or ccx2/20 (sub-cycled rows);
// vm = am*max(v,0), vp = ap*min(v,0) with am/ap of adv_lat_coef().
__device__ __forceinline__ void adv_lon_full_pm(const Flux& f, const float T[12], const float w[12],
                                                const float um[4], const float up[4], float d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    const float em2 = w[c - 2] * (T[c] - T[c - 2]), ep2 = w[c + 2] * (T[c] - T[c + 2]);
    d[i] = up[i] * (ep2 - f.Pp[c]) - um[i] * (f.Pm[c - 1] + em2);
  }
}
__device__ __forceinline__ void adv_lon_sub_pm(const Flux& f, const float T[12], const float w[12],
                                               const float um[4], const float up[4], bool last_quad, float d[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = 4 + i;
    const float am = 10.f * f.Pm[c - 1] + (4.f * f.Pm[c - 2] + f.Pm[c - 3]);
    float ap = 10.f * f.Pp[c] + (4.f * f.Pp[c + 1] + f.Pp[c + 2]);
    if (last_quad && i == 1) ap = 10.f * f.Pp[c] - w[c + 3] * (T[c + 1] - T[c + 3]); // :881
    d[i] = -up[i] * ap - um[i] * am;
  }
}
__device__ __forceinline__ void lat_pm(const f4& T0, const f4& Tm2, const f4& Tm1, const f4& Tp1, const f4& Tp2,
                                       const f4& wm2, const f4& wm1, const f4& wp1, const f4& wp2,
                                       const float vm[4], const float vp[4], float ccy_dif, float ddif[4],
                                       float dadv[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t0 = T0.v[i];
    const float gm1 = wm1.v[i] * (Tm1.v[i] - t0), gp1 = wp1.v[i] * (Tp1.v[i] - t0);
    const float dm2 = wm2.v[i] * (t0 - Tm2.v[i]), dp2 = wp2.v[i] * (t0 - Tp2.v[i]);
    ddif[i] = ccy_dif * (gm1 + gp1);
    dadv[i] = vp[i] * (dp2 - gp1) - vm[i] * (dm2 - gm1);
  }
}

// Sum over the 64 lanes of a wavefront by shuffles (ds_bpermute butterflies); every lane gets the total.
// Used for the diagnostic global mean (src/greb.f90:954) in FAST arithmetic; STRICT keeps the reference's
// sequential order, which is what flang's sum() lowers to.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ============================================================================================
// Point physics (src/greb.f90:367-525).  One column of the model; everything fp32.
// STRICT keeps the reference's operation order (x**4 = ((x*x)*x)*x as flang -O2 lowers it); the
// only non-bit-exact parts are OCML expf/logf vs glibc (<= 1-2 ulp on fluxes).
// ============================================================================================
// Math policy of the point physics.  EXACT: IEEE division / OCML logf, expf, sqrtf (STRICT engine,
// bit-exact where no transcendental is involved).  Otherwise the hardware approximations
// v_rcp/v_log/v_exp/v_sqrt (<= 1-2 ulp): the fluxes they feed enter the state scaled by
// dt/cap ~ 1e-3..1e-1 K, far below the 1e-4 K tolerance, and they cut the per-point instruction
// count ~3x (the physics phase is VALU-bound: ~14 divisions, 4 transcendentals per point).
// (HIP's __fdividef is a plain IEEE division unless the whole file is built with fast-math: ~10 instructions.  The
// FAST policy wants ONE v_rcp_f32 + one multiply; 19 divisions per point made up 44 % of the point-physics phase.)
template <bool EXACT> __device__ __forceinline__ float fdiv(float a, float b) { return EXACT ? a / b : a * __builtin_amdgcn_rcpf(b); }
// (HIP's __logf / __expf are not bare hardware instructions either -- OCML adds denormal scaling and an
// extended-precision ln2 / log2e product, ~11 instructions per call, 16 calls per quad.  Replacing them by
// v_log_f32 * ln2 and v_exp_f32(x * log2e) keeps parity (Tsurf 1.42e-5 K RMS) but measured SLOWER, A/B in one gpurun
// call: point-physics phase 26 600 -> 28 300 cycles, sub-step 4 933 -> 4 990 -- the phase waits on memory, not on
// these instructions, and the shorter code moved the compiler's load/store scheduling the wrong way.  Not taken.)
template <bool EXACT> __device__ __forceinline__ float flog(float x) { return EXACT ? logf(x) : __logf(x); }
template <bool EXACT> __device__ __forceinline__ float fexp(float x) { return EXACT ? expf(x) : __expf(x); }
template <bool EXACT> __device__ __forceinline__ float fsqrt(float x) { return EXACT ? sqrtf(x) : __builtin_amdgcn_sqrtf(x); }

// experiment switches: mirror of GREB_X_* (include/greb_engine.h; equality asserted in greb_engine.cpp)
constexpr unsigned kXNoIce = 1u << 0, kXNoHydro = 1u << 1, kXNoDeepOcean = 1u << 2, kXLwLinear = 1u << 3,
                   kXNoCirc = 1u << 4, kXNoQTransport = 1u << 5, kXQDiffOnly = 1u << 6, kXSstPlus1 = 1u << 7;

struct Phys { // namelist physics_par + derived capacities, broadcast to the kernel
  float sig, ct_sens, da_ice, a_no_ice, a_cloud, Tl_ice1, Tl_ice2, To_ice1, To_ice2;
  float co_turb, ce, cq_latent, cq_rain, z_air, r_qviwv, rho_air;
  float p_emi[10];
  float cap_ocean, cap_land, cap_air;
  float dt; // float(dt)
};

__device__ __forceinline__ float pow4_ref(float x) {
#pragma clang fp contract(off)
  return x * x * x * x;
}

// a4 SWradiation :380-401 -> albedo, sw
template <bool EXACT = true>
__device__ __forceinline__ void sw_radiation(const Phys& P, float Ts, float z_topo, float glacier, float cld,
                                             float sw_solar, float& albedo, float& sw, unsigned xsw = 0) {
#pragma clang fp contract(off)
  const float a_atmos = cld * P.a_cloud;
  float a_surf = 0.f;
  if (z_topo >= 0.f) {
    if (Ts <= P.Tl_ice1) a_surf = P.a_no_ice + P.da_ice;
    if (Ts >= P.Tl_ice2) a_surf = P.a_no_ice;
    if (Ts > P.Tl_ice1 && Ts < P.Tl_ice2)
      a_surf = P.a_no_ice + P.da_ice * (1.f - fdiv<EXACT>(Ts - P.Tl_ice1, P.Tl_ice2 - P.Tl_ice1));
  } else {
    if (Ts <= P.To_ice1) a_surf = P.a_no_ice + P.da_ice;
    if (Ts >= P.To_ice2) a_surf = P.a_no_ice;
    if (Ts > P.To_ice1 && Ts < P.To_ice2)
      a_surf = P.a_no_ice + P.da_ice * (1.f - fdiv<EXACT>(Ts - P.To_ice1, P.To_ice2 - P.To_ice1));
  }
  if (glacier > 0.5f) a_surf = P.a_no_ice + P.da_ice;
  if (xsw & kXNoIce) a_surf = P.a_no_ice; // greb.original.model.f90:394
  albedo = a_surf + a_atmos - a_surf * a_atmos;
  sw = sw_solar * (1.f - albedo);
}

// a5 LWradiation :420-432.  ez = exp(-z_topo/z_air) (== wz_air, :201), dTrad = -0.16*Tclim-5 (:176)
template <bool EXACT = true>
__device__ __forceinline__ void lw_radiation(const Phys& P, float Ts, float Ta, float q, float co2, float ez,
                                             float cld, float tclim, float& LWsurf, float& LWair_down,
                                             float& em, unsigned xsw = 0, float qclim = 0.f) {
#pragma clang fp contract(off)
  const float e_co2 = ez * co2;
  float e_vapor = ez * P.r_qviwv * q;
  if (xsw & kXLwLinear) e_vapor = ez * P.r_qviwv * qclim; // greb.original.model.f90:423
  float e = P.p_emi[3] * flog<EXACT>(P.p_emi[0] * e_co2 + P.p_emi[1] * e_vapor + P.p_emi[2]) + P.p_emi[6]
            + P.p_emi[4] * flog<EXACT>(P.p_emi[0] * e_co2 + P.p_emi[2])
            + P.p_emi[5] * flog<EXACT>(P.p_emi[1] * e_vapor + P.p_emi[2]);
  e = fdiv<EXACT>(P.p_emi[7] - cld, P.p_emi[8]) * (e - P.p_emi[9]) + P.p_emi[9];
  if (xsw & kXLwLinear) e = e + 0.022f / (0.15f * 24.f) * P.r_qviwv * (q - qclim); // greb.original.model.f90:430
  em = e;
  LWsurf = -P.sig * pow4_ref(Ts);
  const float dTrad = -0.16f * tclim - 5.f;
  LWair_down = -e * P.sig * pow4_ref(Ta + dTrad);
}

// a6 hydro :452-467
template <bool EXACT = true>
__device__ __forceinline__ void hydro(const Phys& P, float Ts, float q, float u, float v, float z_topo,
                                      float ez, float swet, float& Qlat, float& Qlat_air, float& dq_eva,
                                      float& dq_rain, unsigned xsw = 0) {
#pragma clang fp contract(off)
  if (xsw & kXNoHydro) { Qlat = Qlat_air = dq_eva = dq_rain = 0.f; return; } // greb.original.model.f90:452-453
  float abswind = fsqrt<EXACT>(u * u + v * v);
  if (z_topo > 0.f) abswind = fsqrt<EXACT>(abswind * abswind + 2.0f * 2.0f);
  if (z_topo < 0.f) abswind = fsqrt<EXACT>(abswind * abswind + 3.0f * 3.0f);
  float qs = 3.75e-3f * fexp<EXACT>(fdiv<EXACT>(17.08085f * (Ts - 273.15f), Ts - 273.15f + 234.175f));
  qs = qs * ez;
  Qlat = (q - qs) * abswind * P.cq_latent * P.rho_air * P.ce * swet;
  dq_eva = fdiv<EXACT>(fdiv<EXACT>(-Qlat, P.cq_latent), P.r_qviwv);
  dq_rain = P.cq_rain * q;
  Qlat_air = -dq_rain * P.cq_latent * P.r_qviwv;
}

// a7 deep_ocean :505-523
template <bool EXACT = true>
__device__ __forceinline__ void deep_ocean(const Phys& P, float Ts, float To, float z_topo, float mld,
                                           float mld_prev, float z_ocean, float& dT_ocean, float& dTo,
                                           unsigned xsw = 0) {
#pragma clang fp contract(off)
  if (xsw & kXNoDeepOcean) { dT_ocean = dTo = 0.f; return; } // greb.original.model.f90:513-515
  float a = 0.f, b = 0.f;
  const float dmld = mld - mld_prev;
  if (z_topo < 0.f && Ts >= P.To_ice2 && dmld < 0.f) a = fdiv<EXACT>(-dmld, z_ocean - mld) * (Ts - To);
  if (z_topo < 0.f && Ts >= P.To_ice2 && dmld > 0.f) b = fdiv<EXACT>(dmld, mld) * (To - Ts);
  a = 0.5f * a; b = 0.5f * b;
  const float Tx = P.To_ice2 > Ts ? P.To_ice2 : Ts;
  a = a + fdiv<EXACT>(P.dt * P.co_turb * (Tx - To), P.cap_ocean * (z_ocean - mld));
  b = b + fdiv<EXACT>(P.dt * P.co_turb * (To - Tx), P.cap_ocean * mld);
  dTo = a; dT_ocean = b;
}

// a8 seaice :483-490 -> new cap_surf
template <bool EXACT = true>
__device__ __forceinline__ float seaice(const Phys& P, float Ts, float z_topo, float glacier, float mld,
                                        float cap_surf, unsigned xsw = 0) {
#pragma clang fp contract(off)
  if (z_topo < 0.f) {
    if (Ts <= P.To_ice1) cap_surf = P.cap_land;
    if (Ts >= P.To_ice2) cap_surf = P.cap_ocean * mld;
    if (Ts > P.To_ice1 && Ts < P.To_ice2)
      cap_surf = P.cap_land + fdiv<EXACT>(P.cap_ocean * mld - P.cap_land, P.To_ice2 - P.To_ice1) * (Ts - P.To_ice1);
  }
  if (xsw & kXNoIce) { // greb.original.model.f90:492-495
    if (z_topo > 0.f) cap_surf = P.cap_land;
    if (z_topo < 0.f) cap_surf = P.cap_ocean * mld;
  }
  if (glacier > 0.5f) cap_surf = P.cap_land;
  return cap_surf;
}

} // namespace greb
