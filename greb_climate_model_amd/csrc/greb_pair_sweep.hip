// greb_pair_sweep.hip -- FAST circulation sub-step of the any-grid engine on (Tair, q) PAIRS (384-wide grids).
//
// The scalar any-grid kernel (greb_kernels.hip: sweep_kernel) treats the two transported tracers of a member as two
// batch items.  At 384x192 every row is sub-cycled and the rows within ~18 degrees of the poles iterate (up to 225
// Jacobi sweeps per diffusion call), so the launch is instruction-bound on those chains.  Both tracers see the same
// winds, row constants and sweep counts: carrying them as v2 pairs makes every arithmetic instruction a packed
// v_pk_*_f32 (greb_pair.h, as in the fused 96x48 engine) -- one chain, one address stream, one wind sign split for
// the two of them.
//   X2 / Xnew2 : [member][ny][nx][{Tair,q}]   W2p : [ny][nx][{wz_air,wz_vapor}]   winds: [ny][nx], shared
// One workgroup = one member x one latitude band.  LDS rows are [half][quad][4] (greb_pair.h) so the dwordx4
// reads of neighbouring lanes do not collide.  Iterating rows live in registers: lane l owns longitudes 6l..6l+5,
// halos by v_mov_b32_dpp wave_ror/rol:1 (the rotate is the periodic boundary).
#include <cstdlib>

#include "greb_kernels.h"
#include "greb_pair.h"
#include "greb_stencil.h"

namespace greb {
namespace {

constexpr int kPairNx = 384, kPairNq = 96, kPairP = 6;
static_assert(kPairP == 6, "the chain's min3 reduction is written for six points per lane");
constexpr int kSplitMinSweeps = 96; // a band's longest diffusion chain is split by tracer from this length on
constexpr int kPHalf = 4 * kPairNq;  // floats per half-row
constexpr int kPRow = 2 * kPHalf;    // floats per LDS row (768)

// float offset of the point pair i = (longitudes 2i, 2i+1) inside a row
__device__ __forceinline__ int ppair_off(int i) { return (i & 1) * kPHalf + (i >> 1) * 4; }
__device__ __forceinline__ q8 ld8g(const lfloat* row, int q) {
  const vfloat4 a = *(const __attribute__((address_space(3))) vfloat4*)(row + 4 * q);
  const vfloat4 b = *(const __attribute__((address_space(3))) vfloat4*)(row + kPHalf + 4 * q);
  q8 r;
  r.v[0] = v2{a.x, a.y}; r.v[1] = v2{a.z, a.w}; r.v[2] = v2{b.x, b.y}; r.v[3] = v2{b.z, b.w};
  return r;
}
__device__ __forceinline__ v2 dpp_prev(v2 x) { return v2{wave_from_prev(x.x), wave_from_prev(x.y)}; }
__device__ __forceinline__ v2 dpp_next(v2 x) { return v2{wave_from_next(x.x), wave_from_next(x.y)}; }

// one iterating row, both tracers: time2 sweeps of the diffusion and of the advection stencil from the same start,
// then the latitudinal terms and the update (src/greb.f90:651-719, :837-911, :585-590, :756-795, :721, :913, :549)
__device__ void pair_chain_row(const lfloat* sT, const lfloat* sW, const lfloat* sU, const lfloat* sV, int r0, int k0,
                               int k, int ny, const RowK& rk, int lane, float* __restrict__ out_row /* global [nx][2] */) {
  constexpr int P = kPairP, W = P + 6;
  const lfloat* Trow = sT + (k - r0) * kPRow;
  const lfloat* Wrow = sW + (k - r0) * kPRow;
  v2 T0[W], w[W];
  // load by single points (ds_read_b64): 12 window points, periodic
#pragma unroll
  for (int i = 0; i < W; ++i) {
    int x = P * lane - 3 + i;
    x = x < 0 ? x + kPairNx : (x >= kPairNx ? x - kPairNx : x);
    const int o = ppair_off(x >> 1) + (x & 1) * 2;
    T0[i] = *(const __attribute__((address_space(3))) v2*)(Trow + o);
    w[i] = *(const __attribute__((address_space(3))) v2*)(Wrow + o);
  }
  float us[P]; // zonal wind scaled by adv_ccx2/20 (staging), so the sign split is one max/min
#pragma unroll
  for (int i = 0; i < P; ++i) us[i] = sU[(k - k0) * kPairNx + P * lane + i];
  const bool bug_lane = lane == 63; // its point P-3 is longitude xdim-2 (1-based), src/greb.f90:881
  const int time2[2] = {__builtin_amdgcn_readfirstlane(rk.dif_time2), __builtin_amdgcn_readfirstlane(rk.adv_time2)};
  if (time2[0] >= kChainPrioSweeps) __builtin_amdgcn_s_setprio(3); // greb_stencil.h: the long chains issue first
  const float cs = rk.dif_cc * 0.05f;
  v2 Th[2][P];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    // The weights, the row constant and the (pre-scaled) wind do not change during the chain: the increment of
    // point c is a fixed linear form in the six differences e[m] = T[m+1] - T[m] around it, per tracer -- the same
    // coefficient form as the scalar any-grid kernel (greb_stencil.h: chain_lon_regs), on (Tair, q) pairs.
    v2 K[P][6];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int c = 3 + i;
      if (which) { // -neg*(10 Pp[c] + 4 Pp[c+1] + Pp[c+2]) - pos*(10 Pm[c-1] + 4 Pm[c-2] + Pm[c-3]), :845-851
        float pos, neg;
        split_sign(us[i], pos, neg);
        K[i][0] = -pos * w[c - 3]; K[i][1] = (-4.f * pos) * w[c - 2]; K[i][2] = (-10.f * pos) * w[c - 1];
        K[i][3] = (-10.f * neg) * w[c + 1]; K[i][4] = (-4.f * neg) * w[c + 2]; K[i][5] = -neg * w[c + 3];
        if (i == P - 3 && bug_lane) { K[i][4] = -neg * w[c + 3]; K[i][5] = -neg * w[c + 3]; } // :881
      } else { // cs*(6(Pp[c] - Pm[c-1]) + 3(Pp[c+1] - Pm[c-2]) + (Pp[c+2] - Pm[c-3])), :595-600
        K[i][0] = -cs * w[c - 3]; K[i][1] = (-3.f * cs) * w[c - 2]; K[i][2] = (-6.f * cs) * w[c - 1];
        K[i][3] = (6.f * cs) * w[c + 1]; K[i][4] = (3.f * cs) * w[c + 2]; K[i][5] = cs * w[c + 3];
      }
    }
    v2 T[W];
#pragma unroll
    for (int i = 0; i < W; ++i) T[i] = T0[i];
    for (int tt = 0; tt < time2[which]; ++tt) {
      if (tt > 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          T[i] = dpp_prev(T[P + i]);
          T[P + 3 + i] = dpp_next(T[3 + i]);
        }
      }
      v2 e[W - 1];
#pragma unroll
      for (int m = 0; m < W - 1; ++m) e[m] = pk_sub(T[m + 1], T[m]);
      v2 Tn[P], dv[P];
#pragma unroll
      for (int i = 0; i < P; ++i) {
        v2 d = K[i][0] * e[i];
#pragma unroll
        for (int m = 1; m < 6; ++m) d = __builtin_elementwise_fma(K[i][m], e[i + m], d);
        dv[i] = d;
        Tn[i] = T[3 + i] + d;
      }
      float mn = min3f(Tn[0].x, Tn[0].y, Tn[1].x);
      mn = min3f(mn, Tn[1].y, Tn[2].x); mn = min3f(mn, Tn[2].y, Tn[3].x); mn = min3f(mn, Tn[3].y, Tn[4].x);
      mn = min3f(mn, Tn[4].y, Tn[5].x); mn = min3f(mn, Tn[5].y, Tn[5].y);
      if (__builtin_expect(!(mn > 0.f), 0)) { // the clamp (:715 / :907), decided per component only where needed
#pragma unroll
        for (int i = 0; i < P; ++i) {
          const int c = 3 + i;
          v2 d = dv[i];
          d.x = (d.x <= -T[c].x) ? -0.9f * T[c].x : d.x;
          d.y = (d.y <= -T[c].y) ? -0.9f * T[c].y : d.y;
          Tn[i] = T[c] + d;
        }
      }
#pragma unroll
      for (int i = 0; i < P; ++i) T[3 + i] = Tn[i];
    }
#pragma unroll
    for (int i = 0; i < P; ++i) Th[which][i] = T[3 + i];
  }
  // latitudinal terms: rows k-2 .. k+2 at the own longitudes; rows outside the grid contribute nothing
  if (time2[0] >= kChainPrioSweeps) __builtin_amdgcn_s_setprio(0);
  const float am = (k == 1) ? 3.f : 1.f, ap = (k == ny - 2) ? 3.f : 1.f; // :766-769, :784-787 (v is scaled by ccy/3)
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int x = P * lane + i;
    const int o = ppair_off(x >> 1) + (x & 1) * 2;
    const v2 own = T0[3 + i];
    auto rowv = [&](const lfloat* base, int kk) { return *(const __attribute__((address_space(3))) v2*)(base + (kk - r0) * kPRow + o); };
    const v2 z = v2{0.f, 0.f};
    const v2 Tm1 = k >= 1 ? rowv(sT, k - 1) : own, Tp1 = k <= ny - 2 ? rowv(sT, k + 1) : own;
    const v2 Tm2 = k >= 2 ? rowv(sT, k - 2) : own, Tp2 = k <= ny - 3 ? rowv(sT, k + 2) : own;
    const v2 Wm1 = k >= 1 ? rowv(sW, k - 1) : z, Wp1 = k <= ny - 2 ? rowv(sW, k + 1) : z;
    const v2 Wm2 = k >= 2 ? rowv(sW, k - 2) : z, Wp2 = k <= ny - 3 ? rowv(sW, k + 2) : z;
    float vpos, vneg;
    split_sign(sV[(k - k0) * kPairNx + x], vpos, vneg);
    const v2 gm1 = Wm1 * pk_sub(Tm1, own), gp1 = Wp1 * pk_sub(Tp1, own);
    const v2 dm2 = Wm2 * pk_sub(own, Tm2), dp2 = Wp2 * pk_sub(own, Tp2);
    const v2 ddy = rk.dif_ccy * (gm1 + gp1);
    const v2 day = (ap * vneg) * (dp2 - gp1) - (am * vpos) * (dm2 - gm1);
    const v2 dd = w[3 + i] * ((Th[0][i] - own) + ddy); // :718, :721
    const v2 da = (Th[1][i] - own) + day;              // :910, :913
    const v2 xn = (own + dd) + da;                     // :549
    *(float2*)(out_row + 2 * x) = make_float2(xn.x, xn.y);
  }
}

// ONE tracer (C = 0: Tair, 1: q) of an iterating row, scalar arithmetic: the longest chain of a latitude band is
// taken by two waves, one per tracer.  A chain's latency is its instruction count (a lone wave issues one instruction
// per ~5 cycles, packed or not) and the pair form's sweep is ~1.7x the scalar one's (both components' selects and
// halo moves, packed instructions contending with whatever shares the SIMD): 225 sweeps take ~65 us as pairs and
// ~38 us per tracer side by side.  Same coefficient form, same operation order per tracer as pair_chain_row.
template <int C>
__device__ void pair_chain_row_comp(const lfloat* sT, const lfloat* sW, const lfloat* sU, const lfloat* sV, int r0, int k0,
                                    int k, int ny, const RowK& rk, int lane, float* __restrict__ out_row /* global [nx][2] */) {
  constexpr int P = kPairP, W = P + 6;
  auto comp = [](v2 a) { return C == 0 ? a.x : a.y; };
  const lfloat* Trow = sT + (k - r0) * kPRow;
  const lfloat* Wrow = sW + (k - r0) * kPRow;
  float T0[W], w[W];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    int x = P * lane - 3 + i;
    x = x < 0 ? x + kPairNx : (x >= kPairNx ? x - kPairNx : x);
    const int o = ppair_off(x >> 1) + (x & 1) * 2 + C;
    T0[i] = Trow[o];
    w[i] = Wrow[o];
  }
  float us[P];
#pragma unroll
  for (int i = 0; i < P; ++i) us[i] = sU[(k - k0) * kPairNx + P * lane + i];
  const bool bug_lane = lane == 63; // :881
  const int time2[2] = {__builtin_amdgcn_readfirstlane(rk.dif_time2), __builtin_amdgcn_readfirstlane(rk.adv_time2)};
  if (time2[0] >= kChainPrioSweeps) __builtin_amdgcn_s_setprio(3); // greb_stencil.h: the long chains issue first
  const float cs = rk.dif_cc * 0.05f;
  float Th[2][P];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float K[P][6];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int c = 3 + i;
      if (which) {
        float pos, neg;
        split_sign(us[i], pos, neg);
        K[i][0] = -pos * w[c - 3]; K[i][1] = (-4.f * pos) * w[c - 2]; K[i][2] = (-10.f * pos) * w[c - 1];
        K[i][3] = (-10.f * neg) * w[c + 1]; K[i][4] = (-4.f * neg) * w[c + 2]; K[i][5] = -neg * w[c + 3];
        if (i == P - 3 && bug_lane) { K[i][4] = -neg * w[c + 3]; K[i][5] = -neg * w[c + 3]; }
      } else {
        K[i][0] = -cs * w[c - 3]; K[i][1] = (-3.f * cs) * w[c - 2]; K[i][2] = (-6.f * cs) * w[c - 1];
        K[i][3] = (6.f * cs) * w[c + 1]; K[i][4] = (3.f * cs) * w[c + 2]; K[i][5] = cs * w[c + 3];
      }
    }
    static_assert(P == 6, "chain_run6 (greb_chain6.h) is written for 6 points per lane");
#pragma unroll
    for (int i = 0; i < P; ++i) Th[which][i] = T0[3 + i];
    chain_run6<false>(Th[which], chain_pack(K), time2[which]); // the sweeps, greb_chain6.h
  }
  if (time2[0] >= kChainPrioSweeps) __builtin_amdgcn_s_setprio(0);
  const float am = (k == 1) ? 3.f : 1.f, ap = (k == ny - 2) ? 3.f : 1.f; // :766-769, :784-787 (v is scaled by ccy/3)
#pragma unroll
  for (int i = 0; i < P; ++i) {
    const int x = P * lane + i;
    const int o = ppair_off(x >> 1) + (x & 1) * 2 + C;
    const float own = T0[3 + i];
    auto rowv = [&](const lfloat* base, int kk) { return base[(kk - r0) * kPRow + o]; };
    const float Tm1 = k >= 1 ? rowv(sT, k - 1) : own, Tp1 = k <= ny - 2 ? rowv(sT, k + 1) : own;
    const float Tm2 = k >= 2 ? rowv(sT, k - 2) : own, Tp2 = k <= ny - 3 ? rowv(sT, k + 2) : own;
    const float Wm1 = k >= 1 ? rowv(sW, k - 1) : 0.f, Wp1 = k <= ny - 2 ? rowv(sW, k + 1) : 0.f;
    const float Wm2 = k >= 2 ? rowv(sW, k - 2) : 0.f, Wp2 = k <= ny - 3 ? rowv(sW, k + 2) : 0.f;
    float vpos, vneg;
    split_sign(sV[(k - k0) * kPairNx + x], vpos, vneg);
    const float gm1 = Wm1 * (Tm1 - own), gp1 = Wp1 * (Tp1 - own);
    const float dm2 = Wm2 * (own - Tm2), dp2 = Wp2 * (own - Tp2);
    const float ddy = rk.dif_ccy * (gm1 + gp1);
    const float day = (ap * vneg) * (dp2 - gp1) - (am * vpos) * (dm2 - gm1);
    const float dd = w[3 + i] * ((Th[0][i] - own) + ddy); // :718, :721
    const float da = (Th[1][i] - own) + day;              // :910, :913
    out_row[2 * x + C] = (own + dd) + da;                 // :549
  }
}

__global__ __launch_bounds__(256) void sweep_pair_kernel(const float* __restrict__ X2, const float* __restrict__ W2p,
                                                         const float* __restrict__ ug, const float* __restrict__ vg,
                                                         float* __restrict__ Xnew2, const RowTables* __restrict__ tabp,
                                                         const int* __restrict__ tab_index, int ny, int rows_per_band) {
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  const int m = blockIdx.x;
  const RowTables& tab = tabp[tab_index ? tab_index[m] : 0];
  const int nbands = gridDim.y; // poles first: the polar bands carry the long chains
  const int band = (blockIdx.y & 1) ? nbands - 1 - (blockIdx.y >> 1) : (blockIdx.y >> 1);
  const int k0 = band * rows_per_band, k1 = min(ny, k0 + rows_per_band);
  const int r0 = max(0, k0 - 2), r1 = min(ny, k1 + 2), nrows = r1 - r0, nb = k1 - k0;
  lfloat* sT = lds;
  lfloat* sW = sT + nrows * kPRow;
  lfloat* sU = sW + nrows * kPRow;
  lfloat* sV = sU + nb * kPairNx;
  lfloat* rowk = sV + nb * kPairNx;
  stage_row_consts(rowk, tab, ny);
  const size_t fo = (size_t)m * ny * kPairNx * 2;
  // global rows are [nx][2] = one dwordx4 per point pair i; LDS rows are [half][quad][4]
  for (int i = threadIdx.x; i < nrows * (kPairNx / 2); i += 256) {
    const int r = i / (kPairNx / 2), pi = i % (kPairNx / 2);
    const size_t g = ((size_t)(r0 + r) * kPairNx + 2 * pi) * 2;
    st4(sT + r * kPRow + ppair_off(pi), ld4(X2 + fo + g));
    st4(sW + r * kPRow + ppair_off(pi), ld4(W2p + g));
  }
  for (int i = threadIdx.x; i < nb * kPairNq; i += 256) { // winds, pre-scaled by the row's advection constants
    const int k = k0 + i / kPairNq;
    const float cu = tab.subcycled[k] ? tab.adv_ccx2[k] * 0.05f : tab.adv_ccx[k] * (1.f / 3.f);
    const float cv = tab.adv_ccy * (1.f / 3.f);
    f4 uq = ld4(ug + (size_t)k0 * kPairNx + 4 * i), vq = ld4(vg + (size_t)k0 * kPairNx + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) { uq.v[e] *= cu; vq.v[e] *= cv; }
    st4(sU + 4 * i, uq); st4(sV + 4 * i, vq);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // iterating rows.  The band's longest chain, if it is long enough to set the length of the launch, is split by
  // tracer over waves 0 and 1 (scalar arithmetic); the other iterating rows go one wave each to the remaining waves.
  int ksplit = -1, tmax = kSplitMinSweeps - 1;
  for (int k = k0; k < k1; ++k) {
    const int t = tab.dif_time2[k];
    if (t > tmax) { tmax = t; ksplit = k; }
  }
  int ci = 0;
  for (int k = k0; k < k1; ++k) {
    const RowK rk = row_consts((const lfloat*)rowk, k);
    if (!(rk.dif_time2 > 1 || rk.adv_time2 > 1)) continue;
    float* orow = Xnew2 + fo + (size_t)k * kPairNx * 2;
    if (k == ksplit) {
      if (wave == 0) pair_chain_row_comp<0>(sT, sW, sU, sV, r0, k0, k, ny, rk, lane, orow);
      if (wave == 1) pair_chain_row_comp<1>(sT, sW, sU, sV, r0, k0, k, ny, rk, lane, orow);
    } else if (ksplit >= 0) {
      if (2 + (ci++ & 1) == wave) pair_chain_row(sT, sW, sU, sV, r0, k0, k, ny, rk, lane, orow);
    } else if ((ci++ & 3) == wave) {
      pair_chain_row(sT, sW, sU, sV, r0, k0, k, ny, rk, lane, orow);
    }
  }
  // single-sweep rows: pair-quads, all threads
  for (int i = threadIdx.x; i < nb * kPairNq; i += 256) {
    const int k = k0 + i / kPairNq, q = i % kPairNq;
    const RowK rk = row_consts((const lfloat*)rowk, k);
    if (rk.dif_time2 > 1 || rk.adv_time2 > 1) continue;
    const int qm = q == 0 ? kPairNq - 1 : q - 1, qp = q == kPairNq - 1 ? 0 : q + 1;
    const lfloat* xr = sT + (k - r0) * kPRow;
    const lfloat* wr = sW + (k - r0) * kPRow;
    const q8 LT = ld8g(xr, qm), CT = ld8g(xr, q), RT = ld8g(xr, qp);
    const q8 LW = ld8g(wr, qm), CW = ld8g(wr, q), RW = ld8g(wr, qp);
    const q8 Tm1 = k >= 1 ? ld8g(xr - kPRow, q) : CT, Tp1 = k <= ny - 2 ? ld8g(xr + kPRow, q) : CT;
    const q8 Tm2 = k >= 2 ? ld8g(xr - 2 * kPRow, q) : CT, Tp2 = k <= ny - 3 ? ld8g(xr + 2 * kPRow, q) : CT;
    const q8 Wm1 = k >= 1 ? ld8g(wr - kPRow, q) : zero8(), Wp1 = k <= ny - 2 ? ld8g(wr + kPRow, q) : zero8();
    const q8 Wm2 = k >= 2 ? ld8g(wr - 2 * kPRow, q) : zero8(), Wp2 = k <= ny - 3 ? ld8g(wr + 2 * kPRow, q) : zero8();
    const f4 xq = ld4(sU + (k - k0) * kPairNx + 4 * q), yq = ld4(sV + (k - k0) * kPairNx + 4 * q);
    v2 T[12], w[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      T[j] = LT.v[j]; T[4 + j] = CT.v[j]; T[8 + j] = RT.v[j];
      w[j] = LW.v[j]; w[4 + j] = CW.v[j]; w[8 + j] = RW.v[j];
    }
    const float fm = k == 1 ? 3.f : 1.f, fp = k == ny - 2 ? 3.f : 1.f; // :766-769, :784-787
    float um[4], up[4], vm[4], vp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      split_sign(xq.v[e], um[e], up[e]);
      float a, b;
      split_sign(yq.v[e], a, b);
      vm[e] = fm * a; vp[e] = fp * b;
    }
    q8 xn;
    if (rk.sub) xn = substep_pair<true>(T, w, Tm2, Tm1, Tp1, Tp2, Wm2, Wm1, Wp1, Wp2, um, up, vm, vp, rk.dif_cc * 0.05f, rk.dif_ccy, q == kPairNq - 1);
    else xn = substep_pair<false>(T, w, Tm2, Tm1, Tp1, Tp2, Wm2, Wm1, Wp1, Wp2, um, up, vm, vp, rk.dif_cc * 0.05f, rk.dif_ccy, q == kPairNq - 1);
    float* o = Xnew2 + fo + ((size_t)k * kPairNx + 4 * q) * 2; // 4 points x 2 = two dwordx4
    st4(o, f4{{xn.v[0].x, xn.v[0].y, xn.v[1].x, xn.v[1].y}});
    st4(o + 4, f4{{xn.v[2].x, xn.v[2].y, xn.v[3].x, xn.v[3].y}});
  }
}

__global__ void pack_pairs_kernel(const float* __restrict__ state, float* __restrict__ X2, int np, int n_members) {
  const size_t n = (size_t)n_members * np;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / np, p = i % np;
    *(float2*)(X2 + 2 * i) = make_float2(state[m * 5 * np + np + p], state[m * 5 * np + 3 * (size_t)np + p]); // (Tair, q)
  }
}

} // namespace

bool pair_sweep_supported(int nx, int ny) { return nx == kPairNx && ny >= 5 && ny <= kMaxNy; }

// rows per band: (rows+4) x 2 arrays x 3 KB + rows x 3 KB of winds + 6 KB of row constants; 4 rows = 67 KB, two workgroups per CU
static int pair_band_rows() {
  static const int r = tuning_int("GREB_PAIR_ROWS", 4); // -DGREB_TUNING builds only
  return r;
}
static size_t pair_lds_bytes(int rows) {
  return (size_t)(((rows + 4) * 2 * kPRow) + 2 * rows * kPairNx + kMaxNy * kRowKWords) * sizeof(float);
}

hipError_t launch_substep_pairs(const float* X2, const float* W2p, const float* u, const float* v, float* Xnew2,
                                const RowTables* tabs, const int* tab_index, int ny, int n_members, hipStream_t s) {
  const int rows = pair_band_rows(), bands = (ny + rows - 1) / rows;
  const size_t lds = pair_lds_bytes(rows);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_pair_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sweep_pair_kernel, dim3(n_members, bands), dim3(256), lds, s, X2, W2p, u, v, Xnew2, tabs, tab_index, ny, rows);
  return hipGetLastError();
}

hipError_t launch_pack_pairs(const float* state, float* X2, int np, int n_members, hipStream_t s) {
  hipLaunchKernelGGL(pack_pairs_kernel, dim3(1024), dim3(256), 0, s, state, X2, np, n_members);
  return hipGetLastError();
}

} // namespace greb
