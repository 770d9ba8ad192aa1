// greb_pair_sweep.hip -- FAST circulation sub-step of the any-grid engine for many members on 384-wide grids.
//
// The scalar any-grid kernel (greb_kernels.hip: sweep_kernel) treats the two transported tracers of a member as two
// batch items.  Here they travel as (Tair, q) PAIRS:
//   X2 / Xnew2 : [member][ny][nx][{Tair,q}]   W2p : [ny][nx][{wz_air,wz_vapor}]   winds: [ny][nx], shared
// One workgroup = one member x one latitude band, tracer rows and weights staged in LDS as [half][quad][4] rows
// (greb_pair.h) so the dwordx4 reads of neighbouring lanes do not collide.
//   * Single-sweep rows (most of the grid): pair-quads, every arithmetic instruction a packed v_pk_*_f32 on a
//     (Tair,q) register pair (greb_pair.h, as in the fused 96x48 engine) -- one address stream, one wind sign split
//     for the two tracers.
//   * Iterating rows (within ~18 degrees of the poles, up to 225 Jacobi sweeps per diffusion call): one task per
//     (row, tracer) in the registers of one wave, lane l owning longitudes 6l..6l+5 (greb_chain6.h: a chain's latency
//     is its instruction count, and the two tracers are independent chains).
#include <cstdlib>

#include "greb_kernels.h"
#include "greb_pair.h"
#include "greb_stencil.h"

namespace greb {
namespace {

constexpr int kPairNx = 384, kPairNq = 96, kPairP = 6;
static_assert(kPairP == 6, "the chain's min3 reduction is written for six points per lane");
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

// ONE tracer (C = 0: Tair, 1: q) of an iterating row: the row's dif_time2 + adv_time2 dependent sweeps as scalar
// chains in the registers of one wave (greb_chain6.h: 36 instructions per sweep; a chain's latency is its instruction
// count, and per tracer the pipe time of a sweep is ~120 cycles where the same sweep on (Tair,q) pairs -- 11 packed
// subtractions, 36 packed multiply-adds, 12 halo moves, 2 x 6 selects -- took ~450 for the two), then the latitudinal
// terms and the update of that tracer.  Tair and q of a row are independent chains, so they are separate tasks.
template <int C>
__device__ void pair_chain_row_comp(const lfloat* sT, const lfloat* sW, const float* __restrict__ ug, const float* __restrict__ vg, int r0,
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
  // the row's winds, scaled by its advection constants (read once per task: nothing to gain from staging them)
  const float cu = rk.sub ? rk.adv_cc * 0.05f : rk.adv_cc * (1.f / 3.f), cv = rk.adv_ccy * (1.f / 3.f);
  float us[P], vs[P];
#pragma unroll
  for (int i = 0; i < P; i += 2) {
    const float2 a = *(const float2*)(ug + (size_t)k * kPairNx + P * lane + i), b = *(const float2*)(vg + (size_t)k * kPairNx + P * lane + i);
    us[i] = cu * a.x; us[i + 1] = cu * a.y; vs[i] = cv * b.x; vs[i + 1] = cv * b.y;
  }
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
    split_sign(vs[i], vpos, vneg);
    const float gm1 = Wm1 * (Tm1 - own), gp1 = Wp1 * (Tp1 - own);
    const float dm2 = Wm2 * (own - Tm2), dp2 = Wp2 * (own - Tp2);
    const float ddy = rk.dif_ccy * (gm1 + gp1);
    const float day = (ap * vneg) * (dp2 - gp1) - (am * vpos) * (dm2 - gm1);
    const float dd = w[3 + i] * ((Th[0][i] - own) + ddy); // :718, :721
    const float da = (Th[1][i] - own) + day;              // :910, :913
    out_row[2 * x + C] = (own + dd) + da;                 // :549
  }
}

// Members per workgroup (256 threads each).  A band's weights (24 KB) are the same for every member (so were its winds, 12 KB, staged then),
// only its 24 KB of tracer rows are the member's own, and skip experiments (62 members: 57 us per launch, of which 26
// with all arithmetic skipped) showed the launch co-limited by exactly that staging.  What paid: all global loads of a
// band requested before anything is waited for (26 -> 19.6 us with the arithmetic skipped, 56.5 -> 52.6 us in all).
// What did not: sharing the weights and winds among 2 members per workgroup (84 KB of LDS, one 8-wave workgroup per
// CU: 56.0 us, and 49.1 instead of 36.0 at 40 members -- the phases of a workgroup are sequential, and one resident
// workgroup has nobody to overlap them with); 3 or 4 members cap the kernel at 168 / 128 VGPRs where it wants 223
// (spills: 76 / 186 us).  The code keeps the group size as a parameter.
constexpr int kPairGroup = 1;
constexpr int kMaxBandRows = 8; // 12 staged rows x 3 KB x (tracers + weights) = 72 KB: still two workgroups per CU
constexpr int kPairThreads = 256 * kPairGroup;

__global__ __launch_bounds__(kPairThreads) void sweep_pair_kernel(const float* __restrict__ X2, const float* __restrict__ W2p,
                                                                  const float* __restrict__ ug, const float* __restrict__ vg,
                                                                  float* __restrict__ Xnew2, const RowTables* __restrict__ tabp,
                                                                  const int* __restrict__ tab_index, int ny, int n_members,
                                                                  int rows_per_band, int nbands, int dbg) {
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255; // member slot of the workgroup, thread within the member
  // Workgroup -> (member group, band), XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs, each with
  // its own L2.  Neighbouring bands of a member share 4 of their 8 tracer rows, so all bands of a member group are
  // given to ONE XCD (id % 8 = group % 8) and follow each other closely there: the second reader of a halo row finds
  // it in that XCD's L2 instead of fetching it from the Infinity Cache again.  Poles first: the polar bands carry the
  // long chains.
  const int ngroups = (n_members + kPairGroup - 1) / kPairGroup, per_xcd = (ngroups + 7) / 8;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int group = (slot % per_xcd) * 8 + xcd, bo = slot / per_xcd;
  if (group >= ngroups) return; // (whole workgroup)
  const int band = (bo & 1) ? nbands - 1 - (bo >> 1) : (bo >> 1);
  const int m = group * kPairGroup + grp;
  const bool live = m < n_members; // (the last group may hold fewer members; its idle slots only join the barrier)
  const RowTables& tab = tabp[tab_index ? tab_index[live ? m : n_members - 1] : 0];
  const int k0 = band * rows_per_band, k1 = min(ny, k0 + rows_per_band);
  const int r0 = max(0, k0 - 2), r1 = min(ny, k1 + 2), nrows = r1 - r0, nb = k1 - k0;
  // shared by the members: weights; per member: tracer rows, row constants of the band's rows
  lfloat* sW = lds;
  lfloat* sT = sW + (rows_per_band + 4) * kPRow + grp * ((rows_per_band + 4) * kPRow + rows_per_band * kRowKWords);
  lfloat* rowk_band = sT + (rows_per_band + 4) * kPRow;
  const lfloat* rowk = rowk_band - k0 * kRowKWords; // indexed by the absolute row
  const size_t fo = (size_t)(live ? m : 0) * ny * kPairNx * 2;
  // Everything the band needs from global memory is requested before anything is waited for: the loads are
  // unconditional (clamped indices) and only the LDS stores are predicated -- a load under a condition is preceded by
  // s_waitcnt vmcnt(0), which turned the staging into six dependent round trips.
  // global rows are [nx][2] = one dwordx4 per point pair i; LDS rows are [half][quad][4]
  constexpr int kXIt = 6; // tracer loads in flight per thread: (4 + 4 rows) x 192 point pairs / 256 threads
  constexpr int kWIt = (kXIt + kPairGroup - 1) / kPairGroup;
  const int npair = nrows * (kPairNx / 2);
  auto goff = [&](int i) { return ((size_t)(r0 + i / (kPairNx / 2)) * kPairNx + 2 * (i % (kPairNx / 2))) * 2; };
  auto loff = [&](int i) { return (i / (kPairNx / 2)) * kPRow + ppair_off(i % (kPairNx / 2)); };
  for (int base = 0; base < npair; base += kXIt * 256) {
    f4 x[kXIt], wq[kWIt];
#pragma unroll
    for (int j = 0; j < kXIt; ++j) x[j] = ld4(X2 + fo + goff(min(base + j * 256 + tid, npair - 1)));
#pragma unroll
    for (int j = 0; j < kWIt; ++j) wq[j] = ld4(W2p + goff(min(base + j * kPairThreads + (int)threadIdx.x, npair - 1)));
    float rc[kRowKWords] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (base == 0) { // the row constants ride in the first batch
      // (the tables hang off tab_index[m]: a dependent round trip, which must not be waited for before the loads above
      // are on their way)
      __builtin_amdgcn_sched_barrier(0);
      const int k = k0 + min(tid, nb - 1), sub = tab.subcycled[k]; // the member's constants of the band's rows
      const float d2 = tab.dif_ccx2[k], d1 = tab.dif_ccx[k], a2 = tab.adv_ccx2[k], a1 = tab.adv_ccx[k];
      rc[0] = sub ? d2 : d1; rc[1] = sub ? a2 : a1; rc[2] = tab.dif_ccy; rc[3] = tab.adv_ccy;
      rc[4] = __int_as_float(sub); rc[5] = __int_as_float(tab.dif_time2[k]); rc[6] = __int_as_float(tab.adv_time2[k]);
    }
#pragma unroll
    for (int j = 0; j < kXIt; ++j) {
      const int i = base + j * 256 + tid;
      if (live && i < npair) st4(sT + loff(i), x[j]);
    }
#pragma unroll
    for (int j = 0; j < kWIt; ++j) {
      const int i = base + j * kPairThreads + (int)threadIdx.x;
      if (i < min(npair, base + kXIt * 256)) st4(sW + loff(i), wq[j]);
    }
    if (base == 0) {
      if (tid < nb) {
        st4(rowk_band + tid * kRowKWords, f4{{rc[0], rc[1], rc[2], rc[3]}});
        st4(rowk_band + tid * kRowKWords + 4, f4{{rc[4], rc[5], rc[6], rc[7]}});
      }
    }
  }
  __syncthreads();
  if (!live) return;
  const int wave = tid >> 6, lane = tid & 63;
  // iterating rows: every (row, tracer) is one chain task; longest row first, each task to the wave with the least
  // chain work so far (the same scalar bookkeeping in all four waves).  In the polar bands that puts the two tracers of
  // the 225-sweep row on two waves -- two SIMDs -- and everything else on the other two.
  {
    const int me = __builtin_amdgcn_readfirstlane(wave);
    int l0 = 0, l1 = 0, l2 = 0, l3 = 0;
    unsigned done = 0;
    for (int it = 0; it < nb && !(dbg & 1); ++it) {
      int kb = -1, cb = 0;
      for (int k = k0; k < k1; ++k) { // (from the staged constants: the tables themselves are a global round trip away)
        const int td = __builtin_amdgcn_readfirstlane(__float_as_int(rowk[k * kRowKWords + 5]));
        const int ta = __builtin_amdgcn_readfirstlane(__float_as_int(rowk[k * kRowKWords + 6]));
        if (!((done >> (k - k0)) & 1u) && (td > 1 || ta > 1) && td + ta > cb) { cb = td + ta; kb = k; }
      }
      if (kb < 0) break;
      done |= 1u << (kb - k0);
      const RowK rk = row_consts((const lfloat*)rowk, kb);
      float* orow = Xnew2 + fo + (size_t)kb * kPairNx * 2;
#pragma unroll
      for (int C = 0; C < 2; ++C) {
        const int m01 = l1 < l0 ? 1 : 0, v01 = l1 < l0 ? l1 : l0, m23 = l3 < l2 ? 3 : 2, v23 = l3 < l2 ? l3 : l2;
        const int to = v23 < v01 ? m23 : m01;
        l0 += to == 0 ? cb : 0; l1 += to == 1 ? cb : 0; l2 += to == 2 ? cb : 0; l3 += to == 3 ? cb : 0;
        if (to == me) {
          if (C == 0) pair_chain_row_comp<0>(sT, sW, ug, vg, r0, kb, ny, rk, lane, orow);
          else pair_chain_row_comp<1>(sT, sW, ug, vg, r0, kb, ny, rk, lane, orow);
        }
      }
    }
  }
  // single-sweep rows: pair-quads, all threads
  for (int i = tid; i < nb * kPairNq && !(dbg & 2); i += 256) {
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
    f4 xq = ld4(ug + (size_t)k * kPairNx + 4 * q), yq = ld4(vg + (size_t)k * kPairNx + 4 * q);
    {
      const float cu = rk.sub ? rk.adv_cc * 0.05f : rk.adv_cc * (1.f / 3.f), cv = rk.adv_ccy * (1.f / 3.f);
#pragma unroll
      for (int e = 0; e < 4; ++e) { xq.v[e] *= cu; yq.v[e] *= cv; }
    }
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

// rows per band; LDS = weights (rows+4) x 3 KB + per member ((rows+4) x 3 KB + row constants): 4 rows, 1 member:
// 48 KB.  (Three workgroups would fit a CU by LDS, but the kernel's 223 VGPRs allow two; capped at 168 VGPRs it
// spills 64 of them and takes 67.8 instead of 52.6 us per launch at 62 members.)  The winds are read by the tasks
// themselves: every wind value is used by exactly one task.
// A band stages rows + 4 rows to update rows: the larger the band, the less of the staging is halo (2x at 4 rows, 1.5x
// at 8) -- and the fewer, longer workgroups there are to fill the chip with.  Measured (us per launch, rows 4 / 6 / 8):
// 40 members 35.1 / 36.8 / 45.4, 48: 40.6 / 40.8 / 46.0, 56: 46.8 / 45.5 / 46.1, 62: 52.0 / 50.3 / 47.8,
// 96: 71.1 / 69.2 / 64.2, 128: 90.7 / 87.8 / 80.1.
static int pair_band_rows(int n_members) {
  static const int forced = tuning_int("GREB_PAIR_ROWS", 0); // -DGREB_TUNING builds only
  if (forced > 0) return forced;
  return n_members >= 58 ? 8 : (n_members >= 50 ? 6 : 4);
}
static size_t pair_lds_bytes(int rows) {
  return (size_t)((rows + 4) * kPRow + kPairGroup * ((rows + 4) * kPRow + rows * kRowKWords)) * sizeof(float);
}

hipError_t launch_substep_pairs(const float* X2, const float* W2p, const float* u, const float* v, float* Xnew2,
                                const RowTables* tabs, const int* tab_index, int ny, int n_members, hipStream_t s) {
  const int rows = pair_band_rows(n_members), bands = (ny + rows - 1) / rows;
  if (rows < 1 || rows > kMaxBandRows) return hipErrorInvalidValue;
  const size_t lds = pair_lds_bytes(rows);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_pair_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
  if (e != hipSuccess) return e;
  const int ngroups = (n_members + kPairGroup - 1) / kPairGroup;
  hipLaunchKernelGGL(sweep_pair_kernel, dim3(8 * ((ngroups + 7) / 8) * bands), dim3(kPairThreads), lds, s, X2, W2p, u, v, Xnew2, tabs,
                     tab_index, ny, n_members, rows, bands,
                     tuning_int("GREB_DEBUG_PAIR", 0)); // -DGREB_TUNING builds only: bit 0 skips the chains, bit 1 the single-sweep rows
  return hipGetLastError();
}

hipError_t launch_pack_pairs(const float* state, float* X2, int np, int n_members, hipStream_t s) {
  hipLaunchKernelGGL(pack_pairs_kernel, dim3(1024), dim3(256), 0, s, state, X2, np, n_members);
  return hipGetLastError();
}

} // namespace greb
