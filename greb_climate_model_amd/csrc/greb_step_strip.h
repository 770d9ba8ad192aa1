// greb_step_strip.h -- one circulation sub-step X <- (X + dX_diffuse) + dX_advec (src/greb.f90:549, with :556-723 and
// :726-915 behind it) for a STRIP of consecutive latitude rows of one (member, tracer) field of a 384-wide grid -- or of a
// 192-wide one, the row laid twice around the wavefront (greb_rows.h: NXR) --, worked by one wavefront.  Shared by the two kernels that run it:
//   greb_step_rows.hip   one launch per sub-step (plain loads and stores; the kernel boundary orders the sub-steps);
//   greb_circ_rows.hip   one launch per circulation CALL: the 24 sub-steps (src/greb.f90:546-550) in one kernel, the
//                        strips handing their rows to their neighbours through memory flags (loads and stores of the
//                        tracer rows at agent scope, `sc1`).
//   * rows k-2 .. k+2 of the tracer and its weight (the meridional stencils of both operators) are a window in registers
//     that slides up one row per step; rows arrive by LDS-DMA up to three ahead of it (greb_rows.h: four landing slots,
//     19.5 KB of LDS per wavefront, eight wavefronts per CU); the zonal halo is a wave rotate (DPP);
//   * the winds of the row travel the same way (a ring of two);
//   * per row: the zonal edge fluxes once, shared by the diffusion and the advection sweep (greb_device.h: edge-flux
//     form); rows that iterate run their sweeps in registers (greb_chain6.h), diffusion and advection chains one after
//     the other in the same wave.
// STRICT keeps the reference's expression trees (bit-exact), FAST the re-associated ones of the other kernels.
#pragma once
#include "greb_rows.h"

namespace greb {
namespace rows {

constexpr int kRing = 4;   // landing slots of the tracer/weight rows: a row waits here until the window takes it
constexpr unsigned kOutBase = 0, kRingBase = kRowB, kWindBase = kRowB + kRing * kSlotB;
constexpr unsigned kStepLdsB = kWindBase + 2 * kSlotB; // 19.5 KB: eight wavefronts per CU
constexpr int kAuxPlain = 0, kAuxSc1 = 16; // cache policy of an LDS-DMA load: default / agent scope (gfx940+: sc1 = bit 4)

// Immutable for the life of a launch and read through the CONSTANT address space: only that lets the compiler select
// scalar loads (s_load) behind the kernel's own global stores and LDS-DMA -- through a generic pointer the per-row
// constants were four global_load_dword + s_waitcnt vmcnt(0) in the row loop, i.e. every row drained the whole LDS-DMA
// ring (tests/test_isa_cpu.py guards it)
typedef __attribute__((address_space(4))) RowTables crow_tables;

// sixteen bytes per lane to global memory; SC1: written through to memory at agent scope, the form a reader on another
// XCD can see after the writer's s_waitcnt vmcnt(0) (MI355X: the per-XCD L2s are not coherent with each other)
template <bool SC1>
__device__ __forceinline__ void store16(float* p, vfloat4 q) {
  if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(q) : "memory");
  else *reinterpret_cast<vfloat4*>(p) = q;
}

// -DGREB_TUNING builds only: where a strip leaves its s_memtime stamps (null otherwise)
struct StripStamps {
  unsigned long long* first;  // [0..5] of the strip's first row: loop entry, own row + wind landed, before the diffusion
                              // chain, before the advection chain, after it, row stored
  unsigned long long* phases; // [8] start, [11] window filled, [9] end, [10] rows, [12..15] phase totals of the strip
};
#ifdef GREB_TUNING
#define GREB_STEP_STAMP(i) if (st.first && r == k0 && lane == 0) st.first[i] = __builtin_amdgcn_s_memtime()
// phase totals: [12] issue + window advance (waits for the row), [13] zonal part, [14] meridional part + store,
// [15] the next row's winds (waits for them)
#define GREB_STEP_PHASE(i)                                                          \
  if (stamp_last) {                                                                 \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                   \
    if (i > 0) phase_sum[(i) - 1] += now_ - phase_t;                                \
    phase_t = now_;                                                                 \
  }
#else
#define GREB_STEP_STAMP(i)
#define GREB_STEP_PHASE(i)
#endif

// the meridional stencils of both operators and the update of one row (:585-590, :756-795, :718-721, :910-913, :549):
// Tw / ww = rows r-2 .. r+2 of the tracer and its weight (a row outside the grid has weight zero), Td / Ta the row after
// its zonal diffusion / advection sweeps (T1h), v the meridional wind
template <bool STRICT>
__device__ __forceinline__ void meridional_update(const float (&Tw)[5][6], const float (&ww)[5][6], const float (&Td)[6],
                                                  const float (&Ta)[6], const float (&v)[6], float ccy_dif, float ccy_adv,
                                                  int r, int ny, float (&o)[6]) {
  const float (&T0)[6] = Tw[2];
  const float (&w0)[6] = ww[2];
  if (STRICT) {
#pragma clang fp contract(off)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float dyd = dif_lat_point_strict<float>(T0[j], Tw[1][j], Tw[3][j], ww[1][j], ww[3][j], ccy_dif, r, ny);
      const float dya = adv_lat_point_strict<float>(T0[j], Tw[0][j], Tw[1][j], Tw[3][j], Tw[4][j], ww[0][j], ww[1][j], ww[3][j],
                                                    ww[4][j], v[j], ccy_adv, r, ny);
      const float dd = w0[j] * ((Td[j] - T0[j]) + dyd); // :718, :721
      const float da = (Ta[j] - T0[j]) + dya;           // :910, :913
      o[j] = T0[j] + dd + da;                           // :549
    }
  } else {
    float am, ap;
    adv_lat_coef(ccy_adv, r, ny, am, ap);
    const float ccyd = ccy_dif;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float gm1 = ww[1][j] * (Tw[1][j] - T0[j]), gp1 = ww[3][j] * (Tw[3][j] - T0[j]);
      const float dm2 = ww[0][j] * (T0[j] - Tw[0][j]), dp2 = ww[4][j] * (T0[j] - Tw[4][j]);
      const float dyd = ccyd * (gm1 + gp1);
      const float dya = ap * wind_neg(v[j]) * (dp2 - gp1) - am * wind_pos(v[j]) * (dm2 - gm1);
      const float dd = w0[j] * ((Td[j] - T0[j]) + dyd);
      const float da = (Ta[j] - T0[j]) + dya;
      {
#pragma clang fp contract(off)
        o[j] = T0[j] + dd + da; // the reference's two roundings, :549
      }
    }
  }
}

// the lane's window of a row for the register chains: its 6 points with three halo points on either side
__device__ __forceinline__ void chain_halo(const float (&x)[6], float (&xc)[12]) {
#pragma unroll
  for (int j = 0; j < 6; ++j) xc[3 + j] = x[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) { xc[j] = wave_from_prev(x[3 + j]); xc[9 + j] = wave_from_next(x[j]); }
}

struct StripIo {
  const float* Xf; // the field's tracer at the start of the sub-step
  const float* wf; // its weight (wz_air / wz_vapor)
  float* of;       // the field's tracer after the sub-step
  const float* u;  // winds of the model step, shared by every member
  const float* v;
};

// Rows [k0, k1) of one field, one sub-step.  On entry the wavefront has no vector-memory operation outstanding (vmcnt is
// counted by hand from zero); on exit the stores of the last rows may still be in flight.
template <bool STRICT, int AUX_X, bool SC1, int NXR = kNx, int RING = kRing>
__device__ __forceinline__ void stream_strip(lfloat* lds, const StripIo& io, const crow_tables& tab, int k0, int k1, int ny,
                                             bool calm, int chains_first, unsigned lane, const StripStamps& st) {
  const float* Xf = io.Xf;
  const float* wf = io.wf;
  float* of = io.of;
  const PairPtrs PX = pair_ptrs<NXR>(Xf, wf, lane), PU = pair_ptrs<NXR>(io.u, io.v, lane);
  const LaneAddr L = lane_addr(lane);
  const unsigned lb = (unsigned)(size_t)lds;
  const bool last_lane = bug_lane<NXR>(lane);
  constexpr int kStores = NXR == kNx ? 2 : 1; // store instructions per row (store_row_quads)
  static_assert(RING == 3 || RING == 4, "landing slots of the tracer / weight rows (gT holds four 16-bit counters)");
  constexpr unsigned kWindAt = kRowB + RING * kSlotB; // (the wind ring follows the row ring)
  auto ring_slot = [](int row) { return RING == 4 ? (row & 3) : row % 3; };
  if (!chains_first) __builtin_amdgcn_s_setprio(2); // streaming rows ahead of the chains (which drop to 0 while they sweep)
  int ops = 0;
  unsigned long long gT = 0, gU = 0; // 16 bits per slot: `ops` right after the slot's LDS-DMA was issued
  auto issue_T = [&](int row) {
    const int slot = ring_slot(row);
    issue_pair_p<AUX_X>(PX, row * NXR, lds + (kRingBase + slot * kSlotB) / 4);
    ops += 3;
    gT = (gT & ~(0xffffull << (16 * slot))) | ((unsigned long long)ops << (16 * slot));
  };
  auto issue_U = [&](int row) {
    const int slot = row & 1;
    issue_pair_p<kAuxPlain>(PU, row * NXR, lds + (kWindAt + slot * kSlotB) / 4);
    ops += 3;
    gU = (gU & ~(0xffffull << (16 * slot))) | ((unsigned long long)ops << (16 * slot));
  };
  // Rows lo .. hi are read, in order, each once.  The five rows k-2 .. k+2 the meridional stencils of both operators
  // need are a WINDOW IN REGISTERS that slides up one row per step; a row waits for the window in one of kRing LDS
  // slots (up to kRing - 1 rows are in flight ahead of the window), and the slot is refilled as soon as it is read.
  const int lo = k0 >= 2 ? k0 - 2 : 0, hi = k1 + 1 < ny ? k1 + 1 : ny - 1;
  int next_issue = lo;
  for (int j = 0; j < RING && next_issue <= hi; ++j) issue_T(next_issue++);
  issue_U(k0);
  float Tw[5][6], ww[5][6]; // rows c-2 .. c+2 of the tracer and its weight; a row outside the grid has weight zero
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) { Tw[i][j] = 0.f; ww[i][j] = 0.f; }
  auto advance = [&](int c) { // the window moves up to centre row c: row c+2 enters
    const int row = c + 2;
    const bool have = row >= lo && row <= hi;
    PairRaw raw;
    if (have) {
      const int slot = ring_slot(row);
      // (mid-strip the row was requested RING steps ago: 2 + (RING - 1) x 8 + 3 operations since, at two stores per row)
      wait_all_but_mostly<5 + (RING - 1) * (6 + kStores)>(ops - (int)((gT >> (16 * slot)) & 0xffff));
      read_pair_issue(L, lb + kRingBase + slot * kSlotB, raw); // ... and the window shifts under the LDS latency
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 6; ++j) { Tw[i][j] = Tw[i + 1][j]; ww[i][j] = ww[i + 1][j]; }
    if (have) {
      read_pair_finish(raw, Tw[4], ww[4]);
      if (next_issue <= hi) issue_T(next_issue++);
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) { Tw[4][j] = 0.f; ww[4][j] = 0.f; }
    }
  };
#ifdef GREB_TUNING
  const bool stamp_last = st.phases && lane == 0;
  if (stamp_last) { st.phases[8] = __builtin_amdgcn_s_memtime(); st.phases[10] = (unsigned long long)(k1 - k0); }
  unsigned long long phase_t = 0, phase_sum[4] = {0, 0, 0, 0};
#endif
  for (int c = lo - 2; c < k0; ++c) advance(c); // fill: after this the window is centred on row k0 - 1
#ifdef GREB_TUNING
  if (stamp_last) st.phases[11] = __builtin_amdgcn_s_memtime();
#endif

  // the winds of row r are read into registers one step early and the slot refilled at once: a row's wind is requested
  // two steps before it is used (requested one step ahead it was not there yet: 4 700 cycles per row instead of ~2 000)
  // the row constants come through the scalar cache (s_load: the table is in the constant address space), requested one
  // row ahead (asked for where they are used, each row waited 300-600 cycles for them)
  const float ccy_dif = tab.dif_ccy, ccy_adv = tab.adv_ccy;
  int t2d_n = tab.dif_time2[k0], t2a_n = tab.adv_time2[k0];
  float ccd_n = tab.dif_ccx2[k0], cca_n = tab.adv_ccx2[k0];
  float u[6], v[6];
  if (k0 + 1 < k1) issue_U(k0 + 1);
  wait_all_but(ops - (int)((gU >> (16 * (k0 & 1))) & 0xffff));
  read_pair(L, lb + kWindAt + (k0 & 1) * kSlotB, u, v);
  for (int r = k0; r < k1; ++r) {
    GREB_STEP_STAMP(0);
    GREB_STEP_PHASE(0);
    if (r + 2 < k1) issue_U(r + 2); // into the slot of row r, whose winds are in registers
    advance(r);
    GREB_STEP_STAMP(1);
    GREB_STEP_PHASE(1);
    if (calm) {
#pragma unroll
      for (int j = 0; j < 6; ++j) { u[j] = 0.f; v[j] = 0.f; }
    }
    const float (&T0)[6] = Tw[2];
    const float (&w0)[6] = ww[2];
    const int t2d = t2d_n, t2a = t2a_n;
    const float ccd = ccd_n, cca = cca_n;
    {
      const int rn = r + 1 < k1 ? r + 1 : r;
      t2d_n = tab.dif_time2[rn]; t2a_n = tab.adv_time2[rn]; ccd_n = tab.dif_ccx2[rn]; cca_n = tab.adv_ccx2[rn];
    }
    // ---- zonal part: the two sub-cycled results T1h (:656-717, :842-909)
    float Td[6], Ta[6];
    if (STRICT || t2d > 1 || t2a > 1) {
      float Tc[12], wc[12];
      chain_halo(T0, Tc);
      chain_halo(w0, wc);
      const float u0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float T2[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) T2[j] = Tc[j];
      GREB_STEP_STAMP(2);
      if (STRICT || t2d > 1) {
        if (!chains_first) __builtin_amdgcn_s_setprio(0);
        chain_window<STRICT, 6>(Tc, wc, u0, ccd, t2d, false, last_lane ? 63 : 0, chains_first != 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) Td[j] = Tc[3 + j];
      }
      GREB_STEP_STAMP(3);
      if (STRICT || t2a > 1) {
        if (!chains_first) __builtin_amdgcn_s_setprio(0);
        chain_window<STRICT, 6>(T2, wc, u, cca, t2a, true, last_lane ? 63 : 0, chains_first != 0); // (63: the lane with the :881 index bug)
#pragma unroll
        for (int j = 0; j < 6; ++j) Ta[j] = T2[3 + j];
      }
    }
    if (!chains_first) __builtin_amdgcn_s_setprio(2);
    if (!STRICT && (t2d <= 1 || t2a <= 1)) {
      RowFlux f;
      row_flux(T0, w0, f);
      if (t2d <= 1) dif_sweep_fast(T0, f, ccd * 0.05f, Td);
      if (t2a <= 1) adv_sweep_fast(T0, u, f, cca * 0.05f, last_lane, Ta);
    }
    GREB_STEP_STAMP(4);
    GREB_STEP_PHASE(2);
    // ---- meridional part and the update
    float o[6];
    meridional_update<STRICT>(Tw, ww, Td, Ta, v, ccy_dif, ccy_adv, r, ny, o);
    vfloat4 q0, q1;
    transpose_out(L, lb + kOutBase, o, q0, q1);
    ops += store_row_quads<NXR>(of + r * NXR, lane, q0, q1, [](float* p_, vfloat4 q_) { store16<SC1>(p_, q_); });
    order_fence();
    GREB_STEP_STAMP(5);
    GREB_STEP_PHASE(3);
    if (r + 1 < k1) { // the next row's winds (requested at the start of the previous step: 13 operations since, mid-strip)
      wait_all_but_mostly<11 + kStores>(ops - (int)((gU >> (16 * ((r + 1) & 1))) & 0xffff));
      read_pair(L, lb + kWindAt + ((r + 1) & 1) * kSlotB, u, v);
    }
    GREB_STEP_PHASE(4);
  }
#ifdef GREB_TUNING
  if (stamp_last) {
    st.phases[9] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4; ++i) st.phases[12 + i] = phase_sum[i];
  }
#endif
}

} // namespace rows
} // namespace greb
