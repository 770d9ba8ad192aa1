// greb_step_rows.hip -- one circulation sub-step X <- (X + dX_diffuse) + dX_advec (src/greb.f90:549, with :556-723 and
// :726-915 behind it) of the any-grid engine on 384-wide grids, as ROW STRIPS: the engine-side twin of greb_rows.hip.
//
// One wavefront = one task = a strip of consecutive latitude rows of one (member, tracer) field; no workgroup, no
// barrier: a wave that owns a 232-sweep polar row (225 diffusion + 7 advection sweeps) starts its chain
// as soon as ITS row and wind have landed and nobody else waits for it.  With few fields that chain is the length of the
// launch (138-147 cycles per sweep -- greb_chain6.h: one instruction per 4.0 cycles, no test between the sweeps -- so
// 14.9 us for that row with its set-up and epilogue, tools/stamp_step_rows.py); the band kernels (greb_kernels.hip:
// sweep_kernel<fused>, and round 2's (Tair,q)-pair band kernel) add staging, a workgroup barrier and the band's epilogue
// to it: 25.1 us per launch for one member, 48 us for 62 -- here 18 and 38.5.  With many fields the launch is bound by
// instruction issue (one vector instruction per SIMD every 4 cycles) and by how evenly the SIMDs are loaded:
// step_rows_tasks below.
//   * rows k-2 .. k+2 of the tracer and its weight (the meridional stencils of both operators) are a window in registers
//     that slides up one row per step; rows arrive by LDS-DMA up to three ahead of it (greb_rows.h: four landing slots,
//     19.5 KB of LDS per wavefront, eight wavefronts per CU); the zonal halo is a wave rotate (DPP);
//   * the winds of the row travel the same way (a ring of two);
//   * per row: the zonal edge fluxes once, shared by the diffusion and the advection sweep (greb_device.h: edge-flux
//     form); rows that iterate run their sweeps in registers (greb_chain6.h), diffusion and advection chains one after
//     the other in the same wave;
//   * the launch order is ONE round of at most as many tasks as the chip has wavefront slots (step_rows_tasks).
// STRICT keeps the reference's expression trees (bit-exact), FAST the re-associated ones of the other kernels.
#include <algorithm>
#include <cstring>
#include <vector>

#include "greb_rows.h"

namespace greb {
namespace {
using namespace rows;

typedef const __attribute__((address_space(4))) unsigned long long* cull_ptr; // scalar-loadable: immutable during a launch
typedef __attribute__((address_space(4))) RowTables crow_tables;
constexpr int kRing = 4;   // landing slots of the tracer/weight rows: a row waits here until the window takes it
constexpr unsigned kOutBase = 0, kRingBase = kRowB, kWindBase = kRowB + kRing * kSlotB;
constexpr unsigned kStepLdsB = kWindBase + 2 * kSlotB; // 19.5 KB: eight wavefronts per CU

struct StepArgs {
  const float* X;          // [n_members][2][ny][nx]  {Tair, q}
  const float* W2;         // [2][ny][nx]             {wz_air, wz_vapor}
  const float* u;          // [ny][nx] winds of the step, shared by every member
  const float* v;
  float* Xnew;
  const RowTables* tabs;
  const int* tab_index;    // [n_members]
  const RowsTask* tasks;   // field word = (2 * member + tracer) | the member's row-table index << 16 (kStepFieldBits)
  // The first tasks of the launch -- with few fields the 232-sweep polar rows, which ARE the launch -- by value: a
  // wavefront's start is a chain of dependent memory latencies (arguments -> task -> rows, ~0.4 us each after the
  // cache invalidation at the kernel boundary), and these tasks skip the middle one
  unsigned long long head[kStepHeadTasks];
  int ny, calm_odd;        // calm_odd: the vapour fields see zero wind (greb.original.model.f90:560-564)
  int chains_first;        // who issues first where a chain and a streaming strip share a SIMD (launch_substep_rows)
  unsigned long long* stamps; // -DGREB_TUNING builds only (null otherwise): s_memtime stamps of task 0
  unsigned long long* timeline; // -DGREB_TUNING builds only: [task][start, end] in s_memrealtime ticks (100 MHz) + [2 n]: hw id
};
#ifdef GREB_TUNING
#define GREB_STEP_STAMP(i) if (a.stamps && blockIdx.x == 0 && r == k0 && lane == 0) a.stamps[i] = __builtin_amdgcn_s_memtime()
// phase totals of the task three quarters down the launch order: [12] issue + window advance (waits for the row),
// [13] zonal part, [14] meridional part + store, [15] the next row's winds (waits for them)
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

template <bool STRICT>
__global__ __launch_bounds__(64) void step_rows_kernel(const StepArgs a) {
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  // (field, rows) as ONE 8-byte scalar load issued with the argument loads: as two fields with a test between them the
  // compiler loaded them one after the other, a memory latency each
  // (each arm ends in its own readfirstlane: as a plain conditional the compiler selects between the two ADDRESSES and
  // issues one flat vector load -- the task, and every loop bound derived from it, in VGPRs; an opaque asm statement
  // instead makes every later scalar load of the kernel a vector load, tab_index first)
  int fld, task_rows; // fld: the field in the low 16 bits, the row-table index above
  if (blockIdx.x < kStepHeadTasks) {
    const unsigned long long t = a.head[blockIdx.x];
    fld = __builtin_amdgcn_readfirstlane((int)(unsigned)t); task_rows = __builtin_amdgcn_readfirstlane((int)(t >> 32));
  } else {
    const unsigned long long t = *(cull_ptr)(a.tasks + blockIdx.x);
    fld = __builtin_amdgcn_readfirstlane((int)(unsigned)t); task_rows = __builtin_amdgcn_readfirstlane((int)(t >> 32));
  }
#ifdef GREB_TUNING
  if (a.timeline && threadIdx.x == 0) {
    a.timeline[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); // wave, SIMD, CU, SE ids
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.timeline[2 * gridDim.x + blockIdx.x] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  const int tab_idx = (int)((unsigned)fld >> kStepFieldBits);
  fld &= (1 << kStepFieldBits) - 1;
  const int k0 = task_rows & 0xff, k1 = (task_rows >> 8) & 0x1ff, ny = a.ny;
  const unsigned lane = threadIdx.x;
  const int member = fld >> 1, tracer = fld & 1;
  // the row table through the CONSTANT address space: it is immutable for the life of the launch, and only that lets the
  // compiler select scalar loads (s_load) behind the kernel's own global stores and LDS-DMA -- as a generic pointer the
  // per-row constants were four global_load_dword + s_waitcnt vmcnt(0) in the row loop: every row drained the whole
  // LDS-DMA ring (the rows requested three ahead, the winds), which the hand-counted waits exist to avoid
  const crow_tables& tab = *(const crow_tables*)(a.tabs + tab_idx); // (index in the task word: one dependent latency less)
  const size_t np = (size_t)kNx * ny;
  const float* Xf = a.X + (size_t)fld * np;
  const float* wf = a.W2 + (size_t)tracer * np;
  float* of = a.Xnew + (size_t)fld * np;
  const float* hXw = second_halves(Xf, wf, lane);
  const float* hUV = second_halves(a.u, a.v, lane);
  const LaneAddr L = lane_addr(lane);
  const unsigned lb = (unsigned)(size_t)lds;
  const bool calm = a.calm_odd && tracer;
  const bool last_lane = lane == 63;
  if (!a.chains_first) __builtin_amdgcn_s_setprio(2); // streaming rows ahead of the chains (which drop to 0 while they sweep)
  int ops = 0;
  unsigned long long gT = 0, gU = 0; // 16 bits per slot: `ops` right after the slot's LDS-DMA was issued
  auto issue_T = [&](int row) {
    const int slot = row & (kRing - 1);
    issue_pair<0>(Xf + row * kNx, wf + row * kNx, hXw + row * kNx, lds + (kRingBase + slot * kSlotB) / 4, lane);
    ops += 3;
    gT = (gT & ~(0xffffull << (16 * slot))) | ((unsigned long long)ops << (16 * slot));
  };
  auto issue_U = [&](int row) {
    const int slot = row & 1;
    issue_pair<0>(a.u + row * kNx, a.v + row * kNx, hUV + row * kNx, lds + (kWindBase + slot * kSlotB) / 4, lane);
    ops += 3;
    gU = (gU & ~(0xffffull << (16 * slot))) | ((unsigned long long)ops << (16 * slot));
  };
  // Rows lo .. hi are read, in order, each once.  The five rows k-2 .. k+2 the meridional stencils of both operators
  // need are a WINDOW IN REGISTERS that slides up one row per step; a row waits for the window in one of kRing LDS
  // slots (up to kRing - 1 rows are in flight ahead of the window), and the slot is refilled as soon as it is read.
  const int lo = k0 >= 2 ? k0 - 2 : 0, hi = k1 + 1 < ny ? k1 + 1 : ny - 1;
  int next_issue = lo;
  for (int j = 0; j < kRing && next_issue <= hi; ++j) issue_T(next_issue++);
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
      const int slot = row & (kRing - 1);
      // (mid-strip the row was requested four steps ago: 2 + 3 x 8 + 3 operations since)
      wait_all_but_mostly<29>(ops - (int)((gT >> (16 * slot)) & 0xffff));
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
  const bool stamp_last = a.stamps && blockIdx.x == (gridDim.x * 3) / 4 && lane == 0; // a task three quarters down the launch order
  if (stamp_last) { a.stamps[8] = __builtin_amdgcn_s_memtime(); a.stamps[10] = (unsigned long long)(k1 - k0); }
  unsigned long long phase_t = 0, phase_sum[4] = {0, 0, 0, 0};
#endif
  for (int c = lo - 2; c < k0; ++c) advance(c); // fill: after this the window is centred on row k0 - 1
#ifdef GREB_TUNING
  if (stamp_last) a.stamps[11] = __builtin_amdgcn_s_memtime();
#endif

  // the winds of row r are read into registers one step early and the slot refilled at once: a row's wind is requested
  // two steps before it is used (requested one step ahead it was not there yet: 4 700 cycles per row instead of ~2 000)
  // the row constants come from global memory through the scalar cache: requested one row ahead (asked for where they
  // are used, each row waited 300-600 cycles for them)
  const float ccy_dif = tab.dif_ccy, ccy_adv = tab.adv_ccy;
  int t2d_n = tab.dif_time2[k0], t2a_n = tab.adv_time2[k0];
  float ccd_n = tab.dif_ccx2[k0], cca_n = tab.adv_ccx2[k0];
  float u[6], v[6];
  if (k0 + 1 < k1) issue_U(k0 + 1);
  wait_all_but(ops - (int)((gU >> (16 * (k0 & 1))) & 0xffff));
  read_pair(L, lb + kWindBase + (k0 & 1) * kSlotB, u, v);
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
#pragma unroll
      for (int j = 0; j < 6; ++j) { Tc[3 + j] = T0[j]; wc[3 + j] = w0[j]; }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Tc[j] = wave_from_prev(T0[3 + j]); Tc[9 + j] = wave_from_next(T0[j]);
        wc[j] = wave_from_prev(w0[3 + j]); wc[9 + j] = wave_from_next(w0[j]);
      }
      const float u0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float T2[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) T2[j] = Tc[j];
      GREB_STEP_STAMP(2);
      if (STRICT || t2d > 1) {
        if (!a.chains_first) __builtin_amdgcn_s_setprio(0);
        chain_window<STRICT, 6>(Tc, wc, u0, ccd, t2d, false, (int)lane, a.chains_first != 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) Td[j] = Tc[3 + j];
      }
      GREB_STEP_STAMP(3);
      if (STRICT || t2a > 1) {
        if (!a.chains_first) __builtin_amdgcn_s_setprio(0);
        chain_window<STRICT, 6>(T2, wc, u, cca, t2a, true, (int)lane, a.chains_first != 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) Ta[j] = T2[3 + j];
      }
    }
    if (!a.chains_first) __builtin_amdgcn_s_setprio(2);
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
    vfloat4 q0, q1;
    transpose_out(L, lb + kOutBase, o, q0, q1);
    float* row = of + r * kNx;
    *reinterpret_cast<vfloat4*>(row + 4 * lane) = q0;
    if (lane < 32) *reinterpret_cast<vfloat4*>(row + 256 + 4 * lane) = q1;
    order_fence();
    ops += 2;
    GREB_STEP_STAMP(5);
    GREB_STEP_PHASE(3);
    if (r + 1 < k1) { // the next row's winds (requested at the start of the previous step: 13 operations since, mid-strip)
      wait_all_but_mostly<13>(ops - (int)((gU >> (16 * ((r + 1) & 1))) & 0xffff));
      read_pair(L, lb + kWindBase + ((r + 1) & 1) * kSlotB, u, v);
    }
    GREB_STEP_PHASE(4);
  }
#ifdef GREB_TUNING
  if (stamp_last) {
    a.stamps[9] = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 4; ++i) a.stamps[12 + i] = phase_sum[i];
  }
  if (a.timeline && threadIdx.x == 0) a.timeline[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// What a row costs, in cycles (tools/stamp_step_rows.py, tools/step_timeline.py; profiles/r03_g384_substep_*):
//   issue  the issue slots it takes on its SIMD -- one instruction per 4 cycles, shared by the SIMD's two wavefronts:
//          ~480 instructions for a streamed row, 36 per chain sweep (33 without the clamp minimum), ~210 to set a chain up;
//   wall   what the row takes a wavefront that has the SIMD to itself: a streamed row waits for memory (3 500 cycles for
//          1 900 of issue -- 2 200 as it shares the SIMD, the figure used), a chain never waits.
// Two strips on one SIMD end after max(their walls, the sum of their issues): measured 94 000 cycles for a polar strip of
// 63 500 beside a 16-row streaming strip (30 400 of issue), 61 000-70 000 for two such streaming strips.
struct RowCost { int issue, wall; };
constexpr int kRowIssue = 2200, kRowWall = 3500, kSweepCycles = 147, kChainSetupCycles = 850, kFillIssue = 700, kFillWall = 3800;
RowCost step_row_cost(const RowTables& t, int k) {
  const int d = t.dif_time2[k], a = t.adv_time2[k];
  static const int row_issue = tuning_int("GREB_STEP_ROWCOST", kRowIssue); // -DGREB_TUNING builds only
  const int chains = (d > 1 ? kChainSetupCycles + kSweepCycles * d : 0) + (a > 1 ? kChainSetupCycles + kSweepCycles * a : 0);
  return {row_issue + chains, kRowWall + chains};
}

} // namespace

bool step_rows_supported(const RowTables* tabs, int n_tabs, int nx, int ny) {
  if (nx != rows::kNx || ny < 5 || ny > kMaxNy) return false;
  for (int t = 0; t < n_tabs; ++t)
    for (int k = 0; k < ny; ++k)
      if (!tabs[t].subcycled[k] || tabs[t].dif_time2[k] < 1 || tabs[t].adv_time2[k] < 1) return false;
  return true;
}

// The launch order of one sub-step.  The chip has slots / 2 SIMDs with two wavefront slots each (187 VGPRs, 19.5 KB of
// LDS per wavefront); workgroup i of a launch lands on SIMD i mod (slots / 2), so tasks i and i + slots / 2 share one.
// A launch is as long as its longest SIMD, and a task started late -- because there are more tasks than slots -- runs
// its full length after the others are done (62 members as 2 388 tasks for 2 048 slots: 40 us, 13 of them for the 340
// late strips).  So ONE round: the rows of all fields are cut into at most `n_slots` strips such that a SIMD's pair ends
// after S cycles -- each strip at most S / 2 of issue and S of wall (RowCost) -- with S the smallest that fits, but no
// less than the dearest row's wall (the 232-sweep polar row: few fields gain nothing from strips that end before it).
// With n tasks for n_simd SIMDs, n - n_simd SIMDs hold a pair: the strips with the most issue run alone, the others are
// paired dearest with cheapest (two chain strips on one SIMD -- both issue without a pause -- take twice as long each).
void step_rows_tasks(const RowTables* tabs, const int* tab_index, int n_members, int ny, int n_slots,
                     std::vector<RowsTask>& tasks) {
  struct T { int field, k0, k1; long long issue, wall; };
  static const int forced = tuning_int("GREB_STEP_TARGET", 0);        // -DGREB_TUNING builds only: S in cycles
  static const int issue_pct = tuning_int("GREB_STEP_ISSUE_PCT", 54); // ... a strip's share of S in issue
  static const int wall_pct = tuning_int("GREB_STEP_WALL_PCT", 70);   // ... and in wall time (measured: 1 member 19.1 us per launch at 85-100, 18.5 at 60-70)
  const int n_simd = std::max(1, n_slots / 2);
  long long total = 0, dearest = 0;
  for (int m = 0; m < n_members; ++m)
    for (int k = 0; k < ny; ++k) {
      const RowCost c = step_row_cost(tabs[tab_index[m]], k);
      total += 2 * c.issue;
      dearest = std::max<long long>(dearest, c.wall);
    }
  long long S = std::max(total / n_simd, dearest + kFillWall);
  if (forced) S = forced;
  std::vector<T> all;
  for (int pass = 0; pass < 96; ++pass) {
    const long long cap_issue = S * issue_pct / 100, cap_wall = S * wall_pct / 100;
    all.clear();
    for (int m = 0; m < n_members; ++m) {
      const RowTables& t = tabs[tab_index[m]];
      // this member's fields: as few strips as the two caps allow, cut where the cumulative issue crosses equal shares
      // (a greedy cut leaves every strip some way below its cap: more strips, or a larger S, than needed)
      long long fi = 0, fw = 0;
      for (int k = 0; k < ny; ++k) { const RowCost c = step_row_cost(t, k); fi += c.issue; fw += c.wall; }
      const long long ci = std::max<long long>(1, cap_issue - kFillIssue), cw = std::max<long long>(1, cap_wall - kFillWall);
      const int n = (int)std::min<long long>(ny, std::max((fi + ci - 1) / ci, (fw + cw - 1) / cw));
      std::vector<T> mine;
      long long acc = 0, issue = kFillIssue, wall = kFillWall;
      int start = 0, cut = 1;
      for (int k = 0; k < ny; ++k) {
        const RowCost c = step_row_cost(t, k);
        // the share boundary cut * fi / n lies nearer the start of row k than its end: close the strip before it
        if (k > start && cut < n && 2 * n * acc + (long long)n * c.issue >= 2 * fi * cut) {
          mine.push_back({0, start, k, issue, wall});
          start = k; issue = kFillIssue; wall = kFillWall;
          while (cut < n && 2 * n * acc + (long long)n * c.issue >= 2 * fi * cut) ++cut; // (a dear row may span shares)
        }
        acc += c.issue; issue += c.issue; wall += c.wall;
      }
      mine.push_back({0, start, ny, issue, wall});
      for (int tr = 0; tr < 2; ++tr)
        for (T x : mine) { x.field = 2 * m + tr; all.push_back(x); }
    }
    if (forced || (int)all.size() <= 2 * n_simd) break;
    S += S / 40;
  }
  std::stable_sort(all.begin(), all.end(), [](const T& x, const T& y) { return x.issue > y.issue; });
  const int n_all = (int)all.size();
  if (n_all > n_simd && n_all <= 2 * n_simd) {
    const int m = n_all - n_simd, alone = n_simd - m; // m SIMDs hold a pair
    std::vector<T> order((size_t)n_all);
    for (int j = 0; j < m; ++j) {
      order[(size_t)j] = all[(size_t)(alone + j)];                // the dearer of pair j ...
      order[(size_t)(n_simd + j)] = all[(size_t)(n_all - 1 - j)]; // ... and the cheapest left
    }
    for (int j = 0; j < alone; ++j) order[(size_t)(m + j)] = all[(size_t)j];
    all.swap(order);
  }
  tasks.clear();
  tasks.reserve(all.size());
  for (const T& x : all) tasks.push_back({x.field | (tab_index[x.field >> 1] << kStepFieldBits), x.k0 | (x.k1 << 8) | kRowsUp});
}

// the launch order on the device (owned by the caller: the engine keeps one per member count and frees it with itself)
hipError_t step_rows_make_tasks(const RowTables* tabs_host, const int* tab_index_host, int n_members, int ny,
                                int n_slots, RowsTask** dev, int* n, RowsTask* head) {
  std::vector<RowsTask> host;
  step_rows_tasks(tabs_host, tab_index_host, n_members, ny, n_slots, host);
  for (int i = 0; i < kStepHeadTasks; ++i) head[i] = i < (int)host.size() ? host[(size_t)i] : RowsTask{0, 0};
  hipError_t e = hipMalloc(dev, host.size() * sizeof(RowsTask));
  if (e != hipSuccess) return e;
  if ((e = hipMemcpy(*dev, host.data(), host.size() * sizeof(RowsTask), hipMemcpyHostToDevice)) != hipSuccess) {
    (void)hipFree(*dev);
    *dev = nullptr;
    return e;
  }
  *n = (int)host.size();
  return hipSuccess;
}

#ifdef GREB_TUNING
// diagnostic builds only: six s_memtime stamps of the dearest task of the last launch (tools/stamp_step_rows.py):
// loop entry, own row + wind landed, before the diffusion chain, before the advection chain, after it, row stored
static unsigned long long* g_step_stamps = nullptr;
static unsigned long long* g_step_timeline = nullptr;
static int g_step_timeline_cap = 0;
// [task][start, end] of the LAST launch in 100 MHz ticks, then one hardware id per task; out == null arms it for `capacity` tasks
extern "C" int greb_tuning_step_timeline(unsigned long long* out, int capacity) {
  if (!out) {
    if (g_step_timeline) (void)hipFree(g_step_timeline);
    g_step_timeline = nullptr; g_step_timeline_cap = 0;
    if (capacity <= 0) return 0;
    if (hipMalloc(&g_step_timeline, (size_t)capacity * 3 * sizeof(unsigned long long)) != hipSuccess) return -1;
    g_step_timeline_cap = capacity;
    return hipMemset(g_step_timeline, 0, (size_t)capacity * 3 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
  }
  if (!g_step_timeline || capacity > g_step_timeline_cap) return -1;
  return hipMemcpy(out, g_step_timeline, (size_t)capacity * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
extern "C" int greb_tuning_step_stamps(unsigned long long* out6) {
  if (!out6) { // arm
    if (!g_step_stamps && hipMalloc(&g_step_stamps, 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemset(g_step_stamps, 0, 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
  }
  if (!g_step_stamps) return -1;
  // [0..5] the dearest task (see above); [8], [11], [9] the LAST task (a streaming strip): start, window filled, end; [10] its rows
  return hipMemcpy(out6, g_step_stamps, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_substep_rows(const float* X, const float* W2, const float* u, const float* v, float* Xnew,
                               const RowTables* tabs_dev, const int* tab_index_dev, const RowsTask* tasks, const RowsTask* head_host,
                               int n_tasks, int n_simd, int ny, bool strict, hipStream_t s, bool calm_vapor) {
  // A chain never waits, so at equal priority (it is the older wavefront) it takes every issue slot of its SIMD and the
  // streaming strip beside it -- which needs few slots but a long time, it waits for memory -- stands still until the
  // chain is over.  Where SIMDs are shared (more tasks than SIMDs) the streaming rows therefore issue first and the
  // chains fill what is left: the pair ends after max(wall, sum of issues) instead of their sum.  With a SIMD per task
  // (few fields) the long chains keep their raised priority.
  static const int forced = tuning_int("GREB_STEP_CHAINS_FIRST", -1); // -DGREB_TUNING builds only (A/B)
  const bool chains_first = forced >= 0 ? forced != 0 : n_tasks <= n_simd;
  StepArgs a{X, W2, u, v, Xnew, tabs_dev, tab_index_dev, tasks, {}, ny, calm_vapor ? 1 : 0, chains_first ? 1 : 0, nullptr, nullptr};
  static_assert(sizeof(RowsTask) == sizeof(unsigned long long), "a task is one 8-byte load: field in the low half");
  std::memcpy(a.head, head_host, sizeof(a.head));
#ifdef GREB_TUNING
  a.stamps = g_step_stamps;
  a.timeline = n_tasks <= g_step_timeline_cap ? g_step_timeline : nullptr;
#endif
  auto kern = strict ? step_rows_kernel<true> : step_rows_kernel<false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)n_tasks), dim3(64), kStepLdsB, s, a);
  return hipGetLastError();
}

} // namespace greb
