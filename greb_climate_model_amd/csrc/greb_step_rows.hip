// greb_step_rows.hip -- ONE circulation sub-step X <- (X + dX_diffuse) + dX_advec (src/greb.f90:549, with :556-723 and
// :726-915 behind it) of the any-grid engine on 384- and 192-wide grids per LAUNCH, as row strips: one wavefront = one
// task = a strip of consecutive latitude rows of one (member, tracer) field (greb_step_strip.h has the strip itself and is
// shared with greb_circ_rows.hip, which runs all 24 sub-steps of a circulation call in one launch).  No workgroup, no
// barrier; the kernel boundary orders the sub-steps, so nothing here waits for another wavefront: this is the form the
// engine takes where a one-launch call cannot be resident as a whole (greb_engine.cpp: the slot ledger), where
// GREB_F_NO_PERSISTENT asks for it, or where the engine's own trial finds it faster.
//
// With few fields the launch is the 232-sweep polar row (225 diffusion + 7 advection sweeps, SURVEY.md App. B: 138-147
// cycles per sweep, greb_chain6.h) plus everything in front of it -- arguments, task, the first rows from memory another
// XCD wrote: 18.6-18.8 us per launch for one member (the one-launch call: 15.3 us per sub-step; round 2's band kernels:
// 25.1).  With many fields it is bound by instruction issue (one vector instruction per SIMD every 4 cycles) and by how
// evenly the SIMDs are loaded: step_rows_tasks below builds the launch order, ONE round of at most as many tasks as the
// chip has wavefront slots.
// STRICT keeps the reference's expression trees (bit-exact), FAST the re-associated ones of the other kernels.
#include <algorithm>
#include <cstring>
#include <vector>

#include "greb_step_order.h"
#include "greb_step_strip.h"

namespace greb {
namespace {
using namespace rows;

typedef const __attribute__((address_space(4))) unsigned long long* cull_ptr; // scalar-loadable: immutable during a launch

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
// (A/B of compile-time choices, tools/build_variant.sh: -DGREB_STEP_RING=3 -DGREB_STEP_WAVES=3: three landing slots and
// 168 VGPRs, nine wavefronts per CU instead of eight)
#ifndef GREB_STEP_RING
#define GREB_STEP_RING 4
#endif
#ifndef GREB_STEP_WAVES
#define GREB_STEP_WAVES 1
#endif
constexpr int kStepRing = GREB_STEP_RING;
constexpr unsigned kStepRowsLdsB = kRowB + (kStepRing + 2) * kSlotB;
template <bool STRICT, int NXR>
__global__ __launch_bounds__(64, GREB_STEP_WAVES) void step_rows_kernel(const StepArgs a) {
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
  const int tracer = fld & 1;
  // the row table through the constant address space (greb_step_strip.h: crow_tables)
  const crow_tables& tab = *(const crow_tables*)(a.tabs + tab_idx); // (index in the task word: one dependent latency less)
  const size_t np = (size_t)NXR * ny;
  const StripIo io{a.X + (size_t)fld * np, a.W2 + (size_t)tracer * np, a.Xnew + (size_t)fld * np, a.u, a.v};
  StripStamps st{nullptr, nullptr};
#ifdef GREB_TUNING
  if (a.stamps && blockIdx.x == 0) st.first = a.stamps;
  if (a.stamps && blockIdx.x == (gridDim.x * 3) / 4) st.phases = a.stamps; // a task three quarters down the launch order
#endif
  stream_strip<STRICT, kAuxPlain, false, NXR, kStepRing>(lds, io, tab, k0, k1, ny, a.calm_odd && tracer, a.chains_first, lane, st);
#ifdef GREB_TUNING
  if (a.timeline && threadIdx.x == 0) a.timeline[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

} // namespace

bool step_rows_supported(const RowTables* tabs, int n_tabs, int nx, int ny) {
  if ((nx != rows::kNx && 2 * nx != rows::kNx) || ny < 5 || ny > kMaxNy) return false; // 384 longitudes, or 192 laid twice round the wavefront
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
  typedef Strip T;
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
      std::vector<T> mine;
      cut_rows(t, 0, ny, cap_issue, cap_wall, mine);
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
                               int n_tasks, int n_simd, int nx, int ny, bool strict, hipStream_t s, bool calm_vapor) {
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
  auto kern = nx == kNx ? (strict ? step_rows_kernel<true, kNx> : step_rows_kernel<false, kNx>)
                        : (strict ? step_rows_kernel<true, kNx / 2> : step_rows_kernel<false, kNx / 2>);
  hipLaunchKernelGGL(kern, dim3((unsigned)n_tasks), dim3(64), kStepRowsLdsB, s, a);
  return hipGetLastError();
}

} // namespace greb
