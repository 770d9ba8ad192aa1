// greb_step_rows.hip -- one circulation sub-step X <- (X + dX_diffuse) + dX_advec (src/greb.f90:549, with :556-723 and
// :726-915 behind it) of the any-grid engine on 384-wide grids, as ROW STRIPS: the engine-side twin of greb_rows.hip.
//
// One wavefront = one task = a strip of consecutive latitude rows of one (member, tracer) field; no workgroup, no
// barrier: a wave that owns a 232-sweep polar row (225 diffusion + 7 advection sweeps) starts its chain
// as soon as ITS row and wind have landed and nobody else waits for it.  That chain is the length of the launch
// (one wavefront issues one instruction per ~5 cycles: 232 x 36 x 5 cycles = 17.4 us); the band kernels around it
// (greb_kernels.hip: sweep_kernel<fused>, greb_pair_sweep.hip) add staging, a workgroup barrier and the band's
// epilogue to it: 24.4 us per launch for one member, 48 us for 62.
//   * rows k-2 .. k+2 of the tracer and its weight (the meridional stencils of both operators) live in a ring of LDS
//     slots filled by LDS-DMA three rows ahead (greb_rows.h); the zonal halo is a wave rotate (DPP);
//   * the winds of the row travel the same way (a ring of two);
//   * per row: the zonal edge fluxes once, shared by the diffusion and the advection sweep (greb_device.h: edge-flux
//     form); rows that iterate run their sweeps in registers (greb_chain6.h), diffusion and advection chains one after
//     the other in the same wave;
//   * tasks are launched dearest first: the long chains start in the first microsecond.
// STRICT keeps the reference's expression trees (bit-exact), FAST the re-associated ones of the other kernels.
#include <algorithm>
#include <vector>

#include "greb_rows.h"

namespace greb {
namespace {
using namespace rows;

constexpr int kRing = 8;   // tracer/weight rows resident per wavefront: k-2 .. k+2 and three ahead
constexpr int kAhead = 3;
constexpr unsigned kOutBase = 0, kRingBase = kRowB, kWindBase = kRowB + kRing * kSlotB;
constexpr unsigned kStepLdsB = kWindBase + 2 * kSlotB; // 31.5 KB: five wavefronts per CU

struct StepArgs {
  const float* X;          // [n_members][2][ny][nx]  {Tair, q}
  const float* W2;         // [2][ny][nx]             {wz_air, wz_vapor}
  const float* u;          // [ny][nx] winds of the step, shared by every member
  const float* v;
  float* Xnew;
  const RowTables* tabs;
  const int* tab_index;    // [n_members]
  const RowsTask* tasks;   // field = 2 * member + tracer
  int ny, calm_odd;        // calm_odd: the vapour fields see zero wind (greb.original.model.f90:560-564)
  unsigned long long* stamps; // -DGREB_TUNING builds only (null otherwise): s_memtime stamps of task 0
};
#ifdef GREB_TUNING
#define GREB_STEP_STAMP(i) if (a.stamps && blockIdx.x == 0 && r == k0 && lane == 0) a.stamps[i] = __builtin_amdgcn_s_memtime()
#else
#define GREB_STEP_STAMP(i)
#endif

struct Ring { // 16 bits per slot: `ops` right after the slot's LDS-DMA was issued (two words, never an indexed array:
              // that would live in scratch)
  unsigned long long g0, g1;
  __device__ __forceinline__ void set(int slot, int ops) {
    const int sh = 16 * (slot & 3);
    const unsigned long long old = (slot >> 2) ? g1 : g0;
    const unsigned long long v = (old & ~(0xffffull << sh)) | ((unsigned long long)ops << sh);
    if (slot >> 2) g1 = v; else g0 = v;
  }
  __device__ __forceinline__ int get(int slot) const { return (int)(((slot >> 2) ? g1 : g0) >> (16 * (slot & 3))) & 0xffff; }
};

template <bool STRICT>
__global__ __launch_bounds__(64) void step_rows_kernel(const StepArgs a) {
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  const RowsTask task = a.tasks[blockIdx.x];
  const int fld = task.field;
  if (fld < 0) return;
  const int k0 = task.rows & 0xff, k1 = (task.rows >> 8) & 0x1ff, ny = a.ny;
  const unsigned lane = threadIdx.x;
  const int member = fld >> 1, tracer = fld & 1;
  const RowTables& tab = a.tabs[a.tab_index[member]];
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
  int ops = 0;
  Ring gT{0, 0};
  unsigned long long gU = 0;
  auto issue_T = [&](int row) {
    const int slot = row & (kRing - 1);
    issue_pair<0>(Xf + row * kNx, wf + row * kNx, hXw + row * kNx, lds + (kRingBase + slot * kSlotB) / 4, lane);
    ops += 3;
    gT.set(slot, ops);
  };
  auto issue_U = [&](int row) {
    const int slot = row & 1;
    issue_pair<0>(a.u + row * kNx, a.v + row * kNx, hUV + row * kNx, lds + (kWindBase + slot * kSlotB) / 4, lane);
    ops += 3;
    gU = (gU & ~(0xffffull << (16 * slot))) | ((unsigned long long)ops << (16 * slot));
  };
  // rows lo .. hi are read; the first row's own data and wind go first, its neighbours after them
  const int lo = k0 >= 2 ? k0 - 2 : 0, hi = k1 + 1 < ny ? k1 + 1 : ny - 1;
  issue_T(k0);
  issue_U(k0);
  int issued = k0 + 2 + kAhead < hi ? k0 + 2 + kAhead : hi; // highest row issued
  for (int r = lo; r <= issued; ++r)
    if (r != k0) issue_T(r);
  const int ops_prologue = ops;

  for (int r = k0; r < k1; ++r) {
    GREB_STEP_STAMP(0);
    if (r + 1 < k1) issue_U(r + 1);
    if (r > k0 && r + 2 + kAhead <= hi) { issued = r + 2 + kAhead; issue_T(issued); }
    // ---- this row and its wind
    {
      const int yT = ops - gT.get(r & (kRing - 1)), yU = ops - (int)((gU >> (16 * (r & 1))) & 0xffff);
      wait_all_but(yT < yU ? yT : yU);
    }
    GREB_STEP_STAMP(1);
    float T0[6], w0[6], u[6], v[6];
    read_pair(L, lb + kRingBase + (r & (kRing - 1)) * kSlotB, T0, w0);
    read_pair(L, lb + kWindBase + (r & 1) * kSlotB, u, v);
    if (calm) {
#pragma unroll
      for (int j = 0; j < 6; ++j) { u[j] = 0.f; v[j] = 0.f; }
    }
    const int t2d = tab.dif_time2[r], t2a = tab.adv_time2[r];
    const float ccd = tab.dif_ccx2[r], cca = tab.adv_ccx2[r];
    // ---- zonal part: the two sub-cycled results T1h (:656-717, :842-909)
    float Td[6], Ta[6];
    if (STRICT || t2d > 1 || t2a > 1) {
      float Tw[12], ww[12];
#pragma unroll
      for (int j = 0; j < 6; ++j) { Tw[3 + j] = T0[j]; ww[3 + j] = w0[j]; }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Tw[j] = wave_from_prev(T0[3 + j]); Tw[9 + j] = wave_from_next(T0[j]);
        ww[j] = wave_from_prev(w0[3 + j]); ww[9 + j] = wave_from_next(w0[j]);
      }
      const float u0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float T2[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) T2[j] = Tw[j];
      GREB_STEP_STAMP(2);
      if (STRICT || t2d > 1) {
        chain_window<STRICT, 6>(Tw, ww, u0, ccd, t2d, false, (int)lane);
#pragma unroll
        for (int j = 0; j < 6; ++j) Td[j] = Tw[3 + j];
      }
      GREB_STEP_STAMP(3);
      if (STRICT || t2a > 1) {
        chain_window<STRICT, 6>(T2, ww, u, cca, t2a, true, (int)lane);
#pragma unroll
        for (int j = 0; j < 6; ++j) Ta[j] = T2[3 + j];
      }
    }
    if (!STRICT && (t2d <= 1 || t2a <= 1)) {
      RowFlux f;
      row_flux(T0, w0, f);
      if (t2d <= 1) dif_sweep_fast(T0, f, ccd * 0.05f, Td);
      if (t2a <= 1) adv_sweep_fast(T0, u, f, cca * 0.05f, last_lane, Ta);
    }
    GREB_STEP_STAMP(4);
    // ---- meridional part: rows k-2 .. k+2 (rows outside the grid: weight zero)
    if (r == k0) wait_all_but(ops - ops_prologue);
    else if (r + 2 <= hi) wait_all_but(ops - gT.get((r + 2) & (kRing - 1)));
    float Tm1[6], wm1[6], Tp1[6], wp1[6], Tm2[6], wm2[6], Tp2[6], wp2[6];
    auto neighbour = [&](int row, float (&Tn)[6], float (&wn)[6]) {
      if (row >= 0 && row < ny) {
        read_pair(L, lb + kRingBase + (row & (kRing - 1)) * kSlotB, Tn, wn);
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) { Tn[j] = T0[j]; wn[j] = 0.f; }
      }
    };
    neighbour(r - 1, Tm1, wm1); neighbour(r + 1, Tp1, wp1);
    neighbour(r - 2, Tm2, wm2); neighbour(r + 2, Tp2, wp2);
    float o[6];
    if (STRICT) {
#pragma clang fp contract(off)
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float dyd = dif_lat_point_strict<float>(T0[j], Tm1[j], Tp1[j], wm1[j], wp1[j], tab.dif_ccy, r, ny);
        const float dya = adv_lat_point_strict<float>(T0[j], Tm2[j], Tm1[j], Tp1[j], Tp2[j], wm2[j], wm1[j], wp1[j], wp2[j], v[j],
                                                      tab.adv_ccy, r, ny);
        const float dd = w0[j] * ((Td[j] - T0[j]) + dyd); // :718, :721
        const float da = (Ta[j] - T0[j]) + dya;           // :910, :913
        o[j] = T0[j] + dd + da;                           // :549
      }
    } else {
      float am, ap;
      adv_lat_coef(tab.adv_ccy, r, ny, am, ap);
      const float ccyd = tab.dif_ccy;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float gm1 = wm1[j] * (Tm1[j] - T0[j]), gp1 = wp1[j] * (Tp1[j] - T0[j]);
        const float dm2 = wm2[j] * (T0[j] - Tm2[j]), dp2 = wp2[j] * (T0[j] - Tp2[j]);
        const float dyd = ccyd * (gm1 + gp1);
        const float dya = ap * split_p(v[j]) * (dp2 - gp1) - am * split_m(v[j]) * (dm2 - gm1);
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
    ops += 2;
    GREB_STEP_STAMP(5);
  }
}

int step_row_cost(const RowTables& t, int k) {
  const int d = t.dif_time2[k], a = t.adv_time2[k];
  return 330 + (d > 1 ? 110 + 36 * d : 0) + (a > 1 ? 110 + 36 * a : 0);
}

} // namespace

bool step_rows_supported(const RowTables* tabs, int n_tabs, int nx, int ny) {
  if (nx != rows::kNx || ny < 5 || ny > kMaxNy) return false;
  for (int t = 0; t < n_tabs; ++t)
    for (int k = 0; k < ny; ++k)
      if (!tabs[t].subcycled[k] || tabs[t].dif_time2[k] < 1 || tabs[t].adv_time2[k] < 1) return false;
  return true;
}

// The launch order of one sub-step: per field the rows are cut into strips of about `target` instructions (a row is never
// split), the strips of all fields are launched dearest first -- the long polar chains start in the first microsecond
// and set the length of the launch, everything else fills the other SIMDs beside them.  The target follows the member
// count: few members are cut fine (every CU gets something to do), many members coarse (fewer halo rows re-read).
void step_rows_tasks(const RowTables* tabs, const int* tab_index, int n_members, int ny, std::vector<RowsTask>& tasks) {
  struct T { int field, k0, k1, cost; };
  std::vector<T> all;
  const int n_fields = 2 * n_members;
  static const int forced = tuning_int("GREB_STEP_TARGET", 0); // -DGREB_TUNING builds only
  const int target = forced ? forced : (n_fields <= 8 ? 1400 : (n_fields <= 48 ? 2400 : 6000));
  for (int m = 0; m < n_members; ++m) {
    const RowTables& t = tabs[tab_index[m]];
    int acc = 0, start = 0;
    std::vector<T> mine;
    for (int k = 0; k < ny; ++k) {
      const int c = step_row_cost(t, k);
      if (acc > 0 && acc + c > target) { mine.push_back({0, start, k, acc}); start = k; acc = 0; }
      acc += c;
    }
    mine.push_back({0, start, ny, acc});
    for (int tr = 0; tr < 2; ++tr)
      for (T x : mine) { x.field = 2 * m + tr; all.push_back(x); }
  }
  std::stable_sort(all.begin(), all.end(), [](const T& x, const T& y) { return x.cost > y.cost; });
  tasks.clear();
  tasks.reserve(all.size());
  for (const T& x : all) tasks.push_back({x.field, x.k0 | (x.k1 << 8) | kRowsUp});
}

// the launch order on the device (owned by the caller: the engine keeps one per member count and frees it with itself)
hipError_t step_rows_make_tasks(const RowTables* tabs_host, const int* tab_index_host, int n_members, int ny,
                                RowsTask** dev, int* n) {
  std::vector<RowsTask> host;
  step_rows_tasks(tabs_host, tab_index_host, n_members, ny, host);
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
extern "C" int greb_tuning_step_stamps(unsigned long long* out6) {
  if (!out6) { // arm
    if (!g_step_stamps && hipMalloc(&g_step_stamps, 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemset(g_step_stamps, 0, 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
  }
  if (!g_step_stamps) return -1;
  return hipMemcpy(out6, g_step_stamps, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_substep_rows(const float* X, const float* W2, const float* u, const float* v, float* Xnew,
                               const RowTables* tabs_dev, const int* tab_index_dev, const RowsTask* tasks, int n_tasks,
                               int ny, bool strict, hipStream_t s, bool calm_vapor) {
  StepArgs a{X, W2, u, v, Xnew, tabs_dev, tab_index_dev, tasks, ny, calm_vapor ? 1 : 0, nullptr};
#ifdef GREB_TUNING
  a.stamps = g_step_stamps;
#endif
  auto kern = strict ? step_rows_kernel<true> : step_rows_kernel<false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)n_tasks), dim3(64), kStepLdsB, s, a);
  return hipGetLastError();
}

} // namespace greb
