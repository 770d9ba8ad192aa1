// greb_rows.hip -- the standalone batched diffusion sweep (src/greb.f90:556-723) on 384-wide grids as ROW STRIPS.
//
// At 384x192 every latitude row takes the sub-cycled branch (:651-719) and 34 of the 192 rows iterate (225, 82, 40 ...
// dependent Jacobi sweeps next to the poles): 1 046 row-sweeps of arithmetic against 192 rows of
// traffic.  A band-per-workgroup kernel (greb_kernels.hip: sweep_kernel) holds a band's LDS and three idle waves for as
// long as its longest chain runs: 0.39 of HBM peak, 98 500 vector instructions per field, reads 1.29 x the algorithmic
// bytes (profiles/r03_g384_diffusion_pmc.txt).  Here nothing waits for anything but its own data:
//   * one WAVEFRONT = one task = a strip of consecutive rows of one field; no workgroup barrier anywhere.  The dependent
//     chains -- which issue one instruction per ~5 cycles whatever they share a SIMD with -- run beside streaming strips
//     (rows_tasks below decides the launch order: chain strips interleaved with streaming strips, ever shorter strips
//     for the fields launched last);
//   * a lane owns 6 consecutive longitudes of a row (64 x 6 = 384), the layout of the register-resident chains
//     (greb_chain6.h): the zonal halo is a wave rotate (DPP), never memory;
//   * a row travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: coalesced 16-byte lanes, no VGPRs, three rows ahead
//     of the arithmetic and in flight across a whole chain), is read back 6 floats per lane (ds_read_b64 x 3,
//     conflict-free: 24-byte lane stride), and the results take the reverse way (ds_write_b64 -> ds_read_b128 ->
//     global_store_dwordx4 nt).  The LDS is private to the wave: ordering is s_waitcnt only;
//   * the strip walks along the meridian and carries the edge flux w(k)*(T(k+1)-T(k)) from row to row: each row of T and
//     wz is read once per strip (two halo rows per strip are the only re-reads), the meridional term costs 5
//     instructions a point.
// All LDS traffic is inline asm: the compiler must not know that LDS-DMA and the ds_ reads touch the same bytes, or it
// drains the DMA queue (s_waitcnt vmcnt(0)) in front of every read.  vmcnt is counted by hand (loads, LDS-DMA and
// stores retire in issue order): `ops` numbers every vector-memory operation the wave issues.
// Measured (MI355X, batch 1 024, settled clocks): 0.168-0.176 ms per launch = 5.2-5.4 TB/s = 0.65-0.67 of HBM peak
// (band kernel: 0.29), the chains hidden completely (0.165-0.18 with every row forced to a single sweep); 57 400 vector
// instructions per field; HBM-side traffic 1.03 x algorithmic; STRICT 0.42 ms (band kernel: 0.90).
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "greb_rows.h"

namespace greb {
namespace {

using namespace rows;
constexpr int kRNx = kNx;
#ifndef GREB_ROWS_SLOTS
#define GREB_ROWS_SLOTS 3
#endif
constexpr int kRowsSlots = GREB_ROWS_SLOTS;       // rows in flight or waiting per wavefront
constexpr unsigned kRowsLdsB = kRowB + kRowsSlots * kSlotB; // the output row, then the slots

// Everything the kernel needs to know about the rows travels BY VALUE (kernarg, scalar loads): no device-side table
// whose lifetime or contents a concurrent call could disturb.
struct RowsArgs {
  float ccy;                       // kappa*dt_crcl/dyy**2, :581
  float cc[kMaxNy];                // ccx2 of the row, :654
  int time2[kMaxNy];               // sweeps of the row, :653 (dwords: a 16-bit table is read by VECTOR loads, and the
                                   // wait for one drains the LDS-DMA queue)
};

struct Walk {
  const float* Tf;   // this field's T1
  const float* wf;   // ... wz
  const float* p2;   // per lane: the second halves of the T row (lanes 0-31) and of the wz row (lanes 32-63)
  float* of;         // ... dX
  lfloat* lds;
  unsigned lane;
  LaneAddr L;        // LDS byte addresses of this lane (inside a slot / the output row)
  unsigned lb;       // byte address of the wavefront's LDS
  int row0, dir, m;  // the walk reads rows row0, row0 + dir, ..., row0 + m*dir
  int ops;           // vector-memory operations issued so far
  unsigned long long gend, gend2; // 16 bits per slot: `ops` right after the LDS-DMA of the row now in slot s was issued
                                  // (two words, not an array: a dynamically indexed array is kept in scratch)
  int slot;          // the slot the next row of the walk arrives in (rows go round the slots)
};

template <int AUX>
__device__ __forceinline__ void issue_row(Walk& c, int x, int slot) {
  const int ro = x * kRNx;
  issue_pair<AUX>(c.Tf + ro, c.wf + ro, c.p2 + ro, c.lds + (kRowB + slot * kSlotB) / 4, c.lane); // ends in an order_fence
  c.ops += 3;
  const int sh = 16 * (slot & 3);
  const unsigned long long old = (slot >> 2) ? c.gend2 : c.gend;
  const unsigned long long v = (old & ~(0xffffull << sh)) | ((unsigned long long)c.ops << sh);
  if (slot >> 2) c.gend2 = v; else c.gend = v;
}

// the row in `slot` has landed: all but the `younger` operations issued after its LDS-DMA may still be in flight
__device__ __forceinline__ void wait_row(const Walk& c, int slot) {
  const int younger = c.ops - (int)((((slot >> 2) ? c.gend2 : c.gend) >> (16 * (slot & 3))) & 0xffff);
  // mid-strip: the row was requested kRowsSlots steps ago, a step issues three LDS-DMA operations and two stores
  wait_all_but_mostly<2 + 5 * (kRowsSlots - 1)>(younger);
}

__device__ __forceinline__ void read_row(const Walk& c, int slot, float (&T)[6], float (&w)[6]) {
  read_pair(c.L, c.lb + kRowB + slot * kSlotB, T, w);
}

// six results per lane -> LDS -> sixteen bytes per lane -> HBM (two stores: 64 + 32 quads)
__device__ __forceinline__ void store_row(Walk& c, const float (&o)[6], int k) {
  vfloat4 q0, q1;
  transpose_out(c.L, c.lb, o, q0, q1);
  float* row = c.of + k * kRNx;
  __builtin_nontemporal_store(q0, reinterpret_cast<vfloat4*>(row + 4 * c.lane));
  if (c.lane < 32) __builtin_nontemporal_store(q1, reinterpret_cast<vfloat4*>(row + 256 + 4 * c.lane));
  order_fence(); // the counted waits need the LDS-DMA and the stores in program order (greb_rows.h)
  c.ops += 2;
}

template <bool STRICT>
struct RowState {
  float T[6], w[6];
  float q[6];   // FAST: w(k-1)*(T(k) - T(k-1));  STRICT: T of row k-1
  float wm[6];  // STRICT: wz of row k-1
};

template <bool STRICT, int AUX>
__device__ __forceinline__ void row_step(Walk& c, const RowsArgs& a, RowState<STRICT>& cur, RowState<STRICT>& nxt, int i,
                                         int k0, int k1, int ny, int dbg) {
  const int r = c.row0 + c.dir * i; // the row this step updates; the walk's next row is r + dir
  if (i + 1 <= c.m) {
    wait_row(c, c.slot);
    read_row(c, c.slot, nxt.T, nxt.w);
    if (i + 1 + kRowsSlots <= c.m) issue_row<AUX>(c, r + (1 + kRowsSlots) * c.dir, c.slot);
    c.slot = c.slot + 1 == kRowsSlots ? 0 : c.slot + 1;
  } else { // the walk ends at the edge of the grid: no row beyond it, its weight is zero (:586, :589)
#pragma unroll
    for (int j = 0; j < 6; ++j) { nxt.T[j] = cur.T[j]; nxt.w[j] = 0.f; }
  }
  float P[6];
  if (!STRICT) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float h = nxt.T[j] - cur.T[j];
      P[j] = nxt.w[j] * h;
      nxt.q[j] = cur.w[j] * h;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 6; ++j) { nxt.q[j] = cur.T[j]; nxt.wm[j] = cur.w[j]; }
  }
  if (r < k0 || r >= k1) return; // the strip's halo row: nothing to write
  const int t2 = (dbg & 1) ? 1 : a.time2[r]; // dbg: -DGREB_TUNING timing experiments only (results wrong), 0 otherwise
  const float cc = a.cc[r];
  float T1h[6];
  if (dbg & 2) {
#pragma unroll
    for (int j = 0; j < 6; ++j) T1h[j] = cur.T[j];
  } else if (STRICT || t2 > 1) {
    float Tw[12], ww[12];
    const float u0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 6; ++j) { Tw[3 + j] = cur.T[j]; ww[3 + j] = cur.w[j]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      Tw[j] = wave_from_prev(cur.T[3 + j]); Tw[9 + j] = wave_from_next(cur.T[j]);
      ww[j] = wave_from_prev(cur.w[3 + j]); ww[9 + j] = wave_from_next(cur.w[j]);
    }
    if (dbg & 8) __builtin_amdgcn_s_setprio(0); // experiment: the streaming waves issue ahead of the chains
    chain_window<STRICT, 6>(Tw, ww, u0, cc, t2, false, (int)c.lane, (dbg & 4) != 0);
    if (dbg & 8) __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int j = 0; j < 6; ++j) T1h[j] = Tw[3 + j];
  } else {
    RowFlux f; // one edge-flux sweep of a row that does not iterate (greb_rows.h)
    row_flux(cur.T, cur.w, f);
    dif_sweep_fast(cur.T, f, cc * 0.05f, T1h);
  }
  float o[6];
  if (!STRICT) {
    // meridional term ccy*(w(k-1)*(T(k-1)-T(k)) + w(k+1)*(T(k+1)-T(k))) (:585-590) from the two edge fluxes of the walk:
    // P - q is the sum of the same two rounded products whichever way the strip walks
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float dTx = T1h[j] - cur.T[j]; // fl(fl(T + d) - T), :718
      float g;
      {
#pragma clang fp contract(off)
        g = P[j] - cur.q[j];
      }
      o[j] = cur.w[j] * (dTx + a.ccy * g);
    }
  } else {
#pragma clang fp contract(off)
    const bool up = c.dir > 0; // the walk's previous row is the southern neighbour
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float Ts = up ? cur.q[j] : nxt.T[j], ws = up ? cur.wm[j] : nxt.w[j];
      const float Tn = up ? nxt.T[j] : cur.q[j], wn = up ? nxt.w[j] : cur.wm[j];
      const float dTx = T1h[j] - cur.T[j];
      float dTy; // :585-590
      if (r >= 1 && r <= ny - 2) dTy = a.ccy * (ws * (Ts - cur.T[j]) + wn * (Tn - cur.T[j]));
      else if (r == 0) dTy = a.ccy * wn * (-cur.T[j] + Tn);
      else dTy = a.ccy * ws * (Ts - cur.T[j]);
      o[j] = cur.w[j] * (dTx + dTy); // :721
    }
  }
  store_row(c, o, r);
}

template <bool STRICT, int AUX>
__global__ __launch_bounds__(64) void dif_rows_kernel(const float* __restrict__ T1, const float* __restrict__ wz,
                                                       float* __restrict__ dX, const RowsArgs a,
                                                       const RowsTask* __restrict__ tasks, int batch, int ny, int dbg) {
  extern __shared__ __align__(16) float lds_raw[];
  // the launch order is a table (rows_tasks below): block i does task i
  const RowsTask task = tasks[blockIdx.x];
  const int b = task.field;
  if (b < 0) return; // padding of an incomplete group of eight fields
  Walk c;
  c.lds = (lfloat*)lds_raw;
  c.lane = threadIdx.x;
  const int k0 = task.rows & 0xff, k1 = (task.rows >> 8) & 0x1ff;
  const size_t fo = (size_t)b * kRNx * ny;
  c.Tf = T1 + fo; c.wf = wz + fo; c.of = dX + fo;
  c.p2 = second_halves(c.Tf, c.wf, c.lane);
  c.lb = (unsigned)(size_t)c.lds;
  c.L = lane_addr(c.lane);
  c.ops = 0; c.gend = 0; c.gend2 = 0;
  if (dbg & 8) __builtin_amdgcn_s_setprio(2);
  static_assert(kRowsSlots <= 8, "gend, gend2 hold eight 16-bit counters");
  // the walk: rows lo .. hi (the strip and its halo rows inside the grid), upwards or downwards
  const int lo = k0 > 0 ? k0 - 1 : 0, hi = k1 < ny ? k1 : ny - 1;
  c.dir = (task.rows & kRowsUp) ? 1 : -1;
  c.row0 = c.dir > 0 ? lo : hi;
  c.m = hi - lo;
  const int last_out = c.dir > 0 ? k1 - 1 - lo : hi - k0; // index of the last step that writes a row
#pragma unroll
  for (int j = 0; j < kRowsSlots; ++j)
    if (j <= c.m) issue_row<AUX>(c, c.row0 + j * c.dir, j);
  RowState<STRICT> A, B;
  wait_row(c, 0);
  read_row(c, 0, A.T, A.w);
  if (kRowsSlots <= c.m) issue_row<AUX>(c, c.row0 + kRowsSlots * c.dir, 0);
  c.slot = 1;
#pragma unroll
  for (int j = 0; j < 6; ++j) { A.q[j] = STRICT ? A.T[j] : 0.f; A.wm[j] = 0.f; }
  for (int i = 0; i <= last_out; i += 2) {
    row_step<STRICT, AUX>(c, a, A, B, i, k0, k1, ny, dbg);
    if (i + 1 > last_out) break;
    row_step<STRICT, AUX>(c, a, B, A, i + 1, k0, k1, ny, dbg);
  }
}

int strip_cost(int time2) { return time2 > 1 ? 130 + 36 * time2 : 120; }

struct Strip { int k0, k1, cost, up; };

// contiguous strips of about `target` instructions over rows [ka, kb), never splitting a row; a strip is closed when
// the next row would take it over the target, unless it is still tiny
void cut_strips(const RowTables& t, int ka, int kb, int target, std::vector<Strip>& out) {
  int acc = 0, start = ka;
  for (int k = ka; k < kb; ++k) {
    const int cst = strip_cost(t.dif_time2[k]);
    if (acc > 0 && acc + cst > target && acc >= 600) { out.push_back({start, k, acc, 0}); start = k; acc = 0; }
    acc += cst;
  }
  if (kb > ka) out.push_back({start, kb, acc, 0});
}

} // namespace

// The launch order.  Two kinds of task: CHAIN strips (the rows next to the poles that iterate: arithmetic, a lone
// wavefront issuing one instruction per ~5 cycles) and STREAMING strips (the single-sweep rows between the caps: memory).
//   * they are interleaved, the chain strips spread evenly over the first `chain_span` per cent of the launch: at any
//     moment a SIMD holds about one chain wave beside streaming ones, so the arithmetic hides under the traffic
//     (dearest-first order ran the chains first and the traffic after them: 0.210 ms against 0.190);
//   * the streaming region is cut into few long strips for most fields (halo rows re-read: 2 per strip) and into ever
//     shorter ones for the fields launched last (levels below): when the last task starts, what is still running is
//     small, so the launch does not end on a handful of wavefronts each streaming at its own latency-bound ~3 GB/s;
//   * tasks come in groups of eight (the same strip of eight consecutive fields): blocks are dealt to the eight XCDs in
//     turn, so all strips of a field run on one XCD, and neighbouring strips walk away from their common border (one
//     down, one up): the halo rows both read are requested together and the second reader finds them in that XCD's L2.
// Speed only: any order gives the same result bit for bit, every row is written by exactly one task.
void rows_tasks(const RowTables& t, int ny, int batch, const RowsTuning& tu, std::vector<RowsTask>& tasks) {
  // the streaming region: the run of single-sweep rows around the equator
  int ks = ny / 2, ke = ny / 2;
  while (ks > 0 && t.dif_time2[ks - 1] == 1) --ks;
  while (ke < ny && t.dif_time2[ke] == 1) ++ke;
  if (t.dif_time2[ny / 2] != 1) ks = ke = ny / 2; // (no such run: everything is a chain strip)
  std::vector<Strip> caps;
  cut_strips(t, 0, ks, tu.chain_target, caps);
  cut_strips(t, ke, ny, tu.chain_target, caps);
  for (size_t i = 0; i < caps.size(); ++i) caps[i].up = (int)(i & 1);
  std::stable_sort(caps.begin(), caps.end(), [](const Strip& x, const Strip& y) { return x.cost > y.cost; });
  const int G = (batch + 7) / 8, len = ke - ks;
  // levels of the streaming cut, coarse to fine; the finer levels take the LAST groups of fields
  int parts[4], first[5];
  for (int l = 0; l < 4; ++l) parts[l] = std::max(1, std::min(len, tu.parts[l]));
  first[4] = G;
  for (int l = 3; l >= 1; --l) {
    const int groups = len > 0 ? (tu.level_tasks[l] + 8 * parts[l] - 1) / (8 * parts[l]) : 0;
    first[l] = std::max(0, first[l + 1] - groups);
  }
  first[0] = 0;
  struct Oct { double pos; int group, k0, k1, up; };
  std::vector<Oct> so, co;
  double sw = 0, cw = 0;
  if (len > 0)
    for (int l = 0; l < 4; ++l)
      for (int g = first[l]; g < first[l + 1]; ++g)
        for (int i = 0; i < parts[l]; ++i) {
          const int a0 = ks + (int)((long long)len * i / parts[l]), a1 = ks + (int)((long long)len * (i + 1) / parts[l]);
          so.push_back({sw, g, a0, a1, i & 1});
          sw += a1 - a0 + 2;
        }
  for (int g = 0; g < G; ++g)
    for (const Strip& c : caps) { co.push_back({cw, g, c.k0, c.k1, c.up}); cw += c.cost; }
  for (Oct& o : so) o.pos /= sw > 0 ? sw : 1;
  const double span = so.empty() ? 1.0 : tu.chain_span * 0.01;
  for (Oct& o : co) o.pos *= span / (cw > 0 ? cw : 1);
  std::vector<Oct> all(so.size() + co.size());
  std::merge(co.begin(), co.end(), so.begin(), so.end(), all.begin(), [](const Oct& x, const Oct& y) { return x.pos < y.pos; });
  tasks.clear();
  tasks.reserve(all.size() * 8);
  for (const Oct& o : all)
    for (int f = 0; f < 8; ++f) {
      const int field = 8 * o.group + f;
      tasks.push_back({field < batch ? field : -1, o.k0 | (o.k1 << 8) | (o.up ? kRowsUp : 0)});
    }
}

namespace {
// device copies of launch orders: immutable once made (a call in flight never sees its table change), freed only by
// rows_release_cache()
struct TaskTable {
  int device, ny, batch;
  RowsTuning tu;
  int time2[kMaxNy];
  RowsTask* dev;
  int n;
  unsigned long long used; // when it was last asked for (g_rows_clock)
};
std::mutex g_rows_mu;
std::vector<TaskTable> g_rows_tables;
unsigned long long g_rows_clock = 0;
constexpr size_t kRowsTablesMax = 32; // a long-lived host that sweeps kappa or batch: the least recently used table goes
} // namespace

void rows_release_cache() {
  std::lock_guard<std::mutex> lock(g_rows_mu);
  int prev = 0;
  const bool have_prev = hipGetDevice(&prev) == hipSuccess;
  for (TaskTable& e : g_rows_tables)
    if (e.dev && hipSetDevice(e.device) == hipSuccess) { (void)hipDeviceSynchronize(); (void)hipFree(e.dev); }
  g_rows_tables.clear();
  if (have_prev) (void)hipSetDevice(prev);
}

static hipError_t rows_task_table(const RowTables& t, int ny, int batch, const RowsTuning& tu, const RowsTask** dev, int* n) {
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_rows_mu);
  for (TaskTable& c : g_rows_tables)
    if (c.device == device && c.ny == ny && c.batch == batch && std::memcmp(&c.tu, &tu, sizeof(tu)) == 0 &&
        std::memcmp(c.time2, t.dif_time2, sizeof(int) * ny) == 0) {
      c.used = ++g_rows_clock;
      *dev = c.dev; *n = c.n;
      return hipSuccess;
    }
  if (g_rows_tables.size() >= kRowsTablesMax) { // retire this device's least recently used table -- once the device is idle:
    size_t lru = g_rows_tables.size();           // a sweep in flight on any stream may still be reading it
    for (size_t i = 0; i < g_rows_tables.size(); ++i)
      if (g_rows_tables[i].device == device && (lru == g_rows_tables.size() || g_rows_tables[i].used < g_rows_tables[lru].used)) lru = i;
    if (lru < g_rows_tables.size()) {
      if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
      (void)hipFree(g_rows_tables[lru].dev);
      g_rows_tables.erase(g_rows_tables.begin() + (long)lru);
    }
  }
  std::vector<RowsTask> tasks;
  rows_tasks(t, ny, batch, tu, tasks);
  TaskTable c{};
  c.device = device; c.ny = ny; c.batch = batch; c.tu = tu; c.n = (int)tasks.size();
  std::memcpy(c.time2, t.dif_time2, sizeof(int) * ny);
  if ((e = hipMalloc(&c.dev, tasks.size() * sizeof(RowsTask))) != hipSuccess) return e;
  if ((e = hipMemcpy(c.dev, tasks.data(), tasks.size() * sizeof(RowsTask), hipMemcpyHostToDevice)) != hipSuccess) {
    (void)hipFree(c.dev);
    return e;
  }
  c.used = ++g_rows_clock;
  g_rows_tables.push_back(c);
  *dev = c.dev; *n = c.n;
  return hipSuccess;
}

bool rows_supported(const RowTables& t, int nx, int ny) {
  if (nx != kRNx || ny < 3 || ny > kMaxNy) return false;
  for (int k = 0; k < ny; ++k)
    if (!t.subcycled[k] || t.dif_time2[k] < 1 || t.dif_time2[k] > (1 << 20)) return false;
  return true;
}

hipError_t launch_diffusion_rows(const float* T1, const float* wz, float* dX, const RowTables& t, int ny, int batch,
                                 bool strict, hipStream_t s) {
  RowsArgs a;
  std::memset(&a, 0, sizeof(a));
  a.ccy = t.dif_ccy;
  for (int k = 0; k < ny; ++k) { a.cc[k] = t.dif_ccx2[k]; a.time2[k] = t.dif_time2[k]; }
  // (-DGREB_TUNING builds read these from the environment; the release library has the constants)
  static const RowsTuning tu = [] {
    RowsTuning r = rows_default_tuning();
    r.chain_target = tuning_int("GREB_ROWS_TARGET", r.chain_target);
    r.chain_span = tuning_int("GREB_ROWS_SPAN", r.chain_span);
    r.parts[0] = tuning_int("GREB_ROWS_P0", r.parts[0]); r.parts[1] = tuning_int("GREB_ROWS_P1", r.parts[1]);
    r.parts[2] = tuning_int("GREB_ROWS_P2", r.parts[2]); r.parts[3] = tuning_int("GREB_ROWS_P3", r.parts[3]);
    r.level_tasks[1] = tuning_int("GREB_ROWS_L1", r.level_tasks[1]); r.level_tasks[2] = tuning_int("GREB_ROWS_L2", r.level_tasks[2]);
    r.level_tasks[3] = tuning_int("GREB_ROWS_L3", r.level_tasks[3]);
    return r;
  }();
  const RowsTask* tasks = nullptr;
  int n = 0;
  hipError_t e = rows_task_table(t, ny, batch, tu, &tasks, &n);
  if (e != hipSuccess) return e;
  static const int aux = tuning_int("GREB_ROWS_NT", 0);
  static const int dbg = tuning_int("GREB_DEBUG_ROWS", 0);
  static const int lds_pad = tuning_int("GREB_ROWS_LDS_PAD", 0); // occupancy experiments
  void (*kern)(const float*, const float*, float*, const RowsArgs, const RowsTask*, int, int, int);
  if (strict) kern = dif_rows_kernel<true, 0>;
  else kern = aux ? dif_rows_kernel<false, 2> : dif_rows_kernel<false, 0>;
  hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(64), kRowsLdsB + lds_pad, s, T1, wz, dX, a, tasks, batch, ny, dbg);
  return hipGetLastError();
}

} // namespace greb
