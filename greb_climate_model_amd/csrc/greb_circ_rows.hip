// greb_circ_rows.hip -- a whole circulation CALL on a 384-wide grid in ONE launch: the 24 sub-steps
// X <- (X + dX_diffuse) + dX_advec of src/greb.f90:546-550 run inside the kernel, as the reference runs them inside one
// call of `circulation`.  (greb_step_rows.hip launches once per sub-step: 18 250 launches per member-year, each with its
// own prologue -- arguments, task, first rows from memory another XCD wrote -- in front of the 232-sweep polar chain that
// is the length of the launch when the fields are few.)
//
// One wavefront = one task for the whole call, and two kinds of task:
//   strip  rows [k0, k1) of one (member, tracer) field, every sub-step the machinery of greb_step_strip.h;
//   chain  ONE row whose zonal diffusion chain is long (>= kChainTaskMinSweeps dependent sweeps: rows 1, 2, 189, 190 of
//          the 384x192 grid at the default diffusivity).  The row, its weights, its winds and the coefficient form of
//          both chains (greb_chain6.h) stay in REGISTERS from sub-step to sub-step: the next chain starts from the row
//          just computed, not from memory.  Its four meridional neighbours (rows r-2 .. r+2 at the sub-step's start)
//          are fetched into LDS one sub-step AHEAD, while the chain runs -- their owners finish a sub-step long before a
//          232-sweep chain does.  A sub-step of such a task is its arithmetic and nothing else.
// Sub-step s reads X[s & 1] and writes X[(s + 1) & 1].  A task may start sub-step s >= 1 when the tasks that own rows
// k0-2, k0-1, k1, k1+1 of its field have completed s - 1: they have then written what it reads, and have read what it is
// about to overwrite.  Each task counts its completed sub-steps in a flag word of its own.
//
// Memory order (MI355X: the L2 of an XCD is not coherent with the others', a CU's L1 is never refreshed by another CU):
// every tracer row is stored write-through at agent scope (`global_store_dwordx4 ... sc1`), the storing wavefront waits
// `s_waitcnt vmcnt(0)`, then one lane stores the flag `sc1`; a reader polls the flag with `sc1` loads and reads the rows
// with `sc1` LDS-DMA (L1 bypassed).  Nothing in the data path is a fence.  The weights, the winds, the row tables and the
// task list are immutable during a launch and read the ordinary way.
//
// Co-residency is the premise of any in-kernel wait and is never assumed: the host builds at most as many tasks as the
// wavefront slots it was given (launch_circulation_rows checks it again), the engine keeps a ledger of the slots of a
// device (greb_engine.cpp), and every wait is bounded by kCircSpinTicks of the 100 MHz clock: a strip that gives up sets
// a sticky abort word that every other wait sees, the launch drains, and the ABI call returns an error.
#include <algorithm>
#include <cstring>
#include <vector>

#include "greb_step_order.h"
#include "greb_step_strip.h"

namespace greb {
namespace {
using namespace rows;

struct CircArgs {
  float* X[2];             // [n_members][2][ny][nx]  {Tair, q}: sub-step s reads X[s & 1], writes X[(s + 1) & 1]
  const float* W2;         // [2][ny][nx]             {wz_air, wz_vapor}
  const float* u;          // [ny][nx] winds of the model step, shared by every member and every sub-step
  const float* v;
  const RowTables* tabs;
  const CircTask* tasks;
  unsigned* flags;         // [n_tasks]
  unsigned* ctrl;
  unsigned epoch0;         // what every flag reads when the launch starts
  unsigned spin_ticks;
  int n_tasks;
  int ny, nsub, calm_odd;  // calm_odd: the vapour fields see zero wind (greb.original.model.f90:560-564)
  int chains_first;        // who issues first where a chain and a streaming strip share a SIMD
#ifdef GREB_TUNING
  int lose_task;           // this task leaves at once and never publishes: the test of the bounded waits (-1: none)
#endif
  int chain_head, chain_tail; // sweeps of a chain task's diffusion chain before it publishes the previous sub-step / after it polls
  unsigned long long* stamps;   // -DGREB_TUNING builds only: [s] = s_memrealtime at the start of sub-step s of task 0, [nsub] its end
  unsigned long long* timeline; // -DGREB_TUNING builds only: [task][start, end] + [2 n + task]: hw id
};

typedef __attribute__((address_space(4))) CircTask ctask;

// A workgroup is FOUR wavefronts = four tasks, one per SIMD of the compute unit it lands on (the wavefronts of a workgroup
// are dealt to the SIMDs in turn): which tasks share a SIMD -- i and i + n_simd, the rule circ_rows_tasks pairs by -- then
// follows from how workgroups are dealt to compute units and no longer from where single-wavefront workgroups happen
// to land (observed: of 998 such workgroups on 1 024 idle SIMDs, 54 SIMDs got two and 80 none).  The four share nothing:
// no barrier, each its own 19.5 KB of the workgroup's LDS.
constexpr int kTasksPerGroup = 4;
__device__ __forceinline__ int task_index() {
  return (int)blockIdx.x * kTasksPerGroup + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

__device__ __forceinline__ void flag_store(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// every vector-memory operation of this wavefront has completed; then its sub-step count becomes visible
__device__ __forceinline__ void publish(unsigned* flag, unsigned value, unsigned lane) {
  drain();
  if (lane == 0) flag_store(flag, value);
}

__device__ __forceinline__ void give_up(const CircArgs& a, int s, unsigned seen, unsigned want, unsigned lane) {
  if (lane == 0) {
    flag_store(a.ctrl + 1, (unsigned)task_index()); flag_store(a.ctrl + 2, (unsigned)s);
    flag_store(a.ctrl + 3, seen); flag_store(a.ctrl + 4, want);
    drain();
    flag_store(a.ctrl, 1u);
    drain();
  }
}

// Lanes 0-3 watch one dependency each (`mine`: its flag word, null where there is none), lane 4 the abort word.
// True when every dependency has completed `want` sub-steps (counted from the order's creation; wrap-safe);
// false when the launch has been given up -- by somebody else, or by this wavefront after spin_ticks.
__device__ __forceinline__ bool deps_ready(unsigned seen, unsigned want, unsigned lane, bool& aborted) {
  const bool ok = lane == 4 ? seen == 0u : (int)(seen - want) >= 0;
  const unsigned long long bad = __builtin_amdgcn_ballot_w64(!ok);
  aborted = (bad & 16ull) != 0;
  return bad == 0;
}
#ifdef GREB_TUNING
#define GREB_CIRC_WAITED(t0) if (a.timeline && lane == 0) a.timeline[3 * a.n_tasks + task_index()] += __builtin_amdgcn_s_memrealtime() - (t0)
#define GREB_CIRC_DRAINED(t0) if (a.timeline && lane == 0) a.timeline[4 * a.n_tasks + task_index()] += __builtin_amdgcn_s_memrealtime() - (t0)
#else
#define GREB_CIRC_WAITED(t0)
#define GREB_CIRC_DRAINED(t0)
#endif
__device__ bool wait_deps(const CircArgs& a, const unsigned* mine, unsigned want, unsigned lane, int s) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    unsigned seen = lane == 4 ? 0u : want;
    if (mine) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(mine) : "memory");
    bool aborted;
    if (deps_ready(seen, want, lane, aborted)) { GREB_CIRC_WAITED(t0); return true; }
    if (aborted) return false;
    if (__builtin_amdgcn_s_memrealtime() - t0 > a.spin_ticks) {
      // (what the first unsatisfied lane saw)
      const unsigned long long bad = __builtin_amdgcn_ballot_w64(lane < 4 && (int)(seen - want) < 0);
      const int l = bad ? __builtin_ctzll(bad) : 0;
      give_up(a, s, (unsigned)__builtin_amdgcn_readlane((int)seen, l), want, lane);
      return false;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

// ---------------------------------------------------------------------------------------------- the chain task
// LDS of a chain task (the same 19.5 KB as a strip's): [out row | halo buffer 0: pairs (r-2, r-1), (r+1, r+2) | halo buffer
// 1 | two staging slots]; the poll words (five of them) sit at the start of the first staging slot once the launch's
// operands have been read out of it.
constexpr unsigned kHaloBase = kRingBase, kStageBase = kWindBase, kPollBase = kWindBase;

template <bool STRICT, int NXR>
__device__ __forceinline__ void chain_task(lfloat* lds, const CircArgs& a, const crow_tables& tab, int fld, int r,
                                           const unsigned* mine, unsigned lane) {
  const int ny = a.ny, tracer = fld & 1;
  const size_t np = (size_t)NXR * ny;
  const float* wf = a.W2 + (size_t)tracer * np;
  const LaneAddr L = lane_addr(lane);
  const unsigned lb = (unsigned)(size_t)lds;
  const bool calm = a.calm_odd && tracer;
  unsigned* const my_flag = a.flags + task_index();
  // rows outside the grid: fetched from the nearest row inside (finite values) and given weight zero below
  const int rm2 = r >= 2 ? r - 2 : 0, rm1 = r >= 1 ? r - 1 : 0, rp1 = r + 1 < ny ? r + 1 : ny - 1, rp2 = r + 2 < ny ? r + 2 : ny - 1;
  auto fetch = [&](const float* A, const float* B, unsigned at, auto aux) { // rows A and B -> the 3 KB slot at byte `at`
    issue_pair_p<decltype(aux)::value>(pair_ptrs<NXR>(A, B, lane), 0, lds + at / 4);
  };
  using plain = std::integral_constant<int, kAuxPlain>;
  using sc1 = std::integral_constant<int, kAuxSc1>;
  auto fetch_halo = [&](const float* Xf, int buf) { // rows r-2, r-1 | r+1, r+2 of the field as it stands in Xf
    fetch(Xf + rm2 * NXR, Xf + rm1 * NXR, kHaloBase + (2 * buf) * kSlotB, sc1{});
    fetch(Xf + rp1 * NXR, Xf + rp2 * NXR, kHaloBase + (2 * buf + 1) * kSlotB, sc1{});
  };
  // ---- once per call: the row, its weights and winds, the neighbours' weights; all in flight together
  const float* X0 = a.X[0] + (size_t)fld * np;
  fetch(wf + rm2 * NXR, wf + rm1 * NXR, kStageBase, plain{});
  fetch(wf + rp1 * NXR, wf + rp2 * NXR, kStageBase + kSlotB, plain{});
  fetch(X0 + r * NXR, wf + r * NXR, kHaloBase + 2 * kSlotB, sc1{});
  fetch(a.u + r * NXR, a.v + r * NXR, kHaloBase + 3 * kSlotB, plain{});
  fetch_halo(X0, 0);
  drain();
  float Tw[5][6], ww[5][6], u[6], v[6];
  read_pair(L, lb + kStageBase, ww[0], ww[1]);
  read_pair(L, lb + kStageBase + kSlotB, ww[3], ww[4]);
  read_pair(L, lb + kHaloBase + 2 * kSlotB, Tw[2], ww[2]);
  read_pair(L, lb + kHaloBase + 3 * kSlotB, u, v);
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if (r < 2) ww[0][j] = 0.f;
    if (r < 1) ww[1][j] = 0.f;
    if (r + 1 >= ny) ww[3][j] = 0.f;
    if (r + 2 >= ny) ww[4][j] = 0.f;
    if (calm) { u[j] = 0.f; v[j] = 0.f; }
  }
  const float ccy_dif = tab.dif_ccy, ccy_adv = tab.adv_ccy;
  const int t2d = tab.dif_time2[r], t2a = tab.adv_time2[r];
  const float ccd = tab.dif_ccx2[r], cca = tab.adv_ccx2[r];
  const bool last_lane = bug_lane<NXR>(lane);
  float wc[12];
  chain_halo(ww[2], wc);
  const float u0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // FAST: the coefficient form of both chains, built once for the 24 sub-steps (a strip builds them every sub-step)
  ChainK Kd, Ka;
  bool convex = false;
  if (!STRICT) {
    float K[6][6];
    chain_coefficients<6>(wc, u0, ccd, false, last_lane, K);
    convex = chain_sweeps_plain_range(t2d) && chain_coefficients_convex(K);
    Kd = chain_pack(K);
    if (t2a > 1) {
      chain_coefficients<6>(wc, u, cca, true, last_lane, K);
      Ka = chain_pack(K);
    }
  }
  // the diffusion chain in three parts: [head] publish the previous sub-step [body] poll for the next [tail]
  int head = a.chain_head, tail = a.chain_tail;
  if (head + tail + 8 > t2d) { head = 0; tail = 0; }
  const bool prio = a.chains_first != 0; // the long chains issue ahead of whatever shares their SIMD -- unless SIMDs are shared by design
  int fetched = 0;                  // halo(0) .. halo(fetched) have been requested
  const int last_halo = a.nsub - 1; // ... of halo(0) .. halo(nsub - 1)
  for (int s = 0; s < a.nsub; ++s) {
    float* dst = a.X[(s + 1) & 1] + (size_t)fld * np;
#ifdef GREB_TUNING
    if (a.stamps && task_index() == 0 && lane == 0) a.stamps[s] = __builtin_amdgcn_s_memrealtime();
#endif
    const float (&T0)[6] = Tw[2];
    float Td[6], Ta[6];
    // ---- zonal part (:656-717, :842-909): from registers
#ifdef GREB_TUNING
    const unsigned long long tz = __builtin_amdgcn_s_memrealtime();
#endif
    if (prio) __builtin_amdgcn_s_setprio(3);
    if (STRICT) {
      float Tc[12];
      chain_halo(T0, Tc);
      chain_window<true, 6>(Tc, wc, u, cca, t2a, true, last_lane ? 63 : 0, false); // (63: the lane with the :881 index bug)
#pragma unroll
      for (int j = 0; j < 6; ++j) Ta[j] = Tc[3 + j];
      chain_halo(T0, Tc);
      int done = 0;
      auto part = [&](int n) {
        if (n <= 0) return;
        if (done > 0) { // the halo of the part's first sweep: chain_window refreshes it only between ITS sweeps
          float own[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) own[j] = Tc[3 + j];
          chain_halo(own, Tc);
        }
        chain_window<true, 6>(Tc, wc, u0, ccd, n, false, last_lane ? 63 : 0, false);
        done += n;
      };
      part(head);
      if (s > 0) publish(my_flag, a.epoch0 + (unsigned)s, lane); // sub-step s - 1 has left (its stores were issued `head` sweeps ago)
      part(t2d - head - tail);
      if (fetched < last_halo && mine) glds4(mine, lds + kPollBase / 4);
      part(tail);
#pragma unroll
      for (int j = 0; j < 6; ++j) Td[j] = Tc[3 + j];
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) { Td[j] = T0[j]; Ta[j] = T0[j]; }
      if (t2a > 1) chain_run6<false>(Ta, Ka, t2a);
      else { // the single sweep in edge-flux form, as a strip computes it
        RowFlux f;
        row_flux(T0, ww[2], f);
        adv_sweep_fast(T0, u, f, cca * 0.05f, last_lane, Ta);
      }
      const bool positive = convex && chain_range_positive(T0);
      if (head > 0) chain_run6<false>(Td, Kd, head, positive);
      if (s > 0) publish(my_flag, a.epoch0 + (unsigned)s, lane); // sub-step s - 1 has left (its stores were issued `head` sweeps ago)
      chain_run6<false>(Td, Kd, t2d - head - tail, positive);
      if (fetched < last_halo && mine) glds4(mine, lds + kPollBase / 4);
      if (tail > 0) chain_run6<false>(Td, Kd, tail, positive);
    }
    __builtin_amdgcn_s_setprio(0);
    GREB_CIRC_DRAINED(tz); // (a chain task's second tuning counter: the time in its chains, publish included)
    // ---- the neighbours' rows.  halo(j) = rows r-2 .. r+2 of X[j & 1] once their owners have completed sub-step j - 1;
    // it is requested as early as the owners allow -- normally one sub-step ahead, during the chains -- and NEEDED only
    // here, so a neighbour may lag this task by up to a sub-step before anybody waits (tasks that share a SIMD with
    // streaming strips are not in lock step with their neighbours).  This task's own count was published after the
    // `head` sweeps above: nothing below can make two neighbouring chain tasks wait for one another.
    drain(); // the poll words (and every halo requested so far)
    if (fetched < last_halo) {
      unsigned seen = lane == 4 ? 0u : a.epoch0 + (unsigned)a.nsub;
      if (mine) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(lb + kPollBase + 4 * lane) : "memory");
      const int before = fetched;
      for (;;) {
        const int next = fetched + 1;
        if (next > last_halo || next > s + 1) break;
        bool aborted;
        if (!deps_ready(seen, a.epoch0 + (unsigned)next, lane, aborted)) {
          if (aborted) return;
          if (next > s) break; // not needed yet: asked for again during the next sub-step's chains
          if (!wait_deps(a, mine, a.epoch0 + (unsigned)next, lane, s)) return; // needed now
        }
        fetch_halo(a.X[next & 1] + (size_t)fld * np, next & 1);
        fetched = next;
      }
      if (before < s) drain(); // this sub-step's own rows were requested only now
    }
    // ---- meridional part and the update
    const unsigned hb = lb + kHaloBase + 2 * (s & 1) * kSlotB;
    read_pair(L, hb, Tw[0], Tw[1]);
    read_pair(L, hb + kSlotB, Tw[3], Tw[4]);
    float o[6];
    meridional_update<STRICT>(Tw, ww, Td, Ta, v, ccy_dif, ccy_adv, r, ny, o);
    vfloat4 q0, q1;
    transpose_out(L, lb + kOutBase, o, q0, q1);
    (void)store_row_quads<NXR>(dst + r * NXR, lane, q0, q1, [](float* p_, vfloat4 q_) { store16<true>(p_, q_); });
    order_fence();
#pragma unroll
    for (int j = 0; j < 6; ++j) Tw[2][j] = o[j];
  }
  publish(my_flag, a.epoch0 + (unsigned)a.nsub, lane);
#ifdef GREB_TUNING
  if (a.stamps && task_index() == 0 && lane == 0) a.stamps[a.nsub] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <bool STRICT, int NXR>
__global__ __launch_bounds__(64 * kTasksPerGroup, 2) void circ_rows_kernel(const CircArgs a) {
  extern __shared__ __align__(16) float lds_raw[];
  const int task = task_index();
  if (task >= a.n_tasks) return; // (the last workgroup of a launch whose task count is not a multiple of four)
#ifdef GREB_TUNING
  if (task == a.lose_task) return; // (tests/test_gpu_tools.py: its neighbours must give up after spin_ticks, not hang)
#endif
  lfloat* lds = (lfloat*)lds_raw + (size_t)(task & (kTasksPerGroup - 1)) * (kStepLdsB / 4);
  const ctask& tk = *(const ctask*)(a.tasks + task); // eight dwords through the scalar cache
  int fld = tk.field;
  const int task_rows = tk.rows;
  const int d0 = tk.dep[0], d1 = tk.dep[1], d2 = tk.dep[2], d3 = tk.dep[3];
#ifdef GREB_TUNING
  if (a.timeline && (threadIdx.x & 63) == 0) {
    a.timeline[2 * task] = __builtin_amdgcn_s_memrealtime();
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); // wave, SIMD, CU, SE ids
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.timeline[2 * a.n_tasks + task] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  const int tab_idx = (int)((unsigned)fld >> kStepFieldBits);
  fld &= (1 << kStepFieldBits) - 1;
  const int k0 = task_rows & 0xff, k1 = (task_rows >> 8) & 0x1ff, ny = a.ny;
  const unsigned lane = threadIdx.x & 63;
  const int tracer = fld & 1;
  const crow_tables& tab = *(const crow_tables*)(a.tabs + tab_idx);
  // what this lane watches while the wavefront waits: lanes 0-3 a dependency each, lane 4 the abort word
  const int dep = lane == 0 ? d0 : (lane == 1 ? d1 : (lane == 2 ? d2 : (lane == 3 ? d3 : -1)));
  const unsigned* mine = lane == 4 ? a.ctrl : (dep >= 0 ? a.flags + dep : nullptr);
  if (task_rows & kCircChain) {
    chain_task<STRICT, NXR>(lds, a, tab, fld, k0, mine, lane);
  } else {
    const size_t np = (size_t)NXR * ny;
    const StripStamps st{nullptr, nullptr};
    for (int s = 0; s < a.nsub; ++s) {
      if (s > 0 && !wait_deps(a, mine, a.epoch0 + (unsigned)s, lane, s)) return;
      const StripIo io{a.X[s & 1] + (size_t)fld * np, a.W2 + (size_t)tracer * np, a.X[(s + 1) & 1] + (size_t)fld * np, a.u, a.v};
      stream_strip<STRICT, kAuxSc1, true, NXR>(lds, io, tab, k0, k1, ny, a.calm_odd && tracer, a.chains_first, lane, st);
#ifdef GREB_TUNING
      const unsigned long long td = __builtin_amdgcn_s_memrealtime();
#endif
      publish(a.flags + task, a.epoch0 + (unsigned)s + 1u, lane);
      GREB_CIRC_DRAINED(td);
    }
  }
#ifdef GREB_TUNING
  if (a.timeline && (threadIdx.x & 63) == 0) a.timeline[2 * task + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

} // namespace

// The tasks of one circulation call: per field the chain rows (one task each) and, between them, strips cut by the
// cost model of greb_step_order.h -- at most n_slots tasks in all, ONE round (see step_rows_tasks for the reasoning:
// the smallest S such that a SIMD's pair of tasks ends after S cycles), the tasks with the most issue alone on their
// SIMD, the others paired dearest with cheapest.  Then every task's dependencies: the owners of the two rows below and
// the two rows above its own.
void circ_rows_tasks(const RowTables* tabs, const int* tab_index, int n_members, int ny, int n_slots,
                     std::vector<CircTask>& tasks) {
  struct T { int field, k0, k1; long long issue, wall; bool chain; };
  static const int issue_pct = tuning_int("GREB_STEP_ISSUE_PCT", 54); // -DGREB_TUNING builds only
  static const int wall_pct = tuning_int("GREB_STEP_WALL_PCT", 70);
  static const int chain_min = tuning_int("GREB_CIRC_CHAIN_MIN", kChainTaskMinSweeps);
  static const int c_alone = tuning_int("GREB_CIRC_CALONE", 0); // experiment: the dearest chain tasks never share a SIMD
  const int n_simd = std::max(1, n_slots / 2);
  auto is_chain = [&](const RowTables& t, int k) { return t.dif_time2[k] >= chain_min; };
  // a chain task: its sweeps and set-up, the meridional part, no streaming
  static const int chain_weight = tuning_int("GREB_CIRC_CHAIN_WEIGHT", 120); // per cent of the modelled cost: a chain task on a shared SIMD is the one that never waits (tools/circ_timeline.py), so it is dealt the cheaper partner
  auto chain_cost = [&](const RowTables& t, int k) {
    const RowCost c = step_row_cost(t, k);
    return (long long)(c.issue - kRowIssue + 800) * chain_weight / 100;
  };
  long long total = 0, dearest = 0;
  for (int m = 0; m < n_members; ++m)
    for (int k = 0; k < ny; ++k) {
      const RowTables& t = tabs[tab_index[m]];
      const RowCost c = step_row_cost(t, k);
      total += 2 * (is_chain(t, k) ? chain_cost(t, k) : c.issue);
      dearest = std::max<long long>(dearest, is_chain(t, k) ? chain_cost(t, k) : c.wall + kFillWall);
    }
  long long S = std::max(total / n_simd, dearest);
  std::vector<T> all;
  tasks.clear();
  for (int pass = 0; pass < 200; ++pass) {
    const long long cap_issue = S * issue_pct / 100, cap_wall = S * wall_pct / 100;
    all.clear();
    for (int m = 0; m < n_members; ++m) {
      const RowTables& t = tabs[tab_index[m]];
      std::vector<T> mine;
      int a = 0;
      for (int k = 0; k <= ny; ++k) {
        if (k < ny && !is_chain(t, k)) continue;
        if (k > a) { // the rows between two chain rows
          std::vector<Strip> seg;
          cut_rows(t, a, k, cap_issue, cap_wall, seg);
          for (const Strip& x : seg) mine.push_back({0, x.k0, x.k1, x.issue, x.wall, false});
        }
        if (k < ny) mine.push_back({0, k, k + 1, chain_cost(t, k), chain_cost(t, k), true});
        a = k + 1;
      }
      for (int tr = 0; tr < 2; ++tr)
        for (T x : mine) { x.field = 2 * m + tr; all.push_back(x); }
    }
    int n_dear = 0;
    if (c_alone) for (const T& x : all) n_dear += x.chain && x.issue >= 25000;
    if ((int)all.size() <= n_slots - n_dear) break;
    S += S / 40;
    if (pass == 199) return; // (cannot happen for ny <= 192: one strip per segment is reached long before) -- no tasks: no launch
  }
  if (c_alone) for (T& x : all) if (x.chain && x.issue >= 25000) x.issue += 1000000; // (sorted first: alone)
  std::stable_sort(all.begin(), all.end(), [](const T& x, const T& y) { return x.issue > y.issue; });
  if (c_alone) for (T& x : all) if (x.issue >= 1000000) x.issue -= 1000000;
  const int n_all = (int)all.size();
  if (n_all > n_simd) {
    const int m = n_all - n_simd, alone = n_simd - m; // m SIMDs hold a pair
    std::vector<T> order((size_t)n_all);
    for (int j = 0; j < m; ++j) {
      order[(size_t)j] = all[(size_t)(alone + j)];                // the dearer of pair j ...
      order[(size_t)(n_simd + j)] = all[(size_t)(n_all - 1 - j)]; // ... and the cheapest left
    }
    for (int j = 0; j < alone; ++j) order[(size_t)(m + j)] = all[(size_t)j];
    // ... as the hardware deals them: the wavefronts of the SECOND workgroup on a compute unit start one SIMD further on
    // (observed, tools/circ_timeline.py: second-round wavefront w sits on SIMD (w + 1) mod 4, so task i shares its SIMD
    // with task i + 1 023 or i + 1 027, never i + 1 024): the partner meant for SIMD w goes to wavefront (w + 3) mod 4
    for (int c = 0; n_simd + 4 * c + 3 < n_all; ++c) {
      const size_t b = (size_t)n_simd + 4 * (size_t)c;
      const T t0 = order[b], t1 = order[b + 1], t2 = order[b + 2], t3 = order[b + 3];
      order[b + 3] = t0; order[b] = t1; order[b + 1] = t2; order[b + 2] = t3;
    }
    all.swap(order);
  }
  // who owns which row of which field
  std::vector<int> owner((size_t)2 * n_members * ny, -1);
  for (int i = 0; i < n_all; ++i)
    for (int k = all[(size_t)i].k0; k < all[(size_t)i].k1; ++k) owner[(size_t)all[(size_t)i].field * ny + k] = i;
  tasks.reserve((size_t)n_all);
  for (int i = 0; i < n_all; ++i) {
    const T& x = all[(size_t)i];
    CircTask c{x.field | (tab_index[x.field >> 1] << kStepFieldBits), x.k0 | (x.k1 << 8) | kRowsUp | (x.chain ? kCircChain : 0),
               {-1, -1, -1, -1}, (int)std::min<long long>(x.issue, 0x7fffffff), 0};
    int nd = 0;
    const int near[4] = {x.k0 - 2, x.k0 - 1, x.k1, x.k1 + 1};
    for (int j = 0; j < 4; ++j) {
      if (near[j] < 0 || near[j] >= ny) continue;
      const int o = owner[(size_t)x.field * ny + near[j]];
      if (o == i) continue;
      bool seen = false;
      for (int q = 0; q < nd; ++q) seen = seen || c.dep[q] == o;
      if (!seen) c.dep[nd++] = o;
    }
    tasks.push_back(c);
  }
}

hipError_t circ_rows_make_order(const RowTables* tabs_host, const int* tab_index_host, int n_members, int ny, int n_slots,
                                CircOrder* out) {
  std::vector<CircTask> host;
  circ_rows_tasks(tabs_host, tab_index_host, n_members, ny, n_slots, host);
  *out = CircOrder{};
  if (host.empty() || (int)host.size() > n_slots) return hipSuccess; // n == 0: the caller takes one launch per sub-step
  hipError_t e = hipMalloc(&out->tasks, host.size() * sizeof(CircTask));
  if (e == hipSuccess) e = hipMalloc(&out->flags, host.size() * sizeof(unsigned));
  if (e == hipSuccess) e = hipMalloc(&out->ctrl, 8 * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemcpy(out->tasks, host.data(), host.size() * sizeof(CircTask), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(out->flags, 0, host.size() * sizeof(unsigned));
  if (e == hipSuccess) e = hipMemset(out->ctrl, 0, 8 * sizeof(unsigned));
  if (e != hipSuccess) { circ_rows_free_order(out); return e; }
  out->n = (int)host.size();
  out->epoch = 0;
  return hipSuccess;
}

void circ_rows_free_order(CircOrder* o) {
  if (o->tasks) (void)hipFree(o->tasks);
  if (o->flags) (void)hipFree(o->flags);
  if (o->ctrl) (void)hipFree(o->ctrl);
  *o = CircOrder{};
}

int circ_rows_status(const CircOrder& o, unsigned* diag5) {
  unsigned h[8] = {0};
  if (!o.ctrl) return 0;
  if (hipMemcpy(h, o.ctrl, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -2;
  if (diag5) std::memcpy(diag5, h + 1, 5 * sizeof(unsigned));
  return h[0] ? -1 : 0;
}

#ifdef GREB_TUNING
// diagnostic builds only: s_memrealtime at the start of each sub-step of task 0 of the last launch, and its end
static unsigned long long* g_circ_stamps = nullptr;
static unsigned long long* g_circ_timeline = nullptr;
static int g_circ_timeline_cap = 0;
extern "C" int greb_tuning_circ_stamps(unsigned long long* out, int n) {
  if (!out) {
    if (!g_circ_stamps && hipMalloc(&g_circ_stamps, 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
    return hipMemset(g_circ_stamps, 0, 64 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
  }
  if (!g_circ_stamps || n > 64) return -1;
  return hipMemcpy(out, g_circ_stamps, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
// [task][start, end] of the LAST launch in 100 MHz ticks, one hardware id per task, then per task the ticks spent waiting
// for neighbours and draining stores, summed over the launches since the last read; out == null arms it for `capacity` tasks
extern "C" int greb_tuning_circ_timeline(unsigned long long* out, int capacity) {
  if (!out) {
    if (g_circ_timeline) (void)hipFree(g_circ_timeline);
    g_circ_timeline = nullptr; g_circ_timeline_cap = 0;
    if (capacity <= 0) return 0;
    if (hipMalloc(&g_circ_timeline, (size_t)capacity * 5 * sizeof(unsigned long long)) != hipSuccess) return -1;
    g_circ_timeline_cap = capacity;
    return hipMemset(g_circ_timeline, 0, (size_t)capacity * 5 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
  }
  if (!g_circ_timeline || capacity > g_circ_timeline_cap) return -1;
  const hipError_t e = hipMemcpy(out, g_circ_timeline, (size_t)capacity * 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  // (the waits are sums: start the next launch's from zero; capacity must equal the launch's task count, which the layout assumes)
  (void)hipMemset(g_circ_timeline + 3 * (size_t)capacity, 0, (size_t)capacity * 2 * sizeof(unsigned long long));
  return e == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_circulation_rows(float* X0, float* X1, const float* W2, const float* u, const float* v,
                                   const RowTables* tabs_dev, CircOrder& order, int n_simd, int nx, int ny, int nsub, bool strict,
                                   hipStream_t s, bool calm_vapor) {
  if (order.n <= 0 || order.n > 2 * n_simd || nsub < 1 || (nx != kNx && 2 * nx != kNx)) return hipErrorInvalidValue; // more tasks than wavefront slots: never launched
  static const int forced = tuning_int("GREB_STEP_CHAINS_FIRST", -1); // -DGREB_TUNING builds only (A/B)
  static const int head = tuning_int("GREB_CIRC_HEAD", 8), tail = tuning_int("GREB_CIRC_TAIL", 24);
  const bool chains_first = forced >= 0 ? forced != 0 : order.n <= n_simd;
  CircArgs a{};
  a.X[0] = X0; a.X[1] = X1; a.W2 = W2; a.u = u; a.v = v; a.tabs = tabs_dev; a.tasks = order.tasks;
  a.flags = order.flags; a.ctrl = order.ctrl; a.epoch0 = order.epoch; a.spin_ticks = kCircSpinTicks;
  a.n_tasks = order.n; a.ny = ny; a.nsub = nsub; a.calm_odd = calm_vapor ? 1 : 0; a.chains_first = chains_first ? 1 : 0;
  a.chain_head = head; a.chain_tail = tail;
#ifdef GREB_TUNING
  a.lose_task = tuning_int("GREB_CIRC_LOSE_TASK", -1);
  if (const int spin_ms = tuning_int("GREB_CIRC_SPIN_MS", 0)) a.spin_ticks = (unsigned)spin_ms * 100000u; // a shorter bound for that test
#endif
#ifdef GREB_TUNING
  a.stamps = g_circ_stamps;
  a.timeline = order.n == g_circ_timeline_cap ? g_circ_timeline : nullptr;
#endif
  auto kern = nx == kNx ? (strict ? circ_rows_kernel<true, kNx> : circ_rows_kernel<false, kNx>)
                        : (strict ? circ_rows_kernel<true, kNx / 2> : circ_rows_kernel<false, kNx / 2>);
  // (four wavefronts' LDS is more than the 64 KB a kernel may ask for by default; the same ceiling as every other kernel
  // of the library, greb_kernels.h: kMaxDynamicLds -- set on every launch, it is idempotent and two host threads may race)
  hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
  if (ea != hipSuccess) return ea;
  hipLaunchKernelGGL(kern, dim3((unsigned)((order.n + kTasksPerGroup - 1) / kTasksPerGroup)), dim3(64 * kTasksPerGroup),
                     kTasksPerGroup * kStepLdsB, s, a);
  order.epoch += (unsigned)nsub;
  return hipGetLastError();
}

} // namespace greb
