// greb_kernels.h -- launch interface between the C-ABI host code (greb_engine.cpp) and the HIP
// kernels (greb_kernels.hip).  Internal; the public surface is include/greb_engine.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

#include "greb_device.h"

namespace greb {

constexpr int kNT = 730;

// Timing-experiment knobs (skip a family of tasks, shorten the sub-step loop, tile sizes) exist only in a
// -DGREB_TUNING build (greb_climate_model_amd/build.py: build_lib(tuning=True) -> libgreb_hip_tuning.so, used by
// tools/).  The release library never reads the environment: a stray variable cannot change the physics.
#ifdef GREB_TUNING
inline int tuning_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#else
constexpr int tuning_int(const char*, int dflt) { return dflt; }
#endif

// Kernels whose dynamic LDS size depends on the call (grid, band size, member count) are all allowed the SAME maximum:
// the attribute is per kernel, not per launch, and two host threads driving two engines side by side
// (ensemble.run_beside) must not lower it between each other's set and launch.  What a launch occupies is the size it
// passes, not this ceiling.
constexpr int kMaxDynamicLds = 160 * 1024;

// Everything the fused member kernel needs (passed by value as a kernel argument).
struct MemberArgs {
  int nx, ny, np;
  // shared static fields (HBM, read-only)
  const float *z_topo, *glacier, *sw_solar;
  const float *tclim, *qclim, *uclim, *vclim, *mldclim, *cldclim, *swetclim;
  const float *toclim, *z_ocean, *wz_air, *wz_vapor;
  // per member
  float* state;          // [nm][5][np]   Ts, Ta, To, q, cap_surf
  float* acc;            // [nm][6][np]   Tmm, Tamm, Tomm, qmm, apmm (src/greb.f90:149), tsmn (:145)
  float* corr;           // [ncorr][3][730][np]  TF_correct, qF_correct, ToF_correct (:110)
  const int* corr_index; // [nm] -> which correction set a member reads/writes
  const RowTables* tabs; // [ntab]
  const int* tab_index;  // [nm]
  const Phys* phys;      // [nm]
  // run control
  int flux_phase;        // 1: qflux_correction (:311-364), 0: scenario (:228-234 + time_loop)
  long long it0;         // `it` of the first step of this launch (1-based)
  int nsteps;
  int nsub;              // max(1,nint(dt/dt_crcl)) = 24 (:543)
  const float* co2;      // scenario: [nm][co2_stride] annual series; flux: unused
  int co2_stride;
  int co2_year0;         // index into the series of the year containing it0
  float co2_flux;
  float* monthly;        // [nm][rec_stride_years][12][5][np]; record for (year_out, month)
  int monthly_years;     // years per member in `monthly`
  int year_out0;         // output-year index of the year containing it0
  float* yearly;         // [nm][yearly_years][2] or nullptr
  int yearly_years, yearly_year0;
  int ipx, ipy;          // 1-based
  unsigned xsw;          // experiment switches (kX*, greb_device.h); 0 = complete model
  // -DGREB_TUNING builds only (null otherwise; the release kernels contain no stamp code): per member and wave, 8
  // cycle totals of one launch -- see tools/stamp_member.py
  unsigned long long* stamps;
  int dbg;               // -DGREB_TUNING builds only: timing experiments (results wrong): bit 0 point physics without the
                         // accumulators, bit 1 without the state write-back, bit 2 a single pass instead of three
};

// fused engine (greb_member.hip): 96x48 with the default sub-cycling layout -- rows 0-9 and 38-47
// sub-cycled, only the two polar rows iterating; whole member resident in one CU
bool member_layout_supported(const RowTables& tab, int nx, int ny);
hipError_t launch_member_kernel(const MemberArgs& a, int n_members, bool strict, hipStream_t s);
hipError_t launch_circulation_g96(const float* X, const float* wz, const float* u, const float* v, float* dX,
                                  const RowTables* tab_dev, int batch, int nsub, bool strict, hipStream_t s);

// batched single-routine kernels, any grid with nx % 4 == 0, ny <= kMaxNy
// tab_host: the same table on the host (the 384-wide row-strip kernel takes its row constants by value)
hipError_t launch_diffusion(const float* T1, const float* wz, float* dX, const RowTables* tab_dev,
                            const RowTables& tab_host, int nx, int ny, int batch, bool strict, hipStream_t s);
// greb_rows.hip: the diffusion sweep of a 384-wide grid as wavefront-sized row strips (FAST and STRICT)
bool rows_supported(const RowTables& t, int nx, int ny);
struct RowsTask { int field, rows; }; // rows = k0 | k1 << 8 | kRowsUp: the task updates rows [k0, k1); field < 0: no task
constexpr int kRowsUp = 1 << 20;       // the strip walks south to north (else north to south)
struct RowsTuning {
  int chain_target;    // instructions per chain strip
  int chain_span;      // per cent of the launch over which the chain strips are spread
  int parts[4];        // strips the streaming region of a field is cut into, level 0 (most fields) .. 3 (the last fields)
  int level_tasks[4];  // about how many tasks levels 1..3 hold ([0] unused)
};
// measured on MI355X, batch 1 024 (tools/rows_ab.sh; settled clocks): streaming cut 4 / 5 / 6 / 8 / 12 strips per field
// 0.1738 / 0.1707 / 0.1697 / 0.1685 / 0.1678 ms per launch; chain span 50 / 60 / 70 / 80 / 100 %: 0.1724 / 0.1717 / 0.1691 /
// 0.181 / 0.209; without the fine levels 0.1724 against 0.1691
inline RowsTuning rows_default_tuning() { return RowsTuning{3000, 70, {8, 8, 16, 39}, {0, 0, 2048, 1024}}; }
void rows_tasks(const RowTables& t, int ny, int batch, const RowsTuning& tu, std::vector<RowsTask>& tasks);
void rows_release_cache();
hipError_t launch_diffusion_rows(const float* T1, const float* wz, float* dX, const RowTables& t, int ny, int batch,
                                 bool strict, hipStream_t s);
// greb_step_rows.hip: the engine's circulation sub-step on a 384- or 192-wide grid as wavefront-sized row strips, one launch each
bool step_rows_supported(const RowTables* tabs, int n_tabs, int nx, int ny);
constexpr int kStepRowsSlotsPerCu = 8; // wavefronts of the row-strip circulation kernels a CU holds (177-201 VGPRs: two per SIMD; 19.5 KB of LDS each)
void step_rows_tasks(const RowTables* tabs, const int* tab_index, int n_members, int ny, int n_slots,
                     std::vector<RowsTask>& tasks);
constexpr int kStepFieldBits = 16;  // a sub-step task's field word: field | row-table index << 16 (at most 32 767 members)
constexpr int kStepHeadTasks = 16; // the first tasks of the launch order travel in the kernel arguments (see StepArgs)
hipError_t step_rows_make_tasks(const RowTables* tabs_host, const int* tab_index_host, int n_members, int ny,
                                int n_slots, RowsTask** dev, int* n, RowsTask* head /* [kStepHeadTasks] */);
hipError_t launch_substep_rows(const float* X, const float* W2, const float* u, const float* v, float* Xnew,
                               const RowTables* tabs_dev, const int* tab_index_dev, const RowsTask* tasks, const RowsTask* head_host,
                               int n_tasks, int n_simd, int nx, int ny, bool strict, hipStream_t s, bool calm_vapor);
// greb_circ_rows.hip: the circulation CALL (all its sub-steps, src/greb.f90:546-550) on a 384-wide grid in ONE launch.
// The strips of a field hand their rows to each other through memory flags, so every task of the launch must be
// resident at once: the caller passes the wavefront slots it may use and the builder never makes more tasks than that.
struct CircTask {
  int field;   // (2 * member + tracer) | the member's row-table index << kStepFieldBits
  int rows;    // k0 | k1 << 8 | kCircChain: rows [k0, k1)
  int dep[4];  // the tasks that own rows k0-2, k0-1, k1, k1+1 of the same field (each listed once; -1: none)
  int issue;   // the builder's cost estimate in cycles (diagnostic)
  int pad;
};
constexpr int kCircChain = 1 << 21;        // the task is ONE row whose zonal chains live in registers for the whole call
constexpr int kChainTaskMinSweeps = 64;    // rows with at least this many dependent diffusion sweeps become chain tasks
constexpr unsigned kCircSpinTicks = 100u * 1000u * 1000u; // a strip waits at most this long for a neighbour: 1 s of the 100 MHz clock
void circ_rows_tasks(const RowTables* tabs, const int* tab_index, int n_members, int ny, int n_slots,
                     std::vector<CircTask>& tasks);
struct CircOrder {
  CircTask* tasks = nullptr;  // device
  unsigned* flags = nullptr;  // device, [n]: sub-steps completed by each task since the order was made
  unsigned* ctrl = nullptr;   // device, [8]: [0] abort (sticky), [1] task, [2] sub-step, [3] flag seen, [4] flag wanted
  int n = 0;
  unsigned epoch = 0;         // what every flag reads between two launches
};
hipError_t circ_rows_make_order(const RowTables* tabs_host, const int* tab_index_host, int n_members, int ny, int n_slots,
                                CircOrder* out);
void circ_rows_free_order(CircOrder* o);
// X[0] / X[1]: sub-step s reads X[s & 1] and writes X[(s + 1) & 1]; the result is in X[nsub & 1]
hipError_t launch_circulation_rows(float* X0, float* X1, const float* W2, const float* u, const float* v,
                                   const RowTables* tabs_dev, CircOrder& order, int n_simd, int nx, int ny, int nsub, bool strict,
                                   hipStream_t s, bool calm_vapor);
// after the stream has been synchronised: 0, or -1 with the strip that gave up waiting in msg[0..4] (ctrl[1..5])
int circ_rows_status(const CircOrder& o, unsigned* diag5);
hipError_t launch_advection(const float* T1, const float* wz, const float* u, const float* v, float* dX,
                            const RowTables* tab_dev, int nx, int ny, int batch, bool strict, hipStream_t s);
// 24 sub-steps; 96x48 uses the fused LDS loop of the engine, other grids launch per sub-step
hipError_t launch_circulation(const float* X, const float* wz, const float* u, const float* v, float* dX,
                              float* scratch /* 3*batch*np */, const RowTables* tab_dev, const RowTables& tab_host,
                              int nx, int ny, int batch, int nsub, bool strict, hipStream_t s);

// any-grid multi-launch engine (greb_kernels.hip)
hipError_t launch_substep_fused(const float* X, const float* W2, const float* u, const float* v, float* Xnew,
                                const RowTables* tabs, const int* tab_index, int nx, int ny, int n_members,
                                bool strict, hipStream_t s, bool calm_vapor = false);
hipError_t launch_physics_step(const MemberArgs& a, const float* X, float* Xout, float* red, int n_members,
                               bool strict, hipStream_t s);
hipError_t launch_yearly(const float* red, float* yearly, int np, int nx, int ipx, int ipy, int yearly_years,
                         int year_index, int n_members, bool strict, hipStream_t s);
hipError_t launch_pack_tracers(const float* state, float* X, int np, int n_members, hipStream_t s);

struct PointArgs {
  int nx, ny, np, ityr;
  float co2;
  const float *z_topo, *glacier, *sw_solar, *tclim, *uclim, *vclim, *mldclim, *cldclim, *swetclim;
  const float *z_ocean, *wz_air;
  Phys phys;
  unsigned xsw;       // experiment switches
  const float* qclim; // for kXLwLinear
  const float* in5; // Ts, Ta, To, q, cap_surf
  float* out15;
};
hipError_t launch_point_physics(const PointArgs& a, hipStream_t s);

} // namespace greb
