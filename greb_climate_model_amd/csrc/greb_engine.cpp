// greb_engine.cpp -- the C ABI of include/greb_engine.h over the HIP kernels (greb_kernels.hip).
//
// Host-side responsibilities (everything the reference does once per run outside the time loops):
//   * per-row grid tables, src/greb.f90:578-582, 652-654, 749-753, 838-840 (fp32, same expression
//     order, glibc cosf like the flang-built reference)
//   * derived fields of greb_model's preamble, src/greb.f90:176-216, and Toclim, :1088-1094
//   * device residency: inputs are copied to HBM once in create; state, corrections and
//     accumulators live in HBM between launches and in LDS/registers inside a launch
//   * one launch of the fused member kernel per model year (730 steps) per phase
// There is NO CPU fallback: without a HIP device create() fails with GREB_E_NOGPU.
#include "../../include/greb_engine.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "greb_kernels.h"

using namespace greb;

namespace {

thread_local std::string g_last_error; // for failures before an engine exists

int fail(greb_engine* e, int code, const std::string& msg);

#define HIP_TRY(e, expr)                                                                      \
  do {                                                                                        \
    hipError_t _err = (expr);                                                                 \
    if (_err != hipSuccess)                                                                   \
      return fail((e), (int)_err, std::string(#expr) + ": " + hipGetErrorString(_err));       \
  } while (0)

int nint_f(float x) { return (int)lroundf(x); }

// src/greb.f90:578-582, 652-654 (diffusion), 749-753, 838-840 (advection)
void compute_row_tables(const greb_params& p, float kappa, int nx, int ny, RowTables& g) {
  std::memset(&g, 0, sizeof(g));
  const float dlon = 360.f / (float)nx, dlat = 180.f / (float)ny; // :43-44
  const float deg = 2.f * p.pi * 6.371e6f / 360.f;                // :578
  const float dx = dlon, dy = dlat, dyy = dy * deg;               // :579
  const float dtc = (float)p.dt_crcl;
  g.dif_ccy = kappa * dtc / (dyy * dyy); // :581
  g.adv_ccy = dtc / dyy / 2.f;           // :752
  for (int k = 0; k < ny; ++k) {
    const float lat = dlat * (float)(k + 1) - dlat / 2.f - 90.f;    // :580
    const float dxlat = dx * deg * cosf(2.f * p.pi / 360.f * lat);  // :580
    g.dif_ccx[k] = kappa * dtc / (dxlat * dxlat);                   // :582
    g.adv_ccx[k] = dtc / dxlat / 2.f;                               // :753
    g.subcycled[k] = !(dxlat > 2.5e5f);                             // :592, :799
    {
      float dd = (float)nint_f(dtc / (1.f * (dxlat * dxlat) / kappa)); // :652
      if (dd < 1.f) dd = 1.f;
      const int dtdff2 = (int)(dtc / dd);
      // dtdff2 == 0 (384x192 polar rows) is undefined behaviour in the reference (NINT(Inf));
      // the flang x86-64 build yields time2 = 1, ccx2 = 0 -- defined so here.
      int t2 = dtdff2 == 0 ? 1 : nint_f(dtc / (float)dtdff2); // :653
      g.dif_time2[k] = t2 < 1 ? 1 : t2;
      g.dif_ccx2[k] = kappa * (float)dtdff2 / (dxlat * dxlat); // :654
    }
    {
      float dd = (float)nint_f(dtc / (dxlat / 10.0f / 1.f)); // :838
      if (dd < 1.f) dd = 1.f;
      const int dtdff2 = (int)(dtc / dd);
      int t2 = dtdff2 == 0 ? 1 : nint_f(dtc / (float)dtdff2); // :839
      g.adv_time2[k] = t2 < 1 ? 1 : t2;
      g.adv_ccx2[k] = (float)dtdff2 / dxlat / 2.f; // :840
    }
  }
}

Phys make_phys(const greb_params& p, const greb_member_overrides* o) {
  Phys P;
  auto pick = [](float base, float ov) { return std::isnan(ov) ? base : ov; };
  P.sig = p.sig; P.ct_sens = p.ct_sens;
  P.da_ice = o ? pick(p.da_ice, o->da_ice) : p.da_ice;
  P.a_no_ice = o ? pick(p.a_no_ice, o->a_no_ice) : p.a_no_ice;
  P.a_cloud = o ? pick(p.a_cloud, o->a_cloud) : p.a_cloud;
  P.Tl_ice1 = p.Tl_ice1; P.Tl_ice2 = p.Tl_ice2; P.To_ice1 = p.To_ice1; P.To_ice2 = p.To_ice2;
  P.co_turb = p.co_turb; P.ce = p.ce; P.cq_latent = p.cq_latent; P.cq_rain = p.cq_rain;
  P.z_air = p.z_air; P.r_qviwv = p.r_qviwv; P.rho_air = p.rho_air;
  for (int i = 0; i < 10; ++i) P.p_emi[i] = p.p_emi[i];
  P.cap_ocean = p.cp_ocean * p.rho_ocean;          // :186
  P.cap_land = p.cp_land * p.rho_land * p.d_land;  // :187
  P.cap_air = p.cp_air * p.rho_air * p.d_air;      // :188
  P.dt = (float)p.dt;
  return P;
}

template <typename T>
hipError_t dev_alloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)); }

} // namespace

struct greb_engine {
  greb_params p{};
  int nx = 0, ny = 0, np = 0, nm = 0, device = 0;
  bool strict = false;
  unsigned xsw = 0; // sensitivity-experiment switches (GREB_X_*)
  bool shared_corr = true; // all members share physics -> one flux-correction set
  bool fused = true;       // every member has the 96x48 default sub-cycling layout -> fused member kernel
  float *Xa = nullptr, *Xb = nullptr, *red = nullptr, *W2 = nullptr; // any-grid (multi-launch) engine work arrays
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;               // device -> host delivery of the monthly means
  hipEvent_t ev_done[2] = {nullptr, nullptr};      // year written into staging slot i
  hipEvent_t ev_free[2] = {nullptr, nullptr};      // staging slot i copied out
  // device
  float *z_topo = nullptr, *glacier = nullptr, *sw_solar = nullptr;
  float *tclim = nullptr, *qclim = nullptr, *uclim = nullptr, *vclim = nullptr, *mldclim = nullptr,
        *cldclim = nullptr, *swetclim = nullptr;
  float *toclim = nullptr, *z_ocean = nullptr, *wz_air = nullptr, *wz_vapor = nullptr;
  float *state = nullptr, *acc = nullptr, *corr = nullptr;
  int *corr_index = nullptr, *tab_index = nullptr;
  RowTables* tabs = nullptr;
  Phys* phys = nullptr;
  float* co2_dev = nullptr; size_t co2_cap = 0;
  float* monthly_dev = nullptr; size_t monthly_cap = 0;
  float* yearly_dev = nullptr; size_t yearly_cap = 0;
  // host copies needed later
  std::vector<RowTables> h_tabs;
  std::vector<int> h_tab_index;
  bool step_rows = false;                                   // 384-wide grid: the row-strip sub-step (greb_step_rows.hip)
  bool step_rows_always = false;                            // GREB_F_ROW_STRIPS
  struct StepOrder { RowsTask* dev; int n; RowsTask head[kStepHeadTasks]; };
  std::map<int, StepOrder> step_tasks;                      // its launch order, per number of members run
  int cus = 0;                                              // compute units of the device (4 SIMDs each)
  bool persistent = false;                                  // ... its circulation call in ONE launch (greb_circ_rows.hip)
  std::map<int, CircOrder> circ_orders;                     // the tasks, flags and abort word of that launch, per members run
  int slots_granted = -1;                                   // wavefront slots of the device this engine may fill (-1: not asked yet)
  bool persistent_always = false;                           // GREB_F_PERSISTENT: no trial, the one-launch form wherever it can run
  struct FormTrial { int form = 0; float ms_substep = 0.f, ms_call = 0.f; }; // form: 0 undecided, 1 per sub-step, 2 per call
  std::map<int, FormTrial> circ_form;                       // which of the two forms runs, per members run
  hipEvent_t ev_trial[3] = {nullptr, nullptr, nullptr};
  std::vector<Phys> h_phys;
  // model clock
  long long it_flux = 0; // steps done in the flux phase
  long long it_scnr = 0; // steps done in the scenario
  unsigned long long* stamps = nullptr; // -DGREB_TUNING builds only: device buffer for the member kernel's stamps
  std::string last_error;
};

namespace {
int fail(greb_engine* e, int code, const std::string& msg) {
  if (e) e->last_error = msg;
  g_last_error = msg;
  return code;
}

int nsub_of(const greb_params& p) {
  int t = nint_f((float)p.dt / (float)p.dt_crcl); // :543
  return t < 1 ? 1 : t;
}

MemberArgs base_args(greb_engine* e) {
  MemberArgs a{};
  a.nx = e->nx; a.ny = e->ny; a.np = e->np;
  a.z_topo = e->z_topo; a.glacier = e->glacier; a.sw_solar = e->sw_solar;
  a.tclim = e->tclim; a.qclim = e->qclim; a.uclim = e->uclim; a.vclim = e->vclim;
  a.mldclim = e->mldclim; a.cldclim = e->cldclim; a.swetclim = e->swetclim;
  a.toclim = e->toclim; a.z_ocean = e->z_ocean; a.wz_air = e->wz_air; a.wz_vapor = e->wz_vapor;
  a.state = e->state; a.acc = e->acc; a.corr = e->corr; a.corr_index = e->corr_index;
  a.tabs = e->tabs; a.tab_index = e->tab_index; a.phys = e->phys;
  a.nsub = nsub_of(e->p);
  a.nsub = tuning_int("GREB_DEBUG_NSUB", a.nsub); // -DGREB_TUNING builds only
  a.co2_flux = e->p.co2_flux;
  a.ipx = e->p.ipx; a.ipy = e->p.ipy;
  a.xsw = e->xsw;
  a.stamps = e->stamps;
  a.dbg = tuning_int("GREB_DEBUG_PHYS", 0);
  if (e->xsw & GREB_X_NO_CIRCULATION) a.nsub = 0; // no transport at all: the tracers come back unchanged
  return a;
}

int ensure(greb_engine* e, float** buf, size_t* cap, size_t n) {
  if (*cap >= n) return 0;
  if (*buf) HIP_TRY(e, hipFree(*buf));
  *buf = nullptr; *cap = 0;
  HIP_TRY(e, dev_alloc(buf, n));
  *cap = n;
  return 0;
}

// The wavefront slots of a device that one-launch circulation calls may fill, across the engines of this process.
// Such a launch waits inside the kernel for its own tasks, so all of them must be resident at once; two engines driven
// side by side (ensemble.run_beside: config 5's 62 + 2 members) share the device, and the sum of what they launch must fit.
// Every 384-wide engine registers with its member count when it is created; an engine's grant is fixed the first time it
// needs one: its share of the slots by members among the engines registered then, and never more than what the grants
// already made leave.  An engine that gets too little for its tasks takes one launch per sub-step, which waits for
// nothing.  (Another PROCESS on the same device is outside this ledger: there the bounded waits turn a launch that is
// not co-resident into an error, never a hang.)
struct SlotLedger {
  std::mutex mu;
  struct Entry { greb_engine* e; int device, members, granted; };
  std::vector<Entry> entries;
} g_slots;

void ledger_register(greb_engine* e) {
  std::lock_guard<std::mutex> lock(g_slots.mu);
  g_slots.entries.push_back({e, e->device, e->nm, 0});
}
void ledger_release(greb_engine* e) {
  std::lock_guard<std::mutex> lock(g_slots.mu);
  for (size_t i = 0; i < g_slots.entries.size(); ++i)
    if (g_slots.entries[i].e == e) { g_slots.entries.erase(g_slots.entries.begin() + (long)i); break; }
}
int ledger_grant(greb_engine* e, int device_slots) {
  std::lock_guard<std::mutex> lock(g_slots.mu);
  long long members = 0, taken = 0;
  SlotLedger::Entry* mine = nullptr;
  for (auto& x : g_slots.entries) {
    if (x.device != e->device) continue;
    members += x.members;
    if (x.e == e) mine = &x; else taken += x.granted;
  }
  if (!mine) return 0;
  const long long share = (long long)device_slots * mine->members / std::max<long long>(1, members);
  mine->granted = (int)std::max<long long>(0, std::min<long long>(share, device_slots - taken));
  return mine->granted;
}

// after the stream has been synchronised: did a one-launch circulation call give up waiting?
int check_circulation(greb_engine* e) {
  for (auto& kv : e->circ_orders) {
    unsigned d[5] = {0, 0, 0, 0, 0};
    const int rc = circ_rows_status(kv.second, d);
    if (rc == -1) {
      char buf[384];
      std::snprintf(buf, sizeof(buf),
                    "circulation launch given up: task %u waited more than %.1f s in sub-step %u for a neighbouring strip "
                    "(its count read %u, wanted %u) -- the launch was not resident as a whole (another process on the "
                    "device?); results are invalid, recreate the engine with GREB_F_NO_PERSISTENT",
                    d[0], kCircSpinTicks / 1e8, d[1], d[2], d[3]);
      return fail(e, GREB_E_STATE, buf);
    }
    if (rc) return fail(e, GREB_E_STATE, "circulation launch: status word unreadable");
  }
  return 0;
}

// One model year (730 steps) for the first `nrun` members, `a` describing that year.
//   fused layout : one launch of the member kernel
//   other grids  : 24 fused band sub-steps + 1 point-physics launch per model step
int run_year(greb_engine* e, const MemberArgs& a, int nrun) {
  if (e->fused) {
    HIP_TRY(e, launch_member_kernel(a, nrun, e->strict, e->stream));
    return 0;
  }
  const size_t np = (size_t)e->np;
  // 384-wide grids, FAST: the row-strip sub-step (greb_step_rows.hip) at every member count -- us per launch against
  // the band kernels it replaces (the scalar sweep_kernel<fused> below 28 members, a (Tair,q)-pair band kernel above, round
  // 2): 1 member 23.2 / 25.1, 8: 27.8 / 28.8, 24: 30.0 / 30.5, 40: 35.6 / 35.2, 48: 39.6 / 40.5, 62: 40.8 / 48.0.
  // STRICT keeps the band kernel (1 member 64.7 against 74.6 us: its two chains per row run one after the other in
  // one wave here); GREB_F_ROW_STRIPS takes the strips there too.
  const bool rows = e->step_rows && (e->step_rows_always || !e->strict);
  // the whole circulation call (its nsub sub-steps) in ONE launch where every task of it can be resident at once ...
  CircOrder* circ = nullptr;
  if (rows && e->persistent && a.nsub > 0) {
    if (e->cus <= 0) HIP_TRY(e, hipDeviceGetAttribute(&e->cus, hipDeviceAttributeMultiprocessorCount, e->device));
    static const int slots_per_cu = tuning_int("GREB_CIRC_SLOTS_PER_CU", kStepRowsSlotsPerCu); // -DGREB_TUNING builds only (occupancy experiments: <= 8)
    if (e->slots_granted < 0) e->slots_granted = ledger_grant(e, e->cus * std::min(slots_per_cu, kStepRowsSlotsPerCu));
    auto it = e->circ_orders.find(nrun);
    if (it == e->circ_orders.end()) {
      CircOrder o;
      HIP_TRY(e, circ_rows_make_order(e->h_tabs.data(), e->h_tab_index.data(), nrun, e->ny, e->slots_granted, &o));
      it = e->circ_orders.emplace(nrun, o).first;
    }
    if (it->second.n > 0) circ = &it->second; // (0: the grant is too small for this many fields)
  }
  // Which of the two wins depends on how many fields there are and what shares a SIMD with what (one member: 15.1 against
  // 18.7 us per sub-step; 40 members: 29.6 against 25.8; 62: 36.0 against 38.7), and the two are bit-identical row by
  // row (tests/test_gpu_parity.py), so the engine MEASURES: the first eight model steps of the first year run two warm-up
  // steps, three timed steps of one form and three of the other between events on its own stream, and the faster form
  // runs from then on.  The results do not depend on the choice.
  static const int steps = tuning_int("GREB_DEBUG_NSTEPS", kNT); // -DGREB_TUNING builds only: a short stretch for counter passes
  greb_engine::FormTrial* trial = nullptr;
  int form = circ ? 2 : 1;
  if (circ && !e->persistent_always) {
    greb_engine::FormTrial& ft = e->circ_form[nrun];
    if (ft.form == 0 && steps >= 16) {
      trial = &ft;
      for (hipEvent_t& ev : e->ev_trial) if (!ev) HIP_TRY(e, hipEventCreate(&ev));
    } else if (ft.form) form = ft.form;
  }
  // ... else one launch per sub-step
  const RowsTask *step_tasks = nullptr, *step_head = nullptr;
  int n_step_tasks = 0;
  if (rows && (form == 1 || trial)) {
    auto it = e->step_tasks.find(nrun);
    if (it == e->step_tasks.end()) {
      RowsTask* dev = nullptr; int n = 0;
      if (e->cus <= 0) HIP_TRY(e, hipDeviceGetAttribute(&e->cus, hipDeviceAttributeMultiprocessorCount, e->device));
      greb_engine::StepOrder so{};
      static const int step_slots = tuning_int("GREB_STEP_SLOTS_PER_CU", kStepRowsSlotsPerCu); // -DGREB_TUNING builds only (occupancy experiments)
      HIP_TRY(e, step_rows_make_tasks(e->h_tabs.data(), e->h_tab_index.data(), nrun, e->ny, e->cus * step_slots, &dev, &n, so.head));
      so.dev = dev; so.n = n;
      it = e->step_tasks.emplace(nrun, so).first;
    }
    step_tasks = it->second.dev; n_step_tasks = it->second.n; step_head = it->second.head;
  }
  HIP_TRY(e, launch_pack_tracers(e->state, e->Xa, e->np, nrun, e->stream));
  for (int s = 0; s < steps; ++s) {
    const long long it = a.it0 + s;
    const int ityr = (int)((it - 1) % kNT) + 1;
    const size_t off = (size_t)(ityr - 1) * np;
    float *cur = e->Xa, *nxt = e->Xb;
    if (trial) { // steps 0-1 warm up, 2-4 one launch per sub-step, 5-7 one launch per call
      if (s == 2 || s == 5 || s == 8) HIP_TRY(e, hipEventRecord(e->ev_trial[s == 2 ? 0 : (s == 5 ? 1 : 2)], e->stream));
      if (s == 8) {
        HIP_TRY(e, hipEventSynchronize(e->ev_trial[2]));
        HIP_TRY(e, hipEventElapsedTime(&trial->ms_substep, e->ev_trial[0], e->ev_trial[1]));
        HIP_TRY(e, hipEventElapsedTime(&trial->ms_call, e->ev_trial[1], e->ev_trial[2]));
        form = trial->form = trial->ms_call <= trial->ms_substep ? 2 : 1;
        trial = nullptr;
      } else form = (s >= 2 && s < 5) ? 1 : 2;
    }
    if (form == 2) {
      HIP_TRY(e, launch_circulation_rows(e->Xa, e->Xb, e->W2, e->uclim + off, e->vclim + off, e->tabs, *circ, e->cus * 4, e->nx, e->ny,
                                         a.nsub, e->strict, e->stream, (e->xsw & GREB_X_VAPOR_DIFFUSION_ONLY) != 0));
      if (a.nsub & 1) cur = e->Xb;
    } else
    for (int tt = 0; tt < a.nsub; ++tt) {
      if (rows)
        HIP_TRY(e, launch_substep_rows(cur, e->W2, e->uclim + off, e->vclim + off, nxt, e->tabs, e->tab_index, step_tasks,
                                       step_head, n_step_tasks, e->cus * 4, e->nx, e->ny, e->strict, e->stream, (e->xsw & GREB_X_VAPOR_DIFFUSION_ONLY) != 0));
      else
        HIP_TRY(e, launch_substep_fused(cur, e->W2, e->uclim + off, e->vclim + off, nxt, e->tabs, e->tab_index, e->nx,
                                        e->ny, nrun, e->strict, e->stream, (e->xsw & GREB_X_VAPOR_DIFFUSION_ONLY) != 0));
      float* t = cur; cur = nxt; nxt = t;
    }
    MemberArgs b = a;
    b.it0 = it; b.nsteps = 1; // the step kernel derives its clock from it0; year indices are those of `a`
    HIP_TRY(e, launch_physics_step(b, cur, e->Xa, e->red, nrun, e->strict, e->stream));
    if (ityr == kNT && a.yearly)
      HIP_TRY(e, launch_yearly(e->red, a.yearly, e->np, e->nx, a.ipx, a.ipy, a.yearly_years, a.yearly_year0, nrun, e->strict, e->stream));
  }
  return 0;
}
} // namespace

extern "C" {

void greb_params_default(greb_params* p) {
  // src/greb.f90:49-53, 68-104; constant expressions folded in fp32 like the compiler does
  std::memset(p, 0, sizeof(*p));
  p->pi = 3.1416f; p->sig = 5.6704e-8f; p->rho_ocean = 999.1f; p->rho_land = 2600.f; p->rho_air = 1.2f;
  p->cp_ocean = 4186.f; p->cp_land = 926.222f; p->cp_air = 1005.f; p->eps = 1.f;
  p->d_ocean = 50.f; p->d_land = 2.f; p->d_air = 5000.f; p->ct_sens = 22.5f; p->da_ice = 0.25f;
  p->a_no_ice = 0.1f; p->a_cloud = 0.35f;
  p->Tl_ice1 = 273.15f - 10.f; p->Tl_ice2 = 273.15f; p->To_ice1 = 273.15f - 7.f; p->To_ice2 = 273.15f - 1.7f;
  p->co_turb = 5.0f; p->kappa = 8e5f; p->ce = 2e-3f; p->cq_latent = 2.257e6f;
  p->cq_rain = -0.1f / 24.f / 3600.f; p->z_air = 8400.f; p->z_vapor = 5000.f; p->r_qviwv = 2.6736e3f;
  const float pe[10] = {9.0721f, 106.7252f, 61.5562f, 0.0179f, 0.0028f, 0.0570f, 0.3462f, 2.3406f, 0.7032f, 1.0662f};
  std::memcpy(p->p_emi, pe, sizeof(pe));
  p->co2_flux = 298.f;
  p->ipx = 1; p->ipy = 1; p->year0 = 1940; p->dt = 12 * 3600; p->dt_crcl = 1800;
}

const char* greb_engine_last_error(const greb_engine* e) {
  return e ? e->last_error.c_str() : g_last_error.c_str();
}

const char* greb_device_info(int device) {
  static thread_local std::string s;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= device || device < 0) {
    s = "{\"error\": \"no HIP device\"}";
    return s.c_str();
  }
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, device) != hipSuccess) { s = "{\"error\": \"hipGetDeviceProperties\"}"; return s.c_str(); }
  char buf[512];
  std::snprintf(buf, sizeof(buf),
                "{\"name\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"mem_clock_mhz\": %d, "
                "\"hbm_gb\": %.1f, \"lds_per_block\": %zu, \"l2_mb\": %.1f, \"wave\": %d}",
                pr.name, pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1000, pr.memoryClockRate / 1000,
                pr.totalGlobalMem / 1073741824.0, pr.sharedMemPerBlock, pr.l2CacheSize / 1048576.0, pr.warpSize);
  s = buf;
  return s.c_str();
}

int greb_engine_create(const greb_params* p, int nx, int ny, const greb_fields* f, int n_members,
                       const greb_member_overrides* overrides, int device, unsigned flags, greb_engine** out) {
  if (!p || !f || !out || n_members < 1 || nx < 12 || (nx & 3) || ny < 5 || ny > kMaxNy)
    return fail(nullptr, GREB_E_INVALID, "greb_engine_create: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
    return fail(nullptr, GREB_E_NOGPU, "greb_engine_create: no HIP device (the engine has no CPU path)");
  bool fused = true;
  for (int m = 0; m < n_members; ++m) {
    const float kap = (overrides && !std::isnan(overrides[m].kappa)) ? overrides[m].kappa : p->kappa;
    RowTables t; compute_row_tables(*p, kap, nx, ny, t);
    fused = fused && member_layout_supported(t, nx, ny);
  }
  if (p->ipx < 1 || p->ipx > nx || p->ipy < 1 || p->ipy > ny)
    return fail(nullptr, GREB_E_INVALID, "greb_engine_create: ipx/ipy outside the grid");
  greb_engine* e = new (std::nothrow) greb_engine();
  if (!e) return fail(nullptr, GREB_E_INVALID, "out of host memory");
  *out = e; // returned even on failure so last_error can be read; caller destroys
  e->p = *p; e->nx = nx; e->ny = ny; e->np = nx * ny; e->nm = n_members; e->device = device;
  e->strict = (flags & GREB_F_STRICT) != 0;
  e->fused = fused && !(flags & GREB_F_MULTILAUNCH);
  const size_t np = (size_t)e->np, n3 = np * kNT, nm = (size_t)n_members;
  HIP_TRY(e, hipSetDevice(device));
  HIP_TRY(e, hipStreamCreate(&e->stream));

  auto up = [&](float** d, const float* h, size_t n) -> hipError_t {
    hipError_t err = dev_alloc(d, n);
    if (err != hipSuccess) return err;
    return hipMemcpy(*d, h, n * sizeof(float), hipMemcpyHostToDevice);
  };
  HIP_TRY(e, up(&e->z_topo, f->z_topo, np));
  HIP_TRY(e, up(&e->glacier, f->glacier, np));
  HIP_TRY(e, up(&e->sw_solar, f->sw_solar, (size_t)kNT * ny));
  HIP_TRY(e, up(&e->tclim, f->tclim, n3));
  HIP_TRY(e, up(&e->qclim, f->qclim, n3));
  HIP_TRY(e, up(&e->uclim, f->uclim, n3));
  HIP_TRY(e, up(&e->vclim, f->vclim, n3));
  HIP_TRY(e, up(&e->mldclim, f->mldclim, n3));
  HIP_TRY(e, up(&e->cldclim, f->cldclim, n3));
  HIP_TRY(e, up(&e->swetclim, f->swetclim, n3));

  // derived fields (host, once): Toclim :1088-1094, z_ocean :179-183, wz_* :201-202
  std::vector<float> toclim(np), z_ocean(np), wz_air(np), wz_vapor(np);
  for (size_t i = 0; i < np; ++i) {
    float mn = f->tclim[i], mx = 0.f;
    for (int t = 0; t < kNT; ++t) {
      const float v = f->tclim[(size_t)t * np + i]; if (v < mn) mn = v;
      const float d = f->mldclim[(size_t)t * np + i]; if (d > mx) mx = d;
    }
    if (mn - 273.15f < -1.7f) mn = -1.7f + 273.15f;
    toclim[i] = mn;
    z_ocean[i] = 3.0f * mx;
    wz_air[i] = expf(-f->z_topo[i] / p->z_air);
    wz_vapor[i] = expf(-f->z_topo[i] / p->z_vapor);
  }
  HIP_TRY(e, up(&e->toclim, toclim.data(), np));
  HIP_TRY(e, up(&e->z_ocean, z_ocean.data(), np));
  HIP_TRY(e, up(&e->wz_air, wz_air.data(), np));
  HIP_TRY(e, up(&e->wz_vapor, wz_vapor.data(), np));

  // per-member physics, grid tables (deduplicated by kappa), correction-set mapping
  e->h_phys.resize(nm);
  std::vector<int> tab_index(nm), corr_index(nm);
  std::vector<float> kappas;
  e->shared_corr = true;
  for (size_t m = 0; m < nm; ++m) {
    const greb_member_overrides* o = overrides ? overrides + m : nullptr;
    e->h_phys[m] = make_phys(*p, o);
    const float kap = (o && !std::isnan(o->kappa)) ? o->kappa : p->kappa;
    size_t ti = 0;
    for (; ti < kappas.size(); ++ti) if (kappas[ti] == kap) break;
    if (ti == kappas.size()) {
      kappas.push_back(kap);
      RowTables t; compute_row_tables(*p, kap, nx, ny, t);
      e->h_tabs.push_back(t);
    }
    tab_index[m] = (int)ti;
    if (o && !(std::isnan(o->da_ice) && std::isnan(o->a_no_ice) && std::isnan(o->a_cloud) && std::isnan(o->kappa)))
      e->shared_corr = false;
  }
  for (size_t m = 0; m < nm; ++m) corr_index[m] = e->shared_corr ? 0 : (int)m;
  HIP_TRY(e, dev_alloc(&e->phys, nm));
  HIP_TRY(e, hipMemcpy(e->phys, e->h_phys.data(), nm * sizeof(Phys), hipMemcpyHostToDevice));
  HIP_TRY(e, dev_alloc(&e->tabs, e->h_tabs.size()));
  HIP_TRY(e, hipMemcpy(e->tabs, e->h_tabs.data(), e->h_tabs.size() * sizeof(RowTables), hipMemcpyHostToDevice));
  HIP_TRY(e, dev_alloc(&e->tab_index, nm));
  HIP_TRY(e, hipMemcpy(e->tab_index, tab_index.data(), nm * sizeof(int), hipMemcpyHostToDevice));
  e->h_tab_index = tab_index;
  HIP_TRY(e, dev_alloc(&e->corr_index, nm));
  HIP_TRY(e, hipMemcpy(e->corr_index, corr_index.data(), nm * sizeof(int), hipMemcpyHostToDevice));

  const size_t ncorr = e->shared_corr ? 1 : nm;
  HIP_TRY(e, dev_alloc(&e->corr, ncorr * 3 * n3));
  HIP_TRY(e, hipMemset(e->corr, 0, ncorr * 3 * n3 * sizeof(float))); // time_flux = 0 => zero corrections (A.9-10)
  HIP_TRY(e, dev_alloc(&e->acc, nm * 6 * np));
  HIP_TRY(e, hipMemset(e->acc, 0, nm * 6 * np * sizeof(float)));

  // initial state :194-197 and initial cap_surf :190-191
  std::vector<float> st(5 * np);
  const size_t last = (size_t)(kNT - 1) * np;
  HIP_TRY(e, dev_alloc(&e->state, nm * 5 * np));
  for (size_t m = 0; m < nm; ++m) {
    const Phys& P = e->h_phys[m];
    for (size_t i = 0; i < np; ++i) {
      st[i] = f->tclim[last + i]; st[np + i] = st[i]; st[2 * np + i] = toclim[i]; st[3 * np + i] = f->qclim[last + i];
      float c = 0.f;
      if (f->z_topo[i] > 0.f) c = P.cap_land;
      if (f->z_topo[i] <= 0.f) c = P.cap_ocean * f->mldclim[i];
      st[4 * np + i] = c;
    }
    HIP_TRY(e, hipMemcpy(e->state + m * 5 * np, st.data(), 5 * np * sizeof(float), hipMemcpyHostToDevice));
  }
  if (!e->fused) { // the member does not fit one CU (or has another sub-cycling layout): multi-launch engine
    HIP_TRY(e, dev_alloc(&e->Xa, nm * 2 * np));
    HIP_TRY(e, dev_alloc(&e->Xb, nm * 2 * np));
    HIP_TRY(e, dev_alloc(&e->red, nm * np));
    HIP_TRY(e, dev_alloc(&e->W2, 2 * np));
    HIP_TRY(e, hipMemcpy(e->W2, wz_air.data(), np * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->W2 + np, wz_vapor.data(), np * sizeof(float), hipMemcpyHostToDevice));
    static const bool no_step_rows = tuning_int("GREB_NO_STEP_ROWS", 0) != 0; // -DGREB_TUNING builds only (A/B)
    e->step_rows = !no_step_rows && n_members < (1 << (kStepFieldBits - 1)) && // (field and table index share a task word)
                   step_rows_supported(e->h_tabs.data(), (int)e->h_tabs.size(), nx, ny);
    e->step_rows_always = (flags & GREB_F_ROW_STRIPS) != 0;
    static const bool no_persistent = tuning_int("GREB_NO_PERSISTENT", 0) != 0; // -DGREB_TUNING builds only (A/B)
    e->persistent = e->step_rows && !no_persistent && !(flags & GREB_F_NO_PERSISTENT);
    e->persistent_always = e->persistent && (flags & GREB_F_PERSISTENT) != 0;
    if (e->persistent) ledger_register(e);
  }
  return 0;
}

int greb_engine_destroy(greb_engine* e) {
  if (!e) return 0;
  (void)hipSetDevice(e->device); // teardown: nothing useful to do with an error here
  void* ptrs[] = {e->z_topo, e->glacier, e->sw_solar, e->tclim, e->qclim, e->uclim, e->vclim, e->mldclim,
                  e->cldclim, e->swetclim, e->toclim, e->z_ocean, e->wz_air, e->wz_vapor, e->state, e->acc,
                  e->corr, e->corr_index, e->tab_index, e->tabs, e->phys, e->co2_dev, e->monthly_dev, e->yearly_dev,
                  e->Xa, e->Xb, e->red, e->W2};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  for (auto& kv : e->step_tasks) if (kv.second.dev) (void)hipFree(kv.second.dev);
  for (auto& kv : e->circ_orders) circ_rows_free_order(&kv.second);
  if (e->persistent) ledger_release(e);
  for (hipEvent_t ev : e->ev_trial) if (ev) (void)hipEventDestroy(ev);
  for (int i = 0; i < 2; ++i) {
    if (e->ev_done[i]) (void)hipEventDestroy(e->ev_done[i]);
    if (e->ev_free[i]) (void)hipEventDestroy(e->ev_free[i]);
  }
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return 0;
}

int greb_engine_flux_correction(greb_engine* e, int years, float* yearly) {
  if (!e || years < 0) return fail(e, GREB_E_INVALID, "flux_correction: bad argument");
  if (years == 0) return 0;
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t np = (size_t)e->np;
  const int nrun = e->shared_corr ? 1 : e->nm; // identical members: integrate one, broadcast
  if (int rc = ensure(e, &e->yearly_dev, &e->yearly_cap, (size_t)e->nm * years * 2)) return rc;
  HIP_TRY(e, hipMemsetAsync(e->yearly_dev, 0, (size_t)e->nm * years * 2 * sizeof(float), e->stream));
  for (int y = 0; y < years; ++y) {
    MemberArgs a = base_args(e);
    a.flux_phase = 1;
    a.it0 = e->it_flux + 1 + (long long)y * kNT; a.nsteps = kNT;
    a.monthly = nullptr; a.monthly_years = years; a.year_out0 = y;
    a.yearly = e->yearly_dev; a.yearly_years = years; a.yearly_year0 = y;
    if (int rc = run_year(e, a, nrun)) return rc;
  }
  if (e->shared_corr && e->nm > 1) { // the spun-up state (incl. cap_surf) is every member's start (A.8)
    for (int m = 1; m < e->nm; ++m)
      HIP_TRY(e, hipMemcpyAsync(e->state + (size_t)m * 5 * np, e->state, 5 * np * sizeof(float),
                                hipMemcpyDeviceToDevice, e->stream));
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (int rc = check_circulation(e)) return rc;
  e->it_flux += (long long)years * kNT;
  if (yearly) {
    HIP_TRY(e, hipMemcpy(yearly, e->yearly_dev, (size_t)e->nm * years * 2 * sizeof(float), hipMemcpyDeviceToHost));
    if (e->shared_corr)
      for (int m = 1; m < e->nm; ++m) std::memcpy(yearly + (size_t)m * years * 2, yearly, (size_t)years * 2 * sizeof(float));
  }
  return 0;
}

int greb_engine_run(greb_engine* e, int years, const float* co2_ppm, float* monthly, float* yearly,
                    unsigned run_flags) {
  if (!e || years < 1 || !co2_ppm || !monthly) return fail(e, GREB_E_INVALID, "run: bad argument");
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t np = (size_t)e->np, nm = (size_t)e->nm;
  const size_t rec_year = 12 * 5 * np; // floats per member-year
  const bool dev_out = (run_flags & GREB_RUN_DEVICE_OUT) != 0;
  if (int rc = ensure(e, &e->co2_dev, &e->co2_cap, nm * years)) return rc;
  HIP_TRY(e, hipMemcpyAsync(e->co2_dev, co2_ppm, nm * years * sizeof(float), hipMemcpyHostToDevice, e->stream));
  if (int rc = ensure(e, &e->yearly_dev, &e->yearly_cap, nm * years * 2)) return rc;
  HIP_TRY(e, hipMemsetAsync(e->yearly_dev, 0, nm * years * 2 * sizeof(float), e->stream));
  if (dev_out) {
    for (int y = 0; y < years; ++y) {
      MemberArgs a = base_args(e);
      a.flux_phase = 0;
      a.it0 = e->it_scnr + 1 + (long long)y * kNT; a.nsteps = kNT;
      a.co2 = e->co2_dev; a.co2_stride = years; a.co2_year0 = y;
      a.monthly = monthly; a.monthly_years = years; a.year_out0 = y;
      a.yearly = e->yearly_dev; a.yearly_years = years; a.yearly_year0 = y;
      if (int rc = run_year(e, a, e->nm)) return rc;
    }
  } else {
    // Host delivery: year y's records leave over PCIe on the copy stream while year y+1 integrates on the compute
    // stream (two staging slots of one model year each, [member][12][5][np]); the host side is strided by the
    // caller's [member][years] layout.  A pinned `monthly` makes the copies true DMA; a pageable one is staged by
    // the runtime and still overlaps the kernels.
    const size_t slot = nm * rec_year;
    if (int rc = ensure(e, &e->monthly_dev, &e->monthly_cap, 2 * slot)) return rc;
    if (!e->copy_stream) HIP_TRY(e, hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      if (!e->ev_done[i]) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_done[i], hipEventDisableTiming));
      if (!e->ev_free[i]) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_free[i], hipEventDisableTiming));
    }
    // copy of year y: issued AFTER year y+1's kernels are enqueued, so that even a copy the runtime performs
    // synchronously (pageable destination) runs beside a kernel
    auto deliver = [&](int y) -> int {
      const int sl = y & 1;
      HIP_TRY(e, hipStreamWaitEvent(e->copy_stream, e->ev_done[sl], 0));
      HIP_TRY(e, hipMemcpy2DAsync(monthly + (size_t)y * rec_year, (size_t)years * rec_year * sizeof(float),
                                  e->monthly_dev + (size_t)sl * slot, rec_year * sizeof(float), rec_year * sizeof(float),
                                  nm, hipMemcpyDeviceToHost, e->copy_stream));
      HIP_TRY(e, hipEventRecord(e->ev_free[sl], e->copy_stream));
      return 0;
    };
    // an error anywhere below must not return while copies into the CALLER's buffer are still in flight (the caller
    // may free it as soon as it sees the error): the body runs in a lambda and both streams are drained on failure
    const int rc_years = [&]() -> int {
    for (int y = 0; y < years; ++y) {
      const int sl = y & 1;
      if (y >= 2) HIP_TRY(e, hipStreamWaitEvent(e->stream, e->ev_free[sl], 0)); // slot's previous year has left
      MemberArgs a = base_args(e);
      a.flux_phase = 0;
      a.it0 = e->it_scnr + 1 + (long long)y * kNT; a.nsteps = kNT;
      a.co2 = e->co2_dev; a.co2_stride = years; a.co2_year0 = y;
      a.monthly = e->monthly_dev + (size_t)sl * slot; a.monthly_years = 1; a.year_out0 = 0;
      a.yearly = e->yearly_dev; a.yearly_years = years; a.yearly_year0 = y;
      if (int rc = run_year(e, a, e->nm)) return rc;
      HIP_TRY(e, hipEventRecord(e->ev_done[sl], e->stream));
      if (y > 0) if (int rc = deliver(y - 1)) return rc;
    }
    if (int rc = deliver(years - 1)) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->copy_stream));
    return 0;
    }();
    if (rc_years) {
      (void)hipStreamSynchronize(e->copy_stream);
      (void)hipStreamSynchronize(e->stream);
      return rc_years;
    }
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (int rc = check_circulation(e)) return rc;
  e->it_scnr += (long long)years * kNT;
  if (yearly) HIP_TRY(e, hipMemcpy(yearly, e->yearly_dev, nm * years * 2 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

const char* greb_engine_describe(greb_engine* e) {
  static thread_local std::string s;
  if (!e) { s = "{}"; return s.c_str(); }
  char buf[256];
  std::snprintf(buf, sizeof(buf), "{\"grid\": [%d, %d], \"members\": %d, \"arithmetic\": \"%s\", \"engine\": \"%s\"", e->nx, e->ny, e->nm,
                e->strict ? "strict" : "fast", e->fused ? "fused member kernel" : (e->step_rows ? "row strips" : "latitude bands"));
  s = buf;
  if (e->persistent) {
    std::snprintf(buf, sizeof(buf), ", \"wavefront_slots_granted\": %d, \"circulation\": [", e->slots_granted);
    s += buf;
    bool first = true;
    for (const auto& kv : e->circ_orders) {
      const auto ft = e->circ_form.find(kv.first);
      const int form = e->persistent_always ? 2 : (ft == e->circ_form.end() ? 0 : ft->second.form);
      std::snprintf(buf, sizeof(buf), "%s{\"members_run\": %d, \"tasks_of_one_launch_per_call\": %d, \"form\": \"%s\", \"trial_ms_per_3_steps\": [%.4f, %.4f]}",
                    first ? "" : ", ", kv.first, kv.second.n,
                    kv.second.n == 0 ? "one launch per sub-step (slots)" : (form == 2 ? "one launch per call" : (form == 1 ? "one launch per sub-step" : "undecided")),
                    ft == e->circ_form.end() ? 0.f : ft->second.ms_substep, ft == e->circ_form.end() ? 0.f : ft->second.ms_call);
      s += buf;
      first = false;
    }
    s += "]";
  }
  s += "}";
  return s.c_str();
}

int greb_engine_get_state(greb_engine* e, int member, float* state5) {
  if (!e || member < 0 || member >= e->nm || !state5) return fail(e, GREB_E_INVALID, "get_state: bad argument");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemcpy(state5, e->state + (size_t)member * 5 * e->np, (size_t)5 * e->np * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int greb_engine_get_corrections(greb_engine* e, int member, float* corr, float* state5) {
  if (!e || member < 0 || member >= e->nm) return fail(e, GREB_E_INVALID, "get_corrections: bad argument");
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t n = (size_t)3 * kNT * e->np, ci = e->shared_corr ? 0 : (size_t)member;
  if (corr) HIP_TRY(e, hipMemcpy(corr, e->corr + ci * n, n * sizeof(float), hipMemcpyDeviceToHost));
  if (state5) return greb_engine_get_state(e, member, state5);
  return 0;
}

static_assert(GREB_X_NO_ICE == kXNoIce && GREB_X_NO_HYDRO == kXNoHydro && GREB_X_NO_DEEP_OCEAN == kXNoDeepOcean &&
              GREB_X_LW_LINEAR_VAPOR == kXLwLinear && GREB_X_NO_CIRCULATION == kXNoCirc &&
              GREB_X_NO_VAPOR_TRANSPORT == kXNoQTransport && GREB_X_VAPOR_DIFFUSION_ONLY == kXQDiffOnly &&
              GREB_X_SST_PLUS1 == kXSstPlus1, "device switch constants mirror the ABI");

// greb.original.model.f90: which process each log_exp value switches off (the conditions are the original's)
unsigned greb_log_exp_switches(int le) {
  unsigned x = 0;
  if (le <= 5) x |= GREB_X_NO_ICE;                                        // :394, :492
  if (le <= 6 || le == 13 || le == 15) x |= GREB_X_NO_HYDRO;              // :453
  if (le <= 9 || le == 11 || (le >= 14 && le <= 16)) x |= GREB_X_NO_DEEP_OCEAN; // :514-515
  if (le == 11) x |= GREB_X_LW_LINEAR_VAPOR;                              // :423, :430
  if (le <= 4) x |= GREB_X_NO_CIRCULATION;                                // :553
  if (le == 7 || le == 16) x |= GREB_X_NO_VAPOR_TRANSPORT;                // :554-555
  if (le == 8) x |= GREB_X_VAPOR_DIFFUSION_ONLY;                          // :560
  if (le >= 14 && le <= 16) x |= GREB_X_SST_PLUS1;                        // :226
  return x;
}

int greb_engine_set_experiment(greb_engine* e, unsigned switches) {
  if (!e || (switches & ~0xffu)) return fail(e, GREB_E_INVALID, "set_experiment: unknown switch bits");
  e->xsw = switches;
  return 0;
}

int greb_engine_set_corrections(greb_engine* e, int member, const float* corr, const float* state5) {
  if (!e || member < -1 || member >= e->nm) return fail(e, GREB_E_INVALID, "set_corrections: bad argument");
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t n = (size_t)3 * kNT * e->np, s5 = (size_t)5 * e->np;
  const int m0 = member < 0 ? 0 : member, m1 = member < 0 ? e->nm : member + 1; // -1 = every member
  for (int m = m0; m < m1; ++m) {
    const size_t ci = e->shared_corr ? 0 : (size_t)m;
    if (corr && (ci == (size_t)m || m == m0)) HIP_TRY(e, hipMemcpy(e->corr + ci * n, corr, n * sizeof(float), hipMemcpyHostToDevice));
    if (state5) HIP_TRY(e, hipMemcpy(e->state + (size_t)m * s5, state5, s5 * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

int greb_engine_set_state(greb_engine* e, int member, const float* state5) {
  if (!state5) return fail(e, GREB_E_INVALID, "set_state: bad argument");
  return greb_engine_set_corrections(e, member, nullptr, state5);
}

// ---------------------------------------------------------------- batched single routines
namespace {
struct DevBuf {
  float* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};
int batched_common(const greb_params* p, int nx, int ny, int batch, int device) {
  if (!p || nx < 12 || (nx & 3) || ny < 5 || ny > kMaxNy || batch < 1) return fail(nullptr, GREB_E_INVALID, "batched: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev)
    return fail(nullptr, GREB_E_NOGPU, "batched: no HIP device (no CPU path)");
  hipError_t err = hipSetDevice(device);
  if (err != hipSuccess) return fail(nullptr, (int)err, "hipSetDevice");
  return 0;
}
} // namespace

#define HIP_TRY0(expr) HIP_TRY(nullptr, expr)

namespace {
// device copies of row tables for greb_diffusion_batched_dev: one entry per (device, table contents), IMMUTABLE once
// uploaded -- a sweep in flight on any stream never sees its table change under it (the upload of a new table goes to
// a new buffer; the synchronous copy completes before the call that made it launches anything) -- and freed only by
// greb_release_caches(), which waits for the device first.
struct TabCache {
  struct Entry { int device; RowTables* dev; RowTables host; unsigned long long used; };
  std::mutex mu;
  std::vector<Entry> entries;
  unsigned long long clock = 0;
  static constexpr size_t kMax = 32; // a long-lived host that sweeps kappa: the least recently used table is retired
} g_tab_cache;
} // namespace

#ifdef GREB_TUNING
// diagnostic builds only (not part of include/greb_engine.h): where the fused member kernel writes its stamps,
// [n_members][8 waves][8] unsigned long long on the device (tools/stamp_member.py); NULL switches them off
int greb_tuning_set_stamps(greb_engine* e, unsigned long long* stamps_dev) {
  if (!e) return GREB_E_INVALID;
  e->stamps = stamps_dev;
  return 0;
}
#endif

int greb_release_caches(void) {
  rows_release_cache();
  std::lock_guard<std::mutex> lock(g_tab_cache.mu);
  int prev = 0;
  const bool have_prev = hipGetDevice(&prev) == hipSuccess;
  for (auto& e : g_tab_cache.entries)
    if (e.dev && hipSetDevice(e.device) == hipSuccess) { (void)hipDeviceSynchronize(); (void)hipFree(e.dev); }
  g_tab_cache.entries.clear();
  if (have_prev) (void)hipSetDevice(prev);
  return 0;
}

int greb_diffusion_batched_dev(const greb_params* p, int nx, int ny, int batch, const float* T1_dev,
                               const float* wz_dev, float* dX_dev, int strict, int sweeps, void* stream) {
  if (!p || nx < 12 || (nx & 3) || ny < 5 || ny > kMaxNy || batch < 1 || sweeps < 1)
    return fail(nullptr, GREB_E_INVALID, "diffusion_batched_dev: bad argument");
  // The row table lives in a small device buffer cached per device and table contents (the launches must not be
  // separated by an allocation or a copy: this entry point is what the HBM-roofline measurement times).  The operands
  // must live on the calling thread's current device.
  int dev = 0;
  HIP_TRY0(hipGetDevice(&dev));
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, T1_dev) != hipSuccess || attr.device != dev) {
    (void)hipGetLastError();
    return fail(nullptr, GREB_E_INVALID, "diffusion_batched_dev: T1_dev is not a device pointer of the current device");
  }
  RowTables t; compute_row_tables(*p, p->kappa, nx, ny, t);
  RowTables* tab_dev = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_tab_cache.mu);
    for (TabCache::Entry& ce : g_tab_cache.entries)
      if (ce.device == dev && std::memcmp(&t, &ce.host, sizeof(t)) == 0) { tab_dev = ce.dev; ce.used = ++g_tab_cache.clock; break; }
    if (!tab_dev && g_tab_cache.entries.size() >= TabCache::kMax) { // (once the device is idle: a sweep in flight may read it)
      size_t lru = g_tab_cache.entries.size();
      for (size_t i = 0; i < g_tab_cache.entries.size(); ++i)
        if (g_tab_cache.entries[i].device == dev && (lru == g_tab_cache.entries.size() || g_tab_cache.entries[i].used < g_tab_cache.entries[lru].used)) lru = i;
      if (lru < g_tab_cache.entries.size()) {
        HIP_TRY0(hipDeviceSynchronize());
        (void)hipFree(g_tab_cache.entries[lru].dev);
        g_tab_cache.entries.erase(g_tab_cache.entries.begin() + (long)lru);
      }
    }
    if (!tab_dev) {
      TabCache::Entry ce{dev, nullptr, t, ++g_tab_cache.clock};
      HIP_TRY0(dev_alloc(&ce.dev, 1));
      hipError_t he = hipMemcpy(ce.dev, &t, sizeof(t), hipMemcpyHostToDevice);
      if (he != hipSuccess) { (void)hipFree(ce.dev); HIP_TRY0(he); }
      g_tab_cache.entries.push_back(ce);
      tab_dev = ce.dev;
    }
  }
  for (int i = 0; i < sweeps; ++i)
    HIP_TRY0(launch_diffusion(T1_dev, wz_dev, dX_dev, tab_dev, t, nx, ny, batch, strict != 0, (hipStream_t)stream));
  return 0;
}

int greb_diffusion_launch_order(const greb_params* p, int nx, int ny, int batch, int* field, int* k0, int* k1, int* up,
                                int capacity) {
  if (!p || nx < 12 || (nx & 3) || ny < 5 || ny > kMaxNy || batch < 1 || capacity < 0 ||
      (capacity > 0 && (!field || !k0 || !k1 || !up)))
    return fail(nullptr, GREB_E_INVALID, "diffusion_launch_order: bad argument");
  RowTables t; compute_row_tables(*p, p->kappa, nx, ny, t);
  if (!rows_supported(t, nx, ny)) return 0;
  std::vector<RowsTask> tasks;
  rows_tasks(t, ny, batch, rows_default_tuning(), tasks);
  for (size_t i = 0; i < tasks.size() && (int)i < capacity; ++i) {
    field[i] = tasks[i].field; k0[i] = tasks[i].rows & 0xff; k1[i] = (tasks[i].rows >> 8) & 0x1ff;
    up[i] = (tasks[i].rows & kRowsUp) != 0;
  }
  return (int)tasks.size();
}

int greb_substep_launch_order(const greb_params* p, int nx, int ny, int n_members, const float* kappa, int* field, int* k0,
                              int* k1, int capacity) {
  if (!p || nx < 12 || (nx & 3) || ny < 5 || ny > kMaxNy || n_members < 1 || capacity < 0 || (capacity > 0 && (!field || !k0 || !k1)))
    return fail(nullptr, GREB_E_INVALID, "substep_launch_order: bad argument");
  std::vector<RowTables> tabs((size_t)n_members);
  std::vector<int> idx((size_t)n_members);
  for (int m = 0; m < n_members; ++m) { compute_row_tables(*p, kappa ? kappa[m] : p->kappa, nx, ny, tabs[m]); idx[m] = m; }
  if (n_members >= (1 << (kStepFieldBits - 1)) || !step_rows_supported(tabs.data(), n_members, nx, ny)) return 0;
  std::vector<RowsTask> tasks;
  step_rows_tasks(tabs.data(), idx.data(), n_members, ny, 256 * kStepRowsSlotsPerCu, tasks); // an MI355X: 256 CUs
  for (size_t i = 0; i < tasks.size() && (int)i < capacity; ++i) {
    field[i] = tasks[i].field & ((1 << kStepFieldBits) - 1); k0[i] = tasks[i].rows & 0xff; k1[i] = (tasks[i].rows >> 8) & 0x1ff;
  }
  return (int)tasks.size();
}

int greb_circulation_launch_plan(const greb_params* p, int nx, int ny, int n_members, const float* kappa, int slots,
                                 int* field, int* k0, int* k1, int* chain, int* dep4, int capacity) {
  if (!p || nx < 12 || (nx & 3) || ny < 5 || ny > kMaxNy || n_members < 1 || slots < 1 || capacity < 0 ||
      (capacity > 0 && (!field || !k0 || !k1 || !chain || !dep4)))
    return fail(nullptr, GREB_E_INVALID, "circulation_launch_plan: bad argument");
  std::vector<RowTables> tabs((size_t)n_members);
  std::vector<int> idx((size_t)n_members);
  for (int m = 0; m < n_members; ++m) { compute_row_tables(*p, kappa ? kappa[m] : p->kappa, nx, ny, tabs[m]); idx[m] = m; }
  if (n_members >= (1 << (kStepFieldBits - 1)) || !step_rows_supported(tabs.data(), n_members, nx, ny)) return 0;
  std::vector<CircTask> tasks;
  circ_rows_tasks(tabs.data(), idx.data(), n_members, ny, slots, tasks);
  if ((int)tasks.size() > slots) return 0;
  for (size_t i = 0; i < tasks.size() && (int)i < capacity; ++i) {
    field[i] = tasks[i].field & ((1 << kStepFieldBits) - 1); k0[i] = tasks[i].rows & 0xff; k1[i] = (tasks[i].rows >> 8) & 0x1ff;
    chain[i] = (tasks[i].rows & kCircChain) != 0;
    for (int j = 0; j < 4; ++j) dep4[4 * i + j] = tasks[i].dep[j];
  }
  return (int)tasks.size();
}

int greb_diffusion_batched(const greb_params* p, int nx, int ny, int batch, const float* T1, const float* wz,
                           float* dX, int strict, int device) {
  if (int rc = batched_common(p, nx, ny, batch, device)) return rc;
  const size_t n = (size_t)batch * nx * ny;
  DevBuf a, b, c;
  HIP_TRY0(dev_alloc(&a.p, n)); HIP_TRY0(dev_alloc(&b.p, n)); HIP_TRY0(dev_alloc(&c.p, n));
  HIP_TRY0(hipMemcpy(a.p, T1, n * 4, hipMemcpyHostToDevice));
  HIP_TRY0(hipMemcpy(b.p, wz, n * 4, hipMemcpyHostToDevice));
  if (int rc = greb_diffusion_batched_dev(p, nx, ny, batch, a.p, b.p, c.p, strict, 1, nullptr)) return rc;
  HIP_TRY0(hipDeviceSynchronize());
  HIP_TRY0(hipMemcpy(dX, c.p, n * 4, hipMemcpyDeviceToHost));
  return 0;
}

static int adv_or_circ(bool circ, const greb_params* p, int nx, int ny, int batch, const float* X, const float* wz,
                       const float* u, const float* v, float* dX, int strict, int device) {
  if (int rc = batched_common(p, nx, ny, batch, device)) return rc;
  const size_t n = (size_t)batch * nx * ny;
  DevBuf a, b, c, d, o, scr;
  RowTables t; compute_row_tables(*p, p->kappa, nx, ny, t);
  RowTables* tab_dev = nullptr;
  HIP_TRY0(dev_alloc(&tab_dev, 1));
  DevBuf tabhold; tabhold.p = reinterpret_cast<float*>(tab_dev);
  HIP_TRY0(hipMemcpy(tab_dev, &t, sizeof(t), hipMemcpyHostToDevice));
  HIP_TRY0(dev_alloc(&a.p, n)); HIP_TRY0(dev_alloc(&b.p, n)); HIP_TRY0(dev_alloc(&c.p, n));
  HIP_TRY0(dev_alloc(&d.p, n)); HIP_TRY0(dev_alloc(&o.p, n));
  HIP_TRY0(hipMemcpy(a.p, X, n * 4, hipMemcpyHostToDevice));
  HIP_TRY0(hipMemcpy(b.p, wz, n * 4, hipMemcpyHostToDevice));
  HIP_TRY0(hipMemcpy(c.p, u, n * 4, hipMemcpyHostToDevice));
  HIP_TRY0(hipMemcpy(d.p, v, n * 4, hipMemcpyHostToDevice));
  if (circ) {
    HIP_TRY0(dev_alloc(&scr.p, 3 * n));
    HIP_TRY0(launch_circulation(a.p, b.p, c.p, d.p, o.p, scr.p, tab_dev, t, nx, ny, batch, nsub_of(*p), strict != 0, nullptr));
  } else {
    HIP_TRY0(launch_advection(a.p, b.p, c.p, d.p, o.p, tab_dev, nx, ny, batch, strict != 0, nullptr));
  }
  HIP_TRY0(hipDeviceSynchronize());
  HIP_TRY0(hipMemcpy(dX, o.p, n * 4, hipMemcpyDeviceToHost));
  return 0;
}

int greb_advection_batched(const greb_params* p, int nx, int ny, int batch, const float* T1, const float* wz,
                           const float* u, const float* v, float* dX, int strict, int device) {
  return adv_or_circ(false, p, nx, ny, batch, T1, wz, u, v, dX, strict, device);
}

int greb_circulation_batched(const greb_params* p, int nx, int ny, int batch, const float* X, const float* wz,
                             const float* u, const float* v, float* dX, int strict, int device) {
  return adv_or_circ(true, p, nx, ny, batch, X, wz, u, v, dX, strict, device);
}

int greb_engine_point_physics(greb_engine* e, int ityr, float co2, const float* in5, float* out15) {
  if (!e || ityr < 1 || ityr > kNT || !in5 || !out15) return fail(e, GREB_E_INVALID, "point_physics: bad argument");
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t np = (size_t)e->np;
  DevBuf in, out;
  HIP_TRY(e, dev_alloc(&in.p, 5 * np)); HIP_TRY(e, dev_alloc(&out.p, 15 * np));
  HIP_TRY(e, hipMemcpy(in.p, in5, 5 * np * 4, hipMemcpyHostToDevice));
  PointArgs a{};
  a.nx = e->nx; a.ny = e->ny; a.np = e->np; a.ityr = ityr; a.co2 = co2;
  a.z_topo = e->z_topo; a.glacier = e->glacier; a.sw_solar = e->sw_solar; a.tclim = e->tclim;
  a.uclim = e->uclim; a.vclim = e->vclim; a.mldclim = e->mldclim; a.cldclim = e->cldclim; a.swetclim = e->swetclim;
  a.z_ocean = e->z_ocean; a.wz_air = e->wz_air; a.phys = e->h_phys[0]; a.in5 = in.p; a.out15 = out.p;
  a.xsw = e->xsw; a.qclim = e->qclim;
  HIP_TRY(e, launch_point_physics(a, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  HIP_TRY(e, hipMemcpy(out15, out.p, 15 * np * 4, hipMemcpyDeviceToHost));
  return 0;
}

} // extern "C"
