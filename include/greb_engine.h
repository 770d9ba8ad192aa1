/* greb_engine.h -- C ABI of the MI355X-native GREB time-integration engine.
 *
 * The reference (sieste/greb-climate-model) has no plugin / FFI interface: its routines are
 * external Fortran procedures that talk through module globals.  The drop-in boundary is
 * therefore the natural seam SURVEY.md 8(b) identifies -- the two time loops
 *     src/greb.f90:325-362   (qflux_correction: flux-correction phase)
 *     src/greb.f90:228-234   (greb_model: scenario phase, one time_loop call per step)
 * and everything the reference hands across that seam today through modules mo_numerics /
 * mo_physics / mo_diagnostics (src/greb.f90:32-158) is passed here explicitly.
 *
 * A thin Fortran host (greb_climate_model_amd/host/greb_host.f90, iso_c_binding) keeps the
 * reference's CLI / namelist / input-file / output-record conventions and calls these entry
 * points; INTEGRATION.md shows the bind(C) interface block.  Everything is plain C: pointers,
 * ints, floats; no C++ or torch types.  All arrays are IEEE fp32 in the reference's own memory
 * order (Fortran column-major == C [t][lat][lon], longitude fastest, latitude row 0 = south).
 *
 * Error convention: every entry returns int; 0 = ok, <0 = engine error (GREB_E_*), >0 = a
 * hipError_t passed through.  greb_engine_last_error() gives a message.  Nothing throws or
 * exits across the ABI.  An engine is used from one host thread at a time.
 */
#ifndef GREB_ENGINE_H
#define GREB_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GREB_NSTEP_YR 730 /* ndays_yr*ndt_days, src/greb.f90:37-41 */
#define GREB_NVAR_OUT 5   /* Tsurf, Tair, Tocean, q, albedo: src/greb.f90:978-982 */

#define GREB_E_INVALID   (-1) /* bad argument / shape */
#define GREB_E_NOGPU     (-2) /* no HIP device: the product path has no CPU fallback */
#define GREB_E_STATE     (-3) /* call order (e.g. run before create) */
#define GREB_E_UNSUPPORTED (-4)

/* namelist group physics_par in declaration order (src/greb.f90:68-101,128-132),
 * then co2_flux (:104), numerics (:51-53) and the two integer time steps (:38-39). */
typedef struct greb_params {
  float pi, sig, rho_ocean, rho_land, rho_air, cp_ocean, cp_land, cp_air, eps;
  float d_ocean, d_land, d_air, ct_sens, da_ice, a_no_ice, a_cloud;
  float Tl_ice1, Tl_ice2, To_ice1, To_ice2;
  float co_turb, kappa, ce, cq_latent, cq_rain, z_air, z_vapor, r_qviwv;
  float p_emi[10];
  float co2_flux;
  int32_t ipx, ipy;   /* 1-based diagnostic point, src/greb.f90:51-52,954 */
  int32_t year0;
  int32_t dt;         /* 43200 */
  int32_t dt_crcl;    /* 1800  */
} greb_params;

/* Fill with the reference defaults (src/greb.f90:49-53,68-104). */
void greb_params_default(greb_params* p);

/* The reference's input set, host pointers (src/greb.f90:1073-1085). */
typedef struct greb_fields {
  const float* z_topo;   /* [ny][nx]       input/topography     */
  const float* glacier;  /* [ny][nx]       input/glacier.masks  */
  const float* sw_solar; /* [730][ny]      input/solar.radiation */
  const float* tclim;    /* [730][ny][nx]  input/tsurf          */
  const float* qclim;    /*                input/vapor          */
  const float* uclim;    /*                input/zonal.wind     */
  const float* vclim;    /*                input/meridional.wind */
  const float* mldclim;  /*                input/ocean.mld      */
  const float* cldclim;  /*                input/cloud.cover    */
  const float* swetclim; /*                input/soil.moisture  */
} greb_fields;

/* Per-member physics overrides for perturbed-physics ensembles (BASELINE config 5).
 * A member is what a separate `ens_id` process is in the reference (src/greb.f90:153,1064-1068).
 * NaN in a slot = keep the engine-wide greb_params value. */
typedef struct greb_member_overrides {
  float da_ice, a_no_ice, a_cloud, kappa;
} greb_member_overrides;

/* engine flags */
#define GREB_F_STRICT 1u /* reference operation order, IEEE division, no FMA contraction
                            (bit-exact stencils; default is the restructured fast arithmetic) */
#define GREB_F_MULTILAUNCH 2u /* force the any-grid engine (one launch per circulation sub-step) even
                                 where the fused one-CU-per-member kernel applies (96x48); grids that
                                 do not fit one CU, e.g. 384x192, always use it */

#define GREB_F_ROW_STRIPS 4u /* 384-wide grids: FAST arithmetic takes the row-strip form of the circulation
                                (greb_step_rows.hip: one wavefront per strip of rows, no workgroup barrier) by default
                                at every member count; this flag extends it to STRICT arithmetic, which otherwise keeps
                                the band kernel (greb_kernels.hip: sweep_kernel<fused>).  The two are bit-identical in
                                STRICT (tests/test_gpu_parity.py::test_row_strip_substep_equals_band_kernel_strict) */
/* 384-wide grids, row strips: the circulation call (src/greb.f90:546-550: 24 sub-steps inside one call) runs either
 * as one launch per SUB-STEP (greb_step_rows.hip) or as ONE launch per call (greb_circ_rows.hip), whose strips hand
 * their rows to each other through memory flags and therefore need every strip of the launch resident at once (asserted
 * against the device's wavefront slots; an engine created while other engines of the process hold the slots of its device
 * takes one launch per sub-step by itself).  The two are bit-identical; which is faster depends on the member count, so
 * by default the engine times both during the first eight model steps it integrates and keeps the faster. */
#define GREB_F_NO_PERSISTENT 8u /* always one launch per sub-step */
#define GREB_F_PERSISTENT 16u   /* one launch per call wherever it can be resident, without the trial */

typedef struct greb_engine greb_engine;

/* Create an engine for n_members ensemble members on HIP device `device`.
 * Any grid with nx % 4 == 0, nx >= 12, 5 <= ny <= 192 (src/greb.f90:36 is the only thing that fixes the grid in the
 * reference).  Which kernels a grid gets:
 *   96x48 with the default sub-cycling layout   the fused member kernel (a whole member resident in one compute unit);
 *   nx = 384 or nx = 192                        the any-grid engine on wavefront-sized row strips (a lane owns six
 *                                               longitudes; a 192-wide row is laid twice around the wavefront);
 *   anything else                               the any-grid engine on latitude bands staged in LDS.
 * What bounds ny: the per-row tables (sub-cycle counts and constants, src/greb.f90:578-582, 652-654, 838-840) are
 * fixed-size arrays of 192 rows that travel by value in kernel arguments and sit in LDS, and a strip's rows are packed
 * k0 in 8 and k1 in 9 bits of a task word; ny > 192 is GREB_E_INVALID, not a slower path.
 * Copies the inputs to HBM, computes the derived fields of greb_model's preamble
 * (src/greb.f90:176-216) and Toclim (src/greb.f90:1088-1094) and sets every member's
 * state to the initial state (src/greb.f90:194-197).  overrides may be NULL. */
int greb_engine_create(const greb_params* p, int nx, int ny, const greb_fields* f, int n_members,
                       const greb_member_overrides* overrides, int device, unsigned flags,
                       greb_engine** out);

/* qflux_correction (src/greb.f90:311-364): `years`*730 steps at co2_flux; leaves the
 * correction arrays, cap_surf and the spun-up state in the engine.
 * yearly may be NULL, else [n_members][years][2] = {global-mean Tsurf, Tsurf(ipx,ipy)} in
 * deg C as printed at src/greb.f90:954. */
int greb_engine_flux_correction(greb_engine* e, int years, float* yearly);

/* What the engine is and which kernels it has settled on, as a JSON object (valid until the next call from this thread). */
const char* greb_engine_describe(greb_engine* e);

/* Scenario run (src/greb.f90:226-234 + time_loop :239-274): `years`*730 steps.
 *   co2_ppm : [n_members][years]   annual CO2, already padded (src/greb.f90:1053-1061)
 *   monthly : [n_members][years][12][5][ny][nx]  monthly means in file-record order
 *             (src/greb.f90:978-982); host memory unless GREB_RUN_DEVICE_OUT
 *   yearly  : [n_members][years][2] as above (may be NULL)
 * May be called repeatedly; the model clock (it, year, month accumulators) continues. */
#define GREB_RUN_DEVICE_OUT 1u /* `monthly` is a device pointer (e.g. for an RCCL gather) */
int greb_engine_run(greb_engine* e, int years, const float* co2_ppm, float* monthly, float* yearly,
                    unsigned run_flags);

/* ---- sensitivity-experiment switches -----------------------------------------
 * Runtime switches on the same kernels that reproduce the `log_exp` experiments of the upstream model
 * variant (src/greb.original.model.f90:60,162-166,394,423-430,452-453,492-495,513-515,553-571; doc in its
 * namelist_original).  greb_log_exp_switches() maps a log_exp value to the process switches below; the
 * experiment's changes to the BOUNDARY DATA (constant topography / clouds / vapour / mixed layer, :162-166),
 * its CO2 series (A1B ramp, :939-951) and the run sequencing (control run, :208-215) stay with the host
 * (greb_climate_model_amd/original.py shows them).  Default 0 = the complete model = src/greb.f90. */
#define GREB_X_NO_ICE            (1u << 0) /* log_exp <= 5: a_surf = a_no_ice (:394); heat capacity ignores sea ice (:492-495) */
#define GREB_X_NO_HYDRO          (1u << 1) /* <= 6, 13, 15: no latent heat, evaporation, rain (:452-453) */
#define GREB_X_NO_DEEP_OCEAN     (1u << 2) /* <= 9, 11, 14-16: dT_ocean = dTo = 0 (:513-515) */
#define GREB_X_LW_LINEAR_VAPOR   (1u << 3) /* 11: emissivity linear in q around qclim (:423,:430) */
#define GREB_X_NO_CIRCULATION    (1u << 4) /* <= 4: no transport of Tair and q (:553; the original's increment is
                                              unassigned there -- defined as 0 here) */
#define GREB_X_NO_VAPOR_TRANSPORT (1u << 5) /* 7, 16: no transport of q (:554-555; same remark) */
#define GREB_X_VAPOR_DIFFUSION_ONLY (1u << 6) /* 8: q is diffused but not advected (:560-564) */
#define GREB_X_SST_PLUS1         (1u << 7) /* 14-16: before every scenario step Tsurf(ocean) = Tclim(previous step's
                                              slice) + 1 K (:226, evaluated before time_loop updates ityr);
                                              not applied in the flux-correction phase.  The host clears it for
                                              the control run and passes CO2 = CO2_ctrl (:225) */
unsigned greb_log_exp_switches(int log_exp);
/* Takes effect from the next flux_correction / run call. */
int greb_engine_set_experiment(greb_engine* e, unsigned switches);

/* Flux-correction cache: TF/qF/ToF_correct [3][730][ny][nx] + cap_surf +
 * the four state fields Ts,Ta,To,q [5][ny][nx] of one member. */
int greb_engine_get_corrections(greb_engine* e, int member, float* corr, float* state5);
int greb_engine_set_corrections(greb_engine* e, int member, const float* corr, const float* state5);

/* Current state of one member: Ts, Ta, To, q, cap_surf [5][ny][nx]; set: member -1 = every member
 * (== greb_engine_set_corrections(e, member, NULL, state5), for hosts that cannot pass a null array). */
int greb_engine_get_state(greb_engine* e, int member, float* state5);
int greb_engine_set_state(greb_engine* e, int member, const float* state5);

const char* greb_engine_last_error(const greb_engine* e);
int greb_engine_destroy(greb_engine* e);

/* Device + build info as a static JSON string (CU count, clocks, arch). */
const char* greb_device_info(int device);

/* ---- single-routine entry points over a batch dimension (tests + roofline bench) --------
 * Mirrors of the reference routines' signatures with a leading batch count; all pointers are
 * HOST pointers unless the name ends in _dev.  `p` supplies pi/kappa/dt_crcl.
 *   diffusion  src/greb.f90:556-723   dX = wz*(dTx+dTy)
 *   advection  src/greb.f90:726-915   winds u,v are the raw climatology slice (sign split inside)
 *   circulation src/greb.f90:528-553  24 sub-steps of X += diffusion + advection
 * Shapes: T1, wz, dX, u, v : [batch][ny][nx].  strict: 0/1 as GREB_F_STRICT. */
int greb_diffusion_batched(const greb_params* p, int nx, int ny, int batch, const float* T1,
                           const float* wz, float* dX, int strict, int device);
int greb_advection_batched(const greb_params* p, int nx, int ny, int batch, const float* T1,
                           const float* wz, const float* u, const float* v, float* dX, int strict,
                           int device);
int greb_circulation_batched(const greb_params* p, int nx, int ny, int batch, const float* X,
                             const float* wz, const float* u, const float* v, float* dX, int strict,
                             int device);
/* Device-pointer form of the diffusion sweep for the HBM-roofline measurement: launches
 * `sweeps` back-to-back sweeps on `stream` (a hipStream_t, may be NULL) and returns without
 * synchronising.  Algorithmic traffic = 12 B/point/member-sweep. */
int greb_diffusion_batched_dev(const greb_params* p, int nx, int ny, int batch, const float* T1_dev,
                               const float* wz_dev, float* dX_dev, int strict, int sweeps,
                               void* stream);
/* greb_diffusion_batched_dev keeps small immutable tables per device between calls (row constants, the launch order of
 * the 384-wide sweep) so that back-to-back sweeps are not separated by an allocation.  A table is never rewritten once
 * made -- calls on different streams or threads, with different kappa or batch, do not disturb each other -- and a
 * long-lived host releases them with this (it waits for the device first).  Engines are unaffected. */
int greb_release_caches(void);
/* Diagnostic, host only (no GPU call): the launch order of the 384-wide diffusion sweep for a batch of fields, as
 * greb_diffusion_batched_dev would use it -- task i updates rows [k0[i], k1[i]) of field[i] (field < 0: an empty
 * slot), walking south to north where up[i] != 0.  Returns the number of tasks (only the first `capacity` are
 * written), 0 when the grid does not take that kernel, < 0 on a bad argument.  Any order is the same arithmetic;
 * tests check that every row of every field is written exactly once. */
int greb_diffusion_launch_order(const greb_params* p, int nx, int ny, int batch, int* field, int* k0, int* k1, int* up,
                                int capacity);
/* The same for the engine's row-strip circulation in its one-launch-per-sub-step form (384- and 192-wide grids): field = 2 * member + tracer; kappa:
 * [n_members] diffusivities (own sub-cycle tables) or NULL for p->kappa everywhere.  The order is the one an MI355X
 * (256 compute units) gets: at most one task per wavefront slot (2 048), tasks i and i + 1 024 share a SIMD. */
int greb_substep_launch_order(const greb_params* p, int nx, int ny, int n_members, const float* kappa, int* field, int* k0,
                              int* k1, int capacity);

/* The tasks of the engine's ONE-LAUNCH circulation call (greb_circ_rows.hip; src/greb.f90:546-550: the 24 sub-steps of a
 * `circulation` call inside one kernel) for `slots` wavefront slots: per task the field (2 * member + tracer), its rows
 * [k0, k1), whether it is a chain task (one row whose zonal chains stay in registers for the whole call) and up to four
 * dependencies dep4[4 * i .. 4 * i + 3] (task indices, -1 = none): the owners of rows k0-2, k0-1, k1, k1+1 of the same
 * field, whose completed sub-step s - 1 a task waits for before it starts sub-step s.  Host-only (no GPU call).
 * Returns the number of tasks -- never more than `slots` -- or 0 when the grid does not take this kernel or the slots
 * do not suffice (the engine then launches once per sub-step), or < 0. */
int greb_circulation_launch_plan(const greb_params* p, int nx, int ny, int n_members, const float* kappa, int slots,
                                 int* field, int* k0, int* k1, int* chain, int* dep4, int capacity);

/* Point physics of one step for a batch of columns sets (tests): SWradiation :367-403,
 * LWradiation :407-434, hydro :438-469, deep_ocean :495-525, seaice :472-492 evaluated by the
 * same device functions the engine uses.  in  : Ts,Ta,To,q,cap_surf [5][ny][nx]
 *                                          out : 15 fields [15][ny][nx] in the order
 * albedo, sw, LW_surf, LWair_down, em, Q_sens, Q_lat, Q_lat_air, dq_eva, dq_rain, dT_ocean, dTo,
 * cap_surf_new(seaice(Ts)), 0, 0.   ityr is 1-based. */
int greb_engine_point_physics(greb_engine* e, int ityr, float co2, const float* in5, float* out15);

/* ---- ensemble statistics across the members resident on one GPU -------------------
 * The reference leaves ensemble statistics to its R scripts over per-`ens_id` files (src/greb.f90:153,
 * 1064-1068).  x_dev: device array [n_members][n] (e.g. the monthly means viewed per member); all output
 * pointers are device pointers of n elements (any of them may be NULL); nothing synchronises.
 *   moments  : fp64 sum and sum of squares, min, max over the members -- the partials a multi-GPU run
 *              all-reduces (greb_climate_model_amd/ensemble.py: mean, variance, range)
 *   quantiles: out_dev [n_probs][n], probabilities in [0,1] (host array, <= 16), linear interpolation of the
 *              order statistics (numpy's default); n_members <= 4096 */
int greb_ensemble_moments_dev(const float* x_dev, int n_members, size_t n, double* sum_dev, double* sumsq_dev,
                              float* min_dev, float* max_dev, void* stream);
int greb_ensemble_quantiles_dev(const float* x_dev, int n_members, size_t n, const float* probs, int n_probs,
                                float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GREB_ENGINE_H */
