/* greb_oracle.h -- CPU restatement of the GREB hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library
 * (as the checker / the timed CPU baseline); the product path (libgreb_hip.so) never links,
 * loads or calls it.  Every function cites the reference lines (src/greb.f90) it restates.
 *
 * Parity pin: checked bit-for-bit against the reference compiled here with amdflang -O2
 * (oracle/_ref/libgreb_ref.so, per routine, and oracle/_ref/greb_ref, whole run) by
 * tests/golden/make_golden.py; the resulting vectors are committed under tests/golden/.
 */
#ifndef GREB_ORACLE_H
#define GREB_ORACLE_H
#include "../include/greb_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct greb_oracle greb_oracle;

/* grid tables of diffusion/advection (src/greb.f90:578-582,749-753 and the sub-cycle
 * bookkeeping :652-654,:838-840); arrays of length ny. */
typedef struct oracle_grid {
  float *dxlat, *dif_ccx, *adv_ccx, *dif_ccx2, *adv_ccx2;
  int *dif_time2, *adv_time2, *subcycled; /* subcycled[k] = !(dxlat(k) > 2.5e5) */
  float dif_ccy, adv_ccy;
} oracle_grid;

greb_oracle* oracle_create(const greb_params* p, int nx, int ny, const greb_fields* f);
void oracle_destroy(greb_oracle* o);
const oracle_grid* oracle_get_grid(const greb_oracle* o);

/* a1-a3.  wz/T1/dX: [ny][nx].  ityr 1-based. */
void oracle_diffusion(const greb_oracle* o, const float* T1, float* dX, const float* wz);
void oracle_advection(const greb_oracle* o, int ityr, const float* T1, float* dX, const float* wz);
void oracle_circulation(const greb_oracle* o, int ityr, const float* X_in, float* dX, const float* wz);
/* advection/circulation with an explicit wind slice instead of the climatology at ityr */
void oracle_advection_uv(const greb_oracle* o, const float* u, const float* v, const float* T1,
                         float* dX, const float* wz);
void oracle_circulation_uv(const greb_oracle* o, const float* u, const float* v, const float* X_in,
                           float* dX, const float* wz);

/* a4-a8 */
void oracle_swradiation(const greb_oracle* o, int ityr, const float* Ts, float* sw, float* albedo);
void oracle_lwradiation(const greb_oracle* o, int ityr, const float* Ts, const float* Ta,
                        const float* q, float co2, float* LWsurf, float* LWair_up,
                        float* LWair_down, float* em);
void oracle_hydro(const greb_oracle* o, int ityr, const float* Ts, const float* q, float* Qlat,
                  float* Qlat_air, float* dq_eva, float* dq_rain);
void oracle_seaice(greb_oracle* o, int ityr, const float* Ts); /* mutates cap_surf */
void oracle_deep_ocean(const greb_oracle* o, int ityr, const float* Ts, const float* To,
                       float* dT_ocean, float* dTo);

/* state access: which = 0 Ts, 1 Ta, 2 To, 3 q, 4 cap_surf, 5 wz_air, 6 wz_vapor, 7 z_ocean,
 * 8 Toclim(2-D) ; 10,11,12 = TF/qF/ToF_correct (730 slices) */
float* oracle_field(greb_oracle* o, int which);

/* a11: qflux_correction, src/greb.f90:311-364.  yearly: [years][2] or NULL */
void oracle_flux_correction(greb_oracle* o, int years, float* yearly);
/* a10 + a12-a14: scenario loop src/greb.f90:226-234.  co2_ppm [years];
 * monthly [years][12][5][ny][nx]; yearly [years][2] or NULL */
void oracle_run(greb_oracle* o, int years, const float* co2_ppm, float* monthly, float* yearly);

/* ---- log_exp sensitivity experiments of the upstream model variant,
 * /root/reference/src/greb.original.model.f90:60,162-166,394,423-430,452-453,492-495,513-515,553-571,939-951.
 * Pinned against that program compiled in place (oracle/_ref/greb_orig) by tests/golden/make_golden.py.
 * log_exp 10 (default) is the complete model == src/greb.f90.  Where the original reads an unassigned
 * intent(out) circulation increment (log_exp <= 4; vapour at 7 and 16) the oracle defines it as 0. */
void oracle_set_log_exp(greb_oracle* o, int log_exp);                   /* once, right after oracle_create */
void oracle_begin_run(greb_oracle* o, const float* state4, float year_start, int is_scenario); /* control / scenario loop start */
float oracle_co2_level(int log_exp, float year);
void oracle_set_co2_flux(greb_oracle* o, float co2); /* CO2_ctrl, :178-179 */

#ifdef __cplusplus
}
#endif
#endif
