/* greb_oracle.c -- CPU restatement of the GREB hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Scalar fp32, written from the reference's equations in the reference's evaluation order
 * (Fortran: a*b*c and a+b+c associate left to right, a*(..)/c == (a*(..))/c; integer literals
 * mixed with reals are converted; x**2 == x*x, x**4 == ((x*x)*x)*x as flang -O2 lowers it).  Build with
 * -ffp-contract=off (oracle/Makefile) so no FMA is formed: the reference's -O2 x86-64 build has
 * none either.  libm expf/logf/cosf/sqrtf are the same glibc routines the
 * flang-built reference calls, so on one machine the two agree bit for bit.
 *
 * Parity pin: tests/golden/make_golden.py compares every routine and whole runs against
 * oracle/_ref (the reference compiled here with amdflang -O2); results in tests/golden/MANIFEST.json.
 * All "src/greb.f90:N" citations are into /root/reference/.
 */
#include "greb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NT GREB_NSTEP_YR

struct greb_oracle {
  greb_params p;
  int nx, ny, np;
  /* inputs (copies) */
  float *z_topo, *glacier, *sw_solar;
  float *tclim, *qclim, *uclim, *vclim, *mldclim, *cldclim, *swetclim;
  /* derived, src/greb.f90:176-216,1088-1094 */
  float *toclim2d, *z_ocean, *wz_air, *wz_vapor, *cap_surf;
  float cap_ocean, cap_land, cap_air;
  float *TF, *qF, *ToF; /* [730][np] */
  /* state */
  float *Ts, *Ta, *To, *q;
  /* diagnostics/output accumulators, src/greb.f90:145-149 */
  float *tsmn, *Tmm, *Tamm, *Tomm, *qmm, *apmm;
  int mon;      /* 1..12, src/greb.f90:227,984 */
  long it_scnr; /* steps done in the scenario so far */
  float year;   /* src/greb.f90:227,233 (implicitly typed REAL) */
  oracle_grid g;
  int log_exp; /* greb.original.model.f90:60; 10 = complete model = src/greb.f90 */
  int scenario; /* 1 inside the original's scenario loop (SST+1 applies there only, :224-226) */
  /* scratch */
  float *scr;
};

static float* fdup(const float* s, size_t n) {
  float* d = (float*)malloc(n * sizeof(float));
  memcpy(d, s, n * sizeof(float));
  return d;
}
static float* fzero(size_t n) { return (float*)calloc(n, sizeof(float)); }

static int nint_f(float x) { return (int)lroundf(x); } /* Fortran NINT: half away from zero */

/* src/greb.f90:578-582 (diffusion), :749-753 (advection), :652-654, :838-840 (sub-cycling) */
static void grid_tables(greb_oracle* o) {
  const greb_params* p = &o->p;
  const int ny = o->ny;
  oracle_grid* g = &o->g;
  g->dxlat = fzero(ny); g->dif_ccx = fzero(ny); g->adv_ccx = fzero(ny);
  g->dif_ccx2 = fzero(ny); g->adv_ccx2 = fzero(ny);
  g->dif_time2 = (int*)calloc(ny, sizeof(int)); g->adv_time2 = (int*)calloc(ny, sizeof(int));
  g->subcycled = (int*)calloc(ny, sizeof(int));
  const float dlon = 360.f / (float)o->nx, dlat = 180.f / (float)o->ny; /* :43-44 */
  const float deg = 2.f * p->pi * 6.371e6f / 360.f;                    /* :578 */
  const float dx = dlon, dy = dlat, dyy = dy * deg;                    /* :579 */
  const float dtc = (float)p->dt_crcl;
  g->dif_ccy = p->kappa * dtc / (dyy * dyy);                           /* :581 */
  g->adv_ccy = dtc / dyy / 2.f;                                        /* :752 */
  for (int k = 0; k < ny; ++k) {
    const float lat = dlat * (float)(k + 1) - dlat / 2.f - 90.f;      /* :580 */
    const float dxlat = dx * deg * cosf(2.f * p->pi / 360.f * lat);   /* :580 */
    g->dxlat[k] = dxlat;
    g->dif_ccx[k] = p->kappa * dtc / (dxlat * dxlat);                  /* :582 */
    g->adv_ccx[k] = dtc / dxlat / 2.f;                                 /* :753 */
    g->subcycled[k] = !(dxlat > 2.5e5f);                               /* :592,:799 */
    { /* :652-654 */
      float dd = (float)(nint_f(dtc / (1.f * (dxlat * dxlat) / p->kappa)));
      if (dd < 1.f) dd = 1.f;
      const int dtdff2 = (int)(dtc / dd); /* integer assignment truncates */
      /* dtdff2 == 0 (384x192 polar rows): float(1800)/0. = +Inf, NINT(Inf) is undefined; the
       * flang x86-64 build yields INT_MIN -> max(1,.) = 1.  Defined here. */
      int t2 = dtdff2 == 0 ? 1 : nint_f(dtc / (float)dtdff2);
      if (t2 < 1) t2 = 1;
      g->dif_time2[k] = t2;
      g->dif_ccx2[k] = p->kappa * (float)dtdff2 / (dxlat * dxlat);
    }
    { /* :838-840 */
      float dd = (float)(nint_f(dtc / (dxlat / 10.0f / 1.f)));
      if (dd < 1.f) dd = 1.f;
      const int dtdff2 = (int)(dtc / dd);
      int t2 = dtdff2 == 0 ? 1 : nint_f(dtc / (float)dtdff2);
      if (t2 < 1) t2 = 1;
      g->adv_time2[k] = t2;
      g->adv_ccx2[k] = (float)dtdff2 / dxlat / 2.f;
    }
  }
}

greb_oracle* oracle_create(const greb_params* p, int nx, int ny, const greb_fields* f) {
  greb_oracle* o = (greb_oracle*)calloc(1, sizeof(*o));
  o->p = *p; o->nx = nx; o->ny = ny; o->np = nx * ny;
  const size_t np = (size_t)o->np, n3 = np * NT;
  o->z_topo = fdup(f->z_topo, np); o->glacier = fdup(f->glacier, np);
  o->sw_solar = fdup(f->sw_solar, (size_t)NT * ny);
  o->tclim = fdup(f->tclim, n3); o->qclim = fdup(f->qclim, n3); o->uclim = fdup(f->uclim, n3);
  o->vclim = fdup(f->vclim, n3); o->mldclim = fdup(f->mldclim, n3);
  o->cldclim = fdup(f->cldclim, n3); o->swetclim = fdup(f->swetclim, n3);
  o->toclim2d = fzero(np); o->z_ocean = fzero(np); o->wz_air = fzero(np); o->wz_vapor = fzero(np);
  o->cap_surf = fzero(np);
  o->TF = fzero(n3); o->qF = fzero(n3); o->ToF = fzero(n3);
  o->Ts = fzero(np); o->Ta = fzero(np); o->To = fzero(np); o->q = fzero(np);
  o->tsmn = fzero(np); o->Tmm = fzero(np); o->Tamm = fzero(np); o->Tomm = fzero(np);
  o->qmm = fzero(np); o->apmm = fzero(np);
  o->scr = fzero(np * 32);
  grid_tables(o);

  /* Toclim: src/greb.f90:1088-1094 */
  for (size_t i = 0; i < np; ++i) {
    float m = o->tclim[i];
    for (int t = 1; t < NT; ++t) { float v = o->tclim[(size_t)t * np + i]; if (v < m) m = v; }
    if (m - 273.15f < -1.7f) m = -1.7f + 273.15f;
    o->toclim2d[i] = m;
  }
  /* z_ocean: src/greb.f90:179-183 */
  for (size_t i = 0; i < np; ++i) {
    float m = 0.f;
    for (int t = 0; t < NT; ++t) { float v = o->mldclim[(size_t)t * np + i]; if (v > m) m = v; }
    o->z_ocean[i] = 3.0f * m;
  }
  /* heat capacities: src/greb.f90:186-191 */
  o->cap_ocean = p->cp_ocean * p->rho_ocean;
  o->cap_land = p->cp_land * p->rho_land * p->d_land;
  o->cap_air = p->cp_air * p->rho_air * p->d_air;
  for (size_t i = 0; i < np; ++i) {
    if (o->z_topo[i] > 0.f) o->cap_surf[i] = o->cap_land;
    if (o->z_topo[i] <= 0.f) o->cap_surf[i] = o->cap_ocean * o->mldclim[i];
    /* src/greb.f90:201-202 */
    o->wz_air[i] = expf(-o->z_topo[i] / p->z_air);
    o->wz_vapor[i] = expf(-o->z_topo[i] / p->z_vapor);
  }
  /* initial state: src/greb.f90:194-197 */
  const size_t last = (size_t)(NT - 1) * np;
  for (size_t i = 0; i < np; ++i) {
    o->Ts[i] = o->tclim[last + i]; o->Ta[i] = o->Ts[i];
    o->To[i] = o->toclim2d[i];     o->q[i] = o->qclim[last + i];
  }
  o->mon = 1; o->it_scnr = 0; o->year = (float)p->year0; /* :227 */
  o->log_exp = 10; o->scenario = 1;
  return o;
}

void oracle_destroy(greb_oracle* o) {
  if (!o) return;
  float* fs[] = {o->z_topo, o->glacier, o->sw_solar, o->tclim, o->qclim, o->uclim, o->vclim,
                 o->mldclim, o->cldclim, o->swetclim, o->toclim2d, o->z_ocean, o->wz_air,
                 o->wz_vapor, o->cap_surf, o->TF, o->qF, o->ToF, o->Ts, o->Ta, o->To, o->q,
                 o->tsmn, o->Tmm, o->Tamm, o->Tomm, o->qmm, o->apmm, o->scr, o->g.dxlat,
                 o->g.dif_ccx, o->g.adv_ccx, o->g.dif_ccx2, o->g.adv_ccx2};
  for (size_t i = 0; i < sizeof(fs) / sizeof(fs[0]); ++i) free(fs[i]);
  free(o->g.dif_time2); free(o->g.adv_time2); free(o->g.subcycled);
  free(o);
}

const oracle_grid* oracle_get_grid(const greb_oracle* o) { return &o->g; }

float* oracle_field(greb_oracle* o, int which) {
  switch (which) {
    case 0: return o->Ts; case 1: return o->Ta; case 2: return o->To; case 3: return o->q;
    case 4: return o->cap_surf; case 5: return o->wz_air; case 6: return o->wz_vapor;
    case 7: return o->z_ocean; case 8: return o->toclim2d;
    case 10: return o->TF; case 11: return o->qF; case 12: return o->ToF;
  }
  return NULL;
}

/* ------------------------------------------------------------------------------------------
 * a1  diffusion, src/greb.f90:556-723
 * ---------------------------------------------------------------------------------------- */

/* The 7-point longitudinal stencil sum of :595-600 (all seven hand-unrolled blocks are this
 * expression with wrapped indices).  T, w: one latitude row padded with a periodic halo of 3
 * (T[-3..nx+2] valid), so the wrap needs no modulo and the loop vectorises (no reassociation:
 * -ffp-contract=off, no -ffast-math). */
static inline float dif_S(const float* T, const float* w, int j) {
  return 10.f * (w[j - 1] * (T[j - 1] - T[j]) + w[j + 1] * (T[j + 1] - T[j]))
         + 4.f * (w[j - 2] * (T[j - 2] - T[j - 1]) + w[j - 1] * (T[j] - T[j - 1]))
         + 4.f * (w[j + 1] * (T[j] - T[j + 1]) + w[j + 2] * (T[j + 2] - T[j + 1]))
         + 1.f * (w[j - 3] * (T[j - 3] - T[j - 2]) + w[j - 2] * (T[j - 1] - T[j - 2]))
         + 1.f * (w[j + 2] * (T[j + 1] - T[j + 2]) + w[j + 3] * (T[j + 3] - T[j + 2]));
}

/* copy a row into a buffer with periodic halo of 3 on both sides; returns pointer to element 0 */
static inline float* pad_row(float* buf, const float* row, int nx) {
  float* p = buf + 3;
  memcpy(p, row, sizeof(float) * nx);
  p[-1] = row[nx - 1]; p[-2] = row[nx - 2]; p[-3] = row[nx - 3];
  p[nx] = row[0]; p[nx + 1] = row[1]; p[nx + 2] = row[2];
  return p;
}
static inline void rehalo(float* p, int nx) {
  p[-1] = p[nx - 1]; p[-2] = p[nx - 2]; p[-3] = p[nx - 3];
  p[nx] = p[0]; p[nx + 1] = p[1]; p[nx + 2] = p[2];
}

void oracle_diffusion(const greb_oracle* o, const float* T1, float* dX, const float* wz) {
  const int nx = o->nx, ny = o->ny;
  const oracle_grid* g = &o->g;
  const float ccy = g->dif_ccy;
  float* buf = (float*)malloc(sizeof(float) * 3 * (nx + 6));
  float* dTxh = buf + 2 * (nx + 6);
  for (int k = 0; k < ny; ++k) {
    const float* Tk = T1 + (size_t)k * nx;
    const float* wk = wz + (size_t)k * nx;
    float* out = dX + (size_t)k * nx;
    float* T1h = pad_row(buf, Tk, nx);
    const float* wp = pad_row(buf + nx + 6, wk, nx);
    /* longitudinal part first into out[] (dTx), then combine with dTy */
    if (!g->subcycled[k]) { /* :592-650 */
      const float ccx = g->dif_ccx[k];
      for (int j = 0; j < nx; ++j) out[j] = ccx * dif_S(T1h, wp, j) / 20.f;
    } else { /* :651-719 */
      const float ccx2 = g->dif_ccx2[k];
      for (int tt2 = 0; tt2 < g->dif_time2[k]; ++tt2) {
        for (int j = 0; j < nx; ++j) dTxh[j] = ccx2 * dif_S(T1h, wp, j) / 20.f;
        for (int j = 0; j < nx; ++j) {
          if (dTxh[j] <= -T1h[j]) dTxh[j] = -0.9f * T1h[j]; /* :715 */
          T1h[j] = T1h[j] + dTxh[j];                       /* :716 */
        }
        rehalo(T1h, nx);
      }
      for (int j = 0; j < nx; ++j) out[j] = T1h[j] - Tk[j]; /* :718 */
    }
    /* latitudinal :585-590, result :721 */
    if (k >= 1 && k <= ny - 2) {
      const float* Tm = Tk - nx; const float* Tp = Tk + nx;
      for (int j = 0; j < nx; ++j) {
        const float dTy = ccy * (wk[j - nx] * (Tm[j] - Tk[j]) + wk[j + nx] * (Tp[j] - Tk[j]));
        out[j] = wk[j] * (out[j] + dTy);
      }
    } else if (k == 0) {
      for (int j = 0; j < nx; ++j) {
        const float dTy = ccy * wk[j + nx] * (-Tk[j] + Tk[j + nx]);
        out[j] = wk[j] * (out[j] + dTy);
      }
    } else {
      for (int j = 0; j < nx; ++j) {
        const float dTy = ccy * wk[j - nx] * (Tk[j - nx] - Tk[j]);
        out[j] = wk[j] * (out[j] + dTy);
      }
    }
  }
  free(buf);
}

/* ------------------------------------------------------------------------------------------
 * a2  advection, src/greb.f90:726-915.  um/up, vm/vp are uclim_m/uclim_p, vclim_m/vclim_p of
 * :203-216: "_m" holds the non-negative part, "_p" the negative part.
 * ---------------------------------------------------------------------------------------- */
static inline float split_m(float u) { return u >= 0.0f ? u : 0.0f; } /* :203-205 */
static inline float split_p(float u) { return u >= 0.0f ? 0.0f : u; } /* :206-208 */

void oracle_advection_uv(const greb_oracle* o, const float* u, const float* v, const float* T1,
                         float* dX, const float* wz) {
  const int nx = o->nx, ny = o->ny;
  const oracle_grid* g = &o->g;
  const float ccy = g->adv_ccy;
  float* buf = (float*)malloc(sizeof(float) * 3 * (nx + 6));
  float* dTxh = buf + 2 * (nx + 6);
  for (int k = 0; k < ny; ++k) {
    const float* Tk = T1 + (size_t)k * nx;
    const float* wk = wz + (size_t)k * nx;
    const float* uk = u + (size_t)k * nx;
    const float* vk = v + (size_t)k * nx;
    float* out = dX + (size_t)k * nx;
    float* T1h = pad_row(buf, Tk, nx);
    const float* wp = pad_row(buf + nx + 6, wk, nx);
    /* longitudinal :798-912 */
    if (!g->subcycled[k]) {
      const float ccx = g->adv_ccx[k];
      for (int j = 0; j < nx; ++j) { /* :802-835 */
        out[j] = ccx * (-split_m(uk[j]) * (wp[j - 1] * (T1h[j] - T1h[j - 1]) + wp[j - 2] * (T1h[j] - T1h[j - 2]))
                        + split_p(uk[j]) * (wp[j + 1] * (T1h[j] - T1h[j + 1]) + wp[j + 2] * (T1h[j] - T1h[j + 2])))
                 / 3.f;
      }
    } else {
      const float ccx2 = g->adv_ccx2[k];
      for (int tt2 = 0; tt2 < g->adv_time2[k]; ++tt2) {
        for (int j = 0; j < nx; ++j) { /* :845-906 */
          dTxh[j] = ccx2 * (-split_m(uk[j]) * (10.f * wp[j - 1] * (T1h[j] - T1h[j - 1])
                                               + 4.f * wp[j - 2] * (T1h[j - 1] - T1h[j - 2])
                                               + 1.f * wp[j - 3] * (T1h[j - 2] - T1h[j - 3]))
                            + split_p(uk[j]) * (10.f * wp[j + 1] * (T1h[j] - T1h[j + 1])
                                                + 4.f * wp[j + 2] * (T1h[j + 1] - T1h[j + 2])
                                                + 1.f * wp[j + 3] * (T1h[j + 2] - T1h[j + 3])))
                    / 20.f;
        }
        { /* :881  j = xdim-2 (1-based): jp1 = xdim-1, jp2 = xdim-1 (should be xdim), jp3 = 1 --
           * reference index bug, reproduced.  0-based: j = nx-3, jp1 = jp2 = nx-2, jp3 = 0. */
          const int j = nx - 3, jp1 = nx - 2, jp2 = nx - 2, jp3 = 0;
          dTxh[j] = ccx2 * (-split_m(uk[j]) * (10.f * wp[j - 1] * (T1h[j] - T1h[j - 1])
                                               + 4.f * wp[j - 2] * (T1h[j - 1] - T1h[j - 2])
                                               + 1.f * wp[j - 3] * (T1h[j - 2] - T1h[j - 3]))
                            + split_p(uk[j]) * (10.f * wp[jp1] * (T1h[j] - T1h[jp1])
                                                + 4.f * wp[jp2] * (T1h[jp1] - T1h[jp2])
                                                + 1.f * wp[jp3] * (T1h[jp2] - T1h[jp3])))
                    / 20.f;
        }
        for (int j = 0; j < nx; ++j) {
          if (dTxh[j] <= -T1h[j]) dTxh[j] = -0.9f * T1h[j]; /* :907 */
          T1h[j] = T1h[j] + dTxh[j];
        }
        rehalo(T1h, nx);
      }
      for (int j = 0; j < nx; ++j) out[j] = T1h[j] - Tk[j]; /* :910 */
    }
    /* latitudinal :756-795 */
#define DLAT(n) (wk[j + (n) * nx] * (T0 - Tk[j + (n) * nx]))
#define ADV_LAT_LOOP(EXPR)                                                          \
    for (int j = 0; j < nx; ++j) {                                                  \
      const float T0 = Tk[j], vm = split_m(vk[j]), vp = split_p(vk[j]);             \
      (void)vm; (void)vp;                                                           \
      out[j] = out[j] + (EXPR); /* :913 */                                          \
    }
    if (k == 0) {
      ADV_LAT_LOOP(ccy * (vp * (DLAT(1) + DLAT(2))) / 3.f)                                /* :759-761 */
    } else if (k == 1) {
      ADV_LAT_LOOP(ccy * (-vm * (DLAT(-1)) + vp * (DLAT(1) + DLAT(2)) / 3.f))             /* :766-769 */
    } else if (k <= ny - 3) {
      ADV_LAT_LOOP(ccy * (-vm * (DLAT(-1) + DLAT(-2)) + vp * (DLAT(1) + DLAT(2))) / 3.f)  /* :774-778 */
    } else if (k == ny - 2) {
      ADV_LAT_LOOP(ccy * (-vm * (DLAT(-1) + DLAT(-2)) / 3.f + vp * (DLAT(1))))            /* :784-787 */
    } else {
      ADV_LAT_LOOP(ccy * (-vm * (DLAT(-1) + DLAT(-2))) / 3.f)                             /* :792-794 */
    }
#undef ADV_LAT_LOOP
#undef DLAT
  }
  free(buf);
}

void oracle_advection(const greb_oracle* o, int ityr, const float* T1, float* dX, const float* wz) {
  const size_t off = (size_t)(ityr - 1) * o->np;
  oracle_advection_uv(o, o->uclim + off, o->vclim + off, T1, dX, wz);
}

/* a3  circulation, src/greb.f90:528-553 */
void oracle_circulation_uv(const greb_oracle* o, const float* u, const float* v, const float* X_in,
                           float* dX, const float* wz) {
  const int np = o->np;
  int time = nint_f((float)o->p.dt / (float)o->p.dt_crcl); /* :543 */
  if (time < 1) time = 1;
  float* X = (float*)malloc(sizeof(float) * 3 * np);
  float *dd = X + np, *da = X + 2 * np;
  memcpy(X, X_in, sizeof(float) * np);
  for (int tt = 0; tt < time; ++tt) {
    oracle_diffusion(o, X, dd, wz);
    oracle_advection_uv(o, u, v, X, da, wz);
    for (int i = 0; i < np; ++i) X[i] = X[i] + dd[i] + da[i]; /* :549 */
  }
  for (int i = 0; i < np; ++i) dX[i] = X[i] - X_in[i]; /* :551 */
  free(X);
}

/* circulation() of the original with log_exp == 8 for the vapour call: diffusion sub-steps only
 * (greb.original.model.f90:560-564) */
static void oracle_diffusion_only(const greb_oracle* o, const float* X_in, float* dX, const float* wz) {
  const int np = o->np;
  int time = nint_f((float)o->p.dt / (float)o->p.dt_crcl);
  if (time < 1) time = 1;
  float* X = (float*)malloc(sizeof(float) * 2 * np);
  float* dd = X + np;
  memcpy(X, X_in, sizeof(float) * np);
  for (int tt = 0; tt < time; ++tt) {
    oracle_diffusion(o, X, dd, wz);
    for (int i = 0; i < np; ++i) X[i] = X[i] + dd[i];
  }
  for (int i = 0; i < np; ++i) dX[i] = X[i] - X_in[i];
  free(X);
}

void oracle_circulation(const greb_oracle* o, int ityr, const float* X_in, float* dX, const float* wz) {
  const size_t off = (size_t)(ityr - 1) * o->np;
  oracle_circulation_uv(o, o->uclim + off, o->vclim + off, X_in, dX, wz);
}

/* ------------------------------------------------------------------------------------------
 * a4  SWradiation, src/greb.f90:367-403
 * ---------------------------------------------------------------------------------------- */
void oracle_swradiation(const greb_oracle* o, int ityr, const float* Ts, float* sw, float* albedo) {
  const greb_params* p = &o->p;
  const int nx = o->nx, ny = o->ny;
  const float* cld = o->cldclim + (size_t)(ityr - 1) * o->np;
  const float* sol = o->sw_solar + (size_t)(ityr - 1) * ny;
  for (int k = 0; k < ny; ++k)
    for (int j = 0; j < nx; ++j) {
      const int i = k * nx + j;
      const float a_atmos = cld[i] * p->a_cloud; /* :380 */
      const float T = Ts[i];
      float a_surf = 0.f;
      if (o->z_topo[i] >= 0.f) { /* :384-387 */
        if (T <= p->Tl_ice1) a_surf = p->a_no_ice + p->da_ice;
        if (T >= p->Tl_ice2) a_surf = p->a_no_ice;
        if (T > p->Tl_ice1 && T < p->Tl_ice2)
          a_surf = p->a_no_ice + p->da_ice * (1.f - (T - p->Tl_ice1) / (p->Tl_ice2 - p->Tl_ice1));
      } else { /* :389-392 */
        if (T <= p->To_ice1) a_surf = p->a_no_ice + p->da_ice;
        if (T >= p->To_ice2) a_surf = p->a_no_ice;
        if (T > p->To_ice1 && T < p->To_ice2)
          a_surf = p->a_no_ice + p->da_ice * (1.f - (T - p->To_ice1) / (p->To_ice2 - p->To_ice1));
      }
      if (o->glacier[i] > 0.5f) a_surf = p->a_no_ice + p->da_ice; /* :395 */
      if (o->log_exp <= 5) a_surf = p->a_no_ice;                  /* greb.original.model.f90:394 */
      albedo[i] = a_surf + a_atmos - a_surf * a_atmos;            /* :398 */
      sw[i] = sol[k] * (1.f - albedo[i]);                         /* :400 */
    }
}

/* a5  LWradiation, src/greb.f90:407-434 */
static inline float pow4(float x) { return x * x * x * x; } /* flang -O2 lowers x**4 as ((x*x)*x)*x */

void oracle_lwradiation(const greb_oracle* o, int ityr, const float* Ts, const float* Ta,
                        const float* q, float co2, float* LWsurf, float* LWair_up,
                        float* LWair_down, float* em) {
  const greb_params* p = &o->p;
  const float* pe = p->p_emi; /* pe[0] == p_emi(1) */
  const size_t off = (size_t)(ityr - 1) * o->np;
  for (int i = 0; i < o->np; ++i) {
    const float ez = expf(-o->z_topo[i] / p->z_air);
    const float e_co2 = ez * co2;              /* :420 */
    float e_vapor = ez * p->r_qviwv * q[i]; /* :421 */
    if (o->log_exp == 11) e_vapor = ez * p->r_qviwv * o->qclim[off + i]; /* greb.original.model.f90:423 */
    const float e_cloud = o->cldclim[off + i]; /* :422 */
    float e = pe[3] * logf(pe[0] * e_co2 + pe[1] * e_vapor + pe[2]) + pe[6]
              + pe[4] * logf(pe[0] * e_co2 + pe[2])
              + pe[5] * logf(pe[1] * e_vapor + pe[2]); /* :425-427 */
    e = (pe[7] - e_cloud) / pe[8] * (e - pe[9]) + pe[9]; /* :428 */
    if (o->log_exp == 11) e = e + 0.022f / (0.15f * 24.f) * p->r_qviwv * (q[i] - o->qclim[off + i]); /* orig :430 */
    em[i] = e;
    LWsurf[i] = -p->sig * pow4(Ts[i]);                                   /* :430 */
    const float dTrad = -0.16f * o->tclim[off + i] - 5.f;                /* :176 */
    LWair_down[i] = -e * p->sig * pow4(Ta[i] + dTrad);                   /* :431 */
    LWair_up[i] = LWair_down[i];                                         /* :432 */
  }
}

/* a6  hydro, src/greb.f90:438-469 */
void oracle_hydro(const greb_oracle* o, int ityr, const float* Ts, const float* q, float* Qlat,
                  float* Qlat_air, float* dq_eva, float* dq_rain) {
  const greb_params* p = &o->p;
  const size_t off = (size_t)(ityr - 1) * o->np;
  if (o->log_exp <= 6 || o->log_exp == 13 || o->log_exp == 15) { /* greb.original.model.f90:452-453 */
    for (int i = 0; i < o->np; ++i) Qlat[i] = Qlat_air[i] = dq_eva[i] = dq_rain[i] = 0.f;
    return;
  }
  for (int i = 0; i < o->np; ++i) {
    const float u = o->uclim[off + i], v = o->vclim[off + i];
    float abswind = sqrtf(u * u + v * v);                                  /* :452 */
    if (o->z_topo[i] > 0.f) abswind = sqrtf(abswind * abswind + 2.0f * 2.0f); /* :453 */
    if (o->z_topo[i] < 0.f) abswind = sqrtf(abswind * abswind + 3.0f * 3.0f); /* :454 */
    float qs = 3.75e-3f * expf(17.08085f * (Ts[i] - 273.15f) / (Ts[i] - 273.15f + 234.175f)); /* :457 */
    qs = qs * expf(-o->z_topo[i] / p->z_air);                              /* :458 */
    Qlat[i] = (q[i] - qs) * abswind * p->cq_latent * p->rho_air * p->ce * o->swetclim[off + i]; /* :460 */
    dq_eva[i] = -Qlat[i] / p->cq_latent / p->r_qviwv;                      /* :463 */
    dq_rain[i] = p->cq_rain * q[i];                                        /* :464 */
    Qlat_air[i] = -dq_rain[i] * p->cq_latent * p->r_qviwv;                 /* :467 */
  }
}

/* a8  seaice, src/greb.f90:472-492 */
void oracle_seaice(greb_oracle* o, int ityr, const float* Ts) {
  const greb_params* p = &o->p;
  const float* mld = o->mldclim + (size_t)(ityr - 1) * o->np;
  for (int i = 0; i < o->np; ++i) {
    const float T = Ts[i];
    if (o->z_topo[i] < 0.f) {
      if (T <= p->To_ice1) o->cap_surf[i] = o->cap_land;               /* :483 */
      if (T >= p->To_ice2) o->cap_surf[i] = o->cap_ocean * mld[i];     /* :484 */
      if (T > p->To_ice1 && T < p->To_ice2)                            /* :485-487 */
        o->cap_surf[i] = o->cap_land + (o->cap_ocean * mld[i] - o->cap_land)
                                           / (p->To_ice2 - p->To_ice1) * (T - p->To_ice1);
    }
    if (o->log_exp <= 5) { /* greb.original.model.f90:492-495 */
      if (o->z_topo[i] > 0.f) o->cap_surf[i] = o->cap_land;
      if (o->z_topo[i] < 0.f) o->cap_surf[i] = o->cap_ocean * mld[i];
    }
    if (o->glacier[i] > 0.5f) o->cap_surf[i] = o->cap_land;            /* :490 */
  }
}

/* a7  deep_ocean, src/greb.f90:495-525 */
void oracle_deep_ocean(const greb_oracle* o, int ityr, const float* Ts, const float* To,
                       float* dT_ocean, float* dTo) {
  const greb_params* p = &o->p;
  const float* mld = o->mldclim + (size_t)(ityr - 1) * o->np;
  const float* mldm = o->mldclim + (size_t)(ityr > 1 ? ityr - 2 : NT - 1) * o->np; /* :507-508 */
  const float dt = (float)p->dt;
  if (o->log_exp <= 9 || o->log_exp == 11 || (o->log_exp >= 14 && o->log_exp <= 16)) { /* orig :513-515 */
    for (int i = 0; i < o->np; ++i) dT_ocean[i] = dTo[i] = 0.f;
    return;
  }
  for (int i = 0; i < o->np; ++i) {
    float a = 0.f, b = 0.f; /* dTo, dT_ocean :505 */
    const float dmld = mld[i] - mldm[i];
    if (o->z_topo[i] < 0.f && Ts[i] >= p->To_ice2 && dmld < 0.f)
      a = -dmld / (o->z_ocean[i] - mld[i]) * (Ts[i] - To[i]);        /* :511-512 */
    if (o->z_topo[i] < 0.f && Ts[i] >= p->To_ice2 && dmld > 0.f)
      b = dmld / mld[i] * (To[i] - Ts[i]);                            /* :513-514 */
    a = 0.5f * a; b = 0.5f * b;                                       /* :516-518 */
    const float Tx = p->To_ice2 > Ts[i] ? p->To_ice2 : Ts[i];         /* :521 */
    a = a + dt * p->co_turb * (Tx - To[i]) / (o->cap_ocean * (o->z_ocean[i] - mld[i])); /* :522 */
    b = b + dt * p->co_turb * (To[i] - Tx) / (o->cap_ocean * mld[i]);                   /* :523 */
    dTo[i] = a; dT_ocean[i] = b;
  }
}

/* ------------------------------------------------------------------------------------------
 * a9  tendencies, src/greb.f90:277-308.  Work arrays are slices of o->scr.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  float *albedo, *sw, *LW_surf, *Q_lat, *Q_sens, *Q_lat_air, *dq_eva, *dq_rain, *dq_crcl, *dTa_crcl,
      *dT_ocean, *dTo, *LWair_down, *LWair_up, *em;
} tend_t;

static tend_t tend_slices(greb_oracle* o) {
  tend_t t; float* s = o->scr; const int np = o->np;
  t.albedo = s; t.sw = s + np; t.LW_surf = s + 2 * np; t.Q_lat = s + 3 * np; t.Q_sens = s + 4 * np;
  t.Q_lat_air = s + 5 * np; t.dq_eva = s + 6 * np; t.dq_rain = s + 7 * np; t.dq_crcl = s + 8 * np;
  t.dTa_crcl = s + 9 * np; t.dT_ocean = s + 10 * np; t.dTo = s + 11 * np; t.LWair_down = s + 12 * np;
  t.LWair_up = s + 13 * np; t.em = s + 14 * np;
  return t;
}

static void tendencies(greb_oracle* o, int ityr, float co2, const tend_t* t) {
  oracle_swradiation(o, ityr, o->Ts, t->sw, t->albedo);                                   /* :291 */
  oracle_lwradiation(o, ityr, o->Ts, o->Ta, o->q, co2, t->LW_surf, t->LWair_up, t->LWair_down, t->em); /* :293 */
  for (int i = 0; i < o->np; ++i) t->Q_sens[i] = o->p.ct_sens * (o->Ta[i] - o->Ts[i]);    /* :295 */
  oracle_hydro(o, ityr, o->Ts, o->q, t->Q_lat, t->Q_lat_air, t->dq_eva, t->dq_rain);     /* :297 */
  /* greb.original.model.f90:553-571.  Where the original returns before assigning dX_crcl (log_exp <= 4; the
   * vapour call at 7 and 16) its intent(out) result is undefined; defined here as no transport (0). */
  if (o->log_exp <= 4) memset(t->dTa_crcl, 0, sizeof(float) * o->np);
  else oracle_circulation(o, ityr, o->Ta, t->dTa_crcl, o->wz_air);                        /* :301 */
  if (o->log_exp <= 4 || o->log_exp == 7 || o->log_exp == 16) memset(t->dq_crcl, 0, sizeof(float) * o->np);
  else if (o->log_exp == 8) oracle_diffusion_only(o, o->q, t->dq_crcl, o->wz_vapor);     /* orig :560-564 */
  else oracle_circulation(o, ityr, o->q, t->dq_crcl, o->wz_vapor);                        /* :303 */
  oracle_deep_ocean(o, ityr, o->Ts, o->To, t->dT_ocean, t->dTo);                          /* :306 */
}

/* a13  diagnostics, src/greb.f90:929-959 (only tsmn is ever read) */
static void diagnostics(greb_oracle* o, int ityr, const float* Ts0, float* yearly2) {
  const int np = o->np;
  for (int i = 0; i < np; ++i) o->tsmn[i] = o->tsmn[i] + Ts0[i]; /* :945 */
  if (ityr == NT) {
    float s = 0.f;
    for (int i = 0; i < np; ++i) { o->tsmn[i] = o->tsmn[i] / (float)NT; s += o->tsmn[i]; } /* :949 */
    if (yearly2) {
      yearly2[0] = s / (float)(np)-273.15f; /* :954 sum(tsmn)/(xdim*ydim)-273.15 */
      yearly2[1] = o->tsmn[(o->p.ipy - 1) * o->nx + (o->p.ipx - 1)] - 273.15f;
    }
    memset(o->tsmn, 0, sizeof(float) * np); /* :955 */
  }
}

/* a11  qflux_correction, src/greb.f90:311-364 */
void oracle_flux_correction(greb_oracle* o, int years, float* yearly) {
  const greb_params* p = &o->p;
  const int np = o->np;
  const float dt = (float)p->dt;
  tend_t t = tend_slices(o);
  float* Ts0 = o->scr + 16 * np; float* Ta0 = o->scr + 17 * np;
  float* To0 = o->scr + 18 * np; float* q0 = o->scr + 19 * np;
  for (long it = 1; it <= (long)years * NT; ++it) {
    const int ityr = (int)((it - 1) % NT) + 1; /* :327 */
    const size_t off = (size_t)(ityr - 1) * np;
    tendencies(o, ityr, p->co2_flux, &t);
    for (int i = 0; i < np; ++i) {
      const float dTs = dt * (t.sw[i] + t.LW_surf[i] - t.LWair_down[i] + t.Q_lat[i] + t.Q_sens[i]) / o->cap_surf[i]; /* :333 */
      float ts0 = o->Ts[i] + dTs + t.dT_ocean[i];                                                      /* :334 */
      const float dTa = dt * (t.LWair_up[i] + t.LWair_down[i] - t.em[i] * t.LW_surf[i] + t.Q_lat_air[i] - t.Q_sens[i]) / o->cap_air; /* :336 */
      const float ta0 = o->Ta[i] + dTa + t.dTa_crcl[i];                                                 /* :337 */
      float to0 = o->To[i] + t.dTo[i];                                                                 /* :339 */
      const float dq = dt * (t.dq_eva[i] + t.dq_rain[i]);                                               /* :341 */
      float qq0 = o->q[i] + dq + t.dq_crcl[i];                                                         /* :342 */
      const float T_error = o->tclim[off + i] - ts0;                                                   /* :344 */
      o->TF[off + i] = T_error * o->cap_surf[i] / dt;                                                  /* :345 */
      ts0 = o->Ts[i] + dTs + t.dT_ocean[i] + o->TF[off + i] * dt / o->cap_surf[i];                      /* :347 */
      o->ToF[off + i] = o->toclim2d[i] - to0;                                                          /* :349 */
      to0 = o->To[i] + t.dTo[i] + o->ToF[off + i];                                                     /* :351 */
      o->qF[off + i] = o->qclim[off + i] - qq0;                                                        /* :353 */
      qq0 = o->q[i] + dq + t.dq_crcl[i] + o->qF[off + i];                                              /* :355 */
      Ts0[i] = ts0; Ta0[i] = ta0; To0[i] = to0; q0[i] = qq0;
    }
    oracle_seaice(o, ityr, Ts0);                                                                       /* :357 */
    diagnostics(o, ityr, Ts0, yearly ? yearly + 2 * ((it - 1) / NT) : NULL);                           /* :359 */
    memcpy(o->Ts, Ts0, sizeof(float) * np); memcpy(o->Ta, Ta0, sizeof(float) * np);                    /* :361 */
    memcpy(o->q, q0, sizeof(float) * np);   memcpy(o->To, To0, sizeof(float) * np);
  }
}

/* a12  output (accumulate half), src/greb.f90:962-987.  rec: where this month's 5 records go. */
static const int jday_mon[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31}; /* :42 */

static int output_step(greb_oracle* o, long it, int jday, const float* Ts0, const float* Ta0,
                       const float* To0, const float* q0, const float* albedo, float* rec) {
  const int np = o->np;
  for (int i = 0; i < np; ++i) { /* :974 */
    o->Tmm[i] += Ts0[i]; o->Tamm[i] += Ta0[i]; o->Tomm[i] += To0[i]; o->qmm[i] += q0[i];
    o->apmm[i] += albedo[i];
  }
  int cum = 0;
  for (int m = 0; m < o->mon; ++m) cum += jday_mon[m];
  if (jday == cum && (it % 2) == 0) { /* :975-976 (it/float(ndt_days) integral <=> it even) */
    const float ndm = (float)(jday_mon[o->mon - 1] * 2); /* :977 */
    for (int i = 0; i < np; ++i) {
      rec[i] = o->Tmm[i] / ndm; rec[np + i] = o->Tamm[i] / ndm; rec[2 * np + i] = o->Tomm[i] / ndm;
      rec[3 * np + i] = o->qmm[i] / ndm; rec[4 * np + i] = o->apmm[i] / ndm; /* :978-982 */
    }
    memset(o->Tmm, 0, sizeof(float) * np); memset(o->Tamm, 0, sizeof(float) * np);
    memset(o->Tomm, 0, sizeof(float) * np); memset(o->qmm, 0, sizeof(float) * np);
    memset(o->apmm, 0, sizeof(float) * np); /* :983 */
    o->mon += 1; if (o->mon == 13) o->mon = 1; /* :984 */
    return 1;
  }
  return 0;
}

/* scenario loop src/greb.f90:226-234, time_loop :239-274, co2_level :918-926 */
void oracle_run(greb_oracle* o, int years, const float* co2_ppm, float* monthly, float* yearly) {
  const greb_params* p = &o->p;
  const int np = o->np;
  const float dt = (float)p->dt;
  tend_t t = tend_slices(o);
  float* Ts0 = o->scr + 16 * np; float* Ta0 = o->scr + 17 * np;
  float* To0 = o->scr + 18 * np; float* q0 = o->scr + 19 * np;
  long nrec = 0;
  const float year_start = o->year;
  for (long n = 1; n <= (long)years * NT; ++n) {
    const long it = o->it_scnr + n;
    const float co2 = co2_ppm[(int)(o->year - year_start + 1.f) - 1]; /* :924 (index relative to this call) */
    const int jday = (int)(((it - 1) / 2) % 365) + 1; /* :251 */
    const int ityr = (int)((it - 1) % NT) + 1;        /* :252 */
    const size_t off = (size_t)(ityr - 1) * np;
    if (o->scenario && o->log_exp >= 14 && o->log_exp <= 16) {
      /* sens. exp. SST+1, greb.original.model.f90:226 (CO2 = CO2_ctrl: the caller's series).  The statement sits
       * BEFORE time_loop updates the module variable ityr (:247), so it reads the climatology slice of the
       * PREVIOUS step -- slice 730 at it = 1, left there by the preceding phase. */
      const size_t offp = (size_t)((it - 2 + NT) % NT) * np;
      for (int i = 0; i < np; ++i) if (o->z_topo[i] < 0.0f) o->Ts[i] = o->tclim[offp + i] + 1.0f;
    }
    tendencies(o, ityr, co2, &t);                      /* :254 */
    for (int i = 0; i < np; ++i) {
      Ts0[i] = o->Ts[i] + t.dT_ocean[i]
               + dt * (t.sw[i] + t.LW_surf[i] - t.LWair_down[i] + t.Q_lat[i] + t.Q_sens[i] + o->TF[off + i]) / o->cap_surf[i]; /* :258 */
      Ta0[i] = o->Ta[i] + t.dTa_crcl[i]
               + dt * (t.LWair_up[i] + t.LWair_down[i] - t.em[i] * t.LW_surf[i] + t.Q_lat_air[i] - t.Q_sens[i]) / o->cap_air; /* :260 */
      To0[i] = o->To[i] + t.dTo[i] + o->ToF[off + i];                                          /* :262 */
      float dq = dt * (t.dq_eva[i] + t.dq_rain[i]) + t.dq_crcl[i] + o->qF[off + i];             /* :264 */
      if (dq <= -o->q[i]) dq = -0.9f * o->q[i];                                                /* :265 */
      q0[i] = o->q[i] + dq;                                                                    /* :266 */
    }
    oracle_seaice(o, ityr, Ts0);                                                               /* :268 */
    nrec += output_step(o, it, jday, Ts0, Ta0, To0, q0, t.albedo, monthly + (size_t)nrec * 5 * np); /* :270 */
    diagnostics(o, ityr, Ts0, yearly ? yearly + 2 * ((n - 1) / NT) : NULL);                     /* :272 */
    memcpy(o->Ts, Ts0, sizeof(float) * np); memcpy(o->Ta, Ta0, sizeof(float) * np);            /* :232 */
    memcpy(o->q, q0, sizeof(float) * np);   memcpy(o->To, To0, sizeof(float) * np);
    if (it % NT == 0) o->year = o->year + 1.f;                                                 /* :233 */
  }
  o->it_scnr += (long)years * NT;
}


/* ------------------------------------------------------------------------------------------
 * log_exp sensitivity experiments of the upstream model variant.
 * All citations in this block are into /root/reference/src/greb.original.model.f90.
 * ---------------------------------------------------------------------------------------- */
/* greb_model preamble :162-197: boundary-data changes of the experiment, then the derived fields that are
 * computed after them (cap_surf, initial q, wz_*).  z_ocean (:155-160) and Toclim (shell) keep their values
 * from the unmodified data.  Call once, right after oracle_create. */
void oracle_set_log_exp(greb_oracle* o, int log_exp) {
  const size_t np = (size_t)o->np, n3 = np * NT;
  o->log_exp = log_exp;
  if (log_exp == 1) for (size_t i = 0; i < np; ++i) if (o->z_topo[i] > 1.f) o->z_topo[i] = 1.0f; /* :162 */
  if (log_exp <= 2) for (size_t i = 0; i < n3; ++i) o->cldclim[i] = 0.7f;                          /* :163 */
  if (log_exp <= 3) for (size_t i = 0; i < n3; ++i) o->qclim[i] = 0.0052f;                         /* :164 */
  if (log_exp <= 9 || log_exp == 11) for (size_t i = 0; i < n3; ++i) o->mldclim[i] = o->p.d_ocean; /* :165-166 */
  const size_t last = (size_t)(NT - 1) * np;
  for (size_t i = 0; i < np; ++i) {
    if (o->z_topo[i] > 0.f) o->cap_surf[i] = o->cap_land;                       /* :169 */
    if (o->z_topo[i] <= 0.f) o->cap_surf[i] = o->cap_ocean * o->mldclim[i];     /* :170 */
    o->q[i] = o->qclim[last + i];                                               /* :176 */
    o->wz_air[i] = expf(-o->z_topo[i] / o->p.z_air);                            /* :182 */
    o->wz_vapor[i] = expf(-o->z_topo[i] / o->p.z_vapor);                        /* :183 */
  }
}

/* Start of the control / scenario loop of the original (:210-211, :219-220): clock, month counter and
 * accumulators restart; the state is (re)set to state4 = Ts,Ta,To,q [4][np] if given.  cap_surf is NOT
 * reset -- it carries over from the previous phase, as in the original. */
void oracle_begin_run(greb_oracle* o, const float* state4, float year_start, int is_scenario) {
  const size_t np = (size_t)o->np;
  o->scenario = is_scenario;
  o->mon = 1; o->it_scnr = 0; o->year = year_start;
  memset(o->Tmm, 0, sizeof(float) * np); memset(o->Tamm, 0, sizeof(float) * np);
  memset(o->Tomm, 0, sizeof(float) * np); memset(o->qmm, 0, sizeof(float) * np);
  memset(o->apmm, 0, sizeof(float) * np);
  if (state4) {
    memcpy(o->Ts, state4, sizeof(float) * np); memcpy(o->Ta, state4 + np, sizeof(float) * np);
    memcpy(o->To, state4 + 2 * np, sizeof(float) * np); memcpy(o->q, state4 + 3 * np, sizeof(float) * np);
  }
}

/* co2_level :939-951 for the model year `year` */
float oracle_co2_level(int log_exp, float year) {
  float CO2 = 680.f;
  if (log_exp == 12 || log_exp == 13) {
    const float CO2_1950 = 310.f, CO2_2000 = 370.f, CO2_2050 = 520.f;
    if (year <= 2000.f) CO2 = CO2_1950 + 60.f / 50.f * (year - 1950.f);
    if (year > 2000.f && year <= 2050.f) CO2 = CO2_2000 + 150.f / 50.f * (year - 2000.f);
    if (year > 2050.f && year <= 2100.f) CO2 = CO2_2050 + 180.f / 50.f * (year - 2050.f);
  }
  return CO2;
}

/* CO2_ctrl of the original (:178-179) is what qflux_correction runs at; src/greb.f90 calls it co2_flux */
void oracle_set_co2_flux(greb_oracle* o, float co2) { o->p.co2_flux = co2; }
