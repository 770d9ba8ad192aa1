/* greb_run.c -- the C ABI of the MI355X GREB engine used from plain C: no Fortran, no Python.
 *
 *   greb_run <input_dir> <output_file> <time_flux> <time_scnr> <co2_ppm> [ipx ipy]
 *
 * Reads the reference's ten raw fp32 input files (src/greb.f90:1018-1027,1073-1085) from <input_dir>, runs the
 * flux-correction phase and a constant-CO2 scenario (src/greb.f90:219-234) on GPU 0 and writes the monthly means
 * in the reference's record order (five records of 96x48 per month, :978-982).  Prints the yearly console
 * values (:954).  Build:  cc examples/greb_run.c -Iinclude -Lgreb_climate_model_amd -lgreb_hip -Wl,-rpath,... */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "greb_engine.h"

enum { NX = 96, NY = 48, NP = NX * NY };

static float* read_file(const char* dir, const char* name, size_t n) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s", dir, name);
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "greb_run: cannot open %s\n", path); exit(2); }
  float* a = (float*)malloc(n * sizeof(float));
  if (!a || fread(a, sizeof(float), n, f) != n) { fprintf(stderr, "greb_run: %s is not %zu floats\n", path, n); exit(2); }
  fclose(f);
  return a;
}

static void check(int rc, greb_engine* e, const char* what) {
  if (rc == 0) return;
  fprintf(stderr, "greb_run: %s failed (%d): %s\n", what, rc, greb_engine_last_error(e));
  exit(1);
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s input_dir output_file time_flux time_scnr co2_ppm [ipx ipy]\n", argv[0]); return 2; }
  const char* dir = argv[1];
  const int time_flux = atoi(argv[3]), time_scnr = atoi(argv[4]);
  const float co2 = (float)atof(argv[5]);
  greb_params p;
  greb_params_default(&p);
  if (argc >= 8) { p.ipx = atoi(argv[6]); p.ipy = atoi(argv[7]); }
  const size_t n3 = (size_t)GREB_NSTEP_YR * NP;
  greb_fields f;
  f.z_topo = read_file(dir, "topography", NP);
  f.glacier = read_file(dir, "glacier.masks", NP);
  f.sw_solar = read_file(dir, "solar.radiation", (size_t)GREB_NSTEP_YR * NY);
  f.tclim = read_file(dir, "tsurf", n3);
  f.qclim = read_file(dir, "vapor", n3);
  f.uclim = read_file(dir, "zonal.wind", n3);
  f.vclim = read_file(dir, "meridional.wind", n3);
  f.mldclim = read_file(dir, "ocean.mld", n3);
  f.cldclim = read_file(dir, "cloud.cover", n3);
  f.swetclim = read_file(dir, "soil.moisture", n3);

  greb_engine* e = NULL;
  check(greb_engine_create(&p, NX, NY, &f, 1, NULL, 0, 0, &e), e, "greb_engine_create");
  float* yflux = (float*)calloc(2 * (size_t)(time_flux > 0 ? time_flux : 1), sizeof(float));
  check(greb_engine_flux_correction(e, time_flux, yflux), e, "greb_engine_flux_correction");
  for (int y = 0; y < time_flux; ++y) printf("%g %g %.6f %.6f\n", 0.0, p.co2_flux, yflux[2 * y], yflux[2 * y + 1]);

  if (time_scnr > 0) {
    const size_t nrec = (size_t)time_scnr * 12 * GREB_NVAR_OUT * NP;
    float* monthly = (float*)malloc(nrec * sizeof(float));
    float* yearly = (float*)calloc(2 * (size_t)time_scnr, sizeof(float));
    float* series = (float*)malloc((size_t)time_scnr * sizeof(float));
    for (int y = 0; y < time_scnr; ++y) series[y] = co2;
    check(greb_engine_run(e, time_scnr, series, monthly, yearly, 0), e, "greb_engine_run");
    for (int y = 0; y < time_scnr; ++y) printf("%d %g %.6f %.6f\n", p.year0 + y, co2, yearly[2 * y], yearly[2 * y + 1]);
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(monthly, sizeof(float), nrec, o) != nrec) { fprintf(stderr, "greb_run: cannot write %s\n", argv[2]); return 2; }
    fclose(o);
    free(monthly); free(yearly); free(series);
  }
  check(greb_engine_destroy(e), NULL, "greb_engine_destroy");
  return 0;
}
