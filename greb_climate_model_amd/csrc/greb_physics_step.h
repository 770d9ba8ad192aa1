// greb_physics_step.h -- one model step of point physics + Euler update + accumulation for one
// quad (4 consecutive longitudes) of one member (src/greb.f90:254-268 scenario, :328-361 flux
// correction, :945 / :974-984 accumulation).  Shared by the fused 96x48 member kernel (tracers in
// LDS) and the any-grid multi-launch engine (tracers in HBM): the caller supplies the tracers after
// the 24 circulation sub-steps (xTa, xq) and stores the new ones (oTa, oq).
#pragma once
#include "greb_kernels.h"

namespace greb {

__device__ __constant__ int kMonthEnd[12] = {31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334, 365};
__device__ __constant__ int kMonthDays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31}; // :42

struct StepClock {
  int ityr, mon, yr_rel; // step in year (1-based), month ending today (0-based) or -1, years since launch start
  size_t off, offm;      // offsets of climatology slices ityr and ityr-1 (:507-508)
};
template <bool FLUX>
__device__ __forceinline__ StepClock step_clock(const MemberArgs& a, long long it, int np) {
  StepClock c;
  c.ityr = (int)((it - 1) % kNT) + 1;               // :252
  const int jday = (int)(((it - 1) / 2) % 365) + 1; // :251
  c.off = (size_t)(c.ityr - 1) * np;
  c.offm = (size_t)(c.ityr > 1 ? c.ityr - 2 : kNT - 1) * np;
  c.yr_rel = (int)((it - 1) / kNT - (a.it0 - 1) / kNT);
  c.mon = -1;
  if (!FLUX && (it % 2 == 0))
    for (int mm = 0; mm < 12; ++mm) if (jday == kMonthEnd[mm]) c.mon = mm; // :975-976
  return c;
}

// One quad in three pieces -- load, compute, store -- so that a caller with several quads per thread can request the
// next quad's operands BETWEEN the arithmetic of the current one and its stores (vector-memory operations retire in
// order: loads issued behind a quad's stores wait for those stores as well).
// EXP: the sensitivity-experiment switches a.xsw are honoured; false compiles them out.
struct PhysIn {
  f4 Ts, Ta, To, q, cap;                      // state
  f4 zt, gl, zo, ez;                          // static fields
  f4 tcl, cld, mld, mldm, swet, u, v;         // forcing of the step (mldm: previous step, :507-508)
  f4 c0, c1, c2;                              // flux: Toclim, qclim, -- ; scenario: TF, qF, ToF
  f4 acc0, acc1, acc2, acc3, acc4, acc5;      // monthly sums, annual Tsurf sum
  f4 qcl, tclp;                               // experiments only
  float solar;
};
struct PhysOut { // everything the stores need: the inputs are dead once this exists
  f4 Ts, Ta, To, q, cap, TF, qF, ToF, tsmn;
  f4 s0, s1, s2, s3, s4; // the monthly sums including this step (:974)
};

template <bool FLUX, bool EXP>
__device__ __forceinline__ PhysIn physics_load(const MemberArgs& a, int qd, const StepClock& ck, const float* __restrict__ state,
                                               const float* __restrict__ acc, const float* __restrict__ corr) {
  const int nx = a.nx, ny = a.ny, np = a.np, p0 = 4 * qd;
  const size_t off = ck.off, offm = ck.offm;
  const unsigned xsw = EXP ? a.xsw : 0u;
  PhysIn i;
  i.Ts = ld4(state + p0); i.Ta = ld4(state + np + p0); i.To = ld4(state + 2 * np + p0); i.q = ld4(state + 3 * np + p0);
  i.cap = ld4(state + 4 * np + p0);
  i.zt = ld4(a.z_topo + p0); i.gl = ld4(a.glacier + p0); i.zo = ld4(a.z_ocean + p0); i.ez = ld4(a.wz_air + p0);
  i.tcl = ld4(a.tclim + off + p0); i.cld = ld4(a.cldclim + off + p0); i.mld = ld4(a.mldclim + off + p0);
  i.mldm = ld4(a.mldclim + offm + p0); i.swet = ld4(a.swetclim + off + p0); i.u = ld4(a.uclim + off + p0);
  i.v = ld4(a.vclim + off + p0);
  i.solar = a.sw_solar[(size_t)(ck.ityr - 1) * ny + p0 / nx]; // a quad never straddles rows
  if (FLUX) { i.c0 = ld4(a.toclim + p0); i.c1 = ld4(a.qclim + off + p0); i.c2 = zero4(); }
  else { i.c0 = ld4(corr + off + p0); i.c1 = ld4(corr + (size_t)kNT * np + off + p0); i.c2 = ld4(corr + (size_t)2 * kNT * np + off + p0); }
  i.acc0 = i.acc1 = i.acc2 = i.acc3 = i.acc4 = zero4();
#ifdef GREB_TUNING
  const bool no_acc = a.dbg & 1; // timing experiments (tools/stamp_member.py, GREB_DEBUG_PHYS)
#else
  constexpr bool no_acc = false;
#endif
  i.acc5 = no_acc ? zero4() : ld4(acc + 5 * np + p0);
  if (!FLUX && !no_acc) {
    i.acc0 = ld4(acc + p0); i.acc1 = ld4(acc + np + p0); i.acc2 = ld4(acc + 2 * np + p0); i.acc3 = ld4(acc + 3 * np + p0);
    i.acc4 = ld4(acc + 4 * np + p0);
  }
  i.qcl = zero4(); i.tclp = zero4(); // experiments only: qclim(ityr) for the linear emissivity, Tclim of the previous step
  if (EXP && (xsw & kXLwLinear)) i.qcl = FLUX ? i.c1 : ld4(a.qclim + off + p0);
  if (EXP && !FLUX && (xsw & kXSstPlus1)) i.tclp = ld4(a.tclim + offm + p0);
  return i;
}

// the point physics and the Euler update of the quad (src/greb.f90:254-268 scenario, :328-361 flux correction);
// xTa, xq: the tracers after the 24 circulation sub-steps
template <bool STRICT, bool FLUX, bool EXP>
__device__ __forceinline__ PhysOut physics_compute(const MemberArgs& a, const Phys& P, const PhysIn& in, float co2, const f4& xTa,
                                                   const f4& xq) {
  const unsigned xsw = EXP ? a.xsw : 0u;
  PhysOut o;
  o.TF = o.qF = o.ToF = zero4();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
    float Ts1 = in.Ts.v[e];
    const float Ta1 = in.Ta.v[e], To1 = in.To.v[e], q1 = in.q.v[e], cap = in.cap.v[e];
    const float zt = in.zt.v[e], gl = in.gl.v[e], ez = in.ez.v[e], tcl = in.tcl.v[e], cld = in.cld.v[e], mld = in.mld.v[e];
    if (EXP && !FLUX && (xsw & kXSstPlus1) && zt < 0.0f) Ts1 = in.tclp.v[e] + 1.0f; // greb.original.model.f90:226
    const float dTa_crcl = xTa.v[e] - Ta1; // :551
    const float dq_crcl = (EXP && (xsw & kXNoQTransport)) ? 0.f : xq.v[e] - q1; // greb.original.model.f90:554-555
    float albedo, sw, LWsurf, LWdown, em, Qlat, Qlat_air, dq_eva, dq_rain, dT_ocean, dTo;
    sw_radiation<STRICT>(P, Ts1, zt, gl, cld, in.solar, albedo, sw, xsw);
    lw_radiation<STRICT>(P, Ts1, Ta1, q1, co2, ez, cld, tcl, LWsurf, LWdown, em, xsw, in.qcl.v[e]);
    const float Qsens = P.ct_sens * (Ta1 - Ts1); // :295
    hydro<STRICT>(P, Ts1, q1, in.u.v[e], in.v.v[e], zt, ez, in.swet.v[e], Qlat, Qlat_air, dq_eva, dq_rain, xsw);
    deep_ocean<STRICT>(P, Ts1, To1, zt, mld, in.mldm.v[e], in.zo.v[e], dT_ocean, dTo, xsw);
    const float LWup = LWdown; // :432
    float Ts0, Ta0, To0, q0;
    if (FLUX) {
      const float dTs = fdiv<STRICT>(P.dt * (sw + LWsurf - LWdown + Qlat + Qsens), cap);                    // :333
      Ts0 = Ts1 + dTs + dT_ocean;                                                              // :334
      const float dTa = fdiv<STRICT>(P.dt * (LWup + LWdown - em * LWsurf + Qlat_air - Qsens), P.cap_air);   // :336
      Ta0 = Ta1 + dTa + dTa_crcl;                                                              // :337
      To0 = To1 + dTo;                                                                         // :339
      const float dq = P.dt * (dq_eva + dq_rain);                                              // :341
      q0 = q1 + dq + dq_crcl;                                                                  // :342
      const float TF = fdiv<STRICT>((tcl - Ts0) * cap, P.dt);                                               // :344-345
      Ts0 = Ts1 + dTs + dT_ocean + fdiv<STRICT>(TF * P.dt, cap);                                            // :347
      const float ToF = in.c0.v[e] - To0;                                                      // :349
      To0 = To1 + dTo + ToF;                                                                   // :351
      const float qF = in.c1.v[e] - q0;                                                        // :353
      q0 = q1 + dq + dq_crcl + qF;                                                             // :355
      o.TF.v[e] = TF; o.qF.v[e] = qF; o.ToF.v[e] = ToF;
    } else {
      const float TF = in.c0.v[e], qF = in.c1.v[e], ToF = in.c2.v[e];
      Ts0 = Ts1 + dT_ocean + fdiv<STRICT>(P.dt * (sw + LWsurf - LWdown + Qlat + Qsens + TF), cap);          // :258
      Ta0 = Ta1 + dTa_crcl + fdiv<STRICT>(P.dt * (LWup + LWdown - em * LWsurf + Qlat_air - Qsens), P.cap_air); // :260
      To0 = To1 + dTo + ToF;                                                                   // :262
      float dq = P.dt * (dq_eva + dq_rain) + dq_crcl + qF;                                     // :264
      if (dq <= -q1) dq = -0.9f * q1;                                                          // :265
      q0 = q1 + dq;                                                                            // :266
    }
    o.Ts.v[e] = Ts0; o.Ta.v[e] = Ta0; o.To.v[e] = To0; o.q.v[e] = q0;
    o.cap.v[e] = seaice<STRICT>(P, Ts0, zt, gl, mld, cap, xsw);                                              // :268/:357
    o.tsmn.v[e] = in.acc5.v[e] + Ts0;                                                          // :945
    o.s0.v[e] = in.acc0.v[e] + Ts0; o.s1.v[e] = in.acc1.v[e] + Ta0; o.s2.v[e] = in.acc2.v[e] + To0; // :974
    o.s3.v[e] = in.acc3.v[e] + q0; o.s4.v[e] = in.acc4.v[e] + albedo;
  }
  return o;
}

// state write-back, correction / accumulation / monthly records (:974-984), annual mean (:945-956).  Returns the
// annual-mean Tsurf quad in tsmn_mean when ityr == 730 (the caller feeds the global-mean sum, :954).
template <bool FLUX>
__device__ __forceinline__ void physics_store(const MemberArgs& a, int m, int qd, const StepClock& ck,
                                              const PhysOut& o, float* __restrict__ state, float* __restrict__ acc,
                                              float* __restrict__ corr, f4& tsmn_mean) {
  const int np = a.np, p0 = 4 * qd, mon = ck.mon;
  const size_t off = ck.off;
#ifdef GREB_TUNING
  const bool no_acc = a.dbg & 1, no_wb = a.dbg & 2;
#else
  constexpr bool no_acc = false, no_wb = false;
#endif
  if (!no_wb) {
    st4(state + p0, o.Ts); st4(state + np + p0, o.Ta); st4(state + 2 * np + p0, o.To); st4(state + 3 * np + p0, o.q);
    st4(state + 4 * np + p0, o.cap);
  }
  if (FLUX) {
    st4(corr + off + p0, o.TF); st4(corr + (size_t)kNT * np + off + p0, o.qF); st4(corr + (size_t)2 * kNT * np + off + p0, o.ToF);
  } else {
#pragma clang fp contract(off)
    f4 s0 = o.s0, s1 = o.s1, s2 = o.s2, s3 = o.s3, s4 = o.s4;
    if (mon >= 0) { // :975-984
      const float ndm = (float)(kMonthDays[mon] * 2);
      float* rec = a.monthly + (((size_t)m * a.monthly_years + (a.year_out0 + ck.yr_rel)) * 12 + mon) * 5 * np;
      f4 r0, r1, r2, r3, r4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r0.v[e] = s0.v[e] / ndm; r1.v[e] = s1.v[e] / ndm; r2.v[e] = s2.v[e] / ndm; r3.v[e] = s3.v[e] / ndm; r4.v[e] = s4.v[e] / ndm;
      }
      // written once, read by nobody on the device: keep the records out of the L2 / MALL the state lives in
      st4_nt(rec + p0, r0); st4_nt(rec + np + p0, r1); st4_nt(rec + 2 * np + p0, r2); st4_nt(rec + 3 * np + p0, r3); st4_nt(rec + 4 * np + p0, r4);
      s0 = s1 = s2 = s3 = s4 = zero4();
    }
    if (!no_acc) { st4(acc + p0, s0); st4(acc + np + p0, s1); st4(acc + 2 * np + p0, s2); st4(acc + 3 * np + p0, s3); st4(acc + 4 * np + p0, s4); }
  }
  if (ck.ityr == kNT) { // :948-956
#pragma clang fp contract(off)
    f4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t.v[e] = o.tsmn.v[e] / (float)kNT;
    tsmn_mean = t;
    if (!no_acc) st4(acc + 5 * np + p0, zero4());
  } else {
    if (!no_acc) st4(acc + 5 * np + p0, o.tsmn);
  }
}

// the three pieces in a row (one quad per thread: the any-grid engine)
template <bool STRICT, bool FLUX, bool EXP = false>
__device__ __forceinline__ void physics_quad(const MemberArgs& a, const Phys& P, int m, int qd, const StepClock& ck,
                                             float co2, float* __restrict__ state, float* __restrict__ acc,
                                             float* __restrict__ corr, const f4& xTa, const f4& xq, f4& oTa_out,
                                             f4& oq_out, f4& tsmn_mean) {
  const PhysIn in = physics_load<FLUX, EXP>(a, qd, ck, state, acc, corr);
  const PhysOut o = physics_compute<STRICT, FLUX, EXP>(a, P, in, co2, xTa, xq);
  physics_store<FLUX>(a, m, qd, ck, o, state, acc, corr, tsmn_mean);
  oTa_out = o.Ta; oq_out = o.q;
}

} // namespace greb
