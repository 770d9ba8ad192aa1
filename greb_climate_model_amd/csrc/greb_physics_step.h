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

// state/acc/corr: this member's arrays.  Returns the annual-mean Tsurf quad in tsmn_mean when
// ityr == 730 (the caller feeds the sequential global-mean sum, :954).
// EXP: the sensitivity-experiment switches a.xsw are honoured; false compiles them out.
template <bool STRICT, bool FLUX, bool EXP = false>
__device__ __forceinline__ void physics_quad(const MemberArgs& a, const Phys& P, int m, int qd, const StepClock& ck,
                                             float co2, float* __restrict__ state, float* __restrict__ acc,
                                             float* __restrict__ corr, const f4& xTa, const f4& xq, f4& oTa_out,
                                             f4& oq_out, f4& tsmn_mean) {
  const int nx = a.nx, ny = a.ny, np = a.np;
  const int ityr = ck.ityr, mon = ck.mon, yr_rel = ck.yr_rel;
  const size_t off = ck.off, offm = ck.offm;
  const unsigned xsw = EXP ? a.xsw : 0u;
    const int p0 = 4 * qd;
    const f4 vTs = ld4(state + p0), vTa = ld4(state + np + p0), vTo = ld4(state + 2 * np + p0),
             vq = ld4(state + 3 * np + p0), vcap = ld4(state + 4 * np + p0);
    const f4 vzt = ld4(a.z_topo + p0), vgl = ld4(a.glacier + p0), vzo = ld4(a.z_ocean + p0), vez = ld4(a.wz_air + p0);
    const f4 vtcl = ld4(a.tclim + off + p0), vcld = ld4(a.cldclim + off + p0), vmld = ld4(a.mldclim + off + p0),
             vmldm = ld4(a.mldclim + offm + p0), vswet = ld4(a.swetclim + off + p0), vu = ld4(a.uclim + off + p0),
             vv = ld4(a.vclim + off + p0);
    const float solar = a.sw_solar[(size_t)(ityr - 1) * ny + p0 / nx]; // a quad never straddles rows
    f4 vc0, vc1, vc2; // flux: Toclim, qclim, -- ; scenario: TF, qF, ToF
    if (FLUX) { vc0 = ld4(a.toclim + p0); vc1 = ld4(a.qclim + off + p0); vc2 = zero4(); }
    else { vc0 = ld4(corr + off + p0); vc1 = ld4(corr + (size_t)kNT * np + off + p0); vc2 = ld4(corr + (size_t)2 * kNT * np + off + p0); }
    f4 acc0, acc1, acc2, acc3, acc4;
#ifdef GREB_TUNING
    const bool no_acc = a.dbg & 1, no_wb = a.dbg & 2; // timing experiments (tools/stamp_member.py, GREB_DEBUG_PHYS)
    const f4 acc5 = no_acc ? zero4() : ld4(acc + 5 * np + p0);
    if (!FLUX) {
      if (no_acc) { acc0 = acc1 = acc2 = acc3 = acc4 = zero4(); }
      else { acc0 = ld4(acc + p0); acc1 = ld4(acc + np + p0); acc2 = ld4(acc + 2 * np + p0); acc3 = ld4(acc + 3 * np + p0); acc4 = ld4(acc + 4 * np + p0); }
    }
#else
    constexpr bool no_acc = false, no_wb = false;
    const f4 acc5 = ld4(acc + 5 * np + p0);
    if (!FLUX) { acc0 = ld4(acc + p0); acc1 = ld4(acc + np + p0); acc2 = ld4(acc + 2 * np + p0); acc3 = ld4(acc + 3 * np + p0); acc4 = ld4(acc + 4 * np + p0); }
#endif
    f4 vqcl = zero4(), vtclp = zero4(); // experiments only: qclim(ityr) for the linear emissivity, Tclim of the previous step
    if (EXP && (xsw & kXLwLinear)) vqcl = FLUX ? vc1 : ld4(a.qclim + off + p0);
    if (EXP && !FLUX && (xsw & kXSstPlus1)) vtclp = ld4(a.tclim + offm + p0);
    f4 oTs, oTa, oTo, oq, ocap, oTF, oqF, oToF, oalb, otsmn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
      float Ts1 = vTs.v[e];
      const float Ta1 = vTa.v[e], To1 = vTo.v[e], q1 = vq.v[e], cap = vcap.v[e];
      const float zt = vzt.v[e], gl = vgl.v[e], ez = vez.v[e], tcl = vtcl.v[e], cld = vcld.v[e], mld = vmld.v[e];
      if (EXP && !FLUX && (xsw & kXSstPlus1) && zt < 0.0f) Ts1 = vtclp.v[e] + 1.0f; // greb.original.model.f90:226
      const float dTa_crcl = xTa.v[e] - Ta1; // :551
      const float dq_crcl = (EXP && (xsw & kXNoQTransport)) ? 0.f : xq.v[e] - q1; // greb.original.model.f90:554-555
      float albedo, sw, LWsurf, LWdown, em, Qlat, Qlat_air, dq_eva, dq_rain, dT_ocean, dTo;
      sw_radiation<STRICT>(P, Ts1, zt, gl, cld, solar, albedo, sw, xsw);
      lw_radiation<STRICT>(P, Ts1, Ta1, q1, co2, ez, cld, tcl, LWsurf, LWdown, em, xsw, vqcl.v[e]);
      const float Qsens = P.ct_sens * (Ta1 - Ts1); // :295
      hydro<STRICT>(P, Ts1, q1, vu.v[e], vv.v[e], zt, ez, vswet.v[e], Qlat, Qlat_air, dq_eva, dq_rain, xsw);
      deep_ocean<STRICT>(P, Ts1, To1, zt, mld, vmldm.v[e], vzo.v[e], dT_ocean, dTo, xsw);
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
        const float ToF = vc0.v[e] - To0;                                                        // :349
        To0 = To1 + dTo + ToF;                                                                   // :351
        const float qF = vc1.v[e] - q0;                                                          // :353
        q0 = q1 + dq + dq_crcl + qF;                                                             // :355
        oTF.v[e] = TF; oqF.v[e] = qF; oToF.v[e] = ToF;
      } else {
        const float TF = vc0.v[e], qF = vc1.v[e], ToF = vc2.v[e];
        Ts0 = Ts1 + dT_ocean + fdiv<STRICT>(P.dt * (sw + LWsurf - LWdown + Qlat + Qsens + TF), cap);          // :258
        Ta0 = Ta1 + dTa_crcl + fdiv<STRICT>(P.dt * (LWup + LWdown - em * LWsurf + Qlat_air - Qsens), P.cap_air); // :260
        To0 = To1 + dTo + ToF;                                                                   // :262
        float dq = P.dt * (dq_eva + dq_rain) + dq_crcl + qF;                                     // :264
        if (dq <= -q1) dq = -0.9f * q1;                                                          // :265
        q0 = q1 + dq;                                                                            // :266
      }
      oTs.v[e] = Ts0; oTa.v[e] = Ta0; oTo.v[e] = To0; oq.v[e] = q0;
      ocap.v[e] = seaice<STRICT>(P, Ts0, zt, gl, mld, cap, xsw);                                              // :268/:357
      oalb.v[e] = albedo;
      otsmn.v[e] = acc5.v[e] + Ts0;                                                              // :945
    }
    if (!no_wb) {
      st4(state + p0, oTs); st4(state + np + p0, oTa); st4(state + 2 * np + p0, oTo); st4(state + 3 * np + p0, oq);
      st4(state + 4 * np + p0, ocap);
    }
    if (FLUX) {
      st4(corr + off + p0, oTF); st4(corr + (size_t)kNT * np + off + p0, oqF); st4(corr + (size_t)2 * kNT * np + off + p0, oToF);
    } else {
#pragma clang fp contract(off)
      f4 s0, s1, s2, s3, s4; // :974
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0.v[e] = acc0.v[e] + oTs.v[e]; s1.v[e] = acc1.v[e] + oTa.v[e]; s2.v[e] = acc2.v[e] + oTo.v[e];
        s3.v[e] = acc3.v[e] + oq.v[e]; s4.v[e] = acc4.v[e] + oalb.v[e];
      }
      if (mon >= 0) { // :975-984
        const float ndm = (float)(kMonthDays[mon] * 2);
        float* rec = a.monthly + (((size_t)m * a.monthly_years + (a.year_out0 + yr_rel)) * 12 + mon) * 5 * np;
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
    if (ityr == kNT) { // :948-956
#pragma clang fp contract(off)
      f4 t;
#pragma unroll
      for (int e = 0; e < 4; ++e) t.v[e] = otsmn.v[e] / (float)kNT;
      tsmn_mean = t;
      if (!no_acc) st4(acc + 5 * np + p0, zero4());
    } else {
      if (!no_acc) st4(acc + 5 * np + p0, otsmn);
    }
  
  oTa_out = oTa; oq_out = oq;
}

} // namespace greb
