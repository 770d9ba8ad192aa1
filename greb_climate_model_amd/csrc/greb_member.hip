// greb_member.hip -- the fused GREB member engine for MI355X (gfx950, wave64), 96x48 grid.
//
// One 512-thread workgroup (8 waves, 2 per SIMD) integrates one ensemble member for a whole launch
// (one model year): point physics, 2 x 24 circulation sub-steps, Euler update, sea ice,
// monthly/annual accumulation (src/greb.f90:239-364, 528-553) without leaving the CU.
//
// Residency
//   LDS (150 KB)  X[2 buffers][Tair,q][48][96]  the two transported tracers, double-buffered
//                 W[wz_air,wz_vapor][48][96]    the stencil weights (static for the run)
//                 WX, WY [48][96]               this step's winds, scaled by the row's advection
//                                               constants (FAST) or raw (STRICT and polar rows)
//                 scratch                       Jacobi row buffers of the polar rows
//   HBM/L2        Tsurf, Tocean, cap_surf, accumulators, forcing: touched once per model step
//   Registers hold nothing across sub-steps but a handful of per-row constants, so the stencil
//   code has the whole VGPR budget and never spills (v1 spilled and ran 10x slower).
//
// Wave roles (class-uniform, so no wave executes both stencil families)
//   waves 0-2  "sub"  tiles 4 lon x 3 lat over rows 1-9 and 38-46  (sub-cycled formulas, one sweep)
//   waves 3-5  "full" tiles 4 lon x 4 lat over rows 10-37          (full-row formulas)
//   waves 6-7  chain  the polar rows 0 / 47 (8 dependent Jacobi sweeps per diffusion call,
//              src/greb.f90:656-717), both tracers side by side in one wave (lanes 0-23 / 32-55)
// Every sub-step a bulk thread slides a 5-row window down its tile for each tracer, reading T and
// w as dwordx4 from LDS, computes X_new = (X + dX_diffuse) + dX_advec and writes the other
// buffer; one s_barrier per sub-step.
#include <cstdlib>

#include "greb_kernels.h"
#include "greb_stencil.h"

namespace greb {

constexpr int NX = 96, NY = 48, NQ = 24, NP = 4608;
constexpr int kThreads = 512;
// LDS map, in floats
constexpr int kOffX = 0;            // [2][2][NP]
constexpr int kOffW = 4 * NP;       // [2][NP]
constexpr int kOffWX = 6 * NP;      // [NP]  cu*u   (raw u in rows 0, 47 and in STRICT)
constexpr int kOffWY = 7 * NP;      // [NP]  ccy/3*v (raw v ...)
constexpr int kOffScr = 8 * NP;     // [2 poles][2 tracers][4][NX]
constexpr int kLdsFloats = kOffScr + 16 * NX;
constexpr size_t kLdsBytes = (size_t)kLdsFloats * sizeof(float);

// rows 0-9 / 38-47 sub-cycled, only rows 0 and 47 iterate (default kappa +-25 %)
bool member_layout_supported(const RowTables& t, int nx, int ny) {
  if (nx != NX || ny != NY) return false;
  for (int k = 0; k < NY; ++k) {
    const bool sub = (k <= 9 || k >= 38);
    if ((t.subcycled[k] != 0) != sub) return false;
    const bool chain = (k == 0 || k == NY - 1);
    if ((t.dif_time2[k] > 1 || t.adv_time2[k] > 1) != chain) return false;
  }
  return true;
}

// FAST fused sub-step of one quad with pre-multiplied winds (see greb_device.h)
template <bool SUB>
__device__ __forceinline__ f4 substep_pm(const QuadIn& in, const f4& um, const f4& up, const f4& vm, const f4& vp,
                                         float cs_dif, float ccy_dif, bool last_quad) {
  Flux f;
  make_flux(in.T, in.w, f);
  float ddx[4], dax[4], ddy[4], day[4];
  dif_lon_fast(f, cs_dif, ddx);
  if (SUB) {
    float T1h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
    float T2h[4] = {in.T[4], in.T[5], in.T[6], in.T[7]};
    clamp_add_fast(T1h, ddx);                                   // :715-716
    adv_lon_sub_pm(f, in.T, in.w, um.v, up.v, last_quad, dax);
    clamp_add_fast(T2h, dax);                                   // :907-908
#pragma unroll
    for (int i = 0; i < 4; ++i) { ddx[i] = T1h[i] - in.T[4 + i]; dax[i] = T2h[i] - in.T[4 + i]; } // :718, :910
  } else {
    adv_lon_full_pm(f, in.T, in.w, um.v, up.v, dax);
  }
  lat_pm(in.T0, in.Tm2, in.Tm1, in.Tp1, in.Tp2, in.wm2, in.wm1, in.wp1, in.wp2, vm.v, vp.v, ccy_dif, ddy, day);
  f4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float dd = in.w0.v[i] * (ddx[i] + ddy[i]); // :721
    const float da = dax[i] + day[i];                // :913
    r.v[i] = (in.T[4 + i] + dd) + da;                // :549
  }
  return r;
}

// per-row constants a bulk thread keeps in registers
struct TileK {
  RowK rk[4];
  float csd[4];      // dif_cc/20
  float fm[4], fp[4]; // 3 where the latitudinal advection term is not divided by 3 (:766-769,:784-787)
};

// one circulation sub-step of a bulk tile: both tracers, rows k0..k0+H-1
template <bool STRICT, int H>
__device__ __forceinline__ void tile_substep(lfloat* lds, int cur, int k0, int tx, const TileK& tk) {
  constexpr bool SUB = (H == 3);
  const int txm = tx == 0 ? NQ - 1 : tx - 1, txp = tx == NQ - 1 ? 0 : tx + 1;
  const int om = 4 * txm, oc = 4 * tx, op = 4 * txp;
#pragma unroll
  for (int tr = 0; tr < 2; ++tr) {
    const lfloat* Xc = lds + kOffX + (cur * 2 + tr) * NP;
    const lfloat* Wc = lds + kOffW + tr * NP;
    lfloat* Xn = lds + kOffX + ((cur ^ 1) * 2 + tr) * NP;
    // sliding window over latitude: 5 centre quads of T and w are live at a time
    f4 CT[H + 4], CW[H + 4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 - 2 + i;
      const int kc = k < 0 ? 0 : k;
      CT[i] = ld4(Xc + kc * NX + oc);
      CW[i] = k < 0 ? zero4() : ld4(Wc + kc * NX + oc); // rows outside the grid carry zero weight
    }
#pragma unroll
    for (int r = 0; r < H; ++r) {
      const int k = k0 + r;
      {
        const int kk = k + 2, kc = kk > NY - 1 ? NY - 1 : kk;
        CT[r + 4] = ld4(Xc + kc * NX + oc);
        CW[r + 4] = kk > NY - 1 ? zero4() : ld4(Wc + kc * NX + oc);
      }
      const f4 LT = ld4(Xc + k * NX + om), RT = ld4(Xc + k * NX + op);
      const f4 LW = ld4(Wc + k * NX + om), RW = ld4(Wc + k * NX + op);
      const f4 xq = ld4(lds + kOffWX + k * NX + oc), yq = ld4(lds + kOffWY + k * NX + oc);
      QuadIn in;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        in.T[j] = LT.v[j]; in.T[4 + j] = CT[r + 2].v[j]; in.T[8 + j] = RT.v[j];
        in.w[j] = LW.v[j]; in.w[4 + j] = CW[r + 2].v[j]; in.w[8 + j] = RW.v[j];
      }
      in.T0 = CT[r + 2]; in.w0 = CW[r + 2];
      in.Tm2 = CT[r]; in.Tm1 = CT[r + 1]; in.Tp1 = CT[r + 3]; in.Tp2 = CT[r + 4];
      in.wm2 = CW[r]; in.wm1 = CW[r + 1]; in.wp1 = CW[r + 3]; in.wp2 = CW[r + 4];
      f4 xn;
      if (STRICT) {
        float dd[4], da[4];
        dif_quad<true>(in, tk.rk[r], k, NY, dd);
        adv_quad<true>(in, xq.v, yq.v, tk.rk[r], k, NY, tx == NQ - 1, da); // raw winds
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int i = 0; i < 4; ++i) xn.v[i] = in.T[4 + i] + dd[i] + da[i]; // :549
        }
      } else {
        f4 um, up, vm, vp;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          um.v[i] = fmaxf(xq.v[i], 0.f); up.v[i] = fminf(xq.v[i], 0.f);
          vm.v[i] = tk.fm[r] * fmaxf(yq.v[i], 0.f); vp.v[i] = tk.fp[r] * fminf(yq.v[i], 0.f);
        }
        xn = substep_pm<SUB>(in, um, up, vm, vp, tk.csd[r], tk.rk[r].dif_ccy, tx == NQ - 1);
      }
      st4(Xn + k * NX + oc, xn);
      __builtin_amdgcn_sched_barrier(0); // do not hoist the next row's loads over this row's arithmetic
    }
  }
}

// the polar-row wave: row k (0 or 47) of both tracers, lanes 0-23 -> Tair, 32-55 -> q
template <bool STRICT>
__device__ __forceinline__ void chain_substep(lfloat* lds, int cur, int pole, const RowK& rk) {
  const int lane = threadIdx.x & 63;
  const int tr = lane >> 5, ql = lane & 31; // ql >= 24: idle lanes
  const int k = pole ? NY - 1 : 0;
  const Rows X{lds + kOffX + (cur * 2 + tr) * NP, 0, NX};
  const Rows W{lds + kOffW + tr * NP, 0, NX};
  const Rows U{lds + kOffWX, 0, NX}, V{lds + kOffWY, 0, NX}; // raw winds in the polar rows
  chain_row<STRICT>(X, W, U, V, rk, k, NQ, NY, ql, kChainFused, lds + kOffScr + (pole * 2 + tr) * 4 * NX,
                    lds + kOffX + ((cur ^ 1) * 2 + tr) * NP + k * NX);
}

// stage this step's winds (src/greb.f90:203-216, 732): raw for STRICT and for the polar rows,
// otherwise scaled by the row's advection constants so the sign split is one max/min per use:
//   x = c*u, c = ccx/3 (full rows) or ccx2/20 (sub-cycled rows);  y = ccy/3 * v
template <bool STRICT>
__device__ __forceinline__ void stage_winds(lfloat* lds, const float* __restrict__ u, const float* __restrict__ v,
                                            const RowTables* __restrict__ tab) {
  for (int i = threadIdx.x; i < NP / 4; i += kThreads) {
    const int k = i / NQ;
    f4 uq = ld4(u + 4 * i), vq = ld4(v + 4 * i);
    if (!(STRICT || k == 0 || k == NY - 1)) {
      const float cu = tab->subcycled[k] ? tab->adv_ccx2[k] * 0.05f : tab->adv_ccx[k] * (1.f / 3.f);
      const float cv = tab->adv_ccy * (1.f / 3.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) { uq.v[j] *= cu; vq.v[j] *= cv; }
    }
    st4(lds + kOffWX + 4 * i, uq); st4(lds + kOffWY + 4 * i, vq);
  }
}

__device__ __constant__ int kMonthEnd[12] = {31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334, 365};
__device__ __constant__ int kMonthDays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31}; // :42

struct Role {
  int kind; // 0 sub tile, 1 full tile, 2 chain, 3 idle lane
  int k0, tx, pole;
};
__device__ __forceinline__ Role my_role() {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  Role r{3, 1, 0, 0};
  if (wave < 3) {
    if (lane < 48) {
      const int t = wave * 48 + lane, tt = t % 72;
      r.kind = 0; r.k0 = (t >= 72 ? 38 : 1) + 3 * (tt / NQ); r.tx = tt % NQ;
    }
  } else if (wave < 6) {
    if (lane < 56) {
      const int t = (wave - 3) * 56 + lane;
      r.kind = 1; r.k0 = 10 + 4 * (t / NQ); r.tx = t % NQ;
    }
  } else {
    r.kind = 2; r.pole = wave - 6;
  }
  return r;
}

// the circulation loop shared by the member kernel and its test mirror
template <bool STRICT>
struct Circ {
  Role role;
  TileK tk;

  __device__ __forceinline__ void init(lfloat* lds, const float* wz_air, const float* wz_vapor,
                                       const RowTables* __restrict__ tab) {
    role = my_role();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = role.kind == 2 ? (role.pole ? NY - 1 : 0) : min(role.k0 + r, NY - 1);
      tk.rk[r] = row_consts(*tab, k);
      tk.csd[r] = tk.rk[r].dif_cc * 0.05f;
      tk.fm[r] = k == 1 ? 3.f : 1.f;
      tk.fp[r] = k == NY - 2 ? 3.f : 1.f;
    }
    for (int i = threadIdx.x; i < NP / 4; i += kThreads) {
      st4(lds + kOffW + 4 * i, ld4(wz_air + 4 * i));
      st4(lds + kOffW + NP + 4 * i, ld4(wz_vapor + 4 * i));
    }
  }

  // dbg: timing experiments only (tools/microbench_circ.py): bit0/1/2 skip sub / full / chain work
  __device__ __forceinline__ void substep(lfloat* lds, int cur, int dbg = 0) {
    if (role.kind == 0) { if (!(dbg & 1)) tile_substep<STRICT, 3>(lds, cur, role.k0, role.tx, tk); }
    else if (role.kind == 1) { if (!(dbg & 2)) tile_substep<STRICT, 4>(lds, cur, role.k0, role.tx, tk); }
    else if (role.kind == 2) { if (!(dbg & 4)) chain_substep<STRICT>(lds, cur, role.pole, tk.rk[0]); }
  }
};

// ---------------------------------------------------------------------------------------------
// test mirror of circulation() (src/greb.f90:528-553): one field per workgroup; the field is
// loaded into both tracer slots so the engine's code path is exercised unchanged
// ---------------------------------------------------------------------------------------------
template <bool STRICT>
__global__ __launch_bounds__(kThreads) void circulation_g96_kernel(const float* __restrict__ Xin,
                                                                   const float* __restrict__ wz,
                                                                   const float* __restrict__ ug,
                                                                   const float* __restrict__ vg,
                                                                   float* __restrict__ dX,
                                                                   const RowTables* __restrict__ tab, int nsub,
                                                                   int dbg) {
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  const size_t fo = (size_t)blockIdx.x * NP;
  Circ<STRICT> c;
  c.init(lds, wz + fo, wz + fo, tab);
  for (int i = threadIdx.x; i < NP / 4; i += kThreads) {
    const f4 x = ld4(Xin + fo + 4 * i);
    st4(lds + kOffX + 4 * i, x); st4(lds + kOffX + NP + 4 * i, x);
  }
  stage_winds<STRICT>(lds, ug + fo, vg + fo, tab);
  __syncthreads();
  int cur = 0;
#pragma unroll 1
  for (int tt = 0; tt < nsub; ++tt) {
    c.substep(lds, cur, dbg);
    __syncthreads();
    cur ^= 1;
  }
  for (int i = threadIdx.x; i < NP / 4; i += kThreads) {
    const f4 a = ld4(lds + kOffX + cur * 2 * NP + 4 * i), b = ld4(Xin + fo + 4 * i);
    st4(dX + fo + 4 * i, f4{{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]}}); // :551
  }
}

hipError_t launch_circulation_g96(const float* X, const float* wz, const float* u, const float* v, float* dX,
                                  const RowTables* tab_dev, int batch, int nsub, bool strict, hipStream_t s) {
  auto kern = strict ? circulation_g96_kernel<true> : circulation_g96_kernel<false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
  if (e != hipSuccess) return e;
  const char* env = getenv("GREB_DEBUG_SKIP"); // timing experiments only
  hipLaunchKernelGGL(kern, dim3(batch), dim3(kThreads), kLdsBytes, s, X, wz, u, v, dX, tab_dev, nsub, env ? atoi(env) : 0);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// the member kernel
// ---------------------------------------------------------------------------------------------
template <bool STRICT, bool FLUX>
__global__ __launch_bounds__(kThreads) void member_kernel(MemberArgs a) {
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  const int m = blockIdx.x, tid = threadIdx.x;
  float* state = a.state + (size_t)m * 5 * NP;
  float* acc = a.acc + (size_t)m * 6 * NP;
  float* corr = a.corr + (size_t)a.corr_index[m] * 3 * kNT * NP;
  const RowTables* tab = a.tabs + a.tab_index[m];

  Circ<STRICT> circ;
  circ.init(lds, a.wz_air, a.wz_vapor, tab);
  for (int i = tid; i < NP / 4; i += kThreads) {
    st4(lds + kOffX + 4 * i, ld4(state + NP + 4 * i));          // Tair
    st4(lds + kOffX + NP + 4 * i, ld4(state + 3 * NP + 4 * i)); // q
  }
  int cur = 0;
  __syncthreads();

#pragma unroll 1
  for (int s = 0; s < a.nsteps; ++s) {
    const long long it = a.it0 + s;
    const int ityr = (int)((it - 1) % kNT) + 1;                       // :252
    const int jday = (int)(((it - 1) / 2) % 365) + 1;                 // :251
    const size_t off = (size_t)(ityr - 1) * NP;
    const size_t offm = (size_t)(ityr > 1 ? ityr - 2 : kNT - 1) * NP; // :507-508
    const int yr_rel = (int)((it - 1) / kNT - (a.it0 - 1) / kNT);     // whole years since launch start

    stage_winds<STRICT>(lds, a.uclim + off, a.vclim + off, tab);
    __syncthreads();

    // ---- circulation of Tair and q: 24 sub-steps (:543-550)
#pragma unroll 1
    for (int tt = 0; tt < a.nsub; ++tt) {
      circ.substep(lds, cur);
      __syncthreads();
      cur ^= 1;
    }

    // ---- point physics on the OLD state + Euler update (:254-268 / :328-361)
    int mon = -1; // 0-based month whose last day this is, else -1
    if (!FLUX && (it % 2 == 0))
      for (int mm = 0; mm < 12; ++mm) if (jday == kMonthEnd[mm]) mon = mm; // :975-976
    const Phys P = a.phys[m];
    const float co2 = FLUX ? a.co2_flux : a.co2[(size_t)m * a.co2_stride + a.co2_year0 + yr_rel]; // :924
    lfloat* Xf = lds + kOffX + cur * 2 * NP;       // the tracers after the 24 sub-steps
    lfloat* red = lds + kOffX + (cur ^ 1) * 2 * NP; // idle buffer: annual-mean reduction scratch
    // Each thread takes whole quads (4 consecutive longitudes): every load/store of the ~26
    // fields a point touches is one dwordx4 and all of a quad's loads are in flight together, so
    // a step pays ~3 dependent HBM/L2 round trips per thread instead of 9.
#pragma unroll 1
    for (int qd = tid; qd < NP / 4; qd += kThreads) {
      const int p0 = 4 * qd;
      const f4 vTs = ld4(state + p0), vTa = ld4(state + NP + p0), vTo = ld4(state + 2 * NP + p0),
               vq = ld4(state + 3 * NP + p0), vcap = ld4(state + 4 * NP + p0);
      const f4 vzt = ld4(a.z_topo + p0), vgl = ld4(a.glacier + p0), vzo = ld4(a.z_ocean + p0), vez = ld4(a.wz_air + p0);
      const f4 vtcl = ld4(a.tclim + off + p0), vcld = ld4(a.cldclim + off + p0), vmld = ld4(a.mldclim + off + p0),
               vmldm = ld4(a.mldclim + offm + p0), vswet = ld4(a.swetclim + off + p0), vu = ld4(a.uclim + off + p0),
               vv = ld4(a.vclim + off + p0);
      const float solar = a.sw_solar[(size_t)(ityr - 1) * NY + p0 / NX]; // a quad never straddles rows
      f4 vc0, vc1, vc2; // flux: Toclim, qclim, -- ; scenario: TF, qF, ToF
      if (FLUX) { vc0 = ld4(a.toclim + p0); vc1 = ld4(a.qclim + off + p0); vc2 = zero4(); }
      else { vc0 = ld4(corr + off + p0); vc1 = ld4(corr + (size_t)kNT * NP + off + p0); vc2 = ld4(corr + (size_t)2 * kNT * NP + off + p0); }
      f4 acc0, acc1, acc2, acc3, acc4;
      const f4 acc5 = ld4(acc + 5 * NP + p0);
      if (!FLUX) { acc0 = ld4(acc + p0); acc1 = ld4(acc + NP + p0); acc2 = ld4(acc + 2 * NP + p0); acc3 = ld4(acc + 3 * NP + p0); acc4 = ld4(acc + 4 * NP + p0); }
      const f4 xTa = ld4(Xf + p0), xq = ld4(Xf + NP + p0);
      f4 oTs, oTa, oTo, oq, ocap, oTF, oqF, oToF, oalb, otsmn;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
        const float Ts1 = vTs.v[e], Ta1 = vTa.v[e], To1 = vTo.v[e], q1 = vq.v[e], cap = vcap.v[e];
        const float zt = vzt.v[e], gl = vgl.v[e], ez = vez.v[e], tcl = vtcl.v[e], cld = vcld.v[e], mld = vmld.v[e];
        const float dTa_crcl = xTa.v[e] - Ta1; // :551
        const float dq_crcl = xq.v[e] - q1;
        float albedo, sw, LWsurf, LWdown, em, Qlat, Qlat_air, dq_eva, dq_rain, dT_ocean, dTo;
        sw_radiation(P, Ts1, zt, gl, cld, solar, albedo, sw);
        lw_radiation(P, Ts1, Ta1, q1, co2, ez, cld, tcl, LWsurf, LWdown, em);
        const float Qsens = P.ct_sens * (Ta1 - Ts1); // :295
        hydro(P, Ts1, q1, vu.v[e], vv.v[e], zt, ez, vswet.v[e], Qlat, Qlat_air, dq_eva, dq_rain);
        deep_ocean(P, Ts1, To1, zt, mld, vmldm.v[e], vzo.v[e], dT_ocean, dTo);
        const float LWup = LWdown; // :432
        float Ts0, Ta0, To0, q0;
        if (FLUX) {
          const float dTs = P.dt * (sw + LWsurf - LWdown + Qlat + Qsens) / cap;                    // :333
          Ts0 = Ts1 + dTs + dT_ocean;                                                              // :334
          const float dTa = P.dt * (LWup + LWdown - em * LWsurf + Qlat_air - Qsens) / P.cap_air;   // :336
          Ta0 = Ta1 + dTa + dTa_crcl;                                                              // :337
          To0 = To1 + dTo;                                                                         // :339
          const float dq = P.dt * (dq_eva + dq_rain);                                              // :341
          q0 = q1 + dq + dq_crcl;                                                                  // :342
          const float TF = (tcl - Ts0) * cap / P.dt;                                               // :344-345
          Ts0 = Ts1 + dTs + dT_ocean + TF * P.dt / cap;                                            // :347
          const float ToF = vc0.v[e] - To0;                                                        // :349
          To0 = To1 + dTo + ToF;                                                                   // :351
          const float qF = vc1.v[e] - q0;                                                          // :353
          q0 = q1 + dq + dq_crcl + qF;                                                             // :355
          oTF.v[e] = TF; oqF.v[e] = qF; oToF.v[e] = ToF;
        } else {
          const float TF = vc0.v[e], qF = vc1.v[e], ToF = vc2.v[e];
          Ts0 = Ts1 + dT_ocean + P.dt * (sw + LWsurf - LWdown + Qlat + Qsens + TF) / cap;          // :258
          Ta0 = Ta1 + dTa_crcl + P.dt * (LWup + LWdown - em * LWsurf + Qlat_air - Qsens) / P.cap_air; // :260
          To0 = To1 + dTo + ToF;                                                                   // :262
          float dq = P.dt * (dq_eva + dq_rain) + dq_crcl + qF;                                     // :264
          if (dq <= -q1) dq = -0.9f * q1;                                                          // :265
          q0 = q1 + dq;                                                                            // :266
        }
        oTs.v[e] = Ts0; oTa.v[e] = Ta0; oTo.v[e] = To0; oq.v[e] = q0;
        ocap.v[e] = seaice(P, Ts0, zt, gl, mld, cap);                                              // :268/:357
        oalb.v[e] = albedo;
        otsmn.v[e] = acc5.v[e] + Ts0;                                                              // :945
      }
      st4(state + p0, oTs); st4(state + NP + p0, oTa); st4(state + 2 * NP + p0, oTo); st4(state + 3 * NP + p0, oq);
      st4(state + 4 * NP + p0, ocap);
      st4(Xf + p0, oTa); st4(Xf + NP + p0, oq);
      if (FLUX) {
        st4(corr + off + p0, oTF); st4(corr + (size_t)kNT * NP + off + p0, oqF); st4(corr + (size_t)2 * kNT * NP + off + p0, oToF);
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
          float* rec = a.monthly + (((size_t)m * a.monthly_years + (a.year_out0 + yr_rel)) * 12 + mon) * 5 * NP;
          f4 r0, r1, r2, r3, r4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            r0.v[e] = s0.v[e] / ndm; r1.v[e] = s1.v[e] / ndm; r2.v[e] = s2.v[e] / ndm; r3.v[e] = s3.v[e] / ndm; r4.v[e] = s4.v[e] / ndm;
          }
          st4(rec + p0, r0); st4(rec + NP + p0, r1); st4(rec + 2 * NP + p0, r2); st4(rec + 3 * NP + p0, r3); st4(rec + 4 * NP + p0, r4);
          s0 = s1 = s2 = s3 = s4 = zero4();
        }
        st4(acc + p0, s0); st4(acc + NP + p0, s1); st4(acc + 2 * NP + p0, s2); st4(acc + 3 * NP + p0, s3); st4(acc + 4 * NP + p0, s4);
      }
      if (ityr == kNT) { // :948-956
#pragma clang fp contract(off)
        f4 t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t.v[e] = otsmn.v[e] / (float)kNT;
        st4(red + p0, t);
        st4(acc + 5 * NP + p0, zero4());
      } else {
        st4(acc + 5 * NP + p0, otsmn);
      }
    }
    if (ityr == kNT) {
      __syncthreads();
      if (tid == 0 && a.yearly) {
#pragma clang fp contract(off)
        float sum = 0.f; // the reference's sum() lowers to a sequential fp32 loop; same order here
        for (int i = 0; i < NP; ++i) sum += red[i];
        float* y = a.yearly + ((size_t)m * a.yearly_years + (a.yearly_year0 + yr_rel)) * 2;
        y[0] = sum / (float)NP - 273.15f;                                               // :954
        y[1] = red[(a.ipy - 1) * NX + (a.ipx - 1)] - 273.15f;
      }
    }
    __syncthreads();
  }
  // all five state fields were written back every step
}

hipError_t launch_member_kernel(const MemberArgs& a, int n_members, bool strict, hipStream_t s) {
  if (a.nx != NX || a.ny != NY) return hipErrorInvalidValue;
  void (*kern)(MemberArgs);
  if (a.flux_phase) kern = strict ? member_kernel<true, true> : member_kernel<false, true>;
  else kern = strict ? member_kernel<true, false> : member_kernel<false, false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(n_members), dim3(kThreads), kLdsBytes, s, a);
  return hipGetLastError();
}

} // namespace greb
