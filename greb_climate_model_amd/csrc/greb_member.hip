// greb_member.hip -- the fused GREB member engine for MI355X (gfx950, wave64), 96x48 grid.
//
// One 512-thread workgroup integrates one ensemble member for a whole launch (one model year):
// point physics, 2 x 24 circulation sub-steps, Euler update, sea ice, monthly/annual accumulation
// (src/greb.f90:239-364, 528-553) without leaving the CU.
//
// Residency
//   LDS (157 KB)  X[2 buffers][48][96][{Tair,q}]  the two transported tracers, INTERLEAVED and
//                                                 double-buffered
//                 W[48][96][{wz_air,wz_vapor}]    the stencil weights (static for the run)
//                 WX, WY [48][96]                 this step's winds, scaled by the row's advection
//                                                 constants (FAST) or raw (STRICT and polar rows)
//                 row constants, Jacobi row buffers of the polar rows
//   HBM/L2        Tsurf, Tocean, cap_surf, accumulators, forcing: touched once per model step
//   Registers hold nothing across sub-steps.
//
// The tracer interleave makes every FAST arithmetic instruction a packed v_pk_*_f32 on an aligned
// (Tair,q) register pair (greb_pair.h): both tracers share the winds, the sign split, the address
// arithmetic and the row constants.
//
// Work distribution per sub-step (measured on MI355X: the bulk is the critical path, bound by VALU issue on the
// busiest SIMD and by LDS bandwidth; a wave issues at most ~1 instruction per 4.5 cycles, a SIMD ~1 per 3):
//   a TASK works on pair-quads: 4 longitudes x (Tair,q) of one row, or of two vertically stacked rows that share
//   the six rows of their column (44 instead of 62 ds_read_b128).  The 46 non-polar rows are the "sub" family
//   (rows 1-9, 38-46: sub-cycled formulas, one sweep) and the "full" family (rows 10-37).  A PASS is 64 tasks of
//   one kind, one per lane; FAST: 3 passes of sub row-pairs, 3 of full row-pairs, 1 + 4.5 of single rows,
//   in both arithmetic modes; dealt statically to the seven (STRICT: six) bulk waves (see "The schedule") with
//   each lane's task addresses computed once per launch.
//   polar rows 0 / 47 (8 dependent Jacobi sweeps per diffusion call, src/greb.f90:656-717): FAST -- ONE wave, each
//   DPP row of 16 lanes x 6 longitudes one (pole, tracer) chain, neighbours by row rotates (quad_chain_substep);
//   STRICT -- one wave per pole, 48 lanes x 2 longitudes x (Tair,q), neighbours by ds_bpermute (chain_substep).
// One s_barrier per sub-step.
#include <cstdlib>

#include "greb_kernels.h"
#include "greb_pair.h"
#include "greb_physics_step.h"
#include "greb_stencil.h"

namespace greb {

constexpr int NX = 96, NY = 48, NQ = 24, NP = 4608;
constexpr int kThreads = 512;
// floats per interleaved row: 2*NX = 192 data + 16 of padding, i.e. a row stride of 52 sixteen-byte slots = 4 mod 16.
// A ds_read_b128 is served in groups of 16 lanes that must hit 16 different slot classes (slot mod 16) to take one
// LDS cycle.  With the unpadded stride (48 slots = 0 mod 16) every row started in the same class and the lanes of a
// pass, which span two to three rows, collided two-way on every read (42 % of the LDS cycles of the round-1 kernel
// were bank conflicts); with 4 mod 16, rows two apart -- the neighbouring row pairs of a pass -- are offset by 8
// classes and the 16 lanes of every group fall on distinct classes.  (The 9.6 KB this costs are those of the polar
// chains' former Jacobi row buffers plus the slack: the chains exchange their halos by ds_bpermute now.)
constexpr int RS = 2 * NX + 16;
// LDS map, in floats
// X buffers and W carry one all-zero GUARD row below row 0 and above row NY-1: the rows k-2 .. k+2 of any bulk
// row are then five rows at constant strides from ONE base address (immediate offsets of the ds_read), and the
// missing neighbours of rows 1 and NY-2 multiply to zero without selects.
constexpr int XB = (NY + 2) * RS;   // floats per guarded buffer
constexpr int kOffX = RS;           // row 0 of X[0]; X[b] row k at kOffX + b*XB + k*RS
constexpr int kOffW = 2 * XB + RS;  // row 0 of W  [NY][NX][{wz_air,wz_vapor}]
constexpr int kOffWX = 3 * XB;      // [NP]  cu*u   (raw u in rows 0, 47 and in STRICT)
constexpr int kOffWY = kOffWX + NP; // [NP]  ccy/3*v (raw v ...)
constexpr int kOffRowK = kOffWY + NP; // [NY][kRowKWords]
constexpr int kLdsFloats = kOffRowK + NY * kRowKWords;
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
  // the FAST engine runs both poles' chains in one wavefront with one (scalar) trip count; cos(-lat) = cos(lat)
  // makes the two rows' constants identical in the reference, so this never fails for its grid
  if (t.dif_time2[0] != t.dif_time2[NY - 1] || t.adv_time2[0] != t.adv_time2[NY - 1]) return false;
  return true;
}

// split a pair-quad into the two tracers' scalar quads (STRICT path)
__device__ __forceinline__ f4 comp(const q8& q, int tr) {
  return tr == 0 ? f4{{q.v[0].x, q.v[1].x, q.v[2].x, q.v[3].x}} : f4{{q.v[0].y, q.v[1].y, q.v[2].y, q.v[3].y}};
}
// interleave two fields' quads into one pair-quad
__device__ __forceinline__ q8 zip(const f4& a, const f4& b) {
  q8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = v2{a.v[i], b.v[i]};
  return r;
}

// ---------------------------------------------------------------------------------------------
// one bulk task: pair-quad (k, q) -> X_new = (X + dX_diffuse) + dX_advec, both tracers
// ---------------------------------------------------------------------------------------------
// A lane's bulk task of one pass, fixed for the whole launch: float offsets (inside a guarded buffer) of its own
// pair-quad and of the quads to its left and right, and (row | quad << 8).  Computed once (make_tasks) and kept in
// VGPRs: recomputing them -- a division by 24, wrap-around selects and a dozen address adds -- cost ~15 % of a
// pass's issue slots in the sub-step loop.
struct TaskAddr { int c, l, r, kq; };

// a pair-quad at a float offset p (two dwordx4: the [half][quad][4] row layout of greb_pair.h)
__device__ __forceinline__ q8 ld8p(const lfloat* p) {
  const vfloat4 a = *(const __attribute__((address_space(3))) vfloat4*)p;
  const vfloat4 b = *(const __attribute__((address_space(3))) vfloat4*)(p + kHalfRow);
  q8 r;
  r.v[0] = v2{a.x, a.y}; r.v[1] = v2{a.z, a.w}; r.v[2] = v2{b.x, b.y}; r.v[3] = v2{b.z, b.w};
  return r;
}

// FAST arithmetic of one bulk row task from its loaded neighbourhood (shared by the one-row and the two-row task)
template <bool SUB>
__device__ __forceinline__ q8 fast_row(const q8& LT, const q8& CT, const q8& RT, const q8& Tm2, const q8& Tm1, const q8& Tp1,
                                       const q8& Tp2, const q8& LW, const q8& CW, const q8& RW, const q8& Wm2, const q8& Wm1,
                                       const q8& Wp1, const q8& Wp2, const f4& xq, const f4& yq, int k, bool last_quad,
                                       float dif_cc, float dif_ccy, bool calm_q) {
  v2 T[12], w[12];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    T[j] = LT.v[j]; T[4 + j] = CT.v[j]; T[8 + j] = RT.v[j];
    w[j] = LW.v[j]; w[4 + j] = CW.v[j]; w[8 + j] = RW.v[j];
  }
  // the latitudinal advection term is not divided by 3 at k = 1 (v>=0 part) and k = ny-2 (v<0 part)
  // (only rows 1 and NY-2, which are in the sub-cycled family: the full family skips the two multiplications)
  const float fm = SUB && k == 1 ? 3.f : 1.f, fp = SUB && k == NY - 2 ? 3.f : 1.f; // :766-769, :784-787
  float um[4], up[4], vm[4], vp[4]; // the sign split is shared by the two tracers
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    split_sign(xq.v[i], um[i], up[i]);
    float a, b;
    split_sign(yq.v[i], a, b);
    vm[i] = SUB ? fm * a : a; vp[i] = SUB ? fp * b : b;
  }
  return substep_pair<SUB>(T, w, Tm2, Tm1, Tp1, Tp2, Wm2, Wm1, Wp1, Wp2, um, up, vm, vp, dif_cc * 0.05f, dif_ccy, last_quad,
                           calm_q);
}

__device__ __forceinline__ void st8p(lfloat* o, const q8& xn) {
  vfloat4 a, b;
  a.x = xn.v[0].x; a.y = xn.v[0].y; a.z = xn.v[1].x; a.w = xn.v[1].y;
  b.x = xn.v[2].x; b.y = xn.v[2].y; b.z = xn.v[3].x; b.w = xn.v[3].y;
  *(__attribute__((address_space(3))) vfloat4*)o = a;
  *(__attribute__((address_space(3))) vfloat4*)(o + kHalfRow) = b;
}

// STRICT arithmetic of one bulk row task from its loaded neighbourhood: the reference's expression trees, both
// tracers packed -- with contraction off every v2 operation is the reference's IEEE operation on each half.  Raw winds.
__device__ __forceinline__ q8 strict_row(const q8& LT, const q8& CT, const q8& RT, const q8& Tm2, const q8& Tm1, const q8& Tp1,
                                         const q8& Tp2, const q8& LW, const q8& CW, const q8& RW, const q8& Wm2, const q8& Wm1,
                                         const q8& Wp1, const q8& Wp2, const f4& xq, const f4& yq, int k, int q,
                                         const RowK& rk, bool calm_q) {
#pragma clang fp contract(off)
  q8 xn;
  v2 T[12], w[12];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    T[j] = LT.v[j]; T[4 + j] = CT.v[j]; T[8 + j] = RT.v[j];
    w[j] = LW.v[j]; w[4 + j] = CW.v[j]; w[8 + j] = RW.v[j];
  }
  v2 dTx[4], dTy[4], dd[4], da[4];
  // diffusion (dif_quad<true>)
  dif_lon_strict(T, w, rk.dif_cc, dTx);
  if (rk.sub) {
    v2 T1h[4] = {T[4], T[5], T[6], T[7]};
    clamp_add(T1h, dTx);
#pragma unroll
    for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - T[4 + i]; // :718
  }
  dif_lat_strict(CT, Tm1, Tp1, Wm1, Wp1, rk.dif_ccy, k, NY, dTy);
#pragma unroll
  for (int i = 0; i < 4; ++i) dd[i] = CW.v[i] * (dTx[i] + dTy[i]); // :721
  // advection (adv_quad<true>), raw winds
  if (!rk.sub) {
    adv_lon_full_strict(T, w, xq.v, rk.adv_cc, dTx);
  } else {
    v2 T1h[4] = {T[4], T[5], T[6], T[7]};
    adv_lon_sub_strict(T, w, xq.v, rk.adv_cc, q == NQ - 1, dTx);
    clamp_add(T1h, dTx);
#pragma unroll
    for (int i = 0; i < 4; ++i) dTx[i] = T1h[i] - T[4 + i]; // :910
  }
  adv_lat_strict(CT, Tm2, Tm1, Tp1, Tp2, Wm2, Wm1, Wp1, Wp2, yq.v, rk.adv_ccy, k, NY, dTy);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    da[i] = dTx[i] + dTy[i];                       // :913
    const v2 xd = T[4 + i] + dd[i];
    v2 x = xd + da[i];                             // :549
    if (calm_q) x.y = xd.y;                        // orig :562: vapour diffused, not advected
    xn.v[i] = x;
  }
  return xn;
}

// the arithmetic of one row in either mode
template <bool STRICT, bool SUB>
__device__ __forceinline__ q8 one_row(const q8& LT, const q8& CT, const q8& RT, const q8& Tm2, const q8& Tm1, const q8& Tp1,
                                      const q8& Tp2, const q8& LW, const q8& CW, const q8& RW, const q8& Wm2, const q8& Wm1,
                                      const q8& Wp1, const q8& Wp2, const f4& xq, const f4& yq, int k, int q, const RowK& rk,
                                      bool calm_q) {
  if (STRICT) return strict_row(LT, CT, RT, Tm2, Tm1, Tp1, Tp2, LW, CW, RW, Wm2, Wm1, Wp1, Wp2, xq, yq, k, q, rk, calm_q);
  return fast_row<SUB>(LT, CT, RT, Tm2, Tm1, Tp1, Tp2, LW, CW, RW, Wm2, Wm1, Wp1, Wp2, xq, yq, k, q == NQ - 1, rk.dif_cc, rk.dif_ccy,
                       calm_q);
}

// the row constants a task needs: everything in STRICT, two words in FAST (the advection constants are folded into
// the staged winds) -- the full 32-byte read costs the FAST loop 3 %
template <bool STRICT>
__device__ __forceinline__ RowK task_consts(const lfloat* lds, int k) {
  if (STRICT) return row_consts((const lfloat*)(lds + kOffRowK), k);
  RowK rk{};
  rk.dif_cc = lds[kOffRowK + k * kRowKWords];
  rk.dif_ccy = lds[kOffRowK + k * kRowKWords + 2];
  return rk;
}

// One bulk row of one quad column.  The five rows k-2 .. k+2 of the column come from ONE base address (the guard
// rows make k-2 = -1 and k+2 = NY valid reads: zero weights in FAST, not referenced by the reference's boundary
// formulas in STRICT), the row's left and right quads from the two precomputed offsets.
template <bool STRICT, bool SUB>
__device__ __forceinline__ void row_task(lfloat* lds, int cur, const TaskAddr& ta, bool calm_q = false) {
  const int k = ta.kq & 255, q = ta.kq >> 8;
  const lfloat* Xc = lds + kOffX + cur * XB;
  const lfloat* Wc = lds + kOffW;
  const RowK rk = row_consts((const lfloat*)(lds + kOffRowK), k); // (the two-word form of task_consts measured 3 % slower HERE)
  const f4 xq = ld4(lds + kOffWX + k * NX + 4 * q), yq = ld4(lds + kOffWY + k * NX + 4 * q);
  const lfloat* xb = Xc + ta.c - 2 * RS;
  const lfloat* wb = Wc + ta.c - 2 * RS;
  const q8 Tm2 = ld8p(xb), Tm1 = ld8p(xb + RS), CT = ld8p(xb + 2 * RS), Tp1 = ld8p(xb + 3 * RS), Tp2 = ld8p(xb + 4 * RS);
  const q8 Wm2 = ld8p(wb), Wm1 = ld8p(wb + RS), CW = ld8p(wb + 2 * RS), Wp1 = ld8p(wb + 3 * RS), Wp2 = ld8p(wb + 4 * RS);
  const q8 LT = ld8p(Xc + ta.l), RT = ld8p(Xc + ta.r), LW = ld8p(Wc + ta.l), RW = ld8p(Wc + ta.r);
  const q8 xn = one_row<STRICT, SUB>(LT, CT, RT, Tm2, Tm1, Tp1, Tp2, LW, CW, RW, Wm2, Wm1, Wp1, Wp2, xq, yq, k, q, rk, calm_q);
  st8p(lds + kOffX + (cur ^ 1) * XB + ta.c, xn); // own quad of the other buffer: same offset
}

// Two vertically adjacent bulk rows (k, k+1) of one quad column in one task: the six rows k-2 .. k+3 of the
// column are loaded once and shared -- 44 instead of 62 ds_read_b128 for the two rows.  The sub-step loop is
// co-limited by LDS bandwidth: dropping 39 % of the bulk reads (timing experiment) made it 12 % faster.
template <bool STRICT, bool SUB>
__device__ __forceinline__ void row_task2(lfloat* lds, int cur, const TaskAddr& ta, bool calm_q = false) {
  const int k = ta.kq & 255, q = ta.kq >> 8;
  const lfloat* Xc = lds + kOffX + cur * XB;
  const lfloat* Wc = lds + kOffW;
  const lfloat* xb = Xc + ta.c - 2 * RS;
  const lfloat* wb = Wc + ta.c - 2 * RS;
  const q8 T0 = ld8p(xb), T1 = ld8p(xb + RS), T2 = ld8p(xb + 2 * RS), T3 = ld8p(xb + 3 * RS), T4 = ld8p(xb + 4 * RS);
  const q8 W0 = ld8p(wb), W1 = ld8p(wb + RS), W2 = ld8p(wb + 2 * RS), W3 = ld8p(wb + 3 * RS), W4 = ld8p(wb + 4 * RS);
  lfloat* out = lds + kOffX + (cur ^ 1) * XB + ta.c;
  {
    const q8 LT = ld8p(Xc + ta.l), RT = ld8p(Xc + ta.r), LW = ld8p(Wc + ta.l), RW = ld8p(Wc + ta.r);
    const f4 xq = ld4(lds + kOffWX + k * NX + 4 * q), yq = ld4(lds + kOffWY + k * NX + 4 * q);
    const RowK rk = task_consts<STRICT>(lds, k);
    st8p(out, one_row<STRICT, SUB>(LT, T2, RT, T0, T1, T3, T4, LW, W2, RW, W0, W1, W3, W4, xq, yq, k, q, rk, calm_q));
  }
  __builtin_amdgcn_sched_barrier(0); // row k+1 after row k: keeps the two rows' temporaries from piling up
  {
    const q8 T5 = ld8p(xb + 5 * RS), W5 = ld8p(wb + 5 * RS); // row k+3: only the second row needs it
    const q8 LT = ld8p(Xc + ta.l + RS), RT = ld8p(Xc + ta.r + RS), LW = ld8p(Wc + ta.l + RS), RW = ld8p(Wc + ta.r + RS);
    const f4 xq = ld4(lds + kOffWX + (k + 1) * NX + 4 * q), yq = ld4(lds + kOffWY + (k + 1) * NX + 4 * q);
    const RowK rk = task_consts<STRICT>(lds, k + 1);
    st8p(out + RS, one_row<STRICT, SUB>(LT, T3, RT, T1, T2, T4, T5, LW, W3, RW, W1, W2, W4, W5, xq, yq, k + 1, q, rk, calm_q));
  }
}

struct BulkTasks { TaskAddr t[3]; };

// The schedule (both arithmetic modes).  Task kinds: one row (S1 sub-cycled family, F1 full family) or two stacked rows (ST, FT).
//   ST  rows (1,2) (3,4) (5,6) (7,8) (39,40) .. (45,46) : 8 pairs x 24 quads = 3 full passes   (426 VALU instructions)
//   FT  rows (10,11) .. (24,25)                          : 8 pairs x 24 quads = 3 full passes   (306)
//   S1  rows 9, 38                                       : 48 tasks, one pass                   (223)
//   F1  rows 26 .. 37                                    : 288 tasks, 4.5 passes                (156)
// Waves w and w+4 share a SIMD; the polar rows (a dependent chain of 8 + 1 Jacobi sweeps per sub-step) have their own
// wave(s): wave 6 in FAST, waves 2 and 3 in STRICT.  The SIMD's issue arbiter favours the OLDER wave; the younger one runs in the
// slots that leaves, and once the older wave is done the younger runs alone at a single wave's issue rate (one
// instruction per ~5 cycles at best, ~6-7 with its LDS waits) -- so the older wave of a pair carries the larger
// share and the two should finish together.  In-kernel stamps (tools/stamp_member.py) give each wave's busy time
// per sub-step.  The loop is VALU-pipe bound: a packed fp32 instruction occupies the SIMD for ~4.9 cycles (measured,
// tools/ubench/valu_rate.hip; a scalar one ~2.3-2.7).
enum { kNone = 0, kS1 = 1, kF1 = 2, kST = 3, kFT = 4 };
struct Pass { int kind, index; };
// FAST: ONE wave runs all four polar chains (quad_chain_substep: 36-instruction sweeps, ~2 600 busy cycles per
// sub-step when it has the SIMD to itself, where two waves took ~4 400 each) and there are seven bulk waves.  The chain
// wave issues one instruction every ~5 cycles for as long as it runs; as the OLDER wave of its SIMD it leaves its
// partner little of the pipe until it is done, as the YOUNGER one (wave 6) it fills the slots its partner leaves and
// both finish together.  Measured sub-step times (tools/deal_search.py: in-kernel stamps, 512 members, one gpurun call
// per group; slots w0 w1 w2 w3 | w4 w5 w6 w7, C = the chains, S = ST, T = FT, H = S1, F = F1):
//   chains on wave 6 (younger wave of SIMD 2)
//     S0+F0 S2+F1 T1+F3 T2+H+F2 | S1    T0    C     F4      4 773   <- the table below
//     S0+F0 S1+H  T1+F3 T2+F1+F2| S2    T0    C     F4      4 814
//     S0+F0 S2+H  T1+F3 T2+F1+F2| S1    T0    C     F4      4 817
//     S0+F0 S2+H  T1+F3 T2+F1   | S1    T0    C     F2+F4   4 886
//     S0+F0 S2+H  T1+F3+F4 T2+F1+F2 | S1 T0   C     -       5 315   (the chains starve behind three passes)
//     S0+F0 S2+H  T1+F3 T2+F1+F2| S1    T0+F4 C     -       5 665
//   chains on wave 2 (older wave of SIMD 2), same call: 4 900
//     S0+F0 S2+H  C     T2+F1+F2| S1    T0    T1+F3 F4      4 900 ... 4 932
//     S0+F0 S2+H  C     T2+F1   | S1    T0    T1+F3 F2+F4   4 975
//     S0+F0 S2+F1 C     T2+H+F2 | S1    T0    T1+F3 F4      4 974
//     S0+F0 S2+H  C     T2+F1+F2| S1    T0    T1    F3+F4   5 070   (wave 7 waits behind wave 3's three passes)
//     S0+F0 S2+T0 C     H+F1+F2 | S1    T2    T1    F3+F4   5 094
//     S0    S2+F0 C     H+F1+F2 | S1    T0    T1+T2 F3+F4   5 988   (two FT passes behind the chain wave)
//   chains on wave 7 (younger wave of SIMD 3): 5 432
// With the table below the busiest wave of each SIMD is busy 4 325 / 4 348 / 4 379 / 4 368 cycles: balanced to 1 %.
__host__ __device__ constexpr Pass deal_fast(int wave, int i) {
  constexpr Pass none{kNone, 0};
#if defined(GREB_TUNING) && defined(GREB_DEAL_FAST) // tools/deal_search.py: a deal given on the compiler command line
  constexpr Pass t[8][3] = {GREB_DEAL_FAST};
#else
  constexpr Pass t[8][3] = {
      /* w0 */ {{kST, 0}, {kF1, 0}, none},     /* w1 */ {{kST, 2}, {kF1, 1}, none},
      /* w2 */ {{kFT, 1}, {kF1, 3}, none},     /* w3 */ {{kFT, 2}, {kS1, 0}, {kF1, 2}},
      /* w4 */ {{kST, 1}, none, none},         /* w5 */ {{kFT, 0}, none, none},
      /* w6 */ {none, none, none},             /* w7 */ {{kF1, 4}, none, none}};
#endif
  return t[wave][i];
}
// STRICT: waves 2 and 3 are the polar waves (chain_substep, one pole each)
__host__ __device__ constexpr Pass deal_strict(int wave, int i) {
  constexpr Pass none{kNone, 0};
  // Measured sub-step times of the deals tried (in-kernel stamps, 512 members, cycles at 2.36 GHz; the first line is
  // round 1's deal):   w0 ST0 FT0 | w4 ST1 ; w1 ST2 F1_0 | w5 S1 F1_1 F1_2 ; w6 FT1 F1_3 ; w7 FT2 F1_4   6 299
  //                    one F1 pass moved from the younger to the older wave of SIMD 1                          5 838
  //                    ... and the other F1 pass of wave 5 moved to wave 6 (pole SIMD)                         6 115
  //                    w0 ST0 F1 F1 | w4 ST1 ; w1 ST2 FT0 | w5 S1 F1 ; w6 FT1 F1 ; w7 FT2 F1                   6 059
  //                    an ST pass on a pole SIMD (w6 ST2), FT1 F1 on w4                                        6 210
  // With the clamp fast path and the polar halos by ds_bpermute (polar waves 4 700 -> 4 100 busy cycles) the second
  // line takes 5 501; the table below -- its S1 pass moved to wave 6, wave 5 left with two F1 passes -- 5 416.
  // Static s_setprio 1 on the younger waves, on waves 6/7 or on the polar waves: 6 847 / 6 661 / 5 855 -- the
  // prioritised wave runs its instructions at ~6 cycles each and its partner then runs alone; age order is best.
  constexpr Pass t[8][3] = {
      /* w0 */ {{kST, 0}, {kFT, 0}, none},     /* w1 */ {{kST, 2}, {kF1, 0}, {kF1, 1}},
      /* w2 */ {none, none, none},             /* w3 */ {none, none, none},
      /* w4 */ {{kST, 1}, none, none},         /* w5 */ {{kF1, 2}, {kF1, 3}, none},
      /* w6 */ {{kFT, 1}, {kS1, 0}, none},     /* w7 */ {{kFT, 2}, {kF1, 4}, none}};
  return t[wave][i];
}
template <bool STRICT>
__host__ __device__ constexpr Pass deal(int wave, int i) { return STRICT ? deal_strict(wave, i) : deal_fast(wave, i); }
#if defined(GREB_TUNING) && defined(GREB_POLAR_WAVE_FAST) // tools/deal_search.py: which wave runs the FAST polar chains
constexpr int kPolarWaveFast = GREB_POLAR_WAVE_FAST;
#else
constexpr int kPolarWaveFast = 6;
#endif
template <bool STRICT>
__host__ __device__ constexpr bool is_polar_wave(int wave) { return STRICT ? (wave == 2 || wave == 3) : wave == kPolarWaveFast; }

// every pass of every kind is dealt to exactly one wave slot (a deal that drops or doubles a pass would still "run")
template <bool STRICT>
__host__ __device__ constexpr int times_dealt(int kind, int index) {
  int n = 0;
  for (int w = 0; w < 8; ++w)
    for (int i = 0; i < 3; ++i) n += (deal<STRICT>(w, i).kind == kind && deal<STRICT>(w, i).index == index) ? 1 : 0;
  return n;
}
template <bool STRICT>
__host__ __device__ constexpr bool deal_is_complete() {
  for (int p = 0; p < 3; ++p) if (times_dealt<STRICT>(kST, p) != 1 || times_dealt<STRICT>(kFT, p) != 1) return false;
  for (int p = 0; p < 5; ++p) if (times_dealt<STRICT>(kF1, p) != 1) return false;
  if (times_dealt<STRICT>(kS1, 0) != 1) return false;
  for (int w = 0; w < 8; ++w)
    for (int i = 0; i < 3; ++i) if (is_polar_wave<STRICT>(w) && deal<STRICT>(w, i).kind != kNone) return false;
  return true;
}
static_assert(deal_is_complete<true>() && deal_is_complete<false>(),
              "a deal must cover 3 ST, 3 FT, 1 S1 and 5 F1 passes exactly once");

// number of tasks of a pass: 64 = every lane has one
__host__ __device__ constexpr int pass_tasks(int kind, int index) {
  const int total = kind == kST || kind == kFT ? 8 * NQ : (kind == kS1 ? 2 * NQ : (kind == kF1 ? 12 * NQ : 0));
  const int left = total - 64 * index;
  return left >= 64 ? 64 : (left > 0 ? left : 0);
}

// once per launch; the asm makes the values opaque, so the compiler keeps them instead of re-deriving them from
// the lane id inside the sub-step loop
template <bool STRICT>
__device__ __forceinline__ BulkTasks make_tasks(int wave, int lane) {
  BulkTasks b;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int k = 1, q = 0, valid = 0;
    {
      const int kind = deal<STRICT>(wave, i).kind, t = deal<STRICT>(wave, i).index * 64 + lane;
      const int r = t / NQ;
      q = t % NQ;
      // row order: consecutive r of a pass sit two rows (= 8 slot classes) apart wherever the rows allow it
      if (kind == kST) { valid = r < 8; k = r < 4 ? 1 + 2 * r : (r == 4 ? 41 : (r == 5 ? 39 : (r == 6 ? 45 : 43))); }
      else if (kind == kFT) { valid = r < 8; k = 10 + 2 * r; }
      else if (kind == kS1) { valid = r < 2; k = r == 0 ? 9 : 38; if (r == 1) q = (q + 4) % NQ; }
      else if (kind == kF1) { valid = r < 12; k = r < 6 ? 26 + 2 * r : 27 + 2 * (r - 6); }
      if (!valid) { k = 1; q = 0; }
    }
    TaskAddr a;
    a.c = k * RS + 4 * q;
    a.l = k * RS + 4 * (q == 0 ? NQ - 1 : q - 1);
    a.r = k * RS + 4 * (q == NQ - 1 ? 0 : q + 1);
    a.kq = valid ? (k | (q << 8)) : -1;
    asm volatile("" : "+v"(a.c), "+v"(a.l), "+v"(a.r), "+v"(a.kq));
    b.t[i] = a;
  }
  return b;
}

// task i of bulk wave SLOT: kind and pass are compile-time, so a wave's sub-step is straight-line code -- no
// per-sub-step dispatch on the wave number (that dispatch, a switch on a VGPR lowered to exec-mask bookkeeping, cost
// every bulk wave ~100 issue slots per sub-step)
template <bool STRICT, int SLOT, int I>
__device__ __forceinline__ void bulk_task(lfloat* lds, int cur, const BulkTasks& tasks, int dbg, bool calm_q) {
  constexpr int kind = deal<STRICT>(SLOT, I).kind, ntask = pass_tasks(kind, deal<STRICT>(SLOT, I).index);
  if constexpr (kind != kNone && ntask > 0) {
    if (ntask < 64 && tasks.t[I].kq < 0) return; // partial pass: lanes without a task
    if constexpr (kind == kS1) { if (!(dbg & 1)) row_task<STRICT, true>(lds, cur, tasks.t[I], calm_q); }
    else if constexpr (kind == kF1) { if (!(dbg & 2)) row_task<STRICT, false>(lds, cur, tasks.t[I], calm_q); }
    else if constexpr (kind == kST) { if (!(dbg & 1)) row_task2<STRICT, true>(lds, cur, tasks.t[I], calm_q); }
    else { if (!(dbg & 2)) row_task2<STRICT, false>(lds, cur, tasks.t[I], calm_q); }
  }
}
template <bool STRICT, int SLOT>
__device__ __forceinline__ void bulk_substep(lfloat* lds, int cur, const BulkTasks& tasks, int dbg, bool calm_q = false) {
  bulk_task<STRICT, SLOT, 0>(lds, cur, tasks, dbg, calm_q);
  bulk_task<STRICT, SLOT, 1>(lds, cur, tasks, dbg, calm_q);
  bulk_task<STRICT, SLOT, 2>(lds, cur, tasks, dbg, calm_q);
}

// ---------------------------------------------------------------------------------------------
// polar rows, STRICT: one wave per pole, 48 lanes x 2 longitudes x (Tair,q), the reference's expression trees
// ---------------------------------------------------------------------------------------------
// float offset, inside a [half][quad][4] row, of the longitude pair (2l, 2l+1)
__device__ __forceinline__ int pair_off(int l) { return (l & 1) * kHalfRow + (l >> 1) * 4; }
__device__ __forceinline__ void load_win10(const lfloat* row, int l, v2 t[10]) {
  // lanes l-2 .. l+2 (mod 48) -> longitudes 2l-4 .. 2l+5
  const int lm2 = l >= 2 ? l - 2 : l + 46, lm1 = l >= 1 ? l - 1 : 47, lp1 = l <= 46 ? l + 1 : 0, lp2 = l <= 45 ? l + 2 : l - 46;
  const int idx[5] = {lm2, lm1, l, lp1, lp2};
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const vfloat4 a = *(const __attribute__((address_space(3))) vfloat4*)(row + pair_off(idx[i]));
    t[2 * i] = v2{a.x, a.y}; t[2 * i + 1] = v2{a.z, a.w};
  }
}
__device__ __forceinline__ void st_pair2(lfloat* p, v2 a, v2 b) {
  vfloat4 x; x.x = a.x; x.y = a.y; x.z = b.x; x.w = b.y;
  *(__attribute__((address_space(3))) vfloat4*)p = x;
}

__device__ __forceinline__ void chain_substep_strict(lfloat* lds, int cur, int pole, int l /* lane */, bool calm_q = false) {
  if (l >= 48) return; // idle lanes (no workgroup barrier inside this function)
  const int k = pole ? NY - 1 : 0;
  const RowK rk = row_consts((const lfloat*)(lds + kOffRowK), k);
  const lfloat* Xc = lds + kOffX + cur * XB;
  const lfloat* Wc = lds + kOffW;
  v2 T0w[10], w[10];
  load_win10(Xc + k * RS, l, T0w);
  load_win10(Wc + k * RS, l, w);
  const float u0 = lds[kOffWX + k * NX + 2 * l], u1 = lds[kOffWX + k * NX + 2 * l + 1]; // raw winds
  const float v0 = lds[kOffWY + k * NX + 2 * l], v1 = lds[kOffWY + k * NX + 2 * l + 1];
  const int k1 = pole ? k - 1 : k + 1, k2 = pole ? k - 2 : k + 2; // towards the interior
  v2 T1[2], T2[2], W1[2], W2[2];
  {
    const vfloat4 a = *(const __attribute__((address_space(3))) vfloat4*)(Xc + k1 * RS + pair_off(l));
    const vfloat4 b = *(const __attribute__((address_space(3))) vfloat4*)(Xc + k2 * RS + pair_off(l));
    const vfloat4 c = *(const __attribute__((address_space(3))) vfloat4*)(Wc + k1 * RS + pair_off(l));
    const vfloat4 d = *(const __attribute__((address_space(3))) vfloat4*)(Wc + k2 * RS + pair_off(l));
    T1[0] = v2{a.x, a.y}; T1[1] = v2{a.z, a.w}; T2[0] = v2{b.x, b.y}; T2[1] = v2{b.z, b.w};
    W1[0] = v2{c.x, c.y}; W1[1] = v2{c.z, c.w}; W2[0] = v2{d.x, d.y}; W2[1] = v2{d.z, d.w};
  }
  const v2 own[2] = {T0w[4], T0w[5]};
  const bool bug_lane = (l == 46); // its 2nd point is longitude xdim-2 (1-based), :881
  // Between two Jacobi sweeps a lane needs the new values of lanes l-2 .. l+2 (mod 48).  They travel by
  // ds_bpermute_b32 -- lane to lane through the LDS crossbar, no memory, no bank conflicts, ONE hop -- instead of a
  // ds_write + s_waitcnt + ds_read round trip through a row buffer behind the bulk waves' read traffic.  The chain
  // of 8 dependent sweeps is the floor of a sub-step: 4 700 busy cycles per sub-step with the row buffers, 4 100 so.
  const int am2 = 4 * (l >= 2 ? l - 2 : l + 46), am1 = 4 * (l >= 1 ? l - 1 : 47), ap1 = 4 * (l <= 46 ? l + 1 : 0),
            ap2 = 4 * (l <= 45 ? l + 2 : l - 46);
  auto from = [](int addr, v2 x) {
    return v2{__int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x.x))),
              __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(x.y)))};
  };
  const float uu[2] = {u0, u1};

  v2 Th[2][2]; // [diffusion, advection][point]
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    // the row constants come from LDS (per lane) but are the same in every lane: a scalar trip count keeps the
    // sweep loop free of exec-mask bookkeeping
    const int time2 = __builtin_amdgcn_readfirstlane(which ? rk.adv_time2 : rk.dif_time2);
    const float cc = which ? rk.adv_cc : rk.dif_cc;
    v2 T[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) T[i] = T0w[i];
    v2 th[2] = {own[0], own[1]};
    for (int tt = 0; tt < time2; ++tt) {
      if (tt > 0) { // T[0] and T[9] are not referenced by either stencil
        T[1] = from(am2, th[1]); T[2] = from(am1, th[0]); T[3] = from(am1, th[1]);
        T[4] = th[0]; T[5] = th[1];
        T[6] = from(ap1, th[0]); T[7] = from(ap1, th[1]); T[8] = from(ap2, th[0]);
      }
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
#pragma clang fp contract(off)
        const int c = 4 + pt;
        v2 dd;
        if (which) dd = adv_lon_sub_point_strict(T, w, uu[pt], cc, c, bug_lane && pt == 1);
        else dd = div20(cc * dif_S_strict(T, w, c));
        dd = clamp_e(dd, T[c]); // :715 / :907
        th[pt] = T[c] + dd;
      }
    }
    Th[which][0] = th[0]; Th[which][1] = th[1];
  }

  // ---- latitudinal terms + update (:585-590, :756-795, :721, :913, :549)
  v2 xn[2];
  const float vv[2] = {v0, v1};
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
#pragma clang fp contract(off)
    const v2 t0 = own[pt], w0 = w[4 + pt], a1 = T1[pt], a2 = T2[pt], b1 = W1[pt], b2 = W2[pt];
    const float vm = split_m(vv[pt]), vp = split_p(vv[pt]);
    v2 dTy, aTy;
    if (pole == 0) {
      dTy = rk.dif_ccy * b1 * (-t0 + a1);                                        // :589
      aTy = div3(rk.adv_ccy * (vp * (b1 * (t0 - a1) + b2 * (t0 - a2))));         // :759-761
    } else {
      dTy = rk.dif_ccy * b1 * (a1 - t0);                                         // :590
      aTy = div3(rk.adv_ccy * (-vm * (b1 * (t0 - a1) + b2 * (t0 - a2))));        // :792-794
    }
    const v2 dd = w0 * ((Th[0][pt] - t0) + dTy); // :718, :721
    const v2 da = (Th[1][pt] - t0) + aTy;        // :910, :913
    const v2 xd = t0 + dd;
    v2 x = xd + da;                              // :549
    if (calm_q) x.y = xd.y;                      // orig :562
    xn[pt] = x;
  }
  st_pair2(lds + kOffX + (cur ^ 1) * XB + k * RS + pair_off(l), xn[0], xn[1]);
}

// ---------------------------------------------------------------------------------------------
// FAST polar rows: all four chains -- (pole, tracer) -- in ONE wave
// ---------------------------------------------------------------------------------------------
// A DPP row of 16 lanes x 6 points is one 96-point latitude circle whose periodic boundary is the row rotate
// (row_ror:1 / row_ror:15), so the four rows of a wavefront hold the four polar chains of a member: lanes 0-15
// (south pole, Tair), 16-31 (south, q), 32-47 (north, Tair), 48-63 (north, q).  The sweeps are greb_chain6.h's
// (36 instructions for all four chains; the earlier form -- one wave per pole, 48 lanes x 2 points x (Tair,q), halos
// by 12 ds_bpermute per sweep -- needed ~45 per pole and waited for the LDS crossbar in every one of the 9 sweeps:
// 4 400 busy cycles per sub-step on two waves, now ~2 100 on one).  The coefficients of the diffusion chain are fixed
// for the run and those of the advection sweep for the model step; both are rebuilt per circulation call so that
// nothing of this lives across the point-physics phase.
struct QuadChain {
  ChainK kd, ka;             // d[i] = sum_m K[i][m] e[i+m] (greb_chain6.h): zonal diffusion, zonal advection
  float w0[6], W1[6], W2[6]; // weights of the own row and of the two rows towards the interior
  float own[6];              // the lane's points: loaded once per circulation call, then carried from sub-step to sub-step
  float cv[6];               // the latitudinal advection coefficient of the point: (ccy/3) min(v,0) south, -(ccy/3) max(v,0) north
  float ccy;                 // dif_ccy
  int at[3];                 // float offsets, within a row, of the lane's three longitude pairs (+ its tracer)
  int row[3];                // float offsets, within a guarded buffer, of rows k, k1, k2
  int time2_dif, time2_adv;
  bool calm;                 // vapour lane of an experiment that diffuses vapour without advecting it
};

__device__ __forceinline__ void quad_chain_setup(const lfloat* lds, int cur, int lane, bool calm_q, QuadChain& c) {
  const int r = lane >> 4, j = lane & 15, pole = r >> 1, C = r & 1;
  const int k = pole ? NY - 1 : 0, k1 = pole ? k - 1 : k + 1, k2 = pole ? k - 2 : k + 2; // towards the interior
  const RowK rk = row_consts((const lfloat*)(lds + kOffRowK), k);
  c.time2_dif = rk.dif_time2; c.time2_adv = rk.adv_time2; c.ccy = rk.dif_ccy;
  c.calm = calm_q && C == 1;
  c.row[0] = k * RS; c.row[1] = k1 * RS; c.row[2] = k2 * RS;
#pragma unroll
  for (int t = 0; t < 3; ++t) c.at[t] = pair_off(3 * j + t) + C;
  const lfloat* Wc = lds + kOffW;
  // (static_for: compile-time indices from the start -- arrays indexed by the variable of a `#pragma unroll` loop are
  // still dynamically indexed when the compiler decides what may live in registers)
  float w[12]; // longitudes 6j-3 .. 6j+8
  static_for<12>([&](auto I) {
    constexpr int i = I;
    int x = 6 * j - 3 + i;
    x = x < 0 ? x + NX : (x >= NX ? x - NX : x);
    w[i] = Wc[c.row[0] + pair_off(x >> 1) + (x & 1) * 2 + C];
  });
  const float third = rk.adv_ccy * (1.f / 3.f), csd = rk.dif_cc * 0.05f, csa = rk.adv_cc * 0.05f;
  float Kd[6][6], Ka[6][6];
  static_for<6>([&](auto I) {
    constexpr int i = I, p = 3 + i;
    const int o = c.at[i >> 1] + (i & 1) * 2;
    c.w0[i] = w[p];
    c.own[i] = lds[kOffX + cur * XB + c.row[0] + o];
    c.W1[i] = Wc[c.row[1] + o];
    c.W2[i] = Wc[c.row[2] + o];
    const float u = lds[kOffWX + k * NX + 6 * j + i], v = lds[kOffWY + k * NX + 6 * j + i]; // raw winds in the polar rows
    // south pole: only the v<0 part couples (rows 1,2); north pole: only the v>=0 part (rows 46,45)  (:759-761, :792-794)
    c.cv[i] = pole == 0 ? third * fminf(v, 0.f) : -third * fmaxf(v, 0.f);
    // 6(Pp[c] - Pm[c-1]) + 3(Pp[c+1] - Pm[c-2]) + (Pp[c+2] - Pm[c-3]), :595-600 in edge-flux form
    Kd[i][0] = -csd * w[p - 3]; Kd[i][1] = (-3.f * csd) * w[p - 2]; Kd[i][2] = (-6.f * csd) * w[p - 1];
    Kd[i][3] = (6.f * csd) * w[p + 1]; Kd[i][4] = (3.f * csd) * w[p + 2]; Kd[i][5] = csd * w[p + 3];
    // -up (10 Pp[c] + 4 Pp[c+1] + Pp[c+2]) - um (10 Pm[c-1] + 4 Pm[c-2] + Pm[c-3]), :845-851
    const float um = csa * fmaxf(u, 0.f), up = csa * fminf(u, 0.f);
    Ka[i][0] = -um * w[p - 3]; Ka[i][1] = (-4.f * um) * w[p - 2]; Ka[i][2] = (-10.f * um) * w[p - 1];
    Ka[i][3] = (-10.f * up) * w[p + 1]; Ka[i][4] = (-4.f * up) * w[p + 2]; Ka[i][5] = -up * w[p + 3];
    if (i == 3 && j == 15) { // longitude xdim-2 (1-based), :881: the 4* term vanishes, the 1* term is w(1)*(T(xdim-1)-T(1))
      Ka[i][4] = -up * w[p + 3]; Ka[i][5] = -up * w[p + 3];
    }
  });
  c.kd = chain_pack(Kd);
  c.ka = chain_pack(Ka);
}

__device__ __forceinline__ void quad_chain_substep(lfloat* lds, int cur, QuadChain& c) {
  const lfloat* Xc = lds + kOffX + cur * XB;
  // (the own row is this wave's alone: what it stored last sub-step is what it would read now, so the chain starts
  // without an LDS round trip; the neighbour rows are requested here and not needed before the epilogue)
  float own[6], T1[6], T2[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int o = c.at[i >> 1] + (i & 1) * 2;
    own[i] = c.own[i];
    T1[i] = Xc[c.row[1] + o];
    T2[i] = Xc[c.row[2] + o];
  }
  float Td[6], Ta[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { Td[i] = own[i]; Ta[i] = own[i]; }
  chain_run6<true>(Td, c.kd, c.time2_dif); // :656-717
  chain_run6<true>(Ta, c.ka, c.time2_adv); // :861-909 (one sweep at this grid)
  // ---- latitudinal terms + update (:585-590, :756-795, :721, :913, :549)
  lfloat* out = lds + kOffX + (cur ^ 1) * XB + c.row[0];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float g1 = c.W1[i] * (T1[i] - own[i]);
    const float d2 = c.W2[i] * (own[i] - T2[i]);
    const float ddy = c.ccy * g1;
    const float day = c.cv[i] * (d2 - g1);
    const float dd = c.w0[i] * ((Td[i] - own[i]) + ddy);
    float da = (Ta[i] - own[i]) + day;
    if (c.calm) da = 0.f; // orig :562
    c.own[i] = (own[i] + dd) + da;
    out[c.at[i >> 1] + (i & 1) * 2] = c.own[i];
  }
}

// stage this step's winds (src/greb.f90:203-216, 732): raw for STRICT and for the polar rows,
// otherwise scaled by the row's advection constants so the sign split is one max/min per use:
//   x = c*u, c = ccx/3 (full rows) or ccx2/20 (sub-cycled rows);  y = ccy/3 * v
template <bool STRICT>
__device__ __forceinline__ void stage_winds(lfloat* lds, const float* __restrict__ u, const float* __restrict__ v,
                                            const RowTables* __restrict__ tab) {
  // All of a thread's wind quads are requested before the first is waited for (unconditional loads at clamped
  // indices, predicated stores): as a plain load-scale-store loop the three iterations were three dependent round
  // trips, 2 700 cycles per model step.  The row's constants come from the LDS copy staged at init.
  constexpr int kIt = (NP / 4 + kThreads - 1) / kThreads;
  f4 uq[kIt], vq[kIt];
#pragma unroll
  for (int j = 0; j < kIt; ++j) {
    const int i = min((int)threadIdx.x + j * kThreads, NP / 4 - 1);
    uq[j] = ld4(u + 4 * i); vq[j] = ld4(v + 4 * i);
  }
#pragma unroll
  for (int j = 0; j < kIt; ++j) {
    const int i = threadIdx.x + j * kThreads;
    if (i < NP / 4) {
      const int k = i / NQ;
      if (!(STRICT || k == 0 || k == NY - 1)) {
        const RowK rk = row_consts((const lfloat*)(lds + kOffRowK), k);
        const float cu = rk.sub ? rk.adv_cc * 0.05f : rk.adv_cc * (1.f / 3.f);
        const float cv = rk.adv_ccy * (1.f / 3.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) { uq[j].v[e] *= cu; vq[j].v[e] *= cv; }
      }
      st4(lds + kOffWX + 4 * i, uq[j]); st4(lds + kOffWY + 4 * i, vq[j]);
    }
  }
  (void)tab;
}

// the circulation loop shared by the member kernel and its test mirror
template <bool STRICT>
struct Circ {
  __device__ __forceinline__ void init(lfloat* lds, const float* wz_air, const float* wz_vapor,
                                       const RowTables* __restrict__ tab) {
    // guard rows of X[0], X[1], W: zero, never written again
    for (int i = threadIdx.x; i < 6 * RS; i += kThreads) {
      const int g = i / RS, o = i % RS; // buffer g >> 1, lower / upper guard g & 1
      lds[(g >> 1) * XB + (g & 1) * (NY + 1) * RS + o] = 0.f;
    }
    stage_row_consts(lds + kOffRowK, *tab, 0, NY);
    for (int i = threadIdx.x; i < NP / 4; i += kThreads)
      st8(lds + kOffW + (i / NQ) * RS, i % NQ, zip(ld4(wz_air + 4 * i), ld4(wz_vapor + 4 * i)));
  }

  // One wave's share of `nsub` sub-steps, one s_barrier after each (every wave of the workgroup executes the same
  // number of barriers, each in its own loop).  WAVE is compile-time: the loop body is the wave's own straight-line
  // task sequence.  dbg: timing experiments only (tools/microbench_circ.py): bit0/1/2 skip sub / full / chain work.
  template <int WAVE>
  __device__ __forceinline__ void role_loop(lfloat* lds, int cur, int nsub, int lane, int dbg, bool calm_q
#ifdef GREB_TUNING
                                            , bool stamp, unsigned long long& busy
#endif
  ) {
    // the lane's task addresses: derived once per circulation call (not per launch -- values that live across the
    // point-physics phase, where the register pressure peaks, come back as scratch reloads inside this loop)
    constexpr bool kPolar = is_polar_wave<STRICT>(WAVE);
    BulkTasks tasks;
    QuadChain quad;
    if constexpr (!kPolar) tasks = make_tasks<STRICT>(WAVE, lane);
    else if constexpr (!STRICT) quad_chain_setup(lds, cur, lane, calm_q, quad);
#pragma unroll 1
    for (int tt = 0; tt < nsub; ++tt) {
#ifdef GREB_TUNING
      const unsigned long long t0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
#endif
      if constexpr (kPolar && STRICT) { if (!(dbg & 4)) chain_substep_strict(lds, cur, WAVE - 2, lane, calm_q); }
      else if constexpr (kPolar) { if (!(dbg & 4)) quad_chain_substep(lds, cur, quad); }
      else bulk_substep<STRICT, WAVE>(lds, cur, tasks, dbg, calm_q);
#ifdef GREB_TUNING
      if (stamp) busy += __builtin_amdgcn_s_memtime() - t0;
#endif
      __syncthreads();
      cur ^= 1;
    }
  }
  // `nsub` sub-steps starting from buffer `cur`; the caller continues with buffer cur ^ (nsub & 1)
  __device__ __forceinline__ void substeps(lfloat* lds, int cur, int nsub, int dbg = 0, bool calm_q = false
#ifdef GREB_TUNING
                                           , bool stamp = false, unsigned long long* busy = nullptr
#endif
  ) {
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane)); // opaque: nothing derived from it is hoisted out of the model-step loop
#ifdef GREB_TUNING
    unsigned long long b = 0;
#define GREB_ROLE(W) case W: role_loop<W>(lds, cur, nsub, lane, dbg, calm_q, stamp, b); break;
#else
#define GREB_ROLE(W) case W: role_loop<W>(lds, cur, nsub, lane, dbg, calm_q); break;
#endif
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) { // a scalar branch per circulation call
      GREB_ROLE(0) GREB_ROLE(1) GREB_ROLE(2) GREB_ROLE(3) GREB_ROLE(4) GREB_ROLE(5) GREB_ROLE(6) GREB_ROLE(7)
    }
#undef GREB_ROLE
#ifdef GREB_TUNING
    if (busy) *busy += b;
#endif
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
    st8(lds + kOffX + (i / NQ) * RS, i % NQ, zip(x, x));
  }
  __syncthreads(); // stage_winds reads the row constants init() staged
  stage_winds<STRICT>(lds, ug + fo, vg + fo, tab);
  __syncthreads();
  c.substeps(lds, 0, nsub, dbg);
  const int cur = nsub & 1;
  for (int i = threadIdx.x; i < NP / 4; i += kThreads) {
    const f4 a = comp(ld8(lds + kOffX + cur * XB + (i / NQ) * RS, i % NQ), 0), b = ld4(Xin + fo + 4 * i);
    st4(dX + fo + 4 * i, f4{{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]}}); // :551
  }
}

hipError_t launch_circulation_g96(const float* X, const float* wz, const float* u, const float* v, float* dX,
                                  const RowTables* tab_dev, int batch, int nsub, bool strict, hipStream_t s) {
  auto kern = strict ? circulation_g96_kernel<true> : circulation_g96_kernel<false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(batch), dim3(kThreads), kLdsBytes, s, X, wz, u, v, dX, tab_dev, nsub,
                     tuning_int("GREB_DEBUG_SKIP", 0)); // -DGREB_TUNING builds only
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// the member kernel
// ---------------------------------------------------------------------------------------------
// EXP: honour the sensitivity-experiment switches a.xsw; the default instantiation has none of it
template <bool STRICT, bool FLUX, bool EXP>
__global__ __launch_bounds__(kThreads) void member_kernel(MemberArgs a) {
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  const int m = blockIdx.x, tid = threadIdx.x;
  float* state = a.state + (size_t)m * 5 * NP;
  float* acc = a.acc + (size_t)m * 6 * NP;
  float* corr = a.corr + (size_t)a.corr_index[m] * 3 * kNT * NP;
  const RowTables* tab = a.tabs + a.tab_index[m];

#ifdef GREB_TUNING
  // timing experiment (GREB_DEBUG_PHYS >> 8 = units of 1024 cycles): workgroup b starts (b mod 8) units late, so that
  // the point-physics phases of the CUs -- the only memory traffic of a step -- do not coincide
  for (int i = 0; i < (a.dbg >> 8) * (m & 7); ++i) __builtin_amdgcn_s_sleep(16);
#endif
  Circ<STRICT> circ;
  circ.init(lds, a.wz_air, a.wz_vapor, tab);
  for (int i = tid; i < NP / 4; i += kThreads)
    st8(lds + kOffX + (i / NQ) * RS, i % NQ, zip(ld4(state + NP + 4 * i), ld4(state + 3 * NP + 4 * i))); // (Tair, q)
  int cur = 0;
  __syncthreads();
#ifdef GREB_TUNING
  // diagnostic stamps (tools/stamp_member.py): shader-clock cycles this wave spent in each phase of the launch, its
  // busy time inside the sub-steps (barrier release -> arrival at the next barrier) and the launch's wall time in
  // s_memrealtime ticks (100 MHz) -- in-kernel clock = cycles / ticks x 100 MHz
  const bool stamp = a.stamps != nullptr;
  unsigned long long st_wind = 0, st_circ = 0, st_busy = 0, st_phys = 0;
  const unsigned long long st_c0 = stamp ? __builtin_amdgcn_s_memtime() : 0, st_r0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
#define GREB_STAMP(var) const unsigned long long var = stamp ? __builtin_amdgcn_s_memtime() : 0
#else
#define GREB_STAMP(var)
#endif

#pragma unroll 1
  for (int s = 0; s < a.nsteps; ++s) {
    const long long it = a.it0 + s;
    const StepClock ck = step_clock<FLUX>(a, it, NP);
    const int ityr = ck.ityr, yr_rel = ck.yr_rel;
    const size_t off = ck.off;

    GREB_STAMP(t_a);
    stage_winds<STRICT>(lds, a.uclim + off, a.vclim + off, tab);
    __syncthreads();
    GREB_STAMP(t_b);

    // ---- circulation of Tair and q: 24 sub-steps (:543-550)
#ifdef GREB_TUNING
    circ.substeps(lds, cur, a.nsub, 0, EXP && (a.xsw & kXQDiffOnly), stamp, &st_busy);
#else
    circ.substeps(lds, cur, a.nsub, 0, EXP && (a.xsw & kXQDiffOnly));
#endif
    cur ^= a.nsub & 1;
    GREB_STAMP(t_c);

    // ---- point physics on the OLD state + Euler update (:254-268 / :328-361)
    const Phys P = a.phys[m];
    const float co2 = FLUX ? a.co2_flux : a.co2[(size_t)m * a.co2_stride + a.co2_year0 + yr_rel]; // :924
    lfloat* Xf = lds + kOffX + cur * XB;       // the tracers after the 24 sub-steps
    lfloat* red = lds + kOffX + (cur ^ 1) * XB; // idle buffer: annual-mean reduction scratch
    // Each thread takes whole quads (4 consecutive longitudes): every load/store of the ~26
    // fields a point touches is one dwordx4 and all of a quad's loads are in flight together.
    // 1152 quads on 512 threads: two full passes and a quarter-full one.  Measured with the stamp build
    // (GREB_DEBUG_PHYS): the first pass of a step costs 13 400 cycles, the other two together 12 500 -- the phase is
    // VALU-bound (1 170 instructions per quad) plus a cold start of ~5 000 cycles; giving the last 512 points to
    // all threads as single points (a second code path) measured 38 700 cycles against 25 900; the per-member operands of
    // the next pass requested one pass ahead (56 more live VGPRs): 37 100; warm-up loads for the phase's 28 arrays
    // issued under the sub-steps: no change; the next step's winds fetched before the phase and stored after it (24 live
    // VGPRs, wind staging 2 600 -> 250 cycles): the phase itself 35 000.  I-cache misses are nil (SQC_ICACHE_MISSES 41
    // per member-year).  What the slower variants share: vector-memory operations retire in order, so the loop's waits
    // for its loads also wait for the previous pass's stores; in the form below the compiler gets away with
    // s_waitcnt vmcnt(6) where the variants end up at vmcnt(0).  (All A/B runs in one gpurun call: boxes differ.)
    // Also measured: the next quad's operands requested BETWEEN the arithmetic of the current quad and its stores
    // (physics_load / physics_compute / physics_store, the inputs dead by then): 116 + 56 live VGPRs of operands and
    // results still spill 68 registers at the 256 cap, and the phase takes 41 600 cycles instead of 24 900.
#pragma unroll 1
    for (int qd = tid; qd < NP / 4; qd += kThreads) {
#ifdef GREB_TUNING
      if ((a.dbg & 4) && qd >= kThreads) break; // timing experiment: one pass of the three
#endif
      const q8 xpair = ld8(Xf + (qd / NQ) * RS, qd % NQ);
      f4 oTa, oq, tsm;
      physics_quad<STRICT, FLUX, EXP>(a, P, m, qd, ck, co2, state, acc, corr, comp(xpair, 0), comp(xpair, 1), oTa, oq, tsm);
      st8(Xf + (qd / NQ) * RS, qd % NQ, zip(oTa, oq));
      if (ityr == kNT) st4(red + 4 * qd, tsm);
    }
    if (ityr == kNT) {
      __syncthreads();
      if (STRICT) {
        if (tid == 0 && a.yearly) {
#pragma clang fp contract(off)
          float sum = 0.f; // the reference's sum() lowers to a sequential fp32 loop; same order here
          for (int i = 0; i < NP; ++i) sum += red[i];
          float* y = a.yearly + ((size_t)m * a.yearly_years + (a.yearly_year0 + yr_rel)) * 2;
          y[0] = sum / (float)NP - 273.15f;                                               // :954
          y[1] = red[(a.ipy - 1) * NX + (a.ipx - 1)] - 273.15f;
        }
      } else if (a.yearly) { // FAST: per-lane partial sums, wavefront shuffle reduction, 8 wave totals through LDS
        float part = 0.f;
        for (int i = tid; i < NP; i += kThreads) part += red[i];
        part = wave_sum(part);
        const float point = red[(a.ipy - 1) * NX + (a.ipx - 1)];
        __syncthreads(); // everyone has read red[]; its first words now carry the wave totals
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        if (tid == 0) {
          float sum = 0.f;
          for (int w = 0; w < kThreads / 64; ++w) sum += red[w];
          float* y = a.yearly + ((size_t)m * a.yearly_years + (a.yearly_year0 + yr_rel)) * 2;
          y[0] = sum / (float)NP - 273.15f;                                               // :954
          y[1] = point - 273.15f;
        }
      }
    }
    __syncthreads();
#ifdef GREB_TUNING
    if (stamp) {
      const unsigned long long t_d = __builtin_amdgcn_s_memtime();
      st_wind += t_b - t_a; st_circ += t_c - t_b; st_phys += t_d - t_c;
    }
#endif
  }
#ifdef GREB_TUNING
  if (stamp && (tid & 63) == 0) {
    unsigned long long* o = a.stamps + ((size_t)m * (kThreads / 64) + (tid >> 6)) * 8;
    o[0] = __builtin_amdgcn_s_memtime() - st_c0; o[1] = __builtin_amdgcn_s_memrealtime() - st_r0;
    o[2] = st_wind; o[3] = st_circ; o[4] = st_busy; o[5] = st_phys; o[6] = (unsigned long long)a.nsteps; o[7] = (unsigned long long)a.nsub;
  }
#endif
  // all five state fields were written back every step
}

hipError_t launch_member_kernel(const MemberArgs& a, int n_members, bool strict, hipStream_t s) {
  if (a.nx != NX || a.ny != NY) return hipErrorInvalidValue;
  void (*kern)(MemberArgs);
  if (a.xsw) { // sensitivity experiment: the switch-aware instantiation
    if (a.flux_phase) kern = strict ? member_kernel<true, true, true> : member_kernel<false, true, true>;
    else kern = strict ? member_kernel<true, false, true> : member_kernel<false, false, true>;
  } else if (a.flux_phase) kern = strict ? member_kernel<true, true, false> : member_kernel<false, true, false>;
  else kern = strict ? member_kernel<true, false, false> : member_kernel<false, false, false>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(n_members), dim3(kThreads), kLdsBytes, s, a);
  return hipGetLastError();
}

} // namespace greb
