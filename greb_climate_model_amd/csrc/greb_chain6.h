// greb_chain6.h -- the sub-cycled zonal sweep of a latitude circle held 6 points per lane, as 33 + 3 instructions.
//
// The circle is either the whole wavefront (64 lanes x 6 = 384 points: the 384x192 grid, one chain per wave) or one
// DPP row of 16 lanes (16 x 6 = 96 points: the 96x48 grid, four independent chains per wave -- the fused engine's
// polar rows).  The long polar chains (src/greb.f90:651-719, :837-911; at 384x192 up to 225 -- 1 800 for
// kappa < 7.27e5 -- DEPENDENT sweeps per diffusion call) are latency: a lone wavefront issues one
// vector instruction per 4.0 cycles whatever the instruction is (tools/ubench/chain_rate.hip): the chain's latency is
// its instruction count -- 33 for the sweep, 3 more where the clamp minimum has to be carried along (chain_run6).  FAST
// arithmetic only (greb_stencil.h: chain_lon_regs): during a chain the weights, the row constant and the wind are
// fixed, so the increment of point c is a fixed linear form in the six differences around it,
//     d[i] = sum_m K[i][m] * e[i+m],  e[j] = T[j+1] - T[j]  over the window T[0..11] = (prev lane's o[3..5], o[0..5],
//     next lane's o[0..2]),           n[i] = o[i] + d[i],   mn = min_i n[i]  (the clamp test, see chain_sweeps6).
// What makes it 33 (+ 3 for the minimum) instructions instead of the compiler's 65 for the same arithmetic:
//   * points are paired (i, i+3): (o0,o3), (o1,o4), (o2,o5).  Term m of the pair then needs (e[i+m], e[i+m+3]) -- one
//     alignment only, so e lives in four register pairs E2..E5 = (e2,e5), (e3,e6), (e4,e7), (e5,e8), two of them a
//     single v_pk_add of the point pairs, and 12 of the 36 multiply-adds are v_pk_fma_f32 (same issue slot as a scalar
//     one for a lone wavefront);
//   * the halo is carried as DIFFERENCES: e0, e1 are the previous lane's e6, e7 and e9, e10 the next lane's e3, e4;
//     the 12 terms that touch them are scalar v_mul/v_fmac whose DPP operand (wave_ror/rol:1 -- the rotate is the
//     row's periodic boundary) fetches the neighbour's value in the same instruction.  Only e2 and e8, which straddle
//     a lane boundary, cost a DPP subtract;
//   * input and output points use two register sets (A -> B, B -> A) and the loop itself is part of the asm
//     statement (chain_sweeps6), so nothing is copied at the loop edge.
// Same operands, same operations, same order per point as differencing a refreshed T halo and summing m = 0..5 (the
// form chain_increments6 below spells out in C++): results are bit-identical to it.
//
// The body addresses halves of the point / difference / increment pairs, which inline-asm operands cannot express,
// so those live in FIXED registers v64..v90 (the points bound with "{vN}" constraints, the rest clobbered); the
// coefficients are ordinary operands.  Every DPP read is >= 3 instructions behind the write of its
// source (the hazard recogniser does not look inside an asm body; the hardware needs 2).
#pragma once
#include <type_traits>

#include "greb_device.h"

namespace greb {

// point pairs: set A = v[64:69], set B = v[70:75]; lo = point i, hi = point i+3
#define GREB_C6_A0 "v[64:65]"
#define GREB_C6_A0L "v64"
#define GREB_C6_A0H "v65"
#define GREB_C6_A1 "v[66:67]"
#define GREB_C6_A1L "v66"
#define GREB_C6_A1H "v67"
#define GREB_C6_A2 "v[68:69]"
#define GREB_C6_A2L "v68"
#define GREB_C6_A2H "v69"
#define GREB_C6_B0 "v[70:71]"
#define GREB_C6_B0L "v70"
#define GREB_C6_B0H "v71"
#define GREB_C6_B1 "v[72:73]"
#define GREB_C6_B1L "v72"
#define GREB_C6_B1H "v73"
#define GREB_C6_B2 "v[74:75]"
#define GREB_C6_B2L "v74"
#define GREB_C6_B2H "v75"
// E2 = (e2,e5) v[76:77], E3 = (e3,e6) v[78:79], E4 = (e4,e7) v[80:81], E5 = (e5,e8) v[82:83]
// D0 = (d0,d3) v[84:85], D1 = (d1,d4) v[86:87], D2 = (d2,d5) v[88:89], mn v90
// coefficients: operands %[kIM] = K[I][M] (scalars) and %[pIM] = (K[I][M], K[I+3][M]) (aligned pairs), see ChainK
// neighbour lanes: flavour W = the circle is the whole wavefront (64 lanes x 6 = 384 points), flavour R = each DPP row
// of 16 lanes is a circle of its own (16 x 6 = 96 points: four independent chains per wavefront)
#define GREB_C6_PREV_W " wave_ror:1 row_mask:0xf bank_mask:0xf\n"
#define GREB_C6_NEXT_W " wave_rol:1 row_mask:0xf bank_mask:0xf\n"
#define GREB_C6_PREV_R " row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define GREB_C6_NEXT_R " row_ror:15 row_mask:0xf bank_mask:0xf\n"
#define GREB_C6_SUB2 " neg_lo:[0,1] neg_hi:[0,1]\n"

// one sweep: points in set I, result in set O, neighbour flavour F
#define GREB_C6_SWEEP_BODY(I, O, F, M1, M2, M3)                                                                       \
  "v_pk_add_f32 v[78:79], " GREB_C6_##I##1 ", " GREB_C6_##I##0 GREB_C6_SUB2  /* E3 = (e3, e6) */                      \
  "v_pk_add_f32 v[80:81], " GREB_C6_##I##2 ", " GREB_C6_##I##1 GREB_C6_SUB2  /* E4 = (e4, e7) */                      \
  "v_sub_f32 v82, " GREB_C6_##I##0H ", " GREB_C6_##I##2L "\n"  /* e5 = o3 - o2 */                                     \
  "v_sub_f32 v77, " GREB_C6_##I##0H ", " GREB_C6_##I##2L "\n"  /* e5 again, as E2.hi */                               \
  "v_subrev_f32_dpp v76, " GREB_C6_##I##2H ", " GREB_C6_##I##0L GREB_C6_PREV_##F  /* e2 = o0 - prev o5 */             \
  "v_sub_f32_dpp v83, " GREB_C6_##I##0L ", " GREB_C6_##I##2H GREB_C6_NEXT_##F  /* e8 = next o0 - o5 */                \
  "v_mul_f32_dpp v84, v79, %[k00]" GREB_C6_PREV_##F  /* d0  = K00 * prev e6 */                                        \
  "v_mul_f32 v85, %[k30], v78\n"  /* d3  = K30 * e3 */                                                                \
  "v_mul_f32_dpp v86, v81, %[k10]" GREB_C6_PREV_##F  /* d1  = K10 * prev e7 */                                        \
  "v_mul_f32 v87, %[k40], v80\n"  /* d4  = K40 * e4 */                                                                \
  M1                                                                                                                  \
  "v_pk_mul_f32 v[88:89], %[p20], v[76:77]\n"  /* (d2,d5)  = m0 * (e2,e5) */                                          \
  "v_fmac_f32_dpp v84, v81, %[k01]" GREB_C6_PREV_##F  /* d0 += K01 * prev e7 */                                       \
  "v_fmac_f32 v85, %[k31], v80\n"  /* d3 += K31 * e4 */                                                               \
  "v_pk_fma_f32 v[86:87], %[p11], v[76:77], v[86:87]\n"  /* (d1,d4) += m1 * (e2,e5) */                                \
  "v_pk_fma_f32 v[88:89], %[p21], v[78:79], v[88:89]\n"  /* (d2,d5) += m1 * (e3,e6) */                                \
  "v_pk_fma_f32 v[84:85], %[p02], v[76:77], v[84:85]\n"  /* (d0,d3) += m2 * (e2,e5) */                                \
  "v_pk_fma_f32 v[86:87], %[p12], v[78:79], v[86:87]\n"  /* (d1,d4) += m2 * (e3,e6) */                                \
  "v_pk_fma_f32 v[88:89], %[p22], v[80:81], v[88:89]\n"  /* (d2,d5) += m2 * (e4,e7) */                                \
  "v_pk_fma_f32 v[84:85], %[p03], v[78:79], v[84:85]\n"  /* (d0,d3) += m3 * (e3,e6) */                                \
  "v_pk_fma_f32 v[86:87], %[p13], v[80:81], v[86:87]\n"  /* (d1,d4) += m3 * (e4,e7) */                                \
  M2                                                                                                                  \
  "v_pk_fma_f32 v[88:89], %[p23], v[82:83], v[88:89]\n"  /* (d2,d5) += m3 * (e5,e8) */                                \
  "v_pk_fma_f32 v[84:85], %[p04], v[80:81], v[84:85]\n"  /* (d0,d3) += m4 * (e4,e7) */                                \
  "v_pk_fma_f32 v[86:87], %[p14], v[82:83], v[86:87]\n"  /* (d1,d4) += m4 * (e5,e8) */                                \
  "v_fmac_f32 v88, %[k24], v79\n"  /* d2 += K24 * e6 */                                                               \
  "v_fmac_f32_dpp v89, v78, %[k54]" GREB_C6_NEXT_##F  /* d5 += K54 * next e3 */                                       \
  "v_pk_fma_f32 v[84:85], %[p05], v[82:83], v[84:85]\n"  /* (d0,d3) += m5 * (e5,e8) */                                \
  "v_fmac_f32 v86, %[k15], v79\n"  /* d1 += K15 * e6 */                                                               \
  "v_fmac_f32_dpp v87, v78, %[k45]" GREB_C6_NEXT_##F  /* d4 += K45 * next e3 */                                       \
  M3                                                                                                                  \
  "v_fmac_f32 v88, %[k25], v81\n"  /* d2 += K25 * e7 */                                                               \
  "v_fmac_f32_dpp v89, v80, %[k55]" GREB_C6_NEXT_##F  /* d5 += K55 * next e4 */                                       \
  "v_pk_add_f32 " GREB_C6_##O##0 ", " GREB_C6_##I##0 ", v[84:85]\n"                                                   \
  "v_pk_add_f32 " GREB_C6_##O##1 ", " GREB_C6_##I##1 ", v[86:87]\n"                                                   \
  "v_pk_add_f32 " GREB_C6_##O##2 ", " GREB_C6_##I##2 ", v[88:89]\n"
// mn = min of the six updated values: the clamp test of this sweep alone ...
#define GREB_C6_MIN_FRESH(O)                                                                                          \
  "v_min3_f32 v90, " GREB_C6_##O##0L ", " GREB_C6_##O##1L ", " GREB_C6_##O##2L "\n"                                   \
  "v_min3_f32 v90, v90, " GREB_C6_##O##0H ", " GREB_C6_##O##1H "\n"                                                   \
  "v_min3_f32 v90, v90, " GREB_C6_##O##2H ", " GREB_C6_##O##2H "\n"
#define GREB_C6_SWEEP(I, O, F) GREB_C6_SWEEP_BODY(I, O, F, , , ) GREB_C6_MIN_FRESH(O)
// ... or mn carried through the whole chain as the min over every sweep's INPUT points (the same three instructions,
// but nothing waits for them: they sit between the multiply-adds, one dependent on the other only eleven
// instructions apart; the caller adds the last sweep's output)
#define GREB_C6_MIN_IN(I, a, b) "v_min3_f32 v90, v90, " GREB_C6_##I##a ", " GREB_C6_##I##b "\n"
#define GREB_C6_SWEEP_CARRIED(I, O, F)                                                                                \
  GREB_C6_SWEEP_BODY(I, O, F, GREB_C6_MIN_IN(I, 0L, 1L), GREB_C6_MIN_IN(I, 2L, 0H), GREB_C6_MIN_IN(I, 1H, 2H))

// The coefficients as the sweep wants them: 12 scalars (the terms that take a DPP operand or stand alone) and 12
// aligned pairs (K[i][m], K[i+3][m]) for the packed multiply-adds.  Ordinary "v" operands -- the sweep never
// addresses half of a coefficient pair -- so two chains with different coefficients share the code and nothing is
// copied around a sweep.
struct ChainK {
  float s[12]; // K00 K30 K01 K31 K10 K40 K15 K45 K24 K54 K25 K55
  v2 p[12];    // i = 0: m = 2..5;  i = 1: m = 1..4;  i = 2: m = 0..3
};
__device__ __forceinline__ v2 chain_pair(float lo, float hi) {
  // (opaque: building the pair straight from two array elements makes the compiler widen the first element's load to
  // a two-float load of the ARRAY and replace its upper half -- which pins the whole coefficient table in scratch)
  asm("" : "+v"(lo));
  return v2{lo, hi};
}
__device__ __forceinline__ ChainK chain_pack(const float (&K)[6][6]) {
  ChainK c;
  c.s[0] = K[0][0]; c.s[1] = K[3][0]; c.s[2] = K[0][1]; c.s[3] = K[3][1]; c.s[4] = K[1][0]; c.s[5] = K[4][0];
  c.s[6] = K[1][5]; c.s[7] = K[4][5]; c.s[8] = K[2][4]; c.s[9] = K[5][4]; c.s[10] = K[2][5]; c.s[11] = K[5][5];
  c.p[0] = chain_pair(K[0][2], K[3][2]); c.p[1] = chain_pair(K[0][3], K[3][3]);
  c.p[2] = chain_pair(K[0][4], K[3][4]); c.p[3] = chain_pair(K[0][5], K[3][5]);
  c.p[4] = chain_pair(K[1][1], K[4][1]); c.p[5] = chain_pair(K[1][2], K[4][2]);
  c.p[6] = chain_pair(K[1][3], K[4][3]); c.p[7] = chain_pair(K[1][4], K[4][4]);
  c.p[8] = chain_pair(K[2][0], K[5][0]); c.p[9] = chain_pair(K[2][1], K[5][1]);
  c.p[10] = chain_pair(K[2][2], K[5][2]); c.p[11] = chain_pair(K[2][3], K[5][3]);
  return c;
}
// f(integral_constant<0>) ... f(integral_constant<N-1>): indices that are compile-time from the start (an array indexed
// by the variable of a `#pragma unroll` loop is still dynamically indexed when the compiler decides what may live in
// registers -- ChainK went to scratch that way)
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
// K[I][M] out of the packed form
template <int I, int M>
__device__ __forceinline__ float chain_k(const ChainK& c) {
  constexpr int lo = I % 3, hi = I >= 3;
  constexpr int first = lo == 0 ? 2 : (lo == 1 ? 1 : 0); // first packed m of the pair (lo, lo+3)
  if constexpr (M >= first && M < first + 4) return hi ? c.p[4 * lo + M - first].y : c.p[4 * lo + M - first].x;
  else { // the scalars: (0: m 0,1) (1: m 0,5) (2: m 4,5)
    constexpr int slot = lo == 0 ? (M == 0 ? 0 : 2) : (lo == 1 ? (M == 0 ? 4 : 6) : (M == 4 ? 8 : 10));
    return c.s[slot + hi];
  }
}
#define GREB_C6_K_OPERANDS(c)                                                                                         \
  [k00] "v"(c.s[0]), [k30] "v"(c.s[1]), [k01] "v"(c.s[2]), [k31] "v"(c.s[3]), [k10] "v"(c.s[4]), [k40] "v"(c.s[5]),   \
      [k15] "v"(c.s[6]), [k45] "v"(c.s[7]), [k24] "v"(c.s[8]), [k54] "v"(c.s[9]), [k25] "v"(c.s[10]),                 \
      [k55] "v"(c.s[11]), [p02] "v"(c.p[0]), [p03] "v"(c.p[1]), [p04] "v"(c.p[2]), [p05] "v"(c.p[3]),                 \
      [p11] "v"(c.p[4]), [p12] "v"(c.p[5]), [p13] "v"(c.p[6]), [p14] "v"(c.p[7]), [p20] "v"(c.p[8]),                  \
      [p21] "v"(c.p[9]), [p22] "v"(c.p[10]), [p23] "v"(c.p[11])
#define GREB_C6_CLOBBERS                                                                                              \
  "vcc", "scc", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", \
      "v85", "v86", "v87", "v88", "v89", "v90"

// `rem` dependent sweeps, A -> B -> A two per trip (the odd last one A -> B and copied back), as long as no lane
// needs the clamp; returns the number of sweeps still to do.  The whole loop is one asm statement: with the loop in
// C++ the compiler copies the six points into and out of the fixed registers around every sweep (12 v_mov each).
// It stops BEFORE a sweep whose result would need the clamp (min of the updated values <= 0, or NaN), with T holding
// the last state that did not.
#define GREB_C6_B_TO_A                                                                                                \
  "v_mov_b64 v[64:65], v[70:71]\n"                                                                                    \
  "v_mov_b64 v[66:67], v[72:73]\n"                                                                                    \
  "v_mov_b64 v[68:69], v[74:75]\n"
#define GREB_C6_LOOP(F)                                                                                               \
  "s_cmp_lt_i32 %[rem], 2\n"                                                                                          \
  "s_cbranch_scc1 4f\n"                                                                                               \
  "1:\n" GREB_C6_SWEEP(A, B, F)                                                                                       \
  "v_cmp_nlt_f32 vcc, 0, v90\n"                                                                                       \
  "s_cbranch_vccnz 3f\n" GREB_C6_SWEEP(B, A, F)                                                                       \
  "v_cmp_nlt_f32 vcc, 0, v90\n"                                                                                       \
  "s_cbranch_vccnz 2f\n"                                                                                              \
  "s_sub_i32 %[rem], %[rem], 2\n"                                                                                     \
  "s_cmp_ge_i32 %[rem], 2\n"                                                                                          \
  "s_cbranch_scc1 1b\n"                                                                                               \
  "s_branch 4f\n"                                                                                                     \
  "2:\n" /* the second sweep of the trip needs the clamp: keep the first */                                           \
  GREB_C6_B_TO_A                                                                                                      \
  "s_sub_i32 %[rem], %[rem], 1\n"                                                                                     \
  "s_branch 3f\n"                                                                                                     \
  "4:\n" /* none or one left */                                                                                       \
  "s_cmp_lt_i32 %[rem], 1\n"                                                                                          \
  "s_cbranch_scc1 3f\n" GREB_C6_SWEEP(A, B, F)                                                                        \
  "v_cmp_nlt_f32 vcc, 0, v90\n"                                                                                       \
  "s_cbranch_vccnz 3f\n"                                                                                              \
  GREB_C6_B_TO_A                                                                                                      \
  "s_mov_b32 %[rem], 0\n"                                                                                             \
  "3:\n"
#define GREB_C6_T_OPERANDS(T)                                                                                         \
  "+{v64}"(T[0]), "+{v65}"(T[3]), "+{v66}"(T[1]), "+{v67}"(T[4]), "+{v68}"(T[2]), "+{v69}"(T[5])
template <bool ROW16>
__device__ __forceinline__ int chain_sweeps6(float (&T)[6], const ChainK& c, int rem /* wave-uniform */) {
  if constexpr (ROW16)
    asm volatile(GREB_C6_LOOP(R) : [rem] "+s"(rem), GREB_C6_T_OPERANDS(T) : GREB_C6_K_OPERANDS(c) : GREB_C6_CLOBBERS);
  else
    asm volatile(GREB_C6_LOOP(W) : [rem] "+s"(rem), GREB_C6_T_OPERANDS(T) : GREB_C6_K_OPERANDS(c) : GREB_C6_CLOBBERS);
  return rem;
}

// The common case: ALL `n` sweeps without a test in between.  Per sweep this saves the v_cmp, the wait for vcc and a
// branch of chain_sweeps6 -- 48 of its 210 cycles on a lone wavefront, which issues one vector instruction per 4.0
// cycles and pays ~30 for a taken branch (hence four or eight sweeps per trip: 162 cycles per sweep with two, 152 with
// four).  Two flavours of sweep:
//   CARRIED  mn = min over every sweep's input points (the caller adds the last output); if it ends <= 0 some sweep
//            needed the clamp and the caller redoes the chain from its start with chain_sweeps6;
//   PLAIN    no min at all (33 instructions): for chains that cannot reach zero, see chain_stays_positive.
#define GREB_C6_SWEEP_PLAIN(I, O, F) GREB_C6_SWEEP_BODY(I, O, F, , , )
#define GREB_C6_PAIR(S, F) GREB_C6_SWEEP_##S(A, B, F) GREB_C6_SWEEP_##S(B, A, F)
#define GREB_C6_TAIL(S, F, B2, B1) /* the sweeps left after the trips: bits B2 (two) and B1 (one) of `rest` */        \
  "s_bitcmp0_b32 %[rest], 1\n"                                                                                        \
  "s_cbranch_scc1 " #B2 "f\n" GREB_C6_PAIR(S, F)                                                                      \
  #B2 ":\n"                                                                                                           \
  "s_bitcmp0_b32 %[rest], 0\n"                                                                                        \
  "s_cbranch_scc1 " #B1 "f\n" GREB_C6_SWEEP_##S(A, B, F) GREB_C6_B_TO_A                                               \
  #B1 ":\n"
#define GREB_C6_LOOP4(S, F) /* trips = n >> 2, rest = n & 3 */                                                        \
  "s_sub_u32 %[trips], %[trips], 1\n" /* scc = borrow: no trip (left) */                                              \
  "s_cbranch_scc1 2f\n"                                                                                               \
  ".p2align 6\n"                                                                                                      \
  "1:\n"                                                                                                              \
  "s_sub_u32 %[trips], %[trips], 1\n" /* nothing in the sweeps writes scc */                                          \
  GREB_C6_PAIR(S, F) GREB_C6_PAIR(S, F)                                                                               \
  "s_cbranch_scc0 1b\n"                                                                                               \
  "2:\n" GREB_C6_TAIL(S, F, 3, 4)
#define GREB_C6_LOOP8(S, F) /* trips = n >> 3, rest = n & 7 */                                                        \
  "s_sub_u32 %[trips], %[trips], 1\n"                                                                                 \
  "s_cbranch_scc1 2f\n"                                                                                               \
  ".p2align 6\n"                                                                                                      \
  "1:\n"                                                                                                              \
  "s_sub_u32 %[trips], %[trips], 1\n"                                                                                 \
  GREB_C6_PAIR(S, F) GREB_C6_PAIR(S, F) GREB_C6_PAIR(S, F) GREB_C6_PAIR(S, F)                                         \
  "s_cbranch_scc0 1b\n"                                                                                               \
  "2:\n"                                                                                                              \
  "s_bitcmp0_b32 %[rest], 2\n"                                                                                        \
  "s_cbranch_scc1 5f\n" GREB_C6_PAIR(S, F) GREB_C6_PAIR(S, F)                                                         \
  "5:\n" GREB_C6_TAIL(S, F, 3, 4)
#ifndef GREB_C6_W_SHIFT // sweeps per trip of the wavefront-sized circles: 2^3 (A/B: -DGREB_C6_W_SHIFT=2)
#define GREB_C6_W_SHIFT 3
#endif
#if GREB_C6_W_SHIFT == 3
#define GREB_C6_W_LOOP GREB_C6_LOOP8
#else
#define GREB_C6_W_LOOP GREB_C6_LOOP4
#endif
#define GREB_C6_CLOBBERS_ALL                                                                                          \
  "scc", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", \
      "v86", "v87", "v88", "v89"
template <bool ROW16>
__device__ __forceinline__ float chain_sweeps6_all(float (&T)[6], const ChainK& c, int n /* wave-uniform */) {
  float mn = __builtin_inff();
  if constexpr (ROW16) { // the fused engine's polar rows: eight sweeps, and an instruction cache shared by eight roles
    int trips = n >> 2;
    const int rest = n & 3;
    asm volatile(GREB_C6_LOOP4(CARRIED, R)
                 : [trips] "+s"(trips), "+{v90}"(mn), GREB_C6_T_OPERANDS(T)
                 : [rest] "s"(rest), GREB_C6_K_OPERANDS(c)
                 : GREB_C6_CLOBBERS_ALL);
  } else {
    int trips = n >> GREB_C6_W_SHIFT;
    const int rest = n & ((1 << GREB_C6_W_SHIFT) - 1);
    asm volatile(GREB_C6_W_LOOP(CARRIED, W)
                 : [trips] "+s"(trips), "+{v90}"(mn), GREB_C6_T_OPERANDS(T)
                 : [rest] "s"(rest), GREB_C6_K_OPERANDS(c)
                 : GREB_C6_CLOBBERS_ALL);
  }
  return fminf(fminf(fminf(mn, T[0]), fminf(T[1], T[2])), fminf(fminf(T[3], T[4]), T[5]));
}
// ... and without the min (wavefront-sized circles only: the long chains of the 384-wide grid)
__device__ __forceinline__ void chain_sweeps6_plain(float (&T)[6], const ChainK& c, int n /* wave-uniform */) {
  int trips = n >> GREB_C6_W_SHIFT;
  const int rest = n & ((1 << GREB_C6_W_SHIFT) - 1);
  asm volatile(GREB_C6_W_LOOP(PLAIN, W)
               : [trips] "+s"(trips), GREB_C6_T_OPERANDS(T)
               : [rest] "s"(rest), GREB_C6_K_OPERANDS(c)
               : GREB_C6_CLOBBERS_ALL);
}

// The same sweep in C++ with its increments handed back (any registers, the compiler's schedule, ~65 instructions):
// the checked path of chain_run6.  Same operations in the same order as the asm body.
template <bool ROW16>
__device__ __forceinline__ float chain_from_prev(float x) { // lane l <- lane l-1 of the circle
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ROW16 ? 0x121 : 0x13C, 0xf, 0xf, false));
}
template <bool ROW16>
__device__ __forceinline__ float chain_from_next(float x) { // lane l <- lane l+1 of the circle
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ROW16 ? 0x12F : 0x134, 0xf, 0xf, false));
}
template <bool ROW16>
__device__ __forceinline__ void chain_increments6(const float (&o)[6], const ChainK& c, float (&d)[6]) {
  float e[11];
#pragma unroll
  for (int m = 3; m <= 7; ++m) e[m] = o[m - 2] - o[m - 3];
  e[2] = o[0] - chain_from_prev<ROW16>(o[5]);
  e[8] = chain_from_next<ROW16>(o[0]) - o[5];
  e[0] = chain_from_prev<ROW16>(e[6]); e[1] = chain_from_prev<ROW16>(e[7]);
  e[9] = chain_from_next<ROW16>(e[3]); e[10] = chain_from_next<ROW16>(e[4]);
  static_for<6>([&](auto I) {
    constexpr int i = I;
    float a = chain_k<i, 0>(c) * e[i];
    a = __builtin_fmaf(chain_k<i, 1>(c), e[i + 1], a);
    a = __builtin_fmaf(chain_k<i, 2>(c), e[i + 2], a);
    a = __builtin_fmaf(chain_k<i, 3>(c), e[i + 3], a);
    a = __builtin_fmaf(chain_k<i, 4>(c), e[i + 4], a);
    a = __builtin_fmaf(chain_k<i, 5>(c), e[i + 5], a);
    d[i] = a;
  });
}

// A DIFFUSION chain that provably never needs the clamp, so that its sweeps need no test at all.
// n(c) = T(c) + sum_m K[m] e[c-3+m] with e[j] = T[j+1] - T[j] is sum_j a_j T(c+j) with
//   a = (-K0, K0-K1, K1-K2, 1+K2-K3, K3-K4, K4-K5, K5),  sum a_j = 1.
// If every a_j >= 0 (with the smooth weights of the model they are: 6 w(c+1) >= 3 w(c+2) >= w(c+3) >= 0 and the
// sub-cycling keeps the centre weight up) the sweep is a convex combination: in exact arithmetic the values stay inside
// [m, M] = [min, max] of the row.  In fp32 (u = 2^-24) a sweep is off by at most u M for the final rounding plus
// 7 u sum|K| (M - m) for the differences and the multiply-add chain, and sum|K| <= 3 (K3 - K2) <= 2.25 when the centre
// weight is >= 0.25: less than 17 u M per sweep.  Over N <= 2048 sweeps the minimum falls by less than 2.1e-3 M, so
// M <= 64 m (margin 7) keeps every state of the chain positive: no point can meet d <= -T.  The tests on K are exact
// (fp32 comparisons of the stored coefficients, which ARE the map); NaN points are ignored by fminf/fmaxf here as they
// are by the clamp's comparison.  Rows that fail any of it (vapour over the ice sheets can) take the carried-min loop.
template <int N>
__device__ __forceinline__ float chain_row_ror(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + N, 0xf, 0xf, false));
}
template <bool MAX>
__device__ __forceinline__ float chain_wave_extreme(float x) { // over the 64 lanes, the same value in every lane
  auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
  x = op(x, chain_row_ror<1>(x)); x = op(x, chain_row_ror<2>(x));
  x = op(x, chain_row_ror<4>(x)); x = op(x, chain_row_ror<8>(x));
  const int i = __float_as_int(x);
  const float a = __int_as_float(__builtin_amdgcn_readlane(i, 0)), b = __int_as_float(__builtin_amdgcn_readlane(i, 16));
  const float c = __int_as_float(__builtin_amdgcn_readlane(i, 32)), d = __int_as_float(__builtin_amdgcn_readlane(i, 48));
  return op(op(a, b), op(c, d));
}
constexpr int kChainPlainMinSweeps = 64, kChainPlainMaxSweeps = 2048; // (the test costs about three sweeps)
// ... its two halves: what depends on the coefficients alone (every a_j >= 0 with the centre margin, in every lane) ...
__device__ __forceinline__ bool chain_coefficients_convex(const float (&K)[6][6]) {
  // the smallest of the seven weights a_j, the centre one less its margin (the sign of an fp32 difference is exact);
  // branch-free: written with && the 42 conditions become 42 exec-mask branches, 2 600 cycles
  float slack = __builtin_inff();
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    slack = fminf(fminf(slack, -K[i][0]), fminf(K[i][0] - K[i][1], K[i][1] - K[i][2]));
    slack = fminf(fminf(slack, K[i][5]), fminf(K[i][3] - K[i][4], K[i][4] - K[i][5]));
    slack = fminf(slack, (0.75f + K[i][2]) - K[i][3]);
  }
  return __builtin_amdgcn_ballot_w64(!(slack >= 0.f)) == 0; // wave-uniform
}
// ... and what depends on the row: 0 < min, max <= 64 min
__device__ __forceinline__ bool chain_range_positive(const float (&T)[6]) {
  const float lo = chain_wave_extreme<false>(fminf(fminf(fminf(T[0], T[1]), fminf(T[2], T[3])), fminf(T[4], T[5])));
  const float hi = chain_wave_extreme<true>(fmaxf(fmaxf(fmaxf(T[0], T[1]), fmaxf(T[2], T[3])), fmaxf(T[4], T[5])));
  return lo >= 1e-30f && hi <= 64.f * lo;
}
__device__ __forceinline__ bool chain_sweeps_plain_range(int time2) { return time2 >= kChainPlainMinSweeps && time2 <= kChainPlainMaxSweeps; }
__device__ __forceinline__ bool chain_stays_positive(const float (&T)[6], const float (&K)[6][6], int time2) {
  if (!chain_sweeps_plain_range(time2)) return false;
  // (a lane whose coefficients fail vetoes through the minimum: one pair of wave reductions for both halves)
  float slack = __builtin_inff();
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    slack = fminf(fminf(slack, -K[i][0]), fminf(K[i][0] - K[i][1], K[i][1] - K[i][2]));
    slack = fminf(fminf(slack, K[i][5]), fminf(K[i][3] - K[i][4], K[i][4] - K[i][5]));
    slack = fminf(slack, (0.75f + K[i][2]) - K[i][3]);
  }
  float lo = fminf(fminf(fminf(T[0], T[1]), fminf(T[2], T[3])), fminf(T[4], T[5]));
  lo = fminf(lo, slack >= 0.f ? __builtin_inff() : -1.f);
  lo = chain_wave_extreme<false>(lo);
  const float hi = chain_wave_extreme<true>(fmaxf(fmaxf(fmaxf(T[0], T[1]), fmaxf(T[2], T[3])), fmaxf(T[4], T[5])));
  return lo >= 1e-30f && hi <= 64.f * lo;
}

// time2 dependent sweeps with the clamp `where(dTxh <= -T1h) dTxh = -0.9*T1h` (:715 / :907).  d <= -T implies
// fl(T + d) <= 0 (rounding is monotonic), so the min over the updated values decides whether any lane needs the
// reference's per-point select.  The chain is first run WITHOUT a test between its sweeps, the min carried along; if
// any state of it was <= 0 (rare: the test inputs with zeros and spikes) it is redone from its start by the stop-before
// loop, whose offending sweeps take the checked path.  time2 must be the same in every lane of the wavefront.
template <bool ROW16>
__device__ __forceinline__ void chain_run6(float (&T)[6], const ChainK& c, int time2, bool positive = false) {
  const int n = __builtin_amdgcn_readfirstlane(time2);
  if constexpr (!ROW16) {
    if (positive) { // chain_stays_positive: wave-uniform
      chain_sweeps6_plain(T, c, n);
      return;
    }
  }
  float T0[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) T0[i] = T[i];
  // min over every state of the chain; a NaN point is skipped by v_min3/fminf -- and by the reference's select
  // (d <= -T is false for it), so it needs no checked path either
  const float mn = chain_sweeps6_all<ROW16>(T, c, n);
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(mn > 0.f)) == 0, 1)) return;
  // (the rare path is kept out of the common one: written as one loop around the asm statement, the compiler surrounds
  // every chain with its flag and copy bookkeeping, ~25 instructions)
#pragma unroll
  for (int i = 0; i < 6; ++i) T[i] = T0[i];
  int rem = chain_sweeps6<ROW16>(T, c, n);
  while (rem > 0) {
    float d[6];
    chain_increments6<ROW16>(T, c, d);
#pragma unroll
    for (int i = 0; i < 6; ++i) T[i] = T[i] + ((d[i] <= -T[i]) ? -0.9f * T[i] : d[i]); // the reference's select
    if (--rem > 0) rem = chain_sweeps6<ROW16>(T, c, rem);
  }
}

} // namespace greb
