// greb_chain6.h -- the sub-cycled zonal sweep of a 384-point row held 6 points per lane, as 36 instructions.
//
// The long polar chains of the 384x192 grid (src/greb.f90:651-719, :837-911; up to 225 -- 1 800 for kappa < 7.27e5 --
// DEPENDENT sweeps per diffusion call) are executed by one wavefront per (row, tracer), and a lone wavefront issues one
// vector instruction per ~5 cycles whatever the instruction is: the chain's latency is its instruction count.  FAST
// arithmetic only (greb_stencil.h: chain_lon_regs): during a chain the weights, the row constant and the wind are
// fixed, so the increment of point c is a fixed linear form in the six differences around it,
//     d[i] = sum_m K[i][m] * e[i+m],  e[j] = T[j+1] - T[j]  over the window T[0..11] = (prev lane's o[3..5], o[0..5],
//     next lane's o[0..2]),           n[i] = o[i] + d[i],   mn = min_i n[i]  (the clamp test, see chain_run6).
// What makes it 36 instructions instead of the compiler's 65 for the same arithmetic:
//   * points are paired (i, i+3): (o0,o3), (o1,o4), (o2,o5).  Term m of the pair then needs (e[i+m], e[i+m+3]) -- one
//     alignment only, so e lives in four register pairs E2..E5 = (e2,e5), (e3,e6), (e4,e7), (e5,e8), two of them a
//     single v_pk_add of the point pairs, and 12 of the 36 multiply-adds are v_pk_fma_f32 (same issue slot as a scalar
//     one for a lone wavefront);
//   * the halo is carried as DIFFERENCES: e0, e1 are the previous lane's e6, e7 and e9, e10 the next lane's e3, e4;
//     the 12 terms that touch them are scalar v_mul/v_fmac whose DPP operand (wave_ror/rol:1 -- the rotate is the
//     row's periodic boundary) fetches the neighbour's value in the same instruction.  Only e2 and e8, which straddle
//     a lane boundary, cost a DPP subtract;
//   * input and output points use two register sets (A -> B, B -> A) and the loop itself is part of the asm
//     statement (chain_pairs6), so nothing is copied at the loop edge.
// Same operands, same operations, same order per point as differencing a refreshed T halo and summing m = 0..5
// (pair_chain_row in greb_pair_sweep.hip is that form on (Tair,q) pairs): results are bit-identical to it.
//
// The body addresses halves of register pairs, which inline-asm operands cannot express, so it works on FIXED
// registers v64..v127 (bound with "{vN}" constraints).  Every DPP read is >= 3 instructions behind the write of its
// source (the hazard recogniser does not look inside an asm body; the hardware needs 2).
#pragma once

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
// scalar coefficients v91..v102: K00 K30 K01 K31 K10 K40 K15 K45 K24 K54 K25 K55
// coefficient pairs (K[i][m], K[i+3][m]): i=0, m=2..5 v[104:111]; i=1, m=1..4 v[112:119]; i=2, m=0..3 v[120:127]
#define GREB_C6_PREV " wave_ror:1 row_mask:0xf bank_mask:0xf\n"
#define GREB_C6_NEXT " wave_rol:1 row_mask:0xf bank_mask:0xf\n"
#define GREB_C6_SUB2 " neg_lo:[0,1] neg_hi:[0,1]\n"

// one sweep: points in set I, result in set O
#define GREB_C6_SWEEP(I, O)                                                                                            \
  "v_pk_add_f32 v[78:79], " GREB_C6_##I##1 ", " GREB_C6_##I##0 GREB_C6_SUB2            /* E3 = (e3, e6) */            \
  "v_pk_add_f32 v[80:81], " GREB_C6_##I##2 ", " GREB_C6_##I##1 GREB_C6_SUB2            /* E4 = (e4, e7) */            \
  "v_sub_f32 v82, " GREB_C6_##I##0H ", " GREB_C6_##I##2L "\n"                          /* e5 = o3 - o2 */             \
  "v_sub_f32 v77, " GREB_C6_##I##0H ", " GREB_C6_##I##2L "\n"                          /* e5 again, as E2.hi */       \
  "v_subrev_f32_dpp v76, " GREB_C6_##I##2H ", " GREB_C6_##I##0L GREB_C6_PREV           /* e2 = o0 - prev o5 */        \
  "v_sub_f32_dpp v83, " GREB_C6_##I##0L ", " GREB_C6_##I##2H GREB_C6_NEXT              /* e8 = next o0 - o5 */        \
  "v_mul_f32_dpp v84, v79, v91" GREB_C6_PREV                                           /* d0  = K00 * prev e6 */      \
  "v_mul_f32 v85, v92, v78\n"                                                          /* d3  = K30 * e3 */           \
  "v_mul_f32_dpp v86, v81, v95" GREB_C6_PREV                                           /* d1  = K10 * prev e7 */      \
  "v_mul_f32 v87, v96, v80\n"                                                          /* d4  = K40 * e4 */           \
  "v_pk_mul_f32 v[88:89], v[120:121], v[76:77]\n"                                      /* (d2,d5)  = m0 * (e2,e5) */  \
  "v_fmac_f32_dpp v84, v81, v93" GREB_C6_PREV                                          /* d0 += K01 * prev e7 */      \
  "v_fmac_f32 v85, v94, v80\n"                                                         /* d3 += K31 * e4 */           \
  "v_pk_fma_f32 v[86:87], v[112:113], v[76:77], v[86:87]\n"                            /* (d1,d4) += m1 * (e2,e5) */  \
  "v_pk_fma_f32 v[88:89], v[122:123], v[78:79], v[88:89]\n"                            /* (d2,d5) += m1 * (e3,e6) */  \
  "v_pk_fma_f32 v[84:85], v[104:105], v[76:77], v[84:85]\n"                            /* (d0,d3) += m2 * (e2,e5) */  \
  "v_pk_fma_f32 v[86:87], v[114:115], v[78:79], v[86:87]\n"                            /* (d1,d4) += m2 * (e3,e6) */  \
  "v_pk_fma_f32 v[88:89], v[124:125], v[80:81], v[88:89]\n"                            /* (d2,d5) += m2 * (e4,e7) */  \
  "v_pk_fma_f32 v[84:85], v[106:107], v[78:79], v[84:85]\n"                            /* (d0,d3) += m3 * (e3,e6) */  \
  "v_pk_fma_f32 v[86:87], v[116:117], v[80:81], v[86:87]\n"                            /* (d1,d4) += m3 * (e4,e7) */  \
  "v_pk_fma_f32 v[88:89], v[126:127], v[82:83], v[88:89]\n"                            /* (d2,d5) += m3 * (e5,e8) */  \
  "v_pk_fma_f32 v[84:85], v[108:109], v[80:81], v[84:85]\n"                            /* (d0,d3) += m4 * (e4,e7) */  \
  "v_pk_fma_f32 v[86:87], v[118:119], v[82:83], v[86:87]\n"                            /* (d1,d4) += m4 * (e5,e8) */  \
  "v_fmac_f32 v88, v99, v79\n"                                                         /* d2 += K24 * e6 */           \
  "v_fmac_f32_dpp v89, v78, v100" GREB_C6_NEXT                                         /* d5 += K54 * next e3 */      \
  "v_pk_fma_f32 v[84:85], v[110:111], v[82:83], v[84:85]\n"                            /* (d0,d3) += m5 * (e5,e8) */  \
  "v_fmac_f32 v86, v97, v79\n"                                                         /* d1 += K15 * e6 */           \
  "v_fmac_f32_dpp v87, v78, v98" GREB_C6_NEXT                                          /* d4 += K45 * next e3 */      \
  "v_fmac_f32 v88, v101, v81\n"                                                        /* d2 += K25 * e7 */           \
  "v_fmac_f32_dpp v89, v80, v102" GREB_C6_NEXT                                         /* d5 += K55 * next e4 */      \
  "v_pk_add_f32 " GREB_C6_##O##0 ", " GREB_C6_##I##0 ", v[84:85]\n"                                                   \
  "v_pk_add_f32 " GREB_C6_##O##1 ", " GREB_C6_##I##1 ", v[86:87]\n"                                                   \
  "v_pk_add_f32 " GREB_C6_##O##2 ", " GREB_C6_##I##2 ", v[88:89]\n"                                                   \
  "v_min3_f32 v90, " GREB_C6_##O##0L ", " GREB_C6_##O##1L ", " GREB_C6_##O##2L "\n"                                   \
  "v_min3_f32 v90, v90, " GREB_C6_##O##0H ", " GREB_C6_##O##1H "\n"                                                   \
  "v_min3_f32 v90, v90, " GREB_C6_##O##2H ", " GREB_C6_##O##2H "\n"

// the coefficient operands, common to both directions
#define GREB_C6_K_OPERANDS(K)                                                                                          \
  "{v91}"(K[0][0]), "{v92}"(K[3][0]), "{v93}"(K[0][1]), "{v94}"(K[3][1]), "{v95}"(K[1][0]), "{v96}"(K[4][0]),          \
      "{v97}"(K[1][5]), "{v98}"(K[4][5]), "{v99}"(K[2][4]), "{v100}"(K[5][4]), "{v101}"(K[2][5]), "{v102}"(K[5][5]),   \
      "{v104}"(K[0][2]), "{v105}"(K[3][2]), "{v106}"(K[0][3]), "{v107}"(K[3][3]), "{v108}"(K[0][4]),                   \
      "{v109}"(K[3][4]), "{v110}"(K[0][5]), "{v111}"(K[3][5]), "{v112}"(K[1][1]), "{v113}"(K[4][1]),                   \
      "{v114}"(K[1][2]), "{v115}"(K[4][2]), "{v116}"(K[1][3]), "{v117}"(K[4][3]), "{v118}"(K[1][4]),                   \
      "{v119}"(K[4][4]), "{v120}"(K[2][0]), "{v121}"(K[5][0]), "{v122}"(K[2][1]), "{v123}"(K[5][1]),                   \
      "{v124}"(K[2][2]), "{v125}"(K[5][2]), "{v126}"(K[2][3]), "{v127}"(K[5][3])
#define GREB_C6_D_OPERANDS(d, mn)                                                                                      \
  "={v84}"(d[0]), "={v85}"(d[3]), "={v86}"(d[1]), "={v87}"(d[4]), "={v88}"(d[2]), "={v89}"(d[5]), "={v90}"(mn)
#define GREB_C6_E_CLOBBERS "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83"

#define GREB_C6_T_CLOBBERS "v70", "v71", "v72", "v73", "v74", "v75", "v84", "v85", "v86", "v87", "v88", "v89", "v90"

// Sweeps two at a time (A -> B -> A) while at least two remain and no lane needs the clamp; returns the number of
// sweeps still to do.  The whole loop is one asm statement: with the loop in C++ the compiler copies the six points
// into and out of the fixed registers around every sweep (12 v_mov per sweep).  It stops BEFORE a sweep whose result
// would need the clamp (min of the updated values <= 0, or NaN), with T holding the last state that did not.
__device__ __forceinline__ int chain_pairs6(float (&T)[6], const float (&K)[6][6], int rem /* wave-uniform */) {
  asm volatile("s_cmp_lt_i32 %[rem], 2\n"
               "s_cbranch_scc1 3f\n"
               "1:\n" GREB_C6_SWEEP(A, B)
               "v_cmp_nlt_f32 vcc, 0, v90\n"
               "s_cbranch_vccnz 3f\n" GREB_C6_SWEEP(B, A)
               "v_cmp_nlt_f32 vcc, 0, v90\n"
               "s_cbranch_vccnz 2f\n"
               "s_sub_i32 %[rem], %[rem], 2\n"
               "s_cmp_ge_i32 %[rem], 2\n"
               "s_cbranch_scc1 1b\n"
               "s_branch 3f\n"
               "2:\n" // the second sweep of the trip needs the clamp: keep the first
               "v_mov_b64 v[64:65], v[70:71]\n"
               "v_mov_b64 v[66:67], v[72:73]\n"
               "v_mov_b64 v[68:69], v[74:75]\n"
               "s_sub_i32 %[rem], %[rem], 1\n"
               "3:\n"
               : [rem] "+s"(rem), "+{v64}"(T[0]), "+{v65}"(T[3]), "+{v66}"(T[1]), "+{v67}"(T[4]), "+{v68}"(T[2]),
                 "+{v69}"(T[5])
               : GREB_C6_K_OPERANDS(K)
               : "vcc", "scc", GREB_C6_E_CLOBBERS, GREB_C6_T_CLOBBERS);
  return rem;
}

// one sweep with its increments and the clamp test handed back (the checked path of chain_run6)
__device__ __forceinline__ void chain_sweep6(const float (&o)[6], const float (&K)[6][6], float (&n)[6], float (&d)[6],
                                             float& mn) {
  asm(GREB_C6_SWEEP(A, B)
      : "={v70}"(n[0]), "={v71}"(n[3]), "={v72}"(n[1]), "={v73}"(n[4]), "={v74}"(n[2]), "={v75}"(n[5]),
        GREB_C6_D_OPERANDS(d, mn)
      : "{v64}"(o[0]), "{v65}"(o[3]), "{v66}"(o[1]), "{v67}"(o[4]), "{v68}"(o[2]), "{v69}"(o[5]), GREB_C6_K_OPERANDS(K)
      : GREB_C6_E_CLOBBERS);
}

// time2 dependent sweeps with the clamp `where(dTxh <= -T1h) dTxh = -0.9*T1h` (:715 / :907).  d <= -T implies
// fl(T + d) <= 0 (rounding is monotonic), so the min over the updated values decides whether any lane needs the
// reference's per-point select; a sweep that does (and the odd last sweep of a chain) takes the checked path.
__device__ __forceinline__ void chain_run6(float (&T)[6], const float (&K)[6][6], int time2) {
  int rem = __builtin_amdgcn_readfirstlane(time2);
  while (rem > 0) {
    rem = chain_pairs6(T, K, rem);
    if (rem == 0) break;
    float U[6], d[6], mn;
    chain_sweep6(T, K, U, d, mn);
    if (__builtin_expect(!(mn > 0.f), 0)) { // also taken for a NaN minimum: the reference's own comparisons then decide
#pragma unroll
      for (int i = 0; i < 6; ++i) U[i] = T[i] + ((d[i] <= -T[i]) ? -0.9f * T[i] : d[i]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) T[i] = U[i];
    --rem;
  }
}

} // namespace greb
