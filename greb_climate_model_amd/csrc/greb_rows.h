// greb_rows.h -- the wavefront-private row machinery of the 384-wide row-strip kernels (greb_rows.hip: the batched
// diffusion sweep; greb_step_rows.hip: the engine's circulation sub-step).
//
// A lane owns 6 consecutive longitudes of a row (64 x 6 = 384).  A PAIR of rows (T and wz, or u and v) travels
// HBM/L2 -> LDS by three LDS-DMA instructions of 1 KiB (global_load_lds_dwordx4: 16 bytes per lane, no VGPRs) into a
// 3 KiB slot laid out  [row A: 1 536 B | second third of row B: 512 B | first two thirds of row B: 1 024 B]  -- the
// middle instruction fetches the last 512 bytes of A (lanes 0-31) and of B (lanes 32-63) -- and is read back 6 floats
// per lane with ds_read_b64 x 3 per row (24-byte lane stride: conflict-free).  Results take the reverse way: ds_write_b64
// x 3 -> ds_read_b128 -> global_store_dwordx4.  The LDS belongs to ONE wavefront: ordering is s_waitcnt only, and all of
// it is inline asm -- the compiler must not know that LDS-DMA and the ds_ reads touch the same bytes, or it drains the
// DMA queue (s_waitcnt vmcnt(0)) in front of every read.  vmcnt is counted by hand: loads, LDS-DMA and stores retire in
// issue order, so "all but the N youngest operations have completed" is the only wait there is.
#pragma once
#include "greb_kernels.h"
#include "greb_stencil.h"

namespace greb {
namespace rows {

constexpr int kNx = 384;
constexpr unsigned kRowB = kNx * 4;   // 1 536 bytes of a row
constexpr unsigned kSlotB = 2 * kRowB; // a pair of rows in LDS

typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

template <int AUX>
__device__ __forceinline__ void glds16(const float* g, lfloat* l) {
  __builtin_amdgcn_global_load_lds((gvoid*)g, (lvoid*)l, 16, 0, AUX);
}

// four bytes per ACTIVE lane, lane l to l[l] (flag words polled without holding a VGPR across the wait)
__device__ __forceinline__ void glds4(const unsigned* g, lfloat* l) {
  __builtin_amdgcn_global_load_lds((gvoid*)g, (lvoid*)l, 4, 0, 16 /* sc1: agent scope */);
  asm volatile("" ::: "memory");
}

// The hand-counted vmcnt scheme needs the vector-memory operations in PROGRAM order: a later LDS-DMA or store scheduled
// ahead of an earlier one would be counted as younger than it is, and a counted wait could then return before the row
// it waits for has landed.  Nothing emitted, the compiler just may not move memory operations across it.
__device__ __forceinline__ void order_fence() { asm volatile("" ::: "memory"); }

#define GREB_ROWS_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define GREB_ROWS_VMCASE(n) case n: GREB_ROWS_VMCNT(n); break;
// every vector-memory operation of this wavefront has completed except (at most) the `younger` most recent ones
__device__ __forceinline__ void wait_all_but(int younger) {
  switch (younger < 31 ? (younger < 0 ? 0 : younger) : 31) {
    GREB_ROWS_VMCASE(0) GREB_ROWS_VMCASE(1) GREB_ROWS_VMCASE(2) GREB_ROWS_VMCASE(3) GREB_ROWS_VMCASE(4)
    GREB_ROWS_VMCASE(5) GREB_ROWS_VMCASE(6) GREB_ROWS_VMCASE(7) GREB_ROWS_VMCASE(8) GREB_ROWS_VMCASE(9)
    GREB_ROWS_VMCASE(10) GREB_ROWS_VMCASE(11) GREB_ROWS_VMCASE(12) GREB_ROWS_VMCASE(13) GREB_ROWS_VMCASE(14)
    GREB_ROWS_VMCASE(15) GREB_ROWS_VMCASE(16) GREB_ROWS_VMCASE(17) GREB_ROWS_VMCASE(18) GREB_ROWS_VMCASE(19)
    GREB_ROWS_VMCASE(20) GREB_ROWS_VMCASE(21) GREB_ROWS_VMCASE(22) GREB_ROWS_VMCASE(23) GREB_ROWS_VMCASE(24)
    GREB_ROWS_VMCASE(25) GREB_ROWS_VMCASE(26) GREB_ROWS_VMCASE(27) GREB_ROWS_VMCASE(28) GREB_ROWS_VMCASE(29)
    GREB_ROWS_VMCASE(30)
    default: GREB_ROWS_VMCNT(31); break;
  }
}

// ... where `younger` is the same number row after row in the middle of a strip: one compare instead of the switch's
// tree of six (a wavefront pays ~100 cycles for that tree, three times per row)
template <int STEADY>
__device__ __forceinline__ void wait_all_but_mostly(int younger) {
  static_assert(STEADY >= 0 && STEADY < 31, "the switch's range");
  if (__builtin_expect(younger == STEADY, 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STEADY) : "memory");
  else wait_all_but(younger);
}

// LDS byte addresses of a lane, relative to the start of a slot / of the output row
struct LaneAddr {
  unsigned a;     // first row of a pair: 24 * lane
  unsigned b[3];  // second row of a pair: its point pairs (2j, 2j+1)
  unsigned r0, r1; // the output row read back 16 bytes per lane: quads 0-63 and (lanes 0-31) 64-95
};
__device__ __forceinline__ LaneAddr lane_addr(unsigned lane) {
  LaneAddr L;
  L.a = 24 * lane;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const unsigned x = 24 * lane + 8 * j;
    L.b[j] = x < 1024 ? 2048 + x : 512 + x;
  }
  L.r0 = 16 * lane;
  L.r1 = lane < 32 ? 1024 + 16 * lane : 0;
  return L;
}

// the per-lane source of the middle LDS-DMA instruction of a pair: the last third of row A (lanes 0-31) or of row B
__device__ __forceinline__ const float* second_halves(const float* A, const float* B, unsigned lane) {
  return lane < 32 ? A + 256 + 4 * lane : B + 256 + 4 * (lane - 32);
}

// The same machinery on a grid HALF as wide (NXR = 192: the 192x96 grid): the wavefront holds the row TWICE, lanes 32-63
// a copy of lanes 0-31.  A latitude circle laid twice around the wavefront's ring of 64 lanes is still that circle --
// lane 31's eastern neighbour is lane 32 = the copy of lane 0, lane 0's western neighbour is lane 63 = the copy of lane
// 31 -- so every wave rotate (the zonal halo, the 33-instruction chain sweep of greb_chain6.h) is right as it stands; the
// copy is made by the LDS-DMA's per-lane source addresses (virtual longitude v of the 384 -> v mod 192) and only lanes
// 0-47 store.  Twice the arithmetic per point, none of it new code.
template <int NXR>
__device__ __forceinline__ int ring_wrap(int v) { // virtual longitude (0 .. 383) -> longitude of the row
  static_assert(NXR == kNx || 2 * NXR == kNx, "the row fills the wavefront's 384 longitudes once or twice");
  return (NXR < kNx && v >= NXR) ? v - NXR : v;
}
// the per-lane sources of the three LDS-DMA instructions of a pair of rows A, B (pointers to the rows' first longitude)
struct PairPtrs { const float* p1; const float* p2; const float* p3; };
template <int NXR>
__device__ __forceinline__ PairPtrs pair_ptrs(const float* A, const float* B, unsigned lane) {
  const int l = (int)lane;
  return PairPtrs{A + ring_wrap<NXR>(4 * l), l < 32 ? A + ring_wrap<NXR>(256 + 4 * l) : B + ring_wrap<NXR>(256 + 4 * (l - 32)),
                  B + ring_wrap<NXR>(4 * l)};
}
// rows A + off, B + off (off: floats, wave-uniform) -> the slot at dst
template <int AUX>
__device__ __forceinline__ void issue_pair_p(const PairPtrs& P, int off, lfloat* dst) {
  glds16<AUX>(P.p1 + off, dst);
  glds16<AUX>(P.p2 + off, dst + 256);
  glds16<AUX>(P.p3 + off, dst + 512);
  order_fence();
}
// the lane whose point 3 is longitude xdim-2 (1-based) of its copy of the row: the reference's index bug (:881)
template <int NXR>
__device__ __forceinline__ bool bug_lane(unsigned lane) { return (lane & (NXR / 6 - 1)) == (unsigned)(NXR / 6 - 1); }
// sixteen result bytes per lane -> the row at `row` (its first longitude): 64 + 32 lanes for 384 longitudes, 48 for 192;
// returns the number of store instructions issued (wave-uniform: the hand-counted vmcnt needs it exact)
template <int NXR, typename Store>
__device__ __forceinline__ int store_row_quads(float* row, unsigned lane, vfloat4 q0, vfloat4 q1, Store&& st) {
  if constexpr (NXR == kNx) {
    st(row + 4 * lane, q0);
    if (lane < 32) st(row + 256 + 4 * lane, q1);
    return 2;
  } else {
    if (4 * lane < (unsigned)NXR) st(row + 4 * lane, q0);
    return 1;
  }
}

// three LDS-DMA instructions: rows A and B (wave-uniform pointers to the rows) -> the slot at dst
template <int AUX>
__device__ __forceinline__ void issue_pair(const float* a_row, const float* b_row, const float* halves_row, lfloat* dst,
                                           unsigned lane) {
  glds16<AUX>(a_row + 4 * lane, dst);
  glds16<AUX>(halves_row, dst + 256);
  glds16<AUX>(b_row + 4 * lane, dst + 512);
  order_fence();
}

// both rows of the pair in the slot at byte address `base`, 6 floats per lane each
__device__ __forceinline__ void read_pair(const LaneAddr& L, unsigned base, float (&A)[6], float (&B)[6]) {
  v2 t0, t1, t2, w0, w1, w2;
  const unsigned at = L.a + base, aw0 = L.b[0] + base, aw1 = L.b[1] + base, aw2 = L.b[2] + base;
  asm volatile("ds_read_b64 %[t0], %[at]\n\t"
               "ds_read_b64 %[t1], %[at] offset:8\n\t"
               "ds_read_b64 %[t2], %[at] offset:16\n\t"
               "ds_read_b64 %[w0], %[aw0]\n\t"
               "ds_read_b64 %[w1], %[aw1]\n\t"
               "ds_read_b64 %[w2], %[aw2]\n\t"
               "s_waitcnt lgkmcnt(0)"
               : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2)
               : [at] "v"(at), [aw0] "v"(aw0), [aw1] "v"(aw1), [aw2] "v"(aw2)
               : "memory");
  A[0] = t0.x; A[1] = t0.y; A[2] = t1.x; A[3] = t1.y; A[4] = t2.x; A[5] = t2.y;
  B[0] = w0.x; B[1] = w0.y; B[2] = w1.x; B[3] = w1.y; B[4] = w2.x; B[5] = w2.y;
}

// the same in two halves, so that independent work can run under the LDS latency: the six ds_read_b64 are issued into
// `raw`, whose registers must not be touched until read_pair_finish has waited for them
struct PairRaw { v2 t0, t1, t2, w0, w1, w2; };
__device__ __forceinline__ void read_pair_issue(const LaneAddr& L, unsigned base, PairRaw& raw) {
  const unsigned at = L.a + base, aw0 = L.b[0] + base, aw1 = L.b[1] + base, aw2 = L.b[2] + base;
  asm volatile("ds_read_b64 %[t0], %[at]\n\t"
               "ds_read_b64 %[t1], %[at] offset:8\n\t"
               "ds_read_b64 %[t2], %[at] offset:16\n\t"
               "ds_read_b64 %[w0], %[aw0]\n\t"
               "ds_read_b64 %[w1], %[aw1]\n\t"
               "ds_read_b64 %[w2], %[aw2]"
               : [t0] "=&v"(raw.t0), [t1] "=&v"(raw.t1), [t2] "=&v"(raw.t2), [w0] "=&v"(raw.w0), [w1] "=&v"(raw.w1), [w2] "=&v"(raw.w2)
               : [at] "v"(at), [aw0] "v"(aw0), [aw1] "v"(aw1), [aw2] "v"(aw2)
               : "memory");
}
__device__ __forceinline__ void read_pair_finish(PairRaw& raw, float (&A)[6], float (&B)[6]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(raw.t0), "+v"(raw.t1), "+v"(raw.t2), "+v"(raw.w0), "+v"(raw.w1), "+v"(raw.w2)::"memory");
  A[0] = raw.t0.x; A[1] = raw.t0.y; A[2] = raw.t1.x; A[3] = raw.t1.y; A[4] = raw.t2.x; A[5] = raw.t2.y;
  B[0] = raw.w0.x; B[1] = raw.w0.y; B[2] = raw.w1.x; B[3] = raw.w1.y; B[4] = raw.w2.x; B[5] = raw.w2.y;
}

// six results per lane -> the output row at byte address `base` -> sixteen bytes per lane (q1: lanes 0-31 only)
__device__ __forceinline__ void transpose_out(const LaneAddr& L, unsigned base, const float (&o)[6], vfloat4& q0, vfloat4& q1) {
  const v2 p0{o[0], o[1]}, p1{o[2], o[3]}, p2{o[4], o[5]};
  const unsigned ao = L.a + base, r0 = L.r0 + base, r1 = L.r1 + base;
  asm volatile("ds_write_b64 %[ao], %[p0]\n\t"
               "ds_write_b64 %[ao], %[p1] offset:8\n\t"
               "ds_write_b64 %[ao], %[p2] offset:16\n\t"
               "ds_read_b128 %[q0], %[r0]\n\t"
               "ds_read_b128 %[q1], %[r1]\n\t"
               "s_waitcnt lgkmcnt(0)"
               : [q0] "=&v"(q0), [q1] "=&v"(q1)
               : [ao] "v"(ao), [r0] "v"(r0), [r1] "v"(r1), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2)
               : "memory");
}

// The zonal edge fluxes of a row in the edge-flux form of greb_device.h (make_flux): with e[i] = T(i+1) - T(i),
//   A[i] = w(i+1)*e[i] (Pp), i = 0..7      Bx[3+i] = w(i)*e[i] (Pm), i = -3..5
// the lane's six edges, the neighbours' by wave rotates (the rotate is the row's periodic boundary).
// Tn0 / wn0: the next lane's first point (longitude 0 for the last lane).
struct RowFlux {
  float A[8], Bx[9], Tn0, wn0;
};
__device__ __forceinline__ void row_flux(const float (&T)[6], const float (&w)[6], RowFlux& f) {
  f.Tn0 = wave_from_next(T[0]);
  f.wn0 = wave_from_next(w[0]);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float e = (i < 5 ? T[i + 1] : f.Tn0) - T[i];
    f.A[i] = (i < 5 ? w[i + 1] : f.wn0) * e;
    f.Bx[3 + i] = w[i] * e;
  }
  f.A[6] = wave_from_next(f.A[0]); f.A[7] = wave_from_next(f.A[1]);
  f.Bx[0] = wave_from_prev(f.Bx[6]); f.Bx[1] = wave_from_prev(f.Bx[7]); f.Bx[2] = wave_from_prev(f.Bx[8]);
}
// the sub-cycle clamp where(dTxh <= -T1h) dTxh = -0.9*T1h (:715, :907) for one sweep: d <= -T implies fl(T + d) <= 0,
// so the minimum of the updated values decides for the whole wavefront whether any point needs the reference's select
__device__ __forceinline__ void clamped_update(const float (&T)[6], const float (&d)[6], float (&Tn)[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) Tn[i] = T[i] + d[i];
  const float mn = min3f(min3f(Tn[0], Tn[1], Tn[2]), min3f(Tn[3], Tn[4], Tn[5]), Tn[5]);
  if (__builtin_expect(__any(!(mn > 0.f)), 0)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) Tn[i] = T[i] + ((d[i] <= -T[i]) ? -0.9f * T[i] : d[i]);
  }
}
// one diffusion sweep (dif_lon_fast), cs = ccx2/20
__device__ __forceinline__ void dif_sweep_fast(const float (&T)[6], const RowFlux& f, float cs, float (&Tn)[6]) {
  float d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float a = f.A[i] - f.Bx[3 + i - 1], b = f.A[i + 1] - f.Bx[3 + i - 2], g = f.A[i + 2] - f.Bx[3 + i - 3];
    d[i] = cs * (6.f * a + (3.f * b + g));
  }
  clamped_update(T, d, Tn);
}
// FAST sign split of a wind (src/greb.f90:203-216) as ONE instruction each: max(u, 0) / min(u, 0) instead of a compare
// and a select (the two differ only in the sign of a zero, which no FAST result keeps; STRICT uses split_m / split_p)
__device__ __forceinline__ float wind_pos(float u) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(u)); return r; }
__device__ __forceinline__ float wind_neg(float u) { float r; asm("v_min_f32 %0, 0, %1" : "=v"(r) : "v"(u)); return r; }

// one sub-cycled advection sweep (adv_lon_sub_fast, :845-851), cs = ccx2/20; last_lane: the lane whose point 3 is
// longitude xdim-2 (1-based), where the reference's index bug (:881) replaces the 4* and 1* terms
__device__ __forceinline__ void adv_sweep_fast(const float (&T)[6], const float (&u)[6], const RowFlux& f, float cs,
                                               bool last_lane, float (&Tn)[6]) {
  float d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float am = 10.f * f.Bx[3 + i - 1] + (4.f * f.Bx[3 + i - 2] + f.Bx[3 + i - 3]);
    float ap = 10.f * f.A[i] + (4.f * f.A[i + 1] + f.A[i + 2]);
    if (i == 3) { // 4*w(jp1)*(T(jp1)-T(jp1)) + w(jp3)*(T(jp1)-T(jp3)) with jp1 = xdim-1, jp3 = 1
      const float bug = 10.f * f.A[i] - f.wn0 * (T[4] - f.Tn0);
      ap = last_lane ? bug : ap;
    }
    d[i] = cs * (-wind_neg(u[i]) * ap - wind_pos(u[i]) * am);
  }
  clamped_update(T, d, Tn);
}

} // namespace rows
} // namespace greb
