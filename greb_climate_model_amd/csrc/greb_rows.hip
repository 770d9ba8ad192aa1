// greb_rows.hip -- the standalone batched diffusion sweep (src/greb.f90:556-723) on 384-wide grids as ROW STRIPS.
//
// At 384x192 every latitude row takes the sub-cycled branch (:651-719) and 34 of the 192 rows iterate (225, 82, 40 ...
// dependent Jacobi sweeps next to the poles): 1 046 row-sweeps of arithmetic against 192 rows of
// traffic.  A band-per-workgroup kernel (greb_kernels.hip: sweep_kernel) holds a band's LDS and three idle waves for as
// long as its longest chain runs; measured 0.39 of HBM peak, 43 % of the wave-cycles parked (profiles/r03_g384_*).
// Here nothing waits for anything but its own data:
//   * one WAVEFRONT = one task = a strip of consecutive rows of one field; no workgroup barrier anywhere.  Tasks are
//     launched longest first (the strip with the 225-sweep row of every field, then the 82-sweep one, ...), so the
//     dependent chains -- which issue one instruction per ~5 cycles whatever they share a SIMD with -- run beside
//     streaming strips from the first microsecond and the tail of the launch is made of the cheapest strips;
//   * a lane owns 6 consecutive longitudes of a row (64 x 6 = 384), the layout of the register-resident chains
//     (greb_chain6.h): the zonal halo is a wave rotate (DPP), never memory;
//   * a row travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: coalesced 16-byte lanes, no VGPRs, two rows ahead
//     of the arithmetic and in flight across a whole chain), is read back 6 floats per lane (ds_read_b64 x 3,
//     conflict-free: 24-byte lane stride), and the results take the reverse way (ds_write_b64 -> ds_read_b128 ->
//     global_store_dwordx4 nt).  The LDS is private to the wave: ordering is s_waitcnt only;
//   * the strip walks south to north and carries w(k)*(T(k+1)-T(k)) from row to row: each row of T and wz is read once
//     per strip (two halo rows per strip are the only re-reads), the meridional term costs 5 instructions a point.
// All LDS traffic is inline asm: the compiler must not know that LDS-DMA and the ds_ reads touch the same bytes, or it
// drains the DMA queue (s_waitcnt vmcnt(0)) in front of every read.  vmcnt is counted by hand (loads, LDS-DMA and
// stores retire in issue order): `ops` numbers every vector-memory operation the wave issues.
#include <algorithm>
#include <cstring>

#include "greb_kernels.h"
#include "greb_stencil.h"

namespace greb {
namespace {

constexpr int kRNx = 384, kRP = 6;
constexpr unsigned kRowB = kRNx * 4;              // 1 536 bytes of a row
constexpr unsigned kSlotB = 2 * kRowB;            // a (T, wz) row pair in LDS: T row | wz tail (512 B) | wz head (1 024 B)
constexpr unsigned kRowsLdsB = 2 * kSlotB + kRowB; // two slots + the output row
constexpr int kMaxStrips = 64;

// Everything the kernel needs to know about the rows travels BY VALUE (kernarg, scalar loads): no device-side table
// whose lifetime or contents a concurrent call could disturb.
struct RowsArgs {
  float ccy;                       // kappa*dt_crcl/dyy**2, :581
  float cc[kMaxNy];                // ccx2 of the row, :654
  int time2[kMaxNy];               // sweeps of the row, :653 (dwords: a 16-bit table is read by VECTOR loads, and the
                                   // wait for one drains the LDS-DMA queue)
  int n_strips;
  int k0[kMaxStrips], k1[kMaxStrips]; // strip s updates rows [k0, k1); sorted by cost, dearest first
};

typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;

template <int AUX>
__device__ __forceinline__ void glds16(const float* g, lfloat* l) {
  __builtin_amdgcn_global_load_lds((gvoid*)g, (lvoid*)l, 16, 0, AUX);
}

#define GREB_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

struct Walk {
  const float* Tf;   // this field's T1
  const float* wf;   // ... wz
  const float* p2;   // per lane: the second halves of the T row (lanes 0-31) and of the wz row (lanes 32-63)
  float* of;         // ... dX
  lfloat* lds;
  unsigned lane;
  unsigned aT, aW[3], aO, aR0, aR1; // LDS byte addresses of this lane (slot 0)
  int last_row;      // the last row any step reads
  int ops;           // vector-memory operations issued so far
  int gend[2];       // `ops` right after the LDS-DMA of the row now in slot s was issued
};

template <int AUX, unsigned SLOT>
__device__ __forceinline__ void issue_row(Walk& c, int x) {
  const int ro = x * kRNx;
  glds16<AUX>(c.Tf + ro + 4 * c.lane, c.lds + SLOT * (kSlotB / 4));
  glds16<AUX>(c.p2 + ro, c.lds + SLOT * (kSlotB / 4) + 256);
  glds16<AUX>(c.wf + ro + 4 * c.lane, c.lds + SLOT * (kSlotB / 4) + 512);
  c.ops += 3;
  c.gend[SLOT] = c.ops;
}

// the row in slot SLOT has landed: all but the `younger` operations issued after its LDS-DMA may still be in flight
template <unsigned SLOT>
__device__ __forceinline__ void wait_row(const Walk& c) {
  const int younger = c.ops - c.gend[SLOT];
  if (younger >= 7) GREB_VMCNT(7);
  else if (younger >= 5) GREB_VMCNT(5);
  else if (younger >= 3) GREB_VMCNT(3);
  else if (younger >= 2) GREB_VMCNT(2);
  else GREB_VMCNT(0);
}

template <unsigned SLOT>
__device__ __forceinline__ void read_row(const Walk& c, float (&T)[6], float (&w)[6]) {
  v2 t0, t1, t2, w0, w1, w2;
  asm volatile("ds_read_b64 %[t0], %[at] offset:%[o0]\n\t"
               "ds_read_b64 %[t1], %[at] offset:%[o1]\n\t"
               "ds_read_b64 %[t2], %[at] offset:%[o2]\n\t"
               "ds_read_b64 %[w0], %[aw0] offset:%[o0]\n\t"
               "ds_read_b64 %[w1], %[aw1] offset:%[o0]\n\t"
               "ds_read_b64 %[w2], %[aw2] offset:%[o0]\n\t"
               "s_waitcnt lgkmcnt(0)"
               : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2)
               : [at] "v"(c.aT), [aw0] "v"(c.aW[0]), [aw1] "v"(c.aW[1]), [aw2] "v"(c.aW[2]), [o0] "i"(SLOT * kSlotB),
                 [o1] "i"(SLOT * kSlotB + 8), [o2] "i"(SLOT * kSlotB + 16)
               : "memory");
  T[0] = t0.x; T[1] = t0.y; T[2] = t1.x; T[3] = t1.y; T[4] = t2.x; T[5] = t2.y;
  w[0] = w0.x; w[1] = w0.y; w[2] = w1.x; w[3] = w1.y; w[4] = w2.x; w[5] = w2.y;
}

// six results per lane -> LDS -> sixteen bytes per lane -> HBM (two stores: 64 + 32 quads)
__device__ __forceinline__ void store_row(Walk& c, const float (&o)[6], int k) {
  const v2 p0{o[0], o[1]}, p1{o[2], o[3]}, p2{o[4], o[5]};
  vfloat4 q0, q1;
  asm volatile("ds_write_b64 %[ao], %[p0] offset:%[o0]\n\t"
               "ds_write_b64 %[ao], %[p1] offset:%[o1]\n\t"
               "ds_write_b64 %[ao], %[p2] offset:%[o2]\n\t"
               "ds_read_b128 %[q0], %[r0] offset:%[o0]\n\t"
               "ds_read_b128 %[q1], %[r1] offset:%[o0]\n\t"
               "s_waitcnt lgkmcnt(0)"
               : [q0] "=&v"(q0), [q1] "=&v"(q1)
               : [ao] "v"(c.aO), [r0] "v"(c.aR0), [r1] "v"(c.aR1), [p0] "v"(p0), [p1] "v"(p1), [p2] "v"(p2),
                 [o0] "i"(2 * kSlotB), [o1] "i"(2 * kSlotB + 8), [o2] "i"(2 * kSlotB + 16)
               : "memory");
  float* row = c.of + k * kRNx;
  __builtin_nontemporal_store(q0, reinterpret_cast<vfloat4*>(row + 4 * c.lane));
  if (c.lane < 32) __builtin_nontemporal_store(q1, reinterpret_cast<vfloat4*>(row + 256 + 4 * c.lane));
  c.ops += 2;
}

// one zonal sweep of a row that does not iterate (time2 = 1), FAST arithmetic: the edge-flux form of greb_device.h
// (dif_lon_fast) with the lane's six edges e[i] = T(i+1) - T(i), the neighbours' fluxes by wave rotates
__device__ __forceinline__ void single_sweep_fast(const float (&T)[6], const float (&w)[6], float cs, float (&Tn)[6]) {
  float A[8], Bx[9]; // A[i] = w(i+1)*e[i], i = 0..7;  Bx[3+i] = w(i)*e[i], i = -3..5
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float e = (i < 5 ? T[i + 1] : wave_from_next(T[0])) - T[i];
    A[i] = (i < 5 ? w[i + 1] : wave_from_next(w[0])) * e;
    Bx[3 + i] = w[i] * e;
  }
  A[6] = wave_from_next(A[0]); A[7] = wave_from_next(A[1]);
  Bx[0] = wave_from_prev(Bx[6]); Bx[1] = wave_from_prev(Bx[7]); Bx[2] = wave_from_prev(Bx[8]);
  float d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float a = A[i] - Bx[3 + i - 1], b = A[i + 1] - Bx[3 + i - 2], g = A[i + 2] - Bx[3 + i - 3];
    d[i] = cs * (6.f * a + (3.f * b + g));
    Tn[i] = T[i] + d[i];
  }
  // the clamp where(dTxh <= -T1h) dTxh = -0.9*T1h (:715): d <= -T implies fl(T + d) <= 0, so the minimum of the
  // updated values decides for the whole wavefront whether any point needs the reference's select
  const float mn = min3f(min3f(Tn[0], Tn[1], Tn[2]), min3f(Tn[3], Tn[4], Tn[5]), Tn[5]);
  if (__builtin_expect(__any(!(mn > 0.f)), 0)) {
#pragma unroll
    for (int i = 0; i < 6; ++i) Tn[i] = T[i] + ((d[i] <= -T[i]) ? -0.9f * T[i] : d[i]);
  }
}

template <bool STRICT>
struct RowState {
  float T[6], w[6];
  float q[6];   // FAST: w(k-1)*(T(k) - T(k-1));  STRICT: T of row k-1
  float wm[6];  // STRICT: wz of row k-1
};

template <bool STRICT, int AUX, unsigned SLOT /* the slot the NEXT row arrives in */>
__device__ __forceinline__ void row_step(Walk& c, const RowsArgs& a, RowState<STRICT>& cur, RowState<STRICT>& nxt, int r,
                                         int k0, int ny) {
  const int n = r + 1;
  if (n <= c.last_row) {
    wait_row<SLOT>(c);
    read_row<SLOT>(c, nxt.T, nxt.w);
    if (n + 2 <= c.last_row) issue_row<AUX, SLOT>(c, n + 2);
  } else { // no row above the last one: its weight is zero
#pragma unroll
    for (int i = 0; i < 6; ++i) { nxt.T[i] = cur.T[i]; nxt.w[i] = 0.f; }
  }
  float P[6];
  if (!STRICT) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float h = nxt.T[i] - cur.T[i];
      P[i] = nxt.w[i] * h;
      nxt.q[i] = cur.w[i] * h;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) { nxt.q[i] = cur.T[i]; nxt.wm[i] = cur.w[i]; }
  }
  if (r < k0) return; // the strip's lower halo row: nothing to write
  const int t2 = a.time2[r];
  const float cc = a.cc[r];
  float T1h[6];
  if (STRICT || t2 > 1) {
    float Tw[12], ww[12];
    const float u0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 6; ++i) { Tw[3 + i] = cur.T[i]; ww[3 + i] = cur.w[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Tw[i] = wave_from_prev(cur.T[3 + i]); Tw[9 + i] = wave_from_next(cur.T[i]);
      ww[i] = wave_from_prev(cur.w[3 + i]); ww[9 + i] = wave_from_next(cur.w[i]);
    }
    chain_window<STRICT, 6>(Tw, ww, u0, cc, t2, false, (int)c.lane);
#pragma unroll
    for (int i = 0; i < 6; ++i) T1h[i] = Tw[3 + i];
  } else {
    single_sweep_fast(cur.T, cur.w, cc * 0.05f, T1h);
  }
  float o[6];
  if (!STRICT) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float dTx = T1h[i] - cur.T[i]; // fl(fl(T + d) - T), :718
      float g;
      {
#pragma clang fp contract(off)
        g = P[i] - cur.q[i]; // two rounded products: the same value whichever way a strip would walk
      }
      o[i] = cur.w[i] * (dTx + a.ccy * g);
    }
  } else {
#pragma clang fp contract(off)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const float dTx = T1h[i] - cur.T[i];
      float dTy; // :585-590
      if (r >= 1 && r <= ny - 2) dTy = a.ccy * (cur.wm[i] * (cur.q[i] - cur.T[i]) + nxt.w[i] * (nxt.T[i] - cur.T[i]));
      else if (r == 0) dTy = a.ccy * nxt.w[i] * (-cur.T[i] + nxt.T[i]);
      else dTy = a.ccy * cur.wm[i] * (cur.q[i] - cur.T[i]);
      o[i] = cur.w[i] * (dTx + dTy); // :721
    }
  }
  store_row(c, o, r);
}

template <bool STRICT, int AUX>
__global__ __launch_bounds__(64) void dif_rows_kernel(const float* __restrict__ T1, const float* __restrict__ wz,
                                                       float* __restrict__ dX, const RowsArgs a, int batch, int ny) {
  extern __shared__ __align__(16) float lds_raw[];
  Walk c;
  c.lds = (lfloat*)lds_raw;
  c.lane = threadIdx.x;
  const int s = blockIdx.x / batch, b = blockIdx.x - s * batch;
  const int k0 = a.k0[s], k1 = a.k1[s];
  const size_t fo = (size_t)b * kRNx * ny;
  c.Tf = T1 + fo; c.wf = wz + fo; c.of = dX + fo;
  c.p2 = c.lane < 32 ? c.Tf + 256 + 4 * c.lane : c.wf + 256 + 4 * (c.lane - 32);
  const unsigned lb = (unsigned)(size_t)c.lds;
  c.aT = lb + 24 * c.lane;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const unsigned x = 24 * c.lane + 8 * j;
    c.aW[j] = lb + (x < 1024 ? 2048 + x : 512 + x);
  }
  c.aO = lb + 24 * c.lane;
  c.aR0 = lb + 16 * c.lane;
  c.aR1 = lb + (c.lane < 32 ? 1024 + 16 * c.lane : 0);
  c.ops = 0; c.gend[0] = c.gend[1] = 0;
  const int a0 = k0 > 0 ? k0 - 1 : 0;        // first row read
  c.last_row = k1 < ny ? k1 : ny - 1;
  issue_row<AUX, 0>(c, a0);
  if (a0 + 1 <= c.last_row) issue_row<AUX, 1>(c, a0 + 1);
  RowState<STRICT> A, B;
  wait_row<0>(c);
  read_row<0>(c, A.T, A.w);
  if (a0 + 2 <= c.last_row) issue_row<AUX, 0>(c, a0 + 2);
#pragma unroll
  for (int i = 0; i < 6; ++i) { A.q[i] = STRICT ? A.T[i] : 0.f; A.wm[i] = 0.f; }
  for (int r = a0; r < k1; r += 2) {
    row_step<STRICT, AUX, 1>(c, a, A, B, r, k0, ny);
    if (r + 1 >= k1) break;
    row_step<STRICT, AUX, 0>(c, a, B, A, r + 1, k0, ny);
  }
}

int strip_cost(int time2) { return time2 > 1 ? 130 + 36 * time2 : 120; }

} // namespace

bool rows_plan(const RowTables& t, int ny, int target_cost, int& n_strips, int* k0, int* k1) {
  // contiguous strips of about `target_cost` instructions, never splitting a row; a strip is closed when the next row
  // would take it over the target, unless it is still tiny
  struct S { int k0, k1, cost; };
  S st[kMaxNy];
  int n = 0, acc = 0, start = 0;
  for (int k = 0; k < ny; ++k) {
    const int cst = strip_cost(t.dif_time2[k]);
    if (acc > 0 && acc + cst > target_cost && acc >= 600) { st[n++] = {start, k, acc}; start = k; acc = 0; }
    acc += cst;
  }
  st[n++] = {start, ny, acc};
  if (n > kMaxStrips) return false;
  std::stable_sort(st, st + n, [](const S& x, const S& y) { return x.cost > y.cost; });
  n_strips = n;
  for (int i = 0; i < n; ++i) { k0[i] = st[i].k0; k1[i] = st[i].k1; }
  return true;
}

bool rows_supported(const RowTables& t, int nx, int ny) {
  if (nx != kRNx || ny < 3 || ny > kMaxNy) return false;
  for (int k = 0; k < ny; ++k)
    if (!t.subcycled[k] || t.dif_time2[k] < 1 || t.dif_time2[k] > (1 << 20)) return false;
  return true;
}

hipError_t launch_diffusion_rows(const float* T1, const float* wz, float* dX, const RowTables& t, int ny, int batch,
                                 bool strict, hipStream_t s) {
  RowsArgs a;
  std::memset(&a, 0, sizeof(a));
  a.ccy = t.dif_ccy;
  for (int k = 0; k < ny; ++k) { a.cc[k] = t.dif_ccx2[k]; a.time2[k] = t.dif_time2[k]; }
  static const int target = tuning_int("GREB_ROWS_TARGET", 4000); // -DGREB_TUNING builds only
  int n = 0;
  if (!rows_plan(t, ny, target, n, a.k0, a.k1)) return hipErrorInvalidValue;
  a.n_strips = n;
  if ((long long)n * batch > 0x7fffffffLL) return hipErrorInvalidValue;
  static const int aux = tuning_int("GREB_ROWS_NT", 0);
  void (*kern)(const float*, const float*, float*, const RowsArgs, int, int);
  if (strict) kern = dif_rows_kernel<true, 0>;
  else kern = aux ? dif_rows_kernel<false, 2> : dif_rows_kernel<false, 0>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(n * batch)), dim3(64), kRowsLdsB, s, T1, wz, dX, a, batch, ny);
  return hipGetLastError();
}

} // namespace greb
