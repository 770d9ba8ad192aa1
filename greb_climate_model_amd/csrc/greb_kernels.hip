// greb_kernels.hip -- batched single-routine HIP kernels of the GREB hot path (gfx950, wave64).
//
//   sweep_kernel<.,dif>  standalone batched diffusion sweep (src/greb.f90:556-723): the kernel the
//                        HBM-roofline metric is defined on (12 B/point: read T1, read wz, write dX)
//   sweep_kernel<.,adv>  its advection twin (src/greb.f90:726-915)
//   point_kernel         a4-a8 for one step (test mirror of the engine's point physics)
//   launch_circulation   24 sub-steps: 96x48 -> the fused LDS loop of the engine (greb_member.hip),
//                        other grids -> one launch per operator per sub-step
//
// Any grid with nx % 4 == 0: a workgroup stages a latitude band (+halo rows) of T and wz in LDS
// with coalesced dwordx4 loads, sweeps it, and streams dX back as dwordx4.  No MFMA: there is no
// dense contraction in this model.
#include <cstdlib>

#include "greb_kernels.h"
#include "greb_physics_step.h"
#include "greb_stencil.h"

namespace greb {

template <bool STRICT, int MODE>
__global__ __launch_bounds__(256) void sweep_kernel(const float* __restrict__ T1, const float* __restrict__ wz,
                                                    const float* __restrict__ ug, const float* __restrict__ vg,
                                                    float* __restrict__ dX, const RowTables* __restrict__ tabp,
                                                    int nx, int ny, int rows_per_band, int wmod, int uv_shared,
                                                    const int* __restrict__ tab_index, int tab_div, int calm_odd) {
  // batch item b: field T1[b]; weights wz[b % wmod] (wmod = 0: wz[b]); winds u,v[b] or shared;
  // row tables tabp[tab_index[b / tab_div]] (tab_index = nullptr: tabp[0]);
  // calm_odd: odd batch items (the vapour fields of the engine) see zero wind -- the experiment that diffuses
  // but does not advect q (greb.original.model.f90:560-564): every advective increment is then exactly 0
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  const int b = blockIdx.x;
  const RowTables& tab = tabp[tab_index ? tab_index[b / tab_div] : 0];
  const int nq = nx >> 2;
  // bands are handed out from the poles inwards (0, last, 1, last-1, ...): the polar bands carry the long Jacobi
  // chains, so they must start first or they become the tail of the launch
  const int nbands = gridDim.y;
  const int band = (blockIdx.y & 1) ? nbands - 1 - (blockIdx.y >> 1) : (blockIdx.y >> 1);
  const int k0 = band * rows_per_band;
  const int k1 = min(ny, k0 + rows_per_band);
  constexpr int halo = MODE == kChainDif ? 1 : 2;
  const int r0 = max(0, k0 - halo), r1 = min(ny, k1 + halo);
  const int nrows = r1 - r0;
  lfloat* sT = lds;
  lfloat* sW = sT + nrows * nx;
  constexpr bool kWinds = MODE != kChainDif;
  lfloat* sU = sW + nrows * nx; // with winds: band rows k0..k1
  lfloat* sV = sU + (kWinds ? (k1 - k0) * nx : 0);
  // per wave: the chain rows' sub-cycled results (greb_stencil.h: chain_row_scratch); the band's row constants
  const int per_wave = chain_row_scratch(nx);
  lfloat* scratch = sV + (kWinds ? (k1 - k0) * nx : 0);
  lfloat* rowk_band = scratch + 4 * per_wave;                       // [rows_per_band][kRowKWords]
  const lfloat* rowk = rowk_band - k0 * kRowKWords;                 // indexed by the absolute row
  const size_t fo = (size_t)b * nx * ny;
  const size_t fw = (size_t)(wmod ? b % wmod : b) * nx * ny;
  const size_t fu = uv_shared ? 0 : fo;
  // Staging.  Everything the band needs from global memory is requested before anything is waited for: unconditional
  // loads (clamped indices) in batches of kIt per thread, only the LDS stores predicated.  (Written as a plain
  // `for (...) st4(lds, ld4(global))` loop this was one dependent round trip per iteration, and a load under a condition
  // is preceded by s_waitcnt vmcnt(0): six to seven round trips ahead of the 225-sweep chain that sets the length of the
  // 384x192 launch.)  The row tables hang off tab_index[]: they are requested after the first batch is on its way.
  constexpr int kIt = 4;
  const int nT = nrows * nq, nU = kWinds ? (k1 - k0) * nq : 0, bd = blockDim.x, tid = threadIdx.x;
  const bool calm = calm_odd && (b & 1);
  for (int base = 0; base < max(nT, nU); base += kIt * bd) {
    f4 a[kIt], w[kIt], uq[kIt], vq[kIt];
#pragma unroll
    for (int j = 0; j < kIt; ++j) {
      const int i = min(base + j * bd + tid, nT - 1);
      a[j] = ld4(T1 + fo + (size_t)r0 * nx + 4 * i);
      w[j] = ld4(wz + fw + (size_t)r0 * nx + 4 * i);
    }
    if (kWinds && base < nU) {
#pragma unroll
      for (int j = 0; j < kIt; ++j) {
        const int i = min(base + j * bd + tid, nU - 1);
        uq[j] = ld4(ug + fu + (size_t)k0 * nx + 4 * i);
        vq[j] = ld4(vg + fu + (size_t)k0 * nx + 4 * i);
      }
    }
    if (base == 0) {
      __builtin_amdgcn_sched_barrier(0);
      stage_row_consts(rowk_band, tab, k0, k1);
    }
#pragma unroll
    for (int j = 0; j < kIt; ++j) {
      const int i = base + j * bd + tid;
      if (i < nT) { st4(sT + 4 * i, a[j]); st4(sW + 4 * i, w[j]); }
    }
    if (kWinds && base < nU) {
#pragma unroll
      for (int j = 0; j < kIt; ++j) {
        const int i = base + j * bd + tid;
        if (i < nU) { st4(sU + 4 * i, calm ? zero4() : uq[j]); st4(sV + 4 * i, calm ? zero4() : vq[j]); }
      }
    }
  }
  __syncthreads();
  const Rows X{sT, r0, nx}, W{sW, r0, nx}, U{sU, k0, nx}, V{sV, k0, nx};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  // chain rows: one wave each, round-robin
  int ci = 0;
  for (int k = k0; k < k1; ++k) {
    const RowK rk = row_consts((const lfloat*)rowk, k);
    if (!is_chain_row(rk, MODE)) continue;
    if ((ci++ % nwaves) == wave)
      chain_row<STRICT>(X, W, U, V, rk, k, nq, ny, lane, MODE, scratch + wave * per_wave, dX + fo + (size_t)k * nx);
  }
  // everything else: quads, all threads
  for (int i = threadIdx.x; i < (k1 - k0) * nq; i += blockDim.x) {
    const int k = k0 + i / nq, q = i % nq;
    const RowK rk = row_consts((const lfloat*)rowk, k);
    if (is_chain_row(rk, MODE)) continue;
    QuadIn in;
    gather(X, W, k, q, nq, ny, MODE != kChainDif, in);
    float out[4];
    if (MODE == kChainDif) {
      dif_quad<STRICT>(in, rk, k, ny, out);
    } else if (MODE == kChainAdv) {
      const f4 uq = ld4(U.row(k) + 4 * q), vq = ld4(V.row(k) + 4 * q);
      adv_quad<STRICT>(in, uq.v, vq.v, rk, k, ny, q == nq - 1, out);
    } else { // fused sub-step: X_new = (X + dX_diffuse) + dX_advec, src/greb.f90:549
      const f4 uq = ld4(U.row(k) + 4 * q), vq = ld4(V.row(k) + 4 * q);
      float dd[4], da[4];
      dif_quad<STRICT>(in, rk, k, ny, dd);
      adv_quad<STRICT>(in, uq.v, vq.v, rk, k, ny, q == nq - 1, da);
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) out[i2] = in.T[4 + i2] + dd[i2] + da[i2];
      }
    }
    st4(dX + fo + (size_t)k * nx + 4 * q, f4{{out[0], out[1], out[2], out[3]}});
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming form of the diffusion sweep for the 96x48 grid (whole field = 2 x 18 KB of LDS).
//   * persistent workgroups walk the batch; while field b is swept out of LDS the loads of the
//     workgroup's next field are already in flight into registers (9 dwordx4 per thread), so HBM
//     latency overlaps the stencil instead of being paid once per field;
//   * waves 0-2: 192 threads, one 4 lon x 6 lat tile each, sliding a 3-row window down the tile
//     (6 ds_read_b128 per row instead of 14);
//   * wave 3: the polar (chain) rows, two rows side by side in the wave (lanes 0-23 / 32-55), 8
//     dependent Jacobi sweeps each (src/greb.f90:656-717);
//   * the compute phase issues no global load (row constants are staged in LDS): s_waitcnt vmcnt
//     is in-order, one table load would wait for the whole prefetch.
//   * loads and stores are non-temporal (each byte is touched once per launch): measured 5.1-5.2 -> 5.6 TB/s,
//     above the box's torch copy rate.
// Traffic = the algorithmic 12 B/point exactly (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE).
// ---------------------------------------------------------------------------------------------
constexpr int kStreamThreads = 256;
constexpr int kStreamQPT = 5; // quads per thread and array: covers 1280 quads (96x48 has 1152)
constexpr int kTileRows = 6;

template <bool STRICT, int NX_, int NY_>
__global__ __launch_bounds__(kStreamThreads) void diffusion_stream_kernel(const float* __restrict__ T1,
                                                                          const float* __restrict__ wz,
                                                                          float* __restrict__ dX,
                                                                          const RowTables* __restrict__ tabp,
                                                                          int batch, int dbg) {
  constexpr int nx = NX_, ny = NY_, nq = NX_ / 4, nquad = NY_ * nq;
  static_assert(NY_ % kTileRows == 0 && nq * (NY_ / kTileRows) == 192, "tile map is for 96x48");
  extern __shared__ __align__(16) float lds_raw[];
  lfloat* lds = (lfloat*)lds_raw;
  lfloat* sT = lds;
  lfloat* sW = sT + ny * nx;
  lfloat* scratch = sW + ny * nx;      // [2 halves][2*nx]: diffusion chains use two row buffers each
  lfloat* rowk = scratch + 2 * 2 * nx; // [ny][kRowKWords]
  lfloat* chainlist = rowk + ny * kRowKWords; // [ny] row indices of the chain rows
  stage_row_consts(rowk, *tabp, 0, ny);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int nchain = 0; // uniform: every thread scans the (global, cached) table once
  for (int kk = 0; kk < ny; ++kk)
    if (is_chain_row(*tabp, kk, kChainDif)) {
      if (tid == 0) chainlist[nchain] = __int_as_float(kk);
      ++nchain;
    }
  const Rows X{sT, 0, nx}, W{sW, 0, nx};
  f4 rT[kStreamQPT], rW[kStreamQPT];
  int b = blockIdx.x;
  if (b < batch) {
    const size_t fo = (size_t)b * nx * ny;
#pragma unroll
    for (int j = 0; j < kStreamQPT; ++j) {
      const int i = tid + j * kStreamThreads;
      if (i < nquad) { rT[j] = ld4_nt(T1 + fo + 4 * i); rW[j] = ld4_nt(wz + fo + 4 * i); }
    }
  }
  // static role data
  const int tq = tid % nq, tg = tid / nq;          // bulk: quad column, row group (tid < 192)
  const int k0 = tg * kTileRows;
  const int qm = tq == 0 ? nq - 1 : tq - 1, qp = tq == nq - 1 ? 0 : tq + 1;
  for (; b < batch; b += gridDim.x) {
#pragma unroll
    for (int j = 0; j < kStreamQPT; ++j) {
      const int i = tid + j * kStreamThreads;
      if (i < nquad) { st4(sT + 4 * i, rT[j]); st4(sW + 4 * i, rW[j]); }
    }
    __syncthreads();
    const int nb = b + gridDim.x;
    if (nb < batch) { // prefetch the next field; consumed at the top of the next iteration
      const size_t fo = (size_t)nb * nx * ny;
#pragma unroll
      for (int j = 0; j < kStreamQPT; ++j) {
        const int i = tid + j * kStreamThreads;
        if (i < nquad) { rT[j] = ld4_nt(T1 + fo + 4 * i); rW[j] = ld4_nt(wz + fo + 4 * i); }
      }
    }
    float* out = dX + (size_t)b * nx * ny;
    if (wave == 3) {
      // chain rows, two at a time (lanes 0-31 / 32-63); the list was built once at kernel start
      if (!(dbg & 4)) {
        const int half = lane >> 5, ql = lane & 31;
        for (int c = 0; c < nchain; c += 2) {
          const int ci = c + half;
          if (ci < nchain) {
            const int k = __float_as_int(chainlist[ci]);
            chain_row<STRICT>(X, W, X, X, row_consts((const lfloat*)rowk, k), k, nq, ny, ql, kChainDif,
                              scratch + half * 2 * nx, out + (size_t)k * nx);
          }
        }
      }
    } else if (!(dbg & 1)) {
      // bulk tile: rows k0 .. k0+5, sliding window (T and w): C[0..2] = rows k-1, k, k+1
      f4 CT[3], CW[3];
      {
        const int km = k0 > 0 ? k0 - 1 : 0;
        CT[0] = ld4(sT + km * nx + 4 * tq); CW[0] = k0 > 0 ? ld4(sW + km * nx + 4 * tq) : zero4();
        CT[1] = ld4(sT + k0 * nx + 4 * tq); CW[1] = ld4(sW + k0 * nx + 4 * tq);
      }
#pragma unroll
      for (int r = 0; r < kTileRows; ++r) {
        const int k = k0 + r;
        const int kp = k < ny - 1 ? k + 1 : k;
        CT[2] = ld4(sT + kp * nx + 4 * tq);
        CW[2] = k < ny - 1 ? ld4(sW + kp * nx + 4 * tq) : zero4();
        const RowK rk = row_consts((const lfloat*)rowk, k);
        if (!is_chain_row(rk, kChainDif)) {
          const f4 LT = ld4(sT + k * nx + 4 * qm), RT = ld4(sT + k * nx + 4 * qp);
          const f4 LW = ld4(sW + k * nx + 4 * qm), RW = ld4(sW + k * nx + 4 * qp);
          QuadIn in;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            in.T[j] = LT.v[j]; in.T[4 + j] = CT[1].v[j]; in.T[8 + j] = RT.v[j];
            in.w[j] = LW.v[j]; in.w[4 + j] = CW[1].v[j]; in.w[8 + j] = RW.v[j];
          }
          in.T0 = CT[1]; in.w0 = CW[1];
          in.Tm1 = CT[0]; in.Tp1 = CT[2]; in.wm1 = CW[0]; in.wp1 = CW[2];
          in.Tm2 = CT[1]; in.Tp2 = CT[1]; in.wm2 = zero4(); in.wp2 = zero4();
          float o[4];
          dif_quad<STRICT>(in, rk, k, ny, o);
          st4_nt(out + (size_t)k * nx + 4 * tq, f4{{o[0], o[1], o[2], o[3]}});
        }
        CT[0] = CT[1]; CW[0] = CW[1]; CT[1] = CT[2]; CW[1] = CW[2];
      }
    }
    __syncthreads(); // everyone is done reading LDS before the next field overwrites it
  }
}

static size_t stream_lds_bytes(int nx, int ny) { return (size_t)(2 * nx * ny + 2 * 2 * nx + ny * kRowKWords + ny) * sizeof(float); }
static bool stream_fits(int nx, int ny) { return nx == 96 && ny == 48; }

static size_t sweep_lds_bytes(int nx, int rows, int halo, int extra_fields) {
  return (size_t)(((rows + 2 * halo) * 2 + rows * extra_fields) * nx + 4 * chain_row_scratch(nx) + rows * kRowKWords) * sizeof(float);
}
static int pick_band_rows(int nx, int ny, int halo, int extra_fields) {
  // whole field per workgroup when it fits ~44 KB (3+ workgroups share a CU's 160 KB); wide
  // grids get latitude bands of <= 64 KB
  size_t limit = nx <= 128 ? 44 * 1024 : 48 * 1024;
  limit = (size_t)tuning_int("GREB_BAND_LIMIT_KB", (int)(limit / 1024)) * 1024; // -DGREB_TUNING builds only
  int rows = ny;
  while (rows > 2 && sweep_lds_bytes(nx, rows, halo, extra_fields) > limit) rows = (rows + 1) / 2;
  return rows;
}

template <int MODE>
static hipError_t launch_sweep(const float* T1, const float* wz, const float* u, const float* v, float* dX,
                               const RowTables* tab_dev, int nx, int ny, int batch, bool strict, hipStream_t s,
                               int wmod = 0, int uv_shared = 0, const int* tab_index = nullptr, int tab_div = 1,
                               int calm_odd = 0) {
  constexpr int halo = MODE == kChainDif ? 1 : 2;
  constexpr int extra = MODE == kChainDif ? 0 : 2;
  const int rows = pick_band_rows(nx, ny, halo, extra);
  const int bands = (ny + rows - 1) / rows;
  const size_t lds = sweep_lds_bytes(nx, rows, halo, extra);
  dim3 grid(batch, bands), block(256);
  auto kern = strict ? sweep_kernel<true, MODE> : sweep_kernel<false, MODE>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, grid, block, lds, s, T1, wz, u, v, dX, tab_dev, nx, ny, rows, wmod, uv_shared, tab_index, tab_div, calm_odd);
  return hipGetLastError();
}

hipError_t launch_diffusion(const float* T1, const float* wz, float* dX, const RowTables* tab_dev,
                            const RowTables& tab_host, int nx, int ny, int batch, bool strict, hipStream_t s) {
  static const bool no_rows = tuning_int("GREB_NO_ROWS", 0) != 0; // -DGREB_TUNING builds only (A/B experiments)
  if (!no_rows && rows_supported(tab_host, nx, ny)) return launch_diffusion_rows(T1, wz, dX, tab_host, ny, batch, strict, s);
  static const bool no_stream = tuning_int("GREB_NO_STREAM", 0) != 0; // -DGREB_TUNING builds only (A/B experiments)
  if (stream_fits(nx, ny) && !no_stream) {
    const size_t lds = stream_lds_bytes(nx, ny);
    auto kern = strict ? diffusion_stream_kernel<true, 96, 48> : diffusion_stream_kernel<false, 96, 48>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    // workgroups per CU: 2 measured 5.46-5.64 TB/s against 5.34-5.45 with 3 and 3.9 with 1; a second field pair in
    // flight per workgroup (two register sets, 144 KB outstanding per CU) measured no better (5.31-5.39): the kernel
    // runs at the box's HBM rate (1.02-1.05 x its float4 copy), not at a latency limit
    static const int wg_per_cu = tuning_int("GREB_STREAM_WGS", 2);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = batch < cus * wg_per_cu ? batch : cus * wg_per_cu;
    static const int dbg = tuning_int("GREB_DEBUG_SKIP", 0); // -DGREB_TUNING builds only
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kStreamThreads), lds, s, T1, wz, dX, tab_dev, batch, dbg);
    return hipGetLastError();
  }
  return launch_sweep<kChainDif>(T1, wz, nullptr, nullptr, dX, tab_dev, nx, ny, batch, strict, s);
}
hipError_t launch_advection(const float* T1, const float* wz, const float* u, const float* v, float* dX,
                            const RowTables* tab_dev, int nx, int ny, int batch, bool strict, hipStream_t s) {
  return launch_sweep<kChainAdv>(T1, wz, u, v, dX, tab_dev, nx, ny, batch, strict, s);
}

__global__ void axpy2_kernel(float* __restrict__ X, const float* __restrict__ a, const float* __restrict__ b, size_t n) {
#pragma clang fp contract(off)
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    X[i] = X[i] + a[i] + b[i]; // :549
}
__global__ void sub_kernel(float* __restrict__ d, const float* __restrict__ a, const float* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    d[i] = a[i] - b[i]; // :551
}

hipError_t launch_circulation(const float* X, const float* wz, const float* u, const float* v, float* dX,
                              float* scratch, const RowTables* tab_dev, const RowTables& tab_host, int nx, int ny,
                              int batch, int nsub, bool strict, hipStream_t s) {
  if (member_layout_supported(tab_host, nx, ny))
    return launch_circulation_g96(X, wz, u, v, dX, tab_dev, batch, nsub, strict, s);
  // other grids: the member does not fit one CU's LDS -> one launch per operator per sub-step
  const size_t n = (size_t)batch * nx * ny;
  float *Xc = scratch, *dd = scratch + n, *da = scratch + 2 * n;
  hipError_t e = hipMemcpyAsync(Xc, X, n * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return e;
  for (int tt = 0; tt < nsub; ++tt) {
    if ((e = launch_diffusion(Xc, wz, dd, tab_dev, tab_host, nx, ny, batch, strict, s)) != hipSuccess) return e;
    if ((e = launch_advection(Xc, wz, u, v, da, tab_dev, nx, ny, batch, strict, s)) != hipSuccess) return e;
    hipLaunchKernelGGL(axpy2_kernel, dim3(1024), dim3(256), 0, s, Xc, dd, da, n);
  }
  hipLaunchKernelGGL(sub_kernel, dim3(1024), dim3(256), 0, s, dX, Xc, X, n);
  return hipGetLastError();
}

// ============================================================================================
// Point physics of one step (test mirror of a4-a8)
// ============================================================================================
__global__ void point_kernel(PointArgs a) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.np) return;
  const size_t off = (size_t)(a.ityr - 1) * a.np, offm = (size_t)(a.ityr > 1 ? a.ityr - 2 : kNT - 1) * a.np;
  const float Ts = a.in5[p], Ta = a.in5[a.np + p], To = a.in5[2 * a.np + p], q = a.in5[3 * a.np + p];
  const float cap = a.in5[4 * a.np + p];
  const float zt = a.z_topo[p], gl = a.glacier[p], ez = a.wz_air[p];
  const float cld = a.cldclim[off + p], mld = a.mldclim[off + p];
  float albedo, sw, LWsurf, LWdown, em, Qlat, Qlat_air, dq_eva, dq_rain, dT_ocean, dTo;
  sw_radiation(a.phys, Ts, zt, gl, cld, a.sw_solar[(size_t)(a.ityr - 1) * a.ny + p / a.nx], albedo, sw, a.xsw);
  lw_radiation(a.phys, Ts, Ta, q, a.co2, ez, cld, a.tclim[off + p], LWsurf, LWdown, em, a.xsw, a.qclim[off + p]);
  hydro(a.phys, Ts, q, a.uclim[off + p], a.vclim[off + p], zt, ez, a.swetclim[off + p], Qlat, Qlat_air, dq_eva, dq_rain, a.xsw);
  deep_ocean(a.phys, Ts, To, zt, mld, a.mldclim[offm + p], a.z_ocean[p], dT_ocean, dTo, a.xsw);
  float Qsens;
  {
#pragma clang fp contract(off)
    Qsens = a.phys.ct_sens * (Ta - Ts); // :295
  }
  const float capn = seaice(a.phys, Ts, zt, gl, mld, cap, a.xsw);
  const float o[15] = {albedo, sw, LWsurf, LWdown, em, Qsens, Qlat, Qlat_air, dq_eva, dq_rain, dT_ocean, dTo, capn, 0.f, 0.f};
  for (int i = 0; i < 15; ++i) a.out15[(size_t)i * a.np + p] = o[i];
}

hipError_t launch_point_physics(const PointArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(point_kernel, dim3((a.np + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ============================================================================================
// Any-grid ("multi-launch") engine pieces for grids whose member does not fit one CU's LDS
// (384x192: 295 KB per field).  One launch per circulation sub-step over (member x tracer) x
// latitude bands, one launch per model step for the point physics, one per year for the
// diagnostic global mean.  Same device functions as the fused 96x48 engine.
// ============================================================================================
hipError_t launch_substep_fused(const float* X, const float* W2, const float* u, const float* v, float* Xnew,
                                const RowTables* tabs, const int* tab_index, int nx, int ny, int n_members,
                                bool strict, hipStream_t s, bool calm_vapor) {
  // batch = n_members x {Tair, q}; weights W2[2] = {wz_air, wz_vapor}; winds shared by everyone
  return launch_sweep<kChainFused>(X, W2, u, v, Xnew, tabs, nx, ny, 2 * n_members, strict, s, 2, 1, tab_index, 2, calm_vapor ? 1 : 0);
}

template <bool STRICT, bool FLUX, bool EXP>
// X and Xout may be the SAME buffer (run_year passes the current tracer buffer as both when the sub-step count is
// even): every thread reads its own quad before it writes it and touches no other, so in-place is safe -- and the
// pointers are therefore not __restrict__.
__global__ __launch_bounds__(256) void physics_step_kernel(MemberArgs a, const float* X, float* Xout,
                                                           float* __restrict__ red) {
  const int m = blockIdx.y, np = a.np;
  const int qd = blockIdx.x * blockDim.x + threadIdx.x;
  if (qd >= np / 4) return;
  const StepClock ck = step_clock<FLUX>(a, a.it0, np);
  const Phys P = a.phys[m];
  const float co2 = FLUX ? a.co2_flux : a.co2[(size_t)m * a.co2_stride + a.co2_year0];
  float* state = a.state + (size_t)m * 5 * np;
  float* acc = a.acc + (size_t)m * 6 * np;
  float* corr = a.corr + (size_t)a.corr_index[m] * 3 * kNT * np;
  const f4 xTa = ld4(X + ((size_t)m * 2) * np + 4 * qd), xq = ld4(X + ((size_t)m * 2 + 1) * np + 4 * qd);
  f4 oTa, oq, tsm;
  physics_quad<STRICT, FLUX, EXP>(a, P, m, qd, ck, co2, state, acc, corr, xTa, xq, oTa, oq, tsm);
  st4(Xout + ((size_t)m * 2) * np + 4 * qd, oTa);
  st4(Xout + ((size_t)m * 2 + 1) * np + 4 * qd, oq);
  if (ck.ityr == kNT) st4(red + (size_t)m * np + 4 * qd, tsm);
}

hipError_t launch_physics_step(const MemberArgs& a, const float* X, float* Xout, float* red, int n_members,
                               bool strict, hipStream_t s) {
  void (*kern)(MemberArgs, const float*, float*, float*);
  if (a.xsw) {
    if (a.flux_phase) kern = strict ? physics_step_kernel<true, true, true> : physics_step_kernel<false, true, true>;
    else kern = strict ? physics_step_kernel<true, false, true> : physics_step_kernel<false, false, true>;
  } else if (a.flux_phase) kern = strict ? physics_step_kernel<true, true, false> : physics_step_kernel<false, true, false>;
  else kern = strict ? physics_step_kernel<true, false, false> : physics_step_kernel<false, false, false>;
  hipLaunchKernelGGL(kern, dim3((a.np / 4 + 255) / 256, n_members), dim3(256), 0, s, a, X, Xout, red);
  return hipGetLastError();
}

// diagnostics at the end of a model year (src/greb.f90:948-954): the reference's sum() is a
// sequential fp32 loop; one thread per member reproduces that order (STRICT)
// FAST arithmetic: one 256-thread workgroup per member, per-lane partial sums + wavefront shuffle reduction
__global__ __launch_bounds__(256) void yearly_reduce_kernel(const float* __restrict__ red, float* __restrict__ yearly,
                                                            int np, int nx, int ipx, int ipy, int yearly_years,
                                                            int year_index) {
  __shared__ float wsum[4];
  const int m = blockIdx.x;
  const float* r = red + (size_t)m * np;
  float part = 0.f;
  for (int i = threadIdx.x; i < np; i += 256) part += r[i];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float* y = yearly + ((size_t)m * yearly_years + year_index) * 2;
    y[0] = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (float)np - 273.15f; // :954
    y[1] = r[(ipy - 1) * nx + (ipx - 1)] - 273.15f;
  }
}

__global__ void yearly_kernel(const float* __restrict__ red, float* __restrict__ yearly, int np, int nx, int ipx,
                              int ipy, int yearly_years, int year_index, int n_members) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_members) return;
  {
#pragma clang fp contract(off)
    const float* r = red + (size_t)m * np;
    float sum = 0.f;
    for (int i = 0; i < np; ++i) sum += r[i];
    float* y = yearly + ((size_t)m * yearly_years + year_index) * 2;
    y[0] = sum / (float)np - 273.15f;
    y[1] = r[(ipy - 1) * nx + (ipx - 1)] - 273.15f;
  }
}

hipError_t launch_yearly(const float* red, float* yearly, int np, int nx, int ipx, int ipy, int yearly_years,
                         int year_index, int n_members, bool strict, hipStream_t s) {
  if (!strict) {
    hipLaunchKernelGGL(yearly_reduce_kernel, dim3(n_members), dim3(256), 0, s, red, yearly, np, nx, ipx, ipy, yearly_years,
                       year_index);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(yearly_kernel, dim3((n_members + 63) / 64), dim3(64), 0, s, red, yearly, np, nx, ipx, ipy,
                     yearly_years, year_index, n_members);
  return hipGetLastError();
}

__global__ void pack_tracers_kernel(const float* __restrict__ state, float* __restrict__ X, int np, int n_members) {
  // X[m][0] = Tair = state[m][1], X[m][1] = q = state[m][3]
  const size_t n = (size_t)n_members * 2 * np;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / (2 * (size_t)np), r = i % (2 * (size_t)np);
    const int tr = (int)(r / np);
    X[i] = state[m * 5 * np + (size_t)(tr ? 3 : 1) * np + r % np];
  }
}
hipError_t launch_pack_tracers(const float* state, float* X, int np, int n_members, hipStream_t s) {
  hipLaunchKernelGGL(pack_tracers_kernel, dim3(1024), dim3(256), 0, s, state, X, np, n_members);
  return hipGetLastError();
}

} // namespace greb
