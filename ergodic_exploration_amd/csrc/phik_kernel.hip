// phi_k path for gfx950: Target::fill (reference target.cpp:78-89, target.hpp:91-102) and
// Basis::spatialCoeff (basis.cpp:122-133) behind ErgodicControl::configTarget
// (ergodic_control.hpp:362-416), plus the point-list form used by Basis::trajCoeff /
// Basis::spatialCoeff as free operations.
//
// The reference materialises a K^2 x P matrix and calls cos 2 K^2 P times.  Here the basis is
// used in its separable form on the regular target grid:
//     phi_k(k1,k2) = sum_iy cos(b_k2 y_iy) [ sum_ix Phi(iy,ix) cos(a_k1 x_ix) ]
// so the grid Phi is streamed from HBM exactly once (coalesced along x, the fastest index of
// the reference's point order col = iy*nx + ix), the y table row of the current grid row is
// wave-uniform (scalar loads), and each lane keeps K running sums.  Two passes, no atomics:
// pass 1 writes K^2 partials per workgroup, pass 2 adds them in a fixed order, so results are
// deterministic run to run.
#include <cstdlib>

#include "common.hpp"

// grid rows whose loads are issued before the first is consumed (memory-level parallelism per lane)
#ifndef EEA_PHIK_ROWS_IN_FLIGHT
#define EEA_PHIK_ROWS_IN_FLIGHT 4
#endif

namespace eea
{
namespace
{
template <typename R>
__device__ __forceinline__ R exp_r(R v);
template <>
__device__ __forceinline__ double exp_r<double>(double v)
{
  return exp(v);
}
template <>
__device__ __forceinline__ float exp_r<float>(float v)
{
  return expf(v);
}
template <typename R>
__device__ __forceinline__ R cos_r(R v);
template <>
__device__ __forceinline__ double cos_r<double>(double v)
{
  return cos(v);
}
template <>
__device__ __forceinline__ float cos_r<float>(float v)
{
  return cosf(v);
}

// sum over the workgroup, result valid in every thread; s_w: kBlock / kWave reals
template <typename R>
__device__ __forceinline__ R block_sum(R v, R* s_w)
{
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  if (lane == 0) s_w[wave] = v;
  __syncthreads();
  R t = R(0);
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) t += s_w[w];
  __syncthreads();
  return t;
}

// cos((k * pi/l) * coord_i): the grouping of basis.cpp:85
template <typename R>
__global__ __launch_bounds__(kBlock) void cos_table_kernel(const R* __restrict__ coord, int n, int K,
                                                           R pi_over_l, R* __restrict__ out,
                                                           int transpose)
{
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n * K) return;
  const int k = idx / n, i = idx - k * n;
  const R v = cos_r((static_cast<R>(k) * pi_over_l) * coord[i]);
  out[transpose ? (i * K + k) : idx] = v;
}

// un-normalised sum of axis-aligned Gaussians on the grid; gauss: [n][4] = mean (Fourier
// frame) and diagonal of the inverse covariance
template <typename R>
__global__ __launch_bounds__(kBlock) void target_fill_kernel(const R* __restrict__ xs,
                                                             const R* __restrict__ ys, int nx, int ny,
                                                             const R* __restrict__ gauss, int n_gauss,
                                                             R* __restrict__ phi,
                                                             R* __restrict__ partials)
{
  __shared__ R s_w[kBlock / kWave];
  const size_t P = static_cast<size_t>(nx) * ny;
  const size_t q = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  R val = R(0);
  if (q < P) {
    const int iy = static_cast<int>(q / nx), ix = static_cast<int>(q - static_cast<size_t>(iy) * nx);
    const R x = xs[ix], y = ys[iy];
    for (int g = 0; g < n_gauss; ++g) {
      const R dx = x - gauss[4 * g + 0], dy = y - gauss[4 * g + 1];
      // dot(diff.t() * cov_inv, diff) with a diagonal cov_inv (target.hpp:101)
      val += exp_r(R(-0.5) * ((dx * gauss[4 * g + 2]) * dx + (dy * gauss[4 * g + 3]) * dy));
    }
    phi[q] = val;
  }
  const R t = block_sum(val, s_w);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

template <typename R>
__global__ __launch_bounds__(kBlock) void target_fill_points_kernel(const R* __restrict__ px,
                                                                    const R* __restrict__ py, unsigned P,
                                                                    const R* __restrict__ gauss, int n_gauss,
                                                                    R* __restrict__ phi,
                                                                    R* __restrict__ partials)
{
  __shared__ R s_w[kBlock / kWave];
  const unsigned q = blockIdx.x * kBlock + threadIdx.x;
  R val = R(0);
  if (q < P) {
    const R x = px[q], y = py[q];
    for (int g = 0; g < n_gauss; ++g) {
      const R dx = x - gauss[4 * g + 0], dy = y - gauss[4 * g + 1];
      val += exp_r(R(-0.5) * ((dx * gauss[4 * g + 2]) * dx + (dy * gauss[4 * g + 3]) * dy));
    }
    phi[q] = val;
  }
  const R t = block_sum(val, s_w);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

template <typename R>
__global__ __launch_bounds__(kBlock) void reduce_sum_kernel(const R* __restrict__ in, int n,
                                                            R* __restrict__ out)
{
  __shared__ R s_w[kBlock / kWave];
  R v = R(0);
  for (int i = threadIdx.x; i < n; i += kBlock) v += in[i];
  const R t = block_sum(v, s_w);
  if (threadIdx.x == 0) out[0] = t;
}

// phi_vals /= sum(phi_vals)  (target.cpp:87)
template <typename R>
__global__ __launch_bounds__(kBlock) void scale_by_inv_kernel(R* __restrict__ phi, size_t n,
                                                              const R* __restrict__ sum)
{
  const R s = sum[0];
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<size_t>(gridDim.x) * kBlock) {
    phi[i] = phi[i] / s;
  }
}

// pass 1: workgroup = (256 * CPT) grid columns x `rows_per_tile` grid rows.  Each lane owns CPT
// adjacent columns (CPT = 2: one 16-byte load per row for fp64) and keeps K running sums per
// column; four rows are in flight per iteration so that a wavefront has 4 KiB of loads outstanding.
// EXACT: K == KMAX is known at compile time (no per-mode bounds tests in the streaming loop).
template <typename R, int KMAX, int CPT, bool EXACT>
__global__ __launch_bounds__(kBlock) void spatial_pass1_kernel(const R* __restrict__ phi, int nx, int ny,
                                                               int K_rt, int rows_per_tile,
                                                               const R* __restrict__ cx,
                                                               const R* __restrict__ cy,
                                                               R* __restrict__ partials)
{
  constexpr int kCols = kBlock * CPT;
  constexpr int kPad = kCols + 1;  // LDS row stride of the per-column sums
  constexpr int kRowsInFlight = EEA_PHIK_ROWS_IN_FLIGHT;
  const int K = EXACT ? KMAX : K_rt;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  R* const s_S = reinterpret_cast<R*>(smem_raw);  // [K][kPad]

  const int tid = threadIdx.x;
  const int ix0 = blockIdx.x * kCols;
  const int ix = ix0 + tid * CPT;
  const int r0 = blockIdx.y * rows_per_tile;
  const int r1 = (r0 + rows_per_tile) < ny ? (r0 + rows_per_tile) : ny;
  // vector loads need the whole CPT group in range and a 16-byte aligned row pitch
  const bool vec_ok = (CPT == 1) || ((ix + CPT <= nx) && ((nx % CPT) == 0));

  R acc[CPT][KMAX];
#pragma unroll
  for (int c = 0; c < CPT; ++c)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[c][k] = R(0);

  // the y-table rows of this tile, staged once: [rows_per_tile][KY] behind the column sums (KY = K
  // rounded up to an even count so that every row starts 16-byte aligned).  In the streaming loop
  // they are read back as LDS broadcasts, which can be issued ahead of the FMAs (scalar loads
  // return out of order and force a full wait in front of every row).
  constexpr int KY = (KMAX + 1) & ~1;
  R* const s_cy = s_S + ((K * kPad + kBlock + 3) & ~3);
  for (int e = tid; e < (r1 - r0) * K; e += kBlock) {
    const int rr = e / K, kk = e - rr * K;
    s_cy[rr * KY + kk] = cy[static_cast<size_t>(r0 + rr) * K + kk];
  }
  __syncthreads();

  auto load_row = [&](int iy, R (&v)[CPT]) {
    const R* const row = phi + static_cast<size_t>(iy) * nx;
    if (vec_ok) {
      if (CPT == 2) {
        typedef R vec2 __attribute__((ext_vector_type(2)));
        const vec2 t = *reinterpret_cast<const vec2*>(row + ix);
        v[0] = t.x;
        v[CPT - 1] = t.y;
      } else {
        v[0] = (ix < nx) ? row[ix] : R(0);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CPT; ++c) v[c] = (ix + c < nx) ? row[ix + c] : R(0);
    }
  };

  // software pipeline: the loads of the next group of rows are issued before the current group is
  // consumed, so HBM latency overlaps the K FMAs per element instead of preceding them
  int iy = r0;
  R vcur[kRowsInFlight][CPT], vnxt[kRowsInFlight][CPT];
  const bool have_first = iy + kRowsInFlight <= r1;
  if (have_first) {
#pragma unroll
    for (int u = 0; u < kRowsInFlight; ++u) load_row(iy + u, vnxt[u]);
  }
  for (; iy + kRowsInFlight <= r1; iy += kRowsInFlight) {
#pragma unroll
    for (int u = 0; u < kRowsInFlight; ++u)
#pragma unroll
      for (int c = 0; c < CPT; ++c) vcur[u][c] = vnxt[u][c];
    if (iy + 2 * kRowsInFlight <= r1) {
#pragma unroll
      for (int u = 0; u < kRowsInFlight; ++u) load_row(iy + kRowsInFlight + u, vnxt[u]);
    }
#pragma unroll
    for (int u = 0; u < kRowsInFlight; ++u) {
      const R* const cyrow = s_cy + (iy + u - r0) * KY;  // same address in every lane: broadcast
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (EXACT || k < K) {
          const R w = cyrow[k];
#pragma unroll
          for (int c = 0; c < CPT; ++c) acc[c][k] += vcur[u][c] * w;
        }
      }
    }
  }
  for (; iy < r1; ++iy) {
    R v[CPT];
    load_row(iy, v);
    const R* const cyrow = s_cy + (iy - r0) * KY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (EXACT || k < K) {
        const R w = cyrow[k];
#pragma unroll
        for (int c = 0; c < CPT; ++c) acc[c][k] += v[c] * w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (EXACT || k < K) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) s_S[k * kPad + tid * CPT + c] = acc[c][k];
    }
  }
  __syncthreads();

  // epilogue: out[k2][k1] = sum_c S[k2][c] cx[k1][ix0 + c] -- a (K x kCols)(kCols x K) product, on the
  // matrix cores: A[i = k2][k = column] from the LDS column sums, B[k = column][j = k1] from the x
  // table (L2-resident), four columns per v_mfma_*_16x16x4, each wavefront a quarter of the column
  // groups, then one add per mode across the four wavefronts.  Rows / columns beyond K are clamped
  // to K - 1 (their products land in ignored accumulator entries); out-of-range grid columns carry
  // zero sums.
  using M = Mfma<R>;
  using acc_t = typename M::acc_t;
  constexpr int NT = (KMAX + 15) / 16;
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  const int li = lane & 15, lk = lane >> 4;
  acc_t macc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) macc[a][b] = acc_t{ R(0), R(0), R(0), R(0) };
  int rowA[NT], rowB[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r = 16 * t + li;
    rowA[t] = (r < K ? r : K - 1) * kPad;                             // k2 row of the column sums
    rowB[t] = r < K ? r : K - 1;                                      // k1 row of the x table
  }
  for (int cg = wave; cg < kCols / 4; cg += kBlock / kWave) {
    const int c = 4 * cg + lk;                   // this lane's column within the tile
    int gx = ix0 + c;
    gx = gx < nx ? gx : nx - 1;                  // clamped: the matching column sum is zero
    R av[NT], bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      av[t] = s_S[rowA[t] + c];
      bv[t] = cx[static_cast<size_t>(rowB[t]) * nx + gx];
    }
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) macc[a][b] = M::run(av[a], bv[b], macc[a][b]);
  }
  __syncthreads();  // every wavefront is done reading the column sums
  R* const s_red = s_S;  // [4 wavefronts][K^2], reuses the column-sum region (K^2 * 4 <= K * kPad)
  const int K2 = K * K;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k2 = 16 * a + M::row(lane, r), k1 = 16 * b + li;
        if (k2 < K && k1 < K) s_red[wave * K2 + k2 * K + k1] = macc[a][b][r];  // col = k2*K + k1
      }
  __syncthreads();
  R* const out = partials + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * K2;
  for (int m = tid; m < K2; m += kBlock) {
    R t = R(0);
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += s_red[w * K2 + m];
    out[m] = t;
  }
}

// pass 2: add the partials of all workgroups in a fixed order
template <typename R>
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const R* __restrict__ partials,
                                                              int n_parts, int K2, R scale,
                                                              R* __restrict__ out)
{
  for (int m = blockIdx.x * kBlock + threadIdx.x; m < K2; m += gridDim.x * kBlock) {
    R s = R(0);
    for (int b = 0; b < n_parts; ++b) s += partials[static_cast<size_t>(b) * K2 + m];
    out[m] = scale * s;
  }
}

// columns per lane: two (16-byte fp64 loads) while the K x 513 LDS tile stays small
inline int spatial_cpt(int K)
{
  static const int forced = [] {
    const char* v = std::getenv("EEA_PHIK_CPT");
    return v ? std::atoi(v) : 0;
  }();
  if (forced == 2 && K == 10) return 2;  // the two-column variant is built for K = 10 only (A/B knob)
  // measured (profiles/r01_phik_rows_sweep.txt): one column per lane halves the LDS per workgroup
  // (7 instead of 3 workgroups per CU) and wins over the wider 16-byte loads of two columns
  return 1;
}

// rows per workgroup: tall tiles amortise the K^2 epilogue (>= 64 rows) while leaving a few
// workgroups per CU on large grids
inline int spatial_rows_per_tile(int nx, int ny, int K)
{
  const int cols = kBlock * spatial_cpt(K);
  const int col_tiles = (nx + cols - 1) / cols;
  // tuning knob for tools/phik_prof.sh: EEA_PHIK_ROWS fixes the rows per tile
  static const int forced = [] {
    const char* v = std::getenv("EEA_PHIK_ROWS");
    return v ? std::atoi(v) : 0;
  }();
  if (forced > 0) return forced < ny ? forced : ny;
  // measured on MI355X (profiles/r01_phik_rows_sweep.txt): 128-row tiles for K <= 20 and 64-row
  // tiles for larger K are best on 8192^2 grids (~8 / ~16 workgroups per CU in total)
  int row_tiles = (K > 20 ? 4096 : 2048) / col_tiles;
  if (row_tiles < 1) row_tiles = 1;
  int rpt = (ny + row_tiles - 1) / row_tiles;
  if (rpt < 32) rpt = 32;
  if (rpt > ny) rpt = ny;
  return rpt;
}

constexpr int kPointChunk = 128;        // points staged in LDS at a time
constexpr int kPointsPerBlock = 2048;   // points one workgroup reduces

// weighted sum of the (separable) basis over an arbitrary point list
template <typename R>
__global__ __launch_bounds__(kBlock) void point_coeff_kernel(const R* __restrict__ xs,
                                                             const R* __restrict__ ys,
                                                             const R* __restrict__ ws, unsigned P,
                                                             int K, R pi_lx, R pi_ly,
                                                             R* __restrict__ partials)
{
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  R* const s_tabx = reinterpret_cast<R*>(smem_raw);  // [chunk][K], weight folded in
  R* const s_taby = s_tabx + kPointChunk * K;
  const int tid = threadIdx.x;
  const int K2 = K * K;
  const int ntx = (K + 1) / 2;
  const int ntiles = ntx * ntx;
  const int G = kBlock / ntiles;
  const int tile = tid % ntiles, grp = tid / ntiles;
  const bool active = grp < G;
  const int i2 = 2 * (tile % ntx), j2 = 2 * (tile / ntx);
  const bool i_pair = (i2 + 1) < K, j_pair = (j2 + 1) < K;
  R a00 = R(0), a10 = R(0), a01 = R(0), a11 = R(0);

  const unsigned p_begin = blockIdx.x * kPointsPerBlock;
  const unsigned p_end = (p_begin + kPointsPerBlock) < P ? (p_begin + kPointsPerBlock) : P;
  for (unsigned c0 = p_begin; c0 < p_end; c0 += kPointChunk) {
    const int npts = static_cast<int>((p_end - c0) < kPointChunk ? (p_end - c0) : kPointChunk);
    if (tid < npts) {
      const unsigned q = c0 + tid;
      const R w = (ws != nullptr) ? ws[q] : R(1);
      // direct evaluation per mode, as basis.cpp:85 does (this op is not on the control path)
      for (int k = 0; k < K; ++k) {
        s_tabx[tid * K + k] = w * cos_r((static_cast<R>(k) * pi_lx) * xs[q]);
        s_taby[tid * K + k] = cos_r((static_cast<R>(k) * pi_ly) * ys[q]);
      }
    }
    __syncthreads();
    if (active) {
      for (int pl = grp; pl < npts; pl += G) {
        const R* const tx = s_tabx + pl * K;
        const R* const ty = s_taby + pl * K;
        const R x0 = tx[i2], x1 = i_pair ? tx[i2 + 1] : R(0);
        const R y0 = ty[j2], y1 = j_pair ? ty[j2 + 1] : R(0);
        a00 += x0 * y0;
        a10 += x1 * y0;
        a01 += x0 * y1;
        a11 += x1 * y1;
      }
    }
    __syncthreads();
  }
  R* const s_red = s_tabx;  // [G][K2] <= 4 * kBlock reals
  if (active) {
    R* const r = s_red + grp * K2;
    r[j2 * K + i2] = a00;
    if (i_pair) r[j2 * K + i2 + 1] = a10;
    if (j_pair) {
      r[(j2 + 1) * K + i2] = a01;
      if (i_pair) r[(j2 + 1) * K + i2 + 1] = a11;
    }
  }
  __syncthreads();
  for (int m = tid; m < K2; m += kBlock) {
    R s = R(0);
    for (int g = 0; g < G; ++g) s += s_red[g * K2 + m];
    partials[static_cast<size_t>(blockIdx.x) * K2 + m] = s;
  }
}

inline size_t point_lds_elems(int K)
{
  const size_t tab = 2 * static_cast<size_t>(kPointChunk) * K;
  return tab > 4 * kBlock ? tab : 4 * kBlock;
}
}  // namespace

template <typename R>
hipError_t launch_cos_tables(const R* d_coord, int n, int K, R pi_over_l, R* d_out, hipStream_t s)
{
  // not transposed; see launch_cos_tables_t for the [n][K] layout
  const int total = n * K;
  hipLaunchKernelGGL(cos_table_kernel<R>, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                     d_coord, n, K, pi_over_l, d_out, 0);
  return hipGetLastError();
}

template <typename R>
hipError_t launch_cos_tables_t(const R* d_coord, int n, int K, R pi_over_l, R* d_out, hipStream_t s)
{
  const int total = n * K;
  hipLaunchKernelGGL(cos_table_kernel<R>, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                     d_coord, n, K, pi_over_l, d_out, 1);
  return hipGetLastError();
}

template <typename R>
hipError_t launch_target_fill(const R* d_xs, const R* d_ys, int nx, int ny, const R* d_gauss,
                              int n_gauss, R* d_phi, R* d_partials, int* n_partials, hipStream_t s)
{
  const size_t P = static_cast<size_t>(nx) * ny;
  const int blocks = static_cast<int>((P + kBlock - 1) / kBlock);
  *n_partials = blocks;
  hipLaunchKernelGGL(target_fill_kernel<R>, dim3(blocks), dim3(kBlock), 0, s, d_xs, d_ys, nx, ny,
                     d_gauss, n_gauss, d_phi, d_partials);
  return hipGetLastError();
}

template <typename R>
hipError_t launch_target_fill_points(const R* d_px, const R* d_py, unsigned P, const R* d_gauss,
                                     int n_gauss, R* d_phi, R* d_partials, int* n_partials,
                                     hipStream_t s)
{
  const int blocks = static_cast<int>((P + kBlock - 1) / kBlock);
  *n_partials = blocks;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(target_fill_points_kernel<R>, dim3(blocks), dim3(kBlock), 0, s, d_px, d_py, P,
                     d_gauss, n_gauss, d_phi, d_partials);
  return hipGetLastError();
}

template <typename R>
hipError_t launch_reduce_sum(const R* d_in, int n, R* d_out, hipStream_t s)
{
  hipLaunchKernelGGL(reduce_sum_kernel<R>, dim3(1), dim3(kBlock), 0, s, d_in, n, d_out);
  return hipGetLastError();
}

template <typename R>
hipError_t launch_scale_by_inv(R* d_phi, size_t n, const R* d_sum, hipStream_t s)
{
  size_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(scale_by_inv_kernel<R>, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s,
                     d_phi, n, d_sum);
  return hipGetLastError();
}

size_t spatial_work_elems(int nx, int ny, int K)
{
  const int cols = kBlock * spatial_cpt(K);
  const int col_tiles = (nx + cols - 1) / cols;
  const int rpt = spatial_rows_per_tile(nx, ny, K);
  const int row_tiles = (ny + rpt - 1) / rpt;
  return static_cast<size_t>(col_tiles) * row_tiles * K * K;
}

template <typename R>
hipError_t launch_spatial_coeff(const R* d_phi, int nx, int ny, int K, const R* d_cx, const R* d_cy,
                                R* d_work, R* d_phik, hipStream_t s)
{
  const int cpt = spatial_cpt(K);
  const int cols = kBlock * cpt;
  const int col_tiles = (nx + cols - 1) / cols;
  const int rpt = spatial_rows_per_tile(nx, ny, K);
  const int row_tiles = (ny + rpt - 1) / rpt;
  // column sums [K][cols + 1] plus the epilogue's per-group partials (<= 256 reals)
  const int KYmax = ((K <= 16 ? (K == 5 ? 5 : (K == 10 ? 10 : 16)) : (K == 20 ? 20 : (K == 30 ? 30 : 32))) + 1) & ~1;
  const size_t lds = (static_cast<size_t>(K) * (cols + 1) + kBlock + 4 + static_cast<size_t>(rpt) * KYmax) * sizeof(R);
  const dim3 grid(col_tiles, row_tiles);
  auto launch = [&](auto kern) -> hipError_t {
    if (lds > 64 * 1024) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(lds));
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, dim3(kBlock), lds, s, d_phi, nx, ny, K, rpt, d_cx, d_cy, d_work);
    return hipGetLastError();
  };
  hipError_t e;
  if (K == 5) e = launch(spatial_pass1_kernel<R, 5, 1, true>);
  else if (K == 10 && cpt == 2) e = launch(spatial_pass1_kernel<R, 10, 2, true>);
  else if (K == 10) e = launch(spatial_pass1_kernel<R, 10, 1, true>);
  else if (K == 20) e = launch(spatial_pass1_kernel<R, 20, 1, true>);
  else if (K == 30) e = launch(spatial_pass1_kernel<R, 30, 1, true>);
  else if (K <= 16) e = launch(spatial_pass1_kernel<R, 16, 1, false>);
  else e = launch(spatial_pass1_kernel<R, 32, 1, false>);
  if (e != hipSuccess) return e;
  const int K2 = K * K;
  hipLaunchKernelGGL(sum_partials_kernel<R>, dim3((K2 + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                     d_work, col_tiles * row_tiles, K2, R(1), d_phik);
  return hipGetLastError();
}

size_t point_work_elems(unsigned P, int K)
{
  const size_t blocks = (static_cast<size_t>(P) + kPointsPerBlock - 1) / kPointsPerBlock;
  return (blocks ? blocks : 1) * K * K;
}

template <typename R>
hipError_t launch_point_coeff(const R* d_x, const R* d_y, const R* d_w, unsigned P, int K, R pi_lx,
                              R pi_ly, R scale, R* d_work, R* d_out, hipStream_t s)
{
  const unsigned blocks = (P + kPointsPerBlock - 1) / kPointsPerBlock;
  const int K2 = K * K;
  if (blocks > 0) {
    hipLaunchKernelGGL(point_coeff_kernel<R>, dim3(blocks), dim3(kBlock), point_lds_elems(K) * sizeof(R),
                       s, d_x, d_y, d_w, P, K, pi_lx, pi_ly, d_work);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(sum_partials_kernel<R>, dim3((K2 + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                     d_work, static_cast<int>(blocks), K2, scale, d_out);
  return hipGetLastError();
}

#define EEA_INSTANTIATE(R)                                                                          \
  template hipError_t launch_cos_tables<R>(const R*, int, int, R, R*, hipStream_t);                 \
  template hipError_t launch_cos_tables_t<R>(const R*, int, int, R, R*, hipStream_t);               \
  template hipError_t launch_target_fill<R>(const R*, const R*, int, int, const R*, int, R*, R*,    \
                                            int*, hipStream_t);                                     \
  template hipError_t launch_target_fill_points<R>(const R*, const R*, unsigned, const R*, int, R*,  \
                                                   R*, int*, hipStream_t);                           \
  template hipError_t launch_reduce_sum<R>(const R*, int, R*, hipStream_t);                         \
  template hipError_t launch_scale_by_inv<R>(R*, size_t, const R*, hipStream_t);                    \
  template hipError_t launch_spatial_coeff<R>(const R*, int, int, int, const R*, const R*, R*, R*,  \
                                              hipStream_t);                                         \
  template hipError_t launch_point_coeff<R>(const R*, const R*, const R*, unsigned, int, R, R, R,   \
                                            R*, R*, hipStream_t);
EEA_INSTANTIATE(double)
EEA_INSTANTIATE(float)
#undef EEA_INSTANTIATE
}  // namespace eea
