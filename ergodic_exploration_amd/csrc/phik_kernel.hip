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
#include <cstdint>
#include <cstdlib>
#include <string>
#include <type_traits>

#include <hip/hip_ext.h>

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

// both tables of a rebuild in one launch: cx[k * nx + i] and the transposed y table cy[i * K + k]; the two axes
// share the coordinate sequence (both start at 0 and add the resolution, ergodic_control.hpp:387-408)
template <typename R>
__global__ __launch_bounds__(kBlock) void axis_tables_kernel(const R* __restrict__ coord, int nx, int ny, int K,
                                                             R pi_lx, R pi_ly, R* __restrict__ cx,
                                                             R* __restrict__ cy)
{
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  const int nxk = nx * K;
  if (idx < nxk) {
    const int k = idx / nx, i = idx - k * nx;
    cx[idx] = cos_r((static_cast<R>(k) * pi_lx) * coord[i]);
  } else if (idx < nxk + ny * K) {
    const int t = idx - nxk;
    const int k = t / ny, i = t - k * ny;
    cy[i * K + k] = cos_r((static_cast<R>(k) * pi_ly) * coord[i]);
  }
}

// Target::fill without the normalisation, Gaussians passed by value (no upload, no synchronisation):
// per_thread grid points per thread (target_fill_per_thread), one partial sum of phi per workgroup (fixed order).  The workgroups
// beyond fill_blocks compute the two axis tables of the rebuild (axis_tables_kernel's job) in the same launch.
template <typename R>
__global__ __launch_bounds__(kBlock) void target_fill_args_kernel(const R* __restrict__ coord, int nx, int ny,
                                                                  const GaussArgs<R> ga, R* __restrict__ phi,
                                                                  R* __restrict__ partials, int fill_blocks, int per_thread,
                                                                  int K,
                                                                  R pi_lx, R pi_ly, R* __restrict__ cx,
                                                                  R* __restrict__ cy)
{
  __shared__ R s_w[kBlock / kWave];
  if (static_cast<int>(blockIdx.x) >= fill_blocks) {  // whole workgroup
    const int idx = (blockIdx.x - fill_blocks) * kBlock + threadIdx.x;
    const int nxk = nx * K;
    if (idx < nxk) {
      const int k = idx / nx, i = idx - k * nx;
      cx[idx] = cos_r((static_cast<R>(k) * pi_lx) * coord[i]);
    } else if (idx < nxk + ny * K) {
      const int t = idx - nxk;
      const int k = t / ny, i = t - k * ny;
      cy[i * K + k] = cos_r((static_cast<R>(k) * pi_ly) * coord[i]);
    }
    return;
  }
  const size_t P = static_cast<size_t>(nx) * ny;
  const size_t base = static_cast<size_t>(blockIdx.x) * (static_cast<size_t>(kBlock) * per_thread);
  R acc = R(0);
#pragma unroll 4
  for (int r = 0; r < per_thread; ++r) {
    const size_t q = base + static_cast<size_t>(r) * kBlock + threadIdx.x;
    if (q < P) {
      const int iy = static_cast<int>(q / nx), ix = static_cast<int>(q - static_cast<size_t>(iy) * nx);
      const R x = coord[ix], y = coord[iy];
      R val = R(0);
      for (int g = 0; g < ga.n; ++g) {
        const R dx = x - ga.g[g][0], dy = y - ga.g[g][1];
        // dot(diff.t() * cov_inv, diff) with a diagonal cov_inv (target.hpp:101)
        val += exp_r(R(-0.5) * ((dx * ga.g[g][2]) * dx + (dy * ga.g[g][3]) * dy));
      }
      phi[q] = val;
      acc += val;
    }
  }
  const R t = block_sum(acc, s_w);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// ---- configTarget for Gaussian targets, in ONE launch of ONE workgroup (round 4) ---------------------------------------
// The reference's target is a sum of AXIS-ALIGNED Gaussians (target.hpp:68-70: Sigma = diagmat(sigma^2)), evaluated on a
// rectangular grid whose coordinates are products of two 1-D sequences (ergodic_control.hpp:387-408).  Both the target
// and the basis therefore factor per axis and so does every sum over the grid:
//   sum_{ix,iy} exp(-1/2 cxx dx^2 - 1/2 cyy dy^2) cos(a_k1 x) cos(b_k2 y)
//     = [sum_ix exp(-1/2 cxx dx^2) cos(a_k1 x_ix)] [sum_iy exp(-1/2 cyy dy^2) cos(b_k2 y_iy)]  =: Ax_g(k1) Ay_g(k2)
// phi_k = sum_g Ax_g(k1) Ay_g(k2) / sum_g Ax_g(0) Ay_g(0)   (Target::fill's normalisation, target.cpp:87, is mode (0,0))
// -- (nx + ny)(G + 1) transcendental evaluations and (nx + ny) G K multiply-adds instead of nx ny (G + K^2), the
// reference's value up to rounding (its one exp of the summed exponent against a product of two: ~2 ulp per point; the
// cosines of a point by the Chebyshev recurrence from cos(pi x / l), error <~ k^2 2^-54 as in the control kernels).  The
// grid itself (eea_get_target_grid) is filled only when somebody asks for it.
//   1. wavefronts 0..3 take the x axis, 4..7 the y axis; a lane owns the points lane + 64 (w + 4 p) of its axis.  Per
//      Gaussian g: per point one exponential and T_k(cos(pi x / l)) by recurrence, accumulated over the lane's points in
//      K registers; then ONE 4-step row sum per (g, k) -- the four rows of the wavefront keep separate partials;
//   2. A[axis][g][k] = the (wavefront, row) partials in fixed order;   3. the K^2 modes.
// One workgroup on one CU: what counts is the number of wavefront instructions (4 cycles each on one of 4 SIMDs), so
// the cross-lane sums are taken once per (g, k) and wavefront, not once per point round (first form: 30 us at 1024^2).
constexpr int kGaussBlock = 512;
constexpr int kGaussAxisWaves = 4;   // wavefronts per axis
constexpr int kGaussKMax = kMaxBasis;
template <typename R>
__global__ __launch_bounds__(kGaussBlock) void gaussian_phik_kernel(const R* __restrict__ coord, int nx, int ny,
                                                                    const GaussArgs<R> ga, int K, R inv_lx, R inv_ly,
                                                                    R* __restrict__ phik, R* __restrict__ mass_out)
{
  extern __shared__ __attribute__((aligned(16))) char gsm_raw[];
  constexpr int kParts = 2 * kGaussAxisWaves * 4;  // (axis, wavefront of the axis, row of 16 lanes)
  const int G = ga.n, GK = G * K;
  R* const sW = reinterpret_cast<R*>(gsm_raw);             // [axis][wave of the axis][row][g][K] partial sums
  R* const sA = sW + static_cast<size_t>(kParts) * GK;      // [axis][g][K]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int axis = wave / kGaussAxisWaves, aw = wave - axis * kGaussAxisWaves;
  const int cnt = axis ? ny : nx;
  const R inv_l = axis ? inv_ly : inv_lx;
  R* const dst = sW + (static_cast<size_t>(wave) * 4 + (lane >> 4)) * GK;
  const int first = aw * kWave + lane, stride = kGaussAxisWaves * kWave;
  for (int g = 0; g < G; ++g) {
    R acc[kGaussKMax];
#pragma unroll
    for (int k = 0; k < kGaussKMax; ++k) acc[k] = R(0);
    const R mean = ga.g[g][axis], cinv = ga.g[g][2 + axis];
    for (int i = first; i < cnt; i += stride) {
      const R x = coord[i];
      const R d = x - mean;
      const R e = exp_r(R(-0.5) * ((d * cinv) * d));  // (dx cinv_xx) dx as target.hpp:101 groups it
      R sn, c1;
      sincospi_r(x * inv_l, &sn, &c1);                // cos(pi x / l): basis.cpp:85 at k = 1
      const R two = c1 + c1;
      R ta = e, tb = e * c1;                          // e T_k, e T_{k+1}: the recurrence is linear
#pragma unroll
      for (int k = 0; k < kGaussKMax; ++k) {
        if (k < K) {  // wavefront-uniform
          acc[k] += ta;
          const R tn = two * tb - ta;
          ta = tb;
          tb = tn;
        }
      }
    }
    // row sums (lanes 15, 31, 47, 63 of the wavefront): four independent chains per step
#pragma unroll
    for (int k = 0; k < kGaussKMax; ++k) {
      if (k < K) {
        R v = acc[k];
        v += dpp_or_zero<0x111, 0xf>(v);
        v += dpp_or_zero<0x112, 0xf>(v);
        v += dpp_or_zero<0x114, 0xf>(v);
        v += dpp_or_zero<0x118, 0xf>(v);
        if ((lane & 15) == 15) dst[g * K + k] = v;
      }
    }
  }
  __syncthreads();
  for (int q = tid; q < 2 * GK; q += kGaussBlock) {  // q = axis * GK + (g * K + k)
    const int ax = q / GK, r = q - ax * GK;
    R a = R(0);
    for (int part = 0; part < kGaussAxisWaves * 4; ++part) a += sW[(static_cast<size_t>(ax) * kGaussAxisWaves * 4 + part) * GK + r];
    sA[q] = a;
  }
  __syncthreads();
  R mass = R(0);
  for (int g = 0; g < G; ++g) mass += sA[g * K] * sA[(G + g) * K];
  for (int m = tid; m < K * K; m += kGaussBlock) {
    const int k2 = m / K, k1 = m - k2 * K;  // mode order of basis.cpp:58-66
    R v = R(0);
    for (int g = 0; g < G; ++g) v += sA[g * K + k1] * sA[(G + g) * K + k2];
    phik[m] = v / mass;
  }
  if (tid == 0) mass_out[0] = mass;
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

// input kinds of the phi_k pass: fp64 values, fp32 values, occupancy cells (bytes)
constexpr int kKindF64 = 0, kKindF32 = 1, kKindCells = 2;

// pass 1 on the matrix cores: S[k2][col] = sum_rows cy[row][k2] * phi[row][col] is a GEMM whose
// B operand (4 rows x 16 columns per v_mfma_*_16x16x4) is exactly what a wavefront loads from the
// row-major grid: lane (k, j) = 16 k + j holds phi[row + k][col + j].  The grid therefore goes from
// HBM into the matrix instruction without a transpose or an LDS stage; the A operand (the y table,
// 4 rows x 16 modes) comes from L2.  Per lane and 4-row step: LPS vector loads of SPL columns
// (16 bytes of values, or one dword of occupancy cells) feeding U = LPS * SPL accumulator sets
// (4 NT registers each) instead of K scalar accumulators per column and K LDS broadcasts per row
// -- the kernel stays on the HBM roofline for every K <= 32.  Rows beyond the tile and columns
// beyond the grid enter as zeros; table rows / modes beyond K are clamped (their products land in
// output rows that are never read).
// Epilogue per wavefront: each accumulator set (16 modes-tile x 16 columns) goes through a small
// LDS buffer into A-operand layout and is multiplied with the x table (second GEMM,
// out[k2][k1] = sum_col S[k2][col] cx[k1][col]); then one add per mode across the wavefronts.
template <typename R, int NT, typename IN>
__global__ __launch_bounds__(kBlock) void spatial_stream_kernel(const IN* __restrict__ phi, int nx, int ny,
                                                                int K, int rows_per_tile,
                                                                const R* __restrict__ cx,
                                                                const R* __restrict__ cy,
                                                                const R* __restrict__ lut,
                                                                R* __restrict__ partials,
                                                                R* __restrict__ phik_direct,
                                                                R* __restrict__ mass_out)
{
  constexpr bool kCells = !std::is_same<IN, R>::value;
  // columns per lane and load: 16 bytes of values; 8 (one 8-byte load) or 4 occupancy cells
  constexpr int SPL = kCells ? (NT == 1 ? 8 : 4) : static_cast<int>(16 / sizeof(R));
  // loads per lane and row step; two mode tiles (K > 16) double the accumulators per column, so one
  // load keeps the kernel at 3-4 wavefronts per SIMD (measured: profiles/r01_phik_rows_sweep.txt)
  constexpr int LPS = (kCells || NT == 2) ? 1 : 2;
  constexpr int U = SPL * LPS;               // accumulator sets per lane
  constexpr int kWaveCols = 16 * U;
  constexpr int kCols = (kBlock / kWave) * kWaveCols;
  // row steps per pipeline stage: enough bytes in flight per CU for the HBM latency at the
  // occupancy the accumulators allow (cells are 8x smaller than fp64 values)
  constexpr int G = kCells ? 8 : (NT == 2 && sizeof(R) == 8 ? 4 : 2);
  constexpr int kSetPad = 17;                // LDS row stride of one accumulator set
  using M = Mfma<R>;
  using acc_t = typename M::acc_t;
  using raw_t = typename std::conditional<kCells, int, R>::type;
  constexpr int RAWN = kCells ? SPL / 4 : SPL;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  R* const sm = reinterpret_cast<R*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  const int lj = lane & 15, lk = lane >> 4;
  R* const s_set = sm + wave * (16 * NT * kSetPad);             // [16 NT modes][17]
  R* const s_lut = sm + (kBlock / kWave) * (16 * NT * kSetPad);  // [256], occupancy input only
  if (kCells) s_lut[tid] = lut[tid];

  const int ix0 = blockIdx.x * kCols + wave * kWaveCols;  // first column of this wavefront
  const int r0 = blockIdx.y * rows_per_tile;
  const int r1 = (r0 + rows_per_tile) < ny ? (r0 + rows_per_tile) : ny;
  const bool pitch_ok = (nx % SPL) == 0;  // vector loads need an aligned row pitch

  acc_t acc[U][NT];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[u][t] = acc_t{ R(0), R(0), R(0), R(0) };

  // first column of this lane's load l; accumulator set u = l * SPL + e <-> column col_of(l) + e
  auto col_of = [&](int l) { return ix0 + l * (16 * SPL) + SPL * lj; };

  struct Stage
  {
    raw_t b[LPS][RAWN];
    R a[NT];
  };
  // interior wavefront: every vector load of every full row step is inside the grid (wave-uniform)
  const bool interior = pitch_ok && (ix0 + kWaveCols <= nx);
  // fast form: no bounds tests, no divergence -- the loads of a whole pipeline stage issue back to back
  auto load_step_fast = [&](int row_base, Stage& st) {
    const int row = row_base + lk;
    const IN* const prow = phi + static_cast<size_t>(row) * nx;
#pragma unroll
    for (int l = 0; l < LPS; ++l) {
      const int c = col_of(l);
      if constexpr (kCells) {
        if constexpr (RAWN == 2) {
          typedef int int2v __attribute__((ext_vector_type(2)));
          const int2v v = *reinterpret_cast<const int2v*>(prow + c);
          st.b[l][0] = v.x;
          st.b[l][RAWN - 1] = v.y;
        } else {
          st.b[l][0] = *reinterpret_cast<const int*>(prow + c);
        }
      } else {
        typedef R vecn __attribute__((ext_vector_type(SPL)));
        const vecn v = *reinterpret_cast<const vecn*>(prow + c);
#pragma unroll
        for (int e = 0; e < SPL; ++e) st.b[l][e] = v[e];
      }
    }
    const R* const crow = cy + static_cast<size_t>(row) * K;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int m = 16 * t + lj;
      st.a[t] = crow[m < K ? m : K - 1];
    }
  };
  // masked form (tile edges): rows beyond the tile and columns beyond the grid enter as zeros
  auto load_step_edge = [&](int row_base, Stage& st) {
    const int row = row_base + lk;
    const bool row_ok = row < r1;
    const int rowc = row_ok ? row : r1 - 1;
    const IN* const prow = phi + static_cast<size_t>(rowc) * nx;
#pragma unroll
    for (int l = 0; l < LPS; ++l) {
      const int c = col_of(l);
      if constexpr (kCells) {
#pragma unroll
        for (int w = 0; w < RAWN; ++w) st.b[l][w] = 0;
        if (row_ok) {
#pragma unroll
          for (int e = 0; e < SPL; ++e) {
            if (c + e < nx) {
              st.b[l][e / 4] |= static_cast<int>(static_cast<unsigned char>(prow[c + e])) << (8 * (e % 4));
            }
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < SPL; ++e) st.b[l][e] = (row_ok && c + e < nx) ? prow[c + e] : R(0);
      }
    }
    // A operand: lane (k, i) = cy[row + k][16 t + i]
    const R* const crow = cy + static_cast<size_t>(rowc) * K;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int m = 16 * t + lj;
      st.a[t] = crow[m < K ? m : K - 1];
    }
  };
  auto load_group = [&](int row, Stage (&st)[G]) {
    if (interior && row + 4 * G <= r1) {  // wave-uniform
#pragma unroll
      for (int g = 0; g < G; ++g) load_step_fast(row + 4 * g, st[g]);
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) load_step_edge(row + 4 * g, st[g]);
    }
  };
  if (kCells) __syncthreads();  // decode table staged
  auto mma_step = [&](int row_base, const Stage& st) {
    const bool row_ok = row_base + lk < r1;
#pragma unroll
    for (int l = 0; l < LPS; ++l) {
#pragma unroll
      for (int e = 0; e < SPL; ++e) {
        R bv;
        if constexpr (kCells) {
          // columns beyond the grid / rows beyond the tile decode to 0
          const R d = s_lut[(st.b[l][e / 4] >> (8 * (e % 4))) & 0xff];
          bv = (row_ok && col_of(l) + e < nx) ? d : R(0);
        } else {
          bv = st.b[l][e];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[l * SPL + e][t] = M::run(st.a[t], bv, acc[l * SPL + e][t]);
      }
    }
  };

  // software pipeline over groups of G row steps: the next group's loads are in flight while the
  // current group feeds the matrix cores
  Stage cur[G], nxt[G];
  int row = r0;
  load_group(row, nxt);
  for (; row < r1; row += 4 * G) {
#pragma unroll
    for (int g = 0; g < G; ++g) cur[g] = nxt[g];
    if (row + 4 * G < r1) load_group(row + 4 * G, nxt);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (row + 4 * g < r1) mma_step(row + 4 * g, cur[g]);  // wave-uniform
    }
  }

  // ---- epilogue: out[k2][k1] += sum over this wavefront's columns of S[k2][col] cx[k1][col]
  acc_t macc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) macc[a][b] = acc_t{ R(0), R(0), R(0), R(0) };
#pragma unroll
  for (int u = 0; u < U; ++u) {
    // accumulator set -> LDS, [mode][column]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_set[(16 * t + M::row(lane, r)) * kSetPad + lj] = acc[u][t][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int l = u / SPL, e = u % SPL;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      // A[i = k2][k = column 4 g4 + k], B[k = column][j = k1]
      const int cset = 4 * g4 + lk;                               // column index inside the set
      int gx = ix0 + l * (16 * SPL) + SPL * cset + e;             // its grid column
      gx = gx < nx ? gx : nx - 1;                                 // clamped: that column's sums are zero
      R av[NT], bv[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        av[t] = s_set[(16 * t + lj) * kSetPad + cset];
        const int k1 = 16 * t + lj;
        bv[t] = cx[static_cast<size_t>(k1 < K ? k1 : K - 1) * nx + gx];
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) macc[a][b] = M::run(av[a], bv[b], macc[a][b]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();  // all wavefronts are done with their set buffers
  R* const s_red = sm;  // [4 wavefronts][K^2] <= 4 * (16 NT)^2 reals, fits the set buffers + table
  const int K2 = K * K;
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k2 = 16 * a + M::row(lane, r), k1 = 16 * b + lj;
        if (k2 < K && k1 < K) s_red[wave * K2 + k2 * K + k1] = macc[a][b][r];  // col = k2*K + k1
      }
  __syncthreads();
  if (phik_direct != nullptr) {
    // the only workgroup of a small grid: its sums are final.  Normalisation of target.cpp:87 folded in: mode (0,0)
    // of the un-normalised sums is the mass of the grid (cos 0 = 1); no second launch
    R mass = R(0);
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) mass += s_red[w * K2];
    for (int m = tid; m < K2; m += kBlock) {
      R t = R(0);
#pragma unroll
      for (int w = 0; w < kBlock / kWave; ++w) t += s_red[w * K2 + m];
      phik_direct[m] = t / mass;
    }
    if (tid == 0) mass_out[0] = mass;
    return;
  }
  R* const out = partials + (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * K2;
  for (int m = tid; m < K2; m += kBlock) {
    R t = R(0);
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += s_red[w * K2 + m];
    out[m] = t;
  }
}

// pass 2: add the partials of all workgroups in a fixed order (run-to-run deterministic): one
// wavefront per mode, lanes stride over the partials, then a fixed shuffle tree.  (One thread per
// mode walking all partials took 3x as long as the streaming pass on a grid with 1400 tiles.)
template <typename R>
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const R* __restrict__ partials,
                                                              int n_parts, int K2, R scale,
                                                              R* __restrict__ out)
{
  const int lane = threadIdx.x & (kWave - 1);
  const int m = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (m >= K2) return;  // whole wavefront
  R s = R(0);
  for (int b = lane; b < n_parts; b += kWave) s += partials[static_cast<size_t>(b) * K2 + m];
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) s += __shfl_down(s, o, kWave);
  if (lane == 0) out[m] = scale * s;
}
constexpr int kModesPerSumBlock = kBlock / kWave;

// the same with the normalisation of Target::fill (target.cpp:87) folded in: the coefficient sums of the
// UN-normalised grid divided by its mass (spatialCoeff is linear in phi_vals); every wavefront adds the mass
// partials itself, in a fixed order; mass_out[0] receives the mass (for eea_get_target_grid)
template <typename R>
__global__ __launch_bounds__(kBlock) void sum_partials_norm_kernel(const R* __restrict__ partials, int n_parts,
                                                                   int K2, const R* __restrict__ mass_partials,
                                                                   int n_mass, R* __restrict__ out,
                                                                   R* __restrict__ mass_out, int mass_stride)
{
  const int lane = threadIdx.x & (kWave - 1);
  const int m = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
  if (m >= K2) return;  // whole wavefront
  R s = R(0), w = R(0);
  for (int b = lane; b < n_parts; b += kWave) s += partials[static_cast<size_t>(b) * K2 + m];
  // (mass_stride = K^2, mass_partials = partials: the mass is mode (0, 0) of the same sums -- cos 0 = 1 --, added in the same order as
  // that mode's own sum: what normalise_by_first divides by, without its launch)
  for (int b = lane; b < n_mass; b += kWave) w += mass_partials[static_cast<size_t>(b) * mass_stride];
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    s += __shfl_down(s, o, kWave);
    w += __shfl_down(w, o, kWave);
  }
  if (lane == 0) {
    out[m] = s / w;
    if (m == 0) mass_out[0] = w;
  }
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
hipError_t launch_axis_tables(const R* d_coord, int nx, int ny, int K, R pi_lx, R pi_ly, R* d_cx, R* d_cy,
                              hipStream_t s)
{
  const int total = (nx + ny) * K;
  hipLaunchKernelGGL(axis_tables_kernel<R>, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d_coord, nx,
                     ny, K, pi_lx, pi_ly, d_cx, d_cy);
  return hipGetLastError();
}

// grid points per thread of the fill kernel: 16 on large grids (fewer partial sums, enough workgroups anyway), 1 on
// small ones -- at 121 x 61 sixteen points per thread are two workgroups running 32 serial exp() each (12 us)
int target_fill_per_thread(size_t P) { return P >= (static_cast<size_t>(1) << 22) ? kFillPerThread : (P >= (static_cast<size_t>(1) << 18) ? 4 : 1); }
int target_fill_blocks(size_t P)
{
  const size_t per_block = static_cast<size_t>(kBlock) * target_fill_per_thread(P);
  return static_cast<int>((P + per_block - 1) / per_block);
}

template <typename R>
hipError_t launch_target_fill_args(const R* d_coord, int nx, int ny, const GaussArgs<R>& ga, R* d_phi,
                                   R* d_partials, int K, R pi_lx, R pi_ly, R* d_cx, R* d_cy, hipStream_t s)
{
  const size_t P = static_cast<size_t>(nx) * ny;
  const int fill_blocks = target_fill_blocks(P);
  // d_cx == nullptr: the tables are current, fill only
  const int tab_blocks = d_cx != nullptr ? ((nx + ny) * K + kBlock - 1) / kBlock : 0;
  hipLaunchKernelGGL(target_fill_args_kernel<R>, dim3(fill_blocks + tab_blocks), dim3(kBlock), 0, s, d_coord, nx,
                     ny, ga, d_phi, d_partials, fill_blocks, target_fill_per_thread(P), K, pi_lx, pi_ly, d_cx, d_cy);
  return hipGetLastError();
}

size_t gaussian_phik_lds_bytes(int nx, int ny, int n_gauss, int K, size_t real_size)
{
  (void)nx;
  (void)ny;
  return real_size * (static_cast<size_t>(2 * kGaussAxisWaves * 4) + 2) * n_gauss * K;
}
template <typename R>
hipError_t launch_gaussian_phik(const R* d_coord, int nx, int ny, const GaussArgs<R>& ga, int K, R inv_lx, R inv_ly,
                                R* d_phik, R* d_mass, hipStream_t s, hipEvent_t stop)
{
  const size_t lds = gaussian_phik_lds_bytes(nx, ny, ga.n, K, sizeof(R));
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gaussian_phik_kernel<R>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  if (stop != nullptr) {
    hipExtLaunchKernelGGL(gaussian_phik_kernel<R>, dim3(1), dim3(kGaussBlock), lds, s, nullptr, stop, 0, d_coord, nx, ny, ga,
                          K, inv_lx, inv_ly, d_phik, d_mass);
  } else {
    hipLaunchKernelGGL(gaussian_phik_kernel<R>, dim3(1), dim3(kGaussBlock), lds, s, d_coord, nx, ny, ga, K, inv_lx, inv_ly,
                       d_phik, d_mass);
  }
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

// ---- geometry of the matrix-core streaming kernel -------------------------------------------
// columns per workgroup: 4 wavefronts x 16 x (columns per lane)
inline int stream_cols(int kind, int NT)
{
  const int spl = kind == kKindF64 ? 2 : (kind == kKindF32 ? 4 : (NT == 1 ? 8 : 4));
  const int lps = (kind == kKindCells || NT == 2) ? 1 : 2;
  return (kBlock / kWave) * 16 * spl * lps;
}
inline int stream_rows_per_tile(int nx, int ny, int kind, int NT)
{
  const int cols = stream_cols(kind, NT);
  const int col_tiles = (nx + cols - 1) / cols;
  // ~6 workgroups per CU in total; at least 64 rows so that the epilogue (16 NT^2 matrix
  // instructions per accumulator set) stays a small part of the tile
  int row_tiles = 1536 / col_tiles;
  if (row_tiles < 1) row_tiles = 1;
  int rpt = (ny + row_tiles - 1) / row_tiles;
  if (rpt < 64) rpt = 64;
  rpt = (rpt + 31) & ~31;  // whole pipeline stages (4 G rows, G <= 8)
  if (rpt > ny) rpt = ny;
  return rpt;
}

size_t spatial_work_elems(int nx, int ny, int K)
{
  // the largest tile count over the input kinds
  size_t tiles = 0;
  const int NT = K <= 16 ? 1 : 2;
  for (int kind = 0; kind < 3; ++kind) {
    const int cols = stream_cols(kind, NT);
    const int rpt = stream_rows_per_tile(nx, ny, kind, NT);
    const size_t t = static_cast<size_t>((nx + cols - 1) / cols) * ((ny + rpt - 1) / rpt);
    tiles = t > tiles ? t : tiles;
  }
  return tiles * K * K;
}

namespace
{
template <typename R, typename IN>
hipError_t launch_spatial_generic(const IN* d_in, int nx, int ny, int K, const R* d_cx, const R* d_cy,
                                  const R* d_lut, R* d_work, R* d_phik, hipStream_t s,
                                  const R* d_mass_partials = nullptr, int n_mass = 0, R* d_mass = nullptr,
                                  hipEvent_t stop = nullptr, bool norm_by_first = false)
{
  int mass_stride = 1;
  // `stop` (optional) is bound to the completion of the LAST launch (hipExtLaunchKernelGGL: the kernel's own
  // completion signal, no separate event-record packet on the stream)
#define EEA_LAUNCH_MAYBE_LAST(last, kernel, grid, block, lds, ...)                                      \
  do {                                                                                                  \
    if ((last) && stop != nullptr) hipExtLaunchKernelGGL(kernel, grid, block, lds, s, nullptr, stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, s, __VA_ARGS__);                                  \
  } while (0)
  constexpr bool kCells = !std::is_same<IN, R>::value;
  constexpr int kind = kCells ? kKindCells : (sizeof(R) == 8 ? kKindF64 : kKindF32);
  const int K2 = K * K;
  {
    const int NT = K <= 16 ? 1 : 2;
    const int cols = stream_cols(kind, NT);
    const int col_tiles = (nx + cols - 1) / cols;
    const int rpt = stream_rows_per_tile(nx, ny, kind, NT);
    const int row_tiles = (ny + rpt - 1) / rpt;
    // per-wavefront accumulator-set buffers (+ decode table), reused for the K^2 reduction
    size_t elems = static_cast<size_t>(kBlock / kWave) * 16 * NT * 17 + (kCells ? 256 : 0);
    if (elems < static_cast<size_t>(4) * K2) elems = static_cast<size_t>(4) * K2;
    const size_t lds = elems * sizeof(R);
    const dim3 grid(col_tiles, row_tiles);
    // one tile and the normalised form asked for: the streaming kernel finishes the job itself
    const bool direct = d_mass_partials != nullptr && !norm_by_first && col_tiles * row_tiles == 1;
    if (norm_by_first) {  // the normaliser is mode (0, 0) of the tiles' own partial sums
      d_mass_partials = d_work;
      n_mass = col_tiles * row_tiles;
      mass_stride = K2;
    }
    R* const d_direct = direct ? d_phik : nullptr;
    R* const d_direct_mass = direct ? d_mass : nullptr;
    if (NT == 1) {
      EEA_LAUNCH_MAYBE_LAST(direct, (spatial_stream_kernel<R, 1, IN>), grid, dim3(kBlock), lds, d_in, nx, ny, K, rpt,
                            d_cx, d_cy, d_lut, d_work, d_direct, d_direct_mass);
    } else {
      EEA_LAUNCH_MAYBE_LAST(direct, (spatial_stream_kernel<R, 2, IN>), grid, dim3(kBlock), lds, d_in, nx, ny, K, rpt,
                            d_cx, d_cy, d_lut, d_work, d_direct, d_direct_mass);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || direct) return e;
    if (d_mass_partials != nullptr) {
      EEA_LAUNCH_MAYBE_LAST(true, sum_partials_norm_kernel<R>, dim3((K2 + kModesPerSumBlock - 1) / kModesPerSumBlock),
                            dim3(kBlock), 0, d_work, col_tiles * row_tiles, K2, d_mass_partials, n_mass, d_phik, d_mass, mass_stride);
    } else {
      EEA_LAUNCH_MAYBE_LAST(true, sum_partials_kernel<R>, dim3((K2 + kModesPerSumBlock - 1) / kModesPerSumBlock),
                            dim3(kBlock), 0, d_work, col_tiles * row_tiles, K2, R(1), d_phik);
    }
    return hipGetLastError();
  }
#undef EEA_LAUNCH_MAYBE_LAST
}

// out[m] = raw[m] / raw[0]: the (0,0) mode of the un-normalised sums is the sum of the target
// values (cos 0 = 1), i.e. the normaliser of target.cpp:87
template <typename R>
__global__ __launch_bounds__(kBlock) void normalise_by_first_kernel(const R* __restrict__ raw, int K2,
                                                                    R* __restrict__ out)
{
  const R s = raw[0];
  for (int m = blockIdx.x * kBlock + threadIdx.x; m < K2; m += gridDim.x * kBlock) out[m] = raw[m] / s;
}
}  // namespace

template <typename R>
hipError_t launch_spatial_coeff(const R* d_phi, int nx, int ny, int K, const R* d_cx, const R* d_cy,
                                R* d_work, R* d_phik, hipStream_t s)
{
  return launch_spatial_generic<R, R>(d_phi, nx, ny, K, d_cx, d_cy, nullptr, d_work, d_phik, s);
}

template <typename R>
hipError_t launch_spatial_coeff_normalised(const R* d_phi_raw, int nx, int ny, int K, const R* d_cx, const R* d_cy,
                                           R* d_work, R* d_phik, const R* d_mass_partials, int n_mass, R* d_mass,
                                           hipStream_t s, hipEvent_t stop)
{
  return launch_spatial_generic<R, R>(d_phi_raw, nx, ny, K, d_cx, d_cy, nullptr, d_work, d_phik, s, d_mass_partials,
                                      n_mass, d_mass, stop);
}

template <typename R>
hipError_t launch_spatial_coeff_cells(const int8_t* d_occ, int nx, int ny, int K, const R* d_cx,
                                      const R* d_cy, const R* d_lut, R* d_work, R* d_raw, hipStream_t s, R* d_mass_out)
{
  // d_mass_out != nullptr: d_raw receives the NORMALISED coefficients (sums / mode (0, 0)'s sum, formed in the reduction launch
  // itself: two launches instead of three), *d_mass_out the normaliser
  return launch_spatial_generic<R, int8_t>(d_occ, nx, ny, K, d_cx, d_cy, d_lut, d_work, d_raw, s, nullptr, 0, d_mass_out, nullptr,
                                           d_mass_out != nullptr);
}

template <typename R>
hipError_t launch_normalise_by_first(const R* d_raw, int K2, R* d_out, hipStream_t s)
{
  hipLaunchKernelGGL(normalise_by_first_kernel<R>, dim3((K2 + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                     d_raw, K2, d_out);
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
  hipLaunchKernelGGL(sum_partials_kernel<R>, dim3((K2 + kModesPerSumBlock - 1) / kModesPerSumBlock), dim3(kBlock), 0, s,
                     d_work, static_cast<int>(blocks), K2, scale, d_out);
  return hipGetLastError();
}

#define EEA_INSTANTIATE(R)                                                                          \
  template hipError_t launch_axis_tables<R>(const R*, int, int, int, R, R, R*, R*, hipStream_t);    \
  template hipError_t launch_target_fill_args<R>(const R*, int, int, const GaussArgs<R>&, R*, R*,   \
                                                 int, R, R, R*, R*, hipStream_t);                   \
  template hipError_t launch_gaussian_phik<R>(const R*, int, int, const GaussArgs<R>&, int, R, R, R*, R*, hipStream_t, \
                                              hipEvent_t);                                          \
  template hipError_t launch_spatial_coeff_normalised<R>(const R*, int, int, int, const R*, const R*, R*, R*, \
                                                         const R*, int, R*, hipStream_t, hipEvent_t); \
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
  template hipError_t launch_spatial_coeff_cells<R>(const int8_t*, int, int, int, const R*,         \
                                                    const R*, const R*, R*, R*, hipStream_t, R*);   \
  template hipError_t launch_normalise_by_first<R>(const R*, int, R*, hipStream_t);                 \
  template hipError_t launch_point_coeff<R>(const R*, const R*, const R*, unsigned, int, R, R, R,   \
                                            R*, R*, hipStream_t);
EEA_INSTANTIATE(double)
EEA_INSTANTIATE(float)
#undef EEA_INSTANTIATE
}  // namespace eea
