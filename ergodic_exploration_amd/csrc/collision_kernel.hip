// Batched collision lookups for gfx950: Collision::collisionCheck (reference
// collision.cpp:126-243) on a GridMap (grid.cpp:96-184) and validate_control
// (numerics.hpp:273-330).  Integer work, one pose per lane; results are bit-exact with the
// reference's x86-64 build, including the wrap of negative world coordinates in world2Grid.
//
// Large batches go through an inflated map: the boolean result of the ring search does not depend on
// the visiting order (the search only stops at a hit), so it equals "some occupied cell lies at one
// of the offsets the Bresenham rings r_bnd..r_max visit and within r_col" -- a dilation of the
// occupied cells by a fixed offset set.  That set is enumerated on the host by the same ring walk,
// the occupied cells scatter a 1 to every centre that would see them, and a collision check becomes
// one byte lookup (a robot centred inside a small obstacle still reports no collision: cells closer
// than r_bnd are not in the set, as in the reference).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#include "../../include/ergodic_amd.h"
#include "common.hpp"

namespace eea
{
namespace
{
// static_cast<unsigned>(double) as x86-64 gcc compiles it (cvttsd2si r64, low 32 bits):
// negative and > 2^32 values wrap mod 2^32, out-of-range / NaN give 0.  The GPU's own
// double->u32 conversion saturates, so it is not used.
__device__ __forceinline__ unsigned cast_u32_x86(double v)
{
  if (!(v > -9.2233720368547758e18 && v < 9.2233720368547758e18)) return 0u;
  return static_cast<unsigned>(static_cast<unsigned long long>(static_cast<long long>(v)));
}

struct RingState
{
  int cx, cy, sqrd_obs;
};

// collision.cpp:216-243 (+ grid.cpp:96-100, 109-117, 177-184)
__device__ __forceinline__ bool check_cell(const CollisionParams& c, const int8_t* __restrict__ grid,
                                           RingState& st, unsigned cj, unsigned ci)
{
  if ((ci <= c.ysize - 1u) && (cj <= c.xsize - 1u)) {
    const double cell = static_cast<double>(grid[ci * c.xsize + cj]) / 100.0;
    if (!(cell < c.occupied_threshold)) {
      const unsigned ddx = static_cast<unsigned>(st.cx) - cj, ddy = static_cast<unsigned>(st.cy) - ci;
      const int sq = static_cast<int>(ddx * ddx + ddy * ddy);
      if (sq < st.sqrd_obs || st.sqrd_obs == -1) st.sqrd_obs = sq;
      if (sq <= c.r_col * c.r_col) return true;
    }
  }
  return false;
}

// collision.cpp:166-214
__device__ __forceinline__ bool bresenham_circle(const CollisionParams& c,
                                                 const int8_t* __restrict__ grid, RingState& st, int r)
{
  int x = -r, y = 0, err = 2 - 2 * r;
  while (x < 0) {
    if (check_cell(c, grid, st, static_cast<unsigned>(st.cx - x), static_cast<unsigned>(st.cy + y))) return true;
    if (check_cell(c, grid, st, static_cast<unsigned>(st.cx - y), static_cast<unsigned>(st.cy - x))) return true;
    if (check_cell(c, grid, st, static_cast<unsigned>(st.cx + x), static_cast<unsigned>(st.cy - y))) return true;
    if (check_cell(c, grid, st, static_cast<unsigned>(st.cx + y), static_cast<unsigned>(st.cy + x))) return true;
    r = err;
    if (r <= y) {
      y++;
      err += 2 * y + 1;
    }
    if (r > x || err > y) {
      x++;
      err += 2 * x + 1;
    }
  }
  return false;
}

// collision.cpp:126-164 with grid.cpp:143-159
__device__ __forceinline__ bool collision_check(const CollisionParams& c,
                                                const int8_t* __restrict__ grid, double px, double py)
{
  unsigned j = cast_u32_x86(floor((px - c.xmin) / c.resolution));
  unsigned i = cast_u32_x86(floor((py - c.ymin) / c.resolution));
  if (j == c.xsize) j--;
  if (i == c.ysize) i--;
  RingState st;
  st.cx = static_cast<int>(j);
  st.cy = static_cast<int>(i);
  st.sqrd_obs = -1;
  for (int r = c.r_bnd; r <= c.r_max; ++r) {
    if (bresenham_circle(c, grid, st, r)) return true;
  }
  return false;
}

// ---- inflated map ----------------------------------------------------------------------------
// hit map over the centres [-R, xsize + R) x [-R, ysize + R), R = r_col, row-major
struct HitMap
{
  const uint8_t* cells;
  int R, w, h;    // w = xsize + 2R, h = ysize + 2R
  uint8_t stamp;  // a cell is marked iff it holds this build's stamp (older stamps are stale, no clear)
};

// grid cell of a world point as collisionCheck derives it (grid.cpp:143-159), as signed ints
__device__ __forceinline__ void centre_of(const CollisionParams& c, double px, double py, int& cx, int& cy)
{
  unsigned j = cast_u32_x86(floor((px - c.xmin) / c.resolution));
  unsigned i = cast_u32_x86(floor((py - c.ymin) / c.resolution));
  if (j == c.xsize) j--;
  if (i == c.ysize) i--;
  cx = static_cast<int>(j);
  cy = static_cast<int>(i);
}

__device__ __forceinline__ bool hit_lookup(const HitMap& m, int cx, int cy)
{
  const int hx = cx + m.R, hy = cy + m.R;
  // centres further than R outside the grid see no cell at all (the unsigned cell indices of
  // collision.cpp:216-243 fall outside gridBounds)
  if (hx < 0 || hy < 0 || hx >= m.w || hy >= m.h) return false;
  return m.cells[static_cast<size_t>(hy) * m.w + hx] == m.stamp;
}

// occupied cells mark every centre that reaches them through one of the ring offsets
__global__ __launch_bounds__(kBlock) void inflate_kernel(const CollisionParams c, const int8_t* __restrict__ grid,
                                                         const short2* __restrict__ offsets, int n_off,
                                                         uint8_t* __restrict__ cells, int R, int w,
                                                         uint8_t stamp)
{
  // Occupied cells are few and the ring has hundreds of offsets: a lane that walked its own cell's ring alone made the
  // launch as long as that walk (27 us on the 240 x 120 demo map, profiles/r05_tick_kernels.txt).  The block first lists
  // its occupied cells, then ALL its threads walk the ring of each listed cell together.
  __shared__ unsigned s_list[kBlock];
  __shared__ unsigned s_n;
  if (threadIdx.x == 0) s_n = 0u;
  __syncthreads();
  const size_t q = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  const size_t n = static_cast<size_t>(c.xsize) * c.ysize;
  if (q < n) {
    const double cell = static_cast<double>(grid[q]) / 100.0;  // GridMap::getCell, grid.cpp:177-184
    if (!(cell < c.occupied_threshold)) s_list[atomicAdd(&s_n, 1u)] = threadIdx.x;  // checkCell tests !(cell < threshold)
  }
  __syncthreads();
  const unsigned cnt = s_n;
  for (unsigned k = 0; k < cnt; ++k) {
    const size_t qq = static_cast<size_t>(blockIdx.x) * kBlock + s_list[k];
    const int i = static_cast<int>(qq / c.xsize), j = static_cast<int>(qq - static_cast<size_t>(i) * c.xsize);
    for (int o = threadIdx.x; o < n_off; o += kBlock) {
      const short2 d = offsets[o];  // cell = centre + d
      cells[static_cast<size_t>(i - d.y + R) * w + (j - d.x + R)] = stamp;  // same value from every writer
    }
  }
}

__global__ __launch_bounds__(kBlock) void collision_check_map_kernel(const CollisionParams c, const HitMap m,
                                                                     const double* __restrict__ pose,
                                                                     unsigned P, int* __restrict__ hit)
{
  const unsigned q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= P) return;
  int cx, cy;
  centre_of(c, pose[3 * static_cast<size_t>(q)], pose[3 * static_cast<size_t>(q) + 1], cx, cy);
  hit[q] = hit_lookup(m, cx, cy) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void collision_check_kernel(const CollisionParams c,
                                                                 const int8_t* __restrict__ grid,
                                                                 const double* __restrict__ pose,
                                                                 unsigned P, int* __restrict__ hit)
{
  const unsigned q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= P) return;
  hit[q] = collision_check(c, grid, pose[3 * static_cast<size_t>(q)], pose[3 * static_cast<size_t>(q) + 1]) ? 1 : 0;
}

// Round 6: the constant-twist rollouts of validate_control / the dynamic window are instruction-issue bound (20 dependent steps
// per lane of sincos + normalize_angle_PI + a byte lookup: ~245 instructions per step with the device library's sincos and the
// literal wrap, whose quotient is an fp64 DIVISION).  EEA_TICK_FAST_TRIG (default): sin / cos by the engine's own sincospi_r
// (exact reduction, max error 1.9e-16: common.hpp) of theta / pi, and the wrap's quotient by a multiplication with 1 / (2 pi) --
// a quotient that lands on the other side of an integer is repaired by the wrap's own range fixes, everything else is bitwise
// the literal form.  The outputs the reference pins are integer decisions (collision verdicts, the chosen sample), which
// tests/test_gpu_collision_parity.py / test_gpu_dwa_parity.py / test_gpu_fleet_tick.py check bit for bit against the oracle.
#ifndef EEA_TICK_FAST_TRIG
#define EEA_TICK_FAST_TRIG 1
#endif
__device__ __forceinline__ double wrap_pi_d(double rad)
{
#if EEA_TICK_FAST_TRIG
  const double pi = kPi, two_pi = 2.0 * kPi;
  const double q = floor((rad + pi) * (1.0 / (2.0 * kPi)));
  rad = (rad + pi) - q * two_pi;
  if (rad < 0.0) rad += two_pi;
  if (!(rad < two_pi)) rad -= two_pi;
  return rad - pi;
#else
  return wrap_pi<double>(rad);
#endif
}
__device__ __forceinline__ void tick_sincos(double th, double* s, double* c)
{
#if EEA_TICK_FAST_TRIG
  sincospi_r<double>(th * (1.0 / kPi), s, c);
#else
  sincos(th, s, c);
#endif
}

// one collision test through the inflated map (MAP) or by the ring search
template <bool MAP>
__device__ __forceinline__ bool pose_collides(const CollisionParams& c, const int8_t* __restrict__ grid,
                                              const HitMap& m, double px, double py)
{
  if (MAP) {
    int cx, cy;
    centre_of(c, px, py, cx, cy);
    return hit_lookup(m, cx, cy);
  }
  return collision_check(c, grid, px, py);
}

// numerics.hpp:273-330
template <bool MAP>
__global__ __launch_bounds__(kBlock) void validate_control_kernel(const CollisionParams c, const HitMap m,
                                                                  const int8_t* __restrict__ grid,
                                                                  const double* __restrict__ x0,
                                                                  const double* __restrict__ u,
                                                                  double dt, unsigned steps, unsigned P,
                                                                  int* __restrict__ valid)
{
  const unsigned q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= P) return;
  double x = x0[3 * static_cast<size_t>(q)], y = x0[3 * static_cast<size_t>(q) + 1],
         th = x0[3 * static_cast<size_t>(q) + 2];
  const double u0 = u[3 * static_cast<size_t>(q)], u1 = u[3 * static_cast<size_t>(q) + 1],
               u2 = u[3 * static_cast<size_t>(q) + 2];
  // body-frame displacement of one step is the same every step (constant twist)
  double d0, d1, d2;
  if (fabs(u2 - 0.0) < 1.0e-12) {
    d0 = u0 * dt;
    d1 = u1 * dt;
    d2 = 0.0;
  } else {
    const double vb0 = u0 * dt, vb1 = u1 * dt, vb2 = u2 * dt;
    double s, cc;
    tick_sincos(vb2, &s, &cc);
    d0 = (vb0 * s + vb1 * (cc - 1.0)) / vb2;
    d1 = (vb1 * s + vb0 * (1.0 - cc)) / vb2;
    d2 = vb2;
  }
  int ok = 1;
  for (unsigned i = 0; i < steps; ++i) {
    double s, cc;
    tick_sincos(th, &s, &cc);
    x = x + (cc * d0 + (-s) * d1);
    y = y + (s * d0 + cc * d1);
    th = wrap_pi_d(th + d2);
    if (pose_collides<MAP>(c, grid, m, x, y)) {
      ok = 0;
      break;
    }
  }
  valid[q] = ok;
}
// DynamicWindow::control (dynamic_window.cpp:92-286): one workgroup per robot, one lane per
// velocity sample; lane 0 then takes the first strict minimum in the reference's loop order.
constexpr int kDwaBlock = 128;
// eea_tick_batch (FLEET): the robots of a fleet tick.  A robot whose twist validate_control accepted does nothing here
// (its source is recorded); the others run the dynamic window in THEIR mode -- a follower towards its own twist
// (exploration.hpp:243-251), the rest along their optTraj (:254-277) -- and lane 0 updates u, follow_dwa and i.
struct FleetTick
{
  const int* valid;
  int* follow;
  unsigned* count;
  double* u;
  int* source;
};
template <bool MAP, bool FLEET>
__global__ __launch_bounds__(kDwaBlock) void dwa_control_kernel(const CollisionParams c, const DwaParams d,
                                                               const HitMap m,
                                                               const int8_t* __restrict__ grid,
                                                               const double* __restrict__ x0s,
                                                               const double* __restrict__ vbs,
                                                               const double* __restrict__ vrefs,
                                                               const double* __restrict__ xt_refs,
                                                               unsigned n_ref, double dt_ref,
                                                               double* __restrict__ u_opt,
                                                               int* __restrict__ found, const FleetTick ft)
{
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* const s_cost = reinterpret_cast<double*>(smem_raw);
  const unsigned r = blockIdx.x;
  bool following = false;
  if (FLEET) {
    following = ft.follow[r] != 0;  // (after step 1 of the tick)
    if (ft.valid[r] != 0) {         // workgroup-uniform
      if (threadIdx.x == 0 && ft.source != nullptr) ft.source[r] = following ? 1 : 0;
      return;
    }
    if (following) {
      xt_refs = nullptr;
      vrefs = ft.u;
    }
    u_opt = ft.u;
  }
  const double* const x0 = x0s + 3 * static_cast<size_t>(r);
  const double* const vb = vbs + 3 * static_cast<size_t>(r);
  const unsigned nsamp = d.ns[0] * d.ns[1] * d.ns[2];
  constexpr double kMax = 1.7976931348623157e308;

  // window (dynamic_window.cpp:191-235)
  double lower[3], delta[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lower[a] = fmax(vb[a] - d.acc_lim[a] * d.acc_dt, d.vmin[a]);
    const double upper = fmin(vb[a] + d.acc_lim[a] * d.acc_dt, d.vmax[a]);
    delta[a] = (d.ns[a] > 1) ? (upper - lower[a]) / static_cast<double>(d.ns[a] - 1) : 0.0;
  }
  const double tf = static_cast<double>(n_ref) * dt_ref;
  const double* const xr = (xt_refs != nullptr) ? xt_refs + 3 * static_cast<size_t>(n_ref) * r : nullptr;

  for (unsigned sidx = threadIdx.x; sidx < nsamp; sidx += blockDim.x) {
    // sample (i, j, k) in the reference's loop order; values by repeated += like the reference
    const unsigned k = sidx % d.ns[2], j = (sidx / d.ns[2]) % d.ns[1], i = sidx / (d.ns[2] * d.ns[1]);
    double u0 = lower[0], u1 = lower[1], u2 = lower[2];
    for (unsigned q = 0; q < i; ++q) u0 += delta[0];
    for (unsigned q = 0; q < j; ++q) u1 += delta[1];
    for (unsigned q = 0; q < k; ++q) u2 += delta[2];

    // constant-twist displacement in the body frame (numerics.hpp:273-297)
    double d0, d1, d2;
    if (fabs(u2 - 0.0) < 1.0e-12) {
      d0 = u0 * d.dt;
      d1 = u1 * d.dt;
      d2 = 0.0;
    } else {
      const double vb0 = u0 * d.dt, vb1 = u1 * d.dt, vb2 = u2 * d.dt;
      double sn, cs;
      tick_sincos(vb2, &sn, &cs);
      d0 = (vb0 * sn + vb1 * (cs - 1.0)) / vb2;
      d1 = (vb1 * sn + vb0 * (1.0 - cs)) / vb2;
      d2 = vb2;
    }
    double x = x0[0], y = x0[1], th = x0[2];
    double cost = 0.0, t = 0.0;
    bool hit = false;
    for (unsigned st = 0; st < d.steps; ++st) {
      double sn, cs;
      tick_sincos(th, &sn, &cs);
      x = x + (cs * d0 + (-sn) * d1);
      y = y + (sn * d0 + cs * d1);
      th = wrap_pi_d(th + d2);
      if (pose_collides<MAP>(c, grid, m, x, y)) {
        hit = true;
        break;
      }
      if (xr != nullptr) {
        const unsigned jj = cast_u32_x86(round(static_cast<double>(n_ref - 1) * t / tf));
        const double ex = xr[3 * jj + 0] - x, ey = xr[3 * jj + 1] - y;
        cost += sqrt(ex * ex + ey * ey);
        cost += fabs(wrap_pi_d(wrap_pi_d(xr[3 * jj + 2]) - th));
        t += d.dt;
      }
    }
    if (hit) {
      cost = kMax;
    } else if (xr == nullptr) {
      // (FLEET: vrefs is the robot's own u, read here by every lane before lane 0 overwrites it behind the barrier)
      const double* const vref = vrefs + 3 * static_cast<size_t>(r);
      const double e0 = vref[0] - u0, e1 = vref[1] - u1, e2 = vref[2] - u2;
      cost = (e0 * e0 + e2 * e2) + e1 * e1;
    }
    s_cost[sidx] = cost;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double best = kMax;
    unsigned arg = 0;
    for (unsigned sidx = 0; sidx < nsamp; ++sidx) {
      if (s_cost[sidx] < best) {
        best = s_cost[sidx];
        arg = sidx;
      }
    }
    const bool ok = !(fabs(best - kMax) < 1.0e-12);
    double o0 = 0.0, o1 = 0.0, o2 = 0.0;  // u_opt stays zero when nothing beat the initial maximum
    if (best < kMax) {
      const unsigned k = arg % d.ns[2], j = (arg / d.ns[2]) % d.ns[1], i = arg / (d.ns[2] * d.ns[1]);
      o0 = lower[0];
      o1 = lower[1];
      o2 = lower[2];
      for (unsigned q = 0; q < i; ++q) o0 += delta[0];
      for (unsigned q = 0; q < j; ++q) o1 += delta[1];
      for (unsigned q = 0; q < k; ++q) o2 += delta[2];
    }
    u_opt[3 * static_cast<size_t>(r) + 0] = o0;
    u_opt[3 * static_cast<size_t>(r) + 1] = o1;
    u_opt[3 * static_cast<size_t>(r) + 2] = o2;
    if (FLEET) {
      if (following) {  // the followed twist led into a collision: whatever the window found, control() replans next tick
        ft.follow[r] = 0;
        if (ft.source != nullptr) ft.source[r] = 3;
      } else {          // follow the solution if there is one
        ft.follow[r] = ok ? 1 : 0;
        if (ok) ft.count[r] = 0u;
        if (ft.source != nullptr) ft.source[r] = 2;
      }
    } else {
      found[r] = ok ? 1 : 0;
    }
  }
}

// step 1 of the fleet tick (exploration.hpp:223-228): a follower counts a step and replans after dwa_steps of them
__global__ __launch_bounds__(256) void tick_begin_kernel(int* __restrict__ follow, unsigned* __restrict__ count,
                                                         int* __restrict__ skip, unsigned dwa_steps, unsigned P)
{
  const unsigned r = blockIdx.x * 256 + threadIdx.x;
  if (r >= P) return;
  int f = follow[r];
  if (f != 0) {
    const unsigned i = count[r] + 1u;
    count[r] = i;
    f = (i != dwa_steps) ? 1 : 0;
    follow[r] = f;
  }
  skip[r] = f;
}
}  // namespace

namespace
{
// offsets (cell - centre) the ring search r_bnd..r_max can report a collision at: the walk of
// collision.cpp:166-214, keeping the cells within r_col (collision.cpp:239)
std::vector<short2> ring_offsets(const CollisionParams& c)
{
  std::vector<std::pair<int, int>> pts;
  for (int r0 = c.r_bnd; r0 <= c.r_max; ++r0) {
    int r = r0, x = -r0, y = 0, err = 2 - 2 * r0;
    while (x < 0) {
      pts.emplace_back(-x, y);
      pts.emplace_back(-y, -x);
      pts.emplace_back(x, -y);
      pts.emplace_back(y, x);
      r = err;
      if (r <= y) {
        y++;
        err += 2 * y + 1;
      }
      if (r > x || err > y) {
        x++;
        err += 2 * x + 1;
      }
    }
  }
  std::sort(pts.begin(), pts.end());
  pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
  std::vector<short2> out;
  for (const auto& p : pts) {
    if (p.first * p.first + p.second * p.second <= c.r_col * c.r_col) {
      short2 o;
      o.x = static_cast<short>(p.first);
      o.y = static_cast<short>(p.second);
      out.push_back(o);
    }
  }
  return out;
}

// The offset list depends on three small integers only: built once per (device, radii) and kept for
// the life of the process (a few KB), so that building a map needs no host synchronisation.
struct OffsetEntry
{
  int device, r_bnd, r_col, r_max, n;
  short2* d_offsets;
};
std::mutex g_offsets_mutex;
std::vector<OffsetEntry> g_offsets;

hipError_t device_offsets(const CollisionParams& c, const short2** d_out, int* n_out)
{
  int device = 0;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_offsets_mutex);
  for (const OffsetEntry& o : g_offsets) {
    if (o.device == device && o.r_bnd == c.r_bnd && o.r_col == c.r_col && o.r_max == c.r_max) {
      *d_out = o.d_offsets;
      *n_out = o.n;
      return hipSuccess;
    }
  }
  const std::vector<short2> off = ring_offsets(c);
  OffsetEntry ent{ device, c.r_bnd, c.r_col, c.r_max, static_cast<int>(off.size()), nullptr };
  if (!off.empty()) {
    e = hipMalloc(reinterpret_cast<void**>(&ent.d_offsets), off.size() * sizeof(short2));
    if (e != hipSuccess) return e;
    e = hipMemcpy(ent.d_offsets, off.data(), off.size() * sizeof(short2), hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
  }
  g_offsets.push_back(ent);
  *d_out = ent.d_offsets;
  *n_out = ent.n;
  return hipSuccess;
}

// The map buffer of a (device, stream) pair is kept between calls: launches on one stream are
// ordered, so a buffer is never written while an earlier call still reads it.  Every build marks
// with a fresh stamp (1..255), which makes the marks of earlier builds stale without clearing the
// buffer; it is cleared when the stamps wrap.  A build is then a single scatter kernel.
struct MapBuffer
{
  int device;
  hipStream_t stream;
  uint8_t* cells;
  size_t cap;
  unsigned stamp;
  // what the current stamp was built from (eea_tick_io::grid_epoch != 0: a later build of the same map is skipped)
  const int8_t* built_grid = nullptr;
  unsigned long long built_epoch = 0;
  CollisionParams built_params{};
};
std::mutex g_maps_mutex;
std::vector<MapBuffer> g_maps;
constexpr size_t kMaxMapBuffers = 64;

struct MapScratch
{
  void* async_cells = nullptr;  // stream-ordered allocation (only when the buffer table is full)
  HitMap map{ nullptr, 0, 0, 0, 0 };
};

bool same_params(const CollisionParams& a, const CollisionParams& b)
{
  return a.xmin == b.xmin && a.ymin == b.ymin && a.resolution == b.resolution && a.xsize == b.xsize && a.ysize == b.ysize &&
         a.r_bnd == b.r_bnd && a.r_col == b.r_col && a.r_max == b.r_max && a.occupied_threshold == b.occupied_threshold;
}

// epoch != 0: the caller vouches that (d_grid, epoch) names one content -- the inflated map of the last build on this
// stream is reused when it was built from the same (grid, epoch, parameters)
hipError_t build_hit_map(const CollisionParams& c, const int8_t* d_grid, MapScratch& sc, hipStream_t s,
                         unsigned long long epoch = 0)
{
  if (c.r_col < 0 || c.r_col > 8192 || c.r_max < c.r_bnd) return hipErrorInvalidValue;
  const short2* d_off = nullptr;
  int n_off = 0;
  hipError_t e = device_offsets(c, &d_off, &n_off);
  if (e != hipSuccess) return e;
  const int R = c.r_col;
  const int w = static_cast<int>(c.xsize) + 2 * R, h = static_cast<int>(c.ysize) + 2 * R;
  const size_t bytes = static_cast<size_t>(w) * h;
  int device = 0;
  e = hipGetDevice(&device);
  if (e != hipSuccess) return e;

  uint8_t* cells = nullptr;
  unsigned stamp = 1;
  bool from_table = false;  // the map lives in a per-stream buffer of the table (its cache key is recorded after the launch)
  {
    std::lock_guard<std::mutex> lock(g_maps_mutex);
    MapBuffer* buf = nullptr;
    for (MapBuffer& b : g_maps) {
      if (b.device == device && b.stream == s) buf = &b;
    }
    if (buf == nullptr && g_maps.size() < kMaxMapBuffers) {
      g_maps.push_back(MapBuffer{ device, s, nullptr, 0, 0 });
      buf = &g_maps.back();
    }
    if (buf != nullptr) {
      if (buf->cap < bytes) {
        if (buf->cells) (void)hipFree(buf->cells);  // waits for the kernels that still use it
        buf->cells = nullptr;
        buf->cap = 0;
        e = hipMalloc(reinterpret_cast<void**>(&buf->cells), bytes);
        if (e != hipSuccess) return e;
        buf->cap = bytes;
        buf->stamp = 0;
        e = hipMemsetAsync(buf->cells, 0, bytes, s);
        if (e != hipSuccess) return e;
      }
      if (epoch != 0 && buf->stamp != 0 && buf->built_grid == d_grid && buf->built_epoch == epoch &&
          same_params(buf->built_params, c)) {
        sc.map.cells = buf->cells;
        sc.map.R = R;
        sc.map.w = w;
        sc.map.h = h;
        sc.map.stamp = static_cast<uint8_t>(buf->stamp);
        return hipSuccess;
      }
      if (++buf->stamp > 255u) {
        e = hipMemsetAsync(buf->cells, 0, buf->cap, s);
        if (e != hipSuccess) return e;
        buf->stamp = 1;
      }
      // (the cache key is recorded only once the dilation launch below has succeeded -- ADVICE r05: after a failed launch the
      // next tick with the same (grid, epoch) must not validate against a map that was never stamped)
      buf->built_grid = nullptr;
      buf->built_epoch = 0;
      cells = buf->cells;
      stamp = buf->stamp;
      from_table = true;
    }
  }
  if (cells == nullptr) {
    e = hipMallocAsync(&sc.async_cells, bytes, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(sc.async_cells, 0, bytes, s);
    if (e != hipSuccess) return e;
    cells = static_cast<uint8_t*>(sc.async_cells);
  }
  if (n_off > 0) {
    const size_t n = static_cast<size_t>(c.xsize) * c.ysize;
    hipLaunchKernelGGL(inflate_kernel, dim3(static_cast<unsigned>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, c,
                       d_grid, d_off, n_off, cells, R, w, static_cast<uint8_t>(stamp));
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (from_table && epoch != 0) {  // valid for work ordered behind this launch (the caller's stream contract)
    std::lock_guard<std::mutex> lock(g_maps_mutex);  // (the table may have grown: look the buffer up again)
    for (MapBuffer& b : g_maps) {
      if (b.device == device && b.stream == s && b.cells == cells) {
        b.built_grid = d_grid;
        b.built_epoch = epoch;
        b.built_params = c;
      }
    }
  }
  sc.map.cells = cells;
  sc.map.R = R;
  sc.map.w = w;
  sc.map.h = h;
  sc.map.stamp = static_cast<uint8_t>(stamp);
  return hipSuccess;
}

void release_hit_map(MapScratch& sc, hipStream_t s)
{
  if (sc.async_cells) (void)hipFreeAsync(sc.async_cells, s);
  sc.async_cells = nullptr;
}

// Inflated map or ring search?  A ring search is a chain of ~200 dependent byte loads per pose
// (~40 us on MI355X, and the steps of one rollout follow each other in one lane), the map costs a
// fixed ~25 us of stream operations plus a pass over the grid (measured: profiles/r01_tick_kernels.txt).
// EEA_OPT_COLLISION_IMPL = 1 / 2 forces one of them.
bool use_hit_map(size_t poses, unsigned sequential_steps, const CollisionParams& c)
{
  const int forced = option(EEA_OPT_COLLISION_IMPL);
  if (forced != 0) return forced == 2;
  if (poses >= 4096) return true;
  const double cells = static_cast<double>(c.xsize) * c.ysize;
  const double t_map_us = 25.0 + 2.0e-6 * cells, t_ring_us = 40.0 * (sequential_steps ? sequential_steps : 1u);
  return t_map_us < t_ring_us;
}
}  // namespace

void release_collision_caches()
{
  int current = 0;
  const bool have_device = hipGetDevice(&current) == hipSuccess;
  {
    std::lock_guard<std::mutex> lock(g_maps_mutex);
    for (MapBuffer& b : g_maps) {
      if (b.cells != nullptr && hipSetDevice(b.device) == hipSuccess) (void)hipFree(b.cells);  // waits for its users
    }
    g_maps.clear();
  }
  {
    std::lock_guard<std::mutex> lock(g_offsets_mutex);
    for (OffsetEntry& o : g_offsets) {
      if (o.d_offsets != nullptr && hipSetDevice(o.device) == hipSuccess) (void)hipFree(o.d_offsets);
    }
    g_offsets.clear();
  }
  if (have_device) (void)hipSetDevice(current);
}

hipError_t launch_dwa_control(const CollisionParams& c, const DwaParams& d, const int8_t* d_grid,
                              const double* d_x0, const double* d_vb, const double* d_vref,
                              const double* d_xt_ref, unsigned n_ref, double dt_ref, unsigned P,
                              double* d_u_opt, int* d_found, hipStream_t s)
{
  if (P == 0) return hipSuccess;
  const size_t nsamp = static_cast<size_t>(d.ns[0]) * d.ns[1] * d.ns[2];
  const size_t lds = sizeof(double) * nsamp;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  // one lane per velocity sample: a single wavefront when the window has at most 64 samples
  const unsigned block = nsamp <= 64 ? 64 : kDwaBlock;
  if (use_hit_map(static_cast<size_t>(P) * nsamp * d.steps, d.steps, c)) {
    MapScratch sc;
    hipError_t e = build_hit_map(c, d_grid, sc, s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL((dwa_control_kernel<true, false>), dim3(P), dim3(block), lds, s, c, d, sc.map, d_grid, d_x0, d_vb,
                         d_vref, d_xt_ref, n_ref, dt_ref, d_u_opt, d_found, FleetTick{});
      e = hipGetLastError();
    }
    release_hit_map(sc, s);
    return e;
  }
  hipLaunchKernelGGL((dwa_control_kernel<false, false>), dim3(P), dim3(block), lds, s, c, d, HitMap{ nullptr, 0, 0, 0, 0 },
                     d_grid, d_x0, d_vb, d_vref, d_xt_ref, n_ref, dt_ref, d_u_opt, d_found, FleetTick{});
  return hipGetLastError();
}

hipError_t launch_tick_begin(int* d_follow, unsigned* d_count, int* d_skip, unsigned dwa_steps, unsigned P, hipStream_t s)
{
  if (P == 0) return hipSuccess;
  hipLaunchKernelGGL(tick_begin_kernel, dim3((P + 255) / 256), dim3(256), 0, s, d_follow, d_count, d_skip, dwa_steps, P);
  return hipGetLastError();
}

// Steps 3 and 4 of a fleet tick: validate_control of every robot's twist, then the dynamic window of the robots whose twist
// was rejected (one workgroup per robot as above; robots with a valid twist leave at once).  ONE inflated map serves both
// launches (and, with grid_epoch != 0, the ticks that follow on an unchanged grid).  The cost model takes every robot as
// one that searches (the worst case: the inflated map pays for itself at a fraction of that).
hipError_t launch_validate_and_dwa_fleet(const CollisionParams& c, const DwaParams& d, const int8_t* d_grid,
                                         unsigned long long grid_epoch, const double* d_x0, const double* d_vb,
                                         const double* d_traj, unsigned n_ref, double dt_ref, double val_dt, unsigned val_steps,
                                         int* d_valid, int* d_follow, unsigned* d_count, double* d_u, int* d_source, unsigned P,
                                         hipStream_t s)
{
  if (P == 0) return hipSuccess;
  const size_t nsamp = static_cast<size_t>(d.ns[0]) * d.ns[1] * d.ns[2];
  const size_t lds = sizeof(double) * nsamp;
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const unsigned block = nsamp <= 64 ? 64 : kDwaBlock;
  const FleetTick ft{ d_valid, d_follow, d_count, d_u, d_source };
  const dim3 vgrid((P + kBlock - 1) / kBlock);
  if (use_hit_map(static_cast<size_t>(P) * (val_steps + nsamp * d.steps), d.steps, c)) {
    MapScratch sc;
    hipError_t e = build_hit_map(c, d_grid, sc, s, grid_epoch);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(validate_control_kernel<true>, vgrid, dim3(kBlock), 0, s, c, sc.map, d_grid, d_x0, d_u, val_dt,
                         val_steps, P, d_valid);
      hipLaunchKernelGGL((dwa_control_kernel<true, true>), dim3(P), dim3(block), lds, s, c, d, sc.map, d_grid, d_x0, d_vb,
                         static_cast<const double*>(nullptr), d_traj, n_ref, dt_ref, static_cast<double*>(nullptr),
                         static_cast<int*>(nullptr), ft);
      e = hipGetLastError();
    }
    release_hit_map(sc, s);
    return e;
  }
  hipLaunchKernelGGL(validate_control_kernel<false>, vgrid, dim3(kBlock), 0, s, c, HitMap{ nullptr, 0, 0, 0, 0 }, d_grid, d_x0,
                     d_u, val_dt, val_steps, P, d_valid);
  hipLaunchKernelGGL((dwa_control_kernel<false, true>), dim3(P), dim3(block), lds, s, c, d, HitMap{ nullptr, 0, 0, 0, 0 },
                     d_grid, d_x0, d_vb, static_cast<const double*>(nullptr), d_traj, n_ref, dt_ref,
                     static_cast<double*>(nullptr), static_cast<int*>(nullptr), ft);
  return hipGetLastError();
}

hipError_t launch_collision_check(const CollisionParams& c, const int8_t* d_grid,
                                  const double* d_pose, unsigned P, int* d_hit, hipStream_t s)
{
  if (P == 0) return hipSuccess;
  const dim3 grid((P + kBlock - 1) / kBlock);
  if (use_hit_map(P, 1, c)) {
    MapScratch sc;
    hipError_t e = build_hit_map(c, d_grid, sc, s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(collision_check_map_kernel, grid, dim3(kBlock), 0, s, c, sc.map, d_pose, P, d_hit);
      e = hipGetLastError();
    }
    release_hit_map(sc, s);
    return e;
  }
  hipLaunchKernelGGL(collision_check_kernel, grid, dim3(kBlock), 0, s, c, d_grid, d_pose, P, d_hit);
  return hipGetLastError();
}

// integrate_twist (numerics.hpp:273-297) for P poses: x + Rot(theta) dq_b, the heading NOT wrapped (the reference's callers
// normalise it: validate_control :325, the nodes' motion updates).  One thread per pose; this translation unit is compiled
// without FMA contraction, the sums in the reference's order.
__global__ __launch_bounds__(kBlock) void integrate_twist_kernel(const double* __restrict__ x0, const double* __restrict__ u, double dt,
                                                                 unsigned P, double* __restrict__ out, int wrap)
{
  const unsigned q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= P) return;
  const double x = x0[3 * static_cast<size_t>(q)], y = x0[3 * static_cast<size_t>(q) + 1], th = x0[3 * static_cast<size_t>(q) + 2];
  const double u0 = u[3 * static_cast<size_t>(q)], u1 = u[3 * static_cast<size_t>(q) + 1], u2 = u[3 * static_cast<size_t>(q) + 2];
  double d0, d1, d2;
  if (fabs(u2 - 0.0) < 1.0e-12) {
    d0 = u0 * dt;
    d1 = u1 * dt;
    d2 = 0.0;
  } else {
    const double vb0 = u0 * dt, vb1 = u1 * dt, vb2 = u2 * dt;
    double s, cc;
    sincos(vb2, &s, &cc);  // (the device library's: one call per robot and tick, kept closest to libm)
    d0 = (vb0 * s + vb1 * (cc - 1.0)) / vb2;
    d1 = (vb1 * s + vb0 * (1.0 - cc)) / vb2;
    d2 = vb2;
  }
  double s, cc;
  sincos(th, &s, &cc);
  out[3 * static_cast<size_t>(q)] = x + (cc * d0 + (-s) * d1);
  out[3 * static_cast<size_t>(q) + 1] = y + (s * d0 + cc * d1);
  out[3 * static_cast<size_t>(q) + 2] = wrap ? wrap_pi_d(th + d2) : th + d2;
}

hipError_t launch_integrate_twist(const double* d_x0, const double* d_u, double dt, unsigned P, double* d_out, bool wrap, hipStream_t s)
{
  if (P == 0) return hipSuccess;
  hipLaunchKernelGGL(integrate_twist_kernel, dim3((P + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d_x0, d_u, dt, P, d_out, wrap ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_validate_control(const CollisionParams& c, const int8_t* d_grid,
                                   const double* d_x0, const double* d_u, double dt, unsigned steps,
                                   unsigned P, int* d_valid, hipStream_t s)
{
  if (P == 0) return hipSuccess;
  const dim3 grid((P + kBlock - 1) / kBlock);
  if (use_hit_map(static_cast<size_t>(P) * steps, steps, c)) {
    MapScratch sc;
    hipError_t e = build_hit_map(c, d_grid, sc, s);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(validate_control_kernel<true>, grid, dim3(kBlock), 0, s, c, sc.map, d_grid, d_x0, d_u, dt,
                         steps, P, d_valid);
      e = hipGetLastError();
    }
    release_hit_map(sc, s);
    return e;
  }
  hipLaunchKernelGGL(validate_control_kernel<false>, grid, dim3(kBlock), 0, s, c, HitMap{ nullptr, 0, 0, 0, 0 }, d_grid,
                     d_x0, d_u, dt, steps, P, d_valid);
  return hipGetLastError();
}
}  // namespace eea
