// C ABI (include/ergodic_amd.h) of the MI355X ergodic receding-horizon engine: argument
// checking, device-memory ownership and kernel dispatch.  All arithmetic of the hot path
// runs in the HIP kernels (control_kernel.hip, phik_kernel.hip, collision_kernel.hip); the
// host code here only prepares their inputs the way the reference's host code does
// (steps_ truncation, grid coordinates by accumulation, covariance inverse).
#include "../../include/ergodic_amd.h"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "abi_util.hpp"
#include "common.hpp"

namespace
{
using eea::fail;

// process-wide dispatch options (eea_set_option); index = EEA_OPT_*
std::atomic<int> g_options[EEA_OPT_COUNT] = { { 0 }, { 0 }, { 0 }, { 1 }, { 0 }, { 0 }, { 0 }, { 250 } };
}  // namespace

namespace eea
{
int option(int id) { return (id >= 0 && id < EEA_OPT_COUNT) ? g_options[id].load(std::memory_order_relaxed) : 0; }
}  // namespace eea

namespace
{
using eea::fail;

// grid.hpp:61-64 of the reference (x86-64 cast semantics do not matter here: non-negative)
unsigned axis_length(double lower, double upper, double resolution)
{
  return static_cast<unsigned>(std::round((upper - lower) / resolution));
}

struct DevBuf
{
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes)
  {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    const hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release()
  {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// host mailbox of the single-agent path, device-visible (zero-copy over PCIe)
struct Mailbox
{
  double pose[3];  // viewed as R[3]
  double u0[3];    // viewed as R[3]
  int status;
  int done;  // completion sequence number written by the kernel (polled by eea_control)
};
}  // namespace

struct eea_engine
{
  eea_config cfg;
  int T = 0, K = 0, K2 = 0;
  bool f32 = false;
  size_t rs = 8;  // sizeof(real)
  int chunk = 128;

  // Basis state (basis_.lx_, ly_ start at 0: ergodic_control.hpp:208)
  double lx = 0.0, ly = 0.0, map_x = 0.0, map_y = 0.0;
  bool have_phik = false;
  std::vector<double> mu, sigma;  // target Gaussians, map frame
  bool have_gauss = false;

  DevBuf d_phik, d_lamdak;
  // phi grid of the last rebuild
  unsigned nx = 0, ny = 0;
  bool have_fill_grid = false;  // d_phi holds the Target::fill output of the last rebuild ...
  bool phi_is_raw = false;      // ... un-normalised: d_sum[0] is its mass (eea_get_target_grid divides)
  // ... or WILL hold it: a Gaussian rebuild computes phi_k from the per-axis factors without the grid; the grid is
  // filled when eea_get_target_grid asks for it, from the Gaussians as they were at the rebuild (Fourier frame)
  bool fill_deferred = false;
  std::vector<double> fill_gauss;  // [n][4] of the last rebuild: mean - map_pos, diag(cov_inv)
  DevBuf d_phi, d_axis, d_cx, d_cy, d_work, d_gauss, d_sum;
  // accumulated grid coordinates 0, res, res + res, ... (ergodic_control.hpp:387-408): one device array serves
  // both axes of every grid of this engine (the resolution is fixed); grown on demand
  unsigned axis_n = 0;
  // key of the cos tables in d_cx / d_cy: a repeated rebuild / row tile on the same grid and domain reuses them
  unsigned tab_nx = 0, tab_ny = 0;
  double tab_lx = 0.0, tab_ly = 0.0;
  bool have_lut = false;  // the entropy decode table is uploaded once
  hipEvent_t ev_done = nullptr;  // completion of a rebuild: polled (a few microseconds earlier than a blocking wait)
  // a rebuild that was only enqueued (eea_config_domain_async): control calls on OTHER streams wait for this event
  hipEvent_t ev_rebuild = nullptr;
  hipStream_t rebuild_stream = nullptr;
  bool rebuild_pending = false;
  DevBuf d_lut, d_raw, d_occ;  // occupancy targets: decode table, un-normalised sums, staged cells

  // workspaces of eea_ck_records_sum (group records of both levels + the tickets of its tree), one per distinct output
  // buffer: concurrent calls on several streams must not share tickets
  struct SumWs
  {
    const void* key = nullptr;
    unsigned long last_use = 0;
    hipStream_t last_stream = nullptr;  // the stream of the last launch that used the buffers
    DevBuf ws, ctr;
  };
  std::vector<std::unique_ptr<SumWs>> sum_ws;  // (pointers: a workspace in use must not move when the list grows)
  std::mutex sum_mutex;
  unsigned long sum_clock = 0;
  // device buffers replaced while launches may still use them: freed once the event recorded behind their last launch has
  // completed (or with the engine)
  struct Retired
  {
    void* p;
    hipEvent_t done;
  };
  std::vector<Retired> retired;

  // single-agent path
  hipStream_t stream1 = nullptr;
  Mailbox* h_mail = nullptr;
  void* d_mail = nullptr;
  DevBuf d_ut1, d_traj1, d_mem1;
  void* h_stage = nullptr;  // pinned staging for mem_cols / ut transfers
  size_t h_stage_cap = 0;
  double last_pose[3] = { 0, 0, 0 };
  int mail_seq = 0;          // sequence number of the last single-agent launch
  int* mail_done = nullptr;  // set while eea_control fills the launch parameters
  unsigned long phik_gen = 0;  // bumped whenever phi_k / the domain changes

  // resident single-robot workgroup (EEA_OPT_RESIDENT_CONTROL): host-mapped mailbox + replay-memory buffer, its own stream
  void* h_rmail = nullptr;
  void* d_rmail = nullptr;
  void* h_rmem = nullptr;
  void* d_rmem = nullptr;
  DevBuf d_rstage;             // ResidentStage: the request's pose / column count in device memory
  hipStream_t stream_res = nullptr;
  bool res_launched = false;   // a workgroup was launched and has not been seen to leave
  bool res_one_wavefront = false;  // ... the one-wavefront form (horizons of one slot): its controls live in LDS between requests
  unsigned res_seq = 0;        // last request number
  unsigned long res_gen = 0;   // phik_gen it was launched with
};

namespace
{
template <typename R>
void to_real(const double* in, R* out, size_t n)
{
  for (size_t i = 0; i < n; ++i) out[i] = static_cast<R>(in[i]);
}

eea_status use_device(const eea_engine* e)
{
  EEA_HIP(hipSetDevice(e->cfg.device));
  return EEA_OK;
}

eea_status stage_reserve(eea_engine* e, size_t bytes)
{
  if (bytes <= e->h_stage_cap) return EEA_OK;
  // grows geometrically: the replay memory adds a column per tick until it reaches the batch size
  size_t want = e->h_stage_cap ? 2 * e->h_stage_cap : 4096;
  if (want < bytes) want = bytes;
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  e->h_stage = nullptr;
  e->h_stage_cap = 0;
  EEA_HIP(hipHostMalloc(&e->h_stage, want, hipHostMallocDefault));
  e->h_stage_cap = want;
  return EEA_OK;
}

// waits for everything enqueued on `s` so far: records an event and polls it (bounded), then the ordinary wait
eea_status wait_stream_spin(eea_engine* e, hipStream_t s)
{
  if (e->ev_done == nullptr) EEA_HIP(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
  EEA_HIP(hipEventRecord(e->ev_done, s));
  for (int spin = 0; spin < 200000; ++spin) {
    const hipError_t q = hipEventQuery(e->ev_done);
    if (q == hipSuccess) return EEA_OK;
    if (q != hipErrorNotReady) return fail(EEA_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
  }
  EEA_HIP(hipStreamSynchronize(s));
  return EEA_OK;
}

template <typename R>
eea_status upload_lamdak(eea_engine* e)
{
  // basis.cpp:69-75: lamdak = 1 / (1 + sqrt(k1^2 + k2^2))^1.5, col = k2*K + k1
  std::vector<R> lam(e->K2);
  for (int k2 = 0; k2 < e->K; ++k2) {
    for (int k1 = 0; k1 < e->K; ++k1) {
      const double ss = static_cast<double>(k1 * k1 + k2 * k2);
      lam[k2 * e->K + k1] = static_cast<R>(1.0 / std::pow(1.0 + std::sqrt(ss), 1.5));
    }
  }
  EEA_HIP(e->d_lamdak.reserve(sizeof(R) * e->K2));
  EEA_HIP(e->d_phik.reserve(sizeof(R) * e->K2));
  EEA_HIP(hipMemcpy(e->d_lamdak.p, lam.data(), sizeof(R) * e->K2, hipMemcpyHostToDevice));
  EEA_HIP(hipMemset(e->d_phik.p, 0, sizeof(R) * e->K2));
  return EEA_OK;
}

// coordinates of configTarget's grid: repeated += resolution (ergodic_control.hpp:387-408).  The sequence does
// not depend on the domain, only on the resolution, so it is generated once per engine (and extended when a
// larger grid appears): no upload, no synchronisation on the rebuild path.
template <typename R>
eea_status ensure_axis(eea_engine* e, unsigned n)
{
  if (n <= e->axis_n) return EEA_OK;
  unsigned cap = e->axis_n ? 2 * e->axis_n : 2048;
  if (cap < n) cap = n;
  std::vector<R> v(cap);
  double x = 0.0;
  for (unsigned i = 0; i < cap; ++i) {
    v[i] = static_cast<R>(x);
    x += e->cfg.resolution;
  }
  EEA_HIP(hipDeviceSynchronize());  // kernels of earlier rebuilds may still read the old array
  EEA_HIP(e->d_axis.reserve(sizeof(R) * cap));
  EEA_HIP(hipMemcpy(e->d_axis.p, v.data(), sizeof(R) * cap, hipMemcpyHostToDevice));
  e->axis_n = cap;
  return EEA_OK;
}

// cos tables of the current domain (e->lx, e->ly) on an nx x ny grid: one launch, reused while grid and domain
// stay the same
// returns through *stale whether the tables have to be (re)computed; the caller that fuses their computation
// into its own launch passes stale != nullptr, everybody else gets the separate table launch
template <typename R>
eea_status upload_axes_and_tables(eea_engine* e, unsigned nx, unsigned ny, hipStream_t s, bool* stale = nullptr)
{
  if (stale) *stale = false;
  eea_status st = ensure_axis<R>(e, nx > ny ? nx : ny);
  if (st != EEA_OK) return st;
  const size_t need_cx = sizeof(R) * nx * e->K, need_cy = sizeof(R) * ny * e->K;
  const size_t need_work = sizeof(R) * eea::spatial_work_elems(nx, ny, e->K);
  if (need_cx > e->d_cx.cap || need_cy > e->d_cy.cap || need_work > e->d_work.cap) {
    EEA_HIP(hipDeviceSynchronize());  // the buffers about to be replaced may still be in use
    EEA_HIP(e->d_cx.reserve(need_cx));
    EEA_HIP(e->d_cy.reserve(need_cy));
    EEA_HIP(e->d_work.reserve(need_work));
    e->tab_nx = e->tab_ny = 0;
  }
  if (e->tab_nx == nx && e->tab_ny == ny && e->tab_lx == e->lx && e->tab_ly == e->ly) return EEA_OK;
  const R pi_lx = static_cast<R>(eea::kPi / e->lx), pi_ly = static_cast<R>(eea::kPi / e->ly);
  if (stale) {
    // the caller computes the tables inside its own launch and records the key once that launch is enqueued: a
    // failure in between must not leave the key on tables that were never written
    *stale = true;
    return EEA_OK;
  }
  EEA_HIP(eea::launch_axis_tables<R>(static_cast<const R*>(e->d_axis.p), nx, ny, e->K, pi_lx, pi_ly,
                                     static_cast<R*>(e->d_cx.p), static_cast<R*>(e->d_cy.p), s));
  e->tab_nx = nx;
  e->tab_ny = ny;
  e->tab_lx = e->lx;
  e->tab_ly = e->ly;
  return EEA_OK;
}

// A row tile is launched with ny = nrows and its tile geometry is recomputed from nrows; the rows-per-tile
// rule is not monotone in ny, so the partials buffer is sized for both the whole grid and the tile
template <typename R>
eea_status reserve_tile_work(eea_engine* e, unsigned nx, unsigned ny_total, unsigned nrows)
{
  size_t elems = eea::spatial_work_elems(nx, ny_total, e->K);
  const size_t tile = eea::spatial_work_elems(nx, nrows, e->K);
  if (tile > elems) elems = tile;
  EEA_HIP(e->d_work.reserve(sizeof(R) * elems));
  return EEA_OK;
}

// Target::fill + Basis::spatialCoeff on the device: three launches (tables + fill, streaming pass, final sums
// with the normalisation folded in) and ONE host synchronisation at the end
// orders stream s behind a rebuild that was only enqueued on another stream (no-op once it has completed)
eea_status order_after_rebuild(eea_engine* e, hipStream_t s)
{
  if (!e->rebuild_pending || s == e->rebuild_stream) return EEA_OK;
  if (hipEventQuery(e->ev_rebuild) == hipSuccess) {
    e->rebuild_pending = false;
    return EEA_OK;
  }
  EEA_HIP(hipStreamWaitEvent(s, e->ev_rebuild, 0));
  return EEA_OK;
}
// host-side readers of what a rebuild writes
eea_status finish_rebuild(eea_engine* e)
{
  if (!e->rebuild_pending) return EEA_OK;
  EEA_HIP(hipEventSynchronize(e->ev_rebuild));
  e->rebuild_pending = false;
  return EEA_OK;
}

template <typename R>
eea_status rebuild_phik(eea_engine* e, hipStream_t s, bool wait)
{
  {  // the engine's grid / table / phi_k buffers: behind a rebuild still in flight on another stream
    const eea_status st0 = order_after_rebuild(e, s);
    if (st0 != EEA_OK) return st0;
  }
  // Add 1 to include the boundary (ergodic_control.hpp:383-385)
  const unsigned nx = axis_length(0.0, e->lx, e->cfg.resolution) + 1;
  const unsigned ny = axis_length(0.0, e->ly, e->cfg.resolution) + 1;
  const size_t P = static_cast<size_t>(nx) * ny;
  if (P == 0 || P > (static_cast<size_t>(1) << 31)) {
    return fail(EEA_ERR_UNSUPPORTED, "target grid size out of range");
  }
  e->nx = nx;
  e->ny = ny;
  const int ng = static_cast<int>(e->mu.size() / 2);
  // Gaussian parameters as the reference prepares them: mean translated into the Fourier
  // frame (target.hpp:99), cov_inv = inv(diagmat(sigma^2)) (target.hpp:69; 2x2 inverse)
  std::vector<R> g(static_cast<size_t>(4) * (ng ? ng : 1));
  e->fill_gauss.assign(static_cast<size_t>(4) * ng, 0.0);
  for (int i = 0; i < ng; ++i) {
    const double a = e->sigma[2 * i] * e->sigma[2 * i], d = e->sigma[2 * i + 1] * e->sigma[2 * i + 1];
    const double det = a * d - 0.0 * 0.0;
    e->fill_gauss[4 * i + 0] = e->mu[2 * i] - e->map_x;
    e->fill_gauss[4 * i + 1] = e->mu[2 * i + 1] - e->map_y;
    e->fill_gauss[4 * i + 2] = d / det;
    e->fill_gauss[4 * i + 3] = a / det;
    for (int c = 0; c < 4; ++c) g[4 * i + c] = static_cast<R>(e->fill_gauss[4 * i + c]);
  }
  // A sum of axis-aligned Gaussians on the rectangular grid factors per axis, and so do phi_k and the mass: ONE launch
  // of one workgroup, (nx + ny)(G + K) transcendentals, no grid (gaussian_phik_kernel).  The grid itself is filled only
  // when eea_get_target_grid asks for it.  EEA_OPT_REBUILD_IMPL = 1 forces the streaming form below (A/B, tests).
  if (ng >= 1 && ng <= eea::kMaxGaussArgs && eea::option(EEA_OPT_REBUILD_IMPL) != 1 &&
      eea::gaussian_phik_lds_bytes(nx, ny, ng, e->K, sizeof(R)) <= 160 * 1024) {
    eea_status st = ensure_axis<R>(e, nx > ny ? nx : ny);
    if (st != EEA_OK) return st;
    if (e->d_sum.cap < sizeof(R)) {
      EEA_HIP(hipDeviceSynchronize());
      EEA_HIP(e->d_sum.reserve(sizeof(R) * 64));
    }
    eea::GaussArgs<R> ga;
    std::memset(&ga, 0, sizeof(ga));
    ga.n = ng;
    for (int i = 0; i < ng; ++i) {
      for (int c = 0; c < 4; ++c) ga.g[i][c] = g[4 * i + c];
    }
    if (!wait && e->ev_rebuild == nullptr) EEA_HIP(hipEventCreateWithFlags(&e->ev_rebuild, hipEventDisableTiming | hipEventDisableSystemFence));
    EEA_HIP(eea::launch_gaussian_phik<R>(static_cast<const R*>(e->d_axis.p), nx, ny, ga, e->K,
                                         static_cast<R>(1.0 / e->lx), static_cast<R>(1.0 / e->ly),
                                         static_cast<R*>(e->d_phik.p), static_cast<R*>(e->d_sum.p), s,
                                         wait ? nullptr : e->ev_rebuild));
    if (wait) {
      st = wait_stream_spin(e, s);
      if (st != EEA_OK) return st;
      e->rebuild_pending = false;
    } else {
      e->rebuild_stream = s;
      e->rebuild_pending = true;
    }
    e->have_phik = true;
  ++e->phik_gen;  // (the resident single-robot workgroup restarts on the next call)
    e->have_fill_grid = true;
    e->phi_is_raw = true;
    e->fill_deferred = true;
    return EEA_OK;
  }
  e->fill_deferred = false;
  bool tables_stale = false;
  eea_status st = upload_axes_and_tables<R>(e, nx, ny, s, ng <= eea::kMaxGaussArgs ? &tables_stale : nullptr);
  if (st != EEA_OK) return st;
  const size_t need_phi = sizeof(R) * P;
  const int fill_blocks_args = eea::target_fill_blocks(P);
  const int fill_blocks_buf = static_cast<int>((P + eea::kBlock - 1) / eea::kBlock);
  const size_t need_sum = sizeof(R) * (static_cast<size_t>(ng <= eea::kMaxGaussArgs ? fill_blocks_args : fill_blocks_buf) + 1);
  if (need_phi > e->d_phi.cap || need_sum > e->d_sum.cap) {
    EEA_HIP(hipDeviceSynchronize());
    EEA_HIP(e->d_phi.reserve(need_phi));
    EEA_HIP(e->d_sum.reserve(need_sum));
  }
  R* const d_mass = static_cast<R*>(e->d_sum.p);
  R* const d_partials = d_mass + 1;
  int n_partials = 0;
  if (ng <= eea::kMaxGaussArgs) {
    eea::GaussArgs<R> ga;
    std::memset(&ga, 0, sizeof(ga));
    ga.n = ng;
    for (int i = 0; i < ng; ++i) {
      for (int c = 0; c < 4; ++c) ga.g[i][c] = g[4 * i + c];
    }
    n_partials = fill_blocks_args;
    // the fill and (when the domain changed) the two axis tables in ONE launch
    EEA_HIP(eea::launch_target_fill_args<R>(static_cast<const R*>(e->d_axis.p), nx, ny, ga,
                                            static_cast<R*>(e->d_phi.p), d_partials, e->K,
                                            static_cast<R>(eea::kPi / e->lx), static_cast<R>(eea::kPi / e->ly),
                                            tables_stale ? static_cast<R*>(e->d_cx.p) : nullptr,
                                            static_cast<R*>(e->d_cy.p), s));
    if (tables_stale) {
      e->tab_nx = nx;
      e->tab_ny = ny;
      e->tab_lx = e->lx;
      e->tab_ly = e->ly;
    }
  } else {
    // more Gaussians than the kernel arguments hold: parameters through a device buffer (one more sync)
    EEA_HIP(e->d_gauss.reserve(sizeof(R) * g.size()));
    EEA_HIP(hipMemcpyAsync(e->d_gauss.p, g.data(), sizeof(R) * g.size(), hipMemcpyHostToDevice, s));
    EEA_HIP(hipStreamSynchronize(s));
    EEA_HIP(eea::launch_target_fill<R>(static_cast<const R*>(e->d_axis.p), static_cast<const R*>(e->d_axis.p),
                                       nx, ny, static_cast<const R*>(e->d_gauss.p), ng,
                                       static_cast<R*>(e->d_phi.p), d_partials, &n_partials, s));
  }
  // phi_k = spatialCoeff(phi / sum(phi)) = spatialCoeff(phi) / sum(phi)  (target.cpp:87, basis.cpp:122-133)
  // enqueue-only form: the event other streams (and the getters) wait for is bound to the last launch itself
  if (!wait && e->ev_rebuild == nullptr) EEA_HIP(hipEventCreateWithFlags(&e->ev_rebuild, hipEventDisableTiming | hipEventDisableSystemFence));
  EEA_HIP(eea::launch_spatial_coeff_normalised<R>(static_cast<const R*>(e->d_phi.p), nx, ny, e->K,
                                                  static_cast<const R*>(e->d_cx.p), static_cast<const R*>(e->d_cy.p),
                                                  static_cast<R*>(e->d_work.p), static_cast<R*>(e->d_phik.p),
                                                  d_partials, n_partials, d_mass, s, wait ? nullptr : e->ev_rebuild));
  if (wait) {
    st = wait_stream_spin(e, s);
    if (st != EEA_OK) return st;
    e->rebuild_pending = false;
  } else {
    // enqueue only: whatever follows on s is ordered by the stream; other streams wait for the event
    e->rebuild_stream = s;
    e->rebuild_pending = true;
  }
  e->have_phik = true;
  ++e->phik_gen;  // (the resident single-robot workgroup restarts on the next call)
  e->have_fill_grid = true;
  e->phi_is_raw = true;
  return EEA_OK;
}

// Target::fill of the last Gaussian rebuild on demand (eea_get_target_grid): the un-normalised grid into d_phi; d_sum[0]
// keeps the mass the rebuild computed from the per-axis factors
template <typename R>
eea_status fill_deferred_grid(eea_engine* e)
{
  const unsigned nx = e->nx, ny = e->ny;
  const size_t P = static_cast<size_t>(nx) * ny;
  const int ng = static_cast<int>(e->fill_gauss.size() / 4);
  const int blocks = eea::target_fill_blocks(P);
  EEA_HIP(hipDeviceSynchronize());
  EEA_HIP(e->d_phi.reserve(sizeof(R) * P));
  DevBuf partials;  // (the fill's per-workgroup sums are not needed: the mass is already known)
  EEA_HIP(partials.reserve(sizeof(R) * (static_cast<size_t>(blocks) + 1)));
  eea::GaussArgs<R> ga;
  std::memset(&ga, 0, sizeof(ga));
  ga.n = ng;
  for (int i = 0; i < ng; ++i) {
    for (int c = 0; c < 4; ++c) ga.g[i][c] = static_cast<R>(e->fill_gauss[4 * i + c]);
  }
  const hipError_t err = eea::launch_target_fill_args<R>(static_cast<const R*>(e->d_axis.p), nx, ny, ga, static_cast<R*>(e->d_phi.p),
                                                         static_cast<R*>(partials.p), e->K, R(0), R(0), nullptr, nullptr, nullptr);
  const hipError_t err2 = hipDeviceSynchronize();
  partials.release();
  EEA_HIP(err);
  EEA_HIP(err2);
  e->fill_deferred = false;
  return EEA_OK;
}

template <typename R>
eea_status set_target_grid_impl(eea_engine* e, unsigned nx, unsigned ny, const void* phi_vals,
                                int on_device, hipStream_t s)
{
  const size_t P = static_cast<size_t>(nx) * ny;
  e->nx = nx;
  e->ny = ny;
  e->have_fill_grid = false;
  eea_status st = upload_axes_and_tables<R>(e, nx, ny, s);
  if (st != EEA_OK) return st;
  const R* d_phi = static_cast<const R*>(phi_vals);
  if (!on_device) {
    EEA_HIP(e->d_phi.reserve(sizeof(R) * P));
    EEA_HIP(hipMemcpyAsync(e->d_phi.p, phi_vals, sizeof(R) * P, hipMemcpyHostToDevice, s));
    EEA_HIP(hipStreamSynchronize(s));
    d_phi = static_cast<const R*>(e->d_phi.p);
  }
  EEA_HIP(eea::launch_spatial_coeff<R>(d_phi, nx, ny, e->K, static_cast<const R*>(e->d_cx.p),
                                       static_cast<const R*>(e->d_cy.p), static_cast<R*>(e->d_work.p),
                                       static_cast<R*>(e->d_phik.p), s));
  EEA_HIP(hipStreamSynchronize(s));
  e->have_phik = true;
  ++e->phik_gen;  // (the resident single-robot workgroup restarts on the next call)
  return EEA_OK;
}

template <typename R>
void fill_params(const eea_engine* e, eea::ControlParams<R>& p)
{
  std::memset(&p, 0, sizeof(p));
  p.T = e->T;
  p.K = e->K;
  p.chunk = e->chunk;
  p.dt = static_cast<R>(e->cfg.dt);
  p.dt6 = static_cast<R>(e->cfg.dt / 6.0);
  p.half_dt = static_cast<R>(0.5 * e->cfg.dt);
  p.lx = static_cast<R>(e->lx);
  p.ly = static_cast<R>(e->ly);
  p.map_x = static_cast<R>(e->map_x);
  p.map_y = static_cast<R>(e->map_y);
  p.expl_weight = static_cast<R>(e->cfg.expl_weight);
  p.pi_lx = static_cast<R>(eea::kPi / e->lx);
  p.pi_ly = static_cast<R>(eea::kPi / e->ly);
  p.inv_lx = static_cast<R>(1.0 / e->lx);
  p.inv_ly = static_cast<R>(1.0 / e->ly);
  for (int i = 0; i < 9; ++i) p.Rinv[i] = static_cast<R>(e->cfg.Rinv[i]);
  for (int i = 0; i < 3; ++i) {
    p.umin[i] = static_cast<R>(e->cfg.umin[i]);
    p.umax[i] = static_cast<R>(e->cfg.umax[i]);
  }
  p.phik = static_cast<const R*>(e->d_phik.p);
  p.lamdak = static_cast<const R*>(e->d_lamdak.p);
  p.n_steps = 1;
}

// workspace of the sum written to `key` for up to B records.  One per distinct output buffer (concurrent sums on
// several streams must not share tickets), found under the engine's lock (the header allows concurrent calls with
// distinct d_sum buffers); at most kMaxSumWs of them -- a caller that passes a fresh output buffer every call recycles
// the least recently used workspace instead of growing the list.  The tickets are zeroed on the launch stream.
constexpr size_t kMaxSumWs = 32;
// (the caller holds e->sum_mutex, and keeps it across its launch: a second thread that misses with every workspace in use
// must not recycle this one between the lookup and the launch)
template <typename R>
eea_status sum_workspace(eea_engine* e, const void* key, unsigned B, hipStream_t s, eea_engine::SumWs** out)
{
  eea_engine::SumWs* w = nullptr;
  for (auto& cand : e->sum_ws) {
    if (cand->key == key) w = cand.get();
  }
  bool recycled = false;
  if (w == nullptr) {
    if (e->sum_ws.size() < kMaxSumWs) {
      e->sum_ws.emplace_back(new eea_engine::SumWs());
      w = e->sum_ws.back().get();
    } else {  // least recently used
      w = e->sum_ws.front().get();
      for (auto& cand : e->sum_ws) {
        if (cand->last_use < w->last_use) w = cand.get();
      }
      recycled = true;
    }
    w->key = key;
  }
  w->last_use = ++e->sum_clock;
  const size_t need_ws = sizeof(R) * eea::ck_sum_ws_elems(B, e->K2);
  const size_t need_ctr = sizeof(unsigned) * eea::ck_sum_tickets(B, e->K2);
  if (need_ws > w->ws.cap || need_ctr > w->ctr.cap || recycled) {
    // NO device synchronisation here (a per-pass call must not stall every stream of the device): buffers an
    // earlier launch may still use are retired behind an event on that launch's stream and freed once it has completed
    // (no device-wide sweep); a recycled workspace gets fresh buffers for the same reason.
    for (size_t i = 0; i < e->retired.size();) {
      if (e->retired[i].done != nullptr && hipEventQuery(e->retired[i].done) == hipSuccess) {
        (void)hipEventDestroy(e->retired[i].done);
        (void)hipFree(e->retired[i].p);
        e->retired[i] = e->retired.back();
        e->retired.pop_back();
      } else {
        ++i;
      }
    }
    (void)hipGetLastError();  // (hipErrorNotReady of the queries is not an error of this call)
    // Each buffer moves to the retired list individually, and the workspace forgets it the moment it has (a failed
    // record must neither leak the event nor leave w->ws.p pointing at a buffer the list will free).  last_stream is a
    // CALLER-owned handle remembered from an earlier call: if the caller has destroyed it the record fails -- the event is
    // destroyed and the buffer stays in the list without one, until eea_destroy (its last launch cannot be proven complete).
    for (DevBuf* buf : { &w->ws, &w->ctr }) {
      void* const q = buf->p;
      if (q == nullptr) continue;
      hipEvent_t ev = nullptr;
      if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
      if (ev != nullptr && hipEventRecord(ev, w->last_stream) != hipSuccess) {
        (void)hipEventDestroy(ev);
        ev = nullptr;
      }
      (void)hipGetLastError();
      e->retired.push_back({ q, ev });
      buf->p = nullptr;  // (ownership moved to the retired list)
      *buf = DevBuf();
    }
    EEA_HIP(w->ws.reserve(need_ws));
    EEA_HIP(w->ctr.reserve(need_ctr));
    EEA_HIP(hipMemsetAsync(w->ctr.p, 0, w->ctr.cap, s));  // the tickets reset themselves from here on
  }
  w->last_stream = s;
  *out = w;
  return EEA_OK;
}

template <typename R>
eea_status control_batch_impl(eea_engine* e, unsigned B, const eea_batch_io* io, bool rollout_only,
                              hipStream_t s, long long* d_stamps = nullptr, unsigned n_steps = 1,
                              unsigned pose_step_stride = 0, unsigned u0_step_stride = 0)
{
  eea::ControlParams<R> p;
  fill_params<R>(e, p);
  p.n_steps = static_cast<int>(n_steps);
  p.pose_step_stride = pose_step_stride;
  p.u0_step_stride = u0_step_stride;
  p.pose = static_cast<const R*>(io->d_pose);
  p.ut = static_cast<R*>(io->d_ut);
  p.mem_cols = static_cast<const R*>(io->d_mem_cols);
  p.n_mem = io->d_n_mem;
  p.mem_stride = io->d_mem_cols ? io->mem_stride : 0u;
  p.u0 = static_cast<R*>(io->d_u0);
  p.traj = static_cast<R*>(io->d_traj);
  p.ck = static_cast<R*>(io->d_ck);
  p.ck_shared = static_cast<const R*>(io->d_ck_shared);
  p.ck_shared_parts = io->d_ck_shared != nullptr ? static_cast<int>(io->ck_shared_parts) : 0;
  p.rec_len = eea::ck_record_len(e->K2);
  p.rec_ready = rollout_only ? nullptr : io->d_rec_ready;
  p.rec_wave = io->rec_per_wavefront != 0 ? 1 : 0;
  p.rec_seq = io->rec_seq;
  p.ck_flag = io->d_ck_shared != nullptr ? io->d_ck_flag : nullptr;
  p.ck_flag_seq = io->ck_flag_seq;
  p.edx = static_cast<R*>(io->d_edx);
  p.bdx = static_cast<R*>(io->d_bdx);
  p.rhot = static_cast<R*>(io->d_rhot);
  p.status = io->d_status;
  p.skip = io->d_skip;
  p.done = e->mail_done;
  p.done_seq = e->mail_seq;
  const int n_mem_max = rollout_only ? 0 : static_cast<int>(p.mem_stride);
  p.dbg = d_stamps;  // phase stamps: null in the product (only the A/B library's kernels read it, tools/ab/)
  // two kernels ship: one wavefront per agent (horizons <= 256 steps, K <= 16 or K = 20) and one workgroup per agent
  // (everything else).  Batches take the throughput kernel unless EEA_OPT_CONTROL_KERNEL says otherwise.  The single-agent
  // entry (eea_control: one agent, latency) takes it for horizons of one slot (T <= 64: one step per lane, no barrier --
  // 3.0 / 4.9 / 4.3 us of device time at configs[0] / configs[1] / the yaml's T = 50 against 5.1 / 7.4 / 5.2 of the
  // workgroup) and keeps four wavefronts per agent beyond (T = 200: 7.2 against 11.6 us; profiles/r05_one_agent_kernels.txt)
  const bool use_wave = eea::option(EEA_OPT_CONTROL_KERNEL) == 0 && (e->mail_done == nullptr || p.T <= 64) &&
                        eea::control_wave_eligible<R>(p, rollout_only);
  p.ck_rec = rollout_only ? nullptr : static_cast<R*>(io->d_ck_rec);
  if (use_wave) {
    // short horizons: several agents share a wavefront (control_pack_impl.hpp) when the batch still fills the chip
    if constexpr (sizeof(R) == 8) {
      const int lanes = eea::control_pack_lanes(p, B, eea::option(EEA_OPT_AGENT_LANES));
      if (lanes != 0) {
        EEA_HIP(eea::launch_control_pack(p, B, e->cfg.model, rollout_only, lanes, s));
        return EEA_OK;
      }
    }
    EEA_HIP(eea::launch_control_wave<R>(p, B, e->cfg.model, rollout_only, s));
    return EEA_OK;
  }
  const size_t lds = eea::control_lds_bytes<R>(p.T, p.K, n_mem_max, p.chunk);
  if (lds > 160 * 1024) {
    return fail(EEA_ERR_UNSUPPORTED, "horizon/memory/basis too large for one workgroup's 160 KiB LDS");
  }
  // the workgroup-per-agent kernel takes one step per launch: a multi-step call is that many launches on the stream
  p.n_steps = 1;
  for (unsigned n = 0; n < n_steps; ++n) {
    p.pose = static_cast<const R*>(io->d_pose) + 3 * static_cast<size_t>(n) * pose_step_stride;
    p.u0 = static_cast<R*>(io->d_u0) + 3 * static_cast<size_t>(n) * u0_step_stride;
    EEA_HIP(eea::launch_control<R>(p, B, e->cfg.model, n_mem_max, rollout_only, s));
  }
  return EEA_OK;
}

// numerics.hpp:164-179 of the reference: information entropy of one occupancy cell, p = cell / 100
// (unknown cells, p < 0, count 0.7; p == 0 or 1 count 1e-3)
double cell_entropy(double p)
{
  if (std::fabs(0.0 - p) < 1.0e-12 || std::fabs(1.0 - p) < 1.0e-12) return 1e-3;
  if (p < 0.0) return 0.7;
  return -p * std::log(p) - (1.0 - p) * std::log(1.0 - p);
}

// decode table of the occupancy path, indexed by the cell's byte (int8 two's complement)
template <typename R>
eea_status upload_entropy_table(eea_engine* e, hipStream_t s)
{
  std::vector<R> lut(256);
  for (int b = 0; b < 256; ++b) {
    const int cell = static_cast<int>(static_cast<int8_t>(static_cast<uint8_t>(b)));
    lut[b] = static_cast<R>(cell_entropy(static_cast<double>(cell) / 100.0));  // getCell: grid.cpp:176-184
  }
  EEA_HIP(e->d_raw.reserve(sizeof(R) * e->K2));
  if (e->have_lut) return EEA_OK;
  EEA_HIP(e->d_lut.reserve(sizeof(R) * 256));
  EEA_HIP(hipMemcpyAsync(e->d_lut.p, lut.data(), sizeof(R) * 256, hipMemcpyHostToDevice, s));
  EEA_HIP(hipStreamSynchronize(s));
  e->have_lut = true;
  return EEA_OK;
}

// un-normalised coefficient sums of rows [row0, row0 + nrows) of an occupancy grid
template <typename R>
eea_status occupancy_rows_impl(eea_engine* e, unsigned nx, unsigned ny_total, unsigned row0, unsigned nrows,
                               const int8_t* d_occ_rows, void* d_raw_out, hipStream_t s, void* d_mass_out = nullptr)
{
  eea_status st = upload_axes_and_tables<R>(e, nx, ny_total, s);
  if (st != EEA_OK) return st;
  st = upload_entropy_table<R>(e, s);
  if (st != EEA_OK) return st;
  st = reserve_tile_work<R>(e, nx, ny_total, nrows);
  if (st != EEA_OK) return st;
  EEA_HIP(eea::launch_spatial_coeff_cells<R>(d_occ_rows, nx, nrows, e->K, static_cast<const R*>(e->d_cx.p),
                                             static_cast<const R*>(e->d_cy.p) + static_cast<size_t>(row0) * e->K,
                                             static_cast<const R*>(e->d_lut.p), static_cast<R*>(e->d_work.p),
                                             static_cast<R*>(d_raw_out), s, static_cast<R*>(d_mass_out)));
  return EEA_OK;
}

// DynamicWindow::DynamicWindow (dynamic_window.cpp:71-90) from the C configuration
eea_status make_dwa_params(const eea_dwa_cfg* dcfg, eea::DwaParams& d)
{
  if (dcfg == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null DWA configuration");
  d.dt = dcfg->dt;
  d.acc_dt = dcfg->acc_dt;
  d.acc_lim[0] = dcfg->acc_lim_x;
  d.acc_lim[1] = dcfg->acc_lim_y;
  d.acc_lim[2] = dcfg->acc_lim_th;
  d.vmax[0] = dcfg->max_vel_x;
  d.vmax[1] = dcfg->max_vel_y;
  d.vmax[2] = dcfg->max_rot_vel;
  d.vmin[0] = dcfg->min_vel_x;
  d.vmin[1] = dcfg->min_vel_y;
  d.vmin[2] = dcfg->min_rot_vel;
  // a sample count of 0 is raised to 1 (DynamicWindow::DynamicWindow, dynamic_window.cpp:71-90)
  d.ns[0] = dcfg->vx_samples ? dcfg->vx_samples : 1;
  d.ns[1] = dcfg->vy_samples ? dcfg->vy_samples : 1;
  d.ns[2] = dcfg->vth_samples ? dcfg->vth_samples : 1;
  d.steps = static_cast<unsigned>(std::abs(dcfg->horizon / dcfg->dt));
  if (static_cast<size_t>(d.ns[0]) * d.ns[1] * d.ns[2] > 8192) return fail(EEA_ERR_UNSUPPORTED, "more than 8192 DWA samples");
  return EEA_OK;
}

// ---- resident single-robot workgroup (control_kernel_impl.hpp control_resident_kernel) --------------------------------------
// idle time of a resident workgroup in ticks of the constant 100 MHz clock (EEA_OPT_RESIDENT_IDLE_MS, default 250 ms: a 10 Hz
// loop keeps it alive)
long long resident_idle_ticks() { return 100000LL * static_cast<long long>(eea::option(EEA_OPT_RESIDENT_IDLE_MS)); }

// tells the workgroup to leave and waits until it has (no-op when none was launched)
eea_status resident_stop(eea_engine* e)
{
  if (!e->res_launched) return EEA_OK;
  // (the two mailbox layouts differ only in the width of the reals in front of cmd / req: go through the right one)
  if (e->f32) {
    auto* m = static_cast<eea::ResidentMail<float>*>(e->h_rmail);
    __atomic_store_n(&m->cmd, 1, __ATOMIC_RELAXED);
    __atomic_store_n(&m->req, ++e->res_seq, __ATOMIC_RELEASE);
  } else {
    auto* m = static_cast<eea::ResidentMail<double>*>(e->h_rmail);
    __atomic_store_n(&m->cmd, 1, __ATOMIC_RELAXED);
    __atomic_store_n(&m->req, ++e->res_seq, __ATOMIC_RELEASE);
  }
  EEA_HIP(hipStreamSynchronize(e->stream_res));  // it leaves within microseconds (or has left already: idle)
  if (e->f32) static_cast<eea::ResidentMail<float>*>(e->h_rmail)->cmd = 0;
  else static_cast<eea::ResidentMail<double>*>(e->h_rmail)->cmd = 0;
  e->res_launched = false;
  return EEA_OK;
}

template <typename R>
eea_status resident_start(eea_engine* e)
{
  using Mail = eea::ResidentMail<R>;
  if (e->h_rmail == nullptr) {
    EEA_HIP(hipHostMalloc(&e->h_rmail, sizeof(Mail), hipHostMallocMapped));
    std::memset(e->h_rmail, 0, sizeof(Mail));
    EEA_HIP(hipHostGetDevicePointer(&e->d_rmail, e->h_rmail, 0));
    EEA_HIP(hipHostMalloc(&e->h_rmem, sizeof(R) * 3 * eea::kResidentMemCols, hipHostMallocMapped));
    std::memset(e->h_rmem, 0, sizeof(R) * 3 * eea::kResidentMemCols);
    EEA_HIP(hipHostGetDevicePointer(&e->d_rmem, e->h_rmem, 0));
    EEA_HIP(hipStreamCreateWithFlags(&e->stream_res, hipStreamNonBlocking));
    EEA_HIP(e->d_rstage.reserve(sizeof(eea::ResidentStage<R>)));
  }
  EEA_HIP(hipStreamSynchronize(e->stream_res));  // (an earlier workgroup that left by itself has completed)
  eea_status st = finish_rebuild(e);             // phi_k of an enqueued rebuild is complete before the workgroup reads it
  if (st != EEA_OK) return st;
  EEA_HIP(hipStreamSynchronize(e->stream1));     // ... and so is everything else the launch path left behind (d_ut1)
  Mail* const hm = static_cast<Mail*>(e->h_rmail);
  Mail* const dm = static_cast<Mail*>(e->d_rmail);
  eea::ControlParams<R> p;
  fill_params<R>(e, p);
  p.pose = dm->pose;   // (the workgroup points these two at its device-memory stage per request)
  p.n_mem = &dm->n_mem;
  p.ut = static_cast<R*>(e->d_ut1.p);
  p.u0 = dm->u0;
  p.status = &dm->status;
  p.mem_cols = static_cast<const R*>(e->d_rmem);
  p.mem_stride = static_cast<unsigned>(eea::kResidentMemCols);
  p.rec_len = eea::ck_record_len(e->K2);
  // horizons of one slot: ONE wavefront (the kernel the launch path takes at these shapes, control_batch_impl), otherwise the
  // workgroup
  bool one_wavefront = false;
  if constexpr (sizeof(R) == 8) one_wavefront = eea::option(EEA_OPT_CONTROL_KERNEL) == 0 && eea::control_wave_resident_eligible(p);
  const size_t lds = eea::control_lds_bytes<R>(p.T, p.K, eea::kResidentMemCols, p.chunk);
  if (!one_wavefront && lds > 160 * 1024) return EEA_ERR_UNSUPPORTED;  // (no message: the caller takes the launch path)
  hm->alive = 1;
  hm->cmd = 0;
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  if constexpr (sizeof(R) == 8) {
    if (one_wavefront) {
      p.done = &dm->done;
      p.res_mail = e->d_rmail;
      p.res_first = e->res_seq;
      p.res_idle = resident_idle_ticks();
      EEA_HIP(eea::launch_control_wave_resident(p, e->cfg.model, e->stream_res));
    }
  }
  if (!one_wavefront) {
    EEA_HIP(eea::launch_control_resident<R>(p, e->cfg.model, eea::kResidentMemCols, e->d_rmail, e->d_rstage.p, e->res_seq,
                                            resident_idle_ticks(), e->stream_res));
  }
  e->res_one_wavefront = one_wavefront;
  e->res_launched = true;
  e->res_gen = e->phik_gen;
  return EEA_OK;
}

// eea_control through the resident workgroup: post the request in the mailbox, poll the answer
template <typename R>
eea_status control_resident(eea_engine* e, const double x[3], const double* h_mem_cols, unsigned n_mem, double u_out[3])
{
  using Mail = eea::ResidentMail<R>;
  eea_status st = EEA_OK;
  if (e->res_launched && e->res_gen != e->phik_gen) {  // the domain / phi_k changed: the launch parameters are stale
    st = resident_stop(e);
    if (st != EEA_OK) return st;
  }
  Mail* hm = static_cast<Mail*>(e->h_rmail);
  if (!e->res_launched || __atomic_load_n(&hm->alive, __ATOMIC_ACQUIRE) == 0) {
    e->res_launched = false;
    st = resident_start<R>(e);
    if (st != EEA_OK) return st;
    hm = static_cast<Mail*>(e->h_rmail);
  }
  to_real<R>(x, hm->pose, 3);
  hm->map_x = static_cast<R>(e->map_x);
  hm->map_y = static_cast<R>(e->map_y);
  hm->n_mem = static_cast<int>(n_mem);
  if (n_mem > 0) to_real<R>(h_mem_cols, static_cast<R*>(e->h_rmem), 3 * static_cast<size_t>(n_mem));
  hm->status = 0;
  const unsigned seq = ++e->res_seq;
  __atomic_store_n(&hm->req, seq, __ATOMIC_RELEASE);
  // the answer (bounded: ~2 s of polling), or the news that the workgroup left before it saw the request
  bool relaunched = false;
  for (long spin = 0;; ++spin) {
    if (__atomic_load_n(&hm->done, __ATOMIC_ACQUIRE) == static_cast<int>(seq)) break;
    if (__atomic_load_n(&hm->alive, __ATOMIC_ACQUIRE) == 0 &&
        __atomic_load_n(&hm->done, __ATOMIC_ACQUIRE) != static_cast<int>(seq)) {
      if (relaunched) return fail(EEA_ERR_HIP, "the resident control workgroup left twice without answering");
      // It announced that it is leaving around the time of this request.  It may still be SERVING it (a request that lands
      // between "alive = 0" and its last look at the mailbox is served, and then it leaves -- it never polls again): wait
      // until it has left, then look at the answer once more.  Only a request that is still unanswered then gets a new
      // workgroup, which starts from the request before this one and sees it at once.  (Relaunching without this wait served
      // such a request twice: the warm start shifted twice, u0 rewritten under the host's read -- ADVICE r05.)
      EEA_HIP(hipStreamSynchronize(e->stream_res));
      if (__atomic_load_n(&hm->done, __ATOMIC_ACQUIRE) == static_cast<int>(seq)) {
        e->res_launched = false;  // answered by the leaving workgroup; the next call starts another one
        break;
      }
      e->res_launched = false;
      --e->res_seq;
      st = resident_start<R>(e);
      ++e->res_seq;
      if (st != EEA_OK) return st;
      relaunched = true;
    }
    if (spin > 400000000L) return fail(EEA_ERR_TIMEOUT, "the resident control workgroup does not answer");
  }
  if (hm->status == EEA_ERR_INVALID_TWIST) return fail(EEA_ERR_INVALID_TWIST, "Invalid twist y-velocity must be 0.");
  for (int i = 0; i < 3; ++i) u_out[i] = static_cast<double>(hm->u0[i]);
  return EEA_OK;
}

eea_status check_engine(const eea_engine* e)
{
  if (e == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null engine");
  return EEA_OK;
}
}  // namespace

extern "C" {

const char* eea_last_error(void) { return eea::g_last_error.c_str(); }

eea_status eea_set_option(int option, int value)
{
  bool ok = false;
  switch (option) {
    case EEA_OPT_CONTROL_KERNEL: ok = value == 0 || value == 1; break;
    case EEA_OPT_WORKGROUP_THREADS: ok = value == 0 || value == 64 || value == 128 || value == 256; break;
    case EEA_OPT_COLLISION_IMPL: ok = value >= 0 && value <= 2; break;
    case EEA_OPT_MAILBOX_POLL: ok = value == 0 || value == 1; break;
    case EEA_OPT_REBUILD_IMPL: ok = value == 0 || value == 1; break;
    case EEA_OPT_AGENT_LANES: ok = value == 0 || value == 8 || value == 16 || value == 32 || value == 64; break;
    case EEA_OPT_RESIDENT_CONTROL: ok = value == 0 || value == 1; break;
    case EEA_OPT_RESIDENT_IDLE_MS: ok = value >= 1 && value <= 60000; break;
    default: return fail(EEA_ERR_INVALID_ARGUMENT, "unknown option");
  }
  if (!ok) return fail(EEA_ERR_INVALID_ARGUMENT, "option value out of range");
  g_options[option].store(value, std::memory_order_relaxed);
  return EEA_OK;
}
int eea_get_option(int option) { return eea::option(option); }
unsigned eea_abi_version(void) { return EEA_ABI_VERSION; }

eea_status eea_create(const eea_config* cfg, eea_engine** out)
{
  if (cfg == nullptr || out == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (cfg->model != EEA_MODEL_OMNI && cfg->model != EEA_MODEL_SIMPLE_CART) {
    return fail(EEA_ERR_INVALID_ARGUMENT,
                "model must be Omni or SimpleCart (Cart/Mecanum cannot run under ErgodicControl)");
  }
  if (cfg->precision != EEA_PREC_F64 && cfg->precision != EEA_PREC_F32) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "unknown precision");
  }
  // steps_ = static_cast<unsigned>(std::abs(horizon / dt))  (ergodic_control.hpp:199)
  const double ratio = std::abs(cfg->horizon / cfg->dt);
  if (!(ratio < 1.0e6)) return fail(EEA_ERR_INVALID_ARGUMENT, "horizon / dt out of range");
  const unsigned steps = static_cast<unsigned>(ratio);
  if (steps == 1) {
    return fail(EEA_ERR_INVALID_ARGUMENT,
                "Need at least two steps in forward simulation. Increase the horizon or decrease "
                "the time step.");
  }
  if (steps == 0) return fail(EEA_ERR_INVALID_ARGUMENT, "horizon shorter than one time step");
  if (cfg->num_basis == 0 || cfg->num_basis > static_cast<unsigned>(eea::kMaxBasis)) {
    return fail(EEA_ERR_UNSUPPORTED, "num_basis must be in [1, 32]");
  }
  int ndev = 0;
  EEA_HIP(hipGetDeviceCount(&ndev));
  if (cfg->device < 0 || cfg->device >= ndev) return fail(EEA_ERR_HIP, "no such HIP device");
  EEA_HIP(hipSetDevice(cfg->device));

  eea_engine* e = new eea_engine();
  e->cfg = *cfg;
  e->T = static_cast<int>(steps);
  e->K = static_cast<int>(cfg->num_basis);
  e->K2 = e->K * e->K;
  e->f32 = cfg->precision == EEA_PREC_F32;
  e->rs = e->f32 ? 4 : 8;
  eea_status st = e->f32 ? upload_lamdak<float>(e) : upload_lamdak<double>(e);
  if (st != EEA_OK) {
    eea_destroy(e);
    return st;
  }
  hipError_t err = hipStreamCreateWithFlags(&e->stream1, hipStreamNonBlocking);
  if (err == hipSuccess) {
    err = hipHostMalloc(reinterpret_cast<void**>(&e->h_mail), sizeof(Mailbox),
                        hipHostMallocMapped | hipHostMallocCoherent);
  }
  if (err == hipSuccess) {
    std::memset(e->h_mail, 0, sizeof(Mailbox));
    err = hipHostGetDevicePointer(&e->d_mail, e->h_mail, 0);
  }
  if (err == hipSuccess) err = e->d_ut1.reserve(e->rs * 3 * e->T);
  if (err == hipSuccess) err = hipMemset(e->d_ut1.p, 0, e->rs * 3 * e->T);  // ut_ starts at zero (:201)
  if (err == hipSuccess) err = e->d_traj1.reserve(e->rs * 3 * e->T);
  if (err != hipSuccess) {
    const std::string msg = std::string("engine allocation: ") + hipGetErrorString(err);
    eea_destroy(e);
    return fail(EEA_ERR_HIP, msg);
  }
  *out = e;
  return EEA_OK;
}

void eea_destroy(eea_engine* e)
{
  if (e == nullptr) return;
  (void)hipSetDevice(e->cfg.device);
  (void)resident_stop(e);  // the resident single-robot workgroup leaves before anything it reads is freed
  (void)finish_rebuild(e);
  if (e->stream1) {
    (void)hipStreamSynchronize(e->stream1);
    (void)hipStreamDestroy(e->stream1);
  }
  DevBuf* bufs[] = { &e->d_phik, &e->d_lamdak, &e->d_phi, &e->d_axis, &e->d_cx, &e->d_cy,
                     &e->d_work, &e->d_gauss, &e->d_sum, &e->d_ut1, &e->d_traj1, &e->d_mem1, &e->d_rstage,
                     &e->d_lut, &e->d_raw, &e->d_occ };
  for (DevBuf* b : bufs) b->release();
  for (auto& w : e->sum_ws) {
    w->ws.release();
    w->ctr.release();
  }
  for (auto& r : e->retired) {
    if (r.done != nullptr) (void)hipEventDestroy(r.done);
    (void)hipFree(r.p);
  }
  if (e->ev_done) (void)hipEventDestroy(e->ev_done);
  if (e->ev_rebuild) (void)hipEventDestroy(e->ev_rebuild);
  if (e->h_mail) (void)hipHostFree(e->h_mail);
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  if (e->h_rmail) (void)hipHostFree(e->h_rmail);
  if (e->h_rmem) (void)hipHostFree(e->h_rmem);
  if (e->stream_res) (void)hipStreamDestroy(e->stream_res);
  delete e;
}

unsigned eea_steps(const eea_engine* e) { return e ? static_cast<unsigned>(e->T) : 0u; }
unsigned eea_batch_agent_lanes(const eea_engine* e, unsigned B)
{
  if (e == nullptr || eea::option(EEA_OPT_CONTROL_KERNEL) != 0) return 0u;
  if (e->cfg.precision == EEA_PREC_F64) {
    eea::ControlParams<double> p;
    fill_params<double>(e, p);
    if (!eea::control_wave_eligible<double>(p, false)) return 0u;
    const int lanes = eea::control_pack_lanes(p, B, eea::option(EEA_OPT_AGENT_LANES));
    return lanes != 0 ? static_cast<unsigned>(lanes) : 64u;
  }
  eea::ControlParams<float> p;
  fill_params<float>(e, p);
  return eea::control_wave_eligible<float>(p, false) ? 64u : 0u;
}
unsigned eea_batch_record_count(const eea_engine* e, unsigned B)
{
  const unsigned lanes = eea_batch_agent_lanes(e, B);
  if (e == nullptr) return 0u;
  if (lanes == 0u || lanes >= 64u) return B;
  const unsigned A = 64u / lanes;
  return (B + A - 1u) / A;
}
unsigned eea_num_modes(const eea_engine* e) { return e ? static_cast<unsigned>(e->K2) : 0u; }
size_t eea_real_size(const eea_engine* e) { return e ? e->rs : 0; }
unsigned eea_ck_record_len(const eea_engine* e) { return e ? static_cast<unsigned>(eea::ck_record_len(e->K2)) : 0u; }
double eea_time_step(const eea_engine* e) { return e ? e->cfg.dt : 0.0; }

eea_status eea_set_target_gaussians(eea_engine* e, unsigned n, const double* mu, const double* sigma)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (n > 0 && (mu == nullptr || sigma == nullptr)) return fail(EEA_ERR_INVALID_ARGUMENT, "null target");
  e->mu.assign(mu, mu + 2 * static_cast<size_t>(n));
  e->sigma.assign(sigma, sigma + 2 * static_cast<size_t>(n));
  e->have_gauss = true;
  return EEA_OK;
}

eea_status eea_set_target_grid(eea_engine* e, unsigned nx, unsigned ny, const void* phi_vals,
                               int on_device, double lx, double ly, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (finish_rebuild(e) != EEA_OK) return EEA_ERR_HIP;  // a rebuild that was only enqueued owns the buffers below
  if (phi_vals == nullptr || nx == 0 || ny == 0 || !(lx > 0.0) || !(ly > 0.0)) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "bad target grid");
  }
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  e->lx = lx;
  e->ly = ly;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return e->f32 ? set_target_grid_impl<float>(e, nx, ny, phi_vals, on_device, s)
                : set_target_grid_impl<double>(e, nx, ny, phi_vals, on_device, s);
}

eea_status eea_spatial_coeff_rows(eea_engine* e, unsigned nx, unsigned ny_total, unsigned row0,
                                  unsigned nrows, const void* d_phi_rows, double lx, double ly,
                                  void* d_phik_partial, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (finish_rebuild(e) != EEA_OK) return EEA_ERR_HIP;  // a rebuild that was only enqueued owns the buffers below
  if (d_phi_rows == nullptr || d_phik_partial == nullptr || nx == 0 || ny_total == 0 || nrows == 0 ||
      row0 + nrows > ny_total || !(lx > 0.0) || !(ly > 0.0)) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "bad grid tile");
  }
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double keep_lx = e->lx, keep_ly = e->ly;
  e->lx = lx;  // the tables are built for the tile's domain; the engine's own domain is restored
  e->ly = ly;
  st = e->f32 ? upload_axes_and_tables<float>(e, nx, ny_total, s) : upload_axes_and_tables<double>(e, nx, ny_total, s);
  e->lx = keep_lx;
  e->ly = keep_ly;
  if (st != EEA_OK) return st;
  st = e->f32 ? reserve_tile_work<float>(e, nx, ny_total, nrows) : reserve_tile_work<double>(e, nx, ny_total, nrows);
  if (st != EEA_OK) return st;
  if (e->f32) {
    EEA_HIP(eea::launch_spatial_coeff<float>(static_cast<const float*>(d_phi_rows), nx, nrows, e->K,
                                             static_cast<const float*>(e->d_cx.p),
                                             static_cast<const float*>(e->d_cy.p) + static_cast<size_t>(row0) * e->K,
                                             static_cast<float*>(e->d_work.p), static_cast<float*>(d_phik_partial), s));
  } else {
    EEA_HIP(eea::launch_spatial_coeff<double>(static_cast<const double*>(d_phi_rows), nx, nrows, e->K,
                                              static_cast<const double*>(e->d_cx.p),
                                              static_cast<const double*>(e->d_cy.p) + static_cast<size_t>(row0) * e->K,
                                              static_cast<double*>(e->d_work.p), static_cast<double*>(d_phik_partial), s));
  }
  return EEA_OK;
}

eea_status eea_set_target_occupancy(eea_engine* e, unsigned nx, unsigned ny, const int8_t* occ,
                                    int on_device, double lx, double ly, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (finish_rebuild(e) != EEA_OK) return EEA_ERR_HIP;  // a rebuild that was only enqueued owns the buffers below
  if (occ == nullptr || nx == 0 || ny == 0 || !(lx > 0.0) || !(ly > 0.0)) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "bad occupancy grid");
  }
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t P = static_cast<size_t>(nx) * ny;
  const int8_t* d_occ = occ;
  if (!on_device) {
    EEA_HIP(e->d_occ.reserve(P));
    EEA_HIP(hipMemcpyAsync(e->d_occ.p, occ, P, hipMemcpyHostToDevice, s));
    EEA_HIP(hipStreamSynchronize(s));
    d_occ = static_cast<const int8_t*>(e->d_occ.p);
  }
  e->lx = lx;
  e->ly = ly;
  e->nx = nx;
  e->ny = ny;
  e->have_fill_grid = false;
  EEA_HIP(e->d_raw.reserve(e->rs * e->K2));
  // the whole grid on this engine: the reduction launch normalises by mode (0, 0)'s sum itself (phi_k / sum(phi), target.cpp:87) --
  // two launches, the same bits as sums -> launch_normalise_by_first (which the row-tiled multi-rank form still uses, behind its
  // all-reduce); the normaliser goes to d_raw[0]
  st = e->f32 ? occupancy_rows_impl<float>(e, nx, ny, 0, ny, d_occ, e->d_phik.p, s, e->d_raw.p)
              : occupancy_rows_impl<double>(e, nx, ny, 0, ny, d_occ, e->d_phik.p, s, e->d_raw.p);
  if (st != EEA_OK) return st;
  EEA_HIP(hipStreamSynchronize(s));
  e->have_phik = true;
  ++e->phik_gen;  // (the resident single-robot workgroup restarts on the next call)
  return EEA_OK;
}

eea_status eea_spatial_coeff_occupancy_rows(eea_engine* e, unsigned nx, unsigned ny_total, unsigned row0,
                                            unsigned nrows, const int8_t* d_occ_rows, double lx, double ly,
                                            void* d_sums_partial, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (finish_rebuild(e) != EEA_OK) return EEA_ERR_HIP;  // a rebuild that was only enqueued owns the buffers below
  if (d_occ_rows == nullptr || d_sums_partial == nullptr || nx == 0 || ny_total == 0 || nrows == 0 ||
      row0 + nrows > ny_total || !(lx > 0.0) || !(ly > 0.0)) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "bad occupancy tile");
  }
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const double keep_lx = e->lx, keep_ly = e->ly;
  e->lx = lx;  // the tables are built for the tile's domain; the engine's own domain is restored
  e->ly = ly;
  st = e->f32 ? occupancy_rows_impl<float>(e, nx, ny_total, row0, nrows, d_occ_rows, d_sums_partial, s)
              : occupancy_rows_impl<double>(e, nx, ny_total, row0, nrows, d_occ_rows, d_sums_partial, s);
  e->lx = keep_lx;
  e->ly = keep_ly;
  return st;
}

eea_status eea_set_phik(eea_engine* e, const void* phik, int on_device, double lx, double ly)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (finish_rebuild(e) != EEA_OK) return EEA_ERR_HIP;  // a rebuild that was only enqueued owns the buffers below
  if (phik == nullptr || !(lx > 0.0) || !(ly > 0.0)) return fail(EEA_ERR_INVALID_ARGUMENT, "bad phi_k");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  EEA_HIP(hipMemcpy(e->d_phik.p, phik, e->rs * e->K2, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  e->lx = lx;
  e->ly = ly;
  e->have_phik = true;
  ++e->phik_gen;  // (the resident single-robot workgroup restarts on the next call)
  return EEA_OK;
}

eea_status eea_set_phik_from_sums(eea_engine* e, const void* d_sums, double lx, double ly, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (finish_rebuild(e) != EEA_OK) return EEA_ERR_HIP;  // a rebuild that was only enqueued owns the buffers below
  if (d_sums == nullptr || !(lx > 0.0) || !(ly > 0.0)) return fail(EEA_ERR_INVALID_ARGUMENT, "bad sums");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (e->f32) {
    EEA_HIP(eea::launch_normalise_by_first<float>(static_cast<const float*>(d_sums), e->K2,
                                                  static_cast<float*>(e->d_phik.p), s));
  } else {
    EEA_HIP(eea::launch_normalise_by_first<double>(static_cast<const double*>(d_sums), e->K2,
                                                   static_cast<double*>(e->d_phik.p), s));
  }
  e->lx = lx;
  e->ly = ly;
  e->have_phik = true;
  ++e->phik_gen;  // (the resident single-robot workgroup restarts on the next call)
  return EEA_OK;
}

static eea_status config_domain_impl(eea_engine* e, double xmin, double xmax, double ymin, double ymax, int* rebuilt,
                                     void* stream, bool wait)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (rebuilt) *rebuilt = 0;
  // translation from map to fourier domain is refreshed on every call (:366-367)
  e->map_x = xmin;
  e->map_y = ymin;
  const double mx = xmax - xmin, my = ymax - ymin;
  // almost_equal, numerics.hpp:68-71
  if (std::fabs(mx - e->lx) < 1.0e-12 && std::fabs(my - e->ly) < 1.0e-12) return EEA_OK;
  if (!e->have_gauss) return fail(EEA_ERR_NO_TARGET, "configTarget before setTarget");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  const double keep_lx = e->lx, keep_ly = e->ly;
  e->lx = mx;
  e->ly = my;
  hipStream_t s = static_cast<hipStream_t>(stream);
  st = e->f32 ? rebuild_phik<float>(e, s, wait) : rebuild_phik<double>(e, s, wait);
  if (st != EEA_OK) {
    // the reference throws out of the controller here; a failed rebuild must not leave the new extent
    // behind (the next call would take the almost_equal early return with a stale phi_k)
    e->lx = keep_lx;
    e->ly = keep_ly;
    return st;
  }
  if (rebuilt) *rebuilt = 1;
  return st;
}

eea_status eea_config_domain(eea_engine* e, double xmin, double xmax, double ymin, double ymax,
                             int* rebuilt, void* stream)
{
  return config_domain_impl(e, xmin, xmax, ymin, ymax, rebuilt, stream, true);
}

eea_status eea_config_domain_async(eea_engine* e, double xmin, double xmax, double ymin, double ymax,
                                   int* rebuilt, void* stream)
{
  return config_domain_impl(e, xmin, xmax, ymin, ymax, rebuilt, stream, false);
}

static eea_status download_reals(eea_engine* e, const void* d, size_t n, double* out)
{
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  if (e->f32) {
    std::vector<float> tmp(n);
    EEA_HIP(hipMemcpy(tmp.data(), d, sizeof(float) * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) out[i] = tmp[i];
  } else {
    EEA_HIP(hipMemcpy(out, d, sizeof(double) * n, hipMemcpyDeviceToHost));
  }
  return EEA_OK;
}

eea_status eea_get_phik(eea_engine* e, double* h_phik)
{
  if (check_engine(e) != EEA_OK || h_phik == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  if (finish_rebuild(e) != EEA_OK) return EEA_ERR_HIP;
  return download_reals(e, e->d_phik.p, e->K2, h_phik);
}

eea_status eea_get_lamdak(eea_engine* e, double* h_lamdak)
{
  if (check_engine(e) != EEA_OK || h_lamdak == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  return download_reals(e, e->d_lamdak.p, e->K2, h_lamdak);
}

eea_status eea_target_grid_size(const eea_engine* e, unsigned* nx, unsigned* ny)
{
  if (e == nullptr || nx == nullptr || ny == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  *nx = e->nx;
  *ny = e->ny;
  return EEA_OK;
}

eea_status eea_get_target_grid(eea_engine* e, double* h_phi_vals)
{
  if (check_engine(e) != EEA_OK || h_phi_vals == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  if (e->nx == 0 || !e->have_fill_grid || (e->d_phi.p == nullptr && !e->fill_deferred)) {
    return fail(EEA_ERR_NO_TARGET, "no Target::fill grid on the device (explicit / occupancy targets are not kept)");
  }
  if (finish_rebuild(e) != EEA_OK) return EEA_ERR_HIP;
  const size_t P = static_cast<size_t>(e->nx) * e->ny;
  if (e->fill_deferred) {  // the Gaussian rebuild never needed the grid: fill it now (un-normalised; the mass is in d_sum[0])
    const eea_status stf = e->f32 ? fill_deferred_grid<float>(e) : fill_deferred_grid<double>(e);
    if (stf != EEA_OK) return stf;
  }
  eea_status st = download_reals(e, e->d_phi.p, P, h_phi_vals);
  if (st != EEA_OK || !e->phi_is_raw) return st;
  // the device keeps the un-normalised grid and its mass; phi_vals / sum(phi_vals) as target.cpp:87 forms it
  double mass = 0.0;
  st = download_reals(e, e->d_sum.p, 1, &mass);
  if (st != EEA_OK) return st;
  if (e->f32) {
    const float m = static_cast<float>(mass);
    for (size_t i = 0; i < P; ++i) h_phi_vals[i] = static_cast<double>(static_cast<float>(h_phi_vals[i]) / m);
  } else {
    for (size_t i = 0; i < P; ++i) h_phi_vals[i] = h_phi_vals[i] / mass;
  }
  return EEA_OK;
}

eea_status eea_control_batch(eea_engine* e, unsigned B, const eea_batch_io* io, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (io == nullptr || io->d_pose == nullptr || io->d_ut == nullptr || io->d_u0 == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "d_pose, d_ut and d_u0 are required");
  }
  if (!e->have_phik) return fail(EEA_ERR_NO_TARGET, "no phi_k: call eea_config_domain or eea_set_target_grid first");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  st = order_after_rebuild(e, s);  // a phi_k rebuild that was only enqueued on another stream (eea_config_domain_async)
  if (st != EEA_OK) return st;
  return e->f32 ? control_batch_impl<float>(e, B, io, false, s)
                : control_batch_impl<double>(e, B, io, false, s);
}

eea_status eea_control_batch_steps(eea_engine* e, unsigned B, const eea_batch_io* io, unsigned n_steps,
                                   unsigned pose_step_stride, unsigned u0_step_stride, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (io == nullptr || io->d_pose == nullptr || io->d_ut == nullptr || io->d_u0 == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "d_pose, d_ut and d_u0 are required");
  }
  if (n_steps == 0 || n_steps > (1u << 20)) return fail(EEA_ERR_INVALID_ARGUMENT, "n_steps must be in 1 .. 2^20");
  if ((pose_step_stride != 0 && pose_step_stride < B) || (u0_step_stride != 0 && u0_step_stride < B)) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "a step stride is 0 (the same row every step) or >= B agents");
  }
  if (n_steps > 1 && (io->d_rec_ready != nullptr || io->d_ck_flag != nullptr)) {
    // a step of the launch would wait, inside the kernel, for an exchange the host can only enqueue after this call: that
    // needs truly concurrent hardware queues (streams may share one) -- the device-bound exchange is one step per launch
    return fail(EEA_ERR_UNSUPPORTED, "the device-bound exchange (d_rec_ready / d_ck_flag) takes one step per launch");
  }
  if (!e->have_phik) return fail(EEA_ERR_NO_TARGET, "no phi_k: call eea_config_domain or eea_set_target_grid first");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  st = order_after_rebuild(e, s);
  if (st != EEA_OK) return st;
  return e->f32 ? control_batch_impl<float>(e, B, io, false, s, nullptr, n_steps, pose_step_stride, u0_step_stride)
                : control_batch_impl<double>(e, B, io, false, s, nullptr, n_steps, pose_step_stride, u0_step_stride);
}

static eea_status records_sum_impl(eea_engine* e, unsigned B, const void* d_ck_rec, void* d_sum, void* stream,
                                   const unsigned* d_ready, unsigned seq, unsigned* d_flag)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (d_ck_rec == nullptr || d_sum == nullptr || B == 0) return fail(EEA_ERR_INVALID_ARGUMENT, "d_ck_rec, d_sum and B > 0 are required");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  eea_engine::SumWs* w = nullptr;
  std::lock_guard<std::mutex> lock(e->sum_mutex);  // across the launch: the workspace must not be recycled under it
  st = e->f32 ? sum_workspace<float>(e, d_sum, B, s, &w) : sum_workspace<double>(e, d_sum, B, s, &w);
  if (st != EEA_OK) return st;
  if (e->f32) {
    EEA_HIP(eea::launch_ck_records_sum<float>(static_cast<const float*>(d_ck_rec), B, e->K2, static_cast<float*>(w->ws.p),
                                              static_cast<unsigned*>(w->ctr.p), static_cast<float*>(d_sum), s, d_ready, seq,
                                              d_flag));
  } else {
    EEA_HIP(eea::launch_ck_records_sum<double>(static_cast<const double*>(d_ck_rec), B, e->K2, static_cast<double*>(w->ws.p),
                                               static_cast<unsigned*>(w->ctr.p), static_cast<double*>(d_sum), s, d_ready, seq,
                                               d_flag));
  }
  return EEA_OK;
}

eea_status eea_ck_records_sum(eea_engine* e, unsigned B, const void* d_ck_rec, void* d_sum, void* stream)
{
  return records_sum_impl(e, B, d_ck_rec, d_sum, stream, nullptr, 0u, nullptr);
}

eea_status eea_ck_records_sum_ws_bytes(const eea_engine* e, unsigned B, size_t* ws_bytes, size_t* ticket_bytes)
{
  if (check_engine(e) != EEA_OK || B == 0) return fail(EEA_ERR_INVALID_ARGUMENT, "null engine / B == 0");
  if (ws_bytes) *ws_bytes = (e->f32 ? sizeof(float) : sizeof(double)) * eea::ck_sum_ws_elems(B, e->K2);
  if (ticket_bytes) *ticket_bytes = sizeof(unsigned) * eea::ck_sum_tickets(B, e->K2);
  return EEA_OK;
}

// the plain record sum with a caller-owned workspace: no allocation, no cache look-up, nothing but the launch (capturable)
eea_status eea_ck_records_sum_ws(eea_engine* e, unsigned B, const void* d_ck_rec, void* d_sum, void* d_ws, void* d_tickets,
                                 void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (d_ck_rec == nullptr || d_sum == nullptr || d_ws == nullptr || d_tickets == nullptr || B == 0) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "d_ck_rec, d_sum, d_ws, d_tickets and B > 0 are required");
  }
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (e->f32) {
    EEA_HIP(eea::launch_ck_records_sum<float>(static_cast<const float*>(d_ck_rec), B, e->K2, static_cast<float*>(d_ws),
                                              static_cast<unsigned*>(d_tickets), static_cast<float*>(d_sum), s, nullptr, 0u, nullptr));
  } else {
    EEA_HIP(eea::launch_ck_records_sum<double>(static_cast<const double*>(d_ck_rec), B, e->K2, static_cast<double*>(d_ws),
                                               static_cast<unsigned*>(d_tickets), static_cast<double*>(d_sum), s, nullptr, 0u, nullptr));
  }
  return EEA_OK;
}

eea_status eea_ck_records_sum_bound(eea_engine* e, unsigned B, const void* d_ck_rec, const unsigned* d_rec_ready, unsigned seq,
                                    void* d_sum, unsigned* d_flag, void* stream)
{
  if (d_rec_ready == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "d_rec_ready is required (eea_batch_io::d_rec_ready of the producing calls)");
  return records_sum_impl(e, B, d_ck_rec, d_sum, stream, d_rec_ready, seq, d_flag);
}

eea_status eea_publish_record(eea_engine* e, const void* d_src, void* d_pub, unsigned* d_flag, unsigned seq, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (d_src == nullptr || d_pub == nullptr || d_flag == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  const int n = eea::ck_record_len(e->K2);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (e->f32) EEA_HIP(eea::launch_publish_record<float>(static_cast<const float*>(d_src), n, static_cast<float*>(d_pub), d_flag, seq, s));
  else EEA_HIP(eea::launch_publish_record<double>(static_cast<const double*>(d_src), n, static_cast<double*>(d_pub), d_flag, seq, s));
  return EEA_OK;
}

eea_status eea_rollout_batch(eea_engine* e, unsigned B, const void* d_pose, const void* d_ut,
                             void* d_traj, int* d_status, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (d_pose == nullptr || d_ut == nullptr || d_traj == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "d_pose, d_ut and d_traj are required");
  }
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  eea_batch_io io;
  std::memset(&io, 0, sizeof(io));
  io.d_pose = d_pose;
  io.d_ut = const_cast<void*>(d_ut);  // rollout_only never writes ut
  io.d_traj = d_traj;
  io.d_status = d_status;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return e->f32 ? control_batch_impl<float>(e, B, &io, true, s)
                : control_batch_impl<double>(e, B, &io, true, s);
}

eea_status eea_control(eea_engine* e, double xmin, double xmax, double ymin, double ymax,
                       const double x[3], const double* h_mem_cols, unsigned n_mem, double u_out[3])
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (x == nullptr || u_out == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  if (n_mem > 0 && h_mem_cols == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null mem_cols");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  // pose_ = x; configTarget(grid)  (:227-230): enqueued on the same stream as the control kernel, no host wait
  st = eea_config_domain_async(e, xmin, xmax, ymin, ymax, nullptr, e->stream1);
  if (st != EEA_OK) return st;
  if (!e->have_phik) return fail(EEA_ERR_NO_TARGET, "no target set");
  for (int i = 0; i < 3; ++i) e->last_pose[i] = x[i];
  // one robot, one control() per tick without a launch per call (EEA_OPT_RESIDENT_CONTROL); a replay-memory sample beyond
  // the mapped buffer takes the launch path
  if (eea::option(EEA_OPT_RESIDENT_CONTROL) != 0 && n_mem <= static_cast<unsigned>(eea::kResidentMemCols)) {
    st = e->f32 ? control_resident<float>(e, x, h_mem_cols, n_mem, u_out) : control_resident<double>(e, x, h_mem_cols, n_mem, u_out);
    if (st != EEA_ERR_UNSUPPORTED) return st;  // (UNSUPPORTED: the shape does not fit a resident workgroup -- launch path)
  } else {
    st = resident_stop(e);  // (the option was switched off, or this call does not fit: the launch path owns d_ut1 now)
    if (st != EEA_OK) return st;
  }

  eea_batch_io io;
  std::memset(&io, 0, sizeof(io));
  if (n_mem > 0) {
    const size_t bytes = e->rs * 3 * n_mem;
    st = stage_reserve(e, bytes);
    if (st != EEA_OK) return st;
    {
      size_t want = 4096;  // power-of-two steps: the replay memory grows by one column per tick
      while (want < bytes) want *= 2;
      EEA_HIP(e->d_mem1.reserve(want));
    }
    if (e->f32) to_real<float>(h_mem_cols, static_cast<float*>(e->h_stage), 3 * static_cast<size_t>(n_mem));
    else std::memcpy(e->h_stage, h_mem_cols, bytes);
    EEA_HIP(hipMemcpyAsync(e->d_mem1.p, e->h_stage, bytes, hipMemcpyHostToDevice, e->stream1));
    io.d_mem_cols = e->d_mem1.p;
    io.mem_stride = n_mem;  // d_n_mem == NULL: every reserved column is valid
  }
  if (e->f32) to_real<float>(x, reinterpret_cast<float*>(e->h_mail->pose), 3);
  else std::memcpy(e->h_mail->pose, x, sizeof(double) * 3);
  e->h_mail->status = 0;
  Mailbox* const dm = static_cast<Mailbox*>(e->d_mail);
  io.d_pose = dm->pose;
  io.d_u0 = dm->u0;
  io.d_status = &dm->status;
  io.d_ut = e->d_ut1.p;
  e->mail_seq = (e->mail_seq % 1000000) + 1;
  e->mail_done = &dm->done;
  st = e->f32 ? control_batch_impl<float>(e, 1, &io, false, e->stream1)
              : control_batch_impl<double>(e, 1, &io, false, e->stream1);
  e->mail_done = nullptr;
  if (st != EEA_OK) return st;
  // the kernel publishes u0 / status and then the sequence number with a system-scope release: poll it
  // (a few microseconds earlier than the stream's completion signal); bounded, then the ordinary wait
  const bool poll = eea::option(EEA_OPT_MAILBOX_POLL) != 0;
  bool seen = false;
  if (poll) {
    const volatile int* const flag = &e->h_mail->done;
    for (int spin = 0; spin < 200000; ++spin) {
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == e->mail_seq) {
        seen = true;
        break;
      }
    }
  }
  if (!seen) EEA_HIP(hipStreamSynchronize(e->stream1));
  if (e->h_mail->status == EEA_ERR_INVALID_TWIST) {
    return fail(EEA_ERR_INVALID_TWIST, "Invalid twist y-velocity must be 0.");
  }
  if (e->f32) {
    const float* u = reinterpret_cast<const float*>(e->h_mail->u0);
    for (int i = 0; i < 3; ++i) u_out[i] = u[i];
  } else {
    std::memcpy(u_out, e->h_mail->u0, sizeof(double) * 3);
  }
  return EEA_OK;
}

eea_status eea_resident_stop(eea_engine* e)
{
  if (check_engine(e) != EEA_OK) return fail(EEA_ERR_INVALID_ARGUMENT, "null engine");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  return resident_stop(e);
}

eea_status eea_opt_traj(eea_engine* e, double* h_traj)
{
  if (check_engine(e) != EEA_OK || h_traj == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  if (e->f32) to_real<float>(e->last_pose, reinterpret_cast<float*>(e->h_mail->pose), 3);
  else std::memcpy(e->h_mail->pose, e->last_pose, sizeof(double) * 3);
  e->h_mail->status = 0;
  Mailbox* const dm = static_cast<Mailbox*>(e->d_mail);
  st = eea_rollout_batch(e, 1, dm->pose, e->d_ut1.p, e->d_traj1.p, &dm->status, e->stream1);
  if (st != EEA_OK) return st;
  EEA_HIP(hipStreamSynchronize(e->stream1));
  if (e->h_mail->status == EEA_ERR_INVALID_TWIST) {
    return fail(EEA_ERR_INVALID_TWIST, "Invalid twist y-velocity must be 0.");
  }
  return download_reals(e, e->d_traj1.p, 3 * static_cast<size_t>(e->T), h_traj);
}

eea_status eea_get_ut(eea_engine* e, double* h_ut)
{
  if (check_engine(e) != EEA_OK || h_ut == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  EEA_HIP(hipStreamSynchronize(e->stream1));
  return download_reals(e, e->d_ut1.p, 3 * static_cast<size_t>(e->T), h_ut);
}

eea_status eea_set_ut(eea_engine* e, const double* h_ut)
{
  if (check_engine(e) != EEA_OK || h_ut == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  eea_status st = use_device(e);
  if (st != EEA_OK) return st;
  const size_t n = 3 * static_cast<size_t>(e->T);
  if (e->res_launched && e->res_one_wavefront) {  // its warm start is in LDS: the next call starts another one from d_ut1
    st = resident_stop(e);
    if (st != EEA_OK) return st;
  }
  EEA_HIP(hipStreamSynchronize(e->stream1));
  if (e->f32) {
    std::vector<float> tmp(n);
    to_real<float>(h_ut, tmp.data(), n);
    EEA_HIP(hipMemcpy(e->d_ut1.p, tmp.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  } else {
    EEA_HIP(hipMemcpy(e->d_ut1.p, h_ut, sizeof(double) * n, hipMemcpyHostToDevice));
  }
  return EEA_OK;
}

// ---- Basis free functions ------------------------------------------------------------------
static eea_status basis_points(int device, double lx, double ly, unsigned K, const double* xs,
                               const double* ys, const double* ws, unsigned P, double scale,
                               double* out)
{
  if (K == 0 || K > static_cast<unsigned>(eea::kMaxBasis)) return fail(EEA_ERR_UNSUPPORTED, "num_basis must be in [1, 32]");
  EEA_HIP(hipSetDevice(device));
  const size_t K2 = static_cast<size_t>(K) * K;
  DevBuf dx, dy, dw, dwork, dout;
  auto cleanup = [&]() {
    dx.release();
    dy.release();
    dw.release();
    dwork.release();
    dout.release();
  };
  hipError_t err = dx.reserve(sizeof(double) * (P ? P : 1));
  if (err == hipSuccess) err = dy.reserve(sizeof(double) * (P ? P : 1));
  if (err == hipSuccess && ws) err = dw.reserve(sizeof(double) * (P ? P : 1));
  if (err == hipSuccess) err = dwork.reserve(sizeof(double) * eea::point_work_elems(P, K));
  if (err == hipSuccess) err = dout.reserve(sizeof(double) * K2);
  if (err == hipSuccess && P) err = hipMemcpy(dx.p, xs, sizeof(double) * P, hipMemcpyHostToDevice);
  if (err == hipSuccess && P) err = hipMemcpy(dy.p, ys, sizeof(double) * P, hipMemcpyHostToDevice);
  if (err == hipSuccess && P && ws) err = hipMemcpy(dw.p, ws, sizeof(double) * P, hipMemcpyHostToDevice);
  if (err == hipSuccess) {
    err = eea::launch_point_coeff<double>(static_cast<const double*>(dx.p), static_cast<const double*>(dy.p),
                                          ws ? static_cast<const double*>(dw.p) : nullptr, P, K,
                                          eea::kPi / lx, eea::kPi / ly, scale,
                                          static_cast<double*>(dwork.p), static_cast<double*>(dout.p), nullptr);
  }
  if (err == hipSuccess) err = hipMemcpy(out, dout.p, sizeof(double) * K2, hipMemcpyDeviceToHost);
  cleanup();
  if (err != hipSuccess) return fail(EEA_ERR_HIP, std::string("basis op: ") + hipGetErrorString(err));
  return EEA_OK;
}

eea_status eea_basis_traj_coeff(int device, double lx, double ly, unsigned num_basis,
                                const double* h_xt, unsigned rows, unsigned n, double* h_ck)
{
  if (h_xt == nullptr || h_ck == nullptr || rows < 2 || n == 0) return fail(EEA_ERR_INVALID_ARGUMENT, "bad trajectory");
  std::vector<double> xs(n), ys(n);
  for (unsigned i = 0; i < n; ++i) {
    xs[i] = h_xt[static_cast<size_t>(rows) * i];
    ys[i] = h_xt[static_cast<size_t>(rows) * i + 1];
  }
  return basis_points(device, lx, ly, num_basis, xs.data(), ys.data(), nullptr, n, 1.0 / static_cast<double>(n), h_ck);
}

eea_status eea_basis_spatial_coeff(int device, double lx, double ly, unsigned num_basis,
                                   const double* h_phi_vals, const double* h_phi_grid, unsigned P,
                                   double* h_phik)
{
  if (h_phi_vals == nullptr || h_phi_grid == nullptr || h_phik == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  std::vector<double> xs(P), ys(P);
  for (unsigned i = 0; i < P; ++i) {
    xs[i] = h_phi_grid[2 * static_cast<size_t>(i)];
    ys[i] = h_phi_grid[2 * static_cast<size_t>(i) + 1];
  }
  return basis_points(device, lx, ly, num_basis, xs.data(), ys.data(), h_phi_vals, P, 1.0, h_phik);
}

eea_status eea_rk4_rollout(int device, int model, double dt, double horizon, const double x0[3],
                           const double* h_ut, double* h_xt)
{
  if (x0 == nullptr || h_ut == nullptr || h_xt == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  if (model != EEA_MODEL_OMNI && model != EEA_MODEL_SIMPLE_CART) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "device rollout needs a body-twist model (Omni / SimpleCart)");
  }
  const double ratio = std::abs(horizon / dt);
  if (!(ratio < 1.0e6)) return fail(EEA_ERR_INVALID_ARGUMENT, "horizon / dt out of range");
  const unsigned steps = static_cast<unsigned>(ratio);  // integrator.hpp:141
  if (steps == 0) return EEA_OK;
  EEA_HIP(hipSetDevice(device));
  DevBuf dpose, dut, dtraj, dstat;
  hipError_t err = dpose.reserve(sizeof(double) * 3);
  if (err == hipSuccess) err = dut.reserve(sizeof(double) * 3 * steps);
  if (err == hipSuccess) err = dtraj.reserve(sizeof(double) * 3 * steps);
  if (err == hipSuccess) err = dstat.reserve(sizeof(int));
  if (err == hipSuccess) err = hipMemcpy(dpose.p, x0, sizeof(double) * 3, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dut.p, h_ut, sizeof(double) * 3 * steps, hipMemcpyHostToDevice);
  int status = 0;
  if (err == hipSuccess) {
    eea::ControlParams<double> p;
    std::memset(&p, 0, sizeof(p));
    p.T = static_cast<int>(steps);
    p.K = 1;
    p.n_steps = 1;
    p.chunk = 64;
    p.dt = dt;
    p.dt6 = dt / 6.0;
    p.half_dt = 0.5 * dt;
    p.lx = p.ly = 1.0;
    p.inv_lx = p.inv_ly = 1.0;
    p.pose = static_cast<const double*>(dpose.p);
    p.ut = static_cast<double*>(dut.p);
    p.traj = static_cast<double*>(dtraj.p);
    p.status = static_cast<int*>(dstat.p);
    if (eea::control_lds_bytes<double>(p.T, p.K, 0, p.chunk) > 160 * 1024) {
      err = hipErrorInvalidValue;
    } else {
      err = eea::launch_control<double>(p, 1, model, 0, true, nullptr);
    }
  }
  if (err == hipSuccess) err = hipMemcpy(&status, dstat.p, sizeof(int), hipMemcpyDeviceToHost);
  if (err == hipSuccess && status == 0) err = hipMemcpy(h_xt, dtraj.p, sizeof(double) * 3 * steps, hipMemcpyDeviceToHost);
  dpose.release();
  dut.release();
  dtraj.release();
  dstat.release();
  if (err != hipSuccess) return fail(EEA_ERR_HIP, std::string("rk4 rollout: ") + hipGetErrorString(err));
  if (status == EEA_ERR_INVALID_TWIST) return fail(EEA_ERR_INVALID_TWIST, "Invalid twist y-velocity must be 0.");
  return EEA_OK;
}

eea_status eea_target_fill(int device, unsigned n_gauss, const double* mu, const double* sigma,
                           const double trans[2], const double* h_phi_grid, unsigned P,
                           double* h_phi_vals)
{
  if (trans == nullptr || h_phi_grid == nullptr || h_phi_vals == nullptr || (n_gauss > 0 && (mu == nullptr || sigma == nullptr))) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  }
  if (P == 0) return EEA_OK;
  EEA_HIP(hipSetDevice(device));
  std::vector<double> xs(P), ys(P), g(4 * static_cast<size_t>(n_gauss ? n_gauss : 1));
  for (unsigned i = 0; i < P; ++i) {
    xs[i] = h_phi_grid[2 * static_cast<size_t>(i)];
    ys[i] = h_phi_grid[2 * static_cast<size_t>(i) + 1];
  }
  for (unsigned i = 0; i < n_gauss; ++i) {  // target.hpp:69,99
    const double a = sigma[2 * i] * sigma[2 * i], d = sigma[2 * i + 1] * sigma[2 * i + 1];
    const double det = a * d - 0.0 * 0.0;
    g[4 * i + 0] = mu[2 * i] - trans[0];
    g[4 * i + 1] = mu[2 * i + 1] - trans[1];
    g[4 * i + 2] = d / det;
    g[4 * i + 3] = a / det;
  }
  const int blocks = static_cast<int>((P + eea::kBlock - 1) / eea::kBlock);
  DevBuf dx, dy, dg, dphi, dsum;
  hipError_t err = dx.reserve(sizeof(double) * P);
  if (err == hipSuccess) err = dy.reserve(sizeof(double) * P);
  if (err == hipSuccess) err = dg.reserve(sizeof(double) * g.size());
  if (err == hipSuccess) err = dphi.reserve(sizeof(double) * P);
  if (err == hipSuccess) err = dsum.reserve(sizeof(double) * (static_cast<size_t>(blocks) + 1));
  if (err == hipSuccess) err = hipMemcpy(dx.p, xs.data(), sizeof(double) * P, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dy.p, ys.data(), sizeof(double) * P, hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMemcpy(dg.p, g.data(), sizeof(double) * g.size(), hipMemcpyHostToDevice);
  int n_partials = 0;
  double* const d_partials = dsum.p ? static_cast<double*>(dsum.p) + 1 : nullptr;
  if (err == hipSuccess) {
    err = eea::launch_target_fill_points<double>(static_cast<const double*>(dx.p), static_cast<const double*>(dy.p), P,
                                                 static_cast<const double*>(dg.p), static_cast<int>(n_gauss),
                                                 static_cast<double*>(dphi.p), d_partials, &n_partials, nullptr);
  }
  if (err == hipSuccess) err = eea::launch_reduce_sum<double>(d_partials, n_partials, static_cast<double*>(dsum.p), nullptr);
  if (err == hipSuccess) err = eea::launch_scale_by_inv<double>(static_cast<double*>(dphi.p), P, static_cast<const double*>(dsum.p), nullptr);
  if (err == hipSuccess) err = hipMemcpy(h_phi_vals, dphi.p, sizeof(double) * P, hipMemcpyDeviceToHost);
  dx.release();
  dy.release();
  dg.release();
  dphi.release();
  dsum.release();
  if (err != hipSuccess) return fail(EEA_ERR_HIP, std::string("target fill: ") + hipGetErrorString(err));
  return EEA_OK;
}

// ---- collision lookups ---------------------------------------------------------------------
static eea_status make_collision_params(const eea_collision_cfg* cfg, eea::CollisionParams& c)
{
  if (cfg == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null collision config");
  // Collision::Collision (collision.cpp:46-63)
  if (cfg->search_radius < cfg->boundary_radius) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "Search radius must be at least the same size as the boundary radius");
  }
  if (cfg->occupied_threshold > 100.0 || cfg->occupied_threshold < 0.0) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "Occupied threshold must be between 0 and 100");
  }
  c.xmin = cfg->xmin;
  c.ymin = cfg->ymin;
  c.resolution = cfg->resolution;
  c.xsize = cfg->xsize;
  c.ysize = cfg->ysize;
  // CollisionConfig radii in cells (collision.cpp:130-133)
  c.r_bnd = static_cast<int>(std::floor(cfg->boundary_radius / cfg->resolution));
  c.r_col = static_cast<int>(std::floor((cfg->boundary_radius + cfg->obstacle_threshold) / cfg->resolution));
  c.r_max = static_cast<int>(std::floor(cfg->search_radius / cfg->resolution));
  c.occupied_threshold = cfg->occupied_threshold;
  return EEA_OK;
}

eea_status eea_collision_check_batch(int device, const eea_collision_cfg* cfg, const int8_t* d_grid,
                                     const double* d_pose, unsigned P, int* d_hit, void* stream)
{
  eea::CollisionParams c;
  eea_status st = make_collision_params(cfg, c);
  if (st != EEA_OK) return st;
  if (d_grid == nullptr || d_pose == nullptr || d_hit == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  EEA_HIP(hipSetDevice(device));
  EEA_HIP(eea::launch_collision_check(c, d_grid, d_pose, P, d_hit, static_cast<hipStream_t>(stream)));
  return EEA_OK;
}

eea_status eea_validate_control_batch(int device, const eea_collision_cfg* cfg, const int8_t* d_grid,
                                      const double* d_x0, const double* d_u, double dt,
                                      double horizon, unsigned P, int* d_valid, void* stream)
{
  eea::CollisionParams c;
  eea_status st = make_collision_params(cfg, c);
  if (st != EEA_OK) return st;
  if (d_grid == nullptr || d_x0 == nullptr || d_u == nullptr || d_valid == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  }
  const unsigned steps = static_cast<unsigned>(std::abs(horizon / dt));  // numerics.hpp:317
  EEA_HIP(hipSetDevice(device));
  EEA_HIP(eea::launch_validate_control(c, d_grid, d_x0, d_u, dt, steps, P, d_valid,
                                       static_cast<hipStream_t>(stream)));
  return EEA_OK;
}

eea_status eea_integrate_twist_batch(int device, const double* d_x0, const double* d_u, double dt, unsigned P, double* d_out,
                                     int normalize_heading, void* stream)
{
  if (d_x0 == nullptr || d_u == nullptr || d_out == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  EEA_HIP(hipSetDevice(device));
  EEA_HIP(eea::launch_integrate_twist(d_x0, d_u, dt, P, d_out, normalize_heading != 0, static_cast<hipStream_t>(stream)));
  return EEA_OK;
}

void eea_release_collision_caches(void) { eea::release_collision_caches(); }

// One tick of Exploration<ModelT>::control's loop body for B robots (exploration.hpp:220-279): see ergodic_amd.h
eea_status eea_tick_batch(eea_engine* e, unsigned B, const eea_batch_io* io, const eea_tick_io* tick,
                          const eea_collision_cfg* ccfg, const eea_dwa_cfg* dcfg, void* stream)
{
  if (check_engine(e) != EEA_OK) return EEA_ERR_INVALID_ARGUMENT;
  if (io == nullptr || tick == nullptr || dcfg == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  if (e->f32) return fail(EEA_ERR_UNSUPPORTED, "eea_tick_batch takes fp64 engines (poses and twists are doubles)");
  if (io->d_pose == nullptr || io->d_ut == nullptr || tick->d_follow_dwa == nullptr || tick->d_dwa_count == nullptr ||
      tick->d_u == nullptr || tick->d_vb == nullptr || tick->d_grid == nullptr || tick->d_traj == nullptr ||
      tick->d_valid == nullptr || tick->d_skip == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "null tick buffer");
  }
  if (io->d_rec_ready != nullptr || io->d_ck_flag != nullptr) {
    return fail(EEA_ERR_UNSUPPORTED, "eea_tick_batch does not take the device-bound exchange buffers");
  }
  if (B == 0) return EEA_OK;
  eea::CollisionParams c;
  eea_status st = make_collision_params(ccfg, c);
  if (st != EEA_OK) return st;
  eea::DwaParams d;
  st = make_dwa_params(dcfg, d);
  if (st != EEA_OK) return st;
  if (!e->have_phik) return fail(EEA_ERR_NO_TARGET, "no target set");
  st = use_device(e);
  if (st != EEA_OK) return st;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // 1. followers count a step (exploration.hpp:223-228); d_skip = who follows a DWA twist this tick
  EEA_HIP(eea::launch_tick_begin(tick->d_follow_dwa, tick->d_dwa_count, tick->d_skip, d.steps, B, s));
  // 2. control() of everybody else (:230-236), the twist straight into d_u; then optTraj() of the updated controls (:257)
  eea_batch_io cio = *io;
  cio.d_u0 = tick->d_u;
  cio.d_skip = tick->d_skip;
  cio.d_traj = nullptr;
  st = control_batch_impl<double>(e, B, &cio, false, s);
  if (st != EEA_OK) return st;
  eea_batch_io rio{};
  rio.d_pose = io->d_pose;
  rio.d_ut = io->d_ut;
  rio.d_traj = tick->d_traj;
  rio.d_skip = tick->d_skip;
  st = control_batch_impl<double>(e, B, &rio, true, s);
  if (st != EEA_OK) return st;
  // 3. validate_control (:238, numerics.hpp:312-330); 4. the dynamic window where the twist was rejected (:240-277), u /
  // follow_dwa / i updated in place.  One inflated map for both
  const unsigned val_steps = static_cast<unsigned>(std::abs(tick->val_horizon / tick->val_dt));
  EEA_HIP(eea::launch_validate_and_dwa_fleet(c, d, tick->d_grid, tick->grid_epoch, static_cast<const double*>(io->d_pose),
                                             tick->d_vb, tick->d_traj, static_cast<unsigned>(e->T), e->cfg.dt, tick->val_dt,
                                             val_steps, tick->d_valid, tick->d_follow_dwa, tick->d_dwa_count, tick->d_u,
                                             tick->d_source, B, s));
  return EEA_OK;
}

eea_status eea_dwa_control_batch(int device, const eea_collision_cfg* ccfg, const eea_dwa_cfg* dcfg,
                                 const int8_t* d_grid, const double* d_x0, const double* d_vb,
                                 const double* d_vref, const double* d_xt_ref, unsigned n_ref,
                                 double dt_ref, unsigned P, double* d_u_opt, int* d_found, void* stream)
{
  eea::CollisionParams c;
  eea_status st = make_collision_params(ccfg, c);
  if (st != EEA_OK) return st;
  if (dcfg == nullptr || d_grid == nullptr || d_x0 == nullptr || d_vb == nullptr || d_u_opt == nullptr ||
      d_found == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  }
  if ((d_vref == nullptr) == (d_xt_ref == nullptr)) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "give either d_vref or d_xt_ref");
  }
  if (d_xt_ref != nullptr && n_ref == 0) return fail(EEA_ERR_INVALID_ARGUMENT, "empty reference trajectory");
  eea::DwaParams d;
  st = make_dwa_params(dcfg, d);
  if (st != EEA_OK) return st;
  EEA_HIP(hipSetDevice(device));
  EEA_HIP(eea::launch_dwa_control(c, d, d_grid, d_x0, d_vb, d_vref, d_xt_ref, n_ref, dt_ref, P, d_u_opt,
                                  d_found, static_cast<hipStream_t>(stream)));
  return EEA_OK;
}

}  // extern "C"

// the A/B library (make AB=1; tools/ab/) adds its diagnostic entry point here; the product does not
#ifdef EEA_AB_BUILD
#include "../../tools/ab/engine_ab.inc"
#endif
