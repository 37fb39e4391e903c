// Shared declarations of the gfx950 kernels behind include/ergodic_amd.h.
// Host launchers are declared here and defined next to their kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace eea
{
constexpr int kBlock = 256;    // threads per workgroup: 4 wavefronts of 64 lanes
constexpr int kWave = 64;
constexpr int kMaxBasis = 32;  // K <= 32 (K^2 <= 1024 modes)
constexpr unsigned kSumGroup = 32, kSumFan = 8;  // record sum: agents per level-0 unit, level-0 records per level-1 record
__host__ __device__ constexpr int ck_record_len(int K2) { return (K2 + 2) & ~1; }

// numerics.hpp:59 of the reference
constexpr double kPi = 3.14159265358979323846;

enum : int { kModelOmni = 0, kModelSimpleCart = 1 };

// process-wide dispatch option (eea_set_option, EEA_OPT_* of include/ergodic_amd.h); defined in engine.cpp
int option(int id);
// Everything one control launch needs; passed by value (kernarg).
template <typename R>
struct ControlParams
{
  int T;              // steps_ (ergodic_control.hpp:199)
  int K;              // num_basis
  int chunk;          // points staged per contraction pass (multiple of 64)
  unsigned mem_stride;
  R dt, dt6, lx, ly, map_x, map_y, expl_weight;  // dt6 = dt / 6 (integrator.hpp:183,193)
  R half_dt;                                     // dt / 2, formed on the host: stays in scalar registers
  R pi_lx, pi_ly;     // PI / lx, PI / ly (basis.cpp:85)
  R inv_lx, inv_ly;   // 1 / lx, 1 / ly (sin/cos(pi x / lx) evaluation)
  R Rinv[9];          // column-major
  R umin[3], umax[3];
  const R* phik;      // [K^2]
  const R* lamdak;    // [K^2]
  // per-agent buffers (see eea_batch_io)
  const R* pose;
  R* ut;
  const R* mem_cols;
  const int* n_mem;
  R* u0;
  R* traj;
  R* ck;
  const R* ck_shared;  // [K^2] optional: consensus c_k used in place of the agent's own (one per launch); with
                       // ck_shared_parts > 0: that many sum records of rec_len reals (c_bar = sum of sums / sum of counts)
  int ck_shared_parts;
  int rec_len;         // K^2 + 1 rounded up to even: [sums over agents of c_k, number of agents, pad]
  // per-agent sum records (eea_batch_io::d_ck_rec): [B][rec_len] = [c_k, 1 (0 for a rejected agent), pad]; optional
  R* ck_rec;
  int rec_wave;        // eea_batch_io::rec_per_wavefront: the packed kernel writes one record per wavefront (sum of its agents')
  // device-bound exchange (eea_batch_io::d_rec_ready / d_ck_flag): rec_ready[b] = rec_seq once agent b's record is
  // visible device-wide (the record sum waits for the marks, not for the kernel); the consumer waits until
  // *ck_flag - ck_flag_seq >= 0 (mod 2^32) right before it reads ck_shared ("late binding": ~45 % into the wavefront's
  // lifetime) and reads it past its L1
  unsigned* rec_ready;
  unsigned rec_seq;
  const unsigned* ck_flag;
  unsigned ck_flag_seq;
  R* edx;
  R* bdx;
  R* rhot;
  int* status;
  const int* skip;    // [B] optional (eea_batch_io::d_skip): agents with a non-zero entry are left out of the launch
  long long* dbg;     // phase stamps of the A/B library's kernels (tools/ab/); null in the product
  // single-agent path: host-visible completion word, set to done_seq (system-scope release) after
  // u0 / status of agent 0 are written; null for batches
  int* done;
  int done_seq;
  // receding-horizon steps per launch (eea_control_batch_steps; 1 = eea_control_batch): step n reads pose row
  // n * pose_step_stride + b and writes u0 row n * u0_step_stride + b (strides in agents: 0 = the same row every step)
  int n_steps;
  unsigned pose_step_stride, u0_step_stride;
  // resident single-robot wavefront (control_wave_impl.hpp, RESIDENT instances): the host-mapped mailbox (ResidentMail<R>),
  // the request number it has seen already and the idle time (100 MHz ticks) after which it leaves; unused elsewhere
  void* res_mail;
  unsigned res_first;
  long long res_idle;
};

template <typename R>
size_t control_lds_bytes(int T, int K, int n_mem_max, int chunk);

// Launches the fused control kernel for B agents.  rollout_only: stop after the forward
// pass (optTraj), using ut as is (no shift).  Returns hipSuccess or the launch error.
template <typename R>
hipError_t launch_control(const ControlParams<R>& p, unsigned B, int model, int n_mem_max,
                          bool rollout_only, hipStream_t stream);

// The resident single-robot workgroup (control_kernel_impl.hpp control_resident_kernel; EEA_OPT_RESIDENT_CONTROL): one launch
// serves one robot's control() calls from a host-mapped mailbox until it has been idle for idle_ticks (100 MHz) or is told to
// leave.  The mailbox layout (host side: engine.cpp) is ResidentMailF64 / F32 below.
constexpr int kResidentMemCols = 128;  // capacity of the mapped replay-memory buffer (the shipped batch_size is 100)
template <typename R>
struct ResidentMail
{
  // host -> device: ONE 64-byte line, the request number written last -- the workgroup's poll fetches the whole line (16
  // lanes, one dword each: one PCIe read), so a new request number arrives together with its data
  alignas(64) R pose[3];
  R map_x, map_y;      // map_pos_ is refreshed on every call (ergodic_control.hpp:366-367)
  int n_mem;           // valid columns of the mapped replay-memory buffer
  int cmd;             // 1 = leave
  unsigned req;        // request number (only grows)
  // device -> host
  alignas(64) R u0[3];
  int status;
  int done;            // = the request's number once u0 / status are visible
  int alive;           // 0 once the workgroup has left
};
static_assert(offsetof(ResidentMail<double>, req) + sizeof(unsigned) <= 64 && offsetof(ResidentMail<double>, u0) == 64,
              "the request is one cache line");
// what the body reads instead of the mailbox: device memory, filled by the workgroup from the fetched line
template <typename R>
struct ResidentStage
{
  R pose[3];
  int n_mem;
};

template <typename R>
hipError_t launch_control_resident(const ControlParams<R>& p, int model, int n_mem_max, void* d_mail, void* d_stage,
                                   unsigned first_seen, long long idle_ticks, hipStream_t stream);

// Wavefront-per-agent control kernel (control_wave_impl.hpp): horizons of at most 256 steps, K <= 16 or K = 20.
// launch_control_wave needs control_wave_eligible.
template <typename R>
bool control_wave_eligible(const ControlParams<R>& p, bool rollout_only);
template <typename R>
hipError_t launch_control_wave(const ControlParams<R>& p, unsigned B, int model, bool rollout_only,
                               hipStream_t stream);
// The wavefront kernel as a RESIDENT single-robot server (EEA_OPT_RESIDENT_CONTROL at horizons of one slot: fp64, T <= 64,
// K <= 16): one wavefront that serves p.res_mail until told to leave or idle.  hipErrorInvalidValue: the shape has no instance
bool control_wave_resident_eligible(const ControlParams<double>& p);
hipError_t launch_control_wave_resident(const ControlParams<double>& p, int model, hipStream_t stream);
// Several agents per wavefront (control_pack_impl.hpp): groups of 8 / 16 / 32 lanes per agent, fp64, K = 5 / 10,
// T <= 4 lanes.  control_pack_lanes: the group size for a batch of B agents (0 = one wavefront per agent); forced = the
// value of EEA_OPT_AGENT_LANES
bool control_pack_eligible(const ControlParams<double>& p, int lanes);
int control_pack_lanes(const ControlParams<double>& p, unsigned B, int forced);
hipError_t launch_control_pack(const ControlParams<double>& p, unsigned B, int model, bool rollout_only, int lanes,
                               hipStream_t stream);
// sum of B per-agent records (ControlParams::ck_rec): one launch, a fixed tree (groups of kSumGroup agents in agent
// order, kSumFan group records per level-1 record, the level-1 records in order), finished by ticket inside the launch.
// d_ws: >= ck_sum_ws_elems(B, K2) reals, d_ctr: ck_sum_tickets(B, K2) tickets (zero before the first use; they reset
// themselves)
inline unsigned ck_sum_groups(unsigned B) { return (B + kSumGroup - 1) / kSumGroup; }
inline unsigned ck_sum_slices(int K2) { return static_cast<unsigned>(ck_record_len(K2) + 63) / 64u; }
inline size_t ck_sum_ws_elems(unsigned B, int K2)
{
  const size_t g0 = ck_sum_groups(B), g1 = (g0 + kSumFan - 1) / kSumFan;
  return (g0 + g1) * static_cast<size_t>(ck_record_len(K2));
}
inline size_t ck_sum_tickets(unsigned B, int K2)
{
  const size_t g1 = (ck_sum_groups(B) + kSumFan - 1) / kSumFan;
  return ck_sum_slices(K2) * (g1 + 1) + 1;  // per slice: level-1 tickets + the level-2 ticket; then the slices' ticket
}
// d_ready / seq / d_flag: the device-bound form (eea_ck_records_sum_bound): wait for rec_ready[b] == seq per agent inside
// the launch, publish *d_flag = seq behind the finished record; all null / 0: the plain form
template <typename R>
hipError_t launch_ck_records_sum(const R* d_rec, unsigned B, int K2, R* d_ws, unsigned* d_ctr, R* d_out, hipStream_t stream,
                                 const unsigned* d_ready = nullptr, unsigned seq = 0, unsigned* d_flag = nullptr);
// d_pub [n] = d_src [n] written through, then *d_flag = seq (one small launch behind an all-reduce)
template <typename R>
hipError_t launch_publish_record(const R* d_src, int n, R* d_pub, unsigned* d_flag, unsigned seq, hipStream_t stream);

// ---- phi_k path ----------------------------------------------------------------------
// Gaussians of a target passed to the fill kernel by value: [mean x, mean y (Fourier frame), cov_inv xx, yy]
constexpr int kMaxGaussArgs = 8;
constexpr int kFillPerThread = 16;  // grid points per thread of the fill kernel
template <typename R>
struct GaussArgs
{
  int n;
  R g[kMaxGaussArgs][4];
};
// both axis tables of a rebuild in one launch (cx: [K][nx], cy: [ny][K]); d_coord: the accumulated coordinates
template <typename R>
hipError_t launch_axis_tables(const R* d_coord, int nx, int ny, int K, R pi_lx, R pi_ly, R* d_cx, R* d_cy,
                              hipStream_t s);
// un-normalised Target::fill with the Gaussians in the kernel arguments; target_fill_blocks(P) partial sums
int target_fill_blocks(size_t P);
// (and, when d_cx != nullptr, the two axis tables in the same launch)
template <typename R>
hipError_t launch_target_fill_args(const R* d_coord, int nx, int ny, const GaussArgs<R>& ga, R* d_phi,
                                   R* d_partials, int K, R pi_lx, R pi_ly, R* d_cx, R* d_cy, hipStream_t s);
// configTarget for a sum of axis-aligned Gaussians in ONE launch of one workgroup: phi_k and the target's mass from the
// per-axis factors (phik_kernel.hip gaussian_phik_kernel); needs gaussian_phik_lds_bytes <= 160 KiB of LDS
size_t gaussian_phik_lds_bytes(int nx, int ny, int n_gauss, int K, size_t real_size);
template <typename R>
hipError_t launch_gaussian_phik(const R* d_coord, int nx, int ny, const GaussArgs<R>& ga, int K, R inv_lx, R inv_ly,
                                R* d_phik, R* d_mass, hipStream_t s, hipEvent_t stop = nullptr);
// spatialCoeff of an UN-normalised grid divided by its mass (the sum of d_mass_partials); d_mass[0] receives
// the mass.  stop (optional): an event bound to the completion of the last launch
template <typename R>
hipError_t launch_spatial_coeff_normalised(const R* d_phi_raw, int nx, int ny, int K, const R* d_cx, const R* d_cy,
                                           R* d_work, R* d_phik, const R* d_mass_partials, int n_mass, R* d_mass,
                                           hipStream_t s, hipEvent_t stop = nullptr);
// cos tables: out[k * n + i] = cos((k * pi_over_l) * coord[i]), k < K
template <typename R>
hipError_t launch_cos_tables(const R* d_coord, int n, int K, R pi_over_l, R* d_out, hipStream_t s);
// transposed layout: out[i * K + k]
template <typename R>
hipError_t launch_cos_tables_t(const R* d_coord, int n, int K, R pi_over_l, R* d_out, hipStream_t s);

// Target::fill without the normalisation: phi[iy*nx+ix] = sum_g exp(-0.5 d^T Sigma^-1 d).
// d_gauss: [n_gauss][4] = mean in the Fourier frame (mu - map_pos) and diag(cov_inv).
// Block partial sums of phi go to d_partials[*n_partials] (>= ceil(nx*ny/256) reals).
template <typename R>
hipError_t launch_target_fill(const R* d_xs, const R* d_ys, int nx, int ny, const R* d_gauss,
                              int n_gauss, R* d_phi, R* d_partials, int* n_partials, hipStream_t s);

// same on an arbitrary point list (d_px, d_py: P coordinates each)
template <typename R>
hipError_t launch_target_fill_points(const R* d_px, const R* d_py, unsigned P, const R* d_gauss,
                                     int n_gauss, R* d_phi, R* d_partials, int* n_partials,
                                     hipStream_t s);

// deterministic sum of n values -> d_out[0]
template <typename R>
hipError_t launch_reduce_sum(const R* d_in, int n, R* d_out, hipStream_t s);

// phi_vals *= 1 / d_sum[0]   (Target::fill's normalisation, target.cpp:87)
template <typename R>
hipError_t launch_scale_by_inv(R* d_phi, size_t n, const R* d_sum, hipStream_t s);

// spatialCoeff on a regular grid (separable form): two passes, atomics-free.
// d_cx: [K][nx], d_cy: [ny][K] (note the transposed layout of the y table),
// d_work: >= spatial_work_elems(nx, ny, K) reals, d_phik: [K^2] (col = k2*K + k1)
size_t spatial_work_elems(int nx, int ny, int K);
template <typename R>
hipError_t launch_spatial_coeff(const R* d_phi, int nx, int ny, int K, const R* d_cx, const R* d_cy,
                                R* d_work, R* d_phik, hipStream_t s);
// occupancy cells (int8) decoded through d_lut[256] inside the streaming kernel; d_raw receives the
// un-normalised sums, launch_normalise_by_first divides by element 0 -- or, with d_mass_out, the reduction launch divides
// by mode (0, 0)'s sum itself (the same bits, one launch less): d_raw = the normalised coefficients, *d_mass_out = the normaliser
template <typename R>
hipError_t launch_spatial_coeff_cells(const int8_t* d_occ, int nx, int ny, int K, const R* d_cx,
                                      const R* d_cy, const R* d_lut, R* d_work, R* d_raw, hipStream_t s,
                                      R* d_mass_out = nullptr);
template <typename R>
hipError_t launch_normalise_by_first(const R* d_raw, int K2, R* d_out, hipStream_t s);

// Weighted basis sum over an arbitrary point list (Basis::trajCoeff / spatialCoeff):
// out[m] = scale * sum_p w_p f_m(x_p, y_p); d_w may be null (w = 1).
// d_work: >= point_work_elems(P, K) reals.
size_t point_work_elems(unsigned P, int K);
template <typename R>
hipError_t launch_point_coeff(const R* d_x, const R* d_y, const R* d_w, unsigned P, int K, R pi_lx,
                              R pi_ly, R scale, R* d_work, R* d_out, hipStream_t s);

// ---- collision path (integer, bit-exact) ------------------------------------------------
struct CollisionParams
{
  double xmin, ymin, resolution;
  unsigned xsize, ysize;
  int r_bnd, r_col, r_max;
  double occupied_threshold;
};
hipError_t launch_collision_check(const CollisionParams& c, const int8_t* d_grid,
                                  const double* d_pose, unsigned P, int* d_hit, hipStream_t s);
// integrate_twist (numerics.hpp:273-297) per pose; wrap: the heading normalised to [-pi, pi) afterwards
hipError_t launch_integrate_twist(const double* d_x0, const double* d_u, double dt, unsigned P, double* d_out, bool wrap, hipStream_t s);
hipError_t launch_validate_control(const CollisionParams& c, const int8_t* d_grid,
                                   const double* d_x0, const double* d_u, double dt, unsigned steps,
                                   unsigned P, int* d_valid, hipStream_t s);

struct DwaParams
{
  double dt, acc_dt, acc_lim[3], vmax[3], vmin[3];
  unsigned ns[3];  // vx, vy, vth samples (>= 1)
  unsigned steps;  // (unsigned)|horizon / dt|
};
hipError_t launch_dwa_control(const CollisionParams& c, const DwaParams& d, const int8_t* d_grid,
                              const double* d_x0, const double* d_vb, const double* d_vref,
                              const double* d_xt_ref, unsigned n_ref, double dt_ref, unsigned P,
                              double* d_u_opt, int* d_found, hipStream_t s);
// eea_tick_batch: step 1 (follow counters, skip mask) and steps 3 + 4 (validate_control; the dynamic window where it failed,
// per robot towards its own twist or along its optTraj, and the state update) of the fleet tick
hipError_t launch_tick_begin(int* d_follow, unsigned* d_count, int* d_skip, unsigned dwa_steps, unsigned P, hipStream_t s);
hipError_t launch_validate_and_dwa_fleet(const CollisionParams& c, const DwaParams& d, const int8_t* d_grid,
                                         unsigned long long grid_epoch, const double* d_x0, const double* d_vb,
                                         const double* d_traj, unsigned n_ref, double dt_ref, double val_dt, unsigned val_steps,
                                         int* d_valid, int* d_follow, unsigned* d_count, double* d_u, int* d_source, unsigned P,
                                         hipStream_t s);
// frees the cached ring offsets and inflated-map buffers of every device
void release_collision_caches();

// ======================================================================================
// device helpers
// ======================================================================================
#if defined(__HIPCC__)

template <typename R>
__device__ __forceinline__ void sincos_r(R a, R* s, R* c);
template <>
__device__ __forceinline__ void sincos_r<double>(double a, double* s, double* c)
{
  sincos(a, s, c);
}
template <>
__device__ __forceinline__ void sincos_r<float>(float a, float* s, float* c)
{
  sincosf(a, s, c);
}

// sin(pi t), cos(pi t): exact argument reduction, no large-argument path
template <typename R>
__device__ __forceinline__ void sincospi_r(R t, R* s, R* c);
// fp64: n = rint(2t), r = t - n/2 in [-1/4, 1/4] (exact), Taylor series of sin(pi r), cos(pi r)
// in r^2 (8 and 9 terms: truncation < 5e-17), then the quadrant swap / sign from n mod 4.
// Max abs error 1.9e-16 against long-double references on 4.2e6 arguments (tools/ubench/sincos_check);
// about half the instructions of the device library's sincospi.
//  - the rounding adds 1.5 * 2^52: the sum's low word is n mod 2^32 (no conversion) for |t| < 2^50;
//    larger arguments (never produced by headings or in-domain positions) are first reduced
//    modulo 2 with v_fract (exact)
//  - Horner steps are written as v_fma_f64 with the coefficient in a scalar register pair: the
//    compiler's own choice (v_fmac_f64 on a copy of a coefficient held in vector registers) costs
//    a second instruction per step and 32 vector registers per kernel
__device__ __forceinline__ double fma_sc(double a, double b, double coeff)
{
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(coeff));
  return d;
}
__device__ __forceinline__ double mul_sc(double a, double coeff)
{
  double d;
  asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(coeff));
  return d;
}
__device__ __forceinline__ double add_sc(double a, double coeff)
{
  double d;
  asm("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(coeff));
  return d;
}
template <>
__device__ __forceinline__ void sincospi_r<double>(double t, double* s, double* c)
{
  if (__builtin_expect(!(fabs(t) < 0x1p50), 0)) t = 2.0 * __builtin_amdgcn_fract(0.5 * t);
  // every constant comes from a scalar register pair (one per instruction is what the encoding allows): the
  // compiler's own lowering copies 64-bit literals into vector registers first (6 v_mov per evaluation)
  const double magic = 0x1.8p52;
  double nb;
  asm("v_fma_f64 %0, %1, 2.0, %2" : "=v"(nb) : "v"(t), "s"(magic));
  const int q = __double2loint(nb);
  const double n = add_sc(nb, -magic);
  const double r = fma(n, -0.5, t);
  const double z = r * r;
  // top Horner step as multiply + add (two scalar-register constants cannot share one instruction); its terms
  // are below 1e-5 of the result, so the extra rounding is invisible
  double ps = add_sc(mul_sc(z, -0x1.6fadb9f155744p-16), 0x1.e8f434d018d63p-12);
  ps = fma_sc(ps, z, -0x1.e3074fde8871fp-8);
  ps = fma_sc(ps, z, 0x1.50783487ee782p-4);
  ps = fma_sc(ps, z, -0x1.32d2cce62bd86p-1);
  ps = fma_sc(ps, z, 0x1.466bc6775aae2p+1);
  ps = fma_sc(ps, z, -0x1.4abbce625be53p+2);
  ps = fma_sc(ps, z, 0x1.921fb54442d18p+1);
  const double sv = ps * r;
  double pc = add_sc(mul_sc(z, 4.3030695870329473e-06), -0.0001046381049248457);
  pc = fma_sc(pc, z, 0.0019295743094039231);
  pc = fma_sc(pc, z, -0.025806891390014061);
  pc = fma_sc(pc, z, 0.23533063035889321);
  pc = fma_sc(pc, z, -1.3352627688545895);
  pc = fma_sc(pc, z, 4.0587121264167685);
  pc = fma_sc(pc, z, -4.934802200544679);
  pc = fma(pc, z, 1.0);
  const bool odd = (q & 1) != 0;
  const double s1 = odd ? pc : sv;
  const double c1 = odd ? sv : pc;
  // sin changes sign in quadrants 2,3; cos in quadrants 1,2
  const int ssign = (q & 2) << 30;
  const int csign = ((q + 1) & 2) << 30;
  *s = __hiloint2double(__double2hiint(s1) ^ ssign, __double2loint(s1));
  *c = __hiloint2double(__double2hiint(c1) ^ csign, __double2loint(c1));
}
// sin(pi r), cos(pi r) for |r| <= 1/16 (increments of a heading or of a basis angle over one step): no argument
// reduction, Taylor series in r^2 with 6 + 7 terms (truncation < 1e-19), 16 instructions instead of ~40
__device__ __forceinline__ void sincospi_small(double r, double* s, double* c)
{
  const double z = r * r;
  double ps = add_sc(mul_sc(z, -0.007370430945714351), 0.08214588661112823);
  ps = fma_sc(ps, z, -0.5992645293207921);
  ps = fma_sc(ps, z, 2.550164039877345);
  ps = fma_sc(ps, z, -5.16771278004997);
  ps = fma_sc(ps, z, 3.141592653589793);
  *s = ps * r;
  double pc = add_sc(mul_sc(z, 0.0019295743094039231), -0.02580689139001406);
  pc = fma_sc(pc, z, 0.2353306303588935);
  pc = fma_sc(pc, z, -1.3352627688545893);
  pc = fma_sc(pc, z, 4.058712126416768);
  pc = fma_sc(pc, z, -4.934802200544679);
  *c = fma(pc, z, 1.0);
}
__device__ __forceinline__ void sincospi_small(float r, float* s, float* c) { sincospif(r, s, c); }

template <>
__device__ __forceinline__ void sincospi_r<float>(float t, float* s, float* c)
{
  sincospif(t, s, c);
}

// numerics.hpp:78-90 of the reference: wrap to [-pi, pi)
template <typename R>
__device__ __forceinline__ R wrap_pi(R rad)
{
  const R pi = static_cast<R>(kPi);
  const R q = floor((rad + pi) / (R(2) * pi));
  rad = (rad + pi) - q * R(2) * pi;
  if (rad < R(0)) rad += R(2) * pi;
  return rad - pi;
}

// consensus c_k of mode m (eea_batch_io::d_ck_shared): the K^2 values themselves, or -- ck_shared_parts > 0 -- the
// sum of that many sum records divided by the sum of their agent counts (element K^2 of a record).  A record set that
// no agent contributed to (count 0: every agent of the producing pass was rejected, or the buffer is still zero) is no
// consensus at all: the agent keeps its own c_k (`own`) -- the reference's behaviour -- instead of 0 / 0
// agent-scope (sc1: write-through / L1-bypassing) accesses for data that crosses XCDs inside one launch
template <typename R>
__device__ __forceinline__ void store_agent(R* q, R v)
{
  __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename R>
__device__ __forceinline__ R load_agent(const R* q)
{
  return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The divisor is the same for every mode: the reciprocal of the summed agent counts is formed ONCE per wavefront (wavefront-
// uniform, kept in scalar registers) and the modes are multiplied by it -- the quotient per mode was nine fp64 divisions per lane
// of the wavefront kernels, 4 - 7 % of a consensus pass (profiles/r06_ablation.txt item 13).
template <typename R>
struct SharedCk
{
  R inv;      // 1 / (sum of the records' agent counts); unused with ck_shared_parts = 0
  bool have;  // some agent contributed
};
__device__ __forceinline__ double uniform_value(double v)
{
  union { double d; unsigned u[2]; } x;
  x.d = v;
  x.u[0] = __builtin_amdgcn_readfirstlane(x.u[0]);
  x.u[1] = __builtin_amdgcn_readfirstlane(x.u[1]);
  return x.d;
}
__device__ __forceinline__ float uniform_value(float v)
{
  union { float f; unsigned u; } x;
  x.f = v;
  x.u = __builtin_amdgcn_readfirstlane(x.u);
  return x.f;
}
template <typename R, typename P>  // P: ControlParams<R>, possibly in the kernel-argument address space
__device__ __forceinline__ SharedCk<R> shared_ck_begin(const P& p, const R* shared, int K2)
{
  SharedCk<R> sc{ R(1), true };
  if (p.ck_shared_parts <= 0) return sc;
  // a buffer another kernel fills WHILE this one runs (device-bound exchange) is read past the L1
  const bool bound = p.ck_flag != nullptr;
  R n = R(0);
  for (int i = 0; i < p.ck_shared_parts; ++i) {
    const R* const rec = shared + static_cast<size_t>(i) * p.rec_len;
    n += bound ? load_agent(rec + K2) : rec[K2];
  }
  n = uniform_value(n);
  sc.have = n > R(0);
  sc.inv = sc.have ? uniform_value(R(1) / n) : R(0);
  return sc;
}
template <typename R, typename P>
__device__ __forceinline__ R shared_ck_value(const P& p, const SharedCk<R>& sc, const R* shared, int m, R own)
{
  const bool bound = p.ck_flag != nullptr;
  if (p.ck_shared_parts <= 0) return bound ? load_agent(shared + m) : shared[m];
  R s = R(0);
  for (int i = 0; i < p.ck_shared_parts; ++i) {
    const R* const rec = shared + static_cast<size_t>(i) * p.rec_len;
    s += bound ? load_agent(rec + m) : rec[m];
  }
  return sc.have ? s * sc.inv : own;
}

// Device-bound exchange, consumer side: wait (bounded) until *flag has reached seq.  One lane polls past the L1 with a
// sleep between polls (MI355X_MICROARCH.md "polling-cost"); ~2 us per poll, 400 000 polls: about a second, then the
// caller reports EEA_ERR_TIMEOUT and goes on with the agent's own c_k.  Wavefront-uniform result.  The bound is a safety
// net against a producer that can never become resident, not a pacing device (every wait of the protocol is for work
// enqueued before the waiter), and generous: a loaded box must not turn a long queue into a time-out.
constexpr int kFlagPolls = 400000;
__device__ __forceinline__ bool wait_flag(const unsigned* flag, unsigned seq)
{
  for (int i = 0; i < kFlagPolls; ++i) {
    const unsigned v = __builtin_amdgcn_readfirstlane(load_agent(flag));
    if (static_cast<int>(v - seq) >= 0) return true;
    __builtin_amdgcn_s_sleep(32);
  }
  return false;
}

// std::clamp semantics (NaN passes through), ergodic_control.hpp:447-449
template <typename R>
__device__ __forceinline__ R clamp_std(R v, R lo, R hi)
{
  return (v < lo) ? lo : ((hi < v) ? hi : v);
}

template <typename R>
__device__ __forceinline__ R wave_inclusive_scan(R v)
{
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const R t = __shfl_up(v, o, kWave);
    if (lane >= o) v += t;
  }
  return v;
}

// ---- matrix cores: D = A(16x4) B(4x16) + C, one operand element per lane -----------------------
// A[i][k]: lane = 16 k + i;  B[k][j]: lane = 16 k + j.  C/D: col = lane % 16 and
// row = lane/16 + 4 r (f64) or 4 (lane/16) + r (f32), r = accumulator register.
template <typename R>
struct Mfma;
template <>
struct Mfma<double>
{
  using acc_t = double __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c)
  {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float>
{
  using acc_t = float __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
  {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// ---- DPP scans: row shifts inside 16-lane rows, row broadcasts across rows; no LDS traffic
// full row mask: lanes without a source read 0 through bound_ctrl (no "old" register to clear);
// partial row mask (row broadcasts): the unselected rows keep old = 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_or_zero(double v)
{
  constexpr bool kBound = ROW_MASK == 0xf;
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, kBound);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, kBound);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_or_zero(float v)
{
  constexpr bool kBound = ROW_MASK == 0xf;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, kBound));
}

// wave64 inclusive sum scan: row_shr 1,2,4,8 then row_bcast15 (rows 1,3) and row_bcast31
// (rows 2,3).  Lanes without a source keep the identity 0.
template <typename R>
__device__ __forceinline__ R wave_inclusive_scan_dpp(R v)
{
  v += dpp_or_zero<0x111, 0xf>(v);
  v += dpp_or_zero<0x112, 0xf>(v);
  v += dpp_or_zero<0x114, 0xf>(v);
  v += dpp_or_zero<0x118, 0xf>(v);
  v += dpp_or_zero<0x142, 0xa>(v);
  v += dpp_or_zero<0x143, 0xc>(v);
  return v;
}

// Inclusive scan over the workgroup (kBlock threads).  s_w: kBlock/kWave + 1 reals of LDS
// scratch.  Returns the inclusive prefix; *total receives the workgroup sum.
// Contains two __syncthreads().
template <typename R>
__device__ __forceinline__ R block_inclusive_scan(R v, R* s_w, R* total)
{
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const R s = wave_inclusive_scan(v);
  if (lane == kWave - 1) s_w[wave] = s;
  __syncthreads();
  R off = R(0), tot = R(0);
#pragma unroll
  for (int w = 0; w < kBlock / kWave; ++w) {
    const R ws = s_w[w];
    if (w < wave) off += ws;
    tot += ws;
  }
  __syncthreads();
  *total = tot;
  return s + off;
}

#endif  // __HIPCC__
}  // namespace eea
