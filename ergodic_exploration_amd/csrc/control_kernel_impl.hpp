// Fused receding-horizon ergodic control kernel for gfx950 (MI355X), version 2.
//
// One workgroup (256 threads = 4 wavefronts) per agent performs one complete
// `ErgodicControl<ModelT>::control` call (reference ergodic_control.hpp:224-311, minus
// configTarget) for ModelT in {Omni, SimpleCart}:
//
//   shift controls (:233-234) -> forward RK4 rollout (integrator.hpp:135-152,176-184)
//   -> c_k (basis.cpp:109-120) -> ergodic-metric gradient (:418-436) + barrier (:453-474)
//   -> backward co-state RK4 (integrator.hpp:154-174,186-194; rhodot :65-69)
//   -> control update + clamp (:438-451).
//
// It is not a translation of the reference's loops.  Three identities of the reference's
// own formulas make the horizon parallel (derivations in DESIGN.md section 2):
//   1. The basis is separable, f_k(x,y) = cos(a_k1 x) cos(b_k2 y) (basis.cpp:85), and
//      cos(k a), sin(k a) follow from one sincos by the angle-addition recurrence.
//   2. For Omni/SimpleCart theta' = w does not depend on the state, so RK4 collapses to
//      Simpson's rule in closed form and the rollout is two prefix sums over the horizon.
//   3. fdx has only A(0,2), A(1,2) non-zero and edx(2) = bdx(2) = 0, so the co-state pass is
//      two chained suffix sums (RK4 == trapezoid there).
//
// CDNA4 mapping (what changed against control_kernel_v1.hip):
//   * horizon scans: DPP row shifts / row broadcasts inside each 64-lane wavefront (no LDS
//     traffic), one LDS hop between the 4 wavefronts;
//   * c_k = (1/N) Cx Cy^T is the one GEMM-shaped piece: each wavefront stages the cos tables of
//     16 of its points in a private LDS tile laid out [point][k] so that the MFMA operand of
//     lane l is the contiguous element (4g + l/16) * KS + l%16, and accumulates with
//     v_mfma_f64_16x16x4_f64 (v_mfma_f32_16x16x4_f32 for fp32); no workgroup barrier inside;
//   * sin(pi t) / cos(pi t) evaluation (exact argument reduction) for headings and basis angles;
//   * LDS regions with disjoint lifetimes are aliased: 32 KB per agent at K=10, T=200 (fp64),
//     4 workgroups per CU.

#include "common.hpp"

// Phase markers EEA_STAMP(n): nothing in the product; tools/ab/control_kernel_timing.hip (A/B library) defines them
// as shader-clock stamps into p.dbg before including this file.

// waves per SIMD the K <= 12 instances are compiled for (register budget 512 / waves)
#ifndef EEA_WAVES_SMALL_K
#define EEA_WAVES_SMALL_K 5
#endif

#ifndef EEA_STAMP
#define EEA_STAMP(n) ((void)0)
#endif

namespace eea
{
namespace
{
template <typename R>
__device__ __forceinline__ void sc_pi(R t, R* s, R* c)
{
  sincospi_r(t, s, c);
}

// acc + k * v for a small mode number k: fp64 takes the constant from a scalar register pair
template <typename R>
__device__ __forceinline__ R fma_mode(R v, int k, R acc)
{
  return acc + static_cast<R>(k) * v;
}
template <>
__device__ __forceinline__ double fma_mode<double>(double v, int k, double acc)
{
  double d;
  asm("v_fma_f64 %0, %1, %3, %2" : "=v"(d) : "v"(v), "v"(acc), "s"(static_cast<double>(k)));
  return d;
}

// points staged per wavefront per MFMA pass: 16; 8 for the K = 30 instance, whose tiles would
// otherwise hold the kernel at one workgroup per CU (K = 30, T = 500: 83 KB -> 67 KB of LDS).
// (K = 30 always runs the exact-K instance, so host-side sizing and kernel agree.)
#ifndef EEA_STAGE8_K10
#define EEA_STAGE8_K10 0
#endif
__host__ __device__ constexpr int stage_points(int K) { return (K == 30 || (EEA_STAGE8_K10 && K == 10)) ? 8 : 16; }

__host__ __device__ inline int up4(int n) { return (n + 3) & ~3; }
__host__ __device__ inline int table_stride(int K) { return (K + 1) & ~1; }  // even: 16-byte rows
// operand reads of the last staged point run past its row into the pad (modes >= K, ignored):
// at most 16 - KS elements for one 16x16 tile, 32 - KS for two
__host__ __device__ inline int wave_tab_elems(int K)
{
  return up4(2 * stage_points(K) * table_stride(K) + (K <= 16 ? 16 : 32));
}

// LDS carve (element offsets; every segment a multiple of 4 elements => 16-byte aligned)
struct LdsLayout
{
  int vx, vy, w;           // shifted controls, SoA [T]
  int ct, st;              // cos/sin of the pre-step heading, [T+1] (index T = final heading)
  int c1x, s1x, c1y, s1y;  // sin/cos(pi x / lx), sin/cos(pi y / ly) per point, [Nmax]
                           // (memory points first, rollout last: buffer.cpp:78-108)
  int g0, g1;              // barrier gradient rows 0,1 carried from the forward to the backward half, [T]
  int D;                   // lambda_k * (c_k - phi_k), [K^2]
  int sw;                  // scan scratch (one slot set per scan) + flags
  int E;                   // per-wavefront MFMA tiles; afterwards each wavefront's c_k partials [K^2]
  int total;
};

// waves: wavefronts per agent (workgroup size / 64)
__host__ __device__ inline LdsLayout lds_layout(int T, int Nmax, int K, int waves)
{
  LdsLayout L;
  int o = 0;
  L.vx = o; o += up4(T);
  L.vy = o; o += up4(T);
  L.w = o; o += up4(T);
  L.ct = o; o += up4(T + 1);
  L.st = o; o += up4(T + 1);
  L.c1x = o; o += up4(Nmax);
  L.s1x = o; o += up4(Nmax);
  L.c1y = o; o += up4(Nmax);
  L.s1y = o; o += up4(Nmax);
  L.g0 = o; o += up4(T);
  L.g1 = o; o += up4(T);
  L.D = o; o += up4(K * K);
  L.sw = o; o += 48;
  L.E = o;
  {
    // the tiles, later the wavefronts' c_k partials: one K^2 buffer per wavefront inside its own tile
    // region when that fits, else two shared buffers (control_agent's "folded" reduction)
    const int tiles = waves * wave_tab_elems(K);
    const int red = (K * K <= wave_tab_elems(K)) ? 0 : (waves < 2 ? waves : 2) * K * K;
    o += up4(tiles > red ? tiles : red);
  }
  L.total = o;
  return L;
}

// ---- workgroup scans on top of the DPP wavefront scan -------------------------------------
// `reuse`: the scratch slots will be written again before another barrier (multi-chunk
// horizons); otherwise every scan owns its slots and the trailing barrier is not needed
template <typename R, int WAVES>
__device__ __forceinline__ R block_scan(R v, R* s_w, R& total, bool reuse)
{
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // provably wave-uniform
  const R s = wave_inclusive_scan_dpp(v);
  if (lane == kWave - 1) s_w[wave] = s;
  __syncthreads();
  R off = R(0), tot = R(0);
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    const R ws = s_w[w];
    if (w < wave) off += ws;
    tot += ws;
  }
  if (reuse) __syncthreads();
  total = tot;
  return s + off;
}

template <typename R, int WAVES>
__device__ __forceinline__ void block_scan2(R& a, R& b, R* s_w, R& tot_a, R& tot_b, bool reuse)
{
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const R sa = wave_inclusive_scan_dpp(a);
  const R sb = wave_inclusive_scan_dpp(b);
  if (lane == kWave - 1) {
    s_w[wave] = sa;
    s_w[WAVES + wave] = sb;
  }
  __syncthreads();
  R oa = R(0), ob = R(0), ta = R(0), tb = R(0);
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    const R wa = s_w[w], wb = s_w[WAVES + w];
    if (w < wave) {
      oa += wa;
      ob += wb;
    }
    ta += wa;
    tb += wb;
  }
  if (reuse) __syncthreads();
  a = sa + oa;
  b = sb + ob;
  tot_a = ta;
  tot_b = tb;
}

// orders a wavefront's own LDS writes before its own LDS reads (same-wave DS ops execute in
// order; this only stops the compiler from moving them across)
__device__ __forceinline__ void wave_lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// model right-hand side x/y components at heading (c, s) (omni.hpp:177-181, cart.hpp:172)
template <typename R, int MODEL>
__device__ __forceinline__ void model_xy(R vx, R vy, R c, R s, R& fx, R& fy)
{
  if (MODEL == kModelOmni) {
    fx = vx * c - vy * s;
    fy = vx * s + vy * c;
  } else {
    fx = vx * c;
    fy = vx * s;
  }
}

// Register budget: waves per SIMD the kernel is compiled for (VGPR + AGPR <= 512 / waves).
// K <= 12 fits 5 waves per SIMD = 5 workgroups per CU, matching the LDS footprint (29.6 KB at
// T = 200, K = 10, fp64); larger K needs the wider budget.
constexpr int min_waves_per_simd(int KC) { return (KC > 20) ? 2 : ((KC > 12) ? 3 : EEA_WAVES_SMALL_K); }

// wrap to [-pi, pi) like normalize_angle_PI (numerics.hpp:78-90) with the quotient taken by a
// multiplication; a quotient off by one at an exact multiple is repaired by the two range fixes
template <typename R>
__device__ __forceinline__ R wrap_pi_fast(R rad)
{
  const R pi = static_cast<R>(kPi), two_pi = static_cast<R>(2.0 * kPi);
  const R q = floor((rad + pi) * static_cast<R>(1.0 / (2.0 * kPi)));
  rad = (rad + pi) - q * two_pi;
  if (rad < R(0)) rad += two_pi;
  if (rad >= two_pi) rad -= two_pi;
  return rad - pi;
}

// BLK: threads per agent (64, 128 or 256): short horizons take fewer wavefronts per agent.
// The body of one agent's control() call as a device function: control_kernel (one launch per call) and
// control_resident_kernel (a workgroup that stays and serves one robot's calls from a host mailbox) share it.
template <typename R, int MODEL, int KC, int BLK>
__device__ __forceinline__ void control_agent(const ControlParams<R>& p, const int Nmax, const int rollout_only, const int b)
{
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  R* const sm = reinterpret_cast<R*>(smem_raw);

  if (p.skip != nullptr && p.skip[b] != 0) return;  // eea_batch_io::d_skip: the agent is left out (workgroup-uniform)
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int T = p.T;
  const int K = (KC > 0) ? KC : p.K;
  const int K2 = K * K;
  const int KS = table_stride(K);
  constexpr int WAVES = BLK / kWave;
  const LdsLayout L = lds_layout(T, Nmax, K, WAVES);

  R* const s_vx = sm + L.vx;
  R* const s_vy = sm + L.vy;
  R* const s_w = sm + L.w;
  R* const s_ct = sm + L.ct;
  R* const s_st = sm + L.st;
  R* const s_c1x = sm + L.c1x;
  R* const s_s1x = sm + L.s1x;
  R* const s_c1y = sm + L.c1y;
  R* const s_s1y = sm + L.s1y;
  R* const s_g0 = sm + L.g0;
  R* const s_g1 = sm + L.g1;
  R* const s_D = sm + L.D;
  R* const s_sw = sm + L.sw;
  R* const s_E = sm + L.E;
  int* const s_bad = reinterpret_cast<int*>(s_sw + 40);  // one flag per wavefront
  const bool multi_chunk = T > BLK;

  EEA_STAMP(0);
  int nmem = 0;
  if (p.mem_cols != nullptr && !rollout_only) {
    nmem = (p.n_mem != nullptr) ? p.n_mem[b] : static_cast<int>(p.mem_stride);
    nmem = nmem < 0 ? 0 : (nmem > static_cast<int>(p.mem_stride) ? static_cast<int>(p.mem_stride) : nmem);
  }
  const int N = T + nmem;

  const R* const pose = p.pose + 3 * static_cast<size_t>(b);
  R* const ut = p.ut + 3 * static_cast<size_t>(T) * b;
  const R x0 = pose[0], y0 = pose[1], th0 = pose[2];

  // lambda_k, phi_k of this thread's mode: issued now, consumed after the c_k reduction
  R lam_m = R(0), phi_m = R(0);
  if (!rollout_only && tid < K2) {
    lam_m = p.lamdak[tid];
    phi_m = p.phik[tid];
  }

  // ---- controls: shift left by one column, last column zero (ergodic_control.hpp:233-234)
  R w_own = R(0);  // this thread's yaw rate of the first chunk: the heading scan starts from registers
  {
    bool bad = false;
    for (int i = tid; i < T; i += BLK) {
      const int src = rollout_only ? i : i + 1;
      R vx = R(0), vy = R(0), w = R(0);
      if (src < T) {
        vx = ut[3 * src + 0];
        vy = ut[3 * src + 1];
        w = ut[3 * src + 2];
      }
      s_vx[i] = vx;
      s_vy[i] = vy;
      s_w[i] = w;
      if (i == tid) w_own = w;
      // SimpleCart::operator() rejects a lateral velocity (cart.hpp:167-170)
      if (MODEL == kModelSimpleCart && !(fabs(vy) < R(1.0e-12))) bad = true;
    }
    // every wavefront publishes its own flag: no initialisation pass; the flags and the controls
    // become visible with the barrier inside the first heading scan (nothing is written before it)
    const bool wave_bad = __any(bad);
    if (lane == 0) s_bad[wave] = wave_bad ? 1 : 0;
  }
  if (WAVES == 1) __syncthreads();  // single wavefront: a fence, no s_barrier
  EEA_STAMP(1);

  const R dt = p.dt;
  const R dt6 = p.dt6;
  const R inv_pi = static_cast<R>(1.0 / kPi);
  R* const traj = (p.traj != nullptr) ? p.traj + 3 * static_cast<size_t>(T) * b : nullptr;

  // ================= forward half: one pass per chunk of 256 steps ============================
  // heading: theta_i = wrap(theta_{i-1} + dt/6 (w + 2w + 2w + w)) (integrator.hpp:146-148,183)
  //          == wrap(theta_0 + prefix sum) up to rounding
  // position: x_i = x_{i-1} + dt/6 (k1 + 2 k2 + 2 k3 + k4) with k2 == k3 (integrator.hpp:176-184)
  // basis sin/cos of this thread's own rollout point (last chunk): with no memory columns and a
  // single chunk the contraction takes them from registers and needs no barrier in front of it
  R own_c1x = R(0), own_s1x = R(0), own_c1y = R(0), own_s1y = R(0);
  {
    R carry_th = wrap_pi_fast(th0), carry_x = x0, carry_y = y0;
    for (int base = 0; base < T; base += BLK) {
      const int i = base + tid;
      const bool act = i < T;
      R d = R(0), w = R(0);
      if (act) {
        w = (base == 0) ? w_own : s_w[i];
        d = dt6 * (((w + R(2) * w) + R(2) * w) + w);
      }
      R tot_th;
      const R inc = block_scan<R, WAVES>(d, s_sw, tot_th, multi_chunk);
      if (base == 0) {
        int any_bad = 0;
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv) any_bad |= s_bad[wv];
        if (any_bad) {
          // the reference throws out of rk4_.solve; nothing else of this agent is touched
          if (tid == 0 && p.status != nullptr) p.status[b] = 2;  // EEA_ERR_INVALID_TWIST
          if (p.ck_rec != nullptr && !rollout_only) {  // an all-zero sum record: the agent does not count
            for (int m = tid; m < p.rec_len; m += BLK) store_agent(p.ck_rec + static_cast<size_t>(b) * p.rec_len + m, R(0));
            if (p.rec_ready != nullptr) {  // device-bound exchange: the ready mark behind the drained record
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __syncthreads();
              if (tid == 0) store_agent(p.rec_ready + b, p.rec_seq);
            }
          }
          if (tid == 0 && p.done != nullptr && b == 0) {
            __hip_atomic_store(p.done, p.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
          return;
        }
        // (device-bound exchange: a time-out an earlier pass left in a reused buffer stays until the caller clears it)
        if (tid == 0 && p.status != nullptr && !(p.ck_flag != nullptr && p.status[b] == 6)) p.status[b] = 0;
      }
      EEA_STAMP(2);

      // pre-step heading (own prefix minus own increment) and mid stage theta + dt (0.5 w), shared by
      // k2 and k3 (integrator.hpp:179-180), by sin/cos evaluation; the post-step heading is
      // 2 mid - pre, so its sin/cos follow from the two by the double-angle and addition formulas
      // (a few flops, ~4e-16) -- no exchange with the next thread, no barrier in the forward half
      // between the two scans.  The table entries (pre-step values, read by the backward half as
      // "heading after step i - 1") stay the directly evaluated ones.
      R dx = R(0), dy = R(0);
      if (act) {
        // sin(pi t), cos(pi t) reduce any argument exactly: only the reported heading is wrapped
        const R th_pre = carry_th + (inc - d);
        const R th_post = carry_th + inc;
        if (traj != nullptr) traj[3 * i + 2] = wrap_pi_fast(th_post);
        R s, c, smid, cm;
        sc_pi(th_pre * inv_pi, &s, &c);
        s_ct[i] = c;
        s_st[i] = s;
        sc_pi((th_pre + dt * (R(0.5) * w)) * inv_pi, &smid, &cm);
        const R c2m = R(1) - R(2) * smid * smid, s2m = R(2) * smid * cm;
        const R cpost = c2m * c + s2m * s, spost = s2m * c - c2m * s;
        // the heading after the chunk's / horizon's last step has no later thread to tabulate it
        if (tid == BLK - 1 || i == T - 1) {
          s_ct[i + 1] = cpost;
          s_st[i + 1] = spost;
        }
        const R vx = s_vx[i], vy = s_vy[i];
        R k1x, k1y, k2x, k2y, k4x, k4y;
        model_xy<R, MODEL>(vx, vy, c, s, k1x, k1y);
        model_xy<R, MODEL>(vx, vy, cm, smid, k2x, k2y);
        model_xy<R, MODEL>(vx, vy, cpost, spost, k4x, k4y);
        dx = dt6 * (((k1x + R(2) * k2x) + R(2) * k2x) + k4x);
        dy = dt6 * (((k1y + R(2) * k2y) + R(2) * k2y) + k4y);
      }
      EEA_STAMP(3);
      R tx, ty;
      block_scan2<R, WAVES>(dx, dy, s_sw + 4, tx, ty, multi_chunk);
      if (act) {
        const R X = carry_x + dx, Y = carry_y + dy;
        if (traj != nullptr) {
          traj[3 * i + 0] = X;
          traj[3 * i + 1] = Y;
        }
        if (!rollout_only) {
          // map frame -> Fourier frame (ergodic_control.hpp:243-244)
          const R x = X - p.map_x, y = Y - p.map_y;
          // one sin/cos pair per axis: angle = pi x / lx (basis.cpp:85 with k = 1)
          R s, c;
          sc_pi(x * p.inv_lx, &s, &c);
          s_c1x[nmem + i] = c;
          s_s1x[nmem + i] = s;
          own_c1x = c;
          own_s1x = s;
          sc_pi(y * p.inv_ly, &s, &c);
          s_c1y[nmem + i] = c;
          s_s1y[nmem + i] = s;
          own_c1y = c;
          own_s1y = s;
          // barrier gradient (ergodic_control.hpp:453-474), carried to the backward half
          const R eps = R(0.05), weight = R(25);
          R b0 = R(0), b1 = R(0);
          b0 += R(2) * static_cast<R>(x > p.lx - eps) * (x - (p.lx - eps));
          b1 += R(2) * static_cast<R>(y > p.ly - eps) * (y - (p.ly - eps));
          b0 += R(2) * static_cast<R>(x < eps) * (x - eps);
          b1 += R(2) * static_cast<R>(y < eps) * (y - eps);
          s_g0[i] = b0 * weight;
          s_g1[i] = b1 * weight;
        }
      }
      carry_th += tot_th;
      carry_x += tx;
      carry_y += ty;
      EEA_STAMP(4);
    }
  }
  if (rollout_only) return;

  // sampled past states are prepended (buffer.cpp:78-108) and shifted like the rollout
  if (nmem > 0) {
    const R* const mem = p.mem_cols + 3 * static_cast<size_t>(p.mem_stride) * b;
    for (int j = tid; j < nmem; j += BLK) {
      R s, c;
      sc_pi((mem[3 * j + 0] - p.map_x) * p.inv_lx, &s, &c);
      s_c1x[j] = c;
      s_s1x[j] = s;
      sc_pi((mem[3 * j + 1] - p.map_y) * p.inv_ly, &s, &c);
      s_c1y[j] = c;
      s_s1y[j] = s;
    }
  }
  // point q of the contraction is this thread's own rollout point when nothing is prepended and the
  // horizon is one chunk; the other threads' points are first read after the reduction's barrier
  const bool own_points = (nmem == 0) && !multi_chunk && KC != 30;  // K = 30 stages cooperatively (below): LDS
  if (!own_points) __syncthreads();
  EEA_STAMP(5);

  // ---- c_k = (1/N) sum_p cos(a_k1 x_p) cos(b_k2 y_p)  (basis.cpp:109-120) on the matrix cores
  {
    using M = Mfma<R>;
    using acc_t = typename M::acc_t;
    constexpr int NT = (KC > 0) ? ((KC + 15) / 16) : 2;  // 16x16 tiles per dimension
    const int nt = (K + 15) / 16;
    acc_t acc[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int c = 0; c < NT; ++c) acc[a][c] = acc_t{ R(0), R(0), R(0), R(0) };

    R* const tabx = s_E + wave * wave_tab_elems(K);
    constexpr int kSub = stage_points(KC);
    R* const taby = tabx + kSub * KS;
    const int sub = lane / kSub;  // which pass stages this lane's point
    const int pl = lane % kSub;   // its row in the staged tile
    const int mk = lane >> 4, mi = lane & 15;  // matrix-instruction operand coordinates of this lane

    for (int c0 = 0; c0 < N; c0 += BLK) {
      const int q = c0 + wave * kWave + lane;
      int nvalid = N - (c0 + wave * kWave);
      nvalid = nvalid < 0 ? 0 : (nvalid > kWave ? kWave : nvalid);
      const bool have = q < N;
      R c1 = R(0), s1 = R(0), d1 = R(0), e1 = R(0);
      if (have) {
        if (own_points) {  // wave-uniform
          c1 = own_c1x;
          s1 = own_s1x;
          d1 = own_c1y;
          e1 = own_s1y;
        } else {
          c1 = s_c1x[q];
          s1 = s_s1x[q];
          d1 = s_c1y[q];
          e1 = s_s1y[q];
        }
      }

      // K = 30 (the BASELINE config 5 shape) stages a pass with ALL 64 lanes: see stage30 below
      constexpr bool kCoop = (KC == 30);
      constexpr int KA = (KC > 0 && !kCoop) ? KC : 1;
      R cxr[KA], cyr[KA];
      if (KC > 0 && !kCoop) {
        // cos(k a) for k < K by the angle-addition recurrence; zero rows for padding points
        R ck = have ? R(1) : R(0), sk = R(0), dk = have ? R(1) : R(0), ek = R(0);
#pragma unroll
        for (int k = 0; k < KA; ++k) {
          cxr[k] = ck;
          cyr[k] = dk;
          const R cn = ck * c1 - sk * s1;
          sk = sk * c1 + ck * s1;
          ck = cn;
          const R dn = dk * d1 - ek * e1;
          ek = ek * d1 + dk * e1;
          dk = dn;
        }
      }

      // stage one pass of 16 points (this lane's point if it belongs to pass s)
      auto stage = [&](int s) {
        if (sub == s) {
          R* const tx = tabx + pl * KS;
          R* const ty = taby + pl * KS;
          if (KC > 0) {
#pragma unroll
            for (int k = 0; k < KA; ++k) {
              tx[k] = cxr[k];
              ty[k] = cyr[k];
            }
          } else {
            R ck = have ? R(1) : R(0), sk = R(0), dk = have ? R(1) : R(0), ek = R(0);
            for (int k = 0; k < K; ++k) {
              tx[k] = ck;
              ty[k] = dk;
              const R cn = ck * c1 - sk * s1;
              sk = sk * c1 + ck * s1;
              ck = cn;
              const R dn = dk * d1 - ek * e1;
              ek = ek * d1 + dk * e1;
              dk = dn;
            }
          }
        }
      };
      // K = 30: a pass is 8 points x 2 axes x 30 modes = 480 table entries.  Written by the 8 lanes that own the
      // points they are 60 LDS stores of 8 active lanes each -- and an LDS write costs its 6 / 13 cycles per
      // instruction whatever the mask (profiles/r02_ubench_coissue.txt): 47k LDS cycles per CU in this phase, its
      // bottleneck.  Instead all 64 lanes stage: lane = (point l % 8, axis (l / 8) % 2, mode group l / 16): it reads
      // the point's cos / sin of that axis from LDS, reaches cos / sin of 8, 16, 24 times the angle by doubling,
      // takes its group's first two modes by one rotation and the other six by the Chebyshev recurrence, and
      // writes its 8 (last group: 6) consecutive entries: 4 full 16-byte stores per pass.
      auto stage30 = [&](int s) {
        const int pt = lane & 7, ax = (lane >> 3) & 1, mg = lane >> 4;
        const int qq = c0 + wave * kWave + s * kSub + pt;
        const bool hv = qq < N;
        const int qc = hv ? qq : 0;
        const R c = ax ? s_c1y[qc] : s_c1x[qc];
        const R sn = ax ? s_s1y[qc] : s_s1x[qc];
        const R c2 = (c + c) * c - R(1), s2 = (sn + sn) * c;
        const R c4 = (c2 + c2) * c2 - R(1), s4 = (s2 + s2) * c2;
        const R c8 = (c4 + c4) * c4 - R(1), s8 = (s4 + s4) * c4;
        const R c16 = (c8 + c8) * c8 - R(1), s16 = (s8 + s8) * c8;
        const R c24 = c16 * c8 - s16 * s8, s24 = s16 * c8 + c16 * s8;
        R cs = (mg == 0) ? R(1) : (mg == 1 ? c8 : (mg == 2 ? c16 : c24));
        R ss = (mg == 0) ? R(0) : (mg == 1 ? s8 : (mg == 2 ? s16 : s24));
        R a = hv ? cs : R(0);                         // T_{8 mg}; zero rows for padding points
        R bq = hv ? (cs * c - ss * sn) : R(0);        // T_{8 mg + 1}
        const R two = c + c;
        R* const dst = (ax ? taby : tabx) + pt * KS + 8 * mg;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < 3 || mg < 3) {  // modes 30, 31 do not exist (row stride 30)
            if (sizeof(R) == 8) {
              *reinterpret_cast<double2*>(dst + 2 * i) = double2{ static_cast<double>(a), static_cast<double>(bq) };
            } else {
              *reinterpret_cast<float2*>(dst + 2 * i) = float2{ static_cast<float>(a), static_cast<float>(bq) };
            }
          }
          const R n0 = two * bq - a, n1 = two * n0 - bq;
          a = n0;
          bq = n1;
        }
      };
      // four points per matrix instruction: operand of lane l = element (4g + l/16) * KS + l%16
      auto mma_group = [&](int g) {
        const int off = (4 * g + mk) * KS + mi;
        R av[NT], bv[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          av[t] = (t < nt) ? tabx[off + 16 * t] : R(0);
          bv[t] = (t < nt) ? taby[off + 16 * t] : R(0);
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int c = 0; c < NT; ++c)
            if (a < nt && c < nt) acc[a][c] = M::run(av[a], bv[c], acc[a][c]);
      };

      if (nvalid == kWave) {
        // full wavefront (the common case): no bounds tests, everything unrolled
#pragma unroll
        for (int s = 0; s < kWave / kSub; ++s) {
          if (kCoop) stage30(s);
          else stage(s);
          wave_lds_fence();
#pragma unroll
          for (int g = 0; g < kSub / 4; ++g) mma_group(g);
          wave_lds_fence();
        }
      } else {
        for (int s = 0; s < kWave / kSub; ++s) {
          if (s * kSub >= nvalid) break;  // wave-uniform
          if (kCoop) stage30(s);
          else stage(s);
          wave_lds_fence();
#pragma unroll
          for (int g = 0; g < kSub / 4; ++g) {
            if (s * kSub + 4 * g < nvalid) mma_group(g);  // wave-uniform
          }
          wave_lds_fence();
        }
      }
    }

    // cross-wavefront reduction: red[wave][mode], mode = k2*K + k1 (basis.cpp:58-66)
    EEA_STAMP(6);
    // each wavefront's partial sums go into its own tile region (K^2 <= wave_tab_elems(K)), in
    // program order after its last operand read: no barrier between the tiles and the reduction
    const bool own_region = K2 <= wave_tab_elems(K);
    const int red_stride = own_region ? wave_tab_elems(K) : K2;
    const int red_bufs = own_region ? WAVES : (WAVES < 2 ? WAVES : 2);
    R* const s_red = s_E;
    auto put_partials = [&](int buf, bool add) {
      const int j = lane & 15;
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k1 = 16 * a + M::row(lane, r);
            const int k2 = 16 * c + j;
            if (a < nt && c < nt && k1 < K && k2 < K) {
              R* const dst = s_red + buf * red_stride + k2 * K + k1;
              *dst = add ? *dst + acc[a][c][r] : acc[a][c][r];
            }
          }
    };
    if (own_region) {
      put_partials(wave, false);
    } else {
      // large bases: K^2 partials do not fit a wavefront's tile region.  Two shared buffers, filled
      // by wavefronts 0 and 1 and added to by 2 and 3 (same lane <-> same entries, fixed order)
      __syncthreads();  // every wavefront is done reading its tiles
      if (wave < 2) put_partials(wave, false);
      if (WAVES > 2) {
        __syncthreads();
        if (wave >= 2) put_partials(wave - 2, true);
      }
    }
    __syncthreads();
    const R invN = R(1) / static_cast<R>(N);
    for (int m = tid; m < K2; m += BLK) {
      R s = R(0);
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        if (w < red_bufs) s += s_red[w * red_stride + m];
      }
      R c = invN * s;
      if (p.ck != nullptr) p.ck[static_cast<size_t>(b) * K2 + m] = c;
      if (p.ck_rec != nullptr) store_agent(p.ck_rec + static_cast<size_t>(b) * p.rec_len + m, c);  // eea_batch_io::d_ck_rec
      s_D[m] = c;  // (same thread reads it back below)
    }
    if (p.ck_rec != nullptr) {  // [c_k, 1 (this agent counts), pad]
      for (int m = K2 + tid; m < p.rec_len; m += BLK) {
        store_agent(p.ck_rec + static_cast<size_t>(b) * p.rec_len + m, (m == K2) ? R(1) : R(0));
      }
      if (p.rec_ready != nullptr) {  // device-bound exchange: every thread's part has left, then the agent's ready mark
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) store_agent(p.rec_ready + b, p.rec_seq);
      }
    }
    bool use_shared = p.ck_shared != nullptr;
    if (use_shared && p.ck_flag != nullptr) {
      // ... consumer side: wavefront 0 waits for the shared c_k's flag and hands its outcome to the others through LDS -- one
      // polling stream per agent, and ONE outcome: near the poll bound a second poller could see the flag the first gave
      // up on, and the agent would mix its own and the shared c_k
      if (wave == 0) {
        bool ok = wait_flag(p.ck_flag, p.ck_flag_seq);
        if (ok && p.ck_shared_parts > 0) ok = !(load_agent(p.ck_shared + K2) < R(0));
        if (!ok && tid == 0 && p.status != nullptr) p.status[b] = 6;  // EEA_ERR_TIMEOUT
        if (tid == 0) s_sw[46] = ok ? R(1) : R(0);  // (a scratch slot no scan uses)
      }
      __syncthreads();
      use_shared = s_sw[46] != R(0);
      __syncthreads();
    }
    SharedCk<R> shared_ck{ R(1), true };
    if (use_shared) shared_ck = shared_ck_begin<R>(p, p.ck_shared, K2);  // workgroup-uniform
    for (int m = tid; m < K2; m += BLK) {
      R c = s_D[m];
      // decentralised consensus (eea_batch_io::d_ck_shared): the agents' shared c_k replaces the own one
      if (use_shared) c = shared_ck_value(p, shared_ck, p.ck_shared, m, c);
      // fourier_diff = lamdak % (ck - phik)  (ergodic_control.hpp:422)
      const R lam = (m == tid) ? lam_m : p.lamdak[m];
      const R phi = (m == tid) ? phi_m : p.phik[m];
      s_D[m] = lam * (c - phi);
    }
    __syncthreads();
    EEA_STAMP(7);
  }

  // ================= backward half: one pass per chunk, last chunk first ======================
  // per step: ergodic-metric gradient (:418-436, basis.cpp:91-107)
  //   edx_x = w sum_k2 cos(b y) [ sum_k1 D(k1,k2) (-a_k1 sin(a_k1 x)) ]
  //   edx_y = w sum_k2 (-b_k2 sin(b y)) [ sum_k1 D(k1,k2) cos(a_k1 x) ]
  // co-state rows 0,1: rho_i = rho_{i+1} + dt g_i (suffix sums); row 2:
  //   rho2_i = rho2_{i+1} + dt (S_i(rho_{i+1}) + dt/2 S_i(g_i)), S_i(v) = A(0,2) v0 + A(1,2) v1,
  //   A = fdx(x_i, u_i) (omni.hpp:194-197, cart.hpp:183-186); then u_i = clamp(-Rinv B^T rho_i)
  {
    R c0 = R(0), c1r = R(0), c2 = R(0);  // rhoT_ = 0 (ergodic_control.hpp:203)
    for (int base = 0; base < T; base += BLK) {
      const int i = T - 1 - (base + tid);
      const bool act = i >= 0;
      R g0 = R(0), g1 = R(0);
      if (act) {
        const int q = nmem + i;
        const R c1 = s_c1x[q], s1 = s_s1x[q];
        const R d1 = s_c1y[q], e1 = s_s1y[q];
        R Ex = R(0), Ey = R(0);
        if (KC > 0) {
          // edx_x = -pi/lx sum_k1 k1 sin(a_k1 x) G(k1),  G(k1) = sum_k2 D(k1,k2) cos(b_k2 y)
          // edx_y = -pi/ly sum_k2 k2 sin(b_k2 y) H(k2),  H(k2) = sum_k1 D(k1,k2) cos(a_k1 x)
          // one pass over D: K accumulators G (independent chains) and one H per row; the mode
          // numbers multiply through scalar-register constants (no per-mode factors in registers)
          constexpr int KA = KC > 0 ? KC : 1;
          R cxa[KA], G[KA];
          {
            R ck = R(1), sk = R(0);
#pragma unroll
            for (int k = 0; k < KA; ++k) {
              cxa[k] = ck;
              G[k] = R(0);
              const R cn = ck * c1 - sk * s1;
              sk = sk * c1 + ck * s1;
              ck = cn;
            }
          }
          R dk = R(1), ek = R(0), accy = R(0);
          constexpr int kRowUnroll = (KA <= 12 && sizeof(R) == 8) ? KA : 1;
#pragma unroll kRowUnroll
          for (int k2 = 0; k2 < KA; ++k2) {
            const R* const Drow = s_D + k2 * KA;
            R ha = R(0), hb = R(0);
#pragma unroll
            for (int k1 = 0; k1 + 1 < KA; k1 += 2) {
              const R da = Drow[k1], db = Drow[k1 + 1];
              G[k1] += da * dk;
              ha += da * cxa[k1];
              G[k1 + 1] += db * dk;
              hb += db * cxa[k1 + 1];
            }
            if (KA & 1) {
              const R da = Drow[KA - 1];
              G[KA - 1] += da * dk;
              ha += da * cxa[KA - 1];
            }
            if (k2 > 0) accy = fma_mode(ek * (ha + hb), k2, accy);
            const R dn = dk * d1 - ek * e1;
            ek = ek * d1 + dk * e1;
            dk = dn;
          }
          R accx = R(0);
          {
            R ck = c1, sk = s1;  // k1 = 1
#pragma unroll
            for (int k = 1; k < KA; ++k) {
              accx = fma_mode(sk * G[k], k, accx);
              const R cn = ck * c1 - sk * s1;
              sk = sk * c1 + ck * s1;
              ck = cn;
            }
          }
          Ex = -p.pi_lx * accx;
          Ey = -p.pi_ly * accy;
        } else {
          R dk = R(1), ek = R(0);
          for (int k2 = 0; k2 < K; ++k2) {
            const R* const Drow = s_D + k2 * K;
            R t1 = R(0), t2 = R(0);
            R ck = R(1), sk = R(0);
            for (int k1 = 0; k1 < K; ++k1) {
              const R dd = Drow[k1];
              t1 += dd * (-(static_cast<R>(k1) * p.pi_lx) * sk);
              t2 += dd * ck;
              const R cn = ck * c1 - sk * s1;
              sk = sk * c1 + ck * s1;
              ck = cn;
            }
            Ex += dk * t1;
            Ey += (-(static_cast<R>(k2) * p.pi_ly) * ek) * t2;
            const R dn = dk * d1 - ek * e1;
            ek = ek * d1 + dk * e1;
            dk = dn;
          }
        }
        const R ex = Ex * p.expl_weight, ey = Ey * p.expl_weight;
        const R b0 = s_g0[i], b1 = s_g1[i];
        if (p.edx != nullptr) {
          R* const o = p.edx + 3 * (static_cast<size_t>(T) * b + i);
          o[0] = ex;
          o[1] = ey;
          o[2] = R(0);
        }
        if (p.bdx != nullptr) {
          R* const o = p.bdx + 3 * (static_cast<size_t>(T) * b + i);
          o[0] = b0;
          o[1] = b1;
          o[2] = R(0);
        }
        g0 = ex + b0;
        g1 = ey + b1;
      }
      EEA_STAMP(8);

      R h0 = dt * g0, h1 = dt * g1;
      R t0, t1;
      block_scan2<R, WAVES>(h0, h1, s_sw + 12, t0, t1, multi_chunk);
      const R rho0 = c0 + h0, rho1 = c1r + h1;  // inclusive suffix: rho after step i
      EEA_STAMP(9);

      R qv = R(0), cth = R(0), sth = R(0);
      if (act) {
        cth = s_ct[i + 1];
        sth = s_st[i + 1];
        const R vx = s_vx[i], vy = s_vy[i];
        R a02, a12;
        if (MODEL == kModelOmni) {
          a02 = -vx * sth - vy * cth;
          a12 = vx * cth - vy * sth;
        } else {
          a02 = -vx * sth;
          a12 = vx * cth;
        }
        // rho_{i+1} = rho_i - dt g_i (rows 0,1): the exclusive suffix from this thread's own values
        const R sE = a02 * (rho0 - dt * g0) + a12 * (rho1 - dt * g1);
        const R sG = a02 * g0 + a12 * g1;
        qv = dt * (sE + p.half_dt * sG);
      }
      R tot2;
      const R inc2 = block_scan<R, WAVES>(qv, s_sw + 20, tot2, multi_chunk);
      const R rho2 = c2 + inc2;
      EEA_STAMP(10);

      // ---- u_i = clamp(-Rinv B(x_i)^T rho_i)  (ergodic_control.hpp:438-451)
      if (act) {
        R v0, v1, v2;
        if (MODEL == kModelOmni) {  // omni.hpp:205-212
          v0 = cth * rho0 + sth * rho1;
          v1 = -sth * rho0 + cth * rho1;
          v2 = rho2;
        } else {  // cart.hpp:194-203
          v0 = cth * rho0 + sth * rho1;
          // B^T rho has an exact zero in row 1; opaque to the compiler so that the three
          // Rinv(r,1) * (-0) products are formed here instead of living in registers (or
          // scratch) across the whole backward half
          R zero = R(0);
          asm volatile("" : "+v"(zero));
          v1 = zero;
          v2 = rho2;
        }
        const R n0 = -v0, n1 = -v1, n2 = -v2;
        R u[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const R ur = (p.Rinv[r] * n0 + p.Rinv[r + 3] * n1) + p.Rinv[r + 6] * n2;
          u[r] = clamp_std(ur, p.umin[r], p.umax[r]);
        }
        ut[3 * i + 0] = u[0];
        ut[3 * i + 1] = u[1];
        ut[3 * i + 2] = u[2];
        if (p.rhot != nullptr) {
          R* const o = p.rhot + 3 * (static_cast<size_t>(T) * b + i);
          o[0] = rho0;
          o[1] = rho1;
          o[2] = rho2;
        }
        if (i == 0) {
          R* const o = p.u0 + 3 * static_cast<size_t>(b);
          o[0] = u[0];
          o[1] = u[1];
          o[2] = u[2];
          if (p.done != nullptr && b == 0) {
            // the host polls this word instead of waiting for the kernel's completion signal.  The
            // sequence number stays in its scalar register until here (hoisted into a vector register
            // it would live -- spilled -- across the whole backward half)
            int seq = p.done_seq;
            asm volatile("" : "+s"(seq));
            __hip_atomic_store(p.done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
      }
      c0 += t0;
      c1r += t1;
      c2 += tot2;
    }
  }
  EEA_STAMP(11);
}

template <typename R, int MODEL, int KC, int BLK>
__global__ __launch_bounds__(BLK, min_waves_per_simd(KC)) void control_kernel(
    const ControlParams<R> p, const int Nmax, const int rollout_only)
{
  control_agent<R, MODEL, KC, BLK>(p, Nmax, rollout_only, static_cast<int>(blockIdx.x));
}

// ---- one robot, one control() per tick, WITHOUT a launch per call (round 5) ------------------------------------------------
// The reference's real use is one robot at 10 Hz (exploration.hpp:232); a launch per call costs the host -> device -> host
// round trip of a dispatch (18-22 us whatever the shape; the reference's smallest shipped shape takes 6 us on one CPU thread).
// This workgroup STAYS: it polls a host-mapped mailbox (pose + request number in one cache line, written by eea_control),
// runs the same control_agent body, and answers through the mailbox (u0, status, done = request number: system-scope
// release, as the one-launch path).  Bounded in every direction: it leaves by itself after `idle_ticks` of the constant 100
// MHz clock without a request (alive = 0 tells the host, which launches it again on the next call), on an exit command, and
// the body has no unbounded wait.  Opt-in: EEA_OPT_RESIDENT_CONTROL.
template <typename R, int MODEL, int KC, int BLK>
__global__ __launch_bounds__(BLK, min_waves_per_simd(KC)) void control_resident_kernel(
    const ControlParams<R> p0, const int Nmax, ResidentMail<R>* const mail, ResidentStage<R>* const stage,
    const unsigned first_seen, const long long idle_ticks)
{
  __shared__ unsigned s_line[16];
  __shared__ unsigned s_leaving;  // alive = 0 has been announced: the request being served (if any) is the last one
  if (threadIdx.x == 0) s_leaving = 0u;
  constexpr int kReq = offsetof(ResidentMail<R>, req) / 4, kCmd = offsetof(ResidentMail<R>, cmd) / 4;
  constexpr int kNmem = offsetof(ResidentMail<R>, n_mem) / 4, kMapX = offsetof(ResidentMail<R>, map_x) / 4;
  unsigned last = first_seen;
  for (;;) {
    if (threadIdx.x < kWave) {  // wavefront 0 polls: lanes 0..15 fetch the request line, one dword each
      const unsigned* const line = reinterpret_cast<const unsigned*>(mail);
      const int l = threadIdx.x & 15;
      const long long t0 = wall_clock64();
      unsigned v, r;
      bool idle = false;
      for (;;) {
        v = __hip_atomic_load(line + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        r = __builtin_amdgcn_readlane(v, kReq);
        if (r != last) break;
        if (wall_clock64() - t0 > idle_ticks) {
          idle = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (idle) {
        // idle for too long: say so FIRST, then look once more -- a request posted before the host can have seen alive = 0 is
        // still served, one posted later finds alive = 0 and launches a new workgroup (eea_control)
        if (threadIdx.x == 0) {
          __hip_atomic_store(&mail->alive, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          s_leaving = 1u;
        }
        __builtin_amdgcn_s_sleep(64);
        v = __hip_atomic_load(line + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (threadIdx.x < 16) s_line[threadIdx.x] = v;
    }
    __syncthreads();
    const unsigned r = s_line[kReq];
    if (r == last) return;  // (alive is 0)
    last = r;
    if (s_line[kCmd] != 0u) {
      if (threadIdx.x == 0) {
        __hip_atomic_store(&mail->alive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mail->done, static_cast<int>(r), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    // the body reads the pose and the column count from DEVICE memory (one more trip over PCIe per field otherwise)
    ControlParams<R> p = p0;
    {
      constexpr int kPoseDwords = 3 * sizeof(R) / 4;
      unsigned* const st = reinterpret_cast<unsigned*>(stage);
      if (threadIdx.x < kPoseDwords) st[threadIdx.x] = s_line[threadIdx.x];
      if (threadIdx.x == kPoseDwords) st[offsetof(ResidentStage<R>, n_mem) / 4] = s_line[kNmem];
      R mx, my;
      if (sizeof(R) == 8) {
        mx = static_cast<R>(__hiloint2double(static_cast<int>(s_line[kMapX + 1]), static_cast<int>(s_line[kMapX])));
        my = static_cast<R>(__hiloint2double(static_cast<int>(s_line[kMapX + 3]), static_cast<int>(s_line[kMapX + 2])));
      } else {
        mx = static_cast<R>(__int_as_float(static_cast<int>(s_line[kMapX])));
        my = static_cast<R>(__int_as_float(static_cast<int>(s_line[kMapX + 1])));
      }
      p.map_x = mx;
      p.map_y = my;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // whatever other kernels (or the host: eea_set_ut) wrote since the last request, and the stage just written, are behind
    // this workgroup's L1
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    p.pose = stage->pose;
    p.n_mem = &stage->n_mem;
    // the answer is announced HERE, behind every store of the request (the one-launch path announces it as soon as u0 is out:
    // there the stream orders whatever follows behind the kernel; here nothing does)
    p.done = nullptr;
    control_agent<R, MODEL, KC, BLK>(p, Nmax, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&mail->done, static_cast<int>(r), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // a request that landed between "alive = 0" and the last look has been served; the workgroup has told the host that it
    // is gone, so it leaves now (back in the poll loop it would race the successor the host launches)
    if (s_leaving != 0u) return;  // (written before the barrier that published s_line: workgroup-uniform)
  }
}

template <typename R, int MODEL, int KC, int BLK>
hipError_t launch_resident_one(const ControlParams<R>& p, int Nmax, size_t lds, void* mail, void* stage, unsigned first_seen,
                               long long idle_ticks, hipStream_t stream)
{
  auto kern = control_resident_kernel<R, MODEL, KC, BLK>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(1), dim3(BLK), lds, stream, p, Nmax, static_cast<ResidentMail<R>*>(mail),
                     static_cast<ResidentStage<R>*>(stage), first_seen, idle_ticks);
  return hipGetLastError();
}

template <typename R, int MODEL, int KC, int BLK>
hipError_t launch_one(const ControlParams<R>& p, unsigned B, int Nmax, bool rollout_only,
                      size_t lds, hipStream_t stream)
{
  auto kern = control_kernel<R, MODEL, KC, BLK>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize,
                                             static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(BLK), lds, stream, p, Nmax, rollout_only ? 1 : 0);
  return hipGetLastError();
}

}  // namespace
}  // namespace eea
