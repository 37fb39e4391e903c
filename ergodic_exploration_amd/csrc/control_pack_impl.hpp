// Fused receding-horizon ergodic control kernel for gfx950 (MI355X): SEVERAL AGENTS PER WAVEFRONT (short horizons).
//
// The wavefront-per-agent kernel (control_wave_impl.hpp) maps lane <-> horizon step: at T = 20 (BASELINE configs[1])
// 44 of its 64 lanes idle, at T = 5 (configs[0]) 59, and a vector instruction costs the same whatever the lane mask.
// Here an agent is a GROUP OF L = 8, 16 or 32 LANES and a wavefront carries A = 64 / L agents; a lane owns up to
// S = ceil(T / L) <= 4 consecutive horizon steps of its agent (T <= 4 L).  Same computation, same identities -- one complete
// `ErgodicControl<ModelT>::control` call (reference ergodic_control.hpp:224-311 minus configTarget) per agent:
//
//   * the six horizon scans are SEGMENTED DPP scans: row_shr 1/2/4/8 already stop at the 16-lane rows, the 8-lane groups
//     mask the steps that would cross, the 32-lane groups add row_bcast15; a group's total comes back over ds_bpermute.
//     Nothing an agent computes depends on a value of another agent (a NaN agent cannot reach its neighbours);
//   * c_k on the matrix cores with BLOCK <-> AGENT: v_mfma_f64_4x4x4_4b multiplies four independent 4x4x4 blocks; the
//     four blocks of an instruction are four different agents (L = 8), two agents x two point quads (L = 16) or four point
//     quads of one agent (L = 32).  The 32-row tile of a half-pass is staged exactly as in the wavefront kernel (one axis
//     per lane, the partner's cosine over v_permlane32_swap, even tile rows read by lanes 0..31 and odd rows by lanes
//     32..63: conflict-free ds_read_b64), only the point -> tile row map differs; one accumulator set per half;
//   * D = lambda (c - phi) per agent in LDS, agents 2 K^2 dwords apart (K = 10: 8 banks: the A row reads of a gradient
//     step -- every lane of an agent the same address -- are conflict-free);
//   * top-heavy horizons (T = 3 L + 1 .. 3 L + L / 8: the yaml's T = 50 on 16 lanes): the first r lanes of an agent own four
//     steps, the others three, and the gradient of the r-step tail slot is taken by all lanes together, 8 lanes per step
//     (control_wave_kernel's cooperative tail with the 8-lane groups inside the agents): 0.292 -> 0.314 at T = 50;
//   * compiled for 3 wavefronts per SIMD (168 registers): the barrier gradient stays in registers, the LDS holds the
//     tiles / the A agents' D and the parked headings (<= 10.8 KB per wavefront, 12 wavefronts per CU).
//
// What it does not take (the engine falls back to the wavefront kernel): the device-bound exchange (sum records, flags),
// fp32, bases other than K = 5 / 10, the single-agent mailbox.  Which L a batch gets: engine.cpp (pack_lanes).
#pragma once

#include "control_wave_impl.hpp"

namespace eea
{
namespace pack
{
using wave::kMaxS;
using wave::kStageRows;
using wave::lds_fence;
using wave::tab_stride;

__host__ __device__ constexpr int d_stride(int K) { return (K * K + 3) & ~3; }  // K = 10: 100 (200 dwords = 8 mod 64), K = 5: 28
constexpr int kTailElems = 24;  // hand-over space of the cooperative tail gradient, behind the agents' D: [3][8] reals
// Between the two 16-row groups of a table: 8 reals (16 dwords).  A 16-byte staging store is served in groups of 8 contiguous
// lanes on 32 banks of 4 bytes; with 8 or 16 lanes per agent those 8 lanes are rows {0..3} of BOTH 16-row groups of one block,
// and 16 rows of 20 (K = 10) or 12 (K = 5) dwords are a multiple of 32 banks apart: every staging store was a two-way
// conflict (SQ_LDS_BANK_CONFLICT 1.3 cycles per LDS instruction against the wavefront kernel's 0.13, profiles/r06_pack_pmc.txt).
// Shifted by 16 banks the second group's four windows tile the banks the first group's leave free.
#ifndef EEA_PACK_GROUP_PAD
#define EEA_PACK_GROUP_PAD 8
#endif
constexpr int kGroupPad = EEA_PACK_GROUP_PAD;
__host__ __device__ constexpr int region_elems(int K, int A)
{
  const int t = 2 * (kStageRows * tab_stride(K) + kGroupPad), d = A * d_stride(K) + kTailElems;
  return ((t > d ? t : d) + 3) & ~3;
}
// the parked post-step headings: cos and sin, [S][64] each.  S is a launch argument: a horizon of <= 3 steps per lane leaves
// LDS for more resident wavefronts.  Never less than what the controls' hand-over between the steps of a multi-step launch
// takes: [3][4 L A = 256] reals from the start of the region
__host__ __device__ constexpr int park_elems(int S) { return 2 * S * kWave; }
__host__ __device__ constexpr int wave_lds_elems(int K, int A, int S)
{
  const int e = region_elems(K, A) + park_elems(S);
  return e > 3 * kMaxS * kWave ? e : 3 * kMaxS * kWave;
}
// wavefronts per SIMD the kernel is compiled for.  K = 5: four (118 registers).  K = 10, round 6: the timed 16-lane instances
// (no stage outputs; yaml's T = 50) take <= 128 registers without scratch -- FOUR per SIMD -- since
//  (a) lambda_k / phi_k are read where D is formed instead of being preloaded for all NB^2 entries: that preload, on top of the
//      accumulators and the per-step state, was the kernel's register peak (146 of 162; found with a liveness profile of the
//      ISA -- not in the contraction, where round 5 looked for it): 162 -> 138 for every group size;
//  (b) with 16 lanes per agent the four blocks of a matrix instruction are the wavefront's FOUR AGENTS (kOneSet below): one
//      accumulator set instead of one per half of the wavefront (-18), half the D section, and half-passes without a point
//      are skipped altogether.
// 8 lanes per agent (eight agents: two sets by necessity) and 32 stay at three per SIMD.  EEA_PACK_LEAN = 1 (experiment, same
// box: 4 wavefronts per SIMD at 126 registers but 13 % more time per wavefront -- not kept) single-buffers the matrix operands.
// LDS: 10 KB per wavefront at S = 4 = 16 wavefronts per CU exactly.
#ifndef EEA_PACK_WAVES_K10
#define EEA_PACK_WAVES_K10 4
#endif
#ifndef EEA_PACK_LEAN
#define EEA_PACK_LEAN 0
#endif
constexpr int waves_per_simd(int KC, bool STAGES, int L, int SM)
{
  // K = 10, four per SIMD: 16 lanes per agent (one accumulator set), and 8 lanes per agent at <= 3 steps per lane (SM = 3:
  // configs[1]'s T = 20 -- a quarter less per-step state instead)
  return KC == 5 ? 4 : ((!STAGES && (L == 16 || (L == 8 && SM == 3) || EEA_PACK_LEAN != 0)) ? EEA_PACK_WAVES_K10 : 3);
}

// row_shr:N inside the 16-lane row, lanes without a source read 0
template <int N>
__device__ __forceinline__ double row_shr(double v)
{
  return dpp_or_zero<0x110 + N, 0xf>(v);
}
// inclusive sum scan over the L lanes of each agent (tl = lane % L)
template <int L>
__device__ __forceinline__ double seg_inclusive_scan(double v, int tl)
{
  if constexpr (L == 8) {
    // the shifted value of a lane whose source lies in the other agent of the row is dropped (select, not multiply: the
    // neighbour may hold anything)
    double t = row_shr<1>(v);
    v += (tl >= 1) ? t : 0.0;
    t = row_shr<2>(v);
    v += (tl >= 2) ? t : 0.0;
    t = row_shr<4>(v);
    v += (tl >= 4) ? t : 0.0;
  } else {
    v += row_shr<1>(v);
    v += row_shr<2>(v);
    v += row_shr<4>(v);
    v += row_shr<8>(v);
    if constexpr (L == 32) v += dpp_or_zero<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3
  }
  return v;
}
// the value of the agent's last lane in every lane of the agent
template <int L>
__device__ __forceinline__ double seg_last(double v, int lane)
{
  const int src = (lane | (L - 1)) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Point of lane r (0..31) of a half-pass -> row of the 32-row tile.  An instruction group gq (16 rows) multiplies four
// blocks bb of four points k; its operand reads take row 16 gq + 4 bb + 2 (k & 1) + (k >> 1) (control_wave_impl.hpp: even
// rows in lanes 0..31, odd rows in lanes 32..63).  Which (gq, bb, k) a point gets decides what a block sums:
//   L = 8 : bb = agent of the half (4 agents x 8 points: gq = point quad)
//   L = 16: bb = agent + 2 * (quad & 1), gq = quad >> 1   (the two quads of an agent in a group are blocks bb and bb ^ 2)
//   L = 32: bb = quad & 3, gq = quad >> 2                 (one agent, its four quads of a group are the four blocks)
template <int L>
__device__ __forceinline__ int tile_row(int r)
{
  int gq, bb;
  const int k = r & 3;
  if constexpr (L == 8) {
    gq = (r >> 2) & 1;
    bb = r >> 3;
  } else if constexpr (L == 16) {
    const int q = (r >> 2) & 3;
    gq = q >> 1;
    bb = (r >> 4) + 2 * (q & 1);
  } else {
    const int q = r >> 2;
    gq = q >> 2;
    bb = q & 3;
  }
  return 16 * gq + 4 * bb + 2 * (k & 1) + (k >> 1);
}
// agent (index inside the wavefront) whose points block bb of half h sums
template <int L>
__device__ __forceinline__ int block_agent(int h, int bb)
{
  if constexpr (L == 8) return 4 * h + bb;
  else if constexpr (L == 16) return 2 * h + (bb & 1);
  else return h;
}

// MODEL, KC (5 or 10), STAGES as in control_wave_kernel; L = lanes per agent; WPB = wavefronts per workgroup; SM = the most
// steps a lane owns in this instance (3 or 4 = wave::kMaxS): the per-step state is SM-element register arrays, and a horizon
// of <= 3 L steps (configs[1]: T = 20 on 8 lanes) does not have to carry a fourth, empty element of each through the kernel
template <int MODEL, int KC, bool STAGES, int L, int WPB, int SM = wave::kMaxS>
__global__ __launch_bounds__(WPB* kWave, waves_per_simd(KC, STAGES, L, SM)) void control_pack_kernel(const ControlParams<double> p_arg, const unsigned B,
                                                                     const int S_arg, const int rollout_arg)
{
  using R = double;
  (void)p_arg;  // read through the kernel-argument segment below
  constexpr int kLay = wave::kMaxS;  // the LDS layouts (hand-over of the controls between steps) are the same for every SM
  constexpr int kMaxS = SM;          // ... the register arrays and the slot loops are not (shadows wave::kMaxS from here on)
  static_assert(SM == 3 || SM == wave::kMaxS, "steps per lane: 3 or 4");
  static_assert(L == 8 || L == 16 || L == 32, "lanes per agent");
  static_assert(KC == 5 || KC == 10, "block contraction: K = 5 or 10");
  constexpr int A = kWave / L;
  constexpr int K = KC, K2 = K * K;
  constexpr int KS = tab_stride(KC);
  constexpr int NB = (KC + 3) / 4;
  constexpr int DS = d_stride(KC);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef const __attribute__((address_space(4))) ControlParams<R> KernArgParams;
  const int wave_of_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int n_steps = ((KernArgParams*)__builtin_amdgcn_kernarg_segment_ptr())->n_steps;
  // receding-horizon steps per launch (eea_control_batch_steps): as in control_wave_kernel, every step starts from the
  // hardware lane id and re-reads the launch parameters.  An agent SimpleCart rejected stays out for the rest of the launch
  // (control_wave_kernel: its wavefront returns): one bit per lane, wavefront-uniform
  unsigned long long rejected = 0ull;
  for (int step = 0; step < n_steps; ++step) {
  KernArgParams* ka = (KernArgParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  KernArgParams& p = *ka;
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  int wv = wave_of_block;
  asm volatile("" : "+s"(wv));
  int S = S_arg, rollout_only = rollout_arg;
  asm volatile("" : "+s"(S), "+s"(rollout_only));
  const unsigned wave_base = (blockIdx.x * WPB + wv) * A;  // first agent of this wavefront
  if (wave_base >= B) return;                              // wavefront-uniform
  const int tl = lane & (L - 1), al = lane / L;
  const unsigned b = wave_base + al;
  // (eea_batch_io::d_skip: an agent that is left out of the call is treated like one beyond the batch)
  const bool agent_in = b < B && !(p.skip != nullptr && p.skip[b < B ? b : 0] != 0);
  // the agent of lane ^ 32 (the partner whose other axis this lane stages)
  // kOneSet (16 lanes per agent): the four blocks of a matrix instruction are the four agents of the wavefront.  A half-pass
  // then takes the points of lanes tl in [8 h, 8 h + 8) of EVERY agent (not all points of the agents of one half of the
  // wavefront), the lane that stages the other axis of a point is lane ^ 8 -- a lane of the same agent --, and one accumulator
  // set holds all four agents' sums.
  constexpr bool kOneSet = (L == 16);
  constexpr int NSETS = kOneSet ? 1 : 2;
  const unsigned pb = kOneSet ? b : wave_base + (al ^ (A / 2));
  const bool partner_in = pb < B && !(p.skip != nullptr && p.skip[pb < B ? pb : 0] != 0);

  const int T = p.T;
  // (the wavefront's LDS carve is a multiple of 32 bytes -- region_elems is rounded to 4 reals, the park is 128 S reals -- but S is
  // a launch argument: told so, the compiler reads the rows of D in the gradient as 16-byte ds_read_b128 (4 LDS cycles, 64
  // banks) like the wavefront kernel does, instead of ds_read2_b64 pairs (8 cycles, 32 banks))
  R* const sm = static_cast<R*>(__builtin_assume_aligned(
      reinterpret_cast<R*>(smem_raw) + static_cast<size_t>(wv) * wave_lds_elems(KC, A, S), 32));
  R* const tabx = sm;  // [32 rows][KS], kGroupPad reals between rows 15 and 16
  R* const taby = tabx + kStageRows * KS + kGroupPad;
  R* const s_cp = sm + region_elems(KC, A);  // cos of the post-step heading, [j][lane]
  R* const s_sp = s_cp + S * kWave;          // sin
  R* const s_D = tabx;                       // D of agent a at a * DS (after the contraction)

  // lane -> horizon steps of its agent: S consecutive steps from S * tl.  Top-heavy horizons -- T = L (S - 1) + r with
  // r <= L / 8 at S = 4 (the yaml's T = 50 on 16 lanes: 3 x 16 + 2) -- as in control_wave_kernel: the first r lanes own S
  // steps, all others S - 1; the last slot then holds r steps per agent, whose gradient all lanes take together (8 lanes per
  // step, below) instead of a full pass at r / L of the lanes
  const int r_top = T - L * (S - 1);
  const bool top_heavy = SM == kLay && S == kLay && r_top >= 1 && r_top <= L / 8;  // wavefront-uniform
  const int i0 = top_heavy ? (tl < r_top ? S * tl : S * r_top + (S - 1) * (tl - r_top)) : S * tl;
  const int cnt = top_heavy ? (tl < r_top ? S : S - 1) : max(0, min(S, T - i0));
  // lanes of an agent that own a step in slot j
  auto lanes_in_slot = [&](int j) { return top_heavy ? (j < S - 1 ? L : r_top) : min(L, max(0, (T - j + S - 1) / S)); };
  R* const ut = p.ut + 3 * static_cast<size_t>(T) * b;
  const R* const pose = p.pose + 3 * (static_cast<size_t>(step) * p.pose_step_stride + b);
  // ---- controls: shift left by one column, last column zero (ergodic_control.hpp:233-234) ------------
  R vx[kMaxS], vy[kMaxS], w[kMaxS];
  bool bad = false;
  // the controls a step leaves for the next one change hands in LDS (everything there is dead between the update and the
  // next forward half): s_next[r][4 L al + step index], read back one column to the right (a rejected agent left nothing
  // there: it stays rejected, below)
  R* const s_next = sm + kLay * L * al;
  if (step == 0) {  // wavefront-uniform
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      vx[j] = vy[j] = w[j] = R(0);
      if (j < S) {
        const int src = rollout_only ? i0 + j : i0 + j + 1;  // optTraj rolls the controls out as they are
        if (agent_in && j < cnt && src < T) {
          vx[j] = ut[3 * src + 0];
          vy[j] = ut[3 * src + 1];
          w[j] = ut[3 * src + 2];
        }
        // SimpleCart::operator() rejects a lateral velocity (cart.hpp:167-170)
        if (MODEL == kModelSimpleCart && j < cnt && !(fabs(vy[j]) < R(1.0e-12))) bad = true;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      vx[j] = vy[j] = w[j] = R(0);
      if (j < S) {
        const int src = i0 + j + 1;
        if (agent_in && j < cnt && src < T) {
          vx[j] = s_next[0 * kLay * kWave + src];
          vy[j] = s_next[1 * kLay * kWave + src];
          w[j] = s_next[2 * kLay * kWave + src];
        }
        if (MODEL == kModelSimpleCart && j < cnt && !(fabs(vy[j]) < R(1.0e-12))) bad = true;
      }
    }
    lds_fence();  // (read before the forward half writes the park and the tiles)
  }
  R x0 = R(0), y0 = R(0), th0 = R(0);
  if (agent_in) {
    x0 = pose[0];
    y0 = pose[1];
    th0 = pose[2];
  }
  // the reference throws out of rk4_.solve: nothing of such an agent is touched (its lanes run on, their stores are off)
  bool agent_ok = agent_in;
  if (MODEL == kModelSimpleCart) {
    rejected |= __ballot(bad);
    const unsigned long long bm = rejected;
    const unsigned long long grp = (L == 32) ? 0xffffffffull : ((1ull << L) - 1ull);
    const bool agent_bad = ((bm >> (lane & ~(L - 1))) & grp) != 0ull;
    agent_ok = agent_in && !agent_bad;
    if (tl == 0 && agent_in && p.status != nullptr) {
      if (agent_bad) p.status[b] = 2;  // EEA_ERR_INVALID_TWIST
      else if (step == 0 && !(p.ck_flag != nullptr && p.status[b] == 6)) p.status[b] = 0;
    }
  } else if (tl == 0 && agent_in && p.status != nullptr && step == 0 && !(p.ck_flag != nullptr && p.status[b] == 6)) {
    p.status[b] = 0;  // (device-bound exchange: a time-out an earlier pass left in a reused buffer stays, as in control_wave_kernel)
  }

  const R dt = p.dt, dt6 = p.dt6;
  const R inv_pi = static_cast<R>(1.0 / kPi);

  // ================= forward half ========================================================================
  // heading: theta_i = wrap(theta_{i-1} + dt/6 (w + 2w + 2w + w)) (integrator.hpp:146-148,183)
  R thp[kMaxS];
  {
    R run = R(0);
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      const R d = dt6 * (((w[j] + R(2) * w[j]) + R(2) * w[j]) + w[j]);
      run += d;
      thp[j] = run;
    }
    const R incl = seg_inclusive_scan<L>(run, tl);
    const R base = wave::wrap_pi_fast(th0) + (incl - run);  // heading before the lane's first step
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) thp[j] += base;
  }
  // position: x_i = x_{i-1} + dt/6 (k1 + 2 k2 + 2 k3 + k4) with k2 == k3 (integrator.hpp:176-184)
  R px[kMaxS], py[kMaxS], incx[kMaxS], incy[kMaxS];
  {
    R c, s;
    {
      const R th_pre0 = thp[0] - dt6 * (((w[0] + R(2) * w[0]) + R(2) * w[0]) + w[0]);
      sincospi_r(th_pre0 * inv_pi, &s, &c);
    }
    bool small = true;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) small = small && (fabs(dt * (R(0.5) * w[j]) * inv_pi) <= R(0.0625));
    const bool fast_h = __all(small);
    R rx = R(0), ry = R(0);
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      incx[j] = incy[j] = R(0);
      if (j < S) {
        R sm_, cm, cpost, spost;
        if (fast_h) {  // wavefront-uniform
          R sd, cd;
          sincospi_small(dt * (R(0.5) * w[j]) * inv_pi, &sd, &cd);
          cm = c * cd - s * sd;
          sm_ = s * cd + c * sd;
          cpost = cm * cd - sm_ * sd;
          spost = sm_ * cd + cm * sd;
        } else {
          const R th_pre = (j == 0) ? thp[0] - dt6 * (((w[0] + R(2) * w[0]) + R(2) * w[0]) + w[0]) : thp[j - 1];
          sincospi_r((th_pre + dt * (R(0.5) * w[j])) * inv_pi, &sm_, &cm);
          const R c2m = R(1) - R(2) * sm_ * sm_, s2m = R(2) * sm_ * cm;
          cpost = c2m * c + s2m * s;
          spost = s2m * c - c2m * s;
        }
        R k1x, k1y, k2x, k2y, k4x, k4y;
        wave::model_xy<R, MODEL>(vx[j], vy[j], c, s, k1x, k1y);
        wave::model_xy<R, MODEL>(vx[j], vy[j], cm, sm_, k2x, k2y);
        wave::model_xy<R, MODEL>(vx[j], vy[j], cpost, spost, k4x, k4y);
        incx[j] = dt6 * (((k1x + R(2) * k2x) + R(2) * k2x) + k4x);
        incy[j] = dt6 * (((k1y + R(2) * k2y) + R(2) * k2y) + k4y);
        rx += incx[j];
        ry += incy[j];
        s_cp[j * kWave + lane] = cpost;
        s_sp[j * kWave + lane] = spost;
        c = cpost;
        s = spost;
      }
      px[j] = rx;
      py[j] = ry;
    }
    const R ix = seg_inclusive_scan<L>(rx, tl), iy = seg_inclusive_scan<L>(ry, tl);
    const R bx = x0 + (ix - rx), by = y0 + (iy - ry);
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      px[j] += bx;
      py[j] += by;
    }
  }
  if (STAGES && p.traj != nullptr && agent_ok) {
    R* const traj = p.traj + 3 * static_cast<size_t>(T) * b;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < cnt) {
        traj[3 * (i0 + j) + 0] = px[j];
        traj[3 * (i0 + j) + 1] = py[j];
        traj[3 * (i0 + j) + 2] = wave::wrap_pi_fast(thp[j]);
      }
    }
  }
  if (STAGES && rollout_only) return;  // optTraj / path (ergodic_control.hpp:313-342): the rollout is all

  // basis angles of the rollout points and the barrier gradient (:453-474), as in control_wave_kernel
  R c1x[kMaxS], s1x[kMaxS], c1y[kMaxS], s1y[kMaxS], g0[kMaxS], g1[kMaxS];
  bool small_b = true;
#pragma unroll
  for (int j = 1; j < kMaxS; ++j) {
    small_b = small_b && (fabs(incx[j] * p.inv_lx) <= R(0.0625)) && (fabs(incy[j] * p.inv_ly) <= R(0.0625));
  }
  const bool fast_b = __all(small_b);
#pragma unroll
  for (int j = 0; j < kMaxS; ++j) {
    c1x[j] = s1x[j] = c1y[j] = s1y[j] = g0[j] = g1[j] = R(0);
    if (j < S) {
      const R x = px[j] - p.map_x, y = py[j] - p.map_y;
      if (j > 0 && fast_b) {  // wavefront-uniform
        R sd, cd;
        sincospi_small(incx[j] * p.inv_lx, &sd, &cd);
        c1x[j] = c1x[j - 1] * cd - s1x[j - 1] * sd;
        s1x[j] = s1x[j - 1] * cd + c1x[j - 1] * sd;
        sincospi_small(incy[j] * p.inv_ly, &sd, &cd);
        c1y[j] = c1y[j - 1] * cd - s1y[j - 1] * sd;
        s1y[j] = s1y[j - 1] * cd + c1y[j - 1] * sd;
      } else {
        sincospi_r(x * p.inv_lx, &s1x[j], &c1x[j]);
        sincospi_r(y * p.inv_ly, &s1y[j], &c1y[j]);
      }
      const R eps = R(0.05), weight2 = R(50);
      g0[j] = (fmax(x - (p.lx - eps), R(0)) + fmin(x - eps, R(0))) * weight2;
      g1[j] = (fmax(y - (p.ly - eps), R(0)) + fmin(y - eps, R(0))) * weight2;
    }
  }

  // ---- c_k = (1/N) sum_p cos(a_k1 x_p) cos(b_k2 y_p)  (basis.cpp:109-120) on the matrix cores -----------
  int nmem = 0;
  if (p.mem_cols != nullptr && agent_in) {
    nmem = (p.n_mem != nullptr) ? p.n_mem[b] : static_cast<int>(p.mem_stride);
    nmem = nmem < 0 ? 0 : (nmem > static_cast<int>(p.mem_stride) ? static_cast<int>(p.mem_stride) : nmem);
  }
  R cacc[NSETS][NB][NB];  // one accumulator set per half of the wavefront (kOneSet: one for all four agents)
#pragma unroll
  for (int h = 0; h < NSETS; ++h) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
      for (int J = 0; J < NB; ++J) cacc[h][I][J] = R(0);
    }
  }
  // lo: this lane's own point belongs to the FIRST half-pass of a slot (lanes 0..31; kOneSet: lanes tl < 8 of every agent)
  const bool lo = kOneSet ? ((lane & 8) == 0) : (lane < 32);
  const int trow = kOneSet ? tile_row<8>(8 * al + (tl & 7)) : tile_row<L>(lane & 31);
  // steps of the PARTNER lane, whose other axis this lane stages: the lane at the same place of another agent (the same
  // count) -- kOneSet: lane ^ 8 of the same agent, another place in the horizon
  const int ptl = tl ^ 8;
  const int cnt_p = !kOneSet ? cnt : (top_heavy ? (ptl < r_top ? S : S - 1) : max(0, min(S, T - S * ptl)));
  const int cnt_lower = lo ? cnt : cnt_p;  // steps of the lane whose point this lane stages in the first / second half-pass
  const int cnt_upper = lo ? cnt_p : cnt;
  const int trow_off = trow * KS + (trow >= 16 ? kGroupPad : 0);
  R* const st_lower = (lo ? tabx : taby) + trow_off;  // staging the tile of the first half-pass's points
  R* const st_upper = (lo ? taby : tabx) + trow_off;  // ... of the second
  struct Tab1
  {
    R a, b, two;  // T_k, T_{k+1}, 2 cos
  };
  auto tab1_init = [&](R c, bool valid) {
    Tab1 t;
    t.a = valid ? R(1) : R(0);
    t.b = valid ? c : R(0);
    t.two = c + c;
    return t;
  };
  auto tab1_store = [&](const Tab1& t, R* dst, int k) { *reinterpret_cast<double2*>(dst + k) = double2{ t.a, t.b }; };
  auto tab1_step = [&](Tab1& t) {
    const R c = t.two * t.b - t.a, d = t.two * c - t.b;
    t.a = c;
    t.b = d;
  };
  auto stage_cos = [&](R ca, R cb, R& c_lower, R& c_upper) {
    if constexpr (kOneSet) {
      const R partner_cb = dpp_or_zero<0x128, 0xf>(cb);  // row_ror:8 -- the cosine of lane ^ 8, a lane of the same agent
      c_lower = lo ? ca : partner_cb;
      c_upper = lo ? partner_cb : ca;
    } else {
      R from_lower, from_upper;
      wave::half_swap(cb, from_lower, from_upper);
      c_lower = lo ? ca : from_lower;
      c_upper = lo ? from_upper : ca;
    }
  };
  constexpr int kPairs = KS / 2;
  const int orow = 4 * ((lane >> 2) & 3) + 2 * ((lane >> 4) & 1) + (lane >> 5), oi = lane & 3;
  R qa[2][NB], qb[2][NB];
  const unsigned oaddr = wave::lds_addr(tabx + orow * KS + oi);
  constexpr bool kLean = (KC == 10) && (EEA_PACK_LEAN != 0) && !kOneSet;
  auto read_operands4 = [&]() { wave::OperandReads4<KS, NB, 0, 0, kGroupPad>::run(oaddr, qa, qb); };
  auto operands4_ready = [&]() { wave::wait_operands4<NB>(qa, qb); };
  // lean: group q of the tile into the ONE operand set (index 0)
  auto read_group = [&](int q) {
    if (q == 0) wave::OperandReads4One<KS, NB, 0, 0, kGroupPad>::run(oaddr, qa[0], qb[0]);
    else wave::OperandReads4One<KS, NB, 1, 0, kGroupPad>::run(oaddr, qa[0], qb[0]);
  };
  auto group_ready = [&]() { wave::wait_operands4_one<NB>(qa[0], qb[0]); };
  auto mma4_group = [&](int h_arg, int q) {
    const int h = kOneSet ? 0 : h_arg;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
      for (int J = 0; J < NB; ++J) cacc[h][I][J] = wave::mfma4(qa[q][I], qb[q][J], cacc[h][I][J]);
    }
  };
  // the matrix instructions of one 16-row group with the staging of `dst` (kPairs stores) spread between them
  auto mma4_group_staging = [&](int h_arg, int q, Tab1& u, R* dst) {
    const int h = kOneSet ? 0 : h_arg;
    constexpr int kEvery = (NB * NB + kPairs - 1) / kPairs;
    int done = 0;
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
      for (int J = 0; J < NB; ++J) {
        cacc[h][I][J] = wave::mfma4(qa[q][I], qb[q][J], cacc[h][I][J]);
        const int n = I * NB + J + 1;
        if (n % kEvery == 0 && done < kPairs) {
          tab1_store(u, dst, 2 * done);
          tab1_step(u);
          ++done;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int q2 = 0; q2 < kPairs; ++q2) {
      if (q2 >= done) {
        tab1_store(u, dst, 2 * q2);
        tab1_step(u);
      }
    }
  };
  {
    // Rollout points, software-pipelined as in control_wave_kernel: the operands of a half-pass are read first, then the
    // recurrence + stores of the NEXT half-pass run between the matrix instructions of the first 16-row group; the second
    // group (lanes tl >= L / 2 of every agent) is skipped when no agent has a point there.
    R cl, cu;
    stage_cos(c1x[0], c1y[0], cl, cu);
    {
      Tab1 u = tab1_init(cl, (lo ? agent_in : partner_in) && 0 < cnt_lower);
#pragma unroll
      for (int q = 0; q < kPairs; ++q) {
        tab1_store(u, st_lower, 2 * q);
        tab1_step(u);
      }
    }
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < S) {  // wavefront-uniform
        // which 16-row groups of the two half-passes hold a point (wavefront-uniform): a half-pass takes lanes tl in
        // [0, L / 2) resp. [L / 2, L) of the agents of its half of the wavefront -- kOneSet: tl in [0, 8) resp. [8, 16) of every
        // agent, so that the whole second half-pass goes when no agent owns a step there (the tail slot of yaml's T = 50)
        const int lis = lanes_in_slot(j);
        const bool second0 = kOneSet ? lis > 4 : lis > L / 2;
        const bool any1 = kOneSet ? lis > 8 : true;
        const bool second1 = kOneSet ? lis > 12 : lis > L / 2;
        const bool second = second0;
        lds_fence();
        if constexpr (kLean) {
          if (second) {  // wavefront-uniform: the first group's operands, its matrix instructions, then the second group's
            read_group(0);
            group_ready();
            mma4_group(0, 0);
            read_group(1);
          } else {
            read_group(0);
          }
          lds_fence();  // the last group's operands are on their way to the registers: the tile is free
          Tab1 u = tab1_init(cu, (lo ? partner_in : agent_in) && j < cnt_upper);
          group_ready();
          mma4_group_staging(0, 0, u, st_upper);
        } else {
          read_operands4();  // the first half-pass's rows, slot j
          lds_fence();       // operands in registers: the tile is free
          Tab1 u = tab1_init(cu, (lo ? partner_in : agent_in) && j < cnt_upper);
          operands4_ready();
          mma4_group_staging(0, 0, u, st_upper);
          if (second0) mma4_group(0, 1);
        }
        lds_fence();
        const int jn = (j + 1 < kMaxS) ? j + 1 : j;
        if constexpr (kLean) {
          if (second) {
            read_group(0);
            group_ready();
            mma4_group(1, 0);
            read_group(1);
          } else {
            read_group(0);
          }
          lds_fence();
          stage_cos(c1x[jn], c1y[jn], cl, cu);
          Tab1 u = tab1_init(cl, (lo ? agent_in : partner_in) && (j + 1 < S) && (j + 1 < cnt_lower));
          group_ready();
          mma4_group_staging(1, 0, u, st_lower);
        } else if (any1) {  // wavefront-uniform
          read_operands4();  // the second half-pass's rows, slot j
          lds_fence();
          stage_cos(c1x[jn], c1y[jn], cl, cu);
          Tab1 u = tab1_init(cl, (lo ? agent_in : partner_in) && (j + 1 < S) && (j + 1 < cnt_lower));
          operands4_ready();
          mma4_group_staging(1, 0, u, st_lower);
          if (second1) mma4_group(1, 1);
        } else {  // no point in the second half-pass: only the next slot's first tile is staged
          stage_cos(c1x[jn], c1y[jn], cl, cu);
          Tab1 u = tab1_init(cl, (lo ? agent_in : partner_in) && (j + 1 < S) && (j + 1 < cnt_lower));
#pragma unroll
          for (int q = 0; q < kPairs; ++q) {
            tab1_store(u, st_lower, 2 * q);
            tab1_step(u);
          }
        }
      }
    }
    lds_fence();
  }
  // sampled past states are prepended (buffer.cpp:78-108) and shifted like the rollout: rounds of L columns per agent
  if (p.mem_cols != nullptr) {
    const R* const mem = p.mem_cols + 3 * static_cast<size_t>(p.mem_stride) * b;
    for (int c0 = 0; __any(c0 < nmem); c0 += L) {
      const int q = c0 + tl;
      const bool valid = q < nmem;  // (nmem = 0 for agents outside the batch)
      R sa, ca = R(0), sb, cb = R(0);
      if (valid) {
        sincospi_r((mem[3 * q + 0] - p.map_x) * p.inv_lx, &sa, &ca);
        sincospi_r((mem[3 * q + 1] - p.map_y) * p.inv_ly, &sb, &cb);
      }
      const unsigned long long vm = __ballot(valid);
      const bool partner_valid = ((vm >> (lane ^ (kOneSet ? 8 : 32))) & 1ull) != 0ull;
      // columns in the second 16-row group of their half-pass
      const unsigned long long vm2 = __ballot(valid && (kOneSet ? (tl & 7) >= 4 : tl >= L / 2));
      R cl, cu;
      stage_cos(ca, cb, cl, cu);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // the lanes whose own column belongs to half-pass h
        const unsigned long long hmask = kOneSet ? (0x00ff00ff00ff00ffull << (8 * h)) : (0xffffffffull << (32 * h));
        const bool half_any = (vm & hmask) != 0ull, half_second_b = (vm2 & hmask) != 0ull;
        const unsigned half_second = half_second_b ? 1u : 0u;
        if (half_any) {  // wavefront-uniform
          {
            // lanes of half h stage the x axis of their own point, the others the y axis of their partner's
            const bool own = (h == 0) == lo;
            Tab1 t = tab1_init(h ? cu : cl, own ? valid : partner_valid);
            R* const dst = h ? st_upper : st_lower;
#pragma unroll
            for (int qq = 0; qq < kPairs; ++qq) {
              tab1_store(t, dst, 2 * qq);
              tab1_step(t);
            }
          }
          lds_fence();
          if constexpr (kLean) {
            read_group(0);
            group_ready();
            mma4_group(h, 0);
            if (half_second != 0u) {
              read_group(1);
              group_ready();
              mma4_group(h, 0);
            }
            lds_fence();
          } else {
            read_operands4();
            lds_fence();
            operands4_ready();
            mma4_group(h, 0);
            if (half_second != 0u) mma4_group(h, 1);
          }
        }
      }
    }
  }

  // ---- D = lambda (c - phi), fourier_diff of ergodic_control.hpp:422, per agent ---------------------------------
  {
    // lane l = 16 i + 4 bb + j holds, in set h, block bb's sum for c(k1 = 4 I + i, k2 = 4 J + j); the blocks of one agent
    // are added by row rotations (L = 16: bb and bb ^ 2; L = 32: all four)
    const int di = lane >> 4, db = (lane >> 2) & 3, dj = lane & 3;
    // lambda_k and phi_k of this lane's entries.  Preloaded for all NB^2 entries (the loads run under the rotations: what a
    // batch of one wavefront per SIMD wants -- read one pair at a time where D is formed, configs[1] at 4096 agents took 9.4
    // instead of 8.0 us) they are 4 NB^2 registers on top of the accumulators and the per-step state: the kernel's register
    // peak (146 of 162).  The 16-lane instances, compiled for four wavefronts per SIMD, read them where they are used: the
    // tables are 800 bytes, L1-resident, and the other wavefronts of the SIMD cover the latency.
    constexpr bool kPreload = !kOneSet && !(L == 8 && SM == 3);
    R lamv[kPreload ? NB : 1][kPreload ? NB : 1], phiv[kPreload ? NB : 1][kPreload ? NB : 1];
    if constexpr (kPreload) {
#pragma unroll
      for (int I = 0; I < NB; ++I) {
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          const int k1 = 4 * I + di, k2 = 4 * J + dj;
          const int idx = (k1 < K && k2 < K) ? k2 * K + k1 : 0;
          lamv[I][J] = p.lamdak[idx];
          phiv[I][J] = p.phik[idx];
        }
      }
    }
    const bool writer = (L == 8) || kOneSet || (L == 32 && db == 0);  // one copy stores c_k (kOneSet: every block is an agent)
    // The agents' sum records (eea_batch_io::d_ck_rec: [c_k, 1, pad]; decentralised consensus, README ref. [2]) leave from the
    // registers that hold c_k -- the same lanes, the same addresses within the agent as the d_ck stores --, BEFORE anything
    // of the shared c_k is read: with the device-bound exchange (d_rec_ready) they go write-through, the agents' ready marks
    // follow once they have left, and only then does the wavefront wait for the flag of the record it consumes
    // (control_wave_kernel's publish_record / bind_shared_ck, per agent of the wavefront).  A rejected agent's record is all
    // zero (it does not count in the sum), an agent left out of the call (d_skip) writes nothing.
    const bool rec_on = p.ck_rec != nullptr && !rollout_only;  // wavefront-uniform
    const bool rec_wt = p.rec_ready != nullptr;                // wavefront-uniform: write-through + ready marks
    // c_k leaves the wavefront (d_ck, d_ck_rec) THROUGH LDS: the accumulator lanes hold an agent's c_k scattered (k1 = 4 I + i,
    // k2 = 4 J + j: 8-byte stores 80 bytes apart -- with tens of thousands of packed agents per pass that write pattern, not the
    // arithmetic, set the pass time: yaml T = 50, 32 768 agents, records on: 72 against 51 us).  Staged at the agent's place in the
    // D region (D itself is formed afterwards, from the registers), each agent's own lanes then write its K^2 (+ count, pad)
    // values as contiguous runs of L reals
    const bool stage_out = rec_on || p.ck != nullptr;          // wavefront-uniform
#pragma unroll
    for (int h = 0; h < NSETS; ++h) {
      const int ab = kOneSet ? db : block_agent<L>(h, db);
      const unsigned bb = wave_base + ab;
      // (an agent left out of the call is treated like one beyond the batch: nothing of it is read or written)
      const bool in = bb < B && !(p.skip != nullptr && p.skip[bb < B ? bb : 0] != 0);
      int nm = 0;
      if (p.mem_cols != nullptr && in) {
        nm = (p.n_mem != nullptr) ? p.n_mem[bb] : static_cast<int>(p.mem_stride);
        nm = nm < 0 ? 0 : (nm > static_cast<int>(p.mem_stride) ? static_cast<int>(p.mem_stride) : nm);
      }
      const R invN = R(1) / static_cast<R>(T + nm);
#pragma unroll
      for (int I = 0; I < NB; ++I) {
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          R v = cacc[h][I][J];
          if constexpr (L == 16 && !kOneSet) v = wave::add_row_ror<8>(v);
          if constexpr (L == 32) v = wave::add_row_ror<8>(wave::add_row_ror<4>(v));
          const R c = invN * v;
          cacc[h][I][J] = c;
          const int k1 = 4 * I + di, k2 = 4 * J + dj;
          if (stage_out && k1 < K && k2 < K && writer) s_D[ab * DS + k2 * K + k1] = c;
        }
      }
    }
    const bool rec_wave = rec_on && p.rec_wave != 0;  // wavefront-uniform
    if (stage_out) {  // wavefront-uniform
      lds_fence();
      const unsigned rec_len = p.rec_len;
      R* const rec = (rec_on && !rec_wave) ? p.ck_rec + static_cast<size_t>(b < B ? b : 0) * rec_len : nullptr;
      R* const ckb = p.ck != nullptr ? p.ck + static_cast<size_t>(b < B ? b : 0) * K2 : nullptr;
      constexpr int kOutRounds = (ck_record_len(K2) + L - 1) / L;
#pragma unroll
      for (int r = 0; r < kOutRounds; ++r) {
        const int e = L * r + tl;
        const R cv = (e < K2) ? s_D[al * DS + (e < K2 ? e : 0)] : R(0);
        // (d_ck: nothing of a rejected agent is written, like the wavefront kernel, whose wavefront returns)
        if (ckb != nullptr && e < K2 && agent_ok) ckb[e] = cv;
        if (rec != nullptr && agent_in && e < static_cast<int>(rec_len)) {
          // a rejected agent's record is all zero (it does not count in the sum); element K^2 = 1: this agent counts; pad 0
          const R v = !agent_ok ? R(0) : (e < K2 ? cv : (e == K2 ? R(1) : R(0)));
          if (rec_wt) store_agent(rec + e, v);
          else rec[e] = v;
        }
      }
      if (rec_wt && !rec_wave) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the records of every agent of this wavefront have left
        if (tl == 0 && agent_in) store_agent(p.rec_ready + b, p.rec_seq);
      }
      lds_fence();  // (D takes the staging's place below)
    }
    // eea_batch_io::rec_per_wavefront: the wavefront's agents are added here (agent order; the staged c_k are read by all 64
    // lanes, element e = lane, lane + 64) and ONE record [sum, number of accepted agents, pad] leaves: A = 64 / L times
    // fewer bytes through eea_ck_records_sum, which at chip-filling packed batches is what the pass was waiting for
    if (rec_wave) {
      const unsigned long long okm = __ballot(agent_ok), inm = __ballot(agent_in);
      const unsigned rec_len = p.rec_len;
      const unsigned wrec = wave_base / A;
      R* const rec = p.ck_rec + static_cast<size_t>(wrec) * rec_len;
      constexpr int kWaveRounds = (ck_record_len(K2) + 63) / 64;
      if (inm != 0ull) {  // (a wavefront whose agents are all left out of the call writes nothing)
        // bit a L of okm: agent a is accepted (every lane of an agent holds the same flag)
        unsigned long long first = 0ull;
#pragma unroll
        for (int a = 0; a < A; ++a) first |= 1ull << (a * L);
        const int n_ok = __popcll(okm & first);  // wavefront-uniform
#pragma unroll 1
        for (int r = 0; r < kWaveRounds; ++r) {
          // (the lane index is counted afresh: nothing is kept in a register across the contraction for this block)
          const int e = 64 * r + static_cast<int>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
          const R* const src = s_D + (e < K2 ? e : 0);
          R v = R(0);
#pragma unroll 1
          for (int a = 0; a < A; ++a) {
            if (((okm >> (a * L)) & 1ull) != 0ull) v += src[a * DS];  // (wavefront-uniform branch)
          }
          v = e < K2 ? v : (e == K2 ? static_cast<R>(n_ok) : R(0));
          if (e < static_cast<int>(rec_len)) {
            if (rec_wt) store_agent(rec + e, v);
            else rec[e] = v;
          }
        }
        if (rec_wt) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) store_agent(p.rec_ready + wrec, p.rec_seq);
        }
      }
      lds_fence();
    }
    // device-bound exchange, consumer side: the shared c_k may still be on its way -- wait for its flag here, right before
    // its first use.  On a time-out the agents of this wavefront go on with their own c_k and say so
    bool use_shared = p.ck_shared != nullptr;  // wavefront-uniform
    if (use_shared && p.ck_flag != nullptr) {
      bool ok = wait_flag(p.ck_flag, p.ck_flag_seq);
      if (ok && p.ck_shared_parts > 0) ok = !(load_agent(p.ck_shared + K2) < R(0));  // a producer that gave up
      if (!ok && tl == 0 && agent_ok && p.status != nullptr) p.status[b] = 6;  // EEA_ERR_TIMEOUT
      use_shared = ok;
    }
    SharedCk<R> shared_ck{ R(1), true };
    if (use_shared) shared_ck = shared_ck_begin<R>(p, p.ck_shared, K2);  // wavefront-uniform
#pragma unroll
    for (int h = 0; h < NSETS; ++h) {
      const int ab = kOneSet ? db : block_agent<L>(h, db);
#pragma unroll
      for (int I = 0; I < NB; ++I) {
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          const int k1 = 4 * I + di, k2 = 4 * J + dj;
          if (k1 < K && k2 < K) {
            const int idx = k2 * K + k1;
            R c = cacc[h][I][J];
            // decentralised consensus (eea_batch_io::d_ck_shared): the agents' shared c_k replaces the own one
            if (use_shared) c = shared_ck_value(p, shared_ck, p.ck_shared, idx, c);
            if constexpr (kPreload) s_D[ab * DS + idx] = lamv[I][J] * (c - phiv[I][J]);
            else s_D[ab * DS + idx] = p.lamdak[idx] * (c - p.phik[idx]);
          }
        }
      }
    }
    lds_fence();
  }

  // ================= backward half =======================================================================
  // per step: ergodic-metric gradient (:418-436, basis.cpp:91-107), one pass over the agent's D per step
  //   edx_x = -pi/lx sin(a x) sum_k1 k1 U_{k1-1}(cos a x) G(k1),  G(k1) = sum_k2 D(k1,k2) cos(b_k2 y)
  //   edx_y = -pi/ly sin(b y) sum_k2 k2 U_{k2-1}(cos b y) H(k2),  H(k2) = sum_k1 D(k1,k2) cos(a_k1 x)
  R ex[STAGES ? kMaxS : 1], ey[STAGES ? kMaxS : 1];
  const R* const Da = s_D + al * DS;
  // 16-byte reads of D's rows (K even): ds_read_b128 instead of the compiler's ds_read2_b64 pairs.  EEA_PACK_WIDE_D: 0 = never,
  // 1 = slot loop, 2 = slot loop and the cooperative tail.  Same box: configs[1] (8 lanes per agent) +2.8 %, yaml T = 50 (16 lanes)
  // +0.8 % -- but the 16-lane instance with four steps per lane sits AT 128 registers, and the aligned register quads of the
  // wide reads cost it 12 - 28 bytes of scratch: it keeps the narrow reads (126 registers, no scratch)
#ifndef EEA_PACK_WIDE_D
#define EEA_PACK_WIDE_D 2
#endif
  constexpr bool kWideD = (K % 2 == 0) && (EEA_PACK_WIDE_D >= 1) && !(kOneSet && SM == kLay);
  constexpr bool kWideTail = (K % 2 == 0) && (EEA_PACK_WIDE_D >= 2) && !(kOneSet && SM == kLay);
#pragma unroll
  for (int j = 0; j < kMaxS; ++j) {
    if (STAGES) ex[j] = ey[j] = R(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (j < S && !(j == kMaxS - 1 && top_heavy)) {
      R accx = R(0), accy = R(0);
      const R twox = c1x[j] + c1x[j], twoy = c1y[j] + c1y[j];
      R cxa[K], G[K];
      {
        R ta = R(1), tb = c1x[j];
#pragma unroll
        for (int i = 0; i < K; ++i) {
          cxa[i] = ta;
          const R tn = twox * tb - ta;
          ta = tb;
          tb = tn;
        }
      }
#pragma unroll
      for (int i = 0; i < K; ++i) G[i] = (i == 0) ? R(0) : Da[i];  // row k2 = 0: cos(0 y) = 1, k2 sin(0) = 0
      R um = R(0), u0 = R(1);    // U_{k2-2}, U_{k2-1} of the y angle
      R tm = R(1), t0 = c1y[j];  // T_{k2-1}, T_{k2}
#pragma unroll
      for (int k2 = 1; k2 < K; ++k2) {
        const R* const row = Da + k2 * K;
        R h = R(0);
        // the row of D, two entries per LDS read where the rows are 16-byte aligned (K even: 80-byte rows, agents 800 bytes apart):
        // ds_read_b128 -- 4 LDS cycles on 64 banks -- instead of the ds_read2_b64 pairs the compiler forms from 8-byte reads
        // (8 cycles, 32 banks: the alignment of a row of another agent's D is not visible to it)
        auto entry = [&](int i, R d) {
          if (i > 0) G[i] += d * t0;
          if (i == 0) h = d;  // cos(0 x) = 1
          else h += d * cxa[i];
        };
        if constexpr (kWideD) {
#pragma unroll
          for (int i = 0; i < K; i += 2) {
            const double2 dd = *reinterpret_cast<const double2*>(row + i);
            entry(i, dd.x);
            entry(i + 1, dd.y);
          }
        } else {
#pragma unroll
          for (int i = 0; i < K; ++i) entry(i, row[i]);
        }
        accy = wave::fma_k(u0 * h, k2, accy);
        const R un = twoy * u0 - um;
        um = u0;
        u0 = un;
        const R tn = twoy * t0 - tm;
        tm = t0;
        t0 = tn;
      }
      {
        R ua = R(0), ub = R(1);  // U_{k-1}, U_k of the x angle
#pragma unroll
        for (int i = 0; i < K; ++i) {
          if (i > 0) accx = wave::fma_k(ua * G[i], i, accx);
          const R un = twox * ub - ua;
          ua = ub;
          ub = un;
        }
      }
      const R exj = (-p.pi_lx * s1x[j] * accx) * p.expl_weight;
      const R eyj = (-p.pi_ly * s1y[j] * accy) * p.expl_weight;
      if (STAGES) {
        ex[j] = exj;
        ey[j] = eyj;
      } else {
        // g = edx + bdx (inactive steps contribute nothing to the suffix sums)
        const bool act = j < cnt;
        g0[j] = act ? exj + g0[j] : R(0);
        g1[j] = act ? eyj + g1[j] : R(0);
      }
    }
  }
  if constexpr (SM == kLay) {
  if (top_heavy) {  // wavefront-uniform
    // The tail slot, all lanes together (control_wave_kernel's cooperative tail with BLOCK <-> AGENT): lane l = 8 st + sub
    // works on tail step e = (l % L) / 8 of ITS OWN agent -- the step of that agent's lane e --, rows k2 = sub and sub + 8 of
    // that agent's D:
    //   edx_x = -pi/lx sin(a) sum_k2 cos(k2 b) Bx(k2),   Bx(k2) = sum_k1 D(k2,k1) k1 U_{k1-1}(cos a)
    //   edx_y = -pi/ly        sum_k2 k2 sin(k2 b) A(k2),  A(k2)  = sum_k1 D(k2,k1) cos(k1 a)
    // (the same sums as above in another order).  Nothing crosses an agent: the 8 lanes of a step are lanes of its agent.
    constexpr int jt = kMaxS - 1;
    constexpr int E = L / 8;  // tail steps per agent at most
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int st = lane >> 3, sub = lane & 7;
    R* const s_tail = s_D + A * DS;  // [3][8]
    const int own = al * E + (tl & (E - 1));  // the hand-over slot of this lane's own tail step (lanes tl < E)
    if (tl < E) {
      s_tail[own] = c1x[jt];
      s_tail[8 + own] = c1y[jt];
      s_tail[16 + own] = s1y[jt];
    }
    lds_fence();
    const R cxs = s_tail[st], cys = s_tail[8 + st], sys = s_tail[16 + st];
    lds_fence();  // (the hand-over space is written again below)
    R cxa[K], kux[K];  // cos(k1 a), k1 U_{k1-1}(cos a)
    {
      const R two = cxs + cxs;
      R ta = R(1), tb = cxs, ua = R(0), ub = R(1);
#pragma unroll
      for (int k1 = 0; k1 < K; ++k1) {
        cxa[k1] = ta;
        kux[k1] = static_cast<R>(k1) * ua;
        const R tn = two * tb - ta, un = two * ub - ua;
        ta = tb;
        tb = tn;
        ua = ub;
        ub = un;
      }
    }
    // (cos, sin)(k2 b) for k2 = sub by binary powering of z = (cos b, sin b): z^sub = z^(bit 0) z2^(bit 1) z4^(bit 2)
    const R z2c = cys * cys - sys * sys, z2s = (cys + cys) * sys;
    const R z4c = z2c * z2c - z2s * z2s, z4s = (z2c + z2c) * z2s;
    R wc = (sub & 1) ? cys : R(1), ws = (sub & 1) ? sys : R(0);
    {
      const R mc = (sub & 2) ? z2c : R(1), ms = (sub & 2) ? z2s : R(0);
      const R nc = wc * mc - ws * ms, ns = wc * ms + ws * mc;
      wc = nc;
      ws = ns;
    }
    {
      const R mc = (sub & 4) ? z4c : R(1), ms = (sub & 4) ? z4s : R(0);
      const R nc = wc * mc - ws * ms, ns = wc * ms + ws * mc;
      wc = nc;
      ws = ns;
    }
    R tpx = R(0), tpy = R(0);
    auto tail_row = [&](int k2, R cyv, R ksv) {  // k2 < K; cyv = cos(k2 b), ksv = k2 sin(k2 b) (both 0: row not taken)
      const R* const row = Da + k2 * K;
      R a0 = R(0), a1 = R(0), b0 = R(0), b1 = R(0);
      auto entry = [&](int k1, R d) {
        if (k1 & 1) {
          a1 += d * cxa[k1];
          b1 += d * kux[k1];
        } else {
          a0 += d * cxa[k1];
          if (k1 > 0) b0 += d * kux[k1];
        }
      };
      if constexpr (kWideTail) {  // (two entries per LDS read where the rows are 16-byte aligned, as in the slot loop above)
#pragma unroll
        for (int k1 = 0; k1 < K; k1 += 2) {
          const double2 dd = *reinterpret_cast<const double2*>(row + k1);
          entry(k1, dd.x);
          entry(k1 + 1, dd.y);
        }
      } else {
#pragma unroll
        for (int k1 = 0; k1 < K; ++k1) entry(k1, row[k1]);
      }
      tpx += cyv * (b0 + b1);
      tpy += ksv * (a0 + a1);
    };
    tail_row(sub < K ? sub : K - 1, sub < K ? wc : R(0), sub < K ? static_cast<R>(sub) * ws : R(0));
    if (K > 8) {  // rows 8 .. K-1: z^(sub + 8) = z^sub z8
      const R z8c = z4c * z4c - z4s * z4s, z8s = (z4c + z4c) * z4s;
      const R vc = wc * z8c - ws * z8s, vs = wc * z8s + ws * z8c;
      const bool has = sub + 8 < K;
      tail_row(has ? sub + 8 : K - 1, has ? vc : R(0), has ? static_cast<R>(sub + 8) * vs : R(0));
    }
    // the 8 lanes of a step: xor 1, xor 2 (quad permutations), then the mirror image inside the half row
    tpx += dpp_or_zero<0xB1, 0xf>(tpx);
    tpy += dpp_or_zero<0xB1, 0xf>(tpy);
    tpx += dpp_or_zero<0x4E, 0xf>(tpx);
    tpy += dpp_or_zero<0x4E, 0xf>(tpy);
    tpx += dpp_or_zero<0x141, 0xf>(tpx);
    tpy += dpp_or_zero<0x141, 0xf>(tpy);
    if (sub == 0) {
      s_tail[2 * st] = tpx;
      s_tail[2 * st + 1] = tpy;
    }
    lds_fence();
    const R accx = s_tail[2 * own], accy = s_tail[2 * own + 1];
    lds_fence();
    const R exj = (-p.pi_lx * s1x[jt] * accx) * p.expl_weight;
    const R eyj = (-p.pi_ly * accy) * p.expl_weight;  // (sin b is in the row factors)
    if (STAGES) {
      ex[jt] = exj;
      ey[jt] = eyj;
    } else {
      const bool act = jt < cnt;  // lanes tl < r of every agent
      g0[jt] = act ? exj + g0[jt] : R(0);
      g1[jt] = act ? eyj + g1[jt] : R(0);
    }
  }
  }  // SM == kLay
  if (STAGES && agent_ok) {
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < cnt) {
        if (p.edx != nullptr) {
          R* const o = p.edx + 3 * (static_cast<size_t>(T) * b + i0 + j);
          o[0] = ex[j];
          o[1] = ey[j];
          o[2] = R(0);
        }
        if (p.bdx != nullptr) {
          R* const o = p.bdx + 3 * (static_cast<size_t>(T) * b + i0 + j);
          o[0] = g0[j];
          o[1] = g1[j];
          o[2] = R(0);
        }
      }
    }
  }

  // co-state (suffix sums over the agent's horizon), as in control_wave_kernel; the controls again from L2
  R vxr[kMaxS], vyr[kMaxS];
  {
    size_t opaque = 0;
    asm volatile("" : "+v"(opaque));
    const R* const ut_again = ut + opaque;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      const int src = i0 + j + 1;
      const bool ok = agent_in && j < cnt && src < T;
      vxr[j] = vyr[j] = R(0);
      if (ok) {
        vxr[j] = ut_again[3 * src];
        if (MODEL == kModelOmni) vyr[j] = ut_again[3 * src + 1];
      }
    }
  }
  R r0[kMaxS], r1[kMaxS];
  R tot0, tot1, tot2;  // the agent's totals of the three co-state scans
  {
    R s0 = R(0), s1 = R(0);
#pragma unroll
    for (int j = kMaxS - 1; j >= 0; --j) {
      if (STAGES) {
        const bool act = j < cnt;
        g0[j] = act ? ex[j] + g0[j] : R(0);
        g1[j] = act ? ey[j] + g1[j] : R(0);
      }
      s0 += dt * g0[j];
      s1 += dt * g1[j];
      r0[j] = s0;
      r1[j] = s1;
    }
    const R i0s = seg_inclusive_scan<L>(s0, tl), i1s = seg_inclusive_scan<L>(s1, tl);
    tot0 = seg_last<L>(i0s, lane);
    tot1 = seg_last<L>(i1s, lane);
    const R o0 = tot0 - i0s, o1 = tot1 - i1s;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      r0[j] += o0;
      r1[j] += o1;
    }
  }
  R r2[kMaxS], cth[kMaxS], sth[kMaxS];
  {
    R s2 = R(0);
#pragma unroll
    for (int j = kMaxS - 1; j >= 0; --j) {
      cth[j] = s_cp[j * kWave + lane];
      sth[j] = s_sp[j * kWave + lane];
      R a02, a12;
      if (MODEL == kModelOmni) {
        a02 = -vxr[j] * sth[j] - vyr[j] * cth[j];
        a12 = vxr[j] * cth[j] - vyr[j] * sth[j];
      } else {
        a02 = -vxr[j] * sth[j];
        a12 = vxr[j] * cth[j];
      }
      const R sE = a02 * (r0[j] - dt * g0[j]) + a12 * (r1[j] - dt * g1[j]);
      const R sG = a02 * g0[j] + a12 * g1[j];
      const R qv = (j < S) ? dt * (sE + p.half_dt * sG) : R(0);
      s2 += qv;
      r2[j] = s2;
    }
    const R i2s = seg_inclusive_scan<L>(s2, tl);
    tot2 = seg_last<L>(i2s, lane);
    const R o2 = tot2 - i2s;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) r2[j] += o2;
  }

  // ---- u_i = clamp(-Rinv B(x_i)^T rho_i)  (ergodic_control.hpp:438-451) ---------------------------------
  // min(max()) when every total of every agent of the wavefront is finite, std::clamp's comparisons otherwise (a NaN
  // passes std::clamp; Rinv's zeros times an infinite co-state make one)
  const R inf = __builtin_huge_val();
  const bool finite = __all((fabs(tot0) < inf) && (fabs(tot1) < inf) && (fabs(tot2) < inf));
  auto update_controls = [&](auto fast_tag) {
    constexpr bool kFast = decltype(fast_tag)::value;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < cnt && agent_ok) {
        const int i = i0 + j;
        R v0, v1, v2;
        if (MODEL == kModelOmni) {  // omni.hpp:205-212
          v0 = cth[j] * r0[j] + sth[j] * r1[j];
          v1 = -sth[j] * r0[j] + cth[j] * r1[j];
          v2 = r2[j];
        } else {  // cart.hpp:194-203
          v0 = cth[j] * r0[j] + sth[j] * r1[j];
          v1 = R(0);
          v2 = r2[j];
        }
        const R n0 = -v0, n1 = -v1, n2 = -v2;
        R u[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const R ur = (p.Rinv[r] * n0 + p.Rinv[r + 3] * n1) + p.Rinv[r + 6] * n2;
          u[r] = kFast ? fmin(fmax(ur, p.umin[r]), p.umax[r]) : clamp_std(ur, p.umin[r], p.umax[r]);
        }
        ut[3 * i + 0] = u[0];
        ut[3 * i + 1] = u[1];
        ut[3 * i + 2] = u[2];
        if (step + 1 < n_steps) {  // wavefront-uniform: hand-over to the next step
          s_next[0 * kLay * kWave + i] = u[0];
          s_next[1 * kLay * kWave + i] = u[1];
          s_next[2 * kLay * kWave + i] = u[2];
        }
        if (STAGES && p.rhot != nullptr) {
          R* const o = p.rhot + 3 * (static_cast<size_t>(T) * b + i);
          o[0] = r0[j];
          o[1] = r1[j];
          o[2] = r2[j];
        }
        if (i == 0) {
          R* const o = p.u0 + 3 * (static_cast<size_t>(step) * p.u0_step_stride + b);
          o[0] = u[0];
          o[1] = u[1];
          o[2] = u[2];
        }
      }
    }
  };
  if (finite) update_controls(std::true_type{});  // wavefront-uniform
  else update_controls(std::false_type{});
  if (step + 1 < n_steps) {  // the next step reads the controls just stored (its own and its neighbour lanes') and reuses the LDS
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    lds_fence();
  }
  }  // step
}

}  // namespace pack
}  // namespace eea
