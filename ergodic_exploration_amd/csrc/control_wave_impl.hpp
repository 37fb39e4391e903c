// Fused receding-horizon ergodic control kernel for gfx950 (MI355X), version 3: ONE WAVEFRONT PER AGENT.
//
// Same computation as control_kernel_impl.hpp -- one complete `ErgodicControl<ModelT>::control` call
// (reference ergodic_control.hpp:224-311 minus configTarget) per agent, for ModelT in {Omni, SimpleCart} --
// and the same three identities (separable basis, RK4 == Simpson as two prefix sums, co-state as two chained
// suffix sums).  What changes is the mapping onto the machine:
//
//   * an agent is one 64-lane wavefront; lane l owns the S = ceil(T / 64) consecutive horizon steps
//     S*l .. S*l + S - 1 (S <= 4, i.e. T <= 256).  No workgroup barrier exists anywhere in the kernel, no
//     cross-wavefront exchange, no partial-sum reduction: the six horizon scans are a serial prefix over the
//     lane's own steps plus ONE DPP wavefront scan of the lane totals (instead of one DPP scan per wavefront
//     and an LDS hop between four of them);
//   * the S steps of a lane are independent instruction streams: the fp64 sin/cos polynomials and the
//     gradient's multiply-add chains of different steps interleave, so one wavefront keeps the vector pipe
//     busy where the one-step-per-lane kernel waited on its own dependency chains;
//   * c_k = (1/N) Cx Cy^T on the matrix cores, fed through a wavefront-private LDS tile of 32 points
//     [point][k] that all 64 lanes stage (one axis per lane, the partner point's cosine over
//     v_permlane32_swap), software-pipelined (the next pass's table recurrence and stores are issued between
//     the matrix instructions): the accumulator of v_mfma_f64_16x16x4_f64 already holds the COMPLETE sums of
//     the agent, so D = lambda (c - phi) is formed in registers and written once for the gradient;
//   * cos(k a), sin(k a) by the Chebyshev three-term recurrences (T_{k+1} = 2c T_k - T_{k-1}, sin(k a) =
//     sin(a) U_{k-1}): one multiply-add per table entry instead of the four of the rotation recurrence;
//   * the gradient makes ONE pass over D per step (rows from LDS, wavefront-uniform broadcast reads: the LDS
//     return path, 4 cycles per 16-byte row read and wavefront, is what bounded a two-pass form) with the
//     cosine array and the G accumulators of ONE step in registers at a time: <= 128 registers, 4 wavefronts
//     per SIMD, 9.2 KB of LDS per agent -- the 4096-agent batch is resident on the 1024 SIMDs in one round.
//
// What bounds it (profiles/r02_ubench_coissue.txt): fp64 matrix instructions run on the
// vector pipe's own multipliers and hold it for their 64 cycles, so the pipe time of an agent is the SUM of its
// vector and matrix instructions; LDS writes cost 13 cycles per 16-byte instruction whatever the lane mask.
//
// Steps beyond 256 and bases beyond K = 16 (other than 20) stay on the workgroup-per-agent kernel
// (control_kernel_impl.hpp); the engine picks.  rollout_only (optTraj / path) stops after the forward half,
// so a rollout and the trajectory a control call reports are bitwise the same function of (pose, controls).
//
#ifndef EEA_CONTROL_WAVE_HELPERS_HPP
#define EEA_CONTROL_WAVE_HELPERS_HPP

#include <type_traits>

#include "common.hpp"

// phase markers: nothing in the product; the A/B library (make AB=1) pre-includes tools/ab/wave_stamps.hpp, which
// makes them shader-clock stamps into p.dbg (tools/phase_timing.py)
#ifndef EEA_WSTAMP
#define EEA_WSTAMP(n) ((void)0)
#define EEA_WSTAMP_RT(n) ((void)0)
#define EEA_WSTAMP_HWID(n) ((void)0)
#endif

namespace eea
{
namespace wave
{
constexpr int kMaxS = 4;        // steps per lane
constexpr int kStageRows = 32;  // points staged per matrix-core pass (8 MFMAs)
constexpr int kRow1 = 68;  // K = 20 fp32 outer-product tile: row stride in elements (= 4 mod 64: consecutive modes 4 banks apart)
constexpr int kTailScratch = 24;  // [3][8]: cos a, cos b, sin b of the <= 8 points of a cooperative last slot

__host__ __device__ constexpr int tab_stride(int K) { return (K + 1) & ~1; }  // even: 16-byte rows
// LDS carve per wavefront, in elements: the region of the contraction's tiles (x / y, 32 rows each), which D and the
// parked barrier gradient take over once the last operand has been read, then the heading park [2][kMaxS][64].  The
// operand reads of the last rows run past their row (modes beyond K, whose products land in accumulator entries
// nobody reads): past the x tile into the y tile, past the y tile into the park -- no pad
__host__ __device__ constexpr int park_elems() { return 2 * kMaxS * kWave; }
__host__ __device__ constexpr int d_elems(int K) { return (K * K + 3) & ~3; }
__host__ __device__ constexpr int tile_elems(int K)
{
  // the tiles of the contraction; afterwards D [K^2], the parked barrier gradient [2][kMaxS][64] and kTailScratch reals
  // of hand-over space for the cooperative last slot (inside the tiles' footprint at K = 10)
  // (K = 20, fp32: the outer-product form's tile [20 modes][kRow1] -- 32 x points, 32 y points, 4 pad -- is the larger)
  const int t = (K == 20) ? 20 * kRow1 : 2 * kStageRows * tab_stride(K);
  const int d = d_elems(K) + 2 * kMaxS * kWave + kTailScratch;
  return ((t > d ? t : d) + 3) & ~3;
}
__host__ __device__ constexpr int wave_lds_elems(int K) { return park_elems() + tile_elems(K); }

// orders this wavefront's own LDS writes before its own LDS reads (DS operations of one wavefront execute in
// order; this stops the compiler from moving them across)
__device__ __forceinline__ void lds_fence()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename R>
__device__ __forceinline__ R read_lane63(R v);
template <>
__device__ __forceinline__ double read_lane63<double>(double v)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ float read_lane63<float>(float v)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// acc + k * v for a small mode number k: fp64 takes the constant from a scalar register pair
template <typename R>
__device__ __forceinline__ R fma_k(R v, int k, R acc)
{
  return acc + static_cast<R>(k) * v;
}
template <>
__device__ __forceinline__ double fma_k<double>(double v, int k, double acc)
{
  double d;
  asm("v_fma_f64 %0, %1, %3, %2" : "=v"(d) : "v"(v), "v"(acc), "s"(static_cast<double>(k)));
  return d;
}

// v_permlane32_swap (gfx950): from_lower = v of lane l - 32 in the upper 32 lanes, from_upper = v of lane l + 32 in the
// lower 32 lanes (probed: the builtin returns { [a.lower, b.lower], [a.upper, b.upper] })
__device__ __forceinline__ void half_swap(double v, double& from_lower, double& from_upper)
{
  const unsigned lo = static_cast<unsigned>(__double2loint(v)), hi = static_cast<unsigned>(__double2hiint(v));
  const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  from_lower = __hiloint2double(static_cast<int>(rh[0]), static_cast<int>(rl[0]));
  from_upper = __hiloint2double(static_cast<int>(rh[1]), static_cast<int>(rl[1]));
}
__device__ __forceinline__ void half_swap(float v, float& from_lower, float& from_upper)
{
  const unsigned u = static_cast<unsigned>(__float_as_int(v));
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  from_lower = __int_as_float(static_cast<int>(r[0]));
  from_upper = __int_as_float(static_cast<int>(r[1]));
}

// v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products D_b = A_b B_b + C_b in 16 cycles (a quarter of the 16x16x4
// instruction's 64).  Operand layout (probed on the MI355X, tools/ubench/mfma4x4.hip): lane l supplies A_b[i][k] and
// B_b[k][j] with k = l / 16, b = (l / 4) % 4, i (j) = l % 4 and holds D_b[i][j] with i = l / 16, b = (l / 4) % 4, j = l % 4.
__device__ __forceinline__ double mfma4(double a, double b, double c)
{
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float mfma4(float, float, float c) { return c; }  // fp32 keeps the 16x16x4 form

// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 outer products D_b += a_b b_b^T in 8 cycles, the same multiply-adds
// per cycle as the 16x16x4 instruction (tools/ubench/mfma4x4x1.hip).  Layout (probed on the MI355X): lane l = 4 b + i
// supplies a_b[i] and b_b[i]; lane 4 b + j holds column j of D_b, D_b[r][j] in register r.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma1(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma1(double, double, f32x4 c) { return c; }  // (fp64 never takes this form)

// LDS byte address of a pointer into the workgroup's shared memory (for DS instructions written by hand)
__device__ __forceinline__ unsigned lds_addr(const void* q)
{
  return static_cast<unsigned>(reinterpret_cast<size_t>((__attribute__((address_space(3))) const void*)q));
}
// The operand reads of the block contraction as 2 NB x 2 plain ds_read_b64 from one address register with immediate
// offsets: q-th 16-row group, g-th 4-mode block, x tile at `ax`, y tile 32 KS reals behind it.  Written by hand because
// the compiler pairs them into ds_read2_b64 / ds_read2st64_b64: that instruction is served in 16-lane groups on 32 banks
// at HALF the rate of two ds_read_b64 (8 instead of 2 x 2 LDS cycles, MI355X_MICROARCH.md "LDS"), and in 16-lane groups
// the rows of this layout (k, k + 4, k + 8, k + 12: 20 dwords apart) collide pairwise: 8 more cycles per instruction --
// SQ_LDS_BANK_CONFLICT 390 per agent, all of it from these reads (profiles/r04_lds_conflicts.txt).
// PAD (elements; the packed kernel): the second 16-row group of a table starts PAD elements later -- its rows then fall on the
// banks the first group's rows leave free for the 16-byte staging stores (control_pack_impl.hpp kGroupPad)
template <int KS, int NB, int Q, int G, int PAD = 0>
struct OperandReads4
{
  static __device__ __forceinline__ void run(unsigned ax, double (&qa)[2][NB], double (&qb)[2][NB])
  {
    constexpr int off = (16 * Q * KS + Q * PAD + 4 * G) * 8;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qa[Q][G]) : "v"(ax), "n"(off) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qb[Q][G]) : "v"(ax), "n"(off + (32 * KS + PAD) * 8) : "memory");
    if constexpr (G + 1 < NB) OperandReads4<KS, NB, Q, G + 1, PAD>::run(ax, qa, qb);
    else if constexpr (Q == 0) OperandReads4<KS, NB, 1, 0, PAD>::run(ax, qa, qb);
  }
};
// the compiler does not count hand-written DS operations: wait for them before the first use (the values are tied to
// the wait so that no use can move above it)
template <int NB>
__device__ __forceinline__ void wait_operands4(double (&qa)[2][NB], double (&qb)[2][NB])
{
  if constexpr (NB == 3) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(qa[0][0]), "+v"(qa[0][1]), "+v"(qa[0][2]), "+v"(qa[1][0]), "+v"(qa[1][1]), "+v"(qa[1][2]),
                   "+v"(qb[0][0]), "+v"(qb[0][1]), "+v"(qb[0][2]), "+v"(qb[1][0]), "+v"(qb[1][1]), "+v"(qb[1][2]));
  } else if constexpr (NB == 2) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(qa[0][0]), "+v"(qa[0][1]), "+v"(qa[1][0]), "+v"(qa[1][1]), "+v"(qb[0][0]), "+v"(qb[0][1]),
                   "+v"(qb[1][0]), "+v"(qb[1][1]));
  } else if constexpr (NB == 5) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(qa[0][0]), "+v"(qa[0][1]), "+v"(qa[0][2]), "+v"(qa[0][3]), "+v"(qa[0][4]), "+v"(qa[1][0]), "+v"(qa[1][1]),
                   "+v"(qa[1][2]), "+v"(qa[1][3]), "+v"(qa[1][4]), "+v"(qb[0][0]), "+v"(qb[0][1]), "+v"(qb[0][2]), "+v"(qb[0][3]),
                   "+v"(qb[0][4]), "+v"(qb[1][0]), "+v"(qb[1][1]), "+v"(qb[1][2]), "+v"(qb[1][3]), "+v"(qb[1][4]));
  } else {
    static_assert(NB == 2 || NB == 3 || NB == 5, "block contraction: K = 5, 10 or 20");
  }
}

// one 16-row group's operands only (the lean packed instances: single-buffered operands, control_pack_impl.hpp)
template <int KS, int NB, int Q, int G, int PAD = 0>
struct OperandReads4One
{
  static __device__ __forceinline__ void run(unsigned ax, double (&qa)[NB], double (&qb)[NB])
  {
    constexpr int off = (16 * Q * KS + Q * PAD + 4 * G) * 8;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qa[G]) : "v"(ax), "n"(off) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(qb[G]) : "v"(ax), "n"(off + (32 * KS + PAD) * 8) : "memory");
    if constexpr (G + 1 < NB) OperandReads4One<KS, NB, Q, G + 1, PAD>::run(ax, qa, qb);
  }
};
template <int NB>
__device__ __forceinline__ void wait_operands4_one(double (&qa)[NB], double (&qb)[NB])
{
  if constexpr (NB == 3) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]));
  } else {
    static_assert(NB == 2, "lean packed instances: K = 5 or 10");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qa[0]), "+v"(qa[1]), "+v"(qb[0]), "+v"(qb[1]));
  }
}

// v + (v rotated right by N lanes inside its row of 16 lanes)
template <int N>
__device__ __forceinline__ double add_row_ror(double v)
{
  return v + dpp_or_zero<0x120 + N, 0xf>(v);  // row_ror:N, every lane has a source (bound_ctrl: no "old" operand to set up)
}

template <typename R, int MODEL>
__device__ __forceinline__ void model_xy(R vx, R vy, R c, R s, R& fx, R& fy)
{
  if (MODEL == kModelOmni) {  // omni.hpp:177-181
    fx = vx * c - vy * s;
    fy = vx * s + vy * c;
  } else {  // cart.hpp:172
    fx = vx * c;
    fy = vx * s;
  }
}

template <typename R>
__device__ __forceinline__ R wrap_pi_fast(R rad)
{
  // normalize_angle_PI (numerics.hpp:78-90) with the quotient taken by a multiplication; a quotient off by
  // one at an exact multiple is repaired by the two range fixes
  const R pi = static_cast<R>(kPi), two_pi = static_cast<R>(2.0 * kPi);
  const R q = floor((rad + pi) * static_cast<R>(1.0 / (2.0 * kPi)));
  rad = (rad + pi) - q * two_pi;
  if (rad < R(0)) rad += two_pi;
  if (rad >= two_pi) rad -= two_pi;
  return rad - pi;
}

// waves per SIMD the kernel is compiled for: 4 (128 registers) for K <= 10 and for fp32; fp64 K = 20: 2 (the block
// contraction's 25 accumulators + the gradient's 20-mode families: 233 registers, no scratch; compiled for 3 it spills
// 264 B per lane and is 1.5 % slower)
constexpr int waves_per_simd(int KC, int real_size)
{
  return (KC == 20) ? (real_size == 8 ? 2 : 4) : ((KC <= 10 || real_size == 4) ? 4 : 3);
}
// k1 block of the gradient: cosine and G arrays of this many modes are in registers at a time
// K = 20: ALL 20 modes at once in both precisions -- in two blocks of 10 the y recurrences and the weighted sums of H run
// once per block, and the pass is 6 % (fp32: 37.3 -> 34.9 us, 128 registers + 16-24 B of scratch per lane; compiled for 3
// wavefronts per SIMD without scratch: 35.5) / 17 % (fp64: 87.6 -> 72.6 us) slower than the instruction counts say
// (the K <= 16 instance likewise: all 16 modes at once, 150-157 registers in fp64 = 3 wavefronts per SIMD, still 7-9 % faster
// than two blocks of 8 at 4 wavefronts per SIMD: K = 16 69.5 -> 64.1 us, K = 12 54.0 -> 50.0, fp32 K = 16 39.3 -> 35.8)
constexpr int grad_block(int KC) { return KC; }

}  // namespace wave
}  // namespace eea
#endif  // EEA_CONTROL_WAVE_HELPERS_HPP

// ---- the kernel --------------------------------------------------------------------------------------------------------
namespace eea
{
namespace wave
{
// KC: compile-time K (5, 10, 20) or 16 = any K <= 16 at run time (loops unrolled to 16, guarded).
// STAGES: the optional per-stage outputs (traj, edx, bdx, rhot) are compiled in.
// WPB: wavefronts (= agents) per workgroup; they share nothing.
// Registers: the fp64 K <= 10 instances without stage outputs -- what bench.py times -- compile to 116-119 registers,
// no scratch (tests/test_capi_symbols.py reads it from the code object): the allocation granule is 8, so four wavefronts
// per SIMD leave 32 registers free, room for the wavefronts of the record sum (control_kernel.hip) BESIDE a fully
// resident control kernel.  (Rounds 3-4 needed a second, register-capped compilation of this text with two values parked
// in LDS to get there; since the lane id and the launch arguments are re-derived per step it fits by itself.)
//
// RESIDENT (EEA_OPT_RESIDENT_CONTROL, horizons of one slot): the step loop never ends by itself -- every step is one request
// from the host-mapped mailbox p.res_mail (ResidentMail<R>: the poll fetches the 64-byte request line, pose / map position /
// column count come out of it by v_readlane, nothing of the request is read from memory again), u0 and the request's number go
// back to the mailbox; the controls stay in LDS from request to request as between the steps of a multi-step launch.
template <typename R, int MODEL, int KC, bool STAGES, int WPB, bool RESIDENT = false>
__global__ __launch_bounds__(WPB* kWave, waves_per_simd(KC, sizeof(R))) void control_wave_kernel(
    const ControlParams<R> p_arg, const unsigned B, const int S_arg, const int rollout_arg)
{
  (void)p_arg;  // read through the kernel-argument segment below
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // Receding-horizon steps of this agent in ONE launch (eea_control_batch_steps): step n is a complete control() call
  // from the pose of row n and the controls step n - 1 left in ut -- the wavefront reads its own stores back (from L2:
  // the fences at the end of the loop body) instead of the host launching again.  n_steps == 1: eea_control_batch.
  // Every step starts from the thread id alone, through a value the optimiser cannot see through: nothing lane- or
  // agent-derived (step maps, LDS addresses, masks, pointers) is carried across the loop -- hoisted, those invariants
  // cost the body ~460 B of scratch per lane; recomputed they cost what they cost a launch.
  typedef const __attribute__((address_space(4))) ControlParams<R> KernArgParams;
  const int wave_of_block = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int n_steps = RESIDENT ? 0x7fffffff : ((KernArgParams*)__builtin_amdgcn_kernarg_segment_ptr())->n_steps;
  unsigned res_last = RESIDENT ? ((KernArgParams*)__builtin_amdgcn_kernarg_segment_ptr())->res_first : 0u;  // request seen
  bool res_handover = false;  // RESIDENT: a served request has left its controls in LDS
  bool res_leaving = false;   // RESIDENT: alive = 0 has been announced -- the request being served is the last one
  // The controls a step leaves for the next one are handed over in LDS (the tiles and the park are dead between the
  // update and the next forward half): s_next[r][step index], read back one column to the right
  // (ergodic_control.hpp:233-234).  Reading them back from L2 put that latency at the head of every step of every wavefront
  // (all wait at once: nothing to overlap with) -- 8 % of a 50-step launch; carried in registers they pushed the metric
  // instance over its 128.  (ut is still stored every step: the backward half re-reads the controls it needs from there.)
  for (int step = 0; step < n_steps; ++step) {
  // ... and the launch parameters are re-read (scalar loads where they are used, as in a launch) through a pointer that
  // is opaque per step: hoisted out of the loop they would all be live through the body (106 scalar registers + spills)
  KernArgParams* ka = (KernArgParams*)__builtin_amdgcn_kernarg_segment_ptr();  // p_arg is argument 0
  asm volatile("" : "+s"(ka));
  KernArgParams& p = *ka;
  // (the lane id comes from the hardware every step -- v_mbcnt, through volatile asm so that it is neither hoisted nor kept
  // alive across the body -- and the wavefront's index from a scalar register: no vector register of the thread id
  // survives the step)
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  int wv = wave_of_block;
  asm volatile("" : "+s"(wv));
  // (the same for the two scalar launch arguments: hoisted out of the step loop, the ~20 wavefront-uniform predicates of
  // the unrolled slot loops -- S > j, S == 4, ... -- were parked in the lanes of a vector register and read back with
  // v_readlane every step; recomputed they are scalar compares)
  int S = S_arg, rollout_only = rollout_arg;
  asm volatile("" : "+s"(S), "+s"(rollout_only));
  // RESIDENT: wait for the next request (lanes 0..15 fetch the request line, one dword each: one PCIe read that brings a new
  // request number together with its data)
  R res_pose[3] = { R(0), R(0), R(0) }, res_map_x = R(0), res_map_y = R(0);
  int res_nmem = 0;
  if constexpr (RESIDENT) {
    using Mail = ResidentMail<R>;
    constexpr int kReq = offsetof(Mail, req) / 4, kCmd = offsetof(Mail, cmd) / 4;
    constexpr int kNmem = offsetof(Mail, n_mem) / 4, kMapX = offsetof(Mail, map_x) / 4;
    Mail* const mail = static_cast<Mail*>(p.res_mail);
    const unsigned* const line = reinterpret_cast<const unsigned*>(mail);
    const long long t0 = wall_clock64();
    unsigned v, r;
    bool idle = false;
    for (;;) {
      v = __hip_atomic_load(line + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      r = __builtin_amdgcn_readlane(v, kReq);
      if (r != res_last) break;
      if (wall_clock64() - t0 > p.res_idle) {
        idle = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (idle) {
      // idle for too long: say so FIRST, then look once more -- a request posted before the host can have seen alive = 0 is
      // still served, one posted later finds alive = 0 and launches again (engine.cpp control_resident)
      if (lane == 0) __hip_atomic_store(&mail->alive, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_s_sleep(64);
      v = __hip_atomic_load(line + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      r = __builtin_amdgcn_readlane(v, kReq);
      if (r == res_last) return;
      // a request that landed in that window is served -- and then the wavefront LEAVES: it has told the host
      // that it is gone, so it must not go back to polling (the host may already be launching its successor, which would
      // serve the same request a second time and shift the warm start twice)
      res_leaving = true;
    }
    res_last = r;
    if (__builtin_amdgcn_readlane(v, kCmd) != 0u) {  // told to leave
      if (lane == 0) {
        __hip_atomic_store(&mail->alive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mail->done, static_cast<int>(r), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
    if constexpr (sizeof(R) == 8) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        res_pose[i] = static_cast<R>(__hiloint2double(static_cast<int>(__builtin_amdgcn_readlane(v, 2 * i + 1)),
                                                      static_cast<int>(__builtin_amdgcn_readlane(v, 2 * i))));
      }
      res_map_x = static_cast<R>(__hiloint2double(static_cast<int>(__builtin_amdgcn_readlane(v, kMapX + 1)),
                                                  static_cast<int>(__builtin_amdgcn_readlane(v, kMapX))));
      res_map_y = static_cast<R>(__hiloint2double(static_cast<int>(__builtin_amdgcn_readlane(v, kMapX + 3)),
                                                  static_cast<int>(__builtin_amdgcn_readlane(v, kMapX + 2))));
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) res_pose[i] = static_cast<R>(__int_as_float(static_cast<int>(__builtin_amdgcn_readlane(v, i))));
      res_map_x = static_cast<R>(__int_as_float(static_cast<int>(__builtin_amdgcn_readlane(v, kMapX))));
      res_map_y = static_cast<R>(__int_as_float(static_cast<int>(__builtin_amdgcn_readlane(v, kMapX + 1))));
    }
    res_nmem = static_cast<int>(__builtin_amdgcn_readlane(v, kNmem));
    // the replay-memory columns the host wrote for this request are behind this CU's L1 (an earlier request's loads left
    // their cache lines there)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    if (step > (1 << 30)) step = 1;  // (the step index only tells the first request from the later ones)
  }
  const R map_x = RESIDENT ? res_map_x : p.map_x, map_y = RESIDENT ? res_map_y : p.map_y;
  const unsigned b = blockIdx.x * WPB + wv;
  if (b >= B) return;  // wavefront-uniform
  // eea_batch_io::d_skip (the fleet tick: a robot that follows a dynamic-window twist does not call control(),
  // exploration.hpp:230-236): nothing of the agent is read or written
  if (p.skip != nullptr && p.skip[b] != 0) return;
  EEA_WSTAMP_RT(10);
  EEA_WSTAMP_HWID(12);

  const int T = p.T;
  // kGenericK: any K <= 16 at run time (modes beyond K are masked to zero).  kRowGuard: the rows of the gradient
  // (and the table stores) are guarded by a comparison with the run-time K also for K = 20, where the guard is
  // always true: without it the compiler flattens the fully static loops into one block, hoists the LDS loads
  // of all rows above the arithmetic and spills them (1.4 KB of scratch per lane in fp32)
  constexpr bool kGenericK = (KC == 16);
  constexpr bool kRowGuard = (KC == 16 || KC == 20);
  const int K = kRowGuard ? p.K : KC;
  const int K2 = K * K;
  constexpr int KS = (KC == 16) ? 16 : tab_stride(KC);

  R* const sm = reinterpret_cast<R*>(smem_raw) + static_cast<size_t>(wv) * wave_lds_elems(KC == 16 ? 16 : KC);
  R* const tabx = sm;                       // [32 rows][KS]
  R* const taby = tabx + kStageRows * KS;
  R* const s_cp = sm + tile_elems(KC == 16 ? 16 : KC);  // cos of the post-step heading, [j][lane]
  R* const s_sp = s_cp + kMaxS * kWave;                 // sin
  R* const s_D = tabx;                      // D[k2 * K + k1]   (after the contraction)
  R* const s_g = tabx + d_elems(KC == 16 ? 16 : KC);  // barrier gradient parked during the gradient, [2 j + r][lane]

  // Lane -> horizon steps.  Lanes [0, q16) own S consecutive steps each, the following lanes sc < S steps each, the rest
  // none.  Lane order is step order (the scans need nothing else), and the valid lanes of a slot j are whole groups of 16
  // plus at most one partial group -- the contraction multiplies groups of 16 points.  Two shapes:
  //  * T = 64 (S - 1) + r with r <= 8 (T = 200: S = 4, r = 8): the first r lanes own S steps, all others S - 1 -- the
  //    slots 0 .. S-2 are full wavefronts and the last slot holds r steps in lanes 0 .. r-1, whose gradient is then
  //    taken by all 64 lanes together (8 lanes per step, below) instead of a full pass at r / 64 lanes;
  //  * otherwise: q16 = the number of lanes that can own S steps rounded DOWN to a multiple of 16, what is left of the
  //    horizon (< 16 S steps) spread over the next <= 16 lanes, sc = ceil(rem / 16) steps each.
  const int r_top = T - kWave * (S - 1);  // 1 .. 64
  const bool top_heavy = S > 1 && r_top <= 8;  // wavefront-uniform
  const int q16 = top_heavy ? r_top : ((T / S) & ~15);
  const int rem = T - S * q16;
  const int sc = top_heavy ? S - 1 : ((rem + 15) >> 4);
  const int i0 = lane < q16 ? S * lane : S * q16 + sc * (lane - q16);  // first horizon step of this lane
  const int cnt = lane < q16 ? S : max(0, min(sc, T - i0));            // steps of this lane
  // number of lanes that own a step in slot j (wave-uniform)
  auto lanes_in_slot = [&](int j) { return q16 + (j < sc ? (rem - j + sc - 1) / sc : 0); };
  // steps of another lane
  auto cnt_of = [&](int l) { return l < q16 ? S : max(0, min(sc, rem - sc * (l - q16))); };
  R* const ut = p.ut + 3 * static_cast<size_t>(T) * b;
  const R* const pose = p.pose + 3 * (static_cast<size_t>(step) * p.pose_step_stride + b);
  // device-bound exchange: records out / shared c_k in, with their sequence numbers (the same for every step of a
  // multi-step launch: a step must never wait for an exchange the host enqueues AFTER this launch -- that would need
  // truly concurrent hardware queues)
  R* const ck_rec_step = p.ck_rec;
  const R* const ck_shared_step = p.ck_shared;
  const unsigned rec_seq_step = p.rec_seq;
  const unsigned flag_seq_step = p.ck_flag_seq;
  EEA_WSTAMP(0);
  // ---- controls: shift left by one column, last column zero (ergodic_control.hpp:233-234) ------------
  R vx[kMaxS], vy[kMaxS], w[kMaxS];
  bool bad = false;
  if (RESIDENT ? !res_handover : step == 0) {  // wavefront-uniform
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      vx[j] = vy[j] = w[j] = R(0);
      if (j < S) {
        const int src = rollout_only ? i0 + j : i0 + j + 1;  // optTraj rolls the controls out as they are
        if (j < cnt && src < T) {
          vx[j] = ut[3 * src + 0];
          vy[j] = ut[3 * src + 1];
          w[j] = ut[3 * src + 2];
        }
        // SimpleCart::operator() rejects a lateral velocity (cart.hpp:167-170)
        if (MODEL == kModelSimpleCart && j < cnt && !(fabs(vy[j]) < R(1.0e-12))) bad = true;
      }
    }
  } else {  // the kernel's own output of the step before
    const R* const s_next = sm + i0 + 1;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      vx[j] = vy[j] = w[j] = R(0);
      if (j < S) {
        if (j < cnt && i0 + j + 1 < T) {
          vx[j] = s_next[0 * kMaxS * kWave + j];
          vy[j] = s_next[1 * kMaxS * kWave + j];
          w[j] = s_next[2 * kMaxS * kWave + j];
        }
        if (MODEL == kModelSimpleCart && j < cnt && !(fabs(vy[j]) < R(1.0e-12))) bad = true;
      }
    }
    lds_fence();  // (read before the forward half writes the park and the tiles)
  }
  const R x0 = RESIDENT ? res_pose[0] : pose[0], y0 = RESIDENT ? res_pose[1] : pose[1], th0 = RESIDENT ? res_pose[2] : pose[2];
  // record elements per lane of an agent's sum record (K^2 + 1 reals rounded up to even)
  constexpr int kRecRounds = (ck_record_len((KC == 16 ? 16 : KC) * (KC == 16 ? 16 : KC)) + kWave - 1) / kWave;
  if (__any(bad)) {
    // the reference throws out of rk4_.solve; nothing else of this agent is touched
    if (ck_rec_step != nullptr && !rollout_only) {  // it does not count in the sum of the records: all-zero record
#pragma unroll
      for (int r = 0; r < kRecRounds; ++r) {
        const int e = kWave * r + lane;
        if (e < p.rec_len) store_agent(ck_rec_step + static_cast<size_t>(b) * p.rec_len + e, R(0));
      }
      if (p.rec_ready != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) store_agent(p.rec_ready + b, rec_seq_step);
      }
    }
    if (lane == 0 && p.status != nullptr) p.status[b] = 2;  // EEA_ERR_INVALID_TWIST
    if (lane == 0 && p.done != nullptr && b == 0) {
      __hip_atomic_store(p.done, RESIDENT ? static_cast<int>(res_last) : p.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if constexpr (RESIDENT) {
      if (res_leaving) return;
      continue;  // (nothing was written: the controls in ut / LDS are what they were)
    }
    return;
  }
  // (a time-out of an earlier step of the launch stays; device-bound exchange: so does one an earlier PASS left in a reused
  // buffer, until the caller clears it)
  // (RESIDENT: the host clears the mailbox's status with the request)
  if (!RESIDENT && lane == 0 && p.status != nullptr && step == 0 && !(p.ck_flag != nullptr && p.status[b] == 6)) p.status[b] = 0;
  EEA_WSTAMP(1);

  const R dt = p.dt, dt6 = p.dt6;
  const R inv_pi = static_cast<R>(1.0 / kPi);

  // ================= forward half ========================================================================
  // heading: theta_i = wrap(theta_{i-1} + dt/6 (w + 2w + 2w + w)) (integrator.hpp:146-148,183)
  //          == wrap(theta_0 + prefix sum) up to rounding
  R thp[kMaxS];  // inclusive prefix of the heading increments within the lane
  {
    R run = R(0);
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      const R d = dt6 * (((w[j] + R(2) * w[j]) + R(2) * w[j]) + w[j]);
      run += d;  // zero controls (d = 0) in the slots beyond S and beyond the horizon
      thp[j] = run;
    }
    const R incl = wave_inclusive_scan_dpp(run);
    const R base = wrap_pi_fast(th0) + (incl - run);  // heading before the lane's first step
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) thp[j] += base;   // heading after step j
  }

  EEA_WSTAMP(2);
  // position: x_i = x_{i-1} + dt/6 (k1 + 2 k2 + 2 k3 + k4) with k2 == k3 (integrator.hpp:176-184).
  // sin/cos of the pre-step heading is evaluated for the lane's first step only; the later ones are the
  // previous step's post-step values (2 mid - pre by the double-angle / addition formulas, ~4e-16 each)
  R px[kMaxS], py[kMaxS];  // inclusive prefix of the position increments within the lane
  R incx[kMaxS], incy[kMaxS];  // the increments themselves (basis angles of later steps by rotation, below)
  {
    R c, s;
    {
      const R th_pre0 = thp[0] - dt6 * (((w[0] + R(2) * w[0]) + R(2) * w[0]) + w[0]);
      sincospi_r(th_pre0 * inv_pi, &s, &c);
    }
    // Small yaw increments (|dt w / 2| <= pi / 16 for every step of the wavefront -- any realistic time step):
    // the mid-stage and post-step headings follow from the pre-step one by two rotations with sin/cos of the
    // increment (short series, no argument reduction) instead of a full evaluation + double-angle formulas
    bool small = true;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) small = small && (fabs(dt * (R(0.5) * w[j]) * inv_pi) <= R(0.0625));
    const bool fast_h = sizeof(R) == 8 && __all(small);
    R rx = R(0), ry = R(0);
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      incx[j] = incy[j] = R(0);
      if (j < S) {
        // mid stage theta + dt (0.5 w), shared by k2 and k3 (integrator.hpp:179-180)
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
        model_xy<R, MODEL>(vx[j], vy[j], c, s, k1x, k1y);
        model_xy<R, MODEL>(vx[j], vy[j], cm, sm_, k2x, k2y);
        model_xy<R, MODEL>(vx[j], vy[j], cpost, spost, k4x, k4y);
        // steps beyond the horizon carry zero controls (the loads are guarded), so their increments are exact
        // zeros: no select
        incx[j] = dt6 * (((k1x + R(2) * k2x) + R(2) * k2x) + k4x);
        incy[j] = dt6 * (((k1y + R(2) * k2y) + R(2) * k2y) + k4y);
        rx += incx[j];
        ry += incy[j];
        // parked for the backward half: heading after step j (A = fdx(x_j, u_j), B = fdu(x_j))
        s_cp[j * kWave + lane] = cpost;
        s_sp[j * kWave + lane] = spost;
        c = cpost;
        s = spost;
      }
      px[j] = rx;
      py[j] = ry;
    }
    const R ix = wave_inclusive_scan_dpp(rx), iy = wave_inclusive_scan_dpp(ry);
    const R bx = x0 + (ix - rx), by = y0 + (iy - ry);
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      px[j] += bx;
      py[j] += by;
    }
  }
  EEA_WSTAMP(3);
  if (STAGES && p.traj != nullptr) {
    R* const traj = p.traj + 3 * static_cast<size_t>(T) * b;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < cnt) {
        traj[3 * (i0 + j) + 0] = px[j];
        traj[3 * (i0 + j) + 1] = py[j];
        traj[3 * (i0 + j) + 2] = wrap_pi_fast(thp[j]);
      }
    }
  }

  if (STAGES && rollout_only) return;  // optTraj / path (ergodic_control.hpp:313-342): the rollout is all

  // basis angles of the rollout points (map frame -> Fourier frame, ergodic_control.hpp:243-244; one sin/cos
  // pair per axis: angle = pi x / lx, basis.cpp:85 with k = 1) and the barrier gradient (:453-474)
  R c1x[kMaxS], s1x[kMaxS], c1y[kMaxS], s1y[kMaxS], g0[kMaxS], g1[kMaxS];
  // The lane's first point: full evaluation.  Its later points, when every step of the wavefront moves by at most
  // 1/16 of the domain (|dx| <= lx / 16: 0.75 m per step on the 12 m map): rotation of the previous point's
  // sin/cos by the step's increment (short series; ~3e-16 per rotation, chains of at most 3).
  bool small_b = true;
#pragma unroll
  for (int j = 1; j < kMaxS; ++j) {
    small_b = small_b && (fabs(incx[j] * p.inv_lx) <= R(0.0625)) && (fabs(incy[j] * p.inv_ly) <= R(0.0625));
  }
  const bool fast_b = sizeof(R) == 8 && __all(small_b);
#pragma unroll
  for (int j = 0; j < kMaxS; ++j) {
    c1x[j] = s1x[j] = c1y[j] = s1y[j] = g0[j] = g1[j] = R(0);
    if (j < S) {
      const R x = px[j] - map_x, y = py[j] - map_y;
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
      // gradBarrier (:453-474): 25 * (2 [x > lx - eps] (x - (lx - eps)) + 2 [x < eps] (x - eps)) per axis.  The
      // indicator times the difference is max(difference, 0) / min(difference, 0) (x > a <=> x - a > 0 in IEEE
      // arithmetic), at most one of the two is non-zero on any domain wider than 2 eps, and 2 * 25 is exact: the same
      // value from 6 instead of ~11 instructions per axis
      const R eps = R(0.05), weight2 = R(50);
      g0[j] = (fmax(x - (p.lx - eps), R(0)) + fmin(x - eps, R(0))) * weight2;
      g1[j] = (fmax(y - (p.ly - eps), R(0)) + fmin(y - eps, R(0))) * weight2;
    }
  }

  EEA_WSTAMP(4);
  // ---- c_k = (1/N) sum_p cos(a_k1 x_p) cos(b_k2 y_p)  (basis.cpp:109-120) on the matrix cores -----------
  // sampled past states are prepended (buffer.cpp:78-108) and shifted like the rollout
  int nmem = 0;
  if (p.mem_cols != nullptr) {
    nmem = RESIDENT ? res_nmem : (p.n_mem != nullptr) ? p.n_mem[b] : static_cast<int>(p.mem_stride);
    nmem = nmem < 0 ? 0 : (nmem > static_cast<int>(p.mem_stride) ? static_cast<int>(p.mem_stride) : nmem);
  }
  const int N = T + nmem;
  using M = Mfma<R>;
  using acc_t = typename M::acc_t;
  constexpr int NT = (KC > 16) ? 2 : 1;  // 16-mode tiles per axis
  acc_t acc0 = acc_t{ R(0), R(0), R(0), R(0) }, acc1 = acc0;  // NT = 1: two chains on the one tile
  acc_t acc2 = acc0, acc3 = acc0;                              // NT = 2: tiles (0,0) (0,1) (1,0) (1,1)
  const int mk = lane >> 4, mi = lane & 15;  // matrix-instruction operand coordinates of this lane
  // lambda_k, phi_k of this lane's accumulator entries (mode = k2 * K + k1, k2 = mi, k1 = the accumulator row
  // of register r).  Loaded after the contraction: issued earlier (before it, or during its last pass) they
  // hold 16 registers through the pipelined passes and cost 7 % of the launch (profiles/r02_ablation.txt)
  R lam[NT * NT][4], phi[NT * NT][4];
  auto load_lam_phi = [&]() {
#pragma unroll
    for (int t = 0; t < NT * NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k1 = 16 * (t / NT) + M::row(lane, r), k2 = 16 * (t % NT) + mi;
        const bool ok = k1 < K && k2 < K;
        lam[t][r] = ok ? p.lamdak[k2 * K + k1] : R(0);
        phi[t][r] = ok ? p.phik[k2 * K + k1] : R(0);
      }
    }
  };

  // Tiles: 32 rows [point][k] per axis.  One pass of up to 8 matrix instructions consumes the 32 points of one half
  // of the wavefront (lanes [32 h, 32 h + 32); zero rows for invalid points).
  // cos(k a) = T_k(cos a) by the Chebyshev recurrence, two entries (one 16-byte store) per step.
  // Staging, one axis per lane: the tile of a pass is written by ALL 64 lanes -- for the points of lanes 0..31 the
  // lower lanes run the x recurrence of their own point and the upper lanes the y recurrence of the point of lane
  // l - 32 (its cos b comes over v_permlane32_swap), for the points of lanes 32..63 the other way round.  An LDS
  // write costs its 6 / 13 cycles (b64 / b128) per instruction whatever the number of active lanes
  // (profiles/r02_ubench_coissue.txt), and one LDS serves the four SIMDs: per pass one full store per mode pair
  // instead of two half-masked ones, and 2 instead of 4 multiply-adds.
  const bool lo = lane < 32;
  const int row = lane & 31;
  R* const st_lower = (lo ? tabx : taby) + row * KS;  // staging the tile of the points of lanes 0..31
  R* const st_upper = (lo ? taby : tabx) + row * KS;  // ... of lanes 32..63
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
  auto tab1_store = [&](const Tab1& t, R* dst, int k) {  // entries k, k + 1 of the row
    if (sizeof(R) == 8) {
      *reinterpret_cast<double2*>(dst + k) = double2{ static_cast<double>(t.a), static_cast<double>(t.b) };
    } else {
      *reinterpret_cast<float2*>(dst + k) = float2{ static_cast<float>(t.a), static_cast<float>(t.b) };
    }
  };
  auto tab1_step = [&](Tab1& t) {
    const R c = t.two * t.b - t.a, d = t.two * c - t.b;
    t.a = c;
    t.b = d;
  };
  // cos of this lane's axis for the two tiles of 64 points (ca, cb: cos of the x / y angle of this lane's point)
  auto stage_cos = [&](R ca, R cb, R& c_lower, R& c_upper) {
    R from_lower, from_upper;  // cb of lane l - 32 (valid in the upper lanes), of lane l + 32 (in the lower ones)
    half_swap(cb, from_lower, from_upper);
    c_lower = lo ? ca : from_lower;
    c_upper = lo ? from_upper : ca;
  };
  constexpr int kPairs = KS / 2;  // 16-byte stores per row
  // plain (not software-pipelined) pass over the points of half h of the wavefront, of which the first nl_valid
  // lanes of the wavefront hold a valid point: the replay-memory columns and the generic-K / K = 20 instances
  auto stage_and_mma = [&](R c_axis, int h, int nl_valid) {
    {
      Tab1 t = tab1_init(c_axis, 32 * h + row < nl_valid);
      R* const dst = h ? st_upper : st_lower;
#pragma unroll
      for (int q = 0; q < kPairs; ++q) {
        if (!kRowGuard || 2 * q < K) {
          tab1_store(t, dst, 2 * q);
          tab1_step(t);
        }
      }
    }
    const int rows_valid = nl_valid - 32 * h;  // > 0 (the caller skips empty halves)
    lds_fence();
    if (NT == 1) {
#pragma unroll
      for (int m = 0; m < kStageRows / 4; m += 2) {
        if (4 * m < rows_valid) {  // wavefront-uniform
          const int off = (4 * m + mk) * KS + mi;
          acc0 = M::run(tabx[off], taby[off], acc0);
        }
        if (4 * (m + 1) < rows_valid) {
          const int off = (4 * (m + 1) + mk) * KS + mi;
          acc1 = M::run(tabx[off], taby[off], acc1);
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < kStageRows / 4; ++m) {
        if (4 * m < rows_valid) {  // wavefront-uniform
          const int off = (4 * m + mk) * KS + mi;
          const R a0 = tabx[off], a1 = tabx[off + 16], b0 = taby[off], b1 = taby[off + 16];
          acc0 = M::run(a0, b0, acc0);
          acc1 = M::run(a0, b1, acc1);
          acc2 = M::run(a1, b0, acc2);
          acc3 = M::run(a1, b1, acc3);
        }
      }
    }
    lds_fence();
  };

  // ---- fp64, K = 5 / 10: the contraction in 4x4 blocks ----------------------------------------------------------
  // fp64 matrix instructions run on the vector pipe's own multipliers (profiles/r02_ubench_coissue.txt): the 16x16 tile
  // of the form above -- 100 useful entries of 256 at K = 10, 64 cycles per 4 points -- is a quarter of the agent's
  // pipe time.  v_mfma_f64_4x4x4_4b multiplies four independent 4x4x4 blocks in 16 cycles.  The four blocks of an
  // instruction are four GROUPS OF 4 POINTS (16 rows of the tile) for ONE pair (I, J) of 4-mode blocks: the A operand
  // of block I serves the NB instructions (I, 0..NB-1) and the B operand of block J the NB instructions (0..NB-1, J),
  // so a group of 16 points costs 2 NB operand reads and NB^2 instructions: K = 10: 6 reads + 9 x 16 = 144 cycles per 16
  // points instead of 8 reads + 4 x 64 = 256.  Every accumulator holds four partial sums (one per point group), added
  // at the end by two row rotations.  (Round 2's block form gave the four blocks four different (I, J) pairs of the
  // same 4 points: every instruction then needs its own operands -- 288 LDS reads per agent -- and the phase became
  // LDS-bound, tools/ab/block4_contraction.patch.)
  constexpr bool kBlock4 = sizeof(R) == 8 && (KC == 10 || KC == 5 || KC == 20);
  constexpr int NB = (KC + 3) / 4;          // 4-mode blocks per axis
  R cacc[kBlock4 ? NB : 1][kBlock4 ? NB : 1];
#pragma unroll
  for (int I = 0; I < (kBlock4 ? NB : 1); ++I) {
#pragma unroll
    for (int J = 0; J < (kBlock4 ? NB : 1); ++J) cacc[I][J] = R(0);
  }
  // operand coordinates of this lane: row of the 16-row group, mode inside the 4-mode block (modes past K read the
  // next row's first entries -- finite values whose products land in accumulator entries nobody reads).
  // Which of the group's 16 points is (block b, k) of the instruction is free -- sums over points -- as long as A and B
  // agree.  ds_read_b64 is served in two lane groups {0..31}, {32..63} on 64 banks of 4 bytes: a lane group reads 8 rows x
  // 4 modes = 64 dwords, conflict-free iff the rows' windows of 8 dwords tile the banks.  Rows lie 2 KS = 20 dwords apart
  // (K = 10), i.e. at multiples of 4 dwords: EVEN rows (40 m mod 64 = all multiples of 8) for lanes 0..31, odd rows for
  // lanes 32..63.  Round 3's map (row = 4 b + k: rows {0,1,4,5,...} in one lane group) put every window on half of
  // another one's banks: 2 extra LDS cycles per operand read (SQ_LDS_BANK_CONFLICT 390 per agent, profiles/r04_lds_conflicts.txt)
  // K = 20: rows 40 dwords apart, 8 consecutive rows tile the 64 banks: rows 0..7 for lanes 0..31, 8..15 for lanes 32..63.
  const int orow = (KC == 20) ? 8 * (lane >> 5) + 4 * ((lane >> 4) & 1) + ((lane >> 2) & 3)
                              : 4 * ((lane >> 2) & 3) + 2 * ((lane >> 4) & 1) + (lane >> 5);
  const int oi = lane & 3;
  R qa[2][kBlock4 ? NB : 1], qb[2][kBlock4 ? NB : 1];
  const unsigned oaddr = lds_addr(tabx + orow * KS + oi);
  auto read_operands4 = [&]() {  // both 16-row groups of the tile: issued here, waited for by operands4_ready()
    if constexpr (kBlock4) OperandReads4<KS, NB, 0, 0>::run(oaddr, qa, qb);
  };
  auto operands4_ready = [&]() {
    if constexpr (kBlock4) wait_operands4<NB>(qa, qb);
  };
  auto mma4_group = [&](int q) {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
      for (int J = 0; J < NB; ++J) cacc[I][J] = mfma4(qa[q][I], qb[q][J], cacc[I][J]);
    }
  };
  // plain (not software-pipelined) pass of the block form: replay-memory columns
  auto stage_and_mma4 = [&](R c_axis, int h, int nl_valid) {
    {
      Tab1 t = tab1_init(c_axis, 32 * h + row < nl_valid);
      R* const dst = h ? st_upper : st_lower;
#pragma unroll
      for (int q = 0; q < kPairs; ++q) {
        tab1_store(t, dst, 2 * q);
        tab1_step(t);
      }
    }
    const int rows_valid = nl_valid - 32 * h;  // > 0
    lds_fence();
    read_operands4();
    lds_fence();
    operands4_ready();
    mma4_group(0);
    if (rows_valid > 16) mma4_group(1);  // wavefront-uniform
  };

  // ---- fp32, K = 20: the contraction as 4x4 outer products --------------------------------------------------
  // Two 16-mode tiles per axis multiply 1024 accumulator entries for the 400 modes of K = 20 (39 %): 256 instructions of
  // 32 cycles = a third of the agent's pipe time (profiles/r04_k20_f32_isa_budget.txt).  v_mfma_f32_4x4x1_16b multiplies
  // sixteen independent 4x4 blocks for ONE point in 8 cycles: the 25 pairs (I, J) of 4-mode blocks are the blocks of TWO
  // instructions per point (25 of 32 blocks in use, 78 %) -- 512 x 8 cycles, half the pipe time, and every accumulator
  // entry is a complete sum over the points: no rotation at the end.  Block b = 3 I + s of both instructions has the x
  // modes 4 I .. 4 I + 3 (ONE A operand for both), and the y blocks J = s (first instruction) and J = 4 - s (second;
  // s = 2: none, its second block and block 15 multiply something finite that nobody reads).
  // Tile: [mode][kRow1] with 32 x points, then 32 y points per row, so that one ds_read_b128 brings a lane's mode for
  // FOUR points: 3 reads per 8 instructions.  kRow1 = 4 mod 64 puts consecutive modes 4 banks apart; modes m and
  // m + 16 share banks, and no group of 16 lanes reads both: a group holds <= 2 consecutive x blocks, y blocks {0,1,2}
  // in the first and {4,3} in the second instruction.
  constexpr bool kBlock1 = sizeof(R) == 4 && KC == 20;
  f32x4 bacc1 = f32x4{ 0, 0, 0, 0 }, bacc2 = bacc1;
  const int b1blk = (lane >> 2) < 14 ? (lane >> 2) : 14, b1i = lane & 3;
  const int b1I = b1blk / 3, b1s = b1blk - 3 * b1I;
  const int b1J1 = b1s, b1J2 = b1s < 2 ? 4 - b1s : 3;
  const R* const op1A = tabx + (4 * b1I + b1i) * kRow1;
  const R* const op1B1 = tabx + (4 * b1J1 + b1i) * kRow1 + 32;
  const R* const op1B2 = tabx + (4 * b1J2 + b1i) * kRow1 + 32;
  // pass over the points of half h of the wavefront (the first nl_valid lanes of the wavefront hold a valid point):
  // lanes 0..31 run the recurrence of one axis of their row's point, lanes 32..63 the other axis (stage_cos), one
  // 4-byte store per mode; then the valid groups of 4 points
  auto stage_and_mma1 = [&](R c_axis, int h, int nl_valid) {
    if constexpr (kBlock1) {
      {
        Tab1 t = tab1_init(c_axis, 32 * h + row < nl_valid);
        R* const dst = tabx + ((lo ? h : 1 - h) ? 32 : 0) + row;  // (stage_cos: lower lanes x in h = 0, y in h = 1)
#pragma unroll
        for (int q = 0; q < 10; ++q) {
          dst[(2 * q) * kRow1] = t.a;
          dst[(2 * q + 1) * kRow1] = t.b;
          tab1_step(t);
        }
      }
      const int rows_valid = nl_valid - 32 * h;  // > 0
      lds_fence();
#pragma unroll
      for (int g = 0; g < kStageRows / 4; ++g) {
        if (4 * g < rows_valid) {  // wavefront-uniform
          const f32x4 a = *reinterpret_cast<const f32x4*>(op1A + 4 * g);
          const f32x4 y1 = *reinterpret_cast<const f32x4*>(op1B1 + 4 * g);
          const f32x4 y2 = *reinterpret_cast<const f32x4*>(op1B2 + 4 * g);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bacc1 = mfma1(a[q], y1[q], bacc1);
            bacc2 = mfma1(a[q], y2[q], bacc2);
          }
        }
      }
      lds_fence();
    }
  };

  if constexpr (kBlock1) {
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < S) {
        const int nl = lanes_in_slot(j);
        R cl, cu;
        stage_cos(c1x[j], c1y[j], cl, cu);
        stage_and_mma1(cl, 0, nl);
        if (nl > 32) stage_and_mma1(cu, 1, nl);
      }
    }
  } else if constexpr (kBlock4) {
    // Rollout points, software-pipelined as below: the operands of a pass (12 reads) are read first, then the tile is
    // free and the recurrence + stores of the NEXT pass are issued between the matrix instructions of the first
    // 16-row group (pinned with scheduling barriers); the second group is skipped when it holds no valid point.
    R cl, cu;
    stage_cos(c1x[0], c1y[0], cl, cu);
    {
      Tab1 u = tab1_init(cl, 0 < cnt_of(row));
#pragma unroll
      for (int q = 0; q < kPairs; ++q) {
        tab1_store(u, st_lower, 2 * q);
        tab1_step(u);
      }
    }
    // the matrix instructions of one 16-row group with the staging of `dst` (kPairs stores) spread between them
    auto mma4_group_staging = [&](int q, Tab1& u, R* dst) {
      constexpr int kEvery = (NB * NB + kPairs - 1) / kPairs;  // a store after every kEvery-th instruction
      int done = 0;
#pragma unroll
      for (int I = 0; I < NB; ++I) {
#pragma unroll
        for (int J = 0; J < NB; ++J) {
          cacc[I][J] = mfma4(qa[q][I], qb[q][J], cacc[I][J]);
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
      for (int q2 = 0; q2 < kPairs; ++q2) {  // (none left when kPairs * kEvery <= NB * NB)
        if (q2 >= done) {
          tab1_store(u, dst, 2 * q2);
          tab1_step(u);
        }
      }
    };
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < S) {  // wavefront-uniform
        const int nl = lanes_in_slot(j);  // > 0
        lds_fence();
        read_operands4();  // rows of lanes 0..31, step j
        lds_fence();       // operands in registers: the tile is free
        {
          Tab1 u = tab1_init(cu, j < cnt_of(row + 32));
          operands4_ready();
          mma4_group_staging(0, u, st_upper);
          if (nl > 16) mma4_group(1);
        }
        lds_fence();
        read_operands4();  // rows of lanes 32..63, step j
        lds_fence();
        {
          const int jn = (j + 1 < kMaxS) ? j + 1 : j;
          stage_cos(c1x[jn], c1y[jn], cl, cu);
          Tab1 u = tab1_init(cl, (j + 1 < S) && (j + 1 < cnt_of(row)));
          operands4_ready();
          if (nl > 32) {
            mma4_group_staging(0, u, st_lower);
            if (nl > 48) mma4_group(1);
          } else {
#pragma unroll
            for (int q = 0; q < kPairs; ++q) {
              tab1_store(u, st_lower, 2 * q);
              tab1_step(u);
            }
          }
        }
      }
    }
    lds_fence();
  } else if (KC != 16 && NT == 1) {
    // Rollout points, software-pipelined: the 8 matrix instructions of a pass take 65 cycles of the matrix pipe
    // each; the table recurrence and the LDS stores of the NEXT pass are issued in between them (in program
    // order, pinned with scheduling barriers), so that only the operand reads wait.  Branch-free: passes beyond
    // the horizon multiply zero rows.
    R oa[kStageRows / 4], ob[kStageRows / 4];
    auto read_operands = [&]() {
#pragma unroll
      for (int m = 0; m < kStageRows / 4; ++m) {
        const int off = (4 * m + mk) * KS + mi;
        oa[m] = tabx[off];
        ob[m] = taby[off];
      }
    };
    R cl, cu;
    stage_cos(c1x[0], c1y[0], cl, cu);
    {
      Tab1 u = tab1_init(cl, 0 < cnt_of(row));
#pragma unroll
      for (int q = 0; q < kPairs; ++q) {
        tab1_store(u, st_lower, 2 * q);
        tab1_step(u);
      }
    }
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < S) {  // wavefront-uniform
        lds_fence();
        read_operands();  // rows of lanes 0..31, step j
        lds_fence();      // operands in registers: the tile is free
        // pass (j, lower half) on the matrix pipe; meanwhile the tile of the points of lanes 32..63 of step j.
        // Row groups past the horizon (all-zero rows) are skipped where no store is interleaved with them
        // (m >= kPairs): fp64 matrix instructions hold the vector pipe of the whole SIMD for their 64 cycles
        // (profiles/r02_ubench_coissue.txt), so at T = 200 the 3 empty groups of every upper pass are 19 % of the
        // contraction's pipe time
        const int nl = lanes_in_slot(j);
        const int n_lo = ((nl < 32 ? nl : 32) + 3) >> 2, n_up = (nl - 32 + 3) >> 2;  // row groups with a valid row
        {
          Tab1 u = tab1_init(cu, j < cnt_of(row + 32));
#pragma unroll
          for (int m = 0; m < kStageRows / 4; ++m) {
            if (m < kPairs || m < n_lo) {  // wavefront-uniform
              if (m & 1) acc1 = M::run(oa[m], ob[m], acc1);
              else acc0 = M::run(oa[m], ob[m], acc0);
            }
            if (m < kPairs) {
              tab1_store(u, st_upper, 2 * m);
              tab1_step(u);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        lds_fence();
        read_operands();  // rows of lanes 32..63, step j
        lds_fence();
        // pass (j, upper half); meanwhile the tile of the points of lanes 0..31 of step j + 1
        {
          const int jn = (j + 1 < kMaxS) ? j + 1 : j;
          stage_cos(c1x[jn], c1y[jn], cl, cu);
          Tab1 u = tab1_init(cl, (j + 1 < S) && (j + 1 < cnt_of(row)));
#pragma unroll
          for (int m = 0; m < kStageRows / 4; ++m) {
            if (m < kPairs || m < n_up) {  // wavefront-uniform
              if (m & 1) acc1 = M::run(oa[m], ob[m], acc1);
              else acc0 = M::run(oa[m], ob[m], acc0);
            }
            if (m < kPairs) {
              tab1_store(u, st_lower, 2 * m);
              tab1_step(u);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    lds_fence();
  } else {
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < S) {
        const int nl = lanes_in_slot(j);
        R cl, cu;
        stage_cos(c1x[j], c1y[j], cl, cu);
        stage_and_mma(cl, 0, nl);
        if (nl > 32) stage_and_mma(cu, 1, nl);
      }
    }
  }
  if (nmem > 0) {
    const R* const mem = p.mem_cols + 3 * static_cast<size_t>(p.mem_stride) * b;
    for (int c0 = 0; c0 < nmem; c0 += kWave) {
      const int q = c0 + lane;
      const bool valid = q < nmem;
      R sa, ca = R(0), sb, cb = R(0);
      if (valid) {
        sincospi_r((mem[3 * q + 0] - map_x) * p.inv_lx, &sa, &ca);
        sincospi_r((mem[3 * q + 1] - map_y) * p.inv_ly, &sb, &cb);
      }
      const int nl = nmem - c0;  // > 0
      R cl, cu;
      stage_cos(ca, cb, cl, cu);
      if constexpr (kBlock1) {
        stage_and_mma1(cl, 0, nl);
        if (nl > 32) stage_and_mma1(cu, 1, nl);
      } else if constexpr (kBlock4) {
        stage_and_mma4(cl, 0, nl);
        if (nl > 32) stage_and_mma4(cu, 1, nl);
      } else {
        stage_and_mma(cl, 0, nl);
        if (nl > 32) stage_and_mma(cu, 1, nl);
      }
    }
  }

  if constexpr (!kBlock4 && !kBlock1) load_lam_phi();
  EEA_WSTAMP(5);
  // the agent's sum record (eea_batch_io::d_ck_rec): c_k is in s_D [0, K^2) (before D takes the place), element K^2 = 1
  // (this agent counts), pad 0; read back one element per lane: coalesced stores to p.ck_rec [b]
  auto publish_record = [&]() {
    if (lane == 0) {
      s_D[K2] = R(1);
      if (p.rec_len > K2 + 1) s_D[K2 + 1] = R(0);
    }
    lds_fence();
    R* const rec = ck_rec_step + static_cast<size_t>(b) * p.rec_len;
    if (p.rec_ready != nullptr) {
      // device-bound exchange: the record leaves write-through, and once it has left the agent's ready mark follows
      // (MI355X_MICROARCH.md: sc1 payload -> s_waitcnt vmcnt(0) -> sc1 flag); the record sum polls the marks
#pragma unroll
      for (int r = 0; r < kRecRounds; ++r) {
        const int e = kWave * r + lane;
        if (e < p.rec_len) store_agent(rec + e, s_D[e]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) store_agent(p.rec_ready + b, rec_seq_step);
    } else {
#pragma unroll
      for (int r = 0; r < kRecRounds; ++r) {
        const int e = kWave * r + lane;
        if (e < p.rec_len) rec[e] = s_D[e];
      }
    }
    lds_fence();
  };
  // device-bound exchange, consumer side: the shared c_k this step consumes may still be on its way -- wait for its flag
  // here, as late as possible (the first use of c_bar is D).  On a timeout the agent goes on with its own c_k and says so
  auto bind_shared_ck = [&]() -> bool {  // true: the shared c_k replaces the own one
    if (ck_shared_step == nullptr) return false;
    if (p.ck_flag == nullptr) return true;  // wavefront-uniform
    bool ok = wait_flag(p.ck_flag, flag_seq_step);
    // a producer that gave up marks its record with a negative agent count
    if (ok && p.ck_shared_parts > 0) ok = !(load_agent(ck_shared_step + K2) < R(0));
    if (!ok && lane == 0 && p.status != nullptr) p.status[b] = 6;  // EEA_ERR_TIMEOUT
    return ok;
  };
  // D = lambda (c - phi), fourier_diff of ergodic_control.hpp:422, in both orientations
  if constexpr (kBlock4) {
    // every accumulator holds, in the four blocks of its row of 16 lanes, four partial sums over different point
    // groups: two row rotations complete them in all four copies.  Copy d of the row then finishes the pairs
    // (I, J) with (I NB + J) % 4 == d: lane l = 16 i + 4 d + j holds c(k1 = 4 I + i, k2 = 4 J + j).
    const R invN = R(1) / static_cast<R>(N);
    const int di = lane >> 4, db = (lane >> 2) & 3, dj = lane & 3;
    constexpr int TS = (NB * NB + 3) / 4;  // pairs per block copy
    R cv[TS], lamv[TS], phiv[TS];
    int idx[TS];
    bool okv[TS];
    // lambda_k, phi_k of this lane's entries first: the loads run under the rotations
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      const int pi = 4 * t + db, I = pi / NB, J = pi - NB * I;
      const int k1 = 4 * I + di, k2 = 4 * J + dj;
      okv[t] = pi < NB * NB && k1 < K && k2 < K;
      idx[t] = okv[t] ? k2 * K + k1 : 0;
      lamv[t] = p.lamdak[idx[t]];
      phiv[t] = p.phik[idx[t]];
    }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
      for (int J = 0; J < NB; ++J) cacc[I][J] = add_row_ror<8>(add_row_ror<4>(cacc[I][J]));
    }
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      R v = cacc[(4 * t) / NB][(4 * t) % NB];
#pragma unroll
      for (int d = 1; d < 4; ++d) {
        if (4 * t + d < NB * NB) v = (db == d) ? cacc[(4 * t + d) / NB][(4 * t + d) % NB] : v;
      }
      cv[t] = invN * v;
      if (p.ck != nullptr && okv[t]) p.ck[static_cast<size_t>(b) * K2 + idx[t]] = cv[t];
    }
    if (ck_rec_step != nullptr) {  // wavefront-uniform: this agent's record [c_k, 1, pad] through LDS, coalesced
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        if (okv[t]) s_D[idx[t]] = cv[t];
      }
      publish_record();
    }
    const bool use_shared = bind_shared_ck();
    SharedCk<R> shared_ck{ R(1), true };
    if (use_shared) shared_ck = shared_ck_begin<R>(p, ck_shared_step, K2);  // wavefront-uniform
#pragma unroll
    for (int t = 0; t < TS; ++t) {
      // decentralised consensus (eea_batch_io::d_ck_shared): the agents' shared c_k replaces the own one
      if (use_shared) cv[t] = shared_ck_value(p, shared_ck, ck_shared_step, idx[t], cv[t]);
      if (okv[t]) s_D[idx[t]] = lamv[t] * (cv[t] - phiv[t]);
    }
    lds_fence();
  } else if constexpr (kBlock1) {
    // lane 4 b + j, register r of the first (second) accumulator: c(k1 = 4 I + r, k2 = 4 J1 (J2) + j), complete.  The
    // four registers are four consecutive modes of row k2 (K = 20: 16-byte aligned): lambda_k, phi_k and D in fours
    const R invN = R(1) / static_cast<R>(N);
    const bool blk_ok = (lane >> 2) < 15;
    f32x4 cv[2], lamv[2], phiv[2];
    int idx[2];
    bool okv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      okv[t] = blk_ok && (t == 0 || b1s < 2);
      idx[t] = okv[t] ? (4 * (t ? b1J2 : b1J1) + b1i) * KC + 4 * b1I : 0;
      lamv[t] = *reinterpret_cast<const f32x4*>(p.lamdak + idx[t]);
      phiv[t] = *reinterpret_cast<const f32x4*>(p.phik + idx[t]);
      cv[t] = (t ? bacc2 : bacc1) * static_cast<float>(invN);
      if (p.ck != nullptr && okv[t]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p.ck[static_cast<size_t>(b) * K2 + idx[t] + r] = cv[t][r];
      }
    }
    if (ck_rec_step != nullptr) {  // wavefront-uniform
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (okv[t]) *reinterpret_cast<f32x4*>(s_D + idx[t]) = cv[t];
      }
      publish_record();
    }
    const bool use_shared = bind_shared_ck();
    SharedCk<R> shared_ck{ R(1), true };
    if (use_shared) shared_ck = shared_ck_begin<R>(p, ck_shared_step, K2);  // wavefront-uniform
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (use_shared) {
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[t][r] = shared_ck_value(p, shared_ck, ck_shared_step, idx[t] + r, static_cast<R>(cv[t][r]));
      }
      if (okv[t]) *reinterpret_cast<f32x4*>(s_D + idx[t]) = lamv[t] * (cv[t] - phiv[t]);
    }
    lds_fence();
  } else {
    const R invN = R(1) / static_cast<R>(N);
    if (NT == 1) acc0 = acc0 + acc1;
    const acc_t* const accs[4] = { &acc0, &acc1, &acc2, &acc3 };
#pragma unroll
    for (int t = 0; t < NT * NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k1 = 16 * (t / NT) + M::row(lane, r);
        const int k2 = 16 * (t % NT) + mi;
        if (k1 < K && k2 < K) {
          const R c = invN * (*accs[t])[r];
          if (p.ck != nullptr) p.ck[static_cast<size_t>(b) * K2 + k2 * K + k1] = c;
          if (ck_rec_step != nullptr) s_D[k2 * K + k1] = c;
        }
      }
    }
    if (ck_rec_step != nullptr) publish_record();  // wavefront-uniform
    const bool use_shared = bind_shared_ck();
    SharedCk<R> shared_ck{ R(1), true };
    if (use_shared) shared_ck = shared_ck_begin<R>(p, ck_shared_step, K2);  // wavefront-uniform
#pragma unroll
    for (int t = 0; t < NT * NT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k1 = 16 * (t / NT) + M::row(lane, r);
        const int k2 = 16 * (t % NT) + mi;
        if (k1 < K && k2 < K) {
          R c = invN * (*accs[t])[r];
          // decentralised consensus (eea_batch_io::d_ck_shared): the agents' shared c_k replaces the own one
          if (use_shared) c = shared_ck_value(p, shared_ck, ck_shared_step, k2 * K + k1, c);
          s_D[k2 * K + k1] = lam[t][r] * (c - phi[t][r]);
        }
      }
    }
    lds_fence();
  }

  EEA_WSTAMP(6);
  // ================= backward half =======================================================================
  // per step: ergodic-metric gradient (:418-436, basis.cpp:91-107)
  //   edx_x = -pi/lx sin(a x) sum_k1 k1 U_{k1-1}(cos a x) G(k1),  G(k1) = sum_k2 D(k1,k2) cos(b_k2 y)
  //   edx_y = -pi/ly sin(b y) sum_k2 k2 U_{k2-1}(cos b y) H(k2),  H(k2) = sum_k1 D(k1,k2) cos(a_k1 x)
  // (sin(k a) = sin(a) U_{k-1}(cos a)); rows of D^T / D are wavefront-uniform LDS reads
  constexpr int KA = (KC == 16) ? 16 : KC;
  // the barrier gradient waits in LDS while the gradient needs the registers (the tiles are dead)
  if (!STAGES) {
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      if (j < S) {
        s_g[(2 * j + 0) * kWave + lane] = g0[j];
        s_g[(2 * j + 1) * kWave + lane] = g1[j];
      }
    }
  }
  R ex[STAGES ? kMaxS : 1], ey[STAGES ? kMaxS : 1];  // kept apart from the barrier gradient only for the outputs
  // Top-heavy horizon with four slots (T = 193 .. 200): the last slot holds <= 8 steps in lanes 0 .. 7.  A regular pass
  // would spend a full slot's instructions on them (a vector instruction costs the same whatever the lane mask,
  // tools/ubench/exec_mask.hip); instead all 64 lanes take them together, 8 lanes per step (below the loop).
  constexpr bool kCoopTail = sizeof(R) == 8 && (KC == 10 || KC == 5);
  const bool coop_tail = kCoopTail && top_heavy && S == kMaxS;  // wavefront-uniform
  // fp32, K = 20 (BASELINE configs[2]) and K = 10: TWO steps of the lane at a time, every operation of the pass PACKED over the pair
  // (v_pk_fma_f32: (G_j(k1), G_j+1(k1)) += D(k1,k2) (cos b_k2 y_j, cos b_k2 y_j+1) with the element of D as the scalar
  // of both halves, likewise H, the four Chebyshev recurrences and the weighted sums).  Round 3's form packed pairs of x
  // MODES of one step: its recurrences and weighted sums (a third of the gradient's instructions) stayed scalar, at half
  // the packed rate.  The second step of a pair beyond S runs on zeros and is dropped.  (K = 10: 16.05 -> 15.47 us per pass at
  // T = 200; rows of D must start at even offsets: even K only, compile-time K only.)
  constexpr bool kPairGrad = sizeof(R) == 4 && (KC == 20 || KC == 10);
  if constexpr (kPairGrad) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto splat = [](float v) { return f2{ v, v }; };
    // The pairs of D are read as 8-byte pairs, alternately through two pointers the compiler cannot prove equal: merged
    // into 16-byte reads, the fourth element of each costs a register move before it can be the scalar of a packed
    // instruction (op_sel reaches the halves of an aligned register PAIR only).
    int zoff = 0;
    asm volatile("" : "+s"(zoff));
    const R* const s_Dz = s_D + zoff;
    auto d_pair = [&](int k2, int k1) { return *reinterpret_cast<const f2*>(((k1 & 2) ? s_Dz : s_D) + k2 * K + k1); };
#pragma unroll
    for (int jp = 0; jp < kMaxS; jp += 2) {
      if (STAGES) ex[jp] = ey[jp] = ex[jp + 1] = ey[jp + 1] = R(0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (jp < S) {  // wavefront-uniform
        constexpr int KB = grad_block(KC);
        const f2 cxp = f2{ static_cast<float>(c1x[jp]), static_cast<float>(c1x[jp + 1]) };
        const f2 cyp = f2{ static_cast<float>(c1y[jp]), static_cast<float>(c1y[jp + 1]) };
        const f2 twox = cxp + cxp, twoy = cyp + cyp;
        f2 accx = splat(0.0f), accy = splat(0.0f);
        f2 ta = splat(1.0f), tb = cxp;            // T_k, T_{k+1} of the x angles at the block's first mode
        f2 ua = splat(0.0f), ub = splat(1.0f);    // U_{k-1}, U_k
#pragma unroll
        for (int kb0 = 0; kb0 < KC; kb0 += KB) {
          if (kb0 > 0) {  // one block at a time (registers)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
          f2 cx2[KB], G2[KB];
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            cx2[i] = ta;
            const f2 tn = __builtin_elementwise_fma(twox, tb, -ta);
            ta = tb;
            tb = tn;
          }
          // row k2 = 0: cos(0 y) = 1 and k2 sin(0) = 0: only G takes part (k1 = 0 is not special-cased: cos(0 x) = 1 is
          // in the table and G(0) is never used)
#pragma unroll
          for (int q = 0; q < KB / 2; ++q) {
            const f2 d = d_pair(0, kb0 + 2 * q);
            G2[2 * q] = __builtin_elementwise_fma(splat(d.x), splat(1.0f), splat(0.0f));  // (one instruction, not two moves)
            G2[2 * q + 1] = __builtin_elementwise_fma(splat(d.y), splat(1.0f), splat(0.0f));
          }
          f2 um = splat(0.0f), u0 = splat(1.0f);  // U_{k2-2}, U_{k2-1} of the y angles
          f2 tm = splat(1.0f), t0 = cyp;          // T_{k2-1}, T_{k2}
#pragma unroll
          for (int k2 = 1; k2 < KC; ++k2) {
            if (k2 < K) {  // wavefront-uniform, always true (kRowGuard)
              f2 h2;
#pragma unroll
              for (int q = 0; q < KB / 2; ++q) {
                const f2 d = d_pair(k2, kb0 + 2 * q);  // D(k1, k2), D(k1 + 1, k2): rows start even
                G2[2 * q] = __builtin_elementwise_fma(splat(d.x), t0, G2[2 * q]);
                h2 = (q == 0) ? splat(d.x) * cx2[0] : __builtin_elementwise_fma(splat(d.x), cx2[2 * q], h2);
                G2[2 * q + 1] = __builtin_elementwise_fma(splat(d.y), t0, G2[2 * q + 1]);
                h2 = __builtin_elementwise_fma(splat(d.y), cx2[2 * q + 1], h2);
              }
              accy = __builtin_elementwise_fma(u0 * splat(static_cast<float>(k2)), h2, accy);
              const f2 un = __builtin_elementwise_fma(twoy, u0, -um);
              um = u0;
              u0 = un;
              const f2 tn = __builtin_elementwise_fma(twoy, t0, -tm);
              tm = t0;
              t0 = tn;
            }
          }
          // edx_x: sum_k1 k1 U_{k1-1}(cos a x) G(k1) over this block
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int k1 = kb0 + i;
            if (k1 > 0) accx = __builtin_elementwise_fma(ua * splat(static_cast<float>(k1)), G2[i], accx);
            const f2 un = __builtin_elementwise_fma(twox, ub, -ua);
            ua = ub;
            ub = un;
          }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = jp + e;
          if (j < S) {  // wavefront-uniform
            const R exj = (-p.pi_lx * s1x[j] * static_cast<R>(e ? accx.y : accx.x)) * p.expl_weight;
            const R eyj = (-p.pi_ly * s1y[j] * static_cast<R>(e ? accy.y : accy.x)) * p.expl_weight;
            if (STAGES) {
              ex[j] = exj;
              ey[j] = eyj;
            } else {
              const bool act = j < cnt;
              g0[j] = act ? exj + s_g[(2 * j + 0) * kWave + lane] : R(0);
              g1[j] = act ? eyj + s_g[(2 * j + 1) * kWave + lane] : R(0);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kMaxS; ++j) {
    if (kPairGrad) break;
    if (STAGES) ex[j] = ey[j] = R(0);
    // one step at a time: the rows of D are re-read per step; without the fence the compiler keeps them (and
    // the arrays of all four steps) in registers across the unrolled steps and spills
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (j < S && !(j == kMaxS - 1 && coop_tail)) {
      // Pass(es) over D, one per block of KB x-modes k1: G(k1) accumulates over the rows, the block's part of
      // H(k2) is complete at the end of row k2.  The Chebyshev pairs of the x angle run on across the blocks.
      constexpr int KB = grad_block(KC);
      R accx = R(0), accy = R(0);
      const R twox = c1x[j] + c1x[j], twoy = c1y[j] + c1y[j];
      R ta = R(1), tb = c1x[j];   // T_k, T_{k+1} of the x angle at the block's first mode
      R ua = R(0), ub = R(1);     // U_{k-1}, U_k
#pragma unroll
      for (int kb0 = 0; kb0 < KA; kb0 += KB) {
        if (kb0 > 0) {  // one block at a time (registers), like the steps
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
        R cxa[KB], G[KB];
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          cxa[i] = ta;
          const R tn = twox * tb - ta;
          ta = tb;
          tb = tn;
        }
        {
        // row k2 = 0: cos(0 y) = 1 and k2 sin(0) = 0: only G takes part (G(0) is never used: k1 sin(0) = 0)
        {
          const R* const row = s_D + kb0;
#pragma unroll
          for (int i = 0; i < KB; ++i) {
            const int k1 = kb0 + i;
            G[i] = (k1 == 0 || k1 >= KA) ? R(0) : ((kGenericK && k1 >= K) ? R(0) : row[i]);
          }
        }
        R um = R(0), u0 = R(1);         // U_{k2-2}, U_{k2-1} of the y angle
        R tm = R(1), t0 = c1y[j];       // T_{k2-1}, T_{k2}
#pragma unroll
        for (int k2 = 1; k2 < KA; ++k2) {
          if (!kRowGuard || k2 < K) {  // wavefront-uniform
            const R* const row = s_D + k2 * K + kb0;
            R h = R(0);  // (one chain: with four wavefronts per SIMD its latency is covered, and the second chain's final
                         // add was 27 instructions per agent: -0.9 %)
#pragma unroll
            for (int i = 0; i < KB; ++i) {
              const int k1 = kb0 + i;
              if (k1 < KA) {  // compile time
                // generic instance: modes beyond K read the next row's entries (inside D) and are masked
                const R d = (kGenericK && k1 >= K) ? R(0) : row[i];
                if (k1 > 0) G[i] += d * t0;
                if (k1 == 0) h = d;  // cos(0 x) = 1
                else h += d * cxa[i];
              }
            }
            accy = fma_k(u0 * h, k2, accy);
            const R un = twoy * u0 - um;
            um = u0;
            u0 = un;
            const R tn = twoy * t0 - tm;
            tm = t0;
            t0 = tn;
          }
        }
        }
        // edx_x: sum_k1 k1 U_{k1-1}(cos a x) G(k1) over this block
#pragma unroll
        for (int i = 0; i < KB; ++i) {
          const int k1 = kb0 + i;
          if (k1 > 0 && k1 < KA) accx = fma_k(ua * G[i], k1, accx);  // G(k1 >= K) = 0 in the generic instance
          const R un = twox * ub - ua;
          ua = ub;
          ub = un;
        }
      }
      const R sx = s1x[j], sy = s1y[j];
      const R exj = (-p.pi_lx * sx * accx) * p.expl_weight;
      const R eyj = (-p.pi_ly * sy * accy) * p.expl_weight;
      if (STAGES) {
        ex[j] = exj;
        ey[j] = eyj;
      } else {
        // g = edx + bdx (inactive steps contribute nothing to the suffix sums); the basis registers of this
        // step are dead from here on
        const bool act = j < cnt;
        g0[j] = act ? exj + s_g[(2 * j + 0) * kWave + lane] : R(0);
        g1[j] = act ? eyj + s_g[(2 * j + 1) * kWave + lane] : R(0);
      }
    }
  }
  if constexpr (kCoopTail) {
    if (coop_tail) {
      // Lane l = 8 s + t works on the step of lane s (slot 3), rows k2 = t and t + 8 of D:
      //   edx_x = -pi/lx sin(a) sum_k2 cos(k2 b) Bx(k2),   Bx(k2) = sum_k1 D(k2,k1) k1 U_{k1-1}(cos a)
      //   edx_y = -pi/ly        sum_k2 k2 sin(k2 b) A(k2),  A(k2)  = sum_k1 D(k2,k1) cos(k1 a)
      // (the same sums as above in another order).  The x tables by the Chebyshev recurrences in every lane; cos / sin of
      // the lane's own rows k2 b by binary powering of (cos b, sin b) -- the row differs from lane to lane, a recurrence
      // would have to be run to the end and picked from.
      constexpr int jt = kMaxS - 1;
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const int st = lane >> 3, sub = lane & 7;
      R* const s_tail = s_g + 2 * kMaxS * kWave;  // [3][8]
      if (lane < 8) {
        s_tail[lane] = c1x[jt];
        s_tail[8 + lane] = c1y[jt];
        s_tail[16 + lane] = s1y[jt];
      }
      lds_fence();
      const R cxs = s_tail[st], cys = s_tail[8 + st], sys = s_tail[16 + st];
      lds_fence();  // (the hand-over space is written again below)
      R cxa[KC], kux[KC];  // cos(k1 a), k1 U_{k1-1}(cos a)
      {
        const R two = cxs + cxs;
        R ta = R(1), tb = cxs, ua = R(0), ub = R(1);
#pragma unroll
        for (int k1 = 0; k1 < KC; ++k1) {
          cxa[k1] = ta;
          kux[k1] = static_cast<R>(k1) * ua;
          const R tn = two * tb - ta, un = two * ub - ua;
          ta = tb;
          tb = tn;
          ua = ub;
          ub = un;
        }
      }
      // (cos, sin)(k2 b) for k2 = sub: z^sub = z^(bit 0) z2^(bit 1) z4^(bit 2)
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
      R px = R(0), py = R(0);
      auto tail_row = [&](int k2, R cyv, R ksv) {  // k2 < KC; cyv = cos(k2 b), ksv = k2 sin(k2 b) (both 0: row not taken)
        const R* const row = s_D + k2 * K;
        R a0 = R(0), a1 = R(0), b0 = R(0), b1 = R(0);
#pragma unroll
        for (int k1 = 0; k1 < KC; ++k1) {
          const R d = row[k1];
          if (k1 & 1) {
            a1 += d * cxa[k1];
            b1 += d * kux[k1];
          } else {
            a0 += d * cxa[k1];
            if (k1 > 0) b0 += d * kux[k1];
          }
        }
        px += cyv * (b0 + b1);
        py += ksv * (a0 + a1);
      };
      tail_row(sub < KC ? sub : KC - 1, sub < KC ? wc : R(0), sub < KC ? static_cast<R>(sub) * ws : R(0));
      if (KC > 8) {  // rows 8 .. KC-1: z^(sub + 8) = z^sub z8
        const R z8c = z4c * z4c - z4s * z4s, z8s = (z4c + z4c) * z4s;
        const R vc = wc * z8c - ws * z8s, vs = wc * z8s + ws * z8c;
        const bool has = sub + 8 < KC;
        tail_row(has ? sub + 8 : KC - 1, has ? vc : R(0), has ? static_cast<R>(sub + 8) * vs : R(0));
      }
      // the 8 lanes of a step: xor 1, xor 2 (quad permutations), then the mirror image inside the half row
      px += dpp_or_zero<0xB1, 0xf>(px);
      py += dpp_or_zero<0xB1, 0xf>(py);
      px += dpp_or_zero<0x4E, 0xf>(px);
      py += dpp_or_zero<0x4E, 0xf>(py);
      px += dpp_or_zero<0x141, 0xf>(px);
      py += dpp_or_zero<0x141, 0xf>(py);
      if (sub == 0) {
        s_tail[2 * st] = px;
        s_tail[2 * st + 1] = py;
      }
      lds_fence();
      const R accx = s_tail[2 * (lane & 7)], accy = s_tail[2 * (lane & 7) + 1];
      const R sx = s1x[jt];
      const R exj = (-p.pi_lx * sx * accx) * p.expl_weight;
      const R eyj = (-p.pi_ly * accy) * p.expl_weight;  // (sin b is in the row factors)
      if (STAGES) {
        ex[jt] = exj;
        ey[jt] = eyj;
      } else {
        const bool act = jt < cnt;  // lanes 0 .. r-1
        g0[jt] = act ? exj + s_g[(2 * jt + 0) * kWave + lane] : R(0);
        g1[jt] = act ? eyj + s_g[(2 * jt + 1) * kWave + lane] : R(0);
      }
    }
  }
  if (STAGES) {
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

  EEA_WSTAMP(7);
  // co-state rows 0,1: rho_i = rho_{i+1} + dt g_i (suffix sums over the horizon, rho_T = 0,
  // ergodic_control.hpp:203); row 2:
  //   rho2_i = rho2_{i+1} + dt (S_i(rho_{i+1}) + dt/2 S_i(g_i)), S_i(v) = A(0,2) v0 + A(1,2) v1,
  //   A = fdx(x_i, u_i) (omni.hpp:194-197, cart.hpp:183-186); then u_i = clamp(-Rinv B^T rho_i)
  // The (shifted) controls of the lane's steps again, from L2, issued here so that their latency runs under the
  // scans: carried in registers since the first load they were spilled to scratch through the contraction and the
  // gradient (20 MB of HBM traffic per launch).  Branch-free: out-of-range steps read element 0 and select zero.
  R vxr[kMaxS], vyr[kMaxS];
  {
    size_t opaque = 0;  // the compiler must not recognise (and keep alive) the earlier loads of the same words
    asm volatile("" : "+v"(opaque));
    const R* const ut_again = ut + opaque;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) {
      const int src = i0 + j + 1;
      const bool ok = j < cnt && src < T;  // (a slot this lane does not own belongs to another lane's step)
      const int idx = ok ? 3 * src : 0;
      const R a = ut_again[idx];
      vxr[j] = ok ? a : R(0);
      vyr[j] = R(0);
      if (MODEL == kModelOmni) {
        const R bq = ut_again[idx + 1];
        vyr[j] = ok ? bq : R(0);
      }
    }
  }
  R r0[kMaxS], r1[kMaxS];  // inclusive suffix within the lane, then the co-state after step j
  R tot0, tot1, tot2;      // the wavefront totals of the three co-state scans (wavefront-uniform)
  {
    R s0 = R(0), s1 = R(0);
#pragma unroll
    for (int j = kMaxS - 1; j >= 0; --j) {
      if (STAGES) {
        const bool act = j < cnt;
        g0[j] = act ? ex[j] + g0[j] : R(0);  // g = edx + bdx
        g1[j] = act ? ey[j] + g1[j] : R(0);
      }
      s0 += dt * g0[j];
      s1 += dt * g1[j];
      r0[j] = s0;
      r1[j] = s1;
    }
    // suffix over the lanes = wavefront total - inclusive prefix
    const R i0s = wave_inclusive_scan_dpp(s0), i1s = wave_inclusive_scan_dpp(s1);
    tot0 = read_lane63(i0s);
    tot1 = read_lane63(i1s);
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
      // post-step heading parked by the forward half (slots beyond S hold nothing that is used: qv = 0 there)
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
      // rho_{i+1} = rho_i - dt g_i (rows 0,1)
      const R sE = a02 * (r0[j] - dt * g0[j]) + a12 * (r1[j] - dt * g1[j]);
      const R sG = a02 * g0[j] + a12 * g1[j];
      // zero controls beyond the horizon give a02 = a12 = 0 there; slots beyond S never parked a heading (garbage)
      const R qv = (j < S) ? dt * (sE + p.half_dt * sG) : R(0);
      s2 += qv;
      r2[j] = s2;
    }
    const R i2s = wave_inclusive_scan_dpp(s2);
    tot2 = read_lane63(i2s);
    const R o2 = tot2 - i2s;
#pragma unroll
    for (int j = 0; j < kMaxS; ++j) r2[j] += o2;
  }

  EEA_WSTAMP(8);
  // ---- u_i = clamp(-Rinv B(x_i)^T rho_i)  (ergodic_control.hpp:438-451) ---------------------------------
  // std::clamp lets a NaN pass (both comparisons are false); min(max(u, lo), hi) -- 2 instructions per component instead of
  // 2 compares + 4 selects + the moves that feed them: 125 -> ~30 vector instructions per agent -- would turn it into the
  // lower limit.  A NaN anywhere in the agent's state or gradients reaches one of the three co-state scan totals (they
  // sum every step's g and every step's A(x, u) rho terms), so ONE wavefront-uniform test picks the form: the fast one
  // when the totals are finite, the comparing one otherwise (signed zeros: max / min may return +0 where std::clamp
  // returns -0; equal as numbers).
  // (FINITE, not merely numbers: an infinite co-state times Rinv's zeros is a NaN the totals do not show)
  const R inf_r = static_cast<R>(__builtin_huge_val());
  const bool nan_free = __all((fabs(tot0) < inf_r) && (fabs(tot1) < inf_r) && (fabs(tot2) < inf_r));
  auto update_controls = [&](auto fast_tag) {
  constexpr bool kFast = decltype(fast_tag)::value;
#pragma unroll
  for (int j = 0; j < kMaxS; ++j) {
    if (j < cnt) {
      const int i = i0 + j;
      R v0, v1, v2;
      if (MODEL == kModelOmni) {  // omni.hpp:205-212
        v0 = cth[j] * r0[j] + sth[j] * r1[j];
        v1 = -sth[j] * r0[j] + cth[j] * r1[j];
        v2 = r2[j];
      } else {  // cart.hpp:194-203: B^T rho has an exact zero in row 1
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
        sm[0 * kMaxS * kWave + i] = u[0];
        sm[1 * kMaxS * kWave + i] = u[1];
        sm[2 * kMaxS * kWave + i] = u[2];
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
        if (p.done != nullptr && b == 0) {
          // the host polls this word instead of waiting for the kernel's completion signal (RESIDENT: the request's number,
          // behind every store of the request -- they are all this wavefront's)
          __hip_atomic_store(p.done, RESIDENT ? static_cast<int>(res_last) : p.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
  };
  if (nan_free) update_controls(std::true_type{});  // wavefront-uniform
  else update_controls(std::false_type{});
  EEA_WSTAMP(9);
  if (step + 1 < n_steps) {  // wavefront-uniform: the next step's backward half re-reads the controls just stored (and the LDS is reused)
    // only THIS wavefront reads them back: work-group scope -- the stores drained (s_waitcnt vmcnt(0)), the CU's own
    // write-through L1 is coherent for its own wavefronts.  (Agent scope writes back and invalidates the L2 on this
    // multi-XCD part: 150 us per step, profiles/r04_multi_step.txt.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    lds_fence();
  }
  EEA_WSTAMP_RT(11);
  if constexpr (RESIDENT) {
    if (res_leaving) return;  // (the controls of this request are in ut: every step stores them)
    res_handover = true;
  }
  }  // step
}

}  // namespace wave
}  // namespace eea
