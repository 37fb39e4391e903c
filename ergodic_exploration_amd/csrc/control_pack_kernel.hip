// Instantiations and dispatch of the several-agents-per-wavefront control kernel (control_pack_impl.hpp).
#include "control_pack_impl.hpp"

namespace eea
{
namespace
{
constexpr int kPackWPB = 4;  // wavefronts per workgroup (they share nothing)

template <int MODEL, int KC, bool STAGES, int L>
hipError_t launch_pack_one(const ControlParams<double>& p, unsigned B, bool rollout_only, hipStream_t stream)
{
  constexpr int A = kWave / L;
  const int S = (p.T + L - 1) / L;
  const size_t lds = static_cast<size_t>(kPackWPB) * pack::wave_lds_elems(KC, A, S) * sizeof(double);
  const unsigned waves = (B + A - 1) / A;
  // (horizons of <= 3 steps per lane take the instance whose per-step register arrays have three elements)
  if (S <= 3) {
    hipLaunchKernelGGL((pack::control_pack_kernel<MODEL, KC, STAGES, L, kPackWPB, 3>), dim3((waves + kPackWPB - 1) / kPackWPB),
                       dim3(kPackWPB * kWave), lds, stream, p, B, S, rollout_only ? 1 : 0);
  } else {
    hipLaunchKernelGGL((pack::control_pack_kernel<MODEL, KC, STAGES, L, kPackWPB, pack::kMaxS>), dim3((waves + kPackWPB - 1) / kPackWPB),
                       dim3(kPackWPB * kWave), lds, stream, p, B, S, rollout_only ? 1 : 0);
  }
  return hipGetLastError();
}

template <int MODEL, int KC, bool STAGES>
hipError_t launch_pack_l(const ControlParams<double>& p, unsigned B, bool rollout_only, int lanes, hipStream_t stream)
{
  switch (lanes) {
    case 8:
      return launch_pack_one<MODEL, KC, STAGES, 8>(p, B, rollout_only, stream);
    case 16:
      return launch_pack_one<MODEL, KC, STAGES, 16>(p, B, rollout_only, stream);
    default:
      return launch_pack_one<MODEL, KC, STAGES, 32>(p, B, rollout_only, stream);
  }
}

template <int MODEL, bool STAGES>
hipError_t launch_pack_k(const ControlParams<double>& p, unsigned B, bool rollout_only, int lanes, hipStream_t stream)
{
  return p.K == 5 ? launch_pack_l<MODEL, 5, STAGES>(p, B, rollout_only, lanes, stream)
                  : launch_pack_l<MODEL, 10, STAGES>(p, B, rollout_only, lanes, stream);
}
}  // namespace

// Can a group of `lanes` lanes per agent take this call?  fp64, K = 5 / 10, T <= 4 lanes; not the single-agent mailbox.
// (Round 6: the exchange's buffers -- sum records out, ready marks, the flag of the shared c_k -- are served here too, so
// decentralised consensus keeps packing at short horizons.)
bool control_pack_eligible(const ControlParams<double>& p, int lanes)
{
  if (lanes != 8 && lanes != 16 && lanes != 32) return false;
  if (p.K != 5 && p.K != 10) return false;
  if (p.T < 1 || p.T > pack::kMaxS * lanes) return false;
  if (p.done != nullptr || p.dbg != nullptr) return false;
  return true;
}

// Lanes per agent for a batch of B agents: 0 = the wavefront-per-agent kernel.  `forced` (EEA_OPT_AGENT_LANES): 64 = never
// pack, 8 / 16 / 32 = that group size where eligible; 0 = by the cost model below, fitted to profiles/r05_pack_sweep.txt:
//   * a wavefront of either kernel issues I(S) = 600 + 580 S pipe slots (K = 10; 500 + 250 S at K = 5), S = ceil(T / L) steps
//     per lane -- measured SQ_INSTS_VALU: 1071 / 1540 / 2010 at S = 1 / 2 / 3 with L = 64 and 1099 / 1568 / 2039 / 2509 with
//     L = 8, matrix instructions counted four times -- whatever the number of agents in it;
//   * w wavefronts per SIMD retire an instruction per max(16, 5 w) cycles each: one wavefront alone waits on its own
//     dependency chains (T = 20, one wavefront per SIMD: 7.6 us = 16 cycles per instruction), from ~3 per SIMD on the pipe is
//     the limit (0.75 - 0.8 busy);
//   * the call is taken to be one of TWO concurrent agent groups (the launch form of bench.py and AgentBatch): w = 2 x its own
//     wavefronts / 1024 SIMDs.
// Ties go to the narrower group; a batch of fewer than 256 wavefronts is not worth packing.
int control_pack_lanes(const ControlParams<double>& p, unsigned B, int forced)
{
  if (forced == 64) return 0;
  if (forced == 8 || forced == 16 || forced == 32) return control_pack_eligible(p, forced) ? forced : 0;
  double best = 0.0;
  int best_l = 0;
  for (int lanes = 64; lanes >= 8; lanes /= 2) {
    if (lanes < 64 && !control_pack_eligible(p, lanes)) continue;
    if (lanes == 64 && p.T > 4 * 64) continue;
    const int A = 64 / lanes, S = (p.T + lanes - 1) / lanes;
    const unsigned waves = (B + A - 1) / A;
    if (lanes < 64 && waves < 256u) continue;
    const double w = 2.0 * static_cast<double>(waves) / 1024.0;
    // (a top-heavy horizon's last slot -- T = 3 L + 1 .. 3 L + L / 8, its gradient taken by all lanes together -- costs about
    // 0.6 slots: yaml T = 50 on 16 lanes 20.85 -> 19.4 us per 12288-agent pass)
    const int r_top = p.T - lanes * (S - 1);
    const double slots = (lanes < 64 && S == pack::kMaxS && r_top <= lanes / 8) ? S - 0.4 : static_cast<double>(S);
    const double insts = p.K == 5 ? 500.0 + 250.0 * slots : 600.0 + 580.0 * slots;
    const double cost = insts * (5.0 * w > 16.0 ? 5.0 * w : 16.0);
    if (best_l == 0 || cost <= best) {
      best = cost;
      best_l = lanes;
    }
  }
  return best_l == 64 ? 0 : best_l;
}

hipError_t launch_control_pack(const ControlParams<double>& p, unsigned B, int model, bool rollout_only, int lanes,
                               hipStream_t stream)
{
  if (B == 0) return hipSuccess;
  const bool stages = rollout_only || p.traj != nullptr || p.edx != nullptr || p.bdx != nullptr || p.rhot != nullptr;
  if (model == kModelOmni) {
    return stages ? launch_pack_k<kModelOmni, true>(p, B, rollout_only, lanes, stream)
                  : launch_pack_k<kModelOmni, false>(p, B, rollout_only, lanes, stream);
  }
  return stages ? launch_pack_k<kModelSimpleCart, true>(p, B, rollout_only, lanes, stream)
                : launch_pack_k<kModelSimpleCart, false>(p, B, rollout_only, lanes, stream);
}
}  // namespace eea
