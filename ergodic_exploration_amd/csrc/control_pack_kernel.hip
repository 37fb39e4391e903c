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
// pack, 8 / 16 / 32 = that group size where eligible; 0 = by the cost model below, refitted in round 6 to the instances that run
// four wavefronts per SIMD (profiles/r06_pack_points.txt, profiles/r06_ablation.txt item 11):
//   * a wavefront of either kernel issues I(S) = 600 + 580 S pipe slots (K = 10; 500 + 250 S at K = 5), S = ceil(T / L) steps
//     per lane (SQ_INSTS_VALU, matrix instructions counted four times) -- whatever the number of agents in it; a top-heavy
//     horizon's last slot costs about 0.6 slots; with 16 lanes per agent a slot whose steps all lie in lanes 0..7 takes one
//     half-pass instead of two (x 0.8); the 8-lane K = 10 instances retire their instructions ~20 % slower (two accumulator sets);
//   * w wavefronts of a SIMD retire an instruction per c(w) = max(12.5, 4.3 w) cycles each: a lone wavefront waits on its own
//     dependency chains (measured 11.6 - 12.9 cycles per instruction at w <= 2), from ~3 per SIMD on the pipe is the limit
//     (16 - 17 at w = 4); more wavefronts than the instance's residency R (4, or 3 for the 32-lane instances and the 8-lane one at
//     four steps per lane) run in rounds: floor(w / R) rounds at c(R) + the rest at c(w mod R);
//   * the call is taken to be one of TWO concurrent agent groups (the launch form of bench.py and AgentBatch): w = 2 x its own
//     wavefronts / 1024 SIMDs.
// Ties go to the narrower group; a batch of fewer than 256 wavefronts is not worth packing.
int control_pack_lanes(const ControlParams<double>& p, unsigned B, int forced)
{
  if (forced == 64) return 0;
  if (forced == 8 || forced == 16 || forced == 32) return control_pack_eligible(p, forced) ? forced : 0;
  auto c = [](double w) { return 4.3 * w > 12.5 ? 4.3 * w : 12.5; };
  double best = 0.0;
  int best_l = 0;
  for (int lanes = 64; lanes >= 8; lanes /= 2) {
    if (lanes < 64 && !control_pack_eligible(p, lanes)) continue;
    if (lanes == 64 && p.T > 4 * 64) continue;
    const int A = 64 / lanes, S = (p.T + lanes - 1) / lanes;
    const unsigned waves = (B + A - 1) / A;
    if (lanes < 64 && waves < 256u) continue;
    const double w = 2.0 * static_cast<double>(waves) / 1024.0;
    const int r_top = p.T - lanes * (S - 1);
    const double slots = (lanes < 64 && S == pack::kMaxS && r_top <= lanes / 8) ? S - 0.4 : static_cast<double>(S);
    double insts = p.K == 5 ? 500.0 + 250.0 * slots : 600.0 + 580.0 * slots;
    if (lanes == 16 && (p.T + S - 1) / S <= 8) insts *= 0.8;  // every slot's steps in lanes 0..7: one half-pass per slot
    if (lanes == 8 && p.K != 5) insts *= 1.2;
    // wavefronts per SIMD the instance is compiled for (control_pack_impl.hpp waves_per_simd; the wavefront kernel: 4)
    const double R = (lanes == 32 || (lanes == 8 && S > 3 && p.K != 5)) ? 3.0 : 4.0;
    const double full = static_cast<double>(static_cast<int>(w / R)), rest = w - full * R;
    const double cost = insts * (full * c(R) + (rest > 1e-9 ? c(rest) : 0.0));
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
