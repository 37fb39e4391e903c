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
  const size_t lds = static_cast<size_t>(kPackWPB) * pack::wave_lds_elems(KC, A) * sizeof(double);
  const unsigned waves = (B + A - 1) / A;
  hipLaunchKernelGGL((pack::control_pack_kernel<MODEL, KC, STAGES, L, kPackWPB>), dim3((waves + kPackWPB - 1) / kPackWPB),
                     dim3(kPackWPB * kWave), lds, stream, p, B, S, rollout_only ? 1 : 0);
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

// Can a group of `lanes` lanes per agent take this call?  fp64, K = 5 / 10, T <= 4 lanes; none of the device-bound
// exchange's buffers, not the single-agent mailbox.
bool control_pack_eligible(const ControlParams<double>& p, int lanes)
{
  if (lanes != 8 && lanes != 16 && lanes != 32) return false;
  if (p.K != 5 && p.K != 10) return false;
  if (p.T < 1 || p.T > pack::kMaxS * lanes) return false;
  if (p.ck_rec != nullptr || p.rec_ready != nullptr || p.ck_flag != nullptr || p.done != nullptr || p.dbg != nullptr) return false;
  return true;
}

// Lanes per agent for a batch of B agents: 0 = the wavefront-per-agent kernel.  `forced` (EEA_OPT_AGENT_LANES): 64 = never
// pack, 8 / 16 / 32 = that group size where eligible; 0 = by the cost model:
//   the packed kernel issues about I(S) = 440 + 450 S vector instructions per WAVEFRONT of A = 64 / L agents (S = ceil(T / L)
//   steps per lane; control_wave_kernel: the same with A = 1, L = 64), and a SIMD with w resident wavefronts retires one fp64
//   instruction per max(5.4, 12 / w) cycles (profiles/r04_ubench_rates.txt: 0.34 / 0.65 / 0.74 of the pipe at 1 / 2 / 4
//   wavefronts) -- so packing pays only while the batch still fills the 1024 SIMDs with >= 2 wavefronts each.
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
    const double waves = static_cast<double>((B + A - 1) / A) / 1024.0;  // per SIMD
    const double w_res = waves < 1.0 ? 1.0 : (waves > (lanes == 64 ? 4.0 : 3.0) ? (lanes == 64 ? 4.0 : 3.0) : waves);
    const double cpi = 12.0 / w_res > 5.4 ? 12.0 / w_res : 5.4;
    const double rounds = waves < 1.0 ? 1.0 : waves / w_res;  // sequential rounds of resident wavefronts
    const double cost = (440.0 + 450.0 * S) * cpi * w_res * rounds;
    if (best_l == 0 || cost < best * 0.97) {  // (ties go to the wider group: fewer agents share a wavefront's fate)
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
