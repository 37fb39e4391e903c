// Instantiations and dispatch of the wavefront-per-agent control kernel (control_wave_impl.hpp).
#include "control_wave_impl.hpp"

// wavefronts (agents) per workgroup: they share nothing; 4 keeps the dispatch count low
#ifndef EEA_WAVE_WPB
#define EEA_WAVE_WPB 4
#endif

namespace eea
{
namespace
{
template <typename R, int MODEL, int KC, bool STAGES>
hipError_t launch_wave_one(const ControlParams<R>& p, unsigned B, bool rollout_only, hipStream_t stream)
{
  // K = 20 in fp64: 14.3 KB of LDS per agent; workgroups of 4 agents (57 KB) fit twice into the CU's 160 KB (8
  // wavefronts), single agents 11 times (the registers allow 12)
  constexpr int WPB = (KC == 20 && sizeof(R) == 8) ? 1 : EEA_WAVE_WPB;
  const int S = (p.T + kWave - 1) / kWave;
  const size_t lds = static_cast<size_t>(WPB) * wave::wave_lds_elems(KC) * sizeof(R);
  void (*kern)(const ControlParams<R>, const unsigned, const int, const int) = wave::control_wave_kernel<R, MODEL, KC, STAGES, WPB>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, dim3((B + WPB - 1) / WPB), dim3(WPB * kWave), lds, stream, p, B, S, rollout_only ? 1 : 0);
  return hipGetLastError();
}

template <typename R, int MODEL, bool STAGES>
hipError_t launch_wave_k(const ControlParams<R>& p, unsigned B, bool rollout_only, hipStream_t stream)
{
  switch (p.K) {
    case 5:
      return launch_wave_one<R, MODEL, 5, STAGES>(p, B, rollout_only, stream);
    case 10:
      return launch_wave_one<R, MODEL, 10, STAGES>(p, B, rollout_only, stream);
    case 20:
      return launch_wave_one<R, MODEL, 20, STAGES>(p, B, rollout_only, stream);
    default:
      return launch_wave_one<R, MODEL, 16, STAGES>(p, B, rollout_only, stream);
  }
}
}  // namespace

template <typename R>
bool control_wave_eligible(const ControlParams<R>& p, bool rollout_only)
{
  (void)rollout_only;
  return p.T >= 1 && p.T <= wave::kMaxS * kWave && p.K >= 1 && (p.K <= 16 || p.K == 20);
}

template <typename R>
hipError_t launch_control_wave(const ControlParams<R>& p, unsigned B, int model, bool rollout_only,
                               hipStream_t stream)
{
  if (B == 0) return hipSuccess;
  const bool stages = rollout_only || p.traj != nullptr || p.edx != nullptr || p.bdx != nullptr || p.rhot != nullptr;
  if (model == kModelOmni) {
    return stages ? launch_wave_k<R, kModelOmni, true>(p, B, rollout_only, stream)
                  : launch_wave_k<R, kModelOmni, false>(p, B, rollout_only, stream);
  }
  return stages ? launch_wave_k<R, kModelSimpleCart, true>(p, B, rollout_only, stream)
                : launch_wave_k<R, kModelSimpleCart, false>(p, B, rollout_only, stream);
}

// ---- the resident single-robot wavefront (RESIDENT instances: fp64, one slot, K = 5 / 10 / any K <= 16) ---------------------
namespace
{
template <int MODEL, int KC>
hipError_t launch_resident_one(const ControlParams<double>& p, hipStream_t stream)
{
  const size_t lds = wave::wave_lds_elems(KC) * sizeof(double);
  void (*kern)(const ControlParams<double>, const unsigned, const int, const int) =
      wave::control_wave_kernel<double, MODEL, KC, false, 1, true>;
  hipLaunchKernelGGL(kern, dim3(1), dim3(kWave), lds, stream, p, 1u, 1, 0);
  return hipGetLastError();
}
template <int MODEL>
hipError_t launch_resident_k(const ControlParams<double>& p, hipStream_t stream)
{
  switch (p.K) {
    case 5:
      return launch_resident_one<MODEL, 5>(p, stream);
    case 10:
      return launch_resident_one<MODEL, 10>(p, stream);
    default:
      return launch_resident_one<MODEL, 16>(p, stream);
  }
}
}  // namespace

bool control_wave_resident_eligible(const ControlParams<double>& p) { return p.T >= 1 && p.T <= kWave && p.K >= 1 && p.K <= 16; }

hipError_t launch_control_wave_resident(const ControlParams<double>& p, int model, hipStream_t stream)
{
  if (!control_wave_resident_eligible(p) || p.res_mail == nullptr) return hipErrorInvalidValue;
  return model == kModelOmni ? launch_resident_k<kModelOmni>(p, stream) : launch_resident_k<kModelSimpleCart>(p, stream);
}

template bool control_wave_eligible<double>(const ControlParams<double>&, bool);
template bool control_wave_eligible<float>(const ControlParams<float>&, bool);
template hipError_t launch_control_wave<double>(const ControlParams<double>&, unsigned, int, bool, hipStream_t);
template hipError_t launch_control_wave<float>(const ControlParams<float>&, unsigned, int, bool, hipStream_t);
}  // namespace eea
