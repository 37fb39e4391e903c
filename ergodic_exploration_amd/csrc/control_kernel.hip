// Instantiations and dispatch of the fused control kernel (control_kernel_impl.hpp).
#include "../../include/ergodic_amd.h"
#include "control_kernel_impl.hpp"

namespace eea
{
namespace
{
// Threads per agent: one lane per horizon step up to 256; horizons of at most 64 / 128 steps take one /
// two wavefronts per agent (no cross-wavefront exchange at all for one).  EEA_OPT_WORKGROUP_THREADS overrides
// (longer horizons then run several steps per lane through the chunk loop).
int control_threads(int T)
{
  const int forced = option(EEA_OPT_WORKGROUP_THREADS);
  if (forced) return forced;
  return T <= 64 ? 64 : (T <= 128 ? 128 : 256);
}

template <typename R, int MODEL, int BLK>
hipError_t launch_model(const ControlParams<R>& p, unsigned B, int Nmax, bool rollout_only,
                        size_t lds, hipStream_t stream)
{
  switch (p.K) {
    case 5:
      return launch_one<R, MODEL, 5, BLK>(p, B, Nmax, rollout_only, lds, stream);
    case 10:
      return launch_one<R, MODEL, 10, BLK>(p, B, Nmax, rollout_only, lds, stream);
    case 20:
      return launch_one<R, MODEL, 20, BLK>(p, B, Nmax, rollout_only, lds, stream);
    case 30:
      return launch_one<R, MODEL, 30, BLK>(p, B, Nmax, rollout_only, lds, stream);
    default:
      return launch_one<R, MODEL, 0, BLK>(p, B, Nmax, rollout_only, lds, stream);
  }
}

template <typename R, int MODEL>
hipError_t launch_block(const ControlParams<R>& p, unsigned B, int Nmax, bool rollout_only,
                        size_t lds, hipStream_t stream)
{
  switch (control_threads(p.T)) {
    case 64:
      return launch_model<R, MODEL, 64>(p, B, Nmax, rollout_only, lds, stream);
    case 128:
      return launch_model<R, MODEL, 128>(p, B, Nmax, rollout_only, lds, stream);
    default:
      return launch_model<R, MODEL, 256>(p, B, Nmax, rollout_only, lds, stream);
  }
}
}  // namespace

template <typename R>
size_t control_lds_bytes(int T, int K, int n_mem_max, int /*chunk*/)
{
  return static_cast<size_t>(lds_layout(T, T + n_mem_max, K, control_threads(T) / kWave).total) * sizeof(R);
}

template <typename R>
hipError_t launch_control(const ControlParams<R>& p, unsigned B, int model, int n_mem_max,
                          bool rollout_only, hipStream_t stream)
{
  if (B == 0) return hipSuccess;
  const int Nmax = p.T + n_mem_max;
  const size_t lds = control_lds_bytes<R>(p.T, p.K, n_mem_max, p.chunk);
  if (model == kModelOmni) return launch_block<R, kModelOmni>(p, B, Nmax, rollout_only, lds, stream);
  return launch_block<R, kModelSimpleCart>(p, B, Nmax, rollout_only, lds, stream);
}

template size_t control_lds_bytes<double>(int, int, int, int);
template size_t control_lds_bytes<float>(int, int, int, int);
template hipError_t launch_control<double>(const ControlParams<double>&, unsigned, int, int, bool,
                                           hipStream_t);
template hipError_t launch_control<float>(const ControlParams<float>&, unsigned, int, int, bool,
                                          hipStream_t);
}  // namespace eea
