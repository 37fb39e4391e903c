// Instantiations and dispatch of the fused control kernel (control_kernel_impl.hpp).
#include <cstdlib>

#include "control_kernel_impl.hpp"

namespace eea
{
namespace
{
template <typename R, int MODEL>
hipError_t launch_model(const ControlParams<R>& p, unsigned B, int Nmax, bool rollout_only,
                        size_t lds, hipStream_t stream)
{
  switch (p.K) {
    case 5:
      return launch_one<R, MODEL, 5>(p, B, Nmax, rollout_only, lds, stream);
    case 10:
      return launch_one<R, MODEL, 10>(p, B, Nmax, rollout_only, lds, stream);
    case 20:
      return launch_one<R, MODEL, 20>(p, B, Nmax, rollout_only, lds, stream);
    case 30:
      return launch_one<R, MODEL, 30>(p, B, Nmax, rollout_only, lds, stream);
    default:
      return launch_one<R, MODEL, 0>(p, B, Nmax, rollout_only, lds, stream);
  }
}
}  // namespace

template <typename R>
size_t control_lds_bytes(int T, int K, int n_mem_max, int /*chunk*/)
{
  return static_cast<size_t>(lds_layout(T, T + n_mem_max, K).total) * sizeof(R);
}

template <typename R>
hipError_t launch_control(const ControlParams<R>& p, unsigned B, int model, int n_mem_max,
                          bool rollout_only, hipStream_t stream)
{
  if (B == 0) return hipSuccess;
  const int Nmax = p.T + n_mem_max;
  size_t lds = control_lds_bytes<R>(p.T, p.K, n_mem_max, p.chunk);
  // occupancy experiment knob (tools/ab_bench.sh): extra dynamic LDS per workgroup in KiB
  static const int pad_kib = [] {
    const char* v = std::getenv("EEA_LDS_PAD_KIB");
    return v ? std::atoi(v) : 0;
  }();
  if (pad_kib > 0 && lds + static_cast<size_t>(pad_kib) * 1024 <= 160 * 1024) lds += static_cast<size_t>(pad_kib) * 1024;
  if (model == kModelOmni) return launch_model<R, kModelOmni>(p, B, Nmax, rollout_only, lds, stream);
  return launch_model<R, kModelSimpleCart>(p, B, Nmax, rollout_only, lds, stream);
}

template size_t control_lds_bytes<double>(int, int, int, int);
template size_t control_lds_bytes<float>(int, int, int, int);
template hipError_t launch_control<double>(const ControlParams<double>&, unsigned, int, int, bool,
                                           hipStream_t);
template hipError_t launch_control<float>(const ControlParams<float>&, unsigned, int, int, bool,
                                          hipStream_t);
}  // namespace eea
