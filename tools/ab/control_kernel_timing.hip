// Diagnostic build of the workgroup-per-agent control kernel with phase stamps: K = 10 and 30, fp64 only.
// Not used by the product path; reached through eea_debug_phase_timing (engine_ab.inc).
// Each wavefront's lane 0 records the shader clock at the phase boundaries into p.dbg [agent][4 waves][16].
#define EEA_STAMP(n)                                                                            \
  do {                                                                                          \
    if (p.dbg != nullptr && (threadIdx.x & 63) == 0)                                            \
      p.dbg[(static_cast<size_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * 16 + (n)] =            \
          static_cast<long long>(__builtin_readcyclecounter());                                 \
  } while (0)
#include "control_kernel_impl.hpp"

namespace eea
{
hipError_t launch_control_timing(const ControlParams<double>& p, unsigned B, int model, int n_mem_max,
                                 hipStream_t stream)
{
  if (B == 0) return hipSuccess;
  if (p.K != 10 && p.K != 30) return hipErrorInvalidValue;
  const int Nmax = p.T + n_mem_max;
  const size_t lds = static_cast<size_t>(lds_layout(p.T, Nmax, p.K, 4).total) * sizeof(double);
  if (p.K == 30) {  // the BASELINE config 5 shape
    if (model == kModelOmni) return launch_one<double, kModelOmni, 30, 256>(p, B, Nmax, false, lds, stream);
    return launch_one<double, kModelSimpleCart, 30, 256>(p, B, Nmax, false, lds, stream);
  }
  if (model == kModelOmni) return launch_one<double, kModelOmni, 10, 256>(p, B, Nmax, false, lds, stream);
  return launch_one<double, kModelSimpleCart, 10, 256>(p, B, Nmax, false, lds, stream);
}
}  // namespace eea
