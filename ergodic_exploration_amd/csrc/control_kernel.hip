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

// ---- sum of the per-agent records (eea_ck_records_sum) ---------------------------------------------------------------
namespace
{
// One workgroup of 256 threads per group of kSumGroup = 64 agents: thread (h, m) adds element m of the records of the
// group's agents 32 h .. 32 h + 31 in agent order -- all 32 loads in flight at once: the records were just written by
// control kernels on other XCDs, every load is a trip to memory -- and the two halves are added in LDS (h = 0 first).
// The group record goes to the workspace with agent-scope (write-through) stores; the last workgroup to arrive (one
// ticket, atomicInc wraps at the group count: it resets itself) adds the group records in group order, 64 loads in
// flight, and writes the result: one launch, fixed summation order.  Hand-off protocol: MI355X_MICROARCH.md
// (write-through payload -> s_waitcnt vmcnt(0) -> agent-scope ticket; the reader's loads bypass its L1).
constexpr int kSumThreads = 256, kSumHalf = kSumGroup / 2;
template <typename R>
__global__ __launch_bounds__(kSumThreads) void ck_records_sum_kernel(const R* __restrict__ rec, unsigned B, int rec_len,
                                                                      R* groups, unsigned* ctr, R* __restrict__ out)
{
  __shared__ unsigned s_old;
  __shared__ R s_half[kSumThreads / 2];
  const unsigned ngroups = gridDim.x, g = blockIdx.x;
  const unsigned first = g * kSumGroup, n = (B - first) < kSumGroup ? (B - first) : kSumGroup;
  const int h = threadIdx.x / (kSumThreads / 2), lane_m = threadIdx.x % (kSumThreads / 2);
  for (int m0 = 0; m0 < rec_len; m0 += kSumThreads / 2) {
    const int m = m0 + lane_m;
    R acc = R(0);
    if (m < rec_len) {
      const R* const col = rec + (static_cast<size_t>(first) + kSumHalf * h) * rec_len + m;
      const unsigned nh = n > kSumHalf * static_cast<unsigned>(h) ? n - kSumHalf * h : 0u;  // agents of this half
      R v[kSumHalf];
#pragma unroll
      for (int i = 0; i < kSumHalf; ++i) v[i] = (static_cast<unsigned>(i) < nh) ? col[static_cast<size_t>(i) * rec_len] : R(0);
#pragma unroll
      for (int i = 0; i < kSumHalf; ++i) acc += v[i];
    }
    if (m0 > 0) __syncthreads();  // the previous round's halves have been read
    if (h == 1) s_half[lane_m] = acc;
    __syncthreads();
    if (h == 0 && m < rec_len) {
      acc += s_half[lane_m];
      if (ngroups == 1) out[m] = acc;
      else store_agent(groups + static_cast<size_t>(g) * rec_len + m, acc);
    }
  }
  if (ngroups == 1) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // every thread's part of the group record has left
  if (threadIdx.x == 0) s_old = atomicInc(ctr, ngroups - 1);
  __syncthreads();
  if (s_old != ngroups - 1) return;
  for (int m = threadIdx.x; m < rec_len; m += kSumThreads) {
    R acc = R(0);
    unsigned q = 0;
    for (; q + 64 <= ngroups; q += 64) {
      R v[64];
#pragma unroll
      for (int i = 0; i < 64; ++i) v[i] = load_agent(groups + static_cast<size_t>(q + i) * rec_len + m);
#pragma unroll
      for (int i = 0; i < 64; ++i) acc += v[i];
    }
    for (; q + 8 <= ngroups; q += 8) {
      R v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = load_agent(groups + static_cast<size_t>(q + i) * rec_len + m);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    for (; q < ngroups; ++q) acc += load_agent(groups + static_cast<size_t>(q) * rec_len + m);
    out[m] = acc;
  }
}
}  // namespace

template <typename R>
hipError_t launch_ck_records_sum(const R* d_rec, unsigned B, int K2, R* d_ws, unsigned* d_ctr, R* d_out, hipStream_t stream)
{
  if (B == 0) return hipSuccess;
  const unsigned ngroups = (B + kSumGroup - 1) / kSumGroup;
  if (const hipEvent_t stop = take_stop_event()) {
    hipExtLaunchKernelGGL(ck_records_sum_kernel<R>, dim3(ngroups), dim3(kSumThreads), 0, stream, nullptr, stop, 0, d_rec, B,
                          ck_record_len(K2), d_ws, d_ctr, d_out);
  } else {
    hipLaunchKernelGGL(ck_records_sum_kernel<R>, dim3(ngroups), dim3(kSumThreads), 0, stream, d_rec, B, ck_record_len(K2),
                       d_ws, d_ctr, d_out);
  }
  return hipGetLastError();
}

template hipError_t launch_ck_records_sum<double>(const double*, unsigned, int, double*, unsigned*, double*, hipStream_t);
template hipError_t launch_ck_records_sum<float>(const float*, unsigned, int, float*, unsigned*, float*, hipStream_t);
template size_t control_lds_bytes<double>(int, int, int, int);
template size_t control_lds_bytes<float>(int, int, int, int);
template hipError_t launch_control<double>(const ControlParams<double>&, unsigned, int, int, bool,
                                           hipStream_t);
template hipError_t launch_control<float>(const ControlParams<float>&, unsigned, int, int, bool,
                                          hipStream_t);
}  // namespace eea
