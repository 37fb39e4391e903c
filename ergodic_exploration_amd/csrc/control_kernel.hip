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
template <typename R, int MODEL, int BLK>
hipError_t resident_model(const ControlParams<R>& p, int Nmax, size_t lds, void* mail, void* stage, unsigned first_seen,
                          long long idle_ticks, hipStream_t stream)
{
  switch (p.K) {
    case 5:
      return launch_resident_one<R, MODEL, 5, BLK>(p, Nmax, lds, mail, stage, first_seen, idle_ticks, stream);
    case 10:
      return launch_resident_one<R, MODEL, 10, BLK>(p, Nmax, lds, mail, stage, first_seen, idle_ticks, stream);
    case 20:
      return launch_resident_one<R, MODEL, 20, BLK>(p, Nmax, lds, mail, stage, first_seen, idle_ticks, stream);
    case 30:
      return launch_resident_one<R, MODEL, 30, BLK>(p, Nmax, lds, mail, stage, first_seen, idle_ticks, stream);
    default:
      return launch_resident_one<R, MODEL, 0, BLK>(p, Nmax, lds, mail, stage, first_seen, idle_ticks, stream);
  }
}
template <typename R, int MODEL>
hipError_t resident_block(const ControlParams<R>& p, int Nmax, size_t lds, void* mail, void* stage, unsigned first_seen,
                          long long idle_ticks, hipStream_t stream)
{
  switch (control_threads(p.T)) {
    case 64:
      return resident_model<R, MODEL, 64>(p, Nmax, lds, mail, stage, first_seen, idle_ticks, stream);
    case 128:
      return resident_model<R, MODEL, 128>(p, Nmax, lds, mail, stage, first_seen, idle_ticks, stream);
    default:
      return resident_model<R, MODEL, 256>(p, Nmax, lds, mail, stage, first_seen, idle_ticks, stream);
  }
}
}  // namespace

// the resident single-robot workgroup (control_resident_kernel): p.pose / u0 / status / done / n_mem / mem_cols point into
// the host-mapped mailbox and replay-memory buffer; n_mem_max = the buffer's capacity in columns
template <typename R>
hipError_t launch_control_resident(const ControlParams<R>& p, int model, int n_mem_max, void* d_mail, void* d_stage,
                                   unsigned first_seen, long long idle_ticks, hipStream_t stream)
{
  const int Nmax = p.T + n_mem_max;
  const size_t lds = control_lds_bytes<R>(p.T, p.K, n_mem_max, p.chunk);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (model == kModelOmni) return resident_block<R, kModelOmni>(p, Nmax, lds, d_mail, d_stage, first_seen, idle_ticks, stream);
  return resident_block<R, kModelSimpleCart>(p, Nmax, lds, d_mail, d_stage, first_seen, idle_ticks, stream);
}
template hipError_t launch_control_resident<double>(const ControlParams<double>&, int, int, void*, void*, unsigned, long long, hipStream_t);
template hipError_t launch_control_resident<float>(const ControlParams<float>&, int, int, void*, void*, unsigned, long long, hipStream_t);

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
// One WAVEFRONT per unit (g0, e): the elements [64 e, 64 e + 64) of the records of the kSumGroup = 32 agents of group
// g0, added in agent order, 8 loads in flight (the records were just written by control kernels on other XCDs: every
// load is a trip to memory).  A tree of tickets finishes the sum in the same launch, per element slice e: the last
// of the kSumFan = 8 units of a level-1 group adds their records in group order, the last level-1 group adds the
// level-1 records in order -- fixed summation order whatever the arrival order.  Tickets are atomicInc with the
// member count as the wrap value: they reset themselves.  Hand-off protocol: MI355X_MICROARCH.md (write-through
// payload -> s_waitcnt vmcnt(0) -> agent-scope ticket; the reader's loads bypass its L1).
//
// No LDS, one wavefront per workgroup and <= 32 VGPRs (the request is in register pairs on gfx90a+): the wavefronts of
// this kernel fit BESIDE a fully resident fp64 K <= 10 control kernel (4 wavefronts x 120 VGPRs per SIMD, all of the
// LDS) instead of waiting for one of its workgroups to retire -- the exchange of a consensus pass runs under the
// control kernels of the next passes.
constexpr int kSumBatch = 8;

// the record slice of up to `n` rows of `rows` (row stride rec_len), added in row order; AGENT: agent-scope loads
template <typename R, bool AGENT>
__device__ __forceinline__ R sum_rows(const R* rows, unsigned n, int rec_len, int m)
{
  R acc = R(0);
  unsigned q = 0;
#pragma unroll 1
  for (; q + kSumBatch <= n; q += kSumBatch) {
    R v[kSumBatch];
#pragma unroll
    for (int i = 0; i < kSumBatch; ++i) {
      const R* const src = rows + static_cast<size_t>(q + i) * rec_len;  // wavefront-uniform row pointer + lane offset
      v[i] = AGENT ? load_agent(src + m) : src[m];
    }
#pragma unroll
    for (int i = 0; i < kSumBatch; ++i) acc += v[i];
  }
#pragma unroll 1
  for (; q < n; ++q) {
    const R* const src = rows + static_cast<size_t>(q) * rec_len;
    acc += AGENT ? load_agent(src + m) : src[m];
  }
  return acc;
}

// the wavefront's payload has left, then one ticket: true for the last of `members` arrivals
__device__ __forceinline__ bool last_arrival(unsigned* ticket, unsigned members)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned old = 0;
  if (threadIdx.x == 0) old = atomicInc(ticket, members - 1);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old != members - 1) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

// Device-bound form (ready != nullptr; eea_ck_records_sum_bound): the launch does not wait for the control kernels that
// write the records -- every unit polls the ready marks of ITS 32 agents (rec_ready[b] == seq, written behind the agent's
// drained write-through record) and starts as soon as they are there, while the control kernels are still in their
// backward halves; the wavefront that completes the last slice of the sum publishes flag = seq behind the drained sum
// record, which is what the consuming control kernels wait for (ControlParams::ck_flag).  A unit whose agents never
// report (about a second) makes the record's agent count negative: consumers then keep their own c_k and report
// EEA_ERR_TIMEOUT.
template <typename R>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_num_vgpr(16))) void ck_records_sum_kernel(
    const R* __restrict__ rec, unsigned B, int rec_len, int K2, R* ws, unsigned* ctr, R* __restrict__ out,
    const unsigned* ready, unsigned seq, unsigned* flag)
{
  const unsigned ng0 = gridDim.x, ng1 = (ng0 + kSumFan - 1) / kSumFan;
  const unsigned g0 = blockIdx.x, e = blockIdx.y;
  const int m = static_cast<int>(kWave * e + threadIdx.x);
  const bool active = m < rec_len;
  const int mm = active ? m : 0;  // inactive lanes read element 0 and write nothing
  R* const ws0 = ws;                                        // [ng0][rec_len]
  R* const ws1 = ws + static_cast<size_t>(ng0) * rec_len;   // [ng1][rec_len]
  unsigned* const tickets = ctr + static_cast<size_t>(e) * (ng1 + 1);  // [ng1] level-1 tickets, then the level-2 ticket
  unsigned* const slice_ticket = ctr + static_cast<size_t>(gridDim.y) * (ng1 + 1);  // the slices of the finished sum

  const unsigned first = g0 * kSumGroup, n0 = (B - first) < kSumGroup ? (B - first) : kSumGroup;
  const bool bound = ready != nullptr;
  bool timed_out = false;
  if (bound) {
    for (int i = 0;; ++i) {
      const bool ok = threadIdx.x >= n0 || static_cast<int>(load_agent(ready + first + threadIdx.x) - seq) >= 0;
      if (__all(ok)) break;
      if (i >= kFlagPolls) {
        timed_out = true;
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
  R acc = bound ? sum_rows<R, true>(rec + static_cast<size_t>(first) * rec_len, n0, rec_len, mm)
                : sum_rows<R, false>(rec + static_cast<size_t>(first) * rec_len, n0, rec_len, mm);
  if (timed_out && m == K2) acc = R(-1.0e9);  // the agent count of the finished record goes negative
  // the finished slice: out, and -- device-bound form -- the flag behind the last slice
  auto finish = [&](R v) {
    if (flag == nullptr) {
      if (active) out[m] = v;
      return;
    }
    if (active) store_agent(out + m, v);
    if (last_arrival(slice_ticket, gridDim.y) && threadIdx.x == 0) store_agent(flag, seq);
  };
  if (ng0 == 1) {
    finish(acc);
    return;
  }
  if (active) store_agent(ws0 + static_cast<size_t>(g0) * rec_len + m, acc);
  const unsigned g1 = g0 / kSumFan, n1 = (ng0 - g1 * kSumFan) < kSumFan ? (ng0 - g1 * kSumFan) : kSumFan;
  if (!last_arrival(tickets + g1, n1)) return;

  acc = sum_rows<R, true>(ws0 + static_cast<size_t>(g1) * kSumFan * rec_len, n1, rec_len, mm);
  if (ng1 == 1) {
    finish(acc);
    return;
  }
  if (active) store_agent(ws1 + static_cast<size_t>(g1) * rec_len + m, acc);
  if (!last_arrival(tickets + ng1, ng1)) return;

  acc = sum_rows<R, true>(ws1, ng1, rec_len, mm);
  finish(acc);
}

// d_pub [n] = d_src [n] written through (sc1), then *flag = seq: makes a record another kernel produced with plain stores
// (the all-reduce of the ranks' sum records) visible to control kernels that are already running and wait for the flag
template <typename R>
__global__ __launch_bounds__(kBlock) void publish_record_kernel(const R* __restrict__ src, int n, R* pub, unsigned* flag,
                                                               unsigned seq)
{
  for (int i = threadIdx.x; i < n; i += kBlock) store_agent(pub + i, src[i]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) store_agent(flag, seq);
}
}  // namespace

template <typename R>
hipError_t launch_ck_records_sum(const R* d_rec, unsigned B, int K2, R* d_ws, unsigned* d_ctr, R* d_out, hipStream_t stream,
                                 const unsigned* d_ready, unsigned seq, unsigned* d_flag)
{
  if (B == 0) return hipSuccess;
  const dim3 grid(ck_sum_groups(B), ck_sum_slices(K2));
  hipLaunchKernelGGL(ck_records_sum_kernel<R>, grid, dim3(kWave), 0, stream, d_rec, B, ck_record_len(K2), K2, d_ws, d_ctr,
                     d_out, d_ready, seq, d_flag);
  return hipGetLastError();
}

template <typename R>
hipError_t launch_publish_record(const R* d_src, int n, R* d_pub, unsigned* d_flag, unsigned seq, hipStream_t stream)
{
  hipLaunchKernelGGL(publish_record_kernel<R>, dim3(1), dim3(kBlock), 0, stream, d_src, n, d_pub, d_flag, seq);
  return hipGetLastError();
}

template hipError_t launch_publish_record<double>(const double*, int, double*, unsigned*, unsigned, hipStream_t);
template hipError_t launch_publish_record<float>(const float*, int, float*, unsigned*, unsigned, hipStream_t);
template hipError_t launch_ck_records_sum<double>(const double*, unsigned, int, double*, unsigned*, double*, hipStream_t,
                                                  const unsigned*, unsigned, unsigned*);
template hipError_t launch_ck_records_sum<float>(const float*, unsigned, int, float*, unsigned*, float*, hipStream_t,
                                                 const unsigned*, unsigned, unsigned*);
template size_t control_lds_bytes<double>(int, int, int, int);
template size_t control_lds_bytes<float>(int, int, int, int);
template hipError_t launch_control<double>(const ControlParams<double>&, unsigned, int, int, bool,
                                           hipStream_t);
template hipError_t launch_control<float>(const ControlParams<float>&, unsigned, int, int, bool,
                                          hipStream_t);
}  // namespace eea
