// Exchange steps of the agent-batched mode (include/ergodic_amd.h, "multi-GPU exchange"): the reductions of
// the per-agent trajectory coefficients c_k on the device and their collectives over RCCL / xGMI.
//
// The reference is single-agent (no collective anywhere); the semantics come from the decentralised ergodic
// control its README cites as ref. [2] (README.md:225-227): agents share c_k.  Two forms:
//   * all-gather of every agent's c_k (K^2 reals per agent), the exchange north_star names;
//   * consensus c_bar = mean over all agents of c_k: one all-reduce of K^2 + 1 reals (the agent count rides
//     along), which is all the gradient needs (eea_batch_io::d_ck_shared).
// RCCL is bound at run time (dlopen): a process that already carries an RCCL (PyTorch's) shares that
// instance, a plain C++ host gets /opt/rocm's; libergodic_amd.so itself has no link dependency on it.
#include "../../include/ergodic_amd.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstring>
#include <string>
#include <vector>

#include "abi_util.hpp"
#include "common.hpp"

namespace
{
struct RcclApi
{
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

// eea_comm_set_library: the collective library to bind instead of the process's / the default one
std::string g_library_path;
std::atomic<bool> g_bound{ false };

RcclApi& rccl()
{
  static RcclApi api = [] {
    RcclApi a;
    g_bound.store(true);
    if (!g_library_path.empty()) {
      // by path, with its own symbol scope: an RCCL of the same soname may already be mapped (PyTorch's)
      a.handle = dlopen(g_library_path.c_str(), RTLD_NOW | RTLD_LOCAL);
    } else {
      // an RCCL already mapped into the process (e.g. PyTorch's bundled one) is reused
      const char* names[] = { "librccl.so", "librccl.so.1" };
      for (const char* n : names) {
        a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (a.handle) break;
      }
      for (int i = 0; a.handle == nullptr && i < 2; ++i) a.handle = dlopen(names[1 - i], RTLD_NOW | RTLD_GLOBAL);
    }
    if (a.handle == nullptr) return a;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.handle, "ncclAllGather"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce && a.GetErrorString;
    return a;
  }();
  return api;
}

using eea::fail;

#define EEA_RCCL(expr)                                                                        \
  do {                                                                                        \
    const ncclResult_t err__ = (expr);                                                        \
    if (err__ != ncclSuccess) {                                                               \
      return fail(EEA_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(err__));     \
    }                                                                                         \
  } while (0)

// sums[m] = sum over the B local agents of ck[b][m] (fixed order: run-to-run deterministic);
// sums[K2] = B.  One workgroup per 64 modes would starve the chip for K = 10, so: one wavefront per mode,
// lanes stride the agents, DPP-free butterfly through __shfl_xor (cold path: once per exchange).
template <typename R>
__global__ __launch_bounds__(256) void ck_sum_kernel(const R* __restrict__ ck, unsigned B, int K2,
                                                     R* __restrict__ sums)
{
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m > K2) return;
  if (m == K2) {
    if (lane == 0) sums[K2] = static_cast<R>(B);
    return;
  }
  R acc = R(0);
  for (unsigned b = lane; b < B; b += 64) acc += ck[static_cast<size_t>(b) * K2 + m];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) sums[m] = acc;
}

// c_bar[m] = sums[m] / sums[K2]
template <typename R>
__global__ __launch_bounds__(256) void ck_mean_kernel(const R* __restrict__ sums, int K2, R* __restrict__ out)
{
  const R n = sums[K2];
  for (int m = blockIdx.x * 256 + threadIdx.x; m < K2; m += gridDim.x * 256) out[m] = sums[m] / n;
}

template <typename R>
hipError_t launch_ck_sum(const void* d_ck, unsigned B, int K2, void* d_sums, hipStream_t s)
{
  hipLaunchKernelGGL(ck_sum_kernel<R>, dim3((K2 + 1 + 3) / 4), dim3(256), 0, s, static_cast<const R*>(d_ck), B, K2,
                     static_cast<R*>(d_sums));
  return hipGetLastError();
}
template <typename R>
hipError_t launch_ck_mean(const void* d_sums, int K2, void* d_out, hipStream_t s)
{
  hipLaunchKernelGGL(ck_mean_kernel<R>, dim3((K2 + 255) / 256), dim3(256), 0, s, static_cast<const R*>(d_sums), K2,
                     static_cast<R*>(d_out));
  return hipGetLastError();
}
}  // namespace

// Events that only order device streams among each other: no timing, and NO system-scope fence when they complete.
// The default event makes the kernel in front of it write the device's caches back for the host's benefit -- behind a
// control kernel that is 20 MB of dirty controls per completion (measured: tools/ck_cost.py).
constexpr unsigned kDeviceEvent = hipEventDisableTiming | hipEventDisableSystemFence;

struct eea_comm
{
  ncclComm_t comm = nullptr;
  int device = 0, nranks = 1, rank = 0;
  void* d_sums = nullptr;  // K^2 + 1 reals of scratch for the consensus reduction
  size_t sums_cap = 0;
  // asynchronous form: an exchange stream of its own beside the caller's compute stream, completion per slot
  hipStream_t xstream = nullptr;
  hipEvent_t ev_in = nullptr;
  hipEvent_t ev_done[EEA_COMM_SLOTS] = {};
  // completion of the agent groups' control launches of a pass (eea_comm_records_exchange_async records them)
  static constexpr unsigned kMaxGroups = 8;
  hipEvent_t ev_group[EEA_COMM_SLOTS][kMaxGroups] = {};
  // device-bound exchange with more than one rank: per slot, the sum record the ranks all-reduce (the control kernels
  // read the PUBLISHED copy, eea_comm_records_exchange_bound)
  void* d_xrec[EEA_COMM_SLOTS] = {};
  size_t xrec_cap[EEA_COMM_SLOTS] = {};
  std::vector<void*> retired;
};

namespace
{
eea_status sums_reserve(eea_comm* c, size_t bytes)
{
  if (bytes <= c->sums_cap) return EEA_OK;
  if (c->d_sums) (void)hipFree(c->d_sums);
  c->d_sums = nullptr;
  c->sums_cap = 0;
  EEA_HIP(hipMalloc(&c->d_sums, bytes));
  c->sums_cap = bytes;
  return EEA_OK;
}
}  // namespace

extern "C" {

eea_status eea_comm_set_library(const char* path)
{
  if (path == nullptr || path[0] == '\0') return fail(EEA_ERR_INVALID_ARGUMENT, "null / empty library path");
  if (g_bound.load()) return fail(EEA_ERR_UNSUPPORTED, "the collective library of this process is already bound");
  g_library_path = path;
  return EEA_OK;
}

eea_status eea_comm_get_unique_id(void* id)
{
  static_assert(sizeof(ncclUniqueId) == EEA_COMM_ID_BYTES, "ncclUniqueId size");
  if (id == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null id");
  if (!rccl().ok) return fail(EEA_ERR_HIP, "librccl.so not loadable");
  ncclUniqueId u;
  EEA_RCCL(rccl().GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return EEA_OK;
}

eea_status eea_comm_create(int device, int nranks, int rank, const void* id, eea_comm** out)
{
  if (out == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(EEA_ERR_INVALID_ARGUMENT, "bad rank / nranks");
  if (nranks > 1 && id == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null id");
  eea_comm* c = new eea_comm();
  c->device = device;
  c->nranks = nranks;
  c->rank = rank;
  if (id != nullptr) {
    // a real RCCL communicator (also for one rank: the single-GPU tests drive the collectives through it);
    // one rank per GPU (RCCL refuses two ranks of one communicator on the same device)
    if (!rccl().ok) {
      delete c;
      return fail(EEA_ERR_HIP, "librccl.so not loadable");
    }
    const hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) {
      delete c;
      return fail(EEA_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
    }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    const ncclResult_t r = rccl().CommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
      const std::string msg = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r);
      delete c;
      return fail(EEA_ERR_HIP, msg);
    }
  }
  *out = c;
  return EEA_OK;
}

void eea_comm_destroy(eea_comm* c)
{
  if (c == nullptr) return;
  (void)hipSetDevice(c->device);
  if (c->xstream) {
    (void)hipStreamSynchronize(c->xstream);
    (void)hipStreamDestroy(c->xstream);
  }
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  for (hipEvent_t ev : c->ev_done) {
    if (ev) (void)hipEventDestroy(ev);
  }
  for (auto& row : c->ev_group) {
    for (hipEvent_t ev : row) {
      if (ev) (void)hipEventDestroy(ev);
    }
  }
  if (c->comm != nullptr && rccl().ok) (void)rccl().CommDestroy(c->comm);
  if (c->d_sums) (void)hipFree(c->d_sums);
  for (void* q : c->d_xrec) {
    if (q) (void)hipFree(q);
  }
  for (void* q : c->retired) (void)hipFree(q);
  delete c;
}

int eea_comm_rank(const eea_comm* c) { return c ? c->rank : 0; }
int eea_comm_nranks(const eea_comm* c) { return c ? c->nranks : 1; }

eea_status eea_ck_sum(eea_engine* e, unsigned B, const void* d_ck, void* d_sums, void* stream)
{
  if (e == nullptr || d_ck == nullptr || d_sums == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  const int K2 = static_cast<int>(eea_num_modes(e));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (eea_real_size(e) == 4) EEA_HIP(launch_ck_sum<float>(d_ck, B, K2, d_sums, s));
  else EEA_HIP(launch_ck_sum<double>(d_ck, B, K2, d_sums, s));
  return EEA_OK;
}

eea_status eea_comm_allgather_ck(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_local,
                                 void* d_ck_all, void* stream)
{
  if (e == nullptr || d_ck_local == nullptr || d_ck_all == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  const size_t K2 = eea_num_modes(e), rs = eea_real_size(e);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t count = static_cast<size_t>(B_local) * K2;
  if (c == nullptr || c->comm == nullptr) {
    if (d_ck_all != d_ck_local) EEA_HIP(hipMemcpyAsync(d_ck_all, d_ck_local, count * rs, hipMemcpyDeviceToDevice, s));
    return EEA_OK;
  }
  EEA_RCCL(rccl().AllGather(d_ck_local, d_ck_all, count, rs == 4 ? ncclFloat32 : ncclFloat64, c->comm, s));
  return EEA_OK;
}

eea_status eea_comm_allreduce_sum(eea_engine* e, eea_comm* c, void* d_buf, unsigned n, void* stream)
{
  if (e == nullptr || d_buf == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  if (c == nullptr || c->comm == nullptr || n == 0) return EEA_OK;
  EEA_RCCL(rccl().AllReduce(d_buf, d_buf, n, eea_real_size(e) == 4 ? ncclFloat32 : ncclFloat64, ncclSum, c->comm,
                            static_cast<hipStream_t>(stream)));
  return EEA_OK;
}

eea_status eea_comm_consensus_ck(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_local,
                                 void* d_ck_shared, void* stream)
{
  if (e == nullptr || c == nullptr || d_ck_local == nullptr || d_ck_shared == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  }
  const int K2 = static_cast<int>(eea_num_modes(e));
  const size_t rs = eea_real_size(e);
  hipStream_t s = static_cast<hipStream_t>(stream);
  eea_status st = sums_reserve(c, rs * (static_cast<size_t>(K2) + 1));
  if (st != EEA_OK) return st;
  st = eea_ck_sum(e, B_local, d_ck_local, c->d_sums, stream);
  if (st != EEA_OK) return st;
  st = eea_comm_allreduce_sum(e, c, c->d_sums, static_cast<unsigned>(K2 + 1), stream);
  if (st != EEA_OK) return st;
  if (rs == 4) EEA_HIP(launch_ck_mean<float>(c->d_sums, K2, d_ck_shared, s));
  else EEA_HIP(launch_ck_mean<double>(c->d_sums, K2, d_ck_shared, s));
  return EEA_OK;
}

}  // extern "C"

namespace
{
eea_status async_begin(eea_comm* c, void* compute_stream, int slot, bool order_after_compute = true)
{
  if (c == nullptr || slot < 0 || slot >= EEA_COMM_SLOTS) return fail(EEA_ERR_INVALID_ARGUMENT, "bad communicator / slot");
  EEA_HIP(hipSetDevice(c->device));
  if (c->xstream == nullptr) {
    // highest priority: the exchange steps are a handful of wavefronts that must get the first execution slots a
    // finishing control kernel frees, not queue behind the next one's 2048 (measured: tools/ck_cost.py)
    int least = 0, greatest = 0;
    EEA_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    EEA_HIP(hipStreamCreateWithPriority(&c->xstream, hipStreamNonBlocking, greatest));
    EEA_HIP(hipEventCreateWithFlags(&c->ev_in, kDeviceEvent));
  }
  if (c->ev_done[slot] == nullptr) EEA_HIP(hipEventCreateWithFlags(&c->ev_done[slot], kDeviceEvent));
  if (!order_after_compute) return EEA_OK;
  // everything enqueued on the compute stream so far (the pass that produced c_k) comes first
  EEA_HIP(hipEventRecord(c->ev_in, static_cast<hipStream_t>(compute_stream)));
  EEA_HIP(hipStreamWaitEvent(c->xstream, c->ev_in, 0));
  return EEA_OK;
}
}  // namespace

extern "C" {

eea_status eea_comm_consensus_ck_async(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_local,
                                       void* d_ck_shared, void* compute_stream, int slot)
{
  eea_status st = async_begin(c, compute_stream, slot);
  if (st != EEA_OK) return st;
  st = eea_comm_consensus_ck(e, c, B_local, d_ck_local, d_ck_shared, c->xstream);
  if (st != EEA_OK) return st;
  EEA_HIP(hipEventRecord(c->ev_done[slot], c->xstream));
  return EEA_OK;
}

eea_status eea_comm_allgather_ck_async(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_local,
                                       void* d_ck_all, void* compute_stream, int slot)
{
  eea_status st = async_begin(c, compute_stream, slot);
  if (st != EEA_OK) return st;
  st = eea_comm_allgather_ck(e, c, B_local, d_ck_local, d_ck_all, c->xstream);
  if (st != EEA_OK) return st;
  EEA_HIP(hipEventRecord(c->ev_done[slot], c->xstream));
  return EEA_OK;
}

eea_status eea_comm_allreduce_sum_async(eea_engine* e, eea_comm* c, void* d_buf, unsigned n, void* compute_stream,
                                        int slot)
{
  eea_status st = async_begin(c, compute_stream, slot);
  if (st != EEA_OK) return st;
  st = eea_comm_allreduce_sum(e, c, d_buf, n, c->xstream);
  if (st != EEA_OK) return st;
  EEA_HIP(hipEventRecord(c->ev_done[slot], c->xstream));
  return EEA_OK;
}

}  // extern "C"

extern "C" {

// The exchange of one pass, stream-ordered: the exchange stream waits for everything enqueued so far on each group
// stream (one event per group), then record sum, then the all-reduce of the sum record over the ranks; the consuming
// streams eea_comm_wait for the slot.  Correct by construction and the form AgentBatch-style hosts use at their control
// rate; a consensus EVERY pass at the device's own rate is eea_comm_records_exchange_bound.
eea_status eea_comm_records_exchange_async(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_rec,
                                           void* d_sum, void* const* group_streams, unsigned n_streams, int slot)
{
  if (e == nullptr || d_ck_rec == nullptr || d_sum == nullptr || (n_streams > 0 && group_streams == nullptr)) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  }
  if (c == nullptr || slot < 0 || slot >= EEA_COMM_SLOTS) return fail(EEA_ERR_INVALID_ARGUMENT, "bad communicator / slot");
  if (n_streams > eea_comm::kMaxGroups) return fail(EEA_ERR_INVALID_ARGUMENT, "too many group streams");
  eea_status st = async_begin(c, nullptr, slot, false);
  if (st != EEA_OK) return st;
  for (unsigned g = 0; g < n_streams; ++g) {
    if (c->ev_group[slot][g] == nullptr) EEA_HIP(hipEventCreateWithFlags(&c->ev_group[slot][g], kDeviceEvent));
    EEA_HIP(hipEventRecord(c->ev_group[slot][g], static_cast<hipStream_t>(group_streams[g])));
    EEA_HIP(hipStreamWaitEvent(c->xstream, c->ev_group[slot][g], 0));
  }
  st = eea_ck_records_sum(e, B_local, d_ck_rec, d_sum, c->xstream);
  if (st != EEA_OK) return st;
  st = eea_comm_allreduce_sum(e, c, d_sum, eea_ck_record_len(e), c->xstream);  // (nothing without an RCCL communicator)
  if (st != EEA_OK) return st;
  EEA_HIP(hipEventRecord(c->ev_done[slot], c->xstream));
  return EEA_OK;
}

// The exchange of one pass, DEVICE-BOUND (ABI 4): nothing is ordered by the host or by stream waits.  On the
// communicator's own highest-priority stream: the record sum that polls the agents' ready marks (eea_ck_records_sum_bound:
// it starts while the producing control kernels are still in their backward halves), with more than one rank (or a
// one-rank RCCL communicator) the all-reduce of the sum record over the ranks and one small launch that publishes the
// result write-through, and *d_flag = seq behind it.  The control kernels that consume d_sum are launched WITHOUT
// waiting -- eea_batch_io::d_ck_flag / ck_flag_seq make them wait inside the kernel, right before the first use of the
// shared c_k (~45 % into the wavefront's lifetime).  Caller's duties: rotate d_ck_rec / d_sum over >= 3 buffers (slot =
// the buffer index; d_rec_ready and d_flag may be shared by all buffers: sequence numbers only grow), and keep every
// batch that consumes a flag small enough that the producers it waits for can be resident beside it (two agent groups
// per GPU are: each holds half of the execution slots) -- a consumer that cannot be served gives up after about a
// second with EEA_ERR_TIMEOUT in d_status and its own c_k.
// WITH A COLLECTIVE IN THE EXCHANGE (a communicator of more than one rank) "room for the producers" includes the collective
// kernel: a block of 256-512 threads with ~100 registers and LDS of its own does not fit beside a full set of control
// wavefronts (4 x 120 of a SIMD's 512 registers), and when every execution slot is held by control wavefronts that wait for
// the flag it produces, nothing ever frees one (round 5: measured with two ranks on one GPU and a stream-asynchronous test
// double of a realistic footprint, tests/fake_rccl -- every agent timed out).  Letting only ONE of a rank's agent groups
// consume the flag device-bound (the other ordered behind the exchange with eea_comm_wait: the event is recorded here,
// behind the published record) removes the systematic dead-lock but still stalled once in a few thousand passes at exactly
// full occupancy.  The rule: with a communicator use eea_comm_records_exchange_async + eea_comm_wait for EVERY consuming group
// (nothing waits inside a kernel), at a lag of >= 2 passes; this form is for exchanges without a collective kernel.
eea_status eea_comm_records_exchange_bound(eea_engine* e, eea_comm* c, unsigned B_local, const void* d_ck_rec,
                                           const unsigned* d_rec_ready, unsigned seq, void* d_sum, unsigned* d_flag, int slot)
{
  if (e == nullptr || d_ck_rec == nullptr || d_rec_ready == nullptr || d_sum == nullptr || d_flag == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  }
  if (c == nullptr || slot < 0 || slot >= EEA_COMM_SLOTS) return fail(EEA_ERR_INVALID_ARGUMENT, "bad communicator / slot");
  eea_status st = async_begin(c, nullptr, slot, false);
  if (st != EEA_OK) return st;
  if (c->comm == nullptr) {  // one rank, no collective: the sum's last wavefront publishes the flag itself
    st = eea_ck_records_sum_bound(e, B_local, d_ck_rec, d_rec_ready, seq, d_sum, d_flag, c->xstream);
    if (st != EEA_OK) return st;
    EEA_HIP(hipEventRecord(c->ev_done[slot], c->xstream));  // (for the groups that consume stream-ordered: eea_comm_wait)
    return EEA_OK;
  }
  const size_t bytes = eea_real_size(e) * eea_ck_record_len(e);
  if (c->xrec_cap[slot] < bytes) {
    // (a buffer that is too small is retired, not freed: hipFree waits for the whole device)
    if (c->d_xrec[slot]) c->retired.push_back(c->d_xrec[slot]);
    c->d_xrec[slot] = nullptr;
    c->xrec_cap[slot] = 0;
    EEA_HIP(hipMalloc(&c->d_xrec[slot], bytes));
    c->xrec_cap[slot] = bytes;
  }
  st = eea_ck_records_sum_bound(e, B_local, d_ck_rec, d_rec_ready, seq, c->d_xrec[slot], nullptr, c->xstream);
  if (st != EEA_OK) return st;
  st = eea_comm_allreduce_sum(e, c, c->d_xrec[slot], eea_ck_record_len(e), c->xstream);
  if (st != EEA_OK) return st;
  st = eea_publish_record(e, c->d_xrec[slot], d_sum, d_flag, seq, c->xstream);
  if (st != EEA_OK) return st;
  EEA_HIP(hipEventRecord(c->ev_done[slot], c->xstream));  // (for the groups that consume stream-ordered: eea_comm_wait)
  return EEA_OK;
}

eea_status eea_comm_wait(eea_comm* c, int slot, void* stream)
{
  if (c == nullptr || slot < 0 || slot >= EEA_COMM_SLOTS) return fail(EEA_ERR_INVALID_ARGUMENT, "bad communicator / slot");
  if (c->ev_done[slot] == nullptr) return EEA_OK;  // nothing was ever started in this slot
  if (hipEventQuery(c->ev_done[slot]) == hipSuccess) return EEA_OK;  // already finished: nothing to wait for
  EEA_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), c->ev_done[slot], 0));
  return EEA_OK;
}

}  // extern "C"
