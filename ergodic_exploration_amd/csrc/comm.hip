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
#include <memory>
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
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;  // optional (introspection: eea_comm_library_nranks)
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
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(a.handle, "ncclCommCount"));
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
int eea_comm_library_nranks(const eea_comm* c)
{
  if (c == nullptr || c->comm == nullptr) return 0;  // a local communicator: no collective library behind it
  if (rccl().CommCount == nullptr) return -1;
  int n = -1;
  return rccl().CommCount(c->comm, &n) == ncclSuccess ? n : -1;
}

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

// ---- ABI 6: a stream-level wait for a device flag (the GATED exchange) --------------------------------------------------------
// One wavefront that returns once *flag - seq >= 0 (mod 2^32), polling past the L1 with a sleep between polls, bounded like
// wait_flag (about a second: then it counts a time-out and returns -- the launches behind it go on with whatever the record
// holds).  Enqueued IN FRONT of a group's control launch it does what the in-kernel flag wait of the device-bound exchange
// does, but holds ONE execution slot while it waits instead of the group's thousands: the group's half of the chip stays
// empty until the record is there, so the collective kernel that produces it always finds room (the dead-lock of waiting
// control wavefronts that fill every slot, profiles/r05_two_ranks.txt, cannot form), and nothing of the protocol needs an
// event: per pass the host issues launches only (an event record + wait pair costs 4.6 us of host time on this runtime, a
// small launch 0.7 - 2.6: tools/ubench/host_calls.hip).
namespace
{
__global__ __launch_bounds__(64) void flag_gate_kernel(const unsigned* flag, unsigned seq, unsigned* timeouts)
{
  for (int i = 0; i < eea::kFlagPolls; ++i) {
    const unsigned v = __builtin_amdgcn_readfirstlane(eea::load_agent(flag));
    if (static_cast<int>(v - seq) >= 0) return;
    __builtin_amdgcn_s_sleep(32);
  }
  if (timeouts != nullptr && threadIdx.x == 0) atomicAdd(timeouts, 1u);
}
}  // namespace

extern "C" eea_status eea_stream_wait_flag(const unsigned* d_flag, unsigned seq, unsigned* d_timeouts, void* stream)
{
  if (d_flag == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null flag");
  hipLaunchKernelGGL(flag_gate_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), d_flag, seq, d_timeouts);
  EEA_HIP(hipGetLastError());
  return EEA_OK;
}

// ---- ABI 6: the consensus loop of a rank as one replayable device graph (include/ergodic_amd.h) ----------------------------
// `passes_per_launch` passes of the stream-ordered protocol -- per pass: the groups' control launches (records out, the sum
// record of pass i - lag in), then on the exchange branch the record sum (caller-owned workspace: eea_ck_records_sum_ws) and the
// all-reduce over the ranks -- captured ONCE from an exchange stream of the plan's own (the origin) + the group streams, and replayed with
// one hipGraphLaunch.  Dependencies inside a launch are the captured events; across launches the device serialises the
// launches of one executable graph, which is why the first `lag` passes of the capture carry no wait for "their" exchange (it
// ran in the launch before).  Slot rotation is periodic in the launch (passes_per_launch % slots == 0).
struct eea_consensus_plan
{
  eea_engine* e = nullptr;
  eea_comm* c = nullptr;
  int device = 0;
  unsigned n_groups = 0, lag = 0, slots = 0, passes = 0, B = 0, rec_len = 0;
  size_t rs = 8;
  std::vector<unsigned> agents, first, rec_first;  // per group: agents, first agent, first record of a pass
  unsigned n_rec = 0;                               // records per pass (<= B)
  std::vector<eea_batch_io> io;
  std::vector<hipStream_t> gstreams;
  hipStream_t xs = nullptr;
  std::vector<hipEvent_t> ev_group;  // [n_groups]: a group's launch of the pass being captured
  std::vector<hipEvent_t> ev_x;      // [slots]: the exchange of the pass that wrote the slot
  hipEvent_t ev_fork = nullptr;
  void* d_arec = nullptr;     // [slots][B][rec_len]
  void* d_sum = nullptr;      // [slots][rec_len]
  void* d_ws = nullptr;       // [slots] record-sum workspaces
  void* d_tickets = nullptr;  // [slots]
  size_t ws_bytes = 0, ticket_bytes = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

namespace
{
void plan_release(eea_consensus_plan* p)
{
  if (p == nullptr) return;
  (void)hipSetDevice(p->device);
  if (p->exec) (void)hipGraphExecDestroy(p->exec);
  if (p->graph) (void)hipGraphDestroy(p->graph);
  for (hipStream_t s : p->gstreams) {
    if (s) {
      (void)hipStreamSynchronize(s);
      (void)hipStreamDestroy(s);
    }
  }
  if (p->xs) {
    (void)hipStreamSynchronize(p->xs);
    (void)hipStreamDestroy(p->xs);
  }
  for (hipEvent_t ev : p->ev_group) {
    if (ev) (void)hipEventDestroy(ev);
  }
  for (hipEvent_t ev : p->ev_x) {
    if (ev) (void)hipEventDestroy(ev);
  }
  if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
  for (void* q : { p->d_arec, p->d_sum, p->d_ws, p->d_tickets }) {
    if (q) (void)hipFree(q);
  }
  delete p;
}

// the passes of one launch, enqueued on the plan's streams inside a capture that began on the EXCHANGE stream.
// The exchange stream is the capture's origin on purpose: this HIP runtime (ROCm 7) re-parents every NON-origin stream that
// waits for a captured event to the event's stream and lists it there as a "parallel capture stream", every time -- two
// non-origin streams that wait for each other's events (a group stream for the exchange, the exchange stream for the group)
// become each other's children and hipStreamEndCapture recurses for ever (found the hard way: a 174 000-frame stack).  With
// the exchange stream as the origin every wait is between the origin and a group stream, never between two group streams.
eea_status plan_enqueue(eea_consensus_plan* p)
{
  char* const arec = static_cast<char*>(p->d_arec);
  char* const sum = static_cast<char*>(p->d_sum);
  const size_t rec_bytes = p->rs * p->rec_len;
  // fork: the group streams join the capture behind the origin
  EEA_HIP(hipEventRecord(p->ev_fork, p->xs));
  for (unsigned g = 0; g < p->n_groups; ++g) EEA_HIP(hipStreamWaitEvent(p->gstreams[g], p->ev_fork, 0));
  for (unsigned i = 0; i < p->passes; ++i) {
    const unsigned slot = i % p->slots, src = (i + p->slots - p->lag) % p->slots;
    for (unsigned g = 0; g < p->n_groups; ++g) {
      // the sum record this pass consumes is complete: the exchange of pass i - lag (of the launch before for i < lag --
      // ordered by the device's serialisation of the launches, no captured dependency)
      if (i >= p->lag) EEA_HIP(hipStreamWaitEvent(p->gstreams[g], p->ev_x[src], 0));
      eea_batch_io io = p->io[g];
      // (one record per wavefront where agents share one, eea_batch_io::rec_per_wavefront: rec_first / n_rec count records)
      io.d_ck_rec = arec + (static_cast<size_t>(slot) * p->B + p->rec_first[g]) * rec_bytes;
      io.rec_per_wavefront = 1;
      io.d_ck_shared = sum + static_cast<size_t>(src) * rec_bytes;
      io.ck_shared_parts = 1;
      const eea_status st = eea_control_batch(p->e, p->agents[g], &io, p->gstreams[g]);
      if (st != EEA_OK) return st;
      EEA_HIP(hipEventRecord(p->ev_group[g], p->gstreams[g]));
      EEA_HIP(hipStreamWaitEvent(p->xs, p->ev_group[g], 0));
    }
    void* const s_slot = sum + static_cast<size_t>(slot) * rec_bytes;
    eea_status st = eea_ck_records_sum_ws(p->e, p->n_rec, arec + static_cast<size_t>(slot) * p->B * rec_bytes, s_slot,
                                          static_cast<char*>(p->d_ws) + static_cast<size_t>(slot) * p->ws_bytes,
                                          static_cast<char*>(p->d_tickets) + static_cast<size_t>(slot) * p->ticket_bytes, p->xs);
    if (st != EEA_OK) return st;
    st = eea_comm_allreduce_sum(p->e, p->c, s_slot, p->rec_len, p->xs);  // (nothing without an RCCL communicator)
    if (st != EEA_OK) return st;
    EEA_HIP(hipEventRecord(p->ev_x[slot], p->xs));
  }
  // join: the origin ends behind every group stream (its own last node is the last exchange, which follows the last launches)
  for (unsigned g = 0; g < p->n_groups; ++g) {
    EEA_HIP(hipEventRecord(p->ev_group[g], p->gstreams[g]));
    EEA_HIP(hipStreamWaitEvent(p->xs, p->ev_group[g], 0));
  }
  return EEA_OK;
}
}  // namespace

extern "C" {

eea_status eea_consensus_plan_create(eea_engine* e, eea_comm* c, const eea_consensus_desc* d, eea_consensus_plan** out)
{
  if (out == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  *out = nullptr;
  if (e == nullptr || c == nullptr || d == nullptr || d->group_agents == nullptr || d->group_io == nullptr) {
    return fail(EEA_ERR_INVALID_ARGUMENT, "null argument");
  }
  if (d->n_groups < 1 || d->n_groups > eea_comm::kMaxGroups) return fail(EEA_ERR_INVALID_ARGUMENT, "1 .. 8 agent groups");
  if (d->lag < 1 || d->lag + 2 > EEA_COMM_SLOTS) return fail(EEA_ERR_INVALID_ARGUMENT, "lag must be in 1 .. EEA_COMM_SLOTS - 2");
  if (d->passes_per_launch < 1 || d->passes_per_launch > 4096) return fail(EEA_ERR_INVALID_ARGUMENT, "passes_per_launch in 1 .. 4096");
  std::unique_ptr<eea_consensus_plan, void (*)(eea_consensus_plan*)> p(new eea_consensus_plan(), plan_release);
  p->e = e;
  p->c = c;
  p->device = c->device;
  p->n_groups = d->n_groups;
  p->lag = d->lag;
  p->slots = d->lag + 2;
  p->passes = (d->passes_per_launch + p->slots - 1) / p->slots * p->slots;
  p->rec_len = eea_ck_record_len(e);
  p->rs = eea_real_size(e);
  for (unsigned g = 0; g < d->n_groups; ++g) {
    const eea_batch_io& io = d->group_io[g];
    if (d->group_agents[g] == 0 || io.d_pose == nullptr || io.d_ut == nullptr || io.d_u0 == nullptr) {
      return fail(EEA_ERR_INVALID_ARGUMENT, "every group needs agents, d_pose, d_ut and d_u0");
    }
    p->first.push_back(p->B);
    p->agents.push_back(d->group_agents[g]);
    p->B += d->group_agents[g];
    p->rec_first.push_back(p->n_rec);
    p->n_rec += eea_batch_record_count(e, d->group_agents[g]);  // (= the group's agents for one agent per wavefront)
    eea_batch_io mine = io;  // the exchange fields are the plan's
    mine.d_ck_shared = nullptr;
    mine.d_ck_rec = nullptr;
    mine.ck_shared_parts = 0;
    mine.d_rec_ready = nullptr;
    mine.rec_seq = 0;
    mine.d_ck_flag = nullptr;
    mine.ck_flag_seq = 0;
    p->io.push_back(mine);
  }
  EEA_HIP(hipSetDevice(p->device));
  // phi_k of a rebuild that is still enqueued is complete before anything is captured (a capture must not wait for an
  // event of the world outside it)
  {
    std::vector<double> phik(eea_num_modes(e));
    const eea_status st = eea_get_phik(e, phik.data());
    if (st != EEA_OK) return st;
  }
  eea_status st = eea_ck_records_sum_ws_bytes(e, p->B, &p->ws_bytes, &p->ticket_bytes);
  if (st != EEA_OK) return st;
  p->ws_bytes = (p->ws_bytes + 255) / 256 * 256;
  p->ticket_bytes = (p->ticket_bytes + 255) / 256 * 256;
  const size_t rec_bytes = p->rs * p->rec_len;
  EEA_HIP(hipMalloc(&p->d_arec, static_cast<size_t>(p->slots) * p->B * rec_bytes));
  EEA_HIP(hipMalloc(&p->d_sum, static_cast<size_t>(p->slots) * rec_bytes));
  EEA_HIP(hipMalloc(&p->d_ws, static_cast<size_t>(p->slots) * p->ws_bytes));
  EEA_HIP(hipMalloc(&p->d_tickets, static_cast<size_t>(p->slots) * p->ticket_bytes));
  // empty sum records (agent count 0: the first `lag` passes keep their own c_k), zeroed tickets
  EEA_HIP(hipMemset(p->d_arec, 0, static_cast<size_t>(p->slots) * p->B * rec_bytes));
  EEA_HIP(hipMemset(p->d_sum, 0, static_cast<size_t>(p->slots) * rec_bytes));
  EEA_HIP(hipMemset(p->d_tickets, 0, static_cast<size_t>(p->slots) * p->ticket_bytes));
  int least = 0, greatest = 0;
  EEA_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  p->gstreams.assign(p->n_groups, nullptr);
  for (unsigned g = 0; g < p->n_groups; ++g) EEA_HIP(hipStreamCreateWithFlags(&p->gstreams[g], hipStreamNonBlocking));
  EEA_HIP(hipStreamCreateWithPriority(&p->xs, hipStreamNonBlocking, greatest));
  p->ev_group.assign(p->n_groups, nullptr);
  for (unsigned g = 0; g < p->n_groups; ++g) EEA_HIP(hipEventCreateWithFlags(&p->ev_group[g], kDeviceEvent));
  p->ev_x.assign(p->slots, nullptr);
  for (unsigned s = 0; s < p->slots; ++s) EEA_HIP(hipEventCreateWithFlags(&p->ev_x[s], kDeviceEvent));
  EEA_HIP(hipEventCreateWithFlags(&p->ev_fork, kDeviceEvent));
  if (c->comm != nullptr) {
    // the collective library's first call on a communicator sets up its own state (buffers, proxy): outside the capture,
    // on the empty sum record of slot 0 (zeros stay zeros; every rank makes this call in plan_create)
    st = eea_comm_allreduce_sum(e, c, p->d_sum, p->rec_len, p->xs);
    if (st != EEA_OK) return st;
    EEA_HIP(hipStreamSynchronize(p->xs));
  }
  EEA_HIP(hipDeviceSynchronize());
  // capture (relaxed mode: other threads of the process may be making runtime calls of their own)
  hipError_t he = hipStreamBeginCapture(p->xs, hipStreamCaptureModeRelaxed);
  if (he != hipSuccess) return fail(EEA_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(he));
  st = plan_enqueue(p.get());
  const std::string why = st != EEA_OK ? std::string(eea_last_error()) : std::string();
  he = hipStreamEndCapture(p->xs, &p->graph);
  if (st != EEA_OK || he != hipSuccess || p->graph == nullptr) {
    (void)hipGetLastError();
    return fail(st == EEA_ERR_INVALID_ARGUMENT ? st : EEA_ERR_UNSUPPORTED,
                "the consensus passes could not be captured into a graph (" +
                    (st != EEA_OK ? why : std::string(hipGetErrorString(he))) + "): use the per-call exchange");
  }
  he = hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0);
  if (he != hipSuccess) {
    (void)hipGetLastError();
    return fail(EEA_ERR_UNSUPPORTED, std::string("hipGraphInstantiate: ") + hipGetErrorString(he));
  }
  *out = p.release();
  return EEA_OK;
}

eea_status eea_consensus_plan_launch(eea_consensus_plan* p, void* stream)
{
  if (p == nullptr || p->exec == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null plan");
  EEA_HIP(hipSetDevice(p->device));
  EEA_HIP(hipGraphLaunch(p->exec, static_cast<hipStream_t>(stream)));
  return EEA_OK;
}

eea_status eea_consensus_plan_info(const eea_consensus_plan* p, unsigned* passes_per_launch, const void** d_last_sum)
{
  if (p == nullptr) return fail(EEA_ERR_INVALID_ARGUMENT, "null plan");
  if (passes_per_launch) *passes_per_launch = p->passes;
  if (d_last_sum) {
    *d_last_sum = static_cast<const char*>(p->d_sum) + static_cast<size_t>((p->passes - 1) % p->slots) * p->rs * p->rec_len;
  }
  return EEA_OK;
}

void eea_consensus_plan_destroy(eea_consensus_plan* p) { plan_release(p); }

}  // extern "C"
