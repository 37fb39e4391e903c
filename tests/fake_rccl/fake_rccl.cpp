// TEST DOUBLE of the six RCCL entry points csrc/comm.hip binds at run time (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString), built as tests/fake_rccl/librccl.so.1.
//
// Purpose: the multi-rank branches of the engine's exchange code -- eea_comm_create(nranks > 1), the
// rank order of the all-gather, the collective branch of both exchange forms -- have never run with more than one rank,
// because the build and test boxes have one GPU and RCCL refuses two ranks of one communicator on one device.  Here the
// "ranks" are THREADS of one process that share the one GPU: a collective is a host rendezvous (mutex + condition
// variable) around device copies.  Semantics kept: stream order (the call waits for the caller's stream, the result is
// complete when the call returns), rank order of the gathered blocks, sum over the ranks in rank order (the reference
// the tests compare against adds in the same order), in-place operation.  Not kept: asynchrony, performance, every other
// RCCL entry point.  Test infrastructure only: nothing in the product links or loads it; the C++ host test that uses it
// puts this directory in front of LD_LIBRARY_PATH of a process that has no PyTorch (and hence no real RCCL) mapped.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

struct FakeGroup
{
  int nranks = 0, joined = 0, left = 0;
  std::mutex m;
  std::condition_variable cv;
  // one collective at a time: per-rank pointers of the current call, arrival / departure counters, generation
  std::vector<const void*> send;
  std::vector<void*> recv;
  int arrived = 0, gen = 0;
  std::vector<unsigned char> host;  // staging for the reduction
};

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm { FakeGroup* group; int rank; };
typedef ncclComm* ncclComm_t;
}

namespace
{
std::mutex g_mutex;
std::map<std::string, FakeGroup*> g_groups;
int g_next_id = 1;

// all ranks have called; the LAST arrival runs `work` (all pointers are published), then everyone leaves together
template <typename F>
void rendezvous(FakeGroup* g, int rank, const void* send, void* recv, F work)
{
  std::unique_lock<std::mutex> lock(g->m);
  const int my_gen = g->gen;
  g->send[rank] = send;
  g->recv[rank] = recv;
  if (++g->arrived == g->nranks) {
    work();
    g->arrived = 0;
    ++g->gen;
    g->cv.notify_all();
  } else {
    g->cv.wait(lock, [&] { return g->gen != my_gen; });
  }
}
size_t type_size(ncclDataType_t t) { return t == ncclFloat64 ? 8 : (t == ncclFloat32 ? 4 : 0); }
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  std::lock_guard<std::mutex> lock(g_mutex);
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "fake-rccl-%d", g_next_id++);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
  if (comm == nullptr || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  FakeGroup* g = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    const std::string key(id.internal, strnlen(id.internal, sizeof(id.internal)));
    auto it = g_groups.find(key);
    if (it == g_groups.end()) {
      g = new FakeGroup();
      g->nranks = nranks;
      g->send.assign(nranks, nullptr);
      g->recv.assign(nranks, nullptr);
      g_groups[key] = g;
    } else {
      g = it->second;
      if (g->nranks != nranks) return ncclInvalidArgument;
    }
  }
  {  // like the real call: returns once every rank of the communicator has joined
    std::unique_lock<std::mutex> lock(g->m);
    ++g->joined;
    g->cv.notify_all();
    g->cv.wait(lock, [&] { return g->joined >= g->nranks; });
  }
  *comm = new ncclComm{ g, rank };
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  delete comm;  // (the group itself lives to the end of the process: a handful of bytes per test)
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL: invalid argument / HIP failure"; }

// recv [nranks][count] in rank order (in-place capable: send may be recv + rank * count)
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
  const size_t bytes = count * type_size(type);
  if (comm == nullptr || bytes == 0) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  FakeGroup* g = comm->group;
  bool ok = true;
  rendezvous(g, comm->rank, send, recv, [&] {
    g->host.resize(bytes * g->nranks);
    for (int r = 0; r < g->nranks; ++r) ok = ok && hipMemcpy(g->host.data() + bytes * r, g->send[r], bytes, hipMemcpyDeviceToHost) == hipSuccess;
    for (int r = 0; r < g->nranks; ++r) ok = ok && hipMemcpy(g->recv[r], g->host.data(), bytes * g->nranks, hipMemcpyHostToDevice) == hipSuccess;
  });
  return ok ? ncclSuccess : ncclUnhandledCudaError;
}

// recv = sum over the ranks, added in rank order (in-place capable)
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
  const size_t bytes = count * type_size(type);
  if (comm == nullptr || bytes == 0 || op != ncclSum) return ncclInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  FakeGroup* g = comm->group;
  bool ok = true;
  rendezvous(g, comm->rank, send, recv, [&] {
    g->host.resize(bytes * (g->nranks + 1));
    unsigned char* const acc = g->host.data() + bytes * g->nranks;
    for (int r = 0; r < g->nranks; ++r) ok = ok && hipMemcpy(g->host.data() + bytes * r, g->send[r], bytes, hipMemcpyDeviceToHost) == hipSuccess;
    std::memcpy(acc, g->host.data(), bytes);
    for (int r = 1; r < g->nranks; ++r) {
      if (type == ncclFloat64) {
        for (size_t i = 0; i < count; ++i) reinterpret_cast<double*>(acc)[i] += reinterpret_cast<const double*>(g->host.data() + bytes * r)[i];
      } else {
        for (size_t i = 0; i < count; ++i) reinterpret_cast<float*>(acc)[i] += reinterpret_cast<const float*>(g->host.data() + bytes * r)[i];
      }
    }
    for (int r = 0; r < g->nranks; ++r) ok = ok && hipMemcpy(g->recv[r], acc, bytes, hipMemcpyHostToDevice) == hipSuccess;
  });
  return ok ? ncclSuccess : ncclUnhandledCudaError;
}

}  // extern "C"
