// TEST DOUBLE of the six RCCL entry points csrc/comm.hip binds at run time (ncclGetUniqueId, ncclCommInitRank,
// ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString), built as tests/fake_rccl/librccl.so.1.
//
// Purpose: the multi-rank branches of the engine's exchange code have to run with more than one rank on boxes with ONE GPU
// (RCCL refuses two ranks of one communicator on one device).  Here the ranks are PROCESSES (as in production: one process
// per rank) that share the one GPU -- the group's device buffers travel as hipIpcMemHandles through a POSIX shared-memory
// segment named by the unique id -- or threads of one process (then the pointers are used as they are; but a HIP process
// maps its streams onto 4 hardware queues, and two ranks whose streams share a queue can never meet on the device: the
// first kernel spins in front of the second.  tests/fake_rccl/selftest.cpp shows it; processes have their own queues).
// Round 5: the collectives are STREAM-ASYNCHRONOUS and KERNEL-SHAPED, like the real ones --
//   * a call enqueues ONE KERNEL on the caller's stream and returns; no host wait, no host copy, no host rendezvous (round
//     4's double began every collective with hipStreamSynchronize: a host wait, exactly what the device-bound exchange
//     exists to avoid, and nothing that had to become resident beside the waiting control kernels);
//   * the kernels of the ranks meet ON THE DEVICE: each copies its send buffer into the group's staging slot of this
//     operation, raises its arrival word, spins (bounded: ~3 s of the constant 100 MHz clock, then an error word the tests
//     read through fake_rccl_errors) on the other ranks' words, and then reduces / gathers from the staging slots in RANK
//     ORDER into its own receive buffer (in-place capable: the send buffer is not read after the arrival);
//   * with a realistic footprint: 512 threads per block, 96 vector registers, 16 KB of LDS, one block (<= 32 KB per rank)
//     or four -- it has to BECOME RESIDENT beside whatever fills the GPU, which is the property under test.
// Operations of one communicator are numbered per rank in call order (every rank issues the same sequence, on streams that
// serialise them, as RCCL requires): operation n uses staging slot n % 4 -- a rank can be at most one operation ahead of
// the slowest (it cannot pass an arrival), so a slot is never overwritten while it is read.
// Semantics kept: stream order, rank order of the gathered blocks, sum over the ranks in rank order.  Test infrastructure
// only: nothing in the product links or loads it unless a test points eea_comm_set_library at it or puts this directory in
// front of LD_LIBRARY_PATH of a process that has no other RCCL mapped.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>

namespace
{
constexpr unsigned kRing = 4, kMaxBlocks = 4, kMaxRanks = 8;
constexpr size_t kMaxBytes = size_t(4) << 20;  // per rank and operation
constexpr long long kTimeoutTicks = 300000000LL;  // wall_clock64: 100 MHz

// the rendezvous segment of a communicator, /dev/shm/<unique id>: zero-filled when created
struct Shm
{
  std::atomic<int> claimed;  // the first rank to arrive allocates
  std::atomic<int> ready;    // buffers allocated, handles / pointers published
  std::atomic<int> joined;   // ranks that have opened them
  int creator_pid;
  void* p_arrived;           // (for the ranks that live in the creator's process)
  void* p_stage;
  void* p_err;
  hipIpcMemHandle_t h_arrived, h_stage, h_err;
};
struct FakeGroup
{
  int nranks = 0;
  Shm* shm = nullptr;
  unsigned* d_arrived = nullptr;     // [kRing][nranks][kMaxBlocks]
  unsigned char* d_stage = nullptr;  // [kRing][nranks][kMaxBytes]
  int* d_err = nullptr;              // number of blocks that gave up
};
}  // namespace

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5,
               ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm { FakeGroup* group; int rank; unsigned next_op; };
typedef ncclComm* ncclComm_t;
}

namespace
{
std::mutex g_mutex;
std::map<std::string, FakeGroup*> g_groups;  // the groups this process has joined
int g_next_id = 1;

template <typename T>
__device__ __forceinline__ T ld_agent(const T* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void st_agent(T* q, T v) { __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// BIG: the footprint of a real collective kernel (512 threads, 96 registers, 16 KB of LDS: it needs two free wavefront slots on
// every SIMD of ONE compute unit at the same moment).  FAKE_RCCL_SMALL=1 launches the same protocol as one wavefront with a
// minimal footprint -- the A/B that tells resource starvation from protocol errors.
template <typename T, bool BIG>
__global__ __launch_bounds__(BIG ? 512 : 64) void fake_collective_kernel(unsigned* arrived, unsigned char* stage, int* err, int nranks,
                                                                         int rank, unsigned op, const T* send, T* recv, size_t count,
                                                                         int gather)
{
  constexpr unsigned NT = BIG ? 512u : 64u;
  __shared__ double s_pad[BIG ? 2048 : 64];  // 16 KB: the LDS footprint of a collective kernel's staging FIFOs
  __shared__ int s_ok;
  if constexpr (BIG) asm volatile("" ::: "v95");  // ... and its registers: the allocation is 96 vector registers per lane
  const unsigned tid = threadIdx.x, blk = blockIdx.x, nb = gridDim.x;
  const unsigned slot = op % kRing, tag = op + 1u;
  for (unsigned i = tid; i < (BIG ? 2048u : 64u); i += NT) s_pad[i] = static_cast<double>(i);  // (keeps the array)
  const size_t per = (count + nb - 1) / nb, lo = blk * per, hi = (lo + per < count) ? lo + per : count;
  T* const mine = reinterpret_cast<T*>(stage + (static_cast<size_t>(slot) * nranks + rank) * kMaxBytes);
  for (size_t i = lo + tid; i < hi; i += NT) st_agent(mine + i, send[i]);
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(arrived + (slot * nranks + rank) * kMaxBlocks + blk, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 1;
    const long long t0 = wall_clock64();
    for (int r = 0; r < nranks && ok; ++r) {
      const unsigned* const w = arrived + (slot * nranks + r) * kMaxBlocks + blk;
      while (static_cast<int>(__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - tag) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > kTimeoutTicks) {
          ok = 0;
          break;
        }
      }
    }
    if (!ok) atomicAdd(err, 1);
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) return;  // (the receive buffer stays as it is; the test reads the error word)
  if (gather) {
    for (int r = 0; r < nranks; ++r) {
      const T* const src = reinterpret_cast<const T*>(stage + (static_cast<size_t>(slot) * nranks + r) * kMaxBytes);
      for (size_t i = lo + tid; i < hi; i += NT) recv[static_cast<size_t>(r) * count + i] = ld_agent(src + i);
    }
  } else {
    for (size_t i = lo + tid; i < hi; i += NT) {
      T acc = ld_agent(reinterpret_cast<const T*>(stage + static_cast<size_t>(slot) * nranks * kMaxBytes) + i);
      for (int r = 1; r < nranks; ++r) {
        acc += ld_agent(reinterpret_cast<const T*>(stage + (static_cast<size_t>(slot) * nranks + r) * kMaxBytes) + i);
      }
      recv[i] = acc;
    }
  }
  if (s_pad[(tid * 7) & (BIG ? 2047 : 63)] < 0.0) recv[0] = T(0);  // (never true: the array holds its indices)
}

ncclResult_t enqueue(ncclComm_t comm, const void* send, void* recv, size_t count, ncclDataType_t type, int gather, hipStream_t stream)
{
  const size_t ts = type == ncclFloat64 ? 8 : (type == ncclFloat32 ? 4 : 0);
  if (comm == nullptr || ts == 0 || count == 0 || count * ts > kMaxBytes) return ncclInvalidArgument;
  FakeGroup* const g = comm->group;
  const unsigned op = comm->next_op++;
  const unsigned nb = count * ts > (size_t(32) << 10) ? kMaxBlocks : 1u;
  static const bool small = [] {
    const char* const v = std::getenv("FAKE_RCCL_SMALL");
    return v != nullptr && v[0] != '0';
  }();
#define FAKE_LAUNCH(T, BIG)                                                                                                        \
  hipLaunchKernelGGL((fake_collective_kernel<T, BIG>), dim3(nb), dim3(BIG ? 512 : 64), 0, stream, g->d_arrived, g->d_stage, g->d_err, \
                     g->nranks, comm->rank, op, static_cast<const T*>(send), static_cast<T*>(recv), count, gather)
  if (ts == 8) {
    if (small) FAKE_LAUNCH(double, false);
    else FAKE_LAUNCH(double, true);
  } else {
    if (small) FAKE_LAUNCH(float, false);
    else FAKE_LAUNCH(float, true);
  }
#undef FAKE_LAUNCH
  return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
  std::lock_guard<std::mutex> lock(g_mutex);
  std::memset(id, 0, sizeof(*id));
  const long long t = std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof(id->internal), "/fake-rccl-%d-%d-%llx", static_cast<int>(getpid()), g_next_id++, t);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
  if (comm == nullptr || nranks < 1 || nranks > static_cast<int>(kMaxRanks) || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const std::string key(id.internal, strnlen(id.internal, sizeof(id.internal)));
  if (key.empty() || key[0] != '/') return ncclInvalidArgument;
  FakeGroup* g = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_groups.find(key);
    if (it != g_groups.end()) {  // another rank of this process has joined already: same buffers
      g = it->second;
      if (g->nranks != nranks) return ncclInvalidArgument;
    } else {
      const int fd = shm_open(key.c_str(), O_CREAT | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, sizeof(Shm)) != 0) return ncclSystemError;
      void* const m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (m == MAP_FAILED) return ncclSystemError;
      g = new FakeGroup();
      g->nranks = nranks;
      g->shm = static_cast<Shm*>(m);
      const size_t words = static_cast<size_t>(kRing) * nranks * kMaxBlocks;
      if (g->shm->claimed.fetch_add(1) == 0) {
        // (the calling thread has the device of the group current: every rank of the double shares it)
        bool ok = hipMalloc(&g->d_arrived, words * sizeof(unsigned)) == hipSuccess &&
                  hipMemset(g->d_arrived, 0, words * sizeof(unsigned)) == hipSuccess &&
                  hipMalloc(&g->d_stage, static_cast<size_t>(kRing) * nranks * kMaxBytes) == hipSuccess &&
                  hipMalloc(&g->d_err, sizeof(int)) == hipSuccess && hipMemset(g->d_err, 0, sizeof(int)) == hipSuccess &&
                  hipDeviceSynchronize() == hipSuccess;
        ok = ok && hipIpcGetMemHandle(&g->shm->h_arrived, g->d_arrived) == hipSuccess &&
             hipIpcGetMemHandle(&g->shm->h_stage, g->d_stage) == hipSuccess && hipIpcGetMemHandle(&g->shm->h_err, g->d_err) == hipSuccess;
        if (!ok) return ncclUnhandledCudaError;
        g->shm->creator_pid = static_cast<int>(getpid());
        g->shm->p_arrived = g->d_arrived;
        g->shm->p_stage = g->d_stage;
        g->shm->p_err = g->d_err;
        g->shm->ready.store(1);
      } else {
        while (g->shm->ready.load() == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
        if (g->shm->creator_pid == static_cast<int>(getpid())) {  // (cannot happen under g_mutex; kept for clarity)
          g->d_arrived = static_cast<unsigned*>(g->shm->p_arrived);
          g->d_stage = static_cast<unsigned char*>(g->shm->p_stage);
          g->d_err = static_cast<int*>(g->shm->p_err);
        } else {
          void *a = nullptr, *st = nullptr, *e = nullptr;
          if (hipIpcOpenMemHandle(&a, g->shm->h_arrived, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
              hipIpcOpenMemHandle(&st, g->shm->h_stage, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
              hipIpcOpenMemHandle(&e, g->shm->h_err, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            return ncclUnhandledCudaError;
          }
          g->d_arrived = static_cast<unsigned*>(a);
          g->d_stage = static_cast<unsigned char*>(st);
          g->d_err = static_cast<int*>(e);
        }
      }
      g_groups[key] = g;
    }
  }
  // like the real call: returns once every rank of the communicator has joined
  g->shm->joined.fetch_add(1);
  while (g->shm->joined.load() < nranks) std::this_thread::sleep_for(std::chrono::microseconds(200));
  if (rank == 0) shm_unlink(key.c_str());  // (everybody has it mapped; the name can go)
  *comm = new ncclComm{ g, rank, 0u };
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  delete comm;  // (the group itself lives to the end of the process: a few MB per test)
  return ncclSuccess;
}

// the ranks that have JOINED the group (what the real call reports once ncclCommInitRank has returned: all of them)
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count)
{
  if (comm == nullptr || count == nullptr) return ncclInvalidArgument;
  *count = comm->group->shm->joined.load();
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL: invalid argument / HIP failure"; }

// recv [nranks][count] in rank order (in-place capable: send may be recv + rank * count)
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
  return enqueue(comm, send, recv, count, type, 1, stream);
}

// recv = sum over the ranks, added in rank order (in-place capable)
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream)
{
  if (op != ncclSum) return ncclInvalidArgument;
  return enqueue(comm, send, recv, count, type, 0, stream);
}

// test hook (not an RCCL entry point): blocks of collective kernels that gave up waiting for another rank, over all groups
// this process has joined (the count is the group's: every rank's process reads the same word)
int fake_rccl_errors(void)
{
  std::lock_guard<std::mutex> lock(g_mutex);
  int n = 0;
  for (auto& kv : g_groups) {
    int v = 0;
    if (hipMemcpy(&v, kv.second->d_err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    n += v;
  }
  return n;
}

}  // extern "C"
