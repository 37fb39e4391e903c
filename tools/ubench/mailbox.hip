// Round trip of an EMPTY request between a host thread and a workgroup that stays on the device, by where the two mailbox
// lines live -- the floor of the resident single-robot path (control_kernel_impl.hpp control_resident_kernel).
//   request line : host memory (mapped, the GPU polls it over PCIe)  |  device memory (the host writes it through the BAR)
//   answer line  : host memory (the GPU writes it over PCIe)
// and by how the answer is published: payload, wait for the stores, then the number (two dependent trips) | payload and number
// in ONE 64-byte store burst of 16 lanes (the host checks the number at both ends of the line).
// Every experiment runs in its own child process: a placement the platform does not map for the host ends the child with a
// signal, not the probe.   usage: mailbox [round trips = 20000]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                              \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      std::printf("   %s: %s\n", #x, hipGetErrorString(e_));                              \
      std::fflush(stdout);                                                                 \
      std::exit(3);                                                                        \
    }                                                                                      \
  } while (0)

struct alignas(64) Line
{
  unsigned w[16];  // w[15] = number (request), w[0] and w[15] = number (one-burst answer)
};

// work = dependent fp64 operations between request and answer (stands for the body)
__global__ __launch_bounds__(256) void serve(const Line* req, Line* ans, int one_burst, int work, unsigned* total_ticks)
{
  __shared__ unsigned s_line[16];
  unsigned last = 0;
  for (;;) {
    if (threadIdx.x < 64) {
      const int l = threadIdx.x & 15;
      unsigned v, r;
      do {
        v = __hip_atomic_load(&req->w[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        r = __builtin_amdgcn_readlane(v, 15);
      } while (r == last);
      if (threadIdx.x < 16) s_line[threadIdx.x] = v;
    }
    __syncthreads();
    const long long t0 = wall_clock64();
    const unsigned r = s_line[15];
    last = r;
    double acc = static_cast<double>(s_line[1]);
    for (int i = 0; i < work; ++i) acc = acc * 1.0000001 + 1e-9;
    if (r == 0xffffffffu) return;
    if (one_burst) {
      if (threadIdx.x < 16) {
        unsigned v = (threadIdx.x == 0 || threadIdx.x == 15) ? r : (threadIdx.x == 1 ? static_cast<unsigned>(acc) : s_line[threadIdx.x]);
        __hip_atomic_store(&ans->w[threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
      if (threadIdx.x > 0 && threadIdx.x < 15) {
        __hip_atomic_store(&ans->w[threadIdx.x], threadIdx.x == 1 ? static_cast<unsigned>(acc) : s_line[threadIdx.x], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(&ans->w[15], r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) atomicAdd(total_ticks, static_cast<unsigned>(wall_clock64() - t0));
    __syncthreads();
  }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int experiment(int req_place, int one_burst, int work, int trips)
{
  Line *h_req = nullptr, *d_req = nullptr, *h_ans = nullptr, *d_ans = nullptr;
  unsigned* d_ticks = nullptr;
  CK(hipMalloc(&d_ticks, 4));
  CK(hipMemset(d_ticks, 0, 4));
  CK(hipHostMalloc(&h_ans, sizeof(Line), hipHostMallocMapped));
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_ans), h_ans, 0));
  std::memset(h_ans, 0, sizeof(Line));
  if (req_place == 0) {
    CK(hipHostMalloc(&h_req, sizeof(Line), hipHostMallocMapped));
    CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_req), h_req, 0));
    std::memset(h_req, 0, sizeof(Line));
  } else {
    if (req_place == 1) CK(hipMalloc(&d_req, sizeof(Line)));
    if (req_place == 2) CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&d_req), sizeof(Line), hipDeviceMallocFinegrained));
    if (req_place == 3) CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&d_req), sizeof(Line), hipDeviceMallocUncached));
    CK(hipMemset(d_req, 0, sizeof(Line)));
    CK(hipDeviceSynchronize());
    h_req = d_req;  // the host writes device memory through the BAR (a signal ends this child where that is not mapped)
  }
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  serve<<<1, 256, 0, s>>>(d_req, d_ans, one_burst, work, d_ticks);
  CK(hipGetLastError());
  volatile unsigned* ans = h_ans->w;
  double t0 = 0.0;
  const int warm = 200;
  for (int i = 1; i <= trips + warm; ++i) {
    if (i == warm + 1) t0 = now();
    const unsigned seq = static_cast<unsigned>(i);
    for (int k = 0; k < 15; ++k) h_req->w[k] = seq + k;
    __atomic_store_n(&h_req->w[15], seq, __ATOMIC_RELEASE);
    long spin = 0;
    if (one_burst) {
      while (!(ans[15] == seq && ans[0] == seq)) {
        if (++spin > 2000000000L) return 4;
      }
    } else {
      while (__atomic_load_n(&h_ans->w[15], __ATOMIC_ACQUIRE) != seq) {
        if (++spin > 2000000000L) return 4;
      }
    }
    if (ans[2] != seq + 2) {
      std::printf("   wrong payload at trip %d: %u\n", i, ans[2]);
      return 5;
    }
  }
  const double us = 1e6 * (now() - t0) / trips;
  for (int k = 0; k < 15; ++k) h_req->w[k] = 0;
  __atomic_store_n(&h_req->w[15], 0xffffffffu, __ATOMIC_RELEASE);
  CK(hipStreamSynchronize(s));
  unsigned ticks = 0;
  CK(hipMemcpy(&ticks, d_ticks, 4, hipMemcpyDeviceToHost));
  std::printf("   %.2f us per round trip; on the device between request seen and answer issued: %.2f us\n", us,
              0.01 * ticks / (trips + warm));
  return 0;
}

int main(int argc, char** argv)
{
  const int trips = argc > 1 ? std::atoi(argv[1]) : 20000;
  const char* places[] = { "host memory (mapped)", "device memory, hipMalloc", "device memory, fine-grained", "device memory, uncached" };
  for (int work = 0; work <= 400; work += 400) {
    for (int place = 0; place < 4; ++place) {
      for (int burst = 0; burst < 2; ++burst) {
        std::printf("request line in %s; answer %s; body of %d dependent fp64 operations\n", places[place],
                    burst ? "in one burst" : "payload, wait, number", work);
        std::fflush(stdout);
        const pid_t pid = fork();  // (the parent never touches the GPU)
        if (pid == 0) {
          const int rc = experiment(place, burst, work, trips);
          std::fflush(stdout);
          std::_Exit(rc);
        }
        int st = 0;
        waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) std::printf("   ended by signal %d\n", WTERMSIG(st));
        else if (WEXITSTATUS(st) != 0) std::printf("   failed (%d)\n", WEXITSTATUS(st));
        std::fflush(stdout);
      }
    }
  }
  return 0;
}
