// Host-side cost of the runtime calls a stream-ordered exchange is made of (round 6): microseconds per call,
// enqueued back to back onto busy streams (a 20 us kernel keeps each stream occupied so that the calls only ENQUEUE).
//   hipcc --offload-arch=gfx950 -O2 -o host_calls host_calls.hip && ./host_calls
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

#define OK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 1;                                                                \
    }                                                                          \
  } while (0)

__global__ void empty_kernel() {}
__global__ void spin_kernel(long long ticks)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void gate_kernel(const unsigned* flag, unsigned seq)
{
  for (int i = 0; i < 4000000; ++i) {
    if (static_cast<int>(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) >= 0) return;
    __builtin_amdgcn_s_sleep(16);
  }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  hipStream_t a, b;
  OK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  OK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t ev[64];
  for (auto& e : ev) OK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  unsigned* flag = nullptr;
  OK(hipMalloc(&flag, 64));
  OK(hipMemset(flag, 0, 64));  // every wait below (for sequence number 0) is satisfied at once on the device
  const int N = 2000;
  auto run = [&](const char* name, auto&& body) {
    hipDeviceSynchronize();
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, a, 3000000LL);  // 30 ms: the streams stay busy while we enqueue
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, b, 3000000LL);
    const double t0 = now();
    for (int i = 0; i < N; ++i) body(i);
    const double dt = now() - t0;
    hipDeviceSynchronize();
    std::printf("%-58s %7.2f us per call\n", name, 1e6 * dt / N);
    std::fflush(stdout);
    return 0;
  };
  run("empty kernel launch (hipLaunchKernelGGL)", [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, a); });
  run("gate kernel launch (2 args)", [&](int) { hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, a, flag, 0u); });
  run("hipEventRecord (DisableTiming|DisableSystemFence)", [&](int i) { (void)hipEventRecord(ev[i & 63], a); });
  run("hipEventRecord on a + hipStreamWaitEvent on b (pair)", [&](int i) {
    (void)hipEventRecord(ev[i & 63], a);
    (void)hipStreamWaitEvent(b, ev[i & 63], 0);
  });
  run("hipStreamWaitValue32 (GEQ)", [&](int) { (void)hipStreamWaitValue32(a, flag, 0u, hipStreamWaitValueGte, 0xffffffffu); });
  run("hipStreamWriteValue32", [&](int i) { (void)hipStreamWriteValue32(a, flag + 1, static_cast<unsigned>(i), 0); });
  // and the same after an alternating pattern launch / wait (what a pass looks like)
  run("launch on a + launch on b + record a + wait b (one 'pass')", [&](int i) {
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, a);
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, b);
    (void)hipEventRecord(ev[i & 63], a);
    (void)hipStreamWaitEvent(b, ev[i & 63], 0);
  });
  run("launch a + launch b + WaitValue32 a + WaitValue32 b", [&](int) {
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, a);
    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, b);
    (void)hipStreamWaitValue32(a, flag, 0u, hipStreamWaitValueGte, 0xffffffffu);
    (void)hipStreamWaitValue32(b, flag, 0u, hipStreamWaitValueGte, 0xffffffffu);
  });
  return 0;
}
