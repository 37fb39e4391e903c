// Microbenchmark (gfx950): what is the shader clock under load, and what does s_memtime count?
// One wavefront per workgroup executes a known number of wait states (s_nop 15 = 16 wait states of 4 cycles each) between two readings of
// s_memtime and of s_memrealtime (the constant 100 MHz counter); the other wavefronts of the workgroup load the SIMDs with
// fp64 multiply-adds, fp64 matrix instructions or nothing.  clock = wait-state cycles / real time; s_memtime rate = its
// ticks / real time.  hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int LOAD>  // 0: idle, 1: v_fma_f64, 2: v_mfma_f64_16x16x4
__global__ void probe(unsigned long long* out, int nop_iters, int load_iters, double a, double b)
{
  const int wave = threadIdx.x >> 6;
  if (wave == 0) {
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 1
    for (int i = 0; i < nop_iters; ++i) {
      // 64 x s_nop 15 = 1024 wait states per iteration (+ the loop's three scalar instructions)
      asm volatile(
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if ((threadIdx.x & 63) == 0) {
      out[2 * blockIdx.x] = t1 - t0;
      out[2 * blockIdx.x + 1] = r1 - r0;
    }
    return;
  }
  double v[8];
  d4 acc[2] = { d4{ 0, 0, 0, 0 }, d4{ 0, 0, 0, 0 } };
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = a * (c + 1) + threadIdx.x;
  if (LOAD != 0) {
#pragma unroll 1
    for (int i = 0; i < load_iters; ++i) {
      if (LOAD == 1) {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c & 7] = __builtin_fma(v[c & 7], b, a);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[0], v[1], acc[c & 1], 0, 0, 0);
      }
    }
  }
  double s = acc[0][0] + acc[1][0];
#pragma unroll
  for (int c = 0; c < 8; ++c) s += v[c];
  if (s == 12345.678) out[0] = 1;  // (keeps the load alive)
}

int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned long long* d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * 2 * cus);
  std::vector<unsigned long long> h(2 * cus);
  const int nop_iters = 4000;  // 4.1 M wait states ~ 1.7 ms at 2.4 GHz
  printf("# %s: one wavefront per workgroup runs %d x 1024 wait states (s_nop 15); 15 more wavefronts per workgroup carry the load\n", p.name, nop_iters);
#define RUN(LOAD, ITERS, WHAT)                                                                                       \
  do {                                                                                                               \
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<LOAD>), dim3(cus), dim3(1024), 0, 0, d, nop_iters, ITERS, 1.0, 0.5); \
    (void)hipDeviceSynchronize();                                                                                    \
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * cus, hipMemcpyDeviceToHost);                       \
    double ticks = 0, real = 0;                                                                                      \
    for (int b = 0; b < cus; ++b) {                                                                                  \
      ticks += static_cast<double>(h[2 * b]);                                                                        \
      real += static_cast<double>(h[2 * b + 1]);                                                                     \
    }                                                                                                                \
    ticks /= cus;                                                                                                    \
    real /= cus;                                                                                                     \
    const double ns = real * 10.0, cycles = 4.0 * 1027.0 * nop_iters;  /* a wait state is 4 cycles */                                                      \
    printf("%-34s real time %8.1f us: shader clock %.3f GHz (wait states / real time), s_memtime %.3f G ticks/s = %.3f of the clock\n", \
           WHAT, ns * 1e-3, cycles / ns, ticks / ns, ticks / cycles);                                                \
  } while (0)
  RUN(0, 0, "idle SIMDs");
  RUN(1, 30000, "v_fma_f64 on 15 wavefronts per CU");
  RUN(2, 12000, "v_mfma_f64_16x16x4 on 15 wavefronts");
  RUN(0, 0, "idle SIMDs (again)");
  return 0;
}
