// Microbenchmark (gfx950): v_mfma_f64_4x4x4_4b_f64 -- operand layout (one-hot probing) and throughput, alone and
// against a co-resident wavefront of v_fma_f64.  hipcc --offload-arch=gfx950 -O3 -o mfma4x4 mfma4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(int* out)
{
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la) {
    for (int lb = 0; lb < 64; ++lb) {
      const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m ? (__ffsll(static_cast<long long>(m)) - 1) + 100 * __popcll(m) : -1;
    }
  }
}

template <int NM, int NV>
__global__ void cross_wave(double* out, long long* cyc, int iters, double a, double b, unsigned mmask, unsigned vmask)
{
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
  const bool do_m = (mmask >> wv) & 1u, do_v = (vmask >> wv) & 1u;
  double acc[4] = { 0, 0, 0, 0 };
  double v[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) v[c] = a * (c + 1) + threadIdx.x;
  const double av = a + threadIdx.x, bv = b - threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (do_m) {
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[m & 3], 0, 0, 0);
    }
  }
  if (do_v) {
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int c = 0; c < NV; ++c) v[c & 15] = __builtin_fma(v[c & 15], b, a);
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += v[c];
#pragma unroll
  for (int c = 0; c < 4; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wv] = t1 - t0;
}

int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  int* d_probe;
  (void)hipMalloc(&d_probe, sizeof(int) * 4096);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_probe);
  std::vector<int> pr(4096);
  (void)hipMemcpy(pr.data(), d_probe, sizeof(int) * 4096, hipMemcpyDeviceToHost);
  // for every A lane: the B lanes it pairs with and the D lane the product lands in
  printf("# v_mfma_f64_4x4x4_4b_f64 one-hot probe: A lane -> list of (B lane : D lane)\n");
  for (int la = 0; la < 64; ++la) {
    printf("A%02d:", la);
    for (int lb = 0; lb < 64; ++lb) {
      if (pr[la * 64 + lb] >= 0) printf(" (B%02d:D%02d%s)", lb, pr[la * 64 + lb] % 100, pr[la * 64 + lb] / 100 > 1 ? "+" : "");
    }
    printf("\n");
  }
  double* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(double) * 512 * cus);
  (void)hipMalloc(&cyc, sizeof(long long) * 8 * cus);
  const int iters = 2000;
  std::vector<long long> h(8 * cus);
#define CROSS(NM, NV, MM, VM, WHAT)                                                                           \
  do {                                                                                                        \
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((cross_wave<NM, NV>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, 1.0, 0.5, MM, VM); \
    (void)hipDeviceSynchronize();                                                                             \
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * 8 * cus, hipMemcpyDeviceToHost);                       \
    double m = 0, v = 0;                                                                                      \
    for (int b = 0; b < cus; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += static_cast<double>(h[b * 8 + w]); \
    printf("two waves per SIMD, %s: waves 0-3 %7.1f cycles per iteration, waves 4-7 %7.1f\n", WHAT, m / (4.0 * cus) / iters, v / (4.0 * cus) / iters); \
  } while (0)
  CROSS(4, 32, 0x0fu, 0x00u, "0-3: 4 mfma_f64_4x4x4 (4 acc) | 4-7: idle      ");
  CROSS(8, 32, 0x0fu, 0x00u, "0-3: 8 mfma_f64_4x4x4 (4 acc) | 4-7: idle      ");
  CROSS(1, 32, 0x0fu, 0x00u, "0-3: 1 mfma_f64_4x4x4 (1 acc) | 4-7: idle      ");
  CROSS(2, 32, 0x0fu, 0x00u, "0-3: 2 mfma_f64_4x4x4 (2 acc) | 4-7: idle      ");
  CROSS(8, 32, 0x00u, 0xf0u, "0-3: idle                     | 4-7: 32 fma_f64");
  CROSS(8, 32, 0x0fu, 0xf0u, "0-3: 8 mfma_f64_4x4x4         | 4-7: 32 fma_f64");
  CROSS(8, 32, 0xffu, 0x00u, "0-3: 8 mfma_f64_4x4x4         | 4-7: 8 mfma    ");
  return 0;
}
