// Microbenchmark (gfx950): do fp64 vector instructions overlap with v_mfma_f64_16x16x4_f64 on one SIMD?
//   same-wave: a loop body of 1 matrix instruction + NV independent v_fma_f64 (one wavefront per SIMD);
//   cross-wave: two wavefronts per SIMD, one issuing only matrix instructions, the other only v_fma_f64.
// If the two pipes overlap, the time is max(matrix, vector); if the matrix instruction holds the vector pipe,
// it is the sum.  Cycles from s_memtime (shader clock), lane 0 of every wavefront.
// hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ void same_wave(double* out, long long* cyc, int iters, double a, double b)
{
  d4 acc[2] = { d4{ 0, 0, 0, 0 }, d4{ 0, 0, 0, 0 } };
  double v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = a * (c + 1) + threadIdx.x;
  const double av = a + threadIdx.x, bv = b - threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[m & 1], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c & 7] = __builtin_fma(v[c & 7], b, a);
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += v[c];
  s += acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[1][0] + acc[1][1] + acc[1][2] + acc[1][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// blockDim = 512 (two wavefronts per SIMD): wavefront w issues matrix instructions if bit w of mmask is set, vector
// instructions if bit w of vmask is set, nothing otherwise (wavefront-uniform roles: scalar branches only)
template <int NM, int NV>
__global__ void cross_wave(double* out, long long* cyc, int iters, double a, double b, unsigned mmask, unsigned vmask)
{
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / 64);
  const bool do_m = (mmask >> wv) & 1u, do_v = (vmask >> wv) & 1u;
  d4 acc[4] = { d4{ 0, 0, 0, 0 }, d4{ 0, 0, 0, 0 }, d4{ 0, 0, 0, 0 }, d4{ 0, 0, 0, 0 } };
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
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[m & 3], 0, 0, 0);
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
  for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wv] = t1 - t0;
}

static double mean(const std::vector<long long>& v, int first, int step, int n)
{
  double s = 0;
  int c = 0;
  for (int i = first; i < n; i += step) {
    s += static_cast<double>(v[i]);
    ++c;
  }
  return s / c;
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out;
  long long* cyc;
  hipMalloc(&out, sizeof(double) * 512 * cus);
  hipMalloc(&cyc, sizeof(long long) * 8 * cus);
  const int iters = 2000;
  std::vector<long long> h(8 * cus);
  printf("# %s, %d CUs; cycles per loop iteration (s_memtime), one workgroup per CU\n", p.name, cus);
#define SAME(NM, NV)                                                                                          \
  do {                                                                                                        \
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((same_wave<NM, NV>), dim3(cus), dim3(256), 0, 0, out, cyc, iters, 1.0, 0.5); \
    hipDeviceSynchronize();                                                                                   \
    hipMemcpy(h.data(), cyc, sizeof(long long) * 4 * cus, hipMemcpyDeviceToHost);                             \
    printf("same wave: %d mfma_f64 + %2d fma_f64 per iteration: %7.1f cycles\n", NM, NV, mean(h, 0, 1, 4 * cus) / iters); \
  } while (0)
  SAME(1, 0);
  SAME(0, 8);
  SAME(0, 16);
  SAME(1, 4);
  SAME(1, 8);
  SAME(1, 12);
  SAME(1, 16);
  SAME(1, 24);
  SAME(2, 16);
  SAME(2, 24);
#define CROSS(NM, NV, MM, VM, WHAT)                                                                           \
  do {                                                                                                        \
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((cross_wave<NM, NV>), dim3(cus), dim3(512), 0, 0, out, cyc, iters, 1.0, 0.5, MM, VM); \
    hipDeviceSynchronize();                                                                                   \
    hipMemcpy(h.data(), cyc, sizeof(long long) * 8 * cus, hipMemcpyDeviceToHost);                             \
    double m = 0, v = 0;                                                                                      \
    for (int b = 0; b < cus; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += static_cast<double>(h[b * 8 + w]); \
    printf("two waves per SIMD, %s: waves 0-3 %7.1f cycles per iteration, waves 4-7 %7.1f\n", WHAT, m / (4.0 * cus) / iters, v / (4.0 * cus) / iters); \
  } while (0)
  CROSS(4, 32, 0x0fu, 0x00u, "0-3: 4 mfma_f64        | 4-7: idle      ");
  CROSS(4, 32, 0x00u, 0xf0u, "0-3: idle              | 4-7: 32 fma_f64");
  CROSS(4, 32, 0x0fu, 0xf0u, "0-3: 4 mfma_f64        | 4-7: 32 fma_f64");
  CROSS(4, 32, 0xf0u, 0x0fu, "0-3: 32 fma_f64        | 4-7: 4 mfma_f64");
  CROSS(4, 64, 0x00u, 0xf0u, "0-3: idle              | 4-7: 64 fma_f64");
  CROSS(4, 64, 0x0fu, 0xf0u, "0-3: 4 mfma_f64        | 4-7: 64 fma_f64");
  CROSS(4, 32, 0xffu, 0x00u, "0-3: 4 mfma_f64        | 4-7: 4 mfma_f64");
  CROSS(4, 32, 0x00u, 0xffu, "0-3: 32 fma_f64        | 4-7: 32 fma_f64");
  return 0;
}
