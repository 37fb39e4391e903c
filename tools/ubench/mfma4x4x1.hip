// Microbenchmark (gfx950): v_mfma_f32_4x4x1_16b_f32 -- sixteen independent 4x4 outer products per instruction.
// Operand layout (two probe runs: A = lane + 1, B = 1 and A = 1, B = lane + 1) and throughput with 1 / 2 / 4
// independent accumulators, alone and against co-resident v_pk_fma_f32, and v_mfma_f32_16x16x4_f32 for comparison.
// hipcc --offload-arch=gfx950 -O3 -o mfma4x4x1 mfma4x4x1.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void probe(float* out)
{
  const int lane = threadIdx.x;
  const f4 z = f4{ 0, 0, 0, 0 };
  const f4 da = __builtin_amdgcn_mfma_f32_4x4x1f32(static_cast<float>(lane + 1), 1.0f, z, 0, 0, 0);
  const f4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, static_cast<float>(lane + 1), z, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    out[(0 * 64 + lane) * 4 + r] = da[r];
    out[(1 * 64 + lane) * 4 + r] = db[r];
  }
}

// NM matrix instructions on NACC accumulators + NV packed fp32 multiply-adds per iteration
template <int KIND, int NM, int NACC, int NV>
__global__ void rate(float* out, long long* cyc, int iters, float a, float b)
{
  f4 acc[NACC];
#pragma unroll
  for (int c = 0; c < NACC; ++c) acc[c] = f4{ 0, 0, 0, 0 };
  f2 v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = f2{ a * (c + 1) + threadIdx.x, b };
  const f2 b2 = f2{ b, b }, a2 = f2{ a, a };
  const float av = a + threadIdx.x, bv = b - threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (KIND == 0) acc[m % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, acc[m % NACC], 0, 0, 0);
      else acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m % NACC], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c & 7] = __builtin_elementwise_fma(v[c & 7], b2, a2);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += v[c].x + v[c].y;
#pragma unroll
  for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// fp64: NM matrix instructions (KIND 0: 4x4x4_4b, 1: 16x16x4) on NACC accumulators + NV v_fma_f64 per iteration
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KIND, int NM, int NACC, int NV>
__global__ void rate64(float* out, long long* cyc, int iters, double a, double b)
{
  double acc1[NACC];
  d4 acc4[NACC];
#pragma unroll
  for (int c = 0; c < NACC; ++c) {
    acc1[c] = 0.0;
    acc4[c] = d4{ 0, 0, 0, 0 };
  }
  double v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = a * (c + 1) + threadIdx.x;
  const double av = a + threadIdx.x, bv = b - threadIdx.x;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (KIND == 0) acc1[m % NACC] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc1[m % NACC], 0, 0, 0);
      else acc4[m % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc4[m % NACC], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c & 7] = __builtin_fma(v[c & 7], b, a);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_nop 0" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += v[c];
#pragma unroll
  for (int c = 0; c < NACC; ++c) s += acc1[c] + acc4[c][0] + acc4[c][1] + acc4[c][2] + acc4[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<float>(s);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  float* d_probe;
  (void)hipMalloc(&d_probe, sizeof(float) * 512);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_probe);
  std::vector<float> pr(512);
  (void)hipMemcpy(pr.data(), d_probe, sizeof(float) * 512, hipMemcpyDeviceToHost);
  printf("# v_mfma_f32_4x4x1_16b_f32 probe: D[lane][vgpr] = A(lane x) * B(lane y): the A lane and the B lane of every entry\n");
  for (int l = 0; l < 64; ++l) {
    printf("D lane %2d:", l);
    for (int r = 0; r < 4; ++r) printf("  v%d = A%02d x B%02d", r, static_cast<int>(pr[l * 4 + r]) - 1, static_cast<int>(pr[(64 + l) * 4 + r]) - 1);
    printf("\n");
  }
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * 2048 * cus);
  (void)hipMalloc(&cyc, sizeof(long long) * 32 * cus);
  const int iters = 2000;
  std::vector<long long> h(32 * cus);
#define RATE(KIND, NM, NACC, NV, WAVES, WHAT)                                                                   \
  do {                                                                                                          \
    for (int r = 0; r < 2; ++r)                                                                                 \
      hipLaunchKernelGGL((rate<KIND, NM, NACC, NV>), dim3(cus), dim3(64 * WAVES), 0, 0, out, cyc, iters, 1.0f, 0.5f); \
    (void)hipDeviceSynchronize();                                                                               \
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * WAVES * cus, hipMemcpyDeviceToHost);                     \
    double s = 0;                                                                                               \
    for (int i = 0; i < WAVES * cus; ++i) s += static_cast<double>(h[i]);                                       \
    printf("%-64s %2d wavefront(s) per SIMD: %7.1f cycles per iteration and wavefront\n", WHAT, WAVES / 4,      \
           s / (WAVES * cus) / iters);                                                                          \
  } while (0)
  RATE(0, 8, 1, 0, 4, "8 x 4x4x1_16b, 1 accumulator (dependent)");
  RATE(0, 8, 2, 0, 4, "8 x 4x4x1_16b, 2 accumulators");
  RATE(0, 8, 4, 0, 4, "8 x 4x4x1_16b, 4 accumulators");
  RATE(0, 8, 8, 0, 4, "8 x 4x4x1_16b, 8 accumulators");
  RATE(0, 8, 2, 0, 16, "8 x 4x4x1_16b, 2 accumulators");
  RATE(0, 8, 4, 0, 16, "8 x 4x4x1_16b, 4 accumulators");
  RATE(1, 4, 4, 0, 4, "4 x 16x16x4, 4 accumulators");
  RATE(1, 4, 4, 0, 16, "4 x 16x16x4, 4 accumulators");
  RATE(0, 0, 1, 16, 4, "16 v_pk_fma_f32");
  RATE(0, 8, 4, 16, 4, "8 x 4x4x1_16b (4 acc) + 16 v_pk_fma_f32, same wavefront");
  RATE(0, 8, 4, 16, 16, "8 x 4x4x1_16b (4 acc) + 16 v_pk_fma_f32, same wavefront");
  RATE(0, 0, 1, 16, 16, "16 v_pk_fma_f32");
  // wall-clock calibration (HIP events, 20000 iterations, the whole chip): TFLOP/s of the three instruction kinds
#define WALL(KIND, NM, NACC, NV, WAVES, FMA_PER_ITER, WHAT)                                                      \
  do {                                                                                                          \
    hipEvent_t e0, e1;                                                                                          \
    (void)hipEventCreate(&e0);                                                                                  \
    (void)hipEventCreate(&e1);                                                                                  \
    const int it = 20000;                                                                                       \
    hipLaunchKernelGGL((rate<KIND, NM, NACC, NV>), dim3(cus * (WAVES > 16 ? 2 : 1)), dim3(64 * (WAVES > 16 ? WAVES / 2 : WAVES)), 0, 0, out, cyc, it, 1.0f, 0.5f);  \
    (void)hipEventRecord(e0, 0);                                                                                \
    hipLaunchKernelGGL((rate<KIND, NM, NACC, NV>), dim3(cus * (WAVES > 16 ? 2 : 1)), dim3(64 * (WAVES > 16 ? WAVES / 2 : WAVES)), 0, 0, out, cyc, it, 1.0f, 0.5f);  \
    (void)hipEventRecord(e1, 0);                                                                                \
    (void)hipEventSynchronize(e1);                                                                              \
    float ms = 0;                                                                                               \
    (void)hipEventElapsedTime(&ms, e0, e1);                                                                     \
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * WAVES * cus, hipMemcpyDeviceToHost);                     \
    double s = 0;                                                                                               \
    for (int i = 0; i < WAVES * cus; ++i) s += static_cast<double>(h[i]);                                       \
    const double flop = 2.0 * (FMA_PER_ITER) * it * WAVES * cus;                                                  \
    printf("%-40s %2d wavefronts per SIMD: %8.3f ms -> %7.1f TFLOP/s; %9.0f s_memtime ticks per wavefront = %.3f GHz\n", WHAT, \
           WAVES / 4, ms, flop / (ms * 1e-3) / 1e12, s / (WAVES * cus), s / (WAVES * cus) / (ms * 1e-3) / 1e9);  \
  } while (0)
  WALL(0, 0, 1, 16, 16, 16.0 * 128, "16 v_pk_fma_f32 per iteration");
  WALL(0, 8, 4, 0, 16, 8.0 * 256, "8 x 4x4x1_16b per iteration");
  WALL(1, 4, 4, 0, 16, 4.0 * 1024, "4 x 16x16x4 per iteration");
  WALL(0, 0, 1, 16, 8, 16.0 * 128, "16 v_pk_fma_f32 per iteration");
  WALL(0, 0, 1, 16, 4, 16.0 * 128, "16 v_pk_fma_f32 per iteration");
#define WALL64(KIND, NM, NACC, NV, WAVES, FMA_PER_ITER, WHAT)                                                    \
  do {                                                                                                          \
    hipEvent_t e0, e1;                                                                                          \
    (void)hipEventCreate(&e0);                                                                                  \
    (void)hipEventCreate(&e1);                                                                                  \
    const int it = 20000;                                                                                       \
    hipLaunchKernelGGL((rate64<KIND, NM, NACC, NV>), dim3(cus * (WAVES > 16 ? 2 : 1)), dim3(64 * (WAVES > 16 ? WAVES / 2 : WAVES)), 0, 0, out, cyc, it, 1.0, 0.5); \
    (void)hipEventRecord(e0, 0);                                                                                \
    hipLaunchKernelGGL((rate64<KIND, NM, NACC, NV>), dim3(cus * (WAVES > 16 ? 2 : 1)), dim3(64 * (WAVES > 16 ? WAVES / 2 : WAVES)), 0, 0, out, cyc, it, 1.0, 0.5); \
    (void)hipEventRecord(e1, 0);                                                                                \
    (void)hipEventSynchronize(e1);                                                                              \
    float ms = 0;                                                                                               \
    (void)hipEventElapsedTime(&ms, e0, e1);                                                                     \
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * WAVES * cus, hipMemcpyDeviceToHost);                     \
    double s = 0;                                                                                               \
    for (int i = 0; i < WAVES * cus; ++i) s += static_cast<double>(h[i]);                                       \
    const double flop = 2.0 * (FMA_PER_ITER) * it * WAVES * cus;                                                  \
    printf("%-40s %2d wavefronts per SIMD: %8.3f ms -> %7.1f TFLOP/s; %9.0f s_memtime ticks per wavefront = %.3f GHz\n", WHAT, \
           WAVES / 4, ms, flop / (ms * 1e-3) / 1e12, s / (WAVES * cus), s / (WAVES * cus) / (ms * 1e-3) / 1e9);  \
  } while (0)
  WALL(0, 0, 1, 16, 32, 16.0 * 128, "16 v_pk_fma_f32 per iteration");
  WALL64(0, 0, 1, 16, 32, 16.0 * 64, "16 v_fma_f64 per iteration");
  WALL64(0, 0, 1, 16, 16, 16.0 * 64, "16 v_fma_f64 per iteration");
  WALL64(0, 0, 1, 16, 8, 16.0 * 64, "16 v_fma_f64 per iteration");
  WALL64(0, 0, 1, 16, 4, 16.0 * 64, "16 v_fma_f64 per iteration");
  WALL64(0, 8, 4, 0, 16, 8.0 * 256, "8 x f64 4x4x4_4b per iteration");
  WALL64(1, 4, 4, 0, 16, 4.0 * 1024, "4 x f64 16x16x4 per iteration");
  WALL64(0, 4, 4, 12, 16, 4.0 * 256 + 12.0 * 64, "4 x f64 4x4x4_4b + 12 v_fma_f64");
  return 0;
}
