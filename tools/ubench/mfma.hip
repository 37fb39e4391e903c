// Microbenchmark (gfx950): throughput of v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 with CH
// independent accumulators per wave, W waves per SIMD; wall time -> instructions/s per SIMD and the
// implied cycles per instruction at the reported clock.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void mfma64(double* out, int iters, double a, double b)
{
  d4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = d4{ 0, 0, 0, 0 };
  const double av = a + threadIdx.x, bv = b - threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ void mfma32(float* out, int iters, float a, float b)
{
  f4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = f4{ 0, 0, 0, 0 };
  const float av = a + threadIdx.x, bv = b - threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[c], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch)
{
  launch();
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < 5; ++r) launch();
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / 5;
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  int clk_khz = 0;
  hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("# %s, %d CUs, reported clock %.0f MHz\n", p.name, cus, clk_khz / 1e3);
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * cus * 8);
  const int iters = 4096;
  for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD: blocks of 256 threads = 1 wave per SIMD
    const int blocks = cus * wps;
    {
      const double ms = time_ms([&] { hipLaunchKernelGGL(mfma64<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0); });
      const double per_simd = static_cast<double>(iters) * 8 * wps;  // instructions per SIMD
      printf("mfma_f64_16x16x4 x8 acc, %d wave/SIMD: %.3f ms -> %.1f ns per instr per SIMD (%.1f cycles at %.0f MHz), %.1f TFLOP/s\n",
             wps, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * clk_khz / 1e6, clk_khz / 1e3,
             per_simd * 4 * cus * 2048 / (ms * 1e-3) / 1e12);
    }
    {
      const double ms = time_ms([&] { hipLaunchKernelGGL(mfma32<8>, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<float*>(out), iters, 1.0f, 2.0f); });
      const double per_simd = static_cast<double>(iters) * 8 * wps;
      printf("mfma_f32_16x16x4 x8 acc, %d wave/SIMD: %.3f ms -> %.1f ns per instr per SIMD (%.1f cycles at %.0f MHz), %.1f TFLOP/s\n",
             wps, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * clk_khz / 1e6, clk_khz / 1e3,
             per_simd * 4 * cus * 2048 / (ms * 1e-3) / 1e12);
    }
  }
  {
    const double ms = time_ms([&] { hipLaunchKernelGGL(mfma64<2>, dim3(cus), dim3(256), 0, 0, out, iters, 1.0, 2.0); });
    printf("mfma_f64_16x16x4 x2 acc (dependent pairs), 1 wave/SIMD: %.1f ns per instr\n", ms * 1e6 / (iters * 2.0));
  }
  hipFree(out);
  return 0;
}
