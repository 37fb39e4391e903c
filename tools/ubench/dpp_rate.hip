// Microbenchmark (gfx950): issue rate of v_fmac_f64_dpp (row_newbcast) against plain v_fmac_f64, 16 independent
// accumulators, one wavefront per SIMD.  hipcc --offload-arch=gfx950 -O3 -o dpp_rate dpp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N>
__device__ __forceinline__ double fmac_rowbcast(double acc, double x, double y)
{
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(N));
  return acc;
}
__device__ __forceinline__ double fmac_plain(double acc, double x, double y)
{
  asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(y));
  return acc;
}
template <int DPP>
__global__ void rate(double* out, long long* cyc, int iters, double a)
{
  double acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = a * c;
  double x = a + threadIdx.x, y = a - threadIdx.x;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    if (DPP) {
      acc[0] = fmac_rowbcast<0>(acc[0], x, y); acc[1] = fmac_rowbcast<1>(acc[1], x, y);
      acc[2] = fmac_rowbcast<2>(acc[2], x, y); acc[3] = fmac_rowbcast<3>(acc[3], x, y);
      acc[4] = fmac_rowbcast<4>(acc[4], x, y); acc[5] = fmac_rowbcast<5>(acc[5], x, y);
      acc[6] = fmac_rowbcast<6>(acc[6], x, y); acc[7] = fmac_rowbcast<7>(acc[7], x, y);
      acc[8] = fmac_rowbcast<8>(acc[8], x, y); acc[9] = fmac_rowbcast<9>(acc[9], x, y);
      acc[10] = fmac_rowbcast<10>(acc[10], x, y); acc[11] = fmac_rowbcast<11>(acc[11], x, y);
      acc[12] = fmac_rowbcast<12>(acc[12], x, y); acc[13] = fmac_rowbcast<13>(acc[13], x, y);
      acc[14] = fmac_rowbcast<14>(acc[14], x, y); acc[15] = fmac_rowbcast<15>(acc[15], x, y);
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = fmac_plain(acc[c], x, y);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; long long* cyc;
  (void)hipMalloc(&out, sizeof(double) * 1024 * cus);
  (void)hipMalloc(&cyc, sizeof(long long) * 16 * cus);
  std::vector<long long> h(16 * cus);
  const int iters = 200000;
  for (int NTHR = 256; NTHR <= 1024; NTHR *= 2)
  for (int dpp = 0; dpp < 2; ++dpp) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int r = 0; r < 2; ++r) {
      (void)hipEventRecord(e0, 0);
      if (dpp) hipLaunchKernelGGL(rate<1>, dim3(cus), dim3(NTHR), 0, 0, out, cyc, iters, 1.0);
      else hipLaunchKernelGGL(rate<0>, dim3(cus), dim3(NTHR), 0, 0, out, cyc, iters, 1.0);
      (void)hipEventRecord(e1, 0);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("  wall %.3f ms -> %.1f TFLOP/s fp64 (2 flop per lane and instruction)\n", ms,
           2.0 * 64 * 16.0 * iters * (NTHR / 64) * cus / (ms * 1e-3) / 1e12);
    const int nw = NTHR / 64 * cus;
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double s = 0;
    for (int q = 0; q < nw; ++q) s += static_cast<double>(h[q]);
    printf("%s: %.2f cycles per instruction and wavefront, %d wavefront(s) per SIMD -> %.2f cycles per instruction and SIMD\n",
           dpp ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64                 ", s / nw / iters / 16, NTHR / 256, s / nw / iters / 16 / (NTHR / 256));
  }
  return 0;
}
