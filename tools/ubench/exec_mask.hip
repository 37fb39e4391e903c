// Microbenchmark (gfx950): does a VALU instruction cost less when only part of the wavefront is active?
// v_fmac_f64 / v_fmac_f32 / v_mfma-free, 16 independent accumulators, 1 and 4 wavefronts per SIMD, EXEC = the first n lanes
// (n = 64, 48, 32, 16, 8) or a scattered mask with the same count.  hipcc --offload-arch=gfx950 -O3 -o exec_mask exec_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename R>
__device__ __forceinline__ R fmac_plain(R acc, R x, R y);
template <>
__device__ __forceinline__ double fmac_plain<double>(double acc, double x, double y)
{
  asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(y));
  return acc;
}
template <>
__device__ __forceinline__ float fmac_plain<float>(float acc, float x, float y)
{
  asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(x), "v"(y));
  return acc;
}
template <typename R>
__global__ void rate(R* out, long long* cyc, int iters, R a, unsigned long long mask)
{
  R acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = a * c;
  R x = a + threadIdx.x, y = a - threadIdx.x;
  const int lane = threadIdx.x & 63;
  long long t0 = 0, t1 = 0;
  if ((mask >> lane) & 1ull) {  // EXEC = mask inside
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = fmac_plain<R>(acc[c], x, y);
    }
    t1 = __builtin_readcyclecounter();
  }
  R s = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  const int first = __ffsll(static_cast<long long>(mask)) - 1;
  if (lane == first) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <typename R>
void run(const char* name, int cus)
{
  R* out; long long* cyc;
  (void)hipMalloc(&out, sizeof(R) * 1024 * cus);
  (void)hipMalloc(&cyc, sizeof(long long) * 16 * cus);
  std::vector<long long> h(16 * cus);
  const int iters = 100000;
  const struct { const char* what; unsigned long long m; } masks[] = {
    { "64 lanes", ~0ull }, { "lanes 0..47", (1ull << 48) - 1 }, { "lanes 0..31", (1ull << 32) - 1 },
    { "lanes 0..15", 0xffffull }, { "lanes 0..7", 0xffull }, { "lanes 56..63", 0xffull << 56 },
    { "every 4th lane (16)", 0x1111111111111111ull }, { "lanes 0..7 + 32..39", 0xff000000ffull } };
  for (int NTHR = 256; NTHR <= 1024; NTHR *= 4) {
    for (const auto& mk : masks) {
      for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(rate<R>, dim3(cus), dim3(NTHR), 0, 0, out, cyc, iters, R(1), mk.m);
        (void)hipDeviceSynchronize();
      }
      const int nw = NTHR / 64 * cus;
      (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
      double s = 0;
      for (int q = 0; q < nw; ++q) s += static_cast<double>(h[q]);
      printf("%s  %-22s %d wavefront(s)/SIMD: %.2f cycles per instruction and wavefront = %.2f per instruction and SIMD\n", name,
             mk.what, NTHR / 256, s / nw / iters / 16, s / nw / iters / 16 / (NTHR / 256));
    }
  }
}
int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  run<double>("v_fmac_f64", p.multiProcessorCount);
  run<float>("v_fmac_f32", p.multiProcessorCount);
  return 0;
}
