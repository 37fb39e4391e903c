// Microbenchmarks (gfx950): dependent-issue latency of fp64 VALU ops, DPP scan, sincospi,
// LDS round trip and workgroup barrier, in shader cycles per wave.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 512
template <int CH>
__global__ void fma_chain(double* out, long long* cyc, double a, double b)
{
  double x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + c;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) x[c] = __builtin_fma(x[c], a, b);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CH>
__global__ void sincospi_chain(double* out, long long* cyc)
{
  double x[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) x[c] = threadIdx.x * 1e-3 + 0.1 * c;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 32; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      double s, cc;
      sincospi(x[c], &s, &cc);
      x[c] = s * 0.5 + cc * 0.25;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void barrier_chain(double* out, long long* cyc)
{
  __shared__ double sh[8];
  double x = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 64; ++i) {
    if ((threadIdx.x & 63) == 63) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    x += sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void lds_chain(double* out, long long* cyc)
{
  __shared__ double sh[256];
  sh[threadIdx.x] = (double)((threadIdx.x * 7 + 1) & 255);
  __syncthreads();
  int idx = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) idx = (int)sh[idx];
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = idx;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double mean(const std::vector<long long>& v) { double s = 0; for (auto x : v) s += x; return s / v.size(); }

int main()
{
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  hipMalloc(&cyc, sizeof(long long) * 4096);
  std::vector<long long> h(4096);
  auto run = [&](const char* name, auto launch, int blocks, double per) {
    launch(); launch();
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    std::vector<long long> v(h.begin(), h.begin() + blocks);
    printf("%-44s blocks=%5d  %8.1f cycles  -> %.2f per unit\n", name, blocks, mean(v), mean(v) / per);
  };
  // 1 wave per block, 256 blocks: one wave per CU (alone on its SIMD)
  run("fma_f64 chain x1 (1 wave/SIMD)", [&] { fma_chain<1><<<256, 64>>>(out, cyc, 1.0000001, 1e-9); }, 256, N);
  run("fma_f64 chains x2", [&] { fma_chain<2><<<256, 64>>>(out, cyc, 1.0000001, 1e-9); }, 256, N);
  run("fma_f64 chains x4", [&] { fma_chain<4><<<256, 64>>>(out, cyc, 1.0000001, 1e-9); }, 256, N);
  run("fma_f64 chains x8", [&] { fma_chain<8><<<256, 64>>>(out, cyc, 1.0000001, 1e-9); }, 256, N);
  // 4 waves per block, 4 blocks per CU -> 4 waves per SIMD
  run("fma_f64 chain x1 (4 waves/SIMD)", [&] { fma_chain<1><<<1024, 256>>>(out, cyc, 1.0000001, 1e-9); }, 1024, N);
  run("fma_f64 chain x1 (8 waves/SIMD)", [&] { fma_chain<1><<<2048, 256>>>(out, cyc, 1.0000001, 1e-9); }, 2048, N);
  run("sincospi x1 (1 wave/SIMD)", [&] { sincospi_chain<1><<<256, 64>>>(out, cyc); }, 256, 32);
  run("sincospi x2 independent", [&] { sincospi_chain<2><<<256, 64>>>(out, cyc); }, 256, 32);
  run("sincospi x4 independent", [&] { sincospi_chain<4><<<256, 64>>>(out, cyc); }, 256, 32);
  run("barrier pair + 4 LDS reads (1 block/CU)", [&] { barrier_chain<<<256, 256>>>(out, cyc); }, 256, 64);
  run("barrier pair + 4 LDS reads (4 blocks/CU)", [&] { barrier_chain<<<1024, 256>>>(out, cyc); }, 1024, 64);
  run("dependent LDS read (ds_read_b64 + cvt)", [&] { lds_chain<<<256, 64>>>(out, cyc); }, 256, 256);
  return 0;
}
