// Microbenchmark (gfx950): LDS throughput per CU for the access shapes of the control kernel -- ds_read_b64 /
// ds_read_b128 (distinct addresses per lane, conflict-free), wavefront-uniform (broadcast) b128 reads, ds_write_b64 /
// b128, ds_bpermute_b32.  16 wavefronts per CU (one 1024-thread workgroup), cycles per instruction and CU.
// hipcc --offload-arch=gfx950 -O3 -o lds lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void lds_kernel(double* out, long long* cyc, int iters)
{
  __shared__ __attribute__((aligned(16))) double sm[16 * 64 * 2 + 64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16 * 64 * 2; i += 1024) sm[i] = i * 0.5;
  __syncthreads();
  double* const base = sm + wv * 128;
  double acc0 = 0, acc1 = 0;
  int ia = 0;
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {  // b64, lane-distinct
        double v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(static_cast<int>((wv * 128 + lane) * 8)));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        acc0 += 0;
        (void)v;
      } else if (MODE == 1) {  // b128, lane-distinct
        d2v v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(static_cast<int>((wv * 128 + 2 * lane) * 8)));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        (void)v;
      } else if (MODE == 2) {  // b128, wavefront-uniform address
        d2v v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(static_cast<int>((wv * 128 + 2 * u) * 8)));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        (void)v;
      } else if (MODE == 3) {  // write b64
        asm volatile("ds_write_b64 %0, %1" ::"v"(static_cast<int>((wv * 128 + lane) * 8)), "v"(acc0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else if (MODE == 4) {  // write b128
        d2v v{ acc0, acc1 };
        asm volatile("ds_write_b128 %0, %1" ::"v"(static_cast<int>((wv * 128 + 2 * lane) * 8)), "v"(v) : "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else if (MODE == 5) {  // bpermute
        ia = __builtin_amdgcn_ds_bpermute(((lane * 7 + u) & 63) * 4, ia + lane);
      } else if (MODE == 7) {  // write b128, lanes 0..31 only
        d2v v{ acc0, acc1 };
        if (lane < 32) asm volatile("ds_write_b128 %0, %1" ::"v"(static_cast<int>((wv * 128 + 2 * lane) * 8)), "v"(v) : "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else if (MODE == 8) {  // read b64, different bank offsets per instruction (no conflict)
        double v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(static_cast<int>((wv * 128 + ((lane + 8 * u) & 63)) * 8)));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        (void)v;
      } else if (MODE == 6) {  // read2_b64 (two b64 per lane, 16 B apart... distinct)
        d2v v;
        asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64" : "=v"(v) : "v"(static_cast<int>((wv * 128 + lane) * 8)));
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        (void)v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 1024 + threadIdx.x] = acc0 + acc1 + ia + base[lane];
  if (lane == 0) cyc[blockIdx.x * 16 + wv] = t1 - t0;
}

int main()
{
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(double) * 1024 * cus);
  (void)hipMalloc(&cyc, sizeof(long long) * 16 * cus);
  const int iters = 500;
  std::vector<long long> h(16 * cus);
  const char* names[] = { "ds_read_b64   (lane-distinct, 512 B)", "ds_read_b128  (lane-distinct, 1024 B)", "ds_read_b128  (wavefront-uniform)   ",
                          "ds_write_b64  (512 B)               ", "ds_write_b128 (1024 B)              ", "ds_bpermute_b32                     ",
                          "ds_read2_b64  (2 x 512 B, same bank)", "ds_write_b128 (lanes 0..31, 512 B)  ", "ds_read_b64   (rotating addresses)  " };
#define RUN(MODE)                                                                                              \
  do {                                                                                                         \
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(lds_kernel<MODE>, dim3(cus), dim3(1024), 0, 0, out, cyc, iters); \
    (void)hipDeviceSynchronize();                                                                              \
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * 16 * cus, hipMemcpyDeviceToHost);                       \
    double mx = 0;                                                                                             \
    for (int b = 0; b < cus; ++b) { double m = 0; for (int w = 0; w < 16; ++w) m = h[b * 16 + w] > m ? h[b * 16 + w] : m; mx += m; } \
    printf("%s: %6.2f cycles per instruction and CU (16 wavefronts issuing)\n", names[MODE], mx / cus / (iters * 8.0 * 16)); \
  } while (0)
  RUN(0);
  RUN(1);
  RUN(2);
  RUN(3);
  RUN(4);
  RUN(5);
  RUN(6);
  RUN(7);
  RUN(8);
  return 0;
}
