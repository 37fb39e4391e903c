// Accuracy of the engine's fp64 sin(pi t), cos(pi t) (common.hpp sincospi_r) against long double
// references: arguments are reduced exactly (fmodl by 2) before sinl / cosl, so the reference is good to
// ~1e-19.  hipcc --offload-arch=gfx950 -O3 -I ergodic_exploration_amd/csrc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "common.hpp"

__global__ void eval(const double* t, double* s, double* c, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) eea::sincospi_r<double>(t[i], &s[i], &c[i]);
}

int main()
{
  const int n = 1 << 22;
  std::vector<double> t(n);
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> small(-2.0, 2.0), mid(-64.0, 64.0), big(-1e6, 1e6);
  for (int i = 0; i < n; ++i) t[i] = (i % 4 == 0) ? mid(rng) : ((i % 4 == 1) ? big(rng) : small(rng));
  // exact multiples of 1/4 and neighbours, huge arguments (slow path), tiny arguments
  const double specials[] = { 0.0, 0.25, 0.5, 0.75, 1.0, -0.25, -0.5, -1.0, 1e-300, -1e-20, 0.5 + 1e-16, 1e15 + 0.25,
                              3e15, 4.5e15, 1e18, -7e17 };
  for (size_t i = 0; i < sizeof(specials) / sizeof(double); ++i) t[i] = specials[i];
  double *dt, *ds, *dc;
  hipMalloc(&dt, n * 8);
  hipMalloc(&ds, n * 8);
  hipMalloc(&dc, n * 8);
  hipMemcpy(dt, t.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(eval, dim3((n + 255) / 256), dim3(256), 0, 0, dt, ds, dc, n);
  std::vector<double> s(n), c(n);
  hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
  long double worst = 0;
  double at = 0;
  const long double pi = 3.141592653589793238462643383279502884L;
  for (int i = 0; i < n; ++i) {
    const long double r = fmodl(static_cast<long double>(t[i]), 2.0L);
    const long double es = fabsl(sinl(pi * r) - s[i]), ec = fabsl(cosl(pi * r) - c[i]);
    const long double e = es > ec ? es : ec;
    if (e > worst) {
      worst = e;
      at = t[i];
    }
  }
  printf("sincospi_r<double>: %d arguments, max abs error %.3Le at t = %.17g\n", n, worst, at);
  return worst < 4e-16L ? 0 : 1;
}
