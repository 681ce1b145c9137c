// Accuracy of v_rcp_f64 on gfx950 and of one / two Newton steps on top of it (max relative error in ulps of 2^-53 over random
// arguments of both signs and many magnitudes).  hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_rcp64 tools/probe_rcp64.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  r0[i] = r;
  double e = fma(-d, r, 1.0);
  r = fma(e, r, r);
  r1[i] = r;
  e = fma(-d, r, 1.0);
  r = fma(e, r, r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> h(n), a(n), b(n), c(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> m(1.0, 2.0);
  std::uniform_int_distribution<int> ex(-60, 60);
  for (int i = 0; i < n; ++i) h[i] = std::ldexp(m(g), ex(g)) * ((i & 1) ? -1.0 : 1.0);
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, 8ull * n); hipMalloc(&d0, 8ull * n); hipMalloc(&d1, 8ull * n); hipMalloc(&d2, 8ull * n);
  hipMemcpy(dx, h.data(), 8ull * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, 8ull * n, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, 8ull * n, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, 8ull * n, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)h[i];
    e0 = std::fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
    e1 = std::fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
    e2 = std::fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
  }
  const double u = std::ldexp(1.0, -53);
  printf("max rel err: v_rcp_f64 %.3e (%.1f ulp, 2^%.1f)   +1 Newton %.3e (%.2f ulp)   +2 Newton %.3e (%.2f ulp)\n", e0, e0 / u,
         std::log2(e0), e1, e1 / u, e2, e2 / u);
  return 0;
}
