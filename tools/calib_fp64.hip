#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_fma(double* sink, int iters, unsigned long long* clk) {
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double b = 1.0 + 1e-12 * threadIdx.x, c = 1e-13;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fma(a[i], b, c);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
__global__ __launch_bounds__(256) void k_mfma(double* sink, int iters, unsigned long long* clk) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[2] = t1 - t0; clk[3] = r1 - r0; }
}
int main() {
  double* sink; unsigned long long* clk; hipMalloc(&sink, 8); hipMallocManaged(&clk, 64);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  for (int wpc : {4, 8, 16}) {
    int blocks = p.multiProcessorCount * wpc / 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    int iters = 40000;
    k_fma<<<blocks, 256>>>(sink, 1000, clk); hipDeviceSynchronize();
    hipEventRecord(e0); k_fma<<<blocks, 256>>>(sink, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    double fl = 2.0 * 16 * iters * 256.0 * blocks;
    printf("waves/CU=%d  v_fma_f64: %.2f TF/s  clock %.0f MHz  (%.2f cycles per wave-FMA per SIMD)\n", wpc, fl / ms / 1e9, 100.0 * clk[0] / clk[1],
           (double)clk[0] / (16.0 * iters) / (wpc / 4.0));
    iters = 10000;
    k_mfma<<<blocks, 256>>>(sink, 500, clk); hipDeviceSynchronize();
    hipEventRecord(e0); k_mfma<<<blocks, 256>>>(sink, iters, clk); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    fl = 2.0 * 16 * 16 * 4 * 8.0 * iters * 4.0 * blocks;
    printf("waves/CU=%d  mfma_f64 : %.2f TF/s  clock %.0f MHz  (%.2f cycles per MFMA per SIMD)\n", wpc, fl / ms / 1e9, 100.0 * clk[2] / clk[3],
           (double)clk[2] / (8.0 * iters) / (wpc / 4.0));
  }
  return 0;
}
