// do the MFMA-f64 and VALU-f64 pipes overlap on MI355X?  (they do not: the mixed loop takes the sum of the two times)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NM, int NV>
__global__ __launch_bounds__(256) void k_mix(double* sink, int iters) {
  d4 acc[4]; double a[8];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int i = 0; i < 8; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double b = 1.0 + 1e-12 * threadIdx.x, c = 1e-13;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc[i % 4], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i % 8] = fma(a[i % 8], b, c);
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i]; for (int i = 0; i < 4; ++i) s += acc[i][0];
  if (s == 123.456) sink[0] = s;
}
template <int NM, int NV> void run(double* sink, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms; int iters = 4000;
  k_mix<NM, NV><<<blocks, 256>>>(sink, 100); hipDeviceSynchronize();
  hipEventRecord(e0); k_mix<NM, NV><<<blocks, 256>>>(sink, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  double fm = 2.0 * 16 * 16 * 4 * NM * (double)iters * 4.0 * blocks, fv = 2.0 * NV * (double)iters * 256.0 * blocks;
  printf("NM=%d NV=%d: %.3f ms  mfma %.1f TF/s + valu %.1f TF/s = %.1f TF/s\n", NM, NV, ms, fm / ms / 1e9, fv / ms / 1e9, (fm + fv) / ms / 1e9);
}
int main() {
  double* sink; hipMalloc(&sink, 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  for (int wps : {2, 4}) {
    int blocks = p.multiProcessorCount * wps;
    printf("--- %d waves/SIMD\n", wps);
    run<8, 0>(sink, blocks); run<0, 128>(sink, blocks); run<8, 128>(sink, blocks); run<8, 64>(sink, blocks); run<8, 32>(sink, blocks);
  }
}
