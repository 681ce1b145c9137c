// calibrate.hip -- fp64 MFMA peak calibration (the local hardware guide lists no fp64 matrix rate;
// SURVEY.md section 7 asks for a micro-benchmark before quoting a roofline fraction).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace hommx {

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma_f64_peak(double* sink, int iters) {
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) sink[0] = s;  // keep the chain alive
}

hipError_t run_fp64_mfma_calibration(double* flops_per_s) {
  double* sink = nullptr;
  hipError_t e = hipMalloc(&sink, 8);
  if (e != hipSuccess) return e;
  hipDeviceProp_t prop;
  int dev = 0;
  hipGetDevice(&dev);
  hipGetDeviceProperties(&prop, dev);
  const int blocks = prop.multiProcessorCount * 2;  // 8 waves per CU = 2 per SIMD
  const int iters = 20000;
  hipEvent_t t0, t1;
  hipEventCreate(&t0);
  hipEventCreate(&t1);
  hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(256), 0, 0, sink, 200);  // warm-up
  hipEventRecord(t0, 0);
  hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(256), 0, 0, sink, iters);
  hipEventRecord(t1, 0);
  e = hipEventSynchronize(t1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, t0, t1);
  hipEventDestroy(t0);
  hipEventDestroy(t1);
  hipFree(sink);
  if (e != hipSuccess) return e;
  const double flops = 2.0 * 16 * 16 * 4 * 8.0 * iters * 4.0 * blocks;  // per MFMA x 8 x iters x waves
  *flops_per_s = flops / (ms * 1e-3);
  return hipGetLastError();
}

}  // namespace hommx
