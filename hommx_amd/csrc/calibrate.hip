// calibrate.hip -- measured fp64 peak of the device (the local hardware guide lists no fp64 rate; SURVEY.md section 7 asks
// for a micro-benchmark before a roofline fraction is quoted).  Three dependent-chain-free loops (the third: the MFMA fed from LDS like a GEMM), v_mfma_f64_16x16x4_f64 with
// 8 independent accumulator tiles per wave and v_fma_f64 with 16 independent accumulators per lane, each at 2 and 4 waves per
// SIMD, in short bursts and in one long run; the best rate of each instruction is reported.  On MI355X both instructions share ONE fp64 FMA pipe (DESIGN.md
// section 5), so the larger of the two figures is the measured peak a kernel mixing them can be priced against.
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

__global__ __launch_bounds__(256) void k_fma_f64_peak(double* sink, int iters) {
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  const double b = 1.0 + 1e-12 * threadIdx.x, c = 1e-13;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fma(a[i], b, c);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 123.456) sink[0] = s;
}

// The same instruction fed the way a GEMM feeds it: the operands of every MFMA come out of LDS (2 x 2 fragments per k-step, 16 MFMAs per
// 16 ds_read_b64, the inner loop of k_gemm_tile<..., 64, 4>).  The pure loop above clocks down under its own load (power); a loop that also
// waits for LDS holds a higher clock, and this is the fp64 matrix rate real kernels of this code base can reach.
__global__ __launch_bounds__(256) void k_mfma_lds_f64_peak(double* sink, int iters) {
  constexpr int PITCH = 80;
  __shared__ double As[16 * PITCH], Bs[16 * PITCH];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, l15 = l & 15, l4 = l >> 4;
  for (int i = tid; i < 16 * PITCH; i += 256) {
    As[i] = 1.0 + 1e-9 * i;
    Bs[i] = 1.0 - 1e-9 * i;
  }
  __syncthreads();
  const int wi0 = 32 * (w >> 1), wj0 = 32 * (w & 1);
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = As[(4 * ks + l4) * PITCH + wi0 + 16 * a + l15];
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = Bs[(4 * ks + l4) * PITCH + wj0 + 16 * b + l15];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    asm volatile("" ::: "memory");  // the fragments are read again in every iteration
  }
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  if (s == 123.456) sink[0] = s;
}

template <typename K>
static hipError_t time_loop(K kernel, double* sink, int blocks, int iters, float* ms) {
  hipEvent_t t0, t1;
  hipEventCreate(&t0);
  hipEventCreate(&t1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, sink, 200);  // warm-up
  hipEventRecord(t0, 0);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, sink, iters);
  hipEventRecord(t1, 0);
  hipError_t e = hipEventSynchronize(t1);
  hipEventElapsedTime(ms, t0, t1);
  hipEventDestroy(t0);
  hipEventDestroy(t1);
  return e != hipSuccess ? e : hipGetLastError();
}

hipError_t run_fp64_calibration(double* mfma_flops_per_s, double* fma_flops_per_s, double* mfma_lds_fed_flops_per_s) {
  double* sink = nullptr;
  hipError_t e = hipMalloc(&sink, 8);
  if (e != hipSuccess) return e;
  hipDeviceProp_t prop;
  int dev = 0;
  hipGetDevice(&dev);
  hipGetDeviceProperties(&prop, dev);
  double best_mfma = 0.0, best_fma = 0.0, best_lds = 0.0;
  // Short bursts (1 - 3 ms) and long runs (20 ms): under a pure fp64 FMA load of tens of milliseconds the part clocks down (power), so a long
  // loop reports the SUSTAINED rate of that synthetic load while real kernels, which interleave memory and LDS work, run at a higher clock.
  // The best figure over both is the measured peak.
  for (int rep = 0; rep < 6 && e == hipSuccess; ++rep)
    for (int wps : {2, 4, 6}) {  // waves per SIMD: 256-thread blocks = 4 waves = one per SIMD of the CU
      const int blocks = prop.multiProcessorCount * wps;
      float ms = 0.f;
      const int it_m = rep == 0 ? 20000 : 2500, it_f = rep == 0 ? 40000 : 5000;
      if ((e = time_loop(k_mfma_f64_peak, sink, blocks, it_m, &ms)) != hipSuccess) break;
      best_mfma = fmax(best_mfma, 2.0 * 16 * 16 * 4 * 8.0 * it_m * 4.0 * blocks / (ms * 1e-3));
      if ((e = time_loop(k_fma_f64_peak, sink, blocks, it_f, &ms)) != hipSuccess) break;
      best_fma = fmax(best_fma, 2.0 * 16.0 * it_f * 256.0 * blocks / (ms * 1e-3));
      const int it_l = rep == 0 ? 10000 : 1250;
      if ((e = time_loop(k_mfma_lds_f64_peak, sink, blocks, it_l, &ms)) != hipSuccess) break;
      best_lds = fmax(best_lds, 2.0 * 16 * 16 * 4 * 16.0 * it_l * 4.0 * blocks / (ms * 1e-3));
    }
  hipFree(sink);
  if (e != hipSuccess) return e;
  if (mfma_flops_per_s) *mfma_flops_per_s = best_mfma;
  if (fma_flops_per_s) *fma_flops_per_s = best_fma;
  if (mfma_lds_fed_flops_per_s) *mfma_lds_fed_flops_per_s = best_lds;
  return hipSuccess;
}

}  // namespace hommx
