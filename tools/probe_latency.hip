// Dependent-issue latency of the fp64 ops on the sweep's critical path (one wave, nothing else on the CU).
// hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_latency tools/probe_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
// the clock is read by the scalar unit: make it wait for the vector result (v_readfirstlane -> SGPR -> asm use)
#define SYNCV(v) (void)0
// clock read tied into the data flow of v: cannot move across the chain on either side; the v_readfirstlane makes the
// scalar unit wait for the vector result first
#define CLOCK(t, v)                                                                                      \
  do {                                                                                                   \
    int s_ = __builtin_amdgcn_readfirstlane(__double2hiint(v));                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(v) : "s"(s_) : "memory");       \
  } while (0)
__global__ void k(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x * 1e-9, y = 1.0000001, z = 0.25;
  long long t0, t1;
  // dependent v_fma_f64 chain
  CLOCK(t0, x);
#pragma unroll
  for (int i = 0; i < 256; ++i) x = __builtin_fma(x, y, z);
  SYNCV(x);
  CLOCK(t1, x);
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  // dependent v_mul_f64 chain
  CLOCK(t0, x);
#pragma unroll
  for (int i = 0; i < 256; ++i) x = x * y;
  SYNCV(x);
  CLOCK(t1, x);
  if (threadIdx.x == 0) cyc[1] = t1 - t0;
  // dependent v_rcp_f64 chain
  CLOCK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; ++i) { x = __builtin_amdgcn_rcp(x); asm volatile("" : "+v"(x)); }
  SYNCV(x);
  CLOCK(t1, x);
  if (threadIdx.x == 0) cyc[2] = t1 - t0;
  // readlane -> VALU round trip
  CLOCK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
    x = __builtin_fma(x, y, __hiloint2double(hi, lo));
  }
  SYNCV(x);
  CLOCK(t1, x);
  if (threadIdx.x == 0) cyc[3] = t1 - t0;
  // 16 independent fma chains (issue rate of one wave)
  double a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = x + j;
  for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(a[j]));
  CLOCK(t0, a[0]);
#pragma unroll
  for (int i = 0; i < 64; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = __builtin_fma(a[j], y, z);
  double sa = 0.0;
#pragma unroll
  for (int j = 0; j < 16; ++j) sa += a[j];
  CLOCK(t1, sa);
  x += sa * 1e-300;
  if (threadIdx.x == 0) cyc[4] = t1 - t0;
  // LDS write -> read round trip (same wave, other lane), dependent
  __shared__ double sh[64];
  CLOCK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    sh[threadIdx.x] = x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    x = sh[(threadIdx.x + 8) & 63];
    asm volatile("" : "+v"(x));
  }
  CLOCK(t1, x);
  if (threadIdx.x == 0) cyc[5] = t1 - t0;
  double s = x;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += a[j];
  out[threadIdx.x] = s;
}
int main() {
  double* d; long long* c; long long h[6];
  (void)hipMalloc(&d, 64 * 8); (void)hipMalloc(&c, 6 * 8);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, 1.5);
    (void)hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  }
  printf("cycle counter ticks (s_memtime; 100 MHz constant clock if small) per dependent op:\n");
  printf("fma_f64 %.2f  mul_f64 %.2f  rcp_f64 %.2f  readlane+fma %.2f  independent fma (per op) %.2f  lds write->read %.2f\n",
         h[0] / 256.0, h[1] / 256.0, h[2] / 64.0, h[3] / 64.0, h[4] / 1024.0, h[5] / 64.0);
  return 0;
}
