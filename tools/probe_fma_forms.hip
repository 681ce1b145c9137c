// Issue rate of the fp64 FMA encodings on gfx950, all CUs busy: v_fmac_f64 (VOP2, in place) vs v_fma_f64 (VOP3, separate
// destination / neg modifier), at 1, 2 and 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_fma_forms ...
#include <hip/hip_runtime.h>
#include <cstdio>
template <int FORM>
__global__ void k(double* out, int iters) {
  double a[16], y = 1.0000001 + threadIdx.x * 1e-12, z = 0.25;
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = 1.0 + j + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (FORM == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[j]) : "v"(y), "v"(z));
      if (FORM == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[j]) : "v"(y), "v"(z));
      if (FORM == 2) asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(a[j]) : "v"(y), "v"(z));
      if (FORM == 3) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[j]) : "v"(y), "v"(z), "v"(a[(j + 1) & 15]));
      if (FORM == 4) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[j]) : "v"(y), "v"(a[(j + 1) & 15]));
      if (FORM == 5) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(((int*)a)[2 * j]) : "v"(((int*)a)[(2 * j + 2) & 31]), "v"(((int*)a)[(2 * j + 5) & 31]));
      if (FORM == 6) asm volatile("v_mov_b64 %0, %1" : "=v"(a[j]) : "v"(a[(j + 1) & 15]));
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int FORM>
void run(const char* name, double* d) {
  for (int wps : {1, 2, 4}) {
    const int blocks = 256 * wps, iters = 20000;  // 256-thread blocks: 4 waves = one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<FORM>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 16;  // wave-level instructions
    printf("%-28s %d wave/SIMD: %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wps, ms * 1e6 / (ops / 1024), ms * 1e6 / (ops / 1024) * 2.4);
  }
}
int main() {
  double* d; (void)hipMalloc(&d, 8ull * 256 * 4 * 256);
  run<0>("v_fmac_f64 (VOP2)", d);
  run<1>("v_fma_f64 d=c (VOP3)", d);
  run<2>("v_fma_f64 -a (VOP3 neg)", d);
  run<3>("v_fma_f64 d!=c (VOP3)", d);
  run<4>("v_mul_f64", d);
  run<5>("v_cndmask_b32", d);
  run<6>("v_mov_b64", d);
  return 0;
}
