// DP-ALU DPP on gfx950: semantics and issue rate of `v_fmac_f64_dpp ... row_newbcast:n` (the one DPP control 64-bit VALU ops
// accept) against the plain VOP2 form.  row_newbcast:n feeds every lane of a 16-lane row with lane n of that row as src0, i.e. a
// column broadcast of an MFMA-accumulator-layout matrix at no extra instruction.
// hipcc --offload-arch=gfx950 -O2 -o tools/bin/probe_dpp64 tools/probe_dpp64.hip
#include <hip/hip_runtime.h>
#include <cstdio>

// semantics: out[l] = a + bcast_n(b) * c, and the in-place form a += bcast_n(a) * c
template <int N>
__global__ void sem(double* out) {
  const int l = threadIdx.x;
  double a = 1000.0 * l, b = l, c = 1.0;
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(c), "n"(N));
  out[l] = a;
  double d = l + 1.0;
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(c), "n"(N));
  out[64 + l] = d;  // expect (l + 1) + (16 row + N + 1)
  // back-to-back dependent chain WITHOUT nops: does the hardware interlock VALU write -> DPP read?
  double e = l, f = 1.0;
  asm volatile(
      "v_add_f64 %0, %0, 1.0\n\t"
      "v_fmac_f64_dpp %1, %0, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(e), "+v"(f) : "v"(c), "n"(N));
  out[128 + l] = f;  // expect 1 + (16 row + N + 1)
}

template <int FORM>
__global__ void rate(double* out, int iters) {
  double a[16], y = 1.0000001 + threadIdx.x * 1e-12, z = 0.25;
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = 1.0 + j + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (FORM == 0) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[j]) : "v"(y), "v"(z));
      if (FORM == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(y), "v"(z));
      // the sweep's pattern: src0 = another accumulator (written 8 instructions earlier), broadcast
      if (FORM == 2) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(a[(j + 8) & 15]), "v"(z));
      if (FORM == 3) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(a[(j + 8) & 15]));
      if (FORM == 4) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[j]) : "v"(y), "v"(z));
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
    const int blocks = 256 * wps, iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<FORM>, dim3(blocks), dim3(256), 0, 0, d, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(rate<FORM>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 16;
    printf("%-36s %d wave/SIMD: %.2f ns per wave-instr per SIMD\n", name, wps, ms * 1e6 / (ops / 1024));
  }
}

int main() {
  double* d; (void)hipMalloc(&d, 8ull * 256 * 4 * 256);
  double h[192];
  hipLaunchKernelGGL(sem<3>, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int src = 16 * (l / 16) + 3;
    if (h[l] != 1000.0 * l + src) ++bad;
    if (h[64 + l] != (l + 1.0) + (src + 1.0)) ++bad;
    if (h[128 + l] != 1.0 + (src + 1.0)) ++bad;
  }
  printf("semantics row_newbcast:3 : %s (lane 0: %.0f %.0f %.0f, lane 37: %.0f %.0f %.0f)\n", bad ? "MISMATCH" : "ok", h[0], h[64],
         h[128], h[37], h[64 + 37], h[128 + 37]);
  run<0>("v_fmac_f64 (VOP2)", d);
  run<1>("v_fmac_f64_dpp row_newbcast", d);
  run<2>("v_fmac_f64_dpp src0 = accumulator", d);
  run<3>("v_mov_b64_dpp row_newbcast", d);
  run<4>("v_fmac_f64_dpp neg src0", d);
  return bad;
}
