// Prints what v_permlane16_swap / v_permlane32_swap do on gfx950 (row r = lanes 16r .. 16r+15): used to document the
// row algebra the fused kernel's reductions rely on.  hipcc --offload-arch=gfx950 -o tools/bin/probe_permlane tools/probe_permlane.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = 100 + l, b = 200 + l;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"p16.a'", "p16.b'", "p32.a'", "p32.b'"};
  for (int v = 0; v < 4; ++v) {
    printf("%s rows:", names[v]);
    for (int r = 0; r < 4; ++r) printf(" [%u..%u]", h[64 * v + 16 * r], h[64 * v + 16 * r + 15]);
    printf("\n");
  }
  return 0;
}
