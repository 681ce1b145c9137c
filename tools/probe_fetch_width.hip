// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths this code base uses (MI355X_MICROARCH.md: "FETCH_SIZE reports
// exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Every kernel reads the same 1 GiB buffer (4 x the Infinity Cache) exactly once:
//   k_read16   16 B per lane, contiguous per wave (global_load_dwordx4: the GEMM operand loads)
//   k_read8     8 B per lane, contiguous per wave (global_load_dwordx2: k_mf_build, the gather epilogue along a row)
//   k_read8_rows  8 B per lane, the 64 lanes of a wave in 4 different rows of 16 consecutive doubles (128-B segments: the gather epilogue's
//                 accumulator-layout reads of a child's update matrix)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_fetch_width.hip -o tools/bin/probe_fetch_width
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetchw -o run -- tools/bin/probe_fetch_width
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_read16(const double2* __restrict__ p, double* out, long long n2) {
  double s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x) {
    double2 v = p[i];
    s += v.x + v.y;
  }
  if (s == 12345.678) out[0] = s;
}
__global__ void k_read8(const double* __restrict__ p, double* out, long long n) {
  double s = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) s += p[i];
  if (s == 12345.678) out[0] = s;
}
// rows of `ld` doubles; a wave reads a 4 x 16 patch (lane = 16 * r + c -> row 4 * R + r, column 16 * C + c), patches walk the matrix
__global__ void k_read8_rows(const double* __restrict__ p, double* out, long long rows, int ld) {
  double s = 0;
  const int lane = threadIdx.x & 63, r = lane >> 4, c = lane & 15;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long pc = ld / 16, npatch = (rows / 4) * pc;
  for (long long q = wave; q < npatch; q += nwaves) {
    const long long R = q / pc, C = q % pc;
    s += p[(4 * R + r) * ld + 16 * C + c];
  }
  if (s == 12345.678) out[0] = s;
}

int main() {
  const long long n = 1ll << 27;  // doubles = 1 GiB
  double *p, *out;
  hipMalloc(&p, n * 8);
  hipMalloc(&out, 8);
  hipMemset(p, 0, n * 8);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_read16, dim3(256 * 16), dim3(256), 0, 0, (const double2*)p, out, n / 2);
    hipLaunchKernelGGL(k_read8, dim3(256 * 16), dim3(256), 0, 0, p, out, n);
    hipLaunchKernelGGL(k_read8_rows, dim3(256 * 16), dim3(256), 0, 0, p, out, n / 1024, 1024);
  }
  hipDeviceSynchronize();
  printf("each kernel read %lld bytes\n", n * 8);
  return 0;
}
