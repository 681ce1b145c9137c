// kernels.h -- internal launch interface between the C ABI (api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hommx {

// Where the per-element coefficient of a scalar Poisson cell comes from (device samplers).
//   STREAM      coef[cell][n_el]: element means sampled by the caller
//   TWO_PHASE   mask[n_el] (uint8) selects coef[cell][0 / 1]
//   AFFINE      A_K = a + b * table[K]                      coef[cell] = (a, b); table[n_el] = element means of g(y)
//   RECIPROCAL  A_K = sum_q w[q] / (a + b * table[K][q])    coef[cell] = (a, b); table[n_el][nq] = g at the quadrature points
// Every operation of AFFINE / RECIPROCAL is a separately rounded IEEE operation in a fixed order, so a host that evaluates the
// same formula (hommx_amd.hmm.Separable.host_stream) gets the same bits.
// separately rounded IEEE operations: hipcc contracts a * b + c into an fma by default (and __dmul_rn / __dadd_rn are plain
// operators in the HIP headers), which a host evaluating the same formula with NumPy does not do
__device__ __forceinline__ double mul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double add_rn(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ double div_rn(double a, double b) {
#pragma clang fp contract(off)
  return a / b;
}
enum CoefMode { COEF_STREAM = 0, COEF_TWO_PHASE = 1, COEF_AFFINE = 2, COEF_RECIPROCAL = 3 };
struct CoefSource {
  int mode = COEF_STREAM;
  int nq = 0;
  const void* table = nullptr;    // mask (uint8) or table (double)
  const double* weights = nullptr;
};

// fused2d.hip: 2D scalar Poisson (optionally stratified), 3 <= n <= 32, one wave per macro cell.
hipError_t launch_poisson2d_fused(const double* d_coef, const double* d_M, double* d_out, int32_t* d_info,
                                  int n, long long ncells, hipStream_t stream, CoefSource src = CoefSource());

// blocked.hip: coef[cell][el][comp] of a separable coefficient (AFFINE / RECIPROCAL; params[cell][comp] = (a, b)) expanded into the element stream
hipError_t launch_expand_separable(CoefSource src, const double* d_params, double* d_coef, long long n_el, int n_comp, long long ncells,
                                   hipStream_t stream);

// blocked.hip: coef[cell][el][comp] = mask[el] ? values[cell][1][comp] : values[cell][0][comp]
hipError_t launch_expand_two_phase(const unsigned char* d_mask, const double* d_values, double* d_coef, long long n_el,
                                   int n_comp, long long ncells, hipStream_t stream);

// calibrate.hip: best sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates over 2 and 4 waves per SIMD.
hipError_t run_fp64_calibration(double* mfma_flops_per_s, double* fma_flops_per_s, double* mfma_lds_fed_flops_per_s = nullptr);

}  // namespace hommx

namespace hommx {
// blocked.hip: generic block-cyclic path (any dim / kind / n); host-orchestrated batched kernels.
struct BlockedWorkspace;
int blocked_workspace_create(BlockedWorkspace** out, int dim, int n, int kind);
void blocked_workspace_destroy(BlockedWorkspace* ws);
int blocked_solve(BlockedWorkspace* ws, long long ncells, const double* d_coef, const double* d_M,
                  double* d_out, int32_t* d_info, hipStream_t stream, double* d_corr = nullptr);
const char* blocked_last_error();
// allocate the workspace of the route for batches of up to n_cells (what the first solve would otherwise do)
int blocked_reserve(BlockedWorkspace* ws, long long n_cells);
// "small_wave" (b <= 48), "small_fused" (48 < b <= 64) or "blocked": the route blocked_solve takes for effective tensors
const char* blocked_route_name(const BlockedWorkspace* ws);
const char* blocked_route_detail(BlockedWorkspace* ws);
// dense flops one micro-cell solve executes on this route, by the route's own model (multifrontal: sum over the fronts of
// s^3 + 2 s^2 r + s r^2 on the padded sizes; plane elimination: (6 (n - 1) + 2) b^3)
double blocked_flops_per_cell(const BlockedWorkspace* ws);
}  // namespace hommx
