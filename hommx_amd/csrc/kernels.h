// kernels.h -- internal launch interface between the C ABI (api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hommx {

// fused2d.hip: 2D scalar Poisson (optionally stratified), 3 <= n <= 32, one wave per macro cell.
// d_mask == nullptr: d_coef is the element stream [ncells][2 n^2]; otherwise d_coef is [ncells][2] (phase values) and
// d_mask[2 n^2] selects the phase of every element (two-phase media sampled in the kernel).
hipError_t launch_poisson2d_fused(const double* d_coef, const double* d_M, double* d_out, int32_t* d_info,
                                  int n, long long ncells, hipStream_t stream, const unsigned char* d_mask = nullptr);

// blocked.hip: coef[cell][el][comp] = mask[el] ? values[cell][1][comp] : values[cell][0][comp]
hipError_t launch_expand_two_phase(const unsigned char* d_mask, const double* d_values, double* d_coef, long long n_el,
                                   int n_comp, long long ncells, hipStream_t stream);

// calibrate.hip: sustained v_mfma_f64_16x16x4_f64 rate.
hipError_t run_fp64_mfma_calibration(double* flops_per_s);

}  // namespace hommx

namespace hommx {
// blocked.hip: generic block-cyclic path (any dim / kind / n); host-orchestrated batched kernels.
struct BlockedWorkspace;
int blocked_workspace_create(BlockedWorkspace** out, int dim, int n, int kind);
void blocked_workspace_destroy(BlockedWorkspace* ws);
int blocked_solve(BlockedWorkspace* ws, long long ncells, const double* d_coef, const double* d_M,
                  double* d_out, int32_t* d_info, hipStream_t stream, double* d_corr = nullptr);
const char* blocked_last_error();
}  // namespace hommx
