// geo.h -- geometry / stencil descriptor shared by the kernel files of the blocked family (blocked.hip, small.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace hommx {

typedef double d4 __attribute__((ext_vector_type(4)));

struct Geo {
  int dim, n, bs, t, kind, ncomp;
  int nn, npl, b, Bp, nsub, ncode, n_el;
  int voff[6][4][3];     // corner offset of local vertex a of sub-element s
  double grad[6][4][3];  // gradient of its P1 basis function on the unit-size cell (h = 1)
};

// in-plane neighbour q' of in-plane node q for in-plane code ipc
__device__ __forceinline__ int plane_neighbour(const Geo& G, int q, int ipc) {
  const int n = G.n;
  if (G.dim == 2) {
    const int o = ipc - 1;
    return (q + o + n) % n;
  }
  const int ox = ipc % 3 - 1, oy = ipc / 3 - 1;
  const int i = (q % n + ox + n) % n, j = (q / n + oy + n) % n;
  return i + n * j;
}


// small.hip: LDS-resident elimination for plane blocks b <= 64 (small_fused.h); nw = waves per macro cell (0: default)
hipError_t launch_small_fused(const Geo& G, const double* Kst, const double* Brhs, const double* C0, double* out, int32_t* info,
                              long long ncells, int nw, hipStream_t stream);

// small_wave.hip: register-resident elimination for plane blocks b <= 48, one wavefront per macro cell (small_wave.h)
hipError_t launch_small_wave(const Geo& G, const double* Kst, const double* Brhs, const double* C0, double* out, int32_t* info,
                             long long ncells, hipStream_t stream);

}  // namespace hommx
