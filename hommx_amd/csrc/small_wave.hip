// small_wave.hip -- translation unit of the one-wave-per-macro-cell register kernel (small_wave.h, plane blocks b <= 48) and its
// dispatch over (tiles per dimension, components per node, in-plane stencil size, bordered arrow or slab-form load rows).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geo.h"
#include "small_wave.h"

namespace hommx {

hipError_t launch_small_wave(const Geo& G, const double* Kst, const double* Brhs, const double* C0, double* out, int32_t* info,
                             long long nc, hipStream_t st) {
  if (nc <= 0) return hipSuccess;
  const int nipc = G.ncode / 3;
  const int nt = (G.b + 15) / 16;
  const bool aug = G.b + G.t <= 16 * nt;  // the load rows ride in the padding columns of the arrow
#define HOMMX_SW(NT_, BS_, NI_)                                                                                                   \
  do {                                                                                                                            \
    if (aug) hipLaunchKernelGGL((k_small_wave<NT_, BS_, NI_, true>), dim3((unsigned)nc), dim3(64), 0, st, G, Kst, Brhs, C0, out, info, nc); \
    else hipLaunchKernelGGL((k_small_wave<NT_, BS_, NI_, false>), dim3((unsigned)nc), dim3(64), 0, st, G, Kst, Brhs, C0, out, info, nc);    \
  } while (0)
#define HOMMX_SWK(NT_)                                \
  do {                                                \
    if (G.bs == 1 && nipc == 3) HOMMX_SW(NT_, 1, 3);  \
    else if (G.bs == 2) HOMMX_SW(NT_, 2, 3);          \
    else if (G.bs == 1) HOMMX_SW(NT_, 1, 9);          \
    else HOMMX_SW(NT_, 3, 9);                         \
  } while (0)
  if (nt == 1) HOMMX_SWK(1); else if (nt == 2) HOMMX_SWK(2); else HOMMX_SWK(3);
#undef HOMMX_SWK
#undef HOMMX_SW
  return hipGetLastError();
}

}  // namespace hommx
