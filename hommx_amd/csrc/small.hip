// small.hip -- translation unit of the LDS-resident small-block kernel (small_fused.h) and its dispatch over
// (padded block size, components per node, in-plane stencil size, waves per macro cell); blocks b <= 48 go on to small_wave.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geo.h"
#include "small_fused.h"

namespace hommx {

hipError_t launch_small_fused(const Geo& G, const double* Kst, const double* Brhs, const double* C0, double* out, int32_t* info,
                              long long nc, int nw_req, hipStream_t st) {
  if (nc <= 0) return hipSuccess;
  const int nipc = G.ncode / 3;
  if (G.b <= 48 && nw_req != 2 && nw_req != 4) return launch_small_wave(G, Kst, Brhs, C0, out, info, nc, st);  // small_wave.hip
  const int bp = G.b <= 32 ? 32 : G.b <= 48 ? 48 : 64;
  const int nw = (nw_req == 2 || nw_req == 4 || (nw_req == 8 && bp == 64)) ? nw_req : (bp == 64 ? 8 : 2);  // 64: 8 waves (two tiles each) +3..5 % over 4
#define HOMMX_SF(BP_, BS_, NI_, NW_) \
  hipLaunchKernelGGL((k_small_fused<BP_, BS_, NI_, NW_>), dim3((unsigned)nc), dim3(64 * NW_), 0, st, G, Kst, Brhs, C0, out, info, nc)
#define HOMMX_SFK(BP_, NW_)                                \
  do {                                                     \
    if (G.bs == 1 && nipc == 3) HOMMX_SF(BP_, 1, 3, NW_);  \
    else if (G.bs == 2) HOMMX_SF(BP_, 2, 3, NW_);          \
    else if (G.bs == 1) HOMMX_SF(BP_, 1, 9, NW_);          \
    else HOMMX_SF(BP_, 3, 9, NW_);                         \
  } while (0)
  if (bp == 32) { if (nw == 2) HOMMX_SFK(32, 2); else HOMMX_SFK(32, 4); }
  else if (bp == 48) { if (nw == 2) HOMMX_SFK(48, 2); else HOMMX_SFK(48, 4); }
  else { if (nw == 2) HOMMX_SFK(64, 2); else if (nw == 8) HOMMX_SFK(64, 8); else HOMMX_SFK(64, 4); }
#undef HOMMX_SFK
#undef HOMMX_SF
  return hipGetLastError();
}

}  // namespace hommx
