// mf_front.hip -- dispatcher of the register-resident front kernel (mf_front_kernel.h; one translation unit per number of unknowns per node)
#include "mf_front.h"

#include <hip/hip_runtime.h>

namespace hommx {

template <int BS>
void launch_mf_front_bs(const MfFrontDev& g, const double* Kst, const double* Brhs, double* arena, long long nc, long long nbatch, int nn,
                        int ncode, int t, int32_t* info, int stepcode, hipStream_t st);
extern template void launch_mf_front_bs<1>(const MfFrontDev&, const double*, const double*, double*, long long, long long, int, int, int, int32_t*, int, hipStream_t);
extern template void launch_mf_front_bs<2>(const MfFrontDev&, const double*, const double*, double*, long long, long long, int, int, int, int32_t*, int, hipStream_t);
extern template void launch_mf_front_bs<3>(const MfFrontDev&, const double*, const double*, double*, long long, long long, int, int, int, int32_t*, int, hipStream_t);

void launch_mf_front(const MfFrontDev& g, int bs, const double* Kst, const double* Brhs, double* arena, long long nc, long long nbatch, int nn,
                     int ncode, int t, int32_t* info, int stepcode, hipStream_t st) {
  if (bs == 1) launch_mf_front_bs<1>(g, Kst, Brhs, arena, nc, nbatch, nn, ncode, t, info, stepcode, st);
  else if (bs == 2) launch_mf_front_bs<2>(g, Kst, Brhs, arena, nc, nbatch, nn, ncode, t, info, stepcode, st);
  else launch_mf_front_bs<3>(g, Kst, Brhs, arena, nc, nbatch, nn, ncode, t, info, stepcode, st);
}


}  // namespace hommx
