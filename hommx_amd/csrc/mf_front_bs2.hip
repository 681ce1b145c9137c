// mf_front_bs2.hip -- the register-resident front kernel for 2 unknowns per node (mf_front_kernel.h)
#include "mf_front_kernel.h"

namespace hommx {
template void launch_mf_front_bs<2>(const MfFrontDev&, const double*, const double*, double*, long long, long long, int, int, int, int32_t*, int,
                                    hipStream_t);
}  // namespace hommx
