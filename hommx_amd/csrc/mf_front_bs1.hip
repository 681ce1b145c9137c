// mf_front_bs1.hip -- the register-resident front kernel for 1 unknown per node (mf_front_kernel.h)
#include "mf_front_kernel.h"

namespace hommx {
template void launch_mf_front_bs<1>(const MfFrontDev&, const double*, const double*, double*, long long, long long, int, int, int, int32_t*, int,
                                    hipStream_t);
}  // namespace hommx
