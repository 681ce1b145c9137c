// blocked.hip -- generic block-cyclic path (placeholder until the kernels land: fails loudly).
#include "kernels.h"
#include "../../include/hommx_hip.h"
namespace hommx {
struct BlockedWorkspace { int dim, n, kind; };
static thread_local const char* g_berr = "";
const char* blocked_last_error() { return g_berr; }
int blocked_workspace_create(BlockedWorkspace** out, int, int, int) {
  *out = nullptr;
  g_berr = "configuration not implemented yet (only 2D scalar Poisson, 3 <= n_micro <= 32)";
  return HOMMX_EINVAL;
}
void blocked_workspace_destroy(BlockedWorkspace* ws) { delete ws; }
int blocked_solve(BlockedWorkspace*, long long, const double*, const double*, double*, int32_t*, hipStream_t) {
  g_berr = "not implemented";
  return HOMMX_EINVAL;
}
}  // namespace hommx
