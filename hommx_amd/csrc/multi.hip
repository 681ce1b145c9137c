// multi.hip -- C-ABI multi-GPU entry points (include/hommx_hip.h): ONE process drives several MI355X and RCCL all-gathers the
// effective-tensor field over xGMI.  Replaces the reference's MPI partition of the macro cells (hmm.py:307-310) + PETSc assembly
// stash (hmm.py:325-330, :442) for callers without torch.distributed; the torch path is hommx_amd/dist.py.
//
// RCCL is bound lazily with dlopen: libhommx_hip.so has no link-time dependency on librccl, and a process that already loaded a
// copy (PyTorch bundles one) keeps using THAT copy -- two RCCL instances in one process do not mix.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/hommx_hip.h"

extern "C" int hommx_set_error_(int code, const char* msg);  // api.hip: thread-local message of hommx_last_error()

namespace {

typedef void* ncclComm_t;
typedef int ncclResult_t;
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
constexpr int kNcclFloat64 = 8;  // ncclDouble (rccl.h: ncclFloat64 = 8)

Rccl g_rccl;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return hommx_set_error_(code, buf);
}

int load_rccl() {
  if (g_rccl.lib) return 0;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // a copy the process already has (PyTorch's, ...)
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(HOMMX_ENODEV, "cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                   \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (!g_rccl.field) return fail(HOMMX_ENODEV, "librccl lacks %s", name)
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllGather, "ncclAllGather");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_rccl.lib = h;
  return 0;
}

#define HIP_TRY(expr)                                                                                            \
  do {                                                                                                           \
    hipError_t e__ = (expr);                                                                                     \
    if (e__ != hipSuccess)                                                                                       \
      return fail(e__ == hipErrorOutOfMemory ? HOMMX_ENOMEM : HOMMX_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
  } while (0)
#define NCCL_TRY(expr)                                                                                     \
  do {                                                                                                     \
    ncclResult_t r__ = (expr);                                                                             \
    if (r__ != 0) return fail(HOMMX_ERCCL, "%s failed: %s", #expr, g_rccl.GetErrorString(r__));            \
  } while (0)

}  // namespace

struct hommx_comm {
  int ndev = 0;
  std::vector<int> devs;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
  // scratch of hommx_solve_batch_multi, per device (grown on demand)
  std::vector<double*> d_coef, d_M, d_field;
  std::vector<int32_t*> d_info;
  std::vector<int64_t> cap_cells, cap_field;
};

extern "C" {

int hommx_comm_init_all(hommx_comm** out, int ndev, const int* devs) {
  if (!out || ndev <= 0) return fail(HOMMX_EINVAL, "hommx_comm_init_all: bad arguments");
  *out = nullptr;
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return fail(HOMMX_ENODEV, "no HIP device visible");
  if (int rc = load_rccl()) return rc;
  hommx_comm* c = new (std::nothrow) hommx_comm();
  if (!c) return fail(HOMMX_ENOMEM, "host allocation failed");
  c->ndev = ndev;
  for (int i = 0; i < ndev; ++i) {
    const int d = devs ? devs[i] : i;
    if (d < 0 || d >= have) {
      delete c;
      return fail(HOMMX_EINVAL, "device %d out of range [0,%d)", d, have);
    }
    c->devs.push_back(d);
  }
  c->comms.resize(ndev, nullptr);
  c->streams.resize(ndev, nullptr);
  c->d_coef.assign(ndev, nullptr);
  c->d_M.assign(ndev, nullptr);
  c->d_field.assign(ndev, nullptr);
  c->d_info.assign(ndev, nullptr);
  c->cap_cells.assign(ndev, 0);
  c->cap_field.assign(ndev, 0);
  ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), ndev, c->devs.data());
  if (r != 0) {
    delete c;
    return fail(HOMMX_ERCCL, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
  }
  for (int i = 0; i < ndev; ++i) {
    HIP_TRY(hipSetDevice(c->devs[i]));
    HIP_TRY(hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking));
  }
  *out = c;
  return HOMMX_OK;
}

int hommx_comm_destroy(hommx_comm* c) {
  if (!c) return HOMMX_OK;
  for (int i = 0; i < c->ndev; ++i) {
    hipSetDevice(c->devs[i]);
    if (c->d_coef[i]) hipFree(c->d_coef[i]);
    if (c->d_M[i]) hipFree(c->d_M[i]);
    if (c->d_field[i]) hipFree(c->d_field[i]);
    if (c->d_info[i]) hipFree(c->d_info[i]);
    if (c->streams[i]) hipStreamDestroy(c->streams[i]);
    if (c->comms[i] && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comms[i]);
  }
  delete c;
  return HOMMX_OK;
}

int hommx_comm_size(const hommx_comm* c) { return c ? c->ndev : 0; }

int hommx_allgather_field(hommx_comm* c, double* const* d_field_per_dev, int64_t count_per_dev) {
  if (!c || !d_field_per_dev || count_per_dev < 0) return fail(HOMMX_EINVAL, "hommx_allgather_field: bad arguments");
  if (count_per_dev == 0) return HOMMX_OK;
  NCCL_TRY(g_rccl.GroupStart());
  for (int i = 0; i < c->ndev; ++i) {
    // in place: device i's shard sits at offset i * count of its own receive buffer
    ncclResult_t r = g_rccl.AllGather(d_field_per_dev[i] + (size_t)i * count_per_dev, d_field_per_dev[i], (size_t)count_per_dev,
                                      kNcclFloat64, c->comms[i], c->streams[i]);
    if (r != 0) {
      g_rccl.GroupEnd();
      return fail(HOMMX_ERCCL, "ncclAllGather failed: %s", g_rccl.GetErrorString(r));
    }
  }
  NCCL_TRY(g_rccl.GroupEnd());
  for (int i = 0; i < c->ndev; ++i) {
    HIP_TRY(hipSetDevice(c->devs[i]));
    HIP_TRY(hipStreamSynchronize(c->streams[i]));
  }
  return HOMMX_OK;
}

// pack [A_eff (t*t doubles) | info] -> rows of t*t + 1 doubles, so that ONE collective moves both
__global__ void k_pack_field(const double* __restrict__ A, const int32_t* __restrict__ info, double* __restrict__ out, int tt,
                             long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * (tt + 1)) return;
  const long long cell = idx / (tt + 1);
  const int q = (int)(idx % (tt + 1));
  out[idx] = q < tt ? A[cell * tt + q] : (double)info[cell];
}

int hommx_solve_batch_multi(hommx_comm* c, hommx_plan* const* plans, int64_t n_cells, const double* coef, const double* M,
                            double* A_eff, int32_t* info) {
  if (!c || !plans || n_cells < 0) return fail(HOMMX_EINVAL, "hommx_solve_batch_multi: bad arguments");
  if (n_cells == 0) return HOMMX_OK;
  if (!coef || !A_eff) return fail(HOMMX_EINVAL, "null coef / A_eff");
  const int P = c->ndev;
  for (int i = 0; i < P; ++i)
    if (!plans[i]) return fail(HOMMX_EINVAL, "null plan for device slot %d", i);
  const int t = hommx_plan_tensor_size(plans[0]);
  const int64_t per_coef = hommx_plan_num_elements(plans[0]) * hommx_plan_coef_components(plans[0]);
  const int tt = t * t;
  const int d = hommx_plan_dim(plans[0]);
  const int64_t per = (n_cells + P - 1) / P;  // contiguous block partition padded to equal counts
  const int64_t row = tt + 1;
  // stage 1: every device gets ITS shard only, solves it, packs [A | info] at its slot of the gather buffer
  for (int i = 0; i < P; ++i) {
    const int64_t b = std::min<int64_t>(n_cells, i * per), e = std::min<int64_t>(n_cells, (i + 1) * per), nloc = e - b;
    HIP_TRY(hipSetDevice(c->devs[i]));
    if (per > c->cap_cells[i]) {
      if (c->d_coef[i]) hipFree(c->d_coef[i]);
      if (c->d_M[i]) hipFree(c->d_M[i]);
      if (c->d_info[i]) hipFree(c->d_info[i]);
      c->d_coef[i] = c->d_M[i] = nullptr;
      c->d_info[i] = nullptr;
      c->cap_cells[i] = 0;
      HIP_TRY(hipMalloc(&c->d_coef[i], sizeof(double) * per * per_coef));
      HIP_TRY(hipMalloc(&c->d_M[i], sizeof(double) * per * d * d));
      HIP_TRY(hipMalloc(&c->d_info[i], sizeof(int32_t) * per));
      c->cap_cells[i] = per;
    }
    const int64_t need = (int64_t)P * per * row + per * tt;
    if (need > c->cap_field[i]) {
      if (c->d_field[i]) hipFree(c->d_field[i]);
      c->d_field[i] = nullptr;
      c->cap_field[i] = 0;
      HIP_TRY(hipMalloc(&c->d_field[i], sizeof(double) * need));
      c->cap_field[i] = need;
    }
    double* gather = c->d_field[i];
    double* d_A = gather + (int64_t)P * per * row;  // per * tt doubles behind the gather buffer
    HIP_TRY(hipMemsetAsync(gather + (int64_t)i * per * row, 0, sizeof(double) * per * row, c->streams[i]));
    if (nloc > 0) {
      HIP_TRY(hipMemcpyAsync(c->d_coef[i], coef + b * per_coef, sizeof(double) * nloc * per_coef, hipMemcpyHostToDevice, c->streams[i]));
      if (M) HIP_TRY(hipMemcpyAsync(c->d_M[i], M + b * d * d, sizeof(double) * nloc * d * d, hipMemcpyHostToDevice, c->streams[i]));
      int rc = hommx_solve_batch_device(plans[i], nloc, c->d_coef[i], M ? c->d_M[i] : nullptr, d_A, c->d_info[i], c->streams[i]);
      if (rc != HOMMX_OK) return rc;
      const long long work = nloc * row;
      hipLaunchKernelGGL(k_pack_field, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, c->streams[i], d_A, c->d_info[i],
                         gather + (int64_t)i * per * row, tt, (long long)nloc);
      HIP_TRY(hipGetLastError());
    }
  }
  // stage 2: one all-gather of the packed field over RCCL (in place), then one D2H from the first device
  if (int rc = hommx_allgather_field(c, c->d_field.data(), per * row)) return rc;
  std::vector<double> host((size_t)P * per * row);
  HIP_TRY(hipSetDevice(c->devs[0]));
  HIP_TRY(hipMemcpy(host.data(), c->d_field[0], sizeof(double) * host.size(), hipMemcpyDeviceToHost));
  for (int i = 0; i < P; ++i) {
    const int64_t b = std::min<int64_t>(n_cells, i * per), e = std::min<int64_t>(n_cells, (i + 1) * per);
    for (int64_t k = b; k < e; ++k) {
      const double* src = host.data() + ((int64_t)i * per + (k - b)) * row;
      std::memcpy(A_eff + k * tt, src, sizeof(double) * tt);
      if (info) info[k] = (int32_t)src[tt];
    }
  }
  return HOMMX_OK;
}

}  // extern "C"
