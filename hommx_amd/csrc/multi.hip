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
  std::vector<double*> d_coef, d_M, d_field, d_packed;
  std::vector<int32_t*> d_info;
  // capacities in ELEMENTS of each buffer (doubles / int32): a communicator reused with plans of another kind or n_micro must regrow
  // them even when the cell count per device stays the same
  std::vector<int64_t> cap_coef, cap_M, cap_packed, cap_field, cap_info;
};

extern "C" {

int hommx_comm_destroy(hommx_comm* c);

/* Contiguous block partition padded to equal counts: cells_i = [i * ceil(n / P), min(n, (i + 1) * ceil(n / P))).
 * Pure host arithmetic -- the one place the shard bounds and the unpack offsets of the gathered field come from. */
int hommx_shard_range(int64_t n_cells, int32_t ndev, int32_t i, int64_t* begin, int64_t* end, int64_t* per_dev) {
  if (n_cells < 0 || ndev <= 0 || i < 0 || i >= ndev) return fail(HOMMX_EINVAL, "hommx_shard_range: bad arguments");
  const int64_t per = (n_cells + ndev - 1) / ndev;
  if (begin) *begin = std::min<int64_t>(n_cells, (int64_t)i * per);
  if (end) *end = std::min<int64_t>(n_cells, (int64_t)(i + 1) * per);
  if (per_dev) *per_dev = per;
  return HOMMX_OK;
}

/* Unpack the gathered field  packed[ndev][per][t*t + 1]  (row = [A_eff | info as a double]) into A_eff[n_cells][t*t] and info. */
int hommx_unpack_field(int64_t n_cells, int32_t ndev, int32_t tt, const double* packed, double* A_eff, int32_t* info) {
  if (n_cells < 0 || ndev <= 0 || tt <= 0 || !packed || !A_eff) return fail(HOMMX_EINVAL, "hommx_unpack_field: bad arguments");
  const int64_t row = tt + 1;
  for (int i = 0; i < ndev; ++i) {
    int64_t b, e, per;
    hommx_shard_range(n_cells, ndev, i, &b, &e, &per);
    for (int64_t k = b; k < e; ++k) {
      const double* src = packed + ((int64_t)i * per + (k - b)) * row;
      std::memcpy(A_eff + k * tt, src, sizeof(double) * tt);
      if (info) info[k] = (int32_t)src[tt];
    }
  }
  return HOMMX_OK;
}

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
  c->d_packed.assign(ndev, nullptr);
  c->d_info.assign(ndev, nullptr);
  c->cap_coef.assign(ndev, 0);
  c->cap_M.assign(ndev, 0);
  c->cap_packed.assign(ndev, 0);
  c->cap_field.assign(ndev, 0);
  c->cap_info.assign(ndev, 0);
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j)
      if (c->devs[i] == c->devs[j]) {
        const int d = c->devs[i];
        delete c;
        return fail(HOMMX_EINVAL, "device %d listed twice (RCCL needs distinct devices)", d);
      }
  ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), ndev, c->devs.data());
  if (r != 0) {
    delete c;
    return fail(HOMMX_ERCCL, "ncclCommInitAll failed: %s", g_rccl.GetErrorString(r));
  }
  for (int i = 0; i < ndev; ++i) {
    hipError_t e = hipSetDevice(c->devs[i]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->streams[i], hipStreamNonBlocking);
    if (e != hipSuccess) {  // no leak: the streams made so far and the RCCL communicators go with the object
      hommx_comm_destroy(c);
      return fail(HOMMX_EHIP, "stream creation failed: %s", hipGetErrorString(e));
    }
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
    if (c->d_packed[i]) hipFree(c->d_packed[i]);
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

namespace {
// plans[i] must live on device i of the communicator and all of them must describe the same problem
int check_plans(const hommx_comm* c, hommx_plan* const* plans) {
  for (int i = 0; i < c->ndev; ++i) {
    if (!plans[i]) return fail(HOMMX_EINVAL, "null plan for device slot %d", i);
    if (hommx_plan_device(plans[i]) != c->devs[i])
      return fail(HOMMX_EINVAL, "plan %d lives on device %d, the communicator's slot %d is device %d", i, hommx_plan_device(plans[i]), i, c->devs[i]);
    if (hommx_plan_dim(plans[i]) != hommx_plan_dim(plans[0]) || hommx_plan_n_micro(plans[i]) != hommx_plan_n_micro(plans[0]) ||
        hommx_plan_kind(plans[i]) != hommx_plan_kind(plans[0]))
      return fail(HOMMX_EINVAL, "plan %d differs from plan 0 in dim / n_micro / kind", i);
  }
  return HOMMX_OK;
}

// wait for what earlier devices already have in flight before an error leaves the function
void drain(hommx_comm* c, int upto) {
  for (int i = 0; i <= upto && i < c->ndev; ++i) {
    if (hipSetDevice(c->devs[i]) == hipSuccess && c->streams[i]) (void)hipStreamSynchronize(c->streams[i]);
  }
}

// grow one per-device scratch buffer to `need` doubles (capacity tracked per buffer, in doubles)
int ensure(double** buf, int64_t* cap, int64_t need) {
  if (need > *cap) {
    if (*buf) hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(buf, sizeof(double) * need));
    *cap = need;
  }
  return HOMMX_OK;
}
int ensure_field(hommx_comm* c, int i, int64_t need) { return ensure(&c->d_field[i], &c->cap_field[i], need); }
}  // namespace

int hommx_solve_batch_multi_device(hommx_comm* c, hommx_plan* const* plans, int64_t n_cells, const double* const* d_coef_per_dev,
                                   const double* const* d_M_per_dev, double* const* d_packed_per_dev) {
  if (!c || !plans || n_cells < 0 || !d_coef_per_dev || !d_packed_per_dev) return fail(HOMMX_EINVAL, "hommx_solve_batch_multi_device: bad arguments");
  if (n_cells == 0) return HOMMX_OK;
  if (int rc = check_plans(c, plans)) return rc;
  const int P = c->ndev;
  const int t = hommx_plan_tensor_size(plans[0]), tt = t * t;
  const int64_t row = tt + 1;
  int64_t per = 0;
  hommx_shard_range(n_cells, P, 0, nullptr, nullptr, &per);
  for (int i = 0; i < P; ++i) {
    int64_t b, e;
    hommx_shard_range(n_cells, P, i, &b, &e, nullptr);
    const int64_t nloc = e - b;
    if (nloc > 0 && !d_coef_per_dev[i]) return fail(HOMMX_EINVAL, "null coefficient shard for device slot %d", i);
    if (!d_packed_per_dev[i]) return fail(HOMMX_EINVAL, "null output buffer for device slot %d", i);
    int rc = HOMMX_OK;
    do {
      if (hipSetDevice(c->devs[i]) != hipSuccess) { rc = fail(HOMMX_EHIP, "hipSetDevice(%d) failed", c->devs[i]); break; }
      if ((rc = ensure_field(c, i, per * tt)) != HOMMX_OK) break;   // A_eff of the shard before packing
      if (per > c->cap_info[i]) {
        if (c->d_info[i]) hipFree(c->d_info[i]);
        c->d_info[i] = nullptr;
        c->cap_info[i] = 0;
        if (hipMalloc(&c->d_info[i], sizeof(int32_t) * per) != hipSuccess) { rc = fail(HOMMX_ENOMEM, "info scratch on device %d", c->devs[i]); break; }
        c->cap_info[i] = per;
      }
      double* slot = d_packed_per_dev[i] + (int64_t)i * per * row;
      if (hipMemsetAsync(slot, 0, sizeof(double) * per * row, c->streams[i]) != hipSuccess) { rc = fail(HOMMX_EHIP, "memset failed"); break; }
      if (nloc > 0) {
        rc = hommx_solve_batch_device(plans[i], nloc, d_coef_per_dev[i], d_M_per_dev ? d_M_per_dev[i] : nullptr, c->d_field[i], c->d_info[i],
                                      c->streams[i]);
        if (rc != HOMMX_OK) break;
        const long long work = nloc * row;
        hipLaunchKernelGGL(k_pack_field, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, c->streams[i], c->d_field[i], c->d_info[i], slot,
                           tt, (long long)nloc);
        if (hipGetLastError() != hipSuccess) { rc = fail(HOMMX_EHIP, "pack kernel launch failed"); break; }
      }
    } while (0);
    if (rc != HOMMX_OK) {
      drain(c, i);
      return rc;
    }
  }
  return hommx_allgather_field(c, d_packed_per_dev, per * row);
}

int hommx_solve_batch_multi(hommx_comm* c, hommx_plan* const* plans, int64_t n_cells, const double* coef, const double* M,
                            double* A_eff, int32_t* info) {
  if (!c || !plans || n_cells < 0) return fail(HOMMX_EINVAL, "hommx_solve_batch_multi: bad arguments");
  if (n_cells == 0) return HOMMX_OK;
  if (!coef || !A_eff) return fail(HOMMX_EINVAL, "null coef / A_eff");
  if (int rc = check_plans(c, plans)) return rc;
  const int P = c->ndev;
  const int t = hommx_plan_tensor_size(plans[0]);
  const int64_t per_coef = hommx_plan_num_elements(plans[0]) * hommx_plan_coef_components(plans[0]);
  const int tt = t * t;
  const int d = hommx_plan_dim(plans[0]);
  int64_t per = 0;
  hommx_shard_range(n_cells, P, 0, nullptr, nullptr, &per);
  const int64_t row = tt + 1;
  // every device gets ITS shard only (H2D on its own stream), then the device-pointer form solves, packs and gathers
  std::vector<const double*> coefs(P, nullptr), Ms(P, nullptr);
  std::vector<double*> packed(P, nullptr);
  for (int i = 0; i < P; ++i) {
    int64_t b, e;
    hommx_shard_range(n_cells, P, i, &b, &e, nullptr);
    const int64_t nloc = e - b;
    int rc = HOMMX_OK;
    do {
      if (hipSetDevice(c->devs[i]) != hipSuccess) { rc = fail(HOMMX_EHIP, "hipSetDevice(%d) failed", c->devs[i]); break; }
      // sizes depend on the plans (per_coef, d, row), not only on the cells per device: one capacity per buffer
      if ((rc = ensure(&c->d_coef[i], &c->cap_coef[i], per * per_coef)) != HOMMX_OK || (rc = ensure(&c->d_M[i], &c->cap_M[i], per * d * d)) != HOMMX_OK ||
          (rc = ensure(&c->d_packed[i], &c->cap_packed[i], (int64_t)P * per * row)) != HOMMX_OK)
        break;
      if (nloc > 0) {
        if (hipMemcpyAsync(c->d_coef[i], coef + b * per_coef, sizeof(double) * nloc * per_coef, hipMemcpyHostToDevice, c->streams[i]) != hipSuccess ||
            (M && hipMemcpyAsync(c->d_M[i], M + b * d * d, sizeof(double) * nloc * d * d, hipMemcpyHostToDevice, c->streams[i]) != hipSuccess)) {
          rc = fail(HOMMX_EHIP, "H2D copy to device %d failed", c->devs[i]);
          break;
        }
      }
    } while (0);
    if (rc != HOMMX_OK) {
      drain(c, i);
      return rc;
    }
    coefs[i] = c->d_coef[i];
    Ms[i] = M ? c->d_M[i] : nullptr;
    packed[i] = c->d_packed[i];
  }
  if (int rc = hommx_solve_batch_multi_device(c, plans, n_cells, coefs.data(), M ? Ms.data() : nullptr, packed.data())) return rc;
  std::vector<double> host((size_t)P * per * row);
  HIP_TRY(hipSetDevice(c->devs[0]));
  HIP_TRY(hipMemcpy(host.data(), c->d_packed[0], sizeof(double) * host.size(), hipMemcpyDeviceToHost));
  return hommx_unpack_field(n_cells, P, tt, host.data(), A_eff, info);
}

}  // extern "C"
