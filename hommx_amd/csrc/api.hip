// api.hip -- the C ABI of libhommx_hip.so (include/hommx_hip.h): plan objects, buffers, dispatch.
//
// Boundary it implements: the macro-cell loop BaseHMM._assemble_stiffness (hmm.py:298-332) calling
// _compute_local_stiffness (hmm.py:334-369) once per cell.  Here: one call per batch of cells.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "../../include/hommx_hip.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess)                                                                     \
      return fail(e__ == hipErrorOutOfMemory ? HOMMX_ENOMEM : HOMMX_EHIP, "%s failed: %s", #expr, \
                  hipGetErrorString(e__));                                                     \
  } while (0)

enum Family { FAM_FUSED2D = 0, FAM_BLOCKED = 1 };

}  // namespace

struct hommx_plan;
namespace {
template <typename B>
int grow_pinned(B& b, size_t bytes) {
  if (bytes <= b.cap) return HOMMX_OK;
  if (b.p) (void)hipHostFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  HIP_TRY(hipHostMalloc(&b.p, bytes, hipHostMallocDefault));
  b.cap = bytes;
  return HOMMX_OK;
}

template <typename B>
int grow(B& b, size_t bytes) {
  if (bytes <= b.cap) return HOMMX_OK;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  HIP_TRY(hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return HOMMX_OK;
}
}  // namespace

struct hommx_plan {
  hommx_plan_desc desc;
  Family family;
  int64_t n_el;
  int32_t n_comp;
  int32_t t;
  // staging buffers for the host-pointer entry point (grown on demand)
  double* d_coef = nullptr;
  double* d_M = nullptr;
  double* d_out = nullptr;
  int32_t* d_info = nullptr;
  int64_t cap_cells = 0;
  hommx::BlockedWorkspace* ws = nullptr;
  double* d_expand = nullptr;  // two-phase media on the blocked family: expanded element stream
  int64_t cap_expand = 0;
  // plan-owned staging of the sampler entry points (two-phase / separable, host pointers): grown on demand, never per call
  struct Buf {
    void* p = nullptr;
    size_t cap = 0;
  };
  Buf st_mask, st_values, st_table, st_w, st_M, st_out, st_info;
  // pinned host mirrors of the small sampler inputs / outputs: one asynchronous H2D, the kernel, one asynchronous D2H, ONE synchronisation
  Buf pin_in, pin_out, dev_in, dev_out;
  // host-pointer entry point of the fused family: coefficient chunks stream in on s_copy while s_comp solves the previous one
  hipStream_t s_copy = nullptr, s_comp = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool h2d_overlap = true;  // HOMMX_NO_H2D_OVERLAP (dev knob, read when the plan is created) switches the pipelining off
};

extern "C" {

int hommx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* hommx_last_error(void) { return g_err.c_str(); }

int hommx_plan_create(hommx_plan** out, const hommx_plan_desc* d) {
  if (!out || !d) return fail(HOMMX_EINVAL, "null argument");
  *out = nullptr;
  if (d->dim != 2 && d->dim != 3) return fail(HOMMX_EINVAL, "dim must be 2 or 3 (hmm.py:104-105), got %d", d->dim);
  if (d->kind < 0 || d->kind > 3) return fail(HOMMX_EINVAL, "unknown kind %d", d->kind);
  if (d->n_micro < 3) return fail(HOMMX_EINVAL, "n_micro must be >= 3 (got %d): with fewer cells per side periodic neighbours coincide", d->n_micro);
  int ndev = hommx_device_count();
  if (ndev <= 0) return fail(HOMMX_ENODEV, "no HIP device visible");
  if (d->device < 0 || d->device >= ndev) return fail(HOMMX_EINVAL, "device %d out of range [0,%d)", d->device, ndev);

  hommx_plan* p = new (std::nothrow) hommx_plan();
  if (!p) return fail(HOMMX_ENOMEM, "host allocation failed");
  p->desc = *d;
  const int dim = d->dim, n = d->n_micro;
  p->n_el = (dim == 2) ? 2ll * n * n : 6ll * n * n * n;
  const int tp = dim;                    // Poisson tensor size
  const int te = dim * (dim + 1) / 2;    // elasticity (Voigt) tensor size
  switch (d->kind) {
    case HOMMX_KIND_POISSON_SCALAR: p->n_comp = 1; p->t = tp; break;
    case HOMMX_KIND_POISSON_MATRIX: p->n_comp = dim * (dim + 1) / 2; p->t = tp; break;
    case HOMMX_KIND_ELASTICITY_ISO: p->n_comp = 2; p->t = te; break;
    default: p->n_comp = te * (te + 1) / 2; p->t = te; break;
  }
  p->h2d_overlap = getenv("HOMMX_NO_H2D_OVERLAP") == nullptr;
  p->family = (dim == 2 && d->kind == HOMMX_KIND_POISSON_SCALAR && n <= 32 && !(d->flags & 1)) ? FAM_FUSED2D : FAM_BLOCKED;
  if (p->family == FAM_BLOCKED) {
    int rc = hommx::blocked_workspace_create(&p->ws, dim, n, d->kind);
    if (rc != 0) {
      delete p;
      return fail(rc, "blocked path: %s", hommx::blocked_last_error());
    }
  }
  *out = p;
  return HOMMX_OK;
}

int hommx_plan_destroy(hommx_plan* p) {
  if (!p) return HOMMX_OK;
  hipSetDevice(p->desc.device);
  if (p->d_coef) hipFree(p->d_coef);
  if (p->d_M) hipFree(p->d_M);
  if (p->d_out) hipFree(p->d_out);
  if (p->d_info) hipFree(p->d_info);
  if (p->ws) hommx::blocked_workspace_destroy(p->ws);
  if (p->d_expand) hipFree(p->d_expand);
  for (hommx_plan::Buf* b : {&p->st_mask, &p->st_values, &p->st_table, &p->st_w, &p->st_M, &p->st_out, &p->st_info, &p->dev_in, &p->dev_out})
    if (b->p) hipFree(b->p);
  for (hommx_plan::Buf* b : {&p->pin_in, &p->pin_out})
    if (b->p) hipHostFree(b->p);
  if (p->s_copy) hipStreamDestroy(p->s_copy);
  if (p->s_comp) hipStreamDestroy(p->s_comp);
  for (hipEvent_t e : p->ev)
    if (e) hipEventDestroy(e);
  delete p;
  return HOMMX_OK;
}

// multi.hip reports its errors through the same thread-local message
int hommx_set_error_(int code, const char* msg) { return fail(code, "%s", msg); }

int32_t hommx_plan_dim(const hommx_plan* p) { return p ? p->desc.dim : 0; }
int32_t hommx_plan_device(const hommx_plan* p) { return p ? p->desc.device : -1; }
int32_t hommx_plan_n_micro(const hommx_plan* p) { return p ? p->desc.n_micro : 0; }
int32_t hommx_plan_kind(const hommx_plan* p) { return p ? p->desc.kind : -1; }
int64_t hommx_plan_num_elements(const hommx_plan* p) { return p ? p->n_el : 0; }
int32_t hommx_plan_coef_components(const hommx_plan* p) { return p ? p->n_comp : 0; }
int32_t hommx_plan_tensor_size(const hommx_plan* p) { return p ? p->t : 0; }
int hommx_plan_reserve(hommx_plan* p, int64_t n_cells) {
  if (!p) return fail(HOMMX_EINVAL, "null plan");
  if (n_cells < 0) return fail(HOMMX_EINVAL, "negative n_cells");
  if (p->family == FAM_FUSED2D || n_cells == 0) return HOMMX_OK;  // the fused 2D family keeps no scratch
  HIP_TRY(hipSetDevice(p->desc.device));
  int rc = hommx::blocked_reserve(p->ws, n_cells);
  if (rc != 0) return fail(rc, "blocked path: %s", hommx::blocked_last_error());
  return HOMMX_OK;
}

double hommx_plan_flops_per_solve(const hommx_plan* p) {
  if (!p) return 0.0;
  if (p->family == FAM_FUSED2D) {
    const double n = p->desc.n_micro;
    return (6.0 * (n - 1) + 2.0) * n * n * n;
  }
  return hommx::blocked_flops_per_cell(p->ws);
}
const char* hommx_plan_kernel_name(const hommx_plan* p) {
  if (!p) return "";
  return p->family == FAM_FUSED2D ? "fused2d" : hommx::blocked_route_name(p->ws);
}

const char* hommx_plan_route_detail(hommx_plan* p) {
  if (!p) return "";
  if (p->family == FAM_FUSED2D)
    return p->desc.n_micro > 16 ? "fused2d: k_poisson2d_fused<32>, one wavefront per macro cell, every matrix in the f64 MFMA accumulator layout"
                                : "fused2d: k_poisson2d_fused<16>, one wavefront per macro cell, every matrix in the f64 MFMA accumulator layout";
  return hommx::blocked_route_detail(p->ws);
}

int hommx_solve_batch_device(hommx_plan* p, int64_t n_cells, const double* d_coef, const double* d_M,
                             double* d_A_eff, int32_t* d_info, void* stream) {
  if (!p) return fail(HOMMX_EINVAL, "null plan");
  if (n_cells < 0) return fail(HOMMX_EINVAL, "negative n_cells");
  if (n_cells == 0) return HOMMX_OK;
  if (!d_coef || !d_A_eff) return fail(HOMMX_EINVAL, "null coef / A_eff");
  if (n_cells > 0x7fffffffll) return fail(HOMMX_EINVAL, "n_cells too large for one launch");
  HIP_TRY(hipSetDevice(p->desc.device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p->family == FAM_FUSED2D) {
    HIP_TRY(hommx::launch_poisson2d_fused(d_coef, d_M, d_A_eff, d_info, p->desc.n_micro, n_cells, st));
    return HOMMX_OK;
  }
  int rc = hommx::blocked_solve(p->ws, n_cells, d_coef, d_M, d_A_eff, d_info, st);
  if (rc != 0) return fail(rc, "blocked path: %s", hommx::blocked_last_error());
  return HOMMX_OK;
}

int hommx_solve_batch(hommx_plan* p, int64_t n_cells, const double* coef, const double* M, double* A_eff,
                      int32_t* info) {
  if (!p) return fail(HOMMX_EINVAL, "null plan");
  if (n_cells < 0) return fail(HOMMX_EINVAL, "negative n_cells");
  if (n_cells == 0) return HOMMX_OK;
  if (!coef || !A_eff) return fail(HOMMX_EINVAL, "null coef / A_eff");
  HIP_TRY(hipSetDevice(p->desc.device));
  const int d = p->desc.dim, t = p->t;
  if (n_cells > p->cap_cells) {
    if (p->d_coef) hipFree(p->d_coef);
    if (p->d_M) hipFree(p->d_M);
    if (p->d_out) hipFree(p->d_out);
    if (p->d_info) hipFree(p->d_info);
    p->d_coef = p->d_M = p->d_out = nullptr;
    p->d_info = nullptr;
    p->cap_cells = 0;
    HIP_TRY(hipMalloc(&p->d_coef, sizeof(double) * n_cells * p->n_el * p->n_comp));
    HIP_TRY(hipMalloc(&p->d_M, sizeof(double) * n_cells * d * d));
    HIP_TRY(hipMalloc(&p->d_out, sizeof(double) * n_cells * t * t));
    HIP_TRY(hipMalloc(&p->d_info, sizeof(int32_t) * n_cells));
    p->cap_cells = n_cells;
  }
  if (M) HIP_TRY(hipMemcpy(p->d_M, M, sizeof(double) * n_cells * d * d, hipMemcpyHostToDevice));
  const int64_t per = p->n_el * p->n_comp;
  // cells per chunk of the pipelined copy.  Fused 2D kernel: one wave per cell fills the 256 CUs x 8 wave slots exactly once; blocked
  // family: about 256 MB of coefficient stream, at least 256 cells (from there its throughput is flat: C4 / C5 393 KB per cell -> 682)
  int64_t CH = 2048;
  if (p->family != FAM_FUSED2D) {
    CH = (int64_t)((256ll << 20) / (8 * (per > 0 ? per : 1)));
    CH = CH < 256 ? 256 : CH > 4096 ? 4096 : CH;
  }
  if (n_cells >= 2 * CH && p->h2d_overlap) {
    // The coefficient stream (16 KiB per 2D cell, 393 KB per 16^3 elasticity cell) costs PCIe time -- more than the fused kernel costs GPU
    // time, 4 - 8 % of the 3D solves: pipeline it.  The copies are issued from pageable memory, so each blocks this thread -- while the
    // kernels of the previous chunk, already queued on the other stream, run.
    if (!p->s_copy) {
      HIP_TRY(hipStreamCreateWithFlags(&p->s_copy, hipStreamNonBlocking));
      HIP_TRY(hipStreamCreateWithFlags(&p->s_comp, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&p->ev[0], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&p->ev[1], hipEventDisableTiming));
    }
    int k = 0;
    for (int64_t c0 = 0; c0 < n_cells; c0 += CH, k ^= 1) {
      const int64_t nc = (n_cells - c0 < CH) ? n_cells - c0 : CH;
      HIP_TRY(hipMemcpyAsync(p->d_coef + c0 * per, coef + c0 * per, sizeof(double) * nc * per, hipMemcpyHostToDevice, p->s_copy));
      HIP_TRY(hipEventRecord(p->ev[k], p->s_copy));
      HIP_TRY(hipStreamWaitEvent(p->s_comp, p->ev[k], 0));
      int rc = hommx_solve_batch_device(p, nc, p->d_coef + c0 * per, M ? p->d_M + c0 * d * d : nullptr, p->d_out + c0 * t * t,
                                        p->d_info + c0, p->s_comp);
      if (rc != HOMMX_OK) return rc;
    }
    HIP_TRY(hipStreamSynchronize(p->s_comp));
  } else {
    HIP_TRY(hipMemcpy(p->d_coef, coef, sizeof(double) * n_cells * per, hipMemcpyHostToDevice));
    int rc = hommx_solve_batch_device(p, n_cells, p->d_coef, M ? p->d_M : nullptr, p->d_out, p->d_info, nullptr);
    if (rc != HOMMX_OK) return rc;
    HIP_TRY(hipDeviceSynchronize());
  }
  HIP_TRY(hipMemcpy(A_eff, p->d_out, sizeof(double) * n_cells * t * t, hipMemcpyDeviceToHost));
  if (info) HIP_TRY(hipMemcpy(info, p->d_info, sizeof(int32_t) * n_cells, hipMemcpyDeviceToHost));
  return HOMMX_OK;
}

int hommx_solve_batch_two_phase_device(hommx_plan* p, int64_t n_cells, const uint8_t* d_mask, const double* d_values,
                                       const double* d_M, double* d_A_eff, int32_t* d_info, void* stream) {
  if (!p) return fail(HOMMX_EINVAL, "null plan");
  if (n_cells < 0) return fail(HOMMX_EINVAL, "negative n_cells");
  if (n_cells == 0) return HOMMX_OK;
  if (!d_mask || !d_values || !d_A_eff) return fail(HOMMX_EINVAL, "null mask / values / A_eff");
  if (n_cells > 0x7fffffffll) return fail(HOMMX_EINVAL, "n_cells too large for one launch");
  HIP_TRY(hipSetDevice(p->desc.device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p->family == FAM_FUSED2D) {
    hommx::CoefSource src;
    src.mode = hommx::COEF_TWO_PHASE;
    src.table = d_mask;
    HIP_TRY(hommx::launch_poisson2d_fused(d_values, d_M, d_A_eff, d_info, p->desc.n_micro, n_cells, st, src));
    return HOMMX_OK;
  }
  // blocked family: expand on the device, chunk by chunk of at most 1 GiB of element stream
  const int64_t per = p->n_el * p->n_comp;
  int64_t chunk = (int64_t)((1ll << 27) / (per > 0 ? per : 1));
  if (chunk < 1) chunk = 1;
  if (chunk > n_cells) chunk = n_cells;
  if (chunk > p->cap_expand) {
    if (p->d_expand) hipFree(p->d_expand);
    p->d_expand = nullptr;
    p->cap_expand = 0;
    HIP_TRY(hipMalloc(&p->d_expand, sizeof(double) * chunk * per));
    p->cap_expand = chunk;
  }
  const int d = p->desc.dim, t = p->t;
  for (int64_t c0 = 0; c0 < n_cells; c0 += chunk) {
    const int64_t nc = (n_cells - c0 < chunk) ? n_cells - c0 : chunk;
    HIP_TRY(hommx::launch_expand_two_phase(d_mask, d_values + c0 * 2 * p->n_comp, p->d_expand, p->n_el, p->n_comp, nc, st));
    int rc = hommx::blocked_solve(p->ws, nc, p->d_expand, d_M ? d_M + c0 * d * d : nullptr, d_A_eff + c0 * t * t,
                                  d_info ? d_info + c0 : nullptr, st);
    if (rc != 0) return fail(rc, "blocked path: %s", hommx::blocked_last_error());
  }
  return HOMMX_OK;
}

int hommx_solve_batch_two_phase(hommx_plan* p, int64_t n_cells, const uint8_t* mask, const double* values,
                                const double* M, double* A_eff, int32_t* info) {
  if (!p) return fail(HOMMX_EINVAL, "null plan");
  if (n_cells < 0) return fail(HOMMX_EINVAL, "negative n_cells");
  if (n_cells == 0) return HOMMX_OK;
  if (!mask || !values || !A_eff) return fail(HOMMX_EINVAL, "null mask / values / A_eff");
  HIP_TRY(hipSetDevice(p->desc.device));
  const int d = p->desc.dim, t = p->t;
  // The inputs are small (a mask + two values per cell): they are packed into ONE pinned block owned by the plan and travel in one
  // asynchronous copy; the outputs come back the same way, and the call synchronises once.  No hipMalloc / hipFree per call.
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t o_mask = 0, o_val = up((size_t)p->n_el), o_M = o_val + up(sizeof(double) * n_cells * 2 * p->n_comp);
  const size_t in_bytes = o_M + (M ? up(sizeof(double) * n_cells * d * d) : 0);
  const size_t o_info = up(sizeof(double) * n_cells * t * t), out_bytes = o_info + up(sizeof(int32_t) * n_cells);
  if (int rc = grow_pinned(p->pin_in, in_bytes)) return rc;
  if (int rc = grow_pinned(p->pin_out, out_bytes)) return rc;
  if (int rc = grow(p->dev_in, in_bytes)) return rc;
  if (int rc = grow(p->dev_out, out_bytes)) return rc;
  char* hin = static_cast<char*>(p->pin_in.p);
  char* din = static_cast<char*>(p->dev_in.p);
  char* dout = static_cast<char*>(p->dev_out.p);
  memcpy(hin + o_mask, mask, (size_t)p->n_el);
  memcpy(hin + o_val, values, sizeof(double) * n_cells * 2 * p->n_comp);
  if (M) memcpy(hin + o_M, M, sizeof(double) * n_cells * d * d);
  HIP_TRY(hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, nullptr));
  int rc = hommx_solve_batch_two_phase_device(p, n_cells, reinterpret_cast<const uint8_t*>(din + o_mask), reinterpret_cast<const double*>(din + o_val),
                                              M ? reinterpret_cast<const double*>(din + o_M) : nullptr, reinterpret_cast<double*>(dout),
                                              reinterpret_cast<int32_t*>(dout + o_info), nullptr);
  if (rc != HOMMX_OK) return rc;
  HIP_TRY(hipMemcpyAsync(p->pin_out.p, dout, out_bytes, hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  const char* hout = static_cast<const char*>(p->pin_out.p);
  memcpy(A_eff, hout, sizeof(double) * n_cells * t * t);
  if (info) memcpy(info, hout + o_info, sizeof(int32_t) * n_cells);
  return HOMMX_OK;
}

int hommx_solve_batch_separable_device(hommx_plan* p, int64_t n_cells, int32_t family, int32_t n_q, const double* d_table,
                                       const double* d_weights, const double* d_params, const double* d_M, double* d_A_eff,
                                       int32_t* d_info, void* stream) {
  if (!p) return fail(HOMMX_EINVAL, "null plan");
  if (n_cells < 0) return fail(HOMMX_EINVAL, "negative n_cells");
  if (n_cells == 0) return HOMMX_OK;
  if (p->desc.kind != HOMMX_KIND_POISSON_SCALAR && p->desc.kind != HOMMX_KIND_ELASTICITY_ISO)
    return fail(HOMMX_EINVAL, "separable samplers are defined for the scalar Poisson and the isotropic elasticity kinds");
  if (p->desc.kind == HOMMX_KIND_ELASTICITY_ISO && family != HOMMX_SAMPLER_AFFINE)
    return fail(HOMMX_EINVAL, "the isotropic elasticity kind takes the affine sampler only ((lambda, mu) = a + b g)");
  if (family != HOMMX_SAMPLER_AFFINE && family != HOMMX_SAMPLER_RECIPROCAL) return fail(HOMMX_EINVAL, "unknown sampler family %d", family);
  if (!d_table || !d_params || !d_A_eff) return fail(HOMMX_EINVAL, "null table / params / A_eff");
  if (family == HOMMX_SAMPLER_RECIPROCAL && (n_q < 1 || !d_weights)) return fail(HOMMX_EINVAL, "reciprocal sampler needs n_q >= 1 and weights");
  if (n_cells > 0x7fffffffll) return fail(HOMMX_EINVAL, "n_cells too large for one launch");
  HIP_TRY(hipSetDevice(p->desc.device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hommx::CoefSource src;
  src.mode = family == HOMMX_SAMPLER_AFFINE ? hommx::COEF_AFFINE : hommx::COEF_RECIPROCAL;
  src.nq = family == HOMMX_SAMPLER_AFFINE ? 1 : n_q;
  src.table = d_table;
  src.weights = d_weights;
  if (p->family == FAM_FUSED2D) {
    HIP_TRY(hommx::launch_poisson2d_fused(d_params, d_M, d_A_eff, d_info, p->desc.n_micro, n_cells, st, src));
    return HOMMX_OK;
  }
  // blocked family: expand on the device, chunk by chunk of at most 1 GiB of element stream
  const int64_t per = p->n_el * p->n_comp;
  int64_t chunk = (int64_t)((1ll << 27) / (per > 0 ? per : 1));
  if (chunk < 1) chunk = 1;
  if (chunk > n_cells) chunk = n_cells;
  if (chunk > p->cap_expand) {
    if (p->d_expand) hipFree(p->d_expand);
    p->d_expand = nullptr;
    p->cap_expand = 0;
    HIP_TRY(hipMalloc(&p->d_expand, sizeof(double) * chunk * per));
    p->cap_expand = chunk;
  }
  const int d = p->desc.dim, t = p->t;
  for (int64_t c0 = 0; c0 < n_cells; c0 += chunk) {
    const int64_t nc = (n_cells - c0 < chunk) ? n_cells - c0 : chunk;
    HIP_TRY(hommx::launch_expand_separable(src, d_params + 2 * p->n_comp * c0, p->d_expand, p->n_el, p->n_comp, nc, st));
    int rc = hommx::blocked_solve(p->ws, nc, p->d_expand, d_M ? d_M + c0 * d * d : nullptr, d_A_eff + c0 * t * t,
                                  d_info ? d_info + c0 : nullptr, st);
    if (rc != 0) return fail(rc, "blocked path: %s", hommx::blocked_last_error());
  }
  return HOMMX_OK;
}

int hommx_solve_batch_separable(hommx_plan* p, int64_t n_cells, int32_t family, int32_t n_q, const double* table,
                                const double* weights, const double* params, const double* M, double* A_eff, int32_t* info) {
  if (!p) return fail(HOMMX_EINVAL, "null plan");
  if (n_cells < 0) return fail(HOMMX_EINVAL, "negative n_cells");
  if (n_cells == 0) return HOMMX_OK;
  if (!table || !params || !A_eff) return fail(HOMMX_EINVAL, "null table / params / A_eff");
  if (family != HOMMX_SAMPLER_AFFINE && family != HOMMX_SAMPLER_RECIPROCAL) return fail(HOMMX_EINVAL, "unknown sampler family %d", family);
  if (family == HOMMX_SAMPLER_RECIPROCAL && (n_q < 1 || !weights)) return fail(HOMMX_EINVAL, "reciprocal sampler needs n_q >= 1 and weights");
  HIP_TRY(hipSetDevice(p->desc.device));
  const int d = p->desc.dim, t = p->t;
  const int64_t ntab = p->n_el * (family == HOMMX_SAMPLER_AFFINE ? 1 : n_q);
  const bool has_w = weights && n_q > 0;
  if (int rc = grow(p->st_table, sizeof(double) * ntab)) return rc;
  if (int rc = grow(p->st_values, sizeof(double) * n_cells * 2 * p->n_comp)) return rc;
  if (int rc = grow(p->st_out, sizeof(double) * n_cells * t * t)) return rc;
  if (int rc = grow(p->st_info, sizeof(int32_t) * n_cells)) return rc;
  if (has_w)
    if (int rc = grow(p->st_w, sizeof(double) * n_q)) return rc;
  if (M)
    if (int rc = grow(p->st_M, sizeof(double) * n_cells * d * d)) return rc;
  HIP_TRY(hipMemcpyAsync(p->st_table.p, table, sizeof(double) * ntab, hipMemcpyHostToDevice, nullptr));
  HIP_TRY(hipMemcpyAsync(p->st_values.p, params, sizeof(double) * n_cells * 2 * p->n_comp, hipMemcpyHostToDevice, nullptr));
  if (has_w) HIP_TRY(hipMemcpyAsync(p->st_w.p, weights, sizeof(double) * n_q, hipMemcpyHostToDevice, nullptr));
  if (M) HIP_TRY(hipMemcpyAsync(p->st_M.p, M, sizeof(double) * n_cells * d * d, hipMemcpyHostToDevice, nullptr));
  int rc = hommx_solve_batch_separable_device(p, n_cells, family, n_q, static_cast<const double*>(p->st_table.p),
                                              has_w ? static_cast<const double*>(p->st_w.p) : nullptr, static_cast<const double*>(p->st_values.p),
                                              M ? static_cast<const double*>(p->st_M.p) : nullptr, static_cast<double*>(p->st_out.p),
                                              static_cast<int32_t*>(p->st_info.p), nullptr);
  if (rc != HOMMX_OK) return rc;
  HIP_TRY(hipMemcpy(A_eff, p->st_out.p, sizeof(double) * n_cells * t * t, hipMemcpyDeviceToHost));
  if (info) HIP_TRY(hipMemcpy(info, p->st_info.p, sizeof(int32_t) * n_cells, hipMemcpyDeviceToHost));
  return HOMMX_OK;
}

int hommx_solve_batch_correctors(hommx_plan* p, int64_t n_cells, const double* coef, const double* M, double* A_eff,
                                 double* correctors, int32_t* info) {
  if (!p) return fail(HOMMX_EINVAL, "null plan");
  if (n_cells < 0) return fail(HOMMX_EINVAL, "negative n_cells");
  if (n_cells == 0) return HOMMX_OK;
  if (!coef || !A_eff || !correctors) return fail(HOMMX_EINVAL, "null coef / A_eff / correctors");
  HIP_TRY(hipSetDevice(p->desc.device));
  if (!p->ws) {
    int rc = hommx::blocked_workspace_create(&p->ws, p->desc.dim, p->desc.n_micro, p->desc.kind);
    if (rc != 0) return fail(rc, "blocked path: %s", hommx::blocked_last_error());
  }
  const int d = p->desc.dim, t = p->t;
  const int bs = (p->desc.kind >= HOMMX_KIND_ELASTICITY_ISO) ? d : 1;
  long long nn = 1;
  for (int k = 0; k < d; ++k) nn *= p->desc.n_micro;
  double *d_coef = nullptr, *d_M = nullptr, *d_out = nullptr, *d_corr = nullptr;
  int32_t* d_info = nullptr;
  auto cleanup = [&]() {
    if (d_coef) hipFree(d_coef);
    if (d_M) hipFree(d_M);
    if (d_out) hipFree(d_out);
    if (d_corr) hipFree(d_corr);
    if (d_info) hipFree(d_info);
  };
#define HIP_TRY_C(expr)                                                                             \
  do {                                                                                              \
    hipError_t e__ = (expr);                                                                        \
    if (e__ != hipSuccess) {                                                                        \
      cleanup();                                                                                    \
      return fail(e__ == hipErrorOutOfMemory ? HOMMX_ENOMEM : HOMMX_EHIP, "%s failed: %s", #expr,   \
                  hipGetErrorString(e__));                                                          \
    }                                                                                               \
  } while (0)
  const size_t ncoef = sizeof(double) * n_cells * p->n_el * p->n_comp, ncorr = sizeof(double) * n_cells * t * nn * bs;
  HIP_TRY_C(hipMalloc(&d_coef, ncoef));
  HIP_TRY_C(hipMalloc(&d_out, sizeof(double) * n_cells * t * t));
  HIP_TRY_C(hipMalloc(&d_corr, ncorr));
  HIP_TRY_C(hipMalloc(&d_info, sizeof(int32_t) * n_cells));
  HIP_TRY_C(hipMemcpy(d_coef, coef, ncoef, hipMemcpyHostToDevice));
  if (M) {
    HIP_TRY_C(hipMalloc(&d_M, sizeof(double) * n_cells * d * d));
    HIP_TRY_C(hipMemcpy(d_M, M, sizeof(double) * n_cells * d * d, hipMemcpyHostToDevice));
  }
  int rc = hommx::blocked_solve(p->ws, n_cells, d_coef, d_M, d_out, d_info, nullptr, d_corr);
  if (rc != 0) {
    cleanup();
    return fail(rc, "blocked path: %s", hommx::blocked_last_error());
  }
  HIP_TRY_C(hipDeviceSynchronize());
  HIP_TRY_C(hipMemcpy(A_eff, d_out, sizeof(double) * n_cells * t * t, hipMemcpyDeviceToHost));
  HIP_TRY_C(hipMemcpy(correctors, d_corr, ncorr, hipMemcpyDeviceToHost));
  if (info) HIP_TRY_C(hipMemcpy(info, d_info, sizeof(int32_t) * n_cells, hipMemcpyDeviceToHost));
#undef HIP_TRY_C
  cleanup();
  return HOMMX_OK;
}

int hommx_calibrate_fp64(int device, double* mfma_flops_per_s, double* fma_flops_per_s) {
  if (!mfma_flops_per_s && !fma_flops_per_s) return fail(HOMMX_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hommx::run_fp64_calibration(mfma_flops_per_s, fma_flops_per_s));
  return HOMMX_OK;
}

int hommx_calibrate_fp64_mfma(int device, double* flops_per_s) { return hommx_calibrate_fp64(device, flops_per_s, nullptr); }
int hommx_calibrate_fp64_detail(int device, double* mfma_flops_per_s, double* fma_flops_per_s, double* mfma_lds_fed_flops_per_s) {
  if (!mfma_flops_per_s && !fma_flops_per_s && !mfma_lds_fed_flops_per_s) return fail(HOMMX_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hommx::run_fp64_calibration(mfma_flops_per_s, fma_flops_per_s, mfma_lds_fed_flops_per_s));
  return HOMMX_OK;
}

}  // extern "C"
