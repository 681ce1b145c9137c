/*
 * hommx_hip.h -- C ABI of libhommx_hip.so: the MI355X (gfx950) batched micro-cell solver that
 * replaces the per-macro-cell loop of flxrcz/hommx.
 *
 * The reference has no FFI layer; its seam is Python:
 *
 *   BaseHMM._assemble_stiffness()            src/hommx/hmm.py:298-332   loop over owned macro cells
 *     -> _compute_local_stiffness(cell)      src/hommx/hmm.py:334-369   nb corrector solves + nb^2 energies
 *          -> PeriodicLinearProblem.solve()  src/hommx/cell_problem.py:363-388
 *
 * This library replaces the LOOP: one call computes the effective tensor A_H / C_H of every macro
 * cell of a batch; the Python host (hommx_amd/hmm.py) turns it into S_loc = vol(T) G A_H G^T and
 * scatters on the CPU exactly where the reference calls MatSetValues (hmm.py:325-330).
 *
 * All arrays are C-contiguous, float64 / int32.  Plain pointers and sizes only; no torch types.
 * Every function returns 0 on success or a negative HOMMX_E* code; hommx_last_error() returns a
 * thread-local message.  Numerical failures are reported per cell in info[] (the reference logs
 * and continues: hmm.py:320-323, 427-430), never by the return code.
 */
#ifndef HOMMX_HIP_H
#define HOMMX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOMMX_OK 0
#define HOMMX_EINVAL (-1)   /* bad argument / unsupported configuration */
#define HOMMX_EHIP (-2)     /* HIP runtime error (message in hommx_last_error) */
#define HOMMX_ENODEV (-3)   /* no usable GPU */
#define HOMMX_ENOMEM (-4)   /* device allocation failed */
#define HOMMX_ERCCL (-5)    /* RCCL error (message in hommx_last_error) */

/* problem kinds: which bilinear form of hmm.py the plan assembles */
#define HOMMX_KIND_POISSON_SCALAR 0     /* hmm.py:644-667 (and :759-789 when M != NULL); coef[cell][el]            */
#define HOMMX_KIND_POISSON_MATRIX 1     /* same forms, matrix-valued A; coef[cell][el][d(d+1)/2] (00,11,[22,]01[,02,12]) */
#define HOMMX_KIND_ELASTICITY_ISO 2     /* hmm.py:887-922 (and :1024-1067 when M != NULL); coef[cell][el][2]=(lambda,mu) */
#define HOMMX_KIND_ELASTICITY_VOIGT 3   /* same forms, full Hooke tensor; coef[cell][el][t(t+1)/2], upper triangle of
                                           the t x t matrix  E^m : A : E^n  (tensorial unit strains), row-major */

#define HOMMX_FLAG_FORCE_BLOCKED 1     /* use the generic blocked kernel family even where the fused 2D kernel applies */

/* Environment knobs (development / tuning; read once, when a plan is created):
 *   HOMMX_BLOCKED_MEM_GB   workspace budget of the blocked family in GB (default: plane elimination min(64, half of the free HBM),
 *                          nested dissection min(64 -- 128 through hommx_plan_reserve --, 0.6 x free HBM): its fronts are 0.2 GB per C4 / C5 cell)
 *   HOMMX_GEMM128_MIN      smallest M, N routed to the 128x128-tile GEMM (default 256 on the plane elimination; nested-dissection plans
 *                          use the 64x64 tiles throughout)
 *   HOMMX_TILE_SB          big lower-triangle updates walk their tiles in SB x SB super-blocks (default 4; 0: row by row)
 *   HOMMX_SPARSE_V1        any value: generic instead of strip-form sparse E products
 *   HOMMX_LEAF32           any value: 32x32 leaves only in the recursive block inverse
 *   HOMMX_NO_SPLIT64       any value: the recursive block inverse halves 192 into 96 + 96 instead of 64 + 128
 *   HOMMX_NO_H2D_OVERLAP   any value: hommx_solve_batch copies the whole coefficient stream before the first kernel
 *   HOMMX_NO_SMALL_FUSED   any value: plane blocks b <= 64 take the HBM-resident kernels instead of the one-launch kernels
 *   HOMMX_MF_MIN_B         smallest plane block b routed to the nested-dissection (multifrontal) elimination instead of the plane
 *                          elimination (default: 65 in 3D, 49 in 2D, i.e. every plane block the one-launch kernels do not take or lose
 *                          on; 0: never; a value below 65 from the environment needs HOMMX_NO_SMALL_FUSED as well)
 *   HOMMX_MF_FRONT         most 16 x 16 tiles per dimension of a front that is eliminated by the register-resident front kernel
 *                          (csrc/mf_front_kernel.h: one launch per tree level) instead of the build / inverse / GEMM launch sequence
 *                          (default and maximum 19; 0: never)
 *   HOMMX_MF_STREAMS       1: the nested-dissection route runs on the caller's stream alone (default 4: every chunk as two to four pieces side
 *                          by side on the caller's stream and plan-owned ones; the caller's stream waits for all, results are bitwise equal)
 *   HOMMX_MF_CORR          0: hommx_solve_batch_correctors of a nested-dissection plan runs the plane elimination (default: back substitution
 *                          down the elimination tree on a second plan whose fronts all stay resident)
 *   HOMMX_MF_LEAF, HOMMX_MF_SPLIT_DEPTH, HOMMX_MF_G128_MIN_K, HOMMX_MF_NO_BORDER_SPLIT, HOMMX_MF_VERBOSE   tuning / A-B knobs of that route
 *   HOMMX_SMALL_WAVES      2 / 4: plane blocks b <= 48 take the LDS-resident multi-wave kernel (that many waves per macro cell)
 *                          instead of the one-wave-per-cell register kernel; for 48 < b <= 64 it sets that kernel's wave count
 *                          (2 / 4 / 8, default 8)                                                                              */

typedef struct hommx_plan hommx_plan;

typedef struct hommx_plan_desc {
  int32_t dim;      /* 2 or 3 (hmm.py:104-105)                                                    */
  int32_t n_micro;  /* micro cells per side of the unit-cell mesh create_unit_square/cube(n,n[,n]) */
  int32_t kind;     /* HOMMX_KIND_*                                                                */
  int32_t device;   /* HIP device ordinal                                                          */
  int32_t flags;    /* HOMMX_FLAG_* bits, normally 0                                               */
  int32_t reserved[3];
} hommx_plan_desc;

/* Number of visible HIP devices (0 if none / runtime unusable). */
int hommx_device_count(void);

/* Create / destroy a plan = everything that does not depend on the batch: kernel selection,
 * stencil tables, device scratch.  Replaces the per-solve object construction of hmm.py:420-425
 * (dolfinx_mpc.LinearProblem.__init__: sparsity pattern + Mat + Vec + KSP for EVERY rhs of EVERY cell). */
int hommx_plan_create(hommx_plan** out, const hommx_plan_desc* desc);
int hommx_plan_destroy(hommx_plan* plan);

/* Optional: allocate the plan's device workspace for batches of up to n_cells now instead of in the first solve.  The nested-dissection
 * route of the C4 / C5 problem size holds ~0.2 GB of fronts per cell of a chunk: a first solve allocates up to min(64 GB, 0.6 x free HBM),
 * this call up to 128 GB (2 % more throughput over many batches) -- 2 to 6 s of hipMalloc that a benchmark wants outside its timed
 * region.  The fused 2D family keeps no workspace: a no-op there. */
int hommx_plan_reserve(hommx_plan* plan, int64_t n_cells);

/* Shape queries: elements per micro mesh (2 n^2 / 6 n^3), coefficient doubles per element,
 * t = size of the effective tensor (d for Poisson, d(d+1)/2 for elasticity), the descriptor fields, and the name of the
 * kernel route the plan's effective-tensor solves take: "fused2d" (2D scalar Poisson, n <= 32), "small_wave" (plane block
 * b <= 48: one wavefront per cell), "small_fused" (3D meshes with 48 < b <= 64: LDS), "multifrontal" (plane blocks b > 64 and 2D meshes with b > 48, e.g. 3D elasticity
 * from 5^3 micro cells: nested dissection, batched fronts) or "blocked" (everything else: plane elimination; also the corrector entry point of plans whose tensors take a one-launch route). */
int32_t hommx_plan_dim(const hommx_plan* plan);
int32_t hommx_plan_device(const hommx_plan* plan);
int32_t hommx_plan_n_micro(const hommx_plan* plan);
int32_t hommx_plan_kind(const hommx_plan* plan);
int64_t hommx_plan_num_elements(const hommx_plan* plan);
int32_t hommx_plan_coef_components(const hommx_plan* plan);
int32_t hommx_plan_tensor_size(const hommx_plan* plan);
const char* hommx_plan_kernel_name(const hommx_plan* plan);
/* One line describing what that route launches for THIS plan (kernel names with their tile sizes, tree shape of the nested dissection,
 * stage size, streams): for reports -- bench.py's roofline.kernel label is this string, so it cannot drift from the code. */
const char* hommx_plan_route_detail(hommx_plan* plan);
/* Dense flops ONE micro-cell solve executes on the plan's route, by the route's own model: block-cyclic plane
 * elimination (6 (n-1) + 2) b^3; multifrontal: sum over the fronts of s^3 + 2 s^2 r + s r^2 on the padded front sizes. */
double hommx_plan_flops_per_solve(const hommx_plan* plan);

/*
 * Solve a batch of macro cells (host pointers; the call copies in, runs, copies out, synchronises).
 *
 *   coef   [n_cells][n_el][n_comp]  element means of A(c_T, y) (hmm.py:190-198, 349-352) in the element
 *                                   order  el = n_sub*(i + n*j [+ n*n*k]) + s  of the DOLFINx-style mesh
 *   M      [n_cells][d][d] or NULL  Dtheta_transpose(c_T), M[i][j] = d theta_j / d x_i (hmm.py:741, 756-757)
 *   A_eff  [n_cells][t][t]          out: effective tensor = vol(Y)^-1 * the functional of hmm.py:652-667 /
 *                                   774-789 / 905-922 / 1050-1067 on the canonical unit gradients / strains
 *   info   [n_cells] or NULL        out: 0 ok; k>0 non-positive or NaN pivot first seen in block step k-1 of the plane elimination
 *                                   (fused2d / small_* / blocked routes), or in the k-th group of fronts, counted from the leaves, of
 *                                   the multifrontal route; the cell's tensor is then not meaningful (the reference logs and goes on)
 */
int hommx_solve_batch(hommx_plan* plan, int64_t n_cells, const double* coef, const double* M,
                      double* A_eff, int32_t* info);

/* Same with DEVICE pointers, asynchronous on `stream` (a hipStream_t, NULL = default stream).
 * No caller pointer is retained after return; the caller synchronises the stream before reading A_eff.  The blocked and
 * small-block kernel families work out of scratch the PLAN owns (workspace, expanded coefficient stream): a plan is not
 * thread-safe, and two calls on the same plan must not be in flight on different streams at once (distinct plans are
 * independent; the fused 2D family keeps no scratch). */
int hommx_solve_batch_device(hommx_plan* plan, int64_t n_cells, const double* d_coef, const double* d_M,
                             double* d_A_eff, int32_t* d_info, void* stream);

/*
 * Two-phase media sampled on the device ("on-device coefficient samplers").  Every BASELINE.json
 * configuration has a coefficient of the form  A(x, y) = indicator(y) ? a_1(x) : a_0(x)  (laminate.py:101-102,
 * inclusion.py:107-118, rotated_fibers.py:23-38): the fast variable only selects a phase.  Instead of streaming
 * n_el samples per macro cell the caller passes the phase mask ONCE and two values per cell:
 *
 *   mask    [n_el] uint8            phase of every micro element (element order of the DOLFINx-style mesh)
 *   values  [n_cells][2][n_comp]    coefficient of phase 0 / phase 1 at the macro cell midpoint c_T
 *
 * The fused 2D kernel selects in registers (no coefficient stream at all: 16 bytes per cell instead of 16 KiB); the
 * blocked family expands into its scratch stream on the device.  Host pointers; same outputs as hommx_solve_batch.
 */
int hommx_solve_batch_two_phase(hommx_plan* plan, int64_t n_cells, const uint8_t* mask, const double* values,
                                const double* M, double* A_eff, int32_t* info);

/* Same with DEVICE pointers, asynchronous on `stream`. */
int hommx_solve_batch_two_phase_device(hommx_plan* plan, int64_t n_cells, const uint8_t* d_mask, const double* d_values,
                                       const double* d_M, double* d_A_eff, int32_t* d_info, void* stream);

/*
 * Separable coefficients sampled on the device.  The smooth coefficients of the reference's own tests are a
 * slow amplitude times a fixed function of the fast variable:  A(x, y) = a(x) + b(x) g(y)  (test_integration_poisson.py:149-150,
 * 197: 0.33 + 0.15 (sin 2 pi x0 + sin 2 pi y0); :268: 1.1 + x0 + sin 2 pi y0)  or its reciprocal  1 / (a(x) + b(x) g(y))
 * (:124-125: 1 / (2 + cos 2 pi y0)).  g is tabulated ONCE on the micro mesh at the points of the quadrature rule UFL would pick
 * (degree 3: 6 points per triangle); the kernels form the element means from (a, b) of each macro cell, so 16 bytes per cell
 * cross the boundary instead of n_el samples.  Scalar Poisson kind; isotropic elasticity kind with one (a, b) pair per Lame
 * parameter and the same g (test_integration_linear_elasticity.py:78-93: lambda = 1.25, mu = 5 + 4.5 sin 2 pi y0).
 *
 *   family   HOMMX_SAMPLER_AFFINE      A_K = a + b * table[K]                      table[n_el] = sum_q w_q g(y_{K,q})
 *            HOMMX_SAMPLER_RECIPROCAL  A_K = sum_q weights[q] / (a + b * table[K][q])   table[n_el][n_q] = g(y_{K,q})
 *   params   [n_cells][n_comp][2] = (a, b) of every coefficient component at the macro cell midpoint c_T (n_comp = 1, or 2 = (lambda, mu))
 *
 * Every operation is a separately rounded IEEE-754 operation in the written order (q ascending), so a host evaluating the same
 * formula reproduces the element stream bit for bit (hommx_amd.hmm.Separable.host_stream; tests/test_gpu_separable.py).
 */
#define HOMMX_SAMPLER_AFFINE 0
#define HOMMX_SAMPLER_RECIPROCAL 1
int hommx_solve_batch_separable(hommx_plan* plan, int64_t n_cells, int32_t family, int32_t n_q, const double* table,
                                const double* weights, const double* params, const double* M, double* A_eff, int32_t* info);

/* Same with DEVICE pointers, asynchronous on `stream`. */
int hommx_solve_batch_separable_device(hommx_plan* plan, int64_t n_cells, int32_t family, int32_t n_q, const double* d_table,
                                       const double* d_weights, const double* d_params, const double* d_M, double* d_A_eff,
                                       int32_t* d_info, void* stream);

/* Same as hommx_solve_batch, additionally returning the correctors (host pointers):
 *   correctors [n_cells][t][n^d * bs]  chi_m of the canonical load case m (unit gradient e_m / unit strain E^m) at the
 *                                      periodic unknowns, dof = node * bs + component, node = i + n j [+ n^2 k];
 *                                      mean-free per component (the reference removes the constants: cell_problem.py:349-361).
 * These are the functions the reference keeps in self._correctors (hmm.py:204-207, 431) / PoissonPeriodicHMM.correctors
 * (hmm.py:1211-1213, 1239-1240), for the canonical loads instead of the nb macro basis functions: the corrector of a
 * macro basis function is the linear combination  eps * sum_m (grad phi_i)_m chi_m  (SURVEY A.2, row A5).
 * Route: plans whose tensors take the nested-dissection route get the correctors by back substitution down the same elimination tree
 * (a second plan of that tree whose fronts all stay resident; HOMMX_MF_CORR=0: plane elimination); every other plan runs the plane
 * elimination of the blocked family here (the one-launch kernels and the fused 2D kernel never form the factors). */
int hommx_solve_batch_correctors(hommx_plan* plan, int64_t n_cells, const double* coef, const double* M,
                                 double* A_eff, double* correctors, int32_t* info);

/*
 * Multi-GPU from ONE process: the macro cells are block-partitioned over the devices of a communicator --
 * the reference's MPI partition of the cell loop (hmm.py:307-310) -- and ONE RCCL all-gather over xGMI returns the whole
 * effective-tensor field (with the per-cell info flags riding in the same buffer) to every device, where the reference merges
 * rank contributions in PETSc's assembly stash (hmm.py:325-330, :442).  RCCL is loaded lazily (dlopen) on the first call.
 * Python callers that run one process per GPU use hommx_amd/dist.py instead.
 *
 *   hommx_comm_init_all     communicator over devs[0..ndev) (NULL: devices 0..ndev-1); one stream per device
 *   hommx_allgather_field   in-place all-gather: d_field_per_dev[i] is a device buffer of ndev * count doubles on device i whose
 *                           own shard already sits at offset i * count; returns after every device holds all shards
 *   hommx_solve_batch_multi plans[i] = a plan created on device i of the communicator (same dim / n_micro / kind); host
 *                           pointers as hommx_solve_batch.  Device i receives, solves and packs ONLY cells
 *                           [i * ceil(n/ndev), (i+1) * ceil(n/ndev)); the field is all-gathered and copied out once.
 */
typedef struct hommx_comm hommx_comm;
int hommx_comm_init_all(hommx_comm** out, int ndev, const int* devs);
int hommx_comm_destroy(hommx_comm* comm);
int hommx_comm_size(const hommx_comm* comm);
int hommx_allgather_field(hommx_comm* comm, double* const* d_field_per_dev, int64_t count_per_dev);
int hommx_solve_batch_multi(hommx_comm* comm, hommx_plan* const* plans, int64_t n_cells, const double* coef, const double* M,
                            double* A_eff, int32_t* info);

/* DEVICE-pointer form: the inputs stay resident on their devices (no H2D per call).  d_coef_per_dev[i] / d_M_per_dev[i] point at
 * the shard of device i (cells [begin_i, end_i) of hommx_shard_range, its own first cell at offset 0) on device i;
 * d_packed_per_dev[i] is a buffer of ndev * per_dev * (t*t + 1) doubles on device i that receives the whole gathered field, one row
 * [A_eff (t*t) | info as a double] per cell slot, shard after (padded) shard -- hommx_unpack_field() reads that layout on the host.
 * Returns after the all-gather has completed on every device. */
int hommx_solve_batch_multi_device(hommx_comm* comm, hommx_plan* const* plans, int64_t n_cells, const double* const* d_coef_per_dev,
                                   const double* const* d_M_per_dev, double* const* d_packed_per_dev);

/* The block partition and the layout of the gathered field as plain host arithmetic (no GPU needed): cells of device i are
 * [begin, end) = [i * per_dev, min(n_cells, (i+1) * per_dev)), per_dev = ceil(n_cells / ndev)  (the reference's
 * MPI ownership ranges, hmm.py:307-310). */
int hommx_shard_range(int64_t n_cells, int32_t ndev, int32_t i, int64_t* begin, int64_t* end, int64_t* per_dev);
int hommx_unpack_field(int64_t n_cells, int32_t ndev, int32_t tt, const double* packed, double* A_eff, int32_t* info);

/* Calibration micro-benchmark, in FLOP/s: the best sustained rate of v_mfma_f64_16x16x4_f64 (8 independent accumulator tiles per
 * wave) and of v_fma_f64 (16 independent accumulators per lane) over 2 and 4 waves per SIMD on all CUs.  Both instructions share one
 * fp64 pipe on MI355X; bench.py reports the larger as the measured fp64 peak next to the datasheet figure.
 * hommx_calibrate_fp64_mfma is the MFMA figure alone (kept for callers of the first ABI revision). */
int hommx_calibrate_fp64(int device, double* mfma_flops_per_s, double* fma_flops_per_s);
int hommx_calibrate_fp64_mfma(int device, double* flops_per_s);
/* Same plus a third figure: the MFMA fed from LDS the way the GEMM kernels feed it (16 MFMAs per 16 ds_read_b64).  The dependency-free
 * pure-MFMA loop clocks down under its own load; this loop is the fp64 matrix rate a real kernel can hold.  Any pointer may be NULL. */
int hommx_calibrate_fp64_detail(int device, double* mfma_flops_per_s, double* fma_flops_per_s, double* mfma_lds_fed_flops_per_s);

const char* hommx_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HOMMX_HIP_H */
