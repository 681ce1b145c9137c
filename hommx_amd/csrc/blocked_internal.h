// blocked_internal.h -- shared between blocked.hip (block-cyclic plane elimination) and multifrontal.hip (nested dissection):
// the workspace object behind a plan of the blocked family and the batched fp64-MFMA building blocks (GEMM tiles, recursive
// block inverse) both eliminations are made of.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>

#include "geo.h"

namespace hommx {

struct MfPlan;  // multifrontal.hip

struct BlockedWorkspace {
  Geo G;
  // development knobs, read ONCE when the plan is created (include/hommx_hip.h lists them)
  double budget_gb_env = 0.0;  // HOMMX_BLOCKED_MEM_GB (0: automatic)
  int gemm128_min = 256;       // HOMMX_GEMM128_MIN
  bool sparse_v1 = false;      // HOMMX_SPARSE_V1: generic instead of strip-form sparse E products
  bool leaf32 = false;         // HOMMX_LEAF32: 32x32 leaves only in the recursive inverse
  bool split64 = true;         // HOMMX_NO_SPLIT64: halve 192 into 96 + 96 (32- and 64-leaves) instead of 64 + 128
  bool small_fused = true;     // HOMMX_NO_SMALL_FUSED switches the LDS-resident kernel for b <= 64 off (A/B runs)
  int small_waves = 0;         // HOMMX_SMALL_WAVES: 2 / 4 = the LDS kernel with that many waves per macro cell; 0 = default routes
  long long chunk = 0;
  double *Kst = nullptr, *Brhs = nullptr, *C0 = nullptr;
  double *S = nullptr, *W = nullptr, *Sl = nullptr, *V = nullptr, *X = nullptr, *T = nullptr;
  double *R = nullptr, *Rl = nullptr, *Vr = nullptr, *Gm = nullptr;
  // corrector mode: per eliminated plane the inverse Schur block, the arrow block and the load rows are kept
  long long hchunk = 0;
  double *hS = nullptr, *hW = nullptr, *hR = nullptr, *Xa = nullptr, *Xb = nullptr, *Y = nullptr;
  // nested-dissection route (3D, large plane blocks): symbolic analysis + arena, owned by multifrontal.hip
  MfPlan* mf = nullptr;
  MfPlan* mf_keep = nullptr;   // corrector plan of the same route (fronts keep their factors), created by the first corrector call
  bool mf_corr = true;         // HOMMX_MF_CORR=0: correctors of multifrontal plans take the plane elimination (A/B runs)
  int mf_min_b = 192;          // smallest plane block b routed to the multifrontal elimination (set per dim / unknowns per node when the
                               // workspace is created; HOMMX_MF_MIN_B overrides, 0: never)
  bool mf_no_border_split = false;  // HOMMX_MF_NO_BORDER_SPLIT (A/B runs)
  int mf_gather128_min_k = 1024;  // HOMMX_MF_G128_MIN_K: gathering Schur updates of smaller rank take the 64 x 64 tiles
  // tile orders of big lower-triangle updates (gemm): device tables, one per tile count, made on first use
  int tile_sb = 4;                // HOMMX_TILE_SB: tiles walk the lower triangle in SB x SB super-blocks (0: row by row)
  std::map<int, int*> tilemaps;
  std::string detail;             // hommx_plan_route_detail: written once, on first request
};


// One batched operation context: `nc` matrices (cells, or cells x fronts of one shape) on stream `st`.
struct Ctx {
  BlockedWorkspace* ws;
  long long nc;
  hipStream_t st;
  int32_t* info;          // per CELL failure flags (nullable)
  int stepcode;           // value a failing pivot check writes into info
  int ld = 0;             // leading dimension of the matrices invert() works on (0: G.Bp)
  long long sS = 0;       // their batch stride (0: Bp * Bp)
  long long sT = 0;       // batch stride of invert()'s scratch (0: sS)
  int infoDiv = 1;        // info index = batch index / infoDiv (fronts per cell in the multifrontal route)
};

// Child slot of a multifrontal front (multifrontal.hip), as the kernels see it
struct MfChild {
  long long offF;           // per-cell arena offset (doubles) of the child's GROUP buffer (x chunk size at launch)
  int nf, fidx, L, sp;      // fronts in the child's group, the child's index in it, its leading dimension, its padded s
  int valid, rb;            // rb: first border row of the child's boundary block
};

// "C is virtual": instead of beta * C the GEMM epilogue adds, for every child slot, the child's update matrix entry the unknown pair maps
// to (the extend-add of the multifrontal method fused into the parent's Schur update).  batch b = cell * nf + front.
struct GatherC {
  const double* arena = nullptr;
  long long nc = 0;             // cells in the chunk (arena offsets are per cell)
  const MfChild* child = nullptr;   // [nf][2]
  const int32_t* dpos = nullptr;    // [nf][2][rp] unknown of the child's boundary block a boundary unknown of this front maps to, -1: none
  int nf = 0, rp = 0;
  long long batch0 = 0;         // batch index of matrix 0 of this launch (a huge batch is launched in pieces)
  int rowOff = 0;               // C row 0 is boundary unknown rowOff of the front (the border rows are updated by a launch of their own)
};

// C = alpha op(A) op(B) + beta C for every matrix of the batch (blocked.hip: k_gemm_tile, XCD-aware tiles); lowerOnly: tiles on and
// below the diagonal only; Ct: mirrored copy of the result (may be C itself with lowerOnly)
void gemm(const Ctx& c, bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long sA, const double* B,
          int ldb, long long sB, double beta, double* C, int ldc, long long sC, int lowerOnly = 0, double* Ct = nullptr,
          const GatherC* gather = nullptr);

// tile edge (64 or 128) gemm() uses for an M x N x K product of this workspace, and the super-block tile order of a lower triangle of `ty`
// tile rows (device table cached in the workspace; nullptr: row-by-row order)
int gemm_tile_size(const BlockedWorkspace* ws, int M, int N, int K, bool gather);
const int* ensure_tilemap(BlockedWorkspace* ws, int ty);

// in-place inverse of the SPD diagonal block [off, off + size) of every matrix of the batch (recursive Schur-complement form;
// size a multiple of 32); `tmp`: scratch of at least size^2 / 2 doubles per matrix, batch stride c.sT
void invert(const Ctx& c, double* S, int off, int size, double* tmp);

extern thread_local std::string g_berr;

// multifrontal.hip
int mf_plan_create(MfPlan** out, const Geo& G, bool keep = false);
void mf_plan_destroy(MfPlan* p);
double mf_flops_per_cell(const MfPlan* p);
std::string mf_describe(const BlockedWorkspace* ws, const MfPlan* p);  // one line: tree, stages, streams, tile sizes of the dense kernels
int mf_reserve(BlockedWorkspace* ws, MfPlan* P, long long ncells, bool ahead);
// effective tensors, and with the corrector plan (keep = true) and d_corr != nullptr the correctors [cell][t][n^d bs] as well
int mf_solve(BlockedWorkspace* ws, MfPlan* P, long long ncells, const double* d_coef, const double* d_M, double* d_out, int32_t* d_info,
             hipStream_t st, double* d_corr = nullptr);
// remove the mean of every component of nc x t corrector fields (blocked.hip)
void launch_center_corr(BlockedWorkspace* ws, double* corr, long long nc, hipStream_t st);
// K1 of the blocked family (stencil rows, loads, C0 of `nc` cells into the given buffers), shared by both eliminations: each route owns
// its buffers (a plan may serve effective tensors on one route and correctors on the other, with different chunk sizes)
void launch_assembly(BlockedWorkspace* ws, const double* coef, const double* Mm, long long nc, hipStream_t st, double* Kst, double* Brhs,
                     double* C0);

}  // namespace hommx
