// mf_front.h -- k_mf_front: ONE launch per group of fronts of the nested-dissection route (multifrontal.hip) for fronts small enough to
// live in the REGISTERS of one workgroup: a front of L = s + r unknowns (16-granular) is held as its UPPER 16 x 16 tiles in the layout of
// the f64 MFMA accumulator, spread round-robin over the NW waves of the workgroup (T (T + 1) / 2 tiles of 4 doubles per lane; eight waves x 24
// tiles = 192 tiles, T <= 19; the 3D-elasticity leaf front (s, r) = (81, 222 + 8), T = 21, keeps its first two tile rows in LDS).  The kernel
//
//   1. BUILDS the front in registers: stencil entries (K1 output) + the children's update matrices through the child -> parent maps
//      (the extend-add), + the canonical loads in the 8 border rows -- nothing of F11 / F21 / F12 ever touches HBM;
//   2. ELIMINATES the s unknowns in panels of 16: the panel's tile row goes to LDS (tile (p, b) in the accumulator layout IS k-slab form:
//      register r = pivots 4 r .. 4 r + 3 against the columns of block b, i.e. E_b^T), every wave inverts the 16 x 16 diagonal tile with the
//      in-register exchange sweep of sweep_acc.h (T = -D, so the sweep returns -N), the waves share  Y'_a = -N E_a^T  through LDS and
//      every trailing tile takes  acc(a, b) += Y'_a^T E_b^T  -- four MFMAs with both operands read as conflict-free k-major tiles;
//   3. WRITES only the update matrix (lower triangle, the layout the parent's kernels read).
//
// It replaces k_mf_build + k_mf_pad + the launches of the recursive block inverse + X = N E^T + the gathering Schur update of such a group
// (5 to 11 launches), their round trips of the front through HBM, and the padding of s to 32 and of the lower triangle to 64 x 64 tiles.
// Reference semantics: one elimination step of the periodic micro problem's direct solve (cell_problem.py:363-388; forms hmm.py:644-667 /
// 759-789 / 887-922 / 1024-1067); parity is checked against the oracle and against the launch sequence it replaces.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blocked_internal.h"

namespace hommx {

constexpr int MFF_BORDER = 8;  // load rows per front (= MF_BORDER of multifrontal.hip)

struct MfFrontDev {
  int ns, nloc, sp, rb, L, nf;   // eliminated nodes, nodes of the front, ARENA layout of the group: padded s (multiple of 32), boundary unknowns, ld
  int s16, T, P, ntiles;         // 16-granular: padded s, tiles per dimension, panels (= s16 / 16), upper tiles T (T + 1) / 2
  int has_children;
  int R0;                        // tile rows [0, R0) of the front live in LDS (k_mf_front's LROWS variants), the rest in registers; tilemap / ntiles: rows >= R0
  long long offF;                // per-cell arena offset of the group's fronts
  const int32_t* nodes;          // [nf][nloc]       global (periodic) node of a local node
  const int8_t* code;            // [nf][nloc][ns]   stencil code of (row node, eliminated column node), -1: none
  const int32_t* upos;           // [nf][2][16 T]    unknown u of the front -> unknown of child slot c's update matrix (border rows included), -1: none
  const MfChild* child;          // [nf][2]
  const uint16_t* tilemap;       // [ntiles]         a << 8 | b of upper tile e (a <= b), row-major
};

// most 16-tiles per dimension a front may have to take the kernel (register budget: 8 waves x 24 tiles = 192 of the upper tiles)
constexpr int MFF_REG_TILES = 192;
constexpr int MFF_MAX_T = 21;   // 19 in registers alone; 20 / 21 with one / two tile rows in LDS

// one launch over `nbatch` = cells x fronts of the group (mf_front_kernel.h); stepcode: value written to info[cell] by a failing pivot check
void launch_mf_front(const MfFrontDev& g, int bs, const double* Kst, const double* Brhs, double* arena, long long nc, long long nbatch, int nn,
                     int ncode, int t, int32_t* info, int stepcode, hipStream_t st);

}  // namespace hommx
