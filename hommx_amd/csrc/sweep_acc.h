// sweep_acc.h -- in-register Gauss-Jordan exchange sweep of an NB x NB matrix held by ONE wavefront in the layout of the
// f64 MFMA accumulator ("acc layout"), NB = 16 or 32:
//
//     lane l = 16 k + j  (k = l >> 4 "lane row", j = l & 15),   a[ti][tj][r] = T[16 ti + 4 r + k][16 tj + j]
//
// Why this layout: (i) register r of tile row ti IS the k-slab (4 ti + r) of the matrix as an MFMA B operand, and -- for a
// symmetric matrix -- as an A operand, so the inverse feeds the products of the elimination without leaving the register
// file; (ii) the column multiplier of a rank-1 update, T[row][K] for the lane's own rows, sits in the SAME register in the same
// 16-lane row, at lane column K % 16: the DP-ALU DPP control `row_newbcast` delivers it inside the FMA itself
// (v_fmac_f64_dpp, full rate: tools/probe_dpp64.hip), so a pivot costs 16 FMAs + the pivot-row broadcast, no scaling of a
// pivot column and no LDS read per row.
//
// Exchange convention ("conv 3"): for pivot K with d = T[K][K], u = row K:
//     row K <- -u/d,   column K <- column / d,   T[K][K] <- 1/d,   rest += column (x) (-u/d)
// After all NB pivots the registers hold T^-1.  The column rule is folded into the rank-1 update by replacing entry K of the
// broadcast row by 1/d - 1; the row rule is a masked copy of the broadcast row.  The caller carries T = -S (S SPD), so every
// pivot must be negative, normal and finite.
//
// Hazard rule (gfx9): a VGPR written by a VALU instruction must not be read through DPP within the next 2 wait states --
// the hardware does NOT interlock (tools/probe_dpp64.hip shows the stale read).  All DPP reads therefore live in asm blocks
// that start with `s_nop 1`, and every write to a matrix register between two blocks is itself inside such a block or
// precedes its `s_nop`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sweep.h"

namespace hommx {
namespace accl {

// literal exec masks of the acc layout
template <int K> struct RowMask {  // lane row K: lanes 16 K .. 16 K + 15
  static constexpr unsigned lo = K == 0 ? 0x0000FFFFu : K == 1 ? 0xFFFF0000u : 0u;
  static constexpr unsigned hi = K == 2 ? 0x0000FFFFu : K == 3 ? 0xFFFF0000u : 0u;
};
template <int J> struct ColMask {  // lane column J of every lane row
  static constexpr unsigned lo = (1u << J) | (1u << (16 + J));
  static constexpr unsigned hi = lo;
};
template <int K, int J> struct OneLane {
  static constexpr int bit = 16 * K + J;
  static constexpr unsigned lo = bit < 32 ? 1u << (bit & 31) : 0u;
  static constexpr unsigned hi = bit >= 32 ? 1u << (bit & 31) : 0u;
};

// Predicated single instructions under a literal exec mask.  These helpers assume EXEC is all ones on entry and leave it
// all ones: the kernels that use them run ONE full wavefront per workgroup and call them outside any divergent region, so
// there is nothing to save (one scalar instruction less per use than sweep.h's save / restore form).
#define HOMMX_EXEC_SET(LO, HI) "s_mov_b32 exec_lo, %[" LO "]\n\ts_mov_b32 exec_hi, %[" HI "]\n\t"
#define HOMMX_EXEC_ALL "s_mov_b64 exec, -1"
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void amov(double& d, double v) {
  asm(HOMMX_EXEC_SET("lo", "hi") "v_mov_b64 %[d], %[v]\n\t" HOMMX_EXEC_ALL : [d] "+v"(d) : [v] "v"(v), [lo] "i"(LO), [hi] "i"(HI));
}
// d -= v on the lanes of (LO, HI)
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_sub(double& d, double v) {
  asm(HOMMX_EXEC_SET("lo", "hi") "v_add_f64 %[d], %[d], -%[v]\n\t" HOMMX_EXEC_ALL
      : [d] "+v"(d) : [v] "v"(v), [lo] "i"(LO), [hi] "i"(HI));
}
// d_q -= v on the lanes of mask q, q = 0..3 (four registers, four masks, one exec restore)
template <unsigned L0, unsigned H0, unsigned L1, unsigned H1, unsigned L2, unsigned H2, unsigned L3, unsigned H3>
__device__ __forceinline__ void masked_sub4(double& d0, double& d1, double& d2, double& d3, double v) {
  asm(HOMMX_EXEC_SET("l0", "h0") "v_add_f64 %[d0], %[d0], -%[v]\n\t" HOMMX_EXEC_SET("l1", "h1") "v_add_f64 %[d1], %[d1], -%[v]\n\t"
      HOMMX_EXEC_SET("l2", "h2") "v_add_f64 %[d2], %[d2], -%[v]\n\t" HOMMX_EXEC_SET("l3", "h3") "v_add_f64 %[d3], %[d3], -%[v]\n\t"
      HOMMX_EXEC_ALL
      : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3)
      : [v] "v"(v), [l0] "i"(L0), [h0] "i"(H0), [l1] "i"(L1), [h1] "i"(H1), [l2] "i"(L2), [h2] "i"(H2), [l3] "i"(L3), [h3] "i"(H3));
}
// LDS store of one double by the lanes of (LO, HI); `addr` is the byte offset in LDS.  In-order with the wave's other LDS
// traffic (the LDS queue of a wave is FIFO); the compiler's lgkmcnt bookkeeping stays conservative-correct around it.
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_lds_store(unsigned addr, double v) {
  asm volatile(HOMMX_EXEC_SET("lo", "hi") "ds_write_b64 %[a], %[v]\n\t" HOMMX_EXEC_ALL
               : : [a] "v"(addr), [v] "v"(v), [lo] "i"(LO), [hi] "i"(HI) : "memory");
}
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_lds_store2(unsigned addr0, double v0, unsigned addr1, double v1) {
  asm volatile(HOMMX_EXEC_SET("lo", "hi") "ds_write_b64 %[a0], %[v0]\n\tds_write_b64 %[a1], %[v1]\n\t" HOMMX_EXEC_ALL
               : : [a0] "v"(addr0), [v0] "v"(v0), [a1] "v"(addr1), [v1] "v"(v1), [lo] "i"(LO), [hi] "i"(HI) : "memory");
}
// row rule of the exchange: row K <- broadcast row, under the lane-row mask.  The broadcast row carries 1/d - 1 at entry K where
// the exchange wants 1/d; nothing reads a pivoted diagonal entry again during the sweep (later pivots only add to it), so the
// missing "+ 1" of all NB diagonal entries is added once at the end (diag_plus_one): 2 masked moves per pivot instead of 3.
template <unsigned RLO, unsigned RHI>
__device__ __forceinline__ void row_rule2(double& a0, double& a1, double w0, double w1) {
  asm(HOMMX_EXEC_SET("rlo", "rhi") "v_mov_b64 %[a0], %[w0]\n\tv_mov_b64 %[a1], %[w1]\n\t" HOMMX_EXEC_ALL
      : [a0] "+v"(a0), [a1] "+v"(a1) : [w0] "v"(w0), [w1] "v"(w1), [rlo] "i"(RLO), [rhi] "i"(RHI));
}
template <unsigned RLO, unsigned RHI>
__device__ __forceinline__ void row_rule1(double& a0, double w0) {
  asm(HOMMX_EXEC_SET("rlo", "rhi") "v_mov_b64 %[a0], %[w0]\n\t" HOMMX_EXEC_ALL : [a0] "+v"(a0) : [w0] "v"(w0), [rlo] "i"(RLO), [rhi] "i"(RHI));
}
// diagonal of a diagonal tile: register r holds it on the lanes (j = 4 r + k, k)
template <int R> struct DiagMask {
  static constexpr unsigned lo = OneLane<0, 4 * R>::lo | OneLane<1, 4 * R + 1>::lo | OneLane<2, 4 * R + 2>::lo | OneLane<3, 4 * R + 3>::lo;
  static constexpr unsigned hi = OneLane<0, 4 * R>::hi | OneLane<1, 4 * R + 1>::hi | OneLane<2, 4 * R + 2>::hi | OneLane<3, 4 * R + 3>::hi;
};
__device__ __forceinline__ void diag_plus_one(double& d0, double& d1, double& d2, double& d3) {
  asm(HOMMX_EXEC_SET("l0", "h0") "v_add_f64 %[d0], %[d0], 1.0\n\t" HOMMX_EXEC_SET("l1", "h1") "v_add_f64 %[d1], %[d1], 1.0\n\t"
      HOMMX_EXEC_SET("l2", "h2") "v_add_f64 %[d2], %[d2], 1.0\n\t" HOMMX_EXEC_SET("l3", "h3") "v_add_f64 %[d3], %[d3], 1.0\n\t"
      HOMMX_EXEC_ALL
      : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3)
      : [l0] "i"(DiagMask<0>::lo), [h0] "i"(DiagMask<0>::hi), [l1] "i"(DiagMask<1>::lo), [h1] "i"(DiagMask<1>::hi),
        [l2] "i"(DiagMask<2>::lo), [h2] "i"(DiagMask<2>::hi), [l3] "i"(DiagMask<3>::lo), [h3] "i"(DiagMask<3>::hi));
}

// byte offset of a __shared__ object in LDS
template <class T>
__device__ __forceinline__ unsigned lds_offset(T* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)p;
}

// "not a usable pivot" of T = -S: d must be negative, normal and finite, i.e. its high word in [0x80100000, 0xffefffff], i.e.
// hi + 0x00100000 >= 0x80200000 without wrapping.  d comes from v_readlane; the sweep keeps the running minimum of hi + 0x00100000.
// (The compiler folds the 32 minima into a v_min3_u32 chain AFTER the sweep; forcing them onto the scalar unit with inline asm puts
// 2 SALU per pivot into the dependent chain behind v_readlane and costs 1 % of the kernel: measured, rejected.)
__device__ __forceinline__ double readlane_neg_pivot(double v, int lane, unsigned& mn) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  const unsigned t = (unsigned)hi + 0x100000u;
  mn = t < mn ? t : mn;
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bad_pivot_min(unsigned mn) { return mn < 0x80200000u; }

#define HOMMX_BC " row_newbcast:%[jk] row_mask:0xf bank_mask:0xf\n\t"
// y += bcast(x) * wo ; x += bcast(x) * wk   (x = register of the pivot's column tile, y = the other column tile)
#define HOMMX_PAIR(X, Y) "v_fmac_f64_dpp %[" Y "], %[" X "], %[wo]" HOMMX_BC "v_fmac_f64_dpp %[" X "], %[" X "], %[wk]" HOMMX_BC
#define HOMMX_ONE(X) "v_fmac_f64_dpp %[" X "], %[" X "], %[wk]" HOMMX_BC
// reciprocal of the NEXT pivot interleaved with the updates: v_rcp_f64 (2^-24.4 on gfx950, tools/probe_rcp64.hip) and ONE
// third-order step  e = 1 - d r;  r <- r + r (e + e^2)  (error e^3: 1 ulp, one instruction less than two Newton steps)
#define HOMMX_RCP0 "v_rcp_f64 %[r], %[d]\n\t"
#define HOMMX_RCPE "v_fma_f64 %[e], -%[d], %[r], 1.0\n\t"
#define HOMMX_RCPT "v_fma_f64 %[e], %[e], %[e], %[e]\n\t"
#define HOMMX_RCPR "v_fma_f64 %[r], %[r], %[e], %[r]\n\t"

template <int NB> struct Sweep;

// ---------------------------------------------------------------------------------------------------------------------
// NB = 32: 2 x 2 tiles, 16 registers per lane
// ---------------------------------------------------------------------------------------------------------------------
template <>
struct Sweep<32> {
  static constexpr int NB = 32, NT = 2;
  typedef double Mat[2][2][4];

  // ubuf: LDS, 4 row buffers of NB doubles ([lane row][tj][j]); lk = NB * (lane row of this lane)
  template <int K>
  static __device__ __forceinline__ void step(Mat& a, double* ubuf, int lk, int j, unsigned& bad, double u0, double u1, double pinv, int npiv) {
    constexpr int tK = K / 16, jK = K % 16, rK = (K % 16) / 4, kK = K % 4, o = 1 - tK;
    constexpr bool more = K + 1 < NB;
    constexpr int K1 = more ? K + 1 : K;
    constexpr int tK1 = K1 / 16, jK1 = K1 % 16, rK1 = (K1 % 16) / 4, kK1 = K1 % 4;
    constexpr int E = tK1 * 4 + rK1;  // (tile row, register) holding pivot row K + 1: updated first
#define PI(i) ((i) < E ? (i) : (i) + 1)
#define XR(i) a[PI(i) / 4][tK][PI(i) % 4]
#define YR(i) a[PI(i) / 4][o][PI(i) % 4]
    // broadcast row: w = -u / d.  Its entry K arrives as d - 1 (the publisher overwrote that slot), so w[K] = 1/d - 1: the
    // column rule rides on the rank-1 update
    double w[2];
    w[0] = u0 * -pinv;
    w[1] = u1 * -pinv;
    double nu0 = 0.0, nu1 = 0.0, pn = 1.0;
    if constexpr (more) {
      // the registers of pivot row K + 1 first, then publish it (raw) and fetch the next pivot
      asm volatile("s_nop 1\n\t" HOMMX_PAIR("x", "y")
                   : [x] "+v"(a[tK1][tK][rK1]), [y] "+v"(a[tK1][o][rK1])
                   : [wk] "v"(w[tK]), [wo] "v"(w[o]), [jk] "n"(jK));
      // publish: every lane row stores its row of these registers into ITS OWN copy of the row buffer (no exec change);
      // copy kK1 is pivot row K + 1
      ubuf[lk + j] = a[tK1][0][rK1];
      ubuf[lk + 16 + j] = a[tK1][1][rK1];
      const double dn = readlane_neg_pivot(a[tK1][tK1][rK1], 16 * kK1 + jK1, bad);
      ubuf[NB * kK1 + K1] = dn - 1.0;  // every lane stores the same value: slot K + 1 <- d - 1
      // raw pivot row K + 1 for the next step: in flight during the bulk of the update below
      nu0 = ubuf[NB * kK1 + j];
      nu1 = ubuf[NB * kK1 + 16 + j];
      asm volatile("" ::: "memory");  // keep the loads in front of the update block
      double e;
      asm volatile(HOMMX_RCP0 HOMMX_PAIR("x0", "y0") HOMMX_PAIR("x1", "y1") HOMMX_RCPE HOMMX_PAIR("x2", "y2")
                       HOMMX_PAIR("x3", "y3") HOMMX_RCPT HOMMX_PAIR("x4", "y4") HOMMX_PAIR("x5", "y5") HOMMX_RCPR
                           HOMMX_PAIR("x6", "y6")
                   : [x0] "+v"(XR(0)), [y0] "+v"(YR(0)), [x1] "+v"(XR(1)), [y1] "+v"(YR(1)), [x2] "+v"(XR(2)), [y2] "+v"(YR(2)),
                     [x3] "+v"(XR(3)), [y3] "+v"(YR(3)), [x4] "+v"(XR(4)), [y4] "+v"(YR(4)), [x5] "+v"(XR(5)), [y5] "+v"(YR(5)),
                     [x6] "+v"(XR(6)), [y6] "+v"(YR(6)), [r] "=&v"(pn), [e] "=&v"(e)
                   : [wk] "v"(w[tK]), [wo] "v"(w[o]), [d] "s"(dn), [jk] "n"(jK));
    } else {
      asm volatile("s_nop 1\n\t" HOMMX_PAIR("x0", "y0") HOMMX_PAIR("x1", "y1") HOMMX_PAIR("x2", "y2") HOMMX_PAIR("x3", "y3")
                       HOMMX_PAIR("x4", "y4") HOMMX_PAIR("x5", "y5") HOMMX_PAIR("x6", "y6") HOMMX_PAIR("x7", "y7")
                   : [x0] "+v"(a[0][tK][0]), [y0] "+v"(a[0][o][0]), [x1] "+v"(a[0][tK][1]), [y1] "+v"(a[0][o][1]),
                     [x2] "+v"(a[0][tK][2]), [y2] "+v"(a[0][o][2]), [x3] "+v"(a[0][tK][3]), [y3] "+v"(a[0][o][3]),
                     [x4] "+v"(a[1][tK][0]), [y4] "+v"(a[1][o][0]), [x5] "+v"(a[1][tK][1]), [y5] "+v"(a[1][o][1]),
                     [x6] "+v"(a[1][tK][2]), [y6] "+v"(a[1][o][2]), [x7] "+v"(a[1][tK][3]), [y7] "+v"(a[1][o][3])
                   : [wk] "v"(w[tK]), [wo] "v"(w[o]), [jk] "n"(jK));
    }
#undef PI
#undef XR
#undef YR
    // row rule: row K <- broadcast row; (K, K) <- 1/d
    row_rule2<RowMask<kK>::lo, RowMask<kK>::hi>(a[tK][0][rK], a[tK][1][rK], w[0], w[1]);
    if constexpr (more) {
      if (K + 1 < npiv) step<K + 1>(a, ubuf, lk, j, bad, nu0, nu1, pn, npiv);
    }
  }

  // a <- a^-1 (every pivot negative).  ubuf: 4 NB doubles of LDS nobody else touches during the sweep (one row buffer per lane row).
  // npiv < NB: rows / columns >= npiv are an identity padding (diagonal -1, zero elsewhere) -- they are not pivoted and their
  // diagonal comes out as 0 instead of -1, which no product of a padded matrix ever sees.
  static __device__ __forceinline__ void run(Mat& a, double* ubuf, int j, int k, int& bad, int npiv = NB) {
    const int lk = NB * k;
    ubuf[lk + j] = a[0][0][0];
    ubuf[lk + 16 + j] = a[0][1][0];
    unsigned b = 0xffffffffu;
    const double d0 = readlane_neg_pivot(a[0][0][0], 0, b);
    ubuf[0] = d0 - 1.0;
    const double u0 = ubuf[j], u1 = ubuf[16 + j];
    step<0>(a, ubuf, lk, j, b, u0, u1, fast_rcp(d0), npiv);
    diag_plus_one(a[0][0][0], a[0][0][1], a[0][0][2], a[0][0][3]);
    diag_plus_one(a[1][1][0], a[1][1][1], a[1][1][2], a[1][1][3]);
    bad |= bad_pivot_min(b);
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// NB = 16: one tile, 4 registers per lane
// ---------------------------------------------------------------------------------------------------------------------
template <>
struct Sweep<16> {
  static constexpr int NB = 16, NT = 1;
  typedef double Mat[1][1][4];

  template <int K>
  static __device__ __forceinline__ void step(Mat& a, double* ubuf, int lk, int j, unsigned& bad, double u0, double pinv, int npiv) {
    constexpr int jK = K, rK = K / 4, kK = K % 4;
    constexpr bool more = K + 1 < NB;
    constexpr int K1 = more ? K + 1 : K;
    constexpr int jK1 = K1, rK1 = K1 / 4, kK1 = K1 % 4;
#define PI(i) ((i) < rK1 ? (i) : (i) + 1)
    const double w = u0 * -pinv;  // entry K arrives as d - 1: w[K] = 1/d - 1 (column rule)
    double nu0 = 0.0, pn = 1.0;
    if constexpr (more) {
      asm volatile("s_nop 1\n\t" HOMMX_ONE("x") : [x] "+v"(a[0][0][rK1]) : [wk] "v"(w), [jk] "n"(jK));
      ubuf[lk + j] = a[0][0][rK1];
      const double dn = readlane_neg_pivot(a[0][0][rK1], 16 * kK1 + jK1, bad);
      ubuf[NB * kK1 + K1] = dn - 1.0;
      nu0 = ubuf[NB * kK1 + j];
      asm volatile("" ::: "memory");
      double e;
      asm volatile(HOMMX_RCP0 HOMMX_ONE("x0") HOMMX_RCPE HOMMX_ONE("x1") HOMMX_RCPT HOMMX_ONE("x2") HOMMX_RCPR
                   : [x0] "+v"(a[0][0][PI(0)]), [x1] "+v"(a[0][0][PI(1)]), [x2] "+v"(a[0][0][PI(2)]), [r] "=&v"(pn), [e] "=&v"(e)
                   : [wk] "v"(w), [d] "s"(dn), [jk] "n"(jK));
    } else {
      asm volatile("s_nop 1\n\t" HOMMX_ONE("x0") HOMMX_ONE("x1") HOMMX_ONE("x2") HOMMX_ONE("x3")
                   : [x0] "+v"(a[0][0][0]), [x1] "+v"(a[0][0][1]), [x2] "+v"(a[0][0][2]), [x3] "+v"(a[0][0][3])
                   : [wk] "v"(w), [jk] "n"(jK));
    }
#undef PI
#ifdef HOMMX_ROW_RULE_EXEC
    row_rule1<RowMask<kK>::lo, RowMask<kK>::hi>(a[0][0][rK], w);
#else
    a[0][0][rK] = (lk == NB * kK) ? w : a[0][0][rK];  // two v_cndmask_b32 on a loop-invariant lane-row mask: cheaper than 3 SALU + 1 move
#endif
    if constexpr (more) {
      if (K + 1 < npiv) step<K + 1>(a, ubuf, lk, j, bad, nu0, pn, npiv);
    }
  }

  static __device__ __forceinline__ void run(Mat& a, double* ubuf, int j, int k, int& bad, int npiv = NB) {
    const int lk = NB * k;
    ubuf[lk + j] = a[0][0][0];
    unsigned b = 0xffffffffu;
    const double d0 = readlane_neg_pivot(a[0][0][0], 0, b);
    ubuf[0] = d0 - 1.0;
    const double u0 = ubuf[j];
    step<0>(a, ubuf, lk, j, b, u0, fast_rcp(d0), npiv);
    diag_plus_one(a[0][0][0], a[0][0][1], a[0][0][2], a[0][0][3]);
    bad |= bad_pivot_min(b);
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// 32 x 32 as a 2 x 2 block inverse around two 16-sweeps, the six Schur products on the matrix cores
// ---------------------------------------------------------------------------------------------------------------------
// T = [[TA, U], [U^T, TC]] (tiles a[0][0], a[0][1], -, a[1][1]; a[1][0] is not read: T is symmetric):
//     Ai = TA^-1,  X = U^T Ai,  Sc = TC - X U,  N22 = Sc^-1,  N21 = -N22 X,  N12 = N21^T,  N11 = Ai - X^T N21.
// In the acc layout a tile's register r is k-slab r of the tile both as B operand (B(k, j) = M[k][j]) and as A operand of its
// transpose (A(i, k) = M[k][i]), so every product takes its operands straight from the registers of the previous one.  Same
// fp64 pipe time as the 32-sweep (24 MFMAs = 384 FMA slots replace 32 x 16 - 2 x 16 x 4 = 384 DPP FMAs), but 2 x 16 x 12 fewer
// bookkeeping slots and half the dependent pivot chain: a wave issues one fp64 VALU op per ~8 cycles, an MFMA carries 16 of them.
typedef double v4d __attribute__((ext_vector_type(4)));
// transpose of a 16 x 16 tile held in the acc layout, through 16 x 17 doubles of LDS (pitch 17: both passes conflict-free); one wave
__device__ __forceinline__ v4d transpose_tile(v4d x, double* tsc, int j, int k) {
#pragma unroll
  for (int r = 0; r < 4; ++r) tsc[(4 * r + k) * 17 + j] = x[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  v4d y;
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = tsc[j * 17 + 4 * r + k];
  __builtin_amdgcn_wave_barrier();
  return y;
}
// tsc: 16 x 17 doubles of LDS scratch.  npiv (> 16) < 32: rows / columns >= npiv are an identity padding, see Sweep<32>::run.
// Ai U is formed on the matrix cores and its transpose U^T Ai taken through LDS; likewise N12 = N21^T: 16 MFMAs, not 24.
__device__ __forceinline__ void block_inverse32(double (&a)[2][2][4], double* ubuf, double* tsc, int j, int k, int& bad, int npiv = 32) {
  auto mm = [](double x, double y, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); };
  double t00[1][1][4], sc[1][1][4], nu[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { t00[0][0][r] = a[0][0][r]; nu[r] = -a[0][1][r]; }
  Sweep<16>::run(t00, ubuf, j, k, bad);
  v4d nxt = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 4; ++q) nxt = mm(t00[0][0][q], nu[q], nxt);  // -X^T[i][j] = sum_k Ai[k][i] (-U[k][j])
  const v4d nx = transpose_tile(nxt, tsc, j, k);                   // -X = -U^T Ai
  {
    v4d c = v4d{a[1][1][0], a[1][1][1], a[1][1][2], a[1][1][3]};
#pragma unroll
    for (int q = 0; q < 4; ++q) c = mm(nxt[q], a[0][1][q], c);  // Sc = TC + (-X) U
#pragma unroll
    for (int r = 0; r < 4; ++r) sc[0][0][r] = c[r];
  }
  Sweep<16>::run(sc, ubuf, j, k, bad, npiv - 16);
  v4d n21 = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 4; ++q) n21 = mm(sc[0][0][q], nx[q], n21);  // N21 = N22 (-X)
  v4d n11 = v4d{t00[0][0][0], t00[0][0][1], t00[0][0][2], t00[0][0][3]};
#pragma unroll
  for (int q = 0; q < 4; ++q) n11 = mm(nx[q], n21[q], n11);  // N11 = Ai + (-X)^T N21
  const v4d n12 = transpose_tile(n21, tsc, j, k);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a[0][0][r] = n11[r];
    a[0][1][r] = n12[r];
    a[1][0][r] = n21[r];
    a[1][1][r] = sc[0][0][r];
  }
}

}  // namespace accl
}  // namespace hommx
