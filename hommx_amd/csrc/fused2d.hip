// fused2d.hip -- fused micro-cell pipeline for 2D scalar Poisson / stratified Poisson (gfx950).
//
// One wavefront (64 lanes) per macro cell.  For that cell the wave
//   K1  assembles the periodic P1 stiffness of the n x n unit-cell mesh on the fly from the
//       per-element coefficient stream (hmm.py:644-650, 759-772; periodic identification
//       cell_problem.py:38-136 == indices mod n),
//   K2  eliminates it as a block-cyclic-tridiagonal system (block = one node row, b = n <= NB),
//       carrying the two canonical right-hand sides along, so that the effective tensor is the
//       Schur complement  A_H = C0 - B^T K^+ B  (== the energy functional hmm.py:652-667 / 774-789),
//   K3  reduces the 2x2 result with wave shuffles and writes 4 doubles.
//
// Per block row j (node row j of the torus), with S the current Schur block (NB x NB), W the
// "arrow" block that couples the last node row to row j, S_last the Schur block of the last row:
//      N      = -S^-1                     symmetric Gauss-Jordan sweep, matrix held in VGPRs
//      V'     = W N                       v_mfma_f64_16x16x4_f64, A operand from LDS, B from VGPRs
//      S_last += V' W^T                   v_mfma_f64_16x16x4_f64, both operands in VGPRs
//      X      = N E^T                     sparse (E = coupling row j+1 <- row j, bidiagonal), VALU
//      W_next = V' E^T = W X              v_mfma_f64_16x16x4_f64 again: no cross-lane traffic at all
//      S_next = D_{j+1} + E X             sparse, VALU
// Sign convention: the sweep produces N = -S^-1; primes mark quantities carrying that sign.
//
// Register layouts (l = lane):
//   "BLK"      lane owns the BS x BS block (bi = l >> 3, bj = l & 7), BS = NB / 8 (sweep; 2*BS LDS doubles per pivot)
//   "strip"    lane owns column c = l % NB, rows r0 + i, r0 = (l / NB) * RPL   (only to read N back for R N)
//   "operand"  wf[t][kk] = W[16 t + (l & 15)][4 kk + (l >> 4)]      (A and B fragments of the f64 MFMA)
//   "C"        acc[ti][tj][r] = X[16 ti + (l >> 4) + 4 r][16 tj + (l & 15)]  (f64 MFMA accumulator map)
// The product V'^T = N W^T in C layout IS V' in operand layout, so it feeds the second MFMA chain
// without leaving the register file.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "sweep.h"

#ifndef HOMMX_FUSED_WAVES_PER_SIMD
#define HOMMX_FUSED_WAVES_PER_SIMD 2
#endif

namespace hommx {

typedef double d4 __attribute__((ext_vector_type(4)));

// LDS matrix index with an XOR swizzle on odd rows (NB = 32) so that b64 accesses whose lanes
// 0-15 / 16-31 touch consecutive rows fall on disjoint bank halves.
template <int NB>
__device__ __forceinline__ int midx(int row, int col) {
  if (NB == 32) return row * NB + (col ^ ((row & 1) << 4));
  return row * NB + col;
}

// wave-uniform double -> SGPR pair (keeps loop-invariant scalars out of the vector register file)
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

struct CoefRow {
  double a0, a1;    // coefficient of triangle 0 (v0,v1,v3) / 1 (v0,v2,v3) of cell (c, row)
  double a0m, a1m;  // same for cell (c-1, row), cyclic
};

template <int NB>
struct alignas(16) Lds {
  double mat[NB * NB];   // N = -S^-1 (symmetric, swizzled) / band-matrix indexer / S_last transfer
  double ubuf[NB];       // sweep: raw pivot row (published one pivot ahead)
  double rbuf[2][NB];    // R rows of the current block
  double vrbuf[2][NB];   // Vr' = R N
  double e0[NB];         // E[r][r]
  double e1[NB];         // E[r][r-1]
  double vcol[NB];       // V'[:, n-1]: the wrap-around neighbour column of W_next
#ifdef HOMMX_SL_IN_LDS
  double slbuf[NB * NB]; // S_last accumulators parked between two S_last updates (lane-private slots)
#endif
};

template <int NB>
__global__ __launch_bounds__(64, HOMMX_FUSED_WAVES_PER_SIMD) void k_poisson2d_fused(
    const double* __restrict__ coef, const double* __restrict__ Mmat, double* __restrict__ out,
    int32_t* __restrict__ info, int n, long long ncells, const unsigned char* __restrict__ mask) {
  constexpr int RPL = Cfg<NB>::RPL, CG = Cfg<NB>::CG, NT = Cfg<NB>::NT, KK = Cfg<NB>::KK;
  __shared__ Lds<NB> L;

  const long long cell = blockIdx.x;
  if (cell >= ncells) return;
  __builtin_assume(n >= 3);  // checked by the host (hommx_plan_create); lets the compiler drop zero-trip paths
  __builtin_assume(n <= NB);
  const int l = threadIdx.x;
  const int c = l % NB, g = l / NB, r0 = g * RPL;
  const int lb = l - c;
  const bool valid = c < n;
  const int cm = valid ? (c == 0 ? n - 1 : c - 1) : c;  // cyclic left neighbour
  const int cp = valid ? (c == n - 1 ? 0 : c + 1) : c;  // cyclic right neighbour
  const int l15 = l & 15, l4 = l >> 4;
  // LDS matrix addressing.  The XOR swizzle of odd rows (NB = 32) is folded into a handful of per-lane base
  // indices so that every access is base + compile-time offset (affine => one VGPR per base, not per address):
  //   GJ(i)        = element (r0 + i, c)                         parity of the row = parity of i
  //   TILE(a, row) = element (row + l4, 16 a + l15), row % 4 == 0   parity of the row = parity of l4
  constexpr int SW = (NB == 32) ? 16 : 0;
  const int gjE = r0 * NB + c, gjO = r0 * NB + (c ^ SW);
  int tileB[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) tileB[a] = l4 * NB + ((16 * a + l15) ^ ((l4 & 1) ? SW : 0));
#define GJ(i) (((i) & 1 ? gjO : gjE) + (i) * NB)
#define TILE(a, row) (tileB[a] + (row) * NB)
  //   BLK(r, q)    = element (BS bi + r, BS bj + q)                parity of the row = parity of r (BS is even)
  constexpr int BS = NB / 8;
  const int bi = l >> 3, bj = l & 7;
  const int blkE = BS * bi * NB + BS * bj, blkO = BS * bi * NB + ((BS * bj) ^ SW);
#define BLK(r, q) (((r) & 1 ? blkO : blkE) + (r) * NB + (q))

  // ---- stratification matrix M = Dtheta^T(c_T) -> Q = M^T M (hmm.py:759-766) -------------------
  double m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;
  if (Mmat) {
    const double* mp = Mmat + cell * 4;
    m00 = mp[0]; m01 = mp[1]; m10 = mp[2]; m11 = mp[3];
  }
  const double al = uniform_f64(0.5 * (m00 * m00 + m10 * m10));
  const double be = uniform_f64(0.5 * (m01 * m01 + m11 * m11));
  const double ga = uniform_f64(0.5 * (m00 * m01 + m10 * m11));
  const double ab = uniform_f64(al - 2.0 * ga + be);

  // Coefficient source: the element stream coef[cell][2 n^2], or -- two-phase media -- a phase mask shared by all
  // cells (mask[2 n^2], one byte per element) and the two phase values of this cell, coef[cell][2] = (phase 0, phase 1).
  const double* cc = mask ? coef + cell * 2 : coef + cell * (2ll * n * n);
  double ph0 = 0.0, ph1 = 0.0;
  if (mask) {
    ph0 = cc[0];
    ph1 = cc[1];
  }

  auto load_row = [&](int jc) {
    CoefRow r;
    r.a0 = 0.0; r.a1 = 0.0;
    if (valid) {
      if (mask) {
        const uchar2 mk = *reinterpret_cast<const uchar2*>(mask + 2 * (jc * n + c));
        r.a0 = mk.x ? ph1 : ph0;
        r.a1 = mk.y ? ph1 : ph0;
      } else {
        const double2 v = *reinterpret_cast<const double2*>(cc + 2 * (jc * n + c));
        r.a0 = v.x; r.a1 = v.y;
      }
    }
    r.a0m = __shfl(r.a0, lb + cm, 64);
    r.a1m = __shfl(r.a1, lb + cm, 64);
    return r;
  };
  // stencil of node row j from cell rows j (cur, above the nodes) and j-1 (prev, below)
  auto st_diag = [&](const CoefRow& cur, const CoefRow& prev) {
    return cur.a0 * al + cur.a1 * be + cur.a0m * ab + prev.a0m * be + prev.a1m * al + prev.a1 * ab;
  };
  auto st_E = [&](const CoefRow& cur, const CoefRow& prev) { return (cur.a0 + prev.a1) * (ga - al); };
  auto st_N = [&](const CoefRow& cur) { return (cur.a1 + cur.a0m) * (ga - be); };
  auto st_NE = [&](const CoefRow& cur) { return -ga * (cur.a0 + cur.a1); };
  auto st_p0 = [&](const CoefRow& cur, const CoefRow& prev) { return cur.a0 - cur.a0m - prev.a1m + prev.a1; };
  auto st_p1 = [&](const CoefRow& cur, const CoefRow& prev) { return cur.a1 + cur.a0m - prev.a0m - prev.a1; };

  // write the cyclic tridiagonal D (diag dg, coupling c<->c+1 = ce) into mat [row][col]; identity on padding
  auto band_D_to_mat = [&](double dg, double ce) {
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
      for (int q = 0; q < BS; q += 2) *reinterpret_cast<double2*>(&L.mat[BLK(r, q)]) = double2{0.0, 0.0};
    __syncthreads();
    const double cem = __shfl(ce, lb + cm, 64);
    if (g == 0) {
      if (valid) {
        L.mat[midx<NB>(c, c)] = dg;
        L.mat[midx<NB>(cp, c)] = ce;
        L.mat[midx<NB>(cm, c)] = cem;
      } else {
        L.mat[midx<NB>(c, c)] = 1.0;
      }
    }
    __syncthreads();
  };
  // write a bidiagonal coupling X into mat TRANSPOSED ([col][row]) for an operand-layout read:
  //   up == true :  X[x][x] = dv, X[x][x+1] = ov     (U orientation: rows lower node row)
  //   up == false:  X[x][x] = dv, X[x+1][x] = ov     (E = U^T)
  auto band_X_to_matT = [&](double dv, double ov, bool up) {
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
      for (int q = 0; q < BS; q += 2) *reinterpret_cast<double2*>(&L.mat[BLK(r, q)]) = double2{0.0, 0.0};
    __syncthreads();
    if (g == 0 && valid) {
      L.mat[midx<NB>(c, c)] = dv;
      if (up) L.mat[midx<NB>(cp, c)] = ov;  // X[c][cp] stored at [col cp][row c]
      else    L.mat[midx<NB>(c, cp)] = ov;  // X[cp][c] stored at [col c][row cp]
    }
    __syncthreads();
  };

  // ---- prologue: rows n-2, n-1, 0 ---------------------------------------------------------------
  double wf[NT][KK];  // W, operand layout
  double s[RPL];      // S, BLK layout (RPL == BS * BS)
#ifndef HOMMX_SL_IN_LDS
  d4 slr[NT][NT];     // S_last accumulators (lower tiles) resident in registers instead of LDS slots
#endif
  double rr[2], rl[2];
  CoefRow cur;
  // C0 = int_Y A (the corrector-free part of hmm.py:652-667) is accumulated while the coefficient lines stream by:
  // rows n-1 and 0 here, rows 1 .. n-2 in the loop -- every line exactly once (every lane group holds a copy).
  double asum;
  {
    const CoefRow rowA = load_row(n - 2);
    const CoefRow rowB = load_row(n - 1);
    cur = load_row(0);
    asum = (rowB.a0 + rowB.a1) + (cur.a0 + cur.a1);
    // S_last = D_{n-1}
    band_D_to_mat(st_diag(rowB, rowA), st_E(rowB, rowA));
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#ifndef HOMMX_SL_IN_LDS
          if (tj <= ti) slr[ti][tj][r] = L.mat[TILE(tj, 16 * ti + 4 * r)];
#else
          L.slbuf[((ti * NT + tj) * 4 + r) * 64 + l] = L.mat[TILE(tj, 16 * ti + 4 * r)];
#endif
        }
    __syncthreads();
    // W_0 = K[(., n-1), (., 0)] = U_{n-1}
    band_X_to_matT(st_N(rowB), st_NE(rowB), true);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) wf[t][kk] = L.mat[TILE(t, 4 * kk)];
    __syncthreads();
    // S_0 = D_0
    band_D_to_mat(st_diag(cur, rowB), st_E(cur, rowB));
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
      for (int q = 0; q < BS; q += 2) {
        const double2 v = *reinterpret_cast<const double2*>(&L.mat[BLK(r, q)]);
        s[r * BS + q] = v.x; s[r * BS + q + 1] = v.y;
      }
    __syncthreads();
    rr[0] = st_p0(cur, rowB); rr[1] = st_p1(cur, rowB);    // R_0
    rl[0] = st_p0(rowB, rowA); rl[1] = st_p1(rowB, rowA);  // R_last
  }
  double g00 = 0.0, g01 = 0.0, g11 = 0.0;  // -G partial sums (every lane group holds a copy)
  int bad = 0, badstep = 0;
  const int kq = (n - 1) >> 2, lq = (n - 1) & 3;  // where column n-1 lives in operand layout

  // ---- elimination of node rows 0 .. n-2 ----------------------------------------------------------
  for (int j = 0; j <= n - 2; ++j) {
    const bool lastStep = (j == n - 2);
    // next coefficient line early (latency hidden behind the sweep)
    CoefRow nxt = cur;
    if (!lastStep) {
      nxt = load_row(j + 1);
      asum += nxt.a0 + nxt.a1;
    }
    // coupling E = K[(., j+1), (., j)] from cell row j:  E[r][r] = cN[r], E[r][r-1] = cNE[r-1]
    const double e0c = st_N(cur);
    const double e1c = __shfl(st_NE(cur), lb + cm, 64);
    if (lastStep) {
      // the last node row couples to row n-2 through E as well as through the arrow: W += E
      band_X_to_matT(e0c, st_NE(cur), false);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wf[t][kk] += L.mat[TILE(t, 4 * kk)];
      __syncthreads();
    }
    if (g == 0) {
      L.rbuf[0][c] = rr[0];
      L.rbuf[1][c] = rr[1];
      L.e0[c] = e0c;
      L.e1[c] = e1c;
    }

    // (1) N = -S^-1
#ifndef HOMMX_ABLATE_SWEEP
    int badj = 0;
    sweep_blk<NB>(s, L.ubuf, bi, bj, badj);
    if (badj && !bad) { bad = 1; badstep = j + 1; }
#endif

    // (2) N -> LDS: BS rows of BS consecutive doubles per lane (8 lanes cover one 256-byte matrix row)
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
      for (int q = 0; q < BS; q += 2)
        *reinterpret_cast<double2*>(&L.mat[BLK(r, q)]) = double2{s[r * BS + q], s[r * BS + q + 1]};
    __syncthreads();

    // (3) V'^T = N W^T  (C layout == V' in operand layout)
    d4 vt[NT][NT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b) vt[a][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      double af[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) af[a] = L.mat[TILE(a, 4 * kk)];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          vt[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], wf[b][kk], vt[a][b], 0, 0, 0);
    }

#ifndef HOMMX_ABLATE_GEMM2
    // (4) S_last += V' W^T.  The accumulators (C layout) stay in registers across the whole elimination (24 VGPRs for
    //     NB = 32; -DHOMMX_SL_IN_LDS parks them in lane-private LDS slots instead, the layout used while the kernel
    //     still spilled).  S_last is symmetric: only the tiles on and below the diagonal are kept (3 of 4 for NB = 32).
    {
#ifndef HOMMX_SL_IN_LDS
      d4 (&sl)[NT][NT] = slr;
#else
      d4 sl[NT][NT];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) sl[a][b][r] = L.slbuf[((a * NT + b) * 4 + r) * 64 + l];
#endif
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b)
            sl[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[kk >> 2][a][kk & 3], wf[b][kk], sl[a][b], 0, 0, 0);
#ifdef HOMMX_SL_IN_LDS
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) L.slbuf[((a * NT + b) * 4 + r) * 64 + l] = sl[a][b][r];
#endif
    }

#endif
    // (5) Vr' = R N  (partial over the lane's rows, then across lane groups); -G += Vr' R^T
    double vr[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < RPL; i += 2) {  // column strip of N (rows r0 .. r0+RPL-1 of column c) read back from LDS
      const double2 q0 = *reinterpret_cast<const double2*>(&L.rbuf[0][r0 + i]);
      const double2 q1 = *reinterpret_cast<const double2*>(&L.rbuf[1][r0 + i]);
      const double n0 = L.mat[GJ(i)], n1 = L.mat[GJ(i + 1)];
      vr[0] = fma(q0.x, n0, vr[0]); vr[0] = fma(q0.y, n1, vr[0]);
      vr[1] = fma(q1.x, n0, vr[1]); vr[1] = fma(q1.y, n1, vr[1]);
    }
    if (NB == 16) {
      vr[0] = add_xor16(vr[0]);
      vr[1] = add_xor16(vr[1]);
    }
    vr[0] = add_xor32(vr[0]);
    vr[1] = add_xor32(vr[1]);
    g00 = fma(vr[0], rr[0], g00);
    g01 = fma(vr[0], rr[1], g01);
    g11 = fma(vr[1], rr[1], g11);
    if (g == 0) {
      L.vrbuf[0][c] = vr[0];
      L.vrbuf[1][c] = vr[1];
    }
    // column n-1 of V' (wrap-around neighbour of column 0) -> vcol
    if (!lastStep && l4 == lq) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        if (kk == kq) {
#pragma unroll
          for (int t = 0; t < NT; ++t) L.vcol[16 * t + l15] = vt[kk >> 2][t][kk & 3];
        }
    }
    __syncthreads();

#ifndef HOMMX_ABLATE_RL
    // (6) R_last += Vr' W^T
    {
      double part[2][NT];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) part[m][t] = 0.0;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const double v0 = L.vrbuf[0][4 * kk + l4];
        const double v1 = L.vrbuf[1][4 * kk + l4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          part[0][t] = fma(v0, wf[t][kk], part[0][t]);
          part[1][t] = fma(v1, wf[t][kk], part[1][t]);
        }
      }
      // sum over the four 16-lane rows.  NT == 2: lane column c = 16 (row & 1) + (l & 15) wants tile (row & 1); the
      // transposing butterfly delivers exactly that: swap16 -> [A01, B01, A23, B23], swap32 -> [A, B, A, B].
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        double sum;
        if (NT == 2) {
          double a = part[m][0], b = part[m][NT - 1];
          swap16(a, b);
          sum = add_xor32(a + b);
        } else {
          sum = add_xor32(add_xor16(part[m][0]));
        }
        rl[m] += sum;
      }
    }

#endif
    if (!lastStep) {
#ifndef HOMMX_ABLATE_WNEXT
      // (7) W_next = V' E^T (sparse):  W_next[i][col] = V'[i][col] e0[col] + V'[i][col-1] e1[col].
      //     In operand layout col = 4 kk + (l >> 4): the left neighbour of a lane's column sits 16 lanes down; for lanes
      //     0-15 it is lanes 48-63 of register kk-1, and column -1 wraps to n-1 (vcol).  The select is done on the
      //     SOURCE side (lanes 48-63 offer register kk-1), then one 16-lane rotation brings every lane its neighbour.
      {
        double f0[KK], f1[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          f0[kk] = L.e0[4 * kk + l4];
          f1[kk] = L.e1[4 * kk + l4];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const double wrapv = L.vcol[16 * t + l15];
          // left neighbour of column 4 kk + row: rows 1, 3 take rows 0, 2 of the same register (swap16), row 2 takes
          // row 1 and row 0 takes row 3 of register kk - 1 (one swap32 serves both, chained through `carry`)
          double carry = 0.0;
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
            const double x = vt[kk >> 2][t][kk & 3];
            double tt = x, uu = x;
            swap16(tt, uu);  // tt = [x0, x0, x2, x2], uu = [x1, x1, x3, x3]
            double vv = (kk == 0) ? uu : carry;
            swap32(vv, uu);  // vv = [., ., x1, x1], uu = [prev x3, prev x3, x3, x3]
            carry = uu;
#ifndef HOMMX_NO_ASM_MASKS
            double z = tt;                                                   // rows 1, 3 (rows 0, 2 overwritten below)
            masked_mov<0u, 0x0000FFFFu>(z, vv);                              // row 2 = lanes 32..47
            masked_mov<0x0000FFFFu, 0u>(z, (kk == 0) ? wrapv : uu);          // row 0 = lanes  0..15
#else
            const double z = (l4 & 1) ? tt : ((l4 == 2) ? vv : ((kk == 0) ? wrapv : uu));
#endif
            wf[t][kk] = fma(z, f1[kk], x * f0[kk]);
            {  // one register at a time (hoisted, the swap temporaries of all kk spill): the next swap32 waits for this fma
              int clo = __double2loint(carry);
              asm volatile("" : "+v"(clo) : "v"(__double2loint(wf[t][kk])));
              carry = __hiloint2double(__double2hiint(carry), clo);
            }
          }
        }
      }
#endif
#ifndef HOMMX_ABLATE_SNEXT
      // (8) S_next = D_{j+1} + E N E^T in BLK layout.  With X(r, q) = e0[col q] N[r][col q] + e1[col q] N[r][col q - 1]:
      //     S_next[r][q] = D[r][q] + e0[row r] X(r, q) + e1[row r] X(r - 1, q);  row -1 / column -1 are cyclic (n - 1).
      {
        const int rowm = (bi == 0) ? n - 1 : BS * bi - 1;
        const int colm = (bj == 0) ? n - 1 : BS * bj - 1;
        double e0q[BS], e1q[BS], e0r[BS], e1r[BS];
#pragma unroll
        for (int q = 0; q < BS; q += 2) {
          const double2 a0 = *reinterpret_cast<const double2*>(&L.e0[BS * bj + q]);
          const double2 a1 = *reinterpret_cast<const double2*>(&L.e1[BS * bj + q]);
          const double2 b0 = *reinterpret_cast<const double2*>(&L.e0[BS * bi + q]);
          const double2 b1 = *reinterpret_cast<const double2*>(&L.e1[BS * bi + q]);
          e0q[q] = a0.x; e0q[q + 1] = a0.y; e1q[q] = a1.x; e1q[q + 1] = a1.y;
          e0r[q] = b0.x; e0r[q + 1] = b0.y; e1r[q] = b1.x; e1r[q + 1] = b1.y;
        }
        // halo of the block in N: the row above (with its left neighbour) and the column to the left
        double hup[BS + 1], hleft[BS];
        hup[0] = L.mat[midx<NB>(rowm, colm)];
#pragma unroll
        for (int q = 0; q < BS; ++q) hup[q + 1] = L.mat[midx<NB>(rowm, BS * bj + q)];
#pragma unroll
        for (int r = 0; r < BS; ++r) hleft[r] = L.mat[midx<NB>(BS * bi + r, colm)];
        double xup[BS];  // X(r0 - 1, .)
#pragma unroll
        for (int q = 0; q < BS; ++q) xup[q] = fma(e1q[q], hup[q], e0q[q] * hup[q + 1]);
        // X in place
#pragma unroll
        for (int r = 0; r < BS; ++r) {
          double prevN = hleft[r];
#pragma unroll
          for (int q = 0; q < BS; ++q) {
            const double cur_n = s[r * BS + q];
            s[r * BS + q] = fma(e1q[q], prevN, e0q[q] * cur_n);
            prevN = cur_n;
          }
        }
        // D_{j+1} through the LDS indexer (every lane has read its N halo after the barrier inside)
        band_D_to_mat(st_diag(nxt, cur), st_E(nxt, cur));
#pragma unroll
        for (int r = BS - 1; r >= 0; --r) {
#pragma unroll
          for (int q = 0; q < BS; q += 2) {
            const double2 dd = *reinterpret_cast<const double2*>(&L.mat[BLK(r, q)]);
            const double up0 = (r == 0) ? xup[q] : s[(r > 0 ? r - 1 : 0) * BS + q];
            const double up1 = (r == 0) ? xup[q + 1] : s[(r > 0 ? r - 1 : 0) * BS + q + 1];
            s[r * BS + q] = fma(e0r[r], s[r * BS + q], fma(e1r[r], up0, dd.x));
            s[r * BS + q + 1] = fma(e0r[r], s[r * BS + q + 1], fma(e1r[r], up1, dd.y));
          }
        }
      }
#endif
      // (9) R_next = P_{j+1} + Vr' E^T
      const double vm0 = __shfl(vr[0], lb + cm, 64);
      const double vm1 = __shfl(vr[1], lb + cm, 64);
      rr[0] = fma(vm0, e1c, fma(vr[0], e0c, st_p0(nxt, cur)));
      rr[1] = fma(vm1, e1c, fma(vr[1], e0c, st_p1(nxt, cur)));
      cur = nxt;
      __syncthreads();
    }
  }

  // ---- last node row: S_last (C layout) -> GJ layout, pin node (n-1, n-1), sweep ------------------
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b <= a; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#ifndef HOMMX_SL_IN_LDS
        const double v = slr[a][b][r];
#else
        const double v = L.slbuf[((a * NT + b) * 4 + r) * 64 + l];
#endif
        L.mat[TILE(b, 16 * a + 4 * r)] = v;                                          // (16a + l4 + 4r, 16b + l15)
        if (b < a) L.mat[midx<NB>(16 * b + l15, 16 * a + 4 * r + l4)] = v;           // mirror image
      }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int q = 0; q < BS; ++q) {
      double v = L.mat[BLK(r, q)];
      const bool prow = (BS * bi + r == n - 1), pcol = (BS * bj + q == n - 1);
      if (prow || pcol) v = (prow && pcol) ? 1.0 : 0.0;  // gauge: drop the last unknown (cell_problem.py:349-361)
      s[r * BS + q] = v;
    }
  if (c == n - 1) { rl[0] = 0.0; rl[1] = 0.0; }
  if (g == 0) {
    L.rbuf[0][c] = rl[0];
    L.rbuf[1][c] = rl[1];
  }
  __syncthreads();
  {
    int badj = 0;
    sweep_blk<NB>(s, L.ubuf, bi, bj, badj);
    if (badj && !bad) { bad = 1; badstep = n; }
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
      for (int q = 0; q < BS; q += 2)
        *reinterpret_cast<double2*>(&L.mat[BLK(r, q)]) = double2{s[r * BS + q], s[r * BS + q + 1]};
    __syncthreads();
    double vr[2] = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < RPL; i += 2) {  // column strip of N (rows r0 .. r0+RPL-1 of column c) read back from LDS
      const double2 q0 = *reinterpret_cast<const double2*>(&L.rbuf[0][r0 + i]);
      const double2 q1 = *reinterpret_cast<const double2*>(&L.rbuf[1][r0 + i]);
      const double n0 = L.mat[GJ(i)], n1 = L.mat[GJ(i + 1)];
      vr[0] = fma(q0.x, n0, vr[0]); vr[0] = fma(q0.y, n1, vr[0]);
      vr[1] = fma(q1.x, n0, vr[1]); vr[1] = fma(q1.y, n1, vr[1]);
    }
    if (NB == 16) {
      vr[0] = add_xor16(vr[0]);
      vr[1] = add_xor16(vr[1]);
    }
    vr[0] = add_xor32(vr[0]);
    vr[1] = add_xor32(vr[1]);
    g00 = fma(vr[0], rl[0], g00);
    g01 = fma(vr[0], rl[1], g01);
    g11 = fma(vr[1], rl[1], g11);
  }


  // ---- K3: wave reduction and output ----------------------------------------------------------------
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    g00 += __shfl_xor(g00, off, 64);
    g01 += __shfl_xor(g01, off, 64);
    g11 += __shfl_xor(g11, off, 64);
    asum += __shfl_xor(asum, off, 64);
  }
  if (l == 0) {
    const double h = 1.0 / n;
    const double sc = 0.25 * h * h / CG;  // every lane group accumulated a full copy
    const double c0 = 0.5 * h * h * asum / CG;
    // A_H = C0 I + (h^2/4) M Gneg M^T
    const double t00 = m00 * g00 + m01 * g01, t01 = m00 * g01 + m01 * g11;
    const double t10 = m10 * g00 + m11 * g01, t11 = m10 * g01 + m11 * g11;
    double* o = out + cell * 4;
    o[0] = c0 + sc * (t00 * m00 + t01 * m01);
    o[1] = sc * (t00 * m10 + t01 * m11);
    o[2] = sc * (t10 * m00 + t11 * m01);
    o[3] = c0 + sc * (t10 * m10 + t11 * m11);
    if (info) info[cell] = bad ? badstep : 0;
  }
#undef GJ
#undef BLK
#undef TILE
}

// ---- launch ---------------------------------------------------------------------------------------
hipError_t launch_poisson2d_fused(const double* d_coef, const double* d_M, double* d_out, int32_t* d_info,
                                  int n, long long ncells, hipStream_t stream, const unsigned char* d_mask) {
  if (ncells <= 0) return hipSuccess;
  dim3 grid((unsigned)ncells), block(64);
  if (n <= 16)
    hipLaunchKernelGGL(k_poisson2d_fused<16>, grid, block, 0, stream, d_coef, d_M, d_out, d_info, n, ncells, d_mask);
  else
    hipLaunchKernelGGL(k_poisson2d_fused<32>, grid, block, 0, stream, d_coef, d_M, d_out, d_info, n, ncells, d_mask);
  return hipGetLastError();
}

}  // namespace hommx
