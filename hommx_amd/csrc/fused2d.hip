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
// Per block row j (node row j of the torus), with S the current Schur block, W the "arrow" block that couples the last node
// row to row j, S_last the Schur block of the last row, and T = -S the matrix actually carried:
//      N      = T^-1 = -S^-1              exchange sweep in the MFMA accumulator layout (sweep_acc.h): DPP column broadcasts
//      V'^T   = N W^T                     v_mfma_f64_16x16x4_f64, A operand = the sweep's registers (N is symmetric)
//      S_last += V' W^T                   v_mfma_f64_16x16x4_f64, both operands in VGPRs
//      W_next = V' E^T                    sparse (E = coupling row j+1 <- row j, bidiagonal): row shift through LDS
//      T_next = -D_{j+1} - E N E^T        sparse: three neighbours per entry through LDS, band of D under literal exec masks
//      Vr'    = R N ; -G += Vr' R^T ; R_last += Vr' W^T ; R_next = P_{j+1} + Vr' E^T        (2 load rows)
//
// Index convention: the matrices are NB x NB (NB = 16 / 32) with the PADDING FIRST: node column c of the n-periodic row lives
// at matrix index c + p0, p0 = NB - n; the padding carries the identity (T: -1) and zero couplings.  The cyclic neighbour of
// the first real index p0 is therefore always the LAST index NB - 1: wrap-around sources sit at compile-time positions.
//
// Register layouts (lane l = 16 k + j):
//   "acc"      a[ti][tj][r] = X[16 ti + 4 r + k][16 tj + j]          (f64 MFMA accumulator map; also B operand slab 4 ti + r,
//                                                                    and A operand slab of X^T)
//   "operand"  wf[t][kk]    = W[16 t + j][4 kk + k]                  == acc layout of W^T: reg (kk >> 2, t, kk & 3)
// Per-column vectors live one entry per lane: column c = l % NB (replicated 64 / NB times).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "sweep_acc.h"

#ifndef HOMMX_FUSED_WAVES_PER_SIMD
#define HOMMX_FUSED_WAVES_PER_SIMD 2
#endif

namespace hommx {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// wave-uniform double -> SGPR pair (keeps loop-invariant scalars out of the vector register file)
__device__ __forceinline__ double uniform_f64(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

struct CoefRow {
  double a0, a1;    // coefficient of triangle 0 (v0,v1,v3) / 1 (v0,v2,v3) of cell (c, row)
  double a0m, a1m;  // same for cell (c-1, row), cyclic
};

// LDS staging matrix: entry (row, col), row in [-1, NB), at (row + 1) * P + col.  P = 48: lane rows k, k+1 of an acc-layout
// access (16 consecutive doubles each) fall on disjoint halves of the 64 banks; row -1 and the 16 spare columns are slack
// that out-of-range neighbours of padding entries may touch (kept finite: zeroed once).
constexpr int MATP = 48;

template <int NB>
struct alignas(16) Lds {
  double mat[(NB + 1) * MATP];
  double ubuf[4 * NB];  // sweep: raw pivot rows, one buffer per lane row: [k][tj][j]
  double tsc[16 * 17];  // tile transposes of the 2 x 2 block inverse
  double rrow[2][NB];   // R of the current block, row layout: [m][(i & 3) * KK + (i >> 2)]
  double vrow[2][NB];   // Vr' = R N, row layout
  double vnat[2][NB];   // Vr', natural order
  double e0row[NB];     // E[i][i], row layout
  double e1row[NB];     // E[i][i-1], row layout
  double e0nat[NB];     // natural order
  double e1nat[NB];
  double dgnat[NB];     // diag of D_{j+1}
  double cenat[NB + 2]; // [1 + i] = D_{j+1}[i][i+1]; [0] = 0
};

template <int NB>
__global__ __launch_bounds__(64, HOMMX_FUSED_WAVES_PER_SIMD) void k_poisson2d_fused(
    const double* __restrict__ coef, const double* __restrict__ Mmat, double* __restrict__ out,
    int32_t* __restrict__ info, int n, long long ncells, const CoefSource src
#ifdef HOMMX_FUSED_DEBUG
    , double* __restrict__ dbg
#endif
) {
  constexpr int NT = NB / 16, KK = NB / 4, CG = 64 / NB;
  constexpr int P = MATP;
  __shared__ Lds<NB> L;

  const long long cell = blockIdx.x;
  if (cell >= ncells) return;
  __builtin_assume(n >= 3);  // checked by the host (hommx_plan_create)
  __builtin_assume(n <= NB);
  const int l = threadIdx.x;
  const int j = l & 15, k = l >> 4;
  const int c = l % NB, g = l / NB;  // matrix index of the lane's column, replica
  const int lb = l - c;
  const int p0 = NB - n;             // first real index
  const bool valid = c >= p0;
  const int cn = c - p0;             // node column
  const int cm = valid ? (c == p0 ? NB - 1 : c - 1) : c;   // cyclic left neighbour (matrix index)
  const int cp = valid ? (c == NB - 1 ? p0 : c + 1) : c;   // cyclic right neighbour
  const int prow = (c & 3) * KK + (c >> 2);                // position of index c in a row-layout vector

  // LDS addressing of the staging matrix (in doubles)
  const int ownB = (k + 1) * P + j;  // entry (k, j); element (ti, tj, r): + (16 ti + 4 r) * P + 16 tj
  int leftB[NT];                     // entry (k, cyclic-left of column 16 tj + j)
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) {
    const int col = 16 * tj + j;
    leftB[tj] = (k + 1) * P + (col == p0 ? NB - 1 : col - 1);  // col 0 of the padding reads the slack in front (finite, times 0)
  }
#define OWN(ti, tj, r) (ownB + (16 * (ti) + 4 * (r)) * P + 16 * (tj))
#define LEFT(ti, tj, r) (leftB[tj] + (16 * (ti) + 4 * (r)) * P)
  const unsigned matOff = accl::lds_offset(&L.mat[0]);
  const unsigned haloA = matOff + 8u * (unsigned)(p0 * P + j);  // halo row = row index p0 - 1: copy of row NB - 1

  // zero the staging matrix once
  for (int i = l; i < (NB + 1) * P; i += 64) L.mat[i] = 0.0;
  if (l == 0) L.cenat[0] = 0.0;

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

  // Coefficient source (kernels.h): the element stream coef[cell][2 n^2], or a device sampler fed by two numbers per cell
  const int mode = src.mode;
  const unsigned char* mask = static_cast<const unsigned char*>(src.table);
  const double* table = static_cast<const double*>(src.table);
  const double* cc = mode == COEF_STREAM ? coef + cell * (2ll * n * n) : coef + cell * 2;
  double ph0 = 0.0, ph1 = 0.0;  // phase values / (a, b)
  if (mode != COEF_STREAM) {
    ph0 = cc[0];
    ph1 = cc[1];
  }
  const int nq = src.nq;
  auto sample = [&](long long el) {  // RECIPROCAL: sum_q w_q / (a + b g_q), every operation rounded separately, q ascending
    double acc = 0.0;
    for (int q = 0; q < nq; ++q)
      acc = add_rn(acc, mul_rn(src.weights[q], div_rn(1.0, add_rn(ph0, mul_rn(ph1, table[el * nq + q])))));
    return acc;
  };
  auto load_row = [&](int jc) {
    CoefRow r;
    r.a0 = 0.0; r.a1 = 0.0;
    if (valid) {
      const long long el = 2 * (jc * n + cn);
      if (mode == COEF_TWO_PHASE) {
        const uchar2 mk = *reinterpret_cast<const uchar2*>(mask + el);
        r.a0 = mk.x ? ph1 : ph0;
        r.a1 = mk.y ? ph1 : ph0;
      } else if (mode == COEF_AFFINE) {
        const double2 g2 = *reinterpret_cast<const double2*>(table + el);
        r.a0 = add_rn(ph0, mul_rn(ph1, g2.x));
        r.a1 = add_rn(ph0, mul_rn(ph1, g2.y));
      } else if (mode == COEF_RECIPROCAL) {
        r.a0 = sample(el);
        r.a1 = sample(el + 1);
      } else {
        const double2 v = *reinterpret_cast<const double2*>(cc + el);
        r.a0 = v.x; r.a1 = v.y;
      }
    }
    r.a0m = __shfl(r.a0, lb + cm, 64);
    r.a1m = __shfl(r.a1, lb + cm, 64);
    return r;
  };
  // stencil of node row jr from cell rows jr (cur, above the nodes) and jr-1 (prev, below); padding: identity, no coupling
  auto st_diag = [&](const CoefRow& cur, const CoefRow& prev) {
    return valid ? cur.a0 * al + cur.a1 * be + cur.a0m * ab + prev.a0m * be + prev.a1m * al + prev.a1 * ab : 1.0;
  };
  auto st_E = [&](const CoefRow& cur, const CoefRow& prev) { return (cur.a0 + prev.a1) * (ga - al); };   // c <-> c+1
  auto st_N = [&](const CoefRow& cur) { return (cur.a1 + cur.a0m) * (ga - be); };                         // (c, jr+1) <- (c, jr)
  auto st_NE = [&](const CoefRow& cur) { return -ga * (cur.a0 + cur.a1); };                               // (c+1, jr+1) <- (c, jr)
  auto st_p0 = [&](const CoefRow& cur, const CoefRow& prev) { return cur.a0 - cur.a0m - prev.a1m + prev.a1; };
  auto st_p1 = [&](const CoefRow& cur, const CoefRow& prev) { return cur.a1 + cur.a0m - prev.a0m - prev.a1; };
  // (coefficients of invalid lanes are 0, so every stencil entry but the diagonal vanishes on the padding)

  // band matrices through the staging matrix: write the few non-zeros, read in acc layout, clear them again
  //   sym == true :  X[c][c] = dv, X[cp][c] = X[c][cp] = ov          (D blocks)
  //   sym == false:  X[c][c] = dv, X[cp][c] = ov                     (E: row cp couples to column c)
  auto band_write = [&](double dv, double ov, bool sym, bool clear) {
    if (g == 0) {
      const double d_ = clear ? 0.0 : dv, o_ = clear ? 0.0 : ov;
      L.mat[(c + 1) * P + c] = d_;
      if (valid) {
        L.mat[(cp + 1) * P + c] = o_;
        if (sym) L.mat[(c + 1) * P + cp] = o_;
      }
    }
  };

  // ---- prologue: rows n-2, n-1, 0 ---------------------------------------------------------------
  double wf[NT][KK];          // W, operand layout
  double a[NT][NT][4];        // T = -S / N, acc layout
  d4 slr[NT][NT];             // S_last accumulators (lower tiles), acc layout
  double rr[2];               // R of the current block at column c
  double rlm;                 // R_last[k >> 1] at column c
  CoefRow cur;
  double asum;                // C0 = int_Y A accumulated while the coefficient lines stream by (every line exactly once)
  __syncthreads();
  {
    const CoefRow rowA = load_row(n - 2);
    const CoefRow rowB = load_row(n - 1);
    cur = load_row(0);
    asum = (rowB.a0 + rowB.a1) + (cur.a0 + cur.a1);
    // S_last = D_{n-1}
    {
      const double dv = st_diag(rowB, rowA), ov = st_E(rowB, rowA);
      band_write(dv, ov, true, false);
      __syncthreads();
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) slr[ti][tj][r] = L.mat[OWN(ti, tj, r)];
      __syncthreads();
      band_write(dv, ov, true, true);
      __syncthreads();
    }
    // W_0 = K[(., n-1), (., 0)] = E_{n-1}^T: operand layout of W = acc layout of W^T = E_{n-1}
    {
      const double dv = st_N(rowB), ov = st_NE(rowB);
      band_write(dv, ov, false, false);
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wf[t][kk] = L.mat[OWN(kk >> 2, t, kk & 3)];
      __syncthreads();
      band_write(dv, ov, false, true);
      __syncthreads();
    }
    // T_0 = -D_0
    {
      const double dv = -st_diag(cur, rowB), ov = -st_E(cur, rowB);
      band_write(dv, ov, true, false);
      __syncthreads();
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) a[ti][tj][r] = L.mat[OWN(ti, tj, r)];
      __syncthreads();
      band_write(dv, ov, true, true);
      __syncthreads();
    }
    rr[0] = st_p0(cur, rowB); rr[1] = st_p1(cur, rowB);                       // R_0
    rlm = (k >> 1) ? st_p1(rowB, rowA) : st_p0(rowB, rowA);                   // R_last
  }
  double ga_ = 0.0, gb_ = 0.0;  // -G partial sums: lane rows 0,1: (G00, G01); lane rows 2,3: (G10, G11)
  int bad = 0, badstep = 0;
  // wrap-around entries of D (first real index <-> last index): per-lane selectors
  double wselR[NT];  // 1 on the lane holding T[NB-1][p0] in register (NT-1, tj, 3)
#pragma unroll
  for (int tj = 0; tj < NT; ++tj) wselR[tj] = (k == 3 && 16 * tj + j == p0) ? 1.0 : 0.0;
  const double wselC = (j == 15 && k == (p0 & 3)) ? 1.0 : 0.0;  // lane holding T[p0][NB-1] in register (p0/16, NT-1, (p0%16)/4)
  double indw[KK];  // wave-uniform 0/1: which (tile row, register) of column tile NT-1 holds row p0
#pragma unroll
  for (int q = 0; q < KK; ++q) indw[q] = uniform_f64((q == (p0 >> 2)) ? 1.0 : 0.0);

  // ---- elimination of node rows 0 .. n-2 ----------------------------------------------------------
  for (int jr = 0; jr <= n - 2; ++jr) {
    const bool lastStep = (jr == n - 2);
    // next coefficient line early (latency hidden behind the sweep)
    CoefRow nxt = cur;
    if (!lastStep) {
      nxt = load_row(jr + 1);
      asum += nxt.a0 + nxt.a1;
    }
    // coupling E = K[(., jr+1), (., jr)] from cell row jr:  E[i][i] = cN[i], E[i][i-1] = cNE[i-1]
    const double e0c = st_N(cur);
    const double neC = st_NE(cur);
    const double e1c = __shfl(neC, lb + cm, 64);
    if (lastStep) {
      // the last node row couples to row n-2 through E as well as through the arrow: W += E, i.e. W^T += E^T
      for (int i = l; i < (NB + 1) * P; i += 64) L.mat[i] = 0.0;
      __syncthreads();
      if (g == 0) {
        L.mat[(c + 1) * P + c] = e0c;
        if (valid) L.mat[(c + 1) * P + cp] = neC;  // E^T[c][cp] = E[cp][c]
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) wf[t][kk] += L.mat[OWN(kk >> 2, t, kk & 3)];
      __syncthreads();
    }
    // next block's diagonal data
    const double dgc = st_diag(nxt, cur), cec = st_E(nxt, cur);
    if (g == 0) {
      L.rrow[0][prow] = rr[0];
      L.rrow[1][prow] = rr[1];
      L.e0row[prow] = e0c;
      L.e1row[prow] = e1c;
      L.e0nat[c] = e0c;
      L.e1nat[c] = e1c;
      L.dgnat[c] = dgc;
      L.cenat[1 + c] = cec;
    }

    // (1) N = T^-1
    {
      int badj = 0;
#ifdef HOMMX_FUSED_SWEEP32
      accl::Sweep<NB>::run(a, L.ubuf, j, k, badj);
#else
      // the inverse is one dependent chain (pivot after pivot): let it issue ahead of the other wave of the SIMD, whose MFMA / sparse
      // phases have slack (+2 %; priority everywhere but the MFMA block: no gain)
      __builtin_amdgcn_s_setprio(1);
      if constexpr (NB == 32) accl::block_inverse32(a, L.ubuf, L.tsc, j, k, badj);
      else accl::Sweep<NB>::run(a, L.ubuf, j, k, badj);
      __builtin_amdgcn_s_setprio(0);
#endif
      if (badj && !bad) { bad = 1; badstep = jr + 1; }
    }
#ifdef HOMMX_FUSED_DEBUG
    if (dbg && cell == 0) {
      double* q = dbg + (size_t)jr * 4 * NB * NB;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) q[(16 * ti + 4 * r + k) * NB + 16 * tj + j] = a[ti][tj][r];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) q[NB * NB + (16 * t + j) * NB + 4 * kk + k] = wf[t][kk];
    }
#endif

    if (!lastStep) {
      // (2) N -> staging matrix (for T_next), with the halo copy of row NB-1 in front of the first real row
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) L.mat[OWN(ti, tj, r)] = a[ti][tj][r];
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
        accl::masked_lds_store<accl::RowMask<3>::lo, accl::RowMask<3>::hi>(haloA + 128u * tj, a[NT - 1][tj][3]);
    }

    // (3) V'^T = N W^T  (acc layout of V'^T == V' in operand layout); A operand = N's registers (N symmetric)
    d4 vt[NT][NT];
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
      for (int y = 0; y < NT; ++y) vt[x][y] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y)
          vt[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk >> 2][x][kk & 3], wf[y][kk], vt[x][y], 0, 0, 0);

    // (4) S_last += V' W^T: symmetric, tiles on and below the diagonal only
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y <= x; ++y)
          slr[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(vt[kk >> 2][x][kk & 3], wf[y][kk], slr[x][y], 0, 0, 0);

    // (5) Vr' = R N: partial sums over the lane's rows, transposing butterfly over the four lane rows
    //     -> lane row k holds Vr'[k >> 1][column c]
    double z;
    {
      double part[2][NT];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) part[m][tj] = 0.0;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
          const d4 q = *reinterpret_cast<const d4*>(&L.rrow[m][k * KK + 4 * ti]);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) part[m][tj] = fma(q[r], a[ti][tj][r], part[m][tj]);
        }
      if (NT == 2) {
        double s_[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          double x = part[m][0], y = part[m][NT - 1];
          swap16(x, y);  // x = [x0, y0, x2, y2], y = [x1, y1, x3, y3]
          s_[m] = x + y;
        }
        swap32(s_[0], s_[1]);  // [s0_0, s0_1, s1_0, s1_1], [s0_2, s0_3, s1_2, s1_3]
        z = s_[0] + s_[1];
      } else {
        double x = part[0][0], y = part[1][0];
        swap32(x, y);  // x = [x0, x1, y0, y1], y = [x2, x3, y2, y3]
        z = add_xor16(x + y);
      }
    }
    ga_ = fma(z, rr[0], ga_);
    gb_ = fma(z, rr[1], gb_);
    L.vnat[k >> 1][c] = z;
    L.vrow[k >> 1][prow] = z;

    // (6) R_last += Vr' W^T
    {
      double part[2][NT];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        double v[KK];
#pragma unroll
        for (int q4 = 0; q4 < KK; q4 += 4) {
          const d4 q = *reinterpret_cast<const d4*>(&L.vrow[m][k * KK + q4]);
          v[q4] = q[0]; v[q4 + 1] = q[1]; v[q4 + 2] = q[2]; v[q4 + 3] = q[3];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          double acc = 0.0;
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) acc = fma(v[kk], wf[t][kk], acc);
          part[m][t] = acc;
        }
      }
      double z6;
      if (NT == 2) {
        double s_[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          double x = part[m][0], y = part[m][NT - 1];
          swap16(x, y);
          s_[m] = x + y;
        }
        swap32(s_[0], s_[1]);
        z6 = s_[0] + s_[1];
      } else {
        double x = part[0][0], y = part[1][0];
        swap32(x, y);
        z6 = add_xor16(x + y);
      }
      rlm += z6;
    }

    if (!lastStep) {
      // row-layout coupling coefficients: e0r[kk] = E[i][i], e1r[kk] = E[i][i-1] at i = 4 kk + k = 16 ti + 4 r + k
      double e0r[KK], e1r[KK];
#pragma unroll
      for (int q4 = 0; q4 < KK; q4 += 4) {
        const d4 x = *reinterpret_cast<const d4*>(&L.e0row[k * KK + q4]);
        const d4 y = *reinterpret_cast<const d4*>(&L.e1row[k * KK + q4]);
        e0r[q4] = x[0]; e0r[q4 + 1] = x[1]; e0r[q4 + 2] = x[2]; e0r[q4 + 3] = x[3];
        e1r[q4] = y[0]; e1r[q4 + 1] = y[1]; e1r[q4 + 2] = y[2]; e1r[q4 + 3] = y[3];
      }
      // (8) T_next = -D_{j+1} - E N E^T in place.  X(r, q) = e0[q] N[r][q] + e1[q] N[r][q-1];
      //     (E N E^T)[r][q] = e0[r] X(r, q) + e1[r] X(r-1, q); neighbours from the staging matrix (halo row = wrap-around).
      {
        double e0q[NT], e1q[NT], dgq[NT], ceq[NT], cemq[NT];
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
          e0q[tj] = L.e0nat[16 * tj + j];
          e1q[tj] = L.e1nat[16 * tj + j];
          dgq[tj] = L.dgnat[16 * tj + j];
          ceq[tj] = L.cenat[1 + 16 * tj + j];   // D[i][i+1], i = my column
          cemq[tj] = L.cenat[16 * tj + j];      // D[i-1][i] (no wrap: the slot in front of index 0 is 0)
        }
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tj = 0; tj < NT; ++tj) {
              const double nl = L.mat[LEFT(ti, tj, r)];
              const double nu = L.mat[OWN(ti, tj, r) - P];
              const double nul = L.mat[LEFT(ti, tj, r) - P];
              const double x = fma(e1q[tj], nl, e0q[tj] * a[ti][tj][r]);
              const double xu = fma(e1q[tj], nul, e0q[tj] * nu);
              a[ti][tj][r] = -fma(e0r[4 * ti + r], x, e1r[4 * ti + r] * xu);
            }
        // band of -D_{j+1}: diagonal j = 4 r + k, sub-diagonal (row = col + 1) j = 4 r + k - 1, super-diagonal j = 4 r + k + 1
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          using namespace accl;
#define DIAG_LO(r) (OneLane<0, 4 * (r)>::lo | OneLane<1, 4 * (r) + 1>::lo | OneLane<2, 4 * (r) + 2>::lo | OneLane<3, 4 * (r) + 3>::lo)
#define DIAG_HI(r) (OneLane<0, 4 * (r)>::hi | OneLane<1, 4 * (r) + 1>::hi | OneLane<2, 4 * (r) + 2>::hi | OneLane<3, 4 * (r) + 3>::hi)
          masked_sub4<DIAG_LO(0), DIAG_HI(0), DIAG_LO(1), DIAG_HI(1), DIAG_LO(2), DIAG_HI(2), DIAG_LO(3), DIAG_HI(3)>(
              a[t][t][0], a[t][t][1], a[t][t][2], a[t][t][3], dgq[t]);
          // sub-diagonal: lanes (j = 4 r + k - 1, k); for r = 0 lane row 0 has no such lane in this tile
#define SUB_LO(r) (((r) ? OneLane<0, (4 * (r) - 1) & 15>::lo : 0u) | OneLane<1, 4 * (r)>::lo | OneLane<2, 4 * (r) + 1>::lo | OneLane<3, 4 * (r) + 2>::lo)
#define SUB_HI(r) (((r) ? OneLane<0, (4 * (r) - 1) & 15>::hi : 0u) | OneLane<1, 4 * (r)>::hi | OneLane<2, 4 * (r) + 1>::hi | OneLane<3, 4 * (r) + 2>::hi)
          masked_sub4<SUB_LO(0), SUB_HI(0), SUB_LO(1), SUB_HI(1), SUB_LO(2), SUB_HI(2), SUB_LO(3), SUB_HI(3)>(
              a[t][t][0], a[t][t][1], a[t][t][2], a[t][t][3], ceq[t]);
          // super-diagonal: lanes (j = 4 r + k + 1, k); for r = 3 lane row 3 has no such lane in this tile
#define SUP_LO(r) (OneLane<0, 4 * (r) + 1>::lo | OneLane<1, 4 * (r) + 2>::lo | OneLane<2, 4 * (r) + 3>::lo | ((r) < 3 ? OneLane<3, (4 * (r) + 4) & 15>::lo : 0u))
#define SUP_HI(r) (OneLane<0, 4 * (r) + 1>::hi | OneLane<1, 4 * (r) + 2>::hi | OneLane<2, 4 * (r) + 3>::hi | ((r) < 3 ? OneLane<3, (4 * (r) + 4) & 15>::hi : 0u))
          masked_sub4<SUP_LO(0), SUP_HI(0), SUP_LO(1), SUP_HI(1), SUP_LO(2), SUP_HI(2), SUP_LO(3), SUP_HI(3)>(
              a[t][t][0], a[t][t][1], a[t][t][2], a[t][t][3], cemq[t]);
#undef DIAG_LO
#undef DIAG_HI
#undef SUB_LO
#undef SUB_HI
#undef SUP_LO
#undef SUP_HI
        }
        if (NT == 2) {
          using namespace accl;
          // tile crossings: T[16][15] in (1, 0, r = 0) lane (j = 15, k = 0); T[15][16] in (0, 1, r = 3) lane (j = 0, k = 3)
          masked_sub<OneLane<0, 15>::lo, OneLane<0, 15>::hi>(a[NT - 1][0][0], ceq[0]);
          masked_sub<OneLane<3, 0>::lo, OneLane<3, 0>::hi>(a[0][NT - 1][3], cemq[NT - 1]);
        }
        // wrap-around entries D[p0][NB-1] = D[NB-1][p0] = coupling of the last real column to the first
        {
          const double cw = readlane_f64(cec, NB - 1);
#pragma unroll
          for (int tj = 0; tj < NT; ++tj) a[NT - 1][tj][3] = fma(-cw, wselR[tj], a[NT - 1][tj][3]);
          const double vcol = -cw * wselC;
#pragma unroll
          for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[ti][NT - 1][r] = fma(vcol, indw[4 * ti + r], a[ti][NT - 1][r]);
        }
      }

      // (7) W_next = V' E^T: W_next^T[i][.] = e0[i] V'^T[i][.] + e1[i] V'^T[i-1][.]  (row shift through the staging matrix)
#pragma unroll
      for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) L.mat[OWN(x, y, r)] = vt[x][y][r];
#pragma unroll
      for (int y = 0; y < NT; ++y)
        accl::masked_lds_store<accl::RowMask<3>::lo, accl::RowMask<3>::hi>(haloA + 128u * y, vt[NT - 1][y][3]);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          const double up = L.mat[OWN(kk >> 2, t, kk & 3) - P];
          wf[t][kk] = fma(e1r[kk], up, e0r[kk] * vt[kk >> 2][t][kk & 3]);
        }

      // (9) R_next = P_{j+1} + Vr' E^T
      {
        const double v0 = L.vnat[0][c], v1 = L.vnat[1][c];
        const double vm0 = L.vnat[0][cm], vm1 = L.vnat[1][cm];
        rr[0] = fma(vm0, e1c, fma(v0, e0c, st_p0(nxt, cur)));
        rr[1] = fma(vm1, e1c, fma(v1, e0c, st_p1(nxt, cur)));
      }
      cur = nxt;
      __syncthreads();
    }
  }

  // ---- last node row: T_last = -S_last, pin node (n-1, n-1), sweep ------------------------------------
  {
    // upper tile = transpose of the lower one, through the staging matrix
    if (NT == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) L.mat[OWN(1, 0, r)] = slr[NT - 1][0][r];
      __syncthreads();
    }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (tj <= ti) a[ti][tj][r] = -slr[ti][tj][r];
          else a[ti][tj][r] = -L.mat[(16 + j + 1) * P + 4 * r + k];  // S[4r+k][16+j] = S[16+j][4r+k]
        }
    // gauge: drop the last unknown (cell_problem.py:349-361): row / column NB-1 <- 0, diagonal <- -1
    {
      using namespace accl;
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) amov<RowMask<3>::lo, RowMask<3>::hi>(a[NT - 1][tj][3], 0.0);
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) amov<ColMask<15>::lo, ColMask<15>::hi>(a[ti][NT - 1][r], 0.0);
      amov<OneLane<3, 15>::lo, OneLane<3, 15>::hi>(a[NT - 1][NT - 1][3], -1.0);
    }
    if (c == NB - 1) rlm = 0.0;
    __syncthreads();
    L.rrow[k >> 1][prow] = rlm;
    L.vnat[k >> 1][c] = rlm;  // natural copy of R_last (vnat is free now)
    __syncthreads();
    int badj = 0;
    accl::Sweep<NB>::run(a, L.ubuf, j, k, badj);
    if (badj && !bad) { bad = 1; badstep = n; }
    double part[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) part[m][tj] = 0.0;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const d4 q = *reinterpret_cast<const d4*>(&L.rrow[m][k * KK + 4 * ti]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int tj = 0; tj < NT; ++tj) part[m][tj] = fma(q[r], a[ti][tj][r], part[m][tj]);
      }
    double z;
    if (NT == 2) {
      double s_[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        double x = part[m][0], y = part[m][NT - 1];
        swap16(x, y);
        s_[m] = x + y;
      }
      swap32(s_[0], s_[1]);
      z = s_[0] + s_[1];
    } else {
      double x = part[0][0], y = part[1][0];
      swap32(x, y);
      z = add_xor16(x + y);
    }
    ga_ = fma(z, L.vnat[0][c], ga_);
    gb_ = fma(z, L.vnat[1][c], gb_);
  }

  // ---- K3: wave reduction and output ----------------------------------------------------------------
  double g00 = (k < 2) ? ga_ : 0.0, g01 = (k < 2) ? gb_ : 0.0, g11 = (k < 2) ? 0.0 : gb_;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    g00 += __shfl_xor(g00, off, 64);
    g01 += __shfl_xor(g01, off, 64);
    g11 += __shfl_xor(g11, off, 64);
    asum += __shfl_xor(asum, off, 64);
  }
  if (l == 0) {
    const double h = 1.0 / n;
    constexpr int DUP = 32 / NB;  // every (load row, column) pair sits in DUP lanes
    const double sc = 0.25 * h * h / DUP;
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
#undef OWN
#undef LEFT
}

// ---- launch ---------------------------------------------------------------------------------------
hipError_t launch_poisson2d_fused(const double* d_coef, const double* d_M, double* d_out, int32_t* d_info,
                                  int n, long long ncells, hipStream_t stream, CoefSource src) {
  if (ncells <= 0) return hipSuccess;
  dim3 grid((unsigned)ncells), block(64);
#ifndef HOMMX_DEV_LDS_PAD
#define HOMMX_DEV_LDS_PAD 0  // dev builds: dynamic LDS bytes per workgroup, to pin the occupancy for latency experiments
#endif
#ifdef HOMMX_FUSED_DEBUG
  extern double* g_fused_dbg;
  if (n <= 16)
    hipLaunchKernelGGL(k_poisson2d_fused<16>, grid, block, 0, stream, d_coef, d_M, d_out, d_info, n, ncells, src, g_fused_dbg);
  else
    hipLaunchKernelGGL(k_poisson2d_fused<32>, grid, block, 0, stream, d_coef, d_M, d_out, d_info, n, ncells, src, g_fused_dbg);
#else
  if (n <= 16)
    hipLaunchKernelGGL(k_poisson2d_fused<16>, grid, block, HOMMX_DEV_LDS_PAD, stream, d_coef, d_M, d_out, d_info, n, ncells, src);
  else
    hipLaunchKernelGGL(k_poisson2d_fused<32>, grid, block, HOMMX_DEV_LDS_PAD, stream, d_coef, d_M, d_out, d_info, n, ncells, src);
#endif
  return hipGetLastError();
}

}  // namespace hommx
