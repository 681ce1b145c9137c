// small_wave.h -- register-resident elimination for plane blocks b = bs * n^(d-1) <= 48: ONE WAVEFRONT PER MACRO CELL.
//
// For the sizes of the reference's own tests (2D elasticity on 10 x 10 micro cells, b = 20: test_integration_linear_elasticity.py:62-171;
// 3D Poisson on 6^3, b = 36: test_integration_poisson.py:243-294; matrix-valued 2D Poisson) every matrix of the block-cyclic
// elimination (same recurrences as blocked_solve) fits in the register file of one wave in the layout of the f64 MFMA accumulator,
//
//     lane l = 16 k + j,   m[ti][tj][r] = M[16 ti + 4 r + k][16 tj + j]        (NT x NT tiles, BP = 16 NT = 16 / 32 / 48)
//
// and in that layout register r of tile row ti IS k-slab 4 ti + r of M as an MFMA operand:  B(k, j) = M[k][16 tj + j]  and
// A(i, k) = M[k][16 ti + i].  Every product of the elimination has the form  OUT += P^T Q  on such slabs, so accumulators feed
// the next product directly -- no LDS round trip, no barrier, nothing for a second wave to wait for.  fp64 MFMA and fp64 FMA have the
// same peak on gfx950 (one 16x16x4 takes 16 passes), so the kernel is bound by the NUMBER of MFMAs; three things keep it low:
//
//   * bordered arrow (AUG, when b + t <= BP):  A = [ -W^T | -R^T ]  carries the t load rows in the padding columns of the arrow,
//     SLa = [[S_last, R_last^T], [., -G]]  is ONE symmetric matrix, and the four narrow products of the load rows disappear:
//         N = T^-1 ;  nv = N A ;  SLa += nv^T A ;  A' = [0 | -P^T] + ET^T nv ;  nz = N ET ;  T' = -D + (-nz)^T ET
//     (T = -S carried negated: the exchange sweeps of sweep_acc.h work in place and skip the pivots of the identity padding;
//      32 x 32: accl::block_inverse32, two 16-sweeps + 16 MFMAs; BP = 48: one more 2 x 2 level around that and a 16-sweep)
//   * k-slabs behind row b hold zeros: each product is compiled for 4 NT - 3 .. 4 NT slabs and the wave picks its variant
//   * SLa is symmetric and lives in LDS: only its upper tiles are updated
//
// Without the border (b + t > BP) the load rows are BP x 16 slab columns (RT, RLN) with their own products.
// LDS holds SLa (read and written once per step), a BP x BP scatter pad -- the <= 27 stencil entries of every row of E (and D) are
// written dense there, transposed, read back as tiles and erased again (symmetric D: the transposed scatter is conflict-free, the plain
// one is not) -- and the pivot-row buffers of the sweeps.  Occupancy is bounded by registers: 2 waves per SIMD (8 cells per CU) for BP <= 32.
// Correctors are not formed here: hommx_solve_batch_correctors stays on the HBM-resident route.
#pragma once

#include <type_traits>

#include "geo.h"
#include "small_fused.h"  // swz<BP>
#include "sweep_acc.h"

namespace hommx {

// run CALL(K) for the compile-time slab count K = number of k-slabs that hold rows < b
#define HOMMX_SLABS(CALL)                                          \
  switch (NS - KS) {                                               \
    case 0: CALL(NS); break;                                       \
    case 1: CALL((NS - 1)); break;                                 \
    case 2: CALL((NS - 2)); break;                                 \
    default: CALL((NS - 3)); break;                                \
  }

template <int NT, int BSV, int NIPC, bool AUG>
__global__ __launch_bounds__(64, (NT == 3 ? 1 : 2)) void k_small_wave(Geo G, const double* __restrict__ Kst, const double* __restrict__ Brhs,
                                                                     const double* __restrict__ C0, double* __restrict__ out,
                                                                     int32_t* __restrict__ info, long long ncells) {
  constexpr int BP = 16 * NT, NS = 4 * NT, NE = NIPC * BSV;
  constexpr int TM = BSV == 1 ? 3 : BSV == 2 ? 3 : 6;  // t <= TM
  typedef double Mat[NT][NT][4];
  typedef double Col[NT][4];  // BP x 16 matrix (load rows, transposed): columns >= t are zero
  __shared__ double pad[BP * BP];
  __shared__ double SLm[BP * BP];  // S_last (AUG: bordered with R_last^T and -G): touched once per step, upper tiles only
  __shared__ double ubuf[4 * 32];
  __shared__ double tsc[NT >= 2 ? 16 * 17 : 1];  // tile transposes of the 32 x 32 block inverse

  const long long cell = blockIdx.x;
  if (cell >= ncells) return;
  const int lane = threadIdx.x, lj = lane & 15, lk = lane >> 4;
  const int n = G.n, b = G.b, t = G.t, npl = G.npl, nn = G.nn;
  const int KS = (b + 3) >> 2;  // NS - 3 <= KS <= NS
  const double* Kc = Kst + cell * (long long)G.ncode * BSV * BSV * nn;
  const double* Bc = Brhs + cell * (long long)t * BSV * nn;

  // ---- stencil rows: lane ec < b owns row ec of the plane block (node ec / BSV, component ec % BSV) -----------------------------------
  const int ec = lane;
  const bool realrow = ec < b;
  int ex[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) ex[e] = 0;
  if (realrow) {
#pragma unroll
    for (int ipc = 0; ipc < NIPC; ++ipc) {
      const int qn = plane_neighbour(G, ec / BSV, ipc);
#pragma unroll
      for (int be = 0; be < BSV; ++be) ex[ipc * BSV + be] = swz<BP>(qn * BSV + be, ec);  // pad index of (column, row): transposed
    }
  }
  auto fetch_row = [&](double (&dst)[NE], int pl, int o) {  // row ec of K[(., pl), (., pl + o)]
#pragma unroll
    for (int e = 0; e < NE; ++e) dst[e] = 0.0;
    if (realrow) {
#pragma unroll
      for (int ipc = 0; ipc < NIPC; ++ipc)
#pragma unroll
        for (int be = 0; be < BSV; ++be) {
          const int code = ipc + (o + 1) * NIPC;
          dst[ipc * BSV + be] = Kc[(unsigned)(((code * BSV + ec % BSV) * BSV + be) * nn + ec / BSV + npl * pl)];
        }
    }
  };
  // load rows of plane pl.  Border form: lane ec holds its row, P^T[ec][m], m < t
  auto fetch_Prow = [&](double (&dst)[TM], int pl) {
#pragma unroll
    for (int m = 0; m < TM; ++m) dst[m] = (realrow && m < t) ? Bc[(unsigned)((m * BSV + ec % BSV) * nn + ec / BSV + npl * pl)] : 0.0;
  };
  // slab form: dst[ti][r] = sign * P^T[16 ti + 4 r + lk][lj]
  auto fetch_P = [&](Col& dst, int pl, double sign) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * ti + 4 * r + lk;
        dst[ti][r] = (c < b && lj < t) ? sign * Bc[(unsigned)((lj * BSV + c % BSV) * nn + c / BSV + npl * pl)] : 0.0;
      }
  };
  // Scatter pad: dense, transposed image of sign * (stencil rows v) in columns < b, and (border) psign * (load rows pr) in columns
  // b .. b + t - 1.  Several stencil codes can hit one neighbour on tiny meshes: accumulate.  diag != 0: identity padding of rows >= b.
  auto pad_write = [&](const double (&v)[NE], double sign, double diag, const double* pr, double psign) {
    if (realrow) {
#pragma unroll
      for (int e = 0; e < NE; ++e) pad[ex[e]] += sign * v[e];
      if (pr) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
          if (m < t) pad[swz<BP>(ec, b + m)] = psign * pr[m];
      }
    } else if (ec < BP && diag != 0.0) {
      pad[swz<BP>(ec, ec)] = diag;
    }
    __syncthreads();
  };
  auto pad_erase = [&](bool border) {
    __syncthreads();
    if (realrow) {
#pragma unroll
      for (int e = 0; e < NE; ++e) pad[ex[e]] = 0.0;
      if (border) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
          if (m < t) pad[swz<BP>(ec, b + m)] = 0.0;
      }
    } else if (ec < BP) {
      pad[swz<BP>(ec, ec)] = 0.0;
    }
    __syncthreads();
  };
  // tiles of the pad; part = 0: everything, 1: columns < b only, 2: columns >= b only
  auto pad_read = [&](Mat& M, int part) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const bool keep = part == 0 || (part == 1 ? 16 * tj + lj < b : 16 * tj + lj >= b);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double v = pad[swz<BP>(16 * ti + 4 * r + lk, 16 * tj + lj)];
          M[ti][tj][r] = keep ? v : 0.0;
        }
      }
  };
  auto scatter_load = [&](Mat& M, const double (&v)[NE], double sign, double diag) {
    pad_write(v, sign, diag, nullptr, 0.0);
    pad_read(M, 0);
    pad_erase(false);
  };

  // ---- products on the matrix cores, operands and results in registers --------------------------------------------------------------------
  auto mfma = [](double a, double bq, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, c, 0, 0, 0); };
  // OUT += P^T Q over the first KN k-slabs
  auto prod_n = [&](auto kn, Mat& OUT, const Mat& P, const Mat& Q) {
    constexpr int KN = decltype(kn)::value;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        d4 c = d4{OUT[ti][tj][0], OUT[ti][tj][1], OUT[ti][tj][2], OUT[ti][tj][3]};
#pragma unroll
        for (int kk = 0; kk < KN; ++kk) c = mfma(P[kk / 4][ti][kk % 4], Q[kk / 4][tj][kk % 4], c);
#pragma unroll
        for (int r = 0; r < 4; ++r) OUT[ti][tj][r] = c[r];
      }
  };
  auto prod = [&](Mat& OUT, const Mat& P, const Mat& Q) {
#define HOMMX_CALL(K) prod_n(std::integral_constant<int, K>{}, OUT, P, Q)
    HOMMX_SLABS(HOMMX_CALL)
#undef HOMMX_CALL
  };
  auto tile_at = [&](int ti, int tj, int r) { return swz<BP>(16 * ti + 4 * r + lk, 16 * tj + lj); };
  // upper tiles of SLm += P^T Q
  auto prod_sl_n = [&](auto kn, const Mat& P, const Mat& Q) {
    constexpr int KN = decltype(kn)::value;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = ti; tj < NT; ++tj) {
        d4 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = SLm[tile_at(ti, tj, r)];
#pragma unroll
        for (int kk = 0; kk < KN; ++kk) c = mfma(P[kk / 4][ti][kk % 4], Q[kk / 4][tj][kk % 4], c);
#pragma unroll
        for (int r = 0; r < 4; ++r) SLm[tile_at(ti, tj, r)] = c[r];
      }
  };
  auto prod_sl = [&](const Mat& P, const Mat& Q) {
#define HOMMX_CALL(K) prod_sl_n(std::integral_constant<int, K>{}, P, Q)
    HOMMX_SLABS(HOMMX_CALL)
#undef HOMMX_CALL
  };
  // OUT += P^T q  (q, OUT: BP x 16)
  auto prod_col_n = [&](auto kn, Col& OUT, const Mat& P, const Col& q) {
    constexpr int KN = decltype(kn)::value;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      d4 c = d4{OUT[ti][0], OUT[ti][1], OUT[ti][2], OUT[ti][3]};
#pragma unroll
      for (int kk = 0; kk < KN; ++kk) c = mfma(P[kk / 4][ti][kk % 4], q[kk / 4][kk % 4], c);
#pragma unroll
      for (int r = 0; r < 4; ++r) OUT[ti][r] = c[r];
    }
  };
  auto prod_col = [&](Col& OUT, const Mat& P, const Col& q) {
#define HOMMX_CALL(K) prod_col_n(std::integral_constant<int, K>{}, OUT, P, q)
    HOMMX_SLABS(HOMMX_CALL)
#undef HOMMX_CALL
  };
  // g += p^T q  (16 x 16)
  auto dot_col = [&](d4 c, const Col& p, const Col& q) {
#pragma unroll
    for (int kk = 0; kk < NS; ++kk) c = mfma(p[kk / 4][kk % 4], q[kk / 4][kk % 4], c);
    return c;
  };
  auto zero_mat = [](Mat& M) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) M[ti][tj][r] = 0.0;
  };
  auto zero_col = [](Col& c) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[ti][r] = 0.0;
  };

  // ---- T <- T^-1 for T = -S, S SPD with an identity padding behind row b ----------------------------------------------------------------------
  auto invert = [&](Mat& T, int& bad) {
    __builtin_amdgcn_s_setprio(1);  // one dependent chain: issue ahead of the SIMD's other wave (see fused2d.hip)
    if constexpr (NT == 1) {
      accl::Sweep<16>::run(T, ubuf, lj, lk, bad, b);
    } else if constexpr (NT == 2) {
      accl::block_inverse32(T, ubuf, tsc, lj, lk, bad, b);
    } else {
      // T = [[TA, U], [U^T, TC]], TA 32 x 32, TC 16 x 16:  Ai = TA^-1,  X = U^T Ai,  Sc = TC - X U,  N22 = Sc^-1,  N21 = -N22 X,
      // N12 = N21^T,  N11 = Ai - X^T N21.   Held: nx = -X (16 x 32, tiles nx[tj]),  nxt = -X^T (32 x 16, tiles nxt[ti]).
      double a[2][2][4], nu[2][4], nx[2][4], nxt[2][4], sc[1][1][4], n21[2][4], n12[2][4];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          nu[ti][r] = -T[ti][2][r];
#pragma unroll
          for (int tj = 0; tj < 2; ++tj) a[ti][tj][r] = T[ti][tj][r];
        }
      accl::block_inverse32(a, ubuf, tsc, lj, lk, bad, 32);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        d4 cx = d4{0.0, 0.0, 0.0, 0.0}, ct = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          cx = mfma(nu[kk / 4][kk % 4], a[kk / 4][q][kk % 4], cx);  // -X[i][16 q + j] = sum_k -U[k][i] Ai[k][16 q + j]
          ct = mfma(a[kk / 4][q][kk % 4], nu[kk / 4][kk % 4], ct);  // -X^T[16 q + i][j] = sum_k Ai[k][16 q + i] (-U[k][j])
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { nx[q][r] = cx[r]; nxt[q][r] = ct[r]; }
      }
      {
        d4 c = d4{T[2][2][0], T[2][2][1], T[2][2][2], T[2][2][3]};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) c = mfma(nxt[kk / 4][kk % 4], T[kk / 4][2][kk % 4], c);  // Sc = TC + (-X) U
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[0][0][r] = c[r];
      }
      accl::Sweep<16>::run(sc, ubuf, lj, lk, bad, b - 32);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        d4 c1 = d4{0.0, 0.0, 0.0, 0.0}, c2 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          c1 = mfma(sc[0][0][kk], nx[q][kk], c1);  // N21[i][16 q + j] = sum_k N22[k][i] (-X)[k][16 q + j]
          c2 = mfma(nx[q][kk], sc[0][0][kk], c2);  // N12[16 q + i][j] = sum_k (-X)[k][16 q + i] N22[k][j]
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { n21[q][r] = c1[r]; n12[q][r] = c2[r]; }
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          d4 c = d4{a[ti][tj][0], a[ti][tj][1], a[ti][tj][2], a[ti][tj][3]};
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) c = mfma(nx[ti][kk], n21[tj][kk], c);  // N11 = Ai + (-X)^T N21
#pragma unroll
          for (int r = 0; r < 4; ++r) T[ti][tj][r] = c[r];
        }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          T[2][q][r] = n21[q][r];
          T[q][2][r] = n12[q][r];
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) T[2][2][r] = sc[0][0][r];
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: every global load is issued before the first use ---------------------------------------------------------------------------
  for (int i = lane; i < BP * BP; i += 64) pad[i] = 0.0;
  Mat T, A;     // T = -S ;  A = -W^T, bordered with -R^T (AUG)
  Col RT, RLN;  // slab form of the load rows (!AUG): R^T, -R_last^T
  d4 gn = d4{0.0, 0.0, 0.0, 0.0};  // -G (!AUG)
  {
    double d0[NE], w0[NE], dl[NE], p0[TM], pl[TM];
    fetch_row(d0, 0, 0);       // D_0
    fetch_row(w0, n - 1, +1);  // K[(., n-1), (., 0)]
    fetch_row(dl, n - 1, 0);   // D_{n-1}
    if constexpr (AUG) {
      fetch_Prow(p0, 0);
      fetch_Prow(pl, n - 1);
    } else {
      fetch_P(RT, 0, 1.0);
      fetch_P(RLN, n - 1, -1.0);
    }
    __syncthreads();
    pad_write(dl, 1.0, 0.0, AUG ? pl : nullptr, 1.0);  // S_last = D_{n-1}, bordered with R_last^T = P^T_{n-1}
    pad_read(T, 0);
    pad_erase(AUG);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) SLm[tile_at(ti, tj, r)] = T[ti][tj][r];
    scatter_load(T, d0, -1.0, -1.0);  // T = -D_0
    pad_write(w0, -1.0, 0.0, AUG ? p0 : nullptr, -1.0);  // A = [-W^T | -P^T_0]
    pad_read(A, 0);
    pad_erase(AUG);
  }

  int firstbad = 0;
  // first half of an elimination step: inverse, arrow, S_last, load rows.  Leaves nv = N A and (!AUG) nvr = -Vr^T.
  auto eliminate = [&](Mat& nv, Col& nvr, int stepcode) {
    int bad = 0;
    invert(T, bad);  // T = N = -Sinv
    if (bad && !firstbad) firstbad = stepcode;
    zero_mat(nv);
    prod(nv, T, A);   // [V^T | Vr^T] = N A            (N symmetric)
    prod_sl(nv, A);   // S_last += V (-W^T) ;  R_last^T += V (-R^T) ;  -G += Vr (-R^T)
    if constexpr (!AUG) {
      zero_col(nvr);
      prod_col(nvr, T, RT);      // -Vr^T = N R^T
      gn = dot_col(gn, nvr, RT); // -G += nvr^T R^T
      prod_col(RLN, A, nvr);     // -R_last^T += (-W) (-Vr^T)
    }
  };

  for (int jp = 0; jp < n - 2; ++jp) {
    double ev[NE], dv[NE], pr[TM];
    Col pnext;
    fetch_row(ev, jp + 1, -1);  // next plane's stencil rows: in flight during the first half of the step
    fetch_row(dv, jp + 1, 0);
    if constexpr (AUG) fetch_Prow(pr, jp + 1);
    else fetch_P(pnext, jp + 1, 1.0);
    Mat nv;
    Col nvr;
    eliminate(nv, nvr, jp + 1);
    Mat ET;
    pad_write(ev, 1.0, 0.0, AUG ? pr : nullptr, -1.0);
    pad_read(ET, AUG ? 1 : 0);
    if constexpr (AUG) {
      pad_read(A, 2);  // A_next = [0 | -P^T_{j+1}] ...
    } else {
      zero_mat(A);
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r) RT[ti][r] = pnext[ti][r];
      prod_col(RT, ET, nvr);  // R^T_next = P^T - E Vr^T
    }
    pad_erase(AUG);
    prod(A, ET, nv);   //          ... + E [V^T | Vr^T]
    zero_mat(nv);
    prod(nv, T, ET);   // N E^T = -Z^T
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) nv[ti][tj][r] = -nv[ti][tj][r];
    scatter_load(T, dv, -1.0, -1.0);  // T_next = -D_{j+1} ...
    prod(T, nv, ET);                  //          ... + Z E^T
  }
  {  // plane n-2: the last plane couples to it through E as well, K[(., n-1), (., n-2)] joins the arrow
    double el[NE];
    fetch_row(el, n - 1, -1);
    Mat nv;
    scatter_load(nv, el, -1.0, 0.0);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) A[ti][tj][r] += nv[ti][tj][r];
    Col nvr;
    eliminate(nv, nvr, n - 1);
  }

  // ---- last plane: gauge (drop the bs unknowns of the last node), inverse, loads --------------------------------------------------------------
  __syncthreads();
  const int bg = b - BSV;  // rows / columns >= bg: pinned unknowns, border and padding -> identity
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ti + 4 * r + lk;
      if constexpr (AUG) RLN[ti][r] = (row < bg && lj < t) ? -SLm[swz<BP>(row, b + lj)] : 0.0;
      else if (row >= bg) RLN[ti][r] = 0.0;
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const int col = 16 * tj + lj;
        const double v = ti <= tj ? SLm[swz<BP>(row, col)] : SLm[swz<BP>(col, row)];
        T[ti][tj][r] = (row >= bg || col >= bg) ? (row == col ? -1.0 : 0.0) : -v;
      }
    }
  {
    int bad = 0;
    invert(T, bad);
    if (bad && !firstbad) firstbad = n;
    Col nvr;
    zero_col(nvr);
    prod_col(nvr, T, RLN);  // Vr_last^T = N (-R_last^T)
    const d4 c = dot_col(gn, nvr, RLN);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * r + lk, q = lj;
      if (m < t && q < t) {
        double corner = 0.0;  // -G of the eliminated planes
        if constexpr (AUG) corner = m <= q ? SLm[swz<BP>(b + m, b + q)] : SLm[swz<BP>(b + q, b + m)];
        out[cell * t * t + m * t + q] = C0[cell * t * t + m * t + q] + corner + c[r];
      }
    }
    if (lane == 0 && info) info[cell] = firstbad;
  }
}

#undef HOMMX_SLABS

}  // namespace hommx
