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
// the next product directly -- no LDS round trip, no barrier, nothing for a second wave to wait for:
//
//     N    = T^-1                 T = -S carried negated: exchange sweeps of sweep_acc.h in place (pivots >= b skipped);
//                                 BP = 48: 2 x 2 block inverse around a 32- and a 16-sweep, its Schur products on the matrix cores
//     nv   = N WN  ( = V^T )      WN = -W^T, the arrow, carried negated and transposed
//     SL  += nv^T WN              S_last -= V W^T
//     nvr  = N RT ( = -Vr^T ) ;  GN += nvr^T RT  ( = -G ) ;  RLN += WN^T nvr  ( = -R_last^T )        load rows, BP x 16 slabs
//     RT'  = P^T + ET^T nvr ;  WN' = ET^T nv ;  nz = N ET ;  T' = -D + (-nz)^T ET                    E = coupling to the next plane
//
// LDS holds S_last (read and written once per step) and a BP x BP scatter pad: the <= 27 stencil entries of every row of E (and D) are written dense there, transposed,
// read back as tiles and erased again (symmetric D: the transposed scatter is conflict-free, the plain one is not), plus the
// pivot-row buffers of the sweeps.  Occupancy is bounded by registers, not LDS: 2 waves per SIMD (8 cells per CU) for BP <= 32.
// Correctors are not formed here: hommx_solve_batch_correctors stays on the HBM-resident route.
#pragma once

#include "geo.h"
#include "small_fused.h"  // swz<BP>
#include "sweep_acc.h"

namespace hommx {

template <int NT, int BSV, int NIPC>
__global__ __launch_bounds__(64, (NT == 3 ? 1 : 2)) void k_small_wave(Geo G, const double* __restrict__ Kst, const double* __restrict__ Brhs,
                                                                     const double* __restrict__ C0, double* __restrict__ out,
                                                                     int32_t* __restrict__ info, long long ncells) {
  constexpr int BP = 16 * NT, NS = 4 * NT, NE = NIPC * BSV;
  typedef double Mat[NT][NT][4];
  typedef double Col[NT][4];  // BP x 16 matrix (load rows, transposed): columns >= t are zero
  __shared__ double pad[BP * BP];
  __shared__ double SLm[BP * BP];  // S_last: touched once per step
  __shared__ double ubuf[4 * 32];

  const long long cell = blockIdx.x;
  if (cell >= ncells) return;
  const int lane = threadIdx.x, lj = lane & 15, lk = lane >> 4;
  const int n = G.n, b = G.b, t = G.t, npl = G.npl, nn = G.nn;
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
  auto fetch_P = [&](Col& dst, int pl, double sign) {  // sign * P^T of plane pl as slabs: dst[ti][r] = P^T[16 ti + 4 r + lk][lj]
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * ti + 4 * r + lk;
        dst[ti][r] = (c < b && lj < t) ? sign * Bc[(unsigned)((lj * BSV + c % BSV) * nn + c / BSV + npl * pl)] : 0.0;
      }
  };
  // dense, transposed image of sign * (rows v) in the scatter pad -> tiles -> pad erased again.  Several stencil codes can hit one
  // neighbour on tiny meshes: accumulate.  diag != 0: identity padding of rows >= b.
  auto scatter_load = [&](Mat& M, const double (&v)[NE], double sign, double diag) {
    if (realrow) {
#pragma unroll
      for (int e = 0; e < NE; ++e) pad[ex[e]] += sign * v[e];
    } else if (ec < BP && diag != 0.0) {
      pad[swz<BP>(ec, ec)] = diag;
    }
    __syncthreads();
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) M[ti][tj][r] = pad[swz<BP>(16 * ti + 4 * r + lk, 16 * tj + lj)];
    __syncthreads();
    if (realrow) {
#pragma unroll
      for (int e = 0; e < NE; ++e) pad[ex[e]] = 0.0;
    } else if (ec < BP) {
      pad[swz<BP>(ec, ec)] = 0.0;
    }
    __syncthreads();
  };

  // ---- products on the matrix cores, operands and results in registers --------------------------------------------------------------------
  auto mfma = [](double a, double bq, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, c, 0, 0, 0); };
  // OUT += P^T Q
  auto prod = [&](Mat& OUT, const Mat& P, const Mat& Q) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        d4 c = d4{OUT[ti][tj][0], OUT[ti][tj][1], OUT[ti][tj][2], OUT[ti][tj][3]};
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) c = mfma(P[kk / 4][ti][kk % 4], Q[kk / 4][tj][kk % 4], c);
#pragma unroll
        for (int r = 0; r < 4; ++r) OUT[ti][tj][r] = c[r];
      }
  };
  // OUT += P^T q  (q, OUT: BP x 16)
  auto prod_col = [&](Col& OUT, const Mat& P, const Col& q) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      d4 c = d4{OUT[ti][0], OUT[ti][1], OUT[ti][2], OUT[ti][3]};
#pragma unroll
      for (int kk = 0; kk < NS; ++kk) c = mfma(P[kk / 4][ti][kk % 4], q[kk / 4][kk % 4], c);
#pragma unroll
      for (int r = 0; r < 4; ++r) OUT[ti][r] = c[r];
    }
  };
  auto zero_mat = [](Mat& M) {
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) M[ti][tj][r] = 0.0;
  };

  // ---- T <- T^-1 for T = -S, S SPD with an identity padding behind row b ----------------------------------------------------------------------
  auto invert = [&](Mat& T, int& bad) {
    if constexpr (NT == 1) {
      accl::Sweep<16>::run(T, ubuf, lj, lk, bad, b);
    } else if constexpr (NT == 2) {
      accl::Sweep<32>::run(T, ubuf, lj, lk, bad, b);
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
      accl::Sweep<32>::run(a, ubuf, lj, lk, bad, 32);
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
  };

  // S_last in LDS: tile (ti, tj) of SLm (+)= acc
  auto tile_at = [&](int ti, int tj, int r) { return swz<BP>(16 * ti + 4 * r + lk, 16 * tj + lj); };

  // ---- prologue: every global load is issued before the first use ---------------------------------------------------------------------------
  for (int i = lane; i < BP * BP; i += 64) pad[i] = 0.0;
  Mat T, WN;
  Col RT, RLN;
  double gn[4] = {0.0, 0.0, 0.0, 0.0};  // GN = -G
  {
    double d0[NE], w0[NE], dl[NE];
    fetch_row(d0, 0, 0);       // D_0
    fetch_row(w0, n - 1, +1);  // K[(., n-1), (., 0)]
    fetch_row(dl, n - 1, 0);   // D_{n-1}
    fetch_P(RT, 0, 1.0);
    fetch_P(RLN, n - 1, -1.0);
    __syncthreads();
    scatter_load(T, dl, 1.0, 1.0);  // S_last = D_{n-1}
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = 0; tj < NT; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) SLm[tile_at(ti, tj, r)] = T[ti][tj][r];
    scatter_load(T, d0, -1.0, -1.0);  // T = -D_0
    scatter_load(WN, w0, -1.0, 0.0);  // WN = -W^T
  }

  int firstbad = 0;
  // first half of an elimination step: inverse, arrow, S_last, load rows.  Leaves nv = V^T and nvr = -Vr^T.
  auto eliminate = [&](Mat& nv, Col& nvr, int stepcode) {
    int bad = 0;
    invert(T, bad);  // T = N = -Sinv
    if (bad && !firstbad) firstbad = stepcode;
    zero_mat(nv);
    prod(nv, T, WN);  // V^T = N WN            (N symmetric)
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)  // S_last += V WN  ( = -V W^T )
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        d4 c;
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = SLm[tile_at(ti, tj, r)];
#pragma unroll
        for (int kk = 0; kk < NS; ++kk) c = mfma(nv[kk / 4][ti][kk % 4], WN[kk / 4][tj][kk % 4], c);
#pragma unroll
        for (int r = 0; r < 4; ++r) SLm[tile_at(ti, tj, r)] = c[r];
      }
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) nvr[ti][r] = 0.0;
    prod_col(nvr, T, RT);  // -Vr^T = N R^T
    {                      // GN += nvr^T RT
      d4 c = d4{gn[0], gn[1], gn[2], gn[3]};
#pragma unroll
      for (int kk = 0; kk < NS; ++kk) c = mfma(nvr[kk / 4][kk % 4], RT[kk / 4][kk % 4], c);
#pragma unroll
      for (int r = 0; r < 4; ++r) gn[r] = c[r];
    }
    prod_col(RLN, WN, nvr);  // -R_last^T += WN^T nvr
  };

  for (int jp = 0; jp < n - 2; ++jp) {
    double ev[NE], dv[NE];
    Col pnext;
    fetch_row(ev, jp + 1, -1);  // next plane's stencil rows: in flight during the first half of the step
    fetch_row(dv, jp + 1, 0);
    fetch_P(pnext, jp + 1, 1.0);
    Mat nv;
    Col nvr;
    eliminate(nv, nvr, jp + 1);
    Mat ET;
    scatter_load(ET, ev, 1.0, 0.0);
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) RT[ti][r] = pnext[ti][r];
    prod_col(RT, ET, nvr);  // R^T_next = P^T - E Vr^T
    zero_mat(WN);
    prod(WN, ET, nv);       // WN_next = E V^T
    zero_mat(nv);
    prod(nv, T, ET);        // N E^T = -Z^T
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
        for (int r = 0; r < 4; ++r) WN[ti][tj][r] += nv[ti][tj][r];
    Col nvr;
    eliminate(nv, nvr, n - 1);
  }

  // ---- last plane: gauge (drop the bs unknowns of the last node), inverse, loads --------------------------------------------------------------
#pragma unroll
  for (int ti = 0; ti < NT; ++ti)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * ti + 4 * r + lk;
      const bool prow = row >= b - BSV && row < b;
      if (prow) RLN[ti][r] = 0.0;
#pragma unroll
      for (int tj = 0; tj < NT; ++tj) {
        const int col = 16 * tj + lj;
        const bool pcol = col >= b - BSV && col < b;
        T[ti][tj][r] = (prow || pcol) ? (row == col ? -1.0 : 0.0) : -SLm[tile_at(ti, tj, r)];
      }
    }
  {
    int bad = 0;
    invert(T, bad);
    if (bad && !firstbad) firstbad = n;
    Col nvr;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) nvr[ti][r] = 0.0;
    prod_col(nvr, T, RLN);  // Vr_last^T = N (-R_last^T)
    d4 c = d4{gn[0], gn[1], gn[2], gn[3]};
#pragma unroll
    for (int kk = 0; kk < NS; ++kk) c = mfma(nvr[kk / 4][kk % 4], RLN[kk / 4][kk % 4], c);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * r + lk, q = lj;
      if (m < t && q < t) out[cell * t * t + m * t + q] = C0[cell * t * t + m * t + q] + c[r];
    }
    if (lane == 0 && info) info[cell] = firstbad;
  }
}

}  // namespace hommx
