// small_fused.h -- LDS-resident fused elimination for SMALL plane blocks (b = bs * n^(d-1) <= 64), compiled in small.hip.
//
// Default route for 48 < b <= 64 only (a 64 x 64 block does not fit one wave's registers); blocks b <= 48 take the one-wave-per-cell
// register kernel of small_wave.h and come here only with HOMMX_SMALL_WAVES=2|4 (A/B tests).
//
// After K1 (k_assemble_reg, k_c0: the stencil of every node in HBM, read once) ONE workgroup of NW waves per macro cell -- tiles of the
// padded block, BP = 32 / 48 / 64, dealt round-robin to the waves -- runs the whole block-cyclic elimination (same recurrences and signs
// as blocked_solve) with S, W^T and V^T in LDS and S_last in MFMA accumulators:
//
//     Sinv = S^-1          32 x 32 (and 16 x 16) exchange sweeps in the MFMA accumulator layout by ONE wave (sweep_acc.h: DPP
//                          column broadcasts, no barrier per pivot); 48 / 64: 2 x 2 block inverse, Schur products on the matrix cores
//     V^T  = Sinv W^T      v_mfma_f64_16x16x4_f64: the arrow is kept transposed so that every fragment is a ROW read from LDS
//     S_last -= V W^T      (rows k, k+1 of a fragment on disjoint bank halves: XOR swizzle, or pitch 48)
//     load rows            kept transposed as well (BP x 8): Vr^T = Sinv R^T, G += Vr R^T, R_last^T -= W Vr^T on the matrix cores
//     W^T_next = -E V^T ;  S_next = D_{j+1} - (E Sinv) E^T ;  R^T_next = P^T - E Vr^T          E (<= 27 entries per row, from the node
//                          stencil) is scattered DENSE into the buffer the dead arrow leaves free, and the three products run on the
//                          matrix cores as well: gather loops over LDS cost 3x more here than the zeros cost the MFMAs
//
// The stencil rows a step needs are fetched from HBM one whole step ahead (registers).  Correctors are not formed here: hommx_solve_batch_correctors
// stays on the HBM-resident route.
#pragma once

#include "geo.h"
#include "sweep_acc.h"

namespace hommx {

#ifdef HOMMX_SF_PROF
__device__ long long g_sf_prof[16];
#define SF_T(i) do { if (cell == 0 && tid == 0) { const long long now_ = wall_clock64(); g_sf_prof[i] += now_ - t_last_; t_last_ = now_; } } while (0)
#else
#define SF_T(i) do { } while (0)
#endif

// LDS matrix index.  A fragment read takes rows k, k+1 (16 doubles each) in one 32-lane group: they must fall on disjoint bank
// halves.  BP = 48 does that by itself (pitch == 16 mod 32 doubles); BP = 32 / 64 swap the 16-double halves of odd rows.
template <int BP>
__device__ __forceinline__ int swz(int row, int col) {
  if (BP % 32 == 16) return row * BP + col;
  return row * BP + (col ^ ((row & 1) << 4));
}

template <int BP, int BSV, int NIPC, int NW>
__global__ __launch_bounds__(NW * 64) void k_small_fused(Geo G, const double* __restrict__ Kst,
                                                                              const double* __restrict__ Brhs,
                                                                              const double* __restrict__ C0, double* __restrict__ out,
                                                                              int32_t* __restrict__ info, long long ncells) {
  constexpr int NTL = BP / 16;        // 16 x 16 tiles per dimension
  constexpr int NTILE = NTL * NTL;
  constexpr int NTH = NW * 64;        // NW waves per workgroup (= per macro cell); tiles are dealt round-robin to the waves
  constexpr int TPW = (NTILE + NW - 1) / NW;  // tiles per wave
  static_assert(NTH >= BP, "one thread per row of the plane block");
  constexpr int NE = NIPC * BSV;      // entries per row of E
  constexpr int LP = 8;               // pitch of the transposed load-row matrices (t <= 6 rows)
  constexpr int H2 = BP - 32;         // second diagonal block of the 2 x 2 block inverse (0 / 16 / 32)
  __shared__ double Sm[BP * BP];      // S -> Sinv -> S_next; finally S_last
  __shared__ double WT[BP * BP];      // W^T
  __shared__ double VT[BP * BP];      // V^T, Z = E Sinv; scratch of the block inverse
  __shared__ double RT[BP * LP], RlT[BP * LP], VrT[BP * LP];  // R^T, R_last^T, Vr^T
  __shared__ double ubuf[4 * 32];     // pivot-row buffers of the sweeps
  __shared__ double tsc[16 * 17];     // tile transposes of the 32 x 32 block inverse
  __shared__ int badflag;

  const long long cell = blockIdx.x;
  if (cell >= ncells) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lj = lane & 15, lk = lane >> 4;
  const int n = G.n, b = G.b, t = G.t, npl = G.npl, nn = G.nn;
  const double* Kc = Kst + cell * (long long)G.ncode * BSV * BSV * nn;
  const double* Bc = Brhs + cell * (long long)t * BSV * nn;

  // stencil entry: K[(node q of plane pl, comp al), (in-plane neighbour ipc of q in plane pl + o, comp be)]
  auto kst = [&](int pl, int o, int q, int ipc, int al, int be) {
    const int code = ipc + (o + 1) * NIPC;
    return Kc[((long long)(code * BSV + al) * BSV + be) * nn + q + npl * pl];
  };
  // Thread ec = tid < BP owns row ec of the plane block (node ec / BSV, component ec % BSV): it keeps the NE stencil entries of ITS
  // row in registers, fetched from HBM a whole elimination step before they are used.
  const int ec = tid;
  const bool rowthread = tid < BP;
  const bool realrow = rowthread && ec < b;
  int ex[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) ex[e] = 0;
  if (realrow) {
#pragma unroll
    for (int ipc = 0; ipc < NIPC; ++ipc) {
      const int qn = plane_neighbour(G, ec / BSV, ipc);
#pragma unroll
      for (int be = 0; be < BSV; ++be) ex[ipc * BSV + be] = qn * BSV + be;
    }
  }
  auto fetch_row = [&](double (&dst)[NE], int pl, int o) {  // row ec of K[(., pl), (., pl + o)]
#pragma unroll
    for (int e = 0; e < NE; ++e) dst[e] = 0.0;
    if (realrow) {
#pragma unroll
      for (int ipc = 0; ipc < NIPC; ++ipc)
#pragma unroll
        for (int be = 0; be < BSV; ++be) dst[ipc * BSV + be] = kst(pl, o, ec / BSV, ipc, ec % BSV, be);
    }
  };
  // load rows: entry idx = c * LP + m of the transposed (BP x LP) matrices; thread tid holds entries tid, tid + NTH, ...
  constexpr int NPL = (BP * LP + NTH - 1) / NTH;
  auto fetch_P = [&](double (&dst)[NPL], int pl) {
#pragma unroll
    for (int q = 0; q < NPL; ++q) {
      const int idx = tid + q * NTH, c = idx / LP, m = idx % LP;
      dst[q] = (idx < BP * LP && c < b && m < t) ? Bc[((long long)m * BSV + c % BSV) * nn + c / BSV + npl * pl] : 0.0;
    }
  };
  // dst (+)= rows held in registers (thread (ec, 0) owns row ec; several codes can hit one neighbour on tiny meshes: accumulate)
  auto add_rows = [&](double* dst, const double (&v)[NE], bool transposed, bool padIdentity) {
    if (rowthread) {
      if (realrow) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          if (transposed) dst[swz<BP>(ex[e], ec)] += v[e];
          else dst[swz<BP>(ec, ex[e])] += v[e];
        }
      } else if (padIdentity) {
        dst[swz<BP>(ec, ec)] = 1.0;
      }
    }
    __syncthreads();
  };
  auto zero = [&](double* dst) {
    for (int i = tid; i < BP * BP; i += NTH) dst[i] = 0.0;
  };

  // ---- dense products on the matrix cores ---------------------------------------------------------------------------------------------
  // One 16 x 16 output tile:  acc += sum_{k < K} A(i, k) B(k, j),  A(i, k) = AT[ak0 + k][ai0 + i],  B(k, j) = Bm[bk0 + k][bj0 + j];
  // AT / Bm are BP-pitch swizzled matrices, or (narrow*) the pitch-LP load-row matrices, whose columns >= LP read as zero.
  auto frag = [&](const double* Mx, bool narrow, int row, int col) {
    if (narrow) return col < LP ? Mx[row * LP + col] : 0.0;
    return Mx[swz<BP>(row, col)];
  };
  auto tile_mm = [&](d4 acc, const double* AT, bool narrowA, int ak0, int ai0, const double* Bm, bool narrowB, int bk0, int bj0, int K,
                     bool negate) {
    for (int kk = 0; kk < K / 4; ++kk) {
      double af = frag(AT, narrowA, ak0 + 4 * kk + lk, ai0 + lj);
      const double bf = frag(Bm, narrowB, bk0 + 4 * kk + lk, bj0 + lj);
      if (negate) af = -af;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
    }
    return acc;
  };
  // full BP x BP product, fragments of a tile fetched up front:  OUT = A B  with A(i, k) = AT[k][i]
  auto mfma_full = [&](const double* AT, const double* Bm, double* OUT) {
    for (int tile = wave; tile < NTILE; tile += NW) {
      const int ti = tile / NTL, tj = tile % NTL;
      double af[BP / 4], bf[BP / 4];
#pragma unroll
      for (int kk = 0; kk < BP / 4; ++kk) {
        af[kk] = AT[swz<BP>(4 * kk + lk, 16 * ti + lj)];
        bf[kk] = Bm[swz<BP>(4 * kk + lk, 16 * tj + lj)];
      }
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < BP / 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) OUT[swz<BP>(16 * ti + 4 * r + lk, 16 * tj + lj)] = acc[r];
    }
    __syncthreads();
  };
  auto store_tile = [&](double* OUT, int r0, int c0, d4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) OUT[swz<BP>(r0 + 4 * r + lk, c0 + lj)] = acc[r];
  };
  auto load_tile = [&](const double* IN, int r0, int c0) {
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = IN[swz<BP>(r0 + 4 * r + lk, c0 + lj)];
    return acc;
  };

  // ---- inverse of the SPD matrix M (LDS) in place ---------------------------------------------------------------------------------------
  // diagonal block [o, o + 32) or [o, o + 16): ONE wave sweeps T = -block in the accumulator layout (T^-1 = -block^-1)
  auto sweep32 = [&](double* M, int o, int& bad) {
    if (wave == 0) {
      double a[2][2][4];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) a[ti][tj][r] = -M[swz<BP>(o + 16 * ti + 4 * r + lk, o + 16 * tj + lj)];
      __builtin_amdgcn_s_setprio(3);  // the rest of the workgroup waits for this wave
      accl::block_inverse32(a, ubuf, tsc, lj, lk, bad);  // 2 x 2 blocks around two 16-sweeps (sweep_acc.h)
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) M[swz<BP>(o + 16 * ti + 4 * r + lk, o + 16 * tj + lj)] = -a[ti][tj][r];
    }
    __syncthreads();
  };
  auto sweep16 = [&](double* M, int o, int& bad) {
    if (wave == 0) {
      double a[1][1][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) a[0][0][r] = -M[swz<BP>(o + 4 * r + lk, o + lj)];
      __builtin_amdgcn_s_setprio(3);
      accl::Sweep<16>::run(a, ubuf, lj, lk, bad);
      __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) M[swz<BP>(o + 4 * r + lk, o + lj)] = -a[0][0][r];
    }
    __syncthreads();
  };
  // M = [[A, B^T], [B, C]], A 32 x 32, C H2 x H2:  Ai = A^-1,  X = B Ai,  Sc = C - X B^T,  Sci = Sc^-1,
  //   M21 = -Sci X,  M12 = M21^T,  M11 = Ai - X^T M21,  M22 = Sci.   Scratch: X^T (32 x H2) and X (H2 x 32) in T1 / T2.
  auto invert = [&](double* M, double* scratch, int stepcode) {
    int bad = 0;
    sweep32(M, 0, bad);
    if constexpr (H2 > 0) {
      constexpr int HT = H2 / 16;
      double* T1 = scratch;            // X^T: rows 0..31 (k of A), cols 0..H2-1   (BP-pitch view, rows 0..31)
      double* T2 = scratch + 32 * BP;  // X:   rows 0..H2-1, cols 0..31            (BP-pitch view, rows 32..)
      // X^T = Ai B^T (2 x HT tiles) and X = B Ai (HT x 2 tiles)
      for (int tl = wave; tl < 4 * HT; tl += NW) {
        const bool second = tl >= 2 * HT;
        const int q = second ? tl - 2 * HT : tl;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        if (!second) {  // X^T[i][j] = sum_k Ai[k][i] M[k][32 + j]
          const int ti = q / HT, tj = q % HT;
          acc = tile_mm(acc, M, false, 0, 16 * ti, M, false, 0, 32 + 16 * tj, 32, false);
          store_tile(T1, 16 * ti, 16 * tj, acc);
        } else {        // X[i][j] = sum_k M[k][32 + i] Ai[k][j]
          const int ti = q / 2, tj = q % 2;
          acc = tile_mm(acc, M, false, 0, 32 + 16 * ti, M, false, 0, 16 * tj, 32, false);
          store_tile(T2, 16 * ti, 16 * tj, acc);
        }
      }
      __syncthreads();
      // Sc = C - X B^T:  Sc[i][j] -= sum_k X^T[k][i] M[k][32 + j]
      for (int tl = wave; tl < HT * HT; tl += NW) {
        const int ti = tl / HT, tj = tl % HT;
        d4 acc = load_tile(M, 32 + 16 * ti, 32 + 16 * tj);
        acc = tile_mm(acc, T1, false, 0, 16 * ti, M, false, 0, 32 + 16 * tj, 32, true);
        store_tile(M, 32 + 16 * ti, 32 + 16 * tj, acc);
      }
      __syncthreads();
      if constexpr (H2 == 32) sweep32(M, 32, bad);
      else sweep16(M, 32, bad);
      // M21 = -Sci X (HT x 2 tiles), M12 = -X^T Sci (2 x HT tiles)
      for (int tl = wave; tl < 4 * HT; tl += NW) {
        const bool second = tl >= 2 * HT;
        const int q = second ? tl - 2 * HT : tl;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        if (!second) {  // M21[i][j] = -sum_k Sci[k][i] X[k][j]
          const int ti = q / 2, tj = q % 2;
          acc = tile_mm(acc, M, false, 32, 32 + 16 * ti, T2, false, 0, 16 * tj, H2, true);
          store_tile(M, 32 + 16 * ti, 16 * tj, acc);
        } else {        // M12[i][j] = -sum_k X[k][i] Sci[k][j]
          const int ti = q / HT, tj = q % HT;
          acc = tile_mm(acc, T2, false, 0, 16 * ti, M, false, 32, 32 + 16 * tj, H2, true);
          store_tile(M, 16 * ti, 32 + 16 * tj, acc);
        }
      }
      __syncthreads();
      // M11 = Ai - X^T M21:  M11[i][j] -= sum_k X[k][i] M21[k][j]
      for (int tl = wave; tl < 4; tl += NW) {
        const int ti = tl / 2, tj = tl % 2;
        d4 acc = load_tile(M, 16 * ti, 16 * tj);
        acc = tile_mm(acc, T2, false, 0, 16 * ti, M, false, 32, 16 * tj, H2, true);
        store_tile(M, 16 * ti, 16 * tj, acc);
      }
      __syncthreads();
    }
    if (bad) badflag = stepcode;  // wave 0 only
    __syncthreads();
  };

  // ---- coupling E = K[(., plane pl), (., plane pl - 1)]: row ec in ev[] (registers), scattered dense (transposed) for the products ----
  double ev[NE], dv[NE], el[NE];
  // full product with the result left in this wave's accumulator tiles:  acc[q] = -(A B)(tile wave + NW q),  A(i, k) = AT[k][i]
  auto mfma_full_regs = [&](const double* AT, const double* Bm, d4 (&acc)[TPW]) {
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int tile = wave + NW * q;
      acc[q] = d4{0.0, 0.0, 0.0, 0.0};
      if (tile < NTILE) {
        const int ti = tile / NTL, tj = tile % NTL;
        double af[BP / 4], bf[BP / 4];
#pragma unroll
        for (int kk = 0; kk < BP / 4; ++kk) {
          af[kk] = -AT[swz<BP>(4 * kk + lk, 16 * ti + lj)];
          bf[kk] = Bm[swz<BP>(4 * kk + lk, 16 * tj + lj)];
        }
#pragma unroll
        for (int kk = 0; kk < BP / 4; ++kk) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk], acc[q], 0, 0, 0);
      }
    }
  };

  // ---- init: every global load of the prologue is issued before the first use ------------------------------------------------------------
  if (tid == 0) badflag = 0;
  d4 slacc[TPW];                     // S_last tiles of this wave (tile = wave + NW q), accumulator layout
  d4 gacc = d4{0.0, 0.0, 0.0, 0.0};  // G (wave 0)
  {
    double d0[NE], w0[NE], dl[NE], p0[NPL], pl[NPL];
    fetch_row(d0, 0, 0);        // D_0
    fetch_row(w0, n - 1, +1);   // K[(., n-1), (., 0)]
    fetch_row(dl, n - 1, 0);    // D_{n-1}
    fetch_row(el, n - 1, -1);   // K[(., n-1), (., n-2)]: joins the arrow on the last step
    fetch_P(p0, 0);
    fetch_P(pl, n - 1);
    zero(Sm); zero(WT); zero(VT);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NPL; ++q)
      if (tid + q * NTH < BP * LP) { RT[tid + q * NTH] = p0[q]; RlT[tid + q * NTH] = pl[q]; }
    add_rows(VT, dl, false, true);   // S_last = D_{n-1} -> accumulators
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int tile = wave + NW * q;
      if (tile < NTILE) slacc[q] = load_tile(VT, 16 * (tile / NTL), 16 * (tile % NTL));
    }
    add_rows(Sm, d0, false, true);   // S = D_0
    add_rows(WT, w0, true, false);   // W, stored transposed
  }

  int firstbad = 0;
#ifdef HOMMX_SF_PROF
  long long t_last_ = wall_clock64();
#endif
  for (int jp = 0; jp <= n - 2; ++jp) {
    const bool last = (jp == n - 2);
    double pnext[NPL];
    if (!last) {  // next plane's stencil rows: in flight during the whole step
      fetch_row(ev, jp + 1, -1);
      fetch_row(dv, jp + 1, 0);
      fetch_P(pnext, jp + 1);
    } else {
      add_rows(WT, el, true, false);  // the last plane couples to plane n-2 through E as well
    }
    SF_T(0);
    invert(Sm, VT, jp + 1);
    if (badflag && !firstbad) firstbad = badflag;
    SF_T(1);
    mfma_full(Sm, WT, VT);    // V^T = Sinv W^T   (Sinv symmetric: A(i, k) = Sinv[k][i])
    // S_last -= V W^T  (A(i, k) = V[i][k] = VT[k][i], B(k, j) = W^T[k][j]); accumulators stay in registers
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      const int tile = wave + NW * q;
      if (tile < NTILE) {
        const int sti = tile / NTL, stj = tile % NTL;
        double af[BP / 4], bf[BP / 4];
#pragma unroll
        for (int kk = 0; kk < BP / 4; ++kk) {
          af[kk] = -VT[swz<BP>(4 * kk + lk, 16 * sti + lj)];
          bf[kk] = WT[swz<BP>(4 * kk + lk, 16 * stj + lj)];
        }
#pragma unroll
        for (int kk = 0; kk < BP / 4; ++kk) slacc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk], slacc[q], 0, 0, 0);
      }
    }
    SF_T(2);
    // load rows (transposed, pitch LP): Vr^T = Sinv R^T ; G += Vr R^T ; R_last^T -= W Vr^T
    for (int tr = wave; tr < NTL; tr += NW) {  // Vr^T[i][m] = sum_k Sinv[k][i] R^T[k][m]
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
      acc = tile_mm(acc, Sm, false, 0, 16 * tr, RT, true, 0, 0, BP, false);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lj < LP) VrT[(16 * tr + 4 * r + lk) * LP + lj] = acc[r];
    }
    __syncthreads();
    if (wave == 0) gacc = tile_mm(gacc, VrT, true, 0, 0, RT, true, 0, 0, BP, false);  // G[m][q] += sum_c Vr^T[c][m] R^T[c][q]
    for (int tr = (wave + NW - 1) % NW; tr < NTL; tr += NW) {  // R_last^T[r][m] -= sum_c W^T[c][r] Vr^T[c][m]   (wave 0 last: it has G)
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
      acc = tile_mm(acc, WT, false, 0, 16 * tr, VrT, true, 0, 0, BP, false);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lj < LP) RlT[(16 * tr + 4 * r + lk) * LP + lj] -= acc[r];
    }
    __syncthreads();
    SF_T(3);
    if (!last) {
      // E^T dense into the arrow's buffer (W is dead: R_last has used it)
      zero(WT);
      __syncthreads();
      add_rows(WT, ev, true, false);
      // R^T_next = P^T_{j+1} - E Vr^T:  (E Vr^T)[c][m] = sum_k E^T[k][c] Vr^T[k][m]
      for (int tr = wave; tr < NTL; tr += NW) {
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        acc = tile_mm(acc, WT, false, 0, 16 * tr, VrT, true, 0, 0, BP, true);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (lj < LP) RT[(16 * tr + 4 * r + lk) * LP + lj] = acc[r];  // -E Vr^T; P^T is added below by the threads that fetched it
      }
      d4 wnext[TPW];
      mfma_full_regs(WT, VT, wnext);   // W^T_next = -E V^T  (A(i, k) = E[i][k] = E^T[k][i]) -> registers: its buffer still holds E^T
      __syncthreads();                 // V^T and R^T tiles consumed / written
#pragma unroll
      for (int q = 0; q < NPL; ++q)
        if (tid + q * NTH < BP * LP) RT[tid + q * NTH] += pnext[q];
      mfma_full(Sm, WT, VT);           // Z^T = Sinv E^T   (Sinv symmetric)
      {                                // S_next = -Z E^T ...:  A(i, k) = Z[i][k] = Z^T[k][i], B(k, j) = E^T[k][j]
        d4 sn[TPW];
        mfma_full_regs(VT, WT, sn);
        __syncthreads();               // E^T consumed: the arrow's buffer is free again
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
          const int tile = wave + NW * q;
          if (tile < NTILE) {
            store_tile(Sm, 16 * (tile / NTL), 16 * (tile % NTL), sn[q]);
            store_tile(WT, 16 * (tile / NTL), 16 * (tile % NTL), wnext[q]);
          }
        }
      }
      __syncthreads();
      add_rows(Sm, dv, false, true);   //     ... + D_{j+1}
      SF_T(4);
    }
  }

  // ---- last plane: S_last out of the accumulators, gauge (drop the bs unknowns of the last node), inverse, loads ------------------------
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int tile = wave + NW * q;
    if (tile < NTILE) store_tile(Sm, 16 * (tile / NTL), 16 * (tile % NTL), slacc[q]);
  }
  __syncthreads();
  for (int i = tid; i < BP * BSV; i += NTH) {
    const int x = i % BP, p = b - BSV + i / BP;
    Sm[swz<BP>(p, x)] = (x == p) ? 1.0 : 0.0;
    Sm[swz<BP>(x, p)] = (x == p) ? 1.0 : 0.0;
    if (x < LP) RlT[p * LP + x] = 0.0;
  }
  __syncthreads();
  invert(Sm, VT, n);
  if (badflag && !firstbad) firstbad = badflag;
  for (int tr = wave; tr < NTL; tr += NW) {
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    acc = tile_mm(acc, Sm, false, 0, 16 * tr, RlT, true, 0, 0, BP, false);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (lj < LP) VrT[(16 * tr + 4 * r + lk) * LP + lj] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    gacc = tile_mm(gacc, VrT, true, 0, 0, RlT, true, 0, 0, BP, false);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * r + lk, q = lj;
      if (m < t && q < t) out[cell * t * t + m * t + q] = C0[cell * t * t + m * t + q] - gacc[r];
    }
    if (lane == 0 && info) info[cell] = firstbad;
  }
}

}  // namespace hommx

#ifdef HOMMX_SF_PROF
extern "C" void hommx_sf_prof_read(long long* out16, int reset) {  // dev builds only (tools/sf_prof.py)
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(hommx::g_sf_prof), sizeof(long long) * 16);
  if (reset) {
    long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(hommx::g_sf_prof), z, sizeof(z));
  }
}
#endif
