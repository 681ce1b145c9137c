// blocked.hip -- generic block-cyclic micro-cell path: any dimension (2, 3), any problem kind (scalar / matrix-valued
// Poisson, isotropic / general elasticity), optional stratification matrix M, any n_micro >= 3.
//
// Pipeline per chunk of macro cells (host-orchestrated batched kernels, grid = cells x tiles):
//   K1  k_assemble / k_c0 : periodic P1 stencil (3^d slots x bs x bs per node), canonical loads, C0
//                           (hmm.py:644-650 / 759-772 / 891-903 / 1032-1048; periodic map cell_problem.py:38-300)
//   K2  block-cyclic elimination over node planes, block b = bs * n^(d-1) (padded to Bp = 32 k):
//           Sinv = S^-1 (recursive Schur-complement inversion: 32x32 in-register sweeps + fp64-MFMA GEMMs)
//           V = W Sinv ; S_last -= V W^T ; S_next = D_{j+1} - E Sinv E^T ; W_next = -V E^T      (E sparse, from the stencil)
//           Vr = R Sinv ; G += Vr R^T ; R_last -= Vr W^T ; R_next = P_{j+1} - Vr E^T           (t <= 6 load rows, padded to 16)
//   K3  k_finalize : A_H = C0 - G   (== the energy functional hmm.py:652-667 / 774-789 / 905-922 / 1050-1067)
//
// Unified element kernel: with w_{a,alpha} in R^t the (Voigt-weighted) "strain" of basis function (a, alpha) and Cv the
// t x t element matrix  E^m : A : E^n :   K = vol w^T Cv w',  B_m = -vol (Cv w)_m,  C0 = sum vol Cv.
// Poisson is the case bs = 1, w_a = M grad(lambda_a), Cv = A (d x d).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/hommx_hip.h"
#include "blocked_internal.h"
#include "geo.h"
#include "kernels.h"
#include "sweep.h"

namespace hommx {


thread_local std::string g_berr;
const char* blocked_last_error() { return g_berr.c_str(); }

#define BTRY(expr)                                                                       \
  do {                                                                                   \
    hipError_t e__ = (expr);                                                             \
    if (e__ != hipSuccess) {                                                             \
      g_berr = std::string(#expr) + ": " + hipGetErrorString(e__);                       \
      return e__ == hipErrorOutOfMemory ? HOMMX_ENOMEM : HOMMX_EHIP;                     \
    }                                                                                    \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------
// K1: assembly
// ---------------------------------------------------------------------------------------------------------------

// Voigt weights of sym(e_alpha (x) g): diagonal pairs first, then (01)[,(02),(12)] with factor 2 folded in.
__device__ __forceinline__ void strain_weights(const Geo& G, const double* g, int alpha, double* w) {
  const int d = G.dim;
  if (G.bs == 1) {
    for (int k = 0; k < d; ++k) w[k] = g[k];
    return;
  }
  for (int k = 0; k < d; ++k) w[k] = (k == alpha) ? g[k] : 0.0;
  int m = d;
  for (int k = 0; k < d; ++k)
    for (int l = k + 1; l < d; ++l, ++m) w[m] = (k == alpha ? g[l] : 0.0) + (l == alpha ? g[k] : 0.0);
}

// t x t element matrix Cv from the coefficient stream
__device__ __forceinline__ void element_matrix(const Geo& G, const double* c, double* Cv) {
  const int d = G.dim, t = G.t;
  for (int i = 0; i < t * t; ++i) Cv[i] = 0.0;
  if (G.kind == HOMMX_KIND_POISSON_SCALAR) {
    for (int k = 0; k < d; ++k) Cv[k * t + k] = c[0];
  } else if (G.kind == HOMMX_KIND_POISSON_MATRIX) {
    for (int k = 0; k < d; ++k) Cv[k * t + k] = c[k];
    int m = d;
    for (int k = 0; k < d; ++k)
      for (int l = k + 1; l < d; ++l, ++m) Cv[k * t + l] = Cv[l * t + k] = c[m];
  } else if (G.kind == HOMMX_KIND_ELASTICITY_ISO) {
    const double lam = c[0], mu = c[1];
    for (int k = 0; k < d; ++k)
      for (int l = 0; l < d; ++l) Cv[k * t + l] = lam + (k == l ? 2.0 * mu : 0.0);
    for (int m = d; m < t; ++m) Cv[m * t + m] = mu;
  } else {
    int q = 0;
    for (int k = 0; k < t; ++k)
      for (int l = k; l < t; ++l, ++q) Cv[k * t + l] = Cv[l * t + k] = c[q];
  }
}

// compile-time variants of the two helpers above (everything stays in registers, loops unroll)
template <int D, int BSV, int T>
__device__ __forceinline__ void strain_weights_ct(const double* g, int alpha, double* w) {
  if (BSV == 1) {
#pragma unroll
    for (int k = 0; k < D; ++k) w[k] = g[k];
    return;
  }
#pragma unroll
  for (int k = 0; k < D; ++k) w[k] = (k == alpha) ? g[k] : 0.0;
  int m = D;
#pragma unroll
  for (int k = 0; k < D; ++k)
#pragma unroll
    for (int l = k + 1; l < D; ++l, ++m) w[m] = (k == alpha ? g[l] : 0.0) + (l == alpha ? g[k] : 0.0);
}

template <int D, int KIND, int T>
__device__ __forceinline__ void element_matrix_ct(const double* c, double* Cv) {
#pragma unroll
  for (int i = 0; i < T * T; ++i) Cv[i] = 0.0;
  if (KIND == HOMMX_KIND_POISSON_SCALAR) {
#pragma unroll
    for (int k = 0; k < D; ++k) Cv[k * T + k] = c[0];
  } else if (KIND == HOMMX_KIND_POISSON_MATRIX) {
#pragma unroll
    for (int k = 0; k < D; ++k) Cv[k * T + k] = c[k];
    int m = D;
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
      for (int l = k + 1; l < D; ++l, ++m) Cv[k * T + l] = Cv[l * T + k] = c[m];
  } else if (KIND == HOMMX_KIND_ELASTICITY_ISO) {
    const double lam = c[0], mu = c[1];
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
      for (int l = 0; l < D; ++l) Cv[k * T + l] = lam + (k == l ? 2.0 * mu : 0.0);
#pragma unroll
    for (int m = D; m < T; ++m) Cv[m * T + m] = mu;
  } else {
    int q = 0;
#pragma unroll
    for (int k = 0; k < T; ++k)
#pragma unroll
      for (int l = k; l < T; ++l, ++q) Cv[k * T + l] = Cv[l * T + k] = c[q];
  }
}

// Corner offsets of the sub-elements as compile-time constants (the same tables fill_tables() puts into Geo::voff): with them the
// stencil slot `code` of every (sub-element, vertex, vertex) triple is a constant and the node's stencil row can stay in registers.
template <int D>
__host__ __device__ constexpr int voff_ct(int s, int a, int k) {
  constexpr int tri[2][3][2] = {{{0, 0}, {1, 0}, {1, 1}}, {{0, 0}, {0, 1}, {1, 1}}};
  constexpr int vb[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
  constexpr int tet[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
  return D == 2 ? tri[s][a][k] : vb[tet[s][a]][k];
}
template <int D>
__host__ __device__ constexpr int code_ct(int s, int a, int b) {
  int cd = 0, p3 = 1;
  for (int k = 0; k < D; ++k, p3 *= 3) cd += (voff_ct<D>(s, b, k) - voff_ct<D>(s, a, k) + 1) * p3;
  return cd;
}

// K1 with the node's whole stencil row (NCODE x bs x bs) and load entries accumulated in REGISTERS and written once -- no
// read-modify-write chains through L2, no memset of the stencil array.  Same arithmetic, same order of the 24 / 6 incident
// (sub-element, vertex) pairs as k_assemble: bitwise the same numbers.  ALSPLIT == 0: one thread per node (bs^2 * 3^d <= 36:
// scalar kinds, 2D elasticity); ALSPLIT == 1: one thread per (node, row component) -- 3D elasticity, 81 + 6 accumulators per thread.
template <int D, int KIND, int ALSPLIT>
__global__ __launch_bounds__(128) void k_assemble_reg(Geo G, const double* __restrict__ coef, const double* __restrict__ Mmat,
                                                      double* __restrict__ Kst, double* __restrict__ Brhs, long long ncells) {
  constexpr bool EL = KIND >= HOMMX_KIND_ELASTICITY_ISO;
  constexpr int BSV = EL ? D : 1, T = EL ? D * (D + 1) / 2 : D, NV = D + 1, NSUB = (D == 2) ? 2 : 6, NCODE = (D == 2) ? 9 : 27;
  constexpr int NCOMP = KIND == HOMMX_KIND_POISSON_SCALAR ? 1
                        : KIND == HOMMX_KIND_POISSON_MATRIX ? D * (D + 1) / 2
                        : KIND == HOMMX_KIND_ELASTICITY_ISO ? 2
                                                            : T * (T + 1) / 2;
  constexpr int NAL = ALSPLIT ? 1 : BSV;  // row components per thread
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long nodes = ALSPLIT ? idx / BSV : idx;
  const int al0 = ALSPLIT ? (int)(idx % BSV) : 0;
  if (nodes >= ncells * G.nn) return;
  const long long cell = nodes / G.nn;
  const int node = (int)(nodes % G.nn);
  const int n = G.n;
  int pc[3] = {node % n, (node / n) % n, D == 3 ? node / (n * n) : 0};
  double M[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) M[i][j] = Mmat ? Mmat[cell * D * D + i * D + j] : (i == j ? 1.0 : 0.0);
  double vol = 1.0;
#pragma unroll
  for (int k = 0; k < D; ++k) vol /= n;
  vol /= (D == 2 ? 2.0 : 6.0);
  const double* ccell = coef + cell * (long long)G.n_el * NCOMP;
  double Kacc[NCODE * NAL * BSV], Bacc[T * NAL];
#pragma unroll
  for (int i = 0; i < NCODE * NAL * BSV; ++i) Kacc[i] = 0.0;
#pragma unroll
  for (int i = 0; i < T * NAL; ++i) Bacc[i] = 0.0;
#pragma unroll
  for (int s = 0; s < NSUB; ++s) {
#pragma unroll
    for (int a = 0; a < NV; ++a) {
      int cc[3] = {0, 0, 0};
#pragma unroll
      for (int k = 0; k < D; ++k) {
        int v = pc[k] - voff_ct<D>(s, a, k);
        cc[k] = v < 0 ? v + n : v;
      }
      const long long e = (long long)NSUB * (cc[0] + n * (cc[1] + (long long)n * cc[2])) + s;
      double cval[NCOMP];
#pragma unroll
      for (int q = 0; q < NCOMP; ++q) cval[q] = ccell[e * NCOMP + q];
      double Cv[T * T];
      element_matrix_ct<D, KIND, T>(cval, Cv);
      double gt[NV][D];  // g~_b = M (n grad_b)
#pragma unroll
      for (int b = 0; b < NV; ++b)
#pragma unroll
        for (int i = 0; i < D; ++i) {
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) acc += M[i][k] * G.grad[s][b][k];
          gt[b][i] = acc * n;
        }
#pragma unroll
      for (int ai = 0; ai < NAL; ++ai) {
        const int al = ALSPLIT ? al0 : ai;
        double w[T], y[T];
        strain_weights_ct<D, BSV, T>(gt[a], al, w);
#pragma unroll
        for (int m = 0; m < T; ++m) {
          double acc = 0.0;
#pragma unroll
          for (int q = 0; q < T; ++q) acc += Cv[m * T + q] * w[q];
          y[m] = vol * acc;
        }
#pragma unroll
        for (int m = 0; m < T; ++m) Bacc[m * NAL + ai] -= y[m];
#pragma unroll
        for (int b = 0; b < NV; ++b) {
#pragma unroll
          for (int be = 0; be < BSV; ++be) {
            double wb[T];
            strain_weights_ct<D, BSV, T>(gt[b], be, wb);
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < T; ++m) acc += y[m] * wb[m];
            Kacc[(code_ct<D>(s, a, b) * NAL + ai) * BSV + be] += acc;
          }
        }
      }
    }
  }
  double* Kc = Kst + cell * (long long)NCODE * BSV * BSV * G.nn;
  double* Bc = Brhs + cell * (long long)T * BSV * G.nn;
#pragma unroll
  for (int code = 0; code < NCODE; ++code)
#pragma unroll
    for (int ai = 0; ai < NAL; ++ai)
#pragma unroll
      for (int be = 0; be < BSV; ++be)
        Kc[(((long long)code * BSV + (ALSPLIT ? al0 : ai)) * BSV + be) * G.nn + node] = Kacc[(code * NAL + ai) * BSV + be];
#pragma unroll
  for (int m = 0; m < T; ++m)
#pragma unroll
    for (int ai = 0; ai < NAL; ++ai) Bc[((long long)m * BSV + (ALSPLIT ? al0 : ai)) * G.nn + node] = Bacc[m * NAL + ai];
}

// C0[cell][t][t] = sum_e vol Cv_e.  Compile-time (dim, kind): the t x t partial sums stay in registers.  WPC waves per macro cell
// (4: one 256-thread block per cell; 1: small meshes, four cells per block, no LDS, no barrier).  Fixed summation order (lane-strided
// partial sums, wave butterfly, wave totals added in order): bitwise reproducible.
template <int D, int KIND, int WPC>
__global__ __launch_bounds__(256) void k_c0(Geo G, const double* __restrict__ coef, double* __restrict__ C0, long long ncells) {
  constexpr bool EL = KIND >= HOMMX_KIND_ELASTICITY_ISO;
  constexpr int T = EL ? D * (D + 1) / 2 : D, TT = T * T;
  constexpr int NCOMP = KIND == HOMMX_KIND_POISSON_SCALAR ? 1
                        : KIND == HOMMX_KIND_POISSON_MATRIX ? D * (D + 1) / 2
                        : KIND == HOMMX_KIND_ELASTICITY_ISO ? 2
                                                            : T * (T + 1) / 2;
  constexpr int NTH = 64 * WPC;  // threads per cell
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long cell = WPC == 4 ? (long long)blockIdx.x : (long long)blockIdx.x * 4 + wave;
  if (cell >= ncells) return;  // WPC == 1: whole waves leave, nothing below synchronises across waves
  double acc[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) acc[i] = 0.0;
  const double* ccell = coef + cell * (long long)G.n_el * NCOMP;
  for (int e = WPC == 4 ? threadIdx.x : lane; e < G.n_el; e += NTH) {
    double cval[NCOMP], Cv[TT];
#pragma unroll
    for (int q = 0; q < NCOMP; ++q) cval[q] = ccell[(long long)e * NCOMP + q];
    element_matrix_ct<D, KIND, T>(cval, Cv);
#pragma unroll
    for (int i = 0; i < TT; ++i) acc[i] += Cv[i];
  }
  double vol = 1.0;
  for (int k = 0; k < D; ++k) vol /= G.n;
  vol /= (D == 2 ? 2.0 : 6.0);
#pragma unroll
  for (int i = 0; i < TT; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
  }
  if constexpr (WPC == 1) {
#pragma unroll
    for (int i = 0; i < TT; ++i)
      if (lane == i) C0[cell * TT + i] = acc[i] * vol;
  } else {
    __shared__ double red[4][TT];
#pragma unroll
    for (int i = 0; i < TT; ++i)
      if (lane == 0) red[wave][i] = acc[i];
    __syncthreads();
    if (threadIdx.x < TT) C0[cell * TT + threadIdx.x] = (((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]) * vol;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// stencil <-> dense plane blocks
// ---------------------------------------------------------------------------------------------------------------

// dst[r][c] += K[(r in plane rowPlane), (c in plane rowPlane + olast)]; optional identity on the padding diagonal
__global__ void k_scatter_plane(Geo G, const double* __restrict__ Kst, double* __restrict__ dst, long long ncells,
                                int rowPlane, int olast, int padIdentity) {
  const int nipc = G.ncode / 3;
  const long long per = (long long)G.Bp * nipc * G.bs;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * per) return;
  const long long cell = idx / per;
  int rem = (int)(idx % per);
  const int r = rem % G.Bp;
  rem /= G.Bp;
  const int ipc = rem % nipc, be = rem / nipc;
  double* D = dst + cell * (long long)G.Bp * G.Bp;
  if (r >= G.b) {
    if (padIdentity && ipc == 0 && be == 0) D[(long long)r * G.Bp + r] = 1.0;
    return;
  }
  const int q = r / G.bs, al = r % G.bs;
  const int node = q + G.npl * rowPlane;
  const int code = ipc + (olast + 1) * nipc;
  const double v = Kst[((cell * G.ncode + code) * G.bs + al) * G.bs * (long long)G.nn + (long long)be * G.nn + node];
  if (v != 0.0) {
    const int c = plane_neighbour(G, q, ipc) * G.bs + be;
    D[(long long)r * G.Bp + c] += v;
  }
}

// OUT[k][c] (+)= alpha * sum_{k'} IN[k][k'] E[c][k'],  E = K[(., plane rowPlane), (., plane rowPlane + o)]  (OUT = alpha IN E^T);
// o = -1 (codeOff = 0, the elimination) or +1 (codeOff = 2 * 3^(d-1), the back substitution).
// One thread per output column c and tile of RT rows k: the NE = bs * 3^(d-1) entries of E row c and their
// column indices are gathered once into registers and reused for every row of the tile.
// One thread per NODE q (its BSV output columns c = q BSV + al) and tile of RT rows k: the BSV x NE entries of E and
// the NE column indices are gathered once into registers; per row every input IN[k][k'] is loaded once and feeds
// the BSV outputs of the node.
template <int BSV, int NE, int RT>
__global__ __launch_bounds__(256) void k_right_mult_Et(Geo G, const double* __restrict__ Kst,
                                                       const double* __restrict__ IN, double* __restrict__ OUT,
                                                       int nrows, int rowPlane, double alpha, int codeOff,
                                                       int accumulate) {
  const int q = blockIdx.x * 256 + threadIdx.x;  // node in plane (or padding)
  if (q * BSV >= G.Bp) return;
  const long long cell = blockIdx.z;
  const int k0 = blockIdx.y * RT;
  const long long per = (long long)nrows * G.Bp;
  double e[BSV][NE];
  int kx[NE];
  const bool real = q < G.npl;
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    kx[j] = 0;
#pragma unroll
    for (int al = 0; al < BSV; ++al) e[al][j] = 0.0;
  }
  if (real) {
    const int nipc = G.ncode / 3;
    const int node = q + G.npl * rowPlane;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int ipc = j / BSV, be = j % BSV;
      if (ipc < nipc) {
        kx[j] = plane_neighbour(G, q, ipc) * BSV + be;
#pragma unroll
        for (int al = 0; al < BSV; ++al)
          e[al][j] = Kst[((cell * G.ncode + ipc + codeOff) * BSV + al) * BSV * (long long)G.nn + (long long)be * G.nn + node];
      }
    }
  }
  const double* in = IN + cell * per;
  double* out = OUT + cell * per;
  const int k1 = min(nrows, k0 + RT);
  for (int k = k0; k < k1; ++k) {
    const double* row = in + (long long)k * G.Bp;
    double acc[BSV];
#pragma unroll
    for (int al = 0; al < BSV; ++al) acc[al] = 0.0;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const double v = row[kx[j]];
#pragma unroll
      for (int al = 0; al < BSV; ++al) acc[al] = fma(v, e[al][j], acc[al]);
    }
#pragma unroll
    for (int al = 0; al < BSV; ++al) {
      const int c = q * BSV + al;
      if (c < G.Bp) {
        double* o = out + (long long)k * G.Bp + c;
        *o = accumulate ? *o + alpha * acc[al] : alpha * acc[al];
      }
    }
  }
}

// OUT[r][c] = alpha * sum_k E[r][k] X[k][c]   (Bp x Bp).  One workgroup per node q (its bs rows r = q bs + al):
// the bs x NE entries of E and the NE row indices are staged in LDS once; every thread then walks its columns c,
// loading each X[k][c] once for the bs output rows.
template <int BSV, int NE>
__global__ __launch_bounds__(256) void k_left_mult_E(Geo G, const double* __restrict__ Kst,
                                                     const double* __restrict__ X, double* __restrict__ OUT,
                                                     int rowPlane, double alpha) {
  __shared__ double es[BSV][NE];
  __shared__ int ks[NE];
  const long long cell = blockIdx.z;
  const int q = blockIdx.x;  // node in plane; rows q*BSV .. q*BSV+BSV-1 ; q >= npl: padding rows
  const long long per = (long long)G.Bp * G.Bp;
  double* out = OUT + cell * per;
  if (q * BSV >= G.b) {  // padding rows: zero
    for (int al = 0; al < BSV; ++al) {
      const int r = q * BSV + al;
      if (r < G.Bp)
        for (int c = threadIdx.x; c < G.Bp; c += 256) out[(long long)r * G.Bp + c] = 0.0;
    }
    return;
  }
  const int nipc = G.ncode / 3;
  if (threadIdx.x < NE) {
    const int j = threadIdx.x, ipc = j / BSV, be = j % BSV;
    const int node = q + G.npl * rowPlane;
    ks[j] = plane_neighbour(G, q, ipc < nipc ? ipc : 0) * BSV + be;
    for (int al = 0; al < BSV; ++al)
      es[al][j] = (ipc < nipc)
                      ? Kst[((cell * G.ncode + ipc) * BSV + al) * BSV * (long long)G.nn + (long long)be * G.nn + node]
                      : 0.0;
  }
  __syncthreads();
  const double* x = X + cell * per;
  for (int c = threadIdx.x; c < G.Bp; c += 256) {
    double acc[BSV];
#pragma unroll
    for (int al = 0; al < BSV; ++al) acc[al] = 0.0;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const double xv = x[(long long)ks[j] * G.Bp + c];
#pragma unroll
      for (int al = 0; al < BSV; ++al) acc[al] = fma(es[al][j], xv, acc[al]);
    }
#pragma unroll
    for (int al = 0; al < BSV; ++al) out[(long long)(q * BSV + al) * G.Bp + c] = alpha * acc[al];
  }
}

// XCD-aware ids for the strip kernels: the n workgroups (mesh rows) of one (row block, cell) unit read each other's input segments, so they
// should share an L2, i.e. sit on ONE XCD.  Workgroups go to the XCDs round-robin by linear id: linear id L -> XCD L % 8, mesh row
// (L / 8) % n, unit 8 (L / (8 n)) + L % 8.  (With mesh row = blockIdx.x the n neighbours landed on n different XCDs and every segment was
// fetched from HBM three times.)
__device__ __forceinline__ bool strip_ids(int n, int yblocks, long long nunits, int& jrow, int& by, long long& cell) {
  const unsigned L = blockIdx.x;
  const long long unit = 8ll * (L / (8u * n)) + (L & 7u);
  jrow = (int)((L >> 3) % (unsigned)n);
  if (unit >= nunits) return false;
  by = (int)(unit % yblocks);
  cell = unit / yblocks;
  return true;
}
inline unsigned strip_grid(int n, long long nunits) { return (unsigned)(((nunits + 7) / 8) * 8 * n); }

// Same product for 3D planes with n <= 16: one workgroup per MESH ROW of the plane (n nodes, n BSV output rows) and
// 32-column chunks.  The 3 n BSV input rows the strip depends on (mesh rows j-1, j, j+1) are staged through LDS once
// per chunk -- 3x read amplification instead of the 9x of the node-per-workgroup kernel -- with the next chunk in flight
// in registers; thread (i, cp) owns node i of the strip and columns 2 cp, 2 cp + 1.
template <int BSV>
__global__ __launch_bounds__(256, 3) void k_left_mult_E_strip(Geo G, const double* __restrict__ Kst,
                                                           const double* __restrict__ X, double* __restrict__ OUT,
                                                           int rowPlane, double alpha, long long ncells) {
  constexpr int CW = 32, NN = 9, SLMAX = 16 * BSV, LPT = (SLMAX * CW + 255) / 256;  // loads per thread per segment
  constexpr int NEB = NN * BSV * BSV;
  __shared__ double xs[3][SLMAX][CW];
  __shared__ double es[16][NEB];  // E of the strip's nodes: [node][neighbour][be][al]  (read as 16-lane broadcasts)
  const int tid = threadIdx.x, i = tid >> 4, cp = tid & 15;
  const int n = G.n, SL = n * BSV, Bp = G.Bp;
  int jrow, by_;
  long long cell;
  if (!strip_ids(n, 1, ncells, jrow, by_, cell)) return;
  const long long per = (long long)Bp * Bp;
  const double* x = X + cell * per;
  double* out = OUT + cell * per;
  const bool active = i < n;
  for (int el = tid; el < 16 * NEB; el += 256) {
    const int nd = el / NEB, rem = el % NEB, m = rem / (BSV * BSV), be = (rem / BSV) % BSV, al = rem % BSV;
    double v = 0.0;
    if (nd < n) {
      const int node = nd + n * jrow + G.npl * rowPlane;
      v = Kst[((cell * G.ncode + m) * BSV + al) * BSV * (long long)G.nn + (long long)be * G.nn + node];
    }
    es[nd][rem] = v;
  }
  int lrow[NN];  // LDS row of the neighbour's first component: (oy + 1) * SLMAX + i' * BSV
#pragma unroll
  for (int m = 0; m < NN; ++m) {
    const int ox = m % 3 - 1, oy = m / 3 - 1;
    lrow[m] = active ? (oy + 1) * SLMAX + ((i + ox + n) % n) * BSV : 0;
  }
  int grow[3];  // first global row of the three input segments
#pragma unroll
  for (int sgm = 0; sgm < 3; ++sgm) grow[sgm] = ((jrow + sgm - 1 + n) % n) * SL;
  double g[3][LPT];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
      for (int m = 0; m < LPT; ++m) {
        const int el = tid + 256 * m, r = el >> 5, col = el & 31;
        g[sgm][m] = (r < SL) ? x[(long long)(grow[sgm] + r) * Bp + c0 + col] : 0.0;
      }
  };
  auto stash = [&]() {
#pragma unroll
    for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
      for (int m = 0; m < LPT; ++m) {
        const int el = tid + 256 * m, r = el >> 5, col = el & 31;
        if (r < SLMAX) xs[sgm][r][col] = g[sgm][m];
      }
  };
  fetch(0);
  for (int c0 = 0; c0 < Bp; c0 += CW) {
    stash();
    __syncthreads();
    if (c0 + CW < Bp) fetch(c0 + CW);
    double acc[BSV][2];
#pragma unroll
    for (int al = 0; al < BSV; ++al) acc[al][0] = acc[al][1] = 0.0;
    int eo = i * NEB;
    asm volatile("" : "+v"(eo));  // keep the E reads in the loop: hoisted they cost 2 NEB VGPRs and a workgroup per CU
    const double* ei = &es[0][0] + eo;
    const double* base = &xs[0][0][0] + 2 * cp;
#pragma unroll
    for (int m = 0; m < NN; ++m) {
#pragma unroll
      for (int be = 0; be < BSV; ++be) {
        const double2 v = *reinterpret_cast<const double2*>(base + (lrow[m] + be) * CW);
#pragma unroll
        for (int al = 0; al < BSV; ++al) {
          const double ev = ei[(m * BSV + be) * BSV + al];
          acc[al][0] = fma(ev, v.x, acc[al][0]);
          acc[al][1] = fma(ev, v.y, acc[al][1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int al = 0; al < BSV; ++al) {
      acc[al][0] = pin_here(acc[al][0]);
      acc[al][1] = pin_here(acc[al][1]);
    }
    if (active) {
#pragma unroll
      for (int al = 0; al < BSV; ++al)
        *reinterpret_cast<double2*>(out + (long long)((i + n * jrow) * BSV + al) * Bp + c0 + 2 * cp) =
            double2{alpha * acc[al][0], alpha * acc[al][1]};
    }
    __syncthreads();
  }
  if (jrow == 0)  // padding rows b .. Bp-1 of the output are zero
    for (long long idx = (long long)G.b * Bp + tid; idx < per; idx += 256) out[idx] = 0.0;
}

// OUT = alpha IN E^T in the same strip form (the transposed twin of k_left_mult_E_strip): one workgroup per mesh row of
// the plane (its n BSV OUTPUT COLUMNS) and block of rows; 32 rows at a time, the 3 n BSV input columns of each go
// through LDS transposed ([column][row], pitch 34), so the inner loop is the 16 B-read / 16-lane-broadcast loop above.
// E stays in LDS (re-read per chunk: an empty asm hides the loop invariance) to keep 3 workgroups per CU.
template <int BSV>
__global__ __launch_bounds__(256, 2) void k_right_mult_Et_strip(Geo G, const double* __restrict__ Kst,
                                                                const double* __restrict__ IN, double* __restrict__ OUT,
                                                                int nrows, int rowPlane, double alpha, int codeOff,
                                                                int accumulate, int rowsPerBlock, int yblocks, long long ncells) {
  constexpr int CW = 32, CWP = 34, NN = 9, SLMAX = 16 * BSV;
  constexpr int NEB = NN * BSV * BSV;
  __shared__ alignas(16) double xs[3][SLMAX][CWP];
  __shared__ double es[16][NEB];
  __shared__ double ob[CW][SLMAX + 1];
  const int tid = threadIdx.x, i = tid >> 4, rp = tid & 15;
  const int n = G.n, SL = n * BSV, Bp = G.Bp;
  int jrow, by;
  long long cell;
  if (!strip_ids(n, yblocks, (long long)yblocks * ncells, jrow, by, cell)) return;
  const int kbeg = by * rowsPerBlock, kend = min(nrows, kbeg + rowsPerBlock);
  if (kbeg >= kend) return;
  const long long per = (long long)nrows * Bp;
  const double* in = IN + cell * per;
  double* out = OUT + cell * per;
  const bool active = i < n;
  for (int el = tid; el < 16 * NEB; el += 256) {
    const int nd = el / NEB, rem = el % NEB, m = rem / (BSV * BSV), be = (rem / BSV) % BSV, al = rem % BSV;
    double v = 0.0;
    if (nd < n && m < G.ncode / 3) {
      const int node = nd + n * jrow + G.npl * rowPlane;
      v = Kst[((cell * G.ncode + m + codeOff) * BSV + al) * BSV * (long long)G.nn + (long long)be * G.nn + node];
    }
    es[nd][rem] = v;
  }
  int lrow[NN];
#pragma unroll
  for (int m = 0; m < NN; ++m) {
    const int ox = m % 3 - 1, oy = m / 3 - 1;
    lrow[m] = (active ? (oy + 1) * SLMAX + ((i + ox + n) % n) * BSV : 0) * CWP + 2 * rp;
  }
  int gcol[3];  // first global column of the three input segments
#pragma unroll
  for (int sgm = 0; sgm < 3; ++sgm) gcol[sgm] = ((jrow + sgm - 1 + n) % n) * SL;
  const int fr = tid >> 4, fc = tid & 15;  // staging: rows fr, fr + 16 of the chunk, columns fc + 16 m (128 B runs)
  double g[3][2][BSV];
  auto fetch = [&](int k0) {
    // unconditional loads (clamped indices): rows >= kend and columns >= SL land in LDS slots no stored output reads
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      const double* src = in + (long long)min(k0 + fr + 16 * p2, kend - 1) * Bp;
#pragma unroll
      for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
        for (int m = 0; m < BSV; ++m) g[sgm][p2][m] = src[gcol[sgm] + min(fc + 16 * m, SL - 1)];
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
      for (int sgm = 0; sgm < 3; ++sgm)
#pragma unroll
        for (int m = 0; m < BSV; ++m) xs[sgm][fc + 16 * m][fr + 16 * p2] = g[sgm][p2][m];
  };
  fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += CW) {
    stash();
    __syncthreads();
    if (k0 + CW < kend) fetch(k0 + CW);
    double acc[BSV][2];
#pragma unroll
    for (int al = 0; al < BSV; ++al) acc[al][0] = acc[al][1] = 0.0;
    int eo = i * NEB;
    asm volatile("" : "+v"(eo));
    const double* ei = &es[0][0] + eo;
    const double* base = &xs[0][0][0];
#pragma unroll
    for (int m = 0; m < NN; ++m) {
#pragma unroll
      for (int be = 0; be < BSV; ++be) {
        const double2 v = *reinterpret_cast<const double2*>(base + lrow[m] + be * CWP);
#pragma unroll
        for (int al = 0; al < BSV; ++al) {
          const double ev = ei[(m * BSV + be) * BSV + al];
          acc[al][0] = fma(ev, v.x, acc[al][0]);
          acc[al][1] = fma(ev, v.y, acc[al][1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int al = 0; al < BSV; ++al) {  // the products are final here: do not let them sink into the guarded stores
      acc[al][0] = pin_here(acc[al][0]);
      acc[al][1] = pin_here(acc[al][1]);
    }
    // the 32 x SL output tile leaves through LDS so that the stores run along rows (64 B per 8 lanes) as the loads do
#pragma unroll
    for (int al = 0; al < BSV; ++al) {
      ob[2 * rp][i * BSV + al] = alpha * acc[al][0];
      ob[2 * rp + 1][i * BSV + al] = alpha * acc[al][1];
    }
    __syncthreads();
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
      const int rr = fr + 16 * p2;
      if (k0 + rr < kend) {
        double* o = out + (long long)(k0 + rr) * Bp + jrow * SL;
#pragma unroll
        for (int m = 0; m < BSV; ++m) {
          const int cc = fc + 16 * m;
          if (cc < SL) o[cc] = accumulate ? o[cc] + ob[rr][cc] : ob[rr][cc];
        }
      }
    }
  }
  if (jrow == 0 && !accumulate)  // padding columns b .. Bp-1 of the output are zero
    for (int k = kbeg; k < kend; ++k)
      for (int cc = G.b + tid; cc < Bp; cc += 256) out[(long long)k * Bp + cc] = 0.0;
}

// R[m][c] (+)= B[m][(c in plane)]  (16 x Bp load rows)
__global__ void k_add_P(Geo G, const double* __restrict__ Brhs, double* __restrict__ R, long long ncells, int plane,
                        int overwrite) {
  const long long per = 16ll * G.Bp;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * per) return;
  const long long cell = idx / per;
  const int rem = (int)(idx % per);
  const int c = rem % G.Bp, m = rem / G.Bp;
  double v = 0.0;
  if (m < G.t && c < G.b) {
    const int q = c / G.bs, al = c % G.bs;
    v = Brhs[cell * (long long)G.t * G.bs * G.nn + ((long long)m * G.bs + al) * G.nn + q + G.npl * plane];
  }
  if (overwrite) R[idx] = v;
  else R[idx] += v;
}

// gauge: drop the bs unknowns of the last node of the last plane (cell_problem.py:349-361: constants are the kernel)
__global__ void k_pin_last(Geo G, double* __restrict__ Sl, double* __restrict__ Rl, long long ncells) {
  const long long per = (long long)G.Bp * G.bs;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * per) return;
  const long long cell = idx / per;
  const int rem = (int)(idx % per);
  const int x = rem % G.Bp, p = G.b - G.bs + rem / G.Bp;
  double* S = Sl + cell * (long long)G.Bp * G.Bp;
  S[(long long)p * G.Bp + x] = (x == p) ? 1.0 : 0.0;
  S[(long long)x * G.Bp + p] = (x == p) ? 1.0 : 0.0;
  if (x < 16) Rl[cell * 16ll * G.Bp + (long long)x * G.Bp + p] = 0.0;
}

__global__ void k_finalize(Geo G, const double* __restrict__ C0, const double* __restrict__ Gm, double* __restrict__ out,
                           long long ncells) {
  const int tt = G.t * G.t;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * tt) return;
  const long long cell = idx / tt;
  const int m = (int)(idx % tt) / G.t, q = (int)(idx % tt) % G.t;
  out[idx] = C0[idx] - Gm[cell * 256 + m * 16 + q];
}

// corr[cell][m][plane * b + r] = X[cell][m][r]   (t load cases, periodic dof numbering (node, component))
__global__ void k_store_corr(Geo G, const double* __restrict__ X, double* __restrict__ corr, long long ncells, int plane) {
  const long long per = (long long)G.t * G.b;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * per) return;
  const long long cell = idx / per;
  const int rem = (int)(idx % per);
  const int r = rem % G.b, m = rem / G.b;
  corr[(cell * G.t + m) * (long long)G.nn * G.bs + (long long)plane * G.b + r] = X[cell * 16ll * G.Bp + (long long)m * G.Bp + r];
}

// remove the mean of every component (the reference projects the constants out: cell_problem.py:349-361, 382)
__global__ __launch_bounds__(256) void k_center_corr(Geo G, double* __restrict__ corr) {
  __shared__ double red[256];
  double* x = corr + (long long)blockIdx.x * G.nn * G.bs;  // one (cell, load case) per block
  for (int al = 0; al < G.bs; ++al) {
    double acc = 0.0;
    for (int p = threadIdx.x; p < G.nn; p += 256) acc += x[(long long)p * G.bs + al];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
      __syncthreads();
    }
    const double mean = red[0] / G.nn;
    __syncthreads();
    for (int p = threadIdx.x; p < G.nn; p += 256) x[(long long)p * G.bs + al] -= mean;
  }
}


// A[i][j] = A[j][i] for j > i  (mirror the lower triangle; batched, ld = N)
__global__ void k_symmetrize(int N, double* __restrict__ A, long long sA, long long ncells) {
  const long long per = (long long)N * N;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * per) return;
  const long long cell = idx / per;
  const int rem = (int)(idx % per);
  const int j = rem % N, i = rem / N;
  if (j > i) A[cell * sA + (long long)i * N + j] = A[cell * sA + (long long)j * N + i];
}

// OUT[i][j] = IN[j][i]  (sub-blocks, batched)
__global__ void k_transpose(int M, int N, const double* __restrict__ IN, int ldi, long long sI, double* __restrict__ OUT,
                            int ldo, long long sO, long long ncells) {
  const long long per = (long long)M * N;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * per) return;
  const long long cell = idx / per;
  const int rem = (int)(idx % per);
  const int j = rem % N, i = rem / N;
  OUT[cell * sO + (long long)i * ldo + j] = IN[cell * sI + (long long)j * ldi + i];
}

// in-place inverse of the NB x NB SPD diagonal sub-block at (off, off): one wavefront per cell
template <int NB>
__global__ __launch_bounds__(64) void k_leaf_inverse(double* __restrict__ S, int ld, long long stride, int off,
                                                     int32_t* __restrict__ info, int stepcode, int infoDiv) {
  constexpr int RPL = Cfg<NB>::RPL;
  __shared__ alignas(16) double ubuf[NB];
  __shared__ alignas(16) double wbuf[NB];
  const long long cell = blockIdx.x;
  const int l = threadIdx.x, c = l % NB, g = l / NB, r0 = g * RPL;
  double* P = S + cell * stride + (long long)off * ld + off;
  double s[RPL];
#pragma unroll
  for (int i = 0; i < RPL; ++i) s[i] = P[(long long)(r0 + i) * ld + c];
  int bad = 0;
  SweepStep<NB, 0>::run(s, ubuf, wbuf, c, g, r0, bad);
#pragma unroll
  for (int i = 0; i < RPL; ++i) P[(long long)(r0 + i) * ld + c] = -s[i];
  if (bad && l == 0 && info) atomicCAS(&info[cell / infoDiv], 0, stepcode);
}

// same for NB = 64 in the block layout of sweep_blk (lane = 8x8 block, 128 VGPRs of matrix): one launch instead of the
// two 32-leaves, four GEMMs and their launch latencies of a 64-node of the recursion.  Reads the LOWER triangle only
// (blocks above the diagonal are mirrored on the way in), writes the full symmetric inverse.
template <int NB>
__global__ __launch_bounds__(64) void k_leaf_inverse_blk(double* __restrict__ S, int ld, long long stride, int off,
                                                         int32_t* __restrict__ info, int stepcode, int infoDiv) {
  constexpr int BS = NB / 8;
  __shared__ alignas(16) double ubuf[NB];
  const long long cell = blockIdx.x;
  const int l = threadIdx.x, bi = l >> 3, bj = l & 7;
  double* P = S + cell * stride + (long long)off * ld + off;
  double s[BS * BS];
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int q = 0; q < BS; ++q) {
      const int row = BS * bi + r, col = BS * bj + q;
      s[r * BS + q] = (bi >= bj) ? P[(long long)row * ld + col] : P[(long long)col * ld + row];
    }
  int bad = 0;
  sweep_blk<NB>(s, ubuf, bi, bj, bad);
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int q = 0; q < BS; ++q) P[(long long)(BS * bi + r) * ld + BS * bj + q] = -s[r * BS + q];
  if (bad && l == 0 && info) atomicCAS(&info[cell / infoDiv], 0, stepcode);
}

// two-phase media: expand (mask, per-cell phase values) into the element stream the assembly reads
__global__ void k_expand_two_phase(const unsigned char* __restrict__ mask, const double* __restrict__ values,
                                   double* __restrict__ coef, long long n_el, int n_comp, long long ncells) {
  const long long per = n_el * n_comp;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * per) return;
  const long long cell = idx / per;
  const long long rem = idx % per;
  const long long el = rem / n_comp;
  const int comp = (int)(rem % n_comp);
  coef[idx] = values[(cell * 2 + (mask[el] ? 1 : 0)) * n_comp + comp];
}

hipError_t launch_expand_two_phase(const unsigned char* d_mask, const double* d_values, double* d_coef, long long n_el,
                                   int n_comp, long long ncells, hipStream_t stream) {
  const long long work = ncells * n_el * n_comp;
  if (work <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_expand_two_phase, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, d_mask, d_values, d_coef,
                     n_el, n_comp, ncells);
  return hipGetLastError();
}

// separable coefficients (kernels.h): same arithmetic, operation by operation, as the fused kernel's sampler
__global__ void k_expand_separable(CoefSource src, const double* __restrict__ params, double* __restrict__ coef, long long n_el,
                                   int n_comp, long long ncells) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ncells * n_el * n_comp) return;
  const int comp = (int)(idx % n_comp);
  const long long el = (idx / n_comp) % n_el, cell = idx / (n_comp * n_el);
  const double a = params[(cell * n_comp + comp) * 2], b = params[(cell * n_comp + comp) * 2 + 1];  // (a, b) of this component
  const double* table = static_cast<const double*>(src.table);
  if (src.mode == COEF_AFFINE) {
    coef[idx] = add_rn(a, mul_rn(b, table[el]));
  } else {
    double acc = 0.0;
    for (int q = 0; q < src.nq; ++q)
      acc = add_rn(acc, mul_rn(src.weights[q], div_rn(1.0, add_rn(a, mul_rn(b, table[el * src.nq + q])))));
    coef[idx] = acc;
  }
}

hipError_t launch_expand_separable(CoefSource src, const double* d_params, double* d_coef, long long n_el, int n_comp, long long ncells,
                                   hipStream_t stream) {
  const long long work = ncells * n_el * n_comp;
  if (work <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_expand_separable, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, src, d_params, d_coef, n_el, n_comp,
                     ncells);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// host orchestration
// ---------------------------------------------------------------------------------------------------------------
static void fill_tables(Geo& G) {
  static const int tri[2][3][2] = {{{0, 0}, {1, 0}, {1, 1}}, {{0, 0}, {0, 1}, {1, 1}}};
  static const int vb[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
  static const int tet[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
  const int d = G.dim, nv = d + 1;
  for (int s = 0; s < G.nsub; ++s) {
    double X[4][3] = {};
    for (int a = 0; a < nv; ++a)
      for (int k = 0; k < 3; ++k) {
        int o = 0;
        if (k < d) o = (d == 2) ? tri[s][a][k] : vb[tet[s][a]][k];
        G.voff[s][a][k] = o;
        X[a][k] = o;
      }
    // gradients: solve [1 X] coefficients; grad_a = column a of inv([1 X])^T rows 1..d  -> use Cramer via small Gauss-Jordan
    double Aug[4][8] = {};
    for (int a = 0; a < nv; ++a) {
      Aug[a][0] = 1.0;
      for (int k = 0; k < d; ++k) Aug[a][1 + k] = X[a][k];
      Aug[a][nv + a] = 1.0;
    }
    for (int p = 0; p < nv; ++p) {
      int piv = p;
      for (int r = p + 1; r < nv; ++r)
        if (std::fabs(Aug[r][p]) > std::fabs(Aug[piv][p])) piv = r;
      for (int q = 0; q < 2 * nv; ++q) std::swap(Aug[p][q], Aug[piv][q]);
      const double dd = Aug[p][p];
      for (int q = 0; q < 2 * nv; ++q) Aug[p][q] /= dd;
      for (int r = 0; r < nv; ++r)
        if (r != p) {
          const double f = Aug[r][p];
          for (int q = 0; q < 2 * nv; ++q) Aug[r][q] -= f * Aug[p][q];
        }
    }
    // inverse Minv = Aug[:, nv:], lambda_a(x) = Minv[0][a] + sum_k Minv[1+k][a] x_k
    for (int a = 0; a < nv; ++a)
      for (int k = 0; k < 3; ++k) G.grad[s][a][k] = (k < d) ? Aug[1 + k][nv + a] : 0.0;
  }
}

int blocked_workspace_create(BlockedWorkspace** out, int dim, int n, int kind) {
  *out = nullptr;
  BlockedWorkspace* ws = new BlockedWorkspace();
  Geo& G = ws->G;
  G.dim = dim;
  G.n = n;
  G.kind = kind;
  const bool el = kind >= HOMMX_KIND_ELASTICITY_ISO;
  G.bs = el ? dim : 1;
  G.t = el ? dim * (dim + 1) / 2 : dim;
  G.ncomp = kind == HOMMX_KIND_POISSON_SCALAR ? 1
            : kind == HOMMX_KIND_POISSON_MATRIX ? dim * (dim + 1) / 2
            : kind == HOMMX_KIND_ELASTICITY_ISO ? 2
                                                : G.t * (G.t + 1) / 2;
  G.nn = dim == 2 ? n * n : n * n * n;
  G.npl = dim == 2 ? n : n * n;
  G.b = G.bs * G.npl;
  G.Bp = (G.b + 31) / 32 * 32;
  G.nsub = dim == 2 ? 2 : 6;
  G.ncode = dim == 2 ? 9 : 27;
  G.n_el = G.nsub * G.nn;
  fill_tables(G);
  if (const char* e = getenv("HOMMX_BLOCKED_MEM_GB")) ws->budget_gb_env = atof(e);
  if (const char* e = getenv("HOMMX_GEMM128_MIN")) ws->gemm128_min = atoi(e);
  ws->sparse_v1 = getenv("HOMMX_SPARSE_V1") != nullptr;
  ws->leaf32 = getenv("HOMMX_LEAF32") != nullptr;
  ws->split64 = getenv("HOMMX_NO_SPLIT64") == nullptr;
  ws->small_fused = getenv("HOMMX_NO_SMALL_FUSED") == nullptr;
  if (const char* e = getenv("HOMMX_SMALL_WAVES")) ws->small_waves = atoi(e);
  // route: nested dissection (multifrontal.hip) wherever it beats the plane elimination (profiles/r03_kinds_routes.txt) -- every plane
  // block b > 64, i.e. everything the one-launch kernels do not take: 2D scalar n = 80: +51 %, 2D elasticity n = 36: +92 %, 3D elasticity
  // n = 5: +68 %, n = 16: +76 %, scalar 3D n = 9: +18 % (it lost 10 % there before the build kernel batched its loads and the route ran on
  // two streams)
  // Round 4: with the register-resident front kernel (mf_front_kernel.h: one launch per tree level, fronts never leave the registers) the tree
  // also beats the LDS kernel of csrc/small_fused.h on 2D meshes with 48 < b <= 64: 2D Poisson 64^2 175 k -> 256 k solves/s, 2D elasticity
  // 32^2 320 k -> 336 k; it loses on 3D Poisson 8^3 (1.25 M -> 0.79 M: few, larger fronts) and against the one-wave kernel (b <= 48).
  ws->mf_min_b = dim == 2 ? 49 : 65;
  bool mf_env = false;
  if (const char* e = getenv("HOMMX_MF_MIN_B")) {
    ws->mf_min_b = atoi(e);
    mf_env = true;
  }
  if (const char* e = getenv("HOMMX_MF_G128_MIN_K")) ws->mf_gather128_min_k = atoi(e);
  ws->mf_no_border_split = getenv("HOMMX_MF_NO_BORDER_SPLIT") != nullptr;
  if (const char* e = getenv("HOMMX_MF_CORR")) ws->mf_corr = atoi(e) != 0;
  if (const char* e = getenv("HOMMX_TILE_SB")) ws->tile_sb = atoi(e);
  // (a threshold from the environment below 65 takes effect only together with HOMMX_NO_SMALL_FUSED: A/B runs)
  if (ws->mf_min_b > 0 && G.b >= ws->mf_min_b && (G.b > 64 || !ws->small_fused || (!mf_env && G.b > 48))) {
    if (int rc = mf_plan_create(&ws->mf, G)) {
      delete ws;
      return rc;
    }
    // the products of this route are tall and thin or lower-triangular: with the super-block tile order the 64 x 64 tiles (six workgroups
    // per CU) are never slower than the 128 x 128 ones any more -- C4 +2 %, 3D elasticity 20^3 +2 %, scalar 24^3 -1.5 %
    if (!getenv("HOMMX_GEMM128_MIN")) ws->gemm128_min = 1 << 30;
  }
  *out = ws;
  return 0;
}

const char* blocked_route_name(const BlockedWorkspace* ws) {
  if (!ws) return "blocked";
  if (ws->mf) return "multifrontal";
  if (ws->G.b <= 64 && ws->small_fused) return (ws->G.b <= 48 && ws->small_waves != 2 && ws->small_waves != 4) ? "small_wave" : "small_fused";
  return "blocked";
}

// one line for reports (bench.py's roofline.kernel): what the route launches, derived from the plan itself
const char* blocked_route_detail(BlockedWorkspace* ws) {
  if (!ws) return "";
  if (ws->detail.empty()) {
    char buf[512];
    const Geo& G = ws->G;
    if (ws->mf) ws->detail = mf_describe(ws, ws->mf);
    else if (G.b <= 64 && ws->small_fused) {
      snprintf(buf, sizeof(buf), "%s: one launch after K1 (k_assemble_reg), plane block b = %d, f64 MFMA 16x16x4 tiles in %s", blocked_route_name(ws), G.b,
               (G.b <= 48 && ws->small_waves != 2 && ws->small_waves != 4) ? "registers (one wavefront per macro cell)" : "LDS (several waves per macro cell)");
      ws->detail = buf;
    } else {
      snprintf(buf, sizeof(buf), "blocked: plane elimination, b = %d (padded %d), k_gemm_tile %s f64-MFMA tiles, recursive block inverse on 32 / 64 leaves, strip-form sparse products",
               G.b, G.Bp, G.Bp >= ws->gemm128_min ? "128x128 (8 waves) and 64x64 (4 waves)" : "64x64 (4 waves)");
      ws->detail = buf;
    }
  }
  return ws->detail.c_str();
}

double blocked_flops_per_cell(const BlockedWorkspace* ws) {
  if (!ws) return 0.0;
  if (ws->mf) return mf_flops_per_cell(ws->mf);
  const double b = ws->G.b;
  return (6.0 * (ws->G.n - 1) + 2.0) * b * b * b;
}

static void ws_free_main(BlockedWorkspace* ws) {
  double** ptrs[] = {&ws->Kst, &ws->Brhs, &ws->C0, &ws->S, &ws->W, &ws->Sl, &ws->V, &ws->X, &ws->T, &ws->R, &ws->Rl, &ws->Vr, &ws->Gm};
  for (auto p : ptrs) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  ws->chunk = 0;
}

static void ws_free_hist(BlockedWorkspace* ws) {
  double** ptrs[] = {&ws->hS, &ws->hW, &ws->hR, &ws->Xa, &ws->Xb, &ws->Y};
  for (auto p : ptrs) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  ws->hchunk = 0;
}

static void ws_free(BlockedWorkspace* ws) {
  ws_free_main(ws);
  ws_free_hist(ws);
}

void blocked_workspace_destroy(BlockedWorkspace* ws) {
  if (!ws) return;
  if (ws->mf) mf_plan_destroy(ws->mf);
  if (ws->mf_keep) mf_plan_destroy(ws->mf_keep);
  for (auto& kv : ws->tilemaps) (void)hipFree(kv.second);
  ws_free(ws);
  delete ws;
}

static long long per_cell_bytes(const Geo& G) {
  const long long mat = (long long)G.Bp * G.Bp;
  return 8ll * ((long long)G.ncode * G.bs * G.bs * G.nn + (long long)G.t * G.bs * G.nn + 36 + 6 * mat + 3 * 16ll * G.Bp + 256);
}

static int ws_reserve(BlockedWorkspace* ws, long long ncells, bool correctors) {
  const Geo& G = ws->G;
  // workspace budget: the batch kernels keep gaining up to ~1000 cells in flight (small launches of the recursive
  // inverse amortise), and the card has 288 GB: take up to 64 GB, never more than half of what is free
  double budget_gb = 64.0;
  {
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget_gb = std::min(budget_gb, 0.5e-9 * (double)fr);
  }
  if (ws->budget_gb_env > 0.0) budget_gb = ws->budget_gb_env;
  const long long hist_bytes = 8ll * (G.n - 1) * (2ll * G.Bp * G.Bp + 16ll * G.Bp) + 8ll * 3 * 16 * G.Bp;
  if (correctors) {
    long long hc = (long long)(budget_gb * 1e9) / (per_cell_bytes(G) + hist_bytes);
    if (hc < 1) hc = 1;
    if (hc > 8192) hc = 8192;
    if (hc > ncells) hc = ncells;
    if (hc > ws->hchunk) {
      ws_free_hist(ws);
      const long long mat = (long long)G.Bp * G.Bp;
      BTRY(hipMalloc(&ws->hS, 8ll * hc * (G.n - 1) * mat));
      BTRY(hipMalloc(&ws->hW, 8ll * hc * (G.n - 1) * mat));
      BTRY(hipMalloc(&ws->hR, 8ll * hc * (G.n - 1) * 16 * G.Bp));
      BTRY(hipMalloc(&ws->Xa, 8ll * hc * 16 * G.Bp));
      BTRY(hipMalloc(&ws->Xb, 8ll * hc * 16 * G.Bp));
      BTRY(hipMalloc(&ws->Y, 8ll * hc * 16 * G.Bp));
      ws->hchunk = hc;
    }
  }
  long long chunk = (long long)(budget_gb * 1e9) / per_cell_bytes(G);
  if (chunk < 1) chunk = 1;
  if (chunk > 8192) chunk = 8192;
  if (chunk > ncells) chunk = ncells;
  if (chunk <= ws->chunk) return 0;
  ws_free_main(ws);
  const long long mat = (long long)G.Bp * G.Bp;
  BTRY(hipMalloc(&ws->Kst, 8ll * chunk * G.ncode * G.bs * G.bs * G.nn));
  BTRY(hipMalloc(&ws->Brhs, 8ll * chunk * G.t * G.bs * G.nn));
  BTRY(hipMalloc(&ws->C0, 8ll * chunk * 36));
  BTRY(hipMalloc(&ws->S, 8ll * chunk * mat));
  BTRY(hipMalloc(&ws->W, 8ll * chunk * mat));
  BTRY(hipMalloc(&ws->Sl, 8ll * chunk * mat));
  BTRY(hipMalloc(&ws->V, 8ll * chunk * mat));
  BTRY(hipMalloc(&ws->X, 8ll * chunk * mat));
  BTRY(hipMalloc(&ws->T, 8ll * chunk * mat));
  BTRY(hipMalloc(&ws->R, 8ll * chunk * 16 * G.Bp));
  BTRY(hipMalloc(&ws->Rl, 8ll * chunk * 16 * G.Bp));
  BTRY(hipMalloc(&ws->Vr, 8ll * chunk * 16 * G.Bp));
  BTRY(hipMalloc(&ws->Gm, 8ll * chunk * 256));
  ws->chunk = chunk;
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------
// batched fp64 MFMA GEMM  C = alpha op(A) op(B) + beta C  (v_mfma_f64_16x16x4_f64), TM x TM tile per
// workgroup of NW waves in a 2 x NW/2 grid: TM = 128, NW = 8 (64 x 32 per wave; 16 flop per byte of L2 -> LDS
// traffic) for M, N >= 256, TM = 64, NW = 4 for the small levels of the recursive inverse and the 16-row load
// products.  K is staged 16 at a time with the next stage prefetched into registers while the current one is
// multiplied; LDS pitch TM + 16 doubles (== 32 dwords mod 64: conflict-free ds_read_b64 fragments).  The grid is
// one-dimensional and XCD-aware: workgroup g runs on XCD g % 8 (round-robin dispatch), so
// cell = 8 * (slot / T) + g % 8 keeps ALL tiles of one cell on one XCD's 4 MB L2; symmetric updates enumerate the
// lower-triangle tiles only; an optional mirrored store (Ct) writes C^T as well, which replaces transpose passes.
// ---------------------------------------------------------------------------------------------------------------
template <bool TA, bool TB, int TM, int NW, bool GATHER = false>
__global__ __launch_bounds__(64 * NW, 2) void k_gemm_tile(int M, int N, int K, double alpha, const double* __restrict__ A,
                                                    int lda, long long sA, const double* __restrict__ B, int ldb,
                                                    long long sB, double beta, double* __restrict__ C, int ldc,
                                                    long long sC, int lowerOnly, int nc, int tilesX, int tilesPerCell,
                                                    double* Ct, GatherC ga = GatherC(), const int* __restrict__ tilemap = nullptr) {
  constexpr int PITCH = TM + 16;  // 2 PITCH dwords == 32 mod 64 for TM = 64 and 128: conflict-free ds_read_b64 fragments
  constexpr int WTM = TM / 2, WTN = TM / (NW / 2);  // per-wave tile: waves form a 2 x (NW / 2) grid
  constexpr int NFA = WTM / 16, NFB = WTN / 16;     // 16x16 MFMA tiles per wave, rows / columns
  constexpr int PT = TM * 16 / (64 * NW);           // doubles per thread, operand and 16-deep stage
  __shared__ double As[16 * PITCH];
  __shared__ double Bs[16 * PITCH];
  const int g = blockIdx.x, slot = g >> 3;
  const long long cell = 8ll * (slot / tilesPerCell) + (g & 7);
  if (cell >= nc) return;
  const int tile = slot % tilesPerCell;
  int ty, tx;
  if (tilemap) {  // lower triangle in super-blocks (gemm): the panels of a block stay in the XCD's L2
    ty = tilemap[tile] >> 16;
    tx = tilemap[tile] & 0xffff;
  } else if (lowerOnly) {  // tiles of the lower triangle, row by row
    ty = 0;
    while ((ty + 1) * (ty + 2) / 2 <= tile) ++ty;
    tx = tile - ty * (ty + 1) / 2;
  } else {
    ty = tile / tilesX;
    tx = tile % tilesX;
  }
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int m0 = ty * TM, n0 = tx * TM;
  A += cell * sA;
  B += cell * sB;
  C += cell * sC;
  if (Ct) Ct += cell * sC;
  const int wi0 = WTM * (w / (NW / 2)), wj0 = WTN * (w % (NW / 2));
  const int l15 = l & 15, l4 = l >> 4;
  d4 acc[NFA][NFB];
#pragma unroll
  for (int a = 0; a < NFA; ++a)
#pragma unroll
    for (int b = 0; b < NFB; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

  // staging maps.  "row-major along k" operand (A not transposed / B transposed): thread -> row t >> 1, 8 k's;
  // "k-major" operand (A transposed / B not transposed): thread -> k = t >> 4, 8 consecutive rows.
  constexpr int TPR = 16 / PT;   // threads per tile row in the row-major-along-k map
  constexpr int TPK = TM / PT;   // threads per k-row in the k-major map
  const int rk_row = tid / TPR, rk_k = (tid % TPR) * PT;
  const int km_k = tid / TPK, km_row = (tid % TPK) * PT;
  double pa[PT], pb[PT];
  auto fetch = [&](int k0) {
    const double* p;
    bool ok;
    if (!TA) { ok = m0 + rk_row < M; p = A + (long long)(m0 + rk_row) * lda + k0 + rk_k; }
    else     { ok = m0 + km_row < M; p = A + (long long)(k0 + km_k) * lda + m0 + km_row; }
#pragma unroll
    for (int x = 0; x < PT; x += 2) {
      double2 v = double2{0.0, 0.0};
      if (ok) v = *reinterpret_cast<const double2*>(p + x);
      pa[x] = v.x; pa[x + 1] = v.y;
    }
    if (TB) { ok = n0 + rk_row < N; p = B + (long long)(n0 + rk_row) * ldb + k0 + rk_k; }
    else    { ok = n0 + km_row < N; p = B + (long long)(k0 + km_k) * ldb + n0 + km_row; }
#pragma unroll
    for (int x = 0; x < PT; x += 2) {
      double2 v = double2{0.0, 0.0};
      if (ok) v = *reinterpret_cast<const double2*>(p + x);
      pb[x] = v.x; pb[x + 1] = v.y;
    }
  };
  auto stash = [&]() {
    if (!TA) {
#pragma unroll
      for (int x = 0; x < PT; ++x) As[(rk_k + x) * PITCH + rk_row] = pa[x];
    } else {
#pragma unroll
      for (int x = 0; x < PT; x += 2) *reinterpret_cast<double2*>(&As[km_k * PITCH + km_row + x]) = double2{pa[x], pa[x + 1]};
    }
    if (TB) {
#pragma unroll
      for (int x = 0; x < PT; ++x) Bs[(rk_k + x) * PITCH + rk_row] = pb[x];
    } else {
#pragma unroll
      for (int x = 0; x < PT; x += 2) *reinterpret_cast<double2*>(&Bs[km_k * PITCH + km_row + x]) = double2{pb[x], pb[x + 1]};
    }
  };

  fetch(0);
  stash();
  __syncthreads();
  // a diagonal tile of a GATHERING lower-triangle update (its epilogue stores nothing above the diagonal; the plain epilogue stores diagonal
  // tiles whole, and the recursive inverse reads them whole): the wave(s) whose sub-tile lies strictly above the diagonal -- wave 1 of the
  // 2 x 2 grid of a 64-tile -- keep staging operands and keeping the barriers, but issue no LDS reads and no MFMAs
  const bool idle = GATHER && lowerOnly && tx == ty && wj0 >= wi0 + WTM;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const bool more = k0 + 16 < K;
    if (more) fetch(k0 + 16);
    if (!idle)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double af[NFA], bf[NFB];
#pragma unroll
      for (int a = 0; a < NFA; ++a) af[a] = As[(4 * ks + l4) * PITCH + wi0 + 16 * a + l15];
#pragma unroll
      for (int b = 0; b < NFB; ++b) bf[b] = Bs[(4 * ks + l4) * PITCH + wj0 + 16 * b + l15];
#pragma unroll
      for (int a = 0; a < NFA; ++a)
#pragma unroll
        for (int b = 0; b < NFB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      stash();
      __syncthreads();
    }
  }
  if constexpr (GATHER) {
    // multifrontal extend-add fused into the Schur update: C_out = sum over the child slots of U_child[map(row)][map(col)] + alpha acc
    // (valid entries of a child's update matrix are those on and below its diagonal: read through (max, min))
    const int f = (int)((cell + ga.batch0) % ga.nf);
    const long long mcell = (cell + ga.batch0) / ga.nf;
#pragma unroll
    for (int a = 0; a < NFA; ++a)
#pragma unroll
      for (int b = 0; b < NFB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] *= alpha;
#pragma unroll 1
    for (int slot = 0; slot < 2; ++slot) {
      const MfChild ch = ga.child[f * 2 + slot];
      if (!ch.valid) continue;
      const int32_t* dp = ga.dpos + ((long long)f * 2 + slot) * ga.rp;
      const double* U = ga.arena + ga.nc * ch.offF + ((mcell * ch.nf + ch.fidx) * (long long)ch.L + ch.sp) * ch.L + ch.sp;
      int pc[NFB];
#pragma unroll
      for (int b = 0; b < NFB; ++b) {
        const int col = n0 + wj0 + 16 * b + l15;
        pc[b] = col < N ? dp[col] : -1;
      }
#pragma unroll
      for (int a = 0; a < NFA; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wi0 + 16 * a + l4 + 4 * r;
          const int pr = row < M ? dp[row + ga.rowOff] : -1;
          if (pr < 0) continue;
#pragma unroll
          for (int b = 0; b < NFB; ++b)
            if (pc[b] >= 0 && !(lowerOnly && n0 + wj0 + 16 * b + l15 > row)) {
              const int hi = pr > pc[b] ? pr : pc[b], lo = pr > pc[b] ? pc[b] : pr;
              acc[a][b][r] += U[(long long)hi * ch.L + lo];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NFA; ++a)
#pragma unroll
      for (int b = 0; b < NFB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wi0 + 16 * a + l4 + 4 * r, col = n0 + wj0 + 16 * b + l15;
          if (row < M && col < N && !(lowerOnly && col > row)) C[(long long)row * ldc + col] = acc[a][b][r];  // above the diagonal: never read
        }
    return;
  }
#pragma unroll
  for (int a = 0; a < NFA; ++a)
#pragma unroll
    for (int b = 0; b < NFB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wi0 + 16 * a + l4 + 4 * r, col = n0 + wj0 + 16 * b + l15;
        if (row < M && col < N) {
          double* p = C + (long long)row * ldc + col;
          double v = alpha * acc[a][b][r];
          if (beta != 0.0) v += beta * *p;
          *p = v;
          // mirrored copy (same leading dimension and batch stride as C).  With lowerOnly, Ct may be C itself: the tiles
          // below the diagonal then fill the ones above; diagonal tiles are complete and are not mirrored (two lanes
          // would write the same entry with values that differ in the last bit).
          if (Ct && !(lowerOnly && tx == ty)) Ct[(long long)col * ldc + row] = v;
        }
      }
}

namespace {
inline unsigned nblk(long long work, int bs = 256) { return (unsigned)((work + bs - 1) / bs); }
}  // namespace

// Super-block tile order of a lower triangle of `ty` tile rows (device table, cached per tile count).  The multifrontal route builds the
// tables of all its groups when its workspace is reserved (mf_reserve), so that no allocation or blocking copy happens while streams are
// being filled; a first use from anywhere else makes the table here.  A failed allocation falls back to the row-by-row order and leaves no
// sticky HIP error behind.
const int* ensure_tilemap(BlockedWorkspace* ws, int ty) {
  auto it = ws->tilemaps.find(ty);
  if (it != ws->tilemaps.end()) return it->second;
  const int SB = ws->tile_sb;
  std::vector<int> order;
  order.reserve((size_t)ty * (ty + 1) / 2);
  for (int I = 0; I < ty; I += SB)
    for (int J = 0; J <= I; J += SB)
      for (int i = I; i < std::min(I + SB, ty); ++i)
        for (int j = J; j < std::min(J + SB, ty) && j <= i; ++j) order.push_back(i << 16 | j);
  int* d = nullptr;
  if (hipMalloc(&d, sizeof(int) * order.size()) == hipSuccess &&
      hipMemcpy(d, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice) == hipSuccess)
    return ws->tilemaps.emplace(ty, d).first->second;
  if (d) (void)hipFree(d);
  (void)hipGetLastError();  // the fallback is legitimate: do not let the failure surface later as somebody else's error
  return nullptr;
}

// tile size gemm() picks for an M x N (x K) product of this workspace
static inline bool gemm_big(const BlockedWorkspace* ws, int M, int N, int K, bool gather) {
  return M >= ws->gemm128_min && N >= ws->gemm128_min && !(gather && K < ws->mf_gather128_min_k);
}
int gemm_tile_size(const BlockedWorkspace* ws, int M, int N, int K, bool gather) { return gemm_big(ws, M, N, K, gather) ? 128 : 64; }

void gemm(const Ctx& c, bool ta, bool tb, int M, int N, int K, double alpha, const double* A, int lda, long long sA,
          const double* B, int ldb, long long sB, double beta, double* C, int ldc, long long sC, int lowerOnly, double* Ct,
          const GatherC* gather) {
  // gemm128_min is a dev knob: smallest M, N routed to the 128x128 tiles (tests lower it to cover partial tiles); a gathering update of
  // small rank is bound by the traffic of the tiles it touches: 64-tiles waste less of the lower triangle
  const bool big = gemm_big(c.ws, M, N, K, gather != nullptr);
  const int TM = big ? 128 : 64;
  const int tx = (N + TM - 1) / TM, ty = (M + TM - 1) / TM;
  const int T = lowerOnly ? ty * (ty + 1) / 2 : tx * ty;
  const long long groups = (c.nc + 7) / 8;
  {  // a launch holds at most 2^32 - 1 work-items (AQL grid size): huge batches go in pieces
    const long long max_groups = std::max(1ll, (1ll << 30) / (8ll * T * (big ? 512 : 256)));
    if (groups > max_groups) {
      for (long long g0 = 0; g0 < groups; g0 += max_groups) {
        Ctx sub = c;
        const long long b0 = g0 * 8;
        sub.nc = std::min(c.nc - b0, max_groups * 8);
        GatherC gs;
        if (gather) {
          gs = *gather;
          gs.batch0 = gather->batch0 + b0;
        }
        gemm(sub, ta, tb, M, N, K, alpha, A + b0 * sA, lda, sA, B + b0 * sB, ldb, sB, beta, C + b0 * sC, ldc, sC, lowerOnly,
             Ct ? Ct + b0 * sC : nullptr, gather ? &gs : nullptr);
      }
      return;
    }
  }
  // Big lower-triangle updates walk their tiles in SB x SB super-blocks: row by row a tile row of a 1,536-front touches 8 MB of B panels,
  // twice an XCD's L2, and every panel is fetched once per tile (the rank-672 update of C4 fetched 143 MB per cell for 52 MB of operands)
  const int* tilemap = (lowerOnly && c.ws->tile_sb > 1 && ty >= 2 * c.ws->tile_sb) ? ensure_tilemap(c.ws, ty) : nullptr;
  dim3 grid((unsigned)(groups * 8 * T));
  // 128 tiles: 8 waves per workgroup (2 x 4 grid of 64 x 32 wave tiles, 110 VGPRs, 4 waves per SIMD): +2 % over 4 waves
  // of 64 x 64; 64 tiles: 4 waves of 32 x 32 (8 waves measured slower)
#define HOMMX_GT(TA_, TB_)                                                                                                  \
  do {                                                                                                                      \
    if (big)                                                                                                                \
      hipLaunchKernelGGL((k_gemm_tile<TA_, TB_, 128, 8>), grid, dim3(512), 0, c.st, M, N, K, alpha, A, lda, sA, B, ldb, sB,   \
                         beta, C, ldc, sC, lowerOnly, (int)c.nc, tx, T, Ct, GatherC(), tilemap);                            \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_gemm_tile<TA_, TB_, 64, 4>), grid, dim3(256), 0, c.st, M, N, K, alpha, A, lda, sA, B, ldb, sB,    \
                         beta, C, ldc, sC, lowerOnly, (int)c.nc, tx, T, Ct, GatherC(), tilemap);                            \
  } while (0)
  if (gather) {  // virtual C (multifrontal.hip): NN only
    if (big)
      hipLaunchKernelGGL((k_gemm_tile<false, false, 128, 8, true>), grid, dim3(512), 0, c.st, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C,
                         ldc, sC, lowerOnly, (int)c.nc, tx, T, Ct, *gather, tilemap);
    else
      hipLaunchKernelGGL((k_gemm_tile<false, false, 64, 4, true>), grid, dim3(256), 0, c.st, M, N, K, alpha, A, lda, sA, B, ldb, sB, beta, C,
                         ldc, sC, lowerOnly, (int)c.nc, tx, T, Ct, *gather, tilemap);
    return;
  }
  if (!ta && !tb) HOMMX_GT(false, false);
  else if (!ta && tb) HOMMX_GT(false, true);
  else if (ta && !tb) HOMMX_GT(true, false);
  else HOMMX_GT(true, true);
#undef HOMMX_GT
}

namespace {
void right_mult_Et(const Ctx& c, const double* IN, double* OUT, int nrows, int rowPlane, double alpha,
                   int olast = -1, int accumulate = 0) {
  const Geo& G = c.ws->G;
  constexpr int RT = 32;
  const int nodes = (G.Bp + G.bs - 1) / G.bs;
  dim3 grid((nodes + 255) / 256, (nrows + RT - 1) / RT, (unsigned)c.nc), block(256);
  const int ne = G.bs * (G.ncode / 3);
  const int codeOff = (olast + 1) * (G.ncode / 3);
  if (G.dim == 3 && G.n <= 16 && (G.bs == 1 || G.bs == 3) && !c.ws->sparse_v1) {  // strip kernel (HOMMX_SPARSE_V1: dev knob, generic kernels)
    int rpb = (nrows + 31) / 32 * 32;  // rows per workgroup: as many as still leave ~4 workgroups per slot
    while (rpb > 32 && (long long)((nrows + rpb - 1) / rpb) * c.nc * G.n < 4096) rpb = (rpb / 2 + 31) / 32 * 32;
    const int yb = (nrows + rpb - 1) / rpb;
    dim3 g2(strip_grid(G.n, (long long)yb * c.nc));
    if (G.bs == 1)
      hipLaunchKernelGGL((k_right_mult_Et_strip<1>), g2, block, 0, c.st, G, c.ws->Kst, IN, OUT, nrows, rowPlane, alpha, codeOff, accumulate, rpb, yb, c.nc);
    else
      hipLaunchKernelGGL((k_right_mult_Et_strip<3>), g2, block, 0, c.st, G, c.ws->Kst, IN, OUT, nrows, rowPlane, alpha, codeOff, accumulate, rpb, yb, c.nc);
    return;
  }
#define HOMMX_RM(BSV, NE) hipLaunchKernelGGL((k_right_mult_Et<BSV, NE, RT>), grid, block, 0, c.st, G, c.ws->Kst, IN, OUT, nrows, rowPlane, alpha, codeOff, accumulate)
  if (ne == 3) HOMMX_RM(1, 3);
  else if (ne == 6) HOMMX_RM(2, 6);
  else if (ne == 9) HOMMX_RM(1, 9);
  else HOMMX_RM(3, 27);
#undef HOMMX_RM
}

void left_mult_E(const Ctx& c, const double* X, double* OUT, int rowPlane, double alpha) {
  const Geo& G = c.ws->G;
  dim3 grid((G.Bp + G.bs - 1) / G.bs, 1, (unsigned)c.nc), block(256);
  const int ne = G.bs * (G.ncode / 3);
  if (G.dim == 3 && G.n <= 16 && (G.bs == 1 || G.bs == 3) && !c.ws->sparse_v1) {  // strip kernel
    dim3 g2(strip_grid(G.n, c.nc));
    if (G.bs == 1) hipLaunchKernelGGL((k_left_mult_E_strip<1>), g2, block, 0, c.st, G, c.ws->Kst, X, OUT, rowPlane, alpha, c.nc);
    else hipLaunchKernelGGL((k_left_mult_E_strip<3>), g2, block, 0, c.st, G, c.ws->Kst, X, OUT, rowPlane, alpha, c.nc);
    return;
  }
#define HOMMX_LM(BSV, NE) hipLaunchKernelGGL((k_left_mult_E<BSV, NE>), grid, block, 0, c.st, G, c.ws->Kst, X, OUT, rowPlane, alpha)
  if (ne == 3) HOMMX_LM(1, 3);
  else if (ne == 6) HOMMX_LM(2, 6);
  else if (ne == 9) HOMMX_LM(1, 9);
  else HOMMX_LM(3, 27);
#undef HOMMX_LM
}

}  // namespace

// in-place inverse of the SPD diagonal block [off, off+size) of every matrix of the batch (ld / batch stride from the context,
// default Bp / Bp^2), recursive Schur-complement form; `tmp` points at free scratch (consumed stack-like by the nesting levels)
void invert(const Ctx& c, double* S, int off, int size, double* tmp) {
  const Geo& G = c.ws->G;
  const int ld = c.ld ? c.ld : G.Bp;
  const long long sS = c.sS ? c.sS : (long long)G.Bp * G.Bp;
  const long long sT = c.sT ? c.sT : sS;
  if (size <= 32) {
    if (size == 32)
      hipLaunchKernelGGL(k_leaf_inverse<32>, dim3((unsigned)c.nc), dim3(64), 0, c.st, S, ld, sS, off, c.info, c.stepcode, c.infoDiv);
    else
      hipLaunchKernelGGL(k_leaf_inverse<16>, dim3((unsigned)c.nc), dim3(64), 0, c.st, S, ld, sS, off, c.info, c.stepcode, c.infoDiv);
    return;
  }
  if (size == 64 && !c.ws->leaf32) {
    hipLaunchKernelGGL(k_leaf_inverse_blk<64>, dim3((unsigned)c.nc), dim3(64), 0, c.st, S, ld, sS, off, c.info, c.stepcode, c.infoDiv);
    return;
  }
  int s1 = (size / 2) / 32 * 32;
  if (s1 < 32) s1 = 32;
  if (c.ws->split64 && size >= 128 && size % 64 == 0) s1 = (size / 2) / 64 * 64;  // 192 -> 64 + 128: every leaf a 64-block, whole 64-tiles
  const int s2 = size - s1;
  double* A11 = S + (long long)off * ld + off;
  double* A21 = S + (long long)(off + s1) * ld + off;
  double* A12 = S + (long long)off * ld + off + s1;
  double* A22 = S + (long long)(off + s1) * ld + off + s1;
  double* Xm = tmp;  // s2 x s1, ld = s1, batch stride sT
  invert(c, S, off, s1, tmp);                                                  // A11 <- A11^-1
  gemm(c, false, false, s2, s1, s1, 1.0, A21, ld, sS, A11, ld, sS, 0.0, Xm, s1, sT);   // Xm = A21 A11^-1
  gemm(c, false, true, s2, s2, s1, -1.0, Xm, s1, sT, A21, ld, sS, 1.0, A22, ld, sS, 1);  // A22 <- A22 - Xm A21^T (symmetric: lower tiles;
                                                                                          //  the recursion below never reads above the diagonal tiles)
  invert(c, S, off + s1, s2, tmp + (long long)s1 * s2);                        // A22 <- (Schur)^-1
  gemm(c, false, false, s2, s1, s2, -1.0, A22, ld, sS, Xm, s1, sT, 0.0, A21, ld, sS, 0, A12);  // A21 <- -T^-1 Xm, A12 <- A21^T
  gemm(c, true, false, s1, s1, s2, -1.0, Xm, s1, sT, A21, ld, sS, 1.0, A11, ld, sS, 1, A11);  // A11 <- A11^-1 - Xm^T A21: symmetric
                                                                         // (= A11^-1 + Xm^T T^-1 Xm): lower tiles, mirrored in place
}

void launch_center_corr(BlockedWorkspace* ws, double* corr, long long nc, hipStream_t st) {
  hipLaunchKernelGGL(k_center_corr, dim3((unsigned)(nc * ws->G.t)), dim3(256), 0, st, ws->G, corr);
}

void launch_assembly(BlockedWorkspace* ws, const double* coef, const double* Mm, long long nc, hipStream_t st, double* Kst, double* Brhs,
                     double* C0) {
  const Geo& G = ws->G;
  // ---- K1: the stencil row of a node (3D elasticity: of one row component of a node) in registers, written once, no memset
  {
#define HOMMX_ASMR(D_, K_, SPLIT_)                                                                                          \
  hipLaunchKernelGGL((k_assemble_reg<D_, K_, SPLIT_>), dim3(nblk(nc * G.nn * (SPLIT_ ? D_ : 1), 128)), dim3(128), 0, st, G, coef, Mm, \
                     Kst, Brhs, nc)
    if (G.dim == 2) {
      if (G.kind == 0) HOMMX_ASMR(2, 0, 0); else if (G.kind == 1) HOMMX_ASMR(2, 1, 0); else if (G.kind == 2) HOMMX_ASMR(2, 2, 0); else HOMMX_ASMR(2, 3, 0);
    } else {
      if (G.kind == 0) HOMMX_ASMR(3, 0, 0); else if (G.kind == 1) HOMMX_ASMR(3, 1, 0); else if (G.kind == 2) HOMMX_ASMR(3, 2, 1); else HOMMX_ASMR(3, 3, 1);
    }
#undef HOMMX_ASMR
  }
  {
#define HOMMX_C0(D_, K_)                                                                                                  \
  do {                                                                                                                    \
    if (G.n_el <= 4096) hipLaunchKernelGGL((k_c0<D_, K_, 1>), dim3(nblk(nc, 4)), dim3(256), 0, st, G, coef, C0, nc);   \
    else hipLaunchKernelGGL((k_c0<D_, K_, 4>), dim3((unsigned)nc), dim3(256), 0, st, G, coef, C0, nc);                 \
  } while (0)
    if (G.dim == 2) {
      if (G.kind == 0) HOMMX_C0(2, 0); else if (G.kind == 1) HOMMX_C0(2, 1); else if (G.kind == 2) HOMMX_C0(2, 2); else HOMMX_C0(2, 3);
    } else {
      if (G.kind == 0) HOMMX_C0(3, 0); else if (G.kind == 1) HOMMX_C0(3, 1); else if (G.kind == 2) HOMMX_C0(3, 2); else HOMMX_C0(3, 3);
    }
#undef HOMMX_C0
  }
}

int blocked_reserve(BlockedWorkspace* ws, long long n_cells) {
  if (!ws || n_cells <= 0) return 0;
  if (ws->mf) return mf_reserve(ws, ws->mf, n_cells, true);
  return ws_reserve(ws, n_cells, false);
}

int blocked_solve(BlockedWorkspace* ws, long long ncells, const double* d_coef, const double* d_M, double* d_out,
                  int32_t* d_info, hipStream_t st, double* d_corr) {
  if (ws->mf && !d_corr) return mf_solve(ws, ws->mf, ncells, d_coef, d_M, d_out, d_info, st);  // nested dissection (multifrontal.hip)
  if (ws->mf && ws->mf_corr) {  // correctors on the same route: a second plan whose fronts keep their factors for the back substitution
    if (!ws->mf_keep)
      if (int rc = mf_plan_create(&ws->mf_keep, ws->G, true)) return rc;
    return mf_solve(ws, ws->mf_keep, ncells, d_coef, d_M, d_out, d_info, st, d_corr);
  }
  if (int rc = ws_reserve(ws, ncells, d_corr != nullptr)) return rc;
  const Geo& G = ws->G;
  const int n = G.n, Bp = G.Bp;
  const long long mat = (long long)Bp * Bp;
  if (d_info) BTRY(hipMemsetAsync(d_info, 0, sizeof(int32_t) * ncells, st));
  long long step_cells = d_corr ? std::min(ws->chunk, ws->hchunk) : ws->chunk;
  if (step_cells > 0) {  // equal chunks: a short tail chunk would run the small kernels of the inverse underfilled
    const long long nchunks = (ncells + step_cells - 1) / step_cells;
    step_cells = (ncells + nchunks - 1) / nchunks;
  }
  for (long long c0 = 0; c0 < ncells; c0 += step_cells) {
    const long long nc = std::min(step_cells, ncells - c0);
    Ctx c{ws, nc, st, d_info ? d_info + c0 : nullptr, 0};
    const double* coef = d_coef + c0 * G.n_el * G.ncomp;
    const double* Mm = d_M ? d_M + c0 * G.dim * G.dim : nullptr;
    launch_assembly(ws, coef, Mm, nc, st, ws->Kst, ws->Brhs, ws->C0);
    if (G.b <= 64 && !d_corr && ws->small_fused) {
      // small plane blocks: the whole elimination in ONE launch -- b <= 48: one wave per macro cell, matrices in registers
      // (small_wave.h); 48 < b <= 64, or HOMMX_SMALL_WAVES = 2 | 4: that many waves per cell, matrices in LDS (small_fused.h)
      double* o = d_out + c0 * G.t * G.t;
      int32_t* inf = d_info ? d_info + c0 : nullptr;
      BTRY(launch_small_fused(G, ws->Kst, ws->Brhs, ws->C0, o, inf, nc, ws->small_waves, st));
      BTRY(hipGetLastError());
      continue;
    }
    // ---- K2 init
    BTRY(hipMemsetAsync(ws->S, 0, 8ll * nc * mat, st));
    BTRY(hipMemsetAsync(ws->W, 0, 8ll * nc * mat, st));
    BTRY(hipMemsetAsync(ws->Sl, 0, 8ll * nc * mat, st));
    BTRY(hipMemsetAsync(ws->Gm, 0, 8ll * nc * 256, st));
    const long long scat = nc * (long long)Bp * (G.ncode / 3) * G.bs;
    hipLaunchKernelGGL(k_scatter_plane, dim3(nblk(scat)), dim3(256), 0, st, G, ws->Kst, ws->S, nc, 0, 0, 1);
    hipLaunchKernelGGL(k_scatter_plane, dim3(nblk(scat)), dim3(256), 0, st, G, ws->Kst, ws->W, nc, n - 1, +1, 0);
    hipLaunchKernelGGL(k_scatter_plane, dim3(nblk(scat)), dim3(256), 0, st, G, ws->Kst, ws->Sl, nc, n - 1, 0, 1);
    hipLaunchKernelGGL(k_add_P, dim3(nblk(nc * 16ll * Bp)), dim3(256), 0, st, G, ws->Brhs, ws->R, nc, 0, 1);
    hipLaunchKernelGGL(k_add_P, dim3(nblk(nc * 16ll * Bp)), dim3(256), 0, st, G, ws->Brhs, ws->Rl, nc, n - 1, 1);
    // ---- K2 elimination of planes 0 .. n-2
    for (int j = 0; j <= n - 2; ++j) {
      const bool last = (j == n - 2);
      c.stepcode = j + 1;
      if (last)  // the last plane couples to plane n-2 through E as well as through the arrow
        hipLaunchKernelGGL(k_scatter_plane, dim3(nblk(scat)), dim3(256), 0, st, G, ws->Kst, ws->W, nc, n - 1, -1, 0);
      invert(c, ws->S, 0, Bp, ws->T);                                                                   // S <- S^-1
      if (d_corr) {  // keep what the back substitution needs: S_j^-1, W_j (incl. E on the last step), R_j
        BTRY(hipMemcpyAsync(ws->hS + (long long)j * nc * mat, ws->S, 8ll * nc * mat, hipMemcpyDeviceToDevice, st));
        BTRY(hipMemcpyAsync(ws->hW + (long long)j * nc * mat, ws->W, 8ll * nc * mat, hipMemcpyDeviceToDevice, st));
        BTRY(hipMemcpyAsync(ws->hR + (long long)j * nc * 16 * Bp, ws->R, 8ll * nc * 16 * Bp, hipMemcpyDeviceToDevice, st));
      }
      gemm(c, false, false, Bp, Bp, Bp, 1.0, ws->W, Bp, mat, ws->S, Bp, mat, 0.0, ws->V, Bp, mat);      // V = W Sinv
      gemm(c, false, true, Bp, Bp, Bp, -1.0, ws->V, Bp, mat, ws->W, Bp, mat, 1.0, ws->Sl, Bp, mat, 1);  // S_last -= V W^T (lower tiles)
      gemm(c, false, false, 16, Bp, Bp, 1.0, ws->R, Bp, 16ll * Bp, ws->S, Bp, mat, 0.0, ws->Vr, Bp, 16ll * Bp);   // Vr = R Sinv
      gemm(c, false, true, 16, 16, Bp, 1.0, ws->Vr, Bp, 16ll * Bp, ws->R, Bp, 16ll * Bp, 1.0, ws->Gm, 16, 256);   // G += Vr R^T
      gemm(c, false, true, 16, Bp, Bp, -1.0, ws->Vr, Bp, 16ll * Bp, ws->W, Bp, mat, 1.0, ws->Rl, Bp, 16ll * Bp);  // R_last -= Vr W^T
      if (!last) {
        right_mult_Et(c, ws->S, ws->X, Bp, j + 1, 1.0);   // X = Sinv E^T
        left_mult_E(c, ws->X, ws->S, j + 1, -1.0);        // S = -E X
        hipLaunchKernelGGL(k_scatter_plane, dim3(nblk(scat)), dim3(256), 0, st, G, ws->Kst, ws->S, nc, j + 1, 0, 1);                 // S += D_{j+1}
        right_mult_Et(c, ws->V, ws->W, Bp, j + 1, -1.0);  // W = -V E^T
        right_mult_Et(c, ws->Vr, ws->R, 16, j + 1, -1.0);  // R = -Vr E^T
        hipLaunchKernelGGL(k_add_P, dim3(nblk(nc * 16ll * Bp)), dim3(256), 0, st, G, ws->Brhs, ws->R, nc, j + 1, 0);                 // R += P_{j+1}
      }
    }
    // ---- last plane
    c.stepcode = n;
    hipLaunchKernelGGL(k_symmetrize, dim3(nblk(nc * mat)), dim3(256), 0, st, Bp, ws->Sl, mat, nc);  // mirror the lower tiles
    hipLaunchKernelGGL(k_pin_last, dim3(nblk(nc * (long long)Bp * G.bs)), dim3(256), 0, st, G, ws->Sl, ws->Rl, nc);
    invert(c, ws->Sl, 0, Bp, ws->T);
    gemm(c, false, false, 16, Bp, Bp, 1.0, ws->Rl, Bp, 16ll * Bp, ws->Sl, Bp, mat, 0.0, ws->Vr, Bp, 16ll * Bp);
    gemm(c, false, true, 16, 16, Bp, 1.0, ws->Vr, Bp, 16ll * Bp, ws->Rl, Bp, 16ll * Bp, 1.0, ws->Gm, 16, 256);
    // ---- correctors: back substitution  chi_j = S_j^-1 (r_j - E_j^T chi_{j+1} - W_j^T chi_last), rows = load cases
    if (d_corr) {
      double* corr = d_corr + c0 * (long long)G.t * G.nn * G.bs;
      const long long sx = 16ll * Bp;
      hipLaunchKernelGGL(k_store_corr, dim3(nblk(nc * (long long)G.t * G.b)), dim3(256), 0, st, G, ws->Vr, corr, nc, n - 1);
      double* Xn = ws->Xa;  // chi_{j+1}
      double* Xc = ws->Xb;  // chi_j
      for (int j = n - 2; j >= 0; --j) {
        BTRY(hipMemcpyAsync(ws->Y, ws->hR + (long long)j * nc * sx, 8ll * nc * sx, hipMemcpyDeviceToDevice, st));
        gemm(c, false, false, 16, Bp, Bp, -1.0, ws->Vr, Bp, sx, ws->hW + (long long)j * nc * mat, Bp, mat, 1.0, ws->Y, Bp, sx);
        if (j < n - 2) right_mult_Et(c, Xn, ws->Y, 16, j, -1.0, +1, 1);  // Y -= chi_{j+1} E_j  (E_j[r][c] = K[(c, j), (r, j+1)])
        gemm(c, false, false, 16, Bp, Bp, 1.0, ws->Y, Bp, sx, ws->hS + (long long)j * nc * mat, Bp, mat, 0.0, Xc, Bp, sx);
        hipLaunchKernelGGL(k_store_corr, dim3(nblk(nc * (long long)G.t * G.b)), dim3(256), 0, st, G, Xc, corr, nc, j);
        std::swap(Xn, Xc);
      }
      hipLaunchKernelGGL(k_center_corr, dim3((unsigned)(nc * G.t)), dim3(256), 0, st, G, corr);
    }
    // ---- K3
    hipLaunchKernelGGL(k_finalize, dim3(nblk(nc * G.t * G.t)), dim3(256), 0, st, G, ws->C0, ws->Gm, d_out + c0 * G.t * G.t, nc);
    BTRY(hipGetLastError());
  }
  return 0;
}

}  // namespace hommx
