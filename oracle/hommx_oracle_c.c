/*
 * hommx_oracle_c.c -- plain-C CPU restatements of the micro-cell path  --  TEST INFRASTRUCTURE ONLY.
 *
 * Part 1 (below): the 2D scalar (stratified) Poisson path, fast (closed-form stencil, block-cyclic elimination, OpenMP over cells).
 * Part 2 (end of file, hommx_oracle_generic): every kind in 2D and 3D, element by element from the reference's forms, dense Cholesky --
 * small meshes only; the cross-check of the NumPy oracle's 3D and elasticity paths.
 *
 * A second, independent-in-code restatement next to oracle/hommx_oracle.py (which assembles from element gradients and solves
 * with a sparse LU in the energy form of hmm.py:652-667): this one forms the periodic 7-point stencil of the right-diagonal P1
 * mesh in closed form (hmm.py:644-650, 759-772 on element means; periodic identification cell_problem.py:38-136 == indices
 * mod n), eliminates the n x n torus as a block-cyclic tridiagonal system with dense n x n blocks (Gauss-Jordan inverses, no
 * pivoting: the blocks are SPD) and evaluates the Schur form  A_H = C0 - B^T K^+ B.  The two agree to 1e-12
 * (tests/test_oracle_c.py); bench.py times this file with OpenMP over the macro cells as the OPTIMISED CPU baseline (SURVEY
 * 8(d) baseline (ii)) beside the reference-shaped one-core port.  Only tests/, __graft_entry__.smoke()/build() and bench.py's
 * cpu_baseline leg may load it; nothing under hommx_amd/ does.
 *
 *   gcc -O3 -march=native -fopenmp -shared -fPIC -o oracle/_build/libhommx_oracle.so oracle/hommx_oracle_c.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* in-place inverse of the SPD n x n matrix a (row-major), Gauss-Jordan without pivoting; returns 1 on a non-positive pivot */
static int gj_inverse(double* a, int n) {
  for (int k = 0; k < n; ++k) {
    const double d = a[k * n + k];
    if (!(d > 0.0) || !isfinite(d)) return 1;
    const double p = 1.0 / d;
    for (int j = 0; j < n; ++j) a[k * n + j] *= p;
    a[k * n + k] = p;
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const double f = a[i * n + k];
      if (f == 0.0) continue;
      a[i * n + k] = 0.0;
      double* ai = a + (size_t)i * n;
      const double* ak = a + (size_t)k * n;
      for (int j = 0; j < n; ++j) ai[j] -= f * ak[j];
    }
  }
  return 0;
}

/* C (n x n) = alpha * A * B (+ C if acc); all row-major n x n */
static void gemm_nn(int n, double alpha, const double* A, const double* B, double* C, int acc) {
  if (!acc) memset(C, 0, sizeof(double) * (size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < n; ++k) {
      const double f = alpha * A[i * n + k];
      if (f == 0.0) continue;
      const double* bk = B + (size_t)k * n;
      double* ci = C + (size_t)i * n;
      for (int j = 0; j < n; ++j) ci[j] += f * bk[j];
    }
}
/* C += alpha * A * B^T */
static void gemm_nt_acc(int n, double alpha, const double* A, const double* B, double* C) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const double* ai = A + (size_t)i * n;
      const double* bj = B + (size_t)j * n;
      double s = 0.0;
      for (int k = 0; k < n; ++k) s += ai[k] * bj[k];
      C[i * n + j] += alpha * s;
    }
}

typedef struct {
  double *dg, *ce, *cN, *cNE, *p0, *p1; /* [n rows][n columns] */
} Stencil;

/* stencil of node row j from cell rows j (above the nodes) and j-1 (below); coef[2 (i + n j) + s], s = triangle (v0,v1,v3) / (v0,v2,v3) */
static void stencil_rows(const double* coef, const double* M, int n, Stencil* S, double* asum) {
  double m00 = 1, m01 = 0, m10 = 0, m11 = 1;
  if (M) { m00 = M[0]; m01 = M[1]; m10 = M[2]; m11 = M[3]; }
  const double al = 0.5 * (m00 * m00 + m10 * m10), be = 0.5 * (m01 * m01 + m11 * m11), ga = 0.5 * (m00 * m01 + m10 * m11);
  const double ab = al - 2.0 * ga + be;
  double tot = 0.0;
  for (int j = 0; j < n; ++j) {
    const double* cur = coef + 2 * (size_t)n * j;
    const double* prv = coef + 2 * (size_t)n * ((j + n - 1) % n);
    for (int c = 0; c < n; ++c) {
      const int cm = (c + n - 1) % n;
      const double a0 = cur[2 * c], a1 = cur[2 * c + 1], a0m = cur[2 * cm], a1m = cur[2 * cm + 1];
      const double b0 = prv[2 * c], b1 = prv[2 * c + 1], b0m = prv[2 * cm], b1m = prv[2 * cm + 1];
      (void)b0;
      (void)a1m;
      S->dg[j * n + c] = a0 * al + a1 * be + a0m * ab + b0m * be + b1m * al + b1 * ab;
      S->ce[j * n + c] = (a0 + b1) * (ga - al);            /* (c, j) <-> (c+1, j) */
      S->cN[j * n + c] = (a1 + a0m) * (ga - be);           /* (c, j+1) <- (c, j) */
      S->cNE[j * n + c] = -ga * (a0 + a1);                 /* (c+1, j+1) <- (c, j) */
      S->p0[j * n + c] = a0 - a0m - b1m + b1;
      S->p1[j * n + c] = a1 + a0m - b0m - b1;
      tot += a0 + a1;
    }
  }
  *asum = tot;
}

static void band_D(const double* dg, const double* ce, int n, double* D) { /* cyclic tridiagonal */
  memset(D, 0, sizeof(double) * (size_t)n * n);
  for (int c = 0; c < n; ++c) {
    const int cp = (c + 1) % n;
    D[c * n + c] += dg[c];
    D[c * n + cp] += ce[c];
    D[cp * n + c] += ce[c];
  }
}
static void band_E(const double* cN, const double* cNE, int n, double* E) { /* E[r][r] = cN[r], E[r][r-1] = cNE[r-1] (cyclic) */
  memset(E, 0, sizeof(double) * (size_t)n * n);
  for (int c = 0; c < n; ++c) {
    E[c * n + c] += cN[c];
    E[c * n + (c + n - 1) % n] += cNE[(c + n - 1) % n];
  }
}

/* one macro cell; work = 8 n^2 + 6 n^2 (stencil) doubles; returns 0 ok / step of the first bad pivot */
static int solve_cell(const double* coef, const double* M, int n, double* out, double* work) {
  const size_t nn = (size_t)n * n;
  double *S = work, *Sl = S + nn, *W = Sl + nn, *V = W + nn, *E = V + nn, *X = E + nn, *T = X + nn, *Wn = T + nn;
  Stencil st;
  st.dg = Wn + nn; st.ce = st.dg + nn; st.cN = st.ce + nn; st.cNE = st.cN + nn; st.p0 = st.cNE + nn; st.p1 = st.p0 + nn;
  double asum;
  stencil_rows(coef, M, n, &st, &asum);
  double R[2][64], Rl[2][64], Vr[2][64], G[2][2] = {{0, 0}, {0, 0}};
  band_D(st.dg, st.ce, n, S);
  band_D(st.dg + (size_t)(n - 1) * n, st.ce + (size_t)(n - 1) * n, n, Sl);
  band_E(st.cN + (size_t)(n - 1) * n, st.cNE + (size_t)(n - 1) * n, n, E); /* W_0 = E_{n-1}^T */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) W[i * n + j] = E[j * n + i];
  for (int c = 0; c < n; ++c) {
    R[0][c] = st.p0[c]; R[1][c] = st.p1[c];
    Rl[0][c] = st.p0[(size_t)(n - 1) * n + c]; Rl[1][c] = st.p1[(size_t)(n - 1) * n + c];
  }
  for (int j = 0; j <= n - 2; ++j) {
    band_E(st.cN + (size_t)j * n, st.cNE + (size_t)j * n, n, E);
    if (j == n - 2)
      for (size_t q = 0; q < nn; ++q) W[q] += E[q];
    if (gj_inverse(S, n)) return j + 1;                 /* S <- S^-1 */
    gemm_nn(n, 1.0, W, S, V, 0);                        /* V = W S^-1 */
    gemm_nt_acc(n, -1.0, V, W, Sl);                     /* S_last -= V W^T */
    for (int m = 0; m < 2; ++m)
      for (int c = 0; c < n; ++c) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += R[m][k] * S[k * n + c];
        Vr[m][c] = s;
      }
    for (int m = 0; m < 2; ++m)
      for (int q = 0; q < 2; ++q) {
        double s = 0.0;
        for (int c = 0; c < n; ++c) s += Vr[m][c] * R[q][c];
        G[m][q] += s;
      }
    for (int m = 0; m < 2; ++m)
      for (int r = 0; r < n; ++r) {
        double s = 0.0;
        for (int c = 0; c < n; ++c) s += Vr[m][c] * W[r * n + c];
        Rl[m][r] -= s;
      }
    if (j < n - 2) {
      /* X = S^-1 E^T ; S_next = D_{j+1} - E X ; W_next = -V E^T ; R_next = P_{j+1} - Vr E^T */
      for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) {
          const int cm = (c + n - 1) % n;
          X[i * n + c] = S[i * n + c] * E[c * n + c] + S[i * n + cm] * E[c * n + cm];
          Wn[i * n + c] = -(V[i * n + c] * E[c * n + c] + V[i * n + cm] * E[c * n + cm]);
        }
      band_D(st.dg + (size_t)(j + 1) * n, st.ce + (size_t)(j + 1) * n, n, T);
      for (int r = 0; r < n; ++r) {
        const int rm = (r + n - 1) % n;
        for (int c = 0; c < n; ++c) T[r * n + c] -= E[r * n + r] * X[r * n + c] + E[r * n + rm] * X[rm * n + c];
      }
      memcpy(S, T, sizeof(double) * nn);
      memcpy(W, Wn, sizeof(double) * nn);
      for (int m = 0; m < 2; ++m) {
        double nr[64];
        for (int c = 0; c < n; ++c) {
          const int cm = (c + n - 1) % n;
          const double p = (m ? st.p1 : st.p0)[(size_t)(j + 1) * n + c];
          nr[c] = p - (Vr[m][c] * E[c * n + c] + Vr[m][cm] * E[c * n + cm]);
        }
        memcpy(R[m], nr, sizeof(double) * n);
      }
    }
  }
  /* last node row: gauge (drop the last unknown, cell_problem.py:349-361), inverse, loads */
  for (int x = 0; x < n; ++x) { Sl[(n - 1) * n + x] = 0.0; Sl[x * n + n - 1] = 0.0; }
  Sl[(n - 1) * n + n - 1] = 1.0;
  Rl[0][n - 1] = 0.0; Rl[1][n - 1] = 0.0;
  if (gj_inverse(Sl, n)) return n;
  for (int m = 0; m < 2; ++m)
    for (int c = 0; c < n; ++c) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s += Rl[m][k] * Sl[k * n + c];
      Vr[m][c] = s;
    }
  for (int m = 0; m < 2; ++m)
    for (int q = 0; q < 2; ++q) {
      double s = 0.0;
      for (int c = 0; c < n; ++c) s += Vr[m][c] * Rl[q][c];
      G[m][q] += s;
    }
  double m00 = 1, m01 = 0, m10 = 0, m11 = 1;
  if (M) { m00 = M[0]; m01 = M[1]; m10 = M[2]; m11 = M[3]; }
  const double h = 1.0 / n, sc = 0.25 * h * h, c0 = 0.5 * h * h * asum;
  /* A_H = C0 I - (h^2/4) M G M^T   (G accumulates + Vr R^T with positive S^-1, i.e. B^T K^+ B) */
  const double t00 = m00 * G[0][0] + m01 * G[1][0], t01 = m00 * G[0][1] + m01 * G[1][1];
  const double t10 = m10 * G[0][0] + m11 * G[1][0], t11 = m10 * G[0][1] + m11 * G[1][1];
  out[0] = c0 - sc * (t00 * m00 + t01 * m01);
  out[1] = -sc * (t00 * m10 + t01 * m11);
  out[2] = -sc * (t10 * m00 + t11 * m01);
  out[3] = c0 - sc * (t10 * m10 + t11 * m11);
  return 0;
}

/* coef[ncells][2 n^2], M[ncells][2][2] or NULL -> A_eff[ncells][2][2], info[ncells] (or NULL); OpenMP over the cells.
 * Returns the number of threads used, or -1 on bad arguments (3 <= n <= 64). */
int hommx_oracle_poisson2d(int n, int64_t ncells, const double* coef, const double* M, double* A_eff, int32_t* info, int nthreads) {
  if (n < 3 || n > 64 || ncells < 0 || !coef || !A_eff) return -1;
  int used = 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
#else
  {
#endif
    double* work = (double*)malloc(sizeof(double) * 14 * (size_t)n * n);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int64_t c = 0; c < ncells; ++c) {
      const int rc = solve_cell(coef + c * 2 * (int64_t)n * n, M ? M + 4 * c : NULL, n, A_eff + 4 * c, work);
      if (info) info[c] = rc;
      if (rc) A_eff[4 * c] = A_eff[4 * c + 1] = A_eff[4 * c + 2] = A_eff[4 * c + 3] = NAN;
    }
    free(work);
  }
  return used;
}

/* =====================================================================================================================================
 * Generic element-by-element restatement: 2D and 3D, scalar / matrix-valued Poisson and isotropic / general linear elasticity, with or
 * without the stratification matrix M.  Written from the forms of the reference directly, with plain loops over tensor indices (no
 * Voigt algebra, no stencil, no sparse solver) -- independent in code of oracle/hommx_oracle.py and of the GPU kernels:
 *
 *   micro mesh        create_unit_square / create_unit_cube, squares split along v00 - v11, cubes into the six tetrahedra around
 *                     v0 - v7; periodic identification = indices mod n (cell_problem.py:38-300)
 *   Poisson           a(u, z) = int A (M grad u) . (M grad z)                       hmm.py:644-647 / 759-766
 *                     l_m(z)  = - int A e_m . (M grad z)                            hmm.py:649-650 / 768-772 with grad v = e_m
 *   elasticity        a(u, z) = int C : e_D(u) : e_D(z),  e_D(u) = sym(M . nabla_grad u),  (M . nabla_grad u)_ij = sum_k M_ik d_k u_j
 *                                                                                   hmm.py:887-896 / 1024-1041
 *                     l_m(z)  = - int C : E^m : e_D(z),  E^m the unit symmetric strains (11, 22, [33,] 12 [, 13, 23])   hmm.py:898-903 / 1043-1048
 *   effective tensor  A_H[m][n] = int C : E^m : E^n  -  B^T K^+ B                   (= hmm.py:652-667 / 905-922 over eps^2)
 *
 * Dense assembly, the last node pinned (constants are the kernel: cell_problem.py:349-361), dense Cholesky.  For SMALL meshes only
 * (bs n^d unknowns, dense): the cross-check of the NumPy oracle's 3D and elasticity paths in tests/test_oracle_c.py.
 *
 *   kind 0: coef[n_el]            scalar A                      kind 1: coef[n_el][d][d]         matrix-valued A (symmetric)
 *   kind 2: coef[n_el][2]         (lambda, mu)                  kind 3: coef[n_el][d][d][d][d]   Hooke tensor
 * returns 0, or 1 on bad arguments / a non-positive pivot.
 * ===================================================================================================================================== */
static int chol_dense(double* a, int n) { /* lower Cholesky in place */
  for (int j = 0; j < n; ++j) {
    double d = a[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
    if (!(d > 0.0) || !isfinite(d)) return 1;
    d = sqrt(d);
    a[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = a[(size_t)i * n + j];
      const double* ai = a + (size_t)i * n;
      const double* aj = a + (size_t)j * n;
      for (int k = 0; k < j; ++k) s -= ai[k] * aj[k];
      a[(size_t)i * n + j] = s / d;
    }
  }
  return 0;
}

int hommx_oracle_generic(int dim, int n, int kind, const double* coef, const double* Mmat, double* out) {
  if ((dim != 2 && dim != 3) || n < 3 || kind < 0 || kind > 3 || !coef || !out) return 1;
  static const int tri[2][3][2] = {{{0, 0}, {1, 0}, {1, 1}}, {{0, 0}, {0, 1}, {1, 1}}};
  static const int vb[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {1, 1, 0}, {0, 0, 1}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
  static const int tet[6][4] = {{0, 1, 3, 7}, {0, 1, 7, 5}, {0, 5, 7, 4}, {0, 3, 2, 7}, {0, 6, 4, 7}, {0, 2, 6, 7}};
  const int d = dim, nv = d + 1, nsub = d == 2 ? 2 : 6, el = kind >= 2, bs = el ? d : 1, t = el ? d * (d + 1) / 2 : d;
  long nn = 1;
  for (int k = 0; k < d; ++k) nn *= n;
  const long N = nn * bs, Nr = N - bs; /* the last node's unknowns are dropped */
  const int ncomp = kind == 0 ? 1 : kind == 1 ? d * d : kind == 2 ? 2 : d * d * d * d;
  double M[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (Mmat)
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) M[i][j] = Mmat[i * d + j];
  /* unit symmetric strains E^m (elasticity) / unit gradients e_m (Poisson) as d x d tensors (Poisson: row 0 holds e_m) */
  double E[6][3][3];
  memset(E, 0, sizeof(E));
  if (el) {
    int m = 0;
    for (int k = 0; k < d; ++k, ++m) E[m][k][k] = 1.0;
    for (int k = 0; k < d; ++k)
      for (int l = k + 1; l < d; ++l, ++m) E[m][k][l] = E[m][l][k] = 0.5;
  }
  double* K = (double*)calloc((size_t)N * N, sizeof(double));
  double* B = (double*)calloc((size_t)N * t, sizeof(double));
  double C0[36];
  memset(C0, 0, sizeof(C0));
  if (!K || !B) { free(K); free(B); return 1; }
  const double h = 1.0 / n;
  double vol = 1.0;
  for (int k = 0; k < d; ++k) vol *= h;
  vol /= (d == 2 ? 2.0 : 6.0);
  long cell[3] = {0, 0, 0};
  for (long c = 0; c < nn; ++c) {
    cell[0] = c % n; cell[1] = (c / n) % n; cell[2] = d == 3 ? c / ((long)n * n) : 0;
    for (int s = 0; s < nsub; ++s) {
      const double* ce = coef + ((size_t)c * nsub + s) * ncomp;
      /* vertices (in units of h, relative to the cell corner), their periodic node ids, and the P1 gradients */
      int off[4][3];
      long node[4];
      for (int a = 0; a < nv; ++a) {
        long id = 0, mul = 1;
        for (int k = 0; k < d; ++k) {
          off[a][k] = d == 2 ? tri[s][a][k] : vb[tet[s][a]][k];
          id += ((cell[k] + off[a][k]) % n) * mul;
          mul *= n;
        }
        node[a] = id;
      }
      double J[3][3], Ji[3][3], g[4][3]; /* J[k][a-1] = x_a - x_0 ; grad lambda_a = row a-1 of J^-1 ; grad lambda_0 = -sum */
      for (int k = 0; k < d; ++k)
        for (int a = 1; a < nv; ++a) J[k][a - 1] = h * (off[a][k] - off[0][k]);
      if (d == 2) {
        const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
        Ji[0][0] = J[1][1] / det; Ji[0][1] = -J[0][1] / det; Ji[1][0] = -J[1][0] / det; Ji[1][1] = J[0][0] / det;
      } else {
        const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                           J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) / det; Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) / det;
        Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) / det; Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) / det;
        Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) / det; Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) / det;
        Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) / det; Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) / det;
        Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) / det;
      }
      for (int k = 0; k < d; ++k) g[0][k] = 0.0;
      for (int a = 1; a < nv; ++a)
        for (int k = 0; k < d; ++k) { g[a][k] = Ji[a - 1][k]; g[0][k] -= Ji[a - 1][k]; }
      double gt[4][3]; /* M grad lambda_a */
      for (int a = 0; a < nv; ++a)
        for (int i = 0; i < d; ++i) { gt[a][i] = 0.0; for (int k = 0; k < d; ++k) gt[a][i] += M[i][k] * g[a][k]; }
      if (!el) {
        double A[3][3] = {{0}};
        for (int i = 0; i < d; ++i)
          for (int j = 0; j < d; ++j) A[i][j] = kind == 0 ? (i == j ? ce[0] : 0.0) : ce[i * d + j];
        for (int a = 0; a < nv; ++a) {
          double Ag[3]; /* A gt_a */
          for (int i = 0; i < d; ++i) { Ag[i] = 0.0; for (int j = 0; j < d; ++j) Ag[i] += A[i][j] * gt[a][j]; }
          for (int b = 0; b < nv; ++b) {
            double v = 0.0;
            for (int i = 0; i < d; ++i) v += gt[b][i] * Ag[i];
            K[(size_t)node[b] * N + node[a]] += vol * v;
          }
          for (int m = 0; m < d; ++m) { /* - int A e_m . (M grad z_a) */
            double v = 0.0;
            for (int i = 0; i < d; ++i) v += A[i][m] * gt[a][i];
            B[(size_t)node[a] * t + m] -= vol * v;
          }
        }
        for (int m = 0; m < d; ++m)
          for (int q = 0; q < d; ++q) C0[m * t + q] += vol * A[m][q];
      } else {
        /* stress of a strain tensor e: isotropic  lambda tr(e) I + 2 mu e ;  general  C_ijkl e_kl */
#define HOMMX_STRESS(sig, e)                                                                                       \
  do {                                                                                                             \
    if (kind == 2) {                                                                                               \
      double tr = 0.0;                                                                                             \
      for (int i = 0; i < d; ++i) tr += (e)[i][i];                                                                 \
      for (int i = 0; i < d; ++i)                                                                                  \
        for (int j = 0; j < d; ++j) (sig)[i][j] = 2.0 * ce[1] * (e)[i][j] + (i == j ? ce[0] * tr : 0.0);           \
    } else {                                                                                                       \
      for (int i = 0; i < d; ++i)                                                                                  \
        for (int j = 0; j < d; ++j) {                                                                              \
          double v_ = 0.0;                                                                                         \
          for (int k = 0; k < d; ++k)                                                                              \
            for (int l = 0; l < d; ++l) v_ += ce[((i * d + j) * d + k) * d + l] * (e)[k][l];                       \
          (sig)[i][j] = v_;                                                                                        \
        }                                                                                                          \
    }                                                                                                              \
  } while (0)
        double eps[4][3][3][3]; /* e_D of the basis function (vertex a, component al): sym(gt_a (x) e_al) */
        for (int a = 0; a < nv; ++a)
          for (int al = 0; al < d; ++al)
            for (int i = 0; i < d; ++i)
              for (int j = 0; j < d; ++j) eps[a][al][i][j] = 0.5 * ((j == al ? gt[a][i] : 0.0) + (i == al ? gt[a][j] : 0.0));
        for (int a = 0; a < nv; ++a)
          for (int al = 0; al < d; ++al) {
            double sig[3][3];
            HOMMX_STRESS(sig, eps[a][al]);
            for (int b = 0; b < nv; ++b)
              for (int be = 0; be < d; ++be) {
                double v = 0.0;
                for (int i = 0; i < d; ++i)
                  for (int j = 0; j < d; ++j) v += eps[b][be][i][j] * sig[i][j];
                K[(size_t)(node[b] * bs + be) * N + node[a] * bs + al] += vol * v;
              }
          }
        for (int m = 0; m < t; ++m) {
          double sig[3][3];
          HOMMX_STRESS(sig, E[m]);
          for (int a = 0; a < nv; ++a)
            for (int al = 0; al < d; ++al) {
              double v = 0.0;
              for (int i = 0; i < d; ++i)
                for (int j = 0; j < d; ++j) v += eps[a][al][i][j] * sig[i][j];
              B[(size_t)(node[a] * bs + al) * t + m] -= vol * v;
            }
          for (int q = 0; q < t; ++q) {
            double v = 0.0;
            for (int i = 0; i < d; ++i)
              for (int j = 0; j < d; ++j) v += E[q][i][j] * sig[i][j];
            C0[m * t + q] += vol * v;
          }
        }
#undef HOMMX_STRESS
      }
    }
  }
  /* reduced system (last node dropped), Cholesky, Y = L^-1 B, A_H = C0 - Y^T Y */
  double* Kr = (double*)malloc((size_t)Nr * Nr * sizeof(double));
  if (!Kr) { free(K); free(B); return 1; }
  for (long i = 0; i < Nr; ++i) memcpy(Kr + (size_t)i * Nr, K + (size_t)i * N, sizeof(double) * Nr);
  free(K);
  int rc = chol_dense(Kr, (int)Nr);
  if (!rc) {
    for (int m = 0; m < t; ++m)
      for (long i = 0; i < Nr; ++i) { /* forward substitution, column m of B in place */
        double s = B[(size_t)i * t + m];
        const double* li = Kr + (size_t)i * Nr;
        for (long k = 0; k < i; ++k) s -= li[k] * B[(size_t)k * t + m];
        B[(size_t)i * t + m] = s / li[i];
      }
    for (int m = 0; m < t; ++m)
      for (int q = 0; q < t; ++q) {
        double s = 0.0;
        for (long i = 0; i < Nr; ++i) s += B[(size_t)i * t + m] * B[(size_t)i * t + q];
        out[m * t + q] = C0[m * t + q] - s;
      }
  }
  free(Kr);
  free(B);
  return rc;
}
