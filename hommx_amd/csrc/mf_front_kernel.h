// mf_front_kernel.h -- the register-resident front kernel of the nested-dissection route (see mf_front.h) and its launcher for ONE number of
// unknowns per node.  Its fully unrolled instantiations take minutes to compile, so every BS gets a translation unit of its own
// (mf_front_bs1.hip / _bs2 / _bs3 include this file and instantiate launch_mf_front_bs<BS>); mf_front.hip dispatches.
#pragma once
#include "mf_front.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "sweep_acc.h"

namespace hommx {

#ifdef MFF_PROF  // dev builds (ONE translation unit compiled with -DMFF_PROF, e.g. mf_front_bs3.hip): clocks per phase of wave 0 of every
                 // workgroup, read by hommx_mff_prof_read (tools/mff_prof.py)
__device__ unsigned long long mff_prof[16];
#define MFF_T(i) do { if (tid == 0) { const unsigned long long now__ = clock64(); atomicAdd(&mff_prof[i], now__ - t_prev__); t_prev__ = now__; } } while (0)
#else
#define MFF_T(i)
#endif

// compile-time loop: the tiles of a wave are separate registers, never an indexed array (a loop the compiler declines to unroll would send
// the whole front to scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void mff_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    mff_for<I + 1, N>(f);
  }
}

// upper tiles are numbered column by column, e = b (b + 1) / 2 + a (a <= b): the number of a tile does not depend on the size of the front,
// so a one-wave workgroup (tile e = register set e) knows (a, b) of every register set at compile time
constexpr int mff_col(int e) {
  int b = 0;
  while ((b + 1) * (b + 2) / 2 <= e) ++b;
  return b;
}
constexpr int mff_row(int e) { return e - mff_col(e) * (mff_col(e) + 1) / 2; }

template <int NW>
__device__ __forceinline__ void mff_sync() {
  if constexpr (NW == 1) SyncWave::sync();
  else __syncthreads();
}

// BS unknowns per node; NW waves; TMAX: most tiles per dimension; TPW: most (register) tiles per wave; MINB: waves per SIMD the register
// allocation must leave room for (small fronts are bound by the latency of their dependent loads and pivot chains: what hides it is the
// number of fronts in flight); FENCE: tiles whose build loads are in flight together; LROWS: most tile rows that live in LDS instead of
// registers (fronts one or two tiles per dimension over the register budget -- the 3D-elasticity leaf (81, 222 + 8), T = 21: 231 tiles, 190
// of them in registers, the two first tile rows (41 tiles) in LDS: rows 0 and 1 are the first two panels, each is consumed as a panel where
// it stands and row 1 takes the update of panel 0 in place)
template <int BS, int NW, int TMAX, int TPW, int MINB, int FENCE, int LROWS = 0>
__global__ __launch_bounds__(64 * NW, MINB) void k_mf_front(MfFrontDev g, const double* __restrict__ Kst, const double* __restrict__ Brhs,
                                                      double* __restrict__ arena, long long nc, long long batch0, int nn, int ncode, int t,
                                                      int32_t* __restrict__ info, int stepcode) {
  typedef accl::v4d v4d;
  constexpr int NU = TMAX * 16;
  constexpr int YSZ = NW == 1 ? 8 : (TMAX * 256 > NW * 272) ? TMAX * 256 : NW * 272;  // one wave: Y' stays in registers
  // panel tiles (p, b): E_b^T, one dense 16 x 16 k-major tile per column block; with LDS rows: tile (0, b) at slot b, tile (1, b) at slot
  // T - 1 + b, and the panels of the register rows reuse the slots of row 0
  __shared__ double Qp[(LROWS ? 2 * TMAX - 1 : TMAX) * 256];
  __shared__ double Yp[YSZ];          // Y'_a = -N E_a^T per row block; at the end: per-wave 16 x 17 transpose scratch
  __shared__ double ubuf[NW * 64];    // pivot-row buffers of the sweeps, per wave
  // per unknown u of the front, so that an entry of the build costs a few LDS reads and adds instead of divisions and 64-bit products:
  __shared__ int s_upos[2 * NU];      // unknown of child slot c's update matrix, -1: none
  __shared__ int s_gnode[NU];         // global node of a real unknown; -1: padding; -2 - m: border row m
  __shared__ int s_rk[NU];            // as the ROW of a stencil entry:    component * BS * nn + global node
  __shared__ int s_ck[NU];            // as the COLUMN of a stencil entry: component * nn
  __shared__ int s_n1[NU];            // local node * ns (row of the stencil-code table)
  __shared__ int s_nl[NU];            // local node      (column of the stencil-code table; eliminated unknowns only)
  constexpr int MFF_CODE_LDS = NW == 1 ? 1024 : 4096;  // stencil-code tables up to this many bytes are staged in LDS (as 32-bit words)
  __shared__ int32_t s_code32[MFF_CODE_LDS / 4];
  const int8_t* s_code = reinterpret_cast<const int8_t*>(s_code32);
  const int tid = threadIdx.x, l = tid & 63, j = l & 15, k = l >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = g.T, P = g.P, ntiles = g.ntiles, s16 = g.s16, s0 = g.ns * BS;
  const int R0 = LROWS ? g.R0 : 0;  // tile rows [0, R0) live in LDS; g.tilemap / ntiles cover the register tiles (rows >= R0) only
  auto mm = [](double x, double y, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); };

  // ONE front per workgroup (a grid-stride loop over fronts invites the compiler to hoist the per-tile index arithmetic of all tiles out of
  // it -- more registers than the front itself; big batches are launched in pieces by the host)
  {
    const long long batch = batch0 + blockIdx.x;
    const long long cell = batch / g.nf;
    const int f = (int)(batch % g.nf);
    const int32_t* nodes = g.nodes + (long long)f * g.nloc;
    const int8_t* gcode = g.code + (long long)f * g.nloc * g.ns;
    const int ncodes = g.nloc * g.ns;
    const bool code_lds = ncodes <= MFF_CODE_LDS;
    if (code_lds) {  // tables are 4-byte aligned per front when nloc * ns is a multiple of 4; otherwise byte by byte
      if ((((long long)f * ncodes) & 3) == 0 && (reinterpret_cast<uintptr_t>(g.code) & 3) == 0) {
        const int32_t* g32 = reinterpret_cast<const int32_t*>(gcode);
        for (int q = tid; q < (ncodes + 3) / 4; q += 64 * NW) s_code32[q] = g32[q];  // the word behind the table belongs to the next front's (or to the allocation's slack)
      } else {
        int8_t* sc = reinterpret_cast<int8_t*>(s_code32);
        for (int q = tid; q < ncodes; q += 64 * NW) sc[q] = gcode[q];
      }
    }
#ifdef MFF_PROF
    unsigned long long t_prev__ = clock64();
#endif
    // ---- tables of this front into LDS
    for (int u = tid; u < 16 * T; u += 64 * NW) {
      int gn = -1, node = 0, comp = 0;
      if (u < s0) {
        node = u / BS;
        comp = u - node * BS;
        gn = nodes[node];
      } else if (u >= s16) {
        const int p = u - s16;
        if (p < g.rb) {
          node = p / BS;
          comp = p - node * BS;
          node += g.ns;
          gn = nodes[node];
        } else if (p < g.rb + MFF_BORDER) gn = -2 - (p - g.rb);
      }
      s_gnode[u] = gn;
      s_rk[u] = comp * BS * nn + gn;
      s_ck[u] = comp * nn;
      s_n1[u] = node * g.ns;
      s_nl[u] = node;
      s_upos[u] = g.upos[((long long)f * 2) * (16 * T) + u];
      s_upos[NU + u] = g.upos[((long long)f * 2 + 1) * (16 * T) + u];
    }
    const MfChild ch0 = g.child[f * 2], ch1 = g.child[f * 2 + 1];
    const double* U0 = arena + nc * ch0.offF + ((cell * ch0.nf + ch0.fidx) * (long long)ch0.L + ch0.sp) * ch0.L + ch0.sp;
    const double* U1 = arena + nc * ch1.offF + ((cell * ch1.nf + ch1.fidx) * (long long)ch1.L + ch1.sp) * ch1.L + ch1.sp;
    const double* Kc = Kst + cell * (long long)ncode * BS * BS * nn;
    const double* Bc = Brhs + cell * (long long)t * BS * nn;
    mff_sync<NW>();

    MFF_T(0);
    // ---- 1. build: acc[tt] = upper tile e = w + tt NW; lane (k, j), register r: entry (row 16 a + 4 r + k, column 16 b + j).
    // Per tile and contribution the loads are issued unconditionally -- an absent contribution reads a valid dummy address and is dropped by
    // a select -- so that the (up to twelve) loads of a tile are in flight together.
    // The children's update matrices are stored by rows of the LATER unknown (lower triangle): in the orientation of an upper tile the 16
    // lanes of a lane row run along the later unknown, i.e. down a COLUMN of the child -- 64 cache lines per load instruction.  The child
    // part is therefore gathered in the transposed orientation (lanes along the EARLIER unknown: 128-byte segments of the child's rows, four
    // lines per instruction) and turned round through a 16 x 17 LDS scratch.
    v4d acc[TPW];
    const int L0 = ch0.L, L1 = ch1.L, KS = BS * BS * nn;
    double* const tscb = NW == 1 ? Qp : Yp + w * 272;  // both buffers are free until the first panel
    // (a, b) of this wave's tile tt: a compile-time constant on the one-wave variants (tiles numbered column by column), a table look-up
    // otherwise (tiles numbered row by row: the tiles with eliminated rows come first)
    auto tile_ab = [&](auto tc, int& a, int& b) {
      constexpr int tt = decltype(tc)::value;
      if constexpr (NW == 1) {
        a = mff_row(tt);
        b = mff_col(tt);
      } else {
        const int e0 = w + tt * NW;
        const int ab = g.tilemap[e0 < ntiles ? e0 : 0];
        a = ab >> 8;
        b = ab & 255;
      }
    };
    // stencil part of ONE entry set (tile (a, b), this lane's four entries): address (a valid dummy when there is nothing to add), whether it
    // counts, and the identity padding
    auto stencil_addr = [&](int a, int b, bool act, const double* (&pk)[4], bool (&okk)[4], double (&ident)[4]) {
      const int uc = 16 * b + j;
      const int gc = s_gnode[uc], rkc = s_rk[uc], ckc = s_ck[uc], n1c = s_n1[uc], nlc = s_nl[uc];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ur = 16 * a + 4 * r + k;
        const int gr = s_gnode[ur];
        const bool sw = a == b && ur > uc;
        const int ulo = sw ? uc : ur, uhi = sw ? ur : uc;
        const int glo = sw ? gc : gr, ghi = sw ? gr : gc;
        const int cklo = sw ? ckc : s_ck[ur], nllo = sw ? nlc : s_nl[ur];
        const int rkhi = sw ? s_rk[ur] : rkc, n1hi = sw ? s_n1[ur] : n1c;
        const bool pad_s = ulo >= s0;                                // (ulo < s16 when act) identity padding of the eliminated block
        ident[r] = (act && pad_s && uhi == ulo) ? 1.0 : 0.0;
        const bool elim = act && !pad_s && ghi != -1;                // a real eliminated column unknown against a real unknown or a border row
        const bool real_hi = ghi >= 0;
        const int ci = (elim && real_hi) ? n1hi + nllo : 0;
        const int cd = code_lds ? s_code[ci] : gcode[ci];
        const bool okK = elim && real_hi && cd >= 0;
        const int m = -2 - ghi;
        const bool okB = elim && !real_hi && m < t;
        pk[r] = okB ? Bc + (m * BS * nn + cklo + glo) : Kc + (okK ? cd * KS + rkhi + cklo : 0);
        okk[r] = okK || okB;
      }
    };
    if constexpr (LROWS > 0) {
      // the tile rows that live in LDS (a < R0 <= P: their row block is eliminated here) are built FIRST, while no register tile is live yet,
      // where they will be consumed as panels, in k-major form (register r of the accumulator layout = rows 4 r .. 4 r + 3)
      for (int a = 0; a < R0; ++a)
        for (int b = a + w; b < T; b += 2 * NW) {  // two tiles per trip: their loads fly together
          const int b2 = b + NW;
          const bool two = b2 < T;
          v4d val[2] = {v4d{0.0, 0.0, 0.0, 0.0}, v4d{0.0, 0.0, 0.0, 0.0}};
          if (g.has_children) {
            const int ul = 16 * a + j;
            const int p0l = s_upos[ul], p1l = s_upos[NU + ul];
            int o0[2][4], o1[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int uh = 16 * (i ? (two ? b2 : b) : b) + 4 * r + k;
                const int p0h = s_upos[uh], p1h = s_upos[NU + uh];
                const int h0 = p0h > p0l ? p0h : p0l, l0 = p0h > p0l ? p0l : p0h, h1 = p1h > p1l ? p1h : p1l, l1 = p1h > p1l ? p1l : p1h;
                o0[i][r] = (l0 >= 0 && (i == 0 || two)) ? h0 * L0 + l0 : -1;
                o1[i][r] = (l1 >= 0 && (i == 0 || two)) ? h1 * L1 + l1 : -1;
              }
            double v0[2][4], v1[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v0[i][r] = U0[o0[i][r] >= 0 ? o0[i][r] : 0];
                v1[i][r] = U1[o1[i][r] >= 0 ? o1[i][r] : 0];
              }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              v4d vt;
#pragma unroll
              for (int r = 0; r < 4; ++r) vt[r] = (o0[i][r] >= 0 ? v0[i][r] : 0.0) + (o1[i][r] >= 0 ? v1[i][r] : 0.0);
              val[i] = accl::transpose_tile(vt, tscb, j, k);
            }
          }
          {
            const double* pk[2][4];
            bool okk[2][4];
            double ident[2][4];
            stencil_addr(a, b, a < P, pk[0], okk[0], ident[0]);
            stencil_addr(a, two ? b2 : b, two && a < P, pk[1], okk[1], ident[1]);
            double vk[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) vk[i][r] = *pk[i][r];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) val[i][r] += ident[i][r] + (okk[i][r] ? vk[i][r] : 0.0);
          }
          const int slot = a == 0 ? b : T - 1 + b;
#pragma unroll
          for (int r = 0; r < 4; ++r) Qp[slot * 256 + (4 * r + k) * 16 + j] = val[0][r];
          if (two) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Qp[(slot + NW) * 256 + (4 * r + k) * 16 + j] = val[1][r];
          }
        }
    }
    constexpr int NG = (TPW + FENCE - 1) / FENCE;  // groups of FENCE tiles whose loads are in flight together
    // pass A: the children's update matrices.  Straight-line code per group (a tile this wave does not own contributes masked dummy loads),
    // fenced between groups so that the scheduler cannot hoist every tile's loads to the top (more registers than the front itself)
    if (g.has_children) {
      mff_for<0, NG>([&](auto gc) {
        constexpr int g0 = decltype(gc)::value * FENCE;
        constexpr int GN = g0 + FENCE <= TPW ? FENCE : TPW - g0;
        if (w + g0 * NW >= ntiles) {  // (wave-uniform) nothing left for this wave
          mff_for<0, GN>([&](auto ic) { acc[g0 + decltype(ic)::value] = v4d{0.0, 0.0, 0.0, 0.0}; });
          return;
        }
        int o0[GN][4], o1[GN][4];
        mff_for<0, GN>([&](auto ic) {
          constexpr int i = decltype(ic)::value, tt = g0 + i;
          const bool own = w + tt * NW < ntiles;
          int a, b;
          tile_ab(std::integral_constant<int, tt>(), a, b);
          // transposed orientation: lane (k, j), register r: (row 16 b + 4 r + k = the later unknown, column 16 a + j = the earlier one)
          const int ul = 16 * a + j;
          const int p0l = own ? s_upos[ul] : -1, p1l = own ? s_upos[NU + ul] : -1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int uh = 16 * b + 4 * r + k;
            const int p0h = s_upos[uh], p1h = s_upos[NU + uh];
            const int h0 = p0h > p0l ? p0h : p0l, l0 = p0h > p0l ? p0l : p0h, h1 = p1h > p1l ? p1h : p1l, l1 = p1h > p1l ? p1l : p1h;
            o0[i][r] = l0 >= 0 ? h0 * L0 + l0 : -1;   // (upos is -1 for padding and for an absent child)
            o1[i][r] = l1 >= 0 ? h1 * L1 + l1 : -1;
          }
        });
        double v0[GN][4], v1[GN][4];
#pragma unroll
        for (int i = 0; i < GN; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v0[i][r] = U0[o0[i][r] >= 0 ? o0[i][r] : 0];
            v1[i][r] = U1[o1[i][r] >= 0 ? o1[i][r] : 0];
          }
        mff_for<0, GN>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          v4d vt;
#pragma unroll
          for (int r = 0; r < 4; ++r) vt[r] = (o0[i][r] >= 0 ? v0[i][r] : 0.0) + (o1[i][r] >= 0 ? v1[i][r] : 0.0);
          acc[g0 + i] = accl::transpose_tile(vt, tscb, j, k);
        });
        asm volatile("" ::: "memory");
      });
    } else {
      mff_for<0, TPW>([&](auto tc) { acc[decltype(tc)::value] = v4d{0.0, 0.0, 0.0, 0.0}; });
    }
    // pass B: tiles whose row block is eliminated here (a < P): stencil entries / canonical loads / identity padding.  Off the diagonal
    // tiles the row unknown is the earlier one (lo); inside a diagonal tile either order occurs.
    // (row-by-row numbering of the register tiles, rows >= R0) the tiles with a < P come first
    const int e_u = P > R0 ? (P * T - P * (P - 1) / 2) - (R0 * T - R0 * (R0 - 1) / 2) : 0;
    if constexpr (NW == 1) {
      // one wave: (a, b) is a compile-time constant per register set and the tiles are numbered column by column -- one tile at a time
      mff_for<0, TPW>([&](auto tc) {
        constexpr int tt = decltype(tc)::value;
        constexpr int a = mff_row(tt), b = mff_col(tt);
        if (tt < ntiles && a < P) {
          const double* pk[4];
          bool okk[4];
          double ident[4];
          stencil_addr(a, b, true, pk, okk, ident);
          double vk[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) vk[r] = *pk[r];
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[tt][r] += ident[r] + (okk[r] ? vk[r] : 0.0);
        }
      });
    } else {
      mff_for<0, NG>([&](auto gc) {
        constexpr int g0 = decltype(gc)::value * FENCE;
        constexpr int GN = g0 + FENCE <= TPW ? FENCE : TPW - g0;
        // (wave-uniform) the tiles with eliminated rows are a prefix of the row-by-row numbering
        if (w + g0 * NW >= e_u) return;
        const double* pk[GN][4];
        bool okk[GN][4];
        double ident[GN][4];
        mff_for<0, GN>([&](auto ic) {
          constexpr int i = decltype(ic)::value, tt = g0 + i;
          int a, b;
          tile_ab(std::integral_constant<int, tt>(), a, b);
          stencil_addr(a, b, w + tt * NW < ntiles && a < P, pk[i], okk[i], ident[i]);
        });
        double vk[GN][4];
  #pragma unroll
        for (int i = 0; i < GN; ++i)
  #pragma unroll
          for (int r = 0; r < 4; ++r) vk[i][r] = *pk[i][r];
        mff_for<0, GN>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
  #pragma unroll
          for (int r = 0; r < 4; ++r) acc[g0 + i][r] += ident[i][r] + (okk[i][r] ? vk[i][r] : 0.0);
        });
        asm volatile("" ::: "memory");
      });
    }

    MFF_T(1);
    // ---- 2. elimination in panels of 16
    int bad = 0;
    for (int p = 0; p < P; ++p) {
      // tile (p, b) of the panel sits at Qp[qb + 256 b]: an LDS row is consumed where it stands, a register row is copied to slot b
      const int qb = (LROWS > 0 && p < R0 && p > 0) ? (T - 1) * 256 : 0;
      if (!(LROWS > 0 && p < R0)) {
        mff_for<0, TPW>([&](auto tc) {
          constexpr int tt = decltype(tc)::value;
          const int e = w + tt * NW;
          if (e < ntiles) {
            int a, b;
            if constexpr (NW == 1) { a = mff_row(tt); b = mff_col(tt); }
            else { const int ab = g.tilemap[e]; a = ab >> 8; b = ab & 255; }
            if (a == p) {
#pragma unroll
              for (int r = 0; r < 4; ++r) Qp[b * 256 + (4 * r + k) * 16 + j] = acc[tt][r];
            }
          }
        });
      }
      mff_sync<NW>();
      MFF_T(2);
      // every wave inverts the diagonal tile itself (no broadcast of N, no second barrier): T = -D, all pivots negative
      double nm[1][1][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) nm[0][0][r] = -Qp[qb + p * 256 + (4 * r + k) * 16 + j];
      accl::Sweep<16>::run(nm, ubuf + w * 64, j, k, bad);   // nm = T^-1 = -N
      MFF_T(3);
      if constexpr (NW == 1) {
        // one wave: (a, b) of every register set is a compile-time constant and Y'_a stays in registers (in the accumulator layout register q
        // of a tile IS its k-slab q as the A operand of the transpose): no second LDS buffer, no barrier between the products
        v4d yreg[TMAX];
        mff_for<1, TMAX>([&](auto ac) {
          constexpr int a = decltype(ac)::value;
          if (a > p && a < T) {
            v4d c = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) c = mm(nm[0][0][q], Qp[a * 256 + (4 * q + k) * 16 + j], c);
            yreg[a] = c;
          }
        });
        mff_for<0, TPW>([&](auto tc) {
          constexpr int tt = decltype(tc)::value;
          constexpr int a = mff_row(tt), b = mff_col(tt);
          if constexpr (a >= 1) {
            if (a > p && b < T) {
              v4d c = acc[tt];
#pragma unroll
              for (int q = 0; q < 4; ++q) c = mm(yreg[a][q], Qp[b * 256 + (4 * q + k) * 16 + j], c);
              acc[tt] = c;
            }
          }
        });
      } else {
        // Y'_a = (-N) E_a^T for the row blocks a = p + 1 + w, + NW, ...:  C[k][i] = sum_l (-N)[l][k] E_a^T[l][i]
        for (int a = p + 1 + w; a < T; a += NW) {
          v4d c = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int q = 0; q < 4; ++q) c = mm(nm[0][0][q], Qp[qb + a * 256 + (4 * q + k) * 16 + j], c);
#pragma unroll
          for (int r = 0; r < 4; ++r) Yp[a * 256 + (4 * r + k) * 16 + j] = c[r];
        }
        mff_sync<NW>();
        MFF_T(4);
        // trailing tiles (a, b), p < a <= b:  acc += Y'_a^T E_b^T
        mff_for<0, TPW>([&](auto tc) {
          constexpr int tt = decltype(tc)::value;
          const int e = w + tt * NW;
          if (e < ntiles) {
            int a, b;
            tile_ab(tc, a, b);
            if (a > p) {
              v4d c = acc[tt];
#pragma unroll
              for (int q = 0; q < 4; ++q) c = mm(Yp[a * 256 + (4 * q + k) * 16 + j], Qp[qb + b * 256 + (4 * q + k) * 16 + j], c);
              acc[tt] = c;
            }
          }
        });
        if constexpr (LROWS > 0) {
          // the LDS rows behind the panel (p < a < R0: row 1 at step 0) take the same update in place
          for (int a = p + 1; a < R0; ++a)
            for (int b = a + w; b < T; b += NW) {
              double* tile = Qp + (T - 1 + b) * 256;  // (a == 1)
              v4d c;
#pragma unroll
              for (int r = 0; r < 4; ++r) c[r] = tile[(4 * r + k) * 16 + j];
#pragma unroll
              for (int q = 0; q < 4; ++q) c = mm(Yp[a * 256 + (4 * q + k) * 16 + j], Qp[qb + b * 256 + (4 * q + k) * 16 + j], c);
#pragma unroll
              for (int r = 0; r < 4; ++r) tile[(4 * r + k) * 16 + j] = c[r];
            }
        }
      }
      MFF_T(5);
      mff_sync<NW>();  // the next panel overwrites Qp / Yp
      MFF_T(6);
    }

    // ---- 3. the update matrix (lower triangle, arena layout of the group: F22 at (sp, sp), ld = L), rows up to the border
    {
      double* U = arena + nc * g.offF + batch * (long long)g.L * g.L + (long long)g.sp * g.L + g.sp;
      double* tsc = NW == 1 ? Qp : Yp + w * 272;  // (one wave: Qp is free after the last panel; TMAX >= 2)
      const int nrow = g.rb + MFF_BORDER;
      mff_for<0, TPW>([&](auto tc) {
        constexpr int tt = decltype(tc)::value;
        const int e = w + tt * NW;
        int a, b;
        if constexpr (NW == 1) { a = mff_row(tt); b = mff_col(tt); }
        else { const int ab = e < ntiles ? g.tilemap[e] : 0; a = ab >> 8; b = ab & 255; }
        if (e < ntiles && a >= P) {
          const v4d y = accl::transpose_tile(acc[tt], tsc, j, k);  // y[r] at lane (k, j) = F[row 16 b + 4 r + k][column 16 a + j]
          const int col = 16 * a + j - s16;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * b + 4 * r + k - s16;
            if (row < nrow && col <= row) U[(long long)row * g.L + col] = y[r];
          }
        }
      });
    }
    MFF_T(7);
    if (bad && l == 0 && info) atomicCAS(&info[cell], 0, stepcode);
  }
}


template <int BS>
void launch_mf_front_bs(const MfFrontDev& g, const double* Kst, const double* Brhs, double* arena, long long nc, long long nbatch, int nn,
                        int ncode, int t, int32_t* info, int stepcode, hipStream_t st) {
  // a launch holds at most 2^32 - 1 work-items (AQL grid size): big batches go in pieces of 2^21 fronts
  for (long long b0 = 0; b0 < nbatch; b0 += 1ll << 21) {
  const unsigned grid = (unsigned)std::min(nbatch - b0, 1ll << 21);
#define HOMMX_MFF(BS_, NW_, TMAX_, TPW_, MINB_, FENCE_, ...)                                                                                 \
  hipLaunchKernelGGL((k_mf_front<BS_, NW_, TMAX_, TPW_, MINB_, FENCE_, ##__VA_ARGS__>), dim3(grid), dim3(64 * NW_), 0, st, g, Kst, Brhs, arena, \
                     nc, b0, nn, ncode, t, info, stepcode)
  // register budgets (waves per SIMD): 10 tiles = 80 VGPRs of matrix -> 4; 21 tiles -> 2; four waves x 9 tiles -> 3; x 20 -> 2; eight x 24 -> 2
#ifdef MFF_ONLY_ONE  // dev builds: one instantiation (register experiments)
#define HOMMX_MFF_BS(BS_) HOMMX_MFF(3, 8, 19, 24, 2, 3)
#elif defined(MFF_DEV_BS3)  // dev builds: the two multi-wave variants of three unknowns per node only (phase timing, seconds to compile)
#define HOMMX_MFF_BS(BS_)                                \
  do {                                                   \
    if (g.T <= 12) HOMMX_MFF(3, 4, 12, 20, 2, 4);        \
    else if (g.T <= 19) HOMMX_MFF(3, 8, 19, 24, 2, 3);   \
    else HOMMX_MFF(3, 8, 21, 24, 2, 2, 2);               \
  } while (0)
#else
#define HOMMX_MFF_BS(BS_)                                \
  do {                                                   \
    if (g.T <= 4) HOMMX_MFF(BS_, 1, 4, 10, 4, 1);        \
    else if (g.T <= 6) HOMMX_MFF(BS_, 1, 6, 21, 2, 2);   \
    else if (g.T <= 8) HOMMX_MFF(BS_, 4, 8, 9, 3, 3);    \
    else if (g.T <= 12) HOMMX_MFF(BS_, 4, 12, 20, 2, 4); \
    else if (g.T <= 19) HOMMX_MFF(BS_, 8, 19, 24, 2, 3); \
    else HOMMX_MFF(BS_, 8, 21, 24, 2, 2, 2);             \
  } while (0)
#endif
  HOMMX_MFF_BS(BS);
#undef HOMMX_MFF_BS
#undef HOMMX_MFF
  }
}

#ifdef MFF_PROF
extern "C" void hommx_mff_prof_read(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(mff_prof), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(mff_prof), z, sizeof(z));
  }
}
#endif

}  // namespace hommx
