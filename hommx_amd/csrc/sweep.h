// sweep.h -- in-register symmetric Gauss-Jordan sweep of an NB x NB SPD matrix by ONE wavefront.
//
// "GJ layout": lane l owns column c = l % NB and the RPL = NB*NB/64 rows r0 + i, r0 = (l / NB) * RPL.
// After the sweep the registers hold  -S^-1  (symmetric).  One pivot row per step is broadcast through
// 2 x NB doubles of LDS (ubuf: raw row, wbuf: row / pivot); the loop over pivots is unrolled by
// template recursion so that every register index is a compile-time constant.
#pragma once
#include <hip/hip_runtime.h>

namespace hommx {

template <int NB>
struct Cfg {
  static constexpr int RPL = NB * NB / 64;  // rows per lane in GJ layout
  static constexpr int CG = 64 / NB;        // lane groups (each owns RPL rows of every column)
  static constexpr int NT = NB / 16;        // 16x16 tiles per dimension
  static constexpr int KK = NB / 4;         // k-steps of the 16x16x4 MFMA
};

// "not a usable pivot" (d <= 0, denormal, inf or NaN) of a wave-uniform double, in integer arithmetic on its high word:
// d comes out of v_readlane, i.e. out of SGPRs, and this keeps the test on the scalar unit (a v_cmp_f64 per pivot is a
// VALU issue slot, and those are what the sweep is short of).
__device__ __forceinline__ int bad_pivot_hi(int hi) {
  return (unsigned)(hi - 1) >= 0x7fefffffu;  // hi in [0x00000001, 0x7fefffff] <=> positive normal finite
}
__device__ __forceinline__ int bad_pivot(double d) { return bad_pivot_hi(__double2hiint(d)); }
// v_readlane of a double that also flags an unusable pivot, tested on the SGPR the high word arrives in
__device__ __forceinline__ double readlane_pivot(double v, int lane, int& bad) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  bad |= bad_pivot_hi(hi);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/d to ~1 ulp: v_rcp_f64 seed + two Newton steps (operands are well inside the normal range).
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = fma(-d, r, 1.0);
  r = fma(e, r, r);
  e = fma(-d, r, 1.0);
  r = fma(e, r, r);
  return r;
}

// gfx950 row swaps on doubles (row r = lanes 16r .. 16r+15; measured with tools/probe_permlane.hip):
//   swap16(a, b): a' = [a0, b0, a2, b2], b' = [a1, b1, a3, b3]        (v_permlane16_swap_b32 on both dwords)
//   swap32(a, b): a' = [a0, a1, b0, b1], b' = [a2, a3, b2, b3]        (v_permlane32_swap_b32)
// VALU-latency cross-row traffic: what the reductions over the four 16-lane rows use instead of ds_bpermute.
__device__ __forceinline__ void swap16(double& a, double& b) {
  auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap32(double& a, double& b) {
  auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]);
  b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// v(l) + v(l ^ 32)  resp.  + v(l ^ 16), in every lane
__device__ __forceinline__ double add_xor32(double v) {
  double t = v, u = v;
  swap32(t, u);  // t = [v0, v1, v0, v1], u = [v2, v3, v2, v3]
  return t + u;
}
__device__ __forceinline__ double add_xor16(double v) {
  double t = v, u = v;
  swap16(t, u);  // t = [v0, v0, v2, v2], u = [v1, v1, v3, v3]
  return t + u;
}

// Empty volatile asm through which a value is threaded: pins the computation of `v` before this program point.
__device__ __forceinline__ double pin_here(double v) {
  int hi = __double2hiint(v), lo = __double2loint(v);
  asm volatile("" : "+v"(hi), "+v"(lo));
  return __hiloint2double(hi, lo);
}

// WSYNC: barrier among the lanes that share ubuf/wbuf.  With one wave per workgroup __syncthreads()
// lowers to a wait on the LDS counter; kernels with several independent waves per workgroup pass a
// wave-local fence instead.
// Ordering point for LDS traffic WITHIN one wavefront: the LDS executes a wave's instructions in issue order, so a
// ds_read issued after a ds_write of the same wave observes it without an s_waitcnt in between; all that is needed is
// that the compiler keeps the program order (HOMMX_SWEEP_WAIT restores the full wait, for A/B runs).
__device__ __forceinline__ void lds_order() {
#ifdef HOMMX_SWEEP_WAIT
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

struct SyncBlock {
  static __device__ __forceinline__ void sync() { __syncthreads(); }
};
struct SyncWave {
  static __device__ __forceinline__ void sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};

template <int NB, int K, class SYNC = SyncBlock>
struct SweepStep {
  static __device__ __forceinline__ void run(double (&s)[Cfg<NB>::RPL], double* __restrict__ ubuf,
                                             double* __restrict__ wbuf, int c, int g, int r0, int& bad) {
    constexpr int RPL = Cfg<NB>::RPL;
    constexpr int gk = K / RPL, ik = K % RPL;
    const double d = readlane_f64(s[ik], gk * NB + K);
    bad |= bad_pivot(d);
    const double pinv = fast_rcp(d);
    if (g == gk) {
      const double u = s[ik];
      ubuf[c] = u;
      wbuf[c] = (c == K) ? -pinv : u * pinv;
    }
    SYNC::sync();
    const double w = wbuf[c];
    double x[RPL];
#pragma unroll
    for (int i = 0; i < RPL; i += 2) {
      const double2 t = *reinterpret_cast<const double2*>(&ubuf[r0 + i]);
      x[i] = t.x;
      x[i + 1] = t.y;
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i) s[i] = fma(-x[i], w, s[i]);
    if (c == K) {  // pivot column: new column k == scaled pivot row (symmetry)
#pragma unroll
      for (int i = 0; i < RPL; i += 2) {
        const double2 t = *reinterpret_cast<const double2*>(&wbuf[r0 + i]);
        s[i] = t.x;
        s[i + 1] = t.y;
      }
    }
    if (g == gk) s[ik] = w;  // pivot row
    // no barrier needed here: the next step's ubuf/wbuf stores follow these loads in program order
    // of the same wave, and the LDS pipeline is in-order per wave
    SweepStep<NB, K + 1, SYNC>::run(s, ubuf, wbuf, c, g, r0, bad);
  }
};
template <int NB, class SYNC>
struct SweepStep<NB, NB, SYNC> {
  static __device__ __forceinline__ void run(double (&)[Cfg<NB>::RPL], double*, double*, int, int, int, int&) {}
};

}  // namespace hommx

namespace hommx {

// ---- predicated fix-ups with COMPILE-TIME exec masks ---------------------------------------------------------------
// In the block layout the lanes of block row / column k are known constants (bi = l >> 3, bj = l & 7), so the pivot
// row / column fix-ups can run under a literal exec mask instead of one the compiler derives from v_cmp and has to
// keep in (or spill from) SGPRs for all 8 + 8 values of k.  exec is saved and restored inside the one asm statement.
template <int KB> struct LaneMask {
  static constexpr unsigned col_lo = 0x01010101u << KB, col_hi = 0x01010101u << KB;              // bj == KB
  static constexpr unsigned row_lo = KB < 4 ? 0xFFu << (8 * (KB & 3)) : 0u;                      // bi == KB
  static constexpr unsigned row_hi = KB < 4 ? 0u : 0xFFu << (8 * (KB & 3));
  static constexpr unsigned dia_lo = KB < 4 ? 1u << ((9 * KB) & 31) : 0u;                        // lane 9 KB
  static constexpr unsigned dia_hi = KB < 4 ? 0u : 1u << ((9 * KB - 32) & 31);
};
#define HOMMX_EXEC_IN "s_mov_b64 %[sv], exec\n\ts_mov_b32 exec_lo, %[lo]\n\ts_mov_b32 exec_hi, %[hi]\n\t"
#define HOMMX_EXEC_OUT "s_mov_b64 exec, %[sv]"
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_mov(double& d, double v) {
  unsigned long long sv;
  asm(HOMMX_EXEC_IN "v_mov_b64 %[d], %[v]\n\t" HOMMX_EXEC_OUT
      : [d] "+v"(d), [sv] "=&s"(sv) : [v] "v"(v), [lo] "i"(LO), [hi] "i"(HI));
}
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_mov4(double& d0, double& d1, double& d2, double& d3, double v0, double v1, double v2,
                                            double v3) {
  unsigned long long sv;
  asm(HOMMX_EXEC_IN "v_mov_b64 %[d0], %[v0]\n\tv_mov_b64 %[d1], %[v1]\n\tv_mov_b64 %[d2], %[v2]\n\tv_mov_b64 %[d3], %[v3]\n\t"
      HOMMX_EXEC_OUT
      : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [sv] "=&s"(sv)
      : [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3), [lo] "i"(LO), [hi] "i"(HI));
}
// d_q = u_q * p on the lanes of (LO, HI)
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_scale4(double& d0, double& d1, double& d2, double& d3, double u0, double u1, double u2,
                                              double u3, double p) {
  unsigned long long sv;
  asm(HOMMX_EXEC_IN
      "v_mul_f64 %[d0], %[u0], %[p]\n\tv_mul_f64 %[d1], %[u1], %[p]\n\tv_mul_f64 %[d2], %[u2], %[p]\n\t"
      "v_mul_f64 %[d3], %[u3], %[p]\n\t" HOMMX_EXEC_OUT
      : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [sv] "=&s"(sv)
      : [u0] "v"(u0), [u1] "v"(u1), [u2] "v"(u2), [u3] "v"(u3), [p] "v"(p), [lo] "i"(LO), [hi] "i"(HI));
}
// d = -p on the lanes of (LO, HI)
template <unsigned LO, unsigned HI>
__device__ __forceinline__ void masked_neg(double& d, double p) {
  unsigned long long sv;
  asm(HOMMX_EXEC_IN "v_mul_f64 %[d], -1.0, %[p]\n\t" HOMMX_EXEC_OUT
      : [d] "+v"(d), [sv] "=&s"(sv) : [p] "v"(p), [lo] "i"(LO), [hi] "i"(HI));
}

// ---- same sweep, "BLK layout": lane l owns the BS x BS block (bi = l >> 3, bj = l & 7), BS = NB / 8 ----------
// Element (r, q) of the block is matrix entry (BS*bi + r, BS*bj + q) and lives in s[r * BS + q].
// Per pivot a lane needs only BS entries of the pivot row for its rows and BS for its columns (2*BS LDS
// doubles instead of RPL + 1 in the column-strip layout): the sweep is LDS-bandwidth bound, so this is what
// sets its speed.  The pivot column needs no LDS at all (BS predicated multiplies).
template <int NB, int K>
struct SweepStepBlk {
  static constexpr int BS = NB / 8;
  // Entered with the RAW pivot row K already in ubuf (published one step earlier) and `d`, `pinv` = pivot K and
  // its reciprocal.  Only the raw row goes through LDS (one write of BS doubles by the 8 owner lanes, 2*BS doubles
  // read per lane); the scaling by 1/pivot is BS + BS multiplies per lane.
  static __device__ __forceinline__ void run(double (&s)[BS * BS], double* __restrict__ ubuf, int bi, int bj,
                                             int& bad, double d, double pinv) {
    constexpr int kb = K / BS, kr = K % BS;
    constexpr bool more = (K + 1 < NB);
    constexpr int K1 = more ? K + 1 : K;
    constexpr int kb1 = K1 / BS, kr1 = K1 % BS;
    lds_order();  // the reads below queue behind this wave's own ubuf stores: no wait for their completion
    double ur[BS], uc[BS], t[BS];
#pragma unroll
    for (int q = 0; q < BS; q += 2) {
      const double2 a = *reinterpret_cast<const double2*>(&ubuf[BS * bi + q]);
      const double2 b = *reinterpret_cast<const double2*>(&ubuf[BS * bj + q]);
      ur[q] = a.x; ur[q + 1] = a.y;
      uc[q] = b.x; uc[q + 1] = b.y;
    }
#pragma unroll
    for (int r = 0; r < BS; ++r) t[r] = ur[r] * pinv;  // scaled pivot-row entries at my rows (== new pivot column)
    double dn = 1.0, pn = 1.0;
    if (more) {
      // pivot row K+1 first: update its BS entries, fix the one in pivot column K, publish raw, start 1/pivot
      double e[BS];
#pragma unroll
      for (int q = 0; q < BS; ++q) e[q] = fma(-t[kr1], uc[q], s[kr1 * BS + q]);
#ifndef HOMMX_NO_ASM_MASKS
      if constexpr (BS == 4) masked_mov<LaneMask<kb>::col_lo, LaneMask<kb>::col_hi>(e[kr], t[kr1]);
      else
#endif
      if (bj == kb) e[kr] = t[kr1];
#pragma unroll
      for (int q = 0; q < BS; ++q) s[kr1 * BS + q] = e[q];
      if (bi == kb1) {
#pragma unroll
        for (int q = 0; q < BS; q += 2) *reinterpret_cast<double2*>(&ubuf[BS * bj + q]) = double2{e[q], e[q + 1]};
      }
      dn = readlane_pivot(e[kr1], 9 * kb1, bad);
      pn = pin_here(fast_rcp(dn));
    }
#pragma unroll
    for (int r = 0; r < BS; ++r)
      if (!(more && r == kr1)) {
#pragma unroll
        for (int q = 0; q < BS; ++q) s[r * BS + q] = fma(-t[r], uc[q], s[r * BS + q]);
      }
#ifndef HOMMX_NO_ASM_MASKS
    if constexpr (BS == 4) {
      using LM = LaneMask<kb>;
      masked_mov4<LM::col_lo, LM::col_hi>(s[0 * BS + kr], s[1 * BS + kr], s[2 * BS + kr], s[3 * BS + kr], t[0], t[1], t[2], t[3]);
      masked_scale4<LM::row_lo, LM::row_hi>(s[kr * BS + 0], s[kr * BS + 1], s[kr * BS + 2], s[kr * BS + 3], uc[0], uc[1], uc[2],
                                            uc[3], pinv);
      masked_neg<LM::dia_lo, LM::dia_hi>(s[kr * BS + kr], pinv);
    } else
#endif
    {
      if (bj == kb) {  // pivot column
#pragma unroll
        for (int r = 0; r < BS; ++r) s[r * BS + kr] = t[r];
      }
      if (bi == kb) {  // pivot row: scaled raw row; (K, K) = -1/pivot
#pragma unroll
        for (int q = 0; q < BS; ++q) s[kr * BS + q] = uc[q] * pinv;
        if (bj == kb) s[kr * BS + kr] = -pinv;
      }
    }
    SweepStepBlk<NB, K + 1>::run(s, ubuf, bi, bj, bad, dn, pn);
  }
};
template <int NB>
struct SweepStepBlk<NB, NB> {
  static __device__ __forceinline__ void run(double (&)[(NB / 8) * (NB / 8)], double*, int, int, int&, double,
                                             double) {}
};

// entry: pivot 0 and its reciprocal, then the pipelined steps
template <int NB>
__device__ __forceinline__ void sweep_blk(double (&s)[(NB / 8) * (NB / 8)], double* ubuf, int bi, int bj,
                                          int& bad) {
  constexpr int BS = NB / 8;
  if (bi == 0) {
#pragma unroll
    for (int q = 0; q < BS; q += 2) *reinterpret_cast<double2*>(&ubuf[BS * bj + q]) = double2{s[q], s[q + 1]};
  }
  int ub = 0;  // wave-uniform flag (fed by v_readlane results only): stays on the scalar unit until the single OR below
  const double d0 = readlane_pivot(s[0], 0, ub);
  SweepStepBlk<NB, 0>::run(s, ubuf, bi, bj, ub, d0, fast_rcp(d0));
  bad |= ub;
}

}  // namespace hommx
