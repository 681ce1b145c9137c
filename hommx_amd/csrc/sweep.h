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

// WSYNC: barrier among the lanes that share ubuf/wbuf.  With one wave per workgroup __syncthreads()
// lowers to a wait on the LDS counter; kernels with several independent waves per workgroup pass a
// wave-local fence instead.
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
    bad |= !(d > 0.0);
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
