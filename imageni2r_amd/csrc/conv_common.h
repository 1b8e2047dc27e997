// Shared pieces of the implicit-GEMM convolution kernels (gfx950 fp32 MFMA).
#pragma once
#include "i2r_common.h"

// One K-chunk of BKC rows: acc[tm][tn] += A[k][m-tile] * B[k][n-tile], operands staged in LDS as [k][m] / [k][n]
// (one float per lane per operand for v_mfma_f32_32x32x2_f32: lane l holds row/col l&31 of k-slice 2*ks + (l>>5)).
template <int WM, int WN, int TM, int TN, int BKC>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ As, const float* __restrict__ Bs, int lda,
                                          int ldb, f32x16 (&acc)[TM][TN], int wm, int wn, int l31, int h) {
  // operands of k-step ks+1 are fetched from LDS before the MFMAs of k-step ks are issued (register double buffer),
  // so the ds_read latency sits under 4 x 64 MFMA cycles instead of in front of them
  const float* ap = As + h * lda + wm * TM * 32 + l31;
  const float* bp = Bs + h * ldb + wn * TN * 32 + l31;
  float a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) a0[t] = ap[t * 32];
#pragma unroll
  for (int t = 0; t < TN; ++t) b0[t] = bp[t * 32];
#pragma unroll
  for (int ks = 0; ks < BKC / 2; ++ks) {
    if (ks + 1 < BKC / 2) {
#pragma unroll
      for (int t = 0; t < TM; ++t) a1[t] = ap[(2 * ks + 2) * lda + t * 32];
#pragma unroll
      for (int t = 0; t < TN; ++t) b1[t] = bp[(2 * ks + 2) * ldb + t * 32];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ABOVE the MFMAs (hipcc otherwise sinks it below them)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[tm], b0[tn], acc[tm][tn], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TM; ++t) a0[t] = a1[t];
#pragma unroll
    for (int t = 0; t < TN; ++t) b0[t] = b1[t];
  }
}
