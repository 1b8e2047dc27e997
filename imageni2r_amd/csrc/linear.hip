// Small dense layers of the noise-embedding path (models/networks.py:39-43 used at :440-447 and :170):
// map_layer0/1 and the 36 per-block `affine` projections.  Rows = batch (1 in sampling, B in training), so these
// are GEMV-shaped and bound by reading the weight matrix once: one wavefront streams one weight row, keeps it in
// registers and applies it to every batch row (weight-stationary), wave-shuffle reduction per output.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

#define LIN_MAXK_PER_LANE 32  // K <= 2048

// y[m][n] = act( sum_k x[m][k] * w[n][k] + b[n] );  pre[m][n] (optional) keeps the pre-activation for backward
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         float* __restrict__ pre, int M, int K, int N, int act) {
  int lane = threadIdx.x & 63;
  int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;  // wave-uniform
  float wr[LIN_MAXK_PER_LANE];
  int nk = (K + 63) / 64;
#pragma unroll
  for (int j = 0; j < LIN_MAXK_PER_LANE; ++j) {
    int k = j * 64 + lane;
    wr[j] = (j < nk && k < K) ? w[(long)n * K + k] : 0.0f;
  }
  float bn = bias ? bias[n] : 0.0f;
  for (int m = 0; m < M; ++m) {
    const float* xr = x + (long)m * K;
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < LIN_MAXK_PER_LANE; ++j) {
      int k = j * 64 + lane;
      if (j < nk && k < K) acc += xr[k] * wr[j];
    }
    acc = wave_sum(acc) + bn;
    if (lane == 0) {
      if (pre) pre[(long)m * N + n] = acc;
      y[(long)m * N + n] = act ? silu_f(acc) : acc;
    }
  }
}

// dx[m][k] = sum_n dy[m][n] * w[n][k]
__global__ void linear_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                 int M, int K, int N) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)M * K) return;
  int k = (int)(i % K), m = (int)(i / K);
  const float* dyr = dy + (long)m * N;
  float acc = 0.0f;
  for (int n = 0; n < N; ++n) acc += dyr[n] * w[(long)n * K + k];
  dx[i] = acc;
}
// dw[n][k] = sum_m dy[m][n] * x[m][k]
__global__ void linear_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                 int M, int K, int N) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)N * K) return;
  int k = (int)(i % K), n = (int)(i / K);
  float acc = 0.0f;
  for (int m = 0; m < M; ++m) acc += dy[(long)m * N + n] * x[(long)m * K + k];
  dw[i] = acc;
}

extern "C" int i2r_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, float* pre, int M,
                                  int K, int N, int act_silu, i2r_stream_t stream) {
  I2R_REQUIRE(x && w && y && M > 0 && K > 0 && N > 0, "linear_fwd: bad argument");
  I2R_REQUIRE(K <= 64 * LIN_MAXK_PER_LANE, "linear_fwd: in_features %d > %d unsupported", K, 64 * LIN_MAXK_PER_LANE);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(i2r_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, pre,
                     M, K, N, act_silu);
  return i2r_check_launch("linear_fwd");
}

extern "C" int i2r_linear_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, int M,
                                  int K, int N, i2r_stream_t stream) {
  I2R_REQUIRE(x && w && dy && M > 0 && K > 0 && N > 0, "linear_bwd: bad argument");
  if (dx)
    hipLaunchKernelGGL(linear_dx_kernel, dim3(i2r_cdiv((long)M * K, 256)), dim3(256), 0, (hipStream_t)stream, dy, w,
                       dx, M, K, N);
  if (dw)
    hipLaunchKernelGGL(linear_dw_kernel, dim3(i2r_cdiv((long)N * K, 256)), dim3(256), 0, (hipStream_t)stream, dy, x,
                       dw, M, K, N);
  return i2r_check_launch("linear_bwd");
}
