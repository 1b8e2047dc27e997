// Small dense layers of the noise-embedding path (models/networks.py:39-43 used at :440-447 and :170):
// map_layer0/1 and the 36 per-block `affine` projections, forward and backward.  Rows = batch (1 in sampling, the
// training batch otherwise), K, N <= 2048: a few MFLOP each, so what matters is latency, not throughput.
//
// One kernel serves the three products (y = x W^T, dx = dy W, dW = dy^T x) through operand strides:
//     C[i][j] = sum_k A(i,k) * B(j,k),   A(i,k) = A[i*a_rs + k*a_ks],   B(j,k) = B[j*b_rs + k*b_ks]
// A workgroup owns one 32x32 output tile; its 4 wavefronts split the contraction range, each running a chain of exact
// fp32 MFMAs (v_mfma_f32_32x32x2_f32, operands straight from global memory: 16-B loads along a unit-stride
// contraction, 128-B coalesced row segments otherwise), then the four partial tiles are summed through LDS in a fixed
// order (deterministic) and the epilogue adds the bias, keeps the pre-activation for backward and applies SiLU.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

struct GemmP {
  const float* A;
  const float* B;
  const float* bias;   // per output column j (may be null)
  float* C;            // (M, N) row-major, leading dimension N
  float* pre;          // optional copy of C before the activation
  float* rowsum;       // optional: rowsum[i] = sum_k A(i,k)   (bias gradient of the dW product)
  int M, N, K;
  long a_rs, a_ks, b_rs, b_ks;
  int act;
};

__device__ __forceinline__ void load4(const float* base, long rs, long ks, int idx, int lim, int k, int kend, bool vec,
                                      float (&v)[4]) {
  const bool ok = idx < lim;
  if (vec && ok && k + 3 < kend) {
    float4 t = *reinterpret_cast<const float4*>(base + (long)idx * rs + k);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (ok && k + j < kend) ? base[(long)idx * rs + (long)(k + j) * ks] : 0.0f;
  }
}

template <int NW>
__device__ __forceinline__ void gemm_tile(const GemmP& p, int vecA, int vecB, int bx, int by) {
  __shared__ float red[NW][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int i0 = by * 32, j0 = bx * 32;
  int kc = (p.K + NW - 1) / NW;
  kc = (kc + 7) & ~7;  // contraction slice per wave, multiple of the 8-deep step
  const int kbeg = wave * kc;
  int kend = kbeg + kc;
  if (kend > p.K) kend = p.K;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  // step of 8: MFMA jj pairs A's and B's slot h with contraction index k0 + 4h + jj
  // four steps' operands are requested before the first MFMA consumes any: the chain is latency-bound otherwise
  // (kc is a multiple of 8; steps past kend load zeros)
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    float a[4][4], b[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      load4(p.A, p.a_rs, p.a_ks, i0 + l31, p.M, k0 + 8 * u + 4 * h, kend, vecA, a[u]);
      load4(p.B, p.b_rs, p.b_ks, j0 + l31, p.N, k0 + 8 * u + 4 * h, kend, vecB, b[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][jj], b[u][jj], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][l31] = acc[r];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16 / NW; ++q) {
    int idx = tid + 64 * NW * q;
    int r = idx >> 5, c = idx & 31;
    int i = i0 + r, j = j0 + c;
    if (i < p.M && j < p.N) {
      float v = red[0][r][c];
#pragma unroll
      for (int w = 1; w < NW; ++w) v += red[w][r][c];
      if (p.bias) v += p.bias[j];
      long o = (long)i * p.N + j;
      if (p.pre) p.pre[o] = v;
      p.C[o] = p.act ? silu_f(v) : v;
    }
  }
  if (p.rowsum && bx == 0 && tid < 32 && i0 + tid < p.M) {
    float s = 0.0f;
    for (int k = 0; k < p.K; ++k) s += p.A[(long)(i0 + tid) * p.a_rs + (long)k * p.a_ks];
    p.rowsum[i0 + tid] = s;
  }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void small_gemm_kernel(GemmP p, int vecA, int vecB) {
  gemm_tile<NW>(p, vecA, vecB, blockIdx.x, blockIdx.y);
}

__device__ __forceinline__ bool vec_ok_dev(const float* ptr, long rs, long ks) {
  return ks == 1 && (rs % 4) == 0 && (((uintptr_t)ptr) & 15) == 0;
}

// All the `affine` layers of the network (models/networks.py:170, one Linear(4M -> 2 Cout) per UNetBlock, every one a
// function of the same embedding) in ONE launch per product instead of one per layer:
//   mode 0  y_l  (M, N_l) = emb W_l^T + b_l                          -> out_base + off_l
//   mode 1  dW_l (N_l, K) = dy_l^T emb,  db_l = column sums of dy_l   (dy_l at base + off_l)
//   mode 2  partial dx_l (M, K) = dy_l W_l                           -> part_base + l * M * K  (summed by reduce_rows)
// table rows {W, bias, dW, db, N, off} (int64); blockmap {layer, tile x, tile y, -}.
template <int MODE>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(const long* __restrict__ table,
                                                           const int4* __restrict__ blockmap, const float* emb,
                                                           float* base, int M, int K) {
  const int4 bm = blockmap[blockIdx.x];
  const long* e = table + 6L * bm.x;
  const float* W = (const float*)e[0];
  const float* bias = (const float*)e[1];
  float* dW = (float*)e[2];
  float* db = (float*)e[3];
  const int N = (int)e[4];
  float* slab = base + e[5];
  float* part = base + (long)bm.w;
  GemmP p;
  if (MODE == 0) {
    p.A = emb; p.B = W; p.bias = bias; p.C = slab; p.pre = nullptr; p.rowsum = nullptr;
    p.M = M; p.N = N; p.K = K; p.a_rs = K; p.a_ks = 1; p.b_rs = K; p.b_ks = 1;
  } else if (MODE == 1) {
    p.A = slab; p.B = emb; p.bias = nullptr; p.C = dW; p.pre = nullptr; p.rowsum = db;
    p.M = N; p.N = K; p.K = M; p.a_rs = 1; p.a_ks = N; p.b_rs = 1; p.b_ks = K;
  } else {
    p.A = slab; p.B = W; p.bias = nullptr; p.C = part; p.pre = nullptr; p.rowsum = nullptr;
    p.M = M; p.N = K; p.K = N; p.a_rs = N; p.a_ks = 1; p.b_rs = 1; p.b_ks = K;
  }
  p.act = 0;
  gemm_tile<4>(p, vec_ok_dev(p.A, p.a_rs, p.a_ks), vec_ok_dev(p.B, p.b_rs, p.b_ks), bm.y, bm.z);
}

static bool vec_ok(const float* ptr, long rs, long ks) {
  return ks == 1 && (rs % 4) == 0 && (((uintptr_t)ptr) & 15) == 0;
}

static void launch_gemm(const GemmP& p, hipStream_t s) {
  dim3 grid(i2r_cdiv(p.N, 32), i2r_cdiv(p.M, 32));
  // few output tiles and a long contraction (batch-row products against a whole weight matrix): 16 wavefronts split
  // the contraction so every operand is requested in one round trip; otherwise 4
  if ((long)grid.x * grid.y <= 128 && p.K >= 256)
    hipLaunchKernelGGL(small_gemm_kernel<16>, grid, dim3(1024), 0, s, p, (int)vec_ok(p.A, p.a_rs, p.a_ks),
                       (int)vec_ok(p.B, p.b_rs, p.b_ks));
  else
    hipLaunchKernelGGL(small_gemm_kernel<4>, grid, dim3(256), 0, s, p, (int)vec_ok(p.A, p.a_rs, p.a_ks),
                       (int)vec_ok(p.B, p.b_rs, p.b_ks));
}

extern "C" int i2r_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, float* pre, int M,
                                  int K, int N, int act_silu, i2r_stream_t stream) {
  I2R_REQUIRE(x && w && y && M > 0 && K > 0 && N > 0, "linear_fwd: bad argument");
  GemmP p{x, w, bias, y, pre, nullptr, M, N, K, K, 1, K, 1, act_silu};
  launch_gemm(p, (hipStream_t)stream);
  return i2r_check_launch("linear_fwd");
}

// dx (M,K) = dy (M,N) . W (N,K);   dW (N,K) = dy^T . x;   db (N) = column sums of dy   (each optional)
extern "C" int i2r_linear_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                  int M, int K, int N, i2r_stream_t stream) {
  I2R_REQUIRE(x && w && dy && M > 0 && K > 0 && N > 0, "linear_bwd: bad argument");
  I2R_REQUIRE(!db || dw, "linear_bwd: the bias gradient is produced together with the weight gradient");
  if (dx) {
    GemmP p{dy, w, nullptr, dx, nullptr, nullptr, M, K, N, N, 1, 1, K, 0};
    launch_gemm(p, (hipStream_t)stream);
  }
  if (dw) {
    GemmP p{dy, x, nullptr, dw, nullptr, db, N, K, M, 1, N, 1, K, 0};
    launch_gemm(p, (hipStream_t)stream);
  }
  return i2r_check_launch("linear_bwd");
}

// Grouped form over the network's `affine` layers; see gemm_grouped_kernel.  `emb` (M, K); `base` = the output slab
// (mode 0) or the dy slab (modes 1, 2; mode 2 writes its partials behind it at the float offsets the blockmap carries).
extern "C" int i2r_gemm_grouped_f32(const void* table, const void* blockmap, int nblocks, int mode, const float* emb,
                                    float* base, int M, int K, i2r_stream_t stream) {
  I2R_REQUIRE(table && blockmap && nblocks > 0 && emb && base && M > 0 && K > 0 && mode >= 0 && mode <= 2,
              "gemm_grouped: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const long* t = (const long*)table;
  const int4* bmap = (const int4*)blockmap;
  if (mode == 0) hipLaunchKernelGGL(gemm_grouped_kernel<0>, dim3(nblocks), dim3(256), 0, s, t, bmap, emb, base, M, K);
  else if (mode == 1) hipLaunchKernelGGL(gemm_grouped_kernel<1>, dim3(nblocks), dim3(256), 0, s, t, bmap, emb, base, M, K);
  else hipLaunchKernelGGL(gemm_grouped_kernel<2>, dim3(nblocks), dim3(256), 0, s, t, bmap, emb, base, M, K);
  return i2r_check_launch("gemm_grouped");
}
