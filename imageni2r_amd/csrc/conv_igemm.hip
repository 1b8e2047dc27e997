// Implicit-GEMM convolutions (3x3 pad 1 / 1x1, stride 1, NCHW fp32) on the gfx950 fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD).  Replaces torch.nn.functional.conv2d as called
// from models/networks.py:87 (forward) and autograd's conv backward (dgrad, wgrad).
//
// GEMM view (forward / dgrad):  Out[co][n] = sum_k  W[co][k] * Im2col[k][n]
//     k = (ci, r, s) in the weight's own OIHW order (no repack), n = (b, y, x) flattened positions.
//     MFMA rows  (i, accumulator registers) = output channel, MFMA columns (j, lanes) = position, so each accumulator
//     register is stored as 32 consecutive positions of one output plane: 128-B coalesced NCHW stores.
// wgrad:                        dW[co][j]  = sum_n  dY[co][n] * Im2col[j][n]      (reduction over positions, split over
//     blockIdx.z with fp32 atomics when one pass would leave CUs idle).
//
// Tiling: 256 threads = 4 waves in a WM x WN grid, each wave TM x TN MFMA tiles of 32x32; K is walked in chunks of 16
// staged through LDS as [k][m] / [k][n] (m, n contiguous -> conflict-free ds_read_b32 for the one-float-per-lane f32
// MFMA operands), register-staged double buffering (global loads of chunk c+1 in flight under the MFMAs of chunk c).
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

#define BK 16

struct ConvParams {
  const float* x;     // input activations (B, Cin, H, W)
  const float* w;     // weights (Cout, Cin*KS*KS)
  const float* bias;  // (Cout) or null
  const float* res;   // residual (B, Cout, H, W) or null
  float* out;         // (B, Cout, H, W)
  int B, Cin, H, W, Cout, HW, K;
  long Npos;
};

template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void mma_chunk(const float* __restrict__ As, const float* __restrict__ Bs, int lda,
                                          int ldb, f32x16 (&acc)[TM][TN], int wm, int wn, int l31, int h) {
#pragma unroll
  for (int ks = 0; ks < BK / 2; ++ks) {
    float a[TM], b[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) a[t] = As[(2 * ks + h) * lda + (wm * TM + t) * 32 + l31];
#pragma unroll
    for (int t = 0; t < TN; ++t) b[t] = Bs[(2 * ks + h) * ldb + (wn * TN + t) * 32 + l31];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------ forward / dgrad
template <int WM, int WN, int TM, int TN, int KS>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int EA = BM * BK / 256;  // A elements per thread per chunk
  constexpr int EB = BN * BK / 256;  // B elements per thread per chunk
  constexpr int KSTEP = 256 / BN;    // k rows covered by one pass of the 256 threads over the B tile
  __shared__ float smem[2 * BK * LDA + 2 * BK * LDB];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM;
  const long n0 = (long)blockIdx.x * BN;

  // A loader: kk = tid % 16 (k fastest: 64-B runs of a weight row), rows tid/16 + 16*j
  const int a_kk = tid & 15, a_i0 = tid >> 4;
  // B loader: one position per thread for the whole kernel
  const int b_j = tid % BN, b_kk0 = tid / BN;
  const long nj = n0 + b_j;
  const bool nvalid = nj < p.Npos;
  int pb = 0, py = 0, px = 0;
  if (nvalid) {
    pb = (int)(nj / p.HW);
    int pos = (int)(nj % p.HW);
    py = pos / p.W;
    px = pos % p.W;
  }
  const float* xb = p.x + (long)pb * p.Cin * p.HW;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float ra[EA], rb[EB];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int j = 0; j < EA; ++j) {
      int row = m0 + a_i0 + 16 * j, k = k0 + a_kk;
      ra[j] = (row < p.Cout && k < p.K) ? p.w[(long)row * p.K + k] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < EB; ++j) {
      int k = k0 + b_kk0 + KSTEP * j;
      float v = 0.0f;
      if (KS == 3) {
        int ci = k / 9, tap = k - 9 * ci;
        int r = tap / 3, s = tap - 3 * r;
        int iy = py + r - 1, ix = px + s - 1;
        if (nvalid && k < p.K && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
          v = xb[(long)ci * p.HW + iy * p.W + ix];
      } else {
        if (nvalid && k < p.K) v = xb[(long)k * p.HW + py * p.W + px];
      }
      rb[j] = v;
    }
  };
  auto store_chunk = [&](int buf) {
    float* a = As + buf * BK * LDA;
    float* b = Bs + buf * BK * LDB;
#pragma unroll
    for (int j = 0; j < EA; ++j) a[a_kk * LDA + a_i0 + 16 * j] = ra[j];
#pragma unroll
    for (int j = 0; j < EB; ++j) b[(b_kk0 + KSTEP * j) * LDB + b_j] = rb[j];
  };

  const int nchunks = (p.K + BK - 1) / BK;
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    int cur = c & 1;
    if (c + 1 < nchunks) load_chunk((c + 1) * BK);
    mma_chunk<WM, WN, TM, TN>(As + cur * BK * LDA, Bs + cur * BK * LDB, LDA, LDB, acc, wm, wn, l31, h);
    if (c + 1 < nchunks) store_chunk(cur ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane&31 (position), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (output channel)
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    long n = n0 + (wn * TN + tn) * 32 + l31;
    if (n >= p.Npos) continue;
    int b = (int)(n / p.HW);
    int pos = (int)(n % p.HW);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < p.Cout) {
          long o = ((long)b * p.Cout + co) * p.HW + pos;
          float v = acc[tm][tn][r];
          if (p.bias) v += p.bias[co];
          if (p.res) v += p.res[o];
          p.out[o] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ wgrad
struct WgradParams {
  const float* x;   // forward input (B, Cin, H, W)
  const float* dy;  // output gradient (B, Cout, H, W)
  float* dw;        // (Cout, Cin*KS*KS); zero-filled by the caller when nsplit > 1
  int B, Cin, H, W, Cout, HW, K;
  long Npos;
  int chunks_per_split, nsplit;
};

template <int WM, int WN, int TM, int TN, int KS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int EA = BM * BK / 256, EB = BN * BK / 256;
  __shared__ float smem[2 * BK * LDA + 2 * BK * LDB];
  float* As = smem;
  float* Bs = smem + 2 * BK * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM;  // output channel tile
  const int j0 = blockIdx.x * BN;  // (ci, tap) tile
  const int kk = tid & 15, i0 = tid >> 4;

  // per-thread constant decode of its EB im2col columns
  int coff[EB], dr[EB], ds[EB];
  bool jok[EB];
#pragma unroll
  for (int e = 0; e < EB; ++e) {
    int j = j0 + i0 + 16 * e;
    jok[e] = j < p.K;
    int ci = (KS == 3) ? j / 9 : j;
    int tap = (KS == 3) ? j - 9 * ci : 0;
    int r = tap / 3, s = tap - 3 * r;
    coff[e] = ci * p.HW;
    dr[e] = (KS == 3) ? r - 1 : 0;
    ds[e] = (KS == 3) ? s - 1 : 0;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float ra[EA], rb[EB];
  auto load_chunk = [&](long nbase) {
    long n = nbase + kk;
    bool nv = n < p.Npos;
    int b = 0, y = 0, x = 0, pos = 0;
    if (nv) {
      b = (int)(n / p.HW);
      pos = (int)(n % p.HW);
      y = pos / p.W;
      x = pos % p.W;
    }
    const float* dyb = p.dy + (long)b * p.Cout * p.HW + pos;
    const float* xb = p.x + (long)b * p.Cin * p.HW;
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      int co = m0 + i0 + 16 * e;
      ra[e] = (nv && co < p.Cout) ? dyb[(long)co * p.HW] : 0.0f;
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      int iy = y + dr[e], ix = x + ds[e];
      float v = 0.0f;
      if (nv && jok[e] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v = xb[coff[e] + iy * p.W + ix];
      rb[e] = v;
    }
  };
  auto store_chunk = [&](int buf) {
    float* a = As + buf * BK * LDA;
    float* b = Bs + buf * BK * LDB;
#pragma unroll
    for (int e = 0; e < EA; ++e) a[kk * LDA + i0 + 16 * e] = ra[e];
#pragma unroll
    for (int e = 0; e < EB; ++e) b[kk * LDB + i0 + 16 * e] = rb[e];
  };

  const long total_chunks = (p.Npos + BK - 1) / BK;
  long c_begin = (long)blockIdx.z * p.chunks_per_split;
  long c_end = c_begin + p.chunks_per_split;
  if (c_end > total_chunks) c_end = total_chunks;
  if (c_begin < c_end) {
    load_chunk(c_begin * BK);
    store_chunk(0);
    __syncthreads();
    for (long c = c_begin; c < c_end; ++c) {
      int cur = (int)((c - c_begin) & 1);
      if (c + 1 < c_end) load_chunk((c + 1) * BK);
      mma_chunk<WM, WN, TM, TN>(As + cur * BK * LDA, Bs + cur * BK * LDB, LDA, LDB, acc, wm, wn, l31, h);
      if (c + 1 < c_end) store_chunk(cur ^ 1);
      __syncthreads();
    }
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int j = j0 + (wn * TN + tn) * 32 + l31;
    if (j >= p.K) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < p.Cout) {
          float* dst = p.dw + (long)co * p.K + j;
          if (p.nsplit > 1)
            atomicAdd(dst, acc[tm][tn][r]);
          else
            *dst = acc[tm][tn][r];
        }
      }
    }
  }
}

// wt[ci][co][8 - tap] = w[co][ci][tap]  (KS = 3);  wt[ci][co] = w[co][ci]  (KS = 1): dgrad == forward conv of dY with wt
__global__ void weight_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int KK) {
  long total = (long)Cout * Cin * KK;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int tap = (int)(i % KK);
    long r = i / KK;
    int co = (int)(r % Cout);
    int ci = (int)(r / Cout);
    wt[i] = w[((long)co * Cin + ci) * KK + (KK - 1 - tap)];
  }
}

// db[co] = sum_{b,pos} dy[b][co][pos]: one wave per channel
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int B,
                                                        int C, int HW) {
  int lane = threadIdx.x & 63;
  int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  float acc = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* p = dy + ((long)b * C + c) * HW;
    for (int i = lane; i < HW; i += 64) acc += p[i];
  }
  acc = wave_sum(acc);
  if (lane == 0) db[c] = acc;
}

// ------------------------------------------------------------------------------------------------ host side
template <int WM, int WN, int TM, int TN>
static void launch_fwd(const ConvParams& p, int ks, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  dim3 grid(i2r_cdiv(p.Npos, BN), i2r_cdiv(p.Cout, BM));
  if (ks == 3)
    hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, TM, TN, 3>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, TM, TN, 1>), grid, dim3(256), 0, s, p);
}

static int conv_fwd_dispatch(const ConvParams& p, int ks, hipStream_t s) {
  long t128 = (long)i2r_cdiv(p.Npos, 128) * i2r_cdiv(p.Cout, 128);
  if (p.Cout <= 32)
    launch_fwd<1, 4, 1, 1>(p, ks, s);  // 32 x 128
  else if (p.Npos <= 32)
    launch_fwd<4, 1, 1, 1>(p, ks, s);  // 128 x 32
  else if (t128 >= 192 && p.Cout > 64)
    launch_fwd<2, 2, 2, 2>(p, ks, s);  // 128 x 128
  else
    launch_fwd<2, 2, 1, 1>(p, ks, s);  // 64 x 64
  return i2r_check_launch("conv2d");
}

extern "C" int i2r_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* residual,
                                  float* out, int B, int Cin, int H, int W, int Cout, int ksize,
                                  i2r_stream_t stream) {
  I2R_REQUIRE(x && w && out, "conv2d_fwd: null pointer");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_fwd: kernel size %d unsupported (1 or 3)", ksize);
  I2R_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d_fwd: non-positive dimension");
  I2R_REQUIRE((long)Cin * ksize * ksize < (1L << 30) && (long)Cin * H * W < (1L << 31),
              "conv2d_fwd: per-sample tensor too large for 32-bit plane offsets");
  ConvParams p{x, w, bias, residual, out, B, Cin, H, W, Cout, H * W, Cin * ksize * ksize, (long)B * H * W};
  return conv_fwd_dispatch(p, ksize, (hipStream_t)stream);
}

extern "C" int i2r_conv2d_weight_flip_f32(const float* w, float* wt, int Cout, int Cin, int ksize,
                                          i2r_stream_t stream) {
  I2R_REQUIRE(w && wt && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "weight_flip: bad argument");
  long total = (long)Cout * Cin * ksize * ksize;
  int nb = i2r_cdiv(total, 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(weight_flip_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wt, Cout, Cin,
                     ksize * ksize);
  return i2r_check_launch("weight_flip");
}

// dx = conv(dy, flip(w)); `wt_ws` is a caller-provided workspace of Cout*Cin*ksize^2 floats.
extern "C" int i2r_conv2d_dgrad_f32(const float* dy, const float* w, float* wt_ws, float* dx, int B, int Cin, int H,
                                    int W, int Cout, int ksize, i2r_stream_t stream) {
  I2R_REQUIRE(dy && w && wt_ws && dx, "conv2d_dgrad: null pointer");
  int e = i2r_conv2d_weight_flip_f32(w, wt_ws, Cout, Cin, ksize, stream);
  if (e) return e;
  return i2r_conv2d_fwd_f32(dy, wt_ws, nullptr, nullptr, dx, B, Cout, H, W, Cin, ksize, stream);
}

template <int WM, int WN, int TM, int TN>
static void launch_wgrad(WgradParams& p, int ks, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  long tiles = (long)i2r_cdiv(p.K, BN) * i2r_cdiv(p.Cout, BM);
  long total_chunks = (p.Npos + BK - 1) / BK;
  long want = (1024 + tiles - 1) / tiles;  // aim for ~4 blocks per CU
  if (want > total_chunks / 4) want = total_chunks / 4;  // keep >= 4 chunks per split
  if (want < 1) want = 1;
  p.chunks_per_split = (int)((total_chunks + want - 1) / want);
  p.nsplit = (int)((total_chunks + p.chunks_per_split - 1) / p.chunks_per_split);
  if (p.nsplit > 1) (void)hipMemsetAsync(p.dw, 0, sizeof(float) * (size_t)p.Cout * p.K, s);
  dim3 grid(i2r_cdiv(p.K, BN), i2r_cdiv(p.Cout, BM), p.nsplit);
  if (ks == 3)
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 3>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 1>), grid, dim3(256), 0, s, p);
}

extern "C" int i2r_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int H,
                                    int W, int Cout, int ksize, i2r_stream_t stream) {
  I2R_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_wgrad: kernel size %d unsupported (1 or 3)", ksize);
  I2R_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d_wgrad: non-positive dimension");
  WgradParams p{x, dy, dw, B, Cin, H, W, Cout, H * W, Cin * ksize * ksize, (long)B * H * W, 0, 1};
  hipStream_t s = (hipStream_t)stream;
  if (Cout <= 32)
    launch_wgrad<1, 4, 1, 1>(p, ksize, s);
  else if (p.K <= 32)
    launch_wgrad<4, 1, 1, 1>(p, ksize, s);
  else if ((long)i2r_cdiv(p.K, 128) * i2r_cdiv(Cout, 128) >= 8 && Cout > 64)
    launch_wgrad<2, 2, 2, 2>(p, ksize, s);
  else
    launch_wgrad<2, 2, 1, 1>(p, ksize, s);
  if (dbias)
    hipLaunchKernelGGL(bias_grad_kernel, dim3(i2r_cdiv(Cout, 4)), dim3(256), 0, s, dy, dbias, B, Cout, H * W);
  return i2r_check_launch("conv2d_wgrad");
}
