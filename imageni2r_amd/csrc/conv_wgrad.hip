// Weight gradient of the 3x3 / 1x1 convolutions (autograd of torch.nn.functional.conv2d, models/networks.py:87) on
// the fp32 MFMA:   dW[co][j] = sum_n dY[co][n] * Im2col[j][n],  j = (ci, r, s) in OIHW order, n = (b, y, x).
// The reduction runs over positions; it is split over blockIdx.z (fp32 atomics into a zeroed dW) when a single pass
// would leave CUs idle.  The bias gradient (row sums of dY) falls out of the dY tiles the first column of workgroups
// stages anyway.  `accumulate` adds into dW/db instead of overwriting them (the caller's pre-zeroed gradient buffer).
#include <cstdlib>
#include "conv_common.h"
#include "../../include/i2r_hip.h"

#define BK 16

// ------------------------------------------------------------------------------------------------ wgrad
struct WgradParams {
  const float* x;   // forward input (B, Cin, H, W)
  const float* dy;  // output gradient (B, Cout, H, W)
  float* dw;        // (Cout, Cin*KS*KS); zero-filled (or holding a partial sum) when `atomic`
  float* db;        // (Cout) bias gradient = row sums of dY, produced by the blockIdx.x == 0 column of blocks; or null
  int B, Cin, H, W, Cout, HW, K;
  long Npos;
  int chunks_per_split, nsplit;
  int atomic;       // 1: add into dw/db with fp32 atomics (split reduction, or accumulate into existing values)
  long slab_stride; // > 0 (deterministic mode): split z stores its partial dW, then its partial db, at dw + z * slab_stride
                    // (a workspace; plain stores, no atomics) and wgrad_slab_reduce_kernel sums the slabs in z order
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

  // per-thread constants: its EB im2col columns (channel plane, tap shift) and its EA output-channel rows
  int coff[EB], dr[EB], ds[EB], doff[EB];
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
    doff[e] = dr[e] * p.W + ds[e];
  }
  int arow[EA];
  bool aok[EA];
#pragma unroll
  for (int e = 0; e < EA; ++e) {
    int co = m0 + i0 + 16 * e;
    aok[e] = co < p.Cout;
    arow[e] = aok[e] ? co * p.HW : 0;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float ra[EA], rb[EB];
  float bsum[EA];
#pragma unroll
  for (int e = 0; e < EA; ++e) bsum[e] = 0.0f;
  const bool want_bias = p.db != nullptr && blockIdx.x == 0;
  float* const dw_out = p.dw + (long)blockIdx.z * p.slab_stride;
  float* const db_out = p.slab_stride > 0 ? dw_out + (long)p.Cout * p.K : p.db;
  // Position of this thread's slot kk in the chunk being loaded, kept incrementally (one division per workgroup
  // instead of four per chunk: the loader, not the MFMA, was the busiest unit of this kernel).
  const long total_chunks = (p.Npos + BK - 1) / BK;
  long c_begin = (long)blockIdx.z * p.chunks_per_split;
  long c_end = c_begin + p.chunks_per_split;
  if (c_end > total_chunks) c_end = total_chunks;
  int pb, ppos;
  {
    long n = c_begin * BK + kk;
    pb = (int)(n / p.HW);
    ppos = (int)(n - (long)pb * p.HW);
  }
  const int wsh = (p.W & (p.W - 1)) == 0 ? __builtin_ctz(p.W) : -1;
  auto load_chunk = [&]() {  // loads the chunk at (pb, ppos), then advances them by one chunk
    const bool nv = pb < p.B;
    const int y = wsh >= 0 ? (ppos >> wsh) : ppos / p.W;
    const int x = ppos - y * p.W;
    const float* dyb = p.dy + (long)pb * p.Cout * p.HW + ppos;
    const float* xb = p.x + (long)pb * p.Cin * p.HW + ppos;
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      ra[e] = (nv && aok[e]) ? dyb[arow[e]] : 0.0f;
      bsum[e] += ra[e];
    }
#pragma unroll
    for (int e = 0; e < EB; ++e) {
      int iy = y + dr[e], ix = x + ds[e];
      float v = 0.0f;
      if (nv && jok[e] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) v = xb[coff[e] + doff[e]];
      rb[e] = v;
    }
    ppos += BK;
    while (ppos >= p.HW) { ppos -= p.HW; ++pb; }
  };
  auto store_chunk = [&](int buf) {
    float* a = As + buf * BK * LDA;
    float* b = Bs + buf * BK * LDB;
#pragma unroll
    for (int e = 0; e < EA; ++e) a[kk * LDA + i0 + 16 * e] = ra[e];
#pragma unroll
    for (int e = 0; e < EB; ++e) b[kk * LDB + i0 + 16 * e] = rb[e];
  };

  if (c_begin < c_end) {
    load_chunk();
    store_chunk(0);
    __syncthreads();
    for (long c = c_begin; c < c_end; ++c) {
      int cur = (int)((c - c_begin) & 1);
      if (c + 1 < c_end) load_chunk();
      mma_chunk<WM, WN, TM, TN, BK>(As + cur * BK * LDA, Bs + cur * BK * LDB, LDA, LDB, acc, wm, wn, l31, h);
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
          float* dst = dw_out + (long)co * p.K + j;
          if (p.atomic)
            atomicAdd(dst, acc[tm][tn][r]);
          else
            *dst = acc[tm][tn][r];
        }
      }
    }
  }
  if (want_bias) {  // the 16 threads that share i0 hold the 16 position phases of one output channel
#pragma unroll
    for (int e = 0; e < EA; ++e) {
      float v = bsum[e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      int co = m0 + i0 + 16 * e;
      if (kk == 0 && co < p.Cout) {
        if (p.atomic)
          atomicAdd(db_out + co, v);
        else
          db_out[co] = v;
      }
    }
  }
}

// Deterministic mode: out[i] (+)= slab[0][i] + slab[1][i] + ... in that fixed order; n = Cout*K (+ Cout for the bias
// gradient, which sits behind dW in every slab and goes to its own output).
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ slabs, long slab_stride,
                                                                int nsplit, float* __restrict__ dw,
                                                                float* __restrict__ db, long n_dw, long n_all,
                                                                int accumulate) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_all; i += (long)gridDim.x * blockDim.x) {
    float v = slabs[i];
    for (int z = 1; z < nsplit; ++z) v += slabs[z * slab_stride + i];
    float* dst = i < n_dw ? dw + i : db + (i - n_dw);
    *dst = accumulate ? *dst + v : v;
  }
}

// ------------------------------------------------------------------------------------------------ host side
static void wgrad_split_plan(long tiles, long Npos, int& chunks_per_split, int& nsplit) {
  long total_chunks = (Npos + BK - 1) / BK;
  static const long target = getenv("I2R_WGRAD_TARGET") ? atol(getenv("I2R_WGRAD_TARGET")) : 1024;
  long want = (target + tiles - 1) / tiles;  // aim for ~4 blocks per CU
  if (want > total_chunks / 4) want = total_chunks / 4;  // keep >= 4 chunks per split
  static const long maxsplit = getenv("I2R_WGRAD_MAXSPLIT") ? atol(getenv("I2R_WGRAD_MAXSPLIT")) : 1 << 20;
  if (want > maxsplit) want = maxsplit;
  if (want < 1) want = 1;
  chunks_per_split = (int)((total_chunks + want - 1) / want);
  nsplit = (int)((total_chunks + chunks_per_split - 1) / chunks_per_split);
}

template <int WM, int WN, int TM, int TN>
static void launch_wgrad_det(WgradParams& p, int ks, float* ws, float* dw, float* db, int accumulate, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  wgrad_split_plan((long)i2r_cdiv(p.K, BN) * i2r_cdiv(p.Cout, BM), p.Npos, p.chunks_per_split, p.nsplit);
  const long n_dw = (long)p.Cout * p.K, n_all = n_dw + (db ? p.Cout : 0);
  p.dw = ws;                       // every split writes its own slab with plain stores
  p.db = db ? ws : nullptr;        // (non-null = "bias wanted"; the kernel derives the slab's db address itself)
  p.atomic = 0;
  p.slab_stride = n_dw + p.Cout;
  dim3 grid(i2r_cdiv(p.K, BN), i2r_cdiv(p.Cout, BM), p.nsplit);
  if (ks == 3)
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 3>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 1>), grid, dim3(256), 0, s, p);
  int nb = i2r_cdiv(n_all, 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(nb), dim3(256), 0, s, ws, p.slab_stride, p.nsplit, dw, db, n_dw,
                     n_all, accumulate);
}

template <int WM, int WN, int TM, int TN>
static long wgrad_det_floats(int K, int Cout, long Npos) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  int cps, ns;
  wgrad_split_plan((long)i2r_cdiv(K, BN) * i2r_cdiv(Cout, BM), Npos, cps, ns);
  return (long)ns * ((long)Cout * K + Cout);
}

template <int WM, int WN, int TM, int TN>
static void launch_wgrad(WgradParams& p, int ks, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  wgrad_split_plan((long)i2r_cdiv(p.K, BN) * i2r_cdiv(p.Cout, BM), p.Npos, p.chunks_per_split, p.nsplit);
  if (p.nsplit > 1 && !p.atomic) {  // split reduction into fresh outputs: zero them, then add atomically
    (void)hipMemsetAsync(p.dw, 0, sizeof(float) * (size_t)p.Cout * p.K, s);
    if (p.db) (void)hipMemsetAsync(p.db, 0, sizeof(float) * (size_t)p.Cout, s);
    p.atomic = 1;
  }
  dim3 grid(i2r_cdiv(p.K, BN), i2r_cdiv(p.Cout, BM), p.nsplit);
  if (ks == 3)
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 3>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, TM, TN, 1>), grid, dim3(256), 0, s, p);
}

// 128x128 tiles only pay once the position range is long: below that the split reduction's atomic epilogue
// (BM*BN atomics per block) outweighs the operand reuse, and 64x64 tiles win (B1, batch 32: 19.1 -> 17.7 ms/step)
static long wgrad_big_pos() {
  static const long v = getenv("I2R_WGRAD_BIGPOS") ? atol(getenv("I2R_WGRAD_BIGPOS")) : 32768;
  return v;
}

extern "C" int i2r_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int H,
                                    int W, int Cout, int ksize, int accumulate, i2r_stream_t stream) {
  I2R_REQUIRE(x && dy && dw, "conv2d_wgrad: null pointer");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_wgrad: kernel size %d unsupported (1 or 3)", ksize);
  I2R_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d_wgrad: non-positive dimension");
  WgradParams p{x, dy, dw, dbias, B, Cin, H, W, Cout, H * W, Cin * ksize * ksize, (long)B * H * W, 0, 1,
                accumulate ? 1 : 0, 0};
  hipStream_t s = (hipStream_t)stream;
  if (Cout <= 32)
    launch_wgrad<1, 4, 1, 1>(p, ksize, s);
  else if (p.K <= 32)
    launch_wgrad<4, 1, 1, 1>(p, ksize, s);
  else if ((long)i2r_cdiv(p.K, 128) * i2r_cdiv(Cout, 128) >= 8 && Cout > 64 && p.Npos >= wgrad_big_pos())
    launch_wgrad<2, 2, 2, 2>(p, ksize, s);
  else
    launch_wgrad<2, 2, 1, 1>(p, ksize, s);
  return i2r_check_launch("conv2d_wgrad");
}

// Which tile shape a problem gets (shared by the two deterministic entry points so that the workspace query and the
// launch agree).
static int wgrad_variant(int K, int Cout, long Npos) {
  if (Cout <= 32) return 0;
  if (K <= 32) return 1;
  if ((long)i2r_cdiv(K, 128) * i2r_cdiv(Cout, 128) >= 8 && Cout > 64 && Npos >= wgrad_big_pos()) return 2;
  return 3;
}

extern "C" long i2r_conv2d_wgrad_det_floats(int B, int Cin, int H, int W, int Cout, int ksize) {
  const int K = Cin * ksize * ksize;
  const long Npos = (long)B * H * W;
  switch (wgrad_variant(K, Cout, Npos)) {
    case 0: return wgrad_det_floats<1, 4, 1, 1>(K, Cout, Npos);
    case 1: return wgrad_det_floats<4, 1, 1, 1>(K, Cout, Npos);
    case 2: return wgrad_det_floats<2, 2, 2, 2>(K, Cout, Npos);
    default: return wgrad_det_floats<2, 2, 1, 1>(K, Cout, Npos);
  }
}

extern "C" int i2r_conv2d_wgrad_det_f32(const float* x, const float* dy, float* dw, float* dbias, float* ws,
                                        long ws_floats, int B, int Cin, int H, int W, int Cout, int ksize,
                                        int accumulate, i2r_stream_t stream) {
  I2R_REQUIRE(x && dy && dw && ws, "conv2d_wgrad_det: null pointer");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_wgrad_det: kernel size %d unsupported (1 or 3)", ksize);
  I2R_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d_wgrad_det: non-positive dimension");
  const long need = i2r_conv2d_wgrad_det_floats(B, Cin, H, W, Cout, ksize);
  I2R_REQUIRE(ws_floats >= need, "conv2d_wgrad_det: workspace of %ld floats, needs %ld", ws_floats, need);
  WgradParams p{x, dy, dw, dbias, B, Cin, H, W, Cout, H * W, Cin * ksize * ksize, (long)B * H * W, 0, 1, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  switch (wgrad_variant(p.K, Cout, p.Npos)) {
    case 0: launch_wgrad_det<1, 4, 1, 1>(p, ksize, ws, dw, dbias, accumulate, s); break;
    case 1: launch_wgrad_det<4, 1, 1, 1>(p, ksize, ws, dw, dbias, accumulate, s); break;
    case 2: launch_wgrad_det<2, 2, 2, 2>(p, ksize, ws, dw, dbias, accumulate, s); break;
    default: launch_wgrad_det<2, 2, 1, 1>(p, ksize, ws, dw, dbias, accumulate, s); break;
  }
  return i2r_check_launch("conv2d_wgrad_det");
}
