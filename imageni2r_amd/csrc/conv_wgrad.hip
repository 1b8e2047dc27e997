// Weight gradient of the 3x3 / 1x1 convolutions (autograd of torch.nn.functional.conv2d, models/networks.py:87) on
// the fp32 MFMA:   dW[co][j] = sum_n dY[co][n] * Im2col[j][n],  j = (ci, r, s) in OIHW order, n = (b, y, x).
// The reduction runs over positions; it is split over blockIdx.z (fp32 atomics into a zeroed dW) when a single pass
// would leave CUs idle.  The bias gradient (row sums of dY) falls out of the dY tiles the first column of workgroups
// stages anyway.  `accumulate` adds into dW/db instead of overwriting them (the caller's pre-zeroed gradient buffer).
#include <cmath>
#include <cstdint>
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

// ------------------------------------------------------------------------------------------------ 3x3 from halo tiles
// The kernel above gathers every im2col column with its own scalar loads and bound checks, nine times over for the nine
// taps of a channel, and does one MFMA per wave between them: its loader is the bound (~40 TF).  Here a wave stages, per
// chunk of 64 positions (R = 64 / W whole image rows of one sample), the input rows with their zero halo ONCE --
// position-major in LDS, 32 channels wide -- and the 32 dY rows beside them, and the nine taps are nine shifted reads
// of that one tile: 288 MFMAs per chunk per wave against 24 16-byte loads per lane.  The four waves of a workgroup own
// the SAME 32 (co) x 32 (ci) x 9 (taps) output tile and take every fourth chunk of the workgroup's position range, each
// out of its own LDS tiles (no workgroup barrier in the loop); their accumulators are summed in wave order through LDS
// at the end, so a workgroup adds one tile -- in the (ci, tap) order of dW's rows, 256 consecutive bytes per store --
// where four workgroups of the gather kernel would each add theirs.
template <int W_>
struct WgTile {
  static constexpr int R = 64 / W_;                // image rows per chunk
  static constexpr int XR = (R + 2) * (W_ + 2);    // positions of the halo tile
  static constexpr int LD = 33;                    // floats per position (32 channels + 1: conflict-free b32 reads)
  static constexpr int XF = XR * LD, YF = 64 * LD;
  static constexpr int WAVE_F = XF + YF;           // floats of LDS per wave
  static constexpr int NX = 32 * (R + 2) * (W_ / 4) / 64;  // 16-byte loads per lane and chunk: input rows
  static constexpr int NY = 8;                              //                                   dY rows
  static constexpr int LDS_BYTES = 4 * WAVE_F * 4;
  static_assert(4 * WAVE_F >= 4 * 16 * 288 + 4 * 32, "the reduction space lives in the operand space");
};

template <int W_>
__global__ __launch_bounds__(256) void conv_wgrad3_tile_kernel(WgradParams p) {
  using T = WgTile<W_>;
  constexpr int R = T::R, LD = T::LD, W2 = W_ + 2, Q = W_ / 4, NX = T::NX, NY = T::NY;
  extern __shared__ float wg_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  float* const Xs = wg_smem + wave * T::WAVE_F;
  float* const Ys = Xs + T::XF;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int cpi = p.H / R;  // chunks per image
  const long total = (long)p.B * cpi;
  const long c_begin = (long)blockIdx.z * p.chunks_per_split;
  long c_end = c_begin + p.chunks_per_split;
  if (c_end > total) c_end = total;
  float* const dw_out = p.dw + (long)blockIdx.z * p.slab_stride;
  float* const db_out = p.slab_stride > 0 ? dw_out + (long)p.Cout * p.K : p.db;
  const bool want_bias = p.db != nullptr && blockIdx.x == 0;

  for (int i = lane; i < (R + 2) * 2 * LD; i += 64) {  // the halo columns stay zero; the rest is rewritten per chunk
    const int row = i / (2 * LD), rest = i - row * 2 * LD, side = rest / LD;
    Xs[(row * W2 + side * (W_ + 1)) * LD + (rest - side * LD)] = 0.0f;
  }

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  float bsum[NY];
#pragma unroll
  for (int e = 0; e < NY; ++e) bsum[e] = 0.0f;

  float4 rx[NX], ry[NY];
  auto load_chunk = [&](long c) {
    const int b = (int)(c / cpi);
    const int y0 = (int)(c - (long)b * cpi) * R;
    const float* xb = p.x + ((long)b * p.Cin + ci0) * p.HW;
    const float* yb = p.dy + ((long)b * p.Cout + co0) * p.HW + y0 * W_;
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int id = lane + 64 * e, q = id % Q, row = (id / Q) % (R + 2), ci = id / (Q * (R + 2));
      const int y = y0 - 1 + row;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (ci0 + ci < p.Cin && (unsigned)y < (unsigned)p.H)
        v = *reinterpret_cast<const float4*>(xb + (long)ci * p.HW + y * W_ + q * 4);
      rx[e] = v;
    }
#pragma unroll
    for (int e = 0; e < NY; ++e) {
      const int id = lane + 64 * e, q = id & 15, co = id >> 4;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (co0 + co < p.Cout) v = *reinterpret_cast<const float4*>(yb + (long)co * p.HW + q * 4);
      ry[e] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int e = 0; e < NX; ++e) {
      const int id = lane + 64 * e, q = id % Q, row = (id / Q) % (R + 2), ci = id / (Q * (R + 2));
      float* d = Xs + (row * W2 + 1 + q * 4) * LD + ci;
      d[0] = rx[e].x; d[LD] = rx[e].y; d[2 * LD] = rx[e].z; d[3 * LD] = rx[e].w;
    }
#pragma unroll
    for (int e = 0; e < NY; ++e) {
      const int id = lane + 64 * e, q = id & 15, co = id >> 4;
      float* d = Ys + (q * 4) * LD + co;
      d[0] = ry[e].x; d[LD] = ry[e].y; d[2 * LD] = ry[e].z; d[3 * LD] = ry[e].w;
      bsum[e] += (ry[e].x + ry[e].y) + (ry[e].z + ry[e].w);
    }
  };
  // lane (l31, h): A = dY[co = l31][position 2*ks + h], B = X[ci = l31][that position shifted by the tap]
  const float* const ya = Ys + h * LD + l31;
  const float* const xa = Xs + h * LD + l31;
  auto fetch = [&](int ks, float& a, float (&b)[9]) {
    const int pos = 2 * ks, r = pos / W_, x = pos % W_;
    a = ya[pos * LD];
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int ts = 0; ts < 3; ++ts) b[tr * 3 + ts] = xa[((r + tr) * W2 + x + ts) * LD];
  };

  long c = c_begin + wave;
  if (c < c_end) {
    load_chunk(c);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zero fill is behind us
    store_chunk();
  }
  while (c < c_end) {
    const long cn = c + 4;
    if (cn < c_end) load_chunk(cn);  // in flight under the MFMAs of this chunk
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float a0, b0[9], a1, b1[9];
    fetch(0, a0, b0);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
      if (ks + 1 < 32) fetch(ks + 1, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a0 = a1;
#pragma unroll
      for (int t = 0; t < 9; ++t) b0[t] = b1[t];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (cn < c_end) store_chunk();
    c = cn;
  }

  // The four partial tiles are summed in wave order (a fixed order: the result does not depend on timing), sixteen
  // output rows at a time: every wave parks those rows of its accumulators in LDS side by side, then all 256 threads
  // add the four copies of their elements and send the sums out -- consecutive threads, consecutive floats of dW.
  float* const part = wg_smem;                   // [4 waves][16 rows][288 = 32 ci x 9 taps]
  float* const bpart = wg_smem + 4 * 16 * 288;   // [4 waves][32 co]
  const int ncol = (p.Cin - ci0 < 32 ? p.Cin - ci0 : 32) * 9;
#pragma unroll 1
  for (int ph = 0; ph < 2; ++ph) {
    __syncthreads();  // operand tiles (ph 0) / the previous sixteen rows (ph 1) are read out
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = ph == 0 ? acc[t][4 * g + i] : acc[t][8 + 4 * g + i];
          part[(wave * 16 + 8 * g + i + 4 * h) * 288 + l31 * 9 + t] = v;
        }
    __syncthreads();
#pragma unroll 6
    for (int e = 0; e < 18; ++e) {
      const int idx = tid + 256 * e, row = idx / 288, col = idx - row * 288;
      const float v = ((part[idx] + part[4608 + idx]) + part[2 * 4608 + idx]) + part[3 * 4608 + idx];
      const int co = co0 + 16 * ph + row;
      if (co < p.Cout && col < ncol) {
        float* dst = dw_out + (long)co * p.K + ci0 * 9 + col;
        if (p.atomic)
          atomicAdd(dst, v);
        else
          *dst = v;
      }
    }
  }
  if (want_bias) {  // the 16 lanes that share lane / 16 hold the 16 position quads of output channel 4 * e + lane / 16
#pragma unroll
    for (int e = 0; e < NY; ++e) {
      float v = bsum[e];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if ((lane & 15) == 0) bpart[wave * 32 + 4 * e + (lane >> 4)] = v;
    }
    __syncthreads();
    if (tid < 32 && co0 + tid < p.Cout) {
      const float v = ((bpart[tid] + bpart[32 + tid]) + bpart[64 + tid]) + bpart[96 + tid];
      if (p.atomic)
        atomicAdd(db_out + co0 + tid, v);
      else
        db_out[co0 + tid] = v;
    }
  }
}

// Position split of the tile kernel: one workgroup per CU at most (its LDS), so the split that minimises
// rounds-of-256-workgroups x (chunks per wave + the epilogue's worth of about half a chunk).
static bool wgrad_tile_shape_ok(int Cin, int H, int W, int Cout, long Npos, int ks) {
  const char* sw = getenv("I2R_WGRAD_TILE");  // read per call: the A/B switch of the tests and probes
  if ((sw && atoi(sw) == 0) || ks != 3) return false;
  if (W != 8 && W != 16 && W != 32) return false;
  if (H % (64 / W) != 0 || Cin < 32 || Cout < 32) return false;
  static const long minpos = getenv("I2R_WGRAD_TILE_MINPOS") ? atol(getenv("I2R_WGRAD_TILE_MINPOS")) : 2048;
  return Npos >= minpos;
}

static bool wgrad_tile_ok(const WgradParams& p, int ks) {  // + the 16-byte loads' alignment
  return wgrad_tile_shape_ok(p.Cin, p.H, p.W, p.Cout, p.Npos, ks) &&
         (((uintptr_t)p.x | (uintptr_t)p.dy) & 15) == 0;
}

static void wgrad_tile_plan(const WgradParams& p, int& chunks_per_split, int& nsplit) {
  const long tiles = (long)i2r_cdiv(p.Cin, 32) * i2r_cdiv(p.Cout, 32);
  const long total = p.Npos / 64;
  long best_ns = 1;
  double best = 1e30;
  for (long ns = 1; ns <= total && ns <= 64; ++ns) {
    const long cps = (total + ns - 1) / ns;
    const long real_ns = (total + cps - 1) / cps;
    const long rounds = (tiles * real_ns + 255) / 256;
    const double t = (double)rounds * ((double)((cps + 3) / 4) + 0.5);
    if (t < best - 1e-9) { best = t; best_ns = real_ns; }
  }
  chunks_per_split = (int)((total + best_ns - 1) / best_ns);
  nsplit = (int)((total + chunks_per_split - 1) / chunks_per_split);
}

static bool wgrad_tile_launch(const WgradParams& p, hipStream_t s) {
  dim3 grid(i2r_cdiv(p.Cin, 32), i2r_cdiv(p.Cout, 32), p.nsplit);
#define WG_TILE(W_)                                                                                          \
  do {                                                                                                       \
    auto kern = conv_wgrad3_tile_kernel<W_>;                                                                 \
    static bool attr[64] = {false}; \
    if (!i2r_lds_attr((const void*)kern, WgTile<W_>::LDS_BYTES, attr)) return false; \
    hipLaunchKernelGGL(kern, grid, dim3(256), WgTile<W_>::LDS_BYTES, s, p);                                  \
  } while (0)
  if (p.W == 8)
    WG_TILE(8);
  else if (p.W == 16)
    WG_TILE(16);
  else
    WG_TILE(32);
#undef WG_TILE
  return true;
}

// ------------------------------------------------------------------------------------------------ 1x1, vector loader
// dW[co][ci] = sum_n dY[co][n] X[ci][n] is a plain product of two position-contiguous matrices; at the shapes of the
// path its outputs are few (49 k - 200 k) and its reduction long, so the position split that fills the chip is paid in
// atomics -- measured here at ~0.2 per ns device-wide (they execute memory-side, behind the per-XCD L2s): 128 x 256
// tiles, ideal for the matrix cores, lost to the gather kernel (0.43-0.84x) because few tiles force a deep split.  So:
// the gather kernel's 64 x 64 tile (a wave 32 x 32), but fed by 16-byte loads along the positions in chunks of 32,
// prefetched into registers under the previous chunk's MFMAs and stored position-major (odd pitch: conflict-free both
// ways) -- no per-element index arithmetic -- and a split chosen against the atomics' cost (wgrad_tile1_plan).
template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_wgrad1_tile_kernel(WgradParams p) {
  constexpr int PC = 32, LDA = BM + 1, LDB = BN + 1;
  constexpr int TM = BM / 64, TN = BN / 64;    // 32-wide tiles per wave (waves 2 x 2)
  constexpr int NA = BM * PC / 4 / 256, NB = BN * PC / 4 / 256;
  __shared__ float As[PC * LDA];
  __shared__ float Bs[PC * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ci0 = blockIdx.x * BN, co0 = blockIdx.y * BM;
  const long total = p.Npos / PC;
  const long c_begin = (long)blockIdx.z * p.chunks_per_split;
  long c_end = c_begin + p.chunks_per_split;
  if (c_end > total) c_end = total;
  float* const dw_out = p.dw + (long)blockIdx.z * p.slab_stride;
  float* const db_out = p.slab_stride > 0 ? dw_out + (long)p.Cout * p.K : p.db;
  const bool want_bias = p.db != nullptr && blockIdx.x == 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float bsum[NA];
#pragma unroll
  for (int e = 0; e < NA; ++e) bsum[e] = 0.0f;

  float4 ra[NA], rb[NB];
  auto load_chunk = [&](long c) {
    const long n0 = c * PC;
    const int b = (int)(n0 / p.HW);
    const int pos = (int)(n0 - (long)b * p.HW);  // HW % 32 == 0: a chunk never leaves its sample
    const float* yb = p.dy + ((long)b * p.Cout + co0) * p.HW + pos;
    const float* xb = p.x + ((long)b * p.Cin + ci0) * p.HW + pos;
#pragma unroll
    for (int e = 0; e < NA; ++e) {
      const int id = tid + 256 * e, q = id & 7, co = id >> 3;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (co0 + co < p.Cout) v = *reinterpret_cast<const float4*>(yb + (long)co * p.HW + q * 4);
      ra[e] = v;
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int id = tid + 256 * e, q = id & 7, ci = id >> 3;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (ci0 + ci < p.Cin) v = *reinterpret_cast<const float4*>(xb + (long)ci * p.HW + q * 4);
      rb[e] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int e = 0; e < NA; ++e) {
      const int id = tid + 256 * e, q = id & 7, co = id >> 3;
      float* d = As + (q * 4) * LDA + co;
      d[0] = ra[e].x; d[LDA] = ra[e].y; d[2 * LDA] = ra[e].z; d[3 * LDA] = ra[e].w;
      bsum[e] += (ra[e].x + ra[e].y) + (ra[e].z + ra[e].w);
    }
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int id = tid + 256 * e, q = id & 7, ci = id >> 3;
      float* d = Bs + (q * 4) * LDB + ci;
      d[0] = rb[e].x; d[LDB] = rb[e].y; d[2 * LDB] = rb[e].z; d[3 * LDB] = rb[e].w;
    }
  };

  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk();
    __syncthreads();
    for (long c = c_begin; c < c_end; ++c) {
      if (c + 1 < c_end) load_chunk(c + 1);  // in flight under this chunk's MFMAs
      mma_chunk<2, 2, TM, TN, PC>(As, Bs, LDA, LDB, acc, wm, wn, l31, h);
      __syncthreads();  // everyone is done reading this chunk
      if (c + 1 < c_end) {
        store_chunk();
        __syncthreads();
      }
    }
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int ci = ci0 + (wn * TN + tn) * 32 + l31;
    if (ci >= p.Cin) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < p.Cout) {
          float* dst = dw_out + (long)co * p.K + ci;
          if (p.atomic)
            atomicAdd(dst, acc[tm][tn][r]);
          else
            *dst = acc[tm][tn][r];
        }
      }
  }
  if (want_bias) {  // the 8 lanes that share id / 8 hold the 8 position quads of output channel id / 8
#pragma unroll
    for (int e = 0; e < NA; ++e) {
      float v = bsum[e];
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int co = co0 + ((tid + 256 * e) >> 3);
      if ((tid & 7) == 0 && co < p.Cout) {
        if (p.atomic)
          atomicAdd(db_out + co, v);
        else
          db_out[co] = v;
      }
    }
  }
}

static bool wgrad_tile1_shape_ok(int Cin, int H, int W, int Cout, long Npos, int ks) {
  const char* sw = getenv("I2R_WGRAD_TILE");  // 0: gather kernel everywhere, 3: only the 3x3 tile kernel
  if ((sw && (atoi(sw) == 0 || atoi(sw) == 3)) || ks != 1) return false;
  if ((H * W) % 32 != 0 || Cin < 64 || Cout < 64) return false;
  static const long minpos = getenv("I2R_WGRAD_TILE_MINPOS") ? atol(getenv("I2R_WGRAD_TILE_MINPOS")) : 2048;
  return Npos >= minpos;
}
static bool wgrad_tile1_ok(const WgradParams& p, int ks) {
  return wgrad_tile1_shape_ok(p.Cin, p.H, p.W, p.Cout, p.Npos, ks) && (((uintptr_t)p.x | (uintptr_t)p.dy) & 15) == 0;
}

// Position split against the atomics' cost: a split of ns costs (chunks / ns) * t_chunk of matrix work per workgroup
// and outputs * ns atomics device-wide; the sum is least at ns = sqrt(chunks * t_chunk * rate / outputs).  t_chunk
// ~2 us (32 positions of a 64 x 64 tile incl. its two barriers and its share of L2 traffic; swept 0.5 - 5), rate ~0.2 atomics per ns (measured: 16.7 M in 80 us).
static void wgrad_tile1_plan(const WgradParams& p, int& chunks_per_split, int& nsplit) {
  const long tiles = (long)i2r_cdiv(p.Cin, 64) * i2r_cdiv(p.Cout, 64);
  const long total = p.Npos / 32;
  static const double t_chunk = getenv("I2R_WGRAD1_TCHUNK") ? atof(getenv("I2R_WGRAD1_TCHUNK")) : 2.0;
  static const double rate = getenv("I2R_WGRAD1_RATE") ? atof(getenv("I2R_WGRAD1_RATE")) : 2.0e5;  // per us
  double ns_d = sqrt((double)total * t_chunk * rate / ((double)p.Cin * p.Cout));
  long ns = (long)(ns_d + 0.5);
  if (ns * tiles > 2048) ns = 2048 / tiles;  // no more than ~8 workgroups per CU
  if (ns > total) ns = total;
  if (ns < 1) ns = 1;
  chunks_per_split = (int)((total + ns - 1) / ns);
  nsplit = (int)((total + chunks_per_split - 1) / chunks_per_split);
}

static void wgrad_tile1_launch(const WgradParams& p, hipStream_t s) {
  dim3 grid(i2r_cdiv(p.Cin, 64), i2r_cdiv(p.Cout, 64), p.nsplit);
  hipLaunchKernelGGL((conv_wgrad1_tile_kernel<64, 64>), grid, dim3(256), 0, s, p);
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
  const bool t3 = wgrad_tile_ok(p, ksize), t1 = !t3 && wgrad_tile1_ok(p, ksize);
  if (t3 || t1) {
    if (t3)
      wgrad_tile_plan(p, p.chunks_per_split, p.nsplit);
    else
      wgrad_tile1_plan(p, p.chunks_per_split, p.nsplit);
    if (p.nsplit > 1 && !p.atomic) {
      (void)hipMemsetAsync(p.dw, 0, sizeof(float) * (size_t)p.Cout * p.K, s);
      if (p.db) (void)hipMemsetAsync(p.db, 0, sizeof(float) * (size_t)p.Cout, s);
      p.atomic = 1;
    }
    if (t3) {
      if (!wgrad_tile_launch(p, s)) return I2R_ERR_HIP;
    } else {
      wgrad_tile1_launch(p, s);
    }
    return i2r_check_launch("conv2d_wgrad");
  }
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
  long need = 0;
  switch (wgrad_variant(K, Cout, Npos)) {
    case 0: need = wgrad_det_floats<1, 4, 1, 1>(K, Cout, Npos); break;
    case 1: need = wgrad_det_floats<4, 1, 1, 1>(K, Cout, Npos); break;
    case 2: need = wgrad_det_floats<2, 2, 2, 2>(K, Cout, Npos); break;
    default: need = wgrad_det_floats<2, 2, 1, 1>(K, Cout, Npos); break;
  }
  if (wgrad_tile_shape_ok(Cin, H, W, Cout, Npos, ksize)) {  // the halo-tile kernel's slabs (it needs aligned operands;
    WgradParams q{};                                        // the query covers whichever kernel the call then takes)
    q.Cin = Cin; q.Cout = Cout; q.Npos = Npos;
    int cps, ns;
    wgrad_tile_plan(q, cps, ns);
    const long t = (long)ns * ((long)Cout * K + Cout);
    if (t > need) need = t;
  }
  if (wgrad_tile1_shape_ok(Cin, H, W, Cout, Npos, ksize)) {
    WgradParams q{};
    q.Cin = Cin; q.Cout = Cout; q.Npos = Npos;
    int cps, ns;
    wgrad_tile1_plan(q, cps, ns);
    const long t = (long)ns * ((long)Cout * K + Cout);
    if (t > need) need = t;
  }
  return need;
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
  const bool t3 = wgrad_tile_ok(p, ksize), t1 = !t3 && wgrad_tile1_ok(p, ksize);
  if (t3 || t1) {
    if (t3)
      wgrad_tile_plan(p, p.chunks_per_split, p.nsplit);
    else
      wgrad_tile1_plan(p, p.chunks_per_split, p.nsplit);
    const long n_dw = (long)p.Cout * p.K, n_all = n_dw + (dbias ? p.Cout : 0);
    p.dw = ws;
    p.db = dbias ? ws : nullptr;
    p.atomic = 0;
    p.slab_stride = n_dw + p.Cout;
    if (t3) {
      if (!wgrad_tile_launch(p, s)) return I2R_ERR_HIP;
    } else {
      wgrad_tile1_launch(p, s);
    }
    int nb = i2r_cdiv(n_all, 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(nb), dim3(256), 0, s, ws, p.slab_stride, p.nsplit, dw, dbias,
                       n_dw, n_all, accumulate);
    return i2r_check_launch("conv2d_wgrad_det");
  }
  switch (wgrad_variant(p.K, Cout, p.Npos)) {
    case 0: launch_wgrad_det<1, 4, 1, 1>(p, ksize, ws, dw, dbias, accumulate, s); break;
    case 1: launch_wgrad_det<4, 1, 1, 1>(p, ksize, ws, dw, dbias, accumulate, s); break;
    case 2: launch_wgrad_det<2, 2, 2, 2>(p, ksize, ws, dw, dbias, accumulate, s); break;
    default: launch_wgrad_det<2, 2, 1, 1>(p, ksize, ws, dw, dbias, accumulate, s); break;
  }
  return i2r_check_launch("conv2d_wgrad_det");
}
