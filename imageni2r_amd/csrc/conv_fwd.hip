// Implicit-GEMM convolution forward / data-gradient (3x3 pad 1 and 1x1, stride 1, NCHW fp32) on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chain, 64 FLOP/clk/SIMD).  Replaces torch.nn.functional.conv2d as called
// from models/networks.py:87 and autograd's conv input-gradient.
//
//   Out[co][n] = sum_k  Wp[k][co] * Im2col[k][n],   k = (ci, r, s) in the weight's OIHW order, n = (b, y, x)
//
//   MFMA rows (accumulator registers) = output channel, MFMA columns (lanes) = position: each accumulator register is
//   stored as 32 consecutive positions of one output plane = 128-B coalesced NCHW stores.
//
// Design points (what the rocprof counters of the first version asked for -- it spent 278 VALU instructions per
// 32 MFMAs on address arithmetic and sat at 45% MFMA utilisation):
//   * weights are pre-packed K-major, Wp[Kpad][Cout4] (i2r_conv2d_pack_weight_f32; also does the transpose + flip of
//     the data-gradient), so a weight tile is rows of contiguous floats: 16-B buffer loads, 16-B LDS stores;
//   * K is walked in chunks of 18 = 2 channels x 9 taps for 3x3 (16 channels for 1x1), so the tap, the halo test and
//     the element offset of every im2col element a thread fetches are LOOP-INVARIANT: they live in registers as
//     buffer voffsets, the chunk advance is the instruction's scalar soffset, and out-of-image taps carry an
//     out-of-range voffset that the buffer unit turns into 0.0 -- the main loop has no address VALU at all;
//   * register-staged prefetch, two chunks deep: the loads of chunks c+1 and c+2 are in flight under the MFMAs of c;
//   * split-K over blockIdx.z (deterministic: partial slabs + a fused reduce/bias/residual pass) for the low-resolution
//     layers whose M x N tile count cannot fill 256 CUs.
#include "conv_common.h"
#include "../../include/i2r_hip.h"
#include <stdlib.h>

#define OOR 0x80000000u  // voffset sentinel: beyond any buffer we accept (< 2 GiB) => buffer load returns 0

struct Conv2Params {
  const float* x;     // activations (B, Cin, H, W)
  const float* wp;    // packed weights [Kpad][Cout4]
  const float* bias;  // (Cout) or null
  const float* res;   // residual (B, Cout, H, W) or null
  float* out;         // (B, Cout, H, W), or the split-K slab base when nsplit > 1
  unsigned x_bytes, wp_bytes;
  int B, Cin, H, W, Cout, Cout4, HW, Kpad;
  long Npos;
  int chunks_per_split, nsplit;
  long slab_stride;   // elements between split-K slabs
  int slab_cn;        // 1: split-K slabs are laid out [co][n] (n = position over the whole batch) instead of NCHW: with
                      // HW < 32 an NCHW tile leaves as 4*HW-byte pieces, once per split; [co][n] leaves as 128-byte runs
                      // and the reduce pass scatters once
  int ablate;         // tuning only (I2R_CONV_ABLATE): 8 skip the epilogue
};

__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 buf_load_f32x4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int WM, int WN, int TM, int TN, int KS>
__global__ __launch_bounds__(256) void conv_fwd_kernel(Conv2Params p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int BKC = (KS == 3) ? 18 : 16;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A4 = BM / 4;                          // float4 per weight-tile row
  constexpr int NA = (BKC * A4 + 255) / 256;          // float4 loads per thread per chunk
  constexpr int KSTEP = 256 / BN;                     // k rows covered by one sweep of the block over the B tile
  constexpr int NB = (BKC + KSTEP - 1) / KSTEP;       // im2col loads per thread per chunk
  __shared__ __attribute__((aligned(16))) float smem[2 * BKC * LDA + 2 * BKC * LDB];
  float* As = smem;
  float* Bs = smem + 2 * BKC * LDA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM;
  const long n0 = (long)blockIdx.x * BN;

  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

  // ---- loop-invariant per-thread load descriptors
  unsigned voffA[NA];
  int ldsA[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int idx = tid + 256 * j;
    int row = idx / A4, c4 = idx % A4;
    bool ok = idx < BKC * A4 && (m0 + c4 * 4) < p.Cout4;
    voffA[j] = ok ? (unsigned)((row * p.Cout4 + m0 + c4 * 4) * 4) : OOR;
    ldsA[j] = (idx < BKC * A4) ? row * LDA + c4 * 4 : -1;
  }
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
  unsigned voffB[NB];
  int ldsB[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int kk = b_kk0 + KSTEP * j;
    bool ok = nvalid && kk < BKC;
    long off;
    if (KS == 3) {
      int cl = kk / 9, tap = kk - 9 * cl;
      int r = tap / 3, s = tap - 3 * r;
      int iy = py + r - 1, ix = px + s - 1;
      ok = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      off = ((long)pb * p.Cin + cl) * p.HW + iy * p.W + ix;
    } else {
      off = ((long)pb * p.Cin + kk) * p.HW + py * p.W + px;
    }
    voffB[j] = ok ? (unsigned)(off * 4) : OOR;
    ldsB[j] = (kk < BKC) ? kk * LDB + b_j : -1;
  }
  constexpr int CH_PER_CHUNK = (KS == 3) ? 2 : 16;
  const unsigned stepA = (unsigned)(BKC * p.Cout4 * 4);       // bytes per chunk in the packed weights
  const unsigned stepB = (unsigned)(CH_PER_CHUNK * p.HW * 4); // bytes per chunk in the activations

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Register stages: the loads of chunks c+1 and c+2 are in flight while chunk c is multiplied.  One chunk is only
  // 9 (3x3) or 8 (1x1) MFMAs per wave -- ~550 cycles against ~2000 of load latency -- and the low-resolution layers
  // that come here in training run two or three workgroups per CU, so with a single chunk in flight (the first form
  // of this loop) every iteration waited for memory: 1.2 us per chunk.
  constexpr int DEPTH = 3;
  f32x4 ra[DEPTH][NA];
  float rb[DEPTH][NB];
  auto load_chunk = [&](int c, f32x4 (&qa)[NA], float (&qb)[NB]) {
    unsigned sA = (unsigned)c * stepA, sB = (unsigned)c * stepB;
#pragma unroll
    for (int j = 0; j < NA; ++j) qa[j] = buf_load_f32x4(rA, voffA[j], sA);
#pragma unroll
    for (int j = 0; j < NB; ++j) qb[j] = buf_load_f32(rB, voffB[j], sB);
  };
  auto store_chunk = [&](int buf, const f32x4 (&qa)[NA], const float (&qb)[NB]) {
    float* a = As + buf * BKC * LDA;
    float* b = Bs + buf * BKC * LDB;
#pragma unroll
    for (int j = 0; j < NA; ++j)
      if (ldsA[j] >= 0) *reinterpret_cast<f32x4*>(a + ldsA[j]) = qa[j];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (ldsB[j] >= 0) b[ldsB[j]] = qb[j];
  };

  const int total_chunks = p.Kpad / BKC;
  int c_begin = blockIdx.z * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  if (c_end > total_chunks) c_end = total_chunks;
  if (c_begin < c_end) {
    // The loop body has no conditional loads (past the end the LAST chunk is fetched again and never used), so the
    // compiler can count: each store waits with a partial vmcnt that leaves the two younger chunks in flight.
    const int c_last = c_end - 1;
    load_chunk(c_begin, ra[0], rb[0]);
    load_chunk(c_begin + 1 < c_end ? c_begin + 1 : c_last, ra[1], rb[1]);
    for (int c = c_begin; c < c_end; c += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int cc = c + u;
        if (cc < c_end) {  // uniform over the workgroup
          const int cur = (cc - c_begin) & 1;
          load_chunk(cc + 2 < c_end ? cc + 2 : c_last, ra[(u + 2) % DEPTH], rb[(u + 2) % DEPTH]);
          // LDS buffer `cur` was last read by the MFMAs of chunk cc - 2; every wave finished them before it passed
          // the barrier of chunk cc - 1
          store_chunk(cur, ra[u], rb[u]);
          __syncthreads();
          mma_chunk<WM, WN, TM, TN, BKC>(As + cur * BKC * LDA, Bs + cur * BKC * LDB, LDA, LDB, acc, wm, wn, l31, h);
        }
      }
    }
  }

  // epilogue: C/D layout col = lane&31 (position), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (output channel)
  if (p.ablate & 8) return;
  // Bias and residual are loaded for the whole tile BEFORE the first store: the output may alias them as far as the
  // compiler knows, so a load placed after a store would wait for it (64 serial round trips per lane otherwise).
  const bool direct = p.nsplit == 1;
  float* outp = p.out + (direct ? 0 : (long)blockIdx.z * p.slab_stride);
  const bool use_bias = direct && p.bias != nullptr, use_res = direct && p.res != nullptr;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    long n = n0 + (wn * TN + tn) * 32 + l31;
    const bool nok = n < p.Npos;
    int b = nok ? (int)(n / p.HW) : 0;
    int pos = nok ? (int)(n % p.HW) : 0;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      float add[16];  // one 32x32 sub-tile at a time keeps the register footprint of the main loop
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        bool ok = nok && co < p.Cout;
        long o = ((long)b * p.Cout + co) * p.HW + pos;
        float a = (use_bias && ok) ? p.bias[co] : 0.0f;
        if (use_res && ok) a += p.res[o];
        add[r] = a;
      }
      if (!direct && p.slab_cn) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int co = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (nok && co < p.Cout) outp[(long)co * p.Npos + n] = acc[tm][tn][r];
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = m0 + (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (nok && co < p.Cout) outp[((long)b * p.Cout + co) * p.HW + pos] = acc[tm][tn][r] + add[r];
      }
    }
  }
}

// out = sum_z slab[z] + bias + residual   (the deterministic split-K combine).  slab_cn: the slabs are [co][n] (see
// Conv2Params) and are walked in that order -- contiguous reads of every slab, one scattered write.
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int nsplit,
                                     const float* __restrict__ bias, const float* __restrict__ res,
                                     float* __restrict__ out, long total, int Cout, int HW, long Npos, int slab_cn) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v = slabs[i];
    int z = 1;
    for (; z + 3 < nsplit; z += 4) {  // four independent loads in flight; summation order stays z-ascending
      float a = slabs[z * slab_stride + i], b = slabs[(z + 1) * slab_stride + i];
      float c = slabs[(z + 2) * slab_stride + i], d = slabs[(z + 3) * slab_stride + i];
      v = (((v + a) + b) + c) + d;
    }
    for (; z < nsplit; ++z) v += slabs[z * slab_stride + i];
    long o = i;
    int co;
    if (slab_cn) {
      co = (int)(i / Npos);
      const long n = i - (long)co * Npos;
      const long b = n / HW;
      o = (b * Cout + co) * HW + (n - b * HW);
    } else {
      co = (int)((i / HW) % Cout);
    }
    if (bias) v += bias[co];
    if (res) v += res[o];
    out[o] = v;
  }
}

// Wp[k][co] (Kpad x Cout4, zero padded).
//   forward : k = ci*KK + tap,          Wp[k][co] = w[co][ci][tap]
//   dgrad   : conv of dY with the transposed, spatially flipped weights: "Cout" = Cin, "Cin" = Cout,
//             k = co*KK + tap,          Wp[k][ci] = w[co][ci][KK-1-tap]
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int KK,
                                   int dgrad, int M, int M4, int K, int Kpad) {
  long total = (long)Kpad * M4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int m = (int)(i % M4);
    int k = (int)(i / M4);
    float v = 0.0f;
    if (m < M && k < K) {
      int c = k / KK, tap = k - c * KK;
      if (!dgrad)
        v = w[((long)m * Cin + c) * KK + tap];
      else
        v = w[((long)c * Cin + m) * KK + (KK - 1 - tap)];
    }
    wp[i] = v;
  }
}

// Multi-tensor form: every conv weight of the network (forward and data-gradient layouts) in ONE launch.
// table rows {w, wp, Cout, Cin, ksize, dgrad} (int64); blockmap {tensor, block} with PACK_BLOCK packed floats per block.
#define PACK_BLOCK 4096
__device__ __forceinline__ float pack_src(const float* __restrict__ w, int m, int k, int M, int K, int KK, int Cin,
                                          int dgrad) {
  if (m >= M || k >= K) return 0.0f;
  int c = k / KK, tap = k - c * KK;
  return dgrad ? w[((long)c * Cin + m) * KK + (KK - 1 - tap)] : w[((long)m * Cin + c) * KK + tap];
}
// It is a transpose, done in 64 (k) x 64 (m) tiles through LDS so that BOTH global sides move 256 contiguous bytes per
// wave: the first form gave every block a band of 4096 / M4 whole k-rows, which for the wide layers (M = 512: 8 rows)
// read the source in 32-byte pieces -- 1.9 TB/s over B1's 70 M low-resolution weights.  Read order per layout:
//   forward        : lanes along k (w[m][k0 ..] is contiguous in OIHW);
//   dgrad, 1x1     : lanes along m (w[c][m0 ..] is contiguous: no transpose at all);
//   dgrad, 3x3     : the (ci, tap) slab of each co the tile touches, front to back (64 ci x 9 taps contiguous), each
//                    element dropped at its flipped tap's row.
// A tensor keeps ceil(total / PACK_BLOCK) blocks (the host's block map); they stride over its ceil(Kpad / 64) x
// ceil(M4 / 64) tiles.
#define PT 64
#define PTP 65
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const long* __restrict__ table,
                                                                const int2* __restrict__ blockmap) {
  __shared__ float tile[PT * PTP];
  const int2 bm = blockmap[blockIdx.x];
  const long* e = table + 6L * bm.x;
  const float* w = (const float*)e[0];
  float* wp = (float*)e[1];
  const int Cout = (int)e[2], Cin = (int)e[3], ks = (int)e[4], dgrad = (int)e[5];
  const int M = dgrad ? Cin : Cout, C = dgrad ? Cout : Cin;
  const int KK = ks * ks, K = C * KK, bkc = ks == 3 ? 18 : 16;
  const int Kpad = (K + bkc - 1) / bkc * bkc, M4 = (M + 3) / 4 * 4;
  const long total = (long)Kpad * M4;
  const int nb_t = (int)((total + PACK_BLOCK - 1) / PACK_BLOCK);  // blocks the host gave this tensor
  const int tiles_m = (M4 + PT - 1) / PT, tiles_k = (Kpad + PT - 1) / PT;
  const int tid = threadIdx.x;
  for (int t = bm.y; t < tiles_m * tiles_k; t += nb_t) {
    const int k0 = (t / tiles_m) * PT, m0 = (t % tiles_m) * PT;
    __syncthreads();  // the previous tile is written out
    if (dgrad && KK == 9) {
      for (int j = tid; j < PT * PTP; j += 256) tile[j] = 0.0f;  // rows / columns the slabs do not reach (padding)
      __syncthreads();
      const int c_lo = k0 / 9, c_hi = (k0 + PT - 1) / 9;  // the co whose (ci, tap) slabs meet this tile's k rows
      const int mw = (M - m0 < PT ? M - m0 : PT) * 9;     // floats of a slab inside this tile's m range
      for (int c = c_lo; c <= c_hi && c < C; ++c) {
        const float* src = w + ((long)c * Cin + m0) * 9;
        for (int r = tid; r < mw; r += 256) {
          const int mm = r / 9, tap = 8 - (r - 9 * mm), kk = c * 9 + tap - k0;
          if (kk >= 0 && kk < PT) tile[kk * PTP + mm] = src[r];
        }
      }
    } else if (dgrad) {  // 1x1: Wp[c][m] = w[c][m]
#pragma unroll 4
      for (int i = 0; i < PT * PT / 256; ++i) {
        const int j = tid + 256 * i, mm = j & (PT - 1), kk = j >> 6;
        tile[kk * PTP + mm] = pack_src(w, m0 + mm, k0 + kk, M, K, KK, Cin, 1);
      }
    } else {
#pragma unroll 4
      for (int i = 0; i < PT * PT / 256; ++i) {
        const int j = tid + 256 * i, kk = j & (PT - 1), mm = j >> 6;
        tile[kk * PTP + mm] = pack_src(w, m0 + mm, k0 + kk, M, K, KK, Cin, 0);
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < PT * PT / 256; ++i) {
      const int j = tid + 256 * i, mm = j & (PT - 1), kk = j >> 6;
      if (k0 + kk < Kpad && m0 + mm < M4) wp[(long)(k0 + kk) * M4 + m0 + mm] = tile[kk * PTP + mm];
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static inline int conv_bkc(int ks) { return ks == 3 ? 18 : 16; }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

extern "C" long i2r_conv2d_packed_floats(int Cout, int Cin, int ksize, int dgrad) {
  int M = dgrad ? Cin : Cout, C = dgrad ? Cout : Cin;
  return (long)round_up(C * ksize * ksize, conv_bkc(ksize)) * round_up(M, 4);
}

extern "C" int i2r_conv2d_pack_weight_f32(const float* w, float* wp, int Cout, int Cin, int ksize, int dgrad,
                                          i2r_stream_t stream) {
  I2R_REQUIRE(w && wp && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "pack_weight: bad argument");
  int M = dgrad ? Cin : Cout, C = dgrad ? Cout : Cin;
  int KK = ksize * ksize, K = C * KK, Kpad = round_up(K, conv_bkc(ksize)), M4 = round_up(M, 4);
  long total = (long)Kpad * M4;
  int nb = i2r_cdiv(total, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, KK, dgrad, M,
                     M4, K, Kpad);
  return i2r_check_launch("pack_weight");
}

extern "C" int i2r_conv2d_pack_block_floats(void) { return PACK_BLOCK; }

extern "C" int i2r_conv2d_pack_multi_f32(const void* table, const void* blockmap, int nblocks, i2r_stream_t stream) {
  I2R_REQUIRE(table && blockmap && nblocks > 0, "pack_multi: bad argument");
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, (const long*)table,
                     (const int2*)blockmap);
  return i2r_check_launch("pack_multi");
}

struct ConvPlan {
  int variant;  // 0: 128x128, 1: 64x64, 2: 32x128, 3: 128x32
  int bm, bn, nsplit, chunks_per_split;
};

static ConvPlan conv_plan(long Npos, int Cout, int Kpad, int ks, bool allow_split) {
  ConvPlan pl;
  // Tile choice, from a sweep over the B1/C layer shapes at batch 32..1024 (tools/conv_sweep.py): 128x128 tiles once
  // they alone fill the chip; below that 64x64 tiles (4x the workgroups, no slower per flop on the 64-cycle fp32 MFMA)
  // and split-K up to ~1024 workgroups with at least 8 chunks (144 / 128 k) per split.
  const long tiles128 = (long)i2r_cdiv(Npos, 128) * i2r_cdiv(Cout, 128);
  if (Cout <= 32) {
    pl.variant = 2; pl.bm = 32; pl.bn = 128;
  } else if (Npos <= 32) {
    pl.variant = 3; pl.bm = 128; pl.bn = 32;
  } else if (Cout <= 64 || Npos <= 64 || tiles128 < 256) {
    pl.variant = 1; pl.bm = 64; pl.bn = 64;
  } else {
    pl.variant = 0; pl.bm = 128; pl.bn = 128;
  }
  if (const char* ev = getenv("I2R_CONV_VARIANT")) {  // tuning/debug override
    int v = atoi(ev);
    static const int dims[6][2] = {{128, 128}, {64, 64}, {32, 128}, {128, 32}, {128, 64}, {64, 128}};
    if (v >= 0 && v < 6) { pl.variant = v; pl.bm = dims[v][0]; pl.bn = dims[v][1]; }
  }
  long tiles = (long)i2r_cdiv(Npos, pl.bn) * i2r_cdiv(Cout, pl.bm);
  int chunks = Kpad / conv_bkc(ks);
  int nsplit = 1;
  if (allow_split && tiles < 512) {
    static const long target = getenv("I2R_CONV_TARGET") ? atol(getenv("I2R_CONV_TARGET")) : 1024;
    static const long minch = getenv("I2R_CONV_MINCHUNKS") ? atol(getenv("I2R_CONV_MINCHUNKS")) : 8;
    static const long maxsplit = getenv("I2R_CONV_MAXSPLIT") ? atol(getenv("I2R_CONV_MAXSPLIT")) : 32;
    long want = (target + tiles - 1) / tiles;
    long maxs = chunks / minch;
    if (want > maxs) want = maxs;
    if (want > maxsplit) want = maxsplit;
    if (want > 1) nsplit = (int)want;
  }
  if (const char* ev = getenv("I2R_CONV_NSPLIT")) {  // tuning/debug override
    int v = atoi(ev);
    if (v >= 1 && allow_split) nsplit = v > chunks ? chunks : v;
  }
  pl.chunks_per_split = (chunks + nsplit - 1) / nsplit;
  pl.nsplit = (chunks + pl.chunks_per_split - 1) / pl.chunks_per_split;
  return pl;
}

extern "C" long i2r_conv2d_splitk_floats(int B, int Cin, int H, int W, int Cout, int ksize) {
  long Npos = (long)B * H * W;
  int Kpad = round_up(Cin * ksize * ksize, conv_bkc(ksize));
  ConvPlan pl = conv_plan(Npos, Cout, Kpad, ksize, true);
  return pl.nsplit > 1 ? (long)pl.nsplit * Npos * Cout : 0;
}

template <int WM, int WN, int TM, int TN>
static void launch_fwd(const Conv2Params& p, int ks, dim3 grid, hipStream_t s) {
  if (ks == 3)
    hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, TM, TN, 3>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<WM, WN, TM, TN, 1>), grid, dim3(256), 0, s, p);
}

extern "C" int i2r_conv2d_fwd_packed_f32(const float* x, const float* wp, const float* bias, const float* residual,
                                         float* out, float* splitk_ws, long splitk_floats, int B, int Cin, int H,
                                         int W, int Cout, int ksize, i2r_stream_t stream) {
  I2R_REQUIRE(x && wp && out, "conv2d_fwd: null pointer");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_fwd: kernel size %d unsupported (1 or 3)", ksize);
  I2R_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d_fwd: non-positive dimension");
  long x_bytes = (long)B * Cin * H * W * 4;
  int Kpad = round_up(Cin * ksize * ksize, conv_bkc(ksize)), Cout4 = round_up(Cout, 4);
  long wp_bytes = (long)Kpad * Cout4 * 4;
  I2R_REQUIRE(x_bytes < 0x7FFFFFFFL && wp_bytes < 0x7FFFFFFFL,
              "conv2d_fwd: input (%ld B) or packed weights (%ld B) exceed the 2 GiB buffer-addressing limit; "
              "split the batch", x_bytes, wp_bytes);
  long Npos = (long)B * H * W;
  ConvPlan pl = conv_plan(Npos, Cout, Kpad, ksize, splitk_ws != nullptr);
  if (pl.nsplit > 1 && (long)pl.nsplit * Npos * Cout > splitk_floats) {
    pl.nsplit = 1;  // workspace too small: fall back to a single pass (still correct)
    pl.chunks_per_split = Kpad / conv_bkc(ksize);
  }
  Conv2Params p;
  p.x = x; p.wp = wp; p.bias = bias; p.res = residual;
  p.out = pl.nsplit > 1 ? splitk_ws : out;
  p.x_bytes = (unsigned)x_bytes; p.wp_bytes = (unsigned)wp_bytes;
  p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.Cout4 = Cout4; p.HW = H * W; p.Kpad = Kpad;
  p.Npos = Npos; p.chunks_per_split = pl.chunks_per_split; p.nsplit = pl.nsplit; p.slab_stride = Npos * Cout;
  { const char* ab = getenv("I2R_CONV_ABLATE"); p.ablate = ab ? atoi(ab) : 0; }
  { const char* cn = getenv("I2R_CONV_SLAB_CN"); p.slab_cn = (pl.nsplit > 1 && H * W < 32 && !(cn && cn[0] == '0')) ? 1 : 0; }
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(i2r_cdiv(Npos, pl.bn), i2r_cdiv(Cout, pl.bm), pl.nsplit);
  switch (pl.variant) {
    case 0: launch_fwd<2, 2, 2, 2>(p, ksize, grid, s); break;
    case 1: launch_fwd<2, 2, 1, 1>(p, ksize, grid, s); break;
    case 2: launch_fwd<1, 4, 1, 1>(p, ksize, grid, s); break;
    case 3: launch_fwd<4, 1, 1, 1>(p, ksize, grid, s); break;
    case 4: launch_fwd<2, 2, 2, 1>(p, ksize, grid, s); break;
    default: launch_fwd<2, 2, 1, 2>(p, ksize, grid, s); break;
  }
  if (pl.nsplit > 1) {
    long total = Npos * Cout;
    int nb = i2r_cdiv(total, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, s, splitk_ws, p.slab_stride, pl.nsplit, bias,
                       residual, out, total, Cout, H * W, Npos, p.slab_cn);
  }
  return i2r_check_launch("conv2d_fwd");
}

// Convenience forms that pack on the fly.  ws must hold i2r_conv2d_packed_floats(...) (+ i2r_conv2d_splitk_floats(...)
// to enable split-K) floats.
extern "C" int i2r_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* residual,
                                  float* out, float* ws, long ws_floats, int B, int Cin, int H, int W, int Cout,
                                  int ksize, i2r_stream_t stream) {
  I2R_REQUIRE(w && ws, "conv2d_fwd: null weight / workspace");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_fwd: kernel size %d unsupported (1 or 3)", ksize);
  long pf = i2r_conv2d_packed_floats(Cout, Cin, ksize, 0);
  I2R_REQUIRE(ws_floats >= pf, "conv2d_fwd: workspace of %ld floats < %ld needed for the packed weights", ws_floats,
              pf);
  if (int e = i2r_conv2d_pack_weight_f32(w, ws, Cout, Cin, ksize, 0, stream)) return e;
  long rest = ws_floats - pf;
  return i2r_conv2d_fwd_packed_f32(x, ws, bias, residual, out, rest > 0 ? ws + pf : nullptr, rest, B, Cin, H, W,
                                   Cout, ksize, stream);
}

// dx = conv2d(dy, flip(w)^T): the forward kernel on weights packed in dgrad mode.
extern "C" int i2r_conv2d_dgrad_f32(const float* dy, const float* w, float* ws, long ws_floats, float* dx, int B,
                                    int Cin, int H, int W, int Cout, int ksize, i2r_stream_t stream) {
  I2R_REQUIRE(dy && w && ws && dx, "conv2d_dgrad: null pointer");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_dgrad: kernel size %d unsupported (1 or 3)", ksize);
  long pf = i2r_conv2d_packed_floats(Cout, Cin, ksize, 1);
  I2R_REQUIRE(ws_floats >= pf, "conv2d_dgrad: workspace of %ld floats < %ld needed for the packed weights",
              ws_floats, pf);
  if (int e = i2r_conv2d_pack_weight_f32(w, ws, Cout, Cin, ksize, 1, stream)) return e;
  long rest = ws_floats - pf;
  return i2r_conv2d_fwd_packed_f32(dy, ws, nullptr, nullptr, dx, rest > 0 ? ws + pf : nullptr, rest, B, Cout, H, W,
                                   Cin, ksize, stream);
}
