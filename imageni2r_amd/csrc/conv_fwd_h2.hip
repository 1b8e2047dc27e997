// Convolution forward with fp32-grade accuracy on the gfx950 *half-precision* matrix cores ("f16x2 split").
//
// Every fp32 operand x is written as x = hi + lo with hi = f16(x), lo = f16(x - hi): 22 significand bits, and the product
// of two fp16 numbers is exact in the MFMA's fp32 accumulator.  Then
//        a*b = a_hi*b_hi + a_hi*b_lo + a_lo*b_hi     (+ a_lo*b_lo ~ 2^-22 |ab|, dropped)
// i.e. THREE v_mfma_f32_32x32x16_f16 (32 cycles each for a 32x32x16 tile-step) replace EIGHT v_mfma_f32_32x32x2_f32
// (64 cycles each): 5.3x the matrix throughput of the exact-fp32 kernel (conv_fwd.hip) at ~1e-6 relative error,
// far inside the 1e-4 bar of the path.  Used for inference (the sampler's 35 denoiser evaluations per series);
// training keeps the exact fp32 kernels because gradients can underflow fp16's range.
//
// Range: activations must satisfy |x| < 65504 (fp16 max); weights are pre-scaled by 2^6 when packed (so that the lo
// halves of ~1/sqrt(fan_in) weights stay normal fp16 numbers) and the epilogue multiplies by 2^-6 (exact).
//
// GEMM mapping is the one of conv_fwd.hip (MFMA rows = output channel, lanes = position).  K order of the packed
// weights is chosen for the f16 fragments (8 consecutive k per lane): a chunk is one kernel ROW r and 16 input channels,
// k = s*16 + c (3 k-steps of 16 channels at tap column s); 1x1: a chunk is 64 channels.  Per-lane im2col offsets are
// invariant across the channel blocks of one kernel row (buffer voffsets, recomputed 3 times per kernel), the channel
// advance is the scalar soffset, halo taps are out-of-range voffsets that the buffer unit returns as 0.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

#define OOR 0x80000000u
#define WSCALE 64.0f
#define WSCALE_INV 0.015625f

struct ConvHParams {
  const float* x;       // activations (B, Cin, H, W) fp32
  const void* wp;       // packed split weights: [chunk][Cout][2][KC] f16
  const float* bias;
  const float* res;
  float* out;
  unsigned x_bytes, wp_bytes;
  int B, Cin, H, W, Cout, HW, nCB;
  long Npos;
  int chunks_per_split, nsplit;
  long slab_stride;
};

__device__ __forceinline__ float bufload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ uint4v bufload4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// split 8 floats into packed hi / lo half8 (round-to-nearest both)
__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)(v[j] - (float)h);
  }
}

// KS = 3: chunk = (kernel row r, 16-channel block), KC = 48 (k = s*16 + c).   KS = 1: chunk = 64 channels, KC = 64.
template <int KS>
__global__ __launch_bounds__(256, 2) void conv_fwd_h2_kernel(ConvHParams p) {
  constexpr int BM = 128, BN = 128;
  constexpr int KC = (KS == 3) ? 48 : 64;
  constexpr int NKS = KC / 16;             // MFMA k-steps per chunk
  constexpr int ROWL = KC * 2 + 16;        // LDS row bytes (odd multiple of 16 B -> conflict-free b128 rows)
  constexpr int RB = 2 * KC * 2;           // global bytes per packed weight row (hi | lo)
  constexpr int PPR = RB / 16;             // 16-B pieces per weight row
  constexpr int NA = BM * PPR / 256;       // weight pieces per thread per chunk (6 or 8)
  constexpr int NSEG = (KS == 3) ? 3 : 4;  // 8-channel segments a thread converts per chunk (one ds_write pair each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ah = smem;                       // [BM][ROWL]
  unsigned char* Al = Ah + BM * ROWL;
  unsigned char* Bh = Al + BM * ROWL;             // [BN][ROWL]
  unsigned char* Bl = Bh + BN * ROWL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * BM;
  const long n0 = (long)blockIdx.x * BN;

  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

  // ---- weight-tile descriptors (loop invariant)
  unsigned voffA[NA];
  int ldsA[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int idx = tid + 256 * j;
    int row = idx / PPR, piece = idx % PPR;
    voffA[j] = (m0 + row < p.Cout) ? (unsigned)((m0 + row) * RB + piece * 16) : OOR;
    int lo = piece >= PPR / 2;
    ldsA[j] = (lo ? BM * ROWL : 0) + row * ROWL + (piece - lo * (PPR / 2)) * 16;
  }
  const unsigned stepA = (unsigned)p.Cout * RB;  // bytes per chunk of packed weights

  // ---- im2col descriptors: thread = (position n, channel half g)
  const int b_j = tid & 127, g = tid >> 7;
  const long nj = n0 + b_j;
  const bool nvalid = nj < p.Npos;
  int pb = 0, py = 0, px = 0;
  if (nvalid) {
    pb = (int)(nj / p.HW);
    int pos = (int)(nj % p.HW);
    py = pos / p.W;
    px = pos % p.W;
  }
  const unsigned chan_bytes = (unsigned)p.HW * 4;
  unsigned voffB[3];
  auto set_row = [&](int r) {  // voffsets of the three tap columns of kernel row r for this thread's position
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      int iy = py + r - 1, ix = px + s - 1;
      bool ok = nvalid && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      long off = ((long)pb * p.Cin + 8 * g) * p.HW + iy * p.W + ix;
      voffB[s] = ok ? (unsigned)(off * 4) : OOR;
    }
  };
  if (KS == 1) {
    long off = ((long)pb * p.Cin + 8 * g) * p.HW + py * p.W + px;
    voffB[0] = nvalid ? (unsigned)(off * 4) : OOR;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  uint4v ra[NA];
  float rb[NSEG][8];
  // chunk c -> (r, cb) for 3x3 (r-major so voffB changes only 3 times), cb for 1x1
  auto load_chunk = [&](int c) {
    unsigned sA = (unsigned)c * stepA;
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = bufload4(rA, voffA[j], sA);
    if (KS == 3) {
      int cb = c % p.nCB;
      int cvalid = p.Cin - cb * 16 - 8 * g;  // channels of this thread's half that exist (uniform per half-block)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned sB = (unsigned)(cb * 16 + j) * chan_bytes;
        bool cok = j < cvalid;
#pragma unroll
        for (int s = 0; s < 3; ++s) rb[s][j] = bufload(rB, cok ? voffB[s] : OOR, sB);
      }
    } else {
      int cvalid = p.Cin - c * 64 - 8 * g;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          unsigned sB = (unsigned)(c * 64 + q * 16 + j) * chan_bytes;
          rb[q][j] = bufload(rB, (q * 16 + j < cvalid) ? voffB[0] : OOR, sB);
        }
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<uint4v*>(Ah + ldsA[j]) = ra[j];
#pragma unroll
    for (int q = 0; q < NSEG; ++q) {
      half8 hi, lo;
      split8(rb[q], hi, lo);
      int o = b_j * ROWL + (q * 16 + 8 * g) * 2;
      *reinterpret_cast<half8*>(Bh + o) = hi;
      *reinterpret_cast<half8*>(Bl + o) = lo;
    }
  };

  const int total_chunks = (KS == 3) ? 3 * p.nCB : p.nCB;
  int c_begin = blockIdx.z * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  if (c_end > total_chunks) c_end = total_chunks;
  if (c_begin < c_end) {
    int cur_r = -1;
    auto prep = [&](int c) {
      if (KS == 3) {
        int r = c / p.nCB;
        if (r != cur_r) { set_row(r); cur_r = r; }
      }
      load_chunk(c);
    };
    prep(c_begin);
    store_chunk();
    __syncthreads();
    const unsigned char* a_base = Ah + ((wm * 64 + l31) * ROWL + 16 * h);
    const unsigned char* b_base = Bh + ((wn * 64 + l31) * ROWL + 16 * h);
    for (int c = c_begin; c < c_end; ++c) {
      if (c + 1 < c_end) prep(c + 1);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          ah[t] = *reinterpret_cast<const half8*>(a_base + t * 32 * ROWL + ks * 32);
          al[t] = *reinterpret_cast<const half8*>(a_base + BM * ROWL + t * 32 * ROWL + ks * 32);
          bh[t] = *reinterpret_cast<const half8*>(b_base + t * 32 * ROWL + ks * 32);
          bl[t] = *reinterpret_cast<const half8*>(b_base + BN * ROWL + t * 32 * ROWL + ks * 32);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bh[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tm], bl[tn], acc[tm][tn], 0, 0, 0);
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tm], bh[tn], acc[tm][tn], 0, 0, 0);
          }
      }
      __syncthreads();            // every wave has read chunk c out of LDS
      if (c + 1 < c_end) store_chunk();
      __syncthreads();            // chunk c+1 visible
    }
  }

  const bool direct = p.nsplit == 1;
  float* outp = p.out + (direct ? 0 : (long)blockIdx.z * p.slab_stride);
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    long n = n0 + (wn * 2 + tn) * 32 + l31;
    if (n >= p.Npos) continue;
    int b = (int)(n / p.HW);
    int pos = (int)(n % p.HW);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = m0 + (wm * 2 + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (co < p.Cout) {
          long o = ((long)b * p.Cout + co) * p.HW + pos;
          float v = acc[tm][tn][r] * WSCALE_INV;
          if (direct) {
            if (p.bias) v += p.bias[co];
            if (p.res) v += p.res[o];
          }
          outp[o] = v;
        }
      }
    }
  }
}

// out = sum_z slab[z] + bias + residual
__global__ void splitk_reduce_h2_kernel(const float* __restrict__ slabs, long slab_stride, int nsplit,
                                        const float* __restrict__ bias, const float* __restrict__ res,
                                        float* __restrict__ out, long total, int Cout, int HW) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v = slabs[i];
    for (int z = 1; z < nsplit; ++z) v += slabs[z * slab_stride + i];
    if (bias) v += bias[(i / HW) % Cout];
    if (res) v += res[i];
    out[i] = v;
  }
}

// Packed split weights  Wp[chunk][co][hl][KC]  (f16, scaled by 2^6):
//   3x3: chunk = r * nCB + cb (nCB = ceil(Cin/16)), k = s*16 + c  <-  w[co][cb*16 + c][r][s]
//   1x1: chunk = cb (nCB = ceil(Cin/64)),            k = c         <-  w[co][cb*64 + c]
__global__ void pack_weight_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int Cout, int Cin,
                                      int ks, int nCB, long total) {
  const int KC = ks == 3 ? 48 : 64;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int k = (int)(i % KC);
    long t = i / KC;
    int co = (int)(t % Cout);
    int chunk = (int)(t / Cout);
    float v = 0.0f;
    if (ks == 3) {
      int r = chunk / nCB, cb = chunk % nCB;
      int s = k / 16, c = cb * 16 + (k % 16);
      if (c < Cin) v = w[((long)co * Cin + c) * 9 + r * 3 + s];
    } else {
      int c = chunk * 64 + k;
      if (c < Cin) v = w[(long)co * Cin + c];
    }
    v *= WSCALE;
    _Float16 hi = (_Float16)v;
    _Float16 lo = (_Float16)(v - (float)hi);
    long base = ((long)chunk * Cout + co) * 2 * KC;
    wp[base + k] = hi;
    wp[base + KC + k] = lo;
  }
}

// ------------------------------------------------------------------------------------------------ host side
static inline int h2_kc(int ks) { return ks == 3 ? 48 : 64; }
static inline int h2_ncb(int Cin, int ks) { return ks == 3 ? (Cin + 15) / 16 : (Cin + 63) / 64; }
static inline int h2_chunks(int Cin, int ks) { return (ks == 3 ? 3 : 1) * h2_ncb(Cin, ks); }

extern "C" long i2r_conv2d_packed_h2_bytes(int Cout, int Cin, int ksize) {
  return (long)h2_chunks(Cin, ksize) * Cout * 2 * h2_kc(ksize) * 2;
}

extern "C" int i2r_conv2d_pack_weight_h2(const float* w, void* wp, int Cout, int Cin, int ksize,
                                         i2r_stream_t stream) {
  I2R_REQUIRE(w && wp && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "pack_weight_h2: bad argument");
  long total = (long)h2_chunks(Cin, ksize) * Cout * h2_kc(ksize);
  int nb = i2r_cdiv(total, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(pack_weight_h2_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)wp, Cout, Cin,
                     ksize, h2_ncb(Cin, ksize), total);
  return i2r_check_launch("pack_weight_h2");
}

static void h2_plan(long Npos, int Cout, int chunks, bool allow_split, int& nsplit, int& cps) {
  long tiles = (long)i2r_cdiv(Npos, 128) * i2r_cdiv(Cout, 128);
  nsplit = 1;
  if (allow_split && tiles < 256) {
    long want = (512 + tiles - 1) / tiles;
    long maxs = chunks / 3;
    if (want > maxs) want = maxs;
    if (want > 32) want = 32;
    if (want > 1) nsplit = (int)want;
  }
  cps = (chunks + nsplit - 1) / nsplit;
  nsplit = (chunks + cps - 1) / cps;
}

extern "C" long i2r_conv2d_splitk_h2_floats(int B, int Cin, int H, int W, int Cout, int ksize) {
  int nsplit, cps;
  long Npos = (long)B * H * W;
  h2_plan(Npos, Cout, h2_chunks(Cin, ksize), true, nsplit, cps);
  return nsplit > 1 ? (long)nsplit * Npos * Cout : 0;
}

extern "C" int i2r_conv2d_fwd_h2(const float* x, const void* wp, const float* bias, const float* residual,
                                 float* out, float* splitk_ws, long splitk_floats, int B, int Cin, int H, int W,
                                 int Cout, int ksize, i2r_stream_t stream) {
  I2R_REQUIRE(x && wp && out, "conv2d_fwd_h2: null pointer");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_fwd_h2: kernel size %d unsupported (1 or 3)", ksize);
  I2R_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d_fwd_h2: non-positive dimension");
  long x_bytes = (long)B * Cin * H * W * 4;
  long wp_bytes = i2r_conv2d_packed_h2_bytes(Cout, Cin, ksize);
  I2R_REQUIRE(x_bytes < 0x7FFFFFFFL && wp_bytes < 0x7FFFFFFFL,
              "conv2d_fwd_h2: input (%ld B) or packed weights (%ld B) exceed the 2 GiB buffer-addressing limit",
              x_bytes, wp_bytes);
  long Npos = (long)B * H * W;
  int chunks = h2_chunks(Cin, ksize), nsplit, cps;
  h2_plan(Npos, Cout, chunks, splitk_ws != nullptr, nsplit, cps);
  if (nsplit > 1 && (long)nsplit * Npos * Cout > splitk_floats) { nsplit = 1; cps = chunks; }
  ConvHParams p;
  p.x = x; p.wp = wp; p.bias = bias; p.res = residual;
  p.out = nsplit > 1 ? splitk_ws : out;
  p.x_bytes = (unsigned)x_bytes; p.wp_bytes = (unsigned)wp_bytes;
  p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.HW = H * W; p.nCB = h2_ncb(Cin, ksize);
  p.Npos = Npos; p.chunks_per_split = cps; p.nsplit = nsplit; p.slab_stride = Npos * Cout;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(i2r_cdiv(Npos, 128), i2r_cdiv(Cout, 128), nsplit);
  static bool attr_set = false;
  const int lds3 = 4 * 128 * (48 * 2 + 16), lds1 = 4 * 128 * (64 * 2 + 16);
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_fwd_h2_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds3);
    (void)hipFuncSetAttribute((const void*)conv_fwd_h2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    attr_set = true;
  }
  if (ksize == 3)
    hipLaunchKernelGGL(conv_fwd_h2_kernel<3>, grid, dim3(256), lds3, s, p);
  else
    hipLaunchKernelGGL(conv_fwd_h2_kernel<1>, grid, dim3(256), lds1, s, p);
  if (nsplit > 1) {
    long total = Npos * Cout;
    int nb = i2r_cdiv(total, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(splitk_reduce_h2_kernel, dim3(nb), dim3(256), 0, s, splitk_ws, p.slab_stride, nsplit, bias,
                       residual, out, total, Cout, H * W);
  }
  return i2r_check_launch("conv2d_fwd_h2");
}
