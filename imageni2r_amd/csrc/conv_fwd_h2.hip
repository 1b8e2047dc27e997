// Convolution forward with fp32-grade accuracy on the gfx950 *half-precision* matrix cores ("f16x2 split").
//
// Every fp32 operand x is written as x = hi + lo with hi = f16(x), lo = f16(x - hi): 22 significand bits, and the product
// of two fp16 numbers is exact in the MFMA's fp32 accumulator.  Then
//        a*b = a_hi*b_hi + a_hi*b_lo + a_lo*b_hi     (+ a_lo*b_lo ~ 2^-22 |ab|, dropped)
// i.e. THREE v_mfma_f32_32x32x16_f16 (32 cycles each for a 32x32x16 tile-step) replace EIGHT v_mfma_f32_32x32x2_f32
// (64 cycles each): 5.3x the matrix throughput of the exact-fp32 kernel (conv_fwd.hip) at ~1e-6 relative error,
// far inside the 1e-4 bar of the path.  Used for inference (the sampler's 35 denoiser evaluations per series) and for
// the larger training forwards; the backward keeps the exact fp32 kernels because gradients can underflow fp16's range.
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
#include <type_traits>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

#define OOR 0x80000000u
#define WSCALE 64.0f
#define WSCALE_INV 0.015625f

struct ConvHParams {
  const float* x;       // activations (B, Cin, H, W) fp32 -- or the first Cin1 channels of a virtual channel concat
  const float* x2;      // 1x1 only: the remaining Cin - Cin1 channels (B, Cin - Cin1, H, W), or null
  const void* wp;       // packed split weights: [chunk][Cout][2][KC] f16
  const float* bias;
  const float* res;
  float* out;
  unsigned x_bytes, x2_bytes, wp_bytes;
  int B, Cin, Cin1, H, W, Cout, HW, nCB;
  long Npos;
  int chunks_per_split, nsplit, mtiles;
  long slab_stride;
  int ablate;  // tuning only (I2R_H2_ABLATE): 1 = skip the epilogue stores, 2 = skip the main loop
  int up_in;   // 1 (3x3 DPP path): x is (B, Cin, H/2, W/2) and is nearest-upsampled x2 on the fly (Conv2d up=True, :79-81)
  int up_res;  // 1: the residual is (B, Cout, H/2, W/2), nearest-upsampled x2 on the fly (the kernel-0 `skip` of up blocks)
  const float* in_scale;  // null, or device float[2] = {s, 1/s}, s a power of two (i2r_absmax_pow2_scale_f32): un-split
                          // inputs are multiplied by s before the f16 split and the result by 1/s -- both exact -- so
                          // that tensors far below fp16's normal range (gradients: the data-gradient use of this
                          // kernel, i2r_conv2d_dgrad_h2) keep their 22 significand bits
  int hw1_vec; // 1 (1x1 conv on 1x1 images, 16-byte-aligned rows): the tiled kernel fetches a position's channels with 16-byte loads
  int qkv_out; // 1: the output feeds i2r_attention_fwd_h2(presplit): q channels (c % 3 == 0) are scaled by I2R_QSCALE and
               // every element is stored as the split pair f16 hi | f16 lo << 16 (the attention kernel then only permutes)
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
    _Float16 h, l;
    f16_split(v[j], h, l);
    hi[j] = h;
    lo[j] = l;
  }
}

// the same on s * v (s a power of two: exact)
__device__ __forceinline__ void split8s(const float* v, float s, half8& hi, half8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 h, l;
    f16_split(v[j] * s, h, l);
    hi[j] = h;
    lo[j] = l;
  }
}

// pre-split activations: one 32-bit word per element = f16 hi | f16 lo << 16 (written by the producing kernel, e.g.
// GroupNorm with split_out); 8 words -> the hi and lo fragments with byte permutes only.
__device__ __forceinline__ void unpack8(const float* v, half8& hi, half8& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned w0 = __builtin_bit_cast(unsigned, v[2 * i]), w1 = __builtin_bit_cast(unsigned, v[2 * i + 1]);
    h[i] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
    l[i] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
  }
  uint4v hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
  hi = __builtin_bit_cast(half8, hv);
  lo = __builtin_bit_cast(half8, lv);
}

// Epilogue of both kernels: lane = (position, channel half h), acc[t][r] = channel m0 + 32t + (r&3) + 8(r>>2) + 4h.
// Buffer addressing (lane voffset + per-channel scalar offset) keeps it at one address register.  Every load (bias,
// residual) is issued BEFORE the first store: the compiler must assume the output may alias them, so a load placed
// after a store waits for it -- interleaved, the 64 elements become 64 serial load -> wait -> store round trips
// (measured: 25 us per workgroup, more than the whole main loop of a 1x1 conv).
template <int NTL>
__device__ __forceinline__ void h2_epilogue(const ConvHParams& p, const f32x16 (&acc)[NTL], float* outp, bool full,
                                            bool nvalid, unsigned pb, unsigned pos, int m0, int h, int ablate,
                                            float omul) {
  const unsigned out_bytes = (unsigned)(p.Npos * p.Cout * 4);
  __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, (int)out_bytes, 0x00020000);
  const unsigned plane = (unsigned)p.HW * 4;
  const unsigned voff = nvalid ? ((pb * (unsigned)p.Cout + (unsigned)(m0 + 4 * h)) * (unsigned)p.HW + pos) * 4u : OOR;
  // residual: same element, or (up_res) its nearest-x2 source in a half-resolution tensor
  unsigned rplane = plane, rvoff = voff, res_bytes = out_bytes;
  if (p.up_res) {
    const unsigned y = pos / (unsigned)p.W, x = pos - y * (unsigned)p.W, hwl = (unsigned)p.HW >> 2;
    rplane = hwl * 4;
    res_bytes = out_bytes >> 2;
    rvoff = nvalid ? ((pb * (unsigned)p.Cout + (unsigned)(m0 + 4 * h)) * hwl + (y >> 1) * ((unsigned)p.W >> 1) + (x >> 1)) * 4u
                   : OOR;
  }
  const int lim = p.Cout - m0 - 4 * h;  // channel offsets below lim exist
  const bool use_bias = full && p.bias != nullptr, use_res = full && p.res != nullptr;
  __amdgpu_buffer_rsrc_t rR =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, use_res ? (int)res_bytes : 0, 0x00020000);
  // bias (and residual, above) through a buffer descriptor of size 0 when there is none (or on split-K partials): every lane-level
  // condition is an out-of-range offset instead of a branch around a load (the branchy form cost ~7 us per workgroup)
  __amdgpu_buffer_rsrc_t rBias =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, use_bias ? p.Cout * 4 : 0, 0x00020000);
  const unsigned boff = (unsigned)(m0 + 4 * h) * 4u;
  const int cmod = (m0 + 4 * h) % 3;  // (head, d, {q,k,v}) interleave: channel % 3 == 0 is a q row
  float bv[NTL][16], rv[NTL][16];
#pragma unroll
  for (int t = 0; t < NTL; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cof = t * 32 + (r & 3) + 8 * (r >> 2);
      const bool ok = cof < lim;
      bv[t][r] = bufload(rBias, ok ? boff : OOR, (unsigned)cof * 4u);  // branch-free: out of range reads 0
      rv[t][r] = bufload(rR, ok ? rvoff : OOR, (unsigned)cof * rplane);
    }
  auto store_all = [&](auto qkv) {  // one straight-line run of 64 stores per output form (no per-element branch)
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cof = t * 32 + (r & 3) + 8 * (r >> 2);
        float v = acc[t][r] * omul + bv[t][r] + rv[t][r];
        if (decltype(qkv)::value) {
          if ((cmod + cof) % 3 == 0) v *= I2R_QSCALE;
          v = split_pack(v);
        }
        unsigned vo = (cof < lim && (!(ablate & 1) || v == 12345.678f)) ? voff : OOR;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, vo, (unsigned)cof * plane, 0);
      }
  };
  if (p.qkv_out) store_all(std::true_type{});
  else store_all(std::false_type{});
}

// Same epilogue with 16-byte global accesses.  The vector-memory INSTRUCTION count, not bytes, bounds the short-K
// convs: the lane-per-position form above issues 64 dword stores + 64 bias + 64 residual loads per tile and lane; a
// workgroup that does nothing else still takes ~6 us (measured with the main loop and the stores compiled out), on a
// par with the whole MFMA loop of a 128-channel 1x1 conv.  Here each 32-channel x 32-position accumulator tile
// is transposed through a private 4.5 KB LDS patch (the weight stages are idle by now) so that a lane owns FOUR
// consecutive positions of one channel: 16 b128 stores + 16 b128 residual loads + 16 bias loads per tile.
// Needs W % 4 == 0 (a position quad inside one image row).  `scratch`: this wave's 32 x 36 floats.
#define EPI_PITCH 36
// The quad epilogue's 16 bias values of a lane (channel offsets 8 q + (lane >> 3) of the tile at m0).
template <int NQ>
__device__ __forceinline__ void h2_quad_bias(const ConvHParams& p, bool full, int m0, int lane, float (&bv)[NQ]) {
  const int cr = lane >> 3, lim = p.Cout - m0 - cr;
  const bool use_bias = full && p.bias != nullptr;
  // (readfirstlane: a uniform select that the compiler lowers to v_cndmask puts the descriptor into vector registers,
  // and every load through it then becomes a waterfall loop -- seen in the 16x16x32 kernels' epilogues, 49 of them)
  __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.bias, 0, __builtin_amdgcn_readfirstlane(use_bias ? p.Cout * 4 : 0), 0x00020000);
  const unsigned boff = (unsigned)(m0 + cr) * 4u;
#pragma unroll
  for (int q = 0; q < NQ; ++q) bv[q] = bufload(rBias, 8 * q < lim ? boff : OOR, (unsigned)(8 * q) * 4u);
}

// bias_pre: the lane's h2_quad_bias values when the caller already holds them (a position loop loads them once).
// RES = false: the caller knows there is no residual -- not even the (null-descriptor) loads are issued: they would sit
// behind the position loop's prefetches in the in-order vmcnt queue and make the epilogue wait for them.
// `put(t, scratch)`: writes the wave's 32-channel x 32-position accumulator block t into the patch as
// scratch[channel * EPI_PITCH + position] -- the only part that depends on the MFMA shape's accumulator layout.
template <int NTL, bool RES, typename PUT>
__device__ __forceinline__ void h2_epilogue_quad_put(const ConvHParams& p, PUT put, float* outp, bool full,
                                                     long nw, int m0, int lane, float* scratch, int ablate, float omul,
                                                     const float* bias_pre = nullptr, const float* bias_lds = nullptr) {
  const int cr = lane >> 3, pq = (lane & 7) * 4;
  const long nq = nw + pq;
  const bool qvalid = nq < p.Npos;
  unsigned pb = 0, pos = 0;
  if (qvalid) {
    pb = (unsigned)(nq / p.HW);
    pos = (unsigned)(nq - (long)pb * p.HW);
  }
  const unsigned out_bytes = (unsigned)(p.Npos * p.Cout * 4);
  const bool use_res = full && p.res != nullptr;
  const unsigned plane = (unsigned)p.HW * 4;
  const unsigned crow = pb * (unsigned)p.Cout + (unsigned)(m0 + cr);  // (sample, channel) plane index of this lane
  const unsigned voff = qvalid ? (crow * (unsigned)p.HW + pos) * 4u : OOR;
  unsigned rplane = plane, rvoff = voff, res_bytes = out_bytes;
  if (p.up_res) {  // half-resolution residual: the quad's sources are pixels (x0/2, x0/2 + 1) of row y/2
    const unsigned y = pos / (unsigned)p.W, x = pos - y * (unsigned)p.W, hwl = (unsigned)p.HW >> 2;
    rplane = hwl * 4;
    res_bytes = out_bytes >> 2;
    rvoff = qvalid ? (crow * hwl + (y >> 1) * ((unsigned)p.W >> 1) + (x >> 1)) * 4u : OOR;
  }
  rplane = __builtin_amdgcn_readfirstlane(rplane);   // (scalar offset of every residual load: see h2_quad_bias)
  __amdgpu_buffer_rsrc_t rO =
      __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, __builtin_amdgcn_readfirstlane((int)out_bytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.res, 0, __builtin_amdgcn_readfirstlane(use_res ? (int)res_bytes : 0), 0x00020000);
  const int lim = p.Cout - m0 - cr;  // channel offsets below lim exist for this lane
  const int cmod = (m0 + cr) % 3;
  // every load before the first store (see the aliasing note above)
  constexpr int NQ = 4 * NTL;
  float bv[NQ], rv[NQ][4];
  if (bias_pre) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) bv[q] = bias_pre[q];
  } else if (bias_lds) {   // the tile's bias values (zeros where there is none) staged in LDS by the caller
#pragma unroll
    for (int q = 0; q < NQ; ++q) bv[q] = bias_lds[8 * q + cr];
  } else {
    h2_quad_bias<NQ>(p, full, m0, lane, bv);
  }
  if (!RES || !use_res) {   // (uniform branch: without a residual not even the null-descriptor loads are issued)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) rv[q][e] = 0.0f;  // (still added below: x + 0 keeps the bits of the loaded-zero form)
  } else if (p.up_res) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {  // tile t = q / 4 covers channel offsets 32 t + 8 i + cr = 8 q + cr
      const float a = bufload(rR, 8 * q < lim ? rvoff : OOR, (unsigned)(8 * q) * rplane);
      const float b = bufload(rR, 8 * q < lim ? rvoff + 4u : OOR, (unsigned)(8 * q) * rplane);
      rv[q][0] = a; rv[q][1] = a; rv[q][2] = b; rv[q][3] = b;
    }
  } else {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const uint4v t4 = bufload4(rR, 8 * q < lim ? rvoff : OOR, (unsigned)(8 * q) * rplane);
      // (scalar copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this hipcc)
      const unsigned u0 = t4.x, u1 = t4.y, u2 = t4.z, u3 = t4.w;
      rv[q][0] = __builtin_bit_cast(float, u0); rv[q][1] = __builtin_bit_cast(float, u1);
      rv[q][2] = __builtin_bit_cast(float, u2); rv[q][3] = __builtin_bit_cast(float, u3);
    }
  }
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    put(t, scratch);
    // the four reads of a block go out together (one LDS round trip per block instead of four in a row)
    f32x4 a4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const f32x4*>(scratch + (8 * i + cr) * EPI_PITCH + pq);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = 4 * t + i, cof = 8 * q;
      float v[4] = {a4[i][0], a4[i][1], a4[i][2], a4[i][3]};
      uint4v o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float y = v[e] * omul + bv[q] + rv[q][e];
        if (p.qkv_out) {
          if ((cmod + cof) % 3 == 0) y *= I2R_QSCALE;
          y = split_pack(y);
        }
        o[e] = __builtin_bit_cast(unsigned, y);
      }
      // The channel offset goes into the VECTOR offset, the scalar offset stays the constant 0.  A 16-byte store whose
      // data registers are overwritten by the very next VALU instruction needs one wait state in between; the compiler
      // inserts it only for stores WITHOUT a register in the scalar-offset field (the ISA manuals exempt those), and on
      // gfx950 the exemption does not hold: with `cof * plane` as an SGPR offset and a v_mov into o[0] scheduled right
      // behind the store, 16 lanes of a wave stored the NEW value of o[0] (seen in conv3x3_h2_m16_kernel's in-loop
      // epilogue: raw accumulators of the next channel block at element 0 of a quad, timing dependent).
      unsigned vo = (cof < lim && (!(ablate & 1) || o[0] == 0x12345678u)) ? voff + (unsigned)cof * plane : OOR;
      __builtin_amdgcn_raw_buffer_store_b128(o, rO, vo, 0, 0);
    }
  }
}

// 32x32x16 accumulators: acc[t][r] = channel 32 t + (r & 3) + 8 (r >> 2) + 4 h, position l31
template <int NTL, bool RES = true>
__device__ __forceinline__ void h2_epilogue_quad(const ConvHParams& p, const f32x16 (&acc)[NTL], float* outp, bool full,
                                                 long nw, int m0, int lane, float* scratch, int ablate, float omul,
                                                 const float* bias_pre = nullptr) {
  const int l31 = lane & 31, h = lane >> 5;
  auto put = [&](int t, float* sc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * h) * EPI_PITCH + l31] = acc[t][r];
  };
  h2_epilogue_quad_put<NTL, RES>(p, put, outp, full, nw, m0, lane, scratch, ablate, omul, bias_pre);
}

// value of lane-1 / lane+1 (0 into the wave's first / last lane): DPP wave shifts, no LDS, no memory
__device__ __forceinline__ float from_lane_below(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_lane_above(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// KS = 3: chunk = (kernel row r, 16-channel block), KC = 48 (k = s*16 + c).   KS = 1: chunk = 64 channels, KC = 64.
// Workgroup tile: 128 output channels x 32*NW positions.  The NW wavefronts split the POSITIONS (32 each) and every
// wave covers all 128 channels: the weight tile is shared through LDS (double buffered, one barrier per chunk), while
// each lane loads exactly the activation elements its own MFMAs consume -- straight from global memory into
// registers, no LDS round trip, no cross-wave traffic.
//   SPLIT: activations arrive pre-split (see unpack8).
//   DPP (3x3, image width a power of two <= 32): only the centre tap column is loaded; the left / right taps are the
//   neighbouring lanes' values (lane = position along the image row), fetched with DPP wave shifts and zeroed at the
//   row ends -- a third of the vector-memory instructions and L1/L2 bytes of the plain im2col gather.
// TERMS = 3: the full split product.  TERMS = 1: a_hi*b_hi only -- plain f16 operands, fp32 accumulation (~5e-4 per
// layer): the optional reduced-precision inference path (set_conv_math("f16")).
// BM: rows (output channels) of the workgroup's tile -- 128, or 64 / 32 for layers with that few output channels (the
// 64-channel net's convs, every net's output conv): the MFMAs of the empty rows are not issued.
template <int KS, bool SPLIT, int NW, bool DPP, int TERMS = 3, int BM = 128>
__global__ __launch_bounds__(64 * NW, 2) void conv_fwd_h2_kernel(ConvHParams p) {
  constexpr int BN = 32 * NW, NT = 64 * NW, NTL = BM / 32;
  constexpr int KC = (KS == 3) ? 48 : 64;
  constexpr int NKS = KC / 16;             // MFMA k-steps per chunk
  constexpr int ROWL = KC * 2 + 16;        // LDS row bytes (odd multiple of 16 B -> conflict-free b128 rows)
  constexpr int RB = 2 * KC * 2;           // global bytes per packed weight row (hi | lo)
  constexpr int PPR = RB / 16;             // 16-B pieces per weight row
  constexpr int PPRL = (TERMS == 3) ? PPR : PPR / 2;  // pieces of a row that are fetched (single-term: the hi half only)
  constexpr int NA = (BM * PPRL + NT - 1) / NT;       // weight pieces per thread per chunk
  constexpr int ABUF = 2 * BM * ROWL;      // bytes of one weight stage (hi rows, then lo rows)
  constexpr int NRAW = (KS == 3 && DPP) ? 1 : NKS;  // raw activation register sets per chunk
  static_assert(BM == 128 || BM == 64 || BM == 32, "Cout tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (id % 8), each with its own L2.  All the
  // output-channel tiles of one position tile get ids 8 apart -> same XCD, dispatched back to back, so the
  // activation tile they share is fetched from HBM once and then served by that XCD's L2.  (Speed only: any
  // placement computes the same result.)
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int mt = q % p.mtiles;
  const long nt = (long)(q / p.mtiles) * 8 + xcd;
  const int m0 = mt * BM;
  const long n0 = nt * BN;
  if (n0 >= p.Npos) return;  // padding of the position-tile count to a multiple of 8 (whole workgroup exits)

  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);

  // ---- weight-tile descriptors (loop invariant)
  unsigned voffA[NA];
  int ldsA[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int idx = tid + NT * j;
    const bool used = idx < BM * PPRL;  // (single-term 8 waves, short tiles): a spare slot loads
    int row = idx / PPRL, piece = idx % PPRL;  // nothing (out of range) and parks its zeros in a row's padding
    voffA[j] = (used && m0 + row < p.Cout) ? (unsigned)((m0 + row) * RB + piece * 16) : OOR;
    int lo = piece >= PPR / 2;
    ldsA[j] = used ? (lo ? BM * ROWL : 0) + row * ROWL + (piece - lo * (PPR / 2)) * 16 : (idx % BM) * ROWL + KC * 2;
  }
  const unsigned stepA = (unsigned)p.Cout * RB;  // bytes per chunk of packed weights

  // ---- im2col descriptors: lane = (position n0 + 32*wave + l31, channel half h)
  const long nj = n0 + wave * 32 + l31;
  const bool nvalid = nj < p.Npos;
  int pb = 0, py = 0, px = 0;
  if (nvalid) {
    pb = (int)(nj / p.HW);
    int pos = (int)(nj % p.HW);
    py = pos / p.W;
    px = pos % p.W;
  }
  const bool has_left = px > 0, has_right = px < p.W - 1;
  const int us = p.up_in;  // 0 / 1: source rows, columns and planes are halved when the input is upsampled on the fly
  const unsigned chan_bytes = (unsigned)(p.HW >> (2 * us)) * 4;
  unsigned voffB[3];
  auto set_row = [&](int r) {  // voffsets of the tap columns of kernel row r for this lane's position
    int iy = py + r - 1;
    bool rok = nvalid && (unsigned)iy < (unsigned)p.H;
    long off = ((long)pb * p.Cin + 8 * h) * p.HW + iy * p.W + px;
    if (DPP && us)  // nearest x2: tap (iy, px) of the upsampled image is source pixel (iy >> 1, px >> 1); the DPP
                    // neighbours (px -+ 1) then hold exactly the left / right taps of the upsampled row
      off = ((long)pb * p.Cin + 8 * h) * (p.HW >> 2) + (iy >> 1) * (p.W >> 1) + (px >> 1);
    if (DPP) {
      voffB[0] = rok ? (unsigned)(off * 4) : OOR;
    } else {
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        int ix = px + s - 1;
        voffB[s] = (rok && (unsigned)ix < (unsigned)p.W) ? (unsigned)((off + s - 1) * 4) : OOR;
      }
    }
  };
  // 1x1 over torch.cat([x, x2], dim=1) without materialising it (models/networks.py:458): channel blocks below Cin1
  // come from x, the rest from x2 (Cin1 is a multiple of 16, so a k-step never straddles the two).
  __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, (int)p.x2_bytes, 0x00020000);
  unsigned voffB2 = OOR;
  if (KS == 1) {
    long off = ((long)pb * p.Cin1 + 8 * h) * p.HW + py * p.W + px;
    voffB[0] = nvalid ? (unsigned)(off * 4) : OOR;
    long off2 = ((long)pb * (p.Cin - p.Cin1) + 8 * h) * p.HW + py * p.W + px;
    voffB2 = (nvalid && p.x2) ? (unsigned)(off2 * 4) : OOR;
  }

  f32x16 acc[NTL];
#pragma unroll
  for (int i = 0; i < NTL; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  uint4v ra[NA];
  float rb[NRAW][8];
  // (1x1 convs on 1x1 images with 16-byte-aligned operands and Cin splits: see load_b)
  const bool hw1 = KS == 1 && p.HW == 1 && p.hw1_vec;
  // live = false (a chunk past this workgroup's range): the scalar offset is pushed out of the buffer, so the loads
  // return 0 without touching memory -- keeps the main loop branch-free
  auto load_a = [&](int c, bool live) {
    unsigned sA = live ? (unsigned)c * stepA : OOR;
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = bufload4(rA, voffA[j], sA);
  };
  auto store_a = [&](int buf) {
    unsigned char* dst = smem + buf * ABUF;
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<uint4v*>(dst + ldsA[j]) = ra[j];
  };
  // Channels past Cin in the last channel block need no masking: their packed weights are zero, and what the load
  // returns there is either a finite activation of the next sample or the buffer unit's out-of-range 0.
  auto load_b = [&](int c, bool live) {
    const unsigned dead = live ? 0u : OOR;
    if (KS == 3) {
      int cb = c % p.nCB;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned sB = (unsigned)(cb * 16 + j) * chan_bytes + dead;
#pragma unroll
        for (int s = 0; s < NRAW; ++s) rb[s][j] = bufload(rB, voffB[s], sB);
      }
    } else if (hw1) {
      // 1x1 images: a position's channels are CONSECUTIVE floats, so the lane's eight channels of a k-step are two
      // 16-byte loads instead of eight dword loads that each touch 64 different cache lines across the wave (the
      // bottom level of the 8x8-image nets: 40 us per launch for 6 MB of operands)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cbase = c * 64 + q * 16;
        const bool first = cbase < p.Cin1;  // wave-uniform
        const unsigned sB = (unsigned)(first ? cbase : cbase - p.Cin1) * 4u + dead;
        const uint4v lo4 = bufload4(first ? rB : rB2, first ? voffB[0] : voffB2, sB);
        const uint4v hi4 = bufload4(first ? rB : rB2, first ? voffB[0] : voffB2, sB + 16u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          rb[q][j] = __builtin_bit_cast(float, (unsigned)lo4[j]);
          rb[q][4 + j] = __builtin_bit_cast(float, (unsigned)hi4[j]);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int cbase = c * 64 + q * 16;
          const bool first = cbase < p.Cin1;  // wave-uniform
          unsigned sB = (unsigned)((first ? cbase : cbase - p.Cin1) + j) * chan_bytes + dead;
          rb[q][j] = bufload(first ? rB : rB2, first ? voffB[0] : voffB2, sB);  // scalar selects, no branch
        }
    }
  };
  // input scale (wave-uniform; 1 and exact when there is none): only the un-split path multiplies
  float s_in = 1.0f, omul = WSCALE_INV;
  if (!SPLIT && p.in_scale != nullptr) {
    s_in = p.in_scale[0];
    omul = WSCALE_INV * p.in_scale[1];
  }
  auto to_frag = [&](const float* v, half8& hi, half8& lo) {
    if (SPLIT) unpack8(v, hi, lo);
    else split8s(v, s_in, hi, lo);
  };

  const int total_chunks = (KS == 3) ? 3 * p.nCB : p.nCB;
  int c_begin = blockIdx.z * p.chunks_per_split;
  int c_end = c_begin + p.chunks_per_split;
  if (c_end > total_chunks) c_end = total_chunks;
  if (c_begin < c_end && !(p.ablate & 2)) {
    int cur_r = -1;
    auto prep_b = [&](int c) {
      if (KS == 3) {
        int r = c / p.nCB;
        if (r != cur_r) { set_row(r); cur_r = r; }
      }
      load_b(c, c < c_end);
    };
    // fragments of one chunk's activations out of the raw registers
    auto build_frags = [&](half8* fh, half8* fl) {
      if (KS == 3 && DPP) {
        float tl[8], tr[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float lft = from_lane_below(rb[0][j]), rgt = from_lane_above(rb[0][j]);
          tl[j] = has_left ? lft : 0.0f;
          tr[j] = has_right ? rgt : 0.0f;
        }
        to_frag(tl, fh[0], fl[0]);
        to_frag(rb[0], fh[1], fl[1]);
        to_frag(tr, fh[2], fl[2]);
      } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) to_frag(rb[ks], fh[ks], fl[ks]);
      }
    };
    load_a(c_begin, true);
    prep_b(c_begin);
    store_a(0);
    half8 bh[NKS], bl[NKS];
    build_frags(bh, bl);
    load_a(c_begin + 1, c_begin + 1 < c_end);
    prep_b(c_begin + 1);
    __syncthreads();
    // Steady state of chunk c: weight stage c in LDS, activation fragments of c in bh/bl, and the loads of chunk c+1
    // (weights -> ra, activations -> rb) in flight.  The fragment build for c+1 is VALU work placed beside the last
    // k-step's MFMAs of chunk c; the loads of c+2 are issued as soon as ra / rb are free again.
    for (int c = c_begin; c < c_end; ++c) {
      const int buf = (c - c_begin) & 1;
      const unsigned char* a_base = smem + buf * ABUF + (l31 * ROWL + 16 * h);
      half8 af[2][2 * NTL];
      auto load_frags = [&](int slot, int ks) {
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          af[slot][2 * t] = *reinterpret_cast<const half8*>(a_base + t * 32 * ROWL + ks * 32);
          if (TERMS == 3)
            af[slot][2 * t + 1] = *reinterpret_cast<const half8*>(a_base + BM * ROWL + t * 32 * ROWL + ks * 32);
        }
      };
      auto mfma12 = [&](int ks) {
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][2 * t], bh[ks], acc[t], 0, 0, 0);
          if (TERMS == 3) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][2 * t], bl[ks], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][2 * t + 1], bh[ks], acc[t], 0, 0, 0);
          }
        }
      };
      load_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < NKS - 1; ++ks) {
        load_frags((ks + 1) & 1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma12(ks);
        __builtin_amdgcn_sched_barrier(0);
      }
      // last k-step: its 12 MFMAs interleaved with the next chunk's fragment build (VALU only)
      // Everything below is unconditional so that it stays ONE scheduling region: past the last chunk the stores hit
      // the idle LDS stage, the fragment build chews on stale registers and the loads run off the end of the buffers
      // (the buffer unit returns 0) -- all harmless, none of it is read.
      half8 nh[NKS], nl[NKS];
      store_a(buf ^ 1);  // stage c+1 (its last readers, chunk c-1, are behind the previous barrier)
      mfma12(NKS - 1);
      build_frags(nh, nl);
#pragma unroll
      for (int i = 0; i < NTL * TERMS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, (TERMS == 3 ? 6 : 12) * 4 / NTL, 0);  // VALU of the fragment build
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) { bh[ks] = nh[ks]; bl[ks] = nl[ks]; }
      load_a(c + 2, c + 2 < c_end);
      prep_b(c + 2);
      __syncthreads();
    }
  }

  const bool direct = p.nsplit == 1;
  float* outp = p.out + (direct ? 0 : (long)blockIdx.z * p.slab_stride);
  if ((p.W & 3) == 0 && !(p.ablate & 8))  // quad form: its LDS patch reuses the (now idle) weight stages
    h2_epilogue_quad<NTL>(p, acc, outp, direct, n0 + wave * 32, m0, lane,
                     reinterpret_cast<float*>(smem) + wave * (32 * EPI_PITCH), p.ablate, omul);
  else
    h2_epilogue<NTL>(p, acc, outp, direct, nvalid, (unsigned)pb, nvalid ? (unsigned)(nj % p.HW) : 0u, m0, h, p.ablate,
                omul);
}

// ---------------------------------------------------------------------------------------------------------------
// The large 3x3 layers on v_mfma_f32_16x16x32_f16.  Measured (tools/mfma_shape_probe.hip: this kernel's MFMA pattern as
// a bare loop on random operands in registers): 557 TF algorithmic with 32x32x16, 662 TF with 16x16x32 -- on all-zero
// operands both reach the nominal 2.47 PF, so the difference is the clock the chip holds under matrix load, and the
// 32x32x16 kernel above already sits at ~0.7 of ITS ceiling.  Same algorithm and the same packed weights as
// conv_fwd_h2_kernel<3, SPLIT, 8, true>; what changes with the instruction shape:
//   * a k-step is 32 input channels at one tap, a chunk = (kernel row r, 32-channel block): 3 k-steps, 144 MFMAs per
//     wave and barrier.  The weight stage is filled from TWO consecutive 16-channel chunks of the packed layout (a
//     fragment's 8 consecutive k are 8 consecutive channels of one of them), so no second weight layout exists.
//   * lane = (position pair n16 = lane & 15, channel group kg = lane >> 4): lane n16 owns positions 2 n16 and
//     2 n16 + 1 of the wave's 32; column tile 0 = the even, column tile 1 = the odd positions.  One 8-byte load per
//     channel brings both (for an input upsampled on the fly one dword IS both); its 128 channels are eight 16-row
//     tiles; accumulators acc[8][2] of 4 floats.
//   * the centre tap is converted to hi / lo fragments ONCE; the left tap of an odd position and the right tap of an
//     even one are the OTHER column tile's centre fragment as it stands, the remaining two are DPP ROW shifts of the
//     packed fragment registers (two channels per register): per 32 channels and lane 8 loads, 16 permutes and 16 DPP
//     moves (the kernel above: 16 loads, 48 permutes, 32 DPP moves + 32 selects; one position per lane and tile in
//     this kernel: 16 loads, 32 DPP moves -- the pair form measured +3 to +10 % on every probe shape).  Image rows of
//     32 need no border select at all (the row shift's zero fill IS the border); narrower rows mask two taps.
// Taken for W in {2, 4, 8, 16, 32}, Cin % 32 == 0, Cout > 64, >= 256 tiles of 256 positions x 128 channels, never
// split-K (one 8-wave workgroup per CU: 250 registers per lane; two weight stages are 96 KB of LDS).  WM: 2 = image rows of 32, 1 = rows of 2 (2x2 images), 0 = the widths between.
#ifndef M16_ABL
#define M16_ABL 0  /* tuning builds only (tools/r3_m16_ablate.sh): 1 no fragment reads from LDS, 2 no fragment build,
                      4 no weight staging, 8 no global loads in the loop, 16 no barrier -- results are then garbage */
#endif
template <int CTRL, bool ZERO_FILL>
__device__ __forceinline__ unsigned row_dpp(unsigned fill, unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, CTRL, 0xf, 0xf, ZERO_FILL);
}
#define DPP_ROW_SHR1 0x111  /* lane i <- lane i-1 of its 16-lane row */
#define DPP_ROW_SHL1 0x101  /* lane i <- lane i+1 */
#define DPP_ROW_ROR1 0x121  /* lane i <- lane (i-1) mod 16 */
#define DPP_ROW_ROR15 0x12F /* lane i <- lane (i+1) mod 16 */

template <bool SPLIT, int WM, bool UP>
__global__ __launch_bounds__(512, 2) void conv3x3_h2_m16_kernel(ConvHParams p) {
  constexpr int BM = 128, NT = 512, BN = 256;
  constexpr int KC = 96;                // halves per weight row and stage half: 3 taps x 32 channels
  // LDS rows are the bare 192 bytes (12 pieces of 16).  A fragment read has lanes (row n16, piece kg) and
  // ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... of 16 over 64 banks
  // (MI355X_MICROARCH.md, LDS): rows {0-3, 12-15} at one kg beside rows {4-11} at the next.  With a padded (odd)
  // pitch those sixteen 16-byte pieces fall on eleven bank quads (measured: the fragment reads were 14 % of the
  // kernel); swapping the piece pairs (0,1) <-> (2,3) in rows 8-15 of every 16 makes them sixteen distinct ones.
  constexpr int ROWL = KC * 2;          // LDS row pitch in bytes
  constexpr int ABUF = 2 * BM * ROWL;   // one weight stage: hi rows, then lo rows
  constexpr int NA = BM * 24 / NT;      // 16-byte weight pieces per thread and chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // (the wave index through readfirstlane: the stream counters below are advanced under `if (grp)`, and they must stay
  // scalar for the compiler -- a counter it takes for divergent turns every buffer load into a waterfall loop)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  // PERSISTENT, and ONE software pipeline over all the chunks of all the workgroup's tiles.  One workgroup per CU
  // walks its tiles in the XCD-aware order of conv_fwd_h2_kernel: the 32 workgroups of an XCD (ids 8 apart) take 32
  // consecutive entries of that XCD's tile list (output-channel tile fastest), so the Cout tiles of a position tile run
  // side by side on one L2.  Per-shape rates of the first persistent form (a prologue, the chunk loop and an epilogue
  // per tile) fitted time = chunks + 4.9 chunk-times per tile at a main-loop rate of 551 TF -- the MFMA-only ceiling
  // of the ablation -- i.e. ALL of the remaining loss was per-tile: exposed load latency in the prologue and the
  // epilogue and the two half-phases of skew between the wave groups, 29 % of a 128-channel layer's time.  Now the
  // weight and activation load streams run ahead ACROSS tile boundaries (each stream has its own tile / chunk counter
  // and recomputes its lane offsets when it wraps), the chunk loop never drains, and a tile boundary is just a staging
  // phase that also holds the tile's epilogue -- beside the other group's MFMA phase.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const long npt = (p.Npos + BN - 1) / BN;                         // position tiles in all
  const long lim = (npt > xcd ? (npt - xcd + 7) / 8 : 0) * p.mtiles;  // entries of this XCD's tile list
  const int ntile = lim > slot ? (int)((lim - slot + nslots - 1) / nslots) : 0;
  if (ntile == 0) return;   // (whole workgroup, before any barrier)
  const int nCB32 = p.nCB >> 1;
  const int c_end = 3 * nCB32;      // chunks per tile: (kernel row, 32-channel block)
  const int S = ntile * c_end;      // chunks this workgroup computes
  constexpr int us = UP ? 1 : 0;    // input upsampled x2 on the fly (compile-time: the loads differ in width)
  const unsigned chan_bytes = (unsigned)(p.HW >> (2 * us)) * 4;
  const unsigned stepA = (unsigned)p.Cout * 384;  // bytes of packed weights per 32-channel chunk
  float s_in = 1.0f, omul = WSCALE_INV;
  if (!SPLIT && p.in_scale != nullptr) {
    s_in = p.in_scale[0];
    omul = WSCALE_INV * p.in_scale[1];
  }
  auto tile_m0 = [&](int k) { return ((slot + k * nslots) % p.mtiles) * BM; };
  auto tile_n0 = [&](int k) { return ((long)((slot + k * nslots) / p.mtiles) * 8 + xcd) * BN; };

  // ---- weight stream.  A row's 24 pieces = 2 packed 16-channel chunks x (hi | lo) x 3 taps x 2 channel octets,
  // fetched in the packed layout's own order (192 contiguous bytes per chunk and row).  Waves 4-7 stage rows 0-63 of a
  // weight tile, waves 0-3 rows 64-127 (see the main loop for why).
  unsigned voffA[NA];
  int ldsA[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int idx = (tid & 255) + 256 * j;
    const int row = (tid < 256 ? 64 : 0) + idx / 24, pq = idx % 24, kgh = pq / 12, q12 = pq % 12;
    const int hl = q12 / 6, sx = (q12 % 6) >> 1, kgl = q12 & 1;
    ldsA[j] = (hl * BM + row) * ROWL + (sx * 4 + ((kgh * 2 + kgl) ^ ((row >> 2) & 2))) * 16;
  }
  auto a_tile = [&](int k) {   // lane offsets of tile k's weight pieces (past the last tile: out of range -> zeros)
    const int m0 = tile_m0(k);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = (tid & 255) + 256 * j;
      const int row = (tid < 256 ? 64 : 0) + idx / 24, pq = idx % 24, kgh = pq / 12, q12 = pq % 12;
      voffA[j] = (k < ntile && m0 + row < p.Cout) ? (unsigned)((kgh * p.Cout + m0 + row) * 192 + q12 * 16) : OOR;
    }
  };
  int ka = 0, ca = 0;   // tile and chunk of the weight stream's next load
  uint4v ra[NA];
  auto load_a = [&]() {
    const unsigned sA = (unsigned)ca * stepA;
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = bufload4(rA, voffA[j], sA);
    if (++ca == c_end) { ca = 0; a_tile(++ka); }
  };
  auto store_a = [&](int buf) {
    unsigned char* dst = smem + buf * ABUF;
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<uint4v*>(dst + ldsA[j]) = ra[j];
  };

  // ---- activation stream.  Lane n16 owns the PAIR of positions n0 + 32 wave + 2 n16 + {0, 1}: column tile 0 = the
  // even positions, column tile 1 = the odd ones.  One 8-byte load per channel fetches both (half the vector-memory
  // instructions of one position per lane and tile), and a pixel's neighbour is mostly the OTHER column tile's value
  // in the same lane: left of (even, lane n) = (odd, lane n-1), right of it = (odd, lane n); left of (odd, lane n) =
  // (even, lane n), right of it = (even, lane n+1) -- two row shifts per channel pair instead of four.  W is even,
  // so a pair never straddles an image row and both of its positions are valid or invalid together.  A tile starts at
  // a multiple of 256 positions and W divides 256: the column px -- and with it the border masks -- is the same in
  // every tile; sample and row (pb, py) belong to the stream's current tile.
  const int px = (wave * 32 + 2 * n16) % p.W;
  const bool has_left0 = px > 0, has_right1 = px + 2 < p.W;   // (even position's left tap, odd position's right tap)
  bool nvalid = false;
  int pb = 0, py = 0;
  auto b_tile = [&](int k) {
    const long nj = tile_n0(k) + wave * 32 + 2 * n16;
    nvalid = k < ntile && nj < p.Npos;
    pb = py = 0;
    if (nvalid) {
      pb = (int)(nj / p.HW);
      py = (int)(nj - (long)pb * p.HW) / p.W;
    }
  };
  unsigned voffB = OOR;
  auto set_row = [&](int r) {
    const int iy = py + r - 1;
    const bool rok = nvalid && (unsigned)iy < (unsigned)p.H;
    long off = ((long)pb * p.Cin + 8 * kg) * p.HW + iy * p.W + px;
    // upsampled input: both positions of the pair are source pixel (iy >> 1, px >> 1)
    if (us) off = ((long)pb * p.Cin + 8 * kg) * (p.HW >> 2) + (iy >> 1) * (p.W >> 1) + (px >> 1);
    voffB = rok ? (unsigned)(off * 4) : OOR;
  };
  int kb = 0, rb_r = 0, rb_blk = 0;   // tile, kernel row and 32-channel block of the activation stream's next load
  float rb[2][8];                      // [even | odd position][channel]
  auto prep_b = [&]() {
    if (rb_blk == 0) set_row(rb_r);   // a new kernel row (of a new tile, possibly)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned sB = (unsigned)(rb_blk * 32 + j) * chan_bytes;
      if (us) {
        rb[0][j] = bufload(rB, voffB, sB);
        rb[1][j] = rb[0][j];
      } else {
        typedef unsigned uint2v __attribute__((ext_vector_type(2)));
        const uint2v v2 = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(rB, voffB, sB, 0));
        const unsigned u0 = v2.x, u1 = v2.y;
        rb[0][j] = __builtin_bit_cast(float, u0);
        rb[1][j] = __builtin_bit_cast(float, u1);
      }
    }
    if (++rb_blk == nCB32) {
      rb_blk = 0;
      if (++rb_r == 3) { rb_r = 0; b_tile(++kb); }
    }
  };
  // fragments [tap][column tile] of one chunk out of the raw centre-tap registers: two conversions, two row shifts
  auto build_frags = [&](half8 (*fh)[2], half8 (*fl)[2]) {
    uint4v ch[2], cl[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      half8 h8, l8;
      if (SPLIT) unpack8(rb[ct], h8, l8);
      else split8s(rb[ct], s_in, h8, l8);
      fh[1][ct] = h8;
      fl[1][ct] = l8;
      ch[ct] = __builtin_bit_cast(uint4v, h8);
      cl[ct] = __builtin_bit_cast(uint4v, l8);
    }
    uint4v lh, ll, rh, rl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // left of the even position: the odd position one lane down (0 into a 16-lane row's first lane: with image rows
      // of 32 that IS the border; narrower rows also end inside the 16 lanes)
      if (WM == 1) {   // image rows of 2: a pair IS a row -- the even position has no left, the odd one no right
        lh[e] = ll[e] = rh[e] = rl[e] = 0u;
        continue;
      }
      lh[e] = row_dpp<DPP_ROW_SHR1, true>(0u, ch[1][e]);
      ll[e] = row_dpp<DPP_ROW_SHR1, true>(0u, cl[1][e]);
      rh[e] = row_dpp<DPP_ROW_SHL1, true>(0u, ch[0][e]);
      rl[e] = row_dpp<DPP_ROW_SHL1, true>(0u, cl[0][e]);
      if (WM != 2) {
        lh[e] = has_left0 ? lh[e] : 0u;
        ll[e] = has_left0 ? ll[e] : 0u;
        rh[e] = has_right1 ? rh[e] : 0u;
        rl[e] = has_right1 ? rl[e] : 0u;
      }
    }
    fh[0][0] = __builtin_bit_cast(half8, lh);   // left taps:  even <- shifted odd,  odd <- even
    fl[0][0] = __builtin_bit_cast(half8, ll);
    fh[0][1] = fh[1][0];
    fl[0][1] = fl[1][0];
    fh[2][0] = fh[1][1];                         // right taps: even <- odd,          odd <- shifted even
    fl[2][0] = fl[1][1];
    fh[2][1] = __builtin_bit_cast(half8, rh);
    fl[2][1] = __builtin_bit_cast(half8, rl);
  };

  f32x4 acc[8][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][ct][r] = 0.0f;
  };
  // Epilogue of tile k: the quad epilogue behind a functor that knows the accumulator layout.  The patches have an LDS
  // area of their own behind the two weight stages (the other group reads a stage meanwhile).
  float* patches = reinterpret_cast<float*>(smem + 2 * ABUF) + wave * (32 * EPI_PITCH);
  // the bias of every output-channel tile (zeros where there is none) in LDS, once per workgroup: the epilogue then
  // starts without a global-memory round trip (16 dword loads and their latency per tile and wave)
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * ABUF) + 8 * 32 * EPI_PITCH;
  for (int i = tid; i < p.mtiles * BM; i += NT) bias_lds[i] = (p.bias != nullptr && i < p.Cout) ? p.bias[i] : 0.0f;
  auto put = [&](int T, float* sc) {  // acc[t][ct][i] = channel 16 t + 4 kg + i, position 2 n16 + ct
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (an ext-vector store: a store through HIP's float2 STRUCT may be moved past the epilogue's float4-struct
        // reads of the same patch by type-based alias analysis -- it was: stale patches in about half the samples)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 v2 = {acc[2 * T + tt][0][i], acc[2 * T + tt][1][i]};
        *reinterpret_cast<f32x2*>(sc + (tt * 16 + 4 * kg + i) * EPI_PITCH + 2 * n16) = v2;
      }
  };

  // ---- main loop: the two waves of a SIMD ALTERNATE.  Waves 0-3 (one per SIMD) and waves 4-7 run half a period apart:
  // while one group issues the 144 MFMAs of a chunk (with nothing but the weight-fragment reads between them) the other
  // stages weights, builds its activation fragments and issues its loads -- each phase ends in a workgroup barrier.
  // With both waves of a SIMD in the same phase (one barrier per chunk) the matrix pipe idles whenever both are in
  // their staging part and is contended whenever both are in their MFMA part: measured 389 TF against 562 TF for the
  // same loop with everything but the MFMAs compiled out (tools/r3_m16_ablate.sh).  s = chunk number over all tiles:
  //   phase 2s:   group 0 MFMA(s)      group 1 stages (chunk s-1's tail: weights s+1, fragments s, loads)
  //   phase 2s+1: group 0 stages       group 1 MFMA(s)
  // A weight stage (s & 1) is filled half by each group: stage s+1 by group 1 in phase 2s (its readers of chunk s-1
  // finished in 2s-1) and by group 0 in phase 2s+1 -- group 1 therefore runs its weight stream one chunk further
  // ahead (`lead`).  Row halves: group 1 stages rows 0-63, group 0 rows 64-127, so that at the END of its staging phase
  // a group can already fetch the first fragment group of its next chunk (tap 0, rows 0-63: written by group 1 at
  // least one barrier earlier) -- those reads land while the wave waits at the barrier instead of at the head of its
  // MFMA phase (fetched after the barrier they cost ~10 % of the kernel: 8 reads x 4 waves queue on the LDS before one
  // MFMA).  After the last chunk of a tile the staging phase first stores the pending weight pieces, then runs the
  // tile's epilogue (the accumulators, 80 registers of bias / residual and nothing else are live then) and clears the
  // accumulators, then goes on as usual: the stage, fragment and load pipelines know nothing of tiles.
  const int grp = wave >> 2, lead = 1 + grp;
  half8 bh[3][2], bl[3][2];
  // A fragments a GROUP ahead: group g = (tap g / 2, row tiles 4 (g & 1) .. +3): 8 ds_read_b128 that have the 24
  // MFMAs (384 cycles) of the previous group to land
  half8 af[2][4][2];
  const int a_lane = n16 * ROWL + (kg ^ ((n16 >> 2) & 2)) * 16;
  auto load_group = [&](int slot_, int g, int c) {
    const unsigned char* a = smem + (c & 1) * ABUF + a_lane + (g & 1) * 64 * ROWL + (g >> 1) * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      af[slot_][t][0] = *reinterpret_cast<const half8*>(a + t * 16 * ROWL);
      af[slot_][t][1] = *reinterpret_cast<const half8*>(a + t * 16 * ROWL + BM * ROWL);
    }
  };
  // (rows of 2, WM = 1: the left tap of the even column tile and the right tap of the odd one are all padding -- those
  // MFMAs would add exact zeros and are not issued: 96 instead of 144 per chunk)
  auto mfma_group = [&](int slot_, int g) {
    const int sx = g >> 1;
    const bool do0 = !(WM == 1 && sx == 0), do1 = !(WM == 1 && sx == 2);
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
      const int t = (g & 1) * 4 + tq;
      if (do0) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][0], bh[sx][0], acc[t][0], 0, 0, 0);
      if (do1) acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][0], bh[sx][1], acc[t][1], 0, 0, 0);
      if (do0) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][0], bl[sx][0], acc[t][0], 0, 0, 0);
      if (do1) acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][0], bl[sx][1], acc[t][1], 0, 0, 0);
      if (do0) acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][1], bh[sx][0], acc[t][0], 0, 0, 0);
      if (do1) acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][1], bh[sx][1], acc[t][1], 0, 0, 0);
    }
  };
  a_tile(0);
  b_tile(0);
  zero_acc();
  load_a();               // chunk 0
  prep_b();               // chunk 0
  store_a(0);
  if (grp) load_a();      // chunk 1 (S >= 3)
  build_frags(bh, bl);
  if (grp) store_a(1);
  load_a();               // group 0: chunk 1, group 1: chunk 2
  prep_b();               // chunk 1
  __syncthreads();        // stage 0 complete
  load_group(0, 0, 0);
  if (grp) __syncthreads();   // phase 0: group 0 computes chunk 0
  int kc = 0, cc = 0;         // tile and chunk-in-tile of the MFMA phase
  for (int s = 0;; ++s) {
    if (!(p.ablate & 64)) __builtin_amdgcn_s_setprio(1);  // the MFMA phase outranks its partner's staging (+1-2 %)
    const bool last = s + 1 == S;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      // the next group's 8 fragment reads go out ONE per three MFMAs: issued as a block they hold the wave's
      // instruction stream for ~50 cycles in which no MFMA issues and the matrix pipe runs dry (measured: 12 %)
      if (g < 5) load_group((g + 1) & 1, g + 1, s);
      mfma_group(g & 1, g);
      if (WM == 1 && (g >> 1) != 1) {   // the 12-MFMA groups: 2, 1, 2, 1, ... MFMAs per fragment read
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(p.ablate & 64)) __builtin_amdgcn_s_setprio(0);
    // (the very last chunk: group 0 meets group 1's last staging barrier, group 1 just leaves.  The barrier one MFMA
    // group EARLIER, so that the other group's first MFMAs issue beside this group's last 24 instead of after a barrier
    // round trip: measured 6-7 % SLOWER on three bench shapes -- the groups' MFMAs contend, they do not interleave)
    if (!last || !grp) __syncthreads();
    // ---- staging phase
    if (s + lead < S) store_a((s + lead) & 1);
    if (++cc == c_end) {
      if (!(p.ablate & 128) || last)   // (128: timing experiment -- only the last tile's epilogue runs)
        h2_epilogue_quad_put<4, true>(p, put, p.out, true, tile_n0(kc) + wave * 32, tile_m0(kc), lane, patches,
                                      p.ablate, omul, nullptr, bias_lds + tile_m0(kc));
      zero_acc();
      cc = 0;
      ++kc;
    }
    if (last) break;
    build_frags(bh, bl);   // chunk s+1's fragments over chunk s's (their MFMAs have issued)
    load_a();
    prep_b();
    load_group(0, 0, s + 1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions with a SHORT contraction (Cin <= 128: the qkv / proj convs of the 128-channel levels) as a
// weight-stationary, barrier-free position loop.  Measured on the tiled kernel above (128 -> 384 qkv conv, 512 samples
// at 16x16: 94.7 us): a workgroup with its main loop AND its stores compiled out still costs 34 us of that -- 3072
// workgroups of two K chunks each are all prologue / epilogue, and with two 256-VGPR workgroups per CU nothing
// overlaps one's loads with the other's stores.  Here one 8-wave workgroup per CU stages the whole 128-row weight tile
// (both chunks, hi + lo: 72 KB) in LDS ONCE, then every wave walks its own 32-position slices of the workgroup's
// position tiles with no barrier at all: waves drift apart, so one wave's epilogue stores run beside another's
// activation loads and MFMAs.  Same fragments, same MFMA order per output element, same epilogue as the tiled kernel
// => the same bits.
template <bool SPLIT>
__global__ __launch_bounds__(512, 2) void conv1x1_h2_persist_kernel(ConvHParams p, int nworkers, int tile0,
                                                                    int ntiles) {
  constexpr int BM = 128, NW = 8, BN = 32 * NW, KC = 64, NKS = KC / 16;
  constexpr int ROWL = KC * 2 + 16, RB = 2 * KC * 2, PPR = RB / 16, ABUF = 2 * BM * ROWL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  // same XCD-aware order as the tiled kernel: the Cout tiles that share a position range sit on one XCD's L2
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int mt = q % p.mtiles;
  const int worker = (q / p.mtiles) * 8 + xcd;
  if (worker >= nworkers) return;  // grid padding (whole workgroup, before the only barrier)
  const int first = tile0 + worker;  // this launch walks position tiles [tile0, ntiles)
  const int m0 = mt * BM;
  const int nch = p.nCB;           // 1 or 2 chunks of 64 channels (checked on the host)

  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  for (int idx = tid; idx < nch * BM * PPR; idx += 64 * NW) {
    const int c = idx / (BM * PPR), r = idx - c * (BM * PPR);
    const int row = r / PPR, piece = r - row * PPR, lo = piece >= PPR / 2;
    const unsigned voff = (m0 + row < p.Cout) ? (unsigned)((m0 + row) * RB + piece * 16) : OOR;
    const uint4v v = bufload4(rA, voff, (unsigned)c * (unsigned)p.Cout * RB);
    *reinterpret_cast<uint4v*>(smem + c * ABUF + (lo ? BM * ROWL : 0) + row * ROWL + (piece - lo * (PPR / 2)) * 16) = v;
  }
  __syncthreads();
  float* scratch = reinterpret_cast<float*>(smem + nch * ABUF) + wave * (32 * EPI_PITCH);

  float s_in = 1.0f, omul = WSCALE_INV;
  if (!SPLIT && p.in_scale != nullptr) {
    s_in = p.in_scale[0];
    omul = WSCALE_INV * p.in_scale[1];
  }
  const unsigned chan_bytes = (unsigned)p.HW * 4;
  float bv[16];  // loop-invariant: this lane's channels are the same in every position tile
  h2_quad_bias<16>(p, true, m0, lane, bv);
  auto tile_voff = [&](int nt) -> unsigned {  // this lane's activation offset in position tile nt (OOR past the end)
    const long nj = (long)nt * BN + wave * 32 + l31;
    if (nt >= ntiles || nj >= p.Npos) return OOR;
    const int pb = (int)(nj / p.HW), pos = (int)(nj - (long)pb * p.HW);
    return (unsigned)((((long)pb * p.Cin + 8 * h) * p.HW + pos) * 4);
  };
  // rb[kq] = the 8 activations of k-step kq.  Software pipeline: as soon as a k-step's registers have been turned into
  // fragments they are re-requested for the NEXT tile, so those loads fly under this tile's MFMAs and epilogue stores
  // (the epilogue itself loads nothing: no residual on this path, bias hoisted).
  float rb[2 * NKS][8];
  auto request = [&](int kq, unsigned voff) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      rb[kq][j] = bufload(rB, (kq < nch * NKS) ? voff : OOR, (unsigned)(kq * 16 + j) * chan_bytes);
  };
  {
    const unsigned v0 = tile_voff(first);
#pragma unroll
    for (int kq = 0; kq < 2 * NKS; ++kq) request(kq, v0);
  }
  for (int nt = first; nt < ntiles; nt += nworkers) {
    const unsigned vnext = tile_voff(nt + nworkers);
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int kq = 0; kq < 2 * NKS; ++kq) {
      half8 bh, bl;
      if (SPLIT) unpack8(rb[kq], bh, bl);
      else split8s(rb[kq], s_in, bh, bl);
      request(kq, vnext);
      if (kq < nch * NKS && !(p.ablate & 2)) {  // (wave-uniform)
        const unsigned char* a_base = smem + (kq / NKS) * ABUF + (l31 * ROWL + 16 * h) + (kq % NKS) * 32;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const half8 ah = *reinterpret_cast<const half8*>(a_base + t * 32 * ROWL);
          const half8 al = *reinterpret_cast<const half8*>(a_base + BM * ROWL + t * 32 * ROWL);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
        }
      }
    }
    h2_epilogue_quad<4, false>(p, acc, p.out, true, (long)nt * BN + wave * 32, m0, lane, scratch, p.ablate, omul, bv);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 convolutions with a LONG contraction (Cin >= 256: the qkv / proj convs of the 256-channel levels, the decoder's
// skip convs over a channel concat) -- and the 128-channel ones that add a residual -- on v_mfma_f32_16x16x32_f16, in
// the form of conv3x3_h2_m16_kernel: one persistent 8-wave workgroup per CU, the two waves of a SIMD alternating MFMA
// and staging phases, weights streamed through two LDS stages, a lane owning a pair of positions (8-byte loads).
// These layers ran at 205-250 TF on the tiled 32x32x16 kernel: neither at the matrix pipe's rate nor at HBM's.
// A chunk = the packed layout's 64-channel chunk = 2 k-steps of 32 channels = 96 MFMAs per wave; its weight tile is
// 128 rows x (hi 128 B + lo 128 B).  LDS rows are the bare 128 bytes (8 pieces of 16): a fragment read has lanes
// (row n16, piece 4 ks + kg) and ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31},
// ...: with pitch 128 the bank quad of a piece is 8 (row & 1) + piece, so piece ^= (row >> 1) & 7 spreads the sixteen
// pieces of a group over the sixteen quads (rows {0,2,12,14} x kg 0 get XOR 0,1,6,7, rows {4,6,8,10} x kg 1 get
// 2,3,4,5 on piece | 1: eight distinct values per row parity).
// Channel concat (x | x2): a chunk comes from one source (Cin1 % 64 == 0, checked on the host).
template <bool SPLIT>
__global__ __launch_bounds__(512, 2) void conv1x1_h2_m16_kernel(ConvHParams p) {
  constexpr int BM = 128, NT = 512, BN = 256;
  constexpr int ROWL = 128;             // LDS row pitch in bytes: 64 halves
  constexpr int ABUF = 2 * BM * ROWL;   // one weight stage: hi rows, then lo rows
  constexpr int NA = BM * 16 / NT;      // 16-byte weight pieces per thread and chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, (int)p.x2_bytes, 0x00020000);
  // tiles and the one software pipeline over all their chunks: see conv3x3_h2_m16_kernel
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const long npt = (p.Npos + BN - 1) / BN;
  const long lim = (npt > xcd ? (npt - xcd + 7) / 8 : 0) * p.mtiles;
  const int ntile = lim > slot ? (int)((lim - slot + nslots - 1) / nslots) : 0;
  if (ntile == 0) return;   // (whole workgroup, before any barrier)
  const int c_end = p.nCB, nc1 = p.Cin1 >> 6;   // chunks per tile / of them from the first source
  const int S = ntile * c_end;
  const int grp = wave >> 2, lead = 1 + grp;
  float s_in = 1.0f, omul = WSCALE_INV;
  if (!SPLIT && p.in_scale != nullptr) {
    s_in = p.in_scale[0];
    omul = WSCALE_INV * p.in_scale[1];
  }
  const unsigned chan_bytes = (unsigned)p.HW * 4;
  const unsigned stepA = (unsigned)p.Cout * 256;  // bytes of packed weights per chunk
  auto tile_m0 = [&](int k) { return ((slot + k * nslots) % p.mtiles) * BM; };
  auto tile_n0 = [&](int k) { return ((long)((slot + k * nslots) / p.mtiles) * 8 + xcd) * BN; };

  // ---- weight stream: waves 4-7 stage rows 0-63, waves 0-3 rows 64-127; a row's 16 pieces = hi 0-7 | lo 8-15
  unsigned voffA[NA];
  int ldsA[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int idx = (tid & 255) + 256 * j;
    const int row = (tid < 256 ? 64 : 0) + (idx >> 4), q16 = idx & 15;
    ldsA[j] = ((q16 >> 3) * BM + row) * ROWL + (((q16 & 7) ^ ((row >> 1) & 7)) << 4);
  }
  auto a_tile = [&](int k) {
    const int m0 = tile_m0(k);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int idx = (tid & 255) + 256 * j;
      const int row = (tid < 256 ? 64 : 0) + (idx >> 4), q16 = idx & 15;
      voffA[j] = (k < ntile && m0 + row < p.Cout) ? (unsigned)((m0 + row) * 256 + q16 * 16) : OOR;
    }
  };
  int ka = 0, ca = 0;
  uint4v ra[NA];
  auto load_a = [&]() {
    const unsigned sA = (unsigned)ca * stepA;
#pragma unroll
    for (int j = 0; j < NA; ++j) ra[j] = bufload4(rA, voffA[j], sA);
    if (++ca == c_end) { ca = 0; a_tile(++ka); }
  };
  auto store_a = [&](int buf) {
    unsigned char* dst = smem + buf * ABUF;
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<uint4v*>(dst + ldsA[j]) = ra[j];
  };

  // ---- activation stream: lane n16 owns the pair n0 + 32 wave + 2 n16 + {0, 1} (HW % 4 == 0: never across samples)
  unsigned voffB1 = OOR, voffB2 = OOR;
  auto b_tile = [&](int k) {
    const long nj = tile_n0(k) + wave * 32 + 2 * n16;
    voffB1 = voffB2 = OOR;
    if (k < ntile && nj < p.Npos) {
      const int pb = (int)(nj / p.HW), pos = (int)(nj - (long)pb * p.HW);
      voffB1 = (unsigned)((((long)pb * p.Cin1 + 8 * kg) * p.HW + pos) * 4);
      if (p.x2) voffB2 = (unsigned)((((long)pb * (p.Cin - p.Cin1) + 8 * kg) * p.HW + pos) * 4);
    }
  };
  int kb = 0, cb = 0;
  float rb[2][2][8];   // [k-step][even | odd position][channel]
  auto prep_b = [&]() {
    const bool first = cb < nc1;   // workgroup-uniform
    const unsigned cbase = (unsigned)((first ? cb : cb - nc1) * 64);
    const unsigned voff = first ? voffB1 : voffB2;
    if (first) {   // (a uniform branch, not a select of the descriptor: that would put it into vector registers)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          typedef unsigned uint2v __attribute__((ext_vector_type(2)));
          const uint2v v2 = __builtin_bit_cast(
              uint2v, __builtin_amdgcn_raw_buffer_load_b64(rB, voff, (cbase + ks * 32 + j) * chan_bytes, 0));
          const unsigned u0 = v2.x, u1 = v2.y;
          rb[ks][0][j] = __builtin_bit_cast(float, u0);
          rb[ks][1][j] = __builtin_bit_cast(float, u1);
        }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          typedef unsigned uint2v __attribute__((ext_vector_type(2)));
          const uint2v v2 = __builtin_bit_cast(
              uint2v, __builtin_amdgcn_raw_buffer_load_b64(rB2, voff, (cbase + ks * 32 + j) * chan_bytes, 0));
          const unsigned u0 = v2.x, u1 = v2.y;
          rb[ks][0][j] = __builtin_bit_cast(float, u0);
          rb[ks][1][j] = __builtin_bit_cast(float, u1);
        }
    }
    if (++cb == c_end) { cb = 0; b_tile(++kb); }
  };
  auto build_frags = [&](half8 (*fh)[2], half8 (*fl)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        if (SPLIT) unpack8(rb[ks][ct], fh[ks][ct], fl[ks][ct]);
        else split8s(rb[ks][ct], s_in, fh[ks][ct], fl[ks][ct]);
      }
  };

  f32x4 acc[8][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][ct][r] = 0.0f;
  };
  float* patches = reinterpret_cast<float*>(smem + 2 * ABUF) + wave * (32 * EPI_PITCH);
  float* bias_lds = reinterpret_cast<float*>(smem + 2 * ABUF) + 8 * 32 * EPI_PITCH;
  for (int i = tid; i < p.mtiles * BM; i += NT) bias_lds[i] = (p.bias != nullptr && i < p.Cout) ? p.bias[i] : 0.0f;
  auto put = [&](int T, float* sc) {  // acc[t][ct][i] = channel 16 t + 4 kg + i, position 2 n16 + ct
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));   // (not HIP's float2 struct: see the 3x3 kernel)
        const f32x2 v2 = {acc[2 * T + tt][0][i], acc[2 * T + tt][1][i]};
        *reinterpret_cast<f32x2*>(sc + (tt * 16 + 4 * kg + i) * EPI_PITCH + 2 * n16) = v2;
      }
  };

  half8 bh[2][2], bl[2][2];
  // A fragments a GROUP ahead: group g = (k-step g / 2, row tiles 4 (g & 1) .. +3), 8 ds_read_b128 per group
  half8 af[2][4][2];
  const int a_lane[2] = {n16 * ROWL + ((kg ^ (n16 >> 1)) << 4), n16 * ROWL + (((4 + kg) ^ (n16 >> 1)) << 4)};
  auto load_group = [&](int slot_, int g, int c) {
    const unsigned char* a = smem + (c & 1) * ABUF + a_lane[g >> 1] + (g & 1) * 64 * ROWL;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      af[slot_][t][0] = *reinterpret_cast<const half8*>(a + t * 16 * ROWL);
      af[slot_][t][1] = *reinterpret_cast<const half8*>(a + t * 16 * ROWL + BM * ROWL);
    }
  };
  auto mfma_group = [&](int slot_, int g) {
    const int ks = g >> 1;
#pragma unroll
    for (int tq = 0; tq < 4; ++tq) {
      const int t = (g & 1) * 4 + tq;
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][0], bh[ks][0], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][0], bh[ks][1], acc[t][1], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][0], bl[ks][0], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][0], bl[ks][1], acc[t][1], 0, 0, 0);
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][1], bh[ks][0], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[slot_][tq][1], bh[ks][1], acc[t][1], 0, 0, 0);
    }
  };
  // phases and barriers exactly as in conv3x3_h2_m16_kernel (group 1 runs its weight stream `lead` = 2 ahead)
  a_tile(0);
  b_tile(0);
  zero_acc();
  load_a();               // chunk 0
  prep_b();               // chunk 0
  store_a(0);
  if (grp) load_a();      // chunk 1 (S >= 2)
  build_frags(bh, bl);
  if (grp) store_a(1);
  load_a();               // group 0: chunk 1, group 1: chunk 2
  prep_b();               // chunk 1
  __syncthreads();        // stage 0 (and the bias) complete
  load_group(0, 0, 0);
  if (grp) __syncthreads();   // phase 0: group 0 computes chunk 0
  int kc = 0, cc = 0;
  for (int s = 0;; ++s) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) load_group((g + 1) & 1, g + 1, s);
      mfma_group(g & 1, g);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    const bool last = s + 1 == S;
    if (!last || !grp) __syncthreads();
    // ---- staging phase
    if (s + lead < S) store_a((s + lead) & 1);
    if (++cc == c_end) {
      h2_epilogue_quad_put<4, true>(p, put, p.out, true, tile_n0(kc) + wave * 32, tile_m0(kc), lane, patches, p.ablate,
                                    omul, nullptr, bias_lds + tile_m0(kc));
      zero_acc();
      cc = 0;
      ++kc;
    }
    if (last) break;
    build_frags(bh, bl);
    load_a();
    prep_b();
    load_group(0, 0, s + 1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Epilogue of the weight-stationary 16x16x32 kernel below, straight from the accumulators (no LDS): with the pair
// layout acc[t][ct][i] of lane (n16, kg) = channel 16 t + 4 kg + i, position 2 n16 + ct -- the two column tiles of a
// lane are two ADJACENT positions of one channel, so (acc[t][0][i], acc[t][1][i]) is one 8-byte store and the sixteen
// lanes n16 of a channel write one 128-byte run per instruction.  32 store instructions per wave and slice (the
// transposing quad epilogue: 16) -- affordable where waves are decoupled and one wave's epilogue runs beside another
// wave's MFMAs; in the lock-step kernels above this form measured no better than the quad epilogue.
// The residual is loaded one channel block ahead (a lane reads exactly the elements it writes: an output that IS the
// residual's buffer stays correct).  `bias_tile`: the tile's 128 bias values in LDS (zeros where there is none).
template <bool HAS_RES, bool QKV>
__device__ __forceinline__ void ws_epilogue_pairs(const ConvHParams& p, const f32x4 (&acc)[8][2], long nw, int m0,
                                                  int lane, int ablate, float omul, const float* bias_tile) {
  typedef unsigned uint2v __attribute__((ext_vector_type(2)));
  const int n16 = lane & 15, kg = lane >> 4;
  const unsigned out_bytes = (unsigned)(p.Npos * p.Cout * 4);
  const unsigned HWu = (unsigned)p.HW, plane = HWu * 4;
  const int c0 = m0 + 4 * kg;                    // this lane's channel for (t, i): c0 + 16 t + i
  const long nq = nw + 2 * n16;                  // its position pair (never across samples: HW is even)
  const bool pvalid = nq < p.Npos;
  unsigned pb = 0, pos = 0;
  if (pvalid) {   // (32-bit: positions stay below 2^29 -- the host checks out_bytes < 2 GiB)
    pb = (unsigned)nq / HWu;
    pos = (unsigned)nq - pb * HWu;
  }
  const unsigned voff = pvalid ? ((pb * (unsigned)p.Cout + (unsigned)c0) * HWu + pos) * 4u : OOR;
  __amdgpu_buffer_rsrc_t rO =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, __builtin_amdgcn_readfirstlane((int)out_bytes), 0x00020000);
  __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.res, 0, __builtin_amdgcn_readfirstlane(HAS_RES ? (int)out_bytes : 0), 0x00020000);
  const int lim = p.Cout - c0;                   // channel offsets 16 t + i below lim exist for this lane
  const int cmod = c0 % 3;
  auto load_res = [&](int t, uint2v (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      r[i] = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(rR, 16 * t + i < lim ? voff : OOR,
                                                                             (unsigned)(16 * t + i) * plane, 0));
  };
  uint2v rcur[4], rnext[4];
  if (HAS_RES && !(ablate & 16)) load_res(0, rcur);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (HAS_RES && (ablate & 16)) load_res(t, rcur);   // (debug: no prefetch)
    if (HAS_RES && !(ablate & 16) && t + 1 < 8) load_res(t + 1, rnext);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_tile + 16 * t + 4 * kg);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int cof = 16 * t + i;
      // (the bias through two separately laundered copies: left alone, the compiler packs the two fmas into one
      // v_pk_fma_f32 that broadcasts the bias register pair's high element through op_sel -- and on gfx950 the LOW
      // result took the low element instead in lanes 48-63 now and then, right behind a v_mov into the packed source:
      // element 0 of a pair got its neighbour channel's bias, timing dependent, tools/ws_diag.py)
      float b0 = bv[i], b1 = bv[i];
      asm volatile("" : "+v"(b0));
      asm volatile("" : "+v"(b1));
      float y0 = __builtin_fmaf(acc[t][0][i], omul, b0);
      float y1 = __builtin_fmaf(acc[t][1][i], omul, b1);
      if (HAS_RES) {
        const unsigned u0 = rcur[i].x, u1 = rcur[i].y;
        y0 += __builtin_bit_cast(float, u0);
        y1 += __builtin_bit_cast(float, u1);
      }
      if (QKV) {
        if ((cmod + cof) % 3 == 0) {
          y0 *= I2R_QSCALE;
          y1 *= I2R_QSCALE;
        }
        y0 = split_pack(y0);
        y1 = split_pack(y1);
      }
      uint2v o = {__builtin_bit_cast(unsigned, y0), __builtin_bit_cast(unsigned, y1)};
      unsigned vo = (cof < lim && voff != OOR && (!(ablate & 1) || o.x == 0x12345678u))
                        ? voff + (unsigned)cof * plane : OOR;
      __builtin_amdgcn_raw_buffer_store_b64(o, rO, vo, 0, 0);
      // Two wait states before anything may overwrite the store's registers.  The ISA manuals ask for one only behind
      // stores of MORE than 8 bytes and the compiler follows them; on gfx950 an 8-byte store whose data pair was
      // re-used by the next two v_mov lost elements in ~9 % of the samples (first form of this epilogue, residual
      // variant; the 16-byte case: see the quad epilogue).  The empty-looking asm ties the registers: whatever writes
      // them next is ordered behind it.
      asm volatile("s_nop 1" : "+v"(o), "+v"(vo));
    }
    if (HAS_RES && !(ablate & 16)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rcur[i] = rnext[i];
    }
  }
}

// 1x1 convolutions with Cin = 128 or 256 (NCH = 2 / 4 packed 64-channel chunks) WEIGHT-STATIONARY on
// v_mfma_f32_16x16x32_f16: one 8-wave workgroup per CU stages its whole 128-row weight tile (hi + lo, NCH x 32 KB:
// 128 KB at Cin = 256 -- unpadded 128-byte rows, swizzled as in conv1x1_h2_m16_kernel, leave room for it) in LDS ONCE,
// and after that single barrier every wave walks 32-position slices on its own.  No barrier, no lock step: a wave's
// epilogue (ws_epilogue_pairs) runs beside the other waves' MFMAs, which is what the lock-step pipeline above cannot
// give these layers -- there a tile is 2-4 chunks and its epilogue a third of the kernel.  Activations:
// a lane owns a pair of positions (8-byte loads), prefetched FOUR k-steps of 32 channels ahead through a register ring
// that runs on into the next slice; weight fragments a group of four row tiles ahead, one read per three MFMAs.
// Channel concat (x | x2): Cin1 % 64 == 0.  Tiles are dealt XCD-aware: the Cout-tile workers that share a position
// range sit on one XCD's L2.
template <bool SPLIT, int NCH>
__global__ __launch_bounds__(512, 2) void conv1x1_h2_ws_kernel(ConvHParams p, int nworkers) {
  constexpr int BM = 128, ROWL = 128, CHB = 2 * BM * ROWL;   // bytes of one chunk's weight tile: hi rows, lo rows
  constexpr int KS = 2 * NCH;                                 // k-steps of 32 channels per slice
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, kg = lane >> 4;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int mt = q % p.mtiles;
  const int worker = (q / p.mtiles) * 8 + xcd;
  if (worker >= nworkers) return;   // (whole workgroup, before the only barrier)
  const int m0 = mt * BM;
  __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)p.wp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rB2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, (int)p.x2_bytes, 0x00020000);
  // ---- the weight tile, once: piece (chunk c, row, q16 = hi 0-7 | lo 8-15) -> swizzled LDS slot
  for (int idx = tid; idx < NCH * 2048; idx += 512) {
    const int c = idx >> 11, r = idx & 2047, row = r >> 4, q16 = r & 15;
    const unsigned voff = (m0 + row < p.Cout) ? (unsigned)((m0 + row) * 256 + q16 * 16) : OOR;
    const uint4v v = bufload4(rA, voff, (unsigned)c * (unsigned)p.Cout * 256u);
    *reinterpret_cast<uint4v*>(smem + c * CHB + ((q16 >> 3) * BM + row) * ROWL + (((q16 & 7) ^ ((row >> 1) & 7)) << 4)) = v;
  }
  float* bias_lds = reinterpret_cast<float*>(smem + NCH * CHB);
  if (tid < BM) bias_lds[tid] = (p.bias != nullptr && m0 + tid < p.Cout) ? p.bias[m0 + tid] : 0.0f;
  __syncthreads();

  float s_in = 1.0f, omul = WSCALE_INV;
  if (!SPLIT && p.in_scale != nullptr) {
    s_in = p.in_scale[0];
    omul = WSCALE_INV * p.in_scale[1];
  }
  const unsigned chan_bytes = (unsigned)p.HW * 4;
  const int nc1 = p.Cin1 >> 6;                    // chunks from the first source
  const long nslices = (p.Npos + 31) >> 5;
  const long stride = (long)nworkers * 8;
  // lane offsets of slice s in the two sources (out of range past the end)
  auto slice_off = [&](long sl, unsigned& v1, unsigned& v2) {
    const long nj = sl * 32 + 2 * n16;
    v1 = v2 = OOR;
    if (sl < nslices && nj < p.Npos) {
      const unsigned pb = (unsigned)nj / (unsigned)p.HW, pos = (unsigned)nj - pb * (unsigned)p.HW;
      v1 = ((pb * (unsigned)p.Cin1 + 8 * kg) * (unsigned)p.HW + pos) * 4;
      if (p.x2) v2 = ((pb * (unsigned)(p.Cin - p.Cin1) + 8 * kg) * (unsigned)p.HW + pos) * 4;
    }
  };
  // rb[slot]: the raw activations of one k-step (8 channels x 2 positions); k-step j lives in slot j & 3
  float rb[4][2][8];
  auto request = [&](int slot, int j, unsigned v1, unsigned v2) {   // k-step j (0 .. KS-1) of the slice at (v1, v2)
    const int c = j >> 1;                          // its chunk
    const bool first = c < nc1;                    // (uniform)
    const unsigned cbase = (unsigned)((first ? c : c - nc1) * 64 + (j & 1) * 32);
    const unsigned voff = first ? v1 : v2;
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    if (first) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint2v v = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(rB, voff, (cbase + e) * chan_bytes, 0));
        const unsigned u0 = v.x, u1 = v.y;
        rb[slot][0][e] = __builtin_bit_cast(float, u0);
        rb[slot][1][e] = __builtin_bit_cast(float, u1);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint2v v = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(rB2, voff, (cbase + e) * chan_bytes, 0));
        const unsigned u0 = v.x, u1 = v.y;
        rb[slot][0][e] = __builtin_bit_cast(float, u0);
        rb[slot][1][e] = __builtin_bit_cast(float, u1);
      }
    }
  };
  // weight fragments: group g = (k-step g >> 1, row tiles 4 (g & 1) .. + 3), 8 ds_read_b128
  half8 af[2][4][2];
  const int a_lane[2] = {n16 * ROWL + ((kg ^ (n16 >> 1)) << 4), n16 * ROWL + (((4 + kg) ^ (n16 >> 1)) << 4)};
  auto load_group = [&](int slot_, int g) {
    const int j = g >> 1;
    const unsigned char* a = smem + (j >> 1) * CHB + a_lane[j & 1] + (g & 1) * 64 * ROWL;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      af[slot_][t][0] = *reinterpret_cast<const half8*>(a + t * 16 * ROWL);
      af[slot_][t][1] = *reinterpret_cast<const half8*>(a + t * 16 * ROWL + BM * ROWL);
    }
  };

  long sl = (long)worker * 8 + wave;
  unsigned v1, v2, nv1, nv2;
  slice_off(sl, v1, v2);
#pragma unroll
  for (int j = 0; j < 4; ++j) request(j, j, v1, v2);   // (KS >= 4)
  for (; sl < nslices; sl += stride) {
    slice_off(sl + stride, nv1, nv2);
    f32x4 acc[8][2];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][ct][r] = 0.0f;
    load_group(0, 0);
    half8 bh[2], bl[2];
#pragma unroll
    for (int g = 0; g < 2 * KS; ++g) {
      const int j = g >> 1;
      if ((g & 1) == 0) {   // a new k-step: its fragments out of the ring, the slot re-requested four k-steps on
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          if (SPLIT) unpack8(rb[j & 3][ct], bh[ct], bl[ct]);
          else split8s(rb[j & 3][ct], s_in, bh[ct], bl[ct]);
        }
        if (j + 4 < KS) request(j & 3, j + 4, v1, v2);
        else request(j & 3, j + 4 - KS, nv1, nv2);
      }
      if (g + 1 < 2 * KS) load_group((g + 1) & 1, g + 1);
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        const int t = (g & 1) * 4 + tq;
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[g & 1][tq][0], bh[0], acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[g & 1][tq][0], bh[1], acc[t][1], 0, 0, 0);
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[g & 1][tq][0], bl[0], acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[g & 1][tq][0], bl[1], acc[t][1], 0, 0, 0);
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[g & 1][tq][1], bh[0], acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[g & 1][tq][1], bh[1], acc[t][1], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (p.res != nullptr) {
      if (p.qkv_out) ws_epilogue_pairs<true, true>(p, acc, sl * 32, m0, lane, p.ablate, omul, bias_lds);
      else ws_epilogue_pairs<true, false>(p, acc, sl * 32, m0, lane, p.ablate, omul, bias_lds);
    } else {
      if (p.qkv_out) ws_epilogue_pairs<false, true>(p, acc, sl * 32, m0, lane, p.ablate, omul, bias_lds);
      else ws_epilogue_pairs<false, false>(p, acc, sl * 32, m0, lane, p.ablate, omul, bias_lds);
    }
    v1 = nv1;
    v2 = nv2;
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
__device__ __forceinline__ void pack_h2_elem(const float* __restrict__ w, _Float16* __restrict__ wp, int Cout, int Cin,
                                             int ks, int nCB, long i) {
  const int KC = ks == 3 ? 48 : 64;
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
  _Float16 hi, lo;
  f16_split(v, hi, lo);
  long base = ((long)chunk * Cout + co) * 2 * KC;
  wp[base + k] = hi;
  wp[base + KC + k] = lo;
}
__global__ void pack_weight_h2_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int Cout, int Cin,
                                      int ks, int nCB, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    pack_h2_elem(w, wp, Cout, Cin, ks, nCB, i);
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

// Packed split weights of the DATA-GRADIENT problem: dx = conv2d(dy, Wd), Wd[ci][co][r][s] = w[co][ci][2-r][2-s]
// (the transposed, tap-flipped filter; autograd of conv2d w.r.t. its input): the forward layout above with the roles
// of Cout and Cin swapped -- rows = ci (the output channels of this problem), k runs over (tap, co).
__device__ __forceinline__ void pack_h2_dgrad_elem(const float* __restrict__ w, _Float16* __restrict__ wp, int Cout,
                                                   int Cin, int ks, int nCB, long i) {
  const int KC = ks == 3 ? 48 : 64;
  int k = (int)(i % KC);
  long t = i / KC;
  int ci = (int)(t % Cin);      // row of the packed matrix = output channel of the dgrad problem
  int chunk = (int)(t / Cin);
  float v = 0.0f;
  if (ks == 3) {
    int r = chunk / nCB, cb = chunk % nCB;
    int s = k / 16, co = cb * 16 + (k % 16);
    if (co < Cout) v = w[((long)co * Cin + ci) * 9 + (2 - r) * 3 + (2 - s)];
  } else {
    int co = chunk * 64 + k;
    if (co < Cout) v = w[(long)co * Cin + ci];
  }
  v *= WSCALE;
  _Float16 hi, lo;
  f16_split(v, hi, lo);
  long base = ((long)chunk * Cin + ci) * 2 * KC;
  wp[base + k] = hi;
  wp[base + KC + k] = lo;
}
__global__ void pack_weight_h2_dgrad_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int Cout, int Cin,
                                            int ks, int nCB, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    pack_h2_dgrad_elem(w, wp, Cout, Cin, ks, nCB, i);
}

// Multi-tensor form (training: one launch per step instead of one or two per layer).  table rows {w, wp, Cout, Cin,
// ksize, dgrad} (int64); blockmap {tensor, block}, a block = H2_PACK_BLOCK split pairs of that tensor.
#define H2_PACK_BLOCK 4096
__global__ __launch_bounds__(256) void pack_weight_h2_multi_kernel(const long* __restrict__ table,
                                                                   const int2* __restrict__ blockmap) {
  const int2 bm = blockmap[blockIdx.x];
  const long* e = table + 6L * bm.x;
  const float* w = (const float*)e[0];
  _Float16* wp = (_Float16*)e[1];
  const int Cout = (int)e[2], Cin = (int)e[3], ks = (int)e[4], dgrad = (int)e[5];
  const int KC = ks == 3 ? 48 : 64;
  const int red = dgrad ? Cout : Cin, rows = dgrad ? Cin : Cout;  // reduction / row channels of this layout
  const int nCB = ks == 3 ? (red + 15) / 16 : (red + 63) / 64;
  const long total = (long)(ks == 3 ? 3 : 1) * nCB * rows * KC;
  const long begin = (long)bm.y * H2_PACK_BLOCK;
  long end = begin + H2_PACK_BLOCK;
  if (end > total) end = total;
  for (long i = begin + threadIdx.x; i < end; i += 256) {
    if (dgrad)
      pack_h2_dgrad_elem(w, wp, Cout, Cin, ks, nCB, i);
    else
      pack_h2_elem(w, wp, Cout, Cin, ks, nCB, i);
  }
}

// bytes: i2r_conv2d_packed_h2_bytes(Cin, Cout, ksize) -- the transposed problem's size
extern "C" int i2r_conv2d_pack_weight_h2_dgrad(const float* w, void* wp, int Cout, int Cin, int ksize,
                                               i2r_stream_t stream) {
  I2R_REQUIRE(w && wp && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3), "pack_weight_h2_dgrad: bad argument");
  long total = (long)h2_chunks(Cout, ksize) * Cin * h2_kc(ksize);
  int nb = i2r_cdiv(total, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(pack_weight_h2_dgrad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)wp, Cout,
                     Cin, ksize, h2_ncb(Cout, ksize), total);
  return i2r_check_launch("pack_weight_h2_dgrad");
}

extern "C" int i2r_conv2d_pack_h2_block_pairs(void) { return H2_PACK_BLOCK; }

extern "C" int i2r_conv2d_pack_h2_multi(const void* table, const void* blockmap, int nblocks, i2r_stream_t stream) {
  I2R_REQUIRE(table && blockmap && nblocks > 0, "pack_h2_multi: bad argument");
  hipLaunchKernelGGL(pack_weight_h2_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                     (const long*)table, (const int2*)blockmap);
  return i2r_check_launch("pack_h2_multi");
}

// positions per workgroup: 256 (8 waves, one workgroup per CU) once that still gives every CU >= 2 workgroups of work,
// else 128 (4 waves, two workgroups per CU)
static int h2_bn(long Npos, int Cout, int ksize) {
  if (const char* ev = getenv("I2R_H2_BN")) {  // tuning/debug override
    int v = atoi(ev);
    if (v == 128 || v == 256) return v;
  }
  if (ksize == 1) return 128;  // short K loops: two independent workgroups per CU overlap better (measured)
  return ((long)i2r_cdiv(Npos, 256) * i2r_cdiv(Cout, 128) >= 512) ? 256 : 128;
}

static void h2_plan(long Npos, int Cout, int ksize, int chunks, bool allow_split, int& bn, int& nsplit, int& cps) {
  bn = h2_bn(Npos, Cout, ksize);
  long tiles = (long)i2r_cdiv(Npos, bn) * i2r_cdiv(Cout, 128);
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
  int bn, nsplit, cps;
  long Npos = (long)B * H * W;
  h2_plan(Npos, Cout, ksize, h2_chunks(Cin, ksize), true, bn, nsplit, cps);
  return nsplit > 1 ? (long)nsplit * Npos * Cout : 0;
}

static int h2_launch(const float* x, const float* x2, int Cin1, const void* wp, const float* bias,
                     const float* residual, float* out, float* splitk_ws, long splitk_floats, int B, int Cin, int H,
                     int W, int Cout, int ksize, int x_is_split, i2r_stream_t stream, const float* in_scale = nullptr);

extern "C" int i2r_conv2d_fwd_h2(const float* x, const void* wp, const float* bias, const float* residual,
                                 float* out, float* splitk_ws, long splitk_floats, int B, int Cin, int H, int W,
                                 int Cout, int ksize, int x_is_split, i2r_stream_t stream) {
  return h2_launch(x, nullptr, Cin, wp, bias, residual, out, splitk_ws, splitk_floats, B, Cin, H, W, Cout, ksize,
                   x_is_split, stream);
}

// Data gradient on the f16x2-split kernel: dx (B, Cin, H, W) = conv2d(dy (B, Cout, H, W), Wd).  `scale`: device
// float[2] {s, 1/s} from i2r_absmax_pow2_scale_f32(dy): gradients sit far below fp16's normal range; multiplied by the
// power of two s they are split at full precision, and the result is multiplied back (both exact).
extern "C" int i2r_conv2d_dgrad_h2(const float* dy, const void* wpd, float* dx, const float* scale, float* splitk_ws,
                                   long splitk_floats, int B, int Cin, int H, int W, int Cout, int ksize,
                                   i2r_stream_t stream) {
  I2R_REQUIRE(scale != nullptr, "conv2d_dgrad_h2: needs the {s, 1/s} pair of i2r_absmax_pow2_scale_f32");
  return h2_launch(dy, nullptr, Cout, wpd, nullptr, nullptr, dx, splitk_ws, splitk_floats, B, Cout, H, W, Cin, ksize,
                   0, stream, scale);
}

// ---------------------------------------------------------------------------------------------------------------
// The one entry point that allocates nothing (no counterpart in the reference; include/i2r_hip.h).  Its split-K
// slabs live in this file-scope device array instead of a caller workspace.  `noalloc_scratch` below is a HOST-side
// stand-in for the symbol, not a device address: host code only ever uses the per-device pointer that
// hipGetSymbolAddress returns (every GPU holds its own copy of the array at its own address).
#define I2R_NOALLOC_SCRATCH_FLOATS (4L << 20) /* 16 MiB per GPU, held until the process ends */
static __device__ __align__(256) float noalloc_scratch[I2R_NOALLOC_SCRATCH_FLOATS];

#include <mutex>
static float* noalloc_scratch_ptr() {
  static std::mutex mu;
  static float* ptrs[64] = {nullptr};
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
    i2r_set_error("conv2d_fwd_h2_noalloc: cannot identify the current device");
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(mu);
  if (!ptrs[dev]) {  // once per device, with that device current
    float* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(noalloc_scratch));
    if (e != hipSuccess || !ptr) {
      i2r_set_error("conv2d_fwd_h2_noalloc: hipGetSymbolAddress(noalloc_scratch) failed on device %d: %s", dev,
                    hipGetErrorString(e));
      return nullptr;
    }
    ptrs[dev] = ptr;
  }
  return ptrs[dev];
}

extern "C" long i2r_conv2d_fwd_h2_noalloc_max_floats(void) { return I2R_NOALLOC_SCRATCH_FLOATS; }

extern "C" int i2r_conv2d_fwd_h2_noalloc(const float* x, const void* wp, const float* bias, const float* residual,
                                         float* out, int B, int Cin, int H, int W, int Cout, int ksize,
                                         int x_is_split, i2r_stream_t stream) {
  I2R_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && (ksize == 1 || ksize == 3),
              "conv2d_fwd_h2_noalloc: bad dimensions");
  // the same plan i2r_conv2d_fwd_h2 makes when it is handed the workspace i2r_conv2d_splitk_h2_floats asks for
  const bool may_split = !((x_is_split >> 1) & 1) && !((x_is_split >> 3) & 1);
  long want = may_split ? i2r_conv2d_splitk_h2_floats(B, Cin, H, W, Cout, ksize) : 0;
  I2R_REQUIRE(want <= I2R_NOALLOC_SCRATCH_FLOATS,
              "conv2d_fwd_h2_noalloc: this shape (B=%d Cin=%d %dx%d Cout=%d k=%d) wants %ld floats of split-K scratch, "
              "above the built-in %ld; call i2r_conv2d_fwd_h2 with a caller-owned workspace instead",
              B, Cin, H, W, Cout, ksize, want, (long)I2R_NOALLOC_SCRATCH_FLOATS);
  float* scratch = nullptr;
  if (want > 0) {
    scratch = noalloc_scratch_ptr();
    if (!scratch) return I2R_ERR_HIP;
  }
  // every slab element the reduction reads is written by exactly one workgroup of this call first: nothing to clear
  return h2_launch(x, nullptr, Cin, wp, bias, residual, out, scratch, want, B, Cin, H, W, Cout, ksize, x_is_split,
                   stream);
}

// 1x1 convolution over the channel concatenation [x1 | x2] (C1 + C2 input channels) without materialising it.
extern "C" int i2r_conv1x1_cat_fwd_h2(const float* x1, const float* x2, int C1, int C2, const void* wp,
                                      const float* bias, const float* residual, float* out, float* splitk_ws,
                                      long splitk_floats, int B, int H, int W, int Cout, i2r_stream_t stream) {
  I2R_REQUIRE(x1 && x2 && C1 > 0 && C2 > 0 && C1 % 16 == 0,
              "conv1x1_cat_fwd_h2: needs two inputs and C1 (%d) a multiple of 16", C1);
  return h2_launch(x1, x2, C1, wp, bias, residual, out, splitk_ws, splitk_floats, B, C1 + C2, H, W, Cout, 1, 0,
                   stream);
}

static int h2_launch(const float* x, const float* x2, int Cin1, const void* wp, const float* bias,
                     const float* residual, float* out, float* splitk_ws, long splitk_floats, int B, int Cin, int H,
                     int W, int Cout, int ksize, int x_is_split, i2r_stream_t stream, const float* in_scale) {
  I2R_REQUIRE(x && wp && out, "conv2d_fwd_h2: null pointer");
  I2R_REQUIRE(ksize == 1 || ksize == 3, "conv2d_fwd_h2: kernel size %d unsupported (1 or 3)", ksize);
  I2R_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0, "conv2d_fwd_h2: non-positive dimension");
  long x_bytes = (long)B * Cin1 * H * W * 4, x2_bytes = (long)B * (Cin - Cin1) * H * W * 4;
  long wp_bytes = i2r_conv2d_packed_h2_bytes(Cout, Cin, ksize);
  long out_bytes = (long)B * Cout * H * W * 4;
  I2R_REQUIRE(x_bytes < 0x7FFFFFFFL && x2_bytes < 0x7FFFFFFFL && wp_bytes < 0x7FFFFFFFL && out_bytes < 0x7FFFFFFFL,
              "conv2d_fwd_h2: input (%ld B), output (%ld B) or packed weights (%ld B) exceed the 2 GiB "
              "buffer-addressing limit; split the batch", x_bytes, out_bytes, wp_bytes);
  long Npos = (long)B * H * W;
  int chunks = h2_chunks(Cin, ksize), bn, nsplit, cps;
  const int qkv_out = (x_is_split >> 1) & 1;  // flag bit 1: split-pair qkv output for the attention kernel
  const int up_in = (x_is_split >> 2) & 1;    // flag bit 2: x is half resolution, upsampled x2 on the fly
  const int up_res = (x_is_split >> 3) & 1;   // flag bit 3: residual is half resolution, upsampled x2 on the fly
  const int one_term = (x_is_split >> 4) & 1; // flag bit 4: plain-f16 product where such a kernel is built (3x3, split x, DPP)
  x_is_split &= 1;
  I2R_REQUIRE(!(up_in || up_res) || (H % 2 == 0 && W % 2 == 0), "conv2d_fwd_h2: x2 upsampling needs even H, W");
  I2R_REQUIRE(!up_in || (ksize == 3 && !x2 && W <= 32 && (W & (W - 1)) == 0 && !getenv("I2R_H2_NODPP")),
              "conv2d_fwd_h2: fused upsampling is built for the 3x3 DPP path only (W a power of two <= 32)");
  I2R_REQUIRE(!up_res || residual, "conv2d_fwd_h2: up_res without a residual");
  if (up_in) x_bytes /= 4;
  I2R_REQUIRE(!qkv_out || Cout % 3 == 0, "conv2d_fwd_h2: qkv output needs Cout %% 3 == 0 (got %d)", Cout);
  // split-K partials are fp32 and its reduce kernel adds a full-resolution residual: not for qkv_out / up_res
  h2_plan(Npos, Cout, ksize, chunks, splitk_ws != nullptr && !qkv_out && !up_res, bn, nsplit, cps);
  if (nsplit > 1 && (long)nsplit * Npos * Cout > splitk_floats) { nsplit = 1; cps = chunks; }
  ConvHParams p;
  p.x = x; p.x2 = x2; p.wp = wp; p.bias = bias; p.res = residual;
  p.out = nsplit > 1 ? splitk_ws : out;
  p.x_bytes = (unsigned)x_bytes; p.x2_bytes = (unsigned)(x2 ? x2_bytes : x_bytes); p.wp_bytes = (unsigned)wp_bytes;
  p.B = B; p.Cin = Cin; p.Cin1 = Cin1; p.H = H; p.W = W; p.Cout = Cout; p.HW = H * W; p.nCB = h2_ncb(Cin, ksize);
  p.Npos = Npos; p.chunks_per_split = cps; p.nsplit = nsplit; p.slab_stride = Npos * Cout;
  { const char* ab = getenv("I2R_H2_ABLATE"); p.ablate = ab ? atoi(ab) : 0; }
  p.qkv_out = qkv_out; p.up_in = up_in; p.up_res = up_res;
  p.hw1_vec = ksize == 1 && H * W == 1 && (Cin1 & 3) == 0 && ((Cin - Cin1) & 3) == 0 &&
              ((((uintptr_t)x) | ((uintptr_t)(x2 ? x2 : x))) & 15) == 0 && !getenv("I2R_H2_HW1_SCALAR");
  p.in_scale = in_scale;  // (null on every forward call: the un-split path then multiplies by exactly 1)
  hipStream_t s = (hipStream_t)stream;
  p.mtiles = i2r_cdiv(Cout, 128);
  // Cin = 128 / 256 1x1 at inference sizes: weight-stationary 16x16x32 kernel (conv1x1_h2_ws_kernel)
  {
    const char* wse = getenv("I2R_H2_WS");   // (read per call: probes and tests flip it inside one process)
    static const long ws_minpos = getenv("I2R_H2_WS_MINPOS") ? atol(getenv("I2R_H2_WS_MINPOS")) : 128L * 256;   // (65536 until the sampler's half-batch graphs were looked at: +0.8 % on B1, tools/r3_wsmin_ab.sh)
    // (taken without a residual only: measured against the kernels it would replace, qkv 256 -> 768 1.18-1.23x, qkv
    // 128 -> 384 1.15x; with a residual 0.95-1.0x, over a channel concat 1.0-1.02x -- those stay where they were)
    if (ksize == 1 && !(wse && atoi(wse) == 0) && !one_term && !up_in && !residual && !x2 && (Cin == 128 || Cin == 256) &&
        Cout > 64 && (W & 3) == 0 && Npos >= ws_minpos) {
      p.mtiles = i2r_cdiv(Cout, 128);
      if (p.mtiles <= 32) {
        const int nch = Cin / 64;
        const int lds = nch * 2 * 128 * 128 + 512;            // the weight tile + its bias
        const int nworkers = 8 * (32 / p.mtiles);             // per XCD: 32 CUs / Cout tiles, all tiles alike
        dim3 grid(8 * (unsigned)((32 / p.mtiles) * p.mtiles));
        p.out = out;   // (never split-K)
        p.nsplit = 1;
#define H2_WS(SP_, NCH_)                                                         \
  do {                                                                           \
    auto kern = conv1x1_h2_ws_kernel<SP_, NCH_>;                                 \
    static bool attr[64] = {false};                                              \
    if (!i2r_lds_attr((const void*)kern, NCH_ * 2 * 128 * 128 + 512, attr)) return I2R_ERR_HIP; \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p, nworkers);              \
  } while (0)
        if (nch == 2) { if (x_is_split) H2_WS(true, 2); else H2_WS(false, 2); }
        else { if (x_is_split) H2_WS(true, 4); else H2_WS(false, 4); }
#undef H2_WS
        return i2r_check_launch("conv1x1_h2_ws");
      }
    }
  }
  // short-K 1x1: the weight-stationary position loop (conv1x1_h2_persist_kernel)
  const char* pe = getenv("I2R_H2_PERSIST");  // (read per call: the A/B probe flips it inside one process)
  const bool persist_on = !(pe && atoi(pe) == 0);
  if (persist_on && ksize == 1 && !x2 && !residual && chunks <= 2 && nsplit == 1 && !up_in && (W & 3) == 0 &&
      Npos >= 256L * 256) {
    // Launched in spans of at most 1024 position tiles: the Cout-tile workers that share a position range run
    // unsynchronised, and over a longer walk they drift apart until the shared activations have left the L2
    // (measured at 8192 tiles: 0.85x of the tiled kernel; in spans of 1024: 1.3x).
    const int all_tiles = i2r_cdiv(Npos, 256);
    int cap = 256 / p.mtiles;                 // one 8-wave workgroup per CU (110 KB of LDS)
    if (cap < 1) cap = 1;
    const int lds = chunks * 2 * 128 * (64 * 2 + 16) + 8 * 32 * EPI_PITCH * 4;
#define H2_PERSIST(SP_)                                                                                       \
  do {                                                                                                        \
    auto kern = conv1x1_h2_persist_kernel<SP_>;                                                               \
    static bool attr[64] = {false}; \
    if (!i2r_lds_attr((const void*)kern, 2 * 2 * 128 * (64 * 2 + 16) + 8 * 32 * EPI_PITCH * 4, attr)) return I2R_ERR_HIP; \
    for (int tile0 = 0; tile0 < all_tiles; tile0 += 1024) {                                                   \
      const int ntiles = tile0 + 1024 < all_tiles ? tile0 + 1024 : all_tiles;                                 \
      const int rounds = i2r_cdiv(ntiles - tile0, cap);                                                       \
      const int nworkers = i2r_cdiv(ntiles - tile0, rounds); /* each walks `rounds` (or rounds - 1) tiles */  \
      dim3 pgrid((nworkers + 7) / 8 * 8 * p.mtiles);                                                          \
      hipLaunchKernelGGL(kern, pgrid, dim3(512), lds, s, p, nworkers, tile0, ntiles);                         \
    }                                                                                                         \
  } while (0)
    if (x_is_split) H2_PERSIST(true);
    else H2_PERSIST(false);
#undef H2_PERSIST
    return i2r_check_launch("conv1x1_h2_persist");
  }
  // long-K (or residual-adding) 1x1 layers at inference sizes: the 16x16x32 MFMA form (conv1x1_h2_m16_kernel)
  {
    const char* m16k1 = getenv("I2R_H2_M16_K1");  // (read per call: A/B probes and tests flip it inside one process)
    static const long k1_mintiles = getenv("I2R_H2_M16_MINTILES") ? atol(getenv("I2R_H2_M16_MINTILES")) : 256;
    if (ksize == 1 && !(m16k1 && atoi(m16k1) == 0) && !one_term && !up_in && Cin % 64 == 0 && Cin >= 128 &&
        (!x2 || Cin1 % 64 == 0) && Cout > 64 && Cout <= 2048 && (W & 3) == 0 &&
        (long)i2r_cdiv(Npos, 256) * i2r_cdiv(Cout, 128) >= k1_mintiles) {
      p.mtiles = i2r_cdiv(Cout, 128);
      const int lds = 2 * 2 * 128 * 128 + 8 * 32 * EPI_PITCH * 4 + p.mtiles * 512;  // stages, patches, bias
      const long per_xcd = (long)((i2r_cdiv(Npos, 256) + 7) / 8) * p.mtiles;
      dim3 grid(8 * (unsigned)(per_xcd < 32 ? per_xcd : 32));  // one workgroup per CU
      p.out = out;   // (never split-K)
      p.nsplit = 1;
#define H2_M16_K1(SP_)                                                          \
  do {                                                                          \
    auto kern = conv1x1_h2_m16_kernel<SP_>;                                     \
    static bool attr[64] = {false};                                             \
    if (!i2r_lds_attr((const void*)kern, lds - p.mtiles * 512 + 16 * 512, attr)) return I2R_ERR_HIP; \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);                       \
  } while (0)
      if (x_is_split) H2_M16_K1(true);
      else H2_M16_K1(false);
#undef H2_M16_K1
      return i2r_check_launch("conv1x1_h2_m16");
    }
  }
  // 3x3 convs with few output channels run on a shorter Cout tile (the MFMAs of the empty rows are not issued)
  const int bm = (ksize == 3 && !one_term && !getenv("I2R_H2_BM128")) ? (Cout <= 32 ? 32 : Cout <= 64 ? 64 : 128) : 128;
  p.mtiles = i2r_cdiv(Cout, bm);
  // the large 3x3 layers: the 16x16x32 MFMA form (conv3x3_h2_m16_kernel)
  const char* m16 = getenv("I2R_H2_M16");  // (read per call: A/B probes and tests flip it inside one process)
  static const long m16_mintiles = getenv("I2R_H2_M16_MINTILES") ? atol(getenv("I2R_H2_M16_MINTILES")) : 256;
  // (2x2 images from half that size on: their form of the kernel issues a third fewer MFMAs, and the sampler runs such
  // launches of a 2048-series half batch -- 128 tiles -- beside the other half's on a second stream)
  const long m16_tiles = (long)i2r_cdiv(Npos, 256) * i2r_cdiv(Cout, 128);
  const bool m16_size = (bn == 256 && nsplit == 1) || m16_tiles >= m16_mintiles ||
                        (W == 2 && H == 2 && 2 * m16_tiles >= m16_mintiles && !getenv("I2R_H2_M16_W2OFF"));
  if (ksize == 3 && !(m16 && atoi(m16) == 0) && !one_term && !x2 && bm == 128 && Cout > 64 && Cout <= 2048 && m16_size &&
      Cin % 32 == 0 && (W == 4 || W == 8 || W == 16 || W == 32 || (W == 2 && H == 2 && !up_in && !up_res)) &&
      !getenv("I2R_H2_NODPP")) {
    const int lds = 2 * 2 * 128 * (96 * 2) + 8 * 32 * EPI_PITCH * 4 + p.mtiles * 128 * 4;  // stages, patches, bias
    const long per_xcd = (long)((i2r_cdiv(Npos, 256) + 7) / 8) * p.mtiles;  // tile-list entries of one XCD
    // (I2R_H2_M16_CUS: workgroups per XCD, default all 32 CUs -- an experiment switch: fewer leave CUs to a second
    // stream's memory-bound kernels)
    static const long m16_cus = getenv("I2R_H2_M16_CUS") ? atol(getenv("I2R_H2_M16_CUS")) : 32;
    const long cus = m16_cus >= 1 && m16_cus <= 32 ? m16_cus : 32;
    // (I2R_H2_M16_TPW: at least this many tiles per workgroup, default 1 -- an experiment switch: short launches on
    // fewer CUs, each workgroup pipelining over several tiles, the rest of the chip left to the other stream)
    static const long m16_tpw = getenv("I2R_H2_M16_TPW") ? atol(getenv("I2R_H2_M16_TPW")) : 1;
    long wgs = per_xcd < cus ? per_xcd : cus;
    if (m16_tpw > 1 && wgs * m16_tpw > per_xcd) wgs = per_xcd / m16_tpw > 0 ? per_xcd / m16_tpw : 1;
    dim3 grid(8 * (unsigned)wgs);                                          // one workgroup per CU
    p.out = out;  // (never split-K: a plan made for the 128-position tiles may have asked for it)
    p.nsplit = 1;
#define H2_M16(SP_, WM_, UP_)                                                                                   \
  do {                                                                                                          \
    auto kern = conv3x3_h2_m16_kernel<SP_, WM_, UP_>;                                                           \
    static bool attr[64] = {false};                                                                             \
    if (!i2r_lds_attr((const void*)kern, lds - p.mtiles * 512 + 16 * 512, attr)) return I2R_ERR_HIP;            \
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, p);                                                       \
  } while (0)
#define H2_M16_W(SP_)                                  \
  do {                                                 \
    if (W == 32 && up_in) H2_M16(SP_, 2, true);        \
    else if (W == 32) H2_M16(SP_, 2, false);           \
    else if (W == 2 && !getenv("I2R_H2_M16_W2OFF")) H2_M16(SP_, 1, false); \
    else if (up_in) H2_M16(SP_, 0, true);              \
    else H2_M16(SP_, 0, false);                        \
  } while (0)
    if (x_is_split) H2_M16_W(true);
    else H2_M16_W(false);
#undef H2_M16_W
#undef H2_M16
    return i2r_check_launch("conv3x3_h2_m16");
  }
  dim3 grid((i2r_cdiv(Npos, bn) + 7) / 8 * 8 * p.mtiles, 1, nsplit);
  // two weight stages (hi + lo rows each); the quad epilogue's per-wave transpose patches reuse them
  const int patches = (bn / 32) * 32 * EPI_PITCH * 4;
  const int lds3 = max(4 * bm * (48 * 2 + 16), patches), lds1 = 4 * 128 * (64 * 2 + 16);
  // DPP tap reuse needs lanes = consecutive positions of one image row: W a power of two <= 32
  const bool dpp = ksize == 3 && W <= 32 && (W & (W - 1)) == 0 && !getenv("I2R_H2_NODPP");
#define H2_LAUNCH(KS_, SP_, NW_, DP_, TM_, BM_)                                                                    \
  do {                                                                                                             \
    auto kern = conv_fwd_h2_kernel<KS_, SP_, NW_, DP_, TM_, BM_>;                                                  \
    static bool attr[64] = {false}; \
    if (!i2r_lds_attr((const void*)kern, KS_ == 3 ? 4 * 128 * (48 * 2 + 16) : lds1, attr)) return I2R_ERR_HIP; \
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW_), KS_ == 3 ? lds3 : lds1, s, p);                                  \
  } while (0)
#define H2_PICK_NW(KS_, SP_, DP_, TM_, BM_)            \
  do {                                                 \
    if (bn == 256) H2_LAUNCH(KS_, SP_, 8, DP_, TM_, BM_); \
    else H2_LAUNCH(KS_, SP_, 4, DP_, TM_, BM_);        \
  } while (0)
#define H2_PICK_BM(SP_, DP_)                           \
  do {                                                 \
    if (bm == 32) H2_PICK_NW(3, SP_, DP_, 3, 32);      \
    else if (bm == 64) H2_PICK_NW(3, SP_, DP_, 3, 64); \
    else H2_PICK_NW(3, SP_, DP_, 3, 128);              \
  } while (0)
  if (ksize == 3) {
    if (x_is_split) {
      if (dpp) { if (one_term) H2_PICK_NW(3, true, true, 1, 128); else H2_PICK_BM(true, true); }
      else H2_PICK_BM(true, false);
    } else { if (dpp) H2_PICK_BM(false, true); else H2_PICK_BM(false, false); }
  } else {
    if (x_is_split) H2_PICK_NW(1, true, false, 3, 128);
    else H2_PICK_NW(1, false, false, 3, 128);
  }
#undef H2_PICK_BM
#undef H2_PICK_NW
#undef H2_LAUNCH
  if (nsplit > 1) {
    long total = Npos * Cout;
    int nb = i2r_cdiv(total, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(splitk_reduce_h2_kernel, dim3(nb), dim3(256), 0, s, splitk_ws, p.slab_stride, nsplit, bias,
                       residual, out, total, Cout, H * W);
  }
  return i2r_check_launch("conv2d_fwd_h2");
}
