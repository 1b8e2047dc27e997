// Low-resolution self-attention of the ADM UNet (models/networks.py:113-126, 181-184), fp32, head dim 64.
// Input is the qkv 1x1-conv output (B, 3C, T) whose channels are interleaved (head, d, {q,k,v}) exactly as the
// reference's reshape(B*h, C/h, 3, T).unbind(2) reads them; output is a (B, C, T) with channel = head*64 + c.
//
// T >= 64 (multiple of 32): flash-style kernel on the fp32 MFMA (v_mfma_f32_32x32x2_f32).  One wave owns 32 queries.
//   S^T = K^T Q is computed with the QUERY on the MFMA column (= lane): a lane's 16 accumulator registers are 16 of
//   the 32 keys of its own query, its partner lane (l ^ 32) holds the other 16 -> the softmax row reduction is 15
//   in-register ops + one cross-half shuffle, and the per-query rescale of O is a per-lane scalar.
//   P (registers) is fed straight back as the B operand of O += V P: the product reduces over the accumulator's row
//   index, so the k-order of that MFMA chain is simply permuted to the order the rows sit in the registers
//   (key(kk, h) = 4h + (kk & 3) + 8 (kk >> 2)) and V is read from LDS in the same order.  The T x T score matrix the
//   reference materialises (268 MB per block at T = 1024, B = 32) never exists.
// T <= 32: one 64-thread block per (sample, head) entirely in LDS.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

#define HD 64          // head dim (channels_per_head, networks.py:393)
#define KT 32          // keys per tile
#define VLD (KT + 1)   // padded row of the V / K^T images read with the channel on the lane

template <int QW>
__global__ __launch_bounds__(64 * QW) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           float* __restrict__ lse, int heads, int T) {
  __shared__ float Ks[HD * KT];
  __shared__ float Vs[HD * VLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int C = heads * HD;
  const float* base = qkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;  // row (d, s) at base + (3 d + s) T
  const int q0 = (blockIdx.x * QW + wave) * 32;

  float qf[HD / 2];
#pragma unroll
  for (int kk = 0; kk < HD / 2; ++kk) qf[kk] = base[(long)(3 * (2 * kk + h) + 0) * T + q0 + l31] * 0.125f;

  f32x16 O[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[t][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  for (int k0 = 0; k0 < T; k0 += KT) {
    __syncthreads();
    for (int idx = tid; idx < HD * KT; idx += 64 * QW) {
      int d = idx >> 5, key = idx & 31;
      Ks[d * KT + key] = base[(long)(3 * d + 1) * T + k0 + key];
      Vs[d * VLD + key] = base[(long)(3 * d + 2) * T + k0 + key];
    }
    __syncthreads();
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < HD / 2; ++kk)
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * kk + h) * KT + l31], qf[kk], s, 0, 0, 0);
    float mt = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    float m_new = fmaxf(m_run, mt);
    float alpha = expf(m_run - m_new);
    float psum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = expf(s[r] - m_new);
      psum += s[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      int key = 4 * h + (kk & 3) + 8 * (kk >> 2);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        O[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(t * 32 + l31) * VLD + key], s[kk], O[t], 0, 0, 0);
    }
  }
  float inv_l = 1.0f / l_run;
  float* ob = out + ((long)b * C + (long)hd * HD) * T;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      ob[(long)c * T + q0 + l31] = O[t][r] * inv_l;
    }
  if (lse && h == 0) lse[(long)bh * T + q0 + l31] = m_run + logf(l_run);
}

// T <= TM (TM = 4, 16 or 32): everything in LDS, 64 threads per (sample, head).  The head's q/k/v slab is one
// contiguous run of 192*T floats and its output one of 64*T: both are walked linearly (consecutive lanes, consecutive
// addresses), and LDS is sized by TM so that the 2x2 / 4x4 levels keep 10+ workgroups per CU in flight -- at batch
// 4096 these launches stream 270 MB each and were latency-bound at 1.2 TB/s with the fixed 32-wide layout.
// T a multiple of 4 (the 2x2 and 4x4 levels): 16-byte global accesses and four outputs per thread and inner step (two
// LDS reads per four multiply-adds instead of eight).  Every element is still the same sum in the same order as in the
// general form below (the 1/8 on k is a power of two, so applying it while staging changes no bit), so the results are
// the same.  Scores are kept TRANSPOSED (St[key][query]) so that the second product reads four queries of a key with
// one access.  TC: T when it is known at compile time (all loops unroll, the twelve loads of a thread are in flight
// together), 0 otherwise.
template <int TM, int TC>
__device__ __forceinline__ void attn_small_vec(const float* __restrict__ base, float* __restrict__ ob, float* __restrict__ lse_row,
                                               float (*QKV)[HD * TM], float* S, int Trt, int tid) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int T = TC ? TC : Trt;
  const f4* base4 = reinterpret_cast<const f4*>(base);
  const int T4 = T >> 2;
  constexpr int SP = TM + 4;
  constexpr int UNR = TC ? 64 : 1;   // (full unrolling when the trip counts are constants, none otherwise)
#pragma unroll UNR
  for (int it = 0; it < (3 * HD * T4 + 63) / 64; ++it) {
    const int i4 = tid + 64 * it;
    if (i4 >= 3 * HD * T4) break;
    const int row = i4 / T4, t = (i4 - row * T4) * 4;
    const int d = row / 3, sel = row - 3 * d;
    f4 v = base4[i4];
    if (sel == 1) v *= 0.125f;
    *reinterpret_cast<f4*>(&QKV[sel][d * TM + t]) = v;
  }
  __syncthreads();
  const float* Q = QKV[0];
  const float* K = QKV[1];
  const float* V = QKV[2];
#pragma unroll UNR
  for (int it = 0; it < (T * T4 + 63) / 64; ++it) {
    const int idx = tid + 64 * it;
    if (idx >= T * T4) break;
    const int qi = idx / T4, ki = (idx - qi * T4) * 4;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 16
    for (int d = 0; d < HD; ++d) {
      const float q = Q[d * TM + qi];
      const f4 k4 = *reinterpret_cast<const f4*>(&K[d * TM + ki]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += q * k4[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) S[(ki + j) * SP + qi] = acc[j];
  }
  __syncthreads();
  if (tid < T) {   // softmax over the keys of query `tid` (a column of St)
    float m = S[tid];
#pragma unroll UNR
    for (int k = 1; k < T; ++k) m = fmaxf(m, S[k * SP + tid]);
    float sum = 0.0f;
#pragma unroll UNR
    for (int k = 0; k < T; ++k) {
      float e = expf(S[k * SP + tid] - m);
      S[k * SP + tid] = e;
      sum += e;
    }
    float inv = 1.0f / sum;
#pragma unroll UNR
    for (int k = 0; k < T; ++k) S[k * SP + tid] *= inv;
    if (lse_row) lse_row[tid] = m + logf(sum);
  }
  __syncthreads();
#pragma unroll UNR
  for (int it = 0; it < (HD * T4 + 63) / 64; ++it) {
    const int idx = tid + 64 * it;
    if (idx >= HD * T4) break;
    const int c = idx / T4, qi = (idx - c * T4) * 4;
    f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll UNR
    for (int k = 0; k < T; ++k) {
      const float v = V[c * TM + k];
      const f4 p4 = *reinterpret_cast<const f4*>(&S[k * SP + qi]);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += p4[j] * v;
    }
    *reinterpret_cast<f4*>(ob + c * T + qi) = acc;
    // In the fully unrolled form the next item's `acc = 0` lands in the SAME registers right behind this 16-byte
    // store, and on gfx950 nothing keeps a VALU write away from the data registers of a store that has been issued
    // but has not read them yet (the trap of the f16x2 conv epilogues): alone the kernel was always
    // right; with the vector-memory path backed up by another stream's attention kernel on the same CU, 22 of 8192
    // workgroups stored zeros / the next item's partial sums (tools/r3_det_victim3.py) -- and the two-stream sampler
    // stopped being reproducible.  Two wait states tied to the data registers.
    asm volatile("s_nop 1" : "+v"(acc) : : "memory");
  }
}

template <int TM>
__global__ __launch_bounds__(64) void attn_fwd_small_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                            float* __restrict__ lse, int heads, int T) {
  __shared__ __attribute__((aligned(16))) float QKV[3][HD * TM];
  __shared__ __attribute__((aligned(16))) float S[TM * (TM + 4)];
  const int tid = threadIdx.x;
  const int bh = blockIdx.x, b = bh / heads, hd = bh % heads;
  const int C = heads * HD;
  const float* base = qkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;  // row 3 d + s (s: q, k, v), T floats each
  float* ob = out + ((long)b * C + (long)hd * HD) * T;  // (channel, query) row-major: idx = c * T + qi
  if (TM <= 16 && T == TM) {
    attn_small_vec<TM, (TM <= 16 ? TM : 0)>(base, ob, lse ? lse + (long)bh * T : nullptr, QKV, S, T, tid);
    return;
  }
  if ((T & 3) == 0) {
    attn_small_vec<TM, 0>(base, ob, lse ? lse + (long)bh * T : nullptr, QKV, S, T, tid);
    return;
  }
  for (int idx = tid; idx < 3 * HD * T; idx += 64) {
    int row = idx / T, t = idx - row * T;
    int d = row / 3, sel = row - 3 * d;
    QKV[sel][d * TM + t] = base[idx];
  }
  __syncthreads();
  const float* Q = QKV[0];
  const float* K = QKV[1];
  const float* V = QKV[2];
  for (int idx = tid; idx < T * T; idx += 64) {
    int qi = idx / T, ki = idx - qi * T;
    float acc = 0.0f;
    for (int d = 0; d < HD; ++d) acc += Q[d * TM + qi] * (K[d * TM + ki] * 0.125f);
    S[qi * (TM + 1) + ki] = acc;
  }
  __syncthreads();
  if (tid < T) {
    float* row = S + tid * (TM + 1);
    float m = row[0];
    for (int k = 1; k < T; ++k) m = fmaxf(m, row[k]);
    float sum = 0.0f;
    for (int k = 0; k < T; ++k) {
      float e = expf(row[k] - m);
      row[k] = e;
      sum += e;
    }
    float inv = 1.0f / sum;
    for (int k = 0; k < T; ++k) row[k] *= inv;
    if (lse) lse[(long)bh * T + tid] = m + logf(sum);
  }
  __syncthreads();
  for (int idx = tid; idx < HD * T; idx += 64) {
    int c = idx / T, qi = idx - c * T;
    float acc = 0.0f;
    for (int k = 0; k < T; ++k) acc += S[qi * (TM + 1) + k] * V[c * TM + k];
    ob[idx] = acc;
  }
}

// ------------------------------------------------------------------------------ inference forward on the f16 MFMA
// Same flash structure with every product on the HALF-precision matrix cores at fp32-grade accuracy (the "f16x2
// split" of conv_fwd_h2.hip: x = hi + lo in fp16, a*b = a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulate): 24
// v_mfma_f32_32x32x16_f16 (768 cycles) per 32-key tile instead of 64 v_mfma_f32_32x32x2_f32 (4096 cycles).
//   S^T = K^T Q : A = K^T tile from LDS ([key][d] rows, 16-B fragments), B = Q fragments held in registers.
//   O  += V P   : B = P straight from the S accumulator (registers 8s..8s+7 -> k-step s; element j of lane half h is
//                 key 16s + 8(j>>2) + 4h + (j&3)), A = V tile from an LDS image stored in exactly that key order.
// K/V tiles are split to fp16 hi/lo once, while being staged (double buffered, one barrier per tile).
typedef _Float16 ah8 __attribute__((ext_vector_type(8)));
typedef unsigned au4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void a_split8(const float* v, ah8& hi, ah8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 hh, ll;
    f16_split(v[j], hh, ll);
    hi[j] = hh;
    lo[j] = ll;
  }
}

// hi/lo split of softmax weights (plain fp32 values in [0, 1], not products: the double-rounding trap f16_split()
// guards against cannot occur), two elements per conversion instruction (v_cvt_pk_f16_f32, round to nearest)
typedef float af2 __attribute__((ext_vector_type(2)));
typedef _Float16 ah2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void p_split8(const float* v, ah8& hi, ah8& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    af2 x = {v[2 * i], v[2 * i + 1]};
    ah2 h2 = __builtin_convertvector(x, ah2);
    af2 r = x - __builtin_convertvector(h2, af2);
    ah2 l2 = __builtin_convertvector(r, ah2);
    _Float16 h0 = h2.x, h1 = h2.y, l0 = l2.x, l1 = l2.y;
    hi[2 * i] = h0; hi[2 * i + 1] = h1;
    lo[2 * i] = l0; lo[2 * i + 1] = l1;
  }
}

#define AKROW 144  // bytes per key row of the K image: 64 d x 2 B + 16 pad (odd multiple of 16 B)
#define AVROW 80   // bytes per channel row of the V image: 32 keys x 2 B + 16 pad

// words that already hold f16 hi | f16 lo << 16 (written by the qkv conv's epilogue): fragments by byte permutes only
__device__ __forceinline__ void a_unpack8(const float* v, ah8& hi, ah8& lo) {
  unsigned hw[4], lw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned w0 = __builtin_bit_cast(unsigned, v[2 * i]), w1 = __builtin_bit_cast(unsigned, v[2 * i + 1]);
    hw[i] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
    lw[i] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
  }
  au4 hv = {hw[0], hw[1], hw[2], hw[3]}, lv = {lw[0], lw[1], lw[2], lw[3]};
  hi = __builtin_bit_cast(ah8, hv);
  lo = __builtin_bit_cast(ah8, lv);
}

// PRE: qkv arrives pre-split (and q pre-scaled) from i2r_conv2d_fwd_h2's qkv output mode.
template <int QW, bool PRE>
__global__ __launch_bounds__(64 * QW, QW == 4 ? 3 : QW) void attn_fwd_h2_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                              float* __restrict__ lse, int heads, int T) {
  constexpr int NT = 64 * QW, NIT = 256 / NT;                 // staging work items per thread
  constexpr int KBYTES = KT * AKROW, VBYTES = HD * AVROW;     // one of hi / lo
  constexpr int STAGE = 2 * KBYTES + 2 * VBYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int C = heads * HD;
  const float* base = qkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;  // row (d, s) at base + (3 d + s) T
  const int q0 = (blockIdx.x * QW + wave) * 32;

  // Q fragments, pre-scaled by 1/sqrt(64) * log2(e) so the softmax runs on exp2
  ah8 qh[4], ql[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = base[(long)(3 * (16 * s + 8 * h + j) + 0) * T + q0 + l31];
      if (!PRE) v[j] *= I2R_QSCALE;
    }
    if (PRE) a_unpack8(v, qh[s], ql[s]);
    else a_split8(v, qh[s], ql[s]);
  }

  f32x16 O[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[t][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;

  // staging: item = (key, d-group of 8) for K; (channel, piece = k-step s2, half h2) for V.  Loads go through a
  // buffer descriptor over this head's 192 x T slab: per-thread byte offsets are loop invariant, the key tile advances
  // through the scalar offset -- no 64-bit address arithmetic in the loop (it was ~40 % of the loop's VALU work).
  __amdgpu_buffer_rsrc_t rQKV = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 3 * HD * T * 4, 0x00020000);
  unsigned voffK[NIT][8], voffV[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    int item = tid + NT * it;
    int key = item & 31, dg = item >> 5;
#pragma unroll
    for (int j = 0; j < 8; ++j) voffK[it][j] = (unsigned)(((3 * (8 * dg + j) + 1) * T + key) * 4);
    int c = item >> 2, piece = item & 3, s2 = piece >> 1, h2 = piece & 1;
    voffV[it] = (unsigned)(((3 * c + 2) * T + 16 * s2 + 4 * h2) * 4);
  }
  float kreg[NIT][8];
  float4 vreg[NIT][2];
  auto stage_load = [&](int k0) {
    const unsigned soff = (unsigned)k0 * 4;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        kreg[it][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rQKV, voffK[it][j], soff, 0));
      vreg[it][0] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rQKV, voffV[it], soff, 0));
      vreg[it][1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rQKV, voffV[it] + 32, soff, 0));
    }
  };
  auto stage_store = [&](int buf) {
    unsigned char* st = lds + buf * STAGE;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      int item = tid + NT * it;
      int key = item & 31, dg = item >> 5;
      ah8 hi, lo;
      if (PRE) a_unpack8(kreg[it], hi, lo);
      else a_split8(kreg[it], hi, lo);
      *reinterpret_cast<ah8*>(st + key * AKROW + dg * 16) = hi;
      *reinterpret_cast<ah8*>(st + KBYTES + key * AKROW + dg * 16) = lo;
      int c = item >> 2, piece = item & 3;
      float vv[8] = {vreg[it][0].x, vreg[it][0].y, vreg[it][0].z, vreg[it][0].w,
                     vreg[it][1].x, vreg[it][1].y, vreg[it][1].z, vreg[it][1].w};
      if (PRE) a_unpack8(vv, hi, lo);
      else a_split8(vv, hi, lo);
      *reinterpret_cast<ah8*>(st + 2 * KBYTES + c * AVROW + piece * 16) = hi;
      *reinterpret_cast<ah8*>(st + 2 * KBYTES + VBYTES + c * AVROW + piece * 16) = lo;
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
  const int ntiles = T / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const unsigned char* st = lds + (kt & 1) * STAGE;
    if (kt + 1 < ntiles) stage_load((kt + 1) * KT);
    // ---- S^T = K^T Q  (rows = keys, cols = this lane's query)
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      ah8 kh = *reinterpret_cast<const ah8*>(st + l31 * AKROW + s * 32 + h * 16);
      ah8 kl = *reinterpret_cast<const ah8*>(st + KBYTES + l31 * AKROW + s * 32 + h * 16);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sacc, 0, 0, 0);
    }
    // ---- online softmax in base 2 (scores are already in log2 units)
    float mt = sacc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    // the kernel is VALU-bound (profiles/r01_pmc_attention_T1024.txt): the bare v_exp_f32 is enough here -- every
    // argument is <= 0, and a result below 2^-126 may flush to 0 -- where exp2f() expands to ~6 instructions
    float m_new = fmaxf(m_run, mt);
    float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.0f;
    float pv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pv[r] = __builtin_amdgcn_exp2f(sacc[r] - m_new);
      psum += pv[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // after the first tiles the running maximum rarely moves: skip the 32 multiplies by 1 (wave-uniform branch)
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[t][r] *= alpha;
    }
    // ---- O += V P
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ah8 ph, pl;
      p_split8(pv + 8 * s, ph, pl);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned char* vrow = st + 2 * KBYTES + (t * 32 + l31) * AVROW + (2 * s + h) * 16;
        ah8 vh = *reinterpret_cast<const ah8*>(vrow);
        ah8 vl = *reinterpret_cast<const ah8*>(vrow + VBYTES);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, O[t], 0, 0, 0);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, O[t], 0, 0, 0);
        O[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, O[t], 0, 0, 0);
      }
    }
    if (kt + 1 < ntiles) stage_store((kt + 1) & 1);  // the other stage: its readers finished before the last barrier
    __syncthreads();
  }
  float inv_l = 1.0f / l_run;
  float* ob = out + ((long)b * C + (long)hd * HD) * T;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      ob[(long)c * T + q0 + l31] = O[t][r] * inv_l;
    }
  // training forward: the log-sum-exp the backward recomputes the probabilities from, in natural-log units like
  // attn_fwd_kernel's (the scores here are in log2 units: q carries log2(e))
  if (lse && h == 0) lse[(long)bh * T + q0 + l31] = (m_run + log2f(l_run)) * 0.6931471805599453f;
}

// f16x2-split inference forward; T must be a multiple of 32 and > 32 (smaller T: i2r_attention_fwd_f32's LDS kernel).
// presplit: qkv was written by i2r_conv2d_fwd_h2 with flag bit 1 (split pairs, q pre-scaled).
static int attention_fwd_h2_launch(const float* qkv, float* out, float* lse, int B, int heads, int T, int presplit,
                                   i2r_stream_t stream) {
  I2R_REQUIRE(qkv && out && B > 0 && heads > 0 && T > 32 && T % 32 == 0, "attention_fwd_h2: bad argument (T=%d)", T);
  I2R_REQUIRE((((uintptr_t)qkv) & 15) == 0, "attention_fwd_h2: qkv must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
#define ATT_GO(QW_, PRE_)                                                                                          \
  hipLaunchKernelGGL((attn_fwd_h2_kernel<QW_, PRE_>), dim3(T / (32 * QW_), B * heads), dim3(64 * QW_), 0, s, qkv, out, \
                     lse, heads, T)
  if (T % 128 == 0) { if (presplit) ATT_GO(4, true); else ATT_GO(4, false); }
  else if (T % 64 == 0) { if (presplit) ATT_GO(2, true); else ATT_GO(2, false); }
  else { if (presplit) ATT_GO(1, true); else ATT_GO(1, false); }
#undef ATT_GO
  return i2r_check_launch("attention_fwd_h2");
}

extern "C" int i2r_attention_fwd_h2(const float* qkv, float* out, int B, int heads, int T, int presplit,
                                    i2r_stream_t stream) {
  return attention_fwd_h2_launch(qkv, out, nullptr, B, heads, T, presplit, stream);
}

// The same kernel as a TRAINING forward: plain fp32 qkv in, and the per-query log-sum-exp out (B * heads * T floats) for
// i2r_attention_bwd_f32, which recomputes the probabilities from it.
extern "C" int i2r_attention_fwd_h2_lse(const float* qkv, float* out, float* lse, int B, int heads, int T,
                                        i2r_stream_t stream) {
  I2R_REQUIRE(lse, "attention_fwd_h2_lse: null lse");
  return attention_fwd_h2_launch(qkv, out, lse, B, heads, T, 0, stream);
}

extern "C" int i2r_attention_fwd_f32(const float* qkv, float* out, float* lse, int B, int heads, int T,
                                     i2r_stream_t stream) {
  I2R_REQUIRE(qkv && out && B > 0 && heads > 0 && T > 0, "attention_fwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (T <= 32) {
    if (T <= 4)
      hipLaunchKernelGGL(attn_fwd_small_kernel<4>, dim3(B * heads), dim3(64), 0, s, qkv, out, lse, heads, T);
    else if (T <= 16)
      hipLaunchKernelGGL(attn_fwd_small_kernel<16>, dim3(B * heads), dim3(64), 0, s, qkv, out, lse, heads, T);
    else
      hipLaunchKernelGGL(attn_fwd_small_kernel<32>, dim3(B * heads), dim3(64), 0, s, qkv, out, lse, heads, T);
  } else {
    I2R_REQUIRE(T % 32 == 0, "attention_fwd: T=%d must be <= 32 or a multiple of 32", T);
    if (T % 128 == 0)
      hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3(T / 128, B * heads), dim3(256), 0, s, qkv, out, lse, heads, T);
    else if (T % 64 == 0)
      hipLaunchKernelGGL(attn_fwd_kernel<2>, dim3(T / 64, B * heads), dim3(128), 0, s, qkv, out, lse, heads, T);
    else
      hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3(T / 32, B * heads), dim3(64), 0, s, qkv, out, lse, heads, T);
  }
  return i2r_check_launch("attention_fwd");
}

// --------------------------------------------------------------------------------------------------- backward
// AttentionOp.backward (networks.py:121-126) + the backward of einsum('nqk,nck->ncq', w, v) (:184), flash-style: the
// probabilities are recomputed from q, k and the forward's log-sum-exp instead of being saved.
//   delta[q] = sum_c da[c][q] a[c][q];  P = exp(S - lse);  dP[q][k] = sum_c da[c][q] v[c][k];  dS = P (dP - delta)
//   dv[c][k] = sum_q da[c][q] P[q][k];  dq[d][q] = sum_k dS[q][k] k[d][k] / 8;  dk[d][k] = sum_q dS[q][k] q[d][q] / 8
// Two deterministic kernels (no atomics): one owns 32 KEYS per wave and produces dk, dv; one owns 32 QUERIES per wave
// and produces dq.  In each, the MFMA whose result feeds the next product is oriented so that the next product
// reduces over its accumulator ROW index (registers) -- no cross-lane movement between the chained MFMAs.
#define QLD 33

__global__ void attn_delta_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                  float* __restrict__ delta, int heads, int T, long total) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  int q = (int)(i % T);
  long bh = i / T;
  const float* a = out + bh * HD * T + q;  // (b*C + hd*64) * T == bh * 64 * T since C = heads*64
  const float* da = dout + bh * HD * T + q;
  float acc = 0.0f;
  for (int c = 0; c < HD; ++c) acc += a[(long)c * T] * da[(long)c * T];
  delta[i] = acc;
}

template <int QW>
__global__ __launch_bounds__(64 * QW, 2) void attn_bwd_dkv_kernel(const float* __restrict__ qkv,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               const float* __restrict__ dout,
                                                               float* __restrict__ dqkv, int heads, int T) {
  __shared__ float Qs[HD * QLD];
  __shared__ float dAs[HD * QLD];
  __shared__ float lse_s[32], del_s[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int C = heads * HD;
  const float* base = qkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  float* dbase = dqkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  const float* dab = dout + ((long)b * C + (long)hd * HD) * T;
  const int key0 = (blockIdx.x * QW + wave) * 32;

  float kf[HD / 2], vf[HD / 2];
#pragma unroll
  for (int kk = 0; kk < HD / 2; ++kk) {
    kf[kk] = base[(long)(3 * (2 * kk + h) + 1) * T + key0 + l31] * 0.125f;
    vf[kk] = base[(long)(3 * (2 * kk + h) + 2) * T + key0 + l31];
  }
  f32x16 dK[2], dV[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dK[t][r] = dV[t][r] = 0.0f;

  for (int q0 = 0; q0 < T; q0 += 32) {
    __syncthreads();
    for (int idx = tid; idx < HD * 32; idx += 64 * QW) {
      int d = idx >> 5, q = idx & 31;
      Qs[d * QLD + q] = base[(long)(3 * d + 0) * T + q0 + q];
      dAs[d * QLD + q] = dab[(long)d * T + q0 + q];
    }
    if (tid < 32) {
      lse_s[tid] = lse[(long)bh * T + q0 + tid];
      del_s[tid] = delta[(long)bh * T + q0 + tid];
    }
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < HD / 2; ++kk) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[(2 * kk + h) * QLD + l31], kf[kk], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(dAs[(2 * kk + h) * QLD + l31], vf[kk], dp, 0, 0, 0);
    }
    // rows = query q(r,h), cols = key (lane)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int q = (r & 3) + 8 * (r >> 2) + 4 * h;
      float pr = expf(s[r] - lse_s[q]);
      s[r] = pr;                          // P
      dp[r] = pr * (dp[r] - del_s[q]);    // dS
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      int q = 4 * h + (kk & 3) + 8 * (kk >> 2);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dV[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(dAs[(t * 32 + l31) * QLD + q], s[kk], dV[t], 0, 0, 0);
        dK[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[(t * 32 + l31) * QLD + q], dp[kk], dK[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      dbase[(long)(3 * d + 1) * T + key0 + l31] = dK[t][r] * 0.125f;
      dbase[(long)(3 * d + 2) * T + key0 + l31] = dV[t][r];
    }
}

template <int QW>
__global__ __launch_bounds__(64 * QW) void attn_bwd_dq_kernel(const float* __restrict__ qkv,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ delta,
                                                              const float* __restrict__ dout,
                                                              float* __restrict__ dqkv, int heads, int T) {
  __shared__ float Ks[HD * QLD];
  __shared__ float Vs[HD * QLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int C = heads * HD;
  const float* base = qkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  float* dbase = dqkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  const float* dab = dout + ((long)b * C + (long)hd * HD) * T;
  const int q0 = (blockIdx.x * QW + wave) * 32;

  float qf[HD / 2], daf[HD / 2];
#pragma unroll
  for (int kk = 0; kk < HD / 2; ++kk) {
    qf[kk] = base[(long)(3 * (2 * kk + h) + 0) * T + q0 + l31] * 0.125f;
    daf[kk] = dab[(long)(2 * kk + h) * T + q0 + l31];
  }
  const float lse_q = lse[(long)bh * T + q0 + l31];
  const float del_q = delta[(long)bh * T + q0 + l31];
  f32x16 dQ[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dQ[t][r] = 0.0f;

  for (int k0 = 0; k0 < T; k0 += 32) {
    __syncthreads();
    for (int idx = tid; idx < HD * 32; idx += 64 * QW) {
      int d = idx >> 5, key = idx & 31;
      Ks[d * QLD + key] = base[(long)(3 * d + 1) * T + k0 + key];
      Vs[d * QLD + key] = base[(long)(3 * d + 2) * T + k0 + key];
    }
    __syncthreads();
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < HD / 2; ++kk) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * kk + h) * QLD + l31], qf[kk], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(2 * kk + h) * QLD + l31], daf[kk], dp, 0, 0, 0);
    }
    // rows = key(r,h), cols = query (lane)
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = expf(s[r] - lse_q) * (dp[r] - del_q);  // dS^T
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      int key = 4 * h + (kk & 3) + 8 * (kk >> 2);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        dQ[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(t * 32 + l31) * QLD + key], dp[kk], dQ[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      dbase[(long)(3 * d + 0) * T + q0 + l31] = dQ[t][r] * 0.125f;
    }
}

// T <= 32: one 256-thread block per (sample, head), everything in LDS; recomputes the softmax itself.  (Four waves,
// not one: at the training batch there are only B * heads = 128 workgroups on 256 CUs, so the launch is as long as ONE
// workgroup's chain of dependent LDS loops -- 23 us with one wave; every element is the same sum in the same order.)
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(const float* __restrict__ qkv,
                                                             const float* __restrict__ dout,
                                                             float* __restrict__ dqkv, int heads, int T) {
  __shared__ float Q[HD * 32], K[HD * 32], V[HD * 32], dA[HD * 32], P[32 * 33], dS[32 * 33];
  const int tid = threadIdx.x;
  const int bh = blockIdx.x, b = bh / heads, hd = bh % heads;
  const int C = heads * HD;
  const float* base = qkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  float* dbase = dqkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  const float* dab = dout + ((long)b * C + (long)hd * HD) * T;
  for (int idx = tid; idx < HD * T; idx += 256) {
    int d = idx / T, t = idx % T;
    Q[d * 32 + t] = base[(long)(3 * d + 0) * T + t];
    K[d * 32 + t] = base[(long)(3 * d + 1) * T + t];
    V[d * 32 + t] = base[(long)(3 * d + 2) * T + t];
    dA[d * 32 + t] = dab[(long)d * T + t];
  }
  __syncthreads();
  for (int idx = tid; idx < T * T; idx += 256) {
    int qi = idx / T, ki = idx % T;
    float acc = 0.0f, dp = 0.0f;
#pragma unroll 8
    for (int d = 0; d < HD; ++d) {
      acc += Q[d * 32 + qi] * (K[d * 32 + ki] * 0.125f);
      dp += dA[d * 32 + qi] * V[d * 32 + ki];
    }
    P[qi * 33 + ki] = acc;
    dS[qi * 33 + ki] = dp;  // dP for now
  }
  __syncthreads();
  if (tid < T) {
    float m = P[tid * 33];
    for (int k = 1; k < T; ++k) m = fmaxf(m, P[tid * 33 + k]);
    float sum = 0.0f;
    for (int k = 0; k < T; ++k) {
      float e = expf(P[tid * 33 + k] - m);
      P[tid * 33 + k] = e;
      sum += e;
    }
    float inv = 1.0f / sum, del = 0.0f;
    for (int k = 0; k < T; ++k) {
      float pr = P[tid * 33 + k] * inv;
      P[tid * 33 + k] = pr;
      del += pr * dS[tid * 33 + k];
    }
    for (int k = 0; k < T; ++k) dS[tid * 33 + k] = P[tid * 33 + k] * (dS[tid * 33 + k] - del);
  }
  __syncthreads();
  // lane = channel (head-dim index), wave = every fourth key / query
  const int c = tid & 63, part = tid >> 6;
  for (int k = part; k < T; k += 4) {
    float dv = 0.0f, dk = 0.0f;
    for (int q = 0; q < T; ++q) {
      dv += dA[c * 32 + q] * P[q * 33 + k];
      dk += dS[q * 33 + k] * Q[c * 32 + q];
    }
    dbase[(long)(3 * c + 1) * T + k] = dk * 0.125f;
    dbase[(long)(3 * c + 2) * T + k] = dv;
  }
  for (int q = part; q < T; q += 4) {
    float dq = 0.0f;
    for (int k = 0; k < T; ++k) dq += dS[q * 33 + k] * K[c * 32 + k];
    dbase[(long)(3 * c + 0) * T + q] = dq * 0.125f;
  }
}

extern "C" int i2r_attention_bwd_f32(const float* qkv, const float* out, const float* lse, const float* dout,
                                     float* dqkv, float* delta_ws, int B, int heads, int T, i2r_stream_t stream) {
  I2R_REQUIRE(qkv && dout && dqkv && B > 0 && heads > 0 && T > 0, "attention_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (T <= 32) {
    hipLaunchKernelGGL(attn_bwd_small_kernel, dim3(B * heads), dim3(256), 0, s, qkv, dout, dqkv, heads, T);
    return i2r_check_launch("attention_bwd");
  }
  I2R_REQUIRE(T % 32 == 0, "attention_bwd: T=%d must be <= 32 or a multiple of 32", T);
  I2R_REQUIRE(out && lse && delta_ws, "attention_bwd: out, lse and delta workspace are required for T > 32");
  long total = (long)B * heads * T;
  hipLaunchKernelGGL(attn_delta_kernel, dim3(i2r_cdiv(total, 256)), dim3(256), 0, s, out, dout, delta_ws, heads, T,
                     total);
  if (T % 128 == 0) {
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<4>, dim3(T / 128, B * heads), dim3(256), 0, s, qkv, lse, delta_ws, dout,
                       dqkv, heads, T);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<4>, dim3(T / 128, B * heads), dim3(256), 0, s, qkv, lse, delta_ws, dout,
                       dqkv, heads, T);
  } else if (T % 64 == 0) {
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<2>, dim3(T / 64, B * heads), dim3(128), 0, s, qkv, lse, delta_ws, dout,
                       dqkv, heads, T);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<2>, dim3(T / 64, B * heads), dim3(128), 0, s, qkv, lse, delta_ws, dout,
                       dqkv, heads, T);
  } else {
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<1>, dim3(T / 32, B * heads), dim3(64), 0, s, qkv, lse, delta_ws, dout,
                       dqkv, heads, T);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<1>, dim3(T / 32, B * heads), dim3(64), 0, s, qkv, lse, delta_ws, dout,
                       dqkv, heads, T);
  }
  return i2r_check_launch("attention_bwd");
}

// ------------------------------------------------------------------------------------- backward on the f16 MFMA
// The same two kernels with every product on the half-precision matrix cores in the f16x2 split (operands hi + lo in
// fp16, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate): 48 (dk, dv) and 36 (dq) MFMAs of 32 cycles per
// 32 x 32 tile where the fp32 kernels issue 128 and 96 of 64 cycles.  Gradients are far below fp16's range, so
//   dA' = dA * s        s = the power of two that puts max|dA| into [2^13, 2^14)  (i2r_absmax_pow2_scale_f32),
//   dS' = P (dP' - delta') * 2^-10                      (|dS'| <= 2^11 max|v|: inside fp16 for |v| < 32),
// and dk, dv, dq are scaled back exactly at the end.  A value that leaves the range comes out as inf / NaN, never as
// a finite wrong number; the optimiser skips such a step and the caller goes back to the exact kernels (train.py).
// Chained products read the first product's ACCUMULATOR as their B operand (registers 8 s2 .. 8 s2 + 7 = k-step s2;
// element j of lane half h is row 16 s2 + 8 (j >> 2) + 4 h + (j & 3)), so the other operand comes from an LDS image
// stored in exactly that order ("perm" image); first products read [position][channel] rows ("rows" image).
#define BROW 144                        // bytes per position row of a rows image: 64 channels x 2 B + 16 pad
#define PROW 80                         // bytes per channel row of a perm image: 32 positions x 2 B + 16 pad
#define ROWS_B (32 * BROW)              // one of hi / lo
#define PERM_B (HD * PROW)
#define DS_SCALE 0.0009765625f          // 2^-10
#define K_SCALE 0.18033688011112042f    // 1/sqrt(64) * log2(e): scores in log2 units, softmax on exp2
#define LOG2E 1.4426950408889634f

// split of a plain fp32 value held in a register (not a product about to be folded into a conversion)
__device__ __forceinline__ void r_split8(const float* v, ah8& hi, ah8& lo) { p_split8(v, hi, lo); }

// One thread's share of staging a 64-channel x 32-position tile X[ch * cs + pos] as a rows image and a perm image.
template <int NIT>
struct TileRegs {
  float r[NIT][8];
  float4 q[NIT][2];
};
template <int NT, int NIT>
__device__ __forceinline__ void tile_load(TileRegs<NIT>& g, const float* __restrict__ x, long cs, int tid, bool rows,
                                          bool perm) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = tid + NT * it;
    if (rows) {
      const int pos = item & 31, grp = item >> 5;
#pragma unroll
      for (int j = 0; j < 8; ++j) g.r[it][j] = x[(long)(8 * grp + j) * cs + pos];
    }
    if (perm) {
      const int ch = item >> 2, piece = item & 3;
      const float* src = x + (long)ch * cs + 16 * (piece >> 1) + 4 * (piece & 1);
      g.q[it][0] = *reinterpret_cast<const float4*>(src);
      g.q[it][1] = *reinterpret_cast<const float4*>(src + 8);
    }
  }
}
template <int NT, int NIT>
__device__ __forceinline__ void tile_store(const TileRegs<NIT>& g, unsigned char* rows_img, unsigned char* perm_img,
                                           int tid, float mul) {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int item = tid + NT * it;
    ah8 hi, lo;
    if (rows_img) {
      const int pos = item & 31, grp = item >> 5;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = g.r[it][j] * mul;
      a_split8(v, hi, lo);
      *reinterpret_cast<ah8*>(rows_img + pos * BROW + grp * 16) = hi;
      *reinterpret_cast<ah8*>(rows_img + ROWS_B + pos * BROW + grp * 16) = lo;
    }
    if (perm_img) {
      const int ch = item >> 2, piece = item & 3;
      float v[8] = {g.q[it][0].x * mul, g.q[it][0].y * mul, g.q[it][0].z * mul, g.q[it][0].w * mul,
                    g.q[it][1].x * mul, g.q[it][1].y * mul, g.q[it][1].z * mul, g.q[it][1].w * mul};
      a_split8(v, hi, lo);
      *reinterpret_cast<ah8*>(perm_img + ch * PROW + piece * 16) = hi;
      *reinterpret_cast<ah8*>(perm_img + PERM_B + ch * PROW + piece * 16) = lo;
    }
  }
}
// acc += A B in the f16x2 split: A = (ah, al), B = (bh, bl)
__device__ __forceinline__ void mma3(f32x16& acc, ah8 ah, ah8 al, ah8 bh, ah8 bl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
}

#define DKV_STAGE (4 * ROWS_B + 4 * PERM_B + 256)  // Q rows, dA rows, Q perm, dA perm (hi + lo each), lse[32], delta[32]

template <int QW>
__global__ __launch_bounds__(64 * QW, QW == 4 ? 2 : 1) void attn_bwd_dkv_h2_kernel(const float* __restrict__ qkv,
                                                                  const float* __restrict__ lse,
                                                                  const float* __restrict__ delta,
                                                                  const float* __restrict__ dout,
                                                                  const float* __restrict__ scale,
                                                                  float* __restrict__ dqkv, int heads, int T) {
  constexpr int NT = 64 * QW, NIT = 256 / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char bwd_lds[];  // 2 stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int C = heads * HD;
  const float* base = qkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;  // row (d, sel) at base + (3 d + sel) T
  float* dbase = dqkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  const float* dab = dout + ((long)b * C + (long)hd * HD) * T;
  const int key0 = (blockIdx.x * QW + wave) * 32;
  const float sc = scale[0], isc = scale[1];

  // this wave's 32 keys as B operands: lane = key, element j of k-step s = channel 16 s + 8 h + j
  ah8 kh[4], kl[4], vh[4], vl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float kv[8], vv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 16 * s + 8 * h + j;
      kv[j] = base[(long)(3 * d + 1) * T + key0 + l31] * K_SCALE;
      vv[j] = base[(long)(3 * d + 2) * T + key0 + l31];
    }
    a_split8(kv, kh[s], kl[s]);
    a_split8(vv, vh[s], vl[s]);
  }
  f32x16 dK[2], dV[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dK[t][r] = dV[t][r] = 0.0f;

  TileRegs<NIT> gq, ga;
  float rl = 0.0f, rd = 0.0f;
  auto stage_load = [&](int q0) {
    tile_load<NT, NIT>(gq, base + q0, 3L * T, tid, true, true);
    tile_load<NT, NIT>(ga, dab + q0, (long)T, tid, true, true);
    if (tid < 32) {
      rl = lse[(long)bh * T + q0 + tid];
      rd = delta[(long)bh * T + q0 + tid];
    }
  };
  auto stage_store = [&](int buf) {
    unsigned char* st = bwd_lds + buf * DKV_STAGE;
    tile_store<NT, NIT>(gq, st, st + 4 * ROWS_B, tid, 1.0f);
    tile_store<NT, NIT>(ga, st + 2 * ROWS_B, st + 4 * ROWS_B + 2 * PERM_B, tid, sc);
    if (tid < 32) {
      float* ls = reinterpret_cast<float*>(st + 4 * ROWS_B + 4 * PERM_B);
      ls[tid] = rl * LOG2E;
      ls[32 + tid] = rd * sc;
    }
  };

  // Four waves: the next tile's loads are in flight under this tile's MFMAs (two LDS stages).  Two waves (T = 64, two
  // tiles in all): each thread stages twice as much, and holding that across the MFMAs would spill -- load, store,
  // compute, one stage.
  constexpr bool PIPE = QW == 4;
  stage_load(0);
  stage_store(0);
  __syncthreads();
  const int ntiles = T / 32;
  for (int qt = 0; qt < ntiles; ++qt) {
    const unsigned char* st = bwd_lds + (PIPE ? (qt & 1) : 0) * DKV_STAGE;
    if (PIPE && qt + 1 < ntiles) stage_load((qt + 1) * 32);
    const unsigned char* Qr = st;
    const unsigned char* Ar = st + 2 * ROWS_B;
    const unsigned char* Qp = st + 4 * ROWS_B;
    const unsigned char* Ap = st + 4 * ROWS_B + 2 * PERM_B;
    const float* ls = reinterpret_cast<const float*>(st + 4 * ROWS_B + 4 * PERM_B);
    // S (log2 units) and dP': rows = query, cols = this lane's key
    f32x16 sa, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = dp[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int off = l31 * BROW + s * 32 + h * 16;
      mma3(sa, *reinterpret_cast<const ah8*>(Qr + off), *reinterpret_cast<const ah8*>(Qr + ROWS_B + off), kh[s], kl[s]);
      mma3(dp, *reinterpret_cast<const ah8*>(Ar + off), *reinterpret_cast<const ah8*>(Ar + ROWS_B + off), vh[s], vl[s]);
    }
    float pv[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float pr = __builtin_amdgcn_exp2f(sa[r] - ls[q]);
      pv[r] = pr;
      ds[r] = pr * (dp[r] - ls[32 + q]) * DS_SCALE;
    }
    // dV' += dA' P,  dK' += Q dS'   (reduction over the 32 queries = the accumulator rows above)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      ah8 ph, pl, sh, sl;
      r_split8(pv + 8 * s2, ph, pl);
      r_split8(ds + 8 * s2, sh, sl);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int off = (t * 32 + l31) * PROW + (2 * s2 + h) * 16;
        mma3(dV[t], *reinterpret_cast<const ah8*>(Ap + off), *reinterpret_cast<const ah8*>(Ap + PERM_B + off), ph, pl);
        mma3(dK[t], *reinterpret_cast<const ah8*>(Qp + off), *reinterpret_cast<const ah8*>(Qp + PERM_B + off), sh, sl);
      }
    }
    if (PIPE) {
      if (qt + 1 < ntiles) stage_store((qt + 1) & 1);  // the other stage: its readers finished before the last barrier
      __syncthreads();
    } else {
      __syncthreads();  // this tile is read out
      if (qt + 1 < ntiles) {
        stage_load((qt + 1) * 32);
        stage_store(0);
        __syncthreads();
      }
    }
  }
  const float ck = 0.125f * isc * (1.0f / DS_SCALE);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      dbase[(long)(3 * d + 1) * T + key0 + l31] = dK[t][r] * ck;
      dbase[(long)(3 * d + 2) * T + key0 + l31] = dV[t][r] * isc;
    }
}

#define DQ_STAGE (4 * ROWS_B + 2 * PERM_B)  // K rows, V rows, K perm (hi + lo each)

template <int QW>
__global__ __launch_bounds__(64 * QW, QW == 4 ? 2 : 1) void attn_bwd_dq_h2_kernel(const float* __restrict__ qkv,
                                                                 const float* __restrict__ lse,
                                                                 const float* __restrict__ delta,
                                                                 const float* __restrict__ dout,
                                                                 const float* __restrict__ scale,
                                                                 float* __restrict__ dqkv, int heads, int T) {
  constexpr int NT = 64 * QW, NIT = 256 / NT;
  __shared__ __attribute__((aligned(16))) unsigned char dq_lds[2 * DQ_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int C = heads * HD;
  const float* base = qkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  float* dbase = dqkv + ((long)b * 3 * C + (long)hd * 3 * HD) * T;
  const float* dab = dout + ((long)b * C + (long)hd * HD) * T;
  const int q0 = (blockIdx.x * QW + wave) * 32;
  const float sc = scale[0], isc = scale[1];

  // this wave's 32 queries as B operands: lane = query
  ah8 qh[4], ql[4], dah[4], dal[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float qv[8], av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = 16 * s + 8 * h + j;
      qv[j] = base[(long)(3 * d + 0) * T + q0 + l31] * K_SCALE;
      av[j] = dab[(long)d * T + q0 + l31] * sc;
    }
    a_split8(qv, qh[s], ql[s]);
    a_split8(av, dah[s], dal[s]);
  }
  const float lse_q = lse[(long)bh * T + q0 + l31] * LOG2E;
  const float del_q = delta[(long)bh * T + q0 + l31] * sc;
  f32x16 dQ[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dQ[t][r] = 0.0f;

  TileRegs<NIT> gk, gv;
  auto stage_load = [&](int k0) {
    tile_load<NT, NIT>(gk, base + T + k0, 3L * T, tid, true, true);
    tile_load<NT, NIT>(gv, base + 2 * T + k0, 3L * T, tid, true, false);
  };
  auto stage_store = [&](int buf) {
    unsigned char* st = dq_lds + buf * DQ_STAGE;
    tile_store<NT, NIT>(gk, st, st + 4 * ROWS_B, tid, 1.0f);
    tile_store<NT, NIT>(gv, st + 2 * ROWS_B, nullptr, tid, 1.0f);
  };

  constexpr bool PIPE = QW == 4;  // as in the dk / dv kernel
  stage_load(0);
  stage_store(0);
  __syncthreads();
  const int ntiles = T / 32;
  for (int kt = 0; kt < ntiles; ++kt) {
    const unsigned char* st = dq_lds + (PIPE ? (kt & 1) : 0) * DQ_STAGE;
    if (PIPE && kt + 1 < ntiles) stage_load((kt + 1) * 32);
    const unsigned char* Kr = st;
    const unsigned char* Vr = st + 2 * ROWS_B;
    const unsigned char* Kp = st + 4 * ROWS_B;
    // S^T (log2 units) and dP'^T: rows = key, cols = this lane's query
    f32x16 sa, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = dp[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int off = l31 * BROW + s * 32 + h * 16;
      mma3(sa, *reinterpret_cast<const ah8*>(Kr + off), *reinterpret_cast<const ah8*>(Kr + ROWS_B + off), qh[s], ql[s]);
      mma3(dp, *reinterpret_cast<const ah8*>(Vr + off), *reinterpret_cast<const ah8*>(Vr + ROWS_B + off), dah[s], dal[s]);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(sa[r] - lse_q) * (dp[r] - del_q) * DS_SCALE;
    // dQ' += K dS'^T  (reduction over the 32 keys = the accumulator rows above)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      ah8 sh, sl;
      r_split8(ds + 8 * s2, sh, sl);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int off = (t * 32 + l31) * PROW + (2 * s2 + h) * 16;
        mma3(dQ[t], *reinterpret_cast<const ah8*>(Kp + off), *reinterpret_cast<const ah8*>(Kp + PERM_B + off), sh, sl);
      }
    }
    if (PIPE) {
      if (kt + 1 < ntiles) stage_store((kt + 1) & 1);
      __syncthreads();
    } else {
      __syncthreads();
      if (kt + 1 < ntiles) {
        stage_load((kt + 1) * 32);
        stage_store(0);
        __syncthreads();
      }
    }
  }
  const float cq = 0.125f * isc * (1.0f / DS_SCALE);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      dbase[(long)(3 * d + 0) * T + q0 + l31] = dQ[t][r] * cq;
    }
}

// The f16x2 form of i2r_attention_bwd_f32 for T a multiple of 64 (four waves = 128 keys / queries per workgroup when
// T % 128 == 0, else two).
// scale: device float[2] {s, 1/s} from i2r_absmax_pow2_scale_f32 over dout.  All pointers 16-byte aligned.
extern "C" int i2r_attention_bwd_h2(const float* qkv, const float* out, const float* lse, const float* dout,
                                    float* dqkv, float* delta_ws, const float* scale, int B, int heads, int T,
                                    i2r_stream_t stream) {
  I2R_REQUIRE(qkv && out && lse && dout && dqkv && delta_ws && scale && B > 0 && heads > 0,
              "attention_bwd_h2: bad argument");
  I2R_REQUIRE(T >= 64 && T % 64 == 0, "attention_bwd_h2: T=%d must be a multiple of 64", T);
  I2R_REQUIRE(((((uintptr_t)qkv) | ((uintptr_t)dout)) & 15) == 0, "attention_bwd_h2: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  long total = (long)B * heads * T;
  hipLaunchKernelGGL(attn_delta_kernel, dim3(i2r_cdiv(total, 256)), dim3(256), 0, s, out, dout, delta_ws, heads, T,
                     total);
#define BWD_GO(QW_)                                                                                                 \
  do {                                                                                                              \
    auto kern = attn_bwd_dkv_h2_kernel<QW_>;                                                                        \
    static bool attr[64] = {false}; \
    if (!i2r_lds_attr((const void*)kern, 2 * DKV_STAGE, attr)) return I2R_ERR_HIP; \
    hipLaunchKernelGGL(kern, dim3(T / (32 * QW_), B * heads), dim3(64 * QW_), 2 * DKV_STAGE, s, qkv, lse, delta_ws, \
                       dout, scale, dqkv, heads, T);                                                                \
    hipLaunchKernelGGL(attn_bwd_dq_h2_kernel<QW_>, dim3(T / (32 * QW_), B * heads), dim3(64 * QW_), 0, s, qkv, lse, \
                       delta_ws, dout, scale, dqkv, heads, T);                                                      \
  } while (0)
  if (T % 128 == 0)
    BWD_GO(4);
  else
    BWD_GO(2);
#undef BWD_GO
  return i2r_check_launch("attention_bwd_h2");
}
