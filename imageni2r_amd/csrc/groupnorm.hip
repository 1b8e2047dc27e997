// Fused GroupNorm family (models/networks.py:104-106 as used at :168, :173, :182, :460):
//   y = dropout( act( gn(x) * (1 + scale) + shift ) )      act in {identity, SiLU}; scale/shift optional (adaptive
//   modulation from the block's `affine` Linear, :170-173); dropout only in training (:177).
// A group is (C/G)*H*W contiguous fp32 in NCHW, at most 16 KiB in every shipped config.  One wavefront (or a 16-lane
// quarter of one for groups of <= 64 elements) owns one (sample, group) and keeps the whole group in REGISTERS: one
// 16-B-per-lane read of x (and of dy in backward), shuffle reductions, one 16-B-per-lane write.  HBM-bound:
// 8 B/element forward, 12 B/element backward (+ the small per-channel outputs).
// Shapes the register kernels do not cover (H*W not a power of two >= 4, groups > 4096 elements in backward,
// > 16384 in forward, where a whole workgroup can own a group) take the generic
// strided-loop kernels at the bottom.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

#define GN_WAVES 4

// Dropout keep-mask: one Philox4x32 call covers 4 consecutive elements (counter = element index / 4).
__device__ __forceinline__ uint4 keep_bits(uint64_t seed, uint64_t offset, long i4) {
  return philox4x32(make_uint4((uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)),
                    make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}
__device__ __forceinline__ float keep_of(uint32_t r, float p, float inv_keep) {
  return (u32_to_unit(r) > p) ? inv_keep : 0.0f;
}
__device__ __forceinline__ uint64_t eff_offset(uint64_t offset, const uint64_t* rng_step) {
  return rng_step ? offset + (rng_step[0] << 20) : offset;
}

template <int LPG>
__device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
  for (int o = LPG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the LPG lanes that own a group; LPG = 256 is the whole workgroup (groups of 4097..16384 elements: the
// 32x32 levels of the ETTh1-768 / weather configs), reduced through LDS (SLOT keeps the two uses on separate words)
template <int LPG, int SLOT>
__device__ __forceinline__ float grp_sum(float v) {
  if constexpr (LPG <= 64) {
    return seg_sum<LPG>(v);
  } else {
    __shared__ float red[4];
    v = seg_sum<64>(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  }
}
// sum over all 256 threads of the workgroup (every thread must call it); `phase` alternates the LDS words so that a
// wave already in the next call cannot overwrite what a slower wave is still reading
__device__ __forceinline__ float wg256_sum(float v, int& phase) {
  __shared__ float red[2][4];
  v = seg_sum<64>(v);
  if ((threadIdx.x & 63) == 0) red[phase][threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = (red[phase][0] + red[phase][1]) + (red[phase][2] + red[phase][3]);
  phase ^= 1;
  return r;
}
// sum over aligned runs of `w` lanes (w a power of two <= 64, wave-uniform)
__device__ __forceinline__ float run_sum(float v, int w) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    if (o < w) v += __shfl_xor(v, o, 64);
  return v;
}

struct GnP {
  const float* x;       // input (B, C, HW) -- or the first C1 channels of a virtual channel concat [x | x2]
  const float* x2;      // forward only: the remaining C - C1 channels (B, C - C1, HW), or null
  const float* gamma;
  const float* beta;
  const float* ss;      // [scale | shift] (ss_batch, 2C) or null
  const float* mean_in; // backward
  const float* rstd_in;
  const float* dy;
  float* y;             // forward output / backward dx
  float* mean_out;
  float* rstd_out;
  float* dgb_part;      // backward: (B, 2C) per-sample [dgamma | dbeta]  (or null when dg_acc / db_acc are given)
  float* dg_acc;        // backward: (C) dgamma / dbeta accumulated with fp32 atomics over the batch (pre-zeroed by
  float* db_acc;        //           the caller: the step's gradient slots) -- saves the (B, 2C) partials + a reduction
  float* dss;           // backward: (B, 2C) [dscale | dshift]
  const float* add;     // backward, optional (B, C, HW): added to dx (the gradient of the block's bypass of x)
  const uint64_t* rng_step;
  int ss_batch, B, C, C1, HW, G, sh, act, split_out, gb_lds;
  float eps, drop_p;
  uint64_t seed, offset;
};

// LPG lanes own one group; each lane holds NV float4 (elements (sl + LPG*k)*4 .. +3).  Requires HW = 2^sh >= 4.
template <int LPG, int NV>
__global__ __launch_bounds__(256) void gn_fwd_reg_kernel(GnP p) {
  const int sl = threadIdx.x % LPG;
  const long wg = (long)blockIdx.x * (256 / LPG) + threadIdx.x / LPG;
  const bool valid = wg < (long)p.B * p.G;
  const int b = valid ? (int)(wg / p.G) : 0, g = valid ? (int)(wg % p.G) : 0;
  const int cpg = p.C / p.G, n = cpg * p.HW;
  const long base = ((long)b * p.C + (long)g * cpg) * p.HW;
  const float4* xp = reinterpret_cast<const float4*>(p.x + base);

  float4 v[NV];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int i4 = sl + LPG * k;
    const float4* src = xp + i4;
    if (p.x2) {  // torch.cat([x, x2], 1) read in place: a float4 never straddles a channel (HW >= 4)
      int c = g * cpg + ((i4 * 4) >> p.sh), hw = (i4 * 4) & (p.HW - 1);
      const float* q = c < p.C1 ? p.x + ((long)b * p.C1 + c) * p.HW + hw
                                : p.x2 + ((long)b * (p.C - p.C1) + (c - p.C1)) * p.HW + hw;
      src = reinterpret_cast<const float4*>(q);
    }
    v[k] = (valid && i4 * 4 < n) ? *src : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  const float mean = grp_sum<LPG, 0>(s) / (float)n;
  float q = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int i4 = sl + LPG * k;
    if (i4 * 4 < n) {
      float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  }
  const float rstd = 1.0f / sqrtf(grp_sum<LPG, 1>(q) / (float)n + p.eps);
  if (!valid) return;
  if (sl == 0 && p.mean_out) {
    p.mean_out[wg] = mean;
    p.rstd_out[wg] = rstd;
  }
  const float* ssp = p.ss ? p.ss + (long)(p.ss_batch == 1 ? 0 : b) * 2 * p.C : nullptr;
  const uint64_t off = eff_offset(p.offset, p.rng_step);
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  float4* yp = reinterpret_cast<float4*>(p.y + base);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int i4 = sl + LPG * k;
    if (i4 * 4 >= n) continue;
    int c = g * cpg + ((i4 * 4) >> p.sh);
    float gm = p.gamma[c], bt = p.beta[c];
    float u[4] = {(v[k].x - mean) * rstd * gm + bt, (v[k].y - mean) * rstd * gm + bt,
                  (v[k].z - mean) * rstd * gm + bt, (v[k].w - mean) * rstd * gm + bt};
    if (ssp) {
      float sc = ssp[c] + 1.0f, shf = ssp[p.C + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = u[j] * sc + shf;
    }
    if (p.act) {
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = silu_fast(u[j]);
    }
    if (p.drop_p > 0.0f) {
      uint4 r = keep_bits(p.seed, off, base / 4 + i4);
      u[0] *= keep_of(r.x, p.drop_p, inv_keep);
      u[1] *= keep_of(r.y, p.drop_p, inv_keep);
      u[2] *= keep_of(r.z, p.drop_p, inv_keep);
      u[3] *= keep_of(r.w, p.drop_p, inv_keep);
    }
    if (p.split_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = split_pack(u[j]);
    }
    yp[i4] = make_float4(u[0], u[1], u[2], u[3]);
  }
}

// Forward for SMALL groups (n <= 4 * LPG: one float4 per lane) when a launch has many of them (the 4x4 / 2x2 levels at
// inference batches): every LPG-lane slot walks GPS groups with all GPS loads issued before the first reduction -- the
// one-group form keeps 16 bytes per lane in flight and streamed these levels at 2.8-3.3 TB/s.
template <int LPG, int GPS>
__global__ __launch_bounds__(256) void gn_fwd_small_kernel(GnP p) {
  const int sl = threadIdx.x % LPG;
  const long slot = (long)blockIdx.x * (256 / LPG) + threadIdx.x / LPG;
  const int cpg = p.C / p.G, n = cpg * p.HW;
  const long ngroups = (long)p.B * p.G;
  const bool lane_in = sl * 4 < n;
  const int cl = (sl * 4) >> p.sh, hw = (sl * 4) & (p.HW - 1);  // this lane's channel within the group, pixel
  float4 v[GPS];
  bool valid[GPS];
  int bb[GPS], gg[GPS];
#pragma unroll
  for (int gi = 0; gi < GPS; ++gi) {
    const long wg = slot * GPS + gi;
    valid[gi] = wg < ngroups;
    bb[gi] = valid[gi] ? (int)(wg / p.G) : 0;
    gg[gi] = valid[gi] ? (int)(wg % p.G) : 0;
    const int c = gg[gi] * cpg + cl;
    const float* q;
    if (p.x2)
      q = c < p.C1 ? p.x + ((long)bb[gi] * p.C1 + c) * p.HW + hw
                   : p.x2 + ((long)bb[gi] * (p.C - p.C1) + (c - p.C1)) * p.HW + hw;
    else
      q = p.x + ((long)bb[gi] * p.C + c) * p.HW + hw;
    v[gi] = (valid[gi] && lane_in) ? *reinterpret_cast<const float4*>(q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const uint64_t off = eff_offset(p.offset, p.rng_step);
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
#pragma unroll
  for (int gi = 0; gi < GPS; ++gi) {
    const float s = (v[gi].x + v[gi].y) + (v[gi].z + v[gi].w);
    const float mean = seg_sum<LPG>(s) / (float)n;
    float q = 0.0f;
    if (lane_in) {
      float d0 = v[gi].x - mean, d1 = v[gi].y - mean, d2 = v[gi].z - mean, d3 = v[gi].w - mean;
      q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    const float rstd = 1.0f / sqrtf(seg_sum<LPG>(q) / (float)n + p.eps);
    if (!valid[gi]) continue;
    const long wg = slot * GPS + gi;
    if (sl == 0 && p.mean_out) {
      p.mean_out[wg] = mean;
      p.rstd_out[wg] = rstd;
    }
    if (!lane_in) continue;
    const int c = gg[gi] * cpg + cl;
    const float gm = p.gamma[c], bt = p.beta[c];
    float u[4] = {(v[gi].x - mean) * rstd * gm + bt, (v[gi].y - mean) * rstd * gm + bt,
                  (v[gi].z - mean) * rstd * gm + bt, (v[gi].w - mean) * rstd * gm + bt};
    if (p.ss) {
      const float* ssp = p.ss + (long)(p.ss_batch == 1 ? 0 : bb[gi]) * 2 * p.C;
      float sc = ssp[c] + 1.0f, shf = ssp[p.C + c];
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = u[j] * sc + shf;
    }
    if (p.act) {
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = silu_fast(u[j]);
    }
    const long base = ((long)bb[gi] * p.C + (long)gg[gi] * cpg) * p.HW;
    if (p.drop_p > 0.0f) {
      uint4 r = keep_bits(p.seed, off, base / 4 + sl);
      u[0] *= keep_of(r.x, p.drop_p, inv_keep);
      u[1] *= keep_of(r.y, p.drop_p, inv_keep);
      u[2] *= keep_of(r.z, p.drop_p, inv_keep);
      u[3] *= keep_of(r.w, p.drop_p, inv_keep);
    }
    if (p.split_out) {
#pragma unroll
      for (int j = 0; j < 4; ++j) u[j] = split_pack(u[j]);
    }
    reinterpret_cast<float4*>(p.y + base)[sl] = make_float4(u[0], u[1], u[2], u[3]);
  }
}

// Backward, same ownership.  Per-channel sums (dgamma/dbeta partials, dscale/dshift) are reductions over runs of
// min(LPG, HW/4) lanes, accumulated over the HW/(4*LPG) register slots a channel spans when HW > 4*LPG.
template <int LPG, int NV>
__global__ __launch_bounds__(256) void gn_bwd_reg_kernel(GnP p) {
  const int sl = threadIdx.x % LPG;
  // The 256 / LPG group slots of a workgroup own the SAME group of consecutive samples (not consecutive groups of one
  // sample): their per-channel dgamma / dbeta sums then meet in LDS and leave as ONE global atomic per channel and
  // workgroup.  All B * C atomics of a launch land on C / 32 cache lines, where they are served one after another:
  // one per sample and channel, they cost 0.26 ms of the 13 ms B1 train step (tools/r3_gnabl_ab.sh).
  constexpr int SLOTS = 256 / LPG;
  const int nbt = (p.B + SLOTS - 1) / SLOTS;   // grid = G * nbt
  const int g_blk = blockIdx.x / nbt, b_blk = (blockIdx.x - g_blk * nbt) * SLOTS + threadIdx.x / LPG;
  const bool valid = b_blk < p.B;
  const int b = valid ? b_blk : 0, g = g_blk;
  const long wg = (long)b * p.G + g;
  const int cpg = p.C / p.G, n = cpg * p.HW;
  __shared__ float gb_lds[2][64];
  // The whole-workgroup forms (SLOTS = 1) go through LDS too: the cpg channel sums then leave in ONE atomic
  // instruction of cpg adjacent lanes instead of one per channel from its own wave -- what queues on a line are
  // requests, not lanes (32 x 128 x 16 x 16: 19.3 -> see tools/r3_gn_bwd_prof.sh).
  const bool gb_in_lds = p.gb_lds && p.dg_acc && cpg <= 64;   // (uniform over the workgroup)
  if (gb_in_lds) {
    if (threadIdx.x < 128) (&gb_lds[0][0])[threadIdx.x] = 0.0f;
    __syncthreads();
  }
  const long base = ((long)b * p.C + (long)g * cpg) * p.HW;
  const float4* xp = reinterpret_cast<const float4*>(p.x + base);
  const float4* dyp = reinterpret_cast<const float4*>(p.dy + base);
  const float mean = valid ? p.mean_in[wg] : 0.0f, rstd = valid ? p.rstd_in[wg] : 0.0f;
  const float* ssp = p.ss ? p.ss + (long)(p.ss_batch == 1 ? 0 : b) * 2 * p.C : nullptr;
  const uint64_t off = eff_offset(p.offset, p.rng_step);
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const int q4 = p.HW >> 2;                    // float4 per channel
  const int runw = q4 < LPG ? q4 : LPG;        // lanes of one slot that share a channel
  const int spc = q4 <= LPG ? 1 : q4 / LPG;    // slots per channel

  // LPG = 256 (groups of 4097..16384 elements: the concatenated inputs of the 32x32 decoder blocks): the whole
  // workgroup owns one group; a channel's lanes are a run inside one wave (HW <= 256) or the whole workgroup
  int phase = 0;
  auto chan_sum = [&](float v) -> float {
    if constexpr (LPG <= 64) return run_sum(v, runw);
    else return runw <= 64 ? run_sum(v, runw) : wg256_sum(v, phase);
  };
  auto group_sum = [&](float v) -> float {
    if constexpr (LPG <= 64) return seg_sum<LPG>(v);
    else return wg256_sum(v, phase);
  };
  float4 xh[NV], dn[NV];  // xhat and d(loss)/d(normalised*gamma+beta) after modulation  (dn = du * (1+scale))
  float S1 = 0.0f, S2 = 0.0f;
  float a_du = 0.0f, a_dun = 0.0f, a_dn = 0.0f, a_dnx = 0.0f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int i4 = sl + LPG * k;
    bool in = valid && i4 * 4 < n;
    int c = g * cpg + ((i4 * 4) >> p.sh);
    float gm = 0.0f, bt = 0.0f, sc1 = 1.0f, shf = 0.0f;
    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), dv = xv;
    if (in) {
      xv = xp[i4];
      dv = dyp[i4];
      gm = p.gamma[c];
      bt = p.beta[c];
      if (ssp) {
        sc1 = ssp[c] + 1.0f;
        shf = ssp[p.C + c];
      }
    }
    float xs[4] = {xv.x, xv.y, xv.z, xv.w}, du[4] = {dv.x, dv.y, dv.z, dv.w};
    if (in && p.drop_p > 0.0f) {
      uint4 r = keep_bits(p.seed, off, base / 4 + i4);
      du[0] *= keep_of(r.x, p.drop_p, inv_keep);
      du[1] *= keep_of(r.y, p.drop_p, inv_keep);
      du[2] *= keep_of(r.z, p.drop_p, inv_keep);
      du[3] *= keep_of(r.w, p.drop_p, inv_keep);
    }
    float xo[4], dno[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float h = (xs[j] - mean) * rstd;
      float nn = h * gm + bt;
      float d = du[j];
      if (p.act) d *= dsilu_f(nn * sc1 + shf);
      float dnj = d * sc1;
      if (!in) { h = 0.0f; d = 0.0f; dnj = 0.0f; nn = 0.0f; }
      xo[j] = h;
      dno[j] = dnj;
      a_du += d;
      a_dun += d * nn;
      a_dn += dnj;
      a_dnx += dnj * h;
    }
    xh[k] = make_float4(xo[0], xo[1], xo[2], xo[3]);
    dn[k] = make_float4(dno[0] * gm, dno[1] * gm, dno[2] * gm, dno[3] * gm);  // now d/dxhat
    if ((k + 1) % spc == 0 || k == NV - 1) {  // channel complete in this lane's slots (spc, NV powers of two)
      float r_dn = chan_sum(a_dn), r_dnx = chan_sum(a_dnx);
      S1 += gm * a_dn;
      S2 += gm * a_dnx;
      if (ssp) {
        float r_du = chan_sum(a_du), r_dun = chan_sum(a_dun);
        if (in && (sl & (runw - 1)) == 0) {
          p.dss[(long)b * 2 * p.C + c] = r_dun;
          p.dss[(long)b * 2 * p.C + p.C + c] = r_du;
        }
      }
      if (in && (sl & (runw - 1)) == 0) {
        if (gb_in_lds) {
          atomicAdd(&gb_lds[0][c - g * cpg], r_dnx);
          atomicAdd(&gb_lds[1][c - g * cpg], r_dn);
        } else if (p.dg_acc) {
          atomicAdd(p.dg_acc + c, r_dnx);
          atomicAdd(p.db_acc + c, r_dn);
        } else {
          p.dgb_part[(long)b * 2 * p.C + c] = r_dnx;
          p.dgb_part[(long)b * 2 * p.C + p.C + c] = r_dn;
        }
      }
      a_du = a_dun = a_dn = a_dnx = 0.0f;
    }
  }
  const float inv_n = 1.0f / (float)n;
  const float m1 = group_sum(S1) * inv_n, m2 = group_sum(S2) * inv_n;
  if (gb_in_lds) {
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
      atomicAdd(p.dg_acc + g * cpg + threadIdx.x, gb_lds[0][threadIdx.x]);
      atomicAdd(p.db_acc + g * cpg + threadIdx.x, gb_lds[1][threadIdx.x]);
    }
  }
  if (!valid) return;
  float4* dxp = reinterpret_cast<float4*>(p.y + base);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    int i4 = sl + LPG * k;
    if (i4 * 4 >= n) continue;
    float4 o = make_float4(rstd * (dn[k].x - m1 - xh[k].x * m2), rstd * (dn[k].y - m1 - xh[k].y * m2),
                           rstd * (dn[k].z - m1 - xh[k].z * m2), rstd * (dn[k].w - m1 - xh[k].w * m2));
    if (p.add) {
      const float4 a = reinterpret_cast<const float4*>(p.add + base)[i4];
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    dxp[i4] = o;
  }
}

// ---------------------------------------------------------------------------------------------- generic fallbacks
__device__ __forceinline__ float keep_scale1(uint64_t seed, uint64_t offset, long i, float p) {
  uint4 r = keep_bits(seed, offset, i >> 2);
  uint32_t w = (i & 3) == 0 ? r.x : (i & 3) == 1 ? r.y : (i & 3) == 2 ? r.z : r.w;
  return keep_of(w, p, 1.0f / (1.0f - p));
}

__global__ __launch_bounds__(64 * GN_WAVES) void groupnorm_fwd_kernel(GnP p) {
  int lane = threadIdx.x & 63;
  long wg = (long)blockIdx.x * GN_WAVES + (threadIdx.x >> 6);  // (b, g) index, wave-uniform
  if (wg >= (long)p.B * p.G) return;
  int b = (int)(wg / p.G), g = (int)(wg % p.G);
  int cpg = p.C / p.G, HW = p.HW, C = p.C;
  int n = cpg * HW;
  long base = ((long)b * C + (long)g * cpg) * HW;
  const float* xp = p.x + base;
  const uint64_t off = eff_offset(p.offset, p.rng_step);

  float s = 0.0f;
  for (int i = lane; i < n; i += 64) s += xp[i];
  float mean = wave_sum(s) / (float)n;
  float v = 0.0f;
  for (int i = lane; i < n; i += 64) {
    float d = xp[i] - mean;
    v += d * d;
  }
  float rstd = 1.0f / sqrtf(wave_sum(v) / (float)n + p.eps);
  if (lane == 0 && p.mean_out) {
    p.mean_out[wg] = mean;
    p.rstd_out[wg] = rstd;
  }
  const float* ssp = p.ss ? p.ss + (long)(p.ss_batch == 1 ? 0 : b) * 2 * C : nullptr;
  for (int i = lane; i < n; i += 64) {
    int c = g * cpg + i / HW;
    float u = (xp[i] - mean) * rstd * p.gamma[c] + p.beta[c];
    if (ssp) u = u * (ssp[c] + 1.0f) + ssp[C + c];
    if (p.act) u = silu_f(u);
    if (p.drop_p > 0.0f) u *= keep_scale1(p.seed, off, base + i, p.drop_p);
    p.y[base + i] = p.split_out ? split_pack(u) : u;
  }
}

// Pass A: per-channel sums over HW of dn and dn*xhat (the per-sample dbeta/dgamma partials; combined with gamma they
// give the two group moments that dx needs); pass B: dx.
__global__ __launch_bounds__(64 * GN_WAVES) void groupnorm_bwd_kernel(GnP p) {
  int lane = threadIdx.x & 63;
  long wg = (long)blockIdx.x * GN_WAVES + (threadIdx.x >> 6);
  if (wg >= (long)p.B * p.G) return;
  int b = (int)(wg / p.G), g = (int)(wg % p.G);
  int cpg = p.C / p.G, HW = p.HW, C = p.C;
  int n = cpg * HW;
  long base = ((long)b * C + (long)g * cpg) * HW;
  const float* xp = p.x + base;
  const float* dyp = p.dy + base;
  float mean = p.mean_in[wg], rstd = p.rstd_in[wg];
  const float* ssp = p.ss ? p.ss + (long)(p.ss_batch == 1 ? 0 : b) * 2 * C : nullptr;
  const uint64_t off = eff_offset(p.offset, p.rng_step);

  float S1 = 0.0f, S2 = 0.0f;  // sum_c gamma_c * sum_hw dn,  sum_c gamma_c * sum_hw dn*xhat
  for (int cc = 0; cc < cpg; ++cc) {
    int c = g * cpg + cc;
    float gm = p.gamma[c], bt = p.beta[c];
    float sc1 = ssp ? ssp[c] + 1.0f : 1.0f, sh = ssp ? ssp[C + c] : 0.0f;
    float a_du = 0.0f, a_dun = 0.0f, a_dn = 0.0f, a_dnx = 0.0f;
    for (int q = lane; q < HW; q += 64) {
      int i = cc * HW + q;
      float xh = (xp[i] - mean) * rstd;
      float nn = xh * gm + bt;
      float u = nn * sc1 + sh;
      float du = dyp[i];
      if (p.drop_p > 0.0f) du *= keep_scale1(p.seed, off, base + i, p.drop_p);
      if (p.act) du *= dsilu_f(u);
      float dn = du * sc1;
      a_du += du;
      a_dun += du * nn;
      a_dn += dn;
      a_dnx += dn * xh;
    }
    a_du = wave_sum(a_du);
    a_dun = wave_sum(a_dun);
    a_dn = wave_sum(a_dn);
    a_dnx = wave_sum(a_dnx);
    if (lane == 0) {
      if (p.dg_acc) {
        atomicAdd(p.dg_acc + c, a_dnx);
        atomicAdd(p.db_acc + c, a_dn);
      } else {
        p.dgb_part[(long)b * 2 * C + c] = a_dnx;
        p.dgb_part[(long)b * 2 * C + C + c] = a_dn;
      }
      if (p.dss) {
        p.dss[(long)b * 2 * C + c] = a_dun;      // d scale
        p.dss[(long)b * 2 * C + C + c] = a_du;   // d shift
      }
    }
    S1 += gm * a_dn;
    S2 += gm * a_dnx;
  }
  float inv_n = 1.0f / (float)n;
  float m1 = S1 * inv_n, m2 = S2 * inv_n;
  for (int i = lane; i < n; i += 64) {
    int c = g * cpg + i / HW;
    float gm = p.gamma[c];
    float sc1 = ssp ? ssp[c] + 1.0f : 1.0f;
    float xh = (xp[i] - mean) * rstd;
    float du = dyp[i];
    if (p.drop_p > 0.0f) du *= keep_scale1(p.seed, off, base + i, p.drop_p);
    if (p.act) {
      float u = (xh * gm + p.beta[c]) * sc1 + (ssp ? ssp[C + c] : 0.0f);
      du *= dsilu_f(u);
    }
    float dxh = du * sc1 * gm;
    p.y[base + i] = rstd * (dxh - m1 - xh * m2) + (p.add ? p.add[base + i] : 0.0f);
  }
}

// Forward for 1x1 images (HW = 1; the register forms need HW >= 4): one lane per channel, LPG lanes per (sample, group).
// The generic kernel gave each 8..64-element group a whole wave and three passes: 52 us per launch at 4096 samples x 512
// channels (16 MB: 4 us of HBM), 5 % of a shape-A evaluation.
template <int LPG>
__global__ __launch_bounds__(256) void gn_fwd_hw1_kernel(GnP p) {
  const int sl = threadIdx.x % LPG;
  const long wg = (long)blockIdx.x * (256 / LPG) + threadIdx.x / LPG;   // (b, g) index
  const bool valid = wg < (long)p.B * p.G;
  const int b = valid ? (int)(wg / p.G) : 0, g = valid ? (int)(wg % p.G) : 0;
  const int cpg = p.C / p.G, C = p.C;
  const bool in = valid && sl < cpg;
  const int c = g * cpg + (sl < cpg ? sl : 0);
  const long idx = (long)b * C + c;
  const float xv = in ? p.x[idx] : 0.0f;
  const float mean = seg_sum<LPG>(xv) / (float)cpg;
  const float d = in ? xv - mean : 0.0f;
  const float rstd = 1.0f / sqrtf(seg_sum<LPG>(d * d) / (float)cpg + p.eps);
  if (!in) return;
  if (sl == 0 && p.mean_out) {
    p.mean_out[wg] = mean;
    p.rstd_out[wg] = rstd;
  }
  float u = d * rstd * p.gamma[c] + p.beta[c];
  if (p.ss) {
    const float* ssp = p.ss + (long)(p.ss_batch == 1 ? 0 : b) * 2 * C;
    u = u * (ssp[c] + 1.0f) + ssp[C + c];
  }
  if (p.act) u = silu_f(u);
  if (p.drop_p > 0.0f) u *= keep_scale1(p.seed, eff_offset(p.offset, p.rng_step), idx, p.drop_p);
  p.y[idx] = p.split_out ? split_pack(u) : u;
}

// Backward for 1x1 images (HW = 1: the bottom level of the 8x8-image configs; the register forms need HW >= 4).  One
// lane per channel, LPG lanes per (sample, group), the 256 / LPG slots of a workgroup on the same group of consecutive
// samples; dgamma / dbeta as in gn_bwd_reg_kernel (LDS, then one atomic instruction per workgroup and parameter).  The
// generic kernel spent a wave and four wave reductions per CHANNEL here and one atomic request per sample and channel:
// 56 us per launch at 128 samples x 512 channels, 9 % of shape A's train step.
template <int LPG>
__global__ __launch_bounds__(256) void gn_bwd_hw1_kernel(GnP p) {
  constexpr int SLOTS = 256 / LPG;
  const int sl = threadIdx.x % LPG;
  const int nbt = (p.B + SLOTS - 1) / SLOTS;   // grid = G * nbt
  const int g = blockIdx.x / nbt, b_blk = (blockIdx.x - g * nbt) * SLOTS + threadIdx.x / LPG;
  const int cpg = p.C / p.G, C = p.C;
  const bool valid = b_blk < p.B;
  const int b = valid ? b_blk : 0;
  const bool in = valid && sl < cpg;
  const int c = g * cpg + (sl < cpg ? sl : 0);
  const long idx = (long)b * C + c;
  __shared__ float gb_lds[2][64];
  const bool gb_in_lds = p.dg_acc != nullptr;   // (uniform)
  if (gb_in_lds) {
    if (threadIdx.x < 128) (&gb_lds[0][0])[threadIdx.x] = 0.0f;
    __syncthreads();
  }
  const float mean = p.mean_in[(long)b * p.G + g], rstd = p.rstd_in[(long)b * p.G + g];
  const float* ssp = p.ss ? p.ss + (long)(p.ss_batch == 1 ? 0 : b) * 2 * C : nullptr;
  float xh = 0.0f, dxh = 0.0f, dn = 0.0f, gm = 0.0f;
  if (in) {
    gm = p.gamma[c];
    const float bt = p.beta[c];
    const float sc1 = ssp ? ssp[c] + 1.0f : 1.0f, shf = ssp ? ssp[C + c] : 0.0f;
    xh = (p.x[idx] - mean) * rstd;
    const float nn = xh * gm + bt;
    float du = p.dy[idx];
    if (p.drop_p > 0.0f) du *= keep_scale1(p.seed, eff_offset(p.offset, p.rng_step), idx, p.drop_p);
    if (p.act) du *= dsilu_f(nn * sc1 + shf);
    dn = du * sc1;
    dxh = dn * gm;
    if (p.dss) {
      p.dss[(long)b * 2 * C + c] = du * nn;   // d scale
      p.dss[(long)b * 2 * C + C + c] = du;    // d shift
    }
    if (gb_in_lds) {
      atomicAdd(&gb_lds[0][sl], dn * xh);
      atomicAdd(&gb_lds[1][sl], dn);
    } else {
      p.dgb_part[(long)b * 2 * C + c] = dn * xh;
      p.dgb_part[(long)b * 2 * C + C + c] = dn;
    }
  }
  const float inv_n = 1.0f / (float)cpg;
  const float m1 = seg_sum<LPG>(dxh) * inv_n, m2 = seg_sum<LPG>(dxh * xh) * inv_n;
  if (in) p.y[idx] = rstd * (dxh - m1 - xh * m2) + (p.add ? p.add[idx] : 0.0f);
  if (gb_in_lds) {
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
      atomicAdd(p.dg_acc + g * cpg + threadIdx.x, gb_lds[0][threadIdx.x]);
      atomicAdd(p.db_acc + g * cpg + threadIdx.x, gb_lds[1][threadIdx.x]);
    }
  }
}

// out[c] = sum_r in[r][c]   (deterministic batch reduction of the per-sample dgamma/dbeta partials, bias grads ...).
// One lane per column would serialise `rows` dependent load+add pairs (latency-bound at the 1-8 blocks these launch),
// so 4 lanes share a column, each summing every 4th row with 4 independent loads in flight; the 4 partials are
// combined in a fixed order by shuffles (deterministic).
__device__ __forceinline__ float col_sum4(const float* __restrict__ in, int rows, int cols, int c, int sub) {
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  int r = sub;
  for (; r + 12 < rows; r += 16) {
    float v0 = in[(long)r * cols + c], v1 = in[(long)(r + 4) * cols + c];
    float v2 = in[(long)(r + 8) * cols + c], v3 = in[(long)(r + 12) * cols + c];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; r < rows; r += 4) a0 += in[(long)r * cols + c];
  float acc = (a0 + a1) + (a2 + a3);
  acc += __shfl_xor(acc, 16, 64);  // the 4 sub-lanes of a column sit 16 lanes apart (16 columns per wave)
  acc += __shfl_xor(acc, 32, 64);
  return acc;
}
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          int rows, int cols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 4 + wave) * 16 + (lane & 15), sub = lane >> 4;
  const bool ok = c < cols;
  float acc = col_sum4(in, ok ? rows : 0, cols, ok ? c : 0, sub);
  if (ok && sub == 0) out[c] = acc;
}

// same, columns [0, cols_a) -> out_a and [cols_a, cols_a + cols_b) -> out_b (dgamma | dbeta into their own buffers)
__global__ __launch_bounds__(256) void reduce_rows2_kernel(const float* __restrict__ in, float* __restrict__ out_a,
                                                           float* __restrict__ out_b, int rows, int cols_a,
                                                           int cols_b) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cols = cols_a + cols_b;
  const int c = (blockIdx.x * 4 + wave) * 16 + (lane & 15), sub = lane >> 4;
  const bool ok = c < cols;
  float acc = col_sum4(in, ok ? rows : 0, cols, ok ? c : 0, sub);
  if (ok && sub == 0) {
    if (c < cols_a) out_a[c] = acc;
    else out_b[c - cols_a] = acc;
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int pow2_shift(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return ((1 << s) == v) ? s : -1;
}

template <bool BWD>
static void gn_launch(const GnP& p, hipStream_t s) {
  const long groups = (long)p.B * p.G;
  const int n = (p.C / p.G) * p.HW;
  const bool reg_ok = p.sh >= 2 && n <= 16384 && (((uintptr_t)p.x) & 15) == 0 &&
                      (((uintptr_t)p.y) & 15) == 0 && (!BWD || ((((uintptr_t)p.dy) | ((uintptr_t)p.add)) & 15) == 0);
  // Few groups (training batches: 32 samples x 32 groups = one wave per SIMD in the 64-lane forms, every load's
  // latency exposed): the whole workgroup takes one group, four times the waves with a quarter of the loads each.
  const char* wide = getenv("I2R_GN_WIDE");
  const bool few = groups <= 2048 && !(wide && wide[0] == '0');
#define GN_GO(LPG, NV)                                                                                          \
  do {                                                                                                          \
    dim3 grid(BWD ? p.G * i2r_cdiv(p.B, 256 / LPG) : i2r_cdiv(groups, 256 / LPG));                              \
    if (BWD)                                                                                                    \
      hipLaunchKernelGGL((gn_bwd_reg_kernel<LPG, NV>), grid, dim3(256), 0, s, p);                               \
    else                                                                                                        \
      hipLaunchKernelGGL((gn_fwd_reg_kernel<LPG, NV>), grid, dim3(256), 0, s, p);                               \
  } while (0)
  const char* hw1 = getenv("I2R_GN_HW1");
  if (BWD && p.HW == 1 && p.C / p.G <= 64 && !(hw1 && hw1[0] == '0')) {
    const int cpg = p.C / p.G;
    if (cpg <= 16)
      hipLaunchKernelGGL(gn_bwd_hw1_kernel<16>, dim3(p.G * i2r_cdiv(p.B, 16)), dim3(256), 0, s, p);
    else if (cpg <= 32)
      hipLaunchKernelGGL(gn_bwd_hw1_kernel<32>, dim3(p.G * i2r_cdiv(p.B, 8)), dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL(gn_bwd_hw1_kernel<64>, dim3(p.G * i2r_cdiv(p.B, 4)), dim3(256), 0, s, p);
  } else if (!BWD && p.HW == 1 && p.C / p.G <= 64 && !p.x2 && !(hw1 && hw1[0] == '0')) {
    const int cpg = p.C / p.G;
    if (cpg <= 16)
      hipLaunchKernelGGL(gn_fwd_hw1_kernel<16>, dim3(i2r_cdiv(groups, 16)), dim3(256), 0, s, p);
    else if (cpg <= 32)
      hipLaunchKernelGGL(gn_fwd_hw1_kernel<32>, dim3(i2r_cdiv(groups, 8)), dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL(gn_fwd_hw1_kernel<64>, dim3(i2r_cdiv(groups, 4)), dim3(256), 0, s, p);
  } else if (!reg_ok) {
    dim3 grid(i2r_cdiv(groups, GN_WAVES));
    if (BWD)
      hipLaunchKernelGGL(groupnorm_bwd_kernel, grid, dim3(64 * GN_WAVES), 0, s, p);
    else
      hipLaunchKernelGGL(groupnorm_fwd_kernel, grid, dim3(64 * GN_WAVES), 0, s, p);
  } else if (!BWD && n <= 256 && groups >= 16384 && !(getenv("I2R_GN_SMALL") && getenv("I2R_GN_SMALL")[0] == '0')) {
    // many small groups: four groups per lane slot, their loads in flight together
    if (n <= 64)
      hipLaunchKernelGGL((gn_fwd_small_kernel<16, 4>), dim3(i2r_cdiv(groups, 4 * 16)), dim3(256), 0, s, p);
    else if (n <= 128)
      hipLaunchKernelGGL((gn_fwd_small_kernel<32, 4>), dim3(i2r_cdiv(groups, 4 * 8)), dim3(256), 0, s, p);
    else   // (256 elements: 8x8 x 4 channels, shape A's top level; the decoder concats of the 4x4 level)
      hipLaunchKernelGGL((gn_fwd_small_kernel<64, 4>), dim3(i2r_cdiv(groups, 4 * 4)), dim3(256), 0, s, p);
  } else if (n <= 64) {
    GN_GO(16, 1);
  } else if (n <= 128) {
    GN_GO(32, 1);  // 8 channels x 4x4 (256-wide levels at 4x4): 32 lanes carry data, not half of 64
  } else if (n <= 256) {
    GN_GO(64, 1);
  } else if (n <= 512) {
    GN_GO(64, 2);
  } else if (n <= 1024) {
    if (few && n > 512) GN_GO(256, 1); else GN_GO(64, 4);
  } else if (n <= 2048) {
    if (few) GN_GO(256, 2); else GN_GO(64, 8);
  } else if (n <= 4096) {
    if (few) GN_GO(256, 4); else GN_GO(64, 16);
  } else if (n <= 8192) {
    GN_GO(256, 8);
  } else {
    GN_GO(256, 16);
  }
#undef GN_GO
}

extern "C" int i2r_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, const float* ss,
                                     int ss_batch, float* y, float* mean, float* rstd, int B, int C, int HW, int G,
                                     float eps, int act, float drop_p, uint64_t seed, uint64_t offset,
                                     const uint64_t* rng_step, int split_out, i2r_stream_t stream) {
  I2R_REQUIRE(x && gamma && beta && y, "groupnorm_fwd: null pointer");
  I2R_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "groupnorm_fwd: C=%d not divisible by G=%d", C, G);
  I2R_REQUIRE((mean == nullptr) == (rstd == nullptr), "groupnorm_fwd: mean and rstd must both be given or both null");
  I2R_REQUIRE(!ss || ss_batch == 1 || ss_batch == B, "groupnorm_fwd: scale/shift batch %d vs %d", ss_batch, B);
  I2R_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "groupnorm_fwd: dropout p out of range");
  GnP p{};
  p.x = x; p.gamma = gamma; p.beta = beta; p.ss = ss; p.y = y; p.mean_out = mean; p.rstd_out = rstd;
  p.rng_step = rng_step; p.ss_batch = ss_batch; p.B = B; p.C = C; p.HW = HW; p.G = G; p.sh = pow2_shift(HW);
  p.act = act; p.eps = eps; p.drop_p = drop_p; p.seed = seed; p.offset = offset; p.split_out = split_out;
  gn_launch<false>(p, (hipStream_t)stream);
  return i2r_check_launch("groupnorm_fwd");
}

// Forward over the channel concatenation [x1 | x2] (C1 + C2 channels) without materialising it (decoder skips,
// models/networks.py:458 followed by norm0 at :168).  Register kernels only: H*W a power of two >= 4, groups <= 16384 elements.
extern "C" int i2r_groupnorm_cat_fwd_f32(const float* x1, const float* x2, int C1, int C2, const float* gamma,
                                         const float* beta, const float* ss, int ss_batch, float* y, int B, int HW,
                                         int G, float eps, int act, int split_out, i2r_stream_t stream) {
  I2R_REQUIRE(x1 && x2 && gamma && beta && y && C1 > 0 && C2 > 0, "groupnorm_cat_fwd: bad argument");
  const int C = C1 + C2;
  I2R_REQUIRE(B > 0 && HW > 0 && G > 0 && C % G == 0, "groupnorm_cat_fwd: C=%d not divisible by G=%d", C, G);
  I2R_REQUIRE(!ss || ss_batch == 1 || ss_batch == B, "groupnorm_cat_fwd: scale/shift batch %d vs %d", ss_batch, B);
  GnP p{};
  p.x = x1; p.x2 = x2; p.C1 = C1; p.gamma = gamma; p.beta = beta; p.ss = ss; p.y = y; p.ss_batch = ss_batch;
  p.B = B; p.C = C; p.HW = HW; p.G = G; p.sh = pow2_shift(HW); p.act = act; p.eps = eps; p.split_out = split_out;
  I2R_REQUIRE(p.sh >= 2 && (C / G) * HW <= 16384 && ((((uintptr_t)x1) | ((uintptr_t)x2) | ((uintptr_t)y)) & 15) == 0,
              "groupnorm_cat_fwd: needs H*W a power of two >= 4 (got %d), groups <= 16384 elements, 16-B aligned "
              "tensors", HW);
  gn_launch<false>(p, (hipStream_t)stream);
  return i2r_check_launch("groupnorm_cat_fwd");
}

extern "C" int i2r_groupnorm_bwd_f32(const float* x, const float* gamma, const float* beta, const float* ss,
                                     int ss_batch, const float* mean, const float* rstd, const float* dy, float* dx,
                                     float* dgb_part, float* dss, const float* dx_add, int B, int C, int HW, int G,
                                     int act, float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_step,
                                     i2r_stream_t stream) {
  I2R_REQUIRE(x && gamma && beta && mean && rstd && dy && dx && dgb_part, "groupnorm_bwd: null pointer");
  I2R_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "groupnorm_bwd: C=%d not divisible by G=%d", C, G);
  I2R_REQUIRE(!ss || ss_batch == B, "groupnorm_bwd: scale/shift must be per-sample (batch %d vs %d)", ss_batch, B);
  I2R_REQUIRE((ss != nullptr) == (dss != nullptr), "groupnorm_bwd: dss must be given iff ss is");
  GnP p{};
  p.x = x; p.gamma = gamma; p.beta = beta; p.ss = ss; p.mean_in = mean; p.rstd_in = rstd; p.dy = dy; p.y = dx;
  p.dgb_part = dgb_part; p.dss = dss; p.add = dx_add; p.rng_step = rng_step; p.ss_batch = ss_batch; p.B = B; p.C = C; p.HW = HW;
  p.G = G; p.sh = pow2_shift(HW); p.act = act; p.drop_p = drop_p; p.seed = seed; p.offset = offset;
  gn_launch<true>(p, (hipStream_t)stream);
  return i2r_check_launch("groupnorm_bwd");
}

// Same, with dgamma / dbeta ADDED into caller-zeroed (C) buffers by fp32 atomics (32 adds per address at batch 32):
// no (B, 2C) partial tensor and no reduction launch.  Summation order over the batch is not fixed.
extern "C" int i2r_groupnorm_bwd_acc_f32(const float* x, const float* gamma, const float* beta, const float* ss,
                                         int ss_batch, const float* mean, const float* rstd, const float* dy,
                                         float* dx, float* dgamma_acc, float* dbeta_acc, float* dss,
                                         const float* dx_add, int B, int C, int HW, int G, int act, float drop_p,
                                         uint64_t seed, uint64_t offset, const uint64_t* rng_step,
                                         i2r_stream_t stream) {
  I2R_REQUIRE(x && gamma && beta && mean && rstd && dy && dx && dgamma_acc && dbeta_acc,
              "groupnorm_bwd_acc: null pointer");
  I2R_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "groupnorm_bwd_acc: C=%d not divisible by G=%d", C, G);
  I2R_REQUIRE(!ss || ss_batch == B, "groupnorm_bwd_acc: scale/shift must be per-sample (batch %d vs %d)", ss_batch, B);
  I2R_REQUIRE((ss != nullptr) == (dss != nullptr), "groupnorm_bwd_acc: dss must be given iff ss is");
  GnP p{};
  p.x = x; p.gamma = gamma; p.beta = beta; p.ss = ss; p.mean_in = mean; p.rstd_in = rstd; p.dy = dy; p.y = dx;
  { const char* e = getenv("I2R_GN_BWD_LDS"); p.gb_lds = !(e && e[0] == '0'); }
  p.dg_acc = dgamma_acc; p.db_acc = dbeta_acc; p.dss = dss; p.add = dx_add; p.rng_step = rng_step; p.ss_batch = ss_batch; p.B = B;
  p.C = C; p.HW = HW; p.G = G; p.sh = pow2_shift(HW); p.act = act; p.drop_p = drop_p; p.seed = seed; p.offset = offset;
  gn_launch<true>(p, (hipStream_t)stream);
  return i2r_check_launch("groupnorm_bwd_acc");
}

extern "C" int i2r_reduce_rows_f32(const float* in, float* out, int rows, int cols, i2r_stream_t stream) {
  I2R_REQUIRE(in && out && rows > 0 && cols > 0, "reduce_rows: bad argument");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(i2r_cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream, in, out, rows,
                     cols);
  return i2r_check_launch("reduce_rows");
}

extern "C" int i2r_reduce_rows2_f32(const float* in, float* out_a, float* out_b, int rows, int cols_a, int cols_b,
                                    i2r_stream_t stream) {
  I2R_REQUIRE(in && out_a && out_b && rows > 0 && cols_a > 0 && cols_b > 0, "reduce_rows2: bad argument");
  hipLaunchKernelGGL(reduce_rows2_kernel, dim3(i2r_cdiv(cols_a + cols_b, 64)), dim3(256), 0, (hipStream_t)stream, in,
                     out_a, out_b, rows, cols_a, cols_b);
  return i2r_check_launch("reduce_rows2");
}
