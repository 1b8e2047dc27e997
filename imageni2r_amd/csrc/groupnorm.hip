// Fused GroupNorm family (models/networks.py:104-106 as used at :168, :173, :182, :460):
//   y = dropout( act( gn(x) * (1 + scale) + shift ) )      act in {identity, SiLU}; scale/shift optional (adaptive
//   modulation from the block's `affine` Linear, :170-173); dropout only in training (:177).
// One wavefront owns one (sample, group): a group is (C/G)*H*W contiguous fp32 in NCHW, at most 16 KiB in every
// shipped config, so the statistics are a wave-local shuffle reduction and the tensor makes one trip from HBM
// (passes 2 and 3 re-read the group from L1/L2).  HBM-bound: 8 B/element forward, 16 B/element backward.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

#define GN_WAVES 4

__device__ __forceinline__ float keep_scale(uint64_t seed, uint64_t offset, long i, float p) {
  uint4 r = philox4x32(make_uint4((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)),
                       make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
  return (u32_to_unit(r.x) > p) ? 1.0f / (1.0f - p) : 0.0f;
}

__global__ __launch_bounds__(64 * GN_WAVES) void groupnorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ ss, int ss_batch, float* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out, int B, int C, int HW, int G, float eps, int act, float drop_p, uint64_t seed,
    uint64_t offset) {
  int lane = threadIdx.x & 63;
  long wg = (long)blockIdx.x * GN_WAVES + (threadIdx.x >> 6);  // (b, g) index, wave-uniform
  if (wg >= (long)B * G) return;
  int b = (int)(wg / G), g = (int)(wg % G);
  int cpg = C / G;
  int n = cpg * HW;
  long base = ((long)b * C + (long)g * cpg) * HW;
  const float* xp = x + base;

  float s = 0.0f;
  for (int i = lane; i < n; i += 64) s += xp[i];
  float mean = wave_sum(s) / (float)n;
  float v = 0.0f;
  for (int i = lane; i < n; i += 64) {
    float d = xp[i] - mean;
    v += d * d;
  }
  float rstd = 1.0f / sqrtf(wave_sum(v) / (float)n + eps);
  if (lane == 0 && mean_out) {
    mean_out[wg] = mean;
    rstd_out[wg] = rstd;
  }
  const float* ssp = ss ? ss + (long)(ss_batch == 1 ? 0 : b) * 2 * C : nullptr;
  for (int i = lane; i < n; i += 64) {
    int c = g * cpg + i / HW;
    float u = (xp[i] - mean) * rstd * gamma[c] + beta[c];
    if (ssp) u = u * (ssp[c] + 1.0f) + ssp[C + c];
    if (act) u = silu_f(u);
    if (drop_p > 0.0f) u *= keep_scale(seed, offset, base + i, drop_p);
    y[base + i] = u;
  }
}

// Backward.  Pass A: per-channel sums over HW of dn and dn*xhat (these are the per-sample dbeta/dgamma partials, and
// combined with gamma they give the two group moments that dx needs); pass B: dx.
__global__ __launch_bounds__(64 * GN_WAVES) void groupnorm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ ss, int ss_batch, const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dgamma_part,
    float* __restrict__ dbeta_part, float* __restrict__ dss, int B, int C, int HW, int G, int act, float drop_p,
    uint64_t seed, uint64_t offset) {
  int lane = threadIdx.x & 63;
  long wg = (long)blockIdx.x * GN_WAVES + (threadIdx.x >> 6);
  if (wg >= (long)B * G) return;
  int b = (int)(wg / G), g = (int)(wg % G);
  int cpg = C / G;
  int n = cpg * HW;
  long base = ((long)b * C + (long)g * cpg) * HW;
  const float* xp = x + base;
  const float* dyp = dy + base;
  float mean = mean_in[wg], rstd = rstd_in[wg];
  const float* ssp = ss ? ss + (long)(ss_batch == 1 ? 0 : b) * 2 * C : nullptr;

  float S1 = 0.0f, S2 = 0.0f;  // sum_c gamma_c * sum_hw dn,  sum_c gamma_c * sum_hw dn*xhat
  for (int cc = 0; cc < cpg; ++cc) {
    int c = g * cpg + cc;
    float gm = gamma[c], bt = beta[c];
    float sc1 = ssp ? ssp[c] + 1.0f : 1.0f, sh = ssp ? ssp[C + c] : 0.0f;
    float a_du = 0.0f, a_dun = 0.0f, a_dn = 0.0f, a_dnx = 0.0f;
    for (int p = lane; p < HW; p += 64) {
      int i = cc * HW + p;
      float xh = (xp[i] - mean) * rstd;
      float nn = xh * gm + bt;
      float u = nn * sc1 + sh;
      float du = dyp[i];
      if (drop_p > 0.0f) du *= keep_scale(seed, offset, base + i, drop_p);
      if (act) du *= dsilu_f(u);
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
      dgamma_part[(long)b * C + c] = a_dnx;
      dbeta_part[(long)b * C + c] = a_dn;
      if (dss) {
        dss[(long)b * 2 * C + c] = a_dun;      // d scale
        dss[(long)b * 2 * C + C + c] = a_du;   // d shift
      }
    }
    S1 += gm * a_dn;
    S2 += gm * a_dnx;
  }
  float inv_n = 1.0f / (float)n;
  float m1 = S1 * inv_n, m2 = S2 * inv_n;
  for (int i = lane; i < n; i += 64) {
    int c = g * cpg + i / HW;
    float gm = gamma[c];
    float sc1 = ssp ? ssp[c] + 1.0f : 1.0f;
    float xh = (xp[i] - mean) * rstd;
    float du = dyp[i];
    if (drop_p > 0.0f) du *= keep_scale(seed, offset, base + i, drop_p);
    if (act) {
      float u = (xh * gm + beta[c]) * sc1 + (ssp ? ssp[C + c] : 0.0f);
      du *= dsilu_f(u);
    }
    float dxh = du * sc1 * gm;
    dx[base + i] = rstd * (dxh - m1 - xh * m2);
  }
}

// out[c] = sum_r in[r][c]   (deterministic batch reduction of the per-sample dgamma/dbeta partials, bias grads ...)
__global__ void reduce_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float acc = 0.0f;
  for (int r = 0; r < rows; ++r) acc += in[(long)r * cols + c];
  out[c] = acc;
}

extern "C" int i2r_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, const float* ss,
                                     int ss_batch, float* y, float* mean, float* rstd, int B, int C, int HW, int G,
                                     float eps, int act, float drop_p, uint64_t seed, uint64_t offset,
                                     i2r_stream_t stream) {
  I2R_REQUIRE(x && gamma && beta && y, "groupnorm_fwd: null pointer");
  I2R_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "groupnorm_fwd: C=%d not divisible by G=%d", C, G);
  I2R_REQUIRE((mean == nullptr) == (rstd == nullptr), "groupnorm_fwd: mean and rstd must both be given or both null");
  I2R_REQUIRE(!ss || ss_batch == 1 || ss_batch == B, "groupnorm_fwd: scale/shift batch %d vs %d", ss_batch, B);
  I2R_REQUIRE(drop_p >= 0.0f && drop_p < 1.0f, "groupnorm_fwd: dropout p out of range");
  long waves = (long)B * G;
  hipLaunchKernelGGL(groupnorm_fwd_kernel, dim3(i2r_cdiv(waves, GN_WAVES)), dim3(64 * GN_WAVES), 0,
                     (hipStream_t)stream, x, gamma, beta, ss, ss_batch, y, mean, rstd, B, C, HW, G, eps, act, drop_p,
                     seed, offset);
  return i2r_check_launch("groupnorm_fwd");
}

extern "C" int i2r_groupnorm_bwd_f32(const float* x, const float* gamma, const float* beta, const float* ss,
                                     int ss_batch, const float* mean, const float* rstd, const float* dy, float* dx,
                                     float* dgamma_part, float* dbeta_part, float* dss, int B, int C, int HW, int G,
                                     int act, float drop_p, uint64_t seed, uint64_t offset, i2r_stream_t stream) {
  I2R_REQUIRE(x && gamma && beta && mean && rstd && dy && dx && dgamma_part && dbeta_part,
              "groupnorm_bwd: null pointer");
  I2R_REQUIRE(B > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, "groupnorm_bwd: C=%d not divisible by G=%d", C, G);
  I2R_REQUIRE(!ss || ss_batch == B, "groupnorm_bwd: scale/shift must be per-sample (batch %d vs %d)", ss_batch, B);
  I2R_REQUIRE((ss != nullptr) == (dss != nullptr), "groupnorm_bwd: dss must be given iff ss is");
  long waves = (long)B * G;
  hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3(i2r_cdiv(waves, GN_WAVES)), dim3(64 * GN_WAVES), 0,
                     (hipStream_t)stream, x, gamma, beta, ss, ss_batch, mean, rstd, dy, dx, dgamma_part, dbeta_part,
                     dss, B, C, HW, G, act, drop_p, seed, offset);
  return i2r_check_launch("groupnorm_bwd");
}

extern "C" int i2r_reduce_rows_f32(const float* in, float* out, int rows, int cols, i2r_stream_t stream) {
  I2R_REQUIRE(in && out && rows > 0 && cols > 0, "reduce_rows: bad argument");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(i2r_cdiv(cols, 256)), dim3(256), 0, (hipStream_t)stream, in, out, rows,
                     cols);
  return i2r_check_launch("reduce_rows");
}
