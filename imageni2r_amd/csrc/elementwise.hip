// HBM-bound elementwise / small-reduction kernels of the masked-EDM path:
//   EDM preconditioning (models/networks.py:660-667), masked noise injection (models/our.py:71-77),
//   mask-weighted loss + its gradient seed (models/our.py:63-66), fp64 Heun updates (models/sampler.py:45-63),
//   2x2 box down / nearest up resampling (models/networks.py:82-85 with resample_filter [1,1]),
//   channel concat/split for the decoder skips (models/networks.py:458), SiLU backward, positional embedding
//   (models/networks.py:200-206).
// Built with -ffp-contract=off so the fp64 sampler arithmetic keeps the reference's operation order.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

static inline int ew_grid(long total, int block = 256) {
  long g = (total + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
#define GRID_STRIDE(i, total) \
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ EDM preconditioning
struct EdmCoef {
  float c_skip, c_out, c_in;
};
__device__ __forceinline__ EdmCoef edm_coef(float sigma, float sd) {
  float s2 = sigma * sigma + sd * sd;
  EdmCoef c;
  c.c_skip = (sd * sd) / s2;
  c.c_out = sigma * sd / sqrtf(s2);
  c.c_in = 1.0f / sqrtf(sd * sd + sigma * sigma);
  return c;
}

// xin = c_in(sigma_b) * float(x);  c_noise[b] = ln(sigma_b)/4.   sigma has nsig entries (1 = broadcast).
template <typename T>
__global__ void edm_scale_input_kernel(const T* __restrict__ x, const float* __restrict__ sigma, int nsig,
                                       float* __restrict__ xin, float* __restrict__ c_noise, int B, long chw,
                                       float sd) {
  long total = (long)B * chw;
  GRID_STRIDE(i, total) {
    int b = (int)(i / chw);
    float s = sigma[nsig == 1 ? 0 : b];
    xin[i] = edm_coef(s, sd).c_in * (float)x[i];
  }
  if (blockIdx.x == 0)
    for (int j = threadIdx.x; j < nsig; j += blockDim.x) c_noise[j] = logf(sigma[j]) / 4.0f;
}

// D = c_skip * float(x) + c_out * F
template <typename T>
__global__ void edm_combine_kernel(const T* __restrict__ x, const float* __restrict__ F,
                                   const float* __restrict__ sigma, int nsig, float* __restrict__ D, int B, long chw,
                                   float sd) {
  long total = (long)B * chw;
  GRID_STRIDE(i, total) {
    int b = (int)(i / chw);
    EdmCoef c = edm_coef(sigma[nsig == 1 ? 0 : b], sd);
    D[i] = c.c_skip * (float)x[i] + c.c_out * F[i];
  }
}

// dF = c_out * dD (the only gradient the training step needs through the preconditioner: SURVEY.md §0.10)
__global__ void edm_combine_bwd_kernel(const float* __restrict__ dD, const float* __restrict__ sigma, int nsig,
                                       float* __restrict__ dF, int B, long chw, float sd) {
  long total = (long)B * chw;
  GRID_STRIDE(i, total) {
    int b = (int)(i / chw);
    dF[i] = edm_coef(sigma[nsig == 1 ? 0 : b], sd).c_out * dD[i];
  }
}

extern "C" int i2r_edm_scale_input(const void* x, int x_is_f64, const float* sigma, int nsig, float* xin,
                                   float* c_noise, int B, long chw, float sigma_data, i2r_stream_t stream) {
  I2R_REQUIRE(x && sigma && xin && c_noise, "edm_scale_input: null pointer");
  I2R_REQUIRE(nsig == 1 || nsig == B, "edm_scale_input: sigma has %d entries, batch is %d", nsig, B);
  int nb = ew_grid((long)B * chw);
  if (x_is_f64)
    hipLaunchKernelGGL(edm_scale_input_kernel<double>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const double*)x,
                       sigma, nsig, xin, c_noise, B, chw, sigma_data);
  else
    hipLaunchKernelGGL(edm_scale_input_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                       sigma, nsig, xin, c_noise, B, chw, sigma_data);
  return i2r_check_launch("edm_scale_input");
}

extern "C" int i2r_edm_combine(const void* x, int x_is_f64, const float* F, const float* sigma, int nsig, float* D,
                               int B, long chw, float sigma_data, i2r_stream_t stream) {
  I2R_REQUIRE(x && F && sigma && D, "edm_combine: null pointer");
  I2R_REQUIRE(nsig == 1 || nsig == B, "edm_combine: sigma has %d entries, batch is %d", nsig, B);
  int nb = ew_grid((long)B * chw);
  if (x_is_f64)
    hipLaunchKernelGGL(edm_combine_kernel<double>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const double*)x, F,
                       sigma, nsig, D, B, chw, sigma_data);
  else
    hipLaunchKernelGGL(edm_combine_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, F,
                       sigma, nsig, D, B, chw, sigma_data);
  return i2r_check_launch("edm_combine");
}

extern "C" int i2r_edm_combine_bwd(const float* dD, const float* sigma, int nsig, float* dF, int B, long chw,
                                   float sigma_data, i2r_stream_t stream) {
  I2R_REQUIRE(dD && sigma && dF, "edm_combine_bwd: null pointer");
  I2R_REQUIRE(nsig == 1 || nsig == B, "edm_combine_bwd: sigma has %d entries, batch is %d", nsig, B);
  hipLaunchKernelGGL(edm_combine_bwd_kernel, dim3(ew_grid((long)B * chw)), dim3(256), 0, (hipStream_t)stream, dD,
                     sigma, nsig, dF, B, chw, sigma_data);
  return i2r_check_launch("edm_combine_bwd");
}

// ------------------------------------------------------------------ training-time noise (our.py:71-77)
// sigma_b = exp(rnd_b * P_std + P_mean); weight_b = (sigma^2 + sd^2) / (sigma*sd)^2;
// xn = x + (noise * sigma_b) * mask.   rnd/noise are either injected (tests) or drawn here with Philox + Box-Muller.
__global__ void edm_noise_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                 const float* __restrict__ rnd_in, const float* __restrict__ noise_in,
                                 float* __restrict__ xn, float* __restrict__ sigma, float* __restrict__ weight, int B,
                                 long chw, float P_mean, float P_std, float sd, uint64_t seed, uint64_t offset,
                                 const uint64_t* __restrict__ rng_step) {
  long total = (long)B * chw;
  if (rng_step) offset += rng_step[0] << 20;
  uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
  GRID_STRIDE(i, total) {
    int b = (int)(i / chw);
    float rnd;
    if (rnd_in) {
      rnd = rnd_in[b];
    } else {
      uint4 r = philox4x32(make_uint4((uint32_t)b, 0x5167u, (uint32_t)offset, (uint32_t)(offset >> 32)), key);
      rnd = sqrtf(-2.0f * logf(u32_to_unit(r.x))) * cosf(6.283185307179586f * u32_to_unit(r.y));
    }
    float s = expf(rnd * P_std + P_mean);
    float n;
    if (noise_in) {
      n = noise_in[i];
    } else {
      uint4 r = philox4x32(make_uint4((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)offset, (uint32_t)(offset >> 32) ^ 0xA5A5u), key);
      n = sqrtf(-2.0f * logf(u32_to_unit(r.x))) * cosf(6.283185307179586f * u32_to_unit(r.y));
    }
    xn[i] = x[i] + (n * s) * mask[i];
    if (i % chw == 0) {
      sigma[b] = s;
      float t = s * sd;
      weight[b] = (s * s + sd * sd) / (t * t);
    }
  }
}

extern "C" int i2r_edm_noise(const float* x, const float* mask, const float* rnd_in, const float* noise_in,
                             float* xn, float* sigma, float* weight, int B, long chw, float P_mean, float P_std,
                             float sigma_data, uint64_t seed, uint64_t offset, const uint64_t* rng_step,
                             i2r_stream_t stream) {
  I2R_REQUIRE(x && mask && xn && sigma && weight, "edm_noise: null pointer");
  hipLaunchKernelGGL(edm_noise_kernel, dim3(ew_grid((long)B * chw)), dim3(256), 0, (hipStream_t)stream, x, mask,
                     rnd_in, noise_in, xn, sigma, weight, B, chw, P_mean, P_std, sigma_data, seed, offset,
                     rng_step);
  return i2r_check_launch("edm_noise");
}

// Device-resident step counter of the training RNG streams (dropout masks, noise draws): kernels that take
// `rng_step` add (*rng_step << 20) to their Philox offset, so a captured HIP graph draws fresh randomness on every
// replay once this increment is part of the graph.
__global__ void counter_add_kernel(uint64_t* ctr, uint64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += inc;
}
extern "C" int i2r_counter_add_u64(uint64_t* counter, uint64_t inc, i2r_stream_t stream) {
  I2R_REQUIRE(counter, "counter_add: null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, inc);
  return i2r_check_launch("counter_add");
}

// ------------------------------------------------------------------ masked EDM loss (our.py:63-66)
// loss = mean_i( weight_b * (D*mask - x*mask)^2 );  dD = 2 * weight_b * mask * (D*mask - x*mask) / N.
// Stage 1: per-block fp32 partials into `partials` (<= 1024 blocks); stage 2: one wave sums them in fp64.
__global__ void masked_loss_stage1(const float* __restrict__ D, const float* __restrict__ x,
                                   const float* __restrict__ mask, const float* __restrict__ weight,
                                   float* __restrict__ dD, float* __restrict__ partials, int B, long chw,
                                   float inv_n) {
  long total = (long)B * chw;
  float acc = 0.0f;
  GRID_STRIDE(i, total) {
    int b = (int)(i / chw);
    float m = mask[i], w = weight[b];
    float diff = D[i] * m - x[i] * m;
    acc += w * (diff * diff);
    if (dD) dD[i] = 2.0f * w * m * diff * inv_n;
  }
  acc = wave_sum(acc);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void masked_loss_stage2(const float* __restrict__ partials, int n, float* __restrict__ loss,
                                   double inv_n) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) acc += (double)partials[i];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) loss[0] = (float)(acc * inv_n);
}

extern "C" int i2r_masked_edm_loss(const float* D, const float* x, const float* mask, const float* weight,
                                   float* loss, float* dD, float* partials, int n_partials, int B, long chw,
                                   i2r_stream_t stream) {
  I2R_REQUIRE(D && x && mask && weight && loss && partials, "masked_edm_loss: null pointer");
  I2R_REQUIRE(n_partials >= 1, "masked_edm_loss: need a partials workspace");
  long total = (long)B * chw;
  int nb = ew_grid(total);
  if (nb > n_partials) nb = n_partials;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(masked_loss_stage1, dim3(nb), dim3(256), 0, (hipStream_t)stream, D, x, mask, weight, dD,
                     partials, B, chw, (float)(1.0 / (double)total));
  hipLaunchKernelGGL(masked_loss_stage2, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, nb, loss,
                     1.0 / (double)total);
  return i2r_check_launch("masked_edm_loss");
}

// ------------------------------------------------------------------ Heun sampler updates (sampler.py:45-63), fp64 state
__global__ void heun_init_kernel(const float* __restrict__ latents, double* __restrict__ x, long n, double t0) {
  GRID_STRIDE(i, n) x[i] = (double)latents[i] * t0;
}
__global__ void heun_init_d_kernel(const double* __restrict__ latents, double* __restrict__ x, long n, double t0) {
  GRID_STRIDE(i, n) x[i] = latents[i] * t0;
}
// d_cur = (x_hat - D) / t_hat;  x_next = x_hat + (t_next - t_hat) * d_cur
__global__ void heun_euler_kernel(const double* __restrict__ x_hat, const float* __restrict__ D,
                                  double* __restrict__ d_cur, double* __restrict__ x_next, long n, double t_hat,
                                  double t_next) {
  double dt = t_next - t_hat;
  GRID_STRIDE(i, n) {
    double xh = x_hat[i];
    double d = (xh - (double)D[i]) / t_hat;
    d_cur[i] = d;
    x_next[i] = xh + dt * d;
  }
}
// d' = (x_next - D') / t_next;  x_out = x_hat + (t_next - t_hat) * (0.5 d_cur + 0.5 d')
__global__ void heun_correct_kernel(const double* __restrict__ x_hat, const double* __restrict__ x_next,
                                    const float* __restrict__ Dp, const double* __restrict__ d_cur,
                                    double* __restrict__ x_out, long n, double t_hat, double t_next) {
  double dt = t_next - t_hat;
  GRID_STRIDE(i, n) {
    double dp = (x_next[i] - (double)Dp[i]) / t_next;
    x_out[i] = x_hat[i] + dt * (0.5 * d_cur[i] + 0.5 * dp);
  }
}

extern "C" int i2r_heun_init_f64(const float* latents, double* x, long n, double t0, i2r_stream_t stream) {
  I2R_REQUIRE(latents && x && n > 0, "heun_init: bad argument");
  hipLaunchKernelGGL(heun_init_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, latents, x, n, t0);
  return i2r_check_launch("heun_init");
}
extern "C" int i2r_heun_init_f64_from_f64(const double* latents, double* x, long n, double t0, i2r_stream_t stream) {
  I2R_REQUIRE(latents && x && n > 0, "heun_init: bad argument");
  hipLaunchKernelGGL(heun_init_d_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, latents, x, n, t0);
  return i2r_check_launch("heun_init");
}
extern "C" int i2r_heun_euler_f64(const double* x_hat, const float* D, double* d_cur, double* x_next, long n,
                                  double t_hat, double t_next, i2r_stream_t stream) {
  I2R_REQUIRE(x_hat && D && d_cur && x_next && n > 0 && t_hat != 0.0, "heun_euler: bad argument");
  hipLaunchKernelGGL(heun_euler_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x_hat, D, d_cur, x_next,
                     n, t_hat, t_next);
  return i2r_check_launch("heun_euler");
}
extern "C" int i2r_heun_correct_f64(const double* x_hat, const double* x_next, const float* Dp, const double* d_cur,
                                    double* x_out, long n, double t_hat, double t_next, i2r_stream_t stream) {
  I2R_REQUIRE(x_hat && x_next && Dp && d_cur && x_out && n > 0 && t_next != 0.0, "heun_correct: bad argument");
  hipLaunchKernelGGL(heun_correct_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x_hat, x_next, Dp,
                     d_cur, x_out, n, t_hat, t_next);
  return i2r_check_launch("heun_correct");
}

// ------------------------------------------------------------------ resampling (networks.py:82-85)
// y[p, oy, ox] = scale * sum_{2x2} x[p, 2oy+dy, 2ox+dx]   (scale 0.25 = forward box filter; 1.0 = adjoint of nearest-up)
__global__ void down2_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int H, int W,
                             float scale) {
  int Ho = H / 2, Wo = W / 2;
  long total = planes * Ho * Wo;
  GRID_STRIDE(i, total) {
    int ox = (int)(i % Wo);
    long r = i / Wo;
    int oy = (int)(r % Ho);
    long p = r / Ho;
    const float* s = x + (p * H + 2 * oy) * W + 2 * ox;
    // same summation order as a 2x2 depthwise conv: row-major taps
    y[i] = (((s[0] * scale + s[1] * scale) + s[W] * scale) + s[W + 1] * scale);
  }
}
// y[p, oy, ox] = scale * x[p, oy/2, ox/2]   (scale 1 = nearest up; 0.25 = adjoint of the box filter)
__global__ void up2_kernel(const float* __restrict__ x, float* __restrict__ y, long planes, int H, int W,
                           float scale) {
  int Ho = H * 2, Wo = W * 2;
  long total = planes * Ho * Wo;
  GRID_STRIDE(i, total) {
    int ox = (int)(i % Wo);
    long r = i / Wo;
    int oy = (int)(r % Ho);
    long p = r / Ho;
    y[i] = scale * x[(p * H + (oy >> 1)) * W + (ox >> 1)];
  }
}
extern "C" int i2r_down2_f32(const float* x, float* y, long planes, int H, int W, float scale, i2r_stream_t stream) {
  I2R_REQUIRE(x && y && planes > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "down2: bad shape %dx%d", H, W);
  hipLaunchKernelGGL(down2_kernel, dim3(ew_grid(planes * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, x, y,
                     planes, H, W, scale);
  return i2r_check_launch("down2");
}
extern "C" int i2r_up2_f32(const float* x, float* y, long planes, int H, int W, float scale, i2r_stream_t stream) {
  I2R_REQUIRE(x && y && planes > 0 && H >= 1 && W >= 1, "up2: bad shape");
  hipLaunchKernelGGL(up2_kernel, dim3(ew_grid(planes * H * W * 4)), dim3(256), 0, (hipStream_t)stream, x, y, planes, H,
                     W, scale);
  return i2r_check_launch("up2");
}

// ------------------------------------------------------------------ channel concat / split (networks.py:458)
// out[b, 0:C1] = a[b], out[b, C1:C1+C2] = c[b]      (hw = H*W elements per channel)
__global__ void concat_kernel(const float* __restrict__ a, const float* __restrict__ c, float* __restrict__ out,
                              int B, long n1, long n2) {
  long per = n1 + n2, total = (long)B * per;
  GRID_STRIDE(i, total) {
    long b = i / per, r = i % per;
    out[i] = (r < n1) ? a[b * n1 + r] : c[b * n2 + (r - n1)];
  }
}
// a[b] (+)= in[b, 0:C1], c[b] (+)= in[b, C1:]
__global__ void split_kernel(const float* __restrict__ in, float* __restrict__ a, float* __restrict__ c, int B,
                             long n1, long n2, int accumulate_c) {
  long per = n1 + n2, total = (long)B * per;
  GRID_STRIDE(i, total) {
    long b = i / per, r = i % per;
    float v = in[i];
    if (r < n1) {
      a[b * n1 + r] = v;
    } else {
      long j = b * n2 + (r - n1);
      c[j] = accumulate_c ? c[j] + v : v;
    }
  }
}
extern "C" int i2r_concat_channels_f32(const float* a, const float* c, float* out, int B, long n1, long n2,
                                       i2r_stream_t stream) {
  I2R_REQUIRE(a && c && out && B > 0 && n1 > 0 && n2 > 0, "concat: bad argument");
  hipLaunchKernelGGL(concat_kernel, dim3(ew_grid((long)B * (n1 + n2))), dim3(256), 0, (hipStream_t)stream, a, c, out,
                     B, n1, n2);
  return i2r_check_launch("concat");
}
extern "C" int i2r_split_channels_f32(const float* in, float* a, float* c, int B, long n1, long n2,
                                      int accumulate_c, i2r_stream_t stream) {
  I2R_REQUIRE(in && a && c && B > 0 && n1 > 0 && n2 > 0, "split: bad argument");
  hipLaunchKernelGGL(split_kernel, dim3(ew_grid((long)B * (n1 + n2))), dim3(256), 0, (hipStream_t)stream, in, a, c,
                     B, n1, n2, accumulate_c);
  return i2r_check_launch("split");
}

// ------------------------------------------------------------------ small pointwise helpers
__global__ void silu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ dy, float* __restrict__ dx,
                                long n) {
  GRID_STRIDE(i, n) dx[i] = dy[i] * dsilu_f(pre[i]);
}
extern "C" int i2r_silu_bwd_f32(const float* pre, const float* dy, float* dx, long n, i2r_stream_t stream) {
  I2R_REQUIRE(pre && dy && dx && n > 0, "silu_bwd: bad argument");
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, pre, dy, dx, n);
  return i2r_check_launch("silu_bwd");
}
// y = a + b  (gradient accumulation where a tensor has two consumers)
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
  GRID_STRIDE(i, n) y[i] = a[i] + b[i];
}
extern "C" int i2r_add_f32(const float* a, const float* b, float* y, long n, i2r_stream_t stream) {
  I2R_REQUIRE(a && b && y && n > 0, "add: bad argument");
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
  return i2r_check_launch("add");
}

// mask = 1 - isnan(x); x_clean = nan_to_num(x, 0)   (our.py:61-62)
__global__ void nan_mask_kernel(const float* __restrict__ x, float* __restrict__ x_clean, float* __restrict__ mask,
                                long n) {
  GRID_STRIDE(i, n) {
    float v = x[i];
    bool isn = v != v;
    x_clean[i] = isn ? 0.0f : v;
    mask[i] = isn ? 0.0f : 1.0f;
  }
}
extern "C" int i2r_nan_mask_f32(const float* x, float* x_clean, float* mask, long n, i2r_stream_t stream) {
  I2R_REQUIRE(x && x_clean && mask && n > 0, "nan_mask: bad argument");
  hipLaunchKernelGGL(nan_mask_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, x_clean, mask, n);
  return i2r_check_launch("nan_mask");
}

// positional embedding (networks.py:200-206): out[b, i] = cos(x_b f_i), out[b, half + i] = sin(x_b f_i),
// f_i = (1/max_positions)^(i/half), all in fp32 like the reference.
__global__ void pos_embedding_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int nch,
                                     float max_positions) {
  int half = nch / 2;
  long total = (long)B * half;
  GRID_STRIDE(i, total) {
    int k = (int)(i % half);
    int b = (int)(i / half);
    float f = powf(1.0f / max_positions, (float)k / (float)half);
    float arg = x[b] * f;
    out[(long)b * nch + k] = cosf(arg);
    out[(long)b * nch + half + k] = sinf(arg);
  }
}
extern "C" int i2r_pos_embedding_f32(const float* x, float* out, int B, int num_channels, int max_positions,
                                     i2r_stream_t stream) {
  I2R_REQUIRE(x && out && B > 0 && num_channels >= 2 && num_channels % 2 == 0, "pos_embedding: bad argument");
  hipLaunchKernelGGL(pos_embedding_kernel, dim3(ew_grid((long)B * num_channels / 2)), dim3(256), 0,
                     (hipStream_t)stream, x, out, B, num_channels, (float)max_positions);
  return i2r_check_launch("pos_embedding");
}

// ------------------------------------------------------------------ power-of-two scale of a tensor (f16x2 dgrad)
// scale_out = {s, 1/s} with s = 2^(13 - floor(log2 max|x|)): s * x peaks in [2^13, 2^14) -- inside fp16's range with a
// factor 4 to spare, and everything down to 2^-27 of the peak stays representable.  max|x| = 0, inf or NaN: s = 1 (a
// non-finite gradient then reaches the optimiser as such and its step is skipped).  `scratch`: two uint32 (running
// maximum, arrival count) that are 0 on entry and are left 0 (the bit pattern of a non-negative float orders like the
// float).  16-byte loads; one atomic per WORKGROUP: with one atomicMax per wave on the same word the 4096 waves of a
// 17 MB tensor queued at the L2 for 50 us -- ten times what reading the tensor takes.
__device__ __forceinline__ void pow2_scale_of(unsigned m, float* __restrict__ scale_out) {
  int e = (int)(m >> 23) - 127;              // floor(log2 max|x|) for normal numbers
  float s = 1.0f;
  if (m != 0 && m < 0x7F800000u) {           // finite, non-zero
    int k = 13 - e;
    if (k > 100) k = 100;                    // (subnormal or tiny peaks: keep 2^k and 2^-k both normal)
    if (k < -100) k = -100;
    s = __builtin_bit_cast(float, (unsigned)(k + 127) << 23);
  }
  scale_out[0] = s;
  scale_out[1] = 1.0f / s;
}
// ONE launch: every workgroup folds its slice's max |x| bits into scratch[0] (memory-side atomic), then takes a ticket
// on scratch[1]; the last to arrive reads the final maximum, writes {s, 1/s} and leaves both words zero again.  The
// grid is bounded (<= ABSMAX_MAX_WG workgroups) because the tickets serialise memory-side at ~0.2 per ns: with one
// ticket per 256 threads of a 1024-workgroup grid the tickets alone took longer than the separate one-thread kernel
// they replace (round 2); at <= 128 of them they cost well under a microsecond.
#define ABSMAX_MAX_WG 128
__global__ __launch_bounds__(256) void absmax_pow2_kernel(const float* __restrict__ x, long n,
                                                          unsigned* __restrict__ scratch,
                                                          float* __restrict__ scale_out) {
  __shared__ unsigned red[4];
  unsigned m = 0;
  const long n4 = (((uintptr_t)x) & 15) == 0 ? (n >> 2) : 0;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  const long stride = (long)gridDim.x * blockDim.x, t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long i = t0;
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four 16-byte loads in flight per thread
    const uint4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
#define ABSMAX4(v)                                                                                                 \
    {                                                                                                               \
      const unsigned qa = v.x & 0x7FFFFFFFu, qb = v.y & 0x7FFFFFFFu, qc = v.z & 0x7FFFFFFFu, qd = v.w & 0x7FFFFFFFu; \
      const unsigned ab = qa > qb ? qa : qb, cd = qc > qd ? qc : qd, q = ab > cd ? ab : cd;                         \
      m = q > m ? q : m;                                                                                            \
    }
    ABSMAX4(v0) ABSMAX4(v1) ABSMAX4(v2) ABSMAX4(v3)
  }
  for (; i < n4; i += stride) {
    const uint4 v = x4[i];
    ABSMAX4(v)
  }
#undef ABSMAX4
  for (long j = 4 * n4 + t0; j < n; j += stride) {
    const unsigned b = __builtin_bit_cast(unsigned, x[j]) & 0x7FFFFFFFu;
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned t = (unsigned)__shfl_xor((int)m, o, 64);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = red[0] > red[1] ? red[0] : red[1], b = red[2] > red[3] ? red[2] : red[3];
    m = a > b ? a : b;
    if (m != 0) atomicMax(scratch, m);
    __threadfence();                                   // the maximum is out before the ticket is
    const unsigned ticket = atomicAdd(scratch + 1, 1u);
    if (ticket == gridDim.x - 1) {                     // every other workgroup's maximum has landed
      __threadfence();
      const unsigned final_m = atomicExch(scratch, 0u);  // read memory-side, and leave the word zero
      atomicExch(scratch + 1, 0u);
      pow2_scale_of(final_m, scale_out);
    }
  }
}
extern "C" int i2r_absmax_pow2_scale_f32(const float* x, long n, unsigned* scratch, float* scale_out,
                                         i2r_stream_t stream) {
  I2R_REQUIRE(x && scratch && scale_out && n > 0, "absmax_pow2_scale: bad argument");
  long nb = (n / 4 + 256 * 8 - 1) / (256 * 8);  // >= 8 16-byte loads per thread before another workgroup is worth it
  if (nb < 1) nb = 1;
  if (nb > ABSMAX_MAX_WG) nb = ABSMAX_MAX_WG;
  hipLaunchKernelGGL(absmax_pow2_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, n, scratch,
                     scale_out);
  return i2r_check_launch("absmax_pow2_scale");
}
