// Fused optimiser tail of the diffusion train step (run_irregular.py:228-230):
//     clip_grad_norm_(params, max_norm)  ->  AdamW.step()  ->  LitEma update (models/ema.py:26-47)
// as three launches over ALL parameter tensors at once (multi-tensor: a device table of pointers + a chunk table),
// instead of ~15 foreach passes plus a per-parameter EMA loop.  HBM-bound: 16 B read + 12 B written per parameter
// for AdamW, +8/+4 for the EMA shadow.  Every per-step scalar (step count, bias corrections, clip coefficient, EMA
// decay, num_updates) lives in device memory and is advanced by the kernels themselves, so the whole tail is safe
// to capture into a HIP graph and replay.
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

#define OPT_CHUNK 8192  // elements per workgroup (256 threads x 8 float4)

struct OptEntry {  // one parameter tensor; mirrors the (T, 6) int64 table built by the host
  float* p;
  const float* g;
  float* m;
  float* v;
  float* ema;  // may be null
  long n;
};

// scalars[]: 0 total_norm, 1 clip_coef, 2 step_size (lr / bias_correction1), 3 sqrt(bias_correction2),
//            4 ema (1 - decay), 5 ema mode (0 off, 1 copy, 2 lerp), 6 steps skipped so far, 7 skip THIS step (0/1)
// Guard (only when clipping is on, i.e. the norm is computed anyway): a step whose total gradient norm is not finite
// is SKIPPED -- no weight, moment, EMA or counter changes -- and counted in scalars[6].  torch's clip_grad_norm_ +
// AdamW would turn every weight into NaN there (the reference has no guard, run_irregular.py:228-229); with finite
// gradients the two are identical.
__global__ __launch_bounds__(256) void opt_sumsq_kernel(const OptEntry* __restrict__ tab,
                                                        const int2* __restrict__ chunks,
                                                        float* __restrict__ partials) {
  const int2 ch = chunks[blockIdx.x];
  const OptEntry e = tab[ch.x];
  const long start = (long)ch.y * OPT_CHUNK;
  long cnt = e.n - start;
  if (cnt > OPT_CHUNK) cnt = OPT_CHUNK;
  const float* g = e.g + start;
  float acc = 0.0f;
  if ((((uintptr_t)g) & 15) == 0) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = threadIdx.x; i < (cnt >> 2); i += 256) {
      float4 t = g4[i];
      acc += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
    }
    for (long i = (cnt & ~3L) + threadIdx.x; i < cnt; i += 256) acc += g[i] * g[i];
  } else {
    for (long i = threadIdx.x; i < cnt; i += 256) acc += g[i] * g[i];
  }
  acc = wave_sum(acc);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup: total norm (fp64 tree over the partials, fixed order), then every per-step scalar.
__global__ __launch_bounds__(256) void opt_scalars_kernel(const float* __restrict__ partials, int nparts,
                                                          float* __restrict__ scalars, float* __restrict__ step,
                                                          int* __restrict__ ema_num_updates, float max_norm, float lr,
                                                          float beta1, float beta2, float ema_decay, int ema_warmup,
                                                          int ema_on) {
  __shared__ double red[256];
  double acc = 0.0;
  if (max_norm > 0.0f)
    for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)partials[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float total = (float)sqrt(red[0]);
    float coef = 1.0f;
    if (max_norm > 0.0f) {
      coef = max_norm / (total + 1e-6f);  // torch.nn.utils.clip_grad_norm_
      if (coef > 1.0f) coef = 1.0f;
    }
    const bool skip = max_norm > 0.0f && !(total <= 3.0e38f);  // inf or NaN gradient norm
    scalars[0] = total;
    scalars[7] = skip ? 1.0f : 0.0f;
    if (skip) {
      scalars[6] += 1.0f;
      return;
    }
    float t = step[0] + 1.0f;
    step[0] = t;
    double bc1 = 1.0 - pow((double)beta1, (double)t);
    double bc2 = 1.0 - pow((double)beta2, (double)t);
    scalars[0] = total;
    scalars[1] = coef;
    scalars[2] = (float)((double)lr / bc1);
    scalars[3] = (float)sqrt(bc2);
    float omd = 0.0f, mode = 0.0f;
    if (ema_on) {
      float decay = ema_decay;
      int n = ema_num_updates[0];
      if (n >= 0) {  // use_num_updates (ema.py:29-32)
        n += 1;
        ema_num_updates[0] = n;
        float d2 = (1.0f + (float)n) / (10.0f + (float)n);
        if (d2 < decay) decay = d2;
      }
      omd = 1.0f - decay;
      mode = (n > ema_warmup) ? 2.0f : 1.0f;  // copy while num_updates <= warmup (ema.py:37-47)
    }
    scalars[4] = omd;
    scalars[5] = mode;
  }
}

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float* ema, float coef, float decay_mul,
                                          float b1, float b2, float step_size, float sqrt_bc2, float eps,
                                          float omd, int mode) {
  g *= coef;
  p *= decay_mul;                        // p.mul_(1 - lr * weight_decay)
  m = m + (g - m) * (1.0f - b1);         // exp_avg.lerp_(grad, 1 - beta1)
  v = v * b2 + (1.0f - b2) * (g * g);    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  float denom = sqrtf(v) / sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
  if (mode == 1) *ema = p;
  else if (mode == 2) *ema = *ema - omd * (*ema - p);
}

__global__ __launch_bounds__(256) void opt_adamw_ema_kernel(const OptEntry* __restrict__ tab,
                                                            const int2* __restrict__ chunks,
                                                            const float* __restrict__ scalars, float lr, float beta1,
                                                            float beta2, float eps, float weight_decay) {
  const int2 ch = chunks[blockIdx.x];
  const OptEntry e = tab[ch.x];
  const long start = (long)ch.y * OPT_CHUNK;
  long cnt = e.n - start;
  if (cnt > OPT_CHUNK) cnt = OPT_CHUNK;
  if (scalars[7] != 0.0f) return;  // non-finite gradient norm: the whole step is skipped (uniform over the grid)
  const float coef = scalars[1], step_size = scalars[2], isb2 = scalars[3], omd = scalars[4];
  const int mode = e.ema ? (int)scalars[5] : 0;
  const float decay_mul = 1.0f - lr * weight_decay;
  float* p = e.p + start;
  const float* g = e.g + start;
  float* m = e.m + start;
  float* v = e.v + start;
  float* s = e.ema ? e.ema + start : nullptr;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)s)) & 15) == 0;
  long done = 0;
  if (vec) {
    for (long i = threadIdx.x; i < (cnt >> 2); i += 256) {
      float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
      float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
      float4 S = make_float4(0.f, 0.f, 0.f, 0.f);
      if (mode == 2) S = reinterpret_cast<float4*>(s)[i];
      adamw_one(P.x, G.x, M.x, V.x, &S.x, coef, decay_mul, beta1, beta2, step_size, isb2, eps, omd, mode);
      adamw_one(P.y, G.y, M.y, V.y, &S.y, coef, decay_mul, beta1, beta2, step_size, isb2, eps, omd, mode);
      adamw_one(P.z, G.z, M.z, V.z, &S.z, coef, decay_mul, beta1, beta2, step_size, isb2, eps, omd, mode);
      adamw_one(P.w, G.w, M.w, V.w, &S.w, coef, decay_mul, beta1, beta2, step_size, isb2, eps, omd, mode);
      reinterpret_cast<float4*>(p)[i] = P;
      reinterpret_cast<float4*>(m)[i] = M;
      reinterpret_cast<float4*>(v)[i] = V;
      if (mode) reinterpret_cast<float4*>(s)[i] = S;
    }
    done = cnt & ~3L;
  }
  for (long i = done + threadIdx.x; i < cnt; i += 256) {
    float P = p[i], M = m[i], V = v[i], S = (mode == 2) ? s[i] : 0.0f;
    adamw_one(P, g[i], M, V, &S, coef, decay_mul, beta1, beta2, step_size, isb2, eps, omd, mode);
    p[i] = P;
    m[i] = M;
    v[i] = V;
    if (mode) s[i] = S;
  }
}

extern "C" int i2r_opt_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int i2r_fused_adamw_ema_step(const void* table, const void* chunks, int nchunks, float* partials,
                                        float* scalars, float* step, int* ema_num_updates, float max_norm, float lr,
                                        float beta1, float beta2, float eps, float weight_decay, float ema_decay,
                                        int ema_warmup, i2r_stream_t stream) {
  I2R_REQUIRE(table && chunks && partials && scalars && step && nchunks > 0, "fused_adamw_ema_step: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (max_norm > 0.0f)
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3(nchunks), dim3(256), 0, s, (const OptEntry*)table, (const int2*)chunks,
                       partials);
  hipLaunchKernelGGL(opt_scalars_kernel, dim3(1), dim3(256), 0, s, partials, nchunks, scalars, step, ema_num_updates,
                     max_norm, lr, beta1, beta2, ema_decay, ema_warmup, ema_num_updates != nullptr);
  hipLaunchKernelGGL(opt_adamw_ema_kernel, dim3(nchunks), dim3(256), 0, s, (const OptEntry*)table,
                     (const int2*)chunks, scalars, lr, beta1, beta2, eps, weight_decay);
  return i2r_check_launch("fused_adamw_ema_step");
}
