// Delay-embedding transforms: time series (B,L,C) <-> square image (B,C,S,S).
// Replaces the reference's Python slice-copy loops (models/img_transformations.py:79-108, 140-170).
// HBM-bound byte movers: one read of the series, one coalesced write of image (+mask); no LDS needed because
// every image cell maps to exactly one series element (closed form t = col*delay + row).
#include "i2r_common.h"
#include "../../include/i2r_hip.h"

// One thread per image cell, x fastest (coalesced 4-B stores along the image row; the matching loads walk the
// series with stride delay*C floats, served from L2: the series is 1/k the size of the image).
__global__ void ts_to_img_kernel(const float* __restrict__ ts, float* __restrict__ img, float* __restrict__ mask,
                                 int B, int L, int C, int delay, int E, int cols, int S, float pad_value) {
  long total = (long)B * C * S * S;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int col = (int)(i % S);
    long r1 = i / S;
    int row = (int)(r1 % S);
    long r2 = r1 / S;
    int c = (int)(r2 % C);
    int b = (int)(r2 / C);
    float v;
    if (row < E && col < cols) {
      int t = col * delay + row;
      v = (t < L) ? ts[((long)b * L + t) * C + c] : 0.0f;  // tail of the partial column stays 0 (:86,:98)
    } else {
      v = pad_value;  // pad_to_square (:59-70)
    }
    img[i] = v;
    if (mask) mask[i] = (v != v) ? 0.0f : 1.0f;  // run_irregular.py:216
  }
}

// d ts[b,t,c] = sum over the cells that hold t of d img (the adjoint of the gather above).
__global__ void ts_to_img_bwd_kernel(const float* __restrict__ dimg, float* __restrict__ dts, int B, int L, int C,
                                     int delay, int E, int cols, int S) {
  long total = (long)B * L * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long r1 = i / C;
    int t = (int)(r1 % L);
    int b = (int)(r1 / L);
    const float* p = dimg + ((long)b * C + c) * S * S;
    float acc = 0.0f;
    for (int row = t % delay; row < E && row <= t; row += delay) {
      int col = (t - row) / delay;
      if (col < cols) acc += p[row * S + col];
    }
    dts[i] = acc;
  }
}

// ts[b,t,c] = (sum of the k cells that hold t, ascending image row, in T) / k, stored as fp32
// (img_transformations.py:152-165; order and true division established in SURVEY.md §0.7).
template <typename T>
__global__ void img_to_ts_kernel(const T* __restrict__ img, float* __restrict__ ts, int B, int L, int C, int delay,
                                 int E, int cols, int S) {
  long total = (long)B * L * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // t fastest inside a (b,c) plane so that neighbouring lanes read neighbouring image cells
    int t = (int)(i % L);
    long r1 = i / L;
    int c = (int)(r1 % C);
    int b = (int)(r1 / C);
    const T* p = img + ((long)b * C + c) * S * S;
    T acc = (T)0;
    int k = 0;
    for (int row = t % delay; row < E && row <= t; row += delay) {
      int col = (t - row) / delay;
      if (col < cols) {
        T v = p[row * S + col];
        acc = (k == 0) ? v : acc + v;
        ++k;
      }
    }
    float o = 0.0f;
    if (k > 0) o = (float)(acc / (T)k);
    ts[((long)b * L + t) * C + c] = o;
  }
}

static int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

static int check_geom(int B, int L, int C, int delay, int E, int cols, int S) {
  I2R_REQUIRE(B > 0 && L > 0 && C > 0 && delay > 0 && E > 0, "delay-embed: non-positive dimension");
  int n_full = L < E ? 0 : (L - E) / delay + 1;
  int want = n_full + ((n_full * delay != L) ? 1 : 0);
  I2R_REQUIRE(cols == want, "delay-embed: cols=%d does not match geometry (expected %d)", cols, want);
  I2R_REQUIRE(cols <= E, "delay-embed: cols=%d > embedding=%d (reference allocates an ExE image)", cols, E);
  I2R_REQUIRE(S >= E && S >= cols, "delay-embed: side %d smaller than image %dx%d", S, E, cols);
  return I2R_OK;
}

extern "C" int i2r_ts_to_img_f32(const float* ts, float* img, float* mask, int B, int L, int C, int delay, int E,
                                 int cols, int S, float pad_value, i2r_stream_t stream) {
  I2R_REQUIRE(ts && img, "ts_to_img: null pointer");
  if (int e = check_geom(B, L, C, delay, E, cols, S)) return e;
  long total = (long)B * C * S * S;
  hipLaunchKernelGGL(ts_to_img_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, ts, img, mask,
                     B, L, C, delay, E, cols, S, pad_value);
  return i2r_check_launch("ts_to_img");
}

extern "C" int i2r_ts_to_img_bwd_f32(const float* dimg, float* dts, int B, int L, int C, int delay, int E, int cols,
                                     int S, i2r_stream_t stream) {
  I2R_REQUIRE(dimg && dts, "ts_to_img_bwd: null pointer");
  if (int e = check_geom(B, L, C, delay, E, cols, S)) return e;
  long total = (long)B * L * C;
  hipLaunchKernelGGL(ts_to_img_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dimg, dts,
                     B, L, C, delay, E, cols, S);
  return i2r_check_launch("ts_to_img_bwd");
}

extern "C" int i2r_img_to_ts_f32(const float* img, float* ts, int B, int L, int C, int delay, int E, int cols, int S,
                                 i2r_stream_t stream) {
  I2R_REQUIRE(img && ts, "img_to_ts: null pointer");
  if (int e = check_geom(B, L, C, delay, E, cols, S)) return e;
  long total = (long)B * L * C;
  hipLaunchKernelGGL(img_to_ts_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, ts,
                     B, L, C, delay, E, cols, S);
  return i2r_check_launch("img_to_ts_f32");
}

extern "C" int i2r_img_to_ts_f64(const double* img, float* ts, int B, int L, int C, int delay, int E, int cols, int S,
                                 i2r_stream_t stream) {
  I2R_REQUIRE(img && ts, "img_to_ts: null pointer");
  if (int e = check_geom(B, L, C, delay, E, cols, S)) return e;
  long total = (long)B * L * C;
  hipLaunchKernelGGL(img_to_ts_kernel<double>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, img, ts,
                     B, L, C, delay, E, cols, S);
  return i2r_check_launch("img_to_ts_f64");
}
