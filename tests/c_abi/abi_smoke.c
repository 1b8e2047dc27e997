/* A host program in plain C on the C ABI of libi2r_hip.so (include/i2r_hip.h): no Python, no torch -- device memory
 * from the HIP runtime, plain pointers and sizes in, status codes out.  What a non-Python host (or another language's
 * FFI) binds.  Checks, against loops written here:
 *   1. i2r_ts_to_img_f32 / i2r_img_to_ts_f32 on a series with missing steps: bit-exact image, mask and round trip;
 *   2. i2r_conv2d_pack_weight_h2 + i2r_conv2d_fwd_h2 (3x3, bias + residual) against a double-precision direct
 *      convolution: 1e-5;  the same through i2r_conv2d_fwd_h2_noalloc: equal bits;
 *   3. the error convention: a bad argument returns a negative code and leaves a message in i2r_last_error().
 * Built by __graft_entry__.build() (tests/c_abi/Makefile) as a shared object and as a program; run by
 * tests/test_gpu_c_abi.py.  Test infrastructure. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "i2r_hip.h"

#define CHECK_HIP(e)                                                                  \
  do {                                                                                \
    hipError_t r_ = (e);                                                              \
    if (r_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)r_, __LINE__); return 2; } \
  } while (0)
#define CHECK_I2R(e)                                                                              \
  do {                                                                                            \
    int r_ = (e);                                                                                 \
    if (r_ != 0) { printf("i2r error %d at line %d: %s\n", r_, __LINE__, i2r_last_error()); return 3; } \
  } while (0)

static unsigned lcg = 12345u;
static float frand(void) { lcg = lcg * 1664525u + 1013904223u; return (float)(lcg >> 8) / 16777216.0f; }

/* Exported for tests/test_gpu_c_abi.py (loaded in-process: the GPU boxes refuse an exec from a process that has
 * initialised the GPU); `main` below makes the same file a stand-alone program. */
int abi_smoke_run(void) {
  printf("i2r_version %d\n", i2r_version());
  hipStream_t stream;
  CHECK_HIP(hipStreamCreate(&stream));

  /* ---- 1. delay embedding: L = 24, delay 3, embedding 8 -> 7 columns, padded to 8 x 8 */
  enum { B = 2, L = 24, C = 3, D = 3, E = 8, COLS = 7, S = 8 };
  float ts[B * L * C], img[B * C * S * S], mask[B * C * S * S], back[B * L * C];
  for (int i = 0; i < B * L * C; ++i) ts[i] = frand();
  for (int c = 0; c < C; ++c) { ts[(0 * L + 5) * C + c] = NAN; ts[(1 * L + 17) * C + c] = NAN; }
  float *d_ts, *d_img, *d_mask, *d_back;
  CHECK_HIP(hipMalloc((void**)&d_ts, sizeof ts));
  CHECK_HIP(hipMalloc((void**)&d_img, sizeof img));
  CHECK_HIP(hipMalloc((void**)&d_mask, sizeof mask));
  CHECK_HIP(hipMalloc((void**)&d_back, sizeof back));
  CHECK_HIP(hipMemcpy(d_ts, ts, sizeof ts, hipMemcpyHostToDevice));
  CHECK_I2R(i2r_ts_to_img_f32(d_ts, d_img, d_mask, B, L, C, D, E, COLS, S, 0.0f, stream));
  CHECK_I2R(i2r_img_to_ts_f32(d_img, d_back, B, L, C, D, E, COLS, S, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(img, d_img, sizeof img, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(mask, d_mask, sizeof mask, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(back, d_back, sizeof back, hipMemcpyDeviceToHost));
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int r = 0; r < S; ++r)
        for (int j = 0; j < S; ++j) {
          int t = j * D + r;
          float want = (r < E && j < COLS && t < L) ? ts[(b * L + t) * C + c] : 0.0f;
          float got = img[((b * C + c) * S + r) * S + j], m = mask[((b * C + c) * S + r) * S + j];
          if (memcmp(&want, &got, 4) != 0 && !(isnan(want) && isnan(got))) { printf("ts_to_img mismatch\n"); return 1; }
          if (m != (isnan(want) ? 0.0f : 1.0f)) { printf("mask mismatch\n"); return 1; }
        }
  for (int i = 0; i < B * L * C; ++i) {  /* every step sits in <= 3 cells holding the same value: the mean is it, to 1 ulp */
    if (isnan(ts[i]) ? !isnan(back[i]) : fabsf(back[i] - ts[i]) > 1.2e-7f) { printf("round trip mismatch at %d\n", i); return 1; }
  }
  printf("delay embedding ok\n");

  /* ---- 2. 3x3 convolution with bias and residual on the f16x2 kernels */
  enum { CB = 3, CIN = 20, COUT = 36, H = 8, W = 8 };
  const int nx = CB * CIN * H * W, ny = CB * COUT * H * W, nw = COUT * CIN * 9;
  float *x = malloc(nx * 4), *w = malloc(nw * 4), *bias = malloc(COUT * 4), *res = malloc(ny * 4), *y = malloc(ny * 4),
        *y2 = malloc(ny * 4);
  for (int i = 0; i < nx; ++i) x[i] = 2.0f * frand() - 1.0f;
  for (int i = 0; i < nw; ++i) w[i] = (2.0f * frand() - 1.0f) / sqrtf(9.0f * CIN);
  for (int i = 0; i < COUT; ++i) bias[i] = frand();
  for (int i = 0; i < ny; ++i) res[i] = frand();
  float *d_x, *d_w, *d_b, *d_r, *d_y, *d_ws = NULL;
  void* d_wp;
  long wp_bytes = i2r_conv2d_packed_h2_bytes(COUT, CIN, 3), nsk = i2r_conv2d_splitk_h2_floats(CB, CIN, H, W, COUT, 3);
  CHECK_HIP(hipMalloc((void**)&d_x, nx * 4));
  CHECK_HIP(hipMalloc((void**)&d_w, nw * 4));
  CHECK_HIP(hipMalloc((void**)&d_b, COUT * 4));
  CHECK_HIP(hipMalloc((void**)&d_r, ny * 4));
  CHECK_HIP(hipMalloc((void**)&d_y, ny * 4));
  CHECK_HIP(hipMalloc(&d_wp, wp_bytes));
  if (nsk > 0) CHECK_HIP(hipMalloc((void**)&d_ws, nsk * 4));
  CHECK_HIP(hipMemcpy(d_x, x, nx * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_w, w, nw * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_b, bias, COUT * 4, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_r, res, ny * 4, hipMemcpyHostToDevice));
  CHECK_I2R(i2r_conv2d_pack_weight_h2(d_w, d_wp, COUT, CIN, 3, stream));
  CHECK_I2R(i2r_conv2d_fwd_h2(d_x, d_wp, d_b, d_r, d_y, d_ws, nsk, CB, CIN, H, W, COUT, 3, 0, stream));
  CHECK_HIP(hipStreamSynchronize(stream));
  CHECK_HIP(hipMemcpy(y, d_y, ny * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, scale = 0.0;
  for (int b = 0; b < CB; ++b)
    for (int co = 0; co < COUT; ++co)
      for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx) {
          double acc = bias[co] + res[((b * COUT + co) * H + yy) * W + xx];
          for (int ci = 0; ci < CIN; ++ci)
            for (int r = 0; r < 3; ++r)
              for (int s = 0; s < 3; ++s) {
                int iy = yy + r - 1, ix = xx + s - 1;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                  acc += (double)w[((co * CIN + ci) * 3 + r) * 3 + s] * x[((b * CIN + ci) * H + iy) * W + ix];
              }
          double e = fabs(acc - y[((b * COUT + co) * H + yy) * W + xx]);
          if (e > worst) worst = e;
          if (fabs(acc) > scale) scale = fabs(acc);
        }
  printf("conv3x3 f16x2: max error %.2e of %.2f\n", worst, scale);
  if (!(worst <= 1e-5 * scale)) { printf("conv mismatch\n"); return 1; }
  if (nsk <= i2r_conv2d_fwd_h2_noalloc_max_floats()) {
    CHECK_HIP(hipMemset(d_y, 0, ny * 4));
    CHECK_I2R(i2r_conv2d_fwd_h2_noalloc(d_x, d_wp, d_b, d_r, d_y, CB, CIN, H, W, COUT, 3, 0, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipMemcpy(y2, d_y, ny * 4, hipMemcpyDeviceToHost));
    if (memcmp(y, y2, ny * 4) != 0) { printf("no-alloc entry point differs\n"); return 1; }
    printf("no-alloc entry point: equal bits\n");
  }

  /* ---- 3. error convention */
  int rc = i2r_conv2d_fwd_h2(NULL, d_wp, d_b, d_r, d_y, d_ws, nsk, CB, CIN, H, W, COUT, 3, 0, stream);
  if (rc >= 0 || strlen(i2r_last_error()) == 0) { printf("null pointer was not refused (rc %d)\n", rc); return 1; }
  rc = i2r_ts_to_img_f32(d_ts, d_img, d_mask, B, L, C, D, E, COLS + 1, S, 0.0f, stream);
  if (rc >= 0) { printf("inconsistent geometry was not refused\n"); return 1; }
  printf("errors refused: \"%s\"\n", i2r_last_error());
  printf("C ABI smoke ok\n");
  return 0;
}

int main(void) { return abi_smoke_run(); }
