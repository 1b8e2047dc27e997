// Ceiling probe for the f16x2-split 3x3 conv: the kernel's MFMA pattern -- per wave a 128-channel x
// 32-position output tile, every product as hi*hi + hi*lo + lo*hi -- as a BARE loop on random operands held in
// registers, once with v_mfma_f32_32x32x16_f16 (what conv_fwd_h2_kernel issues) and once with v_mfma_f32_16x16x32_f16.
// Same flops, same operand registers; the question is only which shape the chip clocks higher under load
// (MI355X_MICROARCH.md, "Shape").  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// operands: per lane 20 half8 (16 A + 4 B) read once from `src`
template <int SHAPE>
__global__ __launch_bounds__(256, 2) void probe(const half8* __restrict__ src, float* __restrict__ dst, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  half8 a[16], b[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = src[(size_t)(tid % 4096) * 20 + i];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = src[(size_t)(tid % 4096) * 20 + 16 + i];
  float out = 0.f;
  if (SHAPE == 32) {
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)  // two k16 steps = one k32
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          // a[ks*8 + 2t] = hi rows of tile t, a[ks*8 + 2t + 1] = lo rows; b[2ks] = hi, b[2ks+1] = lo
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks * 8 + 2 * t], b[2 * ks], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks * 8 + 2 * t], b[2 * ks + 1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks * 8 + 2 * t + 1], b[2 * ks], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) out += acc[t][r];
  } else {
    f32x4 acc[8][2];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t)     // a[2t] = hi rows of 16-row tile t (k32), a[2t+1] = lo; b[2c] = hi of col tile c
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * t], b[2 * c], acc[t][c], 0, 0, 0);
          acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * t], b[2 * c + 1], acc[t][c], 0, 0, 0);
          acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * t + 1], b[2 * c], acc[t][c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) out += acc[t][c][r];
  }
  if (out == 123.456f) dst[tid] = out;  // keep the loop alive
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const size_t n = 4096 * 20;
  std::vector<half8> h(n);
  for (int zero = 0; zero < 2; ++zero) {
    srand(1);
    for (size_t i = 0; i < n; ++i)
      for (int j = 0; j < 8; ++j) h[i][j] = zero ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    half8* src; float* dst;
    CK(hipMalloc(&src, n * sizeof(half8)));
    CK(hipMalloc(&dst, 1 << 22));
    CK(hipMemcpy(src, h.data(), n * sizeof(half8), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
      for (int shape : {32, 16, 32, 16}) {
        const int grid = 256 * wgs_per_cu;
        for (int rep = 0; rep < 2; ++rep) {  // rep 0 warms up
          CK(hipEventRecord(e0));
          if (shape == 32) hipLaunchKernelGGL(probe<32>, dim3(grid), dim3(256), 0, 0, src, dst, iters);
          else hipLaunchKernelGGL(probe<16>, dim3(grid), dim3(256), 0, 0, src, dst, iters);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          // per wave and iteration: 128 x 32 x 32 (k32) x 2 flops x 3 terms
          const double mfma_flop = (double)grid * 4 * iters * 128.0 * 32 * 32 * 2 * 3;
          if (rep) printf("%s data, %d wave(s)/SIMD, mfma %s: %.2f ms  %.0f TF f16-MFMA  (%.0f TF algorithmic)\n",
                          zero ? "zero  " : "random", wgs_per_cu, shape == 32 ? "32x32x16" : "16x16x32", ms,
                          mfma_flop / ms * 1e-9, mfma_flop / 3 / ms * 1e-9);
        }
      }
    CK(hipFree(src)); CK(hipFree(dst));
  }
  return 0;
}
