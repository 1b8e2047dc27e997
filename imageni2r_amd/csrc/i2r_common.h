// Shared device/host helpers for libi2r_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define I2R_OK 0
#define I2R_ERR_ARG -1
#define I2R_ERR_HIP -2
#define I2R_ERR_UNSUPPORTED -3

void i2r_set_error(const char* fmt, ...);
int i2r_check_launch(const char* what);

#define I2R_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      i2r_set_error(__VA_ARGS__);         \
      return I2R_ERR_ARG;                 \
    }                                     \
  } while (0)

static inline int i2r_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Raises a kernel's dynamic-LDS limit once per (kernel, DEVICE): the attribute belongs to the device's copy of the
// code object, so a process that launches on a second GPU has to set it there too.  `done`: the call site's static
// per-device flags.  False (and the error text set) when the runtime refuses.
static inline bool i2r_lds_attr(const void* kern, int bytes, bool (&done)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (done[dev]) return true;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    i2r_set_error("cannot raise the dynamic LDS limit of a kernel to %d bytes on device %d: %s", bytes, dev,
                  hipGetErrorString(e));
    return false;
  }
  done[dev] = true;
  return true;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// wave64 all-reduce helpers (CDNA wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + expf(-u)); }
// Same value to ~2 ulp on the hardware transcendental units (v_exp_f32, v_rcp_f32: 1 ulp each) instead of the
// ~20-instruction IEEE expf + division: the fused GroupNorm+SiLU kernel is VALU-bound, not HBM-bound, without it.
__device__ __forceinline__ float silu_fast(float u) {
  float e = __builtin_amdgcn_exp2f(u * -1.4426950408889634f);  // exp(-u); +inf for very negative u -> rcp = 0
  return u * __builtin_amdgcn_rcpf(1.0f + e);
}
// d silu(u) / du
__device__ __forceinline__ float dsilu_f(float u) {
  float s = 1.0f / (1.0f + expf(-u));
  return s * (1.0f + u * (1.0f - s));
}

// x = hi + lo with hi = f16(x), lo = f16(x - hi): the operand split of the f16x2 kernels.  The value is made opaque
// first: when x is the result of a multiply, hipcc folds f16(a*b) into v_fma_mixlo_f16 (ONE rounding of the exact
// product) for the stored hi but may convert the separately rounded f32 product for the subtraction -- at an exact
// f16 tie the two disagree by one f16 ulp and hi + lo is off by 2^-11 |x| (seen in the attention Q scaling).
__device__ __forceinline__ void f16_split(float x, _Float16& hi, _Float16& lo) {
  asm volatile("" : "+v"(x));
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// y -> f16 hi | f16 lo << 16: the pre-split operand word of the f16x2 kernels, carried in a float register
__device__ __forceinline__ float split_pack(float y) {
  _Float16 hi, lo;
  f16_split(y, hi, lo);
  unsigned w = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
  return __builtin_bit_cast(float, w);
}
// softmax scale of the attention logits folded into q: 1/sqrt(64) * log2(e)  (the kernels run the softmax on exp2)
#define I2R_QSCALE 0.18033688011112042f

// Philox4x32-10 counter RNG (used for dropout keep-masks and training noise; regenerated, never stored)
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}
__device__ __forceinline__ float u32_to_unit(uint32_t x) {  // (0,1]
  return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
