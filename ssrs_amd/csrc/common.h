// Shared device/host helpers for libmfnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define MFNET_OK 0
#define MFNET_ERR_ARG (-1)
#define MFNET_ERR_HIP (-2)
#define MFNET_ERR_UNSUPPORTED (-3)

// dtype enum shared with include/mfnet_hip.h
#define MFNET_F32 0
#define MFNET_BF16 1

void mfnet_set_error(const char* fmt, ...);

#define MF_CHECK_ARG(cond, ...)                  \
  do {                                           \
    if (!(cond)) {                               \
      mfnet_set_error(__VA_ARGS__);              \
      return MFNET_ERR_ARG;                      \
    }                                            \
  } while (0)

#define MF_LAUNCH_CHECK(name)                                                     \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      mfnet_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return MFNET_ERR_HIP;                                                       \
    }                                                                             \
  } while (0)

typedef unsigned short bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (8 bf16)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;                 // result of ds_read_b64_tr_b16
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned int)v) << 16);
}
// round-to-nearest-even by the gfx950 conversion instruction (v_cvt_pk_bf16_f32): one VALU op per pair
// instead of a dozen with a NaN branch
typedef __attribute__((ext_vector_type(2))) float mf_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 mf_bf16x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const mf_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, mf_bf16x2_t));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact (erf) GELU and its derivative, as nn.GELU() default
// Exact (erf) GELU of nn.GELU, with erfc from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32
// round-off level): one v_rcp + one v_exp + 7 FMAs instead of the branchy library erff - the GELU epilogues
// of the MLP GEMMs evaluate it 64 times per lane. cdf(x) = 1 - erfc(x/sqrt2)/2 for x >= 0, erfc(|x|/sqrt2)/2
// below (no cancellation in the left tail).
__device__ __forceinline__ float gelu_cdf_pdf(float x, float* pdf) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float pl = fmaf(t, 1.061405429f, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = __expf(-z * z);           // = exp(-x^2/2)
  const float hq = 0.5f * pl * t * e;       // erfc(z) / 2
  *pdf = 0.3989422804014327f * e;
  return x < 0.f ? hq : 1.0f - hq;
}
__device__ __forceinline__ float gelu_f(float x) { float pdf; return x * gelu_cdf_pdf(x, &pdf); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  float pdf;
  const float cdf = gelu_cdf_pdf(x, &pdf);
  return fmaf(x, pdf, cdf);
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
