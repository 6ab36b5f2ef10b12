// LayerNorm over the last (channel) dimension of a row-major [rows, C] token matrix.
// HBM-bound: one 64-lane wave per row, the row lives in registers (float4 per lane per step),
// two-pass mean / variance with wave shuffles. Algorithmic bytes: read x + write y per call.
//
// Reference semantics: nn.LayerNorm(eps=1e-6) in the ViT blocks (medsam/models/sam/build_sam.py:80),
// LayerNorm2d on NCHW (medsam/models/common/layer_norm.py:12-17) and Norm2d
// (unetformer_mmsam.py:14-22) - all the same op once tokens are channels-last.
#include "common.h"

#define LN_MAXV 8   // float4 chunks per lane -> C <= 64*4*8 = 2048

template <typename TIn>
__device__ __forceinline__ float4 ld4(const TIn* p);
template <>
__device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p) {
  u32x2_t w = *reinterpret_cast<const u32x2_t*>(p);
  return make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
}
template <typename TOut>
__device__ __forceinline__ void st4(TOut* p, float4 v);
template <>
__device__ __forceinline__ void st4<float>(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <>
__device__ __forceinline__ void st4<bf16_t>(bf16_t* p, float4 v) {
  u32x2_t w;
  w.x = pack_bf16x2(v.x, v.y);
  w.y = pack_bf16x2(v.z, v.w);
  *reinterpret_cast<u32x2_t*>(p) = w;
}

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TIn* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TOut* __restrict__ y,
                                                     int64_t ldy, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TIn* xr = x + row * ldx;
  const int nv = C >> 2;
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      v[i] = ld4<TIn>(xr + 4 * c4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    if (mean) mean[row] = mu;
    if (rstd) rstd[row] = rs;
  }
  TOut* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c4);
      const float4 b = *reinterpret_cast<const float4*>(beta + 4 * c4);
      float4 o;
      o.x = (v[i].x - mu) * rs * g.x + b.x;
      o.y = (v[i].y - mu) * rs * g.y + b.y;
      o.z = (v[i].z - mu) * rs * g.z + b.z;
      o.w = (v[i].w - mu) * rs * g.w + b.w;
      st4<TOut>(yr + 4 * c4, o);
    }
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)) [+ dres]; optional bf16 copy of dx;
// optional dgamma/dbeta accumulation (float atomics, one add per column per wave).
// NVL = float4 chunks per lane (C <= 256 * NVL): the per-row register arrays are sized for the row at hand, which keeps
// enough waves resident for the loads in flight to cover HBM latency (sized for C = 2048 the kernel ran 2 waves / SIMD).
template <typename TDy, typename TX, int NVL, bool WANT_W>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const TDy* __restrict__ dy, int64_t lddy,
                                                     const TX* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ mean,
                                                     const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma,
                                                     const float* dres, float* dx,
                                                     bf16_t* __restrict__ dx_bf16, int64_t lddx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     int64_t rows, int C) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nv = C >> 2;
  float4 ag[NVL], ab[NVL];
  constexpr bool want_w = WANT_W;
  if (want_w) {
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      ag[i] = make_float4(0, 0, 0, 0);
      ab[i] = make_float4(0, 0, 0, 0);
    }
  }
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float4 xh[NVL], gd[NVL], rr[NVL];
    float s1 = 0.f, s2 = 0.f;
    if (dres) {        // requested with x and dy: the row costs one memory round trip, not two
#pragma unroll
      for (int i = 0; i < NVL; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nv) rr[i] = *reinterpret_cast<const float4*>(dres + row * lddx + 4 * c4);
      }
    }
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        const float4 xv = ld4<TX>(x + row * ldx + 4 * c4);
        const float4 d = ld4<TDy>(dy + row * lddy + 4 * c4);
        const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c4);
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        gd[i] = make_float4(g.x * d.x, g.y * d.y, g.z * d.z, g.w * d.w);
        s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
        s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
        if (want_w) {
          ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
          ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        }
      }
    }
    const float m1 = wave_sum(s1) / (float)C;
    const float m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        float4 o;
        o.x = rs * (gd[i].x - m1 - xh[i].x * m2);
        o.y = rs * (gd[i].y - m1 - xh[i].y * m2);
        o.z = rs * (gd[i].z - m1 - xh[i].z * m2);
        o.w = rs * (gd[i].w - m1 - xh[i].w * m2);
        if (dres) { o.x += rr[i].x; o.y += rr[i].y; o.z += rr[i].z; o.w += rr[i].w; }
        if (dx) *reinterpret_cast<float4*>(dx + row * lddx + 4 * c4) = o;
        if (dx_bf16) st4<bf16_t>(dx_bf16 + row * lddx + 4 * c4, o);
      }
    }
  }
  if (want_w) {
#pragma unroll
    for (int i = 0; i < NVL; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        atomicAdd(dgamma + 4 * c4 + 0, ag[i].x); atomicAdd(dgamma + 4 * c4 + 1, ag[i].y);
        atomicAdd(dgamma + 4 * c4 + 2, ag[i].z); atomicAdd(dgamma + 4 * c4 + 3, ag[i].w);
        atomicAdd(dbeta + 4 * c4 + 0, ab[i].x); atomicAdd(dbeta + 4 * c4 + 1, ab[i].y);
        atomicAdd(dbeta + 4 * c4 + 2, ab[i].z); atomicAdd(dbeta + 4 * c4 + 3, ab[i].w);
      }
    }
  }
}

extern "C" int mfnet_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma,
                                   const float* beta, void* y, int y_dtype, int64_t ldy, float* mean,
                                   float* rstd, int64_t rows, int64_t C, float eps, hipStream_t stream) {
  MF_CHECK_ARG(x && y && gamma && beta, "layernorm_fwd: null pointer");
  MF_CHECK_ARG(rows >= 0 && C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXV,
               "layernorm_fwd: C=%lld must be a multiple of 4 and <= %d", (long long)C, 64 * 4 * LN_MAXV);
  MF_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= C && ldy >= C, "layernorm_fwd: bad leading dims");
  if (rows == 0) return MFNET_OK;
  dim3 grid((unsigned)ceil_div64(rows, 4)), block(256);
#define LN_FWD(TI, TO)                                                                                      \
  hipLaunchKernelGGL((ln_fwd_kernel<TI, TO>), grid, block, 0, stream, (const TI*)x, ldx, gamma, beta, (TO*)y, \
                     ldy, mean, rstd, rows, (int)C, eps)
  if (x_dtype == MFNET_F32 && y_dtype == MFNET_BF16) LN_FWD(float, bf16_t);
  else if (x_dtype == MFNET_F32 && y_dtype == MFNET_F32) LN_FWD(float, float);
  else if (x_dtype == MFNET_BF16 && y_dtype == MFNET_BF16) LN_FWD(bf16_t, bf16_t);
  else if (x_dtype == MFNET_BF16 && y_dtype == MFNET_F32) LN_FWD(bf16_t, float);
  else { mfnet_set_error("layernorm_fwd: bad dtype"); return MFNET_ERR_ARG; }
#undef LN_FWD
  MF_LAUNCH_CHECK("layernorm_fwd");
  return MFNET_OK;
}

extern "C" int mfnet_layernorm_bwd(const void* dy, int dy_dtype, int64_t lddy, const void* x, int x_dtype,
                                   int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                   const float* dres, float* dx, void* dx_bf16, int64_t lddx, float* dgamma,
                                   float* dbeta, int64_t rows, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(dy && x && mean && rstd && gamma && (dx || dx_bf16), "layernorm_bwd: null pointer");
  MF_CHECK_ARG(C > 0 && C % 4 == 0 && C <= 64 * 4 * LN_MAXV, "layernorm_bwd: bad C=%lld", (long long)C);
  MF_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma/dbeta must come together");
  MF_CHECK_ARG(lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0, "layernorm_bwd: bad leading dims");
  if (rows == 0) return MFNET_OK;
  int64_t nblk = ceil_div64(rows, 4);
  if (dgamma && nblk > 512) nblk = 512;       // fewer waves -> fewer column atomics
  if (nblk > 65535 * 16) nblk = 65535 * 16;
  dim3 grid((unsigned)nblk), block(256);
#define LN_BWD_N(TD, TX, NVL)                                                                                      \
  do {                                                                                                             \
    if (dgamma)                                                                                                    \
      hipLaunchKernelGGL((ln_bwd_kernel<TD, TX, NVL, true>), grid, block, 0, stream, (const TD*)dy, lddy, (const TX*)x, ldx, \
                         mean, rstd, gamma, dres, dx, (bf16_t*)dx_bf16, lddx, dgamma, dbeta, rows, (int)C);        \
    else                                                                                                           \
      hipLaunchKernelGGL((ln_bwd_kernel<TD, TX, NVL, false>), grid, block, 0, stream, (const TD*)dy, lddy, (const TX*)x, ldx, \
                         mean, rstd, gamma, dres, dx, (bf16_t*)dx_bf16, lddx, dgamma, dbeta, rows, (int)C);        \
  } while (0)
#define LN_BWD(TD, TX)                                                                                             \
  do {                                                                                                             \
    if (C <= 256) LN_BWD_N(TD, TX, 1);                                                                             \
    else if (C <= 768) LN_BWD_N(TD, TX, 3);                                                                        \
    else if (C <= 1280) LN_BWD_N(TD, TX, 5);                                                                       \
    else LN_BWD_N(TD, TX, LN_MAXV);                                                                                \
  } while (0)
  if (dy_dtype == MFNET_BF16 && x_dtype == MFNET_F32) LN_BWD(bf16_t, float);
  else if (dy_dtype == MFNET_F32 && x_dtype == MFNET_F32) LN_BWD(float, float);
  else if (dy_dtype == MFNET_BF16 && x_dtype == MFNET_BF16) LN_BWD(bf16_t, bf16_t);
  else if (dy_dtype == MFNET_F32 && x_dtype == MFNET_BF16) LN_BWD(float, bf16_t);
  else { mfnet_set_error("layernorm_bwd: bad dtype"); return MFNET_ERR_ARG; }
#undef LN_BWD_N
#undef LN_BWD
  MF_LAUNCH_CHECK("layernorm_bwd");
  return MFNET_OK;
}
