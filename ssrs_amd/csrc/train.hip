// Training / inference glue kernels (all HBM-bound):
//  * cross-entropy over NCHW logits without the transpose copy of utils2.py:115-127
//    (+ ignore_index variant train_multiscale_main.py:30-43), forward + backward
//  * fused flat SGD (momentum, weight decay) = torch.optim.SGD as set up in trainer.py:111-116
//  * sliding-window logit accumulation + argmax (evaluator.py:169-176), DSM min-max (evaluator.py:100-111)
//  * MMAdapter cross-modal mix (adapter_fusionblock.py:105-108), GELU, bias gradient
#include "common.h"
#include <float.h>

static inline unsigned grid_for(int64_t work_items, int64_t cap = 8192) {
  int64_t g = ceil_div64(work_items, 256);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ----------------------------------------------------------------------------------- cross entropy
#define CE_MAXC 32
// acc[0] += sum w[t]*nll ; acc[1] += sum w[t] ; acc[2] += #(argmax == t) ; acc[3] += #valid pixels
// (pixels with t == ignore contribute nothing)
// CT > 0: the class count at compile time - the pixel's CT logits are requested together and kept in registers (the runtime-C
// loop issued one dependent load per class, twice: 37 us for a 25 MB map; the MFNet heads have 6 or 2 classes).
template <int CT>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, float* __restrict__ acc,
                                                     unsigned char* __restrict__ pred, int64_t B, int Crt, int64_t HW,
                                                     int64_t ignore) {
  const int C = CT > 0 ? CT : Crt;
  float ls = 0.f, ws = 0.f, nc = 0.f, nv = 0.f;
  const int64_t total = B * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, px = i - b * HW;
    const float* lp = logits + b * C * HW + px;
    float mx = -FLT_MAX;
    int am = 0;
    float vr[CT > 0 ? CT : 1];
    if (CT > 0) {
#pragma unroll
      for (int c = 0; c < CT; ++c) vr[c] = lp[(int64_t)c * HW];
#pragma unroll
      for (int c = 0; c < CT; ++c) if (vr[c] > mx) { mx = vr[c]; am = c; }
    } else {
      for (int c = 0; c < C; ++c) {
        const float v = lp[(int64_t)c * HW];
        if (v > mx) { mx = v; am = c; }
      }
    }
    if (pred) pred[i] = (unsigned char)am;
    const int64_t t = target[i];
    if (t == ignore || t < 0 || t >= C) continue;
    float se = 0.f;
    if (CT > 0) {
#pragma unroll
      for (int c = 0; c < CT; ++c) se += __expf(vr[c] - mx);
    } else {
      for (int c = 0; c < C; ++c) se += __expf(lp[(int64_t)c * HW] - mx);
    }
    const float w = weight ? weight[t] : 1.f;
    ls += w * (mx + __logf(se) - lp[t * HW]);
    ws += w;
    nc += (am == (int)t) ? 1.f : 0.f;
    nv += 1.f;
  }
  ls = wave_sum(ls);
  ws = wave_sum(ws);
  nc = wave_sum(nc);
  nv = wave_sum(nv);
  __shared__ float sh[16];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[wave] = ls; sh[4 + wave] = ws; sh[8 + wave] = nc; sh[12 + wave] = nv; }
  __syncthreads();
  if (threadIdx.x < 4) atomicAdd(acc + threadIdx.x, sh[4 * threadIdx.x] + sh[4 * threadIdx.x + 1] + sh[4 * threadIdx.x + 2] + sh[4 * threadIdx.x + 3]);
}
// dlogits = gscale * w[t] * (softmax - onehot) / acc[1]
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, const float* __restrict__ acc,
                                                     float* __restrict__ dlogits, float gscale, int64_t B, int C,
                                                     int64_t HW, int64_t ignore) {
  const float inv = gscale / acc[1];
  const int64_t total = B * HW;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, px = i - b * HW;
    const float* lp = logits + b * C * HW + px;
    float* dp = dlogits + b * C * HW + px;
    const int64_t t = target[i];
    if (t == ignore || t < 0 || t >= C) {
      for (int c = 0; c < C; ++c) dp[(int64_t)c * HW] = 0.f;
      continue;
    }
    float mx = -FLT_MAX;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[(int64_t)c * HW]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(lp[(int64_t)c * HW] - mx);
    const float w = (weight ? weight[t] : 1.f) * inv;
    const float rs = 1.f / se;
    for (int c = 0; c < C; ++c) {
      const float pr = __expf(lp[(int64_t)c * HW] - mx) * rs;
      dp[(int64_t)c * HW] = w * (pr - (c == t ? 1.f : 0.f));
    }
  }
}
extern "C" int mfnet_ce_loss_fwd(const float* logits, const int64_t* target, const float* weight, float* acc,
                                 unsigned char* pred, int64_t B, int64_t C, int64_t HW, int64_t ignore, hipStream_t stream) {
  MF_CHECK_ARG(logits && target && acc && C > 0 && C < 256, "ce_loss_fwd: bad args");
  if (B * HW <= 0) return MFNET_OK;
  const dim3 grid(grid_for(B * HW, 1024));      // every workgroup ends with 4 same-address float atomics: keep them few
  if (C == 6) hipLaunchKernelGGL(ce_fwd_kernel<6>, grid, dim3(256), 0, stream, logits, target, weight, acc, pred, B, (int)C, HW, ignore);
  else if (C == 2) hipLaunchKernelGGL(ce_fwd_kernel<2>, grid, dim3(256), 0, stream, logits, target, weight, acc, pred, B, (int)C, HW, ignore);
  else hipLaunchKernelGGL(ce_fwd_kernel<0>, grid, dim3(256), 0, stream, logits, target, weight, acc, pred, B, (int)C, HW, ignore);
  MF_LAUNCH_CHECK("ce_loss_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_ce_loss_bwd(const float* logits, const int64_t* target, const float* weight, const float* acc,
                                 float* dlogits, float gscale, int64_t B, int64_t C, int64_t HW, int64_t ignore, hipStream_t stream) {
  MF_CHECK_ARG(logits && target && acc && dlogits && C > 0 && C < 256, "ce_loss_bwd: bad args");
  if (B * HW <= 0) return MFNET_OK;
  hipLaunchKernelGGL(ce_bwd_kernel, dim3(grid_for(B * HW)), dim3(256), 0, stream, logits, target, weight, acc, dlogits, gscale, B, (int)C, HW, ignore);
  MF_LAUNCH_CHECK("ce_loss_bwd");
  return MFNET_OK;
}

// ----------------------------------------------------------------------------------- SGD
// d = gscale*g + wd*p ; buf = mom*buf + d ; p -= lr*buf   (buf zero-initialised => first step buf = d).
// gscale = 1/world folds the data-parallel gradient average into the update.
__global__ __launch_bounds__(256) void sgd_flat_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ buf, int64_t n, float lr, float mom, float wd,
                                                       float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float pv = p[i];
    float d = gscale * g[i] + wd * pv;
    if (buf) { d = mom * buf[i] + d; buf[i] = d; }
    p[i] = pv - lr * d;
  }
}
extern "C" int mfnet_sgd_flat(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd,
                              float gscale, hipStream_t stream) {
  MF_CHECK_ARG(p && g && n >= 0, "sgd_flat: bad args");
  MF_CHECK_ARG(momentum == 0.f || buf, "sgd_flat: momentum needs a buffer");
  if (n == 0) return MFNET_OK;
  hipLaunchKernelGGL(sgd_flat_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, momentum != 0.f ? buf : nullptr, n, lr, momentum, wd, gscale);
  MF_LAUNCH_CHECK("sgd_flat");
  return MFNET_OK;
}

// ----------------------------------------------------------------------------------- sliding-window merge
// pred [C,H,W] fp32 += logits[n] [C,h,w] at (x=row, y=col) = coords[n]; evaluator.py:169-171.
// atomic=0 requires the tiles of this launch to be pairwise disjoint.
__global__ __launch_bounds__(256) void tile_acc_kernel(float* __restrict__ pred, const float* __restrict__ logits,
                                                       const int* __restrict__ coords, int64_t n, int C, int64_t H,
                                                       int64_t W, int h, int w, int atomic) {
  const int w4 = w >> 2;
  const int64_t per = (int64_t)C * h * w4, total = n * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / per;
    int64_t r = i - t * per;
    const int c = (int)(r / ((int64_t)h * w4));
    r -= (int64_t)c * h * w4;
    const int yy = (int)(r / w4), x4 = (int)(r - (int64_t)yy * w4);
    const int64_t row = coords[2 * t] + yy, col = coords[2 * t + 1] + 4 * x4;
    const float4 v = *reinterpret_cast<const float4*>(logits + ((t * C + c) * h + yy) * (int64_t)w + 4 * x4);
    float* d = pred + ((int64_t)c * H + row) * W + col;
    if (atomic) {
      atomicAdd(d, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
    } else {
      d[0] += v.x; d[1] += v.y; d[2] += v.z; d[3] += v.w;
    }
  }
}
extern "C" int mfnet_tile_accumulate(float* pred, const float* logits, const int* coords, int64_t n, int64_t C,
                                     int64_t H, int64_t W, int64_t h, int64_t w, int atomic, hipStream_t stream) {
  MF_CHECK_ARG(pred && logits && coords && C > 0 && h > 0 && w > 0 && w % 4 == 0 && h <= H && w <= W, "tile_accumulate: bad args");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(tile_acc_kernel, dim3(grid_for(n * C * h * (w / 4))), dim3(256), 0, stream, pred, logits, coords, n, (int)C, H, W, (int)h, (int)w, atomic);
  MF_LAUNCH_CHECK("tile_accumulate");
  return MFNET_OK;
}
// argmax over the leading C dim of [C, N] (first max wins, like np.argmax)
__global__ __launch_bounds__(256) void argmax_chw_kernel(const float* __restrict__ pred, unsigned char* __restrict__ out, int C, int64_t N) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    float mx = pred[i];
    int am = 0;
    for (int c = 1; c < C; ++c) {
      const float v = pred[(int64_t)c * N + i];
      if (v > mx) { mx = v; am = c; }
    }
    out[i] = (unsigned char)am;
  }
}
extern "C" int mfnet_argmax_chw(const float* pred, unsigned char* out, int64_t C, int64_t N, hipStream_t stream) {
  MF_CHECK_ARG(pred && out && C > 0 && C < 256, "argmax_chw: bad args");
  if (N <= 0) return MFNET_OK;
  hipLaunchKernelGGL(argmax_chw_kernel, dim3(grid_for(N)), dim3(256), 0, stream, pred, out, (int)C, N);
  MF_LAUNCH_CHECK("argmax_chw");
  return MFNET_OK;
}

// float64 accumulator twins: the reference merges window logits in a float64 numpy array in window order (evaluator.py:134,
// 169-176). Sums of a few dozen fp32 values are exact in float64, so the order the atomics land in cannot change a bit:
// the merged scores, and with them the argmax mask, are those of the reference for equal logits - and deterministic.
// src is fp32 logits (src_f64 = 0) or a float64 band received from a neighbouring rank (src_f64 = 1).
template <typename TSrc>
__global__ __launch_bounds__(256) void tile_acc_f64_kernel(double* __restrict__ pred, const TSrc* __restrict__ src, const int* __restrict__ coords,
                                                           int64_t n, int C, int64_t H, int64_t W, int h, int w, int atomic) {
  const int64_t per = (int64_t)C * h * w, total = n * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / per;
    int64_t r = i - t * per;
    const int c = (int)(r / ((int64_t)h * w));
    r -= (int64_t)c * h * w;
    const int yy = (int)(r / w), xx = (int)(r - (int64_t)yy * w);
    double* d = pred + ((int64_t)c * H + coords[2 * t] + yy) * W + coords[2 * t + 1] + xx;
    const double v = (double)src[i];
    if (atomic) atomicAdd(d, v); else *d += v;
  }
}
extern "C" int mfnet_tile_accumulate_f64(double* pred, const void* src, int src_f64, const int* coords, int64_t n, int64_t C,
                                         int64_t H, int64_t W, int64_t h, int64_t w, int atomic, hipStream_t stream) {
  MF_CHECK_ARG(pred && src && coords && C > 0 && h > 0 && w > 0 && h <= H && w <= W, "tile_accumulate_f64: bad args");
  if (n <= 0) return MFNET_OK;
  const dim3 grid(grid_for(n * C * h * w));
  if (src_f64) hipLaunchKernelGGL(tile_acc_f64_kernel<double>, grid, dim3(256), 0, stream, pred, (const double*)src, coords, n, (int)C, H, W, (int)h, (int)w, atomic);
  else hipLaunchKernelGGL(tile_acc_f64_kernel<float>, grid, dim3(256), 0, stream, pred, (const float*)src, coords, n, (int)C, H, W, (int)h, (int)w, atomic);
  MF_LAUNCH_CHECK("tile_accumulate_f64");
  return MFNET_OK;
}
__global__ __launch_bounds__(256) void argmax_chw_f64_kernel(const double* __restrict__ pred, unsigned char* __restrict__ out, int C, int64_t N) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    double mx = pred[i];
    int am = 0;
    for (int c = 1; c < C; ++c) {
      const double v = pred[(int64_t)c * N + i];
      if (v > mx) { mx = v; am = c; }
    }
    out[i] = (unsigned char)am;
  }
}
extern "C" int mfnet_argmax_chw_f64(const double* pred, unsigned char* out, int64_t C, int64_t N, hipStream_t stream) {
  MF_CHECK_ARG(pred && out && C > 0 && C < 256, "argmax_chw_f64: bad args");
  if (N <= 0) return MFNET_OK;
  hipLaunchKernelGGL(argmax_chw_f64_kernel, dim3(grid_for(N)), dim3(256), 0, stream, pred, out, (int)C, N);
  MF_LAUNCH_CHECK("argmax_chw_f64");
  return MFNET_OK;
}

// ----------------------------------------------------------------------------------- min-max normalise
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__global__ void minmax_init_kernel(int* mm) { mm[0] = 0x7fffffff; mm[1] = (int)0x80000000; }
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, int* __restrict__ mm, int64_t n) {
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    lo = fminf(lo, v); hi = fmaxf(hi, v);
  }
  hi = wave_max(hi);
  lo = -wave_max(-lo);
  if ((threadIdx.x & 63) == 0) { atomicMin(mm, f2ord(lo)); atomicMax(mm + 1, f2ord(hi)); }
}
__global__ void minmax_fin_kernel(int* mm) { float* f = (float*)mm; const float a = ord2f(mm[0]), b = ord2f(mm[1]); f[0] = a; f[1] = b; }
// out[0] = min(x), out[1] = max(x)
extern "C" int mfnet_minmax(const float* x, float* out, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(x && out && n > 0, "minmax: bad args");
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, stream, (int*)out);
  hipLaunchKernelGGL(minmax_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, stream, x, (int*)out, n);
  hipLaunchKernelGGL(minmax_fin_kernel, dim3(1), dim3(1), 0, stream, (int*)out);
  MF_LAUNCH_CHECK("minmax");
  return MFNET_OK;
}
// mode 0: y = (x - mm[0]) / (mm[1] - mm[0])  (evaluator.py:109-111)
// mode 1: same, but y = 0 when max == min     (inference.py:551-555)
__global__ __launch_bounds__(256) void scale_shift_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ mm, int64_t n, int mode) {
  const float lo = mm[0], rng = mm[1] - mm[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = (mode == 1 && !(rng > 0.f)) ? 0.f : (x[i] - lo) / rng;
}
extern "C" int mfnet_scale_shift(const float* x, float* y, const float* mm, float unused, int64_t n, int mode, hipStream_t stream) {
  (void)unused;
  MF_CHECK_ARG(x && y && mm && n >= 0, "scale_shift: bad args");
  if (n == 0) return MFNET_OK;
  hipLaunchKernelGGL(scale_shift_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, mm, n, mode);
  MF_LAUNCH_CHECK("scale_shift");
  return MFNET_OK;
}

// ----------------------------------------------------------------------------------- MMAdapter mix
// out[x_t] = h[x_t] + wx*ada[x_t] + (1-wx)*ada[y_t] ; out[y_t] = h[y_t] + wy*ada[y_t] + (1-wy)*ada[x_t]
__global__ __launch_bounds__(256) void adapter_mix_kernel(const float* h, const float* __restrict__ ada,
                                                          const float* __restrict__ wx, const float* __restrict__ wy,
                                                          float* out, int64_t half4) {
  const float fx = wx[0], fy = wy[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < half4; i += (int64_t)gridDim.x * 256) {
    const float4 ax = reinterpret_cast<const float4*>(ada)[i], ay = reinterpret_cast<const float4*>(ada)[half4 + i];
    const float4 hx = reinterpret_cast<const float4*>(h)[i], hy = reinterpret_cast<const float4*>(h)[half4 + i];
    reinterpret_cast<float4*>(out)[i] = make_float4(hx.x + fx * ax.x + (1 - fx) * ay.x, hx.y + fx * ax.y + (1 - fx) * ay.y,
                                                     hx.z + fx * ax.z + (1 - fx) * ay.z, hx.w + fx * ax.w + (1 - fx) * ay.w);
    reinterpret_cast<float4*>(out)[half4 + i] = make_float4(hy.x + fy * ay.x + (1 - fy) * ax.x, hy.y + fy * ay.y + (1 - fy) * ax.y,
                                                             hy.z + fy * ay.z + (1 - fy) * ax.z, hy.w + fy * ay.w + (1 - fy) * ax.w);
  }
}
extern "C" int mfnet_adapter_mix(const float* h, const float* ada, const float* wx, const float* wy, float* out,
                                 void* reserved, int64_t M, int64_t D, hipStream_t stream) {
  (void)reserved;
  MF_CHECK_ARG(h && ada && wx && wy && out && M % 2 == 0 && D % 4 == 0, "adapter_mix: bad args");
  if (M == 0) return MFNET_OK;
  hipLaunchKernelGGL(adapter_mix_kernel, dim3(grid_for(M / 2 * D / 4)), dim3(256), 0, stream, h, ada, wx, wy, out, M / 2 * D / 4);
  MF_LAUNCH_CHECK("adapter_mix");
  return MFNET_OK;
}
// dada[x_t] = wx*dh[x_t] + (1-wy)*dh[y_t] ; dada[y_t] = (1-wx)*dh[x_t] + wy*dh[y_t]   (bf16 out)
// dw[slot][0] += sum dh[x]*(ada[x]-ada[y]) ; dw[slot][1] += sum dh[y]*(ada[y]-ada[x]) over ADMIX_SLOTS slots (same-address
// float atomics serialise: 2048 per address took 50 of the kernel's 61 us); the caller sums the slots
#define ADMIX_SLOTS 16
template <bool F32OUT>
__global__ __launch_bounds__(256) void adapter_mix_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ ada,
                                                              const float* __restrict__ wx, const float* __restrict__ wy,
                                                              void* __restrict__ dada_, float* __restrict__ dw, int64_t half4) {
  const float fx = wx[0], fy = wy[0];
  float sx = 0.f, sy = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < half4; i += (int64_t)gridDim.x * 256) {
    const float4 ax = reinterpret_cast<const float4*>(ada)[i], ay = reinterpret_cast<const float4*>(ada)[half4 + i];
    const float4 gx = reinterpret_cast<const float4*>(dh)[i], gy = reinterpret_cast<const float4*>(dh)[half4 + i];
    const float4 dx4 = make_float4(fx * gx.x + (1 - fy) * gy.x, fx * gx.y + (1 - fy) * gy.y, fx * gx.z + (1 - fy) * gy.z, fx * gx.w + (1 - fy) * gy.w);
    const float4 dy4 = make_float4((1 - fx) * gx.x + fy * gy.x, (1 - fx) * gx.y + fy * gy.y, (1 - fx) * gx.z + fy * gy.z, (1 - fx) * gx.w + fy * gy.w);
    if (F32OUT) {                        // the fp32 training path keeps the gradient in fp32
      reinterpret_cast<float4*>(dada_)[i] = dx4;
      reinterpret_cast<float4*>(dada_)[half4 + i] = dy4;
    } else {
      u32x2_t o;
      o.x = pack_bf16x2(dx4.x, dx4.y); o.y = pack_bf16x2(dx4.z, dx4.w);
      reinterpret_cast<u32x2_t*>(dada_)[i] = o;
      o.x = pack_bf16x2(dy4.x, dy4.y); o.y = pack_bf16x2(dy4.z, dy4.w);
      reinterpret_cast<u32x2_t*>(dada_)[half4 + i] = o;
    }
    const float dxs = (ax.x - ay.x), dys = (ax.y - ay.y), dzs = (ax.z - ay.z), dws = (ax.w - ay.w);
    sx += gx.x * dxs + gx.y * dys + gx.z * dzs + gx.w * dws;
    sy -= gy.x * dxs + gy.y * dys + gy.z * dzs + gy.w * dws;
  }
  __shared__ float red[2][4];
  sx = wave_sum(sx); sy = wave_sum(sy);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sx; red[1][threadIdx.x >> 6] = sy; }
  __syncthreads();
  if (threadIdx.x < 2)
    atomicAdd(dw + 2 * (blockIdx.x % ADMIX_SLOTS) + threadIdx.x, (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}
extern "C" int mfnet_adapter_mix_bwd(const float* dh, const float* ada, const float* wx, const float* wy, void* dada,
                                     float* dw, void* reserved, int64_t M, int64_t D, hipStream_t stream) {
  (void)reserved;
  MF_CHECK_ARG(dh && ada && wx && wy && dada && dw && M % 2 == 0 && D % 4 == 0, "adapter_mix_bwd: bad args");
  if (M == 0) return MFNET_OK;
  hipLaunchKernelGGL(adapter_mix_bwd_kernel<false>, dim3(grid_for(M / 2 * D / 4, 512)), dim3(256), 0, stream, dh, ada, wx, wy, dada, dw, M / 2 * D / 4);
  MF_LAUNCH_CHECK("adapter_mix_bwd");
  return MFNET_OK;
}
extern "C" int mfnet_adapter_mix_bwd_f32(const float* dh, const float* ada, const float* wx, const float* wy, float* dada, float* dw, int64_t M,
                                         int64_t D, hipStream_t stream) {
  MF_CHECK_ARG(dh && ada && wx && wy && dada && dw && M % 2 == 0 && D % 4 == 0, "adapter_mix_bwd_f32: bad args");
  if (M == 0) return MFNET_OK;
  hipLaunchKernelGGL(adapter_mix_bwd_kernel<true>, dim3(grid_for(M / 2 * D / 4, 512)), dim3(256), 0, stream, dh, ada, wx, wy, (void*)dada, dw, M / 2 * D / 4);
  MF_LAUNCH_CHECK("adapter_mix_bwd_f32");
  return MFNET_OK;
}

// ----------------------------------------------------------------------------------- GELU / add / bias grad
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = gelu_f(x[i]);
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] = dy[i] * gelu_grad_f(x[i]);
}
extern "C" int mfnet_gelu_fwd(const float* x, float* y, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(x && y, "gelu_fwd: null");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, y, n);
  MF_LAUNCH_CHECK("gelu_fwd");
  return MFNET_OK;
}
extern "C" int mfnet_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(x && dy && dx, "gelu_bwd: null");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, stream, x, dy, dx, n);
  MF_LAUNCH_CHECK("gelu_bwd");
  return MFNET_OK;
}
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ o, int64_t n2) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
    const unsigned int x = reinterpret_cast<const unsigned int*>(a)[i], y = reinterpret_cast<const unsigned int*>(b)[i];
    reinterpret_cast<unsigned int*>(o)[i] = pack_bf16x2(bf16_lo(x) + bf16_lo(y), bf16_hi(x) + bf16_hi(y));
  }
}
extern "C" int mfnet_add_bf16(const void* a, const void* b, void* out, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(a && b && out && n % 2 == 0, "add_bf16: bad args");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(add_bf16_kernel, dim3(grid_for(n / 2)), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n / 2);
  MF_LAUNCH_CHECK("add_bf16");
  return MFNET_OK;
}
// out[n] += sum_m dy[m, n]   (dy bf16; out fp32, accumulated with float atomics)
__global__ __launch_bounds__(256) void bias_grad_kernel(const bf16_t* __restrict__ dy, int64_t ld, float* __restrict__ out,
                                                        int64_t M, int N, int64_t rows_per_block) {
  const int n2 = blockIdx.x * 256 + threadIdx.x;   // pair of columns
  if (2 * n2 >= N) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float a = 0.f, b = 0.f;
  for (int64_t m = r0; m < r1; ++m) {
    const unsigned int w = *reinterpret_cast<const unsigned int*>(dy + m * ld + 2 * n2);
    a += bf16_lo(w); b += bf16_hi(w);
  }
  atomicAdd(out + 2 * n2, a);
  atomicAdd(out + 2 * n2 + 1, b);
}
extern "C" int mfnet_bias_grad(const void* dy, int64_t ld, float* out, int64_t M, int64_t N, hipStream_t stream) {
  MF_CHECK_ARG(dy && out && N % 2 == 0 && ld >= N && ld % 2 == 0, "bias_grad: bad args");
  if (M <= 0) return MFNET_OK;
  const int64_t rpb = 64;
  dim3 grid((unsigned)ceil_div64(N / 2, 256), (unsigned)ceil_div64(M, rpb));
  hipLaunchKernelGGL(bias_grad_kernel, grid, dim3(256), 0, stream, (const bf16_t*)dy, ld, out, M, (int)N, rpb);
  MF_LAUNCH_CHECK("bias_grad");
  return MFNET_OK;
}
