// Sliding-window large-raster inference, on-device pre/post steps around the tile loop:
//   * RGB normalisation (inference.py:464-484: /255 then ImageNet mean / std), tile extraction with the
//     reflect padding of short edge tiles (inference.py:713-737), per-tile DSM min-max (inference.py:547-555)
//   * softmax class probability of each tile accumulated into a raster sum + coverage count and blended
//     (mean over the covering tiles), thresholded building mask (inference.py:562-566,582)
// The evaluator-style merge (sum of raw logits, argmax; evaluator.py:169-176) lives in train.hip
// (mfnet_tile_accumulate / mfnet_argmax_chw).
#include "common.h"
#include <float.h>

static inline unsigned grid_for(int64_t work_items, int64_t cap = 16384) {
  int64_t g = ceil_div64(work_items, 256);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// out[c,y,x] = (in[c,y,x] * scale - mean[c]) / std[c], in uint8
__global__ __launch_bounds__(256) void normalize_u8_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, int64_t hw, int C,
                                                           float scale, float m0, float m1, float m2, float s0, float s1, float s2) {
  const int64_t total = hw * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i / hw);
    const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), s = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[i] = ((float)in[i] * scale - m) / s;
  }
}
extern "C" int mfnet_normalize_rgb_u8(const unsigned char* in, float* out, int64_t H, int64_t W, float scale, const float* mean3,
                                      const float* std3, hipStream_t stream) {
  MF_CHECK_ARG(in && out && mean3 && std3 && H > 0 && W > 0, "normalize_rgb_u8: bad args (mean3 / std3 are HOST arrays of 3 floats)");
  hipLaunchKernelGGL(normalize_u8_kernel, dim3(grid_for(3 * H * W)), dim3(256), 0, stream, in, out, H * W, 3, scale, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2]);
  MF_LAUNCH_CHECK("normalize_rgb_u8");
  return MFNET_OK;
}

// numpy 'reflect' index for a tile of valid length n padded to a longer window: the reflection is periodic
// (period 2n-2), so a pad longer than n-1 reflects a second time as np.pad(mode='reflect') does (inference.py:732)
__device__ __forceinline__ int reflect_idx(int i, int n) {
  if (i < n) return i;
  if (n <= 1) return 0;
  const int m = i % (2 * n - 2);
  return m < n ? m : 2 * n - 2 - m;
}
// coords[t] = (row, col, valid_h, valid_w); rgb [C,H,W] fp32, dsm [H,W] fp32 -> rgb_out [n,C,h,w], dsm_out [n,h,w]
__global__ __launch_bounds__(256) void extract_tiles_kernel(const float* __restrict__ rgb, const float* __restrict__ dsm, const int* __restrict__ coords,
                                                            float* __restrict__ rgb_out, float* __restrict__ dsm_out, int64_t n, int C, int64_t H,
                                                            int64_t W, int h, int w) {
  const int64_t per = (int64_t)(C + 1) * h * w, total = n * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / per;
    int64_t r = i - t * per;
    const int c = (int)(r / ((int64_t)h * w));
    r -= (int64_t)c * h * w;
    const int y = (int)(r / w), x = (int)(r - (int64_t)y * w);
    const int* cd = coords + 4 * t;
    const int64_t sy = cd[0] + reflect_idx(y, cd[2]), sx = cd[1] + reflect_idx(x, cd[3]);
    if (c < C) rgb_out[((t * C + c) * h + y) * (int64_t)w + x] = rgb[((int64_t)c * H + sy) * W + sx];
    else dsm_out[(t * h + y) * (int64_t)w + x] = dsm[sy * W + sx];
  }
}
extern "C" int mfnet_extract_tiles(const float* rgb, const float* dsm, const int* coords, float* rgb_out, float* dsm_out, int64_t n, int64_t C,
                                   int64_t H, int64_t W, int64_t h, int64_t w, hipStream_t stream) {
  MF_CHECK_ARG(rgb && dsm && coords && rgb_out && dsm_out && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, "extract_tiles: bad args");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(extract_tiles_kernel, dim3(grid_for(n * (C + 1) * h * w)), dim3(256), 0, stream, rgb, dsm, coords, rgb_out, dsm_out, n, (int)C, H, W, (int)h, (int)w);
  MF_LAUNCH_CHECK("extract_tiles");
  return MFNET_OK;
}

// per-tile min-max normalisation in place: (x - min) / (max - min), 0 when max == min; one block per tile
__global__ __launch_bounds__(256) void tile_minmax_norm_kernel(float* __restrict__ x, int64_t per, int keep_flat) {
  __shared__ float lo_s[4], hi_s[4];
  float* p = x + (int64_t)blockIdx.x * per;
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int64_t i = threadIdx.x; i < per; i += 256) { const float v = p[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  hi = wave_max(hi); lo = -wave_max(-lo);
  if ((threadIdx.x & 63) == 0) { lo_s[threadIdx.x >> 6] = lo; hi_s[threadIdx.x >> 6] = hi; }
  __syncthreads();
  lo = fminf(fminf(lo_s[0], lo_s[1]), fminf(lo_s[2], lo_s[3]));
  hi = fmaxf(fmaxf(hi_s[0], hi_s[1]), fmaxf(hi_s[2], hi_s[3]));
  const float rng = hi - lo;
  if (!(rng > 0.f) && keep_flat) return;       // multiscale_dataset.py:206 leaves a flat patch as it is
  for (int64_t i = threadIdx.x; i < per; i += 256) p[i] = rng > 0.f ? (p[i] - lo) / rng : 0.f;
}
// tiles of at most 65536 values (the 256 x 256 windows): one 1024-thread workgroup holds its tile in registers (16 float4 per
// thread, all loads in flight together) - one pass over memory instead of two scalar ones (118 -> 14 us for 32 tiles).
// min / max are order-independent: same bits as the kernel above.
__global__ __launch_bounds__(1024) void tile_minmax_norm_reg_kernel(float* __restrict__ x, int per4, int keep_flat) {
  __shared__ float lo_s[16], hi_s[16];
  float4* p = reinterpret_cast<float4*>(x) + (int64_t)blockIdx.x * per4;
  float4 v[16];
  float lo = FLT_MAX, hi = -FLT_MAX;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = threadIdx.x + 1024 * j;
    if (i < per4) v[j] = p[i];
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if ((int)threadIdx.x + 1024 * j < per4) {
      lo = fminf(fminf(lo, v[j].x), fminf(v[j].y, fminf(v[j].z, v[j].w)));
      hi = fmaxf(fmaxf(hi, v[j].x), fmaxf(v[j].y, fmaxf(v[j].z, v[j].w)));
    }
  }
  hi = wave_max(hi); lo = -wave_max(-lo);
  if ((threadIdx.x & 63) == 0) { lo_s[threadIdx.x >> 6] = lo; hi_s[threadIdx.x >> 6] = hi; }
  __syncthreads();
  lo = lo_s[0]; hi = hi_s[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) { lo = fminf(lo, lo_s[w]); hi = fmaxf(hi, hi_s[w]); }
  const float rng = hi - lo;
  if (!(rng > 0.f) && keep_flat) return;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = threadIdx.x + 1024 * j;
    if (i < per4)
      p[i] = rng > 0.f ? make_float4((v[j].x - lo) / rng, (v[j].y - lo) / rng, (v[j].z - lo) / rng, (v[j].w - lo) / rng) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
extern "C" int mfnet_tile_minmax_norm(float* tiles, int64_t n, int64_t per_tile, int keep_flat, hipStream_t stream) {
  MF_CHECK_ARG(tiles && per_tile > 0 && n < (1ll << 31), "tile_minmax_norm: bad args");
  if (n <= 0) return MFNET_OK;
  if (per_tile % 4 == 0 && per_tile <= 65536 && (uintptr_t)tiles % 16 == 0)
    hipLaunchKernelGGL(tile_minmax_norm_reg_kernel, dim3((unsigned)n), dim3(1024), 0, stream, tiles, (int)(per_tile / 4), keep_flat);
  else
  hipLaunchKernelGGL(tile_minmax_norm_kernel, dim3((unsigned)n), dim3(256), 0, stream, tiles, per_tile, keep_flat);
  MF_LAUNCH_CHECK("tile_minmax_norm");
  return MFNET_OK;
}

// sum[row+y, col+x] += softmax(logits[t,:,y,x])[cls] ; cnt[...] += 1 over the valid extent of each tile
// Order-independent blend: the sums are 64-bit FIXED-POINT integers (probability * 2^40, rounded once per tile in double)
// and the tile counts integers - integer atomic adds commute, so the blended raster is bit-identical from run to run and
// for every window order / rank count (fp32 atomics left order-dependent last bits). 2^40 resolves 9e-13, finer than fp32
// near any probability the 0.5 threshold can see; 2^23 overlapping tiles fit before the sum could overflow.
#define MF_PROB_FIX 1099511627776.0   /* 2^40 */
__global__ __launch_bounds__(256) void tile_prob_accumulate_kernel(unsigned long long* __restrict__ sum, unsigned long long* __restrict__ cnt,
                                                                   const float* __restrict__ logits,
                                                                   const int* __restrict__ coords, int64_t n, int C, int64_t W, int h, int w, int cls) {
  const int64_t per = (int64_t)h * w, total = n * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / per;
    const int64_t r = i - t * per;
    const int y = (int)(r / w), x = (int)(r - (int64_t)y * w);
    const int* cd = coords + 4 * t;
    if (y >= cd[2] || x >= cd[3]) continue;
    const float* lp = logits + t * C * per + r;
    float mx = -FLT_MAX;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[(int64_t)c * per]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(lp[(int64_t)c * per] - mx);
    const float pr = __expf(lp[(int64_t)cls * per] - mx) / se;
    const int64_t o = (int64_t)(cd[0] + y) * W + cd[1] + x;
    const double q = (double)pr * MF_PROB_FIX + 0.5;
    atomicAdd(sum + o, (pr == pr && q > 0.0) ? (unsigned long long)q : 0ull);      // (a NaN logit contributes 0, not garbage)
    atomicAdd(cnt + o, 1ull);
  }
}
extern "C" int mfnet_tile_prob_accumulate(int64_t* sum, int64_t* cnt, const float* logits, const int* coords, int64_t n, int64_t C, int64_t H, int64_t W,
                                          int64_t h, int64_t w, int64_t cls, hipStream_t stream) {
  MF_CHECK_ARG(sum && cnt && logits && coords && C > 0 && cls >= 0 && cls < C && h > 0 && w > 0, "tile_prob_accumulate: bad args");
  MF_CHECK_ARG(((uintptr_t)sum | (uintptr_t)cnt) % 8 == 0, "tile_prob_accumulate: the 64-bit accumulators must be 8-byte aligned");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(tile_prob_accumulate_kernel, dim3(grid_for(n * h * w)), dim3(256), 0, stream, (unsigned long long*)sum, (unsigned long long*)cnt,
                     logits, coords, n, (int)C, W, (int)h, (int)w, (int)cls);
  MF_LAUNCH_CHECK("tile_prob_accumulate");
  return MFNET_OK;
}
// prob = sum / cnt (0 where no tile covers the pixel) ; mask = prob >= threshold (inference.py:582)
__global__ __launch_bounds__(256) void blend_finish_kernel(const int64_t* __restrict__ sum, const int64_t* __restrict__ cnt, float* __restrict__ prob,
                                                           unsigned char* __restrict__ mask, float thr, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t c = cnt[i];
    const float p = c > 0 ? (float)(((double)sum[i] / MF_PROB_FIX) / (double)c) : 0.f;
    if (prob) prob[i] = p;
    if (mask) mask[i] = p >= thr ? 1 : 0;
  }
}
extern "C" int mfnet_blend_finish(const int64_t* sum, const int64_t* cnt, float* prob, unsigned char* mask, float threshold, int64_t n, hipStream_t stream) {
  MF_CHECK_ARG(sum && cnt && (prob || mask) && n >= 0, "blend_finish: bad args");
  if (n == 0) return MFNET_OK;
  hipLaunchKernelGGL(blend_finish_kernel, dim3(grid_for(n)), dim3(256), 0, stream, sum, cnt, prob, mask, threshold, n);
  MF_LAUNCH_CHECK("blend_finish");
  return MFNET_OK;
}

// ----------------------------------------------------------------------------------------------------------
// BuildingInference._normalize_image (inference.py:444-496) on a whole raster [C,H,W] of uint8 / uint16 / fp32:
//   nodata[y,x] = any_c(in[c,y,x] == nodata_value)                                       (:456-459)
//   RGB:  u = in / 255 (uint8) | in / 65535 (uint16) | clip((in - p2) / (p98 - p2), 0, 1) (fp32; the percentiles
//         come from the host side), then (u - mean[c]) / std[c] when C == 3               (:464-484)
//   DEM:  the raw value as fp32                                                           (:485-488)
//   nodata pixels are set to 0 in every channel                                           (:490-494)
// Divisions are IEEE fp32 divisions as in numpy (no reciprocal), so the result is bit-equal to the reference's.
struct RasterNorm { float sub, div, mean[3], stdv[3]; int scale, clip, imagenet, has_nodata; double nodata; };
template <typename T>
__global__ __launch_bounds__(256) void normalize_raster_kernel(const T* __restrict__ in, float* __restrict__ out, unsigned char* __restrict__ mask,
                                                               int64_t hw, int C, RasterNorm rp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    bool nd = false;
    if (rp.has_nodata)
      for (int c = 0; c < C; ++c) nd = nd || ((double)in[(int64_t)c * hw + i] == rp.nodata);
    if (mask) mask[i] = nd ? 1 : 0;
    for (int c = 0; c < C; ++c) {
      float u = (float)in[(int64_t)c * hw + i];
      if (rp.scale) {
        if (rp.clip) { u = (u - rp.sub) / rp.div; u = u < 0.f ? 0.f : (u > 1.f ? 1.f : u); }   // keeps a NaN, like np.clip
        else u = u / rp.div;
      }
      if (rp.imagenet) u = (u - rp.mean[c]) / rp.stdv[c];
      out[(int64_t)c * hw + i] = nd ? 0.f : u;
    }
  }
}
extern "C" int mfnet_normalize_raster(const void* in, int in_kind, int64_t C, int64_t H, int64_t W, int is_rgb, int clip, float sub,
                                      float div, const float* mean3, const float* std3, int has_nodata, double nodata_value,
                                      float* out, unsigned char* nodata_mask, hipStream_t stream) {
  MF_CHECK_ARG(in && out && C > 0 && C <= 16 && H > 0 && W > 0, "normalize_raster: bad args (1 <= C <= 16)");
  MF_CHECK_ARG(in_kind >= 0 && in_kind <= 2, "normalize_raster: in_kind is 0 (uint8), 1 (uint16) or 2 (fp32)");
  RasterNorm rp{};
  rp.scale = is_rgb ? 1 : 0; rp.clip = clip ? 1 : 0; rp.sub = sub; rp.div = div;
  rp.has_nodata = has_nodata ? 1 : 0; rp.nodata = nodata_value;
  MF_CHECK_ARG(!is_rgb || clip || div != 0.f, "normalize_raster: zero divisor");
  rp.imagenet = (is_rgb && C == 3 && mean3 && std3) ? 1 : 0;
  if (rp.imagenet)
    for (int c = 0; c < 3; ++c) { rp.mean[c] = mean3[c]; rp.stdv[c] = std3[c]; }
  const dim3 grid(grid_for(H * W));
  if (in_kind == 0) hipLaunchKernelGGL(normalize_raster_kernel<unsigned char>, grid, dim3(256), 0, stream, (const unsigned char*)in, out, nodata_mask, H * W, (int)C, rp);
  else if (in_kind == 1) hipLaunchKernelGGL(normalize_raster_kernel<unsigned short>, grid, dim3(256), 0, stream, (const unsigned short*)in, out, nodata_mask, H * W, (int)C, rp);
  else hipLaunchKernelGGL(normalize_raster_kernel<float>, grid, dim3(256), 0, stream, (const float*)in, out, nodata_mask, H * W, (int)C, rp);
  MF_LAUNCH_CHECK("normalize_raster");
  return MFNET_OK;
}

// Per-tile building mask for the CPU vectoriser (inference.py:562-566, 582, 742-744): prob = softmax(logits[t])[cls],
// mask = prob >= threshold over the tile's valid extent (the reflect padding is cropped away: 0 outside).
__global__ __launch_bounds__(256) void tile_prob_mask_kernel(const float* __restrict__ logits, const int* __restrict__ coords, float* __restrict__ prob,
                                                             unsigned char* __restrict__ mask, int64_t n, int C, int h, int w, int cls, float thr) {
  const int64_t per = (int64_t)h * w, total = n * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / per;
    const int64_t r = i - t * per;
    const int y = (int)(r / w), x = (int)(r - (int64_t)y * w);
    const int* cd = coords + 4 * t;
    float pr = 0.f;
    bool inside = y < cd[2] && x < cd[3];
    if (inside) {
      const float* lp = logits + t * C * per + r;
      float mx = -FLT_MAX;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[(int64_t)c * per]);
      float se = 0.f;
      for (int c = 0; c < C; ++c) se += __expf(lp[(int64_t)c * per] - mx);
      pr = __expf(lp[(int64_t)cls * per] - mx) / se;
    }
    if (prob) prob[i] = pr;
    mask[i] = (inside && pr >= thr) ? 1 : 0;
  }
}
extern "C" int mfnet_tile_prob_mask(const float* logits, const int* coords, float* prob, unsigned char* mask, int64_t n, int64_t C,
                                    int64_t h, int64_t w, int64_t cls, float threshold, hipStream_t stream) {
  MF_CHECK_ARG(logits && coords && mask && C > 0 && cls >= 0 && cls < C && h > 0 && w > 0, "tile_prob_mask: bad args");
  if (n <= 0) return MFNET_OK;
  hipLaunchKernelGGL(tile_prob_mask_kernel, dim3(grid_for(n * h * w)), dim3(256), 0, stream, logits, coords, prob, mask, n, (int)C, (int)h, (int)w, (int)cls, threshold);
  MF_LAUNCH_CHECK("tile_prob_mask");
  return MFNET_OK;
}

// confusion matrix cm[gt*n + pred] += 1 over pixels with 0 <= gt < n (utils2.py:168-176: sklearn
// confusion_matrix(gts, predictions, labels=range(n))); pred uint8, gt int64, cm int64 [n, n] (zero first)
__global__ __launch_bounds__(256) void confusion_kernel(const unsigned char* __restrict__ pred, const int64_t* __restrict__ gt, unsigned long long* __restrict__ cm,
                                                        int n, int64_t total) {
  __shared__ unsigned int loc[32 * 32];
  for (int i = threadIdx.x; i < n * n; i += 256) loc[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t g = gt[i];
    const int p = pred[i];
    if (g >= 0 && g < n && p < n) atomicAdd(&loc[(int)g * n + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n * n; i += 256)
    if (loc[i]) atomicAdd(cm + i, (unsigned long long)loc[i]);
}
extern "C" int mfnet_confusion(const unsigned char* pred, const int64_t* gt, int64_t* cm, int64_t n_classes, int64_t total, hipStream_t stream) {
  MF_CHECK_ARG(pred && gt && cm && n_classes > 0 && n_classes <= 32 && total >= 0, "confusion: bad args (n_classes <= 32)");
  if (total == 0) return MFNET_OK;
  hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(total, 1024)), dim3(256), 0, stream, pred, gt, (unsigned long long*)cm, (int)n_classes, total);
  MF_LAUNCH_CHECK("confusion");
  return MFNET_OK;
}

// =====================================================================================================
// Training input pipeline on the device (multiscale_dataset.py:190-215 normalisation, :228-245 geometric
// augmentation): per tile, out = rot90^k(vflip?(hflip?(in))) applied to RGB, DSM and labels alike, RGB uint8 ->
// (v/255 - mean)/std. code[t]: bit 0 horizontal flip (np.flip axis W), bit 1 vertical flip (axis H), bits 2-3 =
// k of np.rot90(axes=(H, W)) (counter-clockwise). The per-tile DSM min-max that follows in the reference
// (:205-208) is mfnet_tile_minmax_norm on dsm_out.
struct PrepParams { float scale[3], shift[3]; };
__global__ __launch_bounds__(256) void prepare_tiles_kernel(const unsigned char* __restrict__ rgb, const float* __restrict__ dsm,
                                                            const int64_t* __restrict__ lab, const int* __restrict__ code,
                                                            float* __restrict__ rgb_out, float* __restrict__ dsm_out,
                                                            int64_t* __restrict__ lab_out, int64_t n, int S, PrepParams pp) {
  const int64_t per = (int64_t)S * S, total = n * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / per;
    const int r = (int)(i - t * per);
    const int y = r / S, x = r - y * S;
    const int c = code[t];
    const int k = (c >> 2) & 3;
    int sy, sx;                       // undo the rotation: np.rot90(m, 1)[i][j] = m[j][S-1-i]
    if (k == 0) { sy = y; sx = x; }
    else if (k == 1) { sy = x; sx = S - 1 - y; }
    else if (k == 2) { sy = S - 1 - y; sx = S - 1 - x; }
    else { sy = S - 1 - x; sx = y; }
    if (c & 2) sy = S - 1 - sy;       // undo the vertical flip
    if (c & 1) sx = S - 1 - sx;       // undo the horizontal flip
    const int64_t src = (int64_t)sy * S + sx;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
      rgb_out[(t * 3 + ch) * per + r] = fmaf((float)rgb[(t * 3 + ch) * per + src], pp.scale[ch], pp.shift[ch]);
    if (dsm) dsm_out[t * per + r] = dsm[t * per + src];
    if (lab) lab_out[t * per + r] = lab[t * per + src];
  }
}
extern "C" int mfnet_prepare_tiles(const unsigned char* rgb, const float* dsm, const int64_t* labels, const int* codes,
                                   float* rgb_out, float* dsm_out, int64_t* labels_out, int64_t n, int64_t S,
                                   const float* mean3, const float* std3, hipStream_t stream) {
  MF_CHECK_ARG(rgb && codes && rgb_out && mean3 && std3 && S > 0 && S < 32768, "prepare_tiles: bad args");
  MF_CHECK_ARG((!dsm || dsm_out) && (!labels || labels_out), "prepare_tiles: an input without its output");
  if (n <= 0) return MFNET_OK;
  PrepParams pp;
  for (int c = 0; c < 3; ++c) {
    MF_CHECK_ARG(std3[c] != 0.f, "prepare_tiles: std must be non-zero");
    pp.scale[c] = 1.0f / (255.0f * std3[c]);
    pp.shift[c] = -mean3[c] / std3[c];
  }
  hipLaunchKernelGGL(prepare_tiles_kernel, dim3(grid_for(n * S * S)), dim3(256), 0, stream, rgb, dsm, labels, codes, rgb_out, dsm_out,
                     labels_out, n, (int)S, pp);
  MF_LAUNCH_CHECK("prepare_tiles");
  return MFNET_OK;
}

// ----------------------------------------------------------------------------------------------------------
// Brightness / contrast jitter of the training loader (multiscale_dataset.py:250-265), on the raw uint8 patches, per tile and
// RGB channel, in the reference's fp32 statement order:
//   v = x * brightness ; m = mean(v) ; v = (v - m) * contrast + m ; v = clip(v, 0, 255) ; x = (uint8) v   (truncation)
// params[t] = (brightness, contrast); brightness <= 0 leaves tile t untouched (the p = 0.3 "no jitter" draw). One workgroup
// per (tile, channel): pass 1 sums the fp32 products in double (exact: <= 2^24 values below 2^9 with 24-bit mantissas), pass 2
// applies. numpy's mean adds the same fp32 products pairwise in fp32; the two means differ in the last fp32 bits, which moves a
// pixel that lands within ~1e-5 of an integer by one grey level (tests bound it: <= 1 level on < 0.1 % of the pixels).
__global__ __launch_bounds__(256) void jitter_tiles_kernel(unsigned char* __restrict__ rgb, const float* __restrict__ params, int64_t per) {
  __shared__ double red[256];
  const int64_t tc = blockIdx.x;                       // tile * 3 + channel
  const float b = params[2 * (tc / 3)], c = params[2 * (tc / 3) + 1];
  if (!(b > 0.f)) return;
  unsigned char* p = rgb + tc * per;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < per; i += 256) s += (double)((float)p[i] * b);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float m = (float)(red[0] / (double)per);
  for (int64_t i = threadIdx.x; i < per; i += 256) {
    float v = (float)p[i] * b;
    v = (v - m) * c + m;
    v = fminf(fmaxf(v, 0.f), 255.f);
    p[i] = (unsigned char)v;
  }
}
extern "C" int mfnet_jitter_tiles(unsigned char* rgb, const float* params, int64_t n, int64_t S, hipStream_t stream) {
  MF_CHECK_ARG(rgb && params && S > 0 && S <= 4096, "jitter_tiles: bad args");
  if (n <= 0) return MFNET_OK;
  MF_CHECK_ARG(n * 3 < (1ll << 31), "jitter_tiles: too many tiles");
  hipLaunchKernelGGL(jitter_tiles_kernel, dim3((unsigned)(n * 3)), dim3(256), 0, stream, rgb, params, S * S);
  MF_LAUNCH_CHECK("jitter_tiles");
  return MFNET_OK;
}
