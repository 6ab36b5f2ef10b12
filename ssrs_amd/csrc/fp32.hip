// fp32-accurate forward path of MFNet (north star: "logits within 1e-4 relative for the fp32 path"; the reference computes
// everything in fp32: unetformer_mmsam.py:533-563, lora_block.py:169-196), and the contractions of fp32 training.
//
//   gemm_f32       out = act(alpha * A.W^T + bias) (+ residual) (+ out): fp32 operands on v_mfma_f32_16x16x4_f32, whose result is
//                  a k-ordered fp32 fma chain (one rounding per product) - the arithmetic F.linear / 1x1 / im2col'ed convs have
//   attn_f32       SAM ViT attention (windowed with zero-padded windows, or global) with the decomposed relative-position
//                  bias, fp32 throughout: one lane = one query, keys / values streamed through LDS, online softmax per key tile
//   attn_f32_bwd   its backward (dqkv), for fp32 training
//   im2col*_f32    fp32 twins of the patch / 3x3 gathers
// GLA's 8x8-window attention already computes in fp32; its fp32-input twin lives next to it in decoder.hip.
#include "common.h"
#include <float.h>

static inline unsigned f32_grid_for(int64_t work_items, int64_t cap = 16384) {
  int64_t b = (work_items + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// =============================================================================== GEMM
#define F_BM 64
#define F_BN 64
#define F_BK 16
#define F_LD 20          // LDS row stride in floats: 16 + 4 keeps float4 stores aligned and spreads rows over banks
struct GemmF32Params {
  const float* A; int64_t lda;
  const float* W; int64_t ldw;
  const float* bias;
  const float* residual; int64_t ldr, res_row_mod;
  float* out; int64_t ldo;
  int64_t M, N, K;
  float alpha;
  int act, accumulate;
};
__device__ __forceinline__ float f32_act(float v, int act) {
  if (act == 1) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));     // nn.GELU() (erf form)
  if (act == 3) return fminf(fmaxf(v, 0.f), 6.f);                               // ReLU6
  if (act == 6) return fmaxf(v, 0.f);                                           // ReLU
  return v;
}
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmF32Params p) {
  __shared__ float As[2][F_BM][F_LD], Ws[2][F_BN][F_LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int64_t m0 = (int64_t)blockIdx.y * F_BM, n0 = (int64_t)blockIdx.x * F_BN;
  const int lr = tid >> 2, lk = (tid & 3) * 4;                  // this thread's row / k offset of the 64x16 operand tiles
  const bool a_row = m0 + lr < p.M, w_row = n0 + lr < p.N;
  const float* ap = p.A + (m0 + lr) * p.lda + lk;
  const float* wp = p.W + (n0 + lr) * p.ldw + lk;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ra = (a_row && lk < p.K) ? *reinterpret_cast<const float4*>(ap) : zero4;     // K % 4 == 0: a float4 is in or out as a whole
  float4 rw = (w_row && lk < p.K) ? *reinterpret_cast<const float4*>(wp) : zero4;
  const int nk = (int)((p.K + F_BK - 1) / F_BK);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    *reinterpret_cast<float4*>(&As[buf][lr][lk]) = ra;
    *reinterpret_cast<float4*>(&Ws[buf][lr][lk]) = rw;
    __syncthreads();                                            // two buffers: the tile written two iterations ago is fully consumed
    const int64_t kn = (int64_t)(kt + 1) * F_BK + lk;
    ra = (a_row && kn < p.K) ? *reinterpret_cast<const float4*>(ap + (int64_t)(kt + 1) * F_BK) : zero4;
    rw = (w_row && kn < p.K) ? *reinterpret_cast<const float4*>(wp + (int64_t)(kt + 1) * F_BK) : zero4;
#pragma unroll
    for (int kk = 0; kk < F_BK / 4; ++kk) {
      const int kc = kk * 4 + (lane >> 4);
      const float a0 = As[buf][wm * 32 + (lane & 15)][kc], a1 = As[buf][wm * 32 + 16 + (lane & 15)][kc];
      const float b0 = Ws[buf][wn * 32 + (lane & 15)][kc], b1 = Ws[buf][wn * 32 + 16 + (lane & 15)][kc];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 32 + j * 16 + (lane & 15);
      if (n >= p.N) continue;
      const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        if (m >= p.M) continue;
        float v = f32_act(fmaf(p.alpha, acc[i][j][r], b), p.act);
        if (p.residual) v += p.residual[(p.res_row_mod ? m % p.res_row_mod : m) * p.ldr + n];
        float* o = p.out + m * p.ldo + n;
        *o = p.accumulate ? *o + v : v;
      }
    }
}
extern "C" int mfnet_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, int64_t M, int64_t N, int64_t K, const float* bias,
                              float alpha, int act, const float* residual, int64_t ldr, int64_t res_row_mod, float* out, int64_t ldo,
                              int accumulate, hipStream_t stream) {
  MF_CHECK_ARG(A && W && out, "gemm_f32: null pointer");
  MF_CHECK_ARG(M >= 0 && N > 0 && K > 0 && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K && ldo >= N,
               "gemm_f32: M=%lld N=%lld K=%lld lda=%lld ldw=%lld ldo=%lld (K and the operand row strides must be multiples of 4)",
               (long long)M, (long long)N, (long long)K, (long long)lda, (long long)ldw, (long long)ldo);
  MF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "gemm_f32: operands must be 16-byte aligned");
  MF_CHECK_ARG(act == 0 || act == 1 || act == 3 || act == 6, "gemm_f32: act %d (0 none, 1 gelu, 3 relu6, 6 relu)", act);
  MF_CHECK_ARG(!residual || ldr >= N, "gemm_f32: residual row stride %lld < N", (long long)ldr);
  if (M == 0) return MFNET_OK;
  const int64_t gy = ceil_div64(M, F_BM), gx = ceil_div64(N, F_BN);
  MF_CHECK_ARG(gy <= 65535, "gemm_f32: M=%lld too large for one launch", (long long)M);
  GemmF32Params p{A, lda, W, ldw, bias, residual, ldr, res_row_mod, out, ldo, M, N, K, alpha, act, accumulate};
  hipLaunchKernelGGL(gemm_f32_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, p);
  MF_LAUNCH_CHECK("gemm_f32");
  return MFNET_OK;
}

// =============================================================================== SAM ViT attention, fp32
// qkv [G*Sh*Sw, 3*heads*HD] fp32 (bias and LoRA already added), channel = (which*heads + head)*HD + d (lora_block.py:175-179).
// Windowed blocks (ws > 0): the map is zero-padded to a multiple of ws AFTER norm1 (lora_block.py:199-220), so a padded token
// enters as key / value with qkv = the layer's bias; queries at padded positions are dropped by window_unpartition.
// Scores use the SCALED q, the decomposed relative-position bias the UNSCALED q (lora_block.py:181-193, 281-317):
//   s(q, k) = scale * q.k + q.Th[qy - ky + kh - 1] + q.Tw[qx - kx + kw - 1],   Th [2kh-1, HD], Tw [2kw-1, HD] (resized tables).
#define AF_KMAX 64       // window / map side up to 64 tokens (1024 px input)
struct AttnF32Params {
  const float* qkv; const float* qkv_bias; const float* th; const float* tw; float* out;
  int G, Sh, Sw, heads, kh, kw, nwh, nww; float scale;
};
template <int HD>
__global__ __launch_bounds__(64) void attn_f32_kernel(const AttnF32Params p) {
  __shared__ float kt[64][HD], vt[64][HD];
  __shared__ float bh[64][AF_KMAX + 1], bw[64][AF_KMAX + 1];
  const int lane = threadIdx.x;
  const int head = blockIdx.x % p.heads;
  const int win = blockIdx.x / p.heads;
  const int g = win / (p.nwh * p.nww), wy = (win / p.nww) % p.nwh, wx = win % p.nww;
  const int nk = p.kh * p.kw;
  const int t = blockIdx.y * 64 + lane;
  const int ty = t / p.kw, tx = t % p.kw;
  const int qy = wy * p.kh + ty, qx = wx * p.kw + tx;
  const bool live = t < nk && qy < p.Sh && qx < p.Sw;
  const int64_t C3 = 3ll * p.heads * HD;
  float q[HD], acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { q[d] = 0.f; acc[d] = 0.f; }
  if (live) {
    const float* qp = p.qkv + (((int64_t)g * p.Sh + qy) * p.Sw + qx) * C3 + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 v = *reinterpret_cast<const float4*>(qp + d);
      q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
    }
    for (int j = 0; j < p.kh; ++j) {
      const float* r = p.th + (int64_t)(ty - j + p.kh - 1) * HD;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) a = fmaf(q[d], r[d], a);
      bh[lane][j] = a;
    }
    for (int j = 0; j < p.kw; ++j) {
      const float* r = p.tw + (int64_t)(tx - j + p.kw - 1) * HD;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) a = fmaf(q[d], r[d], a);
      bw[lane][j] = a;
    }
  }
  float mx = -FLT_MAX, l = 0.f;
  constexpr int V4 = HD / 4;                                     // float4s per key row
  for (int j0 = 0; j0 < nk; j0 += 64) {
    __syncthreads();                                             // previous key tile fully consumed
    for (int i = lane; i < 64 * V4; i += 64) {
      const int kj = i / V4, part = (i % V4) * 4;
      const int j = j0 + kj;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (j < nk) {
        const int ky = wy * p.kh + j / p.kw, kx = wx * p.kw + j % p.kw;
        if (ky < p.Sh && kx < p.Sw) {
          const float* base = p.qkv + (((int64_t)g * p.Sh + ky) * p.Sw + kx) * C3 + head * HD + part;
          kv = *reinterpret_cast<const float4*>(base + (int64_t)p.heads * HD);
          vv = *reinterpret_cast<const float4*>(base + 2ll * p.heads * HD);
        } else {                                                 // zero-padded token: its qkv row is the bias
          const float* base = p.qkv_bias + head * HD + part;
          kv = *reinterpret_cast<const float4*>(base + (int64_t)p.heads * HD);
          vv = *reinterpret_cast<const float4*>(base + 2ll * p.heads * HD);
        }
      }
      *reinterpret_cast<float4*>(&kt[kj][part]) = kv;
      *reinterpret_cast<float4*>(&vt[kj][part]) = vv;
    }
    __syncthreads();
    if (!live) continue;
    const int nj = min(64, nk - j0);
    for (int j1 = 0; j1 < nj; j1 += 16) {                        // 16 keys at a time: scores in registers, one rescale per group
      float s[16];
      float tmx = -FLT_MAX;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = j1 + jj;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], kt[j][d], a);
        const int jg = min(j0 + j, nk - 1);
        a = a * p.scale + bh[lane][jg / p.kw] + bw[lane][jg % p.kw];
        a = j < nj ? a : -FLT_MAX;
        s[jj] = a;
        tmx = fmaxf(tmx, a);
      }
      const float nm = fmaxf(mx, tmx);
      const float corr = expf(mx - nm);                          // first group: exp(-FLT_MAX - nm) = 0, acc and l are 0 anyway
      l *= corr;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] *= corr;
      mx = nm;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const int j = j1 + jj;
        const float e = j < nj ? expf(s[jj] - mx) : 0.f;
        l += e;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = fmaf(e, vt[j][d], acc[d]);
      }
    }
  }
  if (!live) return;
  const float inv = 1.f / l;
  float* op = p.out + (((int64_t)g * p.Sh + qy) * p.Sw + qx) * ((int64_t)p.heads * HD) + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4)
    *reinterpret_cast<float4*>(op + d) = make_float4(acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv);
}
extern "C" int mfnet_attn_f32(const float* qkv, const float* qkv_bias, const float* th, const float* tw, float* out, int64_t G, int64_t Sh,
                              int64_t Sw, int64_t heads, int64_t hd, int64_t ws, float scale, hipStream_t stream) {
  MF_CHECK_ARG(qkv && qkv_bias && th && tw && out, "attn_f32: null pointer");
  MF_CHECK_ARG(G > 0 && Sh > 0 && Sw > 0 && heads > 0 && ws >= 0, "attn_f32: bad geometry");
  const int64_t kh = ws > 0 ? ws : Sh, kw = ws > 0 ? ws : Sw;
  MF_CHECK_ARG(kh <= AF_KMAX && kw <= AF_KMAX, "attn_f32: window / map side %lldx%lld exceeds %d tokens", (long long)kh, (long long)kw, AF_KMAX);
  MF_CHECK_ARG(hd == 64 || hd == 80 || hd == 32, "attn_f32: head dim %lld (32, 64 and 80 are built)", (long long)hd);
  AttnF32Params p;
  p.qkv = qkv; p.qkv_bias = qkv_bias; p.th = th; p.tw = tw; p.out = out;
  p.G = (int)G; p.Sh = (int)Sh; p.Sw = (int)Sw; p.heads = (int)heads; p.kh = (int)kh; p.kw = (int)kw;
  p.nwh = (int)ceil_div64(Sh, kh); p.nww = (int)ceil_div64(Sw, kw); p.scale = scale;
  const int64_t gx = G * p.nwh * p.nww * heads, gy = ceil_div64(kh * kw, 64);
  MF_CHECK_ARG(gx < (1ll << 31) && gy <= 65535, "attn_f32: too many windows");
  const dim3 grid((unsigned)gx, (unsigned)gy), block(64);
  if (hd == 64) hipLaunchKernelGGL(attn_f32_kernel<64>, grid, block, 0, stream, p);
  else if (hd == 80) hipLaunchKernelGGL(attn_f32_kernel<80>, grid, block, 0, stream, p);
  else hipLaunchKernelGGL(attn_f32_kernel<32>, grid, block, 0, stream, p);
  MF_LAUNCH_CHECK("attn_f32");
  return MFNET_OK;
}

// =============================================================================== SAM ViT attention backward, fp32
// (fp32 training: the reference trains in fp32, trainer.py:146-159.) Same geometry as attn_f32 above, rel-pos tables and the qkv bias
// frozen (MFNet trains only the PEFT tensors: unetformer_mmsam.py:522-528), so the gradient is dqkv alone.
//   dq kernel  (lane = query): sweep 1 over the window's keys -> row max / sum; sweep 2 -> p, dp = dO.v, ds = p (dp - delta),
//              dq += ds (scale k + Th[.] + Tw[.]); leaves lse, delta and the query's bias rows (q.Th[*], q.Tw[*]) in the workspace
//   dkv kernel (lane = key): the window's queries streamed through LDS 32 at a time; dk += scale ds q, dv += p dO
// Padded window positions have no row in memory: their keys enter the softmax (forward), their gradients are dropped.
struct AttnF32BwdParams {
  const float* qkv; const float* qkv_bias; const float* th; const float* tw; const float* out; const float* dout; float* dqkv;
  float* lse; float* delta; float* wbh; float* wbw;          // workspace: [slots], [slots], [slots][kh], [slots][kw]
  int G, Sh, Sw, heads, kh, kw, nwh, nww, nkp; float scale;
};
template <int HD>
__global__ __launch_bounds__(64) void attn_f32_bwd_dq_kernel(const AttnF32BwdParams p) {
  __shared__ float kt[64][HD], vt[64][HD];
  __shared__ float bh[64][AF_KMAX + 1], bw[64][AF_KMAX + 1];
  const int lane = threadIdx.x;
  const int head = blockIdx.x % p.heads;
  const int win = blockIdx.x / p.heads;
  const int g = win / (p.nwh * p.nww), wy = (win / p.nww) % p.nwh, wx = win % p.nww;
  const int nk = p.kh * p.kw;
  const int t = blockIdx.y * 64 + lane;
  const int ty = t / p.kw, tx = t % p.kw;
  const int qy = wy * p.kh + ty, qx = wx * p.kw + tx;
  const bool live = t < nk && qy < p.Sh && qx < p.Sw;
  const int64_t C3 = 3ll * p.heads * HD, C1 = (int64_t)p.heads * HD;
  const int64_t slot = (int64_t)blockIdx.x * p.nkp + t;
  float q[HD], dO[HD], dq[HD];
  float delta = 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) { q[d] = 0.f; dO[d] = 0.f; dq[d] = 0.f; }
  if (live) {
    const int64_t row = ((int64_t)g * p.Sh + qy) * p.Sw + qx;
    const float* qp = p.qkv + row * C3 + head * HD;
    const float* op = p.out + row * C1 + head * HD;
    const float* dp = p.dout + row * C1 + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 v = *reinterpret_cast<const float4*>(qp + d);
      const float4 o = *reinterpret_cast<const float4*>(op + d);
      const float4 e = *reinterpret_cast<const float4*>(dp + d);
      q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
      dO[d] = e.x; dO[d + 1] = e.y; dO[d + 2] = e.z; dO[d + 3] = e.w;
      delta = fmaf(o.x, e.x, fmaf(o.y, e.y, fmaf(o.z, e.z, fmaf(o.w, e.w, delta))));
    }
    for (int j = 0; j < p.kh; ++j) {
      const float* r = p.th + (int64_t)(ty - j + p.kh - 1) * HD;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) a = fmaf(q[d], r[d], a);
      bh[lane][j] = a;
      p.wbh[slot * p.kh + j] = a;
    }
    for (int j = 0; j < p.kw; ++j) {
      const float* r = p.tw + (int64_t)(tx - j + p.kw - 1) * HD;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) a = fmaf(q[d], r[d], a);
      bw[lane][j] = a;
      p.wbw[slot * p.kw + j] = a;
    }
  }
  float mx = -FLT_MAX, l = 0.f, inv = 0.f;
  constexpr int V4 = HD / 4;
  for (int sweep = 0; sweep < 2; ++sweep) {
    for (int j0 = 0; j0 < nk; j0 += 64) {
      __syncthreads();
      for (int i = lane; i < 64 * V4; i += 64) {
        const int kj = i / V4, part = (i % V4) * 4;
        const int j = j0 + kj;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (j < nk) {
          const int ky = wy * p.kh + j / p.kw, kx = wx * p.kw + j % p.kw;
          const float* base = (ky < p.Sh && kx < p.Sw) ? p.qkv + (((int64_t)g * p.Sh + ky) * p.Sw + kx) * C3 + head * HD + part
                                                       : p.qkv_bias + head * HD + part;
          kv = *reinterpret_cast<const float4*>(base + C1);
          vv = *reinterpret_cast<const float4*>(base + 2 * C1);
        }
        *reinterpret_cast<float4*>(&kt[kj][part]) = kv;
        *reinterpret_cast<float4*>(&vt[kj][part]) = vv;
      }
      __syncthreads();
      if (!live) continue;
      const int nj = min(64, nk - j0);
      for (int j = 0; j < nj; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], kt[j][d], a);
        const int jg = j0 + j, jy = jg / p.kw, jx = jg % p.kw;
        a = a * p.scale + bh[lane][jy] + bw[lane][jx];
        if (sweep == 0) {
          const float nm = fmaxf(mx, a);
          l = l * expf(mx - nm) + expf(a - nm);
          mx = nm;
        } else {
          const float pr = expf(a - mx) * inv;
          float dpv = 0.f;
#pragma unroll
          for (int d = 0; d < HD; ++d) dpv = fmaf(dO[d], vt[j][d], dpv);
          const float ds = pr * (dpv - delta);
          const float* rh = p.th + (int64_t)(ty - jy + p.kh - 1) * HD;
          const float* rw = p.tw + (int64_t)(tx - jx + p.kw - 1) * HD;
          const float dss = ds * p.scale;
#pragma unroll
          for (int d = 0; d < HD; ++d) dq[d] = fmaf(dss, kt[j][d], fmaf(ds, rh[d] + rw[d], dq[d]));
        }
      }
    }
    if (sweep == 0) inv = live ? 1.f / l : 0.f;
  }
  if (!live) return;
  p.lse[slot] = mx + logf(l);
  p.delta[slot] = delta;
  float* o = p.dqkv + (((int64_t)g * p.Sh + qy) * p.Sw + qx) * C3 + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) *reinterpret_cast<float4*>(o + d) = make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]);
}
template <int HD>
__global__ __launch_bounds__(64) void attn_f32_bwd_dkv_kernel(const AttnF32BwdParams p) {
  __shared__ float ql[32][HD], dl[32][HD];
  __shared__ float bh[32][AF_KMAX + 1], bw[32][AF_KMAX + 1];
  __shared__ float lsel[32], dell[32];
  __shared__ int livel[32];
  const int lane = threadIdx.x;
  const int head = blockIdx.x % p.heads;
  const int win = blockIdx.x / p.heads;
  const int g = win / (p.nwh * p.nww), wy = (win / p.nww) % p.nwh, wx = win % p.nww;
  const int nk = p.kh * p.kw;
  const int t = blockIdx.y * 64 + lane;
  const int ty = t / p.kw, tx = t % p.kw;
  const int ky = wy * p.kh + ty, kx = wx * p.kw + tx;
  const bool live = t < nk && ky < p.Sh && kx < p.Sw;
  const int64_t C3 = 3ll * p.heads * HD, C1 = (int64_t)p.heads * HD;
  float k[HD], v[HD], dk[HD], dv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
  const int64_t krow = ((int64_t)g * p.Sh + ky) * p.Sw + kx;
  if (live) {
    const float* base = p.qkv + krow * C3 + head * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const float4 a = *reinterpret_cast<const float4*>(base + C1 + d), b = *reinterpret_cast<const float4*>(base + 2 * C1 + d);
      k[d] = a.x; k[d + 1] = a.y; k[d + 2] = a.z; k[d + 3] = a.w;
      v[d] = b.x; v[d + 1] = b.y; v[d + 2] = b.z; v[d + 3] = b.w;
    }
  }
  constexpr int V4 = HD / 4;
  for (int i0 = 0; i0 < nk; i0 += 32) {
    __syncthreads();
    for (int i = lane; i < 32 * V4; i += 64) {
      const int qi = i / V4, part = (i % V4) * 4;
      const int tq = i0 + qi;
      const int qy = wy * p.kh + tq / p.kw, qx = wx * p.kw + tq % p.kw;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (tq < nk && qy < p.Sh && qx < p.Sw) {
        const int64_t row = ((int64_t)g * p.Sh + qy) * p.Sw + qx;
        a = *reinterpret_cast<const float4*>(p.qkv + row * C3 + head * HD + part);
        b = *reinterpret_cast<const float4*>(p.dout + row * C1 + head * HD + part);
      }
      *reinterpret_cast<float4*>(&ql[qi][part]) = a;
      *reinterpret_cast<float4*>(&dl[qi][part]) = b;
    }
    if (lane < 32) {
      const int tq = i0 + lane;
      const int qy = wy * p.kh + tq / p.kw, qx = wx * p.kw + tq % p.kw;
      const bool lq = tq < nk && qy < p.Sh && qx < p.Sw;
      const int64_t slot = (int64_t)blockIdx.x * p.nkp + tq;
      livel[lane] = lq ? 1 : 0;
      lsel[lane] = lq ? p.lse[slot] : 0.f;
      dell[lane] = lq ? p.delta[slot] : 0.f;
    }
    for (int i = lane; i < 32 * (p.kh + p.kw); i += 64) {
      const int qi = i / (p.kh + p.kw), c = i % (p.kh + p.kw);
      const int tq = i0 + qi;
      const int qy = wy * p.kh + tq / p.kw, qx = wx * p.kw + tq % p.kw;
      const bool lq = tq < nk && qy < p.Sh && qx < p.Sw;
      const int64_t slot = (int64_t)blockIdx.x * p.nkp + tq;
      if (c < p.kh) bh[qi][c] = lq ? p.wbh[slot * p.kh + c] : 0.f;
      else bw[qi][c - p.kh] = lq ? p.wbw[slot * p.kw + (c - p.kh)] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    for (int qi = 0; qi < 32; ++qi) {
      if (!livel[qi]) continue;
      float a = 0.f, dpv = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { a = fmaf(ql[qi][d], k[d], a); dpv = fmaf(dl[qi][d], v[d], dpv); }
      a = a * p.scale + bh[qi][ty] + bw[qi][tx];
      const float pr = expf(a - lsel[qi]);
      const float ds = pr * (dpv - dell[qi]) * p.scale;
#pragma unroll
      for (int d = 0; d < HD; ++d) { dk[d] = fmaf(ds, ql[qi][d], dk[d]); dv[d] = fmaf(pr, dl[qi][d], dv[d]); }
    }
  }
  if (!live) return;
  float* o = p.dqkv + krow * C3 + head * HD;
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    *reinterpret_cast<float4*>(o + C1 + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
    *reinterpret_cast<float4*>(o + 2 * C1 + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
  }
}
static int64_t attn_f32_bwd_ws(int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t ws) {
  const int64_t kh = ws > 0 ? ws : Sh, kw = ws > 0 ? ws : Sw;
  const int64_t slots = G * ceil_div64(Sh, kh) * ceil_div64(Sw, kw) * heads * (ceil_div64(kh * kw, 64) * 64);
  return slots * (2 + kh + kw);
}
extern "C" int64_t mfnet_attn_f32_bwd_workspace_floats(int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t ws) {
  return (G > 0 && Sh > 0 && Sw > 0 && heads > 0 && ws >= 0) ? attn_f32_bwd_ws(G, Sh, Sw, heads, ws) : 0;
}
extern "C" int mfnet_attn_f32_bwd(const float* qkv, const float* qkv_bias, const float* th, const float* tw, const float* out, const float* dout,
                                  float* dqkv, int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws, float scale,
                                  float* workspace, int64_t workspace_floats, hipStream_t stream) {
  MF_CHECK_ARG(qkv && qkv_bias && th && tw && out && dout && dqkv && workspace, "attn_f32_bwd: null pointer");
  MF_CHECK_ARG(G > 0 && Sh > 0 && Sw > 0 && heads > 0 && ws >= 0, "attn_f32_bwd: bad geometry");
  const int64_t kh = ws > 0 ? ws : Sh, kw = ws > 0 ? ws : Sw;
  MF_CHECK_ARG(kh <= AF_KMAX && kw <= AF_KMAX, "attn_f32_bwd: window / map side %lldx%lld exceeds %d tokens", (long long)kh, (long long)kw, AF_KMAX);
  MF_CHECK_ARG(hd == 64 || hd == 80 || hd == 32, "attn_f32_bwd: head dim %lld (32, 64 and 80 are built)", (long long)hd);
  MF_CHECK_ARG(workspace_floats >= attn_f32_bwd_ws(G, Sh, Sw, heads, ws), "attn_f32_bwd: workspace of %lld floats, need %lld",
               (long long)workspace_floats, (long long)attn_f32_bwd_ws(G, Sh, Sw, heads, ws));
  AttnF32BwdParams p;
  p.qkv = qkv; p.qkv_bias = qkv_bias; p.th = th; p.tw = tw; p.out = out; p.dout = dout; p.dqkv = dqkv;
  p.G = (int)G; p.Sh = (int)Sh; p.Sw = (int)Sw; p.heads = (int)heads; p.kh = (int)kh; p.kw = (int)kw;
  p.nwh = (int)ceil_div64(Sh, kh); p.nww = (int)ceil_div64(Sw, kw); p.scale = scale;
  const int64_t gx = G * p.nwh * p.nww * heads, gy = ceil_div64(kh * kw, 64);
  MF_CHECK_ARG(gx < (1ll << 31) && gy <= 65535, "attn_f32_bwd: too many windows");
  p.nkp = (int)(gy * 64);
  const int64_t slots = gx * p.nkp;
  p.lse = workspace; p.delta = workspace + slots; p.wbh = workspace + 2 * slots; p.wbw = p.wbh + slots * kh;
  const dim3 grid((unsigned)gx, (unsigned)gy), block(64);
  if (hd == 64) {
    hipLaunchKernelGGL(attn_f32_bwd_dq_kernel<64>, grid, block, 0, stream, p);
    hipLaunchKernelGGL(attn_f32_bwd_dkv_kernel<64>, grid, block, 0, stream, p);
  } else if (hd == 80) {
    hipLaunchKernelGGL(attn_f32_bwd_dq_kernel<80>, grid, block, 0, stream, p);
    hipLaunchKernelGGL(attn_f32_bwd_dkv_kernel<80>, grid, block, 0, stream, p);
  } else {
    hipLaunchKernelGGL(attn_f32_bwd_dq_kernel<32>, grid, block, 0, stream, p);
    hipLaunchKernelGGL(attn_f32_bwd_dkv_kernel<32>, grid, block, 0, stream, p);
  }
  MF_LAUNCH_CHECK("attn_f32_bwd");
  return MFNET_OK;
}

// =============================================================================== gathers, fp32
// im2col_patch: as mfnet_im2col_patch with fp32 columns. One thread = 4 consecutive px.
__global__ __launch_bounds__(256) void im2col_patch_f32_kernel(const float* __restrict__ rgb, const float* __restrict__ dsm, float* __restrict__ cols,
                                                               int B, int H, int W, int P, int dsm_ch) {
  const int Sh = H / P, Sw = W / P;
  const int K = 3 * P * P, K4 = K >> 2, P4 = P >> 2;
  const int64_t total = (int64_t)2 * B * Sh * Sw * K4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int k4 = (int)(i % K4);
    const int64_t row = i / K4;
    const int px4 = k4 % P4, py = (k4 / P4) % P, c = k4 / (P4 * P);
    const int pw = (int)(row % Sw), ph = (int)((row / Sw) % Sh);
    const int b = (int)((row / ((int64_t)Sw * Sh)) % B), mod = (int)(row / ((int64_t)Sw * Sh * B));
    const int y = ph * P + py, x = pw * P + px4 * 4;
    const float* src = mod == 0 ? rgb + (((int64_t)b * 3 + c) * H + y) * W + x
                                : dsm + (((int64_t)b * dsm_ch + (dsm_ch == 3 ? c : 0)) * H + y) * W + x;
    *reinterpret_cast<float4*>(cols + row * K + (int64_t)k4 * 4) = *reinterpret_cast<const float4*>(src);
  }
}
extern "C" int mfnet_im2col_patch_f32(const float* rgb, const float* dsm, int dsm_channels, float* cols, int64_t B, int64_t H, int64_t W,
                                      int64_t P, hipStream_t stream) {
  MF_CHECK_ARG(rgb && dsm && cols, "im2col_patch_f32: null pointer");
  MF_CHECK_ARG(dsm_channels == 1 || dsm_channels == 3, "im2col_patch_f32: the second modality has 1 or 3 channels, not %d", dsm_channels);
  MF_CHECK_ARG(P > 0 && P % 4 == 0 && H % P == 0 && W % P == 0 && W % 4 == 0, "im2col_patch_f32: bad geometry H=%lld W=%lld P=%lld",
               (long long)H, (long long)W, (long long)P);
  if (B <= 0) return MFNET_OK;
  const int64_t total = 2 * B * (H / P) * (W / P) * (3 * P * P / 4);
  hipLaunchKernelGGL(im2col_patch_f32_kernel, dim3(f32_grid_for(total)), dim3(256), 0, stream, rgb, dsm, cols, (int)B, (int)H, (int)W, (int)P, dsm_channels);
  MF_LAUNCH_CHECK("im2col_patch_f32");
  return MFNET_OK;
}
// x [G,H,W,C] fp32 -> cols [G*H*W, 9*C] fp32, k = tap*C + c, tap = ky*3+kx, zero padding 1.
__global__ __launch_bounds__(256) void im2col3x3_f32_kernel(const float* __restrict__ x, float* __restrict__ cols, int G, int H, int W, int C) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)G * H * W * 9 * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const int tap = (int)((i / C4) % 9);
    const int64_t pix = i / (9 * C4);
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const int64_t g = pix / ((int64_t)W * H);
    const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sy >= 0 && sy < H && sx >= 0 && sx < W) v = *reinterpret_cast<const float4*>(x + ((g * H + sy) * W + sx) * C + c4 * 4);
    *reinterpret_cast<float4*>(cols + pix * 9 * C + (int64_t)tap * C + c4 * 4) = v;
  }
}
extern "C" int mfnet_im2col3x3_f32(const float* x, float* cols, int64_t G, int64_t H, int64_t W, int64_t C, hipStream_t stream) {
  MF_CHECK_ARG(x && cols && G > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "im2col3x3_f32: bad args");
  hipLaunchKernelGGL(im2col3x3_f32_kernel, dim3(f32_grid_for(G * H * W * 9 * (C / 4))), dim3(256), 0, stream, x, cols, (int)G, (int)H, (int)W, (int)C);
  MF_LAUNCH_CHECK("im2col3x3_f32");
  return MFNET_OK;
}
