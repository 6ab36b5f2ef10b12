// Fused attention forward for the SAM ViT blocks: windowed (14x14, padded keys synthesised from the
// qkv bias) and global, with decomposed relative-position bias, flash-style online softmax.
// See attn_common.h for the reference semantics (lora_block.py:169-196, 248-317).
//
// Workgroup = 4 waves x 16 queries of one (image-modality, head, window); keys are staged in LDS in
// chunks of 256: K row-major [key][64] (XOR-swizzled 16-B chunks, ds_read_b128 fragments) and V
// transposed [hd][key] so that P.V contracts over a register-contiguous key index.
// Scores are computed "swapped" (S^T = K.Q^T: key on the accumulator rows, query on the lane), so
//   * softmax statistics are per-lane scalars (4 lane groups combined with two shuffles),
//   * the bf16 P^T accumulator registers ARE the B operand of O^T = V^T.P^T (keys permuted identically
//     in the V^T fragment read), no LDS round trip.
#include "attn_common.h"

struct AttnFwdParams {
  AttnGeom g;
  const bf16_t* qkv;      // [G*T, 3D]
  const float* qkv_bias;  // [3D] fp32 (padded keys / values)
  const bf16_t* tbl;      // [NT, 64] bf16, rows >= Lh+Lw are zero
  bf16_t* out;            // [G*T, D]
  float* lse;             // [G, heads, T]
};

__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int T = g.Sh * g.Sw;
  const int nwin = g.nWh * g.nWw;
  const int gi = blockIdx.z / nwin, win = blockIdx.z - gi * nwin;
  const int wi = win / g.nWw, wj = win - wi * g.nWw;
  const int head = blockIdx.y;
  const int nq = num_queries(g, wi, wj);
  const int qt0 = blockIdx.x * ATT_QT;
  if (qt0 >= nq) return;
  const int Nk = g.KH * g.KW;
  const int TLD = g.NT + 4;

  unsigned char* K_lds = smem;                                     // [256][64] bf16, swizzled
  bf16_t* Vt_lds = (bf16_t*)(smem + ATT_KC * ATT_HD * 2);          // [64][ATT_VT_LD]
  int* kidx_lds = (int*)(smem + ATT_KC * ATT_HD * 2 + ATT_HD * ATT_VT_LD * 2);   // [256] packed (idx_h | idx_w<<16)
  float* T_lds = (float*)(kidx_lds + ATT_KC);                      // [64][NT+4]

  const int64_t row_base = (int64_t)gi * T;
  const int64_t ld = 3 * (int64_t)g.D;

  // ---- this lane's query (column of every accumulator tile)
  const QueryPos qp = query_pos(g, wi, wj, qt0 + wave * 16 + fr);
  bf16x8_t qf[2];
  {
    const bf16_t* qptr = p.qkv + (row_base + qp.tok) * ld + head * ATT_HD + 8 * fq;
    qf[0] = *reinterpret_cast<const bf16x8_t*>(qptr);
    qf[1] = *reinterpret_cast<const bf16x8_t*>(qptr + 32);
    if (!qp.valid) { qf[0] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0}; qf[1] = qf[0]; }
  }
  // ---- T^T = tbl . Q^T  -> T_lds[query][table row]
  float* Trow = T_lds + (wave * 16 + fr) * TLD;
  att_table_product(p.tbl, g.NT, qf, Trow, fr, fq);
  const int qoff_h = qp.qh, qoff_w = qp.qw;

  float m_run = -INFINITY, l_run = 0.f;
  f32x4_t o[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) o[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // this thread always stages 16-B chunk `ch` of a key row: the padded-key value of that chunk (the qkv bias,
  // rounded to bf16 like every other qkv entry) is kept in registers
  const int ch = tid & 7;
  const PaddedKV pad = att_padded_kv(p.qkv_bias, g.D, head, ch);

  for (int kc = 0; kc < Nk; kc += ATT_KC) {
    // ---- stage K (row-major, swizzled), V^T and the per-key bias indices. All global loads of the chunk
    // are issued first, unconditionally (padded / out-of-range keys read row 0 and are replaced afterwards):
    // one memory round trip per chunk, overlapping the other waves' tail of the previous chunk.
    u32x4_t kv[ATT_KC / 32], vv[ATT_KC / 32];
    int kmeta[ATT_KC / 32];
#pragma unroll
    for (int it = 0; it < ATT_KC / 32; ++it) {
      const int kg = kc + it * 32 + (tid >> 3);
      int kh = 0, kw = 0;
      const int tok = kg < Nk ? key_token(g, wi, wj, kg, &kh, &kw) : -2;
      const bf16_t* src = p.qkv + (row_base + max(tok, 0)) * ld + g.D + head * ATT_HD + ch * 8;
      kv[it] = *reinterpret_cast<const u32x4_t*>(src);
      vv[it] = *reinterpret_cast<const u32x4_t*>(src + g.D);
      kmeta[it] = (tok < 0 ? (tok == -2 ? 2 : 1) : 0) | (((g.KH - 1 - kh) | ((g.Lh + g.KW - 1 - kw) << 16)) << 2);
    }
    __syncthreads();   // previous chunk fully consumed (also orders T_lds writes on the first pass)
#pragma unroll
    for (int it = 0; it < ATT_KC / 32; ++it) {
      const int kl = it * 32 + (tid >> 3);
      const int kind = kmeta[it] & 3;          // 0 real, 1 padded, 2 beyond the window
      u32x4_t k4 = kind == 0 ? kv[it] : pad.k, v4 = kind == 0 ? vv[it] : pad.v;
      if (kind == 2) { k4 = (u32x4_t){0u, 0u, 0u, 0u}; v4 = k4; }
      *reinterpret_cast<u32x4_t*>(K_lds + kl * 128 + ((ch ^ kswz(kl)) << 4)) = k4;
      bf16_t* vt = Vt_lds + (ch * 8) * ATT_VT_LD + kl;
      vt[0 * ATT_VT_LD] = (bf16_t)(v4.x & 0xffff); vt[1 * ATT_VT_LD] = (bf16_t)(v4.x >> 16);
      vt[2 * ATT_VT_LD] = (bf16_t)(v4.y & 0xffff); vt[3 * ATT_VT_LD] = (bf16_t)(v4.y >> 16);
      vt[4 * ATT_VT_LD] = (bf16_t)(v4.z & 0xffff); vt[5 * ATT_VT_LD] = (bf16_t)(v4.z >> 16);
      vt[6 * ATT_VT_LD] = (bf16_t)(v4.w & 0xffff); vt[7 * ATT_VT_LD] = (bf16_t)(v4.w >> 16);
      if (ch == 0) kidx_lds[kl] = kmeta[it] >> 2;
    }
    __syncthreads();
    const int kvalid = min(ATT_KC, Nk - kc);
    for (int kt = 0; kt * 64 < kvalid; ++kt) {
      // ---- S^T tiles (16 keys x 16 queries each)
      f32x4_t s[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        const int kr = kt * 64 + 16 * t + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(K_lds + kr * 128 + (((fq + 4 * ks) ^ kswz(kr)) << 4));
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[t], 0, 0, 0);
        }
      }
      // ---- scale + rel-pos bias + mask, tile max
      float tmax = -INFINITY;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = kt * 64 + 16 * t + 4 * fq + r;
          const int pk = kidx_lds[kl];
          const float bias = Trow[(pk & 0xffff) + qoff_h] + Trow[(pk >> 16) + qoff_w];
          float v = s[t][r] * g.scale + bias;
          v = (kl < kvalid) ? v : -INFINITY;
          s[t][r] = v;
          tmax = fmaxf(tmax, v);
        }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __expf(m_run - m_new);    // m_run = -inf on the first tile -> 0
      m_run = m_new;
      float psum = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(s[t][r] - m_new);
          s[t][r] = e;
          psum += e;
        }
      l_run = l_run * alpha + psum;
#pragma unroll
      for (int u = 0; u < 4; ++u) { o[u][0] *= alpha; o[u][1] *= alpha; o[u][2] *= alpha; o[u][3] *= alpha; }
      // ---- O^T += V^T . P^T ; element j of k-step s2 <-> key 32*s2 + 16*(j>>2) + 4*fq + (j&3)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        union { bf16x8_t v; unsigned int w[4]; } pf;
        pf.w[0] = pack_bf16x2(s[2 * s2][0], s[2 * s2][1]);
        pf.w[1] = pack_bf16x2(s[2 * s2][2], s[2 * s2][3]);
        pf.w[2] = pack_bf16x2(s[2 * s2 + 1][0], s[2 * s2 + 1][1]);
        pf.w[3] = pack_bf16x2(s[2 * s2 + 1][2], s[2 * s2 + 1][3]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bf16_t* vp = Vt_lds + (16 * u + fr) * ATT_VT_LD + kt * 64 + 32 * s2 + 4 * fq;
          union { bf16x8_t v; u32x2_t h[2]; } vf;
          vf.h[0] = *reinterpret_cast<const u32x2_t*>(vp);
          vf.h[1] = *reinterpret_cast<const u32x2_t*>(vp + 16);
          o[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[u], 0, 0, 0);
        }
      }
    }
  }
  // ---- finish: combine the 4 lane groups' partial sums, normalise, store
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qp.valid) {
    const float inv = 1.f / l_run;
    bf16_t* op = p.out + (row_base + qp.tok) * g.D + head * ATT_HD + 4 * fq;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x2_t w;
      w.x = pack_bf16x2(o[u][0] * inv, o[u][1] * inv);
      w.y = pack_bf16x2(o[u][2] * inv, o[u][3] * inv);
      *reinterpret_cast<u32x2_t*>(op + 16 * u) = w;
    }
    if (fq == 0 && p.lse) p.lse[((int64_t)gi * g.heads + head) * T + qp.tok] = m_run + __logf(l_run);
  }
}

int attn_fill_geom(AttnGeom* g, int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws,
                   int64_t Lh, int64_t Lw, int64_t NT, float scale, const char* who) {
  MF_CHECK_ARG(hd == ATT_HD, "%s: head dim %lld unsupported (only 64)", who, (long long)hd);
  MF_CHECK_ARG(G > 0 && Sh > 0 && Sw > 0 && heads > 0 && ws >= 0, "%s: bad geometry", who);
  g->G = (int)G; g->Sh = (int)Sh; g->Sw = (int)Sw; g->heads = (int)heads; g->D = (int)(heads * hd);
  g->ws = (int)ws;
  if (ws > 0) {
    g->nWh = (int)ceil_div64(Sh, ws); g->nWw = (int)ceil_div64(Sw, ws); g->KH = g->KW = (int)ws;
  } else {
    g->nWh = g->nWw = 1; g->KH = (int)Sh; g->KW = (int)Sw;
  }
  MF_CHECK_ARG(Lh == 2 * g->KH - 1 && Lw == 2 * g->KW - 1, "%s: rel-pos tables must have 2*k-1 rows (got %lld,%lld for key grid %dx%d)", who, (long long)Lh, (long long)Lw, g->KH, g->KW);
  MF_CHECK_ARG(NT % 32 == 0 && NT >= Lh + Lw && NT <= 256, "%s: NT=%lld must be a multiple of 32 in [Lh+Lw, 256]", who, (long long)NT);
  g->Lh = (int)Lh; g->NT = (int)NT; g->scale = scale;
  return MFNET_OK;
}

static size_t attn_fwd_lds(const AttnGeom& g) {
  return (size_t)ATT_KC * ATT_HD * 2 + (size_t)ATT_HD * ATT_VT_LD * 2 + ATT_KC * 4 + (size_t)ATT_QT * (g.NT + 4) * 4;
}

extern "C" int mfnet_attn_fwd(const void* qkv, const float* qkv_bias, const void* tbl, void* out, float* lse,
                              int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws,
                              int64_t Lh, int64_t Lw, int64_t NT, float scale, hipStream_t stream) {
  MF_CHECK_ARG(qkv && qkv_bias && tbl && out, "attn_fwd: null pointer");
  AttnFwdParams p;
  int rc = attn_fill_geom(&p.g, G, Sh, Sw, heads, hd, ws, Lh, Lw, NT, scale, "attn_fwd");
  if (rc) return rc;
  p.qkv = (const bf16_t*)qkv; p.qkv_bias = qkv_bias; p.tbl = (const bf16_t*)tbl; p.out = (bf16_t*)out; p.lse = lse;
  const size_t lds = attn_fwd_lds(p.g);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { mfnet_set_error("attn_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MFNET_ERR_HIP; }
    attr_set = true;
  }
  const int maxq = p.g.ws > 0 ? p.g.ws * p.g.ws : p.g.Sh * p.g.Sw;
  dim3 grid((unsigned)ceil_div64(maxq, ATT_QT), (unsigned)heads, (unsigned)(G * p.g.nWh * p.g.nWw));
  MF_CHECK_ARG(grid.z <= 65535 && grid.y <= 65535, "attn_fwd: grid too large");
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), lds, stream, p);
  MF_LAUNCH_CHECK("attn_fwd");
  return MFNET_OK;
}
