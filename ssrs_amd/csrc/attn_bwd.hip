// Attention backward (flash-style recompute from Q, K, V, O, LSE) for the SAM ViT blocks; companion of
// attn.hip, same geometry / padded-key semantics (attn_common.h). Padded keys receive no gradient
// (they are the frozen qkv bias); rel-pos tables are frozen, so their only gradient path is into q.
//
// Two kernels, no float atomics in global memory:
//  * attn_bwd_dq : one workgroup per 64-query tile (query on the lane, like the forward):
//        S^T, P^T recomputed; dP^T = V.dO^T; dS^T = P^T*(dP^T - delta);
//        dQ^T = scale * K^T.dS^T + tbl^T.dT^T, where dT[q][row] = sum of dS over the keys that used
//        table row `row` for that query. A key uses rows (KH-1-kh)+qh and Lh+(KW-1-kw)+qw: the key part is
//        independent of the query, so the sums over keys are one more MFMA product per tile,
//        dG^T = G^T.dS^T with the 0/1 indicator G[key][j] (j < KH: KH-1-kh == j; else KW-1-kw == j-KH),
//        and the query offset (qh, qw) is applied once at the end. (The first version used LDS float
//        atomics here and spent 12 of 50 ms per training step in this kernel.)
//  * attn_bwd_dkv: one workgroup per 64 REAL keys (key on the lane), loops over the window's queries:
//        dV^T += dO^T.P, dK^T += scale * Q^T.dS  with P / dS accumulator tiles reused as B operands.
#include "attn_common.h"

struct AttnBwdParams {
  AttnGeom g;
  const bf16_t* qkv;      // [G*T, 3D]
  const float* qkv_bias;  // [3D]
  const bf16_t* tbl;      // [NT, 64]
  const bf16_t* tblT;     // [64, NT]
  const bf16_t* out;      // [G*T, D] forward output O
  const bf16_t* dout;     // [G*T, D]
  const float* lse;       // [G, heads, T]
  bf16_t* dqkv;           // [G*T, 3D]
};

// =============================================================================================== dQ
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnBwdParams p) {
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

  unsigned char* K_lds = smem;                                        // [KC][64] swizzled
  unsigned char* V_lds = smem + ATT_KC * ATT_HD * 2;                  // [KC][64] swizzled
  bf16_t* Kt_lds = (bf16_t*)(smem + 2 * ATT_KC * ATT_HD * 2);         // [64][ATT_VT_LD]
  int* kidx_lds = (int*)((unsigned char*)Kt_lds + ATT_HD * ATT_VT_LD * 2);
  float* T_lds = (float*)(kidx_lds + ATT_KC);                         // [64][NT+4]
  bf16_t* Gt_lds = (bf16_t*)(T_lds + ATT_QT * TLD);                   // [16*nG][ATT_VT_LD] key -> table-row indicators
  float* dT_lds = (float*)smem;                                       // [64][NT+4], aliases K/V after the key loop
  const int nG = (g.KH + g.KW + 15) >> 4;

  const int64_t row_base = (int64_t)gi * T;
  const int64_t ld = 3 * (int64_t)g.D;
  const QueryPos qp = query_pos(g, wi, wj, qt0 + wave * 16 + fr);

  bf16x8_t qf[2], dof[2];
  float delta = 0.f, lse = 0.f;
  {
    const bf16x8_t z = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    const bf16_t* qptr = p.qkv + (row_base + qp.tok) * ld + head * ATT_HD + 8 * fq;
    const bf16_t* dptr = p.dout + (row_base + qp.tok) * g.D + head * ATT_HD + 8 * fq;
    const bf16_t* optr = p.out + (row_base + qp.tok) * g.D + head * ATT_HD + 8 * fq;
    qf[0] = *reinterpret_cast<const bf16x8_t*>(qptr);
    qf[1] = *reinterpret_cast<const bf16x8_t*>(qptr + 32);
    dof[0] = *reinterpret_cast<const bf16x8_t*>(dptr);
    dof[1] = *reinterpret_cast<const bf16x8_t*>(dptr + 32);
    const bf16x8_t o0 = *reinterpret_cast<const bf16x8_t*>(optr);
    const bf16x8_t o1 = *reinterpret_cast<const bf16x8_t*>(optr + 32);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      delta += bf16_to_f32((bf16_t)dof[0][e]) * bf16_to_f32((bf16_t)o0[e]);
      delta += bf16_to_f32((bf16_t)dof[1][e]) * bf16_to_f32((bf16_t)o1[e]);
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    lse = p.lse[((int64_t)gi * g.heads + head) * T + qp.tok];
    if (!qp.valid) { qf[0] = qf[1] = dof[0] = dof[1] = z; delta = 0.f; lse = 0.f; }
  }
  // T^T = tbl . Q^T -> T_lds
  float* Trow = T_lds + (wave * 16 + fr) * TLD;
  float* dTrow = dT_lds + (wave * 16 + fr) * TLD;
  att_table_product(p.tbl, g.NT, qf, Trow, fr, fq);

  f32x4_t dq[4], dgt[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { dq[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dgt[u] = dq[u]; }

  const int ch = tid & 7;          // this thread always stages 16-B chunk `ch` of a key row
  const PaddedKV pad = att_padded_kv(p.qkv_bias, g.D, head, ch);

  for (int kc = 0; kc < Nk; kc += ATT_KC) {
    // all global loads of the chunk first, unconditionally (see attn_fwd_kernel)
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
      kmeta[it] = (tok < 0 ? (tok == -2 ? 2 : 1) : 0) | (kh << 2) | (kw << 10);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ATT_KC / 32; ++it) {
      const int kl = it * 32 + (tid >> 3);
      const int kind = kmeta[it] & 3, kh = (kmeta[it] >> 2) & 0xff, kw = (kmeta[it] >> 10) & 0xff;
      u32x4_t k4 = kind == 0 ? kv[it] : pad.k, v4 = kind == 0 ? vv[it] : pad.v;
      if (kind == 2) { k4 = (u32x4_t){0u, 0u, 0u, 0u}; v4 = k4; }
      *reinterpret_cast<u32x4_t*>(K_lds + kl * 128 + ((ch ^ kswz(kl)) << 4)) = k4;
      *reinterpret_cast<u32x4_t*>(V_lds + kl * 128 + ((ch ^ kswz(kl)) << 4)) = v4;
      bf16_t* kt_ = Kt_lds + (ch * 8) * ATT_VT_LD + kl;
      kt_[0 * ATT_VT_LD] = (bf16_t)(k4.x & 0xffff); kt_[1 * ATT_VT_LD] = (bf16_t)(k4.x >> 16);
      kt_[2 * ATT_VT_LD] = (bf16_t)(k4.y & 0xffff); kt_[3 * ATT_VT_LD] = (bf16_t)(k4.y >> 16);
      kt_[4 * ATT_VT_LD] = (bf16_t)(k4.z & 0xffff); kt_[5 * ATT_VT_LD] = (bf16_t)(k4.z >> 16);
      kt_[6 * ATT_VT_LD] = (bf16_t)(k4.w & 0xffff); kt_[7 * ATT_VT_LD] = (bf16_t)(k4.w >> 16);
      if (ch == 0) kidx_lds[kl] = (g.KH - 1 - kh) | ((g.Lh + g.KW - 1 - kw) << 16);
      // indicator rows: thread (kl, ch) fills rows ch, ch+8, ... of column kl
      const int jh = g.KH - 1 - kh, jw = g.KH + g.KW - 1 - kw;
      const bool kin = kind != 2;
      for (int j = ch; j < 16 * nG; j += 8) Gt_lds[j * ATT_VT_LD + kl] = (kin && (j == jh || j == jw)) ? (bf16_t)0x3f80 : (bf16_t)0;
    }
    __syncthreads();
    const int kvalid = min(ATT_KC, Nk - kc);
    for (int kt = 0; kt * 64 < kvalid; ++kt) {
      f32x4_t s[4], dp[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        s[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        dp[t] = s[t];
        const int kr = kt * 64 + 16 * t + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = kr * 128 + (((fq + 4 * ks) ^ kswz(kr)) << 4);
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(K_lds + off), qf[ks], s[t], 0, 0, 0);
          dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(V_lds + off), dof[ks], dp[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = kt * 64 + 16 * t + 4 * fq + r;
          const int pk = kidx_lds[kl];
          const int ih = (pk & 0xffff) + qp.qh, iw = (pk >> 16) + qp.qw;
          const float v = s[t][r] * g.scale + Trow[ih] + Trow[iw];
          const float pr = (kl < kvalid && qp.valid) ? __expf(v - lse) : 0.f;
          s[t][r] = pr * (dp[t][r] - delta);          // dS (unscaled)
        }
      // dQ^T += scale * K^T . dS^T and dG^T += G^T . dS^T   (key-permuted k-steps, as in the forward's P.V)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        union { bf16x8_t v; unsigned int w[4]; } pf;
        pf.w[0] = pack_bf16x2(s[2 * s2][0], s[2 * s2][1]);
        pf.w[1] = pack_bf16x2(s[2 * s2][2], s[2 * s2][3]);
        pf.w[2] = pack_bf16x2(s[2 * s2 + 1][0], s[2 * s2 + 1][1]);
        pf.w[3] = pack_bf16x2(s[2 * s2 + 1][2], s[2 * s2 + 1][3]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bf16_t* kp = Kt_lds + (16 * u + fr) * ATT_VT_LD + kt * 64 + 32 * s2 + 4 * fq;
          union { bf16x8_t v; u32x2_t h[2]; } kf;
          kf.h[0] = *reinterpret_cast<const u32x2_t*>(kp);
          kf.h[1] = *reinterpret_cast<const u32x2_t*>(kp + 16);
          dq[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, pf.v, dq[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u < nG) {
            const bf16_t* gp = Gt_lds + (16 * u + fr) * ATT_VT_LD + kt * 64 + 32 * s2 + 4 * fq;
            union { bf16x8_t v; u32x2_t h[2]; } gf;
            gf.h[0] = *reinterpret_cast<const u32x2_t*>(gp);
            gf.h[1] = *reinterpret_cast<const u32x2_t*>(gp + 16);
            dgt[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf.v, pf.v, dgt[u], 0, 0, 0);
          }
        }
      }
    }
  }
  __syncthreads();   // K / V images dead: their LDS becomes dT
#pragma unroll
  for (int u = 0; u < 4; ++u) { dq[u][0] *= g.scale; dq[u][1] *= g.scale; dq[u][2] *= g.scale; dq[u][3] *= g.scale; }
  for (int t = 0; t < g.NT / 16; ++t) *reinterpret_cast<float4*>(dTrow + 16 * t + 4 * fq) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  // dT[q][j + qh] = dG[q][j] (j < KH) ; dT[q][Lh + j' + qw] = dG[q][KH + j'] : one lane per (q, j), plain stores
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u < nG) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * u + 4 * fq + r;
        if (j < g.KH) dTrow[j + qp.qh] = dgt[u][r];
        else if (j < g.KH + g.KW) dTrow[g.Lh + (j - g.KH) + qp.qw] = dgt[u][r];
      }
    }
  }
  __syncthreads();
  // dQ^T += tbl^T . dT^T
  for (int s3 = 0; s3 < g.NT / 32; ++s3) {
    const float4 d0 = *reinterpret_cast<const float4*>(dTrow + 32 * s3 + 8 * fq);
    const float4 d1 = *reinterpret_cast<const float4*>(dTrow + 32 * s3 + 8 * fq + 4);
    union { bf16x8_t v; unsigned int w[4]; } bf;
    bf.w[0] = pack_bf16x2(d0.x, d0.y); bf.w[1] = pack_bf16x2(d0.z, d0.w);
    bf.w[2] = pack_bf16x2(d1.x, d1.y); bf.w[3] = pack_bf16x2(d1.z, d1.w);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(p.tblT + (int64_t)(16 * u + fr) * g.NT + 32 * s3 + 8 * fq);
      dq[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf.v, dq[u], 0, 0, 0);
    }
  }
  if (qp.valid) {
    bf16_t* op = p.dqkv + (row_base + qp.tok) * ld + head * ATT_HD + 4 * fq;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x2_t w;
      w.x = pack_bf16x2(dq[u][0], dq[u][1]);
      w.y = pack_bf16x2(dq[u][2], dq[u][3]);
      *reinterpret_cast<u32x2_t*>(op + 16 * u) = w;
    }
  }
}

// ============================================================================================ dK, dV
#define QT_LD (ATT_QT + 8)   // row length (elements) of the transposed [hd][query] images

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int T = g.Sh * g.Sw;
  const int nwin = g.nWh * g.nWw;
  const int gi = blockIdx.z / nwin, win = blockIdx.z - gi * nwin;
  const int wi = win / g.nWw, wj = win - wi * g.nWw;
  const int head = blockIdx.y;
  const int nq = num_queries(g, wi, wj);      // real keys of a window == its real queries
  const int k0 = blockIdx.x * 64;
  if (k0 >= nq) return;
  const int TLD = g.NT + 4;

  unsigned char* Q_lds = smem;                                   // [64][64] swizzled
  unsigned char* dO_lds = smem + ATT_QT * ATT_HD * 2;
  bf16_t* Qt_lds = (bf16_t*)(smem + 2 * ATT_QT * ATT_HD * 2);    // [64][QT_LD]
  bf16_t* dOt_lds = Qt_lds + ATT_HD * QT_LD;
  float* lse_lds = (float*)(dOt_lds + ATT_HD * QT_LD);           // [64]
  float* delta_lds = lse_lds + ATT_QT;                           // [64]
  int* qpos_lds = (int*)(delta_lds + ATT_QT);                    // [64] packed (qh | qw<<8 | valid<<16)
  float* T_lds = (float*)(qpos_lds + ATT_QT);                    // [64][NT+4]

  const int64_t row_base = (int64_t)gi * T;
  const int64_t ld = 3 * (int64_t)g.D;

  // this lane's key (column of S / dP tiles)
  const QueryPos kp = query_pos(g, wi, wj, k0 + wave * 16 + fr);
  bf16x8_t kb[2], vb[2];
  {
    const bf16_t* kptr = p.qkv + (row_base + kp.tok) * ld + g.D + head * ATT_HD + 8 * fq;
    kb[0] = *reinterpret_cast<const bf16x8_t*>(kptr);
    kb[1] = *reinterpret_cast<const bf16x8_t*>(kptr + 32);
    vb[0] = *reinterpret_cast<const bf16x8_t*>(kptr + g.D);
    vb[1] = *reinterpret_cast<const bf16x8_t*>(kptr + g.D + 32);
  }
  const int kbias_h = g.KH - 1 - kp.qh, kbias_w = g.Lh + g.KW - 1 - kp.qw;

  f32x4_t dk[4], dv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { dk[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[u] = dk[u]; }

  for (int qc = 0; qc < nq; qc += ATT_QT) {
    // ---- stage Q / dO (row-major + transposed), lse, delta, query coordinates. All global loads first and
    // unconditional (an invalid query reads the window's first token and is zeroed): one round trip.
    const int ch = tid & 7;
    u32x4_t qv[2], dv_[2], ov[2];
    float lsev[2];
    int qmeta[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const QueryPos qq = query_pos(g, wi, wj, qc + it * 32 + (tid >> 3));
      qv[it] = *reinterpret_cast<const u32x4_t*>(p.qkv + (row_base + qq.tok) * ld + head * ATT_HD + ch * 8);
      dv_[it] = *reinterpret_cast<const u32x4_t*>(p.dout + (row_base + qq.tok) * g.D + head * ATT_HD + ch * 8);
      ov[it] = *reinterpret_cast<const u32x4_t*>(p.out + (row_base + qq.tok) * g.D + head * ATT_HD + ch * 8);
      lsev[it] = p.lse[((int64_t)gi * g.heads + head) * T + qq.tok];
      qmeta[it] = qq.qh | (qq.qw << 8) | (qq.valid << 16);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ql = it * 32 + (tid >> 3);
      const bool valid = (qmeta[it] >> 16) & 1;
      if (!valid) { qv[it] = (u32x4_t){0u, 0u, 0u, 0u}; dv_[it] = qv[it]; ov[it] = qv[it]; }
      *reinterpret_cast<u32x4_t*>(Q_lds + ql * 128 + ((ch ^ kswz(ql)) << 4)) = qv[it];
      *reinterpret_cast<u32x4_t*>(dO_lds + ql * 128 + ((ch ^ kswz(ql)) << 4)) = dv_[it];
      bf16_t* a = Qt_lds + (ch * 8) * QT_LD + ql;
      bf16_t* b = dOt_lds + (ch * 8) * QT_LD + ql;
      const unsigned int qw_[4] = {qv[it].x, qv[it].y, qv[it].z, qv[it].w}, dw_[4] = {dv_[it].x, dv_[it].y, dv_[it].z, dv_[it].w},
                         ow_[4] = {ov[it].x, ov[it].y, ov[it].z, ov[it].w};
      float dl = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[(2 * e) * QT_LD] = (bf16_t)(qw_[e] & 0xffff); a[(2 * e + 1) * QT_LD] = (bf16_t)(qw_[e] >> 16);
        b[(2 * e) * QT_LD] = (bf16_t)(dw_[e] & 0xffff); b[(2 * e + 1) * QT_LD] = (bf16_t)(dw_[e] >> 16);
        dl += bf16_lo(dw_[e]) * bf16_lo(ow_[e]) + bf16_hi(dw_[e]) * bf16_hi(ow_[e]);
      }
      dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64); dl += __shfl_xor(dl, 4, 64);
      if (ch == 0) {
        delta_lds[ql] = dl;
        lse_lds[ql] = valid ? lsev[it] : 0.f;
        qpos_lds[ql] = qmeta[it];
      }
    }
    // ---- T^T = tbl . Q^T for this wave's 16 queries of the chunk -> T_lds[query][row]
    {
      const QueryPos qq = query_pos(g, wi, wj, qc + wave * 16 + fr);
      const bf16_t* qptr = p.qkv + (row_base + qq.tok) * ld + head * ATT_HD + 8 * fq;
      bf16x8_t qf0 = *reinterpret_cast<const bf16x8_t*>(qptr), qf1 = *reinterpret_cast<const bf16x8_t*>(qptr + 32);
      if (!qq.valid) { qf0 = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0}; qf1 = qf0; }
      float* Trow = T_lds + (wave * 16 + fr) * TLD;
      const bf16x8_t qfx[2] = {qf0, qf1};
      att_table_product(p.tbl, g.NT, qfx, Trow, fr, fq);
    }
    __syncthreads();
    const int qvalid = min(ATT_QT, nq - qc);
    for (int q32 = 0; q32 * 32 < qvalid; ++q32) {
      // S[q][key], dP[q][key] for two 16-query sub-tiles
      f32x4_t s[2], dp[2];
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        s[sub] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        dp[sub] = s[sub];
        const int qr = q32 * 32 + 16 * sub + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = qr * 128 + (((fq + 4 * ks) ^ kswz(qr)) << 4);
          s[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(Q_lds + off), kb[ks], s[sub], 0, 0, 0);
          dp[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(dO_lds + off), vb[ks], dp[sub], 0, 0, 0);
        }
      }
      union { bf16x8_t v; unsigned int w[4]; } pf, dsf;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        float pr[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = q32 * 32 + 16 * sub + 4 * fq + r;
          const int qpk = qpos_lds[ql];
          const int qh = qpk & 0xff, qw = (qpk >> 8) & 0xff, ok = (qpk >> 16) & kp.valid;
          const float* Tr = T_lds + ql * TLD;
          const float v = s[sub][r] * g.scale + Tr[kbias_h + qh] + Tr[kbias_w + qw];
          pr[r] = ok ? __expf(v - lse_lds[ql]) : 0.f;
          ds[r] = pr[r] * (dp[sub][r] - delta_lds[ql]) * g.scale;
        }
        pf.w[2 * sub] = pack_bf16x2(pr[0], pr[1]); pf.w[2 * sub + 1] = pack_bf16x2(pr[2], pr[3]);
        dsf.w[2 * sub] = pack_bf16x2(ds[0], ds[1]); dsf.w[2 * sub + 1] = pack_bf16x2(ds[2], ds[3]);
      }
      // dV^T += dO^T . P ; dK^T += Q^T . dS   (query slot j <-> query 16*(j>>2) + 4*fq + (j&3))
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bf16_t* dop = dOt_lds + (16 * u + fr) * QT_LD + q32 * 32 + 4 * fq;
        const bf16_t* qtp = Qt_lds + (16 * u + fr) * QT_LD + q32 * 32 + 4 * fq;
        union { bf16x8_t v; u32x2_t h[2]; } af, bf;
        af.h[0] = *reinterpret_cast<const u32x2_t*>(dop); af.h[1] = *reinterpret_cast<const u32x2_t*>(dop + 16);
        bf.h[0] = *reinterpret_cast<const u32x2_t*>(qtp); bf.h[1] = *reinterpret_cast<const u32x2_t*>(qtp + 16);
        dv[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, pf.v, dv[u], 0, 0, 0);
        dk[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf.v, dsf.v, dk[u], 0, 0, 0);
      }
    }
  }
  if (kp.valid) {
    bf16_t* op = p.dqkv + (row_base + kp.tok) * ld + g.D + head * ATT_HD + 4 * fq;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      u32x2_t w;
      w.x = pack_bf16x2(dk[u][0], dk[u][1]); w.y = pack_bf16x2(dk[u][2], dk[u][3]);
      *reinterpret_cast<u32x2_t*>(op + 16 * u) = w;
      w.x = pack_bf16x2(dv[u][0], dv[u][1]); w.y = pack_bf16x2(dv[u][2], dv[u][3]);
      *reinterpret_cast<u32x2_t*>(op + g.D + 16 * u) = w;
    }
  }
}

int attn_fill_geom(AttnGeom* g, int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws,
                   int64_t Lh, int64_t Lw, int64_t NT, float scale, const char* who);

extern "C" int mfnet_attn_bwd(const void* qkv, const float* qkv_bias, const void* tbl, const void* tblT,
                              const void* out, const void* dout, const float* lse, void* dqkv, int64_t G,
                              int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws, int64_t Lh,
                              int64_t Lw, int64_t NT, float scale, hipStream_t stream) {
  MF_CHECK_ARG(qkv && qkv_bias && tbl && tblT && out && dout && lse && dqkv, "attn_bwd: null pointer");
  AttnBwdParams p;
  int rc = attn_fill_geom(&p.g, G, Sh, Sw, heads, hd, ws, Lh, Lw, NT, scale, "attn_bwd");
  if (rc) return rc;
  if (NT > 96) { mfnet_set_error("attn_bwd: NT=%lld > 96 (token grids above 24x24) not supported yet", (long long)NT); return MFNET_ERR_UNSUPPORTED; }
  MF_CHECK_ARG(p.g.KH < 256 && p.g.KW < 256, "attn_bwd: key grid too large");
  p.qkv = (const bf16_t*)qkv; p.qkv_bias = qkv_bias; p.tbl = (const bf16_t*)tbl; p.tblT = (const bf16_t*)tblT;
  p.out = (const bf16_t*)out; p.dout = (const bf16_t*)dout; p.lse = lse; p.dqkv = (bf16_t*)dqkv;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { mfnet_set_error("attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MFNET_ERR_HIP; }
    attr_set = true;
  }
  const int maxq = p.g.ws > 0 ? p.g.ws * p.g.ws : p.g.Sh * p.g.Sw;
  dim3 grid((unsigned)ceil_div64(maxq, ATT_QT), (unsigned)heads, (unsigned)(G * p.g.nWh * p.g.nWw));
  MF_CHECK_ARG(grid.z <= 65535 && grid.y <= 65535, "attn_bwd: grid too large");
  const size_t TL = (size_t)ATT_QT * (p.g.NT + 4) * 4;
  const size_t nGrows = 16 * (size_t)((p.g.KH + p.g.KW + 15) / 16);
  MF_CHECK_ARG(nGrows <= 64 && TL <= 2 * (size_t)ATT_KC * ATT_HD * 2, "attn_bwd: key grid too large for the dT staging");
  const size_t lds_dq = 2 * (size_t)ATT_KC * ATT_HD * 2 + (size_t)ATT_HD * ATT_VT_LD * 2 + ATT_KC * 4 + TL + nGrows * ATT_VT_LD * 2;
  const size_t lds_dkv = 2 * (size_t)ATT_QT * ATT_HD * 2 + 2 * (size_t)ATT_HD * QT_LD * 2 + 3 * ATT_QT * 4 + TL;
  // the two kernels write disjoint column ranges of dqkv (q | k,v); window mode leaves nothing unwritten
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), lds_dq, stream, p);
  MF_LAUNCH_CHECK("attn_bwd_dq");
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), lds_dkv, stream, p);
  MF_LAUNCH_CHECK("attn_bwd_dkv");
  return MFNET_OK;
}
