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
#include <mutex>
#define F4Z make_float4(0.f, 0.f, 0.f, 0.f)

struct AttnBwdParams {
  AttnGeom g;
  const bf16_t* qkv;      // [G*T, 3D]
  const float* qkv_bias;  // [3D]
  const bf16_t* tbl;      // [NT, 32*ceil(hd/32)], zero padded columns
  const bf16_t* tblT;     // [hd, NT]
  const bf16_t* out;      // [G*T, D] forward output O
  const bf16_t* dout;     // [G*T, D]
  const float* lse;       // [G, heads, T]
  bf16_t* dqkv;           // [G*T, 3D]
  float* bias_ws;         // dQ writes, dK/dV reads, per query: ROWT [ATT_WS_LD] fp32 - Th[kh] / scale (KH) then Tw[kw] / scale (KW);
                          // else [nsg][hi | lo][32 slots] bf16 - the slot rows of T' / scale (attn.hip)
};
#define ATT_WS_LD 128

// =============================================================================================== dQ
// Same structure as attn_fwd_kernel (attn.hip): compact (window, tile) grid, K / V row-major in LDS, rel-pos
// bias as the indicator product G.T'^T. The two transposed operands of this kernel, K^T (dQ^T += K^T.dS^T) and
// G^T (dT'^T += G^T.dS^T), are read from the SAME row-major images with ds_read_b64_tr_b16.
// ROWT (see attn_fwd_kernel): the bias enters as the initial accumulator; its gradient needs no indicator product either -
// the w part is the lane's own 4 x 4 patch of dS summed over the tiles, the h part one per-query sum of dS per tile.
template <int NSG, int HD, bool ROWT>
__global__ __launch_bounds__(256, HD > 64 ? 1 : ((NSG == 1 || ROWT) ? 3 : 2)) void attn_bwd_dq_kernel(const AttnBwdParams p) {
  typedef AttHd<HD> H;
  constexpr int KS = H::KS, UT = H::UT, ROWB = H::ROWB, CHS = H::CHS, CHV = H::CHV, RPI = H::RPI, NIT = ATT_KC / H::RPI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int T = g.Sh * g.Sw;
  int gi, head, tile_id;
  if (!att_locate_wg(g.ntile, g.heads, g.G, &tile_id, &head, &gi)) return;
  int wi, wj, qt0;
  att_locate_tile(g, tile_id, &wi, &wj, &qt0);
  const int nq = num_queries(g, wi, wj);
  const int Nk = g.KH * g.KW;
  const int TLD = g.NT + 4;
  const int nsg = (g.KH + g.KW + 31) >> 5;

  unsigned char* K_lds = smem;                                        // [128][64] swizzled
  unsigned char* V_lds = smem + ATT_KC * ROWB;                        // [128][hd] swizzled
  float* T_lds = (float*)smem;                                        // [64][NT+4] before the key loop; dT after it
  unsigned char* G_lds = smem + att_region_a(g.NT, ROWB, ROWT);       // [nsg][128][32] bf16 indicator rows

  const int64_t row_base = (int64_t)gi * T;
  const int ld = 3 * g.D;                                          // (one image's qkv rows span < 2^31 elements)
  const bf16_t* qkv_img = p.qkv + row_base * ld;
  bf16_t* dqkv_img = p.dqkv + row_base * ld;
  const bf16_t* dout_img = p.dout + row_base * g.D;
  const bf16_t* out_img = p.out + row_base * g.D;
  const QueryPos qp = query_pos(g, wi, wj, qt0 + wave * 16 + fr);
  const bool wave_active = qt0 + wave * 16 < nq;

  bf16x8_t qf[KS], dof[KS];
  float delta = 0.f, lse2 = 0.f;
  {
    bf16x8_t of[KS];
    att_load_row_frags<HD>(qkv_img + qp.tok * ld + head * HD, fq, qp.valid, qf);
    att_load_row_frags<HD>(dout_img + qp.tok * g.D + head * HD, fq, qp.valid, dof);
    att_load_row_frags<HD>(out_img + qp.tok * g.D + head * HD, fq, qp.valid, of);
    lse2 = qp.valid ? p.lse[((int64_t)gi * g.heads + head) * T + qp.tok] * 1.4426950408889634f : 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) delta += bf16_to_f32((bf16_t)dof[ks][e]) * bf16_to_f32((bf16_t)of[ks][e]);
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
  }
  // T^T = tbl . Q^T -> T_lds, then this lane's slots of T' / scale (attn.hip)
  bf16x8_t tqh[NSG], tql[NSG];
  float wb[4][4], dwb[4][4];                                          // ROWT: w-term bias / scale of this lane's keys, and its gradient
  float* Thq = (float*)G_lds;                                         // ROWT: [64 queries][KH] h term / scale
  float* dThq = Thq;                                                  // ROWT: its gradient, IN PLACE: a tile is one key row, visited once per query -
                                                                      // entry [q][row] is read (hb) before the same wave overwrites it with the row's dS sum
  float* Tb = T_lds + (wave * 16 + fr) * ATT_TB_LD;                   // ROWT: this lane's query row of the wave's 64-row table block
  if (ROWT) {
    // the table is walked 64 rows at a time through the wave's own block buffer (attn.hip)
    const float inv = 1.0f / g.scale;
    float* wr = p.bias_ws + (((int64_t)gi * g.heads + head) * T + qp.tok) * ATT_WS_LD;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { wb[t][r] = 0.f; dwb[t][r] = 0.f; }
    for (int b0 = 0; b0 < g.NT; b0 += 64) {
      att_table_block<HD>(p.tbl, g.NT, b0, qf, Tb, fr, fq);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = g.Lh + (g.KW - 1 - (16 * t + 4 * fq + r)) + qp.qw - b0;
          if (idx >= 0 && idx < 64) wb[t][r] = Tb[idx] * inv;
        }
      for (int j = fq; j < g.KH; j += 4) {
        const int idx = (g.KH - 1 - j) + qp.qh - b0;
        if (idx >= 0 && idx < 64) {
          const float v = Tb[idx] * inv;
          Thq[(wave * 16 + fr) * g.KH + j] = v;
          if (qp.valid) wr[j] = v;    // the dK/dV kernel reads these rows instead of repeating the table product per key tile
        }
      }
    }
    if (qp.valid) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        *reinterpret_cast<float4*>(wr + g.KH + 16 * t + 4 * fq) = make_float4(wb[t][0], wb[t][1], wb[t][2], wb[t][3]);
    }
  } else {
    float* Trow = T_lds + (wave * 16 + fr) * TLD;
    att_table_product<HD>(p.tbl, g.NT, qf, Trow, fr, fq);
    att_slot_fragments<NSG>(g, Trow, qp.qh, qp.qw, fq, nsg, tqh, tql);
    if (qp.valid) {                   // the dK/dV kernel reads the query's slot rows (hi, lo) instead of recomputing them per key tile
      unsigned char* wr = (unsigned char*)p.bias_ws + (((int64_t)gi * g.heads + head) * T + qp.tok) * (int64_t)(nsg * 128);
#pragma unroll
      for (int c = 0; c < NSG; ++c)
        if (c < nsg) {
          *reinterpret_cast<bf16x8_t*>(wr + c * 128 + 16 * fq) = tqh[c];
          *reinterpret_cast<bf16x8_t*>(wr + c * 128 + 64 + 16 * fq) = tql[c];
        }
    }
  }
  const float cs = g.scale * 1.4426950408889634f;

  f32x4_t dq[UT], dgt[NSG][2];
#pragma unroll
  for (int u = 0; u < UT; ++u) dq[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NSG; ++c) dgt[c][0] = dgt[c][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int ch = tid % CHS, chl = ch < CHV ? ch : CHV - 1;   // this thread always stages 16-B chunk slot `ch` of a key row
  const PaddedKV pad = att_padded_kv(p.qkv_bias, g.D, head, HD, chl);

  // all global loads of a chunk first, unconditionally (see attn_fwd_kernel)
  u32x4_t kv[NIT], vv[NIT];
  int kkind[NIT];
  auto issue_chunk = [&](int kc) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int kg = kc + it * RPI + tid / CHS;
      int kh = 0, kw = 0;
      const int tok = kg < Nk ? key_token(g, wi, wj, kg, &kh, &kw) : -2;
      const bf16_t* src = qkv_img + max(tok, 0) * ld + g.D + head * HD + chl * 8;
      kv[it] = *reinterpret_cast<const u32x4_t*>(src);
      vv[it] = *reinterpret_cast<const u32x4_t*>(src + g.D);
      kkind[it] = ch >= CHV ? 2 : (tok < 0 ? (tok == -2 ? 2 : 1) : 0);
    }
  };
  for (int kc = 0; kc < Nk; kc += ATT_KC) {
    issue_chunk(kc);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int kl = it * RPI + tid / CHS;
      // three exclusive stores under the execution mask instead of 16 selects per row: a window without padded keys (most of the
      // work) takes the first branch as a whole
      u32x4_t* kd = reinterpret_cast<u32x4_t*>(K_lds + kl * ROWB + ((ch ^ kswz(kl)) << 4));
      u32x4_t* vd = reinterpret_cast<u32x4_t*>(V_lds + kl * ROWB + ((ch ^ kswz(kl)) << 4));
      if (kkind[it] == 0) { *kd = kv[it]; *vd = vv[it]; }
      else if (kkind[it] == 1) { *kd = pad.k; *vd = pad.v; }
      else { *kd = (u32x4_t){0u, 0u, 0u, 0u}; *vd = (u32x4_t){0u, 0u, 0u, 0u}; }
    }
    if (!ROWT) att_stage_indicator(g, G_lds, kc, Nk, nsg, tid);
    __syncthreads();
    const int kvalid = min(ATT_KC, Nk - kc);
    if (wave_active) {
      for (int kt = 0; kt * 64 < kvalid; ++kt) {
        f32x4_t s[4], dp[4];
        const int krow_h = (kc + kt * 64) >> 6;                        // ROWT: the key row of this tile
        const float hb = ROWT ? Thq[(wave * 16 + fr) * g.KH + krow_h] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          s[t] = ROWT ? (f32x4_t){wb[t][0] + hb, wb[t][1] + hb, wb[t][2] + hb, wb[t][3] + hb} : (f32x4_t){0.f, 0.f, 0.f, 0.f};
          dp[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          const int kr = kt * 64 + 16 * t + fr;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int off = kr * ROWB + (((fq + 4 * ks) ^ kswz(kr)) << 4);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(K_lds + off), qf[ks], s[t], 0, 0, 0);
            dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(V_lds + off), dof[ks], dp[t], 0, 0, 0);
          }
#pragma unroll
          for (int c = 0; c < NSG; ++c) {
            if (!ROWT && c < nsg) {
              const bf16x8_t gf = *reinterpret_cast<const bf16x8_t*>(G_lds + c * (ATT_KC * ATT_GROW) + kr * ATT_GROW + ((fq ^ gswz(kr)) << 4));
              s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, tqh[c], s[t], 0, 0, 0);
              s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf, tql[c], s[t], 0, 0, 0);
            }
          }
        }
        if (kt * 64 + 64 > kvalid) {                   // only the window's last tile holds keys beyond Nk: a wave-uniform branch, so
#pragma unroll                                         // the other tiles carry no compare / select per score (48 of 154 VALU per tile)
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pr = __builtin_amdgcn_exp2f(fmaf(s[t][r], cs, -lse2));
              if (kt * 64 + 16 * t + 4 * fq + r >= kvalid) pr = 0.f;
              s[t][r] = pr * (dp[t][r] - delta);        // dS (unscaled)
            }
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], cs, -lse2)) * (dp[t][r] - delta);
        }
        if (ROWT) {
          float hs = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { dwb[t][r] += s[t][r]; hs += s[t][r]; }
          hs += __shfl_xor(hs, 16, 64);
          hs += __shfl_xor(hs, 32, 64);
          if (fq == 0) dThq[(wave * 16 + fr) * g.KH + krow_h] = hs;       // each key row is visited once
        }
        // dQ^T += K^T . dS^T and dT'^T += G^T . dS^T   (key-permuted k-steps, as in the forward's P.V; the
        // transposed fragments come from the row-major images by ds_read_b64_tr_b16)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          union { bf16x8_t v; unsigned int w[4]; } pf;
          pf.w[0] = pack_bf16x2(s[2 * s2][0], s[2 * s2][1]);
          pf.w[1] = pack_bf16x2(s[2 * s2][2], s[2 * s2][3]);
          pf.w[2] = pack_bf16x2(s[2 * s2 + 1][0], s[2 * s2 + 1][1]);
          pf.w[3] = pack_bf16x2(s[2 * s2 + 1][2], s[2 * s2 + 1][3]);
          const int krow = kt * 64 + 32 * s2 + 4 * fq + (fr >> 2);
#pragma unroll
          for (int u = 0; u < UT; ++u) {
            const unsigned char* ka = K_lds + krow * ROWB + (((2 * u + ((fr >> 1) & 1)) ^ kswz(krow)) << 4) + ((fr & 1) << 3);
            union { bf16x8_t v; s16x4_t h[2]; } kf;
            kf.h[0] = att_tr_read(ka);
            kf.h[1] = att_tr_read(ka + 16 * ROWB);
            dq[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf.v, pf.v, dq[u], 0, 0, 0);
          }
#pragma unroll
          for (int c = 0; c < NSG; ++c) {
            if (!ROWT && c < nsg) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const unsigned char* ga = G_lds + c * (ATT_KC * ATT_GROW) + krow * ATT_GROW + (((2 * u + ((fr >> 1) & 1)) ^ gswz(krow)) << 4) + ((fr & 1) << 3);
                union { bf16x8_t v; s16x4_t h[2]; } gf;
                gf.h[0] = att_tr_read(ga);
                gf.h[1] = att_tr_read(ga + 16 * ATT_GROW);
                dgt[c][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gf.v, pf.v, dgt[c][u], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();   // K / V images dead: their LDS becomes dT
#pragma unroll
  for (int u = 0; u < UT; ++u) { dq[u][0] *= g.scale; dq[u][1] *= g.scale; dq[u][2] *= g.scale; dq[u][3] *= g.scale; }
  if (ROWT) {
    // dQ^T += tbl^T . dT^T, 64 table rows at a time through the wave's block buffer: zero it, drop in the entries of dT that fall
    // into the block (dT[q][(KH-1-j) + qh] = dThq[q][j]; dT[q][Lh + (KW-1-kw) + qw] = dwb of key column kw), multiply
    if (wave_active) {
      for (int b0 = 0; b0 < g.NT; b0 += 64) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(Tb + 16 * fq + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = fq; j < g.KH; j += 4) {
          const int idx = (g.KH - 1 - j) + qp.qh - b0;
          if (idx >= 0 && idx < 64) Tb[idx] = dThq[(wave * 16 + fr) * g.KH + j];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = g.Lh + (g.KW - 1 - (16 * t + 4 * fq + r)) + qp.qw - b0;
            if (idx >= 0 && idx < 64) Tb[idx] = dwb[t][r];
          }
#pragma unroll
        for (int s3 = 0; s3 < 2; ++s3) {
          if (b0 + 32 * s3 < g.NT) {
            const float4 d0 = *reinterpret_cast<const float4*>(Tb + 32 * s3 + 8 * fq);
            const float4 d1 = *reinterpret_cast<const float4*>(Tb + 32 * s3 + 8 * fq + 4);
            union { bf16x8_t v; unsigned int w[4]; } bf;
            bf.w[0] = pack_bf16x2(d0.x, d0.y); bf.w[1] = pack_bf16x2(d0.z, d0.w);
            bf.w[2] = pack_bf16x2(d1.x, d1.y); bf.w[3] = pack_bf16x2(d1.z, d1.w);
#pragma unroll
            for (int u = 0; u < UT; ++u) {
              const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(p.tblT + (int64_t)(16 * u + fr) * g.NT + b0 + 32 * s3 + 8 * fq);
              dq[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf.v, dq[u], 0, 0, 0);
            }
          }
        }
      }
    }
  } else {
    float* dTrow = T_lds + (wave * 16 + fr) * TLD;
    for (int t = 0; t < g.NT / 16; ++t) *reinterpret_cast<float4*>(dTrow + 16 * t + 4 * fq) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // dT[q][j + qh] = dT'[q][j] (j < KH) ; dT[q][Lh + j' + qw] = dT'[q][KH + j'] : one lane per (q, slot), plain stores
#pragma unroll
    for (int c = 0; c < NSG; ++c)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 32 * c + 16 * u + 4 * fq + r;
          if (j < g.KH) dTrow[j + qp.qh] = dgt[c][u][r];
          else if (j < g.KH + g.KW) dTrow[g.Lh + (j - g.KH) + qp.qw] = dgt[c][u][r];
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
      for (int u = 0; u < UT; ++u) {
        const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(p.tblT + (int64_t)(16 * u + fr) * g.NT + 32 * s3 + 8 * fq);
        dq[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf.v, dq[u], 0, 0, 0);
      }
    }
  }
  if (qp.valid) {
    bf16_t* op = dqkv_img + qp.tok * ld + head * HD + 4 * fq;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
      u32x2_t w;
      w.x = pack_bf16x2(dq[u][0], dq[u][1]);
      w.y = pack_bf16x2(dq[u][2], dq[u][3]);
      *reinterpret_cast<u32x2_t*>(op + 16 * u) = w;
    }
  }
}

// ============================================================================================ dK, dV
// One workgroup per 64 REAL keys of a window (key on the lane), looping over the window's queries in chunks
// of 64: dV^T += dO^T.P, dK^T += scale * Q^T.dS with the P / dS accumulator tiles reused as B operands. Q and
// dO are staged row-major only; their transposes come from ds_read_b64_tr_b16. The rel-pos bias is the
// product T'.G^T: T' (per query, hi + lo bf16 rows staged in LDS) times this lane's key indicator (registers).
// ROWT (see attn_fwd_kernel): the workgroup's 64 keys are one key row; the bias of (query, key) is read from two fp32
// LDS images of the chunk's queries, Th[q][kh] and Tw[q][kw] (row stride ATT_TS floats: conflict-free for the 4 query
// rows x 16 key columns a wave reads at once), instead of the T'.G^T indicator product.
#define ATT_TS 68
template <int NSG, int HD, bool ROWT>
__global__ __launch_bounds__(256, HD > 64 ? 1 : (NSG == 1 ? 4 : ((NSG == 4 && !ROWT) ? 2 : 3))) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
  typedef AttHd<HD> H;
  constexpr int KS = H::KS, UT = H::UT, ROWB = H::ROWB, CHS = H::CHS, CHV = H::CHV, RPI = H::RPI, NIT = ATT_QT / H::RPI;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnGeom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int T = g.Sh * g.Sw;
  int gi, head, tile_id;
  if (!att_locate_wg(g.ntile, g.heads, g.G, &tile_id, &head, &gi)) return;
  int wi, wj, k0;
  att_locate_tile(g, tile_id, &wi, &wj, &k0);
  const int nq = num_queries(g, wi, wj);      // real keys of a window == its real queries
  const int nsg = (g.KH + g.KW + 31) >> 5;

  unsigned char* Q_lds = smem;                                   // [64][hd] swizzled
  unsigned char* dO_lds = smem + ATT_QT * ROWB;
  float* lse_lds = (float*)(smem + 2 * ATT_QT * ROWB);           // [64] lse * log2(e); +huge for an absent query
  float* delta_lds = lse_lds + ATT_QT;                           // [64]
  unsigned char* TqH_lds = (unsigned char*)(delta_lds + ATT_QT); // [nsg][64][32 slots] bf16, hi part of T'/scale
  const int tq_bytes = ROWT ? ATT_QT * ATT_TS * 4 : nsg * ATT_QT * ATT_GROW;
  unsigned char* TqL_lds = TqH_lds + tq_bytes;                   // lo part
  float* Thq = (float*)TqH_lds;                                  // ROWT: [64 queries][ATT_TS] h term / scale by key row
  float* Twq = (float*)TqL_lds;                                  // ROWT: [64 queries][ATT_TS] w term / scale by key column

  const int64_t row_base = (int64_t)gi * T;
  const int ld = 3 * g.D;                                          // (one image's qkv rows span < 2^31 elements)
  const bf16_t* qkv_img = p.qkv + row_base * ld;
  bf16_t* dqkv_img = p.dqkv + row_base * ld;
  const bf16_t* dout_img = p.dout + row_base * g.D;
  const bf16_t* out_img = p.out + row_base * g.D;

  // this lane's key (column of S / dP tiles) and its indicator slots
  const QueryPos kp = query_pos(g, wi, wj, k0 + wave * 16 + fr);
  const bool wave_active = k0 + wave * 16 < nq;
  bf16x8_t kb[KS], vb[KS];
  att_load_row_frags<HD>(qkv_img + kp.tok * ld + g.D + head * HD, fq, true, kb);
  att_load_row_frags<HD>(qkv_img + kp.tok * ld + 2 * g.D + head * HD, fq, true, vb);
  bf16x8_t gk[NSG];
  {
    const int jh = g.KH - 1 - kp.qh, jw = g.KH + g.KW - 1 - kp.qw;
#pragma unroll
    for (int c = 0; c < NSG; ++c) {
      union { bf16x8_t v; unsigned int w[4]; } u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int a = 32 * c + 8 * fq + 2 * i, b = a + 1;
        u.w[i] = ((a == jh || a == jw) ? 0x3f80u : 0u) | ((b == jh || b == jw) ? 0x3f800000u : 0u);
      }
      gk[c] = u.v;
    }
  }
  const float cs = g.scale * 1.4426950408889634f;

  f32x4_t dk[UT], dv[UT];
#pragma unroll
  for (int u = 0; u < UT; ++u) { dk[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[u] = dk[u]; }

  for (int qc = 0; qc < nq; qc += ATT_QT) {
    // ---- stage Q / dO (row-major), lse, delta and the T' rows of the chunk's queries. All global loads first
    // and unconditional (an absent query reads the window's first token and is zeroed): one round trip.
    const int ch = tid % CHS, chl = ch < CHV ? ch : CHV - 1;       // slots >= CHV hold zeros
    u32x4_t qv[NIT], dv_[NIT], ov[NIT];
    float lsev[NIT];
    bool qok[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const QueryPos qq = query_pos(g, wi, wj, qc + it * RPI + tid / CHS);
      qv[it] = *reinterpret_cast<const u32x4_t*>(qkv_img + qq.tok * ld + head * HD + chl * 8);
      dv_[it] = *reinterpret_cast<const u32x4_t*>(dout_img + qq.tok * g.D + head * HD + chl * 8);
      ov[it] = *reinterpret_cast<const u32x4_t*>(out_img + qq.tok * g.D + head * HD + chl * 8);
      lsev[it] = p.lse[((int64_t)gi * g.heads + head) * T + qq.tok];
      qok[it] = qq.valid;
    }
    const QueryPos qw_ = query_pos(g, wi, wj, qc + wave * 16 + fr);      // this wave's 16 queries: table product
    constexpr bool from_ws = ROWT;                                        // the dQ kernel left the bias rows in global memory
    float4 wsv0 = F4Z, wsv1 = F4Z, wsv2 = F4Z, wsv3 = F4Z;
    float wsh = 0.f;
    if constexpr (from_ws) {
      // thread -> query tid >> 2, 16 of its Tw values (4 float4) and, for the first of the four, its Th[kh of this key row]
      const QueryPos qq = query_pos(g, wi, wj, qc + (tid >> 2));
      const float* wr = p.bias_ws + (((int64_t)gi * g.heads + head) * T + qq.tok) * ATT_WS_LD;
      const float4* w4 = reinterpret_cast<const float4*>(wr + g.KH + 16 * (tid & 3));
      wsv0 = w4[0]; wsv1 = w4[1]; wsv2 = w4[2]; wsv3 = w4[3];
      wsh = wr[kp.qh];
    }
    bf16x8_t th[NSG], tl[NSG];
    if constexpr (!from_ws) {
      const unsigned char* wr = (const unsigned char*)p.bias_ws + (((int64_t)gi * g.heads + head) * T + qw_.tok) * (int64_t)(nsg * 128);
#pragma unroll
      for (int c = 0; c < NSG; ++c) {
        const int cc = c < nsg ? c : 0;
        th[c] = *reinterpret_cast<const bf16x8_t*>(wr + cc * 128 + 16 * fq);
        tl[c] = *reinterpret_cast<const bf16x8_t*>(wr + cc * 128 + 64 + 16 * fq);
        if (!qw_.valid) { th[c] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0}; tl[c] = th[c]; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ql = it * RPI + tid / CHS;
      if (!qok[it] || ch >= CHV) { qv[it] = (u32x4_t){0u, 0u, 0u, 0u}; dv_[it] = qv[it]; ov[it] = qv[it]; }
      *reinterpret_cast<u32x4_t*>(Q_lds + ql * ROWB + ((ch ^ kswz(ql)) << 4)) = qv[it];
      *reinterpret_cast<u32x4_t*>(dO_lds + ql * ROWB + ((ch ^ kswz(ql)) << 4)) = dv_[it];
      const unsigned int dw_[4] = {dv_[it].x, dv_[it].y, dv_[it].z, dv_[it].w}, ow_[4] = {ov[it].x, ov[it].y, ov[it].z, ov[it].w};
      float dl = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) dl += bf16_lo(dw_[e]) * bf16_lo(ow_[e]) + bf16_hi(dw_[e]) * bf16_hi(ow_[e]);
      dl += __shfl_xor(dl, 1, 64); dl += __shfl_xor(dl, 2, 64); dl += __shfl_xor(dl, 4, 64);
      if (CHS > 8) dl += __shfl_xor(dl, 8, 64);
      if (ch == 0) {
        delta_lds[ql] = dl;
        lse_lds[ql] = qok[it] ? lsev[it] * 1.4426950408889634f : 1e30f;     // absent query: P = exp2(-huge) = 0
      }
    }
    if constexpr (from_ws) {
      const int ql = tid >> 2;
      float4* w4 = reinterpret_cast<float4*>(Twq + ql * ATT_TS + 16 * (tid & 3));
      w4[0] = wsv0; w4[1] = wsv1; w4[2] = wsv2; w4[3] = wsv3;
      if ((tid & 3) == 0) Thq[ql * ATT_TS + kp.qh] = wsh;
    } else {
      const int ql = wave * 16 + fr;
#pragma unroll
      for (int c = 0; c < NSG; ++c) {
        if (c < nsg) {
          const int off = c * (ATT_QT * ATT_GROW) + ql * ATT_GROW + ((fq ^ gswz(ql)) << 4);
          *reinterpret_cast<bf16x8_t*>(TqH_lds + off) = th[c];
          *reinterpret_cast<bf16x8_t*>(TqL_lds + off) = tl[c];
        }
      }
    }
    __syncthreads();
    const int qvalid = min(ATT_QT, nq - qc);
    if (wave_active) {
      for (int q32 = 0; q32 * 32 < qvalid; ++q32) {
        // S[q][key], dP[q][key] for two 16-query sub-tiles
        union { bf16x8_t v; unsigned int w[4]; } pf, dsf;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
          if (ROWT) {
            const int q0 = q32 * 32 + 16 * sub + 4 * fq;
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] = Thq[(q0 + r) * ATT_TS + kp.qh] + Twq[(q0 + r) * ATT_TS + kp.qw];
          }
          const int qr = q32 * 32 + 16 * sub + fr;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const int off = qr * ROWB + (((fq + 4 * ks) ^ kswz(qr)) << 4);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(Q_lds + off), kb[ks], s, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(dO_lds + off), vb[ks], dp, 0, 0, 0);
          }
#pragma unroll
          for (int c = 0; c < NSG; ++c) {
            if (!ROWT && c < nsg) {
              const int off = c * (ATT_QT * ATT_GROW) + qr * ATT_GROW + ((fq ^ gswz(qr)) << 4);
              s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(TqH_lds + off), gk[c], s, 0, 0, 0);
              s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8_t*>(TqL_lds + off), gk[c], s, 0, 0, 0);
            }
          }
          const int ql0 = q32 * 32 + 16 * sub + 4 * fq;
          const float4 l4 = *reinterpret_cast<const float4*>(lse_lds + ql0), d4 = *reinterpret_cast<const float4*>(delta_lds + ql0);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dlv[4] = {d4.x, d4.y, d4.z, d4.w};
          float pr[4], ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pr[r] = __builtin_amdgcn_exp2f(fmaf(s[r], cs, -lv[r]));
            ds[r] = pr[r] * (dp[r] - dlv[r]) * g.scale;
          }
          pf.w[2 * sub] = pack_bf16x2(pr[0], pr[1]); pf.w[2 * sub + 1] = pack_bf16x2(pr[2], pr[3]);
          dsf.w[2 * sub] = pack_bf16x2(ds[0], ds[1]); dsf.w[2 * sub + 1] = pack_bf16x2(ds[2], ds[3]);
        }
        // dV^T += dO^T . P ; dK^T += Q^T . dS   (query slot j <-> query 16*(j>>2) + 4*fq + (j&3))
        const int qrow = q32 * 32 + 4 * fq + (fr >> 2);
#pragma unroll
        for (int u = 0; u < UT; ++u) {
          const int off = qrow * ROWB + (((2 * u + ((fr >> 1) & 1)) ^ kswz(qrow)) << 4) + ((fr & 1) << 3);
          union { bf16x8_t v; s16x4_t h[2]; } af, bf;
          af.h[0] = att_tr_read(dO_lds + off); af.h[1] = att_tr_read(dO_lds + off + 16 * ROWB);
          bf.h[0] = att_tr_read(Q_lds + off); bf.h[1] = att_tr_read(Q_lds + off + 16 * ROWB);
          dv[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, pf.v, dv[u], 0, 0, 0);
          dk[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf.v, dsf.v, dk[u], 0, 0, 0);
        }
      }
    }
  }
  if (kp.valid) {
    bf16_t* op = dqkv_img + kp.tok * ld + g.D + head * HD + 4 * fq;
#pragma unroll
    for (int u = 0; u < UT; ++u) {
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
int attn_query_tiles(const AttnGeom& g);
bool attn_row_tiles(const AttnGeom& g);

// floats of scratch mfnet_attn_bwd needs: the dQ kernel hands each query's rel-pos bias rows to the dK/dV kernel through
// it (every key tile used to repeat the table product for all the window's queries)
extern "C" int64_t mfnet_attn_bwd_workspace_floats(int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t ws) {
  if (G <= 0 || Sh <= 0 || Sw <= 0 || heads <= 0) return 0;
  AttnGeom g{};
  g.ws = (int)ws; g.KW = ws > 0 ? (int)ws : (int)Sw; g.KH = ws > 0 ? (int)ws : (int)Sh;
  const int64_t nsg = (g.KH + g.KW + 31) / 32;
  return G * heads * Sh * Sw * (attn_row_tiles(g) ? (int64_t)ATT_WS_LD : 32 * nsg);
}

extern "C" int mfnet_attn_bwd(const void* qkv, const float* qkv_bias, const void* tbl, const void* tblT,
                              const void* out, const void* dout, const float* lse, void* dqkv, int64_t G,
                              int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws, int64_t Lh,
                              int64_t Lw, int64_t NT, float scale, float* workspace, int64_t workspace_floats,
                              hipStream_t stream) {
  MF_CHECK_ARG(qkv && qkv_bias && tbl && tblT && out && dout && lse && dqkv, "attn_bwd: null pointer");
  AttnBwdParams p;
  int rc = attn_fill_geom(&p.g, G, Sh, Sw, heads, hd, ws, Lh, Lw, NT, scale, "attn_bwd");
  if (rc) return rc;
  MF_CHECK_ARG(p.g.KH + p.g.KW <= 128, "attn_bwd: key grid %dx%d too large (KH + KW <= 128)", p.g.KH, p.g.KW);
  p.qkv = (const bf16_t*)qkv; p.qkv_bias = qkv_bias; p.tbl = (const bf16_t*)tbl; p.tblT = (const bf16_t*)tblT;
  p.out = (const bf16_t*)out; p.dout = (const bf16_t*)dout; p.lse = lse; p.dqkv = (bf16_t*)dqkv;
  {   // once per device, thread-safe (see attn.hip)
    static std::mutex mu;
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { mfnet_set_error("attn_bwd: hipGetDevice failed"); return MFNET_ERR_HIP; }
    std::lock_guard<std::mutex> lk(mu);
    if (!done[dev]) {
      const void* fns[16] = {(const void*)attn_bwd_dq_kernel<1, 64, false>, (const void*)attn_bwd_dq_kernel<2, 64, false>, (const void*)attn_bwd_dq_kernel<4, 64, false>,
                             (const void*)attn_bwd_dkv_kernel<1, 64, false>, (const void*)attn_bwd_dkv_kernel<2, 64, false>, (const void*)attn_bwd_dkv_kernel<4, 64, false>,
                             (const void*)attn_bwd_dq_kernel<1, 80, false>, (const void*)attn_bwd_dq_kernel<2, 80, false>, (const void*)attn_bwd_dq_kernel<4, 80, false>,
                             (const void*)attn_bwd_dkv_kernel<1, 80, false>, (const void*)attn_bwd_dkv_kernel<2, 80, false>, (const void*)attn_bwd_dkv_kernel<4, 80, false>,
                             (const void*)attn_bwd_dq_kernel<4, 64, true>, (const void*)attn_bwd_dkv_kernel<4, 64, true>,
                             (const void*)attn_bwd_dq_kernel<4, 80, true>, (const void*)attn_bwd_dkv_kernel<4, 80, true>};
      for (int i = 0; i < 16; ++i) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { mfnet_set_error("attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MFNET_ERR_HIP; }
      }
      done[dev] = true;
    }
  }
  const int nsg = (p.g.KH + p.g.KW + 31) / 32;
  const int rowb = hd <= 64 ? AttHd<64>::ROWB : AttHd<80>::ROWB;
  const bool rowt = attn_row_tiles(p.g);
  const size_t lds_dq = rowt ? (size_t)att_region_a(p.g.NT, rowb, true) + (size_t)ATT_QT * p.g.KH * 4            // K / V + Thq (dThq in place)
                             : (size_t)att_region_a(p.g.NT, rowb) + (size_t)nsg * ATT_KC * ATT_GROW;
  MF_CHECK_ARG(workspace && workspace_floats >= mfnet_attn_bwd_workspace_floats(G, Sh, Sw, heads, ws) && (uintptr_t)workspace % 16 == 0,
               "attn_bwd: needs a 16-byte aligned workspace of mfnet_attn_bwd_workspace_floats(G, Sh, Sw, heads, ws) floats");
  p.bias_ws = workspace;
  const size_t lds_dkv = 2 * (size_t)ATT_QT * rowb + 2 * ATT_QT * 4 + 2 * (rowt ? (size_t)ATT_QT * ATT_TS * 4 : (size_t)nsg * ATT_QT * ATT_GROW);
  MF_CHECK_ARG(lds_dq <= 160 * 1024 && lds_dkv <= 160 * 1024, "attn_bwd: LDS demand too large");
  // the two kernels write disjoint column ranges of dqkv (q | k,v); window mode leaves nothing unwritten
  p.g.ntile = attn_query_tiles(p.g);
  const int64_t nwg = att_grid(p.g.ntile, heads * G);
  MF_CHECK_ARG(nwg < (1ll << 31), "attn_bwd: grid too large");
  dim3 grid_q((unsigned)nwg);
#define ATT_BWD(KERNEL, LDS, HDV)                                                                          \
  do {                                                                                                     \
    if (rowt) hipLaunchKernelGGL((KERNEL<4, HDV, true>), grid_q, dim3(256), LDS, stream, p);               \
    else if (nsg <= 1) hipLaunchKernelGGL((KERNEL<1, HDV, false>), grid_q, dim3(256), LDS, stream, p);     \
    else if (nsg == 2) hipLaunchKernelGGL((KERNEL<2, HDV, false>), grid_q, dim3(256), LDS, stream, p);     \
    else hipLaunchKernelGGL((KERNEL<4, HDV, false>), grid_q, dim3(256), LDS, stream, p);                   \
  } while (0)
  if (hd == 64) ATT_BWD(attn_bwd_dq_kernel, lds_dq, 64); else ATT_BWD(attn_bwd_dq_kernel, lds_dq, 80);
  MF_LAUNCH_CHECK("attn_bwd_dq");
  if (hd == 64) ATT_BWD(attn_bwd_dkv_kernel, lds_dkv, 64); else ATT_BWD(attn_bwd_dkv_kernel, lds_dkv, 80);
  MF_LAUNCH_CHECK("attn_bwd_dkv");
#undef ATT_BWD
  return MFNET_OK;
}
