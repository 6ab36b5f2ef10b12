// Geometry shared by the attention forward / backward kernels.
//
// Reference semantics (medsam/models/ImageEncoder/vit/lora_block.py):
//  * window blocks (:70-79,199-245): the (Sh,Sw) token grid is zero-padded AFTER norm1 to a multiple of
//    ws and cut into ws x ws windows. A padded token's qkv row is exactly the qkv bias (its input is 0,
//    so the LoRA update vanishes too); padded KEYS still enter the softmax with their own rel-pos
//    bias, padded QUERIES are discarded by window_unpartition. We therefore run only the real queries
//    and synthesise padded keys from the bias vector - no 3.06x token inflation.
//  * scores (:181-193,281-317): softmax(scale * q.k + q.Rh[qh,kh] + q.Rw[qw,kw]) with the rel-pos term
//    taken from the UNSCALED q. With tbl = [rel_h ; rel_w] (already resized by get_rel_pos's linear
//    interpolation, :260-271) the bias of (query, key) is T[q][qh-kh+KH-1] + T[q][Lh+qw-kw+KW-1],
//    T = q @ tbl^T : one extra skinny MFMA product per query tile instead of two einsums.
#pragma once
#include "common.h"

#define ATT_HD 64          // head dim (ViT-B / ViT-L)
#define ATT_KC 128         // keys resident in LDS per chunk (small enough for 2-3 workgroups per CU)
#define ATT_QT 64          // queries per workgroup (4 waves x 16)
#define ATT_VT_LD (ATT_KC + 8)   // bf16 elements per row of the transposed [hd][key] images

struct AttnGeom {
  int G, Sh, Sw, heads, D;   // G = image-modalities, token grid Sh x Sw, D = heads*64
  int ws;                    // 0 = global attention
  int nWh, nWw;              // windows per axis (1,1 for global)
  int KH, KW;                // key grid per window (ws,ws) or (Sh,Sw)
  int Lh;                    // rows of the h table inside tbl; w table starts at row Lh
  int NT;                    // padded table rows (multiple of 32, <= 256)
  float scale;
};

struct QueryPos { int valid, tok, qh, qw; };

// query index qi inside (window wi,wj) -> token + in-window coordinates
__device__ __forceinline__ QueryPos query_pos(const AttnGeom& g, int wi, int wj, int qi) {
  QueryPos r;
  int rh = g.Sh, rw = g.Sw, oh = 0, ow = 0;
  if (g.ws > 0) {
    oh = wi * g.ws; ow = wj * g.ws;
    rh = min(g.ws, g.Sh - oh); rw = min(g.ws, g.Sw - ow);
  }
  r.valid = qi < rh * rw;
  const int q = r.valid ? qi : 0;
  r.qh = q / rw; r.qw = q - r.qh * rw;
  r.tok = (oh + r.qh) * g.Sw + (ow + r.qw);
  return r;
}
__device__ __forceinline__ int num_queries(const AttnGeom& g, int wi, int wj) {
  if (g.ws == 0) return g.Sh * g.Sw;
  return min(g.ws, g.Sh - wi * g.ws) * min(g.ws, g.Sw - wj * g.ws);
}
// key index k (row-major over KH x KW) -> token or -1 for a padded position
__device__ __forceinline__ int key_token(const AttnGeom& g, int wi, int wj, int k, int* kh, int* kw) {
  *kh = k / g.KW; *kw = k - *kh * g.KW;
  const int y = (g.ws > 0 ? wi * g.ws : 0) + *kh, x = (g.ws > 0 ? wj * g.ws : 0) + *kw;
  return (y < g.Sh && x < g.Sw) ? y * g.Sw + x : -1;
}

__device__ __forceinline__ int kswz(int row) { return row & 7; }   // 16-B chunk XOR for [row][64 hd] images

// ---- shared device helpers of the forward / backward kernels ------------------------------------------

// T^T = tbl . Q^T for this wave's 16 queries -> Trow[table row] (this lane's query). All table loads are
// issued before the first MFMA (rows beyond NT re-read the last tile), so the product costs one memory
// round trip instead of NT/16.
#define ATT_MAX_NT16 16
__device__ __forceinline__ void att_table_product(const bf16_t* tbl, int NT, const bf16x8_t (&qf)[2], float* Trow, int fr, int fq) {
  const int nt16 = NT >> 4;
  for (int t0 = 0; t0 < nt16; t0 += 6) {
    bf16x8_t a0[6], a1[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int t = min(t0 + i, nt16 - 1);
      const bf16_t* tp = tbl + (int64_t)(16 * t + fr) * ATT_HD + 8 * fq;
      a0[i] = *reinterpret_cast<const bf16x8_t*>(tp);
      a1[i] = *reinterpret_cast<const bf16x8_t*>(tp + 32);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (t0 + i < nt16) {
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], qf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], qf[1], acc, 0, 0, 0);
        *reinterpret_cast<float4*>(Trow + 16 * (t0 + i) + 4 * fq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
    }
  }
}

// 16-B chunk `ch` of a padded key's K and V rows: the qkv bias (fp32) rounded to bf16
struct PaddedKV { u32x4_t k, v; };
__device__ __forceinline__ PaddedKV att_padded_kv(const float* qkv_bias, int D, int head, int ch) {
  const float* bk = qkv_bias + D + head * ATT_HD + ch * 8;
  const float4 a = *reinterpret_cast<const float4*>(bk), b = *reinterpret_cast<const float4*>(bk + 4);
  const float4 c = *reinterpret_cast<const float4*>(bk + D), d = *reinterpret_cast<const float4*>(bk + D + 4);
  PaddedKV r;
  r.k.x = pack_bf16x2(a.x, a.y); r.k.y = pack_bf16x2(a.z, a.w); r.k.z = pack_bf16x2(b.x, b.y); r.k.w = pack_bf16x2(b.z, b.w);
  r.v.x = pack_bf16x2(c.x, c.y); r.v.y = pack_bf16x2(c.z, c.w); r.v.z = pack_bf16x2(d.x, d.y); r.v.w = pack_bf16x2(d.z, d.w);
  return r;
}
