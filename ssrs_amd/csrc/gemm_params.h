// Parameters shared by the two bf16 GEMM kernels (gemm.hip: 128x128 tiles, 2 workgroups / CU; gemm_big.hip: 256 x 32*NF tiles,
// one 8-wave workgroup / CU with a 2-3 stage LDS ring).
#pragma once
#include "common.h"

#define GEMM_MAX_R 32

struct GemmParams {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  int M, N, K;
  const float* bias;
  const float* U; int64_t ldu;
  const float* V; int64_t v_group_stride; int64_t rows_per_group; int n_groups; int R; float lr_scale;
  int act;                              // 0 none, 1 gelu, 2 *gelu'(aux), 3 relu6, 4 *relu6'(aux = relu6 output), 5 relu6 AFTER the residual add
  const bf16_t* aux; int64_t ldaux;
  bf16_t* out_pre; int64_t ldpre;
  const float* residual; int64_t ldres; int64_t res_row_mod;
  void* out; int out_dtype; int64_t ldc;
  int tiles_m, tiles_n;
  int k_splits;                         // > 1: blockIdx.y owns a K range and stores its fp32 partial into slice blockIdx.y of out
  int64_t w_group_stride, bias_group_stride;   // != 0: W / bias of row group g start g * stride elements further (per-modality layers)
  int conv_H, conv_W, conv_cb;          // conv_cb != 0 (gemm.hip only): A is a [B*conv_H*conv_W, 64*conv_cb] channels-last map and the product
                                        // is its 3x3 / pad 1 convolution, K = 9 * 64 * conv_cb with k = tap * C + c (implicit im2col)
#ifdef GEMM_STAMPS
  unsigned long long* stamps;
  int stagger;
#endif
};

#ifdef GEMM_STAMPS
// Diagnostic build (tools/gemm_stamps.py; never part of libmfnet_hip.so): per-workgroup real-time stamps (100 MHz) at
// entry / first tile landed / main loop done / exit, into a buffer of their own.
extern unsigned long long* g_stamps;
extern int g_stagger;
#define MF_STAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MF_STAMP(i) do {} while (0)
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// tile: 0 = choose, -1 = the 128x128 kernel, 3 / 4 / 6 / 8 / 9 = the 256 x 32*tile kernel. Returns 1 if the big-tile kernel took
// the launch, 0 if the shape is not one of its (caller falls back), < 0 on a launch error.
int gemm_big_launch(const GemmParams& p, int act, bool lora, int tile, hipStream_t stream);
