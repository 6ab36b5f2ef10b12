/* libmfnet_hip - C ABI of the MI355X (gfx950) kernels behind the MFNet hot path.
 *
 * The reference (AgPriyanshu/SSRS) has no FFI / plugin layer: its hot path is a chain of ATen ops
 * inside torch.nn.Modules. Every entry point below therefore cites the reference
 * Python code whose arithmetic it replaces (paths relative to the reference root), and
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: device pointers (`void*`, `float*`), int64_t sizes / leading dimensions (in ELEMENTS),
 *     `int` dtype enum, `hipStream_t` last. No torch types.
 *   - return 0 on success, <0 on error (MFNET_ERR_*); the message is in mfnet_last_error()
 *     (thread-local). Nothing throws across the boundary.
 *   - all work is enqueued on `stream`; no entry point synchronises, allocates or frees device
 *     memory, or keeps device state. Caller owns every buffer. Two documented exceptions, both thread-safe: the
 *     attention entry points and mfnet_lora_down_mix (LDS images above 64 KB) raise their kernels' dynamic-LDS limit once
 *     per device (a per-device "done" flag is all that is kept), and mfnet_colsum_static owns a scratch array (see there). The library reads no environment variable.
 *   - "bf16" buffers hold raw bfloat16 bits; matrices are row-major; *_dtype is MFNET_F32 / MFNET_BF16.
 *   - shapes that would make a kernel read or write out of bounds are rejected with MFNET_ERR_ARG
 *     before anything is launched.
 */
#ifndef MFNET_HIP_H
#define MFNET_HIP_H

#include <stdint.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFNET_OK 0
#define MFNET_ERR_ARG (-1)
#define MFNET_ERR_HIP (-2)
#define MFNET_ERR_UNSUPPORTED (-3)

#define MFNET_F32 0
#define MFNET_BF16 1

int mfnet_version(void);              /* 100 = 0.1.0 */
const char* mfnet_last_error(void);   /* message of the last failing call on this thread */

/* ---- LayerNorm over the last dim (eps 1e-6): nn.LayerNorm norm1/norm2 of the ViT blocks
 * (medsam/models/ImageEncoder/vit/lora_block.py:67-68,83-84; adapter_fusionblock.py:78-79,101-103),
 * LayerNorm2d of the neck (medsam/models/common/layer_norm.py:12-17) and Norm2d of fpn1x
 * (unetformer_mmsam.py:14-22), all computed channels-last. x [rows, C] (ld = ldx) fp32|bf16 ->
 * y fp32|bf16; mean/rstd [rows] fp32 optional (saved for backward). C % 4 == 0, C <= 2048. */
int mfnet_layernorm_fwd(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta,
                        void* y, int y_dtype, int64_t ldy, float* mean, float* rstd, int64_t rows,
                        int64_t C, float eps, hipStream_t stream);
/* dx = LN'(dy) (+ dres: the residual-stream gradient, fused), written as fp32 (dx) and/or bf16
 * (dx_bf16, ld = lddx); dgamma/dbeta [C] fp32 are ACCUMULATED into when non-null. */
/* layernorm_fwd fused with the LoRA down-projection of the layer that consumes it (loralib/layers.py:157-158, 253, 261): y bf16,
 * a_out[row, j] = sum_c y[row, c] * A[g][j, c] with A fp32 [n_groups][R][C] (group stride a_group_stride floats; n_groups = 2:
 * the first rows/2 rows are modality x, the rest y), R in {4, 8}. With U != NULL (lora.Linear, n_groups = 2) also the
 * cross-modal mix of layers.py:159-160: U [rows, 2R] = [wx*ax, (1-wx)*ay] for x rows, [(1-wy)*ax, wy*ay] for y rows. */
int mfnet_layernorm_fwd_lora(const void* x, int x_dtype, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                             float* mean, float* rstd, int64_t rows, int64_t C, float eps, const float* A, int64_t a_group_stride,
                             int n_groups, int R, float* a_out, const float* wx, const float* wy, float* U, hipStream_t stream);
/* the same down-projection (+ mix) for a layer whose input is not a LayerNorm output (lin2 reads GELU(lin1)): x bf16 [rows, C],
 * C % 8 == 0, C <= 4096; replaces mfnet_skinny_nt + mfnet_lora_mix_fwd on that path. */
int mfnet_lora_down_mix(const void* x, int64_t ldx, int64_t rows, int64_t C, const float* A, int64_t a_group_stride, int n_groups, int R,
                        float* a_out, const float* wx, const float* wy, float* U, hipStream_t stream);
int mfnet_layernorm_bwd(const void* dy, int dy_dtype, int64_t lddy, const void* x, int x_dtype, int64_t ldx,
                        const float* mean, const float* rstd, const float* gamma, const float* dres,
                        float* dx, void* dx_bf16, int64_t lddx, float* dgamma, float* dbeta, int64_t rows,
                        int64_t C, hipStream_t stream);

/* ---- bf16 MFMA GEMM with fused epilogues: every dense contraction of the encoder.
 *   out[m,n] = act( sum_k A[m,k] W[n,k] + bias[n] + lr_scale * sum_j U[m,j] V[g(m)][n,j] ) + residual[m % res_row_mod, n]
 * A [M,K] bf16 (lda), W [N,K] bf16 (ldw) - F.linear layout, fp32 accumulation. K % 32 == 0, N % 16 == 0.
 * Low-rank term (fp32 U [M,R] (ldu), V [n_groups][N,R], g(m) = (m / rows_per_group) % n_groups, R in
 * {4,8,12,16}): the LoRA update of lora.MergedLinear (medsam/models/common/loralib/layers.py:244-268,
 * lr_scale = 1/r) and of lora.Linear incl. the cross-modal mix (layers.py:149-165), and their dgrads.
 * act: 0 none | 1 exact GELU (out_pre, if non-null, receives the bf16 pre-activation) | 2 multiply by
 * GELU'(aux[m,n]) (aux bf16: the saved pre-activation; MLP backward). With act 0, out_pre receives the
 * value before the residual add. residual fp32 (ldres), may alias a fp32 `out`; res_row_mod > 0
 * broadcasts a [res_row_mod, N] table over the rows (abs-pos-embed add, image_encoder.py:126-135).
 * Replaces: qkv / proj (lora_block.py:169-196), MLPBlock (lora_block.py:104-108), Adapter fc1/fc2
 * (common/adapter.py:16-25), PatchEmbed as im2col GEMM (image_encoder.py:189-197), neck convs (:104-120),
 * and the transposed-weight dgrads of all of them. k_splits > 1 (weight gradients: tiny M x N, K = number of
 * pixels): the K range is cut over gridDim.y and slice s stores its fp32 partial at out + s*M*ldc, i.e. `out` is
 * a [k_splits][M][ldc] scratch that mfnet_reduce_splits sums (fp32 out, no bias / low-rank / act / residual). */
int mfnet_gemm_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N, int64_t K,
                    const float* bias, const float* U, int64_t ldu, const float* V, int64_t v_group_stride,
                    int64_t rows_per_group, int n_groups, int R, float lr_scale, int act, const void* aux,
                    int64_t ldaux, void* out_pre, int64_t ldpre, const float* residual, int64_t ldres,
                    int64_t res_row_mod, void* out, int out_dtype, int64_t ldc, int k_splits, hipStream_t stream);
/* The same call with the tile shape named by the caller (benchmarks and tests; the reference has no counterpart - its GEMMs are
 * ATen's): tile 0 = the library's choice (what mfnet_gemm_bf16 does), -1 = the 128x128-tile kernel (2 workgroups / CU), 3 / 4 / 6 /
 * 8 / 9 = the 256 x (32*tile)-tile kernel (one 8-wave workgroup / CU, 2-3 stage LDS ring; needs M % 256 == 0, K % 64 == 0,
 * N % (32*tile) == 0, k_splits == 1, else MFNET_ERR_ARG). */
int mfnet_gemm_bf16_tiled(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t M, int64_t N, int64_t K,
                    const float* bias, const float* U, int64_t ldu, const float* V, int64_t v_group_stride,
                    int64_t rows_per_group, int n_groups, int R, float lr_scale, int act, const void* aux,
                    int64_t ldaux, void* out_pre, int64_t ldpre, const float* residual, int64_t ldres,
                    int64_t res_row_mod, void* out, int out_dtype, int64_t ldc, int k_splits, int tile, hipStream_t stream);
/* The same GEMM with per-group weights: rows [g*rows_per_group, (g+1)*rows_per_group) (g cyclic over n_groups) use
 * W + g*w_group_stride and bias + g*bias_group_stride (elements; multiples of 8 / 4). One launch for the per-modality
 * dense layers of the MMAdapter block - Img_Adapter / DSM_Adapter and MLPx_ / MLPy_Adapter, adapter.py:16-25 applied at
 * adapter_fusionblock.py:88-108 - instead of two half-batch GEMMs. No low-rank term, no split-K. */
int mfnet_gemm_bf16_grouped(const void* A, int64_t lda, const void* W, int64_t ldw, int64_t w_group_stride, int64_t M,
                            int64_t N, int64_t K, const float* bias, int64_t bias_group_stride, int64_t rows_per_group,
                            int n_groups, int act, const void* aux, int64_t ldaux, void* out_pre, int64_t ldpre,
                            const float* residual, int64_t ldres, void* out, int out_dtype, int64_t ldc,
                            hipStream_t stream);

/* ---- skinny products of the low-rank paths (layers.py:149-165,244-268 forward "x @ A^T" and the
 * wgrads of lora_A / lora_B). nt (MFMA): out[m,j] = scale * sum_k X[m,k] Wt[g(m)][j,k], X bf16 [M,K], Wt fp32
 * [n_groups][R,K] (group stride w_group_stride elements; g(m) = (m / rows_per_group) % n_groups: the per-modality
 * LoRA matrices), R <= 16, K % 32 == 0, ldo % 4 == 0. tn: for each group z < n_groups, out_z[c,j] (or out_z[j,c]
 * if transposed) += scale * sum_{m < M} Y[z*M+m, c] U[z*M+m, j]; Y bf16, U fp32 [.,R], out_z = out +
 * z*out_group_stride (accumulates: zero `out` first). tn with out == NULL (a workspace is then required): only the
 * partial tiles are produced, workspace[z][s][c][j] for s < splits = workspace_floats(M, C, R, n_groups) / (n_groups*C*R),
 * unscaled; their sum is left to mfnet_gather_segments (red_n = splits). */
int mfnet_skinny_nt(const void* X, int64_t ldx, const float* Wt, int64_t ldw, int64_t w_group_stride, int64_t rows_per_group,
                    int n_groups, float* out, int64_t ldo, int64_t M, int64_t K, int R, float scale, hipStream_t stream);
int mfnet_skinny_tn(const void* Y, int64_t ldy, const float* U, int64_t ldu, float* out, int64_t M, int64_t C,
                    int R, int transposed, float scale, int n_groups, int64_t out_group_stride, float* workspace,
                    int64_t workspace_floats, hipStream_t stream);
/* scratch floats that make mfnet_skinny_tn atomic-free and deterministic (two-stage row reduction); with a
 * null workspace the row splits meet in float atomics on `out`. */
int64_t mfnet_skinny_tn_workspace_floats(int64_t M, int64_t C, int R, int n_groups);

/* ---- LoRA cross-modal mix (layers.py:159-160): a [M,r] fp32 = [x@Ax^T ; y@Ay^T] (RGB rows first) ->
 * U [M,2r]: RGB rows [wx*ax, (1-wx)*ay], DSM rows [(1-wy)*ax, wy*ay]; backward returns da and ADDS
 * (d lora_wx, d lora_wy) into dw[2]. wx/wy: device pointers to the trainable scalars. */
int mfnet_lora_mix_fwd(const float* a, const float* wx, const float* wy, float* U, int64_t M, int r, hipStream_t stream);
/* P = the sum of `nparts` partial matrices [M,2r], part_stride floats apart (nparts = 1: P itself; the column strips of
 * mfnet_lora_side otherwise). */
int mfnet_lora_mix_bwd(const float* P, int nparts, int64_t part_stride, const float* a, const float* wx, const float* wy, float* da,
                       float* dw, int64_t M, int r, hipStream_t stream);

/* ---- both rank-r products of a LoRA layer's backward in ONE pass over the output gradient Y [n_groups*Mg, C] bf16
 * (loralib/layers.py:149-165, 244-268 differentiated; group g = rows [g*Mg, (g+1)*Mg) = one modality):
 *   p_part[strip][m][j]     = sum_{c in strip} Y[m,c] * Vt[g][j][c]   (strips of 256 columns; [ceil(C/256)][n_groups*Mg][8] fp32,
 *                             j >= RP are zero) - its sum over the strips is Y . B, the gradient w.r.t. the down-projection x A^T;
 *   dv_part[g][s][c][j]     = sum_{m in split s} Y[m,c] * U[m][j]     ([n_groups][mfnet_lora_side_splits(Mg, C)][C][R] fp32) - its
 *                             sum over the splits is dB = Y^T . u.
 * U (fp32 [n_groups*Mg, R], R = 4 or 8, row stride ldu) or Vt (fp32 [RP <= 8][C] per group, vt_group_stride floats apart) may be
 * NULL: then only the other product is computed (Vt == NULL, U = d(xA^T): dA^T = x^T . da of the same layer). C % 64 == 0. */
int64_t mfnet_lora_side_splits(int64_t Mg, int64_t C);
int mfnet_lora_side(const void* Y, int64_t ldy, int64_t Mg, int64_t C, int n_groups, const float* U, int64_t ldu, int R,
                    float* dv_part, const float* Vt, int64_t ldvt, int64_t vt_group_stride, int RP, float* p_part,
                    hipStream_t stream);
/* Several column products dV = Y^T . U of mfnet_lora_side (no Vt / P) in one launch - the dA = x^T . da products of the LoRA
 * layers' backward (loralib/layers.py:149-165 differentiated w.r.t. lora_A), which feed nothing but the optimizer step. desc: 8
 * int64 per problem - Y, ldy, Mg, C, n_groups, U, ldu, dv_part; one R (4 or 8) for all. Results per problem as mfnet_lora_side. */
int mfnet_lora_side_batch(const int64_t* desc, int64_t n, int R, hipStream_t stream);

/* ---- fused attention of the SAM blocks (lora_block.py:169-196 / adapter_fusionblock.py:176-198 with
 * add_decomposed_rel_pos :281-317, window_partition / window_unpartition :199-245).
 * qkv [G*Sh*Sw, 3*heads*hd] bf16 (q | k | v thirds, head-major), G = images x modalities, hd = 64 (ViT-B / ViT-L) or
 * 80 (ViT-H).
 * ws = 0: global attention over the Sh x Sw grid; ws > 0: ws x ws windows of the grid zero-padded at the
 * bottom/right; only real queries are computed, padded keys/values are synthesised from qkv_bias (their
 * input row is exactly zero after norm1), so nothing is inflated to the padded token count.
 * tbl [NT, hdp] bf16 = [rel_h resized to Lh = 2*KH-1 rows ; rel_w resized to Lw = 2*KW-1 rows ; zero rows], hdp = hd
 * rounded up to a multiple of 32 with zero columns (64 -> 64, 80 -> 96),
 * NT % 32 == 0 (get_rel_pos' linear interpolation is done once on the host side, the tables are frozen);
 * scores = scale * q.k + q.rel_h[qh-kh+KH-1] + q.rel_w[qw-kw+KW-1] with the UNSCALED q in the bias terms.
 * Online softmax: no N x N tensor is materialised. out [G*Sh*Sw, heads*hd] bf16, lse [G,heads,Sh*Sw] fp32. */
int mfnet_attn_fwd(const void* qkv, const float* qkv_bias, const void* tbl, void* out, float* lse, int64_t G,
                   int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws, int64_t Lh, int64_t Lw,
                   int64_t NT, float scale, hipStream_t stream);
/* dqkv [G*Sh*Sw, 3D] bf16 from dout; recomputes probabilities from lse. tblT [hd,NT] bf16 = tbl[:, :hd]^T.
 * Key grids up to KH + KW <= 128 (token grids up to 64 x 64). */
int mfnet_attn_bwd(const void* qkv, const float* qkv_bias, const void* tbl, const void* tblT, const void* out,
                   const void* dout, const float* lse, void* dqkv, int64_t G, int64_t Sh, int64_t Sw,
                   int64_t heads, int64_t hd, int64_t ws, int64_t Lh, int64_t Lw, int64_t NT, float scale,
                   float* workspace, int64_t workspace_floats, hipStream_t stream);
/* scratch of mfnet_attn_bwd (required, 16-byte aligned): the dQ kernel leaves each query's rel-pos bias rows there and
 * the dK/dV kernel, which visits every query of the window once per 64-key tile, reads them instead of repeating the
 * table product. */
int64_t mfnet_attn_bwd_workspace_floats(int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t ws);

/* ---- data movement of the encoder. cast: fp32 <-> bf16, n % 8 == 0. im2col_patch: rgb [B,3,H,W] fp32 and
 * dsm [B,H,W] fp32 (dsm_channels = 1: replicated to 3 channels, unetformer_mmsam.py:535) or [B,3,H,W] (dsm_channels = 3: a
 * second 3-channel image, the general ImageEncoderViT.forward(x, y) of image_encoder.py:122-144) -> cols [2*B*(H/P)*(W/P), 3*P*P]
 * bf16, RGB rows first, k in conv-weight order (image_encoder.py:189-197). im2col3x3 / col2im3x3: NHWC 3x3
 * pad-1 unfold / fold for the neck conv and its dgrad (image_encoder.py:112-118). transpose_bf16:
 * x [M,N] -> y [N,Mp] zero-padded (operand prep for adapter wgrads). */
int mfnet_cast(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, hipStream_t stream);
int mfnet_im2col_patch(const float* rgb, const float* dsm, int dsm_channels, void* cols, int64_t B, int64_t H, int64_t W, int64_t P,
                       hipStream_t stream);
int mfnet_im2col3x3(const void* x, void* cols, int64_t G, int64_t H, int64_t W, int64_t C, hipStream_t stream);
int mfnet_col2im3x3(const void* dcols, float* dx, int64_t G, int64_t H, int64_t W, int64_t C, hipStream_t stream);
int mfnet_transpose_bf16(const void* x, int64_t ld, void* y, int64_t M, int64_t N, int64_t Mp, hipStream_t stream);
/* out[i] (= | +=) scale * sum_s partial[s*n + i], n % 4 == 0: second stage of the split-K weight gradients. */
/* Weight-gradient GEMM out[n][k] = sum_m dy[m][n] * x[m][k] on the row-major activations themselves (bf16, row
 * strides ldy / ldx), fp32 result: the wgrad of every trainable conv / linear of the decoder, FPN and adapters
 * (what autograd's conv / linear backward computes for unetformer_mmsam.py:43-84, adapter.py:17-25). The row range
 * is cut into `splits` (mfnet_gemm_tn_splits gives a good count); split s writes its partial [N][ldo] tile at
 * out + s*N*ldo, to be summed by mfnet_reduce_splits (splits == 1: out is the result).
 * bias_part (nullable): [splits][N] fp32, row s = column sums of dy over split s's rows - the bias gradient of the same
 * layer from the fragments already in registers (its partial rows are summed like the weight partials). */
int mfnet_gemm_tn_bf16(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t M, int64_t N, int64_t K,
                       float* out, int64_t ldo, int splits, float* bias_part, hipStream_t stream);
/* n_groups independent products in one launch: group g multiplies rows [g*M, (g+1)*M) of dy and x (M = rows per group);
 * partial tile (split s, group g) goes to out + (s*n_groups + g)*N*ldo. The per-modality adapter weight gradients
 * (adapter.py:13-14 D_fc1 / D_fc2 of Img_ / DSM_ and MLPx_ / MLPy_Adapter). bias_part (nullable): [splits][n_groups][N]. */
int mfnet_gemm_tn_bf16_grouped(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t M, int64_t N, int64_t K,
                               float* out, int64_t ldo, int splits, int n_groups, float* bias_part, hipStream_t stream);
/* Several of the products above in ONE launch (they are independent of each other and of the backward chain: the weight
 * gradients autograd would compute node by node for unetformer_mmsam.py:43-84 / adapter.py:17-25 only have to exist before the
 * optimizer step, trainer.py:170-172). desc: 16 int64 per problem - dy, ldy, x, ldx, M (rows per group), N, K, out, ldo, splits,
 * n_groups, bias_part (0: none), conv_H, conv_W, conv_C (0: a plain product; else x is a channels-last map and K = 9 * conv_C as
 * in mfnet_gemm_tn_conv3x3), 0. Results and partial-tile layout per problem exactly as the one-problem entry points. */
int mfnet_gemm_tn_batch(const int64_t* desc, int64_t n, hipStream_t stream);
/* 3x3 / stride 1 / zero-pad 1 convolution of a channels-last bf16 map as an implicit GEMM (no column tensor): x [B*H*W, C]
 * (row stride ldx, C % 64 == 0), w [N, 9*C] with k = tap*C + c (tap = 3*ky + kx), out [B*H*W, N] bf16 or fp32: the 3x3 convs of
 * the decoder and the neck (unetformer_mmsam.py:43-60 ConvBNReLU / ConvBN / Conv with kernel_size=3, image_encoder.py:112-118).
 * The data gradient is the same call on dy with flipped, transposed weights [C, 9*N] (k = (8 - tap)*N + n);
 * mfnet_gemm_tn_conv3x3 is the weight gradient out[n][tap*C + c] = sum_p dy[p][n] * x[p shifted by tap][c]
 * (splits / partial tiles / bias_part as mfnet_gemm_tn_bf16 with M = B*H*W, K = 9*C). */
int mfnet_conv3x3_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, int64_t B, int64_t H, int64_t W, int64_t C,
                       int64_t N, void* out, int out_dtype, int64_t ldc, hipStream_t stream);
int mfnet_gemm_tn_conv3x3(const void* dy, int64_t ldy, const void* x, int64_t ldx, int64_t B, int64_t H, int64_t W, int64_t C,
                          int64_t N, float* out, int64_t ldo, int splits, float* bias_part, hipStream_t stream);
int64_t mfnet_gemm_tn_splits(int64_t M, int64_t N, int64_t K);
int mfnet_reduce_splits(const float* partial, float* out, int64_t n, int splits, float scale, int accumulate, hipStream_t stream);

/* Strided gather of n small fp32 tensors in one launch (per 48 segments). desc is a HOST array of n x 17
 * int64: {src device ptr, dst device ptr, dims[4], src strides[4], dst strides[4], red_n, red_stride, scale} with
 * strides in elements and scale = the bits of a float in the low 32 bits; the value moved is
 * scale * sum_{k < red_n} src[... + k*red_stride] (red_n = 1, scale = 1.0f: a plain copy), which folds the sum over the
 * partial tiles of a split reduction (mfnet_skinny_tn with out == NULL, mfnet_gemm_tn_bf16) into the delivery;
 * dst_dtype MFNET_F32 (assign, or += when accumulate != 0) or MFNET_BF16 (assign). Replaces the framework
 * copies the reference's step performs implicitly: weight casts / permutes ahead of every conv and linear
 * (unetformer_mmsam.py:43-84 Conv/ConvBN/ConvTranspose modules, loralib/layers.py:149-165 B@A products) and
 * autograd's per-parameter gradient accumulation (trainer.py:160 loss.backward()). */
int mfnet_gather_segments(const int64_t* desc, int n, int dst_dtype, int accumulate, hipStream_t stream);

/* ---- MMAdapter glue (adapter_fusionblock.py:104-108): out = h + w*ada_own + (1-w)*ada_other per modality
 * (rows [0,M/2) RGB, [M/2,M) DSM); backward gives d ada (bf16) and ADDS (d wx_Adapter, d wy_Adapter) into
 * dw[16][2]: sixteen slots the caller sums (mfnet_gather_segments with red_n = 16), so no address takes more than a few
 * dozen float atomics. gelu_fwd/bwd: exact GELU; add_bf16: elementwise sum; bias_grad: out[n] += sum_m dy[m,n]. */
int mfnet_adapter_mix(const float* h, const float* ada, const float* wx, const float* wy, float* out, void* reserved,
                      int64_t M, int64_t D, hipStream_t stream);
int mfnet_adapter_mix_bwd(const float* dh, const float* ada, const float* wx, const float* wy, void* dada, float* dw,
                          void* reserved, int64_t M, int64_t D, hipStream_t stream);
int mfnet_gelu_fwd(const float* x, float* y, int64_t n, hipStream_t stream);
int mfnet_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t stream);
int mfnet_add_bf16(const void* a, const void* b, void* out, int64_t n, hipStream_t stream);
int mfnet_bias_grad(const void* dy, int64_t ld, float* out, int64_t M, int64_t N, hipStream_t stream);

/* ---- loss / optimiser. CrossEntropy2d (utils2.py:115-127; ignore_index variant
 * train_multiscale_main.py:30-43) on NCHW fp32 logits without the transpose copy. fwd ADDS into acc[4] =
 * {sum w[t]*nll, sum w[t], #(argmax == t), #valid} (zero it first; loss = acc[0]/acc[1]) and optionally writes
 * the arg-max class per pixel (uint8). bwd: dlogits = gscale * w[t] * (softmax - onehot) / acc[1].
 * sgd_flat: torch.optim.SGD of trainer.py:111-116 over one flat buffer: d = gscale*g + wd*p;
 * buf = momentum*buf + d; p -= lr*buf (buf zero-initialised; gscale = 1/world folds the DP average). */
int mfnet_ce_loss_fwd(const float* logits, const int64_t* target, const float* weight, float* acc,
                      unsigned char* pred, int64_t B, int64_t C, int64_t HW, int64_t ignore_index, hipStream_t stream);
int mfnet_ce_loss_bwd(const float* logits, const int64_t* target, const float* weight, const float* acc,
                      float* dlogits, float gscale, int64_t B, int64_t C, int64_t HW, int64_t ignore_index,
                      hipStream_t stream);
int mfnet_sgd_flat(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float wd,
                   float gscale, hipStream_t stream);

/* ---- sliding-window inference (evaluator.py:113-176, inference.py:535-568). tile_accumulate:
 * pred [C,H,W] fp32 += logits [n,C,h,w] at (row, col) = coords[n][2] int32 - the reference's
 * `pred[x:x+w, y:y+h] += out` overlap merge, on the device (atomic != 0 allows overlapping tiles within one
 * call). argmax_chw: uint8 class map (first maximum wins, like np.argmax). minmax + scale_shift: DSM
 * min-max normalisation (mode 0 evaluator.py:100-111 whole raster; mode 1 inference.py:547-555 per tile,
 * zero when max == min). */
int mfnet_tile_accumulate(float* pred, const float* logits, const int* coords, int64_t n, int64_t C, int64_t H,
                          int64_t W, int64_t h, int64_t w, int atomic, hipStream_t stream);
int mfnet_argmax_chw(const float* pred, unsigned char* out, int64_t C, int64_t N, hipStream_t stream);
/* float64-accumulator twins of the two above: the reference merges window logits in a float64 numpy array (evaluator.py:134,
 * 169-176); sums of a few dozen fp32 logits are exact there, so the merged scores and the argmax mask do not depend on the
 * order the (atomic) adds land in. src: fp32 logits (src_f64 = 0) or a float64 band from a neighbouring rank (src_f64 = 1). */
int mfnet_tile_accumulate_f64(double* pred, const void* src, int src_f64, const int* coords, int64_t n, int64_t C, int64_t H,
                              int64_t W, int64_t h, int64_t w, int atomic, hipStream_t stream);
int mfnet_argmax_chw_f64(const double* pred, unsigned char* out, int64_t C, int64_t N, hipStream_t stream);
int mfnet_minmax(const float* x, float* out, int64_t n, hipStream_t stream);
int mfnet_scale_shift(const float* x, float* y, const float* mm, float unused, int64_t n, int mode, hipStream_t stream);

/* ===================== FPN / SE-fusion / UNetFormer decoder (unetformer_mmsam.py:24-563) =====================
 * Channels-last maps flattened to [rows = B*H*W, C]; *_dtype selects fp32 / bf16 storage per tensor. The
 * decoder's dense contractions (1x1 convs :43-62, ConvTranspose2d k2s2 :494-516 as a GEMM with N = 4*C,
 * 3x3 convs through mfnet_im2col3x3) run on mfnet_gemm_bf16 (act 3 = ReLU6, 4 = multiply by the ReLU6 mask
 * of aux, 5 = ReLU6 after the residual add); the kernels below are the HBM-bound remainder. */

/* per-(group, channel) sums over the P rows of each of G groups: sum [G,C] += sum_p x, sumsq += sum_p x*x
 * (sumsq may be null). BatchNorm statistics (G = 1), F.adaptive_avg_pool2d of SqueezeAndExcitation (:262)
 * and of FeatureRefinementHead.ca (:333). coldot: out [G,C] += sum_p a*b (gate gradients). Zero outputs first.
 * The reductions are deterministic (no float atomics): `workspace` holds the workgroups' partial slabs
 * (mfnet_colreduce_workspace_floats(G, P, C) floats, contents irrelevant) and `tickets` G unsigned ints that must
 * be ZERO on entry and are zero again on completion (so one zeroed buffer serves every later call on the stream). */
int64_t mfnet_colreduce_workspace_floats(int64_t G, int64_t P, int64_t C);
int mfnet_colsum(const void* x, int x_dtype, float* sum, float* sumsq, int64_t G, int64_t P, int64_t C,
                 float* workspace, unsigned int* tickets, hipStream_t stream);
/* mfnet_colsum without a workspace argument, for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc. It has NO counterpart in the reference (neither has mfnet_colsum's workspace: the
 * reference's sums are ATen reductions) - it exists for those callers only. Same kernels and bit-identical results; the partial
 * slabs and tickets live in a file-scope device array of the library (4 MiB + 4 KiB per GPU the library runs on, held until
 * the process ends). Limits: mfnet_colreduce_workspace_floats(G, P, C) <= 1048576 floats and G <= 1024, else MFNET_ERR_ARG
 * (every BatchNorm / SE / FRH reduction of MFNet at 16 tiles per GPU fits). It calls no hipMalloc / hipFree; the tickets are
 * cleared by a one-workgroup kernel on `stream`. All calls on one GPU share that GPU's array: calls that overlap in time on one
 * GPU (different streams, different threads) race - serialise them on one stream. The first call on each device looks the
 * array's address up (hipGetSymbolAddress): make it outside any stream capture. */
int mfnet_colsum_static(const void* x, int x_dtype, float* sum, float* sumsq, int64_t G, int64_t P, int64_t C, hipStream_t stream);
int mfnet_coldot(const void* a, int a_dtype, const void* b, int b_dtype, float* out, int64_t G, int64_t P, int64_t C,
                 float* workspace, unsigned int* tickets, hipStream_t stream);
/* LayerNorm weight gradients, deterministic: dbeta [C] += sum_rows dy ; dgamma [C] += sum_rows dy * (x - mean[row]) * rstd[row]
 * (the affine parameters of nn.LayerNorm / Norm2d, unetformer_mmsam.py:14-22, 499-503). dy / x are contiguous [rows, C]
 * (ld == C); workspace = mfnet_colreduce_workspace_floats(1, rows, C) floats and one zeroed ticket. */
int mfnet_layernorm_bwd_weights(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                                float* dgamma, float* dbeta, int64_t rows, int64_t C, float* workspace, unsigned int* tickets,
                                hipStream_t stream);

/* nn.BatchNorm2d (:27-73,126-129,508). finalize: training != 0: mean / rstd from the batch sums (biased
 * variance), running_mean / running_var updated with `momentum` (unbiased variance), num_batches_tracked += 1;
 * training == 0: mean / rstd from the running statistics. apply: y = act((x-mean)*rstd*gamma+beta) *
 * chan_scale[row / rows_per_image] with act 0 none | 1 ReLU6 (ConvBNReLU) | 2 GELU; chan_scale (optional,
 * [B,C]) is the Dropout2d mask of the segmentation head (:440); y2 is an optional second copy in another
 * dtype. bwd_reduce: dsum [C] += sum dy', dxh [C] += sum dy'*xhat (= dbeta, dgamma) with dy' = dy * chan_scale
 * * act'(.); bwd_apply: dx = gamma*rstd*(dy' - [training](dsum + xhat*dxh)/rows). */
int mfnet_bn_finalize(const float* sum, const float* sumsq, float* mean, float* rstd, float* run_mean, float* run_var,
                      int64_t* num_batches_tracked, int64_t C, int64_t n_rows, float eps, float momentum, int training,
                      hipStream_t stream);
/* finalize (training) + apply in ONE launch: mean / rstd are derived from the column sums by every workgroup and stored
 * (with the running-statistics update) by the first; same arithmetic as the two calls above. C <= 1024. */
int mfnet_bn_train_apply(const void* x, int x_dtype, const float* sum, const float* sumsq, float* mean, float* rstd, float* run_mean,
                         float* run_var, int64_t* num_batches_tracked, float eps, float momentum, const float* gamma,
                         const float* beta, int act, const float* chan_scale, int64_t rows_per_image, void* y, int y_dtype, void* y2,
                         int y2_dtype, int64_t rows, int64_t C, hipStream_t stream);
int mfnet_bn_apply(const void* x, int x_dtype, const float* mean, const float* rstd, const float* gamma, const float* beta,
                   int act, const float* chan_scale, int64_t rows_per_image, void* y, int y_dtype, void* y2, int y2_dtype,
                   int64_t rows, int64_t C, hipStream_t stream);
int mfnet_bn_bwd_reduce(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, int act, const float* chan_scale, int64_t rows_per_image,
                        float* dsum, float* dxh, int64_t rows, int64_t C, float* workspace, unsigned int* tickets,
                        hipStream_t stream);   /* workspace / tickets as for mfnet_colsum with G = 1 */
int mfnet_bn_bwd_apply(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, int act, const float* chan_scale, int64_t rows_per_image,
                       const float* dsum, const float* dxh, int training, void* dx, int dx_dtype, int64_t rows, int64_t C,
                       hipStream_t stream);

/* elementwise activation (0 none | 1 ReLU6 | 2 exact GELU) and its backward dx = dy * act'(ref) (nn.GELU of
 * fpn1 :497,509; the final nn.ReLU6 of FeatureRefinementHead :353). n % 4 == 0. */
int mfnet_act(const void* x, int x_dtype, void* y, int y_dtype, int act, int64_t n, hipStream_t stream);
int mfnet_act_bwd(const void* dy, int dy_dtype, const void* ref, int ref_dtype, void* dx, int dx_dtype, int act, int64_t n,
                  hipStream_t stream);

/* depthwise k x k conv, zero padding (k-1)/2, x / y [B,H,W,C] fp32, w [C, k*k] (nn.Conv2d(groups=C) weight):
 * FeatureRefinementHead.pa (k 3, bias, :329-331), SeparableConvBN (:64-73) of FRH.proj (k 3) and of
 * GlobalLocalAttention.proj (k 8). ext != 0 folds GLA's `F.pad(out, (0,1,0,1), 'reflect')` (:192-196) into the
 * indexing. bwd_weight ADDS into dw [slots][C, k*k] and dbias [slots][C] (optional): workgroup i adds into copy i % slots
 * and the caller sums the copies (mfnet_gather_segments with red_n = slots; slots = 1: the plain gradient) - float atomics
 * on one address cost ~140 ns each on this chip. */
int mfnet_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int64_t B, int64_t H, int64_t W, int64_t C,
                     int64_t k, int ext, hipStream_t stream);
int mfnet_dwconv_bwd_data(const void* dy, int dy_dtype, const float* w, float* dx, int64_t B, int64_t H, int64_t W, int64_t C,
                          int64_t k, int ext, hipStream_t stream);
int mfnet_dwconv_bwd_weight(const void* dy, int dy_dtype, const float* x, float* dw, float* dbias, int64_t B, int64_t H,
                            int64_t W, int64_t C, int64_t k, int ext, int slots, hipStream_t stream);
/* Several depthwise weight gradients in one launch (independent of the backward chain, collected by the host like
 * mfnet_gemm_tn_batch; what autograd computes node by node for the depthwise convs of unetformer_mmsam.py:52-84,117-131).
 * desc: 12 int64 per problem - dy, dy_dtype, x, dw, dbias (0: none), B, H, W, C, k, ext, slots. */
int mfnet_dwconv_bwd_weight_batch(const int64_t* desc, int64_t n, hipStream_t stream);

/* GlobalLocalAttention's window attention (:169-186): qkv [B*H*W, 3*heads*8] bf16 in the 1x1 qkv conv's
 * channel order (which, head, d); 8x8 windows (H, W multiples of 8), softmax(q.k*scale +
 * relative_position_bias_table[index(i,j)][head]).v -> out [B*H*W, heads*8] fp32. bwd writes dqkv (bf16) and
 * ADDS the table gradient into dtable [225, heads]. Only hd = 8, ws = 8 are implemented. */
int mfnet_gla_attn_fwd(const void* qkv, const float* table, float* out, int64_t B, int64_t H, int64_t W, int64_t heads,
                       int64_t hd, int64_t ws, float scale, hipStream_t stream);
int mfnet_gla_attn_bwd(const void* qkv, const float* table, const float* dout, void* dqkv, float* dtable, int64_t B,
                       int64_t H, int64_t W, int64_t heads, int64_t hd, int64_t ws, float scale, hipStream_t stream);
/* out = AvgPool2d((8,1),1,(3,0))(pad_bottom_reflect(o)) + AvgPool2d((1,8),1,(0,3))(pad_right_reflect(o)) + l1 + l2
 * (:187-191; attn_x / attn_y / local); bwd returns d o (d l1 = d l2 = ds). All fp32 [B,H,W,C]. */
int mfnet_gla_pool_fwd(const float* o, const float* l1, const float* l2, float* out, int64_t B, int64_t H, int64_t W,
                       int64_t C, hipStream_t stream);
int mfnet_gla_pool_bwd(const float* ds, float* d_o, int64_t B, int64_t H, int64_t W, int64_t C, hipStream_t stream);

/* WF / FeatureRefinementHead fusion (:242-246,342-345): out[b,Y,X] = fw0*skip[perm(b,Y,X)] + fw1*bilinear_x2(x),
 * fw = relu(w)/(relu(w0)+relu(w1)+eps) read from the device pointer w[2]; x [B,H/2,W/2,C] fp32. `perm`: row
 * order of skip = number of ConvTranspose(k2,s2) GEMMs stacked above the raster grid (0 raster, 1, 2), so no
 * pixel-shuffle pass exists. bwd: dx (fp32), dskip (skip's order), dw[2] += fusion-weight gradient; `sums` is a
 * 64-float scratch buffer (32 slots x 2) supplied by the caller. */
int mfnet_upfuse_fwd(const float* x, const void* skip, int skip_dtype, int perm, const float* w, float eps, void* out,
                     int out_dtype, int64_t B, int64_t H, int64_t W, int64_t C, hipStream_t stream);
int mfnet_upfuse_bwd(const float* df, const float* x, const void* skip, int skip_dtype, int perm, const float* w, float eps,
                     float* dx, void* dskip, int dskip_dtype, float* dw, float* sums, int64_t B, int64_t H, int64_t W,
                     int64_t C, hipStream_t stream);
/* final F.interpolate(bilinear, align_corners=False) by integer factor s (:455): x [B,h,w,ldx] fp32 (first ncls
 * channels) -> logits [B,ncls,h*s,w*s] fp32 NCHW; bwd: dlogits -> dx [B,h,w,ldx] bf16 (channels >= ncls zero). */
int mfnet_up_final_fwd(const float* x, int64_t ldx, float* logits, int64_t B, int64_t h, int64_t w, int64_t ncls, int64_t s,
                       hipStream_t stream);
int mfnet_up_final_bwd(const float* dlogits, void* dx, int64_t ldx, int64_t B, int64_t h, int64_t w, int64_t ncls, int64_t s,
                       hipStream_t stream);
/* fpn4 = nn.MaxPool2d(2,2) (:501,513): y [B,H/2,W/2,C] fp32 + arg (0..3) uint8; bwd scatters (accumulate != 0: +=). */
int mfnet_maxpool2_fwd(const float* x, float* y, unsigned char* arg, int64_t B, int64_t H, int64_t W, int64_t C,
                       hipStream_t stream);
int mfnet_maxpool2_bwd(const void* dy, int dy_dtype, const unsigned char* arg, float* dx, int64_t B, int64_t H, int64_t W,
                       int64_t C, int accumulate, hipStream_t stream);

/* SqueezeAndExcitation gate (:251-266; C 256, Cr 16, biases, ReLU) and FRH channel attention (:333-337; C 64,
 * Cr 4, no biases, relu6 != 0): z = act(W1*pooled_sum/P + b1), g = sigmoid(W2*z + b2), one block per image.
 * bwd: dg [B,C] -> dpool [B,C] (= d pooled mean / P, the per-pixel share) and ADDS dW1, db1, dW2, db2.
 * se_scale_sum = SEFusion (:268-281): out = rx*gx[b] + ry*gy[b]; se_bwd_apply: drx = dres*gx[b] + dpx[b], same y. */
int mfnet_se_gate_fwd(const float* pooled_sum, int64_t P, const float* W1, const float* b1, const float* W2, const float* b2,
                      int relu6, float* z, float* g, int64_t B, int64_t C, int64_t Cr, hipStream_t stream);
int mfnet_se_gate_bwd(const float* dg, const float* pooled_sum, int64_t P, const float* W1, const float* W2, int relu6,
                      const float* z, const float* g, float* dpool, float* dW1, float* db1, float* dW2, float* db2, int64_t B,
                      int64_t C, int64_t Cr, hipStream_t stream);
int mfnet_se_scale_sum(const void* rx, int rx_dtype, const void* ry, int ry_dtype, const float* gx, const float* gy, void* out,
                       int out_dtype, int64_t B, int64_t P, int64_t C, hipStream_t stream);
int mfnet_se_bwd_apply(const void* dres, int dres_dtype, const float* gx, const float* gy, const float* dpx, const float* dpy,
                       void* drx, int drx_dtype, void* dry, int dry_dtype, int64_t B, int64_t P, int64_t C, hipStream_t stream);
/* FRH: pc = xf*(sigmoid(dwo) + gate[b]) = pa(x)*x + ca(x)*x (:349-351); bwd: dxf = dpc*(sig+gate) (+ dpool[b]),
 * ddwo = dpc*xf*sig*(1-sig). add3: out = a + b (+ c), gradient joins. */
int mfnet_frh_gate_fwd(const float* xf, const float* dwo, const float* gate, float* pc, int64_t B, int64_t P, int64_t C,
                       hipStream_t stream);
int mfnet_frh_gate_bwd(const float* dpc, const float* xf, const float* dwo, const float* gate, const float* dpool, float* dxf,
                       float* ddwo, int64_t B, int64_t P, int64_t C, hipStream_t stream);
int mfnet_add3(const void* a, int a_dtype, const void* b, int b_dtype, const void* c, int c_dtype, void* out, int out_dtype,
               int64_t n, hipStream_t stream);

/* ===================== sliding-window raster inference, pre / post steps (inference.py) =====================
 * normalize_rgb_u8: out[c] = (in[c]*scale - mean3[c]) / std3[c] on a [3,H,W] uint8 raster (:464-484; mean3 /
 * std3 are HOST arrays). extract_tiles: coords[t] = (row, col, valid_h, valid_w) int32; tiles shorter than the
 * h x w window are filled with numpy 'reflect' padding (:726-737). tile_minmax_norm: per-tile DSM min-max in
 * place; a flat tile (max == min) becomes 0 (:547-555) or, with keep_flat, stays as it is (multiscale_dataset.py:206). tile_prob_accumulate: sum[row+y, col+x] += softmax(logits[t])[cls],
 * cnt += 1 over each tile's valid extent (:562-566); both accumulators are int64 - sum in fixed point (probability * 2^40), so the
 * atomic adds commute and the blend is bit-identical for any window order / rank count. blend_finish: prob = sum/2^40/cnt,
 * mask = prob >= threshold (:582).
 * normalize_raster: BuildingInference._normalize_image (:444-496) on a whole [C,H,W] raster; in_kind 0 uint8 (/255),
 * 1 uint16 (/65535), 2 fp32 (clip((v - sub)/div, 0, 1) with the host-computed 2nd / 98th percentiles when clip != 0);
 * is_rgb == 0 keeps raw values (the DEM, :485-488); ImageNet mean / std (HOST arrays) only when C == 3; pixels where any
 * channel equals nodata_value are zeroed and flagged in nodata_mask (optional, uint8 [H,W]).
 * tile_prob_mask: per-tile softmax(logits)[cls] (optional fp32 out) and uint8 mask = prob >= threshold over the valid
 * extent of each tile, 0 in the cropped padding (:562-566, 582, 742-744): what the CPU vectoriser consumes. */
int mfnet_normalize_rgb_u8(const unsigned char* in, float* out, int64_t H, int64_t W, float scale, const float* mean3,
                           const float* std3, hipStream_t stream);
int mfnet_extract_tiles(const float* rgb, const float* dsm, const int* coords, float* rgb_out, float* dsm_out, int64_t n,
                        int64_t C, int64_t H, int64_t W, int64_t h, int64_t w, hipStream_t stream);
int mfnet_tile_minmax_norm(float* tiles, int64_t n, int64_t per_tile, int keep_flat, hipStream_t stream);

/* Training input pipeline on the device (multiscale_dataset.py:190-215 normalisation, :228-245 flips / rot90):
 * per square S x S tile t, out = rot90^k(vflip?(hflip?(in))) for RGB (uint8 [n,3,S,S] -> fp32 (v/255 - mean)/std),
 * DSM (fp32 [n,S,S], optional) and labels (int64 [n,S,S], optional); codes[t]: bit 0 horizontal flip, bit 1
 * vertical flip, bits 2-3 = k (np.rot90, counter-clockwise). mean3 / std3 are HOST arrays. The reference's per-tile
 * DSM min-max (:205-208) is mfnet_tile_minmax_norm on dsm_out. */
int mfnet_prepare_tiles(const unsigned char* rgb, const float* dsm, const int64_t* labels, const int* codes,
                        float* rgb_out, float* dsm_out, int64_t* labels_out, int64_t n, int64_t S,
                        const float* mean3, const float* std3, hipStream_t stream);
/* Brightness / contrast jitter of the loader's augmentation (multiscale_dataset.py:250-265) in place on uint8 patches
 * [n,3,S,S]: per tile t and channel, v = x * b; v = (v - mean(v)) * c + mean(v); clip to [0, 255]; truncate to uint8, fp32 in the
 * reference's statement order; params [n,2] fp32 on the device = (b, c) per tile, b <= 0 leaves the tile untouched. */
int mfnet_jitter_tiles(unsigned char* rgb, const float* params, int64_t n, int64_t S, hipStream_t stream);
int mfnet_tile_prob_accumulate(int64_t* sum, int64_t* cnt, const float* logits, const int* coords, int64_t n, int64_t C, int64_t H,
                               int64_t W, int64_t h, int64_t w, int64_t cls, hipStream_t stream);
int mfnet_normalize_raster(const void* in, int in_kind, int64_t C, int64_t H, int64_t W, int is_rgb, int clip, float sub,
                           float div, const float* mean3, const float* std3, int has_nodata, double nodata_value, float* out,
                           unsigned char* nodata_mask, hipStream_t stream);
int mfnet_tile_prob_mask(const float* logits, const int* coords, float* prob, unsigned char* mask, int64_t n, int64_t C,
                         int64_t h, int64_t w, int64_t cls, float threshold, hipStream_t stream);
int mfnet_blend_finish(const int64_t* sum, const int64_t* cnt, float* prob, unsigned char* mask, float threshold, int64_t n,
                       hipStream_t stream);
/* cm[gt*n + pred] += 1 for pixels with 0 <= gt < n: the confusion matrix behind utils2.metrics (utils2.py:168-216),
 * pred uint8, gt int64, cm int64 [n,n] (zero first), n <= 32. */
int mfnet_confusion(const unsigned char* pred, const int64_t* gt, int64_t* cm, int64_t n_classes, int64_t total, hipStream_t stream);

/* ---- fp32-accurate forward path (csrc/fp32.hip): the reference's arithmetic is fp32 throughout (unetformer_mmsam.py:533-563,
 * lora_block.py:169-196); these entry points reproduce its logits to 1e-4 relative where the bf16 MFMA path gives 2e-2.
 *
 * gemm_f32: out[m,n] = act(alpha * sum_k A[m,k] W[n,k] + bias[n]) (+ residual[m or m % res_row_mod, n]) (+ out[m,n] if accumulate),
 * all fp32, on v_mfma_f32_16x16x4_f32 (a k-ordered fp32 fma chain) - what F.linear / 1x1 convs / im2col'ed 3x3 convs and the
 * rank-r LoRA products (layers.py:149-165, 244-268) compute. K % 4 == 0, lda / ldw % 4 == 0, 16-byte aligned operands.
 * act: 0 none, 1 GELU (erf), 3 ReLU6, 6 ReLU.
 * attn_f32: SAM ViT attention (lora_block.py:169-196, adapter_fusionblock.py:142-198) on qkv [G*Sh*Sw, 3*heads*hd] fp32:
 * ws > 0 = ws x ws windows over the zero-padded map (a padded token's qkv row is qkv_bias, lora_block.py:199-220), ws == 0 =
 * global; scores = scale * q.k + q.th[qy-ky+kh-1] + q.tw[qx-kx+kw-1] with the resized tables th [2kh-1, hd], tw [2kw-1, hd]
 * (get_rel_pos / add_decomposed_rel_pos, :248-317); out [G*Sh*Sw, heads*hd]. hd in {32, 64, 80}; window / map side <= 64.
 * im2col_patch_f32 / im2col3x3_f32: fp32 twins of mfnet_im2col_patch / mfnet_im2col3x3. gla_attn_fwd_f32: mfnet_gla_attn_fwd
 * on fp32 qkv. */
int mfnet_gemm_f32(const float* A, int64_t lda, const float* W, int64_t ldw, int64_t M, int64_t N, int64_t K, const float* bias,
                   float alpha, int act, const float* residual, int64_t ldr, int64_t res_row_mod, float* out, int64_t ldo,
                   int accumulate, hipStream_t stream);
int mfnet_attn_f32(const float* qkv, const float* qkv_bias, const float* th, const float* tw, float* out, int64_t G, int64_t Sh,
                   int64_t Sw, int64_t heads, int64_t hd, int64_t ws, float scale, hipStream_t stream);
int mfnet_im2col_patch_f32(const float* rgb, const float* dsm, int dsm_channels, float* cols, int64_t B, int64_t H, int64_t W,
                           int64_t P, hipStream_t stream);
int mfnet_im2col3x3_f32(const float* x, float* cols, int64_t G, int64_t H, int64_t W, int64_t C, hipStream_t stream);
int mfnet_gla_attn_fwd_f32(const float* qkv, const float* table, float* out, int64_t B, int64_t H, int64_t W, int64_t heads,
                           int64_t hd, int64_t ws, float scale, hipStream_t stream);

/* ---- fp32 training (the reference trains in fp32: trainer.py:146-159): the backward twins the fp32 path needs beyond the
 * dtype-flagged kernels above. The forward + backward of `UNetFormer(precision="fp32")` in train mode runs the same module
 * backward passes as the bf16 path with every contraction on mfnet_gemm_f32 (dgrad = dy.W, wgrad = dy^T.x on transposed copies).
 *
 * attn_f32_bwd: dqkv of mfnet_attn_f32 (lora_block.py:169-196 differentiated; rel-pos tables and the qkv bias are frozen,
 * unetformer_mmsam.py:522-528): `out` = the forward's output, dout its gradient, dqkv [G*Sh*Sw, 3*heads*hd] fully written.
 * workspace: mfnet_attn_f32_bwd_workspace_floats(G, Sh, Sw, heads, ws) floats (row max+log-sum, delta and the per-query
 * bias rows, handed from the dq kernel to the dk / dv kernel).
 * gla_attn_bwd_f32 / up_final_bwd_f32 / adapter_mix_bwd_f32: mfnet_gla_attn_bwd / mfnet_up_final_bwd / mfnet_adapter_mix_bwd
 * with fp32 qkv / dqkv, fp32 dx and fp32 dada. */
int64_t mfnet_attn_f32_bwd_workspace_floats(int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t ws);
int mfnet_attn_f32_bwd(const float* qkv, const float* qkv_bias, const float* th, const float* tw, const float* out, const float* dout,
                       float* dqkv, int64_t G, int64_t Sh, int64_t Sw, int64_t heads, int64_t hd, int64_t ws, float scale,
                       float* workspace, int64_t workspace_floats, hipStream_t stream);
int mfnet_gla_attn_bwd_f32(const float* qkv, const float* table, const float* dout, float* dqkv, float* dtable, int64_t B, int64_t H,
                           int64_t W, int64_t heads, int64_t hd, int64_t ws, float scale, hipStream_t stream);
int mfnet_up_final_bwd_f32(const float* dlogits, float* dx, int64_t ldx, int64_t B, int64_t h, int64_t w, int64_t ncls, int64_t s,
                           hipStream_t stream);
int mfnet_adapter_mix_bwd_f32(const float* dh, const float* ada, const float* wx, const float* wy, float* dada, float* dw, int64_t M,
                              int64_t D, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MFNET_HIP_H */
