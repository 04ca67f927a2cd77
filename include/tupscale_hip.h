/* C ABI of libtupscale_hip.so -- the MI355X (gfx950) kernels behind the FastTransformer
 * plugin surface (models/FastTransformer/model.py: TransformerModel).
 *
 * The reference has no native layer: its hot path is a chain of PyTorch aten calls inside
 * nn.Module.forward.  Each entry point below replaces the aten call site(s) cited next to it
 * (reference file:line, relative to the reference repo root).  A binding needs only plain
 * pointers to device memory, ints and a hipStream_t (passed as void*); no torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a hipError_t value; launches are asynchronous on
 *     `stream` (pass the caller's current stream, never assume the null stream);
 *   - bf16 tensors are raw uint16 storage; NHWC = [B][H][W][C] with C = 64;
 *   - "window layout" = token rows ordered [B][window_y][window_x][8*8 tokens], 192 features,
 *     the order window_partition (model.py:31-45) produces, including zero rows for the
 *     tokens added by the bottom/right padding (model.py:273-280);
 *   - packed weights are produced by transformerupscaler_amd/packing.py (layouts documented
 *     at each function).
 */
#ifndef TUPSCALE_HIP_H
#define TUPSCALE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* library / ABI version, bumped when a signature changes */
int tup_abi_version(void);

/* conv1: Conv2d(3,64,k3,p1)+ReLU. model.py:202-203,251.
 * x fp32 [B][3][H][W]; wp bf16 [64][32] (row ct*16+4g+e = channel g*16+ct*4+e, k = tap*3+cin,
 * zero pad 27..31); bias fp32 [64] or NULL; out bf16 NHWC.
 * Also the input-gradient conv of the 64->3 convs (transposed/flipped weight as wp): in_mask fp32
 * (shape of x; input *= in_mask > 0) and out_mask bf16 NHWC (out *= out_mask > 0) fuse the ReLU backward. */
int tup_conv3x3_c3_fwd(const float* x, const void* wp, const float* bias, const float* in_mask,
                       const void* out_mask, void* out, int B, int H, int W, int relu, void* stream);

/* Conv2d(64*in_r^2, Cout, k3, p1) on NHWC bf16:
 *   conv2 model.py:204,252 | decoder_conv1 model.py:228,312 | Upsampler convs + PixelShuffle(r)
 *   utils.py:62-63,74-75,83-84 (out_mode 0, ntiles = r*r, the cout tile index is the sub-pixel
 *   i*r+j) | up1_conv utils.py:32-40 and decoder_conv2 model.py:229,313 (out_mode 1).
 * x bf16 [B][H*in_r][W*in_r][64]: in_r = 1 is a plain map; in_r > 1 reads 64*in_r^2 logical channels
 * through PixelShuffle^-1 (the input-gradient conv of an Upsampler stage).
 * wp bf16 [ntiles][in_r^2][9][rows][64]; out_mode 0: rows = 64, bias fp32 [ntiles][64] or NULL,
 * out bf16 [B][H*r][W*r][64], optional add / mask (bf16, shape of out):
 * out = (conv + bias [relu] + add) * (mask > 0); out_mode 1: rows = 16 (cout_valid real), bias fp32
 * [cout_valid] or NULL, out fp32 [B][cout_valid][H][W]. */
int tup_conv3x3_c64_fwd(const void* x, const void* wp, const float* bias, const void* add,
                        const void* mask, void* out, int B, int H, int W, int ntiles, int r,
                        int cout_valid, int relu, int out_mode, int in_r, void* stream);

/* Inference-only composition of branch A: the LAST Upsampler conv (64->64rr, +bias) + PixelShuffle(r)
 * (utils.py:62-63,74-75,83-84) and up1_conv (64->3, no bias, ReLU; utils.py:32-40), called back to back at
 * model.py:264-265 with no non-linearity between them, evaluated as one 5x5-tap conv with 3rr outputs
 * (exact algebra; the outermost HR pixel ring uses per-border weight variants because the reference
 * zero-pads the HR intermediate).  x bf16 NHWC [B][H][W][64]; wp bf16 [25][rows][64] (rows 16/32/112 for
 * r 2/3/6); bias fp32 [3rr]; wv bf16 [9][3rr][25][64]; bv fp32 [9][3rr]; out fp32 [B][3][H*r][W*r]. */
int tup_conv5x5_c64_planar_fwd(const void* x, const void* wp, const float* bias, const void* wv,
                               const float* bv, float* out, int B, int H, int W, int r, int relu, void* stream);

/* Planar fp32 Conv2d(3, 3*r*r, k3, p1) + PixelShuffle(r) [+ add] [+ clamp(0,1)]:
 *   final_upscale utils.py:62-63,74-75,83-84 (n_feats=3) | final_upscale_conv model.py:212,317
 *   fused with "out = upscaled_input + residual_up" model.py:320 and torch.clamp model.py:327.
 * x fp32 [B][3][H][W]; w28 fp32 [3*r*r][28] (27 taps (cin,ky,kx) + pad); out/add fp32 [B][3][H*r][W*r].
 * clamp01 = 2 (training): out is [2][B][3][H*r][W*r]: the unclamped sum (the clamp's backward gate), then the clamped output. */
int tup_conv3x3_planar_fwd(const float* x, const float* w28, const float* bias, const float* add,
                           float* out, int B, int H, int W, int r, int clamp01, void* stream);

/* transforms.Resize on a tensor = antialiased bilinear (model.py:323-325, train.py:127-130)
 * [+ clamp(0,1) model.py:327].  Tap tables as aten's _compute_indices_weights_aa (float32).
 * clamp01: 0 = out is the resized tensor; 1 = clamped; 2 (training) = out is [2][planes][Ho][Wo]: first the unclamped values
 * (the gate of the clamp's backward), then the clamped ones (the model output), written in the same pass. */
int tup_resize_aa_fwd(const float* in, float* out, const int* ymin, const int* ysize, const float* yw,
                      int KY, const int* xmin, const int* xsize, const float* xw, int KX, int planes,
                      int Hi, int Wi, int Ho, int Wo, int clamp01, void* stream);

/* The same output tail for a last stage of r = 2 (scales 2 and 4) WITHOUT the Resize, as a register-streaming stencil (no LDS):
 * final_upscale's last Conv2d(3,12,3)+PixelShuffle(2) utils.py:62-63,74-75, final_upscale_conv + "+ upscaled_input"
 * model.py:316-320 [+ clamp model.py:327].  x fp32 [B][3][H][W]; wfu_t fp32 [27][12], wfc_t fp32 [27][4] tap-major
 * (k = cin*9 + ky*3 + kx; packing.pack_planar_t); bfu [12], bfc [3]; ui / out fp32 [B][3][2H][2W]. */
int tup_tail_stream_r2_fwd(const float* x, const float* wfu_t, const float* bfu, const float* wfc_t, const float* bfc,
                           const float* ui, float* out, int B, int H, int W, int clamp01, void* stream);

/* ... followed by the antialiased Resize to Ho x Wo (model.py:323-325; tap tables as tup_resize_aa_fwd, at most 4 taps per output)
 * and the clamp, in the same kernel: out fp32 [B][3][Ho][Wo].  sc = LR columns a wave strip advances by
 * (<= 60 - ceil((max taps - 1) / 2)), band_h = LR rows per band (multiple of 3), ext = LR rows a band runs past its end
 * (ceil((max taps - 1) / 2)); oxb int [ceil(W / sc) + 1] / oyb int [ceil(H / band_h) + 1]: first output column / row whose
 * first tap lies in strip s's HR columns [2 s sc, ...) / band k's HR rows [2 k band_h, ...). */
int tup_tail_stream_r2_resize_fwd(const float* x, const float* wfu_t, const float* bfu, const float* wfc_t, const float* bfc,
                                  const float* ui, float* out, const int* ymin, const int* ysize, const float* yw, int KY,
                                  const int* xmin, const int* xsize, const float* xw, int KX, const int* oxb, const int* oyb,
                                  int B, int H, int W, int Ho, int Wo, int sc, int band_h, int ext, int clamp01, void* stream);

/* Inference fusion of the output tail (model.py:316-327): last final_upscale stage (Conv2d(3,3rr,3)+PixelShuffle),
 * final_upscale_conv, "+ upscaled_input", Resize (tap tables; identity tables when sizes match) and clamp.
 * x fp32 [B][3][H][W]; ui fp32 [B][3][H*r][W*r]; out fp32 [B][3][Ho][Wo]; EH/EW = largest HR window a 16x64
 * output tile's taps touch. */
int tup_tail_fused_fwd(const float* x, const float* wfu, const float* bfu, const float* wfc, const float* bfc,
                       const float* ui, float* out, const int* ymin, const int* ysize, const float* yw, int KY,
                       const int* xmin, const int* xsize, const float* xw, int KX, int B, int H, int W, int r,
                       int Ho, int Wo, int EH, int EW, int clamp01, void* stream);

/* torch.clamp(out, 0, 1) model.py:327 when no resize precedes it. */
int tup_clamp01_fwd(const float* in, float* out, long long n, void* stream);

/* nn.LayerNorm(192) model.py:142,144,163,169.  x fp32 [M][192] -> y bf16 [M][192];
 * mean/rstd fp32 [M] optional (saved for backward). */
int tup_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y,
                      float* mean, float* rstd, int M, void* stream);

/* relative_position_bias_table[relative_position_index] model.py:120-123 -> dense per-head bias
 * in the attention kernel's fragment order, fp32 [12][4][4][64][4]. */
int tup_relpos_bias_expand(const float* table, float* frag, void* stream);

/* WindowAttention core model.py:114-130 (q*scale, qk^T + bias, softmax, @v, head concat).
 * qkv bf16 [nwin][64][576]; out bf16 [nwin][64][192].  drop_p > 0: attn_drop (model.py:80,127) with a
 * stateless hash mask keyed by drop_seed (csrc/common.h); the backward re-derives the same mask.
 * lse: NULL (inference) or fp32 [nwin][12][64] <- log-sum-exp of every score row, saved for tup_window_attn_bwd. */
int tup_window_attn_fwd(const void* qkv, const float* bias_frag, void* out, float* lse, int nwin, float drop_p,
                        unsigned int drop_seed, void* stream);

/* nn.Linear family model.py:79,81,146-151 with fused epilogues (forward AND the input-gradient
 * GEMMs of the backward, which are the same product with the transposed weight packed as Wt).
 * Wt bf16 [N][K] (rows permuted per 64-group: row ct*16+4g+e = feature g*16+ct*4+e), bias fp32 [N].
 * a_dtype 0: A bf16 [M][lda]; 1: A fp32.  epilogue 0: (+bias) -> bf16 | 1: +bias, erf-GELU -> bf16
 * (model.py:148) | 2: +bias + res -> fp32 (residual adds model.py:164,171) | 3: * gelu'(aux) -> bf16
 * (aux bf16 [M][ldo] = saved pre-activation; backward of model.py:148).  With epilogue 1 a non-NULL
 * aux is an OUTPUT that receives the bf16 pre-activation (saved for the backward).  drop_p > 0 (epilogue 2):
 * out = dropout(acc + bias) + res -- proj_drop / the MLP's Dropout (model.py:82,132,150). */
int tup_gemm_tokens_fwd(const void* A, int a_dtype, int lda, const void* Wt, const float* bias,
                        const float* res, const void* aux, void* out, int ldo, int M, int N, int K,
                        int epilogue, float drop_p, unsigned int drop_seed, void* stream);

/* norm1 fused into attn.qkv (model.py:163 + :115): out bf16 [M][N] = LayerNorm(x) Wt^T + bias; x fp32 [M][192]. */
int tup_ln_gemm_fwd(const float* x, const float* gamma, const float* beta, const void* Wt,
                    const float* bias, void* out, int M, int N, void* stream);

/* Inference fusion of the MLP half of a block: x += mlp.2(GELU(mlp.0(norm2(x)))) (model.py:144-151,168-171);
 * the [M][768] hidden tensor stays in registers.  x fp32 [M][192] in place; w1 bf16 [768][192] = mlp.0.weight / 4 (rows
 * permuted per 64-group AND columns in the kernel's K order) and b1 = mlp.0.bias / 4 (packing.pack_fc1_fused_q: exact powers of
 * two), w2 FP16 [192][768] = 4 mlp.2.weight (rows permuted per 64-group, packing.pack_fc2_h4): erf-GELU (nn.GELU(), model.py:148)
 * is evaluated in packed fp16 on x / 4 and FC2 runs on the fp16 MFMA, the factors of 4 cancel. */
int tup_fused_mlp_fwd(float* x, const float* gamma, const float* beta, const void* w1, const float* b1,
                      const void* w2, const float* b2, int M, void* stream);

/* reflect pad + patch_embed Conv2d(64,192,k8,s8) + NHWC permute + zero token pad +
 * window_partition: model.py:256-261,268-285.  feat bf16 NHWC; Wt bf16 [192][4096],
 * k = (i*8+j)*64+c; x_out fp32 window layout. */
int tup_patch_embed_fwd(const void* feat, const void* Wt, const float* bias, float* x_out,
                        int B, int H, int W, void* stream);

/* window_reverse + crop + patch_unembed ConvTranspose2d(192,64,k8,s8) + crop + skip add:
 * model.py:292-309.  x window layout, fp32 (x_bf16 = 0) or bf16 (x_bf16 = 1: tup_blocks_stream_fwd's out_bf16 -- the GEMM rounds an
 * fp32 x to the same bf16 values on load); Wt bf16 [4096][192], n = (i*8+j)*64+o; bias fp32 [64];
 * skip/out bf16 NHWC [B][H][W][64]. */
int tup_patch_unembed_fwd(const void* x, int x_bf16, const void* Wt, const float* bias, const void* skip,
                          void* out, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ResidualTransformer path (reference models/ResidualTransformer/model.py; forward / inference).
 * Reuses conv1/conv2/decoder kernels above; the stride-2 `downsample` conv (model.py:89,132) runs on
 * tup_conv3x3_c64_fwd with in_r = 2 (space-to-depth read) and a zero-padded 3x3x4 weight pack.
 * ------------------------------------------------------------------------------------------- */

/* patch_embed Conv2d(64,128,k8,s8) + flatten/transpose + pos_embed (model.py:135-140): feat bf16 NHWC
 * [B][H][W][64]; Wt bf16 [128][4096]; pos fp32 [T][128]; x_out fp32 [B*T][128], T = (H/8)*(W/8). */
int tup_rt_patch_embed_fwd(const void* feat, const void* Wt, const float* bias, const float* pos, float* x_out,
                           int B, int H, int W, void* stream);

/* transpose/view + patch_unembed ConvTranspose2d(128,64,k8,s8) + skip add (model.py:147-153). */
int tup_rt_patch_unembed_fwd(const float* x, const void* Wt, const float* bias, const void* skip, void* out,
                             int B, int H, int W, void* stream);

/* nn.MultiheadAttention(128, 8 heads) core, eval mode (model.py:31,43): qkv bf16 [B][N][384] -> out bf16 [B][N][128];
 * flash-style (online softmax), any N; lse (optional) receives the per-query log-sum-exp for the backward.
 * drop_p > 0: nn.MultiheadAttention's dropout on the attention probabilities, stateless hash mask on (image, head, q, k). */
int tup_rt_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, float drop_p, unsigned int drop_seed,
                         void* stream);

/* Its backward (P recomputed from the saved log-sum-exp `lse` fp32 [B][8][N]; two passes, no atomics):
 * out/gout bf16 [B][N][128], work fp32 [B][8][N] scratch, gqkv bf16 [B][N][384]. */
int tup_rt_attention_bwd(const void* qkv, const void* out, const void* gout, const float* lse, float* work,
                         void* gqkv, int B, int N, float drop_p, unsigned int drop_seed, void* stream);

/* nn.LayerNorm(128) (model.py:30,32): x fp32 [M][128] -> y bf16. */
int tup_layernorm128_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                         int M, void* stream);

/* out = clamp(F.interpolate(a, bicubic) + F.interpolate(b, bicubic)) (model.py:125,160-164); tables [Ho][4]/[Wo][4]
 * of clamped source indices and weights per source (align_corners=False, A=-0.75). */
int tup_rt_bicubic_sum_fwd(const float* a, const float* b, float* out, const int* ayi, const float* ayw,
                           const int* axi, const float* axw, const int* byi, const float* byw, const int* bxi,
                           const float* bxw, int planes, int Ha, int Wa, int Hb, int Wb, int Ho, int Wo,
                           int clamp01, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Backward (what torch autograd executes for the same call sites under train.py:138).
 * Accumulating outputs (documented per function) must be zeroed by the caller.
 * ------------------------------------------------------------------------------------------- */

/* Weight gradient of a Linear: out[NI][NJ] (fp32, ldo) += P^T Q, P [M][NI], Q [M][NJ]; dtypes 0 = bf16,
 * 1 = fp32.  For nn.Linear: P = grad_output, Q = input -> out = weight.grad layout [out][in]. */
int tup_gemm_wgrad(const void* P, int p_dtype, int ldp, const void* Q, int q_dtype, int ldq,
                   float* out, int ldo, int M, int NI, int NJ, void* stream);
/* The same with the layer's bias gradient in the same pass (model.py:79,81,146-151 under train.py:138):
 * colsum_out[NI] (fp32, zeroed by the caller) += column sums of P. */
int tup_gemm_wgrad_bias(const void* P, int p_dtype, int ldp, const void* Q, int q_dtype, int ldq,
                        float* out, int ldo, float* colsum_out, int M, int NI, int NJ, void* stream);

/* Weight gradient of patch_embed (reflect=1: P = grad tokens, map = feat) and of patch_unembed
 * (reflect=0: P = tokens, map = grad of its output).  out fp32 [192][4096] +=, column (i*8+j)*64+c. */
int tup_patch_wgrad(const float* P, const void* map, float* out, int B, int H, int W, int reflect, void* stream);
/* The same from bf16 token rows P [M][192] (the rounding tup_patch_wgrad applies on load, done by the caller) on the wide-tile
 * kernel: one workgroup per CU owns 192 x 128 of the output, both operands by LDS-DMA, the map is read once instead of three
 * times and the token rows 32 instead of 64 times.  B * H * W * 128 < 2^31. */
int tup_patch_wgrad_bf16(const void* P, const void* map, float* out, int B, int H, int W, int reflect, void* stream);

/* Bias gradients: out[N] += column sums of G [M][N] (dtype 0 bf16 / 1 fp32) over the rows with
 * rowmask[m] != 0 (uint8 [M]; NULL = all rows). */
int tup_colsum(const void* G, int dtype, int ld, float* out, int M, int N, const void* rowmask, void* stream);

/* LayerNorm backward: dx = LN'(gy) (+ gres), dgamma/dbeta fp32 [192] +=.  gdrop: NULL, or bf16 [M][192] <- dx * dropout mask /
 * (1 - drop_p) for the site keyed by drop_seed (tup_dropout_bwd of dx fused in: the gradient's next stop is the Dropout behind
 * attn.proj / mlp.2, model.py:82,132,150). */
int tup_layernorm_bwd(const void* gy, const float* x, const float* mean, const float* rstd,
                      const float* gamma, const float* gres, float* dx, float* dgamma, float* dbeta,
                      int M, void* gdrop, float drop_p, unsigned int drop_seed, void* stream);

/* Dense relative-position bias in the second (query-row) fragment order used by the backward. */
int tup_relpos_bias_expand_n(const float* table, float* frag, void* stream);

/* Attention-core backward: P is rebuilt from q, k, the dense bias (bias_n, tup_relpos_bias_expand_n) and the forward's lse;
 * rowsum(P dP) = gout . att with att bf16 [nwin][64][192] the forward's output.  gqkv bf16 [nwin][64][576]; dbias_n fp32
 * [12][4][4][64][4] overwritten (the dense gradient of the relative-position bias, in bias_n's layout).  scratch: fp32
 * [tup_window_attn_bwd_scratch(nwin, heads)], uninitialised -- each persistent wave writes its own partial sum there and a
 * second kernel adds the slots up (no float atomics). */
int tup_window_attn_bwd(const void* qkv, const void* gout, const void* att, const float* lse, const float* bias_n,
                        void* gqkv, float* dbias_n, float* scratch, int nwin, float drop_p, unsigned int drop_seed, void* stream);
long long tup_window_attn_bwd_scratch(int nwin, int heads);

/* Backward of nn.Dropout after proj / mlp.2: gout bf16 = gin fp32 * mask / (1 - p), element index m*192+n. */
int tup_dropout_bwd(const float* gin, void* gout, long long n, float drop_p, unsigned int drop_seed, void* stream);

/* dense bias gradient -> relative_position_bias_table.grad fp32 [225][12] (overwritten). */
int tup_relpos_bias_reduce(const float* dbias_n, float* dtable, void* stream);

/* Input gradients of patch_unembed (gx fp32 window layout, overwritten) and patch_embed (gmap_pad bf16
 * NHWC [B][ceil8(H)][ceil8(W)][64], the reflect-PADDED map, overwritten). */
int tup_patch_unembed_bwd(const void* gmap, const void* Wt, float* gx, int B, int H, int W, void* stream);
int tup_patch_embed_bwd(const float* gx, const void* Wt, void* gmap_pad, int B, int H, int W, void* stream);

/* ... with the gradient merge at `feat` in the epilogue (H, W multiples of 8): out bf16 NHWC [B][H][W][64] = (gx Wt^T + add1 + add2)
 * * (relu_src > 0) -- tup_patch_embed_bwd followed by tup_feat_grad_combine without the padded map in between.  add2 may be NULL.
 * add1_colsum: NULL, or fp32 [16][64] (zeroed by the caller) += per-channel sums of add1 in 16 replicas (tup_colsum of add1 for free). */
int tup_patch_embed_bwd_merge(const float* gx, const void* Wt, const void* add1, const void* add2, const void* relu_src,
                              void* out, float* add1_colsum, int B, int H, int W, void* stream);

/* Weight/bias gradients of the 3x3 convs (accumulating).
 *   c64:    x bf16 NHWC, gmap bf16 NHWC [B][H*gr][W*gr][64] sub-pixel plane sp -> dwp fp32 [64][9][64]
 *           (co, tap, ci) for couts {c*gr*gr + sp}, dbias fp32 [64] or NULL
 *   thin:   gpl fp32 [B][3][H][W] -> dwp fp32 [3][9][64], dbias [3] or NULL   (up1_conv, decoder_conv2)
 *   c3:     x fp32 [B][3][H][W], gmap bf16 NHWC -> dw fp32 [64][3][3][3], dbias [64]   (conv1)
 *   planar: x fp32 [B][3][H][W], gpl fp32 [B][3][H*r][W*r] -> dw fp32 [3rr][3][3][3], dbias [3rr] */
int tup_conv3x3_c64_wgrad(const void* x, const void* gmap, float* dwp, float* dbias,
                          int B, int H, int W, int gr, int sp, void* stream);
int tup_conv3x3_thin_wgrad(const void* x, const float* gpl, float* dwp, float* dbias, int B, int H, int W, void* stream);
int tup_conv3x3_c3_wgrad(const float* x, const void* gmap, float* dw, float* dbias, int B, int H, int W, void* stream);
int tup_conv3x3_planar_wgrad(const float* x, const float* gpl, float* dw, float* dbias,
                             int B, int H, int W, int r, void* stream);

/* Input gradient of Conv2d(3,3rr,k3)+PixelShuffle(r) on planar fp32 (final_upscale): w fp32 [3rr][3][3][3]. */
int tup_conv3x3_planar_dgrad(const float* gpl, const float* w, float* gx, int B, int H, int W, int r, void* stream);

/* Backward of Resize(antialias) [+ clamp]: gin fp32 [planes][Hi][Wi]; pre = pre-clamp output or NULL;
 * oy0/oyn/ox0/oxn: per input row/col the range of outputs that reference it.
 * l1_scale: NULL, or device float[2] = {d loss / numel, plain} -- then `gout` is the TARGET of nn.L1Loss (train.py:103,132) and the
 * gradient is formed in the kernel (pre required): plain = 0: sign(clamp(pre) - target) * l1_scale[0], gated by the clamp (pre = the
 * pre-clamp output); plain != 0: sign(pre - target) * l1_scale[0] with pre = the Resize output itself (train.py:127-130 resizes the
 * clamped model output, nothing clamps after it).  The loss's backward pass and its 100 MB gradient tensor do not exist. */
int tup_resize_aa_bwd(const float* gout, const float* pre, float* gin, const int* ymin, const float* yw, int KY,
                      const int* xmin, const float* xw, int KX, const int* oy0, const int* oyn,
                      const int* ox0, const int* oxn, int planes, int Hi, int Wi, int Ho, int Wo,
                      const float* l1_scale, void* stream);

/* gin = gout * [0 <= pre <= 1] * [relu_src > 0] (clamp / ReLU backward; either source may be NULL); l1_scale as above. */
int tup_mask_bwd(const float* gout, const float* pre, const float* relu_src, float* gin, long long n,
                 const float* l1_scale, void* stream);

/* Gradient merge at `feat` (model.py:264,268,308 fan-out) + conv2's ReLU backward:
 * out = (a + b + fold_reflect(gpe)) * (feat > 0), NHWC bf16; gpe = padded map from tup_patch_embed_bwd. */
int tup_feat_grad_combine(const void* a, const void* b, const void* gpe, const void* feat, void* out,
                          int B, int H, int W, void* stream);

/* ---- ResidualTransformer backward (autograd through models/ResidualTransformer/model.py:121-165 in train.py:138) ---- */

/* LayerNorm(128) backward; same contract as tup_layernorm_bwd (dgamma / dbeta fp32 [128] accumulated). */
int tup_layernorm128_bwd(const void* gy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         const float* gres, float* dx, float* dgamma, float* dbeta, int M, void* gdrop, float drop_p,
                         unsigned int drop_seed, void* stream);

/* patch_embed / patch_unembed weight gradients on the plain token grid: out fp32 [128][4096] += P^T patches(map);
 * P fp32 [B*T][128], map NHWC bf16 [B][H][W][64], column (i*8+j)*64 + c. */
int tup_rt_patch_wgrad(const float* P, const void* map, float* out, int B, int H, int W, void* stream);

/* Weight gradient of the stride-2 `downsample` conv (model.py:132) as a conv over the space-to-depth input:
 * x NHWC bf16 [B][H*xr][W*xr][64] plane xsp, gmap NHWC bf16 [B][H][W][64]; dwp fp32 [64][9][64] +=, dbias [64] += or NULL. */
int tup_conv3x3_c64_wgrad_s2d(const void* x, const void* gmap, float* dwp, float* dbias, int B, int H, int W,
                              int xr, int xsp, void* stream);

/* Backward of tup_rt_bicubic_sum_fwd w.r.t. source a (F.interpolate bicubic + clamp, model.py:160-164): gout / out fp32
 * [planes][Ho][Wo] (out = saved forward output, gate 0 < out < 1; NULL = no clamp), ga fp32 [planes][Ha][Wa], tmp fp32
 * [planes][Ha][Wo]; (ystart [Ha+1], yo, yw) / (xstart [Wa+1], xo, xw) = transposed tap lists. */
int tup_rt_bicubic_bwd(const float* gout, const float* out, float* ga, float* tmp, const int* ystart, const int* yo,
                       const float* yw, const int* xstart, const int* xo, const float* xw, int planes, int Ha, int Wa,
                       int Ho, int Wo, void* stream);
/* The same with the row pass in bands of 16 source rows (every output row is read once per band instead of once per source row it
 * touches) and dense column tables: band_r0 / band_n int [ceil(Ha/16)] = first output row / row count of band b, band_w fp32
 * [nbands][nr_max][16] = weight of output row band_r0[b] + i for source row 16 b + k (zero where none); xoT int / xwT fp32
 * [kmax][Wa] = entry k of source column x's transposed tap list (padding: weight 0); blk_c0 / blk_n int [ceil(Wa/256)] = the
 * stretch (<= 4096 columns) of a tmp row that source columns 256 j .. 256 j + 255 read.  l1_scale (device scalar, or NULL): when
 * given, `gout` is the TARGET of nn.L1Loss (train.py:103,132) applied to the forward output and the upstream gradient
 * sign(out - target) * l1_scale[0] is formed inside the row pass instead of being read. */
int tup_rt_bicubic_bwd_banded(const float* gout, const float* out, float* ga, float* tmp, const int* band_r0, const int* band_n,
                              const float* band_w, int nr_max, const int* xoT, const float* xwT, int kmax, const int* blk_c0,
                              const int* blk_n, int planes, int Ha, int Wa, int Ho, int Wo, const float* l1_scale, void* stream);

/* ---- frame pre/post-processing either side of the model ---- */

/* torchvision ToTensor on a uint8 frame (reference data_handling/data_class.py:61-71, inference.py:65-75,
 * app_overlay.py preproc): src u8 [B][H][W][3] -> dst fp32 [B][3][H][W] = src / 255.  swap_rb = 1: BGR source. */
int tup_u8hwc_to_f32chw(const void* src, float* dst, int B, int H, int W, int swap_rb, void* stream);

/* (x * 255).clamp(0, 255).to(uint8).permute(1, 2, 0)[..., [2, 1, 0]] (reference app_overlay.py:381-388; ToPILImage
 * in inference.py:123-124): src fp32 [B][3][H][W] -> dst u8 [B][H][W][3], truncating.  swap_rb = 1: BGR output. */
int tup_f32chw_to_u8hwc(const float* src, void* dst, int B, int H, int W, int swap_rb, void* stream);

/* ---- WindowTransformer plugin (models/WindowTransformer/model.py): the FastTransformer window
 * block at embedding width 128 / 8 heads ---- */

/* relative_position_bias_table [225][heads] -> dense S^T-fragment bias [heads*16*256] (heads = 8 or 12). */
int tup_relpos_bias_expand_h(const float* table, float* frag, int heads, void* stream);

/* WindowAttention core (model.py:125-143) for heads x 16 channels: qkv bf16 [nwin][64][48*heads] -> out bf16
 * [nwin][64][16*heads]; lse NULL or fp32 [nwin][heads][64] as tup_window_attn_fwd. */
int tup_window_attn_fwd_h(const void* qkv, const float* bias_frag, void* out, float* lse, int nwin, int heads, float drop_p,
                          unsigned int drop_seed, void* stream);

/* patch_embed (stride-8 conv, no padding: floor(H/8) x floor(W/8) tokens) + permute + zero token pad + window_partition
 * (model.py:247-268): feat bf16 NHWC [B][H][W][64], Wt bf16 [N][4096], x_out fp32 window layout [B*nWy*nWx*64][N]. */
int tup_wt_patch_embed_fwd(const void* feat, const void* Wt, const float* bias, float* x_out, int B, int H, int W, int N,
                           void* stream);

/* window_reverse + crop + patch_unembed + cropped skip add (model.py:272-291): x fp32 window layout [M][K], Wt bf16
 * [4096][K], skip / out bf16 NHWC [B][Hs][Ws][64] (Hs, Ws multiples of 8 = the token grid times 8). */
int tup_wt_patch_unembed_fwd(const float* x, const void* Wt, const float* bias, const void* skip, void* out,
                             int B, int Hs, int Ws, int K, void* stream);

/* nn.L1Loss() of the training step (reference train.py:103,132) and its backward.  a, b fp32 [n] (n % 4 == 0):
 * partial[nblocks] receives per-workgroup sums of |a - b| (the caller adds them and divides by n);
 * ga = sign(a - b) * gout[0] / n. */
int tup_l1_loss_partial(const float* a, const float* b, float* partial, long long n, int nblocks, void* stream);
int tup_l1_loss_bwd(const float* a, const float* b, const float* gout, float* ga, long long n, void* stream);

/* Measurement aid (bench.py `sustained.clock_GHz`; no reference counterpart): writes s_memtime (shader cycles) and s_memrealtime
 * (100 MHz ticks) of one wave to out2[0], out2[1] (two uint64 on the device), in stream order. */
int tup_clock_probe(void* out2, void* stream);

/* WindowTransformer backward pieces: the FastTransformer entries for `heads` = 8 or 12 and the window-layout patch weight
 * gradient on the floor(H/8) x floor(W/8) token grid (out fp32 [NI][4096] += P^T patches(map), no reflect padding). */
int tup_relpos_bias_expand_n_h(const float* table, float* frag, int heads, void* stream);
int tup_window_attn_bwd_h(const void* qkv, const void* gout, const void* att, const float* lse, const float* bias_n, void* gqkv,
                          float* dbias_n, float* scratch, int nwin, int heads, float drop_p, unsigned int drop_seed, void* stream);
int tup_relpos_bias_reduce_h(const float* dbias_n, float* dtable, int heads, void* stream);
int tup_wt_patch_wgrad(const float* P, const void* map, float* out, int B, int H, int W, int NI, void* stream);

/* Inference fusion of norm1 + attn.qkv + the WindowAttention core (models/FastTransformer/model.py:104-130,163): the qkv
 * tensor never exists.  x fp32 [64*nwin][192] (window layout); wh bf16 [12][64][192] = per head the q, k, v weight rows
 * (16 each, natural channel order) + 16 zero rows; bh fp32 [12][48]; bias_frag from tup_relpos_bias_expand;
 * out bf16 [64*nwin][192] (the input of attn.proj). */
int tup_fused_qkv_attn_fwd(const float* x, const float* gamma, const float* beta, const void* wh, const float* bh,
                           const float* bias_frag, void* out, int nwin, void* stream);

/* ... and with attn.proj + the residual add as well (the attention half of a block, model.py:163-164), x updated in place.
 * wproj bf16 [192][192] from packing.pack_proj_pairs (rows permuted per 64-group, columns in head-pair K order). */
int tup_fused_attn_block_fwd(float* x, const float* gamma, const float* beta, const void* wh, const float* bh,
                             const float* bias_frag, const void* wproj, const float* bproj, int nwin, void* stream);

/* One whole WindowTransformerBlock in place (model.py:153-172): x += proj(attention(qkv(norm1(x)))) followed by
 * x += mlp.2(GELU(mlp.0(norm2(x)))) in ONE kernel: the residual stream of a 128-token tile stays in registers between
 * the two halves.  norm1 / norm2's scale and shift arrive folded into attn.qkv / mlp.0 (packing.fold_layernorm: W diag(gamma),
 * b + W beta): wh / bh = pack_qkv_heads of the folded qkv, w1 / b1 = pack_fc1_fused_q of the folded mlp.0 (x 1/4), w2 = 4 mlp.2.weight
 * in fp16 (pack_fc2_h4); bias_frag, wproj, bproj as in tup_fused_attn_block_fwd. */
int tup_fused_block_fwd(float* x, const void* wh, const float* bh, const float* bias_frag, const void* wproj, const float* bproj,
                        const void* w1, const float* b1, const void* w2, const float* b2, int nwin, void* stream);

/* Branch A in TRAINING through its exact composition (r = 2): replaces autograd through `self.up1(feat)` + `self.up1_conv(...)`,
 * model.py:264-265 (Upsampler utils.py:62-63 + BasicConv utils.py:32-40; no non-linearity between the two convs, utils.py:50).
 * compose: wu fp32 [256][64][3][3], bu fp32 [256], w3 fp32 [3][64][3][3] -> the composed 5x5 conv's weights in every layout the
 * kernels use: wv bf16 [9 border variants][12 n][25 taps][64] + bv fp32 [9][12]; wp bf16 [25][16][64] (forward main kernel,
 * zero-initialised by the caller); wd bf16 [13][64][32] (input-gradient kernel, zero-initialised by the caller). */
int tup_bra_compose(const float* wu, const float* bu, const float* w3, void* wv, float* bv, void* wp, void* wd, void* stream);
/* backward: g fp32 [B][3][2H][2W] (gradient w.r.t. upscaled_input, before its ReLU mask), ui fp32 [B][3][2H][2W], feat bf16
 * [B][H][W][64] -> dfeat bf16 [B][H][W][64] (written); G fp32 [16][9][12][25][64], Gb fp32 [16][9][12] = gradient w.r.t. the
 * composed weights / biases per variant, spread over 16 replicas against atomic contention (ACCUMULATED: zero first; the
 * gradient is the sum over the replicas); g12 bf16 [B][H][W][16] workspace.  H, W >= 6. */
int tup_bra_backward(const float* g, const float* ui, const void* feat, const void* wd, const void* wv,
                     void* g12, void* dfeat, float* G, float* Gb, int B, int H, int W, void* stream);
/* chain rule through the composition: G, Gb (16 replicas each, summed here) -> dwu fp32 [256][64][3][3], dbu fp32 [256], dw3 fp32 [3][64][3][3] (written);
 * dM fp32 [62208] and dMb fp32 [108] workspaces. */
int tup_bra_chain(const float* G, const float* Gb, const float* wu, const float* bu, const float* w3,
                  float* dM, float* dMb, float* dwu, float* dbu, float* dw3, void* stream);

/* transforms.Resize on a uint8 PIL image (reference data_handling/data_class.py:61-71, inference.py:65-75) = Pillow's two-pass
 * 8-bit BILINEAR resampler (third-party Pillow libImaging/Resample.c: triangle filter widened by the down-scale factor,
 * weights normalised in double and rounded to 22 fractional bits, out = clip8((2^21 + sum(pixel * k)) >> 22), horizontal pass
 * into a uint8 image first).  Bit-exact with Pillow.  Tables (device, int32) from resize_taps.pil_bilinear_coeffs:
 * min / size [out], k [out][ksize].
 * rows: src u8 [B][H][W][3] -> dst u8 [B][H][Wo][3]. */
int tup_resize_u8_rows(const void* src, void* dst, const int* xmin, const int* xsize, const int* k, int ksize,
                       int B, int H, int W, int Wo, void* stream);
/* cols: src u8 [B][H][W][3] -> dst_u8 u8 [B][Ho][W][3] (or NULL) and / or dst_f32 fp32 [B][3][Ho][W] = value / 255 (or NULL;
 * ToTensor fused, swap_rb = 1 for BGR frames). */
int tup_resize_u8_cols(const void* src, void* dst_u8, float* dst_f32, const int* ymin, const int* ysize, const int* k,
                       int ksize, int B, int H, int W, int Ho, int swap_rb, void* stream);

/* nblk (<= 8) consecutive WindowTransformerBlocks in ONE launch (the loop `for block in self.window_blocks`, model.py:288-289), with
 * the default kernel's geometry (two waves per window, two workgroups per CU): between blocks the residual stream passes through
 * memory as each wave's own stores followed by its own loads (L2), so the launch boundaries and their chip-wide load / store bursts
 * disappear.  x fp32 [64*nwin][192] in place; table: HOST array [nblk][9] of device pointers, per block the arguments of
 * tup_fused_block_fwd after x in that order and packing. */
int tup_fused_blocks32_fwd(float* x, const void* const* table, int nblk, int nwin, void* stream);

/* The same loop (model.py:288-289; blocks :153-172) as the streamed 32x32x16-MFMA kernel (csrc/block_stream.hip): one workgroup of
 * eight waves carries four windows through all nblk blocks; LayerNorm / softmax / GELU instructions sit between the matrix
 * instructions of the same wave.  x fp32 [64*nwin][192]; out_bf16 = NULL: in place; out_bf16 != NULL: the result goes there as bf16
 * [64*nwin][192] (round to nearest even) and x is left holding the kernel's parked intermediate -- for tup_patch_unembed_fwd with
 * x_bf16 = 1, which then reads half the bytes for the same operand.  table: HOST array [nblk][7] of device pointers, per block the
 * tensors of packing.pack_stream_block: wqk bf16 [12][3][32][64], wv bf16 [6][3][32][64], wproj bf16 [6][3][32][64],
 * w1 bf16 [24][3][32][64], w2 fp16 [24][6][32][32], tab fp32 [1536], sbias fp32 [12][2][2][64][16]. */
int tup_blocks_stream_fwd(float* x, void* out_bf16, const void* const* table, int nblk, int nwin, void* stream);

/* Re-packing the weights after an optimizer step (training; replaces the torch index / permute / cat / cast calls of
 * packing.py that follow reference train.py:139 `optimizer.step()`): dst[i] = map[i] < 0 ? 0 : concat(src[0..nparam-1])[map[i]].
 * src: device array of nparam device pointers to the fp32 parameters; offs: device int [nparam + 1] prefix sum of their sizes;
 * map: device int [n] (built once by pack_plan.py from packing.py itself); dst: bf16 (to_bf16 != 0, round to nearest even) or fp32 [n]. */
int tup_pack_gather(const void* src, const int* offs, int nparam, const int* map, void* dst, long long n, int to_bf16, void* stream);

/* torch.optim.Adam's step (reference train.py:104,139; default betas / eps, no weight decay, no amsgrad) for all parameters in one
 * launch.  segs: device array [nseg] of 64-byte records {float* p; const float* g; float* m; float* v; long long n; float step_size
 * (= lr / bias_correction1), inv_sqrt_bc2 (= 1 / sqrt(bias_correction2)), beta2, 1 - beta1, 1 - beta2, eps}; chunks: device int [nchunks][2] =
 * (segment index, first element), one workgroup per 4096 elements.  p, m, v are updated in place. */
int tup_adam_step(const void* segs, const int* chunks, int nchunks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TUPSCALE_HIP_H */
