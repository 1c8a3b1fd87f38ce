/* mickey_hip.h -- C ABI of libmickey_hip.so: the MI355X (gfx950) kernels behind the MicKey
 * inference hot path.
 *
 * The reference (nianticlabs/mickey) has no native code and no FFI: every GPU op is an ATen call
 * made from Python.  Each entry point below therefore cites the reference
 * PYTHON call site(s) (file:line under /root/reference) whose arithmetic it replaces; the Python
 * binding a maintainer adds on the reference side is shown in INTEGRATION.md (ctypes).
 *
 * Conventions
 *   - plain C: raw DEVICE pointers (tensor.data_ptr()), explicit sizes/strides, a stream handle
 *     (hipStream_t passed as void*); no torch types.
 *   - every function returns MK_OK (0) or an MK_ERR_* code; mk_last_error() gives the text.
 *   - functions only enqueue work on `stream`: no allocation, no synchronisation, no host<->device
 *     copies.  Scratch memory is passed in by the caller (sizes documented per function).
 *   - re-entrant: no mutable globals besides the thread-local error string (the process-wide schedule
 *     selectors for benchmarks / tests live in mickey_hip_dev.h, not in this ABI); state a kernel reports
 *     (e.g. the saturation word of the split-operand planes) is a per-call argument.
 *   - "lp" (low precision) buffers hold bf16 or fp16 according to `dtype` (MK_BF16 / MK_F16);
 *     accumulation is always fp32.  dtype MK_F32 is the exact parity mode: "lp" buffers hold fp32 and
 *     every contraction runs on the fp32-input MFMA (the reference's FLOAT16: False path and its
 *     always-fp32 heads, mickey_extractor.py:31-35,49-56); about 16x slower, same entry points.
 */
#ifndef MICKEY_HIP_H
#define MICKEY_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mk_stream_t; /* hipStream_t */

enum { MK_OK = 0, MK_ERR_INVALID_ARGUMENT = 1, MK_ERR_LAUNCH = 2 };
enum { MK_BF16 = 0, MK_F16 = 1, MK_F32 = 2 };
enum { MK_ACT_NONE = 0, MK_ACT_RELU = 1, MK_ACT_GELU = 2 };
/* internal epilogue selectors of the GEMM kernel (exposed for tests/tools only) */
enum { MK_EPI_STORE = 0, MK_EPI_LS_RESIDUAL = 1, MK_EPI_QKV = 2, MK_EPI_PATCH = 3 };

int mk_version(void);
const char* mk_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline (the caller in FRONT of the hot path)
 * ---------------------------------------------------------------------------------------------- */

/* Decoded frames -> model input: replaces cv2.resize + float + permute(2,0,1) + /255 of read_color_image
 * (lib/datasets/utils.py:61-78; same in demo_inference.py:12-29) for n frames at once.
 *   src  uint8 [n, Hs, Ws, 3] RGB on the device, frame stride stride_img bytes (>= Hs*Ws*3)
 *   dst  fp32  [n, 3, H, W] in [0, 1]
 * cv2.resize(uint8, INTER_LINEAR) exactly as OpenCV 4.8.0 computes it: half-pixel centres, edge clamp, 11-bit fixed-point
 * weights (INTER_RESIZE_COEF_BITS) and its rounding of the two passes, the area-fast path for exact 2x decimation; byte-equal to
 * oracle/input_oracle.py.  Hs == H and Ws == W is the identity resize and gives exactly float(v) / 255. */
int mk_preprocess_u8(const unsigned char* src, long long stride_img, int n, int Hs, int Ws, float* dst, int H, int W,
                     mk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder: DINOv2 ViT/14 (reference lib/models/MicKey/modules/DINO_modules/)
 * ---------------------------------------------------------------------------------------------- */

/* out[M,N] = act(A[M,K] . W[N,K]^T + bias).  A, W lp with K contiguous; out lp or fp32.
 * Replaces nn.Linear call sites: mlp.fc1+GELU (layers/mlp.py:36-37) and the heads' small linears
 * (att_layers/transformer_utils.py:55-57,60,64).  K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0. */
int mk_gemm(const void* A, int lda, const void* W, int ldw, const float* bias, void* out, int ldc, int M, int N, int K,
            int act, int out_is_f32, int dtype, mk_stream_t stream);

/* `groups` independent GEMMs of identical shape in one launch (element strides per group; a stride
 * of 0 shares the operand).  Used to run the four heads side by side. */
int mk_gemm_grouped(const void* A, int lda, long long strideA, const void* W, int ldw, long long strideW, const float* bias,
                    long long strideBias, void* out, int ldc, long long strideOut, int groups, int M, int N, int K, int act,
                    int out_is_f32, int dtype, mk_stream_t stream);

/* x[M,N] (fp32, in place) += gamma[N] * (A . W^T + bias): attn.proj+ls1+residual and mlp.fc2+ls2+
 * residual (layers/attention.py:60, mlp.py:39, layer_scale.py:27-28, block.py:105-106). */
int mk_gemm_ls_residual(const void* A, int lda, const void* W, int ldw, const float* bias, const float* gamma, float* x,
                        int ldx, int M, int N, int K, int dtype, mk_stream_t stream);

/* attn.qkv (layers/attention.py:51-53): [nimg*ntok, D] x [3D, D]^T + bias, split per head into
 *   q  [nimg, heads, ntok_pad, 64]  (multiplied by qscale = 64^-0.5 * log2(e): the attention kernel
 *                                    exponentiates with exp2)
 *   k  [nimg, heads, ntok_pad, 64]
 *   vt [nimg, heads, 64, ntok_pad]  V transposed, token index stored with bits 2 and 3 swapped
 * Pad rows (token >= ntok) are never written: the caller zero-fills q/k/vt once.  head_dim is 64
 * for every DINOv2 size (S/B/L/g). */
int mk_gemm_qkv(const void* A, int lda, const void* W, int ldw, const float* bias, void* q, void* k, void* vt, int nimg,
                int ntok, int ntok_pad, int heads, float qscale, int dtype, mk_stream_t stream);

/* Patch embedding (layers/patch_embed.py:76-78) + pos-embed add (dinov2.py:198): A = im2col rows
 * [nimg*npatch, K], W = conv weight [D, K]; x[img, 1+p, :] = A.W^T + bias + pos[1+p, :] (fp32). */
int mk_gemm_patch_embed(const void* A, int lda, const void* W, int ldw, const float* bias, const float* pos, float* x,
                        int nimg, int npatch, int D, int K, int dtype, mk_stream_t stream);

/* im2col for the 14x14/stride-14 patch conv.  img fp32 NCHW (3 channels) with explicit element
 * strides (so the /14 crop of mickey_extractor.py:46 is just a smaller gh/gw); out lp
 * [nimg*gh*gw, ldo], column ch*196 + dy*14 + dx, columns 588..ldo-1 zero. */
int mk_im2col_patch14(const float* img, long long stride_img, long long stride_ch, int stride_row, int nimg, int gh,
                      int gw, void* out, int ldo, int dtype, mk_stream_t stream);

/* x[img, 0, :] = cls + pos[0, :]  (dinov2.py:197-198) */
int mk_cls_token(const float* cls, const float* pos, float* x, int nimg, int ntok, int D, mk_stream_t stream);

/* ---- LayerNorm folded into the GEMMs around it (the encoder's pre-norm blocks, block.py:84-88,105-106) ----
 * y = LN(x) . W^T + b  ==  rstd_m * (x . (W diag(w_ln))^T) - rstd_m * mean_m * colsum + (b + W . b_ln).
 * The residual stream x is kept in HBM as TWO 16-bit planes, x = hi + lo with hi = rn16(x), lo = rn16(x - hi): the 4
 * bytes per element of an fp32 stream, 17 (bf16) / 22 (fp16) mantissa bits -- and hi is, as it stands, the raw A operand
 * of the next GEMM.  The GEMM that PRODUCES rows of x (patch embedding, cls row, proj / fc2 + LayerScale + residual)
 * writes hi / lo and the partial row statistics of the fp32 values it rounded; the GEMM that CONSUMES norm1(x) /
 * norm2(x) (qkv, fc1) reads hi and applies mean / rstd per row in its epilogue.  No stand-alone LayerNorm pass and no
 * extra copy of the token matrix remain in a block (ViT-L: 48 passes of fp32 read + 16-bit write each); mk_layernorm
 * stays for the final norm and for the exact-fp32 mode, where x is a plain fp32 matrix.  16-bit dtypes only.
 *   xh, xl  [rows, ldxs] 16 bit;   stats fp32 [rows][N / 64][2]: (sum, sum of squares) of the row over each 64-column
 *           slot (deterministic: one writer per slot, fixed summation order);   N % 64 == 0.
 *   x_f32_out (mk_gemm_ls_residual_ln): if not NULL the updated rows go there as fp32 [M, ldx] and hi / lo / stats are
 *           left alone -- the last block, whose output feeds the final norm.
 *   colsum fp32 [N]: sum_k W'[n][k] over the 16-bit-ROUNDED folded weights (so that the mean term cancels exactly what
 *           the MFMA accumulates);  bias = b + W . b_ln (fp32, folded on the host);  eps as nn.LayerNorm (1e-6).
 *   Row centring (shift_out / shift_in, fp32 [M], each may be NULL = off): LayerNorm is blind to a constant added to all
 *           channels of a row and nothing but LayerNorm reads the residual stream, so the stream may carry a per-row offset.
 *           A consumer publishes every row's current mean in shift_out; the next producer subtracts shift_in while it adds the
 *           branch output.  Rows then stay centred to within one residual update and the 16-bit hi plane rounds x - mean
 *           instead of x: the folded form's operand rounding becomes that of a LayerNorm OUTPUT at any common-mode level. */
int mk_gemm_ls_residual_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const float* gamma, void* xh,
                           void* xl, int ldxs, float* stats, const float* shift_in, float* x_f32_out, int ldx, int M, int N,
                           int K, int dtype, mk_stream_t stream);
int mk_gemm_patch_embed_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const float* pos, void* xh,
                           void* xl, float* stats, int nimg, int npatch, int D, int K, int dtype, mk_stream_t stream);
int mk_cls_token_ln(const float* cls, const float* pos, void* xh, void* xl, float* stats, int nimg, int ntok, int D, int dtype,
                    mk_stream_t stream);
/* Row centring of a freshly produced stream (after mk_gemm_patch_embed_ln + mk_cls_token_ln): every row of hi + lo gets its
 * own mean subtracted in place and its statistics rewritten, so that the first consumer also meets centred rows. */
int mk_recentre_split(void* xh, void* xl, float* stats, long long rows, int D, int dtype, mk_stream_t stream);
/* consumers: mk_gemm (bias, optional GELU, 16-bit output) and mk_gemm_qkv with A = the hi plane [M, K] (lda == K) */
int mk_gemm_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const float* colsum, const float* stats,
               float eps, float* shift_out, void* out, int ldc, int M, int N, int K, int act, int dtype, mk_stream_t stream);
int mk_gemm_qkv_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const float* colsum, const float* stats,
                   float eps, float* shift_out, void* q, void* k, void* vt, int nimg, int ntok, int ntok_pad, int heads,
                   float qscale, int dtype, mk_stream_t stream);

/* LayerNorm over the last dim (dinov2.py:87,230; block.py:84,87; transformer_utils.py:37-38).
 * x fp32 [*, ldx]; output row r reads input row (r / (rows_per_img - skip)) * rows_per_img + skip +
 * r % (rows_per_img - skip)  (skip = 1 drops the CLS token, dinov2.py:233).  out lp or fp32.
 * If resid != NULL (fp32 [rows_out, ldr]): resid += LN(x) and `out` receives the updated resid
 * (transformer_utils.py:64-66).  wgroup_rows > 0: output rows [g*wgroup_rows, (g+1)*wgroup_rows) use
 * w + g*D, b + g*D (the four heads normalised in one launch).
 * bord_h > 0: `out` is a stack of BORDERED feature maps (below, mk_bordered_rows) of bord_m = nimg * bord_h * bord_w pixels
 * each: output row r goes to row (r / bord_m) * mk_bordered_rows(nimg, bord_h, bord_w) + bordered(r % bord_m).  The
 * border rows are not written.  bord_h = 0: dense rows. */
int mk_layernorm(const float* x, int ldx, const float* w, const float* b, float eps, void* out, int ldo, int out_is_f32,
                 float* resid, int ldr, int rows_out, int D, int rows_per_img, int skip, int wgroup_rows, int bord_h,
                 int bord_w, int bord_m, int dtype, mk_stream_t stream);

/* A 128-wide linear and the LayerNorm behind it in one pass (the linear-attention layers of the heads: merge -> norm1 and
 * mlp[2] -> norm2 + the layer's residual, att_layers/transformer_utils.py:58-66): per group g and row r
 *   y = LayerNorm_128(A[g][r, :K] . W[g]^T) * ln_w[g] + ln_b[g];   resid != NULL: resid[g*M + r] += y, out = the updated resid
 * A lp [groups][M, lda], W lp [groups][128, ldw] (no bias), ln_w / ln_b fp32 [groups, 128], resid fp32 [groups * M, ldr], out lp
 * [groups * M, ldo] dense, or (bord_h > 0) a stack of bordered feature maps as in mk_layernorm.  K a multiple of 32, <= 256.  The
 * fp32 GEMM output never reaches memory; K steps are summed in order (the accumulators are mk_gemm_grouped's, bit for bit). */
int mk_gemm_ln128(const void* A, int lda, long long strideA, const void* W, int ldw, long long strideW, const float* ln_w,
                  const float* ln_b, float eps, float* resid, int ldr, void* out, int ldo, int groups, int M, int K, int bord_h,
                  int bord_w, int dtype, mk_stream_t stream);

/* The same LayerNorm writing its rows as the (hi, lo) fp16 operand planes of the split-operand head kernels (mk_conv3x3_split,
 * mk_gemm_grouped_split; AMD.HEADS_DTYPE: split): LN(x) * plane_scale = hi + lo, both [.., ldo] fp16, dense or bordered as above
 * (no fp32 copy, no separate mk_split_planes pass).  out_lo == NULL: only the hi plane is written -- the rows ROUNDED to fp16,
 * what the reference's fp16 encoder hands its heads (mickey_extractor.py:49-52: forward_features(x.to(amp_dtype)) ... .float());
 * a zero-initialised lo plane then stays zero.  sat_flag: see mk_split_planes. */
int mk_layernorm_planes(const float* x, int ldx, const float* w, const float* b, float eps, void* out_hi, void* out_lo, int ldo,
                        float plane_scale, float* resid, int ldr, int rows_out, int D, int rows_per_img, int skip, int wgroup_rows,
                        int bord_h, int bord_w, int bord_m, int* sat_flag, mk_stream_t stream);

/* Non-causal multi-head attention, softmax(q k^T) v with head_dim 64 (layers/attention.py:53-59),
 * flash style (the ntok x ntok matrix is never materialised).  q/k/vt as written by mk_gemm_qkv;
 * out lp [nimg*ntok, ldo] with column head*64 + d. */
int mk_flash_attn_fwd(const void* q, const void* k, const void* vt, void* out, int ldo, int nimg, int heads, int ntok,
                      int ntok_pad, int dtype, mk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Heads (reference lib/models/MicKey/modules/mickey_extractor.py:67-251)
 * ---------------------------------------------------------------------------------------------- */

/* BORDERED feature map: the layout of every activation a 3x3 convolution reads.  Pixel (b, y, x) of a [nimg, H, Wd]
 * grid is row ((b (H+1) + y + 1)(Wd+1) + x + 1) of a [mk_bordered_rows(nimg, H, Wd), C] matrix; all other rows are the
 * zero padding of nn.Conv2d(padding=1) (one shared column between grid rows, one shared grid row between images) and
 * must be zero -- allocate the buffer zeroed once; no kernel of this library ever writes them.  Every 3x3 tap of every
 * pixel is then the same row shift dy (Wd+1) + dx: the implicit GEMM addresses its A operand like a dense matrix (one
 * scalar-addressed LDS-DMA instruction per 8 rows; per-lane tap / border arithmetic cost the round-2 kernel 8 %). */
long long mk_bordered_rows(int nimg, int H, int Wd);

/* 3x3 conv, pad 1, as implicit GEMM over bordered NHWC lp activations, BatchNorm folded into W/bias by the
 * caller (utils/extractor_utils.py:28-35):
 *   out[g][pix, co] = act( sum_{tap,ci} W[g][co, tap*C1+ci] * in1[g][pix+tap, ci]
 *                        + sum_{ci} W[g][co, 9*C1+ci] * in2[g][pix, ci]      (1x1 shortcut, optional)
 *                        + bias[g][co] + resid[g][pix, co] (identity shortcut, optional) )
 * tap = ky*3+kx.  in1, in2, resid: bordered feature maps (resid [.., Cout]).  out_kind: MK_CONV_OUT_DENSE = lp rows
 * [nimg*H*Wd, Cout]; MK_CONV_OUT_BORDERED = lp, bordered (the next conv's input); MK_CONV_OUT_F32 = fp32 dense rows.
 * C1, C2 % 64 == 0; Cout % 4 == 0.  Strides are element strides per group (0 = shared input). */
enum { MK_CONV_OUT_DENSE = 0, MK_CONV_OUT_BORDERED = 1, MK_CONV_OUT_F32 = 2 };
int mk_conv3x3(const void* in1, long long stride_in1, int C1, const void* in2, long long stride_in2, int C2, const void* W,
               int ldw, long long strideW, const float* bias, long long strideBias, const void* resid, long long strideResid,
               void* out, int Cout, long long strideOut, int groups, int nimg, int H, int Wd, int act, int out_kind, int dtype,
               mk_stream_t stream);

/* The same convolution with SPLIT operands -- fp32-grade products on the 16-bit matrix cores (the reference runs its heads in
 * fp32, mickey_extractor.py:53-56; the fp32-input MFMA is 1/16 of the 16-bit rate).  Activations and weights are each held as
 * two fp16 planes, x * s = hi + lo (22 mantissa bits; s a power of two that keeps lo out of fp16's subnormals), and every
 * product is evaluated as hi.hi + lo.hi + hi.lo with fp32 accumulation.  SHARED operands (round 6): a K step of 32 stages the
 * four planes a_hi, a_lo, w_hi, w_lo ONCE in LDS (one 128-byte row = 32 hi | 32 lo elements) and issues the three MFMA sets from
 * those fragments -- L2 -> LDS bytes and LDS fragment reads per product are 2/3 of three plain sweeps'.
 *   in1_hi / in1_lo, in2_hi / in2_lo: bordered fp16 feature maps (mk_split_planes of the fp32 maps, scale s_a); the two planes
 *      of a source must lie within 2 GiB of each other (one allocation); in1_lo == NULL (then without in2): the source IS its
 *      hi plane -- fp16 features of an fp16 encoder -- and hi_w . lo_a is not computed: two MFMA sets per K step instead of three;
 *   W: fp16 [Cout, 2 K], K = 9 C1 + C2: the BatchNorm-folded weights times s_w as INTERLEAVED planes -- for every block of 32
 *      K columns, 32 hi values then 32 lo values (weights.split_conv_weight); an identity shortcut is passed as in2 = the
 *      block input with identity columns in W;
 *   out: out_lo == NULL: fp32 [.., Cout], dense rows or (out_bordered) a bordered feature map; out_lo != NULL: the result
 *      goes out as the NEXT split conv's operand planes instead, fp16 out = hi, out_lo = lo of result * plane_scale (same
 *      layout; saturating at fp16's largest finite value, NaN kept); acc_scale = 1 / (s_a s_w), applied to the accumulators
 *      before bias / activation;
 *   sat_flag: NULL, or a zero-initialised int in device memory that gets bit 0 set when a plane value had to be clamped
 *      (|result * plane_scale| > 65504) or was NaN -- per call, so that two models / streams never share a word.
 *   C1, C2 % 32 == 0. */
int mk_conv3x3_split(const void* in1_hi, const void* in1_lo, long long stride_in1, int C1, const void* in2_hi, const void* in2_lo,
                     long long stride_in2, int C2, const void* W, int ldw, long long strideW, const float* bias,
                     long long strideBias, void* out, void* out_lo, int Cout, long long strideOut, int groups, int nimg, int H,
                     int Wd, int act, int out_bordered, float acc_scale, float plane_scale, int* sat_flag, mk_stream_t stream);

/* fp32 [rows, cols] (row stride ld_src) -> fp16 planes hi = rn16(x scale), lo = rn16(x scale - hi) (saturating at +-65504, NaN
 * kept), each [rows, cols] with row stride ld_dst.  cols, ld_src, ld_dst % 4 == 0.  sat_flag: NULL, or a zero-initialised
 * device int that gets bit 0 set when a value was clamped or NaN (activations are held as x * 64: |x| > 1023 saturates). */
int mk_split_planes(const float* src, long long rows, int cols, long long ld_src, float scale, void* hi, void* lo,
                    long long ld_dst, int* sat_flag, mk_stream_t stream);

/* Grouped GEMM with SPLIT operands (the small linears of the heads' attention layers in AMD.HEADS_DTYPE: split; reference
 * att_layers/transformer_utils.py:51-66 runs them in fp32): out[g] = act(A[g] W[g]^T + bias[g]) with A = (A_hi + A_lo) / s_a
 * as fp16 planes [M, lda] (within 2 GiB of each other) and W fp16 [N, 2 K] = the interleaved (32 hi | 32 lo) planes of the
 * weights times s_w -- hi.hi + lo.hi + hi.lo on the 16-bit matrix cores from operands staged once, fp32 accumulation (as
 * mk_conv3x3_split).  out: fp32 [M, ldc], or with out_lo != NULL the (hi, lo) planes of result * plane_scale.  K % 32 == 0. */
int mk_gemm_grouped_split(const void* A_hi, const void* A_lo, int lda, long long strideA, const void* W, int ldw, long long strideW,
                          const float* bias, long long strideBias, void* out, void* out_lo, int ldc, long long strideOut, int groups,
                          int M, int N, int K, int act, float acc_scale, float plane_scale, int* sat_flag, mk_stream_t stream);

/* Start of Transformer_self_att (att_layers/transformer.py:92-95): xs = x + pe (fp32 stream) and an
 * lp copy into columns [0,C) of a [rows, ld_cat] buffer.  x lp [G][rows, C]; pe fp32 [npix, C] or NULL. */
int mk_posenc_add(const void* x, const float* pe, float* xs, void* cat, int ld_cat, int groups, int nimg, int npix, int C,
                  int dtype, mk_stream_t stream);

/* Linear attention (att_layers/attention.py:46-64), heads of 16 channels.
 * qkv fp32 [G][nimg*L, 3C] (q | k | v).  Step 1 computes per (g, img, head)
 *   KV[16][16] = sum_s (elu(k_s)+1) (x) v_s / L   and   Ksum[16] = sum_s (elu(k_s)+1)
 * into kv fp32 [G*nimg*(C/16)][272], deterministically (per-chunk partials in `work`, then a fixed-
 * order reduction).  work: mk_linattn_work_floats(G, nimg, L, C) fp32 elements. */
long long mk_linattn_work_floats(int groups, int nimg, int L, int C);
int mk_linattn_kv(const float* qkv, float* kv, float* work, int groups, int nimg, int L, int C, mk_stream_t stream);
/* Step 2: msg[s, h*16+v] = (phi(q_s) . KV[:, v]) * L / (phi(q_s) . Ksum + 1e-6), written lp to
 * out [G][nimg*L, ldo]. */
int mk_linattn_apply(const float* qkv, const float* kv, void* out, int ldo, int groups, int nimg, int L, int C, int dtype,
                     mk_stream_t stream);

/* Head tails (mickey_extractor.py:134-138,173-176,213-216,246-249 and
 * compute_correspondences.py:20-31).  feat* fp32 [nimg*h*w, C] (resblock4 outputs).
 *   scr   [nimg, h*w]      border-masked temperature-100 softmax (or sigmoid) of w_score . feat_det
 *   kps   [nimg, 2, h*w]   (sigmoid(w_xy . feat_off) + cell) * down   (x row, then y row)
 *   depth [nimg, h*w]      w_depth . feat_depth  (max_depth * sigmoid(.) if use_depth_sigmoid)
 *   dsc   [nimg, Cd, h*w]  feat_dsc / sqrt(sum_c feat_dsc^2 + 1e-10) if norm_dsc, transposed */
int mk_head_tails(const float* feat_det, const float* w_score, const float* feat_off, const float* w_xy,
                  const float* feat_depth, const float* w_depth, const float* feat_dsc, float* scr, float* kps, float* depth,
                  float* dsc, int nimg, int h, int w, int C, int Cd, int border, int use_softmax, int use_depth_sigmoid,
                  float max_depth, int norm_dsc, float down, mk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Matcher (reference lib/models/MicKey/modules/utils/feature_matcher.py)
 * ---------------------------------------------------------------------------------------------- */

/* dualSoftmax.forward (feature_matcher.py:64-83) + kp_matrix_scores (compute_correspondences.py:46-50)
 * + final_scores (compute_pose.py:23), exact-fp32 MFMA correlation.
 *   dsc0 [B, C, n0], dsc1 [B, C, n1] fp32;  scr0 [B, n0], scr1 [B, n1] fp32 (may be NULL with kp/final NULL)
 *   scores / kp_scores / final_scores [B, n0, n1] fp32, each may be NULL (lean mode).
 *   use_dustbin: append the scalar `dustbin` as extra row/column/corner before both softmaxes.
 *   work: fp32 scratch, mk_dual_softmax_work_floats(B, n0, n1, own_copy) elements, 16-byte aligned: the softmax partials,
 *          plus -- with own_copy != 0 -- one [B, n0, n1] copy of the scaled correlation.  The copy lives in `scores` /
 *          `final_scores` when one of them is requested: pass own_copy = 1 only for a call with scores == final_scores ==
 *          NULL (kp_scores alone). */
long long mk_dual_softmax_work_floats(int B, int n0, int n1, int own_copy);
int mk_dual_softmax(const float* dsc0, const float* dsc1, const float* scr0, const float* scr1, float inv_temperature,
                    int use_dustbin, float dustbin, float* scores, float* kp_scores, float* final_scores, float* work, int B,
                    int C, int n0, int n1, mk_stream_t stream);

/* dualSoftmax.forward with the descriptor correlation on the 16-bit matrix cores (BASELINE.json configs[4]: "fp16 MFMA
 * descriptor correlation"; reference feature_matcher.py:65 is an fp32 matmul): every descriptor entry is split into fp16 hi + lo
 * parts (x 2^10 = hi + lo, 22 mantissa bits) and x0.x1 evaluated as lo.hi + hi.lo + hi.hi on v_mfma_f32_32x32x16_f16 with fp32
 * accumulation -- fp32-grade values (<= 1e-5 rel vs the fp32 reference, same row / column arg-max), a sixth of the matrix
 * time of the exact fp32 MFMA; the correlation is evaluated twice (statistics, then outputs) instead of being stored.
 * PRECONDITIONS (else use mk_dual_softmax): C == 128; |dsc| <= 1 everywhere (L2-normalised descriptors,
 * MICKEY.DSC_HEAD.NORM_DSC: True, extractor_utils.py:6-10); inv_temperature * log2(e) <= 100 (the row / column sums are
 * taken without a running maximum).  Arguments as mk_dual_softmax; work: mk_dual_softmax_split_work_floats fp32 elements. */
long long mk_dual_softmax_split_work_floats(int B, int n0, int n1);
int mk_dual_softmax_split(const float* dsc0, const float* dsc1, const float* scr0, const float* scr1, float inv_temperature,
                          int use_dustbin, float dustbin, float* scores, float* kp_scores, float* final_scores, float* work,
                          int B, int C, int n0, int n1, mk_stream_t stream);

/* sinkhorn.forward (feature_matcher.py:93-137): 10 log-domain iterations on u, v only.
 *   scr0/scr1/kp_scores/final_scores as in mk_dual_softmax (optional).
 *   work: mk_sinkhorn_work_floats(B, n0, n1) fp32 elements (holds the (n0+1)x(n1+1) coupling matrix). */
long long mk_sinkhorn_work_floats(int B, int n0, int n1);
int mk_sinkhorn(const float* dsc0, const float* dsc1, const float* scr0, const float* scr1, float alpha, int iters,
                float* scores, float* kp_scores, float* final_scores, float* work, int B, int C, int n0, int n1,
                mk_stream_t stream);

/* featureMatcher.get_matches_list (feature_matcher.py:19-46), batched: mutual nearest neighbours on
 * scores[b, :n0-1, :n1-1], sorted by score descending.  matches int32 [B, n0, 2] (row i, col j),
 * count int32 [B].  work: 2*B*(n0+n1) ints. */
int mk_mutual_nn(const float* scores, int* matches, int* count, int* work, int B, int n0, int n1, mk_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Probabilistic Procrustes solver (reference .../utils/probabilisticProcrustes.py:183-348)
 * ---------------------------------------------------------------------------------------------- */

/* Weighted sampling without replacement == top-k of p / Exp(1) ("exponential race"); what
 * torch.multinomial(p, k) computes (probabilisticProcrustes.py:230-231).  For every pair b and every
 * r in [0, rows_per_pair): idx[b*rows_per_pair + r, 0..k) = indices of the k largest p[b, c] / e[b, r, c],
 * in descending key order (the order torch.topk returns).
 *   p      [B, ncell] fp32, >= 0
 *   noise  NULL: the race is drawn on device from Philox4x32-10(seed, offset) -- not cell by cell: with the threshold T of the
 *          keys known from the histogram of p, the candidates {p / e > T} of a row are generated directly as a thinned
 *          Bernoulli process (geometric skipping under a per-16-cell bound of p, acceptance s / S, e drawn from Exp(1)
 *          conditioned on e < p / T; mk_solver.hip), the same law as drawing every e; else fp32 [B*rows_per_pair, ncell]
 *          injected Exp(1) draws (tests: bit-comparable with torch)
 *   idx    int32 [B*rows_per_pair, k];  cnt int32 [B*rows_per_pair] = number of non-zero-key entries
 *          actually available (< k only if fewer than k cells have p > 0; the tail is then filled
 *          with zero-probability cells in ascending index order)
 *   invalid int32 [1] or NULL: OR-ed with 1 when torch.multinomial would have raised (a NaN / inf /
 *          negative probability, or a row without any positive cell) -- the reference then returns
 *          the zero pose for the whole batch (probabilisticProcrustes.py:331-336)
 *   work   bytes: mk_exprace_topk_work_bytes(B, rows_per_pair, k, ncell), 16-byte aligned.  Its first
 *          mk_exprace_topk_state_bytes(B, rows_per_pair) bytes are SELF-CLEANING state (per-row candidate counts, per-pair
 *          flags and histograms, arrival counters): they must be ZERO when a call starts and every call leaves them zero, so
 *          the chain carries no zero-fill launch -- zero the buffer once when it is allocated and reuse it; a buffer must not
 *          be shared by calls that may run concurrently (two streams).  The chain is four launches: histogram of p (+ the
 *          threshold as its tail), the collect pass (+ the shortfall check as its tail), the exact fallback (idle unless a
 *          pair came up short), the select kernel.
 *   pair_base  GLOBAL index of pair 0 of this call.  The Philox streams are keyed by (seed, offset, global pair index,
 *          draw, cell), so a batch may be split arbitrarily -- over calls or over the GPUs of a node -- without changing
 *          any pair's draws: pair i of a B = 32 call and the same pair alone with pair_base = i sample identically. */
long long mk_exprace_topk_work_bytes(int B, int rows_per_pair, int k, long long ncell);
long long mk_exprace_topk_state_bytes(int B, int rows_per_pair);
int mk_exprace_topk(const float* p, const float* noise, unsigned long long seed, unsigned long long offset,
                    const unsigned long long* offset_dev, int* idx, int* cnt, int* invalid, void* work, int B, int rows_per_pair,
                    long long ncell, int k, int pair_base, mk_stream_t stream);

/* *counter += inc on the stream.  `offset_dev` of mk_exprace_topk / mk_ransac_hypotheses (may be NULL) is such a
 * counter: the Philox stream offset used is offset + *offset_dev, read on the device when the kernel runs, so a
 * captured hipGraph of the forward draws fresh samples at every replay (the reference advances torch's generator). */
int mk_counter_add(unsigned long long* counter, unsigned long long inc, mk_stream_t stream);

/* Index decode + gathers + back-projection (probabilisticProcrustes.py:233-244, training_utils.py:7-22):
 * for every sampled cell c = idx[r, s]: i = c / n1 (image-0 keypoint), j = c % n1;
 *   X[r, s, :] = depth0[b, i] * K0[b]^-1 [kps0[b, :, i], 1],  Y likewise from image 1,
 *   wts[r, s] = final_scores[b, c],  corr[r, s, :] = (u0, v0, u1, v1, d0, d1)   (for the inlier list)
 * kps [B, 2, n], depth [B, n], K [B, 3, 3] fp32.  r = b*rows_per_pair + o. */
int mk_gather_backproject(const int* idx, const float* final_scores, const float* kps0, const float* depth0,
                          const float* kps1, const float* depth1, const float* K0, const float* K1, float* X, float* Y,
                          float* wts, float* corr, int B, int rows_per_pair, int k, int n0, int n1, mk_stream_t stream);

/* Backward of mk_gather_backproject w.r.t. keypoints and depths (training: loss_class.py:139-146 + training_utils.py:7-22 under
 * autograd; intrinsics and scores get no gradient).  gX, gY [B*rows_per_pair, k, 3] = dL/dX, dL/dY; corr as written by the forward;
 * gkps [B, 2, n], gdepth [B, n] must be ZEROED by the caller: contributions are accumulated with fp32 atomic adds (a keypoint is
 * drawn by many cells; the order of the additions is not fixed, as in torch's index backward). */
int mk_gather_backproject_bwd(const int* idx, const float* corr, const float* gX, const float* gY, const float* K0, const float* K1,
                              float* gkps0, float* gdepth0, float* gkps1, float* gdepth1, int B, int rows_per_pair, int k, int n0,
                              int n1, mk_stream_t stream);

/* Hypothesis generation and scoring (probabilisticProcrustes.py:247-268; loss/solvers.py:31-52;
 * training_utils.py:55-61).  For every correspondence set r (k points) and every h in [0, it_ransac):
 * draw 3 of k without replacement weighted by wts (exponential race; noise3 = NULL: Philox, else fp32
 * [R*it_ransac, k] injected; idx3_in != NULL injects the 3 indices directly), fit R,t by Kabsch on the
 * 3 point pairs, score = sum_j sigmoid(5/th * (th - sqrt(|R X_j + t - Y_j|^2 + 1e-6))).
 *   Rh [R*it_ransac, 9], th [R*it_ransac, 3], score [R*it_ransac], idx3 int32 [R*it_ransac, 3] (out)
 *   set_base  GLOBAL index of correspondence set 0 of this call (= pair_base * rows_per_pair): keys the Philox streams of
 *          the 3-samples the same way as pair_base in mk_exprace_topk */
int mk_ransac_hypotheses(const float* X, const float* Y, const float* wts, const float* noise3, const int* idx3_in,
                         unsigned long long seed, unsigned long long offset, const unsigned long long* offset_dev,
                         float th_soft, float* Rh, float* th, float* score, int* idx3, int nsets, int it_ransac, int k,
                         long long set_base, mk_stream_t stream);

/* Arg-max over a pair's hypotheses, <= num_ref rounds of {hard-inlier recount, masked weighted Kabsch},
 * final confidence (probabilisticProcrustes.py:275-303, loss/solvers.py:13-26, training_utils.py:71-75).
 * One workgroup per pair.  hyp_per_pair = it_matches*it_ransac; the winning set is best / it_ransac.
 *   R [B, 9], t [B, 3], conf [B], best int32 [B], inl_mask uint8 [B, k] (hard inliers of the final pose),
 *   rounds int32 [B] (refits done), invalid int32 [1]: set to 1 if any hypothesis of the BATCH is
 *   non-finite (the reference then zeroes the whole batch, :261-262,329-342; applied by mk_pose_finalize). */
int mk_refine_pose(const float* X, const float* Y, const float* Rh, const float* th, const float* score, float th_inlier,
                   int num_ref, int min_inliers, float* R, float* t, float* conf, int* best, unsigned char* inl_mask,
                   int* rounds, int* invalid, int B, int it_matches, int it_ransac, int k, mk_stream_t stream);

/* If *invalid != 0: zero R, t, conf for the whole batch (reference :338-342). */
int mk_pose_finalize(float* R, float* t, float* conf, const int* invalid, int B, mk_stream_t stream);

/* ---- training-time RANSAC -----------------------------------------------------------------------
 * The no-grad core of MetricPoseLoss.single_iteration_RANSAC (reference lib/models/MicKey/modules/loss/loss_class.py:141-184):
 * for every sampled match set r in [0, nsets) (S matches: X, Y [nsets, S, 3], wts [nsets, S]) and every hypothesis
 * h in [0, it_ransac): draw num_corr matches without replacement ~ wts (:148; exponential race; noise = NULL: Philox
 * keyed by (seed, offset, set_base * it_ransac + r * it_ransac + h); noise != NULL: fp32 Exp(1) [nsets*it_ransac, S]
 * injected; idx_in != NULL: int32 [nsets*it_ransac, num_corr] indices injected), then <= num_ref refinement rounds
 * {masked weighted Procrustes over the current set (loss/solvers.py:13-26,45-52) -> matches within th_ref
 * (training_utils.py:71-75) -> accept iff their number grew} exactly as the reference's masked tensor updates do (:152-184).
 *   final_mask float32 [nsets*it_ransac, S]  `inliers_final`: 0/1, the set that produced the last accepted pose -- the
 *                                            weights of the differentiable Procrustes the caller runs next (:187)
 *   idx_out    int32 [nsets*it_ransac, num_corr]  the drawn matches, in draw order (descending race key)
 *   rounds     int32 [nsets*it_ransac]       accepted refinement rounds
 * S <= 1024, 3 <= num_corr <= S. */
int mk_train_ransac_masks(const float* X, const float* Y, const float* wts, const float* noise, const int* idx_in,
                          unsigned long long seed, unsigned long long offset, const unsigned long long* offset_dev,
                          float th_ref, int num_ref, int num_corr, float* final_mask, int* idx_out, int* rounds, int nsets,
                          int it_ransac, int S, long long set_base, mk_stream_t stream);

/* The DIFFERENTIABLE tail of the same function (loss_class.py:187-246), forward and backward: for every hypothesis
 * hyp = r * it_ransac + h of every match set r (pair b = r / it_matches)
 *   forward   weighted_procrustes(X_r, Y_r, mask_hyp) (loss/solvers.py:13-26,45-52: centroids with w / (sum|w| + 1e-16),
 *             H = A_c^T (w B_c), H = U S V^T, R = V diag(1, 1, det(U V^T)) U^T, t = b_mean - a_mean R^T), the soft inlier score
 *             sum_j sigmoid(5 / th (th - |R x_j + t - y_j|)) over all S matches (training_utils.py:55-61) and the loss:
 *             loss_type 0 = compute_vcre_loss (loss_utils.py:41-69 with metrics.py:56-80: the 7 x 4 x 7 eye grid projected with
 *             K0 / K1, clipped to [0, img_h], tanh(. / 80) with soft_clip), 1 = compute_pose_loss (:27-39);
 *               out   float32 [nhyp, 4]  (loss_value, rot_angle_loss, trans_l1_loss, score)
 *               Rt    float32 [nhyp, 12] (R row-major, t)
 *               saved float32 [nhyp, 32] (U, V, S, det sign, centroids, sum of weights: for the backward pass)
 *   backward  grad_out float32 [nhyp, 2] = dL/d(loss_value, score)  ->  gX, gY float32 [nsets, S, 3] = dL/dX, dL/dY, summed over
 *             the it_ransac hypotheses of a set in hypothesis order (no atomics); the SVD is differentiated in closed form
 *             (the adjoint of H -> R; finite for equal singular values, where torch.svd's backward is not);
 *             work float32 [nhyp, 16] scratch.  mask, the poses and the intrinsics receive no gradient (the reference detaches
 *             them: the mask comes out of torch.no_grad, :152-184).
 * X, Y [nsets, S, 3], mask [nsets * it_ransac, S], Rgt [B, 9], tgt [B, 3], K0, K1 [B, 9] (VCRE only), S <= 1024. */
int mk_train_tail_fwd(const float* X, const float* Y, const float* mask, const float* Rgt, const float* tgt, const float* K0,
                      const float* K1, int nsets, int it_ransac, int S, int it_matches, float th_soft, int loss_type,
                      int soft_clip, float img_h, float* out, float* Rt, float* saved, mk_stream_t stream);
int mk_train_tail_bwd(const float* X, const float* Y, const float* mask, const float* Rgt, const float* tgt, const float* K0,
                      const float* K1, int nsets, int it_ransac, int S, int it_matches, float th_soft, int loss_type,
                      int soft_clip, float img_h, const float* Rt, const float* saved, const float* grad_out, float* work,
                      float* gX, float* gY, mk_stream_t stream);

/* The reductions between the tail and the loss (loss_class.py:229-246, :263-268), forward and backward:
 *   forward   per set (row = b*it_matches + o) of it_ransac hypotheses, from out [nhyp, 4] of mk_train_tail_fwd:
 *               sm      = softmax over [score_k / temperature (, null_score / temperature if add_null)]
 *               loss_value[row] = sum_k sm_k loss_k (+ sm_null * null_loss)                                   (:240-246)
 *               rot / trans     = sum_k softmax(score / temperature)_k (rot_k | trans_k), WITHOUT the null column     (:229-238)
 *               coef [nhyp, 2]  = d loss_value[row] / d (loss_k, score_k) = (sm_k, sm_k (loss_k - loss_value[row]) / temperature)
 *             per pair: per_pair [B, 3] = sums over its it_matches rows of (loss_value, rot, trans), in row order        (:263-268)
 *             flags int32 [2] (caller zeroes): [0] |= 1 if any R / t of Rt [nhyp, 12] is not finite (:225-227), [1] += number of
 *             hypotheses whose cross-covariance has rank one (singular values saved[:, 18:21]; torch.linalg.matrix_rank, :190).
 *   backward  grad_out [nhyp, 2] = g_pair[b] * coef  (g_pair [B] = dL/d per_pair[:, 0]; what mk_train_tail_bwd takes)
 * it_matches <= 64. */
int mk_train_aggregate_fwd(const float* out, const float* Rt, const float* saved, int B, int it_matches, int it_ransac,
                           float temperature, int add_null, float null_loss, float null_score, float* loss_value, float* per_pair,
                           float* coef, int* flags, mk_stream_t stream);
int mk_train_aggregate_bwd(const float* coef, const float* g_pair, int B, int it_matches, int it_ransac, float* grad_out,
                           mk_stream_t stream);

/* REINFORCE bookkeeping of the same function (loss_class.py:251-261, a python loop over B*it_matches rows in the
 * reference): for row = b*it_matches + r, r ascending, and every sampled cell c = idx[row, s]:
 *   gradients[b, c] += loss_value[row];  gradients_b[b, c] += 1
 * fp32, rows applied in the reference's order (bit-identical sums).  Cells of one row must be distinct (they come from
 * sampling without replacement).  idx int32 [B*it_matches, S]; gradients, gradients_b float32 [B, ncell], zeroed by the
 * caller. */
int mk_reinforce_scatter(const int* idx, const float* loss_value, float* gradients, float* gradients_b, int B,
                         int it_matches, int S, long long ncell, mk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MICKEY_HIP_H */
