/* libi2r_hip.so -- C ABI of the MI355X (gfx950) masked-diffusion hot path of ImagenI2R.
 *
 * The reference (azencot-group/ImagenI2R) has no FFI: its boundary for this path is the Python object API
 * (TS2img_Karras / EDMPrecond / DiffusionProcess / DelayEmbedder).  The host-side mirror of that API lives in
 * the imageni2r_amd Python package; underneath it every arithmetic step of the path is one of the entry points below.  Each entry
 * point cites the reference code it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous memory (NCHW fp32 unless stated); the library never allocates,
 *     frees or synchronises: outputs and workspaces are caller-owned (torch-allocated in the Python mirror);
 *   - `stream` is a hipStream_t; all work is enqueued on it asynchronously and is safe under hipGraph stream capture;
 *   - return value 0 = ok, negative = error (I2R_ERR_*), message via i2r_last_error() (thread-local);
 *     shape/aliasing preconditions are checked on the host before anything is launched.
 */
#ifndef I2R_HIP_H
#define I2R_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* i2r_stream_t; /* hipStream_t */

int i2r_version(void);
const char* i2r_last_error(void);

/* ---- delay embedding: models/img_transformations.py ------------------------------------------------------------ */
/* DelayEmbedder.ts_to_img + pad_to_square (:79-108, :59-70), fused with the observed-mask derivation
 * run_irregular.py:216 (mask may be NULL).  ts (B,L,C) -> img, mask (B,C,S,S).  cols = image columns before padding. */
int i2r_ts_to_img_f32(const float* ts, float* img, float* mask, int B, int L, int C, int delay, int E, int cols,
                      int S, float pad_value, i2r_stream_t stream);
/* adjoint of the above (autograd's CopySlices chain through :91/:98): dts (B,L,C) = gather-sum of dimg */
int i2r_ts_to_img_bwd_f32(const float* dimg, float* dts, int B, int L, int C, int delay, int E, int cols, int S,
                          i2r_stream_t stream);
/* DelayEmbedder.img_to_ts (:140-170): mean over the cells of each time step, summed in ascending image row in the
 * input dtype, true division by the count, stored fp32.  f64 variant = what the sampler's fp64 output goes through. */
int i2r_img_to_ts_f32(const float* img, float* ts, int B, int L, int C, int delay, int E, int cols, int S,
                      i2r_stream_t stream);
int i2r_img_to_ts_f64(const double* img, float* ts, int B, int L, int C, int delay, int E, int cols, int S,
                      i2r_stream_t stream);

/* ---- convolutions: models/networks.py:68-90 (Conv2d.forward, non-fused resample path) ----------------------------- */
/* out = conv2d(x, w, padding = ksize/2) + bias + residual   (bias, residual may be NULL; ksize 1 or 3).
 * `residual` fuses the x.add_(skip) / proj(...).add_(x) of UNetBlock.forward (:178, :185).
 * The kernels read weights pre-packed K-major (Wp[Kpad][Cout4]); pack once per weight version: */
long i2r_conv2d_packed_floats(int Cout, int Cin, int ksize, int dgrad);
int i2r_conv2d_pack_weight_f32(const float* w /* OIHW */, float* wp, int Cout, int Cin, int ksize, int dgrad,
                               i2r_stream_t stream);
/* all conv weights of a network in one launch: table = device (T,6) int64 rows {w, wp, Cout, Cin, ksize, dgrad},
 * blockmap = device (nblocks,2) int32 {tensor, block}, one block per pack_block_floats packed floats (query below) */
int i2r_conv2d_pack_block_floats(void);
int i2r_conv2d_pack_multi_f32(const void* table, const void* blockmap, int nblocks, i2r_stream_t stream);
/* floats of split-K workspace this shape wants (0 = none); passing less (or NULL) only disables split-K */
long i2r_conv2d_splitk_floats(int B, int Cin, int H, int W, int Cout, int ksize);
int i2r_conv2d_fwd_packed_f32(const float* x, const float* wp, const float* bias, const float* residual, float* out,
                              float* splitk_ws, long splitk_floats, int B, int Cin, int H, int W, int Cout, int ksize,
                              i2r_stream_t stream);
/* same, packing w (OIHW) into ws on the fly; ws_floats >= packed_floats (+ splitk_floats to enable split-K) */
int i2r_conv2d_fwd_f32(const float* x, const float* w, const float* bias, const float* residual, float* out,
                       float* ws, long ws_floats, int B, int Cin, int H, int W, int Cout, int ksize,
                       i2r_stream_t stream);
/* autograd of conv2d w.r.t. x: dx = conv2d(dy, flip(w)^T) = the forward kernel on weights packed with dgrad=1;
 * ws_floats >= i2r_conv2d_packed_floats(Cout, Cin, ksize, 1) (+ split-K floats of the transposed problem). */
int i2r_conv2d_dgrad_f32(const float* dy, const float* w, float* ws, long ws_floats, float* dx, int B, int Cin,
                         int H, int W, int Cout, int ksize, i2r_stream_t stream);
/* Inference-only forward on the HALF-precision matrix cores with fp32-grade accuracy ("f16x2 split": x = hi + lo in
 * fp16, three f16 MFMAs per tile-step, fp32 accumulate; ~1e-6 relative error; requires |x| < 65504).  Same contract as
 * i2r_conv2d_fwd_packed_f32 with weights packed by the h2 pack routine below (its byte count from the _bytes query). */
long i2r_conv2d_packed_h2_bytes(int Cout, int Cin, int ksize);
int i2r_conv2d_pack_weight_h2(const float* w /* OIHW */, void* wp, int Cout, int Cin, int ksize, i2r_stream_t stream);
long i2r_conv2d_splitk_h2_floats(int B, int Cin, int H, int W, int Cout, int ksize);
/* x_is_split is a flag word.  Bit 0: the activations were already split by their producer (i2r_groupnorm_fwd_f32 with
 * split_out): each 32-bit element holds f16 hi | f16 lo << 16 in the same NCHW positions.  Bit 1: the OUTPUT is the qkv
 * tensor of an attention block and is stored in that split form, q rows (channel % 3 == 0) pre-multiplied by the softmax
 * scale, for i2r_attention_fwd_h2 with presplit = 1; needs Cout % 3 == 0, and split-K is not used for it.
 * Bit 2 (3x3, W a power of two <= 32): x is (B, Cin, H/2, W/2) and is nearest-upsampled x2 while it is read (Conv2d
 * up=True, networks.py:79-81).  Bit 3: the residual is (B, Cout, H/2, W/2), upsampled the same way; no split-K.
 * Bit 4: reduced precision (optional, 1e-2 class): where a kernel is built for it (3x3, bit 0 set, W a power of two
 * <= 32) only the f16 high halves are multiplied -- one MFMA per tile-step instead of three; ignored elsewhere. */
int i2r_conv2d_fwd_h2(const float* x, const void* wp, const float* bias, const float* residual, float* out,
                      float* splitk_ws, long splitk_floats, int B, int Cin, int H, int W, int Cout, int ksize,
                      int x_is_split, i2r_stream_t stream);
/* Data gradient of the same convolution (autograd of conv2d w.r.t. x, as i2r_conv2d_dgrad_f32) on the f16x2-split
 * kernel: weights packed once per weight version by the _dgrad pack routine (byte count:
 * i2r_conv2d_packed_h2_bytes(Cin, Cout, ksize), the transposed problem), split-K workspace as for the forward of the
 * transposed problem (i2r_conv2d_splitk_h2_floats(B, Cout, H, W, Cin, ksize)).  `scale`: device float[2] {s, 1/s}
 * made by i2r_absmax_pow2_scale_f32 over dy -- gradients lie far below fp16's normal range; scaled by the power of
 * two s they are split at full precision and the result is scaled back (both exact). */
int i2r_conv2d_pack_weight_h2_dgrad(const float* w /* OIHW */, void* wp, int Cout, int Cin, int ksize,
                                    i2r_stream_t stream);
/* Both split-f16 packs for many layers in ONE launch (the training step packs every layer's weights once per step).
 * table: device int64 rows {w, wp, Cout, Cin, ksize, dgrad} -- wp as for i2r_conv2d_pack_weight_h2 (dgrad 0) or
 * i2r_conv2d_pack_weight_h2_dgrad (dgrad 1); blockmap: device int32 pairs {row, block}, a row needing
 * ceil(pairs / block_pairs) blocks: block_pairs from the query below, pairs = a quarter of the layout's packed_h2_bytes. */
int i2r_conv2d_pack_h2_block_pairs(void);
int i2r_conv2d_pack_h2_multi(const void* table, const void* blockmap, int nblocks, i2r_stream_t stream);
int i2r_conv2d_dgrad_h2(const float* dy, const void* wpd, float* dx, const float* scale, float* splitk_ws,
                        long splitk_floats, int B, int Cin, int H, int W, int Cout, int ksize, i2r_stream_t stream);
/* scale_out[0] = s = 2^(13 - floor(log2 max|x|)), scale_out[1] = 1/s (s = 1 when max|x| is 0 or not finite).
 * scratch: TWO uint32 (running maximum, arrival count), zero on entry, left zero: one launch -- the last workgroup to
 * arrive writes the pair.  (No counterpart in the reference: it serves the kernel above.) */
int i2r_absmax_pow2_scale_f32(const float* x, long n, unsigned* scratch, float* scale_out, i2r_stream_t stream);
/* i2r_conv2d_fwd_h2 for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that may
 * not call hipMalloc or hand over a workspace -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid allocation.  Same arithmetic, same kernels, same split-K plan and therefore the same bits as
 * i2r_conv2d_fwd_h2 given the workspace i2r_conv2d_splitk_h2_floats asks for, but the split-K slabs live in a
 * file-scope device array whose float count the _max_floats query below returns (16 MiB, one copy per GPU the library runs
 * on, held until the process ends).  A shape whose plan wants more is rejected with an error (use the workspace
 * variant); shapes that need no split-K use no scratch at all.  The function calls no hipMalloc / hipFree or any
 * allocating routine.  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two
 * threads) race; serialise them.  Nothing in the array has to be cleared between calls.  Make the FIRST call on each
 * device outside stream capture (it looks the array's device address up once per device).  The array used is the one
 * of the CURRENT device (hipGetDevice): call with the device that owns x / out / the stream current, as for every
 * other entry point of this library. */
long i2r_conv2d_fwd_h2_noalloc_max_floats(void);
int i2r_conv2d_fwd_h2_noalloc(const float* x, const void* wp, const float* bias, const float* residual, float* out,
                              int B, int Cin, int H, int W, int Cout, int ksize, int x_is_split, i2r_stream_t stream);
/* 1x1 convolution over torch.cat([x1, x2], dim=1) (:458 feeding a block's `skip` conv, :156-159) without materialising
 * the concatenation; C1 a multiple of 16; weights packed for Cin = C1 + C2 by i2r_conv2d_pack_weight_h2. */
int i2r_conv1x1_cat_fwd_h2(const float* x1, const float* x2, int C1, int C2, const void* wp, const float* bias,
                           const float* residual, float* out, float* splitk_ws, long splitk_floats, int B, int H, int W,
                           int Cout, i2r_stream_t stream);
/* autograd w.r.t. w and bias: dw (Cout,Cin,k,k) and dbias (Cout, may be NULL) are overwritten, or added to when
 * `accumulate` is set (gradient buffers zeroed once per step by the caller). */
int i2r_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int H, int W,
                         int Cout, int ksize, int accumulate, i2r_stream_t stream);
/* the same gradients with a run-to-run reproducible summation order (deterministic mode): every split of the position
 * range stores its partial dW / dbias into its own slab of `ws` with plain stores and one more launch adds the slabs in
 * split order -- no atomics.  ws_floats >= i2r_conv2d_wgrad_det_floats(...) of the same shape. */
long i2r_conv2d_wgrad_det_floats(int B, int Cin, int H, int W, int Cout, int ksize);
int i2r_conv2d_wgrad_det_f32(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats,
                             int B, int Cin, int H, int W, int Cout, int ksize, int accumulate, i2r_stream_t stream);
/* resample_filter [1,1] (:64-65, :82-85): y = scale * 2x2 box sum (down; scale 0.25 forward) and
 * y = scale * nearest x2 (up; scale 1 forward); each is the other's adjoint with the scales swapped. */
int i2r_down2_f32(const float* x, float* y, long planes, int H, int W, float scale, i2r_stream_t stream);
int i2r_up2_f32(const float* x, float* y, long planes, int H, int W, float scale, i2r_stream_t stream);
/* torch.cat([x, skip], dim=1) of DhariwalUNet.forward (:458) and its adjoint. n1, n2 = C1*H*W, C2*H*W. */
int i2r_concat_channels_f32(const float* a, const float* c, float* out, int B, long n1, long n2, i2r_stream_t stream);
int i2r_split_channels_f32(const float* in, float* a, float* c, int B, long n1, long n2, int accumulate_c,
                           i2r_stream_t stream);

/* ---- GroupNorm family: models/networks.py:104-106 fused with its consumers (:168, :173, :177, :182, :460) ---------- */
/* y = dropout_p( act( gn(x; gamma, beta, G, eps) * (1 + scale) + shift ) );  ss = [scale | shift] (ss_batch, 2C) or
 * NULL; act: 0 identity, 1 SiLU; mean/rstd (B*G) optional outputs for backward; dropout mask = Philox(seed,
 * offset + (*rng_step << 20)); rng_step (device counter, see i2r_counter_add_u64) may be NULL.  split_out: store each
 * y as the pair f16 hi | f16 lo << 16 (hi = f16(y), lo = f16(y - hi)) for i2r_conv2d_fwd_h2(x_is_split). */
int i2r_groupnorm_fwd_f32(const float* x, const float* gamma, const float* beta, const float* ss, int ss_batch,
                          float* y, float* mean, float* rstd, int B, int C, int HW, int G, float eps, int act,
                          float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_step, int split_out,
                          i2r_stream_t stream);
/* forward over torch.cat([x1, x2], dim=1) read in place (decoder skip connections, :458 then norm0 :168).
 * H*W must be a power of two >= 4 and a group at most 4096 elements (all shipped configs). */
int i2r_groupnorm_cat_fwd_f32(const float* x1, const float* x2, int C1, int C2, const float* gamma, const float* beta,
                              const float* ss, int ss_batch, float* y, int B, int HW, int G, float eps, int act,
                              int split_out, i2r_stream_t stream);
/* backward: dx (B,C,HW); dgb_part (B,2C) per-sample [dgamma | dbeta] partials (reduce with i2r_reduce_rows_f32);
 * dss (B,2C) = [dscale | dshift] iff ss given; dx_add (B,C,HW) or NULL is added to dx (the gradient that reaches x
 * through the block's residual / skip bypass, :166-187: saves autograd's separate accumulation pass). */
int i2r_groupnorm_bwd_f32(const float* x, const float* gamma, const float* beta, const float* ss, int ss_batch,
                          const float* mean, const float* rstd, const float* dy, float* dx, float* dgb_part,
                          float* dss, const float* dx_add, int B, int C, int HW, int G, int act, float drop_p,
                          uint64_t seed, uint64_t offset, const uint64_t* rng_step, i2r_stream_t stream);
/* same backward with dgamma / dbeta ADDED by fp32 atomics into caller-zeroed (C) buffers (the training step's
 * gradient slots): no per-sample partials, no reduction launch; batch summation order is not fixed. */
int i2r_groupnorm_bwd_acc_f32(const float* x, const float* gamma, const float* beta, const float* ss, int ss_batch,
                              const float* mean, const float* rstd, const float* dy, float* dx, float* dgamma_acc,
                              float* dbeta_acc, float* dss, const float* dx_add, int B, int C, int HW, int G, int act,
                              float drop_p, uint64_t seed, uint64_t offset, const uint64_t* rng_step,
                              i2r_stream_t stream);
int i2r_reduce_rows_f32(const float* in, float* out, int rows, int cols, i2r_stream_t stream);
/* same with the columns split over two outputs: [0,cols_a) -> out_a, [cols_a, cols_a+cols_b) -> out_b */
int i2r_reduce_rows2_f32(const float* in, float* out_a, float* out_b, int rows, int cols_a, int cols_b,
                         i2r_stream_t stream);

/* ---- attention: models/networks.py:113-126 (AttentionOp) + :182-184 ---------------------------------------------- */
/* qkv (B, 3*heads*64, T) interleaved (head, d, {q,k,v}); out (B, heads*64, T); lse (B*heads*T) optional (backward). */
int i2r_attention_fwd_f32(const float* qkv, float* out, float* lse, int B, int heads, int T, i2r_stream_t stream);
/* inference forward of the same op on the f16 matrix cores with fp32-grade accuracy (f16x2 split, see
 * i2r_conv2d_fwd_h2); T a multiple of 32 and > 32; qkv 16-byte aligned.  presplit: qkv holds split pairs with q
 * pre-scaled (i2r_conv2d_fwd_h2 flag bit 1) -- the kernel then converts nothing. */
int i2r_attention_fwd_h2(const float* qkv, float* out, int B, int heads, int T, int presplit, i2r_stream_t stream);
/* The f16x2 kernel as the TRAINING forward of AttentionOp (networks.py:113-119): fp32 qkv in, and the per-query
 * log-sum-exp out (B*heads*T floats, natural-log units) from which i2r_attention_bwd_f32 recomputes the probabilities. */
int i2r_attention_fwd_h2_lse(const float* qkv, float* out, float* lse, int B, int heads, int T, i2r_stream_t stream);
/* AttentionOp.backward (:121-126) + einsum backward: dqkv (B,3C,T); delta_ws: workspace of B*heads*T floats. */
int i2r_attention_bwd_f32(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                          float* delta_ws, int B, int heads, int T, i2r_stream_t stream);

/* The same backward on the f16x2-split matrix-core path (models/networks.py:121-126, :184 like i2r_attention_bwd_f32):
 * every product as three f16 MFMAs with fp32 accumulation, dout scaled into fp16's range by `scale` = device float[2]
 * {s, 1/s} from i2r_absmax_pow2_scale_f32 over dout and scaled back exactly.  T a multiple of 64; qkv and dout
 * 16-byte aligned.  A value that leaves fp16's range comes out non-finite, never finite and wrong. */
int i2r_attention_bwd_h2(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                         float* delta_ws, const float* scale, int B, int heads, int T, i2r_stream_t stream);

/* ---- embedding path: models/networks.py:200-206 (PositionalEmbedding), :39-43 (Linear) ---------------------------- */
int i2r_pos_embedding_f32(const float* x, float* out, int B, int num_channels, int max_positions,
                          i2r_stream_t stream);
/* y = act(x W^T + b); x (M,K), w (N,K); pre (M,N) optional pre-activation copy; act_silu fuses silu (:440,:447). */
int i2r_linear_fwd_f32(const float* x, const float* w, const float* bias, float* y, float* pre, int M, int K, int N,
                       int act_silu, i2r_stream_t stream);
/* autograd of the above: dx (M,K) = dy W, dw (N,K) = dy^T x, db (N) = column sums of dy; each output may be NULL
 * (db needs dw). */
int i2r_linear_bwd_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int M, int K,
                       int N, i2r_stream_t stream);
/* Every `affine` Linear of the UNet (models/networks.py:170; all read the same embedding) in one launch per product.
 * table: int64 rows {W, bias, dW, db, N, float offset of the layer in the slab}; blockmap: int32 {layer, tile x,
 * tile y, partial offset}.  mode 0: y_l = emb W_l^T + b_l -> base + off_l;  mode 1: dW_l = dy_l^T emb, db_l = column
 * sums (dy_l at base + off_l);  mode 2: partial dx_l = dy_l W_l -> base + blockmap.w (sum them with
 * i2r_reduce_rows_f32).  emb is (M, K). */
int i2r_gemm_grouped_f32(const void* table, const void* blockmap, int nblocks, int mode, const float* emb,
                         float* base, int M, int K, i2r_stream_t stream);
int i2r_silu_bwd_f32(const float* pre, const float* dy, float* dx, long n, i2r_stream_t stream);
int i2r_add_f32(const float* a, const float* b, float* y, long n, i2r_stream_t stream);

/* ---- EDM preconditioning: models/networks.py:654-668 --------------------------------------------------------------- */
/* xin = c_in(sigma) * float(x), c_noise = ln(sigma)/4 (:662-665); x fp32 or fp64; sigma: nsig = 1 (broadcast) or B. */
int i2r_edm_scale_input(const void* x, int x_is_f64, const float* sigma, int nsig, float* xin, float* c_noise, int B,
                        long chw, float sigma_data, i2r_stream_t stream);
/* D = c_skip * float(x) + c_out * F (:660-661, :667) */
int i2r_edm_combine(const void* x, int x_is_f64, const float* F, const float* sigma, int nsig, float* D, int B,
                    long chw, float sigma_data, i2r_stream_t stream);
int i2r_edm_combine_bwd(const float* dD, const float* sigma, int nsig, float* dF, int B, long chw, float sigma_data,
                        i2r_stream_t stream);

/* ---- masked EDM loss: models/our.py:53-78 -------------------------------------------------------------------------- */
/* forward_irregular (:71-77): sigma = exp(rnd*P_std+P_mean), weight, xn = x + (noise*sigma)*mask.  rnd_in (B) and
 * noise_in (B*chw) inject the two random draws (parity tests); when NULL they are drawn in-kernel (Philox). */
int i2r_edm_noise(const float* x, const float* mask, const float* rnd_in, const float* noise_in, float* xn,
                  float* sigma, float* weight, int B, long chw, float P_mean, float P_std, float sigma_data,
                  uint64_t seed, uint64_t offset, const uint64_t* rng_step, i2r_stream_t stream);
/* *counter += inc on the stream: the device-resident step of the training RNG streams.  Kernels that take `rng_step`
 * add (*rng_step << 20) to their Philox offset, so a captured HIP graph that contains this increment draws fresh
 * dropout masks / noise on every replay.  (No counterpart in the reference, whose RNG is torch's global generator.) */
int i2r_counter_add_u64(uint64_t* counter, uint64_t inc, i2r_stream_t stream);
/* mask = 1 - isnan(x), x_clean = nan_to_num(x, 0): the mask=None branch of loss_fn_irregular (:60-62) */
int i2r_nan_mask_f32(const float* x, float* x_clean, float* mask, long n, i2r_stream_t stream);
/* loss_fn_irregular (:63-66): loss = mean(weight * (D*mask - x*mask)^2) over all B*chw elements; dD (optional) is
 * d loss / d D.  partials: workspace of n_partials (>= 1, ideally 1024) floats. */
int i2r_masked_edm_loss(const float* D, const float* x, const float* mask, const float* weight, float* loss,
                        float* dD, float* partials, int n_partials, int B, long chw, i2r_stream_t stream);

/* ---- Heun sampler updates: models/sampler.py:45, :56-57, :62-63 (fp64 state, fp32 denoiser output) ------------------ */
int i2r_heun_init_f64(const float* latents, double* x, long n, double t0, i2r_stream_t stream);
/* the same for latents the caller hands over in fp64: :45 is `latents.to(torch.float64) * t_steps[0]`, so fp64 latents
 * enter the state unrounded */
int i2r_heun_init_f64_from_f64(const double* latents, double* x, long n, double t0, i2r_stream_t stream);
int i2r_heun_euler_f64(const double* x_hat, const float* D, double* d_cur, double* x_next, long n, double t_hat,
                       double t_next, i2r_stream_t stream);
int i2r_heun_correct_f64(const double* x_hat, const double* x_next, const float* Dp, const double* d_cur,
                         double* x_out, long n, double t_hat, double t_next, i2r_stream_t stream);

/* ---- fused optimiser tail: run_irregular.py:228-230 (clip_grad_norm_ -> AdamW.step -> LitEma, models/ema.py:26-47) -- */
/* One multi-tensor step over every parameter.  table: device (T,6) int64 rows {param, grad, exp_avg, exp_avg_sq,
 * ema_shadow|0, numel}; chunks: device (nchunks,2) int32 {tensor, chunk index} with opt_chunk_elems (query below) elements per
 * chunk; partials: nchunks floats; scalars: 8 floats (out: [0] total grad norm, [1] clip coefficient);
 * step: device float (AdamW step count, incremented here); ema_num_updates: LitEma's int32 counter (incremented here)
 * or NULL for no EMA.  max_norm <= 0 disables clipping.  All per-step state is device-resident: graph-capturable.
 * With clipping on, a step whose total gradient norm is inf / NaN is skipped as a whole (nothing is written but
 * scalars[0], scalars[7] = 1 and the skipped-step count scalars[6]); the reference would write NaN into every weight. */
int i2r_opt_chunk_elems(void);
int i2r_fused_adamw_ema_step(const void* table, const void* chunks, int nchunks, float* partials, float* scalars,
                             float* step, int* ema_num_updates, float max_norm, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float ema_decay, int ema_warmup, i2r_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* I2R_HIP_H */
