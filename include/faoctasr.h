/*
 * faoctasr.h -- C ABI of the MI355X (gfx950) kernels behind the frequency-aware OCTA
 * super-resolution train step.
 *
 * The reference has no native code: its hot path (train.py:166-269) runs stock ATen
 * ops from Python.  These entry points are what a binding for that path would bind,
 * one per ATen op class the path executes.  Conventions:
 *   - every tensor is a raw DEVICE pointer to contiguous fp32 NCHW data owned by the caller;
 *   - `stream` is a hipStream_t passed as void*; kernels are only enqueued: no allocation,
 *     no synchronisation, no host read-back (safe under hipGraph capture);
 *   - workspaces are caller-provided (sizes from the *_workspace_floats queries);
 *   - return 0 on success, negative on error; faoctasr_last_error() gives the thread-local text.
 * Each function cites the reference call site(s) it stands in for (paths under /root/reference).
 */
#ifndef FAOCTASR_H
#define FAOCTASR_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* faoctasr_stream_t;

enum { FAOCTASR_ACT_NONE = 0, FAOCTASR_ACT_RELU = 1, FAOCTASR_ACT_LRELU = 2, FAOCTASR_ACT_TANH = 3 };
enum { FAOCTASR_OK = 0, FAOCTASR_EINVAL = -1, FAOCTASR_EUNSUPPORTED = -2, FAOCTASR_EHIP = -3 };

int faoctasr_version(void);
const char* faoctasr_last_error(void);
/* diagnostics (bench.py's per-family roofline): the kernel family the calling thread's last convolution-type call went to:
 * 1 flat implicit GEMM, 2 LDS-patch implicit GEMM, 3 Winograd F(2x2,3x3), 4 bf16x3 split, 5 M=1 head (VALU), 6 narrow-map GEMM,
 * 7 stem input gradient (1..4 input channels, VALU); 11 flat weight gradient, 12 LDS-patch weight gradient, 13 stride-1 / 4x4
 * stride-2 weight gradient (wgrad_s1), 14 M=1 weight gradient, 15 bf16x3 weight gradient (wgrad_x3), 16 stem weight gradient (VALU) */
int faoctasr_last_route(void);

/* ---- convolution family (implicit GEMM on f32 MFMA) --------------------------------------
 * nn.Conv2d forward: model.py:102,109,117,122 (discriminator), 242-244,250,258,275-277,286
 * (stems/skip), 412-414,438 (ResnetBlock, head), 451,458,473 (ResnetGenerator), 494,499.
 * y[N,M,OH,OW] = act(conv(x[N,C,IH,IW], w[M,C,KH,KW]) + bias); OH=(IH+2*pad-KH)/stride+1.
 * reflect!=0 folds nn.ReflectionPad2d(pad) (model.py:450,472) into the gather.          */
int faoctasr_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                        int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                        int reflect, int act, float slope, float* wpack, int wpack_state, int precision,
                        faoctasr_stream_t stream);
/* Packed-weight images for the LDS-patch kernel.  Every gather-type call (conv2d_fwd/dgrad,
 * conv_transpose2d_fwd/dgrad) takes an optional caller-owned buffer `wpack` of
 * faoctasr_conv_wpack_floats(kind, C, M, KH, KW, stride, pad) floats (kind 0..3 in that order; C, M
 * as in the matching call) and `wpack_state`: 0 = none (flat im2col kernel), 1 = pack `w` into it
 * now, 2 = it already holds this `w` (valid until the weights change).
 * `precision`: 0 = exact fp32 on v_mfma_f32_32x32x2_f32 (Winograd F(2x2,3x3) on the dense stride-1 3x3 layers), 1 = the same
 * without Winograd, 3 = "f16x2" (below: faoctasr_conv_set_scales); 2 = "bf16x3": operands split hi/lo into bf16, three
 * v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate (fp32-parity, ~5x the MFMA rate; needs a wpack buffer and
 * C >= 16, width >= 24 -- other shapes silently use the fp32 kernels).  The packed image depends on `precision` and on
 * whether the map is wide enough for the split kernel, so keep one buffer per (weights, precision, input size).
 * OR-ing FAOCTASR_CONV_NO_SPLIT_K into `precision` of a gather call keeps the whole reduction of an output element in one
 * block: no fp32 atomics, a bit-reproducible result, at the price of under-filled grids on narrow maps (the kernels otherwise
 * split K across blocks when a layer has fewer blocks than the chip has CUs).                                          */
#define FAOCTASR_CONV_NO_SPLIT_K 0x100
long faoctasr_conv_wpack_floats(int kind, int C, int M, int KH, int KW, int stride, int pad, int precision);
/* Batched packing: all packed-weight images of a step in ONE launch (csrc/conv_pack.hip).  Each convolution call of
 * train.py:166-269 re-reads weights the optimizer has just changed (train.py:239,268), i.e. ~240 images per step;
 * packed one by one they are ~240 tiny dependent launches.  faoctasr_conv_pack_job writes, into a HOST slot of
 * FAOCTASR_PACK_JOB_BYTES, the job that the gather call `kind` (0 conv2d_fwd, 1 conv2d_dgrad, 2 conv_transpose2d_fwd,
 * 3 conv_transpose2d_dgrad) with exactly these arguments and wpack_state 1 would launch; it returns the job's block
 * count (0: that call uses no packed image -- skip the slot; < 0: error).  `block_base` is the sum of the block counts
 * of the jobs before it.  The caller copies the slots, contiguous, to device memory once and calls
 * faoctasr_conv_pack_run(table, njobs, total_blocks, stream) after every weight update; the matching gather calls then
 * pass wpack_state 2.  `w` / `wpack` addresses are baked into the job.                                              */
#define FAOCTASR_PACK_JOB_BYTES 1024
long faoctasr_conv_pack_job(void* job_host, long block_base, int kind, const float* w, float* wpack,
                            int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                            int reflect, int out_pad, int precision);
int faoctasr_conv_pack_run(const void* jobs_dev, int njobs, long nblocks, faoctasr_stream_t stream);
/* precision 3 ("f16x2") tables only, BEFORE faoctasr_conv_pack_run on the same stream: the absmax slots of every job's weight
 * tensor (8 blocks per job, one partial maximum each, in the last 8 words of the job's image).                          */
int faoctasr_conv_pack_scales(const void* jobs_dev, int njobs, faoctasr_stream_t stream);
/* ---- precision 3, "f16x2": fp32-exact-class contraction at the 16-bit matrix-core rate ------------------------------
 * Every fp32 operand is split into hi = f16(x s), lo = f16(x s - hi) (22 significant bits) and a product is accumulated in
 * fp32 as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16; s is the power of two that puts the operand tensor's largest
 * magnitude in [2^14, 2^15) and is divided out of the accumulators.  Measured against fp64 the result is at or below the
 * error of the exact-f32 MFMA kernels (csrc/split16.h, profiles/r04_split_precision_error.log), at ~5x their rate.
 * The largest magnitude of an ACTIVATION operand travels as its fp32 bit pattern in a caller-owned "absmax slot" of
 * FAOCTASR_ABSMAX_SLOT_WORDS device words (512 bytes; 8 of them are used, one per 64-byte line, so that the producers' atomics
 * do not queue on one address; a reader takes their maximum): zero the slot, then faoctasr_absmax_bits(x, n, slot) (atomicMax:
 * several calls may fold several tensors into one slot).  The slots of the next convolution-type call of the calling thread are handed over with
 * faoctasr_conv_set_scales(a, b): gather calls (conv2d_fwd / dgrad, conv_transpose2d_fwd / dgrad) read `a` = the slot of their
 * gathered tensor (x, resp. dy); weight-gradient calls read `a` = slot of x and `b` = slot of dy.  The call consumes them
 * (a precision-3 call without slots fails with FAOCTASR_EINVAL); the weights' own slot lives in the packed image.  Shapes
 * the split kernels do not take (maps narrower than 24, fewer than 16 channels) silently run on the exact-f32 kernels. */
#define FAOCTASR_ABSMAX_SLOT_WORDS 128
int faoctasr_absmax_bits(const float* x, long n, unsigned* slot, faoctasr_stream_t stream);
/* The producer's side of the same slot: the NEXT faoctasr_batchnorm_train_fwd (its y) / faoctasr_batchnorm_train_bwd (its dx) /
 * faoctasr_cat2_act_fwd (its y) call of the calling thread folds the largest magnitude of its output into `slot` (zeroed by the caller) in its own store loop, so
 * the convolution that reads that tensor needs no separate faoctasr_absmax_bits pass over it.  Map sizes (H*W) that are not a
 * multiple of 4 are refused (FAOCTASR_EUNSUPPORTED): use faoctasr_absmax_bits there.                                       */
int faoctasr_out_absmax(unsigned* slot);
int faoctasr_conv_set_scales(const unsigned* slot_a, const unsigned* slot_b);
/* Which slots the precision-3 form of a call reads: bit 0 = slot a, bit 1 = slot b, 0 = none (the shape runs on an exact-f32
 * kernel).  kind 0..3 as faoctasr_conv_pack_job (conv2d_fwd, conv2d_dgrad, conv_transpose2d_fwd, conv_transpose2d_dgrad),
 * 4 = conv2d_wgrad, 5 = conv_transpose2d_wgrad; the other arguments exactly as that call receives them.                      */
int faoctasr_conv_needs_scales(int kind, int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                               int reflect, int out_pad);
/* Two-pass weight-gradient reduction (precisions 2 and 3, the shapes the split kernels take): with a caller-owned, 16-byte aligned
 * workspace of faoctasr_conv_wgrad_workspace_floats(C, M, KH, KW, stride) floats handed over for the NEXT weight-gradient call of the
 * calling thread, every pixel range of the kernel stores its partial dW there (plain stores) and a second kernel adds them to dw
 * in a fixed order -- instead of one fp32 atomic per partial and element.  Faster (the atomics were 29 us of a 67 us launch on the
 * 256 -> 256 3x3 layer) and, with a step's weight gradients on one stream, bit-reproducible.  Without a workspace (or one that is
 * too small) the call accumulates with atomics as before.  The workspace is only used between the call's two launches: calls on
 * the same stream may share it.                                                                                            */
int faoctasr_conv_set_workspace(float* workspace, long nfloats);
/* y = gather(...) + residual for the NEXT gather call (conv2d_fwd / dgrad, conv_transpose2d_fwd / dgrad) of the calling thread;
 * `residual` has the output's shape.  x + conv_block(x) (model.py:420,505) sends two gradients to x -- the skip's and the first
 * convolution's input gradient -- and autograd adds them with an elementwise kernel (66 per train step); handing the skip's gradient
 * to that input-gradient call adds it in the split kernels' epilogue instead (other routes: one in-place pass after the kernel). */
int faoctasr_conv_set_residual(const float* residual);
long faoctasr_conv_wgrad_workspace_floats(int C, int M, int KH, int KW, int stride);
/* aten::convolution_backward, input gradient.  dx[N,C,IH,IW] from dy[N,M,OH,OW].  With
 * reflect!=0 dx is the gradient w.r.t. the PADDED input [N,C,IH+2p,IW+2p] (fold it with
 * faoctasr_reflect_pad_bwd).                                                              */
int faoctasr_conv2d_dgrad(const float* dy, const float* w, float* dx,
                          int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                          float* wpack, int wpack_state, int precision, faoctasr_stream_t stream);
/* aten::convolution_backward, weight gradient.  dw[M,C,KH,KW] is overwritten, or added to
 * when accumulate != 0 (the gradient arena is zeroed once per step instead).  precision as above: 2 = bf16x3 (hi/lo-split dY
 * and X on v_mfma_f32_32x32x16_bf16) for the stride-1 3x3 (pad 1) and 7x7 (pad 3, reflection or zero padding) layers and the
 * stride-2 3x3 / 4x4 (pad 1) layers, with C, M multiples of 64, an output width that is a multiple of 32 and an even output
 * height (the transposed convolution's weight gradient takes the stride-2 form with x and dy swapped); other shapes silently
 * use the fp32 kernels.  faoctasr_last_route() tells which (15 = bf16x3).                    */
int faoctasr_conv2d_wgrad(const float* x, const float* dy, float* dw,
                          int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                          int reflect, int accumulate, int precision, faoctasr_stream_t stream);
/* nn.ConvTranspose2d forward: model.py:431 (4x4 s2 p1), 469 (3x3 s2 p1 output_padding 1).
 * x[N,C,IH,IW], w[C,M,KH,KW], y[N,M,OH,OW], OH=(IH-1)*stride-2*pad+KH+out_pad.          */
int faoctasr_conv_transpose2d_fwd(const float* x, const float* w, const float* bias, float* y,
                                  int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                                  int out_pad, int act, float slope, float* wpack, int wpack_state, int precision,
                                  faoctasr_stream_t stream);
int faoctasr_conv_transpose2d_dgrad(const float* dy, const float* w, float* dx,
                                    int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                                    int out_pad, float* wpack, int wpack_state, int precision, faoctasr_stream_t stream);
int faoctasr_conv_transpose2d_wgrad(const float* x, const float* dy, float* dw,
                                    int N, int C, int IH, int IW, int M, int KH, int KW, int stride, int pad,
                                    int out_pad, int accumulate, int precision, faoctasr_stream_t stream);
/* gradient of nn.ReflectionPad2d(p): dx[NC,H,W] += fold of dxp[NC,H+2p,W+2p] (dx overwritten) */
int faoctasr_reflect_pad_bwd(const float* dxp, float* dx, int NC, int H, int W, int p, faoctasr_stream_t stream);
/* per-channel sum over (N,HW): conv bias gradient.  db[C] overwritten (or added to).      */
int faoctasr_channel_sum(const float* dy, float* db, int N, int C, int HW, int accumulate, faoctasr_stream_t stream);

/* ---- BatchNorm2d (training mode) + fused activation / residual ----------------------------
 * nn.BatchNorm2d calls: model.py:110,118 (D), 244-245,251 (stems/skip), 412-414,431 (shallowNet),
 * 452,459,470 (ResnetGenerator), 494,499 (ResidualBlock).  Batch statistics over (N,H,W),
 * biased variance for normalisation, unbiased for running_var, momentum/eps as given.
 * y = act(gamma*xhat + beta + residual)   (residual may be NULL).
 * workspace: faoctasr_bn_workspace_floats(C) floats.                                       */
long faoctasr_bn_workspace_floats(int C);
int faoctasr_batchnorm_train_fwd(const float* x, const float* gamma, const float* beta, const float* residual,
                                 float* y, float* save_mean, float* save_invstd,
                                 float* running_mean, float* running_var,
                                 int N, int C, int HW, float eps, float momentum, int act, float slope,
                                 float* workspace, faoctasr_stream_t stream);
/* dx overwritten; dgamma[C], dbeta[C] overwritten or (accumulate_affine != 0) added to; y is
 * the saved forward output (activation mask); the residual gradient dy*act'(y) is written to
 * dres when dres != NULL.  y may be NULL for act NONE / RELU / LRELU when the forward had NO residual: the mask is then taken
 * from the recomputed pre-activation x*gamma*invstd + (beta - mean*gamma*invstd) (the forward's own expression), which saves
 * one tensor read per pass; `beta` is only read in that case.                                */
int faoctasr_batchnorm_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd,
                                 float* dx, float* dgamma, float* dbeta, float* dres,
                                 int N, int C, int HW, int act, float slope, int accumulate_affine,
                                 float* workspace, faoctasr_stream_t stream);
/* nn.BatchNorm2d in eval mode (running statistics): the inference path, utils.py:186,221 `model.eval()`.
 * y = act(gamma*(x-running_mean)/sqrt(running_var+eps)+beta); the backward gives dx only (statistics are constants). */
int faoctasr_batchnorm_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float* y, int N, int C, int HW, float eps, int act, float slope,
                                faoctasr_stream_t stream);
int faoctasr_batchnorm_eval_bwd(const float* dy, const float* y, const float* gamma, const float* running_var, float* dx,
                                int N, int C, int HW, float eps, int act, float slope, faoctasr_stream_t stream);
/* InstanceNorm2d (named by north_star; not on the reference's path): per-(n,c) statistics;
 * save_mean/save_invstd have N*C entries, workspace faoctasr_bn_workspace_floats(N*C) floats. */
int faoctasr_instancenorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                              float* save_mean, float* save_invstd, int N, int C, int HW, float eps,
                              int act, float slope, float* workspace, faoctasr_stream_t stream);
int faoctasr_instancenorm_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* beta,
                              const float* save_mean, const float* save_invstd, float* dx,
                              float* dgamma, float* dbeta, int N, int C, int HW, int act, float slope,
                              float* workspace, faoctasr_stream_t stream);

/* ---- pointwise / data movement ----------------------------------------------------------- */
/* nn.ReLU / nn.LeakyReLU(0.2) / nn.Tanh (model.py:102,111,119,243,249,254,413,431,438,...) */
int faoctasr_act_fwd(const float* x, float* y, long n, int act, float slope, faoctasr_stream_t stream);
/* dx = dy * act'(y) with y the activation OUTPUT */
int faoctasr_act_bwd(const float* dy, const float* y, float* dx, long n, int act, float slope, faoctasr_stream_t stream);
/* torch.cat([a,b],1) followed by an optional activation (model.py:266,268,298 + 249,431) */
int faoctasr_cat2_act_fwd(const float* a, const float* b, float* y, int N, int Ca, int Cb, int HW,
                          int act, float slope, faoctasr_stream_t stream);
int faoctasr_cat2_act_bwd(const float* dy, const float* y, float* da, float* db, int N, int Ca, int Cb, int HW,
                          int act, float slope, faoctasr_stream_t stream);
/* y = alpha*a + beta*b (residual add x + conv_block(x): model.py:420,505) */
int faoctasr_axpby(const float* a, const float* b, float* y, long n, float alpha, float beta, faoctasr_stream_t stream);

/* ---- Haar DWT / IDWT, mode 'reflect', even H and W ----------------------------------------
 * DWTForward.forward transform2d.py:44-74 -> AFB2D lowlevel.py:312-365; DWTInverse.forward
 * transform2d.py:111-148 -> SFB2D lowlevel.py:647-694.  One level per call.
 * x[NC,H,W] -> ll[NC,H/2,W/2], hi[NC,3,H/2,W/2] (band order LH,HL,HH).                     */
int faoctasr_haar_dwt2d_fwd(const float* x, float* ll, float* hi, long NC, int H, int W, faoctasr_stream_t stream);
/* AFB2D.backward lowlevel.py:349-365 (= synthesis): dx[NC,H,W] from dll, dhi (either may be NULL = zeros) */
int faoctasr_haar_dwt2d_bwd(const float* dll, const float* dhi, float* dx, long NC, int H, int W, faoctasr_stream_t stream);
/* fused discriminator front ends: model.py:166-179 (D_A: LL only) and model.py:222-235
 * (D_B: cat(LH,HL,HH)*0.5+0.5, x[N,1,H,W] -> y[N,3,H/2,W/2]); mode 0 = LL, 1 = cat-normalised */
int faoctasr_haar_dfront_fwd(const float* x, float* y, int N, int H, int W, int mode, faoctasr_stream_t stream);
int faoctasr_haar_dfront_bwd(const float* dy, float* dx, int N, int H, int W, int mode, faoctasr_stream_t stream);

/* ---- FFT Gaussian frequency split as circulant GEMMs ---------------------------------------
 * utils.high_pass / utils.low_pass (utils.py:71-117) as called at train.py:173-175,189-191,
 * 197-199,211-213.  The shifted Gaussian mask is separable, so ifft2(mask*fft2(x)) = Ch x Cw^T
 * with real symmetric circulant matrices built once per (n, radius) (faoctasr_circulant_lowpass).
 * Batched row-major SGEMM on f32 MFMA: for b in [0,batch): C_b = A_b(MxK) * B_b(KxN).         */
int faoctasr_sgemm_batched(const float* A, const float* B, float* C, int M, int N, int K,
                           int lda, int ldb, int ldc, long strideA, long strideB, long strideC, int batch,
                           faoctasr_stream_t stream);
/* The filter cache entry for one (n, radius): out[n*n] = the real symmetric circulant of the centred Gaussian mask of
 * utils.py:71-80 (guais_low_pass; the high-pass mask is 1 - it), built in double precision on the device.  SURVEY 8b asks
 * for an immutable (H,W,r)-keyed cache behind a create/destroy handle; because this ABI never allocates, the entry is a
 * caller-owned buffer instead: build it once per (n, radius, device) with this call and keep it as long as needed.    */
int faoctasr_circulant_lowpass(float* out, int n, float radius, faoctasr_stream_t stream);
/* hf = (|x - low_hp| + x)/2, lf = -|low_lp|  (train.py:173-175) */
int faoctasr_freq_mix_fwd(const float* x, const float* low_hp, const float* low_lp, float* hf, float* lf,
                          long n, faoctasr_stream_t stream);
/* s_hp = 0.5*g_hf*sign(x-low_hp), s_lp = -g_lf*sign(low_lp), dx_direct = 0.5*g_hf + s_hp
 * (the caller then subtracts Ch s_hp Cw and adds Ch s_lp Cw)                                */
int faoctasr_freq_mix_bwd(const float* x, const float* low_hp, const float* low_lp, const float* g_hf, const float* g_lf,
                          float* s_hp, float* s_lp, float* dx_direct, long n, faoctasr_stream_t stream);

/* ---- SSIM (ssim.py:17-37), fused separable 11-tap Gaussian window, zero padding -------------
 * per-image sums of the ssim map are written to sums[N] (mean = sums/(C*H*W)).              */
int faoctasr_ssim_fwd(const float* a, const float* b, float* sums, int N, int C, int H, int W, faoctasr_stream_t stream);
/* da, db (either may be NULL) = g[n or 0] * d(sum of map)/d(a|b); gscale multiplies, g is a device scalar
 * array of length gN (1 = shared) */
int faoctasr_ssim_bwd(const float* a, const float* b, const float* g, int gN, float gscale, float* da, float* db,
                      int N, int C, int H, int W, faoctasr_stream_t stream);

/* ---- losses (train.py:91-99) -----------------------------------------------------------------
 * kind 0: sum (a-b)^2 (MSELoss), 1: sum |a-b| (L1Loss), 2: BCEWithLogits(input=a, target=b) sum.
 * out[0] = scale * sum (overwritten); workspace: faoctasr_loss_workspace_floats() floats.  */
long faoctasr_loss_workspace_floats(void);
int faoctasr_loss_fwd(const float* a, const float* b, float* out, long n, int kind, float scale, float* workspace,
                      faoctasr_stream_t stream);
/* gradient wrt `wrt` (0 = a, 1 = b): d = g[0]*scale * dloss/d(...) ; g is a device scalar */
int faoctasr_loss_bwd(const float* a, const float* b, const float* g, float* d, long n, int kind, float scale, int wrt,
                      faoctasr_stream_t stream);
/* discriminator head model.py:158-164: out[n] = wa*mean(a[n,:]) + wb*mean(b[n,:]) */
int faoctasr_mean_mix_fwd(const float* a, const float* b, float* out, int N, int La, int Lb, float wa, float wb,
                          faoctasr_stream_t stream);
int faoctasr_mean_mix_bwd(const float* g, float* da, float* db, int N, int La, int Lb, float wa, float wb,
                          faoctasr_stream_t stream);

/* ---- optimizer: torch.optim.AdamW step (train.py:102-103,239,269) over one flat arena -------- */
int faoctasr_adamw_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, float grad_scale, faoctasr_stream_t stream);
/* The same update with the scalars in DEVICE memory (for a hipGraph-captured step, SURVEY 8f-1: the graph replays with each
 * step's own values): hyper[8] = {lr, beta1, beta2, eps, weight_decay, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), grad_scale}. */
int faoctasr_adamw_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, faoctasr_stream_t stream);

/* ---- input pipeline: the tensor transforms of train.py:129-140 fused into one pass ---------------------------
 * img [N,H,W] uint8 grayscale (what Image.open(..).convert('L') + ToTensor hold before the /255), tops/lefts [N] device ints =
 * the RandomCrop offsets (drawn by the caller), out [N,1,out_size,out_size] fp32:
 *   out = (resize(crop(img)/255) - mean) / std,  resize = torch bicubic (align_corners false, A = -0.75) when out_size != crop
 * transforms_A = crop 128 -> 256 bicubic; transforms_B = crop 256, no resize (the Normalize/RandomCrop order there commutes). */
int faoctasr_prep_crop_resize(const unsigned char* img, const int* tops, const int* lefts, float* out, int N, int H, int W,
                              int crop, int out_size, float mean, float std, faoctasr_stream_t stream);

/* ---- data-parallel gradient exchange over RCCL ---------------------------------------------------------
 * The reference has no distributed code; a DDP wrap of its loop would all-reduce after loss_G.backward() (train.py:238) and
 * after the discriminator backwards (train.py:255,267).  Here that is ONE in-place SUM all-reduce per flat gradient arena,
 * enqueued on the caller's stream (capturable in a hipGraph); the 1/world average is folded into adamw_step's grad_scale.
 * librccl is bound at run time (dlopen; the copy already in the process when there is one) -- single-GPU hosts never load it.
 * comm handles: rank 0 fills a 128-byte id (ncclUniqueId) and ships it to the other ranks by any host channel; every rank
 * then calls comm_create with the device it will use current.  These three calls are the only ones that create state.  */
int faoctasr_comm_unique_id(void* id128);
int faoctasr_comm_create(void** comm, int nranks, int rank, const void* id128);
int faoctasr_comm_destroy(void* comm);
int faoctasr_comm_size(void* comm);                 /* number of ranks (> 0) or a negative error */
/* dtype: 0 = fp32 (the only arena type) */
int faoctasr_grad_allreduce(float* bucket, long count, int dtype, void* comm, faoctasr_stream_t stream);
/* identical replicas at start: rank `root`'s parameter arena / BatchNorm buffers to everyone, in place */
int faoctasr_param_broadcast(float* buf, long count, int root, void* comm, faoctasr_stream_t stream);

/* ---- evaluation path: utils.py:182-242 `eval` / `eval_6m` ------------------------------------------------------
 * The four skimage metrics of utils.py:209-212 on device images: y, gt [N][H][W] fp32; out [N][4] doubles = {PSNR
 * (peak_signal_noise_ratio, data_range), SSIM (structural_similarity defaults: 7x7 uniform window, sample covariance, map cropped
 * by 3), MSE, NMI (normalized_mutual_information: joint bins x bins histogram over each image's [min, max], numpy.histogram2d
 * bin semantics)}.  data_range = 2, bins = 100 reproduce the reference.  workspace: faoctasr_eval_workspace_bytes(N, bins) bytes. */
long faoctasr_eval_workspace_bytes(int N, int bins);
int faoctasr_eval_metrics(const float* y, const float* gt, double* out, void* workspace, int N, int H, int W, float data_range, int bins,
                          faoctasr_stream_t stream);
/* `model.eval()` (utils.py:186,221) makes every BatchNorm2d a per-channel affine map; folded into the preceding convolution:
 * w_folded = w * gamma / sqrt(running_var + eps) per output channel m, bias_folded = (bias - running_mean) * that + beta.
 * w is [M][K] (Conv2d; transposed = 0) or [K0][M][K] (ConvTranspose2d weight [C][M][kh*kw]; transposed = 1).            */
int faoctasr_bn_fold(const float* w, const float* bias, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, float* w_folded, float* bias_folded, int M, long K, int transposed, long K0,
                     faoctasr_stream_t stream);

/* ---- utility ---------------------------------------------------------------------------------- */
int faoctasr_fill(float* p, long n, float value, faoctasr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FAOCTASR_H */
