/*
 * stem_hip.h -- C ABI of libstem_hip.so: hand-written HIP kernels (gfx950 / MI355X)
 * for the STEM hot path of mmSir/SpatioTemporalEntropyModel.
 *
 * The reference has no device FFI: its "operator interface" for this path is the
 * set of torch ops its modules call.  Each entry point below
 * names the reference call site it replaces (paths relative to /root/reference).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors);
 *    outputs are pre-allocated by the caller; nothing here allocates or syncs;
 *  - kernels are asynchronous on `stream` (a hipStream_t passed as void*);
 *  - return 0 on success, <0 on error; stem_last_error() describes the last
 *    failure on the calling thread;
 *  - activations are NHWC fp32 ("channels_last"): element (b,y,x,c) of a view
 *    lives at p[((b*H + y)*W + x)*ld + c]; `ld` >= C lets a tensor be a channel
 *    slice of a wider buffer, which is how torch.cat / chunk on the channel axis
 *    (spatiotemporalpriors.py:846,858-859) are done without a copy;
 *  - convolution weights are consumed in a packed K-contiguous layout made by
 *    stem_pack_weight() from the reference's OIHW / IOHW tensors.
 */
#ifndef STEM_HIP_H
#define STEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *stem_last_error(void);
int stem_abi_version(void);

/* Plan selectors for tests and sweep tools: force a tile shape / split factor that the library would otherwise choose
 * ("fx3_tile": 0 automatic | 64 | 128 pixel workgroups of stem_conv2d_f16x3_fwd; "fx3_split" / "wg3_split": split factor
 * of stem_conv2d_f16x3_gen_fwd / stem_conv2d_wgrad_f16x3, 0 = the planner's; "fx3_depth": 0 automatic | 2 | 4 chunks in
 * flight in stem_conv2d_f16x3_fwd; "fx3_gen_tile": 0 automatic | 64 | 128 pixel workgroups of stem_conv2d_f16x3_gen_fwd; "arp_workers": workgroups of the persistent decoder).  Every setting computes the same
 * contraction (summation order aside); process-wide; not for concurrent use with launches.  There is no reference
 * counterpart (torch picks its kernels internally).  Nothing in the library reads the environment to change results:
 * ablated / instrumented variants exist only in builds with -DSTEM_EXPERIMENTS, which
 * stem_built_with_experiments() reports (0 for the shipped library).                                                  */
/* ---- launch tape: the native executor of a static launch schedule (csrc/tape.hip; reference loop body stem/trainSTEM.py:194-218,
 * which the reference walks through the Python interpreter and autograd every step).  A tape holds calls of THIS library's
 * int-returning entry points (function address + integer-class / float arguments; integer arguments may advance by a fixed
 * amount per replay: Adam's step count, Philox offsets), stream hand-overs and event records / waits; stem_tape_replay re-issues
 * a range of it on the streams it was recorded on. */
void *stem_tape_create(void);
void stem_tape_destroy(void *tape);
int stem_tape_length(void *tape);
int stem_tape_add_call(void *tape, void *fn, int nargs, const unsigned char *kinds, const long long *ivals, const double *fvals,
                       const long long *ideltas);
int stem_tape_add_wait(void *tape, void *waiting_stream, void *signalling_stream);
int stem_tape_add_event(void *tape, void *event, void *stream, int wait);
int stem_tape_replay(void *tape, int lo, int hi, long long n);
int stem_tape_set_iarg(void *tape, int entry, int arg, long long value);
/* SSE-class argument `arg` of call `entry` <- the 64-bit register pattern (a double, or a float's bits in the low half): the
 * hyper-parameters a scheduler edits between steps (stem/trainSTEM.py:123,290) */
int stem_tape_set_farg(void *tape, int entry, int arg, double pattern);
/* 1 if fn is an int-returning entry point of this library whose prototype passed the trampolines' compile-time contract check */
int stem_tape_entry_recordable(void *fn);
int stem_copy_d2d(void *dst, const void *src, size_t nbytes, void *stream);
/* optimizer.zero_grad() (stem/trainSTEM.py:203) as a recordable library call */
int stem_zero_bytes(void *dst, size_t nbytes, void *stream);
/* stream flags (8 bytes of signal memory): a stream waits for work the host has not issued yet -- the compute stream of a
 * data-parallel rank behind the gradient all-reduces its helper thread issues (torch.distributed / DistributedDataParallel in
 * stem_roi/train_stem_roi.py; here distributed.OverlappedGradReducer).  wait_ge: proceed once *flag >= value (a step counter: monotonic);
 * write: *flag <- value when the stream gets there, or at once from the host with stream == (void*)-1 */
int stem_stream_flag_create(void **flag);
int stem_stream_flag_destroy(void *flag);
int stem_stream_flag_wait_ge(void *flag, unsigned value, void *stream);
int stem_stream_flag_write(void *flag, unsigned value, void *stream);
int stem_tuning_set(const char *name, int value);
int stem_tuning_get(const char *name);
int stem_built_with_experiments(void);

/* ---- weight packing ------------------------------------------------------
 * roles (what the packed copy will be multiplied with):                       */
enum {
    STEM_PACK_CONV_FWD = 0,     /* nn.Conv2d weight [K,C,R,S] -> [R*S][K][C]  (forward)            */
    STEM_PACK_CONV_DGRAD = 1,   /* nn.Conv2d weight [K,C,R,S] -> [R*S][C][K]  (input gradient)      */
    STEM_PACK_DECONV_FWD = 2,   /* nn.ConvTranspose2d weight [C,K,R,S] -> [R*S][K][C] (forward)      */
    STEM_PACK_DECONV_DGRAD = 3, /* nn.ConvTranspose2d weight [C,K,R,S] -> [R*S][C][K]                */
    STEM_PACK_CONV_FWD_C4 = 4   /* Conv2d weight [K,3or4,R,S] -> [K][32 taps][4] zero padded         */
};
/* `masked`: bits 0-1 = mode, bit 2 (value 4) = mask type B instead of A (compressai/layers/layers.py:39-47: type A zeroes
 * the taps at row > R/2 or (row == R/2, col >= S/2); type B keeps the centre tap).  Mode 1 applies the mask to the
 * packed copy; mode 2 additionally zeroes the masked taps of `w` itself, which is what the reference's forward does
 * (`self.weight.data *= self.mask`, layers.py:46).                                                                 */
int stem_pack_weight(const float *w, float *wp, int K, int C, int R, int S, int role, int masked, void *stream);
size_t stem_packed_weight_elems(int K, int C, int R, int S, int role);
/* gradient in packed layout [splits][R*S][K][C] (from stem_conv2d_wgrad) -> reference layout,
 * summing the split-K slabs in a fixed order.  flags: STEM_UNPACK_DECONV writes the ConvTranspose2d layout
 * [C,K,R,S]; STEM_UNPACK_ACCUMULATE adds to `dw` instead of overwriting it (what autograd's `.grad +=` does when
 * gradients of several backward passes accumulate: stem_roi/train_stem_roi.py:533,560).        */
enum { STEM_UNPACK_DECONV = 1, STEM_UNPACK_ACCUMULATE = 2 };
int stem_unpack_wgrad(const float *dwp, float *dw, int K, int C, int R, int S, int splits, int flags, void *stream);

/* Multi-tensor forms: every layer of a model in one launch (host arrays of descriptors, device pointers inside). */
typedef struct {
    const float *w;   /* reference-layout weight (masked == 2 also writes it, see above) */
    float *wp;        /* packed output */
    int K, C, R, S, role, masked;
} stem_pack_desc;
typedef struct {
    const float *dwp; /* [splits][R*S][..][..] slabs from stem_conv2d_wgrad / stem_deconv2d_wgrad */
    float *dw;        /* gradient in the reference layout */
    int K, C, R, S, splits;
    int flags;        /* STEM_UNPACK_DECONV | STEM_UNPACK_ACCUMULATE */
} stem_unpack_desc;
int stem_pack_weights_multi(const stem_pack_desc *descs, int n, void *stream);
int stem_unpack_wgrads_multi(const stem_unpack_desc *descs, int n, void *stream);
/* Second stage of the bias gradient (the `.bias.grad` autograd leaves for a Conv2d, stem/trainSTEM.py:212) for up to 24 layers
 * with one launch: db (+)= sum over `parts` rows of part[parts][K] -- the per-split column sums stem_conv2d_wgrad_f16x3 leaves. */
typedef struct {
    const float *part;
    float *db;
    int K, parts, accumulate, reserved;
} stem_bias_final_desc;
int stem_bias_grad_final_multi(const stem_bias_final_desc *descs, int n, void *stream);

/* ---- epilogues --------------------------------------------------------- */
enum { STEM_ACT_NONE = 0, STEM_ACT_LRELU = 1 };
/* OR into `act` of stem_conv2d_fwd (and into `kind` of stem_conv_workspace_bytes): the weight is a type-A MaskedConv2d
 * (compressai/layers/layers.py:39-42), whose taps at row > R/2 or (row == R/2, col >= S/2) are zero -- the kernel skips
 * them (12 of 25 for the 5x5 context model) instead of multiplying zeros. */
#define STEM_CONV_MASKED_A 0x100

/* Split-K workspace.  Layers whose output is too small to fill 256 CUs (the 16x16 / 8x8 / 4x4 STEM latents)
 * split their reduction over taps x channels across workgroups; the fp32 partial tiles go to `ws` and the LAST
 * workgroup to arrive at an output tile (integer arrival counter at the head of `ws`) sums them in split order and
 * applies bias / activation inside the same kernel: bit-reproducible (the order does not depend on who arrives
 * last, no float atomics), no second launch.
 * kind: 0 conv fwd, 1 conv dgrad, 2 deconv fwd, 3 deconv dgrad; dims are the LAYER's (B,H,W,C in / K out).
 * Passing ws = NULL (or too few bytes) is legal and runs the unsplit schedule.
 * CONTRACT (ABI version 2): the first 64 KiB of `ws` must be ZERO when the buffer is first handed to the library;
 * every launch leaves them zero again, so one buffer serves all layers launched on one stream.  Two launches that
 * may overlap in time (different streams) need different buffers.                                                 */
enum { STEM_KIND_CONV_FWD = 0, STEM_KIND_CONV_DGRAD = 1, STEM_KIND_DECONV_FWD = 2, STEM_KIND_DECONV_DGRAD = 3 };
size_t stem_conv_workspace_bytes(int kind, int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad);

/* nn.Conv2d forward (+bias, optional fused LeakyReLU).  Replaces F.conv2d under
 * compressai/models/utils.py:112-120 and spatiotemporalpriors.py:807-838.
 * x[B,H,W,C] (ldx)  wp = STEM_PACK_CONV_FWD  ->  y[B,Ho,Wo,K] (ldy)                             */
int stem_conv2d_fwd(const float *x, int ldx, const float *wp, const float *bias, float *y, int ldy,
                    int B, int H, int W, int C, int K, int R, int S, int stride, int pad,
                    int act, float slope, void *ws, size_t ws_bytes, void *stream);
/* first analysis layer (C_in = 3, priors.py:422): x is NHWC with 4 channels (4th zero) made by
 * stem_nchw3_to_nhwc4; wp = STEM_PACK_CONV_FWD_C4.                                               */
int stem_conv2d_fwd_c4(const float *x4, const float *wp, const float *bias, float *y, int ldy,
                       int B, int H, int W, int K, int R, int S, int stride, int pad, void *stream);
/* dX of nn.Conv2d.  dy[B,Ho,Wo,K] -> dx[B,H,W,C].  wp = STEM_PACK_CONV_DGRAD.  If `xact` != NULL the
 * result is multiplied by LeakyReLU'(xact) (xact = the layer's *input* activation = output of the
 * preceding LeakyReLU), fusing the activation backward.                                           */
int stem_conv2d_dgrad(const float *dy, int lddy, const float *wp, float *dx, int lddx,
                      const float *xact, int ldxact, float slope,
                      int B, int H, int W, int C, int K, int R, int S, int stride, int pad,
                      void *ws, size_t ws_bytes, void *stream);
/* dW (packed, `splits` slabs of [R*S][K][C]) and db[K] of nn.Conv2d.  All R*S taps are produced
 * (the reference's autograd does not mask MaskedConv2d's weight gradient).  dwp must hold
 * stem_wgrad_workspace_elems() floats: the slabs plus scratch for the two-stage bias-gradient sum.  */
#define STEM_WGRAD_SQUARE_G 2     /* flags: use x^2 instead of x (GDN gamma gradient) */
#define STEM_WGRAD_TABLE_VALID 1   /* flags: `dwp` still holds the gather table of an earlier call with the same geometry */
#define STEM_WGRAD_ACCUMULATE_DB 4 /* flags: db += column sums instead of db = (gradient accumulation over several backward passes) */
#define STEM_WGRAD_DEFER_DB 8      /* flags: leave the bias gradient's per-part column sums behind the slabs (dwp + splits * R * S * K * C) and skip
                                    * the second stage: the caller adds them into db later (stem_bias_grad_final / stem_bias_grad_final_multi
                                    * with stem_wgrad_bias_parts parts) -- one launch per module group instead of one per layer */
int stem_wgrad_bias_parts(const float *x, int ldx, const float *dy, int lddy, long npix_dy, int C, int K, int splits, int flags, int deconv);
int stem_conv2d_wgrad(const float *x, int ldx, const float *dy, int lddy, float *dwp, float *db,
                      int B, int H, int W, int C, int K, int R, int S, int stride, int pad,
                      int splits, int flags, void *stream);
int stem_wgrad_splits(int B, int Ho, int Wo, int C, int K, int R, int S);      /* Ho,Wo = the loop grid (Conv2d: output) */
size_t stem_wgrad_workspace_elems(int splits, int C, int K, int R, int S, int npix);   /* npix = B*Ho*Wo of the loop grid */

/* nn.ConvTranspose2d forward as sub-pixel phases (no zero insertion).  models/utils.py:122-130,
 * spatiotemporalpriors.py:821-826.  x[B,H,W,C] -> y[B,Ho,Wo,K], wp = STEM_PACK_DECONV_FWD.        */
int stem_deconv2d_fwd(const float *x, int ldx, const float *wp, const float *bias, float *y, int ldy,
                      int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad,
                      int act, float slope, void *ws, size_t ws_bytes, void *stream);
int stem_deconv2d_dgrad(const float *dy, int lddy, const float *wp, float *dx, int lddx,
                        const float *xact, int ldxact, float slope,
                        int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad,
                        void *ws, size_t ws_bytes, void *stream);
/* dW packed as [splits][R*S][K][C] with K = out channels, C = in channels; unpack with deconv=1. */
int stem_deconv2d_wgrad(const float *x, int ldx, const float *dy, int lddy, float *dwp, float *db,
                        int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int opad,
                        int splits, int flags, void *stream);

/* GDN / IGDN forward, compressai/layers/gdn.py:52-67 with the NonNegativeParametrizer
 * (compressai/ops/parametrizers.py:42-45) applied to the stored beta[C], gamma[C,C] on the fly. */
int stem_gdn_fwd(const float *x, int ldx, const float *beta, const float *gamma, float *y, int ldy,
                 int B, int H, int W, int C, int inverse, float beta_min, void *stream);

/* Convolution (or transposed convolution) with the following GDN / IGDN fused into its epilogue (inference path
 * of g_a / g_s, priors.py:421-439): the conv output of a pixel tile stays in registers, its squares are parked in
 * LDS and the normalisation is a second K = N contraction in the same kernel.  N <= 192, N % 4 == 0.            */
int stem_conv2d_gdn_fwd(const float *x, int ldx, const float *wp, const float *bias, const float *beta,
                        const float *gamma, float *y, int ldy, int B, int H, int W, int C, int K, int R, int S,
                        int stride, int pad, int inverse, float beta_min, void *stream);
int stem_conv2d_fwd_c4_gdn(const float *x4, const float *wp, const float *bias, const float *beta,
                           const float *gamma, float *y, int ldy, int B, int H, int W, int K, int R, int S,
                           int stride, int pad, int inverse, float beta_min, void *stream);
int stem_deconv2d_gdn_fwd(const float *x, int ldx, const float *wp, const float *bias, const float *beta,
                          const float *gamma, float *y, int ldy, int B, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, int opad, int inverse, float beta_min, void *stream);

/* GDN / IGDN backward (torch autograd of gdn.py:52-67 incl. the NonNegativeParametrizer / LowerBound chain):
 * x, dy -> dx, dbeta[C], dgamma[C,C] (gradients of the STORED parameters).  The denominator is recomputed
 * (one K=C contraction), dx needs a second one with gamma'^T, dgamma a pixel reduction of g (x) x^2.
 * ws: stem_gdn_bwd_workspace_bytes(B,H,W,C) bytes of scratch.                                                    */
size_t stem_gdn_bwd_workspace_bytes(int B, int H, int W, int C);
int stem_gdn_bwd(const float *x, int ldx, const float *dy, int lddy, const float *beta, const float *gamma,
                 float *dx, int lddx, float *dbeta, float *dgamma, int B, int H, int W, int C, int inverse,
                 float beta_min, void *ws, size_t ws_bytes, void *stream);

/* LeakyReLU backward given the activation OUTPUT (sign-preserving): dx = dy * (yact>0 ? 1 : slope). */
int stem_lrelu_bwd(const float *yact, const float *dy, float *dx, size_t n, float slope, void *stream);

/* ---- variable-rate (ROI) models: SFT modulation and quality-map pooling (compressai/models/stem_utils.py:24-63) ----
 * out = act(x * (1 + gamma) + beta), act = LeakyReLU(slope) (slope 1: none).  Dense tensors of n floats.           */
int stem_sft_fwd(const float *x, const float *gamma, const float *beta, float *out, size_t n, float slope, void *stream);
int stem_sft_bwd(const float *x, const float *gamma, const float *out, const float *dout, float *dx, float *dgamma,
                 float *dbeta, size_t n, float slope, void *stream);
int stem_lrelu_fwd(const float *x, float *y, size_t n, float slope, void *stream);
/* adaptive_avg_pool2d for integer ratios: x[B,H,W,C] -> y[B,Ho,Wo,C] */
int stem_avgpool_fwd(const float *x, int ldx, float *y, int ldy, int B, int H, int W, int C, int Ho, int Wo, void *stream);
/* its adjoint: dx[B,H,W,C] = dy[B,Ho,Wo,C] / window, broadcast over each window */
int stem_avgpool_bwd(const float *dy, int ldy, float *dx, int ldx, int B, int H, int W, int C, int Ho, int Wo, void *stream);
/* pixel-wise weighted distortion (PixelwiseRateDistortionLoss, reference utils.py:53-74): NCHW images, lambda [B,1,H,W].
 * *acc (fp64, device) += sum lambda (xhat-x)^2 ;  dxhat = (*g) * coef * 2 lambda (xhat-x) with g a device scalar. */
int stem_weighted_sqerr_sum(const float *xhat, const float *x, const float *lambda, int B, int C, size_t HW, double *acc, void *stream);
int stem_weighted_sqerr_bwd(const float *xhat, const float *x, const float *lambda, float *dxhat, int B, int C, size_t HW,
                            const double *g, float coef, void *stream);
/* ---- training-data pipeline on the device (stem/dataset_vidseq.py, stem_roi/stem_roi_dataset.py) ----
 * src: decoded frames uint8 [B][T][H][W][3]; params int32 [B][3] = (top, left, flip); dst float [T][B][3][crop][crop]:
 * frame t of sample b is source frame (flip ? T-1-t : t), cropped, as ToTensor would give it (byte / 255).            */
int stem_crop_u8_to_f32(const unsigned char *src, float *dst, const int *params, int B, int T, int H, int W, int crop, void *stream);
/* quality maps: params double [B][stem_qmap_params_per_sample()] = mode (0 uniform, 1 gradation, 2 Gaussians), value|v1|count,
 * v2|final factor, transpose, then (mu_row, mu_col, var_row, var_col) x 20; out float [B][crop][crop] = map * inv_range. */
int stem_qmap_params_per_sample(void);
int stem_qmap_render(const double *params, float *out, int B, int crop, float inv_range, void *stream);

/* ---- layout ------------------------------------------------------------ */
int stem_nchw_to_nhwc(const float *x, float *y, int ldy, int B, int C, int H, int W, void *stream);
int stem_nhwc_to_nchw(const float *x, int ldx, float *y, int B, int C, int H, int W, int clamp01, void *stream);
/* q (optional): scale record of the image for stem_conv2d_c4_gdn_f16x3, stem_nhwc4_qrec_floats(B, H, W) floats */
size_t stem_nhwc4_qrec_floats(int B, int H, int W);
int stem_nchw3_to_nhwc4(const float *x, float *y, int B, int H, int W, float *q, void *stream);
/* dst[p][c] = src[p][c] for c < C with independent pixel pitches: writes a tensor into a channel slice
 * of a wider buffer (what is left of torch.cat, spatiotemporalpriors.py:846).                         */
int stem_copy_channels(const float *src, int lds, float *dst, int ldd, size_t npix, int C, void *stream);

/* ---- entropy models ---------------------------------------------------- */
#define STEM_EB_NPARAM 58
/* pack the EntropyBottleneck parameters (entropy_models.py:310-328) into [C][58]:
 * per layer i: matrix_i (out x in), bias_i, factor_i (i<4).  Pointers are the 14 tensors in that order. */
int stem_eb_pack(const float *const *tensors14, float *pack, int C, void *stream);
/* scatter d(pack) [C][58] back into the 14 gradient tensors; accumulate != 0 adds (autograd's `.grad +=`) */
int stem_eb_unpack_grads(const float *dpack, float *const *tensors14, int C, int accumulate, void *stream);
/* EntropyBottleneck.forward (entropy_models.py:424-452) on z[B,H,W,C] NHWC.
 * mode 0: z_hat = z + noise (noise != NULL, same layout);  mode 1: z_hat = round(z - median) + median.
 * lik = max(|sigmoid(s*upper) - sigmoid(s*lower)|, bound).                                        */
int stem_eb_forward(const float *z, int ldz, const float *noise, const float *pack, const float *medians,
                    float *z_hat, float *lik, int B, int H, int W, int C, int mode, float bound, void *stream);
/* backward: dlik -> dz (+= dzhat_in if non-NULL) and dpack[C][58].                                */
int stem_eb_backward(const float *z_hat, const float *pack, const float *dlik, const float *dzhat_in,
                     float *dz, float *dpack, int B, int H, int W, int C, float bound, void *stream);
/* ... also leaving the scale record of dz (16 + C floats; NULL: none) for the split in front of the hyper encoder's backward */
int stem_eb_backward_rec(const float *z_hat, const float *pack, const float *dlik, const float *dzhat_in,
                         float *dz, float *dpack, int B, int H, int W, int C, float bound, float *dz_rec, void *stream);
/* EntropyBottleneck.loss (entropy_models.py:383-386): loss[1], dquantiles[C][3]                   */
int stem_eb_aux_loss(const float *quantiles, const float *pack, const float *target3, float *loss,
                     float *dquantiles, int C, void *stream);

/* GaussianConditional.forward (entropy_models.py:570-596) fused with quantize:
 * mode 0: out = y + noise ; mode 1: out = round(y - mean) + mean.  y/noise/out/lik contiguous NHWC [n_pix][C];
 * scales / means are channel slices (ld) of the EPM output.                                       */
int stem_gc_forward(const float *y, const float *noise, const float *scales, const float *means, int ldsm,
                    float *out, float *lik, size_t npix, int C, int mode, float scale_bound, float lik_bound,
                    void *stream);
int stem_gc_backward(const float *out, const float *scales, const float *means, int ldsm, const float *dlik,
                     float *dscales, float *dmeans, int lddsm, float *dy, size_t npix, int C,
                     float scale_bound, float lik_bound, float *q, void *stream);
/* q (optional, stem_rate_partials(npix * C) + 16 floats): scale record of (dscales | dmeans), one max |value| per workgroup */
/* sum(log2(lik)) accumulated in double: acc[0] += sum.  (EMLoss, utils.py:18-27)                   */
int stem_log2_sum(const float *lik, size_t n, double *acc, void *stream);
/* dlik = coef / lik  (gradient of coef*sum(log lik))                                              */
int stem_dlog(const float *lik, float *dlik, size_t n, float coef, void *stream);
/* elementwise helpers on contiguous buffers */
int stem_sub(const float *a, const float *b, float *out, size_t n, void *stream);
int stem_add(const float *a, const float *b, float *out, size_t n, void *stream);
int stem_round(const float *a, float *out, size_t n, void *stream);
/* counter-based uniform noise in [-0.5, 0.5) (Philox4x32-10), replaces Tensor.uniform_ at
 * entropy_models.py:119; (seed, offset) make the stream reproducible and rank-dependent.          */
int stem_uniform_noise(float *out, size_t n, uint64_t seed, uint64_t offset, void *stream);
/* same stream, advanced by a DEVICE-resident draw count: counter = offset + epoch_dev[0] * epoch_stride + i.  For launches
 * captured in a hipGraph (arguments are frozen at capture; the epoch is bumped by stem_counter_add inside the graph). */
int stem_uniform_noise_epoch(float *out, size_t n, uint64_t seed, uint64_t offset, const long long *epoch_dev,
                             uint64_t epoch_stride, void *stream);
/* ---- fused training glue (one P-frame optimisation step, stem/trainSTEM.py:203-218) --------------------------------
 * The elementwise work between the convolutions, regrouped so that a step issues 4 small kernels instead of ~30.
 * Noise: `noise` != NULL supplies U(-1/2,1/2) explicitly (dense [npix][C]); otherwise it is drawn from the Philox stream
 * of stem_uniform_noise at (seed, offset [+ epoch_dev[0] * epoch_stride]) -- the same numbers stem_uniform_noise(out,
 * npix*C, seed, offset) would produce.
 * stem_prior_prologue (spatiotemporalpriors.py:846-856,863): he_in[:, :C] = y_cur, he_in[:, C:2C] = y_cond (pitch ldh);
 *   target = y_cur - y_cond (residual) or y_cur; t_hat (may be NULL) = target + noise (training) / round(target);
 *   y_hat (may be NULL) = t_hat + y_cond (residual) / t_hat.  target, t_hat, y_hat are dense.
 * stem_eb_forward_train / stem_gc_forward_train: training-mode forward of the entropy models (entropy_models.py:424-452,
 *   588-596) that also emits dlik = coef / lik (EMLoss is a sum of logs, utils.py:18-27: coef = -1 / (ln 2 * N*H*W)) and
 *   per-workgroup partial sums of log2(lik) (double, `partials` holds stem_rate_partials(npix*C) entries).
 * stem_em_loss_finalize: out3 = {y_bpp, z_bpp, loss} = scale * fixed-order sums of the partials (scale = -1/(N*H*W)).
 * stem_eb_aux_loss_grad: EntropyBottleneck.loss (entropy_models.py:383-386) and its gradient w.r.t. the quantiles in
 *   one workgroup: loss[0] is written (not accumulated), dquantiles written or added (accumulate != 0).             */
int stem_prior_prologue(const float *y_cur, int ldc, const float *y_cond, int ldd, float *he_in, int ldh, float *target,
                        float *t_hat, float *y_hat, const float *noise, uint64_t seed, uint64_t offset,
                        const long long *epoch_dev, uint64_t epoch_stride, size_t npix, int C, int residual, int training,
                        float *q_in, float *q_t, void *stream);
/* q_in / q_t (optional, stem_rate_partials(npix * C / 4) + 16 floats each): scale records of he_in (max over y_cur and y_cond)
 * and of t_hat, for stem_f16x2_split_nhwc(..., src_q) of these tensors */
int stem_rate_partials(size_t n);
int stem_eb_forward_train(const float *z, int ldz, const float *pack, const float *noise, uint64_t seed, uint64_t offset,
                          const long long *epoch_dev, uint64_t epoch_stride, float *z_hat, float *lik, float *dlik,
                          double *partials, size_t npix, int C, float bound, float coef, void *stream);
/* ... also leaving the scale record of z_hat (16 + ceil(npix * C / 256) floats; NULL: none) for the split in front of the hyper
 * decoder's fp16 layers: no separate maximum pass over z_hat */
int stem_eb_forward_train_rec(const float *z, int ldz, const float *pack, const float *noise, uint64_t seed, uint64_t offset,
                              const long long *epoch_dev, uint64_t epoch_stride, float *z_hat, float *lik, float *dlik,
                              double *partials, size_t npix, int C, float bound, float coef, float *zhat_rec, void *stream);
int stem_gc_forward_train(const float *y, const float *scales, const float *means, int ldsm, const float *noise,
                          uint64_t seed, uint64_t offset, const long long *epoch_dev, uint64_t epoch_stride, float *out,
                          float *lik, float *dlik, double *partials, size_t npix, int C, float scale_bound, float lik_bound,
                          float coef, void *stream);
/* the same with the backward of stem_gc_backward folded in (EMLoss: d loss / d likelihood = coef / lik is known in the forward,
 * utils.py:18-27): dscales / dmeans [npix][lddsm] and, optionally, their scale record q -- one launch less per P-frame step */
int stem_gc_forward_backward_train(const float *y, const float *scales, const float *means, int ldsm, const float *noise,
                                   uint64_t seed, uint64_t offset, const long long *epoch_dev, uint64_t epoch_stride, float *out,
                                   float *lik, float *dlik, double *partials, size_t npix, int C, float scale_bound,
                                   float lik_bound, float coef, float *dscales, float *dmeans, int lddsm, float *q, void *stream);
int stem_em_loss_finalize(const double *partials_y, int ny, const double *partials_z, int nz, double scale, double *out3,
                          void *stream);
int stem_eb_aux_loss_grad(const float *quantiles, const float *pack, const float *target3, float *loss, float *dquantiles,
                          int C, int accumulate, void *stream);

/* GaussianConditional.build_indexes + quantize("symbols") (entropy_models.py:598-604,137-150)     */
int stem_build_indexes(const float *scales, int lds, const float *table, int T, int32_t *idx, size_t npix, int C,
                       float scale_bound, void *stream);

/* ---- autoregressive coding loop (spatiotemporalpriors.py:916-961, 1015-1054) ----------------------
 * y[n] = act(bias[n] + sum over up to three contiguous input segments of W[n][woff_i + k] * x_i[k]);
 * one wavefront per output row, shuffle reduction.  Segments let the 5x5 context window and
 * cat(tp, hp, ctx) be consumed in place.  Lengths / offsets are multiples of 4 floats, 16-byte aligned. */
int stem_gemv3(const float *W, int ldw, const float *bias, const float *x0, int len0, int woff0,
               const float *x1, int len1, int woff1, const float *x2, int len2, int woff2, float *y, int N,
               int act, float slope, void *stream);
/* MaskedConv2d weight [K][C][5][5] -> [K][12 live taps][C] for stem_gemv3 */
int stem_pack_ctx_gemv(const float *w, float *out, int K, int C, void *stream);
/* gp = scales[M] | means[M] of one pixel.  encode: idx = build_indexes(scale), sym = round(pix - mean),
 * pix <- sym + mean (spatiotemporalpriors.py:945-952).  decode: pix <- sym + mean (:1050-1054).        */
int stem_ar_finish_encode(const float *gp, const float *table, int T, float scale_bound, float *pix,
                          int32_t *sym, int32_t *idx, int M, void *stream);
int stem_ar_index(const float *gp, const float *table, int T, float scale_bound, int32_t *idx, int M, void *stream);
int stem_ar_finish_decode(const float *gp, const int32_t *sym, float *pix, int M, void *stream);
/* Decoder variants of stem_gemv3 (two launches fewer per position).  sym_prev != NULL: workgroup 0 writes the previous
 * position's y_hat = sym_prev + mean_prev to pix_prev, and with prev_is_left the second half of segment 2 (the left
 * neighbour, len2 == 2M) is taken from (sym_prev, mean_prev) instead of memory.  table != NULL: outputs n < M are also
 * turned into CDF indexes idx[n] (build_indexes, entropy_models.py:556-562).  sym_prev / idx may be pinned host memory. */
int stem_gemv3_decode(const float *W, int ldw, const float *bias, const float *x0, int len0, int woff0,
                      const float *x1, int len1, int woff1, const float *x2, int len2, int woff2, float *y, int N,
                      int act, float slope, const int32_t *sym_prev, const float *mean_prev, float *pix_prev, int M,
                      int prev_is_left, const float *table, int T, float scale_bound, int32_t *idx, void *stream);
/* The whole raster-order decode of one image (spatiotemporalpriors.py:1015-1054) in one call: per position the four
 * products above, a stream synchronisation, and `decode` -- the host symbol decoder, injected as a C function pointer with
 * the signature of stem_rans_decoder_decode (include/stem_rans.h) so that libstem_hip does not link libstem_rans -- pops M
 * symbols for the M indexes in the pinned mailbox.  buf: zero-initialised padded latent [(H+4)][(W+4)][M] of this image,
 * filled with y_hat on return; tp (may be NULL) / hp: [H*W][2M]. */
/* Encoder counterpart: the W + 3(H-1) wavefront steps (stem_gemv3_wave x4 + stem_ar_finish_encode_wave each) queued by one
 * call; wctx/wh1/wh2/wgp: scratch [min(H,(W+2)/3)][2M | n0 | n1 | 2M]; sym/idx: [H*W][M] in raster order for ONE host rANS
 * call (spatiotemporalpriors.py:916-961).  buf holds the padded target on entry and the reconstruction on return. */
int stem_ar_encode_image(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0, int n0,
                         const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2, const float *b2,
                         float *buf, int H, int W, int M, int pad, const float *tp, const float *hp,
                         float *wctx, float *wh1, float *wh2, float *wgp, const float *table, int T, float scale_bound,
                         float slope, int32_t *sym, int32_t *idx, void *stream);
typedef int (*stem_symbol_decoder_fn)(void *dec, const int32_t *indexes, size_t n, const int32_t *cdfs, int ncdf, int cdf_stride,
                                      const int32_t *sizes, const int32_t *offsets, int32_t *out);
int stem_ar_decode_image(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0, int n0,
                         const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2, const float *b2,
                         float *buf, int H, int W, int M, int pad, const float *tp, const float *hp,
                         float *ctx, float *h1, float *h2, float *gp, const float *table, int T, float scale_bound, float slope,
                         int32_t *idx_host, int32_t *sym_host, stem_symbol_decoder_fn decode, void *dec,
                         const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets, void *stream);
/* The same loop as ONE persistent kernel (csrc/ar_persistent.hip): 64 resident workgroups taken from one XCD walk through
 * the positions together, grid barriers instead of kernel boundaries between the four products of a position, the host's
 * symbol decoder reached through pinned mailboxes and sequence flags (held by the library, per calling thread) instead of a
 * stream synchronisation.  Same arguments and results as stem_ar_decode_image (bit for bit: same dot-product order) without
 * the caller's mailboxes; every wait is bounded, a non-zero return leaves `buf` and the decoder state undefined (decode the
 * image again with stem_ar_decode_image from a fresh decoder).  Replaces the loop body of spatiotemporalpriors.py:1015-1054. */
/* 1 if stem_ar_decode_image_persistent handles these widths (M latent channels; n0 / n1 outputs of EPM.0 / EPM.2): its workgroups
 * keep the weights of fixed output rows resident (M <= 204, n0, n1 <= 768: every model of spatiotemporalpriors.py) */
int stem_ar_decode_image_persistent_supported(int M, int n0, int n1);
/* calling thread: its next persistent decodes run on XCD `xcc` (0..7; -1 = first come): several images at once, one XCD + one stream +
 * one host thread each (per-thread mailboxes inside the library) */
int stem_ar_decode_image_persistent_prefer_xcc(int xcc);
int stem_ar_decode_image_persistent(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0, int n0,
                                    const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2, const float *b2,
                                    float *buf, int H, int W, int M, int pad, const float *tp, const float *hp,
                                    float *ctx, float *h1, float *h2, float *gp, const float *table, int T, float scale_bound, float slope,
                                    stem_symbol_decoder_fn decode, void *dec,
                                    const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets, void *stream);
/* The same loop for G (1..8) independent images in lockstep -- the batch elements of decompress(), which the reference decodes
 * one after the other (spatiotemporalpriors.py:1015-1054): one set of four launches + one synchronisation advances all G
 * images by a position, so the per-position latency (what bounds the decoder) is shared.  Per image the arithmetic, and
 * therefore every symbol / index / byte, is that of stem_ar_decode_image.  buf [G][(H+4)][(W+4)][M]; tp / hp [G][H*W][2M];
 * ctx [G][2M], h1 [G][n0], h2 [G][n1], gp [G][2M]; idx_host / sym_host [G][M] pinned; decs[G] = one decoder handle per image. */
int stem_ar_decode_batch(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0, int n0,
                         const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2, const float *b2,
                         float *buf, int G, int H, int W, int M, int pad, const float *tp, const float *hp,
                         float *ctx, float *h1, float *h2, float *gp, const float *table, int T, float scale_bound, float slope,
                         int32_t *idx_host, int32_t *sym_host, stem_symbol_decoder_fn decode, void *const *decs,
                         const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets, void *stream);

/* stem_ar_decode_batch without a stream synchronisation per position: device and host hand the indexes / symbols over
 * through flags in pinned memory (owned by the library), the launches of position p + 2 are issued while the device works on
 * p + 1, and the images alternate in two groups so that the host decodes one group while the device advances the other.
 * Every wait is bounded (device ~1 s, host 5 s -> error return).  Same arithmetic per image, same symbols. */
int stem_ar_decode_batch_pipelined(const float *w_ctx, int ld_ctx, const float *b_ctx, const float *w0, int ld0, const float *b0,
                                   int n0, const float *w1, int ld1, const float *b1, int n1, const float *w2, int ld2,
                                   const float *b2, float *buf, int G, int H, int W, int M, int pad, const float *tp,
                                   const float *hp, float *ctx, float *h1, float *h2, float *gp, const float *table, int T,
                                   float scale_bound, float slope, stem_symbol_decoder_fn decode, void *const *decs,
                                   const int32_t *cdfs, int ncdf, int cdf_stride, const int32_t *sizes, const int32_t *offsets,
                                   void *stream);

/* ---- fp32-accurate convolution on the 16-bit matrix cores (csrc/conv_f16x3.hip, wgrad_f16x3.hip, c4gdn_f16x3.hip) --------
 * Replaces the fp32 convolutions of the reference on this path (compressai/models/priors.py:613-621 under no_grad in
 * stem/trainSTEM.py:128,171; spatiotemporalpriors.py:814-838 with torch autograd).  Every fp32 operand a is stored as two fp16
 * numbers a0 = rn(a * 2^e), a1 = rn(a * 2^e - a0) (|a * 2^e - a0 - a1| <= 2^-22 |a| 2^e); the three fp16 products a0.b0, a0.b1,
 * a1.b0 are exact in the fp32 accumulator and carry the fp32 product to 2^-21; accumulation is fp32.  (Round 2 used three
 * bf16 planes and six products: no scaling, twice the matrix instructions.)
 * "planes" layout of an NHWC tensor with C % 32 == 0: [pixel][C/32][2][32] fp16 (128 B per pixel and 32-channel slab),
 * followed by the tensor's SCALE RECORD at byte stem_f16x2_planes_qrec_offset():
 *     int nslots; float inv = 2^-e; 14 reserved words; float max_abs[nslots]   (one slot per producing workgroup)
 * The record is written by whatever kernel writes the planes (plain stores, nothing to initialise): a split takes e from the
 * measured maximum of its input, a convolution from the bound K * max|x| * max|w| + max|bias| and records the measured
 * maximum of its output for the next consumer.  Records travel as explicit pointers (`xq`, `yq`) because channel views of a
 * planes tensor share the record of the whole tensor.  Packed weight images carry their record behind the image
 * (stem_f16x2_conv_weight*_bytes include it).  Operands must satisfy K * max|x| * max|w| < 3e38.                          */
size_t stem_f16x2_planes_bytes(long npix, int C);        /* payload + scale record */
size_t stem_f16x2_planes_qrec_offset(long npix, int C);  /* = payload bytes */
size_t stem_f16x2_conv_weight_bytes(int C, int R, int S);
/* max |x| of an NHWC fp32 tensor into the slots of record q (at most max_slots of them, <= 1024) */
int stem_amax_nhwc(const float *x, int ldx, long npix, int C, float *q, long max_slots, void *stream);
/* src_q: a record whose slots already hold max |x| (written by the kernel that produced x); null = measure here first */
int stem_f16x2_split_nhwc(const float *x, int ldx, void *xp, float *xq, const float *src_q, long npix, int C, void *stream);
int stem_f16x2_merge_nhwc(const void *xp, const float *xq, float *x, int ldx, long npix, int C, void *stream);
/* planes of x * (z > 0 ? 1 : slope): the gradient entering a convolution whose output z went through leaky_relu(., slope)
 * (what autograd computes as LeakyReluBackward before ConvolutionBackward, stem_roi.py's conv + LeakyReLU pairs), split in
 * the same pass for the input-gradient / weight-gradient kernels */
int stem_f16x2_split_dact_nhwc(const float *x, int ldx, const float *z, int ldz, float slope, void *xp, float *xq, const float *src_q,
                                long npix, int C, void *stream);
/* w: the torch Conv2d weight [N][C][R][S] (NOT one of the stem_pack_* layouts); N <= 192, R*S <= 25 */
int stem_f16x2_pack_conv_weight(const float *w, void *wp, int N, int C, int R, int S, void *stream);
/* y = conv(x) + bias, followed by GDN when beta / gp are given (gdn.py:52-67).  beta: the STORED parameter (reparametrised on
 * the fly); gp: gamma' = max(gamma, 2^-18)^2 - 2^-36 (parametrizers.py:42-45) as the packed image of a 1x1 convolution weight
 * [N][ceil32(N)] (zero-padded columns), i.e. stem_f16x2_pack_conv_weight(gamma', gp, N, ceil32(N), 1, 1): the GDN's
 * contraction over the squared outputs runs on the same fp16 instruction as the convolution, with the squares scaled by
 * the workgroup tile's own maximum.  Repack when gamma changes.
 * Output as fp32 NHWC (y, ldy) and / or as planes (yp, with its record yq); either may be null.  yq without yp: only the
 * measured maximum of y is recorded (for a later stem_f16x2_split_nhwc(..., src_q = yq)). */
int stem_conv2d_f16x3_fwd(const void *xp, const float *xq, const void *wp, const float *bias, const float *beta, const void *gp,
                           float beta_min, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int R, int S,
                           int stride, int pad, void *stream);
/* the same kernel for the conv + LeakyReLU / ReLU pairs of the layer-wise models at large pixel counts (stem_roi.py:520-608,
 * stem_utils.py:24-63): act = 1 applies leaky_relu(., slope) after the bias.  stem_f16x2_pack_conv_weight_flip packs the
 * operand of the INPUT GRADIENT of a stride-1 convolution with torch weight w[C][N][R][S] (N = its input channels <= 192):
 * that gradient is stem_conv2d_f16x3_fwd_act(dy planes, flipped pack, no bias) with the same padding.                      */
int stem_conv2d_f16x3_fwd_act(const void *xp, const float *xq, const void *wp, const float *bias, int act, float slope, float *y, int ldy,
                               void *yp, float *yq, int B, int H, int W, int C, int N, int R, int S, int stride, int pad, void *stream);
int stem_f16x2_pack_conv_weight_flip(const float *w, void *wp, int N, int C, int R, int S, void *stream);
/* First analysis layer + its GDN as ONE kernel (csrc/c4gdn_f16x3.hip): replaces `self.g_a[0:2]` = conv(3, N) ; GDN(N) of
 * compressai/models/priors.py:421-423 (gdn.py:52-67) under no_grad (stem/trainSTEM.py:128,171).  N = 64, 128 or 192
 * (stem_c4gdn_supported), R*S <= 25.  Both contractions run on the 16-bit matrix cores with fp32-class products and fp32
 * accumulation; the squared conv outputs go from the accumulators into the GDN contraction as registers.  stem_c4gdn_pack
 * builds the A-operand stream (stem_c4gdn_stream_bytes bytes, scale records included) from the STEM_PACK_CONV_FWD_C4 weight
 * and the STORED gamma (reparametrised there: max(gamma, 2^-18)^2 - 2^-36); repack when either changes.
 * x4 = stem_nchw3_to_nhwc4's buffer, xq = the record that call filled; output as fp32 NHWC (y, ldy) and / or planes (yp, yq). */
int stem_c4gdn_supported(int N, int R, int S);
size_t stem_c4gdn_stream_bytes(int N, int R, int S);
int stem_c4gdn_pack(const float *wp_c4, const float *gamma, void *astream, int N, int R, int S, void *stream);
int stem_conv2d_c4_gdn_f16x3(const float *x4, const float *xq, const void *astream, const float *bias, const float *beta, float beta_min,
                              float *y, int ldy, void *yp, float *yq, int B, int H, int W, int N, int R, int S, int stride, int pad,
                              void *stream);

/* General variant of stem_conv2d_f16x3_fwd for the small-M layers of the STEM network at training time: any N (tiles of 128
 * output channels), split-K with an in-kernel deterministic reduction, epilogue epi = 0 bias | 1 bias + leaky ReLU(slope) |
 * 2 times the leaky ReLU derivative selected by z (the input-gradient of a convolution whose input was activated:
 * torch autograd of spatiotemporalpriors.py:814-838).  Weights: stem_f16x2_pack_conv_weight_gen of the torch weight
 * [K][C][R][S]; flip = 1 packs the operand of the input-gradient of a stride-1 convolution (then N = C, and the planes input is
 * dy with K channels).  ws: stem_conv2d_f16x3_gen_workspace_bytes bytes whose first 64 KiB are zero before the first use
 * (the kernel leaves them zero -- the same contract as stem_conv_workspace_bytes, the buffer may be shared); null = unsplit.  Results do not depend on which workgroup arrives last. */
size_t stem_f16x2_conv_weight_gen_bytes(int N, int C, int R, int S);
/* taps: 0 = all R*S taps; 0 < taps < R*S = a MASKED convolution (layers.py:21-47: mask type A keeps the first (R/2)*S + S/2 taps in
 * row-major order, type B one more): only those are packed, and the others are ZEROED IN PLACE in w, as the reference does at
 * every forward (`self.weight.data *= self.mask`); forward role only.  The same value goes to stem_conv2d_f16x3_gen_fwd. */
int stem_f16x2_pack_conv_weight_gen(const float *w, void *wp, int N, int C, int R, int S, int flip, int taps, void *stream);
/* all layers of a training model with two launches (maxima, images: their weights change every optimiser step); N, C as in
 * the single call, i.e. already swapped for flip = 1 */
typedef struct {
    const void *w;
    void *wp;
    int N, C, R, S, flip, taps;
    const float *bmax;      /* optional: per-chunk maxima left by stem_adam_step_bmax for the flat buffer w lives in ... */
    int b0, nb, rsv0, rsv1; /* ... chunks b0 .. b0 + nb - 1 cover the tensor: no maximum pass is launched when EVERY descriptor has them */
} stem_f16x2_pack_desc;
int stem_f16x2_pack_conv_weights_multi(const stem_f16x2_pack_desc *descs, int n, void *stream);
/* BOTH images of a layer (forward role + input-gradient role: two transposes of the same numbers) from ONE read of its weights;
 * replaces two stem_f16x2_pack_conv_weights_multi calls per optimiser step (stem/trainSTEM.py:213: every weight changed).
 * Tensor w[A][B][R][S]; image r (wp0 / wp1, null = none): mode bit 0: 0 = rows are a (contraction over b: a Conv2d's forward
 * image, a ConvTranspose2d's input-gradient image), 1 = rows are b (contraction over a); mode >> 1: tap order 0 identity,
 * 1 mirrored (flip = 1 above), 2 sub-pixel phases (flip = 2); taps0 > 0: masked convolution, the image holds the first taps0
 * taps and the others are zeroed in w.  bmax / b0 / nb: the optimiser pass's chunk maxima covering the tensor (mandatory).
 * Padding rows (row count not a multiple of 128) are not written: zero-fill the images once. */
typedef struct {
    const void *w;
    int A, B, R, S;
    void *wp0;
    int mode0, taps0;
    void *wp1;
    int mode1, taps1;
    const float *bmax;
    int b0, nb;
} stem_f16x2_pair_desc;
int stem_f16x2_pack_conv_weights_pair_multi(const stem_f16x2_pair_desc *descs, int n, void *stream);
size_t stem_conv2d_f16x3_gen_workspace_bytes(int B, int H, int W, int C, int N, int R, int S, int stride, int pad, int taps);
/* xpix: bytes per pixel of the planes buffer xp points into (0 = dense, (C/32) * 128); xp may point at a 32-channel-aligned
 * slab of a wider planes tensor (xq: the record of that whole tensor) */
int stem_conv2d_f16x3_gen_fwd(const void *xp, const float *xq, int xpix, const void *wp, const float *bias, int epi, float slope,
                               const float *z, int ldz, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int R,
                               int S, int stride, int pad, int taps, void *ws, size_t ws_bytes, void *stream);
/* ... for rows [n0, n0 + N) of a weight image of N_image rows (whole 128-row tiles): a layer's outputs range by range, so that the
 * consumer of one range need not wait for the others (the input gradient of EPM.0 feeds three independent chains) */
int stem_conv2d_f16x3_gen_fwd_rows(const void *xp, const float *xq, int xpix, const void *wp_image, int N_image, int n0, const float *bias,
                                    int epi, float slope, const float *z, int ldz, float *y, int ldy, void *yp, float *yq, int B, int H, int W,
                                    int C, int N, int R, int S, int stride, int pad, int taps, void *ws, size_t ws_bytes, void *stream);
/* The TRANSPOSED face of a stride-2, R x R (odd), padding R/2 layer whose fine grid is exactly twice the coarse one, on the same
 * kernel as ONE launch over its four sub-pixel phases (each a stride-1 convolution of the coarse grid with the taps of its parity:
 * 3x3 / 3x2 / 2x3 / 2x2 for 5x5 -- no structural zeros):
 *   forward of nn.ConvTranspose2d(C, N, R, stride=2, padding=R/2, output_padding=1)     HD.0 / HD.2 (spatiotemporalpriors.py:822-826)
 *   input gradient of nn.Conv2d(N, C, R, stride=2, padding=R/2), even-sized input       HE.2 / HE.4 (:814-818; torch autograd)
 * xp: planes of the coarse tensor [B, H, W, C]; wp: the four phase images made by stem_f16x2_pack_conv_weights_multi with
 * flip = 2 (rows = the N outputs, contraction channels = C, i.e. the torch weight read as w[c][n][r][s]); y / yp / z: fp32 rows /
 * planes / activation rows of the fine tensor [B, OHf, OWf, N] with OHf = 2H, or 2H - 1 for the input gradient of a strided
 * convolution whose input had an odd number of rows (likewise OWf); epi, slope, ws as in stem_conv2d_f16x3_gen_fwd. */
size_t stem_tconv2d_f16x3_workspace_bytes(int B, int H, int W, int C, int N, int R);
int stem_tconv2d_f16x3_fwd(const void *xp, const float *xq, int xpix, const void *wp, const float *bias, int epi, float slope,
                            const float *z, int ldz, float *y, int ldy, void *yp, float *yq, int B, int H, int W, int C, int N, int R,
                            int OHf, int OWf, void *ws, size_t ws_bytes, void *stream);

/* Weight gradient of a stride-1 nn.Conv2d on the 16-bit matrix cores (csrc/wgrad_f16x3.hip): x and dy as planes with their
 * records (pitches in bytes per pixel, 0 = dense; 32-aligned channel views allowed), result as `splits` slabs [R*S][K][C] like
 * stem_conv2d_wgrad (sum / transpose with stem_unpack_wgrads_multi).  splits = stem_wgrad_f16x3_splits(...); dwp holds
 * splits*R*S*K*C floats.  The bias gradient (column sums of the fp32 dy) is stem_bias_grad. */
int stem_wgrad_f16x3_splits(int B, int H, int W, int C, int K, int R, int S, int pad);
/* bias_part (optional, splits * K floats): per-split column sums of dy, to be finished by stem_bias_grad_final */
int stem_conv2d_wgrad_f16x3(const void *xp, const float *xq, int xpix, const void *dyp, const float *dyq, int dypix, float *dwp,
                             float *bias_part, int B, int H, int W, int C, int K, int R, int S, int pad, int splits, void *stream);
/* ... of a STRIDED layer (per-tap form): dW[k][c][r][s] = sum over coarse pixels of g[b, oy, ox][k] * f[b, oy*stride + r - pad, ...][c].
 * g = `dyp` (K channels, coarse grid), f = `xp` (C channels, fine grid H x W).  nn.Conv2d (HE.2 / HE.4, spatiotemporalpriors.py:814-818):
 * f = input, g = output gradient, slabs [t][K][C].  nn.ConvTranspose2d (HD.0 / HD.2, :822-826; weight [Cin][Cout][R][S]): g = the
 * layer's input (K = Cin), f = the gradient of its output (C = Cout), slabs [t][Cin][Cout] (unpack with STEM_UNPACK_DECONV). */
int stem_wgrad_f16x3_strided_splits(int B, int H, int W, int C, int K, int R, int S, int stride, int pad);
int stem_conv2d_wgrad_f16x3_strided(const void *xp, const float *xq, int xpix, const void *dyp, const float *dyq, int dypix, float *dwp,
                                     float *bias_part, int B, int H, int W, int C, int K, int R, int S, int stride, int pad, int splits,
                                     void *stream);
int stem_bias_grad_final(const float *part, int K, int parts, float *db, int accumulate, void *stream);
size_t stem_bias_grad_scratch_elems(long npix, int K);
int stem_bias_grad(const float *dy, int lddy, long npix, int K, float *scratch, float *db, int accumulate, void *stream);

/* Wavefront-parallel encoder: all latent positions with the same t = w + 3h are independent under the 5x5
 * type-A mask, so a H x W frame is coded in W + 3(H-1) batched steps instead of H*W sequential ones.  Input
 * segment of position p at (h, w): x + sh*h + sw*w + sp*p (element offsets); output y[p*ldy + n].            */
typedef struct {
    const float *x;
    int len, woff;
    long sh, sw, sp;
} stem_wave_seg;
int stem_gemv3_wave(const float *W, int ldw, const float *bias, const stem_wave_seg *segs3, float *y, int ldy, int N,
                    int act, float slope, int t, int H, int Wd, void *stream);
int stem_ar_finish_encode_wave(const float *gp, const float *table, int T, float scale_bound, float *buf,
                               int32_t *sym, int32_t *idx, int M, int t, int H, int Wd, int Wp, int pad, void *stream);

/* ---- optimiser --------------------------------------------------------- */
/* acc[0] (double) += sum of squares of a flat gradient buffer, in two stages without atomics (bit-reproducible):
 * `acc` must hold 1 + STEM_SUMSQ_SCRATCH doubles, acc[1..] is scratch for the per-workgroup partial sums. */
#define STEM_SUMSQ_SCRATCH 2048
int stem_sumsq(const float *g, size_t n, double *acc, void *stream);
/* same reduction, acc[0] = sum (no pre-zeroed accumulator needed) */
int stem_sumsq_set(const float *g, size_t n, double *acc, void *stream);
/* stand-alone torch.nn.utils.clip_grad_norm_ (stem_roi/train_stem_roi.py:536,563 clips once per frame while gradients
 * accumulate over the GOP): g *= min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)); sumsq may span several buffers. */
int stem_clip_scale(float *g, size_t n, const double *sumsq, float max_norm, void *stream);
/* y += a * x on flat buffers (running GOP gradient += 1/world * all-reduced frame gradient) */
int stem_axpy(float *y, const float *x, float a, size_t n, void *stream);
/* torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step fused over a flat buffer:
 * scale = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) (max_norm <= 0: no clipping), g *= scale * gscale. */
int stem_adam_step(float *p, const float *g, float *m, float *v, size_t n, const double *sumsq, float max_norm,
                   float gscale, float lr, float beta1, float beta2, float eps, int step, void *stream);
/* stem_adam_step that also clears `g` in the same pass (explicit training schedule: the next backward accumulates into it) */
/* stem_adam_step[_zero] over contiguous chunks of stem_adam_chunk() parameters, leaving max |p_new| per chunk in bmax
 * (4 * cdiv(n, chunk) floats: one value per wavefront of the chunk's workgroup): what stem_f16x2_pack_conv_weights_multi
 * takes its scales from right after an optimiser step */
size_t stem_adam_chunk(void);
int stem_adam_step_bmax(float *p, float *g, float *m, float *v, size_t n, const double *sumsq, float max_norm, float gscale,
                        float lr, float beta1, float beta2, float eps, int step, int zero_grad, float *bmax, void *stream);
int stem_adam_step_zero(float *p, float *g, float *m, float *v, size_t n, const double *sumsq, float max_norm,
                        float gscale, float lr, float beta1, float beta2, float eps, int step, void *stream);
/* The same update with the step count and learning rate in DEVICE memory, for optimiser steps captured in a hipGraph
 * (kernel arguments are frozen at capture): step_dev[0] is incremented first, then the update uses
 * lr_dev[0] / (1 - beta1^t) and 1 / sqrt(1 - beta2^t); scal_dev = 2 floats of scratch.  A learning-rate scheduler
 * (ReduceLROnPlateau, stem/trainSTEM.py:123,283) acts by rewriting lr_dev between replays. */
int stem_adam_step_dev(float *p, const float *g, float *m, float *v, size_t n, const double *sumsq, float max_norm,
                       float gscale, const float *lr_dev, float beta1, float beta2, float eps, long long *step_dev,
                       float *scal_dev, void *stream);
/* ctr[0] += inc on the device (epoch counters of captured graphs) */
int stem_counter_add(long long *ctr, long long inc, void *stream);

#ifdef __cplusplus
}
#endif
#endif
