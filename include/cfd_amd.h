/*
 * cfd_amd.h -- C ABI of libcfd_amd.so, the MI355X (gfx950) hot path of the CFD surrogate.
 *
 * The reference (annguyenphanhuu/CFD) has no FFI: its hot path is the arithmetic that
 * TensorFlow/Keras executes for the graphs built in {U,P,DeltaP}/models and the scipy
 * nearest-neighbour resample in pixels_coords_testing.py.  Each entry point below names
 * the reference construct it replaces (file:line under the reference root).  A binding
 * only needs plain pointers and sizes (Python: ctypes).
 *
 * Conventions
 *   - every data pointer is DEVICE memory unless the comment says "host";
 *   - activations are NHWC (Keras layout); a tensor argument is (pointer, ld) where `ld`
 *     is the channel stride between consecutive pixels of the underlying buffer, so a
 *     kernel can read or write a channel slice of a concat buffer in place;
 *   - convolution weights are packed [Cout][kh][kw][Cin] ("OHWI", K contiguous) -- use
 *     cfd_pack_* to convert from the Keras layouts;
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous on it;
 *   - every function returns 0 on success, a negative cfd_status otherwise, and never
 *     falls back to a CPU path; cfd_last_error() describes the last failure of the
 *     calling thread.
 */
#ifndef CFD_AMD_H
#define CFD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CFD_OK = 0, CFD_EINVAL = -1, CFD_EHIP = -2, CFD_EUNSUPPORTED = -3 } cfd_status;
typedef enum { CFD_F32 = 0, CFD_BF16 = 1 } cfd_dtype;
typedef enum { CFD_ACT_NONE = 0, CFD_ACT_LEAKY = 1, CFD_ACT_RELU = 2, CFD_ACT_SIGMOID = 3 } cfd_act;

/* library */
int cfd_version(void);
const char *cfd_last_error(void);
/* Tuning / A-B switches (no counterpart in the reference; process-wide, not thread-safe against running calls).
 *   CFD_TUNE_IGEMM256     -1 heuristic (default) | 0 never | 1 whenever eligible: use the 256x256 ping-pong
 *                         bf16 implicit-GEMM kernel instead of the 128x128 one
 *   CFD_TUNE_IGEMM256_PD  schedule variant of that kernel (see igemm256.hip; default = two phases per K tile)
 *   CFD_TUNE_WGRAD256     -1 heuristic | 0 never | 1 whenever eligible: the 256x256 ping-pong weight-gradient kernel
 *   CFD_TUNE_CLASS_ORDER  -1 heuristic | 0 class-major | 1 class-interleaved block order of the sub-pixel classes of a
 *                         transposed convolution (interleaved: the four classes of one pixel tile share L2 lines)
 *   CFD_TUNE_STAGED_OUT   -1 / 1 on where legal | 0 off: bf16 output tiles of the 256-wide kernel leave through LDS as whole
 *                         pixel rows instead of 32-B fragments
 *   CFD_TUNE_RESAMPLE_DENSITY  -1 default (6) | n > 0: the resample's cell grid aims at n / 4 points per cell (any value gives
 *                         the same results; it trades distance evaluations per pixel against extra search rings)
 *   CFD_TUNE_NTILE        -1 heuristic | 128 | 256: n tile of the 256-row ping-pong kernels for layers whose Cout is a
 *                         multiple of 256 (128 = the 128(n) x 256(m) variant: twice the blocks, no split-K slabs)
 *   CFD_TUNE_PATCH        0 off | 1 on where legal: bf16 3x3, 4x4 stride-2 and transposed-4x4 layers on whole 8 x 32 pixel
 *                         tiles stage their activation operand once per 64-channel chunk as an input patch (igemm256h.hip)
 *                         | 2: as 1, and every layer on 128-wide n tiles runs the CO-RESIDENT form (32-channel K tiles, 69 KiB
 *                         of LDS, two workgroups per CU) | 3: every patch-eligible layer does | -1 (default): as 1, plus the
 *                         co-resident form for 128-wide layers of at most 16 K tiles (per-tile overhead dominates there)
 *   CFD_TUNE_RESAMPLE_FUSED  -1 / 1: everything before the resample's search runs as ONE launch, a block per case (cases of up to
 *                         32 Ki points on average) | 0: the separate init / points / set-up / counting-sort launches
 *   CFD_TUNE_ONEPHASE     1: the 256-wide patch-staged kernel runs ONE cluster of 64 MFMAs per K tile (A1 fragments re-read
 *                         inside the cluster, LDS-DMA duty split by wave group) | 0 / -1: two clusters of 32 (igemm256h.hip)
 *   CFD_TUNE_SKINNY       -1 / 1: layers with at most 64 output positions per class (batch 1: U e5, b, d1, d2) run on 32-row m tiles
 *                         with a deeper K split | 0: on the 128-row tiles like every other layer
 */
typedef enum { CFD_TUNE_IGEMM256 = 0, CFD_TUNE_IGEMM256_PD = 1, CFD_TUNE_WGRAD256 = 2, CFD_TUNE_CLASS_ORDER = 3,
               CFD_TUNE_STAGED_OUT = 4, CFD_TUNE_RESAMPLE_DENSITY = 5, CFD_TUNE_NTILE = 6, CFD_TUNE_PATCH = 7,
               CFD_TUNE_RESAMPLE_FUSED = 8, CFD_TUNE_ONEPHASE = 9, CFD_TUNE_SKINNY = 10, CFD_TUNE_COUNT = 11 } cfd_tuning_key;
int cfd_set_tuning(int key, int value);

/* Measurement aid, no counterpart in the reference: `iters` rounds of 32 independent v_mfma_f32_16x16x32_bf16 per wave on
 * REGISTER operands (8 waves on each of 256 workgroups; no LDS, no memory traffic inside the loop) -- what the matrix
 * cores sustain on the operand values given, against which bench.py sets the conv kernels' TFLOP/s beside the 2.5 PFLOP/s
 * peak (the chip's clock under MFMA load depends on how many operand bits toggle: DESIGN.md section 5 (k2)).
 * operands: 65 536 x 16 bytes (bf16 values, device memory), out: 256 x 512 floats (device memory, sink).
 * Work: 2 * 16*16*32 * 32 * iters * 8 * 256 FLOP per call. */
int cfd_mfma_probe(const void *operands, float *out, int iters, void *stream);

/* ------------------------------------------------------------------------------------
 * Convolutions as implicit GEMM on MFMA.
 * Replaces Conv2D / Conv2DTranspose / Dense of
 *   U/models/generator.py:18,31,59,70   P/models/generator.py:43,56,85,102
 *   U/models/discriminator.py:22-41     DeltaP/models/generator.py:23-46
 * ---------------------------------------------------------------------------------- */
typedef enum {
  CFD_CONV_K4S2 = 0,  /* Conv2D 4x4 stride 2 'same' (pad 1,1)              out = in/2 */
  CFD_CONV_K4S1 = 1,  /* Conv2D 4x4 stride 1 'same' (pad 1 before, 2 after) out = in  */
  CFD_CONV_K3S1 = 2,  /* Conv2D 3x3 stride 1 'same' (pad 1,1)              out = in   */
  CFD_CONVT_K4S2 = 3, /* Conv2DTranspose 4x4 stride 2 'same'                out = 2*in */
  CFD_DENSE = 4       /* Dense: Hi=Wi=1, Cin = features in                             */
  /* 5 = CFD_CONV_K4S1_BWD, see the training section */
} cfd_conv_kind;

typedef struct {
  int kind;            /* cfd_conv_kind */
  int dtype;           /* cfd_dtype of x and w (arithmetic: bf16 MFMA or f32 MFMA, f32 accumulate) */
  int out_dtype;       /* cfd_dtype of y */
  int B, Hi, Wi, Cin;  /* input  [B][Hi][Wi][Cin] */
  int Cout;            /* output [B][Ho][Wo][Cout], Ho/Wo implied by kind */
  int ldx, ldy;        /* channel strides of the x and y buffers (elements) */
  int act;             /* cfd_act fused after bias */
  float leaky_slope;   /* for CFD_ACT_LEAKY */
  int split_k;         /* >=1; >1 needs `workspace` of cfd_conv_workspace_bytes() */
} cfd_conv_desc;

/* bytes of fp32 scratch needed for desc->split_k > 1 (0 otherwise) */
int64_t cfd_conv_workspace_bytes(const cfd_conv_desc *desc);
/* picks a split_k that fills the chip for this shape (host heuristic) */
int cfd_conv_suggest_split_k(const cfd_conv_desc *desc);

/* y = act(conv(x, w) + bias).  bias: fp32 [Cout] or NULL. */
int cfd_conv_forward(const cfd_conv_desc *desc, const void *x, const void *w, const float *bias, void *y,
                     void *workspace, void *stream);

/* Same as cfd_conv_forward, and additionally accumulates (atomics; the caller zeroes them) the per-sample,
 * per-channel sum and sum of squares of the stored pre-activation into stat_sum / stat_sq, fp32 [B][Cout]:
 * the batch statistics BatchNormalization(training=True) needs (generator.py:20-21,33), taken from the fp32
 * accumulators so the conv output is not re-read for them.  Only shapes with cfd_conv_stats_fusable() != 0
 * (no split-K, whole output tiles inside one sample); desc->act must be CFD_ACT_NONE. */
int cfd_conv_stats_fusable(const cfd_conv_desc *desc);
int cfd_conv_forward_stats(const cfd_conv_desc *desc, const void *x, const void *w, const float *bias, void *y,
                           void *workspace, float *stat_sum, float *stat_sq, void *stream);

/* conv (stride 1: CFD_CONV_K3S1 / CFD_CONV_K4S1) + bias + activation + MaxPooling2D((2,2)) in one launch: y is the
 * POOLED tensor [B][Hi/2][Wi/2][Cout] with channel stride desc->ldy; the unpooled activation is never written.
 * Replaces the Conv2D -> LeakyReLU -> MaxPooling2D triples of define_cnn_model at inference
 * (DeltaP/models/generator.py:27-34).  Only shapes with cfd_conv_pool2_fusable() != 0 (split_k = 1, Cout % 256 == 0,
 * Wi % 128 == 0, Hi even, out_dtype == dtype, activation NONE / RELU / LEAKY with slope >= 0, which commute with max). */
int cfd_conv_pool2_fusable(const cfd_conv_desc *desc);
int cfd_conv_forward_pool2(const cfd_conv_desc *desc, const void *x, const void *w, const float *bias, void *y,
                           void *stream);

/* The training form of the same fusion: y is the unpooled activation [B][Hi][Wi][Cout] (channel stride desc->ldy; the
 * pool's backward needs it) AND y_pooled [B][Hi/2][Wi/2][Cout] (channel stride ld_pooled) its MaxPooling2D((2,2)), both
 * written from the one staged output tile -- the separate cfd_maxpool2 launch re-read the whole activation
 * (DeltaP/models/generator.py:31-35 inside DeltaP/train.py:131's train_on_batch).  bf16 only; shapes with
 * cfd_conv_pool2_keep_fusable() != 0 (those of cfd_conv_pool2_fusable that the patch-staged kernel takes). */
int cfd_conv_pool2_keep_fusable(const cfd_conv_desc *desc);
int cfd_conv_forward_pool2_keep(const cfd_conv_desc *desc, const void *x, const void *w, const float *bias, void *y,
                                void *y_pooled, int ld_pooled, void *stream);

/* weight re-layout: Keras Conv2D (kh,kw,Cin,Cout) fp32 -> packed [Cout][kh][kw][Cin] of `dtype` */
int cfd_pack_conv_weight(const float *keras_hwio, void *packed, int kh, int kw, int cin, int cout, int dtype,
                         void *stream);
/* Keras Conv2DTranspose (kh,kw,Cout,Cin) fp32 -> packed [Cout][kh][kw][Cin] */
int cfd_pack_convT_weight(const float *keras_hwoi, void *packed, int kh, int kw, int cin, int cout, int dtype,
                          void *stream);
/* Keras Dense (in,out) fp32 -> packed [out][in] */
int cfd_pack_dense_weight(const float *keras_io, void *packed, int in, int out, int dtype, void *stream);

/* ------------------------------------------------------------------------------------
 * First / last layers that are not GEMM shaped (Cin <= 4 or Cout == 1): direct kernels.
 * ---------------------------------------------------------------------------------- */
/* Conv2D kxk (k=3 s1 or k=4 s2) 'same' with Cin<=4 (e1: generator.py:52, c1: discriminator.py:22,
 * g1: DeltaP generator.py:23) + bias + activation.  x: fp32 [B][Hi][Wi][Cin]; a second input x2
 * (may be NULL) is concatenated channel-wise after x (discriminator.py:20).
 * w: fp32 packed [Cout][k][k][Cin1+Cin2].  y: `out_dtype`, channel stride ldy.  If y2 != NULL the
 * same pre-activation is also written through act2 to y2 (stride ldy2): the skip copy that the
 * decoder's concat+ReLU would otherwise re-read (generator.py:38-40). */
int cfd_conv_first(const float *x, const float *x2, int cin1, int cin2, const float *w, const float *bias,
                   void *y, int ldy, void *y2, int ldy2, int B, int Hi, int Wi, int Cout, int ksize, int stride,
                   int act, int act2, float leaky_slope, int out_dtype, void *stream);

/* Conv2DTranspose 4x4 s2 'same' to ONE channel + bias (+ sigmoid) (generator.py:70-71).
 * x: `dtype` [B][Hi][Wi][Cin] stride ldx; w packed [1][4][4][Cin] `dtype`; taps: fp32 scratch
 * [B*Hi*Wi][16]; y fp32 [B][2Hi][2Wi][1].  logits (may be NULL) receives the pre-sigmoid values. */
int cfd_convT_out(const void *x, int ldx, const void *w, const float *bias, float *taps, float *y, float *logits,
                  int B, int Hi, int Wi, int Cin, int dtype, int apply_sigmoid, void *stream);

/* The same with ReLU applied to x as it is loaded (relu_on_load != 0): the inference path keeps one copy of the first
 * encoder activation -- the LeakyReLU one, written straight into the concat buffer -- and the Concatenate -> ReLU of
 * generator.py:38-40 happens here, ReLU(LeakyReLU(v)) == ReLU(v). */
int cfd_convT_out_relu(const void *x, int ldx, const void *w, const float *bias, float *taps, float *y, float *logits,
                       int B, int Hi, int Wi, int Cin, int dtype, int apply_sigmoid, int relu_on_load, void *stream);

/* The output layer with the LAST DECODER BLOCK's BatchNorm + ReLU applied as its operand is loaded (inference;
 * generator.py:33-40 for d5 followed by :70-71): x[pixel] = [ReLU(BN(raw[pixel][0..C1))) | ReLU(skip[pixel][0..C2))].
 * raw: pre-BatchNorm output of the last Conv2DTranspose, `raw_dtype` (fp32 is allowed under a bf16 compute dtype),
 * [B][Hi][Wi][C1] stride ldraw; stat_sum / stat_sq: its per-sample sums [B][C1] (cfd_conv_forward_stats or cfd_bn_stats);
 * gamma / beta / eps: the layer's BatchNorm; skip: `dtype` [B][Hi][Wi][C2] stride ldskip (the first encoder activation);
 * w packed [1][4][4][C1 + C2].  Bit-identical to cfd_bn_apply(ReLU, no dropout) into a concat buffer followed by
 * cfd_convT_out_relu -- minus one write and one read of the normalised tensor.  Hi * Wi must be a multiple of 16;
 * built for C1 = C2 in {64, 128, 256}. */
int cfd_convT_out_bn(const void *raw, int ldraw, int raw_dtype, const float *stat_sum, const float *stat_sq,
                     const float *gamma, const float *beta, float eps, const void *skip, int ldskip, const void *w,
                     const float *bias, float *taps, float *y, float *logits, int B, int Hi, int Wi, int C1, int C2,
                     int dtype, int apply_sigmoid, void *stream);

/* Conv2D 3x3 'same' on ONE input channel + bias + activation + MaxPooling2D(2,2) in one pass (inference path of
 * DeltaP/models/generator.py:23-26: g1 -> LeakyReLU -> pool).  x fp32 [B][Hi][Wi][1]; w fp32 packed [Cout][3][3][1];
 * y `out_dtype` dense [B][Hi/2][Wi/2][Cout].  act must be monotone (NONE / LEAKY with slope >= 0 / RELU): the maximum is
 * taken over the four pre-activations, which equals pooling the activations exactly.  The full-resolution activation
 * is never written (training keeps cfd_conv_first + cfd_maxpool2: the pool's backward needs it). */
int cfd_conv3_pool2(const float *x, const float *w, const float *bias, void *y, int B, int Hi, int Wi, int Cout, int act,
                    float leaky_slope, int out_dtype, void *stream);

/* ------------------------------------------------------------------------------------
 * Fused bandwidth-bound ops.
 * ---------------------------------------------------------------------------------- */
/* BatchNormalization(training) + activation, statistics per sample over (H,W)
 * (generator.py:21-23 encoder: BN -> LeakyReLU; generator.py:33-40 decoder:
 * BN -> Dropout -> Concatenate -> ReLU; discriminator.py:26-27).
 *   x        `x_dtype` [B][HW][C] stride ldx (conv output incl. bias); fp32 x with bf16 y is allowed
 *   gamma,beta fp32 [C];  eps = 1e-3
 *   use_moving != 0: normalise with mean/var given in `mean`,`var` fp32 [C] (inference-mode BN,
 *                    D inside gan.test_on_batch); otherwise batch statistics are computed and
 *                    written to mean/var fp32 [B][C] (may be NULL)
 *   dropout (Dropout(0.5)(g, training=True), generator.py:36): drop_rate == 0 -> none;
 *            drop_mask != NULL -> explicit uint8 [B][HW][C] keep-mask; otherwise a counter-based RNG
 *            keyed by (drop_seed, global element index with the batch starting at sample `sample0`),
 *            so the draw is independent of batch split / world size.  Kept values * 1/(1-rate).
 *   y        `dtype` stride ldy, through `act`;  y2 (may be NULL) the same through act2 */
int cfd_bn_act(const void *x, int ldx, int x_dtype, const float *gamma, const float *beta, float *mean, float *var,
               int use_moving, const uint8_t *drop_mask, float drop_rate, uint64_t drop_seed, int64_t sample0,
               void *y, int ldy, void *y2, int ldy2, int B, int HW, int C, float eps, int act, int act2,
               float leaky_slope, int dtype, void *stream);
/* cfd_bn_act for a conv output whose statistics came from cfd_conv_forward_stats: one coalesced read of x.
 * stat_sum / stat_sq fp32 [B][C]; mean / var (may be NULL) receive the per-sample statistics [B][C] that
 * cfd_bn_act_bwd needs.  Same dropout / activation arguments as cfd_bn_act.  C in {64,128,...,2048}. */
int cfd_bn_apply(const void *x, int ldx, int x_dtype, const float *stat_sum, const float *stat_sq, const float *gamma,
                 const float *beta, float *mean, float *var, const uint8_t *drop_mask, float drop_rate,
                 uint64_t drop_seed, int64_t sample0, void *y, int ldy, void *y2, int ldy2, int B, int HW, int C,
                 float eps, int act, int act2, float leaky_slope, int dtype, void *stream);
/* Per-sample statistics for cfd_bn_apply when the producing conv could not fuse them (split-K): accumulates
 * sum / sum of squares of x [B][HW][C] into stat_sum / stat_sq fp32 [B][C] (zeroed by the caller), pixel-parallel.
 * Replaces the first pass of BatchNormalization(training=True) (generator.py:20,33).  C in {64,128,...,2048}. */
int cfd_bn_stats(const void *x, int ldx, int x_dtype, float *stat_sum, float *stat_sq, int B, int HW, int C, void *stream);
/* Device-resident dropout epoch (no counterpart in the reference): forward kernels in RNG mode whose drop_seed has
 * bit 63 set use seed + epoch * odd_constant.  set != 0: epoch = value; else epoch += value.  Lets a captured HIP
 * graph (whose kernel arguments are frozen) draw fresh dropout per replay.  Seeds below 2^63 ignore the epoch, so a
 * training step's backward always reproduces its forward mask; cfd_bn_act_bwd refuses bit-63 seeds. */
int cfd_dropout_epoch(uint64_t value, int set, void *stream);
/* the keep-mask (0/1 bytes, [B][HW][C]) that cfd_bn_act's RNG mode applies for the same arguments */
int cfd_dropout_mask(uint8_t *mask, int64_t sample0, int B, int HW, int C, float rate, uint64_t seed, void *stream);

/* MaxPooling2D(2,2) (DeltaP generator.py:26,32,35).  x,y `dtype` NHWC dense. */
int cfd_maxpool2(const void *x, void *y, int B, int Hi, int Wi, int C, int dtype, void *stream);

/* dtype conversion / strided channel copy: y[p][0..C) = convert(x[p][0..C)) */
int cfd_cast_copy(const void *x, int ldx, int x_dtype, void *y, int ldy, int y_dtype, int64_t pixels, int C,
                  void *stream);

/* ------------------------------------------------------------------------------------
 * Training step: what Keras runs inside train_on_batch (U/train.py:121-125, P/train.py:133-137,
 * DeltaP/train.py:131) for the graphs above -- backward of every layer, losses, Adam.
 * Data gradients of the GEMM-shaped layers reuse cfd_conv_forward on transposed weights:
 *   dX of CONV_K4S2  = CONVT_K4S2(dY, Wt)      dX of CONVT_K4S2 = CONV_K4S2(dY, Wt)
 *   dX of CONV_K3S1  = CONV_K3S1(dY, Wt flipped)   dX of DENSE = DENSE(dY, Wt)
 *   dX of CONV_K4S1  = CFD_CONV_K4S1_BWD(dY, Wt flipped)   (4x4 stride 1, pad 2 before / 1 after)
 * with Wt[ci][tap'][co] = W[co][tap][ci] from cfd_transpose_weight.
 * ---------------------------------------------------------------------------------- */
#define CFD_CONV_K4S1_BWD 5

/* dW[Cout][taps][Cin] (fp32) = sum over pixels of dY (x) gathered X, on MFMA; `desc` describes the FORWARD
 * layer (x = its input, dy = gradient of its output, dtype of both = desc->dtype).  split_m == 1 stores,
 * > 1 accumulates with atomics into a buffer the caller zeroed. */
int cfd_conv_wgrad(const cfd_conv_desc *desc, const void *x, const void *dy, float *dw, int split_m, void *stream);
int cfd_conv_wgrad_suggest_split(const cfd_conv_desc *desc);
/* Same, with scratch for split_m > 1: when cfd_conv_wgrad_workspace_bytes(desc, split_m) > 0 and `workspace` holds
 * that many bytes, the partial tiles are stored and summed in a fixed order -- dw is overwritten (no zeroing
 * needed), the result is bitwise reproducible and no atomics are issued.  Otherwise identical to cfd_conv_wgrad. */
int64_t cfd_conv_wgrad_workspace_bytes(const cfd_conv_desc *desc, int split_m);
int cfd_conv_wgrad_ws(const cfd_conv_desc *desc, const void *x, const void *dy, float *dw, int split_m,
                      void *workspace, void *stream);
/* dst[Cin][taps][Cout] = src[Cout][taps][Cin], taps reversed when flip != 0 */
int cfd_transpose_weight(const void *src, void *dst, int Cout, int taps, int Cin, int flip, int dtype, void *stream);
/* out[c] += sum_m x[m][c]: bias gradients (out must be zeroed by the caller) */
int cfd_colsum(const void *x, int ldx, float *out, int64_t M, int C, int dtype, void *stream);

/* Backward of cfd_bn_act (BatchNorm batch statistics per sample + dropout + activation(s)):
 *   g1 = dL/dy, g2 = dL/dy2 (may be NULL);  dx = dL/dx (`dtype`, may be NULL);  dgamma/dbeta fp32 [C] are
 *   ACCUMULATED (atomics; zero them first).  mean/var: the fp32 [B][C] statistics the forward wrote. */
int cfd_bn_act_bwd(const void *x, int ldx, int x_dtype, const float *mean, const float *var, const float *gamma,
                   const float *beta, const void *g1, int ldg1, const void *g2, int ldg2, const uint8_t *drop_mask,
                   float drop_rate, uint64_t drop_seed, int64_t sample0, void *dx, int lddx, float *dgamma,
                   float *dbeta, int B, int HW, int C, float eps, int act1, int act2, float leaky_slope, int dtype,
                   void *stream);
/* Same, with scratch: `sums` fp32 [2][B][C], ZEROED by the caller (may be NULL = cfd_bn_act_bwd).  Large tensors
 * then run as two pixel-parallel, fully coalesced passes (partial sums by atomics into `sums`, then dx). */
int cfd_bn_act_bwd_ws(const void *x, int ldx, int x_dtype, const float *mean, const float *var, const float *gamma,
                      const float *beta, const void *g1, int ldg1, const void *g2, int ldg2, const uint8_t *drop_mask,
                      float drop_rate, uint64_t drop_seed, int64_t sample0, void *dx, int lddx, float *dgamma,
                      float *dbeta, int B, int HW, int C, float eps, int act1, int act2, float leaky_slope, int dtype,
                      float *sums, void *stream);
/* layers without BN: dpre = g1*act1'(y) + g2*act2'(y), y = the saved activation (its sign decides) */
int cfd_act_bwd(const void *y, int ldy, const void *g1, int ldg1, const void *g2, int ldg2, void *dx, int lddx,
                int64_t pixels, int C, int act1, int act2, float leaky_slope, int dtype, void *stream);
/* cfd_act_bwd that also accumulates (atomics; the caller zeroes it) the bias gradient of the layer whose pre-activation
 * gradient it writes: dbias[c] += sum over pixels of dx[p][c], summed over the values as stored -- what a separate
 * cfd_colsum pass over dx gives, without re-reading dx (Conv2D / Dense bias of the layers of
 * DeltaP/models/generator.py:28-46 and the generators' bottleneck, U/models/generator.py:59-60).  C / 8 must divide 256. */
int cfd_act_bwd_dbias(const void *y, int ldy, const void *g1, int ldg1, const void *g2, int ldg2, void *dx, int lddx,
                      int64_t pixels, int C, int act1, int act2, float leaky_slope, int dtype, float *dbias, void *stream);
/* MaxPooling2D(2,2) backward fused with the activation backward of the layer that fed the pool;
 * x = that layer's activation [B][Hi][Wi][C], dy = gradient of the pooled tensor, dx = gradient of its pre-activation */
int cfd_maxpool2_act_bwd(const void *x, const void *dy, void *dx, int B, int Hi, int Wi, int C, int act,
                         float leaky_slope, int dtype, void *stream);
/* the same, also accumulating that layer's bias gradient (see cfd_act_bwd_dbias) */
int cfd_maxpool2_act_bwd_dbias(const void *x, const void *dy, void *dx, int B, int Hi, int Wi, int C, int act,
                               float leaky_slope, int dtype, float *dbias, void *stream);
/* Dropout(rate) on a dense tensor (fc1 output, DeltaP/models/generator.py:42) and on its gradient */
int cfd_dropout_apply(const void *x, void *y, const uint8_t *mask, int64_t total, float rate, uint64_t seed,
                      int64_t elem0, int dtype, void *stream);

/* Losses, mean over all n elements, with gradients.  loss (fp32 scalar, device) is ACCUMULATED.
 * binary_crossentropy on the logits of a sigmoid output (discriminator.py:42,48; generator.py:93):
 *   target = target[i] or the constant; dz = grad_scale*(sigmoid(z)-t)/n (dz may be NULL) */
int cfd_bce_logits(const float *z, const float *target, float target_const, int64_t n, float *loss, float *dz,
                   float grad_scale, void *stream);
/* kind 0: MAE ('mae' / MC_mae, loss.py:47-88, whose mass-conservation term is identically 0), 1: MSE
 * (MC_mse, loss.py:4-45), 2: Huber(delta) (DeltaP generator.py:53).  dout = (weight*l'(y-t)/n + dy_extra) *
 * (sigmoid_grad ? y(1-y) : 1): the gradient w.r.t. the logits of a sigmoid output (or w.r.t. y). */
int cfd_field_loss(const float *y, const float *target, int64_t n, int kind, float delta, float weight, float *loss,
                   const float *dy_extra, float *dout, int sigmoid_grad, void *stream);

/* Adam, Keras form (epsilon outside the sqrt): m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= alpha*m/(sqrt(v)+eps),
 * g scaled by grad_scale first (1/world after an all-reduce SUM).  alpha = lr*sqrt(1-b2^t)/(1-b1^t) from the host.
 * w_compute (may be NULL): the bf16 shadow copy of w, refreshed in the same pass. */
int cfd_adam_step(float *w, const float *g, float *m, float *v, void *w_compute, int compute_dtype, int64_t n,
                  float alpha, float beta1, float beta2, float eps, float grad_scale, void *stream);
/* moving = moving*momentum + batch*(1-momentum), once per sample in order (BatchNormalization momentum 0.99) */
int cfd_bn_update_moving(float *moving_mean, float *moving_var, const float *mean, const float *var, int B, int C,
                         float momentum, void *stream);

/* cfd_bn_update_moving for up to 16 BatchNorm layers in one launch (the discriminator's four per update).  The arrays
 * are HOST arrays of `nlayers` device pointers / channel counts. */
int cfd_bn_update_moving_multi(int nlayers, float *const *moving_mean, float *const *moving_var,
                               const float *const *mean, const float *const *var, const int *C, int B, float momentum,
                               void *stream);
/* base[off .. off+cnt) = 0 for each of the n (off, cnt) pairs in `ranges` (DEVICE int64 [n][2]; max_count = the largest
 * cnt, sizes the grid): what a train step must clear before its backward -- the gradient tensors that kernels
 * accumulate into (biases, BatchNorm gamma/beta, first/last layers); the GEMM weight gradients are stored, not added
 * (no counterpart in the reference; Keras zeroes nothing because it owns its gradient buffers). */
int cfd_zero_ranges(float *base, const int64_t *ranges, int n, int64_t max_count, void *stream);

/* One-output-channel layers (generator `out` Conv2DTranspose, discriminator c6 Conv2D): backward.
 * geom 0: Conv2DTranspose k4 s2; geom 1: Conv2D k4 s1 'same'.  dtaps fp32 [B*Hi*Wi][16]. */
int cfd_tap_scatter_grad(const float *dz, float *dtaps, int B, int Hi, int Wi, int geom, void *stream);
/* dx[row][c] = sum_t d[row][t]*w[t][c]  (ntaps = 16, or 1 for the Dense(1) layer fc3) */
int cfd_tap_dx(const float *d, const void *w, void *dx, int lddx, int64_t rows, int C, int ntaps, int dtype,
               void *stream);
/* dw[t][c] += sum_rows d[row][t]*x[row][c]  (fp32, accumulated) */
int cfd_tap_dw(const float *d, const void *x, int ldx, float *dw, int64_t rows, int C, int ntaps, int dtype,
               void *stream);
/* first layers (Cin <= 2): dw[Cout][k][k][Cin] and db[Cout] accumulated from dpre (`dtype`, stride ldd) */
int cfd_conv_first_wgrad(const float *x, const float *x2, int cin1, int cin2, const void *dpre, int ldd, float *dw,
                         float *db, int B, int Hi, int Wi, int Cout, int ksize, int stride, int dtype, void *stream);

/* ------------------------------------------------------------------------------------
 * Nearest-neighbour resample + wall mask, fp64, bit-exact.
 * Replaces process_csv_with_pixel_coordinates_nearest, pixels_coords_testing.py:151-227
 * (scipy griddata(method='nearest') :214-215, wall detection :188-211, mask :218-221,
 * p_convert :225).  Ties between equidistant source points resolve to the lowest index.
 * The points of each case are binned into a cell grid on the device and every pixel searches
 * rings of cells with an exact (rounded-arithmetic) stop test: the argmin of the full scan, at
 * ~20-50 distance evaluations per pixel instead of N.
 *   pts          fp64 [total_points][5] rows x,y,z,p,Umean, cases concatenated
 *   case_offsets int64 [ncase+1] (device) row offsets of each case in pts; case_offsets[0] = 0,
 *                case_offsets[ncase] = total_points (the host passes that number too: grids are sized from it)
 *   y_final[H], z_final[W], z_samples[S] fp64 (device): pixel centres (:167-168) and slice
 *                positions (:188, ASCENDING as np.linspace gives them), computed by the host exactly as numpy does
 *   out          fp64 [ncase][H*W][5] rows y_center,z_center,p_center,Umean_center,p_convert
 *   workspace    cfd_resample_workspace_bytes(ncase, W, S, total_points) bytes; its first
 *                ncase * (2S + 2W + 2) doubles hold, per case, the slice minima / maxima [S][S], the wall
 *                interpolated to the columns [W][W], and (as an int) the number of valid slices -- 0 means
 *                np.interp would have raised (:210)
 * ---------------------------------------------------------------------------------- */
int64_t cfd_resample_workspace_bytes(int ncase, int W, int S, int64_t total_points);
int cfd_resample_nearest(const double *pts, const int64_t *case_offsets, int ncase, int64_t total_points,
                         const double *y_final, int H, const double *z_final, int W, const double *z_samples, int S,
                         double tol1, double tol2, double p_convert, double *out, void *workspace, void *stream);
/* The same call, additionally adding the number of fp64 distance evaluations its search performed (one = two subtractions,
 * two products, one sum) to *n_evals (device, zeroed by the caller).  Measurement aid for the FP64-rate roofline of the
 * search kernel (pixels_coords_testing.py:214-215 costs one KD-tree query per pixel; this path's cost is these
 * evaluations); same results as cfd_resample_nearest. */
int cfd_resample_nearest_counted(const double *pts, const int64_t *case_offsets, int ncase, int64_t total_points,
                                 const double *y_final, int H, const double *z_final, int W, const double *z_samples,
                                 int S, double tol1, double tol2, double p_convert, double *out, void *workspace,
                                 unsigned long long *n_evals, void *stream);

/* cfd_resample_nearest WITHOUT a workspace argument and without any allocation.  No counterpart in the reference: it
 * exists for callers that cannot allocate device memory -- code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  The scratch is a fixed device array inside the library (cfd_resample_noalloc_max_bytes() =
 * 6 MiB per GPU, held until the process ends; 64 cases x 1024 points on a 64x512 grid with 101 slices need 3.7 MiB):
 * a call whose cfd_resample_workspace_bytes(...) exceeds it is refused with CFD_EINVAL and a message naming both
 * numbers.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU RACE -- serialise them.
 *   valid_slices  int [ncase] (device, may be NULL): per case the number of valid wall slices (0: np.interp would
 *                 have raised, :210) -- what the workspace's status words tell the caller of cfd_resample_nearest
 * All other arguments as cfd_resample_nearest; same kernels, same results bit for bit. */
int64_t cfd_resample_noalloc_max_bytes(void);
int cfd_resample_nearest_noalloc(const double *pts, const int64_t *case_offsets, int ncase, int64_t total_points,
                                 const double *y_final, int H, const double *z_final, int W, const double *z_samples,
                                 int S, double tol1, double tol2, double p_convert, double *out, int *valid_slices,
                                 void *stream);

/* Host-side text form of the resample's result table (no device work, no stream).
 * Replaces `df_out.to_csv(output_csv, index=False)`, pixels_coords_testing.py:223-226 (pandas' writer: every fp64 as
 * repr(float) = shortest round-trip decimal in CPython's layout, NaN as the empty field).  SAME BYTES as pandas, without
 * the header line; ~30x faster, and the preprocess of a folder is otherwise nothing but this call.
 *   rows   HOST fp64 [n_rows][n_cols], row-major
 *   out    HOST char buffer of `cap` >= n_rows * n_cols * 26 bytes
 * Returns the number of bytes written (>= 0), or a negative cfd_status. */
int64_t cfd_csv_format_rows(const double *rows, int64_t n_rows, int n_cols, char *out, int64_t cap);
/* HOST-side reader for the ingest: two named columns of a numeric CSV as fp64, each field parsed as Python's float() parses
 * it -- /root/reference utils.py:72-83 (`csv.DictReader`, `float(row['p_convert'])`, `float(row['Umean_center'])`); both are
 * correctly rounded, so the doubles are the same bit for bit.  Runs without Python's GIL (pandas' round-trip parser holds it:
 * 0.8 s a case from 16 threads).
 *   text/len  the file's bytes (HOST);  name_a, name_b  column names (NUL-terminated);  out_a, out_b  HOST fp64 [cap]
 * Returns the number of data rows (>= 0); -2 when the text holds anything this plain parser does not vouch for (quotes, '_'
 * separators, hex floats, empty / malformed fields, ragged rows, a missing column, more than `cap` rows): the caller then
 * uses a general CSV reader; another negative cfd_status on bad arguments. */
int64_t cfd_csv_parse_columns(const char *text, int64_t len, const char *name_a, const char *name_b, double *out_a,
                              double *out_b, int64_t cap);

/* ------------------------------------------------------------------------------------
 * Dataset build: what utils.load_data computes per case after parsing the CSV (utils.py:89-103).
 *   p            fp64 [ncase][H][W]: the `p_convert` column reshaped (device)
 *   sdf          fp64 [ncase][H][W]: skfmm.distance(pts, dx = cell) with pts = (p > 0 ? -1 : +1) and cell = 4/H
 *                (utils.py:100-103): the model input.  scikit-fmm's fast-marching scheme, order 2, no mask (frozen band by
 *                per-axis linear interpolation combined harmonically, heap-ordered upwind updates); scikit-fmm itself is
 *                absent here, so the scheme is restated from its published source (PARITY UNPINNED) and checked bit for
 *                bit against the independent CPU restatement oracle/sdf_fmm.c.  One thread marches one case.
 *   delta_p      fp64 [ncase]: mean(p[:,0]) - mean(p[:,-1]) in numpy's summation order (utils.py:97)
 *   flags        int  [ncase]: bit 0 = some cell is fluid, bit 1 = some cell is solid; a case with (flags & 3) != 3 has
 *                no zero contour (skfmm raises ValueError there; the host mirror does too)
 *   workspace    cfd_dataset_workspace_bytes(ncase, H, W) bytes (17 B per cell: heap keys, heap cells, heap positions,
 *                cell states)
 * ---------------------------------------------------------------------------------- */
int64_t cfd_dataset_workspace_bytes(int ncase, int H, int W);
int cfd_dataset_grids(const double *p, int ncase, int H, int W, double cell, double *sdf, double *delta_p, int *flags,
                      void *workspace, void *stream);

/* ------------------------------------------------------------------------------------
 * Evaluation metrics of the valid / test drivers (U/valid.py:86-111): per case, from the prediction y = pred * scale and
 * the CFD field Y (both fp32 [ncase][H][W], device):
 *   out[c][0] = max |Y - y|                        (:100-101)
 *   out[c][1] = |Y - y| / Y * 100 at the arg-max pixel, first occurrence in row-major order, fp32 arithmetic (:103-105)
 *   out[c][2] = sum |Y - y| / (H * W)              (:96-97)
 *   out[c][3] = mean over rows with Y[i] != 0 of the cosine similarity of y[i] and Y[i]; NaN if there is none (:107-111)
 *   colmax    fp32 [ncase][W] (may be NULL) = max over rows of |Y - y| per column (:100, the error surface of :119-139)
 * ---------------------------------------------------------------------------------- */
int cfd_eval_metrics(const float *pred, float scale, const float *target, int ncase, int H, int W, double *out,
                     float *colmax, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CFD_AMD_H */
