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
/* the keep-mask (0/1 bytes, [B][HW][C]) that cfd_bn_act's RNG mode applies for the same arguments */
int cfd_dropout_mask(uint8_t *mask, int64_t sample0, int B, int HW, int C, float rate, uint64_t seed, void *stream);

/* MaxPooling2D(2,2) (DeltaP generator.py:26,32,35).  x,y `dtype` NHWC dense. */
int cfd_maxpool2(const void *x, void *y, int B, int Hi, int Wi, int C, int dtype, void *stream);

/* dtype conversion / strided channel copy: y[p][0..C) = convert(x[p][0..C)) */
int cfd_cast_copy(const void *x, int ldx, int x_dtype, void *y, int ldy, int y_dtype, int64_t pixels, int C,
                  void *stream);

/* ------------------------------------------------------------------------------------
 * Nearest-neighbour resample + wall mask, fp64, bit-exact.
 * Replaces process_csv_with_pixel_coordinates_nearest, pixels_coords_testing.py:151-227
 * (scipy griddata(method='nearest') :214-215, wall detection :188-211, mask :218-221,
 * p_convert :225).  Ties between equidistant source points resolve to the lowest index.
 *   pts          fp64 [total_points][5] rows x,y,z,p,Umean, cases concatenated
 *   case_offsets int64 [ncase+1] (device) row offsets of each case in pts
 *   y_final[H], z_final[W], z_samples[S] fp64 (device): pixel centres (:167-168) and slice
 *                positions (:188), computed by the host exactly as numpy does
 *   out          fp64 [ncase][H*W][5] rows y_center,z_center,p_center,Umean_center,p_convert
 *   workspace    cfd_resample_workspace_bytes(ncase, W, S) bytes
 * ---------------------------------------------------------------------------------- */
int64_t cfd_resample_workspace_bytes(int ncase, int W, int S);
int cfd_resample_nearest(const double *pts, const int64_t *case_offsets, int ncase, const double *y_final, int H,
                         const double *z_final, int W, const double *z_samples, int S, double tol1, double tol2,
                         double p_convert, double *out, void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CFD_AMD_H */
