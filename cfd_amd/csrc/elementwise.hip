// Fused bandwidth-bound kernels: BatchNorm(batch statistics per sample) + dropout + activation with
// optional second (skip/concat) output, 2x2 max-pool, strided cast/copy.
//
//   encoder  BN(training=True) -> LeakyReLU(0.2)                      U/models/generator.py:20-23
//   decoder  BN(training=True) -> Dropout(0.5) -> Concatenate -> ReLU U/models/generator.py:33-40
//            (the kernel writes its half of the concat buffer in place; the skip half was written by
//             the encoder's kernel through act2 = ReLU, since ReLU(LeakyReLU(x)) == ReLU(x))
//   discriminator BN -> LeakyReLU                                     U/models/discriminator.py:26-39
//   MaxPooling2D(2,2)                                                 DeltaP/models/generator.py:26,32,35
#include "common.h"

namespace cfd {

template <typename T> struct Vec;
template <> struct Vec<bf16> { typedef bf16x8 type; static constexpr int N = 8; };
template <> struct Vec<float> { typedef f32x4 type; static constexpr int N = 4; };

// Block = (sample b, slab of CS channels = 128 bytes per pixel).  TPP threads cover one pixel's slab with
// 16-B vectors; 256/TPP pixel phases stride over the HW pixels.  Pass 1: shifted sums -> mean, biased var.
// Pass 2: normalise, dropout, activation(s), store.
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const T *__restrict__ x, int ldx, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, float *__restrict__ mean_io,
                                                    float *__restrict__ var_io, int use_moving,
                                                    const uint8_t *__restrict__ mask, float drop_scale,
                                                    T *__restrict__ y, int ldy, T *__restrict__ y2, int ldy2, int HW,
                                                    int C, float eps, int act, int act2, float slope) {
  typedef typename Vec<T>::type V;
  constexpr int VN = Vec<T>::N;
  constexpr int TPP = 8;          // 8 x 16 B = 128 B slab per pixel
  constexpr int CS = TPP * VN;    // channels per slab
  constexpr int PH = 256 / TPP;   // pixel phases
  __shared__ float red[2][PH][CS];
  __shared__ float s_scale[CS], s_shift[CS];

  const int slabs = (C + CS - 1) / CS;
  const int b = blockIdx.x / slabs;
  const int c0 = (blockIdx.x % slabs) * CS;
  const int tid = threadIdx.x;
  const int cv = (tid % TPP) * VN;  // first channel of my vector inside the slab
  const int ph = tid / TPP;
  const bool cvalid = c0 + cv < C;  // C is a multiple of VN (checked on the host)
  const T *xb = x + (size_t)b * HW * ldx + c0 + cv;

  if (!use_moving) {
    float sh[VN], s1[VN], s2[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) { sh[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
    if (cvalid) {
      const V v0 = *(const V *)xb;  // pixel 0 as the shift: removes the mean from the sums of squares
#pragma unroll
      for (int i = 0; i < VN; ++i) sh[i] = to_f32(v0[i]);
      for (int p = ph; p < HW; p += PH) {
        const V v = *(const V *)(xb + (size_t)p * ldx);
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          const float d = to_f32(v[i]) - sh[i];
          s1[i] += d;
          s2[i] = fmaf(d, d, s2[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) { red[0][ph][cv + i] = s1[i]; red[1][ph][cv + i] = s2[i]; }
    __syncthreads();
    if (tid < CS) {
      float a = 0.f, q = 0.f;
      for (int k = 0; k < PH; ++k) { a += red[0][k][tid]; q += red[1][k][tid]; }
      const int c = c0 + tid;
      if (c < C) {
        const float shc = to_f32(x[(size_t)b * HW * ldx + c]);
        const float inv_n = 1.f / (float)HW;
        const float dm = a * inv_n;
        const float mean = shc + dm;
        float var = q * inv_n - dm * dm;
        var = var > 0.f ? var : 0.f;
        const float sc = gamma[c] * rsqrtf(var + eps);
        s_scale[tid] = sc;
        s_shift[tid] = beta[c] - mean * sc;
        if (mean_io) mean_io[(size_t)b * C + c] = mean;
        if (var_io) var_io[(size_t)b * C + c] = var;
      }
    }
  } else {
    if (tid < CS && c0 + tid < C) {
      const int c = c0 + tid;
      const float sc = gamma[c] * rsqrtf(var_io[c] + eps);
      s_scale[tid] = sc;
      s_shift[tid] = beta[c] - mean_io[c] * sc;
    }
  }
  __syncthreads();
  if (!cvalid) return;
  float sc[VN], sf[VN];
#pragma unroll
  for (int i = 0; i < VN; ++i) { sc[i] = s_scale[cv + i]; sf[i] = s_shift[cv + i]; }
  T *yb = y + (size_t)b * HW * ldy + c0 + cv;
  T *y2b = y2 ? y2 + (size_t)b * HW * ldy2 + c0 + cv : nullptr;
  const uint8_t *mb = mask ? mask + (size_t)b * HW * C + c0 + cv : nullptr;
  for (int p = ph; p < HW; p += PH) {
    const V v = *(const V *)(xb + (size_t)p * ldx);
    float f[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) f[i] = fmaf(to_f32(v[i]), sc[i], sf[i]);
    if (mb) {
      const uint8_t *mp = mb + (size_t)p * C;
#pragma unroll
      for (int i = 0; i < VN; ++i) f[i] = mp[i] ? f[i] * drop_scale : 0.f;
    }
    V o;
#pragma unroll
    for (int i = 0; i < VN; ++i) o[i] = from_f32<T>(apply_act(f[i], act, slope));
    *(V *)(yb + (size_t)p * ldy) = o;
    if (y2b) {
#pragma unroll
      for (int i = 0; i < VN; ++i) o[i] = from_f32<T>(apply_act(f[i], act2, slope));
      *(V *)(y2b + (size_t)p * ldy2) = o;
    }
  }
}

template <typename T>
__global__ void maxpool2_kernel(const T *__restrict__ x, T *__restrict__ y, int B, int Hi, int Wi, int C) {
  typedef typename Vec<T>::type V;
  constexpr int VN = Vec<T>::N;
  const int Ho = Hi / 2, Wo = Wi / 2, CV = C / VN;
  const int64_t total = (int64_t)B * Ho * Wo * CV;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % CV);
    const int64_t op = idx / CV;
    const int ox = (int)(op % Wo);
    const int oy = (int)((op / Wo) % Ho);
    const int b = (int)(op / ((int64_t)Wo * Ho));
    const T *p = x + (((size_t)b * Hi + 2 * oy) * Wi + 2 * ox) * C + cv * VN;
    const V a = *(const V *)p, bq = *(const V *)(p + C), c = *(const V *)(p + (size_t)Wi * C),
            d = *(const V *)(p + (size_t)Wi * C + C);
    V o;
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      const float m = fmaxf(fmaxf(to_f32(a[i]), to_f32(bq[i])), fmaxf(to_f32(c[i]), to_f32(d[i])));
      o[i] = from_f32<T>(m);
    }
    *(V *)(y + (size_t)op * C + cv * VN) = o;
  }
}

template <typename TI, typename TO>
__global__ void cast_copy_kernel(const TI *__restrict__ x, int ldx, TO *__restrict__ y, int ldy, int64_t pixels,
                                 int C) {
  const int64_t total = pixels * C;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / C;
    const int c = (int)(idx - p * C);
    y[p * ldy + c] = from_f32<TO>(to_f32(x[p * ldx + c]));
  }
}

}  // namespace cfd

using namespace cfd;

extern "C" int cfd_bn_act(const void *x, int ldx, const float *gamma, const float *beta, float *mean, float *var,
                          int use_moving, const uint8_t *drop_mask, float drop_scale, void *y, int ldy, void *y2,
                          int ldy2, int B, int HW, int C, float eps, int act, int act2, float slope, int dtype,
                          void *stream) {
  CFD_REQUIRE(x && gamma && beta && y, "NULL pointer");
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  CFD_REQUIRE(!use_moving || (mean && var), "use_moving needs mean and var");
  const int vn = dtype == CFD_BF16 ? 8 : 4;
  CFD_REQUIRE(C % vn == 0 && ldx % vn == 0 && ldy % vn == 0 && (!y2 || ldy2 % vn == 0),
              "C/ldx/ldy must be multiples of %d", vn);
  CFD_REQUIRE(B > 0 && HW > 0, "bad shape");
  const int cs = 8 * vn;
  const int blocks = B * ((C + cs - 1) / cs);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CFD_BF16)
    hipLaunchKernelGGL(bn_act_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16 *)x, ldx, gamma, beta, mean,
                       var, use_moving, drop_mask, drop_scale, (bf16 *)y, ldy, (bf16 *)y2, ldy2, HW, C, eps, act, act2,
                       slope);
  else
    hipLaunchKernelGGL(bn_act_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)x, ldx, gamma, beta, mean,
                       var, use_moving, drop_mask, drop_scale, (float *)y, ldy, (float *)y2, ldy2, HW, C, eps, act,
                       act2, slope);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_maxpool2(const void *x, void *y, int B, int Hi, int Wi, int C, int dtype, void *stream) {
  CFD_REQUIRE(x && y, "NULL pointer");
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  const int vn = dtype == CFD_BF16 ? 8 : 4;
  CFD_REQUIRE(Hi % 2 == 0 && Wi % 2 == 0 && C % vn == 0, "maxpool2: bad shape %dx%dx%d", Hi, Wi, C);
  const int64_t total = (int64_t)B * (Hi / 2) * (Wi / 2) * (C / vn);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CFD_BF16)
    hipLaunchKernelGGL(maxpool2_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16 *)x, (bf16 *)y, B, Hi, Wi, C);
  else
    hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float *)x, (float *)y, B, Hi, Wi, C);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_cast_copy(const void *x, int ldx, int x_dtype, void *y, int ldy, int y_dtype, int64_t pixels, int C,
                             void *stream) {
  CFD_REQUIRE(x && y, "NULL pointer");
  CFD_REQUIRE(pixels > 0 && C > 0 && ldx >= C && ldy >= C, "bad shape");
  const int64_t total = pixels * C;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t st = (hipStream_t)stream;
#define CC(TI, TO) hipLaunchKernelGGL((cast_copy_kernel<TI, TO>), dim3(blocks), dim3(256), 0, st, (const TI *)x, ldx, (TO *)y, ldy, pixels, C)
  if (x_dtype == CFD_F32 && y_dtype == CFD_F32) CC(float, float);
  else if (x_dtype == CFD_F32 && y_dtype == CFD_BF16) CC(float, bf16);
  else if (x_dtype == CFD_BF16 && y_dtype == CFD_F32) CC(bf16, float);
  else if (x_dtype == CFD_BF16 && y_dtype == CFD_BF16) CC(bf16, bf16);
  else CFD_FAIL(CFD_EINVAL, "bad dtype");
#undef CC
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}
