// Direct (non-GEMM) kernels for the first and last layers of the generators / discriminator / CNN,
// whose channel counts (Cin <= 4 or Cout == 1) do not fill an MFMA tile, and weight re-layout.
//
//   cfd_conv_first : e1 (U/models/generator.py:52, P/...:78), c1 incl. the channel concat of
//                    (src, target) (U/models/discriminator.py:20-23), g1 (DeltaP/models/generator.py:23-24)
//   cfd_convT_out  : output Conv2DTranspose(1,(4,4),strides 2) + sigmoid (U/models/generator.py:70-71)
#include "common.h"

namespace cfd {

// ---------------------------------------------------------------------------------------------
// conv_first: tile = one output row segment of TX pixels (a block walks many tiles); the input patch (KS rows x
// ((TX-1)*S+KS) cols x CIN) is staged in LDS with zero padding (TF 'same': pad 1 before),
// each thread keeps the weights of its CPT output channels in registers.
// ---------------------------------------------------------------------------------------------
// pixels of one output row per tile: 128 amortises the patch staging and its two barriers over 16 pixels per thread
// (32 left 4 per thread between barriers)
constexpr int CONV_FIRST_TX = 128;
constexpr int WL_LD = 256 + 4;           // conv_first: Cout <= 256 (CG <= 256 / ... checked on the host)

template <int KS, int S, int CIN, int CPT, typename TO>
__global__ __launch_bounds__(256) void conv_first_kernel(const float *__restrict__ x, const float *__restrict__ x2,
                                                         int cin1, const float *__restrict__ w,
                                                         const float *__restrict__ bias, TO *__restrict__ y, int ldy,
                                                         TO *__restrict__ y2, int ldy2, int Hi, int Wi, int Ho, int Wo,
                                                         int Cout, int act, int act2, float slope, int ntiles) {
  constexpr int TX = CONV_FIRST_TX;
  constexpr int PW = (TX - 1) * S + KS;
  constexpr int NW = KS * KS * CIN;
  __shared__ float patch[KS][PW][CIN];

  const int tid = threadIdx.x;
  const int CG = Cout / CPT;           // channel groups; 256 % CG == 0 checked on the host
  const int cg = tid % CG;
  const int slot = tid / CG;
  const int pstep = 256 / CG;

  const int xtiles = (Wo + TX - 1) / TX;
  // activation kinds resolved once (a per-element switch on the run-time `act` compiled to ~6 scalar branches per
  // output element, next to its 16 FMAs)
  const float ns1 = act_neg_scale(act, slope), ns2 = act_neg_scale(act2, slope);
  const bool sig1 = act == CFD_ACT_SIGMOID, sig2 = act2 == CFD_ACT_SIGMOID;

  // weights of my channels: wreg[c][tap*CIN + ci] -- loaded ONCE per block; the block then walks many tiles
  // (a per-tile block re-read 128 KB of weights to write 32 KB of output).  They come through LDS: read straight from
  // global memory every lane's NW * CPT loads hit a different 64-B row (stride NW floats between channel groups), which
  // at small batch -- one tile per block -- cost more than the tile itself (e1 at B = 16: 75 us for 134 MB of output).
  __shared__ __align__(16) float wl[NW][WL_LD];      // [tap][channel], rows padded to keep 16-B alignment
  for (int e = tid; e < Cout * NW; e += 256) wl[e % NW][e / NW] = w[e];
  __syncthreads();
  float wreg[CPT][NW];
  float breg[CPT];
#pragma unroll
  for (int k = 0; k < NW; ++k) {
#pragma unroll
    for (int c4 = 0; c4 < CPT; c4 += 4) {
      const f32x4 t = *(const f32x4 *)&wl[k][cg * CPT + c4];
#pragma unroll
      for (int c = 0; c < 4; ++c) wreg[c4 + c][k] = t[c];
    }
  }
#pragma unroll
  for (int c = 0; c < CPT; ++c) breg[c] = bias ? bias[cg * CPT + c] : 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int xt = bid % xtiles; bid /= xtiles;
    const int oy = bid % Ho;
    const int b = bid / Ho;
    const int ox0 = xt * TX;

    // stage the patch
    const int iy0 = oy * S - 1, ix0 = ox0 * S - 1;
    __syncthreads();   // the previous tile's readers are done
    for (int e = tid; e < KS * PW * CIN; e += 256) {
      const int ci = e % CIN;
      const int px = (e / CIN) % PW;
      const int py = e / (CIN * PW);
      const int iy = iy0 + py, ix = ix0 + px;
      float v = 0.f;
      if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
        const size_t pix = ((size_t)b * Hi + iy) * Wi + ix;
        v = (ci < cin1) ? x[pix * cin1 + ci] : x2[pix * (CIN - cin1) + (ci - cin1)];
      }
      patch[py][px][ci] = v;
    }
    __syncthreads();

    for (int px = slot; px < TX; px += pstep) {
      const int ox = ox0 + px;
      if (ox >= Wo) break;
      float accv[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) accv[c] = breg[c];
#pragma unroll
      for (int kh = 0; kh < KS; ++kh)
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) {
            const float v = patch[kh][px * S + kw][ci];
#pragma unroll
            for (int c = 0; c < CPT; ++c) accv[c] = fmaf(v, wreg[c][(kh * KS + kw) * CIN + ci], accv[c]);
          }
      const size_t opix = ((size_t)b * Ho + oy) * Wo + ox;
      TO o[CPT];
      float fo[CPT];
      apply_act_vec(fo, accv, ns1, sig1);
#pragma unroll
      for (int c = 0; c < CPT; ++c) o[c] = from_f32<TO>(fo[c]);
      TO *dst = y + opix * ldy + cg * CPT;
#pragma unroll
      for (int c = 0; c < CPT; ++c) dst[c] = o[c];  // contiguous CPT elements: the compiler merges into one wide store
      if (y2) {
        apply_act_vec(fo, accv, ns2, sig2);
#pragma unroll
        for (int c = 0; c < CPT; ++c) o[c] = from_f32<TO>(fo[c]);
        TO *dst2 = y2 + opix * ldy2 + cg * CPT;
#pragma unroll
        for (int c = 0; c < CPT; ++c) dst2[c] = o[c];
      }
    }
  }
}

template <int KS, int S, int CIN, int CPT>
static int launch_conv_first(const float *x, const float *x2, int cin1, const float *w, const float *bias, void *y,
                             int ldy, void *y2, int ldy2, int B, int Hi, int Wi, int Cout, int act, int act2,
                             float slope, int out_dtype, hipStream_t st) {
  const int Ho = Hi / S, Wo = Wi / S;
  const int CG = Cout / CPT;
  CFD_REQUIRE(Cout % CPT == 0 && CG <= 256 && 256 % CG == 0, "conv_first: Cout=%d unsupported", Cout);
  const int ntiles = B * Ho * ((Wo + CONV_FIRST_TX - 1) / CONV_FIRST_TX);
  const int blocks = ntiles < 2048 ? ntiles : 2048;
  if (out_dtype == CFD_BF16)
    hipLaunchKernelGGL((conv_first_kernel<KS, S, CIN, CPT, bf16>), dim3(blocks), dim3(256), 0, st, x, x2, cin1, w, bias,
                       (bf16 *)y, ldy, (bf16 *)y2, ldy2, Hi, Wi, Ho, Wo, Cout, act, act2, slope, ntiles);
  else
    hipLaunchKernelGGL((conv_first_kernel<KS, S, CIN, CPT, float>), dim3(blocks), dim3(256), 0, st, x, x2, cin1, w,
                       bias, (float *)y, ldy, (float *)y2, ldy2, Hi, Wi, Ho, Wo, Cout, act, act2, slope, ntiles);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

// ---------------------------------------------------------------------------------------------
// conv_first (3x3 stride 1, one input channel) + activation + MaxPooling2D(2,2) in one pass: the inference path of
// DeltaP's g1 -> pool (DeltaP/models/generator.py:23-26).  The unfused pair writes the full-resolution activation
// (B x H x W x 64: 537 MB in bf16 at B = 128) and reads it straight back to keep a quarter of it; here a thread
// computes the four pre-activations under one pooled pixel, takes their maximum (LeakyReLU / ReLU / identity are
// monotone, so act(max) == max(act), exactly) and writes one value.  Training keeps the unfused pair: the pool's
// backward needs the full-resolution activation.
// Tile = TXP pooled pixels of one pooled row; patch = 4 input rows x (2 TXP + 2) columns.
// ---------------------------------------------------------------------------------------------
constexpr int CONV_POOL_TX = 64;

template <int CPT, typename TO>
__global__ __launch_bounds__(256) void conv3_pool2_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, TO *__restrict__ y, int Hi,
                                                          int Wi, int Cout, int act, float slope, int ntiles) {
  constexpr int TX = CONV_POOL_TX;
  constexpr int PW = 2 * TX + 2;
  __shared__ float patch[4][PW];
  const int tid = threadIdx.x;
  const int CG = Cout / CPT;
  const int cg = tid % CG, slot = tid / CG, pstep = 256 / CG;
  const int Hp = Hi / 2, Wp = Wi / 2;
  const int xtiles = (Wp + TX - 1) / TX;
  const float ns = act_neg_scale(act, slope);
  float wreg[CPT][9], breg[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wreg[c][k] = w[(size_t)(cg * CPT + c) * 9 + k];
    breg[c] = bias ? bias[cg * CPT + c] : 0.f;
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int xt = bid % xtiles; bid /= xtiles;
    const int py = bid % Hp;
    const int b = bid / Hp;
    const int px0 = xt * TX;
    const int iy0 = 2 * py - 1, ix0 = 2 * px0 - 1;
    __syncthreads();
    for (int e = tid; e < 4 * PW; e += 256) {
      const int cx = e % PW, cy = e / PW;
      const int iy = iy0 + cy, ix = ix0 + cx;
      float v = 0.f;
      if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) v = x[((size_t)b * Hi + iy) * Wi + ix];
      patch[cy][cx] = v;
    }
    __syncthreads();
    for (int px = slot; px < TX; px += pstep) {
      if (px0 + px >= Wp) break;
      float best[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) best[c] = -3.402823466e38f;
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float accv[CPT];
#pragma unroll
          for (int c = 0; c < CPT; ++c) accv[c] = breg[c];
          // the same accumulation order as conv_first_kernel (kh, kw ascending): identical pre-activations
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const float v = patch[dy + kh][2 * px + dx + kw];
#pragma unroll
              for (int c = 0; c < CPT; ++c) accv[c] = fmaf(v, wreg[c][kh * 3 + kw], accv[c]);
            }
#pragma unroll
          for (int c = 0; c < CPT; ++c) best[c] = accv[c] > best[c] ? accv[c] : best[c];
        }
      TO *dst = y + (((size_t)b * Hp + py) * Wp + px0 + px) * Cout + cg * CPT;
#pragma unroll
      for (int c = 0; c < CPT; ++c) dst[c] = from_f32<TO>(apply_act_linear(best[c], ns));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// convT_out, step 1: taps[pixel][t] = sum_c x[pixel][c] * W[t][c]   (t = kh*4+kw), on MFMA.
// One wave owns 16 pixels per iteration; the 16 x Cin weight matrix lives in registers.  Each lane
// loads 32 contiguous bytes of its pixel per step so a wave-load covers whole 128-B lines; MFMA sums
// over k in any order, so both operands simply use the same (permuted) k assignment.
// ---------------------------------------------------------------------------------------------
// `relu`: ReLU applied to x as it is loaded -- the inference path keeps ONE copy of the first encoder activation (the
// LeakyReLU one, inside the concat buffer) and rectifies it here, ReLU(LeakyReLU(v)) == ReLU(v) (generator.py:38-40)
__device__ __forceinline__ bf16x8 relu8(bf16x8 v) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 b = __builtin_bit_cast(s16x8, v);
  b = b & ~(b >> 15);                 // negative (sign bit set, -0 and NaN with sign included) -> +0
  return __builtin_bit_cast(bf16x8, b);
}

template <int CIN>
__global__ __launch_bounds__(256) void tapgemm_bf16_kernel(const bf16 *__restrict__ x, int ldx,
                                                          const bf16 *__restrict__ w, float *__restrict__ taps,
                                                          int npix, int relu) {
  constexpr int STEPS = CIN / 64;  // each step: 64 channels = 2 MFMA k-steps of 32
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 wf[STEPS][2];
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) wf[s][h] = *(const bf16x8 *)(w + (size_t)fr * CIN + s * 64 + fq * 16 + h * 8);
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int ntile = (npix + 15) / 16;
  for (int t = wave_global; t < ntile; t += nwaves) {
    const int pix = t * 16 + fr;
    const int pc = pix < npix ? pix : npix - 1;
    const bf16 *xp = x + (size_t)pc * ldx + fq * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 xa[STEPS][2];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      xa[s][0] = *(const bf16x8 *)(xp + s * 64);
      xa[s][1] = *(const bf16x8 *)(xp + s * 64 + 8);
      if (relu) { xa[s][0] = relu8(xa[s][0]); xa[s][1] = relu8(xa[s][1]); }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][0], xa[s][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][1], xa[s][1], acc, 0, 0, 0);
    }
    if (pix < npix) *(f32x4 *)(taps + (size_t)pix * 16 + fq * 4) = acc;  // rows (taps) fq*4..+3 of column (pixel) fr
  }
}

template <int CIN>
__global__ __launch_bounds__(256) void tapgemm_f32_kernel(const float *__restrict__ x, int ldx,
                                                         const float *__restrict__ w, float *__restrict__ taps,
                                                         int npix, int relu) {
  constexpr int STEPS = CIN / 32;  // each step: 32 channels: lane group fq supplies k = 32 s + 8 fq + e
  const int lane = threadIdx.x & 63;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 wf[STEPS][2];
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) wf[s][h] = *(const f32x4 *)(w + (size_t)fr * CIN + s * 32 + fq * 8 + h * 4);
  const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int ntile = (npix + 15) / 16;
  for (int t = wave_global; t < ntile; t += nwaves) {
    const int pix = t * 16 + fr;
    const int pc = pix < npix ? pix : npix - 1;
    const float *xp = x + (size_t)pc * ldx + fq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      f32x4 a0 = *(const f32x4 *)(xp + s * 32);
      f32x4 a1 = *(const f32x4 *)(xp + s * 32 + 4);
      if (relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[e] = a0[e] > 0.f ? a0[e] : 0.f; a1[e] = a1[e] > 0.f ? a1[e] : 0.f; }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][0][e], a0[e], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][1][e], a1[e], acc, 0, 0, 0);
    }
    if (pix < npix) *(f32x4 *)(taps + (size_t)pix * 16 + fq * 4) = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// convT_out, step 1 with the last decoder block's BatchNorm + ReLU applied AS THE OPERAND IS LOADED (inference):
//   x[pixel] = [ ReLU(BN(raw[pixel][0..C1))) | ReLU(skip[pixel][0..C2)) ]      (generator.py:33-40, 70-71)
// `raw` is the pre-BatchNorm output of the last transposed convolution, whose per-sample sums the conv epilogue (or
// cfd_bn_stats) accumulated; the normalised activation is never written and never re-read (U at B = 128: 0.5 GB each
// way).  Same arithmetic, in the same order, as cfd_bn_apply followed by tapgemm: fmaf(x, scale, shift), ReLU, rounding
// to the compute dtype, one MFMA chain over [raw | skip] -- bit-identical taps.
// Block = (sample, chunk of pixels): the sample's scale / shift sit in LDS, one wave owns 16 pixels per iteration.
// ---------------------------------------------------------------------------------------------
template <int C1>
__device__ __forceinline__ void bn_coeffs_to_lds(float *s_sc, float *s_sh, const float *__restrict__ ssum,
                                                 const float *__restrict__ ssq, const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, float eps, int b, int HW) {
  const float inv_n = 1.f / (float)HW;
  for (int c = threadIdx.x; c < C1; c += 256) {
    // (bn_apply_kernel's expressions, so that both paths normalise with the same numbers)
    const float mean = ssum[(size_t)b * C1 + c] * inv_n;
    const float v = fmaf(-mean, mean, ssq[(size_t)b * C1 + c] * inv_n);
    const float var = v > 0.f ? v : 0.f;
    const float sc = gamma[c] * rsqrtf(var + eps);
    s_sc[c] = sc;
    s_sh[c] = beta[c] - mean * sc;
  }
  __syncthreads();
}

template <typename TR> __device__ __forceinline__ void load8(const TR *p, float (&f)[8]);
template <> __device__ __forceinline__ void load8<float>(const float *p, float (&f)[8]) {
  const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16 *p, float (&f)[8]) {
  const bf16x8 a = *(const bf16x8 *)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)a[i];
}

template <int C1, int C2, typename TR>
__global__ __launch_bounds__(256) void tapgemm_bn_bf16_kernel(const TR *__restrict__ raw, int ldraw,
                                                             const float *__restrict__ ssum, const float *__restrict__ ssq,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta,
                                                             float eps, const bf16 *__restrict__ skip, int ldskip,
                                                             const bf16 *__restrict__ w, float *__restrict__ taps, int HW,
                                                             int chunks, int pix_per_chunk) {
  constexpr int CIN = C1 + C2, S1 = C1 / 64, S2 = C2 / 64;
  constexpr int G1 = S1 % 2 == 0 ? 2 : 1, G2 = S2 % 2 == 0 ? 2 : 1;   // 64-channel steps handled together
  __shared__ __align__(16) float s_sc[C1], s_sh[C1];
  const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
  bn_coeffs_to_lds<C1>(s_sc, s_sh, ssum, ssq, gamma, beta, eps, b, HW);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  bf16x8 wf[S1 + S2][2];
#pragma unroll
  for (int s = 0; s < S1 + S2; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) wf[s][h] = *(const bf16x8 *)(w + (size_t)fr * CIN + s * 64 + fq * 16 + h * 8);
  const int p0 = chunk * pix_per_chunk;
  const int p1 = p0 + pix_per_chunk < HW ? p0 + pix_per_chunk : HW;       // (host: HW % 16 == 0, pix_per_chunk % 16 == 0)
  for (int pt = p0 + wave * 16; pt < p1; pt += 64) {
    const size_t pix = (size_t)b * HW + pt + fr;
    const TR *rp = raw + pix * ldraw + fq * 16;
    const bf16 *sp = skip + pix * ldskip + fq * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // two 64-channel steps at a time, each group's loads, conversion and MFMAs kept together (sched_barrier): with every
    // load of a pixel row hoisted to the top the kernel needed 296 registers, i.e. ONE wave per SIMD for an HBM-bound loop
#pragma unroll
    for (int s0 = 0; s0 < S1; s0 += G1) {
      bf16x8 xa[G1][2];
#pragma unroll
      for (int ds = 0; ds < G1; ++ds)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int s = s0 + ds;
          float f[8];
          load8<TR>(rp + s * 64 + h * 8, f);
          const int c = s * 64 + fq * 16 + h * 8;
          const f32x4 a0 = *(const f32x4 *)&s_sc[c], a1 = *(const f32x4 *)&s_sc[c + 4];
          const f32x4 b0 = *(const f32x4 *)&s_sh[c], b1 = *(const f32x4 *)&s_sh[c + 4];
          bf16x8 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] = (bf16)apply_act_linear(fmaf(f[i], a0[i], b0[i]), 0.f);
            o[4 + i] = (bf16)apply_act_linear(fmaf(f[4 + i], a1[i], b1[i]), 0.f);
          }
          xa[ds][h] = o;
        }
#pragma unroll
      for (int ds = 0; ds < G1; ++ds) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s0 + ds][0], xa[ds][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s0 + ds][1], xa[ds][1], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s0 = 0; s0 < S2; s0 += G2) {
      bf16x8 xa[G2][2];
#pragma unroll
      for (int ds = 0; ds < G2; ++ds) {
        xa[ds][0] = relu8(*(const bf16x8 *)(sp + (s0 + ds) * 64));
        xa[ds][1] = relu8(*(const bf16x8 *)(sp + (s0 + ds) * 64 + 8));
      }
#pragma unroll
      for (int ds = 0; ds < G2; ++ds) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[S1 + s0 + ds][0], xa[ds][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[S1 + s0 + ds][1], xa[ds][1], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    *(f32x4 *)(taps + pix * 16 + fq * 4) = acc;
  }
}

template <int C1, int C2>
__global__ __launch_bounds__(256) void tapgemm_bn_f32_kernel(const float *__restrict__ raw, int ldraw,
                                                            const float *__restrict__ ssum, const float *__restrict__ ssq,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            float eps, const float *__restrict__ skip, int ldskip,
                                                            const float *__restrict__ w, float *__restrict__ taps, int HW,
                                                            int chunks, int pix_per_chunk) {
  constexpr int CIN = C1 + C2, S1 = C1 / 32, S2 = C2 / 32;
  __shared__ __align__(16) float s_sc[C1], s_sh[C1];
  const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
  bn_coeffs_to_lds<C1>(s_sc, s_sh, ssum, ssq, gamma, beta, eps, b, HW);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 wf[S1 + S2][2];
#pragma unroll
  for (int s = 0; s < S1 + S2; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) wf[s][h] = *(const f32x4 *)(w + (size_t)fr * CIN + s * 32 + fq * 8 + h * 4);
  const int p0 = chunk * pix_per_chunk;
  const int p1 = p0 + pix_per_chunk < HW ? p0 + pix_per_chunk : HW;
  for (int pt = p0 + wave * 16; pt < p1; pt += 64) {
    const size_t pix = (size_t)b * HW + pt + fr;
    const float *rp = raw + pix * ldraw + fq * 8;
    const float *sp = skip + pix * ldskip + fq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S1 + S2; ++s) {
      f32x4 a[2];
      if (s < S1) {
        const int c = s * 32 + fq * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 v = *(const f32x4 *)(rp + s * 32 + h * 4);
          const f32x4 sc = *(const f32x4 *)&s_sc[c + h * 4], sh = *(const f32x4 *)&s_sh[c + h * 4];
#pragma unroll
          for (int e = 0; e < 4; ++e) a[h][e] = apply_act_linear(fmaf(v[e], sc[e], sh[e]), 0.f);
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 v = *(const f32x4 *)(sp + (s - S1) * 32 + h * 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) a[h][e] = v[e] > 0.f ? v[e] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][0][e], a[0][e], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[s][1][e], a[1][e], acc, 0, 0, 0);
      if (s & 1) __builtin_amdgcn_sched_barrier(0);   // (keeps the loads of at most two steps in flight per wave: registers)
    }
    *(f32x4 *)(taps + pix * 16 + fq * 4) = acc;
  }
}

// step 2: out[b][oy][ox] = bias + sum over the (<=4) (input pixel, tap) pairs with oy = 2*iy + kh - 1
__global__ void convT_out_gather_kernel(const float *__restrict__ taps, const float *__restrict__ bias,
                                        float *__restrict__ y, float *__restrict__ logits, int B, int Hi, int Wi,
                                        int apply_sigmoid) {
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int64_t total = (int64_t)B * Ho * Wo;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(idx % Wo);
    const int oy = (int)((idx / Wo) % Ho);
    const int b = (int)(idx / ((int64_t)Wo * Ho));
    float s = bias ? bias[0] : 0.f;
    // fixed summation order: kh ascending, kw ascending
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int kh = ((oy + 1) & 1) + 2 * a;
      const int iy = (oy + 1 - kh) >> 1;
      if ((unsigned)iy >= (unsigned)Hi) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int kw = ((ox + 1) & 1) + 2 * c;
        const int ix = (ox + 1 - kw) >> 1;
        if ((unsigned)ix >= (unsigned)Wi) continue;
        s += taps[(((int64_t)b * Hi + iy) * Wi + ix) * 16 + kh * 4 + kw];
      }
    }
    if (logits) logits[idx] = s;
    y[idx] = apply_sigmoid ? 1.f / (1.f + expf(-s)) : s;
  }
}

// ---------------------------------------------------------------------------------------------
// weight re-layout (Keras -> packed OHWI of the compute dtype)
// ---------------------------------------------------------------------------------------------
// src index strides are given explicitly so one kernel serves Conv2D (kh,kw,ci,co), Conv2DTranspose
// (kh,kw,co,ci) and Dense (ci,co):  packed[co][t][ci] = src[t*st + ci*sci + co*sco]
template <typename T>
__global__ void pack_weight_kernel(const float *__restrict__ src, T *__restrict__ dst, int taps, int cin, int cout,
                                   int64_t st, int64_t sci, int64_t sco) {
  const int64_t total = (int64_t)taps * cin * cout;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(idx % cin);
    const int t = (int)((idx / cin) % taps);
    const int co = (int)(idx / ((int64_t)cin * taps));
    dst[idx] = from_f32<T>(src[t * st + ci * sci + co * sco]);
  }
}

static int pack_weight(const float *src, void *dst, int taps, int cin, int cout, int64_t st, int64_t sci, int64_t sco,
                       int dtype, void *stream) {
  CFD_REQUIRE(src && dst, "NULL pointer");
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  const int64_t total = (int64_t)taps * cin * cout;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == CFD_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16 *)dst,
                       taps, cin, cout, st, sci, sco);
  else
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (float *)dst,
                       taps, cin, cout, st, sci, sco);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

}  // namespace cfd

using namespace cfd;

extern "C" int cfd_pack_conv_weight(const float *keras_hwio, void *packed, int kh, int kw, int cin, int cout,
                                    int dtype, void *stream) {
  return pack_weight(keras_hwio, packed, kh * kw, cin, cout, (int64_t)cin * cout, cout, 1, dtype, stream);
}
extern "C" int cfd_pack_convT_weight(const float *keras_hwoi, void *packed, int kh, int kw, int cin, int cout,
                                     int dtype, void *stream) {
  return pack_weight(keras_hwoi, packed, kh * kw, cin, cout, (int64_t)cin * cout, 1, cin, dtype, stream);
}
extern "C" int cfd_pack_dense_weight(const float *keras_io, void *packed, int in, int out, int dtype, void *stream) {
  return pack_weight(keras_io, packed, 1, in, out, 0, out, 1, dtype, stream);
}

extern "C" int cfd_conv_first(const float *x, const float *x2, int cin1, int cin2, const float *w, const float *bias,
                              void *y, int ldy, void *y2, int ldy2, int B, int Hi, int Wi, int Cout, int ksize,
                              int stride, int act, int act2, float slope, int out_dtype, void *stream) {
  CFD_REQUIRE(x && w && y, "NULL pointer");
  CFD_REQUIRE(cin2 == 0 || x2, "x2 is NULL but cin2=%d", cin2);
  CFD_REQUIRE(out_dtype == CFD_BF16 || out_dtype == CFD_F32, "bad out_dtype");
  CFD_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Hi % stride == 0 && Wi % stride == 0, "bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int cin = cin1 + cin2;
  if (ksize == 4 && stride == 2 && cin == 1)
    return launch_conv_first<4, 2, 1, 8>(x, x2, cin1, w, bias, y, ldy, y2, ldy2, B, Hi, Wi, Cout, act, act2, slope,
                                         out_dtype, st);
  if (ksize == 4 && stride == 2 && cin == 2)
    return launch_conv_first<4, 2, 2, 4>(x, x2, cin1, w, bias, y, ldy, y2, ldy2, B, Hi, Wi, Cout, act, act2, slope,
                                         out_dtype, st);
  if (ksize == 3 && stride == 1 && cin == 1)
    return launch_conv_first<3, 1, 1, 8>(x, x2, cin1, w, bias, y, ldy, y2, ldy2, B, Hi, Wi, Cout, act, act2, slope,
                                         out_dtype, st);
  CFD_FAIL(CFD_EUNSUPPORTED, "conv_first: k=%d s=%d cin=%d not built", ksize, stride, cin);
}

extern "C" int cfd_conv3_pool2(const float *x, const float *w, const float *bias, void *y, int B, int Hi, int Wi,
                               int Cout, int act, float leaky_slope, int out_dtype, void *stream) {
  CFD_REQUIRE(x && w && y, "NULL pointer");
  CFD_REQUIRE(out_dtype == CFD_BF16 || out_dtype == CFD_F32, "bad out_dtype");
  CFD_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && Hi % 2 == 0 && Wi % 2 == 0, "bad shape (H and W must be even)");
  CFD_REQUIRE(act == CFD_ACT_NONE || act == CFD_ACT_LEAKY || act == CFD_ACT_RELU, "activation must be monotone");
  CFD_REQUIRE(act != CFD_ACT_LEAKY || leaky_slope >= 0.f, "a negative LeakyReLU slope is not monotone");
  constexpr int CPT = 8;
  const int CG = Cout / CPT;
  CFD_REQUIRE(Cout % CPT == 0 && CG <= 256 && 256 % CG == 0, "conv3_pool2: Cout=%d unsupported", Cout);
  const int64_t nt64 = (int64_t)B * (Hi / 2) * (((Wi / 2) + CONV_POOL_TX - 1) / CONV_POOL_TX);
  CFD_REQUIRE(nt64 < (1LL << 31), "too many tiles");
  const int ntiles = (int)nt64, blocks = ntiles < 2048 ? ntiles : 2048;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == CFD_BF16)
    hipLaunchKernelGGL((conv3_pool2_kernel<CPT, bf16>), dim3(blocks), dim3(256), 0, st, x, w, bias, (bf16 *)y, Hi, Wi,
                       Cout, act, leaky_slope, ntiles);
  else
    hipLaunchKernelGGL((conv3_pool2_kernel<CPT, float>), dim3(blocks), dim3(256), 0, st, x, w, bias, (float *)y, Hi, Wi,
                       Cout, act, leaky_slope, ntiles);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_convT_out(const void *x, int ldx, const void *w, const float *bias, float *taps, float *y,
                             float *logits, int B, int Hi, int Wi, int Cin, int dtype, int apply_sigmoid,
                             void *stream) {
  return cfd_convT_out_relu(x, ldx, w, bias, taps, y, logits, B, Hi, Wi, Cin, dtype, apply_sigmoid, 0, stream);
}

extern "C" int cfd_convT_out_relu(const void *x, int ldx, const void *w, const float *bias, float *taps, float *y,
                                  float *logits, int B, int Hi, int Wi, int Cin, int dtype, int apply_sigmoid,
                                  int relu_on_load, void *stream) {
  CFD_REQUIRE(x && w && taps && y, "NULL pointer");
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  CFD_REQUIRE(ldx >= Cin && ldx % (dtype == CFD_BF16 ? 8 : 4) == 0, "bad ldx=%d", ldx);
  hipStream_t st = (hipStream_t)stream;
  const int64_t npix64 = (int64_t)B * Hi * Wi;
  CFD_REQUIRE(npix64 > 0 && npix64 < (1LL << 31), "bad pixel count");
  const int npix = (int)npix64;
  const int ntile = (npix + 15) / 16;
  int blocks = (ntile + 3) / 4;
  if (blocks > 2048) blocks = 2048;
#define TAPGEMM(C)                                                                                              \
  if (Cin == C) {                                                                                               \
    if (dtype == CFD_BF16)                                                                                      \
      hipLaunchKernelGGL(tapgemm_bf16_kernel<C>, dim3(blocks), dim3(256), 0, st, (const bf16 *)x, ldx,          \
                         (const bf16 *)w, taps, npix, relu_on_load);                                            \
    else                                                                                                        \
      hipLaunchKernelGGL(tapgemm_f32_kernel<C>, dim3(blocks), dim3(256), 0, st, (const float *)x, ldx,          \
                         (const float *)w, taps, npix, relu_on_load);                                           \
  } else
  TAPGEMM(512) TAPGEMM(256) TAPGEMM(128) TAPGEMM(64) { CFD_FAIL(CFD_EUNSUPPORTED, "convT_out: Cin=%d not built", Cin); }
#undef TAPGEMM
  CFD_LAUNCH_CHECK();
  const int64_t total = (int64_t)B * 4 * Hi * Wi;
  const int gb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(convT_out_gather_kernel, dim3(gb), dim3(256), 0, st, (const float *)taps, bias, y, logits, B, Hi,
                     Wi, apply_sigmoid);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_convT_out_bn(const void *raw, int ldraw, int raw_dtype, const float *stat_sum, const float *stat_sq,
                                const float *gamma, const float *beta, float eps, const void *skip, int ldskip,
                                const void *w, const float *bias, float *taps, float *y, float *logits, int B, int Hi,
                                int Wi, int C1, int C2, int dtype, int apply_sigmoid, void *stream) {
  CFD_REQUIRE(raw && stat_sum && stat_sq && gamma && beta && skip && w && taps && y, "NULL pointer");
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  CFD_REQUIRE(raw_dtype == dtype || (raw_dtype == CFD_F32 && dtype == CFD_BF16), "raw dtype %d with compute dtype %d", raw_dtype, dtype);
  const int va = dtype == CFD_BF16 ? 8 : 4, vr = raw_dtype == CFD_BF16 ? 8 : 4;
  CFD_REQUIRE(ldraw >= C1 && ldraw % vr == 0 && ldskip >= C2 && ldskip % va == 0, "bad ldraw=%d / ldskip=%d", ldraw, ldskip);
  CFD_REQUIRE(((uintptr_t)raw & 15) == 0 && ((uintptr_t)skip & 15) == 0 && ((uintptr_t)w & 15) == 0, "operands must be 16-B aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t hw64 = (int64_t)Hi * Wi, npix64 = (int64_t)B * hw64;
  CFD_REQUIRE(B > 0 && hw64 > 0 && npix64 < (1LL << 31) && hw64 % 16 == 0, "bad grid %d x %d x %d (H*W must be a multiple of 16)", B, Hi, Wi);
  const int HW = (int)hw64;
  // block = (sample, chunk of pixels); chunks as large as still give >= 1024 blocks (the weights are re-read per block)
  int ppc = 1024;
  while (ppc > 64 && (int64_t)B * cdiv(HW, ppc) < 1024) ppc >>= 1;
  const int chunks = cdiv(HW, ppc);
  const int blocks = B * chunks;
#define TAPGEMM_BN(A, C)                                                                                               \
  if (C1 == A && C2 == C) {                                                                                            \
    if (dtype == CFD_F32)                                                                                              \
      hipLaunchKernelGGL((tapgemm_bn_f32_kernel<A, C>), dim3(blocks), dim3(256), 0, st, (const float *)raw, ldraw,     \
                         stat_sum, stat_sq, gamma, beta, eps, (const float *)skip, ldskip, (const float *)w, taps, HW, \
                         chunks, ppc);                                                                                 \
    else if (raw_dtype == CFD_BF16)                                                                                    \
      hipLaunchKernelGGL((tapgemm_bn_bf16_kernel<A, C, bf16>), dim3(blocks), dim3(256), 0, st, (const bf16 *)raw,      \
                         ldraw, stat_sum, stat_sq, gamma, beta, eps, (const bf16 *)skip, ldskip, (const bf16 *)w,      \
                         taps, HW, chunks, ppc);                                                                       \
    else                                                                                                               \
      hipLaunchKernelGGL((tapgemm_bn_bf16_kernel<A, C, float>), dim3(blocks), dim3(256), 0, st, (const float *)raw,    \
                         ldraw, stat_sum, stat_sq, gamma, beta, eps, (const bf16 *)skip, ldskip, (const bf16 *)w,      \
                         taps, HW, chunks, ppc);                                                                       \
  } else
  TAPGEMM_BN(256, 256) TAPGEMM_BN(128, 128) TAPGEMM_BN(64, 64) {
    CFD_FAIL(CFD_EUNSUPPORTED, "convT_out_bn: channels %d + %d not built (256+256, 128+128, 64+64)", C1, C2);
  }
#undef TAPGEMM_BN
  CFD_LAUNCH_CHECK();
  const int64_t total = (int64_t)B * 4 * Hi * Wi;
  const int gb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(convT_out_gather_kernel, dim3(gb), dim3(256), 0, st, (const float *)taps, bias, y, logits, B, Hi,
                     Wi, apply_sigmoid);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}
