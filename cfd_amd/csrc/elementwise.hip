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

// Device-resident dropout epoch: added (times an odd constant) to RNG-mode dropout seeds of the FORWARD kernels whose
// bit 63 is set (CFD_DROP_SEED_EPOCH_BIT).  A captured HIP graph bakes its kernel arguments, so a graph-replayed
// predict passes such a seed and advances this word with a one-thread kernel at the head of the graph to get a fresh
// draw per replay (generator.py:36 keeps Dropout active at inference).  Seeds below 2^63 -- everything a training
// step uses, whose backward kernels must reproduce the forward mask -- are never touched by the epoch.
__device__ unsigned long long g_drop_epoch = 0ULL;

__global__ void drop_epoch_kernel(unsigned long long delta, int set) {
  if (threadIdx.x == 0 && blockIdx.x == 0) g_drop_epoch = set ? delta : g_drop_epoch + delta;
}

// splitmix64 finaliser over (seed, group index): 8 random bytes, one per element of an aligned group of 8
__device__ __forceinline__ uint64_t dropout_hash(uint64_t seed, uint64_t group) {
  if (seed >> 63) seed += g_drop_epoch * 0xD6E8FEB86659FD93ULL;
  uint64_t z = seed + (group + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// the mask the RNG mode of bn_act applies, written out as 0/1 bytes (parity tests feed it to the oracle)
__global__ void dropout_mask_kernel(uint8_t *__restrict__ out, int64_t total, int64_t e_begin, uint64_t seed,
                                    unsigned thr) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t e = (uint64_t)(e_begin + idx);
    const uint64_t h = dropout_hash(seed, e >> 3);
    out[idx] = (((h >> ((e & 7) * 8)) & 0xff) >= thr) ? 1 : 0;
  }
}

template <typename T> struct Vec;
template <> struct Vec<bf16> { typedef bf16x8 type; static constexpr int N = 8; };
template <> struct Vec<float> { typedef f32x4 type; static constexpr int N = 4; };

// vector load/store of VN consecutive elements (16-B pieces)
template <typename T, int VN> struct VecIO;
template <> struct VecIO<bf16, 8> {
  static __device__ __forceinline__ void load(const bf16 *p, float (&f)[8]) {
    const bf16x8 v = *(const bf16x8 *)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  static __device__ __forceinline__ void store(bf16 *p, const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
    *(bf16x8 *)p = v;
  }
};
template <> struct VecIO<float, 8> {
  static __device__ __forceinline__ void load(const float *p, float (&f)[8]) {
    const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  }
  static __device__ __forceinline__ void store(float *p, const float (&f)[8]) {
    *(f32x4 *)p = (f32x4){f[0], f[1], f[2], f[3]};
    *(f32x4 *)(p + 4) = (f32x4){f[4], f[5], f[6], f[7]};
  }
};
template <> struct VecIO<float, 4> {
  static __device__ __forceinline__ void load(const float *p, float (&f)[4]) {
    const f32x4 a = *(const f32x4 *)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = a[i];
  }
  static __device__ __forceinline__ void store(float *p, const float (&f)[4]) {
    *(f32x4 *)p = (f32x4){f[0], f[1], f[2], f[3]};
  }
};

// Block = (sample b, slab of CS = 8*VN channels).  TPP = 8 threads cover one pixel's slab; 256/TPP pixel
// phases stride over the HW pixels.  Pass 1: shifted sums -> mean, biased var.  Pass 2: normalise,
// dropout, activation(s), store.  TI = dtype of the conv output (fp32 or bf16), TO = activation dtype.
template <typename TI, typename TO, int VN>
__global__ __launch_bounds__(256) void bn_act_kernel(const TI *__restrict__ x, int ldx, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, float *__restrict__ mean_io,
                                                    float *__restrict__ var_io, int use_moving,
                                                    const uint8_t *__restrict__ mask, float drop_scale,
                                                    unsigned drop_thr, uint64_t drop_seed, int64_t sample0,
                                                    TO *__restrict__ y, int ldy, TO *__restrict__ y2, int ldy2, int HW,
                                                    int C, float eps, int act, int act2, float slope) {
  constexpr int TPP = 8;
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
  const TI *xb = x + (size_t)b * HW * ldx + c0 + cv;

  if (!use_moving) {
    float sh[VN], s1[VN], s2[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) { sh[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
    if (cvalid) {
      VecIO<TI, VN>::load(xb, sh);  // pixel 0 as the shift: removes the mean from the sums of squares
      for (int p = ph; p < HW; p += PH) {
        float v[VN];
        VecIO<TI, VN>::load(xb + (size_t)p * ldx, v);
#pragma unroll
        for (int i = 0; i < VN; ++i) {
          const float d = v[i] - sh[i];
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
  TO *yb = y + (size_t)b * HW * ldy + c0 + cv;
  TO *y2b = y2 ? y2 + (size_t)b * HW * ldy2 + c0 + cv : nullptr;
  const uint8_t *mb = mask ? mask + (size_t)b * HW * C + c0 + cv : nullptr;
  for (int p = ph; p < HW; p += PH) {
    float f[VN];
    VecIO<TI, VN>::load(xb + (size_t)p * ldx, f);
#pragma unroll
    for (int i = 0; i < VN; ++i) f[i] = fmaf(f[i], sc[i], sf[i]);
    if (mb) {
      const uint8_t *mp = mb + (size_t)p * C;
#pragma unroll
      for (int i = 0; i < VN; ++i) f[i] = mp[i] ? f[i] * drop_scale : 0.f;
    } else if (drop_thr) {
      // counter-based RNG: element e = ((sample0+b)*HW + p)*C + c draws byte (e & 7) of hash(seed, e >> 3)
      const uint64_t e0 = ((uint64_t)(sample0 + b) * HW + p) * C + c0 + cv;
      const uint64_t h = dropout_hash(drop_seed, e0 >> 3);
      const int sh0 = (int)(e0 & 7) * 8;
#pragma unroll
      for (int i = 0; i < VN; ++i) f[i] = (((h >> (sh0 + 8 * i)) & 0xff) >= drop_thr) ? f[i] * drop_scale : 0.f;
    }
    float o[VN];
    apply_act_vec(o, f, act_neg_scale(act, slope), act == CFD_ACT_SIGMOID);
    VecIO<TO, VN>::store(yb + (size_t)p * ldy, o);
    if (y2b) {
      apply_act_vec(o, f, act_neg_scale(act2, slope), act2 == CFD_ACT_SIGMOID);
      VecIO<TO, VN>::store(y2b + (size_t)p * ldy2, o);
    }
  }
}


// BatchNorm apply for conv outputs whose per-sample statistics were accumulated by the conv epilogue
// (cfd_conv_forward_stats): ONE read of x, fully coalesced (a pixel row of C channels is covered by C/8
// consecutive threads, 16 B each), normalise + dropout + activation(s), write y (and y2).
// Block = (sample b, chunk of pixels); thread = fixed 8-channel vector, so scale/shift live in registers.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TI *__restrict__ x, int ldx, const float *__restrict__ ssum,
                                                      const float *__restrict__ ssq, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float *__restrict__ mean_out,
                                                      float *__restrict__ var_out, const uint8_t *__restrict__ mask,
                                                      float drop_scale, unsigned drop_thr, uint64_t drop_seed,
                                                      int64_t sample0, TO *__restrict__ y, int ldy, TO *__restrict__ y2,
                                                      int ldy2, int HW, int C, float eps, int act, int act2, float slope,
                                                      int chunks, int pix_per_chunk) {
  const int CV = C >> 3;          // threads per pixel row (host: CV <= 256 and 256 % CV == 0)
  const int rpar = 256 / CV;
  const int cv = (threadIdx.x % CV) * 8, rl = threadIdx.x / CV;
  const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
  float sc[8], sf[8];
  {
    const float inv_n = 1.f / (float)HW;
    float s1[8], s2[8], g[8], be[8];
    VecIO<float, 8>::load(ssum + (size_t)b * C + cv, s1);
    VecIO<float, 8>::load(ssq + (size_t)b * C + cv, s2);
    VecIO<float, 8>::load(gamma + cv, g);
    VecIO<float, 8>::load(beta + cv, be);
    float mean[8], var[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      mean[i] = s1[i] * inv_n;
      const float v = fmaf(-mean[i], mean[i], s2[i] * inv_n);
      var[i] = v > 0.f ? v : 0.f;
      sc[i] = g[i] * rsqrtf(var[i] + eps);
      sf[i] = be[i] - mean[i] * sc[i];
    }
    if (chunk == 0 && rl == 0) {
      if (mean_out) VecIO<float, 8>::store(mean_out + (size_t)b * C + cv, mean);
      if (var_out) VecIO<float, 8>::store(var_out + (size_t)b * C + cv, var);
    }
  }
  const int p0 = chunk * pix_per_chunk;
  const int p1 = p0 + pix_per_chunk < HW ? p0 + pix_per_chunk : HW;
  const TI *xb = x + (size_t)b * HW * ldx + cv;
  TO *yb = y + (size_t)b * HW * ldy + cv;
  TO *y2b = y2 ? y2 + (size_t)b * HW * ldy2 + cv : nullptr;
  const uint8_t *mb = mask ? mask + (size_t)b * HW * C + cv : nullptr;
  for (int p = p0 + rl; p < p1; p += rpar) {
    float f[8];
    VecIO<TI, 8>::load(xb + (size_t)p * ldx, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = fmaf(f[i], sc[i], sf[i]);
    if (mb) {
      const uint8_t *mp = mb + (size_t)p * C;
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = mp[i] ? f[i] * drop_scale : 0.f;
    } else if (drop_thr) {
      const uint64_t e0 = ((uint64_t)(sample0 + b) * HW + p) * C + cv;   // multiple of 8
      const uint64_t h = dropout_hash(drop_seed, e0 >> 3);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = (((h >> (8 * i)) & 0xff) >= drop_thr) ? f[i] * drop_scale : 0.f;
    }
    float o[8];
    apply_act_vec(o, f, act_neg_scale(act, slope), act == CFD_ACT_SIGMOID);
    VecIO<TO, 8>::store(yb + (size_t)p * ldy, o);
    if (y2b) {
      apply_act_vec(o, f, act_neg_scale(act2, slope), act2 == CFD_ACT_SIGMOID);
      VecIO<TO, 8>::store(y2b + (size_t)p * ldy2, o);
    }
  }
}

// Per-sample BatchNorm statistics of a tensor whose producer could not fuse them (split-K convs, i.e. the small-batch
// case): sum and sum of squares per (sample, channel), pixel-parallel like bn_apply -- block = (sample, chunk of pixels),
// thread = fixed 8-channel vector; rows of a block are reduced through LDS, then one atomic per channel per block.
// bn_act_kernel's own statistics pass uses one block per (sample, 64 channels): 8 blocks for a 512-channel tensor at
// batch 1 (97 us for 2 MB); this + bn_apply fills the chip.
template <typename TI>
__global__ __launch_bounds__(256) void bn_stats_kernel(const TI *__restrict__ x, int ldx, float *__restrict__ ssum,
                                                      float *__restrict__ ssq, int HW, int C, int chunks,
                                                      int pix_per_chunk) {
  __shared__ float red[2][256][9];
  const int CV = C >> 3, rpar = 256 / CV;
  const int cvi = threadIdx.x % CV, cv = cvi * 8, rl = threadIdx.x / CV;
  const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
  const int p0 = chunk * pix_per_chunk;
  const int p1 = p0 + pix_per_chunk < HW ? p0 + pix_per_chunk : HW;
  const TI *xb = x + (size_t)b * HW * ldx + cv;
  float s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.f;
  for (int p = p0 + rl; p < p1; p += rpar) {
    float f[8];
    VecIO<TI, 8>::load(xb + (size_t)p * ldx, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] += f[i]; s2[i] = fmaf(f[i], f[i], s2[i]); }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = s1[i]; red[1][threadIdx.x][i] = s2[i]; }
  __syncthreads();
  // thread t < 2 * C: statistic (t / C), channel (t % C): sum over the rpar row groups
  for (int t = threadIdx.x; t < 2 * C; t += 256) {
    const int which = t / C, c = t - which * C;
    float tot = 0.f;
    for (int r = 0; r < rpar; ++r) tot += red[which][r * CV + (c >> 3)][c & 7];
    atomicAdd((which ? ssq : ssum) + (size_t)b * C + c, tot);
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

static inline unsigned drop_threshold(float rate) { return (unsigned)(rate * 256.f + 0.5f); }

extern "C" int cfd_dropout_epoch(uint64_t value, int set, void *stream) {
  hipLaunchKernelGGL(drop_epoch_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)value, set);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_dropout_mask(uint8_t *mask, int64_t sample0, int B, int HW, int C, float rate, uint64_t seed,
                                void *stream) {
  CFD_REQUIRE(mask && B > 0 && HW > 0 && C > 0 && C % 8 == 0, "bad arguments");
  CFD_REQUIRE(rate > 0.f && rate < 1.f, "rate must be in (0,1)");
  const int64_t total = (int64_t)B * HW * C;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, total,
                     sample0 * HW * C, seed, drop_threshold(rate));
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_bn_act(const void *x, int ldx, int x_dtype, const float *gamma, const float *beta, float *mean,
                          float *var, int use_moving, const uint8_t *drop_mask, float drop_rate, uint64_t drop_seed,
                          int64_t sample0, void *y, int ldy, void *y2, int ldy2, int B, int HW, int C, float eps,
                          int act, int act2, float slope, int dtype, void *stream) {
  CFD_REQUIRE(drop_rate >= 0.f && drop_rate < 1.f, "drop_rate must be in [0,1)");
  const float drop_scale = 1.f / (1.f - drop_rate);
  const unsigned drop_thr = (drop_mask || drop_rate == 0.f) ? 0u : drop_threshold(drop_rate);
  CFD_REQUIRE(!drop_thr || C % 8 == 0, "RNG dropout needs C %% 8 == 0");
  CFD_REQUIRE(x && gamma && beta && y, "NULL pointer");
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  CFD_REQUIRE(x_dtype == CFD_F32 || (x_dtype == CFD_BF16 && dtype == CFD_BF16), "bad x_dtype/dtype combination");
  CFD_REQUIRE(!use_moving || (mean && var), "use_moving needs mean and var");
  const int vn = dtype == CFD_BF16 ? 8 : 4;
  CFD_REQUIRE(C % vn == 0 && ldx % vn == 0 && ldy % vn == 0 && (!y2 || ldy2 % vn == 0),
              "C/ldx/ldy must be multiples of %d", vn);
  CFD_REQUIRE(B > 0 && HW > 0, "bad shape");
  const int cs = 8 * vn;
  const int blocks = B * ((C + cs - 1) / cs);
  hipStream_t st = (hipStream_t)stream;
#define BN_LAUNCH(TI, TO, VN)                                                                                        \
  hipLaunchKernelGGL((bn_act_kernel<TI, TO, VN>), dim3(blocks), dim3(256), 0, st, (const TI *)x, ldx, gamma, beta,   \
                     mean, var, use_moving, drop_mask, drop_scale, drop_thr, drop_seed, sample0, (TO *)y, ldy,       \
                     (TO *)y2, ldy2, HW, C, eps, act, act2, slope)
  if (dtype == CFD_BF16 && x_dtype == CFD_BF16) BN_LAUNCH(bf16, bf16, 8);
  else if (dtype == CFD_BF16) BN_LAUNCH(float, bf16, 8);
  else BN_LAUNCH(float, float, 4);
#undef BN_LAUNCH
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_bn_apply(const void *x, int ldx, int x_dtype, const float *stat_sum, const float *stat_sq,
                            const float *gamma, const float *beta, float *mean, float *var, const uint8_t *drop_mask,
                            float drop_rate, uint64_t drop_seed, int64_t sample0, void *y, int ldy, void *y2, int ldy2,
                            int B, int HW, int C, float eps, int act, int act2, float slope, int dtype, void *stream) {
  CFD_REQUIRE(drop_rate >= 0.f && drop_rate < 1.f, "drop_rate must be in [0,1)");
  const float drop_scale = 1.f / (1.f - drop_rate);
  const unsigned drop_thr = (drop_mask || drop_rate == 0.f) ? 0u : drop_threshold(drop_rate);
  CFD_REQUIRE(x && stat_sum && stat_sq && gamma && beta && y, "NULL pointer");
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  CFD_REQUIRE(x_dtype == CFD_F32 || (x_dtype == CFD_BF16 && dtype == CFD_BF16), "bad x_dtype/dtype combination");
  CFD_REQUIRE(B > 0 && HW > 0 && C >= 64 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0,
              "bn_apply: C=%d must be 64..2048 with C/8 a divisor of 256", C);
  CFD_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && (!y2 || ldy2 % 8 == 0), "ldx/ldy must be multiples of 8");
  // ~16 pixel rows per thread-row keeps >= 2048 blocks in flight on the big layers without shrinking a block's run
  const int rpar = 256 / (C / 8);
  int ppc = rpar * 16;
  if (ppc > HW) ppc = HW;
  const int chunks = (HW + ppc - 1) / ppc;
  hipStream_t st = (hipStream_t)stream;
#define BNA_LAUNCH(TI, TO)                                                                                            \
  hipLaunchKernelGGL((bn_apply_kernel<TI, TO>), dim3(B * chunks), dim3(256), 0, st, (const TI *)x, ldx, stat_sum,     \
                     stat_sq, gamma, beta, mean, var, drop_mask, drop_scale, drop_thr, drop_seed, sample0, (TO *)y,   \
                     ldy, (TO *)y2, ldy2, HW, C, eps, act, act2, slope, chunks, ppc)
  if (dtype == CFD_BF16 && x_dtype == CFD_BF16) BNA_LAUNCH(bf16, bf16);
  else if (dtype == CFD_BF16) BNA_LAUNCH(float, bf16);
  else BNA_LAUNCH(float, float);
#undef BNA_LAUNCH
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_bn_stats(const void *x, int ldx, int x_dtype, float *stat_sum, float *stat_sq, int B, int HW, int C,
                            void *stream) {
  CFD_REQUIRE(x && stat_sum && stat_sq, "NULL pointer");
  CFD_REQUIRE(x_dtype == CFD_BF16 || x_dtype == CFD_F32, "bad dtype");
  CFD_REQUIRE(B > 0 && HW > 0 && C >= 64 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0,
              "bn_stats: C=%d must be 64..2048 with C/8 a divisor of 256", C);
  CFD_REQUIRE(ldx % 8 == 0, "ldx must be a multiple of 8");
  const int rpar = 256 / (C / 8);
  int ppc = rpar * 32;
  if (ppc > HW) ppc = HW;
  const int chunks = (HW + ppc - 1) / ppc;
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == CFD_BF16)
    hipLaunchKernelGGL(bn_stats_kernel<bf16>, dim3(B * chunks), dim3(256), 0, st, (const bf16 *)x, ldx, stat_sum, stat_sq,
                       HW, C, chunks, ppc);
  else
    hipLaunchKernelGGL(bn_stats_kernel<float>, dim3(B * chunks), dim3(256), 0, st, (const float *)x, ldx, stat_sum,
                       stat_sq, HW, C, chunks, ppc);
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
