// Backward / optimiser kernels of the training step (all bandwidth-bound, fused).
//
// What Keras executes inside train_on_batch (U/train.py:121-125, P/train.py, DeltaP/train.py:131):
//   BatchNorm(batch stats) backward fused with the activation / dropout / skip-concat backward,
//   plain activation backward, MaxPooling2D backward, losses + their gradients
//   (binary_crossentropy on sigmoid outputs, MAE = MC_mae U/models/loss.py:47-88, MSE = MC_mse :4-45,
//   Huber DeltaP/models/generator.py:53), Adam (Keras form), BN moving-average update,
//   plus weight transposition for the data-gradient GEMMs and bias-gradient column sums.
#include "common.h"

namespace cfd {

__device__ __forceinline__ uint64_t dropout_hash2(uint64_t seed, uint64_t group) {  // same as elementwise.hip
  uint64_t z = seed + (group + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

template <typename T> struct IO8 {
  static __device__ __forceinline__ void load(const T *p, float (&f)[8]);
  static __device__ __forceinline__ void store(T *p, const float (&f)[8]);
};
template <> struct IO8<bf16> {
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
template <> struct IO8<float> {
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

// derivative by the sign of v: 1 above zero, else slope / 0 / 1 for LEAKY / RELU / none -- the scalar choice is
// loop-invariant, so a run-time `act` costs one compare + select per element instead of a switch
__device__ __forceinline__ float act_grad(float v, int act, float slope) {
  const float ns = act == CFD_ACT_LEAKY ? slope : (act == CFD_ACT_RELU ? 0.f : 1.f);
  return v > 0.f ? 1.f : ns;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm(batch stats, per sample) backward fused with the backward of what followed it.
// Forward was: bn = (x-mean)*rstd*gamma+beta; y1 = act1(keep*scale*bn); y2 = act2(bn) (optional skip copy).
// u = dL/dbn = g1*act1'*keep*scale + g2*act2';  dbeta += sum u; dgamma += sum u*xhat;
// dx = gamma*rstd*(u - mean(u) - xhat*mean(u*xhat)).
// Block = (sample, slab of 64 channels); 8 threads x 8 channels cover a pixel's slab.
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TG>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(
    const TI *__restrict__ x, int ldx, const float *__restrict__ mean, const float *__restrict__ var,
    const float *__restrict__ gamma, const float *__restrict__ beta, const TG *__restrict__ g1, int ldg1,
    const TG *__restrict__ g2, int ldg2, const uint8_t *__restrict__ mask, float drop_scale, unsigned drop_thr,
    uint64_t drop_seed, int64_t sample0, TG *__restrict__ dx, int lddx, float *__restrict__ dgamma,
    float *__restrict__ dbeta, int HW, int C, float eps, int act1, int act2, float slope) {
  constexpr int VN = 8, TPP = 8, CS = 64, PH = 32;
  __shared__ float red[2][PH][CS];
  __shared__ float s_mu[CS], s_rstd[CS], s_g[CS], s_b[CS], s_m1[CS], s_m2[CS];
  const int slabs = (C + CS - 1) / CS;
  const int b = blockIdx.x / slabs;
  const int c0 = (blockIdx.x % slabs) * CS;
  const int tid = threadIdx.x;
  const int cv = (tid % TPP) * VN;
  const int ph = tid / TPP;
  const bool cvalid = c0 + cv < C;
  if (tid < CS && c0 + tid < C) {
    const int c = c0 + tid;
    s_mu[tid] = mean[(size_t)b * C + c];
    s_rstd[tid] = rsqrtf(var[(size_t)b * C + c] + eps);
    s_g[tid] = gamma[c];
    s_b[tid] = beta[c];
  }
  __syncthreads();
  float mu[VN], rs[VN], ga[VN], be[VN];
#pragma unroll
  for (int i = 0; i < VN; ++i) { mu[i] = s_mu[cv + i]; rs[i] = s_rstd[cv + i]; ga[i] = s_g[cv + i]; be[i] = s_b[cv + i]; }
  const TI *xb = x + (size_t)b * HW * ldx + c0 + cv;
  const TG *g1b = g1 + (size_t)b * HW * ldg1 + c0 + cv;
  const TG *g2b = g2 ? g2 + (size_t)b * HW * ldg2 + c0 + cv : nullptr;
  const uint8_t *mb = mask ? mask + (size_t)b * HW * C + c0 + cv : nullptr;

  auto compute_u = [&](int p, float (&xh)[VN], float (&u)[VN]) {
    float xv[VN], gv[VN];
    IO8<TI>::load(xb + (size_t)p * ldx, xv);
    IO8<TG>::load(g1b + (size_t)p * ldg1, gv);
    float keep[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) keep[i] = 1.f;
    if (mb) {
      const uint8_t *mp = mb + (size_t)p * C;
#pragma unroll
      for (int i = 0; i < VN; ++i) keep[i] = mp[i] ? drop_scale : 0.f;
    } else if (drop_thr) {
      const uint64_t e0 = ((uint64_t)(sample0 + b) * HW + p) * C + c0 + cv;
      const uint64_t h = dropout_hash2(drop_seed, e0 >> 3);
#pragma unroll
      for (int i = 0; i < VN; ++i) keep[i] = (((h >> (8 * i)) & 0xff) >= drop_thr) ? drop_scale : 0.f;
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      xh[i] = (xv[i] - mu[i]) * rs[i];
      const float bn = fmaf(xh[i], ga[i], be[i]);
      u[i] = gv[i] * act_grad(bn, act1, slope) * keep[i];   // keep >= 0 does not change the sign fed to act1
    }
    if (g2b) {
      float g2v[VN];
      IO8<TG>::load(g2b + (size_t)p * ldg2, g2v);
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        const float bn = fmaf(xh[i], ga[i], be[i]);
        u[i] = fmaf(g2v[i], act_grad(bn, act2, slope), u[i]);
      }
    }
  };

  float s1[VN], s2[VN];
#pragma unroll
  for (int i = 0; i < VN; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (cvalid)
    for (int p = ph; p < HW; p += PH) {
      float xh[VN], u[VN];
      compute_u(p, xh, u);
#pragma unroll
      for (int i = 0; i < VN; ++i) { s1[i] += u[i]; s2[i] = fmaf(u[i], xh[i], s2[i]); }
    }
#pragma unroll
  for (int i = 0; i < VN; ++i) { red[0][ph][cv + i] = s1[i]; red[1][ph][cv + i] = s2[i]; }
  __syncthreads();
  if (tid < CS) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < PH; ++k) { a += red[0][k][tid]; q += red[1][k][tid]; }
    s_m1[tid] = a / (float)HW;
    s_m2[tid] = q / (float)HW;
    if (c0 + tid < C) {
      if (dbeta) atomicAdd(dbeta + c0 + tid, a);
      if (dgamma) atomicAdd(dgamma + c0 + tid, q);
    }
  }
  __syncthreads();
  if (!cvalid || !dx) return;
  float m1[VN], m2[VN];
#pragma unroll
  for (int i = 0; i < VN; ++i) { m1[i] = s_m1[cv + i]; m2[i] = s_m2[cv + i]; }
  TG *dxb = dx + (size_t)b * HW * lddx + c0 + cv;
  for (int p = ph; p < HW; p += PH) {
    float xh[VN], u[VN], o[VN];
    compute_u(p, xh, u);
#pragma unroll
    for (int i = 0; i < VN; ++i) o[i] = ga[i] * rs[i] * (u[i] - m1[i] - xh[i] * m2[i]);
    IO8<TG>::store(dxb + (size_t)p * lddx, o);
  }
}

// ---------------------------------------------------------------------------------------------
// The same backward for LARGE tensors, split over pixels so that it fills the chip and reads whole pixel rows
// (block = (sample, chunk of pixels), thread = fixed 8-channel vector, as bn_apply_kernel):
//   pass 1 (PASS == 1): per-block partial sums of u and u*xhat -> LDS reduce -> contiguous atomics into
//                       sums[0][b][c], sums[1][b][c] (zeroed by the caller)
//   pass 2 (PASS == 2): dx from the finished sums; the chunk-0 block of every sample adds them to dbeta / dgamma
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TG, int PASS>
__global__ __launch_bounds__(256) void bn_act_bwd_split_kernel(
    const TI *__restrict__ x, int ldx, const float *__restrict__ mean, const float *__restrict__ var,
    const float *__restrict__ gamma, const float *__restrict__ beta, const TG *__restrict__ g1, int ldg1,
    const TG *__restrict__ g2, int ldg2, const uint8_t *__restrict__ mask, float drop_scale, unsigned drop_thr,
    uint64_t drop_seed, int64_t sample0, TG *__restrict__ dx, int lddx, float *__restrict__ dgamma,
    float *__restrict__ dbeta, float *__restrict__ sums, int B, int HW, int C, float eps, int act1, int act2,
    float slope, int chunks, int pix_per_chunk) {
  constexpr int VN = 8;
  __shared__ float red[2][256 * 8];
  const int CV = C >> 3, rpar = 256 / CV;
  const int cv = (threadIdx.x % CV) * 8, rl = threadIdx.x / CV;
  const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
  float mu[VN], rs[VN], ga[VN], be[VN], vv[VN];
  IO8<float>::load(mean + (size_t)b * C + cv, mu);
  IO8<float>::load(var + (size_t)b * C + cv, vv);
  IO8<float>::load(gamma + cv, ga);
  IO8<float>::load(beta + cv, be);
#pragma unroll
  for (int i = 0; i < VN; ++i) rs[i] = rsqrtf(vv[i] + eps);
  const TI *xb = x + (size_t)b * HW * ldx + cv;
  const TG *g1b = g1 + (size_t)b * HW * ldg1 + cv;
  const TG *g2b = g2 ? g2 + (size_t)b * HW * ldg2 + cv : nullptr;
  const uint8_t *mb = mask ? mask + (size_t)b * HW * C + cv : nullptr;

  auto compute_u = [&](int p, float (&xh)[VN], float (&u)[VN]) {
    float xv[VN], gv[VN];
    IO8<TI>::load(xb + (size_t)p * ldx, xv);
    IO8<TG>::load(g1b + (size_t)p * ldg1, gv);
    float keep[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) keep[i] = 1.f;
    if (mb) {
      const uint8_t *mp = mb + (size_t)p * C;
#pragma unroll
      for (int i = 0; i < VN; ++i) keep[i] = mp[i] ? drop_scale : 0.f;
    } else if (drop_thr) {
      const uint64_t e0 = ((uint64_t)(sample0 + b) * HW + p) * C + cv;
      const uint64_t h = dropout_hash2(drop_seed, e0 >> 3);
#pragma unroll
      for (int i = 0; i < VN; ++i) keep[i] = (((h >> (8 * i)) & 0xff) >= drop_thr) ? drop_scale : 0.f;
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
      xh[i] = (xv[i] - mu[i]) * rs[i];
      const float bn = fmaf(xh[i], ga[i], be[i]);
      u[i] = gv[i] * act_grad(bn, act1, slope) * keep[i];
    }
    if (g2b) {
      float g2v[VN];
      IO8<TG>::load(g2b + (size_t)p * ldg2, g2v);
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        const float bn = fmaf(xh[i], ga[i], be[i]);
        u[i] = fmaf(g2v[i], act_grad(bn, act2, slope), u[i]);
      }
    }
  };

  const int p0 = chunk * pix_per_chunk;
  const int p1 = p0 + pix_per_chunk < HW ? p0 + pix_per_chunk : HW;
  float *s1g = sums + (size_t)b * C, *s2g = sums + (size_t)B * C + (size_t)b * C;
  if constexpr (PASS == 1) {
    float s1[VN], s2[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    for (int p = p0 + rl; p < p1; p += rpar) {
      float xh[VN], u[VN];
      compute_u(p, xh, u);
#pragma unroll
      for (int i = 0; i < VN; ++i) { s1[i] += u[i]; s2[i] = fmaf(u[i], xh[i], s2[i]); }
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) { red[0][rl * C + cv + i] = s1[i]; red[1][rl * C + cv + i] = s2[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float a = 0.f, q = 0.f;
      for (int k = 0; k < rpar; ++k) { a += red[0][k * C + c]; q += red[1][k * C + c]; }
      atomicAdd(s1g + c, a);
      atomicAdd(s2g + c, q);
    }
  } else {
    float t1[VN], t2[VN], m1[VN], m2[VN];
    IO8<float>::load(s1g + cv, t1);
    IO8<float>::load(s2g + cv, t2);
    const float inv_n = 1.f / (float)HW;
#pragma unroll
    for (int i = 0; i < VN; ++i) { m1[i] = t1[i] * inv_n; m2[i] = t2[i] * inv_n; }
    if (chunk == 0 && rl == 0) {
#pragma unroll
      for (int i = 0; i < VN; ++i) {
        if (dbeta) atomicAdd(dbeta + cv + i, t1[i]);
        if (dgamma) atomicAdd(dgamma + cv + i, t2[i]);
      }
    }
    if (!dx) return;
    TG *dxb = dx + (size_t)b * HW * lddx + cv;
    for (int p = p0 + rl; p < p1; p += rpar) {
      float xh[VN], u[VN], o[VN];
      compute_u(p, xh, u);
#pragma unroll
      for (int i = 0; i < VN; ++i) o[i] = ga[i] * rs[i] * (u[i] - m1[i] - xh[i] * m2[i]);
      IO8<TG>::store(dxb + (size_t)p * lddx, o);
    }
  }
}

// Column sums of a pixel-parallel backward kernel, for the bias gradient of the layer whose pre-activation gradient it
// writes (a separate colsum pass re-read the whole tensor: 6.9 % of the DeltaP training step at 1.9 TB/s).  Needs
// blockDim % CV == 0 and a grid stride that is a multiple of CV, so that a thread keeps ONE channel group (cv = tid % CV)
// over its whole grid-stride loop: it accumulates 8 channel sums in registers, the block folds its 256 / CV row slots
// through LDS and issues one atomic per channel.  The sums are taken over the values AS STORED (rounded to T), like colsum.
template <typename T>
__device__ __forceinline__ void block_colsum8(const float (&acc)[8], int CV, float *__restrict__ dbias) {
  __shared__ float red[256][9];
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) red[tid][i] = acc[i];
  __syncthreads();
  for (int e = tid; e < CV * 8; e += 256) {          // (C up to 2048: more channels than threads)
    const int cv = e >> 3, i = e & 7;
    float s = 0.f;
    for (int r = cv; r < 256; r += CV) s += red[r][i];
    atomicAdd(dbias + cv * 8 + i, s);
  }
}

// dpre = g1*act1'(y) + g2*act2'(y), the sign taken from the saved activation y (LeakyReLU/ReLU keep it)
template <typename T, bool DBIAS>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T *__restrict__ y, int ldy, const T *__restrict__ g1, int ldg1,
                               const T *__restrict__ g2, int ldg2, T *__restrict__ dx, int lddx, int64_t pixels,
                               int C, int act1, int act2, float slope, float *__restrict__ dbias) {
  const int CV = C / 8;
  const int64_t total = pixels * CV;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / CV;
    const int c = (int)(idx - p * CV) * 8;
    float yv[8], gv[8], o[8];
    IO8<T>::load(y + p * ldy + c, yv);
    IO8<T>::load(g1 + p * ldg1 + c, gv);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = gv[i] * act_grad(yv[i], act1, slope);
    if (g2) {
      IO8<T>::load(g2 + p * ldg2 + c, gv);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = fmaf(gv[i], act_grad(yv[i], act2, slope), o[i]);
    }
    IO8<T>::store(dx + p * lddx + c, o);
    if constexpr (DBIAS) {
#pragma unroll
      for (int i = 0; i < 8; ++i) bsum[i] += to_f32(from_f32<T>(o[i]));
    }
  }
  if constexpr (DBIAS) block_colsum8<T>(bsum, CV, dbias);
}

// MaxPooling2D(2,2) backward fused with the producing layer's activation backward:
// dpre[b,2oy+i,2ox+j,c] = (first max of the window ? dy[b,oy,ox,c] : 0) * act'(x)
template <typename T, bool DBIAS>
__global__ __launch_bounds__(256) void maxpool2_act_bwd_kernel(const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ dx, int B,
                                        int Hi, int Wi, int C, int act, float slope, float *__restrict__ dbias) {
  const int Ho = Hi / 2, Wo = Wi / 2, CV = C / 8;
  const int64_t total = (int64_t)B * Ho * Wo * CV;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % CV);
    const int64_t op = idx / CV;
    const int ox = (int)(op % Wo);
    const int oy = (int)((op / Wo) % Ho);
    const int b = (int)(op / ((int64_t)Wo * Ho));
    const size_t base = (((size_t)b * Hi + 2 * oy) * Wi + 2 * ox) * C + cv * 8;
    const size_t offs[4] = {0, (size_t)C, (size_t)Wi * C, (size_t)Wi * C + C};
    float v[4][8], g[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) IO8<T>::load(x + base + offs[k], v[k]);
    IO8<T>::load(dy + (size_t)op * C + cv * 8, g);
    float o[4][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int best = 0;
      float m = v[0][i];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][i] > m) { m = v[k][i]; best = k; }
      const float gg = g[i] * act_grad(m, act, slope);       // (m = v[best][i])
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][i] = (k == best) ? gg : 0.f;
      if constexpr (DBIAS) bsum[i] += to_f32(from_f32<T>(gg));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) IO8<T>::store(dx + base + offs[k], o[k]);
  }
  if constexpr (DBIAS) block_colsum8<T>(bsum, CV, dbias);
}

// dst[ci][t'][co] = src[co][t][ci], t' = flip ? T-1-t : t   (64x64 tiles through LDS, per tap)
template <typename T>
__global__ __launch_bounds__(256) void transpose_weight_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                              int Cout, int taps, int Cin, int flip) {
  __shared__ T tile[64][65];
  const int t = blockIdx.z;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? src[((size_t)co * taps + t) * Cin + ci] : from_f32<T>(0.f);
  }
  __syncthreads();
  const int td = flip ? taps - 1 - t : t;
  for (int r = ty; r < 64; r += 4) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) dst[((size_t)ci * taps + td) * Cout + co] = tile[tx][r];
  }
}

// bf16, Cin % 8 == 0 && Cout % 8 == 0: 128(co) x 128(ci) tile, 16-B accesses on both sides (256-B runs per row)
__global__ __launch_bounds__(256) void transpose_weight_bf16v_kernel(const bf16 *__restrict__ src, bf16 *__restrict__ dst,
                                                                    int Cout, int taps, int Cin, int flip) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[128][136];
  const int t = blockIdx.z;
  const int co0 = blockIdx.y * 128, ci0 = blockIdx.x * 128;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 256 + tid, r = idx >> 4, c8 = (idx & 15) * 8;
    const int co = co0 + r, ci = ci0 + c8;
    uint4 v = {0u, 0u, 0u, 0u};
    if (co < Cout && ci < Cin) v = *(const uint4 *)(src + ((size_t)co * taps + t) * Cin + ci);
    *(uint4 *)&tile[r][c8] = v;
  }
  __syncthreads();
  const int td = flip ? taps - 1 - t : t;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = i * 256 + tid, r = idx & 127, c8 = (idx >> 7) * 8;   // consecutive lanes: consecutive ci rows
    const int ci = ci0 + r, co = co0 + c8;
    if (ci < Cin && co < Cout) {
      unsigned short e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = tile[c8 + k][r];
      uint4 v = {(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                 (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
      *(uint4 *)(dst + ((size_t)ci * taps + td) * Cout + co) = v;
    }
  }
}

// out[c] += sum_m x[m][c]   (bias gradients)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T *__restrict__ x, int ldx, float *__restrict__ out,
                                                    int64_t M, int C, int rows_per_block) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ry = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
  float s = 0.f;
  if (c < C)
    for (int64_t m = m0 + ry; m < m1; m += 4) s += to_f32(x[m * ldx + c]);
  red[ry][threadIdx.x & 63] = s;
  __syncthreads();
  if (ry == 0 && c < C) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- losses ------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v) {
  __shared__ float sm[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm[0] + sm[1] + sm[2] + sm[3];
}

// binary_crossentropy on logits z with target t: mean(max(z,0) - z t + log1p(exp(-|z|))); dz = gscale*(sigmoid(z)-t)/n
__global__ __launch_bounds__(256) void bce_logits_kernel(const float *__restrict__ z, const float *__restrict__ tgt,
                                                        float tconst, int64_t n, float *__restrict__ loss,
                                                        float *__restrict__ dz, float gscale) {
  float acc = 0.f;
  const float inv_n = 1.f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float zi = z[i], t = tgt ? tgt[i] : tconst;
    acc += fmaxf(zi, 0.f) - zi * t + log1pf(expf(-fabsf(zi)));
    if (dz) dz[i] = gscale * (1.f / (1.f + expf(-zi)) - t) * inv_n;
  }
  const float s = block_sum(acc);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, s * inv_n);
}

// kind 0 MAE, 1 MSE, 2 Huber(delta).  loss += mean(l(y-t)); dL/dy = weight*l'(y-t)/n + dy_extra;
// dout = dL/dy * (sigmoid_grad ? y(1-y) : 1)
__global__ __launch_bounds__(256) void field_loss_kernel(const float *__restrict__ y, const float *__restrict__ t,
                                                        int64_t n, int kind, float delta, float weight,
                                                        float *__restrict__ loss, const float *__restrict__ dy_extra,
                                                        float *__restrict__ dout, int sigmoid_grad) {
  float acc = 0.f;
  const float inv_n = 1.f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float yi = y[i], e = yi - t[i], a = fabsf(e);
    float l, d;
    if (kind == 0) { l = a; d = (e > 0.f) ? 1.f : (e < 0.f ? -1.f : 0.f); }
    else if (kind == 1) { l = e * e; d = 2.f * e; }
    else { l = (a <= delta) ? 0.5f * e * e : delta * (a - 0.5f * delta); d = (a <= delta) ? e : (e > 0.f ? delta : -delta); }
    acc += l;
    if (dout) {
      float g = weight * d * inv_n + (dy_extra ? dy_extra[i] : 0.f);
      if (sigmoid_grad) g *= yi * (1.f - yi);
      dout[i] = g;
    }
  }
  const float s = block_sum(acc);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, s * inv_n);
}

// ---- Adam (Keras form): m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= alpha*m/(sqrt(v)+eps) ---------
template <typename TC>
__global__ void adam_kernel(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, TC *__restrict__ wc, int64_t n, float alpha, float b1, float b2,
                            float eps, float gscale) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 wv = ((f32x4 *)w)[i], gv = ((const f32x4 *)g)[i], mv = ((f32x4 *)m)[i], vv = ((f32x4 *)v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gv[k] * gscale;
      mv[k] += (gk - mv[k]) * (1.f - b1);
      vv[k] += (gk * gk - vv[k]) * (1.f - b2);
      wv[k] -= (mv[k] * alpha) / (sqrtf(vv[k]) + eps);
    }
    ((f32x4 *)w)[i] = wv; ((f32x4 *)m)[i] = mv; ((f32x4 *)v)[i] = vv;
    if (wc) {
      if constexpr (sizeof(TC) == 2) {
        bf16x4 o = {(bf16)wv[0], (bf16)wv[1], (bf16)wv[2], (bf16)wv[3]};
        ((bf16x4 *)wc)[i] = o;
      }
    }
  }
}

// moving = moving*mom + stat_b*(1-mom), applied once per sample in order (B small)
__global__ void bn_update_moving_kernel(float *__restrict__ mm, float *__restrict__ mv, const float *__restrict__ mean,
                                        const float *__restrict__ var, int B, int C, float mom) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = mm[c], q = mv[c];
  for (int b = 0; b < B; ++b) {
    a = a * mom + mean[(size_t)b * C + c] * (1.f - mom);
    q = q * mom + var[(size_t)b * C + c] * (1.f - mom);
  }
  mm[c] = a; mv[c] = q;
}

// the same for up to 16 BatchNorm layers in one launch (a discriminator update touches four: 12 launches of ~8 us per
// training step became 3); blockIdx.y = layer
constexpr int BN_MULTI_MAX = 16;
struct BnMultiArgs {
  float *mm[BN_MULTI_MAX], *mv[BN_MULTI_MAX];
  const float *mean[BN_MULTI_MAX], *var[BN_MULTI_MAX];
  int C[BN_MULTI_MAX];
};
__global__ void bn_update_moving_multi_kernel(BnMultiArgs a, int B, float mom) {
  const int l = blockIdx.y, C = a.C[l];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float m = a.mm[l][c], q = a.mv[l][c];
  const float *mean = a.mean[l], *var = a.var[l];
  for (int b = 0; b < B; ++b) {
    m = m * mom + mean[(size_t)b * C + c] * (1.f - mom);
    q = q * mom + var[(size_t)b * C + c] * (1.f - mom);
  }
  a.mm[l][c] = m; a.mv[l][c] = q;
}

// base[off .. off + cnt) = 0 for every (off, cnt) pair of a device-resident table: the gradient tensors that kernels
// ACCUMULATE into (bias / BatchNorm / first- and last-layer gradients), in one launch instead of a memset of the whole
// flat gradient buffer (512 MB for the U generator) -- the GEMM-shaped weight gradients are plain stores
__global__ __launch_bounds__(256) void zero_ranges_kernel(float *__restrict__ base, const int64_t *__restrict__ ranges) {
  const int64_t off = ranges[2 * blockIdx.y], cnt = ranges[2 * blockIdx.y + 1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) base[off + i] = 0.f;
}

// y = x * keep * scale over a dense [rows][C] tensor (Dropout(0.5) after fc1, DeltaP/models/generator.py:42; same
// kernel applies the mask to the gradient in the backward pass)
template <typename T>
__global__ void dropout_apply_kernel(const T *__restrict__ x, T *__restrict__ y, const uint8_t *__restrict__ mask,
                                     int64_t total, float scale, unsigned thr, uint64_t seed, int64_t e_begin) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    bool keep;
    if (mask) keep = mask[i] != 0;
    else {
      const uint64_t e = (uint64_t)(e_begin + i);
      keep = ((dropout_hash2(seed, e >> 3) >> ((e & 7) * 8)) & 0xff) >= thr;
    }
    y[i] = from_f32<T>(keep ? to_f32(x[i]) * scale : 0.f);
  }
}

// ---- tap family (layers with ONE output channel): backward -----------------------------------------
// dtaps[pix][t] = dz[output pixel fed by (pix, tap t)] or 0.  geom 0: Conv2DTranspose k4 s2 (generator out layer),
// out = (2iy+kh-1, 2ix+kw-1) on (2Hi,2Wi).  geom 1: Conv2D k4 s1 'same' pad(1,2) (discriminator c6),
// out = (iy-kh+1, ix-kw+1) on (Hi,Wi).
__global__ void tap_scatter_grad_kernel(const float *__restrict__ dz, float *__restrict__ dtaps, int B, int Hi, int Wi,
                                        int geom) {
  const int64_t total = (int64_t)B * Hi * Wi * 16;
  const int Ho = geom == 0 ? 2 * Hi : Hi, Wo = geom == 0 ? 2 * Wi : Wi;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(idx & 15);
    const int64_t pix = idx >> 4;
    const int ix = (int)(pix % Wi);
    const int iy = (int)((pix / Wi) % Hi);
    const int b = (int)(pix / ((int64_t)Wi * Hi));
    const int kh = t >> 2, kw = t & 3;
    const int oy = geom == 0 ? 2 * iy + kh - 1 : iy - kh + 1;
    const int ox = geom == 0 ? 2 * ix + kw - 1 : ix - kw + 1;
    float v = 0.f;
    if ((unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo) v = dz[((int64_t)b * Ho + oy) * Wo + ox];
    dtaps[idx] = v;
  }
}

// dx[row][c] = sum_{t<T} d[row][t] * w[t][c]     (T = 16 taps, or 1 for a Dense with one output)
template <typename T>
__global__ void tap_dx_kernel(const float *__restrict__ d, const T *__restrict__ w, T *__restrict__ dx, int lddx,
                              int64_t rows, int C, int ntaps) {
  const int CV = C / 8;
  const int64_t total = rows * CV;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = idx / CV;
    const int c = (int)(idx - r * CV) * 8;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    for (int t = 0; t < ntaps; ++t) {
      const float dv = d[r * ntaps + t];
      float wv[8];
      IO8<T>::load(w + (size_t)t * C + c, wv);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = fmaf(dv, wv[i], o[i]);
    }
    IO8<T>::store(dx + r * lddx + c, o);
  }
}

// dw[t][c] += sum_rows d[row][t] * x[row][c];  block = chunk of rows, thread = 8 channels, 16 taps in registers.
// The 256/CV row lanes of a block are summed through LDS, then every thread adds CONTIGUOUS channels with
// atomics (one 256-B run per wave instruction: the shape global float atomics run at full rate with); with
// <= 256 blocks the atomic traffic is a few MB instead of the ~130 MB a per-thread scatter costs.
template <typename T>
__global__ __launch_bounds__(256) void tap_dw_kernel(const float *__restrict__ d, const T *__restrict__ x, int ldx,
                                                    float *__restrict__ dw, int64_t rows, int C, int ntaps,
                                                    int rows_per_block) {
  __shared__ float red[256 * 8];
  const int CV = C / 8;                // threads per row
  const int rpar = 256 / CV;           // rows in flight (host guarantees CV <= 256 and 256 % CV == 0)
  const int cv = threadIdx.x % CV, rl = threadIdx.x / CV;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float acc[16][8];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
  if (ntaps == 16) {
    for (int64_t r = r0 + rl; r < r1; r += rpar) {
      float xv[8];
      IO8<T>::load(x + r * ldx + cv * 8, xv);
      const f32x4 *dp = (const f32x4 *)(d + r * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 dv = dp[q];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[q * 4 + e][i] = fmaf(dv[e], xv[i], acc[q * 4 + e][i]);
      }
    }
  } else {
    for (int64_t r = r0 + rl; r < r1; r += rpar) {
      float xv[8];
      IO8<T>::load(x + r * ldx + cv * 8, xv);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (t < ntaps) {
          const float dv = d[r * ntaps + t];
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[t][i] = fmaf(dv, xv[i], acc[t][i]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    if (t < ntaps) {   // uniform
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) red[rl * C + cv * 8 + i] = acc[t][i];
      __syncthreads();
      for (int c = threadIdx.x; c < C; c += 256) {
        float sum = 0.f;
        for (int k = 0; k < rpar; ++k) sum += red[k * C + c];
        atomicAdd(dw + (size_t)t * C + c, sum);
      }
    }
  }
}

// direct first-layer weight gradient: dw[co][tap][ci] += sum_pix dpre[pix][co] * patch(pix)[tap][ci], db[co] += sum dpre
// (e1, c1, g1: Cin <= 2).  thread = output channel co, block = chunk of output pixels; the block's partial
// sums are transposed through LDS so that the atomics of a wave hit contiguous addresses.
template <typename T, int KS, int S, int CIN, int CGRP>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ x2,
                                                              int cin1, const T *__restrict__ dpre, int ldd,
                                                              float *__restrict__ dw, float *__restrict__ db, int B,
                                                              int Hi, int Wi, int Ho, int Wo, int Cout, int pix_per_block) {
  // thread = (channel co, pixel group grp): CGRP = 64 / 128 / 256 threads cover the channels, the 256 / CGRP groups split
  // the pixels of a step (with one thread per channel and nothing else, a 64-channel layer -- DeltaP g1 -- kept one wave of
  // four busy).  Pixel indices are 32-bit (host: B * Ho * Wo < 2^31): the staging loop's three 64-bit divisions per element
  // cost more than the step's FMAs.
  constexpr int NW = KS * KS * CIN;
  constexpr int G = 256 / CGRP, PPT = 32;   // pixel groups; pixels per thread and step = dpre loads in flight per thread
  constexpr int PIXB = PPT * G;             // pixels per staged step
  __shared__ float patch[PIXB][NW];
  __shared__ float outb[256 * (NW + 1)];   // [thread][NW] padded by one against bank conflicts
  __shared__ float outbias[256];
  const unsigned npix = (unsigned)B * Ho * Wo;
  const unsigned p0 = blockIdx.x * (unsigned)pix_per_block;
  const unsigned p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
  const int co = threadIdx.x % CGRP, grp = threadIdx.x / CGRP;
  float acc[NW];
  float accb = 0.f;
#pragma unroll
  for (int k = 0; k < NW; ++k) acc[k] = 0.f;
  const int cc = co < Cout ? co : Cout - 1;          // (threads beyond Cout load a valid address and contribute nothing)
  for (unsigned pb = p0; pb < p1; pb += PIXB) {
    // the step's gradient rows first: their latency runs under the patch staging (both were serial: 2 round trips per step)
    float gv[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const unsigned pix = pb + grp + G * i;
      const float g = to_f32(dpre[(size_t)(pix < p1 ? pix : p1 - 1) * ldd + cc]);
      gv[i] = pix < p1 ? g : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PIXB * NW; e += 256) {
      const int pi = e / NW, k = e % NW;
      const unsigned pix = pb + pi;
      float v = 0.f;
      if (pix < p1) {
        const int ci = k % CIN, kw = (k / CIN) % KS, kh = k / (CIN * KS);
        const unsigned row = pix / (unsigned)Wo;
        const int ox = (int)(pix - row * Wo), b = (int)(row / (unsigned)Ho), oy = (int)(row - (unsigned)b * Ho);
        const int iy = oy * S - 1 + kh, ix = ox * S - 1 + kw;
        if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
          const size_t ip = ((size_t)b * Hi + iy) * Wi + ix;
          v = (ci < cin1) ? x[ip * cin1 + ci] : x2[ip * (CIN - cin1) + (ci - cin1)];
        }
      }
      patch[pi][k] = v;
    }
    __syncthreads();
    if (co < Cout) {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        accb += gv[i];
#pragma unroll
        for (int k = 0; k < NW; ++k) acc[k] = fmaf(gv[i], patch[grp + G * i][k], acc[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NW; ++k) outb[threadIdx.x * (NW + 1) + k] = acc[k];
  outbias[threadIdx.x] = accb;
  __syncthreads();
  for (int e = threadIdx.x; e < Cout * NW; e += 256) {
    const int c = e / NW, k = e - c * NW;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) t += outb[(g * CGRP + c) * (NW + 1) + k];
    atomicAdd(dw + e, t);
  }
  if (threadIdx.x < Cout && db) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) t += outbias[g * CGRP + threadIdx.x];
    atomicAdd(db + threadIdx.x, t);
  }
}

// The same gradient, lane = (pixel lane, group of CPL output channels) (round 3).  The kernel above spends one LDS read per
// FMA (thread = one channel: a patch value serves one FMA) and loads the gradient rows 2 bytes per lane; DeltaP g1 at
// B = 64 took 165 us for 268 MB of gradient rows (1.8 TB/s) with the vector ALU issuing two instructions per FMA.  Here a
// lane keeps CPL channels x NW taps of accumulators (72 for g1's 3 x 3 taps x 8 channels), loads its CPL gradient values of eight
// pixels with 16 / 8 / 4-byte accesses ahead of the patch staging (one pixel per thread: two divisions per staged pixel
// instead of three per staged element) and does CPL FMAs per LDS read.
template <typename T, int KS, int S, int CIN, int CPL>
__global__ __launch_bounds__(256) void conv_first_wgrad2_kernel(const float *__restrict__ x, const float *__restrict__ x2,
                                                               int cin1, const T *__restrict__ dpre, int ldd,
                                                               float *__restrict__ dw, float *__restrict__ db, int B,
                                                               int Hi, int Wi, int Ho, int Wo, int Cout, int lg_ncgp,
                                                               int pix_per_block) {
  constexpr int NW = KS * KS * CIN, NWP = NW | 1;    // odd row pitch: the staging writes and the reads spread over the banks
  constexpr int PPT = 8;                             // pixels per thread and step
  __shared__ float red[256 * CPL];
  __shared__ float sh[256 * (NW + 1)];               // the step's patches [pixel][NWP]; after the loop the block's sums [co][NW]
  const int tid = threadIdx.x;
  const int ncgp = 1 << lg_ncgp, npl = 256 >> lg_ncgp;
  const int cg = tid & (ncgp - 1), pl = tid >> lg_ncgp;
  const int ncg = Cout / CPL;
  const int c0 = (cg < ncg ? cg : 0) * CPL;          // (lanes beyond the channels load a valid address; never summed)
  const int pixb = npl * PPT;                        // pixels per step (<= 256)
  float acc[CPL][NW], accb[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    accb[c] = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) acc[c][k] = 0.f;
  }
  const unsigned npix = (unsigned)B * Ho * Wo;
  const unsigned p0 = blockIdx.x * (unsigned)pix_per_block;
  const unsigned p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
  const int cin2 = CIN - cin1;
  for (unsigned pb = p0; pb < p1; pb += pixb) {
    // the step's gradient rows first: their latency runs under the patch staging
    float gv[PPT][CPL];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const unsigned pix = pb + pl + npl * i;
      const T *src = dpre + (size_t)(pix < p1 ? pix : p1 - 1) * ldd + c0;
      const bool ok = pix < p1;
      if constexpr (sizeof(T) == 2) {
        if constexpr (CPL == 8) { const bf16x8 v = *(const bf16x8 *)src;
#pragma unroll
          for (int c = 0; c < 8; ++c) gv[i][c] = ok ? (float)v[c] : 0.f; }
        else if constexpr (CPL == 4) { const bf16x4 v = *(const bf16x4 *)src;
#pragma unroll
          for (int c = 0; c < 4; ++c) gv[i][c] = ok ? (float)v[c] : 0.f; }
        else { const bf16x2 v = *(const bf16x2 *)src; gv[i][0] = ok ? (float)v[0] : 0.f; gv[i][1] = ok ? (float)v[1] : 0.f; }
      } else {
#pragma unroll
        for (int c = 0; c < CPL; c += 2) {
          const f32x2 v = *(const f32x2 *)(src + c);
          gv[i][c] = ok ? v[0] : 0.f; gv[i][c + 1] = ok ? v[1] : 0.f;
        }
      }
    }
    __syncthreads();
    if (tid < pixb) {                                // one pixel per thread
      const unsigned pix = pb + tid;
      float pv[NW];
#pragma unroll
      for (int k = 0; k < NW; ++k) pv[k] = 0.f;
      if (pix < p1) {
        const unsigned row = pix / (unsigned)Wo;
        const int ox = (int)(pix - row * Wo), b = (int)(row / (unsigned)Ho), oy = (int)(row - (unsigned)b * Ho);
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
          const int iy = oy * S - 1 + kh;
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            const int ix = ox * S - 1 + kw;
            if ((unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi) {
              const size_t ip = ((size_t)b * Hi + iy) * Wi + ix;
#pragma unroll
              for (int ci = 0; ci < CIN; ++ci)
                pv[(kh * KS + kw) * CIN + ci] = ci < cin1 ? x[ip * cin1 + ci] : x2[ip * cin2 + (ci - cin1)];
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < NW; ++k) sh[tid * NWP + k] = pv[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float *pp = sh + (pl + npl * i) * NWP;
#pragma unroll
      for (int c = 0; c < CPL; ++c) accb[c] += gv[i][c];
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        const float pvk = pp[k];
#pragma unroll
        for (int c = 0; c < CPL; ++c) acc[c][k] = fmaf(gv[i][c], pvk, acc[c][k]);
      }
    }
  }
  // ---- block reduction over the pixel lanes, one tap at a time through 8 KiB of LDS; the sums leave as one contiguous
  // burst of atomics (dw is [co][tap][ci] = [co][k])
  float *osum = sh;
#pragma unroll
  for (int k = 0; k <= NW; ++k) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPL; ++c) red[tid * CPL + c] = k < NW ? acc[c][k < NW ? k : 0] : accb[c];
    __syncthreads();
    float t[(256 + 255) / 256];
    for (int co = tid, n = 0; co < Cout; co += 256, ++n) {
      const int g = co / CPL, c = co - g * CPL;
      float tt = 0.f;
      for (int q = 0; q < npl; ++q) tt += red[((q << lg_ncgp) + g) * CPL + c];
      t[n] = tt;
    }
    if (k == 0) __syncthreads();                    // (every thread is done reading the last step's patches before sh is reused)
    for (int co = tid, n = 0; co < Cout; co += 256, ++n) {
      if (k < NW) osum[co * NW + (k < NW ? k : 0)] = t[n];
      else if (db) atomicAdd(db + co, t[n]);
    }
  }
  __syncthreads();
  for (int e = tid; e < Cout * NW; e += 256) atomicAdd(dw + e, osum[e]);
}

// Streaming kernels get ONE vector per thread (no cap): measured on the Adam update over 128 Mi parameters
// (scripts/micro/adam_bw.hip), a grid-stride loop over 8 192 blocks runs at 4.0-4.1 TB/s, the uncapped grid at 5.4 -- with a
// capped grid every block's next iteration lands 32 MB further on in eight streams at once.  Kernels that keep
// per-block state (loss sums -> one atomic per block) pass their own cap.
static inline int grid_for(int64_t total, int cap = 1 << 30) {
  const int64_t b = (total + 255) / 256;
  return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}
static inline unsigned drop_threshold2(float rate) { return (unsigned)(rate * 256.f + 0.5f); }

}  // namespace cfd

using namespace cfd;

static int bn_act_bwd_impl(const void *x, int ldx, int x_dtype, const float *mean, const float *var,
                              const float *gamma, const float *beta, const void *g1, int ldg1, const void *g2,
                              int ldg2, const uint8_t *drop_mask, float drop_rate, uint64_t drop_seed,
                              int64_t sample0, void *dx, int lddx, float *dgamma, float *dbeta, int B, int HW, int C,
                              float eps, int act1, int act2, float slope, int dtype, float *sums, void *stream) {
  CFD_REQUIRE(x && mean && var && gamma && beta && g1, "NULL pointer");
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  CFD_REQUIRE(x_dtype == CFD_F32 || (x_dtype == CFD_BF16 && dtype == CFD_BF16), "bad x_dtype/dtype combination");
  CFD_REQUIRE(C % 8 == 0 && ldx % 8 == 0 && ldg1 % 8 == 0 && (!g2 || ldg2 % 8 == 0) && (!dx || lddx % 8 == 0),
              "C and strides must be multiples of 8");
  CFD_REQUIRE(drop_rate >= 0.f && drop_rate < 1.f, "drop_rate must be in [0,1)");
  CFD_REQUIRE(drop_mask || drop_rate == 0.f || !(drop_seed >> 63),
              "epoch-keyed dropout seeds (bit 63, cfd_dropout_epoch) are forward-only: a backward cannot reproduce them");
  const float drop_scale = 1.f / (1.f - drop_rate);
  const unsigned thr = (drop_mask || drop_rate == 0.f) ? 0u : drop_threshold2(drop_rate);
  hipStream_t st = (hipStream_t)stream;
  // large tensors with a sums workspace: two coalesced, chip-filling passes
  // (also at small batch: the one-pass kernel runs one block per (sample, 64 channels), 8 blocks for a 512-channel
  //  tensor at batch 1)
  const bool big = (int64_t)B * HW * C >= (1 << 22) || ((int64_t)HW * C >= (1 << 18) && B * (C / 64) < 128);
  if (sums && C >= 64 && C / 8 <= 256 && 256 % (C / 8) == 0 && big) {
    const int rpar = 256 / (C / 8);
    int ppc = rpar * 16;
    if (ppc > HW) ppc = HW;
    const int chunks = (HW + ppc - 1) / ppc;
#define BWD2(TI, TG, PASS)                                                                                           \
  hipLaunchKernelGGL((bn_act_bwd_split_kernel<TI, TG, PASS>), dim3(B * chunks), dim3(256), 0, st, (const TI *)x, ldx, \
                     mean, var, gamma, beta, (const TG *)g1, ldg1, (const TG *)g2, ldg2, drop_mask, drop_scale, thr, \
                     drop_seed, sample0, (TG *)dx, lddx, dgamma, dbeta, sums, B, HW, C, eps, act1, act2, slope,      \
                     chunks, ppc)
    if (dtype == CFD_BF16 && x_dtype == CFD_BF16) { BWD2(bf16, bf16, 1); BWD2(bf16, bf16, 2); }
    else if (dtype == CFD_BF16) { BWD2(float, bf16, 1); BWD2(float, bf16, 2); }
    else { BWD2(float, float, 1); BWD2(float, float, 2); }
#undef BWD2
    CFD_LAUNCH_CHECK();
    return CFD_OK;
  }
  const int blocks = B * ((C + 63) / 64);
#define BWD(TI, TG)                                                                                                  \
  hipLaunchKernelGGL((bn_act_bwd_kernel<TI, TG>), dim3(blocks), dim3(256), 0, st, (const TI *)x, ldx, mean, var,     \
                     gamma, beta, (const TG *)g1, ldg1, (const TG *)g2, ldg2, drop_mask, drop_scale, thr, drop_seed, \
                     sample0, (TG *)dx, lddx, dgamma, dbeta, HW, C, eps, act1, act2, slope)
  if (dtype == CFD_BF16 && x_dtype == CFD_BF16) BWD(bf16, bf16);
  else if (dtype == CFD_BF16) BWD(float, bf16);
  else BWD(float, float);
#undef BWD
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_bn_act_bwd(const void *x, int ldx, int x_dtype, const float *mean, const float *var,
                              const float *gamma, const float *beta, const void *g1, int ldg1, const void *g2,
                              int ldg2, const uint8_t *drop_mask, float drop_rate, uint64_t drop_seed,
                              int64_t sample0, void *dx, int lddx, float *dgamma, float *dbeta, int B, int HW, int C,
                              float eps, int act1, int act2, float slope, int dtype, void *stream) {
  return bn_act_bwd_impl(x, ldx, x_dtype, mean, var, gamma, beta, g1, ldg1, g2, ldg2, drop_mask, drop_rate, drop_seed,
                         sample0, dx, lddx, dgamma, dbeta, B, HW, C, eps, act1, act2, slope, dtype, nullptr, stream);
}

extern "C" int cfd_bn_act_bwd_ws(const void *x, int ldx, int x_dtype, const float *mean, const float *var,
                                 const float *gamma, const float *beta, const void *g1, int ldg1, const void *g2,
                                 int ldg2, const uint8_t *drop_mask, float drop_rate, uint64_t drop_seed,
                                 int64_t sample0, void *dx, int lddx, float *dgamma, float *dbeta, int B, int HW, int C,
                                 float eps, int act1, int act2, float slope, int dtype, float *sums, void *stream) {
  return bn_act_bwd_impl(x, ldx, x_dtype, mean, var, gamma, beta, g1, ldg1, g2, ldg2, drop_mask, drop_rate, drop_seed,
                         sample0, dx, lddx, dgamma, dbeta, B, HW, C, eps, act1, act2, slope, dtype, sums, stream);
}

// grid for the kernels that may carry a bias-gradient reduction: whole blocks of 256 threads whose global stride is a
// multiple of C / 8 (C / 8 divides 256), so a thread keeps one channel group
static int act_bwd_impl(const void *y, int ldy, const void *g1, int ldg1, const void *g2, int ldg2, void *dx,
                        int lddx, int64_t pixels, int C, int act1, int act2, float slope, int dtype, float *dbias,
                        hipStream_t st) {
  CFD_REQUIRE(y && g1 && dx, "NULL pointer");
  CFD_REQUIRE(C % 8 == 0 && ldy % 8 == 0 && ldg1 % 8 == 0 && lddx % 8 == 0 && (!g2 || ldg2 % 8 == 0),
              "C and strides must be multiples of 8");
  CFD_REQUIRE(!dbias || (C <= 2048 && 256 % (C / 8) == 0), "bias-gradient reduction: C / 8 must divide 256 (C = %d)", C);
  // (with the bias gradient: a bounded grid, so that the per-block atomics stay few -- 2048 blocks x C)
  const int blocks = dbias ? grid_for(pixels * (C / 8), 2048) : grid_for(pixels * (C / 8));
#define CFD_ACT_BWD(T, DB)                                                                                              \
  hipLaunchKernelGGL((act_bwd_kernel<T, DB>), dim3(blocks), dim3(256), 0, st, (const T *)y, ldy, (const T *)g1, ldg1,    \
                     (const T *)g2, ldg2, (T *)dx, lddx, pixels, C, act1, act2, slope, dbias)
  if (dtype == CFD_BF16) { if (dbias) CFD_ACT_BWD(bf16, true); else CFD_ACT_BWD(bf16, false); }
  else if (dtype == CFD_F32) { if (dbias) CFD_ACT_BWD(float, true); else CFD_ACT_BWD(float, false); }
  else CFD_FAIL(CFD_EINVAL, "bad dtype");
#undef CFD_ACT_BWD
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_act_bwd(const void *y, int ldy, const void *g1, int ldg1, const void *g2, int ldg2, void *dx,
                           int lddx, int64_t pixels, int C, int act1, int act2, float slope, int dtype, void *stream) {
  return act_bwd_impl(y, ldy, g1, ldg1, g2, ldg2, dx, lddx, pixels, C, act1, act2, slope, dtype, nullptr, (hipStream_t)stream);
}

extern "C" int cfd_act_bwd_dbias(const void *y, int ldy, const void *g1, int ldg1, const void *g2, int ldg2, void *dx,
                                 int lddx, int64_t pixels, int C, int act1, int act2, float slope, int dtype, float *dbias,
                                 void *stream) {
  CFD_REQUIRE(dbias, "dbias is NULL");
  return act_bwd_impl(y, ldy, g1, ldg1, g2, ldg2, dx, lddx, pixels, C, act1, act2, slope, dtype, dbias, (hipStream_t)stream);
}

static int maxpool2_act_bwd_impl(const void *x, const void *dy, void *dx, int B, int Hi, int Wi, int C, int act,
                                 float slope, int dtype, float *dbias, hipStream_t st) {
  CFD_REQUIRE(x && dy && dx, "NULL pointer");
  CFD_REQUIRE(Hi % 2 == 0 && Wi % 2 == 0 && C % 8 == 0, "bad shape");
  CFD_REQUIRE(!dbias || (C <= 2048 && 256 % (C / 8) == 0), "bias-gradient reduction: C / 8 must divide 256 (C = %d)", C);
  const int64_t work = (int64_t)B * (Hi / 2) * (Wi / 2) * (C / 8);
  const int blocks = dbias ? grid_for(work, 2048) : grid_for(work);
#define CFD_POOL_BWD(T, DB)                                                                                             \
  hipLaunchKernelGGL((maxpool2_act_bwd_kernel<T, DB>), dim3(blocks), dim3(256), 0, st, (const T *)x, (const T *)dy,      \
                     (T *)dx, B, Hi, Wi, C, act, slope, dbias)
  if (dtype == CFD_BF16) { if (dbias) CFD_POOL_BWD(bf16, true); else CFD_POOL_BWD(bf16, false); }
  else if (dtype == CFD_F32) { if (dbias) CFD_POOL_BWD(float, true); else CFD_POOL_BWD(float, false); }
  else CFD_FAIL(CFD_EINVAL, "bad dtype");
#undef CFD_POOL_BWD
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_maxpool2_act_bwd(const void *x, const void *dy, void *dx, int B, int Hi, int Wi, int C, int act,
                                    float slope, int dtype, void *stream) {
  return maxpool2_act_bwd_impl(x, dy, dx, B, Hi, Wi, C, act, slope, dtype, nullptr, (hipStream_t)stream);
}

extern "C" int cfd_maxpool2_act_bwd_dbias(const void *x, const void *dy, void *dx, int B, int Hi, int Wi, int C, int act,
                                          float slope, int dtype, float *dbias, void *stream) {
  CFD_REQUIRE(dbias, "dbias is NULL");
  return maxpool2_act_bwd_impl(x, dy, dx, B, Hi, Wi, C, act, slope, dtype, dbias, (hipStream_t)stream);
}

extern "C" int cfd_transpose_weight(const void *src, void *dst, int Cout, int taps, int Cin, int flip, int dtype,
                                    void *stream) {
  CFD_REQUIRE(src && dst && Cout > 0 && taps > 0 && Cin > 0, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((Cin + 63) / 64, (Cout + 63) / 64, taps);
  if (dtype == CFD_BF16 && Cin % 8 == 0 && Cout % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0)
    hipLaunchKernelGGL(transpose_weight_bf16v_kernel, dim3((Cin + 127) / 128, (Cout + 127) / 128, taps), dim3(256), 0, st,
                       (const bf16 *)src, (bf16 *)dst, Cout, taps, Cin, flip);
  else if (dtype == CFD_BF16)
    hipLaunchKernelGGL(transpose_weight_kernel<bf16>, grid, dim3(256), 0, st, (const bf16 *)src, (bf16 *)dst, Cout, taps, Cin, flip);
  else if (dtype == CFD_F32)
    hipLaunchKernelGGL(transpose_weight_kernel<float>, grid, dim3(256), 0, st, (const float *)src, (float *)dst, Cout, taps, Cin, flip);
  else CFD_FAIL(CFD_EINVAL, "bad dtype");
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_colsum(const void *x, int ldx, float *out, int64_t M, int C, int dtype, void *stream) {
  CFD_REQUIRE(x && out && M > 0 && C > 0, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int64_t rpb = (M + 511) / 512;
  if (rpb < 64) rpb = 64;
  dim3 grid((C + 63) / 64, (unsigned)((M + rpb - 1) / rpb));
  if (dtype == CFD_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16>, grid, dim3(256), 0, st, (const bf16 *)x, ldx, out, M, C, (int)rpb);
  else if (dtype == CFD_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, st, (const float *)x, ldx, out, M, C, (int)rpb);
  else CFD_FAIL(CFD_EINVAL, "bad dtype");
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_bce_logits(const float *z, const float *target, float target_const, int64_t n, float *loss,
                              float *dz, float grad_scale, void *stream) {
  CFD_REQUIRE(z && n > 0, "bad arguments");
  hipLaunchKernelGGL(bce_logits_kernel, dim3(grid_for(n, 512)), dim3(256), 0, (hipStream_t)stream, z, target,
                     target_const, n, loss, dz, grad_scale);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_field_loss(const float *y, const float *target, int64_t n, int kind, float delta, float weight,
                              float *loss, const float *dy_extra, float *dout, int sigmoid_grad, void *stream) {
  CFD_REQUIRE(y && target && n > 0 && kind >= 0 && kind <= 2, "bad arguments");
  hipLaunchKernelGGL(field_loss_kernel, dim3(grid_for(n, 512)), dim3(256), 0, (hipStream_t)stream, y, target, n, kind,
                     delta, weight, loss, dy_extra, dout, sigmoid_grad);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_adam_step(float *w, const float *g, float *m, float *v, void *w_compute, int compute_dtype,
                             int64_t n, float alpha, float beta1, float beta2, float eps, float grad_scale,
                             void *stream) {
  CFD_REQUIRE(w && g && m && v && n > 0 && n % 4 == 0, "bad arguments (n must be a multiple of 4)");
  hipStream_t st = (hipStream_t)stream;
  // one 16-B vector per stream and thread, as many blocks as that takes: measured against a bounded grid with two vectors in
  // flight per thread (scripts/micro/adam_bw.py, 128 M parameters): 737 us = 5.46 TB/s against 804-946 us = 4.3-5.0 TB/s
  const int blocks = grid_for(n / 4);
  if (w_compute && compute_dtype == CFD_BF16)
    hipLaunchKernelGGL(adam_kernel<bf16>, dim3(blocks), dim3(256), 0, st, w, g, m, v, (bf16 *)w_compute, n, alpha,
                       beta1, beta2, eps, grad_scale);
  else
    hipLaunchKernelGGL(adam_kernel<float>, dim3(blocks), dim3(256), 0, st, w, g, m, v, (float *)nullptr, n, alpha,
                       beta1, beta2, eps, grad_scale);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_bn_update_moving(float *moving_mean, float *moving_var, const float *mean, const float *var, int B,
                                    int C, float momentum, void *stream) {
  CFD_REQUIRE(moving_mean && moving_var && mean && var && B > 0 && C > 0, "bad arguments");
  hipLaunchKernelGGL(bn_update_moving_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, moving_mean,
                     moving_var, mean, var, B, C, momentum);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_bn_update_moving_multi(int nlayers, float *const *moving_mean, float *const *moving_var,
                                          const float *const *mean, const float *const *var, const int *C, int B,
                                          float momentum, void *stream) {
  CFD_REQUIRE(nlayers > 0 && nlayers <= BN_MULTI_MAX && moving_mean && moving_var && mean && var && C && B > 0,
              "bad arguments (at most %d layers per call)", BN_MULTI_MAX);
  BnMultiArgs a;
  int cmax = 0;
  for (int l = 0; l < nlayers; ++l) {
    CFD_REQUIRE(moving_mean[l] && moving_var[l] && mean[l] && var[l] && C[l] > 0, "layer %d: NULL pointer or C <= 0", l);
    a.mm[l] = moving_mean[l]; a.mv[l] = moving_var[l]; a.mean[l] = mean[l]; a.var[l] = var[l]; a.C[l] = C[l];
    cmax = C[l] > cmax ? C[l] : cmax;
  }
  hipLaunchKernelGGL(bn_update_moving_multi_kernel, dim3((cmax + 255) / 256, nlayers), dim3(256), 0, (hipStream_t)stream, a,
                     B, momentum);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_zero_ranges(float *base, const int64_t *ranges, int n, int64_t max_count, void *stream) {
  CFD_REQUIRE(base && ranges && n > 0 && n <= 65535 && max_count > 0, "bad arguments");
  int bx = (int)((max_count + 2047) / 2048);
  bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, base, ranges);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_dropout_apply(const void *x, void *y, const uint8_t *mask, int64_t total, float rate,
                                 uint64_t seed, int64_t elem0, int dtype, void *stream) {
  CFD_REQUIRE(x && y && total > 0 && rate > 0.f && rate < 1.f, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const float scale = 1.f / (1.f - rate);
  const unsigned thr = drop_threshold2(rate);
  if (dtype == CFD_BF16)
    hipLaunchKernelGGL(dropout_apply_kernel<bf16>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16 *)x, (bf16 *)y,
                       mask, total, scale, thr, seed, elem0);
  else if (dtype == CFD_F32)
    hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float *)x,
                       (float *)y, mask, total, scale, thr, seed, elem0);
  else CFD_FAIL(CFD_EINVAL, "bad dtype");
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_tap_scatter_grad(const float *dz, float *dtaps, int B, int Hi, int Wi, int geom, void *stream) {
  CFD_REQUIRE(dz && dtaps && B > 0 && Hi > 0 && Wi > 0 && (geom == 0 || geom == 1), "bad arguments");
  hipLaunchKernelGGL(tap_scatter_grad_kernel, dim3(grid_for((int64_t)B * Hi * Wi * 16)), dim3(256), 0,
                     (hipStream_t)stream, dz, dtaps, B, Hi, Wi, geom);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_tap_dx(const float *d, const void *w, void *dx, int lddx, int64_t rows, int C, int ntaps, int dtype,
                          void *stream) {
  CFD_REQUIRE(d && w && dx && rows > 0 && C % 8 == 0 && lddx % 8 == 0 && ntaps >= 1 && ntaps <= 16, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = grid_for(rows * (C / 8));
  if (dtype == CFD_BF16)
    hipLaunchKernelGGL(tap_dx_kernel<bf16>, dim3(blocks), dim3(256), 0, st, d, (const bf16 *)w, (bf16 *)dx, lddx, rows, C, ntaps);
  else if (dtype == CFD_F32)
    hipLaunchKernelGGL(tap_dx_kernel<float>, dim3(blocks), dim3(256), 0, st, d, (const float *)w, (float *)dx, lddx, rows, C, ntaps);
  else CFD_FAIL(CFD_EINVAL, "bad dtype");
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_tap_dw(const float *d, const void *x, int ldx, float *dw, int64_t rows, int C, int ntaps, int dtype,
                          void *stream) {
  CFD_REQUIRE(d && x && dw && rows > 0 && C % 8 == 0 && ldx % 8 == 0 && ntaps >= 1 && ntaps <= 16, "bad arguments");
  const int cv = C / 8;
  CFD_REQUIRE(cv <= 256 && 256 % cv == 0, "tap_dw: C=%d unsupported", C);
  hipStream_t st = (hipStream_t)stream;
  int64_t rpb = (rows + 255) / 256;
  const int rpar = 256 / cv;
  if (rpb < 8 * rpar) rpb = 8 * rpar;
  const int blocks = (int)((rows + rpb - 1) / rpb);
  if (dtype == CFD_BF16)
    hipLaunchKernelGGL(tap_dw_kernel<bf16>, dim3(blocks), dim3(256), 0, st, d, (const bf16 *)x, ldx, dw, rows, C, ntaps, (int)rpb);
  else if (dtype == CFD_F32)
    hipLaunchKernelGGL(tap_dw_kernel<float>, dim3(blocks), dim3(256), 0, st, d, (const float *)x, ldx, dw, rows, C, ntaps, (int)rpb);
  else CFD_FAIL(CFD_EINVAL, "bad dtype");
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_conv_first_wgrad(const float *x, const float *x2, int cin1, int cin2, const void *dpre, int ldd,
                                    float *dw, float *db, int B, int Hi, int Wi, int Cout, int ksize, int stride,
                                    int dtype, void *stream) {
  CFD_REQUIRE(x && dpre && dw, "NULL pointer");
  CFD_REQUIRE(Cout <= 256, "conv_first_wgrad: Cout=%d > 256", Cout);
  CFD_REQUIRE(dtype == CFD_BF16 || dtype == CFD_F32, "bad dtype");
  hipStream_t st = (hipStream_t)stream;
  const int Ho = Hi / stride, Wo = Wi / stride;
  const int64_t npix = (int64_t)B * Ho * Wo;
  CFD_REQUIRE(npix > 0 && npix < (1LL << 31) - 4096, "conv_first_wgrad: too many pixels");
  // ~1024 blocks of 32-pixel steps (32 dpre loads in flight per thread and barrier pair); each block ends with one
  // contiguous 4-32 KB atomic burst
  int64_t ppb = (npix + 1023) / 1024;
  if (ppb < 64) ppb = 64;
  ppb = (ppb + 127) / 128 * 128;     // a multiple of every variant's step (32 pixels x 1, 2 or 4 groups)
  const int blocks = (int)((npix + ppb - 1) / ppb);
  const int cin = cin1 + cin2;
  // lane = (pixel lane, group of CPL channels) when the channels divide into at most 64 groups (every first layer of the
  // reference's graphs: g1 64, e1 256 / 128, c1 64 channels); else the thread-per-channel kernel below
  {
    const int nw = ksize * ksize * cin;
    const int cpl = nw <= 9 ? 8 : (nw <= 16 ? 4 : 2);
    int lg = 3;
    while ((1 << lg) * cpl < Cout) ++lg;
    const size_t es = dtype == CFD_BF16 ? 2 : 4;
    const bool aligned = ((uintptr_t)dpre % (cpl * es)) == 0 && ((size_t)ldd * es) % (cpl * es) == 0;
    const bool x2ok = (cin2 > 0) == (x2 != nullptr);
    if (Cout % cpl == 0 && lg <= 6 && aligned && x2ok) {
      const int step = (256 >> lg) * 8;            // pixels per staged step
      int64_t ppb2 = (npix + 1023) / 1024;
      ppb2 = (ppb2 + step - 1) / step * step;
      const dim3 grid((unsigned)((npix + ppb2 - 1) / ppb2));
#define CFW2(T, KS, S, CIN, CPL)                                                                                       \
  hipLaunchKernelGGL((conv_first_wgrad2_kernel<T, KS, S, CIN, CPL>), grid, dim3(256), 0, st, x, x2, cin1, (const T *)dpre, \
                     ldd, dw, db, B, Hi, Wi, Ho, Wo, Cout, lg, (int)ppb2)
      // measured (interleaved A/B, one box): DeltaP's step 3.336 -> 3.269 ms with g1 here (3 x 3, 8 channels per lane); the
      // 4 x 4 first layers (e1: 4 channels per lane, c1: 2) have 32 / 64-pixel steps here and ran the U step 0.7 % SLOWER
      // than the thread-per-channel kernel, so they stay there
      bool done = true;
      if (ksize == 3 && stride == 1 && cin == 1) { if (dtype == CFD_BF16) CFW2(bf16, 3, 1, 1, 8); else CFW2(float, 3, 1, 1, 8); }
      else done = false;
#undef CFW2
      if (done) {
        CFD_LAUNCH_CHECK();
        return CFD_OK;
      }
    }
  }
#define CFW_(T, KS, S, CIN, CG)                                                                                   \
  hipLaunchKernelGGL((conv_first_wgrad_kernel<T, KS, S, CIN, CG>), dim3(blocks), dim3(256), 0, st, x, x2, cin1,    \
                     (const T *)dpre, ldd, dw, db, B, Hi, Wi, Ho, Wo, Cout, (int)ppb)
#define CFW(T, KS, S, CIN)                                                                                         \
  do {                                                                                                             \
    if (Cout <= 64) CFW_(T, KS, S, CIN, 64); else if (Cout <= 128) CFW_(T, KS, S, CIN, 128);                      \
    else CFW_(T, KS, S, CIN, 256);                                                                                 \
  } while (0)
  if (ksize == 4 && stride == 2 && cin == 1) { if (dtype == CFD_BF16) CFW(bf16, 4, 2, 1); else CFW(float, 4, 2, 1); }
  else if (ksize == 4 && stride == 2 && cin == 2) { if (dtype == CFD_BF16) CFW(bf16, 4, 2, 2); else CFW(float, 4, 2, 2); }
  else if (ksize == 3 && stride == 1 && cin == 1) { if (dtype == CFD_BF16) CFW(bf16, 3, 1, 1); else CFW(float, 3, 1, 1); }
  else CFD_FAIL(CFD_EUNSUPPORTED, "conv_first_wgrad: k=%d s=%d cin=%d not built", ksize, stride, cin);
#undef CFW_
#undef CFW
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}
