// Weight gradients of the conv / transposed-conv / dense layers on MFMA.
//
// Keras computes these inside `train_on_batch` (U/train.py:121-125, DeltaP/train.py:131) as the
// backward of Conv2D / Conv2DTranspose / Dense.  For one kernel tap t:
//     dW[co][t][ci] = sum_m  P_t[m][co] * Q_t[m][ci]
//   conv kinds : m over the OUTPUT grid;  P = dY[m] (plain rows),  Q = X[pixel(m) * S - pad + tap] (gathered)
//   transposed : m over the INPUT grid;   Q = X[m]  (plain rows),  P = dY[2*pixel(m) + tap - 1]    (gathered)
//   dense      : m = batch row, one tap, both plain
// i.e. a GEMM whose REDUCTION index (m) is the slow memory axis of both operands.  Tiles of 64 rows
// (m) x 256 B (128 bf16 / 64 f32 channels) are staged with global_load_lds in their natural row-major
// form; bf16 fragments are then read with the gfx950 transposing LDS read (ds_read_b64_tr_b16), which
// hands each lane 4 consecutive m of one channel -- the k-contiguous layout MFMA wants -- so neither
// operand is ever transposed in memory.  f32 fragments are plain 4-byte LDS reads.
// 16-B chunks are XOR-swizzled with (row & 15) (source side + read side) against bank conflicts.
// Grid = co tiles x ci tiles x taps x split_m.  split_m > 1 accumulates with fp32 atomics into a
// zeroed buffer (summation order then varies run to run); split_m == 1 stores.
#include "common.h"
#include "wgrad256.h"

namespace cfd {

struct WgradParams {
  const char *x;   // input activations  [B][Hx][Wx] stride ldx
  const char *dy;  // output gradients   [B][Hy][Wy] stride ldy
  float *dw;       // [Cout][taps][Cin] fp32
  int B, Hx, Wx, Hy, Wy, Cin, Cout, ldx, ldy;
  int Hm, Wm, M;   // reduction grid
  int kind, ntaps, flip;
  int pair, tap_groups;   // pair: bf16 conv kinds with Cin == 64 -- a 128-channel tile row holds taps t and t + 1 (64 channels each)
  int cotiles, citiles, split_m;
  unsigned long long magicW, magicHW;  // exact division of m by Wm and by Hm*Wm: q = (m * magic) >> (32 + shift)
  int shiftW, shiftHW;
  int nblocks;
  float *slab;          // split-M partial results [split_m][Cout][taps][Cin], summed by wgrad_slab_reduce in a fixed order; NULL = atomics
  int64_t dw_elems;
};

__device__ __attribute__((aligned(16))) const unsigned int g_zero_page_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};

constexpr int W_TILE_BYTES = 64 * 256;       // 64 rows (m) x 256 B
constexpr int W_STAGE_BYTES = 2 * W_TILE_BYTES;

__device__ __forceinline__ void glds16w(const char *gsrc, char *lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ int fastdiv(int m, unsigned long long magic, int shift) {
  return (int)(((unsigned long long)(unsigned)m * magic) >> (32 + shift));
}

template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  constexpr int ES = sizeof(T);
  constexpr int CT = 256 / ES;   // channels per tile row: 128 bf16 / 64 f32
  constexpr int EPC = 16 / ES;
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bid = blockIdx.x;
  const int cit = bid % p.citiles; bid /= p.citiles;
  const int cot = bid % p.cotiles; bid /= p.cotiles;
  const int tap = (bid % p.tap_groups) * (p.pair ? 2 : 1);
  const int sm = bid / p.tap_groups;
  const int co0 = cot * CT, ci0 = cit * CT;

  // tap geometry: offsets applied to the gathered operand
  int kh, kw;
  if (p.kind == CFD_CONV_K3S1) { kh = tap / 3; kw = tap - kh * 3; }
  else if (p.kind == CFD_DENSE) { kh = 0; kw = 0; }
  else { kh = tap >> 2; kw = tap & 3; }
  const bool transposed = p.kind == CFD_CONVT_K4S2;
  const int S = (p.kind == CFD_CONV_K4S2 || transposed) ? 2 : 1;
  const int pad = (p.kind == CFD_DENSE) ? 0 : 1;
  const int oy = kh - pad, ox = kw - pad;  // gathered pixel = grid pixel * S + (oy, ox)
  // pair mode: the second tap of the tile (columns 64..127); an odd tap count leaves the last pair without one
  const bool has2 = p.pair && tap + 1 < p.ntaps;
  int kh2, kw2;
  if (p.kind == CFD_CONV_K3S1) { kh2 = (tap + 1) / 3; kw2 = (tap + 1) - kh2 * 3; }
  else { kh2 = (tap + 1) >> 2; kw2 = (tap + 1) & 3; }
  const int oy2 = kh2 - pad, ox2 = kw2 - pad;

  // staging: one wave-instruction = 1 KiB = 4 rows x 256 B; lane -> row lane/16, chunk slot lane%16.
  // round r (0..3) of wave w covers tile rows (r*4 + w)*4 + lane/16
  const int lrow = lane >> 4, slot = lane & 15;
  const int mchunks = (p.M + 63) >> 6;
  const int cbeg = (int)((long long)mchunks * sm / p.split_m);
  const int cend = (int)((long long)mchunks * (sm + 1) / p.split_m);
  const char *zero = (const char *)g_zero_page_w;

  auto stage = [&](int chunk_idx, int buf) {
    char *pdst = lds + buf * W_STAGE_BYTES + wave * 1024;  // P tile (dY) first, Q tile (X) second
    char *qdst = pdst + W_TILE_BYTES;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int R = (r * 4 + wave) * 4 + lrow;
      const int lchunk = slot ^ (R & 15);      // logical 16-B chunk fetched into this slot
      const int m = chunk_idx * 64 + R;
      const char *ps = zero, *qs = zero;
      if (m < p.M) {
        const int b = fastdiv(m, p.magicHW, p.shiftHW);
        const int rem = m - b * p.Hm * p.Wm;
        const int qy = fastdiv(rem, p.magicW, p.shiftW);
        const int qx = rem - qy * p.Wm;
        // plain operand pixel == m; gathered operand pixel = (qy*S+oy, qx*S+ox) on its own grid
        const bool second = p.pair && lchunk * EPC >= 64;
        const int gy = qy * S + (second ? oy2 : oy), gx = qx * S + (second ? ox2 : ox);
        const int cop = co0 + lchunk * EPC, cip = p.pair ? (lchunk * EPC) & 63 : ci0 + lchunk * EPC;
        if (!transposed) {
          if (cop < p.Cout) ps = p.dy + ((int64_t)m * p.ldy + cop) * ES;
          if (cip < p.Cin && (!second || has2) && (unsigned)gy < (unsigned)p.Hx && (unsigned)gx < (unsigned)p.Wx)
            qs = p.x + ((int64_t)((b * p.Hx + gy) * p.Wx + gx) * p.ldx + cip) * ES;
        } else {
          if (cip < p.Cin) qs = p.x + ((int64_t)m * p.ldx + cip) * ES;
          if (cop < p.Cout && (unsigned)gy < (unsigned)p.Hy && (unsigned)gx < (unsigned)p.Wy)
            ps = p.dy + ((int64_t)((b * p.Hy + gy) * p.Wy + gx) * p.ldy + cop) * ES;
        }
      }
      glds16w(ps, pdst + r * 4096);
      glds16w(qs, qdst + r * 4096);
    }
  };

  // wave tile: 64 (co) x 64 (ci) for bf16 (4x4 frags); for f32 the tile is 64x64 total -> 32x32 per wave (2x2 frags)
  constexpr int FR = (ES == 2) ? 4 : 2;
  const int wco = (wave >> 1) * (FR * 16), wci = (wave & 1) * (FR * 16);
  f32x4 acc[FR][FR];
#pragma unroll
  for (int i = 0; i < FR; ++i)
#pragma unroll
    for (int j = 0; j < FR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;

  if (cbeg < cend) stage(cbeg, 0);
  for (int c = cbeg; c < cend; ++c) {
    const int buf = (c - cbeg) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (c + 1 < cend) stage(c + 1, buf ^ 1);
    const char *pb = lds + buf * W_STAGE_BYTES;
    const char *qb = pb + W_TILE_BYTES;
    if constexpr (ES == 2) {
      // k-step s covers rows m = 32 s .. 32 s + 31; lane (fq, fr): k = 8 fq + j.  Transposing read: within a
      // 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 and receives column `fr`.
      const int q = fr >> 2, pp = fr & 3;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 a[FR], b[FR];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = s * 32 + fq * 8 + h * 4 + q;
#pragma unroll
          for (int i = 0; i < FR; ++i) {
            const int ca = wco + i * 16 + pp * 4;   // column (channel) inside the tile row
            const int offa = row * 256 + ((((ca >> 3) ^ (row & 15)) << 4) | ((ca & 7) << 1));
            const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(pb + offa));
            const int cb = wci + i * 16 + pp * 4;
            const int offb = row * 256 + ((((cb >> 3) ^ (row & 15)) << 4) | ((cb & 7) << 1));
            const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(qb + offb));
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[i][h * 4 + e] = va[e]; b[i][h * 4 + e] = vb[e]; }
          }
        }
#pragma unroll
        for (int i = 0; i < FR; ++i)
#pragma unroll
          for (int j = 0; j < FR; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // f32 16x16x4: lane supplies A[co = fr][k = fq], B[k = fq][ci = fr]; 16 k-steps of 4 rows
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int row = s * 4 + fq;
        float a[FR], b[FR];
#pragma unroll
        for (int i = 0; i < FR; ++i) {
          const int ca = wco + i * 16 + fr;
          a[i] = *(const float *)(pb + row * 256 + ((((ca >> 2) ^ (row & 15)) << 4) | ((ca & 3) << 2)));
          const int cb = wci + i * 16 + fr;
          b[i] = *(const float *)(qb + row * 256 + ((((cb >> 2) ^ (row & 15)) << 4) | ((cb & 3) << 2)));
        }
#pragma unroll
        for (int i = 0; i < FR; ++i)
#pragma unroll
          for (int j = 0; j < FR; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- store / accumulate: lane holds dW[co = .. + fq*4 + r][ci = .. + fr]
  // Whole tiles that are STORED (no atomics, one tap per tile) leave through LDS as full rows of CT channels (256 B f32 /
  // 512 B bf16-mode): straight from the fragments a store instruction covers four 64-byte pieces, and the weight-gradient
  // launches whose output is most of their traffic (batch 1: U b, 67 MB for 8 pixels; DeltaP fc1, 268 MB) wrote at 0.7-2 TB/s.
  if (!p.pair && (p.slab || p.split_m == 1) && co0 + CT <= p.Cout && ci0 + CT <= p.Cin && (p.Cin & 3) == 0) {
    constexpr int PITCH = CT + 4;
    float *tile = (float *)lds;
    __syncthreads();   // every wave is done with the operand stages
#pragma unroll
    for (int i = 0; i < FR; ++i)
#pragma unroll
      for (int j = 0; j < FR; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tile[(wco + i * 16 + fq * 4 + r) * PITCH + wci + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int wtap = p.flip ? (p.ntaps - 1 - tap) : tap;
    float *dst = p.slab ? p.slab + (int64_t)sm * p.dw_elems : p.dw;
    constexpr int CPR = CT / 4;                       // 16-B chunks per row
    for (int e = tid; e < CT * CPR; e += 256) {
      const int row = e / CPR, c4 = (e - row * CPR) * 4;
      const f32x4 v = *(const f32x4 *)(tile + row * PITCH + c4);
      *(f32x4 *)(dst + ((int64_t)(co0 + row) * p.ntaps + wtap) * p.Cin + ci0 + c4) = v;
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < FR; ++i)
#pragma unroll
    for (int j = 0; j < FR; ++j) {
      int ci = ci0 + wci + j * 16 + fr;
      int tapx = tap;
      if (p.pair) {
        if (ci >= 64) { if (!has2) continue; tapx = tap + 1; ci -= 64; }
      } else if (ci >= p.Cin) continue;
      const int wtap = p.flip ? (p.ntaps - 1 - tapx) : tapx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wco + i * 16 + fq * 4 + r;
        if (co >= p.Cout) continue;
        const int64_t e = ((int64_t)co * p.ntaps + wtap) * p.Cin + ci;
        if (p.slab) p.slab[(int64_t)sm * p.dw_elems + e] = acc[i][j][r];
        else if (p.split_m > 1) atomicAdd(p.dw + e, acc[i][j][r]);
        else p.dw[e] = acc[i][j][r];
      }
    }
}

static void make_magic(int d, unsigned long long &magic, int &shift) {
  // round-up method: s = ceil(log2 d), magic = ceil(2^(32+s) / d) < 2^33; floor(m/d) == (m*magic) >> (32+s)
  // exactly for every 0 <= m < 2^32, and m*magic < 2^64 for m < 2^31
  shift = 0;
  while ((1LL << shift) < d) ++shift;
  const unsigned __int128 num = ((unsigned __int128)1 << (32 + shift)) + (unsigned)d - 1;
  magic = (unsigned long long)(num / (unsigned)d);
}

}  // namespace cfd

using namespace cfd;

/* declared in include/cfd_amd.h */
static int conv_wgrad_impl(const cfd_conv_desc *d, const void *x, const void *dy, float *dw, int split_m,
                           void *workspace, void *stream) {
  CFD_REQUIRE(d && x && dy && dw, "NULL pointer");
  CFD_REQUIRE(d->dtype == CFD_BF16 || d->dtype == CFD_F32, "bad dtype");
  CFD_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0 && d->Cout > 0, "non-positive shape");
  const int epc = d->dtype == CFD_BF16 ? 8 : 4;
  CFD_REQUIRE(d->Cin % epc == 0 && d->Cout % epc == 0 && d->ldx % epc == 0 && d->ldy % epc == 0,
              "wgrad: Cin/Cout/ldx/ldy must be multiples of %d (Cin=%d Cout=%d)", epc, d->Cin, d->Cout);
  CFD_REQUIRE(split_m >= 1, "split_m must be >= 1");
  {
    Wgrad256Params p2;
    if (wgrad256_selected(d, p2)) return wgrad256_launch(p2, x, dy, dw, split_m, (float *)workspace, (hipStream_t)stream);
  }
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char *)x; p.dy = (const char *)dy; p.dw = dw;
  p.B = d->B; p.Hx = d->Hi; p.Wx = d->Wi; p.Cin = d->Cin; p.Cout = d->Cout; p.ldx = d->ldx; p.ldy = d->ldy;
  p.kind = d->kind; p.flip = 0;
  switch (d->kind) {
    case CFD_CONV_K4S2: p.Hy = d->Hi / 2; p.Wy = d->Wi / 2; p.ntaps = 16; p.Hm = p.Hy; p.Wm = p.Wy; break;
    case CFD_CONV_K4S1: p.Hy = d->Hi; p.Wy = d->Wi; p.ntaps = 16; p.Hm = p.Hy; p.Wm = p.Wy; break;
    case CFD_CONV_K3S1: p.Hy = d->Hi; p.Wy = d->Wi; p.ntaps = 9; p.Hm = p.Hy; p.Wm = p.Wy; break;
    case CFD_CONVT_K4S2: p.Hy = d->Hi * 2; p.Wy = d->Wi * 2; p.ntaps = 16; p.Hm = d->Hi; p.Wm = d->Wi; break;
    case CFD_DENSE:
      CFD_REQUIRE(d->Hi == 1 && d->Wi == 1, "dense needs Hi=Wi=1");
      p.Hy = 1; p.Wy = 1; p.ntaps = 1; p.Hm = 1; p.Wm = 1; break;
    default: CFD_FAIL(CFD_EINVAL, "wgrad: unknown kind %d", d->kind);
  }
  const int64_t M = (int64_t)d->B * p.Hm * p.Wm;
  CFD_REQUIRE(M < (1LL << 31) && (int64_t)d->B * p.Hy * p.Wy < (1LL << 31), "too many pixels");
  p.M = (int)M;
  const int ct = d->dtype == CFD_BF16 ? 128 : 64;
  p.cotiles = cdiv(d->Cout, ct);
  p.citiles = cdiv(d->Cin, ct);
  p.pair = d->dtype == CFD_BF16 && d->Cin == 64 &&
           (d->kind == CFD_CONV_K4S2 || d->kind == CFD_CONV_K4S1 || d->kind == CFD_CONV_K3S1);
  p.tap_groups = p.pair ? (p.ntaps + 1) / 2 : p.ntaps;
  const int mchunks = (p.M + 63) / 64;
  p.split_m = split_m < mchunks ? split_m : mchunks;
  make_magic(p.Wm, p.magicW, p.shiftW);
  make_magic(p.Hm * p.Wm, p.magicHW, p.shiftHW);
  const int64_t nb = (int64_t)p.cotiles * p.citiles * p.tap_groups * p.split_m;
  CFD_REQUIRE(nb < (1LL << 31), "grid too large");
  p.nblocks = (int)nb;
  // split-M partial results as slabs when the caller brought the workspace for them (cfd_conv_wgrad_workspace_bytes): a small
  // weight tensor with a long reduction (DeltaP g3: 73 728 weights, 524 288 pixels at B = 64, 128 slices) otherwise takes
  // ~10 M fp32 atomics onto 73 K addresses
  p.dw_elems = (int64_t)d->Cout * p.ntaps * d->Cin;
  p.slab = (p.split_m > 1 && workspace && p.dw_elems % 4 == 0) ? (float *)workspace : nullptr;
  hipStream_t st = (hipStream_t)stream;
  // (the staged epilogue's tile: CT x (CT + 4) floats -- 66 KiB in bf16 mode, a little over the two operand stages)
  const size_t ctile = d->dtype == CFD_BF16 ? 128 : 64;
  const size_t shmem = std::max<size_t>(2 * W_STAGE_BYTES, ctile * (ctile + 4) * 4);
  if (d->dtype == CFD_BF16) {
    if (int e_ = set_max_lds((const void *)wgrad_kernel<bf16>, (int)shmem)) return e_;
    hipLaunchKernelGGL(wgrad_kernel<bf16>, dim3(p.nblocks), dim3(256), shmem, st, p);
  } else {
    if (int e_ = set_max_lds((const void *)wgrad_kernel<float>, (int)shmem)) return e_;
    hipLaunchKernelGGL(wgrad_kernel<float>, dim3(p.nblocks), dim3(256), shmem, st, p);
  }
  CFD_LAUNCH_CHECK();
  if (p.slab) return wgrad_slab_reduce_launch(p.slab, dw, p.dw_elems / 4, p.split_m, st);
  return CFD_OK;
}

// bytes of split-M slabs the 128 x 128 kernel can use (0: none -- atomics): small weight tensors only
static int64_t wgrad128_slab_bytes(const cfd_conv_desc *d, int split_m) {
  if (split_m <= 1) return 0;
  const int ntaps = d->kind == CFD_CONV_K3S1 ? 9 : (d->kind == CFD_DENSE ? 1 : 16);
  const int64_t elems = (int64_t)d->Cout * ntaps * d->Cin;
  const int64_t bytes = elems * 4 * split_m;
  return (elems % 4 == 0 && bytes <= (256LL << 20)) ? bytes : 0;
}

extern "C" int cfd_conv_wgrad(const cfd_conv_desc *d, const void *x, const void *dy, float *dw, int split_m,
                              void *stream) {
  return conv_wgrad_impl(d, x, dy, dw, split_m, nullptr, stream);
}

extern "C" int cfd_conv_wgrad_ws(const cfd_conv_desc *d, const void *x, const void *dy, float *dw, int split_m,
                                 void *workspace, void *stream) {
  return conv_wgrad_impl(d, x, dy, dw, split_m, workspace, stream);
}

extern "C" int64_t cfd_conv_wgrad_workspace_bytes(const cfd_conv_desc *d, int split_m) {
  Wgrad256Params p2;
  if (d && wgrad256_selected(d, p2)) return wgrad256_workspace_bytes(p2, split_m);
  return d ? wgrad128_slab_bytes(d, split_m) : 0;
}

extern "C" int cfd_conv_wgrad_suggest_split(const cfd_conv_desc *d) {
  if (!d) return 1;
  {
    Wgrad256Params p2;
    if (wgrad256_selected(d, p2)) return wgrad256_suggest_split(p2);
  }
  int ntaps, Hm, Wm;
  switch (d->kind) {
    case CFD_CONV_K4S2: ntaps = 16; Hm = d->Hi / 2; Wm = d->Wi / 2; break;
    case CFD_CONV_K3S1: ntaps = 9; Hm = d->Hi; Wm = d->Wi; break;
    case CFD_DENSE: ntaps = 1; Hm = 1; Wm = 1; break;
    default: ntaps = 16; Hm = d->Hi; Wm = d->Wi; break;
  }
  const int ct = d->dtype == CFD_BF16 ? 128 : 64;
  const bool pair = d->dtype == CFD_BF16 && d->Cin == 64 &&
                    (d->kind == CFD_CONV_K4S2 || d->kind == CFD_CONV_K4S1 || d->kind == CFD_CONV_K3S1);
  const int64_t tiles = (int64_t)cdiv(d->Cout, ct) * cdiv(d->Cin, ct) * (pair ? (ntaps + 1) / 2 : ntaps);
  const int64_t mchunks = ((int64_t)d->B * Hm * Wm + 63) / 64;
  int s = 1;
  while (tiles * s * 2 <= 1024 && mchunks / (s * 2) >= 8) s *= 2;
  return s;
}
