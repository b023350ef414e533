// bf16 weight gradients on the 256x256 ping-pong structure of igemm256.hip.
//
// Per kernel tap t:   dW[co][t][ci] = sum_m  P_t[m][co] * Q_t[m][ci]        (see wgrad.hip for the GEMM view)
//   conv kinds : P = dY rows (plain),  Q = X gathered at (qy*S + oy, qx*S + ox)
//   transposed : Q = X rows (plain),   P = dY gathered at (2 qy + kh - 1, 2 qx + kw - 1)
//   dense      : both plain
// Output tile 256 (co) x 256 (ci) of one tap; the reduction index m is the K dimension, 64 rows per K tile.
// Both operands are staged in their natural row-major form ([m][channels], 256 B per half-tile row) and the
// fragments are read with the transposing LDS read ds_read_b64_tr_b16, as in wgrad.hip.
//
// Schedule = igemm256's two-phase ping-pong (8 waves, waves 4..7 one barrier behind, LDS-DMA in flight across
// barriers with counted vmcnt):
//   half tiles per K tile: slot 0 = A0 (P, co 0..127), 1 = B0 (Q, ci 0..127), 2 = B1 (Q, ci 128..255), 3 = A1
//   phase A: read A0, B0, B1; issue B1, A1 of K tile k+1;  32 MFMAs A0 x (B0 | B1)
//   phase B: read A1;         issue A0, B0 of K tile k+2;  32 MFMAs A1 x (B0 | B1)
// Addressing (needs power-of-two reduction-grid sides, true for the reference's 64x512 and 128x1024 grids):
// a K tile is 64 consecutive, 64-aligned m, so m = (64 kt) | R splits into a scalar and a per-lane part with no
// carries.  The plain operand is voffset (lane constant) + soffset (scalar); the gathered operand's offset is
// lane constant + scalar as well, and only the in-image test costs VALU (out-of-image rows get an out-of-range
// voffset, which buffer_load zero-fills).  The gathered buffer descriptor starts MARGIN bytes before the tensor so
// that lane constants with a negative tap offset stay non-negative.
#include "common.h"
#include "igemm.h"
#include "wgrad256.h"

namespace cfd {

constexpr int WHT = 64 * 256;       // half tile: 64 rows (m) x 256 B (128 channels)
constexpr int WST = 4 * WHT;        // stage = four half tiles = 64 KiB
// LDS map: the A half tiles of both stages fill the first 64 KiB, the B half tiles the second, so that every
// fragment read of one operand lies within the 16-bit immediate range of ONE base register.
__host__ __device__ constexpr int wslot_off(int st, int sl) {
  return (sl == 0 || sl == 3) ? (st * 2 + (sl == 3 ? 1 : 0)) * WHT : 4 * WHT + (st * 2 + (sl == 2 ? 1 : 0)) * WHT;
}
constexpr unsigned WOOB = 0x80000000u;


__device__ __forceinline__ void wbload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char *lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_dst, 16, (int)voff,
                                           (int)soff, 0, 0);
}

__device__ __forceinline__ bf16x8 tr_read2(const char *lds, unsigned o0, unsigned o1) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(lds + o0));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4 *)(lds + o1));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// PAIR (conv kinds with Cin == 128: the discriminator's c2, DeltaP g4): the 256 ci columns of a tile are TWO consecutive
// taps x 128 channels -- B0 = tap t, B1 = tap t + 1 (the same 128 channels at that tap's own pixel offset, with its own
// in-image test); dW's [tap][ci] columns are contiguous, so the epilogue is unchanged except that an odd tap count (3x3)
// leaves the last pair without a second tap: its B1 loads nothing and its columns are not stored.
template <bool GATHER_A, bool PAIR>
__global__ __launch_bounds__(512, 2) void wgrad256_kernel(Wgrad256Params p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- block -> (tap, ci tile, co tile, m slice); taps fastest so that blocks sharing operand rows share an XCD
  int bid = blockIdx.x;
  {
    const int nwg = p.nblocks, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tap = (bid % p.tap_groups) * (PAIR ? 2 : 1); bid /= p.tap_groups;
  const int cit = bid % p.citiles; bid /= p.citiles;
  const int cot = bid % p.cotiles;
  const int sm = bid / p.cotiles;
  const int co0 = cot * 256, ci0 = cit * 256;
  const int kh = tap / p.ktaps_w, kw = tap - kh * p.ktaps_w;
  const int oy = kh + p.oy_base, ox = kw + p.ox_base;
  const bool has2 = PAIR && tap + 1 < p.ntaps;
  const int kh2 = (tap + 1) / p.ktaps_w, kw2 = (tap + 1) - kh2 * p.ktaps_w;
  const int oy2 = kh2 + p.oy_base, ox2 = kw2 + p.ox_base;

  const int ktiles = (p.M + 63) >> 6;
  const int kbeg = k_slice(ktiles, sm, p.split_m);       // 32-bit: ktiles * split_m < 2^31 (host)
  const int kend = k_slice(ktiles, sm + 1, p.split_m);
  const int nkt = kend - kbeg;

  const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void *)p.plain, 0, (int)p.pbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void *)p.gath, 0, (int)p.gbytes, 0x00020000);

  // ---- staging geometry: a half tile = 16 wave-instructions of 1 KiB (4 rows x 256 B); wave w issues pieces w and
  // 8 + w: rows R_i = (i*8 + w)*4 + lane/16, physical 16-B slot lane%16 holding LOGICAL chunk slot ^ (R & 15)
  const int lrow = lane >> 4, slot = lane & 15;
  const int cplain = GATHER_A ? ci0 : co0, cgath = GATHER_A ? co0 : ci0;
  unsigned pv[2], gl[2], gl2[2], cur[2], cur1[2];
  int gyl[2], gxl[2], gyl2[2], gxl2[2];
  const int Wm_mask = (1 << p.lw) - 1, Hm_mask = (1 << p.lh) - 1;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int R = (i * 8 + wave) * 4 + lrow;
    const int lchunk = slot ^ (R & 15);
    pv[i] = (unsigned)((R * p.ldp + cplain + lchunk * 8) * 2);
    const int qxl = R & Wm_mask, qyl = (R >> p.lw) & Hm_mask, bl = R >> (p.lw + p.lh);
    gyl[i] = qyl * p.S + oy;
    gxl[i] = qxl * p.S + ox;
    gl[i] = (unsigned)((((bl * p.Hg + gyl[i]) * p.Wg + gxl[i]) * p.ldg + cgath + lchunk * 8) * 2 + (int)p.margin);
    gyl2[i] = qyl * p.S + oy2;
    gxl2[i] = qxl * p.S + ox2;
    gl2[i] = (unsigned)((((bl * p.Hg + gyl2[i]) * p.Wg + gxl2[i]) * p.ldg + cgath + lchunk * 8) * 2 + (int)p.margin);
  }

  // ---- issue stream state (uniform)
  int iss_kt = kbeg, iss_left = nkt;
  unsigned s_p = 0, s_g = 0;
  auto iss_update = [&]() {
    const int ms = iss_kt << 6;
    s_p = (unsigned)ms * (unsigned)(p.ldp * 2);
    const int qxs = ms & Wm_mask, qys = (ms >> p.lw) & Hm_mask, bs = ms >> (p.lw + p.lh);
    const int gys = qys * p.S, gxs = qxs * p.S;
    s_g = (unsigned)(((bs * p.Hg + gys) * p.Wg + gxs) * p.ldg * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      cur[i] = (unsigned)(gyl[i] + gys) < (unsigned)p.Hg && (unsigned)(gxl[i] + gxs) < (unsigned)p.Wg ? gl[i] : WOOB;
      if (PAIR)
        cur1[i] = has2 && (unsigned)(gyl2[i] + gys) < (unsigned)p.Hg && (unsigned)(gxl2[i] + gxs) < (unsigned)p.Wg ? gl2[i] : WOOB;
    }
  };
  auto iss_advance = [&]() {
    if (iss_left > 1) { --iss_left; ++iss_kt; iss_update(); }
  };
  iss_update();

  char *const ldsw = lds + wave * 1024;
  auto issue = [&](int sl, int stage) {   // compile-time constants at every call site
    char *dst = ldsw + wslot_off(stage, sl);
    const bool is_a = (sl == 0 || sl == 3);
    const unsigned half = (sl == 2 || sl == 3) ? 256u : 0u;
    if (PAIR && sl == 2) {
      wbload16(grs, cur1[0], s_g, dst); wbload16(grs, cur1[1], s_g, dst + 8192);
    } else if (is_a == GATHER_A) {
      wbload16(grs, cur[0], s_g + half, dst); wbload16(grs, cur[1], s_g + half, dst + 8192);
    } else {
      wbload16(prs, pv[0], s_p + half, dst); wbload16(prs, pv[1], s_p + half, dst + 8192);
    }
  };

  // ---- fragment read addresses.  k-step s covers tile rows 32 s .. 32 s + 31; lane (fq, fr): k = 8 fq + e.
  // Transposing read: within a 16-lane group, lane 4q+pp supplies the address of row q, columns 4pp..4pp+3.
  const int fr = lane & 15, fq = lane >> 4;
  const int q = fr >> 2, pp = fr & 3;
  // Byte offset of the read (h, i, s) inside a half tile:
  //   row = 32 s + 8 fq + 4 h + q,  chunk = (wr*8 + 2 i + pp/2) ^ (row & 15),  row & 15 = 8 (fq&1) + 4 h + q
  //   => off = [ (8 fq + q)*256 + ((wr*8 + pp/2) ^ (8 (fq&1) + q))*16 + (pp&1)*8 ]  ^  ((i ^ 2h) << 5)   (bits 5,6)
  //            + h*1024 + s*8192                                                          (bits 10, 13: free)
  // so ONE lane base per operand XOR-ed with {0,32,64,96} gives every address; the rest are instruction
  // immediates (< 64 KiB inside an operand's LDS region).  Four variants x two stages x two operands = 16 address registers,
  // no address arithmetic in the loop.
  unsigned aX[4], bX[4];
  {
    const int row0 = fq * 8 + q;
    const unsigned abase = (unsigned)(row0 * 256 + ((((wr * 8 + (pp >> 1)) ^ (row0 & 15)) << 4) | ((pp & 1) << 3)));
    const unsigned bbase = (unsigned)(row0 * 256 + ((((wc * 4 + (pp >> 1)) ^ (row0 & 15)) << 4) | ((pp & 1) << 3)));
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      aX[v] = abase ^ (unsigned)(v << 5);
      bX[v] = (bbase ^ (unsigned)(v << 5)) + 4 * WHT;   // base of the B region
      asm volatile("" : "+v"(aX[v]));   // opaque: keep exactly these 8 values, fold the rest as immediates
      asm volatile("" : "+v"(bX[v]));
    }
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: H0 .. H5; H0..H2 must have landed
#pragma unroll
  for (int h = 0; h < 6; ++h) {
    issue(h & 3, (h >> 2) & 1);
    if ((h & 3) == 3) iss_advance();
  }
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // second group runs one barrier behind

  bf16x8 a[4][2], b0[2][2], b1[2][2];

#define W_PHASE(SLOTA, STA, SLOTB, STB, VMASM, READS, MFMAS)                           \
    {                                                                                  \
      READS                                                                            \
      issue(SLOTA, STA); issue(SLOTB, STB);                                            \
      if (SLOTB == 3) iss_advance();                                                   \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
      asm volatile(VMASM ::: "memory");                                                \
      __builtin_amdgcn_s_barrier();                                                    \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_setprio(1);                                                   \
      MFMAS                                                                            \
      __builtin_amdgcn_s_setprio(0);                                                   \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_barrier();                                                    \
    }
#define W_MFMA(AI, BJ, BREG)                                                           \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                \
            acc[AI + i][BJ + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][s], BREG[j][s], acc[AI + i][BJ + j], 0, 0, 0);
#define W_KTILE(ST)                                                                    \
    W_PHASE(2, (ST + 1) & 1, 3, (ST + 1) & 1, "s_waitcnt vmcnt(8)",                    \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
          b0[j][s] = tr_read2(lds, bX[j] + (wslot_off(ST, 1) - 4 * WHT + s * 8192), bX[j ^ 2] + (wslot_off(ST, 1) - 4 * WHT + s * 8192 + 1024)); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          a[i][s] = tr_read2(lds, aX[i] + (wslot_off(ST, 0) + s * 8192), aX[i ^ 2] + (wslot_off(ST, 0) + s * 8192 + 1024)); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
          b1[j][s] = tr_read2(lds, bX[j] + (wslot_off(ST, 2) - 4 * WHT + s * 8192), bX[j ^ 2] + (wslot_off(ST, 2) - 4 * WHT + s * 8192 + 1024)); \
      },                                                                               \
      W_MFMA(0, 0, b0) W_MFMA(0, 2, b1))                                               \
    W_PHASE(0, ST, 1, ST, "s_waitcnt vmcnt(6)",                                        \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          a[i][s] = tr_read2(lds, aX[i] + (wslot_off(ST, 3) + s * 8192), aX[i ^ 2] + (wslot_off(ST, 3) + s * 8192 + 1024)); \
      ,                                                                                \
      W_MFMA(4, 0, b0) W_MFMA(4, 2, b1))

  // pairs + odd tail (a mid-loop break made the register allocator rotate the accumulators and spill)
  for (int k = 0; k + 1 < nkt; k += 2) {
    W_KTILE(0)
    W_KTILE(1)
  }
  if (nkt & 1) { W_KTILE(0) }
#undef W_KTILE
#undef W_MFMA
#undef W_PHASE
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail loads must not outlive the workgroup's LDS allocation

  // ---- store / accumulate: lane holds dW[co = .. + fq*4 + r][ci = .. + fr].  The destination (dw, this block's slab, or
  // atomics into dw) is chosen ONCE: a per-element three-way branch here compiled to ~400 scalar branches per wave.
  float *const dst = p.split_m == 1 ? p.dw : (p.slab ? p.slab + (int64_t)sm * p.dw_elems : p.dw);
  const bool atomics = p.split_m > 1 && !p.slab;
  // 32-bit element offsets inside the (< 2^31 element) gradient tensor: one multiply per row, adds after that
  const int row_stride = p.ntaps * p.Cin;
  const int base = (co0 + wr * 64 + fq * 4) * row_stride + tap * p.Cin + ci0 + wc * 32 + fr;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rowoff = base + ((i >> 2) * 128 + (i & 3) * 16 + r) * row_stride;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (PAIR && (j >> 1) && !has2) continue;      // (uniform) the odd last tap has no partner: those columns do not exist
        float *q = dst + rowoff + (j >> 1) * 128 + (j & 1) * 16;
        if (atomics) atomicAdd(q, acc[i][j][r]);
        else *q = acc[i][j][r];
      }
    }
}

// dw[i] = sum_s slab[s][i]: the split-M partial tiles, written with plain stores, are summed in a fixed order
// (bitwise reproducible; a 256x256 fp32 tile per block as atomics would cost more than 30 K tiles of MFMA time)
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float *__restrict__ slab, float *__restrict__ dw,
                                                               int64_t n4, int split) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 a = ((const f32x4 *)slab)[i];
    for (int s = 1; s < split; ++s) a += ((const f32x4 *)slab)[(int64_t)s * n4 + i];
    ((f32x4 *)dw)[i] = a;
  }
}

int wgrad_slab_reduce_launch(const float *slab, float *dw, int64_t n4, int split, hipStream_t st) {
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, slab, dw, n4, split);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// fills `p` (without pointers) and returns true when the 256x256 kernel can run this layer
bool wgrad256_plan(const cfd_conv_desc *d, Wgrad256Params &p) {
  memset(&p, 0, sizeof(p));
  const bool pair = d->Cin == 128 && (d->kind == CFD_CONV_K4S2 || d->kind == CFD_CONV_K4S1 || d->kind == CFD_CONV_K3S1);
  if (d->dtype != CFD_BF16 || (d->Cin % 256 && !pair) || d->Cout % 256) return false;
  int Hy, Wy, Hm, Wm;
  switch (d->kind) {
    case CFD_CONV_K4S2: Hy = d->Hi / 2; Wy = d->Wi / 2; p.ntaps = 16; p.ktaps_w = 4; p.S = 2; p.oy_base = -1; p.ox_base = -1; Hm = Hy; Wm = Wy; break;
    case CFD_CONV_K4S1: Hy = d->Hi; Wy = d->Wi; p.ntaps = 16; p.ktaps_w = 4; p.S = 1; p.oy_base = -1; p.ox_base = -1; Hm = Hy; Wm = Wy; break;
    case CFD_CONV_K3S1: Hy = d->Hi; Wy = d->Wi; p.ntaps = 9; p.ktaps_w = 3; p.S = 1; p.oy_base = -1; p.ox_base = -1; Hm = Hy; Wm = Wy; break;
    case CFD_CONVT_K4S2: Hy = d->Hi * 2; Wy = d->Wi * 2; p.ntaps = 16; p.ktaps_w = 4; p.S = 2; p.oy_base = -1; p.ox_base = -1; Hm = d->Hi; Wm = d->Wi; break;
    case CFD_DENSE: Hy = 1; Wy = 1; p.ntaps = 1; p.ktaps_w = 1; p.S = 1; p.oy_base = 0; p.ox_base = 0; Hm = 1; Wm = 1; break;
    default: return false;
  }
  if (!is_pow2(Hm) || !is_pow2(Wm)) return false;
  p.gather_a = d->kind == CFD_CONVT_K4S2;
  p.lw = ilog2(Wm); p.lh = ilog2(Hm);
  p.Cin = d->Cin; p.Cout = d->Cout;
  const int64_t M = (int64_t)d->B * Hm * Wm;
  if (M >= (1LL << 31) - 64) return false;
  p.M = (int)M;
  int64_t pb, gb;
  if (p.gather_a) {   // plain = X (input grid == reduction grid), gathered = dY (output grid)
    p.ldp = d->ldx; p.ldg = d->ldy; p.Hg = Hy; p.Wg = Wy;
    pb = ((M - 1) * d->ldx + d->Cin) * 2;
    gb = (((int64_t)d->B * Hy * Wy - 1) * d->ldy + d->Cout) * 2;
  } else {            // plain = dY, gathered = X
    p.ldp = d->ldy; p.ldg = d->ldx; p.Hg = d->Hi; p.Wg = d->Wi;
    pb = ((M - 1) * d->ldy + d->Cout) * 2;
    gb = (((int64_t)d->B * d->Hi * d->Wi - 1) * d->ldx + d->Cin) * 2;
  }
  const int64_t margin = (int64_t)(p.Wg + 1) * p.ldg * 2;
  // every offset the kernel forms (incl. the 64-row tail past M and the lane constants) must stay below 2^31
  const int64_t slack = (int64_t)64 * (p.ldp > p.ldg ? p.ldp : p.ldg) * 2 * 4 + margin * 4;
  if (pb + slack >= (1LL << 31) || gb + margin + slack >= (1LL << 31)) return false;
  p.pbytes = (unsigned)pb; p.gbytes = (unsigned)(gb + margin); p.margin = (unsigned)margin;
  p.pair = pair;
  p.cotiles = d->Cout / 256; p.citiles = pair ? 1 : d->Cin / 256;
  p.tap_groups = pair ? (p.ntaps + 1) / 2 : p.ntaps;
  return true;
}

int wgrad256_suggest_split(const Wgrad256Params &p) {
  const int tiles = p.cotiles * p.citiles * p.tap_groups;
  const int ktiles = (p.M + 63) / 64;
  int s = 1;
  while (tiles * s * 2 <= 512 && ktiles / (s * 2) >= 16) s *= 2;
  // tile counts that are no power of two (nine taps, five tap pairs): doubling lands on 288 or 320 blocks, i.e. a
  // second round on a quarter of the CUs; take the largest factor that still fits two full rounds instead
  if ((tiles & (tiles - 1)) != 0 && tiles < 512) {
    int t = 512 / tiles;
    if (t > ktiles / 16) t = ktiles / 16;
    if (t > s) s = t;
  }
  return s;
}

bool wgrad256_selected(const cfd_conv_desc *d, Wgrad256Params &p) {
  if (!wgrad256_plan(d, p)) return false;
  const int mode = tuning(CFD_TUNE_WGRAD256);
  if (mode >= 0) return mode != 0;
  return true;   // measured (scripts/layer_bench.py --modes wgrad): wins on every eligible layer, B = 16 and 64
}

int64_t wgrad256_workspace_bytes(const Wgrad256Params &p, int split_m) {
  return split_m > 1 ? (int64_t)split_m * p.Cout * p.ntaps * p.Cin * 4 : 0;
}

int wgrad256_launch(Wgrad256Params &p, const void *x, const void *dy, float *dw, int split_m, float *workspace,
                    hipStream_t st) {
  const char *plain = (const char *)(p.gather_a ? x : dy);
  const char *gath = (const char *)(p.gather_a ? dy : x);
  p.plain = plain;
  p.gath = gath - p.margin;
  p.dw = dw;
  const int ktiles = (p.M + 63) / 64;
  p.split_m = split_m < ktiles ? split_m : ktiles;
  CFD_REQUIRE((int64_t)ktiles * p.split_m < (1LL << 31), "reduction too long for 32-bit slice arithmetic");
  p.nblocks = p.cotiles * p.citiles * p.tap_groups * p.split_m;
  p.dw_elems = (int64_t)p.Cout * p.ntaps * p.Cin;
  CFD_REQUIRE(p.dw_elems < (1LL << 31), "weight tensor too large for 32-bit element offsets");
  p.slab = p.split_m > 1 ? workspace : nullptr;
  const size_t shmem = 2 * WST;
  {
    const void *fn = p.gather_a ? (const void *)wgrad256_kernel<true, false>
                                : (p.pair ? (const void *)wgrad256_kernel<false, true> : (const void *)wgrad256_kernel<false, false>);
    if (int e_ = set_max_lds(fn, (int)shmem)) return e_;
  }
  CFD_REQUIRE(!(p.pair && p.gather_a), "tap pairs need a gathered ci-side operand");
  if (p.gather_a) hipLaunchKernelGGL((wgrad256_kernel<true, false>), dim3(p.nblocks), dim3(512), shmem, st, p);
  else if (p.pair) hipLaunchKernelGGL((wgrad256_kernel<false, true>), dim3(p.nblocks), dim3(512), shmem, st, p);
  else hipLaunchKernelGGL((wgrad256_kernel<false, false>), dim3(p.nblocks), dim3(512), shmem, st, p);
  CFD_LAUNCH_CHECK();
  if (p.slab) {
    const int64_t n4 = p.dw_elems / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(wgrad_slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, p.slab, dw, n4, p.split_m);
    CFD_LAUNCH_CHECK();
  }
  return CFD_OK;
}

}  // namespace cfd
