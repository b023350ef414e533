// bf16 implicit-GEMM convolution, 256(n) x 256(m) ping-pong tile, with the ACTIVATION operand staged once per 64-channel
// chunk as an input patch instead of once per tap.
//
// Why (note (f)): the PMC counters show igemm256's L2 channels 90 % busy at ~11.8 TB/s of requests --
// the kernel is bound by the bytes it fills into LDS (64 KiB per K tile: 32 KiB of weights + 32 KiB of gathered
// activations), not by the MFMA pipe or HBM.  The gathered operand of a stride-1 tap grid is the same input pixels shifted
// by the tap: a tile of 8 x 32 output pixels needs an (8 + kh - 1) x (32 + kw - 1) input patch, which is loaded ONCE per
// chunk (43 KiB for 3x3, 39 KiB for one sub-pixel class of the transposed 4x4) and read by the MFMA side at tap-shifted
// rows.  Fill per K tile falls from 64 KiB to 32 + 43/9 = 37 KiB (3x3) and 32 + 39/4 = 42 KiB (transposed).
//
// Differences from igemm256_kernel (igemm256.hip), whose structure (8 waves, two groups one barrier apart, two phases
// of 32 MFMAs per K tile, LDS-DMA with counted vmcnt, XOR-swizzled 128-B rows) is kept:
//   * m tile = 8 rows x 32 columns of ONE sample's class grid (host: Hq % 8 == 0, Wq % 32 == 0), m local = ty*32 + tx;
//     wave (wr, wc) owns tile rows wc and 4 + wc (its four B fragments = 16-pixel runs of those rows).
//   * K order is chunk-major: K tile k = chunk * NT + tap.  LDS = two stages of (A0, A1) weight half tiles (64 KiB) + two
//     patch stages.  A patch row is one patch pixel (128 B = 64 channels), pixel pp = py * 34 + px, its 16-B chunks
//     swizzled by pp & 7; the B fragment of tap (dy, dx) for tile pixel (ty, tx) is patch pixel (ty + dy - dymin) * 34 +
//     (tx + dx - dxmin): ONE lane base address per tap, the four fragments at immediate offsets (the row pitch 34 keeps
//     4 * 34 = 0 mod 8, so the swizzle key is the same for all four).
//   * the patch of chunk c + 1 is loaded during the first K tiles of chunk c (1 or 2 extra loads per wave per K tile, none
//     in the chunk's last K tile, so the wait that ends it -- vmcnt(4) -- retires the whole patch); padding = patch pixels
//     outside the image get an out-of-range voffset, which the hardware zero-fills.
//   * vmcnt bookkeeping: phase A of K tile k issues [patch loads n_t] + A1(k+1), phase B issues A0(k+2); both phases end
//     with vmcnt(4 + n_t), which retires A1(k) / A0(k+1) respectively.
//   * no split-K (the host keeps those layers on igemm256_kernel); the pooled epilogue works on the 8 x 32 tile.
#include "common.h"
#include "igemm.h"
#include "igemm256.h"
#include <type_traits>

namespace cfd {

constexpr int H_TH = 8, H_TW = 32, H_PW = 34;
constexpr int H_ABYTES = 4 * HT_BYTES;   // two stages of (A0, A1)
constexpr int H_BROW = 4 * H_PW * 128;   // byte distance between a wave's two tile rows (wc and 4 + wc) in the patch

template <int NT, bool S2 = false> struct HaloGeo {
  static constexpr int PH = H_TH + ((NT == 9 || S2) ? 2 : 1);
  static constexpr int PP = PH * H_PW;           // patch pixels
  static constexpr int NPI = (PP + 7) / 8;       // wave-level load instructions (8 pixels = 1 KiB each)
  static constexpr int PPW = (NPI + 7) / 8;      // ... per wave
  static constexpr int PER = (PPW + NT - 2) / (NT - 1);   // per K tile, so that the chunk's last K tile carries none
  static constexpr int PATCH = NPI * 1024;
  static constexpr int BIAS_OFF = H_ABYTES + 2 * PATCH;   // the n tile's bias (1 KiB), fetched with the first loads
  static constexpr int LDS = BIAS_OFF + 1024;
  static constexpr int first(int t) { return t * PER < PPW ? t * PER : PPW; }
  static constexpr int count(int t) { return first(t + 1) - first(t); }
  static_assert(first(NT - 1) == PPW, "the last K tile of a chunk must not carry patch loads");
  static_assert(LDS <= 160 * 1024 && BIAS_OFF >= OUT_LDS, "LDS budget (the staged epilogue must not reach the bias)");
};

struct IgemmHaloParams {
  IgemmParams g;
  unsigned xbytes, wbytes;
  int cls_fast, stage_out;
  int toff[4][9];      // per class, per tap: pixel offset of the tap inside the patch, (dy - dymin) * 34 + (dx - dxmin)
  int wtoff[4][9];     // per class, per tap: byte offset of the tap's Cin weights inside a weight row
  int dymin[4], dxmin[4];
  int tps, tpr;        // m tiles per sample, per tile row
  unsigned dv[5][3];   // exact unsigned division by ntiles, nclass, mtiles, tps, tpr: (multiplier, shift 1, shift 2)
};

// tile decode shared by the two kernels.  Every kernel-argument field the set-up needs is fetched in ONE batch up front
// (the empty asm pins the values there): left to itself the compiler loads each field right before its first use, which
// made the set-up a chain of seven dependent scalar-load round trips (2 K tiles' worth of time per tile, stamp build).
struct HaloTile { int nt, mt, cls, b, ty0, tx0; };
__device__ __forceinline__ HaloTile halo_tile(const IgemmHaloParams &P) {
  const IgemmParams &p = P.g;
  const int nblocks = p.nblocks, ntiles = p.ntiles, nclass = p.nclass, mtiles = p.mtiles, cls_fast = P.cls_fast;
  const int tps = P.tps, tpr = P.tpr;
  unsigned dv[5][3];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) dv[i][j] = P.dv[i][j];
  {   // (used by the caller right after: buffer descriptors, patch and weight addresses)
    const char *x = p.x, *w = p.w;
    const unsigned xb = P.xbytes, wb = P.wbytes;
    const int Hi = p.Hi, Wi = p.Wi, ldx = p.ldx, Kw = p.Kw, cpt = p.cpt;
    asm volatile("" ::"s"(x), "s"(w), "s"(xb), "s"(wb), "s"(Hi), "s"(Wi), "s"(ldx), "s"(Kw), "s"(cpt));
  }
  asm volatile("" ::"s"(nblocks), "s"(ntiles), "s"(nclass), "s"(mtiles), "s"(cls_fast), "s"(tps), "s"(tpr), "s"(dv[0][0]),
               "s"(dv[0][1]), "s"(dv[0][2]), "s"(dv[1][0]), "s"(dv[1][1]), "s"(dv[1][2]), "s"(dv[2][0]), "s"(dv[2][1]),
               "s"(dv[2][2]), "s"(dv[3][0]), "s"(dv[3][1]), "s"(dv[3][2]), "s"(dv[4][0]), "s"(dv[4][1]), "s"(dv[4][2]));
  // XCD-aware block remap (bijective), as in igemm256_kernel
  int bid = blockIdx.x;
  {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  HaloTile t;
  int q = fastdiv(bid, dv[0]);
  t.nt = bid - q * ntiles; bid = q;
  if (cls_fast) { q = fastdiv(bid, dv[1]); t.cls = bid - q * nclass; t.mt = q; }
  else { q = fastdiv(bid, dv[2]); t.mt = bid - q * mtiles; t.cls = q; }
  t.b = fastdiv(t.mt, dv[3]);
  const int trem = t.mt - t.b * tps;
  const int tyi = fastdiv(trem, dv[4]);
  t.ty0 = tyi * 8;
  t.tx0 = (trem - tyi * tpr) * 32;
  return t;
}

template <int N> __device__ __forceinline__ void hwait() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

template <int V> using IC = std::integral_constant<int, V>;

// The chunk walk.  A chunk = 64 input channels of one GROUP; 3x3 and transposed layers have one group.  A 4x4 stride-2
// convolution (S2) is four 2x2-tap stride-1 convolutions over the four input parities (r, s) -- input pixel (2y + r, 2x + s)
// of the space-to-depth grid (y, x) -- all accumulating into the same output tile: four groups, each with the patch
// geometry of the 3x3 kernel (origin (ty0 - 1, tx0 - 1), 10 x 34 pixels).  Tap (ty, tx) of group (r, s) reads the grid at
// (y + ty - r, x + tx - s) and the weights of kernel position (2 ty + 1 - r, 2 tx + 1 - s): the tables hold the r = s = 0
// values, the group shifts them by a constant.
struct HaloChunk {
  int a;   // byte offset inside a weight row: group shift + chunk * 128
  int t;   // patch-pixel shift of the group's taps
  int p;   // scalar byte offset of the chunk's patch source: parity pixel + chunk * 128
};
template <bool S2>
__device__ __forceinline__ HaloChunk halo_chunk(int g, int c, int cin2, int wi, int ldx2) {
  if constexpr (S2) {
    const int r = g >> 1, s = g & 1;
    return {c * 128 - (r * 4 + s) * cin2, -(r * H_PW + s), (r * wi + s) * ldx2 + c * 128};
  } else {
    return {c * 128, 0, c * 128};
  }
}

// OP ("one phase", round 3): ONE barrier pair and one cluster of 64 MFMAs per K tile instead of two of 32.  The fragment
// registers stay 64: the cluster starts with the A0 half tile's fragments and re-fills each A register quad with the A1
// half tile's fragment as soon as its eight MFMAs have issued (LDS reads inside the cluster, ~a K tile ahead of their
// use).  That keeps the A1 slot of a stage under read until the END of a group's cluster, so the LDS-DMA duty is split
// by group: waves 0-3 load the A0 half tile of K tile k + 1 (whose old content both groups finished in their pre-cluster
// parts), waves 4-7 -- one barrier behind, i.e. in their pre-cluster part while waves 0-3 run cluster k -- load the A1
// half tile (whose old content waves 0-3 finished with cluster k - 1 and they themselves with theirs).  Every wave waits
// for its own loads (vmcnt(0)) AFTER its cluster, before the closing barrier: the loads have a pre-cluster part and a whole
// cluster (~1 500 cycles) to land.  Barrier intervals double (1 024 MFMA cycles against the ~190 cycles of hand-over that
// kept the two-phase loop at 0.70-0.73 MFMA-busy).
// CFD_DIAG (build-time, scripts/diag_variants.sh; never set in the shipped library): what bounds the K loop, found by
// taking pieces of it away -- results are garbage, only the time means anything.  1: no LDS-DMA loads inside the K loop;
// 2: no fragment reads after the first K tile; 4: no MFMAs; 8: no barriers inside the K loop.
#ifndef CFD_DIAG
#define CFD_DIAG 0
#endif
template <int NT, bool S2 = false, bool OP = false>
__global__ __launch_bounds__(512, 2) void igemm256h_kernel(IgemmHaloParams P) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  using G = HaloGeo<NT, S2>;
  const IgemmParams &p = P.g;
  typedef bf16x8 frag_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const float act_ns = act_neg_scale(p.act, p.slope);
  const bool act_any = p.act == CFD_ACT_LEAKY || p.act == CFD_ACT_RELU;   // (uniform)
  const bool act_sig = p.act == CFD_ACT_SIGMOID;

  const HaloTile tile = halo_tile(P);
  const int cls = tile.cls, ph = cls >> 1, pw = cls & 1;
  const int n0 = tile.nt * 256;
  const int b = tile.b, ty0 = tile.ty0, tx0 = tile.tx0;
  const int cpt = p.cpt;
  const int ncc = S2 ? 4 * cpt : cpt;                       // chunks of the whole K range
  const int cin2 = p.Cin * 2, ldx2 = p.ldx * 2;
  int ng = 0, nc = 0;                                       // group / chunk-in-group of `nxt`
  auto chunk_step = [&]() {                                 // (clamped at the last chunk: the stream's overrun re-reads it)
    if (nc + 1 < cpt) ++nc;
    else if (S2 && ng < 3) { nc = 0; ++ng; }
  };
  HaloChunk cur = halo_chunk<S2>(0, 0, cin2, p.Wi, ldx2);
  chunk_step();
  HaloChunk nxt = halo_chunk<S2>(ng, nc, cin2, p.Wi, ldx2);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)P.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)P.wbytes, 0x00020000);

  int toff[NT], wto[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    toff[t] = __builtin_amdgcn_readfirstlane(P.toff[cls][t]);
    wto[t] = __builtin_amdgcn_readfirstlane(P.wtoff[cls][t]);
  }

  // ---- staging geometry: a wave-level load = 1 KiB = 8 rows x 128 B; lane -> row lane/8, physical chunk lane%8, which
  // holds the LOGICAL chunk (lane%8) ^ (row & 7) (swizzle on the source address; the LDS image stays lane-linear)
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  unsigned avoff[4];   // weights: row (n0 + half*128 + piece*64 + wave*8 + lrow), r = half*2 + piece
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if constexpr (OP)   // group g = wave >> 2 loads half tile g: pieces (wave & 3) * 4 + r, rows piece * 8 + lrow
      avoff[r] = (unsigned)(((n0 + wr * 128 + ((wave & 3) * 4 + r) * 8 + lrow) * p.Kw + chunk * 8) * 2);
    else
      avoff[r] = (unsigned)(((n0 + (r >> 1) * 128 + (r & 1) * 64 + wave * 8 + lrow) * p.Kw + chunk * 8) * 2);
  }
  unsigned pvoff[G::PPW];   // patch: load j of this wave covers patch pixels q*8 .. q*8+7, q = min(j*8 + wave, NPI-1)
  {
    const int py0 = ty0 + P.dymin[cls], px0 = tx0 + P.dxmin[cls];
#pragma unroll
    for (int j = 0; j < G::PPW; ++j) {
      const int q = min(j * 8 + wave, G::NPI - 1);
      const int pp = q * 8 + lrow;
      const int py = pp / H_PW, px = pp - py * H_PW;
      const int y = py0 + py, x = px0 + px;
      constexpr int S = S2 ? 2 : 1;   // (y, x) is on the class grid; S2: the space-to-depth grid of the input
      const bool ok = pp < G::PP && (unsigned)y < (unsigned)p.Hq && (unsigned)x < (unsigned)p.Wq;
      pvoff[j] = ok ? (unsigned)((((b * p.Hi + S * y) * p.Wi + S * x) * p.ldx + chunk * 8) * 2) : OOB;
    }
  }

  if constexpr ((CFD_DIAG & 16) != 0) {   // every load out of range: zeros reach LDS, nothing is fetched
#pragma unroll
    for (int r = 0; r < 4; ++r) avoff[r] = OOB;
#pragma unroll
    for (int j = 0; j < G::PPW; ++j) pvoff[j] = OOB;
  }
  if constexpr ((CFD_DIAG & 32) != 0) {   // every weight load re-reads K tile 0 (L1 / L2 hits), the patch loads chunk 0
#pragma unroll
    for (int t = 0; t < NT; ++t) wto[t] = 0;
  }
  char *const ldsw = lds + wave * 1024;
  auto issueA0 = [&](int st, unsigned sa) {
    char *dst = ldsw + st * (2 * HT_BYTES);
    bload16(wrs, avoff[0], sa, dst); bload16(wrs, avoff[1], sa, dst + 8192);
  };
  auto issueA1 = [&](int st, unsigned sa) {
    char *dst = ldsw + st * (2 * HT_BYTES) + HT_BYTES;
    bload16(wrs, avoff[2], sa, dst); bload16(wrs, avoff[3], sa, dst + 8192);
  };
  auto issueP = [&](int j, int pst, unsigned sp) {   // j: compile-time at every call site
    const int q = min(j * 8 + wave, G::NPI - 1);
    bload16(xrs, pvoff[j], sp, lds + H_ABYTES + pst * G::PATCH + q * 1024);
  };
  auto issueHalf = [&](int st, unsigned sa) {        // OP: this wave's four pieces of its group's half tile
    char *dst = lds + st * (2 * HT_BYTES) + wr * HT_BYTES + (wave & 3) * 4096;
#pragma unroll
    for (int r = 0; r < 4; ++r) bload16(wrs, avoff[r], sa, dst + r * 1024);
  };

  // ---- fragment read offsets
  const int fr = lane & 15, fq = lane >> 4;
  const int laneoff = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const char *ard[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) ard[s] = lds + ((wr * 64 * 128 + laneoff) ^ (s << 6));
  const int L0 = wc * H_PW + fr;   // patch pixel of (tile row wc, column fr) for a tap at the patch origin

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: patch of chunk 0, A0(0), A1(0), A0(1)
  // the bias of this n tile goes to LDS with the very first load (every wave issues the same 1 KiB, so the vmcnt
  // bookkeeping stays uniform; it is the oldest load and retires with the first wait): fetched from global memory in
  // the epilogue it cost a full round trip per tile (~1 us: P d5 447 -> 463 us, DeltaP g3 203 -> 221 with a bias)
  if (p.bias) {
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, p.Cout * 4, 0x00020000);
    bload16(brs, (unsigned)(n0 * 4 + lane * 16), 0u, lds + G::BIAS_OFF);
  }
#pragma unroll
  for (int j = 0; j < G::PPW; ++j) issueP(j, 0, (unsigned)cur.p);
  if constexpr (OP) {
    issueHalf(0, (unsigned)(wto[0] + cur.a));
    hwait<0>();
  } else {
    issueA0(0, (unsigned)(wto[0] + cur.a));
    issueA1(0, (unsigned)(wto[0] + cur.a));
    issueA0(1, (unsigned)(wto[1 % NT] + (1 / NT ? nxt.a : cur.a)));
    hwait<4>();
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // second group runs one barrier behind

  frag_t a[4][2], b0[2][2], b1[2][2];

#define CFD_MFMA(AI, BJ, BREG)                                                         \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                \
            acc[AI + i][BJ + j] = mma(a[i][s], BREG[j][s], acc[AI + i][BJ + j]);
#define CFD_KBARRIER() do { if constexpr (!(CFD_DIAG & 8)) __builtin_amdgcn_s_barrier(); } while (0)
#define CFD_SYNC_MFMA(VM, MFMAS)                                                       \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
      hwait<(VM) + ((CFD_DIAG & 64) ? 2 : 0) + ((CFD_DIAG & 128) ? 4 : 0)>();            \
      CFD_KBARRIER();                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_setprio(1);                                                   \
      if constexpr (!(CFD_DIAG & 4)) { MFMAS }                                         \
      else {                                                                           \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                \
          _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(a[i][s]));   \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) { asm volatile("" :: "v"(b0[j][s])); asm volatile("" :: "v"(b1[j][s])); } \
        }                                                                              \
      }                                                                                \
      __builtin_amdgcn_s_setprio(0);                                                   \
      __builtin_amdgcn_sched_barrier(0);                                               \
      CFD_KBARRIER();

  auto ktile = [&](auto tc, int c) {
    constexpr int t = decltype(tc)::value;
    constexpr int NP = G::count(t), J0 = G::first(t);
    constexpr int t1 = (t + 1) % NT, k1 = (t + 1) / NT, t2 = (t + 2) % NT, k2 = (t + 2) / NT;
    const int st = (NT & 1) ? ((c + t) & 1) : (t & 1);
    const int pst = c & 1;
    const unsigned sa1 = (unsigned)(wto[t1] + (k1 ? nxt.a : cur.a));
    const unsigned sa2 = (unsigned)(wto[t2] + (k2 ? nxt.a : cur.a));
    const unsigned sp = (unsigned)nxt.p;
    const int stoff = st * (2 * HT_BYTES);
    if constexpr (OP) {
      const int pp = L0 + toff[t] + cur.t;
      const int bo = H_ABYTES + pst * G::PATCH + (pp << 7) + ((fq ^ (pp & 7)) << 4);
#pragma unroll
      for (int jj = 0; jj < NP; ++jj) issueP(J0 + jj, pst ^ 1, sp);
      issueHalf(st ^ 1, sa1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const char *bb = lds + (bo ^ (s << 6));
#pragma unroll
        for (int j = 0; j < 2; ++j) b0[j][s] = *(const frag_t *)(bb + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i][s] = *(const frag_t *)(ard[s] + stoff + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) b1[j][s] = *(const frag_t *)(bb + H_BROW + j * 2048);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = mma(a[i][s], b0[j][s], acc[i][j]);
            acc[i][2 + j] = mma(a[i][s], b1[j][s], acc[i][2 + j]);
          }
        // this A register quad is free: the A1 half tile's fragment takes its place, used 24+ MFMAs from now
#pragma unroll
        for (int s = 0; s < 2; ++s) a[i][s] = *(const frag_t *)(ard[s] + stoff + HT_BYTES + i * 2048);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[4 + i][j] = mma(a[i][s], b0[j][s], acc[4 + i][j]);
            acc[4 + i][2 + j] = mma(a[i][s], b1[j][s], acc[4 + i][2 + j]);
          }
      // pin the interleave: (8 MFMAs, 2 LDS reads) x 4, then 32 MFMAs
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      hwait<0>();
      __builtin_amdgcn_s_barrier();
    } else {
    const bool rd = !(CFD_DIAG & 2) || (c == 0 && t == 0);
    {   // phase A: A0 x (B rows wc | 4 + wc)
      const int pp = L0 + toff[t] + cur.t;
      const int bo = H_ABYTES + pst * G::PATCH + (pp << 7) + ((fq ^ (pp & 7)) << 4);
      if (rd)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const char *bb = lds + (bo ^ (s << 6));
#pragma unroll
        for (int j = 0; j < 2; ++j) b0[j][s] = *(const frag_t *)(bb + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i][s] = *(const frag_t *)(ard[s] + stoff + i * 2048);
#pragma unroll
        for (int j = 0; j < 2; ++j) b1[j][s] = *(const frag_t *)(bb + H_BROW + j * 2048);
      }
      if constexpr (!(CFD_DIAG & 1)) {
#pragma unroll
        for (int jj = 0; jj < NP; ++jj) issueP(J0 + jj, pst ^ 1, sp);
        issueA1(st ^ 1, sa1);
      }
      CFD_SYNC_MFMA(4 + NP, CFD_MFMA(0, 0, b0) CFD_MFMA(0, 2, b1))
    }
    {   // phase B: A1 x (the same B fragments)
      if (rd)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i][s] = *(const frag_t *)(ard[s] + stoff + HT_BYTES + i * 2048);
      if constexpr (!(CFD_DIAG & 1)) issueA0(st, sa2);
      CFD_SYNC_MFMA(4 + NP, CFD_MFMA(4, 0, b0) CFD_MFMA(4, 2, b1))
    }
    }
  };

  for (int c = 0; c < ncc; ++c) {
    ktile(IC<0>{}, c); ktile(IC<1>{}, c); ktile(IC<2>{}, c); ktile(IC<3>{}, c);
    if constexpr (NT == 9) {
      ktile(IC<4>{}, c); ktile(IC<5>{}, c); ktile(IC<6>{}, c); ktile(IC<7>{}, c); ktile(IC<8>{}, c);
    }
    cur = nxt;
    chunk_step();
    nxt = halo_chunk<S2>(ng, nc, cin2, p.Wi, ldx2);
  }
#undef CFD_SYNC_MFMA
#undef CFD_KBARRIER
#undef CFD_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();
  hwait<0>();   // the tail loads must not outlive the workgroup's LDS allocation

  // tile-local row -> output pixel
  auto out_pixel = [&](int local) -> int64_t {
    const int qy = ty0 + (local >> 5), qx = tx0 + (local & 31);
    if (p.kind == CFD_CONVT_K4S2) return ((int64_t)b * p.Ho + 2 * qy + ph) * p.Wo + 2 * qx + pw;
    return ((int64_t)b * p.Hq + qy) * p.Wq + qx;
  };

  // ---- fused BatchNorm statistics (as in igemm256_kernel; the tile lies in sample b by construction)
  if (p.stat_sum) {
    __syncthreads();   // every wave's LDS-DMA has landed and every fragment read is done: LDS is free
    float *part = (float *)lds;   // [2][4 wc][256 n]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int nl = (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *(const f32x4 *)(lds + G::BIAS_OFF + nl * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[i][j][r] + bv[r];
          t1 += v;
          t2 = fmaf(v, v, t2);
        }
        t1 = row16_sum(t1);
        t2 = row16_sum(t2);
        if (fr == 0) {
          part[(0 * 4 + wc) * 256 + nl + r] = t1;
          part[(1 * 4 + wc) * 256 + nl + r] = t2;
        }
      }
    }
    __syncthreads();
    {
      const int which = tid >> 8, c = tid & 255;
      const float *pp = part + which * 1024 + c;
      const float tot = pp[0] + pp[256] + pp[512] + pp[768];
      atomicAdd((which ? p.stat_sq : p.stat_sum) + (size_t)b * p.Cout + n0 + c, tot);
    }
  }

  if (p.pool) {
    // ---- pooled epilogue (conv + bias + monotone activation + 2x2 max-pool, as igemm256_kernel's, on the 8 x 32 tile:
    // 4 x 16 pooled pixels, every pooling window inside the tile): fragments -> LDS [tile pixel][channel], then each
    // pooled pixel takes the maximum of its four rows and leaves as one 512-B run
    __syncthreads();
    char *stg = lds + OUT_OFF;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int nl = (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *(const f32x4 *)(lds + G::BIAS_OFF + nl * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int local = (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + fr;
        f32x4 v = acc[i][j] + bv;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act_linear(v[r], act_ns);
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *(bf16x4 *)(stg + local * OUT_ROW + nl * 2) = o;
      }
    }
    __syncthreads();
    const int half = lane >> 5, ck = lane & 31;
    auto mx = [](unsigned u0, unsigned u1, unsigned u2, unsigned u3) {
      // two bf16 per word: compare as floats (a bf16 is the high half of its f32), repack the winners' bits
      const float l = fmaxf(fmaxf(__uint_as_float(u0 << 16), __uint_as_float(u1 << 16)),
                            fmaxf(__uint_as_float(u2 << 16), __uint_as_float(u3 << 16)));
      const float g = fmaxf(fmaxf(__uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 & 0xffff0000u)),
                            fmaxf(__uint_as_float(u2 & 0xffff0000u), __uint_as_float(u3 & 0xffff0000u)));
      return (__float_as_uint(g) & 0xffff0000u) | (__float_as_uint(l) >> 16);
    };
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = wave * 8 + it * 2 + half;   // pooled pixel inside the tile: row q / 16, column q % 16
      const int py = q >> 4, px = q & 15;
      const char *s0 = stg + (py * 64 + px * 2) * OUT_ROW + ck * 16;
      const uint4 a4 = *(const uint4 *)s0, b4 = *(const uint4 *)(s0 + OUT_ROW);
      const uint4 c4 = *(const uint4 *)(s0 + 32 * OUT_ROW), d4 = *(const uint4 *)(s0 + 33 * OUT_ROW);
      uint4 o;
      o.x = mx(a4.x, b4.x, c4.x, d4.x); o.y = mx(a4.y, b4.y, c4.y, d4.y);
      o.z = mx(a4.z, b4.z, c4.z, d4.z); o.w = mx(a4.w, b4.w, c4.w, d4.w);
      const int64_t opix = ((int64_t)b * (p.Hq >> 1) + (ty0 >> 1) + py) * (p.Wq >> 1) + (tx0 >> 1) + px;
      if (p.pool == 2) *(uint4 *)(p.y2 + (opix * p.ldy2 + n0) * 2 + ck * 16) = o;
      else *(uint4 *)(p.y + (opix * p.ldy + n0) * 2 + ck * 16) = o;
    }
    if (p.pool == 2) {   // training: the unpooled activation too (the pool's backward needs it), from the same staged tile
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int ml = wave * 32 + it * 2 + half;
        const int64_t opix = out_pixel(ml);
        const uint4 v = *(const uint4 *)(stg + ml * OUT_ROW + ck * 16);
        *(uint4 *)(p.y + (opix * p.ldy + n0) * 2 + ck * 16) = v;
      }
    }
  } else if (P.stage_out) {
    // ---- staged epilogue: fragments (bias + activation applied, bf16) -> LDS [pixel][channel] -> whole pixel rows
    __syncthreads();
    char *stg = lds + OUT_OFF;
    auto to_lds = [&](auto with_act) {
      char *wbase = stg + (wc * 32 + fr) * OUT_ROW + (wr * 64 + fq * 4) * 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int nl = (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4 *)(lds + G::BIAS_OFF + nl * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 v = acc[i][j] + bv;
          if constexpr (decltype(with_act)::value) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act_linear(v[r], act_ns);
          }
          bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4 *)(wbase + ((j >> 1) * 128 + (j & 1) * 16) * OUT_ROW + ((i >> 2) * 128 + (i & 3) * 16) * 2) = o;
        }
      }
    };
    if (act_any) to_lds(std::true_type{}); else to_lds(std::false_type{});
    __syncthreads();
    const int half = lane >> 5, ck = lane & 31;   // two pixel rows per wave instruction, 32 chunks of 16 B each
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int ml = wave * 32 + it * 2 + half;
      const int64_t opix = out_pixel(ml);
      const uint4 v = *(const uint4 *)(stg + ml * OUT_ROW + ck * 16);
      *(uint4 *)(p.y + (opix * p.ldy + n0) * 2 + ck * 16) = v;
    }
  } else {
    auto direct_epi = [&](auto mode_c) {
      constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t opix = out_pixel((j >> 1) * 128 + wc * 32 + (j & 1) * 16 + fr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int n = n0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
          f32x4 v = acc[i][j];
          if (p.bias) v += *(const f32x4 *)(lds + G::BIAS_OFF + (n - n0) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act_linear(v[r], act_ns);   // ns = 1: identity
          if (act_sig) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 1.f / (1.f + __expf(-v[r]));
          }
          if constexpr (MODE == 1) {
            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *(bf16x4 *)(p.y + (opix * p.ldy + n) * 2) = o;
          } else {
            *(f32x4 *)(p.y + (opix * p.ldy + n) * 4) = v;
          }
        }
      }
    };
    if (p.out_bf16) direct_epi(IC<1>{}); else direct_epi(IC<2>{});
  }
}

// ---------------------------------------------------------------------------------------------------------
// 128(n) x 256(m) variant (Cout = 128 mod 256, and the small-M layers the host gives the 128-wide n tile): the machinery of
// igemm128n_kernel -- ONE phase of 32 MFMAs per K tile, a three-stage ring for the weight half tile (48 KiB) -- with the
// patch in place of the two gathered B half tiles.  Per K tile a wave issues 2 weight loads + at most 2 patch loads
// instead of 6 loads, which is what lets the memory cluster of one wave group fit under the MFMA cluster of the other.
//   wave (wr, wc) owns n rows wr*64 + [0,64) and tile rows 2 wc, 2 wc + 1 (four 16-pixel runs); the two rows' swizzle
//   keys differ (row pitch 34 = 2 mod 8), so there is one lane base address per row and tap.
//   phase k: read stage k % 3 and the patch; issue [patch loads n_t] + A(k+2) into stage (k+2) % 3; vmcnt(2 + n_t)
//   (A(k+1) landed; in the chunk's last K tile n_t = 0, so the whole next patch has landed too); barrier; 32 MFMAs; barrier.
// ---------------------------------------------------------------------------------------------------------
constexpr int H3_ABYTES = 3 * HT_BYTES;

template <int NT, bool S2 = false> struct HaloGeo3 : HaloGeo<NT, S2> {
  static constexpr int BIAS_OFF = H3_ABYTES + 2 * HaloGeo<NT, S2>::PATCH;
  static constexpr int LDS = BIAS_OFF + 1024;
};

template <int N> __device__ __forceinline__ void hwait3() {
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

template <int NT, bool S2 = false>
__global__ __launch_bounds__(512, 2) void igemm128nh_kernel(IgemmHaloParams P) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  using G = HaloGeo3<NT, S2>;
  const IgemmParams &p = P.g;
  typedef bf16x8 frag_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const float act_ns = act_neg_scale(p.act, p.slope);
  const bool act_sig = p.act == CFD_ACT_SIGMOID;

  const HaloTile tile = halo_tile(P);
  const int cls = tile.cls, ph = cls >> 1, pw = cls & 1;
  const int n0 = tile.nt * 128;
  const int b = tile.b, ty0 = tile.ty0, tx0 = tile.tx0;
  const int cpt = p.cpt;
  const int ncc = S2 ? 4 * cpt : cpt;                       // chunks of the whole K range
  const int cin2 = p.Cin * 2, ldx2 = p.ldx * 2;
  int ng = 0, nc = 0;                                       // group / chunk-in-group of `nxt`
  auto chunk_step = [&]() {                                 // (clamped at the last chunk: the stream's overrun re-reads it)
    if (nc + 1 < cpt) ++nc;
    else if (S2 && ng < 3) { nc = 0; ++ng; }
  };
  HaloChunk cur = halo_chunk<S2>(0, 0, cin2, p.Wi, ldx2);
  chunk_step();
  HaloChunk nxt = halo_chunk<S2>(ng, nc, cin2, p.Wi, ldx2);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)P.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)P.wbytes, 0x00020000);

  int toff[NT], wto[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    toff[t] = __builtin_amdgcn_readfirstlane(P.toff[cls][t]);
    wto[t] = __builtin_amdgcn_readfirstlane(P.wtoff[cls][t]);
  }

  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  unsigned avoff[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) avoff[r] = (unsigned)(((n0 + r * 64 + wave * 8 + lrow) * p.Kw + chunk * 8) * 2);
  unsigned pvoff[G::PPW];
  {
    const int py0 = ty0 + P.dymin[cls], px0 = tx0 + P.dxmin[cls];
#pragma unroll
    for (int j = 0; j < G::PPW; ++j) {
      const int q = min(j * 8 + wave, G::NPI - 1);
      const int pp = q * 8 + lrow;
      const int py = pp / H_PW, px = pp - py * H_PW;
      const int y = py0 + py, x = px0 + px;
      constexpr int S = S2 ? 2 : 1;   // (y, x) is on the class grid; S2: the space-to-depth grid of the input
      const bool ok = pp < G::PP && (unsigned)y < (unsigned)p.Hq && (unsigned)x < (unsigned)p.Wq;
      pvoff[j] = ok ? (unsigned)((((b * p.Hi + S * y) * p.Wi + S * x) * p.ldx + chunk * 8) * 2) : OOB;
    }
  }

  char *const ldsw = lds + wave * 1024;
  auto issueA = [&](int st, unsigned sa) {
    char *dst = ldsw + st * HT_BYTES;
    bload16(wrs, avoff[0], sa, dst); bload16(wrs, avoff[1], sa, dst + 8192);
  };
  auto issueP = [&](int j, int pst, unsigned sp) {
    const int q = min(j * 8 + wave, G::NPI - 1);
    bload16(xrs, pvoff[j], sp, lds + H3_ABYTES + pst * G::PATCH + q * 1024);
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int laneoff = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const char *ard[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) ard[s] = lds + ((wr * 64 * 128 + laneoff) ^ (s << 6));
  const int L0 = 2 * wc * H_PW + fr;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the bias of this n tile goes to LDS with the very first load (every wave issues the same 1 KiB, so the vmcnt
  // bookkeeping stays uniform; it is the oldest load and retires with the first wait): fetched from global memory in
  // the epilogue it cost a full round trip per tile (~1 us: P d5 447 -> 463 us, DeltaP g3 203 -> 221 with a bias)
  if (p.bias) {
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, p.Cout * 4, 0x00020000);
    bload16(brs, (unsigned)(n0 * 4 + lane * 16), 0u, lds + G::BIAS_OFF);
  }
#pragma unroll
  for (int j = 0; j < G::PPW; ++j) issueP(j, 0, (unsigned)cur.p);
  issueA(0, (unsigned)(wto[0] + cur.a));
  issueA(1, (unsigned)(wto[1 % NT] + (1 / NT ? nxt.a : cur.a)));
  hwait3<2>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();

  frag_t a[4][2], bf[4][2];
  int stg = 0;   // ring stage of the current K tile (NT = 4; with 9 taps it is t % 3)

  auto ktile = [&](auto tc, int c) {
    constexpr int t = decltype(tc)::value;
    constexpr int NP = G::count(t), J0 = G::first(t);
    constexpr int t2 = (t + 2) % NT, k2 = (t + 2) / NT;
    int st;
    if constexpr (NT % 3 == 0) st = t % 3; else st = stg;
    const int st2 = st == 0 ? 2 : st - 1;   // (st + 2) % 3
    const int pst = c & 1;
    const unsigned sa2 = (unsigned)(wto[t2] + (k2 ? nxt.a : cur.a));
    const unsigned sp = (unsigned)nxt.p;
    const int pp0 = L0 + toff[t] + cur.t, pp1 = pp0 + H_PW;
    const int pb = H3_ABYTES + pst * G::PATCH;
    const int bo0 = pb + (pp0 << 7) + ((fq ^ (pp0 & 7)) << 4);
    const int bo1 = pb + (pp1 << 7) + ((fq ^ (pp1 & 7)) << 4);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const char *b0p = lds + (bo0 ^ (s << 6)), *b1p = lds + (bo1 ^ (s << 6));
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i][s] = *(const frag_t *)(ard[s] + st * HT_BYTES + i * 2048);
      bf[0][s] = *(const frag_t *)(b0p);
      bf[1][s] = *(const frag_t *)(b0p + 2048);
      bf[2][s] = *(const frag_t *)(b1p);
      bf[3][s] = *(const frag_t *)(b1p + 2048);
    }
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) issueP(J0 + jj, pst ^ 1, sp);
    issueA(st2, sa2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    hwait3<2 + NP>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mma(a[i][s], bf[j][s], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if constexpr (NT % 3 != 0) stg = stg == 2 ? 0 : stg + 1;
  };

  for (int c = 0; c < ncc; ++c) {
    ktile(IC<0>{}, c); ktile(IC<1>{}, c); ktile(IC<2>{}, c); ktile(IC<3>{}, c);
    if constexpr (NT == 9) {
      ktile(IC<4>{}, c); ktile(IC<5>{}, c); ktile(IC<6>{}, c); ktile(IC<7>{}, c); ktile(IC<8>{}, c);
    }
    cur = nxt;
    chunk_step();
    nxt = halo_chunk<S2>(ng, nc, cin2, p.Wi, ldx2);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  hwait<0>();

  auto out_pixel = [&](int local) -> int64_t {
    const int qy = ty0 + (local >> 5), qx = tx0 + (local & 31);
    if (p.kind == CFD_CONVT_K4S2) return ((int64_t)b * p.Ho + 2 * qy + ph) * p.Wo + 2 * qx + pw;
    return ((int64_t)b * p.Hq + qy) * p.Wq + qx;
  };

  if (p.stat_sum) {
    __syncthreads();
    float *part = (float *)lds;   // [2][4 wc][128 n]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = wr * 64 + i * 16 + fq * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *(const f32x4 *)(lds + G::BIAS_OFF + nl * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[i][j][r] + bv[r];
          t1 += v;
          t2 = fmaf(v, v, t2);
        }
        t1 = row16_sum(t1);
        t2 = row16_sum(t2);
        if (fr == 0) {
          part[(0 * 4 + wc) * 128 + nl + r] = t1;
          part[(1 * 4 + wc) * 128 + nl + r] = t2;
        }
      }
    }
    __syncthreads();
    if (tid < 256) {
      const int which = tid >> 7, c = tid & 127;
      const float *pp = part + which * 512 + c;
      const float tot = pp[0] + pp[128] + pp[256] + pp[384];
      atomicAdd((which ? p.stat_sq : p.stat_sum) + (size_t)b * p.Cout + n0 + c, tot);
    }
  }

  auto direct_epi = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t opix = out_pixel(wc * 64 + j * 16 + fr);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + wr * 64 + i * 16 + fq * 4;
        f32x4 v = acc[i][j];
        if (p.bias) v += *(const f32x4 *)(lds + G::BIAS_OFF + (n - n0) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act_linear(v[r], act_ns);
        if (act_sig) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = 1.f / (1.f + __expf(-v[r]));
        }
        if constexpr (MODE == 1) {
          bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4 *)(p.y + (opix * p.ldy + n) * 2) = o;
        } else {
          *(f32x4 *)(p.y + (opix * p.ldy + n) * 4) = v;
        }
      }
    }
  };
  if (p.out_bf16) direct_epi(IC<1>{}); else direct_epi(IC<2>{});
}

// ---------------------------------------------------------------------------------------------------------
// 128(n) x 256(m) CO-RESIDENT variant (round 3): the same patch-staged tile with a K tile of ONE TAP x 32 CHANNELS, sized so
// that TWO workgroups share a CU (69 KiB of LDS, <= 128 registers per lane).
//
// Why: one 135-150 KiB workgroup per CU leaves nothing to run under a tile's set-up, cold first loads, epilogue and store
// drain (~7 K tiles' worth per tile, stamp build: 10 % of a 64-K-tile layer, 40 % of P d5's tiles, 64 % of DeltaP g3's), and
// inside the K loop each SIMD holds exactly two waves, so a barrier turn-around (~190 cycles against a 512-cycle MFMA
// cluster) is dead time on the matrix pipe.  With two workgroups per CU every SIMD holds four waves from two independent
// barrier domains: one workgroup's epilogue runs under the other's K loop, and a wave may spend three quarters of its
// time outside its MFMA cluster before the pipe idles.
//
// What had to shrink: the 64-channel patch double-buffered is 86 KiB.  Here a chunk is 32 channels (patch 22 KiB per stage)
// and a K tile is one tap of it: weights 128 rows x 64 B = 8 KiB per stage in a three-stage ring, ONE k-step of 16 MFMAs per
// wave and K tile (fragments: 4 + 4 registers quads instead of 8 + 8).  Rows are 64 B (one pixel / one weight row of 32
// channels); their four 16-B chunks are XOR-ed with bits 1-2 of the row index (chunk ^ ((row >> 1) & 3)), which makes
// ds_read_b128 conflict-free for 16 consecutive rows at ANY starting row (tap shifts are odd as often as even; searched by
// brute force over the instruction's lane groups) and is invariant under +16 rows, so fragments sit at immediate offsets.
// A wave-level LDS-DMA piece = 1 KiB = 16 rows; lane -> row lane >> 2, physical chunk lane & 3.
//   phase k: read stage k % 3 and the patch; issue [patch load n_t <= 1] + A(k+2) into stage (k+2) % 3; vmcnt(1 + n_t)
//   (A(k+1) landed; the chunk's last K tile carries no patch load, so the whole next patch has landed too); barrier;
//   16 MFMAs; barrier.  The two wave groups run one barrier apart as in the other kernels.
// ---------------------------------------------------------------------------------------------------------
constexpr int C_ST = 128 * 64;          // one weight stage
constexpr int C_ABYTES = 3 * C_ST;

template <int NT, bool S2 = false> struct HaloGeoC {
  static constexpr int PH = H_TH + ((NT == 9 || S2) ? 2 : 1);
  static constexpr int PP = PH * H_PW;             // patch pixels
  static constexpr int NPI = (PP + 15) / 16;       // wave-level load instructions (16 pixels x 64 B = 1 KiB each)
  static constexpr int PPW = (NPI + 7) / 8;        // ... per wave
  static constexpr int PER = (PPW + NT - 2) / (NT - 1);
  static constexpr int PATCH = NPI * 1024;
  static constexpr int BIAS_OFF = C_ABYTES + 2 * PATCH;   // the n tile's bias (512 B; the load writes 1 KiB)
  static constexpr int LDS = BIAS_OFF + 1024;
  static constexpr int first(int t) { return t * PER < PPW ? t * PER : PPW; }
  static constexpr int count(int t) { return first(t + 1) - first(t); }
  static_assert(first(NT - 1) == PPW, "the last K tile of a chunk must not carry patch loads");
  static_assert(PER <= 1, "vmcnt bookkeeping below assumes at most one patch load per K tile");
  static_assert(2 * LDS <= 160 * 1024, "two workgroups must fit one CU");
};

template <int N> __device__ __forceinline__ void hwaitc() {
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

template <bool S2>
__device__ __forceinline__ HaloChunk halo_chunk_c(int g, int c, int cin2, int wi, int ldx2) {   // 32-channel chunks: 64 B
  if constexpr (S2) {
    const int r = g >> 1, s = g & 1;
    return {c * 64 - (r * 4 + s) * cin2, -(r * H_PW + s), (r * wi + s) * ldx2 + c * 64};
  } else {
    return {c * 64, 0, c * 64};
  }
}

// NW = 128: waves 2 (n) x 4 (m), a wave owns 64 n rows and tile rows 2 wc, 2 wc + 1.  NW = 64 (Cout = 64 mod 128: DeltaP g3's
// data gradient): waves 1 x 8, a wave owns all 64 n rows and tile row wc; the weight stage keeps its 128-row frame, waves 4-7
// issue their weight load out of range (zero fill into rows no one reads) so that every wave's vmcnt count stays the same.
template <int NT, bool S2 = false, int NW = 128>
__global__ __launch_bounds__(512, 4) void igemm128c_kernel(IgemmHaloParams P) {
  constexpr int MJ = NW == 128 ? 4 : 2;        // 16-pixel fragments per wave
  constexpr int NWC = NW == 128 ? 4 : 8;       // waves along m
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  using G = HaloGeoC<NT, S2>;
  const IgemmParams &p = P.g;
  typedef bf16x8 frag_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                // ping-pong group: waves 4-7 run one barrier behind
  const int wr = NW == 128 ? wave >> 2 : 0, wc = NW == 128 ? wave & 3 : wave;

  const HaloTile tile = halo_tile(P);
  const int cls = tile.cls, ph = cls >> 1, pw = cls & 1;
  const int n0 = tile.nt * NW;
  const int b = tile.b, ty0 = tile.ty0, tx0 = tile.tx0;
  const int cpt = p.cpt;                                    // chunks of 32 channels per group (host: Cin / 32)
  const int ncc = S2 ? 4 * cpt : cpt;
  const int cin2 = p.Cin * 2, ldx2 = p.ldx * 2;
  int ng = 0, nc = 0;
  auto chunk_step = [&]() {                                 // (clamped at the last chunk: the stream's overrun re-reads it)
    if (nc + 1 < cpt) ++nc;
    else if (S2 && ng < 3) { nc = 0; ++ng; }
  };
  HaloChunk cur = halo_chunk_c<S2>(0, 0, cin2, p.Wi, ldx2);
  chunk_step();
  HaloChunk nxt = halo_chunk_c<S2>(ng, nc, cin2, p.Wi, ldx2);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)P.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)P.wbytes, 0x00020000);

  int toff[NT], wto[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    toff[t] = __builtin_amdgcn_readfirstlane(P.toff[cls][t]);
    wto[t] = __builtin_amdgcn_readfirstlane(P.wtoff[cls][t]);
  }

  // ---- staging geometry: piece = 16 rows x 64 B; lane -> row lane >> 2, physical chunk lane & 3 = logical chunk ^ ((row >> 1) & 3)
  const int lrow = lane >> 2;
  const int kc = (lane & 3) ^ ((lrow >> 1) & 3);
  const unsigned avoff = (NW == 128 || wave < 4) ? (unsigned)(((n0 + wave * 16 + lrow) * p.Kw + kc * 8) * 2) : OOB;
  unsigned pvoff[G::PPW];
  {
    const int py0 = ty0 + P.dymin[cls], px0 = tx0 + P.dxmin[cls];
#pragma unroll
    for (int j = 0; j < G::PPW; ++j) {
      const int q = min(j * 8 + wave, G::NPI - 1);
      const int pp = q * 16 + lrow;
      const int py = pp / H_PW, px = pp - py * H_PW;
      const int y = py0 + py, x = px0 + px;
      constexpr int S = S2 ? 2 : 1;
      const bool ok = pp < G::PP && (unsigned)y < (unsigned)p.Hq && (unsigned)x < (unsigned)p.Wq;
      pvoff[j] = ok ? (unsigned)((((b * p.Hi + S * y) * p.Wi + S * x) * p.ldx + kc * 8) * 2) : OOB;
    }
  }

  char *const ldsw = lds + wave * 1024;
  auto issueA = [&](int st, unsigned sa) { bload16(wrs, avoff, sa, ldsw + st * C_ST); };
  auto issueP = [&](int j, int pst, unsigned sp) {
    const int q = min(j * 8 + wave, G::NPI - 1);
    bload16(xrs, pvoff[j], sp, lds + C_ABYTES + pst * G::PATCH + q * 1024);
  };

  // ---- fragment read offsets (row r of a stage / pixel pp of the patch at r * 64, chunk fq at (fq ^ ((r >> 1) & 3)) * 16)
  const int fr = lane & 15, fq = lane >> 4;
  const char *const ard = lds + wr * 64 * 64 + fr * 64 + ((fq ^ ((fr >> 1) & 3)) << 4);
  const int L0 = (NW == 128 ? 2 * wc : wc) * H_PW + fr;

  f32x4 acc[4][MJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (p.bias) {   // the n tile's bias rides to LDS with the first loads (lanes 32-63 fetch the next tile's or zeros: unused)
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, p.Cout * 4, 0x00020000);
    bload16(brs, (unsigned)(n0 * 4 + lane * 16), 0u, lds + G::BIAS_OFF);
  }
#pragma unroll
  for (int j = 0; j < G::PPW; ++j) issueP(j, 0, (unsigned)cur.p);
  issueA(0, (unsigned)(wto[0] + cur.a));
  issueA(1, (unsigned)(wto[1 % NT] + (1 / NT ? nxt.a : cur.a)));
  hwaitc<1>();
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();

  frag_t a[4], bf[MJ];
  int stg = 0;   // ring stage of the current K tile

  auto ktile = [&](auto tc, int c) {
    constexpr int t = decltype(tc)::value;
    constexpr int NP = G::count(t), J0 = G::first(t);
    constexpr int t2 = (t + 2) % NT, k2 = (t + 2) / NT;
    int st;
    if constexpr (NT % 3 == 0) st = t % 3; else st = stg;
    const int st2 = st == 0 ? 2 : st - 1;   // (st + 2) % 3
    const int pst = c & 1;
    const unsigned sa2 = (unsigned)(wto[t2] + (k2 ? nxt.a : cur.a));
    const unsigned sp = (unsigned)nxt.p;
    // (opaque: without a group shift the tap's lane addresses are loop-invariant, and hoisting nine taps' worth of them
    //  out of the chunk loop spills -- scratch reloads count in vmcnt and would break the counted waits)
    int pp0 = L0 + toff[t] + cur.t;
    asm volatile("" : "+v"(pp0));
    const int pp1 = pp0 + H_PW;
    const int pb = C_ABYTES + pst * G::PATCH;
    const char *b0p = lds + pb + (pp0 << 6) + ((fq ^ ((pp0 >> 1) & 3)) << 4);
    const char *b1p = lds + pb + (pp1 << 6) + ((fq ^ ((pp1 >> 1) & 3)) << 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *(const frag_t *)(ard + st * C_ST + i * 1024);
    bf[0] = *(const frag_t *)(b0p);
    bf[1] = *(const frag_t *)(b0p + 1024);
    if constexpr (MJ == 4) {
      bf[2] = *(const frag_t *)(b1p);
      bf[3] = *(const frag_t *)(b1p + 1024);
    }
#pragma unroll
    for (int jj = 0; jj < NP; ++jj) issueP(J0 + jj, pst ^ 1, sp);
    issueA(st2, sa2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    hwaitc<1 + NP>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] = mma(a[i], bf[j], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if constexpr (NT % 3 != 0) stg = stg == 2 ? 0 : stg + 1;
  };

  for (int c = 0; c < ncc; ++c) {
    ktile(IC<0>{}, c); ktile(IC<1>{}, c); ktile(IC<2>{}, c); ktile(IC<3>{}, c);
    if constexpr (NT == 9) {
      ktile(IC<4>{}, c); ktile(IC<5>{}, c); ktile(IC<6>{}, c); ktile(IC<7>{}, c); ktile(IC<8>{}, c);
    }
    cur = nxt;
    chunk_step();
    nxt = halo_chunk_c<S2>(ng, nc, cin2, p.Wi, ldx2);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  hwait<0>();   // the tail loads must not outlive the workgroup's LDS allocation

  if (p.stat_sum) {   // fused BatchNorm statistics, as in igemm128nh_kernel
    __syncthreads();
    float *part = (float *)lds;   // [2][NWC wc][NW n]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = wr * 64 + i * 16 + fq * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *(const f32x4 *)(lds + G::BIAS_OFF + nl * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const float v = acc[i][j][r] + bv[r];
          t1 += v;
          t2 = fmaf(v, v, t2);
        }
        t1 = row16_sum(t1);
        t2 = row16_sum(t2);
        if (fr == 0) {
          part[(0 * NWC + wc) * NW + nl + r] = t1;
          part[(1 * NWC + wc) * NW + nl + r] = t2;
        }
      }
    }
    __syncthreads();
    if (tid < 2 * NW) {
      const int which = tid / NW, c = tid - which * NW;
      const float *pp = part + which * NWC * NW + c;
      float tot = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < NWC; ++k2) tot += pp[k2 * NW];
      atomicAdd((which ? p.stat_sq : p.stat_sum) + (size_t)b * p.Cout + n0 + c, tot);
    }
  }

  const float act_ns = act_neg_scale(p.act, p.slope);
  const bool act_sig = p.act == CFD_ACT_SIGMOID;
  auto direct_epi = [&](auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int local = wc * (16 * MJ) + j * 16 + fr;
      const int qy = ty0 + (local >> 5), qx = tx0 + (local & 31);
      const int64_t opix = p.kind == CFD_CONVT_K4S2 ? ((int64_t)b * p.Ho + 2 * qy + ph) * p.Wo + 2 * qx + pw
                                                     : ((int64_t)b * p.Hq + qy) * p.Wq + qx;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = n0 + wr * 64 + i * 16 + fq * 4;
        f32x4 v = acc[i][j];
        if (p.bias) v += *(const f32x4 *)(lds + G::BIAS_OFF + (n - n0) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act_linear(v[r], act_ns);
        if (act_sig) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = 1.f / (1.f + __expf(-v[r]));
        }
        if constexpr (MODE == 1) {
          bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4 *)(p.y + (opix * p.ldy + n) * 2) = o;
        } else {
          *(f32x4 *)(p.y + (opix * p.ldy + n) * 4) = v;
        }
      }
    }
  };
  if (p.out_bf16) direct_epi(IC<1>{}); else direct_epi(IC<2>{});
}

// which layers take this path: bf16, a stride-1 tap grid (3x3 'same' conv, one sub-pixel class of the transposed 4x4
// stride-2 conv, or one input parity of the 4x4 stride-2 conv), whole n tiles, no split-K, a class grid that 8 x 32 tiles
// cover exactly
bool igemm256h_eligible(const IgemmParams &p, int dtype, int ntile) {
  if (tuning(CFD_TUNE_PATCH) == 0) return false;
  const bool s1 = (p.kind == CFD_CONV_K3S1 || p.kind == CFD_CONVT_K4S2) && p.Hi == p.Hq && p.Wi == p.Wq;
  const bool s2 = p.kind == CFD_CONV_K4S2 && p.Hi == 2 * p.Hq && p.Wi == 2 * p.Wq && !p.pool;   // four parity groups
  return dtype == CFD_BF16 && (s1 || s2) && p.Cout % ntile == 0 && p.Cin % 64 == 0 && p.split_k == 1 &&
         (!p.pool || (ntile == 256 && p.nclass == 1 && p.out_bf16)) && p.Hq % H_TH == 0 && p.Wq % H_TW == 0;
}

int igemm256h_launch(const IgemmParams &pin, int ntile, unsigned xbytes, unsigned wbytes, int cls_fast, int stage_out,
                     hipStream_t st, bool coresident) {
  IgemmHaloParams P;
  memset(&P, 0, sizeof(P));
  P.g = pin;
  if (coresident) {
    CFD_REQUIRE((ntile == 128 || ntile == 64) && !pin.pool && pin.Cin % 32 == 0,
                "co-resident patch kernel: 128- or 64-wide n tiles, no pooled epilogue");
    P.g.cpt = pin.Cin / 32;      // this kernel's chunks are 32 channels
  }
  const IgemmParams &p = P.g;
  CFD_REQUIRE((ntile == 128 || ntile == 256 || (ntile == 64 && coresident)) && igemm256h_eligible(p, CFD_BF16, ntile) &&
                  p.ntiles * ntile == p.Cout,
              "patch-staged kernel: shape not eligible");
  P.xbytes = xbytes; P.wbytes = wbytes; P.cls_fast = cls_fast; P.stage_out = stage_out;
  P.tpr = p.Wq / H_TW;
  P.tps = (p.Hq / H_TH) * P.tpr;
  fastdiv_make((unsigned)p.ntiles, P.dv[0]); fastdiv_make((unsigned)p.nclass, P.dv[1]); fastdiv_make((unsigned)p.mtiles, P.dv[2]);
  fastdiv_make((unsigned)P.tps, P.dv[3]); fastdiv_make((unsigned)P.tpr, P.dv[4]);
  CFD_REQUIRE((int64_t)P.tps * p.B == p.mtiles && (int64_t)p.mtiles * p.ntiles * p.nclass == p.nblocks,
              "patch-staged kernel: tile count %d does not match the 8 x 32 tiling", p.mtiles);
  const bool s2 = p.kind == CFD_CONV_K4S2;
  const int ph_rows = H_TH + ((p.ntaps == 9 || s2) ? 2 : 1);
  CFD_REQUIRE(p.ntaps == 9 || p.ntaps == 4 || (s2 && p.ntaps == 16 && p.nclass == 1), "patch-staged kernel: %d taps", p.ntaps);
  if (s2) {
    // parity group (0, 0): tap (ty, tx) reads the space-to-depth grid at (y + ty, x + tx) = patch pixel (ty + 1, tx + 1) and
    // the weights of kernel position (2 ty + 1, 2 tx + 1); group (r, s) shifts both by a constant (halo_chunk)
    P.dymin[0] = -1; P.dxmin[0] = -1;
    for (int t = 0; t < 4; ++t) {
      const int ty = t >> 1, tx = t & 1;
      P.toff[0][t] = (ty + 1) * H_PW + tx + 1;
      P.wtoff[0][t] = ((2 * ty + 1) * 4 + 2 * tx + 1) * p.Cin * 2;
    }
  }
  for (int cls = 0; cls < (s2 ? 0 : p.nclass); ++cls) {
    int dy[9], dx[9], wt[9], dymin = 1 << 20, dxmin = 1 << 20;
    for (int t = 0; t < p.ntaps; ++t) {
      tap_geom(p.kind, cls >> 1, cls & 1, t, dy[t], dx[t], wt[t]);
      dymin = dy[t] < dymin ? dy[t] : dymin;
      dxmin = dx[t] < dxmin ? dx[t] : dxmin;
    }
    P.dymin[cls] = dymin; P.dxmin[cls] = dxmin;
    for (int t = 0; t < p.ntaps; ++t) {
      CFD_REQUIRE(dy[t] - dymin + H_TH <= ph_rows && dx[t] - dxmin + H_TW <= H_PW, "patch-staged kernel: tap outside the patch");
      P.toff[cls][t] = (dy[t] - dymin) * H_PW + (dx[t] - dxmin);
      P.wtoff[cls][t] = wt[t] * p.Cin * 2;
    }
  }
#define CFD_LAUNCH_H(KERNEL, GEO)                                                                                   \
  do {                                                                                                              \
    if (int e_ = set_max_lds((const void *)KERNEL, GEO::LDS)) return e_;                                            \
    hipLaunchKernelGGL(KERNEL, dim3(p.nblocks), dim3(512), GEO::LDS, st, P);                                        \
  } while (0)
  typedef HaloGeo<4, true> GeoS2;
  typedef HaloGeo3<4, true> Geo3S2;
  typedef HaloGeoC<4, true> GeoCS2;
  // cfd_set_tuning(CFD_TUNE_ONEPHASE, 0 | 1): the 256-wide kernel's one-cluster-per-K-tile schedule (A/B switch)
  const bool onephase = tuning(CFD_TUNE_ONEPHASE) == 1;
  if (coresident && ntile == 64) {
    if (s2) CFD_LAUNCH_H((igemm128c_kernel<4, true, 64>), GeoCS2);
    else if (p.ntaps == 9) CFD_LAUNCH_H((igemm128c_kernel<9, false, 64>), HaloGeoC<9>);
    else CFD_LAUNCH_H((igemm128c_kernel<4, false, 64>), HaloGeoC<4>);
  } else if (coresident) {
    if (s2) CFD_LAUNCH_H((igemm128c_kernel<4, true>), GeoCS2);
    else if (p.ntaps == 9) CFD_LAUNCH_H(igemm128c_kernel<9>, HaloGeoC<9>);
    else CFD_LAUNCH_H(igemm128c_kernel<4>, HaloGeoC<4>);
  } else if (ntile == 256 && onephase) {
    if (s2) CFD_LAUNCH_H((igemm256h_kernel<4, true, true>), GeoS2);
    else if (p.ntaps == 9) CFD_LAUNCH_H((igemm256h_kernel<9, false, true>), HaloGeo<9>);
    else CFD_LAUNCH_H((igemm256h_kernel<4, false, true>), HaloGeo<4>);
  } else if (ntile == 256) {
    if (s2) CFD_LAUNCH_H((igemm256h_kernel<4, true>), GeoS2);
    else if (p.ntaps == 9) CFD_LAUNCH_H(igemm256h_kernel<9>, HaloGeo<9>);
    else CFD_LAUNCH_H(igemm256h_kernel<4>, HaloGeo<4>);
  } else {
    if (s2) CFD_LAUNCH_H((igemm128nh_kernel<4, true>), Geo3S2);
    else if (p.ntaps == 9) CFD_LAUNCH_H(igemm128nh_kernel<9>, HaloGeo3<9>);
    else CFD_LAUNCH_H(igemm128nh_kernel<4>, HaloGeo3<4>);
  }
#undef CFD_LAUNCH_H
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

}  // namespace cfd
