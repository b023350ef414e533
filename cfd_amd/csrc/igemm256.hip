// bf16 / f32 implicit-GEMM convolution, 256(n) x 256(m) tile, 8 waves in two ping-pong groups.
//
// Same GEMM view, operand roles and NHWC epilogue as igemm.hip (weights = MFMA A side, gathered
// activations = B side); used for the large layers (Cout % 256 == 0), where the 128x128 kernel's
// one-barrier-per-K-tile loop tops out near 900 TFLOP/s.
//
// Structure (one workgroup per CU, 512 threads, 128 KiB LDS):
//   * a K tile is 64 bf16 (128 B per row) and is staged as four HALF tiles of 128 rows x 128 B:
//       slot 0 = A0 (n rows 0..127), 1 = B0 (m rows 0..127), 2 = B1 (m rows 128..255), 3 = A1 (n rows 128..255);
//     two stages of four slots = 128 KiB.  Half tiles form one stream H0, H1, ... (4 per K tile).
//   * wave (wr, wc) = (wave>>2, wave&3) owns n rows {wr*64 + [0,64)} of BOTH A halves and m rows
//     {wc*32 + [0,32)} of BOTH B halves: 8 x 4 fragments of 16x16, 128 accumulator registers.
//   * per K tile four phases; phase p = memory cluster (LDS fragment reads for quadrant p + LDS-DMA
//     issue of half tile H(q+PD)) -> barrier -> 16 MFMAs of quadrant p -> barrier.
//       p0: read A0,B0 -> A0xB0     p1: read B1 -> A0xB1     p2: read A1 -> A1xB1     p3: (no read) A1xB0
//   * waves 4..7 run one barrier behind waves 0..3, so on every SIMD one wave is in its MFMA cluster
//     while the other is in its memory cluster (the two waves of a SIMD are w and w+4).  A wave issues
//     at most one instruction per ~4 cycles, so the memory cluster must stay well under 64 instructions
//     to fit beside a 256-cycle MFMA cluster: addresses are therefore 32-bit buffer offsets
//     (buffer_load ... lds: per-lane voffset precomputed per tap, K position in the scalar soffset,
//     out-of-image taps = an out-of-range voffset, which the hardware zero-fills), the per-tap geometry
//     comes from a host-built table in the kernel arguments, and the K loop is unrolled by two so
//     stage addresses are immediates.
//   * LDS-DMA stays in flight across barriers: counted s_waitcnt vmcnt(2*(PD-2)) + raw s_barrier; a half
//     tile is read one phase after the wait that retires it and restaged >= 2 phases after its last read.
//     Past the end of the K range the stream re-issues the last K tile (into slots nobody reads again),
//     so the counts stay uniform.
#include "common.h"
#include "igemm.h"
#include "igemm256.h"
#include <type_traits>

namespace cfd {

// validity bits (bit t = tap t reads inside the image) of one class for the pixel whose stride-scaled origin is
// (y0, x0): 8 compares on the affine tap grid instead of one table load per tap (the address set-up of a tile is
// serial scalar-load latency otherwise: 16 taps x 4 rows x ~200 cycles)
__device__ __forceinline__ unsigned tap_mask(const int *g, int y0, int x0, int Hi, int Wi, bool mv) {
  const int kw = g[0], kh = g[1], dy0 = g[2], dys = g[3], dx0 = g[4], dxs = g[5];
  unsigned col = 0;
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
    if (kx < kw) col |= ((unsigned)(x0 + dx0 + dxs * kx) < (unsigned)Wi ? 1u : 0u) << kx;
  unsigned msk = 0;
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
    if (ky < kh) msk |= ((unsigned)(y0 + dy0 + dys * ky) < (unsigned)Hi ? col : 0u) << (ky * kw);
  return mv ? msk : 0u;
}

// walk over the taps of one class in table order (t = ky*kw + kx) by scalar increments; g = Igemm256Params::tgeo[cls]
struct TapWalk {
  int pix, wt;               // current tap: input-pixel offset dy*Wi + dx, weight tap index
  int kx, kw;                // column in the tap grid, columns per row
  int dpx, dpy, dwx, dwy;    // increments: next column / (next row, from the last column)
  __device__ __forceinline__ void init(const int *g, int Wi, int t) {
    kw = g[0];
    const int dy0 = g[2], dys = g[3], dx0 = g[4], dxs = g[5], wt0 = g[6], wxs = g[7] & 0xffff, wys = g[7] >> 16;
    const int ky = t / kw;
    kx = t - ky * kw;
    pix = (dy0 + dys * ky) * Wi + dx0 + dxs * kx;
    wt = wt0 + wys * ky + wxs * kx;
    dpx = dxs; dwx = wxs;
    dpy = dys * Wi - (kw - 1) * dxs;
    dwy = wys - (kw - 1) * wxs;
  }
  __device__ __forceinline__ void next() {
    if (++kx == kw) { kx = 0; pix += dpy; wt += dwy; }
    else { pix += dpx; wt += dwx; }
  }
};

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

// The kernel-argument fields a tile's set-up needs, fetched in ONE batch (the empty asm pins the values there): left to itself
// the compiler loads each field right before its first use, which made the set-up a chain of ~nine dependent scalar-load
// round trips; the block-index decode divides by host-made constants (fastdiv) instead of five run-time divisions.
#define CFD_KARGS_BATCH(P)                                                                                            \
  const int k_nblocks = (P).g.nblocks, k_ntiles = (P).g.ntiles, k_nclass = (P).g.nclass, k_mtiles = (P).g.mtiles,     \
            k_cls_fast = (P).cls_fast;                                                                                \
  unsigned kdv[3][3];                                                                                                 \
  _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_)                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) kdv[i_][j_] = (P).dv[i_][j_];                                    \
  {                                                                                                                   \
    const char *x_ = (P).g.x, *w_ = (P).g.w;                                                                          \
    const unsigned xb_ = (P).xbytes, wb_ = (P).wbytes;                                                                \
    const int a_ = (P).g.Hi, b_ = (P).g.Wi, c_ = (P).g.ldx, d_ = (P).g.Kw, e_ = (P).g.cpt, f_ = (P).g.ntaps,          \
              g_ = (P).g.split_k, h_ = (P).g.M, i2_ = (P).g.hw_shift, j2_ = (P).g.wq_shift, k_ = (P).g.kind,          \
              l_ = (P).g.pool, m_ = (P).g.Cin, n_ = (P).g.Hq, o_ = (P).g.Wq, p_ = (P).g.Ho, q_ = (P).g.Wo,            \
              r_ = (P).g.Cout;                                                                                        \
    const float *bias_ = (P).g.bias;                                                                                  \
    asm volatile("" ::"s"(x_), "s"(w_), "s"(xb_), "s"(wb_), "s"(a_), "s"(b_), "s"(c_), "s"(d_), "s"(e_), "s"(f_),     \
                 "s"(g_), "s"(h_), "s"(i2_), "s"(j2_), "s"(k_), "s"(l_), "s"(m_), "s"(n_), "s"(o_), "s"(p_), "s"(q_), \
                 "s"(r_), "s"(bias_));                                                                                         \
  }                                                                                                                   \
  asm volatile("" ::"s"(k_nblocks), "s"(k_ntiles), "s"(k_nclass), "s"(k_mtiles), "s"(k_cls_fast), "s"(kdv[0][0]),    \
               "s"(kdv[0][1]), "s"(kdv[0][2]), "s"(kdv[1][0]), "s"(kdv[1][1]), "s"(kdv[1][2]), "s"(kdv[2][0]),        \
               "s"(kdv[2][1]), "s"(kdv[2][2]));

// T = bf16: K tile = 64 elements, mfma_f32_16x16x32_bf16.  T = float: K tile = 32 elements, exact-f32
// mfma_f32_16x16x4f32; a 16-B fragment read then feeds four MFMAs (lane group fq supplies k = 16 s + 4 fq + e
// on BOTH operands -- any consistent k assignment sums to the same dot product).  Rows are 128 B either way.
template <typename T, int PD, int DIAG = 0, int NPH = 4>
__global__ __launch_bounds__(512, 2) void igemm256_kernel(Igemm256Params P) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const IgemmParams &p = P.g;
  constexpr int ES = sizeof(T);
  constexpr int EPC = 16 / ES;     // elements per 16-B chunk
  constexpr int BKE = 128 / ES;    // K elements per tile row
  typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type frag_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const float act_ns = act_neg_scale(p.act, p.slope);
  const bool act_any = p.act == CFD_ACT_LEAKY || p.act == CFD_ACT_RELU;   // (uniform)
  const bool act_sig = p.act == CFD_ACT_SIGMOID;   // (uniform; never set by the nets for these kernels)
  // DIAG == 2: four stamps per wave (entry, pipeline primed, K loop done, stores issued) + where it ran
  unsigned long long tg[6] = {0, 0, 0, 0, 0, 0};
#define CFD_TSTAMP(i)                                                                   \
  if constexpr (DIAG == 2) {                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tg[i]) :: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                                                  \
  }
  CFD_TSTAMP(0)

  // ---- XCD-aware block remap (bijective)
  CFD_KARGS_BATCH(P)
  int bid = blockIdx.x;
  {
    const int nwg = k_nblocks, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int nt, mt, cls;
  {
    int q = fastdiv(bid, kdv[0]);
    nt = bid - q * k_ntiles; bid = q;
    if (k_cls_fast) {
      q = fastdiv(bid, kdv[1]); cls = bid - q * k_nclass; bid = q;
      q = fastdiv(bid, kdv[2]); mt = bid - q * k_mtiles; bid = q;
    } else {
      q = fastdiv(bid, kdv[2]); mt = bid - q * k_mtiles; bid = q;
      q = fastdiv(bid, kdv[1]); cls = bid - q * k_nclass; bid = q;
    }
  }
  const int sk = bid;
  int tg8[8];   // this class's tap grid, one batch (tap_mask and the tap walk read it)
#pragma unroll
  for (int i = 0; i < 8; ++i) tg8[i] = P.tgeo[cls][i];
  asm volatile("" ::"s"(tg8[0]), "s"(tg8[1]), "s"(tg8[2]), "s"(tg8[3]), "s"(tg8[4]), "s"(tg8[5]), "s"(tg8[6]), "s"(tg8[7]));
  const int ph = cls >> 1, pw = cls & 1;
  const int m0 = mt * 256, n0 = nt * 256;
  const int S = (p.kind == CFD_CONV_K4S2) ? 2 : 1;

  const int ktiles = p.ntaps * p.cpt;
  const int kbeg = k_slice(ktiles, sk, p.split_k);
  const int kend = k_slice(ktiles, sk + 1, p.split_k);
  const int nkt = kend - kbeg;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)P.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)P.wbytes, 0x00020000);

  // ---- staging geometry: a half tile = 16 wave-instructions of 1 KiB (8 rows x 128 B); wave w issues
  // pieces w and 8 + w.  Lane -> row lane/8 of its piece, physical chunk lane%8, which holds the
  // LOGICAL chunk (lane%8) ^ (lane/8): the swizzle is applied on the source address (LDS image stays lane-linear).
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  // A side: byte offset of this lane's chunk in row (n0 + half*128 + piece*64 + wave*8 + lrow), r = half*2 + piece
  unsigned avoff[4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
    avoff[r] = (unsigned)(((n0 + (r >> 1) * 128 + (r & 1) * 64 + wave * 8 + lrow) * p.Kw + chunk * EPC) * ES);
  // B side: four pixel rows per lane, r = half*2 + piece
  unsigned rowoff[4], vmask[4], cur[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int local = (r >> 1) * 128 + (r & 1) * 64 + wave * 8 + lrow;
    const int m = p.pool ? pool_pixel(p, mt, local) : m0 + local;
    const bool mv = m < p.M;
    const int mm = mv ? m : 0;
    int b, qy, qx;
    split_pixel(p, mm, b, qy, qx);
    const int y0 = qy * S, x0 = qx * S;
    rowoff[r] = (unsigned)((((b * p.Hi + y0) * p.Wi + x0) * p.ldx + chunk * EPC) * ES);
    vmask[r] = tap_mask(tg8, y0, x0, p.Hi, p.Wi, mv);
  }

  // ---- issue stream state (uniform): the K tile whose half tiles are being issued
  int iss_left = nkt;                       // K tiles not yet fully issued (incl. the current one)
  int iss_t = kbeg ? kbeg / p.cpt : 0, iss_c = kbeg - iss_t * p.cpt;
  unsigned s_a = 0, s_b = 0;                // scalar byte offsets: A = (wtap*Cin + c*64)*2, B = c*128
  // A tap switch sits inside a memory cluster, i.e. on the critical path of its phase, and layers with few K tiles per
  // tap (Cin 64..256) switch every 1..4 K tiles.  Looking the tap up in the kernel-argument table cost a dependent
  // scalar load (a few hundred cycles, and its wait also drains the fragment reads sharing lgkmcnt) per switch; the
  // taps of every kind form an affine grid (checked by the host), so the walk is a handful of scalar adds instead:
  // tw_pix = (dy*Wi + dx) of the current tap, tw_wt = its weight tap index, (tw_kx) its column in the kh x kw grid.
  TapWalk tw;
  tw.init(tg8, p.Wi, iss_t);
  auto tap_switch = [&]() {
    s_a = (unsigned)((tw.wt * p.Cin + iss_c * BKE) * ES);
    s_b = (unsigned)(iss_c * 128);
    const int tapx = tw.pix * p.ldx * ES;
#pragma unroll
    for (int r = 0; r < 4; ++r) cur[r] = ((vmask[r] >> iss_t) & 1u) ? rowoff[r] + (unsigned)tapx : OOB;
  };
  auto iss_advance = [&]() {
    if (iss_left > 1) {
      --iss_left;
      s_a += 128; s_b += 128;
      if (++iss_c == p.cpt) { iss_c = 0; ++iss_t; tw.next(); tap_switch(); }
    }
  };
  tap_switch();

  char *const ldsw = lds + wave * 1024;
  auto issue = [&](int slot, int stage) {   // compile-time constants at every call site
    char *dst = ldsw + stage * ST_BYTES + slot * HT_BYTES;
    if (slot == 0) {
      bload16(wrs, avoff[0], s_a, dst); bload16(wrs, avoff[1], s_a, dst + 8192);
    } else if (slot == 3) {
      bload16(wrs, avoff[2], s_a, dst); bload16(wrs, avoff[3], s_a, dst + 8192);
    } else if (slot == 1) {
      bload16(xrs, cur[0], s_b, dst); bload16(xrs, cur[1], s_b, dst + 8192);
    } else {
      bload16(xrs, cur[2], s_b, dst); bload16(xrs, cur[3], s_b, dst + 8192);
    }
  };

  // ---- fragment read offsets (bytes inside a half tile); k-step 1 is the same address ^ 64
  const int fr = lane & 15, fq = lane >> 4;
  const int laneoff = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const char *ard[2], *brd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    ard[s] = lds + ((wr * 64 * 128 + laneoff) ^ (s << 6));
    brd[s] = lds + ((wc * 32 * 128 + laneoff) ^ (s << 6));
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  CFD_TSTAMP(5)
  // the n tile's bias goes to LDS with the very first load (every wave issues the same 1 KiB so the vmcnt bookkeeping
  // stays uniform; it is the oldest load and retires with the first wait): fetched from global memory in the epilogue it
  // cost a round trip per tile that nothing overlapped (~1 us: P d5 447 -> 463 us, DeltaP g3 203 -> 221 with a bias)
  if (p.bias) {
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, p.Cout * 4, 0x00020000);
    bload16(brs, (unsigned)(n0 * 4 + lane * 16), 0u, lds + BIAS256_OFF);
  }
  // ---- prologue: H0 .. H(PD-1)   (NPH == 2: H0 .. H5, and H0..H2 must have landed)
  constexpr int PRO = NPH == 2 ? 6 : PD;
#pragma unroll
  for (int h = 0; h < PRO; ++h) {
    issue(h & 3, (h >> 2) & 1);
    if ((h & 3) == 3) iss_advance();
  }
  if constexpr (NPH == 2) wait_vmcnt<6>(); else wait_vmcnt<2 * (PD - 2)>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // second group runs one barrier behind
  CFD_TSTAMP(1)

  frag_t a[4][2], b0[2][2], b1[2][2];

  // diagnostic build only (DIAG): cycle stamps around the four segments of a phase, summed per wave
  unsigned long long dg[5] = {0, 0, 0, 0, 0}, dt0 = 0, dt1 = 0;
#define CFD_STAMP(dst)                                                                 \
      if constexpr (DIAG == 1) {                                                       \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst) :: "memory");  \
        __builtin_amdgcn_sched_barrier(0);                                             \
      }
#define CFD_PHASE(ST, PH, READS, MFMAS)                                                \
    {                                                                                  \
      CFD_STAMP(dt0)                                                                   \
      READS                                                                            \
      issue((PH + PD) & 3, (ST + ((PH + PD) >> 2)) & 1);                               \
      if (((PH + PD) & 3) == 3) iss_advance();                                         \
      CFD_STAMP(dt1) if constexpr (DIAG == 1) { dg[0] += dt1 - dt0; }                       \
      wait_vmcnt<2 * (PD - 2)>();                                                      \
      CFD_STAMP(dt0) if constexpr (DIAG == 1) { dg[1] += dt0 - dt1; }                       \
      __builtin_amdgcn_s_barrier();                                                    \
      CFD_STAMP(dt1) if constexpr (DIAG == 1) { dg[2] += dt1 - dt0; }                       \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_setprio(1);                                                   \
      MFMAS                                                                            \
      __builtin_amdgcn_s_setprio(0);                                                   \
      __builtin_amdgcn_sched_barrier(0);                                               \
      CFD_STAMP(dt0) if constexpr (DIAG == 1) { dg[3] += dt0 - dt1; }                       \
      __builtin_amdgcn_s_barrier();                                                    \
      CFD_STAMP(dt1) if constexpr (DIAG == 1) { dg[4] += dt1 - dt0; }                       \
    }
#define CFD_MFMA(AI, BJ, BREG)                                                         \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                \
            acc[AI + i][BJ + j] = mma(a[i][s], BREG[j][s], acc[AI + i][BJ + j]);
#define CFD_KTILE(ST)                                                                  \
    CFD_PHASE(ST, 0,                                                                   \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
          b0[j][s] = *(const frag_t *)(brd[s] + ST * ST_BYTES + 1 * HT_BYTES + j * 2048); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          a[i][s] = *(const frag_t *)(ard[s] + ST * ST_BYTES + 0 * HT_BYTES + i * 2048); \
      },                                                                               \
      CFD_MFMA(0, 0, b0))                                                              \
    CFD_PHASE(ST, 1,                                                                   \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
          b1[j][s] = *(const frag_t *)(brd[s] + ST * ST_BYTES + 2 * HT_BYTES + j * 2048); \
      ,                                                                                \
      CFD_MFMA(0, 2, b1))                                                              \
    CFD_PHASE(ST, 2,                                                                   \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          a[i][s] = *(const frag_t *)(ard[s] + ST * ST_BYTES + 3 * HT_BYTES + i * 2048); \
      ,                                                                                \
      CFD_MFMA(4, 2, b1))                                                              \
    CFD_PHASE(ST, 3, , CFD_MFMA(4, 0, b0))

  // NPH == 2: two phases of 32 MFMAs per K tile (half as many barrier intervals as the 4 x 16 schedule: each
  // interval carries ~170 cycles of barrier / pipeline-restart overhead on top of its MFMA time).
  //   phase A: read A0, B0, B1; issue H(4k+6), H(4k+7) (B1, A1 of K tile k+1);  MFMA A0 x (B0 | B1)
  //   phase B: read A1;         issue H(4k+8), H(4k+9) (A0, B0 of K tile k+2);  MFMA A1 x (B0 | B1)
  // Fragment reads are retired (lgkmcnt(0)) BEFORE the phase's first barrier, so a slot may be restaged one phase
  // after its last read; vmcnt(8) / vmcnt(6) leave 4 / 3 half tiles in flight.
#define CFD_PHASE2(ST, SLOTA, STA, SLOTB, STB, VM, READS, MFMAS)                        \
    {                                                                                  \
      CFD_STAMP(dt0)                                                                   \
      READS                                                                            \
      issue(SLOTA, STA); issue(SLOTB, STB);                                            \
      if (SLOTB == 3) iss_advance();                                                   \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
      CFD_STAMP(dt1) if constexpr (DIAG == 1) { dg[0] += dt1 - dt0; }                       \
      wait_vmcnt<VM>();                                                                \
      CFD_STAMP(dt0) if constexpr (DIAG == 1) { dg[1] += dt0 - dt1; }                       \
      __builtin_amdgcn_s_barrier();                                                    \
      CFD_STAMP(dt1) if constexpr (DIAG == 1) { dg[2] += dt1 - dt0; }                       \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_setprio(1);                                                   \
      MFMAS                                                                            \
      __builtin_amdgcn_s_setprio(0);                                                   \
      __builtin_amdgcn_sched_barrier(0);                                               \
      CFD_STAMP(dt0) if constexpr (DIAG == 1) { dg[3] += dt0 - dt1; }                       \
      __builtin_amdgcn_s_barrier();                                                    \
      CFD_STAMP(dt1) if constexpr (DIAG == 1) { dg[4] += dt1 - dt0; }                       \
    }
#define CFD_KTILE2(ST)                                                                 \
    CFD_PHASE2(ST, 2, (ST + 1) & 1, 3, (ST + 1) & 1, 8,                                \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
          b0[j][s] = *(const frag_t *)(brd[s] + ST * ST_BYTES + 1 * HT_BYTES + j * 2048); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          a[i][s] = *(const frag_t *)(ard[s] + ST * ST_BYTES + 0 * HT_BYTES + i * 2048); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
          b1[j][s] = *(const frag_t *)(brd[s] + ST * ST_BYTES + 2 * HT_BYTES + j * 2048); \
      },                                                                               \
      CFD_MFMA(0, 0, b0) CFD_MFMA(0, 2, b1))                                           \
    CFD_PHASE2(ST, 0, ST, 1, ST, 6,                                                    \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          a[i][s] = *(const frag_t *)(ard[s] + ST * ST_BYTES + 3 * HT_BYTES + i * 2048); \
      ,                                                                                \
      CFD_MFMA(4, 0, b0) CFD_MFMA(4, 2, b1))

  if constexpr (NPH == 2) {
    for (int k = 0; k < nkt; k += 2) {
      CFD_KTILE2(0)
      if (k + 1 >= nkt) break;
      CFD_KTILE2(1)
    }
  } else {
    for (int k = 0; k < nkt; k += 2) {
      CFD_KTILE(0)
      if (k + 1 >= nkt) break;
      CFD_KTILE(1)
    }
  }
#undef CFD_KTILE2
#undef CFD_PHASE2
#undef CFD_KTILE
#undef CFD_MFMA
#undef CFD_PHASE
#undef CFD_STAMP
  if constexpr (DIAG == 1) {
    // stamps leave the kernel only through the workspace (split_k must be 1): [block][wave][8]
    if (lane == 0 && p.ws) {
      unsigned long long *o = (unsigned long long *)p.ws + ((size_t)blockIdx.x * 8 + wave) * 8;
      for (int i = 0; i < 5; ++i) o[i] = dg[i];
      o[5] = (unsigned long long)nkt;
    }
  }
  CFD_TSTAMP(2)
  if (wr == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();   // the tail loads must not outlive the workgroup's LDS allocation
  CFD_TSTAMP(4)

  // ---- fused BatchNorm statistics (host guarantees: split_k == 1, the 256 pixel rows of a tile lie in ONE
  // sample and none is ragged): per channel sum and sum of squares of (acc + bias) over the tile's pixels,
  // reduced over the 16 pixel lanes by DPP, over the 4 pixel waves through LDS, then one atomic per channel
  if (p.stat_sum) {
    __syncthreads();   // every wave's LDS-DMA has landed and every fragment read is done: LDS is free
    float *part = (float *)lds;   // [2][4 wc][256 n]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int nl = (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *(const f32x4 *)(lds + BIAS256_OFF + nl * 4);
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
      const int b = m0 / (p.Hq * p.Wq);
      atomicAdd((which ? p.stat_sq : p.stat_sum) + (size_t)b * p.Cout + n0 + c, tot);
    }
  }

  if (p.pool) {
    // ---- pooled epilogue (conv + bias + monotone activation + 2x2 max-pool, DeltaP/models/generator.py:27-34 at
    // inference): fragments -> LDS [tile row = 2 image rows x 128 columns][channels], then each pooled pixel takes the
    // maximum of its four rows and leaves as whole 512-B runs.  The unpooled activation never reaches HBM.  f32 goes in
    // two passes of 128 channels (a 256-channel f32 tile would need 256 KiB).
    __syncthreads();   // every wave's tail LDS-DMA has landed (each waited for its own): the stages are dead
    char *stg = lds + OUT_OFF;
    const int cb = p.Wq >> 7, rp = mt / cb, cblk = mt - rp * cb;
    const int half = lane >> 5, ck = lane & 31;   // two pooled pixels per wave instruction, 32 chunks of 16 B each
    constexpr int NPASS = ES == 2 ? 1 : 2;
#pragma unroll
    for (int h = 0; h < NPASS; ++h) {
      if (h) __syncthreads();                     // the previous half's readers are done
#pragma unroll
      for (int ii = 0; ii < 8 / NPASS; ++ii) {
        const int i = h * (8 / NPASS) + ii;
        const int nl = (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4 *)(lds + BIAS256_OFF + nl * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int local = (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + fr;
          f32x4 v = acc[i][j] + bv;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act_linear(v[r], act_ns);   // ns = 1: identity
          if constexpr (ES == 2) {
            bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
            *(bf16x4 *)(stg + local * OUT_ROW + nl * 2) = o;
          } else {
            *(f32x4 *)(stg + local * OUT_ROW + (nl - h * 128) * 4) = v;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = wave * 8 + it * 2 + half;   // pooled column inside the tile (0..63)
        const char *s0 = stg + (2 * q) * OUT_ROW + ck * 16;
        const uint4 a = *(const uint4 *)s0, b = *(const uint4 *)(s0 + OUT_ROW);
        const uint4 c = *(const uint4 *)(s0 + 128 * OUT_ROW), d = *(const uint4 *)(s0 + 129 * OUT_ROW);
        uint4 o;
        if constexpr (ES == 2) {
          auto mx = [](unsigned u0, unsigned u1, unsigned u2, unsigned u3) {
            // two bf16 per word: compare as floats (a bf16 is the high half of its f32), repack the winners' bits
            const float l = fmaxf(fmaxf(__uint_as_float(u0 << 16), __uint_as_float(u1 << 16)),
                                  fmaxf(__uint_as_float(u2 << 16), __uint_as_float(u3 << 16)));
            const float g = fmaxf(fmaxf(__uint_as_float(u0 & 0xffff0000u), __uint_as_float(u1 & 0xffff0000u)),
                                  fmaxf(__uint_as_float(u2 & 0xffff0000u), __uint_as_float(u3 & 0xffff0000u)));
            return (__float_as_uint(g) & 0xffff0000u) | (__float_as_uint(l) >> 16);
          };
          o.x = mx(a.x, b.x, c.x, d.x); o.y = mx(a.y, b.y, c.y, d.y);
          o.z = mx(a.z, b.z, c.z, d.z); o.w = mx(a.w, b.w, c.w, d.w);
        } else {
          auto mx = [](unsigned u0, unsigned u1, unsigned u2, unsigned u3) {
            return __float_as_uint(fmaxf(fmaxf(__uint_as_float(u0), __uint_as_float(u1)),
                                         fmaxf(__uint_as_float(u2), __uint_as_float(u3))));
          };
          o.x = mx(a.x, b.x, c.x, d.x); o.y = mx(a.y, b.y, c.y, d.y);
          o.z = mx(a.z, b.z, c.z, d.z); o.w = mx(a.w, b.w, c.w, d.w);
        }
        const int64_t opix = (int64_t)rp * (p.Wq >> 1) + cblk * 64 + q;
        *(uint4 *)(p.y + (opix * p.ldy + n0 + h * 128) * ES + ck * 16) = o;
      }
    }
  } else if (P.stage_out) {
    // ---- staged epilogue: fragments (bias + activation applied, bf16) -> LDS [pixel][channel] -> whole pixel rows
    __syncthreads();   // every wave's tail LDS-DMA has landed (each waited for its own): the stages are dead
    char *stg = lds + OUT_OFF;
    // two copies of the fragment loop, chosen once: most launches write the raw pre-BatchNorm sum (no activation),
    // and the host keeps sigmoid layers off this path
    auto to_lds = [&](auto with_act) {
      char *wbase = stg + (wc * 32 + fr) * OUT_ROW + (wr * 64 + fq * 4) * 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int nl = (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4 *)(lds + BIAS256_OFF + nl * 4);
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
      const int m = m0 + ml;
      if (m < p.M) {
        int64_t opix = m;
        if (p.kind == CFD_CONVT_K4S2) {
          int b, qy, qx;
          split_pixel(p, m, b, qy, qx);
          opix = ((int64_t)b * p.Ho + 2 * qy + ph) * p.Wo + 2 * qx + pw;
        }
        const uint4 v = *(const uint4 *)(stg + ml * OUT_ROW + ck * 16);
        *(uint4 *)(p.y + (opix * p.ldy + n0) * 2 + ck * 16) = v;
      }
    }
  } else {
  // ---- epilogue: lane holds channels n..n+3 (register index) of pixel m.  Three copies (split-K slab / bf16 / f32
  // output), chosen once per launch instead of per fragment.
  auto direct_epi = [&](auto mode_c) {
  constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + fr;
    if (m >= p.M) continue;
    int b, qy, qx;
    split_pixel(p, m, b, qy, qx);
    int64_t opix;
    if (p.kind == CFD_CONVT_K4S2)
      opix = ((int64_t)b * p.Ho + 2 * qy + ph) * p.Wo + 2 * qx + pw;
    else
      opix = m;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int n = n0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
      f32x4 v = acc[i][j];
      if constexpr (MODE == 0) {          // split-K slab: raw partial sums
        *(f32x4 *)(p.ws + ((int64_t)sk * p.B * p.Ho * p.Wo + opix) * p.Cout + n) = v;
      } else {
        if (p.bias) v += *(const f32x4 *)(lds + BIAS256_OFF + (n - n0) * 4);
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
  }
  };
  if (p.split_k > 1) direct_epi(std::integral_constant<int, 0>{});
  else if (p.out_bf16) direct_epi(std::integral_constant<int, 1>{});
  else direct_epi(std::integral_constant<int, 2>{});
  }
  if constexpr (DIAG == 2) {
    CFD_TSTAMP(3)
    if (lane == 0 && p.split_k == 1 && p.ws) {   // (the workspace is free when there are no split-K slabs)
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned long long *o = (unsigned long long *)p.ws + ((size_t)blockIdx.x * 8 + wave) * 8;
      for (int i = 0; i < 4; ++i) o[i] = tg[i];
      o[4] = hw; o[5] = (unsigned long long)xcc | (tg[5] << 8); o[6] = (unsigned long long)nkt; o[7] = tg[4];
    }
  }
#undef CFD_TSTAMP
}

// ---------------------------------------------------------------------------------------------------------
// Persistent form of the bf16 two-phase schedule: one workgroup per CU walks a list of tiles, and the LDS-DMA issue
// stream simply CONTINUES into the next tile while the MFMA side is still on the last K tiles of the current one.
// A tile boundary then costs the epilogue (accumulators -> global, while the next tile's first half tiles are
// already in flight) instead of workgroup launch + address set-up + a cold pipeline prologue + epilogue: measured
// on the one-tile-per-workgroup kernel that overhead is ~10 K tiles' worth per tile (14 % of a 64-K-tile layer).
//   * tiles of a workgroup: XCD x owns a contiguous range of logical tiles (same bijection as above); its G/8
//     workgroups take tiles j, j + G/8, ... of that range, so an XCD always works on a window of adjacent tiles.
//   * issue side and MFMA side each decode "their" tile; the per-lane address state belongs to the issue side only.
//   * every K range must be even (each tile then starts in stage 0); the host falls back otherwise.
//   * the epilogue's global stores share vmcnt with the DMA loads: it ends with vmcnt(0), by which time the
//     prefetched half tiles have landed anyway.
//   * fused statistics use their own 8 KiB of LDS (the stages are live) and re-align the two wave groups for
//     their exchange: group 0 takes one extra barrier before it, group 1 one after it.
// ---------------------------------------------------------------------------------------------------------
constexpr int PART_BYTES = 2 * 4 * 256 * 4;

__global__ __launch_bounds__(512, 2) void igemm256p_kernel(Igemm256Params P) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const IgemmParams &p = P.g;
  constexpr int ES = 2, EPC = 8, BKE = 64, PD = 6;
  typedef bf16x8 frag_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const float act_ns = act_neg_scale(p.act, p.slope);
  const bool act_any = p.act == CFD_ACT_LEAKY || p.act == CFD_ACT_RELU;   // (uniform)
  const bool act_sig = p.act == CFD_ACT_SIGMOID;   // (uniform; never set by the nets for these kernels)
  const int S = (p.kind == CFD_CONV_K4S2) ? 2 : 1;
  const int ktiles = p.ntaps * p.cpt;

  // ---- this workgroup's tile list
  const int total = p.nblocks, gx = __builtin_amdgcn_readfirstlane((int)gridDim.x >> 3);
  const int xcd = __builtin_amdgcn_readfirstlane((int)blockIdx.x & 7);
  const int jloc = __builtin_amdgcn_readfirstlane((int)blockIdx.x >> 3);
  const int q8 = total >> 3, r8 = total & 7;
  const int xstart = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int xcnt = q8 + (xcd < r8 ? 1 : 0);
  if (jloc >= xcnt) return;   // (whole workgroup; the host sizes the grid so that this never happens)

  auto decode = [&](int id, int &nt, int &mt, int &cls, int &sk) {
    nt = id % p.ntiles; id /= p.ntiles;
    if (P.cls_fast) { cls = id % p.nclass; id /= p.nclass; mt = id % p.mtiles; id /= p.mtiles; }
    else { mt = id % p.mtiles; id /= p.mtiles; cls = id % p.nclass; id /= p.nclass; }
    sk = id;
  };

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)P.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)P.wbytes, 0x00020000);

  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;

  // ---- issue-side state
  unsigned avoff[4], rowoff[4], vmask[4], cur[4];
  int iss_i = jloc, iss_cls = 0, iss_left = 0, iss_t = 0, iss_c = 0;
  unsigned s_a_ = 0, s_b_ = 0;
  auto tap_switch = [&]() {
    const int info = P.tapinfo[iss_cls][iss_t];
    // (readfirstlane: these are uniform by construction; said explicitly, or the compiler wraps every LDS-DMA load
    //  whose scalar offset it cannot prove uniform in a waterfall loop)
    s_a_ = (unsigned)__builtin_amdgcn_readfirstlane(((info & 255) * p.Cin + iss_c * BKE) * ES);
    s_b_ = (unsigned)__builtin_amdgcn_readfirstlane(iss_c * 128);
    const int tapx = (info >> 8) * p.ldx * ES;
#pragma unroll
    for (int r = 0; r < 4; ++r) cur[r] = ((vmask[r] >> iss_t) & 1u) ? rowoff[r] + (unsigned)tapx : OOB;
  };
  auto iss_setup = [&](int id) {
    int nt, mt, cls, sk;
    decode(id, nt, mt, cls, sk);
    iss_cls = __builtin_amdgcn_readfirstlane(cls);
    const int kbeg = k_slice(ktiles, sk, p.split_k);
    const int kend = k_slice(ktiles, sk + 1, p.split_k);
    iss_left = __builtin_amdgcn_readfirstlane(kend - kbeg);
    iss_t = __builtin_amdgcn_readfirstlane(kbeg / p.cpt);
    iss_c = __builtin_amdgcn_readfirstlane(kbeg - iss_t * p.cpt);
    const int m0 = mt * 256, n0 = nt * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      avoff[r] = (unsigned)(((n0 + (r >> 1) * 128 + (r & 1) * 64 + wave * 8 + lrow) * p.Kw + chunk * EPC) * ES);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + (r >> 1) * 128 + (r & 1) * 64 + wave * 8 + lrow;
      const bool mv = m < p.M;
      const int mm = mv ? m : 0;
      int b, qy, qx;
      split_pixel(p, mm, b, qy, qx);
      const int y0 = qy * S, x0 = qx * S;
      rowoff[r] = (unsigned)((((b * p.Hi + y0) * p.Wi + x0) * p.ldx + chunk * EPC) * ES);
      vmask[r] = tap_mask(P.tgeo[cls], y0, x0, p.Hi, p.Wi, mv);
    }
    tap_switch();
  };
  auto iss_advance = [&]() {
    if (__builtin_amdgcn_readfirstlane(iss_left) > 1) {   // scalar branches: everything assigned below stays uniform
      --iss_left;
      s_a_ = (unsigned)__builtin_amdgcn_readfirstlane(s_a_ + 128);
      s_b_ = (unsigned)__builtin_amdgcn_readfirstlane(s_b_ + 128);
      if (__builtin_amdgcn_readfirstlane(++iss_c) == p.cpt) { iss_c = 0; ++iss_t; tap_switch(); }
    } else if (__builtin_amdgcn_readfirstlane(iss_i + gx) < xcnt) {   // this tile is fully issued: the stream moves on to the next one
      iss_i = __builtin_amdgcn_readfirstlane(iss_i + gx);
      iss_setup(xstart + iss_i);
    }                                 // (after the last tile: the last K tile is issued again, nobody reads it)
  };

  char *const ldsw = lds + wave * 1024;
  auto issue = [&](int slot, int stage) {
    char *dst = ldsw + stage * ST_BYTES + slot * HT_BYTES;
    const unsigned s_a = __builtin_amdgcn_readfirstlane(s_a_), s_b = __builtin_amdgcn_readfirstlane(s_b_);
    if (slot == 0) {
      bload16(wrs, avoff[0], s_a, dst); bload16(wrs, avoff[1], s_a, dst + 8192);
    } else if (slot == 3) {
      bload16(wrs, avoff[2], s_a, dst); bload16(wrs, avoff[3], s_a, dst + 8192);
    } else if (slot == 1) {
      bload16(xrs, cur[0], s_b, dst); bload16(xrs, cur[1], s_b, dst + 8192);
    } else {
      bload16(xrs, cur[2], s_b, dst); bload16(xrs, cur[3], s_b, dst + 8192);
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int laneoff = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const char *ard[2], *brd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    ard[s] = lds + ((wr * 64 * 128 + laneoff) ^ (s << 6));
    brd[s] = lds + ((wc * 32 * 128 + laneoff) ^ (s << 6));
  }

  // ---- prologue of the whole stream: H0 .. H5
  iss_setup(xstart + iss_i);
#pragma unroll
  for (int h = 0; h < 6; ++h) {
    issue(h & 3, (h >> 2) & 1);
    if ((h & 3) == 3) iss_advance();
  }
  wait_vmcnt<6>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // second group runs one barrier behind

  frag_t a[4][2], b0[2][2], b1[2][2];
  f32x4 acc[8][4];

#define CFD_MFMA(AI, BJ, BREG)                                                         \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                \
            acc[AI + i][BJ + j] = mma(a[i][s], BREG[j][s], acc[AI + i][BJ + j]);
#define CFD_PHASE2(ST, SLOTA, STA, SLOTB, STB, VM, READS, MFMAS)                        \
    {                                                                                  \
      READS                                                                            \
      issue(SLOTA, STA); issue(SLOTB, STB);                                            \
      if (SLOTB == 3) iss_advance();                                                   \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
      wait_vmcnt<VM>();                                                                \
      __builtin_amdgcn_s_barrier();                                                    \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_setprio(1);                                                   \
      MFMAS                                                                            \
      __builtin_amdgcn_s_setprio(0);                                                   \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_barrier();                                                    \
    }
#define CFD_KTILE2(ST)                                                                 \
    CFD_PHASE2(ST, 2, (ST + 1) & 1, 3, (ST + 1) & 1, 8,                                \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
          b0[j][s] = *(const frag_t *)(brd[s] + ST * ST_BYTES + 1 * HT_BYTES + j * 2048); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          a[i][s] = *(const frag_t *)(ard[s] + ST * ST_BYTES + 0 * HT_BYTES + i * 2048); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                  \
          b1[j][s] = *(const frag_t *)(brd[s] + ST * ST_BYTES + 2 * HT_BYTES + j * 2048); \
      },                                                                               \
      CFD_MFMA(0, 0, b0) CFD_MFMA(0, 2, b1))                                           \
    CFD_PHASE2(ST, 0, ST, 1, ST, 6,                                                    \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          a[i][s] = *(const frag_t *)(ard[s] + ST * ST_BYTES + 3 * HT_BYTES + i * 2048); \
      ,                                                                                \
      CFD_MFMA(4, 0, b0) CFD_MFMA(4, 2, b1))

  for (int ci = jloc; ci < xcnt; ci += gx) {
    int nt, mt, cls, sk;
    decode(xstart + ci, nt, mt, cls, sk);
    const int ph = cls >> 1, pw = cls & 1;
    const int m0 = mt * 256, n0 = nt * 256;
    const int nkt = __builtin_amdgcn_readfirstlane(
        k_slice(ktiles, sk + 1, p.split_k) - k_slice(ktiles, sk, p.split_k));
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k = 0; k < nkt; k += 2) {   // nkt is even (host)
      CFD_KTILE2(0)
      CFD_KTILE2(1)
    }

    // ---- fused BatchNorm statistics (as in igemm256_kernel; own LDS region, wave groups re-aligned around it)
    if (p.stat_sum) {
      if (wr == 0) __builtin_amdgcn_s_barrier();
      float *part = (float *)(lds + 2 * ST_BYTES);   // [2][4 wc][256 n]
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int nl = (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4 *)(p.bias + n0 + nl);
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      {
        const int which = tid >> 8, c = tid & 255;
        const float *pp = part + which * 1024 + c;
        const float tot = pp[0] + pp[256] + pp[512] + pp[768];
        const int b = m0 / (p.Hq * p.Wq);
        atomicAdd((which ? p.stat_sq : p.stat_sum) + (size_t)b * p.Cout + n0 + c, tot);
      }
      if (wr == 1) __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: lane holds channels n..n+3 (register index) of pixel m
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + fr;
      if (m >= p.M) continue;
      int b, qy, qx;
      split_pixel(p, m, b, qy, qx);
      int64_t opix;
      if (p.kind == CFD_CONVT_K4S2)
        opix = ((int64_t)b * p.Ho + 2 * qy + ph) * p.Wo + 2 * qx + pw;
      else
        opix = m;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = n0 + (i >> 2) * 128 + wr * 64 + (i & 3) * 16 + fq * 4;
        f32x4 v = acc[i][j];
        if (p.split_k > 1) {
          *(f32x4 *)(p.ws + ((int64_t)sk * p.B * p.Ho * p.Wo + opix) * p.Cout + n) = v;
          continue;
        }
        if (p.bias) v += *(const f32x4 *)(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (act_any) v[r] = apply_act_linear(v[r], act_ns);
        if (act_sig) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = 1.f / (1.f + __expf(-v[r]));
        }
        if (p.out_bf16) {
          bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
          *(bf16x4 *)(p.y + (opix * p.ldy + n) * 2) = o;
        } else {
          *(f32x4 *)(p.y + (opix * p.ldy + n) * 4) = v;
        }
      }
    }
    wait_vmcnt<0>();   // stores and DMA loads share the counter: start the next tile from a known state
  }
#undef CFD_KTILE2
#undef CFD_PHASE2
#undef CFD_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------------------------------------------------
// 128(n) x 256(m) variant for layers with Cout = 128 (mod 256): P d5, DeltaP g3, ...
// Same ping-pong machinery; a K tile is three half tiles (A, B0, B1 = 48 KiB), three stages (144 KiB), and ONE
// phase of 32 MFMAs per K tile:
//   wave (wr, wc) owns n rows wr*64 + [0,64) and m rows wc*64 + [0,64) (B half wc/2): 4 x 4 fragments.
//   phase k: read stage k%3; issue the six loads of K tile k+2 into stage (k+2)%3 (last read in phase k-1, whose
//   fragment reads every wave retired before its first barrier); lgkmcnt(0); vmcnt(6) (K tile k+1 landed); barrier;
//   32 MFMAs; barrier.
// ---------------------------------------------------------------------------------------------------------
constexpr int ST3_BYTES = 3 * HT_BYTES;
constexpr int BIAS128N_OFF = 3 * ST3_BYTES;   // 1 KiB after the three stages

template <typename T>
__global__ __launch_bounds__(512, 2) void igemm128n_kernel(Igemm256Params P) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const IgemmParams &p = P.g;
  constexpr int ES = sizeof(T);
  constexpr int EPC = 16 / ES;
  constexpr int BKE = 128 / ES;
  typedef typename std::conditional<sizeof(T) == 2, bf16x8, f32x4>::type frag_t;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const float act_ns = act_neg_scale(p.act, p.slope);
  const bool act_any = p.act == CFD_ACT_LEAKY || p.act == CFD_ACT_RELU;   // (uniform)
  const bool act_sig = p.act == CFD_ACT_SIGMOID;   // (uniform; never set by the nets for these kernels)

  CFD_KARGS_BATCH(P)
  int bid = blockIdx.x;
  {
    const int nwg = k_nblocks, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int nt, mt, cls;
  {
    int q = fastdiv(bid, kdv[0]);
    nt = bid - q * k_ntiles; bid = q;
    if (k_cls_fast) {
      q = fastdiv(bid, kdv[1]); cls = bid - q * k_nclass; bid = q;
      q = fastdiv(bid, kdv[2]); mt = bid - q * k_mtiles; bid = q;
    } else {
      q = fastdiv(bid, kdv[2]); mt = bid - q * k_mtiles; bid = q;
      q = fastdiv(bid, kdv[1]); cls = bid - q * k_nclass; bid = q;
    }
  }
  const int sk = bid;
  int tg8[8];   // this class's tap grid, one batch (tap_mask and the tap walk read it)
#pragma unroll
  for (int i = 0; i < 8; ++i) tg8[i] = P.tgeo[cls][i];
  asm volatile("" ::"s"(tg8[0]), "s"(tg8[1]), "s"(tg8[2]), "s"(tg8[3]), "s"(tg8[4]), "s"(tg8[5]), "s"(tg8[6]), "s"(tg8[7]));
  const int ph = cls >> 1, pw = cls & 1;
  const int m0 = mt * 256, n0 = nt * 128;
  const int S = (p.kind == CFD_CONV_K4S2) ? 2 : 1;

  const int ktiles = p.ntaps * p.cpt;
  const int kbeg = k_slice(ktiles, sk, p.split_k);
  const int kend = k_slice(ktiles, sk + 1, p.split_k);
  const int nkt = kend - kbeg;

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)P.xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, 0, (int)P.wbytes, 0x00020000);

  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  unsigned avoff[2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
    avoff[r] = (unsigned)(((n0 + r * 64 + wave * 8 + lrow) * p.Kw + chunk * EPC) * ES);
  unsigned rowoff[4], vmask[4], cur[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + (r >> 1) * 128 + (r & 1) * 64 + wave * 8 + lrow;
    const bool mv = m < p.M;
    const int mm = mv ? m : 0;
    int b, qy, qx;
    split_pixel(p, mm, b, qy, qx);
    const int y0 = qy * S, x0 = qx * S;
    rowoff[r] = (unsigned)((((b * p.Hi + y0) * p.Wi + x0) * p.ldx + chunk * EPC) * ES);
    vmask[r] = tap_mask(tg8, y0, x0, p.Hi, p.Wi, mv);
  }

  int iss_left = nkt;
  int iss_t = kbeg ? kbeg / p.cpt : 0, iss_c = kbeg - iss_t * p.cpt;
  unsigned s_a = 0, s_b = 0;
  // A tap switch sits inside a memory cluster, i.e. on the critical path of its phase, and layers with few K tiles per
  // tap (Cin 64..256) switch every 1..4 K tiles.  Looking the tap up in the kernel-argument table cost a dependent
  // scalar load (a few hundred cycles, and its wait also drains the fragment reads sharing lgkmcnt) per switch; the
  // taps of every kind form an affine grid (checked by the host), so the walk is a handful of scalar adds instead:
  // tw_pix = (dy*Wi + dx) of the current tap, tw_wt = its weight tap index, (tw_kx) its column in the kh x kw grid.
  TapWalk tw;
  tw.init(tg8, p.Wi, iss_t);
  auto tap_switch = [&]() {
    s_a = (unsigned)((tw.wt * p.Cin + iss_c * BKE) * ES);
    s_b = (unsigned)(iss_c * 128);
    const int tapx = tw.pix * p.ldx * ES;
#pragma unroll
    for (int r = 0; r < 4; ++r) cur[r] = ((vmask[r] >> iss_t) & 1u) ? rowoff[r] + (unsigned)tapx : OOB;
  };
  auto iss_advance = [&]() {
    if (iss_left > 1) {
      --iss_left;
      s_a += 128; s_b += 128;
      if (++iss_c == p.cpt) { iss_c = 0; ++iss_t; tw.next(); tap_switch(); }
    }
  };
  tap_switch();

  char *const ldsw = lds + wave * 1024;
  auto issue_ktile = [&](int stage) {   // compile-time constant at every call site
    char *dst = ldsw + stage * ST3_BYTES;
    bload16(wrs, avoff[0], s_a, dst); bload16(wrs, avoff[1], s_a, dst + 8192);
    bload16(xrs, cur[0], s_b, dst + HT_BYTES); bload16(xrs, cur[1], s_b, dst + HT_BYTES + 8192);
    bload16(xrs, cur[2], s_b, dst + 2 * HT_BYTES); bload16(xrs, cur[3], s_b, dst + 2 * HT_BYTES + 8192);
    iss_advance();
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int laneoff = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const char *ard[2], *brd[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    ard[s] = lds + ((wr * 64 * 128 + laneoff) ^ (s << 6));
    brd[s] = lds + HT_BYTES + (wc >> 1) * HT_BYTES + (((wc & 1) * 64 * 128 + laneoff) ^ (s << 6));
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (p.bias) {   // (as in igemm256_kernel: the n tile's bias rides to LDS ahead of the first loads)
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)p.bias, 0, p.Cout * 4, 0x00020000);
    bload16(brs, (unsigned)(n0 * 4 + lane * 16), 0u, lds + BIAS128N_OFF);
  }
  issue_ktile(0);
  issue_ktile(1);
  wait_vmcnt<6>();   // K tile 0 landed
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();

  frag_t a[4][2], b[4][2];
#define CFD_KT3(ST)                                                                    \
    {                                                                                  \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
          a[i][s] = *(const frag_t *)(ard[s] + ST * ST3_BYTES + i * 2048);             \
          b[i][s] = *(const frag_t *)(brd[s] + ST * ST3_BYTES + i * 2048);             \
        }                                                                              \
      }                                                                                \
      issue_ktile((ST + 2) % 3);                                                       \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
      wait_vmcnt<6>();                                                                 \
      __builtin_amdgcn_s_barrier();                                                    \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_setprio(1);                                                   \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                  \
          _Pragma("unroll") for (int j = 0; j < 4; ++j)                                \
            acc[i][j] = mma(a[i][s], b[j][s], acc[i][j]);                              \
      __builtin_amdgcn_s_setprio(0);                                                   \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_barrier();                                                    \
    }
  int k = 0;
  for (; k + 2 < nkt; k += 3) {
    CFD_KT3(0)
    CFD_KT3(1)
    CFD_KT3(2)
  }
  if (k < nkt) {
    CFD_KT3(0)
    if (k + 1 < nkt) { CFD_KT3(1) }
  }
#undef CFD_KT3
  if (wr == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();

  if (p.stat_sum) {
    __syncthreads();
    float *part = (float *)lds;   // [2][4 wc][128 n]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = wr * 64 + i * 16 + fq * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *(const f32x4 *)(lds + BIAS128N_OFF + nl * 4);
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
      const int b = m0 / (p.Hq * p.Wq);
      atomicAdd((which ? p.stat_sq : p.stat_sum) + (size_t)b * p.Cout + n0 + c, tot);
    }
  }

  // (output mode chosen once per launch, as in igemm256_kernel)
  auto direct_epi = [&](auto mode_c) {
  constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wc * 64 + j * 16 + fr;
    if (m >= p.M) continue;
    int b, qy, qx;
    split_pixel(p, m, b, qy, qx);
    int64_t opix;
    if (p.kind == CFD_CONVT_K4S2)
      opix = ((int64_t)b * p.Ho + 2 * qy + ph) * p.Wo + 2 * qx + pw;
    else
      opix = m;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + wr * 64 + i * 16 + fq * 4;
      f32x4 v = acc[i][j];
      if constexpr (MODE == 0) {
        *(f32x4 *)(p.ws + ((int64_t)sk * p.B * p.Ho * p.Wo + opix) * p.Cout + n) = v;
      } else {
        if (p.bias) v += *(const f32x4 *)(lds + BIAS128N_OFF + (n - n0) * 4);
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
  }
  };
  if (p.split_k > 1) direct_epi(std::integral_constant<int, 0>{});
  else if (p.out_bf16) direct_epi(std::integral_constant<int, 1>{});
  else direct_epi(std::integral_constant<int, 2>{});
}

static int64_t x_bytes(const IgemmParams &p, int es) { return (((int64_t)p.B * p.Hi * p.Wi - 1) * p.ldx + p.Cin) * es; }
static int64_t w_bytes(const IgemmParams &p, int es) { return (int64_t)p.Cout * p.Kw * es; }

bool igemm256_eligible(const IgemmParams &p, int dtype) {
  const int es = dtype == CFD_BF16 ? 2 : 4;
  return p.Cout % 128 == 0 && p.Cin % (128 / es) == 0 && p.ntaps <= 16 && x_bytes(p, es) < (1LL << 31) &&
         w_bytes(p, es) < (1LL << 31);
}

// n-tile of the variant that runs this layer: 256 when Cout allows it, else the 128 x 256 variant
// n-tile of the variant that runs this layer: 256 when Cout allows it, else the 128 x 256 variant -- and also the 128 x 256
// variant for bf16 layers that give the 256 x 256 kernel at most half a chip of tiles (small M: U e4 / d2 and every
// discriminator layer at B = 16, P e4 .. d2 at B = 128): twice the blocks means half the split-K factor, i.e. half the fp32
// slab traffic, or none (measured per layer, scripts/layer_bench.py --ntile 128: 8-20 % faster up to 128 tiles, slower from
// 256 tiles on)
int igemm256_ntile(const IgemmParams &p) {
  if (p.Cout % 256 != 0) return 128;
  if (p.pool) return 256;
  const int t = tuning(CFD_TUNE_NTILE);
  if (t == 128 || t == 256) return t;
  const bool bf = p.cpt * 64 == p.Cin;
  const int64_t blocks256 = (int64_t)cdiv(p.M, 256) * (p.Cout / 256) * p.nclass;
  return bf && blocks256 <= 128 ? 128 : 256;
}

int igemm256_block_count(const IgemmParams &p) {
  return cdiv(p.M, 256) * (p.Cout / igemm256_ntile(p)) * p.nclass * p.split_k;
}

// Cout = 64 (mod 128), bf16, patch-eligible, no split-K: the co-resident patch kernel with a 64-wide n tile (the only kernel
// that does not spend half its MFMAs on an empty half tile there -- DeltaP g3's data gradient, 128 -> 64 channels)
bool igemm_n64_eligible(const IgemmParams &pin, int dtype) {
  if (dtype != CFD_BF16 || pin.Cout % 128 != 64 || pin.split_k != 1 || pin.pool || pin.stat_sum) return false;
  if (tuning(CFD_TUNE_PATCH) == 0 || tuning(CFD_TUNE_IGEMM256) == 0 || tuning(CFD_TUNE_IGEMM256_PD) >= 0) return false;
  if (pin.Cin % 64 != 0 || pin.ntaps > 16 || x_bytes(pin, 2) >= (1LL << 31) || w_bytes(pin, 2) >= (1LL << 31)) return false;
  return igemm256h_eligible(pin, dtype, 64) && (int64_t)cdiv(pin.M, 256) * (pin.Cout / 64) * pin.nclass >= 256;
}

int igemm_n64_launch(const IgemmParams &pin, hipStream_t st) {
  IgemmParams p = pin;
  CFD_REQUIRE(igemm_n64_eligible(p, CFD_BF16), "64-wide co-resident kernel: shape not eligible");
  p.mtiles = cdiv(p.M, 256);
  p.ntiles = p.Cout / 64;
  p.nblocks = p.mtiles * p.ntiles * p.nclass;
  const int co = tuning(CFD_TUNE_CLASS_ORDER);
  const int cls_fast = p.nclass > 1 && (co < 0 ? x_bytes(p, 2) > w_bytes(p, 2) : co == 1);
  return igemm256h_launch(p, 64, (unsigned)x_bytes(p, 2), (unsigned)w_bytes(p, 2), cls_fast, 0, st, true);
}

int igemm256_launch(const IgemmParams &pin, int dtype, hipStream_t st) {
  const int es = dtype == CFD_BF16 ? 2 : 4;
  Igemm256Params P;
  memset(&P, 0, sizeof(P));
  P.g = pin;
  IgemmParams &p = P.g;
  p.mtiles = cdiv(p.M, 256);
  int ntile = igemm256_ntile(p);
  // CFD_TUNE_PATCH: 1 = patch kernels, never the co-resident one; 2 = the co-resident patch kernel (two workgroups per CU,
  // igemm256h.hip) for every layer that takes 128-wide n tiles anyway; 3 = for every patch-eligible layer (a 256-wide layer
  // then runs as two 128-wide n tiles); -1 = heuristic below
  // Measured (scripts/layer_bench.py --patch=-1,2,3, B = 128, one box): the co-resident kernel's 64 x 64 wave tiles read
  // 1.4 x the LDS bytes per FLOP of the 256-wide kernel's 128 x 64 ones and its K loop runs at 0.55 of the MFMA peak where
  // igemm256h reaches 0.64 (U d5 1 378 -> 1 580 us, U d4 690 -> 794, DeltaP g4 487 -> 537); against igemm128nh it is even
  // on long K loops (P d5 452.3 -> 451.8 us) and wins where the per-tile overhead dominates (DeltaP g3, 9 K tiles of 64
  // channels: 215.9 -> 178.7 us).  Heuristic (-1): 128-wide layers with at most 16 such K tiles.
  const int pmode = tuning(CFD_TUNE_PATCH);
  const bool cores_want = pmode == 3 || (ntile == 128 && (pmode == 2 || (pmode < 0 && p.ntaps * (p.Cin / 64) <= 16)));
  const bool cores = cores_want && tuning(CFD_TUNE_IGEMM256_PD) < 0 && !p.pool && igemm256h_eligible(p, dtype, 128);
  if (cores) ntile = 128;
  p.ntiles = p.Cout / ntile;
  p.nblocks = p.mtiles * p.ntiles * p.nclass * p.split_k;
  P.xbytes = (unsigned)x_bytes(p, es);
  P.wbytes = (unsigned)w_bytes(p, es);
  // block order of the sub-pixel classes: interleaved (the four classes of a pixel tile run side by side on one XCD
  // and share the input rows in its L2) when the activations outweigh the weights, else class-major (each class
  // streams only its own quarter of the weights)
  const int co = tuning(CFD_TUNE_CLASS_ORDER);
  fastdiv_make((unsigned)p.ntiles, P.dv[0]); fastdiv_make((unsigned)p.nclass, P.dv[1]); fastdiv_make((unsigned)p.mtiles, P.dv[2]);
  P.cls_fast = p.nclass > 1 && (co < 0 ? x_bytes(p, es) > w_bytes(p, es) : co == 1);
  for (int cls = 0; cls < p.nclass; ++cls)
    for (int t = 0; t < p.ntaps; ++t) {
      int dy, dx, wt;
      tap_geom(p.kind, cls >> 1, cls & 1, t, dy, dx, wt);
      P.tapinfo[cls][t] = ((dy * p.Wi + dx) << 8) | wt;
      P.tapdy[cls][t] = (short)dy;
      P.tapdx[cls][t] = (short)dx;
    }
  for (int cls = 0; cls < p.nclass; ++cls) {
    const int kw = p.kind == CFD_CONV_K3S1 ? 3 : (p.kind == CFD_DENSE ? 1 : (p.kind == CFD_CONVT_K4S2 ? 2 : 4));
    const int kh = p.ntaps / kw;
    int *g = P.tgeo[cls];
    g[0] = kw; g[1] = kh;
    g[2] = P.tapdy[cls][0]; g[3] = kh > 1 ? P.tapdy[cls][kw] - P.tapdy[cls][0] : 0;
    g[4] = P.tapdx[cls][0]; g[5] = kw > 1 ? P.tapdx[cls][1] - P.tapdx[cls][0] : 0;
    const int wt0 = P.tapinfo[cls][0] & 255;
    const int wxs = kw > 1 ? (P.tapinfo[cls][1] & 255) - wt0 : 0, wys = kh > 1 ? (P.tapinfo[cls][kw] & 255) - wt0 : 0;
    g[6] = wt0; g[7] = (wys << 16) | (wxs & 0xffff);
    CFD_REQUIRE(wxs >= 0 && wys >= 0 && wxs < 256 && wys < 256, "weight taps of kind %d do not ascend", p.kind);
    for (int t = 0; t < p.ntaps; ++t)
      CFD_REQUIRE(kh * kw == p.ntaps && P.tapdy[cls][t] == g[2] + g[3] * (t / kw) && P.tapdx[cls][t] == g[4] + g[5] * (t % kw) &&
                      (P.tapinfo[cls][t] & 255) == wt0 + wys * (t / kw) + wxs * (t % kw),
                  "taps of kind %d are not an affine grid", p.kind);
  }
  // staged bf16 output: whole-tile kernel only (n tile 256), no split-K, 16-B aligned rows
  P.stage_out = ntile == 256 && p.out_bf16 && p.split_k == 1 && p.ldy % 8 == 0 && ((uintptr_t)p.y & 15) == 0 &&
                p.act != CFD_ACT_SIGMOID && tuning(CFD_TUNE_STAGED_OUT) != 0;
  const int pdt = tuning(CFD_TUNE_IGEMM256_PD);
  if (p.pool) {
    // (cfd_conv_forward_pool2 checked the shape: stride-1 conv, no split-K, Wq % 128 == 0, Hq even, monotone activation)
    CFD_REQUIRE(ntile == 256 && p.nclass == 1 && p.split_k == 1 && !p.stat_sum && p.M % 256 == 0, "pooled epilogue: bad launch");
    CFD_REQUIRE(p.pool == 1 || (pdt < 0 && igemm256h_eligible(p, dtype, ntile) && p.y2),
                "pooled epilogue that keeps the unpooled tensor: patch-staged kernel only");
    P.stage_out = 0;
  }
  const size_t shmem = (size_t)BIAS256_OFF + 1024;   // stages / staged output tile, then the n tile's bias
  // stride-1 tap grids on whole 8 x 32 pixel tiles: the activation operand is staged once per channel chunk (igemm256h.hip)
  if (pdt < 0 && igemm256h_eligible(p, dtype, ntile))
    return igemm256h_launch(p, ntile, P.xbytes, P.wbytes, P.cls_fast, P.stage_out, st, cores);
  // persistent form (igemm256p_kernel): bf16, 256-wide n tiles, more tiles than CUs, every K range even
  // (CFD_TUNE_IGEMM256_PD: 36 forces it where legal, 26 forbids it)
  const int ktiles_all = p.ntaps * p.cpt;
  // (37 = forced on 8 workgroups only: the tests' way to walk several tiles per workgroup with small shapes)
  const bool p_legal = !p.pool && dtype == CFD_BF16 && ntile == 256 && ktiles_all % (2 * p.split_k) == 0 && p.nblocks >= 8;
  const int p_grid = pdt == 37 ? 8 : (p.nblocks >= 256 ? 256 : (p.nblocks / 8) * 8);
  constexpr bool P_DEFAULT = false;   // until measured: the persistent kernel runs only when asked for
  if (p_legal && (pdt == 36 || pdt == 37 || (P_DEFAULT && pdt < 0 && p.nblocks > 256))) {
    const size_t shp = 2 * ST_BYTES + PART_BYTES;
    if (int e_ = set_max_lds((const void *)igemm256p_kernel, (int)shp)) return e_;
    hipLaunchKernelGGL(igemm256p_kernel, dim3(p_grid), dim3(512), shp, st, P);
    CFD_LAUNCH_CHECK();
    return CFD_OK;
  }
  // variants (cfd_set_tuning(CFD_TUNE_IGEMM256_PD, v)): default / 26 = two phases of 32 MFMAs per K tile;
  // 5, 6 = four phases of 16, prefetch distance 5 / 6 half tiles; 106, 126 = diagnostic builds with cycle stamps
#define CFD_LAUNCH256(...)                                                                                          \
  do {                                                                                                              \
    if (int e_ = set_max_lds((const void *)igemm256_kernel<__VA_ARGS__>, (int)BIAS256_OFF + 1024)) return e_;       \
    hipLaunchKernelGGL((igemm256_kernel<__VA_ARGS__>), dim3(p.nblocks), dim3(512), shmem, st, P);                   \
  } while (0)
  if (ntile == 128) {
    const size_t shmem3 = BIAS128N_OFF + 1024;
    if (int e_ = set_max_lds(dtype == CFD_F32 ? (const void *)igemm128n_kernel<float> : (const void *)igemm128n_kernel<bf16>, (int)shmem3)) return e_;
    if (dtype == CFD_F32) hipLaunchKernelGGL(igemm128n_kernel<float>, dim3(p.nblocks), dim3(512), shmem3, st, P);
    else hipLaunchKernelGGL(igemm128n_kernel<bf16>, dim3(p.nblocks), dim3(512), shmem3, st, P);
  } else if (dtype == CFD_F32) CFD_LAUNCH256(float, 6, 0, 4);
  else if (pdt == 226) CFD_LAUNCH256(bf16, 6, 2, 2);   // tile-stage stamps (scripts/tile_timeline.py)
  else if (pdt == 126) CFD_LAUNCH256(bf16, 6, 1, 2);
  else if (pdt == 106) CFD_LAUNCH256(bf16, 6, 1, 4);
  else if (pdt == 5) CFD_LAUNCH256(bf16, 5, 0, 4);
  else if (pdt == 6) CFD_LAUNCH256(bf16, 6, 0, 4);
  else CFD_LAUNCH256(bf16, 6, 0, 2);
#undef CFD_LAUNCH256
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

}  // namespace cfd
