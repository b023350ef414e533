// Implicit-GEMM convolutions on gfx950 MFMA (bf16 16x16x32 and exact-f32 16x16x4).
//
// Replaces the Conv2D / Conv2DTranspose / Dense arithmetic Keras runs for
//   U/models/generator.py:18,31,59   P/models/generator.py:43,56,85
//   U/models/discriminator.py:25-41  DeltaP/models/generator.py:28-46
//
// GEMM view (per output "class"):   C[n][m] = sum_k  W[n][k] * X[k][m]
//   n = output channel,  m = (b, qy, qx) position in the class grid,  k = (tap, ci)
//   conv kinds  : one class, class grid = output grid, tap = (kh,kw), input pixel
//                 (qy*S - pad + kh, qx*S - pad + kw), zero outside the image (TF 'same')
//   transposed  : four classes = output parity (ph,pw); class grid = INPUT grid; each class is a
//                 2x2-tap convolution (sub-pixel decomposition -> no scatter-add, no atomics);
//                 output pixel (2qy+ph, 2qx+pw);  oh = 2*ih + kh - 1
//   dense       : 1x1 grid, one tap
// The weight operand is the MFMA "A" side (rows = n) and the gathered activations the "B" side
// (cols = m), so that every lane ends up holding 4 CONSECUTIVE channels of one pixel and the
// NHWC store is 8 B (bf16) / 16 B (f32) per lane.
//
// Tile: 128(n) x 128(m) x 128 bytes of K (64 bf16 / 32 f32), 256 threads = 2x2 waves of 64x64,
// double-buffered LDS (64 KiB) filled by global_load_lds (16 B per lane).  The LDS image is
// lane-linear (rows of 128 B); bank conflicts are avoided by XOR-swizzling the 16-B chunk index
// with (row & 7) on the per-lane SOURCE address and on the fragment read.  Out-of-image taps
// and rows past M/N read a 16-B zero page, so padding costs no branches in the K loop.
#include "common.h"
#include "igemm.h"

namespace cfd {


__device__ __attribute__((aligned(16))) const unsigned int g_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};

template <typename T> struct Elem;
template <> struct Elem<bf16> { static constexpr int EPC = 8; static constexpr int BKE = 64; };
template <> struct Elem<float> { static constexpr int EPC = 4; static constexpr int BKE = 32; };

constexpr int TILE_BYTES = 128 * 128;      // one operand tile: 128 rows x 128 B
constexpr int STAGE_BYTES = 2 * TILE_BYTES;


// MT = rows of the m tile: 128, or 32 for the layers with a handful of output pixels per class (round 3: at batch 1 -- the
// reference's own batch size -- U's e5, b, d1, d2 have 8 or 32 pixels; on a 128-row tile their fp32 MFMAs ran 94 % on
// padding and the weight-bound layers took 47-80 us for 67-134 MB of weights).  MT = 32: the four waves split the 128
// n rows (32 x 32 wave tiles), a stage is 16 + 4 KiB, four workgroups share a CU and the host splits K further.
template <typename T, int MT = 128>
__global__ __launch_bounds__(256, MT == 128 ? 2 : 4) void igemm_kernel(IgemmParams p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BKE = Elem<T>::BKE;
  constexpr int ES = sizeof(T);
  constexpr int XR = MT / 32;                       // staging rounds of the activation tile (32 rows each)
  constexpr int STAGE = TILE_BYTES + MT * 128;      // weights 128 x 128 B + activations MT x 128 B
  constexpr int NI = MT == 128 ? 4 : 2, MJ = MT == 128 ? 4 : 2;   // 16 x 16 fragments per wave: n x m
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware block remap (bijective): blocks that share an XCD get a contiguous id range
  int bid = blockIdx.x;
  {
    const int nwg = p.nblocks, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = bid % p.ntiles; bid /= p.ntiles;
  const int mt = bid % p.mtiles; bid /= p.mtiles;
  const int cls = bid % p.nclass;
  const int sk = bid / p.nclass;
  const int ph = cls >> 1, pw = cls & 1;
  const int m0 = mt * MT, n0 = nt * 128;

  const int S = (p.kind == CFD_CONV_K4S2) ? 2 : 1;

  // ---- per-lane staging rows: round i covers tile rows (i*4+wave)*8 + lane/8, chunk slot lane%8
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;  // logical 16-B chunk this lane fetches (swizzle on the source)
  int ybase[4], xbase[4], pix0[4];
  bool mval[4];
  const char *wrow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = (i * 4 + wave) * 8 + lrow;
    if (i < XR) {
      const int m = m0 + R;
      mval[i] = m < p.M;
      const int mm = mval[i] ? m : 0;
      int b, qy, qx;
      split_pixel(p, mm, b, qy, qx);
      ybase[i] = qy * S;
      xbase[i] = qx * S;
      pix0[i] = (b * p.Hi + qy * S) * p.Wi + qx * S;
    }
    const int n = n0 + R;
    wrow[i] = (n < p.Cout) ? p.w + ((int64_t)n * p.Kw + chunk * EPC) * ES : nullptr;
  }
  const char *zero = (const char *)g_zero_page;

  const int ktiles = p.ntaps * p.cpt;
  const int kbeg = k_slice(ktiles, sk, p.split_k);
  const int kend = k_slice(ktiles, sk + 1, p.split_k);

  // K tiles are staged strictly in order, so the (tap, channel chunk) position is a pair of counters, and everything
  // that depends on the tap only -- geometry, the in-image test, the 64-bit row pointers -- is computed at a tap
  // switch; a K tile then costs one add per pointer (this kernel has only 32 MFMAs per wave per K tile to hide
  // its address arithmetic behind)
  int st_t = kbeg / p.cpt, st_c = kbeg - st_t * p.cpt;
  const char *xrow[4], *wtap_row[4];
  auto tap_setup = [&]() {
    int dy, dx, wtap;
    tap_geom(p.kind, ph, pw, st_t, dy, dx, wtap);
    const int64_t woff = (int64_t)wtap * p.Cin * ES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wtap_row[i] = wrow[i] ? wrow[i] + woff : nullptr;
      if (i < XR) {
        const int iy = ybase[i] + dy, ix = xbase[i] + dx;
        const bool ok = mval[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        xrow[i] = ok ? p.x + ((int64_t)(pix0[i] + dy * p.Wi + dx) * p.ldx + chunk * EPC) * ES : nullptr;
      }
    }
  };
  tap_setup();
  auto stage = [&](int /*kt*/, int buf) {
    char *wdst = lds + buf * STAGE + wave * 1024;
    char *xdst = wdst + TILE_BYTES;
    const int cib = st_c * BKE * ES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(wtap_row[i] ? wtap_row[i] + cib : zero, wdst + i * 4096);
      if (i < XR) glds16(xrow[i] ? xrow[i] + cib : zero, xdst + i * 4096);
    }
    if (++st_c == p.cpt) {
      st_c = 0;
      if (++st_t < p.ntaps) tap_setup();
    }
  };

  // ---- accumulators: wave tile 64(n) x 64(m) = 4x4 fragments of 16x16 (MT = 32: 32(n) x 32(m) = 2x2, waves along n)
  const int wm = MT == 128 ? (wave & 1) : 0, wn = MT == 128 ? (wave >> 1) : wave;
  const int noff = wn * (NI * 16), moff = wm * (MJ * 16);
  const float act_ns = act_neg_scale(p.act, p.slope);
  const bool act_sig = p.act == CFD_ACT_SIGMOID;
  f32x4 acc[NI][MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  // fragment byte offsets inside a tile for k-step 0 (k-step 1 is chunk + 4)
  int aoff[NI], boff[MJ];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ra = noff + i * 16 + fr;
    aoff[i] = ra * 128 + ((fq ^ (ra & 7)) << 4);
  }
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int rb = moff + j * 16 + fr;
    boff[j] = TILE_BYTES + rb * 128 + ((fq ^ (rb & 7)) << 4);
  }

  if (kbeg < kend) stage(kbeg, 0);
  for (int kt = kbeg; kt < kend; ++kt) {
    const int buf = (kt - kbeg) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // tile kt has landed for every wave; every wave is done reading the other buffer
    if (kt + 1 < kend) stage(kt + 1, buf ^ 1);
    const char *base = lds + buf * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // chunk index for this k-step is (s*4 + fq); XOR with (row&7) only touches the low 3 bits, and
      // s*4 flips bit 2 -> equivalent to xor-ing the byte offset with 64
      if constexpr (sizeof(T) == 2) {
        bf16x8 a[NI], b[MJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = *(const bf16x8 *)(base + (aoff[i] ^ (s << 6)));
#pragma unroll
        for (int j = 0; j < MJ; ++j) b[j] = *(const bf16x8 *)(base + (boff[j] ^ (s << 6)));
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < MJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      } else {
        // f32: lane group fq supplies k = 16 s + 4 fq + e for e = 0..3 on BOTH operands (any
        // consistent k assignment sums to the same dot product), so one 16-B read feeds 4 MFMAs
        f32x4 a[NI], b[MJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = *(const f32x4 *)(base + (aoff[i] ^ (s << 6)));
#pragma unroll
        for (int j = 0; j < MJ; ++j) b[j] = *(const f32x4 *)(base + (boff[j] ^ (s << 6)));
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- fused BatchNorm statistics (host guarantees: split_k == 1, Cout % 128 == 0, the 128 pixel rows of a
  // tile lie in ONE sample, none ragged): see igemm256.hip
  if (MT == 128 && p.stat_sum) {
    __syncthreads();
    float *part = (float *)lds;   // [2][2 wm][128 n]
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int nl = noff + i * 16 + fq * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *(const f32x4 *)(p.bias + n0 + nl);
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
          part[(0 * 2 + wm) * 128 + nl + r] = t1;
          part[(1 * 2 + wm) * 128 + nl + r] = t2;
        }
      }
    }
    __syncthreads();
    {
      const int which = tid >> 7, c = tid & 127;
      const float *pp = part + which * 256 + c;
      const float tot = pp[0] + pp[128];
      const int b = m0 / (p.Hq * p.Wq);
      atomicAdd((which ? p.stat_sq : p.stat_sum) + (size_t)b * p.Cout + n0 + c, tot);
    }
  }

  // ---- epilogue: lane holds channels n..n+3 (reg index) of pixel m
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int m = m0 + moff + j * 16 + fr;
    if (m >= p.M) continue;
    int b, qy, qx;
    split_pixel(p, m, b, qy, qx);
    int64_t opix;
    if (p.kind == CFD_CONVT_K4S2)
      opix = ((int64_t)b * p.Ho + 2 * qy + ph) * p.Wo + 2 * qx + pw;
    else
      opix = m;  // class grid == output grid
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int n = n0 + noff + i * 16 + fq * 4;
      if (n >= p.Cout) continue;
      f32x4 v = acc[i][j];
      if (p.split_k > 1) {
        float *dst = p.ws + ((int64_t)sk * p.B * p.Ho * p.Wo + opix) * p.Cout + n;
        if (n + 3 < p.Cout) {
          *(f32x4 *)dst = v;
        } else {
          for (int r = 0; r < 4 && n + r < p.Cout; ++r) dst[r] = v[r];
        }
        continue;
      }
      if (n + 3 < p.Cout) {
        if (p.bias) {
          const f32x4 bv = *(const f32x4 *)(p.bias + n);
          v += bv;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act_linear(v[r], act_ns);
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
      } else {
        for (int r = 0; r < 4 && n + r < p.Cout; ++r) {
          float s = v[r] + (p.bias ? p.bias[n + r] : 0.f);
          s = apply_act(s, p.act, p.slope);
          if (p.out_bf16)
            *(bf16 *)(p.y + (opix * p.ldy + n + r) * 2) = (bf16)s;
          else
            *(float *)(p.y + (opix * p.ldy + n + r) * 4) = s;
        }
      }
    }
  }
}

// sum split-K slabs + bias + activation + convert + strided NHWC store.  One thread = four consecutive channels of one
// pixel (16-B slab loads), one pass, no grid cap; the scalar form serves Cout % 4 != 0 (the one-channel layers).
template <int V>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float *__restrict__ ws, const float *__restrict__ bias,
                                                             char *__restrict__ y, int64_t npix, int Cout, int ldy,
                                                             int split_k, int act, float slope, int out_bf16) {
  const int64_t total = npix * Cout;
  const int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (idx >= total) return;
  const int64_t pix = idx / Cout;
  const int n = (int)(idx - pix * Cout);
  if constexpr (V == 4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < split_k; ++k) s += *(const f32x4 *)(ws + (int64_t)k * total + idx);
    if (bias) s += *(const f32x4 *)(bias + n);
    const float ns = act_neg_scale(act, slope);
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = act == CFD_ACT_SIGMOID ? apply_act(s[r], act, slope) : apply_act_linear(s[r], ns);
    if (out_bf16) {
      bf16x4 o = {(bf16)s[0], (bf16)s[1], (bf16)s[2], (bf16)s[3]};
      *(bf16x4 *)(y + (pix * ldy + n) * 2) = o;
    } else {
      *(f32x4 *)(y + (pix * ldy + n) * 4) = s;
    }
  } else {
    float s = 0.f;
    for (int k = 0; k < split_k; ++k) s += ws[(int64_t)k * total + idx];
    if (bias) s += bias[n];
    s = apply_act(s, act, slope);
    if (out_bf16)
      *(bf16 *)(y + (pix * ldy + n) * 2) = (bf16)s;
    else
      *(float *)(y + (pix * ldy + n) * 4) = s;
  }
}

// The same sum for SMALL outputs behind a deep K split (round 3: U b at batch 1 is 8 pixels x 1 024 channels behind 128
// slabs): above, a thread walks all the slabs of its four channels -- 2 048 threads each with a chain of 128 loads.  Here a
// block owns 64 consecutive channel quads (1 KiB of every slab), its four waves take every fourth slab, eight loads in
// flight each, and fold through LDS.
__global__ __launch_bounds__(256) void splitk_epilogue_small_kernel(const float *__restrict__ ws, const float *__restrict__ bias,
                                                                   char *__restrict__ y, int64_t npix, int Cout, int ldy,
                                                                   int split_k, int act, float slope, int out_bf16) {
  __shared__ f32x4 part[4][64];
  const int64_t total = npix * Cout;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t idx = ((int64_t)blockIdx.x * 64 + lane) * 4;
  const bool live = idx < total;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    int k = wave;
    for (; k + 28 < split_k; k += 32) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4 *)(ws + (int64_t)(k + 4 * u) * total + idx);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < split_k; k += 4) s += *(const f32x4 *)(ws + (int64_t)k * total + idx);
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave != 0 || !live) return;
  s = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  const int64_t pix = idx / Cout;
  const int n = (int)(idx - pix * Cout);
  if (bias) s += *(const f32x4 *)(bias + n);
  const float ns = act_neg_scale(act, slope);
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] = act == CFD_ACT_SIGMOID ? apply_act(s[r], act, slope) : apply_act_linear(s[r], ns);
  if (out_bf16) {
    bf16x4 o = {(bf16)s[0], (bf16)s[1], (bf16)s[2], (bf16)s[3]};
    *(bf16x4 *)(y + (pix * ldy + n) * 2) = o;
  } else {
    *(f32x4 *)(y + (pix * ldy + n) * 4) = s;
  }
}

static int fill_params(const cfd_conv_desc *d, IgemmParams &p) {
  CFD_REQUIRE(d != nullptr, "desc is NULL");
  CFD_REQUIRE(d->dtype == CFD_BF16 || d->dtype == CFD_F32, "bad dtype %d", d->dtype);
  CFD_REQUIRE(d->out_dtype == CFD_BF16 || d->out_dtype == CFD_F32, "bad out_dtype %d", d->out_dtype);
  CFD_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0 && d->Cout > 0, "non-positive shape");
  const int bke = d->dtype == CFD_BF16 ? 64 : 32;
  CFD_REQUIRE(d->Cin % bke == 0, "Cin=%d must be a multiple of %d for the implicit-GEMM path (use cfd_conv_first)",
              d->Cin, bke);
  CFD_REQUIRE(d->ldx >= d->Cin && d->ldy >= d->Cout, "ldx/ldy smaller than the channel count");
  const int epc = d->dtype == CFD_BF16 ? 8 : 4;
  CFD_REQUIRE(d->ldx % epc == 0, "ldx=%d must keep 16-B alignment", d->ldx);
  CFD_REQUIRE(d->ldy % 4 == 0 || d->Cout < 4, "ldy=%d must be a multiple of 4", d->ldy);
  CFD_REQUIRE(d->split_k >= 1, "split_k must be >= 1");
  memset(&p, 0, sizeof(p));
  p.B = d->B; p.Hi = d->Hi; p.Wi = d->Wi; p.Cin = d->Cin; p.Cout = d->Cout; p.ldx = d->ldx; p.ldy = d->ldy;
  p.kind = d->kind; p.act = d->act; p.slope = d->leaky_slope; p.out_bf16 = d->out_dtype == CFD_BF16;
  p.nclass = 1;
  switch (d->kind) {
    case CFD_CONV_K4S2:
      CFD_REQUIRE(d->Hi % 2 == 0 && d->Wi % 2 == 0, "k4s2 'same' path needs even input (got %dx%d)", d->Hi, d->Wi);
      p.Ho = d->Hi / 2; p.Wo = d->Wi / 2; p.Hq = p.Ho; p.Wq = p.Wo; p.ntaps = 16; break;
    case CFD_CONV_K4S1:
    case CFD_CONV_K4S1_BWD: p.Ho = d->Hi; p.Wo = d->Wi; p.Hq = p.Ho; p.Wq = p.Wo; p.ntaps = 16; break;
    case CFD_CONV_K3S1: p.Ho = d->Hi; p.Wo = d->Wi; p.Hq = p.Ho; p.Wq = p.Wo; p.ntaps = 9; break;
    case CFD_CONVT_K4S2:
      p.Ho = d->Hi * 2; p.Wo = d->Wi * 2; p.Hq = d->Hi; p.Wq = d->Wi; p.ntaps = 4; p.nclass = 4; break;
    case CFD_DENSE:
      CFD_REQUIRE(d->Hi == 1 && d->Wi == 1, "dense needs Hi=Wi=1");
      p.Ho = 1; p.Wo = 1; p.Hq = 1; p.Wq = 1; p.ntaps = 1; break;
    default: CFD_FAIL(CFD_EINVAL, "unknown conv kind %d", d->kind);
  }
  const int alltaps = (d->kind == CFD_CONV_K3S1) ? 9 : (d->kind == CFD_DENSE ? 1 : 16);
  p.Kw = alltaps * d->Cin;
  const int64_t M = (int64_t)d->B * p.Hq * p.Wq;
  CFD_REQUIRE(M < (1LL << 31) && (int64_t)d->B * d->Hi * d->Wi < (1LL << 31), "too many pixels");
  p.M = (int)M;
  {
    const int hw = p.Hq * p.Wq;
    const bool pow2 = (hw & (hw - 1)) == 0 && (p.Wq & (p.Wq - 1)) == 0;
    p.hw_shift = pow2 ? __builtin_ctz((unsigned)hw) : -1;
    p.wq_shift = pow2 ? __builtin_ctz((unsigned)p.Wq) : -1;
  }
  p.mtiles = cdiv(M, 128);
  p.ntiles = cdiv(d->Cout, 128);
  p.cpt = d->Cin / bke;
  p.split_k = d->split_k;
  CFD_REQUIRE(p.split_k <= p.ntaps * p.cpt, "split_k=%d exceeds the %d K tiles", p.split_k, p.ntaps * p.cpt);
  const int64_t nb = (int64_t)p.mtiles * p.ntiles * p.nclass * p.split_k;
  CFD_REQUIRE(nb < (1LL << 31), "grid too large");
  p.nblocks = (int)nb;
  return CFD_OK;
}

}  // namespace cfd

using namespace cfd;

extern "C" int64_t cfd_conv_workspace_bytes(const cfd_conv_desc *d) {
  IgemmParams p;
  if (fill_params(d, p) != CFD_OK) return -1;
  if (p.split_k <= 1) return 0;
  return (int64_t)p.split_k * p.B * p.Ho * p.Wo * p.Cout * 4;
}

// kernel choice: the 256x256 ping-pong kernel (igemm256.hip) for large bf16 layers, else the 128x128 kernel.
// cfd_set_tuning(CFD_TUNE_IGEMM256, 0 | 1) overrides the heuristic (tuning / A-B runs).
// split-K choice for the 256x256 kernel: one 512-thread block per CU, so fill ~256 blocks, and keep the K loop
// of a slice long enough (>= 8 K tiles) to amortise the pipeline prologue
static int split_k_256(const IgemmParams &p) {
  const int blocks = cdiv(p.M, 256) * (p.Cout / igemm256_ntile(p)) * p.nclass;
  const int ktiles = p.ntaps * p.cpt;
  int s = 1;
  while (blocks * s * 2 <= 256 && ktiles / (s * 2) >= 8) s *= 2;
  return s;
}

static bool use_igemm256(const IgemmParams &p, int dtype) {
  if (!igemm256_eligible(p, dtype)) return false;
  const int mode = tuning(CFD_TUNE_IGEMM256);
  if (mode >= 0) return mode != 0;
  // measured (scripts/layer_bench.py, B = 16..64): the ping-pong kernel wins once a slice runs >= 32 K tiles
  // and the launch fills at least half the chip; below that the 128x128 kernel's 2 blocks/CU hide more
  // (f32 with at most half a 256-row m tile of real rows -- DeltaP fc1 at B = 128 -- is MFMA-bound on the empty half:
  //  320 us on this kernel against 230 on the 128-row one)
  //  and with two 256-row m tiles -- U e3 at batch 1, the reference's own batch size -- 82.6 us here against 57.8: the
  //  split-K slabs of 256 x 256 tiles are four times the 128 x 128 kernel's per real row)
  if (dtype == CFD_F32 && p.M <= 512) return false;
  const int blocks = cdiv(p.M, 256) * (p.Cout / igemm256_ntile(p)) * p.nclass;
  const int s = split_k_256(p);
  // (an f32 K tile is 8x the MFMA time of a bf16 one, so a much shorter slice already amortises the prologue)
  // (B = 128: with >= 4 tiles queued per CU a 16-K-tile bf16 layer -- DeltaP g4, K = 9 x 128 -- runs 0.97 PFLOP/s
  //  here against 0.70 on the 128x128 kernel)
  const int kt = (p.ntaps * p.cpt) / s;
  // (the 128(n) x 256(m) variant with >= 4 tiles queued per CU already wins at 9 K tiles: DeltaP g3, K = 9 x 64, 272 us
  //  against 311 on the 128x128 kernel at B = 128)
  const int kt_many = igemm256_ntile(p) == 128 ? 8 : 16;
  return blocks * s >= 128 && (kt >= (dtype == CFD_BF16 ? 32 : 8) || (blocks >= 1024 && kt >= kt_many));
}

// the 32-row m tile: layers with at most 64 positions per class (two m tiles), which the 128-row tile would run mostly on
// padding; cfd_set_tuning(CFD_TUNE_SKINNY, 0) turns it off (A/B)
// (f32 only: the bf16 MFMA is 16 times faster, so the padding costs little there and the deeper split's slabs more --
//  measured at batch 1, bf16: e5 21.8 -> 27.4 us, b 22.8 -> 28.5, d2 22.5 -> 26.3 on the 32-row tile)
static bool use_skinny(const IgemmParams &p, int dtype) {
  return dtype == CFD_F32 && p.M <= 64 && tuning(CFD_TUNE_SKINNY) != 0;
}

extern "C" int cfd_conv_suggest_split_k(const cfd_conv_desc *d) {
  cfd_conv_desc t = *d;
  t.split_k = 1;
  IgemmParams p;
  if (fill_params(&t, p) != CFD_OK) return 1;
  if (igemm_n64_eligible(p, d->dtype)) return 1;
  if (use_skinny(p, d->dtype)) {
    // four small workgroups per CU (1 024), at least four K tiles per slice: these layers stream their weights once
    const int ktiles = p.ntaps * p.cpt, blocks = cdiv(p.M, 32) * p.ntiles * p.nclass;
    int s = 1;
    while (blocks * s * 2 <= 1024 && ktiles / (s * 2) >= 4) s *= 2;
    return s;
  }
  if (use_igemm256(p, d->dtype)) return split_k_256(p);
  const int ktiles = p.ntaps * p.cpt;
  int s = 1;
  const int blocks = p.mtiles * p.ntiles * p.nclass;
  // fill ~2 blocks per CU (512) but keep >= 8 K tiles per slice
  while (blocks * s * 2 <= 512 && ktiles / (s * 2) >= 8) s *= 2;
  return s;
}

// fused statistics need: no split-K, every tile inside one sample, no ragged tile
static bool stats_fusable(const IgemmParams &p, int dtype) {
  if (p.split_k != 1) return false;
  if (use_igemm256(p, dtype)) return (p.Hq * p.Wq) % 256 == 0;   // Cout is a multiple of its n tile
  return (p.Hq * p.Wq) % 128 == 0 && p.Cout % 128 == 0;
}

extern "C" int cfd_conv_stats_fusable(const cfd_conv_desc *d) {
  IgemmParams p;
  if (fill_params(d, p) != CFD_OK) return 0;
  return stats_fusable(p, d->dtype) ? 1 : 0;
}

static int conv_forward_impl(const cfd_conv_desc *d, const void *x, const void *w, const float *bias, void *y,
                             void *workspace, float *stat_sum, float *stat_sq, void *stream) {
  IgemmParams p;
  int rc = fill_params(d, p);
  if (rc != CFD_OK) return rc;
  CFD_REQUIRE(x && w && y, "NULL tensor pointer");
  CFD_REQUIRE(p.split_k == 1 || workspace, "split_k=%d needs a workspace", p.split_k);
  p.x = (const char *)x; p.w = (const char *)w; p.bias = bias; p.y = (char *)y; p.ws = (float *)workspace;
  if (stat_sum || stat_sq) {
    CFD_REQUIRE(stat_sum && stat_sq, "both statistics buffers are needed");
    CFD_REQUIRE(stats_fusable(p, d->dtype), "this shape cannot fuse BatchNorm statistics (ask cfd_conv_stats_fusable)");
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t shmem = 2 * STAGE_BYTES;
  if (igemm_n64_eligible(p, d->dtype)) {
    rc = igemm_n64_launch(p, st);
    if (rc != CFD_OK) return rc;
  } else if (use_skinny(p, d->dtype) && !p.stat_sum) {
    p.mtiles = cdiv(p.M, 32);
    p.nblocks = p.mtiles * p.ntiles * p.nclass * p.split_k;
    const size_t sh = 2 * (TILE_BYTES + 32 * 128);
    hipLaunchKernelGGL((igemm_kernel<float, 32>), dim3(p.nblocks), dim3(256), sh, st, p);
    CFD_LAUNCH_CHECK();
  } else if (use_igemm256(p, d->dtype)) {
    rc = igemm256_launch(p, d->dtype, st);
    if (rc != CFD_OK) return rc;
  } else if (d->dtype == CFD_BF16) {
    if (int e_ = set_max_lds((const void *)igemm_kernel<bf16>, (int)shmem)) return e_;
    hipLaunchKernelGGL(igemm_kernel<bf16>, dim3(p.nblocks), dim3(256), shmem, st, p);
    CFD_LAUNCH_CHECK();
  } else {
    if (int e_ = set_max_lds((const void *)igemm_kernel<float>, (int)shmem)) return e_;
    hipLaunchKernelGGL(igemm_kernel<float>, dim3(p.nblocks), dim3(256), shmem, st, p);
    CFD_LAUNCH_CHECK();
  }
  if (p.split_k > 1) {
    const int64_t npix = (int64_t)p.B * p.Ho * p.Wo;
    const int64_t total = npix * p.Cout;
    const bool v4 = p.Cout % 4 == 0 && p.ldy % 4 == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0);
    const int64_t threads = v4 ? total / 4 : total;
    CFD_REQUIRE((threads + 255) / 256 < (1LL << 31), "split-K epilogue grid too large");
    if (v4 && p.split_k >= 16 && threads <= 65536)
      hipLaunchKernelGGL(splitk_epilogue_small_kernel, dim3((unsigned)((threads + 63) / 64)), dim3(256), 0, st,
                         (const float *)workspace, bias, (char *)y, npix, p.Cout, p.ldy, p.split_k, p.act, p.slope, p.out_bf16);
    else if (v4)
      hipLaunchKernelGGL(splitk_epilogue_kernel<4>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                         (const float *)workspace, bias, (char *)y, npix, p.Cout, p.ldy, p.split_k, p.act, p.slope, p.out_bf16);
    else
      hipLaunchKernelGGL(splitk_epilogue_kernel<1>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                         (const float *)workspace, bias, (char *)y, npix, p.Cout, p.ldy, p.split_k, p.act, p.slope, p.out_bf16);
    CFD_LAUNCH_CHECK();
  }
  return CFD_OK;
}

extern "C" int cfd_conv_forward(const cfd_conv_desc *d, const void *x, const void *w, const float *bias, void *y,
                                void *workspace, void *stream) {
  return conv_forward_impl(d, x, w, bias, y, workspace, nullptr, nullptr, stream);
}

extern "C" int cfd_conv_forward_stats(const cfd_conv_desc *d, const void *x, const void *w, const float *bias, void *y,
                                      void *workspace, float *stat_sum, float *stat_sq, void *stream) {
  return conv_forward_impl(d, x, w, bias, y, workspace, stat_sum, stat_sq, stream);
}

// conv (stride 1) + bias + monotone activation + MaxPooling2D(2x2): the pooled tensor is the only output
// (DeltaP/models/generator.py:27-34 at inference).  Runs on the 256-wide kernel with 2-row x 128-column m tiles.
static bool pool2_fusable(const cfd_conv_desc *d, IgemmParams &p) {
  if (fill_params(d, p) != CFD_OK) return false;
  const int es_out = d->out_dtype == CFD_BF16 ? 2 : 4;
  return (d->kind == CFD_CONV_K3S1 || d->kind == CFD_CONV_K4S1) && d->split_k == 1 && d->Cout % 256 == 0 &&
         p.Wq % 128 == 0 && p.Hq % 2 == 0 && igemm256_eligible(p, d->dtype) && d->dtype == d->out_dtype &&
         (d->act == CFD_ACT_NONE || d->act == CFD_ACT_RELU || (d->act == CFD_ACT_LEAKY && d->leaky_slope >= 0.f)) &&
         d->ldy % (16 / es_out) == 0;
}

extern "C" int cfd_conv_pool2_fusable(const cfd_conv_desc *d) {
  IgemmParams p;
  return d && pool2_fusable(d, p) ? 1 : 0;
}

// training form: y = the unpooled activation, y_pooled = MaxPooling2D(2,2) of it, one launch (patch-staged bf16 kernel)
static bool pool2_keep_fusable(const cfd_conv_desc *d, IgemmParams &p) {
  if (!pool2_fusable(d, p) || d->dtype != CFD_BF16 || tuning(CFD_TUNE_IGEMM256_PD) >= 0) return false;
  p.pool = 2;
  p.mtiles = cdiv(p.M, 256);
  return igemm256h_eligible(p, d->dtype, 256);
}

extern "C" int cfd_conv_pool2_keep_fusable(const cfd_conv_desc *d) {
  IgemmParams p;
  return d && pool2_keep_fusable(d, p) ? 1 : 0;
}

extern "C" int cfd_conv_forward_pool2_keep(const cfd_conv_desc *d, const void *x, const void *w, const float *bias, void *y,
                                           void *y_pooled, int ld_pooled, void *stream) {
  IgemmParams p;
  CFD_REQUIRE(d != nullptr, "desc is NULL");
  CFD_REQUIRE(pool2_keep_fusable(d, p),
              "conv + kept activation + 2x2 max-pool cannot be fused for this shape (ask cfd_conv_pool2_keep_fusable): bf16, "
              "3x3 stride-1 conv, split_k = 1, Cout %% 256 == 0, Cin %% 64 == 0, height %% 8 == 0, width %% 128 == 0, a "
              "monotone activation");
  CFD_REQUIRE(x && w && y && y_pooled, "NULL tensor pointer");
  CFD_REQUIRE(((uintptr_t)y & 15) == 0 && ((uintptr_t)y_pooled & 15) == 0 && ld_pooled % 8 == 0 && ld_pooled >= d->Cout,
              "y and y_pooled must be 16-byte aligned, ld_pooled a multiple of 8 and >= Cout");
  p.x = (const char *)x; p.w = (const char *)w; p.bias = bias; p.y = (char *)y; p.ws = nullptr;
  p.y2 = (char *)y_pooled; p.ldy2 = ld_pooled;
  p.pool = 2;
  return igemm256_launch(p, d->dtype, (hipStream_t)stream);
}

extern "C" int cfd_conv_forward_pool2(const cfd_conv_desc *d, const void *x, const void *w, const float *bias, void *y,
                                      void *stream) {
  IgemmParams p;
  CFD_REQUIRE(d != nullptr, "desc is NULL");
  CFD_REQUIRE(pool2_fusable(d, p),
              "conv + 2x2 max-pool cannot be fused for this shape (ask cfd_conv_pool2_fusable): needs a stride-1 conv, "
              "split_k = 1, Cout %% 256 == 0, width %% 128 == 0, even height, equal in/out dtype, a monotone activation");
  CFD_REQUIRE(x && w && y, "NULL tensor pointer");
  CFD_REQUIRE(((uintptr_t)y & 15) == 0, "y must be 16-byte aligned");
  p.x = (const char *)x; p.w = (const char *)w; p.bias = bias; p.y = (char *)y; p.ws = nullptr;
  p.pool = 1;
  return igemm256_launch(p, d->dtype, (hipStream_t)stream);
}
