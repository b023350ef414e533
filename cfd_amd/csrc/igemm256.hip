// bf16 implicit-GEMM convolution, 256(n) x 256(m) tile, 8 waves in two ping-pong groups.
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
//     while the other is in its memory cluster (the two waves of a SIMD are w and w+4).
//   * LDS-DMA stays in flight across barriers: counted s_waitcnt vmcnt(2*(PD-2)) + raw s_barrier; a half
//     tile is read one phase after the wait that retires it and restaged >= 2 phases after its last read.
//     Past the end of the K range the stream issues zero-page loads, so the counts stay uniform.
#include "common.h"
#include "igemm.h"

namespace cfd {

__device__ __attribute__((aligned(16))) const unsigned int g_zero_page256[8] = {0, 0, 0, 0, 0, 0, 0, 0};

constexpr int HT_BYTES = 128 * 128;   // half tile
constexpr int ST_BYTES = 4 * HT_BYTES;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else static_assert(N < 0, "unsupported vmcnt");
}

template <int PD>
__global__ __launch_bounds__(512, 2) void igemm256_kernel(IgemmParams p) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- XCD-aware block remap (bijective)
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
  const int m0 = mt * 256, n0 = nt * 256;
  const int S = (p.kind == CFD_CONV_K4S2) ? 2 : 1;

  const int ktiles = p.ntaps * p.cpt;
  const int kbeg = (int)((int64_t)ktiles * sk / p.split_k);
  const int kend = (int)((int64_t)ktiles * (sk + 1) / p.split_k);
  const char *zero = (const char *)g_zero_page256;

  // ---- staging geometry: a half tile = 16 wave-instructions of 1 KiB (8 rows x 128 B); wave w issues
  // pieces w and 8 + w.  Lane -> row lane/8 of its piece, physical chunk lane%8, which holds the
  // LOGICAL chunk (lane%8) ^ (lane/8): the swizzle is applied on the source address (LDS image stays lane-linear).
  const int lrow = lane >> 3;
  const int chunk = (lane & 7) ^ lrow;
  // A side: one per-lane pointer; halves / pieces differ by uniform multiples of Kw
  const char *wbase = p.w + ((int64_t)(n0 + wave * 8 + lrow) * p.Kw + chunk * 8) * 2;
  const int64_t wstep64 = (int64_t)64 * p.Kw * 2;   // 64 rows further
  // B side: four pixel rows per lane, r = half*2 + piece
  const char *xrow[4];
  unsigned vmask[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + (r >> 1) * 128 + (r & 1) * 64 + wave * 8 + lrow;
    const bool mv = m < p.M;
    const int mm = mv ? m : 0;
    const int b = mm / (p.Hq * p.Wq);
    const int rem = mm - b * (p.Hq * p.Wq);
    const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
    const int y0 = qy * S, x0 = qx * S;
    xrow[r] = p.x + ((int64_t)((b * p.Hi + y0) * p.Wi + x0) * p.ldx + chunk * 8) * 2;
    unsigned msk = 0;
#pragma clang loop vectorize(disable) unroll(disable)
    for (int t = 0; t < p.ntaps; ++t) {
      int dy, dx, wt;
      tap_geom(p.kind, ph, pw, t, dy, dx, wt);
      const bool ok = mv && (unsigned)(y0 + dy) < (unsigned)p.Hi && (unsigned)(x0 + dx) < (unsigned)p.Wi;
      msk |= (ok ? 1u : 0u) << t;
    }
    vmask[r] = msk;
  }

  // ---- issue stream state (uniform): the K tile whose half tiles are being issued
  int iss_kt = kbeg, iss_t = kbeg / p.cpt, iss_c = kbeg - iss_t * p.cpt;
  int64_t iss_woff = 0, iss_xoff = 0;
  bool iss_ok = false;
  auto iss_update = [&]() {
    iss_ok = iss_kt < kend;
    int dy, dx, wt;
    tap_geom(p.kind, ph, pw, iss_ok ? iss_t : 0, dy, dx, wt);
    const int ci0 = iss_c * 64;
    iss_woff = ((int64_t)wt * p.Cin + ci0) * 2;
    iss_xoff = ((int64_t)(dy * p.Wi + dx) * p.ldx + ci0) * 2;
  };
  auto iss_advance = [&]() {
    ++iss_kt;
    if (++iss_c == p.cpt) { iss_c = 0; ++iss_t; }
    iss_update();
  };
  iss_update();

  auto issue = [&](int slot, int stage) {   // slot is a compile-time constant at every call site
    char *dst = lds + stage * ST_BYTES + slot * HT_BYTES + wave * 1024;
    if (slot == 0 || slot == 3) {
      const int64_t hoff = (slot == 3 ? 2 * wstep64 : 0) + iss_woff;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char *src = iss_ok ? wbase + hoff + i * wstep64 : zero;
        glds16(src, dst + i * 8192);
      }
    } else {
      const int h = slot - 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = h * 2 + i;
        const bool ok = iss_ok && ((vmask[r] >> iss_t) & 1u);
        const char *src = ok ? xrow[r] + iss_xoff : zero;
        glds16(src, dst + i * 8192);
      }
    }
  };

  // ---- fragment read offsets (bytes inside a half tile); k-step 1 is the same address ^ 64
  const int fr = lane & 15, fq = lane >> 4;
  const int laneoff = fr * 128 + ((fq ^ (fr & 7)) << 4);
  const int aoff = wr * 64 * 128 + laneoff;   // + i*2048
  const int boff = wc * 32 * 128 + laneoff;   // + j*2048

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: H0 .. H(PD-1)
#pragma unroll
  for (int h = 0; h < PD; ++h) {
    issue(h & 3, (h >> 2) & 1);
    if ((h & 3) == 3) iss_advance();
  }
  wait_vmcnt<2 * (PD - 2)>();   // H0, H1 landed
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();   // second group runs one barrier behind

  bf16x8 a[4][2], b0[2][2], b1[2][2];
  const int nkt = kend - kbeg;
  for (int k = 0; k < nkt; ++k) {
    const char *sb = lds + (k & 1) * ST_BYTES;

#define CFD_PHASE(P, READS, MFMAS)                                                     \
    {                                                                                  \
      READS                                                                            \
      issue((P + PD) & 3, (k + ((P + PD) >> 2)) & 1);                                  \
      if (((P + PD) & 3) == 3) iss_advance();                                          \
      wait_vmcnt<2 * (PD - 2)>();                                                      \
      __builtin_amdgcn_s_barrier();                                                    \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_setprio(1);                                                   \
      MFMAS                                                                            \
      __builtin_amdgcn_s_setprio(0);                                                   \
      __builtin_amdgcn_sched_barrier(0);                                               \
      __builtin_amdgcn_s_barrier();                                                    \
    }

    // phase 0: A0 x B0
    CFD_PHASE(0,
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {
        _Pragma("unroll") for (int j = 0; j < 2; ++j)
          b0[j][s] = *(const bf16x8 *)(sb + 1 * HT_BYTES + j * 2048 + (boff ^ (s << 6)));
        _Pragma("unroll") for (int i = 0; i < 4; ++i)
          a[i][s] = *(const bf16x8 *)(sb + 0 * HT_BYTES + i * 2048 + (aoff ^ (s << 6)));
      },
      _Pragma("unroll") for (int s = 0; s < 2; ++s)
        _Pragma("unroll") for (int i = 0; i < 4; ++i)
          _Pragma("unroll") for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][s], b0[j][s], acc[i][j], 0, 0, 0);
    )
    // phase 1: A0 x B1
    CFD_PHASE(1,
      _Pragma("unroll") for (int s = 0; s < 2; ++s)
        _Pragma("unroll") for (int j = 0; j < 2; ++j)
          b1[j][s] = *(const bf16x8 *)(sb + 2 * HT_BYTES + j * 2048 + (boff ^ (s << 6)));
      ,
      _Pragma("unroll") for (int s = 0; s < 2; ++s)
        _Pragma("unroll") for (int i = 0; i < 4; ++i)
          _Pragma("unroll") for (int j = 0; j < 2; ++j)
            acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][s], b1[j][s], acc[i][2 + j], 0, 0, 0);
    )
    // phase 2: A1 x B1
    CFD_PHASE(2,
      _Pragma("unroll") for (int s = 0; s < 2; ++s)
        _Pragma("unroll") for (int i = 0; i < 4; ++i)
          a[i][s] = *(const bf16x8 *)(sb + 3 * HT_BYTES + i * 2048 + (aoff ^ (s << 6)));
      ,
      _Pragma("unroll") for (int s = 0; s < 2; ++s)
        _Pragma("unroll") for (int i = 0; i < 4; ++i)
          _Pragma("unroll") for (int j = 0; j < 2; ++j)
            acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][s], b1[j][s], acc[4 + i][2 + j], 0, 0, 0);
    )
    // phase 3: A1 x B0 (operands already in registers)
    CFD_PHASE(3,
      ,
      _Pragma("unroll") for (int s = 0; s < 2; ++s)
        _Pragma("unroll") for (int i = 0; i < 4; ++i)
          _Pragma("unroll") for (int j = 0; j < 2; ++j)
            acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][s], b0[j][s], acc[4 + i][j], 0, 0, 0);
    )
#undef CFD_PHASE
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  wait_vmcnt<0>();   // the zero-page tail loads must not outlive the workgroup's LDS allocation

  // ---- epilogue: lane holds channels n..n+3 (register index) of pixel m
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + (j >> 1) * 128 + wc * 32 + (j & 1) * 16 + fr;
    if (m >= p.M) continue;
    const int b = m / (p.Hq * p.Wq);
    const int rem = m - b * (p.Hq * p.Wq);
    const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
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
      for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act, p.slope);
      if (p.out_bf16) {
        bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *(bf16x4 *)(p.y + (opix * p.ldy + n) * 2) = o;
      } else {
        *(f32x4 *)(p.y + (opix * p.ldy + n) * 4) = v;
      }
    }
  }
}

bool igemm256_eligible(const IgemmParams &p, int dtype) {
  return dtype == CFD_BF16 && p.Cout % 256 == 0 && p.Cin % 64 == 0 && p.ntaps <= 16;
}

int igemm256_block_count(const IgemmParams &p) {
  return cdiv(p.M, 256) * (p.Cout / 256) * p.nclass * p.split_k;
}

int igemm256_launch(const IgemmParams &pin, hipStream_t st) {
  IgemmParams p = pin;
  p.mtiles = cdiv(p.M, 256);
  p.ntiles = p.Cout / 256;
  p.nblocks = p.mtiles * p.ntiles * p.nclass * p.split_k;
  const size_t shmem = 2 * ST_BYTES;
  static bool attr_done = false;
  if (!attr_done) {
    CFD_HIP(hipFuncSetAttribute((const void *)igemm256_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    CFD_HIP(hipFuncSetAttribute((const void *)igemm256_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    attr_done = true;
  }
  if (tuning(CFD_TUNE_IGEMM256_PD) == 6)
    hipLaunchKernelGGL(igemm256_kernel<6>, dim3(p.nblocks), dim3(512), shmem, st, p);
  else
    hipLaunchKernelGGL(igemm256_kernel<5>, dim3(p.nblocks), dim3(512), shmem, st, p);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

}  // namespace cfd
