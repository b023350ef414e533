// Shared between the 256-wide ping-pong kernels (igemm256.hip: one LDS tile per tap; igemm256h.hip: the activation
// operand staged once per channel chunk as an input patch).
#pragma once
#include "common.h"
#include "igemm.h"

namespace cfd {

constexpr int HT_BYTES = 128 * 128;   // half tile
constexpr int ST_BYTES = 4 * HT_BYTES;
constexpr unsigned OOB = 0x80000000u;  // voffset beyond any num_records we accept (< 2^31): reads as zero

struct Igemm256Params {
  IgemmParams g;
  unsigned xbytes, wbytes;   // buffer sizes for the range check
  int tapinfo[4][16];        // per class, per tap: (dy*Wi + dx) pixel offset << 8 | weight tap index
  short tapdy[4][16], tapdx[4][16];
  int cls_fast;              // block order: 1 = the sub-pixel classes of one pixel tile are adjacent blocks
  int tgeo[4][8];            // per class: kw, kh, dy0, dys, dx0, dxs -- the taps form a kh x kw grid, tap t = ky*kw + kx
                             // sits at (dy0 + dys*ky, dx0 + dxs*kx)
  int stage_out;             // bf16 output tile goes through LDS and leaves as whole 512-B pixel rows
  unsigned dv[3][3];         // fastdiv constants of ntiles, nclass, mtiles (block index -> tile)
};

// LDS-staged output (bf16, no split-K): measured with the stamp build, the direct epilogue -- 32 stores per lane, each
// wave instruction scattering 16 segments of 32 B -- took 7.6 K tiles' worth per tile (~11 us); staged, a wave
// instruction stores two whole 512-B pixel rows.
constexpr int OUT_ROW = 528;               // 512 B of channels + 16 B pad (bank spread of the fragment writes)
constexpr int OUT_OFF = 8192;              // after the statistics partials
constexpr int OUT_LDS = OUT_OFF + 256 * OUT_ROW;
constexpr int BIAS256_OFF = OUT_LDS;       // igemm256_kernel: the n tile's bias (1 KiB) sits after everything else

__device__ __forceinline__ void bload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char *lds_dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)lds_dst, 16, (int)voff,
                                           (int)soff, 0, 0);
}

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(f32x4 a, f32x4 b, f32x4 c) {
#pragma unroll
  for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
  return c;
}

// q = n / d for any 32-bit n with the host-made constants of d (Granlund-Montgomery): three scalar instructions instead
// of the ~25 of a run-time division -- a tile's set-up does five of them before it can issue its first load
__device__ __forceinline__ int fastdiv(int n, const unsigned (&c)[3]) {
  const unsigned t = __umulhi((unsigned)n, c[0]);
  return (int)((t + (((unsigned)n - t) >> c[1])) >> c[2]);
}
static void fastdiv_make(unsigned d, unsigned (&c)[3]) {
  unsigned L = 0;
  while ((1ull << L) < d) ++L;
  c[0] = (unsigned)((((1ull << 32) * ((1ull << L) - d)) / d) + 1);
  c[1] = L < 1 ? L : 1;
  c[2] = L < 1 ? 0 : L - 1;
}

// igemm256h.hip: patch-staged variant (bf16, 3x3 stride 1 and the transposed 4x4 stride 2, no split-K).  `p` as prepared
// by igemm256_launch (mtiles / ntiles / nblocks for 256-row m tiles and `ntile`-wide n tiles).
bool igemm256h_eligible(const IgemmParams &p, int dtype, int ntile);
// coresident: the 128(n) x 256(m) kernel with 32-channel K tiles, two workgroups per CU (ntile must be 128)
int igemm256h_launch(const IgemmParams &p, int ntile, unsigned xbytes, unsigned wbytes, int cls_fast, int stage_out,
                     hipStream_t st, bool coresident = false);

}  // namespace cfd
