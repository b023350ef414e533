// Shared between the implicit-GEMM kernels (igemm.hip: 128x128 tile, bf16 + f32; igemm256.hip: 256x256
// ping-pong tile, bf16).
#pragma once
#include "common.h"

namespace cfd {

struct IgemmParams {
  const char *x;
  const char *w;
  const float *bias;
  char *y;
  float *ws;
  int B, Hi, Wi, Cin, Cout, ldx, ldy;
  int Hq, Wq, M;       // class grid and rows per class
  int Ho, Wo;          // output grid
  int Kw;              // weight row length (all taps * Cin)
  int ntaps;           // taps per class
  int kind, act, out_bf16;
  float slope;
  int mtiles, ntiles, nclass, split_k;
  int cpt;             // K tiles per tap = Cin / BKE
  int nblocks;
  float *stat_sum, *stat_sq;   // optional fused BatchNorm statistics [B][Cout] (accumulated with atomics), or NULL
  int hw_shift, wq_shift;      // log2(Hq*Wq), log2(Wq) when both are powers of two (every reference grid), else -1
  int pool;                    // 1: the epilogue max-pools 2x2 and y is the POOLED tensor [B][Ho/2][Wo/2][Cout] (igemm256 only)
                               // 2: y is the unpooled activation as usual AND y2 (channel stride ldy2) receives the pooled
                               //    tensor from the same staged tile (igemm256h only: training keeps both)
  char *y2;
  int ldy2;
};

// first K tile of split-K slice sk (slice = [k_slice(sk), k_slice(sk + 1))): ktiles * sk < 2^31 (host: ktiles <= 16 taps *
// Cin / 32, split_k <= ktiles), so 32-bit arithmetic -- a 64-bit division is ~150 instructions in every tile's set-up;
// no division at all without split-K
__device__ __forceinline__ int k_slice(int ktiles, int sk, int split_k) {
  return split_k == 1 ? ktiles * sk : (int)((unsigned)(ktiles * sk) / (unsigned)split_k);
}

// class-grid row m -> (sample, qy, qx); shifts instead of two integer divisions when the grid sides are powers of two
// (a tile's address set-up and epilogue do this for 8 rows each)
__device__ __forceinline__ void split_pixel(const IgemmParams &p, int m, int &b, int &qy, int &qx) {
  if (p.hw_shift >= 0) {
    b = m >> p.hw_shift;
    const int rem = m & ((1 << p.hw_shift) - 1);
    qy = rem >> p.wq_shift;
    qx = rem & ((1 << p.wq_shift) - 1);
  } else {
    b = m / (p.Hq * p.Wq);
    const int rem = m - b * (p.Hq * p.Wq);
    qy = rem / p.Wq;
    qx = rem - qy * p.Wq;
  }
}

// pooled launches (IgemmParams::pool): an m tile is 2 image rows x 128 columns, so that every 2x2 pooling window lies
// inside one tile.  Row `local` (0..255) of tile `mt` -> linear pixel index.  Host: Wq % 128 == 0, Hq even.
__device__ __forceinline__ int pool_pixel(const IgemmParams &p, int mt, int local) {
  const int cb = p.Wq >> 7;                 // 128-column blocks per image row
  const int rp = mt / cb, c = mt - rp * cb; // row pair (over B * Hq / 2), column block
  return (2 * rp + (local >> 7)) * p.Wq + c * 128 + (local & 127);
}

// tap t of a class -> input-pixel offset (dy,dx) relative to the class-grid pixel * stride, and the
// weight tap index; uniform (scalar) arithmetic
__host__ __device__ __forceinline__ void tap_geom(int kind, int ph, int pw, int t, int &dy, int &dx, int &wtap) {
  if (kind == CFD_CONVT_K4S2) {
    const int kh = (1 - ph) + 2 * (t >> 1), kw = (1 - pw) + 2 * (t & 1);
    dy = (ph + 1 - kh) / 2;
    dx = (pw + 1 - kw) / 2;
    wtap = kh * 4 + kw;
  } else if (kind == CFD_CONV_K3S1) {
    const int kh = t / 3, kw = t - kh * 3;
    dy = kh - 1; dx = kw - 1; wtap = t;
  } else if (kind == CFD_DENSE) {
    dy = 0; dx = 0; wtap = 0;
  } else if (kind == CFD_CONV_K4S1_BWD) {
    dy = (t >> 2) - 2; dx = (t & 3) - 2; wtap = t;   // pad 2 before / 1 after: data gradient of the k4 s1 'same' conv
  } else {
    dy = (t >> 2) - 1; dx = (t & 3) - 1; wtap = t;
  }
}

// sum over the 16 lanes of a DPP row (lanes that share lane>>4); every lane of the row gets the total
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

__device__ __forceinline__ void glds16(const char *gsrc, char *lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                   (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// igemm256.hip
bool igemm256_eligible(const IgemmParams &p, int dtype);
int igemm256_launch(const IgemmParams &p, int dtype, hipStream_t st);
int igemm256_block_count(const IgemmParams &p);
int igemm256_ntile(const IgemmParams &p);
// igemm256.hip: Cout = 64 (mod 128) on the co-resident patch kernel's 64-wide n tile
bool igemm_n64_eligible(const IgemmParams &p, int dtype);
int igemm_n64_launch(const IgemmParams &p, hipStream_t st);
// igemm256h.hip: does the patch-staged kernel take this launch (p prepared as by igemm256_launch)?
bool igemm256h_eligible(const IgemmParams &p, int dtype, int ntile);

}  // namespace cfd
