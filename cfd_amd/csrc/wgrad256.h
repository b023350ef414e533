// Interface of wgrad256.hip (256x256 ping-pong bf16 weight-gradient kernel) used by the dispatcher in wgrad.hip.
#pragma once
#include "common.h"

namespace cfd {

struct Wgrad256Params {
  const char *plain, *gath;   // operand base pointers (gath already moved back by `margin`)
  float *dw;
  float *slab;                // split-M partial tiles [split_m][Cout][ntaps][Cin], or NULL (atomics)
  int64_t dw_elems;
  int ldp, ldg;               // channel strides (elements)
  int Hg, Wg;                 // gathered operand's grid
  int lw, lh;                 // log2 of the reduction grid sides Wm, Hm
  int S, oy_base, ox_base;    // gathered pixel = grid pixel * S + (kh + oy_base, kw + ox_base)
  int ktaps_w;                // kernel width in taps (4, 3 or 1)
  int Cin, Cout, ntaps, M;
  int cotiles, citiles, split_m, nblocks;
  unsigned pbytes, gbytes, margin;
  int gather_a;               // 1: the co-side operand (dY) is the gathered one (transposed conv)
  int pair, tap_groups;       // pair = 1: a tile's 256 ci columns are two adjacent taps x 128 channels; blocks per (co, ci) tile
};

bool wgrad256_plan(const cfd_conv_desc *d, Wgrad256Params &p);
bool wgrad256_selected(const cfd_conv_desc *d, Wgrad256Params &p);
int wgrad256_suggest_split(const Wgrad256Params &p);
int64_t wgrad256_workspace_bytes(const Wgrad256Params &p, int split_m);
// dw[i] = sum over s of slab[s][i], i < 4 * n4, in a fixed order (shared with wgrad.hip's 128 x 128 kernel)
int wgrad_slab_reduce_launch(const float *slab, float *dw, int64_t n4, int split, hipStream_t st);
int wgrad256_launch(Wgrad256Params &p, const void *x, const void *dy, float *dw, int split_m, float *workspace,
                    hipStream_t st);

}  // namespace cfd
