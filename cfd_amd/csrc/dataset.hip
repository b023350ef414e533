// Dataset build on the GPU: processed-CSV columns -> model input (signed distance field) and DeltaP.
//
// Replaces the per-case arithmetic of utils.load_data (utils.py:64-107):
//   pts = -1 where p > 0 else +1                         (:89-93, the 32,768-iteration Python loop)
//   DeltaP = mean(p[:, 0]) - mean(p[:, -1])              (:97)
//   sdf = skfmm.distance(pts, dx = 4/H)                  (:100-103)  -> the model input
// skfmm.distance is scikit-fmm's fast-marching signed distance (third party, unpinned in the reference, absent from
// the build image: PARITY UNPINNED).  This file implements its published scheme -- order 2, no mask:
//   1. frozen band: a cell with a 4-neighbour of opposite sign gets, per axis, the linearly interpolated distance to
//      the zero crossing d_axis = dx * phi_i / (phi_i - phi_n) (the smaller of the two directions) and the axes are
//      combined harmonically, d = sqrt(1 / sum 1/d_axis^2), signed like phi_i  (dx/2 at a straight wall, dx/(2 sqrt 2)
//      in a corner);
//   2. every Far neighbour of the band becomes Narrow with a tentative distance and enters a min-heap keyed by |d|;
//   3. pop the smallest, freeze it, re-evaluate its non-frozen 4-neighbours and the Narrow cell behind a frozen one;
//   4. tentative distance: per axis the frozen neighbour nearest the contour (value1) and, if frozen and not further
//      out, the cell behind it (value2): second-order term tp = (4 value1 - value2)/3 with weight 9/4, else the
//      first-order term; root of a d^2 + b d + c - 1 = 0 on the side of phi's sign, or the parabola's minimum -b/(2a)
//      when the discriminant is negative (speckled masks reach that branch).
// The CPU oracle (oracle/sdf_fmm.c) is an independent implementation of the same scheme; both pop the heap in the total
// order (|d|, cell address) and evaluate every expression in scikit-fmm's operand order without contraction
// (-ffp-contract=off), so the two agree bit for bit.
//
// Fast marching is sequential inside a case and the cases are independent, so the parallel axis is the case: ONE THREAD
// marches one case (heap, flags and heap positions in the caller's workspace, 17 B per cell).  Up to 2048 cases run one
// per wave (lowest latency: a lone case takes ~0.1 s, which is what a wave's single lane can do); beyond that the
// lanes of a wave take different cases, so that each wave instruction has up to 64 independent memory accesses in
// flight -- 10 000 cases keep every CU's memory pipeline busy.  HBM bytes per case: 8 B read + 8 B written per cell;
// the heap traffic (~17 B x log n per cell) stays in L2.
#include "common.h"
#include <float.h>

namespace cfd {

enum : unsigned char { FMM_FAR = 0, FMM_NARROW = 1, FMM_FROZEN = 2 };

struct Marcher {
  int H, W;
  const double *p;          // p_convert grid: phi = p > 0 ? -1 : +1
  double *dist;             // output
  unsigned char *flag;
  double *hkey;             // heap: key and cell at each position, position of each cell
  int *hcell, *hpos;
  int hn;
  double dx, idx2;

  __device__ __forceinline__ double phi(int i) const { return p[i] > 0.0 ? -1.0 : 1.0; }

  // neighbour along dim (0 rows, 1 columns) at offset step, or -1
  __device__ __forceinline__ int nb(int i, int dim, int step) const {
    const int r = i / W, c = i - r * W;
    if (dim == 0) { const int rr = r + step; return (rr < 0 || rr >= H) ? -1 : rr * W + c; }
    const int cc = c + step;
    return (cc < 0 || cc >= W) ? -1 : r * W + cc;
  }

  // ---- min-heap on (key, cell)
  __device__ __forceinline__ bool before(double ka, int ca, double kb, int cb) const {
    return ka < kb || (ka == kb && ca < cb);
  }
  __device__ void place(int pos, int cell, double key) { hkey[pos] = key; hcell[pos] = cell; hpos[cell] = pos; }
  __device__ void sift_up(int pos, int cell, double key) {
    while (pos > 0) {
      const int par = (pos - 1) >> 1;
      const double kp = hkey[par];
      const int cp = hcell[par];
      if (!before(key, cell, kp, cp)) break;
      place(pos, cp, kp);
      pos = par;
    }
    place(pos, cell, key);
  }
  __device__ void sift_down(int pos, int cell, double key) {
    for (;;) {
      int ch = 2 * pos + 1;
      if (ch >= hn) break;
      double kc = hkey[ch];
      int cc = hcell[ch];
      if (ch + 1 < hn) {
        const double k2 = hkey[ch + 1];
        const int c2 = hcell[ch + 1];
        if (before(k2, c2, kc, cc)) { ++ch; kc = k2; cc = c2; }
      }
      if (!before(kc, cc, key, cell)) break;
      place(pos, cc, kc);
      pos = ch;
    }
    place(pos, cell, key);
  }
  __device__ void push(int cell, double key) { sift_up(hn++, cell, key); }
  __device__ void rekey(int cell, double key) {
    const int pos = hpos[cell];
    const double old = hkey[pos];
    if (before(key, cell, old, cell)) sift_up(pos, cell, key);
    else sift_down(pos, cell, key);
  }
  __device__ int pop() {
    const int top = hcell[0];
    --hn;
    if (hn > 0) sift_down(0, hcell[hn], hkey[hn]);
    return top;
  }

  // tentative distance of cell i from its frozen neighbours (order 2)
  __device__ double tentative(int i) {
    double a = 0.0, b = 0.0, c = 0.0;
#pragma unroll
    for (int dim = 0; dim < 2; ++dim) {
      double v1 = DBL_MAX, v2 = DBL_MAX;
#pragma unroll
      for (int j = -1; j <= 1; j += 2) {
        const int n1 = nb(i, dim, j);
        if (n1 < 0 || flag[n1] != FMM_FROZEN) continue;
        const double d1 = dist[n1];
        if (fabs(d1) < fabs(v1)) {
          v1 = d1;
          const int n2 = nb(i, dim, 2 * j);
          if (n2 >= 0 && flag[n2] == FMM_FROZEN) {
            const double d2 = dist[n2];
            if ((d2 <= v1 && v1 >= 0.0) || (d2 >= v1 && v1 <= 0.0)) v2 = d2;   // (no reset otherwise: as published)
          }
        }
      }
      if (v2 < DBL_MAX) {
        const double tp = (1.0 / 3.0) * (4.0 * v1 - v2);
        a += idx2 * (9.0 / 4.0);
        b -= idx2 * 2.0 * (9.0 / 4.0) * tp;
        c += idx2 * (9.0 / 4.0) * (tp * tp);
      } else if (v1 < DBL_MAX) {
        a += idx2;
        b -= idx2 * 2.0 * v1;
        c += idx2 * (v1 * v1);
      }
    }
    c -= 1.0;
    const double det = b * b - 4.0 * a * c;
    if (!(det >= 0.0)) return -b / (2.0 * a);     // scikit-fmm catches its own "negative discriminant" here: the parabola's minimum
    return phi(i) > DBL_EPSILON ? (-b + sqrt(det)) / 2.0 / a : (-b - sqrt(det)) / 2.0 / a;
  }

  __device__ int march() {     // -> bit 0: some fluid cell, bit 1: some solid cell
    const int n = H * W;
    int seen = 0;
    // frozen band
    for (int i = 0; i < n; ++i) {
      const double pi = phi(i);
      seen |= pi < 0.0 ? 1 : 2;
      double ld[2];
      bool borders = false;
#pragma unroll
      for (int dim = 0; dim < 2; ++dim) {
        ld[dim] = 0.0;
#pragma unroll
        for (int j = -1; j <= 1; j += 2) {
          const int na = nb(i, dim, j);
          if (na < 0) continue;
          const double pn = phi(na);
          if (pi * pn < 0.0) {
            borders = true;
            const double d = dx * pi / (pi - pn);
            if (ld[dim] == 0.0 || ld[dim] > d) ld[dim] = d;
          }
        }
      }
      double v = 0.0;
      unsigned char f = FMM_FAR;
      if (borders) {
        double dsum = 0.0;
        if (ld[0] > 0.0) dsum += 1.0 / ld[0] / ld[0];
        if (ld[1] > 0.0) dsum += 1.0 / ld[1] / ld[1];
        v = pi < 0.0 ? -sqrt(1.0 / dsum) : sqrt(1.0 / dsum);
        f = FMM_FROZEN;
      }
      dist[i] = v;
      flag[i] = f;
    }
    if (seen != 3) return seen;            // no zero contour: skfmm raises
    // narrow band around it, in address order
    hn = 0;
    for (int i = 0; i < n; ++i) {
      if (flag[i] != FMM_FROZEN) continue;
      for (int dim = 0; dim < 2; ++dim)
        for (int j = -1; j <= 1; j += 2) {
          const int na = nb(i, dim, j);
          if (na >= 0 && flag[na] == FMM_FAR) {
            flag[na] = FMM_NARROW;
            const double d = tentative(na);
            dist[na] = d;
            push(na, fabs(d));
          }
        }
    }
    // march
    while (hn > 0) {
      const int addr = pop();
      flag[addr] = FMM_FROZEN;
      for (int dim = 0; dim < 2; ++dim)
        for (int j = -1; j <= 1; j += 2) {
          const int na = nb(addr, dim, j);
          if (na < 0) continue;
          const unsigned char fa = flag[na];
          if (fa != FMM_FROZEN) {
            const double d = tentative(na);
            if (d != 0.0) {
              dist[na] = d;
              if (fa == FMM_NARROW) rekey(na, fabs(d));
              else { flag[na] = FMM_NARROW; push(na, fabs(d)); }
            }
          } else {                          // second order: the narrow cell behind a frozen neighbour
            const int n2 = nb(addr, dim, 2 * j);
            if (n2 >= 0 && flag[n2] == FMM_NARROW) {
              const double d = tentative(n2);
              if (d != 0.0) { rekey(n2, fabs(d)); dist[n2] = d; }
            }
          }
        }
    }
    return seen;
  }
};

static inline int64_t fmm_ws_per_case(int H, int W) {
  const int64_t n = (int64_t)H * W;
  return ((n * 8 + 15) & ~15LL) + 2 * ((n * 4 + 15) & ~15LL) + ((n + 15) & ~15LL);
}

// thread `lane < per_wave` of wave w marches case w * per_wave + lane
__global__ __launch_bounds__(64) void sdf_fmm_kernel(const double *__restrict__ p, int ncase, int H, int W, double cell,
                                                    double *__restrict__ sdf, int *__restrict__ flags,
                                                    unsigned char *__restrict__ ws, int64_t ws_per_case, int per_wave) {
  const int lane = threadIdx.x;
  if (lane >= per_wave) return;
  const int64_t cs = (int64_t)blockIdx.x * per_wave + lane;
  if (cs >= ncase) return;
  const int64_t n = (int64_t)H * W;
  unsigned char *base = ws + cs * ws_per_case;
  Marcher m;
  m.H = H; m.W = W;
  m.p = p + cs * n;
  m.dist = sdf + cs * n;
  m.hkey = (double *)base;
  m.hcell = (int *)(base + ((n * 8 + 15) & ~15LL));
  m.hpos = m.hcell + ((n * 4 + 15) & ~15LL) / 4;
  m.flag = (unsigned char *)(m.hpos + ((n * 4 + 15) & ~15LL) / 4);
  m.hn = 0;
  m.dx = cell;
  m.idx2 = 1.0 / cell / cell;
  flags[cs] = m.march();
}

// numpy's pairwise summation of n <= 128 strided doubles: 8 running sums, then a fixed tree (n % 8 tail last)
__device__ double np_pairwise_sum(const double *a, int n, size_t stride) {
  if (n < 8) {
    double res = 0.;
    for (int i = 0; i < n; ++i) res += a[i * stride];
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[(i + j) * stride];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i * stride];
  return res;
}
__device__ double np_sum(const double *a, int n, size_t stride) {   // numpy splits blocks larger than 128 in halves
  if (n <= 128) return np_pairwise_sum(a, n, stride);
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_sum(a, n2, stride) + np_sum(a + n2 * stride, n - n2, stride);
}

__global__ void delta_p_kernel(const double *__restrict__ p, int ncase, int H, int W, double *__restrict__ out) {
  const int cs = blockIdx.x * blockDim.x + threadIdx.x;
  if (cs >= ncase) return;
  const double *pc = p + (size_t)cs * H * W;
  const double m0 = np_sum(pc, H, (size_t)W) / (double)H;
  const double m1 = np_sum(pc + (W - 1), H, (size_t)W) / (double)H;
  out[cs] = m0 - m1;
}

}  // namespace cfd

using namespace cfd;

extern "C" int64_t cfd_dataset_workspace_bytes(int ncase, int H, int W) {
  if (ncase <= 0 || H <= 0 || W <= 0) return -1;
  return (int64_t)ncase * fmm_ws_per_case(H, W);
}

extern "C" int cfd_dataset_grids(const double *p, int ncase, int H, int W, double cell, double *sdf, double *delta_p,
                                 int *flags, void *workspace, void *stream) {
  CFD_REQUIRE(p && sdf && delta_p && flags && workspace, "NULL pointer");
  CFD_REQUIRE(ncase > 0 && H > 0 && W > 0 && (int64_t)H * W < (1LL << 28), "bad shape");
  CFD_REQUIRE(cell > 0.0, "cell size must be positive");
  hipStream_t st = (hipStream_t)stream;
  // one case per wave up to 2048 cases, then several lanes of a wave take a case each (see the header)
  int per_wave = (ncase + 2047) / 2048;
  per_wave = per_wave > 64 ? 64 : per_wave;
  hipLaunchKernelGGL(sdf_fmm_kernel, dim3(cdiv(ncase, per_wave)), dim3(64), 0, st, p, ncase, H, W, cell, sdf, flags,
                     (unsigned char *)workspace, fmm_ws_per_case(H, W), per_wave);
  CFD_LAUNCH_CHECK();
  hipLaunchKernelGGL(delta_p_kernel, dim3((ncase + 63) / 64), dim3(64), 0, st, p, ncase, H, W, delta_p);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}
