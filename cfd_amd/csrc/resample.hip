// Scattered-point -> pixel-grid nearest-neighbour resample with wall mask, fp64, bit-exact.
//
// Replaces process_csv_with_pixel_coordinates_nearest, pixels_coords_testing.py:151-227:
//   :182      P[P<0] = 0
//   :188-205  y_min/y_max of the points within 1e-5 (else 1e-4) of each of 101 z slices
//   :208-211  np.interp of the valid slices to every pixel column
//   :214-215  scipy griddata(method='nearest') = cKDTree.query: argmin of (dy*dy + dz*dz), fp64
//   :218-221  zero outside [y_min, y_max]
//   :225      p_convert = p * 1072.033
// This translation unit is compiled with -ffp-contract=off: every product and sum rounds once, as
// in numpy/scipy's x86-64 builds, so the argmin and the interpolated wall are the same doubles.
// Equidistant source points resolve to the LOWEST index (scipy's choice is traversal dependent).
//
// Layout: points stay AoS [n][5] in HBM (40 B rows); the nearest kernel stages (y,z) pairs of a tile
// of points into LDS once per block and every thread (one pixel each) scans them with broadcast
// reads; the 5 output columns of a pixel are written as one 40-B row, rows contiguous across lanes.
#include "common.h"

namespace cfd {

struct ResampleWs {          // per case, in the caller's workspace
  // [S] ymin, [S] ymax, [W] ylo, [W] yhi, then 1 int status
};

__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000LL); }

// one block per (slice, case): min/max of y over points with |z - zs| < tol1, else tol2
__global__ __launch_bounds__(256) void wall_slices_kernel(const double *__restrict__ pts,
                                                         const int64_t *__restrict__ offs,
                                                         const double *__restrict__ zs, int S, double tol1,
                                                         double tol2, double *__restrict__ ws, int64_t ws_stride) {
  const int s = blockIdx.x, cs = blockIdx.y;
  const int64_t beg = offs[cs], end = offs[cs + 1];
  const double zv = zs[s];
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  double mn1 = inf, mx1 = -inf, mn2 = inf, mx2 = -inf;
  int c1 = 0, c2 = 0;
  for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
    const double y = pts[i * 5 + 1], z = pts[i * 5 + 2];
    const double d = fabs(z - zv);
    if (d < tol1) { mn1 = fmin(mn1, y); mx1 = fmax(mx1, y); ++c1; }
    if (d < tol2) { mn2 = fmin(mn2, y); mx2 = fmax(mx2, y); ++c2; }
  }
  __shared__ double r[4][256];
  __shared__ int rc[2][256];
  r[0][threadIdx.x] = mn1; r[1][threadIdx.x] = mx1; r[2][threadIdx.x] = mn2; r[3][threadIdx.x] = mx2;
  rc[0][threadIdx.x] = c1; rc[1][threadIdx.x] = c2;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      r[0][threadIdx.x] = fmin(r[0][threadIdx.x], r[0][threadIdx.x + w]);
      r[1][threadIdx.x] = fmax(r[1][threadIdx.x], r[1][threadIdx.x + w]);
      r[2][threadIdx.x] = fmin(r[2][threadIdx.x], r[2][threadIdx.x + w]);
      r[3][threadIdx.x] = fmax(r[3][threadIdx.x], r[3][threadIdx.x + w]);
      rc[0][threadIdx.x] += rc[0][threadIdx.x + w];
      rc[1][threadIdx.x] += rc[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double *w = ws + cs * ws_stride;
    double lo, hi;
    if (rc[0][0] > 0) { lo = r[0][0]; hi = r[1][0]; }
    else if (rc[1][0] > 0) { lo = r[2][0]; hi = r[3][0]; }
    else { lo = dnan(); hi = dnan(); }
    w[s] = lo;
    w[S + s] = hi;
  }
}

// np.interp(x, xp, fp) for one x (numpy/_core/src/multiarray/compiled_base.c, arr_interp)
__device__ double np_interp(double x, const double *xp, const double *fp, int n) {
  if (n == 1) return fp[0];
  if (x > xp[n - 1]) return fp[n - 1];
  if (x < xp[0]) return fp[0];
  int lo = 0, hi = n;  // find j with xp[j] <= x < xp[j+1]
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (x >= xp[mid]) lo = mid + 1; else hi = mid;
  }
  const int j = lo - 1;
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double res = slope * (x - xp[j]) + fp[j];
  if (res != res) {
    res = slope * (x - xp[j + 1]) + fp[j + 1];
    if (res != res && fp[j] == fp[j + 1]) res = fp[j];
  }
  return res;
}

// one block per case: compact the valid slices, interpolate both walls to the W pixel columns
__global__ __launch_bounds__(256) void wall_interp_kernel(const double *__restrict__ zs, int S,
                                                         const double *__restrict__ zf, int W,
                                                         double *__restrict__ ws, int64_t ws_stride) {
  extern __shared__ double sm[];  // [3][S]: xp, lo, hi
  double *xp = sm, *lo = sm + S, *hi = sm + 2 * S;
  __shared__ int nvalid;
  double *w = ws + blockIdx.x * ws_stride;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int s = 0; s < S; ++s) {
      const double a = w[s], b = w[S + s];
      if (a == a && b == b) { xp[n] = zs[s]; lo[n] = a; hi[n] = b; ++n; }
    }
    nvalid = n;
    *(int *)(w + 2 * S + 2 * W) = n;  // status word: number of valid slices (0 -> numpy would raise)
  }
  __syncthreads();
  const int n = nvalid;
  for (int j = threadIdx.x; j < W; j += 256) {
    double a = dnan(), b = dnan();
    if (n > 0) { a = np_interp(zf[j], xp, lo, n); b = np_interp(zf[j], xp, hi, n); }
    w[2 * S + j] = a;
    w[2 * S + W + j] = b;
  }
}

// one thread per pixel; points staged through LDS in tiles
__global__ __launch_bounds__(256) void nearest_kernel(const double *__restrict__ pts, const int64_t *__restrict__ offs,
                                                     const double *__restrict__ yf, int H,
                                                     const double *__restrict__ zf, int W, int S,
                                                     const double *__restrict__ ws, int64_t ws_stride,
                                                     double p_convert, double *__restrict__ out) {
  constexpr int TILE = 1024;
  __shared__ double2 tile[TILE];
  const int cs = blockIdx.y;
  const int64_t beg = offs[cs], end = offs[cs + 1];
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < H * W;
  const int i = live ? pix / W : 0, j = live ? pix - i * W : 0;
  const double qy = yf[i], qz = zf[j];
  double best = __longlong_as_double(0x7ff0000000000000LL);
  int64_t besti = -1;
  for (int64_t t0 = beg; t0 < end; t0 += TILE) {
    const int cnt = (int)((end - t0) < TILE ? (end - t0) : TILE);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += 256) {
      const double *p = pts + (t0 + k) * 5;
      tile[k] = make_double2(p[1], p[2]);
    }
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      const double2 s = tile[k];
      const double dy = s.x - qy, dz = s.y - qz;
      const double d = dy * dy + dz * dz;   // two products, one sum, each rounded (contraction is off)
      if (d < best) { best = d; besti = t0 + k; }   // strict: the lowest index wins ties
    }
  }
  if (!live) return;
  const double *w = ws + cs * ws_stride;
  const double ylo = w[2 * S + j], yhi = w[2 * S + W + j];
  double P = 0.0, U = 0.0;
  if (besti >= 0) {
    P = pts[besti * 5 + 3];
    U = pts[besti * 5 + 4];
    if (P < 0.0) P = 0.0;                      // :182
  }
  if (qy < ylo || qy > yhi) { P = 0.0; U = 0.0; }  // :220-221 (comparisons with NaN are false, as in numpy)
  double *o = out + ((int64_t)cs * H * W + pix) * 5;
  o[0] = qy; o[1] = qz; o[2] = P; o[3] = U; o[4] = P * p_convert;
}

}  // namespace cfd

using namespace cfd;

static inline int64_t ws_stride_doubles(int W, int S) { return 2 * (int64_t)S + 2 * (int64_t)W + 2; }

extern "C" int64_t cfd_resample_workspace_bytes(int ncase, int W, int S) {
  return (int64_t)ncase * ws_stride_doubles(W, S) * 8;
}

extern "C" int cfd_resample_nearest(const double *pts, const int64_t *case_offsets, int ncase, const double *y_final,
                                    int H, const double *z_final, int W, const double *z_samples, int S, double tol1,
                                    double tol2, double p_convert, double *out, void *workspace, void *stream) {
  CFD_REQUIRE(pts && case_offsets && y_final && z_final && z_samples && out && workspace, "NULL pointer");
  CFD_REQUIRE(ncase > 0 && H > 0 && W > 0 && S > 0 && S <= 2048, "bad sizes ncase=%d H=%d W=%d S=%d", ncase, H, W, S);
  CFD_REQUIRE((int64_t)H * W < (1LL << 31), "grid too large");
  hipStream_t st = (hipStream_t)stream;
  double *ws = (double *)workspace;
  const int64_t stride = ws_stride_doubles(W, S);
  hipLaunchKernelGGL(wall_slices_kernel, dim3(S, ncase), dim3(256), 0, st, pts, case_offsets, z_samples, S, tol1, tol2,
                     ws, stride);
  CFD_LAUNCH_CHECK();
  hipLaunchKernelGGL(wall_interp_kernel, dim3(ncase), dim3(256), 3 * S * sizeof(double), st, z_samples, S, z_final, W,
                     ws, stride);
  CFD_LAUNCH_CHECK();
  hipLaunchKernelGGL(nearest_kernel, dim3((H * W + 255) / 256, ncase), dim3(256), 0, st, pts, case_offsets, y_final, H,
                     z_final, W, S, (const double *)ws, stride, p_convert, out);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}
