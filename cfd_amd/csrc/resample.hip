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
// HBM-bound by contract (40 B per input point + 40 B per output pixel), so the search must not cost
// H*W*N distance evaluations: the points of a case are binned into a uniform cell grid (about two
// points per cell, cells square in metres) and every pixel searches rings of cells around its own
// until no unvisited cell can hold a nearer -- or equally near -- point.  The stop test is exact in
// ROUNDED arithmetic: a point outside the visited box [cy0..cy1] x [cz0..cz1] lies beyond one of the
// box's boundary coordinates b, and because fp64 subtraction, squaring and addition are monotone its
// rounded distance is >= (b - q)*(b - q) computed the same way; the search stops only when the best
// distance is STRICTLY below the smallest such bound, so ties (lowest index wins) are never cut off.
// The result is therefore the same argmin the brute-force scan gives, bit for bit, for any grid.
//
// Kernels of one call (all cases at once; ragged cases via the offsets array):
//   resample_init      accumulators (+-inf encodings, zero counts)
//   points_pass        per point: bounding box of its case (wave-aggregated atomics) and the min/max of y in
//                      every z slice it is near (replaces 101 scans of the table per case)
//   case_setup         per case: decode the slices, np.interp the wall to the W columns, choose the cell grid
//   bin_count / bin_scan / bin_scatter   counting sort of the points by cell (cell lists)
//   nearest_kernel     16 x 32-pixel tiles (two pixels per lane); ring search; the tile's 5-column rows leave through LDS as
//                      contiguous runs (whole 128-B lines), not as five 8-B stores at 40-B lane stride
#include "common.h"

namespace cfd {

__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7ff0000000000000LL); }

// order-preserving map double -> uint64 (for atomicMin / atomicMax on doubles)
__device__ __forceinline__ unsigned long long enc(double d) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ __forceinline__ double dec(unsigned long long e) {
  const unsigned long long u = (e >> 63) ? (e & 0x7fffffffffffffffULL) : ~e;
  return __longlong_as_double((long long)u);
}

// ---- workspace layout (host and device agree through this struct) -------------------------------------
struct RsLayout {
  int ncase, W, S;
  int64_t total;        // points of all cases
  int64_t wall_stride;  // doubles per case in the wall block: [S] lo, [S] hi, [W] ylo, [W] yhi, 2 status
  // byte offsets from the workspace base
  int64_t o_wall, o_sacc, o_scnt, o_bbox, o_grid, o_start, o_cnt, o_cell, o_syz, o_spu, o_sidx, bytes;
};

static inline int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }

static RsLayout rs_layout(int ncase, int W, int S, int64_t total) {
  RsLayout L;
  L.ncase = ncase; L.W = W; L.S = S; L.total = total;
  L.wall_stride = 2 * (int64_t)S + 2 * (int64_t)W + 2;
  const int64_t cells = total + 64 * (int64_t)ncase;    // sum over cases of cap_i = max(64, N_i) <= this
  int64_t o = 0;
  L.o_wall = o;  o = align16(o + (int64_t)ncase * L.wall_stride * 8);
  L.o_sacc = o;  o = align16(o + (int64_t)ncase * 4 * S * 8);            // uint64 [ncase][4][S]: mn1 mx1 mn2 mx2
  L.o_scnt = o;  o = align16(o + (int64_t)ncase * 4 * S * 4);            // int    [ncase][4][S]: n1 n2 nan1 nan2
  L.o_bbox = o;  o = align16(o + (int64_t)ncase * 4 * 8);                // uint64 [ncase][4]: ymin ymax zmin zmax
  L.o_grid = o;  o = align16(o + (int64_t)ncase * 8 * 8);                // double [ncase][8]: ny nz y0 z0 1/hy 1/hz hy hz
  L.o_start = o; o = align16(o + (cells + (int64_t)ncase) * 4);          // int cell_start, per case ncell+1
  L.o_cnt = o;   o = align16(o + cells * 4);                             // int cell count / fill
  L.o_cell = o;  o = align16(o + total * 4);                             // int cell id of every point
  L.o_syz = o;   o = align16(o + total * 16);                            // double2 (y,z) sorted by cell
  L.o_spu = o;   o = align16(o + total * 16);                            // double2 (p,U) in the same order
  L.o_sidx = o;  o = align16(o + total * 4);                             // int original index (within the case), sorted
  L.bytes = o;
  return L;
}

// per-case bases inside the cell arrays: cap_i = max(64, N_i) <= 64 + N_i, so 64*cs + offs[cs] never overlaps
__device__ __forceinline__ int64_t cell_base(int cs, int64_t beg) { return 64 * (int64_t)cs + beg; }

__global__ __launch_bounds__(256) void resample_init_kernel(unsigned long long *__restrict__ sacc,
                                                           int *__restrict__ scnt,
                                                           unsigned long long *__restrict__ bbox, int ncase, int S) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nacc = (int64_t)ncase * 4 * S;
  if (i < nacc) {
    const int which = (int)((i / S) & 3);                 // 0, 2: minima; 1, 3: maxima
    sacc[i] = enc((which & 1) ? -dinf() : dinf());
    scnt[i] = 0;
  }
  if (i < (int64_t)ncase * 4) bbox[i] = enc((i & 1) ? -dinf() : dinf());
}

// case of global point index g: the last cs with offs[cs] <= g
__device__ __forceinline__ int case_of(const int64_t *__restrict__ offs, int ncase, int64_t g) {
  int lo = 0, hi = ncase;   // invariant: offs[lo] <= g < offs[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (offs[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// Bounding box of every case and the min/max of y in every z slice a point is near (:190-199).
// grid = (K, ncase): K blocks share a case (block b takes points b*256 + tid, stride K*256).  All accumulation happens in
// LDS (64-bit min/max atomics on order-preserving encodings); a block then folds only the entries it touched into the
// case's global accumulators, so the global atomics per case are O(K * slices touched), not O(points).
// A point within tol1 of a slice counts for class 1 only, a point in [tol1, tol2) for class 2 only: class 2 is consulted
// only when no point is within tol1 (:196), and then the two definitions coincide (needs tol1 <= tol2, else both).
__global__ __launch_bounds__(256) void points_pass_kernel(const double *__restrict__ pts, const int64_t *__restrict__ offs,
                                                         const double *__restrict__ zs, int S, double tol1, double tol2,
                                                         unsigned long long *__restrict__ sacc, int *__restrict__ scnt,
                                                         unsigned long long *__restrict__ bbox) {
  extern __shared__ unsigned long long lacc[];      // [4][S] mn1 mx1 mn2 mx2, then int [2][S] NaN counts, then 4 bbox
  const int cs = blockIdx.y, tid = threadIdx.x;
  int *lnan = (int *)(lacc + 4 * S);
  unsigned long long *lbox = lacc + 4 * S + (2 * S + 1) / 2;
  for (int s = tid; s < 4 * S; s += 256) lacc[s] = enc(((s / S) & 1) ? -dinf() : dinf());
  for (int s = tid; s < 2 * S; s += 256) lnan[s] = 0;
  if (tid < 4) lbox[tid] = enc((tid & 1) ? -dinf() : dinf());
  __syncthreads();
  const int64_t beg = offs[cs], end = offs[cs + 1];
  const bool both = !(tol1 <= tol2);
  double ymn = dinf(), ymx = -dinf(), zmn = dinf(), zmx = -dinf();
  for (int64_t g = beg + (int64_t)blockIdx.x * 256 + tid; g < end; g += (int64_t)gridDim.x * 256) {
    const double y = pts[g * 5 + 1], z = pts[g * 5 + 2];
    if ((y - y == 0.0) && (z - z == 0.0)) {          // bounding box: finite coordinates only
      ymn = fmin(ymn, y); ymx = fmax(ymx, y); zmn = fmin(zmn, z); zmx = fmax(zmx, z);
    }
    if (!(z - z == 0.0)) continue;                   // |NaN - zv| < tol is false for every slice
    // candidate slices from a window twice the tolerance wide (zs ascending), the exact test inside
    const double wlo = z - 2.0 * tol2 - 2.0 * tol1, whi = z + 2.0 * tol2 + 2.0 * tol1;
    int lo = 0, hi = S;                              // first slice with zs >= wlo
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (zs[mid] < wlo) lo = mid + 1; else hi = mid;
    }
    const bool ynan = y != y;
    for (int s = lo; s < S && zs[s] <= whi; ++s) {
      const double d = fabs(z - zs[s]);
      const bool in1 = d < tol1, in2 = d < tol2;
      if (in1) {
        if (ynan) atomicAdd(lnan + s, 1);
        else { atomicMin(lacc + 0 * S + s, enc(y)); atomicMax(lacc + 1 * S + s, enc(y)); }
      }
      if (in2 && (both || !in1)) {
        if (ynan) atomicAdd(lnan + S + s, 1);
        else { atomicMin(lacc + 2 * S + s, enc(y)); atomicMax(lacc + 3 * S + s, enc(y)); }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ymn = fmin(ymn, __shfl_xor(ymn, o)); ymx = fmax(ymx, __shfl_xor(ymx, o));
    zmn = fmin(zmn, __shfl_xor(zmn, o)); zmx = fmax(zmx, __shfl_xor(zmx, o));
  }
  if ((tid & 63) == 0) {
    atomicMin(lbox + 0, enc(ymn)); atomicMax(lbox + 1, enc(ymx));
    atomicMin(lbox + 2, enc(zmn)); atomicMax(lbox + 3, enc(zmx));
  }
  __syncthreads();
  // fold what this block touched into the case's accumulators
  unsigned long long *a = sacc + (int64_t)cs * 4 * S;
  int *c = scnt + (int64_t)cs * 4 * S;
  for (int s = tid; s < 2 * S; s += 256) {           // s < S: class 1, else class 2
    const int k = s < S ? 0 : 2, q = s < S ? s : s - S;
    const unsigned long long mn = lacc[k * S + q], mx = lacc[(k + 1) * S + q];
    const int nn = lnan[s];
    if (mn <= mx) {                                  // at least one finite y landed here
      atomicMin(a + k * S + q, mn);
      atomicMax(a + (k + 1) * S + q, mx);
      atomicAdd(c + (k >> 1) * S + q, 1);            // "some point in this class": > 0 is all that is read
    }
    if (nn > 0) {
      atomicAdd(c + (k >> 1) * S + q, 1);
      atomicAdd(c + (2 + (k >> 1)) * S + q, nn);
    }
  }
  if (tid < 4) {
    if (tid & 1) atomicMax(bbox + (int64_t)cs * 4 + tid, lbox[tid]);
    else atomicMin(bbox + (int64_t)cs * 4 + tid, lbox[tid]);
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

// one block per case: slice results -> wall block, interpolate both walls to the W pixel columns, choose the cell grid
__global__ __launch_bounds__(256) void case_setup_kernel(const int64_t *__restrict__ offs, const double *__restrict__ zs,
                                                        int S, const double *__restrict__ zf, int W,
                                                        const unsigned long long *__restrict__ sacc,
                                                        const int *__restrict__ scnt,
                                                        const unsigned long long *__restrict__ bbox,
                                                        double *__restrict__ wall, int64_t wall_stride,
                                                        double *__restrict__ grid,
                                                        double pts_per_cell) {
  extern __shared__ double sm[];  // [3][S]: xp, lo, hi
  double *xp = sm, *lo = sm + S, *hi = sm + 2 * S;
  __shared__ int nvalid;
  const int cs = blockIdx.x;
  double *w = wall + (int64_t)cs * wall_stride;
  const unsigned long long *a = sacc + (int64_t)cs * 4 * S;
  const int *c = scnt + (int64_t)cs * 4 * S;
  for (int s = threadIdx.x; s < S; s += 256) {
    double l, h;
    if (c[0 * S + s] > 0) {                                     // some point within tol1 (:192-195)
      if (c[2 * S + s] > 0) { l = dnan(); h = dnan(); }         // np.min / np.max propagate a NaN
      else { l = dec(a[0 * S + s]); h = dec(a[1 * S + s]); }
    } else if (c[1 * S + s] > 0) {                              // else within tol2 (:197-199)
      if (c[3 * S + s] > 0) { l = dnan(); h = dnan(); }
      else { l = dec(a[2 * S + s]); h = dec(a[3 * S + s]); }
    } else { l = dnan(); h = dnan(); }
    w[s] = l;
    w[S + s] = h;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
    for (int s = 0; s < S; ++s) {
      const double l = w[s], h = w[S + s];
      if (l == l && h == h) { xp[n] = zs[s]; lo[n] = l; hi[n] = h; ++n; }
    }
    nvalid = n;
    *(int *)(w + 2 * S + 2 * W) = n;  // status word: number of valid slices (0 -> numpy would raise)
    // ---- cell grid: ~2 points per cell, cells square in metres, at most cap = max(64, N) cells
    const int64_t beg = offs[cs], N = offs[cs + 1] - beg;
    const double ymn = dec(bbox[cs * 4 + 0]), ymx = dec(bbox[cs * 4 + 1]);
    const double zmn = dec(bbox[cs * 4 + 2]), zmx = dec(bbox[cs * 4 + 3]);
    const bool any = ymn <= ymx && zmn <= zmx;
    const double ry = any ? ymx - ymn : 0.0, rz = any ? zmx - zmn : 0.0;
    const int64_t cap = N > 64 ? N : 64;
    const double tcells = floor((double)N / pts_per_cell);
    const double target = tcells > 1.0 ? tcells : 1.0;
    int64_t ny = 1, nz = 1;
    if (ry > 0.0 && rz > 0.0) {
      const double h = sqrt(ry * rz / target);
      const double fy = floor(ry / h) + 1.0, fz = floor(rz / h) + 1.0;
      ny = fy < (double)cap ? (int64_t)fy : cap;
      nz = fz < (double)cap ? (int64_t)fz : cap;
    } else if (ry > 0.0) {
      ny = (int64_t)target < cap ? (int64_t)target : cap;
    } else if (rz > 0.0) {
      nz = (int64_t)target < cap ? (int64_t)target : cap;
    }
    while (ny * nz > cap) { if (ny >= nz) ny = (ny + 1) / 2; else nz = (nz + 1) / 2; }
    const double g_y0 = any ? ymn : 0.0, g_z0 = any ? zmn : 0.0;
    const double g_hy = ry > 0.0 ? ry / (double)ny : 1.0, g_hz = rz > 0.0 ? rz / (double)nz : 1.0;
    double *gp = grid + (int64_t)cs * 8;
    gp[0] = (double)ny; gp[1] = (double)nz; gp[2] = g_y0; gp[3] = g_z0; gp[4] = 1.0 / g_hy; gp[5] = 1.0 / g_hz;
    gp[6] = g_hy; gp[7] = g_hz;
  }
  __syncthreads();
  const int n = nvalid;
  for (int j = threadIdx.x; j < W; j += 256) {
    double l = dnan(), h = dnan();
    if (n > 0) { l = np_interp(zf[j], xp, lo, n); h = np_interp(zf[j], xp, hi, n); }
    w[2 * S + j] = l;
    w[2 * S + W + j] = h;
  }
  // (cell boundary k of an axis is y0 + k*hy, evaluated where needed: cell_bound(); only their ORDER matters)
}

// cell of coordinate v along one axis: cell k holds b[k] <= v < b[k+1]; cells 0 and n-1 are open-ended.
// The guess only saves steps; the two loops make the result agree with the boundary table exactly, which is what
// the search's stop test relies on.  NaN -> cell 0.
// boundary k of an axis: ONE expression everywhere (binning of the points, of the pixels, and the stop test), each
// operation rounded once (-ffp-contract=off), so every kernel sees the same doubles without a boundary table in memory
// (the table cost two or three dependent loads per axis per pixel, and the kernel is a chain of load latencies)
__device__ __forceinline__ double cell_bound(double o, double h, int k) { return o + (double)k * h; }

__device__ __forceinline__ int bin1(double v, double o, double inv, double h, int n) {
  const double t = (v - o) * inv;
  int k = (t >= 0.0) ? (t < (double)(n - 1) ? (int)t : n - 1) : 0;
  while (k + 1 < n && v >= cell_bound(o, h, k + 1)) ++k;
  while (k > 0 && v < cell_bound(o, h, k)) --k;
  return k;
}

struct CaseGrid {
  int ny, nz;
  double y0, z0, invy, invz, hy, hz;
  __device__ __forceinline__ double by(int k) const { return cell_bound(y0, hy, k); }
  __device__ __forceinline__ double bz(int k) const { return cell_bound(z0, hz, k); }
};
__device__ __forceinline__ CaseGrid load_grid(const double *__restrict__ grid, int cs) {
  const double *gp = grid + (int64_t)cs * 8;     // one 64-B record per case
  CaseGrid G;
  G.ny = (int)gp[0]; G.nz = (int)gp[1]; G.y0 = gp[2]; G.z0 = gp[3]; G.invy = gp[4]; G.invz = gp[5]; G.hy = gp[6]; G.hz = gp[7];
  return G;
}

__global__ __launch_bounds__(256) void bin_count_kernel(const double *__restrict__ pts, const int64_t *__restrict__ offs,
                                                       int ncase, int64_t total, const double *__restrict__ grid,
                                                       int *__restrict__ cnt,
                                                       int *__restrict__ cell) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int cs = case_of(offs, ncase, g);
  const int64_t beg = offs[cs];
  const CaseGrid G = load_grid(grid, cs);
  const int cy = bin1(pts[g * 5 + 1], G.y0, G.invy, G.hy, G.ny);
  const int cz = bin1(pts[g * 5 + 2], G.z0, G.invz, G.hz, G.nz);
  const int c = cy * G.nz + cz;
  cell[g] = c;
  atomicAdd(cnt + cell_base(cs, beg) + c, 1);
}

// one block per case: exclusive scan of the cell counts -> cell_start[0..ncell]; counts reset to 0 (scatter's fill)
__global__ __launch_bounds__(256) void bin_scan_kernel(const int64_t *__restrict__ offs, const double *__restrict__ grid,
                                                      int *__restrict__ cnt, int *__restrict__ start) {
  __shared__ int part[256];
  const int cs = blockIdx.x, tid = threadIdx.x;
  const int64_t beg = offs[cs];
  const double *gp = grid + (int64_t)cs * 8;
  const int ncell = (int)gp[0] * (int)gp[1];
  int *c = cnt + cell_base(cs, beg);
  int *st = start + cell_base(cs, beg) + cs;
  const int per = (ncell + 255) / 256;
  const int a = tid * per, b = (a + per < ncell) ? a + per : ncell;
  int sum = 0;
  for (int i = a; i < b; ++i) sum += c[i];
  part[tid] = sum;
  __syncthreads();
  // Hillis-Steele inclusive scan of the 256 partials
  for (int o = 1; o < 256; o <<= 1) {
    const int v = (tid >= o) ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;   // exclusive prefix of this thread's segment
  for (int i = a; i < b; ++i) {
    const int v = c[i];
    st[i] = run;
    run += v;
    c[i] = 0;
  }
  if (tid == 255) st[ncell] = part[255];
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(const double *__restrict__ pts, const int64_t *__restrict__ offs,
                                                         int ncase, int64_t total, const int *__restrict__ cell,
                                                         const int *__restrict__ start, int *__restrict__ cnt,
                                                         double2 *__restrict__ syz, double2 *__restrict__ spu,
                                                         int *__restrict__ sidx) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= total) return;
  const int cs = case_of(offs, ncase, g);
  const int64_t beg = offs[cs];
  const int c = cell[g];
  const int pos = start[cell_base(cs, beg) + cs + c] + atomicAdd(cnt + cell_base(cs, beg) + c, 1);
  syz[beg + pos] = make_double2(pts[g * 5 + 1], pts[g * 5 + 2]);
  spu[beg + pos] = make_double2(pts[g * 5 + 3], pts[g * 5 + 4]);
  sidx[beg + pos] = (int)(g - beg);
}

// ---- everything that precedes the search, for ONE case per block (round 3) -----------------------------------------
// resample_init + points_pass + case_setup + bin_count / bin_scan / bin_scatter + the memset of the cell counts were six
// launches and a memset whose ~100 us were a third of a 256-case call although each case's share is a few microseconds of
// work: all of it is per-case, so one block of 1024 threads walks the phases with block barriers in between -- slice
// accumulators and bounding box in LDS (no global accumulators at all), then the wall, the cell grid, and the counting sort
// with this case's counters in global memory (zeroed here, by the block that uses them).  Same arithmetic, same results;
// used when the average case fits a block's sweep (the multi-launch path stays for huge cases, K blocks per case).
constexpr int PREP_T = 1024;
constexpr int PREP_LDS_CELLS = 8192;

// count -> exclusive scan -> scatter for one case, by one block of PREP_T threads.  cntp (zeroed) / startp: the counters and
// cell starts it works on (LDS or global); start_out (may be NULL = startp is already the global array) receives the starts.
__device__ __forceinline__ void prepare_sort(int *cntp, int *startp, int *start_out, const CaseGrid &G, int ncell,
                                             const double *__restrict__ pts, int64_t beg, int64_t end,
                                             int *__restrict__ cell, double2 *__restrict__ syz, double2 *__restrict__ spu,
                                             int *__restrict__ sidx, int *part) {
  const int tid = threadIdx.x;
  for (int64_t g = beg + tid; g < end; g += PREP_T) {
    const int cy = bin1(pts[g * 5 + 1], G.y0, G.invy, G.hy, G.ny);
    const int cz = bin1(pts[g * 5 + 2], G.z0, G.invz, G.hz, G.nz);
    const int cc = cy * G.nz + cz;
    cell[g] = cc;
    atomicAdd(cntp + cc, 1);
  }
  __syncthreads();
  const int per = (ncell + PREP_T - 1) / PREP_T;
  const int a = tid * per < ncell ? tid * per : ncell, b = (a + per < ncell) ? a + per : ncell;
  int sum = 0;
  for (int i = a; i < b; ++i) sum += cntp[i];
  part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < PREP_T; o <<= 1) {
    const int v = (tid >= o) ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
  for (int i = a; i < b; ++i) {
    const int v = cntp[i];
    startp[i] = run;
    if (start_out) start_out[i] = run;
    run += v;
    cntp[i] = 0;
  }
  if (tid == PREP_T - 1) {
    startp[ncell] = part[PREP_T - 1];
    if (start_out) start_out[ncell] = part[PREP_T - 1];
  }
  __syncthreads();
  for (int64_t g = beg + tid; g < end; g += PREP_T) {
    const int cc = cell[g];
    const int pos = startp[cc] + atomicAdd(cntp + cc, 1);
    syz[beg + pos] = make_double2(pts[g * 5 + 1], pts[g * 5 + 2]);
    spu[beg + pos] = make_double2(pts[g * 5 + 3], pts[g * 5 + 4]);
    sidx[beg + pos] = (int)(g - beg);
  }
}

__global__ __launch_bounds__(PREP_T) void case_prepare_kernel(
    const double *__restrict__ pts, const int64_t *__restrict__ offs, const double *__restrict__ zs, int S, double tol1,
    double tol2, const double *__restrict__ zf, int W, double *__restrict__ wall, int64_t wall_stride,
    double *__restrict__ grid, int *__restrict__ start, int *__restrict__ cnt, int *__restrict__ cell,
    double2 *__restrict__ syz, double2 *__restrict__ spu, int *__restrict__ sidx, double pts_per_cell) {
  extern __shared__ unsigned long long lacc[];      // [4][S] mn1 mx1 mn2 mx2 | double [6][S] zs, xp, lo, hi, l_all, h_all | int [2][S] NaN counts
  double *lzs = (double *)(lacc + 4 * S), *xp = lzs + S, *lo = xp + S, *hi = lo + S, *lall = hi + S, *hall = lall + S;
  int *lnan = (int *)(hall + S);
  __shared__ unsigned long long lbox[4];
  __shared__ int part[PREP_T];
  __shared__ int nvalid;
  __shared__ double gsh[8];
  const int cs = blockIdx.x, tid = threadIdx.x;
  const int64_t beg = offs[cs], end = offs[cs + 1], N = end - beg;
  const int64_t cap = N > 64 ? N : 64;
  int *c = cnt + cell_base(cs, beg);
  int *st = start + cell_base(cs, beg) + cs;
  for (int s = tid; s < 4 * S; s += PREP_T) lacc[s] = enc(((s / S) & 1) ? -dinf() : dinf());
  for (int s = tid; s < 2 * S; s += PREP_T) lnan[s] = 0;
  for (int s = tid; s < S; s += PREP_T) lzs[s] = zs[s];
  if (tid < 4) lbox[tid] = enc((tid & 1) ? -dinf() : dinf());
  for (int64_t i = tid; i < cap; i += PREP_T) c[i] = 0;
  __syncthreads();
  // ---- points pass (points_pass_kernel's loop with one block per case)
  {
    const bool both = !(tol1 <= tol2);
    double ymn = dinf(), ymx = -dinf(), zmn = dinf(), zmx = -dinf();
    for (int64_t g = beg + tid; g < end; g += PREP_T) {
      const double y = pts[g * 5 + 1], z = pts[g * 5 + 2];
      if ((y - y == 0.0) && (z - z == 0.0)) {
        ymn = fmin(ymn, y); ymx = fmax(ymx, y); zmn = fmin(zmn, z); zmx = fmax(zmx, z);
      }
      if (!(z - z == 0.0)) continue;
      const double wlo = z - 2.0 * tol2 - 2.0 * tol1, whi = z + 2.0 * tol2 + 2.0 * tol1;
      int a = 0, b = S;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (lzs[mid] < wlo) a = mid + 1; else b = mid;
      }
      const bool ynan = y != y;
      for (int s = a; s < S && lzs[s] <= whi; ++s) {
        const double d = fabs(z - lzs[s]);
        const bool in1 = d < tol1, in2 = d < tol2;
        if (in1) {
          if (ynan) atomicAdd(lnan + s, 1);
          else { atomicMin(lacc + 0 * S + s, enc(y)); atomicMax(lacc + 1 * S + s, enc(y)); }
        }
        if (in2 && (both || !in1)) {
          if (ynan) atomicAdd(lnan + S + s, 1);
          else { atomicMin(lacc + 2 * S + s, enc(y)); atomicMax(lacc + 3 * S + s, enc(y)); }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ymn = fmin(ymn, __shfl_xor(ymn, o)); ymx = fmax(ymx, __shfl_xor(ymx, o));
      zmn = fmin(zmn, __shfl_xor(zmn, o)); zmx = fmax(zmx, __shfl_xor(zmx, o));
    }
    if ((tid & 63) == 0) {
      atomicMin(lbox + 0, enc(ymn)); atomicMax(lbox + 1, enc(ymx));
      atomicMin(lbox + 2, enc(zmn)); atomicMax(lbox + 3, enc(zmx));
    }
  }
  __syncthreads();
  // ---- slices -> wall block (case_setup_kernel's decode, reading the LDS accumulators)
  double *w = wall + (int64_t)cs * wall_stride;
  for (int s = tid; s < S; s += PREP_T) {
    const bool has1 = lacc[0 * S + s] <= lacc[1 * S + s] || lnan[s] > 0;          // some point within tol1 (:192-195)
    const bool has2 = lacc[2 * S + s] <= lacc[3 * S + s] || lnan[S + s] > 0;      // else within tol2 (:197-199)
    double l, h;
    if (has1) {
      if (lnan[s] > 0) { l = dnan(); h = dnan(); }                                // np.min / np.max propagate a NaN
      else { l = dec(lacc[0 * S + s]); h = dec(lacc[1 * S + s]); }
    } else if (has2) {
      if (lnan[S + s] > 0) { l = dnan(); h = dnan(); }
      else { l = dec(lacc[2 * S + s]); h = dec(lacc[3 * S + s]); }
    } else { l = dnan(); h = dnan(); }
    w[s] = l;
    w[S + s] = h;
    lall[s] = l;
    hall[s] = h;
  }
  __syncthreads();
  // valid slices, compacted in order, by one wave (ballot + popcount per 64 slices): a single thread walking the S slices
  // through global memory was a chain of S round trips -- most of the 23 us case_setup_kernel took
  if (tid < 64) {
    int n = 0;
    for (int base = 0; base < S; base += 64) {
      const int sl = base + tid;
      double l = dnan(), h = dnan();
      if (sl < S) { l = lall[sl]; h = hall[sl]; }
      const bool ok = l == l && h == h;
      const unsigned long long mask = __ballot(ok);
      const int pos = n + __popcll(mask & ((1ULL << tid) - 1ULL));
      if (ok) { xp[pos] = lzs[sl]; lo[pos] = l; hi[pos] = h; }
      n += __popcll(mask);
    }
    if (tid == 0) {
      nvalid = n;
      *(int *)(w + 2 * S + 2 * W) = n;
    }
  }
  if (tid == 64) {
    // ---- cell grid: the same choice as case_setup_kernel
    const double ymn = dec(lbox[0]), ymx = dec(lbox[1]), zmn = dec(lbox[2]), zmx = dec(lbox[3]);
    const bool any = ymn <= ymx && zmn <= zmx;
    const double ry = any ? ymx - ymn : 0.0, rz = any ? zmx - zmn : 0.0;
    const double tcells = floor((double)N / pts_per_cell);
    const double target = tcells > 1.0 ? tcells : 1.0;
    int64_t ny = 1, nz = 1;
    if (ry > 0.0 && rz > 0.0) {
      const double h = sqrt(ry * rz / target);
      const double fy = floor(ry / h) + 1.0, fz = floor(rz / h) + 1.0;
      ny = fy < (double)cap ? (int64_t)fy : cap;
      nz = fz < (double)cap ? (int64_t)fz : cap;
    } else if (ry > 0.0) {
      ny = (int64_t)target < cap ? (int64_t)target : cap;
    } else if (rz > 0.0) {
      nz = (int64_t)target < cap ? (int64_t)target : cap;
    }
    while (ny * nz > cap) { if (ny >= nz) ny = (ny + 1) / 2; else nz = (nz + 1) / 2; }
    const double g_y0 = any ? ymn : 0.0, g_z0 = any ? zmn : 0.0;
    const double g_hy = ry > 0.0 ? ry / (double)ny : 1.0, g_hz = rz > 0.0 ? rz / (double)nz : 1.0;
    gsh[0] = (double)ny; gsh[1] = (double)nz; gsh[2] = g_y0; gsh[3] = g_z0; gsh[4] = 1.0 / g_hy; gsh[5] = 1.0 / g_hz;
    gsh[6] = g_hy; gsh[7] = g_hz;
  }
  __syncthreads();
  if (tid < 8) grid[(int64_t)cs * 8 + tid] = gsh[tid];
  {
    const int n = nvalid;
    for (int j = tid; j < W; j += PREP_T) {
      double l = dnan(), h = dnan();
      if (n > 0) { l = np_interp(zf[j], xp, lo, n); h = np_interp(zf[j], xp, hi, n); }
      w[2 * S + j] = l;
      w[2 * S + W + j] = h;
    }
  }
  // ---- counting sort of the points by cell: counters and prefix sums in LDS when the case's cells fit (PREP_LDS_CELLS),
  // else in the global arrays; the cell starts go to global memory either way (the search reads them)
  CaseGrid G;
  G.ny = (int)gsh[0]; G.nz = (int)gsh[1]; G.y0 = gsh[2]; G.z0 = gsh[3]; G.invy = gsh[4]; G.invz = gsh[5]; G.hy = gsh[6]; G.hz = gsh[7];
  const int ncell = G.ny * G.nz;
  int *lc = lnan + 2 * S;                       // [PREP_LDS_CELLS] counts / fill, then [PREP_LDS_CELLS + 1] starts
  int *ls = lc + PREP_LDS_CELLS;
  const bool in_lds = ncell <= PREP_LDS_CELLS;  // (uniform)
  if (in_lds) {
    for (int i = tid; i < ncell; i += PREP_T) lc[i] = 0;
    __syncthreads();
    prepare_sort(lc, ls, st, G, ncell, pts, beg, end, cell, syz, spu, sidx, part);
  } else {
    prepare_sort(c, st, (int *)nullptr, G, ncell, pts, beg, end, cell, syz, spu, sidx, part);
  }
}

// ordering point for LDS traffic between the lanes of ONE wave (its LDS operations execute in program order; this keeps
// the compiler from moving them across and waits for the writes)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int NPX = 2;                   // pixels per lane: rows lr and lr + 8 of the wave's 16 x 8 pixels
constexpr int TILE_H = 8 * NPX, TILE_W = 32;   // pixels per block: 16 rows x 32 columns; one wave = 16 rows x 8 columns
constexpr int WIN_CAP = 128;             // points a wave stages in LDS for its pixels
constexpr int WIN_ROWS = 8;              // cell rows of a wave's window at most (else the general search)

// General search of one pixel: rings of cells around its own cell, read from global memory, with the exact stop test
// (see the header).  Used when a wave's window does not fit its LDS budget or cannot decide a pixel.
__device__ void ring_search(double qy, double qz, const CaseGrid &G, const int *__restrict__ st,
                            const double2 *__restrict__ sp, const int *__restrict__ si, double &best, int &bestk,
                            unsigned &nev) {
  best = dinf();
  bestk = -1;            // position in the sorted order; the original index is looked up only on ties and at the end
  const int cqy = bin1(qy, G.y0, G.invy, G.hy, G.ny), cqz = bin1(qz, G.z0, G.invz, G.hz, G.nz);
  for (int r = 1;; ++r) {
    const int y0 = cqy - r > 0 ? cqy - r : 0, y1 = cqy + r < G.ny - 1 ? cqy + r : G.ny - 1;
    const int z0 = cqz - r > 0 ? cqz - r : 0, z1 = cqz + r < G.nz - 1 ? cqz + r : G.nz - 1;
    for (int cy = y0; cy <= y1; ++cy) {
      const bool full = r == 1 || cy == cqy - r || cy == cqy + r;
      // full row of the box: cells (cy, z0..z1) are consecutive in the sorted order.  Middle rows of ring r >= 2:
      // only the two end cells (cy, cqz - r) and (cy, cqz + r) are new.
      for (int part = 0; part < 2; ++part) {
        int ca, cb;
        if (full) {
          if (part) break;
          ca = cy * G.nz + z0; cb = cy * G.nz + z1 + 1;
        } else {
          const int cz = part ? cqz + r : cqz - r;
          if (cz < 0 || cz > G.nz - 1) continue;
          ca = cy * G.nz + cz; cb = ca + 1;
        }
        const int pa = st[ca], pb = st[cb];
        nev += (unsigned)(pb - pa);
        for (int k = pa; k < pb; ++k) {
          const double2 s = sp[k];
          const double dy = s.x - qy, dz = s.y - qz;
          const double d = dy * dy + dz * dz;   // two products, one sum, each rounded (contraction is off)
          if (d < best) { best = d; bestk = k; }
          else if (d == best && bestk >= 0 && si[k] < si[bestk]) bestk = k;   // ties: the lowest original index
        }
      }
    }
    if (y0 == 0 && y1 == G.ny - 1 && z0 == 0 && z1 == G.nz - 1) break;    // every cell visited
    // smallest rounded distance any point outside the box can have (see the header): stop only STRICTLY below it
    double B = dinf();
    if (y1 < G.ny - 1) { const double t = G.by(y1 + 1) - qy; B = fmin(B, t * t); }
    if (y0 > 0)        { const double t = G.by(y0) - qy;     B = fmin(B, t * t); }
    if (z1 < G.nz - 1) { const double t = G.bz(z1 + 1) - qz; B = fmin(B, t * t); }
    if (z0 > 0)        { const double t = G.bz(z0) - qz;     B = fmin(B, t * t); }
    if (best < B) break;
  }
}

// The 64 pixels of a wave (8 x 8, ~0.5 x 0.8 mm at the reference grid) fall into a handful of cells, so the wave stages
// ONE window of cells -- the cells its pixels fall in plus one ring -- into LDS (whole rows of the window are consecutive
// in the sorted order: one range per cell row) and every lane scans the same ~30-50 points with broadcast LDS reads, no
// divergence and no dependent global loads.  The scan is exact for a pixel iff its best distance is STRICTLY below the
// smallest rounded distance a point outside the window can have (the same bound as the ring search, for the window's
// box); any other pixel -- empty neighbourhoods, oversized windows -- takes the general search.  Pixels outside the
// walls are written as zeros whatever their nearest point is (:220-221), so they search nothing.
// Measured (r02, 256 cases): the kernel is a chain of dependent round trips (case record -> cell starts -> points ->
// values) plus ~35 FP64 distance evaluations per pixel; (p, U) ride with the window so the last round trip is LDS.
// COUNT: also tally the distance evaluations (one = two subtractions, two products, one sum, in fp64) into *evals -- the
// bench's FP64 roofline of this kernel is measured work, not an estimate (cfd_resample_nearest_counted)
// Occupancy: 18 KiB of LDS per block = 8 blocks = 32 waves per CU; left to itself the compiler takes 80+ registers (6 waves
// per SIMD), held to 8 waves it needs 63 and spills nothing (the counting variant spills; it runs once per bench) -- the
// kernel is a latency chain, so the two extra waves pay: 257 -> 251 us per 256-case call.
// Two pixels per lane (rows lr and lr + 8: a wave owns 16 x 8 pixels, round 3): a wave's chain of dependent round trips
// (case record -> cell starts -> points) is paid once per 128 pixels instead of once per 64, every window point read from
// LDS serves two pixels and the column term (z - qz)^2 is shared; the window grows by one or two cell rows, so the
// evaluations per pixel go 23.8 -> 31.7 -- and the call 251 -> 226 us.  Four pixels per lane: 41.7 evaluations, 234 us.
template <bool COUNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void nearest_kernel(const int64_t *__restrict__ offs,
                                                     int ncase, const double *__restrict__ yf, int H,
                                                     const double *__restrict__ zf, int W, int S,
                                                     const double *__restrict__ wall, int64_t wall_stride,
                                                     const double *__restrict__ grid,
                                                     const int *__restrict__ start, const double2 *__restrict__ syz,
                                                     const double2 *__restrict__ spu,
                                                     const int *__restrict__ sidx, double p_convert,
                                                     double *__restrict__ out, unsigned long long *__restrict__ evals) {
  unsigned nev = 0;
  // per wave: [WIN_CAP] (y,z), then [WIN_CAP] (p,U); the wave's 8 x 8 x 5 output doubles reuse the same bytes once the
  // wave has finished with its window.  The four waves of a block share NOTHING and never wait for each other (a
  // block-wide barrier made every wave as slow as the block's slowest lane, e.g. one pixel in the general search).
  __shared__ __align__(16) double lds[4 * WIN_CAP * 4];
  __shared__ int wids[4][WIN_CAP];
  static_assert(64 * 5 <= WIN_CAP * 4, "a wave's output rows must fit its window storage");
  const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  const int tiles = tiles_x * tiles_y;
  // XCD-aware order: blocks L, L+8, L+16, ... share an XCD (round-robin dispatch), so give each XCD whole cases --
  // its L2 then holds the sorted points and cell lists of the cases it works on and no other XCD re-reads them
  const int64_t L = blockIdx.x;
  const int ncase8 = ncase & ~7;
  int cs, tile;
  if (L < (int64_t)ncase8 * tiles) {
    const int xcd = (int)(L & 7);
    const int64_t j = L >> 3;
    cs = (int)(j / tiles) * 8 + xcd;
    tile = (int)(j % tiles);
  } else {
    cs = (int)(L / tiles);
    tile = (int)(L % tiles);
  }
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane >> 3, lc = wave * 8 + (lane & 7);          // row (of the first pixel) / column inside the block's tile
  const int j = tx * TILE_W + lc;
  // everything that depends on nothing but the block's coordinates is requested first, in one round trip
  const int64_t beg = offs[cs];
  const CaseGrid G = load_grid(grid, cs);
  const double *w = wall + (int64_t)cs * wall_stride;
  const bool livec = j < W;
  const double qz = zf[livec ? j : 0];
  const double ylo = w[2 * S + (livec ? j : 0)], yhi = w[2 * S + W + (livec ? j : 0)];
  double qy[NPX];
  bool wanted[NPX];
#pragma unroll
  for (int h = 0; h < NPX; ++h) {
    const int i = ty * TILE_H + h * 8 + lr;
    const bool live = livec && i < H;
    qy[h] = yf[live ? i : 0];
    wanted[h] = live && !(qy[h] < ylo || qy[h] > yhi);   // (comparisons with NaN are false, as in numpy: then search)
  }
  const int *st = start + cell_base(cs, beg) + cs;
  const double2 *sp = syz + beg;
  const double2 *su = spu + beg;
  const int *si = sidx + beg;
  double2 *wpts = (double2 *)lds + wave * 2 * WIN_CAP;
  double2 *wpu = wpts + WIN_CAP;
  int *wid = wids[wave];

  // ---- the wave's window of cells (over the pixels that need a nearest point)
  int ymin = 0x7fffffff, ymax = -1, zmin = 0x7fffffff, zmax = -1;
#pragma unroll
  for (int h = 0; h < NPX; ++h)
    if (wanted[h]) {
      const int cqy = bin1(qy[h], G.y0, G.invy, G.hy, G.ny), cqz = bin1(qz, G.z0, G.invz, G.hz, G.nz);
      ymin = cqy < ymin ? cqy : ymin; ymax = cqy > ymax ? cqy : ymax;
      zmin = cqz < zmin ? cqz : zmin; zmax = cqz > zmax ? cqz : zmax;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int a = __shfl_xor(ymin, o), b = __shfl_xor(ymax, o), c = __shfl_xor(zmin, o), d = __shfl_xor(zmax, o);
    ymin = a < ymin ? a : ymin; ymax = b > ymax ? b : ymax; zmin = c < zmin ? c : zmin; zmax = d > zmax ? d : zmax;
  }
  bool fast = false;
  int T = 0, wy0 = 0, wy1 = -1, wz0 = 0, wz1 = -1;
  if (ymax >= 0) {                                   // (uniform) some pixel of the wave needs a search
    wy0 = ymin > 0 ? ymin - 1 : 0; wy1 = ymax < G.ny - 1 ? ymax + 1 : G.ny - 1;
    wz0 = zmin > 0 ? zmin - 1 : 0; wz1 = zmax < G.nz - 1 ? zmax + 1 : G.nz - 1;
    const int nrows = wy1 - wy0 + 1;
    if (nrows <= WIN_ROWS) {
      int pa = 0, pb = 0;
      if (lane < nrows) { pa = st[(wy0 + lane) * G.nz + wz0]; pb = st[(wy0 + lane) * G.nz + wz1 + 1]; }
      const int cnt = pb - pa;
      int pre = cnt;                                 // inclusive prefix over the rows (lanes 0..7; cnt = 0 beyond nrows)
#pragma unroll
      for (int o = 1; o < WIN_ROWS; o <<= 1) {
        const int t = __shfl_up(pre, o);
        if (lane >= o) pre += t;
      }
      T = __shfl(pre, WIN_ROWS - 1);
      if (T <= WIN_CAP) {
        fast = true;
        // window slot e (two per lane) <- its row's range of the sorted order; all loads leave before any lands
        int src[2] = {-1, -1};
#pragma unroll
        for (int r = 0; r < WIN_ROWS; ++r) {
          const int pr = __shfl(pre, r), rc = __shfl(cnt, r), ra = __shfl(pa, r);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int e = lane + 64 * h;
            if (e >= pr - rc && e < pr) src[h] = ra + (e - (pr - rc));
          }
        }
        double2 a0, a1, u0, u1;
        int i0 = 0, i1 = 0;
        if (src[0] >= 0) { a0 = sp[src[0]]; u0 = su[src[0]]; i0 = si[src[0]]; }
        if (src[1] >= 0) { a1 = sp[src[1]]; u1 = su[src[1]]; i1 = si[src[1]]; }
        if (src[0] >= 0) { wpts[lane] = a0; wpu[lane] = u0; wid[lane] = i0; }
        if (src[1] >= 0) { wpts[lane + 64] = a1; wpu[lane + 64] = u1; wid[lane + 64] = i1; }
      }
    }
  }
  wave_sync();                                       // the wave's LDS writes before its reads

  double P[NPX], U[NPX];
#pragma unroll
  for (int h = 0; h < NPX; ++h) { P[h] = 0.0; U[h] = 0.0; }
  bool decided[NPX];
#pragma unroll
  for (int h = 0; h < NPX; ++h) decided[h] = !wanted[h];
  if (fast && (wanted[0] || wanted[NPX - 1])) {
    // first pass, both pixels of the lane against every point read: strict minimum in scan order, and whether any candidate
    // EQUALLED the running minimum -- only then (rare) that pixel's pass is repeated with the lowest-original-index tie rule
    double best[NPX];
    int bk[NPX];
    bool tie[NPX];
#pragma unroll
    for (int h = 0; h < NPX; ++h) { best[h] = dinf(); bk[h] = -1; tie[h] = false; }
    int k = 0;
    for (; k + 2 <= T; k += 2) {
      const double2 s0 = wpts[k], s1 = wpts[k + 1];
      const double z0 = s0.y - qz, z1 = s1.y - qz;
      const double zz0 = z0 * z0, zz1 = z1 * z1;      // (the column term is shared by the lane's pixels: same rounding)
#pragma unroll
      for (int h = 0; h < NPX; ++h) {
        const double y0 = s0.x - qy[h], y1 = s1.x - qy[h];
        const double d0 = y0 * y0 + zz0, d1 = y1 * y1 + zz1;   // two products, one sum, each rounded (contraction is off)
        tie[h] |= d0 == best[h]; if (d0 < best[h]) { best[h] = d0; bk[h] = k; }
        tie[h] |= d1 == best[h]; if (d1 < best[h]) { best[h] = d1; bk[h] = k + 1; }
      }
    }
    for (; k < T; ++k) {
      const double2 s = wpts[k];
      const double dz = s.y - qz;
      const double zz = dz * dz;
#pragma unroll
      for (int h = 0; h < NPX; ++h) {
        const double dy = s.x - qy[h];
        const double d = dy * dy + zz;
        tie[h] |= d == best[h]; if (d < best[h]) { best[h] = d; bk[h] = k; }
      }
    }
#pragma unroll
    for (int h = 0; h < NPX; ++h) {
      if (!wanted[h]) continue;
      nev += (unsigned)T;
      if (tie[h]) {
        nev += (unsigned)T;
        best[h] = dinf(); bk[h] = -1;
        for (k = 0; k < T; ++k) {
          const double2 s = wpts[k];
          const double dy = s.x - qy[h], dz = s.y - qz;
          const double d = dy * dy + dz * dz;
          if (d < best[h]) { best[h] = d; bk[h] = k; }
          else if (d == best[h] && bk[h] >= 0 && wid[k] < wid[bk[h]]) bk[h] = k;   // ties: the lowest original index
        }
      }
      if (wy0 == 0 && wy1 == G.ny - 1 && wz0 == 0 && wz1 == G.nz - 1) decided[h] = true;   // the window is the whole grid
      else {
        double B = dinf();
        if (wy1 < G.ny - 1) { const double t = G.by(wy1 + 1) - qy[h]; B = fmin(B, t * t); }
        if (wy0 > 0)        { const double t = G.by(wy0) - qy[h];     B = fmin(B, t * t); }
        if (wz1 < G.nz - 1) { const double t = G.bz(wz1 + 1) - qz; B = fmin(B, t * t); }
        if (wz0 > 0)        { const double t = G.bz(wz0) - qz;     B = fmin(B, t * t); }
        decided[h] = best[h] < B;
      }
      if (decided[h] && bk[h] >= 0) { const double2 pu = wpu[bk[h]]; P[h] = pu.x; U[h] = pu.y; }
    }
  }
#pragma unroll
  for (int h = 0; h < NPX; ++h) {
    if (!decided[h]) {
      double best;
      int bestk;
      ring_search(qy[h], qz, G, st, sp, si, best, bestk, nev);
      if (bestk >= 0) { const double2 pu = su[bestk]; P[h] = pu.x; U[h] = pu.y; }
    }
    if (P[h] < 0.0) P[h] = 0.0;                      // :182
  }
  if constexpr (COUNT) {
    unsigned t = nev;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0 && t) atomicAdd(evals, (unsigned long long)t);
  }
  // a row of the wave's pixels = up to 8 pixels x 5 columns = 40 consecutive doubles of the output table; the wave's two
  // groups of 8 rows leave one after the other through the window's LDS bytes
  double *stage = (double *)wpts;
  const int col0 = tx * TILE_W + wave * 8;
  const int cols = (W - col0) < 8 ? (W - col0) : 8;  // (<= 0: the whole wave is outside the image)
#pragma unroll
  for (int h = 0; h < NPX; ++h) {
    wave_sync();                                     // every lane is done with the window / the previous group's rows
    double *s5 = stage + lane * 5;                   // lane = row lr (8) x column (8) of the group's 8 x 8 pixels
    s5[0] = qy[h]; s5[1] = qz; s5[2] = P[h]; s5[3] = U[h]; s5[4] = P[h] * p_convert;
    wave_sync();
#pragma unroll
    for (int e = lane; e < 64 * 5; e += 64) {
      const int r = e / 40, k = e - r * 40;
      const int row = ty * TILE_H + h * 8 + r;
      if (row < H && k < cols * 5)
        out[(((int64_t)cs * H + row) * W + col0) * 5 + k] = stage[e];
    }
  }
}

}  // namespace cfd

using namespace cfd;

extern "C" int64_t cfd_resample_workspace_bytes(int ncase, int W, int S, int64_t total_points) {
  if (ncase <= 0 || W <= 0 || S <= 0 || total_points < 0) return -1;
  return rs_layout(ncase, W, S, total_points).bytes;
}

namespace cfd {
// shared by cfd_resample_nearest and its no-allocation variant
int resample_launch(const double *pts, const int64_t *case_offsets, int ncase, int64_t total, const double *y_final,
                    int H, const double *z_final, int W, const double *z_samples, int S, double tol1, double tol2,
                    double p_convert, double *out, void *workspace, hipStream_t st, unsigned long long *evals = nullptr) {
  CFD_REQUIRE(pts && case_offsets && y_final && z_final && z_samples && out && workspace, "NULL pointer");
  CFD_REQUIRE(ncase > 0 && ncase <= 65535 && H > 0 && W > 0 && S > 0 && S <= 2048, "bad sizes ncase=%d H=%d W=%d S=%d",
              ncase, H, W, S);
  CFD_REQUIRE((int64_t)H * W < (1LL << 31), "grid too large");
  CFD_REQUIRE(total >= 0 && total < (1LL << 31), "total_points %lld out of range", (long long)total);
  CFD_REQUIRE(tol1 >= 0.0 && tol2 >= 0.0, "negative tolerance");
  const RsLayout L = rs_layout(ncase, W, S, total);
  char *base = (char *)workspace;
  double *wall = (double *)(base + L.o_wall);
  unsigned long long *sacc = (unsigned long long *)(base + L.o_sacc);
  int *scnt = (int *)(base + L.o_scnt);
  unsigned long long *bbox = (unsigned long long *)(base + L.o_bbox);
  double *grid = (double *)(base + L.o_grid);
  int *start = (int *)(base + L.o_start);
  int *cnt = (int *)(base + L.o_cnt);
  int *cell = (int *)(base + L.o_cell);
  double2 *syz = (double2 *)(base + L.o_syz);
  double2 *spu = (double2 *)(base + L.o_spu);
  int *sidx = (int *)(base + L.o_sidx);
  const double ppc = tuning(CFD_TUNE_RESAMPLE_DENSITY) > 0 ? tuning(CFD_TUNE_RESAMPLE_DENSITY) / 4.0 : 1.5;
  const int pblocks = cdiv(total, 256);
  // one block per case does everything before the search when the average case is a block's sweep (<= 32 Ki points) and the
  // slice tables fit its LDS; cfd_set_tuning(CFD_TUNE_RESAMPLE_FUSED, 0) keeps the multi-launch path (tests run both)
  const size_t prep_lds = (size_t)10 * S * 8 + (size_t)2 * S * 4 + (size_t)(2 * PREP_LDS_CELLS + 1) * 4;
  const bool fused = tuning(CFD_TUNE_RESAMPLE_FUSED) != 0 && S <= 1024 && total / ncase <= 32768;
  if (fused) {
    if (int e_ = set_max_lds((const void *)case_prepare_kernel, (int)prep_lds)) return e_;
    hipLaunchKernelGGL(case_prepare_kernel, dim3(ncase), dim3(PREP_T), prep_lds, st, pts, case_offsets, z_samples, S, tol1,
                       tol2, z_final, W, wall, L.wall_stride, grid, start, cnt, cell, syz, spu, sidx, ppc);
    CFD_LAUNCH_CHECK();
  } else {
  const int64_t ninit = (int64_t)ncase * 4 * S;
  hipLaunchKernelGGL(resample_init_kernel, dim3(cdiv(ninit, 256)), dim3(256), 0, st, sacc, scnt, bbox, ncase, S);
  CFD_LAUNCH_CHECK();
  CFD_HIP(hipMemsetAsync(cnt, 0, (size_t)(L.o_cell - L.o_cnt), st));
  if (pblocks > 0) {
    // blocks per case from the AVERAGE case size (the per-case sizes live on the device): ~8 points per thread
    int64_t K = (total / ncase + 2047) / 2048;
    K = K < 1 ? 1 : (K > 256 ? 256 : K);
    const size_t lds = (size_t)4 * S * 8 + (size_t)((2 * S + 1) / 2) * 8 + 4 * 8;
    hipLaunchKernelGGL(points_pass_kernel, dim3((unsigned)K, ncase), dim3(256), lds, st, pts, case_offsets, z_samples, S,
                       tol1, tol2, sacc, scnt, bbox);
    CFD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(case_setup_kernel, dim3(ncase), dim3(256), 3 * S * sizeof(double), st, case_offsets, z_samples, S,
                     z_final, W, (const unsigned long long *)sacc, (const int *)scnt, (const unsigned long long *)bbox,
                     wall, L.wall_stride, grid, ppc);
  CFD_LAUNCH_CHECK();
  if (pblocks > 0) {
    hipLaunchKernelGGL(bin_count_kernel, dim3(pblocks), dim3(256), 0, st, pts, case_offsets, ncase, total,
                       (const double *)grid, cnt, cell);
    CFD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(bin_scan_kernel, dim3(ncase), dim3(256), 0, st, case_offsets, (const double *)grid, cnt, start);
  CFD_LAUNCH_CHECK();
  if (pblocks > 0) {
    hipLaunchKernelGGL(bin_scatter_kernel, dim3(pblocks), dim3(256), 0, st, pts, case_offsets, ncase, total,
                       (const int *)cell, (const int *)start, cnt, syz, spu, sidx);
    CFD_LAUNCH_CHECK();
  }
  }
  const int64_t tiles = (int64_t)cdiv(W, TILE_W) * cdiv(H, TILE_H);
  CFD_REQUIRE(tiles * ncase < (1LL << 31), "too many tiles in one call");
  if (evals)
    hipLaunchKernelGGL(nearest_kernel<true>, dim3((unsigned)(tiles * ncase)), dim3(256), 0, st, case_offsets, ncase, y_final,
                       H, z_final, W, S, (const double *)wall, L.wall_stride, (const double *)grid, (const int *)start,
                       (const double2 *)syz, (const double2 *)spu, (const int *)sidx, p_convert, out, evals);
  else
    hipLaunchKernelGGL(nearest_kernel<false>, dim3((unsigned)(tiles * ncase)), dim3(256), 0, st, case_offsets, ncase, y_final,
                       H, z_final, W, S, (const double *)wall, L.wall_stride, (const double *)grid, (const int *)start,
                       (const double2 *)syz, (const double2 *)spu, (const int *)sidx, p_convert, out, evals);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}
}  // namespace cfd

extern "C" int cfd_resample_nearest(const double *pts, const int64_t *case_offsets, int ncase, int64_t total_points,
                                    const double *y_final, int H, const double *z_final, int W,
                                    const double *z_samples, int S, double tol1, double tol2, double p_convert,
                                    double *out, void *workspace, void *stream) {
  return resample_launch(pts, case_offsets, ncase, total_points, y_final, H, z_final, W, z_samples, S, tol1, tol2,
                         p_convert, out, workspace, (hipStream_t)stream);
}

extern "C" int cfd_resample_nearest_counted(const double *pts, const int64_t *case_offsets, int ncase, int64_t total_points,
                                            const double *y_final, int H, const double *z_final, int W,
                                            const double *z_samples, int S, double tol1, double tol2, double p_convert,
                                            double *out, void *workspace, unsigned long long *n_evals, void *stream) {
  CFD_REQUIRE(n_evals, "n_evals is NULL");
  return resample_launch(pts, case_offsets, ncase, total_points, y_final, H, z_final, W, z_samples, S, tol1, tol2,
                         p_convert, out, workspace, (hipStream_t)stream, n_evals);
}

// ---------------------------------------------------------------------------------------------------------------
// The one entry point that allocates nothing (no counterpart in the reference).  For callers that cannot call
// hipMalloc: code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same kernels as
// cfd_resample_nearest; the workspace is this file-scope device array instead of a caller-supplied buffer.  Every GPU
// the library runs on holds its own copy (RS_SCRATCH_BYTES, until the process ends) at its own address, so the address
// is looked up once per device with that device current, and only that pointer -- never the array's name -- is handed
// to kernels and memsets.  The array keeps the last call's contents: resample_launch re-initialises every accumulator
// and counter it reads (resample_init_kernel, the memset of the cell counts), so nothing has to be cleared here.
// Calls that overlap on one GPU (two streams, two threads) share the array and RACE: serialise them.
// ---------------------------------------------------------------------------------------------------------------
#include <mutex>

namespace cfd {
constexpr size_t RS_SCRATCH_BYTES = 6u << 20;     // 6 MiB per GPU; e.g. 64 cases x 1024 points, 64x512 grid, 101 slices = 3.7 MiB
constexpr int RS_MAX_DEVICES = 64;
static __device__ __align__(256) unsigned char scratch[RS_SCRATCH_BYTES];
static void *g_scratch_ptr[RS_MAX_DEVICES];       // per device: device address of `scratch`, 0 = not looked up yet
static std::mutex g_scratch_mutex;

__global__ void copy_valid_slices_kernel(const double *__restrict__ wall, int64_t wall_stride, int S, int W, int ncase,
                                         int *__restrict__ out) {
  const int cs = blockIdx.x * blockDim.x + threadIdx.x;
  if (cs < ncase) out[cs] = *(const int *)(wall + (int64_t)cs * wall_stride + 2 * S + 2 * W);
}
}  // namespace cfd

extern "C" int64_t cfd_resample_noalloc_max_bytes(void) { return (int64_t)RS_SCRATCH_BYTES; }

extern "C" int cfd_resample_nearest_noalloc(const double *pts, const int64_t *case_offsets, int ncase,
                                            int64_t total_points, const double *y_final, int H, const double *z_final,
                                            int W, const double *z_samples, int S, double tol1, double tol2,
                                            double p_convert, double *out, int *valid_slices, void *stream) {
  CFD_REQUIRE(ncase > 0 && W > 0 && S > 0 && total_points >= 0, "bad sizes ncase=%d W=%d S=%d total_points=%lld", ncase,
              W, S, (long long)total_points);
  const int64_t need = rs_layout(ncase, W, S, total_points).bytes;
  CFD_REQUIRE(need <= (int64_t)RS_SCRATCH_BYTES,
              "cfd_resample_nearest_noalloc: this call needs %lld bytes of scratch, the built-in buffer holds %lld "
              "(ncase=%d W=%d S=%d total_points=%lld); use cfd_resample_nearest with a caller-supplied workspace",
              (long long)need, (long long)RS_SCRATCH_BYTES, ncase, W, S, (long long)total_points);
  int dev = -1;
  CFD_HIP(hipGetDevice(&dev));
  CFD_REQUIRE(dev >= 0 && dev < RS_MAX_DEVICES, "device index %d out of range", dev);
  void *ws = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
      void *p = nullptr;
      CFD_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));     // the current device's copy of the array
      CFD_REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null address");
      g_scratch_ptr[dev] = p;
    }
    ws = g_scratch_ptr[dev];
  }
  hipStream_t st = (hipStream_t)stream;
  const int rc = resample_launch(pts, case_offsets, ncase, total_points, y_final, H, z_final, W, z_samples, S, tol1, tol2,
                                 p_convert, out, ws, st);
  if (rc != CFD_OK) return rc;
  if (valid_slices) {
    const RsLayout L = rs_layout(ncase, W, S, total_points);
    hipLaunchKernelGGL(copy_valid_slices_kernel, dim3(cdiv(ncase, 64)), dim3(64), 0, st,
                       (const double *)((char *)ws + L.o_wall), L.wall_stride, S, W, ncase, valid_slices);
    CFD_LAUNCH_CHECK();
  }
  return CFD_OK;
}
