/*
 * CPU restatement of skfmm.distance(phi, dx) -- scikit-fmm's signed-distance fast marching -- for a 2-D grid.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).  The product path is
 * cfd_dataset_grids on the GPU (cfd_amd/csrc/dataset.hip), an independent implementation of the same scheme.
 *
 * Where the algorithm lives: the reference calls `skfmm.distance(phi, d_x)` (/root/reference/utils.py:100-103,
 * P/test.py:69-71, U/test.py).  scikit-fmm is a third-party dependency, version UNPINNED in the reference (no
 * requirements file) and NOT installed here, so this file restates its published algorithm (skfmm/pfmm.py `distance`,
 * skfmm/base_marcher.cpp, skfmm/distance_marcher.cpp, skfmm/heap.cpp, as of the 2022-2024 releases) from the source
 * as the author of this file knows it.  PARITY UNPINNED: nothing in the reference or in this image can confirm it.
 * Call-site parameters: order = 2 (skfmm's default), narrow = 0, no mask, not periodic, dx equal along both axes.
 *
 * The scheme (names as in scikit-fmm):
 *   initalizeFrozen   cells with phi == 0 are frozen at 0.  A cell whose 4-neighbourhood crosses the zero level set
 *                     (phi_i * phi_n < 0) is frozen at the distance obtained by LINEAR INTERPOLATION along each axis,
 *                     d_axis = dx * phi_i / (phi_i - phi_n) (the smaller of the two directions), combined over the
 *                     axes harmonically: d = sqrt(1 / sum(1 / d_axis^2)), with the sign of phi_i.
 *                     For phi = +-1 that is dx/2 beside a straight wall and dx/(2*sqrt(2)) in a corner.
 *   initalizeNarrow   every Far 4-neighbour of a frozen cell becomes Narrow, gets its tentative distance
 *                     (updatePoint) and enters the heap with key |d|; cells are visited in address order.
 *   solve             pop the smallest |d|, freeze it, re-evaluate its non-frozen 4-neighbours (Narrow: heap key
 *                     updated; Far: pushed), and -- second order -- the Narrow cell two steps away when the cell in
 *                     between is frozen.  A tentative value of exactly 0 is ignored (`if (d)`).
 *   updatePointOrderTwo   per axis, value1 = the frozen neighbour of smallest |distance|; value2 = the cell behind
 *                     it if that is frozen and not further from the contour than value1 (value2 is NOT reset when the
 *                     second direction replaces value1 without a valid cell behind it: the published code has no
 *                     else-branch there; kept, see SKFMM_RESET_VALUE2).  Second-order term
 *                     tp = (4*value1 - value2)/3 with weight 9/4, else first-order term.  Quadratic a d^2 + b d +
 *                     (c - 1) = 0; root (+) where phi > epsilon, (-) otherwise.  A negative discriminant (solveQuadratic
 *                     throws "Negative discriminant in distance marcher quadratic.") is caught by
 *                     updatePointOrderTwo, which then returns the position of the parabola's minimum, -b / (2 a)
 *                     ("a more reasonable approximation"): speckled masks -- single solid cells inside the fluid, as a
 *                     noisy p field produces -- do reach that branch.
 * Heap order: scikit-fmm's binary heap breaks ties between equal keys by its sift history; this restatement and the
 * GPU kernel both pop in the total order (key, address), so the two agree bit for bit.  (In exact arithmetic the order
 * among equal keys does not change the result: a neighbour whose value equals the cell's own contributes nothing.)
 * Arithmetic: every expression is written in scikit-fmm's operand order; pow(x, 2) is x*x (what gcc emits for it);
 * compile WITHOUT floating-point contraction (-ffp-contract=off), as x86-64 wheels of scikit-fmm run (no FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <float.h>

#ifndef SKFMM_RESET_VALUE2
#define SKFMM_RESET_VALUE2 0
#endif

enum { FAR = 0, NARROW = 1, FROZEN = 2 };

typedef struct {
  int H, W;
  const double *phi;
  double *dist;
  unsigned char *flag;
  double dx, idx2;
  /* heap of narrow cells: position p holds cell hcell[p] with key hkey[p]; hpos[cell] = p */
  int *hcell, *hpos, hn;
  double *hkey;
  int error;
} Fmm;

/* neighbour of `i` along `dim` (0 = rows, 1 = columns) at offset `step`, or -1 outside the grid */
static int nb(const Fmm *m, int i, int dim, int step) {
  const int r = i / m->W, c = i - r * m->W;
  if (dim == 0) {
    const int rr = r + step;
    return (rr < 0 || rr >= m->H) ? -1 : rr * m->W + c;
  }
  const int cc = c + step;
  return (cc < 0 || cc >= m->W) ? -1 : r * m->W + cc;
}

/* ---- heap with the total order (key, cell) ---------------------------------------------------------------- */
static int hless(const Fmm *m, int p, int q) {
  return m->hkey[p] < m->hkey[q] || (m->hkey[p] == m->hkey[q] && m->hcell[p] < m->hcell[q]);
}
static void hswap(Fmm *m, int p, int q) {
  const int cp = m->hcell[p], cq = m->hcell[q];
  const double kp = m->hkey[p];
  m->hcell[p] = cq; m->hkey[p] = m->hkey[q]; m->hpos[cq] = p;
  m->hcell[q] = cp; m->hkey[q] = kp; m->hpos[cp] = q;
}
static void hup(Fmm *m, int p) {
  while (p > 0) {
    const int par = (p - 1) / 2;
    if (!hless(m, p, par)) break;
    hswap(m, p, par);
    p = par;
  }
}
static void hdown(Fmm *m, int p) {
  for (;;) {
    const int l = 2 * p + 1, r = l + 1;
    int s = p;
    if (l < m->hn && hless(m, l, s)) s = l;
    if (r < m->hn && hless(m, r, s)) s = r;
    if (s == p) break;
    hswap(m, p, s);
    p = s;
  }
}
static void hpush(Fmm *m, int cell, double key) {
  const int p = m->hn++;
  m->hcell[p] = cell; m->hkey[p] = key; m->hpos[cell] = p;
  hup(m, p);
}
static void hset(Fmm *m, int cell, double key) {
  const int p = m->hpos[cell];
  m->hkey[p] = key;
  hup(m, p);
  hdown(m, m->hpos[cell]);
}
static int hpop(Fmm *m) {
  const int cell = m->hcell[0];
  m->hn--;
  if (m->hn > 0) {
    m->hcell[0] = m->hcell[m->hn]; m->hkey[0] = m->hkey[m->hn]; m->hpos[m->hcell[0]] = 0;
    hdown(m, 0);
  }
  return cell;
}

/* distance_marcher.cpp: updatePointOrderTwo + solveQuadratic */
static double update_point(Fmm *m, int i) {
  const double aa = 9.0 / 4.0, oneThird = 1.0 / 3.0;
  double a = 0, b = 0, c = 0;
  for (int dim = 0; dim < 2; dim++) {
    double value1 = DBL_MAX, value2 = DBL_MAX;
    for (int j = -1; j < 2; j += 2) {
      const int n1 = nb(m, i, dim, j);
      if (n1 != -1 && m->flag[n1] == FROZEN) {
        if (fabs(m->dist[n1]) < fabs(value1)) {
          value1 = m->dist[n1];
          const int n2 = nb(m, i, dim, j * 2);
          if (n2 != -1 && m->flag[n2] == FROZEN &&
              ((m->dist[n2] <= value1 && value1 >= 0) || (m->dist[n2] >= value1 && value1 <= 0))) {
            value2 = m->dist[n2];
          }
#if SKFMM_RESET_VALUE2
          else value2 = DBL_MAX;
#endif
        }
      }
    }
    if (value2 < DBL_MAX) {
      const double tp = oneThird * (4 * value1 - value2);
      a += m->idx2 * aa;
      b -= m->idx2 * 2 * aa * tp;
      c += m->idx2 * aa * (tp * tp);
    } else if (value1 < DBL_MAX) {
      a += m->idx2;
      b -= m->idx2 * 2 * value1;
      c += m->idx2 * (value1 * value1);
    }
  }
  c -= 1;
  const double det = b * b - 4 * a * c;
  if (det >= 0) {
    if (m->phi[i] > DBL_EPSILON) return (-b + sqrt(det)) / 2.0 / a;
    return (-b - sqrt(det)) / 2.0 / a;
  }
  return -b / (2.0 * a); /* solveQuadratic throws; updatePointOrderTwo catches and returns the parabola's minimum */
}

/*
 * out[H*W] <- signed distance of phi[H*W] (row-major), cell size dx.
 * Returns 0, 2 = no zero contour (pfmm.py raises ValueError), 1 = no memory.
 */
int sdf_fmm_distance(const double *phi, int H, int W, double dx, double *out) {
  const int n = H * W;
  Fmm m;
  m.H = H; m.W = W; m.phi = phi; m.dist = out; m.dx = dx; m.idx2 = 1 / dx / dx; m.hn = 0; m.error = 0;
  m.flag = (unsigned char *)calloc((size_t)n, 1);
  m.hcell = (int *)malloc((size_t)n * sizeof(int));
  m.hpos = (int *)malloc((size_t)n * sizeof(int));
  m.hkey = (double *)malloc((size_t)n * sizeof(double));
  if (!m.flag || !m.hcell || !m.hpos || !m.hkey) { free(m.flag); free(m.hcell); free(m.hpos); free(m.hkey); return 1; }
  for (int i = 0; i < n; i++) out[i] = 0.0;
  /* ---- initalizeFrozen */
  int frozen = 0;
  for (int i = 0; i < n; i++)
    if (phi[i] == 0.0) { m.flag[i] = FROZEN; out[i] = 0.0; frozen++; }
  for (int i = 0; i < n; i++) {
    if (m.flag[i] != FAR) continue;
    double ld[2];
    int borders = 0;
    for (int dim = 0; dim < 2; dim++) {
      ld[dim] = 0;
      for (int j = -1; j < 2; j += 2) {
        const int na = nb(&m, i, dim, j);
        if (na != -1 && phi[i] * phi[na] < 0) {
          borders = 1;
          const double d = dx * phi[i] / (phi[i] - phi[na]);
          if (ld[dim] == 0 || ld[dim] > d) ld[dim] = d;
        }
      }
    }
    if (borders) {
      double dsum = 0;
      for (int dim = 0; dim < 2; dim++)
        if (ld[dim] > 0) dsum += 1 / ld[dim] / ld[dim];
      out[i] = phi[i] < 0 ? -sqrt(1 / dsum) : sqrt(1 / dsum);
      m.flag[i] = FROZEN;
      frozen++;
    }
  }
  if (!frozen) { free(m.flag); free(m.hcell); free(m.hpos); free(m.hkey); return 2; }
  /* ---- initalizeNarrow */
  for (int i = 0; i < n; i++) {
    if (m.flag[i] != FROZEN) continue;
    for (int dim = 0; dim < 2; dim++)
      for (int j = -1; j < 2; j += 2) {
        const int na = nb(&m, i, dim, j);
        if (na != -1 && m.flag[na] == FAR) {
          m.flag[na] = NARROW;
          const double d = update_point(&m, na);
          out[na] = d;
          hpush(&m, na, fabs(d));
        }
      }
  }
  /* ---- solve */
  while (m.hn > 0 && !m.error) {
    const int addr = hpop(&m);
    m.flag[addr] = FROZEN;
    for (int dim = 0; dim < 2; dim++)
      for (int j = -1; j < 2; j += 2) {
        const int na = nb(&m, addr, dim, j);
        if (na != -1 && m.flag[na] != FROZEN) {
          const double d = update_point(&m, na);
          if (d) {
            out[na] = d;
            if (m.flag[na] == NARROW) hset(&m, na, fabs(d));
            else { m.flag[na] = NARROW; hpush(&m, na, fabs(d)); }
          }
        }
        /* second order: "jump" over a frozen neighbour to the narrow cell behind it */
        if (na != -1 && m.flag[na] == FROZEN) {
          const int n2 = nb(&m, addr, dim, j * 2);
          if (n2 != -1 && m.flag[n2] == NARROW) {
            const double d = update_point(&m, n2);
            if (d) { hset(&m, n2, fabs(d)); out[n2] = d; }
          }
        }
      }
  }
  const int err = m.error;
  free(m.flag); free(m.hcell); free(m.hpos); free(m.hkey);
  return err;
}

/* batch form for the bench's CPU baseline and the tests: ncase grids, returns the first non-zero status */
int sdf_fmm_distance_batch(const double *phi, int ncase, int H, int W, double dx, double *out) {
  for (int c = 0; c < ncase; c++) {
    const int rc = sdf_fmm_distance(phi + (size_t)c * H * W, H, W, dx, out + (size_t)c * H * W);
    if (rc) return rc;
  }
  return 0;
}
