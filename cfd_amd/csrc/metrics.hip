// Evaluation metrics of the valid / test drivers, on the device (U/valid.py:86-111, P/valid.py, P/test.py):
// per case  y_error = |Y - y|,  its maximum, the percentage error at the arg-max pixel, its mean, the mean row-wise
// cosine similarity over the rows whose target is not all zero, and the per-column maximum (the "Error_mean" surface,
// U/valid.py:100,119).  One block per case; each case is 2 x H x W fp32 = 256 KiB at the reference shape, read once
// from HBM (the column pass re-reads it from L2).  HBM-bound: algorithmic bytes = 8 * H * W per case.
#include "common.h"

namespace cfd {

struct ArgMax {
  float v;
  int i;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {
  // numpy.argmax / ndarray.max: first occurrence of the maximum in row-major order, and a NaN IS the maximum (the
  // first NaN wins): a diverged model's all-NaN prediction then reports NaN, as the reference would, instead of
  // leaving the index at its start value
  const bool an = a.v != a.v, bn = b.v != b.v;
  if (an || bn) return (bn && (!an || b.i < a.i)) ? b : a;
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

__global__ __launch_bounds__(256) void eval_metrics_kernel(const float *__restrict__ pred, float scale,
                                                           const float *__restrict__ target, int H, int W,
                                                           double *__restrict__ out, float *__restrict__ colmax) {
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *y = pred + (size_t)c * H * W, *Y = target + (size_t)c * H * W;
  double esum = 0.0, cos_sum = 0.0;
  int cos_n = 0;
  ArgMax am = {-1.f, 0};      // every |Y - y| is >= 0 or NaN, so the start value never survives; index 0 keeps Y[am.i] in bounds regardless
  for (int r = wave; r < H; r += 4) {
    double dot = 0.0, na = 0.0, nb = 0.0;
    int nz = 0;
    for (int j = lane; j < W; j += 64) {
      const float a = y[r * W + j] * scale, b = Y[r * W + j];
      const float e = fabsf(b - a);
      esum += (double)e;
      am = better(am, ArgMax{e, r * W + j});
      dot += (double)a * (double)b;
      na += (double)a * (double)a;
      nb += (double)b * (double)b;
      nz |= (b != 0.f);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      dot += __shfl_xor(dot, o);
      na += __shfl_xor(na, o);
      nb += __shfl_xor(nb, o);
      nz |= __shfl_xor(nz, o);
    }
    if (nz) {   // `if not np.all(Y[i] == 0)` (U/valid.py:109); sklearn leaves a zero-norm row at zero -> cosine 0
      cos_sum += (na > 0.0 && nb > 0.0) ? dot / (sqrt(na) * sqrt(nb)) : 0.0;
      cos_n += 1;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    esum += __shfl_xor(esum, o);
    ArgMax t = {__shfl_xor(am.v, o), __shfl_xor(am.i, o)};
    am = better(am, t);
  }
  __shared__ double s_e[4], s_c[4];
  __shared__ int s_n[4], s_i[4];
  __shared__ float s_v[4];
  if (lane == 0) { s_e[wave] = esum; s_c[wave] = cos_sum; s_n[wave] = cos_n; s_v[wave] = am.v; s_i[wave] = am.i; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      esum += s_e[w]; cos_sum += s_c[w]; cos_n += s_n[w];
      am = better(am, ArgMax{s_v[w], s_i[w]});
    }
    double *o = out + (size_t)c * 4;
    o[0] = (double)am.v;
    o[1] = (double)(am.v / Y[am.i] * 100.f);          // fp32 arithmetic as numpy does on float32 arrays (:105)
    o[2] = esum / ((double)H * (double)W);
    o[3] = cos_n ? cos_sum / cos_n : nan("");
  }
  if (colmax)
    for (int j = tid; j < W; j += 256) {
      float m = 0.f;
      for (int r = 0; r < H; ++r) {
        const float e = fabsf(Y[r * W + j] - y[r * W + j] * scale);
        m = (e > m || e != e) ? e : m;            // ndarray.max(axis=0) propagates NaN (fmaxf would drop it)
      }
      colmax[(size_t)c * W + j] = m;
    }
}

}  // namespace cfd

using namespace cfd;

extern "C" int cfd_eval_metrics(const float *pred, float scale, const float *target, int ncase, int H, int W, double *out,
                                float *colmax, void *stream) {
  CFD_REQUIRE(ncase >= 0 && H > 0 && W > 0 && (int64_t)H * W < (1LL << 31), "bad shape");
  if (ncase == 0) return CFD_OK;
  CFD_REQUIRE(pred && target && out, "NULL pointer");
  hipLaunchKernelGGL(eval_metrics_kernel, dim3(ncase), dim3(256), 0, (hipStream_t)stream, pred, scale, target, H, W, out,
                     colmax);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}
