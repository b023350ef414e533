// Microbenchmark behind DESIGN.md section 5's Adam note: the same update as train.hip's adam_kernel in a few variants
// (grid size, non-temporal accesses, two vectors per thread, approximate reciprocal / sqrt), 128 Mi parameters.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/adam_bw.hip -o scripts/micro/adam_bw && scripts/micro/adam_bw
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NT, int U, int FAST>
__global__ __launch_bounds__(256) void adam(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                                            float *__restrict__ v, __bf16 *__restrict__ wc, int64_t n, float alpha,
                                            float b1, float b2, float eps, float gscale) {
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += stride * U) {
    f32x4 wv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        if (NT) {
          wv[u] = __builtin_nontemporal_load((f32x4 *)w + i); gv[u] = __builtin_nontemporal_load((const f32x4 *)g + i);
          mv[u] = __builtin_nontemporal_load((f32x4 *)m + i); vv[u] = __builtin_nontemporal_load((f32x4 *)v + i);
        } else {
          wv[u] = ((f32x4 *)w)[i]; gv[u] = ((const f32x4 *)g)[i]; mv[u] = ((f32x4 *)m)[i]; vv[u] = ((f32x4 *)v)[i];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i >= n4) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gk = gv[u][k] * gscale;
        mv[u][k] += (gk - mv[u][k]) * (1.f - b1);
        vv[u][k] += (gk * gk - vv[u][k]) * (1.f - b2);
        if (FAST) wv[u][k] -= (mv[u][k] * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv[u][k]) + eps);
        else wv[u][k] -= (mv[u][k] * alpha) / (sqrtf(vv[u][k]) + eps);
      }
      bf16x4 o = {(__bf16)wv[u][0], (__bf16)wv[u][1], (__bf16)wv[u][2], (__bf16)wv[u][3]};
      if (NT) {
        __builtin_nontemporal_store(wv[u], (f32x4 *)w + i); __builtin_nontemporal_store(mv[u], (f32x4 *)m + i);
        __builtin_nontemporal_store(vv[u], (f32x4 *)v + i); __builtin_nontemporal_store(o, (bf16x4 *)wc + i);
      } else {
        ((f32x4 *)w)[i] = wv[u]; ((f32x4 *)m)[i] = mv[u]; ((f32x4 *)v)[i] = vv[u]; ((bf16x4 *)wc)[i] = o;
      }
    }
  }
}

template <int NT, int U, int FAST>
static void run(const char *name, int blocks, float *w, float *g, float *m, float *v, __bf16 *wc, int64_t n) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((adam<NT, U, FAST>), dim3(blocks), dim3(256), 0, 0, w, g, m, v, wc, n, 1e-3f, .5f, .999f, 1e-7f, 1.f);
  CK(hipEventRecord(a));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((adam<NT, U, FAST>), dim3(blocks), dim3(256), 0, 0, w, g, m, v, wc, n, 1e-3f, .5f, .999f, 1e-7f, 1.f);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps;
  printf("%-28s blocks %6d: %8.1f us  %6.2f TB/s\n", name, blocks, us, n * 30.0 / us / 1e6);
}

int main() {
  const int64_t n = 128LL << 20;
  float *w, *g, *m, *v; __bf16 *wc;
  CK(hipMalloc(&w, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&wc, n * 2));
  CK(hipMemset(w, 0, n * 4)); CK(hipMemset(g, 0x3c, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
  for (int blocks : {8192, 131072}) {
    run<0, 1, 0>("plain", blocks, w, g, m, v, wc, n);
    run<1, 1, 0>("nontemporal", blocks, w, g, m, v, wc, n);
  }
  // one pass, no loop: blocks = n4 / (256 * U)
  run<0, 2, 0>("plain x2 full", 65536, w, g, m, v, wc, n);
  run<1, 2, 0>("nontemporal x2 full", 65536, w, g, m, v, wc, n);
  run<0, 4, 0>("plain x4 full", 32768, w, g, m, v, wc, n);
  run<1, 4, 0>("nontemporal x4 full", 32768, w, g, m, v, wc, n);
  run<1, 8, 0>("nontemporal x8 full", 16384, w, g, m, v, wc, n);
  run<1, 4, 1>("nontemporal x4 full approx", 32768, w, g, m, v, wc, n);
  // smaller ranges (the discriminator: 2.8 M; a DP shard: 16 M)
  for (int64_t nn : {(int64_t)(3 << 20), (int64_t)(16 << 20)}) {
    printf("n = %lld\n", (long long)nn);
    run<0, 1, 0>("plain 8192 cap", (int)((nn / 4 + 255) / 256 < 8192 ? (nn / 4 + 255) / 256 : 8192), w, g, m, v, wc, nn);
    run<1, 2, 0>("nontemporal x2 full", (int)((nn / 4 + 511) / 512), w, g, m, v, wc, nn);
    run<1, 4, 0>("nontemporal x4 full", (int)((nn / 4 + 1023) / 1024), w, g, m, v, wc, nn);
  }
  return 0;
}
