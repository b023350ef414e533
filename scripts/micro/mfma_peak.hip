// What the matrix cores sustain with NOTHING else in the loop (no LDS, no memory): bf16 16x16x32 MFMAs on register operands,
// 32 independent accumulator tiles per wave, 8 waves per CU -- with random operands and with zeros.  The 2.5 PFLOP/s
// figure the rooflines are priced against assumes 2.4 GHz; this prints what the chip holds under the same instruction
// stream the conv kernels' MFMA clusters issue.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_peak.hip -o scripts/micro/mfma_peak && scripts/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512, 2) void mfma_loop(const bf16x8 *__restrict__ src, float *__restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  bf16x8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = src[(tid * 12 + i) & 0xffff];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = src[(tid * 12 + 8 + j) & 0xffff];
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[tid] = s[0] + s[1] + s[2] + s[3];
}

int main() {
  const int blocks = 256, threads = 512, iters = 4096;
  const size_t n = 1 << 16;
  std::vector<unsigned short> h(n * 8);
  bf16x8 *src;
  float *out;
  hipMalloc(&src, n * 16);
  hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    srand(1);
    for (auto &v : h) {
      // random: a bf16 with random sign / mantissa and an exponent near 1 (what activations and weights look like); zeros
      v = mode == 0 ? (unsigned short)(((rand() & 1) << 15) | ((120 + (rand() % 12)) << 7) | (rand() & 0x7f)) : 0;
    }
    hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, src, out, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
      sum += ms;
    }
    const double flop = 2.0 * 16 * 16 * 32 * 32.0 * iters * (threads / 64) * blocks;
    printf("%s operands: %.3f ms per launch (best %.3f): %.0f TFLOP/s sustained = %.3f of 2500 (best %.0f)\n",
           mode == 0 ? "random" : "zero  ", sum / reps, best, flop / (sum / reps * 1e-3) / 1e12,
           flop / (sum / reps * 1e-3) / 1e12 / 2500.0, flop / (best * 1e-3) / 1e12);
  }
  return 0;
}
