// 16x16x32 against 32x32x16 bf16 MFMAs on register operands (random values): does the instruction shape change what the chip
// sustains?  (32x32x16 reads half the operand registers per FLOP.)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/mfma_shape.hip -o scripts/micro/mfma_shape && scripts/micro/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void loop(const bf16x8 *__restrict__ src, float *__restrict__ out, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  float r = 0.f;
  if constexpr (SHAPE == 16) {
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = src[(tid * 12 + i) & 0xffff];
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
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][3];
  } else {
    // the same 128 x 64 wave tile as 4 x 2 tiles of 32 x 32: 4 A fragments, 2 B fragments, 8 accumulators of 16 registers
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = src[(tid * 12 + i) & 0xffff];
    for (int j = 0; j < 2; ++j) b[j] = src[(tid * 12 + 8 + j) & 0xffff];
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)   // two k-steps of 16 = the K = 32 of one 16x16x32 round
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) r += acc[i][j][0] + acc[i][j][15];
  }
  out[tid] = r;
}

int main() {
  const int blocks = 256, threads = 512, iters = 4096;
  const size_t n = 1 << 16;
  std::vector<unsigned short> h(n * 8);
  srand(1);
  for (auto &v : h) v = (unsigned short)(((rand() & 1) << 15) | ((120 + (rand() % 12)) << 7) | (rand() & 0x7f));
  bf16x8 *src; float *out;
  (void)hipMalloc(&src, n * 16); (void)hipMalloc(&out, (size_t)blocks * threads * 4);
  (void)hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const double flop = 2.0 * 128 * 64 * 32 * iters * (threads / 64) * blocks;   // both shapes: a 128 x 64 x 32 wave-tile round
  for (int round = 0; round < 2; ++round)
    for (int shape = 0; shape < 2; ++shape) {
      float sum = 0.f;
      for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0, 0);
        if (shape == 0) hipLaunchKernelGGL(loop<16>, dim3(blocks), dim3(threads), 0, 0, src, out, iters);
        else hipLaunchKernelGGL(loop<32>, dim3(blocks), dim3(threads), 0, 0, src, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3) sum += ms;
      }
      printf("%s: %.3f ms per launch, %.0f TFLOP/s\n", shape == 0 ? "16x16x32" : "32x32x16", sum / 5, flop / (sum / 5 * 1e-3) / 1e12);
    }
  return 0;
}
