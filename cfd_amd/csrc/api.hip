// Library-level entry points of libcfd_amd.so.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

namespace cfd {
int set_max_lds(const void *fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, int> done;   // (kernel, device) -> bytes set
  int dev = 0;
  CFD_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({fn, dev});
  if (it != done.end() && it->second >= bytes) return CFD_OK;
  CFD_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done[{fn, dev}] = bytes;
  return CFD_OK;
}

char *last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

// tuning knobs (process-wide; -1 = built-in heuristic)
static int g_tuning[CFD_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
int tuning(int key) { return (key >= 0 && key < CFD_TUNE_COUNT) ? g_tuning[key] : -1; }
}  // namespace cfd

namespace cfd {
// see cfd_mfma_probe in include/cfd_amd.h
__global__ __launch_bounds__(512, 2) void mfma_probe_kernel(const bf16x8 *__restrict__ src, float *__restrict__ out, int iters) {
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
}  // namespace cfd

extern "C" int cfd_mfma_probe(const void *operands, float *out, int iters, void *stream) {
  CFD_REQUIRE(operands && out && iters > 0 && iters <= (1 << 20), "bad arguments");
  CFD_REQUIRE(((uintptr_t)operands & 15) == 0, "operands must be 16-byte aligned");
  hipLaunchKernelGGL(cfd::mfma_probe_kernel, dim3(256), dim3(512), 0, (hipStream_t)stream, (const cfd::bf16x8 *)operands, out,
                     iters);
  CFD_LAUNCH_CHECK();
  return CFD_OK;
}

extern "C" int cfd_version(void) { return 100; }
extern "C" const char *cfd_last_error(void) { return cfd::last_error_buf(); }
extern "C" int cfd_set_tuning(int key, int value) {
  CFD_REQUIRE(key >= 0 && key < CFD_TUNE_COUNT, "unknown tuning key %d", key);
  cfd::g_tuning[key] = value;
  return CFD_OK;
}
