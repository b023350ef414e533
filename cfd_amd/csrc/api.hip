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

extern "C" int cfd_version(void) { return 100; }
extern "C" const char *cfd_last_error(void) { return cfd::last_error_buf(); }
extern "C" int cfd_set_tuning(int key, int value) {
  CFD_REQUIRE(key >= 0 && key < CFD_TUNE_COUNT, "unknown tuning key %d", key);
  cfd::g_tuning[key] = value;
  return CFD_OK;
}
