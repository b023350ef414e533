// Library-level entry points of libcfd_amd.so.
#include "common.h"

namespace cfd {
char *last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace cfd

extern "C" int cfd_version(void) { return 100; }
extern "C" const char *cfd_last_error(void) { return cfd::last_error_buf(); }
