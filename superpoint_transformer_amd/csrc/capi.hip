// Error channel + version of the C ABI (include/spt_hip.h).
#include <stdarg.h>

#include "common.hpp"

namespace spt {

char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace spt

extern "C" int spt_version(void) { return 1000; }

extern "C" const char* spt_last_error(void) { return spt::last_error_buf(); }
