// Error reporting for the C ABI (thread-local message, never throws across the boundary).
#include "common.h"

#include <stdarg.h>

namespace clipfs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace clipfs

extern "C" int clipfs_abi_version(void) { return CLIPFS_ABI_VERSION; }
extern "C" const char* clipfs_last_error(void) { return clipfs::g_err; }
