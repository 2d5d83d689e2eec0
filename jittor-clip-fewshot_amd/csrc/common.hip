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

// Source stamps baked in by build.py (sha256 prefixes of the kernel sources this library was built from): lets a
// measurement file (profiles/*/gemm_traffic.json) state WHICH kernel it was taken on, and bench.py refuse a stale one.
#ifndef CLIPFS_SOURCE_STAMP
#define CLIPFS_SOURCE_STAMP "unstamped"
#endif
#ifndef CLIPFS_GEMM_STAMP
#define CLIPFS_GEMM_STAMP "unstamped"
#endif
extern "C" const char* clipfs_source_stamp(void) { return CLIPFS_SOURCE_STAMP; }
extern "C" const char* clipfs_gemm_source_stamp(void) { return CLIPFS_GEMM_STAMP; }
