// host_common.hpp -- error reporting shared by the translation units of libcarl_amd.so
#pragma once

#include <cstddef>

namespace carl_host {
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);  // formats into g_err, returns code
int check_launch(const char* what);        // hipGetLastError -> 0 / fail(...)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size) instead of on every launch
// (it is a driver call: ~2 us of the ~10 us a launch costs the host)
int ensure_dynamic_lds(const void* kernel, size_t bytes, const char* who);
}  // namespace carl_host
