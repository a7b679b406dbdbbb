// host_common.hpp -- error reporting shared by the translation units of libcarl_amd.so
#pragma once

namespace carl_host {
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);  // formats into g_err, returns code
int check_launch(const char* what);        // hipGetLastError -> 0 / fail(...)
}  // namespace carl_host
