// njode_error.h -- the thread-local message njode_last_error() returns is owned by
// njode_api.hip; other translation units of the library record theirs through this.
#pragma once
#include <stdarg.h>

namespace njode {
int set_error_v(int code, const char* fmt, va_list ap);
}
