// Shared host-side helpers of libnnmpc_hip.so.
#pragma once
#include <stdarg.h>
#include <stdio.h>

namespace nnmpc {
char* error_buffer();  // thread-local, 512 bytes
inline void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
}
}  // namespace nnmpc
