#include "common.h"
#include "../../include/nnmpc.h"

namespace nnmpc {
char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}
}  // namespace nnmpc

extern "C" const char* nnmpc_last_error(void) { return nnmpc::error_buffer(); }
