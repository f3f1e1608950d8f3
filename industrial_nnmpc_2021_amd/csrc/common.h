// Shared host-side helpers of libnnmpc_hip.so.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>

namespace nnmpc {
char* error_buffer();  // thread-local, 512 bytes
inline void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
}
// Wait for a stream by polling first: the blocking hipStreamSynchronize can take milliseconds to wake the calling
// thread on some hosts (measured: +5 ms per 12 ms NN forward), and the round loop of the active-set pass waits once
// per round.  Falls back to the blocking wait after ~0.2 s of polling.
inline hipError_t stream_sync(hipStream_t s) {
  for (int i = 0; i < 200000; ++i) {
    const hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e;
  }
  return hipStreamSynchronize(s);
}
}  // namespace nnmpc
// (library-internal, not part of the C ABI) the stream a regulator handle launches on: nnmpc_chain_run puts its own kernels of a
// lock-step step on it, so that a step needs no host wait between them and the solve
struct nnmpc_qp;
hipStream_t nnmpc_qp_stream_internal(nnmpc_qp* h);
