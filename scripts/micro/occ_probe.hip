// Register / scratch footprint of single-class instances of the register-resident multiplier kernel at other occupancies
// (compile only: hipcc ... --cuda-device-only -S, then read .vgpr_count / .private_segment_fixed_size).
#include <hip/hip_runtime.h>
#include "qp_asm.h"
using namespace nnmpc;
#ifndef PROBE_MB
#define PROBE_MB 8
#endif
#ifndef PROBE_OCC
#define PROBE_OCC 3
#endif
extern "C" __global__ __launch_bounds__(256, PROBE_OCC) void probe_k(AsmDev d) {
  asm_lambda_reg<float, PROBE_MB, 4>(d, ASM_NBIN + PROBE_MB - 4, blockIdx.x);
}
