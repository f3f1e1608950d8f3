// The workgroup kernels of qp_wg.h (sets beyond 144 / 176 bounds: four or eight waves per problem, every tile in registers), ONE
// PER OBJECT: compiled five times with -DASM_WG_TU=0..4 (Makefile).  In a single translation unit with the rest of the solver they
// were eleven minutes of compile time on one core; the launches stay in qp_solver.hip, which declares the kernels (qp_wg.h).
//   0 asm_lambda_wg32_k   1 asm_lambda_wg64_k   2 asm_lambda_wg32b_k   3 asm_lambda_wg64r_k   4 asm_lambda_wg64s_k
#ifndef ASM_WG_TU
#error "compile with -DASM_WG_TU=<0..4>"
#endif
#include "qp_asm.h"
