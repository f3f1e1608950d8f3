// Stand-alone NT GEMM kernels built on tile_gemm.h:
//   C[M x N] = A[M x K] * B[N x K]'     (row-major, K contiguous in A and B)
// f32: MFMA 32x32x2 (exact f32), NB x NB tile per 256-thread workgroup, optional
//      fused epilogue  C = act(C + bias[n]).
// f64: VALU, 64 x 64 tile (used only for the O(n^2) iterative-refinement
//      residuals and q = tq x0; <1 % of the flops of a solve).
#pragma once
#include "tile_gemm.h"

namespace nnmpc {

template <int NB, bool RELU, bool BIAS>
__global__ __launch_bounds__(256) void gemm_nt_f32_k(float* __restrict__ C, size_t ldc,
                                                     const float* __restrict__ A, size_t lda,
                                                     const float* __restrict__ B, size_t ldb,
                                                     int K, const float* __restrict__ bias) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using Cf = TileCfg<NB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * NB, n0 = blockIdx.x * NB;
  f32x16 acc[Cf::MT][Cf::MT];
  zero_acc<NB>(acc);
  PlainOp a{A + (size_t)m0 * lda, lda};
  PlainOp b{B + (size_t)n0 * ldb, ldb};
  tile_gemm_nt<NB>(acc, a, b, K, lds, false);
#pragma unroll
  for (int mi = 0; mi < Cf::MT; ++mi)
#pragma unroll
    for (int mj = 0; mj < Cf::MT; ++mj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wr * Cf::WT + mi * 32 + acc_row(r, lane);
        const int col = n0 + wc * Cf::WT + mj * 32 + acc_col(lane);
        float v = acc[mi][mj][r];
        if (BIAS) v += bias[col];
        if (RELU) v = v > 0.f ? v : 0.f;
        C[(size_t)row * ldc + col] = v;
      }
}

// M, N multiples of 64, K multiple of 16.  rowphase (nullable): per-row tag, see below.
static __global__ __launch_bounds__(256) void gemm_nt_f64_k(double* __restrict__ C, size_t ldc,
                                                     const double* __restrict__ A, size_t lda,
                                                     const double* __restrict__ B, size_t ldb,
                                                     int K, const int* __restrict__ rowphase,
                                                     int want) {
  __shared__ double As[16][65];
  __shared__ double Bs[16][65];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  if (rowphase) {  // skip 64-row blocks in which no row (= solver slot) is in phase `want`
    const int need = tid < 64 ? (rowphase[m0 + tid] == want) : 0;
    if (!__syncthreads_or(need)) return;
  }
  double c[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) c[i][j] = 0.0;
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, row = e >> 4, k = e & 15;
      As[k][row] = A[(size_t)(m0 + row) * lda + k0 + k];
      Bs[k][row] = B[(size_t)(n0 + row) * ldb + k0 + k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      C[(size_t)(m0 + ty * 4 + i) * ldc + n0 + tx * 4 + j] = c[i][j];
}

}  // namespace nnmpc
