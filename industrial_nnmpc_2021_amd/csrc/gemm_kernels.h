// Stand-alone NT GEMM kernels built on tile_gemm.h:
//   C[M x N] = A[M x K] * B[N x K]'     (row-major, K contiguous in A and B)
// f32: MFMA 32x32x2 (exact f32), NB x NB tile per 256-thread workgroup, optional
//      fused epilogue  C = act(C + bias[n]).
// f64: MFMA 16x16x4 f64, 64 x 64 tile (PCG residuals P v and q = tq x0; <1 % of the flops of a solve).
#pragma once
#include "tile_gemm.h"
#include "gemm64.h"

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
  // Workgroups go to the 8 XCDs (each with its own L2) round-robin by linear id: remap id -> tile so that the
  // gridDim.x column tiles of one row panel run back to back on ONE XCD (the A panel then comes from HBM once and
  // from that L2 for the other tiles).  A bijection of the grid; the tail (gridDim.y % 8 panels) keeps the plain order.
  int tm, tn;
  {
    const int ntn = gridDim.x, ntm = gridDim.y, bid = blockIdx.x + ntn * blockIdx.y, full = (ntm >> 3) * 8 * ntn;
    if (bid < full) { const int sq = bid >> 3; tm = (sq / ntn) * 8 + (bid & 7); tn = sq % ntn; }
    else { const int rem = bid - full; tm = (ntm >> 3) * 8 + rem / ntn; tn = rem % ntn; }
  }
  const int m0 = tm * NB, n0 = tn * NB;
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

// Plain f32 NT GEMM whose k-loop stops at a device-side bound (kdyn: index of the last non-zero column of A, per
// launch or per row block):
// XH32 = LAM32 * Pinv32 of the f32 active-set rounds.
template <int NB>
__global__ __launch_bounds__(256) void gemm_nt_f32_kdyn_k(float* __restrict__ C, size_t ldc,
                                                          const float* __restrict__ A, size_t lda,
                                                          const float* __restrict__ B, size_t ldb,
                                                          int K, const int* __restrict__ kdyn, int kper) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using Cf = TileCfg<NB>;
  // 1-D grid of ntm * ntn workgroups (gridDim.y = ntm carries the row-tile count, gridDim.x = ntn * ... see the launch):
  // workgroups go to the 8 XCDs round-robin by id, so the column tiles of ONE row panel get ids that are equal mod 8 --
  // the panel of LAM32 comes from HBM once and from that XCD's L2 for the other column tiles (a plain (x = column tile)
  // grid spread them over four XCDs: 3.2 GB of the step's HBM traffic for 0.8 GB of operands).
  int tm, tn;
  {
    const int ntn = gridDim.x, ntm = gridDim.y, bid = blockIdx.x + ntn * blockIdx.y, full = (ntm >> 3) * 8 * ntn;
    if (bid < full) { const int sq = bid >> 3; tm = (sq / ntn) * 8 + (bid & 7); tn = sq % ntn; }
    else { const int rem = bid - full; tm = (ntm >> 3) * 8 + rem / ntn; tn = rem % ntn; }
  }
  {
    // kper = 0: one bound for the launch; else kper consecutive entries per row block (64-row granularity)
    int kl = kdyn[kper * tm];
    for (int i = 1; i < kper; ++i) kl = max(kl, kdyn[kper * tm + i]);
    K = min(K, ((kl + KC) / KC) * KC);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = tm * NB, n0 = tn * NB;
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
        C[(size_t)row * ldc + col] = acc[mi][mj][r];
      }
}

// f64 NT GEMM on v_mfma_f64_16x16x4_f64.  M, N multiples of 64, K multiple of 16.
// 256 threads = 2 x 2 waves, each wave a 32 x 32 block = 2 x 2 MFMA tiles (4 f64 results per
// lane and tile; f64 C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg).  K-chunks of 16
// staged through LDS as [64 rows][16 k] with row stride 18 doubles (conflict-free ds_read_b64:
// lane (i = l & 15, kq = l >> 4) reads k = 4 s + kq of row i).
// rowphase (nullable): per-row tag; 64-row blocks in which no row has tag `want` are skipped.
static __global__ __launch_bounds__(256) void gemm_nt_f64_k(double* __restrict__ C, size_t ldc,
                                                     const double* __restrict__ A, size_t lda,
                                                     const double* __restrict__ B, size_t ldb,
                                                     int K, const int* __restrict__ rowphase,
                                                     int want, const int* __restrict__ kdyn = nullptr,
                                                     const int* __restrict__ mdyn = nullptr, int kper = 0) {
  constexpr int LD = 18;
  // kdyn (nullable): device-side bound on the non-zero columns of A (last non-zero column index);
  // the k-loop stops there -- LAM rows of the active-set pass are zero beyond the last active bound.
  // kper = 0: one bound for the launch; else kper consecutive entries per row block (64-row granularity)
  if (kdyn) {
    int kl = kdyn[kper * blockIdx.y];
    for (int i = 1; i < kper; ++i) kl = max(kl, kdyn[kper * blockIdx.y + i]);
    K = min(K, ((kl + 16) / 16) * 16);
  }
  __shared__ __attribute__((aligned(16))) double As[2][64 * LD];
  __shared__ __attribute__((aligned(16))) double Bs[2][64 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  if (mdyn && m0 >= *mdyn) return;                          // device-side row count (rows beyond it are not needed)
  if (rowphase) {
    const int need = tid < 64 ? (rowphase[m0 + tid] == want) : 0;
    if (!__syncthreads_or(need)) return;
  }
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  // staging: 64 rows x 16 doubles = 512 double2 per operand -> 2 per thread
  const int lrow0 = tid >> 3, lc = (tid & 7) * 2;           // rows lrow0, lrow0 + 32
  const double* Ag = A + (size_t)m0 * lda;
  const double* Bg = B + (size_t)n0 * ldb;
  f64x2 ra[2], rb[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    ra[h] = *reinterpret_cast<const f64x2*>(Ag + (size_t)(lrow0 + 32 * h) * lda + lc);
    rb[h] = *reinterpret_cast<const f64x2*>(Bg + (size_t)(lrow0 + 32 * h) * ldb + lc);
  }
  const int li = lane & 15, kq = lane >> 4;
  const int nk = K / 16;
  for (int kc = 0; kc < nk; ++kc) {
    double* sA = As[kc & 1];
    double* sB = Bs[kc & 1];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<f64x2*>(sA + (lrow0 + 32 * h) * LD + lc) = ra[h];
      *reinterpret_cast<f64x2*>(sB + (lrow0 + 32 * h) * LD + lc) = rb[h];
    }
    __syncthreads();
    if (kc + 1 < nk) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ra[h] = *reinterpret_cast<const f64x2*>(Ag + (size_t)(lrow0 + 32 * h) * lda + (kc + 1) * 16 + lc);
        rb[h] = *reinterpret_cast<const f64x2*>(Bg + (size_t)(lrow0 + 32 * h) * ldb + (kc + 1) * 16 + lc);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = sA[(wr * 32 + t * 16 + li) * LD + 4 * s + kq];
        b[t] = sB[(wc * 32 + t * 16 + li) * LD + 4 * s + kq];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 32 + i * 16 + (lane >> 4) + 4 * r;
        const int col = n0 + wc * 32 + j * 16 + (lane & 15);
        C[(size_t)row * ldc + col] = acc[i][j][r];
      }
}

// (The 128 x 128 tile of rounds 1-2 -- register staging, padded LDS rows, ds_read_b64 fragments -- lives on in
// scripts/micro/gemm64_micro.hip as the A/B reference of gemm64.h's tile.)

}  // namespace nnmpc
