// bf16 NT tile GEMM core for gfx950 (structured-NN forward):
//   acc[r][c] += sum_k A[r][k] * B[c][k],  A, B bf16 row-major (K contiguous), f32 accumulate,
// v_mfma_f32_32x32x16_bf16, 256 threads = 2x2 waves per NB x NB tile, K-chunks of 64.
// LDS image per operand/chunk: [NB rows][64 bf16], row stride 72 bf16 = 144 B = 36 dwords: the
// same stride as the f32 core, so a lane's ds_read_b128 (its 8 consecutive k of one MFMA
// step) is conflict-free inside every 16-lane group.  Lane l feeds step s with
// k = 16 s + 8 (l >> 5) + (0..7) for both operands.
#pragma once
#include <hip/hip_runtime.h>
#include "tile_gemm.h"

namespace nnmpc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short bf16raw;

constexpr int KC16 = 64;
constexpr int LDS_LD16 = 72;

template <int NB>
struct TileCfg16 {
  static constexpr int WT = NB / 2;
  static constexpr int MT = WT / 32;
  static constexpr int LD4 = NB * (KC16 / 8) / 256;          // 16-byte loads / thread / operand / chunk
  static constexpr int STAGE_ELEMS = NB * LDS_LD16;
  static constexpr int LDS_BYTES = 4 * STAGE_ELEMS * 2;       // A,B x double buffer
};

template <int NB>
__device__ __forceinline__ void mma_chunk16(f32x16 (&acc)[TileCfg16<NB>::MT][TileCfg16<NB>::MT],
                                            const bf16raw* __restrict__ sA, const bf16raw* __restrict__ sB,
                                            int wr, int wc, int lane) {
  constexpr int MT = TileCfg16<NB>::MT, WT = TileCfg16<NB>::WT;
  const int lr = lane & 31, kh = (lane >> 5) * 8;
  bf16x8 a[MT][4], b[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a[m][s] = *reinterpret_cast<const bf16x8*>(sA + (wr * WT + m * 32 + lr) * LDS_LD16 + 16 * s + kh);
      b[m][s] = *reinterpret_cast<const bf16x8*>(sB + (wc * WT + m * 32 + lr) * LDS_LD16 + 16 * s + kh);
    }
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int mj = 0; mj < MT; ++mj)
        acc[mi][mj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][s], b[mj][s], acc[mi][mj], 0, 0, 0);
}

template <int NB>
__device__ __forceinline__ void load_chunk16(f32x4 (&r)[TileCfg16<NB>::LD4], const bf16raw* __restrict__ g,
                                             size_t ld, int tid) {
#pragma unroll
  for (int i = 0; i < TileCfg16<NB>::LD4; ++i) {
    const int f = tid + 256 * i, row = f >> 3, c = f & 7;
    r[i] = *reinterpret_cast<const f32x4*>(g + (size_t)row * ld + 8 * c);
  }
}
template <int NB>
__device__ __forceinline__ void store_chunk16(const f32x4 (&r)[TileCfg16<NB>::LD4], bf16raw* s, int tid) {
#pragma unroll
  for (int i = 0; i < TileCfg16<NB>::LD4; ++i) {
    const int f = tid + 256 * i, row = f >> 3, c = f & 7;
    *reinterpret_cast<f32x4*>(s + row * LDS_LD16 + 8 * c) = r[i];
  }
}

// C = act(A W' + bias); A [M][K] bf16, Wt [N][K] bf16; C bf16 (hidden layers) or f32 (head).
template <int NB, bool RELU, bool BIAS, bool OUT_BF16>
__global__ __launch_bounds__(256) void gemm_nt_bf16_k(void* __restrict__ Cv, size_t ldc,
                                                      const bf16raw* __restrict__ A, size_t lda,
                                                      const bf16raw* __restrict__ B, size_t ldb, int K,
                                                      const float* __restrict__ bias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bf16raw* lds = reinterpret_cast<bf16raw*>(lds_raw);
  using Cf = TileCfg16<NB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * NB, n0 = blockIdx.x * NB;
  f32x16 acc[Cf::MT][Cf::MT];
#pragma unroll
  for (int i = 0; i < Cf::MT; ++i)
#pragma unroll
    for (int j = 0; j < Cf::MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bf16raw* Ag = A + (size_t)m0 * lda;
  const bf16raw* Bg = B + (size_t)n0 * ldb;
  const int nk = K / KC16;
  f32x4 ra[Cf::LD4], rb[Cf::LD4];
  load_chunk16<NB>(ra, Ag, lda, tid);
  load_chunk16<NB>(rb, Bg, ldb, tid);
  for (int kc = 0; kc < nk; ++kc) {
    bf16raw* sA = lds + (kc & 1) * 2 * Cf::STAGE_ELEMS;
    bf16raw* sB = sA + Cf::STAGE_ELEMS;
    store_chunk16<NB>(ra, sA, tid);
    store_chunk16<NB>(rb, sB, tid);
    __syncthreads();
    if (kc + 1 < nk) {
      load_chunk16<NB>(ra, Ag + (kc + 1) * KC16, lda, tid);
      load_chunk16<NB>(rb, Bg + (kc + 1) * KC16, ldb, tid);
    }
    mma_chunk16<NB>(acc, sA, sB, wr, wc, lane);
  }
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
        if (OUT_BF16) reinterpret_cast<__bf16*>(Cv)[(size_t)row * ldc + col] = (__bf16)v;
        else reinterpret_cast<float*>(Cv)[(size_t)row * ldc + col] = v;
      }
}

}  // namespace nnmpc
