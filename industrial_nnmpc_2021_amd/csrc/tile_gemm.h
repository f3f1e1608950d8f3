// NT tile GEMM core for gfx950:  acc[r][c] += sum_k A[r][k] * B[c][k]
// (both operands row-major with K contiguous), exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32), 256 threads = 4 wave64 per NB x NB output tile.
//
// LDS image per operand and K-chunk: [NB rows][KC=32 floats], row stride 36
// floats (144 B): a lane's ds_read_b128 of 4 consecutive k then lands on
// 16-B slot (9*row + const) mod 16 -> the 16 lanes of every b128 lane group
// hit 16 distinct slots (conflict-free), and rows stay 16-B aligned.
// Lane l of a wave feeds MFMA step s (0..15) with k = 16*(l>>5) + s, i.e. the
// two k of one 32x32x2 instruction are (s, 16+s); A and B use the same map so
// the permutation of k inside the chunk is harmless.
#pragma once
#include <hip/hip_runtime.h>

namespace nnmpc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;          // K chunk staged per barrier
constexpr int LDS_LD = 36;      // padded row stride (floats) of a staged chunk

template <int NB>
struct TileCfg {
  static constexpr int WT = NB / 2;            // wave tile edge (2x2 waves)
  static constexpr int MT = WT / 32;           // 32x32 MFMA tiles per wave edge
  static constexpr int LD4 = NB * (KC / 4) / 256;  // float4 loads / thread / operand / chunk
  static constexpr int STAGE_FLOATS = NB * LDS_LD;  // one operand, one buffer
  static constexpr int LDS_FLOATS = 4 * STAGE_FLOATS;  // A,B x double buffer
  static_assert(NB == 64 || NB == 128, "NB must be 64 or 128");
};

// Row r (0..31), col c of a 32x32 accumulator register `reg` held by `lane`.
__device__ __forceinline__ int acc_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int lane) { return lane & 31; }

// One staged chunk (already in LDS): acc += A_chunk * B_chunk'.
template <int NB>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[TileCfg<NB>::MT][TileCfg<NB>::MT],
                                          const float* __restrict__ sA,
                                          const float* __restrict__ sB, int wr, int wc,
                                          int lane) {
  constexpr int MT = TileCfg<NB>::MT;
  constexpr int WT = TileCfg<NB>::WT;
  const int lr = lane & 31, kh = (lane >> 5) * 16;
  f32x4 a[MT][4], b[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float* pa = sA + (wr * WT + m * 32 + lr) * LDS_LD + kh;
    const float* pb = sB + (wc * WT + m * 32 + lr) * LDS_LD + kh;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      a[m][v] = *reinterpret_cast<const f32x4*>(pa + 4 * v);
      b[m][v] = *reinterpret_cast<const f32x4*>(pb + 4 * v);
    }
  }
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int mj = 0; mj < MT; ++mj)
          acc[mi][mj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][v][e], b[mj][v][e],
                                                              acc[mi][mj], 0, 0, 0);
}

// Same, but only the 32x32 sub-tiles (mi, mj) for which pred(mi, mj) holds
// (pred must be wave-uniform): triangular operands skip their zero blocks.
template <int NB, class Pred>
__device__ __forceinline__ void mma_chunk_pred(f32x16 (&acc)[TileCfg<NB>::MT][TileCfg<NB>::MT],
                                               const float* __restrict__ sA,
                                               const float* __restrict__ sB, int wr, int wc,
                                               int lane, Pred pred) {
  constexpr int MT = TileCfg<NB>::MT;
  constexpr int WT = TileCfg<NB>::WT;
  const int lr = lane & 31, kh = (lane >> 5) * 16;
  f32x4 a[MT][4], b[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float* pa = sA + (wr * WT + m * 32 + lr) * LDS_LD + kh;
    const float* pb = sB + (wc * WT + m * 32 + lr) * LDS_LD + kh;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      a[m][v] = *reinterpret_cast<const f32x4*>(pa + 4 * v);
      b[m][v] = *reinterpret_cast<const f32x4*>(pb + 4 * v);
    }
  }
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int mj = 0; mj < MT; ++mj)
      if (pred(mi, mj)) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[mi][mj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][v][e], b[mj][v][e],
                                                                acc[mi][mj], 0, 0, 0);
      }
}

// Global -> registers for one operand chunk: rows [0,NB) x k [k0, k0+32).
// `ld` is the row stride in floats; `g` points at row 0, k = k0.
template <int NB>
__device__ __forceinline__ void load_chunk(f32x4 (&r)[TileCfg<NB>::LD4],
                                           const float* __restrict__ g, size_t ld, int tid) {
#pragma unroll
  for (int i = 0; i < TileCfg<NB>::LD4; ++i) {
    const int f = tid + 256 * i;
    const int row = f >> 3, c4 = f & 7;
    r[i] = *reinterpret_cast<const f32x4*>(g + (size_t)row * ld + 4 * c4);
  }
}
template <int NB>
__device__ __forceinline__ void store_chunk(const f32x4 (&r)[TileCfg<NB>::LD4], float* s,
                                            int tid) {
#pragma unroll
  for (int i = 0; i < TileCfg<NB>::LD4; ++i) {
    const int f = tid + 256 * i;
    const int row = f >> 3, c4 = f & 7;
    *reinterpret_cast<f32x4*>(s + row * LDS_LD + 4 * c4) = r[i];
  }
}

// Operand addressing: K index kk -> pointer to (row 0, k = kk) and row stride.
// PlainOp: one row-major matrix with leading dimension ld.
struct PlainOp {
  const float* base;
  size_t ld;
  __device__ __forceinline__ const float* at(int kk) const { return base + kk; }
  __device__ __forceinline__ size_t stride() const { return ld; }
};
// TileRowOp: a block row of a tile-packed matrix: consecutive NB x NB
// row-major tiles along K.
template <int NB>
struct TileRowOp {
  const float* base;
  __device__ __forceinline__ const float* at(int kk) const {
    return base + (size_t)(kk / NB) * NB * NB + (kk % NB);
  }
  __device__ __forceinline__ size_t stride() const { return NB; }
};

// acc += A[0:NB, 0:K] * B[0:NB, 0:K]'   (K multiple of 32).  All 256 threads.
// One barrier per chunk: chunk kc is written to buffer kc&1, then a barrier,
// then computed on; the next write to that buffer happens after the barrier of
// chunk kc+1, which every wave reaches only after finishing chunk kc.
template <int NB, class OpA, class OpB>
__device__ __forceinline__ void tile_gemm_nt(f32x16 (&acc)[TileCfg<NB>::MT][TileCfg<NB>::MT],
                                             const OpA A, const OpB B, int K, float* lds,
                                             bool same_ab = false) {
  using C = TileCfg<NB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int nk = K / KC;
  if (nk == 0) return;
  f32x4 ra[C::LD4], rb[C::LD4];
  load_chunk<NB>(ra, A.at(0), A.stride(), tid);
  if (!same_ab) load_chunk<NB>(rb, B.at(0), B.stride(), tid);
  for (int kc = 0; kc < nk; ++kc) {
    float* sA = lds + (kc & 1) * 2 * C::STAGE_FLOATS;
    float* sB = same_ab ? sA : sA + C::STAGE_FLOATS;
    store_chunk<NB>(ra, sA, tid);
    if (!same_ab) store_chunk<NB>(rb, sB, tid);
    __syncthreads();
    if (kc + 1 < nk) {
      load_chunk<NB>(ra, A.at((kc + 1) * KC), A.stride(), tid);
      if (!same_ab) load_chunk<NB>(rb, B.at((kc + 1) * KC), B.stride(), tid);
    }
    mma_chunk<NB>(acc, sA, sB, wr, wc, lane);
  }
  __syncthreads();  // LDS free for the caller
}

template <int NB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TileCfg<NB>::MT][TileCfg<NB>::MT]) {
#pragma unroll
  for (int i = 0; i < TileCfg<NB>::MT; ++i)
#pragma unroll
    for (int j = 0; j < TileCfg<NB>::MT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

}  // namespace nnmpc
