// Batched left-looking blocked Cholesky of  K_p = mask_p mask_p' o P + diag(dvec_p)
// (P shared by every problem of the batch, mask/dvec per problem) and the two
// triangular solves, for gfx950.
//
// Storage: every problem owns a tile-packed lower factor: tiles (i,j), i >= j,
// NB x NB row-major, tile (i,j) at index i(i+1)/2 + j, so block row i is one
// contiguous run of i+1 tiles (the K-stream of the left-looking update and of
// the forward solve).  The inverses Y_j = L_jj^-1 of the diagonal blocks are
// kept next to it ([T][NB][NB]); panel TRSMs and both TRSVs then become GEMM /
// GEMV with Y_j instead of substitutions.
//
// Step j of the factorisation (host loop, two launches):
//   chol_diag_k : C = K[j,j] - sum_k L[j,k] L[j,k]'  (MFMA SYRK), potrf + trtri in LDS
//   chol_panel_k: for every i > j:  L[i,j] = (K[i,j] - sum_k L[i,k] L[j,k]') Y_j'
//                (two chained MFMA GEMMs, the second fed through LDS)
#pragma once
#include "tile_gemm.h"

namespace nnmpc {

struct CholArgs {
  int n;        // true dimension
  int np;       // padded to a multiple of NB
  int T;        // np / NB
  int tiles;    // T (T + 1) / 2
  const float* Pt;    // tile-packed lower P (pad diagonal = 1)
  float* L;           // [slots][tiles][NB*NB]
  float* Y;           // [slots][T][NB*NB]
  const float* dvec;  // [slots][np]
  const float* mask;  // [slots][np]
  const int* flag;    // [slots] factor this slot?
  int* fail;          // [slots] set to 1 on a non-positive pivot
};

__device__ __forceinline__ size_t tile_off(int i, int j, int nb2) {
  return ((size_t)i * (i + 1) / 2 + j) * nb2;
}

template <int NB>
constexpr int chol_diag_lds_bytes() {
  constexpr int a = TileCfg<NB>::LDS_FLOATS * 4;
  constexpr int b = 2 * NB * (NB + 1) * 4;
  return a > b ? a : b;
}
template <int NB>
constexpr int chol_panel_lds_bytes() {
  return TileCfg<NB>::LDS_FLOATS * 4;
}

template <int NB>
__global__ __launch_bounds__(256) void chol_diag_k(CholArgs a, int j) {
  const int p = blockIdx.x;
  if (!a.flag[p]) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using C = TileCfg<NB>;
  constexpr int S = NB + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  float* Lp = a.L + (size_t)p * a.tiles * NB * NB;

  f32x16 acc[C::MT][C::MT];
  zero_acc<NB>(acc);
  TileRowOp<NB> A{Lp + tile_off(j, 0, NB * NB)};
  tile_gemm_nt<NB>(acc, A, A, j * NB, lds, true);

  // C = mask mask' o P[j,j] + diag(dvec) - acc  -> LDS [NB][NB+1]
  float* Cs = lds;
  float* Ys = lds + NB * S;
  const float* Pjj = a.Pt + tile_off(j, j, NB * NB);
  const float* mk = a.mask + (size_t)p * a.np + j * NB;
  const float* dv = a.dvec + (size_t)p * a.np + j * NB;
#pragma unroll
  for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
    for (int mj = 0; mj < C::MT; ++mj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * C::WT + mi * 32 + acc_row(r, lane);
        const int col = wc * C::WT + mj * 32 + acc_col(lane);
        float v = mk[row] * mk[col] * Pjj[row * NB + col] - acc[mi][mj][r];
        if (row == col) v += dv[row];
        Cs[row * S + col] = v;
      }
  __syncthreads();

  // ---- potrf (right-looking, in LDS)
  int bad = 0;
  for (int c = 0; c < NB; ++c) {
    float piv = Cs[c * S + c];
    if (!(piv > 1e-30f)) { piv = 1e-30f; bad = 1; }
    const float sq = sqrtf(piv);
    const float inv = 1.0f / sq;
    for (int r = c + 1 + tid; r < NB; r += 256) Cs[r * S + c] *= inv;
    __syncthreads();
    if (tid == 0) Cs[c * S + c] = sq;
    for (int r = c + 1 + wave; r < NB; r += 4) {
      const float lrc = Cs[r * S + c];
      for (int cc = c + 1 + lane; cc <= r; cc += 64) Cs[r * S + cc] -= lrc * Cs[cc * S + c];
    }
    __syncthreads();
  }
  if (bad && tid == 0) a.fail[p] = 1;

  // ---- write L[j,j] (strict upper zeroed)
  float* Ljj = Lp + tile_off(j, j, NB * NB);
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e / NB, c = e % NB;
    Ljj[e] = (c <= r) ? Cs[r * S + c] : 0.f;
  }

  // ---- trtri: column c of Y = L^-1 by forward substitution (thread c)
  if (tid < NB) {
    const int c = tid;
    for (int r = 0; r < NB; ++r) {
      float s = (r == c) ? 1.f : 0.f;
      for (int m = 0; m < r; ++m) s -= Cs[r * S + m] * Ys[m * S + c];
      Ys[r * S + c] = (r >= c) ? s / Cs[r * S + r] : 0.f;
    }
  }
  __syncthreads();
  float* Yj = a.Y + ((size_t)p * a.T + j) * NB * NB;
  for (int e = tid; e < NB * NB; e += 256) {
    const int r = e / NB, c = e % NB;
    Yj[e] = Ys[r * S + c];
  }
}

// One K-chunk (32 columns MC*32.. of C) of  out += C * Y';  C lives in the
// accumulators of the waves with wc == (MC*32)/WT, sub-tile column MJ (static).
template <int NB, int MC>
__device__ __forceinline__ void second_gemm_chunk(
    const f32x16 (&acc)[TileCfg<NB>::MT][TileCfg<NB>::MT],
    f32x16 (&out)[TileCfg<NB>::MT][TileCfg<NB>::MT], const float* Yj, float* sA, float* sB,
    int wr, int wc, int lane, int tid) {
  using C = TileCfg<NB>;
  constexpr int WT = C::WT;
  constexpr int NEED_WC = (MC * 32) / WT;
  constexpr int MJ = ((MC * 32) % WT) / 32;
  if (wc == NEED_WC) {
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sA[(wr * WT + mi * 32 + acc_row(r, lane)) * LDS_LD + acc_col(lane)] = acc[mi][MJ][r];
  }
  f32x4 rb[C::LD4];
  load_chunk<NB>(rb, Yj + MC * 32, NB, tid);
  store_chunk<NB>(rb, sB, tid);
  __syncthreads();
  mma_chunk<NB>(out, sA, sB, wr, wc, lane);
  __syncthreads();
}

template <int NB>
__global__ __launch_bounds__(256) void chol_panel_k(CholArgs a, int j) {
  const int p = blockIdx.y;
  if (!a.flag[p]) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using C = TileCfg<NB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = j + 1 + blockIdx.x;
  float* Lp = a.L + (size_t)p * a.tiles * NB * NB;

  f32x16 acc[C::MT][C::MT];
  zero_acc<NB>(acc);
  TileRowOp<NB> A{Lp + tile_off(i, 0, NB * NB)};
  TileRowOp<NB> B{Lp + tile_off(j, 0, NB * NB)};
  tile_gemm_nt<NB>(acc, A, B, j * NB, lds, false);

  // C = mask_i mask_j' o P[i,j] - acc   (kept in registers)
  const float* Pij = a.Pt + tile_off(i, j, NB * NB);
  const float* mki = a.mask + (size_t)p * a.np + i * NB;
  const float* mkj = a.mask + (size_t)p * a.np + j * NB;
#pragma unroll
  for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
    for (int mj = 0; mj < C::MT; ++mj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * C::WT + mi * 32 + acc_row(r, lane);
        const int col = wc * C::WT + mj * 32 + acc_col(lane);
        acc[mi][mj][r] = mki[row] * mkj[col] * Pij[row * NB + col] - acc[mi][mj][r];
      }

  // out = C * Y_j'  : K-chunks of C go registers -> LDS, chunks of Y global -> LDS
  f32x16 out[C::MT][C::MT];
  zero_acc<NB>(out);
  const float* Yj = a.Y + ((size_t)p * a.T + j) * NB * NB;
  float* sA = lds;
  float* sB = lds + C::STAGE_FLOATS;
  second_gemm_chunk<NB, 0>(acc, out, Yj, sA, sB, wr, wc, lane, tid);
  second_gemm_chunk<NB, 1>(acc, out, Yj, sA, sB, wr, wc, lane, tid);
  if constexpr (NB == 128) {
    second_gemm_chunk<NB, 2>(acc, out, Yj, sA, sB, wr, wc, lane, tid);
    second_gemm_chunk<NB, 3>(acc, out, Yj, sA, sB, wr, wc, lane, tid);
  }

  float* Lij = Lp + tile_off(i, j, NB * NB);
#pragma unroll
  for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
    for (int mj = 0; mj < C::MT; ++mj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * C::WT + mi * 32 + acc_row(r, lane);
        const int col = wc * C::WT + mj * 32 + acc_col(lane);
        Lij[row * NB + col] = out[mi][mj][r];
      }
}

// ---------------------------------------------------------------------------
// x = K^-1 rhs  via  L y = rhs,  L' x = y.   One workgroup (256 threads) per
// problem; L is streamed once per sweep (memory-bound), y/x live in LDS.
// GEMV lane map: a tile row is NB floats = NB/4 float4; LPR = NB/4 lanes cover
// one row, a wave-instruction covers RPI = 64/LPR rows, a wave owns NB/4 rows.
struct TrsvArgs {
  int n, np, T, tiles;
  const float* L;
  const float* Y;
  const float* rhs;   // [slots][np]
  float* sol;         // [slots][np]
  const int* flag;    // [slots]
};

template <int NB>
constexpr int trsv_lds_bytes(int np) {
  return (np + 5 * NB) * 4;
}

template <int NB>
__device__ __forceinline__ void gemv_rows(float (&acc)[NB * NB / 1024], const float* tile,
                                          const float* xs, int wave, int lane) {
  constexpr int LPR = NB / 4, RPI = 64 / LPR, NI = NB * NB / 1024;
  const int c4 = lane % LPR, rsub = lane / LPR;
  const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + 4 * c4);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = wave * (NB / 4) + i * RPI + rsub;
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * NB + 4 * c4);
    acc[i] += v[0] * xv[0] + v[1] * xv[1] + v[2] * xv[2] + v[3] * xv[3];
  }
}
template <int NB>
__device__ __forceinline__ void gemv_cols(f32x4& acc, const float* tile, const float* xs,
                                          int wave, int lane) {
  constexpr int LPR = NB / 4, RPI = 64 / LPR, NI = NB * NB / 1024;
  const int c4 = lane % LPR, rsub = lane / LPR;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = wave * (NB / 4) + i * RPI + rsub;
    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row * NB + 4 * c4);
    const float xr = xs[row];
    acc[0] += v[0] * xr; acc[1] += v[1] * xr; acc[2] += v[2] * xr; acc[3] += v[3] * xr;
  }
}

template <int NB>
__global__ __launch_bounds__(256) void trsv_k(TrsvArgs a) {
  const int p = blockIdx.x;
  if (!a.flag[p]) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LPR = NB / 4, RPI = 64 / LPR, NI = NB * NB / 1024;
  float* ys = lds;                 // [np]
  float* tv = lds + a.np;          // [NB]
  float* red = tv + NB;            // [4][NB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c4 = lane % LPR, rsub = lane / LPR;
  const float* Lp = a.L + (size_t)p * a.tiles * NB * NB;
  const float* Yp = a.Y + (size_t)p * a.T * NB * NB;
  const float* rhs = a.rhs + (size_t)p * a.np;

  // ---- forward:  y_j = Y_j (rhs_j - sum_{k<j} L[j,k] y_k)
  for (int j = 0; j < a.T; ++j) {
    float acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = 0.f;
    const float* row0 = Lp + tile_off(j, 0, NB * NB);
    for (int k = 0; k < j; ++k) gemv_rows<NB>(acc, row0 + (size_t)k * NB * NB, ys + k * NB, wave, lane);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float v = acc[i];
      for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o);
      if (c4 == 0) {
        const int row = wave * (NB / 4) + i * RPI + rsub;
        tv[row] = rhs[j * NB + row] - v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = 0.f;
    gemv_rows<NB>(acc, Yp + (size_t)j * NB * NB, tv, wave, lane);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float v = acc[i];
      for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o);
      if (c4 == 0) ys[j * NB + wave * (NB / 4) + i * RPI + rsub] = v;
    }
    __syncthreads();
  }

  // ---- backward:  x_j = Y_j' (y_j - sum_{i>j} L[i,j]' x_i)    (x overwrites y)
  for (int j = a.T - 1; j >= 0; --j) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = j + 1; i < a.T; ++i)
      gemv_cols<NB>(acc, Lp + tile_off(i, j, NB * NB), ys + i * NB, wave, lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[e];
      for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o);
      acc[e] = v;
    }
    if (rsub == 0) *reinterpret_cast<f32x4*>(red + wave * NB + 4 * c4) = acc;
    __syncthreads();
    if (tid < NB) tv[tid] = ys[j * NB + tid] - (red[tid] + red[NB + tid] + red[2 * NB + tid] + red[3 * NB + tid]);
    __syncthreads();
    acc = {0.f, 0.f, 0.f, 0.f};
    gemv_cols<NB>(acc, Yp + (size_t)j * NB * NB, tv, wave, lane);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = acc[e];
      for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o);
      acc[e] = v;
    }
    if (rsub == 0) *reinterpret_cast<f32x4*>(red + wave * NB + 4 * c4) = acc;
    __syncthreads();
    if (tid < NB) ys[j * NB + tid] = red[tid] + red[NB + tid] + red[2 * NB + tid] + red[3 * NB + tid];
    __syncthreads();
  }
  float* sol = a.sol + (size_t)p * a.np;
  for (int r = tid; r < a.np; r += 256) sol[r] = ys[r];
}

}  // namespace nnmpc
