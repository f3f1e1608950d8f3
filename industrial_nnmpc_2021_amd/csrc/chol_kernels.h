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
//   chol_diag_k : C = K[j,j] - Dacc[j], potrf + trtri in LDS  (Dacc[j] = sum_k L[j,k] L[j,k]'
//                 was accumulated by the panel kernels of steps k < j)
//   chol_panel_k: for every i > j:  L[i,j] = (K[i,j] - sum_k L[i,k] L[j,k]') Y_j'
//                (two chained MFMA GEMMs, the second fed through LDS), then
//                Dacc[i] += L[i,j] L[i,j]'  (third MFMA product, same LDS image as A and B)
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
  float* Dacc;        // [slots][T][NB*NB] running sum_k L[i,k] L[i,k]' of every diagonal tile
  const float* dvec;  // [slots][np]
  const float* mask;  // [slots][np]
  const int* flag;    // [slots] factor this slot?
  int* fail;          // [slots] set to 1 on a non-positive pivot
};

__device__ __forceinline__ size_t tile_off(int i, int j, int nb2) {
  return ((size_t)i * (i + 1) / 2 + j) * nb2;
}

template <int NB>
constexpr int chol_panel_lds_bytes() {
  return TileCfg<NB>::LDS_FLOATS * 4;
}

// ---- 32 x 32 building blocks of the diagonal-tile factorisation ----------------
__device__ __forceinline__ float rdlane(float x, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}

// One wave; lane (l & 31) owns row (l & 31) of a 32 x 32 SPD block in a[0..31].
// On exit a = its Cholesky factor (entries right of the diagonal are garbage) and
// y = column (l & 31) of the inverse factor.  Pivot rows travel by v_readlane.
__device__ __forceinline__ int potrf_trtri_32(float (&a)[32], float (&y)[32], int col) {
  int bad = 0;
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    float d = rdlane(a[c], c);
    if (!(d > 1e-30f)) { d = 1e-30f; bad = 1; }
    const float inv = 1.0f / sqrtf(d);
    const float lc = a[c] * inv;
    a[c] = lc;
#pragma unroll
    for (int cc = c + 1; cc < 32; ++cc) a[cc] -= lc * rdlane(lc, cc);
  }
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    float s = (r == col) ? 1.f : 0.f;
#pragma unroll
    for (int m = 0; m < r; ++m) s -= rdlane(a[m], r) * y[m];
    y[r] = s / rdlane(a[r], r);
  }
  return bad;
}

// 32x32x32 products on LDS operands (one wave).  A is read as rows [r][k]
// (ds_read_b128).  B is read either as rows [c][k] (NT) or as [k][c] (NN).
__device__ __forceinline__ f32x16 mma32_nt(const float* A, int lda, const float* B, int ldb, int lane,
                                           f32x16 acc) {
  const int lr = lane & 31, kh = (lane >> 5) * 16;
  f32x4 a4[4], b4[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    a4[v] = *reinterpret_cast<const f32x4*>(A + lr * lda + kh + 4 * v);
    b4[v] = *reinterpret_cast<const f32x4*>(B + lr * ldb + kh + 4 * v);
  }
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[v][e], b4[v][e], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 mma32_nn(const float* A, int lda, const float* Bt, int ldb, int lane,
                                           f32x16 acc) {
  const int lr = lane & 31, kh = (lane >> 5) * 16;
  f32x4 a4[4];
  float b[16];
#pragma unroll
  for (int v = 0; v < 4; ++v) a4[v] = *reinterpret_cast<const f32x4*>(A + lr * lda + kh + 4 * v);
#pragma unroll
  for (int k = 0; k < 16; ++k) b[k] = Bt[(kh + k) * ldb + lr];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[v][e], b[4 * v + e], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

template <int NB>
constexpr int chol_diag_lds_bytes() {
  return (2 * NB * (NB + 4) + 4 * 32 * 36) * 4;
}

// Diagonal tile of step j:  C = K[j,j] - Dacc[j];  L_jj = chol(C);  Y_j = L_jj^-1.
// Blocked with 32 x 32 sub-blocks, all in LDS (row stride NB+4: 16-B aligned rows,
// conflict-free ds_read_b128 operand reads): sub-block potrf + trtri in the
// registers of wave 0, TRSM / SYRK / assembly of the inverse on MFMA by all waves.
template <int NB>
__global__ __launch_bounds__(256) void chol_diag_k(CholArgs a, int j) {
  const int p = blockIdx.x;
  if (!a.flag[p]) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LD = NB + 4;
  constexpr int NS = NB / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Lp = a.L + (size_t)p * a.tiles * NB * NB;
  float* Cs = lds;
  float* Ys = lds + NB * LD;
  float* Ts = Ys + NB * LD + wave * 32 * 36;   // per-wave 32 x 32 scratch (stride 36)

  const float* Pjj = a.Pt + tile_off(j, j, NB * NB);
  float* Dj = a.Dacc + ((size_t)p * a.T + j) * NB * NB;
  const float* mk = a.mask + (size_t)p * a.np + j * NB;
  const float* dv = a.dvec + (size_t)p * a.np + j * NB;
  for (int e4 = tid; e4 < NB * NB / 4; e4 += 256) {
    const int row = (4 * e4) / NB, col = (4 * e4) % NB;
    const f32x4 pv = *reinterpret_cast<const f32x4*>(Pjj + 4 * e4);
    const f32x4 dd = *reinterpret_cast<const f32x4*>(Dj + 4 * e4);
    const f32x4 mc = *reinterpret_cast<const f32x4*>(mk + col);
    const float mr = mk[row];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = mr * mc[e] * pv[e] - dd[e];
      if (row == col + e) v[e] += dv[row];
    }
    *reinterpret_cast<f32x4*>(Cs + row * LD + col) = v;
    *reinterpret_cast<f32x4*>(Ys + row * LD + col) = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(Dj + 4 * e4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();

  for (int s = 0; s < NS; ++s) {
    if (wave == 0) {
      const int r = lane & 31;
      float av[32], yv[32];
      const float* src = Cs + (32 * s + r) * LD + 32 * s;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * v);
        av[4 * v] = t[0]; av[4 * v + 1] = t[1]; av[4 * v + 2] = t[2]; av[4 * v + 3] = t[3];
      }
#pragma unroll
      for (int m = 0; m < 32; ++m) yv[m] = 0.f;
      const int bad = potrf_trtri_32(av, yv, r);
      if (lane < 32) {
        float* dst = Cs + (32 * s + r) * LD + 32 * s;
#pragma unroll
        for (int c = 0; c < 32; ++c) dst[c] = (c <= r) ? av[c] : 0.f;
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) Ys[(32 * s + rr) * LD + 32 * s + r] = yv[rr];
        if (bad && lane == 0) a.fail[p] = 1;
      }
    }
    __syncthreads();
    // TRSM: C[i,s] <- C[i,s] Y_ss'   (one sub-block per wave)
    for (int i = s + 1 + wave; i < NS; i += 4) {
      f32x16 acc = mma32_nt(Cs + 32 * i * LD + 32 * s, LD, Ys + 32 * s * LD + 32 * s, LD, lane, zero16());
#pragma unroll
      for (int r = 0; r < 16; ++r) Cs[(32 * i + acc_row(r, lane)) * LD + 32 * s + acc_col(lane)] = acc[r];
    }
    __syncthreads();
    // SYRK: C[i,k] -= L[i,s] L[k,s]'  for s < k <= i
    {
      int b = 0;
      for (int i = s + 1; i < NS; ++i)
        for (int k = s + 1; k <= i; ++k, ++b) {
          if ((b & 3) != wave) continue;
          f32x16 acc = mma32_nt(Cs + 32 * i * LD + 32 * s, LD, Cs + 32 * k * LD + 32 * s, LD, lane, zero16());
#pragma unroll
          for (int r = 0; r < 16; ++r) Cs[(32 * i + acc_row(r, lane)) * LD + 32 * k + acc_col(lane)] -= acc[r];
        }
    }
    __syncthreads();
  }

  // ---- write L[j,j] (strictly upper part zeroed)
  float* Ljj = Lp + tile_off(j, j, NB * NB);
  for (int e4 = tid; e4 < NB * NB / 4; e4 += 256) {
    const int row = (4 * e4) / NB, col = (4 * e4) % NB;
    f32x4 v = *reinterpret_cast<const f32x4*>(Cs + row * LD + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) if (col + e > row) v[e] = 0.f;
    *reinterpret_cast<f32x4*>(Ljj + 4 * e4) = v;
  }

  // ---- assemble Y = L^-1 below the diagonal sub-blocks, distance d = i - k at a time:
  //      Y[i,k] = -Y[i,i] * sum_{m=k}^{i-1} L[i,m] Y[m,k]
  for (int d = 1; d < NS; ++d) {
    for (int i = d + wave; i < NS; i += 4) {
      const int k = i - d;
      f32x16 t = zero16();
      for (int m = k; m < i; ++m)
        t = mma32_nn(Cs + 32 * i * LD + 32 * m, LD, Ys + 32 * m * LD + 32 * k, LD, lane, t);
#pragma unroll
      for (int r = 0; r < 16; ++r) Ts[acc_row(r, lane) * 36 + acc_col(lane)] = t[r];
      // same wave wrote Ts; LDS ops of one wave complete in order
      f32x16 y = mma32_nn(Ys + 32 * i * LD + 32 * i, LD, Ts, 36, lane, zero16());
#pragma unroll
      for (int r = 0; r < 16; ++r) Ys[(32 * i + acc_row(r, lane)) * LD + 32 * k + acc_col(lane)] = -y[r];
    }
    __syncthreads();
  }
  float* Yj = a.Y + ((size_t)p * a.T + j) * NB * NB;
  for (int e4 = tid; e4 < NB * NB / 4; e4 += 256) {
    const int row = (4 * e4) / NB, col = (4 * e4) % NB;
    *reinterpret_cast<f32x4*>(Yj + 4 * e4) = *reinterpret_cast<const f32x4*>(Ys + row * LD + col);
  }
}

// Stage columns [32 MC, 32 MC + 32) of an NB x NB matrix held in accumulators
// (wave (wr, wc) owns rows wr*WT.., cols wc*WT..) into an LDS chunk image.
template <int NB, int MC>
__device__ __forceinline__ void stage_acc_chunk(const f32x16 (&x)[TileCfg<NB>::MT][TileCfg<NB>::MT],
                                                float* sA, int wr, int wc, int lane) {
  using C = TileCfg<NB>;
  constexpr int WT = C::WT;
  constexpr int NEED_WC = (MC * 32) / WT;
  constexpr int MJ = ((MC * 32) % WT) / 32;
  if (wc == NEED_WC) {
#pragma unroll
    for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sA[(wr * WT + mi * 32 + acc_row(r, lane)) * LDS_LD + acc_col(lane)] = x[mi][MJ][r];
  }
}

// Panel tile (i, j), i > j, of step j:
//   acc  = sum_k L[i,k] L[j,k]' - mask o P[i,j]                 (= -C; P preloaded into acc)
//   nout = acc Y_j'  (= -L[i,j]; Y_j lower triangular: zero 32-blocks skipped)
//   Dacc[i] += nout nout'  (lower 32-blocks only), accumulated on top of the tile loaded from HBM.
// The two epilogue products run chunk by chunk through alternating LDS buffers
// (one barrier per chunk), the next Y chunk is fetched while the current one is multiplied.
template <int NB>
__global__ __launch_bounds__(256, 2) void chol_panel_k(CholArgs a, int j) {
  const int p = blockIdx.y;
  if (!a.flag[p]) return;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using C = TileCfg<NB>;
  constexpr int NC = NB / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = j + 1 + blockIdx.x;
  float* Lp = a.L + (size_t)p * a.tiles * NB * NB;

  f32x16 acc[C::MT][C::MT];
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
        acc[mi][mj][r] = -(mki[row] * mkj[col] * Pij[row * NB + col]);
      }
  TileRowOp<NB> A{Lp + tile_off(i, 0, NB * NB)};
  TileRowOp<NB> B{Lp + tile_off(j, 0, NB * NB)};
  tile_gemm_nt<NB>(acc, A, B, j * NB, lds, false);

  // ---- nout = acc * Y_j'
  f32x16 out[C::MT][C::MT];
  zero_acc<NB>(out);
  const float* Yj = a.Y + ((size_t)p * a.T + j) * NB * NB;
  f32x4 rb[C::LD4];
  load_chunk<NB>(rb, Yj, NB, tid);
  auto tri = [&](int mc) {  // chunk mc of Y' only feeds output 32-blocks cb >= mc
    return [=](int, int mj) { return mc <= wc * C::MT + mj; };
  };
#define NNMPC_TRSM_CHUNK(MC)                                                        \
  {                                                                                 \
    float* sA = lds + ((MC) & 1) * 2 * C::STAGE_FLOATS;                             \
    float* sB = sA + C::STAGE_FLOATS;                                               \
    stage_acc_chunk<NB, MC>(acc, sA, wr, wc, lane);                                 \
    store_chunk<NB>(rb, sB, tid);                                                   \
    __syncthreads();                                                                \
    if ((MC) + 1 < NC) load_chunk<NB>(rb, Yj + ((MC) + 1) * 32, NB, tid);           \
    mma_chunk_pred<NB>(out, sA, sB, wr, wc, lane, tri(MC));                         \
  }
  NNMPC_TRSM_CHUNK(0)
  NNMPC_TRSM_CHUNK(1)
  if constexpr (NB == 128) {
    NNMPC_TRSM_CHUNK(2)
    NNMPC_TRSM_CHUNK(3)
  }
#undef NNMPC_TRSM_CHUNK

  float* Lij = Lp + tile_off(i, j, NB * NB);
#pragma unroll
  for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
    for (int mj = 0; mj < C::MT; ++mj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * C::WT + mi * 32 + acc_row(r, lane);
        const int col = wc * C::WT + mj * 32 + acc_col(lane);
        Lij[row * NB + col] = -out[mi][mj][r];
      }

  // ---- Dacc[i] += nout nout'   (32-blocks with column block <= row block)
  float* Di = a.Dacc + ((size_t)p * a.T + i) * NB * NB;
  auto low = [=](int mi, int mj) { return wc * C::MT + mj <= wr * C::MT + mi; };
#pragma unroll
  for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
    for (int mj = 0; mj < C::MT; ++mj)
      if (low(mi, mj)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr * C::WT + mi * 32 + acc_row(r, lane);
          const int col = wc * C::WT + mj * 32 + acc_col(lane);
          acc[mi][mj][r] = Di[row * NB + col];
        }
      }
#define NNMPC_SYRK_CHUNK(MC)                                                        \
  {                                                                                 \
    float* sA = lds + (((MC) + NC) & 1) * 2 * C::STAGE_FLOATS;                      \
    stage_acc_chunk<NB, MC>(out, sA, wr, wc, lane);                                 \
    __syncthreads();                                                                \
    mma_chunk_pred<NB>(acc, sA, sA, wr, wc, lane, low);                             \
  }
  NNMPC_SYRK_CHUNK(0)
  NNMPC_SYRK_CHUNK(1)
  if constexpr (NB == 128) {
    NNMPC_SYRK_CHUNK(2)
    NNMPC_SYRK_CHUNK(3)
  }
#undef NNMPC_SYRK_CHUNK
#pragma unroll
  for (int mi = 0; mi < C::MT; ++mi)
#pragma unroll
    for (int mj = 0; mj < C::MT; ++mj)
      if (low(mi, mj)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wr * C::WT + mi * 32 + acc_row(r, lane);
          const int col = wc * C::WT + mj * 32 + acc_col(lane);
          Di[row * NB + col] = acc[mi][mj][r];
        }
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
  unsigned long long* count;  // number of slot-solves executed (statistics)
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
  if (tid == 0 && a.count) atomicAdd(a.count, 1ULL);
}

}  // namespace nnmpc
