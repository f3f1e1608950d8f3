// fp64 NT tile GEMM core for gfx950 (round 3):  acc[m][n] += sum_k A[m][k] * B[n][k]  on v_mfma_f64_16x16x4_f64.
//
// 128 x 128 output tile per 256-thread workgroup (2 x 2 waves of 64 x 64 = 4 x 4 MFMA tiles), K-chunks of 16.
// What changed against gemm_nt_f64_128_k of rounds 1-2 (40 % of its LDS time were bank conflicts, one ds_read_b64
// per operand element, staging through 16 registers per thread):
//   * the chunk goes global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write), one chunk
//     ahead of the MFMAs, into a two-stage ring (64 KB: two workgroups per CU);
//   * LDS image per operand and chunk: [128 rows][16 doubles], 128-byte rows (two rows per 256-byte bank row), the
//     16-byte slot s (0..7) of row r stored at slot  s ^ ((r >> 1) & 7).  LDS-DMA writes are lane-linear, so the
//     swizzle sits in the per-lane SOURCE address of the load and in the read address -- the same involution;
//   * operand fragments come in as ds_read_b128: lane (i = l & 15, kq = l >> 4) reads the doubles k = 8 t + 2 kq,
//     8 t + 2 kq + 1 of row i (t = 0, 1) -- first double to MFMA 2 t, second to MFMA 2 t + 1.  A and B use the same
//     map, so the order in which a chunk's k are summed is a permutation and nothing else.  With the swizzle every
//     16-lane group of a ds_read_b128 (lanes {0-3, 12-15, 20-27}, ...) touches 16 distinct 16-byte slots of a bank
//     row: conflict-free (16 reads per 64 MFMAs and wave);
//   * the K loop runs over TWO segments (A0 B0' then A1 B1', the second optionally subtracted): the full-width pass
//     of the active-set path is ONE GEMM  x = [x0 | lam] [Kunc | -Pinv]'  (qp_wide.h) -- x_unc beyond the column
//     window never exists in HBM;
//   * 1-D grid, tiles dealt to the XCDs so that one XCD (one L2) works on 4 row panels x all column tiles at a time.
#pragma once
#include <hip/hip_runtime.h>

namespace nnmpc {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int G64_KC = 16;                       // doubles of K per chunk
constexpr int G64_OP = 128 * G64_KC;             // doubles of one operand's chunk image
constexpr int G64_STAGE = 2 * G64_OP;            // A image | B image
constexpr int G64_LDS = 2 * G64_STAGE * 8;       // two stages: 65 536 B
constexpr int G64_R = 4;                         // row panels an XCD works on together

// One K segment: nk chunks of A[128 rows][..] (row 0 = first row of the tile, leading dimension lda) against
// B[128 rows][..] (first row = first column of the tile).  rowoff (nullable): per-row element offsets of A's rows
// relative to A (a gather: row i of the tile is A + rowoff[i]; lda unused).
struct G64Seg {
  const double* A; size_t lda;
  const double* B; size_t ldb;
  int nk;
};

// tile of workgroup `bid` of a 1-D grid over ntm x ntn tiles; false: no tile (the grid is padded).
// Workgroups go to the 8 XCDs round-robin by id; XCD x takes the row panels tm = 8 lm + x and walks them in groups of
// G64_R panels x all column tiles, column tile outermost: the 64 workgroups an XCD runs at a time share 4 A panels and
// ~16 B tiles through its L2.
__host__ __device__ inline int g64_grid(int ntm, int ntn) {
  const int nloc = (ntm + 7) / 8;
  return 8 * ((nloc + G64_R - 1) / G64_R) * G64_R * ntn;
}
__device__ __forceinline__ bool g64_tile_of(int bid, int ntm, int ntn, int& tm, int& tn) {
  const int per = G64_R * ntn, xcd = bid & 7, s = bid >> 3;
  const int gl = s / per, idx = s - gl * per;
  tn = idx / G64_R;
  tm = (gl * G64_R + (idx - tn * G64_R)) * 8 + xcd;
  return tm < ntm;
}

__device__ __forceinline__ void g64_glds(const char* src, double* lds_dst) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                   (void __attribute__((address_space(3)))*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ double g64_neg(double x, int mask) {   // mask = 0x80000000: -x; 0: x
  return __hiloint2double(__double2hiint(x) ^ mask, __double2loint(x));
}

// acc += A0 B0' (+ or -) A1 B1'.  sm: G64_LDS bytes of LDS, 16-byte aligned.  All 256 threads; ends with the LDS free.
// rowA0 (nullable, LDS): element offsets of the tile's 128 rows of A0 from s0.A (a gather; s0.lda is then unused).
template <bool NEG1>
__device__ __forceinline__ void g64_tile(f64x4 (&acc)[4][4], const G64Seg& s0, const G64Seg& s1, double* sm,
                                         const long long* rowA0 = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // ---- staging: wave-instruction q (0..3) of this wave fills rows 32 wave + 8 q + (lane >> 3) of both images;
  // lane -> physical slot lane & 7 of its row, i.e. logical slot (lane & 7) ^ ((row >> 1) & 7)
  unsigned vA0[4], vB0[4], vA1[4], vB1[4];                   // byte offsets from the segment's tile base
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 32 * wave + 8 * q + (lane >> 3);
    const int s = (lane & 7) ^ ((row >> 1) & 7);
    const long long ra0 = rowA0 ? rowA0[row] : (long long)row * (long long)s0.lda;   // (gather: offsets from the array base,
    vA0[q] = (unsigned)((ra0 + 2 * s) * 8);                                            // which the caller keeps below 4 GB)
    vB0[q] = (unsigned)(((size_t)row * s0.ldb + 2 * s) * 8);
    vA1[q] = (unsigned)(((size_t)row * s1.lda + 2 * s) * 8);
    vB1[q] = (unsigned)(((size_t)row * s1.ldb + 2 * s) * 8);
  }
  const int nk0 = s0.nk, nk = s0.nk + s1.nk;
  auto issue = [&](int c) {
    double* st = sm + (c & 1) * G64_STAGE + (32 * wave) * G64_KC;
    const bool first = c < nk0;
    const char* ab = reinterpret_cast<const char*>(first ? s0.A : s1.A) + (size_t)(first ? c : c - nk0) * (G64_KC * 8);
    const char* bb = reinterpret_cast<const char*>(first ? s0.B : s1.B) + (size_t)(first ? c : c - nk0) * (G64_KC * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) g64_glds(ab + (first ? vA0[q] : vA1[q]), st + 8 * q * G64_KC);
#pragma unroll
    for (int q = 0; q < 4; ++q) g64_glds(bb + (first ? vB0[q] : vB1[q]), st + G64_OP + 8 * q * G64_KC);
  };
  // ---- fragment reads: lane (i, kq), row block t16: bytes (64 w + 16 t16 + i) * 128 + ((4 t + kq) ^ ((i >> 1) & 7)) * 16
  const int li = lane & 15, kq = lane >> 4, sw = (li >> 1) & 7;
  const int ra = (wr * 64 + li) * 128, rb = (wc * 64 + li) * 128 + G64_OP * 8;
  const int o0 = ((kq) ^ sw) * 16, o1 = ((4 + kq) ^ sw) * 16;
  if (nk == 0) return;
  issue(0);
  __syncthreads();                                           // (with an LDS-DMA in flight this is vmcnt(0) + barrier)
  for (int c = 0; c < nk; ++c) {
    if (c + 1 < nk) issue(c + 1);                            // into the stage chunk c - 1 was read from: every wave is past
    const char* st = reinterpret_cast<const char*>(sm + (c & 1) * G64_STAGE);   // that chunk's MFMAs (the barrier below)
    f64x2 a[4][2], b[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      a[t][0] = *reinterpret_cast<const f64x2*>(st + ra + t * 2048 + o0);
      a[t][1] = *reinterpret_cast<const f64x2*>(st + ra + t * 2048 + o1);
      b[t][0] = *reinterpret_cast<const f64x2*>(st + rb + t * 2048 + o0);
      b[t][1] = *reinterpret_cast<const f64x2*>(st + rb + t * 2048 + o1);
    }
    if (NEG1) {
      const int mask = c >= nk0 ? (int)0x80000000 : 0;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) { a[t][h][0] = g64_neg(a[t][h][0], mask); a[t][h][1] = g64_neg(a[t][h][1], mask); }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][h][e], b[j][h][e], acc[i][j], 0, 0, 0);
    __syncthreads();                                         // chunk c + 1 has landed (vmcnt(0)), chunk c is read
  }
}

__device__ __forceinline__ void g64_zero(f64x4 (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
}

// C[M x N] = A[M x K] B[N x K]'  (row-major, K contiguous; M, N multiples of 128 as far as the grid goes, K of 16).
// 1-D grid of g64_grid(ntm, ntn) workgroups.
//   rowphase (nullable): per-row tag; 128-row tiles in which no row has tag `want` are skipped.
//   kdyn (nullable): device-side bound on the non-zero columns of A (index of the last one): kper = 0 one bound for the
//     launch, else kper consecutive entries per 128-row tile (the maximum counts).
//   mdyn (nullable): device-side row count; tiles beyond it are skipped.
//   rowmap (nullable): row i of A and of C is row rowmap[i] of the arrays (gather / scatter by problem; entries < 0 and
//     rows >= *mdyn: no row -- nothing is stored, row 0 is read).
static __global__ __launch_bounds__(256, 2) void gemm_nt_f64_t128_k(double* __restrict__ C, size_t ldc,
                                                                   const double* __restrict__ A, size_t lda,
                                                                   const double* __restrict__ B, size_t ldb,
                                                                   int K, int ntm, int ntn,
                                                                   const int* __restrict__ rowphase, int want,
                                                                   const int* __restrict__ kdyn, int kper,
                                                                   const int* __restrict__ mdyn,
                                                                   const int* __restrict__ rowmap) {
  extern __shared__ __attribute__((aligned(16))) double g64_sm[];
  __shared__ long long rowoff[128];
  int tm, tn;
  if (!g64_tile_of(blockIdx.x, ntm, ntn, tm, tn)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = tm * 128, n0 = tn * 128;
  const int mlim = mdyn ? *mdyn : 0x7fffffff;
  if (m0 >= mlim) return;
  if (kdyn) {
    int kl = kdyn[kper * tm];
    for (int i = 1; i < kper; ++i) kl = max(kl, kdyn[kper * tm + i]);
    K = min(K, ((kl + G64_KC) / G64_KC) * G64_KC);
  }
  if (rowphase) {
    const int need = tid < 128 ? (rowphase[m0 + tid] == want) : 0;
    if (!__syncthreads_or(need)) return;
  }
  if (rowmap) {                                              // (the caller keeps the gathered array below 4 GB: 32-bit offsets)
    if (tid < 128) {
      const int r = (m0 + tid < mlim) ? rowmap[m0 + tid] : -1;
      rowoff[tid] = (long long)max(r, 0) * (long long)lda;
    }
    __syncthreads();
  }
  f64x4 acc[4][4];
  g64_zero(acc);
  const G64Seg s0{rowmap ? A : A + (size_t)m0 * lda, lda, B + (size_t)n0 * ldb, ldb, K / G64_KC};
  const G64Seg s1{A, 0, B, 0, 0};
  g64_tile<false>(acc, s0, s1, g64_sm, rowmap ? rowoff : nullptr);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rt = wr * 64 + i * 16 + (lane >> 4) + 4 * r;
      long long crow = m0 + rt;
      if (rowmap) {
        const int pr = (m0 + rt < mlim) ? rowmap[m0 + rt] : -1;
        if (pr < 0) continue;
        crow = pr;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) C[(size_t)crow * ldc + n0 + wc * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    }
}

}  // namespace nnmpc
