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
        acc[mi][mj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[mj][s], a[mi][s], acc[mi][mj], 0, 0, 0);   // transposed block: lane = row of C
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

// C = act(A W' + bias); A [M][K] bf16, Wt [ntn * NB][K] bf16 (rows beyond the layer's width are zero);
// C bf16 (hidden layers) or f32 (head), only its first ldc columns exist (ldc a multiple of 4): the last
// column tile may be partial, a wave whose columns all lie beyond ldc skips its MFMAs.
// 1-D grid of ntm * ntn workgroups.  Workgroups go to the 8 XCDs round-robin by id, and each XCD has its own
// L2: id -> (row panel, column tile) is chosen so that the ntn column tiles of one row panel run back to
// back on ONE XCD -- the A panel comes from HBM once and from that L2 for the other ntn - 1 tiles.
// The MFMAs compute the transposed 32 x 32 blocks (operands swapped): a lane then holds 4 CONSECUTIVE
// columns of one row per register quad, so the epilogue stores 8 (bf16) / 16 (f32) bytes per lane.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// OUT: 0 f32, 1 bf16, 2 split bf16 -- two planes of ldc columns in rows of 2 ldc: [hi | lo], hi = bf16(x), lo = bf16(x - hi).
// Such rows are the A operand of a next layer run with nk0 > 0 (the planes have nk0 K-chunks each): its K loop walks the planes
// hi, hi, lo against weights stacked [hi ; lo ; hi], so that ONE bf16 GEMM of three times the depth forms
// hi hi' + hi lo' + lo hi' -- the product to ~2^-17 relative (only lo lo' is dropped): f32-grade results from the bf16 matrix
// pipes (nnmpc_nn_create, use_bf16 = 2).
__device__ __forceinline__ int split_chunk(int c, int nk0) {   // K-chunk c of [hi, hi, lo] -> chunk of the two-plane row [hi | lo]
  return nk0 == 0 ? c : (c < nk0 ? c : c - nk0);
}
template <int NB, bool RELU, bool BIAS, int OUT>
__global__ __launch_bounds__(256) void gemm_nt_bf16_k(void* __restrict__ Cv, int ldc,
                                                      const bf16raw* __restrict__ A, size_t lda,
                                                      const bf16raw* __restrict__ B, size_t ldb, int K,
                                                      const float* __restrict__ bias, int ntm, int ntn, int nk0 = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bf16raw* lds = reinterpret_cast<bf16raw*>(lds_raw);
  using Cf = TileCfg16<NB>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int tm, tn;
  {
    const int bid = blockIdx.x, full = (ntm >> 3) * 8 * ntn;
    if (bid < full) { const int sq = bid >> 3; tm = (sq / ntn) * 8 + (bid & 7); tn = sq % ntn; }
    else { const int rem = bid - full; tm = (ntm >> 3) * 8 + rem / ntn; tn = rem % ntn; }
  }
  const int m0 = tm * NB, n0 = tn * NB;
  const bool live = n0 + wc * Cf::WT < ldc;                 // wave-uniform
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
      load_chunk16<NB>(ra, Ag + split_chunk(kc + 1, nk0) * KC16, lda, tid);
      load_chunk16<NB>(rb, Bg + (kc + 1) * KC16, ldb, tid);
    }
    if (live) mma_chunk16<NB>(acc, sA, sB, wr, wc, lane);
  }
  if (!live) return;
  const int row0 = m0 + wr * Cf::WT + (lane & 31);
#pragma unroll
  for (int mj = 0; mj < Cf::MT; ++mj)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wc * Cf::WT + mj * 32 + 8 * j + 4 * (lane >> 5);
      if (col < ldc) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (BIAS) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
        for (int mi = 0; mi < Cf::MT; ++mi) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[mi][mj][4 * j + e] + bv[e];
            v[e] = RELU ? (x > 0.f ? x : 0.f) : x;
          }
          const size_t o = (size_t)(row0 + mi * 32) * (OUT == 2 ? 2 * ldc : ldc) + col;
          if (OUT == 2) {
            const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            const bf16x4 l = {(__bf16)(v[0] - (float)h[0]), (__bf16)(v[1] - (float)h[1]), (__bf16)(v[2] - (float)h[2]), (__bf16)(v[3] - (float)h[3])};
            __bf16* cp = reinterpret_cast<__bf16*>(Cv) + o;
            *reinterpret_cast<bf16x4*>(cp) = h;
            *reinterpret_cast<bf16x4*>(cp + ldc) = l;
          } else if (OUT == 1) {
            bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(Cv) + o) = h;
          } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cv) + o) = v;
          }
        }
      }
    }
}


// ---- wide-tile variant for the hidden layers: 256 rows x 208 columns per workgroup (832 = 4 x 208: the CDU
// widths need no column padding), 512 threads = 8 waves as 4 (M) x 2 (N): a wave owns 64 rows x 7 (wn = 0) or
// 6 (wn = 1) column tiles of v_mfma_f32_16x16x32_bf16 -- at most 112 accumulator registers, so two waves share a
// SIMD (one workgroup per CU) and cover each other's LDS latency.
// Why: the 128 x 128 kernel stages 32 KB per 2 MFLOP (64 flop/B: 39 TB/s of L2 reads at the bf16 peak) and has
// 512 MFMA cycles per K-chunk to hide a global load; this tile stages 58 KB per 6.8 MFLOP (117 flop/B) and runs
// ~1700 MFMA cycles per chunk and SIMD.
// LDS image per operand and K-chunk of 64: [rows][64 bf16], 128-byte rows, 16-byte slot s of row r stored at
// slot s ^ (r & 7): a wave's ds_read_b128 of one fragment (lane l: row l & 15, slot 4 ks + (l >> 4)) is
// conflict-free in each of the instruction's four 16-lane groups, and so are the 8-lane groups of the
// ds_write_b128 that fill it (8 lanes = the 8 slots of one row).
// Operands swapped like above: a lane ends up with 4 consecutive columns of one row of C.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int WBM = 256, WBN = 208, WNT = 7;               // WNT: column tiles of the wn = 0 waves (wn = 1: 6)
constexpr int W_STAGE = (WBM + 256) * 64;                  // bf16 elements per stage (A, then B padded to 256 rows: staging writes need no branch)
constexpr int W_LDS_BYTES = 2 * W_STAGE * 2 + 1024;        // two stages + the workgroup's bias values

// SPLIT: input AND output rows are the two planes [hi | lo] of gemm_nt_bf16_k's OUT = 2 (A: rows of lda = 2 K / 3 elements, the K
// loop walks hi, hi, lo; C: rows of 2 ldc elements).
template <bool RELU, bool BIAS, bool SPLIT = false>
__global__ __launch_bounds__(512) void gemm_nt_bf16_wide_k(__bf16* __restrict__ C, int ldc,
                                                           const bf16raw* __restrict__ A, size_t lda,
                                                           const bf16raw* __restrict__ B, size_t ldb, int K,
                                                           const float* __restrict__ bias, int ntm, int ntn, int npg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bf16raw* lds = reinterpret_cast<bf16raw*>(lds_raw);
  float* sbias = reinterpret_cast<float*>(lds_raw + 2 * W_STAGE * 2);   // this workgroup's 208 bias values
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  // Persistent workgroups (grid = 8 XCDs x npg panel groups x ntn column tiles, at most one per CU): workgroup w keeps
  // ONE column tile (its 208 weight rows stay L2-hot, its bias sits in LDS) and walks row panels
  //   p = (w & 7) + 8 q + 8 npg it,   q = (w >> 3) / ntn,   it = 0, 1, ...
  // The K-chunks of all its panels form ONE software pipeline: the first chunks of the next panel are loaded while
  // the last chunks of this one are multiplied and its result is stored (a workgroup per tile spent a quarter of its
  // time filling and draining the pipeline).  Workgroups go to the XCDs round-robin by id, so the ntn workgroups that
  // share a panel run on the SAME XCD at the same time: the panel comes from HBM once, then from that L2.
  const int wq = (int)blockIdx.x >> 3, tn = wq % ntn;
  const int p0 = ((int)blockIdx.x & 7) + 8 * (wq / ntn), pstride = 8 * npg;
  if (p0 >= ntm) return;
  const int np = (ntm - p0 + pstride - 1) / pstride;        // panels of this workgroup
  const int n0 = tn * WBN;
  if (tid < WBN) sbias[tid] = (BIAS && n0 + tid < ldc) ? bias[n0 + tid] : 0.f;
  f32x4 acc[4][WNT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // staging: piece f = tid + 512 i -> row f >> 3 = (tid >> 3) + 64 i, slot f & 7 = tid & 7.  Buffer loads: the
  // per-thread byte offset is loop-invariant (one VGPR per operand), row block and K-chunk go into the scalar
  // offset -- no vector address arithmetic in the loop (hipcc recomputes global_load addresses into the registers
  // of the previous loads and guards that with vmcnt(0), which would drain the chunk in flight).  The 4th B piece
  // (rows 192..255 of a 208-row tile) is loaded without a branch: it reads the next tile's rows, or the 64 zero rows
  // the caller appends to the weights (nnmpc_nn_create), and lands in the padding rows of the stage.
  const int prow = tid >> 3, pslot = tid & 7;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16raw*>(A), 0, (int)((size_t)ntm * WBM * lda * 2), 0x00020000);   // the caller keeps this below 2^31
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16raw*>(B), 0, (int)((size_t)ntn * WBN * ldb * 2), 0x00020000);
  const int voA = (int)((prow * lda + pslot * 8) * 2), voB = (int)((prow * ldb + pslot * 8) * 2);
  const int rsA = (int)(64 * lda * 2), rsB = (int)(64 * ldb * 2);   // bytes per block of 64 rows
  const int soff = prow * 64 + ((pslot ^ (prow & 7)) * 8);  // + 64 rows * 64 per i
  // Registers in flight: TWO sets for A (activations, streamed from HBM: the loads of chunk c + 2 are issued before
  // the MFMAs of chunk c -- two chunk times, ~3500 cycles) and ONE for B (weights, L2 hits: loaded one chunk ahead,
  // right after the previous chunk's registers went to LDS).  Two full sets would not fit 256 registers.
  u32x4 ra0[4], ra1[4], rb[4];
  auto gloadA = [&](u32x4 (&ra)[4], int off) {               // off: byte offset of (panel, chunk)
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rA, voA, i * rsA + off, 0);
  };
  auto gloadB = [&](int off) {                              // off: byte offset of (tile, chunk)
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, voB, i * rsB + off, 0);
  };
  auto swrite = [&](const u32x4 (&ra)[4], bf16raw* sA) {
    bf16raw* sB = sA + WBM * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(sA + soff + 64 * 64 * i) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(sB + soff + 64 * 64 * i) = rb[i];   // rows 208..255: padding
  };
  const int lr = lane & 15, g = lane >> 4;
  // One K-chunk: two steps of 32.  STAGE: the LDS writes of the next chunk's registers and the loads of the B chunk
  // after it are issued BETWEEN the MFMAs of the second step (sched_group_barrier: 2 MFMAs, then one of them) --
  // left to the end of the chunk, all 8 waves would write at once with the matrix pipes idle (58 KB at ~79 B/clk).
  auto kstep = [&](const bf16raw* sA, const bf16raw* sB, int ks, auto&& stage) {
    const int so = ((4 * ks + g) ^ (lr & 7)) * 8;
    bf16x8 af[4], bf[WNT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const bf16x8*>(sA + 16 * mt * 64 + so);
#pragma unroll
    for (int nt = 0; nt < WNT; ++nt)
      if (nt < WNT - 1 || wn == 0) bf[nt] = *reinterpret_cast<const bf16x8*>(sB + 16 * nt * 64 + so);
#pragma unroll
    for (int nt = 0; nt < WNT - 1; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt], af[mt], acc[mt][nt], 0, 0, 0);
    stage();
    if (wn == 0) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        acc[mt][WNT - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[WNT - 1], af[mt], acc[mt][WNT - 1], 0, 0, 0);
    }
  };
  auto compute = [&](const bf16raw* st, auto&& stage) {
    const bf16raw* sA = st + (64 * wm + lr) * 64;
    const bf16raw* sB = st + WBM * 64 + (16 * WNT * wn + lr) * 64;
    kstep(sA, sB, 0, [] {});
    kstep(sA, sB, 1, stage);
  };
  const int nk = K / 64;
  bf16raw* st0 = lds;
  bf16raw* st1 = lds + W_STAGE;
  // Loads and LDS writes are unconditional inside the loop (s_waitcnt counts are static: a conditional load would
  // force vmcnt(0) before every LDS write).  Past the last step the load cursors wrap (A) / stay on the last tile
  // (B): valid bytes, written to a stage that is never computed.  Order of the loads in flight at an LDS write of
  // step s + 1: A(s + 1), B(s + 1), A(s + 2).
  const int S = np * nk;                                    // steps = panels x chunks
  const int panelA = (int)(WBM * lda * 2), offB = tn * (int)(WBN * ldb * 2);
  int ca = 0, pa = 0, oa = p0 * panelA, cb = 0;             // load cursors: chunk / panel / panel byte offset of the next A load; chunk of the next B load
  const int nk0 = SPLIT ? nk / 3 : 0;                       // chunks per plane of a split input row
  auto nextA = [&](u32x4 (&ra)[4]) {
    gloadA(ra, oa + split_chunk(ca, nk0) * 128);
    if (++ca == nk) { ca = 0; if (pa + 1 < np) { ++pa; oa += pstride * panelA; } }
  };
  auto nextB = [&] {
    gloadB(offB + cb * 128);
    if (++cb == nk) cb = 0;
  };
  auto interleave = [] {
    // 24 MFMAs of the step are in this scheduling region: (2 MFMA, 1 LDS write) x 8, (2 MFMA, 1 buffer load) x 4
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  };
  // ---- end of a tile.  Straight from the accumulators a store instruction would write 16 rows x 32 bytes (a quarter
  // of a cache line per row); instead every wave transposes its own 64 x 112 block, 32 rows at a time, through a
  // private region of the stage that was just multiplied (free: the barrier of the step is behind us; the OTHER stage
  // already holds the next tile's first chunk) and stores 16-byte pieces of whole rows (224 contiguous bytes per row).
  // DS operations of a wave run in order, so only the hand-back of the stage needs a workgroup barrier.
  // Row stride 240 B: 2-way conflicts on the 8-byte writes, 16-byte aligned reads.
  constexpr int ESTRIDE = 240;
  const int npc = wn ? 2 * (WNT - 1) : 2 * WNT;             // 16-byte pieces per row of this wave's block
  auto epilogue = [&](int pc, bf16raw* stage) {
    unsigned char* er = reinterpret_cast<unsigned char*>(stage) + wave * (32 * ESTRIDE);
    const int m0 = (p0 + pstride * pc) * WBM;
    const size_t ldr = SPLIT ? (size_t)2 * ldc : (size_t)ldc;   // elements per row of C
    __bf16* Cw = C + (size_t)(m0 + 64 * wm) * ldr + n0 + 16 * WNT * wn;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll 1
      for (int plane = 0; plane < (SPLIT ? 2 : 1); ++plane) { // 0: hi (bf16 of the value) -> columns [0, ldc);
#pragma unroll                                              // 1 (SPLIT): lo (bf16 of value - hi) -> [ldc, 2 ldc)
        for (int nt = 0; nt < WNT; ++nt) {
          if (nt < WNT - 1 || wn == 0) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + 16 * (WNT * wn + nt) + 4 * g);
#pragma unroll
            for (int m2 = 0; m2 < 2; ++m2) {
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x = acc[2 * h + m2][nt][e] + bv[e];
                v[e] = RELU ? (x > 0.f ? x : 0.f) : x;
                if (!SPLIT || plane == 1) acc[2 * h + m2][nt][e] = 0.f;
              }
              bf16x4 hv = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
              if (SPLIT && plane == 1) hv = bf16x4{(__bf16)(v[0] - (float)hv[0]), (__bf16)(v[1] - (float)hv[1]), (__bf16)(v[2] - (float)hv[2]), (__bf16)(v[3] - (float)hv[3])};
              *reinterpret_cast<bf16x4*>(er + (16 * m2 + lr) * ESTRIDE + 32 * nt + 8 * g) = hv;
            }
          }
        }
        __bf16* Cp = Cw + (SPLIT && plane == 1 ? (size_t)ldc : 0);
        int row = lane / npc, c = lane - row * npc;          // piece q = lane + 64 i -> (row, c) = (q / npc, q % npc)
        const int dr = 64 / npc, dc = 64 % npc;
#pragma unroll 1
        for (int i = 0; i < WNT; i += 2) {                   // two pieces per trip (their LDS reads overlap); not unrolled
          int row1 = row + dr, c1 = c + dc;                  // further: the registers belong to the loads in flight
          if (c1 >= npc) { c1 -= npc; ++row1; }
          const bool ok0 = row < 32, ok1 = i + 1 < WNT && row1 < 32;
          u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
          if (ok0) v0 = *reinterpret_cast<const u32x4*>(er + row * ESTRIDE + 16 * c);
          if (ok1) v1 = *reinterpret_cast<const u32x4*>(er + row1 * ESTRIDE + 16 * c1);
          const bool in0 = ok0 && n0 + 16 * WNT * wn + 8 * c < ldc, in1 = ok1 && n0 + 16 * WNT * wn + 8 * c1 < ldc;
          if (in0) *reinterpret_cast<u32x4*>(Cp + (size_t)(32 * h + row) * ldr + 8 * c) = v0;
          if (in1) *reinterpret_cast<u32x4*>(Cp + (size_t)(32 * h + row1) * ldr + 8 * c1) = v1;
          row = row1 + dr; c = c1 + dc;
          if (c >= npc) { c -= npc; ++row; }
        }
      }
    }
    __syncthreads();                                        // the stage goes back to the pipeline
  };
  int cs = 0, pcur = 0;                                     // chunk / panel (of this workgroup) being multiplied
  auto endstep = [&](bf16raw* stage) {
    if (++cs == nk) { epilogue(pcur, stage); cs = 0; ++pcur; }
  };
  nextA(ra0);
  nextB();
  nextA(ra1);
  swrite(ra0, st0);
  nextB();
  __syncthreads();
  int sp = 0;
  for (; sp + 1 < S; sp += 2) {
    nextA(ra0);
    compute(st0, [&] { swrite(ra1, st1); nextB(); interleave(); });
    __syncthreads();
    endstep(st0);
    nextA(ra1);
    compute(st1, [&] { swrite(ra0, st0); nextB(); interleave(); });
    __syncthreads();
    endstep(st1);
  }
  if (sp < S) {                                             // odd number of steps: the last one is in stage 0
    compute(st0, [] {});
    __syncthreads();
    endstep(st0);
  }
}

}  // namespace nnmpc
