// Multiplier systems of 145 .. 256 active bounds (10 .. 16 blocks of 16): ONE WORKGROUP OF FOUR WAVES PER PROBLEM, every
// tile in registers.  (Included by qp_asm.h after the single-wave kernels, whose building blocks it uses.)
//
// Why: a single wave holds at most 32 + a few LDS-resident tiles (asm_lambda_reg: <= 11 blocks, and from 10 blocks on at half
// the rate of the 9-block class), and the LDS-tile kernels beyond (asm_lambda_tile32_k, asm_lambda_tile_k) move every
// operand of every MFMA through LDS behind three barriers per block column: 7 - 9 TFLOP/s at 192 .. 256 bounds against the
// 51 TFLOP/s of the 9-block register kernel (scripts/micro/lambda_micro.hip).  Here the lower tiles of an MB-block set are
// spread over the four waves BY BLOCK ROW, so that
//   * TRSM  L(I,K)' = Y_K S(I,K)'  and the trailing update  S(I,J)' += L(J,K) L(I,K)'  run MFMA register to register exactly
//     as in asm_lambda_reg (the accumulator of a transposed tile IS its operand fragment); the only operand a wave does not
//     own, L(J,K) of another wave's row J, is read from a panel in LDS (one ds_read_b128 per tile and block column);
//   * the forward substitution rides along in the owner of each row, the backward one sums the four waves' partial
//     products of a block column through 64 LDS words;
//   * the owner of the next diagonal tile updates that tile first and runs its 16-step pivot chain (asm_diag16) in the same
//     straight-line stretch of code as its share of the trailing update, while the other three waves do theirs: two
//     workgroup barriers per block column.
// Rows are dealt top-down in snake order (longest row first: waves 0 1 2 3 3 2 1 0 0 1 ...), so that every wave owns the same
// number of tiles to within the length of one row.  Slot a of a wave holds ONE row of at most MB - 4a tiles; the wave whose
// row in that slot is shorter carries up to three tiles ABOVE the diagonal along (never read by anything that counts): with
// them the block-column loop has no branch that depends on the wave -- which is what lets the compiler interleave MFMAs and
// the pivot chain -- at 18 % (16 blocks) to 33 % (10 blocks) more MFMA work, which is not what bounds the kernel.
// One instantiation per number of blocks (the owner of every block row is then a compile-time constant).
#pragma once

namespace nnmpc {

constexpr int ASM_WG_MB = 16;                              // blocks of the largest set
#ifndef ASM_WG_GUARD32
#define ASM_WG_GUARD32 15          // (see GUARD in asm_lambda_wg; overridden by scripts/micro for A/B runs)
#endif
#ifndef ASM_WG_GUARD64
#define ASM_WG_GUARD64 13
#endif
constexpr int ASM_WG_MAX = 16 * ASM_WG_MB;
constexpr int ASM_WG_MB8 = 24;                             // ... of the eight-wave f32 instances (257 .. 384 bounds)
__host__ __device__ constexpr int asm_wg_off(int MB, int NW, int a) { int o = 0; for (int b = 0; b < a; ++b) o += MB - NW * b; return o; }   // first tile of slot a
__host__ __device__ constexpr int asm_wg_tiles(int MB, int NW) { return asm_wg_off(MB, NW, (MB + NW - 1) / NW); }
__host__ __device__ constexpr int asm_wg_owner(int MB, int NW, int K) {   // wave that owns block row K (its slot: (MB - 1 - K) / NW)
  const int rp = MB - 1 - K, a = rp / NW;
  return (a & 1) ? NW - 1 - rp % NW : rp % NW;
}
// LDS, in elements of T: diagonal tile, its inverse, identity tile | y [ROWS] | rhs [ROWS] | partial sums [2][8 waves][16] |
// panel [MB8 + 1][256] (the last tile: where the rows above the diagonal go); then ints: index list [ROWS], flag
// (MBX: the largest number of blocks of the kernel's instances -- 16 for the four- and two-wave kernels, 24 for the eight-wave one)
template <class T, int MBX> constexpr int asm_wg_lds_bytes_refine();
template <class T, int MBX = ASM_WG_MB> constexpr int asm_wg_lds_bytes() { return (3 * ASM_TS + 2 * 16 * MBX + 256 + (MBX + 1) * 256) * (int)sizeof(T) + (16 * MBX + 4) * 4; }

// REFINE (f32 instances only): the fp64 solve of a set too large for fp64 tiles in registers -- the f32 factor stays in the
// registers and the multipliers are refined against fp64 residuals  r = b - S lam  (S gathered again from the fp64 inverse, row by
// row, lanes along the row; b, lam, r in fp64) until max |r| <= 1e-13 max |b|: an fp64 RESULT from f32 arithmetic on the tiles.
// A set it cannot bring there in ASM_WG_NREF corrections (cond(S) 6e-8 per step) is handed to the fp64 slab kernel (prec = 2).
constexpr int ASM_WG_NREF = 5;
template <class T, int MBX> constexpr int asm_wg_lds_bytes_refine() { return asm_wg_lds_bytes<T, MBX>() + (2 * 16 * MBX + 16) * 8 + 16; }
template <class T, int MB, int NW, bool REFINE = false>
__device__ __forceinline__ void asm_lambda_wg(const AsmDev& d, int p, int m) {
  using N = AsmNum<T>;
  using V4 = typename N::v4;
  constexpr int NS = (MB + NW - 1) / NW, NT = asm_wg_tiles(MB, NW), NTH = 64 * NW;
  // GUARD: the largest sets skip the tiles above the diagonal behind wave-uniform branches instead of carrying them along: the
  // branches cut the code into small scheduling regions, which is what keeps 40 tiles + working set inside the registers there
  // (branch-free, the 15- and 16-block f32 instances spill ~220 registers and run at 4.97 instead of 7.1 problems/us)
  constexpr bool GUARD = N::F32 ? MB >= ASM_WG_GUARD32 : MB >= ASM_WG_GUARD64;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
  T* dt = reinterpret_cast<T*>(sm_raw);                    // diagonal tile, row-major stride 17
  T* Yt = dt + ASM_TS;                                     // its inverse factor
  T* idt = Yt + ASM_TS;                                    // identity tile
  T* ys = idt + ASM_TS;                                    // y (forward result) [16][16]
  constexpr int MBX = MB > ASM_WG_MB ? ASM_WG_MB8 : ASM_WG_MB, ROWS = 16 * MBX;   // (the LDS layout of the kernel this instance belongs to)
  T* rv = ys + ROWS;                                       // right-hand side
  T* part = rv + ROWS;                                     // backward substitution: [2][NW waves][16]
  T* panel = part + 256;                                   // L(J,K), J > K, of the current block column: tile J at 256 J, lane-major V4
  int* ix = reinterpret_cast<int*>(panel + (MBX + 1) * 256);   // active indices (padded with the last one)
  int* s_bad = ix + ROWS;
  double* b64 = reinterpret_cast<double*>(s_bad + 4);      // REFINE only (asm_wg_lds_bytes_refine): rhs in fp64 [ROWS], multipliers [ROWS], maxima [2 NW]
  double* lam64 = b64 + ROWS;
  double* red = lam64 + ROWS;
  constexpr int PSPARE = MBX;                              // panel tile that takes the rows above the diagonal
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const size_t o = (size_t)p * d.np;
  const unsigned char* st = d.st + (size_t)p * d.n;
  const int* idx = d.idxg + (size_t)p * d.max_active;
#ifdef ASM_STAMPS
  const int wg = blockIdx.x;
#endif
  ASM_STAMP(0);
  for (int i = tid; i < 16 * MB; i += NTH) {
    const int a = idx[min(i, m - 1)], k = a % d.nu;
    ix[i] = a;
    const double v = d.xunc[o + a] - (st[a] == 1 ? d.ub[(size_t)p * d.nu + k] : d.lb[(size_t)p * d.nu + k]);
    rv[i] = i < m ? (T)v : T(0);
    if constexpr (REFINE) { b64[i] = i < m ? v : 0.0; lam64[i] = 0.0; }
  }
  for (int j = tid; j < ASM_TS; j += NTH) idt[j] = (j / 17 == j % 17) ? T(1) : T(0);
  if (tid == 0) *s_bad = 0;
  __syncthreads();
  // the row of slot a: position NW a + (a odd ? NW - 1 - wave : wave) from the top, i.e. row MB - 1 - position (negative: none)
  int Ia[NS];
#pragma unroll
  for (int a = 0; a < NS; ++a) Ia[a] = MB - 1 - (NW * a + ((a & 1) ? NW - 1 - wave : wave));
  // ---- gather: tile (I,J)[r] = S[16 I + li][16 J + kr(lq, r)]  (PLUS the Schur complement, unlike asm_lambda_reg: the MFMAs have
  // no negate modifier, so the trailing update negates the panel tile it has just read -- four v_xor per tile and column; negating
  // the gathered values kept a second copy of them alive until their first use, i.e. spills at 15 - 16 blocks);
  // beyond m the identity: 16 (MB - 1) < m <= 16 MB (asm_lambda_wg_any), so only the last block row -- slot 0 of wave 0 --
  // reaches beyond m, and only its loads go through a select (a select per load costs a register per load in flight)
  V4 C[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) C[t] = V4{T(0), T(0), T(0), T(0)};
  {
    using HT = typename std::conditional<N::F32, float, double>::type;
    const char* const Hbase = N::F32 ? reinterpret_cast<const char*>(d.H32) : reinterpret_cast<const char*>(d.H);
    asm_sfor<0, NS>([&](auto ac) __attribute__((always_inline)) {
      constexpr int a = decltype(ac)::value;
      constexpr int oa = asm_wg_off(MB, NW, a);
      const int I = Ia[a];
      if (I >= 0) {
        const int gi = 16 * I + li;
        const unsigned gco = (unsigned)ix[gi] * (unsigned)sizeof(HT);
        const bool edge = a == 0 && I == MB - 1;
        asm_sfor<0, MB - NW * a>([&](auto Jc) __attribute__((always_inline)) {
          constexpr int J = decltype(Jc)::value;
          if (J <= I) {
            if (edge) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int gj = 16 * J + N::kr(lq, r);
                const unsigned rowoff = (unsigned)ix[gj] * (unsigned)d.np * (unsigned)sizeof(HT);
                const HT v = *reinterpret_cast<const HT*>(Hbase + (rowoff + gco));
                C[oa + J][r] = (gi < m && gj < m) ? (T)v : (gi == gj ? T(1) : T(0));
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int gj = 16 * J + N::kr(lq, r);
                const unsigned rowoff = (unsigned)ix[gj] * (unsigned)d.np * (unsigned)sizeof(HT);
                C[oa + J][r] = (T)*reinterpret_cast<const HT*>(Hbase + (rowoff + gco));
              }
            }
          }
          if constexpr (J % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // (a few tiles' addresses at a time: registers)
        });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  ASM_STAMP(1);
  T ps[NS];                                                // lane-local partial sums of  sum_J L(I,J) y_J  of the own rows
#pragma unroll
  for (int a = 0; a < NS; ++a) ps[a] = T(0);
  int bad = 0;
  // owner of block row K: slot and wave, compile-time
  // ---- diagonal tile 0
  {
    constexpr int a0 = (MB - 1) / NW, w0 = asm_wg_owner(MB, NW, 0);
    if (wave == w0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) dt[li * 17 + N::kr(lq, r)] = C[asm_wg_off(MB, NW, a0)][r];
      ASM_FENCE();
      bad |= asm_diag16<T>(dt, Yt, lane, idt);
      ASM_FENCE();
    }
  }
  asm_sfor<0, MB>([&](auto Kc) __attribute__((always_inline)) {
    constexpr int K = decltype(Kc)::value;
    constexpr int aK = (MB - 1 - K) / NW, wK = asm_wg_owner(MB, NW, K), oK = asm_wg_off(MB, NW, aK);
    ASM_STAMP(4 + 3 * (K % 9));
    if (wave == wK) {                                      // Y_K' replaces the diagonal tile; y_K = Y_K (r_K - sum_{J<K} L(K,J) y_J)
      V4 Yc;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) Yc[s4] = Yt[N::kr(lq, s4) * 17 + li];
      C[oK + K] = Yc;
      const T tK = rv[16 * K + li] - (K ? xsum4<T>(ps[aK]) : T(0));
      T yq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) yq[r] = Yc[r] * tK;
      rowsum16x4<T>(yq);
      if (li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ys[16 * K + N::kr(lq, r)] = yq[r];
      }
      if (bad && lane == 0) *s_bad = 1;
    }
    __syncthreads();                                       // Y_K and y_K are in LDS
    if constexpr (K + 1 < MB) {
      // TRSM of the own rows below K:  L(I,K)' = Y_K S(I,K)', into the panel for the other waves
      {
        T yf[4], yq[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { yf[s4] = Yt[li * 17 + N::kr(lq, s4)]; yq[s4] = ys[16 * K + N::kr(lq, s4)]; }
        asm_sfor<0, NS>([&](auto ac) __attribute__((always_inline)) {
          constexpr int a = decltype(ac)::value;
          constexpr int oa = asm_wg_off(MB, NW, a);
          if constexpr (K < MB - NW * a) {
            const bool real = Ia[a] > K;                   // (else: the diagonal tile = Y', or a tile above the diagonal)
            if (!GUARD || real) {
              const V4 b = C[oa + K];
              V4 acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4) acc = N::mfma(yf[s4], b[s4], acc);
              T dot = T(0);
#pragma unroll
              for (int r = 0; r < 4; ++r) { C[oa + K][r] = (GUARD || real) ? acc[r] : b[r]; dot += acc[r] * yq[r]; }
              ps[a] += (GUARD || real) ? dot : T(0);
              *reinterpret_cast<V4*>(panel + ((GUARD || real) ? Ia[a] : PSPARE) * 256 + lane * 4) = acc;
            }
          }
        });
      }
      __syncthreads();                                     // the panel of column K is in LDS
      __builtin_amdgcn_sched_barrier(0);
      ASM_STAMP(5 + 3 * (K % 9));
      constexpr int a1 = (MB - 2 - K) / NW, w1 = asm_wg_owner(MB, NW, K + 1), o1 = asm_wg_off(MB, NW, a1);
      // trailing update: tile (slot a, J) -= L(J,K) L(row of a, K)',  K < J (tiles above the diagonal included)
      auto trailing = [&](auto skipc, auto Jc) __attribute__((always_inline)) {   // block column J of the update
        constexpr bool skip = decltype(skipc)::value != 0;
        constexpr int J = decltype(Jc)::value;
        if (GUARD && Ia[0] < J) return;                    // (slot 0 holds the wave's longest row)
        const V4 PJ = -*reinterpret_cast<const V4*>(panel + J * 256 + lane * 4);
        asm_sfor<0, NS>([&](auto ac) __attribute__((always_inline)) {
          constexpr int a = decltype(ac)::value;
          constexpr int oa = asm_wg_off(MB, NW, a);
          if constexpr (J < MB - NW * a && !(skip && a == a1 && J == K + 1)) {
            if (!GUARD || Ia[a] >= J) {
              const V4 l = C[oa + K];
              V4 acc = C[oa + J];
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4) acc = N::mfma(PJ[s4], l[s4], acc);
              C[oa + J] = acc;
            }
          }
        });
      };
      if (wave == w1) {
        // the next diagonal tile first, then its factorisation (a long dependent VALU chain) in one stretch of code with this
        // wave's share of the trailing update
        {
          const V4 l = C[o1 + K];
          V4 acc = C[o1 + K + 1];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) acc = N::mfma(-l[s4], l[s4], acc);
          C[o1 + K + 1] = acc;
#pragma unroll
          for (int r = 0; r < 4; ++r) dt[li * 17 + N::kr(lq, r)] = acc[r];
        }
        ASM_FENCE();
        __builtin_amdgcn_sched_barrier(0);
        // pivot step cc of the chain and block column K + 1 + cc of this wave's share of the trailing update go into one
        // scheduling region each (the MFMAs fill the chain's dependency stalls; bounded regions keep the registers bounded)
        T x[16];
        asm_diag16_begin<T>(x, dt, idt, lane);
        asm_sfor<0, 16>([&](auto cc) __attribute__((always_inline)) {
          constexpr int c = decltype(cc)::value;
          asm_diag16_step<T, c>(x);
          if constexpr (K + 1 + c < MB) trailing(asm_ic<1>{}, asm_ic<K + 1 + c>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        // (more than 17 blocks: the block columns beyond the sixteenth after the chain)
        if constexpr (K + 17 < MB) {
          asm_sfor<K + 17, MB>([&](auto Jc) __attribute__((always_inline)) {
            trailing(asm_ic<1>{}, Jc);
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        ASM_STAMP(6 + 3 * (K % 9));
        bad |= asm_diag16_end<T>(x, Yt, lane);
        ASM_FENCE();
      } else {
        asm_sfor<K + 1, MB>([&](auto Jc) __attribute__((always_inline)) {
          trailing(asm_ic<0>{}, Jc);
          __builtin_amdgcn_sched_barrier(0);
        });
        ASM_STAMP(6 + 3 * (K % 9));
      }
    }
  });
  __syncthreads();
  ASM_STAMP(40);
  if (*s_bad) {
    // f32: S is not positive definite in this precision -- the round is void (the LAM32 row is still zero), the next one
    // runs in fp64.  fp64: hand the problem to the PDIP path.
    if (tid == 0) { if (REFINE) { d.prec[p] = 2; d.redo[p] = 1; } else if (N::F32) { d.prec[p] = 1; d.redo[p] = 1; } else d.state[p] = ASM_FALLBACK; }
    return;
  }
  if ((!N::F32 || REFINE) && tid == 0) d.prec[p] = 1;      // solved in fp64 (REFINE: to fp64 residuals)
  // ---- backward substitution  L' lam = y:  lam_K = Y_K' (y_K - sum_{I>K} L(I,K)' lam_I), the sum over the rows of all four
  // waves through LDS (double-buffered by the parity of K: one barrier per block column)
  T lam[NS];
  auto backward = [&]() __attribute__((always_inline)) {
#pragma unroll
  for (int a = 0; a < NS; ++a) lam[a] = T(0);
  asm_sfor<0, MB>([&](auto Kr) __attribute__((always_inline)) {
    constexpr int K = MB - 1 - decltype(Kr)::value;
    constexpr int aK = (MB - 1 - K) / NW, wK = asm_wg_owner(MB, NW, K), oK = asm_wg_off(MB, NW, aK);
    T s4[4] = {T(0), T(0), T(0), T(0)};
    asm_sfor<0, NS>([&](auto ac) __attribute__((always_inline)) {
      constexpr int a = decltype(ac)::value;
      constexpr int oa = asm_wg_off(MB, NW, a);
      if constexpr (K < MB - NW * a) {
        const bool real = Ia[a] > K;
#pragma unroll
        for (int r = 0; r < 4; ++r) s4[r] += real ? C[oa + K][r] * lam[a] : T(0);
      }
    });
    rowsum16x4<T>(s4);
    T* pb = part + (K & 1) * (16 * NW);
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pb[wave * 16 + N::kr(lq, r)] = s4[r];
    }
    __syncthreads();
    if (wave == wK) {
      const V4 Yc = C[oK + K];
      T acc = T(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = N::kr(lq, r);
        T sum = pb[k];
#pragma unroll
        for (int w = 1; w < NW; ++w) sum += pb[16 * w + k];
        const T t = ys[16 * K + k] - sum;
        acc += Yc[r] * t;                                  // Y_K[k][li] t[k]
      }
      lam[aK] = xsum4<T>(acc);
    }
  });
  };
  backward();
  if constexpr (REFINE) {
    // ---- fp64 refinement on the f32 factor
    bool conv = false;
    for (int itr = 0; itr <= ASM_WG_NREF && !conv; ++itr) {
      // lam64 += the correction just solved for (the first pass: the f32 solution itself)
#pragma unroll
      for (int a = 0; a < NS; ++a) {
        const int i = 16 * Ia[a] + li;
        if (Ia[a] >= 0 && lq == 0 && i < m) lam64[i] += (double)lam[a];
      }
      __syncthreads();
      // r = b - S lam64: row i by wave i % NW, lanes along the row (the gathered entries of one row of the inverse lie within a few KB)
      double rmax = 0.0, bmax = 0.0;
      // (four rows at a time: their gathers -- L2 / Infinity Cache round trips -- are all in flight before the first product; one row
      // at a time the loop was a chain of ~0.5 us waits, 20 us per residual)
      for (int i0 = wave; i0 < 16 * MB; i0 += 4 * NW) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        const double* Hr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) Hr[u] = d.H + (size_t)ix[min(i0 + u * NW, m - 1)] * d.np;
        for (int j = lane; j < m; j += 64) {
          const int cj = ix[j];
          const double lj = lam64[j];
          double hv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) hv[u] = Hr[u][cj];
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] += hv[u] * lj;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[u] += __shfl_xor(acc[u], off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * NW;
          if (i < 16 * MB) {
            const double r = i < m ? b64[i] - acc[u] : 0.0;
            if (lane == 0) rv[i] = (T)r;
            rmax = fmax(rmax, fabs(r)); bmax = fmax(bmax, i < m ? fabs(b64[i]) : 0.0);
          }
        }
      }
      if (lane == 0) { red[wave] = rmax; red[NW + wave] = bmax; }
      __syncthreads();
      double rm = 0.0, bm = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { rm = fmax(rm, red[w]); bm = fmax(bm, red[NW + w]); }
      conv = rm <= 1e-13 * bm || !(rm == rm);             // (a NaN leaves the loop: handled below)
      if (!(rm == rm)) { conv = false; break; }
      if (conv || itr == ASM_WG_NREF) break;
      // correction: forward substitution of r (the stored L and Y_K, one barrier per block column), then backward
      T ps2[NS];
#pragma unroll
      for (int a = 0; a < NS; ++a) ps2[a] = T(0);
      asm_sfor<0, MB>([&](auto Kc) __attribute__((always_inline)) {
        constexpr int K = decltype(Kc)::value;
        constexpr int aK = (MB - 1 - K) / NW, wK = asm_wg_owner(MB, NW, K), oK = asm_wg_off(MB, NW, aK);
        if (wave == wK) {
          const V4 Yc = C[oK + K];
          const T tK = rv[16 * K + li] - (K ? xsum4<T>(ps2[aK]) : T(0));
          T yq[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) yq[r] = Yc[r] * tK;
          rowsum16x4<T>(yq);
          if (li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ys[16 * K + N::kr(lq, r)] = yq[r];
          }
        }
        __syncthreads();
        if constexpr (K + 1 < MB) {
          T yq[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) yq[s4] = ys[16 * K + N::kr(lq, s4)];
          asm_sfor<0, NS>([&](auto ac) __attribute__((always_inline)) {
            constexpr int a = decltype(ac)::value;
            constexpr int oa = asm_wg_off(MB, NW, a);
            if constexpr (K < MB - NW * a) {
              if (Ia[a] > K) {
                T dot = T(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) dot += C[oa + K][r] * yq[r];
                ps2[a] += dot;
              }
            }
          });
        }
      });
      __syncthreads();
      backward();
    }
    if (!conv) {                                           // not brought to fp64 residuals: the round is void, the slab kernel takes the set
      if (tid == 0) { d.prec[p] = 2; d.redo[p] = 1; }
      return;
    }
    double* lrow64 = d.lam + (size_t)d.row[p] * d.np;
    for (int i = tid; i < m; i += NTH) lrow64[ix[i]] = lam64[i];
    return;
  }
  ASM_STAMP(41);
  using LT = typename std::conditional<N::F32, float, double>::type;   // f32 rounds: row of LAM32 (f32 GEMM)
  LT* lrow = (N::F32 ? (LT*)d.lam32 : (LT*)d.lam) + (size_t)d.row[p] * d.np;
#pragma unroll
  for (int a = 0; a < NS; ++a) {
    const int i = 16 * Ia[a] + li;
    if (Ia[a] >= 0 && lq == 0 && i < m) lrow[ix[i]] = (LT)lam[a];
  }
}

#if defined(ASM_WG_TU)
#define ASM_WG_DEF(n) (ASM_WG_TU == (n))        // one kernel per translation unit (see qp_asm.h); -1: declarations only
#else
#define ASM_WG_DEF(n) 1
#endif
template <class T>
__device__ __forceinline__ void asm_lambda_wg_any(const AsmDev& d, int p) {
  const int m = __builtin_amdgcn_readfirstlane(d.mg[p]);
  if (m <= 0 || m > ASM_WG_MAX) return;
  switch (max((m + 15) >> 4, 10)) {
    case 16: asm_lambda_wg<T, 16, 4>(d, p, m); break;
    case 15: asm_lambda_wg<T, 15, 4>(d, p, m); break;
    case 14: asm_lambda_wg<T, 14, 4>(d, p, m); break;
    case 13: asm_lambda_wg<T, 13, 4>(d, p, m); break;
    case 12: asm_lambda_wg<T, 12, 4>(d, p, m); break;
#ifdef ASM_WG_MICRO                                        // (scripts/micro only: the solver's lists hold sets of 177 .. 256 bounds)
    case 11: asm_lambda_wg<T, 11, 4>(d, p, m); break;
    default: asm_lambda_wg<T, 10, 4>(d, p, m); break;
#else
    default: if (threadIdx.x == 0) d.state[p] = ASM_FALLBACK; break;   // not reachable through asm_scan_col; the PDIP path if it ever is
#endif
  }
}

// f32 rounds of the sets of 177 .. 256 bounds (list ASM_NLIST), fp64 rounds (list ASM_NLIST + 1): one workgroup each.
// (The 10- and 11-block classes stay with the single-wave kernels asm_lambda_reg32b_k / asm_lambda_reg2_k: four independent
// pivot chains per CU and no barrier beat four waves on one chain there -- 17.9 against 14.9 problems per microsecond at 160
// bounds in f32; from 12 blocks on the single-wave kernels have no room.  scripts/micro/lambda_micro.hip, problems per
// microsecond at 192 / 224 / 256 bounds: f32 11.8 / 8.6 / 6.4 here against 2.9 / 2.2 / 1.65 of asm_lambda_tile32_k; fp64
// 4.2 / 2.8 / 2.3.  Eight waves per problem -- half the tiles per wave, no spills at 16 blocks -- were SLOWER: 4.5 at 256
// bounds in f32, one workgroup per CU and twice the barrier traffic.)
#if ASM_WG_DEF(0)
__global__ __launch_bounds__(256, 2) void asm_lambda_wg32_k(AsmDev d) {
  if ((int)blockIdx.x < d.counters[ASM_CNT_BIG32]) asm_lambda_wg_any<float>(d, d.binlist[(size_t)ASM_NLIST * d.nseg + blockIdx.x]);
}
#else
__global__ void asm_lambda_wg32_k(AsmDev d);
#endif
#if ASM_WG_DEF(1)
__global__ __launch_bounds__(256, 1) void asm_lambda_wg64_k(AsmDev d) {
  if ((int)blockIdx.x < d.counters[ASM_CNT_BIG64]) asm_lambda_wg_any<double>(d, d.binlist[(size_t)(ASM_NLIST + 1) * d.nseg + blockIdx.x]);
}
#else
__global__ void asm_lambda_wg64_k(AsmDev d);
#endif

// f32 rounds of the sets of 257 .. 384 bounds (list ASM_NLIST + 2): EIGHT waves per problem, one workgroup per CU.  Until round 4
// these sets ran every round -- f32 or not -- in the fp64 L2-slab kernel (1.9 us per solve at 290 bounds); an f32 result only moves
// the set, and the tiles of 17 .. 24 blocks fit the registers of eight waves in f32 (153 .. 300 tiles: at most 48 per wave).
#if ASM_WG_DEF(2)
__global__ __launch_bounds__(512, 1) void asm_lambda_wg32b_k(AsmDev d) {
  if ((int)blockIdx.x >= d.counters[ASM_CNT_BIG32B]) return;
  const int p = d.binlist[(size_t)(ASM_NLIST + 2) * d.nseg + blockIdx.x];
  const int m = __builtin_amdgcn_readfirstlane(d.mg[p]);
  if (m <= 256 || m > ASM_BIG32B) return;
  switch ((m + 15) >> 4) {
    case 24: asm_lambda_wg<float, 24, 8>(d, p, m); break;
    case 23: asm_lambda_wg<float, 23, 8>(d, p, m); break;
    case 22: asm_lambda_wg<float, 22, 8>(d, p, m); break;
    case 21: asm_lambda_wg<float, 21, 8>(d, p, m); break;
    case 20: asm_lambda_wg<float, 20, 8>(d, p, m); break;
    case 19: asm_lambda_wg<float, 19, 8>(d, p, m); break;
    case 18: asm_lambda_wg<float, 18, 8>(d, p, m); break;
    default: asm_lambda_wg<float, 17, 8>(d, p, m); break;
  }
}
#else
__global__ void asm_lambda_wg32b_k(AsmDev d);
#endif

// fp64 solves of the sets of 257 .. 384 bounds (list ASM_NLIST + 3): the same eight-wave f32 factorisation, refined to fp64
// residuals (REFINE above).  Replaces the fp64 L2-slab kernel there (1.9 us per solve at 290 bounds).
#if ASM_WG_DEF(3)
__global__ __launch_bounds__(512, 1) void asm_lambda_wg64r_k(AsmDev d) {
  if ((int)blockIdx.x >= d.counters[ASM_CNT_BIG64R]) return;
  const int p = d.binlist[(size_t)(ASM_NLIST + 3) * d.nseg + blockIdx.x];
  const int m = __builtin_amdgcn_readfirstlane(d.mg[p]);
  if (m <= 256 || m > ASM_BIG32B) return;
  switch ((m + 15) >> 4) {
    case 24: asm_lambda_wg<float, 24, 8, true>(d, p, m); break;
    case 23: asm_lambda_wg<float, 23, 8, true>(d, p, m); break;
    case 22: asm_lambda_wg<float, 22, 8, true>(d, p, m); break;
    case 21: asm_lambda_wg<float, 21, 8, true>(d, p, m); break;
    case 20: asm_lambda_wg<float, 20, 8, true>(d, p, m); break;
    case 19: asm_lambda_wg<float, 19, 8, true>(d, p, m); break;
    case 18: asm_lambda_wg<float, 18, 8, true>(d, p, m); break;
    default: asm_lambda_wg<float, 17, 8, true>(d, p, m); break;
  }
}
#else
__global__ void asm_lambda_wg64r_k(AsmDev d);
#endif

// (The same for the sets of 177 .. 256 bounds -- four waves, f32 tiles, two workgroups per CU, refined -- is no faster than the fp64
// register kernel asm_lambda_wg64_k there: 4.4 / 3.0 / 2.5 against 4.3 / 3.3 / 2.3 problems per microsecond at 192 / 224 / 256
// bounds (scripts/micro/lambda_micro.hip, variant 14 against 6): the three residual passes and two extra solves cost what the f32
// tiles save.  Not instantiated in the library.)
#ifdef ASM_WG_MICRO
__global__ __launch_bounds__(256, 2) void asm_lambda_wg64r4_k(AsmDev d) {
  if ((int)blockIdx.x >= d.counters[ASM_CNT_BIG64]) return;
  const int p = d.binlist[(size_t)(ASM_NLIST + 1) * d.nseg + blockIdx.x];
  const int m = __builtin_amdgcn_readfirstlane(d.mg[p]);
  if (m <= 176 || m > ASM_WG_MAX) return;
  switch ((m + 15) >> 4) {
    case 16: asm_lambda_wg<float, 16, 4, true>(d, p, m); break;
    case 15: asm_lambda_wg<float, 15, 4, true>(d, p, m); break;
    case 14: asm_lambda_wg<float, 14, 4, true>(d, p, m); break;
    case 13: asm_lambda_wg<float, 13, 4, true>(d, p, m); break;
    default: asm_lambda_wg<float, 12, 4, true>(d, p, m); break;
  }
}
#endif

// The 10- and 11-block classes (145 .. 176 bounds) with TWO waves per problem: the same number of problems per CU as the
// single-wave kernels (four in f32, two in fp64), half the tiles and half the trailing update per wave.  The solver uses the
// fp64 instance (7.3 / 5.2 problems per microsecond at 160 / 176 bounds, asm_lambda_reg2_k: 6.0 / 4.5); the f32 instance
// loses to asm_lambda_reg32b_k (14.3 against 18.1 at 160) and is kept for scripts/micro only.
template <class T>
__device__ __forceinline__ void asm_lambda_wg2(const AsmDev& d, int c7, int c6) {
  int w = blockIdx.x, list;
  const int n1 = d.counters[c7], n0 = d.counters[c6];
  if (w < n1) list = asm_list_of_counter(c7);
  else if ((w -= n1) < n0) list = asm_list_of_counter(c6);
  else return;
  const int p = d.binlist[(size_t)list * d.nseg + w];
  const int m = __builtin_amdgcn_readfirstlane(d.mg[p]);
  if (m <= 0 || m > 176) return;
  if (m > 160) asm_lambda_wg<T, 11, 2>(d, p, m);
  else asm_lambda_wg<T, 10, 2>(d, p, m);
}
#ifdef ASM_WG_MICRO
__global__ __launch_bounds__(128, 4) void asm_lambda_wg32s_k(AsmDev d) { asm_lambda_wg2<float>(d, ASM_CNT_F32 + 7, ASM_CNT_F32 + 6); }
#endif
#if ASM_WG_DEF(4)
__global__ __launch_bounds__(128, 2) void asm_lambda_wg64s_k(AsmDev d) { asm_lambda_wg2<double>(d, 4 + 7, 4 + 6); }
#else
__global__ void asm_lambda_wg64s_k(AsmDev d);
#endif

}  // namespace nnmpc
