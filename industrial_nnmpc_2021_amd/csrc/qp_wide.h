// Full-width pass of the active-set path: fp64 GEMMs with the check in the epilogue (gfx950).
//
// A problem that settles inside the column window [0, W) gets ONE pass over the columns beyond it:
//     x[W:] = x_unc[W:] - Pinv[W:, A] lam  =  M z,    M = [Kunc[W:] | -Pinv[W:, 0:W]],   z = [x0 ; lam[0:W]]
// feasibility of the free variables there, u written out; asm_wide_k is left with the columns inside the window, the
// multiplier statistics and the decision.  Three forms of the product, newest first:
//
//  * FAR FIELD (round 3; nnmpc_qp_set_farfield).  Beyond the last active bound the optimum follows the unconstrained
//    recursion of the terminal-cost LQ problem, so everything out there is a linear function of the (augmented) state at
//    the window's end: M has numerical rank ~Nx (CDU: 252 of 796 columns, the 253rd singular value is 4e-15 of the first).
//    With the one-time factorisation  M = U V'  (host, fp64 SVD, verified on the device: max |U V' - M| enters the
//    certificate)
//        T = [x0 | lam] V            asm_wide_t_k       rows x rp,    k = n_aug + the row block's k-range
//        x[W:] = T U'                asm_wide_gemm_k<FAR>             k = ffk[column tile] <= rp = 256
//    2.6 x fewer flops than the dense product at the CDU size -- the same numbers to rounding.  The basis is ordered so that
//    the coordinates a column tile still feels come first (the closed loop forgets its fast modes first: the row space of
//    M[j:, :] shrinks with j); U is a staircase, and a tile's K loop ends where its rows of U do (CDU, W = 512: 103 of 256 on
//    average over the 31 tiles).
//    First-move calls (NNMPC_OUT_FIRST_MOVE: nothing beyond the window is delivered, only checked) skip every 128-column
//    tile that Cauchy-Schwarz certifies:  |x_j| <= |U_j| |T_p| <= min(ub, -lb)  for all rows of the tile and all columns at or
//    beyond it (|U_j| decays geometrically along the horizon: CDU, W = 512: all but the first ~6 of 31 tiles).
//  * LAZY (round 3): one GEMM whose K loop has two segments (gemm64.h), x0 against Kunc and the multiplier row against
//    Pinv, subtracted: x_unc beyond the window never exists in HBM.
//  * round 2: x_unc for all columns up front, the tile of lamw Pinv in the accumulators, x_unc loaded in the epilogue.
#pragma once
#include "gemm64.h"
#include "qp_asm.h"

namespace nnmpc {

enum { WIDE_XUNC = 0, WIDE_LAZY = 1, WIDE_FAR = 2 };

// Rows of the pass: the fp64 rows 0..wrows-1 of LAM of the round just finished (AsmDev::rowprob: the problem that settled in the
// row, or -1); k-range of a 128-row tile: the last active bound of its two 64-row blocks (AsmDev::kblk, asm_bins_b_k).
__device__ __forceinline__ int wide_tile_k(const AsmDev& d, int tm) {
  const int kl = max(d.kblk[2 * tm], d.kblk[2 * tm + 1]);
  return ((kl + G64_KC) / G64_KC) * G64_KC;
}

// ---- epilogue shared by the three forms: x, feasibility of the free variables, u out.  All loads of a row are issued
// unconditionally and together (a branch per element would turn the epilogue into a chain of dependent round trips).
// BEYOND: every column of the tile lies beyond the window the rows settled in (c0 > 0): all of them are free variables -- no
// bound state to load -- and, nu dividing 32 or 16, a lane's four columns (16 apart) see two inputs at most.
template <bool XUNC, bool BEYOND>
__device__ __forceinline__ void wide_epilogue(const AsmDev& d, const f64x4 (&acc)[4][4], int m0, int n0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  int colj[4], kj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    colj[j] = min(n0 + wc * 64 + j * 16 + (lane & 15), d.n - 1);   // (clamped: columns >= n are padding, never stored)
    kj[j] = colj[j] % d.nu;
  }
  const bool colok = n0 + wc * 64 + 63 < d.n;                      // whole 64-column half inside the problem (wave-uniform)
  const bool pair = BEYOND && colok && (32 % d.nu == 0);           // columns j and j + 2 share their input (wave-uniform)
  int prow[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wr * 64 + i * 16 + (lane >> 4) + 4 * r;
      prow[4 * i + r] = d.rowprob[row];                            // (-1 up to the end of the last tile)
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = prow[4 * i + r];
      const int pc = max(p, 0);
      unsigned char* st = d.st + (size_t)pc * d.n;
      int sv[4];
      double xu[4], lbv[4], ubv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sv[j] = BEYOND ? 0 : st[colj[j]];
        xu[j] = XUNC ? d.xunc[(size_t)pc * d.np + colj[j]] : 0.0;
      }
      if (pair) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          lbv[j] = lbv[j + 2] = d.lb[(size_t)pc * d.nu + kj[j]];
          ubv[j] = ubv[j + 2] = d.ub[(size_t)pc * d.nu + kj[j]];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          lbv[j] = d.lb[(size_t)pc * d.nu + kj[j]];
          ubv[j] = d.ub[(size_t)pc * d.nu + kj[j]];
        }
      }
      int viol = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double xf = XUNC ? xu[j] - acc[i][j][r] : acc[i][j][r];
        const int ns = sv[j] != 0 ? sv[j] : (xf > ubv[j] + d.bound_tol ? 1 : (xf < lbv[j] - d.bound_tol ? 2 : 0));
        const double x = sv[j] == 0 ? xf : (sv[j] == 1 ? ubv[j] : lbv[j]);
        const bool live = p >= 0 && (colok || n0 + wc * 64 + j * 16 + (lane & 15) < d.n);
        if (live && ns != sv[j]) { st[colj[j]] = (unsigned char)ns; viol = 1; }
        if (live && colj[j] < d.nout) d.u_out[(size_t)pc * d.ldu + colj[j]] = x;     // final if nothing changes
      }
      if (viol) d.wflag[pc] = 1;                                   // (same value from every writer)
    }
}

// columns: c0 + 128 tn ...; 1-D grid of g64_grid(ntm, ntn) workgroups, ntm = row tiles of AsmDev::wrows.
template <int MODE>
static __global__ __launch_bounds__(256, 2) void asm_wide_gemm_k(AsmDev d, int c0, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) double g64_sm[];
  __shared__ long long rowoff[128];
  int tm, tn;
  if (!g64_tile_of(blockIdx.x, ntm, ntn, tm, tn)) return;
  const int m0 = tm * 128, n0 = c0 + tn * 128, tid = threadIdx.x;
  {                                                          // a tile without a settled problem has nothing to do
    const int any = tid < 128 && d.rowprob[m0 + tid] >= 0;
    if (!__syncthreads_or(any)) return;
  }
  f64x4 acc[4][4];
  g64_zero(acc);
  if (MODE == WIDE_FAR) {
    if (d.ff_skip) {
      // first-move call: nothing in these columns is delivered.  |x_j| <= |U_j| |T_p| (Cauchy-Schwarz); ffcu[tn] bounds |U_j| for
      // every column at or beyond this tile, tnorm / tslack hold |T_p| (rounded up) and min_k min(ub_k, -lb_k) of the row's problem
      const int row = m0 + (tid & 127);
      const int need = tid < 128 && !(d.ffcu[tn] * d.tnorm[row] <= d.tslack[row]);   // (a NaN needs the check)
      if (!__syncthreads_or(need)) return;
    }
    // staircase factors: the rows of U of this column tile are zero from column ffk[tn] on (the far field forgets the fast modes
    // first -- qp.py prepare_farfield orders the basis that way; nnmpc_qp_set_farfield finds the zeros): k = ffk[tn], not the rank
    const int kk = d.ffk[tn];
    if (d.ff_skip && tid == 0) atomicAdd(&d.counters[ASM_CNT_FFTILES], kk / G64_KC);
    if (kk > 0) {
      const G64Seg st{d.T + (size_t)m0 * d.ffr, (size_t)d.ffr, d.ffU + (size_t)(n0 - c0) * d.ffr, (size_t)d.ffr, kk / G64_KC};
      const G64Seg s1{d.T, 0, d.ffU, 0, 0};
      g64_tile<false>(acc, st, s1, g64_sm);
    }
    wide_epilogue<false, true>(d, acc, m0, n0);
    return;
  }
  const int K1 = min(d.np, wide_tile_k(d, tm));
  const G64Seg sl{d.lam + (size_t)m0 * d.np, (size_t)d.np, d.H + (size_t)n0 * d.np, (size_t)d.np, K1 / G64_KC};
  if (MODE == WIDE_LAZY) {                                   // x0 rows of the tile's problems (rows without one: row 0, never stored)
    if (tid < 128) rowoff[tid] = (long long)max(d.rowprob[m0 + tid], 0) * (long long)d.ka;
    __syncthreads();
    const G64Seg sx{d.x0, (size_t)d.ka, d.Kunc + (size_t)n0 * d.ka, (size_t)d.ka, d.ka / G64_KC};
    g64_tile<true>(acc, sx, sl, g64_sm, rowoff);             // acc = x0 Kunc' - lam Pinv'
    wide_epilogue<false, true>(d, acc, m0, n0);
  } else {
    const G64Seg s1{d.lam, 0, d.H, 0, 0};
    g64_tile<false>(acc, sl, s1, g64_sm);                    // acc = lam Pinv'
    if (c0 > 0) wide_epilogue<true, true>(d, acc, m0, n0);
    else wide_epilogue<true, false>(d, acc, m0, n0);
  }
}

// T = [x0 | lam] [Vx | Vl]'  for the rows of the pass (rows of a round's LAM that did not settle give rows of T nobody reads).
static __global__ __launch_bounds__(256, 2) void asm_wide_t_k(AsmDev d, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) double g64_sm[];
  __shared__ long long rowoff[128];
  int tm, tn;
  if (!g64_tile_of(blockIdx.x, ntm, ntn, tm, tn)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, m0 = tm * 128, n0 = tn * 128;
  {
    const int any = tid < 128 && d.rowprob[m0 + tid] >= 0;
    if (!__syncthreads_or(any)) return;
  }
  const int K1 = min(d.ffW, wide_tile_k(d, tm));             // (Vl has ffW columns; every active bound lies below ffW)
  if (tid < 128) rowoff[tid] = (long long)max(d.rowprob[m0 + tid], 0) * (long long)d.ka;
  __syncthreads();
  f64x4 acc[4][4];
  g64_zero(acc);
  const G64Seg sx{d.x0, (size_t)d.ka, d.ffVx + (size_t)n0 * d.ka, (size_t)d.ka, d.ka / G64_KC};
  const G64Seg sl{d.lam + (size_t)m0 * d.np, (size_t)d.np, d.ffVl + (size_t)n0 * d.ffW, (size_t)d.ffW, K1 / G64_KC};
  g64_tile<false>(acc, sx, sl, g64_sm, rowoff);              // (the minus sign of the Pinv block is in Vl)
  double* Tt = d.T + (size_t)m0 * d.ffr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * 64 + i * 16 + (lane >> 4) + 4 * r;
#pragma unroll
      for (int j = 0; j < 4; ++j) Tt[(size_t)row * d.ffr + n0 + wc * 64 + j * 16 + (lane & 15)] = acc[i][j][r];
    }
}

// First-move calls: |T_p| and the smallest distance of a bound from zero per row of T (one wave per row).
static __global__ __launch_bounds__(256) void asm_wide_tnorm_k(AsmDev d, int ntm) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= ntm * 128) return;
  const int p = d.rowprob[row];
  const bool ok = p >= 0;
  double s = 0.0, sl = 1e300, z1 = 0.0;
  if (ok) {
    for (int k = lane; k < d.ffr; k += 64) { const double t = d.T[(size_t)row * d.ffr + k]; s += t * t; }
    for (int k = lane; k < d.nu; k += 64) sl = fmin(sl, fmin(d.ub[(size_t)p * d.nu + k], -d.lb[(size_t)p * d.nu + k]));
    // |z|_1 = |x0|_1 + |lam|_1: the factored product is off from M z by at most ff_efar |z|_1 per entry (far_verify_k's bound)
    for (int k = lane; k < d.ka; k += 64) z1 += fabs(d.x0[(size_t)p * d.ka + k]);
    const int m = d.mg[p];
    const int* idx = d.idxg + (size_t)p * d.max_active;
    for (int i = lane; i < m; i += 64) z1 += fabs(d.lam[(size_t)row * d.np + idx[i]]);
  }
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); sl = fmin(sl, __shfl_xor(sl, off)); z1 += __shfl_xor(z1, off); }
  if (lane == 0) {
    d.tnorm[row] = ok ? sqrt(s) * (1.0 + 1e-12) : 0.0;       // rows without a problem never ask for a tile
    // |x_j| <= |U_j| |T_p| + ff_efar |z|_1 must stay inside the box shrunk by the feasibility tolerance
    d.tslack[row] = ok ? sl - d.bound_tol - d.ff_efar * z1 * (1.0 + 1e-12) : 1e300;   // (a NaN bound or T entry fails "<=": the tile is evaluated)
  }
}

// max |U V' - M| over the far block: row j of the block (column W + j of the problem), all n_aug + W columns of M.
static __global__ __launch_bounds__(256) void far_verify_k(const double* __restrict__ U, const double* __restrict__ Vx, const double* __restrict__ Vl,
                                                           const double* __restrict__ Kunc, const double* __restrict__ H, int W, int rp, int ka, int np,
                                                           unsigned long long* __restrict__ emax) {
  extern __shared__ double urow[];
  const int j = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < rp; i += 256) urow[i] = U[(size_t)j * rp + i];
  __syncthreads();
  double e = 0.0;
  for (int k = tid; k < ka + W; k += 256) {
    double s = 0.0;
    if (k < ka) { for (int i = 0; i < rp; ++i) s += urow[i] * Vx[(size_t)i * ka + k]; s -= Kunc[(size_t)(W + j) * ka + k]; }
    else { for (int i = 0; i < rp; ++i) s += urow[i] * Vl[(size_t)i * W + (k - ka)]; s += H[(size_t)(W + j) * np + (k - ka)]; }
    e = fmax(e, fabs(s));
    if (!(s == s)) e = 1e300;
  }
  for (int off = 32; off > 0; off >>= 1) e = fmax(e, __shfl_xor(e, off));
  if ((tid & 63) == 0) atomicMax(emax, (unsigned long long)__double_as_longlong(e));   // (non-negative doubles order like integers)
}

}  // namespace nnmpc
