// Full-width pass of the active-set path with the check in the GEMM's epilogue (gfx950).
//
// A problem that settles inside the column window gets ONE pass over the columns beyond it: x = x_unc - lamw * Pinv there,
// feasibility of the free variables, u written out.  Separate kernels made that  GEMM -> XHW (8 B per element written),
// asm_wide_k (XHW and x_unc read back, u written: 24 B)  -- HBM-bound bookkeeping next to an MFMA-bound GEMM.  Here the
// 128 x 128 fp64 tile of gemm_nt_f64_128_k keeps its product in the accumulators and its epilogue does the check: x_unc and the
// bound states come in (9 B per element), u goes out (8 B), a violated bound changes state in place and raises the
// problem's flag; asm_wide_k is left with the columns inside the window, the multiplier statistics and the decision.
#pragma once
#include "gemm_kernels.h"
#include "qp_asm.h"

namespace nnmpc {

// rows: the problems of k-group g awaiting the check (row w of region g of LAMW -> problem wlist[g * wcap + w]);
// columns: c0 + 128 blockIdx.x ...; k-range: the group's last possible active bound (counters[ASM_CNT_WKMAX + g]).
static __global__ __launch_bounds__(256, 2) void asm_wide_gemm_k(AsmDev d, int g, int c0) {
  constexpr int LD = 18, TS = 128 * LD;
  extern __shared__ __attribute__((aligned(16))) double sm128[];   // [2][A 128 x LD | B 128 x LD]
  const int cntg = d.counters[ASM_CNT_WIDEG + g];
  const int m0 = blockIdx.y * 128, n0 = c0 + blockIdx.x * 128;
  if (m0 >= cntg) return;
  const int K = min(d.np, ((d.counters[ASM_CNT_WKMAX + g] + 16) / 16) * 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const size_t lda = d.np, ldb = d.np;
  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  const int lrow0 = tid >> 3, lc = (tid & 7) * 2;           // rows lrow0 + 32 h
  const double* Ag = d.lamw + ((size_t)g * d.wcap + m0) * lda;
  const double* Bg = d.H + (size_t)n0 * ldb;
  f64x2 ra[4], rb[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    ra[h] = *reinterpret_cast<const f64x2*>(Ag + (size_t)(lrow0 + 32 * h) * lda + lc);
    rb[h] = *reinterpret_cast<const f64x2*>(Bg + (size_t)(lrow0 + 32 * h) * ldb + lc);
  }
  const int li = lane & 15, kq = lane >> 4;
  const int nk = K / 16;
  for (int kc = 0; kc < nk; ++kc) {
    double* sA = sm128 + (kc & 1) * 2 * TS;
    double* sB = sA + TS;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      *reinterpret_cast<f64x2*>(sA + (lrow0 + 32 * h) * LD + lc) = ra[h];
      *reinterpret_cast<f64x2*>(sB + (lrow0 + 32 * h) * LD + lc) = rb[h];
    }
    __syncthreads();
    if (kc + 1 < nk) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        ra[h] = *reinterpret_cast<const f64x2*>(Ag + (size_t)(lrow0 + 32 * h) * lda + (kc + 1) * 16 + lc);
        rb[h] = *reinterpret_cast<const f64x2*>(Bg + (size_t)(lrow0 + 32 * h) * ldb + (kc + 1) * 16 + lc);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      double a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = sA[(wr * 64 + t * 16 + li) * LD + 4 * s + kq];
        b[t] = sB[(wc * 64 + t * 16 + li) * LD + 4 * s + kq];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // ---- epilogue: x = x_unc - (lamw Pinv), feasibility of the free variables, u out.  All loads of a row are issued
  // unconditionally and together (a branch per element would turn the epilogue into a chain of dependent round trips).
  const int* wl = d.wlist + (size_t)g * d.wcap;
  int colj[4], kj[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    colj[j] = min(n0 + wc * 64 + j * 16 + (lane & 15), d.n - 1);   // (clamped: columns >= n are padding, never stored)
    kj[j] = colj[j] % d.nu;
  }
  const bool colok = n0 + wc * 64 + 63 < d.n;                      // whole 64-column half inside the problem (wave-uniform)
  int prow[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wr * 64 + i * 16 + (lane >> 4) + 4 * r;
      prow[4 * i + r] = row < cntg ? wl[row] : -1;
    }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = prow[4 * i + r];
      const int pc = max(p, 0);
      const size_t o = (size_t)pc * d.np;
      unsigned char* st = d.st + (size_t)pc * d.n;
      int sv[4];
      double xu[4], lbv[4], ubv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sv[j] = st[colj[j]];
        xu[j] = d.xunc[o + colj[j]];
        lbv[j] = d.lb[(size_t)pc * d.nu + kj[j]];
        ubv[j] = d.ub[(size_t)pc * d.nu + kj[j]];
      }
      int viol = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double xf = xu[j] - acc[i][j][r];
        const int ns = sv[j] != 0 ? sv[j] : (xf > ubv[j] + d.bound_tol ? 1 : (xf < lbv[j] - d.bound_tol ? 2 : 0));
        const double x = sv[j] == 0 ? xf : (sv[j] == 1 ? ubv[j] : lbv[j]);
        const bool live = p >= 0 && (colok || n0 + wc * 64 + j * 16 + (lane & 15) < d.n);
        if (live && ns != sv[j]) { st[colj[j]] = (unsigned char)ns; viol = 1; }
        if (live && colj[j] < d.nout) d.u_out[(size_t)pc * d.ldu + colj[j]] = x;     // final if nothing changes
      }
      if (viol) d.wflag[pc] = 1;                                   // (same value from every writer)
    }
}

}  // namespace nnmpc
