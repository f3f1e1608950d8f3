// Small problems: the whole primal-dual active-set iteration of a problem in ONE wave (gfx950).
//
// The lock-step rounds of qp_asm.h are built for the CDU size (n = 4480 variables, ~110 active bounds): a round is eight
// launches and a host read-back, amortised over 100 000 problems and milliseconds of MFMA work.  At the CSTRs size (n = 540, ~17
// active bounds, SURVEY 8d configs[0]) a round is ~0.25 ms of launch and read-back latency around microseconds of work, and a
// 10 000-problem step was 12 such rounds plus a device tail: 4.6 ms.  Here one wave owns one problem from the first set to the
// certificate -- no host round trip, no row of LAM / XH, no lists:
//     ordered list of the active bounds (wave ballots over the bound states, which live in LDS)
//     lam = (H_AA)^-1 (x_unc,A - b_A)        asm_reg_core: the register-tile Cholesky of the round kernels, fp64
//     x = x_unc - lam H[A, :]                lanes = columns, rows of Pinv straight from L2 (n^2 * 8 bytes = 2.3 MB at n = 540);
//                                            inside the column window (past the last active bound) while the set still moves,
//                                            all columns once when it settles
//     fp64 feasibility / multiplier-sign tests, the exchange rule of asm_update_k (all infeasible indices change sides while their
//     number keeps falling, ASM_SM_GRACE iterations of grace, then Murty's least-index single exchanges), the certificate of
//     asm_update_k (inverse-error bound, else ASM_DONE: asm_certify_k checks with P itself).
// Two instances, launched one after the other over all problems of the segment (a wave whose problem is not running exits at once):
//     asm_small_k<2, 3, 4>   sets of up to 32 bounds -- the common case (CSTRs batch: 17 on average) -- in 146 VGPRs: three waves per
//                            SIMD hide the L2 round trips of the gathers (the iteration is a chain of them)
//     asm_small_k<7, 1, 8>   sets of up to 112 bounds: the problems the first instance handed back (one in twenty of that batch); a
//                            four-block solve up to 64 bounds, a seven-block one (first five block columns' tiles in LDS, identity
//                            padding for any size) beyond
//     asm_small_k<9, 1, 8>   sets of up to 144 bounds: what outgrew the second (rare -- one problem per batch or none --, but the lock-step
//                            rounds and the device tail behind them cost that batch a millisecond)
// A problem whose set outgrows an instance, or that is still moving after `budget` iterations, is handed back as it stands
// (bound states, exchange-rule memory, iteration count): it stays ASM_RUN and the next instance, then the lock-step rounds / the
// device tail carry on.  Same arithmetic (fp64 everywhere), same tests, same certificate: the answers are the ones the rounds give.
// One wave per workgroup: a problem that needs 26 iterations does not hold the LDS of three finished neighbours.
// What did not pay (CSTRs-size 10 000 problems, ms per step): a third instance for 33 .. 64 bounds in between (0.32 + 0.52 + 0.69
// against 0.32 + 0.70: the slowest problem sets the time of each launch and passes through all of them); starting the problems
// whose FIRST set has more than 32 bounds in the large instance at once, on a second stream beside the small one (1.8 against
// 1.4: the slowest problem starts small); seven solve instances in one kernel (60 .. 230 registers spilled to scratch).
#pragma once
#include "qp_asm.h"

namespace nnmpc {

constexpr int ASM_SM_GRACE = 24;                 // iterations without a new minimum of infeasible indices before single exchanges (see solve_segment_asm)
constexpr int ASM_SM_NMAX = 1024;                // largest n (x_unc, bound states and decisions of the problem live in LDS)
// Tiles of the first block columns kept in LDS instead of accumulator registers (asm_reg_core's NL), by set size in 16-blocks.
// No instance of this kernel uses scratch memory: spilled registers would sit in the dependent chain of an iteration.
__host__ __device__ constexpr int asm_small_nl(int mb) { return mb <= 3 ? 0 : (mb <= 5 ? 1 : (mb == 6 ? 2 : 5)); }
__host__ __device__ constexpr int asm_small_nlt(int mb) { return asm_small_nl(mb) * (mb - 1) - asm_small_nl(mb) * (asm_small_nl(mb) - 1) / 2; }
// LDS bytes of a wave: dt, Yt | ys, rv [16 MB] | lamv [16 MB] doubles | LDS-resident tiles | al, ao [16 MB] ints | x_unc [nst] doubles |
// lb, ub [nu] | st, dec [nst] bytes
__host__ __device__ constexpr int asm_small_lds_bytes(int mbmax, int n, int nu) {
  return 2 * ASM_TS * 8 + 3 * 16 * mbmax * 8 + asm_small_nlt(mbmax) * 256 * 8 + 2 * 16 * mbmax * 4 + (((n + 63) / 64) * 64) * 8 + 2 * (((nu + 1) / 2) * 2) * 8 +
         2 * (((n + 63) / 64) * 64);
}

#ifdef ASM_SM_PROF
__device__ unsigned long long asm_small_prof[2][8];          // diagnostics build only: shader-clock sums per phase and instance
#define ASM_SP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&asm_small_prof[MBMAX > 2][i], t_ - tp_); tp_ = t_; } while (0)
#else
#define ASM_SP(i) do { } while (0)
#endif
template <int MBMAX, int OCC, int XR>
__global__ __launch_bounds__(64, OCC) void asm_small_k(AsmDev d, int budget, int grace0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
  constexpr int MS = 16 * MBMAX;                             // largest set of this instance
  const int lane = threadIdx.x;
  const int li = lane & 15, lq = lane >> 4;
  const int nst = ((d.n + 63) / 64) * 64, nup = ((d.nu + 1) / 2) * 2;
  double* dt = reinterpret_cast<double*>(sm_raw);
  double* Yt = dt + ASM_TS;
  double* ys = Yt + ASM_TS;                                  // [MS] forward result; the index list of the gather before that
  double* rv = ys + MS;                                      // [MS] right-hand side
  double* lamv = rv + MS;                                    // [MS] multipliers in list order
  double* ltl = lamv + MS;                                   // LDS-resident tiles of the larger sets (sized for MBMAX; smaller sets use a prefix)
  double* xul = ltl + asm_small_nlt(MBMAX) * 256;            // [nst] x_unc of the problem
  double* lbl = xul + nst;                                   // [nu] its bounds
  double* ubl = lbl + nup;
  int* al = reinterpret_cast<int*>(ubl + nup);               // [MS] active indices, ascending
  unsigned* ao = reinterpret_cast<unsigned*>(al + MS);       // [MS] element offset of their rows of Pinv
  unsigned char* stl = reinterpret_cast<unsigned char*>(ao + MS);   // [nst] bound states (0 free, 1 upper, 2 lower)
  unsigned char* dec = stl + nst;                            // [nst] this iteration's decisions (255: stays)
  const int p = blockIdx.x;
  if (p >= d.nseg || d.state[p] != ASM_RUN) return;
  const size_t o = (size_t)p * d.np;
  unsigned char* stg = d.st + (size_t)p * d.n;
  for (int r = lane; r < nst; r += 64) {
    stl[r] = r < d.n ? stg[r] : (unsigned char)0;
    xul[r] = r < d.n ? d.xunc[o + r] : 0.0;
  }
  for (int k = lane; k < d.nu; k += 64) { lbl[k] = d.lb[(size_t)p * d.nu + k]; ubl[k] = d.ub[(size_t)p * d.nu + k]; }
  int best = d.ninf_best[p], grace = d.alpha[p], rounds = d.rounds[p];
  double wflops = 0.0, wbytes_g = 0.0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int m = 0, outcome = 0;                                    // outcome: 0 hand back (still running), 1 settled, 2 not positive definite
  bool sure = false;
  ASM_FENCE();
#ifdef ASM_SM_PROF
  unsigned long long tp_ = __builtin_amdgcn_s_memtime();
#endif
  for (int it = 0; it < budget; ++it) {
    ASM_SP(5);
    // ---- ordered list of the active bounds
    m = 0;
    for (int c = 0; c < nst; c += 64) {
      const int s = stl[c + lane];
      const unsigned long long mask = __ballot(s != 0);
      const int k = m + __popcll(mask & lt_mask);
      if (s != 0 && k < MS) { al[k] = c + lane; ao[k] = (unsigned)(c + lane) * (unsigned)d.np; }
      m += __popcll(mask);
    }
    if (m > MS) break;                                       // beyond the register tiles of this instance: the next one takes over
    ASM_FENCE();
    const int lastact = m > 0 ? al[m - 1] : -1;
    ASM_SP(0);
    // ---- multipliers
    if (m > 0) {
      int* ix = reinterpret_cast<int*>(ys);
      for (int i = lane; i < MS; i += 64) {
        const int a = al[min(i, m - 1)], k = a % d.nu;
        ix[i] = a;
        rv[i] = i < m ? xul[a] - (stl[a] == 1 ? ubl[k] : lbl[k]) : 0.0;
      }
      ASM_FENCE();
      int bad = 0;
      const int mb = (m + 15) >> 4;
      // (two instances of the solve per kernel: more of them in one kernel and the register allocator no longer fits the largest
      // -- 60 .. 230 registers spilled to scratch with seven; up to four blocks every tile has the padding select, so a smaller
      // set runs the larger instance with identity padding; the seven-block instance does the same with ANYM)
      auto solve = [&](auto MBc, auto ANYc) {
        constexpr int MB = decltype(MBc)::value;
        double lam[MB];
        bad = asm_reg_core<double, MB, asm_small_nl(MB), (decltype(ANYc)::value != 0)>(d, m, ix, dt, Yt, ys, rv, ltl + lane, (const double*)nullptr, lam, lane, 0, 0);
        if (!bad) {
#pragma unroll
          for (int I = 0; I < MB; ++I) { const int i = 16 * I + li; if (lq == 0 && i < m) lamv[i] = lam[I]; }
        }
      };
      if constexpr (MBMAX <= 2) { if (mb == 1) solve(asm_ic<1>{}, asm_ic<0>{}); else solve(asm_ic<2>{}, asm_ic<0>{}); }
      else if constexpr (MBMAX <= 4) solve(asm_ic<4>{}, asm_ic<0>{});
      else { if (mb <= 4) solve(asm_ic<4>{}, asm_ic<0>{}); else solve(asm_ic<MBMAX>{}, asm_ic<1>{}); }
      if (bad) { outcome = 2; break; }
      ASM_FENCE();
      const double md = (double)m;
      wflops += md * md * md / 3.0 + 2.0 * md * md;
      wbytes_g += 8.0 * (md * (md + 1.0) / 2.0 + 2.0 * md);
    }
    ASM_SP(1);
    // ---- x over the columns [c0, c1), tests of the free variables there, u written out (final once nothing changes), decisions
    // recorded.  A lane owns FOUR consecutive columns of a 256-column chunk: one 32-byte load per row of Pinv and lane (2 KB per
    // wave and instruction), XR rows in flight -- the loop is a chain of L2 round trips, and at 8 bytes per lane and load (the
    // first version) a set of 90 bounds needed 450 of them per lane and iteration.
    int ninf = 0, rmin = 0x7fffffff;
    auto xpass = [&](int c0, int c1) {
      for (int c = c0; c < c1; c += 256) {
        const int v0 = c + 4 * lane;
        const bool inr = v0 < nst;                           // (nst is a multiple of 64: all four columns or none)
        const int vq = min(v0, d.np - 4);                    // load address inside the row of Pinv (np is a multiple of 4)
        const int vl = min(v0, nst - 4);                     // ... and inside the LDS arrays
        f64x4_t acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        int i = 0;
        for (; i + XR <= m; i += XR) {
          f64x4_t h[XR];
#pragma unroll
          for (int u = 0; u < XR; ++u) h[u] = *reinterpret_cast<const f64x4_t*>(d.H + ao[i + u] + vq);
#pragma unroll
          for (int u = 0; u < XR; u += 2) { acc += h[u] * lamv[i + u]; acc2 += h[u + 1] * lamv[i + u + 1]; }
        }
        for (; i < m; ++i) acc += *reinterpret_cast<const f64x4_t*>(d.H + ao[i] + vq) * lamv[i];
        acc += acc2;
        const uint32_t sw = *reinterpret_cast<const uint32_t*>(stl + vl);
        const f64x4_t xu4 = *reinterpret_cast<const f64x4_t*>(xul + vl);
        uint32_t dw = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = v0 + j;
          const bool in = inr && r < d.n;
          const int sv = (sw >> (8 * j)) & 0xff;             // (padding entries are 0 and never leave it)
          const int k = min(r, d.n - 1) % d.nu;
          const double lb = lbl[k], ub = ubl[k];
          const double xf = xu4[j] - acc[j];
          int dc = 255;
          if (in && sv == 0) { if (xf > ub + d.bound_tol) dc = 1; else if (xf < lb - d.bound_tol) dc = 2; }
          const double x = sv == 0 ? xf : (sv == 1 ? ub : lb);
          if (in) {
            if (r < d.nout) d.u_out[(size_t)p * d.ldu + r] = x;
            d.x[o + r] = x;                                  // ... and as a GEMM row, should P itself have to confirm it
          }
          dw |= (uint32_t)dc << (8 * j);
          const unsigned long long mask = __ballot(dc != 255);
          if (mask) { ninf += __popcll(mask); rmin = min(rmin, c + 4 * (int)__builtin_ctzll(mask) + j); }
        }
        if (inr) *reinterpret_cast<uint32_t*>(dec + v0) = dw;
      }
    };
    const int xlim = min(nst, ((lastact + 1 + d.nu + 255) / 256) * 256);   // (an empty set: the first 256 columns)
    xpass(0, xlim);
    int xdone = xlim;
    ASM_FENCE();
    ASM_SP(2);
    // ---- multiplier signs of the active bounds (keep iff the multiplier has the sign of its side)
    for (int i0 = 0; i0 < m; i0 += 64) {
      const int i = i0 + lane;
      bool wrong = false;
      if (i < m) {
        const int a = al[i], sa = stl[a];
        const double l = lamv[i];
        wrong = (sa == 1 && l <= 0.0) || (sa == 2 && l >= 0.0);
        if (wrong) dec[a] = 0;
      }
      const unsigned long long mask = __ballot(wrong);
      if (mask) { ninf += __popcll(mask); rmin = min(rmin, al[i0 + (int)__builtin_ctzll(mask)]); }   // (the list ascends)
    }
    ++rounds;
    ASM_SP(3);
#ifdef ASM_SM_PROF
    if (lane == 0) atomicAdd(&asm_small_prof[MBMAX > 2][6], 1ull);
#endif
    if (ninf == 0 && xlim < nst) {                           // settled inside the window: the columns beyond it, once
      xpass(xlim, nst);
      xdone = nst;
      if (ninf > 0) { best = 0x7fffffff; grace = grace0; }   // bounds out there join: the exchange rule starts afresh (as asm_wide_k)
    }
    if (ninf == 0) {
      // ---- settled: the certificate of asm_update_k
      double l1 = 0.0, lmin = 1e300, x1 = 0.0;
      for (int i = lane; i < m; i += 64) { const double l = fabs(lamv[i]); l1 += l; lmin = fmin(lmin, l); }
      for (int k = lane; k < d.ka; k += 64) x1 += fabs(d.x0[(size_t)p * d.ka + k]);
      for (int off = 32; off > 0; off >>= 1) {
        l1 += __shfl_xor(l1, off); x1 += __shfl_xor(x1, off);
        lmin = fmin(lmin, __shfl_xor(lmin, off));
      }
      const double QI = d.tqmax * x1;
      const double bnd = 2.0 * (d.e1max * x1 + d.e2max * l1) + 1e-14 * (QI + l1);
      sure = bnd <= d.stat_tol * d.pscale && lmin > bnd;
      outcome = 1;
      break;
    }
    int single = 0;
    if (ninf < best) { best = ninf; grace = grace0; }
    else if (grace > 0) --grace;
    else single = 1;
    ASM_FENCE();
    for (int r = lane; r < xdone; r += 64) {
      const int dc = dec[r];
      if (dc != 255 && (!single || r == rmin)) stl[r] = (unsigned char)dc;
    }
    ASM_FENCE();
  }
  // ---- out: bound states, iteration count, exchange-rule memory (a problem handed back carries on from here)
  for (int r = lane; r < d.n; r += 64) stg[r] = stl[r];
  if (lane == 0) {
    d.rounds[p] = rounds;
    d.mg[p] = m;
    d.prec[p] = 1;
    d.hi[p] = d.n;
    d.ninf_best[p] = best;
    d.alpha[p] = (unsigned char)min(grace, 255);
    if (d.work) { d.work[3 * p] += wflops; d.work[3 * p + 1] += wbytes_g; }
    if (outcome == 1) {
      d.state[p] = sure ? ASM_CERT : ASM_DONE;
      if (!sure) atomicAdd(&d.counters[ASM_CNT_DONE], 1);
    } else if (outcome == 2) d.state[p] = ASM_FALLBACK;
    else if (m > d.max_active) d.state[p] = ASM_FALLBACK;
  }
}

}  // namespace nnmpc
