// Shared-inverse primal-dual active-set path of the batched box QP (gfx950).
//
// Every sample of the offline data generation has the SAME Hessian P (reference
// lib/linearMPC.py:472, :503-504: only q = tq x0 and the bounds change), so with
// H = P^-1 (fp64, one-time host setup) the equality-constrained problem on an active set A
//      min 1/2 x'Px + q'x   s.t.  x_A = b_A
// has the closed form
//      lam = (H_AA)^-1 (x_unc,A - b_A),     x = x_unc - H[:,A] lam,     x_unc = -H q = Kunc x0,
// (lam = multipliers: > 0 at an upper bound, < 0 at a lower bound when the set is right).
// A primal-dual active-set iteration (add violated bounds, drop wrong-sign multipliers)
// therefore costs  |A|^3/3 + O(|A|^2)  per problem (dense Cholesky of the |A| x |A| block of H, held in
// the MFMA accumulators of one wave) plus one row of the batched fp64 GEMM  LAM * H  -- no n^3 work at all.
// Problems whose set is too large, or that do not settle, fall back to the PDIP path.
//
// Per round (all unfinished problems of a segment in lock-step; the host reads 24 counters once):
//   asm_wide_t_k / asm_wide_gemm_k / asm_wide_k   full-width pass of the problems that settled inside last round's column
//                         window (qp_wide.h: far-field form, check in the GEMM's epilogue)
//   asm_count_k           ordered list of the active indices of every running problem
//   asm_bins_a/b_k        scans: row of LAM / XH for this round (running problems packed), size-class lists
//   asm_lambda_reg*_k     gather S = H_AA, Cholesky, lam -> row of LAM   (f32 until the set settles, then fp64)
//   gemm_nt_f64_t128_k    XH = LAM * H inside the column window          (MFMA f64, gemm64.h)
//   asm_update_k          x = x_unc - XH (free), x = bound (active); fp64 feasibility / multiplier-sign tests ->
//                         next set (exchange rule with anti-cycling fallback), or settled -> full-width pass
// and once per segment asm_init_k (first sets) and asm_certify_k (active-set bits, status, check with P itself
// for the rows the inverse-error bound cannot certify).  DESIGN.md section 2a has the details.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// Translation units.  The kernels of this header are compiled into qp_solver.o; the workgroup kernels of qp_wg.h -- minutes of
// compile time each -- into one object per kernel (qp_wg_kernels.hip with -DASM_WG_TU=0..4, which skips the kernels here), and
// qp_solver.hip (ASM_WG_TU = -1) only declares those.  Without ASM_WG_TU (scripts/micro) everything is defined in the including file.
#if defined(ASM_WG_TU) && ASM_WG_TU >= 0
#define ASM_OWN_KERNELS 0
#else
#define ASM_OWN_KERNELS 1
#endif

namespace nnmpc {

constexpr int ASM_MLDS = 176;      // largest active set factored in LDS (11 x 11 lower 16x16 fp64 tiles)
constexpr int ASM_TS = 16 * 17;    // doubles per LDS tile (16 rows, stride 17: conflict-free MFMA operand reads)
enum { ASM_RUN = 0, ASM_DONE = 1, ASM_FALLBACK = 2, ASM_CERT = 3, ASM_WIDE = 4, ASM_INVALID = 5 };   // CERT: finished and certified by the
                                   // inverse-error bound; INVALID: NaN / Inf in x0 or lb > ub (status NNMPC_ST_NUMERIC, u = NaN)
constexpr int ASM_NBIN = 8;        // size classes by the number of 16-blocks: class b holds sets of 16 (b + 4) or fewer
constexpr int ASM_NREG = 6;        // classes 0..5 (<= 144 bounds): four problems per workgroup (asm_lambda_reg_k); 6, 7: two
constexpr int ASM_NCNT = 40;       // ints in AsmDev::counters
constexpr int ASM_CNT_F32 = 16;    // counters[16 + b]: length of the f32 list of size class b
constexpr int ASM_NLIST = 2 * ASM_NBIN;   // lists in AsmDev::binlist: fp64 classes, then f32 classes, then (list ASM_NLIST) the
                                          // f32 rounds of sets of ASM_MLDS + 1 .. ASM_BIG32 bounds (asm_lambda_tile32_k)
constexpr int ASM_BIG32 = 256;            // largest set whose f32 rounds run with the tiles in LDS (16 x 16 lower f32 tiles, one
                                          // workgroup per problem); beyond: fp64 from the start, tiles in an L2 slab
constexpr int ASM_CNT_BIG64 = 37;         // counters[37]: length of list ASM_NLIST + 1 (fp64 rounds of sets of 177 .. 256 bounds: asm_lambda_wg64_k)
constexpr int ASM_CNT_BIG32 = 36;         // counters[36]: length of list ASM_NLIST
constexpr int ASM_BIG32B = 384;           // largest set whose f32 rounds run in registers at all: 257 .. 384 bounds on EIGHT waves per problem
                                          // (asm_lambda_wg32b_k, qp_wg.h); beyond, and for every fp64 solve beyond 256: the L2-slab kernel
constexpr int ASM_CNT_BIG64R = 31;        // counters[31]: length of list ASM_NLIST + 3 (fp64 solves of sets of 257 .. 384 bounds: f32 factor + fp64 refinement)
constexpr int ASM_CNT_BIG32B = 39;        // counters[39]: length of list ASM_NLIST + 2 (f32 rounds of sets of 257 .. 384 bounds)
constexpr double ASM_REFINE_TOL = 1e-12;   // residual of a corrected f32 solve, relative to max |b|, that counts as an fp64 solve (asm_update_k)
constexpr int ASM_GRACE = 10;       // rounds without a new minimum of infeasible indices before single exchanges take over
constexpr int ASM_CNT_WIDE = 12;   // counters[13]: problems handled by the last asm_wide_k, counters[ASM_CNT_WKSUM]: the sum of their
constexpr int ASM_CNT_WKSUM = 30;  // (last active index + 1) -- statistics, from the scans of asm_bins (no atomics)
constexpr int ASM_TAIL_MB = 8;     // asm_tail_k keeps the tiles of sets of up to 128 bounds in LDS
constexpr int ASM_CNT_TAIL = 35;   // counters[35]: problems handed to asm_tail_k
constexpr int ASM_CNT_FFTILES = 38; // counters[38]: 128 x 128 tiles the far-field pass evaluated in first-move calls (statistics; cumulative over a segment)
constexpr int ASM_CNT_DONE = 14;   // counters[14]: problems finished but not certified by the inverse-error bound
// list `list` of AsmDev::binlist and its length: 0..ASM_NBIN-1 the fp64 size classes (counters[4 + b]),
// ASM_NBIN + b the f32 ones (counters[ASM_CNT_F32 + b])
__host__ __device__ constexpr int asm_list_counter(int list) {
  return list < ASM_NBIN ? 4 + list : (list < ASM_NLIST ? ASM_CNT_F32 + list - ASM_NBIN : (list == ASM_NLIST ? ASM_CNT_BIG32 : (list == ASM_NLIST + 1 ? ASM_CNT_BIG64 : (list == ASM_NLIST + 2 ? ASM_CNT_BIG32B : ASM_CNT_BIG64R))));
}
__host__ __device__ constexpr int asm_list_of_counter(int c) { return c >= ASM_CNT_F32 ? ASM_NBIN + c - ASM_CNT_F32 : c - 4; }   // size-class lists only
constexpr int ASM_NKG = 3;         // rows of a round are ordered by the last active stage (groups: <= median, +1, beyond),
                                   // so that a 128-row block of the GEMM stops its k-loop at ITS last active bound
constexpr int ASM_NSCAN = ASM_NLIST + 9 + 2 * ASM_NKG;  // scan columns: large sets, the lists, (fp64, f32) x group rows, the problems the
                                                        // full-width pass handled (number, sum of last active index + 1), sum and max of
                                                        // the last active indices of the running problems
constexpr int ASM_CNT_ROWS32 = 15; // counters[15]: rows of LAM32 / XH32 handed out ([2]: rows of LAM / XH)
__host__ __device__ constexpr int asm_bin_cap(int b) { return 16 * (b + 4); }

struct AsmDev {
  int n, np, nu, nseg;
  int max_active;                  // larger sets -> fallback
  int max_rounds;
  double bound_tol, stat_tol, pscale;
  double e1max, e2max;             // max |P Kunc + tq|, max |P Pinv - I| (verified once at setup)
  const double* x0;                // [nseg][ka] padded initial states (for the certificate; first K segment of the full-width pass)
  int ka;
  const double* Kunc;              // [np][ka]: x_unc = Kunc x0
  int pred_w;                      // bound states of the columns [0, pred_w) were named by asm_predict_k (qp_predict.h): asm_init_k leaves them
  int pred_f64;                    // ... and the rounds start in fp64 (a predicted set is close to the final one: an f32 round would mostly confirm it)
  int winit;                       // columns the first sets are drawn from (without a guess): the leading eighth of the horizon, 512 at least
  int Wx;                          // x_unc exists in HBM for the columns [0, Wx) only (Wx = np: all of them); beyond, the full-width
                                   // pass forms it inside its GEMM (qp_wide.h), the rare consumers below from Kunc and x0
  // far-field form of the full-width pass (qp_wide.h; nnmpc_qp_set_farfield): x[ffW:] = U (Vx x0 + Vl lam[0:ffW])
  const double* ffU;               // [np - ffW][ffr]
  const double* ffVx;              // [ffr][ka]
  const double* ffVl;              // [ffr][ffW]  (minus sign of the Pinv block included)
  const double* ffcu;              // [(np - ffW) / 128] bound on |U_j| over all columns at or beyond each 128-column tile
  const int* ffk;                  // [(np - ffW) / 128] columns of U the tile's rows use (staircase: zero from there on), multiple of 16
  int ffr, ffW;                    // padded rank (multiple of 128), the window the factors belong to
  double* T;                       // [row tiles * 128][ffr]
  double *tnorm, *tslack;          // [row tiles * 128] |T_p| and min_k min(ub_k, -lb_k) (first-move calls)
  int ff_skip;                     // first-move call: column tiles certified feasible by |U_j| |T_p| are not evaluated
  double ff_err;                   // |P|_inf * max |U V' - M| of the factors the last full-width pass used (0: dense form): enters the certificate
  double ff_efar;                  // max |U V' - M| itself: x beyond the window is off by at most ff_efar (|x0|_1 + |lam|_1) -- enters the skip test of first-move calls
  const double* H;                 // [np][np] fp64 inverse Hessian
  const float* H32;                // the same rounded to f32 (operand of the f32 GEMM; the f32 rounds gather S from it: same values as
                                   // rounding the fp64 entries on the fly, half the bytes)
  const double* lb;                // [nseg][nu]
  const double* ub;
  const double* xunc;              // [nseg][np]
  const double* q64;               // [nseg][np]
  double* x;                       // [nseg][np]
  double* lam;                     // [nseg][np] dense multiplier rows (GEMM operand)
  const double* xh;                // [nseg][np] = lam * H
  const double* px;                // [nseg][np] = x * P (certification)
  unsigned char* st;               // [nseg][n]
  const unsigned char* guess;      // [nseg][n] caller's active-set estimate or NULL
  int* state;                      // [nseg] ASM_RUN / DONE / FALLBACK
  int* rounds;                     // [nseg]
  int* counters;                   // [0] still running, [1] big-set list length, [2] rows handed out, [3] largest active variable
                                   // index of this round (GEMM k-range), [4 + b] length of size-bin list b
  int* biglist;                    // [nseg] problems whose set does not fit LDS
  int* binlist;                    // [ASM_NBIN][nseg] problems by active-set size (LDS size / occupancy classes)
  int* idxg;                       // [nseg][max_active] ordered active indices (asm_count_k)
  int* mg;                         // [nseg] their number
  int* row;                        // [nseg] row of lam / xh this round: the running problems are packed into rows
                                   // 0..nrun-1 (counters[2]), so the GEMM only covers those
  double tqmax;                    // max |tq| entry: |q|_inf <= tqmax |x0|_1 (q itself is only formed for the full check)
  unsigned char* rowk;             // [nseg] 1: this round's row is one of LAM32 / XH32 (f32 solve and GEMM), 0: of LAM / XH
  float* lam32;                    // [rows] f32 multiplier rows (GEMM operand of the f32 rounds)
  const float* xh32;               // [rows] = lam32 * H (f32)
  unsigned char* prec;             // [nseg] 0: rounds in f32 until the set settles, 1: fp64 (only these results are accepted)
  int* kblk;                       // [2][nkblk / 2] last active bound per 64-row block of LAM / LAM32 (k-range of the GEMMs)
  int nkblk;
  int kref;                        // last round's overall last active bound (row grouping of asm_bins)
  int* hi;                         // [nseg] no bound at or beyond this index is active (asm_count_k scans [0, hi) only)
  int* ninf_best;                  // [nseg] smallest number of infeasible indices seen so far (exchange rule of asm_update_k)
  unsigned char* alpha;            // [nseg] rounds of grace left before single exchanges
  unsigned char* redo;             // [nseg] set by the f32 kernel when S is not positive definite in f32: the round is void
  int* lrank;                      // [nseg] asm_bins scratch: rank inside its chunk and list
  int* ctot;                       // [chunks][ASM_NSCAN] asm_bins scratch: per-chunk totals, last column: max active index
  int use_f32;                     // run the rounds in f32 until the set settles (then fp64)
  int W;                           // columns evaluated in this round (multiple of 64, past the last active bound of any
                                   // running problem + a margin); W < n: a problem that settles inside the window is
                                   // handed to the full-width check (asm_wide_k) through the lists below
  // A problem that settles inside the window keeps its row of LAM (and of XH) until the full-width pass at the start of the next
  // round: that pass runs over the fp64 rows 0..wrows-1 of the round just finished as they stand -- rows are ordered by the stage
  // of the last active bound, kblk holds each 64-row block's k-range -- with rowprob saying whose row it is.  (Rounds 1-2 copied
  // the multipliers into rows of a second array handed out by one atomic per problem: 80 000 same-address atomics in the round
  // most sets settle in, 0.65 of that kernel's 0.95 ms, and 10 GB of workspace.)
  int* rowprob;                    // [rows] problem awaiting the full-width check in this row of LAM, or -1
  int wrows;                       // fp64 rows of the round whose settled problems the pass at hand checks
  unsigned char* wmark;            // [nseg] set by asm_wide_k for the problems it handled, counted and cleared by asm_bins (statistics)
  const double* xhw;               // [rows] = lam * H beyond the window (shapes the fused kernels' tiles do not fit)
  int tail_gi;                     // asm_tail_k: dual active-set steps (Goldfarb-Idnani) instead of Murty's single exchanges once block exchanges stop making progress (off: measured slower, see there)
  int use_wg;                      // sets of 145 .. 256 bounds go to the four-wave register kernels (qp_wg.h); 0: the single-wave / LDS-tile / slab kernels (A/B)
  int refine;                      // f32 rounds of sets of <= ASM_MLDS bounds get rows of LAM (fp64) and correct a result that can be final in fp64 (asm_reg_core)
  double refine_tol;               // residual of a corrected f32 solve, relative to max |b|, that counts as an fp64 solve (ASM_REFINE_TOL)
  int refine_later;                // ... from the next round on (this one may still be a plain f32 screen): a set that settles in f32 now is NOT sent to the fp64 kernel
  int early64;                     // an f32 round that moves at most this many bounds is followed by an fp64 round (0: only a settled set is)
  int* wflag;                      // [nseg] set by asm_wide_gemm_k when a bound beyond the window is violated (cleared by asm_wide_k)
  double* work;                    // [nseg][3] statistics: flops (m^3/3 + 2 m^2) and gathered bytes of the lambda kernels, flops of the f32 rounds
  double* scratch;                 // [pool][tiles(max_active) * ASM_TS] tile slabs of the queue kernel
  // outputs (problem-indexed, may be null except u)
  double* u_out;                   // [nseg][ldu]: the first nout entries of every solution (nout = n: whole sequences,
  int ldu, nout;                   // nout = nu: first moves only -- all the offline simulation keeps, lib/linearMPC.py:856)
  uint32_t* act_out;
  int* status_out;
  int* iters_out;
  int words;
};

__device__ __forceinline__ size_t tri(int i, int j) { return (size_t)i * (i + 1) / 2 + j; }

// x_unc -> first active-set estimate.  One WAVE per problem, four per workgroup (a workgroup of 256 threads per problem was
// launch-bound: 100 000 workgroups for 5 KB each).
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256) void asm_init_k(AsmDev d, int nrows) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= nrows) return;
  const size_t o = (size_t)p * d.np;
  if (p >= d.nseg) {                 // padding rows of the GEMM operands: never active (and not ASM_DONE: the check with P itself
    if (lane == 0) d.state[p] = ASM_CERT;   // selects its rows by that tag -- 107 padding rows of a 149-chain step would cost 0.1 ms)
    return;
  }
  // first set: the bounds x_unc violates in the leading part of the horizon (d.winit columns: where MPC saturates; a violation
  // further out is found by the full-width pass every problem goes through before it is accepted), or the guess
  // (without a guess the host has zeroed the bound states of the segment: only the leading wi are written here)
  const int wi = d.guess ? d.n : min(d.n, d.winit);
  const bool kfix = 64 % d.nu == 0;                          // k = r % nu is then the same for every chunk of 64 columns
  const double lbf = d.lb[(size_t)p * d.nu + lane % d.nu], ubf = d.ub[(size_t)p * d.nu + lane % d.nu];
  for (int r = (d.guess ? 0 : d.pred_w) + lane; r < wi; r += 64) {
    int s = 0;
    if (d.guess) { s = d.guess[(size_t)p * d.n + r]; if (s > 2) s = 0; }
    else {
      const double x = d.xunc[o + r];
      double lb = lbf, ub = ubf;
      if (!kfix) { const int k = r % d.nu; lb = d.lb[(size_t)p * d.nu + k]; ub = d.ub[(size_t)p * d.nu + k]; }
      s = x > ub ? 1 : (x < lb ? 2 : 0);
    }
    d.st[(size_t)p * d.n + r] = (unsigned char)s;
  }
  // inputs the tests below cannot reason about (every comparison with a NaN is false, so a NaN would pass for
  // "feasible"): NaN / Inf in x0, a NaN bound, lb > ub.  Such a problem is not solved at all.
  int invalid = 0;
  for (int k = lane; k < d.nu; k += 64) invalid |= !(d.lb[(size_t)p * d.nu + k] <= d.ub[(size_t)p * d.nu + k]);
  for (int k = lane; k < d.ka; k += 64) invalid |= !(fabs(d.x0[(size_t)p * d.ka + k]) <= 1.79e308);
  invalid = __any(invalid);
  // an empty set runs one round like the others: x = x_unc is checked and certified by asm_update_k / asm_wide_k
  if (lane == 0) {
    d.rounds[p] = 0; d.state[p] = invalid ? ASM_INVALID : ASM_RUN;
    d.prec[p] = (d.use_f32 && !d.guess && !(d.pred_w && d.pred_f64)) ? 0 : 1;   // a caller's guess (a predicted set) is expected to be right: fp64 at once
    d.redo[p] = 0;
    d.ninf_best[p] = 0x7fffffff; d.alpha[p] = ASM_GRACE;
    d.hi[p] = max(wi, d.guess ? 0 : d.pred_w);           // (a predicted set may hold bounds anywhere in the predictor's window)
  }
}
#endif

// Round stage 0a: ordered list of the active indices of every running problem and its length.
// (one workgroup; returns the size of the set to all threads; wsum: 4 ints of LDS)
__device__ __forceinline__ int asm_count_one(const AsmDev& d, int p, int hi_p, int* wsum) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned char* st = d.st + (size_t)p * d.n;
  const bool words = (d.n & 3) == 0;                         // rows of st are then 4-byte aligned: 4 bounds per load
  const int hi = min(d.n, hi_p);                             // nothing is active at or beyond hi
  const int nw = words ? (hi + 3) >> 2 : hi;                 // items (dwords or bytes), a contiguous run per thread
  const int per = (nw + 255) / 256;
  const int j0 = min(nw, tid * per), j1 = min(nw, j0 + per);
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(st);
  int c = 0;
  if (words) {
    for (int j = j0; j < j1; ++j) {
      const uint32_t w = sw[j];
      c += ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
    }
  } else {
    for (int j = j0; j < j1; ++j) c += st[j] != 0;
  }
  int inc = c;
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
  __syncthreads();                                           // wsum free again
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - c;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const int m = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (m <= d.max_active && c > 0) {
    int* idx = d.idxg + (size_t)p * d.max_active;
    int k = base;
    if (words) {
      for (int j = j0; j < j1; ++j) {
        const uint32_t w = sw[j];
        if (w & 0xffu) idx[k++] = 4 * j;
        if (w & 0xff00u) idx[k++] = 4 * j + 1;
        if (w & 0xff0000u) idx[k++] = 4 * j + 2;
        if (w & 0xff000000u) idx[k++] = 4 * j + 3;
      }
    } else {
      for (int j = j0; j < j1; ++j) if (st[j]) idx[k++] = j;
    }
  }
  return m;
}
// One WAVE per problem (four per workgroup): a round's sets live in the first few hundred bound states, a workgroup of 256
// threads and two barriers per problem was launch and barrier latency (0.1 ms per round at 100 000 problems).
__device__ __forceinline__ int asm_count_wave(const AsmDev& d, int p, int hi_p) {
  const int lane = threadIdx.x & 63;
  const unsigned char* st = d.st + (size_t)p * d.n;
  const bool words = (d.n & 3) == 0;
  const int hi = min(d.n, hi_p);
  const int nw = words ? (hi + 3) >> 2 : hi;
  const int per = (nw + 63) / 64;
  const int j0 = min(nw, lane * per), j1 = min(nw, j0 + per);
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(st);
  int c = 0;
  if (words) {
    for (int j = j0; j < j1; ++j) {
      const uint32_t w = sw[j];
      c += ((w & 0xffu) != 0) + ((w & 0xff00u) != 0) + ((w & 0xff0000u) != 0) + ((w & 0xff000000u) != 0);
    }
  } else {
    for (int j = j0; j < j1; ++j) c += st[j] != 0;
  }
  int inc = c;
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
  const int m = __shfl(inc, 63);
  if (m <= d.max_active && c > 0) {
    int* idx = d.idxg + (size_t)p * d.max_active;
    int k = inc - c;
    if (words) {
      for (int j = j0; j < j1; ++j) {
        const uint32_t w = sw[j];
        if (w & 0xffu) idx[k++] = 4 * j;
        if (w & 0xff00u) idx[k++] = 4 * j + 1;
        if (w & 0xff0000u) idx[k++] = 4 * j + 2;
        if (w & 0xff000000u) idx[k++] = 4 * j + 3;
      }
    } else {
      for (int j = j0; j < j1; ++j) if (st[j]) idx[k++] = j;
    }
  }
  return m;
}
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256) void asm_count_k(AsmDev d) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= d.nseg || d.state[p] != ASM_RUN) return;
  const int m = asm_count_wave(d, p, d.hi[p]);
  if ((threadIdx.x & 63) == 0) {
    d.mg[p] = m;
    if (m <= d.max_active && d.work) {
      const double md = (double)m;
      d.work[3 * p] += md * md * md / 3.0 + 2.0 * md * md;             // per-problem slots: same-address atomics
      d.work[3 * p + 1] += 8.0 * (md * (md + 1.0) / 2.0 + 2.0 * md);   // (one per workgroup) serialise the launch
      if (d.prec[p] == 0 && m <= (d.use_wg ? ASM_BIG32B : ASM_BIG32)) d.work[3 * p + 2] += md * md * md / 3.0 + 2.0 * md * md;   // ... of which in an f32 round
    }
    if (m > d.max_active) d.state[p] = ASM_FALLBACK;
  }
}
#endif

// Round stage 0b: the running problems get the rows 0..nrun-1 of LAM / XH (the GEMM covers only those)
// and a place in the list of their size class -- by exclusive scans over the problems (chunks of 1024 per
// workgroup: ranks inside the chunk from wave ballots, then the chunk totals) instead of one same-address
// atomic per problem, which cost more than the factorisations' launch.
// counters: [2] / [ASM_CNT_ROWS32] running problems solved in fp64 / f32 this round (the host adds them up), [1] sets too large for LDS, [3] largest active variable index, [4 + b] length
// of size-class list b.  counters[ASM_CNT_WIDEG..] (filled by asm_update_k, consumed by asm_wide_k earlier in
// the round) is reset here.
// scan columns: 0 large sets, 1 + list (lists 0 .. ASM_NLIST + 3), ASM_COL_ROW + prec * ASM_NKG + group (rows of LAM / LAM32), then the sum of
// (last active index + 1) and the max index
constexpr int ASM_COL_ROW = 5 + ASM_NLIST;
constexpr int ASM_COL_WCNT = ASM_COL_ROW + 2 * ASM_NKG, ASM_COL_WK = ASM_COL_WCNT + 1;
constexpr int ASM_WG_SETS = 256;   // largest set of the four-wave register kernels (qp_wg.h)
__device__ __forceinline__ int asm_scan_col(const AsmDev& d, int p, bool& run) {   // 0 large set, 1 + list otherwise
  run = p < d.nseg && d.state[p] == ASM_RUN;
  if (!run) return -1;
  const int m = d.mg[p];
  if (m > ASM_MLDS) return (d.prec[p] == 0 && m <= ASM_BIG32) ? 1 + ASM_NLIST : ((d.use_wg && m <= ASM_WG_SETS) ? 2 + ASM_NLIST : ((d.use_wg && d.prec[p] == 0 && m <= ASM_BIG32B) ? 3 + ASM_NLIST : ((d.use_wg && d.prec[p] == 1 && m <= ASM_BIG32B) ? 4 + ASM_NLIST : 0)));   // (prec 2: the refinement gave up on this set)
  const int b = max((m + 15) / 16, 4) - 4;
  return 1 + (d.prec[p] == 0 ? ASM_NBIN + b : b);
}
// group by the stage of the last active bound, relative to last round's overall last stage (d.kref)
__host__ __device__ inline int asm_kgroup_of(int kl, int kref, int nu) {
  const int st10 = 10 * (kl / nu), sref = kref / nu;
  return st10 <= 7 * sref ? 0 : (st10 <= 8 * sref ? 1 : 2);
}
__host__ __device__ inline int asm_kgroup_bound_of(int g, int kref, int nu) {   // largest index a member of group g < 2 can have
  const int sref = kref / nu;
  return ((g == 0 ? 7 : 8) * sref / 10 + 1) * nu - 1;
}
__device__ __forceinline__ int asm_kgroup(const AsmDev& d, int kl) { return asm_kgroup_of(kl, d.kref, d.nu); }
__device__ __forceinline__ int asm_kgroup_bound(const AsmDev& d, int g) { return asm_kgroup_bound_of(g, d.kref, d.nu); }
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(1024) void asm_bins_a_k(AsmDev d) {
  __shared__ int wtot[ASM_NSCAN][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = blockIdx.x * 1024 + tid;
  for (int i = p; i < d.nkblk; i += gridDim.x * 1024) d.kblk[i] = 0;      // filled by asm_bins_b_k
  bool run;
  const int col = asm_scan_col(d, p, run);
  int kl = 0;
  if (run) { const int m = d.mg[p]; if (m > 0) kl = d.idxg[(size_t)p * d.max_active + m - 1]; }
  const unsigned long long lt = (1ull << lane) - 1ull;
  const bool r32 = (col > ASM_NBIN + (d.refine ? ASM_NBIN : 0) && col <= 1 + ASM_NLIST) || col == 3 + ASM_NLIST;   // solved in f32 this round AND not corrected: row of LAM32 / XH32
  const int rcol = run ? ASM_COL_ROW + (r32 ? ASM_NKG : 0) + asm_kgroup(d, kl) : -1;
  int myrank = 0, myrow = 0;
#pragma unroll
  for (int c = 0; c < 2 * ASM_NKG; ++c) {
    const unsigned long long mk = __ballot(rcol == ASM_COL_ROW + c);
    if (rcol == ASM_COL_ROW + c) myrow = __popcll(mk & lt);
    if (lane == 0) wtot[ASM_COL_ROW + c][wave] = __popcll(mk);
  }
#pragma unroll
  for (int c = 0; c <= ASM_NLIST + 4; ++c) {
    const unsigned long long mk = __ballot(col == c);
    if (col == c) myrank = __popcll(mk & lt);
    if (lane == 0) wtot[c][wave] = __popcll(mk);
  }
  int km = kl, ks = run ? kl + 1 : 0;                        // ks: columns of LAM this row really has
  // problems the full-width pass at the start of this round handled (asm_wide_k marked them): number, sum of (last active index + 1)
  const bool wm = p < d.nseg && d.wmark[p] != 0;
  int wk = 0;
  if (wm) { const int m = d.mg[p]; wk = m > 0 ? d.idxg[(size_t)p * d.max_active + m - 1] + 1 : 1; }
  const unsigned long long wmk = __ballot(wm);
  for (int off = 32; off > 0; off >>= 1) { km = max(km, __shfl_xor(km, off)); ks += __shfl_xor(ks, off); wk += __shfl_xor(wk, off); }
  if (lane == 0) { wtot[ASM_NSCAN - 1][wave] = km; wtot[ASM_NSCAN - 2][wave] = ks; wtot[ASM_COL_WCNT][wave] = __popcll(wmk); wtot[ASM_COL_WK][wave] = wk; }
  __syncthreads();
  if (run) {
    for (int w = 0; w < wave; ++w) { myrow += wtot[rcol][w]; myrank += wtot[col][w]; }
    d.row[p] = myrow;
    d.lrank[p] = myrank;
  }
  if (tid < ASM_NSCAN) {
    int t = 0;
    if (tid == ASM_NSCAN - 1) { for (int w = 0; w < 16; ++w) t = max(t, wtot[tid][w]); }
    else { for (int w = 0; w < 16; ++w) t += wtot[tid][w]; }
    d.ctot[(size_t)blockIdx.x * ASM_NSCAN + tid] = t;
  }
}
#endif
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(1024) void asm_bins_b_k(AsmDev d) {
  __shared__ int base[ASM_NSCAN], total[ASM_NSCAN];
  const int tid = threadIdx.x;
  const int p = blockIdx.x * 1024 + tid;
  if (tid < ASM_NSCAN) {
    int t = 0, all = 0;
    if (tid == ASM_NSCAN - 1) { for (int j = 0; j < (int)gridDim.x; ++j) t = max(t, d.ctot[(size_t)j * ASM_NSCAN + tid]); all = t; }
    else {
      for (int j = 0; j < (int)gridDim.x; ++j) { const int v = d.ctot[(size_t)j * ASM_NSCAN + tid]; if (j < (int)blockIdx.x) t += v; all += v; }
    }
    base[tid] = t; total[tid] = all;
  }
  __syncthreads();
  if (blockIdx.x == gridDim.x - 1 && tid < ASM_NSCAN) {      // one workgroup publishes the totals
    if (tid == ASM_NSCAN - 1) {
      d.counters[3] = total[tid];
    } else if (tid == ASM_NSCAN - 2) d.counters[0] = total[tid];   // sum of (last active index + 1): algorithmic k of the GEMM
    else if (tid == ASM_COL_WCNT) d.counters[ASM_CNT_WIDE + 1] = total[tid];
    else if (tid == ASM_COL_WK) d.counters[ASM_CNT_WKSUM] = total[tid];
    else if (tid == 0) d.counters[1] = total[0];
    else if (tid <= ASM_NLIST + 4) d.counters[asm_list_counter(tid - 1)] = total[tid];
    else if (tid == ASM_COL_ROW) { int t = 0; for (int g = 0; g < ASM_NKG; ++g) t += total[ASM_COL_ROW + g]; d.counters[2] = t; }
    else if (tid == ASM_COL_ROW + ASM_NKG) { int t = 0; for (int g = 0; g < ASM_NKG; ++g) t += total[ASM_COL_ROW + ASM_NKG + g]; d.counters[ASM_CNT_ROWS32] = t; }
  }
  if (p < d.nseg) d.wmark[p] = 0;
  bool run;
  const int col = asm_scan_col(d, p, run);
  if (!run) return;
  const int m = d.mg[p];
  const int kl = m > 0 ? d.idxg[(size_t)p * d.max_active + m - 1] : 0;
  const bool r32 = (col > ASM_NBIN + (d.refine ? ASM_NBIN : 0) && col <= 1 + ASM_NLIST) || col == 3 + ASM_NLIST;
  const int c0 = ASM_COL_ROW + (r32 ? ASM_NKG : 0), g = asm_kgroup(d, kl);
  int row = d.row[p] + base[c0 + g];
  for (int gg = 0; gg < g; ++gg) row += total[c0 + gg];      // rows ordered by group
  d.row[p] = row;
  d.rowk[p] = r32;
  atomicMax(&d.kblk[(r32 ? d.nkblk / 2 : 0) + (row >> 6)], kl);   // last active bound of this 64-row block
  const int pos = base[col] + d.lrank[p];
  if (col == 0) d.biglist[pos] = p;
  else d.binlist[(size_t)(col - 1) * d.nseg + pos] = p;
}
#endif

// Workgroup-per-problem variant (the first implementation; the solver uses it for the rare sets beyond 176
// bounds, BIG = 1): r_A = x_unc,A - b_A, S = H_AA as 16 x 16 fp64 tiles, blocked right-looking Cholesky:
// diagonal tile + its inverse by wave 0 (asm_diag16), TRSM and trailing updates on v_mfma_f64_16x16x4_f64 by
// all four waves, blocked triangular solves by wave 0.  3 barriers per block column.
// BIG = 1: persistent workgroups walk the big-set queue with the tiles in a global scratch slab (L2).
// BIG = 0: one workgroup per entry of size-class list `bin`, tiles in LDS (kept for scripts/micro comparisons).
typedef double f64x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rdlane_d(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ f64x4_t tile_mma_nt(const double* A, const double* B, int lane) {
  f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int li = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[li * 17 + 4 * s + kq], B[li * 17 + 4 * s + kq], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ double* asm_tile(double* T, int I, int J) { return T + ((size_t)I * (I + 1) / 2 + J) * ASM_TS; }

#define ASM_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <class T> __device__ __forceinline__ int asm_diag16(const T* Tk, T* Yt, int lane, const T* Id = nullptr);
__device__ __noinline__ int asm_diag16_call(const double* Tk, double* Yt, int lane);

// One workgroup, one problem: rA <- lam = (H_AA)^-1 (x_unc,A - b_A) for the set idx[0..m).  rA [>= 16 ceil(m/16)]
// and Yt [ASM_TS] in LDS, T: the lower 16 x 16 tiles (LDS or a global slab).  Returns 1 (to all threads) when H_AA
// is not positive definite in fp64.  *s_bad: an int in LDS.
__device__ __forceinline__ int asm_tile_solve(const AsmDev& d, int p, int m, const int* idx, double* rA, double* Yt, double* T,
                                              int* s_bad) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __syncthreads();                                       // previous use of the buffers fully retired
  if (tid == 0) *s_bad = 0;
  const size_t o = (size_t)p * d.np;
  const unsigned char* st = d.st + (size_t)p * d.n;
  const int mb = (m + 15) / 16;
  __syncthreads();
  for (int i = tid; i < mb * 16; i += 256) {
    double v = 0.0;
    if (i < m) {
      const int a = idx[i], k = a % d.nu;
      v = d.xunc[o + a] - (st[a] == 1 ? d.ub[(size_t)p * d.nu + k] : d.lb[(size_t)p * d.nu + k]);
    }
    rA[i] = v;
  }
  // ---- gather S = H_AA into tiles (diagonal tiles complete, pad = identity)
  const int ntile = mb * (mb + 1) / 2;
  {
    // thread (ti, tj) fetches element (ti, tj) of every tile; 4 tiles per trip keep 4 loads in flight
    const int ti = tid >> 4, tj = tid & 15;
    int I = 0, J = 0;
    for (int t0 = 0; t0 < ntile; t0 += 4) {
      double v[4];
      int tI[4], tJ[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        tI[u] = I; tJ[u] = J;
        const int gi = 16 * I + ti, gj = 16 * J + tj;
        v[u] = gi == gj ? 1.0 : 0.0;
        if (t0 + u < ntile && gi < m && gj < m) v[u] = d.H[(size_t)idx[gi] * d.np + idx[gj]];
        if (++J > I) { J = 0; ++I; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (t0 + u < ntile) asm_tile(T, tI[u], tJ[u])[ti * 17 + tj] = v[u];
    }
  }
  __syncthreads();
  // ---- blocked Cholesky
  for (int K = 0; K < mb; ++K) {
    double* TKK = asm_tile(T, K, K);
    if (wave == 0) {
      const int bad = asm_diag16_call(TKK, Yt, lane);  // Y_K = L_KK^-1 -> Yt (L_KK itself is not needed again)
      if (bad && lane == 0) *s_bad = 1;
    }
    __syncthreads();
    for (int I = K + 1 + wave; I < mb; I += 4) {       // TRSM: T(I,K) <- T(I,K) Y'
      double* TIK = asm_tile(T, I, K);
      const f64x4_t acc = tile_mma_nt(TIK, Yt, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) TIK[((lane >> 4) + 4 * r) * 17 + (lane & 15)] = acc[r];
    }
    __syncthreads();
    {                                                   // trailing: T(I,J) -= T(I,K) T(J,K)'
      int b = 0;
      for (int I = K + 1; I < mb; ++I)
        for (int J = K + 1; J <= I; ++J, ++b) {
          if ((b & 3) != wave) continue;
          const f64x4_t acc = tile_mma_nt(asm_tile(T, I, K), asm_tile(T, J, K), lane);
          double* TIJ = asm_tile(T, I, J);
#pragma unroll
          for (int r = 0; r < 4; ++r) TIJ[((lane >> 4) + 4 * r) * 17 + (lane & 15)] -= acc[r];
        }
    }
    // keep Y_K for the solves: its transpose goes into the upper part of T(K,K) (diagonal included)
    if (wave == 0 && lane < 16) {
      const int row = lane;
#pragma unroll
      for (int k = 0; k < 16; ++k) if (k >= row) TKK[row * 17 + k] = Yt[k * 17 + row];   // Y[k][row], k >= row
    }
    __syncthreads();
  }
  if (*s_bad) return 1;
  // ---- solves by wave 0:  L y = r (forward), L' lam = y (backward), 16-blocks
  if (wave == 0) {
    const int i = lane & 15, kq = lane >> 4;
    for (int K = 0; K < mb; ++K) {                       // forward
      double t = 0.0;
      for (int J = 0; J < K; ++J) {
        const double* TKJ = asm_tile(T, K, J);
#pragma unroll
        for (int k = 0; k < 4; ++k) t += TKJ[i * 17 + 4 * kq + k] * rA[16 * J + 4 * kq + k];
      }
      t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
      t = rA[16 * K + i] - t;                            // all 64 lanes hold t_i (i = lane & 15)
      // y_K = Y_K t,  Y_K[i][k] (k <= i) stored at TKK[k][i]
      const double* TKK = asm_tile(T, K, K);
      double yv = t * TKK[i * 17 + i];
#pragma unroll
      for (int k = 0; k < 16; ++k) { const double tk = __shfl(t, k); if (k < i) yv += TKK[k * 17 + i] * tk; }
      if (lane < 16) rA[16 * K + i] = yv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // y_K visible to every lane before the next block row
    }
    for (int K = mb - 1; K >= 0; --K) {                  // backward
      double t = 0.0;
      for (int I = K + 1; I < mb; ++I) {
        const double* TIK = asm_tile(T, I, K);
#pragma unroll
        for (int k = 0; k < 4; ++k) t += TIK[(4 * kq + k) * 17 + i] * rA[16 * I + 4 * kq + k];
      }
      t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
      t = rA[16 * K + i] - t;
      // lam_K = Y_K' t:  lam_i = sum_{k >= i} Y_K[k][i] t_k,  Y_K[k][i] (k > i) stored at TKK[i][k]
      const double* TKK = asm_tile(T, K, K);
      double lv = t * TKK[i * 17 + i];
#pragma unroll
      for (int k = 0; k < 16; ++k) { const double tk = __shfl(t, k); if (k > i) lv += TKK[i * 17 + k] * tk; }
      if (lane < 16) rA[16 * K + i] = lv;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  return 0;
}

#if ASM_OWN_KERNELS
template <int BIG>
__global__ __launch_bounds__(256, 4) void asm_lambda_tile_k(AsmDev d, int bin) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  const int cap = BIG ? d.max_active : asm_bin_cap(bin);  // rhs capacity of this variant
  double* rA = sm;                                       // [cap]
  double* Yt = rA + cap;                                 // inverse of the current diagonal tile
  double* T = BIG ? d.scratch + (size_t)blockIdx.x * ((size_t)(cap / 16) * (cap / 16 + 1) / 2 * ASM_TS)
                  : Yt + ASM_TS;                         // lower tiles
  const int nitem = BIG ? d.counters[1] : d.counters[4 + bin];
  const int* list = BIG ? d.biglist : d.binlist + (size_t)bin * d.nseg;
  for (int it = blockIdx.x; it < nitem; it += gridDim.x) {
    const int p = list[it];
    const int* idx = d.idxg + (size_t)p * d.max_active;
    const int m = d.mg[p];
    if (asm_tile_solve(d, p, m, idx, rA, Yt, T, &s_bad)) { if (tid == 0) d.state[p] = ASM_FALLBACK; continue; }
    if (tid == 0) d.prec[p] = 1;                           // an fp64 solve
    double* lrow = d.lam + (size_t)d.row[p] * d.np;
    for (int i = tid; i < m; i += 256) lrow[idx[i]] = rA[i];   // the rest of the row is zero (asm_update_k)
  }
}
#endif

// ---- register-resident version, one instantiation per number MB of 16-blocks: ONE WAVE PER PROBLEM,
// no workgroup barriers, straight-line code.
// Accumulator t(I,J) of the wave holds the TRANSPOSE of tile (I,J) (I >= J) in the MFMA C/D layout
// (reg r of lane (li, lq): element [li][lq + 4r] of the tile).  The C/D registers of a transposed tile
// are exactly the A/B operand fragments of the tile itself (lane (li, lq), step s: [li][4s + lq]), so
//   TRSM      L(I,K)' = Y_K S(I,K)'            A = fragments of Y_K (LDS),  B = t(I,K)
//   trailing  S(I,J)' -= L(J,K) L(I,K)'         A = -t(J,K),                 B = t(I,K)
// run MFMA register to register; LDS only transposes the diagonal tile (C layout -> rows on lanes)
// and its inverse (columns on lanes -> fragments): two fences per block column.  Both substitutions
// stay in registers: sums over a tile row are DPP reductions inside the 16-lane rows, sums over the
// four lane rows use v_permlane16/32_swap.
// The 512 registers of a wave hold 32 tiles next to the working set; for MB = 8, 9 (10, 11) the strictly
// lower tiles of the first NL = 2 (4) block columns -- touched by one TRSM, one pass as operands and the
// backward substitution only -- live in LDS instead (C layout, 2 KB each).  10- and 11-block sets then need
// ~70 KB of LDS per wave: two waves per workgroup (asm_lambda_reg2_k).
//
// The same code runs in f32 (AsmNum<float>, v_mfma_f32_16x16x4_f32): tiles take half the registers, so two
// waves share a SIMD and hide each other's pivot chains, and the chain itself is shorter (one v_readlane
// per broadcast, v_rsq_f32 without refinement).  f32 multipliers are good to ~cond(S) * 6e-8 -- enough to
// decide which bounds to add or drop, not to deliver them: a problem runs its rounds in f32 until its set
// settles, then in fp64 (prec[p]) until it settles again, and only fp64 results are ever accepted.
template <class T> struct AsmNum;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <> struct AsmNum<double> {
  using v4 = f64x4_t;
  static constexpr bool F32 = false;
  // C/D layout of v_mfma_f64_16x16x4_f64: reg r of lane (li, lq) is row lq + 4r, column li
  __host__ __device__ static constexpr int kr(int lq, int r) { return lq + 4 * r; }
  __device__ static __forceinline__ v4 mfma(double a, double b, v4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ double rdlane(double x, int l) { return rdlane_d(x, l); }
  __device__ static __forceinline__ double rsq(double x) { return rsqrt(x); }

  template <int CTRL> __device__ static __forceinline__ double dpp(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  }
  __device__ static __forceinline__ double swap16(double x) {   // x[l] + x[l ^ 16]
    const auto a = __builtin_amdgcn_permlane16_swap(__double2loint(x), __double2loint(x), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__double2hiint(x), __double2hiint(x), false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  }
  __device__ static __forceinline__ double swap32(double x) {   // x[l] + x[l ^ 32]
    const auto a = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  }
};
template <> struct AsmNum<float> {
  using v4 = f32x4_t;
  static constexpr bool F32 = true;
  // C/D layout of v_mfma_f32_16x16x4_f32: reg r of lane (li, lq) is row 4 lq + r, column li.  The k index a
  // register stands for only has to be the same for both operands of an MFMA, which it is: both come from
  // accumulators (or from fragments read with the same kr()).
  __host__ __device__ static constexpr int kr(int lq, int r) { return 4 * lq + r; }
  __device__ static __forceinline__ v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  __device__ static __forceinline__ float rdlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
  __device__ static __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }

  template <int CTRL> __device__ static __forceinline__ float dpp(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, false));
  }
  __device__ static __forceinline__ float swap16(float x) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(x), __float_as_int(x), false, false);
    return __int_as_float(a[0]) + __int_as_float(a[1]);
  }
  __device__ static __forceinline__ float swap32(float x) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_int(x), __float_as_int(x), false, false);
    return __int_as_float(a[0]) + __int_as_float(a[1]);
  }
};
// sum over the 16 lanes of a lane row, result in every lane of the row
template <class T> __device__ __forceinline__ T rowsum16(T x) {
  using N = AsmNum<T>;
  x += N::template dpp<0xB1>(x);        // quad_perm [1,0,3,2]
  x += N::template dpp<0x4E>(x);        // quad_perm [2,3,0,1]
  x += N::template dpp<0x141>(x);       // row_half_mirror
  x += N::template dpp<0x140>(x);       // row_mirror
  return x;
}
// Four row sums at once.  Generic: four rowsum16.  f32: the four chains interleaved by hand as v_add_f32_dpp (the DPP
// operand folded into the add; hipcc emits v_mov_b32_dpp + v_add_f32 + s_nop 1 per step -- 240 moves and ~160 nops per
// problem).  A DPP read needs two wait states after the vector write of its source: the three other chains' adds sit in
// between, so no s_nop is needed inside the block; the one in front covers whatever wrote the inputs.
template <class T> __device__ __forceinline__ void rowsum16x4(T (&v)[4]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = rowsum16<T>(v[r]);
}
template <> __device__ __forceinline__ void rowsum16x4<float>(float (&v)[4]) {
  asm volatile(
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n"
      "v_add_f32_dpp %3, %3, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}
// sum over the four lane rows (lanes l, l^16, l^32, l^48), result in all of them
template <class T> __device__ __forceinline__ T xsum4(T x) { return AsmNum<T>::swap32(AsmNum<T>::swap16(x)); }

// tiles of the first NL block columns kept in LDS; LDS elements per wave
#ifndef ASM_NL_OF_8
#define ASM_NL_OF_8 2              // (scripts/micro/occ_probe.hip overrides this to try other register / LDS splits)
#endif
template <class T> __host__ __device__ constexpr int asm_nl(int mb) {
  return mb <= 7 ? 0 : (mb == 8 ? ASM_NL_OF_8 : (mb <= 9 ? 2 : 4));
}
template <class T> __host__ __device__ constexpr int asm_nlt(int mb) { return asm_nl<T>(mb) * (mb - 1) - asm_nl<T>(mb) * (asm_nl<T>(mb) - 1) / 2; }
template <class T> __host__ __device__ constexpr int asm_rw(int mb) { return 2 * ASM_TS + 2 * mb * 16 + asm_nlt<T>(mb) * 256; }

__host__ __device__ constexpr int asm_tix(int I, int J) { return I * (I + 1) / 2 + J; }
// linear index t over the pairs (J, I), 0 <= J <= I < n, J-major (t = 0: (0, 0)): the J / the I of pair t
__host__ __device__ constexpr int asm_tri_row(int n, int t) { int J = 0; while (t >= n - J) { t -= n - J; ++J; } return J; }
__host__ __device__ constexpr int asm_tri_col(int n, int t) { int J = 0; while (t >= n - J) { t -= n - J; ++J; } return J + t; }
template <int V> struct asm_ic { static constexpr int value = V; };
template <int B, int E, class F>
__device__ __forceinline__ void asm_sfor(F&& f) {          // f(asm_ic<B>{}), ..., f(asm_ic<E-1>{}): indices are constants
  if constexpr (B < E) { f(asm_ic<B>{}); asm_sfor<B + 1, E>(f); }
}

// 16 x 16 diagonal tile: Cholesky L and Y = L^-1 in one sweep of 16 column steps, no LDS traffic in
// the chain.  Lanes 0..31 (two mirrored 16-lane rows) hold the rows of the tile, lanes 32..63 the
// columns of Y (lane 32 + j: Y[.][j], starting from e_j): with w = x[cc] / L[cc][cc] both halves run
// the SAME update  x[c2] -= w * L[c2][cc]  -- it is the elimination step for the first half and the
// forward substitution L Y = I for the second.  Pivot and L[c2][cc] are wave-uniform (v_readlane from
// lanes cc / c2), used once each, so they do not pile up in SGPRs.  16 live values per lane.
// Tk: the tile in LDS (row-major, stride 17, read only); Yt receives Y (same layout).
// Id (optional): an identity tile in LDS (same layout): the lanes of the inverse half then READ their unit vectors
// instead of selecting them (one pointer select instead of 16 v_cndmask per tile).
// The same sweep in three pieces (begin / 16 steps / end) for callers that interleave other work with the steps (qp_wg.h).
template <class T> __device__ __forceinline__ void asm_diag16_begin(T (&x)[16], const T* Tk, const T* Id, int lane) {
  const T* src = lane >= 32 ? Id : Tk;
  const int row = lane & 15;
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = src[row * 17 + k];
}
template <class T, int CC> __device__ __forceinline__ void asm_diag16_step(T (&x)[16]) {
  using N = AsmNum<T>;
  const T dd = N::rdlane(x[CC], CC);
  const T w = x[CC] * N::rsq(dd);
  x[CC] = w;
#pragma unroll
  for (int c2 = CC + 1; c2 < 16; ++c2) x[c2] -= w * N::rdlane(w, c2);
}
template <class T> __device__ __forceinline__ int asm_diag16_end(const T (&x)[16], T* Yt, int lane) {
  using N = AsmNum<T>;
  const int bad = !(N::rdlane(x[15], 15) > T(0));
  if (lane >= 48) {
    const int row = lane & 15;
#pragma unroll
    for (int k = 0; k < 16; ++k) Yt[k * 17 + row] = x[k];
  }
  return bad;
}

template <class T>
__device__ __forceinline__ int asm_diag16(const T* Tk, T* Yt, int lane, const T* Id) {
  using N = AsmNum<T>;
  const int row = lane & 15;
  const bool inv_half = lane >= 32;
  T x[16];
  if (Id) {
    const T* src = inv_half ? Id : Tk;
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = src[row * 17 + k];
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      T a = Tk[row * 17 + k];
      asm volatile("" : "+v"(a));                          // keeps the 16 loads unconditional and back to back (the
      x[k] = inv_half ? (k == row ? T(1) : T(0)) : a;      // select would otherwise become 16 divergent branches)
    }
  }
  // (the broadcasts stay on v_readlane: the same sweep with ds_bpermute -- DS instructions, 47 % fewer vector
  // instructions per tile -- ran 25 % SLOWER: the crossbar's latency sits in the pivot chain; so did a sweep shared by the
  // four waves of a workgroup over DPP row_newbcast, behind two barriers per block column: +26 %)
#pragma unroll
  for (int cc = 0; cc < 16; ++cc) {
    const T dd = N::rdlane(x[cc], cc);
    const T w = x[cc] * N::rsq(dd);                        // L[row][cc]  |  Y[cc][row]
    x[cc] = w;
#pragma unroll
    for (int c2 = cc + 1; c2 < 16; ++c2) x[c2] -= w * N::rdlane(w, c2);
  }
  // Not positive definite in this precision <=> some pivot was <= 0 (or NaN): its rsq is NaN / Inf, which every later
  // column inherits through the updates -- the LAST diagonal entry L[15][15] (lane 15) then fails "> 0".  One test
  // instead of a running minimum beside the chain (which cost a v_writelane and a compare per pivot).
  const int bad = !(N::rdlane(x[15], 15) > T(0));
  if (lane >= 48) {
#pragma unroll
    for (int k = 0; k < 16; ++k) Yt[k * 17 + row] = x[k];
  }
  return bad;
}

__device__ __noinline__ int asm_diag16_call(const double* Tk, double* Yt, int lane) { return asm_diag16<double>(Tk, Yt, lane); }


// In-kernel phase stamps of ONE wave (scripts/micro only: -DASM_STAMPS): shader-clock reads at the phase boundaries of
// wave 0 of workgroup ASM_STAMP_WG, dumped to a device array.  Compiles to nothing otherwise.
#ifdef ASM_STAMPS
__device__ unsigned long long asm_stamp_buf[64];
#ifndef ASM_STAMP_WG
#define ASM_STAMP_WG 0
#endif
#define ASM_STAMP(i) do { if (wg == ASM_STAMP_WG && wave == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) asm_stamp_buf[i] = t_; } } while (0)
#else
#define ASM_STAMP(i) do { } while (0)
#endif

// The solve of one wave, shared by the round kernels (asm_lambda_reg) and the fused small-problem kernel (asm_small_k):
// gather -S = -H_AA from the index list ix[0 .. 16 MB) (LDS; entries beyond m repeat the last index), blocked Cholesky with the
// forward substitution riding along, backward substitution.  rv: right-hand side [16 MB] (zero beyond m).  lam[I] receives the
// multiplier of bound 16 I + li (every lane row holds a copy).  Returns non-zero (lam untouched) when S is not positive
// definite in this precision.  lt: base of the LDS-resident tiles of this wave PLUS lane.  NL: block columns whose tiles live there.
// MB >= 5 expects 16 (MB - 1) < m <= 16 MB (the size classes of the rounds) unless ANYM is set.
// REF (the f32 instances of the rounds, do_refine): when every multiplier of the f32 solve has the sign that keeps its bound -- the
// set may be final --, ONE correction in fp64 with the factor at hand: r = b - S lam with S gathered again from the fp64 inverse (fp64
// FMAs, lam broadcast from LDS, the loads of two columns in flight), L L' dl = r through the tiles in the registers, lam + dl in
// fp64 (rf.lam64; returns 2).  The error goes from cond(S) * 6e-8 to ~(cond(S) * 6e-8)^2: 1e-14 on the CDU plant, whose sets have
// cond(S) = 11 .. 18 -- what an fp64 factorisation leaves.  It is not taken on trust: the window GEMM of the round forms S (lam + dl)
// for everybody, and asm_update_k accepts the set only if that residual is at the fp64 floor (else the set runs again in the fp64
// kernel).  The round that used to confirm a settled set in fp64 -- the largest launch of a step -- is this one.
template <int MB> struct AsmRefine {
  const double *xunc, *lb, *ub;    // of this problem: [np], [nu], [nu]
  const unsigned char* st;         // [n]
  const int* idx;                  // the ordered active indices (global memory)
  double lam64[MB];                // out: multiplier of bound 16 I + li
};
template <class T, int MB, int NL, bool ANYM, bool REF>
__device__ __forceinline__ int asm_reg_core(const AsmDev& d, int m, const int* ix, T* dt, T* Yt, T* ys, const T* rv, T* lt, const T* idt,
                                            T (&lam)[MB], int lane, int wg, int wave, AsmRefine<MB>& rf, bool do_refine) {
  using N = AsmNum<T>;
  using V4 = typename N::v4;
  const int li = lane & 15, lq = lane >> 4;
  auto slot = [](int I, int J) { return J * (MB - 1) - J * (J - 1) / 2 + I - J - 1; };   // tile (I,J), J < NL, I > J
  (void)wg; (void)wave;
  ASM_STAMP(60);                                           // (right-hand side and index list in LDS)
  // ---- gather: t(I,J)[r] = -S[16 I + li][16 J + kr(lq, r)], read as Pinv[row of (J, lq, r)][col of (I, li)].
  // The not yet factored tiles hold MINUS the Schur complement, so the trailing update is a plain
  // accumulation (the MFMAs have no negate modifier; a VALU negation would cost a pass over the operands).
  V4 C[MB * (MB + 1) / 2];
  {
    int gcol[MB];                                          // Pinv index of active bound 16 I + li
#pragma unroll
    for (int I = 0; I < MB; ++I) gcol[I] = ix[16 * I + li];            // (entries beyond m repeat the last index)
    // addresses as 32-bit byte offsets from the (wave-uniform) base of Pinv: one v_add per load instead of 64-bit pointer
    // arithmetic (np^2 entries of 4 / 8 bytes stay below 2^32 for every n the library accepts: np <= 23 k)
    using HT0 = typename std::conditional<N::F32, float, double>::type;
    const char* const Hbase = N::F32 ? reinterpret_cast<const char*>(d.H32) : reinterpret_cast<const char*>(d.H);
    unsigned gco[MB];
#pragma unroll
    for (int I = 0; I < MB; ++I) gco[I] = (unsigned)gcol[I] * (unsigned)sizeof(HT0);
#pragma unroll
    for (int J = 0; J < MB; ++J) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gj = 16 * J + N::kr(lq, r);
        using HT = typename std::conditional<N::F32, float, double>::type;
        const unsigned rowoff = (unsigned)ix[gj] * (unsigned)d.np * (unsigned)sizeof(HT);
#pragma unroll
        for (int I = J; I < MB; ++I) {                     // unconditional (clamped) loads, then select
          const int gi = 16 * I + li;
          const HT v = *reinterpret_cast<const HT*>(Hbase + (rowoff + gco[I]));
          // a class of MB >= 5 blocks holds sets with 16 (MB - 1) < m <= 16 MB: only the last block row / column can
          // reach beyond m (padding = identity); the other tiles need no select
          // (ANYM: any m <= 16 MB -- every tile gets the select)
          const bool edge = ANYM || MB <= 4 || I == MB - 1;
          const T e = !edge ? (T)(-v) : ((gi < m && gj < m) ? (T)(-v) : (gi == gj ? T(-1) : T(0)));
          if (J < NL && I > J) lt[slot(I, J) * 256 + r * 64] = e;
          else C[asm_tix(I, J)][r] = e;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  ASM_STAMP(1);                                            // (the gathered values are not waited for yet)
  T ps[MB];                                                // lane-local partial sums of  sum_J L(I,J) y_J
#pragma unroll
  for (int I = 0; I < MB; ++I) ps[I] = T(0);
  int bad = 0;
  // ---- blocked Cholesky (right-looking) with the forward substitution riding along.  The diagonal step of
  // block column K + 1 (a long dependent VALU chain) is issued right after the one tile it needs, in the same
  // stretch of code as the rest of column K's trailing MFMAs, which do not depend on it.
  auto trail = [&](auto Kc, auto Jc, auto Ic, const V4* P) {   // tile (I,J) += L(J,K) L(I,K)'   (accumulates -Schur)
    constexpr int K = decltype(Kc)::value, J = decltype(Jc)::value, I = decltype(Ic)::value;
    constexpr bool in_lds = J < NL && I > J;
    V4 acc;
    if (in_lds) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = lt[slot(I, J) * 256 + r * 64];
    } else {
      acc = C[asm_tix(I, J)];
    }
    const V4 a = K < NL ? P[J] : C[asm_tix(J, K)], b = K < NL ? P[I] : C[asm_tix(I, K)];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = N::mfma(a[s4], b[s4], acc);
    if (in_lds) {
#pragma unroll
      for (int r = 0; r < 4; ++r) lt[slot(I, J) * 256 + r * 64] = acc[r];
    } else {
      C[asm_tix(I, J)] = acc;
    }
  };
#pragma unroll
  for (int r = 0; r < 4; ++r) dt[li * 17 + N::kr(lq, r)] = -C[asm_tix(0, 0)][r];
  ASM_FENCE();
  ASM_STAMP(2);
  bad |= asm_diag16<T>(dt, Yt, lane, idt);
  ASM_FENCE();
  ASM_STAMP(3);
  asm_sfor<0, MB>([&](auto Kc) {
    constexpr int K = decltype(Kc)::value;
    __builtin_amdgcn_sched_barrier(0);
    ASM_STAMP(4 + 3 * K);
    T yf[4];                                               // fragments of -Y_K
    V4 Yc;                                                 // Y_K' in C layout: [li][kr] of Y' = Y[kr(lq, r)][li]
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { yf[s4] = -Yt[li * 17 + N::kr(lq, s4)]; Yc[s4] = Yt[N::kr(lq, s4) * 17 + li]; }
    C[asm_tix(K, K)] = Yc;
    // y_K = Y_K (r_K - sum_{J<K} L(K,J) y_J)
    const T tK = rv[16 * K + li] - (K ? xsum4<T>(ps[K]) : T(0));
    T yq[4];                                               // y_K[kr(lq, r)]
#pragma unroll
    for (int r = 0; r < 4; ++r) yq[r] = Yc[r] * tK;
    rowsum16x4<T>(yq);
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ys[16 * K + N::kr(lq, r)] = yq[r];
    }
    V4 P[MB];                                              // the panel of an LDS-resident column (K < NL)
#pragma unroll
    for (int I = K + 1; I < MB; ++I) {                     // TRSM: L(I,K)' = Y_K S(I,K)'
      V4 b;
      if (K < NL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = lt[slot(I, K) * 256 + r * 64];
      } else {
        b = C[asm_tix(I, K)];
      }
      V4 acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = N::mfma(yf[s4], b[s4], acc);
      if (K < NL) {
        P[I] = acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) lt[slot(I, K) * 256 + r * 64] = acc[r];
      } else {
        C[asm_tix(I, K)] = acc;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) ps[I] += acc[r] * yq[r];
    }
    ASM_STAMP(5 + 3 * K);                                  // (TRSM of column K issued)
    if constexpr (K + 1 < MB) {
      trail(Kc, asm_ic<K + 1>{}, asm_ic<K + 1>{}, P);      // the next diagonal tile first ...
#pragma unroll
      for (int r = 0; r < 4; ++r) dt[li * 17 + N::kr(lq, r)] = -C[asm_tix(K + 1, K + 1)][r];
      ASM_FENCE();
      __builtin_amdgcn_sched_barrier(0);
#ifdef ASM_REG_ILV
      // ... then its factorisation INTERLEAVED with the rest of the trailing update: pivot step cc of the chain and the cc-th
      // sixteenth of the remaining tiles share a scheduling region (the MFMAs fill the chain's dependency stalls -- with one wave
      // per SIMD, the fp64 instance, nothing else does)
      {
        constexpr int NR = (MB - K - 1) * (MB - K) / 2 - 1;  // tiles of the trailing update besides (K + 1, K + 1)
        T x[16];
        asm_diag16_begin<T>(x, dt, idt, lane);
        asm_sfor<0, 16>([&](auto cc) __attribute__((always_inline)) {
          constexpr int c = decltype(cc)::value;
          asm_diag16_step<T, c>(x);
          asm_sfor<c * NR / 16, (c + 1) * NR / 16>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value + 1;       // linear index in the trailing block (0 = its first tile, done above)
            constexpr int J = asm_tri_row(MB - K - 1, t), I = asm_tri_col(MB - K - 1, t);
            trail(Kc, asm_ic<K + 1 + J>{}, asm_ic<K + 1 + I>{}, P);
          });
          __builtin_amdgcn_sched_barrier(0);
        });
        ASM_STAMP(6 + 3 * K);
        bad |= asm_diag16_end<T>(x, Yt, lane);
        ASM_FENCE();
      }
#else
      // ... then its factorisation together with the rest of the trailing update
      asm_sfor<K + 1, MB>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        asm_sfor<J, MB>([&](auto Ic) {
          constexpr int I = decltype(Ic)::value;
          if constexpr (!(I == K + 1 && J == K + 1)) trail(Kc, Jc, Ic, P);
        });
      });
      ASM_STAMP(6 + 3 * K);                                // (trailing update issued)
      bad |= asm_diag16<T>(dt, Yt, lane, idt);
      ASM_FENCE();
#endif
    }
  });
  ASM_STAMP(40);
  if (bad) return 1;
  ASM_FENCE();
  // ---- backward substitution  L' lam = y
  auto backward = [&](T (&out)[MB]) __attribute__((always_inline)) {
#pragma unroll
    for (int K = MB - 1; K >= 0; --K) {
      T part = T(0);
      T s4[4] = {T(0), T(0), T(0), T(0)};                  // (sum_I L(I,K)' lam_I)[kr(lq, r)], summed over li
      if (K < MB - 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int I = K + 1; I < MB; ++I) s4[r] += (K < NL ? lt[slot(I, K) * 256 + r * 64] : C[asm_tix(I, K)][r]) * out[I];
        rowsum16x4<T>(s4);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) part += C[asm_tix(K, K)][r] * (ys[16 * K + N::kr(lq, r)] - s4[r]);   // Y_K[kr(lq, r)][li] tt[kr(lq, r)]
      out[K] = xsum4<T>(part);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  backward(lam);
  if constexpr (REF) {
    if (!do_refine) return 0;
    // ---- one fp64 correction of a solve whose multipliers all keep their bounds
    int ai[MB], sa[MB];
    double bb[MB], bnd[MB];
    bool wrong = false;
#pragma unroll
    for (int I = 0; I < MB; ++I) ai[I] = rf.idx[min(16 * I + li, m - 1)];        // (three rounds of unconditional loads, then the selects)
#pragma unroll
    for (int I = 0; I < MB; ++I) { sa[I] = rf.st[ai[I]]; bb[I] = rf.xunc[ai[I]]; }
#pragma unroll
    for (int I = 0; I < MB; ++I) { const int k = ai[I] % d.nu; bnd[I] = sa[I] == 1 ? rf.ub[k] : rf.lb[k]; }
#pragma unroll
    for (int I = 0; I < MB; ++I) {
      const bool in = 16 * I + li < m;
      bb[I] = in ? bb[I] - bnd[I] : 0.0;
      const double l = (double)lam[I];
      wrong |= in && ((sa[I] == 1 && l <= 0.0) || (sa[I] == 2 && l >= 0.0));   // asm_update_k's test: the set moves anyway
    }
#ifndef ASM_REFINE_ALWAYS               // (scripts/micro/lambda_micro.hip: synthetic sets, random signs)
    if (__any(wrong)) return 0;
#endif
    int* ixr = reinterpret_cast<int*>(const_cast<T*>(rv));  // [16 MB] (rv's last reader was the forward substitution above)
    double* l64 = reinterpret_cast<double*>(dt);            // [16 MB] over dt / Yt (2 * ASM_TS floats: 16 MB <= 272)
    static_assert(16 * MB * 8 <= 2 * ASM_TS * (int)sizeof(T) || !N::F32, "lam64 does not fit the diagonal-tile buffers");
    unsigned gco[MB];
#pragma unroll
    for (int I = 0; I < MB; ++I) {
      gco[I] = (unsigned)ai[I] * 8u;
      if (lq == 0) { ixr[16 * I + li] = ai[I]; l64[16 * I + li] = 16 * I + li < m ? (double)lam[I] : 0.0; }
    }
    ASM_FENCE();
    const char* const H64 = reinterpret_cast<const char*>(d.H);
    double acc[MB];
#pragma unroll
    for (int I = 0; I < MB; ++I) acc[I] = 0.0;
    {
      // column j = 4 jj + lq of S (entries beyond m: index repeated, lam = 0); the MB loads of TWO columns in flight (left to the
      // compiler -- 256 registers, the tiles live -- every load was followed by s_waitcnt vmcnt(0))
      double hv[2][MB], lj[2];
      auto issue = [&](int jj, auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        const int j = 4 * jj + lq;
        const unsigned rowoff = (unsigned)ixr[j] * (unsigned)d.np * 8u;
        lj[b] = l64[j];
#pragma unroll
        for (int I = 0; I < MB; ++I) hv[b][I] = *reinterpret_cast<const double*>(H64 + (rowoff + gco[I]));
      };
      issue(0, asm_ic<0>{});
#pragma unroll 1
      for (int jj = 0; jj < 4 * MB; jj += 2) {
        __builtin_amdgcn_sched_barrier(0);
        issue(jj + 1, asm_ic<1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int I = 0; I < MB; ++I) acc[I] = fma(hv[0][I], lj[0], acc[I]);
        __builtin_amdgcn_sched_barrier(0);
        issue(min(jj + 2, 4 * MB - 1), asm_ic<0>{});         // (the last trip loads a column again: no branch in the loop)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int I = 0; I < MB; ++I) acc[I] = fma(hv[1][I], lj[1], acc[I]);
      }
    }
    T rr[MB];
#pragma unroll
    for (int I = 0; I < MB; ++I) rr[I] = 16 * I + li < m ? (T)(bb[I] - xsum4<double>(acc[I])) : T(0);
    // forward substitution  L y = r  (the factorisation's, without the MFMAs)
#pragma unroll
    for (int I = 0; I < MB; ++I) ps[I] = T(0);
    asm_sfor<0, MB>([&](auto Kc) __attribute__((always_inline)) {
      constexpr int K = decltype(Kc)::value;
      const V4 Yc = C[asm_tix(K, K)];
      const T tK = rr[K] - (K ? xsum4<T>(ps[K]) : T(0));
      T yq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) yq[r] = Yc[r] * tK;
      rowsum16x4<T>(yq);
      if (li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ys[16 * K + N::kr(lq, r)] = yq[r];
      }
#pragma unroll
      for (int I = K + 1; I < MB; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) ps[I] += (K < NL ? lt[slot(I, K) * 256 + r * 64] : C[asm_tix(I, K)][r]) * yq[r];
      __builtin_amdgcn_sched_barrier(0);
    });
    ASM_FENCE();
    T dl[MB];
    backward(dl);
#pragma unroll
    for (int I = 0; I < MB; ++I) rf.lam64[I] = (double)lam[I] + (double)dl[I];
    return 2;
  }
  return 0;
}
template <class T, int MB, int NL = asm_nl<T>(MB), bool ANYM = false>
__device__ __forceinline__ int asm_reg_core(const AsmDev& d, int m, const int* ix, T* dt, T* Yt, T* ys, const T* rv, T* lt, const T* idt,
                                            T (&lam)[MB], int lane, int wg, int wave) {
  AsmRefine<MB> none;
  return asm_reg_core<T, MB, NL, ANYM, false>(d, m, ix, dt, Yt, ys, rv, lt, idt, lam, lane, wg, wave, none, false);
}

template <class T, int MB, int WPB>
__device__ __forceinline__ void asm_lambda_reg(const AsmDev& d, int list, int wg) {
  using N = AsmNum<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  T* dt = reinterpret_cast<T*>(sm_raw) + (size_t)wave * asm_rw<T>(MB);   // diagonal tile, row-major stride 17
  T* Yt = dt + ASM_TS;                                     // its inverse factor
  T* ys = Yt + ASM_TS;                                     // y (forward result), [MB][16]
  T* rv = ys + MB * 16;                                    // right-hand side, [MB][16]
  T* lt = rv + MB * 16 + lane;                             // LDS-resident tiles: slot * 256 + r * 64 (+ lane)
  T* idt = reinterpret_cast<T*>(sm_raw) + (size_t)WPB * asm_rw<T>(MB);   // identity tile (every wave writes all of it: same
  for (int i = lane; i < ASM_TS; i += 64) idt[i] = (i / 17 == i % 17) ? T(1) : T(0);   // values, so no barrier is needed)
  const int nitem = d.counters[asm_list_counter(list)];
  const int it = wg * WPB + wave;
  if (it >= nitem) return;
  const int p = __builtin_amdgcn_readfirstlane(d.binlist[(size_t)list * d.nseg + it]);
  const size_t o = (size_t)p * d.np;
  const unsigned char* st = d.st + (size_t)p * d.n;
  const int* idx = d.idxg + (size_t)p * d.max_active;
  const int m = __builtin_amdgcn_readfirstlane(d.mg[p]);
  if (m <= 0) return;
  ASM_STAMP(0);
  // ---- active indices and rhs -> LDS (read back block by block: registers are the scarce resource here).  The index
  // list goes through LDS because every later use of an index is the ADDRESS of a gather: read from global memory it
  // would sit in the same in-order vmcnt queue as the gathers before it, and each block column of the gather would
  // wait for all the loads of the previous one (measured: 26 k of the wave's 85 k cycles went into ISSUING the gather).
  int* ix = reinterpret_cast<int*>(ys);                    // [MB * 16] (ys is not written before the factorisation)
  for (int i = lane; i < MB * 16; i += 64) {
    const int a = idx[min(i, m - 1)], k = a % d.nu;
    ix[i] = a;
    const double v = d.xunc[o + a] - (st[a] == 1 ? d.ub[(size_t)p * d.nu + k] : d.lb[(size_t)p * d.nu + k]);
    rv[i] = i < m ? (T)v : T(0);
  }
  ASM_FENCE();
  T lam[MB];
  // f32 rounds with a row of LAM (fp64): asm_bins hands those out when d.refine is set -- a result that can be final gets its fp64 correction
  AsmRefine<MB> rf;
  const bool refine = N::F32 && d.rowk[p] == 0;
  if (N::F32) { rf.xunc = d.xunc + o; rf.lb = d.lb + (size_t)p * d.nu; rf.ub = d.ub + (size_t)p * d.nu; rf.st = st; rf.idx = idx; }
  const int bad = asm_reg_core<T, MB, asm_nl<T>(MB), false, N::F32>(d, m, ix, dt, Yt, ys, rv, lt, idt, lam, lane, wg, wave, rf, refine);
  if (N::F32 && refine && bad != 1) {
    if (bad == 2 && lane == 0) d.redo[p] = 2;               // corrected in fp64: asm_update_k verifies the residual and may accept the set
    double* lrow64 = d.lam + (size_t)d.row[p] * d.np;
#pragma unroll
    for (int I = 0; I < MB; ++I) {
      const int i = 16 * I + li;
      if (lq == 0 && i < m) lrow64[idx[i]] = bad == 2 ? rf.lam64[I] : (double)lam[I];
    }
    return;
  }
  if (bad) {
    // f32: S is not positive definite in this precision -- this round is void (asm_update_k skips the problem,
    // its LAM row is still zero), the next one runs in fp64.  fp64: hand the problem to the PDIP path.
    if (lane == 0) { if (N::F32) { d.prec[p] = 1; d.redo[p] = 1; } else d.state[p] = ASM_FALLBACK; }
    return;
  }
  if (!N::F32 && lane == 0) d.prec[p] = 1;                 // solved in fp64 (sets beyond the f32 classes start here)
  ASM_STAMP(41);
  using LT = typename std::conditional<N::F32, float, double>::type;   // f32 rounds: row of LAM32 (f32 GEMM)
  LT* lrow = (N::F32 ? (LT*)d.lam32 : (LT*)d.lam) + (size_t)d.row[p] * d.np;
#pragma unroll
  for (int I = 0; I < MB; ++I) {
    const int i = 16 * I + li;
    if (lq == 0 && i < m) lrow[idx[i]] = (LT)lam[I];
  }
}

// ---- f32 rounds of the sets beyond the register kernels (ASM_MLDS + 1 .. ASM_BIG32 bounds): the same blocked Cholesky as
// asm_tile_solve, one workgroup (eight waves) per problem, in f32 with ALL tiles in LDS (136 tiles of 16 x 17 floats at 256
// bounds: 148 KB) -- the eight waves share the TRSM and trailing MFMAs (v_mfma_f32_16x16x4_f32) of a block column.  As in the
// register kernels an f32 result only moves the set; a set that settles is solved again in fp64 (asm_lambda_tile_k<1>).
typedef float f32x4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v_t tile_mma_nt32(const float* A, const float* B, int lane) {
  f32x4v_t acc = {0.f, 0.f, 0.f, 0.f};
  const int li = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[li * 17 + 4 * s + kq], B[li * 17 + 4 * s + kq], acc, 0, 0, 0);
  return acc;                                            // reg r of lane (li, lq): row 4 lq + r, column li
}
__device__ __forceinline__ float* asm_tile32(float* T, int I, int J) { return T + ((size_t)I * (I + 1) / 2 + J) * ASM_TS; }

// rA [>= 16 ceil(m/16)], Yt [ASM_TS], T [tiles] in LDS; returns 1 (to all threads) when H_AA is not positive definite in f32
__device__ __forceinline__ int asm_tile_solve32(const AsmDev& d, int p, int m, const int* idx, float* rA, float* Yt, float* T, int* s_bad) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef ASM_STAMPS
  const int wg = blockIdx.x;
#endif
  __syncthreads();
  ASM_STAMP(48);
  if (tid == 0) *s_bad = 0;
  const size_t o = (size_t)p * d.np;
  const unsigned char* st = d.st + (size_t)p * d.n;
  const int mb = (m + 15) / 16;
  __syncthreads();
  for (int i = tid; i < mb * 16; i += 512) {
    float v = 0.f;
    if (i < m) {
      const int a = idx[i], k = a % d.nu;
      v = (float)(d.xunc[o + a] - (st[a] == 1 ? d.ub[(size_t)p * d.nu + k] : d.lb[(size_t)p * d.nu + k]));
    }
    rA[i] = v;
  }
  const int ntile = mb * (mb + 1) / 2;
  {
    // thread (half, ti, tj) fetches element (ti, tj) of every second tile, 8 at a time: 8 gathers in flight per thread
    // (the entries come from L2 / Infinity Cache, ~1 us each: the gather is a latency chain, not a bandwidth one)
    const int half = tid >> 8, ti = (tid >> 4) & 15, tj = tid & 15;
    for (int t0 = half; t0 < ntile; t0 += 2 * 8) {
      float v[8];
      int tt[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + 2 * u;
        // tile t -> (I, J): I = largest with I (I + 1) / 2 <= t
        int I = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        while (I * (I + 1) / 2 > t) --I;
        const int J = t - I * (I + 1) / 2;
        tt[u] = t;
        const int gi = 16 * I + ti, gj = 16 * J + tj;
        v[u] = gi == gj ? 1.f : 0.f;
        if (t < ntile && gi < m && gj < m) v[u] = d.H32[(size_t)idx[gi] * d.np + idx[gj]];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (tt[u] < ntile) (T + (size_t)tt[u] * ASM_TS)[ti * 17 + tj] = v[u];
    }
  }
  __syncthreads();
  // Two barriers per block column: [TRSM of column K by all waves] | [wave 0: its tile (K+1, K+1) of the trailing update,
  // Y_K' saved for the solves, then the NEXT diagonal step -- the long dependent chain -- while waves 1..7 share the rest
  // of the trailing update].  (The first version ran diagonal step, TRSM and trailing update one after the other behind
  // four barriers: the diagonal steps alone, seven waves waiting, were 45 % of the factorisation.)
  ASM_STAMP(49);
  if (wave == 0) {
    const int bad = asm_diag16<float>(asm_tile32(T, 0, 0), Yt, lane);
    if (bad && lane == 0) *s_bad = 1;
  }
  __syncthreads();
  ASM_STAMP(50);
  for (int K = 0; K < mb; ++K) {
    float* TKK = asm_tile32(T, K, K);
    for (int I = K + 1 + wave; I < mb; I += 8) {         // TRSM: T(I,K) <- T(I,K) Y'   (eight waves)
      float* TIK = asm_tile32(T, I, K);
      const f32x4v_t acc = tile_mma_nt32(TIK, Yt, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) TIK[(4 * (lane >> 4) + r) * 17 + (lane & 15)] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
      if (lane < 16) {                                     // Y_K' into the upper part of T(K,K) for the solves
        const int row = lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k >= row) TKK[row * 17 + k] = Yt[k * 17 + row];
      }
      if (K + 1 < mb) {
        const f32x4v_t acc = tile_mma_nt32(asm_tile32(T, K + 1, K), asm_tile32(T, K + 1, K), lane);
        float* Tn = asm_tile32(T, K + 1, K + 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) Tn[(4 * (lane >> 4) + r) * 17 + (lane & 15)] -= acc[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile and the saved Y' are in LDS before the sweep reads / overwrites
        const int bad = asm_diag16<float>(Tn, Yt, lane);
        if (bad && lane == 0) *s_bad = 1;
      }
    } else {
      int b = 0;
      for (int I = K + 1; I < mb; ++I)
        for (int J = K + 1; J <= I; ++J) {
          if (I == K + 1) continue;                        // (K+1, K+1): wave 0
          if ((b++ % 7) + 1 != wave) continue;
          const f32x4v_t acc = tile_mma_nt32(asm_tile32(T, I, K), asm_tile32(T, J, K), lane);
          float* TIJ = asm_tile32(T, I, J);
#pragma unroll
          for (int r = 0; r < 4; ++r) TIJ[(4 * (lane >> 4) + r) * 17 + (lane & 15)] -= acc[r];
        }
    }
    __syncthreads();
  }
  ASM_STAMP(51);
  if (*s_bad) return 1;
  // ---- substitutions, column oriented, all eight waves (one wave alone spent 29 % of the kernel here): as soon as y_K
  // (lam_K) is known every wave takes it out of the right-hand sides of its share of the remaining block rows.
  {
    const int i = lane & 15, kq = lane >> 4;
    for (int K = 0; K < mb; ++K) {                       // forward:  L y = r
      if (wave == 0) {                                   // y_K = Y_K t_K,  Y_K[i][k] (k <= i) stored at TKK[k][i]
        const float* TKK = asm_tile32(T, K, K);
        const float t = rA[16 * K + i];
        float yv = t * TKK[i * 17 + i];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float tk = __shfl(t, k); if (k < i) yv += TKK[k * 17 + i] * tk; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16) rA[16 * K + i] = yv;
      }
      __syncthreads();
      for (int I = K + 1 + wave; I < mb; I += 8) {       // r_I -= L(I,K) y_K
        const float* TIK = asm_tile32(T, I, K);
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) t += TIK[i * 17 + 4 * kq + k] * rA[16 * K + 4 * kq + k];
        t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
        if (lane < 16) rA[16 * I + i] -= t;
      }
      __syncthreads();
    }
    for (int K = mb - 1; K >= 0; --K) {                  // backward:  L' lam = y
      if (wave == 0) {                                   // lam_K = Y_K' t_K,  Y_K[k][i] (k > i) stored at TKK[i][k]
        const float* TKK = asm_tile32(T, K, K);
        const float t = rA[16 * K + i];
        float lv = t * TKK[i * 17 + i];
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float tk = __shfl(t, k); if (k > i) lv += TKK[i * 17 + k] * tk; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16) rA[16 * K + i] = lv;
      }
      __syncthreads();
      for (int J = K - 1 - wave; J >= 0; J -= 8) {       // y_J -= L(K,J)' lam_K
        const float* TKJ = asm_tile32(T, K, J);
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) t += TKJ[(4 * kq + k) * 17 + i] * rA[16 * K + 4 * kq + k];
        t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
        if (lane < 16) rA[16 * J + i] -= t;
      }
      __syncthreads();
    }
  }
  ASM_STAMP(52);
  __syncthreads();
  return 0;
}

constexpr int ASM_TILE32_LDS = (ASM_BIG32 + ASM_TS + (ASM_BIG32 / 16) * (ASM_BIG32 / 16 + 1) / 2 * ASM_TS) * 4;
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(512, 1) void asm_lambda_tile32_k(AsmDev d) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  float* rA = smf;                                       // [ASM_BIG32]
  float* Yt = rA + ASM_BIG32;
  float* T = Yt + ASM_TS;
  const int nitem = d.counters[ASM_CNT_BIG32];
  const int* list = d.binlist + (size_t)ASM_NLIST * d.nseg;
  for (int it = blockIdx.x; it < nitem; it += gridDim.x) {
    const int p = list[it];
    const int* idx = d.idxg + (size_t)p * d.max_active;
    const int m = d.mg[p];
    if (asm_tile_solve32(d, p, m, idx, rA, Yt, T, &s_bad)) {
      // not positive definite in f32: the round is void (the LAM32 row is still zero), the next one runs in fp64
      if (tid == 0) { d.prec[p] = 1; d.redo[p] = 1; }
      continue;
    }
    float* lrow = d.lam32 + (size_t)d.row[p] * d.np;
    for (int i = tid; i < m; i += 512) lrow[idx[i]] = rA[i];
  }
}
#endif

// All register-resident fp64 size classes 0..5 in ONE launch (their workgroups are independent; separate
// launches would serialise six tails): workgroup w walks the classes from the largest down and takes
// four problems of the class its index falls into.  Grid: sum_b ceil(count_b / 4).
constexpr int ASM_REG_LDS = (4 * asm_rw<double>(ASM_NREG + 3) + ASM_TS) * 8;  // bytes of dynamic LDS (largest class) + the identity tile
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256, 1) void asm_lambda_reg_k(AsmDev d) {
  int w = blockIdx.x;
#define ASM_REG_CLASS(B)                                                   \
  {                                                                        \
    const int nb = (d.counters[4 + B] + 3) >> 2;                           \
    if (w < nb) { asm_lambda_reg<double, B + 4, 4>(d, B, w); return; }     \
    w -= nb;                                                               \
  }
  ASM_REG_CLASS(5) ASM_REG_CLASS(4) ASM_REG_CLASS(3) ASM_REG_CLASS(2) ASM_REG_CLASS(1) ASM_REG_CLASS(0)
#undef ASM_REG_CLASS
  static_assert(ASM_NREG == 6, "one ASM_REG_CLASS line per register-resident size class");
}
#endif
// The 10- and 11-block classes (145..176 bounds): same code, two waves (problems) per workgroup.
constexpr int ASM_REG2_LDS = (2 * asm_rw<double>(11) + ASM_TS) * 8;
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(128, 1) void asm_lambda_reg2_k(AsmDev d) {
  int w = blockIdx.x;
  {
    const int nb = (d.counters[4 + 7] + 1) >> 1;
    if (w < nb) { asm_lambda_reg<double, 11, 2>(d, 7, w); return; }
    w -= nb;
  }
  asm_lambda_reg<double, 10, 2>(d, 6, w);
  static_assert(ASM_NBIN == 8 && ASM_MLDS == 176, "classes 6 and 7 are the 10- and 11-block sets");
}
#endif
// The f32 rounds of classes 0..5: two workgroups per CU (two waves per SIMD).
constexpr int ASM_REG32_LDS = (4 * asm_rw<float>(ASM_NREG + 3) + ASM_TS) * 4;
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256, 2) void asm_lambda_reg32_k(AsmDev d) {
  int w = blockIdx.x;
#define ASM_REG_CLASS(B)                                                            \
  {                                                                                 \
    const int nb = (d.counters[ASM_CNT_F32 + B] + 3) >> 2;                          \
    if (w < nb) { asm_lambda_reg<float, B + 4, 4>(d, ASM_NBIN + B, w); return; }    \
    w -= nb;                                                                        \
  }
  ASM_REG_CLASS(5) ASM_REG_CLASS(4) ASM_REG_CLASS(3) ASM_REG_CLASS(2) ASM_REG_CLASS(1) ASM_REG_CLASS(0)
#undef ASM_REG_CLASS
}
#endif

// ... and of classes 6, 7: four waves per workgroup, one workgroup per CU (LDS).
constexpr int ASM_REG32B_LDS = (4 * asm_rw<float>(11) + ASM_TS) * 4;
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256, 1) void asm_lambda_reg32b_k(AsmDev d) {
  int w = blockIdx.x;
  {
    const int nb = (d.counters[ASM_CNT_F32 + 7] + 3) >> 2;
    if (w < nb) { asm_lambda_reg<float, 11, 4>(d, ASM_NBIN + 7, w); return; }
    w -= nb;
  }
  asm_lambda_reg<float, 10, 4>(d, ASM_NBIN + 6, w);
}
#endif

}  // namespace nnmpc
#ifndef ASM_NO_WG_KERNELS             // (scripts/micro/predict_micro.hip: the workgroup kernels are minutes of compile time)
#include "qp_wg.h"
#endif
namespace nnmpc {

// x from the GEMM result, fp64 KKT tests, next active set.  A problem whose set no longer changes
// is certified right here when the verified inverse allows it:  with E1 = P Kunc + tq and
// E2 = P Pinv - I (maxima e1max, e2max computed once at setup),
//     P x + q = E1 x0 - lam_ext - E2 lam_ext,
// so |stationarity residual on the free set| <= e1max |x0|_1 + e2max |lam|_1 =: bnd, and the
// multiplier signs are certain when every |lam_a| > bnd.  Otherwise (ASM_DONE) the full check
// with P itself (asm_certify_k) decides.
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256) void asm_update_k(AsmDev d) {   // one WAVE per problem: the window is a few hundred columns
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= d.nseg || d.state[p] != ASM_RUN) return;
  const int rdo = d.redo[p];
  if (rdo == 1) {                                            // the f32 kernel gave up on this set: nothing was computed,
    if (lane == 0) d.redo[p] = 0;                            // the same set runs again in fp64
    return;
  }
  if (rdo && lane == 0) d.redo[p] = 0;                       // 2: an f32 solve with its fp64 correction (asm_reg_core): verified below
  const bool f32_phase = d.rowk[p] != 0;                     // f32 solve and GEMM: only good enough to move the set
  const size_t o = (size_t)p * d.np, orow = (size_t)d.row[p] * d.np;
  unsigned char* st = d.st + (size_t)p * d.n;
  const int W = min(d.W, d.n);                               // every active bound lies inside the window
  const int m = d.mg[p];                                     // this round's set: idx[0..m) (asm_count_k)
  const int* idx = d.idxg + (size_t)p * d.max_active;
  // Exchange rule (block principal pivoting with Murty's fallback: Judice & Pires 1994, Kim & Park 2011): all
  // infeasible indices change sides as long as their number keeps falling -- with ASM_GRACE rounds of grace --,
  // otherwise only the infeasible variable with the SMALLEST index does (Murty's least-index rule: the earliest MPC
  // stage first -- taking the largest index instead needed several times as many exchanges on the stragglers of the
  // cond-4e7 plant).  The plain all-at-once rule cycles on
  // ill-conditioned Hessians (6 of 131 072 samples of the cond-4e7 CSTRs-size plant); the fallback is finite.
  __shared__ unsigned short ch_r[4][8][64];                  // this lane's first changes: index, new state
  __shared__ unsigned char ch_s[4][8][64];
  const int wv = threadIdx.x >> 6;
  int chg = 0, rmax = 0x7fffffff;                            // the index a single exchange takes: the smallest infeasible one
  double l1 = 0.0, lmin = 1e300;
  double res_r = 0.0, res_b = 0.0;                           // max |b - S lam| and max |b| over the set (rdo == 2: from the GEMM's row)
  auto scan = [&](int mode) {                                // 0: count and record, 1: apply all, 2: apply only index rmax (the smallest infeasible one)
    // (loading eight chunks of the window at once, unconditionally, before the tests -- instead of bound state, branch, then x_unc
    // and the GEMM's row per chunk -- made this kernel slower: 0.30 -> 0.53 ms per round at 100 000 problems)
    for (int r = lane; r < W; r += 64) {                     // free variables of the window: feasibility
      if (st[r]) continue;
      const int k = r % d.nu;
      const double lb = d.lb[(size_t)p * d.nu + k], ub = d.ub[(size_t)p * d.nu + k];
      const double x = d.xunc[o + r] - (f32_phase ? (double)d.xh32[orow + r] : d.xh[orow + r]);
      const int ns = x > ub + d.bound_tol ? 1 : (x < lb - d.bound_tol ? 2 : 0);
      if (!ns) continue;
      if (mode == 0) { if (chg < 8) { ch_r[wv][chg][lane] = (unsigned short)r; ch_s[wv][chg][lane] = (unsigned char)ns; } ++chg; rmax = min(rmax, r); }
      else if (mode == 1 || r == rmax) st[r] = (unsigned char)ns;
    }
    for (int i = lane; i < m; i += 64) {                     // active bounds: multiplier signs (keep iff multiplier > 0)
      const int a = idx[i], sa = st[a];
      const double l = f32_phase ? (double)d.lam32[orow + a] : d.lam[orow + a];
      if (mode == 0) { l1 += fabs(l); lmin = fmin(lmin, fabs(l)); }
      if (mode == 0 && rdo == 2) {
        const int k = a % d.nu;
        const double b = d.xunc[o + a] - (sa == 1 ? d.ub[(size_t)p * d.nu + k] : d.lb[(size_t)p * d.nu + k]);
        res_b = fmax(res_b, fabs(b));
        res_r = fmax(res_r, fabs(b - d.xh[orow + a]));
      }
      if (!((sa == 1 && l <= 0.0) || (sa == 2 && l >= 0.0))) continue;
      if (mode == 0) { if (chg < 8) { ch_r[wv][chg][lane] = (unsigned short)a; ch_s[wv][chg][lane] = 0; } ++chg; rmax = min(rmax, a); }
      else if (mode == 1 || a == rmax) st[a] = 0;
    }
  };
  scan(0);
  const int mych = chg;
  const bool overflow = __any(mych > 8) || d.n > 65535;
  for (int off = 32; off > 0; off >>= 1) { chg += __shfl_xor(chg, off); rmax = min(rmax, __shfl_xor(rmax, off)); }
  const int tot = chg;
  // fp64-grade multipliers: an fp64 solve, or a corrected f32 solve whose residual -- S lam is in this round's GEMM row -- is at the
  // fp64 floor (the fp64 kernel leaves ~5e-15 |b| at these sizes)
  bool grade64 = !f32_phase && d.prec[p] != 0;
  if (rdo == 2) {
    for (int off = 32; off > 0; off >>= 1) { res_r = fmax(res_r, __shfl_xor(res_r, off)); res_b = fmax(res_b, __shfl_xor(res_b, off)); }
    grade64 = !f32_phase && res_r <= d.refine_tol * res_b;
  }
  if (tot > 0) {
    int single = 0;
    if (lane == 0) {
      const int best = d.ninf_best[p];
      int alpha = d.alpha[p];
      if (tot < best) { d.ninf_best[p] = tot; alpha = ASM_GRACE; }
      else if (alpha > 0) --alpha;
      else { single = 1; d.prec[p] = 1; }                    // the careful rule works on fp64 multipliers only
      d.alpha[p] = (unsigned char)alpha;
    }
    single = __shfl(single, 0);
    if (overflow) scan(single ? 2 : 1);                      // rare (window = all columns): decide again, with the writes
    else {
      for (int i = 0; i < mych; ++i) {
        const int r = ch_r[wv][i][lane];
        if (!single || r == rmax) st[r] = ch_s[wv][i][lane];
      }
    }
  }
  const bool settled = tot == 0 && grade64;                  // a set that settles in f32 is solved again in fp64
  const bool settle_wide = settled && W < d.n;               // settled inside the window: full-width check next
  // the LAM row goes back to zero (rows are handed out anew every round) -- except the row of a problem that settled inside the
  // window: the full-width pass at the start of the next round runs on it as it stands (asm_wide_k clears it afterwards)
  if (settle_wide) { if (lane == 0) d.rowprob[d.row[p]] = p; }
  else {
    for (int i = lane; i < m; i += 64) {
      const int a = idx[i];
      if (f32_phase) d.lam32[orow + a] = 0.f;
      else d.lam[orow + a] = 0.0;
    }
  }
  bool sure = false;
  if (settled && !settle_wide) {                             // finished (window = all columns): certificate, and only
    double x1 = 0.0;                                         // now x is written out, straight into the caller's buffer
    for (int k = lane; k < d.ka; k += 64) x1 += fabs(d.x0[(size_t)p * d.ka + k]);
    for (int off = 32; off > 0; off >>= 1) {
      l1 += __shfl_xor(l1, off); x1 += __shfl_xor(x1, off);
      lmin = fmin(lmin, __shfl_xor(lmin, off));
    }
    const double QI = d.tqmax * x1;                          // >= |q|_inf
    const double bnd = 2.0 * (d.e1max * x1 + d.e2max * l1) + 1e-14 * (QI + l1);
    sure = bnd <= d.stat_tol * d.pscale && lmin > bnd;
    for (int r = lane; r < d.n; r += 64) {
      const int k = r % d.nu, s = st[r];
      const double x = s == 0 ? d.xunc[o + r] - d.xh[orow + r]
                              : (s == 1 ? d.ub[(size_t)p * d.nu + k] : d.lb[(size_t)p * d.nu + k]);
      if (r < d.nout) d.u_out[(size_t)p * d.ldu + r] = x;
      if (!sure) d.x[o + r] = x;                             // as a GEMM row too if P itself must confirm it
    }
  }
  if (lane == 0) {
    const int rd = d.rounds[p] + 1;
    d.rounds[p] = rd;
    if (tot > 0 && d.hi[p] < W) d.hi[p] = W;                 // bounds inside the window may have joined
    const bool refine_next = d.refine_later && m <= ASM_MLDS;   // (the classes asm_lambda_reg serves: larger sets keep the f32 -> fp64 ladder)
    if (tot == 0 && !grade64 && !(f32_phase && refine_next)) d.prec[p] = 1;   // (refine_later: the next round solves it in f32 with the fp64 correction)
    // ... and so is one that is about to: when at most early64 bounds moved, the next set is most often the final one, and
    // solving it in fp64 at once saves the f32 round that would only have confirmed it (CDU batch: 4.65 -> 3.7 f32 solves
    // per problem, still one fp64 solve for nine in ten)
    if (tot > 0 && tot <= d.early64 && f32_phase && !refine_next) d.prec[p] = 1;
    if (settle_wide) d.state[p] = ASM_WIDE;
    else if (settled) {
      d.state[p] = sure ? ASM_CERT : ASM_DONE;
      if (!sure) atomicAdd(&d.counters[ASM_CNT_DONE], 1);    // rare: q and x P are formed only for these
    } else if (rd >= d.max_rounds) d.state[p] = ASM_FALLBACK;
  }
}
#endif

// Full-width check of the problems that settled inside the window (xhw = lamw * H over all columns):
// a bound violated beyond the window joins the set and the problem runs on; otherwise it is finished
// exactly like in asm_update_k.  Runs at the start of a round, before asm_count_k.
// One WAVE per row of LAM (four per workgroup; 256 threads and three barriers per problem were launch and barrier latency:
// 0.44 ms for the 83 000 problems of the round most sets settle in).
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256) void asm_wide_k(AsmDev d, int fused_c0) {
  // fused_c0 >= 0: the columns [fused_c0, n) were checked and written by asm_wide_gemm_k (qp_wide.h), which raised
  // wflag[p] for a violated bound; this kernel does the columns inside the window, the multiplier statistics and the
  // decision.  fused_c0 < 0: everything from the XHW rows of a plain GEMM (shapes the fused kernel's tiles do not fit).
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);         // row of LAM / XH (and XHW) of the round the problem settled in
  if (w >= d.wrows) return;
  const int p = d.rowprob[w];
  if (p < 0) return;
  const size_t o = (size_t)p * d.np, orow = (size_t)w * d.np;
  // columns inside the window the problem settled in (d.W is still that round's) were evaluated by that round's
  // GEMM: its row of XH is untouched until this round's GEMM; only the columns beyond come from the wide pass
  const size_t onar = (size_t)d.row[p] * d.np;
  const int Wp = min(d.W, d.n);
  const bool fused = fused_c0 >= 0;
  const int rend = fused ? min(fused_c0, d.n) : d.n;         // columns this kernel evaluates
  unsigned char* st = d.st + (size_t)p * d.n;
  const int m = d.mg[p];                                     // the settled set: idx[0..m) (unchanged since asm_count_k)
  const int* idx = d.idxg + (size_t)p * d.max_active;
  int chg = 0;
  double l1 = 0.0, lmin = 1e300;
  for (int i = lane; i < m; i += 64) {                       // multipliers of the settled set (the row goes back to zero below)
    const double l = d.lam[orow + idx[i]];
    l1 += fabs(l); lmin = fmin(lmin, fabs(l));
  }
  for (int r = lane; r < rend; r += 64) {
    const int k = r % d.nu;
    const double lb = d.lb[(size_t)p * d.nu + k], ub = d.ub[(size_t)p * d.nu + k];
    const int s = st[r];
    double x;
    if (s == 0) {
      x = d.xunc[o + r] - (r < Wp ? d.xh[onar + r] : d.xhw[orow + r]);
      if (x > ub + d.bound_tol) { st[r] = 1; ++chg; }
      else if (x < lb - d.bound_tol) { st[r] = 2; ++chg; }
    } else x = s == 1 ? ub : lb;
    if (r < d.nout) d.u_out[(size_t)p * d.ldu + r] = x;      // final if nothing changes
  }
  if (fused && lane == 0) { chg += d.wflag[p]; d.wflag[p] = 0; }
  double x1 = 0.0;
  for (int k = lane; k < d.ka; k += 64) x1 += fabs(d.x0[(size_t)p * d.ka + k]);
  for (int off = 32; off > 0; off >>= 1) {
    chg += __shfl_xor(chg, off);
    l1 += __shfl_xor(l1, off); x1 += __shfl_xor(x1, off);
    lmin = fmin(lmin, __shfl_xor(lmin, off));
  }
  const int tot = chg;
  const double QI = d.tqmax * x1;                            // >= |q|_inf
  // (far-field pass: x beyond the window is off by at most max|U V' - M| (|x0|_1 + |lam|_1), its gradient by |P|_inf times that)
  const double bnd = 2.0 * (d.e1max * x1 + d.e2max * l1) + 1e-14 * (QI + l1) + d.ff_err * (x1 + l1);
  const bool sure = bnd <= d.stat_tol * d.pscale && lmin > bnd;
  if (tot == 0 && !sure) {                                   // P itself has to confirm this one: x as a GEMM row
    for (int r = lane; r < d.n; r += 64) {                   // (nothing changed: st still is the set x belongs to)
      const int k = r % d.nu, s = st[r];
      double x;
      if (s != 0) x = s == 1 ? d.ub[(size_t)p * d.nu + k] : d.lb[(size_t)p * d.nu + k];
      else if (r < Wp) x = d.xunc[o + r] - d.xh[onar + r];
      else if (!fused) x = d.xunc[o + r] - d.xhw[orow + r];
      else {                                                 // rare: the product again, straight from Pinv (as asm_tail_k does)
        const double* Hc = d.H + r;
        double a0 = 0.0;
        for (int i = 0; i < m; ++i) { const int a = idx[i]; a0 += Hc[(size_t)a * d.np] * d.lam[orow + a]; }
        double xu;
        if (r < d.Wx) xu = d.xunc[o + r];
        else {                                               // x_unc beyond the columns kept in HBM: Kunc[r] . x0
          xu = 0.0;
          for (int k2 = 0; k2 < d.ka; ++k2) xu += d.Kunc[(size_t)r * d.ka + k2] * d.x0[(size_t)p * d.ka + k2];
        }
        x = xu - a0;
      }
      d.x[o + r] = x;
    }
  }
  for (int i = lane; i < m; i += 64) d.lam[orow + idx[i]] = 0.0;   // the row of LAM goes back to zero
  if (lane == 0) {
    d.wmark[p] = 1;
    if (tot == 0) {
      d.state[p] = sure ? ASM_CERT : ASM_DONE;
      if (!sure) atomicAdd(&d.counters[ASM_CNT_DONE], 1);    // rare: q and x P are formed only for these
    } else {
      d.state[p] = d.rounds[p] >= d.max_rounds ? ASM_FALLBACK : ASM_RUN;
      d.ninf_best[p] = 0x7fffffff; d.alpha[p] = ASM_GRACE;          // new bounds joined: the exchange rule starts afresh
      d.hi[p] = d.n;                                                // ... anywhere
    }
  }
}
#endif

// ---- the tail: when only a few problems are still running, a round of eight launches and a host read-back is all
// latency.  asm_tail_k finishes them on the device: one workgroup per problem loops count -> fp64 solve (tiles in its
// L2 slab) -> x over ALL columns (n |A| MACs straight from Pinv) -> exchange rule, until the set settles or the budget
// is spent.  Same arithmetic and the same certificate as the round kernels; u goes to the caller's buffer.
// ---- dense factor of the tail's single-exchange phase (see asm_tail_k): L (lower, fp64) of S = H_AA for up to
// ASM_FM - 1 bounds in LDS, packed by rows: row i holds its columns 0..i + 1 (one spare slot: the entry a dropped
// row leaves above the diagonal until the Givens rotations have removed it) at offset i (i + 3) / 2.  The routines
// below run in ONE wave (DS operations of a wave complete in order; ASM_FENCE where a later read depends on an
// earlier write of another lane).
constexpr int ASM_FM = 160;
constexpr int ASM_TAIL_GRACE = 2;
constexpr int ASM_TAIL_AREA = ASM_FM * (ASM_FM + 3) / 2 > ASM_TAIL_MB * (ASM_TAIL_MB + 1) / 2 * ASM_TS
                                  ? ASM_FM * (ASM_FM + 3) / 2 : ASM_TAIL_MB * (ASM_TAIL_MB + 1) / 2 * ASM_TS;   // doubles: tiles or factor
constexpr int ASM_TAIL_EXTRA = ASM_FM * (4 + 8 + 8 + 8);  // al, rhs, two work vectors
__device__ __forceinline__ int asm_frow(int i) { return i * (i + 3) / 2; }

// v <- L^-1 v  (v[0..m) in LDS, m < ASM_FM <= 192).  The vector lives in three registers per lane (element lane + 64 u), the
// pivot element goes round by v_readlane, the reciprocals of the diagonal are formed once, in parallel: one LDS read and two
// FMAs per lane in the dependent chain of a step.  (The first version kept v in LDS -- four LDS round trips and an fp64
// division per step: 20 us per substitution at 100 bounds, two or three substitutions per iteration of asm_tail_k.)
static_assert(ASM_FM <= 192, "asm_fwd / asm_bwd hold the vector in at most three registers per lane");
// (NU = registers per lane: the loop is bound by instruction issue, ~25 instructions per step and register -- sets of at most 64
// bounds run the NU = 1 instance, at most 128 the NU = 2 one)
template <int NU>
__device__ __forceinline__ void asm_fwd_t(const double* Ld, double* v, int m, int lane) {
  double x[NU], idg[NU], ln[NU];
  const double* rowp[NU];                                    // row of element lane + 64 u (clamped to the last row: reads stay inside L)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = lane + 64 * u, ic = min(i, m - 1);
    rowp[u] = Ld + asm_frow(ic);
    x[u] = i < m ? v[i] : 0.0;
    idg[u] = i < m ? 1.0 / rowp[u][i] : 0.0;
    ln[u] = rowp[u][0];
  }
  // branch-free steps, the LDS reads of step k + 1 issued before the arithmetic of step k (entries at or above the
  // diagonal are read and not used)
  for (int k = 0; k < m; ++k) {
    const int ku = k >> 6, kl = k & 63;
    double lc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { lc[u] = ln[u]; ln[u] = rowp[u][min(k + 1, m - 1)]; }
    double t = x[0] * idg[0];
#pragma unroll
    for (int u = 1; u < NU; ++u) t = ku == u ? x[u] * idg[u] : t;
    const double yk = rdlane_d(t, kl);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = lane + 64 * u;
      const double upd = x[u] - lc[u] * yk;
      x[u] = i == k ? yk : ((i > k && i < m) ? upd : x[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) { const int i = lane + 64 * u; if (i < m) v[i] = x[u]; }
  ASM_FENCE();
}
__device__ __forceinline__ void asm_fwd(const double* Ld, double* v, int m, int lane) {
  if (m <= 64) asm_fwd_t<1>(Ld, v, m, lane);
  else if (m <= 128) asm_fwd_t<2>(Ld, v, m, lane);
  else asm_fwd_t<3>(Ld, v, m, lane);
}
// v <- L^-T v
template <int NU>
__device__ __forceinline__ void asm_bwd_t(const double* Ld, double* v, int m, int lane) {
  double x[NU], idg[NU], ln[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = lane + 64 * u;
    x[u] = i < m ? v[i] : 0.0;
    idg[u] = i < m ? 1.0 / Ld[asm_frow(i) + i] : 0.0;
    ln[u] = m > 0 ? Ld[asm_frow(m - 1) + min(i, m - 1)] : 0.0;
  }
  for (int k = m - 1; k >= 0; --k) {
    const int ku = k >> 6, kl = k & 63;
    double lc[NU];
    const double* rown = Ld + asm_frow(max(k - 1, 0));       // row k - 1 for the next step: columns 0..k-1 (k is its spare slot)
#pragma unroll
    for (int u = 0; u < NU; ++u) { lc[u] = ln[u]; ln[u] = rown[min(lane + 64 * u, max(k - 1, 0))]; }
    double t = x[0] * idg[0];
#pragma unroll
    for (int u = 1; u < NU; ++u) t = ku == u ? x[u] * idg[u] : t;
    const double lk = rdlane_d(t, kl);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = lane + 64 * u;
      const double upd = x[u] - lc[u] * lk;
      x[u] = i == k ? lk : (i < k ? upd : x[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) { const int i = lane + 64 * u; if (i < m) v[i] = x[u]; }
  ASM_FENCE();
}
__device__ __forceinline__ void asm_bwd(const double* Ld, double* v, int m, int lane) {
  if (m <= 64) asm_bwd_t<1>(Ld, v, m, lane);
  else if (m <= 128) asm_bwd_t<2>(Ld, v, m, lane);
  else asm_bwd_t<3>(Ld, v, m, lane);
}

#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256) void asm_taillist_k(AsmDev d) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p < d.nseg && d.state[p] == ASM_RUN) d.biglist[atomicAdd(&d.counters[ASM_CNT_TAIL], 1)] = p;   // a few hundred at most
}
#endif
// Iterations of the block-exchange phase refactor the set from scratch (asm_tile_solve, as in the round kernels).  Once a
// problem is down to SINGLE exchanges -- where the stragglers of an ill-conditioned plant spend hundreds of iterations --
// the set changes by one index per iteration: the kernel then keeps a dense Cholesky factor in LDS and appends a row
// (one forward substitution) or drops one (Givens rotations over the trailing columns) instead of gathering and
// factoring |A|^2 entries again; multipliers by two substitutions.  Every accepted result still passes the fp64
// certificate, so a factor that lost accuracy can only cost time (the problem then falls back to the PDIP path).
#ifdef ASM_TAIL_PROF
__device__ unsigned long long asm_tail_prof[8];            // diagnostics build only: shader-clock sums per phase over all workgroups
#define ASM_TP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (tid == 0) atomicAdd(&asm_tail_prof[i], t_ - tp_); tp_ = t_; } while (0)
#else
#define ASM_TP(i) do { } while (0)
#endif
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256) void asm_tail_k(AsmDev d, int budget) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_bad, wsum[4], s_i[4], s_n[4], s_m, s_fast;
  __shared__ double s_d[12];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= d.counters[ASM_CNT_TAIL]) return;
  const int p = d.biglist[blockIdx.x];
  double* rA = sm;                                           // [max_active] rhs -> lam
  double* Yt = rA + d.max_active;
  double* Tl = Yt + ASM_TS;                                  // tiles of sets of up to 16 ASM_TAIL_MB bounds (else: L2 slab)
  double* Ld = Tl;                                           // ... or the dense factor of the single-exchange phase
  unsigned char* dec = reinterpret_cast<unsigned char*>(Tl + ASM_TAIL_AREA);   // [n] decisions
  double* rhs = reinterpret_cast<double*>(dec + ((d.n + 15) / 16) * 16);   // [ASM_FM] x_unc,A - b_A in factor order
  double* vv = rhs + ASM_FM;                                 // [ASM_FM] work vector
  double* ww = vv + ASM_FM;                                  // [ASM_FM] second work vector (dual steps)
  int* al = reinterpret_cast<int*>(ww + ASM_FM);             // [ASM_FM] active indices in factor order
  double* Tg = d.scratch + (size_t)blockIdx.x * ((size_t)(d.max_active / 16) * (d.max_active / 16 + 1) / 2 * ASM_TS);
  const size_t o = (size_t)p * d.np;
  unsigned char* st = d.st + (size_t)p * d.n;
  const int* idx = d.idxg + (size_t)p * d.max_active;
  const double* lbp = d.lb + (size_t)p * d.nu;
  const double* ubp = d.ub + (size_t)p * d.nu;
  int best = d.ninf_best[p], grace = d.alpha[p], hi = d.hi[p], rounds = d.rounds[p];
  if (tid == 0) d.hi[p] = d.n;                               // (from here on bounds anywhere may join: asm_certify_k scans [0, hi))
  int single = 0;                                            // the previous iteration ended with a single exchange
  int fast = 0, m = 0;                                       // dense factor valid for the current set (of size m)
  int gi = 0;                                                // dual active-set phase (see the exchange rule below)
  // one row leaves the dense factor (wave 0): the logical arrays and the rows of L move up, Givens rotations restore the triangle
  auto drop_row = [&](int j, int mm) {
    int av[3]; double hv[3], lv[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = j + lane + 64 * u;
      av[u] = 0; hv[u] = 0.0; lv[u] = 0.0;
      if (i + 1 < mm) { av[u] = al[i + 1]; hv[u] = rhs[i + 1]; lv[u] = rA[i + 1]; }
    }
    ASM_FENCE();
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int i = j + lane + 64 * u;
      if (i + 1 < mm) { al[i] = av[u]; rhs[i] = hv[u]; rA[i] = lv[u]; }
    }
    // new row i = old row i + 1 (columns 0..i + 1), in ascending order
    for (int i = j; i + 1 < mm; ++i) {
      const double* src = Ld + asm_frow(i + 1);
      double* dst = Ld + asm_frow(i);
      double tv[3];
#pragma unroll
      for (int u = 0; u < 3; ++u) { const int c = lane + 64 * u; tv[u] = c <= i + 1 ? src[c] : 0.0; }
      ASM_FENCE();
#pragma unroll
      for (int u = 0; u < 3; ++u) { const int c = lane + 64 * u; if (c <= i + 1) dst[c] = tv[u]; }
      ASM_FENCE();
    }
    --mm;
    for (int k = j; k < mm; ++k) {                           // row i >= j now has an entry in column i + 1
      double* rk = Ld + asm_frow(k);
      const double a = rk[k], b = rk[k + 1];
      const double rr = sqrt(a * a + b * b), c = a / rr, s = b / rr;
      ASM_FENCE();
      if (lane == 0) { rk[k] = rr; rk[k + 1] = 0.0; }
      for (int i = k + 1 + lane; i < mm; i += 64) {
        double* ri = Ld + asm_frow(i);
        const double pp = ri[k], qq = ri[k + 1];
        ri[k] = c * pp + s * qq;
        ri[k + 1] = c * qq - s * pp;
      }
      ASM_FENCE();
    }
  };
#ifdef ASM_TAIL_PROF
  unsigned long long tp_ = __builtin_amdgcn_s_memtime();
#endif
  for (int it = 0; it < budget; ++it) {
    ASM_TP(4);
    if (!fast) {
      m = asm_count_one(d, p, hi, wsum);
      if (m > d.max_active) { if (tid == 0) d.state[p] = ASM_FALLBACK; return; }
      if (single && (m > 0 || gi) && m < ASM_FM) {
        // ---- build the dense factor: gather the lower triangle, right-looking Cholesky (three barriers per column)
        __syncthreads();                                       // idx (asm_count_one) complete
        for (int i = tid; i < m; i += 256) {
          const int a = idx[i];
          al[i] = a;
          rhs[i] = d.xunc[o + a] - (st[a] == 1 ? ubp[a % d.nu] : lbp[a % d.nu]);
        }
        if (tid == 0) s_bad = 0;
        __syncthreads();
        for (int i = tid >> 4; i < m; i += 16) {
          const double* Hr = d.H + (size_t)al[i] * d.np;
          for (int j = tid & 15; j <= i; j += 16) Ld[asm_frow(i) + j] = Hr[al[j]];
        }
        for (int k = 0; k < m; ++k) {
          __syncthreads();
          const double piv = Ld[asm_frow(k) + k];
          if (!(piv > 0.0)) { if (tid == 0) s_bad = 1; break; }        // uniform: every thread reads the same value
          const double rs = 1.0 / sqrt(piv);
          __syncthreads();
          if (tid == 0) Ld[asm_frow(k) + k] = piv * rs;
          for (int i = k + 1 + tid; i < m; i += 256) Ld[asm_frow(i) + k] *= rs;
          __syncthreads();
          for (int i = k + 1 + (tid >> 4); i < m; i += 16) {
            const double lik = Ld[asm_frow(i) + k];
            for (int j = k + 1 + (tid & 15); j <= i; j += 16) Ld[asm_frow(i) + j] -= lik * Ld[asm_frow(j) + k];
          }
        }
        __syncthreads();
        if (s_bad) { if (tid == 0) d.state[p] = ASM_FALLBACK; return; }
        fast = 1;
      } else {
        if (asm_tile_solve(d, p, m, idx, rA, Yt, m <= 16 * ASM_TAIL_MB ? Tl : Tg, &s_bad)) { if (tid == 0) d.state[p] = ASM_FALLBACK; return; }
      }
    }
    ASM_TP(0);
    if (fast) {                                              // lam = L^-T L^-1 rhs  (wave 0)
      if (wave == 0) {
        for (int i = lane; i < m; i += 64) vv[i] = rhs[i];
        ASM_FENCE();
        asm_fwd(Ld, vv, m, lane);
        asm_bwd(Ld, vv, m, lane);
        for (int i = lane; i < m; i += 64) rA[i] = vv[i];
      }
      __syncthreads();
    }
    ASM_TP(1);
    const int* lst = fast ? al : idx;                        // the set the multipliers rA[0..m) belong to
    // x and the tests, decisions recorded (255: stays)
    int ninf = 0, rmin = 0x7fffffff;                         // rmin: the infeasible index a single exchange takes (the smallest)
    int nneg = 0;                                            // active bounds whose multiplier has the wrong sign
    double l1 = 0.0, lmin = 1e300;
    // Column window, as in the lock-step rounds: while the set is still moving only the columns up to one stage past the last
    // active bound are evaluated (in MPC the active bounds sit in the first stages: 512 of 4480 columns at CDU size -- the x loop
    // over ALL columns was most of an iteration); a set that settles inside the window gets the remaining columns once, and a
    // bound violated out there sends the problem on.  (On the dense factor -- single exchanges -- the list is not ordered: all
    // columns.)
    const int lastact = m > 0 ? idx[m - 1] : -1;
    const int xlim = fast ? d.n : min(d.n, ((lastact + 1 + d.nu + 255) / 256) * 256);
    int xdone = xlim;                                        // columns whose decisions dec[] are this iteration's
    // x over the columns [r0, r1) and the feasibility tests of the free variables there
    // one column: x from the product (free) or the bound (active), test, write-out, decision
    auto xcol = [&](int r, int sr, double prod, int& ninf, int& rmin) {
      const int k = r % d.nu;
      const double lb = lbp[k], ub = ubp[k];
      unsigned char dc = 255;
      double x;
      if (sr == 0) {
        x = d.xunc[o + r] - prod;
        if (x > ub + d.bound_tol) dc = 1; else if (x < lb - d.bound_tol) dc = 2;
      } else x = sr == 1 ? ub : lb;
      if (r < d.nout) d.u_out[(size_t)p * d.ldu + r] = x;      // final once nothing changes
      d.x[o + r] = x;                                          // ... and as a GEMM row, should P itself have to confirm it
      dec[r] = dc;
      if (dc != 255) { ++ninf; rmin = min(rmin, r); }
    };
    auto xpass = [&](int r0, int r1, int& ninf, int& rmin) {
    // (lam H)[r] = sum_i lam_i H[idx_i][r], the GEMM's indexing: consecutive threads read consecutive columns of row idx_i
    // (H[r][idx_i] -- the same number, H is symmetric -- would touch one cache line per thread).  A thread owns FOUR consecutive
    // columns: one 32-byte load per row of Pinv and thread, eight rows in flight (the entries come from L2 / Infinity Cache at
    // ~1 us each; 8-byte loads, sixteen in flight, made this loop -- (columns / 256) m / 16 dependent round trips -- most of an
    // iteration with a large set: the columns beyond the window of a 149-chain step took 0.1 of its 0.3 ms)
    int rs = r0;                                             // the scalar loop below starts here
    if ((d.np & 3) == 0 && (r0 & 3) == 0) {
      const int rv1 = r0 + ((r1 - r0) & ~3);
      for (int r = r0 + 4 * tid; r < rv1; r += 1024) {
        const int s0 = st[r], s1 = st[r + 1], s2 = st[r + 2], s3 = st[r + 3];
        f64x4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
        if (s0 == 0 || s1 == 0 || s2 == 0 || s3 == 0) {
          const double* Hc = d.H + r;
          int i = 0;
          for (; i + 8 <= m; i += 8) {
            f64x4_t h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) h[u] = *reinterpret_cast<const f64x4_t*>(Hc + (size_t)lst[i + u] * d.np);
#pragma unroll
            for (int u = 0; u < 8; u += 2) { a0 += h[u] * rA[i + u]; a1 += h[u + 1] * rA[i + u + 1]; }
          }
          for (; i < m; ++i) a0 += *reinterpret_cast<const f64x4_t*>(Hc + (size_t)lst[i] * d.np) * rA[i];
          a0 += a1;
        }
        xcol(r, s0, a0[0], ninf, rmin); xcol(r + 1, s1, a0[1], ninf, rmin);
        xcol(r + 2, s2, a0[2], ninf, rmin); xcol(r + 3, s3, a0[3], ninf, rmin);
      }
      rs = rv1;
    }
    for (int r = rs + tid; r < r1; r += 256) {
      const int sr = st[r];
      double prod = 0.0;
      if (sr == 0) {
        const double* Hc = d.H + r;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int i = 0;
        for (; i + 4 <= m; i += 4) {
          a0 += Hc[(size_t)lst[i] * d.np] * rA[i];
          a1 += Hc[(size_t)lst[i + 1] * d.np] * rA[i + 1];
          a2 += Hc[(size_t)lst[i + 2] * d.np] * rA[i + 2];
          a3 += Hc[(size_t)lst[i + 3] * d.np] * rA[i + 3];
        }
        for (; i < m; ++i) a0 += Hc[(size_t)lst[i] * d.np] * rA[i];
        prod = (a0 + a1) + (a2 + a3);
      }
      xcol(r, sr, prod, ninf, rmin);
    }
    };
    xpass(0, xlim, ninf, rmin);
    __syncthreads();                                         // dec of the free variables complete
    ASM_TP(2);
    for (int i = tid; i < m; i += 256) {
      const int a = lst[i], sa = st[a];
      const double l = rA[i];
      l1 += fabs(l); lmin = fmin(lmin, fabs(l));
      if ((sa == 1 && l <= 0.0) || (sa == 2 && l >= 0.0)) { dec[a] = 0; ++ninf; ++nneg; if (!gi) rmin = min(rmin, a); }
    }
    for (int off = 32; off > 0; off >>= 1) {
      ninf += __shfl_xor(ninf, off); rmin = min(rmin, __shfl_xor(rmin, off)); nneg += __shfl_xor(nneg, off);
      l1 += __shfl_xor(l1, off); lmin = fmin(lmin, __shfl_xor(lmin, off));
    }
    __syncthreads();
    if (lane == 0) { wsum[wave] = ninf; s_i[wave] = rmin; s_n[wave] = nneg; s_d[wave] = l1; s_d[4 + wave] = lmin; }
    __syncthreads();
    ninf = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    nneg = s_n[0] + s_n[1] + s_n[2] + s_n[3];
    rmin = min(min(s_i[0], s_i[1]), min(s_i[2], s_i[3]));           // (dual phase: the smallest VIOLATED index)
    if (ninf == 0 && xlim < d.n) {                           // settled inside the window: the columns beyond, once
      int ninf2 = 0, rmin2 = 0x7fffffff;
      xpass(xlim, d.n, ninf2, rmin2);
      for (int off = 32; off > 0; off >>= 1) { ninf2 += __shfl_xor(ninf2, off); rmin2 = min(rmin2, __shfl_xor(rmin2, off)); }
      __syncthreads();
      if (lane == 0) { wsum[wave] = ninf2; s_i[wave] = rmin2; }
      __syncthreads();
      ninf = wsum[0] + wsum[1] + wsum[2] + wsum[3];
      rmin = min(min(s_i[0], s_i[1]), min(s_i[2], s_i[3]));
      xdone = d.n;
    }
    ++rounds;
    ASM_TP(3);
#ifdef ASM_TAIL_PROF
    if (tid == 0) atomicAdd(&asm_tail_prof[fast ? 5 : 6], 1ull);
#endif
    if (ninf == 0 && fast) {                                 // settled on an updated factor: confirm with a fresh
      fast = 0; single = 0;                                  // factorisation of the same set (next iteration)
      __syncthreads();
      continue;
    }
    if (ninf == 0) {                                         // settled: certificate as in asm_update_k
      double x1 = 0.0;
      for (int k = tid; k < d.ka; k += 256) x1 += fabs(d.x0[(size_t)p * d.ka + k]);
      for (int off = 32; off > 0; off >>= 1) x1 += __shfl_xor(x1, off);
      if (lane == 0) s_d[8 + wave] = x1;
      __syncthreads();
      const double L1 = s_d[0] + s_d[1] + s_d[2] + s_d[3], X1 = s_d[8] + s_d[9] + s_d[10] + s_d[11];
      const double LM = fmin(fmin(s_d[4], s_d[5]), fmin(s_d[6], s_d[7]));
      const double QI = d.tqmax * X1;
      const double bnd = 2.0 * (d.e1max * X1 + d.e2max * L1) + 1e-14 * (QI + L1);
      const bool sure = bnd <= d.stat_tol * d.pscale && LM > bnd;
      if (tid == 0) {
        d.state[p] = sure ? ASM_CERT : ASM_DONE;
        if (!sure) atomicAdd(&d.counters[ASM_CNT_DONE], 1);
        d.rounds[p] = rounds;
      }
      return;
    }
    // exchange rule (see asm_update_k)
    single = 0;
    // (the problems that reach the tail have shown that block exchanges do not settle them: a new minimum buys
    // ASM_TAIL_GRACE block exchanges here, not ASM_GRACE, so most iterations are cheap single exchanges)
    if (gi) single = 1;
    else if (ninf < best) { best = ninf; grace = ASM_TAIL_GRACE; }
    else if (grace > 0) --grace;
    else single = 1;
    if (single && !gi && d.tail_gi && m + 1 < ASM_FM) {
      // (NNMPC_TAIL_GI=1 only; OFF by default.)  From here on: DUAL active-set steps (Goldfarb & Idnani 1983) instead of Murty's
      // one-index-per-iteration rule.  Measured: no better -- the stragglers of the cond-4e7 plant need the same number of
      // iterations either way (36 at most in a 10 000-problem batch), a dual iteration costs more (the purge refactors, every
      // step takes two more substitutions): 5.8 against 5.2 ms per CSTRs-size step; on random dense Hessians with cond >= 1e5
      // and half the bounds active (scripts/stress_asm.py, seeds 1 and 3) it exhausts the iteration budget MORE often (174 and
      // 308 of ~1600 problems against 130 and 166): after the purge it has to add the missing bounds one by one.  Kept for A/B.
      // (i) Purge: drop ALL bounds with a wrong-sign multiplier, solve again, until the set is
      // dual feasible -- x is then the optimum on that set with multipliers >= 0.  (ii) Add the violated bound with the smallest
      // index by a dual step: multipliers of the set move along -w, w = S^-1 H[A,r] (two substitutions with the dense factor);
      // a bound whose multiplier reaches zero first leaves (Givens downdate) and the step is repeated, else the new bound
      // joins (append a row).  Every step raises the dual objective: no cycling, and the number of steps is about the number of
      // bounds still missing.  Multipliers and x are recomputed from the factor after every completed step; a set that settles
      // is confirmed by a fresh factorisation and the certificate like any other.
      gi = 1;
    }
    if (gi && m + 1 >= ASM_FM) gi = 0;                       // too large for the dense factor: Murty's rule on fresh factorisations
    const int dsel = dec[rmin < d.n ? rmin : 0];             // what the single exchange does: 1 / 2 add at that bound, 0 drop
    __syncthreads();
    if (gi) {
      hi = d.n;
      if (nneg > 0 || !fast) {
        // (i) purge -- or, dual feasible already, nothing: the next iteration builds the dense factor of the set as it stands
        for (int r = tid; r < xdone; r += 256) if (dec[r] == 0) st[r] = 0;
        fast = 0;
        __syncthreads();
        continue;
      }
      // (ii) dual step for bound rmin (violated; the set is dual feasible, the dense factor valid), wave 0
      if (wave == 0) {
        int ok = 1, mm = m;
        const int r = rmin, sp = dsel == 1 ? 1 : -1;
        const double bp = dsel == 1 ? ubp[r % d.nu] : lbp[r % d.nu];
        double viol = sp * (d.x[o + r] - bp), mup = 0.0;     // > bound_tol
        const double* Hr = d.H + (size_t)r * d.np;
        const double hd = Hr[r];
        int done = 0;
        for (int step = 0; step < ASM_FM + 8 && ok && !done; ++step) {
          for (int i = lane; i < mm; i += 64) vv[i] = Hr[al[i]];
          ASM_FENCE();
          asm_fwd(Ld, vv, mm, lane);
          double s2 = 0.0;
          for (int i = lane; i < mm; i += 64) s2 += vv[i] * vv[i];
          for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
          const double d2 = hd - s2;                         // n_p' (H - H[:,A] S^-1 H[A,:]) n_p > 0
          if (!(d2 > 1e-13 * hd)) { ok = 0; break; }
          for (int i = lane; i < mm; i += 64) ww[i] = vv[i];
          ASM_FENCE();
          asm_bwd(Ld, ww, mm, lane);                         // w = S^-1 H[A,r]
          // largest step that keeps every multiplier of the set >= 0:  mu_j = s_j lam_j falls by t s_j s_p w_j
          double t1 = 1e300; int k1 = -1;
          for (int i = lane; i < mm; i += 64) {
            const int sj = st[al[i]] == 1 ? 1 : -1;
            const double rj = sj * sp * ww[i];
            if (rj > 0.0) { const double q = fmax(sj * rA[i], 0.0) / rj; if (q < t1) { t1 = q; k1 = i; } }
          }
          for (int off = 32; off > 0; off >>= 1) {
            const double t_o = __shfl_xor(t1, off); const int k_o = __shfl_xor(k1, off);
            if (t_o < t1 || (t_o == t1 && k_o >= 0 && (k1 < 0 || k_o < k1))) { t1 = t_o; k1 = k_o; }
          }
          const double t2 = viol / d2;                       // full step: the new bound becomes active
          const double t = t2 <= t1 ? t2 : t1;
          for (int i = lane; i < mm; i += 64) rA[i] -= t * sp * ww[i];
          mup += t;
          ASM_FENCE();
          if (t2 <= t1) {
            if (mm + 1 >= ASM_FM) { ok = 0; break; }
            double* row = Ld + asm_frow(mm);
            for (int i = lane; i < mm; i += 64) row[i] = vv[i];
            if (lane == 0) {
              row[mm] = sqrt(d2); al[mm] = r; rhs[mm] = d.xunc[o + r] - bp; rA[mm] = sp * mup;
              st[r] = (unsigned char)dsel;
            }
            ASM_FENCE();
            ++mm; done = 1;
          } else {
            viol -= t * d2;
            if (lane == 0) st[al[k1]] = 0;
            ASM_FENCE();
            drop_row(k1, mm);
            --mm;
          }
        }
        if (!done) ok = 0;
        if (lane == 0) { s_m = mm; s_fast = ok; }
      }
      __syncthreads();
      m = s_m; fast = s_fast;
      __syncthreads();
      continue;
    }
    for (int r = tid; r < xdone; r += 256) {
      const unsigned char dc = dec[r];
      if (dc != 255 && (!single || r == rmin)) st[r] = dc;
    }
    hi = d.n;
    if (!single) fast = 0;
    else if (fast) {
      // ---- one index enters or leaves the factor (wave 0), or the factor is given up (next iteration rebuilds / refactors)
      if (wave == 0) {
        int ok = 1, mm = m;
        if (dsel != 0) {                                     // add rmin at its upper (1) / lower (2) bound
          if (mm + 1 >= ASM_FM) ok = 0;
          else {
            const double* Hr = d.H + (size_t)rmin * d.np;
            for (int i = lane; i < mm; i += 64) vv[i] = Hr[al[i]];
            ASM_FENCE();
            asm_fwd(Ld, vv, mm, lane);
            double s2 = 0.0;
            for (int i = lane; i < mm; i += 64) s2 += vv[i] * vv[i];
            for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off);
            const double hd = Hr[rmin], d2 = hd - s2;
            if (!(d2 > 1e-13 * hd)) ok = 0;
            else {
              double* row = Ld + asm_frow(mm);
              for (int i = lane; i < mm; i += 64) row[i] = vv[i];
              if (lane == 0) {
                row[mm] = sqrt(d2); al[mm] = rmin;
                rhs[mm] = d.xunc[o + rmin] - (dsel == 1 ? ubp[rmin % d.nu] : lbp[rmin % d.nu]);
              }
              ++mm;
            }
          }
        } else {                                             // drop rmin: its row leaves, Givens rotations restore the triangle
          int j = -1;
          for (int i = lane; i < mm; i += 64) if (al[i] == rmin) j = i;
          for (int off = 32; off > 0; off >>= 1) j = max(j, __shfl_xor(j, off));
          if (j < 0) ok = 0;
          else {
            drop_row(j, mm);
            --mm;
          }
        }
        if (lane == 0) { s_m = mm; s_fast = ok; }
      }
      __syncthreads();
      m = s_m; fast = s_fast;
    }
    __syncthreads();
  }
  if (tid == 0) { d.state[p] = ASM_FALLBACK; d.rounds[p] = rounds; }
}
#endif

// Independent fp64 certification with P itself (px = x P):  stationarity on the free set,
// multiplier signs on the active set, feasibility; active-set bits and status of finished problems
// (u itself was written by asm_update_k / asm_wide_k).
#if ASM_OWN_KERNELS
__global__ __launch_bounds__(256) void asm_certify_k(AsmDev d, double gscale_min) {
  __shared__ int cnt[4];
  __shared__ double gq[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  const int stt = d.state[p];
  if (stt == ASM_INVALID) {                     // rejected by asm_init_k: no solution, no active set
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    for (int r = tid; r < d.nout; r += 256) d.u_out[(size_t)p * d.ldu + r] = qnan;
    if (d.act_out) for (int w = tid; w < d.words; w += 256) d.act_out[(size_t)p * d.words + w] = 0u;
    if (tid == 0) {
      if (d.status_out) d.status_out[p] = 2;    // NNMPC_ST_NUMERIC
      if (d.iters_out) { d.iters_out[2 * p] = 0; d.iters_out[2 * p + 1] = 0; }
    }
    return;
  }
  if (stt != ASM_DONE && stt != ASM_CERT) {
    if (tid == 0 && d.status_out) d.status_out[p] = 3;   // marks "not solved here" for the caller
    return;
  }
  const size_t o = (size_t)p * d.np;
  const unsigned char* st = d.st + (size_t)p * d.n;
  int bad = 0;
  double qm = 0.0, gf = 0.0;
  if (stt == ASM_DONE) {                        // full check with P itself
    for (int r = tid; r < d.n; r += 256) {
      const int k = r % d.nu;
      const double lb = d.lb[(size_t)p * d.nu + k], ub = d.ub[(size_t)p * d.nu + k];
      const double x = d.x[o + r];
      const int s = st[r];
      const double g = d.px[o + r] + d.q64[o + r];
      qm = fmax(qm, fabs(d.q64[o + r]));
      // every test is written so that a NaN fails it (fmax / a plain "violated?" comparison would let it through)
      if (s == 0) { bad += !(fabs(g) <= 1.79e308); gf = fmax(gf, fabs(g)); bad += !(x <= ub + d.bound_tol) || !(x >= lb - d.bound_tol); }
      else if (s == 1) bad += !(g < 0.0);
      else bad += !(g > 0.0);
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    bad += __shfl_xor(bad, off);
    qm = fmax(qm, __shfl_xor(qm, off));
    gf = fmax(gf, __shfl_xor(gf, off));
  }
  if ((tid & 63) == 0) { cnt[tid >> 6] = bad; gq[tid >> 6] = qm; }
  __syncthreads();
  const int nbad = cnt[0] + cnt[1] + cnt[2] + cnt[3];
  const double qs = fmax(fmax(gq[0], gq[1]), fmax(gq[2], gq[3]));
  __syncthreads();
  if ((tid & 63) == 0) gq[tid >> 6] = gf;
  __syncthreads();
  const double gfm = fmax(fmax(gq[0], gq[1]), fmax(gq[2], gq[3]));
  const int ok = nbad == 0 && gfm <= d.stat_tol * fmax(gscale_min, qs);   // (a NaN in q makes qs NaN -> fmax -> gscale_min: still a finite bar)
  if (d.act_out) {
    // bit k*2nu + c: upper bound of variable k*nu + c active, bit k*2nu + nu + c: lower bound (row order of the
    // reference's G).  One pass over the bound states (one division per variable), words assembled in LDS.
    __shared__ uint32_t wb[1024];
    for (int w0 = 0; w0 < d.words; w0 += 1024) {
      const int nwd = min(1024, d.words - w0);
      for (int w = tid; w < nwd; w += 256) wb[w] = 0u;
      __syncthreads();
      // variables whose two bits fall into words [w0, w0 + nwd): stages [32 w0 / 2nu, 32 (w0 + nwd) / 2nu]
      // (no bound at or beyond hi[p] is active: asm_count_k's invariant -- 512 of the 4480 bound states at the CDU size)
      const int k0 = (32 * w0) / (2 * d.nu), k1 = min((min(d.hi[p], d.n) + d.nu - 1) / d.nu, (32 * (w0 + nwd) + 2 * d.nu - 1) / (2 * d.nu));
      for (int r = k0 * d.nu + tid; r < k1 * d.nu; r += 256) {
        const int sv = st[r];
        if (sv) {
          const int kk = r / d.nu, c = r - kk * d.nu;
          const int bit = kk * 2 * d.nu + (sv == 1 ? c : d.nu + c), w = (bit >> 5) - w0;
          if (w >= 0 && w < nwd) atomicOr(&wb[w], 1u << (bit & 31));
        }
      }
      __syncthreads();
      for (int w = tid; w < nwd; w += 256) d.act_out[(size_t)p * d.words + w0 + w] = wb[w];
      __syncthreads();
    }
  }
  if (tid == 0) {
    if (d.status_out) d.status_out[p] = ok ? 0 : 3;      // 3: let the PDIP path redo it
    if (d.iters_out) { d.iters_out[2 * p] = 0; d.iters_out[2 * p + 1] = 0; }   // no PDIP iterations, no n^3 factorisations
  }
}
#endif

// ---- gather / scatter of the problems handed to the PDIP fallback
#if ASM_OWN_KERNELS
__global__ void asm_gather_k(double* x0c, double* lbc, double* ubc, const double* x0, const double* lb,
                             const double* ub, const int* list, int cnt, int n_aug, int nu) {
  const int i = blockIdx.x;
  if (i >= cnt) return;
  const int p = list[i];
  for (int k = threadIdx.x; k < n_aug; k += blockDim.x) x0c[(size_t)i * n_aug + k] = x0[(size_t)p * n_aug + k];
  for (int k = threadIdx.x; k < nu; k += blockDim.x) {
    lbc[(size_t)i * nu + k] = lb[(size_t)p * nu + k];
    ubc[(size_t)i * nu + k] = ub[(size_t)p * nu + k];
  }
}
#endif
#if ASM_OWN_KERNELS
__global__ void asm_gather_guess_k(unsigned char* guess, const unsigned char* st, const int* state, const int* list, int cnt, int n) {
  const int i = blockIdx.x;
  if (i >= cnt) return;
  const int p = list[i];
  const bool settled = state[p] == ASM_DONE;                 // finished here, rejected by the check with P
  for (int k = threadIdx.x; k < n; k += blockDim.x) guess[(size_t)i * n + k] = settled ? st[(size_t)p * n + k] : (unsigned char)255;
}
#endif
#if ASM_OWN_KERNELS
__global__ void asm_scatter_k(double* u, uint32_t* act, int* status, int* iters, const double* uc,
                              const uint32_t* actc, const int* stc, const int* itc, const int* list, int cnt,
                              int n, int words, int ldu, int nout) {
  const int i = blockIdx.x;
  if (i >= cnt) return;
  const int p = list[i];
  for (int k = threadIdx.x; k < nout; k += blockDim.x) u[(size_t)p * ldu + k] = uc[(size_t)i * n + k];
  if (act) for (int k = threadIdx.x; k < words; k += blockDim.x) act[(size_t)p * words + k] = actc[(size_t)i * words + k];
  if (threadIdx.x == 0) {
    if (status) status[p] = stc[i];
    if (iters) { iters[2 * p] = itc[2 * i]; iters[2 * p + 1] = itc[2 * i + 1]; }
  }
}
#endif

}  // namespace nnmpc
