// Batched box-constrained condensed-QP solver for gfx950 (MI355X).
//
//   min 1/2 u'Pu + (tq x0)'u   s.t.  lb <= u_k <= ub  (k = 0..N-1)
//
// replaces DenseQPRegulator.solve -> cvxopt.solvers.qp of the reference
// (lib/linearMPC.py:495-512) for B independent (x0, lb, ub) at once.
//
// Algorithm per problem (all problems of a wave advance in lock-step "rounds",
// a per-problem state machine on the device decides what each round does):
//   PDIP   Mehrotra predictor-corrector on the reduced KKT system
//          (P + diag(z_u/s_u + z_l/s_l)) du = r, f32 Cholesky (MFMA) + 2 solves,
//          warm-started at the clipped unconstrained minimiser, run only until
//          the active set is identifiable (ipm_tol).
//   POLISH active set from z > s; solve the free block exactly: f32 Cholesky of
//          the masked P + iterative refinement with f64 residuals (P, q in f64),
//          then an f64 KKT check; a failed check updates the set (primal-dual
//          active-set step) and repeats.  status OPTIMAL = KKT verified in f64.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include "../../include/nnmpc.h"
#define ASM_WG_TU (-1)              // the workgroup kernels of qp_wg.h are compiled in qp_wg_kernels.hip: declarations only here
#include "chol_kernels.h"
#include "gemm_kernels.h"
#include "qp_asm.h"
#include "qp_wide.h"
#include "qp_small.h"
#include "qp_predict.h"
#include "common.h"
static constexpr int ASM_SMALL9_LDS_MAX = 100 * 1024;   // dynamic LDS the nine-block instance of asm_small_k may ask for (81 KB at n = 1024, nu = 64)

using namespace nnmpc;

namespace {

enum { PH_INIT = 0, PH_IPM = 1, PH_POLISH = 2, PH_DONE = 3 };
enum { PS_START = 0, PS_CG = 1, PS_CHECK = 2 };
enum { CNT_ACTIVE = 0, CNT_FACTOR = 1, CNT_IPM = 2, CNT_POLISH = 3, CNT_SOLVE = 4 };

constexpr int POLISH_GRACE = 10;   // polish rounds without a new minimum of infeasible indices before single exchanges
struct QpDev {
  int n, np, nu, slots;
  int max_ipm, max_polish, max_refine, stale_max_changes, stale_cg_limit;
  float ipm_tol, delta;
  double refine_tol, bound_tol, stat_tol, pscale;
  const float* pdiag;  // [np] diagonal of the normalised P
  // f32 [slots][np]
  float *u, *zu, *zl, *lbv, *ubv, *q, *PU, *rd, *rhs, *sol, *dua, *dvec, *mask, *uunc;
  // f64 [slots][np]
  double *x, *q64, *PX, *r64, *p64, *v64;   // PX = P64 * v64
  unsigned char* st;  // [slots][np]  0 free, 1 at upper, 2 at lower
  double *lb64, *ub64;        // [slots][nu] bounds of the problem resident in the slot
  // segment-level inputs (problem-indexed) the slots are (re)filled from
  const double *q64_all, *lb_all, *ub_all;   // [seg][np], [seg][nu], [seg][nu]
  const float* uunc_all;                      // [seg][np]
  const unsigned char* guess_all;             // [seg][n] active-set guess (0 free/1 upper/2 lower) or NULL
  int *slot_prob, *age, *next_prob;
  int seg_count, max_rounds;
  int *phase, *f_factor, *f_solve, *istep, *ipm_it, *nfac, *prounds, *rcnt, *psub, *fail, *stale;
  int *pninf, *pgrace;   // polish exchange rule: fewest infeasible indices seen, rounds of grace left (see kkt_check)
  int* ptie;             // polish tie round done (see kkt_check): bounds violated by less than bound_tol were made active once
  float *mu, *gap, *smu, *qscale;
  double* rz;
  int* counters;
  double* u_out;       // [problems][ldu]: the first nout entries of every solution
  int ldu, nout;
  uint32_t* act_out;   // [slots][words]
  int* status_out;     // [slots]
  int* iters_out;      // [slots][2]
  int words;
};

__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_maxd(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
// All 256 threads must call; result broadcast to all.
__device__ float block_max(float v, float* sh) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ double block_maxd(double v, double* sh) {
  v = wave_maxd(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
__device__ double block_sumd(double v, double* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ int block_sum_i(int v, int* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

#define SLACK_MIN 1e-12f

__device__ void write_outputs(const QpDev& d, int p, int status);

// ---------------------------------------------------------------------------
// Continuous batching: every free (DONE) slot grabs the next unsolved problem of
// the segment and is initialised for the PDIP (warm start = clipped u_unc).
__global__ __launch_bounds__(256) void refill_k(QpDev d) {
  __shared__ float shf[4];
  __shared__ int s_idx;
  const int p = blockIdx.x, tid = threadIdx.x;
  const size_t o = (size_t)p * d.np;
  if (d.phase[p] != PH_DONE) return;
  if (tid == 0) s_idx = atomicAdd(d.next_prob, 1);
  __syncthreads();
  const int idx = s_idx;
  if (idx >= d.seg_count) {
    if (tid == 0) d.slot_prob[p] = -1;
    return;
  }
  const size_t oi = (size_t)idx * d.np;
  if (tid < d.nu) {
    d.lb64[(size_t)p * d.nu + tid] = d.lb_all[(size_t)idx * d.nu + tid];
    d.ub64[(size_t)p * d.nu + tid] = d.ub_all[(size_t)idx * d.nu + tid];
  }
  float qm = 0.f;
  int invalid = 0;
  for (int r = tid; r < d.np; r += 256) {
    if (r < d.n) {
      const int c = r % d.nu;
      invalid |= !(fabs(d.q64_all[oi + r]) <= 1.79e308);   // (fmaxf below would drop a NaN)
      const float lb = (float)d.lb_all[(size_t)idx * d.nu + c], ub = (float)d.ub_all[(size_t)idx * d.nu + c];
      const double q64 = d.q64_all[oi + r];
      const float q = (float)(q64 / d.pscale);
      const float w = ub - lb;
      float u0 = d.uunc_all[oi + r];
      u0 = fminf(fmaxf(u0, lb + 0.05f * w), ub - 0.05f * w);
      d.lbv[o + r] = lb; d.ubv[o + r] = ub; d.q[o + r] = q; d.u[o + r] = u0; d.q64[o + r] = q64;
      qm = fmaxf(qm, fabsf(q));
    } else {
      d.lbv[o + r] = -1.f; d.ubv[o + r] = 1.f; d.q[o + r] = 0.f; d.u[o + r] = 0.f; d.q64[o + r] = 0.0;
    }
    d.zu[o + r] = 0.f; d.zl[o + r] = 0.f;
    d.mask[o + r] = 1.f; d.dvec[o + r] = 0.f; d.rhs[o + r] = 0.f; d.sol[o + r] = 0.f;
    d.dua[o + r] = 0.f; d.rd[o + r] = 0.f; d.x[o + r] = 0.0; d.st[o + r] = 0;
    d.r64[o + r] = 0.0; d.p64[o + r] = 0.0; d.v64[o + r] = 0.0;
  }
  qm = block_max(qm, shf);
  {
    // NaN / Inf in q (i.e. in x0) or a bound pair with lb > ub (or a NaN): not solved at all -- status NUMERIC, u = NaN
    invalid |= !(qm <= 3.0e38f);
    if (tid < d.nu) invalid |= !(d.lb_all[(size_t)idx * d.nu + tid] <= d.ub_all[(size_t)idx * d.nu + tid]);
    invalid = __syncthreads_or(invalid);
    if (invalid) {
      const double qnan = __longlong_as_double(0x7ff8000000000000ll);
      for (int r = tid; r < d.np; r += 256) d.x[o + r] = qnan;
      if (tid == 0) { d.slot_prob[p] = idx; d.fail[p] = 1; d.ipm_it[p] = d.nfac[p] = 0; }
      __syncthreads();
      write_outputs(d, p, NNMPC_ST_NUMERIC);       // the slot stays free (PH_DONE) and takes the next problem next round
      return;
    }
  }
  // (a guess row that starts with 255 means: no guess for this problem)
  const bool warm = d.guess_all != nullptr && d.guess_all[(size_t)idx * d.n] != 255;
  if (warm) {
    // caller-supplied active set (e.g. the shifted set of the previous step of a closed-loop
    // chain): skip the PDIP, start the polish on it; the fp64 KKT check still certifies the result
    for (int r = tid; r < d.n; r += 256) {
      const int c = r % d.nu;
      int s = d.guess_all[(size_t)idx * d.n + r];
      if (s > 2) s = 0;
      const double xv = s == 1 ? d.ub_all[(size_t)idx * d.nu + c]
                      : s == 2 ? d.lb_all[(size_t)idx * d.nu + c] : (double)d.u[o + r];
      d.st[o + r] = (unsigned char)s;
      d.x[o + r] = xv;
      d.v64[o + r] = xv;
    }
  }
  if (tid == 0) {
    d.slot_prob[p] = idx;
    d.age[p] = 0;
    d.qscale[p] = fmaxf(1.f, qm);
    d.phase[p] = warm ? PH_POLISH : PH_INIT;
    d.f_factor[p] = d.f_solve[p] = 0;
    d.ipm_it[p] = d.nfac[p] = d.prounds[p] = d.rcnt[p] = d.psub[p] = d.fail[p] = d.stale[p] = 0;
    d.pninf[p] = 0x7fffffff; d.pgrace[p] = POLISH_GRACE; d.ptie[p] = 0;
    d.mu[p] = d.gap[p] = d.smu[p] = 0.f; d.rz[p] = 0.0;
  }
}
// Start of a segment: every slot is free.
__global__ void reset_slots_k(QpDev d) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < d.slots) {
    d.phase[p] = PH_DONE; d.slot_prob[p] = -1; d.age[p] = 0;
    d.f_factor[p] = d.f_solve[p] = 0;
  }
  if (p == 0) *d.next_prob = 0;
}

__device__ void write_outputs(const QpDev& d, int p, int status) {
  const int tid = threadIdx.x;
  const size_t o = (size_t)p * d.np;
  const size_t pi = (size_t)d.slot_prob[p];
  for (int r = tid; r < d.nout; r += 256) d.u_out[pi * d.ldu + r] = d.x[o + r];
  const int m = 2 * d.n;
  for (int w = tid; w < d.words; w += 256) {
    uint32_t bits = 0;
    for (int b = 0; b < 32; ++b) {
      const int idx = 32 * w + b;
      if (idx < m) {
        const int k = idx / (2 * d.nu), c = idx % (2 * d.nu);
        const int hit = (c < d.nu) ? (d.st[o + k * d.nu + c] == 1) : (d.st[o + k * d.nu + c - d.nu] == 2);
        bits |= (uint32_t)hit << b;
      }
    }
    if (d.act_out) d.act_out[pi * d.words + w] = bits;
  }
  if (tid == 0) {
    if (d.status_out) d.status_out[pi] = d.fail[p] ? NNMPC_ST_NUMERIC : status;
    if (d.iters_out) { d.iters_out[2 * pi] = d.ipm_it[p]; d.iters_out[2 * pi + 1] = d.nfac[p]; }
  }
}

// Preconditioned CG on the free block  P_FF x_F = -(q + P_FA x_A)_F  in f64, the
// f32 Cholesky of the masked, regularised P being the preconditioner M.
// cg_alpha: needs PX = P p.   x += a p, r -= a P p;  returns 1 if another
// preconditioner solve is wanted (rhs = r), 0 if converged (psub -> CHECK, v = x).
__device__ int cg_alpha(const QpDev& d, int p, double* shd) {
  const int tid = threadIdx.x;
  const size_t o = (size_t)p * d.np;
  double pap = 0.0;
  for (int r = tid; r < d.n; r += 256)
    if (d.st[o + r] == 0) pap += d.p64[o + r] * d.PX[o + r];
  pap = block_sumd(pap, shd);
  const double rz = d.rz[p];
  const bool ok = (pap > 0.0) && (rz > 0.0);
  const double a = ok ? rz / pap : 0.0;
  double pm = 0.0, xm = 0.0;
  for (int r = tid; r < d.n; r += 256) {
    if (d.st[o + r] == 0) {
      const double pv = d.p64[o + r];
      const double x = d.x[o + r] + a * pv;
      d.x[o + r] = x;
      d.r64[o + r] -= a * d.PX[o + r];
      pm = fmax(pm, fabs(pv));
      xm = fmax(xm, fabs(x));
    } else {
      xm = fmax(xm, fabs(d.x[o + r]));
    }
  }
  pm = block_maxd(pm, shd);
  xm = block_maxd(xm, shd);
  const int rc = d.rcnt[p] + 1;
  const bool nan = !(a == a) || !(xm == xm);
  const bool conv = !ok || nan || (fabs(a) * pm <= d.refine_tol * fmax(1.0, xm)) || rc >= d.max_refine;
  if (conv) {
    for (int r = tid; r < d.n; r += 256) d.v64[o + r] = d.x[o + r];
  } else {
    for (int r = tid; r < d.n; r += 256) d.rhs[o + r] = d.st[o + r] ? 0.f : (float)d.r64[o + r];
  }
  // PCG on a stale preconditioner that does not converge quickly: factor the current set instead
  const bool refac = !conv && d.stale[p] && rc >= d.stale_cg_limit;
  if (refac) for (int r = tid; r < d.n; r += 256) d.v64[o + r] = d.x[o + r];
  __syncthreads();
  if (tid == 0) {
    d.rcnt[p] = refac ? 0 : rc;
    if (nan) d.fail[p] = 1;
    if (conv) d.psub[p] = PS_CHECK;
    if (refac) { d.psub[p] = PS_START; d.stale[p] = 0; }
  }
  return (conv || refac) ? 0 : 1;
}
// cg_beta: needs sol = M^-1 r.   p = z + (r'z / rz_old) p,  v = p.
__device__ void cg_beta(const QpDev& d, int p, double* shd) {
  const int tid = threadIdx.x;
  const size_t o = (size_t)p * d.np;
  double rzn = 0.0;
  for (int r = tid; r < d.n; r += 256)
    if (d.st[o + r] == 0) rzn += d.r64[o + r] * (double)d.sol[o + r];
  rzn = block_sumd(rzn, shd);
  const bool first = d.psub[p] == PS_START;
  const double rz = d.rz[p];
  const double beta = (first || !(rz > 0.0)) ? 0.0 : rzn / rz;
  for (int r = tid; r < d.n; r += 256) {
    double pv = 0.0;
    if (d.st[o + r] == 0) pv = (double)d.sol[o + r] + beta * d.p64[o + r];
    d.p64[o + r] = pv;
    d.v64[o + r] = pv;
  }
  __syncthreads();
  if (tid == 0) { d.rz[p] = rzn; d.psub[p] = PS_CG; }
}

// fp64 KKT check of the refined point (PX = P x).  Returns 1 when the slot is finished
// (outputs written, phase DONE); otherwise the active set has been updated, x snapped to
// the new bounds, psub = START, v = x.
__device__ int kkt_check(const QpDev& d, int p, int* shi, double* shd) {
  const int tid = threadIdx.x, n = d.n;
  const size_t o = (size_t)p * d.np;
  int bad = 0;
  double gfree = 0.0;
  for (int r = tid; r < n; r += 256) {
    const int c = r % d.nu;
    const double g = d.PX[o + r] + d.q64[o + r], x = d.x[o + r];
    const double lb = d.lb64[(size_t)p * d.nu + c], ub = d.ub64[(size_t)p * d.nu + c];
    const int s = d.st[o + r];
    // (written so that a NaN fails every test)
    if (s == 0) { bad += !(x <= ub + d.bound_tol) || !(x >= lb - d.bound_tol) || !(fabs(g) <= 1.79e308); gfree = fmax(gfree, fabs(g)); }
    else if (s == 1) bad += !(g < 0.0);   // multiplier -g must be > 0
    else bad += !(g > 0.0);               // multiplier  g must be > 0
  }
  bad = block_sum_i(bad, shi);
  gfree = block_maxd(gfree, shd);
  const int pr = d.prounds[p] + 1;
  // Tie rule ("a bound is active iff its multiplier is > 0", as the oracle and the active-set path have it): a free variable that
  // sits beyond its bound by LESS than bound_tol passes the feasibility test above, but at the exact optimum that bound is active
  // with a tiny positive multiplier.  Once per problem, when everything else is settled, such bounds are made active and the set is
  // solved again; a multiplier that then comes out with the wrong sign drops the bound by the ordinary rule, and it is not
  // re-added (ptie): at most one extra polish round, only for problems that have such a tie.
  if (bad == 0 && pr <= d.max_polish && !d.ptie[p]) {
    int ties = 0;
    for (int r = tid; r < n; r += 256) {
      if (d.st[o + r] != 0) continue;
      const int c = r % d.nu;
      const double x = d.x[o + r];
      const double lb = d.lb64[(size_t)p * d.nu + c], ub = d.ub64[(size_t)p * d.nu + c];
      if (x > ub) { d.st[o + r] = 1; d.x[o + r] = ub; ++ties; }
      else if (x < lb) { d.st[o + r] = 2; d.x[o + r] = lb; ++ties; }
    }
    ties = block_sum_i(ties, shi);
    __syncthreads();
    if (tid == 0) d.ptie[p] = 1;
    if (ties > 0) {
      for (int r = tid; r < n; r += 256) d.v64[o + r] = d.x[o + r];
      if (tid == 0) {
        d.stale[p] = ties <= d.stale_max_changes;
        d.prounds[p] = pr; d.psub[p] = PS_START; d.rcnt[p] = 0;
        d.f_factor[p] = d.f_solve[p] = 0;
      }
      return 0;
    }
  }
  if (bad == 0 || pr > d.max_polish) {
    // stationarity of the free block certifies the refinement itself
    const double gs = d.pscale * (double)d.qscale[p];
    const int ok = bad == 0 && gfree <= d.stat_tol * gs;
    write_outputs(d, p, ok ? NNMPC_ST_OPTIMAL : NNMPC_ST_MAXITER);
    if (tid == 0) { d.phase[p] = PH_DONE; d.f_factor[p] = d.f_solve[p] = 0; }
    return 1;
  }
  // Exchange rule (block principal pivoting with single-exchange fallback, as in asm_update_k): every infeasible
  // index changes sides while their number keeps reaching new minima (POLISH_GRACE rounds of grace), after that only
  // the one with the SMALLEST index (Murty's least-index rule; earliest MPC stage first) -- the all-at-once rule can
  // cycle for ever on ill-conditioned Hessians.
  int single = 0, rsel = -1;
  {
    const int best = d.pninf[p], grace = d.pgrace[p];
    __syncthreads();
    if (bad < best) { if (tid == 0) { d.pninf[p] = bad; d.pgrace[p] = POLISH_GRACE; } }
    else if (grace > 0) { if (tid == 0) d.pgrace[p] = grace - 1; }
    else single = 1;
  }
  if (single) {
    double rm = -1e300;                                      // minus the smallest infeasible index
    for (int r = tid; r < n; r += 256) {
      const int c = r % d.nu;
      const double g = d.PX[o + r] + d.q64[o + r], x = d.x[o + r];
      const double lb = d.lb64[(size_t)p * d.nu + c], ub = d.ub64[(size_t)p * d.nu + c];
      const int s = d.st[o + r];
      const bool inf = s == 0 ? ((x > ub + d.bound_tol) || (x < lb - d.bound_tol)) : (s == 1 ? g >= 0.0 : g <= 0.0);
      if (inf && -(double)r > rm) rm = -(double)r;
    }
    rm = block_maxd(rm, shd);
    rsel = rm > -1e299 ? (int)(-rm) : -1;
  }
  for (int r = tid; r < n; r += 256) {
    const int c = r % d.nu;
    const double g = d.PX[o + r] + d.q64[o + r], x = d.x[o + r];
    const double lb = d.lb64[(size_t)p * d.nu + c], ub = d.ub64[(size_t)p * d.nu + c];
    const int s = d.st[o + r];
    double xn = x;
    if (single && r != rsel) { d.v64[o + r] = xn; continue; }
    if (s == 0) {
      if (x > ub + d.bound_tol) { d.st[o + r] = 1; xn = ub; }
      else if (x < lb - d.bound_tol) { d.st[o + r] = 2; xn = lb; }
    } else if ((s == 1 && g >= 0.0) || (s == 2 && g <= 0.0)) {
      d.st[o + r] = 0;
    }
    d.x[o + r] = xn;
    d.v64[o + r] = xn;
  }
  if (tid == 0) {
    // few changes: keep the previous factor as the PCG preconditioner (a stale SPD
    // preconditioner is still valid; ~1 extra CG step per changed bound, each ~20x
    // cheaper than a factorisation); cg_alpha falls back to a fresh factor if CG stalls
    d.stale[p] = bad <= d.stale_max_changes;
    d.prounds[p] = pr; d.psub[p] = PS_START; d.rcnt[p] = 0;
    d.f_factor[p] = d.f_solve[p] = 0;
  }
  return 0;
}

// Round stage 1: needs PU = P u (f32) for INIT/IPM slots, PX = P x (f64) for POLISH slots.
__global__ __launch_bounds__(256) void stage_pre_k(QpDev d) {
  __shared__ float shf[4];
  __shared__ int shi[4];
  __shared__ double shd[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  const size_t o = (size_t)p * d.np;
  int ph = d.phase[p];
  if (ph == PH_DONE) return;
  const int n = d.n;
  {
    const int age = d.age[p] + 1;   // rounds this problem has been resident
    __syncthreads();
    if (tid == 0) d.age[p] = age;
    if (age > d.max_rounds) {       // budget exhausted: emit the current iterate, uncertified
      if (ph != PH_POLISH) {
        for (int r = tid; r < n; r += 256) { d.x[o + r] = (double)d.u[o + r]; d.st[o + r] = 0; }
        __syncthreads();
      }
      write_outputs(d, p, NNMPC_ST_MAXITER);
      if (tid == 0) { d.phase[p] = PH_DONE; d.f_factor[p] = d.f_solve[p] = 0; }
      return;
    }
  }

  if (ph == PH_INIT) {
    float s = 0.f;
    for (int r = tid; r < n; r += 256) s += fabsf(d.PU[o + r] + d.q[o + r]);
    s = block_sum(s, shf);
    const float mu0 = 0.1f * s / n + 1e-3f;
    for (int r = tid; r < n; r += 256) {
      const float g = d.PU[o + r] + d.q[o + r];
      const float u = d.u[o + r];
      const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
      d.zu[o + r] = fmaxf(-g, 0.f) + mu0 / su;
      d.zl[o + r] = fmaxf(g, 0.f) + mu0 / sl;
    }
    ph = PH_IPM;
  }

  if (ph == PH_IPM) {
    float rdm = 0.f, gs = 0.f;
    for (int r = tid; r < n; r += 256) {
      const float g = d.PU[o + r] + d.q[o + r];
      const float u = d.u[o + r], zu = d.zu[o + r], zl = d.zl[o + r];
      const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
      const float rdv = g + zu - zl;
      d.rd[o + r] = rdv;
      rdm = fmaxf(rdm, fabsf(rdv));
      gs += su * zu + sl * zl;
    }
    rdm = block_max(rdm, shf);
    gs = block_sum(gs, shf);
    const float mu = gs / (2.f * n);
    const int it = d.ipm_it[p];
    const bool nan = !(rdm == rdm) || !(gs == gs);
    const bool conv = (rdm <= d.ipm_tol * d.qscale[p] && mu <= d.ipm_tol * d.qscale[p]) || it >= d.max_ipm || nan;
    if (!conv) {
      for (int r = tid; r < n; r += 256) {
        const float u = d.u[o + r], zu = d.zu[o + r], zl = d.zl[o + r];
        const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
        d.dvec[o + r] = zu / su + zl / sl + d.delta;
        d.mask[o + r] = 1.f;
        d.rhs[o + r] = -(d.PU[o + r] + d.q[o + r]);
      }
      if (tid == 0) {
        d.phase[p] = PH_IPM;
        d.gap[p] = gs; d.mu[p] = mu;
        d.ipm_it[p] = it + 1; d.nfac[p] += 1;
        d.f_factor[p] = 1; d.f_solve[p] = 1; d.istep[p] = 0;
        atomicAdd(&d.counters[CNT_ACTIVE], 1); atomicAdd(&d.counters[CNT_FACTOR], 1);
        atomicAdd(&d.counters[CNT_IPM], 1); atomicAdd(&d.counters[CNT_SOLVE], 1);
      }
      return;
    }
    // ---- PDIP done: guess the active set, start the polish next round
    for (int r = tid; r < n; r += 256) {
      const float u = d.u[o + r], zu = d.zu[o + r], zl = d.zl[o + r];
      const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
      const int c = r % d.nu;
      int s = 0;
      if (!nan) {
        const float pd = d.pdiag[r];
        const bool au = zu > pd * su, al = zl > pd * sl;
        if (au && al) s = (zu * sl > zl * su) ? 1 : 2; else if (au) s = 1; else if (al) s = 2;
      }
      d.st[o + r] = (unsigned char)s;
      const double xv = s == 1 ? d.ub64[(size_t)p * d.nu + c]
                      : s == 2 ? d.lb64[(size_t)p * d.nu + c]
                               : (nan ? 0.0 : (double)u);
      d.x[o + r] = xv;
      d.v64[o + r] = xv;
    }
    if (tid == 0) {
      if (nan) d.fail[p] = 1;
      d.phase[p] = PH_POLISH; d.psub[p] = PS_START; d.rcnt[p] = 0; d.prounds[p] = 0;
      d.f_factor[p] = d.f_solve[p] = 0;
      atomicAdd(&d.counters[CNT_ACTIVE], 1); atomicAdd(&d.counters[CNT_POLISH], 1);
    }
    return;
  }

  // ---- PH_POLISH (PX = P64 * v64, v = x in START / CHECK, v = p in CG)
  int sub = d.psub[p];
  if (sub == PS_CHECK) {
    if (kkt_check(d, p, shi, shd)) return;          // DONE (outputs written)
    if (tid == 0) { atomicAdd(&d.counters[CNT_ACTIVE], 1); atomicAdd(&d.counters[CNT_POLISH], 1); }
    return;                                         // new active set: P x is stale, restart next round
  }
  if (sub == PS_START) {
    // new active set: r = -(P x + q)_F, z = M^-1 r with M = f32 Cholesky of the masked
    // (regularised) P -- freshly factored, or the previous one when only few bounds changed
    const int keep = d.stale[p];
    for (int r = tid; r < n; r += 256) {
      const int s = d.st[o + r];
      const double rr = s ? 0.0 : -(d.PX[o + r] + d.q64[o + r]);
      d.r64[o + r] = rr;
      d.rhs[o + r] = (float)rr;
      if (!keep) { d.mask[o + r] = s ? 0.f : 1.f; d.dvec[o + r] = s ? 1.f : d.delta; }
    }
    if (tid == 0) {
      d.f_factor[p] = !keep; d.f_solve[p] = 1;
      if (!keep) { d.nfac[p] += 1; atomicAdd(&d.counters[CNT_FACTOR], 1); }
      atomicAdd(&d.counters[CNT_ACTIVE], 1); atomicAdd(&d.counters[CNT_POLISH], 1);
      atomicAdd(&d.counters[CNT_SOLVE], 1);
    }
    return;
  }
  // PS_CG carried over from the previous round: the sub-step loop always stops after part B,
  // i.e. the slot is waiting for the next preconditioner solve (rhs = r is already in place).
  if (tid == 0) {
    d.f_factor[p] = 0; d.f_solve[p] = 1;
    atomicAdd(&d.counters[CNT_ACTIVE], 1); atomicAdd(&d.counters[CNT_POLISH], 1);
    atomicAdd(&d.counters[CNT_SOLVE], 1);
  }
}

// Solve sub-step, part A (after a triangular solve): sol = K^-1 rhs is available.
//   IPM  step 0: affine direction -> sigma -> corrector rhs (another solve follows)
//        step 1: combined direction -> step length -> new iterate (done for this round)
//   POLISH: z = sol -> new CG direction p, v = p (P p is computed next)
__global__ __launch_bounds__(256) void stage_sub_a_k(QpDev d) {
  __shared__ float shf[4];
  __shared__ double shd[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  const size_t o = (size_t)p * d.np;
  if (!d.f_solve[p]) return;
  const int ph = d.phase[p], n = d.n;
  if (ph == PH_POLISH) { cg_beta(d, p, shd); return; }
  if (ph != PH_IPM) return;
  if (d.istep[p] == 0) {
  // IPM: affine (predictor) direction -> sigma -> corrector right-hand side
  float t = 0.f;
  for (int r = tid; r < n; r += 256) {
    const float du = d.sol[o + r], u = d.u[o + r], zu = d.zu[o + r], zl = d.zl[o + r];
    const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
    const float dzu = -zu + zu * du / su, dzl = -zl - zl * du / sl;
    t = fmaxf(t, fmaxf(fmaxf(du / su, -du / sl), fmaxf(-dzu / zu, -dzl / zl)));
  }
  t = block_max(t, shf);
  const float a = t > 0.f ? fminf(1.f, 1.f / t) : 1.f;
  float ga = 0.f;
  for (int r = tid; r < n; r += 256) {
    const float du = d.sol[o + r], u = d.u[o + r], zu = d.zu[o + r], zl = d.zl[o + r];
    const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
    const float dzu = -zu + zu * du / su, dzl = -zl - zl * du / sl;
    ga += (su - a * du) * (zu + a * dzu) + (sl + a * du) * (zl + a * dzl);
  }
  ga = block_sum(ga, shf);
  float sg = ga / d.gap[p];
  sg = fminf(1.f, fmaxf(0.f, sg));
  const float smu = sg * sg * sg * d.mu[p];
  for (int r = tid; r < n; r += 256) {
    const float du = d.sol[o + r], u = d.u[o + r], zu = d.zu[o + r], zl = d.zl[o + r];
    const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
    const float dzu = -zu + zu * du / su, dzl = -zl - zl * du / sl;
    d.rhs[o + r] = -d.rd[o + r] + zu - smu / su - du * dzu / su - zl + smu / sl - du * dzl / sl;
    d.dua[o + r] = du;
  }
  if (tid == 0) { d.smu[p] = smu; d.istep[p] = 1; }
    return;
  }
  const float smu = d.smu[p];
  float t = 0.f;
  for (int r = tid; r < n; r += 256) {
    const float du = d.sol[o + r], da = d.dua[o + r], u = d.u[o + r], zu = d.zu[o + r], zl = d.zl[o + r];
    const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
    const float dzua = -zu + zu * da / su, dzla = -zl - zl * da / sl;
    const float rcu = -su * zu + smu + da * dzua, rcl = -sl * zl + smu - da * dzla;
    const float dzu = (rcu + zu * du) / su, dzl = (rcl - zl * du) / sl;
    t = fmaxf(t, fmaxf(fmaxf(du / su, -du / sl), fmaxf(-dzu / zu, -dzl / zl)));
  }
  t = block_max(t, shf);
  const float a = t > 0.f ? fminf(1.f, 0.99f / t) : 1.f;
  for (int r = tid; r < n; r += 256) {
    const float du = d.sol[o + r], da = d.dua[o + r], u = d.u[o + r], zu = d.zu[o + r], zl = d.zl[o + r];
    const float su = fmaxf(d.ubv[o + r] - u, SLACK_MIN), sl = fmaxf(u - d.lbv[o + r], SLACK_MIN);
    const float dzua = -zu + zu * da / su, dzla = -zl - zl * da / sl;
    const float rcu = -su * zu + smu + da * dzua, rcl = -sl * zl + smu - da * dzla;
    const float dzu = (rcu + zu * du) / su, dzl = (rcl - zl * du) / sl;
    d.u[o + r] = u + a * du;
    d.zu[o + r] = fmaxf(zu + a * dzu, 1e-30f);
    d.zl[o + r] = fmaxf(zl + a * dzl, 1e-30f);
  }
  if (tid == 0) d.f_solve[p] = 0;
}

// Solve sub-step, part B (after PX = P v):  POLISH slots only.
//   CG    : x += a p, r -= a P p; converged -> psub = CHECK, v = x (P x is computed next)
//   CHECK : (P x fresh) fp64 KKT check -> DONE, or new active set for the next round
__global__ __launch_bounds__(256) void stage_sub_b_k(QpDev d) {
  __shared__ int shi[4];
  __shared__ double shd[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  if (d.phase[p] != PH_POLISH) return;
  const int sub = d.psub[p];
  if (sub == PS_CG && d.f_solve[p]) {
    const int go = cg_alpha(d, p, shd);
    if (tid == 0) d.f_solve[p] = go;
  } else if (sub == PS_CHECK) {
    kkt_check(d, p, shi, shd);
  }
}

__global__ void f64_to_f32_k(float* dst, const double* src, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) dst[i] = (float)src[i];
}
// x0 [nprob][n_aug] f64 -> padded [rows][ka] f64 and f32
__global__ void pad_x0_k(double* d64, float* d32, const double* src, int nprob, int n_aug, int ka,
                         int rows) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)rows * ka;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int p = (int)(i / ka), k = (int)(i % ka);
    const double v = (p < nprob && k < n_aug) ? src[(size_t)p * n_aug + k] : 0.0;
    d64[i] = v;
    d32[i] = (float)v;
  }
}

}  // namespace

// ---------------------------------------------------------------------------
struct nnmpc_qp {
  int device;
  int n, np, nu, n_aug, ka, NB, T, tiles, slots, words;
  nnmpc_qp_opts opts;
  bool have_kunc;
  // shared problem data
  float* Pt;      // tile-packed lower, f32
  float* P32;     // full np x np, f32
  double* P64;    // full np x np, f64
  double* tq64;   // np x ka
  float* Kunc32;  // np x ka
  float* pdiag;   // np, diagonal of P / pscale
  double pscale;  // median diag(P): the f32 PDIP works on P / pscale, q / pscale
  // per-slot workspace
  float* L;
  float* Y;
  float* Dacc;
  unsigned long long* trsv_count;
  QpDev d;
  // shared-inverse active-set path (qp_asm.h)
  bool have_inverse;
  double* H64;      // np x np
  double* Kunc64;   // np x ka
  double *asm_xunc, *asm_x, *asm_lam, *asm_xh, *asm_scratch, *asm_xhw;
  int *asm_rowprob, *asm_wflag;
  unsigned char* asm_wmark;
  unsigned char* asm_st;
  int *asm_state, *asm_rounds, *asm_counters, *asm_biglist, *asm_status, *asm_binlist, *asm_idxg, *asm_mg, *asm_row, *asm_lrank, *asm_ctot;
  unsigned char *asm_prec, *asm_redo, *asm_alpha, *asm_rowk;
  float *asm_lam32, *asm_xh32, *H32;
  int *asm_ninf, *asm_hi, *asm_kblk;
  double tqmax;         // max |tq| entry
  int asm_pool;
  double asm_e1max, asm_e2max;
  // far-field factorisations of the full-width pass (nnmpc_qp_set_farfield), one per window
  struct Far { int W, r, rp; double *U, *Vx, *Vl, *cu; int* kt; double ksum; double efar; };   // ksum: sum of kt over the column tiles
  std::vector<Far> far;
  std::vector<int> far_missing;   // windows of full-width passes that ran in the dense form for want of factors (handed out once each)
  double p_inf = 0.0;       // max row sum of |P|
  double *asm_tnorm = nullptr, *asm_tslack = nullptr;
  // first-set predictor (qp_predict.h): bf16 fragments of Pinv[0:512, 0:512], step sizes; set by nnmpc_qp_set_inverse when the problem is large enough
  struct PredWin { pu32x4* Hf = nullptr; double L = 0.0; int W = 0; };   // L = 1.05 lambda_max(D^-1/2 Pinv_WW D^-1/2); 0: window not available
  PredWin pred[2];              // [0]: 512 columns, 64 problems per workgroup; [1]: 1024 columns, 32 per workgroup (sets that reach further)
  int* pred_cnt = nullptr;      // device counters: [0] asm_extent_k, [1] sum of the workgroups' iterations after the first (asm_predict_k)
  double* asm_work;
  int seg_max;          // problems per segment (q / warm start precomputed per segment)
  double* x0_64;        // [seg_max][ka]
  float* x0_32;
  double* q64_all;      // [seg_max][np]
  float* uunc_all;      // [seg_max][np]
  double *lb_d, *ub_d;  // [seg_max][nu] staging for host inputs
  double* in_stage;     // [seg_max][n_aug]
  hipStream_t stream;
  hipStream_t stream2 = nullptr;     // side streams of the active-set rounds (large-set kernels: the families do not wait for each other)
  hipStream_t stream3 = nullptr, stream4 = nullptr;
  int asm_tail_budget = 50000;       // iterations of asm_tail_k per problem (set in nnmpc_qp_create)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr, ev_join4 = nullptr;
  // pinned host copies of the active-set pass's read-backs (the round counters, the status of a segment): into pageable memory a
  // "hipMemcpyAsync" is staged and waited for inside the runtime (a blocking wait); pinned, the copy is a packet on the stream and
  // stream_sync polls for it.
  int* pin_cnt = nullptr;            // [ASM_NCNT + 4]: the round counters, then the predictor's iteration sum
  int* pin_st = nullptr;             // [seg_max]
  bool profiling;
  bool gemm_error = false;      // a GEMM was asked for in a form no kernel implements (gemm64): the call in progress fails
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used;
  struct EvRec { int kind; size_t e0, e1; double flops; };
  std::vector<EvRec> ev_recs;
  nnmpc_qp_stats stats;
  std::vector<void*> allocs;
  // grow-only scratch owned by the handle: staging of host-pointer calls, compacted copies for the PDIP fallback
  struct Scratch { void* p = nullptr; size_t cap = 0; };
  enum { SC_U = 0, SC_ACT, SC_ST, SC_IT, SC_GUESS, SC_FB_GUESS, SC_FB_LIST, SC_FB_X0, SC_FB_LB, SC_FB_UB, SC_FB_U, SC_FB_ACT, SC_FB_ST, SC_FB_IT, SC_COUNT };
  Scratch sc[SC_COUNT];
  int ldu, nout;        // layout of the caller's u buffer for the call in progress (nnmpc_qp_solve_batch_ex)
};

namespace {

template <class T>
int dev_alloc(nnmpc_qp* h, T** p, size_t count) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, count * sizeof(T));
  if (e != hipSuccess) { set_error("hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e)); return NNMPC_ENOMEM; }
  e = hipMemset(q, 0, count * sizeof(T));
  if (e != hipSuccess) { set_error("hipMemset: %s", hipGetErrorString(e)); return NNMPC_EHIP; }
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

// scratch buffer `which` of at least `bytes` (contents undefined); grown buffers replace the old ones
template <class T>
int scratch_get(nnmpc_qp* h, int which, T** out, size_t bytes) {
  nnmpc_qp::Scratch& s = h->sc[which];
  if (s.cap < bytes) {
    if (s.p) { hipFree(s.p); s.p = nullptr; s.cap = 0; }
    const size_t want = bytes + bytes / 4 + 256;
    const hipError_t e = hipMalloc(&s.p, want);
    if (e != hipSuccess) { s.p = nullptr; set_error("hipMalloc(%zu bytes of scratch): %s", want, hipGetErrorString(e)); return NNMPC_ENOMEM; }
    s.cap = want;
  }
  *out = (T*)s.p;
  return 0;
}

size_t ev_get(nnmpc_qp* h) {
  if (h->ev_used == h->ev_pool.size()) {
    hipEvent_t e;
    hipEventCreate(&e);
    h->ev_pool.push_back(e);
  }
  return h->ev_used++;
}
struct EvScope {
  nnmpc_qp* h; int kind; double flops; size_t e0; hipStream_t st;
  EvScope(nnmpc_qp* h_, int kind_, double flops_, hipStream_t st_ = nullptr) : h(h_), kind(kind_), flops(flops_), e0(0), st(st_ ? st_ : h_->stream) {
    if (h->profiling) { e0 = ev_get(h); hipEventRecord(h->ev_pool[e0], st); }
  }
  ~EvScope() {
    if (h->profiling) {
      size_t e1 = ev_get(h);
      hipEventRecord(h->ev_pool[e1], st);
      h->ev_recs.push_back({kind, e0, e1, flops});
    }
  }
};
void ev_collect(nnmpc_qp* h) {
  for (auto& r : h->ev_recs) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, h->ev_pool[r.e0], h->ev_pool[r.e1]);
    if (r.kind == 0) { h->stats.panel_ms += ms; h->stats.panel_launches += 1; h->stats.panel_flops += r.flops; }
    else if (r.kind == 1) h->stats.diag_ms += ms;
    else if (r.kind == 2) h->stats.trsv_ms += ms;
    else if (r.kind == 3) h->stats.total_ms += ms;
    else if (r.kind == 4) h->stats.asm_lambda_ms += ms;
    else if (r.kind == 5) { h->stats.asm_gemm_ms += ms; h->stats.asm_gemm_flops += r.flops; h->stats.asm_gemm_launches += 1; }
    else if (r.kind == 6) h->stats.asm_update_ms += ms;
    else if (r.kind == 7) { h->stats.asm_lambda32_ms += ms; h->stats.asm_lambda32_launches += 1; }
    else if (r.kind == 8) { h->stats.asm_lambda64_ms += ms; h->stats.asm_lambda64_launches += 1; }
    else if (r.kind == 9) h->stats.asm_side_ms += ms;
    else if (r.kind == 10) { h->stats.asm_predict_ms += ms; h->stats.asm_predict_flops += r.flops; }   // (launches: counted at the launch, profiling or not)
  }
  h->ev_recs.clear();
  h->ev_used = 0;
}

template <int NB>
void launch_gemm32(hipStream_t s, float* C, size_t ldc, const float* A, size_t lda, const float* B,
                   size_t ldb, int M, int N, int K) {
  dim3 grid(N / NB, M / NB);
  hipLaunchKernelGGL((gemm_nt_f32_k<NB, false, false>), grid, dim3(256), TileCfg<NB>::LDS_FLOATS * 4, s,
                     C, ldc, A, lda, B, ldb, K, (const float*)nullptr);
}
void gemm32(nnmpc_qp* h, float* C, size_t ldc, const float* A, size_t lda, const float* B, size_t ldb,
            int M, int N, int K) {
  if (N % 128 == 0 && M % 128 == 0) launch_gemm32<128>(h->stream, C, ldc, A, lda, B, ldb, M, N, K);
  else launch_gemm32<64>(h->stream, C, ldc, A, lda, B, ldb, M, N, K);
}
void gemm64(nnmpc_qp* h, double* C, size_t ldc, const double* A, size_t lda, const double* B,
            size_t ldb, int M, int N, int K, const int* rowphase = nullptr, int want = 0,
            const int* kdyn = nullptr, const int* mdyn = nullptr, bool kblocks = false, const int* rowmap = nullptr) {
  // kblocks: kdyn holds one bound per 64 rows of A (else one for the launch); rowmap: gather / scatter of the rows (gemm64.h)
  // (fewer 128 x 128 tiles than CUs -- the x_unc product of a chain step: 149 rows -- leave most of the chip idle behind a few
  // long K loops: four times as many 64 x 64 tiles finish sooner)
  static const bool small64 = getenv("NNMPC_NO_SMALL_GEMM") == nullptr;   // (the variable: A/B)
  const bool few = small64 && !rowmap && (M / 128) * (N / 128) < 200;
  if (M % 128 == 0 && N % 128 == 0 && K % G64_KC == 0 && !few) {
    const int ntm = M / 128, ntn = N / 128;
    hipLaunchKernelGGL(gemm_nt_f64_t128_k, dim3(g64_grid(ntm, ntn)), dim3(256), G64_LDS, h->stream, C, ldc, A, lda, B, ldb, K, ntm, ntn,
                       rowphase, want, kdyn, kblocks ? 2 : 0, mdyn, rowmap);
    return;
  }
  if (rowmap) {
    // the 64 x 64 kernel has no gather / scatter of the rows: running it would write the rows of C to the wrong problems.  The one
    // caller with a rowmap (the device tail's x_unc rows) only exists on the 128-grid (`lazy`); anything else is a bug -- say so.
    set_error("gemm64: rowmap with a shape off the 128 x 128 / K %% 16 grid (M = %d, N = %d, K = %d)", M, N, K);
    h->gemm_error = true;
    return;
  }
  dim3 grid(N / 64, M / 64);
  hipLaunchKernelGGL(gemm_nt_f64_k, grid, dim3(256), 0, h->stream, C, ldc, A, lda, B, ldb, K, rowphase, want, kdyn, mdyn, kblocks ? 1 : 0);
}

template <int NB>
int set_lds_attrs() {
  hipError_t e;
  e = hipFuncSetAttribute((const void*)chol_diag_k<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, chol_diag_lds_bytes<NB>());
  if (e != hipSuccess) return -1;
  e = hipFuncSetAttribute((const void*)chol_panel_k<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, chol_panel_lds_bytes<NB>());
  if (e != hipSuccess) return -1;
  e = hipFuncSetAttribute((const void*)trsv_k<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  if (e != hipSuccess) return -1;
  e = hipFuncSetAttribute((const void*)gemm_nt_f32_k<NB, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg<NB>::LDS_FLOATS * 4);
  if (e != hipSuccess) return -1;
  return 0;
}

template <int NB>
void factor_all(nnmpc_qp* h, int nslots, int nfactor) {
  CholArgs a;
  a.n = h->n; a.np = h->np; a.T = h->T; a.tiles = h->tiles;
  a.Pt = h->Pt; a.L = h->L; a.Y = h->Y; a.Dacc = h->Dacc; a.dvec = h->d.dvec; a.mask = h->d.mask;
  a.flag = h->d.f_factor; a.fail = h->d.fail;
  const double nb = NB;
  for (int j = 0; j < h->T; ++j) {
    {
      EvScope es(h, 1, 0.0);
      hipLaunchKernelGGL((chol_diag_k<NB>), dim3(nslots), dim3(256), chol_diag_lds_bytes<NB>(), h->stream, a, j);
    }
    const int below = h->T - 1 - j;
    if (below > 0) {
      // algorithmic flops of this launch: per tile 2 NB^2 (j NB) update + NB^3 triangular solve
      // + NB^3 symmetric rank-NB update of the diagonal tile
      const double fl = (double)nfactor * below * (2.0 * nb * nb * (j * nb) + nb * nb * nb + nb * nb * nb);
      EvScope es(h, 0, fl);
      hipLaunchKernelGGL((chol_panel_k<NB>), dim3(below, nslots), dim3(256), chol_panel_lds_bytes<NB>(), h->stream, a, j);
    }
  }
}
template <int NB>
void solve_all(nnmpc_qp* h, int nslots, const int* flag) {
  TrsvArgs a;
  a.n = h->n; a.np = h->np; a.T = h->T; a.tiles = h->tiles;
  a.L = h->L; a.Y = h->Y; a.rhs = h->d.rhs; a.sol = h->d.sol; a.flag = flag; a.count = h->trsv_count;
  EvScope es(h, 2, 0.0);
  hipLaunchKernelGGL((trsv_k<NB>), dim3(nslots), dim3(256), (h->np + 5 * NB) * 4, h->stream, a);
}
void factor_dispatch(nnmpc_qp* h, int nslots, int nfactor) {
  if (h->NB == 128) factor_all<128>(h, nslots, nfactor); else factor_all<64>(h, nslots, nfactor);
}
void solve_dispatch(nnmpc_qp* h, int nslots, const int* flag) {
  if (h->NB == 128) solve_all<128>(h, nslots, flag); else solve_all<64>(h, nslots, flag);
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e_)); return NNMPC_EHIP; } } while (0)

// One segment (<= h->seg_max problems): q and the warm start for the whole
// segment by two GEMMs, then lock-step rounds over the resident slots with
// finished slots refilled from the segment until it is exhausted.
int solve_segment(nnmpc_qp* h, int nprob, const double* x0_dev, const double* lb_dev, const double* ub_dev,
                  const unsigned char* guess_dev, double* u_dev, uint32_t* act_dev, int32_t* st_dev, int32_t* it_dev) {
  QpDev& d = h->d;
  hipStream_t s = h->stream;
  const int rows = h->slots;
  const int segp = ((nprob + 127) / 128) * 128;
  hipLaunchKernelGGL(pad_x0_k, dim3(512), dim3(256), 0, s, h->x0_64, h->x0_32, x0_dev, nprob, h->n_aug, h->ka, segp);
  gemm64(h, h->q64_all, h->np, h->x0_64, h->ka, h->tq64, h->ka, segp, h->np, h->ka);
  if (h->have_kunc) gemm32(h, h->uunc_all, h->np, h->x0_32, h->ka, h->Kunc32, h->ka, segp, h->np, h->ka);
  else HIPCHK(hipMemsetAsync(h->uunc_all, 0, (size_t)segp * h->np * sizeof(float), s));
  d.q64_all = h->q64_all; d.uunc_all = h->uunc_all; d.lb_all = lb_dev; d.ub_all = ub_dev;
  d.seg_count = nprob;
  d.guess_all = guess_dev;
  d.u_out = u_dev; d.act_out = act_dev; d.status_out = st_dev; d.iters_out = it_dev;
  d.ldu = h->ldu; d.nout = h->nout;
  hipLaunchKernelGGL(reset_slots_k, dim3((rows + 255) / 256), dim3(256), 0, s, d);

  int cnt[8];
  bool any_ipm = true, any_polish = guess_dev != nullptr;
  int issued = 0;  // problems handed to slots so far (host mirror of *next_prob, upper bound)
  const int hard_cap = 1000000;
  for (int round = 0; round < hard_cap; ++round) {
    HIPCHK(hipMemsetAsync(d.counters, 0, 8 * sizeof(int), s));
    if (issued < nprob) {
      hipLaunchKernelGGL(refill_k, dim3(rows), dim3(256), 0, s, d);
      any_ipm = true;
      if (guess_dev) any_polish = true;
    }
    if (any_ipm) gemm32(h, d.PU, h->np, d.u, h->np, h->P32, h->np, rows, h->np, h->np);
    if (any_polish) gemm64(h, d.PX, h->np, d.v64, h->np, h->P64, h->np, rows, h->np, h->np, d.phase, PH_POLISH);
    hipLaunchKernelGGL(stage_pre_k, dim3(rows), dim3(256), 0, s, d);
    HIPCHK(hipMemcpyAsync(cnt, d.counters, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(&issued, d.next_prob, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(stream_sync(s));
    if (cnt[CNT_ACTIVE] == 0 && issued >= nprob) break;
    any_ipm = cnt[CNT_IPM] > 0;
    any_polish = cnt[CNT_POLISH] > 0;
    h->stats.rounds += 1;
    h->stats.factorizations += cnt[CNT_FACTOR];
    h->stats.ipm_iterations += cnt[CNT_IPM];
    if (cnt[CNT_FACTOR] > 0) factor_dispatch(h, rows, cnt[CNT_FACTOR]);
    if (cnt[CNT_SOLVE] > 0) {
      // solve sub-steps: the two PDIP solves, and PCG steps + KKT check of polishing slots, all
      // inside the round of their factorisation (slots not concerned exit at once)
      const int nsub = cnt[CNT_POLISH] > 0 ? h->opts.sub_steps : 2;
      for (int k = 0; k < nsub; ++k) {
        solve_dispatch(h, rows, d.f_solve);
        hipLaunchKernelGGL(stage_sub_a_k, dim3(rows), dim3(256), 0, s, d);
        if (cnt[CNT_POLISH] > 0) {
          gemm64(h, d.PX, h->np, d.v64, h->np, h->P64, h->np, rows, h->np, h->np, d.phase, PH_POLISH);
          hipLaunchKernelGGL(stage_sub_b_k, dim3(rows), dim3(256), 0, s, d);
        }
      }
    }
  }
  HIPCHK(stream_sync(s));
  HIPCHK(hipGetLastError());
  h->stats.problems += nprob;
  return 0;
}


// One window of the first-set predictor (qp_predict.h): bf16 MFMA fragments of H' = Pinv[0:W, 0:W] diag(t), t_k = 1 / (L Pinv_kk),
// L = lambda_max(D^-1/2 Pinv_WW D^-1/2) (power iteration, 5 % margin: a step that is a little short costs nothing but speed).
// hh: the padded inverse (np x np, host).  Leaves pw.L = 0 when the window does not apply.
template <int NT>
int build_predictor(nnmpc_qp* h, const std::vector<double>& hh, int np, nnmpc_qp::PredWin& pw) {
  constexpr int W = PredCfg<NT>::W, KS = PredCfg<NT>::KS;
  pw.L = 0.0; pw.W = W;
  std::vector<double> dg(W);
  for (int j = 0; j < W; ++j) { dg[j] = hh[(size_t)j * np + j]; if (!(dg[j] > 0.0)) return 0; }
  std::vector<double> v(W, 1.0), u(W), isd(W);
  for (int j = 0; j < W; ++j) isd[j] = 1.0 / std::sqrt(dg[j]);
  double lam = 0.0;
  for (int itp = 0; itp < 200; ++itp) {
    double nv = 0.0;
    for (int i = 0; i < W; ++i) nv += v[i] * v[i];
    nv = std::sqrt(nv);
    for (int i = 0; i < W; ++i) v[i] *= isd[i] / nv;
    for (int i = 0; i < W; ++i) {
      double sacc = 0.0;
      const double* hr = &hh[(size_t)i * np];
      for (int j = 0; j < W; ++j) sacc += hr[j] * v[j];
      u[i] = sacc * isd[i];
    }
    double num = 0.0;
    for (int i = 0; i < W; ++i) num += u[i] * v[i] / isd[i];
    const bool conv = std::fabs(num - lam) <= 1e-6 * std::fabs(num);
    lam = num;
    v = u;
    if (conv && itp > 8) break;
  }
  if (!(lam > 0.0)) return 0;
  const double L = 1.05 * lam;
  std::vector<unsigned short> hf((size_t)W * W);
  auto bf = [](double x) { float f = (float)x; unsigned u32; memcpy(&u32, &f, 4); return (unsigned short)((u32 + 0x7fffu + ((u32 >> 16) & 1u)) >> 16); };
  for (int jt = 0; jt < W / 16; ++jt)
    for (int ks = 0; ks < KS; ++ks)
      for (int lane = 0; lane < 64; ++lane) {
        const int li = lane & 15, lq = lane >> 4;
        for (int e = 0; e < 8; ++e) {
          const int k = 32 * ks + 8 * lq + e;
          hf[pred_frag_index<NT>(jt, ks, lane) * 8 + e] = bf(hh[(size_t)(16 * jt + li) * np + k] / (L * dg[k]));   // H' = H diag(t)
        }
      }
  if (!pw.Hf) { int rc = dev_alloc(h, (unsigned short**)&pw.Hf, hf.size()); if (rc) return rc; }
  HIPCHK(hipMemcpy(pw.Hf, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  pw.L = L;
  return 0;
}

// Shared-inverse active-set pass over one segment; problems it cannot finish are marked 3 in
// h->asm_status and re-solved by the PDIP path (solve_segment) on a compacted copy.
int solve_segment_asm(nnmpc_qp* h, int nprob, const double* x0_dev, const double* lb_dev, const double* ub_dev,
                      const unsigned char* guess_dev, double* u_dev, uint32_t* act_dev, int32_t* st_dev, int32_t* it_dev) {
  hipStream_t s = h->stream;
  const int segp = ((nprob + 127) / 128) * 128;
  hipLaunchKernelGGL(pad_x0_k, dim3(512), dim3(256), 0, s, h->x0_64, h->x0_32, x0_dev, nprob, h->n_aug, h->ka, segp);
  static const bool no_fuse = getenv("NNMPC_NO_FUSED_WIDE") != nullptr;   // diagnostics: A/B of the fused epilogue
  static const bool no_lazy = getenv("NNMPC_NO_LAZY_XUNC") != nullptr;    // diagnostics: x_unc for all columns up front (round 2)
  const bool tail_only = h->opts.asm_tail_batch >= 0 && nprob <= std::min(std::min(h->asm_pool, 256), h->opts.asm_tail_batch ? h->opts.asm_tail_batch : 256);
  // x_unc = Kunc x0 -- for the leading columns only (the window the first sets are drawn from; extended should a round's
  // window outgrow it): beyond them x_unc is one K segment of the full-width pass's GEMM (qp_wide.h).  Calls that go to
  // the device tail from the start, and shapes the fused kernel's tiles do not fit, form all of it.  q = tq x0 is only
  // formed for the rows that need the full check with P.
  // Small problems -- Pinv fits the 4 MB L2 of an XCD (n <= 724: the CSTRs size, 540) -- run the whole iteration in one wave each
  // (qp_small.h); what that kernel hands back (sets beyond 64 bounds, problems still moving after its budget) carries on below.
  const bool small = getenv("NNMPC_NO_SMALL") == nullptr && (size_t)h->np * h->np * 8 <= (4u << 20) && h->n <= ASM_SM_NMAX && h->np % 4 == 0;   // (the variable: A/B, tests of the rounds at small sizes)
  const bool lazy = h->np % 128 == 0 && !no_fuse && !no_lazy && !tail_only && !small && (uint64_t)h->seg_max * h->ka * 8 < (1ull << 32);
  // (the window the first sets are drawn from: the leading eighth of the horizon, 512 columns at least -- a quarter until round 3;
  // the CDU batch settles inside 512 columns, a quarter is 1152: 0.4 ms of x_unc GEMM per step.  A bound x_unc violates further
  // out joins through the full-width pass every problem goes through.  Following the previous call's window instead made the
  // path of a call depend on the handle's history: other windows, other far-field factors, and for degenerate problems other sets)
  const int winit = std::min(h->n, std::max(512, ((h->n / 8 + 127) / 128) * 128));
  int Wx = h->np;
  if (lazy) Wx = std::min(h->np, guess_dev ? 512 : ((winit + 127) / 128) * 128);

  {
    EvScope es(h, 5, 2.0 * Wx * (double)h->ka * nprob);
    gemm64(h, h->asm_xunc, h->np, h->x0_64, h->ka, h->Kunc64, h->ka, segp, Wx, h->ka);
  }
  AsmDev a;
  a.Kunc = h->Kunc64; a.Wx = Wx; a.winit = winit;
  a.ffU = a.ffVx = a.ffVl = a.ffcu = nullptr; a.ffk = nullptr; a.ffr = a.ffW = 0; a.T = h->asm_xhw; a.tnorm = h->asm_tnorm; a.tslack = h->asm_tslack;
  a.ff_skip = 0; a.ff_err = 0.0; a.ff_efar = 0.0;
  static const bool no_far = getenv("NNMPC_NO_FARFIELD") != nullptr;      // diagnostics: dense form of the full-width pass (A/B)
  int wide_far_rp = 0, fft_prev = 0;
  double wide_far_ksum = 0.0;                      // the last full-width pass ran in the far-field form with this padded rank
  a.n = h->n; a.np = h->np; a.nu = h->nu; a.nseg = nprob;
  a.max_active = h->opts.asm_max_active; a.max_rounds = h->opts.asm_max_rounds;
  a.bound_tol = h->opts.bound_tol; a.stat_tol = 1e-8; a.pscale = h->pscale;
  a.e1max = h->asm_e1max; a.e2max = h->asm_e2max; a.x0 = h->x0_64; a.ka = h->ka;
  a.H = h->H64; a.lb = lb_dev; a.ub = ub_dev; a.xunc = h->asm_xunc; a.q64 = h->q64_all;
  a.x = h->asm_x; a.lam = h->asm_lam; a.xh = h->asm_xh; a.px = h->asm_xh;
  a.st = h->asm_st; a.guess = guess_dev; a.state = h->asm_state; a.rounds = h->asm_rounds; a.counters = h->asm_counters;
  a.biglist = h->asm_biglist; a.binlist = h->asm_binlist; a.idxg = h->asm_idxg; a.mg = h->asm_mg; a.row = h->asm_row; a.tqmax = h->tqmax; a.lrank = h->asm_lrank; a.ctot = h->asm_ctot; a.prec = h->asm_prec; a.redo = h->asm_redo; a.rowk = h->asm_rowk; a.lam32 = h->asm_lam32; a.H32 = h->H32; a.xh32 = h->asm_xh32; a.alpha = h->asm_alpha; a.ninf_best = h->asm_ninf; a.hi = h->asm_hi; a.kblk = h->asm_kblk; a.nkblk = 2 * (h->seg_max / 64 + 2); a.kref = 0; a.use_f32 = h->opts.asm_f32_rounds >= 0; a.xhw = h->asm_xhw; a.rowprob = h->asm_rowprob; a.wrows = 0; a.wmark = h->asm_wmark; a.wflag = h->asm_wflag; a.W = h->np; a.scratch = h->asm_scratch; a.work = h->asm_work;
  a.u_out = u_dev; a.ldu = h->ldu; a.nout = h->nout; a.act_out = act_dev; a.status_out = h->asm_status; a.iters_out = it_dev; a.words = h->words;
  a.nseg = nprob;
  { static const int tgi = getenv("NNMPC_TAIL_GI") ? 1 : 0; a.tail_gi = tgi; }   // (the variable: diagnostics, A/B of the tail's exchange rule)
  { static const int wg = getenv("NNMPC_NO_WG") ? 0 : 1; a.use_wg = wg; }   // (the variable: diagnostics, A/B against the kernels it replaced)
  { static const int rfn = getenv("NNMPC_REFINE") ? atoi(getenv("NNMPC_REFINE")) : 1; a.refine = 0; a.refine_later = rfn && a.use_f32;
    static const double rtol = getenv("NNMPC_REFINE_TOL") ? atof(getenv("NNMPC_REFINE_TOL")) : ASM_REFINE_TOL; a.refine_tol = rtol; }   // (the variable: diagnostics, A/B against separate f32 and fp64 rounds)
  { static const int e64 = getenv("NNMPC_EARLY64") ? atoi(getenv("NNMPC_EARLY64")) : 4; a.early64 = e64; }   // (the variable: diagnostics, A/B of the rule)
  // LAM / LAMW are all zero between calls: every entry the multiplier kernels write is cleared again by
  // asm_update_k / asm_wide_k of the same round
  if (!guess_dev) HIPCHK(hipMemsetAsync(h->asm_st, 0, (size_t)nprob * h->n, s));   // bound states: asm_init_k writes the leading window only
  // first sets: named by the dual projected-gradient predictor inside its window (qp_predict.h), by the bounds x_unc violates beyond
  // it (and everywhere when the predictor is off, the problem small, or the caller brought a guess)
  a.pred_w = 0; a.pred_f64 = 0;
  double pred_flops_per_it = 0.0;
  {
    static const int env_it = getenv("NNMPC_PRED_ITERS") ? atoi(getenv("NNMPC_PRED_ITERS")) : -1;   // (the variable: diagnostics, A/B; 0 = off)
    // (opts.asm_predict_iters: 0 adaptive, > 0 that many, < 0 off; the variable: 0 off, > 0 that many)
    const int iters_req = env_it >= 0 ? (env_it == 0 ? -1 : std::min(env_it, (int)PRED_MAXIT)) : h->opts.asm_predict_iters;
    const int iters = iters_req == 0 ? (int)PRED_MAXIT : iters_req;
    // (not for the one-wave-per-problem path of small problems: measured on the CSTRs-size batch of 10 000 -- n = 540, nu = 6, cond 4e7 --
    // the prediction takes the mean iteration count from 5.5 to 2, but a launch of asm_small_k lasts as long as its SLOWEST problem, and
    // that one keeps its ~25 iterations: 1.76 ms per step without, 1.94 with the predictor)
    if (iters > 0 && h->pred[0].L > 0.0 && !guess_dev && !small && !tail_only && Wx >= PRED_W) {
      PredArgs pa;
      { static const int f10 = getenv("NNMPC_PRED_FACTOR") ? atoi(getenv("NNMPC_PRED_FACTOR")) : 3; pa.fac10 = f10; }   // (the variable: diagnostics)
      pa.iters = iters; pa.adaptive = iters_req == 0; pa.itsum = h->profiling ? h->pred_cnt + 1 : nullptr;
      if (h->profiling) HIPCHK(hipMemsetAsync(h->pred_cnt + 1, 0, 4, s));
      double t = 1.0;
      for (int k = 0; k < iters; ++k) { const double tn = 0.5 * (1.0 + std::sqrt(1.0 + 4.0 * t * t)); pa.beta[k] = (float)((t - 1.0) / tn); t = tn; }
      for (int k = iters; k < PRED_MAXIT; ++k) pa.beta[k] = 0.f;
      // Which window: the optimum's active bounds reach further along the horizon than the bounds x_unc violates.  When more than a
      // quarter of the problems already violate a bound in the last quarter of the 512-column window (CDU plant: 0.05 % at sx = 2,
      // 8 % at sx = 4, 46 % at sx = 6), the 1024-column instance names the sets -- four times the L2 traffic per problem and x_unc
      // for those columns: 100 000 problems at sx = 6 (291 bounds each, the last one at column 757): 461 -> 412 ms per step; at
      // sx = 4 (216 bounds, last column 629) it costs more than it saves (76 -> 88 ms).  (One 4-byte read-back; only handles that
      // have the wide window pay it.)
      int wide = 0;
      static const int force_w = getenv("NNMPC_PRED_WIDE") ? atoi(getenv("NNMPC_PRED_WIDE")) : -1;   // (the variable: diagnostics, A/B; 0 / 1 forces the choice)
      if (h->pred[1].L > 0.0 && lazy) {
        if (force_w >= 0) wide = force_w;
        else {
          HIPCHK(hipMemsetAsync(h->pred_cnt, 0, 4, s));
          hipLaunchKernelGGL(asm_extent_k, dim3((nprob + 63) / 64), dim3(256), 0, s, a, PRED_W - PRED_W / 4, PRED_W, h->pred_cnt);
          HIPCHK(hipMemcpyAsync(h->pin_cnt, h->pred_cnt, 4, hipMemcpyDeviceToHost, s));
          HIPCHK(stream_sync(s));
          wide = h->pin_cnt[0] * 4 > nprob;
          h->pin_cnt[0] = 0;
        }
      }
      if (wide && Wx < PRED_W2) {
        EvScope es(h, 5, 2.0 * (PRED_W2 - Wx) * (double)h->ka * nprob);
        gemm64(h, h->asm_xunc + Wx, h->np, h->x0_64, h->ka, h->Kunc64 + (size_t)Wx * h->ka, h->ka, segp, PRED_W2 - Wx, h->ka);
        Wx = PRED_W2; a.Wx = Wx;
      }
      const bool nu4 = h->nu % 4 == 0;
      const int W = wide ? PRED_W2 : PRED_W, MR = (wide || !nu4) ? 32 : 64;
      pa.Hf = h->pred[wide].Hf;
      // (flops: the dense count -- every k-step of every iteration after the first; the iterations are summed on the device)
      pred_flops_per_it = 2.0 * W * (double)W * MR;
      h->stats.asm_predict_launches += 1;
      EvScope es(h, 10, 0.0);
      if (wide) hipLaunchKernelGGL((asm_predict_k<8, 2>), dim3((nprob + 31) / 32), dim3(64 * PRED_NW), (pred_lds_bytes<8, 2>(h->nu)), s, a, pa);
      else if (nu4) hipLaunchKernelGGL((asm_predict_k<4, 4>), dim3((nprob + 63) / 64), dim3(64 * PRED_NW), (pred_lds_bytes<4, 4>(h->nu)), s, a, pa);
      else hipLaunchKernelGGL((asm_predict_k<4, 2, false>), dim3((nprob + 31) / 32), dim3(64 * PRED_NW), (pred_lds_bytes<4, 2>(h->nu)), s, a, pa);
      a.pred_w = W;
      { static const int pf = getenv("NNMPC_PRED_F64") ? atoi(getenv("NNMPC_PRED_F64")) : 0; a.pred_f64 = pf; }   // (the variable: diagnostics, A/B)
    }
  }
  hipLaunchKernelGGL(asm_init_k, dim3((segp + 3) / 4), dim3(256), 0, s, a, segp);
  const int lds_big = (a.max_active + ASM_TS) * 8;
  int* cnt = h->pin_cnt;
  memset(cnt, 0, ASM_NCNT * sizeof(int));
  int rounds = 0;
  int prev_rows = 0;                                    // fp64 rows of the last round (of LAM): the problems that settled in them await the full-width check
  int kprev = 0, wide_cols = 0, fused_c0 = -1;
  HIPCHK(hipMemsetAsync(h->asm_counters, 0, ASM_NCNT * sizeof(int), s));
  if (small) {
    EvScope es(h, 4, 0.0);
    // iterations of grace before single exchanges (the variable: diagnostics).  24, not the rounds' ASM_GRACE = 10: on the cond-4e7
    // CSTRs-size plant block exchanges that stall for a dozen iterations mostly recover, and a problem sent to single exchanges
    // early pays for it with hundreds of them -- 131 072 problems at sx = 3: 260 ms per step at 10, 78 at 16, 39 at 24, 40 at 40
    // (the ones that truly cycle wait longer for the fallback: 146 ms at 100); the 10 000-problem batch: 47 -> 26 iterations at most
    static const int g0 = getenv("NNMPC_SMALL_GRACE") ? atoi(getenv("NNMPC_SMALL_GRACE")) : ASM_SM_GRACE;
    hipLaunchKernelGGL((asm_small_k<2, 3, 4>), dim3(nprob), dim3(64), asm_small_lds_bytes(2, h->n, h->nu), s, a, a.max_rounds, g0);
    hipLaunchKernelGGL((asm_small_k<7, 1, 8>), dim3(nprob), dim3(64), asm_small_lds_bytes(7, h->n, h->nu), s, a, a.max_rounds, g0);
    // ... and the sets that outgrow its 112 bounds (one problem in two of three 10 000-problem batches of the CSTRs-size plant: handed
    // to the lock-step machinery it cost a round of bookkeeping launches, a read-back and 30 more iterations in the device tail -- 2.7 -
    // 3.1 ms per step against 1.4 without such a problem) carry on in a nine-block instance; a wave whose problem is finished exits at once
    if (asm_small_lds_bytes(9, h->n, h->nu) <= ASM_SMALL9_LDS_MAX)   // (else: handed on as before)
      hipLaunchKernelGGL((asm_small_k<9, 1, 8>), dim3(nprob), dim3(64), asm_small_lds_bytes(9, h->n, h->nu), s, a, a.max_rounds, g0);
    h->stats.asm_rounds += 1;
    h->stats.asm_small_passes += 1;
    static const bool trace = getenv("NNMPC_TRACE_ROUNDS") != nullptr;   // diagnostics: iterations and set sizes per problem
    if (trace) {
      std::vector<int> rd(nprob), mg(nprob), stt(nprob);
      HIPCHK(stream_sync(s));
      hipMemcpy(rd.data(), h->asm_rounds, nprob * sizeof(int), hipMemcpyDeviceToHost);
      hipMemcpy(mg.data(), h->asm_mg, nprob * sizeof(int), hipMemcpyDeviceToHost);
      hipMemcpy(stt.data(), h->asm_state, nprob * sizeof(int), hipMemcpyDeviceToHost);
      long sum = 0, sm = 0, big = 0, bigit = 0, left = 0; int mx = 0, mm = 0, mxbig = 0, hist[8] = {0};
      for (int i = 0; i < nprob; ++i) {
        sum += rd[i]; mx = std::max(mx, rd[i]); sm += mg[i]; mm = std::max(mm, mg[i]); left += stt[i] == ASM_RUN;
        if (mg[i] > 32) { ++big; bigit += rd[i]; mxbig = std::max(mxbig, rd[i]); }
        hist[std::min(7, rd[i] / 8)]++;
      }
      fprintf(stderr, "asm small: %d problems, iterations mean %.2f max %d, final sets mean %.1f max %d; sets > 32: %ld (iterations mean %.1f max %d); handed on %ld; iterations/8 histogram",
              nprob, (double)sum / nprob, mx, (double)sm / nprob, mm, big, big ? (double)bigit / big : 0.0, mxbig, left);
      for (int b = 0; b < 8; ++b) fprintf(stderr, " %d", hist[b]);
      fprintf(stderr, "\n");
#ifdef ASM_SM_PROF
      unsigned long long tp[2][8];
      hipMemcpyFromSymbol(tp, HIP_SYMBOL(asm_small_prof), sizeof tp);
      for (int k = 0; k < 2; ++k)
        fprintf(stderr, "  small instance %d clock sums: list %llu, rhs + solve %llu, x window %llu, signs %llu, exchange / loop %llu; iterations %llu\n", k, tp[k][0], tp[k][1], tp[k][2], tp[k][3], tp[k][5], tp[k][6]);
      memset(tp, 0, sizeof tp); hipMemcpyToSymbol(HIP_SYMBOL(asm_small_prof), tp, sizeof tp);
#endif
    }
  }
  static const bool trace_tail = getenv("NNMPC_TRACE_ROUNDS") != nullptr;   // diagnostics: iterations per problem
  const bool defer_cnt = tail_only && !trace_tail;
  if (tail_only) {
    // A call of at most 256 problems -- the lock-step chains of a task, a controller's single QP -- is finished on the
    // device from the start (asm_tail_k: count -> fp64 solve -> x over all columns -> exchange rule, one workgroup per
    // problem): lock-step rounds would be eight launches and a host read-back each for a handful of workgroups.
    EvScope es(h, 4, 0.0);
    hipLaunchKernelGGL(asm_taillist_k, dim3((nprob + 255) / 256), dim3(256), 0, s, a);
    const int lds_tail = (a.max_active + ASM_TS + ASM_TAIL_AREA) * 8 + ((h->n + 15) / 16) * 16 + ASM_TAIL_EXTRA;
    hipLaunchKernelGGL(asm_tail_k, dim3(nprob), dim3(256), lds_tail, s, a, h->asm_tail_budget);
    // (the counters are read together with the status at the end of the segment -- one host round trip per call instead of two:
    // a chain step is such a call, 0.45 ms of which ~0.04 ms was this wait; the check with P is launched unconditionally then)
    if (!defer_cnt) {
      HIPCHK(hipMemcpyAsync(cnt, h->asm_counters, ASM_NCNT * sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(stream_sync(s));
    }
    h->stats.asm_rounds += 1;
    if (trace_tail) {
      std::vector<int> rd(nprob), mg(nprob);
      hipMemcpy(rd.data(), h->asm_rounds, nprob * sizeof(int), hipMemcpyDeviceToHost);
      hipMemcpy(mg.data(), h->asm_mg, nprob * sizeof(int), hipMemcpyDeviceToHost);
      long sum = 0, sm = 0; int mx = 0, mm = 0;
      for (int i = 0; i < nprob; ++i) { sum += rd[i]; mx = std::max(mx, rd[i]); sm += mg[i]; mm = std::max(mm, mg[i]); }
      fprintf(stderr, "asm tail only: %d problems, iterations mean %.2f max %d, active bounds mean %.1f max %d\n", nprob, (double)sum / nprob, mx, (double)sm / nprob, mm);
    }
  }
  for (; !tail_only && rounds < 2 * a.max_rounds + 2; ++rounds) {
    if (prev_rows) {
      // problems that settled inside last round's column window: all columns of x, once
      {
        EvScope es(h, 5, 0.0);
        // (a.W is still last round's window: those columns are in that round's XH rows already)
        const int c0 = (a.W < h->np && (h->np - a.W) % 128 == 0) ? a.W : 0;
        fused_c0 = ((h->np - c0) % 128 == 0 && !no_fuse) ? c0 : -1;     // the fused kernel's 128-column tiles fit: check in the GEMM's epilogue
        const int ntm = (prev_rows + 127) / 128, ntn = (h->np - c0) / 128;
        a.wrows = prev_rows;
        const nnmpc_qp::Far* ff = nullptr;
        if (lazy && fused_c0 > 0 && !no_far) {
          for (const auto& f : h->far) if (f.W == c0) ff = &f;
          if (!ff && h->far_missing.size() < 16 && std::find(h->far_missing.begin(), h->far_missing.end(), c0) == h->far_missing.end())
            h->far_missing.push_back(c0);                // (the host wrapper may add the factors for this window: nnmpc_qp_farfield_missing)
        }
        a.ff_err = 0.0; a.ff_efar = 0.0; a.ff_skip = 0; wide_far_rp = 0;
        if (ff) {
          // far-field form: T = [x0 | lamw] V, x[c0:] = T U'; first-move calls skip the column tiles |U_j| |T_p| certifies
          a.ffU = ff->U; a.ffVx = ff->Vx; a.ffVl = ff->Vl; a.ffcu = ff->cu; a.ffk = ff->kt; a.ffr = ff->rp; a.ffW = ff->W;
          a.ff_err = h->p_inf * ff->efar; a.ff_efar = ff->efar;
          a.ff_skip = h->nout <= c0 && h->nout < h->n;
          wide_far_rp = ff->rp; wide_far_ksum = ff->ksum;
          h->stats.asm_far_passes += 1;
          hipLaunchKernelGGL(asm_wide_t_k, dim3(g64_grid(ntm, ff->rp / 128)), dim3(256), G64_LDS, s, a, ntm, ff->rp / 128);
          if (a.ff_skip) hipLaunchKernelGGL(asm_wide_tnorm_k, dim3((ntm * 128 + 3) / 4), dim3(256), 0, s, a, ntm);
          hipLaunchKernelGGL(asm_wide_gemm_k<WIDE_FAR>, dim3(g64_grid(ntm, ntn)), dim3(256), G64_LDS, s, a, c0, ntm, ntn);
        } else if (fused_c0 >= 0 && lazy) {              // (lazy: c0 = W >= the first window -- never 0 -- and x_unc exists up to Wx >= W)
          hipLaunchKernelGGL(asm_wide_gemm_k<WIDE_LAZY>, dim3(g64_grid(ntm, ntn)), dim3(256), G64_LDS, s, a, c0, ntm, ntn);
        } else if (fused_c0 >= 0) {
          hipLaunchKernelGGL(asm_wide_gemm_k<WIDE_XUNC>, dim3(g64_grid(ntm, ntn)), dim3(256), G64_LDS, s, a, c0, ntm, ntn);
        } else {
          // shapes the fused kernels' tiles do not fit: XHW = LAM Pinv beyond c0 by the plain GEMM, k-range per 64-row block
          gemm64(h, h->asm_xhw + c0, h->np, h->asm_lam, h->np, h->H64 + (size_t)c0 * h->np, h->np, ((prev_rows + 127) / 128) * 128,
                 h->np - c0, h->np, nullptr, 0, h->asm_kblk, nullptr, true);
        }
        wide_cols = h->np - c0;
      }
      EvScope es(h, 6, 0.0);
      hipLaunchKernelGGL(asm_wide_k, dim3((prev_rows + 3) / 4), dim3(256), 0, s, a, fused_c0);
    }
    a.kref = kprev;
    if (!a.pred_w) a.refine_later = 0;                   // (only behind predicted first sets: from other starts the f32 rounds are many and their sets still move)
    a.refine = a.refine_later && rounds >= 1;            // round 0 stays the plain f32 screen
    {
      EvScope es(h, 6, 0.0);                            // set bookkeeping: counted with asm_update_k
      hipLaunchKernelGGL(asm_count_k, dim3((nprob + 3) / 4), dim3(256), 0, s, a);
      hipLaunchKernelGGL(asm_bins_a_k, dim3((nprob + 1023) / 1024), dim3(1024), 0, s, a);
      hipLaunchKernelGGL(asm_bins_b_k, dim3((nprob + 1023) / 1024), dim3(1024), 0, s, a);
    }
    HIPCHK(hipMemcpyAsync(cnt, h->asm_counters, ASM_NCNT * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(stream_sync(s));
    const int n64 = cnt[2], n32 = cnt[ASM_CNT_ROWS32], nrun = n64 + n32;   // solved in fp64 / f32 this round
    if (h->profiling)                                   // flops of the full-width pass that opened this round
    {
        // (from the scans of this round's asm_bins: the problems asm_wide_k just handled, the sum of their last active index + 1)
        const double nw = cnt[ASM_CNT_WIDE + 1], ksum = cnt[ASM_CNT_WKSUM];
        if (wide_far_rp) {
          // far-field form: T = z V (k = n_aug + own k-range) and x = T U' (k = the column tile's share of the basis) -- over all
          // columns, or (first-move calls) over the 128 x 128 tiles the certificate did not cover (device count of their k chunks)
          const double chunks = cnt[ASM_CNT_FFTILES] - fft_prev;
          h->stats.asm_gemm_flops += 2.0 * wide_far_rp * (ksum + nw * h->ka) +
                                     (a.ff_skip ? 2.0 * G64_KC * 128.0 * 128.0 * chunks : 2.0 * 128.0 * wide_far_ksum * nw);
        } else {
          h->stats.asm_gemm_flops += 2.0 * wide_cols * (ksum + (lazy ? nw * h->ka : 0.0));   // (lazy: x_unc beyond the window is part of that pass)
        }
        fft_prev = cnt[ASM_CNT_FFTILES];
    }
    kprev = cnt[3];
    {
      static const bool trace = getenv("NNMPC_TRACE_ROUNDS") != nullptr;   // diagnostics: the populations of every round
      if (trace) {
        fprintf(stderr, "asm round %d: fp64 %d f32 %d last-active %d wide-checked %d big %d big32 %d | fp64 classes", rounds, n64, n32, cnt[3],
                cnt[ASM_CNT_WIDE + 1], cnt[1], cnt[ASM_CNT_BIG32] + cnt[ASM_CNT_BIG32B]);
        for (int b = 0; b < ASM_NBIN; ++b) fprintf(stderr, " %d", cnt[4 + b]);
        fprintf(stderr, " | f32 classes");
        for (int b = 0; b < ASM_NBIN; ++b) fprintf(stderr, " %d", cnt[ASM_CNT_F32 + b]);
        fprintf(stderr, "\n");
      }
    }
    if (nrun == 0) break;
    static const int tail_max = getenv("NNMPC_TAIL_MAX") ? atoi(getenv("NNMPC_TAIL_MAX")) : 256;   // (the variable: diagnostics)
    if (nrun <= std::min(h->asm_pool, tail_max) && (rounds >= (a.pred_w ? 2 : 6) || small)) {
      // the tail: a handful of stragglers (the bulk settles in 5-8 rounds) -- finish them on the device (asm_tail_k)
      // instead of paying eight launches and a read-back per round for them.  From round 6 on -- from round 2 when the first sets
      // were predicted (qp_predict.h: the bulk then settles in rounds 1-2, and rounds 3-4 were 0.9 ms for 130 problems) -- (12 before: at 100 000
      // problems per call the rounds 7..12 were launches for a few dozen problems, 5 % of the step)
      EvScope es(h, 4, 0.0);
      hipLaunchKernelGGL(asm_taillist_k, dim3((nprob + 255) / 256), dim3(256), 0, s, a);
      if (Wx < h->np) {
        // the tail evaluates all columns of its problems straight from Pinv: their x_unc rows beyond Wx, gathered by problem
        gemm64(h, h->asm_xunc + Wx, h->np, h->x0_64, h->ka, h->Kunc64 + (size_t)Wx * h->ka, h->ka, ((nrun + 127) / 128) * 128, h->np - Wx, h->ka,
               nullptr, 0, nullptr, h->asm_counters + ASM_CNT_TAIL, false, h->asm_biglist);
        a.Wx = h->np;                                    // (for these problems)
      }
      const int lds_tail = (a.max_active + ASM_TS + ASM_TAIL_AREA) * 8 + ((h->n + 15) / 16) * 16 + ASM_TAIL_EXTRA;
      hipLaunchKernelGGL(asm_tail_k, dim3(nrun), dim3(256), lds_tail, s, a, h->asm_tail_budget);
      HIPCHK(hipMemcpyAsync(cnt, h->asm_counters, ASM_NCNT * sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(stream_sync(s));
      {
        static const bool trace = getenv("NNMPC_TRACE_ROUNDS") != nullptr;   // diagnostics: iterations of the tail's problems
        if (trace) {
          std::vector<int> rd(nprob), bl(nrun);
          hipMemcpy(rd.data(), h->asm_rounds, nprob * sizeof(int), hipMemcpyDeviceToHost);
          hipMemcpy(bl.data(), h->asm_biglist, nrun * sizeof(int), hipMemcpyDeviceToHost);
          long sum = 0; int mx = 0;
          for (int i = 0; i < nrun; ++i) { const int r = rd[bl[i]] - rounds; sum += r; mx = std::max(mx, r); }
          fprintf(stderr, "asm tail after round %d: %d problems, iterations mean %.1f max %d\n", rounds, nrun, (double)sum / nrun, mx);
#ifdef ASM_TAIL_PROF
          unsigned long long tp[8];
          hipMemcpyFromSymbol(tp, HIP_SYMBOL(asm_tail_prof), sizeof tp);
          fprintf(stderr, "  tail clock sums (all workgroups): count/factor/solve %llu, substitutions %llu, x loop %llu, tests %llu, exchange %llu; iterations on the dense factor %llu, on a fresh factorisation %llu\n",
                  tp[0], tp[1], tp[2], tp[3], tp[4], tp[5], tp[6]);
          memset(tp, 0, sizeof tp); hipMemcpyToSymbol(HIP_SYMBOL(asm_tail_prof), tp, sizeof tp);
#endif
        }
      }
      h->stats.asm_rounds += 1;
      rounds = 0;                                         // (regular exit: the counters just read are final)
      break;
    }
    h->stats.asm_rounds += 1;
    // column window of this round: past the last active bound of any running problem plus one stage; a
    // problem that settles inside it gets one full-width pass (asm_wide_k) at the start of the next round
    a.W = std::min(h->np, ((cnt[3] + 1 + h->nu + 127) / 128) * 128);
    if (a.W > Wx) {
      // a set reaches beyond the columns x_unc was formed for (a bound the full-width pass found violated out there): extend
      EvScope es(h, 5, 2.0 * (a.W - Wx) * (double)h->ka * nprob);
      gemm64(h, h->asm_xunc + Wx, h->np, h->x0_64, h->ka, h->Kunc64 + (size_t)Wx * h->ka, h->ka, segp, a.W - Wx, h->ka);
      Wx = a.W; a.Wx = Wx;
    }
    {
      EvScope es(h, 4, 0.0);
      // one wave per problem, S in registers: size classes 0..5 (<= 144 bounds) in one launch, four problems per
      // workgroup; classes 6, 7 (<= 176) two per workgroup and the rare larger sets (tiles in an L2 slab, one
      // workgroup each: long latency chains on a handful of CUs) beside it on the side stream.
      const int nreg2_wg = (cnt[4 + 6] + 1) / 2 + (cnt[4 + 7] + 1) / 2;
      const int nreg32b_wg = (cnt[ASM_CNT_F32 + 6] + 3) / 4 + (cnt[ASM_CNT_F32 + 7] + 3) / 4;
      // (use_wg, the default: sets of 177 .. 256 bounds, one workgroup of four or eight waves each -- qp_wg.h)
      const int nwg64 = a.use_wg ? cnt[ASM_CNT_BIG64] : 0;
      const int nwg32 = a.use_wg ? cnt[ASM_CNT_BIG32] : 0;
      const int nwg32b = a.use_wg ? cnt[ASM_CNT_BIG32B] : 0;   // f32 rounds of 257 .. 384 bounds: eight waves per problem
      const int nwg64r = a.use_wg ? cnt[ASM_CNT_BIG64R] : 0;   // ... and their fp64 solves: the same factorisation, refined to fp64 residuals
      int nbig = cnt[1] + cnt[ASM_CNT_BIG32] + nreg2_wg + nreg32b_wg + nwg64 + nwg32b + nwg64r, nreg_wg = 0, nreg32_wg = 0;
      for (int b = 0; b < ASM_NREG; ++b) { nreg_wg += (cnt[4 + b] + 3) / 4; nreg32_wg += (cnt[ASM_CNT_F32 + b] + 3) / 4; }
      // three side streams: [slab kernel: few workgroups, long chains -- it starts first and runs beside everything else],
      // [four-wave kernels of the 12 .. 16-block sets], [single-wave kernels of the 10- and 11-block classes]
      const bool side4 = cnt[1] > 0, side2 = nwg64 + nwg32 + nwg32b + nwg64r > 0 || (cnt[ASM_CNT_BIG32] && !a.use_wg), side3 = nreg2_wg + nreg32b_wg > 0;
      if (nbig) {
        HIPCHK(hipEventRecord(h->ev_fork, s));
        if (side4) {
          HIPCHK(hipStreamWaitEvent(h->stream4, h->ev_fork, 0));
          { EvScope e9(h, 9, 0.0, h->stream4); hipLaunchKernelGGL((asm_lambda_tile_k<1>), dim3(std::min(cnt[1], h->asm_pool)), dim3(256), lds_big, h->stream4, a, 0); }
          HIPCHK(hipEventRecord(h->ev_join4, h->stream4));
        }
        if (side2) {
          HIPCHK(hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
          EvScope e9(h, 9, 0.0, h->stream2);
          if (nwg64) hipLaunchKernelGGL(asm_lambda_wg64_k, dim3(nwg64), dim3(256), asm_wg_lds_bytes<double>(), h->stream2, a);
          if (nwg64r) hipLaunchKernelGGL(asm_lambda_wg64r_k, dim3(nwg64r), dim3(512), (asm_wg_lds_bytes_refine<float, ASM_WG_MB8>()), h->stream2, a);
          if (nwg32b) hipLaunchKernelGGL(asm_lambda_wg32b_k, dim3(nwg32b), dim3(512), (asm_wg_lds_bytes<float, ASM_WG_MB8>()), h->stream2, a);
          if (nwg32) hipLaunchKernelGGL(asm_lambda_wg32_k, dim3(nwg32), dim3(256), asm_wg_lds_bytes<float>(), h->stream2, a);
          if (cnt[ASM_CNT_BIG32] && !a.use_wg) hipLaunchKernelGGL(asm_lambda_tile32_k, dim3(std::min(cnt[ASM_CNT_BIG32], 4096)), dim3(512), ASM_TILE32_LDS, h->stream2, a);
          HIPCHK(hipEventRecord(h->ev_join, h->stream2));
        }
        if (side3) {
          HIPCHK(hipStreamWaitEvent(h->stream3, h->ev_fork, 0));
          EvScope e9(h, 9, 0.0, h->stream3);
          // fp64, 145 .. 176 bounds: two waves per problem (7.3 / 5.2 problems per microsecond at 160 / 176 bounds against the 6.0 / 4.5
          // of the single-wave kernel; in f32 the single-wave kernel wins, 18.1 against 14.3)
          if (nreg2_wg && a.use_wg) hipLaunchKernelGGL(asm_lambda_wg64s_k, dim3(cnt[4 + 6] + cnt[4 + 7]), dim3(128), asm_wg_lds_bytes<double>(), h->stream3, a);
          else if (nreg2_wg) hipLaunchKernelGGL(asm_lambda_reg2_k, dim3(nreg2_wg), dim3(128), ASM_REG2_LDS, h->stream3, a);
          if (nreg32b_wg) hipLaunchKernelGGL(asm_lambda_reg32b_k, dim3(nreg32b_wg), dim3(256), ASM_REG32B_LDS, h->stream3, a);
          HIPCHK(hipEventRecord(h->ev_join3, h->stream3));
        }
      }
      // fp64 first (its waves are the long ones), then the f32 rounds of the problems whose set still moves
      if (nreg_wg) { EvScope e8(h, 8, 0.0); hipLaunchKernelGGL(asm_lambda_reg_k, dim3(nreg_wg), dim3(256), ASM_REG_LDS, s, a); }
      if (nreg32_wg) { EvScope e7(h, 7, 0.0); hipLaunchKernelGGL(asm_lambda_reg32_k, dim3(nreg32_wg), dim3(256), ASM_REG32_LDS, s, a); }
      if (side2) HIPCHK(hipStreamWaitEvent(s, h->ev_join, 0));
      if (side3) HIPCHK(hipStreamWaitEvent(s, h->ev_join3, 0));
      if (side4) HIPCHK(hipStreamWaitEvent(s, h->ev_join4, 0));
    }
    {
      // the running problems sit in rows 0..n64-1 of LAM (fp64 solves) and 0..n32-1 of LAM32 (f32 solves), the rest
      // of the last row block is zero; algorithmic flops of LAM * Pinv: 2 * columns * (k up to the last active bound)
      // per running problem
      // algorithmic flops: 2 * window columns * (own last active bound + 1) per running problem (cnt[0] is their sum)
      EvScope es(h, 5, 2.0 * a.W * (double)cnt[0]);
      if (n64)
        gemm64(h, h->asm_xh, h->np, h->asm_lam, h->np, h->H64, h->np, ((n64 + 127) / 128) * 128, a.W, h->np, nullptr, 0,
               h->asm_kblk, nullptr, true);
      // (rows are ordered by their last active stage: every 64-row block has its own k-range, asm_bins_b_k)
      if (n32 && a.W % 128 == 0)
        hipLaunchKernelGGL((gemm_nt_f32_kdyn_k<128>), dim3(a.W / 128, (n32 + 127) / 128), dim3(256), TileCfg<128>::LDS_FLOATS * 4, s,
                           h->asm_xh32, (size_t)h->np, h->asm_lam32, (size_t)h->np, h->H32, (size_t)h->np, h->np, h->asm_kblk + a.nkblk / 2, 2);
      else if (n32)
        hipLaunchKernelGGL((gemm_nt_f32_kdyn_k<64>), dim3(a.W / 64, (n32 + 63) / 64), dim3(256), TileCfg<64>::LDS_FLOATS * 4, s,
                           h->asm_xh32, (size_t)h->np, h->asm_lam32, (size_t)h->np, h->H32, (size_t)h->np, h->np, h->asm_kblk + a.nkblk / 2, 1);
    }
    {
      EvScope es(h, 6, 0.0);
      // (rows of LAM whose problem settles inside the window are marked for the full-width pass that opens the next round)
      if (n64) HIPCHK(hipMemsetAsync(h->asm_rowprob, 0xFF, (size_t)((n64 + 127) / 128) * 128 * sizeof(int), s));
      hipLaunchKernelGGL(asm_update_k, dim3((nprob + 3) / 4), dim3(256), 0, s, a);
    }
    prev_rows = a.W < h->n ? n64 : 0;
  }
  // certification with P itself (rows the inverse-error bound could not certify): q = tq x0 and px = x P
  if (rounds >= 2 * a.max_rounds + 2) {                  // left by the round cap: the last counters are not final
    HIPCHK(hipMemcpyAsync(cnt, h->asm_counters, ASM_NCNT * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(stream_sync(s));
  }
  if (h->gemm_error) { h->gemm_error = false; return NNMPC_EINVAL; }
  if (!defer_cnt) h->stats.asm_full_checks += cnt[ASM_CNT_DONE];
  if (defer_cnt || cnt[ASM_CNT_DONE] > 0) {                // (row blocks without an ASM_DONE row leave these GEMMs at once)
    gemm64(h, h->q64_all, h->np, h->x0_64, h->ka, h->tq64, h->ka, segp, h->np, h->ka, h->asm_state, ASM_DONE);
    gemm64(h, h->asm_xh, h->np, h->asm_x, h->np, h->P64, h->np, segp, h->np, h->np, h->asm_state, ASM_DONE);
  }
  hipLaunchKernelGGL(asm_certify_k, dim3(nprob), dim3(256), 0, s, a, h->pscale);
  int* st = h->pin_st;
  HIPCHK(hipMemcpyAsync(st, h->asm_status, (size_t)nprob * sizeof(int), hipMemcpyDeviceToHost, s));
  int* pin_its = h->pin_cnt + ASM_NCNT;
  if (h->profiling && pred_flops_per_it > 0.0) HIPCHK(hipMemcpyAsync(pin_its, h->pred_cnt + 1, 4, hipMemcpyDeviceToHost, s));
  if (defer_cnt) HIPCHK(hipMemcpyAsync(cnt, h->asm_counters, ASM_NCNT * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(stream_sync(s));
  if (defer_cnt) h->stats.asm_full_checks += cnt[ASM_CNT_DONE];
  if (h->profiling && pred_flops_per_it > 0.0) h->stats.asm_predict_flops += pred_flops_per_it * (double)*pin_its;
  HIPCHK(hipGetLastError());
  std::vector<int> fb;
  int ninvalid = 0;
  for (int p = 0; p < nprob; ++p) { if (st[p] == 3) fb.push_back(p); else if (st[p] != 0) ++ninvalid; }   // 2: rejected inputs
  h->stats.asm_solved += nprob - (int64_t)fb.size() - ninvalid;
  if (st_dev) HIPCHK(hipMemcpyAsync(st_dev, h->asm_status, (size_t)nprob * sizeof(int), hipMemcpyDeviceToDevice, s));
  if (fb.empty()) { h->stats.problems += nprob; return 0; }
  if (h->opts.method == 2) {                           // asm only: report the rest as not certified
    for (int p = 0; p < nprob; ++p) st[p] = st[p] == 3 ? NNMPC_ST_MAXITER : st[p];
    if (st_dev) HIPCHK(hipMemcpy(st_dev, st, (size_t)nprob * sizeof(int), hipMemcpyHostToDevice));
    h->stats.problems += nprob;
    return 0;
  }
  // ---- PDIP fallback on the compacted remainder (scratch owned by the handle: nothing to release on an error path)
  const int cntf = (int)fb.size();
  int* list = nullptr; double *x0c = nullptr, *lbc = nullptr, *ubc = nullptr, *uc = nullptr;
  uint32_t* actc = nullptr; int *stc = nullptr, *itc = nullptr;
  unsigned char* guessc = nullptr;
  int rc = 0;
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_GUESS, &guessc, (size_t)cntf * h->n);
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_LIST, &list, cntf * sizeof(int));
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_X0, &x0c, (size_t)cntf * h->n_aug * 8);
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_LB, &lbc, (size_t)cntf * h->nu * 8);
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_UB, &ubc, (size_t)cntf * h->nu * 8);
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_U, &uc, (size_t)cntf * h->n * 8);
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_ACT, &actc, (size_t)cntf * h->words * 4);
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_ST, &stc, (size_t)cntf * 4);
  if (!rc) rc = scratch_get(h, nnmpc_qp::SC_FB_IT, &itc, (size_t)cntf * 8);
  if (rc) return rc;
  HIPCHK(hipMemcpy(list, fb.data(), cntf * sizeof(int), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(asm_gather_k, dim3(cntf), dim3(128), 0, s, x0c, lbc, ubc, x0_dev, lb_dev, ub_dev, list, cntf, h->n_aug, h->nu);
  // a problem whose set had settled but failed the check with P (inverse too inaccurate for its x) starts the polish
  // on that set; the others (not settled, too large, not positive definite) run the PDIP from scratch
  hipLaunchKernelGGL(asm_gather_guess_k, dim3(cntf), dim3(128), 0, s, guessc, h->asm_st, h->asm_state, list, cntf, h->n);
  const int ldu_caller = h->ldu, nout_caller = h->nout;
  h->ldu = h->n; h->nout = h->n;                       // the compacted copies hold whole sequences
  for (int b0 = 0; b0 < cntf && !rc; b0 += h->seg_max) {
    const int nb = std::min(h->seg_max, cntf - b0);
    rc = solve_segment(h, nb, x0c + (size_t)b0 * h->n_aug, lbc + (size_t)b0 * h->nu, ubc + (size_t)b0 * h->nu, guessc + (size_t)b0 * h->n,
                       uc + (size_t)b0 * h->n, actc + (size_t)b0 * h->words, stc + b0, itc + 2 * (size_t)b0);
  }
  h->ldu = ldu_caller; h->nout = nout_caller;
  if (!rc) {
    hipLaunchKernelGGL(asm_scatter_k, dim3(cntf), dim3(128), 0, s, u_dev, act_dev, st_dev, it_dev, uc, actc, stc, itc,
                       list, cntf, h->n, h->words, h->ldu, h->nout);
    HIPCHK(stream_sync(s));
  }
  h->stats.problems += nprob - cntf;   // solve_segment counted the fallback ones
  return rc;
}

}  // namespace

hipStream_t nnmpc_qp_stream_internal(nnmpc_qp* h) { return h->stream; }

extern "C" {

int nnmpc_qp_create(nnmpc_qp** out, int32_t n, int32_t nu, int32_t n_aug, const double* P,
                    const double* tq, const double* Kunc, const nnmpc_qp_opts* opts) {
  if (!out || !P || !tq || n <= 0 || nu <= 0 || nu > 256 || n_aug <= 0 || n % nu != 0) {
    set_error("nnmpc_qp_create: bad arguments (n=%d nu=%d n_aug=%d)", n, nu, n_aug);
    return NNMPC_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("nnmpc_qp_create: no HIP device available (this library has no CPU fallback)");
    return NNMPC_EHIP;
  }
  nnmpc_qp* h = new nnmpc_qp();
  memset(&h->stats, 0, sizeof(h->stats));
  memset(&h->d, 0, sizeof(h->d));
  h->profiling = false; h->ev_used = 0; h->have_inverse = false;
  h->ldu = n; h->nout = n;
  if (opts) h->opts = *opts; else memset(&h->opts, 0, sizeof(h->opts));
  nnmpc_qp_opts& o = h->opts;
  if (o.max_batch <= 0) o.max_batch = 1024;
  if (o.nb == 0) o.nb = (n <= 1024) ? 64 : 128;
  if (o.nb != 64 && o.nb != 128) { set_error("nb must be 64 or 128"); delete h; return NNMPC_EINVAL; }
  if (o.max_ipm_iters <= 0) o.max_ipm_iters = 40;
  if (o.max_polish_rounds <= 0) o.max_polish_rounds = 150;
  if (o.max_refine <= 0) o.max_refine = 60;
  if (o.max_rounds <= 0) o.max_rounds = 1000;
  if (o.sub_steps <= 0) o.sub_steps = 8;
  if (o.stale_max_changes == 0) o.stale_max_changes = 4;   // < 0 disables factor reuse
  if (o.stale_cg_limit <= 0) o.stale_cg_limit = 16;
  if (o.asm_max_active <= 0) o.asm_max_active = 768;
  if (o.asm_max_active > 768) o.asm_max_active = 768;
  o.asm_max_active = std::max(16, (o.asm_max_active / 16) * 16);
  // the device tail (asm_tail_k) gets 50 000 iterations unless the caller set a budget: Murty's rule is finite but slow on dense
  // Hessians with cond >= 1e5 and half the bounds active -- scripts/stress_asm.py, seeds 1 and 3: 130 and 166 of ~1600 problems
  // unfinished after 200 iterations, 27 and 1 after 4000, none after 30 000 (a straggler then holds one workgroup for ~1 s; the
  // problems of the reference's regime settle within 40 iterations and never see the difference)
  h->asm_tail_budget = o.asm_max_rounds <= 0 ? 50000 : o.asm_max_rounds;
  if (o.asm_max_rounds <= 0) o.asm_max_rounds = 200;
  if (o.asm_predict_iters > PRED_MAXIT) o.asm_predict_iters = PRED_MAXIT;   // (0: adaptive, 8 .. PRED_MAXIT by workgroup)
  if (o.sub_steps < 2) o.sub_steps = 2;
  if (o.ipm_tol <= 0.f) o.ipm_tol = 1e-2f;
  if (o.refine_tol <= 0.0) o.refine_tol = 1e-10;
  if (o.bound_tol <= 0.0) o.bound_tol = 1e-9;
  hipGetDevice(&h->device);
  h->n = n; h->nu = nu; h->n_aug = n_aug;
  h->NB = o.nb;
  h->np = ((n + h->NB - 1) / h->NB) * h->NB;
  h->T = h->np / h->NB;
  h->tiles = h->T * (h->T + 1) / 2;
  h->ka = ((n_aug + 31) / 32) * 32;
  h->slots = ((o.max_batch + 127) / 128) * 128;
  h->words = (2 * n + 31) / 32;
  h->have_kunc = Kunc != nullptr;
  if ((size_t)(h->np + 5 * h->NB) * 4 > 96 * 1024) { set_error("n too large for the LDS-resident solve vector"); delete h; return NNMPC_EINVAL; }
  if (hipStreamCreate(&h->stream) != hipSuccess) { set_error("hipStreamCreate failed"); delete h; return NNMPC_EHIP; }
  if (hipStreamCreate(&h->stream2) != hipSuccess || hipStreamCreate(&h->stream3) != hipSuccess || hipStreamCreate(&h->stream4) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join3, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->ev_join4, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
    set_error("hipStreamCreate / hipEventCreate failed"); nnmpc_qp_destroy(h); return NNMPC_EHIP;
  }
  {
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_lambda_reg_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_REG_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_lambda_reg2_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_REG2_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_lambda_reg32_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_REG32_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_tail_k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_lambda_reg32b_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_REG32B_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_lambda_tile32_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_TILE32_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_f64_t128_k, hipFuncAttributeMaxDynamicSharedMemorySize, G64_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_wide_gemm_k<WIDE_XUNC>, hipFuncAttributeMaxDynamicSharedMemorySize, G64_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_wide_gemm_k<WIDE_LAZY>, hipFuncAttributeMaxDynamicSharedMemorySize, G64_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_wide_gemm_k<WIDE_FAR>, hipFuncAttributeMaxDynamicSharedMemorySize, G64_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_wide_t_k, hipFuncAttributeMaxDynamicSharedMemorySize, G64_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)asm_small_k<9, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_SMALL9_LDS_MAX);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_f32_kdyn_k<128>, hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg<128>::LDS_FLOATS * 4);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); nnmpc_qp_destroy(h); return NNMPC_EHIP; }
  }
  if (set_lds_attrs<128>() != 0 || set_lds_attrs<64>() != 0) {
    set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed"); nnmpc_qp_destroy(h); return NNMPC_EHIP;
  }

  const int np = h->np, NB = h->NB, ka = h->ka, S = h->slots;
  const size_t nb2 = (size_t)NB * NB;
  int rc = 0;
#define A_(ptr, cnt) if (!rc) rc = dev_alloc(h, &(ptr), (size_t)(cnt))
  A_(h->Pt, h->tiles * nb2); A_(h->P32, (size_t)np * np); A_(h->P64, (size_t)np * np);
  A_(h->tq64, (size_t)np * ka); A_(h->Kunc32, (size_t)np * ka); A_(h->pdiag, np);
  A_(h->L, (size_t)S * h->tiles * nb2); A_(h->Y, (size_t)S * h->T * nb2); A_(h->Dacc, (size_t)S * h->T * nb2); A_(h->trsv_count, 1);
  QpDev& d = h->d;
  const size_t V = (size_t)S * np;
  A_(d.u, V); A_(d.zu, V); A_(d.zl, V); A_(d.lbv, V); A_(d.ubv, V); A_(d.q, V); A_(d.PU, V);
  A_(d.rd, V); A_(d.rhs, V); A_(d.sol, V); A_(d.dua, V); A_(d.dvec, V); A_(d.mask, V); A_(d.uunc, V);
  A_(d.x, V); A_(d.q64, V); A_(d.PX, V); A_(d.r64, V); A_(d.p64, V); A_(d.v64, V); A_(d.st, V);
  A_(d.phase, S); A_(d.f_factor, S); A_(d.f_solve, S); A_(d.istep, S); A_(d.ipm_it, S);
  A_(d.nfac, S); A_(d.prounds, S); A_(d.pninf, S); A_(d.pgrace, S); A_(d.ptie, S); A_(d.rcnt, S); A_(d.psub, S); A_(d.fail, S); A_(d.stale, S); A_(d.rz, S);
  A_(d.mu, S); A_(d.gap, S); A_(d.smu, S); A_(d.qscale, S); A_(d.counters, 8);
  {
    // segment size: as many problems as a quarter of the free HBM allows (per problem: q f64, warm start f32 and the
    // six f64 rows of the active-set pass, its set list and bound states).  Large segments amortise the thinly
    // populated last rounds of the active-set pass.
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)12e9;
    long long cap = (long long)(0.25 * (double)free_b / (12.0 * np + 88.0 * np + n + 4.0 * o.asm_max_active + 64.0));
    if (o.seg_max > 0) cap = std::min<long long>(cap, o.seg_max);
    cap = std::max<long long>(cap, S);
    cap = std::min<long long>(cap, 1 << 20);
    h->seg_max = (int)((cap / 128) * 128);
  }
  const size_t G = h->seg_max;
  A_(h->x0_64, G * ka); A_(h->x0_32, G * ka);
  A_(h->q64_all, G * np); A_(h->uunc_all, G * np);
  A_(h->lb_d, G * nu); A_(h->ub_d, G * nu);
  A_(h->in_stage, G * n_aug);
  if (!rc && hipHostMalloc((void**)&h->pin_cnt, (ASM_NCNT + 4) * sizeof(int)) != hipSuccess) { set_error("nnmpc_qp_create: hipHostMalloc failed"); rc = NNMPC_EHIP; }
  if (!rc && hipHostMalloc((void**)&h->pin_st, G * sizeof(int)) != hipSuccess) { set_error("nnmpc_qp_create: hipHostMalloc failed"); rc = NNMPC_EHIP; }
  A_(d.lb64, (size_t)S * nu); A_(d.ub64, (size_t)S * nu);
  A_(d.slot_prob, S); A_(d.age, S); A_(d.next_prob, 1);
  // workgroups of the large-set kernel in flight (each with its tile slab in HBM / L2): four per CU for bulk batches --
  // the kernel is a latency chain (barriers, tiles in global memory), occupancy is what hides it
  h->asm_pool = h->seg_max >= 4096 ? (getenv("NNMPC_TAIL_MAX") && atoi(getenv("NNMPC_TAIL_MAX")) > 1024 ? 2048 : 1024) : 256;
  A_(h->H64, (size_t)np * np); A_(h->Kunc64, (size_t)np * ka); A_(h->H32, (size_t)np * np);
  A_(h->asm_xunc, G * np); A_(h->asm_x, G * np); A_(h->asm_lam, G * np); A_(h->asm_xh, G * np);
  A_(h->asm_xhw, (G + 256) * np); A_(h->asm_rowprob, G + 256); A_(h->asm_wflag, G); A_(h->asm_wmark, G);
  A_(h->asm_tnorm, G + 128 * (ASM_NKG + 1)); A_(h->asm_tslack, G + 128 * (ASM_NKG + 1));
  A_(h->asm_st, G * n); A_(h->asm_state, G); A_(h->asm_rounds, G); A_(h->asm_counters, ASM_NCNT);
  A_(h->asm_biglist, G); A_(h->asm_status, G); A_(h->asm_binlist, (size_t)(ASM_NLIST + 4) * G);
  A_(h->asm_idxg, G * o.asm_max_active); A_(h->asm_mg, G); A_(h->asm_row, G); A_(h->asm_lrank, G); A_(h->asm_ctot, ((G + 1023) / 1024) * ASM_NSCAN); A_(h->asm_prec, G); A_(h->asm_redo, G); A_(h->asm_rowk, G); A_(h->asm_lam32, G * np); A_(h->asm_xh32, G * np); A_(h->asm_alpha, G); A_(h->asm_ninf, G); A_(h->asm_hi, G); A_(h->asm_kblk, 2 * (G / 64 + 2)); A_(h->asm_work, 3 * G);
  A_(h->asm_scratch, (size_t)h->asm_pool * ((size_t)(o.asm_max_active / 16) * (o.asm_max_active / 16 + 1) / 2 * ASM_TS));
#undef A_
  if (rc) { nnmpc_qp_destroy(h); return rc; }
  d.n = n; d.np = np; d.nu = nu; d.slots = S; d.words = h->words;
  d.max_ipm = o.max_ipm_iters; d.max_polish = o.max_polish_rounds; d.max_refine = o.max_refine;
  d.max_rounds = o.max_rounds; d.stale_max_changes = o.stale_max_changes; d.stale_cg_limit = o.stale_cg_limit;
  d.ipm_tol = o.ipm_tol; d.refine_tol = o.refine_tol; d.bound_tol = o.bound_tol; d.stat_tol = 1e-8;

  // host-side packing of the shared matrices (one-time setup)
  std::vector<float> pt(h->tiles * nb2, 0.f), p32((size_t)np * np, 0.f), k32((size_t)np * ka, 0.f), pdg(np, 1.f);
  std::vector<double> p64((size_t)np * np, 0.0), t64((size_t)np * ka, 0.0);
  {
    std::vector<double> dg(n);
    for (int r = 0; r < n; ++r) dg[r] = P[(size_t)r * n + r];
    std::nth_element(dg.begin(), dg.begin() + n / 2, dg.end());
    h->pscale = dg[n / 2];
    if (!(h->pscale > 0.0)) { set_error("P has a non-positive diagonal"); nnmpc_qp_destroy(h); return NNMPC_EINVAL; }
  }
  const double ips = 1.0 / h->pscale;
  float pdmax = 0.f;
  for (int r = 0; r < n; ++r) {
    pdg[r] = (float)(P[(size_t)r * n + r] * ips);
    pdmax = std::max(pdmax, pdg[r]);
    for (int c = 0; c <= r; ++c) {
      const double v = P[(size_t)r * n + c];  // lower triangle is authoritative
      p64[(size_t)r * np + c] = v; p64[(size_t)c * np + r] = v;
      p32[(size_t)r * np + c] = (float)(v * ips); p32[(size_t)c * np + r] = (float)(v * ips);
    }
  }
  d.pscale = h->pscale;
  for (int r = 0; r < n; ++r) {
    double rs = 0.0;
    for (int c = 0; c < n; ++c) rs += std::fabs(p64[(size_t)r * np + c]);
    h->p_inf = std::max(h->p_inf, rs);
  }
  d.delta = 16.f * 5.96e-8f * pdmax;   // keeps the f32 Cholesky positive for cond(P) >~ 1e7
  d.pdiag = h->pdiag;
  for (int i = 0; i < h->T; ++i)
    for (int j = 0; j <= i; ++j) {
      float* t = pt.data() + ((size_t)i * (i + 1) / 2 + j) * nb2;
      for (int r = 0; r < NB; ++r)
        for (int c = 0; c < NB; ++c) {
          const int gr = i * NB + r, gc = j * NB + c;
          float v = 0.f;
          if (gr < n && gc < n) v = (float)((gc <= gr ? P[(size_t)gr * n + gc] : P[(size_t)gc * n + gr]) * ips);
          else if (gr == gc) v = 1.f;
          t[(size_t)r * NB + c] = v;
        }
    }
  h->tqmax = 0.0;
  for (int r = 0; r < n; ++r)
    for (int k = 0; k < n_aug; ++k) {
      t64[(size_t)r * ka + k] = tq[(size_t)r * n_aug + k];
      h->tqmax = std::max(h->tqmax, std::fabs(tq[(size_t)r * n_aug + k]));
      if (Kunc) k32[(size_t)r * ka + k] = (float)Kunc[(size_t)r * n_aug + k];
    }
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipMemcpy(h->Pt, pt.data(), pt.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->P32, p32.data(), p32.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->P64, p64.data(), p64.size() * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->tq64, t64.data(), t64.size() * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->Kunc32, k32.data(), k32.size() * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->pdiag, pdg.data(), pdg.size() * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("upload of P/tq failed: %s", hipGetErrorString(e)); nnmpc_qp_destroy(h); return NNMPC_EHIP; }
  *out = h;
  return NNMPC_OK;
}

int nnmpc_qp_destroy(nnmpc_qp* h) {
  if (!h) return NNMPC_OK;
  hipDeviceSynchronize();
  for (void* p : h->allocs) hipFree(p);
  for (auto& g : h->far) { hipFree(g.U); hipFree(g.Vx); hipFree(g.Vl); hipFree(g.cu); hipFree(g.kt); }
  for (auto& sc : h->sc) if (sc.p) hipFree(sc.p);
  for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  if (h->ev_join3) hipEventDestroy(h->ev_join3);
  if (h->ev_join4) hipEventDestroy(h->ev_join4);
  if (h->pin_cnt) hipHostFree(h->pin_cnt);
  if (h->pin_st) hipHostFree(h->pin_st);
  if (h->stream2) hipStreamDestroy(h->stream2);
  if (h->stream3) hipStreamDestroy(h->stream3);
  if (h->stream4) hipStreamDestroy(h->stream4);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return NNMPC_OK;
}

int nnmpc_qp_set_inverse(nnmpc_qp* h, const double* Hinv, const double* Kunc) {
  if (!h || !Hinv || !Kunc) { set_error("nnmpc_qp_set_inverse: bad arguments"); return NNMPC_EINVAL; }
  HIPCHK(hipSetDevice(h->device));
  const int n = h->n, np = h->np, ka = h->ka, n_aug = h->n_aug;
  if ((uint64_t)np * np * 8 >= (1ull << 32)) {            // the multiplier kernels address Pinv with 32-bit byte offsets
    set_error("nnmpc_qp_set_inverse: n = %d is too large for the active-set pass (n < 23168)", n);
    return NNMPC_EINVAL;
  }
  std::vector<double> hh((size_t)np * np, 0.0), kk((size_t)np * ka, 0.0);
  for (int r = 0; r < n; ++r) {
    for (int c = 0; c < n; ++c) hh[(size_t)r * np + c] = 0.5 * (Hinv[(size_t)r * n + c] + Hinv[(size_t)c * n + r]);
    for (int k = 0; k < n_aug; ++k) kk[(size_t)r * ka + k] = Kunc[(size_t)r * n_aug + k];
  }
  HIPCHK(hipMemcpy(h->H64, hh.data(), hh.size() * 8, hipMemcpyHostToDevice));
  {
    std::vector<float> h32(hh.size());
    for (size_t i = 0; i < hh.size(); ++i) h32[i] = (float)hh[i];
    HIPCHK(hipMemcpy(h->H32, h32.data(), h32.size() * 4, hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemcpy(h->Kunc64, kk.data(), kk.size() * 8, hipMemcpyHostToDevice));
  // ---- first-set predictor (qp_predict.h): the 512-column window, and the 1024-column one for batches whose sets reach further
  h->pred[0].L = h->pred[1].L = 0.0;
  if (h->nu <= 64 && h->nu >= 4 && h->opts.asm_predict_iters >= 0) {
    if (np >= PRED_W && n >= PRED_W) {
      const int rc = build_predictor<4>(h, hh, np, h->pred[0]);
      if (rc) return rc;
      if (!h->pred_cnt) { const int rc2 = dev_alloc(h, &h->pred_cnt, 4); if (rc2) return rc2; }
      HIPCHK(hipFuncSetAttribute((const void*)asm_predict_k<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, pred_lds_bytes<4, 4>(h->nu)));
      HIPCHK(hipFuncSetAttribute((const void*)asm_predict_k<4, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, pred_lds_bytes<4, 2>(h->nu)));
    }
    if (np >= PRED_W2 && n >= PRED_W2 && h->pred[0].L > 0.0 && h->nu % 4 == 0) {
      const int rc = build_predictor<8>(h, hh, np, h->pred[1]);
      if (rc) return rc;
      HIPCHK(hipFuncSetAttribute((const void*)asm_predict_k<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, pred_lds_bytes<8, 2>(h->nu)));
    }
  }
  // ---- verify the inverse once, on the device copies the solves will use:
  //      E2 = P Pinv - I,  E1 = P Kunc + tq;  their maxima feed the per-problem certificate
  {
    const int kap = ((n_aug + 63) / 64) * 64;
    double *tmp = nullptr, *kt = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, (size_t)np * np * 8));
    HIPCHK(hipMalloc((void**)&kt, (size_t)kap * np * 8));
    std::vector<double> ktr((size_t)kap * np, 0.0);
    for (int r = 0; r < n; ++r)
      for (int k = 0; k < n_aug; ++k) ktr[(size_t)k * np + r] = Kunc[(size_t)r * n_aug + k];
    HIPCHK(hipMemcpy(kt, ktr.data(), ktr.size() * 8, hipMemcpyHostToDevice));
    gemm64(h, tmp, np, h->P64, np, h->H64, np, np, np, np);                 // P Pinv (Pinv symmetric)
    HIPCHK(stream_sync(h->stream));
    std::vector<double> c((size_t)np * np);
    HIPCHK(hipMemcpy(c.data(), tmp, c.size() * 8, hipMemcpyDeviceToHost));
    double e2 = 0.0;
    for (int r = 0; r < n; ++r)
      for (int cc = 0; cc < n; ++cc) e2 = std::max(e2, std::fabs(c[(size_t)r * np + cc] - (r == cc ? 1.0 : 0.0)));
    gemm64(h, tmp, kap, h->P64, np, kt, np, np, kap, np);                   // P Kunc  -> [np][kap]
    HIPCHK(stream_sync(h->stream));
    HIPCHK(hipMemcpy(c.data(), tmp, (size_t)np * kap * 8, hipMemcpyDeviceToHost));
    std::vector<double> tqh((size_t)np * ka);
    HIPCHK(hipMemcpy(tqh.data(), h->tq64, tqh.size() * 8, hipMemcpyDeviceToHost));
    double e1 = 0.0;
    for (int r = 0; r < n; ++r)
      for (int k = 0; k < n_aug; ++k) e1 = std::max(e1, std::fabs(c[(size_t)r * kap + k] + tqh[(size_t)r * ka + k]));
    hipFree(tmp); hipFree(kt);
    h->asm_e1max = e1; h->asm_e2max = e2;
    if (!(e2 < 1e-6) || !(e1 == e1)) {
      set_error("nnmpc_qp_set_inverse: |P Pinv - I|_max = %.3e: the supplied inverse is not usable", e2);
      return NNMPC_EINVAL;
    }
  }
  h->have_inverse = true;
  return NNMPC_OK;
}

int nnmpc_qp_set_farfield(nnmpc_qp* h, int32_t W, int32_t r, const double* U, const double* Vx, const double* Vl) {
  if (!h || !U || !Vx || !Vl || r <= 0) { set_error("nnmpc_qp_set_farfield: bad arguments"); return NNMPC_EINVAL; }
  if (!h->have_inverse) { set_error("nnmpc_qp_set_farfield: needs nnmpc_qp_set_inverse first"); return NNMPC_EINVAL; }
  HIPCHK(hipSetDevice(h->device));
  const int n = h->n, np = h->np, ka = h->ka, n_aug = h->n_aug;
  const int rp = ((r + 127) / 128) * 128;
  if (W <= 0 || W % 128 != 0 || W >= n || np % 128 != 0) { set_error("nnmpc_qp_set_farfield: W = %d must be a multiple of 128 below n = %d (padded %d)", W, n, np); return NNMPC_EINVAL; }
  // the factored form costs 2 rp (ka + W + columns beyond W) flops per problem, the dense form 2 (ka + W)(columns beyond W): refuse
  // factors that do not pay (a generic Hessian's far block has full rank; the MPC structure is what makes it ~Nx)
  if ((double)rp * (ka + W + (np - W)) > 0.8 * (double)(ka + W) * (np - W)) {
    set_error("nnmpc_qp_set_farfield: rank %d does not pay against the dense form at W = %d (n = %d, n_aug = %d)", r, W, n, n_aug);
    return NNMPC_EINVAL;
  }
  if (rp > np) { set_error("nnmpc_qp_set_farfield: rank %d too large for the workspace", r); return NNMPC_EINVAL; }
  const int nf = np - W;                                  // rows of the far block (padding rows: zero)
  std::vector<double> u((size_t)nf * rp, 0.0), vx((size_t)rp * ka, 0.0), vl((size_t)rp * W, 0.0), cu(nf / 128, 0.0);
  for (int j = 0; j < n - W; ++j)
    for (int i = 0; i < r; ++i) u[(size_t)j * rp + i] = U[(size_t)j * r + i];
  for (int i = 0; i < r; ++i) {
    for (int k = 0; k < n_aug; ++k) vx[(size_t)i * ka + k] = Vx[(size_t)i * n_aug + k];
    for (int k = 0; k < W; ++k) vl[(size_t)i * W + k] = Vl[(size_t)i * W + k];
  }
  // |U_j| (2-norm of the row, rounded up), then its maximum over all columns at or beyond each 128-column tile
  {
    double run = 0.0;
    for (int t = nf / 128 - 1; t >= 0; --t) {
      for (int j = 128 * t; j < 128 * (t + 1); ++j) {
        double s2 = 0.0;
        for (int i = 0; i < rp; ++i) s2 += u[(size_t)j * rp + i] * u[(size_t)j * rp + i];
        run = std::max(run, std::sqrt(s2) * (1.0 + 1e-12));
      }
      cu[t] = run;
    }
  }
  // staircase: columns of U a column tile's rows use at all (exact zeros beyond), in whole k chunks of the GEMM
  std::vector<int> kt(nf / 128, 0);
  double ksum = 0.0;
  for (int t = 0; t < nf / 128; ++t) {
    int last = 0;
    for (int j = 128 * t; j < 128 * (t + 1); ++j)
      for (int i = rp - 1; i >= last; --i)
        if (u[(size_t)j * rp + i] != 0.0) { last = i + 1; break; }
    kt[t] = ((last + G64_KC - 1) / G64_KC) * G64_KC;
    ksum += kt[t];
  }
  nnmpc_qp::Far f;
  f.W = W; f.r = r; f.rp = rp; f.U = f.Vx = f.Vl = f.cu = nullptr; f.kt = nullptr; f.ksum = ksum; f.efar = 0.0;
  // the factors' buffers are owned by their Far entry (not by the handle's allocation list): a refused set and a replaced one are
  // released at once
  unsigned long long* em = nullptr;
  auto release = [&](nnmpc_qp::Far& g) {
    hipFree(g.U); hipFree(g.Vx); hipFree(g.Vl); hipFree(g.cu); hipFree(g.kt);
    g.U = g.Vx = g.Vl = g.cu = nullptr; g.kt = nullptr;
  };
  auto fail = [&](int code) { release(f); if (em) hipFree(em); return code; };
#define FFCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e_)); return fail(e_ == hipErrorOutOfMemory ? NNMPC_ENOMEM : NNMPC_EHIP); } } while (0)
  FFCHK(hipMalloc((void**)&f.U, u.size() * 8));
  FFCHK(hipMalloc((void**)&f.Vx, vx.size() * 8));
  FFCHK(hipMalloc((void**)&f.Vl, vl.size() * 8));
  FFCHK(hipMalloc((void**)&f.cu, cu.size() * 8));
  FFCHK(hipMalloc((void**)&f.kt, kt.size() * sizeof(int)));
  FFCHK(hipMemcpy(f.kt, kt.data(), kt.size() * sizeof(int), hipMemcpyHostToDevice));
  FFCHK(hipMemcpy(f.U, u.data(), u.size() * 8, hipMemcpyHostToDevice));
  FFCHK(hipMemcpy(f.Vx, vx.data(), vx.size() * 8, hipMemcpyHostToDevice));
  FFCHK(hipMemcpy(f.Vl, vl.data(), vl.size() * 8, hipMemcpyHostToDevice));
  FFCHK(hipMemcpy(f.cu, cu.data(), cu.size() * 8, hipMemcpyHostToDevice));
  // ---- verify on the device copies the passes will use: max |U [Vx | Vl] - [Kunc[W:] | -Pinv[W:, 0:W]]|
  {
    FFCHK(hipMalloc((void**)&em, 8));
    FFCHK(hipMemset(em, 0, 8));
    hipLaunchKernelGGL(far_verify_k, dim3(nf), dim3(256), rp * sizeof(double), h->stream, f.U, f.Vx, f.Vl, h->Kunc64, h->H64, W, rp, ka, np, em);
    unsigned long long bits = 0;
    FFCHK(hipMemcpyAsync(&bits, em, 8, hipMemcpyDeviceToHost, h->stream));
    FFCHK(stream_sync(h->stream));
    hipFree(em); em = nullptr;
    double e;
    memcpy(&e, &bits, 8);
    f.efar = e;
    if (!(e < 1e-9)) { set_error("nnmpc_qp_set_farfield: max |U V' - M| = %.3e: the factors are not usable", e); return fail(NNMPC_EINVAL); }
  }
#undef FFCHK
  h->far_missing.erase(std::remove(h->far_missing.begin(), h->far_missing.end(), W), h->far_missing.end());
  for (auto& g : h->far) if (g.W == W) { HIPCHK(stream_sync(h->stream)); release(g); g = f; return NNMPC_OK; }   // replaces the window's factors
  h->far.push_back(f);
  return NNMPC_OK;
}

int nnmpc_qp_farfield_missing(nnmpc_qp* h, int32_t* W) {
  if (!h || !W) { set_error("nnmpc_qp_farfield_missing: bad arguments"); return NNMPC_EINVAL; }
  *W = 0;
  if (!h->far_missing.empty()) { *W = h->far_missing.back(); h->far_missing.pop_back(); }   // handed out once: a window the caller cannot factor is not asked for again until met again
  return NNMPC_OK;
}

int nnmpc_qp_dims(nnmpc_qp* h, int32_t* n, int32_t* nu, int32_t* n_aug) {
  if (!h) { set_error("nnmpc_qp_dims: null handle"); return NNMPC_EINVAL; }
  if (n) *n = h->n;
  if (nu) *nu = h->nu;
  if (n_aug) *n_aug = h->n_aug;
  return NNMPC_OK;
}

int nnmpc_qp_set_profiling(nnmpc_qp* h, int32_t on) {
  if (!h) return NNMPC_EINVAL;
  h->profiling = on != 0;
  return NNMPC_OK;
}

int nnmpc_qp_get_stats(nnmpc_qp* h, nnmpc_qp_stats* out, int32_t reset) {
  if (!h || !out) return NNMPC_EINVAL;
  *out = h->stats;
  out->asm_e1max = h->asm_e1max; out->asm_e2max = h->asm_e2max;
  {
    std::vector<double> w(3 * (size_t)h->seg_max, 0.0);
    hipMemcpy(w.data(), h->asm_work, w.size() * sizeof(double), hipMemcpyDeviceToHost);
    double f = 0.0, b = 0.0, f32 = 0.0;
    for (size_t i = 0; i < w.size(); i += 3) { f += w[i]; b += w[i + 1]; f32 += w[i + 2]; }
    out->asm_lambda_flops = f; out->asm_lambda_bytes = b; out->asm_lambda32_flops = f32;
    if (reset) hipMemset(h->asm_work, 0, w.size() * sizeof(double));
  }
  {
    unsigned long long c = 0;
    hipMemcpy(&c, h->trsv_count, sizeof(c), hipMemcpyDeviceToHost);
    out->trsv_solves = (int64_t)c;
    if (reset) hipMemset(h->trsv_count, 0, sizeof(c));
  }
  if (reset) memset(&h->stats, 0, sizeof(h->stats));
  return NNMPC_OK;
}

int nnmpc_qp_solve_batch(nnmpc_qp* h, int32_t B, const double* x0, const double* lb, const double* ub,
                         double* u, uint32_t* active, int32_t* status, int32_t* iters, int32_t ptr_kind) {
  return nnmpc_qp_solve_batch_ex(h, B, x0, lb, ub, nullptr, u, active, status, iters, ptr_kind, NNMPC_OUT_SEQUENCE);
}

int nnmpc_qp_solve_batch_warm(nnmpc_qp* h, int32_t B, const double* x0, const double* lb, const double* ub,
                              const uint8_t* guess, double* u, uint32_t* active, int32_t* status,
                              int32_t* iters, int32_t ptr_kind) {
  return nnmpc_qp_solve_batch_ex(h, B, x0, lb, ub, guess, u, active, status, iters, ptr_kind, NNMPC_OUT_SEQUENCE);
}

int nnmpc_qp_solve_batch_ex(nnmpc_qp* h, int32_t B, const double* x0, const double* lb, const double* ub,
                            const uint8_t* guess, double* u, uint32_t* active, int32_t* status,
                            int32_t* iters, int32_t ptr_kind, int32_t out_kind) {
  if (!h || B < 0 || !x0 || !lb || !ub || !u) { set_error("nnmpc_qp_solve_batch: bad arguments"); return NNMPC_EINVAL; }
  if (out_kind != NNMPC_OUT_SEQUENCE && out_kind != NNMPC_OUT_FIRST_MOVE) { set_error("nnmpc_qp_solve_batch_ex: bad out_kind %d", out_kind); return NNMPC_EINVAL; }
  if (B == 0) return NNMPC_OK;
  HIPCHK(hipSetDevice(h->device));
  const int G = h->seg_max;
  const int ncol = out_kind == NNMPC_OUT_FIRST_MOVE ? h->nu : h->n;   // columns of the caller's u
  h->ldu = ncol; h->nout = ncol;
  size_t e_tot0 = 0;
  if (h->profiling) { e_tot0 = ev_get(h); hipEventRecord(h->ev_pool[e_tot0], h->stream); }
  // device staging for the outputs when the caller hands host pointers (owned by the handle, grown on demand)
  double* u_stage = nullptr; uint32_t* a_stage = nullptr; int32_t* s_stage = nullptr; int32_t* i_stage = nullptr;
  unsigned char* g_stage = nullptr;
  const int gmax = std::min(G, (int)B);
  int rc = 0;
  if (ptr_kind == NNMPC_HOST) {
    if (!rc) rc = scratch_get(h, nnmpc_qp::SC_U, &u_stage, (size_t)gmax * ncol * sizeof(double));
    if (!rc && active) rc = scratch_get(h, nnmpc_qp::SC_ACT, &a_stage, (size_t)gmax * h->words * sizeof(uint32_t));
    if (!rc && status) rc = scratch_get(h, nnmpc_qp::SC_ST, &s_stage, (size_t)gmax * sizeof(int32_t));
    if (!rc && iters) rc = scratch_get(h, nnmpc_qp::SC_IT, &i_stage, (size_t)gmax * 2 * sizeof(int32_t));
    if (!rc && guess) rc = scratch_get(h, nnmpc_qp::SC_GUESS, &g_stage, (size_t)gmax * h->n);
  }
  hipError_t he = hipSuccess;
#define STEP_(x) do { if (!rc && he == hipSuccess) { he = (x); if (he != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(he)); rc = NNMPC_EHIP; } } } while (0)
  for (int b0 = 0; b0 < B && !rc; b0 += G) {
    const int nb = std::min(G, B - b0);
    const double *x0d, *lbd, *ubd;
    double* ud; uint32_t* ad; int32_t* sd; int32_t* idv;
    const unsigned char* gd = nullptr;
    if (ptr_kind == NNMPC_HOST) {
      if (guess) { STEP_(hipMemcpyAsync(g_stage, guess + (size_t)b0 * h->n, (size_t)nb * h->n, hipMemcpyHostToDevice, h->stream)); gd = g_stage; }
      STEP_(hipMemcpyAsync(h->in_stage, x0 + (size_t)b0 * h->n_aug, (size_t)nb * h->n_aug * 8, hipMemcpyHostToDevice, h->stream));
      STEP_(hipMemcpyAsync(h->lb_d, lb + (size_t)b0 * h->nu, (size_t)nb * h->nu * 8, hipMemcpyHostToDevice, h->stream));
      STEP_(hipMemcpyAsync(h->ub_d, ub + (size_t)b0 * h->nu, (size_t)nb * h->nu * 8, hipMemcpyHostToDevice, h->stream));
      x0d = h->in_stage; lbd = h->lb_d; ubd = h->ub_d;
      ud = u_stage; ad = active ? a_stage : nullptr; sd = status ? s_stage : nullptr; idv = iters ? i_stage : nullptr;
    } else {
      x0d = x0 + (size_t)b0 * h->n_aug; lbd = lb + (size_t)b0 * h->nu; ubd = ub + (size_t)b0 * h->nu;
      if (guess) gd = guess + (size_t)b0 * h->n;
      ud = u + (size_t)b0 * ncol;
      ad = active ? active + (size_t)b0 * h->words : nullptr;
      sd = status ? status + b0 : nullptr;
      idv = iters ? iters + 2 * (size_t)b0 : nullptr;
    }
    if (rc) break;
    if (h->have_inverse && h->opts.method != 1) rc = solve_segment_asm(h, nb, x0d, lbd, ubd, gd, ud, ad, sd, idv);
    else rc = solve_segment(h, nb, x0d, lbd, ubd, gd, ud, ad, sd, idv);
    if (!rc && ptr_kind == NNMPC_HOST) {
      STEP_(hipMemcpy(u + (size_t)b0 * ncol, u_stage, (size_t)nb * ncol * 8, hipMemcpyDeviceToHost));
      if (active) STEP_(hipMemcpy(active + (size_t)b0 * h->words, a_stage, (size_t)nb * h->words * 4, hipMemcpyDeviceToHost));
      if (status) STEP_(hipMemcpy(status + b0, s_stage, (size_t)nb * 4, hipMemcpyDeviceToHost));
      if (iters) STEP_(hipMemcpy(iters + 2 * (size_t)b0, i_stage, (size_t)nb * 8, hipMemcpyDeviceToHost));
    }
  }
#undef STEP_
  if (h->profiling) {
    size_t e1 = ev_get(h);
    hipEventRecord(h->ev_pool[e1], h->stream);
    stream_sync(h->stream);
    h->ev_recs.push_back({3, e_tot0, e1, 0.0});
    ev_collect(h);
  }
  return rc;
}

int nnmpc_qp_debug_factor_solve(nnmpc_qp* h, int32_t B, const float* dvec, const float* mask,
                                const float* rhs, float* sol) {
  if (!h || B <= 0 || B > h->slots || !dvec || !mask || !rhs || !sol) { set_error("debug_factor_solve: bad arguments"); return NNMPC_EINVAL; }
  HIPCHK(hipSetDevice(h->device));
  QpDev& d = h->d;
  const int rows = h->slots;
  std::vector<float> dv((size_t)rows * h->np, 0.f), mk((size_t)rows * h->np, 1.f), rh((size_t)rows * h->np, 0.f);
  std::vector<int> fl(rows, 0);
  for (int p = 0; p < B; ++p) {
    fl[p] = 1;
    for (int r = 0; r < h->n; ++r) {
      dv[(size_t)p * h->np + r] = dvec[(size_t)p * h->n + r];
      mk[(size_t)p * h->np + r] = mask[(size_t)p * h->n + r];
      rh[(size_t)p * h->np + r] = rhs[(size_t)p * h->n + r];
    }
  }
  HIPCHK(hipMemcpy(d.dvec, dv.data(), dv.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.mask, mk.data(), mk.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.rhs, rh.data(), rh.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d.f_factor, fl.data(), rows * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d.fail, 0, rows * 4));
  factor_dispatch(h, rows, B);
  solve_dispatch(h, rows, d.f_factor);
  HIPCHK(stream_sync(h->stream));
  HIPCHK(hipGetLastError());
  std::vector<float> so((size_t)rows * h->np);
  HIPCHK(hipMemcpy(so.data(), d.sol, so.size() * 4, hipMemcpyDeviceToHost));
  for (int p = 0; p < B; ++p)
    for (int r = 0; r < h->n; ++r) sol[(size_t)p * h->n + r] = so[(size_t)p * h->np + r];
  if (h->profiling) { stream_sync(h->stream); ev_collect(h); }
  return NNMPC_OK;
}

}  // extern "C"
