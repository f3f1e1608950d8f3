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
// therefore costs  |A|^3/3 + O(|A|^2)  per problem (dense fp64 Cholesky of the |A| x |A|
// block of H, in LDS) plus one row of the batched fp64 GEMM  LAM * H  -- no n^3 work at all.
// Problems whose set is too large, or that do not settle, fall back to the PDIP path.
//
// Per round (all unfinished problems of a segment in lock-step, 3 launches):
//   asm_lambda_k : compact A, gather S = H_AA, Cholesky, lam -> dense row of LAM
//   gemm_nt_f64  : XH = LAM * H                                   (MFMA f64)
//   asm_update_k : x = x_unc - XH (free), x = bound (active); fp64 KKT sign / feasibility
//                  tests -> new set, or finished
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nnmpc {

constexpr int ASM_MLDS = 192;      // largest active set factored in LDS (packed lower fp64)
enum { ASM_RUN = 0, ASM_DONE = 1, ASM_FALLBACK = 2 };

struct AsmDev {
  int n, np, nu, nseg;
  int max_active;                  // larger sets -> fallback
  int max_rounds;
  double bound_tol, stat_tol, pscale_unused;
  const double* H;                 // [np][np] fp64 inverse Hessian
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
  int* counters;                   // [0] still running, [1] big-set list length
  int* biglist;                    // [nseg] problems whose set does not fit LDS
  double* scratch;                 // [pool][max_active*(max_active+1)/2]
  // outputs (problem-indexed, may be null except u)
  double* u_out;
  uint32_t* act_out;
  int* status_out;
  int* iters_out;
  int words;
};

__device__ __forceinline__ size_t tri(int i, int j) { return (size_t)i * (i + 1) / 2 + j; }

// x_unc -> first active-set estimate: every bound the unconstrained minimiser violates.
__global__ __launch_bounds__(256) void asm_init_k(AsmDev d) {
  __shared__ int cnt[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  const size_t o = (size_t)p * d.np;
  if (p >= d.nseg) {                 // padding rows of the GEMM operands: never active
    if (tid == 0) d.state[p] = ASM_DONE;
    return;
  }
  int c = 0;
  for (int r = tid; r < d.n; r += 256) {
    const int k = r % d.nu;
    const double x = d.xunc[o + r];
    const double lb = d.lb[(size_t)p * d.nu + k], ub = d.ub[(size_t)p * d.nu + k];
    int s = x > ub ? 1 : (x < lb ? 2 : 0);
    if (d.guess) { s = d.guess[(size_t)p * d.n + r]; if (s > 2) s = 0; }
    d.st[(size_t)p * d.n + r] = (unsigned char)s;
    d.x[o + r] = s == 1 ? ub : (s == 2 ? lb : x);
    c += s != 0;
  }
  for (int r = d.n + tid; r < d.np; r += 256) d.x[o + r] = 0.0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((tid & 63) == 0) cnt[tid >> 6] = c;
  __syncthreads();
  if (tid == 0) {
    const int tot = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    d.rounds[p] = 0;
    // empty set: x = x_unc; done if that is feasible, which the first update round decides when the
    // set came from a caller's guess
    d.state[p] = (tot == 0 && !d.guess) ? ASM_DONE : ASM_RUN;
  }
}

// Dense fp64 Cholesky S = L L' (packed lower, in place) + solve S lam = r, one workgroup.
// Returns 0 on a non-positive pivot.
__device__ int asm_chol_solve(double* S, double* r, int m, int tid) {
  __shared__ int bad;
  if (tid == 0) bad = 0;
  const int lane = tid & 63, wave = tid >> 6;
  for (int c = 0; c < m; ++c) {
    __syncthreads();
    double piv = S[tri(c, c)];
    if (!(piv > 0.0)) { piv = 1.0; if (tid == 0) bad = 1; }
    const double sq = sqrt(piv), inv = 1.0 / sq;
    for (int i = c + 1 + tid; i < m; i += 256) S[tri(i, c)] *= inv;
    __syncthreads();
    if (tid == 0) S[tri(c, c)] = sq;
    for (int i = c + 1 + wave; i < m; i += 4) {
      const double lic = S[tri(i, c)];
      for (int j = c + 1 + lane; j <= i; j += 64) S[tri(i, j)] -= lic * S[tri(j, c)];
    }
  }
  __syncthreads();
  // triangular solves by wave 0 (wave-synchronous, no workgroup barriers)
  if (wave == 0) {
    for (int c = 0; c < m; ++c) {               // forward: L y = r
      const double yc = r[c] / S[tri(c, c)];
      for (int i = c + 1 + lane; i < m; i += 64) r[i] -= S[tri(i, c)] * yc;
      if (lane == 0) r[c] = yc;
    }
    for (int c = m - 1; c >= 0; --c) {          // backward: L' lam = y
      const double xc = r[c] / S[tri(c, c)];
      for (int i = lane; i < c; i += 64) r[i] -= S[tri(c, i)] * xc;
      if (lane == 0) r[c] = xc;
    }
  }
  __syncthreads();
  return !bad;
}

// One problem: compact A, r_A, S = H_AA, lam.   BIG = 0: S in LDS (m <= ASM_MLDS), problems with a
// larger set are queued; BIG = 1: persistent workgroups walk that queue with S in global scratch.
template <int BIG>
__global__ __launch_bounds__(256) void asm_lambda_k(AsmDev d) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int wsum[4];
  __shared__ int s_m;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* idx = reinterpret_cast<int*>(sm);                 // [max_active]
  double* rA = sm + (d.max_active + 1) / 2;              // [max_active]
  double* Sl = rA + d.max_active;                        // LDS S (BIG = 0)
  const int nbig = BIG ? d.counters[1] : 0;
  for (int it = blockIdx.x; BIG ? it < nbig : it == (int)blockIdx.x; it += gridDim.x) {
    const int p = BIG ? d.biglist[it] : it;
    if (!BIG && d.state[p] != ASM_RUN) return;
    const size_t o = (size_t)p * d.np;
    const unsigned char* st = d.st + (size_t)p * d.n;
    // ---- ordered compaction of the active indices
    const int per = (d.n + 255) / 256;
    const int r0 = tid * per, r1 = min(d.n, r0 + per);
    int c = 0;
    for (int r = r0; r < r1; ++r) c += st[r] != 0;
    int inc = c;
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = inc - c;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int m = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (m <= d.max_active) {
      int k = base;
      for (int r = r0; r < r1; ++r) if (st[r]) idx[k++] = r;
    }
    if (tid == 0) s_m = m;
    __syncthreads();
    if (m > d.max_active) {                     // too large for this path
      if (tid == 0) d.state[p] = ASM_FALLBACK;
      __syncthreads();
      continue;
    }
    if (!BIG && m > ASM_MLDS) {                 // defer to the global-scratch kernel
      if (tid == 0) d.biglist[atomicAdd(&d.counters[1], 1)] = p;
      return;
    }
    double* S = BIG ? d.scratch + (size_t)blockIdx.x * ((size_t)d.max_active * (d.max_active + 1) / 2) : Sl;
    for (int i = tid; i < m; i += 256) {
      const int a = idx[i], k = a % d.nu;
      const double b = st[a] == 1 ? d.ub[(size_t)p * d.nu + k] : d.lb[(size_t)p * d.nu + k];
      rA[i] = d.xunc[o + a] - b;
    }
    for (int i = wave; i < m; i += 4) {         // one wave per row of S: H[a_i][a_j], j <= i
      const double* Hr = d.H + (size_t)idx[i] * d.np;
      for (int j = lane; j <= i; j += 64) S[tri(i, j)] = Hr[idx[j]];
    }
    const int ok = asm_chol_solve(S, rA, m, tid);
    if (!ok) { if (tid == 0) d.state[p] = ASM_FALLBACK; __syncthreads(); continue; }
    // ---- dense multiplier row for the GEMM
    for (int r = tid; r < d.np; r += 256) d.lam[o + r] = 0.0;
    __syncthreads();
    for (int i = tid; i < m; i += 256) d.lam[o + idx[i]] = rA[i];
    __syncthreads();
  }
}

// x from the GEMM result, fp64 KKT tests, next active set.
__global__ __launch_bounds__(256) void asm_update_k(AsmDev d) {
  __shared__ int cnt[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  if (d.state[p] != ASM_RUN) return;
  const size_t o = (size_t)p * d.np;
  unsigned char* st = d.st + (size_t)p * d.n;
  int chg = 0;
  for (int r = tid; r < d.n; r += 256) {
    const int k = r % d.nu;
    const double lb = d.lb[(size_t)p * d.nu + k], ub = d.ub[(size_t)p * d.nu + k];
    const int s = st[r];
    if (s == 0) {
      const double x = d.xunc[o + r] - d.xh[o + r];
      d.x[o + r] = x;
      if (x > ub + d.bound_tol) { st[r] = 1; ++chg; }
      else if (x < lb - d.bound_tol) { st[r] = 2; ++chg; }
    } else {
      const double l = d.lam[o + r];
      d.x[o + r] = s == 1 ? ub : lb;
      if ((s == 1 && l <= 0.0) || (s == 2 && l >= 0.0)) { st[r] = 0; ++chg; }   // keep iff multiplier > 0
    }
  }
  for (int off = 32; off > 0; off >>= 1) chg += __shfl_xor(chg, off);
  if ((tid & 63) == 0) cnt[tid >> 6] = chg;
  __syncthreads();
  if (tid == 0) {
    const int tot = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    const int rd = d.rounds[p] + 1;
    d.rounds[p] = rd;
    if (tot == 0) d.state[p] = ASM_DONE;
    else if (rd >= d.max_rounds) d.state[p] = ASM_FALLBACK;
    else atomicAdd(&d.counters[0], 1);
  }
}

// Independent fp64 certification with P itself (px = x P):  stationarity on the free set,
// multiplier signs on the active set, feasibility; writes the outputs of finished problems.
__global__ __launch_bounds__(256) void asm_certify_k(AsmDev d, double gscale_min) {
  __shared__ int cnt[4];
  __shared__ double gq[4];
  const int p = blockIdx.x, tid = threadIdx.x;
  if (d.state[p] != ASM_DONE) {
    if (tid == 0 && d.status_out) d.status_out[p] = 3;   // marks "not solved here" for the caller
    return;
  }
  const size_t o = (size_t)p * d.np;
  const unsigned char* st = d.st + (size_t)p * d.n;
  int bad = 0;
  double qm = 0.0, gf = 0.0;
  for (int r = tid; r < d.n; r += 256) {
    const int k = r % d.nu;
    const double lb = d.lb[(size_t)p * d.nu + k], ub = d.ub[(size_t)p * d.nu + k];
    const double x = d.x[o + r], g = d.px[o + r] + d.q64[o + r];
    const int s = st[r];
    qm = fmax(qm, fabs(d.q64[o + r]));
    if (s == 0) { gf = fmax(gf, fabs(g)); bad += (x > ub + d.bound_tol) || (x < lb - d.bound_tol); }
    else if (s == 1) bad += g >= 0.0;
    else bad += g <= 0.0;
    d.u_out[(size_t)p * d.n + r] = x;
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
  const int ok = nbad == 0 && gfm <= d.stat_tol * fmax(gscale_min, qs);
  if (d.act_out) {
    const int m2 = 2 * d.n;
    for (int w = tid; w < d.words; w += 256) {
      uint32_t bits = 0;
      for (int b = 0; b < 32; ++b) {
        const int i = 32 * w + b;
        if (i < m2) {
          const int kk = i / (2 * d.nu), c = i % (2 * d.nu);
          const int hit = (c < d.nu) ? (st[kk * d.nu + c] == 1) : (st[kk * d.nu + c - d.nu] == 2);
          bits |= (uint32_t)hit << b;
        }
      }
      d.act_out[(size_t)p * d.words + w] = bits;
    }
  }
  if (tid == 0) {
    if (d.status_out) d.status_out[p] = ok ? 0 : 3;      // 3: let the PDIP path redo it
    if (d.iters_out) { d.iters_out[2 * p] = 0; d.iters_out[2 * p + 1] = 0; }   // no PDIP iterations, no n^3 factorisations
  }
}

// ---- gather / scatter of the problems handed to the PDIP fallback
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
__global__ void asm_scatter_k(double* u, uint32_t* act, int* status, int* iters, const double* uc,
                              const uint32_t* actc, const int* stc, const int* itc, const int* list, int cnt,
                              int n, int words) {
  const int i = blockIdx.x;
  if (i >= cnt) return;
  const int p = list[i];
  for (int k = threadIdx.x; k < n; k += blockDim.x) u[(size_t)p * n + k] = uc[(size_t)i * n + k];
  if (act) for (int k = threadIdx.x; k < words; k += blockDim.x) act[(size_t)p * words + k] = actc[(size_t)i * words + k];
  if (threadIdx.x == 0) {
    if (status) status[p] = stc[i];
    if (iters) { iters[2 * p] = itc[2 * i]; iters[2 * p + 1] = itc[2 * i + 1]; }
  }
}

}  // namespace nnmpc
