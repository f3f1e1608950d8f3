// Device-resident lock-step chains and the batched steady-state target problems (gfx950).
//
//   nnmpc_chain_*   simulate_offline (lib/linearMPC.py:827-880) for all chains of a task at once: the loop :845-866 with
//                   the chain state, the target pairs, the disturbances and the recorded trajectories in HBM; the
//                   regulator QPs of a step are ONE call of nnmpc_qp_solve_batch_ex (first moves only, warm-started on
//                   the previous step's active set shifted by one stage).
//   nnmpc_ts_*      TargetSelector.solve (lib/linearMPC.py:298-311) for a batch of (ysp, dhat) pairs, reduced to the
//                   inputs on the host (see include/nnmpc.h); one wave per problem.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <chrono>
#include <vector>
#include "../../include/nnmpc.h"
#include "common.h"

using namespace nnmpc;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e_)); return NNMPC_EHIP; } } while (0)

namespace {

// ---- chain step, part 1: the regulator's inputs as get_control_sequence forms them (:682-689), and the records of
// the state before the move (:868-869).  One workgroup per chain.
__global__ __launch_bounds__(256) void chain_pre_k(int nx, int nu, const double* __restrict__ x, const double* __restrict__ uprev,
                                                   const double* __restrict__ xs, const double* __restrict__ us,
                                                   const double* __restrict__ ulb, const double* __restrict__ uub,
                                                   double* __restrict__ x0, double* __restrict__ lb, double* __restrict__ ub,
                                                   double* __restrict__ x_rec, double* __restrict__ uprev_rec) {
  const int c = blockIdx.x, tid = threadIdx.x, na = nx + nu;
  for (int i = tid; i < nx; i += 256) {
    const double xv = x[(size_t)c * nx + i];
    x0[(size_t)c * na + i] = xv - xs[(size_t)c * nx + i];
    x_rec[(size_t)c * nx + i] = xv;
  }
  for (int k = tid; k < nu; k += 256) {
    const double uv = uprev[(size_t)c * nu + k], s = us[(size_t)c * nu + k];
    x0[(size_t)c * na + nx + k] = uv - s;
    lb[(size_t)c * nu + k] = ulb[k] - s;
    ub[(size_t)c * nu + k] = uub[k] - s;
    uprev_rec[(size_t)c * nu + k] = uv;
  }
}

// ---- chain step, part 2: ut = useq[0:Nu] + us (:856, :689), x+ = A x + B ut + Bd d (:860), uprev+ = ut, and the next
// step's warm start: this step's active set shifted by one stage (the last stage repeats).
// Mt = [A'; B'; Bd'] ((nx + nu + nd) x nx, row-major): thread i accumulates x+[i] over consecutive rows -> coalesced.
__global__ __launch_bounds__(256) void chain_post_k(int nx, int nu, int nd, int n, int words, const double* __restrict__ Mt,
                                                    double* __restrict__ x, double* __restrict__ uprev,
                                                    const double* __restrict__ us, const double* __restrict__ dist,
                                                    const double* __restrict__ first, const uint32_t* __restrict__ act,
                                                    const int* __restrict__ status, double* __restrict__ u_rec,
                                                    int* __restrict__ st_rec, unsigned char* __restrict__ guess) {
  extern __shared__ double z[];                            // [nx + nu + nd]
  const int c = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < nx; i += 256) z[i] = x[(size_t)c * nx + i];
  for (int k = tid; k < nu; k += 256) {
    const double u = first[(size_t)c * nu + k] + us[(size_t)c * nu + k];
    z[nx + k] = u;
    u_rec[(size_t)c * nu + k] = u;
    uprev[(size_t)c * nu + k] = u;
  }
  for (int k = tid; k < nd; k += 256) z[nx + nu + k] = dist[(size_t)c * nd + k];
  if (tid == 0) st_rec[c] = status[c];
  __syncthreads();
  const int nz = nx + nu + nd;
  for (int i = tid; i < nx; i += 256) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    int j = 0;
    for (; j < nx; ++j) a0 += Mt[(size_t)j * nx + i] * z[j];          // A x
    for (; j < nx + nu; ++j) a1 += Mt[(size_t)j * nx + i] * z[j];     // B u
    for (; j < nz; ++j) a2 += Mt[(size_t)j * nx + i] * z[j];          // Bd d
    x[(size_t)c * nx + i] = (a0 + a1) + a2;
  }
  // bit k*2nu + j: upper bound of variable k nu + j active, bit k*2nu + nu + j: lower bound
  const uint32_t* aw = act + (size_t)c * words;
  for (int r = tid; r < n; r += 256) {
    const int rs = r + nu < n ? r + nu : r;                // the variable whose state variable r inherits
    const int k = rs / nu, j = rs - k * nu;
    const int bu = k * 2 * nu + j, bl = bu + nu;
    const int su = (aw[bu >> 5] >> (bu & 31)) & 1u, sl = (aw[bl >> 5] >> (bl & 31)) & 1u;
    guess[(size_t)c * n + r] = (unsigned char)(su ? 1 : (sl ? 2 : 0));
  }
}

// out[b][k] = u[b * ldu + k] + us[b][k], k < nu: the absolute first move of every sample (get_control_sequence adds the
// target input back, :689; simulate_offline keeps useq[0:Nu], :856)
__global__ void first_moves_k(double* __restrict__ out, const double* __restrict__ u, size_t ldu, const double* __restrict__ us,
                              int B, int nu) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * nu, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const size_t b = i / nu, k = i - b * nu;
    out[i] = u[b * ldu + k] + (us ? us[i] : 0.0);
  }
}

// ---- steady-state target problems: one wave per problem.
//   min 1/2 u'Pr u + q'u   s.t.  E u = e,  lb <= u <= ub      (nu variables, nz equalities, N = nu + nz <= 64)
// Goldfarb-Idnani dual active-set method (strictly convex: Pr > 0): start at the optimum of the equality-constrained
// problem, then repeatedly pick the most violated bound p and move along the direction that keeps stationarity and the
// working set W (the equalities + the bounds held so far) while the multiplier of p grows,
//     Pr z + E' v + sum_{i in W} s_i v_i e_i = -s_p e_p,   z_W = 0,   E z = 0          (s = +1 at an upper, -1 at a lower bound)
// -- ONE linear system in (z_free, v) per step, Gaussian elimination with partial pivoting, lane = row, matrix in LDS --
// until p is reached (full step: p joins W) or a multiplier of W hits zero first (that bound leaves W, same p again).
// Finite, never visits a working set with dependent constraints (a dependent p shows as z_p = 0: pure dual step), and
// reports infeasibility (no step possible) instead of cycling.  A final solve on the final set removes accumulated rounding
// before the KKT conditions are checked.
constexpr int TS_MAXIT = 600;

// K [N][ld] (column N = right-hand side) -> sol [N]; returns 1 (to all lanes) when a pivot vanishes
__device__ __forceinline__ int ts_ge_solve(double* K, int ld, int N, double* sol, int lane) {
  double kmax = 0.0;
  if (lane < N) for (int c = 0; c < N; ++c) kmax = fmax(kmax, fabs(K[(size_t)lane * ld + c]));
  for (int off = 32; off > 0; off >>= 1) kmax = fmax(kmax, __shfl_xor(kmax, off));
  __syncthreads();
  for (int k = 0; k < N; ++k) {
    double av = (lane >= k && lane < N) ? fabs(K[(size_t)lane * ld + k]) : -1.0;
    int ai = lane;
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_xor(av, off);
      const int oi = __shfl_xor(ai, off);
      if (ov > av || (ov == av && oi < ai)) { av = ov; ai = oi; }
    }
    if (!(av > 1e-13 * kmax)) return 1;                      // (wave-uniform)
    if (ai != k) {                                           // swap rows k and ai, columns k..N
      for (int c = k + lane; c <= N; c += 64) {
        const double a = K[(size_t)k * ld + c], b = K[(size_t)ai * ld + c];
        K[(size_t)k * ld + c] = b; K[(size_t)ai * ld + c] = a;
      }
    }
    __syncthreads();
    if (lane > k && lane < N) {
      double* row = K + (size_t)lane * ld;
      const double* pr = K + (size_t)k * ld;
      const double f = row[k] / pr[k];
      if (f != 0.0) for (int c = k + 1; c <= N; ++c) row[c] -= f * pr[c];
    }
    __syncthreads();
  }
  for (int k = N - 1; k >= 0; --k) {                         // back substitution, column oriented
    if (lane == k) sol[k] = K[(size_t)k * ld + N] / K[(size_t)k * ld + k];
    __syncthreads();
    if (lane < k) K[(size_t)lane * ld + N] -= K[(size_t)lane * ld + k] * sol[k];
    __syncthreads();
  }
  return 0;
}

// rows of the system for the current bound states: free input i: [Pr[i,:], E[:,i]'] = rf;  held input i: unit row = rh;
// equality k: [E[k,:], 0] = re
__device__ __forceinline__ void ts_assemble(double* K, int ld, int nu, int nz, const double* __restrict__ Pr, const double* __restrict__ E,
                                            int lane, int st, double rf, double rh, double re) {
  const int N = nu + nz;
  if (lane < nu) {
    double* row = K + (size_t)lane * ld;
    if (st == 0) {
      for (int c = 0; c < nu; ++c) row[c] = Pr[(size_t)lane * nu + c];
      for (int k = 0; k < nz; ++k) row[nu + k] = E[(size_t)k * nu + lane];
      row[N] = rf;
    } else {
      for (int c = 0; c < N; ++c) row[c] = c == lane ? 1.0 : 0.0;
      row[N] = rh;
    }
  } else if (lane < N) {
    double* row = K + (size_t)lane * ld;
    const int k = lane - nu;
    for (int c = 0; c < nu; ++c) row[c] = E[(size_t)k * nu + c];
    for (int c = nu; c < N; ++c) row[c] = 0.0;
    row[N] = re;
  }
  __syncthreads();
}

__global__ __launch_bounds__(64) void ts_solve_k(int B, int nu, int nz, const double* __restrict__ Pr, const double* __restrict__ E,
                                                 const double* __restrict__ lbv, const double* __restrict__ ubv,
                                                 const double* __restrict__ q, const double* __restrict__ e,
                                                 double* __restrict__ us, double* __restrict__ lam_eq,
                                                 unsigned char* __restrict__ active, int* __restrict__ status, double bound_tol) {
  extern __shared__ double sm[];
  const int p = blockIdx.x, lane = threadIdx.x;
  if (p >= B) return;
  const int N = nu + nz, ld = N + 2;
  double* K = sm;                                            // [N][ld]
  double* sol = K + (size_t)N * ld;                          // [N]
  const double* qp_ = q + (size_t)p * nu;
  const double* ep = e + (size_t)p * nz;
  const bool isu = lane < nu, isy = lane >= nu && lane < N;
  const double lbi = isu ? lbv[lane] : 0.0, ubi = isu ? ubv[lane] : 0.0;
  const double qi = isu ? qp_[lane] : 0.0;
  // inputs a comparison cannot reason about
  int invalid = 0;
  if (isu) invalid = !(lbi <= ubi) || !(fabs(qi) <= 1.79e308);
  if (lane < nz) invalid |= !(fabs(ep[lane]) <= 1.79e308);
  int result = NNMPC_ST_MAXITER;
  int st = 0;                                                // bound state of input `lane`: 0 free, 1 at ub, 2 at lb
  double x = 0.0, y = 0.0, mu = 0.0;                         // input value | equality multiplier | bound multiplier (>= 0)
  if (__any(invalid)) result = NNMPC_ST_NUMERIC;
  else {
    // ---- optimum under the equalities alone
    ts_assemble(K, ld, nu, nz, Pr, E, lane, 0, -qi, 0.0, isy ? ep[lane - nu] : 0.0);
    if (ts_ge_solve(K, ld, N, sol, lane)) result = NNMPC_ST_NUMERIC;     // E itself is rank deficient
    else {
      x = isu ? sol[lane] : 0.0;
      y = isy ? sol[lane] : 0.0;
      int it = 0;
      bool done = false;
      while (!done && it < TS_MAXIT) {
        // ---- most violated bound among the free inputs (ties: smallest index)
        double viol = (isu && st == 0) ? fmax(x - ubi, lbi - x) : -1e300;
        int pi = lane;
        for (int off = 32; off > 0; off >>= 1) {
          const double ov = __shfl_xor(viol, off);
          const int oi = __shfl_xor(pi, off);
          if (ov > viol || (ov == viol && oi < pi)) { viol = ov; pi = oi; }
        }
        if (!(viol > bound_tol)) { result = NNMPC_ST_OPTIMAL; break; }
        const double xp = __shfl(x, pi), ubp = __shfl(ubi, pi), lbp = __shfl(lbi, pi);   // (x of pi moves in the steps below)
        const double sp = xp > ubp ? 1.0 : -1.0, bp = xp > ubp ? ubp : lbp;
        // ---- steps towards bound pi
        for (;; ++it) {
          if (it >= TS_MAXIT) { done = true; break; }
          ts_assemble(K, ld, nu, nz, Pr, E, lane, st, lane == pi ? -sp : 0.0, 0.0, 0.0);
          if (ts_ge_solve(K, ld, N, sol, lane)) { result = NNMPC_ST_NUMERIC; done = true; break; }
          const double z = isu ? sol[lane] : 0.0, v = isy ? sol[lane] : 0.0;
          double vi = 0.0;                                   // rate of this input's bound multiplier
          if (isu && st != 0) {
            double g = 0.0;
            for (int c = 0; c < nu; ++c) g += Pr[(size_t)lane * nu + c] * sol[c];
            for (int k = 0; k < nz; ++k) g += E[(size_t)k * nu + lane] * sol[nu + k];
            vi = st == 1 ? -g : g;
          }
          const double zp = __shfl(z, pi), xpc = __shfl(x, pi);
          // a bound that depends on the working set shows as z = 0 (up to rounding; an independent one moves its input by
          // the order of 1 / Pr_pp per unit multiplier): no primal step possible then, only multipliers move
          const double t2 = (fabs(zp) * Pr[(size_t)pi * nu + pi] > 1e-9 && (bp - xpc) / zp > 0.0) ? (bp - xpc) / zp : 1e300;
          double t1 = (isu && st != 0 && vi < -1e-14) ? mu / -vi : 1e300;
          int kb = lane;
          for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(t1, off);
            const int oi = __shfl_xor(kb, off);
            if (ov < t1 || (ov == t1 && oi < kb)) { t1 = ov; kb = oi; }
          }
          const double t = fmin(t1, t2);
          if (!(t < 1e299)) { result = NNMPC_ST_NUMERIC; done = true; break; }    // no steady state inside the input box
          if (isu) { if (st == 0) x += t * z; else mu += t * vi; }
          if (isy) y += t * v;
          if (lane == pi) mu += t;
          if (t2 <= t1) {                                    // full step: the bound joins the working set
            if (lane == pi) { st = sp > 0.0 ? 1 : 2; x = bp; }
            ++it;
            break;
          }
          if (lane == kb) { st = 0; mu = 0.0; }              // its multiplier reached zero: the bound leaves, same p again
        }
      }
      if (result == NNMPC_ST_OPTIMAL) {
        // ---- final solve on the final set, then the KKT conditions in full
        ts_assemble(K, ld, nu, nz, Pr, E, lane, st, -qi, st == 1 ? ubi : lbi, isy ? ep[lane - nu] : 0.0);
        if (ts_ge_solve(K, ld, N, sol, lane)) result = NNMPC_ST_NUMERIC;
        else {
          x = isu ? sol[lane] : 0.0;
          y = isy ? sol[lane] : 0.0;
          int bad = 0;
          double scale = 1.0;
          if (isu) {
            double g = qi;
            for (int c = 0; c < nu; ++c) g += Pr[(size_t)lane * nu + c] * sol[c];
            for (int k = 0; k < nz; ++k) g += E[(size_t)k * nu + lane] * sol[nu + k];
            scale = fmax(1.0, fabs(qi));
            if (st == 0) bad = !(fabs(g) <= 1e-8 * scale) || !(x <= ubi + bound_tol) || !(x >= lbi - bound_tol);
            else bad = st == 1 ? !(g <= 1e-9 * scale) : !(g >= -1e-9 * scale);     // multiplier -s g >= 0 (zero: weakly active)
            if (st == 1) x = ubi; else if (st == 2) x = lbi;
          }
          if (isy) {
            double r = -ep[lane - nu];
            for (int c = 0; c < nu; ++c) r += E[(size_t)(lane - nu) * nu + c] * sol[c];
            bad = !(fabs(r) <= 1e-9 * fmax(1.0, fabs(ep[lane - nu])));
          }
          if (__any(bad)) result = NNMPC_ST_MAXITER;
        }
      }
    }
  }
  if (isu) {
    us[(size_t)p * nu + lane] = result == NNMPC_ST_NUMERIC ? __longlong_as_double(0x7ff8000000000000ll) : x;
    if (active) active[(size_t)p * nu + lane] = result == NNMPC_ST_NUMERIC ? 0 : (unsigned char)st;
  }
  if (lam_eq && isy) lam_eq[(size_t)p * nz + lane - nu] = y;
  if (lane == 0) status[p] = result;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
struct nnmpc_chain {
  nnmpc_qp* qp = nullptr;
  int device = 0, nc = 0, nx = 0, nu = 0, nd = 0, n = 0, words = 0;
  double *Mt = nullptr, *ulb = nullptr, *uub = nullptr, *x0 = nullptr, *uprev0 = nullptr;   // shared data
  double *x = nullptr, *uprev = nullptr;                   // chain state [nc][nx], [nc][nu]
  double *qx0 = nullptr, *lb = nullptr, *ub = nullptr, *first = nullptr;
  uint32_t* act = nullptr;
  int* status = nullptr;
  unsigned char* guess = nullptr;
  bool have_guess = false;
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double total_ms = 0.0, solve_ms = 0.0;
  std::vector<void*> allocs;
  // grow-only staging for host-pointer runs
  void* stage[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t stage_cap[7] = {0, 0, 0, 0, 0, 0, 0};
};

namespace {
template <class T>
int chain_alloc(nnmpc_chain* c, T** p, size_t count) {
  void* q = nullptr;
  const hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
  if (e != hipSuccess) { set_error("hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e)); return NNMPC_ENOMEM; }
  hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(T));
  c->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}
template <class T>
int chain_stage(nnmpc_chain* c, int which, T** out, size_t bytes) {
  if (c->stage_cap[which] < bytes) {
    if (c->stage[which]) { hipFree(c->stage[which]); c->stage[which] = nullptr; c->stage_cap[which] = 0; }
    const hipError_t e = hipMalloc(&c->stage[which], bytes + 256);
    if (e != hipSuccess) { c->stage[which] = nullptr; set_error("hipMalloc(%zu bytes of staging): %s", bytes, hipGetErrorString(e)); return NNMPC_ENOMEM; }
    c->stage_cap[which] = bytes + 256;
  }
  *out = (T*)c->stage[which];
  return 0;
}
}  // namespace

extern "C" {

int nnmpc_chain_create(nnmpc_chain** out, nnmpc_qp* qp, int32_t nc, int32_t nx, int32_t nu, int32_t nd,
                       const double* A, const double* B, const double* Bd, const double* ulb,
                       const double* uub, const double* x0, const double* uprev0) {
  if (!out || !qp || nc <= 0 || nx <= 0 || nu <= 0 || nd < 0 || !A || !B || (nd && !Bd) || !ulb || !uub || !x0 || !uprev0) {
    set_error("nnmpc_chain_create: bad arguments (nc=%d nx=%d nu=%d nd=%d)", nc, nx, nu, nd);
    return NNMPC_EINVAL;
  }
  int n = 0, qnu = 0, qnaug = 0;
  if (nnmpc_qp_dims(qp, &n, &qnu, &qnaug) != NNMPC_OK) return NNMPC_EINVAL;
  if (qnu != nu || qnaug != nx + nu) {
    set_error("nnmpc_chain_create: regulator has nu=%d n_aug=%d, chains need nu=%d n_aug=nx+nu=%d", qnu, qnaug, nu, nx + nu);
    return NNMPC_EINVAL;
  }
  nnmpc_chain* c = new nnmpc_chain();
  c->qp = qp; c->nc = nc; c->nx = nx; c->nu = nu; c->nd = nd; c->n = n; c->words = (2 * n + 31) / 32;
  if (hipGetDevice(&c->device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
      hipEventCreate(&c->e0) != hipSuccess || hipEventCreate(&c->e1) != hipSuccess) {
    set_error("nnmpc_chain_create: no HIP device / stream"); nnmpc_chain_destroy(c); return NNMPC_EHIP;
  }
  const int nzz = nx + nu + nd;
  int rc = 0;
#define A_(ptr, cnt) if (!rc) rc = chain_alloc(c, &(ptr), (size_t)(cnt))
  A_(c->Mt, (size_t)nzz * nx); A_(c->ulb, nu); A_(c->uub, nu); A_(c->x0, nx); A_(c->uprev0, nu);
  A_(c->x, (size_t)nc * nx); A_(c->uprev, (size_t)nc * nu);
  A_(c->qx0, (size_t)nc * (nx + nu)); A_(c->lb, (size_t)nc * nu); A_(c->ub, (size_t)nc * nu); A_(c->first, (size_t)nc * nu);
  A_(c->act, (size_t)nc * c->words); A_(c->status, nc); A_(c->guess, (size_t)nc * n);
#undef A_
  if (rc) { nnmpc_chain_destroy(c); return rc; }
  std::vector<double> mt((size_t)nzz * nx);
  for (int i = 0; i < nx; ++i) {
    for (int j = 0; j < nx; ++j) mt[(size_t)j * nx + i] = A[(size_t)i * nx + j];
    for (int j = 0; j < nu; ++j) mt[(size_t)(nx + j) * nx + i] = B[(size_t)i * nu + j];
    for (int j = 0; j < nd; ++j) mt[(size_t)(nx + nu + j) * nx + i] = Bd[(size_t)i * nd + j];
  }
  hipError_t e = hipMemcpy(c->Mt, mt.data(), mt.size() * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->ulb, ulb, nu * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->uub, uub, nu * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->x0, x0, nx * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(c->uprev0, uprev0, nu * 8, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("nnmpc_chain_create: upload failed: %s", hipGetErrorString(e)); nnmpc_chain_destroy(c); return NNMPC_EHIP; }
  rc = nnmpc_chain_reset(c);
  if (rc) { nnmpc_chain_destroy(c); return rc; }
  *out = c;
  return NNMPC_OK;
}

int nnmpc_chain_destroy(nnmpc_chain* c) {
  if (!c) return NNMPC_OK;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (void* p : c->allocs) hipFree(p);
  for (void* p : c->stage) if (p) hipFree(p);
  if (c->e0) hipEventDestroy(c->e0);
  if (c->e1) hipEventDestroy(c->e1);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return NNMPC_OK;
}

int nnmpc_chain_reset(nnmpc_chain* c) {
  if (!c) { set_error("nnmpc_chain_reset: null handle"); return NNMPC_EINVAL; }
  HIPCHK(hipSetDevice(c->device));
  for (int k = 0; k < c->nc; ++k) {
    HIPCHK(hipMemcpyAsync(c->x + (size_t)k * c->nx, c->x0, c->nx * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->uprev + (size_t)k * c->nu, c->uprev0, c->nu * 8, hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(stream_sync(c->stream));
  c->have_guess = false;
  return NNMPC_OK;
}

int nnmpc_chain_run(nnmpc_chain* c, int32_t T, const double* xs, const double* us, const double* d,
                    double* x_rec, double* uprev_rec, double* u_rec, int32_t* status,
                    int32_t warm_start, int32_t ptr_kind) {
  if (!c || T < 0 || !xs || !us || (c->nd && !d) || !x_rec || !uprev_rec || !u_rec || !status) {
    set_error("nnmpc_chain_run: bad arguments");
    return NNMPC_EINVAL;
  }
  if (T == 0) return NNMPC_OK;
  HIPCHK(hipSetDevice(c->device));
  const int nc = c->nc, nx = c->nx, nu = c->nu, nd = c->nd;
  const size_t sx = (size_t)T * nc * nx, su = (size_t)T * nc * nu, sd = (size_t)T * nc * nd, ss = (size_t)T * nc;
  const double *xs_d = xs, *us_d = us, *d_d = d;
  double *xr_d = x_rec, *ur_d = uprev_rec, *uu_d = u_rec;
  int* st_d = status;
  if (ptr_kind == NNMPC_HOST) {
    double *a = nullptr, *b = nullptr, *cc = nullptr;
    int rc = chain_stage(c, 0, &a, sx * 8);
    if (!rc) rc = chain_stage(c, 1, &b, su * 8);
    if (!rc) rc = chain_stage(c, 2, &cc, std::max<size_t>(sd, 1) * 8);
    if (!rc) rc = chain_stage(c, 3, &xr_d, sx * 8);
    if (!rc) rc = chain_stage(c, 4, &ur_d, su * 8);
    if (!rc) rc = chain_stage(c, 5, &uu_d, su * 8);
    if (!rc) rc = chain_stage(c, 6, &st_d, ss * 4);
    if (rc) return rc;
    HIPCHK(hipMemcpy(a, xs, sx * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b, us, su * 8, hipMemcpyHostToDevice));
    if (sd) HIPCHK(hipMemcpy(cc, d, sd * 8, hipMemcpyHostToDevice));
    xs_d = a; us_d = b; d_d = cc;
  }
  // the step's own kernels go on the regulator handle's stream: pre -> solve -> post in stream order, no host wait in between
  // (they used to run on the chain's stream behind a host synchronisation per step)
  hipStream_t qs = nnmpc_qp_stream_internal(c->qp);
  HIPCHK(stream_sync(c->stream));                            // (a reset's copies)
  HIPCHK(hipEventRecord(c->e0, qs));
  double solve_s = 0.0;
  const size_t lds = (size_t)(nx + nu + nd) * sizeof(double);
  for (int t = 0; t < T; ++t) {
    const double* xs_t = xs_d + (size_t)t * nc * nx;
    const double* us_t = us_d + (size_t)t * nc * nu;
    const double* d_t = nd ? d_d + (size_t)t * nc * nd : nullptr;
    hipLaunchKernelGGL(chain_pre_k, dim3(nc), dim3(256), 0, qs, nx, nu, c->x, c->uprev, xs_t, us_t, c->ulb, c->uub,
                       c->qx0, c->lb, c->ub, xr_d + (size_t)t * nc * nx, ur_d + (size_t)t * nc * nu);
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = nnmpc_qp_solve_batch_ex(c->qp, nc, c->qx0, c->lb, c->ub, (warm_start && c->have_guess) ? c->guess : nullptr,
                                           c->first, c->act, c->status, nullptr, NNMPC_DEVICE, NNMPC_OUT_FIRST_MOVE);
    solve_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc) {
      // the chain state is half advanced (x, uprev of step t recorded, not stepped): drain the stream and drop the warm-start
      // guess so that a later call starts from a consistent state (the caller should nnmpc_chain_reset)
      hipStreamSynchronize(qs);
      c->have_guess = false;
      return rc;
    }
    hipLaunchKernelGGL(chain_post_k, dim3(nc), dim3(256), lds, qs, nx, nu, nd, c->n, c->words, c->Mt, c->x, c->uprev,
                       us_t, d_t, c->first, c->act, c->status, uu_d + (size_t)t * nc * nu, st_d + (size_t)t * nc, c->guess);
    c->have_guess = true;
  }
  HIPCHK(hipEventRecord(c->e1, qs));
  HIPCHK(stream_sync(qs));
  HIPCHK(hipGetLastError());
  float ms = 0.f;
  hipEventElapsedTime(&ms, c->e0, c->e1);
  c->total_ms = ms; c->solve_ms = 1e3 * solve_s;
  if (ptr_kind == NNMPC_HOST) {
    HIPCHK(hipMemcpy(x_rec, xr_d, sx * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(uprev_rec, ur_d, su * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(u_rec, uu_d, su * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status, st_d, ss * 4, hipMemcpyDeviceToHost));
  }
  return NNMPC_OK;
}

int nnmpc_qp_first_moves(const double* u, int64_t ldu, const double* us, int32_t B, int32_t nu, double* out) {
  if (!u || !out || B < 0 || nu <= 0 || ldu < nu) { set_error("nnmpc_qp_first_moves: bad arguments"); return NNMPC_EINVAL; }
  if (B == 0) return NNMPC_OK;
  const size_t total = (size_t)B * nu;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(first_moves_k, dim3(grid), dim3(256), 0, 0, out, u, (size_t)ldu, us, B, nu);
  HIPCHK(stream_sync(0));      // (polling: common.h)
  return NNMPC_OK;
}

int nnmpc_chain_last_ms(nnmpc_chain* c, double* total_ms, double* solve_ms) {
  if (!c) return NNMPC_EINVAL;
  if (total_ms) *total_ms = c->total_ms;
  if (solve_ms) *solve_ms = c->solve_ms;
  return NNMPC_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
struct nnmpc_ts {
  int device = 0, nu = 0, nz = 0;
  double *Pr = nullptr, *E = nullptr, *lb = nullptr, *ub = nullptr;
  hipStream_t stream = nullptr;
  void* stage[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t stage_cap[6] = {0, 0, 0, 0, 0, 0};
};

namespace {
template <class T>
int ts_stage(nnmpc_ts* h, int which, T** out, size_t bytes) {
  if (h->stage_cap[which] < bytes) {
    if (h->stage[which]) { hipFree(h->stage[which]); h->stage[which] = nullptr; h->stage_cap[which] = 0; }
    const hipError_t e = hipMalloc(&h->stage[which], bytes + bytes / 4 + 256);
    if (e != hipSuccess) { h->stage[which] = nullptr; set_error("hipMalloc(%zu bytes of staging): %s", bytes, hipGetErrorString(e)); return NNMPC_ENOMEM; }
    h->stage_cap[which] = bytes + bytes / 4 + 256;
  }
  *out = (T*)h->stage[which];
  return 0;
}
}  // namespace

extern "C" {

int nnmpc_ts_create(nnmpc_ts** out, int32_t nu, int32_t nz, const double* Pr, const double* E,
                    const double* lb, const double* ub) {
  if (!out || nu <= 0 || nz < 0 || nu + nz > 64 || !Pr || (nz && !E) || !lb || !ub) {
    set_error("nnmpc_ts_create: bad arguments (nu=%d nz=%d; nu + nz <= 64)", nu, nz);
    return NNMPC_EINVAL;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    set_error("nnmpc_ts_create: no HIP device available (this library has no CPU fallback)");
    return NNMPC_EHIP;
  }
  nnmpc_ts* h = new nnmpc_ts();
  h->nu = nu; h->nz = nz;
  hipGetDevice(&h->device);
  hipError_t e = hipStreamCreate(&h->stream);
  if (e == hipSuccess) e = hipMalloc((void**)&h->Pr, (size_t)nu * nu * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&h->E, std::max<size_t>((size_t)nz * nu, 1) * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&h->lb, nu * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&h->ub, nu * 8);
  if (e == hipSuccess) e = hipMemcpy(h->Pr, Pr, (size_t)nu * nu * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess && nz) e = hipMemcpy(h->E, E, (size_t)nz * nu * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->lb, lb, nu * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->ub, ub, nu * 8, hipMemcpyHostToDevice);
  const int N = nu + nz;
  const size_t lds = ((size_t)N * (N + 2) + N) * sizeof(double);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ts_solve_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("nnmpc_ts_create: %s", hipGetErrorString(e)); nnmpc_ts_destroy(h); return NNMPC_EHIP; }
  *out = h;
  return NNMPC_OK;
}

int nnmpc_ts_destroy(nnmpc_ts* h) {
  if (!h) return NNMPC_OK;
  hipSetDevice(h->device);
  hipDeviceSynchronize();
  for (void* p : {(void*)h->Pr, (void*)h->E, (void*)h->lb, (void*)h->ub}) if (p) hipFree(p);
  for (void* p : h->stage) if (p) hipFree(p);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return NNMPC_OK;
}

int nnmpc_ts_solve_batch(nnmpc_ts* h, int32_t B, const double* q, const double* e, double* us,
                         double* lam_eq, uint8_t* active, int32_t* status, int32_t ptr_kind) {
  if (!h || B < 0 || !q || (h->nz && !e) || !us || !status) { set_error("nnmpc_ts_solve_batch: bad arguments"); return NNMPC_EINVAL; }
  if (B == 0) return NNMPC_OK;
  HIPCHK(hipSetDevice(h->device));
  const int nu = h->nu, nz = h->nz, N = nu + nz;
  const double *qd = q, *ed = e;
  double *ud = us, *ld = lam_eq;
  unsigned char* ad = active;
  int* sd = status;
  if (ptr_kind == NNMPC_HOST) {
    double *a = nullptr, *b = nullptr;
    int rc = ts_stage(h, 0, &a, (size_t)B * nu * 8);
    if (!rc) rc = ts_stage(h, 1, &b, std::max<size_t>((size_t)B * nz, 1) * 8);
    if (!rc) rc = ts_stage(h, 2, &ud, (size_t)B * nu * 8);
    if (!rc && lam_eq) rc = ts_stage(h, 3, &ld, std::max<size_t>((size_t)B * nz, 1) * 8);
    if (!rc && active) rc = ts_stage(h, 4, &ad, (size_t)B * nu);
    if (!rc) rc = ts_stage(h, 5, &sd, (size_t)B * 4);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(a, q, (size_t)B * nu * 8, hipMemcpyHostToDevice, h->stream));
    if (nz) HIPCHK(hipMemcpyAsync(b, e, (size_t)B * nz * 8, hipMemcpyHostToDevice, h->stream));
    qd = a; ed = b;
  }
  const size_t lds = ((size_t)N * (N + 2) + N) * sizeof(double);
  hipLaunchKernelGGL(ts_solve_k, dim3(B), dim3(64), lds, h->stream, B, nu, nz, h->Pr, h->E, h->lb, h->ub, qd, ed, ud, ld, ad, sd, 1e-9);
  if (ptr_kind == NNMPC_HOST) {
    HIPCHK(hipMemcpyAsync(us, ud, (size_t)B * nu * 8, hipMemcpyDeviceToHost, h->stream));
    if (lam_eq && nz) HIPCHK(hipMemcpyAsync(lam_eq, ld, (size_t)B * nz * 8, hipMemcpyDeviceToHost, h->stream));
    if (active) HIPCHK(hipMemcpyAsync(active, ad, (size_t)B * nu, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(status, sd, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(stream_sync(h->stream));
  HIPCHK(hipGetLastError());
  return NNMPC_OK;
}

}  // extern "C"
