// First active sets by an accelerated projected-gradient iteration on the DUAL of the box QP, on the bf16 matrix pipes (gfx950).
//
// The primal-dual active-set rounds of qp_asm.h converge in a handful of rounds from ANY first set, but every round is a
// factorisation per problem, and from the classic start -- the bounds x_unc violates -- the CDU batch needs 4.7 of them
// (67 of ~115 bounds change in the first round, 29 in the second).  The dual of
//      min 1/2 x'Px + q'x,  lb <= x <= ub           is           min_mu  1/2 mu' H mu - mu' x_unc + sigma_box(mu),   H = P^-1,
// (mu = signed multipliers: > 0 upper, < 0 lower bound; sigma_box = support function of the box; x(mu) = x_unc - H mu), a
// strongly convex problem whose gradient is ONE row of a GEMM per problem and iteration -- with the same H for every problem.
// FISTA with the diagonal scaling t_j = 1 / (L H_jj), L = lambda_max(D^-1/2 H D^-1/2) over the window:
//      x   = x_unc - H y                                  (MFMA: Y[rows][W] x H[W][W], bf16 operands, f32 accumulate)
//      v   = y + t x,    mu+ = v - clip(v, t lb, t ub)     (the prox of t sigma_box: exact, elementwise)
//      y+  = mu+ + beta_k (mu+ - mu)                      (Nesterov momentum)
// Momentum with ADAPTIVE RESTART (O'Donoghue & Candes 2015, gradient scheme), by workgroup and one iteration late: when
// sum (y - nu+) . (nu+ - nu) > 0 over the workgroup's problems the momentum goes back to zero at the next iteration (Hamming distance
// of the CDU batch's predicted sets after 24 / 32 iterations: 2.2 / 0.94 plain, 1.4 / 0.18 with the restart; CPU emulation).
// The kernel iterates on the SCALED multiplier nu = mu / t (t is fixed, so it is the same iteration): with H' = H diag(t) as the MFMA
// operand,  x = x_unc - H' y,  w = y + x,  nu+ = w - clip(w, lb, ub),  y+ = nu+ + beta (nu+ - nu)  -- no multiplication by t left
// in the elementwise part, and a w inside the box gives an EXACT zero (w - w): the sign of nu is what names the set.
// does not have to CONVERGE: it only has to name the active set.  Measured on the CDU plant (CPU emulation, bf16 operands
// change nothing): Hamming distance of the predicted set from the final one 67 (x_unc start) -> 5.5 after 12 iterations, 2.8
// after 20, 0.9 after 30; rounds until the set stops moving 4.6 -> 2.3 / 2.1 / 1.6 (sx = 2); 5.9 -> 2.6 at sx = 4 (217 bounds).
// Nothing here enters a result: the rounds start from the predicted set instead of the violated one, every accepted answer is
// still an fp64 solve on its final set that passed the certificate.
//
// One workgroup (eight waves, two per SIMD) owns 64 problems for ALL iterations; per iteration only H streams in (bf16, the
// leading W x W block: 512 KB at W = 512 -- L2-resident, read once per workgroup and iteration, straight into MFMA operand
// registers in a fragment-major layout, four k-steps ahead); everything else stays on the chip:
//      Y      [64][W] bf16 in LDS (the MFMA operand every wave reads; rewritten in place by the epilogue)
//      state  x_unc (f16) | mu (bf16) packed in one register per element, acc: 4 x 4 MFMA tiles per wave (64 columns x 64 problems;
//             a four-wave version with 8 x 4 tiles per wave needed 128 state + 128 operand-ring registers next to the accumulators
//             and spilled ~500 of them)
// MFMA orientation: A operand = rows of H (output columns j), B operand = rows of Y (problems), so a lane holds FOUR
// CONSECUTIVE columns of one problem per tile (C layout: reg r of lane (li, lq) = column 4 lq + r, problem li).
// The K loop stops at the last column in which any of the 64 problems has a non-zero y (the multipliers live in the first
// MPC stages).  Cost per iteration and workgroup: 33.5 MFLOP on the matrix pipes (bf16 peak: 3.4 us), 512 KB from L2
// (64 B/clk: 3.4 us), ~1900 vector instructions per wave for the prox / momentum / packing (3.2 us).
#pragma once
#include "qp_asm.h"

namespace nnmpc {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int pu32x2 __attribute__((ext_vector_type(2)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pbf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 pf16x2 __attribute__((ext_vector_type(2)));

constexpr int PRED_NW = 8;                   // waves per workgroup
constexpr int PRED_MAXIT = 64;
// Two instances: <NT = 4 column tiles per wave, PT = 4 problem tiles>: window 512, 64 problems per workgroup (the CDU batch: its
// multipliers live in the first 480 columns); <8, 2>: window 1024, 32 problems per workgroup, for batches whose sets reach further
// (larger state spreads: sx >= 3) -- four times the L2 traffic per problem (2 MB of H' for 32 problems), chosen per call from the
// extent of the bounds x_unc violates (solve_segment_asm).
template <int NT> struct PredCfg {
  static constexpr int W = 16 * NT * PRED_NW;      // columns of the window
  static constexpr int KS = W / 32;                // k-steps of v_mfma_f32_16x16x32_bf16
  static constexpr int LDY = W + 8;                // bf16 per row of Y: 16 B of padding -> the 16 rows of a fragment read hit 64 distinct banks
};
constexpr int PRED_W = PredCfg<4>::W;        // the default window: 512 columns
constexpr int PRED_W2 = PredCfg<8>::W;       // the wide one: 1024

struct PredArgs {
  const pu32x4* Hf;      // [NW waves][KS][NT][64] fragments of bf16(H'[0:W, 0:W]), H' = H diag(t), t_k = 1 / (L H_kk): lane (li, lq) of (w, ks, jt) holds H'[16 (w NT + jt) + li][32 ks + 8 lq .. + 8]
                         // (the NT fragments a wave needs for one k-step are NT KB of consecutive memory: pred_frag_index)
  int iters;             // iterations; with `adaptive` the most a workgroup may take
  int* itsum;            // device counter: += iterations after the first of every workgroup (statistics: the executed flops)
  int fac10;             // adaptive: iterations = fac10 / 10 per bound x_unc violates (per problem, averaged over the workgroup)
  int adaptive;          // 1: every workgroup sets its own count from the bounds x_unc violates in its problems (see asm_predict_k)
  float beta[PRED_MAXIT];   // momentum of iteration k (beta[0] = 0)
};

template <int NT, int PT> __host__ __device__ constexpr int pred_lds_bytes(int nu) {
  return 16 * PT * PredCfg<NT>::LDY * 2 + 2 * 16 * PT * (nu + 4) * 4 + 16;      // (nu a multiple of 4; the last 16 bytes: the violation count, the restart sums)
}

template <int NT> __host__ __device__ constexpr size_t pred_frag_index(int jtg, int ks, int lane) {   // 16-byte fragment of column tile jtg (0 .. W / 16), k-step ks
  return (((size_t)(jtg / NT) * PredCfg<NT>::KS + ks) * NT + (jtg % NT)) * 64 + lane;
}

__device__ __forceinline__ unsigned pred_bf16(float x) {          // round to nearest even, as a 16-bit pattern
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// NU4: nu is a multiple of 4 -- a lane's four consecutive columns are four consecutive inputs of one stage, their bounds one 16-byte
// read; else (nu >= 4, e.g. the CSTRs plant's 6 inputs) every column looks its bounds up by itself.
template <int NT, int PT, bool NU4 = true>
__global__ __launch_bounds__(64 * PRED_NW, 2) void asm_predict_k(AsmDev d, PredArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
  constexpr int KS = PredCfg<NT>::KS, LDY = PredCfg<NT>::LDY, MR = 16 * PT;
  unsigned short* Y = reinterpret_cast<unsigned short*>(sm_raw);              // [MR][LDY] bf16 patterns
  const int ldb = d.nu + 4;                                                    // (rows 16 B aligned; 16 rows of a float4 read hit distinct banks for nu = 32)
  float* lbs = reinterpret_cast<float*>(Y + MR * LDY);                         // [MR][nu + 4]
  float* ubs = lbs + MR * ldb;
  int* nviol = reinterpret_cast<int*>(ubs + MR * ldb);                         // bounds x_unc violates, summed over the workgroup's problems
  float* rsum = reinterpret_cast<float*>(nviol + 1);                           // [2] restart test sums, double-buffered by iteration parity
  constexpr int NTH = 64 * PRED_NW;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int p0 = blockIdx.x * MR;
  for (int i = tid; i < MR * d.nu; i += NTH) {
    const int r = i / d.nu, k = i - r * d.nu;
    const size_t p = (size_t)min(p0 + r, d.nseg - 1);
    lbs[r * ldb + k] = (float)d.lb[p * d.nu + k];
    ubs[r * ldb + k] = (float)d.ub[p * d.nu + k];
  }
  for (int i = tid; i < MR * LDY / 2; i += NTH) reinterpret_cast<unsigned*>(Y)[i] = 0u;
  if (tid == 0) { *nviol = 0; rsum[0] = 0.f; rsum[1] = 0.f; }
  // ---- state, in pairs of consecutive columns (packed f32 arithmetic, v_cvt_pk_bf16_f32): sx = x_unc as two f16, sm = mu as two bf16;
  // pair (jt, pt, h): problem 16 pt + li, columns 16 (w NT + jt) + 4 lq + 2 h, + 1
  unsigned sx[NT][PT][2], sm[NT][PT][2];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const size_t p = (size_t)min(p0 + 16 * pt + li, d.nseg - 1);
      const double* xr = d.xunc + p * d.np + 16 * (w * NT + jt) + 4 * lq;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // (f16 range; a NaN stays a NaN: that problem is rejected elsewhere)
        const pf32x2 xu = {fminf(fmaxf((float)xr[2 * h], -60000.f), 60000.f), fminf(fmaxf((float)xr[2 * h + 1], -60000.f), 60000.f)};
        sx[jt][pt][h] = __builtin_bit_cast(unsigned, __builtin_convertvector(xu, pf16x2));
        sm[jt][pt][h] = 0u;
      }
    }
  int kc0[NT];                                                                 // input index of the first of a lane's four columns
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) kc0[jt] = (16 * (w * NT + jt) + 4 * lq) % d.nu;
  const bool st_words = (d.n & 3) == 0;
  const bool per32 = 32 % d.nu == 0;                                           // (uniform)
  __syncthreads();
  // this wave's fragments by BUFFER loads: the lane's 16 bytes are a loop-invariant vector offset, (w KS + ks) NT + jt KB go into the
  // scalar offset -- one address register for all loads (hipcc turned global loads into one hoisted 64-bit address PER LOAD: 128
  // registers here, spilled)
  const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(const_cast<pu32x4*>(a.Hf), 0, PRED_NW * KS * NT * 1024, 0x00020000);
  const int l16 = lane * 16;
  const int hw0 = w * (KS * NT * 1024);
  auto hload = [&](int ksn, int jt) __attribute__((always_inline)) {
    return __builtin_bit_cast(pu32x4, __builtin_amdgcn_raw_buffer_load_b128(rH, l16 + jt * 1024, hw0 + ksn * (NT * 1024), 0));   // (jt KB: the immediate)
  };
  // Iterations: the harder the batch, the more it takes to name its sets (CDU plant, Hamming distance <= 3 from the converged set: 12
  // iterations at 45 violated bounds per problem, 24 at 100, 40 at 140, 48 at 170, ~60 at 210) -- 0.3 per bound x_unc violates, by
  // workgroup (its 64 problems come from one batch), between 8 and a.iters.  The count is the number of non-zeros after iteration 0:
  // y = 0 there, so nu+ = x_unc - clip(x_unc, lb, ub).
  int nit = a.iters;
  int kb = 0;                                                                  // iterations since the momentum was last reset (index into beta)
  for (int it = 0; it < nit; ++it) {
    // ---- x = x_unc - H y on the matrix pipes: acc[jt][pt][r] = sum_k H[col][k] y[problem][k]
    if (tid == 0) rsum[it & 1] = 0.f;                                            // (this iteration's restart sum: last read an iteration ago; the barrier below orders it)
    pf32x4 acc[NT][PT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[jt][pt] = pf32x4{0.f, 0.f, 0.f, 0.f};
    // (all KS k-steps: the multipliers of the CDU batch reach column ~480 of 512 -- stopping at the last non-zero column of Y saved one
    // step in sixteen and cost a conditional per group of steps; iteration 0 has y = 0 and skips the phase)
#ifdef PRED_NO_MFMA
    if (false) {
#else
    if (it > 0) {
#endif
      constexpr int RD = 3;                                                      // ring of three k-steps of H fragments: loads two steps ahead
      pu32x4 hb[RD][NT];
      asm_sfor<0, RD - 1>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) hb[b][jt] = hload(b, jt);
      });
      // (straight-line code over the KS k-steps: every buffer index is a constant and the wait counts of the loads in flight are exact)
      asm_sfor<0, KS>([&](auto kc) __attribute__((always_inline)) {
        constexpr int ks = decltype(kc)::value, b = ks % RD;
        if constexpr (ks + RD - 1 < KS) {
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) hb[(ks + RD - 1) % RD][jt] = hload(ks + RD - 1, jt);
        }
        pu32x4 yf[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) yf[pt] = *reinterpret_cast<const pu32x4*>(Y + (16 * pt + li) * LDY + 32 * ks + 8 * lq);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt)
            acc[jt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf16x8, hb[b][jt]), __builtin_bit_cast(pbf16x8, yf[pt]),
                                                                  acc[jt][pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                                       // (one k-step per scheduling region: bounded live ranges)
      });
    }
    __syncthreads();                                                             // every wave is done reading Y
    // ---- prox step, momentum, Y rewritten in place (each lane owns its elements)
    // (restart: decided by the previous iteration's sum, read after that iteration's closing barrier; the other slot is cleared for the
    // iteration after this one -- it was last read an iteration ago)
    if (it > 0) kb = rsum[(it - 1) & 1] > 0.f ? 0 : kb + 1;
    const pf32x2 beta2 = {a.beta[kb], a.beta[kb]};
    pf32x2 rdot = {0.f, 0.f};
    // (per-lane bases of the LDS accesses below, hidden from the optimiser once per iteration: left alone it hoists ~50 loop-invariant
    // addresses out of the iteration loop and keeps them in scratch memory; from these bases every address is an immediate offset)
    // (the OFFSETS are hidden, not the pointers: a laundered pointer loses its address space and the accesses become flat loads)
    int yo = li * LDY + 16 * NT * w + 4 * lq, bo = li * ldb;
    asm volatile("" : "+v"(yo), "+v"(bo));
    unsigned short* Yl = Y + yo;
    const float* lbl = lbs + bo;
    const float* ubl = ubs + bo;
    int nv0 = 0;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      // (nu is a multiple of 4 and so is the lane's first column: its four columns are four consecutive inputs of one stage; when nu
      // divides 32 the column tiles jt and jt + 2 see the same inputs: their bounds are read once -- NT is even)
      pf32x4 lb4[2], ub4[2];
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if constexpr (NU4) {
          if (jt < 2 || !per32) {
            lb4[jt & 1] = *reinterpret_cast<const pf32x4*>(lbl + 16 * pt * ldb + kc0[jt]);
            ub4[jt & 1] = *reinterpret_cast<const pf32x4*>(ubl + 16 * pt * ldb + kc0[jt]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int k = kc0[jt] + r, kk = k >= d.nu ? k - d.nu : k;         // (nu >= 4: one wrap at most)
            lb4[jt & 1][r] = lbl[16 * pt * ldb + kk];
            ub4[jt & 1][r] = ubl[16 * pt * ldb + kk];
          }
        }
        unsigned short* yp = Yl + 16 * pt * LDY + 16 * jt;
        const pu32x2 yw = *reinterpret_cast<const pu32x2*>(yp);
        unsigned yb[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const pf32x2 xu = __builtin_convertvector(__builtin_bit_cast(pf16x2, sx[jt][pt][h]), pf32x2);
          const unsigned mw = sm[jt][pt][h], ywd = yw[h];
          const pf32x2 nu = {__uint_as_float(mw << 16), __uint_as_float(mw & 0xffff0000u)};
          const pf32x2 y = {__uint_as_float(ywd << 16), __uint_as_float(ywd & 0xffff0000u)};
          const pf32x2 ac = {acc[jt][pt][2 * h], acc[jt][pt][2 * h + 1]};
#ifdef PRED_NO_EPI
          const pf32x2 wv = xu - ac;
#else
          const pf32x2 wv = y + (xu - ac);
#endif
          // prox of sigma_box in the scaled variable:  w - clip(w, lb, ub)  (the clip as a median of three: one instruction; lb <= ub,
          // else the problem is rejected elsewhere)
          const pf32x2 cl = {__builtin_amdgcn_fmed3f(wv[0], lb4[jt & 1][2 * h], ub4[jt & 1][2 * h]),
                             __builtin_amdgcn_fmed3f(wv[1], lb4[jt & 1][2 * h + 1], ub4[jt & 1][2 * h + 1])};
          const pf32x2 nun = wv - cl;
          const pf32x2 dn = nun - nu;
          const pf32x2 yn = beta2 * dn + nun;
          rdot += (y - nun) * dn;
          yb[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(yn, pbf16x2));
          sm[jt][pt][h] = __builtin_bit_cast(unsigned, __builtin_convertvector(nun, pbf16x2));
        }
        *reinterpret_cast<pu32x2*>(yp) = pu32x2{yb[0], yb[1]};
      }
      __builtin_amdgcn_sched_barrier(0);                                         // (one row tile per scheduling region)
    }
    {
      float rd = rdot[0] + rdot[1];
      for (int off = 32; off > 0; off >>= 1) rd += __shfl_xor(rd, off);
      if (lane == 0) atomicAdd(&rsum[it & 1], rd);
    }
    if (it == 0 && a.adaptive) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
          for (int h = 0; h < 2; ++h) nv0 += ((sm[jt][pt][h] & 0x7fffu) != 0u) + ((sm[jt][pt][h] & 0x7fff0000u) != 0u);
      for (int off = 32; off > 0; off >>= 1) nv0 += __shfl_xor(nv0, off);
      if (lane == 0 && nv0) atomicAdd(nviol, nv0);
    }
    __syncthreads();                                                             // Y of the next iteration is complete
    if (it == 0 && a.adaptive) nit = min(a.iters, max(8, (a.fac10 * *nviol) / (10 * MR)));
    if (it == 0 && tid == 0 && a.itsum) atomicAdd(a.itsum, nit - 1);
  }
  // ---- the predicted set: bound states of the window from the sign of nu -- 1 upper (nu > 0), 2 lower (nu < 0), 0 free.  (From the
  // bf16 image the state holds: bf16 has the exponent range of f32, a non-zero nu stays non-zero; exact zeros come out of the projection.)
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int c0 = 16 * (w * NT + jt) + 4 * lq;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int p = p0 + 16 * pt + li;
      unsigned wd = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned hb16 = (sm[jt][pt][r >> 1] >> (16 * (r & 1))) & 0xffffu;
        wd |= ((hb16 & 0x7fffu) == 0u ? 0u : ((hb16 >> 15) ? 2u : 1u)) << (8 * r);
      }
      if (p < d.nseg && st_words && c0 + 3 < d.n) *reinterpret_cast<unsigned*>(d.st + (size_t)p * d.n + c0) = wd;
      else if (p < d.nseg) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (c0 + r < d.n) d.st[(size_t)p * d.n + c0 + r] = (unsigned char)((wd >> (8 * r)) & 0xff);
      }
    }
  }
}

// How far along the horizon do the bounds x_unc violates reach?  *cnt += problems with a violated bound in the columns [c0, c1).
// One wave per problem, sixteen problems per wave (one atomic per workgroup).
__global__ __launch_bounds__(256) void asm_extent_k(AsmDev d, int c0, int c1, int* cnt) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int hit = 0;
  for (int q = 0; q < 16; ++q) {
    const int p = blockIdx.x * 64 + wave * 16 + q;
    if (p >= d.nseg) break;
    int v = 0;
    for (int r = c0 + lane; r < c1; r += 64) {
      const int k = r % d.nu;
      const double x = d.xunc[(size_t)p * d.np + r];
      v |= x > d.ub[(size_t)p * d.nu + k] || x < d.lb[(size_t)p * d.nu + k];
    }
    hit += __any(v) ? 1 : 0;
  }
  if (lane == 0) wsum[wave] = hit;
  __syncthreads();
  if (threadIdx.x == 0) { const int t = wsum[0] + wsum[1] + wsum[2] + wsum[3]; if (t) atomicAdd(cnt, t); }
}

}  // namespace nnmpc
