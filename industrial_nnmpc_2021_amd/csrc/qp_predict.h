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

constexpr int PRED_NT = 4;                   // 16-column tiles per wave
constexpr int PRED_NW = 8;                   // waves per workgroup
constexpr int PRED_W = 16 * PRED_NT * PRED_NW;   // 512 columns
constexpr int PRED_KS = PRED_W / 32;         // k-steps of v_mfma_f32_16x16x32_bf16
constexpr int PRED_LDY = PRED_W + 8;         // bf16 per row of Y: 16 B of padding -> the 16 rows of a fragment read hit 64 distinct banks
constexpr int PRED_MAXIT = 64;

struct PredArgs {
  const pu32x4* Hf;      // [NW waves][KS][NT][64] fragments of bf16(H[0:W, 0:W]): lane (li, lq) of (w, ks, jt) holds H[16 (w NT + jt) + li][32 ks + 8 lq .. + 8]
                         // (the NT fragments a wave needs for one k-step are 4 KB of consecutive memory: pred_frag_index)
  const float* tt;       // [2][W]: t_j = 1 / (L H_jj), then 1 / t_j
  int iters;
  float beta[PRED_MAXIT];   // momentum of iteration k (beta[0] = 0)
};

__host__ __device__ constexpr int pred_lds_bytes(int nu) {
  return 64 * PRED_LDY * 2 + 2 * PRED_W * 4 + 2 * 64 * (nu + 4) * 4 + 16;      // (nu a multiple of 4)
}

__host__ __device__ constexpr size_t pred_frag_index(int jtg, int ks, int lane) {   // 16-byte fragment of column tile jtg (0 .. W / 16), k-step ks
  return (((size_t)(jtg / PRED_NT) * PRED_KS + ks) * PRED_NT + (jtg % PRED_NT)) * 64 + lane;
}

__device__ __forceinline__ unsigned pred_bf16(float x) {          // round to nearest even, as a 16-bit pattern
  const unsigned u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

__global__ __launch_bounds__(64 * PRED_NW, 2) void asm_predict_k(AsmDev d, PredArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_raw[];
  constexpr int NT = PRED_NT, W = PRED_W, KS = PRED_KS, LDY = PRED_LDY;
  unsigned short* Y = reinterpret_cast<unsigned short*>(sm_raw);              // [64][LDY] bf16 patterns
  float* tl = reinterpret_cast<float*>(Y + 64 * LDY);                          // [2][W]
  const int ldb = d.nu + 4;                                                    // (rows 16 B aligned; 16 rows of a float4 read hit distinct banks for nu = 32)
  float* lbs = tl + 2 * W;                                                     // [64][nu + 4]
  float* ubs = lbs + 64 * ldb;
  int* kmx = reinterpret_cast<int*>(ubs + 64 * ldb);                           // [2] last non-zero column of Y (double-buffered by iteration parity)
  constexpr int NTH = 64 * PRED_NW;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int p0 = blockIdx.x * 64;
  for (int i = tid; i < 2 * W; i += NTH) tl[i] = a.tt[i];
  for (int i = tid; i < 64 * d.nu; i += NTH) {
    const int r = i / d.nu, k = i - r * d.nu;
    const size_t p = (size_t)min(p0 + r, d.nseg - 1);
    lbs[r * ldb + k] = (float)d.lb[p * d.nu + k];
    ubs[r * ldb + k] = (float)d.ub[p * d.nu + k];
  }
  for (int i = tid; i < 64 * LDY / 2; i += NTH) reinterpret_cast<unsigned*>(Y)[i] = 0u;
  if (tid < 2) kmx[tid] = -1;
  // ---- state: x_unc (f16, high half) | mu (bf16, low half); element (jt, pt, r): problem 16 pt + li, column 16 (w NT + jt) + 4 lq + r
  unsigned s[NT][4][4];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const size_t p = (size_t)min(p0 + 16 * pt + li, d.nseg - 1);
      const double* xr = d.xunc + p * d.np + 16 * (w * NT + jt) + 4 * lq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xu = fminf(fmaxf((float)xr[r], -60000.f), 60000.f);          // (f16 range; a NaN stays a NaN: that problem is rejected elsewhere)
        s[jt][pt][r] = ((unsigned)__builtin_bit_cast(unsigned short, (_Float16)xu) << 16);
      }
    }
  int kc0[NT];                                                                 // input index of the first of a lane's four columns
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) kc0[jt] = (16 * (w * NT + jt) + 4 * lq) % d.nu;
  const bool st_words = (d.n & 3) == 0;
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
  for (int it = 0; it < a.iters; ++it) {
    // ---- x = x_unc - H y on the matrix pipes: acc[jt][pt][r] = sum_k H[col][k] y[problem][k]
    pf32x4 acc[NT][4];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) acc[jt][pt] = pf32x4{0.f, 0.f, 0.f, 0.f};
    const int kl = kmx[it & 1];                                                  // (uniform: written before the last barrier)
    if (tid == 0) kmx[(it + 1) & 1] = -1;                                        // (last read an iteration ago; visible after the barrier below)
    const int kend = min(KS, ((kl + 32) >> 5));                                  // k-steps that hold a non-zero y
    const int kend4 = (kend + 3) & ~3;
    if (kend4 > 0) {
      pu32x4 hb[4][NT];                                                          // ring of four k-steps of H fragments (loads three steps ahead)
      asm_sfor<0, 3>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) hb[b][jt] = hload(b, jt);
      });
      // (straight-line code over the KS k-steps, left in whole groups of four: every buffer index is a constant and the wait counts
      // of the loads in flight are exact)
      asm_sfor<0, KS / 4>([&](auto gc) __attribute__((always_inline)) {
        constexpr int ks0 = 4 * decltype(gc)::value;
        if (ks0 < kend4) {
          asm_sfor<0, 4>([&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value, ks = ks0 + b;
            constexpr int kn = ks + 3 < KS ? ks + 3 : KS - 1;                    // (a clamped reload is never multiplied)
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) hb[(b + 3) & 3][jt] = hload(kn, jt);
            pu32x4 yf[4];
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) yf[pt] = *reinterpret_cast<const pu32x4*>(Y + (16 * pt + li) * LDY + 32 * ks + 8 * lq);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
#pragma unroll
              for (int pt = 0; pt < 4; ++pt)
                acc[jt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf16x8, hb[b][jt]), __builtin_bit_cast(pbf16x8, yf[pt]),
                                                                      acc[jt][pt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                                   // (one k-step per scheduling region: bounded live ranges)
          });
        }
      });
    }
    __syncthreads();                                                             // every wave is done reading Y
    // ---- prox step, momentum, Y rewritten in place (each lane owns its elements)
    const float beta = a.beta[it];
    const bool last = it + 1 == a.iters;
    int kmax = -1;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const int c0 = 16 * (w * NT + jt) + 4 * lq;
      const pf32x4 t4 = *reinterpret_cast<const pf32x4*>(tl + c0);
      bool nz = false;
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        unsigned short* yp = Y + (16 * pt + li) * LDY + c0;
        const pu32x2 yw = *reinterpret_cast<const pu32x2*>(yp);
        // (nu is a multiple of 4 and so is the lane's first column: its four columns are four consecutive inputs of one stage)
        const pf32x4 lb4 = *reinterpret_cast<const pf32x4*>(lbs + (16 * pt + li) * ldb + kc0[jt]);
        const pf32x4 ub4 = *reinterpret_cast<const pf32x4*>(ubs + (16 * pt + li) * ldb + kc0[jt]);
        unsigned yb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned sv = s[jt][pt][r];
          const float xu = (float)__builtin_bit_cast(_Float16, (unsigned short)(sv >> 16));
          const float mu = __uint_as_float(sv << 16);
          const float y = __uint_as_float((r & 1) ? (yw[r >> 1] & 0xffff0000u) : (yw[r >> 1] << 16));
          const float x = xu - acc[jt][pt][r];
          const float v = fmaf(t4[r], x, y);
          // prox of t sigma_box:  v - t clip(v / t, lb, ub) = v - clip(v, t lb, t ub)  -- in THIS form a v inside the scaled box gives an
          // exact zero (v - v); through v / t the rounding leaves +-1e-8 of either sign, and the sign is what names the set
          const float mun = v - fminf(fmaxf(v, t4[r] * lb4[r]), t4[r] * ub4[r]);
          const float yn = fmaf(beta, mun - mu, mun);
          yb[r] = pred_bf16(yn);
          s[jt][pt][r] = (sv & 0xffff0000u) | pred_bf16(mun);
          nz |= yb[r] != 0u && yb[r] != 0x8000u;
          if (last) {                                                           // the predicted set: the sign of the multiplier
            // (mun itself, not its bf16 image; exact zeros come out of the projection)
            acc[jt][pt][r] = mun;
          }
        }
        *reinterpret_cast<pu32x2*>(yp) = pu32x2{yb[0] | (yb[1] << 16), yb[2] | (yb[3] << 16)};
      }
      if (__any(nz)) kmax = 16 * (w * NT + jt) + 15;
      __builtin_amdgcn_sched_barrier(0);                                         // (one column tile per scheduling region)
    }
    if (lane == 0 && kmax >= 0) atomicMax(&kmx[(it + 1) & 1], kmax);
    if (last) {
      // ---- bound states of the window: 1 upper (mu > 0), 2 lower (mu < 0), 0 free
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const int c0 = 16 * (w * NT + jt) + 4 * lq;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          const int p = p0 + 16 * pt + li;
          unsigned wd = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float m = acc[jt][pt][r];
            wd |= (m > 0.f ? 1u : (m < 0.f ? 2u : 0u)) << (8 * r);
          }
          if (p < d.nseg && st_words && c0 + 3 < d.n) *reinterpret_cast<unsigned*>(d.st + (size_t)p * d.n + c0) = wd;
          else if (p < d.nseg) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (c0 + r < d.n) d.st[(size_t)p * d.n + c0 + r] = (unsigned char)((wd >> (8 * r)) & 0xff);
          }
        }
      }
    }
    __syncthreads();                                                             // Y and kmx of the next iteration are complete
  }
}

}  // namespace nnmpc
