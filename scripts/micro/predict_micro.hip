// Micro-benchmark and self-check of the first-set predictor (qp_predict.h) on a synthetic window:
//   <bin> <problems> <iterations> [nu]
// Prints the kernel time, the bf16 MFMA rate it implies and the fraction of bound states that differ from a host emulation of
// the same iteration (same roundings of the operands; sums in another order -- a handful of near-zero multipliers may differ).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
#ifndef PRED_MICRO_NT
#define PRED_MICRO_NT 4
#define PRED_MICRO_PT 4
#endif
#define ASM_NO_WG_KERNELS
#include "qp_predict.h"
using namespace nnmpc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static unsigned short bfh(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bff(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char** argv) {
  const int nseg = argc > 1 ? atoi(argv[1]) : 16384, iters = argc > 2 ? atoi(argv[2]) : 24, nu = argc > 3 ? atoi(argv[3]) : 32;
  constexpr int NTm = PRED_MICRO_NT, PTm = PRED_MICRO_PT;
  const int W = PredCfg<NTm>::W, n = 2048, np = 2048;
  std::mt19937_64 rng(3);
  std::normal_distribution<double> g(0.0, 1.0);
  std::vector<double> G((size_t)W * W), H((size_t)W * W);
  for (auto& v : G) v = g(rng) / sqrt((double)W);
  for (int i = 0; i < W; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = i == j ? 0.3 : 0.0;
      for (int k = 0; k < W; ++k) s += G[(size_t)i * W + k] * G[(size_t)j * W + k];
      H[(size_t)i * W + j] = H[(size_t)j * W + i] = s;
    }
  // L by power iteration on D^-1/2 H D^-1/2
  std::vector<double> v(W, 1.0), u(W);
  double lam = 0;
  for (int it = 0; it < 300; ++it) {
    double nv = 0; for (double x : v) nv += x * x; nv = sqrt(nv);
    for (auto& x : v) x /= nv;
    for (int i = 0; i < W; ++i) { double s = 0; for (int j = 0; j < W; ++j) s += H[(size_t)i * W + j] * v[j] / sqrt(H[(size_t)j * W + j]); u[i] = s / sqrt(H[(size_t)i * W + i]); }
    lam = 0; for (int i = 0; i < W; ++i) lam += u[i] * v[i];
    v = u;
  }
  const double L = 1.05 * lam;
  std::vector<unsigned short> hf((size_t)W * W), hb((size_t)W * W);
  for (int j = 0; j < W; ++j) for (int k = 0; k < W; ++k) hb[(size_t)j * W + k] = bfh((float)(H[(size_t)j * W + k] * (1.0 / (L * H[(size_t)k * W + k]))));   // H' = H diag(t)
  for (int jt = 0; jt < W / 16; ++jt)
    for (int ks = 0; ks < PredCfg<NTm>::KS; ++ks)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e)
          hf[pred_frag_index<NTm>(jt, ks, lane) * 8 + e] = hb[(size_t)(16 * jt + (lane & 15)) * W + 32 * ks + 8 * (lane >> 4) + e];
  std::vector<double> xunc((size_t)nseg * np), lb((size_t)nseg * nu), ub((size_t)nseg * nu);
  for (auto& x : xunc) x = 0.8 * g(rng);
  for (int p = 0; p < nseg; ++p) for (int k = 0; k < nu; ++k) { lb[(size_t)p * nu + k] = -1.0 - 0.3 * fabs(g(rng)); ub[(size_t)p * nu + k] = 1.0 + 0.3 * fabs(g(rng)); }
  AsmDev d{};
  d.n = n; d.np = np; d.nu = nu; d.nseg = nseg;
  double *dx, *dlb, *dub; unsigned char* dst; pu32x4* dHf;
  CK(hipMalloc(&dx, xunc.size() * 8)); CK(hipMemcpy(dx, xunc.data(), xunc.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dlb, lb.size() * 8)); CK(hipMemcpy(dlb, lb.data(), lb.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dub, ub.size() * 8)); CK(hipMemcpy(dub, ub.data(), ub.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dst, (size_t)nseg * n)); CK(hipMemset(dst, 0, (size_t)nseg * n));
  CK(hipMalloc(&dHf, hf.size() * 2)); CK(hipMemcpy(dHf, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
  d.xunc = dx; d.lb = dlb; d.ub = dub; d.st = dst;
  PredArgs pa; pa.Hf = dHf; pa.iters = iters; pa.adaptive = 0; pa.fac10 = 3; pa.itsum = nullptr;
  { double t = 1.0; for (int k = 0; k < PRED_MAXIT; ++k) { const double tn = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t * t)); pa.beta[k] = k < iters ? (float)((t - 1.0) / tn) : 0.f; t = tn; } }
  CK(hipFuncSetAttribute((const void*)asm_predict_k<NTm, PTm>, hipFuncAttributeMaxDynamicSharedMemorySize, pred_lds_bytes<NTm, PTm>(nu)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((asm_predict_k<NTm, PTm>), dim3((nseg + 16 * PTm - 1) / (16 * PTm)), dim3(64 * PRED_NW), (pred_lds_bytes<NTm, PTm>(nu)), 0, d, pa);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = std::min(best, ms);
  }
  CK(hipGetLastError());
  std::vector<unsigned char> st((size_t)nseg * n);
  CK(hipMemcpy(st.data(), dst, st.size(), hipMemcpyDeviceToHost));
  // host emulation on a few problems
  long diff = 0, tot = 0, act = 0;
  // (whole workgroups: the momentum restart is decided over the 16 PT problems of a workgroup)
  const int MRm = 16 * PTm, nwg = (nseg + MRm - 1) / MRm, ncheck = std::min(nwg, 3) * MRm;
  for (int cw = 0; cw < std::min(nwg, 3); ++cw) {
    const int g = cw == 0 ? 0 : (cw == 1 ? nwg / 2 : nwg - 1);
    std::vector<float> xu((size_t)MRm * W), mu((size_t)MRm * W, 0.f), y((size_t)MRm * W, 0.f), x(W);
    for (int q = 0; q < MRm; ++q) {
      const int p = std::min(g * MRm + q, nseg - 1);
      for (int j = 0; j < W; ++j) { float f = (float)xunc[(size_t)p * np + j]; _Float16 h = (_Float16)f; xu[(size_t)q * W + j] = (float)h; }
    }
    int kb = 0; double rs = 0.0;
    for (int it = 0; it < iters; ++it) {
      if (it > 0) kb = rs > 0.0 ? 0 : kb + 1;
      rs = 0.0;
      for (int q = 0; q < MRm; ++q) {
        const int p = std::min(g * MRm + q, nseg - 1);
        float* yq = &y[(size_t)q * W]; float* mq = &mu[(size_t)q * W];
        for (int j = 0; j < W; ++j) { double s = 0; for (int k = 0; k < W; ++k) s += (double)bff(hb[(size_t)j * W + k]) * (double)yq[k]; x[j] = xu[(size_t)q * W + j] - (float)s; }
        for (int j = 0; j < W; ++j) {
          const float wv = yq[j] + x[j];
          const float lo = (float)lb[(size_t)p * nu + j % nu], hi = (float)ub[(size_t)p * nu + j % nu];
          const float mun = wv - fminf(fmaxf(wv, lo), hi);
          const float dn = mun - mq[j];
          const float yn = fmaf(pa.beta[kb], dn, mun);
          rs += (double)((yq[j] - mun) * dn);
          x[j] = bff(bfh(yn));                                   // (y of the next iteration: the products above are done)
          if (it + 1 == iters && g * MRm + q < nseg) { const int sref = mun > 0.f ? 1 : (mun < 0.f ? 2 : 0); diff += sref != st[(size_t)p * n + j]; act += sref != 0; ++tot; }
          mq[j] = bff(bfh(mun));
        }
        for (int j = 0; j < W; ++j) yq[j] = x[j];
      }
    }
  }
  const double flops = 2.0 * W * W * (double)(iters - 1) * (double)nseg;
  printf("predict: %d problems, %d iterations, nu %d: %.3f ms (%.1f us per iteration and 64 problems per CU-slot; dense-count %.0f TFLOP/s bf16); "
         "emulation: %ld of %ld bound states differ (%.4f %%), %.1f active per problem\n", nseg, iters, nu, best,
         1e3 * best / iters / std::max(1.0, ceil(nseg / (16.0 * PTm) / 256.0)), flops / (best * 1e-3) / 1e12, diff, tot, 100.0 * diff / std::max(1L, tot), (double)act / ncheck);
  return 0;
}
