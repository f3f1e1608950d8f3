// Micro-benchmark of the multiplier-system kernels (qp_asm.h) on synthetic sets of a fixed size.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I industrial_nnmpc_2021_amd/csrc scripts/micro/lambda_micro.hip -o gpurun_out/lambda_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "qp_asm.h"
using namespace nnmpc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// variant 3: ONE size class of the f32 register kernel at a chosen occupancy / LDS split (-DPROBE_MB=8 -DPROBE_OCC=3 -DASM_NL_OF_8=1)
#ifdef PROBE_MB
#ifndef PROBE_OCC
#define PROBE_OCC 2
#endif
extern "C" __global__ __launch_bounds__(256, PROBE_OCC) void probe_k(AsmDev d) {
  asm_lambda_reg<float, PROBE_MB, 4>(d, ASM_NBIN + PROBE_MB - 4, blockIdx.x);
}
#endif

#ifndef ASM_NO_WG_KERNELS
// variants 10 / 11: the workgroup kernels of qp_wg.h with EIGHT waves per problem (fp64 / f32), 12 .. 16 blocks
template <class T> __device__ __forceinline__ void wg8_any(const AsmDev& d, int p) {
  const int m = __builtin_amdgcn_readfirstlane(d.mg[p]);
  switch (max((m + 15) >> 4, 12)) {
    case 16: asm_lambda_wg<T, 16, 8>(d, p, m); break;
    case 15: asm_lambda_wg<T, 15, 8>(d, p, m); break;
    case 14: asm_lambda_wg<T, 14, 8>(d, p, m); break;
    case 13: asm_lambda_wg<T, 13, 8>(d, p, m); break;
    default: asm_lambda_wg<T, 12, 8>(d, p, m); break;
  }
}
__global__ __launch_bounds__(512, 1) void wg64_8_k(AsmDev d) { if ((int)blockIdx.x < d.counters[ASM_CNT_BIG64]) wg8_any<double>(d, d.binlist[(size_t)(ASM_NLIST + 1) * d.nseg + blockIdx.x]); }
__global__ __launch_bounds__(512, 2) void wg32_8_k(AsmDev d) { if ((int)blockIdx.x < d.counters[ASM_CNT_BIG32]) wg8_any<float>(d, d.binlist[(size_t)ASM_NLIST * d.nseg + blockIdx.x]); }
#endif

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 112;
  const int nseg = argc > 2 ? atoi(argv[2]) : 14336;
  const int variant = argc > 3 ? atoi(argv[3]) : 0;      // 0 tile kernel, 1 register kernel (fp64), 2 register kernel (f32)
  const int n = 512, np = 512, nu = 32, max_active = 768;
  const int win0 = argc > 4 ? atoi(argv[4]) : 416;        // the active indices are drawn from [0, win): win = m makes them contiguous
  const int win = win0 > 0 ? win0 : 512;                   // win < 0: -win inputs saturated over the first steps (indices step * nu + input: the solver's pattern)
  const int gperm = argc > 5 ? atoi(argv[5]) : 0;          // 1: the list in input-major order (same set, another pivot order)
  std::mt19937_64 rng(1);
  std::normal_distribution<double> g(0.0, 1.0);
  std::vector<double> G((size_t)n * n), H((size_t)n * n);
  for (auto& v : G) v = g(rng) / sqrt((double)n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = i == j ? 0.5 : 0.0;
      for (int k = 0; k < n; ++k) s += G[(size_t)i * n + k] * G[(size_t)j * n + k];
      H[(size_t)i * n + j] = H[(size_t)j * n + i] = s;
    }
  std::vector<int> idx((size_t)nseg * max_active, 0), mg(nseg, m), list(nseg);
  std::vector<unsigned char> st((size_t)nseg * n, 0);
  std::vector<double> xunc((size_t)nseg * np), lb(nseg * nu, -1.0), ub(nseg * nu, 0.25);
  for (auto& v : xunc) v = g(rng);
  std::vector<int> perm(win);
  for (int p = 0; p < nseg; ++p) {
    for (int i = 0; i < win; ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    if (win0 < 0) {
      const int k = -win0;
      std::vector<int> inp(nu);
      for (int i = 0; i < nu; ++i) inp[i] = i;
      std::shuffle(inp.begin(), inp.end(), rng);           // the k saturated inputs
      for (int i = 0; i < m; ++i) perm[i] = (i / k) * nu + inp[i % k];
    }
    std::sort(perm.begin(), perm.begin() + m);
    if (gperm) std::stable_sort(perm.begin(), perm.begin() + m, [&](int a, int b) { return a % nu < b % nu; });
    for (int i = 0; i < m; ++i) { idx[(size_t)p * max_active + i] = perm[i]; st[(size_t)p * n + perm[i]] = 1 + (perm[i] & 1); }
    list[p] = p;
  }
  AsmDev d{};
  d.n = n; d.np = np; d.nu = nu; d.nseg = nseg; d.max_active = max_active;
  double *dH, *dlb, *dub, *dxu, *dlam; unsigned char* dst; int *dstate, *dcnt, *dbin, *didx, *dmg;
  CK(hipMalloc(&dH, H.size() * 8)); CK(hipMemcpy(dH, H.data(), H.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dlb, lb.size() * 8)); CK(hipMemcpy(dlb, lb.data(), lb.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dub, ub.size() * 8)); CK(hipMemcpy(dub, ub.data(), ub.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dxu, xunc.size() * 8)); CK(hipMemcpy(dxu, xunc.data(), xunc.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dlam, xunc.size() * 8)); CK(hipMemset(dlam, 0, xunc.size() * 8));
  CK(hipMalloc(&dst, st.size())); CK(hipMemcpy(dst, st.data(), st.size(), hipMemcpyHostToDevice));
  CK(hipMalloc(&dstate, nseg * 4)); CK(hipMemset(dstate, 0, nseg * 4));
  int bin = max((m + 15) / 16, 4) - 4; if (bin >= ASM_NBIN) bin = ASM_NBIN - 1;
  int cnt[ASM_NCNT] = {0}; cnt[4 + bin] = nseg; cnt[ASM_CNT_F32 + bin] = nseg; cnt[ASM_CNT_BIG32] = nseg;
  if (variant == 5) { cnt[ASM_CNT_F32 + 6] = cnt[ASM_CNT_F32 + 7] = 0; }            // four-wave register kernels (qp_wg.h): one list each
  if (variant == 13) cnt[ASM_CNT_BIG64R] = nseg;             // ... and its refined (fp64-residual) instance
  if (variant == 12) cnt[ASM_CNT_BIG32B] = nseg;             // eight-wave f32 kernel of the solver: 257 .. 384 bounds
  if (variant == 6 || variant == 10 || variant == 14) { cnt[4 + 6] = cnt[4 + 7] = 0; cnt[ASM_CNT_BIG64] = nseg; }
  CK(hipMalloc(&dcnt, sizeof cnt)); CK(hipMemcpy(dcnt, cnt, sizeof cnt, hipMemcpyHostToDevice));
  CK(hipMalloc(&dbin, (size_t)(ASM_NLIST + 4) * nseg * 4));
  for (int b = 0; b <= ASM_NLIST + 3; ++b) CK(hipMemcpy(dbin + (size_t)b * nseg, list.data(), nseg * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&didx, idx.size() * 4)); CK(hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dmg, nseg * 4)); CK(hipMemcpy(dmg, mg.data(), nseg * 4, hipMemcpyHostToDevice));
  d.H = dH; d.lb = dlb; d.ub = dub; d.xunc = dxu; d.lam = dlam; d.st = dst; d.state = dstate; d.counters = dcnt;
  d.binlist = dbin; d.idxg = didx; d.mg = dmg;
  unsigned char *dprec, *dredo; CK(hipMalloc(&dprec, nseg)); CK(hipMemset(dprec, 0, nseg)); CK(hipMalloc(&dredo, nseg)); CK(hipMemset(dredo, 0, nseg));
  d.prec = dprec; d.redo = dredo;
  {
    std::vector<float> H32(H.size());
    for (size_t i = 0; i < H.size(); ++i) H32[i] = (float)H[i];
    float* dH32; CK(hipMalloc(&dH32, H32.size() * 4)); CK(hipMemcpy(dH32, H32.data(), H32.size() * 4, hipMemcpyHostToDevice));
    d.H32 = dH32;
  }
  float* dlam32; CK(hipMalloc(&dlam32, xunc.size() * 4)); CK(hipMemset(dlam32, 0, xunc.size() * 4)); d.lam32 = dlam32;
  unsigned char* drowk; CK(hipMalloc(&drowk, nseg)); CK(hipMemset(drowk, variant == 20 ? 0 : 1, nseg)); d.rowk = drowk;   // variant 20: the f32 register kernels with rows of LAM -- every solve gets its fp64 correction (build with -DASM_REFINE_ALWAYS: the synthetic sets have random multiplier signs)
  int* drow; CK(hipMalloc(&drow, nseg * 4)); CK(hipMemcpy(drow, list.data(), nseg * 4, hipMemcpyHostToDevice)); d.row = drow;
  const int mbc = asm_bin_cap(bin) / 16;
  const int lds_tile = (asm_bin_cap(bin) + ASM_TS + mbc * (mbc + 1) / 2 * ASM_TS) * 8;
  CK(hipFuncSetAttribute((const void*)asm_lambda_tile_k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (ASM_MLDS + ASM_TS + 66 * ASM_TS) * 8));
  CK(hipFuncSetAttribute((const void*)asm_lambda_reg_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_REG_LDS));
  CK(hipFuncSetAttribute((const void*)asm_lambda_reg2_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_REG2_LDS));
  CK(hipFuncSetAttribute((const void*)asm_lambda_reg32_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_REG32_LDS));
  CK(hipFuncSetAttribute((const void*)asm_lambda_reg32b_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_REG32B_LDS));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, 0));
#ifndef ASM_NO_WG_KERNELS
    if (variant == 7) hipLaunchKernelGGL(asm_lambda_wg32s_k, dim3(nseg), dim3(128), asm_wg_lds_bytes<float>(), 0, d);
    else if (variant == 8) hipLaunchKernelGGL(asm_lambda_wg64s_k, dim3(nseg), dim3(128), asm_wg_lds_bytes<double>(), 0, d);
    else if (variant == 5) hipLaunchKernelGGL(asm_lambda_wg32_k, dim3(nseg), dim3(256), asm_wg_lds_bytes<float>(), 0, d);
    else if (variant == 6) hipLaunchKernelGGL(asm_lambda_wg64_k, dim3(nseg), dim3(256), asm_wg_lds_bytes<double>(), 0, d);
    else if (variant == 14) hipLaunchKernelGGL(asm_lambda_wg64r4_k, dim3(nseg), dim3(256), (asm_wg_lds_bytes_refine<float, ASM_WG_MB>()), 0, d);
    else if (variant == 13) hipLaunchKernelGGL(asm_lambda_wg64r_k, dim3(nseg), dim3(512), (asm_wg_lds_bytes_refine<float, ASM_WG_MB8>()), 0, d);
    else if (variant == 12) hipLaunchKernelGGL(asm_lambda_wg32b_k, dim3(nseg), dim3(512), (asm_wg_lds_bytes<float, ASM_WG_MB8>()), 0, d);
    else if (variant == 10) hipLaunchKernelGGL(wg64_8_k, dim3(nseg), dim3(512), asm_wg_lds_bytes<double>(), 0, d);
    else if (variant == 11) hipLaunchKernelGGL(wg32_8_k, dim3(nseg), dim3(512), asm_wg_lds_bytes<float>(), 0, d);
    else
#endif
    if (variant == 4) {                                  // f32 LDS-tile workgroup kernel (177..256 bounds)
      static bool once4 = false;
      if (!once4) { CK(hipFuncSetAttribute((const void*)asm_lambda_tile32_k, hipFuncAttributeMaxDynamicSharedMemorySize, ASM_TILE32_LDS)); once4 = true; }
      hipLaunchKernelGGL(asm_lambda_tile32_k, dim3(std::min(nseg, 4096)), dim3(512), ASM_TILE32_LDS, 0, d);
    } else
#ifdef PROBE_MB
    if (variant == 3) {
      static bool once = false;
      const int lds = (4 * asm_rw<float>(PROBE_MB) + ASM_TS) * 4;
      if (!once) { CK(hipFuncSetAttribute((const void*)probe_k, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); once = true; }
      hipLaunchKernelGGL(probe_k, dim3((nseg + 3) / 4), dim3(256), lds, 0, d);
    } else
#endif
    if (variant == 0) hipLaunchKernelGGL((asm_lambda_tile_k<0>), dim3(nseg), dim3(256), lds_tile, 0, d, bin);
    else if ((variant == 2 || variant == 20) && bin < ASM_NREG) hipLaunchKernelGGL(asm_lambda_reg32_k, dim3((nseg + 3) / 4 + ASM_NREG), dim3(256), ASM_REG32_LDS, 0, d);
    else if (variant == 2 || variant == 20) hipLaunchKernelGGL(asm_lambda_reg32b_k, dim3((nseg + 3) / 4 + 2), dim3(256), ASM_REG32B_LDS, 0, d);
    else {
      if (bin < ASM_NREG) hipLaunchKernelGGL(asm_lambda_reg_k, dim3((nseg + 3) / 4 + ASM_NREG), dim3(256), ASM_REG_LDS, 0, d);
      else hipLaunchKernelGGL(asm_lambda_reg2_k, dim3((nseg + 1) / 2 + 2), dim3(128), ASM_REG2_LDS, 0, d);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) best = std::min(best, ms);
  }
  CK(hipGetLastError());
  // check two problems: residual of S lam = r
  std::vector<double> lam((size_t)2 * np);
  std::vector<int> state(nseg);
  CK(hipMemcpy(state.data(), dstate, nseg * 4, hipMemcpyDeviceToHost));
  int nfb = 0; for (int s : state) nfb += s != 0;
  if (variant == 20) {
    std::vector<unsigned char> rd(nseg); CK(hipMemcpy(rd.data(), dredo, nseg, hipMemcpyDeviceToHost));
    int n2 = 0; for (auto v : rd) n2 += v == 2;
    printf("corrected in fp64: %d of %d\n", n2, nseg);
  }
  double worst = 0.0;
  for (int pp = 0; pp < 2; ++pp) {
    const int p = pp == 0 ? 0 : nseg - 1;
    if (variant == 2 || variant == 3 || variant == 4 || variant == 5 || variant == 7 || variant == 11 || variant == 12) {
      std::vector<float> l32(np);
      CK(hipMemcpy(l32.data(), dlam32 + (size_t)p * np, np * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < np; ++i) lam[i] = l32[i];
    } else
    CK(hipMemcpy(lam.data(), dlam + (size_t)p * np, np * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < m; ++i) {
      const int a = idx[(size_t)p * max_active + i];
      double r = xunc[(size_t)p * np + a] - (st[(size_t)p * n + a] == 1 ? 0.25 : -1.0);
      for (int j = 0; j < m; ++j) { const int b2 = idx[(size_t)p * max_active + j]; r -= H[(size_t)a * n + b2] * lam[b2]; }
      worst = std::max(worst, fabs(r));
    }
  }
#ifdef ASM_STAMPS
  {
    unsigned long long st_[64];
    CK(hipMemcpyFromSymbol(st_, HIP_SYMBOL(asm_stamp_buf), sizeof st_));
    printf("stamps (cycles since wave start): rhs in LDS %llu, rhs+gather issued %llu, first tile ready %llu, diag0 %llu\n", st_[60] - st_[0], st_[1] - st_[0], st_[2] - st_[0], st_[3] - st_[2]);
    for (int K = 0; K < 9 && st_[4 + 3 * K]; ++K)
      printf("  column %d: start %llu  trsm-issued +%llu  trail-issued +%llu  (next column at +%llu)\n", K, st_[4 + 3 * K] - st_[0], st_[5 + 3 * K] - st_[4 + 3 * K],
             st_[6 + 3 * K] > st_[5 + 3 * K] ? st_[6 + 3 * K] - st_[5 + 3 * K] : 0ull, st_[4 + 3 * (K + 1)] > st_[4 + 3 * K] ? st_[4 + 3 * (K + 1)] - st_[4 + 3 * K] : 0ull);
    printf("  factor done %llu, backward substitution %llu\n", st_[40] - st_[0], st_[41] - st_[40]);
    if (st_[52]) printf("  tile32: rhs+gather %llu, first diagonal step %llu, factorisation %llu, substitutions %llu\n", st_[49] - st_[48], st_[50] - st_[49], st_[51] - st_[50], st_[52] - st_[51]);
  }
#endif
  printf("variant %d m %d nseg %d: %.3f ms  (%.2f problems/us)  max residual %.2e  fallback %d\n", variant, m, nseg, best, nseg / (best * 1e3), worst, nfb);
  return 0;
}
