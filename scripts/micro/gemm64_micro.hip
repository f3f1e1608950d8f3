// Stand-alone A/B of the fp64 NT GEMM tiles: gemm_nt_f64_128_k (rounds 1-2) against gemm_nt_f64_t128_k (gemm64.h).
//   gemm64_micro [M] [N] [K]      (multiples of 128 / 128 / 16)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include "gemm_kernels.h"
#include "gemm64.h"
namespace nnmpc {
// ---- round 2's tile, kept here as the reference of the A/B (it left the library in round 3)
// Same contract, 128 x 128 tile per workgroup (2 x 2 waves of 64 x 64 = 4 x 4 MFMA tiles): half the
// L2 -> LDS traffic per flop of the 64 x 64 kernel, which is what that one is bound by.  Needs M and the
// grid's N multiples of 128; 74 KB of LDS (two workgroups per CU).
static __global__ __launch_bounds__(256, 2) void gemm_nt_f64_128_k(double* __restrict__ C, size_t ldc,
                                                                  const double* __restrict__ A, size_t lda,
                                                                  const double* __restrict__ B, size_t ldb,
                                                                  int K, const int* __restrict__ rowphase,
                                                                  int want, const int* __restrict__ kdyn = nullptr,
                                                                  const int* __restrict__ mdyn = nullptr, int kper = 0) {
  constexpr int LD = 18, TS = 128 * LD;
  extern __shared__ __attribute__((aligned(16))) double sm128[];   // [2][A 128 x LD | B 128 x LD]
  if (kdyn) {
    int kl = kdyn[kper * blockIdx.y];
    for (int i = 1; i < kper; ++i) kl = max(kl, kdyn[kper * blockIdx.y + i]);
    K = min(K, ((kl + 16) / 16) * 16);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  if (mdyn && m0 >= *mdyn) return;
  if (rowphase) {
    const int need = tid < 128 ? (rowphase[m0 + tid] == want) : 0;
    if (!__syncthreads_or(need)) return;
  }
  f64x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  // staging: 128 rows x 16 doubles = 1024 double2 per operand -> 4 per thread
  const int lrow0 = tid >> 3, lc = (tid & 7) * 2;           // rows lrow0 + 32 h
  const double* Ag = A + (size_t)m0 * lda;
  const double* Bg = B + (size_t)n0 * ldb;
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
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wr * 64 + i * 16 + (lane >> 4) + 4 * r;
        const int col = n0 + wc * 64 + j * 16 + (lane & 15);
        C[(size_t)row * ldc + col] = acc[i][j][r];
      }
}
constexpr int GEMM64_128_LDS = 2 * 2 * 128 * 18 * 8;

}  // namespace nnmpc
using namespace nnmpc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 99968, N = argc > 2 ? atoi(argv[2]) : 4480, K = argc > 3 ? atoi(argv[3]) : 288;
  printf("M %d N %d K %d\n", M, N, K);
  std::vector<double> hA((size_t)M * K), hB((size_t)N * K);
  srand(1);
  for (auto& v : hA) v = (rand() / (double)RAND_MAX) * 2 - 1;
  for (auto& v : hB) v = (rand() / (double)RAND_MAX) * 2 - 1;
  double *A, *B, *C0, *C1;
  CK(hipMalloc(&A, hA.size() * 8)); CK(hipMalloc(&B, hB.size() * 8));
  CK(hipMalloc(&C0, (size_t)M * N * 8)); CK(hipMalloc(&C1, (size_t)M * N * 8));
  CK(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hB.data(), hB.size() * 8, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)gemm_nt_f64_128_k, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM64_128_LDS));
  CK(hipFuncSetAttribute((const void*)gemm_nt_f64_t128_k, hipFuncAttributeMaxDynamicSharedMemorySize, G64_LDS));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ntm = M / 128, ntn = N / 128;
  auto run_old = [&] { hipLaunchKernelGGL(gemm_nt_f64_128_k, dim3(ntn, ntm), dim3(256), GEMM64_128_LDS, 0, C0, (size_t)N, A, (size_t)K, B, (size_t)K, K, (const int*)nullptr, 0, (const int*)nullptr, (const int*)nullptr, 0); };
  auto run_new = [&] { hipLaunchKernelGGL(gemm_nt_f64_t128_k, dim3(g64_grid(ntm, ntn)), dim3(256), G64_LDS, 0, C1, (size_t)N, A, (size_t)K, B, (size_t)K, K, ntm, ntn, (const int*)nullptr, 0, (const int*)nullptr, 0, (const int*)nullptr, (const int*)nullptr); };
  run_old(); run_new();
  CK(hipDeviceSynchronize());
  // compare on a sample of rows
  {
    std::vector<double> r0(N), r1(N);
    double md = 0, mref = 0;
    for (int t = 0; t < 64; ++t) {
      const int row = (int)(((long long)t * 1566083941ll) % M);
      CK(hipMemcpy(r0.data(), C0 + (size_t)row * N, N * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r1.data(), C1 + (size_t)row * N, N * 8, hipMemcpyDeviceToHost));
      for (int c = 0; c < N; ++c) {
        md = fmax(md, fabs(r0[c] - r1[c]));
        if (t < 4 && c % 97 == 0) { double s = 0; for (int k = 0; k < K; ++k) s += hA[(size_t)row * K + k] * hB[(size_t)c * K + k]; mref = fmax(mref, fabs(s - r1[c])); }
      }
    }
    printf("max |old - new| %.3e   max |host - new| %.3e\n", md, mref);
  }
  for (int rep = 0; rep < 3; ++rep)
    for (int v = 0; v < 2; ++v) {
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) { if (v) run_new(); else run_old(); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      printf("%s  %.3f ms  %.1f TFLOP/s\n", v ? "new t128" : "old 128 ", ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    }
  // row map: reversed order of the first 300 rows
  {
    const int nr = 300;
    std::vector<int> map(384, -1);
    for (int i = 0; i < nr; ++i) map[i] = (M - 1) - 7 * i;
    int *dmap, *dcnt; CK(hipMalloc(&dmap, 384 * 4)); CK(hipMalloc(&dcnt, 4));
    CK(hipMemcpy(dmap, map.data(), 384 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcnt, &nr, 4, hipMemcpyHostToDevice));
    CK(hipMemset(C1, 0, (size_t)M * N * 8));
    hipLaunchKernelGGL(gemm_nt_f64_t128_k, dim3(g64_grid(3, ntn)), dim3(256), G64_LDS, 0, C1, (size_t)N, A, (size_t)K, B, (size_t)K, K, 3, ntn, (const int*)nullptr, 0, (const int*)nullptr, 0, dcnt, dmap);
    CK(hipDeviceSynchronize());
    std::vector<double> r0(N), r1(N);
    double md = 0, mz = 0;
    for (int i : {0, 1, 127, 128, 299}) {
      const int row = map[i];
      CK(hipMemcpy(r0.data(), C0 + (size_t)row * N, N * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(r1.data(), C1 + (size_t)row * N, N * 8, hipMemcpyDeviceToHost));
      for (int c = 0; c < N; ++c) md = fmax(md, fabs(r0[c] - r1[c]));
    }
    CK(hipMemcpy(r1.data(), C1 + (size_t)(M - 2) * N, N * 8, hipMemcpyDeviceToHost));   // not in the map: untouched
    for (int c = 0; c < N; ++c) mz = fmax(mz, fabs(r1[c]));
    printf("row map: max |old - new| on mapped rows %.3e, unmapped row max %.3e\n", md, mz);
  }
  return 0;
}
