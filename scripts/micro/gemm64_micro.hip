// Stand-alone A/B of the fp64 NT GEMM tiles: gemm_nt_f64_128_k (rounds 1-2) against gemm_nt_f64_t128_k (gemm64.h).
//   gemm64_micro [M] [N] [K]      (multiples of 128 / 128 / 16)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include "gemm_kernels.h"
#include "gemm64.h"
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
