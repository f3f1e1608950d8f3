// Structured NN controller forward for gfx950:
//   u = clip( us + MLP([x/xs_scale, (uprev), xs/xs_scale, us])
//                - MLP([xs/xs_scale, (us),   xs/xs_scale, us]) )
// Reference: RegulatorLayerWithUprev.call / RegulatorLayerWithoutUprev.call
// (lib/LinearMPCLayers.py:40-61, :91-112) and its numpy twin
// NeuralNetworkController._get_control_input (lib/controller_evaluation.py:863-892).
//
// Both passes are stacked into one 2B-row activation matrix so every layer is a
// single MFMA GEMM (C = act(A W + b), weights stored transposed [out][in] so the
// kernel is the same NT tile GEMM the QP path uses); bias + ReLU are fused in
// the GEMM epilogue, concat/scale in the assemble kernel and us + (o1 - o2) +
// clip in the combine kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include <stdlib.h>
#include "../../include/nnmpc.h"
#include "gemm_kernels.h"
#include "tile_gemm_bf16.h"
#include "common.h"

using namespace nnmpc;

namespace {

// rows [0, Bp): pass 1 inputs [x^, (uprev), xs^, us], rows [Bp, 2Bp): pass 2 inputs [xs^, (us), xs^, us] (x^ = x / xscale);
// columns padded to ldk.  Workgroups walk the SAMPLES: every scaled xs and every us value is computed once and written
// to all of its places in both rows, segment by segment (no per-element source selection; consecutive lanes read
// consecutive doubles).  Samples b >= B (batch padding) and the pad columns are zero.
// (nn_assemble_split_k below: the same rows as the two bf16 planes [hi | lo] of the split-bf16 path)
template <class T>
__global__ __launch_bounds__(256) void nn_assemble_k(T* __restrict__ in, int ldk, int Bp, int B, int nx, int nu,
                              int with_uprev, const double* __restrict__ x,
                              const double* __restrict__ uprev, const double* __restrict__ xs,
                              const double* __restrict__ us, const float* __restrict__ inv_scale) {
  const int o2 = nx + (with_uprev ? nu : 0);               // first column of the xs block
  const int din = o2 + nx + nu;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int b = blockIdx.x; b < Bp; b += gridDim.x) {
    T* d1 = in + (size_t)b * ldk;
    T* d2 = in + (size_t)(Bp + b) * ldk;
    if (b >= B) {
      for (int k = tid; k < ldk; k += nt) { d1[k] = (T)0.f; d2[k] = (T)0.f; }
      continue;
    }
    const double* xa = x + (size_t)b * nx;
    const double* xb = xs + (size_t)b * nx;
    const double* ub = us + (size_t)b * nu;
    for (int k = tid; k < nx; k += nt) {
      const float sc = inv_scale[k];
      const T xv = (T)((float)xa[k] * sc), sv = (T)((float)xb[k] * sc);
      d1[k] = xv; d1[o2 + k] = sv;
      d2[k] = sv; d2[o2 + k] = sv;
    }
    for (int k = tid; k < nu; k += nt) {
      const T uv = (T)(float)ub[k];
      d1[o2 + nx + k] = uv; d2[o2 + nx + k] = uv;
      if (with_uprev) { d1[nx + k] = (T)(float)uprev[(size_t)b * nu + k]; d2[nx + k] = uv; }
    }
    for (int k = din + tid; k < ldk; k += nt) { d1[k] = (T)0.f; d2[k] = (T)0.f; }
  }
}

// Split-bf16 input rows: [hi | lo] planes of ldk columns each (rows of 2 ldk), value = hi + lo to ~2^-17 relative.
__global__ __launch_bounds__(256) void nn_assemble_split_k(__bf16* __restrict__ in, int ldk, int Bp, int B, int nx, int nu,
                              int with_uprev, const double* __restrict__ x,
                              const double* __restrict__ uprev, const double* __restrict__ xs,
                              const double* __restrict__ us, const float* __restrict__ inv_scale) {
  const int o2 = nx + (with_uprev ? nu : 0);
  const int din = o2 + nx + nu;
  const int tid = threadIdx.x, nt = blockDim.x;
  auto put = [&](__bf16* row, int k, float v) {
    const __bf16 h = (__bf16)v, l = (__bf16)(v - (float)h);
    row[k] = h; row[ldk + k] = l;
  };
  for (int b = blockIdx.x; b < Bp; b += gridDim.x) {
    __bf16* d1 = in + (size_t)b * 2 * ldk;
    __bf16* d2 = in + (size_t)(Bp + b) * 2 * ldk;
    if (b >= B) {
      for (int k = tid; k < 2 * ldk; k += nt) { d1[k] = (__bf16)0.f; d2[k] = (__bf16)0.f; }
      continue;
    }
    const double* xa = x + (size_t)b * nx;
    const double* xb = xs + (size_t)b * nx;
    const double* ub = us + (size_t)b * nu;
    for (int k = tid; k < nx; k += nt) {
      const float sc = inv_scale[k];
      const float xv = (float)xa[k] * sc, sv = (float)xb[k] * sc;
      put(d1, k, xv); put(d1, o2 + k, sv);
      put(d2, k, sv); put(d2, o2 + k, sv);
    }
    for (int k = tid; k < nu; k += nt) {
      const float uv = (float)ub[k];
      put(d1, o2 + nx + k, uv); put(d2, o2 + nx + k, uv);
      if (with_uprev) { put(d1, nx + k, (float)uprev[(size_t)b * nu + k]); put(d2, nx + k, uv); }
    }
    for (int k = din + tid; k < ldk; k += nt) { put(d1, k, 0.f); put(d2, k, 0.f); }
  }
}

__global__ void nn_combine_k(double* __restrict__ u, const float* __restrict__ o, int ldo, int Bp,
                             int B, int nu, const double* __restrict__ us,
                             const double* __restrict__ ulb, const double* __restrict__ uub,
                             int clip) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * nu;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int b = (int)(i / nu), c = (int)(i % nu);
    double v = us[i] + ((double)o[(size_t)b * ldo + c] - (double)o[(size_t)(Bp + b) * ldo + c]);
    if (clip) { v = v > uub[c] ? uub[c] : v; v = v < ulb[c] ? ulb[c] : v; }
    u[i] = v;
  }
}

}  // namespace

struct nnmpc_nn {
  int device;
  int nlayers;             // number of weight matrices
  std::vector<int> dims;   // nlayers + 1
  std::vector<int> kpad;   // padded input width of layer l (multiple of 32)
  std::vector<int> npad;   // padded output width of layer l (multiple of 64)
  std::vector<float*> Wt;  // [npad][kpad] transposed weights (f32 path)
  std::vector<bf16raw*> Wt16;  // bf16 path: [n16][k16], k16 = layer input width rounded to 64 only (832 stays 832)
  std::vector<int> k16, n16, ldc16;  // ldc16 = row length of the layer's output = k16 of the next layer
  int use_bf16;            // 0 f32, 1 bf16, 2 split bf16 (activations and weights as hi + lo pairs, three products per layer)
  int split;               // use_bf16 == 2: activation rows [hi | lo], walked hi, hi, lo against weights stacked [hi ; lo ; hi] along K
  std::vector<float*> bias;  // [npad]
  int nx, nu, with_uprev, clip, max_batch;
  float* inv_scale;
  double *ulb, *uub;
  float* act[2];           // ping-pong activations [2*max_batch][maxw] (f32; the bf16 path uses them as bf16 storage, head output f32)
  int maxw;
  double *sx, *suprev, *sxs, *sus, *su;  // staging for host pointers
  hipStream_t stream;
  hipEvent_t e0, e1;
  std::vector<hipEvent_t> eg;   // (start, end of the hidden layers, end) of the GEMMs of every sub-batch of a call
  double hidden_ms; int hidden_launches;
  double gemm_ms, total_ms;
  std::vector<void*> allocs;
};

namespace {
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e_)); return NNMPC_EHIP; } } while (0)

template <class T>
int nn_alloc(nnmpc_nn* h, T** p, size_t count) {
  void* q = nullptr;
  if (hipMalloc(&q, count * sizeof(T)) != hipSuccess) { set_error("hipMalloc(%zu) failed", count * sizeof(T)); return NNMPC_ENOMEM; }
  hipMemset(q, 0, count * sizeof(T));
  h->allocs.push_back(q);
  *p = (T*)q;
  return 0;
}

template <int NB, bool RELU, bool BIAS>
void launch_layer(hipStream_t s, float* C, size_t ldc, const float* A, size_t lda, const float* Wt,
                  size_t ldb, int M, int N, int K, const float* bias) {
  dim3 grid(N / NB, M / NB);
  hipLaunchKernelGGL((gemm_nt_f32_k<NB, RELU, BIAS>), grid, dim3(256), TileCfg<NB>::LDS_FLOATS * 4, s,
                     C, ldc, A, lda, Wt, ldb, K, bias);
}
template <int NB, bool RELU, bool BIAS, int OUT>
void launch_layer16(hipStream_t s, void* C, int ldc, const bf16raw* A, size_t lda, const bf16raw* Wt,
                    size_t ldb, int M, int K, const float* bias, int nk0 = 0) {
  const int ntm = M / NB, ntn = (ldc + NB - 1) / NB;
  hipLaunchKernelGGL((gemm_nt_bf16_k<NB, RELU, BIAS, OUT>), dim3(ntm * ntn), dim3(256), TileCfg16<NB>::LDS_BYTES, s,
                     C, ldc, A, lda, Wt, ldb, K, bias, ntm, ntn, nk0);
}
template <bool RELU, bool BIAS, bool SPLIT>
void launch_layer16_wide(hipStream_t s, __bf16* C, int ldc, const bf16raw* A, size_t lda, const bf16raw* Wt,
                         size_t ldb, int M, int K, const float* bias) {
  const int ntn = (ldc + WBN - 1) / WBN;
  const int npg = std::max(1, 32 / ntn);                    // panel groups per XCD: 8 * npg * ntn workgroups <= 256 CUs
  // the kernel addresses A with 32-bit byte offsets: slices of less than 2^31 bytes
  int max_rows = (int)(((size_t)1 << 31) / (lda * 2) / WBM) * WBM - WBM;
  if (const char* e = getenv("NNMPC_WIDE_MAX_ROWS")) {      // tests: force several slices on a small batch
    const int v = atoi(e) / WBM * WBM;
    if (v >= WBM) max_rows = std::min(max_rows, v);
  }
  for (int m = 0; m < M; m += max_rows) {
    const int rows = std::min(max_rows, M - m), ntm = rows / WBM;
    hipLaunchKernelGGL((gemm_nt_bf16_wide_k<RELU, BIAS, SPLIT>), dim3(8 * npg * ntn), dim3(512), W_LDS_BYTES, s,
                       C + (size_t)m * ldc * (SPLIT ? 2 : 1), ldc, A + (size_t)m * lda, lda, Wt, ldb, K, bias, ntm, ntn, npg);
  }
}
}  // namespace

extern "C" {

int nnmpc_nn_create(nnmpc_nn** out, int32_t nlayers, const int32_t* dims, const double* const* W,
                    const double* const* b, int32_t nx, int32_t nu, int32_t with_uprev,
                    const double* xscale, const double* ulb, const double* uub, int32_t use_bf16,
                    int32_t max_batch) {
  if (!out || nlayers < 1 || !dims || !W || !b || nx <= 0 || nu <= 0) { set_error("nnmpc_nn_create: bad arguments"); return NNMPC_EINVAL; }
  const int din = 2 * nx + (with_uprev ? 2 : 1) * nu;
  if (dims[0] != din || dims[nlayers] != nu) { set_error("nnmpc_nn_create: dims[0]=%d (want %d), dims[L]=%d (want %d)", dims[0], din, dims[nlayers], nu); return NNMPC_EINVAL; }
  if ((ulb == nullptr) != (uub == nullptr)) { set_error("nnmpc_nn_create: ulb and uub must both be given or both NULL"); return NNMPC_EINVAL; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { set_error("nnmpc_nn_create: no HIP device available (no CPU fallback)"); return NNMPC_EHIP; }
  nnmpc_nn* h = new nnmpc_nn();
  hipGetDevice(&h->device);
  h->use_bf16 = use_bf16 == 2 ? 2 : (use_bf16 != 0);
  h->split = h->use_bf16 == 2;
  h->nlayers = nlayers; h->nx = nx; h->nu = nu; h->with_uprev = with_uprev; h->clip = ulb != nullptr;
  h->max_batch = ((std::max(max_batch, 1) + 127) / 128) * 128;
  h->gemm_ms = h->total_ms = h->hidden_ms = 0; h->hidden_launches = 0;
  h->dims.assign(dims, dims + nlayers + 1);
  hipStreamCreate(&h->stream);
  hipEventCreate(&h->e0); hipEventCreate(&h->e1);
  hipFuncSetAttribute((const void*)gemm_nt_f32_k<128, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg<128>::LDS_FLOATS * 4);
  hipFuncSetAttribute((const void*)gemm_nt_f32_k<128, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg<128>::LDS_FLOATS * 4);
  hipFuncSetAttribute((const void*)gemm_nt_bf16_k<128, true, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg16<128>::LDS_BYTES);
  hipFuncSetAttribute((const void*)gemm_nt_bf16_k<128, true, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, TileCfg16<128>::LDS_BYTES);
  hipFuncSetAttribute((const void*)gemm_nt_bf16_wide_k<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES);
  hipFuncSetAttribute((const void*)gemm_nt_bf16_wide_k<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS_BYTES);
  h->maxw = 0;
  int rc = 0;
  for (int l = 0; l < nlayers && !rc; ++l) {
    const int kp = l == 0 ? ((dims[0] + 63) / 64) * 64 : h->npad[l - 1];
    // wide layers use 128-wide tiles (pad to 128), narrow ones (the Nu-wide head) 64
    const int np_ = dims[l + 1] > 64 ? ((dims[l + 1] + 127) / 128) * 128 : 64;
    h->kpad.push_back(kp); h->npad.push_back(np_);
    h->maxw = std::max(h->maxw, std::max(kp, np_));
    std::vector<float> wt((size_t)np_ * kp, 0.f), bb(np_, 0.f);
    for (int i = 0; i < dims[l]; ++i)
      for (int o = 0; o < dims[l + 1]; ++o) wt[(size_t)o * kp + i] = (float)W[l][(size_t)i * dims[l + 1] + o];
    if (l < nlayers - 1) {
      if (!b[l]) { set_error("nnmpc_nn_create: missing bias for hidden layer %d", l); rc = NNMPC_EINVAL; break; }
      for (int o = 0; o < dims[l + 1]; ++o) bb[o] = (float)b[l][o];
    }
    float *dw = nullptr, *db = nullptr;
    rc = nn_alloc(h, &dw, wt.size()); if (rc) break;
    rc = nn_alloc(h, &db, bb.size()); if (rc) break;
    hipMemcpy(dw, wt.data(), wt.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bb.data(), bb.size() * 4, hipMemcpyHostToDevice);
    h->Wt.push_back(dw); h->bias.push_back(db);
    if (h->use_bf16) {
      const bool lastl = l == nlayers - 1;
      const int k16 = ((dims[l] + 63) / 64) * 64;
      const int ldc = lastl ? 64 : ((dims[l + 1] + 63) / 64) * 64;
      const int nb = (!lastl && ldc >= 2 * WBN) ? WBN : (dims[l + 1] > 64 ? 128 : 64);   // WBN: the wide-tile kernel
      const int n16 = ((ldc + nb - 1) / nb) * nb;
      h->k16.push_back(k16); h->n16.push_back(n16); h->ldc16.push_back(ldc);
      // split path: rows of 3 k16 = [hi ; lo ; hi] against the activation planes walked hi, hi, lo
      const int kw = h->split ? 3 * k16 : k16;
      std::vector<bf16raw> w16((size_t)(n16 + 64) * kw, 0);   // + 64 zero rows: the wide kernel's staging loads may run past the last tile
      auto rne = [](float f) { unsigned u; memcpy(&u, &f, 4); u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; return (bf16raw)u; };   // round to nearest even
      auto tof = [](bf16raw b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
      for (int i = 0; i < dims[l]; ++i)
        for (int o = 0; o < dims[l + 1]; ++o) {
          const float f = (float)W[l][(size_t)i * dims[l + 1] + o];
          const bf16raw hi = rne(f);
          w16[(size_t)o * kw + i] = hi;
          if (h->split) { w16[(size_t)o * kw + k16 + i] = rne(f - tof(hi)); w16[(size_t)o * kw + 2 * k16 + i] = hi; }
        }
      bf16raw* d16 = nullptr;
      rc = nn_alloc(h, &d16, w16.size()); if (rc) break;
      hipMemcpy(d16, w16.data(), w16.size() * 2, hipMemcpyHostToDevice);
      h->Wt16.push_back(d16);
    }
  }
  std::vector<float> is(nx, 1.f);
  if (xscale) for (int i = 0; i < nx; ++i) is[i] = (float)(1.0 / xscale[i]);
  if (!rc) rc = nn_alloc(h, &h->inv_scale, nx);
  if (!rc) hipMemcpy(h->inv_scale, is.data(), nx * 4, hipMemcpyHostToDevice);
  if (!rc) rc = nn_alloc(h, &h->ulb, nu);
  if (!rc) rc = nn_alloc(h, &h->uub, nu);
  if (!rc && ulb) { hipMemcpy(h->ulb, ulb, nu * 8, hipMemcpyHostToDevice); hipMemcpy(h->uub, uub, nu * 8, hipMemcpyHostToDevice); }
  const size_t MB = h->max_batch;
  if (h->split) {                                          // rows of two bf16 planes of up to max(k16, ldc16) columns, in units of float
    int w = 0;
    for (int l = 0; l < nlayers; ++l) w = std::max(w, std::max(h->k16[l], h->ldc16[l]));
    h->maxw = std::max(h->maxw, (2 * w * 2 + 3) / 4);
  }
  if (!rc) rc = nn_alloc(h, &h->act[0], 2 * MB * h->maxw);
  if (!rc) rc = nn_alloc(h, &h->act[1], 2 * MB * h->maxw);
  if (!rc) rc = nn_alloc(h, &h->sx, MB * nx);
  if (!rc) rc = nn_alloc(h, &h->sxs, MB * nx);
  if (!rc) rc = nn_alloc(h, &h->suprev, MB * nu);
  if (!rc) rc = nn_alloc(h, &h->sus, MB * nu);
  if (!rc) rc = nn_alloc(h, &h->su, MB * nu);
  if (rc) { nnmpc_nn_destroy(h); return rc; }
  *out = h;
  return NNMPC_OK;
}

int nnmpc_nn_destroy(nnmpc_nn* h) {
  if (!h) return NNMPC_OK;
  hipDeviceSynchronize();
  for (void* p : h->allocs) hipFree(p);
  hipEventDestroy(h->e0); hipEventDestroy(h->e1);
  for (hipEvent_t e : h->eg) hipEventDestroy(e);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return NNMPC_OK;
}

int nnmpc_nn_forward(nnmpc_nn* h, int32_t B, const double* x, const double* uprev, const double* xs,
                     const double* us, double* u, int32_t ptr_kind) {
  if (!h || B < 0 || !x || !xs || !us || !u || (h->with_uprev && !uprev)) { set_error("nnmpc_nn_forward: bad arguments"); return NNMPC_EINVAL; }
  if (B == 0) return NNMPC_OK;
  HIPCHK(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int nx = h->nx, nu = h->nu, MB = h->max_batch;
  static const bool dbg = getenv("NNMPC_DEBUG_TIMING") != nullptr;
  const auto T0 = std::chrono::steady_clock::now();
  double gemm_ms = 0.0;
  hipEventRecord(h->e0, s);
  size_t nsub = 0;
  for (int b0 = 0; b0 < B; b0 += MB, ++nsub) {
    while (h->eg.size() < 3 * (nsub + 1)) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); h->eg.push_back(e); }
    const int nb = std::min(MB, B - b0);
    const int Bp = ((nb + 127) / 128) * 128;
    const double *dx, *dup, *dxs, *dus; double* du;
    if (ptr_kind == NNMPC_HOST) {
      HIPCHK(hipMemcpyAsync(h->sx, x + (size_t)b0 * nx, (size_t)nb * nx * 8, hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(h->sxs, xs + (size_t)b0 * nx, (size_t)nb * nx * 8, hipMemcpyHostToDevice, s));
      HIPCHK(hipMemcpyAsync(h->sus, us + (size_t)b0 * nu, (size_t)nb * nu * 8, hipMemcpyHostToDevice, s));
      if (h->with_uprev) HIPCHK(hipMemcpyAsync(h->suprev, uprev + (size_t)b0 * nu, (size_t)nb * nu * 8, hipMemcpyHostToDevice, s));
      dx = h->sx; dxs = h->sxs; dus = h->sus; dup = h->suprev; du = h->su;
    } else {
      dx = x + (size_t)b0 * nx; dxs = xs + (size_t)b0 * nx; dus = us + (size_t)b0 * nu;
      dup = h->with_uprev ? uprev + (size_t)b0 * nu : nullptr; du = u + (size_t)b0 * nu;
    }
    if (h->split)
      hipLaunchKernelGGL(nn_assemble_split_k, dim3(8192), dim3(256), 0, s, reinterpret_cast<__bf16*>(h->act[0]),
                         h->kpad[0], Bp, nb, nx, nu, h->with_uprev, dx, dup, dxs, dus, h->inv_scale);
    else if (h->use_bf16)
      hipLaunchKernelGGL(nn_assemble_k<__bf16>, dim3(8192), dim3(256), 0, s, reinterpret_cast<__bf16*>(h->act[0]),
                         h->kpad[0], Bp, nb, nx, nu, h->with_uprev, dx, dup, dxs, dus, h->inv_scale);
    else
      hipLaunchKernelGGL(nn_assemble_k<float>, dim3(8192), dim3(256), 0, s, h->act[0], h->kpad[0], Bp, nb, nx, nu,
                         h->with_uprev, dx, dup, dxs, dus, h->inv_scale);
    hipEventRecord(h->eg[3 * nsub], s);
    int cur = 0;
    const int M = 2 * Bp;
    for (int l = 0; l < h->nlayers; ++l) {
      const int K = h->kpad[l], N = h->npad[l];
      float* C = h->act[cur ^ 1];
      const float* A = h->act[cur];
      const bool last = l == h->nlayers - 1;
      if (last) hipEventRecord(h->eg[3 * nsub + 1], s);    // end of the hidden layers
      if (h->use_bf16) {
        const bf16raw* A16 = reinterpret_cast<const bf16raw*>(A);
        // split: one GEMM of three times the depth (K16), rows of two planes (lda), nk0 chunks per plane
        const int K16 = h->split ? 3 * h->k16[l] : h->k16[l], ldc = h->ldc16[l];
        const size_t lda = h->split ? 2 * (size_t)h->k16[l] : (size_t)h->k16[l];
        const int nk0 = h->split ? h->k16[l] / 64 : 0;
        if (!last && ldc >= 2 * WBN) {
          if (h->split) launch_layer16_wide<true, true, true>(s, reinterpret_cast<__bf16*>(C), ldc, A16, lda, h->Wt16[l], K16, M, K16, h->bias[l]);
          else launch_layer16_wide<true, true, false>(s, reinterpret_cast<__bf16*>(C), ldc, A16, lda, h->Wt16[l], K16, M, K16, h->bias[l]);
        } else if (h->n16[l] % 128 == 0) {
          if (last) launch_layer16<128, false, false, 0>(s, C, ldc, A16, lda, h->Wt16[l], K16, M, K16, nullptr, nk0);
          else if (h->split) launch_layer16<128, true, true, 2>(s, C, ldc, A16, lda, h->Wt16[l], K16, M, K16, h->bias[l], nk0);
          else launch_layer16<128, true, true, 1>(s, C, ldc, A16, lda, h->Wt16[l], K16, M, K16, h->bias[l]);
        } else {
          if (last) launch_layer16<64, false, false, 0>(s, C, ldc, A16, lda, h->Wt16[l], K16, M, K16, nullptr, nk0);
          else if (h->split) launch_layer16<64, true, true, 2>(s, C, ldc, A16, lda, h->Wt16[l], K16, M, K16, h->bias[l], nk0);
          else launch_layer16<64, true, true, 1>(s, C, ldc, A16, lda, h->Wt16[l], K16, M, K16, h->bias[l]);
        }
      } else if (N % 128 == 0) {
        if (last) launch_layer<128, false, false>(s, C, N, A, K, h->Wt[l], K, M, N, K, nullptr);
        else launch_layer<128, true, true>(s, C, N, A, K, h->Wt[l], K, M, N, K, h->bias[l]);
      } else {
        if (last) launch_layer<64, false, false>(s, C, N, A, K, h->Wt[l], K, M, N, K, nullptr);
        else launch_layer<64, true, true>(s, C, N, A, K, h->Wt[l], K, M, N, K, h->bias[l]);
      }
      cur ^= 1;
    }
    hipEventRecord(h->eg[3 * nsub + 2], s);
    hipLaunchKernelGGL(nn_combine_k, dim3(1024), dim3(256), 0, s, du, h->act[cur], h->npad[h->nlayers - 1], Bp,
                       nb, nu, dus, h->ulb, h->uub, h->clip);
    if (ptr_kind == NNMPC_HOST) {                           // the staging buffers are reused by the next sub-batch
      HIPCHK(hipMemcpyAsync(u + (size_t)b0 * nu, h->su, (size_t)nb * nu * 8, hipMemcpyDeviceToHost, s));
      HIPCHK(stream_sync(s));
    }
  }
  hipEventRecord(h->e1, s);
  const auto T1 = std::chrono::steady_clock::now();
  HIPCHK(stream_sync(s));
  const auto T2 = std::chrono::steady_clock::now();
  HIPCHK(hipGetLastError());
  float tot = 0.f;
  hipEventElapsedTime(&tot, h->e0, h->e1);
  double hid = 0.0;
  for (size_t i = 0; i < nsub; ++i) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, h->eg[3 * i], h->eg[3 * i + 2]); gemm_ms += ms;
    hipEventElapsedTime(&ms, h->eg[3 * i], h->eg[3 * i + 1]); hid += ms;
  }
  h->hidden_ms = hid; h->hidden_launches = (int)nsub * (h->nlayers - 1);
  h->gemm_ms = gemm_ms; h->total_ms = tot;
  if (dbg) {
    const auto T3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "nnmpc_nn_forward: enqueue %.3f ms, wait %.3f ms, post %.3f ms, device %.3f ms\n", ms(T0, T1), ms(T1, T2), ms(T2, T3), (double)tot);
  }
  return NNMPC_OK;
}

int nnmpc_nn_last_hidden_ms(nnmpc_nn* h, double* hidden_ms, int32_t* launches) {
  if (!h) return NNMPC_EINVAL;
  if (hidden_ms) *hidden_ms = h->hidden_ms;
  if (launches) *launches = h->hidden_launches;
  return NNMPC_OK;
}

int nnmpc_nn_last_ms(nnmpc_nn* h, double* gemm_ms, double* total_ms) {
  if (!h) return NNMPC_EINVAL;
  if (gemm_ms) *gemm_ms = h->gemm_ms;
  if (total_ms) *total_ms = h->total_ms;
  return NNMPC_OK;
}

}  // extern "C"
