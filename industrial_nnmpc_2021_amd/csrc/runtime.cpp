// Device plumbing and the RCCL side of libnnmpc_hip.so.
//
//   nnmpc_dev_* / nnmpc_host_* / nnmpc_set_device ...   HBM buffers, copies and synchronisation for a host program
//                                                       that binds nothing but this library (bench.py, the mirror
//                                                       classes): the reference is plain numpy on the host, so every
//                                                       device-resident array of the path is owned through these.
//   nnmpc_comm_*                                        one process per GPU; the per-rank result blocks of a sharded
//                                                       sample batch meet on one rank through ONE gather over xGMI.
//                                                       Replaces the file-system rendezvous of the reference
//                                                       (per-task .h5py files concatenated by _post_process_data,
//                                                       lib/controller_evaluation.py:273-295; the tasks themselves are
//                                                       OS processes / cluster jobs, lib/linearMPC.py:814-825).
//
// librccl is opened with dlopen at the first nnmpc_comm_* call (a process that never shards never loads it).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/nnmpc.h"
#include "common.h"

using namespace nnmpc;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s: %s", #x, hipGetErrorString(e_)); return NNMPC_EHIP; } } while (0)

extern "C" {

int nnmpc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int nnmpc_set_device(int32_t dev) { HIPCHK(hipSetDevice(dev)); return NNMPC_OK; }
int nnmpc_device_synchronize(void) {
  // polling first (the legacy default stream is done when every blocking stream is): a blocking wait can take milliseconds to
  // wake the caller on some hosts (common.h)
  for (int i = 0; i < 200000 && hipStreamQuery(0) == hipErrorNotReady; ++i) {}
  HIPCHK(hipDeviceSynchronize());
  return NNMPC_OK;
}
int nnmpc_dev_mem_info(uint64_t* free_bytes, uint64_t* total_bytes) {
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return NNMPC_OK;
}
int nnmpc_dev_malloc(void** out, uint64_t bytes) {
  if (!out) { set_error("nnmpc_dev_malloc: null output"); return NNMPC_EINVAL; }
  *out = nullptr;
  if (bytes == 0) return NNMPC_OK;
  const hipError_t e = hipMalloc(out, (size_t)bytes);
  if (e != hipSuccess) { set_error("hipMalloc(%llu bytes): %s", (unsigned long long)bytes, hipGetErrorString(e)); return NNMPC_ENOMEM; }
  return NNMPC_OK;
}
int nnmpc_dev_free(void* p) { if (p) HIPCHK(hipFree(p)); return NNMPC_OK; }
int nnmpc_dev_memset(void* p, int32_t value, uint64_t bytes) { if (bytes) HIPCHK(hipMemset(p, value, (size_t)bytes)); return NNMPC_OK; }
int nnmpc_memcpy_h2d(void* dst, const void* src, uint64_t bytes) {
  if (bytes) HIPCHK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
  return NNMPC_OK;
}
int nnmpc_memcpy_d2h(void* dst, const void* src, uint64_t bytes) {
  if (bytes) HIPCHK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
  return NNMPC_OK;
}
int nnmpc_memcpy_d2d(void* dst, const void* src, uint64_t bytes) {
  if (bytes) HIPCHK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice));
  return NNMPC_OK;
}
int nnmpc_host_alloc_pinned(void** out, uint64_t bytes) {
  if (!out) { set_error("nnmpc_host_alloc_pinned: null output"); return NNMPC_EINVAL; }
  *out = nullptr;
  if (bytes == 0) return NNMPC_OK;
  const hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
  if (e != hipSuccess) { set_error("hipHostMalloc(%llu bytes): %s", (unsigned long long)bytes, hipGetErrorString(e)); return NNMPC_ENOMEM; }
  return NNMPC_OK;
}
int nnmpc_host_free_pinned(void* p) { if (p) HIPCHK(hipHostFree(p)); return NNMPC_OK; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time.  Only the handful of entry points the gather needs.
namespace {

typedef struct { char internal[128]; } rcclUniqueId;       // NCCL_UNIQUE_ID_BYTES = 128 (rccl.h:40-43)
typedef void* rcclComm_t;
enum { RCCL_SUCCESS = 0 };
enum { RCCL_INT8 = 0, RCCL_INT32 = 2, RCCL_FLOAT64 = 8 };  // ncclDataType_t (rccl.h)
enum { RCCL_SUM = 0, RCCL_MAX = 2 };                       // ncclRedOp_t

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId*) = nullptr;
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.lib) return NNMPC_OK;
  const char* names[] = {getenv("NNMPC_RCCL_PATH"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) { set_error("librccl.so.1 not found (set NNMPC_RCCL_PATH): %s", dlerror()); return NNMPC_ENOTIMPL; }
  RcclApi a;
  a.lib = lib;
#define SYM_(field, name) *(void**)(&a.field) = dlsym(lib, name); if (!a.field) { set_error("librccl: missing symbol %s", name); dlclose(lib); return NNMPC_ENOTIMPL; }
  SYM_(GetUniqueId, "ncclGetUniqueId") SYM_(CommInitRank, "ncclCommInitRank") SYM_(CommDestroy, "ncclCommDestroy")
  SYM_(Send, "ncclSend") SYM_(Recv, "ncclRecv") SYM_(AllReduce, "ncclAllReduce")
  SYM_(GroupStart, "ncclGroupStart") SYM_(GroupEnd, "ncclGroupEnd") SYM_(GetErrorString, "ncclGetErrorString")
#undef SYM_
  g_rccl = a;
  return NNMPC_OK;
}

#define RCCLCHK(x) do { const int r_ = (x); if (r_ != RCCL_SUCCESS) { set_error("%s: %s", #x, g_rccl.GetErrorString(r_)); return NNMPC_EHIP; } } while (0)

}  // namespace

struct nnmpc_comm {
  rcclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  double* scratch = nullptr;   // 8 doubles on the device (barrier / max-reduce payload)
};

extern "C" {

int nnmpc_comm_unique_id(void* id128) {
  if (!id128) { set_error("nnmpc_comm_unique_id: null buffer"); return NNMPC_EINVAL; }
  const int rc = rccl_load();
  if (rc) return rc;
  rcclUniqueId id;
  RCCLCHK(g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, sizeof(id.internal));
  return NNMPC_OK;
}

int nnmpc_comm_init(nnmpc_comm** out, const void* id128, int32_t rank, int32_t world) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) {
    set_error("nnmpc_comm_init: bad arguments (rank=%d world=%d)", rank, world);
    return NNMPC_EINVAL;
  }
  const int rc = rccl_load();
  if (rc) return rc;
  nnmpc_comm* c = new nnmpc_comm();
  c->rank = rank; c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess) { set_error("nnmpc_comm_init: no HIP device"); delete c; return NNMPC_EHIP; }
  rcclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  const int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != RCCL_SUCCESS) { set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r)); delete c; return NNMPC_EHIP; }
  if (hipStreamCreate(&c->stream) != hipSuccess || hipMalloc((void**)&c->scratch, 8 * sizeof(double)) != hipSuccess) {
    set_error("nnmpc_comm_init: stream / scratch allocation failed");
    nnmpc_comm_destroy(c);
    return NNMPC_EHIP;
  }
  *out = c;
  return NNMPC_OK;
}

int nnmpc_comm_destroy(nnmpc_comm* c) {
  if (!c) return NNMPC_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm) g_rccl.CommDestroy(c->comm);
  if (c->scratch) hipFree(c->scratch);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return NNMPC_OK;
}

int nnmpc_comm_rank(nnmpc_comm* c) { return c ? c->rank : -1; }
int nnmpc_comm_world(nnmpc_comm* c) { return c ? c->world : -1; }

/* Row blocks of every rank -> `root`, in rank order (rank r's block starts at row sum(rows[0..r))): ncclSend from every
 * rank, the matching ncclRecv's on the root, fused in one group = one gather over xGMI, ragged shards included.
 * send: rows[rank] x row_doubles (device), recv: sum(rows) x row_doubles (device; only read on root).  Returns after the
 * transfer has completed (device work queued before the call on other streams must have been synchronised by the caller:
 * the library's solve entry points return synchronised). */
int nnmpc_comm_gather_rows(nnmpc_comm* c, const double* send, const int64_t* rows, int32_t row_doubles, double* recv, int32_t root) {
  if (!c || !rows || row_doubles <= 0 || root < 0 || root >= c->world) { set_error("nnmpc_comm_gather_rows: bad arguments"); return NNMPC_EINVAL; }
  for (int r = 0; r < c->world; ++r) if (rows[r] < 0) { set_error("nnmpc_comm_gather_rows: negative row count"); return NNMPC_EINVAL; }
  if (rows[c->rank] > 0 && !send) { set_error("nnmpc_comm_gather_rows: null send buffer"); return NNMPC_EINVAL; }
  if (c->rank == root && !recv) { set_error("nnmpc_comm_gather_rows: null receive buffer on the root"); return NNMPC_EINVAL; }
  HIPCHK(hipSetDevice(c->device));
  RCCLCHK(g_rccl.GroupStart());
  // An error between GroupStart and GroupEnd must not leave the group open: every later RCCL call of this thread would be
  // queued into it instead of run.  Close the group (ignoring its result), drain the stream, then report the first error.
  int err = RCCL_SUCCESS;
  if (rows[c->rank] > 0) err = g_rccl.Send(send, (size_t)rows[c->rank] * row_doubles, RCCL_FLOAT64, root, c->comm, c->stream);
  if (err == RCCL_SUCCESS && c->rank == root) {
    size_t off = 0;
    for (int r = 0; r < c->world && err == RCCL_SUCCESS; ++r) {
      if (rows[r] > 0) err = g_rccl.Recv(recv + off, (size_t)rows[r] * row_doubles, RCCL_FLOAT64, r, c->comm, c->stream);
      off += (size_t)rows[r] * row_doubles;
    }
  }
  if (err != RCCL_SUCCESS) {
    g_rccl.GroupEnd();
    hipStreamSynchronize(c->stream);
    set_error("nnmpc_comm_gather_rows: ncclSend / ncclRecv: %s", g_rccl.GetErrorString(err));
    return NNMPC_EHIP;
  }
  RCCLCHK(g_rccl.GroupEnd());
  HIPCHK(stream_sync(c->stream));
  return NNMPC_OK;
}

/* value <- max over ranks (host scalar in/out): the bench's max-over-ranks step time */
int nnmpc_comm_allreduce_max(nnmpc_comm* c, double* value) {
  if (!c || !value) { set_error("nnmpc_comm_allreduce_max: bad arguments"); return NNMPC_EINVAL; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMemcpyAsync(c->scratch, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
  RCCLCHK(g_rccl.AllReduce(c->scratch, c->scratch + 1, 1, RCCL_FLOAT64, RCCL_MAX, c->comm, c->stream));
  HIPCHK(hipMemcpyAsync(value, c->scratch + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(stream_sync(c->stream));
  return NNMPC_OK;
}

/* all ranks have arrived (and this device is idle) when the call returns */
int nnmpc_comm_barrier(nnmpc_comm* c) {
  if (!c) { set_error("nnmpc_comm_barrier: null communicator"); return NNMPC_EINVAL; }
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipDeviceSynchronize());
  double one = 1.0;
  return nnmpc_comm_allreduce_max(c, &one);
}

}  // extern "C"
