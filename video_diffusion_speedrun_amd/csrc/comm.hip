// Sharding runtime of the C ABI (include/vds.h "parameter / gradient sharding"): the collectives the reference
// gets from FSDP2 `fully_shard` (model.py:512-542 -- bf16 all-gather of a shard group's parameters before use, fp32
// reduce-scatter-average of its gradients after its backward), driven from C++ straight on RCCL over xGMI.
//
//   * one communicator per process (one process per GPU), created from a 128-byte unique id the host hands in
//     (rank 0 draws it with vds_comm_unique_id and ships it to the others by any means it has);
//   * every call is asynchronous on the HIP stream passed in -- the host runs them on a dedicated communication
//     stream and orders them against the compute stream with events (fsdp.ShardRuntime);
//   * buffers are the flat per-group buffers of params.FlatGroup: ONE all-gather and ONE reduce-scatter per group
//     per step, sized for 288 GB parts (DiT-XL: 80 MB bf16 / 160 MB fp32 per block group);
//   * two schedules for the reduce-scatter: RCCL's own (default) and an explicit all-pairs exchange
//     (VDS_COMM_SCHEDULE=allpairs): every rank sends chunk j of its gradient buffer directly to rank j -- xGMI is a
//     full mesh of point-to-point links, so all 7 links of a GPU carry one chunk each, one hop -- and a local kernel
//     averages the W chunks in a fixed order (deterministic, unlike a ring whose order depends on the rank).
//
// RCCL is bound at run time (dlopen of the librccl.so.1 torch has already loaded, else the ROCm one): the kernel
// library itself carries no link-time dependency on it and still loads on a box without RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vds.h"
#include "common.h"
#include "config.h"

namespace vdserr {
static thread_local char g_msg[512] = "";
void set(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_msg, sizeof(g_msg), fmt, ap);
  va_end(ap);
}
const char* get() { return g_msg; }
void clear() { g_msg[0] = 0; }
}  // namespace vdserr

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                                hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
};

Rccl R;
ncclComm_t g_comm = nullptr;
int g_rank = -1, g_world = 0;
bool g_allpairs = false;

template <typename F>
bool sym(F& f, const char* name) {
  f = reinterpret_cast<F>(dlsym(R.lib, name));
  if (!f) vdserr::set("RCCL symbol %s not found", name);
  return f != nullptr;
}

bool bind() {
  if (R.lib) return true;
  const char* cands[] = {vdscfg::rccl_path(), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  // the copy the process already holds (torch links one): a second RCCL in one process only wastes memory
  R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  for (const char* c : cands)
    if (!R.lib && c && *c) R.lib = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
  if (!R.lib) {
    vdserr::set("cannot load RCCL (librccl.so.1): %s", dlerror());
    return false;
  }
  bool ok = sym(R.GetUniqueId, "ncclGetUniqueId") && sym(R.CommInitRank, "ncclCommInitRank") &&
            sym(R.CommDestroy, "ncclCommDestroy") && sym(R.AllGather, "ncclAllGather") &&
            sym(R.ReduceScatter, "ncclReduceScatter") && sym(R.AllReduce, "ncclAllReduce") &&
            sym(R.Send, "ncclSend") && sym(R.Recv, "ncclRecv") && sym(R.GroupStart, "ncclGroupStart") &&
            sym(R.GroupEnd, "ncclGroupEnd") && sym(R.GetErrorString, "ncclGetErrorString") &&
            sym(R.GetVersion, "ncclGetVersion");
  if (!ok) R.lib = nullptr;
  return ok;
}

int fail(ncclResult_t r, const char* what) {
  vdserr::set("%s: RCCL error %d (%s)", what, (int)r, R.GetErrorString ? R.GetErrorString(r) : "?");
  return VDS_ERR_LAUNCH;
}
#define RCCL_TRY(expr, what)                  \
  do {                                        \
    ncclResult_t r_ = (expr);                 \
    if (r_ != ncclSuccess) return fail(r_, what); \
  } while (0)

int need_comm(const char* what) {
  if (g_comm) return VDS_OK;
  vdserr::set("%s: no communicator (call vds_comm_init first)", what);
  return VDS_ERR_ARG;
}

// out[i] = (own[i] + sum_p stage[p][i]) / W, summed in rank order (the own chunk takes its rank's place), fp32.
// own: this rank's chunk inside its full gradient buffer; stage: W-1 received chunks, peers in increasing rank order.
__global__ __launch_bounds__(256) void allpairs_average_kernel(const float* __restrict__ own,
                                                               const float* __restrict__ stage, float* __restrict__ out,
                                                               long n4, long n, int W, int rank) {
  const float inv = 1.0f / (float)W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int p = 0;
    for (int r = 0; r < W; ++r) {
      const f32x4 v = (r == rank) ? reinterpret_cast<const f32x4*>(own)[i]
                                  : reinterpret_cast<const f32x4*>(stage + (long)(p++) * n)[i];
      acc += v;
    }
    reinterpret_cast<f32x4*>(out)[i] = acc * inv;
  }
}

}  // namespace

extern "C" int vds_comm_unique_id(void* out, size_t bytes) {
  vdserr::clear();  // a message describes the LAST failing call only
  if (!out || bytes < NCCL_UNIQUE_ID_BYTES) return VDS_ERR_ARG;
  if (!bind()) return VDS_ERR_UNSUPPORTED;
  ncclUniqueId id;
  RCCL_TRY(R.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out, &id, NCCL_UNIQUE_ID_BYTES);
  return VDS_OK;
}

extern "C" int vds_comm_available(void) {
  vdserr::clear();
  return bind() ? VDS_OK : VDS_ERR_UNSUPPORTED;
}

extern "C" int vds_comm_init(int32_t rank, int32_t world, const void* unique_id, size_t bytes) {
  vdserr::clear();  // a message describes the LAST failing call only
  if (!unique_id || bytes < NCCL_UNIQUE_ID_BYTES || world < 1 || rank < 0 || rank >= world) return VDS_ERR_ARG;
  if (g_comm) {
    vdserr::set("vds_comm_init: a communicator already exists (vds_comm_destroy it first)");
    return VDS_ERR_ARG;
  }
  if (!bind()) return VDS_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, unique_id, NCCL_UNIQUE_ID_BYTES);
  RCCL_TRY(R.CommInitRank(&g_comm, world, id, rank), "ncclCommInitRank");  // on the calling thread's current HIP device
  g_rank = rank;
  g_world = world;
  g_allpairs = vdscfg::geti(vdscfg::COMM_ALLPAIRS) != 0;  // (VDS_COMM_SCHEDULE=allpairs at load, or vds_knob_set)
  return VDS_OK;
}

extern "C" int vds_comm_info(int32_t* rank, int32_t* world, int32_t* rccl_version, int32_t* allpairs) {
  if (rank) *rank = g_rank;
  if (world) *world = g_world;
  if (allpairs) *allpairs = g_allpairs ? 1 : 0;
  if (rccl_version) {
    *rccl_version = 0;
    if (bind()) (void)R.GetVersion(rccl_version);
    else vdserr::clear();  // reported through the version field (0), not through vds_last_error
  }
  return g_comm ? VDS_OK : VDS_ERR_ARG;
}

extern "C" int vds_comm_destroy(void) {
  vdserr::clear();  // a message describes the LAST failing call only
  if (!g_comm) return VDS_OK;
  ncclComm_t c = g_comm;
  g_comm = nullptr;
  g_rank = -1;
  g_world = 0;
  RCCL_TRY(R.CommDestroy(c), "ncclCommDestroy");
  return VDS_OK;
}

extern "C" int vds_all_gather_bf16(const void* shard, void* full, int64_t shard_elems, vds_stream_t stream) {
  vdserr::clear();  // a message describes the LAST failing call only
  if (int e = need_comm("vds_all_gather_bf16")) return e;
  if (!shard || !full || shard_elems <= 0) return VDS_ERR_ARG;
  RCCL_TRY(R.AllGather(shard, full, (size_t)shard_elems, ncclBfloat16, g_comm, (hipStream_t)stream), "ncclAllGather(bf16)");
  return VDS_OK;
}

extern "C" int vds_all_gather_f32(const float* shard, float* full, int64_t shard_elems, vds_stream_t stream) {
  vdserr::clear();  // a message describes the LAST failing call only
  if (int e = need_comm("vds_all_gather_f32")) return e;
  if (!shard || !full || shard_elems <= 0) return VDS_ERR_ARG;
  RCCL_TRY(R.AllGather(shard, full, (size_t)shard_elems, ncclFloat32, g_comm, (hipStream_t)stream), "ncclAllGather(f32)");
  return VDS_OK;
}

extern "C" int vds_average_chunks_f32(const float* own, const float* staged, float* out, int64_t n, int32_t world,
                                      int32_t rank, vds_stream_t stream) {
  if (!own || !out || n <= 0 || (n & 3) || world < 1 || rank < 0 || rank >= world || (world > 1 && !staged))
    return VDS_ERR_ARG;
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(allpairs_average_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, own, staged,
                     out, n4, (long)n, world, rank);
  return hipGetLastError() == hipSuccess ? VDS_OK : VDS_ERR_LAUNCH;
}

extern "C" size_t vds_reduce_scatter_workspace_bytes(int64_t shard_elems) {
  if (!g_comm || !g_allpairs || g_world < 2 || shard_elems <= 0) return 0;
  return (size_t)(g_world - 1) * (size_t)shard_elems * sizeof(float);
}

extern "C" int vds_reduce_scatter_f32_avg(const float* full, float* shard, int64_t shard_elems, void* workspace,
                                          size_t ws_bytes, vds_stream_t stream) {
  vdserr::clear();  // a message describes the LAST failing call only
  if (int e = need_comm("vds_reduce_scatter_f32_avg")) return e;
  if (!full || !shard || shard_elems <= 0) return VDS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!g_allpairs || g_world == 1) {
    RCCL_TRY(R.ReduceScatter(full, shard, (size_t)shard_elems, ncclFloat32, ncclAvg, g_comm, s), "ncclReduceScatter(f32, avg)");
    return VDS_OK;
  }
  // all-pairs: chunk j of my buffer goes straight to rank j; the W-1 chunks addressed to me land in the workspace
  if ((shard_elems & 3) || !workspace || ws_bytes < vds_reduce_scatter_workspace_bytes(shard_elems)) {
    vdserr::set("vds_reduce_scatter_f32_avg(allpairs): needs shard_elems %% 4 == 0 and %zu workspace bytes",
                vds_reduce_scatter_workspace_bytes(shard_elems));
    return VDS_ERR_ARG;
  }
  float* stage = (float*)workspace;
  RCCL_TRY(R.GroupStart(), "ncclGroupStart");
  int p = 0;
  ncclResult_t bad = ncclSuccess;
  const char* bad_what = "";
  for (int r = 0; r < g_world && bad == ncclSuccess; ++r) {
    if (r == g_rank) continue;
    bad = R.Send(full + (long)r * shard_elems, (size_t)shard_elems, ncclFloat32, r, g_comm, s);
    bad_what = "ncclSend";
    if (bad != ncclSuccess) break;
    bad = R.Recv(stage + (long)(p++) * shard_elems, (size_t)shard_elems, ncclFloat32, r, g_comm, s);
    bad_what = "ncclRecv";
  }
  const ncclResult_t ge = R.GroupEnd();  // always: an error must not leave the group open
  if (bad != ncclSuccess) return fail(bad, bad_what);
  if (ge != ncclSuccess) return fail(ge, "ncclGroupEnd");
  return vds_average_chunks_f32(full + (long)g_rank * shard_elems, stage, shard, shard_elems, g_world, g_rank, stream);
}

extern "C" int vds_all_reduce_f32_avg(float* buf, int64_t n, vds_stream_t stream) {
  vdserr::clear();  // a message describes the LAST failing call only
  if (int e = need_comm("vds_all_reduce_f32_avg")) return e;
  if (!buf || n <= 0) return VDS_ERR_ARG;
  RCCL_TRY(R.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclAvg, g_comm, (hipStream_t)stream), "ncclAllReduce(f32, avg)");
  return VDS_OK;
}
