// The gradient exchange of the data-parallel training step on RCCL, owned by the library (last row; the
// reference: DDP's bucketed all-reduce under Lightning, train.py:461-474 `strategy="auto"`).
//
//   mh_comm_unique_id   rank 0 makes the 128-byte rendezvous id (ncclGetUniqueId); the host ships it to the other ranks by
//                       whatever channel it has (torch.distributed's store, MPI, a file)
//   mh_comm_init        one communicator per process/GPU (ncclCommInitRank); the handle also owns the pre-multiplied-sum
//                       reduction operators (scale 1/world) for bf16 and fp32
//   mh_comm_allreduce   in-place all-reduce of a contiguous range of the flat gradient buffer on the caller's stream:
//                       SUM, or the data-parallel MEAN as ONE collective (ncclRedOpCreatePreMulSum: every rank's contribution
//                       is multiplied by 1/world inside the reduction -- DDP's "divide, then sum" without the extra pass over
//                       the bucket that a separate division kernel costs)
//   mh_comm_broadcast   parameters from a root rank (DDP's constructor broadcast)
//   mh_comm_destroy
//
// librccl is NOT a link-time dependency of libmidihip.so: it is opened on the first mh_comm_* call (dlopen by soname, so a
// process that already holds RCCL -- PyTorch's own copy, say -- shares that one), and a box without it can still run every
// other entry point.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): ring collectives are per-link bound, so the host
// side ships few, large buckets (32 MB) back to front while the backward still runs (train.py: GradReducer).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <set>

#include "common.h"

namespace {

struct Api {
  void* so = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*RedOpCreatePreMulSum)(ncclRedOp_t*, void*, ncclDataType_t, ncclScalarResidence_t, ncclComm_t) = nullptr;
  ncclResult_t (*RedOpDestroy)(ncclRedOp_t, ncclComm_t) = nullptr;
};
Api g_api;
std::once_flag g_once;
char g_load_error[256] = "";

void load_api() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_api.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_api.so) break;
  }
  if (!g_api.so) {
    snprintf(g_load_error, sizeof(g_load_error), "librccl.so.1 could not be opened: %s", dlerror());
    return;
  }
#define MH_SYM(field, sym)                                                            \
  g_api.field = reinterpret_cast<decltype(g_api.field)>(dlsym(g_api.so, sym));        \
  if (!g_api.field) {                                                                 \
    snprintf(g_load_error, sizeof(g_load_error), "librccl has no symbol %s", sym);    \
    return;                                                                           \
  }
  MH_SYM(GetVersion, "ncclGetVersion")
  MH_SYM(GetUniqueId, "ncclGetUniqueId")
  MH_SYM(CommInitRank, "ncclCommInitRank")
  MH_SYM(CommDestroy, "ncclCommDestroy")
  MH_SYM(GetErrorString, "ncclGetErrorString")
  MH_SYM(AllReduce, "ncclAllReduce")
  MH_SYM(Broadcast, "ncclBroadcast")
  MH_SYM(RedOpCreatePreMulSum, "ncclRedOpCreatePreMulSum")
  MH_SYM(RedOpDestroy, "ncclRedOpDestroy")
#undef MH_SYM
}

bool api_ready() {
  std::call_once(g_once, load_api);
  if (g_load_error[0]) {
    mh_set_error("comm: %s", g_load_error);
    return false;
  }
  return true;
}

#define MH_NCCL(call, what)                                                           \
  do {                                                                                \
    ncclResult_t r_ = (call);                                                         \
    if (r_ != ncclSuccess) {                                                          \
      mh_set_error("comm: %s failed: %s", what, g_api.GetErrorString(r_));            \
      return MH_ERR_LAUNCH;                                                           \
    }                                                                                 \
  } while (0)

struct Comm {
  uint32_t magic;
  ncclComm_t comm;
  int rank, world, device, version;
  ncclRedOp_t mean_bf16, mean_f32;
  bool have_mean;
};
constexpr uint32_t MAGIC = 0x4d48434du;  // "MHCM"

// Live handles: a handle is only dereferenced while it is in this set, so a second mh_comm_destroy, or any call after destroy,
// is refused instead of reading freed memory.
std::mutex g_live_mu;
std::set<void*> g_live;

Comm* as_comm(void* h) {
  if (h == nullptr) return nullptr;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (g_live.find(h) == g_live.end()) return nullptr;
  }
  Comm* c = static_cast<Comm*>(h);
  return (c->magic == MAGIC) ? c : nullptr;
}

}  // namespace

extern "C" int mh_comm_unique_id(void* id128) {
  MH_REQUIRE(id128 != nullptr, "comm_unique_id: null buffer");
  if (!api_ready()) return MH_ERR_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == 128, "the C-ABI hands the id over as 128 bytes");
  ncclUniqueId id;
  MH_NCCL(g_api.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return MH_OK;
}

extern "C" int mh_comm_init(int rank, int world, const void* id128, int device, void** comm_out) {
  MH_REQUIRE(comm_out != nullptr && id128 != nullptr, "comm_init: null argument");
  MH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
  if (!api_ready()) return MH_ERR_UNSUPPORTED;
  // ncclCommInitRank binds the communicator to the CURRENT device: switch for the call and give the caller's device back
  int prev_dev = -1;
  (void)hipGetDevice(&prev_dev);
  hipError_t e = hipSetDevice(device);
  MH_REQUIRE(e == hipSuccess, "comm_init: hipSetDevice(%d): %s", device, hipGetErrorString(e));
  struct Restore {
    int d;
    ~Restore() {
      if (d >= 0) (void)hipSetDevice(d);
    }
  } restore{prev_dev};
  Comm* c = new Comm();
  c->magic = MAGIC;
  c->rank = rank;
  c->world = world;
  c->device = device;
  c->have_mean = false;
  g_api.GetVersion(&c->version);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    mh_set_error("comm: ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_api.GetErrorString(r));
    delete c;
    return MH_ERR_LAUNCH;
  }
  // the data-parallel mean as one collective: contributions pre-multiplied by 1/world inside the reduction
  const float inv = 1.0f / (float)world;
  bf16 inv_b = (bf16)inv;
  float inv_f = inv;
  ncclResult_t r1 = g_api.RedOpCreatePreMulSum(&c->mean_bf16, &inv_b, ncclBfloat16, ncclScalarHostImmediate, c->comm);
  ncclResult_t r2 = g_api.RedOpCreatePreMulSum(&c->mean_f32, &inv_f, ncclFloat32, ncclScalarHostImmediate, c->comm);
  if (r1 != ncclSuccess || r2 != ncclSuccess) {
    mh_set_error("comm: ncclRedOpCreatePreMulSum failed: %s", g_api.GetErrorString(r1 != ncclSuccess ? r1 : r2));
    g_api.CommDestroy(c->comm);
    delete c;
    return MH_ERR_LAUNCH;
  }
  c->have_mean = true;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.insert(c);
  }
  *comm_out = c;
  return MH_OK;
}

extern "C" int mh_comm_info(void* comm, int* rank, int* world, int* rccl_version) {
  Comm* c = as_comm(comm);
  MH_REQUIRE(c != nullptr, "comm_info: not a communicator handle");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (rccl_version) *rccl_version = c->version;
  return MH_OK;
}

extern "C" int mh_comm_allreduce(void* comm, void* buf, int64_t count, int dtype, int mean, void* stream) {
  Comm* c = as_comm(comm);
  MH_REQUIRE(c != nullptr, "comm_allreduce: not a communicator handle");
  MH_REQUIRE(buf != nullptr && count > 0, "comm_allreduce: empty range");
  MH_REQUIRE(dtype == MH_BF16 || dtype == MH_F32, "comm_allreduce: dtype %d", dtype);
  const ncclDataType_t dt = dtype == MH_BF16 ? ncclBfloat16 : ncclFloat32;
  const ncclRedOp_t op = mean ? (dtype == MH_BF16 ? c->mean_bf16 : c->mean_f32) : ncclSum;
  MH_NCCL(g_api.AllReduce(buf, buf, (size_t)count, dt, op, c->comm, (hipStream_t)stream), "ncclAllReduce");
  return MH_OK;
}

extern "C" int mh_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* stream) {
  Comm* c = as_comm(comm);
  MH_REQUIRE(c != nullptr, "comm_broadcast: not a communicator handle");
  MH_REQUIRE(buf != nullptr && count > 0 && root >= 0 && root < c->world, "comm_broadcast: bad arguments");
  MH_REQUIRE(dtype == MH_BF16 || dtype == MH_F32, "comm_broadcast: dtype %d", dtype);
  MH_NCCL(g_api.Broadcast(buf, buf, (size_t)count, dtype == MH_BF16 ? ncclBfloat16 : ncclFloat32, root, c->comm,
                          (hipStream_t)stream),
          "ncclBroadcast");
  return MH_OK;
}

extern "C" int mh_comm_destroy(void* comm) {
  // the membership test and the removal are ONE critical section: of two threads destroying the same handle exactly one gets it
  // (a collective racing a destroy on another thread is still the caller's bug, as with ncclCommDestroy itself)
  Comm* c = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (comm != nullptr && g_live.erase(comm) == 1) c = static_cast<Comm*>(comm);
  }
  MH_REQUIRE(c != nullptr && c->magic == MAGIC, "comm_destroy: not a (live) communicator handle");
  if (c->have_mean) {
    g_api.RedOpDestroy(c->mean_bf16, c->comm);
    g_api.RedOpDestroy(c->mean_f32, c->comm);
  }
  ncclResult_t r = g_api.CommDestroy(c->comm);
  c->magic = 0;
  delete c;
  if (r != ncclSuccess) {
    mh_set_error("comm: ncclCommDestroy failed: %s", g_api.GetErrorString(r));
    return MH_ERR_LAUNCH;
  }
  return MH_OK;
}
