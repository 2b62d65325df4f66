// comm.hip — the sharded path's two exchanges inside the C ABI, over RCCL:
//   * trk_allreduce_f64: sum of a few device doubles over the ranks (inner products, Gram rows: np.dot / np.linalg.norm of
//     the reference's solver loops once the frames of a dynamic problem are spread over GPUs, io.py:420);
//   * trk_halo_exchange: one-frame shift between time-neighbours for the temporal rows of the space-time regulariser
//     (operators.py:39-45).
// A host that binds libtrk.so without PyTorch gets the whole sharded path from the library; the Python layer of this repo
// issues the same two exchanges through torch.distributed by default (one communicator per process, shared with torch) and
// through these entry points with TRK_COMM=rccl (trips_py_amd/dist.py).
//
// RCCL is resolved at run time (dlopen / dlsym), so libtrk.so has no link-time dependency on it: single-GPU users never load
// it, and a process that already holds an RCCL (PyTorch bundles one) shares that copy.
#include "trk_internal.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

using namespace trk;

namespace {

// the slice of the NCCL / RCCL API in use (rccl.h: same ABI as NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclFloat32 = 7, ncclFloat64 = 8 };     // ncclDataType_t
enum { ncclSum = 0 };                          // ncclRedOp_t

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;
bool g_ok = false;
char g_why[256] = "no such library";      // why loading failed: dlerror() is cleared by reading it, so it is saved once, here

void load_rccl() {
  // a copy that is already in the process (PyTorch's) first; then the system's
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
  if (!h)
    for (const char* n : names) {
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
      if (const char* e = dlerror()) std::snprintf(g_why, sizeof(g_why), "%s", e);      // the last real attempt's message
    }
  if (!h) return;
  g_rccl.lib = h;
#define SYM(field, name) *(void**)(&g_rccl.field) = dlsym(h, name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  g_ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.Send && g_rccl.Recv &&
         g_rccl.GroupStart && g_rccl.GroupEnd;
  if (!g_ok) std::snprintf(g_why, sizeof(g_why), "the library lacks part of the NCCL 2 API");
}

int need_rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_ok) return fail(TRK_ENCCL, "RCCL is not available (librccl.so could not be loaded or lacks the NCCL 2 API): %s", g_why);
  return TRK_OK;
}

#define TRK_NCCL(expr)                                                                                      \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess)                                                                                  \
      return fail(TRK_ENCCL, "%s -> %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error"); \
  } while (0)

}  // namespace

struct trk_comm {
  ncclComm_t comm;
  int rank, world;
  bool owned;      // created by trk_comm_init (destroyed with the handle) or attached (the caller's)
};

extern "C" {

int trk_comm_unique_id(void* id128) {
  TRK_REQUIRE(id128, "trk_comm_unique_id: NULL argument");
  if (int rc = need_rccl()) return rc;
  ncclUniqueId id;
  TRK_NCCL(g_rccl.GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return TRK_OK;
}

int trk_comm_init(const void* id128, int rank, int world, trk_comm** out) {
  TRK_REQUIRE(id128 && out && world >= 1 && rank >= 0 && rank < world, "trk_comm_init: bad argument");
  if (int rc = need_rccl()) return rc;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  TRK_NCCL(g_rccl.CommInitRank(&c, world, id, rank));      // one process per GPU: the communicator lives on the current device
  *out = new trk_comm{c, rank, world, true};
  return TRK_OK;
}

int trk_comm_attach(void* nccl_comm, int rank, int world, trk_comm** out) {
  TRK_REQUIRE(nccl_comm && out && world >= 1 && rank >= 0 && rank < world, "trk_comm_attach: bad argument");
  if (int rc = need_rccl()) return rc;
  *out = new trk_comm{(ncclComm_t)nccl_comm, rank, world, false};
  return TRK_OK;
}

int trk_comm_info(const trk_comm* c, int* rank, int* world) {
  TRK_REQUIRE(c && rank && world, "trk_comm_info: NULL argument");
  *rank = c->rank;
  *world = c->world;
  return TRK_OK;
}

int trk_comm_destroy(trk_comm* c) {
  if (!c) return TRK_OK;
  ncclResult_t r = ncclSuccess;
  if (c->owned && g_ok) r = g_rccl.CommDestroy(c->comm);
  delete c;                                   // the handle is gone either way; the result of the teardown is reported
  if (r != ncclSuccess)
    return fail(TRK_ENCCL, "ncclCommDestroy -> %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return TRK_OK;
}

int trk_allreduce_f64(trk_comm* c, double* dev, int count, trk_stream st) {
  TRK_REQUIRE(c && dev && count >= 0, "trk_allreduce_f64: bad argument");
  // one rank: the sum over ranks is the value itself — no RCCL call (trk.h).  TRK_COMM_FORCE=1 (diagnostics, tests, the bench's
  // latency probe) sends one-rank communicators through RCCL all the same: the only way a one-GPU box can exercise the call.
  static const bool force = getenv("TRK_COMM_FORCE") != nullptr;
  if (count == 0 || (c->world == 1 && !force)) return TRK_OK;
  TRK_NCCL(g_rccl.AllReduce(dev, dev, (size_t)count, ncclFloat64, ncclSum, c->comm, (hipStream_t)st));
  return TRK_OK;
}

int trk_halo_exchange(trk_comm* c, const float* send, int send_to, float* recv, int recv_from, int64_t count, trk_stream st) {
  TRK_REQUIRE(c && count >= 0, "trk_halo_exchange: bad argument");
  const bool do_send = send && send_to >= 0 && send_to < c->world, do_recv = recv && recv_from >= 0 && recv_from < c->world;
  if (count == 0 || (!do_send && !do_recv)) return TRK_OK;
  TRK_NCCL(g_rccl.GroupStart());
  // an error inside the group must not leave it open for every later collective of the process: the group is always ended, and
  // the FIRST failure is the one reported
  ncclResult_t first = ncclSuccess;
  const char* what = "";
  if (do_send) {
    first = g_rccl.Send(send, (size_t)count, ncclFloat32, send_to, c->comm, (hipStream_t)st);
    what = "ncclSend";
  }
  if (do_recv && first == ncclSuccess) {
    first = g_rccl.Recv(recv, (size_t)count, ncclFloat32, recv_from, c->comm, (hipStream_t)st);
    what = "ncclRecv";
  }
  const ncclResult_t end = g_rccl.GroupEnd();
  if (first == ncclSuccess && end != ncclSuccess) {
    first = end;
    what = "ncclGroupEnd";
  }
  if (first != ncclSuccess)
    return fail(TRK_ENCCL, "trk_halo_exchange: %s -> %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(first) : "RCCL error");
  return TRK_OK;
}

// Both neighbours in ONE group: what a fused space-time stencil needs of a time-sharded vector is its two boundary frames
// (operators.py:39-45 couples frame t with t+1 only) — one exchange per vector instead of one per direction of L.
int trk_halo_exchange2(trk_comm* c, const float* send_prev, float* recv_prev, const float* send_next, float* recv_next, int64_t count,
                       trk_stream st) {
  TRK_REQUIRE(c && count >= 0, "trk_halo_exchange2: bad argument");
  const bool hp = c->rank > 0, hn = c->rank < c->world - 1;
  TRK_REQUIRE((!hp || (send_prev && recv_prev)) && (!hn || (send_next && recv_next)), "trk_halo_exchange2: NULL buffer for an existing neighbour");
  if (count == 0 || (!hp && !hn)) return TRK_OK;
  TRK_NCCL(g_rccl.GroupStart());
  ncclResult_t first = ncclSuccess;
  const char* what = "";
  auto step = [&](ncclResult_t r, const char* w) {
    if (first == ncclSuccess && r != ncclSuccess) {
      first = r;
      what = w;
    }
  };
  if (hp) {
    step(g_rccl.Send(send_prev, (size_t)count, ncclFloat32, c->rank - 1, c->comm, (hipStream_t)st), "ncclSend");
    if (first == ncclSuccess) step(g_rccl.Recv(recv_prev, (size_t)count, ncclFloat32, c->rank - 1, c->comm, (hipStream_t)st), "ncclRecv");
  }
  if (hn && first == ncclSuccess) {
    step(g_rccl.Send(send_next, (size_t)count, ncclFloat32, c->rank + 1, c->comm, (hipStream_t)st), "ncclSend");
    if (first == ncclSuccess) step(g_rccl.Recv(recv_next, (size_t)count, ncclFloat32, c->rank + 1, c->comm, (hipStream_t)st), "ncclRecv");
  }
  step(g_rccl.GroupEnd(), "ncclGroupEnd");        // always ended: an open group would swallow every later collective of the process
  if (first != ncclSuccess)
    return fail(TRK_ENCCL, "trk_halo_exchange2: %s -> %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(first) : "RCCL error");
  return TRK_OK;
}

}  // extern "C"
