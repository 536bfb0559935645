// libgfft.so: the global-redistribution wire of the C ABI (include/gfft.h, "exchange" section).
//
// What the reference gets from MPI_Alltoallw on a Cartesian sub-communicator (mpi4py_fft/
// pencil.py:182-183,200-201 on communicators from pencil.py:64-93) is here a grouped batch of
// RCCL point-to-point messages over xGMI, enqueued on a HIP stream the caller owns.  RCCL is bound
// at run time (dlopen), so libgfft.so keeps libamdhip64 as its only link-time dependency and a
// process that already carries a librccl (PyTorch bundles its own) shares that copy.
#include "../../include/gfft.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace {

// the slice of rccl.h this file uses (stable since NCCL 2.18: ncclCommSplit is the newest entry)
typedef struct { char internal[128]; } rcclUniqueId;
typedef void *rcclComm_t;
typedef int rcclResult_t;
enum { rcclInt8 = 0 };

struct Rccl {
  void *handle = nullptr;
  std::string path;
  rcclResult_t (*GetUniqueId)(rcclUniqueId *) = nullptr;
  rcclResult_t (*CommInitRank)(rcclComm_t *, int, rcclUniqueId, int) = nullptr;
  rcclResult_t (*CommSplit)(rcclComm_t, int, int, rcclComm_t *, void *) = nullptr;
  rcclResult_t (*CommDestroy)(rcclComm_t) = nullptr;
  rcclResult_t (*CommCount)(rcclComm_t, int *) = nullptr;
  rcclResult_t (*CommUserRank)(rcclComm_t, int *) = nullptr;
  rcclResult_t (*Send)(const void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  rcclResult_t (*Recv)(void *, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  rcclResult_t (*GroupStart)() = nullptr;
  rcclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(rcclResult_t) = nullptr;
  rcclResult_t (*GetVersion)(int *) = nullptr;
};

std::mutex g_mutex;
Rccl g_rccl;
thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

int bind(const char *path) {
  // Already-loaded copies are found by soname first (RTLD_NOLOAD), so a host that linked or
  // imported a librccl shares it; otherwise the usual search path and the ROCm install.
  std::vector<std::string> cands;
  if (path && *path) cands.push_back(path);
  else {
    if (const char *e = getenv("GFFT_RCCL_LIB")) cands.push_back(e);
    cands.push_back("librccl.so.1");
    cands.push_back("librccl.so");
    cands.push_back("/opt/rocm/lib/librccl.so.1");
    cands.push_back("/opt/rocm/lib/librccl.so");
  }
  void *h = nullptr;
  std::string used, tried;
  for (const std::string &c : cands) {
    h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (h) { used = c; break; }
    tried += c + " ";
  }
  if (!h) return fail(GFFT_ERR_UNSUPPORTED, "no RCCL library could be loaded (tried: " + tried + ")");
  Rccl r;
  r.handle = h;
  r.path = used;
  struct { const char *name; void **slot; } syms[] = {
      {"ncclGetUniqueId", (void **)&r.GetUniqueId},   {"ncclCommInitRank", (void **)&r.CommInitRank},
      {"ncclCommSplit", (void **)&r.CommSplit},       {"ncclCommDestroy", (void **)&r.CommDestroy},
      {"ncclCommCount", (void **)&r.CommCount},       {"ncclCommUserRank", (void **)&r.CommUserRank},
      {"ncclSend", (void **)&r.Send},                 {"ncclRecv", (void **)&r.Recv},
      {"ncclGroupStart", (void **)&r.GroupStart},     {"ncclGroupEnd", (void **)&r.GroupEnd},
      {"ncclGetErrorString", (void **)&r.GetErrorString}, {"ncclGetVersion", (void **)&r.GetVersion}};
  for (auto &s : syms) {
    *s.slot = dlsym(h, s.name);
    if (!*s.slot) return fail(GFFT_ERR_UNSUPPORTED, std::string(used) + " lacks " + s.name);
  }
  g_rccl = r;
  return GFFT_OK;
}

int rccl(Rccl **out) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!g_rccl.handle) {
    int rc = bind(nullptr);
    if (rc) return rc;
  }
  *out = &g_rccl;
  return GFFT_OK;
}

#define RCCL_TRY(R, expr)                                                                          \
  do {                                                                                             \
    rcclResult_t _r = (expr);                                                                      \
    if (_r != 0) return fail(GFFT_ERR_HIP, std::string(#expr) + ": " + (R)->GetErrorString(_r));   \
  } while (0)

#define HIPX_TRY(expr)                                                                             \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(GFFT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

}  // namespace

struct gfft_comm_s {
  rcclComm_t comm = nullptr;
  int rank = 0, size = 1;
};

extern "C" {

const char *gfft_exchange_last_error(void) { return g_err.c_str(); }

int gfft_rccl_load(const char *path) {
  std::lock_guard<std::mutex> lock(g_mutex);
  return bind(path);
}

int gfft_rccl_info(char *buf, size_t len) {
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  int v = 0;
  R->GetVersion(&v);
  snprintf(buf, len, "%s (version code %d)", R->path.c_str(), v);
  return GFFT_OK;
}

int gfft_comm_get_unique_id(void *id128) {
  if (!id128) return fail(GFFT_ERR_INVALID, "null id");
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  rcclUniqueId id;
  RCCL_TRY(R, R->GetUniqueId(&id));
  memcpy(id128, id.internal, sizeof id.internal);
  return GFFT_OK;
}

int gfft_comm_create(gfft_comm *comm, const void *id128, int nranks, int rank) {
  if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(GFFT_ERR_INVALID, "bad communicator arguments");
  *comm = nullptr;
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  rcclUniqueId id;
  memcpy(id.internal, id128, sizeof id.internal);
  gfft_comm_s *c = new gfft_comm_s;
  rcclResult_t r = R->CommInitRank(&c->comm, nranks, id, rank);
  if (r != 0) {
    delete c;
    return fail(GFFT_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(r));
  }
  c->rank = rank;
  c->size = nranks;
  *comm = c;
  return GFFT_OK;
}

int gfft_comm_split(gfft_comm parent, int color, int key, gfft_comm *sub) {
  if (!parent || !sub) return fail(GFFT_ERR_INVALID, "null communicator");
  *sub = nullptr;
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  rcclComm_t nc = nullptr;
  RCCL_TRY(R, R->CommSplit(parent->comm, color < 0 ? -1 : color, key, &nc, nullptr));
  if (color < 0 || !nc) return GFFT_OK;     // NCCL_SPLIT_NOCOLOR: this rank joins no group
  gfft_comm_s *c = new gfft_comm_s;
  c->comm = nc;
  RCCL_TRY(R, R->CommCount(nc, &c->size));
  RCCL_TRY(R, R->CommUserRank(nc, &c->rank));
  *sub = c;
  return GFFT_OK;
}

int gfft_comm_rank(gfft_comm comm, int *rank, int *size) {
  if (!comm) return fail(GFFT_ERR_INVALID, "null communicator");
  if (rank) *rank = comm->rank;
  if (size) *size = comm->size;
  return GFFT_OK;
}

int gfft_comm_destroy(gfft_comm comm) {
  if (!comm) return GFFT_OK;
  Rccl *R;
  if (rccl(&R) == GFFT_OK && comm->comm) R->CommDestroy(comm->comm);
  delete comm;
  return GFFT_OK;
}

int gfft_sendrecv(gfft_comm comm, int nsend, const gfft_msg *sends, int nrecv, const gfft_msg *recvs, void *stream) {
  if (!comm || nsend < 0 || nrecv < 0 || (nsend && !sends) || (nrecv && !recvs)) return fail(GFFT_ERR_INVALID, "bad message list");
  Rccl *R;
  int rc = rccl(&R);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < nsend; ++i)
    if (sends[i].peer < 0 || sends[i].peer >= comm->size || sends[i].bytes < 0) return fail(GFFT_ERR_INVALID, "bad send entry");
  for (int i = 0; i < nrecv; ++i)
    if (recvs[i].peer < 0 || recvs[i].peer >= comm->size || recvs[i].bytes < 0) return fail(GFFT_ERR_INVALID, "bad receive entry");
  // Messages to oneself never touch the wire: the k-th self send is copied into the k-th self
  // receive (MPI's matching order), stream ordered like everything else.
  int si = 0;
  for (int ri = 0; ri < nrecv; ++ri) {
    if (recvs[ri].peer != comm->rank) continue;
    while (si < nsend && sends[si].peer != comm->rank) ++si;
    if (si == nsend || sends[si].bytes != recvs[ri].bytes) return fail(GFFT_ERR_INVALID, "self send / receive lists do not match");
    if (recvs[ri].bytes && recvs[ri].ptr != sends[si].ptr)
      HIPX_TRY(hipMemcpyAsync(recvs[ri].ptr, sends[si].ptr, (size_t)recvs[ri].bytes, hipMemcpyDeviceToDevice, s));
    ++si;
  }
  bool wire = false;
  for (int i = 0; i < nsend && !wire; ++i) wire = sends[i].peer != comm->rank && sends[i].bytes > 0;
  for (int i = 0; i < nrecv && !wire; ++i) wire = recvs[i].peer != comm->rank && recvs[i].bytes > 0;
  if (!wire) return GFFT_OK;
  RCCL_TRY(R, R->GroupStart());
  rcclResult_t bad = 0;
  for (int i = 0; i < nrecv && !bad; ++i)
    if (recvs[i].peer != comm->rank && recvs[i].bytes > 0)
      bad = R->Recv(recvs[i].ptr, (size_t)recvs[i].bytes, rcclInt8, recvs[i].peer, comm->comm, s);
  for (int i = 0; i < nsend && !bad; ++i)
    if (sends[i].peer != comm->rank && sends[i].bytes > 0)
      bad = R->Send(sends[i].ptr, (size_t)sends[i].bytes, rcclInt8, sends[i].peer, comm->comm, s);
  rcclResult_t end = R->GroupEnd();
  if (bad) return fail(GFFT_ERR_HIP, std::string("ncclSend/ncclRecv: ") + R->GetErrorString(bad));
  if (end) return fail(GFFT_ERR_HIP, std::string("ncclGroupEnd: ") + R->GetErrorString(end));
  return GFFT_OK;
}

int gfft_alltoallv(gfft_comm comm, const void *d_send, const int64_t *send_counts, const int64_t *send_displs,
                   void *d_recv, const int64_t *recv_counts, const int64_t *recv_displs, int itemsize, void *stream) {
  if (!comm || !send_counts || !send_displs || !recv_counts || !recv_displs || itemsize < 1)
    return fail(GFFT_ERR_INVALID, "bad all-to-all arguments");
  std::vector<gfft_msg> sends(comm->size), recvs(comm->size);
  for (int i = 0; i < comm->size; ++i) {
    sends[i].ptr = const_cast<char *>(static_cast<const char *>(d_send)) + send_displs[i] * itemsize;
    sends[i].bytes = send_counts[i] * itemsize;
    sends[i].peer = i;
    recvs[i].ptr = static_cast<char *>(d_recv) + recv_displs[i] * itemsize;
    recvs[i].bytes = recv_counts[i] * itemsize;
    recvs[i].peer = i;
  }
  return gfft_sendrecv(comm, comm->size, sends.data(), comm->size, recvs.data(), stream);
}

/* ---- streams and events a host without a HIP binding needs to own its pipeline ---- */
int gfft_stream_create(void **stream) {
  if (!stream) return fail(GFFT_ERR_INVALID, "null argument");
  hipStream_t s;
  HIPX_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = s;
  return GFFT_OK;
}
int gfft_stream_destroy(void *stream) { HIPX_TRY(hipStreamDestroy((hipStream_t)stream)); return GFFT_OK; }
int gfft_stream_wait_event(void *stream, void *event) {
  HIPX_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return GFFT_OK;
}
int gfft_event_create_untimed(void **event) {
  if (!event) return fail(GFFT_ERR_INVALID, "null argument");
  hipEvent_t e;
  HIPX_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  *event = e;
  return GFFT_OK;
}

}  // extern "C"
