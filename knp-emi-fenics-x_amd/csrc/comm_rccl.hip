// RCCL point-to-point halo transport and all-reduce, called directly from the library (no Python in the exchange).
//
// The reference moves ghost data with MPI inside DOLFINx / PETSc (`Function.x.scatter_forward()`, parallel KSP:
// pdeSolver.py:24-35 run on the mesh communicator).  Here one process drives one GPU and the ghost dofs travel GPU to
// GPU over xGMI: pack kernel -> ncclSend / ncclRecv pairs in one group -> unpack kernel, all on the handle's stream, no
// host synchronisation.  The communicator is created from a unique id that the caller distributes over whatever
// rendezvous it has (knpemi/fem/distributed.py: torch.distributed's store).
//
// RCCL is resolved at run time: the symbols already loaded in the process are used if there are any (a PyTorch
// process has its own librccl loaded; a second copy must not be mixed in), otherwise /opt/rocm/lib/librccl.so.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "knpemi_internal.h"

namespace {

struct Rccl {
  bool ok = false;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

template <class F>
bool resolve(void* lib, const char* name, F& f) {
  void* p = dlsym(lib ? lib : RTLD_DEFAULT, name);
  f = reinterpret_cast<F>(p);
  return p != nullptr;
}

Rccl& rccl() {
  static Rccl r = [] {
    Rccl q;
    void* lib = nullptr;
    if (!dlsym(RTLD_DEFAULT, "ncclGetUniqueId")) {
      lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) return q;
    }
    q.ok = resolve(lib, "ncclGetUniqueId", q.GetUniqueId) && resolve(lib, "ncclCommInitRank", q.CommInitRank) &&
           resolve(lib, "ncclCommDestroy", q.CommDestroy) && resolve(lib, "ncclSend", q.Send) &&
           resolve(lib, "ncclRecv", q.Recv) && resolve(lib, "ncclAllReduce", q.AllReduce) &&
           resolve(lib, "ncclGroupStart", q.GroupStart) && resolve(lib, "ncclGroupEnd", q.GroupEnd) &&
           resolve(lib, "ncclGetErrorString", q.GetErrorString);
    return q;
  }();
  return r;
}

int comm_fail(const char* what, ncclResult_t e) {
  kn_set_error(std::string(what) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(e) : "RCCL error"));
  return KNPEMI_EHIP;
}

#define KN_NCCL(call)                                  \
  do {                                                 \
    ncclResult_t e_ = (call);                          \
    if (e_ != ncclSuccess) return comm_fail(#call, e_); \
  } while (0)

}  // namespace

extern "C" int knpemi_comm_unique_id(char* out, size_t len) {
  if (!out || len < sizeof(ncclUniqueId)) {
    kn_set_error("knpemi_comm_unique_id: the buffer needs 128 bytes");
    return KNPEMI_EINVAL;
  }
  if (!rccl().ok) {
    kn_set_error("knpemi_comm_unique_id: RCCL is not available in this process");
    return KNPEMI_EHIP;
  }
  ncclUniqueId id;
  KN_NCCL(rccl().GetUniqueId(&id));
  std::memcpy(out, &id, sizeof(id));
  return KNPEMI_OK;
}

int kn_comm_create(int device, int rank, int world, const char* id_bytes, size_t len, void** out);

extern "C" int knpemi_comm_init(knpemi_handle* h, int rank, int world, const char* id_bytes, size_t len) {
  if (!h) {
    kn_set_error("knpemi_comm_init: null handle");
    return KNPEMI_EINVAL;
  }
  if (h->comm) {
    kn_set_error("knpemi_comm_init: communicator already created");
    return KNPEMI_EINVAL;
  }
  int rc = kn_comm_create(h->device, rank, world, id_bytes, len, &h->comm);
  if (rc) return rc;
  h->comm_rank = rank;
  h->comm_world = world;
  return KNPEMI_OK;
}

void kn_comm_destroy(knpemi_handle* h) {
  if (h->comm && rccl().ok) (void)rccl().CommDestroy(static_cast<ncclComm_t>(h->comm));
  h->comm = nullptr;
}

// send_buf[send_off[p] .. + send_cnt[p]) -> peer[p], recv_buf[recv_off[p] .. + recv_cnt[p]) <- peer[p], one group
int kn_comm_sendrecv(void* comm, int world, int device, hipStream_t stream, const double* send_buf_dev, double* recv_buf_dev,
                     int n_parts, const int32_t* peer, const int64_t* send_off, const int64_t* send_cnt,
                     const int64_t* recv_off, const int64_t* recv_cnt) {
  if (!comm) {
    kn_set_error("knpemi_comm_sendrecv: no communicator (knpemi_comm_init)");
    return KNPEMI_EINVAL;
  }
  if (n_parts < 0 || (n_parts > 0 && (!peer || !send_off || !send_cnt || !recv_off || !recv_cnt))) {
    kn_set_error("knpemi_comm_sendrecv: bad argument");
    return KNPEMI_EINVAL;
  }
  for (int p = 0; p < n_parts; ++p)
    if (peer[p] < 0 || peer[p] >= world || send_cnt[p] < 0 || recv_cnt[p] < 0 ||
        (send_cnt[p] > 0 && !send_buf_dev) || (recv_cnt[p] > 0 && !recv_buf_dev)) {
      kn_set_error("knpemi_comm_sendrecv: bad part");
      return KNPEMI_EINVAL;
    }
  KN_HIP(hipSetDevice(device));
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  KN_NCCL(rccl().GroupStart());
  ncclResult_t first = ncclSuccess;   // an error inside the group must not leave it open
  for (int p = 0; p < n_parts && first == ncclSuccess; ++p) {
    if (send_cnt[p] > 0) first = rccl().Send(send_buf_dev + send_off[p], (size_t)send_cnt[p], ncclDouble, peer[p], c, stream);
    if (first == ncclSuccess && recv_cnt[p] > 0)
      first = rccl().Recv(recv_buf_dev + recv_off[p], (size_t)recv_cnt[p], ncclDouble, peer[p], c, stream);
  }
  const ncclResult_t end = rccl().GroupEnd();
  if (first != ncclSuccess) return comm_fail("ncclSend / ncclRecv", first);
  if (end != ncclSuccess) return comm_fail("ncclGroupEnd", end);
  return KNPEMI_OK;
}

int kn_comm_create(int device, int rank, int world, const char* id_bytes, size_t len, void** out) {
  if (!id_bytes || len < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world || !out) {
    kn_set_error("knpemi_comm_init: bad argument");
    return KNPEMI_EINVAL;
  }
  if (!rccl().ok) {
    kn_set_error("knpemi_comm_init: RCCL is not available in this process");
    return KNPEMI_EHIP;
  }
  KN_HIP(hipSetDevice(device));
  ncclUniqueId id;
  std::memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t c = nullptr;
  KN_NCCL(rccl().CommInitRank(&c, world, id, rank));
  *out = c;
  return KNPEMI_OK;
}

void kn_comm_free(void* comm) {
  if (comm && rccl().ok) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm));
}

extern "C" int knpemi_comm_sendrecv(knpemi_handle* h, const double* send_buf_dev, double* recv_buf_dev, int n_parts,
                                    const int32_t* peer, const int64_t* send_off, const int64_t* send_cnt,
                                    const int64_t* recv_off, const int64_t* recv_cnt) {
  if (!h) {
    kn_set_error("knpemi_comm_sendrecv: null handle");
    return KNPEMI_EINVAL;
  }
  return kn_comm_sendrecv(h->comm, h->comm_world, h->device, h->stream, send_buf_dev, recv_buf_dev, n_parts, peer, send_off,
                          send_cnt, recv_off, recv_cnt);
}

extern "C" int knpemi_comm_allreduce(knpemi_handle* h, double* buf_dev, int n) {
  if (!h || !h->comm || !buf_dev || n < 1) {
    kn_set_error("knpemi_comm_allreduce: bad argument or no communicator");
    return KNPEMI_EINVAL;
  }
  KN_HIP(hipSetDevice(h->device));
  KN_NCCL(rccl().AllReduce(buf_dev, buf_dev, (size_t)n, ncclDouble, ncclSum, static_cast<ncclComm_t>(h->comm), h->stream));
  return KNPEMI_OK;
}

// ---- the Krylov solves' communication hooks without Python (knpemi_set_distributed) ---------------------------
// Registered once per system: which entries of a solver vector go to / come from which neighbour.
extern "C" int knpemi_comm_set_vector_plan(knpemi_handle* h, int which, const int32_t* send_idx_dev, int n_send,
                                           const int32_t* recv_idx_dev, int n_recv, double* send_buf_dev,
                                           double* recv_buf_dev, int n_parts, const int32_t* peer, const int64_t* send_off,
                                           const int64_t* send_cnt, const int64_t* recv_off, const int64_t* recv_cnt) {
  if (!h || (which != KNPEMI_B_EMI && which != KNPEMI_B_KNP) || n_parts < 0 || n_send < 0 || n_recv < 0) {
    kn_set_error("knpemi_comm_set_vector_plan: bad argument");
    return KNPEMI_EINVAL;
  }
  KnVecPlan& p = h->vec_plan[which];
  p.send_idx = send_idx_dev; p.recv_idx = recv_idx_dev; p.n_send = n_send; p.n_recv = n_recv;
  p.send_buf = send_buf_dev; p.recv_buf = recv_buf_dev;
  p.peer.assign(peer, peer + n_parts);
  p.send_off.assign(send_off, send_off + n_parts); p.send_cnt.assign(send_cnt, send_cnt + n_parts);
  p.recv_off.assign(recv_off, recv_off + n_parts); p.recv_cnt.assign(recv_cnt, recv_cnt + n_parts);
  p.set = true;
  return KNPEMI_OK;
}

// knpemi_allreduce_fn with ctx = the handle: sums the first n doubles of the registered reduction buffer
extern "C" int knpemi_comm_allreduce_hook(void* ctx, int n) {
  knpemi_handle* h = static_cast<knpemi_handle*>(ctx);
  if (!h || !h->dist.d_red) return KNPEMI_EINVAL;
  return knpemi_comm_allreduce(h, h->dist.d_red, n);
}

// knpemi_halo_fn with ctx = the handle: ghost refresh of a solver vector through the registered plan
extern "C" int knpemi_comm_halo_hook(void* ctx, void* vec_dev, int which) {
  knpemi_handle* h = static_cast<knpemi_handle*>(ctx);
  if (!h || (which != KNPEMI_B_EMI && which != KNPEMI_B_KNP) || !h->vec_plan[which].set) return KNPEMI_EINVAL;
  const KnVecPlan& p = h->vec_plan[which];
  int rc;
  if ((rc = knpemi_vec_gather(h, vec_dev, p.send_idx, p.n_send, p.send_buf))) return rc;
  if ((rc = knpemi_comm_sendrecv(h, p.send_buf, p.recv_buf, (int)p.peer.size(), p.peer.data(), p.send_off.data(),
                                 p.send_cnt.data(), p.recv_off.data(), p.recv_cnt.data()))) return rc;
  return knpemi_vec_scatter(h, vec_dev, p.recv_idx, p.n_recv, p.recv_buf);
}
