// RCCL entry points of the C ABI: what a
// reference-side binding needs to shard proposal batches over the GPUs of a
// node without going through torch.distributed -- the replicate / concatenate
// / add-the-counters pattern of nautilus/bounds/nautilus.py:223-237 as one
// all-gather of the accepted points and one all-reduce of the integer
// counters over xGMI.  RCCL is resolved at run time (the copy a host process
// has already loaded -- e.g. PyTorch's -- or /opt/rocm/lib/librccl.so), so
// the library has no link-time dependency on it.
#include "nb_common.h"
#include "../../include/nautilus_hip.h"

#include <dlfcn.h>

#include <cstring>

namespace {

// the slice of the NCCL API used here (rccl.h; ABI stable across RCCL 2.x)
typedef struct { char internal[128]; } rc_unique_id;
typedef void* rc_comm;
enum { RC_INT64 = 4, RC_FLOAT64 = 8 };      // ncclInt64, ncclFloat64
enum { RC_SUM = 0 };                        // ncclSum

struct RcclApi {
  int (*get_unique_id)(rc_unique_id*);
  int (*comm_init_rank)(rc_comm*, int, rc_unique_id, int);
  int (*comm_destroy)(rc_comm);
  int (*all_gather)(const void*, void*, size_t, int, rc_comm, hipStream_t);
  int (*all_reduce)(const void*, void*, size_t, int, int, rc_comm,
                    hipStream_t);
  const char* (*get_error_string)(int);
  bool ok = false;
};

RcclApi& rccl() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api;
  tried = true;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (h == nullptr) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (h == nullptr) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (h == nullptr) return api;
  api.get_unique_id = (int (*)(rc_unique_id*))dlsym(h, "ncclGetUniqueId");
  api.comm_init_rank =
      (int (*)(rc_comm*, int, rc_unique_id, int))dlsym(h, "ncclCommInitRank");
  api.comm_destroy = (int (*)(rc_comm))dlsym(h, "ncclCommDestroy");
  api.all_gather = (int (*)(const void*, void*, size_t, int, rc_comm,
                            hipStream_t))dlsym(h, "ncclAllGather");
  api.all_reduce = (int (*)(const void*, void*, size_t, int, int, rc_comm,
                            hipStream_t))dlsym(h, "ncclAllReduce");
  api.get_error_string = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  api.ok = api.get_unique_id && api.comm_init_rank && api.comm_destroy &&
           api.all_gather && api.all_reduce;
  return api;
}

int rc_check(int status, const char* what) {
  if (status == 0) return NB_OK;
  RcclApi& api = rccl();
  nb_set_error("%s failed: %s", what,
               api.get_error_string ? api.get_error_string(status) : "RCCL error");
  return NB_ERR_HIP;
}

}  // namespace

struct nb_comm {
  rc_comm comm = nullptr;
  int rank = 0, n_ranks = 1;
};

extern "C" {

int nb_comm_unique_id(uint8_t* id_out) {
  RcclApi& api = rccl();
  if (!api.ok || id_out == nullptr) {
    nb_set_error("RCCL is not available (librccl.so)");
    return NB_ERR_UNSUPPORTED;
  }
  rc_unique_id id;
  const int rc = rc_check(api.get_unique_id(&id), "ncclGetUniqueId");
  if (rc == NB_OK) std::memcpy(id_out, id.internal, NB_COMM_ID_BYTES);
  return rc;
}

int nb_comm_init(int32_t rank, int32_t n_ranks, const uint8_t* id,
                 nb_comm** out) {
  RcclApi& api = rccl();
  if (!api.ok) {
    nb_set_error("RCCL is not available (librccl.so)");
    return NB_ERR_UNSUPPORTED;
  }
  if (id == nullptr || out == nullptr || n_ranks < 1 || rank < 0 ||
      rank >= n_ranks) {
    nb_set_error("bad communicator arguments (rank %d of %d)", rank, n_ranks);
    return NB_ERR_ARG;
  }
  rc_unique_id uid;
  std::memcpy(uid.internal, id, NB_COMM_ID_BYTES);
  nb_comm* c = new nb_comm();
  c->rank = rank;
  c->n_ranks = n_ranks;
  const int rc = rc_check(api.comm_init_rank(&c->comm, n_ranks, uid, rank),
                          "ncclCommInitRank");
  if (rc != NB_OK) { delete c; return rc; }
  *out = c;
  return NB_OK;
}

int nb_comm_destroy(nb_comm* c) {
  if (c == nullptr) return NB_OK;
  if (c->comm != nullptr) (void)rccl().comm_destroy(c->comm);
  delete c;
  return NB_OK;
}

uint64_t nb_comm_rank_key(uint64_t seed, int32_t rank) {
  // Philox key of `rank` (rank 0 keeps the single-GPU stream); the same
  // mixing as parallel.rank_key
  const uint64_t mix = 0x9E3779B97F4A7C15ull;
  const uint64_t lim = 0x7FFFFFFFFFFFFFFFull;
  return (seed ^ (((uint64_t)rank * mix) & lim)) & lim;
}

int nb_comm_allgather_f64(nb_comm* c, const double* send, int64_t count,
                          double* recv, void* stream) {
  if (c == nullptr || send == nullptr || recv == nullptr || count < 0) {
    nb_set_error("bad all-gather arguments");
    return NB_ERR_ARG;
  }
  return rc_check(rccl().all_gather(send, recv, (size_t)count, RC_FLOAT64,
                                    c->comm, (hipStream_t)stream),
                  "ncclAllGather");
}

int nb_comm_allreduce_i64(nb_comm* c, int64_t* buf, int64_t count,
                          void* stream) {
  if (c == nullptr || buf == nullptr || count < 0) {
    nb_set_error("bad all-reduce arguments");
    return NB_ERR_ARG;
  }
  return rc_check(rccl().all_reduce(buf, buf, (size_t)count, RC_INT64, RC_SUM,
                                    c->comm, (hipStream_t)stream),
                  "ncclAllReduce");
}

}  // extern "C"
