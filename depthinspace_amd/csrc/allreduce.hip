// dis_allreduce_*: the gradient exchange of the data-parallel step behind the C ABI (SURVEY.md section 8(b) lists it; 8(e): one
// all-reduce of the flat fp32 gradient per step, mean of the per-rank gradients).  The reference has no communication layer at all
// (train_val.py:55-56: one GPU); a host that is not PyTorch gets the exchange from here, the Python trainer keeps torch.distributed
// (backend 'nccl' = RCCL) unless DIS_ALLREDUCE=abi (trainer.FlatAdam).
//
// RCCL is bound at the first call, not at load time: libdis_hip.so must load on a box without librccl on the loader path (every
// other entry point is independent of it), and a process that already carries torch's copy must end up with ONE RCCL (dlopen by
// soname returns the loaded one).  No device memory is allocated here; the communicator is the only state, owned by the caller.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/dis_hip.h"

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.10: ncclUniqueId is 128 opaque bytes passed by value)
struct RcclUniqueId {
  char internal[128];
};
typedef void* RcclComm;
typedef int (*GetUniqueIdFn)(RcclUniqueId*);
typedef int (*CommInitRankFn)(RcclComm*, int, RcclUniqueId, int);
typedef int (*CommDestroyFn)(RcclComm);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, RcclComm, hipStream_t);
constexpr int kRcclFloat32 = 7, kRcclSum = 0, kRcclAvg = 4;

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  bool tried = false;
};

Rccl& rccl() {
  static Rccl r;
  if (r.tried) return r;
  r.tried = true;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) {   // a copy the process already holds (torch's) first
    r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (r.handle) break;
  }
  for (int i = 0; i < 2 && !r.handle; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!r.handle) return r;
  r.get_unique_id = (GetUniqueIdFn)dlsym(r.handle, "ncclGetUniqueId");
  r.comm_init_rank = (CommInitRankFn)dlsym(r.handle, "ncclCommInitRank");
  r.comm_destroy = (CommDestroyFn)dlsym(r.handle, "ncclCommDestroy");
  r.all_reduce = (AllReduceFn)dlsym(r.handle, "ncclAllReduce");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) r.handle = nullptr;
  return r;
}

constexpr uint64_t kMagic = 0x4449534152434c31ull;   // "DISARCL1"
struct DisComm {
  uint64_t magic;
  RcclComm comm;
  int nranks, rank;
};

// ncclResult_t -> this ABI: 0 stays 0, everything else is reported as a positive code in a range hipError_t does not use
inline int rc_of(int nccl_rc) { return nccl_rc == 0 ? DIS_OK : 10000 + nccl_rc; }

}  // namespace

extern "C" int dis_allreduce_unique_id(void* id128) {
  if (!id128) return DIS_ERR_NULL;
  Rccl& r = rccl();
  if (!r.handle) return DIS_ERR_UNSUPPORTED;
  RcclUniqueId id;
  const int rc = r.get_unique_id(&id);
  if (rc == 0) std::memcpy(id128, id.internal, sizeof id.internal);
  return rc_of(rc);
}

extern "C" int dis_allreduce_init(void** comm, const void* id128, int nranks, int rank) {
  if (!comm || !id128) return DIS_ERR_NULL;
  if (nranks < 1 || rank < 0 || rank >= nranks) return DIS_ERR_BAD_SHAPE;
  Rccl& r = rccl();
  if (!r.handle) return DIS_ERR_UNSUPPORTED;
  RcclUniqueId id;
  std::memcpy(id.internal, id128, sizeof id.internal);
  RcclComm c = nullptr;
  const int rc = r.comm_init_rank(&c, nranks, id, rank);
  if (rc != 0) return rc_of(rc);
  DisComm* d = new DisComm{kMagic, c, nranks, rank};
  *comm = d;
  return DIS_OK;
}

extern "C" int dis_allreduce_sum_f32(void* comm, float* buf, long count, int average, void* stream) {
  if (!comm || !buf) return DIS_ERR_NULL;
  if (count < 0) return DIS_ERR_BAD_SHAPE;
  if (count == 0) return DIS_OK;
  DisComm* d = (DisComm*)comm;
  if (d->magic != kMagic) return DIS_ERR_BAD_SHAPE;
  Rccl& r = rccl();
  if (!r.handle) return DIS_ERR_UNSUPPORTED;
  return rc_of(r.all_reduce(buf, buf, (size_t)count, kRcclFloat32, average ? kRcclAvg : kRcclSum, d->comm, (hipStream_t)stream));
}

extern "C" int dis_allreduce_destroy(void* comm) {
  if (!comm) return DIS_ERR_NULL;
  DisComm* d = (DisComm*)comm;
  if (d->magic != kMagic) return DIS_ERR_BAD_SHAPE;
  Rccl& r = rccl();
  if (!r.handle) return DIS_ERR_UNSUPPORTED;
  const int rc = r.comm_destroy(d->comm);
  d->magic = 0;
  delete d;
  return rc_of(rc);
}
