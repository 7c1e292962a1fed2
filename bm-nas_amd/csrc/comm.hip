// RCCL behind the C ABI (bmnas_comm_* / bmnas_allreduce_f32) — the one
// exchange step of the data-parallel search loop: an in-place all-reduce of the flat fp32 gradient
// bucket over xGMI, replacing the scatter / gather / replicate of torch.nn.DataParallel
// (mmimdb_darts_searchable.py:36-37, ntu_darts_searchable.py:50-52, ego_darts_searchable.py:51-53).
//
// librccl is bound lazily with dlopen (RTLD_NOLOAD first, so a process that already carries
// PyTorch's copy shares it): loading libbmnas_hip.so never needs RCCL, single-GPU runs never touch
// it.  The collective is enqueued on the caller's stream like every other entry point, so it can be
// captured into the hipGraph of a training step (fwd + bwd + all-reduce + Adam in one replay).
#include "../../include/bmnas_hip.h"
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <string.h>

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
constexpr int kNcclFloat32 = 7, kNcclSum = 0, kNcclAvg = 4;

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*CommCount)(NcclComm, int*) = nullptr;
  int (*CommUserRank)(NcclComm, int*) = nullptr;
  int (*CommCuDevice)(NcclComm, int*) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // already in the process (PyTorch's)?
      if (r.handle) break;
    }
    if (!r.handle) {
      for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
      }
    }
    if (!r.handle) return;
    r.GetUniqueId = (int (*)(NcclUniqueId*))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (int (*)(NcclComm))dlsym(r.handle, "ncclCommDestroy");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))dlsym(r.handle, "ncclAllReduce");
    r.GetVersion = (int (*)(int*))dlsym(r.handle, "ncclGetVersion");
    r.CommCount = (int (*)(NcclComm, int*))dlsym(r.handle, "ncclCommCount");
    r.CommUserRank = (int (*)(NcclComm, int*))dlsym(r.handle, "ncclCommUserRank");
    r.CommCuDevice = (int (*)(NcclComm, int*))dlsym(r.handle, "ncclCommCuDevice");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
  });
  return r;
}

constexpr int kErrNoRccl = -4;          // librccl could not be bound
inline int nccl_rc(int rc) { return rc == 0 ? 0 : 1000 + rc; }   // > 0: RCCL's ncclResult_t + 1000

}  // namespace

extern "C" int bmnas_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int bmnas_comm_unique_id_bytes(void) { return (int)sizeof(NcclUniqueId); }

extern "C" int bmnas_comm_get_unique_id(void* id_out) {
  if (!id_out) return BMNAS_E_ARG;
  Rccl& r = rccl();
  if (!r.ok) return kErrNoRccl;
  NcclUniqueId id;
  const int rc = r.GetUniqueId(&id);
  if (rc == 0) memcpy(id_out, &id, sizeof(id));
  return nccl_rc(rc);
}

extern "C" int bmnas_comm_init_rank(void** comm_out, int world, int rank, const void* id) {
  if (!comm_out || !id || world < 1 || rank < 0 || rank >= world) return BMNAS_E_ARG;
  Rccl& r = rccl();
  if (!r.ok) return kErrNoRccl;
  NcclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  NcclComm c = nullptr;
  const int rc = r.CommInitRank(&c, world, uid, rank);
  if (rc == 0) *comm_out = c;
  return nccl_rc(rc);
}

extern "C" int bmnas_comm_destroy(void* comm) {
  if (!comm) return BMNAS_E_ARG;
  Rccl& r = rccl();
  if (!r.ok) return kErrNoRccl;
  return nccl_rc(r.CommDestroy((NcclComm)comm));
}

// What the communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice / ncclGetVersion):
// evidence, printed in bench.py's N > 1 line, that RCCL really spans the ranks the launcher started.
extern "C" int bmnas_comm_info(void* comm, int* n_ranks, int* user_rank, int* hip_device, int* rccl_version) {
  if (!comm) return BMNAS_E_ARG;
  Rccl& r = rccl();
  if (!r.ok) return kErrNoRccl;
  int v = -1, rc = 0;
  if (n_ranks) { v = -1; if (r.CommCount) rc = r.CommCount((NcclComm)comm, &v); if (rc) return nccl_rc(rc); *n_ranks = v; }
  if (user_rank) { v = -1; if (r.CommUserRank) rc = r.CommUserRank((NcclComm)comm, &v); if (rc) return nccl_rc(rc); *user_rank = v; }
  if (hip_device) { v = -1; if (r.CommCuDevice) rc = r.CommCuDevice((NcclComm)comm, &v); if (rc) return nccl_rc(rc); *hip_device = v; }
  if (rccl_version) { v = -1; if (r.GetVersion) rc = r.GetVersion(&v); if (rc) return nccl_rc(rc); *rccl_version = v; }
  return 0;
}

extern "C" int bmnas_allreduce_f32(float* buf, int64_t count, int average, void* comm, void* stream) {
  if (!buf || count < 0 || !comm) return BMNAS_E_ARG;
  if (count == 0) return 0;
  Rccl& r = rccl();
  if (!r.ok) return kErrNoRccl;
  return nccl_rc(r.AllReduce(buf, buf, (size_t)count, kNcclFloat32, average ? kNcclAvg : kNcclSum, (NcclComm)comm,
                             (hipStream_t)stream));
}
