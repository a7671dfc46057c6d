// transport_rccl.hip — RCCL, bound at run time: the process must use ONE HIP runtime, so the library named by the caller is loaded
// (PyTorch-ROCm ships its own librccl.so next to its own libamdhip64; a plain C++ host passes NULL for the system one).  "nccl" IS
// RCCL on ROCm; the one collective of the path is the all-gather of the collision step.
#include "host_internal.h"

namespace mrs_host {
RcclApi g_rccl;

int rccl_load(const char* path) {
  static std::mutex load_mtx;  // swarms of different host threads may initialise their communicators concurrently
  std::lock_guard<std::mutex> lk(load_mtx);
  if (g_rccl.lib) return MRS_OK;
  void* lib = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return fail(MRS_ERR_HIP, std::string("cannot load RCCL: ") + dlerror());
  g_rccl.GetUniqueId    = (int (*)(NcclId*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommDestroy    = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  g_rccl.AllGather      = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(lib, "ncclAllGather");
  g_rccl.CommInitRank   = (int (*)(void**, int, NcclId, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.CommCount      = (int (*)(void*, int*))dlsym(lib, "ncclCommCount");
  if (!g_rccl.GetUniqueId || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.CommInitRank) {
    g_rccl = RcclApi();
    dlclose(lib);
    return fail(MRS_ERR_HIP, "the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
  }
  g_rccl.lib = lib;  // last: the unlocked readers (the communicator calls) only run after a successful load
  return MRS_OK;
}
int rccl_check(int rc, const char* what) {
  if (rc == 0) return MRS_OK;
  return fail(MRS_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}
}  // namespace mrs_host

extern "C" {

int mrs_rccl_unique_id(const char* librccl_path, uint8_t* id128) {
  if (!id128) return fail(MRS_ERR_ARG, "null id");
  int rc = rccl_load(librccl_path);
  if (rc) return rc;
  NcclId id;
  if ((rc = rccl_check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId"))) return rc;
  memcpy(id128, id.internal, 128);
  return MRS_OK;
}

int mrs_swarm_comm_init(mrs_swarm_t* s, const char* librccl_path, int32_t world, int32_t rank, const uint8_t* id128, int64_t n_total) {
  MRS_ENTER(s);
  if (!s || !id128) return fail(MRS_ERR_ARG, "null argument");
  int rc = comm_setup(s, world, rank, n_total);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  if ((rc = rccl_load(librccl_path))) return rc;
  NcclId id;
  memcpy(id.internal, id128, 128);
  if ((rc = rccl_check(g_rccl.CommInitRank(&s->rccl_comm, world, id, rank), "ncclCommInitRank"))) return rc;
  return comm_buffers(s, world, rank, n_total);
}

}  // extern "C"
