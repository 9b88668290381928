// The one exchange on the data path: COCO's cross-GPU gather of the [CLS] rows (COCO/modeling.py:182-190), as a native RCCL
// call for hosts that own an ncclComm_t (the Python host goes through torch.distributed's all_gather_into_tensor, which is
// the same RCCL collective on the process group's communicator).  The library does not link RCCL: the symbols are taken
// from the copy the process already loaded (torch ships one) or, failing that, from librccl.so.1 on the loader path - two
// RCCL copies in one process would each keep their own communicator registry.
#include <dlfcn.h>

#include "common.h"

namespace {
typedef int (*allgather_fn)(const void*, void*, size_t, int /* ncclDataType_t */, void* /* ncclComm_t */, hipStream_t);
typedef const char* (*errstr_fn)(int);
constexpr int kNcclFloat32 = 7;  // rccl.h: ncclFloat32 = 7

// the loader's message of the last failed dlopen / dlsym (dlerror() clears its state when read: read it exactly once, right
// after the failing call)
const char* g_dl_error = nullptr;
void* rccl_handle() {
  static void* h = nullptr;
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy this process already uses
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) {
    dlerror();  // drop the NOLOAD probes' messages
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) g_dl_error = dlerror();
  }
  return h;
}
}  // namespace

extern "C" int cocodr_allgather_rows(const float* local_rows, float* gathered, int rows, int H, void* nccl_comm,
                                     cocodr_stream_t stream) {
  CK_ARG(local_rows && gathered && nccl_comm, "allgather_rows: null pointer");
  CK_ARG(rows > 0 && H > 0, "allgather_rows: bad shape rows=%d H=%d", rows, H);
  void* h = rccl_handle();
  if (!h) {
    cocodr_set_error("allgather_rows: RCCL (librccl.so.1) is not loadable in this process: %s", g_dl_error ? g_dl_error : "dlopen failed");
    return COCODR_ERR_LAUNCH;
  }
  dlerror();
  allgather_fn ag = (allgather_fn)dlsym(h, "ncclAllGather");
  if (!ag) {
    const char* e = dlerror();
    cocodr_set_error("allgather_rows: the loaded RCCL has no ncclAllGather: %s", e ? e : "symbol ncclAllGather missing");
    return COCODR_ERR_LAUNCH;
  }
  const int rc = ag(local_rows, gathered, (size_t)rows * H, kNcclFloat32, nccl_comm, (hipStream_t)stream);
  if (rc != 0) {
    errstr_fn es = (errstr_fn)dlsym(h, "ncclGetErrorString");
    cocodr_set_error("allgather_rows: ncclAllGather failed: %s", es ? es(rc) : "unknown RCCL error");
    return COCODR_ERR_LAUNCH;
  }
  return COCODR_OK;
}
