// Host-side helpers around the HIP runtime (no kernels).
#include "common.h"

// Streams made only to occupy hardware-queue slots while a forked HIP graph is instantiated (see cgrna.h).
extern "C" int cg_queue_fillers_create(int n, void** out) {
  if (n < 0 || n > CG_MAX_QUEUE_FILLERS || (n > 0 && !out)) return CG_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) {
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipStreamDestroy((hipStream_t)out[j]);
      return (int)e;
    }
    out[i] = (void*)s;
  }
  return CG_OK;
}

extern "C" int cg_queue_fillers_destroy(int n, void** handles) {
  if (n < 0 || n > CG_MAX_QUEUE_FILLERS || (n > 0 && !handles)) return CG_ERR_BAD_ARG;
  int status = CG_OK;
  for (int i = 0; i < n; ++i) {
    if (!handles[i]) continue;
    hipError_t e = hipStreamDestroy((hipStream_t)handles[i]);
    if (e != hipSuccess && status == CG_OK) status = (int)e;
    handles[i] = nullptr;
  }
  return status;
}
