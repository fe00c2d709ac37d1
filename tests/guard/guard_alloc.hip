// guard_alloc.hip — TEST HELPER (tests/test_gpu_guard.py), not product code.
// Device memory whose END is followed by unmapped address space: hipMemAddressReserve of (mapped + one granule),
// physical memory mapped over the first part only.  A kernel that reads or writes one byte past a buffer placed at the
// end of the mapped part takes a GPU memory-access fault instead of silently touching a neighbouring allocation
// (unconditional loads past a call shorter than one chunk went unnoticed for exactly that reason).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

extern "C" {

struct guard_buf {
  void *base;        // start of the reservation
  size_t reserved;   // bytes of address space reserved (mapped + guard)
  size_t mapped;     // bytes backed by memory, a multiple of the granularity
  hipMemGenericAllocationHandle_t handle;
  void *ptr;         // base + mapped - bytes: the caller's buffer, ending exactly where the mapping ends
};

#define GCHK(e)                                                                          \
  do {                                                                                   \
    hipError_t _r = (e);                                                                 \
    if (_r != hipSuccess) {                                                              \
      fprintf(stderr, "guard_alloc: %s: %s\n", #e, hipGetErrorString(_r));               \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

int guard_alloc(size_t bytes, guard_buf *out) {
  int dev = 0;
  GCHK(hipGetDevice(&dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  GCHK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  if (!gran || !bytes) return -1;
  out->mapped = (bytes + gran - 1) / gran * gran;
  out->reserved = out->mapped + gran;
  GCHK(hipMemAddressReserve(&out->base, out->reserved, 0, nullptr, 0));
  GCHK(hipMemCreate(&out->handle, out->mapped, &prop, 0));
  GCHK(hipMemMap(out->base, out->mapped, 0, out->handle, 0));
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  GCHK(hipMemSetAccess(out->base, out->mapped, &acc, 1));
  // no fill: a memset (a blit kernel, through the L2) followed by a host upload that bypasses the L2 left stale zero
  // lines in front of the uploaded data on this mapping (seen as whole channels of silence in a later test of the same
  // process); the caller overwrites its buffer completely, by a device-side copy
  out->ptr = static_cast<char *>(out->base) + out->mapped - bytes;
  return 0;
}

// device buffer -> the guarded buffer by a device-side copy (through the L2, like the kernels' own reads), complete on return
int guard_upload(guard_buf *g, const void *dev_src, size_t bytes) {
  GCHK(hipDeviceSynchronize());
  GCHK(hipMemcpy(g->ptr, dev_src, bytes, hipMemcpyDeviceToDevice));
  GCHK(hipDeviceSynchronize());
  return 0;
}

int guard_free(guard_buf *g) {
  GCHK(hipDeviceSynchronize());
  GCHK(hipMemUnmap(g->base, g->mapped));
  GCHK(hipMemRelease(g->handle));
  GCHK(hipMemAddressFree(g->base, g->reserved));
  return 0;
}

}  // extern "C"
