// vmm_alloc.hip -- (placement experiment, tools/placement_vmm.py) device buffers built from the HIP virtual-memory API instead of
// hipMalloc: one reserved virtual range backed by physical handles of a chosen size, mapped in order or in a shuffled order.
// Built here (cross-compiled) into build/libvmm_alloc.so; the GPU box only loads it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <random>
#include <vector>

struct VmmBuf {
  void* va = nullptr;
  size_t size = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;
};

extern "C" {

// granularity[0] = minimum, [1] = recommended
int vmm_granularity(int device, size_t* gran) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  hipError_t e = hipMemGetAllocationGranularity(&gran[0], &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess) return (int)e;
  return (int)hipMemGetAllocationGranularity(&gran[1], &prop, hipMemAllocationGranularityRecommended);
}

// bytes rounded up to a multiple of chunk; chunk = 0: ONE physical handle for the whole range.  shuffle != 0: the chunks are mapped
// in a random order (seed = shuffle).  align: alignment of the virtual range (0 = the chunk size).
int vmm_alloc(int device, size_t bytes, size_t chunk, size_t align, unsigned shuffle, VmmBuf** out, void** ptr) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess) return (int)e;
  if (chunk == 0) chunk = (bytes + gran - 1) / gran * gran;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = (bytes + chunk - 1) / chunk;
  VmmBuf* b = new VmmBuf;
  b->size = n * chunk;
  e = hipMemAddressReserve(&b->va, b->size, align ? align : std::min<size_t>(chunk, (size_t)1 << 30), nullptr, 0);
  if (e != hipSuccess) { delete b; return (int)e; }
  b->handles.resize(n);
  for (size_t i = 0; i < n; i++) {
    e = hipMemCreate(&b->handles[i], chunk, &prop, 0);
    if (e != hipSuccess) { fprintf(stderr, "hipMemCreate chunk %zu/%zu: %s\n", i, n, hipGetErrorString(e)); return (int)e; }
  }
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = i;
  if (shuffle) { std::mt19937 g(shuffle); std::shuffle(order.begin(), order.end(), g); }
  for (size_t i = 0; i < n; i++) {
    e = hipMemMap((char*)b->va + i * chunk, chunk, 0, b->handles[order[i]], 0);
    if (e != hipSuccess) { fprintf(stderr, "hipMemMap %zu: %s\n", i, hipGetErrorString(e)); return (int)e; }
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(b->va, b->size, &acc, 1);
  if (e != hipSuccess) { fprintf(stderr, "hipMemSetAccess: %s\n", hipGetErrorString(e)); return (int)e; }
  *out = b;
  *ptr = b->va;
  return 0;
}

int vmm_free(VmmBuf* b) {
  if (!b) return 0;
  hipDeviceSynchronize();
  hipMemUnmap(b->va, b->size);
  for (auto h : b->handles) hipMemRelease(h);
  hipMemAddressFree(b->va, b->size);
  delete b;
  return 0;
}

}  // extern "C"
