// Can a B = 1 kernel write the x-dependent COO values at their FINAL (scattered) positions in pinned host memory?
// 26,048 doubles written by one launch into a 607,424-entry array: contiguous (the compact layout of today) against scattered with
// the strides of the reference's COO blocks (9, 12, 3 ...), into coherent (fine-grained), non-coherent and default pinned memory.
// hipcc --offload-arch=gfx950 -O3 host_scatter.hip -o /tmp/host_scatter && /tmp/host_scatter
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
__global__ void scatter(double* dst, const int* idx, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[idx[i]] = v + i;
}
int main() {
  const int n = 26048, total = 607424;
  std::vector<int> contiguous(n), strided(n);
  for (int i = 0; i < n; i++) contiguous[i] = i;
  // node-major blocks: entry (slot s, node j) of a 64-node chunk with w slots lands at base + j * w + s  (w = 9, 12, 3, 4 ...)
  { int k = 0, base = 1000; const int ws[6] = {9, 9, 12, 3, 4, 3};
    while (k < n) for (int b = 0; b < 6 && k < n; b++) { const int w = ws[b]; for (int s = 0; s < w && k < n; s++) for (int j = 0; j < 64 && k < n; j++) strided[k++] = base + j * (w + 2) + s; base += 64 * (w + 2) + 17; } }
  int *d_c, *d_s;
  CK(hipMalloc(&d_c, n * 4)); CK(hipMalloc(&d_s, n * 4));
  CK(hipMemcpy(d_c, contiguous.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_s, strided.data(), n * 4, hipMemcpyHostToDevice));
  struct { const char* name; unsigned flags; } kinds[3] = {{"default", hipHostMallocDefault}, {"coherent", hipHostMallocCoherent}, {"non-coherent", hipHostMallocNonCoherent}};
  for (auto& kd : kinds) {
    double* h;
    CK(hipHostMalloc((void**)&h, (size_t)total * 8, kd.flags));
    for (int i = 0; i < total; i++) h[i] = -1.0;
    for (int pat = 0; pat < 2; pat++) {
      const int* idx = pat ? d_s : d_c;
      const std::vector<int>& hidx = pat ? strided : contiguous;
      for (int w = 0; w < 20; w++) { hipLaunchKernelGGL(scatter, dim3((n + 255) / 256), dim3(256), 0, 0, h, idx, n, 1.0); CK(hipDeviceSynchronize()); }
      const int reps = 300;
      auto t0 = std::chrono::steady_clock::now();
      double sum = 0.0;
      for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(scatter, dim3((n + 255) / 256), dim3(256), 0, 0, h, idx, n, (double)r);
        CK(hipStreamSynchronize(0));
        sum += h[hidx[(r * 7919) % n]];        // the host reads a value the launch has just written
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
      // every value visible and right after the last launch?
      int bad = 0;
      for (int i = 0; i < n; i++) bad += (h[hidx[i]] != (double)(reps - 1) + i);
      printf("%-13s %-10s  launch + synchronise + one host read: %7.2f us   wrong values after the last launch: %d   (checksum %g)\n", kd.name, pat ? "scattered" : "contiguous", us, bad, sum);
    }
    CK(hipHostFree(h));
  }
  return 0;
}
