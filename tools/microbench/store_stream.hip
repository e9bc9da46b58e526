// Store-stream microbenchmark: what does the fused kernel's output pattern cost, and which variations of it are cheaper?
// Every wavefront owns one contiguous block of SLOTS x 512 B (like one (vector, phase) block of compact Jacobian values)
// and writes it slot by slot with `work` dependent fp64 FMAs per lane in between (the arithmetic of the sweeps).
//   mode 0: 8 B per lane per store (512 B per instruction), non-temporal        -- what the kernel does
//   mode 1: the same, plain stores
//   mode 2: 16 B per lane (1 KB per instruction: two slots at once)
//   mode 3: no stores (compute only)
//   mode 4: all stores at the END of the wavefront's life (values kept in LDS meanwhile)
//   mode 5: mode 0 but the block of wavefront w is block (w * 2654435761) % nwaves: neighbours in time are not neighbours in memory
// hipcc --offload-arch=gfx950 -O3 store_stream.hip -o store_stream && ./store_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int SLOTS = 62;
typedef unsigned u2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(double* out, int nwaves, int work, double seed) {
  extern __shared__ double lds[];   // 40 KB per workgroup: 4 workgroups per CU like the fused kernel
  const int lane = threadIdx.x & 63;
  const long long w0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (w0 >= nwaves) return;
  const long long w = (MODE == 5) ? (long long)((unsigned long long)w0 * 2654435761ull % (unsigned long long)nwaves) : w0;
  double* blk = out + (size_t)w * SLOTS * 64;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)blk, 0, -1, 0x00020000);
  double v = seed + lane * 1e-3, acc = 0.0, prev = 0.0;
  double* mine = lds + (threadIdx.x >> 6) * 1216 + lane;
  for (int s = 0; s < SLOTS; s++) {
    for (int i = 0; i < work; i++) v = __builtin_fma(v, 0.999999, 1e-7);   // dependent chain: 4 cycles each
    acc += v;
    if (MODE == 0 || MODE == 5) {
      u2 d; __builtin_memcpy(&d, &v, 8);
      __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane * 8, s * 512, 2);
    } else if (MODE == 1) {
      u2 d; __builtin_memcpy(&d, &v, 8);
      __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane * 8, s * 512, 0);
    } else if (MODE == 2) {
      if (s & 1) {
        const double p[2] = {prev, v};
        u4 d; __builtin_memcpy(&d, p, 16);
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, lane * 16, (s >> 1) * 1024, 2);
      } else prev = v;
    } else if (MODE == 4) {
      if (s < 19) mine[s * 64] = v;   // only 19 slots fit: the rest is recomputed below (cheap) -- timing experiment
    }
  }
  if (MODE == 4) {
    for (int s = 0; s < SLOTS; s++) {
      const double t = (s < 19) ? mine[s * 64] : v + s;
      u2 d; __builtin_memcpy(&d, &t, 8);
      __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane * 8, s * 512, 2);
    }
  }
  if (acc == 1.2345e300) out[0] = acc;
}

template <int MODE>
float run(double* d, int nwaves, int work, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = (nwaves * 64 + 255) / 256;
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 40 * 1024, 0, d, nwaves, work, 1.0);
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 40 * 1024, 0, d, nwaves, work, 1.0 + i);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int nwaves = 98304;
  const size_t bytes = (size_t)nwaves * SLOTS * 512;
  double* d; hipMalloc(&d, bytes);
  for (int i = 0; i < 200; i++) run<3>(d, nwaves, 60, 1);   // settle the clocks
  const int works[] = {0, 15, 30, 60, 90};
  printf("bytes per launch %.2f GB; ms per launch by mode (work = dependent FMAs between two stores)\n", bytes / 1e9);
  printf("%6s %10s %10s %10s %10s %10s %10s\n", "work", "nt 8B", "plain 8B", "nt 16B", "no store", "at end", "scattered");
  for (int w : works) {
    printf("%6d %10.4f %10.4f %10.4f %10.4f %10.4f %10.4f\n", w, run<0>(d, nwaves, w, 20), run<1>(d, nwaves, w, 20),
           run<2>(d, nwaves, w, 20), run<3>(d, nwaves, w, 20), run<4>(d, nwaves, w, 20), run<5>(d, nwaves, w, 20));
  }
  return 0;
}
