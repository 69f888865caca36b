// Micro-benchmark: what does a 4-byte GATHER cost on the vector-memory path, by address pattern and active lanes?
// The 2-opt kernels evaluate change(i, j) from two gathers d[t[i-1]][t[j]], d[t[i]][t[j+1]] out of an L2-resident 1 MB
// matrix; this measures the rate at which a CU retires such wave-level gather instructions when nothing else is going on
// (8 waves per SIMD, 8 independent gathers in flight per wave), i.e. the ceiling of the 2-opt pair-evaluation rate.
//   pattern 0: all active lanes read ONE matrix row at random columns (<= 16 cache lines)   [full-row phase of 2-opt]
//   pattern 1: every group of 32 lanes reads its own row                                     [patch phase, G = 32]
//   pattern 2: every lane reads its own row (64 distinct lines)
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_rate.hip -o tools/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) gather(const float *d, int n, int pattern, int active, int iters, float *out) {
  const int lane = threadIdx.x & 63;
  const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float *base = d + (size_t)(blockIdx.x % 64) * n * n;            // 64 instances of 1 MB, as in config 3
  unsigned h = wid * 2654435761u + lane * 40503u;
  float acc = 0.f;
  if (lane < active) {
    for (int it = 0; it < iters; ++it) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        h = h * 1664525u + 1013904223u;
        const unsigned hr = (it * 8 + k + wid) * 2246822519u;          // wave-uniform row hash
        unsigned row = pattern == 0 ? (hr >> 8) % n : pattern == 1 ? ((hr >> 8) + (lane >> 5) * 131u) % n : (h >> 20) % n;
        const unsigned col = (h >> 8) % n;
        v[k] = base[row * (unsigned)n + col];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
  }
  if (acc == 123.f) out[0] = acc;
}
int main() {
  const int n = 500, iters = 200;
  float *d, *out;
  hipMalloc(&d, (size_t)64 * n * n * 4); hipMalloc(&out, 4);
  hipMemset(d, 0, (size_t)64 * n * n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 8;                                             // 8 workgroups (32 waves) per CU
  for (int pattern = 0; pattern < 3; ++pattern)
    for (int active : {64, 32, 16}) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(gather, dim3(grid), dim3(256), 0, 0, d, n, pattern, active, iters, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double winstr = (double)grid * 4 * iters * 8;               // wave-level gather instructions
      printf("pattern %d active %2d: %.3f ms  %.2f G wave-gathers/s  %.1f clk per wave-gather per CU (2.4 GHz)  %.1f G lane-gathers/s\n",
             pattern, active, best, winstr / best * 1e-6, best * 1e-3 * 2.4e9 / (winstr / 256), winstr * active / best * 1e-6);
    }
  return 0;
}
