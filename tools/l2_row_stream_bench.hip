// Micro-benchmark: how fast can wavefronts stream random 2 KB rows of L2-resident matrices?
// Same traffic shape as the tour-construction kernel (TSP-500, 512 ants, 64 instances): every
// half-wave reads one 2 KB row per step, the next row index depends on the data just read.
// It bounds that kernel from above: no arithmetic, no LDS, only the row fetches.
// build: hipcc -O3 --offload-arch=gfx950 tools/l2_row_stream_bench.hip -o tools/l2_row_stream_bench
// run:   tools/l2_row_stream_bench [mode] [dynamic LDS bytes, to cap occupancy]
//   mode 0 independent row indices, 1 index depends on the loaded data, 2 non-temporal loads,
//        3 64 contiguous bytes per lane, 4 whole wave per row (1 KB per instruction), 5 8-byte loads
// MI355X, 2026-09: 1.78 ms = 18.8 TB/s for modes 0/1/4 at any occupancy (profiles/r01_g_l2_row_stream.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void __launch_bounds__(256) rows(const float *P, int n, int ld, int A, int steps, int dep, float *out) {
  extern __shared__ float pad[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 31, up = lane >> 5;
  const int orig = blockIdx.x, nwg = gridDim.x;
  const int q = nwg / 8, r = nwg % 8, x = orig % 8, i = orig / 8;
  const int w = x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;
  const int bpi = A / 8, b = w / bpi, a = ((w % bpi) * 4 + wave) * 2 + up;
  const char *Pb = (const char *)(P + (size_t)b * n * ld);
  unsigned prev = (a * 7919u + 13u) % n;
  float acc = 0.f;
  for (int t = 0; t < steps; ++t) {
    const unsigned off = prev * (unsigned)ld * 4u + s * 16u;
    float v;
    if (dep < 2) {
      float4 r0 = *(const float4 *)(Pb + off), r1 = *(const float4 *)(Pb + off + 512), r2 = *(const float4 *)(Pb + off + 1024),
             r3 = *(const float4 *)(Pb + off + 1536);
      v = r0.x + r1.y + r2.z + r3.w;
    } else if (dep == 2) {   // non-temporal
      typedef float v4 __attribute__((ext_vector_type(4)));
      v4 r0 = __builtin_nontemporal_load((const v4 *)(Pb + off)), r1 = __builtin_nontemporal_load((const v4 *)(Pb + off + 512)),
         r2 = __builtin_nontemporal_load((const v4 *)(Pb + off + 1024)), r3 = __builtin_nontemporal_load((const v4 *)(Pb + off + 1536));
      v = r0.x + r1.y + r2.z + r3.w;
    } else if (dep == 3) {   // each lane 64 contiguous bytes (lane s covers bytes s*64 .. s*64+63)
      const unsigned o2 = prev * (unsigned)ld * 4u + s * 64u;
      float4 r0 = *(const float4 *)(Pb + o2), r1 = *(const float4 *)(Pb + o2 + 16), r2 = *(const float4 *)(Pb + o2 + 32),
             r3 = *(const float4 *)(Pb + o2 + 48);
      v = r0.x + r1.y + r2.z + r3.w;
    } else if (dep == 4) {   // whole wave on one row: 1 KB per instruction, two rows per wave-step (lower / upper ant in turn)
      const unsigned pa = __builtin_amdgcn_readlane(prev, 0), pb = __builtin_amdgcn_readlane(prev, 32);
      const unsigned oa = pa * (unsigned)ld * 4u + lane * 16u, ob = pb * (unsigned)ld * 4u + lane * 16u;
      float4 r0 = *(const float4 *)(Pb + oa), r1 = *(const float4 *)(Pb + oa + 1024), r2 = *(const float4 *)(Pb + ob),
             r3 = *(const float4 *)(Pb + ob + 1024);
      v = r0.x + r1.y + r2.z + r3.w;
    } else {                 // dwordx2: 8 loads of 256 B per half
      const unsigned o2 = prev * (unsigned)ld * 4u + s * 8u;
      float2 q0 = *(const float2 *)(Pb + o2), q1 = *(const float2 *)(Pb + o2 + 256), q2 = *(const float2 *)(Pb + o2 + 512), q3 = *(const float2 *)(Pb + o2 + 768),
             q4 = *(const float2 *)(Pb + o2 + 1024), q5 = *(const float2 *)(Pb + o2 + 1280), q6 = *(const float2 *)(Pb + o2 + 1536), q7 = *(const float2 *)(Pb + o2 + 1792);
      v = q0.x + q1.y + q2.x + q3.y + q4.x + q5.y + q6.x + q7.y;
    }
    acc += v;
    unsigned h = prev * 2654435761u + t * 40503u + a;
    if (dep == 1) h += (unsigned)__builtin_amdgcn_readlane(__float_as_int(v), up * 32) & 1023u;
    prev = (h >> 7) % (unsigned)n;
  }
  if (acc == 123.456f) out[0] = acc + pad[0];
}
int main(int argc, char **argv) {
  int dep = argc > 1 ? atoi(argv[1]) : 1, lds = argc > 2 ? atoi(argv[2]) : 0;
  const int B = 64, n = 500, ld = 512, A = 512, steps = 499;
  float *P, *out;
  hipMalloc(&P, (size_t)B * n * ld * 4); hipMalloc(&out, 4);
  std::vector<float> h((size_t)B * n * ld);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
  hipMemcpy(P, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(rows, dim3(B * A / 8), dim3(256), lds, 0, P, n, ld, A, steps, dep, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)B * A * steps * 2048.0;
    printf("dep=%d lds_pad=%d: %.3f ms, %.2f TB/s row traffic\n", dep, lds, ms, bytes / ms * 1e-9);
  }
  return 0;
}
