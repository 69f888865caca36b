// Micro-benchmark: which access SHAPE sets the L2 -> L1 -> register rate of the tour-construction kernel?
// Every variant moves the same byte count per launch as tsp_scan32_kernel at the headline workload
// (TSP-500 x 512 ants x 64 instances: 32768 half-waves x 499 steps x 2 KB = 33.5 GB) out of L2/MALL-resident
// 1 MB matrices, with the same grid (4096 workgroups x 256 threads) and the same XCD-aware instance mapping.
// Only the address pattern / instruction changes:
//   0 stream     each wave walks its instance's matrix in consecutive 1 KB wave-loads (the friendliest shape)
//   1 rows       random 2 KB row per half-wave and step (the kernel's shape: 512 B per half-wave instruction)
//   2 rows_wave  random 2 KB rows, whole wave on one row (1 KB per instruction, the two ants time-multiplexed)
//   3 rows_2176  as 1 with the row stride padded to 2176 B (off the 2 KB period)
//   4 rows_2304  as 1 with the row stride 2304 B
//   5 same_row   every wave of a workgroup reads the SAME row sequence (L1 hits: what the CU can ingest)
//   6 lines      every 8-lane group reads its own random 128 B line (no row locality at all)
//   7 rows_lds   as 2 through global_load_lds_dwordx4 (LDS-DMA, no VGPR write-back)
//   8 stream_lds as 0 through global_load_lds_dwordx4
//   9 rows_dw    as 1 with dword loads (16 instructions of 128 B per half-wave)
//  10 rows_c64   as 1 with 64 CONTIGUOUS bytes per lane (lane s owns bytes 64s..64s+63: candidates in index order)
// build: hipcc -O3 --offload-arch=gfx950 tools/l2_bw_shapes.hip -o tools/l2_bw_shapes
// run:   tools/l2_bw_shapes [instances=64] [lds_pad_bytes=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) shapes(const float *P, int B, int n, int ldb, int A, int steps, float *out) {
  extern __shared__ float dyn[];
  __shared__ v4 stage[4][4][64];                // LDS-DMA landing zone: [wave][load][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 31, up = lane >> 5;
  const int orig = blockIdx.x, nwg = gridDim.x;
  const int q = nwg / 8, r = nwg % 8, x = orig % 8, i = orig / 8;
  const int w = x < r ? x * (q + 1) + i : r * (q + 1) + (x - r) * q + i;       // XCD x walks whole instances
  const int bpi = A / 8, b = (w / bpi) % B, a = ((w % bpi) * 4 + wave) * 2 + up;
  const size_t inst_bytes = (size_t)n * ldb;
  const char *Pb = (const char *)P + (size_t)b * inst_bytes;
  unsigned prev = (a * 7919u + 13u) % n;
  const unsigned gw = (w % bpi) * 4 + wave;                                     // wave index inside the instance
  v4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < steps; ++t) {
    v4 r0, r1, r2, r3;
    if (MODE == 0 || MODE == 8) {
      const unsigned blk = (gw * 37u + t) * 4u;                                 // 4 consecutive KB, then the next 4
      const unsigned nblk = (unsigned)(inst_bytes / 1024);
      const unsigned o = (blk % (nblk - 3)) * 1024u + lane * 16u;
      if (MODE == 0) {
        r0 = *(const v4 *)(Pb + o); r1 = *(const v4 *)(Pb + o + 1024); r2 = *(const v4 *)(Pb + o + 2048); r3 = *(const v4 *)(Pb + o + 3072);
      } else {
        __builtin_amdgcn_global_load_lds((const void *)(Pb + o), (__attribute__((address_space(3))) void *)&stage[wave][0][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void *)(Pb + o + 1024), (__attribute__((address_space(3))) void *)&stage[wave][1][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void *)(Pb + o + 2048), (__attribute__((address_space(3))) void *)&stage[wave][2][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void *)(Pb + o + 3072), (__attribute__((address_space(3))) void *)&stage[wave][3][0], 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0) lgkmcnt(0)
        r0 = stage[wave][0][lane]; r1 = r2 = r3 = r0;
      }
    } else if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) {
      const unsigned p = MODE == 5 ? (unsigned)((t * 131u + blockIdx.x * 17u) % (unsigned)n) : prev;
      const unsigned o = p * (unsigned)ldb + s * 16u;
      r0 = *(const v4 *)(Pb + o); r1 = *(const v4 *)(Pb + o + 512); r2 = *(const v4 *)(Pb + o + 1024); r3 = *(const v4 *)(Pb + o + 1536);
    } else if (MODE == 2 || MODE == 7) {
      const unsigned pa = __builtin_amdgcn_readlane(prev, 0), pb = __builtin_amdgcn_readlane(prev, 32);
      const unsigned oa = pa * (unsigned)ldb + lane * 16u, ob = pb * (unsigned)ldb + lane * 16u;
      if (MODE == 2) {
        r0 = *(const v4 *)(Pb + oa); r1 = *(const v4 *)(Pb + oa + 1024); r2 = *(const v4 *)(Pb + ob); r3 = *(const v4 *)(Pb + ob + 1024);
      } else {
        __builtin_amdgcn_global_load_lds((const void *)(Pb + oa), (__attribute__((address_space(3))) void *)&stage[wave][0][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void *)(Pb + oa + 1024), (__attribute__((address_space(3))) void *)&stage[wave][1][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void *)(Pb + ob), (__attribute__((address_space(3))) void *)&stage[wave][2][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void *)(Pb + ob + 1024), (__attribute__((address_space(3))) void *)&stage[wave][3][0], 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0070);
        r0 = stage[wave][0][lane]; r1 = r2 = r3 = r0;
      }
    } else if (MODE == 6) {
      const unsigned grp = lane >> 3, within = lane & 7;
      unsigned o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned h = (prev + grp * 977u + j * 131u + t * 7u) * 2654435761u;
        o[j] = ((h >> 8) % (unsigned)(inst_bytes / 128)) * 128u + within * 16u;
      }
      r0 = *(const v4 *)(Pb + o[0]); r1 = *(const v4 *)(Pb + o[1]); r2 = *(const v4 *)(Pb + o[2]); r3 = *(const v4 *)(Pb + o[3]);
    } else if (MODE == 10) {
      const unsigned o = prev * (unsigned)ldb + s * 64u;
      r0 = *(const v4 *)(Pb + o); r1 = *(const v4 *)(Pb + o + 16); r2 = *(const v4 *)(Pb + o + 32); r3 = *(const v4 *)(Pb + o + 48);
    } else {   // 9: dword loads
      const unsigned o = prev * (unsigned)ldb + s * 4u;
      float f[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = *(const float *)(Pb + o + j * 128);
      r0 = (v4){f[0], f[1], f[2], f[3]}; r1 = (v4){f[4], f[5], f[6], f[7]};
      r2 = (v4){f[8], f[9], f[10], f[11]}; r3 = (v4){f[12], f[13], f[14], f[15]};
    }
    acc += r0 + r1 + r2 + r3;
    unsigned h = prev * 2654435761u + t * 40503u + a;
    h += (unsigned)__builtin_amdgcn_readlane(__float_as_int(r0.x), up * 32) & 1023u;      // next row depends on the data
    prev = (h >> 7) % (unsigned)n;
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x + dyn[0];
}

template <int MODE>
static void run(const char *name, const float *P, int B, int n, int ldb, float *out, int lds) {
  const int A = 512, steps = 499, launches = 3;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < launches; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(shapes<MODE>, dim3(64 * A / 8), dim3(256), lds, 0, P, B, n, ldb, A, steps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double bytes = 64.0 * A * steps * 2048.0;
  printf("%-11s instances=%-3d stride=%-5d lds_pad=%-6d %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 2.4 GHz)\n", name, B, ldb, lds, best,
         bytes / best * 1e-9, bytes / (best * 1e-3) / 256.0 / 2.4e9);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, lds = argc > 2 ? atoi(argv[2]) : 0;
  const int n = 500;
  float *P, *out;
  const size_t bytes = (size_t)64 * n * 2304 + 4096;
  hipMalloc(&P, bytes); hipMalloc(&out, 4);
  std::vector<float> h(bytes / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
  hipMemcpy(P, h.data(), bytes, hipMemcpyHostToDevice);
  run<0>("stream", P, B, n, 2048, out, lds);
  run<1>("rows", P, B, n, 2048, out, lds);
  run<2>("rows_wave", P, B, n, 2048, out, lds);
  run<3>("rows_2176", P, B, n, 2176, out, lds);
  run<4>("rows_2304", P, B, n, 2304, out, lds);
  run<5>("same_row", P, B, n, 2048, out, lds);
  run<6>("lines", P, B, n, 2048, out, lds);
  run<7>("rows_lds", P, B, n, 2048, out, lds);
  run<8>("stream_lds", P, B, n, 2048, out, lds);
  run<9>("rows_dw", P, B, n, 2048, out, lds);
  run<10>("rows_c64", P, B, n, 2048, out, lds);
  return 0;
}
