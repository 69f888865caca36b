// daco_device.h -- device-side building blocks shared by the gfx950 kernels.
// wave64 only; DPP row operations of the GFX9 family (row_shr, row_bcast:15/31).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DACO_WAVE 64
// flag in daco_two_opt_auto's per-tour state word (sweeps done | flag): the search of that tour has ended
constexpr int TWO_OPT_DONE = 0x40000000;

namespace daco {

// ------------------------------------------------------------------ error plumbing (host)
void set_error(const char *fmt, ...);

// ------------------------------------------------------------------ candidate layout
// A row of n candidates is dealt to the 64 lanes in vectors of VEC floats: candidate
// k = (c*64 + lane)*VEC + v.  Rows are padded to ld = roundup(n, 64*VEC) so every lane's
// load is in bounds and naturally aligned.
__host__ __device__ inline int vec_for_n(int n) { return n > 128 ? 4 : (n > 64 ? 2 : 1); }
__host__ __device__ inline int ld_for_n(int n) {
  int w = 64 * vec_for_n(n);
  return (n + w - 1) / w * w;
}

// Workgroup -> work-item remap so that every XCD (block b runs on XCD b % 8) walks a
// contiguous range of work items: consecutive items share an instance, hence its rows stay
// in that XCD's private 4 MiB L2.  Bijective for any grid size.
__device__ inline int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// ------------------------------------------------------------------ 2-opt neighbour tables (daco_two_opt_prepare)
// per instance: header | nb[n][n] sorted (d, id) entries | rk[n][n] tolerance ranks; daco_two_opt_nbr.hip documents them
struct NbrEntry { float d; uint32_t id; };

constexpr size_t NBR_HEADER = 256;            // per-instance header: [0] bits of M = max off-diagonal |d|, [1] ~ordered(min off-diagonal d)
__host__ __device__ inline size_t nbr_align(size_t x) { return (x + 255) & ~(size_t)255; }
__host__ __device__ inline size_t nbr_instance_bytes(int n) {
  return NBR_HEADER + nbr_align((size_t)n * n * sizeof(NbrEntry)) + nbr_align((size_t)n * n * sizeof(uint16_t));
}
__device__ inline const NbrEntry *nbr_nb(const unsigned char *tab) { return reinterpret_cast<const NbrEntry *>(tab + NBR_HEADER); }
__device__ inline const uint16_t *nbr_rk(const unsigned char *tab, int n) {
  return reinterpret_cast<const uint16_t *>(tab + NBR_HEADER + nbr_align((size_t)n * n * sizeof(NbrEntry)));
}


// ------------------------------------------------------------------ zero fill as a kernel
// (instead of hipMemsetAsync: a captured HIP graph then consists of kernel nodes only -- the training step captured by
// pipeline.TspNlsTrainer came back with NaN gradients while its accumulators were cleared by memset nodes -- and dozens of
// per-instance memsets become one launch.)  `count` blocks of `bytes` bytes (a multiple of 4, the blocks 4-byte aligned),
// `stride` bytes apart.
template <int UNUSED = 0>
__global__ void __launch_bounds__(256) zero_blocks_kernel(unsigned char *base, size_t stride, size_t bytes) {
  uint32_t *p = reinterpret_cast<uint32_t *>(base + (size_t)blockIdx.y * stride);
  const size_t words = bytes >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
inline hipError_t zero_async(void *ptr, size_t bytes, hipStream_t s, int count = 1, size_t stride = 0) {
  if (bytes == 0 || count <= 0) return hipSuccess;
  size_t blocks = ((bytes >> 2) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((zero_blocks_kernel<0>), dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, s, (unsigned char *)ptr, stride, bytes);
  return hipGetLastError();
}

// ------------------------------------------------------------------ DPP helpers
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ inline float dpp_f(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL,
                                                    ROW_MASK, 0xF, BOUND));
}
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ inline int dpp_i(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xF, BOUND);
}
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

__device__ inline float readlane_f(float x, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
__device__ inline int readlane_i(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }

// Inclusive f32 add-scan over the wave in the order fixed by DESIGN.md 4.2 (and restated in
// oracle/daco_oracle.c lane_scan): Kogge-Stone inside each row of 16 lanes, then rows 1 and 3
// add the preceding row's total, then lanes 32..63 add lane 31.  Lanes without a source add
// +0.0f (x + 0 == x exactly).
__device__ inline float wave_scan_add(float x) {
  x = x + dpp_f<DPP_ROW_SHR(1), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(2), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(4), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_SHR(8), 0xF, true>(0.0f, x);
  x = x + dpp_f<DPP_ROW_BCAST15, 0xA, false>(0.0f, x);
  x = x + dpp_f<DPP_ROW_BCAST31, 0xC, false>(0.0f, x);
  return x;
}
// wave total in the same order (= lane 63 of the scan), broadcast to all lanes
__device__ inline float wave_sum(float x) { return readlane_f(wave_scan_add(x), 63); }

// (key, idx) reduction to lane 63 along the same network; `better(a_key,a_idx,b_key,b_idx)`
// must be a strict total preference so the result is independent of the tree.
struct KeyIdx { float key; int idx; };
template <bool MAX>
__device__ inline bool prefer(float ka, int ia, float kb, int ib) {
  // true if (ka, ia) should replace (kb, ib): better key, or equal key and smaller index
  return MAX ? (ka > kb || (ka == kb && ia < ib)) : (ka < kb || (ka == kb && ia < ib));
}
template <bool MAX, int CTRL, int ROW_MASK>
__device__ inline void arg_step(float &k, int &i) {
  // lanes without a source (or masked rows) read back their own value: a no-op
  const float ok = dpp_f<CTRL, ROW_MASK, false>(k, k);
  const int oi = dpp_i<CTRL, ROW_MASK, false>(i, i);
  if (prefer<MAX>(ok, oi, k, i)) { k = ok; i = oi; }
}
template <bool MAX>
__device__ inline KeyIdx wave_arg(float k, int i) {
  arg_step<MAX, DPP_ROW_SHR(1), 0xF>(k, i);
  arg_step<MAX, DPP_ROW_SHR(2), 0xF>(k, i);
  arg_step<MAX, DPP_ROW_SHR(4), 0xF>(k, i);
  arg_step<MAX, DPP_ROW_SHR(8), 0xF>(k, i);
  arg_step<MAX, DPP_ROW_BCAST15, 0xA>(k, i);
  arg_step<MAX, DPP_ROW_BCAST31, 0xC>(k, i);
  return KeyIdx{readlane_f(k, 63), readlane_i(i, 63)};
}

// ------------------------------------------------------------------ Philox4x32-10
struct u32x4 { uint32_t x, y, z, w; };
__device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
    u32x4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
enum : uint32_t { STREAM_START = 1, STREAM_RACE = 2, STREAM_SCAN = 3 };
__device__ inline u32x4 rng_block(uint64_t seed, uint64_t iter, uint32_t stream, uint32_t ant_gid,
                                  uint32_t idx) {
  u32x4 c;
  c.x = idx; c.y = ant_gid; c.z = (uint32_t)iter;
  c.w = (stream << 24) | (uint32_t)((iter >> 32) & 0xFFFFFFu);
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ inline uint32_t comp(const u32x4 &r, int i) {
  return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : r.w));
}
// uniform in (0,1): (2m+1) * 2^-24, m = top 23 bits -- exactly representable
__device__ inline float u01(uint32_t x) { return (float)(2u * (x >> 9) + 1u) * 0x1p-24f; }
// -log2(1-w): polynomial with explicit fma, bit-identical to oracle/daco_oracle.c neg_log2_1m
__device__ inline float neg_log2_1m(float w) {
  const float y = 1.0f - w;
  uint32_t b = __float_as_uint(y);
  int e = (int)(b >> 23) - 127;
  float t = __uint_as_float((b & 0x007FFFFFu) | 0x3F800000u);
  if (t > 1.41421354f) { t = t * 0.5f; e += 1; }
  const float s = t - 1.0f;
  float q = 0x1.025a2p-3f;
  q = __builtin_fmaf(q, s, -0x1.a8cc5cp-3f);
  q = __builtin_fmaf(q, s, 0x1.b9b11ep-3f);
  q = __builtin_fmaf(q, s, -0x1.e94f12p-3f);
  q = __builtin_fmaf(q, s, 0x1.26d41p-2f);
  q = __builtin_fmaf(q, s, -0x1.715c9cp-2f);
  q = __builtin_fmaf(q, s, 0x1.ec73d4p-2f);
  q = __builtin_fmaf(q, s, -0x1.71547p-1f);
  q = __builtin_fmaf(q, s, 0x1.715476p+0f);
  return -__builtin_fmaf(s, q, (float)e);
}

#define DACO_EPS_F32 1.1920928955078125e-07f
__device__ inline float clamp_log(float pr) {
  pr = pr < DACO_EPS_F32 ? DACO_EPS_F32 : pr;
  pr = pr > 1.0f - DACO_EPS_F32 ? 1.0f - DACO_EPS_F32 : pr;
  return logf(pr);
}

__device__ inline float pw(float x, float a) {
  if (a == 1.0f) return x;
  if (a == 2.0f) return x * x;
  if (a == 0.0f) return 1.0f;
  return powf(x, a);
}

// d(tau^a * eta^b)/d eta = b * tau^a * eta^(b-1), written as b * pk / eta wherever eta != 0 (the expression the
// gradient fixtures pin); at eta == 0 that would be 0/0, while the reference's autograd (tsp/aco.py:171-172)
// gives tau^a for b = 1, 0 for b > 1 and inf for b < 1.
__device__ inline float dprob_deta(float pk, float t, float e, float alpha, float beta) {
  if (e != 0.0f) return beta * (pk / e);
  if (beta == 1.0f) return pw(t, alpha);
  return beta > 1.0f ? 0.0f : __builtin_inff();
}

}  // namespace daco
