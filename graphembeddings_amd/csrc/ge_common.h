// ge_common.h -- shared device helpers for libge_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ge_hip.h"

namespace ge {

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kBlock = 256;       // 4 waves, one per SIMD
constexpr int kMaxBlocks = 256 * 8;  // 256 CUs x 8 resident 256-thread blocks

template <int VEC> struct Vec;
template <> struct Vec<1> { using type = float; };
template <> struct Vec<2> { using type = float2; };
template <> struct Vec<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* __restrict__ p, float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    float4 v = *reinterpret_cast<const float4*>(p);
    r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
  } else if constexpr (VEC == 2) {
    float2 v = *reinterpret_cast<const float2*>(p);
    r[0] = v.x; r[1] = v.y;
  } else {
    r[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* __restrict__ p, const float (&r)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  } else if constexpr (VEC == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]);
  } else {
    *p = r[0];
  }
}

// Sum over a group of LPT consecutive lanes (LPT power of two <= 64); every lane gets the sum.
// Must be called by all 64 lanes of the wave.
template <int LPT>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = LPT / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

// clip_by_norm scale of tf.nn.embedding_lookup(max_norm) (holE.py:162; op chain
// holE-20170724/graph.pbtxt:3159-3596): (t*c) * min(rsqrt(sum t^2), 1/c).
// rsqrt(0) = inf -> the minimum picks 1/c -> scale 1.
__device__ __forceinline__ float clip_scale(float ss, float max_norm, float& inv) {
  inv = rsqrtf(ss);
  return fminf(inv, 1.0f / max_norm) * max_norm;
}

__device__ __forceinline__ float sigmoidf_dev(float x) { return 1.0f / (1.0f + expf(-x)); }

// fp32 atomic add without return: one global_atomic_add_f32 (no CAS loop).
__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline int grid_for(int64_t units, int units_per_block) {
  int64_t g = (units + units_per_block - 1) / units_per_block;
  if (g < 1) g = 1;
  if (g > kMaxBlocks) g = kMaxBlocks;
  return (int)g;
}

inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// Philox4x32-10, the sampler's stream (stated in oracle/hole_oracle.py).
__device__ __forceinline__ uint32_t philox_w0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}

}  // namespace ge
