// ge_common.h -- shared device helpers for libge_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

#define GE_TAG_COIN 0x636F696Eu
#define GE_TAG_SLOT 0x736C6F74u
#define GE_TAG_PICK 0x7069636Bu
#define GE_TAG_SIDE 0x73696465u

// One coin for the whole batch (holE.py:137): heads iff the top bit is 0.
__device__ __forceinline__ bool batch_coin_heads(uint64_t seed, uint64_t step) {
  return (philox_w0((uint32_t)step, (uint32_t)(step >> 32), 0u, 0u, (uint32_t)seed ^ GE_TAG_COIN,
                    (uint32_t)(seed >> 32)) >> 31) == 0;
}

// Type-safe replacement of one entity (holE.py:104-112 + the host resample of holE.py:343-344):
// returns the corrupted column in `col` and the replacement id (or -1 for an unknown type).
__device__ __forceinline__ int32_t corrupt_one(const int32_t (&t)[3], int64_t i, bool batch_heads,
                                               const int32_t* __restrict__ id_to_type, int64_t N,
                                               const int64_t* __restrict__ type_offsets, int32_t n_types,
                                               const int32_t* __restrict__ type_ids, uint64_t seed,
                                               uint64_t step, int32_t padded_size, int32_t mode, int& col) {
  const uint32_t slo = (uint32_t)step, shi = (uint32_t)(step >> 32);
  const uint32_t klo = (uint32_t)seed, khi = (uint32_t)(seed >> 32);
  const uint32_t ilo = (uint32_t)i, ihi = (uint32_t)((uint64_t)i >> 32);
  bool heads;
  if (mode == GE_CORRUPT_BATCH_COIN) heads = batch_heads;
  else if (mode == GE_CORRUPT_ROW_COIN) heads = (philox_w0(slo, shi, ilo, ihi, klo ^ GE_TAG_SIDE, khi) >> 31) == 0;
  else heads = (mode == GE_CORRUPT_HEADS);
  col = heads ? 0 : 1;
  const int32_t x = t[col];
  int32_t repl = -1;  // unknown id -> the '?' default row of -1s (holE.py:39)
  if (x >= 0 && x < N) {
    const int32_t ty = id_to_type[x];  // id_to_type.lookup (holE.py:104)
    if (ty >= 0 && ty < n_types) {
      const int64_t off = type_offsets[ty];
      const uint64_t len = (uint64_t)(type_offsets[ty + 1] - off);
      if (len > 0) {
        uint32_t w = philox_w0(slo, shi, ilo, ihi, klo ^ GE_TAG_SLOT, khi);  // holE.py:108-110
        if (padded_size > 0) {
          const uint32_t slot = w % (uint32_t)padded_size;
          w = philox_w0(slo, shi, (uint32_t)ty, slot, klo ^ GE_TAG_PICK, khi);  // holE.py:343-344
        }
        repl = type_ids[off + (int64_t)(((uint64_t)w * len) >> 32)];
      }
    }
  }
  return repl;
}

}  // namespace ge
