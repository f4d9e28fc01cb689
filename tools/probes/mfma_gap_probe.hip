// How many instructions of each kind fit in the shadow of one v_mfma_f32_32x32x16_f16 issued by the SAME wave (one wave
// per SIMD, as in the rank sweep)?  Loop of { MFMA ; K gap instructions }, cycles per iteration from s_memtime.
// hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_gap_probe.hip -o /tmp/mfma_gap_probe && /tmp/mfma_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int K>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters, float lo) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  float x = threadIdx.x * 0.25f, y = 1.0f, x1 = x + 1, x2 = x + 2, x3 = x + 3;
  int m = 0; unsigned carry = 0;
  __shared__ float4 sm[512];
  sm[threadIdx.x & 511] = make_float4(x, y, x, y);
  if (blockDim.x == 256) sm[threadIdx.x + 256] = make_float4(y, x, y, x);
  __syncthreads();
  float4 l4 = make_float4(0, 0, 0, 0);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
        if (KIND == 1) { unsigned long long s; asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(s) : "v"(x), "v"(lo)); asm volatile("" :: "s"(s)); }
        if (KIND == 2) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(m) : "s"(it));
        if (KIND == 3) { unsigned long long s; asm volatile("s_andn2_b64 %0, %1, exec" : "=s"(s) : "s"((unsigned long long)it)); asm volatile("" :: "s"(s)); }
        if (KIND == 4) { asm volatile("ds_read_b128 %0, %1" : "=v"(l4) : "v"((unsigned)((threadIdx.x + k * 16) & 511) * 16u)); }
        if (KIND == 6) { if (k % 4 == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y)); if (k % 4 == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(y));
                         if (k % 4 == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(y)); if (k % 4 == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(y)); }
        if (KIND == 7) { float r; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(acc[(u + 2) & 3][k & 15])); asm volatile("" :: "v"(r)); }
        if (KIND == 5) {   // the real piece mix: cmp x4 | andn2 x2 + addc x2 | writelane x4, cycling
          const int ph = k % 3;
          if (ph == 0) { unsigned long long s0, s1; asm volatile("v_cmp_lt_f32_e64 %0, %2, %3\n v_cmp_le_f32_e64 %1, %2, %3\n v_cmp_lt_f32_e64 %0, %3, %2\n v_cmp_le_f32_e64 %1, %3, %2" : "=&s"(s0), "=&s"(s1) : "v"(x), "v"(lo)); asm volatile("" :: "s"(s0), "s"(s1)); }
          if (ph == 1) { asm volatile("s_andn2_b64 vcc, exec, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc\n s_andn2_b64 vcc, exec, %1\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(carry) : "s"((unsigned long long)it) : "vcc"); }
          if (ph == 2) { asm volatile("v_writelane_b32 %0, %1, 3\n v_writelane_b32 %0, %1, 7\n s_nop 0\n v_writelane_b32 %0, %1, 11\n v_writelane_b32 %0, %1, 15" : "+v"(m) : "s"(it)); }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = x + x1 + x2 + x3 + m + carry + l4.x;
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int K, int THREADS = 256>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  probe<KIND, K><<<256, THREADS>>>(out, cyc, iters, 0.5f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  probe<KIND, K><<<256, THREADS>>>(out, cyc, iters, 0.5f);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[256];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < 256; ++i) mean += (double)h[i];
  mean /= 256.0 * iters * 4;
  printf("%-10s K=%2d waves/SIMD=%d  %.1f ticks, %.2f ns per MFMA of one wave\n", name, K, THREADS / 256, mean, ms * 1e6 / (iters * 4.0));
}

int main() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8);
#define ROW(KIND, NAME) run<KIND, 0>(NAME, out, cyc); run<KIND, 2>(NAME, out, cyc); run<KIND, 4>(NAME, out, cyc); run<KIND, 6>(NAME, out, cyc); run<KIND, 8>(NAME, out, cyc); run<KIND, 12>(NAME, out, cyc);
  ROW(0, "v_fma") ROW(1, "v_cmp->s") ROW(2, "writelane") ROW(3, "s_andn2") ROW(4, "ds_read128")
  ROW(6, "v_fma x4") ROW(7, "accread")
  run<0, 0, 512>("v_fma", out, cyc); run<0, 4, 512>("v_fma", out, cyc); run<0, 8, 512>("v_fma", out, cyc); run<0, 12, 512>("v_fma", out, cyc);
  run<6, 4, 512>("v_fma x4", out, cyc); run<6, 8, 512>("v_fma x4", out, cyc); run<6, 12, 512>("v_fma x4", out, cyc);
  run<5, 1, 512>("mix(4)", out, cyc); run<5, 2, 512>("mix(8)", out, cyc); run<5, 3, 512>("mix(12)", out, cyc);
  run<5, 1>("mix(4)", out, cyc); run<5, 2>("mix(8)", out, cyc); run<5, 3>("mix(12)", out, cyc);
  return 0;
}
