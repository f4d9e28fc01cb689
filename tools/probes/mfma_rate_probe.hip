// Probe: sustained rate of v_mfma_f32_32x32x2_f32 on gfx950 (prices the 1-vs-K sweep, ge_1vk.hip).
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate_probe tools/probes/mfma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void __launch_bounds__(256) k_mfma(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[i][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out; CK(hipMalloc(&out, 256 * 4 * cus * 8 * sizeof(float)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int grid = cus * wps;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(k_mfma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double mfma_per_simd = (double)iters * 4 * wps;
    const double flops = (double)iters * 4 * 4096.0 * 4 * grid;   // 4096 flop per MFMA, 4 waves per block
    printf("v_mfma_f32_32x32x2_f32  %d wave(s)/SIMD: %.3f ms -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n",
           wps, ms, ms * 1e6 / mfma_per_simd, flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
