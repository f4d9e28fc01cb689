// Probe: issue rate of v_fmac_f32 vs v_pk_fma_f32 on gfx950 (prices the HolE correlation loops).
// Build on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate_probe tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_fma(float* out, int iters, float a, float b) {
  float c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = __builtin_fmaf(c[i], a, b);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_pkfma(float* out, int iters, float a, float b) {
  f2 c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
  const f2 av = f2{a, a * 1.0001f}, bv = f2{b, b * 0.9999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = __builtin_elementwise_fma(c[i], av, bv);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c[i].x + c[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("CUs %d, clock %d kHz\n", cus, prop.clockRate);
  float* out; CK(hipMalloc(&out, 256 * 4 * cus * 8 * sizeof(float)));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD: block = 4 waves, wps blocks per CU
    const int grid = cus * wps;
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a, 0));
        if (which == 0) hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k_pkfma, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      }
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      const double instr_per_simd = (double)iters * 16 * wps;       // wave-instructions issued per SIMD
      const double fma = (double)iters * 16 * 256.0 * grid * (which ? 2 : 1);
      printf("%s  %d wave(s)/SIMD: %.3f ms  -> %.2f ns per wave-instruction per SIMD, %.1f TFLOP/s\n",
             which ? "v_pk_fma_f32" : "v_fma_f32   ", wps, ms, ms * 1e6 / instr_per_simd, 2.0 * fma / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
