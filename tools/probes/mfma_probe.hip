// Microbenchmark behind the rank kernel's inner loop: what does a 64-MFMA chunk cost one wave per SIMD
// (a) from registers, (b) with its 64 operand reads from LDS up front, (c) plus a workgroup barrier,
// (d) with the operand reads of the NEXT chunk issued between this chunk's MFMAs.
// hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int LDA = 201, LDB = 33;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters, const float* __restrict__ src) {
  extern __shared__ float lds[];
  float* A = lds; float* Bs = lds + 128 * LDA;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1, li = lane & 31, lh = lane >> 5;
  for (int i = t; i < 128 * LDA + 2 * 128 * LDB; i += 256) lds[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  float a0[16], a1[16], b0[16], b1[16];
  for (int s = 0; s < 16; ++s) { a0[s] = 0.5f + s; a1[s] = 0.25f * lane; b0[s] = 0.125f * s; b1[s] = 1.f; }
  float n0[16], n1[16], m0[16], m1[16];
  const float* ap = A + (wm * 64 + li) * LDA + lh;
  const float* bp = Bs + (wn * 64 + li) * LDB + lh;
  if (MODE == 3) {
    for (int s = 0; s < 16; ++s) { a0[s] = ap[2 * s]; a1[s] = ap[32 * LDA + 2 * s]; b0[s] = bp[2 * s]; b1[s] = bp[32 * LDB + 2 * s]; }
  }
  float v0 = lane, v1 = 2.f * lane, v2 = 0.5f, v3 = 1.5f;
  float4 g[4] = {};
  float gs = 0.f;
  const float* gp = src + (size_t)(blockIdx.x * 128 + (t >> 1)) * 200 + (t & 1) * 16;
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const int ch = it % 6, buf = it & 1;
    const float* app = ap + ch * 32;
    const float* bpp = bp + buf * 128 * LDB;
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { a0[s] = app[2 * s]; a1[s] = app[32 * LDA + 2 * s]; b0[s] = bpp[2 * s]; b1[s] = bpp[32 * LDB + 2 * s]; }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
      if (MODE == 7) { __builtin_amdgcn_sched_barrier(0); v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f); __builtin_amdgcn_sched_barrier(0); }
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
      if (MODE == 7) { __builtin_amdgcn_sched_barrier(0); v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f); __builtin_amdgcn_sched_barrier(0); }
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
      if (MODE == 7) { __builtin_amdgcn_sched_barrier(0); v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f); __builtin_amdgcn_sched_barrier(0); }
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
      if (MODE == 7) { __builtin_amdgcn_sched_barrier(0); v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f); __builtin_amdgcn_sched_barrier(0); }
      if (MODE == 4 || MODE == 6) {      // 4 independent VALU ops per MFMA group of 4 -> x NV
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < (MODE == 4 ? 2 : 6); ++r) {
          v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
          v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 7) {                   // the same 128 FMAs as mode 4, one per MFMA ... (placed after each MFMA below)
      }
      if (MODE == 8 && s == 0) {         // ... or all in one burst per chunk
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 32; ++r) {
          v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
          v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 5) {
        __builtin_amdgcn_sched_barrier(0);
        if (s < 4) g[s] = *reinterpret_cast<const float4*>(gp + ((it * 32) % 192) + 4 * s);
        if (s == 8) gs += g[0].x + g[1].y + g[2].z + g[3].w;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (MODE == 3) {
        __builtin_amdgcn_sched_barrier(0);
        n0[s] = app[2 * s]; n1[s] = app[32 * LDA + 2 * s]; m0[s] = bpp[2 * s]; m1[s] = bpp[32 * LDB + 2 * s];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE == 3) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { a0[s] = n0[s]; a1[s] = n1[s]; b0[s] = m0[s]; b1[s] = m1[s]; }
    }
    if (MODE == 2 || MODE == 3) __syncthreads();
  }
  long long c1 = clock64();
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int q = 0; q < 16; ++q) s += acc[a][b][q];
  out[blockIdx.x * 256 + t] = s + v0 + v1 + v2 + v3 + gs;
  if (blockIdx.x == 0 && t == 0) *cyc = c1 - c0;
}

// f16 MFMA (32 cycles each): NF independent FMAs after every MFMA -- do they hide in the gap?
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int NF>
__global__ __launch_bounds__(256) void probe_f16(float* out, long long* cyc, int iters) {
  const int t = threadIdx.x, lane = t & 63;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  h8 A[2], B[2];
  for (int i = 0; i < 8; ++i) { A[0][i] = (_Float16)(0.5f + i); A[1][i] = (_Float16)(0.25f * lane); B[0][i] = (_Float16)(0.125f * i); B[1][i] = (_Float16)1.f; }
  float v0 = lane, v1 = 2.f * lane, v2 = 0.5f, v3 = 1.5f;
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 24; ++p) {
      acc[(p >> 1) & 1][p & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[(p >> 1) & 1], B[p & 1], acc[(p >> 1) & 1][p & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < NF; ++r) {
        if ((r & 3) == 0) v0 = __builtin_fmaf(v0, 1.0001f, 0.5f);
        else if ((r & 3) == 1) v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
        else if ((r & 3) == 2) v2 = __builtin_fmaf(v2, 1.0002f, 0.125f);
        else v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long c1 = clock64();
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int q = 0; q < 16; ++q) s += acc[a][b][q];
  out[blockIdx.x * 256 + t] = s + v0 + v1 + v2 + v3;
  if (blockIdx.x == 0 && t == 0) *cyc = c1 - c0;
}

template <int NF>
void run_f16(float* out, long long* cyc, int iters) {
  probe_f16<NF><<<256, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  probe_f16<NF><<<256, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("f16 32x32x16 MFMA, %d FMAs after each: %7.1f ticks per MFMA\n", NF, (double)c / iters / 24);
}

template <int MODE>
void run(const char* name, float* out, long long* cyc, int iters, const float* src) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 256, 150000>>>(out, cyc, iters, src);
  hipEventRecord(e0);
  probe<MODE><<<256, 256, 150000>>>(out, cyc, iters, src);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  double flops = 256.0 * 4 * iters * 64 * 4096.0;
  printf("%-34s %8.3f ms  %7.1f s_memtime ticks/chunk  %6.1f TFLOP/s  (%.0f ns/chunk)\n", name, ms, (double)c / iters, flops / (ms * 1e-3) / 1e12, ms * 1e6 / iters);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  float* src; hipMalloc(&src, (size_t)256 * 128 * 200 * 4 + 4096); hipMemset(src, 0, (size_t)256 * 128 * 200 * 4 + 4096);
  run<0>("mfma from registers", out, cyc, iters, src);
  run<1>("+ 64 LDS operand reads up front", out, cyc, iters, src);
  run<2>("+ workgroup barrier", out, cyc, iters, src);
  run<3>("reads of next chunk between MFMAs", out, cyc, iters, src);
  run<4>("regs + 8 FMAs per 4 MFMAs", out, cyc, iters, src);
  run<6>("regs + 24 FMAs per 4 MFMAs", out, cyc, iters, src);
  run<5>("regs + 4 global dwordx4 per chunk", out, cyc, iters, src);
  run<7>("regs + 128 FMAs, 2 after every MFMA", out, cyc, iters, src);
  run<8>("regs + 128 FMAs in one burst", out, cyc, iters, src);
  run_f16<0>(out, cyc, iters); run_f16<2>(out, cyc, iters); run_f16<4>(out, cyc, iters); run_f16<6>(out, cyc, iters); run_f16<8>(out, cyc, iters);
  return 0;
}
