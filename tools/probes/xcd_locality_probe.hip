// Probe: does data written by a workgroup in kernel 1 stay readable from the SAME XCD's L2 in kernel 2?
// (prices an XCD-aware slot->item assignment for the gradient rows of the training step)
// Kernel 1: workgroup b writes region b (6.4 KB).  Kernel 2: workgroup b reads region (b + shift) % grid.
// shift = 0 -> same workgroup index (same XCD if dispatch is round-robin), shift = 1 -> next XCD, shift = 8 -> same XCD again.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void __launch_bounds__(256) writer(float4* buf, int per_block, float v) {
  float4* p = buf + (size_t)blockIdx.x * per_block;
  for (int i = threadIdx.x; i < per_block; i += 256) p[i] = make_float4(v, v + i, v, v);
}
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void __launch_bounds__(256) reader(const f4v* buf, int per_block, int shift, float* out) {
  const int src = (blockIdx.x + shift) % gridDim.x;
  const f4v* p = buf + (size_t)src * per_block;
  float acc = 0.f;
  for (int i = threadIdx.x; i < per_block; i += 256) {
    f4v x = NT ? __builtin_nontemporal_load(p + i) : p[i];
    acc += x.x + x.y + x.z + x.w;
  }
  if (acc == -1.f) out[0] = acc;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// reader whose loads carry sc1 (aux bit 4): served past this XCD's L2
__global__ void __launch_bounds__(256) reader_sc1(const float4* buf, int per_block, int shift, float* out, int total_bytes) {
  const int src = (blockIdx.x + shift) % gridDim.x;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, total_bytes, 0x00020000);
  float acc = 0.f;
  for (int i = threadIdx.x; i < per_block; i += 256) {
    const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)(((size_t)src * per_block + i) * 16), 0, 16);
    acc += __uint_as_float(u.x) + __uint_as_float(u.y);
  }
  if (acc == -1.f) out[0] = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int grid = 2048, per_block = 400;            // 400 float4 = 6.4 KB per workgroup, 13.1 MB in all
  float4* buf; float* out; CK(hipMalloc(&buf, (size_t)grid * per_block * 16)); CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
  for (int nt = 0; nt < 2; ++nt)
  for (int shift : {0, 1, 4, 8, 9, 1024}) {
    float tw = 0, tr = 0; const int reps = 200;
    for (int r = 0; r < reps + 10; ++r) {
      hipExtLaunchKernelGGL(writer, dim3(grid), dim3(256), 0, 0, e0, e1, 0, buf, per_block, (float)r);
      if (nt) hipExtLaunchKernelGGL(reader<true>, dim3(grid), dim3(256), 0, 0, e2, e3, 0, (const f4v*)buf, per_block, shift, out);
      else hipExtLaunchKernelGGL(reader<false>, dim3(grid), dim3(256), 0, 0, e2, e3, 0, (const f4v*)buf, per_block, shift, out);
      CK(hipEventSynchronize(e3));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e2, e3));
      if (r >= 10) { tw += a; tr += b; }
    }
    printf("%s shift %4d: writer %.2f us, reader %.2f us\n", nt ? "nontemporal reads" : "plain reads      ", shift, tw / reps * 1e3, tr / reps * 1e3);
  }
  // sc1 loads in the reader (shift 1): do they leave the writer on the fast path?
  {
    float tw = 0, tr = 0; const int reps = 200;
    for (int r = 0; r < reps + 10; ++r) {
      hipExtLaunchKernelGGL(writer, dim3(grid), dim3(256), 0, 0, e0, e1, 0, buf, per_block, (float)r);
      hipExtLaunchKernelGGL(reader_sc1, dim3(grid), dim3(256), 0, 0, e2, e3, 0, (const float4*)buf, per_block, 1, out, grid * per_block * 16);
      CK(hipEventSynchronize(e3));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e2, e3));
      if (r >= 10) { tw += a; tr += b; }
    }
    printf("sc1 reads         shift    1: writer %.2f us, reader %.2f us\n", tw / reps * 1e3, tr / reps * 1e3);
  }
  // Table-like pattern: EVERY XCD reads the whole buffer (8 passes with shifts 0..7), then it is rewritten.
  {
    float tw = 0; const int reps = 100;
    for (int r = 0; r < reps + 10; ++r) {
      hipExtLaunchKernelGGL(writer, dim3(grid), dim3(256), 0, 0, e0, e1, 0, buf, per_block, (float)r);
      for (int sh = 0; sh < 8; ++sh)
        hipLaunchKernelGGL(reader<false>, dim3(grid), dim3(256), 0, 0, (const f4v*)buf, per_block, sh, out);
      CK(hipDeviceSynchronize());
      float a; CK(hipEventElapsedTime(&a, e0, e1));
      if (r >= 10) tw += a;
    }
    printf("read by all 8 XCDs, then rewritten: writer %.2f us\n", tw / reps * 1e3);
  }
  // Ring experiment: the pair (writer, reader with shift 1) walks a ring of R regions of 13.1 MB, so a region is
  // rewritten only every R iterations -- does the extra cost of rewriting lines last read on another XCD age out?
  for (int R : {1, 4, 16, 64}) {
    float4* ring; CK(hipMalloc(&ring, (size_t)R * grid * per_block * 16));
    CK(hipMemset(ring, 0, (size_t)R * grid * per_block * 16));
    float tw = 0, tr = 0; const int reps = 256;
    for (int r = 0; r < reps + 64; ++r) {
      float4* reg = ring + (size_t)(r % R) * grid * per_block;
      hipExtLaunchKernelGGL(writer, dim3(grid), dim3(256), 0, 0, e0, e1, 0, reg, per_block, (float)r);
      hipExtLaunchKernelGGL(reader<false>, dim3(grid), dim3(256), 0, 0, e2, e3, 0, (const f4v*)reg, per_block, 1, out);
      CK(hipEventSynchronize(e3));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e2, e3));
      if (r >= 64) { tw += a; tr += b; }
    }
    printf("ring of %2d regions (%4.0f MB), shift 1: writer %.2f us, reader %.2f us\n", R, R * 13.1, tw / reps * 1e3, tr / reps * 1e3);
    CK(hipFree(ring));
  }
  return 0;
}
