#include <hip/hip_runtime.h>
#include <cstdio>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__global__ void k(float* g, int nbytes, float* out, int aux_mode) {
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(g, 0, nbytes, 0x00020000);
  u32x4 v = {__float_as_uint(1.f + threadIdx.x), __float_as_uint(2.f), __float_as_uint(3.f), __float_as_uint(4.f)};
  unsigned off = (blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (aux_mode) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 16);
  else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, off, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  u32x4 r = aux_mode ? __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16) : __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(r.x) + __uint_as_float(r.w);
}
int main() {
  const int n = 256 * 4; float *g, *o; hipMalloc(&g, n * 16); hipMalloc(&o, n * 4);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(g, 0, n * 16); hipMemset(o, 0, n * 4);
    hipLaunchKernelGGL(k, dim3(4), dim3(256), 0, 0, g, n * 16, o, mode);
    float h[8], hg[8]; hipMemcpy(h, o, 32, hipMemcpyDeviceToHost); hipMemcpy(hg, g, 32, hipMemcpyDeviceToHost);
    printf("mode %d: readback %g %g (expect 5 6)  memory %g %g %g %g %g (expect 1 2 3 4 2)\n", mode, h[0], h[1], hg[0], hg[1], hg[2], hg[3], hg[4]);
  }
  return 0;
}
