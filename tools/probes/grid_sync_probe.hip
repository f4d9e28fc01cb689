// Probe: cost of a cooperative grid barrier on gfx950 (decides whether a persistent multi-step
// training kernel can beat two launches per step). Build: hipcc --offload-arch=gfx950 -O3 -o
// gpurun_out/grid_sync_probe tools/probes/grid_sync_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(512) sync_loop(float* buf, int64_t n_per_phase, int iters) {
  cg::grid_group grid = cg::this_grid();
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  for (int it = 0; it < iters; ++it) {
    // phase A: write n floats; phase B: read what a far-away thread wrote
    for (int64_t i = tid; i < n_per_phase; i += nthreads) buf[i] = (float)(it + i);
    grid.sync();
    float acc = 0.f;
    for (int64_t i = tid; i < n_per_phase; i += nthreads) acc += buf[(i + n_per_phase / 2) % n_per_phase];
    if (acc == -1.f) buf[0] = acc;
    grid.sync();
  }
}

// hand-rolled barrier: one atomic counter, sense via generation number
__device__ __forceinline__ void my_grid_sync(unsigned* ctr, unsigned nblocks, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    ++gen;
    const unsigned target = gen * nblocks;
    atomicAdd(ctr, 1u);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();
  }
  __syncthreads();
}

__global__ void __launch_bounds__(512) sync_loop_manual(float* buf, int64_t n_per_phase, int iters, unsigned* ctr) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  unsigned gen = 0;
  for (int it = 0; it < iters; ++it) {
    for (int64_t i = tid; i < n_per_phase; i += nthreads) buf[i] = (float)(it + i);
    my_grid_sync(ctr, gridDim.x, gen);
    float acc = 0.f;
    for (int64_t i = tid; i < n_per_phase; i += nthreads) acc += buf[(i + n_per_phase / 2) % n_per_phase];
    if (acc == -1.f) buf[0] = acc;
    my_grid_sync(ctr, gridDim.x, gen);
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sync_loop, 512, 0));
  printf("CUs %d, blocks/CU(512 thr) %d, cooperativeLaunch %d\n", cus, per_cu, prop.cooperativeLaunch);
  float* buf; CK(hipMalloc(&buf, 64 << 20));
  unsigned* ctr; CK(hipMalloc(&ctr, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 200;
  for (int64_t mb : {0, 1, 4, 13}) {
    int64_t n = mb * (1 << 20) / 4; if (n == 0) n = 1;
    for (int grid_mult = 1; grid_mult <= 1; ++grid_mult) {
      int grid = cus * grid_mult; int it = iters;
      void* args[] = {&buf, &n, &it};
      CK(hipLaunchCooperativeKernel((void*)sync_loop, dim3(grid), dim3(512), args, 0, 0));  // warm
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(a, 0));
      CK(hipLaunchCooperativeKernel((void*)sync_loop, dim3(grid), dim3(512), args, 0, 0));
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("cg    grid %d x512, %2lld MB/phase: %.2f us per (write+sync+read+sync)\n", grid, (long long)mb, ms * 1e3 / iters);
      CK(hipMemset(ctr, 0, 4));
      hipLaunchKernelGGL(sync_loop_manual, dim3(grid), dim3(512), 0, 0, buf, n, it, ctr);  // warm
      CK(hipDeviceSynchronize());
      CK(hipMemset(ctr, 0, 4));
      CK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(sync_loop_manual, dim3(grid), dim3(512), 0, 0, buf, n, it, ctr);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
      printf("manual grid %d x512, %2lld MB/phase: %.2f us per (write+sync+read+sync)\n", grid, (long long)mb, ms * 1e3 / iters);
    }
  }
  return 0;
}
