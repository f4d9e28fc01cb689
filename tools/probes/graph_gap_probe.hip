// Does a captured hipGraph shorten the period of a chain of DEPENDENT short kernels (the config-2 training step: two ~5 us
// kernels a step, each waiting for the one before it)?  A chain of 2N kernels, each a dependent chase of `hops` loads through a
// buffer the previous kernel wrote, (a) launched one by one into a stream with the host far ahead, (b) captured once into a graph
// of 2N kernel nodes and launched as one graph.  Prints the period per kernel of both.
// hipcc -O3 --offload-arch=gfx950 tools/probes/graph_gap_probe.hip -o /tmp/graph_gap_probe && /tmp/graph_gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void chase(const int* __restrict__ in, int* __restrict__ out, int n, int hops) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = i;
  for (int h = 0; h < hops; ++h) j = in[j];          // dependent round trips (L2 / Infinity Cache resident)
  out[i] = (j + 1) % n;
}

int main() {
  const int n = 512 * 256, N = 400;
  int *a, *b;
  CK(hipMalloc(&a, n * sizeof(int)));
  CK(hipMalloc(&b, n * sizeof(int)));
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 13) % n);
  CK(hipMemcpy(a, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
  CK(hipMemcpy(b, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int hops : {1, 3}) {
    auto chain = [&]() {
      for (int s = 0; s < N; ++s) {
        hipLaunchKernelGGL(chase, dim3(512), dim3(256), 0, st, a, b, n, hops);
        hipLaunchKernelGGL(chase, dim3(512), dim3(256), 0, st, b, a, n, hops);
      }
    };
    chain();
    CK(hipStreamSynchronize(st));
    float ms_stream = 0.f, ms_graph = 0.f;
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, st));
    chain();
    auto t1 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms_stream, e0, e1));
    const double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    chain();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms_graph, e0, e1));
    printf("hops %d: stream %.2f us per kernel (host enqueue %.2f us per kernel), graph %.2f us per kernel\n", hops,
           ms_stream * 1e3 / (2 * N), host_us / (2 * N), ms_graph * 1e3 / (2 * N));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
