// Developer probe: device-side cost per dependent kernel launch, stream loop vs hipGraph replay (gfx950, ROCm 7.2).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_tiny(float* p, int spin) {
  float v = p[threadIdx.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  int N = argc > 1 ? atoi(argv[1]) : 1000;
  float* p;
  hipMalloc(&p, 4096);
  hipMemset(p, 0, 4096);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int spin : {0, 20000}) {
    for (int blocks : {1, 1024}) {
      // one kernel alone
      k_tiny<<<blocks, 64, 0, s>>>(p, spin);
      hipStreamSynchronize(s);
      hipEventRecord(e0, s);
      k_tiny<<<blocks, 64, 0, s>>>(p, spin);
      hipEventRecord(e1, s);
      hipStreamSynchronize(s);
      float one = 0;
      hipEventElapsedTime(&one, e0, e1);
      // stream loop
      double h0 = now_ms();
      hipEventRecord(e0, s);
      for (int i = 0; i < N; ++i) k_tiny<<<blocks, 64, 0, s>>>(p, spin);
      hipEventRecord(e1, s);
      double h1 = now_ms();
      hipStreamSynchronize(s);
      float loop = 0;
      hipEventElapsedTime(&loop, e0, e1);
      // graph
      hipGraph_t g;
      hipGraphExec_t ge;
      hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
      for (int i = 0; i < N; ++i) k_tiny<<<blocks, 64, 0, s>>>(p, spin);
      hipStreamEndCapture(s, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphLaunch(ge, s);
      hipStreamSynchronize(s);
      double g0 = now_ms();
      hipEventRecord(e0, s);
      hipGraphLaunch(ge, s);
      hipEventRecord(e1, s);
      double g1 = now_ms();
      hipStreamSynchronize(s);
      float gr = 0;
      hipEventElapsedTime(&gr, e0, e1);
      printf("spin %5d blocks %4d: one kernel %.1f us | stream loop %.2f us/kernel device (host %.2f us/launch) | graph %.2f us/node device (host launch %.2f ms)\n",
             spin, blocks, one * 1e3, loop * 1e3 / N, (h1 - h0) * 1e3 / N, gr * 1e3 / N, g1 - g0);
      hipGraphExecDestroy(ge);
      hipGraphDestroy(g);
    }
  }
  return 0;
}
