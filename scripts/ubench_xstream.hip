// Latency of a cross-stream dependency: stream A runs a ~1 ms kernel and records an event; stream B waits for the event and
// runs an empty kernel.  Reported: (end of B's kernel) - (end of A's kernel), from timing events, for B waiting through
// hipStreamWaitEvent vs. the empty kernel queued behind A's kernel on A's own stream.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_xstream.hip -o scripts/_bin/ubench_xstream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>

__global__ void spin(long long cycles, int* out) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
__global__ void empty(int* out) {
  if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 2;
}

int main() {
  hipStream_t a, b;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  (void)hipStreamCreateWithPriority(&a, hipStreamNonBlocking, 0);
  (void)hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
  int* out;
  (void)hipMalloc(&out, 64);
  hipEvent_t ea, eb, dep;
  (void)hipEventCreate(&ea);
  (void)hipEventCreate(&eb);
  (void)hipEventCreateWithFlags(&dep, hipEventDisableTiming);
  for (int grid : {256, 16384}) {
    for (int mode = 0; mode < 2; mode++) {
      std::vector<float> d;
      for (int it = 0; it < 30; it++) {
        hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, a, 2000000LL, out);  // ~1 ms at 2 GHz per wave
        (void)hipEventRecord(ea, a);
        if (mode == 0) {
          (void)hipEventRecord(dep, a);
          (void)hipStreamWaitEvent(b, dep, 0);
          hipLaunchKernelGGL(empty, dim3(64), dim3(64), 0, b, out);
          (void)hipEventRecord(eb, b);
        } else {
          hipLaunchKernelGGL(empty, dim3(64), dim3(64), 0, a, out);
          (void)hipEventRecord(eb, a);
        }
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ea, eb);
        if (it >= 5) d.push_back(ms * 1000.f);
      }
      std::sort(d.begin(), d.end());
      printf("producer grid %5d waves, consumer %s: end-to-end gap median %.1f us (min %.1f, max %.1f)\n", grid,
             mode == 0 ? "on another stream behind hipStreamWaitEvent" : "on the same stream", d[d.size() / 2], d.front(), d.back());
    }
  }
  return 0;
}
