// Does a kernel's SCRATCH requirement make its launch slow?  Two kernels that exit at once (flag == 0): one without private
// memory, three that reserve 352 / 1376 / 64 bytes of scratch per lane (a dynamically indexed private array behind the
// exit).  1 000 launches each on one stream, wall time per launch; then pairs alternating on two streams.
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_scratch.hip -o scripts/_bin/ubench_scratch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

template <int WORDS>
__global__ void __launch_bounds__(64) k(const int* flag, int* out) {
  if (*flag == 0) return;
  if constexpr (WORDS > 0) {
    int a[WORDS];
    for (int i = 0; i < WORDS; i++) a[i] = i * threadIdx.x + flag[i & 3];
    int s = 0;
    for (int i = 0; i < WORDS; i++) s += a[(i * 7 + flag[1]) % WORDS];
    out[threadIdx.x] = s;
  } else {
    out[threadIdx.x] = *flag;
  }
}

template <int WORDS>
static double run(hipStream_t st, const int* flag, int* out, int n, int grid) {
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k<WORDS>, dim3(grid), dim3(64), 0, st, flag, out);
  (void)hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(k<WORDS>, dim3(grid), dim3(64), 0, st, flag, out);
  (void)hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  int *flag, *out;
  (void)hipMalloc(&flag, 16);
  (void)hipMemset(flag, 0, 16);
  (void)hipMalloc(&out, 4096);
  hipStream_t st;
  (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k<0>));
  printf("k<0>    private bytes per lane %zu\n", (size_t)fa.localSizeBytes);
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k<16>));
  printf("k<16>   private bytes per lane %zu\n", (size_t)fa.localSizeBytes);
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k<88>));
  printf("k<88>   private bytes per lane %zu\n", (size_t)fa.localSizeBytes);
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k<344>));
  printf("k<344>  private bytes per lane %zu\n", (size_t)fa.localSizeBytes);
  for (int grid : {64, 512}) {
    printf("grid %d one-wave workgroups, empty kernels, us per launch: no scratch %.1f | 64 B %.1f | 352 B %.1f | 1376 B %.1f\n", grid,
           run<0>(st, flag, out, 1000, grid), run<16>(st, flag, out, 1000, grid), run<88>(st, flag, out, 1000, grid), run<344>(st, flag, out, 1000, grid));
  }
  // alternating kernels of different scratch sizes on one stream (what a reduction chain looks like)
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 500; i++) {
    hipLaunchKernelGGL(k<88>, dim3(64), dim3(64), 0, st, flag, out);
    hipLaunchKernelGGL(k<0>, dim3(64), dim3(64), 0, st, flag, out);
    hipLaunchKernelGGL(k<344>, dim3(64), dim3(64), 0, st, flag, out);
  }
  (void)hipStreamSynchronize(st);
  printf("alternating 352 B / none / 1376 B: %.1f us per launch\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 1500);
  return 0;
}
