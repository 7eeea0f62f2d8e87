// Returning global atomics from every wave of the chip onto Q counters `stride` ints apart: how long do T of them take?
// (the filter kernel's candidate flush: one atomicAdd per candidate entry on count[query])
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_bench tools/microbench/atomic_contention_bench.hip && /tmp/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) atomics_kernel(int* count, int Q, int stride, int per_wave, int* sink) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * 256 + threadIdx.x) >> 6;
  int acc = 0;
  if (lane < per_wave) {
    const int q = (gw * 7 + lane * 3) % Q;
    acc = atomicAdd(count + (size_t)q * stride, 1);
  }
  if (acc == -1) sink[0] = acc;
}

__global__ void __launch_bounds__(256) empty_kernel(int* sink) {
  if (threadIdx.x == 1024) sink[0] = 1;
}

int main() {
  int* count;
  int* sink;
  hipMalloc(&count, 64 << 20);
  hipMalloc(&sink, 4);
  hipMemset(count, 0, 64 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int waves = 2048, grid = waves / 4;
  auto time_it = [&](auto launch) {
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 50 * 1000.f;
  };
  const float base = time_it([&] { hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, 0, sink); });
  printf("empty launch: %.2f us\n", base);
  for (int Q : {1, 16, 32, 256}) {
    for (int stride : {1, 32, 64, 1024, 16384}) {
      for (int per_wave : {1, 6, 32}) {
        const float us = time_it([&] { hipLaunchKernelGGL(atomics_kernel, dim3(grid), dim3(256), 0, 0, count, Q, stride, per_wave, sink); });
        printf("Q=%4d stride=%6d ints  per_wave=%2d  total=%6d atomics: %7.2f us  (%.2f ns per atomic over the empty launch)\n", Q,
               stride, per_wave, waves * per_wave, us, (us - base) * 1000.f / (waves * per_wave));
      }
    }
  }
  return 0;
}
