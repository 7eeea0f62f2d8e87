// Microbenchmark: v_mfma_f32_32x32x2_f32 issue rate by (waves per SIMD, independent chains per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ void __launch_bounds__(512) k(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float av = a + threadIdx.x * 1e-6f, bv = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
    }
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int threads, const char* name) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4 * 8);
  int iters = 20000 / CHAINS;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS><<<256, threads>>>(d, 100, 1.0f, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS><<<256, threads>>>(d, iters, 1.0001f, 0.9999f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nmfma = 256.0 * (threads / 64) * iters * 16.0 * CHAINS;
  printf("%-40s %8.3f ms  %7.1f TFLOP/s\n", name, ms, nmfma * 4096.0 / ms / 1e9);
  hipFree(d);
}
int main() {
  run<1>(256, "1 wave/SIMD, 1 chain");
  run<2>(256, "1 wave/SIMD, 2 chains");
  run<4>(256, "1 wave/SIMD, 4 chains");
  run<1>(512, "2 waves/SIMD, 1 chain each");
  run<2>(512, "2 waves/SIMD, 2 chains each");
  return 0;
}
