// Microbenchmark 2: MFMA chain whose A operand comes from LDS via ds_read_b128 (one read per 4 MFMAs), as in the top-k kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>  // 0: A from LDS each 4 MFMAs; 1: + v_max epilogue every 128 MFMAs; 2: MODE1 + barrier per 128
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, float a) {
  extern __shared__ float4 smem4[];
  float* smem = (float*)smem4;
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 32 * 260; i += 512) smem[i] = a + i * 1e-7f;
  __syncthreads();
  float breg[128];
#pragma unroll
  for (int m = 0; m < 128; ++m) breg[m] = a + m * 1e-6f + lane * 1e-7f;
  float keep = 0.f;
  const float* arow = smem + j * 260 + h * 128;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, breg[4 * c], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, breg[4 * c + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, breg[4 * c + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, breg[4 * c + 3], acc, 0, 0, 0);
    }
    if (MODE >= 1) {
      float m = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
      if (__any(m >= 1e30f)) keep += m;
    } else {
      asm volatile("" ::"v"(acc[0]), "v"(acc[15]));
    }
    if (MODE == 2) __syncthreads();
    if (MODE == 3 && (it & 15) == 15) __syncthreads();
    if (MODE == 4 && (it & 127) == 127) __syncthreads();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}
template <int MODE>
void run(const char* name) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512, 34 * 1024>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, 512, 34 * 1024>>>(d, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nmfma = 256.0 * 8 * iters * 128.0;
  printf("%-60s %8.3f ms  %7.1f TFLOP/s\n", name, ms, nmfma * 4096.0 / ms / 1e9);
  hipFree(d);
}
int main() {
  run<0>("8 waves, A from LDS (ds_read_b128 per 4 MFMA)");
  run<1>("  + v_max/any epilogue per 128 MFMA");
  run<2>("  + __syncthreads per 128 MFMA");
  run<3>("  + __syncthreads per 16 x 128 MFMA");
  run<4>("  + __syncthreads per 128 x 128 MFMA");
  return 0;
}
