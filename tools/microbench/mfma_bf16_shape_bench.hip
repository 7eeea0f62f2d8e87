// Microbenchmark 6: which bf16 MFMA shape should the filter kernels use?  The inner pattern of topk_filter_kernel -- A
// fragments (1 KiB blocks in fragment order) read from LDS four steps ahead by asm loads, query operands resident in 128
// VGPRs, a v_max epilogue per sub-tile, two waves per SIMD, RANDOM operands (the chip holds a lower clock on random bits;
// MI355X_MICROARCH.md 'DVFS give-back' item 7: the clock it holds can depend on the MFMA shape) -- built twice at the same
// output tile per wave (64 queries x 32 keys per 16 blocks):
//   SHAPE 0: v_mfma_f32_32x32x16_bf16, one A block feeds 2 MFMAs (two groups of 32 queries), 32 cycles each
//   SHAPE 1: v_mfma_f32_16x16x32_bf16, one A block feeds 4 MFMAs (four groups of 16 queries), 16 cycles each
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

template <int SHAPE, int RANDOM>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, float a) {
  extern __shared__ float4 smem4[];
  char* smem = (char*)smem4;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 2; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float v = RANDOM ? ((int)(h & 0xFFFF) - 32768) * (1.f / 32768.f) : a + i * 1e-7f;
    ((__bf16*)smem)[i] = (__bf16)v;
  }
  __syncthreads();
  bf16x8 b[32];  // 128 VGPRs of query operands either way
#pragma unroll
  for (int t = 0; t < 32; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned h = (unsigned)(threadIdx.x * 131 + t * 17 + e) * 2654435761u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      const float v = ((int)(h & 0xFFFF) - 32768) * (1.f / 32768.f) * 0.0625f;
      b[t][e] = RANDOM ? (__bf16)v : (__bf16)(a + t * 1e-3f + e * 1e-2f + lane * 1e-4f);
    }
  float keep = 0.f;
  const unsigned addr = (unsigned)(size_t)(lds_void*)smem + (unsigned)lane * 16u;
  for (int it = 0; it < iters; ++it) {
    f32x4 fr[4];
#define FREAD(n_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(n_)&3]) : "v"(addr), "n"(((n_)&15) * 1024))
#define FWAIT(n_) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr[(n_)&3]))
    FREAD(0); FREAD(1); FREAD(2); FREAD(3);
    if constexpr (SHAPE == 0) {
      f32x16 acc0, acc1;
#define STEP32(n_)                                                                                 \
  {                                                                                                \
    if constexpr ((n_) == 0) { _Pragma("unroll") for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f; } \
    FWAIT(n_);                                                                                     \
    const bf16x8 a_ = __builtin_bit_cast(bf16x8, fr[(n_)&3]);                                      \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b[n_], acc0, 0, 0, 0);                      \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b[16 + (n_)], acc1, 0, 0, 0);               \
    FREAD((n_) + 4);                                                                               \
  }
      STEP32(0) STEP32(1) STEP32(2) STEP32(3) STEP32(4) STEP32(5) STEP32(6) STEP32(7)
      STEP32(8) STEP32(9) STEP32(10) STEP32(11) STEP32(12) STEP32(13) STEP32(14) STEP32(15)
      float m0 = acc0[0], m1 = acc1[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) { m0 = fmaxf(m0, acc0[r]); m1 = fmaxf(m1, acc1[r]); }
      if (__any(m0 >= 1e30f || m1 >= 1e30f)) keep += m0 + m1;
    } else {
      // 16 blocks = two 16-key sub-tiles x 8 k-steps of 32; four groups of 16 queries: b[g * 8 + t]
      f32x4 acc[4];
      float m[4];
#define STEP16(n_)                                                                                 \
  {                                                                                                \
    if constexpr (((n_) & 7) == 0) { _Pragma("unroll") for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; } \
    FWAIT(n_);                                                                                     \
    const bf16x8 a_ = __builtin_bit_cast(bf16x8, fr[(n_)&3]);                                      \
    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                  \
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, b[g * 8 + ((n_) & 7)], acc[g], 0, 0, 0); \
    FREAD((n_) + 4);                                                                               \
    if constexpr (((n_) & 7) == 7) {                                                               \
      _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                \
        m[g] = fmaxf(fmaxf(acc[g][0], acc[g][1]), fmaxf(acc[g][2], acc[g][3]));                    \
      if (__any(m[0] >= 1e30f || m[1] >= 1e30f || m[2] >= 1e30f || m[3] >= 1e30f)) keep += m[0] + m[1] + m[2] + m[3]; \
    }                                                                                              \
  }
      STEP16(0) STEP16(1) STEP16(2) STEP16(3) STEP16(4) STEP16(5) STEP16(6) STEP16(7)
      STEP16(8) STEP16(9) STEP16(10) STEP16(11) STEP16(12) STEP16(13) STEP16(14) STEP16(15)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}
template <int SHAPE, int RANDOM>
void run(const char* name, int threads) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k<SHAPE, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 140 * 1024;  // one workgroup per CU as in the kernel
  k<SHAPE, RANDOM><<<256, threads, lds>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<SHAPE, RANDOM><<<256, threads, lds>>>(d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = 256.0 * (threads / 64) * iters * 32.0 * 32768.0;  // 64 queries x 32 keys x 256 d x 2 per iteration and wave
  printf("%-66s %8.3f ms  %7.1f TFLOP/s\n", name, best, flop / best / 1e9);
  hipFree(d);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 1>("32x32x16, RANDOM operands, 2 waves/SIMD", 512);
    run<1, 1>("16x16x32, RANDOM operands, 2 waves/SIMD", 512);
    run<0, 1>("32x32x16, RANDOM operands, 1 wave/SIMD", 256);
    run<1, 1>("16x16x32, RANDOM operands, 1 wave/SIMD", 256);
  }
  run<0, 0>("32x32x16, near-constant operands, 2 waves/SIMD", 512);
  run<1, 0>("16x16x32, near-constant operands, 2 waves/SIMD", 512);
  return 0;
}
