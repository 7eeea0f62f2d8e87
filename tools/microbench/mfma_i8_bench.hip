// Microbenchmark 7: would an int8 filter level pay?  v_mfma_i32_16x16x64_i8 issues at twice the bf16 rate per clock (same
// cycles as v_mfma_f32_16x16x32_bf16 at twice the K) and an int8 bank copy halves the LDS bytes per key -- but the chip is
// power-limited under the bf16 loop already (1.71 of 2.52 PFLOP/s on random operands, mfma_bf16_shape_bench.hip), so what
// matters is the rate the SAME inner pattern sustains: A fragments (1-KiB blocks: 16 keys x 64 int8 elements) read from LDS
// four steps ahead, query operands resident in VGPRs (64 queries x 256 elements = 64 VGPRs; 128 queries = 128 VGPRs, which
// bf16 cannot hold), an integer v_max epilogue per 16-key half, two waves per SIMD, random operands.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_i8 tools/microbench/mfma_i8_bench.hip && /tmp/mfma_i8
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

// MODE 0: bf16 16x16x32, 4 query groups (the product's pattern);  1: int8 16x16x64, 4 groups;  2: int8, 8 groups
template <int MODE>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters) {
  extern __shared__ float4 smem4[];
  char* smem = (char*)smem4;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    if (MODE == 0) {  // two random bf16 values of magnitude < 1
      const float v0 = ((int)(h & 0xFFFF) - 32768) * (1.f / 32768.f), v1 = ((int)(h >> 16) - 32768) * (1.f / 32768.f);
      ((__bf16*)smem)[2 * i] = (__bf16)v0;
      ((__bf16*)smem)[2 * i + 1] = (__bf16)v1;
    } else {
      ((unsigned*)smem)[i] = h;  // four random int8 values
    }
  }
  __syncthreads();
  constexpr int NG = MODE == 2 ? 8 : 4;             // query groups of 16 per wave
  constexpr int KS = MODE == 0 ? 8 : 4;             // k-steps per 16-key half at D = 256
  i32x4 b[NG * KS];                                 // 128 VGPRs (bf16, 4 groups), 64 (int8, 4 groups), 128 (int8, 8 groups)
#pragma unroll
  for (int t = 0; t < NG * KS; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h = (unsigned)(threadIdx.x * 131 + t * 17 + e) * 2654435761u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      if (MODE == 0) {
        const __bf16 x0 = (__bf16)(((int)(h & 0xFFFF) - 32768) * (0.0625f / 32768.f)), x1 = (__bf16)(((int)(h >> 16) - 32768) * (0.0625f / 32768.f));
        b[t][e] = (int)((unsigned)__builtin_bit_cast(unsigned short, x0) | ((unsigned)__builtin_bit_cast(unsigned short, x1) << 16));
      } else {
        b[t][e] = (int)h;
      }
    }
  float keep = 0.f;
  const unsigned addr = (unsigned)(size_t)(lds_void*)smem + (unsigned)lane * 16u;
  for (int it = 0; it < iters; ++it) {
    i32x4 fr[4];
#define FREAD(n_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(n_)&3]) : "v"(addr), "n"(((n_)&15) * 1024))
#define FWAIT(n_) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr[(n_)&3]))
    FREAD(0); FREAD(1); FREAD(2); FREAD(3);
    if constexpr (MODE == 0) {
      f32x4 acc[4];
      float m[4];
#define STEPB(n_)                                                                                  \
  {                                                                                                \
    if constexpr (((n_) & 7) == 0) { _Pragma("unroll") for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; } \
    FWAIT(n_);                                                                                     \
    const bf16x8 a_ = __builtin_bit_cast(bf16x8, fr[(n_)&3]);                                      \
    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                  \
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_, __builtin_bit_cast(bf16x8, b[g * 8 + ((n_) & 7)]), acc[g], 0, 0, 0); \
    FREAD((n_) + 4);                                                                               \
    if constexpr (((n_) & 7) == 7) {                                                               \
      _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                \
        m[g] = fmaxf(fmaxf(acc[g][0], acc[g][1]), fmaxf(acc[g][2], acc[g][3]));                    \
      if (__any(m[0] >= 1e30f || m[1] >= 1e30f || m[2] >= 1e30f || m[3] >= 1e30f)) keep += m[0] + m[1] + m[2] + m[3]; \
    }                                                                                              \
  }
      STEPB(0) STEPB(1) STEPB(2) STEPB(3) STEPB(4) STEPB(5) STEPB(6) STEPB(7)
      STEPB(8) STEPB(9) STEPB(10) STEPB(11) STEPB(12) STEPB(13) STEPB(14) STEPB(15)
    } else {
      // 16 blocks = four 16-key halves x 4 k-steps of 64; NG groups of 16 queries: b[g * 4 + t]
      i32x4 acc[NG];
      int m[NG];
#define STEPI(n_)                                                                                  \
  {                                                                                                \
    if constexpr (((n_) & 3) == 0) { _Pragma("unroll") for (int g = 0; g < NG; ++g) acc[g] = i32x4{0, 0, 0, 0}; } \
    FWAIT(n_);                                                                                     \
    _Pragma("unroll") for (int g = 0; g < NG; ++g)                                                 \
      acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fr[(n_)&3], b[g * 4 + ((n_) & 3)], acc[g], 0, 0, 0); \
    FREAD((n_) + 4);                                                                               \
    if constexpr (((n_) & 3) == 3) {                                                               \
      bool any = false;                                                                            \
      _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                             \
        m[g] = max(max(acc[g][0], acc[g][1]), max(acc[g][2], acc[g][3]));                          \
        any = any || m[g] >= 0x7FFFFFF0;                                                           \
      }                                                                                            \
      if (__any(any)) keep += (float)m[0];                                                         \
    }                                                                                              \
  }
      STEPI(0) STEPI(1) STEPI(2) STEPI(3) STEPI(4) STEPI(5) STEPI(6) STEPI(7)
      STEPI(8) STEPI(9) STEPI(10) STEPI(11) STEPI(12) STEPI(13) STEPI(14) STEPI(15)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}
template <int MODE>
void run(const char* name, int threads) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 140 * 1024;  // one workgroup per CU as in the kernel
  k<MODE><<<256, threads, lds>>>(d, 10);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE><<<256, threads, lds>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // per iteration and wave: 16 blocks x NG groups MFMAs of 16 x 16 x (32 | 64) x 2 operations
  const double per_mfma = MODE == 0 ? 16384.0 : 32768.0;
  const double ng = MODE == 2 ? 8 : 4;
  double ops = 256.0 * (threads / 64) * iters * 16.0 * ng * per_mfma;
  // the same work as pairs of (query, key) scores over D = 256: ops / 512
  printf("%-58s %8.3f ms  %7.1f Tops/s  = %6.2f G (query,key) scores/s at D = 256\n", name, best, ops / best / 1e9,
         ops / 512.0 / best / 1e6);
  hipFree(d);
}
int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("bf16 16x16x32, 4 query groups, 2 waves/SIMD", 512);
    run<1>("int8 16x16x64, 4 query groups, 2 waves/SIMD", 512);
    run<2>("int8 16x16x64, 8 query groups, 2 waves/SIMD", 512);
    run<1>("int8 16x16x64, 4 query groups, 1 wave/SIMD", 256);
    run<2>("int8 16x16x64, 8 query groups, 1 wave/SIMD", 256);
  }
  return 0;
}
