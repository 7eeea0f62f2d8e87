// Microbenchmark 5: the bf16 filter kernel's inner pattern: v_mfma_f32_32x32x16_bf16, two accumulator chains (two
// groups of 32 queries) sharing one A fragment (8 bf16 of a key row) per k-step, 16 k-steps per 32-key sub-tile.
//   MODE 0: A fragments constant in registers (pure MFMA issue rate, B operands = 128 VGPRs as in the kernel)
//   MODE 1: A fragments from LDS, hipcc's schedule            MODE 2: + v_max epilogue per sub-tile
//   MODE 3: A from LDS, asm reads 4 steps ahead + epilogue    MODE 4: MODE 3 with ONE wave per SIMD (256 threads)
//   MODE 5: 2 x 2 register blocking (two 32-key sub-tiles x two query groups, four accumulators) in snake order, so that
//           consecutive MFMAs differ in ONE operand only -- does less operand switching buy clock on random data?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

template <int MODE, int RANDOM>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, float a) {
  extern __shared__ float4 smem4[];
  char* smem = (char*)smem4;
  const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  // RANDOM != 0: operands are pseudo-random bf16 in [-1, 1) (what real unit rows look like to the data path: the chip
  // holds a lower clock on random bits than on near-constant ones)
  for (int i = threadIdx.x; i < 32768 / 2; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float v = RANDOM ? ((int)(h & 0xFFFF) - 32768) * (1.f / 32768.f) : a + i * 1e-7f;
    ((__bf16*)smem)[i] = (__bf16)v;
  }
  __syncthreads();
  bf16x8 b0[16], b1[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned h = (unsigned)(threadIdx.x * 131 + t * 17 + e) * 2654435761u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      const float v = ((int)(h & 0xFFFF) - 32768) * (1.f / 32768.f) * 0.0625f;
      b0[t][e] = RANDOM ? (__bf16)v : (__bf16)(a + t * 1e-3f + e * 1e-2f + lane * 1e-4f);
      b1[t][e] = RANDOM ? (__bf16)(-v * 0.7f + 0.001f * e) : (__bf16)(a - t * 1e-3f + lane * 1e-4f);
    }
  float keep = 0.f;
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)smem;
  unsigned addr[8];
  const unsigned c0 = (unsigned)(g ^ (j & 15));
#pragma unroll
  for (int i = 0; i < 8; ++i) addr[i] = lds_base + (unsigned)j * 512 + (((unsigned)(2 * i) ^ c0) << 4);
  bf16x8 areg[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) areg[i][e] = (__bf16)(a + i + e * 0.1f);
  for (int it = 0; it < iters; ++it) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    if (MODE == 0) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[t & 3], b0[t], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[t & 3], b1[t], acc1, 0, 0, 0);
      }
    } else if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(smem + j * 512 + (((unsigned)(2 * t + g) ^ (unsigned)(j & 15)) << 4));
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b0[t], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b1[t], acc1, 0, 0, 0);
      }
    } else if (MODE < 5) {
      f32x4 fr[4];
#define FREAD(n_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(n_)&3]) : "v"(addr[(n_)&7]), "i"(((n_) >= 8) ? 256 : 0))
#define FWAIT(c_, n_) asm volatile("s_waitcnt lgkmcnt(" #c_ ")" : "+v"(fr[(n_)&3]))
#define FSTEP(n_)                                                                          \
  {                                                                                        \
    if constexpr ((n_) + 3 < 16) FWAIT(3, n_);                                             \
    else if constexpr ((n_) + 2 < 16) FWAIT(2, n_);                                        \
    else if constexpr ((n_) + 1 < 16) FWAIT(1, n_);                                        \
    else FWAIT(0, n_);                                                                     \
    const bf16x8 a_ = __builtin_bit_cast(bf16x8, fr[(n_)&3]);                              \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b0[n_], acc0, 0, 0, 0);             \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b1[n_], acc1, 0, 0, 0);             \
    if constexpr ((n_) + 4 < 16) FREAD((n_) + 4);                                          \
  }
      FREAD(0); FREAD(1); FREAD(2); FREAD(3);
      FSTEP(0) FSTEP(1) FSTEP(2) FSTEP(3) FSTEP(4) FSTEP(5) FSTEP(6) FSTEP(7)
      FSTEP(8) FSTEP(9) FSTEP(10) FSTEP(11) FSTEP(12) FSTEP(13) FSTEP(14) FSTEP(15)
    }
    if (MODE == 5) {
      f32x16 acc2, acc3;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[r] = acc3[r] = 0.f;
      f32x4 fa[4], fb[4];  // sub-tile 0 / sub-tile 1 fragments, two steps in flight
#define PREAD(n_)                                                                                                  \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(n_)&1]) : "v"(addr[(n_)&7]), "i"(((n_) >= 8) ? 256 : 0)); \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[(n_)&1]) : "v"(addr[(n_)&7]), "i"(16384 + (((n_) >= 8) ? 256 : 0)))
#define PSTEP(n_)                                                                          \
  {                                                                                        \
    if constexpr ((n_) + 1 < 16) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[(n_)&1]), "+v"(fb[(n_)&1])); \
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[(n_)&1]), "+v"(fb[(n_)&1]));        \
    const bf16x8 a0_ = __builtin_bit_cast(bf16x8, fa[(n_)&1]);                             \
    const bf16x8 a1_ = __builtin_bit_cast(bf16x8, fb[(n_)&1]);                             \
    if constexpr (((n_) & 1) == 0) {                                                       \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b0[n_], acc0, 0, 0, 0);           \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b0[n_], acc2, 0, 0, 0);           \
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b1[n_], acc3, 0, 0, 0);           \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b1[n_], acc1, 0, 0, 0);           \
    } else {                                                                               \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b1[n_], acc1, 0, 0, 0);           \
      acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b1[n_], acc3, 0, 0, 0);           \
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1_, b0[n_], acc2, 0, 0, 0);           \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0_, b0[n_], acc0, 0, 0, 0);           \
    }                                                                                      \
    if constexpr ((n_) + 2 < 16) { PREAD((n_) + 2); }                                      \
  }
      PREAD(0); PREAD(1);
      PSTEP(0) PSTEP(1) PSTEP(2) PSTEP(3) PSTEP(4) PSTEP(5) PSTEP(6) PSTEP(7)
      PSTEP(8) PSTEP(9) PSTEP(10) PSTEP(11) PSTEP(12) PSTEP(13) PSTEP(14) PSTEP(15)
      float m0 = acc0[0], m1 = acc1[0], m2 = acc2[0], m3 = acc3[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) { m0 = fmaxf(m0, acc0[r]); m1 = fmaxf(m1, acc1[r]); m2 = fmaxf(m2, acc2[r]); m3 = fmaxf(m3, acc3[r]); }
      if (__any(m0 >= 1e30f || m1 >= 1e30f || m2 >= 1e30f || m3 >= 1e30f)) keep += m0 + m1 + m2 + m3;
    } else if (MODE >= 2) {
      float m0 = acc0[0], m1 = acc1[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) { m0 = fmaxf(m0, acc0[r]); m1 = fmaxf(m1, acc1[r]); }
      if (__any(m0 >= 1e30f || m1 >= 1e30f)) keep += m0 + m1;
    } else {
      asm volatile("" ::"v"(acc0[0]), "v"(acc0[15]), "v"(acc1[0]), "v"(acc1[15]));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}
template <int MODE, int RANDOM>
void run(const char* name, int threads) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k<MODE, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 140 * 1024;  // one workgroup per CU as in the kernel
  k<MODE, RANDOM><<<256, threads, lds>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE, RANDOM><<<256, threads, lds>>>(d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = 256.0 * (threads / 64) * iters * (MODE == 5 ? 64.0 : 32.0) * 32768.0;
  printf("%-60s %8.3f ms  %7.1f TFLOP/s\n", name, best, flop / best / 1e9);
  hipFree(d);
}
// MODE 6 (separate kernel): ONE wave per SIMD (256 threads, up to 512 registers per lane), FOUR accumulator chains sharing
// each A fragment (B operands: 4 x 16 x 4 = 256 registers): half the LDS reads per MFMA of the product kernel.
template <int RANDOM>
__global__ void __launch_bounds__(256, 1) k4(float* out, int iters, float a) {
  extern __shared__ float4 smem4[];
  char* smem = (char*)smem4;
  const int lane = threadIdx.x & 63, j = lane & 31, g = lane >> 5;
  for (int i = threadIdx.x; i < 32768 / 2; i += blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float v = RANDOM ? ((int)(h & 0xFFFF) - 32768) * (1.f / 32768.f) : a + i * 1e-7f;
    ((__bf16*)smem)[i] = (__bf16)v;
  }
  __syncthreads();
  bf16x8 b[4][16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        unsigned h = (unsigned)(threadIdx.x * 131 + t * 17 + e + c * 7919) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((int)(h & 0xFFFF) - 32768) * (1.f / 32768.f) * 0.0625f;
        b[c][t][e] = RANDOM ? (__bf16)v : (__bf16)(a + t * 1e-3f + e * 1e-2f + lane * 1e-4f + c);
      }
  float keep = 0.f;
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)smem;
  unsigned addr[8];
  const unsigned c0 = (unsigned)(g ^ (j & 15));
#pragma unroll
  for (int i = 0; i < 8; ++i) addr[i] = lds_base + (unsigned)j * 512 + (((unsigned)(2 * i) ^ c0) << 4);
  for (int it = 0; it < iters; ++it) {
    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f32x4 fr[4];
#define QREAD(n_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[(n_)&3]) : "v"(addr[(n_)&7]), "i"(((n_) >= 8) ? 256 : 0))
#define QWAIT(c_, n_) asm volatile("s_waitcnt lgkmcnt(" #c_ ")" : "+v"(fr[(n_)&3]))
#define QSTEP(n_)                                                                          \
  {                                                                                        \
    if constexpr ((n_) + 3 < 16) QWAIT(3, n_);                                             \
    else if constexpr ((n_) + 2 < 16) QWAIT(2, n_);                                        \
    else if constexpr ((n_) + 1 < 16) QWAIT(1, n_);                                        \
    else QWAIT(0, n_);                                                                     \
    const bf16x8 a_ = __builtin_bit_cast(bf16x8, fr[(n_)&3]);                              \
    _Pragma("unroll") for (int c = 0; c < 4; ++c)                                          \
      acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b[c][n_], acc[c], 0, 0, 0);     \
    if constexpr ((n_) + 4 < 16) QREAD((n_) + 4);                                          \
  }
    QREAD(0); QREAD(1); QREAD(2); QREAD(3);
    QSTEP(0) QSTEP(1) QSTEP(2) QSTEP(3) QSTEP(4) QSTEP(5) QSTEP(6) QSTEP(7)
    QSTEP(8) QSTEP(9) QSTEP(10) QSTEP(11) QSTEP(12) QSTEP(13) QSTEP(14) QSTEP(15)
    float m = acc[0][0];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[c][r]);
    if (__any(m >= 1e30f)) keep += m;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}
template <int RANDOM>
void run4(const char* name) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k4<RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 140 * 1024;
  k4<RANDOM><<<256, 256, lds>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k4<RANDOM><<<256, 256, lds>>>(d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = 256.0 * 4 * iters * 64.0 * 32768.0;
  printf("%-60s %8.3f ms  %7.1f TFLOP/s\n", name, best, flop / best / 1e9);
  hipFree(d);
}
int main() {
  run<0, 0>("A in registers, 2 waves/SIMD", 512);
  run<0, 0>("A in registers, 1 wave/SIMD", 256);
  run<1, 0>("A from LDS (hipcc schedule), 2 waves/SIMD", 512);
  run<2, 0>("  + v_max epilogue", 512);
  run<3, 0>("A from LDS, asm reads 4 ahead + epilogue, 2 waves/SIMD", 512);
  run<3, 0>("A from LDS, asm reads 4 ahead + epilogue, 1 wave/SIMD", 256);
  run<3, 1>("the same on RANDOM operands, 2 waves/SIMD", 512);
  run<3, 1>("the same on RANDOM operands, 1 wave/SIMD", 256);
  run<0, 1>("A in registers, RANDOM B operands, 2 waves/SIMD", 512);
  run<5, 0>("2x2 blocking, snake order, near-constant operands", 512);
  run<5, 1>("2x2 blocking, snake order, RANDOM operands", 512);
  run<3, 1>("(again) 1x2 on RANDOM operands", 512);
  run4<0>("1 wave/SIMD, FOUR chains per A fragment, near-constant");
  run4<1>("1 wave/SIMD, FOUR chains per A fragment, RANDOM operands");
  run<3, 1>("(again) 1x2 on RANDOM operands, 2 waves/SIMD", 512);
  return 0;
}
