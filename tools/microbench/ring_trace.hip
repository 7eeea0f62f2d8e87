// Timeline of the top-k kernel's ring (burst DMA + FULL/FREE counters) on one CU: per wave and stage, s_memtime stamps
// at the phase boundaries, plus the wave's SIMD id, to see how the two waves of a SIMD share the MFMA pipe.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma_row(const float* g, unsigned lds_dst, unsigned lane16) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(lane16), "s"(lds_dst), "s"(g) : "memory");
}
__device__ __forceinline__ void ring_wait(unsigned* ctr, unsigned target) {
  while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void ring_signal(unsigned* ctr, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
constexpr int T0 = 400, NT = 12, NS = 6;
__global__ void __launch_bounds__(512, 2) k(const float* __restrict__ keys, int64_t nkeys, float* out, int iters, float a,
                                            unsigned long long* trace, unsigned* hwid) {
  extern __shared__ float4 smem4[];
  float* smem = (float*)smem4;
  constexpr int ROW = 260, STAGE = 32 * ROW;
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4 * STAGE; i += 512) smem[i] = a + i * 1e-7f;
  __syncthreads();
  float breg[128];
#pragma unroll
  for (int m = 0; m < 128; ++m) breg[m] = a + m * 1e-6f + lane * 1e-7f;
  float keep = 0.f;
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)smem;
  const unsigned lane16 = 16u * lane;
  unsigned* full = reinterpret_cast<unsigned*>(smem + 4 * STAGE);
  unsigned* freec = full + 4;
  if (threadIdx.x < 8) full[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < 3) full[threadIdx.x] = 8;
  __syncthreads();
  if (blockIdx.x == 0 && lane == 0) hwid[wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
  int pending = -1;
  for (int it = 0; it < iters; ++it) {
    const bool tr = blockIdx.x == 0 && it >= T0 && it < T0 + NT;
    unsigned long long* tp = trace + ((size_t)wave * NT + (it - T0)) * NS;
    const int slot = it & 3;
    const float* arow = smem + slot * STAGE + j * ROW + h * 128;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int64_t key0 = ((int64_t)it * 32) % (nkeys - 32);
    const int ws = (it + 3) & 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    ring_wait(full + slot, 8u * ((it >> 2) + 1));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, breg[4 * c], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, breg[4 * c + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, breg[4 * c + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, breg[4 * c + 3], acc, 0, 0, 0);
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
    if (__any(m >= 1e30f)) keep += m;
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    ring_signal(freec + slot, lane);
    if (pending >= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ring_signal(full + pending, lane);
    }
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    ring_wait(freec + ws, 8u * ((it + 3) >> 2));
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 4 + i;
      dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
    }
    pending = ws;
    if (tr && lane == 0) { tp[0] = t0; tp[1] = t1; tp[2] = t2; tp[3] = t3; tp[4] = t4; tp[5] = t5; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}
int main() {
  const int64_t nkeys = 1000000;
  float *keys, *d;
  unsigned long long* trace; unsigned* hwid;
  hipMalloc(&keys, nkeys * 256 * 4); hipMemset(keys, 0, nkeys * 256 * 4);
  hipMalloc(&d, 256 * 512 * 4);
  hipMalloc(&trace, 8 * NT * NS * 8); hipMalloc(&hwid, 32);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = 4 * 32 * 260 * 4 + 20480;
  k<<<256, 512, lds>>>(keys, nkeys, d, 1000, 1.0f, trace, hwid);
  hipDeviceSynchronize();
  unsigned long long h[8 * NT * NS]; unsigned hw[8];
  hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hw, hwid, sizeof(hw), hipMemcpyDeviceToHost);
  unsigned long long base = ~0ull;
  for (int i = 0; i < 8 * NT * NS; ++i) if (h[i] && h[i] < base) base = h[i];
  for (int w = 0; w < 8; ++w) {
    printf("wave %d  hw_id=0x%08x simd=%u wave_slot=%u\n", w, hw[w], (hw[w] >> 4) & 3, hw[w] & 15);
    for (int it = 0; it < NT; ++it) {
      unsigned long long* t = h + ((size_t)w * NT + it) * NS;
      printf("   stage %d: wait_full %6llu..%6llu  mfma ..%6llu (%5llu)  epi ..%6llu (%4llu)  signal ..%6llu (%4llu)  wait_free ..%6llu (%4llu)\n",
             T0 + it, t[0] - base, t[1] - base, t[2] - base, t[2] - t[1], t[3] - base, t[3] - t[2], t[4] - base, t[4] - t[3],
             t[5] - base, t[5] - t[4]);
    }
  }
  return 0;
}
