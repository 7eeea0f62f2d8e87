// Microbenchmark 3: the top-k kernel's inner pattern with the A-fragment reads software-pipelined by hand
// (ds_read_b128 pair c+1 issued before the 8 MFMAs of pair c; explicit s_waitcnt) vs hipcc's own schedule
// (read pair, wait, 8 MFMAs).  8 waves per workgroup, one workgroup per CU, barrier + v_max epilogue per tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
// wait until at most n LDS reads are outstanding; naming the fragments keeps their MFMAs behind the wait
#define WAIT_LGKM(n, a, b) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b))

template <int MODE, int RANDOM>  // 0: hipcc schedule; 1: hand pipelined one pair ahead; RANDOM: pseudo-random operands
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, float a) {
  extern __shared__ float4 smem4[];
  float* smem = (float*)smem4;
  const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  for (int i = threadIdx.x; i < 32 * 260; i += 512) {
    unsigned hsh = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    smem[i] = RANDOM ? ((int)(hsh & 0xFFFFFF) - 8388608) * (1.f / 8388608.f) * 0.0625f : a + i * 1e-7f;
  }
  __syncthreads();
  float breg[128];
#pragma unroll
  for (int m = 0; m < 128; ++m) {
    unsigned hsh = (unsigned)(threadIdx.x * 131 + m) * 2654435761u;
    hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
    breg[m] = RANDOM ? ((int)(hsh & 0xFFFFFF) - 8388608) * (1.f / 8388608.f) * 0.0625f : a + m * 1e-6f + lane * 1e-7f;
  }
  float keep = 0.f;
  const float* arow = smem + j * 260 + h * 128;
  const unsigned aaddr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)arow;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, breg[4 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, breg[4 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, breg[4 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, breg[4 * c + 3], acc, 0, 0, 0);
      }
    } else {
      f32x4 f[2 * (MODE + 1)];  // (MODE+1) pairs in flight
#pragma unroll
      for (int pp = 0; pp < MODE; ++pp) {
        DS_READ(f[2 * pp], aaddr, 0);
        DS_READ(f[2 * pp + 1], aaddr, 0);
      }
      // fix the offsets of the prologue reads (constant offsets need literals)
#pragma unroll
      for (int c = 0; c < 16; ++c) {  // pair c = chunks 2c, 2c+1
        const int cur = c % (MODE + 1), nxt = (c + MODE) % (MODE + 1);
        if (c + MODE < 16) {
          switch (c + MODE) {
#define CASE(n) case n: DS_READ(f[2 * nxt], aaddr, 32 * n); DS_READ(f[2 * nxt + 1], aaddr, 32 * n + 16); break;
            CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14) CASE(15)
#undef CASE
          }
          if (MODE == 1) WAIT_LGKM(2, f[2 * cur], f[2 * cur + 1]); else WAIT_LGKM(4, f[2 * cur], f[2 * cur + 1]);
        } else {
          if (MODE == 2 && c + 1 < 16) WAIT_LGKM(2, f[2 * cur], f[2 * cur + 1]); else WAIT_LGKM(0, f[2 * cur], f[2 * cur + 1]);
        }
        const f32x4 v0 = f[2 * cur], v1 = f[2 * cur + 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.x, breg[8 * c], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.y, breg[8 * c + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.z, breg[8 * c + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v0.w, breg[8 * c + 3], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.x, breg[8 * c + 4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.y, breg[8 * c + 5], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.z, breg[8 * c + 6], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1.w, breg[8 * c + 7], acc, 0, 0, 0);
      }
    }
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
    if (__any(m >= 1e30f)) keep += m;
    __syncthreads();
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}
template <int MODE, int RANDOM>
void run(const char* name) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k<MODE, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, RANDOM><<<256, 512, 34 * 1024>>>(d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, RANDOM><<<256, 512, 34 * 1024>>>(d, iters, 1.0001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nmfma = 256.0 * 8 * iters * 128.0;
  printf("%-60s %8.3f ms  %7.1f TFLOP/s\n", name, ms, nmfma * 4096.0 / ms / 1e9);
  hipFree(d);
}
int main() {
  run<0, 0>("hipcc schedule (read pair, wait, 8 MFMA)");
  run<1, 0>("hand pipelined, one pair ahead");
  run<0, 0>("hipcc schedule (repeat)");
  run<1, 0>("hand pipelined, one pair ahead (repeat)");
  run<0, 1>("hipcc schedule, RANDOM operands");
  run<1, 1>("hand pipelined, RANDOM operands");
  return 0;
}
