// Microbenchmark 4: the top-k kernel's inner pattern (A fragments from LDS, 128 MFMAs per 32-key tile, v_max epilogue)
// with the key stream arriving by LDS-DMA into a 4-slot ring, WITHOUT flags / lists / inserts: what does the data
// movement alone cost the MFMA rate?  MODE selects who issues the 32 row-DMAs of a stage and where.
#include <hip/hip_runtime.h>
#include <stdio.h>
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

__device__ __forceinline__ void ring_signal_relaxed(unsigned* ctr, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#define EARLY_READ(dst, ptr) asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"((unsigned)(size_t)(lds_void*)(ptr)))
#define EARLY_WAIT(dst) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dst))

// MODE 8: burst DMA + counters, flag values prefetched under the MFMAs.  9: spread DMA + prefetched flags.
// 10: as 9 with relaxed signals.
// MODE 6: the kernel's ring: burst DMA + FULL/FREE counters.  7: spread DMA + counters.
// MODE 0: no DMA.  1: every wave issues its 4 rows in a burst after the tile (the kernel's layout).
// 2: every wave issues one row after every 8th chunk.  3: waves 0-3 issue 8 rows each (burst).  4: wave 0 issues all 32.
// 5: as 1 but barrier-synchronised stages (vmcnt(0) + __syncthreads per tile).
template <int MODE>
__global__ void __launch_bounds__(512, 2) k(const float* __restrict__ keys, int64_t nkeys, float* out, int iters, float a) {
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
  if (MODE >= 6) {
    if (threadIdx.x < 8) full[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < 3) full[threadIdx.x] = 8;
    __syncthreads();
  }
  int pending = -1;
  unsigned e_full = 8, e_free = 0;
  // each workgroup streams its own key range (co-resident workgroups of an XCD share it, as in the kernel)
  const int64_t base_key = (int64_t)(blockIdx.x >> 3 & 0) * 0;
  for (int it = 0; it < iters; ++it) {
    const int slot = it & 3;
    const float* arow = smem + slot * STAGE + j * ROW + h * 128;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int64_t key0 = (base_key + (int64_t)it * 32) % (nkeys - 32);
    const int ws = (it + 3) & 3;
    if (MODE == 6 || MODE == 7 || MODE == 11 || MODE >= 12) ring_wait(full + slot, 8u * ((it >> 2) + 1));
    if (MODE == 12) __builtin_amdgcn_s_setprio(0);
    if (MODE == 13) __builtin_amdgcn_s_setprio(3);
    if (MODE >= 8 && MODE <= 10) {  // value prefetched during the previous tile; fall back to spinning if it was too early
      EARLY_WAIT(e_full);
      if (e_full < 8u * ((it >> 2) + 1)) ring_wait(full + slot, 8u * ((it >> 2) + 1));
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
      if (MODE == 11 && c == 31) ring_signal(freec + slot, lane);  // the slot's last read is in flight: release waits for it
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, breg[4 * c], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, breg[4 * c + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, breg[4 * c + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, breg[4 * c + 3], acc, 0, 0, 0);
      if (MODE == 7 && c == 7) ring_wait(freec + ws, 8u * ((it + 3) >> 2));
      if (MODE >= 9 && MODE <= 10 && c == 0) EARLY_READ(e_free, freec + ws);
      if (MODE >= 9 && MODE <= 10 && c == 7) {
        EARLY_WAIT(e_free);
        if (e_free < 8u * ((it + 3) >> 2)) ring_wait(freec + ws, 8u * ((it + 3) >> 2));
      }
      if (MODE == 8 && c == 20) EARLY_READ(e_free, freec + ws);
      if (MODE >= 8 && MODE <= 10 && c == 24) EARLY_READ(e_full, full + ((it + 1) & 3));
      if ((MODE == 2 || MODE == 7 || MODE == 9 || MODE == 10) && (c & 7) == 7) {
        const int row = wave * 4 + (c >> 3);
        dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
      }
    }
    if (MODE == 12) __builtin_amdgcn_s_setprio(3);  // the few LDS / VMEM / scalar instructions of the hand-over first
    if (MODE == 13) __builtin_amdgcn_s_setprio(0);
    float m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
    if (__any(m >= 1e30f)) keep += m;
    if (MODE == 1 || MODE == 5) {
      if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // last iteration's rows
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 4 + i;
        dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
      }
      if (MODE == 5) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (MODE == 6 || MODE == 11 || MODE == 12 || MODE == 13) {
      if (MODE != 11) ring_signal(freec + slot, lane);
      if (pending >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_signal(full + pending, lane);
      }
      ring_wait(freec + ws, 8u * ((it + 3) >> 2));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 4 + i;
        dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
      }
      pending = ws;
    }
    if (MODE == 8) {
      ring_signal(freec + slot, lane);
      if (pending >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_signal(full + pending, lane);
      }
      EARLY_WAIT(e_free);
      if (e_free < 8u * ((it + 3) >> 2)) ring_wait(freec + ws, 8u * ((it + 3) >> 2));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave * 4 + i;
        dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
      }
      pending = ws;
    }
    if (MODE == 10) {
      ring_signal_relaxed(freec + slot, lane);
      if (pending >= 0) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        ring_signal_relaxed(full + pending, lane);
      }
      pending = ws;
    }
    if (MODE == 7 || MODE == 9) {
      ring_signal(freec + slot, lane);
      if (pending >= 0) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        ring_signal(full + pending, lane);
      }
      pending = ws;
    }
    if (MODE == 3 && wave < 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = wave * 8 + i;
        dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
      }
    }
    if (MODE == 4 && wave == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int row = 0; row < 32; ++row)
        dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}

// MODE 14 design: NOTHING between two tiles' MFMA streams.  Per wave and stage s (slot = s & 3):
//   c = 0     : FULL[s] was checked with a value prefetched at c = 28 of stage s-1 (spin only if it was too early)
//   c = 1     : signal FREE[slot of s-1]  (all of this wave's reads of s-1 completed before its last MFMAs were issued)
//   c = 3..   : epilogue of stage s-1 on the OTHER accumulator (v_max3 chain + any), overlapping this stage's MFMAs
//   c = 5     : prefetch FREE[slot of s-1] (the slot stage s+3 goes to); c = 7: check it (spin if needed), DMA row 0
//   c = 15,23,31: DMA rows 1..3 of stage s+3
//   c = 12    : vmcnt(1) (rows of stage s+2, issued during s-1, have landed; only this stage's row 0 may fly), signal FULL[s+2]
//   c = 28    : prefetch FULL[s+1]
template <bool kEarly>
__global__ void __launch_bounds__(512, 2) k14(const float* __restrict__ keys, int64_t nkeys, float* out, int iters, float a) {
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
  unsigned e_full = 8, e_free = 0;
  f32x16 accA, accB;
#pragma unroll
  for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0.f;
  auto stage = [&](int it, f32x16& acc, f32x16& prev) {
    const int slot = it & 3, ws = (it + 3) & 3;
    const float* arow = smem + slot * STAGE + j * ROW + h * 128;
    const int64_t key0 = ((int64_t)it * 32) % (nkeys - 32);
    EARLY_WAIT(e_full);
    if (e_full < 8u * ((it >> 2) + 1)) ring_wait(full + slot, 8u * ((it >> 2) + 1));
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, breg[4 * c], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, breg[4 * c + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, breg[4 * c + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, breg[4 * c + 3], acc, 0, 0, 0);
      if (c == 1 && it > 0) ring_signal_relaxed(freec + ((it - 1) & 3), lane);
      if (c == 3 && it > 0) {
        float m = prev[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, prev[r]);
        if (__any(m >= 1e30f)) keep += m;
      }
      if (c == 5) EARLY_READ(e_free, freec + ws);
      if (c == 7) {
        EARLY_WAIT(e_free);
        if (e_free < 8u * ((it + 3) >> 2)) ring_wait(freec + ws, 8u * ((it + 3) >> 2));
      }
      if ((c & 7) == 7) {
        const int row = wave * 4 + (c >> 3);
        dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
      }
      if (c == 12 && it > 0) {
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        ring_signal_relaxed(full + ((it + 2) & 3), lane);
      }
      if (c == 28) EARLY_READ(e_full, full + ((it + 1) & 3));
    }
  };
  for (int it = 0; it < iters; it += 2) {
    stage(it, accA, accB);
    stage(it + 1, accB, accA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep + accB[3];
}
void run14(const char* name, const float* keys, int64_t nkeys) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k14<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 4 * 32 * 260 * 4 + 20480;
  k14<true><<<256, 512, lds>>>(keys, nkeys, d, 10, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k14<true><<<256, 512, lds>>>(keys, nkeys, d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double nmfma = 256.0 * 8 * iters * 128.0;
  printf("%-64s %8.3f ms  %7.1f TFLOP/s\n", name, best, nmfma * 4096.0 / best / 1e9);
  hipFree(d);
}


// The same ring with v_mfma_f32_16x16x4_f32: two 16-query groups per wave (two independent 4-VGPR accumulator chains
// fed by the same A fragment), two 16-key sub-tiles per 32-key stage.  Same flops and LDS traffic as the 32x32x2 form,
// half the accumulator write bandwidth.  FLAGS = 0: burst DMA only; 1: burst DMA + FULL/FREE counters (kernel ring).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int FLAGS>
__global__ void __launch_bounds__(512, 2) k16(const float* __restrict__ keys, int64_t nkeys, float* out, int iters, float a) {
  extern __shared__ float4 smem4[];
  float* smem = (float*)smem4;
  constexpr int ROW = 260, STAGE = 32 * ROW;
  const int lane = threadIdx.x & 63, j = lane & 15, sl = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4 * STAGE; i += 512) smem[i] = a + i * 1e-7f;
  __syncthreads();
  float b0[64], b1[64];
#pragma unroll
  for (int m = 0; m < 64; ++m) { b0[m] = a + m * 1e-6f + lane * 1e-7f; b1[m] = a - m * 1e-6f + lane * 1e-7f; }
  float keep = 0.f;
  const unsigned lds_base = (unsigned)(size_t)(lds_void*)smem;
  const unsigned lane16 = 16u * lane;
  unsigned* full = reinterpret_cast<unsigned*>(smem + 4 * STAGE);
  unsigned* freec = full + 4;
  if (threadIdx.x < 8) full[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < 3) full[threadIdx.x] = 8;
  __syncthreads();
  int pending = -1;
  for (int it = 0; it < iters; ++it) {
    const int slot = it & 3, ws = (it + 3) & 3;
    const int64_t key0 = ((int64_t)it * 32) % (nkeys - 32);
    if (FLAGS) ring_wait(full + slot, 8u * ((it >> 2) + 1));
    float mx = -1e30f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const float* arow = smem + slot * STAGE + (sub * 16 + j) * ROW + sl * 64;
      f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(arow + 4 * c);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, b0[4 * c], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, b1[4 * c], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, b0[4 * c + 1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, b1[4 * c + 1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, b0[4 * c + 2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, b1[4 * c + 2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, b0[4 * c + 3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, b1[4 * c + 3], acc1, 0, 0, 0);
      }
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(acc0[0], acc0[1]), fmaxf(acc0[2], acc0[3])),
                           fmaxf(fmaxf(acc1[0], acc1[1]), fmaxf(acc1[2], acc1[3]))));
    }
    if (__any(mx >= 1e30f)) keep += mx;
    if (FLAGS) {
      ring_signal(freec + slot, lane);
      if (pending >= 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ring_signal(full + pending, lane);
      }
      ring_wait(freec + ws, 8u * ((it + 3) >> 2));
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave * 4 + i;
      dma_row(keys + (key0 + row) * 256, lds_base + 4u * (ws * STAGE + row * ROW), lane16);
    }
    pending = ws;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
}
template <int FLAGS>
void run16(const char* name, const float* keys, int64_t nkeys) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k16<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 4 * 32 * 260 * 4 + 20480;
  k16<FLAGS><<<256, 512, lds>>>(keys, nkeys, d, 10, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k16<FLAGS><<<256, 512, lds>>>(keys, nkeys, d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flop = 256.0 * 8 * iters * 256.0 * 2048.0;
  printf("%-64s %8.3f ms  %7.1f TFLOP/s\n", name, best, flop / best / 1e9);
  hipFree(d);
}

template <int MODE>
void run(const char* name, const float* keys, int64_t nkeys) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = 4 * 32 * 260 * 4 + 20480;  // ring + the lists' share: one workgroup per CU as in the kernel
  k<MODE><<<256, 512, lds>>>(keys, nkeys, d, 10, 1.0f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<MODE><<<256, 512, lds>>>(keys, nkeys, d, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double nmfma = 256.0 * 8 * iters * 128.0;
  printf("%-64s %8.3f ms  %7.1f TFLOP/s\n", name, best, nmfma * 4096.0 / best / 1e9);
  hipFree(d);
}
int main() {
  const int64_t nkeys = 1000000;
  float* keys;
  hipMalloc(&keys, nkeys * 256 * 4);
  hipMemset(keys, 0, nkeys * 256 * 4);
  run<0>("no DMA", keys, nkeys);
  run<1>("every wave: 4 rows in a burst after the tile", keys, nkeys);
  run<2>("every wave: one row after every 8th chunk", keys, nkeys);
  run<3>("waves 0-3: 8 rows each, burst", keys, nkeys);
  run<4>("wave 0: all 32 rows, burst", keys, nkeys);
  run<5>("as the first, but vmcnt(0) + barrier per tile", keys, nkeys);
  run<6>("kernel ring: burst DMA + FULL/FREE counters", keys, nkeys);
  run<7>("spread DMA + FULL/FREE counters", keys, nkeys);
  run<8>("burst DMA + counters, flags prefetched under the MFMAs", keys, nkeys);
  run<9>("spread DMA + counters, flags prefetched", keys, nkeys);
  run<10>("spread DMA + counters, flags prefetched, relaxed signals", keys, nkeys);
  run<11>("kernel ring, FREE signalled before the last 4 MFMAs", keys, nkeys);
  run<12>("kernel ring, s_setprio 3 outside the MFMA section", keys, nkeys);
  run<13>("kernel ring, s_setprio 3 inside the MFMA section", keys, nkeys);
  run<6>("kernel ring (repeat)", keys, nkeys);
  run14("ring with everything inside the MFMA stream (two accumulators)", keys, nkeys);
  run16<0>("16x16x4 x 2 chains: burst DMA, no counters", keys, nkeys);
  run16<1>("16x16x4 x 2 chains: kernel ring (burst DMA + counters)", keys, nkeys);
  run<0>("no DMA (repeat)", keys, nkeys);
  return 0;
}
