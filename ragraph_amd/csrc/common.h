// Shared host/device helpers of libragraph_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <limits.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>

#include "ragraph_hip.h"

namespace ragraph {

// ---- error reporting (thread-local message, see ragraph_last_error) -------------------------------------------
void set_error(const char* fmt, ...);

#define RG_REQUIRE(cond, code, ...)      \
  do {                                   \
    if (!(cond)) {                       \
      ::ragraph::set_error(__VA_ARGS__); \
      return (code);                     \
    }                                    \
  } while (0)

// Checks the launch that was just issued.  hipGetLastError does not synchronise, so this is capture-safe.
#define RG_CHECK_LAUNCH(name)                                                            \
  do {                                                                                   \
    hipError_t e__ = hipGetLastError();                                                  \
    if (e__ != hipSuccess) {                                                             \
      ::ragraph::set_error("%s: launch failed: %s", (name), hipGetErrorString(e__));     \
      return RAGRAPH_EDEVICE;                                                            \
    }                                                                                    \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }


// ---- per-device state ------------------------------------------------------------------------------------------
// A process may drive several devices (hipSetDevice between calls): function attributes and the CU count belong to the
// device that is current when a call is made, so the once-flags are kept per device and are atomic (calls may come
// from several host threads; racing setters write the same value).
constexpr int RG_MAX_DEVICES = 64;
static inline int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= RG_MAX_DEVICES) d = 0;
  return d;
}
struct DeviceOnce {
  std::atomic<unsigned char> done[RG_MAX_DEVICES];
};
// Raises the kernel's dynamic-LDS limit once per device.
template <typename KernelT>
static inline hipError_t raise_dynamic_lds(DeviceOnce& once, KernelT kernel, int bytes) {
  const int d = current_device();
  if (once.done[d].load(std::memory_order_acquire)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) once.done[d].store(1, std::memory_order_release);
  return e;
}
// CUs of the current device rounded down to a multiple of 8 (one persistent workgroup per CU, dealt over the 8 XCDs);
// 256 when no device is visible (workspace / plan queries on a CPU box).  RAGRAPH_TOPK_CUS overrides it (test hook).
static inline int device_cus_multiple_of_8() {
  static const int env = [] {
    const char* e = getenv("RAGRAPH_TOPK_CUS");
    const int v = e ? atoi(e) : 0;
    return v >= 8 ? v / 8 * 8 : 0;
  }();
  if (env) return env;
  static std::atomic<int> cached[RG_MAX_DEVICES];
  const int d = current_device();
  int n = cached[d].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 8) n = 256;
    n = n / 8 * 8;
    cached[d].store(n, std::memory_order_relaxed);
  }
  return n;
}

// ---- device helpers ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RG_NEG_INF (-__builtin_huge_valf())
#define RG_IDX_NONE INT_MAX

// Canonical top-k order: score descending, then index ascending.  Strict total order when indices are unique.
__device__ __forceinline__ bool cand_better(float s1, int i1, float s2, int i2) {
  return (s1 > s2) || (s1 == s2 && i1 < i2);
}

// Epilogue activations (selectors in ragraph_hip.h).
__device__ __forceinline__ float apply_act(float x, int act, float alpha) {
  switch (act) {
    case RAGRAPH_ACT_RELU: return x > 0.f ? x : 0.f;                    // F.relu (finite inputs by contract)
    case RAGRAPH_ACT_PRELU:                                             // PReLU / LeakyReLU: x >= 0 ? x : a*x
    case RAGRAPH_ACT_LEAKY: return x >= 0.f ? x : __fmul_rn(alpha, x);
    case RAGRAPH_ACT_ELU: return x > 0.f ? x : __fmul_rn(alpha, expm1f(x));
    default: return x;
  }
}

// ---- wave-wide reductions without LDS (gfx9 wave64 DPP) ----------------------------------------------------------------
// __shfl_xor is a ds_bpermute_b32 -- an LDS round trip per step, six dependent ones per butterfly.  Inside a row of 16 lanes the
// same exchanges are DPP operands of the adding instruction itself.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL,
                                                               ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned src) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, 0xF, false);
}
constexpr int DPP_QUAD_XOR1 = 0xB1, DPP_QUAD_XOR2 = 0x4E;            // quad_perm [1,0,3,2] / [2,3,0,1]
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_ROR4 = 0x124, DPP_ROW_ROR8 = 0x128;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
// Sum over the 64 lanes in ANY association (error norms that carry their own slack): row sums by shifts, rows joined by the
// two broadcasts, the total read from lane 63 -- every lane gets it.
__device__ __forceinline__ float wave_sum_any_order(float v) {
  v = __fadd_rn(v, dpp_f32<DPP_ROW_SHR1>(0.f, v));
  v = __fadd_rn(v, dpp_f32<DPP_ROW_SHR2>(0.f, v));
  v = __fadd_rn(v, dpp_f32<DPP_ROW_SHR4>(0.f, v));
  v = __fadd_rn(v, dpp_f32<DPP_ROW_SHR8>(0.f, v));
  v = __fadd_rn(v, dpp_f32<DPP_ROW_BCAST15, 0xA>(0.f, v));
  v = __fadd_rn(v, dpp_f32<DPP_ROW_BCAST31, 0xC>(0.f, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, dpp_u32<DPP_ROW_SHR1>(0u, v));
  v = max(v, dpp_u32<DPP_ROW_SHR2>(0u, v));
  v = max(v, dpp_u32<DPP_ROW_SHR4>(0u, v));
  v = max(v, dpp_u32<DPP_ROW_SHR8>(0u, v));
  v = max(v, dpp_u32<DPP_ROW_BCAST15, 0xA>(0u, v));
  v = max(v, dpp_u32<DPP_ROW_BCAST31, 0xC>(0u, v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// The FIXED 64-lane tree of the row norms (butterfly over xor 32, 16, 8, 4, 2, 1: DESIGN.md section 2), bit for bit: after the
// levels 32 and 16 (two bpermutes) every lane of a class i mod 16 holds the same partial sum, so the partner at xor 8 / xor 4 can
// be ANY lane of the partner's class -- the row rotations by 8 and 4 pick one -- and a + b = b + a exactly.
__device__ __forceinline__ float wave_sum_fixed_tree(float p) {
  p = __fadd_rn(p, __shfl_xor(p, 32));
  p = __fadd_rn(p, __shfl_xor(p, 16));
  p = __fadd_rn(p, dpp_f32<DPP_ROW_ROR8>(0.f, p));
  p = __fadd_rn(p, dpp_f32<DPP_ROW_ROR4>(0.f, p));
  p = __fadd_rn(p, dpp_f32<DPP_QUAD_XOR2>(0.f, p));
  p = __fadd_rn(p, dpp_f32<DPP_QUAD_XOR1>(0.f, p));
  return p;
}

}  // namespace ragraph
