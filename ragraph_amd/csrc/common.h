// Shared host/device helpers of libragraph_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <limits.h>
#include <math.h>
#include <string.h>

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

}  // namespace ragraph
