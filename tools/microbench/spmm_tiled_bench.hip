// Graph-tiled SpMM hop (csrc/sparse.hip spmm_tiled_kernel), standalone: variants of the inner loop against the panel kernel
// of the library, on c2's shape (n = 100 000, degree 11, D = 256 panel-major) -- random columns and columns confined to a
// window (every gather an L2 hit) --, with per-block entry stamps (do the groups of an XCD walk their runs together?).
//
//   hipcc --offload-arch=gfx950 -O3 -I include -o /tmp/spmm_tiled tools/microbench/spmm_tiled_bench.hip \
//         -L ragraph_amd/csrc -lragraph_hip -Wl,-rpath,$PWD/ragraph_amd/csrc
//   /tmp/spmm_tiled [window] [source_block_bytes] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "ragraph_hip.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

struct Plan {
  const int* gp;
  const int* col2;
  const float* val2;
  const unsigned short* rowl2;
  int RG, S, C, SB;
};

// META: 0 = every lane of a group loads the batch's 80 bytes of edge words itself (no shuffles); 1 = lane lr loads edge
// base + lr and the group broadcasts (ds_swizzle).  PF: prefetch the next source block.  STAMP: block-entry times.
template <int META, bool PF, bool STAMP>
__global__ void __launch_bounds__(1024) tiled_kernel(Plan t, int64_t n, int64_t x_rows, const float* __restrict__ Xp, float* __restrict__ Y,
                                                     unsigned long long* __restrict__ stamps) {
  extern __shared__ float4 tile_acc[];
  constexpr int NG = 128;
  const int lr = threadIdx.x & 7, gi = threadIdx.x >> 3;
  const unsigned xcd = blockIdx.x % 8, w = blockIdx.x / 8, wpx = gridDim.x / 8;
  const int RC = t.RG * NG;
  const int64_t gx = (int64_t)w * NG + gi, ngx = (int64_t)wpx * NG;
  float pf_sum = 0.f;
  float4* mine = tile_acc + (gi * t.RG) * 8 + lr;
  const int panel = (int)xcd;
  const float4* X4 = reinterpret_cast<const float4*>(Xp + (int64_t)panel * x_rows * 32) + lr;
  for (int c = (int)w; c < t.C; c += (int)wpx) {
    for (int j = 0; j < t.RG; ++j) mine[j * 8] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int* gpc = t.gp + ((int64_t)c * NG + gi) * t.S;
    const int e0 = gpc[0], e1 = gpc[t.S];
    int pf_s = 0, pf_e = e0;
    int cur = -1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define EDGE(r_, v_, x_)                                                                           \
  do {                                                                                             \
    if ((r_) != cur) {                                                                             \
      if (cur >= 0) tile_acc[cur * 8 + lr] = acc;                                                  \
      acc = tile_acc[(r_) * 8 + lr];                                                               \
      cur = (r_);                                                                                  \
    }                                                                                              \
    acc.x = fmaf((v_), (x_).x, acc.x); acc.y = fmaf((v_), (x_).y, acc.y);                          \
    acc.z = fmaf((v_), (x_).z, acc.z); acc.w = fmaf((v_), (x_).w, acc.w);                          \
  } while (0)
    if (META == 2) {
      // two batches of line gathers in flight: gathers(i + 1) are issued and the edge words of batch i + 2 requested BEFORE
      // batch i is consumed (loads return in order: waiting for batch i's lines leaves the younger ones outstanding)
#define SWZ(v_, k_) __builtin_amdgcn_ds_swizzle((v_), ((k_) << 5) | 0x18)
#define LOADW(c_, v_, r_, b_)                                          \
  do {                                                                 \
    const int me_ = (b_) + lr < e1 ? (b_) + lr : (e1 > e0 ? e1 - 1 : e0);   \
    c_ = t.col2[me_]; v_ = t.val2[me_]; r_ = t.rowl2[me_];             \
  } while (0)
#define GATH(x_, c_)                                                                                        \
  do {                                                                                                      \
    x_[0] = X4[(int64_t)SWZ(c_, 0) * 8]; x_[1] = X4[(int64_t)SWZ(c_, 1) * 8]; x_[2] = X4[(int64_t)SWZ(c_, 2) * 8]; \
    x_[3] = X4[(int64_t)SWZ(c_, 3) * 8]; x_[4] = X4[(int64_t)SWZ(c_, 4) * 8]; x_[5] = X4[(int64_t)SWZ(c_, 5) * 8]; \
    x_[6] = X4[(int64_t)SWZ(c_, 6) * 8]; x_[7] = X4[(int64_t)SWZ(c_, 7) * 8];                                  \
  } while (0)
#define CONS1(x_, v_, r_, k_, cnt_)                                      \
  do {                                                                   \
    const float vk = __int_as_float(SWZ(__float_as_int(v_), k_));        \
    const int rk = SWZ(r_, k_);                                          \
    if ((k_) < (cnt_)) EDGE(rk, vk, x_[k_]);                             \
  } while (0)
#define CONS(x_, v_, r_, cnt_)                                                                                   \
  do {                                                                                                           \
    CONS1(x_, v_, r_, 0, cnt_); CONS1(x_, v_, r_, 1, cnt_); CONS1(x_, v_, r_, 2, cnt_); CONS1(x_, v_, r_, 3, cnt_); \
    CONS1(x_, v_, r_, 4, cnt_); CONS1(x_, v_, r_, 5, cnt_); CONS1(x_, v_, r_, 6, cnt_); CONS1(x_, v_, r_, 7, cnt_); \
  } while (0)
      int cA, rA, cB, rB, cC, rC, cD, rD;
      float vA, vB, vC, vD;
      float4 xA[8], xB[8];
      LOADW(cA, vA, rA, e0);
      LOADW(cB, vB, rB, e0 + 8);
      if (e0 < e1) GATH(xA, cA);
      for (int base = e0; base < e1; base += 16) {
        if (PF) {
          while (pf_s < t.S && base >= pf_e) {
            const int nb = pf_s + 1 < t.S ? pf_s + 1 : 0;
            const int64_t r0 = (int64_t)nb * t.SB, r1 = r0 + t.SB < x_rows ? r0 + t.SB : x_rows;
            for (int64_t r = r0 + gx; r < r1; r += ngx) pf_sum += X4[r * 8].x;
            ++pf_s;
            pf_e = pf_s < t.S ? gpc[pf_s] : INT_MAX;
          }
        }
        // batch A = [base, base + 8) is in flight; issue B = [base + 8, ...), request C's words, consume A
        if (base + 8 < e1) GATH(xB, cB);
        LOADW(cC, vC, rC, base + 16);
        { const int cnt = e1 - base < 8 ? e1 - base : 8; CONS(xA, vA, rA, cnt); }
        if (base + 8 >= e1) break;
        // issue the batch behind B (its words are C's), request the next words, consume B
        if (base + 16 < e1) GATH(xA, cC);
        LOADW(cD, vD, rD, base + 24);
        { const int cnt = e1 - base - 8 < 8 ? e1 - base - 8 : 8; CONS(xB, vB, rB, cnt); }
        cA = cC; vA = vC; rA = rC;
        cB = cD; vB = vD; rB = rD;
      }
#undef SWZ
#undef LOADW
#undef GATH
#undef CONS1
#undef CONS
    } else
    for (int base = e0; base < e1; base += 8) {
      const int cnt = e1 - base < 8 ? e1 - base : 8;
      if (PF || STAMP) {
        while (pf_s < t.S && base >= pf_e) {
          if (STAMP && c == (int)w && gi == 0 && lr == 0) stamps[(int64_t)blockIdx.x * 32 + pf_s] = wall_clock64();
          if (PF) {
            const int nb = pf_s + 1 < t.S ? pf_s + 1 : 0;
            const int64_t r0 = (int64_t)nb * t.SB, r1 = r0 + t.SB < x_rows ? r0 + t.SB : x_rows;
            for (int64_t r = r0 + gx; r < r1; r += ngx) pf_sum += X4[r * 8].x;
          }
          ++pf_s;
          pf_e = pf_s < t.S ? gpc[pf_s] : INT_MAX;
        }
      }
      if (META == 0) {
        const int4 c0 = *reinterpret_cast<const int4*>(t.col2 + base), c1 = *reinterpret_cast<const int4*>(t.col2 + base + 4);
        const float4 v0 = *reinterpret_cast<const float4*>(t.val2 + base), v1 = *reinterpret_cast<const float4*>(t.val2 + base + 4);
        const uint4 rq = *reinterpret_cast<const uint4*>(t.rowl2 + base);
        const int cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const unsigned rw[4] = {rq.x, rq.y, rq.z, rq.w};
        float4 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = X4[(int64_t)(k < cnt ? cc[k] : cc[0]) * 8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < cnt) {
            const int r = (int)((rw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
            EDGE(r, vv[k], x[k]);
          }
      } else if (META == 2) {
        // (handled below the loop header: the pipelined walk)
      } else {
        // lane lr holds edge base + lr (clamped): ONE dword per lane for the columns, one for the values, one ushort for the rows
        const int me = base + (lr < cnt ? lr : 0);
        const int my_c = t.col2[me];
        const float my_v = t.val2[me];
        const int my_r = t.rowl2[me];
        float4 x[8];
#define SWZ(v_, k_) __builtin_amdgcn_ds_swizzle((v_), ((k_) << 5) | 0x18)   /* lane k of every group of 8 */
#define GATHER(k_) x[k_] = X4[(int64_t)SWZ(my_c, k_) * 8]
        GATHER(0); GATHER(1); GATHER(2); GATHER(3); GATHER(4); GATHER(5); GATHER(6); GATHER(7);
#define CONSUME(k_)                                                            \
  do {                                                                         \
    const float vk = __int_as_float(SWZ(__float_as_int(my_v), k_));            \
    const int rk = SWZ(my_r, k_);                                              \
    if ((k_) < cnt) EDGE(rk, vk, x[k_]);                                       \
  } while (0)
        CONSUME(0); CONSUME(1); CONSUME(2); CONSUME(3); CONSUME(4); CONSUME(5); CONSUME(6); CONSUME(7);
#undef SWZ
#undef GATHER
#undef CONSUME
      }
    }
#undef EDGE
    if (cur >= 0) tile_acc[cur * 8 + lr] = acc;
    const int64_t row0 = (int64_t)c * RC + (int64_t)gi * t.RG;
    for (int j = 0; j < t.RG; ++j) {
      if (row0 + j >= n) break;
      float4 v = mine[j * 8];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      reinterpret_cast<float4*>(Y + (int64_t)panel * n * 32 + (row0 + j) * 32)[lr] = v;
    }
  }
  if (pf_sum == 1.2345678e-30f && n < 0) Y[0] = pf_sum;
}

// Wave-interleaved edge words (round 6, second layout): the eight groups of a wave read their batch's words from ONE place --
// col3 / val3 / row3 [wave-batch][64]: lane l = 8 * group + lr holds edge lr of that group's batch -- so a batch costs the L1
// five line requests for its words instead of twenty-four (each group's run was contiguous, the eight runs 400 bytes apart:
// eight lines per load instruction, streamed from HBM, holding the L1's miss slots for ~2 us each).  A group whose run is
// shorter than its wave's longest is padded with NULL edges (row 0xFFFF: gathered from a valid line, never accumulated).
struct Plan3 {
  const int* wp;                 // [C * 16 + 1] first wave-batch of (chunk, wave)
  const int* col3;
  const float* val3;
  const unsigned short* row3;
  int RG, C;
};
template <bool TWO, int ABL = 0>   // ABL (timing only, wrong results): 1 = no LDS parking, 2 = no line gathers, 3 = neither
__global__ void __launch_bounds__(1024) tiled3_kernel(Plan3 t, int64_t n, int64_t x_rows, const float* __restrict__ Xp, float* __restrict__ Y) {
  extern __shared__ float4 tile_acc[];
  constexpr int NG = 128;
  const int lr = threadIdx.x & 7, gi = threadIdx.x >> 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned xcd = blockIdx.x % 8, w = blockIdx.x / 8, wpx = gridDim.x / 8;
  const int RC = t.RG * NG;
  float4* mine = tile_acc + (gi * t.RG) * 8 + lr;
  const int panel = (int)xcd;
  const float4* X4 = reinterpret_cast<const float4*>(Xp + (int64_t)panel * x_rows * 32) + lr;
#define SWZ(v_, k_) __builtin_amdgcn_ds_swizzle((v_), ((k_) << 5) | 0x18)
  for (int c = (int)w; c < t.C; c += (int)wpx) {
    for (int j = 0; j < t.RG; ++j) mine[j * 8] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int b0 = t.wp[c * 16 + wv], b1 = t.wp[c * 16 + wv + 1];
    int cur = -1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define EDGE3(r_, v_, x_)                                                                          \
  do {                                                                                             \
    if ((r_) != 0xFFFF) {                                                                          \
      if (!(ABL & 1) && (r_) != cur) {                                                             \
        if (cur >= 0) tile_acc[cur * 8 + lr] = acc;                                                \
        acc = tile_acc[(r_) * 8 + lr];                                                             \
        cur = (r_);                                                                                \
      }                                                                                            \
      acc.x = fmaf((v_), (x_).x, acc.x); acc.y = fmaf((v_), (x_).y, acc.y);                        \
      acc.z = fmaf((v_), (x_).z, acc.z); acc.w = fmaf((v_), (x_).w, acc.w);                        \
    }                                                                                              \
  } while (0)
#define GATH3(x_, c_)                                                                                       \
  do {                                                                                                      \
    if (ABL & 2) {                                                                                          \
      for (int q_ = 0; q_ < 8; ++q_) x_[q_] = make_float4((float)SWZ(c_, 0), 1.f, 2.f, 3.f);                \
      break;                                                                                                \
    }                                                                                                       \
    x_[0] = X4[(int64_t)SWZ(c_, 0) * 8]; x_[1] = X4[(int64_t)SWZ(c_, 1) * 8]; x_[2] = X4[(int64_t)SWZ(c_, 2) * 8]; \
    x_[3] = X4[(int64_t)SWZ(c_, 3) * 8]; x_[4] = X4[(int64_t)SWZ(c_, 4) * 8]; x_[5] = X4[(int64_t)SWZ(c_, 5) * 8]; \
    x_[6] = X4[(int64_t)SWZ(c_, 6) * 8]; x_[7] = X4[(int64_t)SWZ(c_, 7) * 8];                                  \
  } while (0)
#define CONS3(x_, v_, r_)                                                                         \
  do {                                                                                            \
    { const int rk = SWZ(r_, 0); const float vk = __int_as_float(SWZ(__float_as_int(v_), 0)); EDGE3(rk, vk, x_[0]); } \
    { const int rk = SWZ(r_, 1); const float vk = __int_as_float(SWZ(__float_as_int(v_), 1)); EDGE3(rk, vk, x_[1]); } \
    { const int rk = SWZ(r_, 2); const float vk = __int_as_float(SWZ(__float_as_int(v_), 2)); EDGE3(rk, vk, x_[2]); } \
    { const int rk = SWZ(r_, 3); const float vk = __int_as_float(SWZ(__float_as_int(v_), 3)); EDGE3(rk, vk, x_[3]); } \
    { const int rk = SWZ(r_, 4); const float vk = __int_as_float(SWZ(__float_as_int(v_), 4)); EDGE3(rk, vk, x_[4]); } \
    { const int rk = SWZ(r_, 5); const float vk = __int_as_float(SWZ(__float_as_int(v_), 5)); EDGE3(rk, vk, x_[5]); } \
    { const int rk = SWZ(r_, 6); const float vk = __int_as_float(SWZ(__float_as_int(v_), 6)); EDGE3(rk, vk, x_[6]); } \
    { const int rk = SWZ(r_, 7); const float vk = __int_as_float(SWZ(__float_as_int(v_), 7)); EDGE3(rk, vk, x_[7]); } \
  } while (0)
    if (!TWO) {
      for (int b = b0; b < b1; ++b) {
        const int64_t o = (int64_t)b * 64 + lane;
        const int my_c = t.col3[o];
        const float my_v = t.val3[o];
        const int my_r = t.row3[o];
        float4 x[8];
        GATH3(x, my_c);
        CONS3(x, my_v, my_r);
      }
    } else {
      // three stages: the words of batch b + 2 are requested, THEN the lines of batch b + 1 gathered (its words were requested
      // one iteration ago, before batch b's lines: loads return in order, so waiting for them leaves batch b's lines in
      // flight), then batch b is consumed.  (The word arrays are padded by two wave-batches of NULL edges.)
      // (unrolled by six with compile-time stage indices: no register is copied while its load is in flight)
      float4 xs[2][8];
      int wc[3], wr[3];
      float wv_[3];
      int64_t o = (int64_t)b0 * 64 + lane;
      wc[0] = t.col3[o]; wv_[0] = t.val3[o]; wr[0] = t.row3[o];
      wc[1] = t.col3[o + 64]; wv_[1] = t.val3[o + 64]; wr[1] = t.row3[o + 64];
      if (b0 < b1) GATH3(xs[0], wc[0]);
      for (int b = b0; b < b1; b += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          if (b + u >= b1) break;
          const int64_t o2 = (int64_t)(b + u + 2) * 64 + lane;
          wc[(u + 2) % 3] = t.col3[o2]; wv_[(u + 2) % 3] = t.val3[o2]; wr[(u + 2) % 3] = t.row3[o2];   // words of b + u + 2
          if (b + u + 1 < b1) GATH3(xs[(u + 1) % 2], wc[(u + 1) % 3]);                                  // lines of b + u + 1
          CONS3(xs[u % 2], wv_[u % 3], wr[u % 3]);                                                      // batch b + u
        }
      }
    }
    if (cur >= 0) tile_acc[cur * 8 + lr] = acc;
    const int64_t row0 = (int64_t)c * RC + (int64_t)gi * t.RG;
    for (int j = 0; j < t.RG; ++j) {
      if (row0 + j >= n) break;
      float4 v = mine[j * 8];
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      reinterpret_cast<float4*>(Y + (int64_t)panel * n * 32 + (row0 + j) * 32)[lr] = v;
    }
  }
#undef SWZ
#undef EDGE3
#undef GATH3
#undef CONS3
}

// Fifth form = the third with the group broadcasts on the VALU (two DPP moves each: quad_perm broadcast + row_half_mirror
// into the other quad) instead of ds_swizzle: the LDS pipe, which the parking needs, was carrying 24 swizzles per batch.
template <int K>
__device__ __forceinline__ int bcast8(int v) {   // lane K of every group of 8 lanes
  constexpr int kk = K & 3;
  constexpr int qp = kk | (kk << 2) | (kk << 4) | (kk << 6);
  const int t = __builtin_amdgcn_update_dpp(0, v, qp, 0xF, 0xF, true);       // every quad: its own lane kk
  // the quad that does not hold lane K takes the other quad's value (row_half_mirror = 0x141; bank_mask picks the quads written)
  return __builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, K < 4 ? 0xA : 0x5, false);
}
template <int ABL = 0>
__global__ void __launch_bounds__(1024) tiled5_kernel(Plan3 t, int64_t n, int64_t x_rows, const float* __restrict__ Xp, float* __restrict__ Y) {
  extern __shared__ float4 tile_acc[];
  constexpr int NG = 128;
  const int lr = threadIdx.x & 7, gi = threadIdx.x >> 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned xcd = blockIdx.x % 8, w = blockIdx.x / 8, wpx = gridDim.x / 8;
  const int RC = t.RG * NG;
  float4* mine = tile_acc + (gi * t.RG) * 8 + lr;
  const int panel = (int)xcd;
  const float4* X4 = reinterpret_cast<const float4*>(Xp + (int64_t)panel * x_rows * 32) + lr;
  for (int c = (int)w; c < t.C; c += (int)wpx) {
    for (int j = 0; j < t.RG; ++j) mine[j * 8] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int b0 = t.wp[c * 16 + wv], b1 = t.wp[c * 16 + wv + 1];
    int cur = -1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t o = (int64_t)b0 * 64 + lane;
    int nc = t.col3[o], nr = t.row3[o];
    float nv = t.val3[o];
    for (int b = b0; b < b1; ++b) {
      const int my_c = nc, my_r = nr;
      const float my_v = nv;
      o += 64;
      nc = t.col3[o]; nv = t.val3[o]; nr = t.row3[o];     // the next batch's words (the arrays end in NULL wave-batches)
      float4 x[8];
#define LINE(k_) x[k_] = (ABL & 2) ? make_float4((float)bcast8<k_>(my_c), 1.f, 2.f, 3.f) : X4[(int64_t)bcast8<k_>(my_c) * 8]
      LINE(0); LINE(1); LINE(2); LINE(3); LINE(4); LINE(5); LINE(6); LINE(7);
#undef LINE
#define EDGE5(k_)                                                                                  \
  do {                                                                                             \
    const int r_ = bcast8<k_>(my_r);                                                               \
    const float v_ = __int_as_float(bcast8<k_>(__float_as_int(my_v)));                             \
    if (r_ != 0xFFFF) {                                                                            \
      if (r_ != cur) {                                                                             \
        if (cur >= 0) tile_acc[cur * 8 + lr] = acc;                                                \
        acc = tile_acc[r_ * 8 + lr];                                                               \
        cur = r_;                                                                                  \
      }                                                                                            \
      acc.x = fmaf(v_, x[k_].x, acc.x); acc.y = fmaf(v_, x[k_].y, acc.y);                          \
      acc.z = fmaf(v_, x[k_].z, acc.z); acc.w = fmaf(v_, x[k_].w, acc.w);                          \
    }                                                                                              \
  } while (0)
      EDGE5(0); EDGE5(1); EDGE5(2); EDGE5(3); EDGE5(4); EDGE5(5); EDGE5(6); EDGE5(7);
#undef EDGE5
    }
    if (cur >= 0) tile_acc[cur * 8 + lr] = acc;
    const int64_t row0 = (int64_t)c * RC + (int64_t)gi * t.RG;
    for (int j = 0; j < t.RG; ++j) {
      if (row0 + j >= n) break;
      float4 vv = mine[j * 8];
      vv.x = fmaxf(vv.x, 0.f); vv.y = fmaxf(vv.y, 0.f); vv.z = fmaxf(vv.z, 0.f); vv.w = fmaxf(vv.w, 0.f);
      reinterpret_cast<float4*>(Y + (int64_t)panel * n * 32 + (row0 + j) * 32)[lr] = vv;
    }
  }
}

// Fourth form: the running sums of a batch's rows are READ from LDS up front (eight ds_read_b128 issued with the line gathers,
// one wait for both) instead of a write + read + wait per row change inside the chain -- the chain of round-trips was what
// the third form spent its time in (87 us of its 125 with the gathers ablated).  Every batch is self-contained: its first edge
// starts from LDS, its last parks there.  The plan guarantees that a row appears in a batch as ONE run of consecutive edges
// (a batch is closed early -- NULL edges -- where the next source block would bring a row back).
template <int ABL = 0>
__global__ void __launch_bounds__(1024) tiled4_kernel(Plan3 t, int64_t n, int64_t x_rows, const float* __restrict__ Xp, float* __restrict__ Y) {
  extern __shared__ float4 tile_acc[];
  constexpr int NG = 128;
  const int lr = threadIdx.x & 7, gi = threadIdx.x >> 3, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned xcd = blockIdx.x % 8, w = blockIdx.x / 8, wpx = gridDim.x / 8;
  const int RC = t.RG * NG;
  float4* mine = tile_acc + (gi * t.RG) * 8 + lr;
  const int panel = (int)xcd;
  const float4* X4 = reinterpret_cast<const float4*>(Xp + (int64_t)panel * x_rows * 32) + lr;
#define SWZ(v_, k_) __builtin_amdgcn_ds_swizzle((v_), ((k_) << 5) | 0x18)
  for (int c = (int)w; c < t.C; c += (int)wpx) {
    for (int j = 0; j < t.RG; ++j) mine[j * 8] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int b0 = t.wp[c * 16 + wv], b1 = t.wp[c * 16 + wv + 1];
    int64_t o = (int64_t)b0 * 64 + lane;
    int nc = t.col3[o], nr = t.row3[o];
    float nv = t.val3[o];
    for (int b = b0; b < b1; ++b) {
      const int my_c = nc, my_r = nr;
      const float my_v = nv;
      o += 64;
      nc = t.col3[o]; nv = t.val3[o]; nr = t.row3[o];     // the next batch's words (the arrays end in a NULL wave-batch)
      int r[8];
      float v[8];
      float4 x[8], a[8];
#define ROWVAL(k_) r[k_] = SWZ(my_r, k_); v[k_] = __int_as_float(SWZ(__float_as_int(my_v), k_))
      ROWVAL(0); ROWVAL(1); ROWVAL(2); ROWVAL(3); ROWVAL(4); ROWVAL(5); ROWVAL(6); ROWVAL(7);
#undef ROWVAL
#define LINE(k_) x[k_] = (ABL & 2) ? make_float4((float)SWZ(my_c, k_), 1.f, 2.f, 3.f) : X4[(int64_t)SWZ(my_c, k_) * 8]
      LINE(0); LINE(1); LINE(2); LINE(3); LINE(4); LINE(5); LINE(6); LINE(7);
#undef LINE
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = tile_acc[(r[k] != 0xFFFF ? r[k] : 0) * 8 + lr];
      float4 acc = a[0];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (r[k] != 0xFFFF) {
          if (k > 0 && r[k] != r[k - 1]) {
            tile_acc[r[k - 1] * 8 + lr] = acc;
            acc = a[k];
          }
          acc.x = fmaf(v[k], x[k].x, acc.x); acc.y = fmaf(v[k], x[k].y, acc.y);
          acc.z = fmaf(v[k], x[k].z, acc.z); acc.w = fmaf(v[k], x[k].w, acc.w);
          if (k == 7 || r[k + 1 < 8 ? k + 1 : 7] == 0xFFFF) tile_acc[r[k] * 8 + lr] = acc;
        }
      }
    }
    const int64_t row0 = (int64_t)c * RC + (int64_t)gi * t.RG;
    for (int j = 0; j < t.RG; ++j) {
      if (row0 + j >= n) break;
      float4 vv = mine[j * 8];
      vv.x = fmaxf(vv.x, 0.f); vv.y = fmaxf(vv.y, 0.f); vv.z = fmaxf(vv.z, 0.f); vv.w = fmaxf(vv.w, 0.f);
      reinterpret_cast<float4*>(Y + (int64_t)panel * n * 32 + (row0 + j) * 32)[lr] = vv;
    }
  }
#undef SWZ
}

int main(int argc, char** argv) {
  const int64_t n = 100000;
  const int deg = 11, D = 256, P = 8;
  const int window = argc > 1 ? atoi(argv[1]) : 0;
  const int64_t src_bytes = argc > 2 ? atoll(argv[2]) : (5 << 18);
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  std::mt19937_64 rng(7);
  std::vector<int64_t> rowptr(n + 1);
  std::vector<int32_t> col((size_t)n * deg);
  std::vector<float> val((size_t)n * deg);
  for (int64_t r = 0; r < n; ++r) {
    rowptr[r] = r * deg;
    for (int j = 0; j < deg; ++j) {
      int64_t c = window > 0 ? r + (int64_t)(rng() % (2 * window + 1)) - window : (int64_t)(rng() % n);
      c = std::min<int64_t>(std::max<int64_t>(c, 0), n - 1);
      col[r * deg + j] = (int32_t)c;
      val[r * deg + j] = (float)((rng() % 1000) / 1000.0 + 0.01);
    }
    std::sort(col.begin() + r * deg, col.begin() + (r + 1) * deg);
  }
  rowptr[n] = n * deg;
  const int64_t nnz = n * deg;
  // plan (ragraph_amd/graph.py CSRGraph.tile_plan)
  const int NG = 128, WPX = 32;
  int passes = 1;
  while ((n + (int64_t)WPX * passes * NG - 1) / ((int64_t)WPX * passes * NG) > 9) ++passes;
  const int RG = (int)((n + (int64_t)WPX * passes * NG - 1) / ((int64_t)WPX * passes * NG));
  const int RC = RG * NG, C = (int)((n + RC - 1) / RC);
  const int S = (int)std::max<int64_t>(1, (n * 128 + src_bytes - 1) / src_bytes);
  const int SB = (int)((n + S - 1) / S);
  std::vector<int> gp((size_t)C * NG * S + 1, 0);
  std::vector<int> bucket(nnz);
  for (int64_t r = 0; r < n; ++r)
    for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const int64_t b = ((r / RC) * NG + (r % RC) / RG) * S + col[e] / SB;
      bucket[e] = (int)b;
      gp[b + 1]++;
    }
  for (size_t i = 1; i < gp.size(); ++i) gp[i] += gp[i - 1];
  std::vector<int> pos(gp.begin(), gp.end() - 1);
  std::vector<int32_t> col2(nnz + 16, 0);
  std::vector<float> val2(nnz + 16, 0.f);
  std::vector<unsigned short> rowl2(nnz + 16, 0);
  for (int64_t r = 0; r < n; ++r)
    for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const int p = pos[bucket[e]]++;
      col2[p] = col[e];
      val2[p] = val[e];
      rowl2[p] = (unsigned short)(r % RC);
    }
  printf("n=%lld nnz=%lld window=%d: RG=%d passes=%d C=%d S=%d SB=%d (%lld KiB blocks)\n", (long long)n, (long long)nnz, window, RG, passes, C, S, SB,
         (long long)SB * 128 / 1024);
  int64_t* d_rowptr;
  int32_t *d_col, *d_col2;
  int* d_gp;
  float *d_val, *d_val2, *d_X, *d_Y, *d_Yref;
  unsigned short* d_rowl2;
  unsigned long long* d_stamps;
  CK(hipMalloc(&d_rowptr, (n + 1) * 8));
  CK(hipMalloc(&d_col, nnz * 4));
  CK(hipMalloc(&d_val, nnz * 4));
  CK(hipMalloc(&d_col2, (nnz + 16) * 4));
  CK(hipMalloc(&d_val2, (nnz + 16) * 4));
  CK(hipMalloc(&d_rowl2, (nnz + 16) * 2));
  CK(hipMalloc(&d_gp, gp.size() * 4));
  CK(hipMalloc(&d_X, n * D * 4));
  CK(hipMalloc(&d_Y, n * D * 4));
  CK(hipMalloc(&d_Yref, n * D * 4));
  CK(hipMalloc(&d_stamps, 256 * 32 * 8));
  CK(hipMemcpy(d_rowptr, rowptr.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col2, col2.data(), (nnz + 16) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val2, val2.data(), (nnz + 16) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rowl2, rowl2.data(), (nnz + 16) * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_gp, gp.data(), gp.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> X((size_t)n * D);
  for (auto& v : X) v = (float)((int)(rng() % 2001) - 1000) / 1000.f;
  CK(hipMemcpy(d_X, X.data(), n * D * 4, hipMemcpyHostToDevice));
  Plan t{d_gp, d_col2, d_val2, d_rowl2, RG, S, C, SB};
  const size_t lds = (size_t)RG * 128 * 128;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& fn) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-44s %8.1f us\n", name, ms / reps * 1e3);
  };
  timeit("library panel kernel (panel -> panel)", [&] {
    if (ragraph_spmm_csr_panels_f32(d_rowptr, d_col, d_val, n, d_X, n, 1, D, RAGRAPH_ACT_RELU, 0.f, d_Yref, 1, nullptr) != 0) exit(2);
  });
  std::vector<float> ref((size_t)n * D), got((size_t)n * D);
  CK(hipMemcpy(ref.data(), d_Yref, n * D * 4, hipMemcpyDeviceToHost));
  auto check = [&](const char* name) {
    CK(hipMemcpy(got.data(), d_Y, n * D * 4, hipMemcpyDeviceToHost));
    printf("    %s: %s\n", name, memcmp(ref.data(), got.data(), (size_t)n * D * 4) == 0 ? "bit-identical" : "DIFFERS");
  };
#define RUN(META_, PF_, name_)                                                                                                      \
  do {                                                                                                                              \
    CK(hipFuncSetAttribute((const void*)tiled_kernel<META_, PF_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128)); \
    CK(hipMemset(d_Y, 0, n * D * 4));                                                                                                \
    timeit(name_, [&] { hipLaunchKernelGGL((tiled_kernel<META_, PF_, false>), dim3(256), dim3(1024), lds, 0, t, n, n, d_X, d_Y, d_stamps); }); \
    check(name_);                                                                                                                   \
  } while (0)
  RUN(0, false, "tiled, per-lane edge words");
  RUN(0, true, "tiled, per-lane edge words + prefetch");
  RUN(2, false, "tiled, two batches in flight");
  RUN(2, true, "tiled, two batches in flight + prefetch");
  RUN(1, false, "tiled, lane-distributed + swizzle");
  RUN(1, true, "tiled, lane-distributed + swizzle + prefetch");
  {  // the wave-interleaved layout from the same runs
    std::vector<int> wp((size_t)C * 16 + 1, 0);
    for (int c = 0; c < C; ++c)
      for (int wv = 0; wv < 16; ++wv) {
        int mx = 0;
        for (int g = 0; g < 8; ++g) {
          const size_t gb = ((size_t)c * NG + wv * 8 + g) * S;
          mx = std::max(mx, (gp[gb + S] - gp[gb] + 7) / 8);
        }
        wp[(size_t)c * 16 + wv + 1] = mx;
      }
    for (size_t i = 1; i < wp.size(); ++i) wp[i] += wp[i - 1];
    const size_t nb = (size_t)wp.back();
    std::vector<int32_t> col3(nb * 64 + 256, 0);
    std::vector<float> val3(nb * 64 + 256, 0.f);
    std::vector<unsigned short> row3(nb * 64 + 256, 0xFFFF);
    for (int c = 0; c < C; ++c)
      for (int wv = 0; wv < 16; ++wv)
        for (int g = 0; g < 8; ++g) {
          const size_t gb = ((size_t)c * NG + wv * 8 + g) * S;
          const int e0 = gp[gb], e1 = gp[gb + S];
          const int nbw = wp[(size_t)c * 16 + wv + 1] - wp[(size_t)c * 16 + wv];
          for (int k = 0; k < nbw * 8; ++k) {
            const size_t o = ((size_t)wp[(size_t)c * 16 + wv] + k / 8) * 64 + g * 8 + k % 8;
            if (e0 + k < e1) {
              col3[o] = col2[e0 + k]; val3[o] = val2[e0 + k]; row3[o] = rowl2[e0 + k];
            } else {
              col3[o] = e1 > e0 ? col2[e1 - 1] : 0;   // a valid line (not accumulated)
            }
          }
        }
    int* d_wp; int32_t* d_col3; float* d_val3; unsigned short* d_row3;
    CK(hipMalloc(&d_wp, wp.size() * 4)); CK(hipMalloc(&d_col3, col3.size() * 4)); CK(hipMalloc(&d_val3, val3.size() * 4)); CK(hipMalloc(&d_row3, row3.size() * 2));
    CK(hipMemcpy(d_wp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col3, col3.data(), col3.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val3, val3.data(), val3.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_row3, row3.data(), row3.size() * 2, hipMemcpyHostToDevice));
    Plan3 t3{d_wp, d_col3, d_val3, d_row3, RG, C};
    printf("  wave-interleaved words: %zu wave-batches (%.1f %% padding)\n", nb, 100.0 * ((double)nb * 64 / (double)nnz - 1.0));
    CK(hipFuncSetAttribute((const void*)tiled3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    CK(hipFuncSetAttribute((const void*)tiled3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    CK(hipMemset(d_Y, 0, n * D * 4));
    timeit("tiled, wave-interleaved words", [&] { hipLaunchKernelGGL((tiled3_kernel<false>), dim3(256), dim3(1024), lds, 0, t3, n, n, d_X, d_Y); });
    check("tiled, wave-interleaved words");
    CK(hipFuncSetAttribute((const void*)tiled3_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    CK(hipFuncSetAttribute((const void*)tiled3_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    CK(hipFuncSetAttribute((const void*)tiled3_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    timeit("  ablation: no LDS parking", [&] { hipLaunchKernelGGL((tiled3_kernel<false, 1>), dim3(256), dim3(1024), lds, 0, t3, n, n, d_X, d_Y); });
    timeit("  ablation: no line gathers", [&] { hipLaunchKernelGGL((tiled3_kernel<false, 2>), dim3(256), dim3(1024), lds, 0, t3, n, n, d_X, d_Y); });
    timeit("  ablation: neither", [&] { hipLaunchKernelGGL((tiled3_kernel<false, 3>), dim3(256), dim3(1024), lds, 0, t3, n, n, d_X, d_Y); });
    CK(hipFuncSetAttribute((const void*)tiled5_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    CK(hipFuncSetAttribute((const void*)tiled5_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    CK(hipMemset(d_Y, 0, n * D * 4));
    timeit("tiled5: DPP broadcasts", [&] { hipLaunchKernelGGL((tiled5_kernel<0>), dim3(256), dim3(1024), lds, 0, t3, n, n, d_X, d_Y); });
    check("tiled5: DPP broadcasts");
    timeit("  ablation: no line gathers", [&] { hipLaunchKernelGGL((tiled5_kernel<2>), dim3(256), dim3(1024), lds, 0, t3, n, n, d_X, d_Y); });
    CK(hipMemset(d_Y, 0, n * D * 4));
    timeit("tiled, wave-interleaved, three stages", [&] { hipLaunchKernelGGL((tiled3_kernel<true>), dim3(256), dim3(1024), lds, 0, t3, n, n, d_X, d_Y); });
    check("tiled, wave-interleaved, three stages");
  }
  {  // batches in which a row appears as ONE run (tiled4_kernel)
    std::vector<std::vector<int>> gb((size_t)C * NG);   // per group: edge ids in plan order, -1 = NULL padding, multiples of 8
    for (size_t g = 0; g < (size_t)C * NG; ++g) {
      auto& out = gb[g];
      const int e0 = gp[g * S], e1 = gp[g * S + S];
      int start = 0;   // index in `out` where the current batch starts
      for (int e = e0; e < e1; ++e) {
        bool close = (int)out.size() - start == 8;
        if (!close)
          for (int q = start; q < (int)out.size(); ++q)
            if (rowl2[out[q]] == rowl2[e] && rowl2[out.back()] != rowl2[e]) { close = true; break; }
        if (close) {
          while ((int)out.size() - start < 8) out.push_back(-1);
          start = (int)out.size();
        }
        out.push_back(e);
      }
      while (out.size() % 8) out.push_back(-1);
    }
    std::vector<int> wp((size_t)C * 16 + 1, 0);
    for (int c = 0; c < C; ++c)
      for (int wv = 0; wv < 16; ++wv) {
        size_t mx = 0;
        for (int g = 0; g < 8; ++g) mx = std::max(mx, gb[(size_t)c * NG + wv * 8 + g].size() / 8);
        wp[(size_t)c * 16 + wv + 1] = (int)mx;
      }
    for (size_t i = 1; i < wp.size(); ++i) wp[i] += wp[i - 1];
    const size_t nb = (size_t)wp.back();
    std::vector<int32_t> col3(nb * 64 + 256, 0);
    std::vector<float> val3(nb * 64 + 256, 0.f);
    std::vector<unsigned short> row3(nb * 64 + 256, 0xFFFF);
    for (int c = 0; c < C; ++c)
      for (int wv = 0; wv < 16; ++wv)
        for (int g = 0; g < 8; ++g) {
          const auto& src = gb[(size_t)c * NG + wv * 8 + g];
          const int nbw = wp[(size_t)c * 16 + wv + 1] - wp[(size_t)c * 16 + wv];
          int last_col = 0;
          for (int k = 0; k < nbw * 8; ++k) {
            const size_t o = ((size_t)wp[(size_t)c * 16 + wv] + k / 8) * 64 + g * 8 + k % 8;
            const int e = k < (int)src.size() ? src[k] : -1;
            if (e >= 0) {
              col3[o] = col2[e]; val3[o] = val2[e]; row3[o] = rowl2[e];
              last_col = col2[e];
            } else {
              col3[o] = last_col;
            }
          }
        }
    int* d_wp; int32_t* d_col3; float* d_val3; unsigned short* d_row3;
    CK(hipMalloc(&d_wp, wp.size() * 4)); CK(hipMalloc(&d_col3, col3.size() * 4)); CK(hipMalloc(&d_val3, val3.size() * 4)); CK(hipMalloc(&d_row3, row3.size() * 2));
    CK(hipMemcpy(d_wp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col3, col3.data(), col3.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val3, val3.data(), val3.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_row3, row3.data(), row3.size() * 2, hipMemcpyHostToDevice));
    Plan3 t4{d_wp, d_col3, d_val3, d_row3, RG, C};
    printf("  run-safe batches: %zu wave-batches (%.1f %% padding)\n", nb, 100.0 * ((double)nb * 64 / (double)nnz - 1.0));
    CK(hipFuncSetAttribute((const void*)tiled4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    CK(hipFuncSetAttribute((const void*)tiled4_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
    CK(hipMemset(d_Y, 0, n * D * 4));
    timeit("tiled4: sums read up front", [&] { hipLaunchKernelGGL((tiled4_kernel<0>), dim3(256), dim3(1024), lds, 0, t4, n, n, d_X, d_Y); });
    check("tiled4: sums read up front");
    timeit("  ablation: no line gathers", [&] { hipLaunchKernelGGL((tiled4_kernel<2>), dim3(256), dim3(1024), lds, 0, t4, n, n, d_X, d_Y); });
  }
  // stamps: when does group 0 of every workgroup enter each block of its FIRST chunk?
  CK(hipFuncSetAttribute((const void*)tiled_kernel<1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 128 * 128));
  CK(hipMemset(d_stamps, 0, 256 * 32 * 8));
  hipLaunchKernelGGL((tiled_kernel<1, true, true>), dim3(256), dim3(1024), lds, 0, t, n, n, d_X, d_Y, d_stamps);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st(256 * 32);
  CK(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull;
  for (int b = 0; b < 256; ++b)
    if (st[b * 32]) t0 = std::min(t0, st[b * 32]);
  printf("  block-entry times of group 0 of the 32 workgroups of XCD 0, first chunk (us since the first entry; min .. max per block):\n   ");
  for (int s = 0; s < S && s < 32; ++s) {
    double lo = 1e30, hi = -1;
    for (int wgi = 0; wgi < 32; ++wgi) {
      const unsigned long long v = st[(wgi * 8) * 32 + s];
      if (!v) continue;
      const double us = (double)(v - t0) / 100.0;
      lo = std::min(lo, us);
      hi = std::max(hi, us);
    }
    printf(" b%d %.1f..%.1f", s, lo, hi);
  }
  printf("\n");
  return 0;
}
